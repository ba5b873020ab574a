"""CPU tier: pins the CPU oracle (oracle/) against
  (1) known-answer vectors from an independent numpy restatement (tests/golden/gen_golden.py),
  (2) the finite-difference convention of the reference's own ProjectionFactor::check()
      (vins_estimator/src/factor/projection_factor.cpp:123-225),
  (3) the MATLAB transcript of createLinearImuMatrices
      (support_files/scripts/createMatricesLinearImuFactor.m:17-101, test_ccT.m:24-36),
  (4) invariants of the solver / marginalization / selector.
The reference ships no tests or golden vectors of its own, so this is the strongest pin available.
"""
import ctypes as C

import numpy as np
import pytest

from helpers import (abi, blank_windows, buffers, golden, imu_window_from_golden, perturb_pose, projection_windows_from_golden, rel,
                     synth)


@pytest.fixture(scope="module")
def opt():
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    return o


def test_projection_factor_matches_golden(oracle, opt):
    g = golden()
    w = projection_windows_from_golden(g)
    raw = oracle.eval_factors(opt, w, apply_loss=False)
    cor = oracle.eval_factors(opt, w, apply_loss=True)
    J = np.concatenate([g["proj_Ji"], g["proj_Jj"], g["proj_Je"][:, :, None]], axis=2)
    assert rel(raw["proj_r"][:, 1], g["proj_r"]) < 1e-12
    assert rel(raw["proj_J"][:, 1], J) < 1e-12
    assert rel(cor["proj_r"][:, 1], g["proj_r_c"]) < 1e-12
    assert rel(cor["proj_J"][:, 1], g["proj_J_c"]) < 1e-12


def test_preintegration_matches_golden(oracle, opt):
    g = golden()
    w = imu_window_from_golden(g)
    d, J, P, sd, sq = oracle.preintegrate(opt, w)
    assert rel(d[0, 0, :3], g["pre_dp"]) < 1e-13
    q = g["pre_dq_wxyz"]
    assert rel(d[0, 0, 3:7], np.array([q[1], q[2], q[3], q[0]])) < 1e-13
    assert rel(d[0, 0, 7:10], g["pre_dv"]) < 1e-13
    assert rel(J[0, 0], g["pre_J"]) < 1e-12
    assert rel(P[0, 0], g["pre_P"]) < 1e-12
    assert abs(sd[0, 0] - g["pre_sum_dt"]) < 1e-15
    # sqrt_info^T sqrt_info == P^-1
    U = sq[0, 0]
    assert rel(U.T @ U @ g["pre_P"], np.eye(15)) < 1e-7
    assert np.allclose(np.tril(U, -1), 0)


def test_imu_residual_matches_golden(oracle, opt):
    g = golden()
    w = imu_window_from_golden(g)
    ev = oracle.eval_factors(opt, w)
    *_, sq = oracle.preintegrate(opt, w)
    assert rel(ev["imu_r"][0, 0], sq[0, 0] @ g["imu_r_raw"]) < 1e-11


def test_linear_imu_matrices_match_matlab_transcript(oracle):
    g = golden()
    Rj = g["lin_Rj"]
    qj = synth.quat_from_R(Rj)
    qi = np.array([0.0, 0, 0, 1.0])
    Om, A = np.zeros((9, 9)), np.zeros((9, 9))
    oracle.lib().avmo_linear_imu_matrices.argtypes = [abi.c_dp, abi.c_dp, C.c_double, C.c_double, C.c_double, C.c_double, abi.c_dp, abi.c_dp]
    oracle.lib().avmo_linear_imu_matrices(abi.dptr(qi), abi.dptr(qj), float(g["lin_n"]), float(g["lin_delta"]), float(g["lin_accVar"]),
                                          float(g["lin_biasVar"]), abi.dptr(Om), abi.dptr(A))
    assert rel(A, g["lin_A"]) < 1e-12
    assert rel(Om, g["lin_Omega"]) < 1e-10
    # test_ccT.m: eigenvalues of the CC^T block incl. the extra nrImu factor on the (1,1) block
    cov = np.linalg.inv(Om)
    assert rel(np.sort(np.linalg.eigvalsh(cov[:6, :6])), g["lin_eig_cct"]) < 1e-8


def test_projection_jacobian_fd_check_convention(oracle, opt):
    w = synth.make_windows(1, tracks="sparse", n_feat=24, max_feat=150)
    base = oracle.eval_factors(opt, w)
    eps, worst = 1e-6, 0.0
    for e in range(0, 24, 5):
        st, nb, no = (int(w.a[k][0, e]) for k in ("feat_start", "feat_obs_begin", "feat_nobs"))
        for t in range(1, no):
            slot = nb + t
            for k in range(6):
                for fr, col in ((st, k), (st + t, 6 + k)):
                    d = (oracle.eval_factors(opt, perturb_pose(w, 0, fr, k, eps))["proj_r"][0, slot] - base["proj_r"][0, slot]) / eps
                    worst = max(worst, np.abs(d - base["proj_J"][0, slot, :, col]).max() / max(1.0, np.abs(d).max()))
            w2 = w.copy()
            w2.a["inv_depth"][0, e] += eps
            d = (oracle.eval_factors(opt, w2)["proj_r"][0, slot] - base["proj_r"][0, slot]) / eps
            worst = max(worst, np.abs(d - base["proj_J"][0, slot, :, 12]).max() / max(1.0, np.abs(d).max()))
    assert worst < 2e-5


def test_imu_jacobian_fd_at_linearization_point(oracle, opt):
    # SURVEY App.A 14b: exact only at Bg_i == linearized_bg, which the generator guarantees
    w = synth.make_windows(1, tracks="sparse", n_feat=4, max_feat=150)
    base = oracle.eval_factors(opt, w)
    worst = 0.0
    for i in (0, 5, 9):
        for k in range(6):
            for fr, c0 in ((i, 0), (i + 1, 15)):
                d = (oracle.eval_factors(opt, perturb_pose(w, 0, fr, k, 1e-6))["imu_r"][0, i] - base["imu_r"][0, i]) / 1e-6
                worst = max(worst, np.abs(d - base["imu_J"][0, i, :, c0 + k]).max() / max(1.0, np.abs(d).max()))
        for k in range(9):
            for fr, c0 in ((i, 6), (i + 1, 21)):
                w2 = w.copy()
                w2.a["speedbias"][0, fr, k] += 1e-8
                d = (oracle.eval_factors(opt, w2)["imu_r"][0, i] - base["imu_r"][0, i]) / 1e-8
                worst = max(worst, np.abs(d - base["imu_J"][0, i, :, c0 + k]).max() / max(1.0, np.abs(d).max()))
    assert worst < 1e-5


def test_eig_sym_matches_numpy(oracle):
    rng = np.random.default_rng(3)
    oracle.lib().avmo_eig_sym.argtypes = [C.c_int, abi.c_dp, abi.c_dp, abi.c_dp]
    for n in (1, 2, 15, 75, 160):
        A = rng.normal(size=(n, n))
        A = A @ A.T + 1e-3 * np.eye(n)
        w, V = np.zeros(n), np.zeros((n, n))
        oracle.lib().avmo_eig_sym(n, abi.dptr(np.ascontiguousarray(A)), abi.dptr(w), abi.dptr(V))
        assert rel(w, np.linalg.eigvalsh(A)) < 1e-12
        assert rel(V @ np.diag(w) @ V.T, A) < 1e-12
        assert rel(V.T @ V, np.eye(n)) < 1e-12


@pytest.mark.parametrize("tracks,nf", [("sparse", 40), ("dense", 30)])
def test_solver_decreases_cost_and_is_deterministic(oracle, opt, tracks, nf):
    w = synth.make_windows(2, tracks=tracks, n_feat=nf, max_feat=150)
    a, b = w.copy(), w.copy()
    sa, sb = buffers.summary_alloc(2), buffers.summary_alloc(2)
    oracle.window_solve(opt, a, None, sa)
    oracle.window_solve(opt, b, None, sb, n_threads=2)
    for k in ("pose", "speedbias", "inv_depth"):
        assert np.array_equal(a.a[k], b.a[k])
    for i in range(2):
        n = sa[i]["num_iterations"]
        tr = np.concatenate([[sa[i]["initial_cost"]], sa[i]["cost_trace"][:n]])
        assert np.all(np.diff(tr) <= 0), tr
        assert sa[i]["final_cost"] < 1e-3 * sa[i]["initial_cost"]
        assert sa[i]["num_successful"] >= 3
    # the solve must move the states
    assert np.abs(a.a["pose"] - w.a["pose"]).max() > 1e-3


def test_non_finite_inputs_do_not_break_the_oracle(oracle, opt):
    """NaN / Inf / singular inputs end in FAILURE after max_num_consecutive_invalid_steps attempts; the dense eigen solver
    of the marginalization stays inside its arrays (the QL scan is bounded by n - 1 also when no comparison is true)."""
    w = synth.make_windows(3, tracks="sparse", n_feat=30, max_feat=150)
    w.a["pose"][0, 3, 0] = np.nan
    w.a["imu_dt"][1, :, :] = 0.0
    o = abi.default_options()
    so, po = buffers.summary_alloc(3), buffers.PriorOutArrays.alloc(3)
    oracle.window_solve(o, w, po, so)
    assert so["termination"].tolist() == [5, 5, 0]
    assert so["num_iterations"].tolist() == [5, 5, 8]
    A = np.full((7, 7), np.nan)
    wv, V = np.zeros(7), np.zeros((7, 7))
    oracle.lib().avmo_eig_sym.argtypes = [C.c_int, abi.c_dp, abi.c_dp, abi.c_dp]
    oracle.lib().avmo_eig_sym(7, abi.dptr(A), abi.dptr(wv), abi.dptr(V))
    assert np.isnan(wv).all()


def _yaw_deg(q):  # Utility::R2ypr(R).x(), utility.h:86-99 (degrees), q = x y z w
    x, y, z, w = q
    r00, r10 = 1 - 2 * (y * y + z * z), 2 * (x * y + z * w)
    return np.degrees(np.arctan2(r10, r00))


def check_gauge_fix(before, after):
    """double2vector (estimator.cpp:521-587): the first frame keeps its yaw and its position; every frame's position and
    velocity is rotated about the first frame's new position by the yaw difference, so distances to frame 0 and speeds
    are the solver's."""
    for b in range(before["pose"].shape[0]):
        assert abs(_yaw_deg(after["pose"][b, 0, 3:]) - _yaw_deg(before["pose"][b, 0, 3:])) < 1e-9
        assert np.abs(after["pose"][b, 0, :3] - before["pose"][b, 0, :3]).max() < 1e-12
        assert np.abs(np.linalg.norm(after["pose"][b, :, 3:], axis=1) - 1).max() < 1e-12


def test_gauge_fix_keeps_first_frame_yaw_and_position(oracle, opt):
    w = synth.make_windows(3, first_id=40, tracks="sparse", n_feat=40, max_feat=150)
    # push the initial guess away so that the solve moves frame 0 before the fix brings it back
    w.a["pose"][:, :, :3] += 0.05
    a = w.copy()
    oracle.window_solve(opt, a, None, buffers.summary_alloc(3))
    assert np.abs(a.a["pose"][:, 1:, :3] - w.a["pose"][:, 1:, :3]).max() > 1e-3
    check_gauge_fix(w.a, a.a)


def test_first_iteration_step_matches_dense_numpy_normal_equations(oracle):
    """Pins the oracle's linear path (Jacobi scaling, LM diagonal, Schur elimination of the inverse depths, Cholesky, the
    Gauss-Newton branch of the dogleg, Plus) against a dense numpy statement of Ceres' first trust-region iteration built
    only from the per-factor residuals / Jacobians (which the golden vectors pin): (H' + mu D^2) y = g' on the full
    315-column system, x <- Plus(x, -S y).  Compared on gauge-invariant quantities, because the post-solve yaw / position
    alignment of double2vector moves the absolute poses."""
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    o.max_num_iterations = 1
    o.initial_trust_region_radius = 1e6   # keeps the Gauss-Newton step inside the region (checked below): no dogleg blend
    w = synth.make_windows(2, tracks="sparse", n_feat=40, max_feat=150)
    f = oracle.eval_factors(o, w, apply_loss=True)
    ws = w.copy()
    summ = buffers.summary_alloc(2)
    oracle.window_solve(o, ws, None, summ)

    def q2R(q):  # x y z w
        x, y, z, ww = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * ww), 2 * (x * z + y * ww)],
                         [2 * (x * y + z * ww), 1 - 2 * (x * x + z * z), 2 * (y * z - x * ww)],
                         [2 * (x * z - y * ww), 2 * (y * z + x * ww), 1 - 2 * (x * x + y * y)]])

    def qmul(a, b):  # x y z w
        ax, ay, az, aw = a
        bx, by, bz, bw = b
        return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by - ax * bz + ay * bw + az * bx,
                         aw * bz + ax * by - ay * bx + az * bw, aw * bw - ax * bx - ay * by - az * bz])

    for b in range(2):
        a = w.a
        nf = int(a["n_feat"][b])
        NF = 165
        rows_J, rows_r = [], []
        for e in range(nf):
            st, no, ob = a["feat_start"][b, e], a["feat_nobs"][b, e], a["feat_obs_begin"][b, e]
            for k in range(1, no):
                J = np.zeros((2, NF + nf))
                Jf = f["proj_J"][b, ob + k]
                J[:, 6 * st:6 * st + 6], J[:, 6 * (st + k):6 * (st + k) + 6], J[:, NF + e] = Jf[:, :6], Jf[:, 6:12], Jf[:, 12]
                rows_J.append(J), rows_r.append(f["proj_r"][b, ob + k])
        for i in range(10):
            J = np.zeros((15, NF + nf))
            Jf = f["imu_J"][b, i]
            J[:, 6 * i:6 * i + 6], J[:, 66 + 9 * i:66 + 9 * i + 9] = Jf[:, :6], Jf[:, 6:15]
            J[:, 6 * (i + 1):6 * (i + 1) + 6], J[:, 66 + 9 * (i + 1):66 + 9 * (i + 1) + 9] = Jf[:, 15:21], Jf[:, 21:30]
            rows_J.append(J), rows_r.append(f["imu_r"][b, i])
        n = int(a["prior_n"][b])
        Jp, off = np.zeros((n, NF + nf)), 0
        for k in range(int(a["prior_nblk"][b])):
            kind, fr = a["prior_blk_kind"][b, k], a["prior_blk_frame"][b, k]
            sz = 9 if kind == abi.BLK_SPEEDBIAS else 6
            if kind == abi.BLK_POSE:
                Jp[:, 6 * fr:6 * fr + 6] = a["prior_J"][b, :n, off:off + 6]
            elif kind == abi.BLK_SPEEDBIAS:
                Jp[:, 66 + 9 * fr:66 + 9 * fr + 9] = a["prior_J"][b, :n, off:off + 9]
            off += sz                                   # (ex_pose: constant in the solve, no columns)
        rows_J.append(Jp), rows_r.append(f["prior_res"][b, :n])
        J, r = np.vstack(rows_J), np.concatenate(rows_r)
        # (the reported cost is sum 1/2 rho(|r|^2); the corrected residuals only reproduce it to first order in the loss)
        assert abs(0.5 * r @ r - summ[b]["initial_cost"]) < 1e-4 * summ[b]["initial_cost"]
        H, g = J.T @ J, J.T @ r
        S = 1.0 / (1.0 + np.sqrt(np.diag(H)))          # Jacobi scaling
        Hs, gs = H * S[:, None] * S[None, :], g * S
        D2 = np.clip(np.diag(Hs), o.min_lm_diagonal, o.max_lm_diagonal)
        y = np.linalg.solve(Hs + 1e-8 * np.diag(D2), gs)   # DoglegStrategy: mu = min_mu = 1e-8 regularizes the Gauss-Newton solve
        assert np.sqrt(np.sum(D2 * y * y)) < o.initial_trust_region_radius   # Gauss-Newton step inside the region
        dx = -S * y
        # Plus
        pose = a["pose"][b].copy()
        for i in range(11):
            d = dx[6 * i:6 * i + 6]
            pose[i, :3] += d[:3]
            q = qmul(pose[i, 3:], np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0]))
            pose[i, 3:] = q / np.linalg.norm(q)
        sbias = a["speedbias"][b] + dx[66:165].reshape(11, 9)
        lam = a["inv_depth"][b, :nf] + dx[NF:]
        assert summ[b]["num_successful"] == 1
        got = ws.a
        assert rel(got["inv_depth"][b, :nf], lam) < 1e-7
        assert rel(got["speedbias"][b, :, 3:], sbias[:, 3:]) < 1e-7          # biases (the velocities are rotated by the gauge fix)
        R0n, R0g = q2R(pose[0, 3:]), q2R(got["pose"][b, 0, 3:])
        reln = np.array([R0n.T @ (pose[i, :3] - pose[0, :3]) for i in range(11)])
        relg = np.array([R0g.T @ (got["pose"][b, i, :3] - got["pose"][b, 0, :3]) for i in range(11)])
        assert rel(relg, reln) < 1e-7
        for i in range(1, 11):
            assert np.abs(R0g.T @ q2R(got["pose"][b, i, 3:]) - R0n.T @ q2R(pose[i, 3:])).max() < 1e-8
        veln = np.array([R0n.T @ sbias[i, :3] for i in range(11)])
        velg = np.array([R0g.T @ got["speedbias"][b, i, :3] for i in range(11)])
        assert rel(velg, veln) < 1e-7


def test_small_trust_region_exercises_dogleg_and_rejections(oracle):
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    o.initial_trust_region_radius = 1e-2
    o.max_num_iterations = 12
    w = synth.make_windows(1, tracks="sparse", n_feat=30, max_feat=150)
    s = buffers.summary_alloc(1)
    oracle.window_solve(o, w, None, s)
    n = s[0]["num_iterations"]
    assert n >= 8
    assert s[0]["radius_trace"][0] != s[0]["radius_trace"][n - 1]  # the radius moved
    assert s[0]["final_cost"] < s[0]["initial_cost"]


def _dense_schur_reference(opt, w_post, oracle, prior_n):
    """numpy: Schur complement of (pose0, sb0, start-0 features) from the factor Jacobians, non-ex kept rows only."""
    ev = oracle.eval_factors(opt, w_post, apply_loss=True)
    nf = int(w_post.a["n_feat"][0])
    f0 = [e for e in range(nf) if w_post.a["feat_start"][0, e] == 0]
    # variable layout: pose f -> 6f, sb f -> 66 + 9f, feature k -> 165 + k
    nz = 165 + len(f0)
    A, b = np.zeros((nz, nz)), np.zeros(nz)

    def add(J, r, cols):
        A[np.ix_(cols, cols)] += J.T @ J
        b[cols] += J.T @ r

    # prior
    if prior_n:
        n = prior_n
        kinds, frames = w_post.a["prior_blk_kind"][0], w_post.a["prior_blk_frame"][0]
        cols, keep = [], []
        off = 0
        for k in range(int(w_post.a["prior_nblk"][0])):
            ls = 9 if kinds[k] == abi.BLK_SPEEDBIAS else 6
            if kinds[k] == abi.BLK_POSE:
                cols += list(range(6 * frames[k], 6 * frames[k] + 6)); keep += list(range(off, off + ls))
            elif kinds[k] == abi.BLK_SPEEDBIAS:
                cols += list(range(66 + 9 * frames[k], 66 + 9 * frames[k] + 9)); keep += list(range(off, off + ls))
            off += ls
        J0 = w_post.a["prior_J"][0, :n, :n][:, keep]
        add(J0, ev["prior_res"][0, :n], cols)
    # imu factor 0
    cols = list(range(0, 6)) + list(range(66, 75)) + list(range(6, 12)) + list(range(75, 84))
    add(ev["imu_J"][0, 0], ev["imu_r"][0, 0], cols)
    for k, e in enumerate(f0):
        s0, no = int(w_post.a["feat_obs_begin"][0, e]), int(w_post.a["feat_nobs"][0, e])
        for t in range(1, no):
            J = ev["proj_J"][0, s0 + t]
            cols = list(range(0, 6)) + list(range(6 * t, 6 * t + 6)) + [165 + k]
            add(J, ev["proj_r"][0, s0 + t], cols)
    m_idx = list(range(0, 6)) + list(range(66, 75)) + list(range(165, nz))
    r_idx = [i for i in range(165) if i not in m_idx]
    Amm = A[np.ix_(m_idx, m_idx)]
    Amm = 0.5 * (Amm + Amm.T)
    ev_, V = np.linalg.eigh(Amm)
    inv = V @ np.diag(np.where(ev_ > 1e-8, 1.0 / ev_, 0.0)) @ V.T
    Arm = A[np.ix_(r_idx, m_idx)]
    S = A[np.ix_(r_idx, r_idx)] - Arm @ inv @ Arm.T
    bb = b[r_idx] - Arm @ inv @ b[m_idx]
    return r_idx, S, bb


def test_marginalization_equals_dense_schur_complement(oracle):
    o = abi.default_options()  # MARGIN_OLD
    w = synth.make_windows(1, tracks="sparse", n_feat=40, max_feat=150)
    po = buffers.PriorOutArrays.alloc(1)
    oracle.window_solve(o, w, po, buffers.summary_alloc(1))
    n, nb = int(po.a["n"][0]), int(po.a["nblk"][0])
    # expected kept poses: prior's (1..9), IMU's (1) and every frame seen by a feature that starts at frame 0
    nf = int(w.a["n_feat"][0])
    seen = {1} | set(range(1, 10))
    for e in range(nf):
        if w.a["feat_start"][0, e] == 0:
            seen |= set(range(1, int(w.a["feat_nobs"][0, e])))
    kept = sorted(seen)
    assert nb == len(kept) + 2 and n == 6 * len(kept) + 9 + 6
    kinds, frames = po.a["blk_kind"][0, :nb], po.a["blk_frame"][0, :nb]
    assert list(kinds) == [abi.BLK_POSE] * len(kept) + [abi.BLK_SPEEDBIAS, abi.BLK_EXPOSE]
    assert list(frames) == [f - 1 for f in kept] + [0, 0]  # addr_shift: pose[i] -> pose[i-1]
    J, r = po.a["J"][0, :n, :n], po.a["r"][0, :n]
    # independent numpy Schur complement on the post-solve state (w now holds it), compare the non-ex part
    o2 = abi.default_options()
    r_idx, S, bb = _dense_schur_reference(o2, w, oracle, int(w.a["prior_n"][0]))
    cols = [c for f in kept for c in range(6 * f, 6 * f + 6)] + list(range(75, 84))
    cols_ref = [r_idx.index(c) for c in cols]
    nk = len(cols)
    H = (J.T @ J)[:nk, :nk]
    g = (J.T @ r)[:nk]
    # the eigen-pseudo-inverse of Amm is conditioning limited (IMU bias weights ~1e12 against vision ~1e4):
    # two correct implementations (this numpy one sums ~1e12-sized terms that cancel to ~1e4) agree only to ~1e-4
    # in the worst entries, so compare with Jacobi (diagonal) scaling and a conditioning-aware tolerance
    Sr = S[np.ix_(cols_ref, cols_ref)]
    dsc = 1.0 / np.sqrt(np.diag(Sr))
    assert rel(H * dsc[:, None] * dsc[None, :], Sr * dsc[:, None] * dsc[None, :]) < 2e-3
    assert rel(H, Sr) < 1e-5
    assert rel(g * dsc, bb[cols_ref] * dsc) < 2e-3
    # x0 = the post-solve blocks (preMarginalize copies them)
    assert np.array_equal(po.a["x0"][0, 0, :7], w.a["pose"][0, 1])
    assert np.array_equal(po.a["x0"][0, len(kept), :9], w.a["speedbias"][0, 1])


def test_margin_second_new_drops_pose9(oracle):
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_SECOND_NEW
    w = synth.make_windows(1, tracks="sparse", n_feat=20, max_feat=150)
    po = buffers.PriorOutArrays.alloc(1)
    oracle.window_solve(o, w, po, buffers.summary_alloc(1))
    n, nb = int(po.a["n"][0]), int(po.a["nblk"][0])
    assert n == 69 and nb == 11
    assert list(po.a["blk_frame"][0, :nb]) == list(range(9)) + [0, 0]
    J = po.a["J"][0, :n, :n]
    assert np.linalg.eigvalsh(J.T @ J).min() > -1e-6


def test_selector_properties(oracle):
    pr = synth.make_fsel(2, horizon=5, n_cand=40, n_used=3, max_features=12)
    om, dl, va = oracle.fsel_information(pr)
    N = 9 * 6
    assert rel(om, np.transpose(om, (0, 2, 1))) < 1e-12
    assert np.linalg.eigvalsh(om[0]).min() > 0
    # block tridiagonal: zero beyond one block off the diagonal
    for i in range(6):
        for j in range(6):
            if abs(i - j) > 1:
                assert not om[0, 9 * i:9 * i + 9, 9 * j:9 * j + 9].any()
    for c in range(40):
        if va[0, c]:
            D = dl[0, c]
            assert rel(D, D.T) < 1e-12 or np.abs(D).max() < 1e-300
            assert np.linalg.eigvalsh(0.5 * (D + D.T)).min() > -1e-9 * max(1.0, np.abs(D).max())
    out = buffers.FselOutArrays.alloc(2, 12)
    nld = oracle.fsel_select(pr, out)
    for p in range(2):
        n = int(out.a["n_selected"][p])
        assert n == 12 - 3
        ids = out.a["selected_ids"][p, :n]
        assert len(set(ids.tolist())) == n
        assert set(ids.tolist()) <= set(pr.a["cand_id"][p].tolist())
        f = out.a["fvalues"][p, :n]
        assert np.all(np.diff(f) >= -1e-9)  # adding information never lowers logdet
    assert nld > 0


def test_horizon_imu_information_matches_numpy(oracle):
    """Pins calcInfoFromRobotMotion + createLinearImuMatrices + addOmegaPrior (feature_selector.cpp:463-609) against a numpy
    statement built from Eigen's documented slerp and a dense inverse of the covariance of eq. (52)."""
    def q2R(q):  # x y z w
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def slerp(qa, qb, t):
        d = float(qa @ qb)
        ad = abs(d)
        if ad >= 1.0 - np.finfo(float).eps:
            s0, s1 = 1.0 - t, t
        else:
            th = np.arccos(ad)
            s0, s1 = np.sin((1.0 - t) * th) / np.sin(th), np.sin(t * th) / np.sin(th)
        if d < 0:
            s1 = -s1
        return s0 * qa + s1 * qb

    H = 4
    pr = synth.make_fsel(3, horizon=H, n_cand=8, n_used=0, n_cloud=4, max_features=4)
    om, _, _ = oracle.fsel_information(pr)
    a, sc = pr.a, pr.scalars
    I3 = np.eye(3)
    for p in range(3):
        n, dt = int(a["nr_imu"][p]), float(a["delta_imu"][p])
        Om = np.zeros((9 * (H + 1), 9 * (H + 1)))
        for h in range(1, H + 1):
            qi, qj = a["hor_quat"][p, h - 1], a["hor_quat"][p, h]
            Nij, Mij, c11, c12 = np.zeros((3, 3)), np.zeros((3, 3)), 0.0, 0.0
            for i in range(n):
                R = q2R(slerp(qi, qj, i / n))
                jkh = n - i - 0.5
                Nij += jkh * R
                Mij += R
                c11 += jkh * jkh
                c12 += jkh
            cov = np.zeros((9, 9))
            cov[0:3, 0:3] = I3 * n * c11 * dt ** 4 * sc["acc_var"]
            cov[0:3, 3:6] = cov[3:6, 0:3] = I3 * c12 * dt ** 3 * sc["acc_var"]
            cov[3:6, 3:6] = I3 * n * dt ** 2 * sc["acc_var"]
            cov[6:9, 6:9] = I3 * n * sc["acc_bias_var"]
            W = np.linalg.inv(cov)
            A = -np.eye(9)
            A[0:3, 3:6] = -I3 * n * dt
            A[0:3, 6:9] = Nij * dt * dt
            A[3:6, 6:9] = Mij * dt
            lo, hi = slice(9 * (h - 1), 9 * h), slice(9 * h, 9 * (h + 1))
            Om[lo, lo] += A.T @ W @ A
            Om[lo, hi] += A.T @ W
            Om[hi, lo] += W @ A
            Om[hi, hi] += W
        Om[:9, :9] += np.eye(9)
        assert rel(om[p], Om) < 1e-10
        assert np.abs(om[p] - om[p].T).max() <= 1e-9 * np.abs(om[p]).max()


def test_feature_information_matches_numpy(oracle):
    """Pins calcInfoFromFeatures (feature_selector.cpp:239-365; PinholeCamera::spaceToPlane / distortion, inFOV, findNNDepth)
    against an independent numpy statement, candidate by candidate: validity and the 3H x 3H position blocks of Delta."""
    def q2R(q):  # x y z w
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def skew(v):
        return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])

    H = 5
    pr = synth.make_fsel(2, horizon=H, n_cand=50, n_used=0, n_cloud=30, max_features=10)
    om, dl, va = oracle.fsel_information(pr)
    a, sc = pr.a, pr.scalars
    Ric, tic = q2R(np.asarray(sc["q_ic"], float)), np.asarray(sc["t_ic"], float)
    n_valid = 0
    for p in range(2):
        Rw = [q2R(a["hor_quat"][p, h]) for h in range(H + 1)]
        tWC = [a["hor_pos"][p, h] + Rw[h] @ tic for h in range(H + 1)]
        RWC = [Rw[h] @ Ric for h in range(H + 1)]
        for c in range(int(a["n_cand"][p])):
            x, y = a["cand_xy"][p, c]
            d = 1.0
            ncl = int(a["n_cloud"][p])
            if ncl:
                dist = ((a["cloud_xy"][p, :ncl] - [x, y]) ** 2).sum(1)
                d = a["cloud_depth"][p, int(np.argmin(dist))]      # first minimum, like the strict '<' scan
            fn = np.array([x, y, 1.0]) / np.linalg.norm([x, y, 1.0])
            pell = tWC[1] + RWC[1] @ (fn * d)
            Ch, EtE, nvis = {}, np.zeros((3, 3)), 1
            for h in range(2, H + 1):
                u = RWC[h].T @ (pell - tWC[h])
                u = u / np.linalg.norm(u)
                xu, yu = u[0] / u[2], u[1] / u[2]
                r2 = xu * xu + yu * yu
                rad = sc["k1"] * r2 + sc["k2"] * r2 * r2
                dx = xu * rad + 2 * sc["p1"] * xu * yu + sc["p2"] * (r2 + 2 * xu * xu)
                dy = yu * rad + 2 * sc["p2"] * xu * yu + sc["p1"] * (r2 + 2 * yu * yu)
                px, py = sc["fx"] * (xu + dx) + sc["cx"], sc["fy"] * (yu + dy) + sc["cy"]
                iu, iv = int(np.floor(abs(px) + 0.5) * np.sign(px)), int(np.floor(abs(py) + 0.5) * np.sign(py))   # std::round
                if not (0 <= iu < sc["image_width"] and 0 <= iv < sc["image_height"]):
                    continue
                Bh = skew(u) @ (RWC[h] @ Ric).T          # (q_WC_h * q_IC)^-1: q_IC applied twice, as in the reference
                Ch[h] = Bh.T @ Bh
                EtE += Ch[h]
                nvis += 1
            assert bool(va[p, c]) == (nvis > 1), (p, c)
            if nvis == 1:
                continue
            B1 = skew(fn) @ (RWC[1] @ Ric).T
            Ch[1] = B1.T @ B1
            EtE += Ch[1]
            W = np.linalg.inv(EtE)
            D = np.zeros((3 * H, 3 * H))
            for j in range(1, H + 1):
                for i in range(j, H + 1):
                    Ci, Cj = Ch.get(i, np.zeros((3, 3))), Ch.get(j, np.zeros((3, 3)))
                    Dij = Ci @ W @ Cj.T
                    if i == j:
                        D[3 * (i - 1):3 * i, 3 * (j - 1):3 * j] = Ci - Dij
                    else:
                        D[3 * (i - 1):3 * i, 3 * (j - 1):3 * j] = -Dij
                        D[3 * (j - 1):3 * j, 3 * (i - 1):3 * i] = -Dij.T
            assert np.abs(dl[p, c] - D).max() < 1e-9 * max(1.0, np.abs(D).max()), (p, c)
            n_valid += 1
    assert n_valid > 20


def test_greedy_selection_matches_numpy_brute_force(oracle):
    """Pins selectInformativeFeatures (feature_selector.cpp:613-686) and the hoists of the restatement (reduced position
    system, Hadamard bound ordering) against a brute-force numpy greedy: every round, slogdet of the FULL 9(H+1) x 9(H+1)
    matrix Omega + OmegaS + p Delta for every remaining candidate, arg max."""
    for H, nc, mf in ((3, 25, 8), (5, 30, 10)):
        pr = synth.make_fsel(2, horizon=H, n_cand=nc, n_used=0, max_features=mf)
        om, dl, va = oracle.fsel_information(pr)
        out = buffers.FselOutArrays.alloc(2, mf)
        oracle.fsel_select(pr, out)
        N, T = 9 * (H + 1), 3 * H
        pos = np.array([9 * (1 + i // 3) + i % 3 for i in range(T)])          # position rows of horizon states 1..H
        for p in range(2):
            M = om[p].copy()
            live = [c for c in range(nc) if va[p, c]]
            ids, fvals = [], []
            for _ in range(mf):
                best, bf = None, -1.0
                for c in live:
                    Mc = M.copy()
                    Mc[np.ix_(pos, pos)] += pr.a["cand_prob"][p, c] * dl[p, c]
                    sign, ld = np.linalg.slogdet(Mc)
                    assert sign > 0
                    if ld > bf:
                        best, bf = c, ld
                if best is None:
                    break
                M[np.ix_(pos, pos)] += pr.a["cand_prob"][p, best] * dl[p, best]
                live.remove(best)
                ids.append(int(pr.a["cand_id"][p, best])), fvals.append(bf)
            n = int(out.a["n_selected"][p])
            assert n == len(ids) and out.a["selected_ids"][p, :n].tolist() == ids
            assert rel(out.a["fvalues"][p, :n], np.array(fvals)) < 1e-10


def test_selector_kappa_zero_and_empty_cloud(oracle):
    pr = synth.make_fsel(1, horizon=3, n_cand=10, n_used=4, max_features=4, n_cloud=0)
    out = buffers.FselOutArrays.alloc(1, 4)
    oracle.fsel_select(pr, out)
    assert int(out.a["n_selected"][0]) == 0  # kappa = max(0, maxFeatures - |subset|) = 0


def _numpy_triangulate(win, b, e):
    """Independent numpy statement of feature_manager.cpp:209-249 (numpy.linalg.svd instead of Eigen::JacobiSVD)."""
    def q2R(q):  # x y z w
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    a = win.a
    ex = a["ex_pose"][b]
    tic, ric = ex[:3], q2R(ex[3:])
    st, no, ob = a["feat_start"][b, e], a["feat_nobs"][b, e], a["feat_obs_begin"][b, e]
    P_ = lambda f: a["pose"][b, f, :3]
    R_ = lambda f: q2R(a["pose"][b, f, 3:])
    t0, R0 = P_(st) + R_(st) @ tic, R_(st) @ ric
    rows = []
    for k in range(no):
        j = st + k
        t1, R1 = P_(j) + R_(j) @ tic, R_(j) @ ric
        t, R = R0.T @ (t1 - t0), R0.T @ R1
        P = np.hstack([R.T, (-R.T @ t)[:, None]])
        f = np.array([a["obs_xy"][b, ob + k, 0], a["obs_xy"][b, ob + k, 1], 1.0])
        f = f / np.linalg.norm(f)
        rows += [f[0] * P[2] - f[2] * P[0], f[1] * P[2] - f[2] * P[1]]
    A = np.array(rows)
    v = np.linalg.svd(A)[2][-1]
    return A, v[2] / v[3]


def test_selector_with_non_finite_inputs(oracle):
    """deltaImu = 0 (the reference's first call), a NaN horizon frame, a NaN candidate, zero probabilities, infinite depths:
    a NaN pixel is outside the image (the reference's double -> int conversion gives INT_MIN), NaN f values never win."""
    pr = synth.make_fsel(6, horizon=5, n_cand=40, n_used=3, n_cloud=20, max_features=12)
    pr.a["delta_imu"][0] = 0.0
    pr.a["hor_pos"][1, 2, 0] = np.nan
    pr.a["cand_xy"][2, 3] = np.nan
    pr.a["cand_prob"][3, :] = 0.0
    pr.a["cloud_depth"][4, :] = np.inf
    out = buffers.FselOutArrays.alloc(6, 12)
    oracle.fsel_select(pr, out)
    assert out.a["n_selected"].tolist() == [0, 9, 9, 9, 0, 9]
    assert int(pr.a["cand_id"][2, 3]) not in out.a["selected_ids"][2].tolist()


def test_triangulate_matches_numpy_svd(oracle):
    """FeatureManager::triangulate. Pins the oracle's one-sided Jacobi SVD and its construction of the
    (2 nobs) x 4 system against numpy.linalg.svd / an independent numpy statement."""
    rng = np.random.default_rng(3)
    for n in (4, 7, 22):
        A = rng.normal(size=(n, 4))
        A[:, 3] = A[:, :3] @ rng.normal(size=3) + 1e-3 * rng.normal(size=n)   # a small last singular value
        v, vr = oracle.smallest_right_singular_vector(A), np.linalg.svd(A)[2][-1]
        assert min(np.abs(v - vr).max(), np.abs(v + vr).max()) < 1e-11
    w = synth.make_windows(3, tracks="sparse", n_feat=40, max_feat=150)
    keep = w.a["inv_depth"].copy()
    w.a["inv_depth"][:, ::2] = -1.0          # estimated_depth = -1: "no depth yet"; odd features keep theirs
    oracle.triangulate(w, init_depth=5.0)
    assert np.array_equal(w.a["inv_depth"][:, 1::2], keep[:, 1::2])
    checked = 0
    for b in range(3):
        for e in range(0, w.a["n_feat"][b], 2):
            _, d = _numpy_triangulate(w, b, e)
            exp = 1.0 / d if d >= 0.1 else 1.0 / 5.0
            assert abs(w.a["inv_depth"][b, e] - exp) <= 1e-9 * abs(exp), (b, e)
            checked += 1
    assert checked > 40
    # the triangulated depths are close to the ones the generator perturbed (poses are 5 cm / 1 deg off, 1.5 px noise)
    r = np.concatenate([w.a["inv_depth"][b, : w.a["n_feat"][b] : 2] / keep[b, : w.a["n_feat"][b] : 2] for b in range(3)])
    assert np.median(np.abs(r - 1)) < 0.3


def test_newest_frame_dead_reckoning_matches_numpy(oracle, opt):
    """Estimator::processIMU (estimator.cpp:100-107) against an independent numpy statement."""
    def q2R(q):  # x y z w, Eigen toRotationMatrix (no normalization)
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    w = synth.make_windows(3, tracks="sparse", n_feat=8, max_feat=150)
    w.a["pose"][:, 10] = w.a["pose"][:, 9]            # what slideWindow() leaves behind
    w.a["speedbias"][:, 10] = w.a["speedbias"][:, 9]
    w.a["imu_n"][1, 9] = 7                             # ragged
    g = np.array(list(opt.g))
    ref = w.copy()
    oracle.imu_propagate(w, g)
    for b in range(3):
        P, V = ref.a["pose"][b, 10, :3].copy(), ref.a["speedbias"][b, 10, :3].copy()
        Ba, Bg = ref.a["speedbias"][b, 10, 3:6], ref.a["speedbias"][b, 10, 6:9]
        R = q2R(ref.a["pose"][b, 10, 3:])
        acc, gyr, dt = ref.a["imu_acc"][b, 9], ref.a["imu_gyr"][b, 9], ref.a["imu_dt"][b, 9]
        a0, w0 = acc[0], gyr[0]
        for s_ in range(ref.a["imu_n"][b, 9]):
            a1, w1, h = acc[s_ + 1], gyr[s_ + 1], dt[s_]
            ua0 = R @ (a0 - Ba) - g
            ug = 0.5 * (w0 + w1) - Bg
            R = R @ q2R(np.array([ug[0] * h / 2, ug[1] * h / 2, ug[2] * h / 2, 1.0]))
            ua = 0.5 * (ua0 + R @ (a1 - Ba) - g)
            P, V = P + h * V + 0.5 * h * h * ua, V + h * ua
            a0, w0 = a1, w1
        assert rel(w.a["pose"][b, 10, :3], P) < 1e-13 and rel(w.a["speedbias"][b, 10, :3], V) < 1e-13
        assert rel(q2R(w.a["pose"][b, 10, 3:]), R) < 1e-9      # R is only orthonormal to first order (unnormalized deltaQ)
        assert np.array_equal(w.a["pose"][b, :10], ref.a["pose"][b, :10]) and np.array_equal(w.a["speedbias"][b, 10, 3:], ref.a["speedbias"][b, 10, 3:])


def _horizon_inputs(P, rng):
    q = rng.normal(size=(P, 2, 4)); q /= np.linalg.norm(q, axis=-1, keepdims=True)
    return dict(k_pos=rng.normal(size=(P, 3)), k_quat=q[:, 0], k_ba=0.02 * rng.normal(size=(P, 3)), k1_pos=rng.normal(size=(P, 3)),
                k1_vel=rng.normal(size=(P, 3)), k1_quat=q[:, 1], acc=np.array([0, 0, 9.8]) + rng.normal(size=(P, 3)),
                gyr=0.3 * rng.normal(size=(P, 3)), nr_imu=rng.integers(1, 25, P), delta_imu=np.full(P, 0.005))


def test_horizon_generator_imu_matches_numpy(oracle):
    """B4: HorizonGenerator::imu (utility/horizon_generator.cpp:25-69) against an independent numpy statement
    (Hamilton product, unnormalized deltaQ, no renormalization, gravity (0, 0, -9.80665))."""
    def qmul(a, b):  # w x y z
        return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                         a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])
    def qrot(q, v):  # Eigen: v + w t + qv x t, t = 2 qv x v
        t = 2 * np.cross(q[1:], v)
        return v + q[0] * t + np.cross(q[1:], t)
    rng = np.random.default_rng(11)
    H, P = 10, 5
    i = _horizon_inputs(P, rng)
    hp, hq = oracle.fsel_horizon_imu(H, **i)
    g = np.array([0, 0, -9.80665])
    for p in range(P):
        assert np.array_equal(hp[p, 0], i["k_pos"][p]) and np.array_equal(hp[p, 1], i["k1_pos"][p])
        assert np.array_equal(hq[p, 0], i["k_quat"][p]) and np.array_equal(hq[p, 1], i["k1_quat"][p])
        dI, w, a, Ba = i["delta_imu"][p], i["gyr"][p], i["acc"][p], i["k_ba"][p]
        Qimu = np.array([1.0, *(w * dI / 2)])
        q = np.array([i["k1_quat"][p][3], *i["k1_quat"][p][:3]])
        pos, vel = i["k1_pos"][p].copy(), i["k1_vel"][p].copy()
        for h in range(2, H + 1):
            for _ in range(i["nr_imu"][p]):
                q = qmul(q, Qimu)
                qa = qrot(q, a - Ba)
                vel = vel + (g + qa) * dI
                pos = pos + vel * dI + 0.5 * g * dI * dI + 0.5 * qa * dI * dI
            assert rel(hp[p, h], pos) < 1e-13 and rel(hq[p, h], np.array([q[1], q[2], q[3], q[0]])) < 1e-13


def test_depth_cloud_matches_numpy(oracle):
    """the cloud FeatureSelector::initKDTree builds (feature_selector.cpp:396-419), against numpy."""
    def q2R(q):  # x y z w
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    B = 3
    w = synth.make_windows(B, tracks="sparse", n_feat=60, max_feat=150)
    a = w.a
    a["inv_depth"][:, 5::11] *= -1.0                    # solve_flag == 2 (negative depth): never in the cloud
    rng = np.random.default_rng(8)
    k1_pos = a["pose"][:, 10, :3] + 0.1 * rng.normal(size=(B, 3))
    k1_quat = a["pose"][:, 10, 3:].copy()
    n, xy, dep = oracle.fsel_build_cloud(w, k1_pos, k1_quat, max_cloud=150)
    for b in range(B):
        ric, tic = q2R(a["ex_pose"][b, 3:]), a["ex_pose"][b, :3]
        Rk1 = q2R(k1_quat[b])
        exp = []
        for e in range(a["n_feat"][b]):
            st = a["feat_start"][b, e]
            if st > 7 or a["inv_depth"][b, e] <= 0:
                continue
            d = 1.0 / a["inv_depth"][b, e]
            o = a["obs_xy"][b, a["feat_obs_begin"][b, e]]
            pw = q2R(a["pose"][b, st, 3:]) @ (ric @ (d * np.array([o[0], o[1], 1.0])) + tic) + a["pose"][b, st, :3]
            pc = ric.T @ (Rk1.T @ (pw - k1_pos[b]) - tic)
            exp.append((pc[0] / pc[2], pc[1] / pc[2], d))
        exp = np.array(exp)
        assert n[b] == len(exp) and 0 < n[b] < a["n_feat"][b]
        assert np.abs(xy[b, : n[b]] - exp[:, :2]).max() < 1e-11 and np.array_equal(dep[b, : n[b]], exp[:, 2])
        assert not xy[b, n[b]:].any()
    # capacity clamp keeps the first max_cloud qualifying features
    n2, xy2, dep2 = oracle.fsel_build_cloud(w, k1_pos, k1_quat, max_cloud=7)
    assert (n2 == 7).all() and np.array_equal(xy2, xy[:, :7]) and np.array_equal(dep2, dep[:, :7])


def _td_factor_inputs(n, rng):
    def unit(a):
        return a / np.linalg.norm(a, axis=-1, keepdims=True)
    def pose(scale):
        q = unit(np.array([0, 0, 0, 1.0]) + 0.2 * rng.normal(size=(n, 4)))
        return np.hstack([scale * rng.normal(size=(n, 3)), q])
    return dict(pose_i=pose(0.5), pose_j=pose(0.5), ex_pose=pose(0.05), inv_depth=1.0 / rng.uniform(2, 15, n), td=0.01 * rng.normal(size=n),
                pts_i=0.4 * rng.normal(size=(n, 2)), pts_j=0.4 * rng.normal(size=(n, 2)), vel_i=0.3 * rng.normal(size=(n, 2)),
                vel_j=0.3 * rng.normal(size=(n, 2)), td_i=0.01 * rng.normal(size=n), td_j=0.01 * rng.normal(size=n),
                row_i=rng.uniform(0, 480, n), row_j=rng.uniform(0, 480, n))


def test_projection_td_factor_residual_and_fd_jacobian(oracle):
    """A7: ProjectionTdFactor::Evaluate (projection_td_factor.cpp:34-141).  Residual against an independent numpy
    statement; all 20 Jacobian columns against central differences through the local parameterization
    (p + dp, q * deltaQ(dtheta): pose_local_parameterization.cpp:3-27), the convention of the reference's own check()."""
    rng = np.random.default_rng(5)
    n, TR, ROW, F = 6, 0.033, 480.0, 460.0
    a = _td_factor_inputs(n, rng)
    r, J = oracle.projection_td_eval(a, TR, ROW, F)

    def q2R(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])

    def residual(k, pi, pj, ex, lam, td):
        Ri, Rj, ric = q2R(pi[3:] / np.linalg.norm(pi[3:])), q2R(pj[3:] / np.linalg.norm(pj[3:])), q2R(ex[3:] / np.linalg.norm(ex[3:]))
        pts_i = np.array([*a["pts_i"][k], 1.0]) - (td - a["td_i"][k] + TR / ROW * (a["row_i"][k] - ROW / 2)) * np.array([*a["vel_i"][k], 0.0])
        pts_j = np.array([*a["pts_j"][k], 1.0]) - (td - a["td_j"][k] + TR / ROW * (a["row_j"][k] - ROW / 2)) * np.array([*a["vel_j"][k], 0.0])
        pc = ric.T @ (Rj.T @ (Ri @ (ric @ (pts_i / lam) + ex[:3]) + pi[:3] - pj[:3]) - ex[:3])
        return F / 1.5 * (pc[:2] / pc[2] - pts_j[:2])

    def plus(p, d):  # x y z qx qy qz qw (+) [dp, dtheta]
        x, y, z, w = p[3:]
        dq = np.array([d[3] / 2, d[4] / 2, d[5] / 2, 1.0])
        q = np.array([w * dq[0] + x * dq[3] + y * dq[2] - z * dq[1], w * dq[1] - x * dq[2] + y * dq[3] + z * dq[0],
                      w * dq[2] + x * dq[1] - y * dq[0] + z * dq[3], w * dq[3] - x * dq[0] - y * dq[1] - z * dq[2]])
        return np.hstack([p[:3] + d[:3], q / np.linalg.norm(q)])

    for k in range(n):
        args = [a["pose_i"][k], a["pose_j"][k], a["ex_pose"][k], a["inv_depth"][k], a["td"][k]]
        assert rel(r[k], residual(k, *args)) < 1e-11
        eps = 1e-6
        for col in range(20):
            blk, off = (col // 6, col % 6) if col < 18 else (col - 15, 0)
            def shifted(sgn):
                b = list(args)
                if blk < 3:
                    d = np.zeros(6); d[off] = sgn * eps
                    b[blk] = plus(b[blk], d)
                else:
                    b[blk] = b[blk] + sgn * eps * (abs(b[blk]) if blk == 3 else 1.0)
                return residual(k, *b)
            scale = abs(args[3]) if blk == 3 else 1.0
            fd = (shifted(+1) - shifted(-1)) / (2 * eps * scale)
            assert np.abs(fd - J[k, :, col]).max() < 2e-5 * max(1.0, np.abs(J[k, :, col]).max()), (k, col)


def _roll_inputs(B=4, seed=2):
    """Windows whose tables look like a feature manager's: ragged tracks plus a few features that the solve would not
    take (single observations, tracks starting in the last frames) and a few untriangulated depths."""
    w = synth.make_windows(B, tracks="sparse", n_feat=60, max_feat=150, max_samp=40)
    a = w.a
    rng = np.random.default_rng(seed)
    a["imu_n"][:] = rng.integers(5, 18, a["imu_n"].shape)
    for b in range(B):
        n, o = int(a["n_feat"][b]), int(a["feat_obs_begin"][b, a["n_feat"][b] - 1] + a["feat_nobs"][b, a["n_feat"][b] - 1])
        extra = [(0, 1), (0, 2), (3, 1), (9, 1), (9, 2), (10, 1), (10, 1)]
        for st, no in sorted(extra):                  # std::list order = non-decreasing start frame
            a["feat_start"][b, n], a["feat_nobs"][b, n], a["feat_obs_begin"][b, n] = st, no, o
            a["obs_xy"][b, o:o + no] = rng.normal(scale=0.3, size=(no, 2))
            a["inv_depth"][b, n] = rng.uniform(0.1, 0.5)
            n, o = n + 1, o + no
        order = np.argsort(a["feat_start"][b, :n], kind="stable")
        for k in ("feat_start", "feat_nobs", "feat_obs_begin", "inv_depth"):
            a[k][b, :n] = a[k][b, :n][order]
        a["n_feat"][b] = n
        a["inv_depth"][b, 1:n:9] = -1.0                 # estimated_depth = -1: not triangulated yet
        a["inv_depth"][b, 2:n:13] = -0.2                # a negative depth (solve_flag == 2 material)
    return w


def _roll_python(a, b, flag, shift_depth, INIT_DEPTH):
    """Independent statement with Python lists: estimator.cpp:996-1107, feature_manager.cpp:275-352."""
    def q2R(q):
        x, y, z, w = q
        return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                         [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                         [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    P = [a["pose"][b, i, :3].copy() for i in range(11)]
    R = [q2R(a["pose"][b, i, 3:]) for i in range(11)]
    poses = [a["pose"][b, i].copy() for i in range(11)]
    sbs = [a["speedbias"][b, i].copy() for i in range(11)]
    ric, tic = q2R(a["ex_pose"][b, 3:]), a["ex_pose"][b, :3]
    feats = []
    for e in range(a["n_feat"][b]):
        ob, no = a["feat_obs_begin"][b, e], a["feat_nobs"][b, e]
        feats.append(dict(start=int(a["feat_start"][b, e]), obs=[tuple(v) for v in a["obs_xy"][b, ob:ob + no]], depth=1.0 / a["inv_depth"][b, e]))
    imu = [dict(dt=list(a["imu_dt"][b, j, :a["imu_n"][b, j]]), acc=[tuple(v) for v in a["imu_acc"][b, j, :a["imu_n"][b, j] + 1]],
                gyr=[tuple(v) for v in a["imu_gyr"][b, j, :a["imu_n"][b, j] + 1]], ba=a["imu_lin_ba"][b, j].copy(), bg=a["imu_lin_bg"][b, j].copy())
           for j in range(10)]
    acc_0, gyr_0 = imu[9]["acc"][-1], imu[9]["gyr"][-1]
    if flag == abi.MARGIN_OLD:
        back_R0, back_P0 = R[0], P[0]
        poses, sbs = poses[1:] + [poses[10]], sbs[1:] + [sbs[10]]
        imu = imu[1:]
        R1, P1 = R[1] @ ric, P[1] + R[1] @ tic
        R0, P0 = back_R0 @ ric, back_P0 + back_R0 @ tic
        out = []
        for f in feats:
            if f["start"] != 0:
                f["start"] -= 1
            else:
                uv = np.array([*f["obs"].pop(0), 1.0])
                if shift_depth:
                    if len(f["obs"]) < 2:
                        continue
                    dep = (R1.T @ (R0 @ (uv * f["depth"]) + P0 - P1))[2]
                    f["depth"] = dep if dep > 0 else INIT_DEPTH
                elif len(f["obs"]) == 0:
                    continue
            out.append(f)
        feats = out
    else:
        imu[8]["dt"] += imu[9]["dt"]; imu[8]["acc"] += imu[9]["acc"][1:]; imu[8]["gyr"] += imu[9]["gyr"][1:]
        imu = imu[:9]
        poses[9], sbs[9] = poses[10], sbs[10]
        out = []
        for f in feats:
            if f["start"] == 10:
                f["start"] = 9
            elif f["start"] + len(f["obs"]) - 1 >= 9:
                f["obs"].pop(9 - f["start"])
                if not f["obs"]:
                    continue
            out.append(f)
        feats = out
    imu.append(dict(dt=[], acc=[acc_0], gyr=[gyr_0], ba=sbs[10][3:6], bg=sbs[10][6:9]))
    return poses, sbs, imu, feats


def check_roll(a, ref, b):
    poses, sbs, imu, feats = ref
    assert np.array_equal(a["pose"][b], np.array(poses)) and np.array_equal(a["speedbias"][b], np.array(sbs))
    assert a["n_feat"][b] == len(feats)
    for e, f in enumerate(feats):
        ob, no = a["feat_obs_begin"][b, e], a["feat_nobs"][b, e]
        assert (a["feat_start"][b, e], no) == (f["start"], len(f["obs"])), (b, e)
        assert np.array_equal(a["obs_xy"][b, ob:ob + no], np.array(f["obs"]).reshape(-1, 2)), (b, e)
        assert abs(a["inv_depth"][b, e] * f["depth"] - 1) < 1e-12, (b, e)
    for j in range(10):
        n = a["imu_n"][b, j]
        assert n == len(imu[j]["dt"]) and np.array_equal(a["imu_dt"][b, j, :n], np.array(imu[j]["dt"]))
        assert np.array_equal(a["imu_acc"][b, j, :n + 1], np.array(imu[j]["acc"])) and np.array_equal(a["imu_gyr"][b, j, :n + 1], np.array(imu[j]["gyr"]))
        assert np.array_equal(a["imu_lin_ba"][b, j], imu[j]["ba"]) and np.array_equal(a["imu_lin_bg"][b, j], imu[j]["bg"])


def test_window_roll_matches_python_lists(oracle):
    """slideWindow + removeBackShiftDepth / removeBack / removeFront on the batch tables."""
    for flag, shift in ((abi.MARGIN_OLD, True), (abi.MARGIN_OLD, False), (abi.MARGIN_SECOND_NEW, True)):
        w = _roll_inputs()
        ref = [_roll_python(w.a, b, flag, shift, 5.0) for b in range(w.n_windows)]
        n0 = w.a["n_feat"].copy()
        assert oracle.slide_window(w, flag, shift, 5.0) == 0
        for b in range(w.n_windows):
            check_roll(w.a, ref[b], b)
        assert (w.a["n_feat"] < n0).all()            # every window erased something
    # the re-anchored depth of a good track is the point's depth in the new first camera (positive, same 3-D point)
    w = _roll_inputs()
    w.a["imu_n"][:, 8], w.a["imu_n"][:, 9] = 30, 20
    assert oracle.slide_window(w, abi.MARGIN_SECOND_NEW, True, 5.0) == -3    # 50 samples do not fit max_samp = 40


def test_vectorisable_linear_algebra_of_the_bench_leg_is_the_literal_one_to_rounding(oracle):
    """bench.py's second CPU baseline ("port_blocked") switches the oracle's Schur update, Cholesky factorization and
    forward substitution to loops the compiler can vectorise (oracle/linalg.hpp: llt_lower_fast - the same subtractions in the same
    order, as row updates of the upper factor instead of dot products; the compiler's vector code rounds a few last bits differently).
    Same decisions; states within 1e-11 of the literal path on a window with a prior, 1e-8 on a prior-less one (its gauge directions
    carry rounding differences into the states: measured 2e-10)."""
    import numpy as np

    from helpers import abi, buffers, rel, synth

    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    for tracks, nf, prior in (("dense", 150, True), ("sparse", 70, False)):
        w = synth.make_windows(3, first_id=77, tracks=tracks, n_feat=nf, max_feat=150, with_prior=prior)
        a, b = w.copy(), w.copy()
        sa, sb = buffers.summary_alloc(3), buffers.summary_alloc(3)
        oracle.window_solve(o, a, None, sa)
        oracle.set_fast_linalg(True)
        try:
            oracle.window_solve(o, b, None, sb)
        finally:
            oracle.set_fast_linalg(False)
        assert np.array_equal(sa["accept_mask"], sb["accept_mask"]) and np.array_equal(sa["num_iterations"], sb["num_iterations"])
        for k in ("pose", "speedbias", "inv_depth"):
            assert rel(b.a[k], a.a[k]) < (1e-11 if prior else 1e-8), (k, rel(b.a[k], a.a[k]))
