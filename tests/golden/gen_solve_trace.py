"""Generates tests/golden/solve_trace.npz: the complete trust-region trace of one small window from an INDEPENDENT numpy
statement of what ceres::Solve does for Estimator::optimization() (estimator.cpp:794-809; Ceres 1.14 TrustRegionMinimizer +
DoglegStrategy(TRADITIONAL_DOGLEG)) - written from the algorithm, not from oracle/ or csrc/:

  * the factors from the reference sources: ProjectionFactor (projection_factor.cpp:21-121) + CauchyLoss / Corrector,
    IMUFactor residual AND analytic Jacobians (imu_factor.h:19-179, with the uncorrected delta_q in d r_R / d bg_i and the
    Qleft / Qright conventions of utility.h), IntegrationBase (integration_base.h:54-186), MarginalizationFactor
    (marginalization_factor.cpp:333-381), PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-19);
  * the solver on the FULL dense Jacobian (no Schur complement, no block structure: numpy.linalg on the 165 + F columns):
    Jacobi scaling, D = sqrt(clamp(diag J^T J)), Gauss-Newton step of (J^T J + mu D^2), Cauchy point, traditional dogleg,
    model_cost_change = -(J s)^T (r + J s / 2), rho, radius update, the mu policy, the termination tests, <= max_num_iterations
    step attempts.

The start point is far enough off that the trace contains REJECTED steps (radius halving, the Gauss-Newton step re-used) and
dogleg steps on the trust-region boundary.  The oracle (CPU tier) and the HIP path (GPU tier) are compared with this file:
cost, radius, accept / reject at every iteration, final states.

Run once in the build container:  python tests/golden/gen_solve_trace.py
"""
import importlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]
from gen_golden import cauchy_correct, imu_residual_raw, preintegrate, projection_factor, q2R, qmul, skew  # noqa: E402

PKG = "anticipated-vins-mono_amd"
G = np.array([0.0, 0.0, 9.81007])
NOISE = (0.08, 0.004, 0.00004, 2.0e-6)
SQRT_INFO = 460.0 / 1.5


def wq(p):  # pose block (x y z qx qy qz qw) -> quaternion (w, x, y, z)
    return np.array([p[6], p[3], p[4], p[5]])


def qconj(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / (q @ q)


def Qleft(q):
    w, v = q[0], q[1:]
    M = np.zeros((4, 4))
    M[0, 0], M[0, 1:], M[1:, 0], M[1:, 1:] = w, -v, v, w * np.eye(3) + skew(v)
    return M


def Qright(q):
    w, v = q[0], q[1:]
    M = np.zeros((4, 4))
    M[0, 0], M[0, 1:], M[1:, 0], M[1:, 1:] = w, -v, v, w * np.eye(3) - skew(v)
    return M


def deltaQ(th):
    return np.array([1.0, th[0] / 2, th[1] / 2, th[2] / 2])


def imu_factor(pre, sqrt_info, lba, lbg, pose_i, sb_i, pose_j, sb_j):
    """IMUFactor::Evaluate: residual (15) and the Jacobian w.r.t. the LOCAL parameters pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9."""
    dp, dq, dv, Jp, P, sdt = pre
    r = imu_residual_raw(pre, G, pose_i, sb_i, pose_j, sb_j, lba, lbg)
    Pi, Qi, Pj, Qj = pose_i[:3], wq(pose_i), pose_j[:3], wq(pose_j)
    Vi, Bgi, Vj = sb_i[:3], sb_i[6:9], sb_j[:3]
    dp_dba, dp_dbg, dq_dbg, dv_dba, dv_dbg = Jp[0:3, 9:12], Jp[0:3, 12:15], Jp[3:6, 12:15], Jp[6:9, 9:12], Jp[6:9, 12:15]
    RiT = q2R(qconj(Qi))
    cdq = qmul(dq, deltaQ(dq_dbg @ (Bgi - lbg)))
    J = np.zeros((15, 30))
    # pose_i
    J[0:3, 0:3] = -RiT
    J[0:3, 3:6] = skew(RiT @ (0.5 * G * sdt * sdt + Pj - Pi - Vi * sdt))
    J[3:6, 3:6] = -(Qleft(qmul(qconj(Qj), Qi)) @ Qright(cdq))[1:, 1:]
    J[6:9, 3:6] = skew(RiT @ (G * sdt + Vj - Vi))
    # speedbias_i (V, BA, BG)
    J[0:3, 6:9] = -RiT * sdt
    J[0:3, 9:12] = -dp_dba
    J[0:3, 12:15] = -dp_dbg
    J[3:6, 12:15] = -Qleft(qmul(qmul(qconj(Qj), Qi), dq))[1:, 1:] @ dq_dbg   # the UNCORRECTED delta_q (imu_factor.h:123-125)
    J[6:9, 6:9] = -RiT
    J[6:9, 9:12] = -dv_dba
    J[6:9, 12:15] = -dv_dbg
    J[9:12, 9:12] = -np.eye(3)
    J[12:15, 12:15] = -np.eye(3)
    # pose_j
    J[0:3, 15:18] = RiT
    J[3:6, 18:21] = Qleft(qmul(qmul(qconj(cdq), qconj(Qi)), Qj))[1:, 1:]
    # speedbias_j
    J[6:9, 21:24] = RiT
    J[9:12, 24:27] = np.eye(3)
    J[12:15, 27:30] = np.eye(3)
    return sqrt_info @ r, sqrt_info @ J


class Problem:
    """The window as a dense least-squares problem over the local (tangent) columns: pose 6 x 11 | speed-bias 9 x 11 | inv depth."""

    def __init__(self, a):
        self.a = a
        self.nf = int(a["n_feat"])
        self.n = 165 + self.nf
        self.pre, self.sqrt = [], []
        for j in range(10):
            ns = int(a["imu_n"][j])
            pre = preintegrate(a["imu_acc"][j, : ns + 1], a["imu_gyr"][j, : ns + 1], a["imu_dt"][j, :ns], a["imu_lin_ba"][j], a["imu_lin_bg"][j], NOISE)
            self.pre.append(pre)
            self.sqrt.append(np.linalg.cholesky(np.linalg.inv(pre[4])).T)   # LLT(cov^-1).matrixL().transpose()
        self.pn = int(a["prior_n"])

    def evaluate(self, x, want_jac):
        """x = dict(pose [11,7], sb [11,9], lam [nf]).  Returns cost (with the robust loss) and, if asked, the CORRECTED stacked
        residual r and Jacobian J (what Ceres' evaluator hands the minimizer)."""
        a = self.a
        rows_r, rows_J, cost = [], [], 0.0
        ex = a["ex_pose"]
        if self.pn > 0:
            n, dx = self.pn, np.zeros(self.pn)
            off, cols = 0, []
            for k in range(int(a["prior_nblk"])):
                kind, fr, x0 = int(a["prior_blk_kind"][k]), int(a["prior_blk_frame"][k]), a["prior_x0"][k]
                if kind == 1:
                    dx[off:off + 9] = x["sb"][fr] - x0[:9]
                    cols.append((off, 66 + 9 * fr, 9))
                    off += 9
                else:
                    cur = x["pose"][fr] if kind == 0 else ex
                    dx[off:off + 3] = cur[:3] - x0[:3]
                    d = qmul(qconj(wq(x0)), wq(cur))
                    dx[off + 3:off + 6] = 2.0 * d[1:] if d[0] >= 0 else -2.0 * d[1:]
                    if kind == 0:
                        cols.append((off, 6 * fr, 6))
                    off += 6
            J0 = a["prior_J"][:n, :n]
            r = a["prior_r"][:n] + J0 @ dx
            cost += 0.5 * r @ r
            if want_jac:
                J = np.zeros((n, self.n))
                for o, c, sz in cols:
                    J[:, c:c + sz] = J0[:, o:o + sz]
                rows_r.append(r), rows_J.append(J)
        for i in range(10):
            if self.pre[i][5] > 10.0:
                continue
            r, Jl = imu_factor(self.pre[i], self.sqrt[i], a["imu_lin_ba"][i], a["imu_lin_bg"][i], x["pose"][i], x["sb"][i], x["pose"][i + 1], x["sb"][i + 1])
            cost += 0.5 * r @ r
            if want_jac:
                J = np.zeros((15, self.n))
                J[:, 6 * i:6 * i + 6], J[:, 66 + 9 * i:66 + 9 * i + 9] = Jl[:, 0:6], Jl[:, 6:15]
                J[:, 6 * (i + 1):6 * (i + 1) + 6], J[:, 66 + 9 * (i + 1):66 + 9 * (i + 1) + 9] = Jl[:, 15:21], Jl[:, 21:30]
                rows_r.append(r), rows_J.append(J)
        for e in range(self.nf):
            s, no, ob = int(a["feat_start"][e]), int(a["feat_nobs"][e]), int(a["feat_obs_begin"][e])
            pts_i = np.array([*a["obs_xy"][ob], 1.0])
            for t in range(1, no):
                pts_j = np.array([*a["obs_xy"][ob + t], 1.0])
                r, Ji, Jj, _, Je = projection_factor(x["pose"][s], x["pose"][s + t], ex, x["lam"][e], pts_i, pts_j, SQRT_INFO)
                Jl = np.zeros((2, self.n))
                Jl[:, 6 * s:6 * s + 6], Jl[:, 6 * (s + t):6 * (s + t) + 6], Jl[:, 165 + e] = Ji, Jj, Je
                rc, Jc, c = cauchy_correct(r, Jl)
                cost += c
                if want_jac:
                    rows_r.append(rc), rows_J.append(Jc)
        if not want_jac:
            return cost
        return cost, np.concatenate(rows_r), np.vstack(rows_J)

    def plus(self, x, d):
        out = dict(pose=x["pose"].copy(), sb=x["sb"].copy(), lam=x["lam"].copy())
        for f in range(11):
            out["pose"][f, :3] = x["pose"][f, :3] + d[6 * f:6 * f + 3]
            q = qmul(wq(x["pose"][f]), deltaQ(d[6 * f + 3:6 * f + 6]))
            q = q / np.sqrt(q @ q)
            out["pose"][f, 3:] = [q[1], q[2], q[3], q[0]]
        out["sb"] = x["sb"] + d[66:165].reshape(11, 9)
        out["lam"] = x["lam"] + d[165:]
        return out

    @staticmethod
    def ambient(x):
        return np.concatenate([x["pose"].ravel(), x["sb"].ravel(), x["lam"]])


def trust_region_solve(P, x, opt):
    """Ceres 1.14 TrustRegionMinimizer::Minimize with DoglegStrategy (traditional), dense linear algebra."""
    n = P.n
    radius, mu = opt["initial_trust_region_radius"], 1e-8
    min_mu, max_mu, mu_inc = 1e-8, 1.0, 10.0
    x_cost, r, J = P.evaluate(x, True)
    scale = 1.0 / (1.0 + np.sqrt((J * J).sum(0)))         # Jacobi scaling from the FIRST Jacobian
    J = J * scale
    x_norm = np.linalg.norm(P.ambient(x))
    trace = dict(cost=[], radius=[], accepted=[], rho=[], model_cost_change=[], step_norm=[], mu=[], kind=[])
    initial_cost, ref_cost = x_cost, x_cost
    reuse, iteration, num_invalid, termination = False, 0, 0, 0
    step_ok, gmax = True, None
    gn = dgrad = diag = alpha = None

    def grad_max_norm():
        g = (J.T @ r) / scale                                # unscaled gradient
        return np.abs(P.ambient(x) - P.ambient(P.plus(x, -g))).max()

    gmax = grad_max_norm()
    while True:
        if iteration > 0:
            trace["cost"].append(x_cost), trace["radius"].append(radius), trace["accepted"].append(step_ok)
        if iteration >= opt["max_num_iterations"]:
            termination = 0
            break
        if step_ok and gmax <= opt["gradient_tolerance"]:
            termination = 1
            break
        if radius <= opt["min_trust_region_radius"]:
            termination = 4
            break
        iteration += 1
        step_ok = False
        solved = True
        if not reuse:
            reuse = True
            diag = np.sqrt(np.clip((J * J).sum(0), opt["min_lm_diagonal"], opt["max_lm_diagonal"]))
            dgrad = (J.T @ r) / diag
            u = dgrad / diag
            alpha = (dgrad @ dgrad) / ((J @ u) @ (J @ u))
            solved = False
            while mu < max_mu:
                H = J.T @ J + np.diag(mu * diag * diag)
                try:
                    L = np.linalg.cholesky(H)
                except np.linalg.LinAlgError:
                    mu *= mu_inc
                    continue
                y = np.linalg.solve(L.T, np.linalg.solve(L, J.T @ r))
                if not np.isfinite(y).all():
                    mu *= mu_inc
                    continue
                gn = -diag * y
                solved = True
                break
        valid, mcc, kind = False, 0.0, -1
        if solved:
            gnorm, gn_norm = np.linalg.norm(dgrad), np.linalg.norm(gn)
            if gn_norm <= radius:
                step, dnorm, kind = gn.copy(), gn_norm, 0
            elif gnorm * alpha >= radius:
                step, dnorm, kind = -(radius / gnorm) * dgrad, radius, 1
            else:
                b_dot_a = -alpha * (dgrad @ gn)
                a2 = (alpha * gnorm) ** 2
                bma2 = a2 - 2 * b_dot_a + gn_norm ** 2
                c = b_dot_a - a2
                d = np.sqrt(c * c + bma2 * (radius ** 2 - a2))
                beta = (d - c) / bma2 if c <= 0 else (radius ** 2 - a2) / (d + c)
                step = (-alpha * (1.0 - beta)) * dgrad + beta * gn
                dnorm, kind = np.linalg.norm(step), 2
            step = step / diag
            Js = J @ step
            mcc = -(Js @ (r + Js / 2.0))
            valid = mcc > 0.0
        trace["mu"].append(mu), trace["kind"].append(kind)
        if not valid:
            trace["rho"].append(np.nan), trace["model_cost_change"].append(mcc), trace["step_norm"].append(np.nan)
            num_invalid += 1
            if num_invalid >= opt["max_num_consecutive_invalid_steps"]:
                termination = 5
                break
            mu *= mu_inc
            reuse = False
            continue
        num_invalid = 0
        cand = P.plus(x, step * scale)
        cand_cost = P.evaluate(cand, False)
        step_norm = np.linalg.norm(P.ambient(x) - P.ambient(cand))
        trace["model_cost_change"].append(mcc), trace["step_norm"].append(dnorm)
        if step_norm <= opt["parameter_tolerance"] * (x_norm + opt["parameter_tolerance"]):
            trace["rho"].append(np.nan)
            termination = 2
            break
        if abs(x_cost - cand_cost) <= opt["function_tolerance"] * x_cost:
            trace["rho"].append(np.nan)
            termination = 3
            break
        rho = (ref_cost - cand_cost) / mcc
        trace["rho"].append(rho)
        if rho > opt["min_relative_decrease"]:
            x = cand
            x_norm = np.linalg.norm(P.ambient(x))
            x_cost, r, J = P.evaluate(x, True)
            J = J * scale
            gmax = grad_max_norm()
            step_ok = True
            if rho < 0.25:
                radius *= 0.5
            if rho > 0.75:
                radius = max(radius, 3.0 * dnorm)
            mu = max(min_mu, 2.0 * mu / mu_inc)
            reuse = False
            ref_cost = cand_cost
        else:
            radius *= 0.5
            reuse = True
    return x, dict(termination=termination, num_iterations=iteration, initial_cost=initial_cost, final_cost=x_cost, **{k: np.array(v) for k, v in trace.items()})


# (seed, attitude noise, position noise [m]) of the first four: one window structure, four start points
CASES = [(5, 0.7, 1.0), (3, 0.6, 1.5), (7, 0.8, 1.0), (99, 0.35, 0.6)]
# round 4: eight more, over the window STRUCTURES the first four do not vary ("8 windows is thin") -
# (seed, attitude noise, position noise, tracks, features, trajectory id, with the prior)
CASES_R4 = [(11, 0.5, 0.8, "dense", 24, 100, True), (12, 0.3, 0.5, "sparse", 40, 200, True), (13, 0.6, 1.0, "sparse", 30, 300, False),
            (14, 0.05, 0.1, "dense", 12, 400, True), (15, 0.9, 2.0, "sparse", 20, 500, True), (16, 0.02, 0.02, "sparse", 14, 600, False),
            (17, 0.7, 1.0, "dense", 40, 700, True), (18, 0.4, 0.3, "sparse", 60, 800, True)]


def make_case(seed, sq, sp, tracks="sparse", n_feat=14, first_id=4242, with_prior=True):
    """One 11-frame window (the first four cases: 14 ragged tracks and the synthetic 75-row prior); frames 1.. knocked far enough off
    for some steps to be rejected (cases 0-2; case 3 is a clean run: every step accepted)."""
    synth = importlib.import_module(PKG + ".synth")
    if n_feat == 14 and first_id == 4242:
        w = synth.make_windows(1, first_id=4242, tracks="sparse", n_feat=14, max_feat=16, max_obs=176)
    else:
        w = synth.make_windows(1, first_id=first_id, tracks=tracks, n_feat=n_feat, max_feat=64, max_obs=704, with_prior=with_prior)
    rng = np.random.default_rng(seed)
    q = w.a["pose"][0, 1:, 3:] + rng.normal(0, sq, w.a["pose"][0, 1:, 3:].shape)
    w.a["pose"][0, 1:, 3:] = q / np.linalg.norm(q, axis=-1, keepdims=True)
    w.a["pose"][0, 1:, :3] += rng.normal(0, sp, w.a["pose"][0, 1:, :3].shape)
    return w


OPT = dict(initial_trust_region_radius=1e4, min_trust_region_radius=1e-32, max_num_iterations=12, gradient_tolerance=1e-10,
           parameter_tolerance=1e-8, function_tolerance=1e-6, min_relative_decrease=1e-3, min_lm_diagonal=1e-6, max_lm_diagonal=1e32,
           max_num_consecutive_invalid_steps=5)


def main():
    cases = [tuple(c) for c in CASES] + CASES_R4
    out = {"n_cases": np.int64(len(cases))}
    out.update({"opt_" + k: np.float64(v) for k, v in OPT.items()})
    for c, spec in enumerate(cases):
        w = make_case(*spec)
        a = {k: v[0] for k, v in w.a.items()}
        P = Problem(a)
        x0 = dict(pose=a["pose"].copy(), sb=a["speedbias"].copy(), lam=a["inv_depth"][: P.nf].copy())
        x, tr = trust_region_solve(P, x0, OPT)
        print(f"case {c}: iterations {tr['num_iterations']} termination {tr['termination']} accepted {tr['accepted'].astype(int).tolist()}")
        print("   cost", tr["initial_cost"], "->", tr["final_cost"], " kinds (0 GN, 1 Cauchy point on the boundary, 2 dogleg)", tr["kind"].tolist())
        out.update({f"c{c}_in_" + k: v for k, v in w.a.items()})
        out.update({f"c{c}_dim_" + k: np.int64(v) for k, v in w.dims.items()})
        out.update({f"c{c}_sol_pose": x["pose"], f"c{c}_sol_speedbias": x["sb"], f"c{c}_sol_inv_depth": x["lam"]})
        out.update({f"c{c}_trace_" + k: v for k, v in tr.items()})
    np.savez_compressed(os.path.join(HERE, "solve_trace.npz"), **out)
    print("wrote solve_trace.npz")


if __name__ == "__main__":
    main()
