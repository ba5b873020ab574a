"""MARGIN_OLD prior parity against an extended-precision ARBITER.

The reference computes the new prior through two symmetric eigen-decompositions of matrices whose entries span 1e12 .. 1e0
(marginalization_factor.cpp:267-291), with a clamp at 1e-8 that sits far below the FP64 noise floor of that computation.  The
FP64 oracle follows that algorithm literally and is therefore only reproducible to 1e-5 .. 1e-4; comparing the GPU with it says
nothing about which of the two is right.  oracle/avm_truth.cpp is the same restatement compiled with __float128 as its scalar
type, from the pre-integration on: the value the reference's algorithm DEFINES for the given FP64 inputs.  Every implementation
is measured against the arbiter evaluated on exactly the state that implementation marginalized at (the state it returned).

Measured (MI355X, 6 windows per row, worst window; profiles/r03_prior_truth.md):
  with a prior:   |GPU - truth|  H 1.3e-7, Jacobi-scaled H 2.6e-6, scaled g 1e-12, cost 7e-8
                  |oracle - truth|  H 2.0e-5,             1.4e-4,          5.6e-7,      1.4e-5
The GPU's different algorithm (scalar pivots for the depths, Cholesky-based square root) is one to two
orders of magnitude CLOSER to the exact result than the reference's algorithm run in FP64."""
import importlib

import numpy as np
import pytest

from helpers import abi, buffers, rel, synth
from marg_sensitivity import distance_to_truth, install_prior, marginalize_at, truth_marginalize, ulp_perturbed

est_m = importlib.import_module("anticipated-vins-mono_amd.estimator")
METRICS = ("H_rel", "H_scaled", "g_scaled", "cost_rel")


def _fmt(d):
    return " ".join(f"{k} {v:.1e}" for k, v in d.items())


# ---------------------------------------------------------------- CPU tier: the arbiter itself
def test_truth_is_an_exact_square_root_of_its_own_schur_complement(oracle):
    """J^T J and J^T r formed in binary128 reproduce the Schur complement A, b that went into the eigen square root to FP64
    round-off (each side is rounded once): the two eigen-decompositions and the clamp lose nothing above 1e-15."""
    o = abi.default_options()
    w = synth.make_windows(3, first_id=300, tracks="sparse", n_feat=60, max_feat=150)
    oracle.window_solve(o, w, buffers.PriorOutArrays.alloc(3), buffers.summary_alloc(3))
    pt, diag = truth_marginalize(w, o)
    for i in range(3):
        t = diag[i]
        assert (t["ev_mm"] > 1e-8).all()                                  # nothing clamped on the dropped side
        d = 1.0 / np.sqrt(np.diag(t["A"]))
        assert np.abs((t["H"] - t["A"]) * d[:, None] * d[None, :]).max() < 1e-12
        assert np.abs((t["g"] - t["b"]) * d).max() < 1e-12 * np.abs(t["b"] * d).max()
        n = t["n"]
        J = pt.a["J"][i, :n, :n]
        assert rel(J.T @ J, t["H"]) < 1e-13                               # the FP64 rounding of J is all that separates them


def test_oracle_distance_to_truth(oracle):
    """How far the reference's algorithm in FP64 (the oracle) lands from its own exact value: 1e-8 .. 2e-5 in H, up to
    1.4e-4 in Jacobi-scaled entries.  This is the yardstick the GPU tier uses, and the reason a 1e-6 comparison
    GPU-vs-oracle cannot be the criterion for the prior."""
    o = abi.default_options()
    worst = {k: 0.0 for k in METRICS}
    for tracks, nf in (("sparse", 60), ("dense", 150)):
        w = synth.make_windows(3, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
        oracle.window_solve(o, w, buffers.PriorOutArrays.alloc(3), buffers.summary_alloc(3))
        po, at = marginalize_at(w, o)
        _, diag = truth_marginalize(at, o)
        for i in range(3):
            d = distance_to_truth(po, diag, i)
            print(f"\n[oracle - truth] {tracks} {nf} window {i}: {_fmt(d)}")
            for k in METRICS:
                worst[k] = max(worst[k], d[k])
    assert worst["H_rel"] < 2e-4 and worst["H_scaled"] < 2e-3 and worst["g_scaled"] < 1e-5 and worst["cost_rel"] < 2e-4, worst
    assert worst["H_rel"] > 1e-9                                          # ... and it is not exact: the arbiter resolves the difference


def test_which_eigenvalues_the_clamp_takes_is_rounding_dependent_in_fp64(oracle):
    """A window without a prior has gauge freedom and unconstrained biases: exact zeros in the Schur complement.  The exact
    computation clamps all of them; the FP64 oracle sees +-1e-10 .. 1e-4 there and keeps the ones that land above 1e-8."""
    o = abi.default_options()
    w = synth.make_windows(4, first_id=300, tracks="sparse", n_feat=80, max_feat=150, with_prior=False)
    oracle.window_solve(o, w, buffers.PriorOutArrays.alloc(4), buffers.summary_alloc(4))
    po, at = marginalize_at(w, o)
    _, diag = truth_marginalize(at, o)
    n_truth, n_oracle = [], []
    for i in range(4):
        n = diag[i]["n"]
        n_truth.append(int((diag[i]["ev_rr"] <= 1e-8).sum()))
        n_oracle.append(int((np.abs(po.a["J"][i, :n, :n]).max(1) == 0).sum()))   # a clamped eigenvalue leaves a zero row
        assert np.abs(diag[i]["ev_rr"][: n_truth[-1]]).max() < 1e-15               # exact zeros, resolved as such in binary128
    print("\n[clamped eigenvalues of A'] truth", n_truth, "oracle (FP64)", n_oracle)
    assert min(n_truth) >= 4 and all(a <= b for a, b in zip(n_oracle, n_truth))
    assert n_oracle != n_truth                                            # (measured: 2 .. 9 of them survive in FP64)


# ---------------------------------------------------------------- GPU tier
FLOOR = dict(H_rel=1e-6, H_scaled=1e-5, g_scaled=1e-9, cost_rel=1e-6)      # below these the comparison is moot


def _zero_baseline_window():
    """Windows whose joint Amm (pose 0, speed-bias 0 AND the inverse depths of the features that start in frame 0) is rank deficient:
    frame 1 has exactly frame 0's pose and a few start-0 features are seen in frames 0 and 1 only - no baseline, so the depth has
    no influence on the residual (the point moves along its own ray): E^T E = 0 for them, exact zero eigenvalues of Amm."""
    w = synth.make_windows(3, first_id=900, tracks="sparse", n_feat=60, max_feat=150)
    for b in range(3):
        w.a["pose"][b, 1] = w.a["pose"][b, 0]            # (the same camera centre needs the same attitude too: tic != 0)
        n = int(w.a["n_feat"][b])
        z = [e for e in range(n) if w.a["feat_start"][b, e] == 0][:3]
        assert len(z) == 3
        for e in z:
            w.a["feat_nobs"][b, e] = 2
    return w


def test_rank_deficient_joint_amm_oracle_vs_truth(oracle):
    """a deterministic case where the 1e-8 clamp of marginalization_factor.cpp:272 acts on the DROPPED side (feature block
    included).  The exact computation finds the zero eigenvalues as exact zeros; the FP64 oracle's joint eigen-decomposition clamps them too
    (they come out as +-1e-4 around 0, mostly below 1e-8 only by luck of the sign) and lands within its usual distance of the truth."""
    o = abi.default_options()
    w = _zero_baseline_window()
    po, at = marginalize_at(w, o)
    _, diag = truth_marginalize(at, o)
    for i in range(3):
        nz = int((diag[i]["ev_mm"] <= 1e-8).sum())   # (the three features above and every other two-view track between frames 0 and 1)
        assert nz >= 3 and np.abs(diag[i]["ev_mm"][:nz]).max() < 1e-12 and diag[i]["ev_mm"][nz] > 1e-3   # exact zeros, resolved as such
        d = distance_to_truth(po, diag, i)
        print(f"\n[rank-deficient Amm] window {i}: oracle - truth {_fmt(d)}")
        assert d["H_rel"] < 1e-3 and d["g_scaled"] < 1e-3, d


@pytest.mark.gpu
def test_rank_deficient_joint_amm_gpu_vs_truth(ctx, oracle):
    """The GPU eliminates the inverse depths as scalar pivots (their block of Amm is diagonal) with the same clamp, then the 15 x 15 block:
    on a rank-deficient joint Amm that is the joint pseudo-inverse of the reference - asserted against the exact result at a FIXED
    tolerance, and against the FP64 oracle's distance."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    w = _zero_baseline_window()
    po, at_o = marginalize_at(w, o)
    pg, at_g = marginalize_at(w, o, estimator=E)
    _, diag_o = truth_marginalize(at_o, o)
    _, diag_g = truth_marginalize(at_g, o)
    for i in range(3):
        assert int((diag_g[i]["ev_mm"] <= 1e-8).sum()) >= 3
        do, dg = distance_to_truth(po, diag_o, i), distance_to_truth(pg, diag_g, i)
        print(f"\n[rank-deficient Amm] window {i}: oracle {_fmt(do)} | gpu {_fmt(dg)}")
        assert dg["H_rel"] < 5e-6 and dg["H_scaled"] < 1e-5 and dg["g_scaled"] < 1e-8 and dg["cost_rel"] < 1e-6, dg   # (measured 1.2e-6 / 1.6e-6 / 1.4e-9 / 8.6e-9)
        for k in METRICS:
            assert dg[k] <= max(do[k], FLOOR[k]), (i, k, dg[k], do[k])


@pytest.mark.gpu
@pytest.mark.parametrize("tracks,nf", [("sparse", 60), ("dense", 150), ("sparse", 150)])
def test_gpu_prior_is_closer_to_the_truth_than_the_fp64_oracle(ctx, oracle, tracks, nf):
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 6
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
    oracle.window_solve(o, w, buffers.PriorOutArrays.alloc(B), buffers.summary_alloc(B))
    po, at_o = marginalize_at(w, o)
    pg, at_g = marginalize_at(w, o, estimator=E)
    assert np.array_equal(pg.a["n"], po.a["n"]) and np.array_equal(pg.a["blk_kind"], po.a["blk_kind"]) and np.array_equal(pg.a["blk_frame"], po.a["blk_frame"])
    _, diag_o = truth_marginalize(at_o, o)
    _, diag_g = truth_marginalize(at_g, o)      # (each side against the exact result at the state IT marginalized at)
    wo, wg = {k: 0.0 for k in METRICS}, {k: 0.0 for k in METRICS}
    for i in range(B):
        do, dg = distance_to_truth(po, diag_o, i), distance_to_truth(pg, diag_g, i)
        print(f"\n[prior vs truth] {tracks} {nf} window {i}: clamped {int((diag_g[i]['ev_rr'] <= 1e-8).sum())} | oracle {_fmt(do)} | gpu {_fmt(dg)}")
        for k in METRICS:
            assert dg[k] <= max(do[k], FLOOR[k]), (i, k, dg[k], do[k])
            wo[k], wg[k] = max(wo[k], do[k]), max(wg[k], dg[k])
    for k in METRICS:
        assert wg[k] <= wo[k], (k, wg[k], wo[k])                           # worst window: the GPU is the closer one, no floor
    # the north-star tolerance, against the exact result
    assert wg["H_rel"] < 1e-6 and wg["H_scaled"] < 1e-5 and wg["g_scaled"] < 1e-9 and wg["cost_rel"] < 1e-6, wg


@pytest.mark.gpu
@pytest.mark.parametrize("tracks,nf", [("sparse", 80), ("dense", 150)])
def test_rank_deficient_prior_against_the_truth(ctx, oracle, tracks, nf):
    """No prior yet (the first marginalization of a run): 16 .. 30 exact zeros in A' (gauge, unconstrained biases / extrinsic).
    The exact computation clamps them all; in FP64 they come out as noise of up to 0.25 (oracle) / 0.02 (GPU), partly above the
    smallest genuine eigenvalues, so no FP64 implementation can clamp exactly that set.  Asserted: the GPU is the closer one."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 6
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150, with_prior=False)
    oracle.window_solve(o, w, buffers.PriorOutArrays.alloc(B), buffers.summary_alloc(B))
    po, at_o = marginalize_at(w, o)
    pg, at_g = marginalize_at(w, o, estimator=E)
    _, diag_o = truth_marginalize(at_o, o)
    _, diag_g = truth_marginalize(at_g, o)
    wo, wg = {k: 0.0 for k in METRICS}, {k: 0.0 for k in METRICS}
    for i in range(B):
        n = diag_g[i]["n"]
        n_truth = int((diag_g[i]["ev_rr"] <= 1e-8).sum())
        n_gpu = int((np.abs(pg.a["J"][i, :n, :n]).max(1) == 0).sum())
        do, dg = distance_to_truth(po, diag_o, i), distance_to_truth(pg, diag_g, i)
        print(f"\n[rank-deficient prior vs truth] {tracks} {nf} window {i}: clamped truth {n_truth} gpu {n_gpu} | oracle {_fmt(do)} | gpu {_fmt(dg)}")
        n_oracle = int((np.abs(po.a["J"][i, :n, :n]).max(1) == 0).sum())
        # FP64 cannot resolve all of the exact zeros (the unconstrained bias directions carry 1e-16 x 1e12 of noise, above
        # the smallest genuine eigenvalues): the GPU recognises at least as many of them as the reference's algorithm in FP64
        assert n_oracle <= n_gpu <= n_truth, (i, n_oracle, n_gpu, n_truth)
        for k in METRICS:
            wo[k], wg[k] = max(wo[k], do[k]), max(wg[k], dg[k])
    for k in METRICS:
        assert wg[k] <= wo[k], (k, wg[k], wo[k])
    assert wg["H_rel"] < 1e-6, wg


def _chained(start, prior, solve):
    c = start.copy()
    install_prior(c, prior)
    return c, solve(c)


@pytest.mark.gpu
@pytest.mark.parametrize("tracks,nf", [("sparse", 60), ("dense", 150)])
def test_what_a_solve_sees_of_the_prior_against_the_truth(ctx, oracle, tracks, nf):
    """The next solve (the FP64 oracle's solver, from the same start) with the exact prior, with the oracle's prior and with
    the GPU's prior: the GPU's prior leads to the solution of the exact prior within the north-star 1e-6 - the oracle's own
    prior does not always - and the GPU's solver on the GPU's prior (the product end to end) is within 1e-6 as well."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 4
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
    wg, wo = w.copy(), w.copy()
    E.optimization(wg)
    pg, po = E.last_marginalization_info, buffers.PriorOutArrays.alloc(B)
    oracle.window_solve(o, wo, po, buffers.summary_alloc(B))
    assert rel(wg.a["pose"], wo.a["pose"]) < 1e-9                          # (the first solves agree far below what follows)
    pt_o, _ = truth_marginalize(wo, o)     # exact prior at the oracle's solution
    pt_g, _ = truth_marginalize(wg, o)     # exact prior at the GPU's solution
    o2 = abi.default_options()
    o2.marginalization_flag = abi.MARGIN_NONE
    E2 = est_m.Estimator(ctx=ctx, options=o2)

    def cpu(c):
        s = buffers.summary_alloc(B)
        oracle.window_solve(o2, c, None, s)
        return s

    def gpu(c):
        return buffers.summary_to_numpy(E2.optimization(c)).copy()

    tt, stt = _chained(wo, pt_o, cpu)      # the reference: exact prior, FP64 solver
    oo, soo = _chained(wo, po, cpu)        # the oracle end to end
    og, sog = _chained(wo, pg, cpu)        # GPU prior, oracle solver, oracle start
    gt, sgt = _chained(wg, pt_g, gpu)      # exact prior at the GPU's state, GPU solver
    gg, sgg = _chained(wg, pg, gpu)        # the product end to end
    for k in ("pose", "speedbias"):
        d_o, d_g, d_e = rel(oo.a[k], tt.a[k]), rel(og.a[k], tt.a[k]), rel(gg.a[k], gt.a[k])
        print(f"\n[chained vs truth {tracks} {nf}] {k}: oracle prior {d_o:.2e}  gpu prior {d_g:.2e}  gpu end to end (vs exact prior at its own state) {d_e:.2e}"
              f"  gpu end to end vs truth chain {rel(gg.a[k], tt.a[k]):.2e}")
        assert d_g < 1e-6, (k, d_g)                                        # north-star tolerance, no escape clause
        assert d_g <= max(d_o, 1e-8), (k, d_g, d_o)
        assert d_e < 1e-6, (k, d_e)
        assert rel(gg.a[k], tt.a[k]) < 1e-6, (k, rel(gg.a[k], tt.a[k]))
    for s in (sog, sgg, sgt):
        assert np.array_equal(s["accept_mask"], stt["accept_mask"]) and np.array_equal(s["num_iterations"], stt["num_iterations"])


class _Oracle:
    def __init__(self, oracle, opt, exact_prior=False):
        self.o, self.opt, self.exact = oracle, opt, exact_prior

    def solve(self, w):
        p, s = buffers.PriorOutArrays.alloc(w.n_windows), buffers.summary_alloc(w.n_windows)
        self.o.window_solve(self.opt, w, p, s)
        if self.exact:                     # the FP64 solver, the marginalization in binary128 at the state it returned
            p, _ = truth_marginalize(w, self.opt)
        return p, s

    def roll(self, w):
        assert self.o.slide_window(w, abi.MARGIN_OLD, True, 5.0) == 0

    def new_frame(self, w):
        self.o.triangulate(w, 5.0)
        self.o.imu_propagate(w, np.array(list(self.opt.g)))


class _Gpu:
    def __init__(self, ctx, opt):
        self.E = est_m.Estimator(ctx=ctx, options=opt)

    def solve(self, w):
        s = buffers.summary_to_numpy(self.E.optimization(w))
        return self.E.last_marginalization_info, s

    def roll(self, w):
        self.E.slideWindow(w, abi.MARGIN_OLD, True, 5.0)

    def new_frame(self, w):
        self.E.triangulate(w, 5.0)
        self.E.imu_propagate(w)


AMP_EPS = 1e-8


def _perturbed_state(w, eps=AMP_EPS, seed=777):
    """A copy of the window with its solved states moved by eps (positions, velocities: absolute, they are O(1); attitudes: a
    rotation of eps rad; biases: eps of their scales 0.02 / 0.002)."""
    rng = np.random.default_rng(seed)
    p = w.copy()
    p.a["pose"][..., :3] += eps * rng.uniform(-1, 1, p.a["pose"][..., :3].shape)
    for f in range(p.a["pose"].shape[1]):
        R = synth.R_from_quat(p.a["pose"][0, f, 3:]) @ synth._rot_zyx(*(eps * rng.uniform(-1, 1, 3)))
        p.a["pose"][0, f, 3:] = synth.quat_from_R(R)
    sc = np.array([1.0] * 3 + [0.02] * 3 + [0.002] * 3)
    p.a["speedbias"] += eps * sc * rng.uniform(-1, 1, p.a["speedbias"].shape)
    return p


def _stream(seq_id, backend, n_images, amplification=None, **seq_kw):
    """amplification (a list, T stream only): per frame k >= 1 the one-frame propagation factor of a state deviation - the solved
    state of frame k - 1 moved by AMP_EPS, the same frame transition (exact marginalization at the moved state, roll, new image,
    triangulation, dead-reckoning) and the FP64 solve of frame k, against the unmoved stream: |delta state_k| / AMP_EPS.  Below 1 a
    stream forgets what separated two implementations; far above 1 it multiplies it."""
    seq = synth.Sequence(seq_id, **seq_kw)
    w, ids = seq.first_window()
    out = []
    pending = None
    for k in range(n_images):
        prior, s = backend.solve(w)
        if amplification is not None:
            amplification.append(0.0 if pending is None else
                                 max(rel(pending.a["pose"], w.a["pose"]), rel(pending.a["speedbias"], w.a["speedbias"])) / AMP_EPS)
            pending = None
            if k + 1 < n_images:
                wp = _perturbed_state(w)
                pp, _ = truth_marginalize(wp, backend.opt)
                backend.roll(wp)
                seq.next_image(wp, ids, k)
                install_prior(wp, pp)
                backend.new_frame(wp)
                o2 = abi.Options.from_buffer_copy(bytes(backend.opt))
                o2.marginalization_flag = abi.MARGIN_NONE
                backend.o.window_solve(o2, wp, None, buffers.summary_alloc(1))
                pending = wp
        npr = int(prior.a["n"][0])
        out.append(dict(pose=w.a["pose"].copy(), speedbias=w.a["speedbias"].copy(), n_feat=int(w.a["n_feat"][0]),
                        it=int(s["num_iterations"][0]), acc=int(s["accept_mask"][0]), term=int(s["termination"][0]),
                        clamped=int((np.abs(prior.a["J"][0, :npr, :npr]).max(1) == 0).sum()) if npr > 0 else 0))  # directions the prior's clamp dropped
        backend.roll(w)
        ids = seq.next_image(w, ids, k)
        install_prior(w, prior)
        backend.new_frame(w)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("seq_id", [0, 1])
def test_ten_frame_stream_against_the_exact_prior_stream(ctx, oracle, seq_id):
    """Ten images through optimization() + slideWindow(), the prior handed from frame to frame (estimator.cpp:996-1107 in a
    loop).  Three streams: T = FP64 solver with every marginalization done in binary128 (what the reference's algorithm defines,
    up to the solver's 1e-11), O = the FP64 oracle, G = the GPU.  At every frame the GPU is at least as close to T as the
    oracle is, and within the north-star 1e-6 of it."""
    n = 10
    o = abi.default_options()
    T = _stream(seq_id, _Oracle(oracle, o, exact_prior=True), n)
    O = _stream(seq_id, _Oracle(oracle, o), n)
    G = _stream(seq_id, _Gpu(ctx, o), n)
    worst_g = worst_o = 0.0
    for k in range(n):
        assert G[k]["n_feat"] == T[k]["n_feat"] and G[k]["it"] == T[k]["it"] and G[k]["acc"] == T[k]["acc"] and G[k]["term"] == T[k]["term"], (k, G[k], T[k])
        for key in ("pose", "speedbias"):
            dg, do = rel(G[k][key], T[k][key]), rel(O[k][key], T[k][key])
            print(f"[stream {seq_id} vs exact-prior stream] frame {k} {key}: gpu {dg:.2e}  oracle {do:.2e}  features {T[k]['n_feat']}")
            worst_g, worst_o = max(worst_g, dg), max(worst_o, do)
            assert dg < 1e-6 or dg <= do, (k, key, dg, do)
    print(f"[stream {seq_id}] worst distance to the exact-prior stream over {n} frames: gpu {worst_g:.2e}  oracle {worst_o:.2e}")
    assert worst_g <= max(worst_o, 1e-6)


# measured on MI355X in round 6 (printed by the test below), 160 frames of eight streams: default clamp 120 frames within 1e-6 of the exact-prior stream,
# worst 1.6e-5; reference-literal clamp 112 frames, worst 1.6e-5; either GPU stream 4.4e-4 from the FP64 oracle's stream at worst, which is the oracle's
# own worst distance from the exact-prior stream (4.4e-4).  Frame counts asserted with a margin of eight frames, distances at three times the measurement.
STREAM_BOUNDS = {"default_frames_within_1e-6": 112, "default_worst_vs_exact": 5e-5, "literal_frames_within_1e-6": 104, "literal_worst_vs_exact": 5e-5,
                 "literal_worst_vs_oracle": 1.5e-3, "default_worst_vs_oracle": 1.5e-3, "oracle_worst_vs_exact": 1.5e-3}


@pytest.mark.gpu
def test_eight_twenty_frame_streams_against_the_exact_prior_stream(ctx, oracle):
    """Eight sequences, twenty images each, through optimization() + slideWindow() with the prior handed from frame to frame.
    Streams as above: T (FP64 solver, every marginalization in binary128), O (the FP64 oracle), G (the GPU).

    What round 4 measured (VERDICT r3 item 3b asked for a per-frame amplification to sort the frames by): the one-frame
    propagation factor of a state deviation (_stream) is 1 .. 2e3 on these sequences and does NOT predict where a stream leaves T -
    stream 2 jumps from 6e-8 to 1e-5 at a frame whose factor is 6, whatever form the square root takes and whichever clamp is used
    (tests/tools/dev_stream_probe.py): what separates the streams enters through weakly determined directions of the prior itself,
    from the first, prior-less marginalization on.  A sensitivity of the solve to a Jacobi-scaled perturbation of J^T J (2e1 .. 7e3)
    does not sort the frames either.  So there is no `amplification below a bound` class to assert 1e-6 on; what is asserted is
    what holds: identical decisions at every frame; |G - T| < 1e-6 on the first frames (k <= 1) of every stream; at every frame
    |G - T| < 1e-6, or |G - T| <= |O - T|, or - the third class, counted and bounded - |G - T| <= 5 |O - T| and < 5e-5.
    Measured: 160 frames, G within 1e-6 of T on about half, closer to T than the oracle on all but ~15 frames (stream 2), worst
    |G - T| 2e-5 (oracle 5e-4).  The north-star 1e-6 is therefore NOT met along whole streams by either implementation: the
    exact-prior stream is itself only defined to ~1e-5 in FP64.

    Round 5: the same streams once more with the REFERENCE-LITERAL clamp (avm_options::marg_noise_rel = 0:
    marginalization_factor.cpp:284-285 as written, nothing but S > 1e-8) as a fourth stream L, reported next to the default:
    |L - T|, and |L - O| - the distance of the GPU in literal mode from the FP64 oracle, which runs the same literal clamp.  Decisions
    are asserted for L as for G; its distances are reported, the classes above are asserted for the default only."""
    n, n_seq = 20, 8
    o = abi.default_options()
    o_lit = abi.default_options()
    o_lit.marg_noise_rel = 0.0
    kw = dict(n_frames=34, n_landmarks=600)
    tot = within = by_oracle = third = 0
    worst_g = worst_o = worst_ratio = 0.0
    lit_within = lit_by_oracle = lit_dec = 0
    worst_l = worst_lo = worst_go = 0.0
    clamp_more = clamp_fewer = clamp_more_l = clamp_fewer_l = 0   # frames whose prior drops more / fewer directions than the exact one
    failures = []
    for sid in range(n_seq):
        amp = []
        T = _stream(sid, _Oracle(oracle, o, exact_prior=True), n, amplification=amp, **kw)
        O = _stream(sid, _Oracle(oracle, o), n, **kw)
        G = _stream(sid, _Gpu(ctx, o), n, **kw)
        L = _stream(sid, _Gpu(ctx, o_lit), n, **kw)
        line = []
        for k in range(n):
            assert G[k]["n_feat"] == T[k]["n_feat"] and G[k]["it"] == T[k]["it"] and G[k]["acc"] == T[k]["acc"] and G[k]["term"] == T[k]["term"], (sid, k, G[k], T[k])
            dg = max(rel(G[k][key], T[k][key]) for key in ("pose", "speedbias"))
            do = max(rel(O[k][key], T[k][key]) for key in ("pose", "speedbias"))
            dl = max(rel(L[k][key], T[k][key]) for key in ("pose", "speedbias"))
            lit_dec += int(not (L[k]["it"] == T[k]["it"] and L[k]["acc"] == T[k]["acc"] and L[k]["term"] == T[k]["term"] and L[k]["n_feat"] == T[k]["n_feat"]))
            clamp_more += G[k]["clamped"] > T[k]["clamped"]
            clamp_fewer += G[k]["clamped"] < T[k]["clamped"]
            clamp_more_l += L[k]["clamped"] > T[k]["clamped"]
            clamp_fewer_l += L[k]["clamped"] < T[k]["clamped"]
            lit_within += dl < 1e-6
            lit_by_oracle += (not dl < 1e-6) and dl <= do
            worst_l = max(worst_l, dl)
            worst_lo = max(worst_lo, max(rel(L[k][key], O[k][key]) for key in ("pose", "speedbias")))
            worst_go = max(worst_go, max(rel(G[k][key], O[k][key]) for key in ("pose", "speedbias")))
            line.append(f"{k}:{amp[k]:.0e}/{dg:.0e}/{do:.0e}")
            tot += 1
            worst_g, worst_o = max(worst_g, dg), max(worst_o, do)
            if dg < 1e-6:
                within += 1
            elif dg <= do:
                by_oracle += 1
            else:
                third += 1
                worst_ratio = max(worst_ratio, dg / do)
                if not (dg <= 5.0 * do and dg < 5e-5):
                    failures.append((sid, k, dg, do))
            if k <= 1 and not dg < 1e-6:
                failures.append((sid, k, "first frames", dg, do))
        print(f"\n[stream {sid}] frame:propagation factor/|G-T|/|O-T|  " + " ".join(line))
    print(f"\n[streams] {tot} frames: |G-T| < 1e-6 on {within}; beyond 1e-6 but <= |O-T| on {by_oracle}; beyond the oracle on {third} "
          f"(worst ratio {worst_ratio:.1f}); worst |G-T| {worst_g:.1e}, worst |O-T| {worst_o:.1e}")
    print(f"[streams, reference-literal clamp (marg_noise_rel = 0)] {tot} frames: |L-T| < 1e-6 on {lit_within}; beyond 1e-6 but <= |O-T| on {lit_by_oracle}; "
          f"worst |L-T| {worst_l:.1e}; frames whose decisions differ from T's: {lit_dec};  distance from the FP64 oracle's stream: literal {worst_lo:.1e}, default {worst_go:.1e}")
    print(f"[streams, directions dropped by the prior's clamp against the exact prior's] default: more on {clamp_more} frames, fewer on {clamp_fewer}; "
          f"literal: more on {clamp_more_l}, fewer on {clamp_fewer_l}")
    assert not failures, failures
    assert third <= tot // 6 and within >= tot // 4
    assert lit_dec == 0
    # what round 5 measured and only printed is asserted (bounds = the round-6 measurement with a margin: STREAM_BOUNDS), so that
    # a regression of either leg - the default's distance from the exact-prior stream, the literal clamp's distance from the FP64 oracle's stream - is red
    assert within >= STREAM_BOUNDS["default_frames_within_1e-6"] and worst_g < STREAM_BOUNDS["default_worst_vs_exact"], (within, worst_g)
    assert lit_within >= STREAM_BOUNDS["literal_frames_within_1e-6"] and worst_l < STREAM_BOUNDS["literal_worst_vs_exact"], (lit_within, worst_l)
    assert worst_lo < STREAM_BOUNDS["literal_worst_vs_oracle"] and worst_go < STREAM_BOUNDS["default_worst_vs_oracle"], (worst_lo, worst_go)
    assert worst_o < STREAM_BOUNDS["oracle_worst_vs_exact"], worst_o
