"""MARGIN_OLD prior parity.

The reference's marginalization (marginalization_factor.cpp:267-291) goes through two symmetric eigen-decompositions of
matrices whose entries span 1e12 .. 1e0; its result is only determined up to the conditioning of that computation.
These tests therefore measure the GPU-vs-oracle gap against the ORACLE'S OWN spread when its inputs move by one unit in
the last place (the oracle follows the reference's algorithm shape literally: one joint eigen pseudo-inverse of Amm), and
pin what a consumer of the prior sees: a chained solve, and a 10-frame solve -> roll -> solve stream."""
import importlib

import numpy as np
import pytest

from helpers import abi, buffers, rel, synth
from marg_sensitivity import install_prior, marginalize_only, prior_metrics, ulp_perturbed

pytestmark = pytest.mark.gpu
est_m = importlib.import_module("anticipated-vins-mono_amd.estimator")
N_SEEDS = 12


@pytest.mark.parametrize("tracks,nf", [("sparse", 60), ("dense", 150), ("sparse", 150)])
def test_margin_old_prior_gap_is_inside_the_oracles_own_one_ulp_spread(ctx, oracle, tracks, nf):
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 4
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
    wo = w.copy()
    oracle.window_solve(o, wo, buffers.PriorOutArrays.alloc(B), buffers.summary_alloc(B))
    # both marginalizations at the BIT-IDENTICAL post-solve state: what differs is the algorithm's rounding only
    pg, po = marginalize_only(wo, o, estimator=E), marginalize_only(wo, o)
    assert np.array_equal(pg.a["n"], po.a["n"]) and np.array_equal(pg.a["blk_kind"], po.a["blk_kind"]) and np.array_equal(pg.a["blk_frame"], po.a["blk_frame"])
    gap = prior_metrics(pg, po)
    spread = [prior_metrics(marginalize_only(ulp_perturbed(wo, s), o), po) for s in range(N_SEEDS)]
    worst = {k: max(s[k] for s in spread) for k in gap}
    print("\n[prior parity]", tracks, nf, "gap", gap, "oracle 1-ulp spread", worst)
    for k in gap:
        assert gap[k] <= worst[k], (k, gap[k], worst[k])
    # absolute ceilings on top (measured: H 6e-6 relative, 1.4e-4 in Jacobi-scaled entries, g 5e-7, cost 1.1e-5)
    assert gap["H_rel"] < 2e-5 and gap["H_scaled"] < 5e-4 and gap["g_scaled"] < 2e-6 and gap["cost_rel"] < 5e-5


@pytest.mark.parametrize("tracks,nf", [("sparse", 60), ("dense", 150)])
def test_what_a_solve_sees_of_the_new_prior(ctx, oracle, tracks, nf):
    """The same solver (the oracle) started from the same state with the GPU's prior and with the oracle's prior: the two
    solutions agree to the north-star tolerance, and the GPU's solver on its own prior follows."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 4
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
    wg, wo = w.copy(), w.copy()
    E.optimization(wg)
    pg, po = E.last_marginalization_info, buffers.PriorOutArrays.alloc(B)
    oracle.window_solve(o, wo, po, buffers.summary_alloc(B))
    o2 = abi.default_options()
    o2.marginalization_flag = abi.MARGIN_NONE
    E2 = est_m.Estimator(ctx=ctx, options=o2)

    def chained(start, prior, gpu):
        c = start.copy()
        install_prior(c, prior)
        s = buffers.summary_alloc(B)
        if gpu:
            s = buffers.summary_to_numpy(E2.optimization(c))
        else:
            oracle.window_solve(o2, c, None, s)
        return c, s

    oo, soo = chained(wo, po, False)
    og, sog = chained(wo, pg, False)   # prior only
    go, sgo = chained(wo, po, True)    # solver only
    gg, sgg = chained(wg, pg, True)    # the product end to end
    # the oracle's own spread of this chained solve when the inputs of ITS marginalization move by one ulp
    po0 = marginalize_only(wo, o)
    ref0, _ = chained(wo, po0, False)
    own = {k: 0.0 for k in ("pose", "speedbias")}
    for sd in range(4):
        ck, _ = chained(wo, marginalize_only(ulp_perturbed(wo, sd), o), False)
        for k in own:
            own[k] = max(own[k], rel(ck.a[k], ref0.a[k]))
    for k in ("pose", "speedbias"):
        print(f"\n[chained {tracks} {nf}] {k}: prior only {rel(og.a[k], oo.a[k]):.2e}  solver only {rel(go.a[k], oo.a[k]):.2e}  "
              f"end to end {rel(gg.a[k], oo.a[k]):.2e}  oracle 1-ulp spread {own[k]:.2e}")
        assert rel(og.a[k], oo.a[k]) < max(1e-6, own[k]), ("prior only", k, rel(og.a[k], oo.a[k]), own[k])
        assert rel(go.a[k], oo.a[k]) < 1e-8, ("solver only", k, rel(go.a[k], oo.a[k]))
        assert rel(gg.a[k], oo.a[k]) < max(1e-6, own[k]), ("end to end", k, rel(gg.a[k], oo.a[k]), own[k])
    for s in (sog, sgo, sgg):
        assert np.array_equal(s["accept_mask"], soo["accept_mask"]) and np.array_equal(s["num_iterations"], soo["num_iterations"])
        assert rel(s["radius_trace"], soo["radius_trace"]) < 1e-5


class _Oracle:
    def __init__(self, oracle, opt):
        self.o, self.opt = oracle, opt

    def solve(self, w):
        p, s = buffers.PriorOutArrays.alloc(w.n_windows), buffers.summary_alloc(w.n_windows)
        self.o.window_solve(self.opt, w, p, s)
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


def _stream(seq_id, backend, n_frames, perturb_seed=None):
    """solve -> marginalize -> roll -> next image (new observations, new tracks, IMU) -> triangulate -> dead-reckon, n times."""
    seq = synth.Sequence(seq_id)
    w, ids = seq.first_window()
    if perturb_seed is not None:
        w = ulp_perturbed(w, perturb_seed, keys=("pose", "speedbias", "inv_depth", "obs_xy"))
    out = []
    for k in range(n_frames):
        prior, s = backend.solve(w)
        out.append(dict(pose=w.a["pose"].copy(), speedbias=w.a["speedbias"].copy(), inv_depth=w.a["inv_depth"].copy(), n_feat=int(w.a["n_feat"][0]),
                        it=int(s["num_iterations"][0]), acc=int(s["accept_mask"][0]), term=int(s["termination"][0]), cost=float(s["final_cost"][0])))
        backend.roll(w)
        ids = seq.next_image(w, ids, k)
        install_prior(w, prior)
        backend.new_frame(w)
    return out


@pytest.mark.parametrize("seq_id", [0, 1])
def test_ten_frame_solve_roll_solve_stream_matches_the_oracle(ctx, oracle, seq_id):
    """estimator.cpp:996-1107 in a loop: ten images through optimization() + slideWindow() with the prior handed from frame to
    frame, ~150 ragged tracks per window that are born and lost along the way.  At every frame the GPU's states sit within
    1e-6 of the oracle's - or within (twice) the oracle's own frame-k spread when ITS first-frame inputs move by one ulp,
    where that is larger.  Measured: the reference's algorithm is only reproducible to 1e-4 in streaming mode (every
    marginalization clamps eigenvalues of a matrix whose noise floor, 1e-16 x 1e12, is far above the 1e-8 threshold), and the
    GPU sits 2-6x closer to the oracle than the oracle's perturbed twins do."""
    n = 10
    o = abi.default_options()
    g = _stream(seq_id, _Gpu(ctx, o), n)
    r = _stream(seq_id, _Oracle(oracle, o), n)
    spread = [_stream(seq_id, _Oracle(oracle, o), n, perturb_seed=s) for s in range(5)]
    worst = 0.0
    for k in range(n):
        assert g[k]["n_feat"] == r[k]["n_feat"] and g[k]["it"] == r[k]["it"] and g[k]["acc"] == r[k]["acc"] and g[k]["term"] == r[k]["term"], (k, g[k], r[k])
        for key in ("pose", "speedbias"):
            gap = rel(g[k][key], r[k][key])
            own = max(rel(p[k][key], r[k][key]) for p in spread)
            print(f"[stream {seq_id}] frame {k} {key}: gpu-oracle {gap:.2e}  oracle 1-ulp spread {own:.2e}  features {r[k]['n_feat']}")
            # (the maximum of five samples is itself a noisy estimate of the spread: 40 comparisons against it need the factor)
            assert gap <= max(1e-6, 2.0 * own), (k, key, gap, own)
            worst = max(worst, gap)
    print(f"[stream {seq_id}] worst gpu-oracle gap over {n} frames: {worst:.2e}")


def test_cholesky_square_root_is_the_same_prior_as_the_eigen_square_root(ctx, oracle, monkeypatch):
    """prior_eig_kernel hands out the transposed Cholesky factor of A' whenever it can certify that no eigenvalue is near the
    1e-8 clamp (the usual case once the window has a prior), and the reference's eigen form diag(sqrt S) V^T otherwise.  Both are
    square roots of the same A' with the matching residual: identical J^T J, J^T r0 and |r0|^2, hence identical next solves.
    Windows without any prior are rank deficient (gauge freedom): they must still take the eigen path."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    for tracks, nf, with_prior in (("sparse", 60, True), ("dense", 150, True), ("sparse", 80, False)):
        w = synth.make_windows(3, first_id=500, tracks=tracks, n_feat=nf, max_feat=150, with_prior=with_prior)
        wa, wb = w.copy(), w.copy()
        monkeypatch.setenv("AVM_PRIOR_LITERAL", "1")
        E.optimization(wa)
        pa = E.last_marginalization_info
        monkeypatch.setenv("AVM_PRIOR_LITERAL", "0")
        E.optimization(wb)
        pb = E.last_marginalization_info
        assert np.array_equal(wa.a["pose"], wb.a["pose"]) and np.array_equal(pa.a["n"], pb.a["n"])
        n_fast = 0
        for i in range(3):
            n = int(pa.a["n"][i])
            Ja, Jb = pa.a["J"][i, :n, :n], pb.a["J"][i, :n, :n]
            clamped = int((np.abs(Ja).max(1) == 0).sum())                 # eigenvalues the literal path zeroed (<= eps)
            if clamped == 0:
                assert np.count_nonzero(Jb) <= n * (n + 1) // 2, (i, "the certified Cholesky path was expected")   # a (permuted) triangle
                assert np.count_nonzero(Ja) > n * (n + 1) // 2
                n_fast += 1
            else:   # an eigenvalue under the clamp: the eigen path either way, bit for bit
                assert np.array_equal(Ja, Jb) and np.array_equal(pa.a["r"][i], pb.a["r"][i]), i
        assert n_fast >= (2 if tracks == "dense" else 0) and (with_prior or n_fast == 0)
        m = prior_metrics(pb, pa)
        print("\n[cholesky vs eigen prior]", tracks, nf, with_prior, m)
        assert m["H_rel"] < 1e-9 and m["g_scaled"] < 1e-7 and m["cost_rel"] < 1e-6, m
        # and the next solve cannot tell them apart
        o2 = abi.default_options()
        o2.marginalization_flag = abi.MARGIN_NONE
        E2 = est_m.Estimator(ctx=ctx, options=o2)
        ca, cb = wa.copy(), wb.copy()
        install_prior(ca, pa), install_prior(cb, pb)
        sa = buffers.summary_to_numpy(E2.optimization(ca)).copy()
        sb = buffers.summary_to_numpy(E2.optimization(cb))
        assert np.array_equal(sa["accept_mask"], sb["accept_mask"])
        for k in ("pose", "speedbias", "inv_depth"):
            assert rel(ca.a[k], cb.a[k]) < 1e-8, (k, rel(ca.a[k], cb.a[k]))
