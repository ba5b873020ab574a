"""GPU tier (-m gpu): the THROUGHPUT form of the marginalization kernel (marginalize_tp_kernel in window_solve_tp.o, round 5: two
256-thread workgroups per CU, the joint system over the 91 variables a marginalization can touch instead of 172).

It follows the solve: a batch larger than the CU count takes both throughput forms.  The marginalization tests of the other files use
small batches (latency forms); here their bodies are collected once more with AVM_SOLVE_TP=1, which forces the throughput forms wherever
they are possible: the oracle comparisons, the binary128 arbiter (tests/test_prior_truth.py), the two square-root forms
(tests/test_prior_parity.py) and the streams all have to hold unchanged.  Plus what is specific to it: the two forms produce the same
A', b' to rounding, the choice rule, and the fall-back for a prior that keeps a speed-bias block of a later frame.
"""
import importlib

import numpy as np
import pytest

from helpers import abi, buffers, rel, synth

from test_gpu_parity import (  # noqa: F401
    test_marginalization_keeps_old_prior_when_second_new_has_nothing_to_drop,
    test_marginalization_rank_deficient_amm_takes_the_eigen_path,
    test_window_roll_matches_oracle_and_chains_solves,
)
from test_prior_parity import (  # noqa: F401
    test_cholesky_square_root_is_the_same_prior_as_the_eigen_square_root,
    test_one_wavefront_factorization_with_deleted_pivots_is_the_pivoted_path_s_prior,
    test_reference_literal_clamp_against_the_fp64_oracle,
    test_the_prior_leaves_the_callers_upper_triangle_alone,
)
from test_prior_truth import (  # noqa: F401
    test_gpu_prior_is_closer_to_the_truth_than_the_fp64_oracle,
    test_rank_deficient_joint_amm_gpu_vs_truth,
    test_rank_deficient_prior_against_the_truth,
    test_ten_frame_stream_against_the_exact_prior_stream,
    test_what_a_solve_sees_of_the_prior_against_the_truth,
)

pytestmark = pytest.mark.gpu

est_m = importlib.import_module("anticipated-vins-mono_amd.estimator")


@pytest.fixture(autouse=True)
def throughput_forms(monkeypatch, ctx):
    monkeypatch.setenv("AVM_SOLVE_TP", "1")
    monkeypatch.delenv("AVM_MARG_TP", raising=False)
    yield


def _quad(p, i):
    n = int(p.a["n"][i])
    J, r = p.a["J"][i, :n, :n], p.a["r"][i, :n]
    return n, J.T @ J, J.T @ r, 0.5 * float(r @ r)


@pytest.mark.parametrize("flag", ["OLD", "SECOND_NEW"])
@pytest.mark.parametrize("tracks,nf,prior", [("dense", 150, True), ("sparse", 60, True), ("sparse", 150, False), ("dense", 12, True)])
def test_the_two_forms_of_the_marginalization_agree_to_rounding(ctx, monkeypatch, flag, tracks, nf, prior):
    """Same window, same solve (the throughput form both times), the marginalization once per form: the same kept set, and the
    same prior as far as a consumer can see it (J^T J, J^T r, |r|^2) - different accumulation orders of the same sums."""
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_OLD if flag == "OLD" else abi.MARGIN_SECOND_NEW
    E = est_m.Estimator(ctx=ctx, options=o)
    w = synth.make_windows(6, tracks=tracks, n_feat=nf, max_feat=150, with_prior=prior)
    out = {}
    for form in ("0", "1"):
        monkeypatch.setenv("AVM_MARG_TP", form)
        g = w.copy()
        E.optimization(g)
        assert ctx.last_solve_form() == "throughput"
        assert ctx.last_marg_form() == ("throughput" if form == "1" else "latency")
        out[form] = (g, E.last_marginalization_info)
    (g0, p0), (g1, p1) = out["0"], out["1"]
    assert np.array_equal(g0.a["pose"], g1.a["pose"])  # the solve does not depend on what follows it
    assert np.array_equal(p0.a["n"], p1.a["n"]) and np.array_equal(p0.a["nblk"], p1.a["nblk"])
    assert np.array_equal(p0.a["blk_kind"], p1.a["blk_kind"]) and np.array_equal(p0.a["blk_frame"], p1.a["blk_frame"])
    assert np.array_equal(p0.a["x0"], p1.a["x0"])
    for i in range(w.n_windows):
        n, H0, b0, c0 = _quad(p0, i)
        if n <= 0:
            continue
        _, H1, b1, c1 = _quad(p1, i)
        d = 1.0 / np.sqrt(np.maximum(np.diag(H0), 1e-300))
        # MARGIN_OLD goes through the pseudo-inverse of an ill-conditioned Amm: two roundings of it differ like this (the oracle, the
        # reference's algorithm in FP64, is 1e-5 / 2e-3 from either: test_marginalization_parity)
        tol = (1e-6, 1e-4) if flag == "OLD" else (1e-11, 1e-9)
        assert rel(H1, H0) < tol[0], (i, rel(H1, H0))
        assert rel(H1 * d[:, None] * d[None, :], H0 * d[:, None] * d[None, :]) < tol[1]
        assert rel(b1 * d, b0 * d) < tol[1]
        assert abs(c1 - c0) <= 1e-4 * max(c0, 1e-300)


def test_which_form_a_marginalization_takes(ctx, monkeypatch):
    """It follows the solve: the latency forms up to the CU count, the throughput forms beyond; AVM_MARG_TP=0 keeps the latency form."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    monkeypatch.delenv("AVM_SOLVE_TP", raising=False)
    base = synth.make_windows(8, tracks="sparse", n_feat=40, max_feat=150)
    E.optimization(base.copy())
    assert (ctx.last_solve_form(), ctx.last_marg_form()) == ("latency", "latency")
    big = synth.tile_windows(base, 600)
    E.optimization(big)
    assert (ctx.last_solve_form(), ctx.last_marg_form()) == ("throughput", "throughput")
    J = E.last_marginalization_info.a["J"].reshape(75, 8, 96, 96)
    assert (J == J[0]).all()  # 75 copies of each window: bit-identical whichever workgroup slot marginalized it
    monkeypatch.setenv("AVM_MARG_TP", "0")
    E.optimization(synth.tile_windows(base, 600))
    assert (ctx.last_solve_form(), ctx.last_marg_form()) == ("throughput", "latency")


@pytest.mark.parametrize("where", ["host", "device"])
def test_a_prior_with_a_later_speed_bias_block_takes_the_latency_marginalization(ctx, oracle, where):
    """The compact joint system holds speed-biases 0 and 1 only (all the reference ever keeps).  A prior on ONE speed-bias block of a
    later frame does not fit it - nor, since round 6, the throughput solve, whose elimination order puts frame 0's block last (chol_regs):
    that window batch is solved and marginalized by the latency kernels - and the result is the oracle's."""
    o = abi.default_options()
    # (MARGIN_SECOND_NEW: under MARGIN_OLD a second kept speed-bias block would not fit the 76 rows a prior may have)
    o.marginalization_flag = abi.MARGIN_SECOND_NEW
    E = est_m.Estimator(ctx=ctx, options=o)
    w = synth.make_windows(3, tracks="sparse", n_feat=50, max_feat=150)
    for b in range(w.n_windows):
        k, f = w.a["prior_blk_kind"][b], w.a["prior_blk_frame"][b]
        sb = [q for q in range(int(w.a["prior_nblk"][b])) if k[q] == abi.BLK_SPEEDBIAS]
        assert len(sb) == 1
        f[sb[0]] = 3
        w.a["prior_x0"][b, sb[0], :9] = w.a["speedbias"][b, 3]
    wo, po = w.copy(), buffers.PriorOutArrays.alloc(w.n_windows)
    oracle.window_solve(o, wo, po, buffers.summary_alloc(w.n_windows))
    g = w.to_device("cuda:0") if where == "device" else w.copy()
    E.optimization(g)
    assert (ctx.last_solve_form(), ctx.last_marg_form()) == ("latency", "latency")
    pg = E.last_marginalization_info
    pg = pg.to_host() if hasattr(pg, "to_host") and where == "device" else pg
    assert np.array_equal(pg.a["n"], po.a["n"]) and np.array_equal(pg.a["nblk"], po.a["nblk"])
    for i in range(w.n_windows):
        n, Hg, gg, cg = _quad(pg, i)
        _, Ho, go, co = _quad(po, i)
        d = 1.0 / np.sqrt(np.diag(Ho))
        assert rel(Hg, Ho) < 1e-9
        assert rel(Hg * d[:, None] * d[None, :], Ho * d[:, None] * d[None, :]) < 1e-9


def test_a_large_batch_of_the_extended_problem_takes_the_throughput_marginalization(ctx, monkeypatch):
    """The extended problem (ex_pose / td as variables, a relocalization frame) is always solved by its own kernel, one workgroup per CU; the
    marginalization that follows is the same problem whatever the solve estimated, so a batch beyond the CU count takes its throughput form (two
    workgroups per CU).  Same solve, and the same prior as far as a consumer can see it, as with AVM_MARG_TP=0."""
    monkeypatch.delenv("AVM_SOLVE_TP", raising=False)
    o = abi.default_options()
    o.estimate_extrinsic, o.estimate_td = 1, 1
    E = est_m.Estimator(ctx=ctx, options=o)
    base = synth.make_windows(6, first_id=5, tracks="sparse", n_feat=60, max_feat=150, td_true=0.004, relo=True)
    small = base.copy()
    E.optimization(small)
    assert (ctx.last_solve_form(), ctx.last_marg_form()) == ("latency", "latency")
    p_small = E.last_marginalization_info
    out = {}
    for form in ("0", "1"):
        monkeypatch.setenv("AVM_MARG_TP", form)
        g = synth.tile_windows(base, 300)
        E.optimization(g)
        assert (ctx.last_solve_form(), ctx.last_marg_form()) == ("latency", "throughput" if form == "1" else "latency")
        out[form] = (g, E.last_marginalization_info)
    (g0, p0), (g1, p1) = out["0"], out["1"]
    assert np.array_equal(g0.a["pose"], g1.a["pose"]) and np.array_equal(g0.a["pose"][:6], small.a["pose"])
    assert np.array_equal(p0.a["J"][:6], p_small.a["J"])  # (the latency form of a tiled batch: bit-identical to the six windows alone)
    assert np.array_equal(p0.a["n"], p1.a["n"]) and np.array_equal(p0.a["blk_kind"], p1.a["blk_kind"]) and np.array_equal(p0.a["x0"], p1.a["x0"])
    J1 = p1.a["J"].reshape(50, 6, 96, 96)
    assert (J1 == J1[0]).all()
    for i in range(6):
        n, H0, b0, c0 = _quad(p0, i)
        _, H1, b1, c1 = _quad(p1, i)
        d = 1.0 / np.sqrt(np.maximum(np.diag(H0), 1e-300))
        assert rel(H1, H0) < 1e-6 and rel(H1 * d[:, None] * d[None, :], H0 * d[:, None] * d[None, :]) < 1e-4
        assert rel(b1 * d, b0 * d) < 1e-4 and abs(c1 - c0) <= 1e-4 * max(c0, 1e-300)
