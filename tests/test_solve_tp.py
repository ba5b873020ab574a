"""GPU tier (-m gpu): the THROUGHPUT form of the solve kernel (window_solve_tp.o: two 256-thread workgroups per CU, the speed-bias
rows of the system in structural form, the factorization on register tiles).

A batch takes it on its own once it is larger than the CU count; the parity tests of the other files use small batches and
therefore run the latency form.  Here the same test bodies are collected once more with AVM_SOLVE_TP=1, which forces the throughput
form wherever it is possible (the base problem): the oracle comparisons, the numpy traces, the binary128 solve,
the speculation test, bit-reproducibility and shard invariance all have to hold for it unchanged.  Plus what is specific to it:
the two forms agree to rounding, the choice rule, and the fall-back for a prior the structural form cannot hold.
"""
import numpy as np
import pytest

from helpers import abi, buffers, rel, synth

# the same bodies, re-collected in this module: the autouse fixture below switches the form for every test of this file
from test_gpu_parity import (  # noqa: F401
    test_device_resident_buffers_match_host_path,
    test_gauge_fix_near_pitch_90_takes_the_full_rotation_branch,
    test_non_finite_inputs_terminate_like_the_oracle_and_do_not_leak_into_other_windows,
    test_observation_table_with_holes_solves_like_the_compact_one,
    test_small_trust_region_dogleg_branches_parity,
    test_solve_is_bit_reproducible_and_shard_invariant,
    test_speculative_evaluation_is_exact_including_rejected_steps,
    test_window_solve_degenerate_inputs,
    test_window_solve_parity,
    test_window_solve_parity_over_many_windows,
    test_window_solve_parity_over_random_track_structures,
    test_window_solve_without_prior_and_mixed_batch,
    test_chained_solves_through_the_new_prior,
    test_marginalization_parity,
)
from test_abi_round3 import test_time_cap_on_the_device  # noqa: F401  (expired cap: no step attempt; generous cap: bit-identical to no cap)
from test_solve_trace import test_hip_path_reproduces_the_independent_numpy_trace  # noqa: F401
from test_solve_truth import test_gpu_solve_is_as_close_to_the_binary128_solve_as_the_fp64_oracle  # noqa: F401

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def throughput_form(monkeypatch, ctx):
    monkeypatch.setenv("AVM_SOLVE_TP", "1")
    yield
    # every test of this file that solved a base-problem batch must have gone through the throughput kernel
    # (the marginalization / chained tests solve more than once; the last solve counts)


def _solve(E, w, form, monkeypatch):
    monkeypatch.setenv("AVM_SOLVE_TP", form)
    g = w.copy()
    s = buffers.summary_to_numpy(E.optimization(g))
    return g, s, E.ctx.last_solve_form()


@pytest.mark.parametrize("tracks,nf,prior", [("dense", 150, True), ("sparse", 60, True), ("sparse", 150, False), ("dense", 12, True)])
def test_the_two_forms_of_the_solve_agree_to_rounding(estimator, monkeypatch, tracks, nf, prior):
    w = synth.make_windows(12, tracks=tracks, n_feat=nf, max_feat=150, with_prior=prior)
    g0, s0, f0 = _solve(estimator, w, "0", monkeypatch)
    g1, s1, f1 = _solve(estimator, w, "1", monkeypatch)
    assert (f0, f1) == ("latency", "throughput")
    for k in ("num_iterations", "accept_mask", "termination"):
        assert np.array_equal(s0[k], s1[k]), k
    assert rel(s1["cost_trace"], s0["cost_trace"]) < 1e-7  # (a cost follows a 1e-11 state difference with its gradient, 1e4 .. 1e6 early on)
    for k in ("pose", "speedbias", "inv_depth"):
        assert rel(g1.a[k], g0.a[k]) < 1e-8, (k, rel(g1.a[k], g0.a[k]))  # measured 1e-12 .. 8e-11; each is ~1e-11 from the oracle


def test_which_form_a_batch_takes(ctx, monkeypatch):
    """Batches up to the CU count: the latency form; larger ones: the throughput form; the extended problem: always the latency form,
    whatever AVM_SOLVE_TP says.  A wall-clock cap (options.max_solver_time_in_seconds, estimator.cpp:803-806) does not decide the form
    (round 5: both kernels check it against a clock that starts with the window's own solve)."""
    import importlib

    opt = abi.default_options()
    opt.marginalization_flag = abi.MARGIN_NONE
    E = importlib.import_module("anticipated-vins-mono_amd.estimator").Estimator(ctx=ctx, options=opt)
    monkeypatch.delenv("AVM_SOLVE_TP", raising=False)
    base = synth.make_windows(8, tracks="sparse", n_feat=40, max_feat=150)
    E.optimization(base.copy())
    assert ctx.last_solve_form() == "latency"
    big = synth.tile_windows(base, 600)
    E.optimization(big)
    assert ctx.last_solve_form() == "throughput"
    a = big.a["pose"].reshape(75, 8, 11, 7)
    assert (a == a[0]).all()  # 75 copies of each window: bit-identical whichever workgroup slot solved it
    monkeypatch.setenv("AVM_SOLVE_TP", "1")
    opt2 = abi.default_options()
    opt2.marginalization_flag = abi.MARGIN_NONE
    opt2.max_solver_time_s = 10.0
    E2 = importlib.import_module("anticipated-vins-mono_amd.estimator").Estimator(ctx=ctx, options=opt2)
    E2.optimization(base.copy())
    assert ctx.last_solve_form() == "throughput"
    opt3 = abi.default_options()
    opt3.marginalization_flag = abi.MARGIN_NONE
    opt3.estimate_extrinsic = 1
    E3 = importlib.import_module("anticipated-vins-mono_amd.estimator").Estimator(ctx=ctx, options=opt3)
    E3.optimization(base.copy())
    assert ctx.last_solve_form() == "latency"


@pytest.mark.parametrize("where", ["host", "device"])
def test_a_prior_with_two_speed_bias_blocks_takes_the_latency_form(estimator, oracle, monkeypatch, where):
    """The structural form of the speed-bias rows has room for ONE prior speed-bias block (the reference never keeps more:
    estimator.cpp:904-916).  A prior with two is still solved correctly - by the other kernel."""
    w = synth.make_windows(3, tracks="sparse", n_feat=50, max_feat=150)
    kind, frame, nblk = w.a["prior_blk_kind"], w.a["prior_blk_frame"], w.a["prior_nblk"]
    # the synthetic prior is [10 poses | speed-bias 0 | ex_pose], n = 75; the same 75 x 75 J read as eight poses and THREE
    # speed-bias blocks (6 * 8 + 9 * 3 = 75, frames 0, 1, 2) is a prior that couples speed-biases of different frames
    for b in range(w.n_windows):
        kinds = [abi.BLK_POSE] * 8 + [abi.BLK_SPEEDBIAS] * 3
        frames = list(range(8)) + [0, 1, 2]
        nblk[b] = len(kinds)
        kind[b, :len(kinds)] = kinds
        frame[b, :len(kinds)] = frames
        # x0 of the blocks: the current states (dx = 0 at the start)
        x0 = w.a["prior_x0"][b]
        x0[:] = 0.0
        for q, (kk, ff) in enumerate(zip(kinds, frames)):
            if kk == abi.BLK_POSE:
                x0[q, :7] = w.a["pose"][b, ff]
            else:
                x0[q, :9] = w.a["speedbias"][b, ff]
    assert int(w.a["prior_n"][0]) == 75
    wo, so = w.copy(), buffers.summary_alloc(w.n_windows)
    oracle.window_solve(estimator.options, wo, None, so)
    g = w.to_device("cuda:0") if where == "device" else w.copy()
    s = buffers.summary_to_numpy(estimator.optimization(g))
    assert estimator.ctx.last_solve_form() == "latency"
    g = g.to_host() if where == "device" else g
    assert np.array_equal(s["accept_mask"], so["accept_mask"])
    for k in ("pose", "speedbias", "inv_depth"):
        assert rel(g.a[k], wo.a[k]) < 1e-6, k


def test_a_prior_on_another_frames_speed_bias_block(estimator, oracle):
    """... and so does a prior whose one speed-bias block is not frame 0's (round 6: the throughput form eliminates the speed-bias blocks last frame
    first so that the block the prior couples to every pose comes last and fills nothing, chol_regs; the reference only ever builds frame 0's,
    estimator.cpp:904-916).  The call takes the latency form and the result is the oracle's."""
    w = synth.make_windows(3, tracks="sparse", n_feat=50, max_feat=150)
    for b in range(w.n_windows):
        k = w.a["prior_blk_kind"][b]
        sb = [q for q in range(int(w.a["prior_nblk"][b])) if k[q] == abi.BLK_SPEEDBIAS]
        assert len(sb) == 1
        w.a["prior_blk_frame"][b, sb[0]] = 3
        w.a["prior_x0"][b, sb[0], :9] = w.a["speedbias"][b, 3]
    wo, so = w.copy(), buffers.summary_alloc(w.n_windows)
    oracle.window_solve(estimator.options, wo, None, so)
    g = w.copy()
    s = buffers.summary_to_numpy(estimator.optimization(g))
    assert estimator.ctx.last_solve_form() == "latency"
    assert np.array_equal(s["accept_mask"], so["accept_mask"]) and np.array_equal(s["termination"], so["termination"])
    for k in ("pose", "speedbias", "inv_depth"):
        assert rel(g.a[k], wo.a[k]) < 1e-6, k


def test_failed_factorizations_retry_like_the_latency_form(estimator, monkeypatch):
    """A window whose first factorizations fail (non-positive pivots: mu is raised tenfold and the system rebuilt) takes the same
    path in both forms: an information-free window (one two-view feature, no prior) has a singular reduced system at mu = 1e-8."""
    w = synth.make_windows(4, tracks="sparse", n_feat=1, max_feat=150, with_prior=False)
    g0, s0, f0 = _solve(estimator, w, "0", monkeypatch)
    g1, s1, f1 = _solve(estimator, w, "1", monkeypatch)
    assert (f0, f1) == ("latency", "throughput")
    for k in ("num_iterations", "accept_mask", "termination"):
        assert np.array_equal(s0[k], s1[k]), k
    assert np.isfinite(g1.a["pose"]).all()
    assert rel(g1.a["pose"], g0.a["pose"]) < 1e-6


@pytest.mark.parametrize("extended", [False, True])
def test_a_batch_mixing_priors_that_fit_the_sparse_factorization_and_priors_that_do_not(ctx, oracle, monkeypatch, extended):
    """The latency and extended kernels choose the factorization PER WINDOW (I_CRFIT, window_solve.hip: chol_regs where the prior's only speed-bias
    block is frame 0's, the left-looking factorization in LDS otherwise).  One call with both kinds of window - window 1's speed-bias block moved to
    frame 3 - gives the oracle's result for every window, and the windows that fit are bit-identical to a batch without the odd one."""
    import importlib
    monkeypatch.setenv("AVM_SOLVE_TP", "0")
    opt = abi.default_options()
    opt.marginalization_flag = abi.MARGIN_NONE
    if extended:
        opt.estimate_extrinsic = 1
    E = importlib.import_module("anticipated-vins-mono_amd.estimator").Estimator(ctx=ctx, options=opt)
    w = synth.make_windows(3, first_id=71, tracks="sparse", n_feat=60, max_feat=150)
    plain = w.copy()
    k = w.a["prior_blk_kind"][1]
    sb = [q for q in range(int(w.a["prior_nblk"][1])) if k[q] == abi.BLK_SPEEDBIAS]
    assert len(sb) == 1
    w.a["prior_blk_frame"][1, sb[0]] = 3
    w.a["prior_x0"][1, sb[0], :9] = w.a["speedbias"][1, 3]
    wo, so = w.copy(), buffers.summary_alloc(3)
    oracle.window_solve(opt, wo, None, so)
    g = w.copy()
    s = buffers.summary_to_numpy(E.optimization(g)).copy()
    assert ctx.last_solve_form() == "latency"
    assert np.array_equal(s["accept_mask"], so["accept_mask"]) and np.array_equal(s["termination"], so["termination"])
    for key in ("pose", "speedbias", "inv_depth", "ex_pose"):
        assert rel(g.a[key], wo.a[key]) < 1e-6, key
    E.optimization(plain)
    for b in (0, 2):
        assert np.array_equal(plain.a["pose"][b], g.a["pose"][b]) and np.array_equal(plain.a["speedbias"][b], g.a["speedbias"][b])
