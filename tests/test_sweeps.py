"""Bounded versions of the development sweeps (tests/dev_sweep.py, tests/dev_sweep_fsel.py) as `-m gpu` tests.

* 2 112 distinct windows of four track shapes through avm_window_solve_batch (the throughput form: batches of this size take it on
  their own) against the FP64 oracle: identical decisions (iterations, accept mask, termination) on every window, states within 1e-8
  (the north star asks for 1e-6; the 51 200-window development sweep measured 3.3e-11 at worst).
* 832 frames of five selector shapes, HORIZON 13 among them, through avm_fsel_select_batch: every frame's ids, in order, are EITHER the
  FP64 oracle's, OR - where two candidates' log-determinants are so close that FP64 rounding turns the pick - the binary128
  arbiter's, with the flipped pair less than 1e-10 apart (relative); such frames are counted and reported.  ("Ids bit-exact" holds as
  far as roundings agree: the GPU hoists the constant pivots - a 30 x 30 Cholesky where the reference factors 99 x 99 - so its
  FP64 log-determinants carry other rounding errors than the oracle's.)
"""
import importlib

import numpy as np
import pytest

from helpers import PKG, abi, buffers, rel, synth
from marg_sensitivity import truth_fsel_select

est_m = importlib.import_module(PKG + ".estimator")

SHAPES = (("sparse", 150, 1024, 20000), ("sparse", 70, 512, 30000), ("dense", 150, 288, 40000), ("sparse", 110, 288, 50000))  # (each above the CU count)


@pytest.mark.gpu
def test_solve_sweep_2112_windows_against_the_oracle(ctx, oracle):
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    E = est_m.Estimator(ctx=ctx, options=o)
    tot = mis = 0
    worst = 0.0
    for tracks, nf, B, fid in SHAPES:
        # (forked numpy-only workers: they never touch the HIP runtime this process has open)
        w = synth.make_windows_parallel(B, first_id=fid, procs=8, tracks=tracks, n_feat=nf, max_feat=150)
        wo, so = w.copy(), buffers.summary_alloc(B)
        oracle.window_solve(o, wo, None, so, n_threads=16)
        wg = w.copy()
        sg = buffers.summary_to_numpy(E.optimization(wg))
        assert ctx.last_solve_form() == "throughput"
        bad = (sg["num_iterations"] != so["num_iterations"]) | (sg["accept_mask"] != so["accept_mask"]) | (sg["termination"] != so["termination"])
        per = np.abs(wg.a["pose"] - wo.a["pose"]).reshape(B, -1).max(1) / np.abs(wo.a["pose"]).max()
        print(f"\n[sweep] {tracks} {nf}: {B} windows, decision mismatches {int(bad.sum())}, worst pose {per.max():.2e} "
              f"speed-bias {rel(wg.a['speedbias'], wo.a['speedbias']):.2e} inverse depth {rel(wg.a['inv_depth'], wo.a['inv_depth']):.2e}")
        tot += B
        mis += int(bad.sum())
        worst = max(worst, float(per.max()), rel(wg.a["speedbias"], wo.a["speedbias"]))
        assert rel(wg.a["inv_depth"], wo.a["inv_depth"]) < 1e-6
    print(f"[sweep] TOTAL {tot} windows: {mis} decision mismatches against the oracle, worst state difference {worst:.2e}")
    assert tot >= 2048 and mis == 0 and worst < 1e-8


FSEL_SHAPES = (("bench shape: 500 -> 150, H 10", dict(), 64),
               ("H 5, 200 -> 60, 4 tracked", dict(horizon=5, n_cand=200, n_used=4, max_features=60), 512),
               ("H 3, 60 -> 25, no cloud", dict(horizon=3, n_cand=60, n_used=0, n_cloud=0, max_features=25), 128),
               ("H 10, 120 -> 40, 10 tracked", dict(n_cand=120, n_used=10, max_features=40), 64),
               ("H 13, 200 -> 60, 5 tracked", dict(horizon=13, n_cand=200, n_used=5, max_features=65), 64))


@pytest.mark.gpu
def test_selector_sweep_832_frames_against_the_oracle_and_the_binary128_arbiter(selector, oracle, monkeypatch):
    monkeypatch.delenv("AVM_FSEL_SOLO", raising=False)
    tot = rounding = 0
    for name, kw, P in FSEL_SHAPES:
        pr = synth.make_fsel(P, first_id=70000, **kw)
        out = selector.select_batch(pr)
        form = selector.ctx.last_fsel_form()
        oo = buffers.FselOutArrays.alloc(P, pr.dims["max_features"])
        oracle.fsel_select(pr, oo, n_threads=16)
        assert np.array_equal(out.a["n_selected"], oo.a["n_selected"])
        diff = [q for q in range(P) if not np.array_equal(out.a["selected_ids"][q], oo.a["selected_ids"][q])]
        for q in diff:
            # who is right, and by how little the two candidates differ: the frame alone through the binary128 statement
            one = type(pr)(dict(pr.dims, n_problems=1), {k: np.ascontiguousarray(v[q:q + 1]) for k, v in pr.a.items()}, pr.scalars)
            tr = truth_fsel_select(one)
            a, b = out.a["selected_ids"][q], oo.a["selected_ids"][q]
            k = int(np.argmax(a != b))
            fa, fb = float(out.a["fvalues"][q, k]), float(oo.a["fvalues"][q, k])
            gap = abs(fa - fb) / max(abs(fa), 1e-300)
            print(f"\n[fsel sweep] {name}, frame {q}, pick {k}: gpu {a[k]} oracle {b[k]} binary128 {tr.a['selected_ids'][0][k]}; relative gap of the two fValues {gap:.1e}")
            assert np.array_equal(a, tr.a["selected_ids"][0]), "the GPU's selection is neither the FP64 oracle's nor the binary128 arbiter's"
            assert gap < 1e-10
        tot += P
        rounding += len(diff)
        print(f"\n[fsel sweep] {name}: {P} frames ({form} form), {P - len(diff)} identical to the FP64 oracle, {len(diff)} rounding-decided (equal to binary128)")
    print(f"[fsel sweep] TOTAL {tot} frames, {rounding} rounding-decided")
    assert tot >= 512 and rounding <= tot // 100
