"""TEST INFRASTRUCTURE (uses oracle/).  Development sweep (GPU box) of the EXTENDED problem (ex_pose, td, a relocalization frame as variables: the -DAVM_X
build of the solve kernel) against the oracle: python tests/dev_sweep_x.py [windows per shape].  Results: profiles/r06z_sweep_extended_*.txt"""
import importlib, sys, time
import numpy as np
import os as _os
_root = _os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))
sys.path.insert(0, _root); sys.path.insert(0, _os.path.join(_root, 'tests'))
from helpers import abi, buffers, rel, synth
from oracle import oracle_py
est_m = importlib.import_module("anticipated-vins-mono_amd.estimator")
libm = importlib.import_module("anticipated-vins-mono_amd.lib")
ctx = libm.Context(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tot_w = tot_mis = 0
worst = 0.0
for (ex, td, relo, tracks, nf, fid) in ((1, 1, True, "sparse", 150, 1000), (1, 1, True, "dense", 150, 21000), (1, 0, True, "sparse", 90, 41000), (0, 1, False, "sparse", 120, 61000),
                                         (2, 1, True, "sparse", 60, 81000)):
    o = abi.default_options(); o.marginalization_flag = abi.MARGIN_NONE; o.estimate_extrinsic = ex; o.estimate_td = td
    E = est_m.Estimator(ctx=ctx, options=o)
    w = synth.make_windows_parallel(B, first_id=fid, procs=16, tracks=tracks, n_feat=nf, max_feat=150, td_true=0.008 if td else None, relo=relo)
    if ex:
        w.a["ex_pose"][:, :3] += 0.01
    wo, so = w.copy(), buffers.summary_alloc(B)
    t = time.time()
    oracle_py.window_solve(o, wo, None, so)
    wg = w.copy(); sg = buffers.summary_to_numpy(E.optimization(wg))
    mis = int((sg["accept_mask"] != so["accept_mask"]).sum()) + int((sg["num_iterations"] != so["num_iterations"]).sum()) + int((sg["termination"] != so["termination"]).sum())
    per = np.abs(wg.a["pose"] - wo.a["pose"]).reshape(B, -1).max(1) / np.abs(wo.a["pose"]).max()
    extra = {k: rel(wg.a[k], wo.a[k]) for k in (["ex_pose"] if ex else []) + (["td"] if td else []) + (["relo_pose"] if relo else [])}
    print(f"ex {ex} td {td} relo {relo} {tracks} {nf} x {B}: decision mismatches {mis}, pose rel worst {per.max():.3e}, speedbias {rel(wg.a['speedbias'], wo.a['speedbias']):.3e},"
          f" inv_depth {rel(wg.a['inv_depth'], wo.a['inv_depth']):.3e}, {extra}, oracle + solve {time.time() - t:.1f} s, form {ctx.last_solve_form()}", flush=True)
    tot_w += B; tot_mis += mis; worst = max(worst, float(per.max()))
print(f"TOTAL: {tot_w} windows of the extended problem, decision mismatches against the oracle {tot_mis}, worst pose difference (relative) {worst:.3e}")
