"""How far apart are the GPU's and the oracle's MARGIN_OLD priors, and how far apart are two runs of the ORACLE whose
inputs differ by one ulp?  (the committed assertion is tests/test_gpu_parity.py::
test_margin_old_prior_gap_is_inside_the_oracles_own_one_ulp_spread.)"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
PKG = "anticipated-vins-mono_amd"
abi, synth, buffers = (importlib.import_module(PKG + "." + m) for m in ("abi", "synth", "buffers"))
est_m = importlib.import_module(PKG + ".estimator")
import oracle_py
from marg_sensitivity import prior_metrics, ulp_perturbed, marginalize_only, install_prior

tracks = sys.argv[1] if len(sys.argv) > 1 else "sparse"
nf = int(sys.argv[2]) if len(sys.argv) > 2 else 60
B = int(sys.argv[3]) if len(sys.argv) > 3 else 4
o = abi.default_options()
E = est_m.Estimator(options=o)
w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
wg, wo = w.copy(), w.copy()
E.optimization(wg)
pg = E.last_marginalization_info
po = buffers.PriorOutArrays.alloc(B)
oracle_py.window_solve(o, wo, po, buffers.summary_alloc(B))
print("state gap after the solve:", {k: float(np.abs(wg.a[k] - wo.a[k]).max() / np.abs(wo.a[k]).max()) for k in ("pose", "speedbias", "inv_depth")})
print("gap, normal flow           :", prior_metrics(pg, po))
# both at the bit-identical state
pg0 = marginalize_only(wo, o, estimator=E)
po0 = marginalize_only(wo, o)
print("gap, same input state      :", prior_metrics(pg0, po0))
spread = []
for seed in range(8):
    pk = marginalize_only(ulp_perturbed(wo, seed), o)
    spread.append(prior_metrics(pk, po0))
print("oracle vs oracle(+-1 ulp)  :", {k: max(s[k] for s in spread) for k in spread[0]})
print("  per seed H_rel           :", [f"{s['H_rel']:.1e}" for s in spread])
# chained solve
o2 = abi.default_options()
o2.marginalization_flag = abi.MARGIN_NONE
E2 = est_m.Estimator(ctx=E.ctx, options=o2)
cg, co = wg.copy(), wo.copy()
install_prior(cg, pg), install_prior(co, po)
E2.optimization(cg)
oracle_py.window_solve(o2, co, None, buffers.summary_alloc(B))
print("chained solve gap          :", {k: float(np.abs(cg.a[k] - co.a[k]).max() / np.abs(co.a[k]).max()) for k in ("pose", "speedbias")})
ch = []
for seed in range(4):
    pk = marginalize_only(ulp_perturbed(wo, seed), o)
    ck = wo.copy()
    install_prior(ck, pk)
    oracle_py.window_solve(o2, ck, None, buffers.summary_alloc(B))
    c0 = wo.copy()
    install_prior(c0, po0)
    oracle_py.window_solve(o2, c0, None, buffers.summary_alloc(B))
    ch.append({k: float(np.abs(ck.a[k] - c0.a[k]).max() / np.abs(c0.a[k]).max()) for k in ("pose", "speedbias")})
print("oracle chained 1-ulp spread:", ch)
