import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
PKG = "anticipated-vins-mono_amd"
abi, synth, buffers = (importlib.import_module(PKG + "." + m) for m in ("abi", "synth", "buffers"))
est_m = importlib.import_module(PKG + ".estimator")
import oracle_py
from marg_sensitivity import prior_metrics, ulp_perturbed, marginalize_only, install_prior
np.set_printoptions(linewidth=200, precision=4)
B = 4
o = abi.default_options()
E = est_m.Estimator(options=o)
w = synth.make_windows(B, first_id=300, tracks="dense", n_feat=150, max_feat=150)
wg, wo = w.copy(), w.copy()
E.optimization(wg)
pg = E.last_marginalization_info
po = buffers.PriorOutArrays.alloc(B)
oracle_py.window_solve(o, wo, po, buffers.summary_alloc(B))
o2 = abi.default_options(); o2.marginalization_flag = abi.MARGIN_NONE
E2 = est_m.Estimator(ctx=E.ctx, options=o2)
def chain(est, start, prior):
    c = start.copy(); install_prior(c, prior)
    if est is None:
        s = buffers.summary_alloc(B); oracle_py.window_solve(o2, c, None, s)
    else:
        s = buffers.summary_to_numpy(est.optimization(c))
    return c, s
# four combinations: solver x prior, all from the oracle's post-solve state
for name, est, pr in (("gpu solver, gpu prior", E2, pg), ("gpu solver, oracle prior", E2, po), ("oracle solver, gpu prior", None, pg), ("oracle solver, oracle prior", None, po)):
    c, s = chain(est, wo, pr)
    print(name, "iters", s["num_iterations"], "acc", s["accept_mask"], "term", s["termination"])
    print("   cost", s["initial_cost"], s["final_cost"])
    if name.startswith("gpu solver, gpu"): ref_gg = c
    if name.startswith("gpu solver, oracle"): ref_go = c
    if name.startswith("oracle solver, gpu"): ref_og = c
    if name.startswith("oracle solver, oracle"): ref_oo = c
rel = lambda a, b: [float(np.abs(a.a["pose"][i] - b.a["pose"][i]).max() / np.abs(b.a["pose"][i]).max()) for i in range(B)]
print("gpu/gpu vs ora/ora", rel(ref_gg, ref_oo))
print("gpu/ora vs ora/ora (solver only)", rel(ref_go, ref_oo))
print("ora/gpu vs ora/ora (prior only)", rel(ref_og, ref_oo))
for i in range(B):
    n = int(po.a["n"][i])
    Jg, Jo = pg.a["J"][i, :n, :n], po.a["J"][i, :n, :n]
    eg, eo = np.linalg.eigvalsh(Jg.T @ Jg), np.linalg.eigvalsh(Jo.T @ Jo)
    print("window", i, "smallest eig of H: gpu", eg[:4], "oracle", eo[:4], " rows with |J row| > 0: gpu", int((np.abs(Jg).max(1) > 0).sum()), "oracle", int((np.abs(Jo).max(1) > 0).sum()))
    print("   |r| gpu", np.linalg.norm(pg.a["r"][i, :n]), "oracle", np.linalg.norm(po.a["r"][i, :n]))
print("---- traces of window 1 (same oracle prior, same start)")
c, s = chain(E2, wo, po); print("gpu spec   cost", s["cost_trace"][1][:8]); print("           radius", s["radius_trace"][1][:8])
os.environ["AVM_NO_SPECULATE"] = "1"
c2, s2 = chain(E2, wo, po); print("gpu classic cost", s2["cost_trace"][1][:8]); print("           radius", s2["radius_trace"][1][:8])
c3, s3 = chain(None, wo, po); print("oracle     cost", s3["cost_trace"][1][:8]); print("           radius", s3["radius_trace"][1][:8])
print("spec vs classic pose", rel(c, c2))
