import importlib, sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi")
buf = importlib.import_module(pkg + ".buffers"); fs_m = importlib.import_module(pkg + ".feature_selector")
import oracle_py
def rel(a, b): return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
FS = fs_m.FeatureSelector()
for (H, nc, nu, mf, P) in ((10, 60, 5, 20, 3), (13, 40, 0, 15, 2), (3, 30, 0, 10, 2), (10, 500, 0, 150, 1)):
    pr = synth.make_fsel(P, horizon=H, n_cand=nc, n_used=nu, max_features=mf)
    om, dl, va = FS.information(pr); oom, odl, ova = oracle_py.fsel_information(pr)
    print(f"H={H} nc={nc}: omega {rel(om,oom):.2e} delta {rel(dl,odl):.2e} valid_eq {bool((va==ova).all())} nvalid {int(va.sum())}")
    t = time.time(); out = FS.select_batch(pr); tg = time.time() - t
    oo = buf.FselOutArrays.alloc(P, mf); t = time.time(); nld = oracle_py.fsel_select(pr, oo); to = time.time() - t
    same = bool((out.a["n_selected"] == oo.a["n_selected"]).all() and (out.a["selected_ids"] == oo.a["selected_ids"]).all())
    n0 = int(oo.a["n_selected"][0])
    print(f"   ids identical {same}; n_sel {out.a['n_selected']} fval rel {rel(out.a['fvalues'][0,:n0], oo.a['fvalues'][0,:n0]):.2e}"
          f" gpu kernel {FS.ctx.kernel_ms('fsel_select'):.2f} ms (wall {tg*1e3:.1f}) oracle {to*1e3:.0f} ms logdets {nld}")
    if not same:
        print(out.a["selected_ids"][0, :12], oo.a["selected_ids"][0, :12])
