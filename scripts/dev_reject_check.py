"""Parity incl. rejected trust-region steps: perturb the initial states so that some dogleg steps get rejected."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi"); buffers = importlib.import_module(pkg + ".buffers")
est_m = importlib.import_module(pkg + ".estimator")
import oracle_py
opt = abi.default_options(); opt.marginalization_flag = abi.MARGIN_NONE
E = est_m.Estimator(options=opt)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 8.0
rng = np.random.default_rng(7)
for tracks in ("sparse", "dense"):
    w = synth.make_windows(n, tracks=tracks)
    # larger perturbation of positions / inverse depths -> rejected steps
    w.a["pose"][:, 1:, :3] += rng.normal(0, 0.05 * scale, w.a["pose"][:, 1:, :3].shape)
    w.a["speedbias"][:, :, :3] += rng.normal(0, 0.05 * scale, w.a["speedbias"][:, :, :3].shape)
    w.a["inv_depth"] *= np.exp(np.clip(rng.normal(0, 0.02 * scale, w.a["inv_depth"].shape), -1.5, 1.5))
    g, o, g2 = w.copy(), w.copy(), w.copy()
    os.environ["AVM_NO_SPECULATE"] = "0"
    sg = buffers.summary_to_numpy(E.optimization(g))
    os.environ["AVM_NO_SPECULATE"] = "1"
    sg2 = buffers.summary_to_numpy(E.optimization(g2))
    fin = np.isfinite(g.a["pose"]).all(axis=(1, 2)) & np.isfinite(g2.a["pose"]).all(axis=(1, 2))
    print("  speculative vs classic on the GPU: accept masks equal", int((sg["accept_mask"] == sg2["accept_mask"]).sum()), "/", n,
          " pose diff", float(np.abs(g.a["pose"][fin] - g2.a["pose"][fin]).max()), " finite", int(fin.sum()))
    so = buffers.summary_alloc(n); oracle_py.window_solve(opt, o, None, so, n_threads=os.cpu_count())
    nrej = int(((sg["num_iterations"] - sg["num_successful"]) > 0).sum())
    same = int((sg["accept_mask"] == so["accept_mask"]).sum())
    def rel(a, b): return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))
    print(tracks, "windows with rejected steps:", nrej, "/", n, " identical accept masks:", same, "/", n,
          " pose rel (finite windows)", rel(g.a["pose"][fin], o.a["pose"][fin]), " term same", int((sg["termination"] == so["termination"]).sum()))
