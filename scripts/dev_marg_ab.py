"""A/B of the two forms of the marginalization kernel on the bench batch (run on the GPU box): kernel ms per 4096 windows, dense and ragged."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
PKG = "anticipated-vins-mono_amd"
abi = importlib.import_module(PKG + ".abi"); synth = importlib.import_module(PKG + ".synth"); est = importlib.import_module(PKG + ".estimator")
E = est.Estimator(options=abi.default_options())
for tracks in ("dense", "sparse"):
    base = synth.make_windows(256, tracks=tracks, n_feat=150, max_feat=150)
    w = synth.tile_windows(base, 4096).to_device("cuda:0")
    for form in ("0", "1", "0", "1"):
        os.environ["AVM_MARG_TP"] = form
        ms = []
        for it in range(4):
            g = w.copy()
            E.optimization(g)
            ms.append((E.ctx.kernel_ms("window_solve"), E.ctx.kernel_ms("marginalize"), E.ctx.kernel_ms("prior_eig")))
        print(tracks, "AVM_MARG_TP=" + form, E.ctx.last_marg_form(), "solve / marginalize / prior ms:", " | ".join("%.3f %.3f %.3f" % m for m in ms[1:]), flush=True)
