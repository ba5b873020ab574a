"""Wall time of one optimization() call with host buffers (AVM_MEM_HOST: what a drop-in C++ host uses) against\ndevice-resident buffers, 1 and 8 windows."""
import importlib, sys, time, numpy as np
sys.path.insert(0, ".")
abi = importlib.import_module("anticipated-vins-mono_amd.abi"); synth = importlib.import_module("anticipated-vins-mono_amd.synth")
est = importlib.import_module("anticipated-vins-mono_amd.estimator")
E = est.Estimator(options=abi.default_options())
for nw in (1, 8):
    w = synth.make_windows(nw, tracks="sparse", n_feat=150, max_feat=150)
    for mode in ("host", "device"):
        ts = []
        for it in range(12):
            x = w.copy() if mode == "host" else w.to_device("cuda:0")
            import torch; torch.cuda.synchronize()
            t = time.perf_counter(); E.optimization(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
        print(nw, mode, "median wall ms", 1e3 * float(np.median(ts[2:])))
