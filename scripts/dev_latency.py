"""Single-window latency of optimization() (solve + marginalization), the reference's real-time use (one window per image)."""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi"); est_m = importlib.import_module(pkg + ".estimator")
import torch
E = est_m.Estimator(options=abi.default_options())
for tracks in ("dense", "sparse"):
    for nw in (1, 8, 256):
        w = synth.tile_windows(synth.make_windows(min(nw, 8), tracks=tracks), nw).to_device("cuda:0")
        E.optimization(w.copy()); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            c = w.copy(); torch.cuda.synchronize()
            t0 = time.perf_counter(); E.optimization(c); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        km = {k: round(E.ctx.kernel_ms(k), 3) for k in ("preint", "window_solve", "marginalize", "prior_eig")}
        print(f"{tracks:6s} n_windows={nw:4d}: wall {min(ts)*1e3:.3f} ms (device-resident buffers), kernels {km}")
