"""Selector only: python scripts/dev_fsel_time.py <frames per call> [reps] [horizon]  (for rocprofv3 --kernel-trace --stats)"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
PKG = "anticipated-vins-mono_amd"
synth, fs_m = importlib.import_module(PKG + ".synth"), importlib.import_module(PKG + ".feature_selector")
P = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
H = int(sys.argv[3]) if len(sys.argv) > 3 else 10
FS = fs_m.FeatureSelector()
fp = synth.make_fsel(min(P, 64), horizon=H)
if P > 64:  # (tiled: the generator is the slow part)
    import numpy as np
    fp = type(fp)(dict(fp.dims, n_problems=P), {k: np.ascontiguousarray(v[np.arange(P) % 64]) for k, v in fp.a.items()}, fp.scalars)
fp = fp.to_device("cuda:0")
FS.select_batch(fp)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(reps):
    FS.select_batch(fp)
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / reps
print(f"{P} frames per call (H = {H}, {FS.ctx.last_fsel_form()}): {dt * 1e3:.3f} ms per call, {dt / P * 1e3:.3f} ms per frame; kernels {FS.ctx.kernel_ms('fsel_select'):.3f} ms")
