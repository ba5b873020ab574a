"""Single-frame select() latency (dev tool): python tests/tools/fsel_single.py [reps] [horizon] [n_cand]"""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); fs_m = importlib.import_module(pkg + ".feature_selector"); lib_m = importlib.import_module(pkg + ".lib")
ctx = lib_m.Context(0)
FS = fs_m.FeatureSelector(ctx=ctx)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
H = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nc = int(sys.argv[3]) if len(sys.argv) > 3 else 500
f1 = synth.make_fsel(1, first_id=0, horizon=H, n_cand=nc).to_device(torch.device("cuda:0"))
FS.select_batch(f1)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    out = FS.select_batch(f1)
torch.cuda.synchronize()
print("horizon", H, "candidates", nc, "single frame ms", (time.perf_counter() - t1) / reps * 1e3, "kernel ms", ctx.kernel_ms("fsel_select"), "n_selected", int(out.to_host().a["n_selected"][0]))
