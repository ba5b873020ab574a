"""Single-frame select() latency (dev tool): python tests/tools/fsel_single.py [reps]"""
import importlib, sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); fs_m = importlib.import_module(pkg + ".feature_selector"); lib_m = importlib.import_module(pkg + ".lib")
ctx = lib_m.Context(0)
FS = fs_m.FeatureSelector(ctx=ctx)
f1 = synth.make_fsel(1, first_id=0).to_device(torch.device("cuda:0"))
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
FS.select_batch(f1)
torch.cuda.synchronize()
t1 = time.perf_counter()
for _ in range(reps):
    out = FS.select_batch(f1)
torch.cuda.synchronize()
print("single frame ms", (time.perf_counter() - t1) / reps * 1e3, "kernel ms", ctx.kernel_ms("fsel_select"), "n_selected", int(out.to_host().a["n_selected"][0]))
