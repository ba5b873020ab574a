"""preint kernel time vs samples per interval (dev tool)."""
import importlib, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); est_m = importlib.import_module(pkg + ".estimator"); abi = importlib.import_module(pkg + ".abi")
o = abi.default_options(); o.marginalization_flag = abi.MARGIN_NONE; o.max_num_iterations = 1
E = est_m.Estimator(options=o)
base = synth.make_windows(32, tracks="sparse", n_feat=20, max_feat=150)
for n in (20, 10, 2, 1):
    w = synth.tile_windows(base, 4096)
    w.a["imu_n"][:] = n
    wd = w.to_device()
    for _ in range(3):
        try:
            E.optimization(wd.copy())
        except Exception as e:
            pass
    print("samples per interval", n, "preint ms", E.ctx.kernel_ms("preint"))
