import importlib, sys, os, ctypes as C
os.environ["AVM_PROFILE"] = "1"
sys.path.insert(0, "/root/repo")
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi"); est_m = importlib.import_module(pkg + ".estimator")
opt = abi.default_options(); opt.marginalization_flag = abi.MARGIN_NONE
E = est_m.Estimator(options=opt)
w = synth.tile_windows(synth.make_windows(32, tracks="dense"), 256)
E.optimization(w.copy()); E.optimization(w.copy())
prof = (C.c_longlong * 32)()
E.ctx._L.avm_debug_copy_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
E.ctx.check(E.ctx._L.avm_debug_copy_profile(E.ctx.h, prof), "prof")
n = prof[31]
for k, nm in enumerate(["loop top/fetch", "proj_eval", "W/PF stores", "staging", "mfma loop", "flush", "", "", "", "wave0 phase A before barrier", "imu_raw wave"]):
    print(f"{nm:32s} {prof[16+k]/n:10.0f}")
print("A total", prof[0]/n)
