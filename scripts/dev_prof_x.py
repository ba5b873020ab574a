"""Per-phase shader-clock profile of the EXTENDED solve kernel (window_solve_x_kernel, AVM_PROFILE=1) next to the base latency form on the same windows."""
import importlib, sys, os, ctypes as C
os.environ["AVM_PROFILE"] = "1"
os.environ["AVM_SOLVE_TP"] = "0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi")
est_m = importlib.import_module(pkg + ".estimator")
NAMES = ["A: frames(MFMA)+imu raw", "B: feat sums+diag", "prior resid", "zero S rows", "  chol: diag block", "  chol: panel solve", "  chol: trailing MFMA", "D: imu sqrt+JtJ", "E: prior + cost", "load+Hp", "scale/gmax", "schur(MFMA)", "cholesky", "tri solve", "backsub", "cand eval"]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
base = synth.make_windows(32, tracks="dense", td_true=0.004, relo=True)
w = synth.tile_windows(base, nw)
for name, ext in (("base problem (latency form)", False), ("extended problem (-DAVM_X)", True)):
    opt = abi.default_options(); opt.marginalization_flag = abi.MARGIN_NONE
    ww = w.copy()
    if ext:
        opt.estimate_extrinsic, opt.estimate_td = 1, 1
    else:
        for k in ("relo_n", "relo_frame", "relo_feat", "relo_xy", "relo_pose", "obs_vel_td", "td"):
            ww.a.pop(k, None)
    E = est_m.Estimator(options=opt)
    E.optimization(ww.copy()); E.optimization(ww.copy())
    ms = E.ctx.kernel_ms("window_solve")
    prof = (C.c_longlong * 64)()
    E.ctx._L.avm_debug_copy_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
    E.ctx.check(E.ctx._L.avm_debug_copy_profile(E.ctx.h, prof), "prof")
    tot = sum(prof[:4]) + sum(prof[7:16]); n = prof[31]
    print(f"{name}: {nw} windows, kernel {ms:.3f} ms; per-window cycles total {tot/n:.0f}, wall {prof[42]/n:.0f}")
    for k, nm in enumerate(NAMES):
        print(f"  {nm:26s} {prof[k]/n:12.0f} cyc/window  {100*prof[k]/tot:5.1f}%")
    print("  phase A busy per wavefront:", " ".join(f"{prof[48+k]/n:.0f}" for k in range(8)))
