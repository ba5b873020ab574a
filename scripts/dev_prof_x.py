"""Per-phase shader-clock profile of the EXTENDED window-solve kernel (-DAVM_X: ex_pose, td, relocalization frame; AVM_PROFILE=1)."""
import importlib, sys, os, ctypes as C
os.environ["AVM_PROFILE"] = "1"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
pkg = "anticipated-vins-mono_amd"
synth = importlib.import_module(pkg + ".synth"); abi = importlib.import_module(pkg + ".abi")
est_m = importlib.import_module(pkg + ".estimator")
NAMES = ["A: frames(MFMA)+imu raw", "B: feat sums+diag", "prior resid", "zero S rows", "  chol: diag block | tp: tile load", "  chol: panel solve | tp: factorization", "  chol: trailing MFMA | tp: back substitution", "D: imu sqrt+JtJ", "E: prior + cost", "load+Hp", "scale/gmax", "schur(MFMA)", "cholesky", "tri solve", "backsub", "cand eval"]
nw = int(sys.argv[1]) if len(sys.argv) > 1 else 256
tracks = sys.argv[2] if len(sys.argv) > 2 else "dense"
opt = abi.default_options(); opt.estimate_extrinsic = 1; opt.estimate_td = 1
if len(sys.argv) > 3 and sys.argv[3] == "nomarg": opt.marginalization_flag = abi.MARGIN_NONE
E = est_m.Estimator(options=opt)
base = synth.make_windows(min(nw, 32), tracks=tracks, td_true=0.004, relo=True)
w = synth.tile_windows(base, nw)
E.optimization(w.copy())
E.optimization(w.copy())
ms = E.ctx.kernel_ms("window_solve")
prof = (C.c_longlong * 64)()
E.ctx._L.avm_debug_copy_profile.argtypes = [C.c_void_p, C.POINTER(C.c_longlong)]
E.ctx.check(E.ctx._L.avm_debug_copy_profile(E.ctx.h, prof), "prof")
tot = sum(prof[:4]) + sum(prof[7:16]); n = prof[31]
print(f"windows {nw} tracks {tracks}: kernel {ms:.3f} ms, {nw/ms*1e3:.0f} solves/s; per-window cycles total {tot/n:.0f}")
for k, nm in enumerate(NAMES):
    print(f"  {nm:22s} {prof[k]/n:12.0f} cyc/window  {100*prof[k]/tot:5.1f}%")
print(f'  chol lookahead: wave0 (tile+diag) {prof[28]/n:.0f} cyc/window ; wave1 (tiles) {prof[27]/n:.0f} cyc/window')
if True:
    QN = ["iteration head: D, |g/D|", "dogleg vectors + model", "Cauchy: |J u|^2", "state_plus + step norm", "accept / reject", "gauge fix + outputs",
          "load: states, tables", "load: cov lists, fs", "load: LPT + zero slot", "load: Hp = J0^T J0", "WINDOW TOTAL (wall)", "  of scale/gmax: gmax part"]
    for k, nm in enumerate(QN):
        print(f"  {nm:26s} {prof[32+k]/n:12.0f} cyc/window  {100*prof[32+k]/tot:5.1f}%")
print("  phase A busy time per wavefront (cyc/window):", " ".join(f"{prof[48+k]/n:.0f}" for k in range(8)))
MN = ["load", "A: frames+imu0", "B: feat sums/PART", "D: imu0 JtJ", "E: prior", "F: feature schur", "G+extract", "eig16", "pinv+schur15", "eig n", "write out"]
if prof[30]:
    mt = sum(prof[16:27]); print(f"preint {E.ctx.kernel_ms('preint'):.3f} ms; marginalize: kernel {E.ctx.kernel_ms('marginalize'):.3f} ms + prior_eig {E.ctx.kernel_ms('prior_eig'):.3f} ms; per-window cycles {mt/prof[30]:.0f}")
    print('  jacobi sweeps per window', prof[29]/prof[30])
    for k, nm in enumerate(MN): print(f"  {nm:22s} {prof[16+k]/prof[30]:12.0f} cyc/window  {100*prof[16+k]/mt:5.1f}%")
if any(prof[56:62]):  # a -DAVM_PROF_CHOL=<wavefront> build: that wavefront's time inside the factorization (chol_regs), cycles per window
    print("  factorization, one wavefront: " + " | ".join(f"{nm} {prof[56+k]/n:.0f}" for k, nm in enumerate(["chain", "wait (b)", "solve", "wait (d)", "update + rest", "tile load"])))
if os.environ.get("AVM_PROF_FT_PRINT") and prof[60]:  # a -DAVM_PROF_FT=<wavefront> build: that wavefront's frame task, cycles per 64-factor chunk
    print("  frame task, one wavefront, per chunk (%d chunks per window): " % (prof[60] / n) + " | ".join(f"{nm} {prof[56+k]/prof[60]:.0f}" for k, nm in enumerate(["inputs' wait", "evaluation + stores", "E^T E", "staging + MFMAs"])))
