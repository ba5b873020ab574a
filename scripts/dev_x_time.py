"""Dev: kernel time of the extended solve (-DAVM_X) on 256 / 1024 dense windows (ex_pose, td, relocalization frame)."""
import importlib, os, sys, statistics
sys.path[:0] = ["."]
PKG = "anticipated-vins-mono_amd"
mod = lambda n: importlib.import_module(PKG + "." + n)
abi, synth = mod("abi"), mod("synth")
import torch
ctx = mod("lib").Context(0)
opt = abi.default_options(); opt.marginalization_flag = abi.MARGIN_NONE; opt.estimate_extrinsic = 1; opt.estimate_td = 1
E = mod("estimator").Estimator(ctx=ctx, options=opt)
base = synth.make_windows(32, tracks="dense", td_true=0.004, relo=True)
for nw in (256, 1024):
    w = synth.tile_windows(base, nw).to_device("cuda:0")
    ks = []
    for rep in range(7):
        x = w.copy()
        E.optimization(x)
        ks.append(ctx.kernel_ms("window_solve"))
    print(nw, "windows: extended solve kernel ms median %.3f" % statistics.median(ks[2:]), flush=True)
