import importlib, os, sys, time, statistics
sys.path[:0] = ["."]
PKG = "anticipated-vins-mono_amd"
mod = lambda n: importlib.import_module(PKG + "." + n)
abi, synth = mod("abi"), mod("synth")
import torch
ctx = mod("lib").Context(0)
opt = abi.default_options(); opt.marginalization_flag = abi.MARGIN_NONE
E = mod("estimator").Estimator(ctx=ctx, options=opt)
for nw in (1, 64, 256):
    base = synth.make_windows(min(nw, 16), tracks="dense")
    w = synth.tile_windows(base, nw).to_device("cuda:0")
    for form in ("0", "1"):
        os.environ["AVM_SOLVE_TP"] = form
        ks = []
        for rep in range(6):
            x = w.copy()
            E.optimization(x)
            ks.append(ctx.kernel_ms("window_solve"))
        print(nw, "windows, form", ctx.last_solve_form(), "kernel ms median %.3f" % statistics.median(ks[1:]), flush=True)
