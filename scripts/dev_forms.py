"""Dev: both forms of the solve / marginalization kernels at small batches (AVM_SOLVE_TP forces the choice): kernel and wall times."""
import importlib, os, sys, time, statistics
sys.path[:0] = ["."]
PKG = "anticipated-vins-mono_amd"
mod = lambda n: importlib.import_module(PKG + "." + n)
abi, synth = mod("abi"), mod("synth")
import torch
ctx = mod("lib").Context(0)
for marg in (abi.MARGIN_NONE, abi.MARGIN_OLD):
    opt = abi.default_options(); opt.marginalization_flag = marg
    E = mod("estimator").Estimator(ctx=ctx, options=opt)
    for nw in (1, 64, 256):
        base = synth.make_windows(min(nw, 16), tracks="dense")
        w = synth.tile_windows(base, nw).to_device("cuda:0")
        for form in ("0", "1"):
            os.environ["AVM_SOLVE_TP"] = form
            ks, ms, wl = [], [], []
            for rep in range(8):
                x = w.copy()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                E.optimization(x)
                torch.cuda.synchronize()
                wl.append((time.perf_counter() - t0) * 1e3)
                ks.append(ctx.kernel_ms("window_solve")); ms.append(ctx.kernel_ms("marginalize") + ctx.kernel_ms("prior_eig"))
            print("marg", marg, nw, "windows, form", ctx.last_solve_form(), "solve kernel ms %.3f" % statistics.median(ks[2:]),
                  "marg+prior ms %.3f" % statistics.median(ms[2:]), "wall ms %.3f" % statistics.median(wl[2:]), flush=True)
