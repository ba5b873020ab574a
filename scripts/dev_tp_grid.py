"""Dev: the throughput kernel's time for 4096 windows against the number of resident workgroups (AVM_TP_GRID)."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT]
PKG = "anticipated-vins-mono_amd"
mod = lambda n: importlib.import_module(PKG + "." + n)
abi, synth = mod("abi"), mod("synth")
ctx = mod("lib").Context(0)
opt = abi.default_options(); opt.marginalization_flag = abi.MARGIN_NONE
E = mod("estimator").Estimator(ctx=ctx, options=opt)
big = synth.tile_windows(synth.make_windows(64, tracks="dense"), 4096).to_device("cuda:0")
os.environ["AVM_SOLVE_TP"] = "1"
for grid in sys.argv[1:] or ["128", "256", "384", "512"]:
    os.environ["AVM_TP_GRID"] = grid
    for rep in range(2):
        E.optimization(big.copy())
    print("grid", grid, ctx.last_solve_form(), "window_solve ms %.3f" % ctx.kernel_ms("window_solve"), flush=True)
