"""Dev: solve 64 dense + 64 ragged windows (throughput and latency forms) and the extended problem with the library in place; save the states to argv[1].npz
(two libraries' files are then compared bit for bit: scripts/dev_bitwise.py a.npz ; ... ; python -c compare)."""
import importlib, os, sys
sys.path[:0] = ["."]
import numpy as np
PKG = "anticipated-vins-mono_amd"
mod = lambda n: importlib.import_module(PKG + "." + n)
abi, synth = mod("abi"), mod("synth")
ctx = mod("lib").Context(0)
out = {}
for form in ("0", "1"):
    os.environ["AVM_SOLVE_TP"] = form
    for tracks in ("dense", "sparse"):
        E = mod("estimator").Estimator(ctx=ctx, options=abi.default_options())
        w = synth.make_windows(64, first_id=900, tracks=tracks)
        E.optimization(w)
        out[f"pose_{form}_{tracks}"] = w.a["pose"].copy(); out[f"sb_{form}_{tracks}"] = w.a["speedbias"].copy()
        out[f"J_{form}_{tracks}"] = E.last_marginalization_info.a["J"].copy()
os.environ.pop("AVM_SOLVE_TP")
o = abi.default_options(); o.estimate_extrinsic = 1; o.estimate_td = 1
E = mod("estimator").Estimator(ctx=ctx, options=o)
w = synth.make_windows(32, first_id=77, tracks="sparse", td_true=0.004, relo=True)
E.optimization(w)
out["pose_x"] = w.a["pose"].copy(); out["td_x"] = w.a["td"].copy(); out["ex_x"] = w.a["ex_pose"].copy()
np.savez(sys.argv[1], **out)
print("saved", sys.argv[1], len(out), "arrays")
