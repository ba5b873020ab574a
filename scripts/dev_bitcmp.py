"""Development: solve one seeded batch (throughput forms) and save what came out, to compare two builds of the library bit for bit.
   python scripts/dev_bitcmp.py out.npz [n_windows]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
from helpers import abi, synth  # noqa: E402

est_m = importlib.import_module("anticipated-vins-mono_amd.estimator")
lib_m = importlib.import_module("anticipated-vins-mono_amd.lib")
n = int(sys.argv[2]) if len(sys.argv) > 2 else 512
ctx = lib_m.Context(0)
E = est_m.Estimator(ctx=ctx, options=abi.default_options())
base = synth.make_windows(8, tracks="dense", n_feat=150, max_feat=150)
w = synth.tile_windows(base, n // 8)
E.optimization(w)
p = E.last_marginalization_info
np.savez(sys.argv[1], pose=w.a["pose"], sb=w.a["speedbias"], lam=w.a["inv_depth"], J=p.a["J"], r=p.a["r"], form=np.array([ctx.last_solve_form()]))
print(ctx.last_solve_form(), ctx.last_marg_form())
