"""Eigenvalue spectra around the clamp: truth (binary128) vs FP64 oracle vs GPU, for rank-deficient windows (no prior)."""
import importlib, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import oracle_py
from helpers import abi, buffers, synth
from marg_sensitivity import marginalize_at, truth_marginalize
gpu = "--no-gpu" not in sys.argv
E = importlib.import_module("anticipated-vins-mono_amd.estimator").Estimator(options=abi.default_options()) if gpu else None
o = abi.default_options()
np.set_printoptions(linewidth=250, precision=2)
for tracks, nf in (("sparse", 80), ("dense", 150)):
    B = 3
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150, with_prior=False)
    oracle_py.window_solve(o, w, buffers.PriorOutArrays.alloc(B), buffers.summary_alloc(B))
    po, at = marginalize_at(w, o)
    _, diag = truth_marginalize(at, o)
    pg = marginalize_at(w, o, estimator=E)[0] if gpu else None
    for i in range(B):
        n = diag[i]["n"]
        et = np.sort(diag[i]["ev_rr"])
        eo = np.sort((po.a["J"][i, :n, :n] ** 2).sum(1))
        print(f"{tracks} {nf} w{i} n={n}\n  truth  ", et[:34])
        print("  oracle ", eo[:34])
        if gpu:
            eg = np.sort((pg.a["J"][i, :n, :n] ** 2).sum(1))
            print("  gpu    ", eg[:34])
        print("  diag A' (truth) min/med/max", np.sort(np.diag(diag[i]["A"]))[[0, n // 2, -1]])
