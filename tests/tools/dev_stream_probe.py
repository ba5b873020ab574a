"""Dev probe: where does a GPU stream leave the exact-prior stream?  usage: python tests/tools/dev_stream_probe.py <seq> <frames>"""
import os, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path[:0] = [os.path.dirname(HERE), os.path.dirname(os.path.dirname(HERE)), os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle")]
import numpy as np
from helpers import abi, rel
import oracle_py as oracle
import test_prior_truth as P
lib_m = P.est_m.__class__  # noqa
import importlib
ctx = importlib.import_module("anticipated-vins-mono_amd.lib").Context(0)
sid, n = int(sys.argv[1]), int(sys.argv[2])
kw = dict(n_frames=34, n_landmarks=600)
o = abi.default_options()
T = P._stream(sid, P._Oracle(oracle, o, exact_prior=True), n, **kw)
O = P._stream(sid, P._Oracle(oracle, o), n, **kw)
def run(tag, env=None, noise=None):
    for k, v in (env or {}).items(): os.environ[k] = v
    oo = abi.default_options()
    if noise is not None: oo.marg_noise_rel = noise
    G = P._stream(sid, P._Gpu(ctx, oo), n, **kw)
    for k in (env or {}): del os.environ[k]
    print(tag.ljust(28), " ".join(f"{max(rel(G[k][q], T[k][q]) for q in ('pose','speedbias')):.0e}" for k in range(n)), flush=True)
print("oracle".ljust(28), " ".join(f"{max(rel(O[k][q], T[k][q]) for q in ('pose','speedbias')):.0e}" for k in range(n)))
run("default")
run("no one-wavefront kernel", {"AVM_PRIOR_NO_FAST": "1"})
run("force eig", {"AVM_PRIOR_FORCE_EIG": "1"})
run("noise_rel 0", noise=0.0)
run("literal", {"AVM_PRIOR_LITERAL": "1"})
run("noise_rel 1e-15", noise=1e-15)
run("latency solve", {"AVM_SOLVE_TP": "0"})
