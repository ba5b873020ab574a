"""TEST INFRASTRUCTURE (uses oracle/).  Development: one frame family of tests/dev_sweep_fsel.py where a form of the selector and the FP64 oracle differ - who is right
(the binary128 arbiter decides), and by how little the two candidates differ."""
import importlib, os, sys
import numpy as np
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, _root); sys.path.insert(0, os.path.join(_root, "tests"))
from helpers import buffers, synth
from oracle import oracle_py
from marg_sensitivity import truth_fsel_select
fs_m = importlib.import_module("anticipated-vins-mono_amd.feature_selector")
FS = fs_m.FeatureSelector()
P = 512
pr = synth.make_fsel(P, first_id=70000, horizon=5, n_cand=200, n_used=4, max_features=60)
outs = {}
for form in ("1", "0"):
    os.environ["AVM_FSEL_SOLO"] = form
    outs[form] = FS.select_batch(pr)
    print("form", FS.ctx.last_fsel_form())
oo = buffers.FselOutArrays.alloc(P, 60)
oracle_py.fsel_select(pr, oo, n_threads=16)
for q in range(P):
    a, b, o = outs["1"].a["selected_ids"][q], outs["0"].a["selected_ids"][q], oo.a["selected_ids"][q]
    if not (np.array_equal(a, o) and np.array_equal(b, o)):
        k = int(np.argmax((a != o) | (b != o)))
        one = type(pr)(dict(pr.dims, n_problems=1), {kk: np.ascontiguousarray(v[q:q + 1]) for kk, v in pr.a.items()}, pr.scalars)
        tr = truth_fsel_select(one)
        print(f"frame {q}: first difference at pick {k}: solo {a[k]} teams {b[k]} oracle {o[k]} binary128 {tr.a['selected_ids'][0][k]}")
        print("   fValue there: solo %.17g teams %.17g oracle %.17g binary128 %.17g" % (outs["1"].a["fvalues"][q, k], outs["0"].a["fvalues"][q, k], oo.a["fvalues"][q, k], tr.a["fvalues"][0, k]))
        print("   whole selection equal to binary128's: solo", np.array_equal(a, tr.a["selected_ids"][0]), "teams", np.array_equal(b, tr.a["selected_ids"][0]), "oracle", np.array_equal(o, tr.a["selected_ids"][0]))
