"""The feature selection against the extended-precision arbiter (oracle/avm_truth.cpp: avmt_fsel_select).

selected ids are discrete: the FP64 oracle and the device agree on them bit for bit in every test, but both decide every round by
comparing FP64 log-determinants of matrices with condition numbers of 1e6.  The binary128 run of the same restatement says which ids
the reference's ALGORITHM selects when no comparison is decided by rounding; the FP64 oracle (CPU tier) and the GPU (GPU tier) have
to select exactly those, in the same order, with fValues within 1e-9 of the exact ones.
"""
import numpy as np
import pytest

from helpers import buffers, rel, synth
from marg_sensitivity import truth_fsel_select


def _check(out, tru, tol):
    assert np.array_equal(out.a["n_selected"], tru.a["n_selected"]) and (tru.a["n_selected"] > 0).all()
    assert np.array_equal(out.a["selected_ids"], tru.a["selected_ids"])
    for p in range(out.a["n_selected"].shape[0]):
        n = int(tru.a["n_selected"][p])
        assert rel(out.a["fvalues"][p, :n], tru.a["fvalues"][p, :n]) < tol


def test_fp64_oracle_selects_what_the_binary128_selection_selects(oracle):
    pr = synth.make_fsel(2, first_id=31, horizon=5, n_cand=100, n_used=3, max_features=20)
    tru = truth_fsel_select(pr)
    oo = buffers.FselOutArrays.alloc(2, 20)
    oracle.fsel_select(pr, oo)
    _check(oo, tru, 1e-11)


@pytest.mark.gpu
@pytest.mark.parametrize("H,nc,mf,P", [(10, 150, 25, 1), (5, 100, 20, 3), (13, 90, 12, 1)])
def test_gpu_selects_what_the_binary128_selection_selects(selector, H, nc, mf, P):
    """Single frames (the DPP evaluation) and a small batch, at the report's horizon and at the reference's compiled HORIZON 13."""
    pr = synth.make_fsel(P, first_id=31, horizon=H, n_cand=nc, n_used=3, max_features=mf)
    tru = truth_fsel_select(pr)
    _check(selector.select_batch(pr), tru, 1e-9)
