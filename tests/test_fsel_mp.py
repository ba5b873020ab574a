"""FeatureSelector::select against an INDEPENDENT statement in 50 digits (round 4).

tests/golden/gen_fsel_mp.py states the whole selection a second time from the reference's sources - the horizon's IMU information with
Eigen's slerp, every feature's information with the pinhole projection, the FOV test and the nearest cloud point, and a brute-force greedy
on log-determinants of the full 9 (H + 1) x 9 (H + 1) matrices - in 50-digit arithmetic.  The binary128 arbiter (avmt_fsel_select: the
oracle's restatement with a wider scalar - reduced position system, Hadamard bounds, std::map order), the FP64 oracle and the GPU must
select exactly those ids in that order; the arbiter's fValues equal the independent ones to the rounding of its FP64 output, the
oracle's and the GPU's to 1e-10 (measured 2e-12 ... 5e-12: FP64 log-determinants of matrices with condition numbers of 1e6)."""
import os

import mpmath as mp
import numpy as np
import pytest

from helpers import buffers
from marg_sensitivity import truth_fsel_select

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fsel_mp.npz"), allow_pickle=False)
mp.mp.dps = 50
NC = int(GOLD["n_cases"])


def _frame(c):
    dims = {k[len(f"c{c}_d_"):]: int(GOLD[k]) for k in GOLD.files if k.startswith(f"c{c}_d_")}
    arrays = {k[len(f"c{c}_a_"):]: GOLD[k].copy() for k in GOLD.files if k.startswith(f"c{c}_a_")}
    scalars = {}
    for k in GOLD.files:
        if k.startswith(f"c{c}_s_"):
            v = GOLD[k]
            scalars[k[len(f"c{c}_s_"):]] = v.copy() if v.ndim else v.item()
    return buffers.FselArrays(dims, arrays, scalars)


def _check(out, c, tol, who):
    ids = GOLD[f"c{c}_ids"]
    n = int(out.a["n_selected"][0])
    assert n == len(ids) and out.a["selected_ids"][0, :n].tolist() == ids.tolist(), who
    worst = max(abs(mp.mpf(float(out.a["fvalues"][0, k])) - (mp.mpf(float(GOLD[f"c{c}_f_hi"][k])) + mp.mpf(float(GOLD[f"c{c}_f_lo"][k])))) /
                abs(mp.mpf(float(GOLD[f"c{c}_f_hi"][k]))) for k in range(n))
    print(f"\n[fsel mp] case {c}: {who}: {n} ids identical to the independent 50-digit selection, fValues within {mp.nstr(worst, 3)}")
    assert worst < tol, who


@pytest.mark.parametrize("c", range(NC))
def test_arbiter_and_oracle_select_what_the_independent_50_digit_statement_selects(oracle, c):
    fr = _frame(c)
    _check(truth_fsel_select(fr), c, 4e-16, "binary128 arbiter")
    oo = buffers.FselOutArrays.alloc(1, fr.dims["max_features"])
    oracle.fsel_select(fr, oo)
    _check(oo, c, 1e-10, "FP64 oracle")


@pytest.mark.gpu
@pytest.mark.parametrize("c", range(NC))
def test_gpu_selects_what_the_independent_50_digit_statement_selects(selector, monkeypatch, c):
    fr = _frame(c)
    _check(selector.select_batch(fr), c, 1e-10, "GPU (teams)")
    monkeypatch.setenv("AVM_FSEL_SOLO", "1")
    out = selector.select_batch(fr)
    if 3 * fr.dims["horizon"] <= 30:
        assert selector.ctx.last_fsel_form() == "solo"
    _check(out, c, 1e-10, "GPU (solo)")
