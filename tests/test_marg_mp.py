"""The marginalization arbiter against an INDEPENDENT statement in 50 digits (round 4).

Every prior of this repository - the FP64 oracle's and the GPU's - is graded against oracle/avm_truth.cpp, the oracle's restatement of
marginalization_factor.cpp compiled with binary128: it removes rounding as an explanation for a difference, not a misreading that both
builds share.  tests/golden/gen_marg_mp.py states MARGIN_OLD a second time, densely, from the numpy factor code the solve traces use
(written from the reference's sources, not from oracle/), in 50-digit arithmetic.  Here: the Schur complement A', b' the arbiter forms
(before the square root) and H = J'^T J', g = J'^T r' of the prior it hands over equal the independent ones to the rounding of the
arbiter's FP64 output (1.1e-16 of an entry; asserted at 4e-16 of the entry's Jacobi scale), on four windows: ragged and dense tracks,
with and without an old prior (the second clamp active on 22 eigenvalues there).  The FP64 comparison this replaces
(test_marginalization_equals_dense_schur_complement) can only assert 2e-3."""
import os

import mpmath as mp
import numpy as np
import pytest

from helpers import abi, buffers
from marg_sensitivity import truth_marginalize

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "marg_mp.npz"))
mp.mp.dps = 50


@pytest.mark.parametrize("c", range(int(GOLD["n_cases"])))
def test_binary128_marginalization_equals_the_independent_50_digit_statement(c):
    dims = {k[len(f"c{c}_dim_"):]: int(GOLD[k]) for k in GOLD.files if k.startswith(f"c{c}_dim_")}
    arrays = {k[len(f"c{c}_in_"):]: GOLD[k].copy() for k in GOLD.files if k.startswith(f"c{c}_in_")}
    w = buffers.WindowArrays(dims, arrays)
    o = abi.default_options()
    assert o.marginalization_flag == abi.MARGIN_OLD
    pr, diag = truth_marginalize(w, o)
    t = diag[0]
    n, nb = t["n"], int(pr.a["nblk"][0])
    # the arbiter's blocks carry the frame index AFTER the window has slid (pose[i] -> pose[i - 1], estimator.cpp:904-916)
    mine = [tuple(int(v) for v in row) for row in GOLD[f"c{c}_blocks"]]
    size = lambda kind: 9 if kind == abi.BLK_SPEEDBIAS else 6
    off_mine, o_ = {}, 0
    for blk in mine:
        off_mine[blk] = o_
        o_ += size(blk[0])
    assert o_ == n and nb == len(mine)
    perm = []
    for k in range(nb):
        kind, fr = int(pr.a["blk_kind"][0, k]), int(pr.a["blk_frame"][0, k])
        key = (kind, fr + 1) if kind in (abi.BLK_POSE, abi.BLK_SPEEDBIAS) else (kind, 0)
        assert key in off_mine, (key, mine)
        perm += list(range(off_mine[key], off_mine[key] + size(kind)))
    assert sorted(perm) == list(range(n))

    def dd(nm):
        hi, lo = GOLD[f"c{c}_{nm}_hi"], GOLD[f"c{c}_{nm}_lo"]
        return hi, lo

    worst = {}
    for M_, v_ in (("A", "b"), ("H", "g")):
        Mh, Ml = dd(M_)
        vh, vl = dd(v_)
        Mh, Ml, vh, vl = Mh[np.ix_(perm, perm)], Ml[np.ix_(perm, perm)], vh[perm], vl[perm]
        Ma, va = t[M_], t[v_]
        # Jacobi scale of an entry: sqrt(M_ii M_jj); a clamped direction can leave a zero on H's diagonal - use A's scale for both
        Ah = dd("A")[0][np.ix_(perm, perm)]
        d = np.sqrt(np.abs(np.diag(Ah)))
        dM = np.abs((Ma - Mh) - Ml) / (d[:, None] * d[None, :])          # (the arbiter's FP64 output minus the 50-digit value, hi + lo)
        dv = np.abs((va - vh) - vl) / (d * np.sqrt(max(float(t["cost"]), 1e-300)) + np.abs(vh))
        worst[M_], worst[v_] = float(dM.max()), float(dv.max())
    print(f"\n[marg mp] case {c}: n = {n}; arbiter vs independent 50-digit statement, scaled: A' {worst['A']:.1e}  b' {worst['b']:.1e}  H {worst['H']:.1e}  g {worst['g']:.1e}")
    assert max(worst.values()) < 4e-16
