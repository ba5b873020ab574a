"""The two forms of the prior's square root: prior_eig_kernel hands out the transposed Cholesky factor of A'
where it can certify that no eigenvalue is near the 1e-8 clamp, the reference's eigen form otherwise.

How close the MARGIN_OLD prior is to what the reference's algorithm defines - the tests that used to live here, graded against the
FP64 oracle's own one-ulp scatter in round 2 - is now decided by an extended-precision arbiter: tests/test_prior_truth.py."""
import importlib

import numpy as np
import pytest

from helpers import abi, buffers, rel, synth
from marg_sensitivity import install_prior, prior_metrics

pytestmark = pytest.mark.gpu
est_m = importlib.import_module("anticipated-vins-mono_amd.estimator")


def test_cholesky_square_root_is_the_same_prior_as_the_eigen_square_root(ctx, oracle, monkeypatch):
    """prior_eig_kernel hands out the transposed Cholesky factor of A' whenever it can certify that no eigenvalue is near the
    1e-8 clamp (the usual case once the window has a prior), and the reference's eigen form diag(sqrt S) V^T otherwise.  Both are
    square roots of the same A' with the matching residual: identical J^T J, J^T r0 and |r0|^2, hence identical next solves.
    Windows without any prior are rank deficient (gauge freedom): they must still take the eigen path."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    for tracks, nf, with_prior in (("sparse", 60, True), ("dense", 150, True), ("sparse", 80, False)):
        w = synth.make_windows(3, first_id=500, tracks=tracks, n_feat=nf, max_feat=150, with_prior=with_prior)
        wa, wb = w.copy(), w.copy()
        monkeypatch.setenv("AVM_PRIOR_FORCE_EIG", "1")   # the eigen-decomposition for every window, the clamp as the options say
        E.optimization(wa)
        pa = E.last_marginalization_info
        monkeypatch.setenv("AVM_PRIOR_FORCE_EIG", "0")
        E.optimization(wb)
        pb = E.last_marginalization_info
        assert np.array_equal(wa.a["pose"], wb.a["pose"]) and np.array_equal(pa.a["n"], pb.a["n"])
        n_fast = n_rank_r = 0
        for i in range(3):
            n = int(pa.a["n"][i])
            Ja, Jb = pa.a["J"][i, :n, :n], pb.a["J"][i, :n, :n]
            clamped = int((np.abs(Ja).max(1) == 0).sum())                 # eigenvalues the literal path zeroed (<= eps)
            if clamped == 0:
                assert np.count_nonzero(Jb) <= n * (n + 1) // 2, (i, "the certified Cholesky path was expected")   # a (permuted) triangle
                assert np.count_nonzero(Ja) > n * (n + 1) // 2
                n_fast += 1
            else:
                # an eigenvalue under the clamp (exact zeros of A': directions nothing constrains).  Round 3: the rank-r Cholesky factor
                # is handed out when the kept part is certified clear of the clamp (zero rows for the dropped directions in both
                # forms), else the eigen path, bit for bit; either way the same prior - asserted on the metrics below
                assert int((np.abs(Jb).max(1) == 0).sum()) >= 1, (i, "zero rows expected for the clamped directions")
                n_rank_r += int(np.count_nonzero(Jb) <= n * (n + 1) // 2 and not np.array_equal(Ja, Jb))
        assert n_fast >= (2 if tracks == "dense" else 0) and (with_prior or n_fast == 0)
        m = prior_metrics(pb, pa)
        print("\n[cholesky vs eigen prior]", tracks, nf, with_prior, m, "full-rank Cholesky form:", n_fast, "rank-r Cholesky form:", n_rank_r)
        # Round 6: the tolerances are the same with and without a prior.  Without one A' has 16 .. 30 EXACT zeros, and with the default
        # marg_noise_rel = 1e-18 (never drop a genuine direction) a few survive the clamp as rounding noise; a direction v that survives carries
        # v (v^T b') in J^T r0 with a v that is itself noise.  A Cholesky form is therefore only handed out when every kept direction clears the
        # MEASURED noise level (pc_cert_noise, prior_eig.hip); a window with a direction in the band between goes to the eigen form in either run -
        # which form finishes a window does not change the prior beyond rounding (round 5 tolerated g_scaled 5e-3 here).
        assert m["H_rel"] < 1e-9 and m["g_scaled"] < 1e-7 and m["cost_rel"] < 1e-6, m
        # and the next solve cannot tell them apart
        o2 = abi.default_options()
        o2.marginalization_flag = abi.MARGIN_NONE
        E2 = est_m.Estimator(ctx=ctx, options=o2)
        ca, cb = wa.copy(), wb.copy()
        install_prior(ca, pa), install_prior(cb, pb)
        sa = buffers.summary_to_numpy(E2.optimization(ca)).copy()
        sb = buffers.summary_to_numpy(E2.optimization(cb))
        assert np.array_equal(sa["accept_mask"], sb["accept_mask"])
        for k in ("pose", "speedbias", "inv_depth"):
            assert rel(ca.a[k], cb.a[k]) < 1e-8, (k, rel(ca.a[k], cb.a[k]))


def test_one_wavefront_factorization_with_deleted_pivots_is_the_pivoted_path_s_prior(ctx, monkeypatch):
    """Ragged tracks: two thirds of the windows leave an A' with two or four exact zeros.  prior_chol_kernel (one wavefront per
    window, natural order) deletes the pivots that are zero up to formation noise and checks what that drops; prior_eig_kernel (four
    wavefronts, diagonal pivoting, AVM_PRIOR_NO_FAST=1 forces it) stops at the rank.  256 windows: the same number of dropped
    directions in both, the same prior through everything a consumer sees of it (J^T J, J^T r0, |r0|^2), and the next solve cannot
    tell them apart."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 256
    w = synth.make_windows(B, first_id=7000, tracks="sparse", n_feat=150, max_feat=150)
    wa, wb = w.copy(), w.copy()
    monkeypatch.setenv("AVM_PRIOR_NO_FAST", "1")
    E.optimization(wa)
    pa = E.last_marginalization_info
    monkeypatch.delenv("AVM_PRIOR_NO_FAST")
    E.optimization(wb)
    pb = E.last_marginalization_info
    assert np.array_equal(wa.a["pose"], wb.a["pose"]) and np.array_equal(pa.a["n"], pb.a["n"])
    n_deleted = n_natural = 0
    for i in range(B):
        n = int(pa.a["n"][i])
        Ja, Jb = pa.a["J"][i, :n, :n], pb.a["J"][i, :n, :n]
        za, zb = int((np.abs(Ja).max(1) == 0).sum()), int((np.abs(Jb).max(1) == 0).sum())
        assert za == zb, (i, za, zb)
        natural = bool(np.array_equal(Jb, np.triu(Jb)))       # an upper triangle in natural order: the one-wavefront form
        n_natural += natural
        n_deleted += natural and zb > 0
    m = prior_metrics(pb, pa)
    print("\n[one-wavefront vs pivoted prior]", m, "natural-order form:", n_natural, "of", B, "with deleted pivots:", n_deleted)
    assert n_natural >= 0.95 * B and n_deleted >= 0.4 * B
    assert m["H_rel"] < 1e-9 and m["H_scaled"] < 1e-6 and m["g_scaled"] < 1e-7 and m["cost_rel"] < 1e-6, m
    o2 = abi.default_options()
    o2.marginalization_flag = abi.MARGIN_NONE
    E2 = est_m.Estimator(ctx=ctx, options=o2)
    ca, cb = wa.copy(), wb.copy()
    install_prior(ca, pa), install_prior(cb, pb)
    sa = buffers.summary_to_numpy(E2.optimization(ca)).copy()
    sb = buffers.summary_to_numpy(E2.optimization(cb))
    assert np.array_equal(sa["accept_mask"], sb["accept_mask"])
    for k in ("pose", "speedbias", "inv_depth"):
        assert rel(ca.a[k], cb.a[k]) < 1e-8, (k, rel(ca.a[k], cb.a[k]))


@pytest.mark.parametrize("how", ["option", "env"])
@pytest.mark.parametrize("tracks,nf", [("sparse", 60), ("dense", 150)])
def test_reference_literal_clamp_against_the_fp64_oracle(ctx, oracle, monkeypatch, how, tracks, nf):
    """avm_options::marg_noise_rel = 0 (or AVM_PRIOR_LITERAL=1) switches the noise test of the eigenvalue clamp off: what is left is
    the reference's S > eps (marginalization_factor.cpp:284-285), which is what the FP64 oracle does.  With a prior in the window
    nothing is near the clamp and the two priors agree to the FP64 oracle's own accuracy (it is 1e-7 .. 2e-5 from the binary128
    result in H, tests/test_prior_truth.py; the GPU 3e-9 .. 1e-7): fixed tolerances, no spread."""
    from marg_sensitivity import marginalize_at

    o = abi.default_options()
    if how == "option":
        o.marg_noise_rel = 0.0
    else:
        monkeypatch.setenv("AVM_PRIOR_LITERAL", "1")
    E = est_m.Estimator(ctx=ctx, options=o)
    B = 4
    w = synth.make_windows(B, first_id=300, tracks=tracks, n_feat=nf, max_feat=150)
    oracle.window_solve(o, w, buffers.PriorOutArrays.alloc(B), buffers.summary_alloc(B))
    po, _ = marginalize_at(w, o)
    pg, _ = marginalize_at(w, o, estimator=E)
    assert np.array_equal(pg.a["n"], po.a["n"]) and np.array_equal(pg.a["blk_kind"], po.a["blk_kind"]) and np.array_equal(pg.a["blk_frame"], po.a["blk_frame"])
    m = prior_metrics(pg, po)
    print("\n[literal clamp vs FP64 oracle]", how, tracks, nf, m)
    assert m["H_rel"] < 1e-4 and m["g_scaled"] < 1e-5 and m["cost_rel"] < 1e-4, m
    if how == "env":
        # the literal form is the eigen form: dense rows (the Cholesky form is a permuted triangle)
        n = int(pg.a["n"][0])
        assert np.count_nonzero(pg.a["J"][0, :n, :n]) > n * (n + 1) // 2


def test_the_prior_leaves_the_callers_upper_triangle_alone(ctx):
    """The scale of every diagonal entry of A' travels from marginalize_kernel to the prior kernels in the ctx's own array (it used to
    ride in the unused upper triangle of the caller's prior_out->J): whatever form the square root takes, J is either a full matrix
    (eigen form) or a triangle with exact zeros on the other side - never a triangle with stray magnitudes in it."""
    o = abi.default_options()
    E = est_m.Estimator(ctx=ctx, options=o)
    w = synth.make_windows(4, first_id=40, tracks="dense", n_feat=150, max_feat=150)
    E.optimization(w)
    p = E.last_marginalization_info
    for i in range(4):
        n = int(p.a["n"][i])
        J = p.a["J"][i, :n, :n]
        assert np.count_nonzero(J) <= n * (n + 1) // 2          # the certified Cholesky form of a dense window with a prior
        assert np.abs(p.a["J"][i, n:, :]).max() == 0.0 and np.abs(p.a["J"][i, :, n:]).max() == 0.0
