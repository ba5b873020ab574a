"""Helpers for the MARGIN_OLD prior parity tests (test infrastructure; uses the CPU oracle).

The reference computes the new prior through two symmetric eigen-decompositions of matrices whose entries span
1e12 (IMU bias random-walk weights) .. 1e0 (weakly observed depths) - marginalization_factor.cpp:267-291.  The result
is only determined up to the conditioning of that computation, so "does the GPU agree with the oracle" is measured
against "does the oracle agree with itself when its inputs move by one unit in the last place"."""
import importlib

import numpy as np

PKG = "anticipated-vins-mono_amd"
abi = importlib.import_module(PKG + ".abi")
buffers = importlib.import_module(PKG + ".buffers")


def _quad(p, i):
    n = int(p.a["n"][i])
    J, r = np.asarray(p.a["J"][i, :n, :n]), np.asarray(p.a["r"][i, :n])
    return n, J.T @ J, J.T @ r, 0.5 * float(r @ r)


def prior_metrics(p, q):
    """Worst case over the windows of: relative H = J^T J, H and g = J^T r in Jacobi-scaled units, relative cost 1/2 |r|^2
    (q is the reference side).  The prior only ever enters a solve through these."""
    out = dict(H_rel=0.0, H_scaled=0.0, g_scaled=0.0, cost_rel=0.0)
    for i in range(len(q.a["n"])):
        n, Hp, gp, cp = _quad(p, i)
        m, Hq, gq, cq = _quad(q, i)
        assert n == m
        d = 1.0 / np.sqrt(np.maximum(np.diag(Hq), 1e-300))
        out["H_rel"] = max(out["H_rel"], float(np.abs(Hp - Hq).max() / np.abs(Hq).max()))
        out["H_scaled"] = max(out["H_scaled"], float(np.abs((Hp - Hq) * d[:, None] * d[None, :]).max()))
        out["g_scaled"] = max(out["g_scaled"], float(np.abs((gp - gq) * d).max() / max(1e-300, np.abs(gq * d).max())))
        out["cost_rel"] = max(out["cost_rel"], abs(cp - cq) / max(cq, 1e-300))
    return out


def ulp_perturbed(w, seed, keys=("pose", "speedbias", "inv_depth", "obs_xy", "prior_J", "prior_r")):
    """A copy of the windows whose marginalization inputs moved by -1, 0 or +1 unit in the last place, at random."""
    rng = np.random.default_rng(1000 + seed)
    out = w.copy()
    for k in keys:
        a = out.a[k]
        s = rng.integers(-1, 2, a.shape).astype(np.float64)
        out.a[k] = a * (1.0 + s * 2.0 ** -52)
    return out


def marginalize_only(w, opt, estimator=None):
    """The marginalization of optimization() at the state the windows are in (max_num_iterations = 0: the solve returns
    its starting point).  Through the oracle, or through the GPU when an Estimator class instance is given."""
    import oracle_py

    o0 = abi.Options.from_buffer_copy(bytes(opt))
    o0.max_num_iterations = 0
    win = w.copy()
    pr = buffers.PriorOutArrays.alloc(w.n_windows)
    if estimator is None:
        oracle_py.window_solve(o0, win, pr, buffers.summary_alloc(w.n_windows))
        return pr
    old = estimator.options
    estimator.options = o0
    try:
        estimator.optimization(win, prior_out=pr)
    finally:
        estimator.options = old
    return pr


def install_prior(win, p):
    a = win.a
    a["prior_n"][:], a["prior_nblk"][:] = np.asarray(p.a["n"]).astype(np.int32), p.a["nblk"]
    a["prior_blk_kind"][:], a["prior_blk_frame"][:] = p.a["blk_kind"], p.a["blk_frame"]
    a["prior_J"][:], a["prior_r"][:], a["prior_x0"][:] = p.a["J"], p.a["r"], p.a["x0"]
