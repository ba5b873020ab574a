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
    return marginalize_at(w, opt, estimator)[0]


def marginalize_at(w, opt, estimator=None):
    """marginalize_only() plus the windows as the call left them: optimization() marginalizes at the state it RETURNS (the
    gauge fix of double2vector / vector2double sits in between, and is the identity only up to the last bit), so that is the
    state an exact evaluation of the same marginalization has to start from."""
    import oracle_py

    o0 = abi.Options.from_buffer_copy(bytes(opt))
    o0.max_num_iterations = 0
    win = w.copy()
    pr = buffers.PriorOutArrays.alloc(w.n_windows)
    if estimator is None:
        oracle_py.window_solve(o0, win, pr, buffers.summary_alloc(w.n_windows))
        return pr, win
    old = estimator.options
    estimator.options = o0
    try:
        estimator.optimization(win, prior_out=pr)
    finally:
        estimator.options = old
    return pr, win


def install_prior(win, p):
    a = win.a
    a["prior_n"][:], a["prior_nblk"][:] = np.asarray(p.a["n"]).astype(np.int32), p.a["nblk"]
    a["prior_blk_kind"][:], a["prior_blk_frame"][:] = p.a["blk_kind"], p.a["blk_frame"]
    a["prior_J"][:], a["prior_r"][:], a["prior_x0"][:] = p.a["J"], p.a["r"], p.a["x0"]


# ---------------------------------------------------------------------------------------------------------------------
# The extended-precision arbiter (oracle/avm_truth.cpp: the oracle's marginalization with __float128 as its scalar type,
# following marginalization_factor.cpp:232-291 literally from the same FP64 inputs).  "How far is X from the value the
# reference's algorithm defines" replaces "how far is X from the oracle, in units of the oracle's own scatter".
_TRUTH = None


def truth_lib():
    global _TRUTH
    if _TRUTH is None:
        import ctypes as C
        import os
        import subprocess

        here = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")
        path = os.path.join(here, "libavm_truth.so")
        if not os.path.exists(path):
            subprocess.check_call(["make", "-C", here, "-s", "libavm_truth.so"])
        _TRUTH = C.CDLL(path)
    return _TRUTH


def truth_marginalize(w, opt):
    """The marginalization of every window of `w` at the state it is in, in binary128.  Returns (PriorOutArrays rounded to
    FP64, diag) with diag[i] = dict(H, g, cost: J^T J, J^T r, |r|^2 / 2 formed in binary128 and rounded once; A, b: the
    Schur complement; ev_mm / ev_rr: the eigenvalues the two clamps looked at)."""
    import ctypes as C

    L = truth_lib()
    B = w.n_windows
    pr = buffers.PriorOutArrays.alloc(B)
    mp, mb = pr.dims["max_prior"], pr.dims["max_pblk"]
    s = w.struct()
    diag = []
    for i in range(B):
        n, nb, nmm = C.c_int32(), C.c_int32(), C.c_int32()
        J, r, H, g = np.zeros(mp * mp), np.zeros(mp), np.zeros(mp * mp), np.zeros(mp)
        A, b, ev_mm, ev_rr = np.zeros(mp * mp), np.zeros(mp), np.zeros(512), np.zeros(mp)
        kind, frame, x0 = np.zeros(mb, np.int32), np.zeros(mb, np.int32), np.zeros((mb, 9))
        cost = C.c_double()
        rc = L.avmt_marginalize(C.byref(opt), C.byref(s), i, mb, C.byref(n), C.byref(nb), abi.iptr(kind), abi.iptr(frame), abi.dptr(x0),
                                abi.dptr(J), abi.dptr(r), abi.dptr(H), abi.dptr(g), C.byref(cost), abi.dptr(A), abi.dptr(b),
                                abi.dptr(ev_mm), 512, C.byref(nmm), abi.dptr(ev_rr))
        nn = n.value
        pr.a["n"][i] = nn
        if rc != 0:
            diag.append(None)
            continue
        pr.a["nblk"][i] = nb.value
        pr.a["blk_kind"][i], pr.a["blk_frame"][i], pr.a["x0"][i] = kind, frame, x0
        pr.a["J"][i, :nn, :nn] = J[: nn * nn].reshape(nn, nn)
        pr.a["r"][i, :nn] = r[:nn]
        diag.append(dict(n=nn, H=H[: nn * nn].reshape(nn, nn).copy(), g=g[:nn].copy(), cost=cost.value, A=A[: nn * nn].reshape(nn, nn).copy(),
                         b=b[:nn].copy(), ev_mm=ev_mm[: nmm.value].copy(), ev_rr=ev_rr[:nn].copy()))
    return pr, diag


def distance_to_truth(p, diag, i):
    """The four metrics of prior_metrics() for window i of prior `p`, against the binary128 H, g, cost of `diag[i]`."""
    n, Hp, gp, cp = _quad(p, i)
    t = diag[i]
    assert n == t["n"]
    Hq, gq, cq = t["H"], t["g"], t["cost"]
    d = 1.0 / np.sqrt(np.maximum(np.diag(Hq), 1e-300))
    return dict(H_rel=float(np.abs(Hp - Hq).max() / np.abs(Hq).max()), H_scaled=float(np.abs((Hp - Hq) * d[:, None] * d[None, :]).max()),
                g_scaled=float(np.abs((gp - gq) * d).max() / max(1e-300, np.abs(gq * d).max())), cost_rel=abs(cp - cq) / max(cq, 1e-300))


def truth_solve(w, opt):
    """Estimator::optimization()'s solve (no marginalization) of every window of `w` in binary128 (oracle/avm_truth.cpp:
    avmt_solve - the restated Ceres dogleg minimizer and the gauge-fix round trip, from the pre-integration on, on the same FP64
    inputs).  Returns (a copy of `w` holding the resulting states rounded once to FP64, the summaries)."""
    import ctypes as C

    L = truth_lib()
    out = w.copy()
    s = w.struct()
    B = w.n_windows
    summ = buffers.summary_alloc(B)
    for i in range(B):
        pose, sb, ex, lam = np.zeros(11 * 7), np.zeros(11 * 9), np.zeros(7), np.zeros(w.dims["max_feat"])
        td, relo = np.zeros(1), np.zeros(7)
        one = buffers.summary_alloc(1)
        rc = L.avmt_solve(C.byref(opt), C.byref(s), i, abi.dptr(pose), abi.dptr(sb), abi.dptr(ex), abi.dptr(td), abi.dptr(relo), abi.dptr(lam),
                          one.ctypes.data_as(C.c_void_p))
        assert rc == 0
        nf = int(w.a["n_feat"][i])
        out.a["pose"][i], out.a["speedbias"][i], out.a["ex_pose"][i] = pose.reshape(11, 7), sb.reshape(11, 9), ex
        out.a["inv_depth"][i, :nf] = lam[:nf]
        if opt.estimate_td and "td" in out.a:
            out.a["td"][i] = td[0]
        if "relo_pose" in out.a:
            out.a["relo_pose"][i] = relo
        summ[i] = one[0]
    return out, summ


def truth_fsel_select(fsel):
    """FeatureSelector::select of every frame of `fsel` in binary128 (oracle/avm_truth.cpp: avmt_fsel_select).  Returns
    FselOutArrays (fvalues rounded to FP64)."""
    import ctypes as C

    L = truth_lib()
    P, mf = fsel.n_problems, fsel.dims["max_features"]
    out = buffers.FselOutArrays.alloc(P, mf)
    s = fsel.struct()
    for p in range(P):
        ids, fv = np.full(mf, -1, np.int32), np.zeros(mf)
        n = L.avmt_fsel_select(C.byref(s), p, abi.iptr(ids), abi.dptr(fv))
        out.a["n_selected"][p], out.a["selected_ids"][p], out.a["fvalues"][p] = n, ids, fv
    return out
