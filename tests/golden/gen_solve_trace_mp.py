"""Generates tests/golden/solve_trace_mp.npz: the independent numpy minimizer of gen_solve_trace.py run in 50-digit arithmetic
(mpmath) on two of its windows - the pin that makes the binary128 arbiter (oracle/avm_truth.cpp) and the
independent reading of the reference agree BEYOND FP64.

How: nothing of gen_solve_trace.py / gen_golden.py is rewritten.  Their module-level `np` is replaced by a proxy that makes every
array they create an object array of mpmath.mpf and routes the handful of float-only calls (sqrt, log, clip, linalg.norm / cholesky
/ inv / solve) to mpmath; the inputs are the FP64 inputs of solve_trace.npz converted exactly; the three constants those files
COMPUTE in floating point (460 / 1.5, the squared noise densities, float() casts) are computed in mpmath instead.  So the same
formulas, the same operation order, the same data, 166-bit significands - against the same formulas' restatement in oracle/
(Schur-eliminated, binary128, 113 bits).  The final state (Ceres' solution BEFORE double2vector's gauge fix), the initial and final
cost are stored as double-double pairs (hi + lo: 32 digits); tests/test_solve_trace_mp.py compares them with avmt_solve_dd.

Run once in the build container (about ten minutes):  python tests/golden/gen_solve_trace_mp.py
"""
import os
import sys
import time

import mpmath as mp
import numpy as real_np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
mp.mp.dps = 50
MPF = mp.mpf


def obj(a):
    """Any array-like -> an object array of mpf (floats are converted exactly)."""
    a = real_np.asarray(a, dtype=object)
    out = real_np.empty(a.shape, dtype=object)
    for idx in real_np.ndindex(a.shape):
        v = a[idx]
        out[idx] = v if isinstance(v, mp.mpf) else MPF(float(v)) if not isinstance(v, (int, real_np.integer)) else MPF(int(v))
    return out


class _Linalg:
    LinAlgError = real_np.linalg.LinAlgError

    @staticmethod
    def norm(x, axis=None, keepdims=False):
        assert axis is None
        x = obj(x).ravel()
        return mp.sqrt(sum((v * v for v in x), MPF(0)))

    @staticmethod
    def cholesky(A):  # lower factor, numpy's contract: raises when not positive definite
        A = obj(A)
        n = A.shape[0]
        L = obj(real_np.zeros((n, n)))
        for j in range(n):
            s = A[j, j] - sum((L[j, k] * L[j, k] for k in range(j)), MPF(0))
            if not s > 0:
                raise real_np.linalg.LinAlgError("not positive definite")
            L[j, j] = mp.sqrt(s)
            for i in range(j + 1, n):
                L[i, j] = (A[i, j] - sum((L[i, k] * L[j, k] for k in range(j)), MPF(0))) / L[j, j]
        return L

    @staticmethod
    def solve(A, b):  # only ever called with a triangular A here: general elimination with partial pivoting all the same
        M = mp.matrix(obj(A).tolist())
        return obj(list(mp.lu_solve(M, mp.matrix(obj(b).tolist()))))

    @staticmethod
    def inv(A):
        return obj((mp.matrix(obj(A).tolist()) ** -1).tolist())


class NpProxy:
    """numpy for the two generator modules: object arrays of mpf in, object arrays of mpf out."""
    linalg = _Linalg()
    nan = real_np.nan

    def __getattr__(self, name):  # everything not listed below is shape bookkeeping that works on object arrays as it is
        return getattr(real_np, name)

    @staticmethod
    def array(x, *a, **k):
        return obj(x)

    @staticmethod
    def zeros(shape, *a, **k):
        return obj(real_np.zeros(shape))

    @staticmethod
    def eye(n, *a, **k):
        return obj(real_np.eye(n))

    @staticmethod
    def diag(v):
        v = obj(v)
        out = obj(real_np.zeros((len(v), len(v))))
        for i in range(len(v)):
            out[i, i] = v[i]
        return out

    @staticmethod
    def sqrt(x):
        if isinstance(x, real_np.ndarray):
            return obj([mp.sqrt(v) for v in obj(x).ravel()]).reshape(x.shape)
        return mp.sqrt(x)

    @staticmethod
    def log(x):
        return mp.log(x)

    @staticmethod
    def clip(x, lo, hi):
        return obj([min(max(v, MPF(float(lo))), MPF(float(hi))) for v in obj(x).ravel()]).reshape(x.shape)

    @staticmethod
    def abs(x):
        return obj([abs(v) for v in obj(x).ravel()]).reshape(x.shape)

    @staticmethod
    def isfinite(x):
        return real_np.array([mp.isfinite(v) for v in obj(x).ravel()])

    @staticmethod
    def sum(x):
        return sum(obj(x).ravel(), MPF(0))


def main():
    import gen_golden as GG
    import gen_solve_trace as GS

    proxy = NpProxy()
    for mod in (GG, GS):
        mod.np = proxy
        mod.float = lambda v: v      # (float(r @ r), float(np.sum(dts)): would round to FP64)
    GS.SQRT_INFO = MPF(460.0 / 1.5)               # FOCAL_LENGTH / 1.5 (estimator.cpp:17) is an FP64 quotient in the reference: a constant of the problem
    GS.NOISE = tuple(MPF(v) for v in GS.NOISE)    # the densities are FP64 inputs; their squares are formed in 50 digits
    GS.G = obj(real_np.array([0.0, 0.0, 9.81007]))
    gold = real_np.load(os.path.join(HERE, "solve_trace.npz"))
    opt = {k: float(gold["opt_" + k]) for k in GS.OPT}
    opt["max_num_iterations"], opt["max_num_consecutive_invalid_steps"] = int(opt["max_num_iterations"]), int(opt["max_num_consecutive_invalid_steps"])
    # usage: gen_solve_trace_mp.py            all cases of solve_trace.npz, one after the other (7 minutes each), then the merge
    #        gen_solve_trace_mp.py 0,3,5      these cases -> solve_trace_mp.part_0_3_5.npz   (run several of these side by side)
    #        gen_solve_trace_mp.py merge      the parts -> solve_trace_mp.npz
    arg = sys.argv[1] if len(sys.argv) > 1 else "all"
    if arg != "merge":
        cases = list(range(int(gold["n_cases"]))) if arg == "all" else [int(v) for v in arg.split(",")]
        out = {}
        for c in cases:
            t0 = time.time()
            a = {k[len(f"c{c}_in_"):]: gold[k][0] for k in gold.files if k.startswith(f"c{c}_in_")}
            am = {k: (obj(v) if v.dtype.kind == "f" else v) for k, v in a.items()}
            P = GS.Problem(am)
            x0 = dict(pose=am["pose"].copy(), sb=am["speedbias"].copy(), lam=am["inv_depth"][: P.nf].copy())
            x, tr = GS.trust_region_solve(P, x0, opt)
            acc = [bool(v) for v in tr["accepted"]]
            want = gold[f"c{c}_trace_accepted"].astype(bool).tolist()
            print(f"case {c}: {time.time() - t0:.0f} s, iterations {tr['num_iterations']} termination {tr['termination']} accepted {acc}", flush=True)
            same = acc == want and tr["num_iterations"] == int(gold[f"c{c}_trace_num_iterations"])
            if not same:
                print(f"   the FP64 run took other decisions: {want}", flush=True)
            xv = list(x["pose"].ravel()) + list(x["sb"].ravel()) + list(x["lam"].ravel())
            costs = [tr["initial_cost"], tr["final_cost"]] + [v for v in tr["cost"]]
            for nm, vals in (("x", xv), ("cost", costs)):
                hi = real_np.array([float(v) for v in vals])
                lo = real_np.array([float(v - MPF(h)) for v, h in zip(vals, hi)])
                out[f"c{c}_{nm}_hi"], out[f"c{c}_{nm}_lo"] = hi, lo
            out[f"c{c}_accepted"] = real_np.array(acc)
            out[f"c{c}_termination"] = real_np.int64(tr["termination"])
        if arg != "all":
            real_np.savez_compressed(os.path.join(HERE, "solve_trace_mp.part_" + "_".join(str(c) for c in cases) + ".npz"), **out)
            return
    else:
        import glob
        out = {}
        for f in sorted(glob.glob(os.path.join(HERE, "solve_trace_mp.part_*.npz"))):
            part = real_np.load(f)
            out.update({k: part[k] for k in part.files})
    cs = sorted({int(k[1:k.index("_")]) for k in out if k.startswith("c") and k.endswith("_x_hi")})
    # How far the FP64 run of the same numpy code (solve_trace.npz) lands from its own 50-digit run, in the measures tests/test_solve_trace.py
    # grades an implementation with: costs after the iterations, max |difference| / max |cost|; the solution, max |difference| / max |x|.
    # A trace whose FP64 run is within a third of the test's tolerances is one FP64 arithmetic can reproduce.
    for c in cs:
        dd = lambda nm: [MPF(float(h)) + MPF(float(l)) for h, l in zip(out[f"c{c}_{nm}_hi"], out[f"c{c}_{nm}_lo"])]
        cost, x = dd("cost")[2:], dd("x")
        fc = gold[f"c{c}_trace_cost"]
        fx = real_np.concatenate([gold[f"c{c}_sol_pose"].ravel(), gold[f"c{c}_sol_speedbias"].ravel(), gold[f"c{c}_sol_inv_depth"].ravel()])
        same = len(fc) == len(cost) and gold[f"c{c}_trace_accepted"].astype(bool).tolist() == out[f"c{c}_accepted"].astype(bool).tolist()
        dc = max(abs(MPF(float(a)) - b) for a, b in zip(fc, cost)) / max(abs(b) for b in cost) if same else MPF("inf")
        dx = max(abs(MPF(float(a)) - b) for a, b in zip(fx, x)) / max(abs(b) for b in x) if same else MPF("inf")
        out[f"c{c}_fp64_cost_rel"], out[f"c{c}_fp64_state_rel"] = real_np.float64(float(dc)), real_np.float64(float(dx))
        out.pop(f"c{c}_fp64_cost_distance", None)
        print(f"case {c}: the FP64 run of the same code is {float(dc):.1e} (costs) / {float(dx):.1e} (solution) from this one")
    out["cases"], out["dps"] = real_np.array(cs, real_np.int64), real_np.int64(mp.mp.dps)
    real_np.savez_compressed(os.path.join(HERE, "solve_trace_mp.npz"), **out)
    print("wrote solve_trace_mp.npz with cases", cs)


if __name__ == "__main__":
    main()
