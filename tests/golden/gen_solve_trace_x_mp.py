"""TEST INFRASTRUCTURE.  The numpy minimizer of gen_solve_trace_x.py - the optional members of Estimator::optimization(): ex_pose as a
variable, para_Td with ProjectionTdFactor, the relocalization frame - run in 50-digit arithmetic (mpmath), exactly as
gen_solve_trace_mp.py runs the base problem: the module's numpy is swapped for the proxy over arrays of mpf, nothing is rewritten.
Output solve_trace_x_mp.npz: per case Ceres' solution before the gauge fix (pose 77 | speed-bias 99 | inverse depths | ex_pose 7 | td |
relo_Pose 7) and the costs (start, end, after every iteration) as double-double pairs, the decisions, and how far the FP64 run of the same
code (solve_trace_x.npz) lands from this one.  tests/test_solve_trace_mp.py compares the binary128 arbiter (-DAVM_X members on) with it.

    python tests/golden/gen_solve_trace_x_mp.py [cases]      (all four: 40 minutes; "0,2" runs two; "merge" joins the parts)
"""
import glob
import os
import sys
import time

import mpmath as mp
import numpy as real_np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_solve_trace_mp as M  # noqa: E402

MPF = mp.mpf


def main():
    import gen_golden as GG
    import gen_solve_trace as GS
    import gen_solve_trace_x as GX

    proxy = M.NpProxy()
    for mod in (GG, GS, GX):
        mod.np = proxy
        mod.float = lambda v: v
    GS.SQRT_INFO = GX.SQRT_INFO = MPF(460.0 / 1.5)
    GS.NOISE = tuple(MPF(v) for v in GS.NOISE)
    GS.G = M.obj(real_np.array([0.0, 0.0, 9.81007]))
    gold = real_np.load(os.path.join(HERE, "solve_trace_x.npz"))
    opt = {k: float(gold["opt_" + k]) for k in GS.OPT}
    opt["max_num_iterations"], opt["max_num_consecutive_invalid_steps"] = int(opt["max_num_iterations"]), int(opt["max_num_consecutive_invalid_steps"])
    arg = sys.argv[1] if len(sys.argv) > 1 else "all"
    out = {}
    if arg != "merge":
        cases = list(range(int(gold["n_cases"]))) if arg == "all" else [int(v) for v in arg.split(",")]
        for c in cases:
            t0 = time.time()
            a = {k[len(f"c{c}_in_"):]: gold[k][0] for k in gold.files if k.startswith(f"c{c}_in_")}
            am = {k: (M.obj(v) if v.dtype.kind == "f" else v) for k, v in a.items()}
            ex, td, relo = int(gold[f"c{c}_est_ex"]), int(gold[f"c{c}_est_td"]), bool(gold[f"c{c}_relo"])
            P = GX.ProblemX(am, ex, td)
            x0 = dict(pose=am["pose"].copy(), sb=am["speedbias"].copy(), lam=am["inv_depth"][: P.nf].copy(), ex=am["ex_pose"].copy(),
                      td=(am["td"].ravel()[0] if td else MPF(0)), relo=am["relo_pose"].copy() if relo else M.obj(real_np.array([0, 0, 0, 0, 0, 0, 1.0])))
            x, tr = GS.trust_region_solve(P, x0, opt)
            acc = [bool(v) for v in tr["accepted"]]
            print(f"case {c} (ex {ex} td {td} relo {relo}): {time.time() - t0:.0f} s, iterations {tr['num_iterations']} termination {tr['termination']} accepted {acc}", flush=True)
            xv = list(x["pose"].ravel()) + list(x["sb"].ravel()) + list(x["lam"].ravel()) + list(x["ex"].ravel()) + [x["td"]] + list(x["relo"].ravel())
            costs = [tr["initial_cost"], tr["final_cost"]] + [v for v in tr["cost"]]
            for nm, vals in (("x", xv), ("cost", costs)):
                vals = [v if isinstance(v, mp.mpf) else MPF(float(v)) for v in vals]
                hi = real_np.array([float(v) for v in vals])
                lo = real_np.array([float(v - MPF(h)) for v, h in zip(vals, hi)])
                out[f"c{c}_{nm}_hi"], out[f"c{c}_{nm}_lo"] = hi, lo
            out[f"c{c}_accepted"] = real_np.array(acc)
        if arg != "all":
            real_np.savez_compressed(os.path.join(HERE, "solve_trace_x_mp.part_" + "_".join(str(c) for c in cases) + ".npz"), **out)
            return
    else:
        for f in sorted(glob.glob(os.path.join(HERE, "solve_trace_x_mp.part_*.npz"))):
            part = real_np.load(f)
            out.update({k: part[k] for k in part.files})
    cs = sorted({int(k[1:k.index("_")]) for k in out if k.startswith("c") and k.endswith("_x_hi")})
    for c in cs:   # the FP64 run of the same code against this one (the measures of gen_solve_trace_mp.py)
        dd = lambda nm: [MPF(float(h)) + MPF(float(l)) for h, l in zip(out[f"c{c}_{nm}_hi"], out[f"c{c}_{nm}_lo"])]
        cost, x = dd("cost")[2:], dd("x")
        fc = gold[f"c{c}_trace_cost"]
        fx = real_np.concatenate([gold[f"c{c}_sol_pose"].ravel(), gold[f"c{c}_sol_speedbias"].ravel(), gold[f"c{c}_sol_inv_depth"].ravel(),
                                  gold[f"c{c}_sol_ex_pose"].ravel(), real_np.atleast_1d(gold[f"c{c}_sol_td"]), gold[f"c{c}_sol_relo_pose"].ravel()])
        same = len(fc) == len(cost) and gold[f"c{c}_trace_accepted"].astype(bool).tolist() == out[f"c{c}_accepted"].astype(bool).tolist() and len(fx) == len(x)
        dc = max(abs(MPF(float(a)) - b) for a, b in zip(fc, cost)) / max(abs(b) for b in cost) if same else MPF("inf")
        dx = max(abs(MPF(float(a)) - b) for a, b in zip(fx, x)) / max(abs(b) for b in x) if same else MPF("inf")
        out[f"c{c}_fp64_cost_rel"], out[f"c{c}_fp64_state_rel"] = real_np.float64(float(dc)), real_np.float64(float(dx))
        print(f"case {c}: the FP64 run of the same code is {float(dc):.1e} (costs) / {float(dx):.1e} (solution) from this one")
    out["cases"], out["dps"] = real_np.array(cs, real_np.int64), real_np.int64(mp.mp.dps)
    real_np.savez_compressed(os.path.join(HERE, "solve_trace_x_mp.npz"), **out)
    print("wrote solve_trace_x_mp.npz with cases", cs)


if __name__ == "__main__":
    main()
