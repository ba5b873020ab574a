#!/usr/bin/env python
"""bench.py — sliding-window solves/sec (10 KF, 150 features) + feature-select ms/frame on MI355X.

A "step" is one pass of the hot path over one batch of synthetic EuRoC-shaped inputs that are
already resident in HBM: avm_window_solve_batch() over `--windows` independent 11-frame windows
per GPU (BASELINE.json configs[3]: 4096 windows / GPU; configs[4]: 8 x 4096 = 32768 over 8 GPUs).
Windows shard embarrassingly: rank r owns window ids [r*W, (r+1)*W); the only collective is one
RCCL all-gather of the final poses per step ("weak" scaling: per-GPU work fixed).

Prints ONE JSON line on rank 0 (see the driver contract in the task statement), with
  roofline     : FP64 FLOP model of the window-solve kernel / its HIP-event duration vs the
                 78.6 TFLOP/s FP64 peak (vector == matrix rate on MI355X); model in DESIGN.md
  cpu_baseline : the CPU oracle ("port" of the reference algorithm) timed on the host cores
"""
import argparse
import importlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
PKG = "anticipated-vins-mono_amd"

FP64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 vector == FP64 matrix; FP32 vector 157.3 / 2
HBM_PEAK_GBS = 8000.0


def flop_model(n_fac, n_feat, summ):
    """Algorithmic FP64 FLOPs of the solves in `summ` (numpy structured summaries). See DESIGN.md §Roofline."""
    import numpy as np

    it = summ["num_iterations"].astype(np.float64)
    ns = summ["num_successful"].astype(np.float64)
    jac_evals = 1.0 + ns
    per_jac = 1800.0 * n_fac + 5.0e5                       # factor r/J + J^T J blocks, 10 IMU factors + prior
    per_lin = 66.0 * 67.0 * n_feat + 165.0**3 / 3.0 + 2.0 * 165.0**2  # Schur rank-150 update + Cholesky + solves
    per_cand = 230.0 * n_fac + 3.0e4                       # residual-only evaluation
    return float((jac_evals * (per_jac + per_lin) + it * per_cand).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--windows", type=int, default=4096, help="windows per GPU per step")
    ap.add_argument("--distinct", type=int, default=128, help="distinct generated windows per rank (tiled up to --windows)")
    ap.add_argument("--tracks", default="dense", choices=["dense", "sparse"])
    ap.add_argument("--fsel-problems", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fsel", action="store_true")
    args = ap.parse_args()

    import numpy as np
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if world > 1:
        import torch.distributed as dist  # noqa: F811

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the product path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"

    abi = importlib.import_module(PKG + ".abi")
    synth = importlib.import_module(PKG + ".synth")
    buffers = importlib.import_module(PKG + ".buffers")
    est_m = importlib.import_module(PKG + ".estimator")
    fs_m = importlib.import_module(PKG + ".feature_selector")
    lib_m = importlib.import_module(PKG + ".lib")

    W = args.windows
    opt = abi.default_options()
    opt.marginalization_flag = abi.MARGIN_NONE if os.environ.get("AVM_BENCH_NO_MARG") else opt.marginalization_flag
    ctx = lib_m.Context(local_rank)
    E = est_m.Estimator(ctx=ctx, options=opt)

    # ---- inputs: generated on-rank from (seed, window id), then resident in HBM
    base = synth.make_windows(min(args.distinct, W), first_id=rank * W, tracks=args.tracks)
    host = synth.tile_windows(base, W)
    n_fac = float((host.a["feat_nobs"] - 1).clip(min=0).sum(1).mean())
    n_feat = float(host.a["n_feat"].mean())
    win = host.to_device(dev)
    pristine = {k: win.a[k].clone() for k in ("pose", "speedbias", "ex_pose", "inv_depth")}
    gathered = torch.empty((world * W, 11, 7), dtype=torch.float64, device=dev) if world > 1 else None

    marg = opt.marginalization_flag != abi.MARGIN_NONE
    prior_slots = buffers.PriorOutArrays.alloc(W, win.dims["max_prior"], win.dims["max_pblk"], dev) if marg else None

    def step():
        for k, v in pristine.items():
            win.a[k].copy_(v)
        summ = E.optimization(win, want_summary=True, prior_out=prior_slots)
        if world > 1:
            dist.all_gather_into_tensor(gathered, win.a["pose"])
        return summ

    kernel_ms, summ = [], None
    for _ in range(args.warmup):
        summ = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        summ = step()
        kernel_ms.append(ctx.kernel_ms("window_solve"))
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    value = world * W * args.steps / elapsed

    result = None
    if rank == 0:
        s = buffers.summary_to_numpy(summ)
        flops = flop_model(n_fac, n_feat, s)
        k_ms = float(np.mean(kernel_ms))
        achieved = flops / (k_ms * 1e-3) / 1e12
        traffic, traffic_src = None, None
        tp = sorted(p for p in os.listdir(os.path.join(ROOT, "profiles")) if p.endswith("_pmc_traffic.json"))
        tp = os.path.join(ROOT, "profiles", tp[-1]) if tp else ""  # newest committed rocprofv3 --pmc summary
        if os.path.exists(tp) and W == 4096 and args.tracks == "dense" and opt.marginalization_flag == abi.MARGIN_OLD:
            # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this same command (scripts/gpu_profile.sh), per launch
            tj = json.load(open(tp))["window_solve_kernel"]
            traffic = tj["traffic_bytes_per_launch"]
            traffic_src = f"profiles/{os.path.basename(tp)}: (2*FETCH_SIZE + WRITE_SIZE) KB per launch, separate --pmc passes"
        alg_bytes = W * (44.0 * n_fac + 23.0e3 + 45.6e3 + 3.0e3 + 2.6e3)  # ~140 KB / solve at K=1500
        result = {
            "metric": "sliding-window solves/sec (10 KF, 150 feats)",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{W} independent 11-frame windows per GPU, {int(n_feat)} features, {int(n_fac)} projection factors "
                            f"({args.tracks} tracks), 10 IMU factors (200 Hz raw samples pre-integrated on device), 75-dim prior, "
                            f"max_num_iterations=8, no time cap; BASELINE.json configs[3] (x{world} GPUs = configs[4] shape)",
                "windows_per_gpu": W,
                "distinct_windows_per_gpu": int(base.n_windows),
                "marginalization": "MARGIN_OLD" if opt.marginalization_flag == abi.MARGIN_OLD else "none",
                "mean_iterations": float(s["num_iterations"].mean()),
                "mean_successful_steps": float(s["num_successful"].mean()),
            },
            "roofline": {
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP64_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP64_PEAK_TFLOPS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "window_solve_kernel",
                "kernel_ms": k_ms,
                "flops_per_launch": flops,
                "hbm_secondary": {"algorithmic_bytes_per_launch": alg_bytes, "achieved_GBs": alg_bytes / (k_ms * 1e-3) / 1e9,
                                  "peak_GBs": HBM_PEAK_GBS},
            },
            "kernel_ms": {k: ctx.kernel_ms(k) for k in ("preint", "window_solve", "marginalize", "prior_eig")},
        }

    # ---- feature selector: ms/frame (batch throughput) and single-frame latency
    if not args.no_fsel:
        FS = fs_m.FeatureSelector(ctx=ctx)
        P = args.fsel_problems
        fp = synth.make_fsel(P, first_id=rank * P).to_device(dev)
        f1 = synth.make_fsel(1, first_id=rank * P).to_device(dev)
        FS.select_batch(fp)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            FS.select_batch(fp)
        torch.cuda.synchronize()
        tb = (time.perf_counter() - t1) / reps
        FS.select_batch(f1)
        t1 = time.perf_counter()
        for _ in range(reps):
            FS.select_batch(f1)
        torch.cuda.synchronize()
        tl = (time.perf_counter() - t1) / reps
        if rank == 0:
            result["feature_select"] = {
                "workload": "500 candidates -> 150 selected, horizon 10 (BASELINE.json configs[2])",
                "ms_per_frame_batched": tb / P * 1e3,
                "batch": P,
                "ms_per_frame_single": tl * 1e3,
            }

    # ---- CPU baseline (oracle = port of the reference algorithm), rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle_py

        cores = os.cpu_count() or 1
        nsamp = min(W, max(8 * cores, 64))
        sample = host.slice(0, nsamp).copy()
        o2 = abi.default_options()
        o2.marginalization_flag = opt.marginalization_flag
        po = buffers.PriorOutArrays.alloc(nsamp) if o2.marginalization_flag != abi.MARGIN_NONE else None
        t2 = time.perf_counter()
        oracle_py.window_solve(o2, sample, po, buffers.summary_alloc(nsamp), n_threads=cores)
        tc = time.perf_counter() - t2
        s1 = host.slice(0, min(4, nsamp)).copy()
        po1 = buffers.PriorOutArrays.alloc(s1.n_windows) if po is not None else None
        t2 = time.perf_counter()
        oracle_py.window_solve(o2, s1, po1, buffers.summary_alloc(s1.n_windows), n_threads=1)
        t1c = (time.perf_counter() - t2) / s1.n_windows
        result["cpu_baseline"] = {
            "value": nsamp / tc,
            "unit": "solves/s",
            "cores": cores,
            "kind": "port",
            "sample": f"{nsamp} of the same windows, one single-threaded solve per host thread ({cores} threads); "
                      f"1-thread rate {1.0 / t1c:.1f} solves/s",
        }
        result["gpu_over_cpu"] = value / (nsamp / tc)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
