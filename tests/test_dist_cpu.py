"""CPU tier: the N>1 sharding path with torch.distributed (gloo, world_size 2).
Each rank generates and solves its own contiguous block of window ids (no scatter), then one
all-gather of the final poses; the result must equal the single-process run over all ids.
The per-rank compute here is the CPU oracle (allowed in tests/); bench.py runs the same sharding with
the HIP path over RCCL."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT


def _worker(rank, world, port, per_rank, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import importlib

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    synth = importlib.import_module("anticipated-vins-mono_amd.synth")
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    import oracle_py

    opt = abi.default_options()
    opt.marginalization_flag = abi.MARGIN_NONE
    w = synth.make_windows(per_rank, first_id=rank * per_rank, tracks="sparse", n_feat=12, max_feat=16)
    oracle_py.window_solve(opt, w)
    mine = torch.from_numpy(w.a["pose"].copy())
    gathered = torch.empty((world * per_rank, 11, 7), dtype=torch.float64)
    dist.all_gather_into_tensor(gathered, mine)
    t = torch.tensor([float(rank + 1)])
    dist.all_reduce(t, op=dist.ReduceOp.MAX)  # the max-over-ranks timing reduction bench.py uses
    if rank == 0:
        q.put((gathered.numpy(), float(t.item())))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_equals_single_process(oracle, synth, abi):
    world, per_rank = 2, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, world, port, per_rank, q)) for r in range(world)]
    for p in procs:
        p.start()
    gathered, tmax = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert tmax == 2.0
    opt = abi.default_options()
    opt.marginalization_flag = abi.MARGIN_NONE
    w = synth.make_windows(world * per_rank, first_id=0, tracks="sparse", n_feat=12, max_feat=16)
    oracle.window_solve(opt, w)
    assert np.array_equal(gathered, w.a["pose"])
