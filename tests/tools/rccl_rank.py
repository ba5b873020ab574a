"""One rank of the library's RCCL gather: solve a block of windows on the GPU, then avm_gather_states the final poses.
usage: rccl_rank.py <rank> <n_ranks> <id file> <out npz> [device]   (rank 0 writes the ncclUniqueId to the id file)"""
import importlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
PKG = "anticipated-vins-mono_amd"
rank, n_ranks, id_file, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
device = int(sys.argv[5]) if len(sys.argv) > 5 else 0
import torch

abi, synth, buffers, lib_m, est_m = (importlib.import_module(PKG + "." + m) for m in ("abi", "synth", "buffers", "lib", "estimator"))
ctx = lib_m.Context(device)
if rank == 0:
    uid = ctx.comm_unique_id()
    with open(id_file + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(id_file + ".tmp", id_file)
else:
    t0 = time.time()
    while not os.path.exists(id_file):
        if time.time() - t0 > 60:
            raise SystemExit("no id file")
        time.sleep(0.05)
    uid = open(id_file, "rb").read()
try:
    ctx.comm_init(n_ranks, rank, uid)
except lib_m.AvmError as e:
    np.savez(out, error=str(e))
    raise SystemExit(3)
W = 4
o = abi.default_options()
o.marginalization_flag = abi.MARGIN_NONE
w = synth.make_windows(W, first_id=rank * W, tracks="sparse", n_feat=40, max_feat=150).to_device(f"cuda:{device}")
est_m.Estimator(ctx=ctx, options=o).optimization(w)          # HIP compute ...
recv = torch.zeros((n_ranks * W, 11, 7), dtype=torch.float64, device=f"cuda:{device}")
ctx.gather_states(w.a["pose"], recv, W * 77)                  # ... and the collective on the same ctx stream
torch.cuda.synchronize()
np.savez(out, gathered=recv.cpu().numpy(), mine=w.a["pose"].cpu().numpy())
ctx.comm_destroy()
