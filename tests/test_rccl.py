"""avm_gather_states - the library's raw-RCCL all-gather of the final poses.

GPU tier: a one-rank communicator next to the solve kernels (always), and two ranks as two processes on the ONE GPU of the
test box - HIP compute and the collective together; RCCL builds that refuse two ranks on one device skip that case with
RCCL's own message.  (The 8-GPU run is the driver's: bench.py --gpus N uses the same call.)
CPU tier: the symbols exist and fail cleanly without a device / without a communicator."""
import importlib
import os
import subprocess
import sys

import numpy as np
import pytest

from helpers import PKG, abi, buffers, synth

lib_m = importlib.import_module(PKG + ".lib")
HERE = os.path.dirname(os.path.abspath(__file__))


def test_comm_entry_points_are_exported():
    L = lib_m.lib()
    for name in ("avm_comm_unique_id", "avm_comm_init", "avm_gather_states", "avm_comm_destroy"):
        assert getattr(L, name)
    assert L.avm_comm_destroy(None) == abi.AVM_ERR_INVALID and L.avm_gather_states(None, None, None, 0) == abi.AVM_ERR_INVALID


@pytest.mark.gpu
def test_single_rank_gather_next_to_the_solve(ctx):
    import torch

    est_m = importlib.import_module(PKG + ".estimator")
    c2 = lib_m.Context(0)   # its own ctx: the session ctx stays without a communicator
    send = torch.arange(77 * 3, dtype=torch.float64, device="cuda:0")
    recv = torch.zeros_like(send)
    with pytest.raises(lib_m.AvmError, match="avm_comm_init"):
        c2.gather_states(send, recv, send.numel())
    c2.comm_init(1, 0, c2.comm_unique_id())
    with pytest.raises(lib_m.AvmError, match="already"):
        c2.comm_init(1, 0, c2.comm_unique_id())
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    w = synth.make_windows(3, tracks="sparse", n_feat=30, max_feat=150).to_device("cuda:0")
    est_m.Estimator(ctx=c2, options=o).optimization(w)
    out = torch.zeros((3, 11, 7), dtype=torch.float64, device="cuda:0")
    c2.gather_states(w.a["pose"], out, 3 * 77)
    assert torch.equal(out, w.a["pose"]) and c2.kernel_ms("gather_states") >= 0.0
    c2.comm_destroy()
    c2.close()


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_solve_and_gather(tmp_path):
    idf = str(tmp_path / "nccl_id")
    outs = [str(tmp_path / f"r{r}.npz") for r in range(2)]
    env = dict(os.environ, NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(HERE, "tools", "rccl_rank.py"), str(r), "2", idf, outs[r]], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=240)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank RCCL run timed out")
    if any(p.returncode == 3 for p in procs):
        msg = "; ".join(str(np.load(o)["error"]) for o in outs if os.path.exists(o) and "error" in np.load(o))
        pytest.skip("this RCCL refuses two ranks on one device: " + msg[:300])
    assert all(p.returncode == 0 for p in procs), logs
    a, b = np.load(outs[0]), np.load(outs[1])
    assert np.array_equal(a["gathered"], b["gathered"])                       # every rank holds every block
    assert np.array_equal(a["gathered"][:4], a["mine"]) and np.array_equal(a["gathered"][4:], b["mine"])
    # and it is the right data: the same windows solved by one process
    est_m = importlib.import_module(PKG + ".estimator")
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_NONE
    w = synth.make_windows(8, first_id=0, tracks="sparse", n_feat=40, max_feat=150)
    est_m.Estimator(options=o).optimization(w)
    assert np.array_equal(w.a["pose"], a["gathered"])                         # bit-exact: shard invariance + the gather
