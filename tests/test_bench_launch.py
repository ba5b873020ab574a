"""bench.py --gpus N starts its own ranks.

CPU tier: `--launch-check` runs the launcher, the rendezvous of the ranks (gloo) and the max-over-ranks reduction, and exits
before HIP is initialised; a failing rank makes the whole call fail.
GPU tier: two ranks on two DIFFERENT devices through the library's raw-RCCL gather (skipped on a one-GPU box), and the
self-launched two-rank bench line itself."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(extra)
    return env


def _json_line(out):
    lines = [ln for ln in out.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out
    return json.loads(lines[0])


@pytest.mark.parametrize("n", [2, 3])
def test_self_launch_rendezvous(n):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--launch-check"], env=_clean_env(), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr
    j = _json_line(r.stdout)
    assert j["launch_check"] and j["self_launched"] and j["n_gpus"] == n and j["ranks"] == list(range(n)) and j["max_over_ranks"] == float(n)


def test_a_failing_rank_fails_the_whole_call():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_clean_env(AVM_BENCH_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode != 0 and "rank 1 failed" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_under_an_external_launcher_it_does_not_spawn_again():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--launch-check"], env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and _json_line(r.stdout)["self_launched"] is False
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--launch-check"], env=_clean_env(RANK="0", LOCAL_RANK="0", WORLD_SIZE="4"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=4" in r.stderr


def _n_devices():
    import torch

    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.gpu
def test_two_ranks_on_two_devices_solve_and_gather(tmp_path):
    if _n_devices() < 2:
        pytest.skip("one visible GPU: two ranks need two devices (RCCL refuses two ranks on one device)")
    idf = str(tmp_path / "nccl_id")
    outs = [str(tmp_path / f"r{r}.npz") for r in range(2)]
    env = _clean_env(NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
    tool = os.path.join(ROOT, "tests", "tools", "rccl_rank.py")
    procs = [subprocess.Popen([sys.executable, tool, str(r), "2", idf, outs[r], str(r)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
             for r in range(2)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0].decode(errors="replace"))
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            pytest.fail("two-rank RCCL run timed out")
    assert all(p.returncode == 0 for p in procs), logs
    a, b = np.load(outs[0]), np.load(outs[1])
    assert np.array_equal(a["gathered"], b["gathered"])
    assert np.array_equal(a["gathered"][:4], a["mine"]) and np.array_equal(a["gathered"][4:], b["mine"])


@pytest.mark.gpu
def test_self_launched_two_gpu_bench_line():
    if _n_devices() < 2:
        pytest.skip("one visible GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "256", "--no-fsel"],
                       env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["pose_gather"].startswith("avm_gather_states")
    assert len(j["config"]["per_rank_window_solve_kernel_ms"]) == 2 and j["config"]["launch"].startswith("self-launched")


@pytest.mark.gpu
def test_self_launched_two_rank_bench_end_to_end_on_whatever_devices_there_are():
    """The whole N > 1 path of bench.py on real HIP work - launcher, per-rank window blocks, solve, the gather, max-over-ranks timing,
    rank 0's JSON line - with the gloo backend, which lets two ranks share the one GPU of the test box (RCCL refuses that; the library's
    own gather is covered by test_rccl.py and, on two devices, by the tests above).  Not a scaling measurement."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--warmup", "1", "--windows", "128", "--no-fsel", "--backend", "gloo",
                        "--gen-procs", "1"], env=_clean_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    j = _json_line(r.stdout)
    assert j["n_gpus"] == 2 and j["config"]["windows_per_gpu"] == 128 and j["value"] > 0
    assert "gloo" in j["config"]["pose_gather"] and len(j["config"]["per_rank_window_solve_kernel_ms"]) == 2
    assert j["config"]["launch"].startswith("self-launched") and "cpu_baseline" not in j


def test_the_committed_rocprof_summary_belongs_to_the_kernel_sources_in_the_tree():
    """bench.py reports roofline.traffic / mfma_util / fabric_GBs from the newest profiles/*_pmc_traffic.json only if that summary was measured
    on the kernel sources that are in the tree now (sha256 of csrc/window_solve.hip, kernels.hpp, devmath.hpp, Makefile stored with it):
    a kernel change without a re-profile (scripts/gpu_profile.sh + scripts/summarize_rocprof.py) fails here instead of reporting stale traffic."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    prof, name, err = b.committed_profile()
    assert err is None, err
    sk = "window_solve_tp_kernel" if "window_solve_tp_kernel" in prof else "window_solve_kernel"  # (the form a 4096-window batch takes)
    assert prof[sk]["traffic_bytes_per_launch"] > 0 and 0 < prof[sk]["mfma_util"] < 1
