"""Loader for the product library libavm_hip.so (HIP/gfx950).  No fallback: a missing library
or a missing GPU raises, it never silently routes to CPU code."""
import ctypes as C
import os
import subprocess

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libavm_hip.so")
_lib = None


class AvmError(RuntimeError):
    pass


def build(force: bool = False):
    """Compile csrc/*.hip for gfx950 with hipcc (works without a GPU)."""
    args = ["make", "-C", os.path.join(_HERE, "csrc"), "-s", "-j4"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise AvmError(
                f"{_LIB_PATH} is missing: build it with __graft_entry__.build() "
                "(hipcc --offload-arch=gfx950). There is no CPU fallback."
            )
        # PyTorch-ROCm ships its own copy of the HIP / HSA runtime under the same sonames.  Whichever is loaded first serves
        # the whole process, and torch cannot find the GPU on top of /opt/rocm's: load torch's first when it is installed,
        # so that `import torch` after this module keeps working (device tensors are how the tests and bench.py hand over
        # HBM-resident buffers).  A C++ host (include/avm_host.hpp) has no such concern.
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(_LIB_PATH)
        vp = C.c_void_p
        L.avm_version.restype = C.c_char_p
        L.avm_last_error.restype = C.c_char_p
        L.avm_last_error.argtypes = [vp]
        L.avm_default_options.argtypes = [C.POINTER(abi.Options)]
        L.avm_create.argtypes = [C.POINTER(abi.Config), C.POINTER(vp)]
        L.avm_destroy.argtypes = [vp]
        L.avm_destroy.restype = None
        L.avm_last_kernel_ms.argtypes = [vp, C.c_char_p, C.POINTER(C.c_float)]
        L.avm_window_solve_batch.argtypes = [vp, C.POINTER(abi.Options), C.c_int, C.POINTER(abi.WindowBatch),
                                             C.POINTER(abi.PriorOut), C.POINTER(abi.SolveSummary)]
        L.avm_window_solve.argtypes = L.avm_window_solve_batch.argtypes
        L.avm_fsel_select.argtypes = [vp, C.c_int, C.POINTER(abi.FselBatch), abi.c_ip, abi.c_ip, abi.c_dp]
        L.avm_fsel_fallback_stats.argtypes = [vp, C.POINTER(C.c_int64)]
        L.avm_imu_preintegrate_batch.argtypes = [vp, C.POINTER(abi.Options), C.c_int, C.POINTER(abi.WindowBatch)] + [abi.c_dp] * 4
        L.avm_window_eval_factors.argtypes = [vp, C.POINTER(abi.Options), C.c_int, C.POINTER(abi.WindowBatch), C.c_int] + [abi.c_dp] * 6
        L.avm_triangulate_batch.argtypes = [vp, C.c_int, C.POINTER(abi.WindowBatch), C.c_double]
        L.avm_imu_propagate_batch.argtypes = [vp, C.c_int, C.POINTER(abi.WindowBatch), C.POINTER(C.c_double)]
        L.avm_fsel_select_batch.argtypes = [vp, C.c_int, C.POINTER(abi.FselBatch), C.POINTER(abi.FselOut)]
        L.avm_fsel_information.argtypes = [vp, C.c_int, C.POINTER(abi.FselBatch), abi.c_dp, abi.c_dp, abi.c_ip]
        L.avm_fsel_nn_depth.argtypes = [vp, C.c_int, C.POINTER(abi.FselBatch), abi.c_dp]
        L.avm_fsel_horizon_imu.argtypes = [vp, C.c_int, C.POINTER(abi.FselHorizonIn), abi.c_dp, abi.c_dp]
        L.avm_projection_td_eval.argtypes = [vp, C.c_int, C.POINTER(abi.TdFactorBatch), abi.c_dp, abi.c_dp]
        L.avm_fsel_build_cloud.argtypes = [vp, C.c_int, C.POINTER(abi.WindowBatch), abi.c_dp, abi.c_dp, C.c_int32, abi.c_ip, abi.c_dp, abi.c_dp]
        L.avm_debug_copy_sqrt_info.argtypes = [vp, C.c_int, abi.c_dp]
        L.avm_debug_last_solve_form.argtypes = [vp]
        L.avm_debug_last_marg_form.argtypes = [vp]
        L.avm_debug_last_fsel_form.argtypes = [vp]
        L.avm_debug_counters.argtypes = [vp, C.POINTER(C.c_int64)]
        L.avm_debug_fsel_evaluations.argtypes = [vp, C.POINTER(C.c_int64)]
        L.avm_debug_solve_tp_occupancy.argtypes = [C.POINTER(C.c_int)]
        L.avm_slide_window.argtypes = [vp, C.c_int, C.POINTER(abi.WindowBatch), C.c_int32, C.c_int32, C.c_double]
        L.avm_comm_unique_id.argtypes = [vp, C.c_void_p]
        L.avm_comm_init.argtypes = [vp, C.c_int32, C.c_int32, C.c_void_p]
        L.avm_gather_states.argtypes = [vp, abi.c_dp, abi.c_dp, C.c_size_t]
        L.avm_comm_destroy.argtypes = [vp]
        L.avm_gt_load_csv.restype = vp
        L.avm_gt_load_csv.argtypes = [C.c_char_p]
        L.avm_gt_from_rows.restype = vp
        L.avm_gt_from_rows.argtypes = [abi.c_dp, C.c_int32]
        L.avm_gt_free.argtypes = [vp]
        L.avm_gt_size.argtypes = [vp]
        L.avm_gt_seek.argtypes = [vp]
        L.avm_fsel_horizon_ground_truth.argtypes = [vp, C.c_int32, C.c_double, abi.c_dp, abi.c_dp, C.c_double, abi.c_dp, abi.c_dp]
        L.avm_image_from_pointcloud.argtypes = [C.c_int32, C.POINTER(C.c_float), C.POINTER(C.POINTER(C.c_float)), C.c_int32, abi.c_ip, abi.c_ip, abi.c_dp]
        _lib = L
    return _lib


EXPORTS = [
    "avm_default_options", "avm_create", "avm_destroy", "avm_last_error", "avm_version", "avm_abi_version",
    "avm_window_solve_batch", "avm_window_solve", "avm_fsel_select", "avm_fsel_fallback_stats", "avm_imu_preintegrate_batch", "avm_window_eval_factors",
    "avm_fsel_select_batch", "avm_fsel_information", "avm_fsel_nn_depth", "avm_last_kernel_ms", "avm_triangulate_batch", "avm_imu_propagate_batch", "avm_fsel_horizon_imu", "avm_projection_td_eval", "avm_fsel_build_cloud",
    "avm_ctx_stream", "avm_comm_unique_id", "avm_comm_init", "avm_gather_states", "avm_comm_destroy", "avm_gt_load_csv", "avm_gt_from_rows", "avm_gt_free", "avm_gt_size", "avm_gt_seek", "avm_fsel_horizon_ground_truth", "avm_image_from_pointcloud", "avm_slide_window",
]


class Context:
    """avm_ctx: one per host thread, owns device scratch and one HIP stream."""

    def __init__(self, device: int = 0):
        self._L = lib()
        cfg = abi.Config()
        cfg.device = device
        cfg.abi_version = abi.AVM_ABI_VERSION
        h = C.c_void_p()
        rc = self._L.avm_create(C.byref(cfg), C.byref(h))
        if rc != abi.AVM_OK:
            raise AvmError(f"avm_create failed with status {rc} (no HIP device? there is no CPU fallback)")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self._L.avm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def check(self, rc: int, what: str):
        if rc != abi.AVM_OK:
            raise AvmError(f"{what} failed: status {rc}: {self._L.avm_last_error(self.h).decode()}")

    # ---- multi-GPU: the library's own RCCL communicator (include/avm.h)
    def comm_unique_id(self) -> bytes:
        buf = C.create_string_buffer(128)
        self.check(self._L.avm_comm_unique_id(self.h, buf), "avm_comm_unique_id")
        return buf.raw

    def comm_init(self, n_ranks: int, rank: int, uid: bytes):
        assert len(uid) == 128
        self.check(self._L.avm_comm_init(self.h, int(n_ranks), int(rank), C.create_string_buffer(uid, 128)), "avm_comm_init")

    def gather_states(self, send, recv, count: int):
        """ncclAllGather of `count` doubles per rank (device tensors): recv[r * count : (r + 1) * count] = rank r's send."""
        self.check(self._L.avm_gather_states(self.h, abi.dptr(send), abi.dptr(recv), int(count)), "avm_gather_states")

    def comm_destroy(self):
        self.check(self._L.avm_comm_destroy(self.h), "avm_comm_destroy")

    def fsel_fallback_stats(self) -> dict:
        """avm_fsel_fallback_stats: how often this ctx's selects fell back from the all-rounds-in-one-launch kernel."""
        out = (C.c_int64 * 4)()
        self.check(self._L.avm_fsel_fallback_stats(self.h, out), "avm_fsel_fallback_stats")
        return {"reruns": int(out[0]), "failed_launches": int(out[1]), "mode": int(out[2]), "calls": int(out[3])}

    def last_solve_form(self) -> str:
        """Which form of the solve kernel the last optimization() took: 'throughput' (two 256-thread workgroups per CU,
        batches larger than the CU count) or 'latency' (one 512-thread workgroup per CU).  AVM_SOLVE_TP=0/1 forces it."""
        return "throughput" if self._L.avm_debug_last_solve_form(self.h) == 1 else "latency"

    def last_marg_form(self) -> str:
        """Which form of the marginalization kernel the last optimization() took: 'throughput' (two 256-thread workgroups per CU; it
        follows the solve's form unless a prior keeps a speed-bias block beyond frame 1, AVM_MARG_TP=0 switches it off) or 'latency'."""
        return "throughput" if self._L.avm_debug_last_marg_form(self.h) == 1 else "latency"

    def last_fsel_form(self) -> str:
        """Which form the last select_batch() started in: 'solo' (one workgroup per frame with lazy evaluation: batches of 33 frames and
        more; AVM_FSEL_SOLO=0/1 forces it), 'teams' (a team of workgroups per frame) or 'rounds' (one launch per round)."""
        return {3: "solo", 2: "teams", 1: "teams", 0: "rounds"}.get(self._L.avm_debug_last_fsel_form(self.h), "none")

    def last_fsel_evaluations(self) -> int:
        """Candidate evaluations the last select_batch() executed on the device: counted by the solo form's kernel (lazy evaluation);
        -1 for the forms that score every live candidate in every round (their count follows from the frame: bench.py)."""
        out = C.c_int64(0)
        self.check(self._L.avm_debug_fsel_evaluations(self.h, C.byref(out)), "avm_debug_fsel_evaluations")
        return int(out.value)

    def counters(self) -> dict:
        """Debug counters of this ctx: device / pinned (re)allocations so far, and how the last marginalization's square roots were taken."""
        out = (C.c_int64 * 4)()
        self.check(self._L.avm_debug_counters(self.h, out), "avm_debug_counters")
        return {"allocations": int(out[0]), "prior_one_wavefront": int(out[1]), "prior_windows": int(out[2]),
                "prior_pivoted_path": int(out[2] - out[1]), "solve_form": "throughput" if out[3] else "latency"}

    def kernel_ms(self, which: str) -> float:
        ms = C.c_float(0)
        self.check(self._L.avm_last_kernel_ms(self.h, which.encode(), C.byref(ms)), "avm_last_kernel_ms")
        return ms.value
