"""Caller-owned buffers behind the C ABI structs (include/avm.h).

A *Arrays object owns a dict of arrays — numpy (host) or torch CUDA tensors (HBM
resident) — and builds the matching ctypes struct with raw pointers.  PyTorch is used
only as the device allocator here; no torch types cross the ABI.
"""
import ctypes as C

import numpy as np

from . import abi

_WINDOW_F64 = [
    "pose", "speedbias", "ex_pose", "inv_depth", "obs_xy", "imu_dt", "imu_acc", "imu_gyr",
    "imu_lin_ba", "imu_lin_bg", "prior_J", "prior_r", "prior_x0",
    "obs_vel_td", "td", "relo_xy", "relo_pose", "last_pose0",   # optional members (absent key -> NULL)
]
_WINDOW_I32 = [
    "n_feat", "feat_start", "feat_nobs", "feat_obs_begin", "imu_n", "prior_n", "prior_nblk",
    "prior_blk_kind", "prior_blk_frame",
    "relo_n", "relo_frame", "relo_feat", "failure_occur",       # optional members
]


class _Arrays:
    F64: list = []
    I32: list = []

    def __init__(self, dims: dict, arrays: dict):
        self.dims = dict(dims)
        self.a = dict(arrays)

    @property
    def on_device(self) -> bool:
        first = next(iter(self.a.values()))
        return not isinstance(first, np.ndarray)

    @property
    def mem(self) -> int:
        return abi.AVM_MEM_DEVICE if self.on_device else abi.AVM_MEM_HOST

    def copy(self):
        if self.on_device:
            return type(self)(self.dims, {k: v.clone() for k, v in self.a.items()})
        return type(self)(self.dims, {k: v.copy() for k, v in self.a.items()})

    def to_device(self, device="cuda:0"):
        import torch

        out = {}
        for k, v in self.a.items():
            out[k] = torch.from_numpy(np.ascontiguousarray(v)).to(device) if isinstance(v, np.ndarray) else v.to(device)
        return type(self)(self.dims, out)

    def to_host(self):
        if not self.on_device:
            return self.copy()
        return type(self)(self.dims, {k: v.cpu().numpy() for k, v in self.a.items()})

    def _fill(self, s):
        for k, v in self.dims.items():
            setattr(s, k, v)
        for k in self.F64:
            setattr(s, k, abi.dptr(self.a.get(k)))
        for k in self.I32:
            setattr(s, k, abi.iptr(self.a.get(k)))
        return s


class WindowArrays(_Arrays):
    """avm_window_batch: the inputs/outputs of Estimator::optimization() for B windows."""

    F64, I32 = _WINDOW_F64, _WINDOW_I32

    def struct(self) -> abi.WindowBatch:
        return self._fill(abi.WindowBatch())

    @property
    def n_windows(self) -> int:
        return self.dims["n_windows"]

    def slice(self, lo: int, hi: int) -> "WindowArrays":
        d = dict(self.dims)
        d["n_windows"] = hi - lo
        return WindowArrays(d, {k: v[lo:hi] for k, v in self.a.items()})


class PriorOutArrays(_Arrays):
    F64, I32 = ["J", "r", "x0"], ["n", "nblk", "blk_kind", "blk_frame"]

    @staticmethod
    def alloc(n_windows: int, max_prior: int = 96, max_pblk: int = 16, device=None) -> "PriorOutArrays":
        shapes = {
            "n": ((n_windows,), "i4"),
            "nblk": ((n_windows,), "i4"),
            "blk_kind": ((n_windows, max_pblk), "i4"),
            "blk_frame": ((n_windows, max_pblk), "i4"),
            "J": ((n_windows, max_prior, max_prior), "f8"),
            "r": ((n_windows, max_prior), "f8"),
            "x0": ((n_windows, max_pblk, 9), "f8"),
        }
        if device:
            import torch  # zero-filled in HBM directly (J alone is 74 KB per window: never staged through the host)

            a = {k: torch.zeros(sh, dtype=torch.int32 if dt == "i4" else torch.float64, device=device) for k, (sh, dt) in shapes.items()}
        else:
            a = {k: np.zeros(sh, np.int32 if dt == "i4" else np.float64) for k, (sh, dt) in shapes.items()}
        return PriorOutArrays({"max_prior": max_prior, "max_pblk": max_pblk}, a)

    def struct(self) -> abi.PriorOut:
        return self._fill(abi.PriorOut())


class FselArrays(_Arrays):
    """avm_fsel_batch: the inputs of FeatureSelector::select() for P frames."""

    F64 = ["hor_pos", "hor_quat", "delta_imu", "cand_xy", "cand_prob", "used_xy", "cloud_xy", "cloud_depth"]
    I32 = ["nr_imu", "n_cand", "cand_id", "n_used", "used_id", "n_cloud"]

    def __init__(self, dims, arrays, scalars=None):
        super().__init__(dims, arrays)
        self.scalars = dict(scalars or {})

    def copy(self):
        c = super().copy()
        c.scalars = dict(self.scalars)
        return c

    def to_device(self, device="cuda:0"):
        c = super().to_device(device)
        c.scalars = dict(self.scalars)
        return c

    def struct(self) -> abi.FselBatch:
        s = self._fill(abi.FselBatch())
        for k, v in self.scalars.items():
            if k in ("q_ic", "t_ic"):
                arr = getattr(s, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                setattr(s, k, v)
        return s

    @property
    def n_problems(self) -> int:
        return self.dims["n_problems"]


class FselOutArrays(_Arrays):
    F64, I32 = ["fvalues", "min_gap"], ["n_selected", "selected_ids"]

    @staticmethod
    def alloc(n_problems: int, max_features: int, device=None, want_min_gap: bool = False) -> "FselOutArrays":
        a = {
            "n_selected": np.zeros(n_problems, np.int32),
            "selected_ids": np.full((n_problems, max_features), -1, np.int32),
            "fvalues": np.zeros((n_problems, max_features)),
        }
        if want_min_gap:  # avm_fsel_out::min_gap (nullable: an absent key is a NULL pointer)
            a["min_gap"] = np.zeros((n_problems, max_features))
        o = FselOutArrays({}, a)
        return o.to_device(device) if device else o

    def struct(self) -> abi.FselOut:
        return self._fill(abi.FselOut())


def summary_alloc(n_windows: int, device=None):
    """[B] avm_solve_summary records (numpy structured array, or a raw byte tensor on device)."""
    if device is None:
        return np.zeros(n_windows, abi.SUMMARY_DTYPE)
    import torch

    return torch.zeros(n_windows * abi.SUMMARY_DTYPE.itemsize, dtype=torch.uint8, device=device)


def summary_ptr(s):
    if isinstance(s, np.ndarray):
        return s.ctypes.data_as(C.POINTER(abi.SolveSummary))
    return C.cast(s.data_ptr(), C.POINTER(abi.SolveSummary))


def summary_to_numpy(s):
    if isinstance(s, np.ndarray):
        return s
    return np.frombuffer(s.cpu().numpy().tobytes(), dtype=abi.SUMMARY_DTYPE).copy()
