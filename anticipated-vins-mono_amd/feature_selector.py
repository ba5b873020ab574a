"""Host-side mirror of FeatureSelector::select() (vins_estimator/src/feature_selector.h:49-50,
feature_selector.cpp:74-202) over the C ABI.

The reference mutates `image` in place and returns {trackedFeatures_, selectedIds}; the host
bookkeeping around the scoring loop (splitOnFeatureId :208-219, the tracked list :108-120,192-196)
is plain index work and stays on the host, the information matrices and the greedy logdet
scoring run on the GPU through avm_fsel_select_batch.
"""
import ctypes as C

import numpy as np

from . import abi, buffers
from .lib import Context


class FeatureSelector:
    def __init__(self, ctx: Context = None, device: int = 0):
        self.ctx = ctx or Context(device)
        self.trackedFeatures_ = []  # feature_selector.h:93
        self.lastFeatureId_ = 0

    def select_batch(self, problems: buffers.FselArrays, want_fvalues: bool = True, want_min_gap: bool = False):
        """Greedy selection for P independent frames. Returns (n_selected[P], ids[P, max_features], fvalues[, min_gap: how firmly every
        round was decided, avm_fsel_out::min_gap])."""
        P, mf = problems.n_problems, problems.dims["max_features"]
        dev = "cuda:%d" % self.ctx.device if problems.on_device else None
        out = buffers.FselOutArrays.alloc(P, mf, dev, want_min_gap=want_min_gap)
        if not want_fvalues:
            out.a["fvalues"] = None
        s, o = problems.struct(), out.struct()
        rc = self.ctx._L.avm_fsel_select_batch(self.ctx.h, problems.mem, C.byref(s), C.byref(o))
        self.ctx.check(rc, "avm_fsel_select_batch")
        return out

    def generateFutureHorizon(self, horizon, k_pos, k_quat, k_ba, k1_pos, k1_vel, k1_quat, acc, gyr, nr_imu, delta_imu):
        """FeatureSelector::generateFutureHorizon in IMU mode = HorizonGenerator::imu (horizon_generator.cpp:25-69) for P
        frames: returns hor_pos [P, H+1, 3], hor_quat [P, H+1, 4] (x y z w), the arrays avm_fsel_batch consumes."""
        s = abi.horizon_in(horizon, k_pos, k_quat, k_ba, k1_pos, k1_vel, k1_quat, acc, gyr, nr_imu, delta_imu)
        P = s.n_problems
        hp, hq = np.zeros((P, horizon + 1, 3)), np.zeros((P, horizon + 1, 4))
        rc = self.ctx._L.avm_fsel_horizon_imu(self.ctx.h, abi.AVM_MEM_HOST, C.byref(s), abi.dptr(hp), abi.dptr(hq))
        self.ctx.check(rc, "avm_fsel_horizon_imu")
        return hp, hq

    def initKDTree(self, windows: buffers.WindowArrays, k1_pos, k1_quat, max_cloud: int = 150):
        """The depth cloud FeatureSelector::initKDTree() builds (feature_selector.cpp:380-433), one per window (host
        arrays): returns n_cloud [B], cloud_xy [B, max_cloud, 2], cloud_depth [B, max_cloud]."""
        assert not windows.on_device
        B = windows.n_windows
        kp, kq = np.ascontiguousarray(k1_pos, float), np.ascontiguousarray(k1_quat, float)
        n, xy, dep = np.zeros(B, np.int32), np.zeros((B, max_cloud, 2)), np.zeros((B, max_cloud))
        s = windows.struct()
        rc = self.ctx._L.avm_fsel_build_cloud(self.ctx.h, windows.mem, C.byref(s), abi.dptr(kp), abi.dptr(kq), int(max_cloud), abi.iptr(n),
                                              abi.dptr(xy), abi.dptr(dep))
        self.ctx.check(rc, "avm_fsel_build_cloud")
        return n, xy, dep

    def information(self, problems: buffers.FselArrays):
        """Omega_kkH (+prior) [P,N,N], compact Delta_ell [P,max_cand,3H,3H], valid [P,max_cand] (host arrays)."""
        assert not problems.on_device
        P, H, mc = problems.n_problems, problems.dims["horizon"], problems.dims["max_cand"]
        N, T = 9 * (H + 1), 3 * H
        om, dl, va = np.zeros((P, N, N)), np.zeros((P, mc, T, T)), np.zeros((P, mc), np.int32)
        s = problems.struct()
        rc = self.ctx._L.avm_fsel_information(self.ctx.h, problems.mem, C.byref(s), abi.dptr(om), abi.dptr(dl), abi.iptr(va))
        self.ctx.check(rc, "avm_fsel_information")
        return om, dl, va

    def nn_depth(self, problems: buffers.FselArrays):
        """FeatureSelector::findNNDepth (feature_selector.cpp:437-459) of every candidate: [P, max_cand] (host array)."""
        assert not problems.on_device
        out = np.zeros((problems.n_problems, problems.dims["max_cand"]))
        s = problems.struct()
        self.ctx.check(self.ctx._L.avm_fsel_nn_depth(self.ctx.h, problems.mem, C.byref(s), abi.dptr(out)), "avm_fsel_nn_depth")
        return out

    def setParameters(self, enable=True, maxFeatures=150, initThresh=0):
        """The bookkeeping half of FeatureSelector::setParameters (feature_selector.cpp:24-34); the noise parameters and
        the horizon mode travel with the FselArrays the problem_builder returns."""
        self.enable_, self.maxFeatures_, self.initThresh_ = bool(enable), int(maxFeatures), int(initThresh)

    def select(self, image: dict, problem_builder, initialized: bool = True):
        """Single-frame select() with the reference's bookkeeping (feature_selector.cpp:74-202): `image` maps feature id ->
        8-vector (x y 1 u v vx vy prob) and is replaced by the subset handed to the back end; `problem_builder(new_ids,
        used_ids)` returns a 1-problem FselArrays for those ids (only called when `initialized`, i.e. solver_flag ==
        NON_LINEAR).  Returns (trackedFeatures_, selectedIds); () when disabled.  include/avm_host.hpp holds the C++
        statement of the same function, including the horizon / depth-cloud marshalling."""
        if not getattr(self, "enable_", True):
            return ()
        ids = sorted(image)
        new_ids = [i for i in ids if i > self.lastFeatureId_]  # splitOnFeatureId
        old = {i: image[i] for i in ids if i <= self.lastFeatureId_}
        if new_ids:
            self.lastFeatureId_ = new_ids[-1]
        subset = {f: old[f] for f in self.trackedFeatures_ if f in old}
        selected = []
        if initialized:
            prob = problem_builder(new_ids, sorted(subset))
            out = self.select_batch(prob).to_host()
            n = int(out.a["n_selected"][0])
            selected = [int(v) for v in out.a["selected_ids"][0, :n]]
            for f in selected:
                subset[f] = image[f]
        elif getattr(self, "firstImage_", True):
            subset = {f: image[f] for f in new_ids}  # the whole first image initializes the back end
            self.trackedFeatures_.extend(new_ids)
            self.firstImage_ = False
        if not initialized and len(subset) < getattr(self, "initThresh_", 0):
            subset.update(old)
        image.clear()
        image.update(subset)
        self.trackedFeatures_.extend(selected)
        return list(self.trackedFeatures_), selected
