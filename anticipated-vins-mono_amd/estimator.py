"""Host-side mirror of the reference's Estimator::optimization() call surface
(vins_estimator/src/estimator.h:47, estimator.cpp:661-994) over the C ABI.

The reference method takes no arguments and works on member arrays (para_Pose,
para_SpeedBias, para_Ex_Pose, para_Feature, pre_integrations, f_manager.feature,
last_marginalization_info).  Here those members live in a WindowArrays batch (host numpy or
HBM-resident torch tensors); optimization() solves all B windows in place and hands back the
new prior, exactly like the reference leaves last_marginalization_info behind.
"""
import ctypes as C

import numpy as np

from . import abi, buffers
from .lib import Context


class Estimator:
    def __init__(self, ctx: Context = None, options: abi.Options = None, device: int = 0):
        self.ctx = ctx or Context(device)
        self.options = options or abi.default_options()
        self.last_summary = None
        self.last_marginalization_info = None  # PriorOutArrays after a MARGIN_OLD / SECOND_NEW solve

    # the reference's name
    def optimization(self, windows: buffers.WindowArrays, want_summary: bool = True, prior_out: buffers.PriorOutArrays = None,
                     summary_out=None):
        """Solve all windows in place.  `prior_out` lets a caller that solves batch after batch hand the same
        output slots back in (the reference allocates a new MarginalizationInfo per call; the slots are plain data);
        `summary_out` (buffers.summary_alloc) likewise for the per-window summaries: every record is rewritten by a call."""
        L = self.ctx._L
        B = windows.n_windows
        s = windows.struct()
        dev = "cuda:%d" % self.ctx.device if windows.on_device else None
        summ = (summary_out if summary_out is not None else buffers.summary_alloc(B, dev)) if want_summary else None
        prior = None
        if self.options.marginalization_flag != abi.MARGIN_NONE:
            prior = prior_out or buffers.PriorOutArrays.alloc(B, windows.dims["max_prior"], windows.dims["max_pblk"], dev)
        po = prior.struct() if prior is not None else None
        rc = L.avm_window_solve_batch(self.ctx.h, C.byref(self.options), windows.mem, C.byref(s),
                                      C.byref(po) if po is not None else None,
                                      buffers.summary_ptr(summ) if summ is not None else None)
        self.ctx.check(rc, "avm_window_solve_batch")
        self.last_summary = summ
        if prior is not None and self.options.marginalization_flag == abi.MARGIN_SECOND_NEW:
            self._keep_old_prior_where_nothing_was_dropped(prior, windows)
        self.last_marginalization_info = prior
        return summ

    @staticmethod
    def _keep_old_prior_where_nothing_was_dropped(prior, windows):
        """MARGIN_SECOND_NEW leaves last_marginalization_info untouched when pose[WINDOW_SIZE - 1] is not in the old prior
        (or there is none): estimator.cpp:926-927.  The library reports that as n == -1; those windows keep the prior
        they were solved with (the batch's prior_* tables), so that the returned slots can always be chained."""
        n = prior.a["n"]
        keep = (n < 0)
        if not bool(keep.any()):
            return
        pa, wa = prior.a, windows.a
        mp = min(prior.dims["max_prior"], windows.dims["max_prior"])
        mb = min(prior.dims["max_pblk"], windows.dims["max_pblk"])
        idx = keep.nonzero()[0] if isinstance(n, np.ndarray) else keep.nonzero().flatten()
        for i in idx.tolist():
            pa["n"][i] = wa["prior_n"][i]
            pa["nblk"][i] = wa["prior_nblk"][i]
            pa["blk_kind"][i, :mb] = wa["prior_blk_kind"][i, :mb]
            pa["blk_frame"][i, :mb] = wa["prior_blk_frame"][i, :mb]
            pa["J"][i, :mp, :mp] = wa["prior_J"][i, :mp, :mp]
            pa["r"][i, :mp] = wa["prior_r"][i, :mp]
            pa["x0"][i, :mb] = wa["prior_x0"][i, :mb]

    def triangulate(self, windows: buffers.WindowArrays, init_depth: float = 5.0):
        """FeatureManager::triangulate (feature_manager.cpp:202-257), the step before optimization() in solveOdometry():
        features whose inverse depth is <= 0 get 1 / depth from the multi-view linear triangulation, in place."""
        s = windows.struct()
        rc = self.ctx._L.avm_triangulate_batch(self.ctx.h, windows.mem, C.byref(s), float(init_depth))
        self.ctx.check(rc, "avm_triangulate_batch")

    def projection_td_eval(self, arrays: dict, tr: float, row: float, focal_length: float = 460.0):
        """ProjectionTdFactor::Evaluate (projection_td_factor.cpp:34-141) for n factors given as host arrays
        (abi.td_factor_batch): returns residual [n, 2] and Jacobian [n, 2, 20] (pose_i 6 | pose_j 6 | ex 6 | lambda | td)."""
        f = abi.td_factor_batch(arrays, tr, row, focal_length)
        r, J = np.zeros((f.n, 2)), np.zeros((f.n, 2, 20))
        rc = self.ctx._L.avm_projection_td_eval(self.ctx.h, abi.AVM_MEM_HOST, C.byref(f), abi.dptr(r), abi.dptr(J))
        self.ctx.check(rc, "avm_projection_td_eval")
        return r, J

    def imu_propagate(self, windows: buffers.WindowArrays):
        """Estimator::processIMU's dead-reckoning of the newest frame (estimator.cpp:100-107), in place."""
        s = windows.struct()
        g = (C.c_double * 3)(*[float(x) for x in self.options.g])
        rc = self.ctx._L.avm_imu_propagate_batch(self.ctx.h, windows.mem, C.byref(s), g)
        self.ctx.check(rc, "avm_imu_propagate_batch")

    def slideWindow(self, windows: buffers.WindowArrays, marginalization_flag=None, shift_depth=True, init_depth=5.0):
        """Estimator::slideWindow (estimator.cpp:996-1107) + removeBackShiftDepth / removeBack / removeFront
        (feature_manager.cpp:275-352), in place on the batch tables (host or device resident)."""
        flag = self.options.marginalization_flag if marginalization_flag is None else marginalization_flag
        s = windows.struct()
        rc = self.ctx._L.avm_slide_window(self.ctx.h, windows.mem, C.byref(s), int(flag), int(bool(shift_depth)), float(init_depth))
        self.ctx.check(rc, "avm_slide_window")

    def preintegrate(self, windows: buffers.WindowArrays):
        """IntegrationBase for every interval: returns delta [B,10,10], jacobian, covariance [B,10,15,15], sum_dt [B,10]."""
        assert not windows.on_device
        B = windows.n_windows
        d, j, cv, sd = np.zeros((B, 10, 10)), np.zeros((B, 10, 15, 15)), np.zeros((B, 10, 15, 15)), np.zeros((B, 10))
        s = windows.struct()
        rc = self.ctx._L.avm_imu_preintegrate_batch(self.ctx.h, C.byref(self.options), windows.mem, C.byref(s),
                                                    abi.dptr(d), abi.dptr(j), abi.dptr(cv), abi.dptr(sd))
        self.ctx.check(rc, "avm_imu_preintegrate_batch")
        return d, j, cv, sd

    def sqrt_info(self, n_windows: int):
        out = np.zeros((n_windows, 10, 15, 15))
        self.ctx.check(self.ctx._L.avm_debug_copy_sqrt_info(self.ctx.h, n_windows, abi.dptr(out)), "avm_debug_copy_sqrt_info")
        return out

    def eval_factors(self, windows: buffers.WindowArrays, apply_loss: bool = False):
        assert not windows.on_device
        B, mo, mp = windows.n_windows, windows.dims["max_obs"], windows.dims["max_prior"]
        out = dict(proj_r=np.zeros((B, mo, 2)), proj_J=np.zeros((B, mo, 2, 13)), imu_r=np.zeros((B, 10, 15)),
                   imu_J=np.zeros((B, 10, 15, 30)), prior_res=np.zeros((B, mp)), cost=np.zeros(B))
        s = windows.struct()
        rc = self.ctx._L.avm_window_eval_factors(self.ctx.h, C.byref(self.options), windows.mem, C.byref(s), int(apply_loss),
                                                 *[abi.dptr(out[k]) for k in ("proj_r", "proj_J", "imu_r", "imu_J", "prior_res", "cost")])
        self.ctx.check(rc, "avm_window_eval_factors")
        return out
