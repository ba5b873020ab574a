"""Host-side format adapters (SURVEY 8(f)4, B4 ground-truth mode) over the C ABI: no device, no context.

HorizonGenerator mirrors utility/horizon_generator.{h,cpp} (loadGroundTruth :169-196, groundTruth :73-123; the IMU mode
lives on the device: FeatureSelector.generateFutureHorizon); image_from_pointcloud mirrors the feature-message decode of
estimator_node.cpp:303-321."""
import ctypes as C

import numpy as np

from . import abi
from .lib import AvmError, lib


class HorizonGenerator:
    def __init__(self):
        self._L = lib()
        self._gt = None

    def loadGroundTruth(self, data_csv: str):
        self._free()
        self._gt = self._L.avm_gt_load_csv(str(data_csv).encode())
        if not self._gt:
            raise AvmError(f"avm_gt_load_csv({data_csv!r}) failed")
        return int(self._L.avm_gt_size(self._gt))

    def setGroundTruth(self, rows):
        """rows [n, 17]: timestamp [ns], p, q (w x y z), v, w, a - the columns of the EuRoC data.csv."""
        self._free()
        r = np.ascontiguousarray(rows, float)
        assert r.ndim == 2 and r.shape[1] == 17
        self._gt = self._L.avm_gt_from_rows(abi.dptr(r), r.shape[0])
        if not self._gt:
            raise AvmError("avm_gt_from_rows failed")

    @property
    def seek_idx(self):
        return int(self._L.avm_gt_seek(self._gt))

    def groundTruth(self, horizon, timestamp_k, k_pos, k_quat, deltaFrame):
        """state_kkH of HorizonGenerator::groundTruth: (pos [H+1, 3], quat [H+1, 4] x y z w)."""
        if not self._gt:
            raise AvmError("no ground truth loaded")
        kp, kq = np.ascontiguousarray(k_pos, float), np.ascontiguousarray(k_quat, float)
        pos, quat = np.zeros((horizon + 1, 3)), np.zeros((horizon + 1, 4))
        rc = self._L.avm_fsel_horizon_ground_truth(self._gt, int(horizon), float(timestamp_k), abi.dptr(kp), abi.dptr(kq), float(deltaFrame),
                                                   abi.dptr(pos), abi.dptr(quat))
        if rc != abi.AVM_OK:
            raise AvmError(f"avm_fsel_horizon_ground_truth failed with status {rc}")
        return pos, quat

    def _free(self):
        if getattr(self, "_gt", None):
            self._L.avm_gt_free(self._gt)
            self._gt = None

    def __del__(self):
        try:
            self._free()
        except Exception:
            pass


def image_from_pointcloud(points, channels, num_cam=1):
    """sensor_msgs::PointCloud (points [n, 3] float32, six float32 channels) -> (feature_id [n], camera_id [n],
    xyz_uv_velocity [n, 8]) in image_t order."""
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    n = pts.shape[0]
    ch = [np.ascontiguousarray(c, np.float32) for c in channels]
    assert len(ch) == 6 and all(c.shape == (n,) for c in ch)
    arr = (C.POINTER(C.c_float) * 6)(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in ch])
    fid, cam, out = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 8))
    rc = lib().avm_image_from_pointcloud(n, pts.ctypes.data_as(C.POINTER(C.c_float)), arr, int(num_cam), abi.iptr(fid), abi.iptr(cam), abi.dptr(out))
    if rc != abi.AVM_OK:
        raise AvmError(f"avm_image_from_pointcloud failed with status {rc}")
    return fid, cam, out
