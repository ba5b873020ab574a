"""ctypes loader for the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes as C
import importlib
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build():
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libavm_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
    return _LIB


def use_library(path):
    """Switch to another build of the same oracle (bench.py compiles one with -march=native on the host it times)."""
    global _LIB
    _LIB = C.CDLL(path)
    return _LIB


def _pkg():
    return importlib.import_module("anticipated-vins-mono_amd")


def window_solve(opt, win, prior_out=None, summary=None, n_threads=1):
    """In-place Estimator::optimization() on host WindowArrays."""
    from_buffers = importlib.import_module("anticipated-vins-mono_amd.buffers")
    s = win.struct()
    po = prior_out.struct() if prior_out is not None else None
    sp = from_buffers.summary_ptr(summary) if summary is not None else None
    rc = lib().avmo_window_solve_batch(C.byref(opt), C.byref(s), C.byref(po) if po is not None else None, sp, int(n_threads))
    assert rc == 0
    return rc


def fsel_select(fsel, out, n_threads=1):
    s, o = fsel.struct(), out.struct()
    n = C.c_int64(0)
    rc = lib().avmo_fsel_select_batch(C.byref(s), C.byref(o), int(n_threads), C.byref(n))
    assert rc == 0
    return n.value


def preintegrate(opt, win):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    B = win.n_windows
    d, j, cv, sd, sq = np.zeros((B, 10, 10)), np.zeros((B, 10, 15, 15)), np.zeros((B, 10, 15, 15)), np.zeros((B, 10)), np.zeros((B, 10, 15, 15))
    s = win.struct()
    rc = lib().avmo_imu_preintegrate_batch(C.byref(opt), C.byref(s), abi.dptr(d), abi.dptr(j), abi.dptr(cv), abi.dptr(sd), abi.dptr(sq))
    assert rc == 0
    return d, j, cv, sd, sq


def eval_factors(opt, win, apply_loss=False):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    B, mo, mp = win.n_windows, win.dims["max_obs"], win.dims["max_prior"]
    out = dict(proj_r=np.zeros((B, mo, 2)), proj_J=np.zeros((B, mo, 2, 13)), imu_r=np.zeros((B, 10, 15)),
               imu_J=np.zeros((B, 10, 15, 30)), prior_res=np.zeros((B, mp)), cost=np.zeros(B))
    s = win.struct()
    rc = lib().avmo_window_eval_factors(C.byref(opt), C.byref(s), int(apply_loss),
                                        *[abi.dptr(out[k]) for k in ("proj_r", "proj_J", "imu_r", "imu_J", "prior_res", "cost")])
    assert rc == 0
    return out


def fsel_information(fsel):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    P, H, mc = fsel.n_problems, fsel.dims["horizon"], fsel.dims["max_cand"]
    N, T = 9 * (H + 1), 3 * H
    om, dl, va = np.zeros((P, N, N)), np.zeros((P, mc, T, T)), np.zeros((P, mc), np.int32)
    s = fsel.struct()
    rc = lib().avmo_fsel_information(C.byref(s), abi.dptr(om), abi.dptr(dl), abi.iptr(va))
    assert rc == 0
    return om, dl, va


def set_fast_linalg(on: bool):
    """bench.py's second CPU leg ("port_blocked"): vectorisable Schur / Cholesky / forward-substitution loops, bit-identical results."""
    lib().avmo_set_fast_linalg(1 if on else 0)


def fsel_nn_depth(fsel):
    """findNNDepth of every candidate, [P, max_cand]."""
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    out = np.zeros((fsel.n_problems, fsel.dims["max_cand"]))
    s = fsel.struct()
    assert lib().avmo_fsel_nn_depth(C.byref(s), abi.dptr(out)) == 0
    return out


def triangulate(win, init_depth=5.0, n_threads=1):
    """FeatureManager::triangulate in place on host WindowArrays (inverse depths <= 0 are replaced)."""
    s = win.struct()
    L = lib()
    L.avmo_triangulate_batch.argtypes = [C.c_void_p, C.c_double, C.c_int]
    rc = L.avmo_triangulate_batch(C.byref(s), float(init_depth), int(n_threads))
    assert rc == 0


def smallest_right_singular_vector(A):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    A = np.ascontiguousarray(A, float)
    v = np.zeros(4)
    rc = lib().avmo_smallest_right_singular_vector(int(A.shape[0]), abi.dptr(A), abi.dptr(v))
    assert rc == 0
    return v


def imu_propagate(win, g):
    """Estimator::processIMU dead-reckoning of the newest frame, in place on host WindowArrays."""
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    s = win.struct()
    gg = np.ascontiguousarray(g, float)
    rc = lib().avmo_imu_propagate_batch(C.byref(s), abi.dptr(gg))
    assert rc == 0


def fsel_horizon_imu(horizon, k_pos, k_quat, k_ba, k1_pos, k1_vel, k1_quat, acc, gyr, nr_imu, delta_imu):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    s = abi.horizon_in(horizon, k_pos, k_quat, k_ba, k1_pos, k1_vel, k1_quat, acc, gyr, nr_imu, delta_imu)
    hp, hq = np.zeros((s.n_problems, horizon + 1, 3)), np.zeros((s.n_problems, horizon + 1, 4))
    rc = lib().avmo_fsel_horizon_imu(C.byref(s), abi.dptr(hp), abi.dptr(hq))
    assert rc == 0
    return hp, hq


def projection_td_eval(arrays, tr, row, focal_length=460.0):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    f = abi.td_factor_batch(arrays, tr, row, focal_length)
    r, J = np.zeros((f.n, 2)), np.zeros((f.n, 2, 20))
    rc = lib().avmo_projection_td_eval(C.byref(f), abi.dptr(r), abi.dptr(J))
    assert rc == 0
    return r, J


def fsel_build_cloud(win, k1_pos, k1_quat, max_cloud=150):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    B = win.n_windows
    kp, kq = np.ascontiguousarray(k1_pos, float), np.ascontiguousarray(k1_quat, float)
    n, xy, dep = np.zeros(B, np.int32), np.zeros((B, max_cloud, 2)), np.zeros((B, max_cloud))
    s = win.struct()
    rc = lib().avmo_fsel_build_cloud(C.byref(s), abi.dptr(kp), abi.dptr(kq), int(max_cloud), abi.iptr(n), abi.dptr(xy), abi.dptr(dep))
    assert rc == 0
    return n, xy, dep


class GroundTruth:
    def __init__(self, rows):
        import numpy as np
        abi = importlib.import_module("anticipated-vins-mono_amd.abi")
        r = np.ascontiguousarray(rows, float)
        L = lib()
        L.avmo_gt_from_rows.restype = C.c_void_p
        L.avmo_gt_from_rows.argtypes = [abi.c_dp, C.c_int]
        L.avmo_gt_free.argtypes = [C.c_void_p]
        L.avmo_gt_seek.argtypes = [C.c_void_p]
        L.avmo_fsel_horizon_ground_truth.argtypes = [C.c_void_p, C.c_int, C.c_double, abi.c_dp, abi.c_dp, C.c_double, abi.c_dp, abi.c_dp]
        self._L, self._abi = L, abi
        self._g = L.avmo_gt_from_rows(abi.dptr(r), r.shape[0])

    @property
    def seek_idx(self):
        return int(self._L.avmo_gt_seek(self._g))

    def horizon(self, H, t0, k_pos, k_quat, deltaFrame):
        import numpy as np
        kp, kq = np.ascontiguousarray(k_pos, float), np.ascontiguousarray(k_quat, float)
        pos, quat = np.zeros((H + 1, 3)), np.zeros((H + 1, 4))
        rc = self._L.avmo_fsel_horizon_ground_truth(self._g, H, float(t0), self._abi.dptr(kp), self._abi.dptr(kq), float(deltaFrame),
                                                    self._abi.dptr(pos), self._abi.dptr(quat))
        return rc, pos, quat

    def __del__(self):
        try:
            self._L.avmo_gt_free(self._g)
        except Exception:
            pass


def image_from_pointcloud(points, channels, num_cam=1):
    import numpy as np
    abi = importlib.import_module("anticipated-vins-mono_amd.abi")
    pts = np.ascontiguousarray(points, np.float32).reshape(-1, 3)
    n = pts.shape[0]
    ch = [np.ascontiguousarray(c, np.float32) for c in channels]
    arr = (C.POINTER(C.c_float) * 6)(*[c.ctypes.data_as(C.POINTER(C.c_float)) for c in ch])
    fid, cam, out = np.zeros(n, np.int32), np.zeros(n, np.int32), np.zeros((n, 8))
    L = lib()
    L.avmo_image_from_pointcloud.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.POINTER(C.c_float)), C.c_int, abi.c_ip, abi.c_ip, abi.c_dp]
    rc = L.avmo_image_from_pointcloud(n, pts.ctypes.data_as(C.POINTER(C.c_float)), arr, int(num_cam), abi.iptr(fid), abi.iptr(cam), abi.dptr(out))
    return rc, fid, cam, out


def slide_window(win, flag, shift_depth=True, init_depth=5.0):
    """Estimator::slideWindow on host WindowArrays, in place; returns the status (0, or -3 on sample-capacity overflow)."""
    s = win.struct()
    L = lib()
    L.avmo_slide_window.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
    return L.avmo_slide_window(C.byref(s), int(flag), int(bool(shift_depth)), float(init_depth))
