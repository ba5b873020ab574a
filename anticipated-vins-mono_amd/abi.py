"""ctypes mirror of include/avm.h (POD structs only).

Field order and types must match the header exactly; tests/test_abi.py checks
sizeof() of every struct against the values the C library reports.
"""
import ctypes as C

import numpy as np

WINDOW_SIZE = 10
NFRAMES = WINDOW_SIZE + 1
MAX_ITER_TRACE = 16

AVM_OK = 0
AVM_MEM_HOST, AVM_MEM_DEVICE = 0, 1
# avm_status (include/avm.h)
AVM_OK, AVM_ERR_INVALID, AVM_ERR_UNSUPPORTED, AVM_ERR_NO_DEVICE, AVM_ERR_HIP, AVM_ERR_CAPACITY = 0, -1, -2, -3, -4, -5
BLK_POSE, BLK_SPEEDBIAS, BLK_EXPOSE, BLK_TD = 0, 1, 2, 3
MARGIN_OLD, MARGIN_SECOND_NEW, MARGIN_NONE = 0, 1, 2
TERM_NAMES = ["NO_CONVERGENCE", "GRADIENT_TOL", "PARAMETER_TOL", "FUNCTION_TOL", "MIN_RADIUS", "FAILURE"]

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)


class Options(C.Structure):
    _fields_ = [
        ("max_num_iterations", C.c_int32),
        ("estimate_extrinsic", C.c_int32),
        ("estimate_td", C.c_int32),
        ("marginalization_flag", C.c_int32),
        ("focal_length", C.c_double),
        ("g", C.c_double * 3),
        ("acc_n", C.c_double),
        ("gyr_n", C.c_double),
        ("acc_w", C.c_double),
        ("gyr_w", C.c_double),
        ("cauchy_a", C.c_double),
        ("max_sum_dt", C.c_double),
        ("initial_trust_region_radius", C.c_double),
        ("max_trust_region_radius", C.c_double),
        ("min_trust_region_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("max_num_consecutive_invalid_steps", C.c_int32),
        ("jacobi_scaling", C.c_int32),
        ("marg_eps", C.c_double),
        ("tr", C.c_double),
        ("row", C.c_double),
        ("max_solver_time_s", C.c_double),
        ("marg_noise_rel", C.c_double),
    ]


class WindowBatch(C.Structure):
    _fields_ = [
        ("n_windows", C.c_int32),
        ("max_feat", C.c_int32),
        ("max_obs", C.c_int32),
        ("max_samp", C.c_int32),
        ("max_prior", C.c_int32),
        ("max_pblk", C.c_int32),
        ("pose", c_dp),
        ("speedbias", c_dp),
        ("ex_pose", c_dp),
        ("inv_depth", c_dp),
        ("n_feat", c_ip),
        ("feat_start", c_ip),
        ("feat_nobs", c_ip),
        ("feat_obs_begin", c_ip),
        ("obs_xy", c_dp),
        ("imu_n", c_ip),
        ("imu_dt", c_dp),
        ("imu_acc", c_dp),
        ("imu_gyr", c_dp),
        ("imu_lin_ba", c_dp),
        ("imu_lin_bg", c_dp),
        ("prior_n", c_ip),
        ("prior_nblk", c_ip),
        ("prior_blk_kind", c_ip),
        ("prior_blk_frame", c_ip),
        ("prior_J", c_dp),
        ("prior_r", c_dp),
        ("prior_x0", c_dp),
        ("obs_vel_td", c_dp),
        ("td", c_dp),
        ("relo_n", c_ip),
        ("relo_frame", c_ip),
        ("relo_feat", c_ip),
        ("relo_xy", c_dp),
        ("relo_pose", c_dp),
        ("failure_occur", c_ip),
        ("last_pose0", c_dp),
    ]


class PriorOut(C.Structure):
    _fields_ = [
        ("max_prior", C.c_int32),
        ("max_pblk", C.c_int32),
        ("n", c_ip),
        ("nblk", c_ip),
        ("blk_kind", c_ip),
        ("blk_frame", c_ip),
        ("J", c_dp),
        ("r", c_dp),
        ("x0", c_dp),
    ]


class SolveSummary(C.Structure):
    _fields_ = [
        ("termination", C.c_int32),
        ("num_iterations", C.c_int32),
        ("num_successful", C.c_int32),
        ("accept_mask", C.c_int32),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("cost_trace", C.c_double * MAX_ITER_TRACE),
        ("radius_trace", C.c_double * MAX_ITER_TRACE),
    ]


SUMMARY_DTYPE = np.dtype(
    [
        ("termination", np.int32),
        ("num_iterations", np.int32),
        ("num_successful", np.int32),
        ("accept_mask", np.int32),
        ("initial_cost", np.float64),
        ("final_cost", np.float64),
        ("cost_trace", np.float64, (MAX_ITER_TRACE,)),
        ("radius_trace", np.float64, (MAX_ITER_TRACE,)),
    ],
    align=True,
)


class TdFactorBatch(C.Structure):
    """avm_td_factor_batch (include/avm.h): inputs of ProjectionTdFactor, one entry per factor."""
    _fields_ = [("n", C.c_int32)] + [(k, c_dp) for k in ("pose_i", "pose_j", "ex_pose", "inv_depth", "td", "pts_i", "pts_j", "vel_i", "vel_j",
                                                        "td_i", "td_j", "row_i", "row_j")] + [("tr", C.c_double), ("row", C.c_double), ("focal_length", C.c_double)]


def td_factor_batch(arrays: dict, tr: float, row: float, focal_length: float = 460.0):
    """Build an avm_td_factor_batch from host numpy arrays (kept alive on the returned struct)."""
    import numpy as np

    keep = {k: np.ascontiguousarray(np.asarray(v, float)) for k, v in arrays.items()}
    s = TdFactorBatch()
    s.n = keep["inv_depth"].shape[0]
    for k, v in keep.items():
        setattr(s, k, dptr(v))
    s.tr, s.row, s.focal_length = float(tr), float(row), float(focal_length)
    s._keep = keep
    return s


class FselHorizonIn(C.Structure):
    """avm_fsel_horizon_in (include/avm.h): inputs of HorizonGenerator::imu."""
    _fields_ = [
        ("n_problems", C.c_int32),
        ("horizon", C.c_int32),
        ("k_pos", c_dp), ("k_quat", c_dp), ("k_ba", c_dp),
        ("k1_pos", c_dp), ("k1_vel", c_dp), ("k1_quat", c_dp),
        ("acc", c_dp), ("gyr", c_dp),
        ("nr_imu", c_ip), ("delta_imu", c_dp),
    ]


def horizon_in(horizon, k_pos, k_quat, k_ba, k1_pos, k1_vel, k1_quat, acc, gyr, nr_imu, delta_imu):
    """Build an avm_fsel_horizon_in from host numpy arrays (kept alive on the returned struct)."""
    import numpy as np

    f64 = lambda a, n: np.ascontiguousarray(np.asarray(a, float).reshape(-1, n))
    arrs = dict(k_pos=f64(k_pos, 3), k_quat=f64(k_quat, 4), k_ba=f64(k_ba, 3), k1_pos=f64(k1_pos, 3), k1_vel=f64(k1_vel, 3),
                k1_quat=f64(k1_quat, 4), acc=f64(acc, 3), gyr=f64(gyr, 3), delta_imu=np.ascontiguousarray(np.asarray(delta_imu, float).reshape(-1)))
    nr = np.ascontiguousarray(np.asarray(nr_imu, np.int32).reshape(-1))
    s = FselHorizonIn()
    s.n_problems, s.horizon = arrs["k_pos"].shape[0], int(horizon)
    for k, v in arrs.items():
        setattr(s, k, dptr(v))
    s.nr_imu = iptr(nr)
    s._keep = (arrs, nr)
    return s


class FselBatch(C.Structure):
    _fields_ = [
        ("n_problems", C.c_int32),
        ("horizon", C.c_int32),
        ("max_cand", C.c_int32),
        ("max_used", C.c_int32),
        ("max_cloud", C.c_int32),
        ("max_features", C.c_int32),
        ("hor_pos", c_dp),
        ("hor_quat", c_dp),
        ("nr_imu", c_ip),
        ("delta_imu", c_dp),
        ("acc_var", C.c_double),
        ("acc_bias_var", C.c_double),
        ("q_ic", C.c_double * 4),
        ("t_ic", C.c_double * 3),
        ("fx", C.c_double),
        ("fy", C.c_double),
        ("cx", C.c_double),
        ("cy", C.c_double),
        ("k1", C.c_double),
        ("k2", C.c_double),
        ("p1", C.c_double),
        ("p2", C.c_double),
        ("image_width", C.c_int32),
        ("image_height", C.c_int32),
        ("n_cand", c_ip),
        ("cand_id", c_ip),
        ("cand_xy", c_dp),
        ("cand_prob", c_dp),
        ("n_used", c_ip),
        ("used_id", c_ip),
        ("used_xy", c_dp),
        ("n_cloud", c_ip),
        ("cloud_xy", c_dp),
        ("cloud_depth", c_dp),
    ]


class FselOut(C.Structure):
    _fields_ = [("n_selected", c_ip), ("selected_ids", c_ip), ("fvalues", c_dp), ("min_gap", c_dp)]


AVM_ABI_VERSION = 6  # include/avm.h: avm_create() refuses a caller built against another header


class Config(C.Structure):
    _fields_ = [
        ("device", C.c_int32),
        ("max_windows", C.c_int32),
        ("max_problems", C.c_int32),
        ("abi_version", C.c_int32),
        ("reserved", C.c_int32 * 4),
    ]


def default_options() -> Options:
    """Values the reference runs with (estimator.cpp:794-806, euroc_config.yaml:54-63) + Ceres defaults.

    Kept in Python so that host code can build options without loading any native library;
    tests check it against avm_default_options() of the product and the oracle.
    """
    o = Options()
    o.max_num_iterations = 8
    o.estimate_extrinsic = 0
    o.estimate_td = 0
    o.marginalization_flag = MARGIN_OLD
    o.focal_length = 460.0
    o.g[0], o.g[1], o.g[2] = 0.0, 0.0, 9.81007
    o.acc_n, o.gyr_n, o.acc_w, o.gyr_w = 0.08, 0.004, 0.00004, 2.0e-6
    o.cauchy_a = 1.0
    o.max_sum_dt = 10.0
    o.initial_trust_region_radius = 1e4
    o.max_trust_region_radius = 1e16
    o.min_trust_region_radius = 1e-32
    o.min_relative_decrease = 1e-3
    o.function_tolerance = 1e-6
    o.gradient_tolerance = 1e-10
    o.parameter_tolerance = 1e-8
    o.min_lm_diagonal = 1e-6
    o.max_lm_diagonal = 1e32
    o.max_num_consecutive_invalid_steps = 5
    o.jacobi_scaling = 1
    o.marg_eps = 1e-8
    o.tr, o.row = 0.0, 480.0  # global shutter (euroc_config.yaml:66), image_height 480
    o.max_solver_time_s = 0.0  # no wall-clock cap (estimator.cpp:803-806 sets SOLVER_TIME; off for parity and the bench)
    o.marg_noise_rel = 1e-18   # the eigenvalue clamp also tests against the rounding noise of the eigenvector's variables (0: reference-literal)
    return o


def _ptr(a, ctype):
    """Pointer to a numpy array (host) or a torch tensor (host or device)."""
    if a is None:
        return C.cast(None, C.POINTER(ctype))
    if isinstance(a, np.ndarray):
        assert a.flags["C_CONTIGUOUS"]
        return a.ctypes.data_as(C.POINTER(ctype))
    # torch tensor
    assert a.is_contiguous()
    return C.cast(a.data_ptr(), C.POINTER(ctype))


def dptr(a):
    return _ptr(a, C.c_double)


def iptr(a):
    return _ptr(a, C.c_int32)
