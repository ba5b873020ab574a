"""Test helpers: hand-built windows around a single factor, error metrics."""
import importlib

import numpy as np

PKG = "anticipated-vins-mono_amd"
abi = importlib.import_module(PKG + ".abi")
buffers = importlib.import_module(PKG + ".buffers")
synth = importlib.import_module(PKG + ".synth")


def rel(a, b):
    a, b = np.asarray(a, float), np.asarray(b, float)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def blank_windows(B, max_feat=8, max_obs=88, max_samp=20, max_prior=96, max_pblk=16):
    """B windows with identity poses, no features, a trivial (zero-motion) IMU stream and no prior."""
    a = {
        "pose": np.zeros((B, 11, 7)), "speedbias": np.zeros((B, 11, 9)), "ex_pose": np.zeros((B, 7)),
        "inv_depth": np.ones((B, max_feat)), "n_feat": np.zeros(B, np.int32), "feat_start": np.zeros((B, max_feat), np.int32),
        "feat_nobs": np.zeros((B, max_feat), np.int32), "feat_obs_begin": np.zeros((B, max_feat), np.int32),
        "obs_xy": np.zeros((B, max_obs, 2)), "imu_n": np.full((B, 10), max_samp, np.int32), "imu_dt": np.full((B, 10, max_samp), 0.005),
        "imu_acc": np.zeros((B, 10, max_samp + 1, 3)), "imu_gyr": np.zeros((B, 10, max_samp + 1, 3)),
        "imu_lin_ba": np.zeros((B, 10, 3)), "imu_lin_bg": np.zeros((B, 10, 3)), "prior_n": np.zeros(B, np.int32),
        "prior_nblk": np.zeros(B, np.int32), "prior_blk_kind": np.zeros((B, max_pblk), np.int32),
        "prior_blk_frame": np.zeros((B, max_pblk), np.int32), "prior_J": np.zeros((B, max_prior, max_prior)),
        "prior_r": np.zeros((B, max_prior)), "prior_x0": np.zeros((B, max_pblk, 9)),
    }
    a["pose"][:, :, 6] = 1.0
    a["ex_pose"][:, 6] = 1.0
    a["imu_acc"][..., 2] = 9.81007
    dims = dict(n_windows=B, max_feat=max_feat, max_obs=max_obs, max_samp=max_samp, max_prior=max_prior, max_pblk=max_pblk)
    return buffers.WindowArrays(dims, a)


def golden():
    import os

    return np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "factors.npz"))


def projection_windows_from_golden(g):
    """One window per golden projection factor: frame 0 = pose_i, frame 1 = pose_j, one feature with 2 observations."""
    K = g["proj_lam"].shape[0]
    w = blank_windows(K)
    w.a["pose"][:, 0] = g["proj_pose_i"]
    w.a["pose"][:, 1] = g["proj_pose_j"]
    w.a["ex_pose"][:] = g["proj_ex"]
    w.a["inv_depth"][:, 0] = g["proj_lam"]
    w.a["n_feat"][:] = 1
    w.a["feat_nobs"][:, 0] = 2
    w.a["obs_xy"][:, 0] = g["proj_pts_i"][:, :2]
    w.a["obs_xy"][:, 1] = g["proj_pts_j"][:, :2]
    return w


def imu_window_from_golden(g):
    w = blank_windows(1)
    ns = g["imu_dt"].shape[0]
    w.a["imu_n"][0, 0] = ns
    w.a["imu_dt"][0, 0, :ns] = g["imu_dt"]
    w.a["imu_acc"][0, 0, : ns + 1] = g["imu_acc"]
    w.a["imu_gyr"][0, 0, : ns + 1] = g["imu_gyr"]
    w.a["imu_lin_ba"][0, 0] = g["imu_lba"]
    w.a["imu_lin_bg"][0, 0] = g["imu_lbg"]
    w.a["pose"][0, 0] = g["imu_pose_i"]
    w.a["pose"][0, 1] = g["imu_pose_j"]
    w.a["speedbias"][0, 0] = g["imu_sb_i"]
    w.a["speedbias"][0, 1] = g["imu_sb_j"]
    return w


def qmul_xyzw(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by, aw * by + ay * bw + az * bx - ax * bz,
                     aw * bz + az * bw + ax * by - ay * bx, aw * bw - ax * bx - ay * by - az * bz])


def perturb_pose(w, b, f, k, eps):
    """check() convention (projection_factor.cpp:187-201): P += delta, Q = Q * deltaQ(delta) (not renormalised)."""
    w2 = w.copy()
    if k < 3:
        w2.a["pose"][b, f, k] += eps
    else:
        d = np.zeros(3)
        d[k - 3] = eps
        w2.a["pose"][b, f, 3:] = qmul_xyzw(w.a["pose"][b, f, 3:], np.array([d[0] / 2, d[1] / 2, d[2] / 2, 1.0]))
    return w2
