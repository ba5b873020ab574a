"""include/avm_host.hpp - the C++ host side above the C ABI (avm_host::Estimator / avm_host::FeatureSelector with the
reference's member names and call surfaces) - driven from pytest through the ctypes hooks of tests/host_cpp/host_shim.cpp.

CPU tier: the marshalling (the inverse of the test's loader must give back the synthetic tables, features that fail the
filter of estimator.cpp:715 are skipped in place), the selector's bookkeeping on the branches that need no device
(feature_selector.cpp:74-120,172-202), and the loud failure without a GPU.
GPU tier: optimization() / triangulate() / select() through the C++ objects against the CPU oracle.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from helpers import abi, buffers, rel, synth

HERE = os.path.dirname(os.path.abspath(__file__))
CAM_KEYS = ("fx", "fy", "cx", "cy", "k1", "k2", "p1", "p2")


def _shim():
    import torch  # noqa: F401  (its bundled HIP runtime has to be the first one in the process, see lib.py)

    d = os.path.join(HERE, "host_cpp")
    subprocess.check_call(["make", "-C", d, "-s"])
    L = C.CDLL(os.path.join(d, "libavm_host_shim.so"))
    L.hs_create.restype = C.c_void_p
    L.hs_last_error.restype = C.c_char_p
    return L


class Host:
    """One avm_host::Estimator (+ FeatureSelector) behind the shim."""

    def __init__(self, device=0):
        self.L = _shim()
        self.h = C.c_void_p(self.L.hs_create(int(device)))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.hs_destroy(self.h)
            self.h = None

    def err(self):
        return self.L.hs_last_error().decode()

    def load(self, win, w=0, feat_id=None, solve_flag=None, extras=()):
        s = win.struct()
        nf = int(win.a["n_feat"][w])
        fid = np.ascontiguousarray(feat_id if feat_id is not None else np.arange(nf), np.int32)
        sf = np.ascontiguousarray(solve_flag if solve_flag is not None else np.ones(nf), np.int32)
        xb = np.ascontiguousarray([e[0] for e in extras], np.int32)
        xs = np.ascontiguousarray([e[1] for e in extras], np.int32)
        xn = np.ascontiguousarray([e[2] for e in extras], np.int32)
        rc = self.L.hs_load_window(self.h, C.byref(s), int(w), abi.iptr(fid), abi.iptr(sf), len(extras), abi.iptr(xb), abi.iptr(xs), abi.iptr(xn))
        assert rc == 0, self.err()

    def marshal(self, max_samp):
        out = dict(pose=np.zeros((11, 7)), speedbias=np.zeros((11, 9)), ex_pose=np.zeros(7), inv_depth=np.zeros(150), n_feat=np.zeros(1, np.int32),
                   feat_start=np.zeros(150, np.int32), feat_nobs=np.zeros(150, np.int32), feat_obs_begin=np.zeros(150, np.int32),
                   obs_xy=np.zeros((1650, 2)), imu_n=np.zeros(10, np.int32), imu_dt=np.zeros((10, max_samp)), imu_acc=np.zeros((10, max_samp + 1, 3)),
                   imu_gyr=np.zeros((10, max_samp + 1, 3)), imu_lin_ba=np.zeros((10, 3)), imu_lin_bg=np.zeros((10, 3)), feat_id=np.full(150, -1, np.int32))
        order = ("pose", "speedbias", "ex_pose", "inv_depth", "n_feat", "feat_start", "feat_nobs", "feat_obs_begin", "obs_xy", "imu_n", "imu_dt",
                 "imu_acc", "imu_gyr", "imu_lin_ba", "imu_lin_bg", "feat_id")
        args = [abi.iptr(out[k]) if out[k].dtype == np.int32 else abi.dptr(out[k]) for k in order]
        rc = self.L.hs_marshal(self.h, int(max_samp), *args)
        assert rc == 0, self.err()
        return out

    def set_flags(self, solver_flag=1, marginalization_flag=0, max_num_iterations=0):
        self.L.hs_set_flags(self.h, int(solver_flag), int(marginalization_flag), int(max_num_iterations))

    def optimization(self):
        return self.L.hs_optimization(self.h)

    def triangulate(self, init_depth=5.0):
        return self.L.hs_triangulate(self.h, C.c_double(init_depth))

    def slide_window(self, init_depth=5.0):
        return self.L.hs_slide_window(self.h, C.c_double(init_depth))

    def features(self):
        cap = 512
        out = dict(id=np.zeros(cap, np.int32), start=np.zeros(cap, np.int32), nobs=np.zeros(cap, np.int32), depth=np.zeros(cap),
                   first=np.zeros((cap, 2)), last=np.zeros((cap, 2)))
        n = self.L.hs_dump_features(self.h, cap, abi.iptr(out["id"]), abi.iptr(out["start"]), abi.iptr(out["nobs"]), abi.dptr(out["depth"]),
                                    abi.dptr(out["first"]), abi.dptr(out["last"]))
        assert n >= 0
        return {k: v[:n] for k, v in out.items()}

    def imu(self, j, cap=256):
        dt, acc, gyr, lin = np.zeros(cap), np.zeros((cap + 1, 3)), np.zeros((cap + 1, 3)), np.zeros(6)
        n = self.L.hs_dump_imu(self.h, int(j), cap, abi.dptr(dt), abi.dptr(acc), abi.dptr(gyr), abi.dptr(lin))
        assert n >= 0
        return n, dt[:n], acc[:n + 1], gyr[:n + 1], lin

    def state(self):
        out = dict(pose=np.zeros((11, 7)), speedbias=np.zeros((11, 9)), ex_pose=np.zeros(7), depth=np.zeros(150), solve_flag=np.zeros(150, np.int32))
        n = C.c_int32(0)
        summ = buffers.summary_alloc(1)
        self.L.hs_get_state(self.h, abi.dptr(out["pose"]), abi.dptr(out["speedbias"]), abi.dptr(out["ex_pose"]), C.byref(n), abi.dptr(out["depth"]),
                            abi.iptr(out["solve_flag"]), buffers.summary_ptr(summ))
        out["n_feat"], out["summary"] = n.value, summ
        return out

    def set_options(self, o):
        self.L.hs_set_options(self.h, C.byref(o))

    def extras(self):
        v = np.zeros(22)
        self.L.hs_get_extras(self.h, abi.dptr(v))
        return dict(td=v[0], relo_pose=v[1:8], drift_correct_yaw=v[8], drift_correct_t=v[9:12], relo_relative_t=v[12:15], relo_relative_q=v[15:19],
                    relo_relative_yaw=v[19], relocalization_info=bool(v[20]), failure_occur=bool(v[21]))

    def prior(self):
        p = buffers.PriorOutArrays.alloc(1)
        n, nb = C.c_int32(0), C.c_int32(0)
        self.L.hs_get_prior(self.h, C.byref(n), C.byref(nb), abi.iptr(p.a["blk_kind"]), abi.iptr(p.a["blk_frame"]), abi.dptr(p.a["J"]), abi.dptr(p.a["r"]),
                            abi.dptr(p.a["x0"]))
        p.a["n"][0], p.a["nblk"][0] = n.value, nb.value
        return p

    # ---- selector
    def sel_create(self, cam, horizon):
        c = np.array([cam[k] for k in CAM_KEYS], float)
        self.L.hs_sel_create(self.h, abi.dptr(c), int(cam["image_width"]), int(cam["image_height"]), int(horizon))

    def sel_set_parameters(self, accVar, accBiasVar, enable, maxFeatures, initThresh, useGT=False):
        self.L.hs_sel_set_parameters(self.h, C.c_double(accVar), C.c_double(accBiasVar), int(enable), int(maxFeatures), int(initThresh), int(useGT))

    def sel_set_next_state(self, stamp, P, Q, V, a, w, Ba):
        arrs = [np.ascontiguousarray(x, float) for x in (P, Q, V, a, w, Ba)]
        self.L.hs_sel_set_next_state(self.h, C.c_double(stamp), *[abi.dptr(x) for x in arrs])

    def select(self, image, stamp, nrImu):
        """image: {id: 8-vector}. Returns (rc, image_ids_after, tracked, selected)."""
        ids = np.ascontiguousarray(list(image.keys()), np.int32)
        rows = np.ascontiguousarray([image[i] for i in image], float).reshape(-1, 8)
        cap = 8192
        io, tr, se = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        ni, nt = C.c_int32(0), C.c_int32(0)
        rc = self.L.hs_sel_select(self.h, len(ids), abi.iptr(ids), abi.dptr(rows), C.c_double(stamp), int(nrImu), abi.iptr(io), C.byref(ni), abi.iptr(tr),
                                  C.byref(nt), abi.iptr(se), cap)
        if rc < 0:
            return rc, None, None, None
        return rc, io[:ni.value].tolist(), tr[:nt.value].tolist(), se[:rc].tolist()


def _frame(rng, ids):
    """image_t rows for the given ids: x y 1 u v vx vy prob (prob float32-rounded, as it arrives in the PointCloud channel)."""
    cam = synth.CAM
    img = {}
    for i in ids:
        u, v = rng.uniform(0, cam["image_width"]), rng.uniform(0, cam["image_height"])
        p = float(np.float32(rng.uniform(0.05, 1.0)))
        img[int(i)] = [(u - cam["cx"]) / cam["fx"], (v - cam["cy"]) / cam["fy"], 1.0, u, v, 0.0, 0.0, p]
    return img


# ------------------------------------------------------------------------------------------------ CPU tier
def test_marshal_restores_the_tables_and_skips_filtered_features():
    w = synth.make_windows(2, tracks="sparse", n_feat=70, max_feat=150, max_obs=1650, max_samp=24)
    w.a["imu_n"][1, 3] = 17  # ragged sample counts
    H = Host()
    # features that fail used_num >= 2 && start_frame < WINDOW_SIZE - 2 sit between the others in the list
    extras = [(0, 2, 1), (5, 9, 2), (5, 8, 3), (70, 0, 1)]
    fid = 500 + 3 * np.arange(70)
    H.load(w, 1, feat_id=fid, extras=extras)
    m = H.marshal(24)
    nf = int(w.a["n_feat"][1])
    assert m["n_feat"][0] == nf
    assert np.array_equal(m["feat_id"][:nf], fid[:nf])
    for k in ("pose", "speedbias", "ex_pose", "imu_lin_ba", "imu_lin_bg"):
        assert np.array_equal(m[k], w.a[k][1]), k
    for k in ("feat_start", "feat_nobs", "inv_depth"):
        assert rel(m[k][:nf], w.a[k][1, :nf]) < 1e-15, k
    # observations are re-packed contiguously in list order: compare track by track
    for e in range(nf):
        n, b0, b1 = w.a["feat_nobs"][1, e], w.a["feat_obs_begin"][1, e], m["feat_obs_begin"][e]
        assert np.array_equal(m["obs_xy"][b1:b1 + n], w.a["obs_xy"][1, b0:b0 + n])
    assert np.array_equal(m["feat_obs_begin"][:nf], np.concatenate([[0], np.cumsum(w.a["feat_nobs"][1, :nf])[:-1]]))
    assert np.array_equal(m["imu_n"], w.a["imu_n"][1])
    for j in range(10):
        n = w.a["imu_n"][1, j]
        assert np.array_equal(m["imu_dt"][j, :n], w.a["imu_dt"][1, j, :n])
        assert np.array_equal(m["imu_acc"][j, :n + 1], w.a["imu_acc"][1, j, :n + 1])
        assert np.array_equal(m["imu_gyr"][j, :n + 1], w.a["imu_gyr"][1, j, :n + 1])


def test_host_objects_fail_loudly_without_a_device():
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    w = synth.make_windows(1, tracks="sparse", n_feat=20, max_feat=150, max_obs=1650)
    H = Host()
    H.load(w)
    assert H.optimization() == abi.AVM_ERR_NO_DEVICE and "no CPU path" in H.err()
    assert H.triangulate() == abi.AVM_ERR_NO_DEVICE
    before = H.state()
    assert np.array_equal(before["pose"], w.a["pose"][0])  # nothing was touched
    H.sel_create(synth.CAM, 10)
    H.sel_set_parameters(0.08, 0.004, True, 30, 10)
    H.set_flags(solver_flag=1)
    rng = np.random.default_rng(1)
    H.sel_set_next_state(0.1, w.a["pose"][0, 10, :3], w.a["pose"][0, 10, 3:], np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3))
    rc, *_ = H.select(_frame(rng, range(1, 20)), 0.1, 20)
    assert rc == abi.AVM_ERR_NO_DEVICE


def test_selector_bookkeeping_before_initialization():
    """feature_selector.cpp:74-120,172-202 while solver_flag != NON_LINEAR: the first image is taken whole and tracked, later
    images pass the tracked ids, and top up with everything left in `image` while fewer than initThresh are tracked."""
    w = synth.make_windows(1, tracks="sparse", n_feat=20, max_feat=150, max_obs=1650)
    H = Host()
    H.load(w)
    H.set_flags(solver_flag=0)
    H.sel_create(synth.CAM, 10)
    rng = np.random.default_rng(2)

    H.sel_set_parameters(0.08, 0.004, False, 30, 10)
    rc, img, tr, se = H.select(_frame(rng, [3, 5, 9]), 1.0, 20)
    assert (rc, img, tr, se) == (0, [3, 5, 9], [], [])  # disabled: returns {} and leaves image alone

    H.sel_set_parameters(0.08, 0.004, True, 30, 6)
    rc, img, tr, se = H.select(_frame(rng, [3, 5, 9, 12]), 1.0, 20)
    assert (rc, img, tr, se) == (0, [3, 5, 9, 12], [3, 5, 9, 12], [])
    assert H.L.hs_sel_last_feature_id(H.h) == 12
    # second image: 5 was lost, 14 and 20 are new.  Not initialized and not the first image: new ones are dropped,
    # but 3 tracked < initThresh 6, so everything still in `image` (the old ids) is let through
    rc, img, tr, se = H.select(_frame(rng, [3, 9, 12, 14, 20]), 1.1, 20)
    assert (rc, img, tr, se) == (0, [3, 9, 12], [3, 5, 9, 12], [])
    assert H.L.hs_sel_last_feature_id(H.h) == 20
    # an old id that was never tracked (7 < lastFeatureId_) only passes through the initThresh top-up
    rc, img, tr, se = H.select(_frame(rng, [3, 7, 9, 25]), 1.2, 20)
    assert (rc, img, tr, se) == (0, [3, 7, 9], [3, 5, 9, 12], [])
    H.sel_set_parameters(0.08, 0.004, True, 30, 2)
    rc, img, tr, se = H.select(_frame(rng, [3, 7, 9, 30]), 1.3, 20)
    assert (rc, img, tr, se) == (0, [3, 9], [3, 5, 9, 12], [])


def test_python_and_cpp_select_bookkeeping_agree_before_initialization():
    """The batch-oriented Python mirror and the C++ host object walk the same uninitialized branches of select() over a
    random sequence of frames (ids lost, re-found, new; initThresh switched on and off)."""
    import importlib

    fs_m = importlib.import_module("anticipated-vins-mono_amd.feature_selector")
    py = object.__new__(fs_m.FeatureSelector)  # no device: only the bookkeeping runs on this branch
    py.trackedFeatures_, py.lastFeatureId_ = [], 0
    w = synth.make_windows(1, tracks="sparse", n_feat=20, max_feat=150, max_obs=1650)
    H = Host()
    H.load(w)
    H.set_flags(solver_flag=0)
    H.sel_create(synth.CAM, 10)
    rng = np.random.default_rng(8)
    alive, nxt = [], 1
    for k in range(12):
        thresh = [0, 6, 40][k % 3]
        py.setParameters(True, 30, thresh)
        H.sel_set_parameters(0.08, 0.004, True, 30, thresh)
        alive = [i for i in alive if rng.uniform() < 0.7] + list(range(nxt, nxt + int(rng.integers(0, 6))))
        nxt = max(alive + [nxt - 1]) + 1 + int(rng.integers(0, 3))
        img = _frame(rng, alive)
        rc, img_c, tr_c, se_c = H.select(dict(img), 1.0 + 0.1 * k, 20)
        img_p = dict(img)
        tr_p, se_p = py.select(img_p, None, initialized=False)
        assert rc == 0 and (img_c, tr_c, se_c) == (sorted(img_p), tr_p, se_p), k
    assert len(tr_c) > 0


def test_cpp_host_capacity_and_unsupported_errors_are_raised_before_any_device_work():
    """More features than the device tables hold (150 / 1650 observations) surface as avm_host::Error with the ABI's status
    codes - on a box without a GPU too, i.e. before the device is touched."""
    w = synth.make_windows(1, tracks="dense", n_feat=150, max_feat=150, max_obs=1650)
    H = Host()
    H.load(w, extras=[(150, 0, 3)])  # a 151st feature that passes the filter of estimator.cpp:715
    assert H.optimization() == abi.AVM_ERR_CAPACITY and "150 features" in H.err()
    assert H.triangulate() == abi.AVM_ERR_CAPACITY
    # ESTIMATE_TD / ESTIMATE_EXTRINSIC are built: without a GPU the call now gets as far as opening the device and fails there
    H.load(w)
    o = abi.default_options()
    o.estimate_td = 1
    H.L.hs_set_options(H.h, C.byref(o))
    import torch
    if not torch.cuda.is_available():
        assert H.optimization() == abi.AVM_ERR_NO_DEVICE


def test_cpp_selector_ground_truth_horizon_moves_its_cursor_on_every_call(oracle):
    """useGT: the reference builds the horizon on every select() call, initialized or not (feature_selector.cpp:131), and
    HorizonGenerator::groundTruth moves its seek cursor each time.  Host-only work: runs without a GPU.  The cursor of the
    C++ object follows the oracle's GroundTruth driven with the same state_k_ / deltaF sequence."""
    w = synth.make_windows(1, tracks="sparse", n_feat=20, max_feat=150, max_obs=1650)
    n = 400
    t = 1403636580.0 + 0.005 * np.arange(n)
    rows = np.zeros((n, 17))
    rows[:, 0] = t * 1e9
    rows[:, 1:4] = np.stack([np.sin(0.3 * (t - t[0])), np.cos(0.2 * (t - t[0])), 0.1 * (t - t[0])], 1)
    ang = 0.4 * (t - t[0])
    rows[:, 4], rows[:, 7] = np.cos(ang / 2), np.sin(ang / 2)  # w x y z
    H = Host()
    H.load(w)
    H.set_flags(solver_flag=0)
    H.sel_create(synth.CAM, 5)
    H.sel_set_parameters(0.08, 0.004, True, 30, 0, useGT=True)
    assert H.L.hs_sel_set_ground_truth(H.h, abi.dptr(np.ascontiguousarray(rows)), n) == 0, H.err()
    gt = oracle.GroundTruth(rows)
    rng = np.random.default_rng(4)
    pose10 = w.a["pose"][0, 10]
    prev_stamp, frame_time = 0.0, None
    for k in range(4):
        stamp = t[0] + 0.1 * (k + 1)
        H.sel_set_next_state(stamp, pose10[:3], pose10[3:], np.zeros(3), np.zeros(3), np.zeros(3), np.zeros(3))
        rc, *_ = H.select(_frame(rng, range(1 + 5 * k, 6 + 5 * k)), stamp, 20)
        assert rc == 0, H.err()
        frame_time = stamp if frame_time is None else frame_time
        gt.horizon(5, prev_stamp, pose10[:3], pose10[3:], stamp - frame_time)   # state_k_ carries the previous image's stamp
        assert H.L.hs_sel_gt_seek(H.h) == gt.seek_idx, k
        prev_stamp, frame_time = stamp, stamp
    assert gt.seek_idx > 0


# ------------------------------------------------------------------------------------------------ GPU tier
def _install_prior(win, p):
    for k_w, k_p in (("prior_n", "n"), ("prior_nblk", "nblk"), ("prior_blk_kind", "blk_kind"), ("prior_blk_frame", "blk_frame"), ("prior_J", "J"),
                     ("prior_r", "r"), ("prior_x0", "x0")):
        win.a[k_w][:] = p.a[k_p]


@pytest.mark.gpu
@pytest.mark.parametrize("flag", [0, 1])
def test_cpp_estimator_optimization_matches_oracle_and_chains_the_prior(oracle, flag):
    w = synth.make_windows(1, tracks="sparse", n_feat=60, max_feat=150, max_obs=1650)
    H = Host()
    H.load(w)
    H.set_flags(solver_flag=1, marginalization_flag=flag)
    assert H.optimization() == 0, H.err()
    o = abi.default_options()
    o.marginalization_flag = abi.MARGIN_OLD if flag == 0 else abi.MARGIN_SECOND_NEW
    wo, po, so = w.copy(), buffers.PriorOutArrays.alloc(1), buffers.summary_alloc(1)
    oracle.window_solve(o, wo, po, so)
    s = H.state()
    nf = int(w.a["n_feat"][0])
    assert s["n_feat"] == nf
    assert rel(s["pose"], wo.a["pose"][0]) < 1e-6 and rel(s["speedbias"], wo.a["speedbias"][0]) < 1e-6
    assert rel(1.0 / s["depth"][:nf], wo.a["inv_depth"][0, :nf]) < 1e-6
    assert np.array_equal(s["solve_flag"][:nf], np.where(wo.a["inv_depth"][0, :nf] < 0, 2, 1))
    assert s["summary"]["accept_mask"][0] == so["accept_mask"][0] and s["summary"]["num_iterations"][0] == so["num_iterations"][0]
    pg = H.prior()
    assert pg.a["n"][0] == po.a["n"][0] and pg.a["nblk"][0] == po.a["nblk"][0]
    nb = int(po.a["nblk"][0])
    assert np.array_equal(pg.a["blk_kind"][0, :nb], po.a["blk_kind"][0, :nb]) and np.array_equal(pg.a["blk_frame"][0, :nb], po.a["blk_frame"][0, :nb])
    # second optimization() on the same members: the prior left behind by the first one is the one that is used
    assert H.optimization() == 0, H.err()
    _install_prior(wo, po)
    oracle.window_solve(o, wo, buffers.PriorOutArrays.alloc(1), so)
    s2 = H.state()
    assert rel(s2["pose"], wo.a["pose"][0]) < 1e-5 and rel(s2["speedbias"], wo.a["speedbias"][0]) < 1e-5


@pytest.mark.gpu
def test_cpp_estimator_optimization_with_extrinsic_td_relocalization_and_failure_anchor(oracle):
    """ESTIMATE_EXTRINSIC, ESTIMATE_TD, relocalization_info and failure_occur through the C++ object: the members come back like
    the reference leaves them (estimator.cpp:521-604): td, tic / ric, relo_Pose, the pose-graph outputs, both flags cleared."""
    w = synth.make_windows(1, first_id=33, tracks="sparse", n_feat=90, max_feat=150, max_obs=1650, td_true=0.01, relo=True)
    anchor = w.a["pose"][:, 0].copy()
    anchor[:, :3] += np.array([0.5, 0.25, -0.125])
    w.a["failure_occur"], w.a["last_pose0"] = np.array([1], np.int32), anchor
    w.a["ex_pose"][:, :3] += 0.01
    o = abi.default_options()
    o.estimate_extrinsic, o.estimate_td = 1, 1
    H = Host()
    H.load(w, feat_id=1000 + 3 * np.arange(int(w.a["n_feat"][0])))
    H.set_options(o)
    H.set_flags(solver_flag=1, marginalization_flag=0)
    assert H.optimization() == 0, H.err()
    wo, po, so = w.copy(), buffers.PriorOutArrays.alloc(1), buffers.summary_alloc(1)
    oracle.window_solve(o, wo, po, so)
    s, x = H.state(), H.extras()
    assert s["summary"]["accept_mask"][0] == so["accept_mask"][0]
    assert rel(s["pose"], wo.a["pose"][0]) < 1e-6 and rel(s["speedbias"], wo.a["speedbias"][0]) < 1e-6 and rel(s["ex_pose"], wo.a["ex_pose"][0]) < 1e-6
    assert abs(x["td"] - wo.a["td"][0]) < 1e-6 * abs(wo.a["td"][0]) and abs(x["td"]) > 1e-5
    assert rel(x["relo_pose"], wo.a["relo_pose"][0]) < 1e-6
    assert not x["relocalization_info"] and not x["failure_occur"]
    assert np.abs(s["pose"][0, :3] - anchor[0, :3]).max() < 1e-12                       # re-anchored on last_P0
    # the pose-graph outputs, restated with numpy from the oracle's relo_r / relo_t and Ps / Rs (estimator.cpp:597-604)
    rp, r = wo.a["relo_pose"][0], int(w.a["relo_frame"][0])
    Rr, Ri = synth.R_from_quat(rp[3:]), synth.R_from_quat(wo.a["pose"][0, r, 3:])
    yaw = lambda R: np.degrees(np.arctan2(R[1, 0], R[0, 0]))
    prev_t = w.a["relo_pose"][0, :3] + np.array([0.3, -0.2, 0.1])
    prev_R = synth.R_from_quat(np.array([0.0, 0.0, np.sin(0.2), np.cos(0.2)]))
    dyaw = yaw(prev_R) - yaw(Rr)
    cy, sy = np.cos(np.radians(dyaw)), np.sin(np.radians(dyaw))
    assert abs(x["drift_correct_yaw"] - dyaw) < 1e-6
    assert rel(x["drift_correct_t"], prev_t - np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]]) @ rp[:3]) < 1e-6
    assert rel(x["relo_relative_t"], Rr.T @ (wo.a["pose"][0, r, :3] - rp[:3])) < 1e-5
    assert rel(synth.R_from_quat(x["relo_relative_q"]), Rr.T @ Ri) < 1e-6
    assert abs(x["relo_relative_yaw"] - (yaw(Ri) - yaw(Rr))) < 1e-6
    # the new prior carries td (n = 76) and chains into the next optimization()
    pg = H.prior()
    assert pg.a["n"][0] == po.a["n"][0] == 76 and (pg.a["blk_kind"][0, :13] == po.a["blk_kind"][0, :13]).all() and pg.a["blk_kind"][0, 12] == abi.BLK_TD
    assert H.optimization() == 0, H.err()


@pytest.mark.gpu
def test_cpp_estimator_triangulate_matches_oracle(oracle):
    w = synth.make_windows(1, tracks="sparse", n_feat=50, max_feat=150, max_obs=1650)
    nf = int(w.a["n_feat"][0])
    keep = w.a["inv_depth"][0].copy()
    w.a["inv_depth"][0, :nf:2] = -1.0  # estimated_depth = -1: not triangulated yet
    H = Host()
    H.load(w)
    assert H.triangulate(5.0) == 0, H.err()
    wo = w.copy()
    oracle.triangulate(wo, init_depth=5.0)
    s = H.state()
    assert rel(1.0 / s["depth"][:nf], wo.a["inv_depth"][0, :nf]) < 1e-9
    assert np.array_equal(1.0 / s["depth"][1:nf:2], 1.0 / (1.0 / keep[1:nf:2]))  # features with a depth keep it


@pytest.mark.gpu
@pytest.mark.parametrize("horizon", [10, 5])
def test_cpp_selector_select_over_frames_matches_oracle_composition(oracle, horizon):
    """select() frame after frame through the C++ object against the same chain put together from the oracle's pieces
    (HorizonGenerator::imu, initKDTree's cloud, the greedy) with the bookkeeping restated in Python."""
    w = synth.make_windows(1, tracks="sparse", n_feat=80, max_feat=150, max_obs=1650)
    nf = int(w.a["n_feat"][0])
    flags = np.ones(nf, np.int32)
    flags[::7] = 2  # some features failed in the last solve: they do not enter the depth cloud
    H = Host()
    H.load(w, solve_flag=flags)
    cam = synth.CAM
    H.sel_create(cam, horizon)
    maxF, accVar, biasVar = 60, synth.ACC_N, synth.ACC_W
    H.sel_set_parameters(accVar, biasVar, True, maxF, 10)
    rng = np.random.default_rng(5)
    pose10, sb10 = w.a["pose"][0, 10], w.a["speedbias"][0, 10]

    tracked, last_id, first, n_selected = [], 0, True, 0
    py = __import__("importlib").import_module("anticipated-vins-mono_amd.feature_selector").FeatureSelector()
    py.setParameters(True, maxF, 10)
    next_id = 1
    stamp, t_prev = 10.0, None
    alive = []
    for k in range(5):
        init = k >= 1
        H.set_flags(solver_flag=int(init))
        n_new = [30, 60, 40, 0, 50][k]
        alive = [i for i in alive if rng.uniform() < 0.8] + list(range(next_id, next_id + n_new))
        next_id += n_new
        image = _frame(rng, alive)
        P = pose10[:3] + rng.normal(0, 0.05, 3)
        Q = pose10[3:] + rng.normal(0, 0.01, 4)
        Q /= np.linalg.norm(Q)
        V, a, gy = sb10[:3] + rng.normal(0, 0.05, 3), rng.normal(0, 0.5, 3) + [0, 0, 9.8], rng.normal(0, 0.1, 3)
        H.sel_set_next_state(stamp, P, Q, V, a, gy, sb10[3:6])
        rc, img_g, tr_g, se_g = H.select(dict(image), stamp, 20)
        assert rc >= 0, H.err()

        # ---- the same frame with the oracle's pieces
        if t_prev is None:
            t_prev = stamp
        deltaF = stamp - t_prev
        new = sorted(i for i in image if i > last_id)
        old = sorted(i for i in image if i <= last_id)
        if new:
            last_id = new[-1]
        subset = [f for f in sorted(set(tracked)) if f in old]
        selected = []
        def build_problem(new, subset):
            hp, hq = oracle.fsel_horizon_imu(horizon, pose10[None, :3], pose10[None, 3:], sb10[None, 3:6], P[None], V[None], Q[None], a[None], gy[None],
                                             np.array([20], np.int32), np.array([deltaF / 20]))
            wc = w.copy()
            wc.a["inv_depth"][0, :nf] = np.where(flags == 1, w.a["inv_depth"][0, :nf], -1.0)
            ncl, cxy, cdep = oracle.fsel_build_cloud(wc, P[None], Q[None], 150)
            nc, nu = max(len(new), 1), max(len(subset), 1)
            arr = {
                "hor_pos": hp, "hor_quat": hq, "nr_imu": np.array([20], np.int32), "delta_imu": np.array([deltaF / 20]),
                "n_cand": np.array([len(new)], np.int32), "cand_id": np.zeros((1, nc), np.int32), "cand_xy": np.zeros((1, nc, 2)),
                "cand_prob": np.zeros((1, nc)), "n_used": np.array([len(subset)], np.int32), "used_id": np.zeros((1, nu), np.int32),
                "used_xy": np.zeros((1, nu, 2)), "n_cloud": ncl.astype(np.int32), "cloud_xy": cxy, "cloud_depth": cdep,
            }
            for j, i in enumerate(new):
                arr["cand_id"][0, j], arr["cand_xy"][0, j], arr["cand_prob"][0, j] = i, image[i][:2], image[i][7]
            for j, i in enumerate(subset):
                arr["used_id"][0, j], arr["used_xy"][0, j] = i, image[i][:2]
            dims = dict(n_problems=1, horizon=horizon, max_cand=nc, max_used=nu, max_cloud=150, max_features=maxF)
            ex = w.a["ex_pose"][0]
            sc = dict(acc_var=accVar, acc_bias_var=biasVar, q_ic=ex[3:], t_ic=ex[:3], **cam)
            return buffers.FselArrays(dims, arr, sc)

        # the batch-oriented Python mirror walks the same frame (its problem comes from the oracle's horizon and cloud)
        img_p = dict(image)
        tr_p, se_p = py.select(img_p, build_problem, initialized=init)
        assert (sorted(img_p), tr_p, se_p) == (img_g, tr_g, se_g), k
        if init:
            pr = build_problem(new, subset)
            out = buffers.FselOutArrays.alloc(1, maxF)
            oracle.fsel_select(pr, out)
            selected = out.a["selected_ids"][0, :out.a["n_selected"][0]].tolist()
            image_out = sorted(subset + selected)
        else:
            if first:
                subset, first = new, False
                tracked = tracked + subset
            if len(subset) < 10:
                subset = sorted(set(subset) | set(old))
            image_out = sorted(subset)
        tracked = tracked + selected
        n_selected += len(selected)
        t_prev = stamp
        assert se_g == selected, (k, se_g, selected)
        assert img_g == image_out, k
        assert tr_g == tracked, k
        stamp += 0.1
    assert n_selected >= 40, n_selected


def _insert_feature_rows(w, rows):
    """rows: (row index, start_frame, nobs) - features that fail the solve's filter, spliced into window 0's tables."""
    a = w.a
    for pos, start, nobs in sorted(rows):
        nf = int(a["n_feat"][0])
        ob = int((a["feat_obs_begin"][0, :nf] + a["feat_nobs"][0, :nf]).max()) if nf else 0
        for k in ("feat_start", "feat_nobs", "feat_obs_begin", "inv_depth"):
            a[k][0, pos + 1:nf + 1] = a[k][0, pos:nf].copy()
        a["feat_start"][0, pos], a["feat_nobs"][0, pos], a["feat_obs_begin"][0, pos], a["inv_depth"][0, pos] = start, nobs, ob, 1.0 / 3.0
        a["obs_xy"][0, ob:ob + nobs] = 0.01 * (1 + np.arange(2 * nobs).reshape(nobs, 2)) + 0.001 * pos
        a["n_feat"][0] = nf + 1


@pytest.mark.gpu
@pytest.mark.parametrize("flag", [0, 1])
def test_cpp_estimator_slide_window_matches_oracle(oracle, flag):
    """slideWindow() on the C++ object - every feature of f_manager goes through the roll, also the ones the solve never
    sees (one observation, late start) - against the oracle's list-based restatement on the same tables."""
    w = synth.make_windows(1, first_id=3, tracks="sparse", n_feat=60, max_feat=150, max_obs=1650, max_samp=48)
    w.a["imu_n"][0, 8], w.a["imu_n"][0, 9] = 20, 17
    nf0 = int(w.a["n_feat"][0])
    n0 = int((w.a["feat_start"][0, :nf0] == 0).sum())
    _insert_feature_rows(w, [(0, 0, 1), (n0 + 1, 0, 2), (nf0 + 2, 9, 1), (nf0 + 3, 9, 2), (nf0 + 4, 10, 1)])
    nf = int(w.a["n_feat"][0])
    assert nf == nf0 + 5 and np.all(np.diff(w.a["feat_start"][0, :nf]) >= 0)
    H = Host()
    H.load(w, feat_id=100 + np.arange(nf))
    H.set_flags(solver_flag=1, marginalization_flag=flag)
    assert H.slide_window(5.0) == 0, H.err()
    wo = w.copy()
    oracle.slide_window(wo, abi.MARGIN_OLD if flag == 0 else abi.MARGIN_SECOND_NEW, True, 5.0)
    s, f = H.state(), H.features()
    assert np.array_equal(s["pose"], wo.a["pose"][0]) and np.array_equal(s["speedbias"], wo.a["speedbias"][0])
    n = int(wo.a["n_feat"][0])
    assert len(f["id"]) == n and n < nf
    st, no, ob = wo.a["feat_start"][0, :n], wo.a["feat_nobs"][0, :n], wo.a["feat_obs_begin"][0, :n]
    assert np.array_equal(f["start"], st) and np.array_equal(f["nobs"], no)
    assert rel(1.0 / f["depth"], wo.a["inv_depth"][0, :n]) < 1e-12
    assert np.array_equal(f["first"], wo.a["obs_xy"][0, ob]) and np.array_equal(f["last"], wo.a["obs_xy"][0, ob + no - 1])
    assert np.all(np.diff(f["id"]) > 0)  # list order survives
    for j in range(10):
        k, dt, acc, gyr, lin = H.imu(j)
        assert k == wo.a["imu_n"][0, j]
        assert np.array_equal(dt, wo.a["imu_dt"][0, j, :k])
        assert np.array_equal(acc, wo.a["imu_acc"][0, j, :k + 1]) and np.array_equal(gyr, wo.a["imu_gyr"][0, j, :k + 1])
        assert np.array_equal(lin[:3], wo.a["imu_lin_ba"][0, j]) and np.array_equal(lin[3:], wo.a["imu_lin_bg"][0, j])
    # and the rolled members solve: optimization() right after, against the oracle on the rolled tables
    H.set_flags(solver_flag=1, marginalization_flag=0)
    assert H.optimization() == 0, H.err()


@pytest.mark.gpu
def test_cpp_estimator_newest_frame_dead_reckoning_matches_oracle(oracle):
    w = synth.make_windows(1, first_id=9, tracks="sparse", n_feat=30, max_feat=150, max_obs=1650)
    w.a["pose"][0, 10] = w.a["pose"][0, 9]
    w.a["speedbias"][0, 10] = w.a["speedbias"][0, 9]
    H = Host()
    H.load(w)
    assert H.L.hs_propagate(H.h) == 0, H.err()
    wo = w.copy()
    o = abi.default_options()
    oracle.imu_propagate(wo, [o.g[0], o.g[1], o.g[2]])
    s = H.state()
    assert rel(s["pose"][10], wo.a["pose"][0, 10]) < 1e-12 and rel(s["speedbias"][10], wo.a["speedbias"][0, 10]) < 1e-12
    assert np.array_equal(s["pose"][:10], w.a["pose"][0, :10])
    assert np.abs(s["pose"][10] - w.a["pose"][0, 9]).max() > 1e-3


@pytest.mark.gpu
def test_cpp_estimator_solve_roll_solve_chain_matches_oracle(oracle):
    """optimization() -> slideWindow() -> optimization() on one C++ object (the prior of the first solve feeds the second,
    the tables are re-marshalled from the rolled members) against the same chain on the oracle."""
    w = synth.make_windows(1, first_id=21, tracks="sparse", n_feat=70, max_feat=150, max_obs=1650)
    H = Host()
    H.load(w)
    H.set_flags(solver_flag=1, marginalization_flag=0)
    assert H.optimization() == 0, H.err()
    assert H.slide_window(5.0) == 0, H.err()
    assert H.optimization() == 0, H.err()
    o = abi.default_options()
    wo, po, so = w.copy(), buffers.PriorOutArrays.alloc(1), buffers.summary_alloc(1)
    oracle.window_solve(o, wo, po, so)
    oracle.slide_window(wo, abi.MARGIN_OLD, True, 5.0)
    _install_prior(wo, po)
    oracle.window_solve(o, wo, buffers.PriorOutArrays.alloc(1), so)
    s = H.state()
    n = int(wo.a["n_feat"][0])
    assert s["n_feat"] == n
    assert rel(s["pose"], wo.a["pose"][0]) < 1e-5 and rel(s["speedbias"], wo.a["speedbias"][0]) < 1e-5
    assert rel(1.0 / s["depth"][:n], wo.a["inv_depth"][0, :n]) < 1e-5
    assert s["summary"]["num_iterations"][0] == so["num_iterations"][0]
