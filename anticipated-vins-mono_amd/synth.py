"""Seeded synthetic EuRoC-shaped inputs for both hot paths.

There is no dataset access in this environment; these generators produce inputs with the
shape and statistics of the reference's operating point (config/euroc/euroc_config.yaml):
11 keyframes @ 10 Hz, 200 Hz IMU (20 samples / interval), 150 landmarks, a 75-dim prior,
and for the selector 500 candidates over a 752x480 pinhole image with a 10-frame horizon.
Window w is generated from seed 0xA17C0000 + w, selector problem p from 0xF5E10000 + p, so
a shard of windows is identical no matter which rank generates it.
"""
import numpy as np

from . import abi
from .buffers import FselArrays, WindowArrays

SEED_WINDOW = 0xA17C0000
SEED_FSEL = 0xF5E10000

# config/euroc/euroc_config.yaml:30-42
RIC = np.array(
    [
        [0.0148655429818, -0.999880929698, 0.00414029679422],
        [0.999557249008, 0.0149672133247, 0.025715529948],
        [-0.0257744366974, 0.00375618835797, 0.999660727178],
    ]
)
TIC = np.array([-0.0216401454975, -0.064676986768, 0.00981073058949])
G_NORM = 9.81007
ACC_N, GYR_N, ACC_W, GYR_W = 0.08, 0.004, 0.00004, 2.0e-6
CAM = dict(fx=461.6, fy=460.3, cx=363.0, cy=248.1, k1=-0.2917, k2=0.08228, p1=5.333e-05, p2=-1.578e-04,
           image_width=752, image_height=480)


def _rot_zyx(y, p, r):
    cy, sy, cp, sp, cr, sr = np.cos(y), np.sin(y), np.cos(p), np.sin(p), np.cos(r), np.sin(r)
    return np.array(
        [
            [cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr],
            [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
            [-sp, cp * sr, cp * cr],
        ]
    )


def quat_from_R(R):
    """(x,y,z,w), w >= 0."""
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0) * 2
        q = np.array([(R[2, 1] - R[1, 2]) / s, (R[0, 2] - R[2, 0]) / s, (R[1, 0] - R[0, 1]) / s, 0.25 * s])
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0) * 2
        v = np.zeros(3)
        v[i] = 0.25 * s
        v[j] = (R[j, i] + R[i, j]) / s
        v[k] = (R[k, i] + R[i, k]) / s
        q = np.array([v[0], v[1], v[2], (R[k, j] - R[j, k]) / s])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def R_from_quat(q):
    x, y, z, w = q
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)],
        ]
    )


def _expm_so3(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th**2 * K @ K


class _Trajectory:
    """Smooth analytic trajectory: position = line + sinusoids, attitude = ZYX Euler sinusoids."""

    def __init__(self, rng):
        speed = rng.uniform(0.5, 2.0)
        hd = rng.uniform(0, 2 * np.pi)
        self.p0 = rng.normal(0, 2.0, 3)
        self.v0 = np.array([speed * np.cos(hd), speed * np.sin(hd), rng.normal(0, 0.1)])
        self.A = rng.uniform(0.05, 0.4, (3, 2)) * np.array([[1.0], [1.0], [0.5]])
        self.w = rng.uniform(0.8, 3.0, (3, 2))
        self.ph = rng.uniform(0, 2 * np.pi, (3, 2))
        self.yaw0 = hd + rng.normal(0, 0.3)
        self.yaw_rate = rng.uniform(-0.5, 0.5)
        self.eA = rng.uniform(0.02, 0.12, (3,))  # yaw wobble, pitch, roll amplitudes
        self.ew = rng.uniform(0.8, 2.5, (3,))
        self.eph = rng.uniform(0, 2 * np.pi, (3,))

    def pos(self, t):
        return self.p0 + self.v0 * t + (self.A * np.sin(self.w * t + self.ph)).sum(1)

    def vel(self, t):
        return self.v0 + (self.A * self.w * np.cos(self.w * t + self.ph)).sum(1)

    def acc(self, t):
        return -(self.A * self.w**2 * np.sin(self.w * t + self.ph)).sum(1)

    def euler(self, t):
        s = self.eA * np.sin(self.ew * t + self.eph)
        return np.array([self.yaw0 + self.yaw_rate * t + s[0], s[1], s[2]])

    def euler_rate(self, t):
        c = self.eA * self.ew * np.cos(self.ew * t + self.eph)
        return np.array([self.yaw_rate + c[0], c[1], c[2]])

    def R(self, t):
        y, p, r = self.euler(t)
        return _rot_zyx(y, p, r)

    def omega_body(self, t):
        y, p, r = self.euler(t)
        yd, pd, rd = self.euler_rate(t)
        return np.array(
            [rd - yd * np.sin(p), pd * np.cos(r) + yd * np.sin(r) * np.cos(p), -pd * np.sin(r) + yd * np.cos(r) * np.cos(p)]
        )


def _one_window(wid, tracks, n_feat, with_prior, max_feat, max_obs, max_samp, max_prior, max_pblk, out, b, td_true=None, relo=False):
    rng = np.random.Generator(np.random.Philox(key=SEED_WINDOW + wid))
    world_pts = []
    tr = _Trajectory(rng)
    G = np.array([0, 0, G_NORM])
    nF, ns, dt = abi.NFRAMES, 20, 0.005
    ba = rng.normal(0, 0.02, 3)
    bg = rng.normal(0, 0.002, 3)
    tk = np.arange(nF) * 0.1
    Rw = [tr.R(t) for t in tk]
    Pw = [tr.pos(t) for t in tk]
    Vw = [tr.vel(t) for t in tk]
    # ---- IMU
    for j in range(abi.WINDOW_SIZE):
        out["imu_n"][b, j] = ns
        out["imu_dt"][b, j, :ns] = dt
        for s in range(ns + 1):
            t = tk[j] + s * dt
            R = tr.R(t)
            out["imu_acc"][b, j, s] = R.T @ (tr.acc(t) + G) + ba + rng.normal(0, ACC_N, 3)
            out["imu_gyr"][b, j, s] = tr.omega_body(t) + bg + rng.normal(0, GYR_N, 3)
    # ---- landmarks and tracks
    if tracks == "dense":
        starts = np.zeros(n_feat, np.int64)
        lens = np.full(n_feat, nF, np.int64)
    else:
        starts = np.sort(rng.integers(0, abi.WINDOW_SIZE - 2, n_feat))  # start_frame < WINDOW_SIZE-2
        lens = np.array([rng.integers(2, nF - s + 1) for s in starts])
    out["n_feat"][b] = n_feat
    o = 0
    sig_px = 1.5 / 460.0
    for e in range(n_feat):
        s, L = int(starts[e]), int(lens[e])
        for _ in range(50):
            xy = np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.45, 0.45)])
            depth = rng.uniform(2.0, 15.0)
            pc = np.array([xy[0], xy[1], 1.0]) * depth
            pw = Rw[s] @ (RIC @ pc + TIC) + Pw[s]
            obs, ok = [], True
            for f in range(s, s + L):
                pcf = RIC.T @ (Rw[f].T @ (pw - Pw[f]) - TIC)
                if pcf[2] < 0.5:
                    ok = False
                    break
                obs.append(pcf[:2] / pcf[2])
            if ok:
                break
        obs = np.array(obs) + rng.normal(0, sig_px, (L, 2))
        outl = rng.uniform(size=L) < 0.05
        outl[0] = False
        obs[outl] += rng.uniform(-0.2, 0.2, (int(outl.sum()), 2))
        out["feat_start"][b, e] = s
        out["feat_nobs"][b, e] = L
        out["feat_obs_begin"][b, e] = o
        out["obs_xy"][b, o : o + L] = obs
        out["inv_depth"][b, e] = (1.0 / depth) * (1.0 + rng.normal(0, 0.10))
        world_pts.append(pw)
        o += L
    assert o <= max_obs
    # ---- initial state = ground truth + perturbation
    ba0 = ba + rng.normal(0, 0.01, 3)
    bg0 = bg + rng.normal(0, 0.001, 3)
    for f in range(nF):
        Rp = Rw[f] @ _expm_so3(rng.normal(0, np.deg2rad(1.0), 3))
        out["pose"][b, f, :3] = Pw[f] + rng.normal(0, 0.05, 3)
        out["pose"][b, f, 3:] = quat_from_R(Rp)
        out["speedbias"][b, f, :3] = Vw[f] + rng.normal(0, 0.05, 3)
        out["speedbias"][b, f, 3:6] = ba0  # biases are copied forward frame to frame (estimator.cpp:104-108)
        out["speedbias"][b, f, 6:9] = bg0
    for j in range(abi.WINDOW_SIZE):
        out["imu_lin_ba"][b, j] = out["speedbias"][b, j, 3:6]
        out["imu_lin_bg"][b, j] = out["speedbias"][b, j, 6:9]
    out["ex_pose"][b, :3] = TIC
    out["ex_pose"][b, 3:] = quat_from_R(RIC)
    # ---- prior over pose[0..9], speedbias[0], ex_pose (what MARGIN_OLD leaves, estimator.cpp:904-916)
    if with_prior:
        kinds = [abi.BLK_POSE] * 10 + [abi.BLK_SPEEDBIAS, abi.BLK_EXPOSE]
        frames = list(range(10)) + [0, 0]
        n = 10 * 6 + 9 + 6
        wts = np.concatenate([np.tile([10.0] * 3 + [50.0] * 3, 10), [10.0] * 3 + [20.0] * 3 + [200.0] * 3, [100.0] * 3 + [200.0] * 3])
        J = (np.eye(n) + 0.05 * rng.normal(size=(n, n))) * wts[None, :]
        out["prior_n"][b] = n
        out["prior_nblk"][b] = len(kinds)
        out["prior_blk_kind"][b, : len(kinds)] = kinds
        out["prior_blk_frame"][b, : len(kinds)] = frames
        out["prior_J"][b, :n, :n] = J
        out["prior_r"][b, :n] = rng.normal(0, 0.3, n)
        for k, (kd, fr) in enumerate(zip(kinds, frames)):
            if kd == abi.BLK_POSE:
                x0 = out["pose"][b, fr].copy()
                x0[:3] += rng.normal(0, 0.01, 3)
                x0[3:] = quat_from_R(R_from_quat(x0[3:]) @ _expm_so3(rng.normal(0, 0.002, 3)))
                out["prior_x0"][b, k, :7] = x0
            elif kd == abi.BLK_SPEEDBIAS:
                out["prior_x0"][b, k, :9] = out["speedbias"][b, fr] + rng.normal(0, 0.005, 9) * np.array([1] * 3 + [0.2] * 3 + [0.02] * 3)
            else:
                out["prior_x0"][b, k, :7] = out["ex_pose"][b]
    # ---- optional members (their own random stream: the tables above do not depend on them)
    rx = np.random.Generator(np.random.Philox(key=SEED_WINDOW + 0x04000000 + wid))

    def project(pw, t):
        pc = RIC.T @ (tr.R(t).T @ (pw - tr.pos(t)) - TIC)
        return pc[:2] / pc[2]

    if td_true is not None:
        # camera-IMU time offset (estimator.cpp:732-747): every observation carries the image velocity of its feature, the
        # td its frame was stamped with (0 here) and its image row; the image was really taken td_true later than stamped,
        # so the raw observation is the projection moved along the velocity by td_true
        for e in range(n_feat):
            s0, L, o0 = int(out["feat_start"][b, e]), int(out["feat_nobs"][b, e]), int(out["feat_obs_begin"][b, e])
            for i in range(L):
                t = tk[s0 + i]
                vel = (project(world_pts[e], t + 1e-3) - project(world_pts[e], t - 1e-3)) / 2e-3
                out["obs_xy"][b, o0 + i] += td_true * vel
                out["obs_vel_td"][b, o0 + i] = [vel[0], vel[1], 0.0, CAM["fy"] * out["obs_xy"][b, o0 + i, 1] + CAM["cy"]]
        out["td"][b] = 0.0
    if relo:
        # relocalization (estimator.cpp:760-792, setReloFrame :1120-1141): an old keyframe that saw the same landmarks from a
        # pose near frame r; relo_Pose starts at the window's frame r
        r = int(rx.integers(3, 8))
        Rl = Rw[r] @ _expm_so3(rx.normal(0, np.deg2rad(4.0), 3))
        Pl = Pw[r] + rx.normal(0, 0.25, 3)
        k = 0
        for e in range(n_feat):
            if int(out["feat_start"][b, e]) > r or rx.uniform() < 0.4:
                continue
            pc = RIC.T @ (Rl.T @ (world_pts[e] - Pl) - TIC)
            if pc[2] < 0.5:
                continue
            out["relo_feat"][b, k] = e
            out["relo_xy"][b, k] = pc[:2] / pc[2] + rx.normal(0, sig_px, 2)
            k += 1
        out["relo_n"][b], out["relo_frame"][b] = k, r
        out["relo_pose"][b] = out["pose"][b, r]


def make_windows(n_windows, first_id=0, tracks="dense", n_feat=150, with_prior=True, max_feat=None, max_obs=None,
                 max_samp=20, max_prior=96, max_pblk=16, td_true=None, relo=False) -> WindowArrays:
    """B synthetic windows [first_id, first_id + n_windows). tracks: 'dense' (K=1500) or 'sparse' (ragged)."""
    max_feat = max_feat or max(n_feat, 1)
    max_obs = max_obs or max_feat * abi.NFRAMES
    B = n_windows
    out = {
        "pose": np.zeros((B, abi.NFRAMES, 7)),
        "speedbias": np.zeros((B, abi.NFRAMES, 9)),
        "ex_pose": np.zeros((B, 7)),
        "inv_depth": np.ones((B, max_feat)),
        "n_feat": np.zeros(B, np.int32),
        "feat_start": np.zeros((B, max_feat), np.int32),
        "feat_nobs": np.zeros((B, max_feat), np.int32),
        "feat_obs_begin": np.zeros((B, max_feat), np.int32),
        "obs_xy": np.zeros((B, max_obs, 2)),
        "imu_n": np.zeros((B, abi.WINDOW_SIZE), np.int32),
        "imu_dt": np.zeros((B, abi.WINDOW_SIZE, max_samp)),
        "imu_acc": np.zeros((B, abi.WINDOW_SIZE, max_samp + 1, 3)),
        "imu_gyr": np.zeros((B, abi.WINDOW_SIZE, max_samp + 1, 3)),
        "imu_lin_ba": np.zeros((B, abi.WINDOW_SIZE, 3)),
        "imu_lin_bg": np.zeros((B, abi.WINDOW_SIZE, 3)),
        "prior_n": np.zeros(B, np.int32),
        "prior_nblk": np.zeros(B, np.int32),
        "prior_blk_kind": np.zeros((B, max_pblk), np.int32),
        "prior_blk_frame": np.zeros((B, max_pblk), np.int32),
        "prior_J": np.zeros((B, max_prior, max_prior)),
        "prior_r": np.zeros((B, max_prior)),
        "prior_x0": np.zeros((B, max_pblk, 9)),
    }
    if td_true is not None:
        out["obs_vel_td"], out["td"] = np.zeros((B, max_obs, 4)), np.zeros(B)
    if relo:
        out.update(relo_n=np.zeros(B, np.int32), relo_frame=np.zeros(B, np.int32), relo_feat=np.zeros((B, max_feat), np.int32),
                   relo_xy=np.zeros((B, max_feat, 2)), relo_pose=np.zeros((B, 7)))
    for b in range(B):
        _one_window(first_id + b, tracks, n_feat, with_prior, max_feat, max_obs, max_samp, max_prior, max_pblk, out, b, td_true, relo)
    dims = dict(n_windows=B, max_feat=max_feat, max_obs=max_obs, max_samp=max_samp, max_prior=max_prior, max_pblk=max_pblk)
    return WindowArrays(dims, out)


def _make_chunk(args):
    kw, first, n = args
    return make_windows(n, first_id=first, **kw).a


def make_windows_parallel(n_windows, first_id=0, procs=1, **kw) -> WindowArrays:
    """make_windows() on `procs` forked worker processes (window w only depends on (seed, w), so the result is identical to
    the serial call).  Call it before the HIP runtime is initialised in this process: the workers are plain forks."""
    procs = max(1, min(int(procs), (n_windows + 7) // 8))
    if procs == 1:
        return make_windows(n_windows, first_id=first_id, **kw)
    import multiprocessing as mp

    per = (n_windows + procs - 1) // procs
    jobs = [(kw, first_id + lo, min(per, n_windows - lo)) for lo in range(0, n_windows, per)]
    with mp.get_context("fork").Pool(len(jobs)) as pool:
        parts = pool.map(_make_chunk, jobs)
    ref = make_windows(1, first_id=first_id, **kw)
    d = dict(ref.dims)
    d["n_windows"] = n_windows
    return WindowArrays(d, {k: np.concatenate([p[k] for p in parts]) for k in ref.a})


def tile_windows(base: WindowArrays, n_windows: int) -> WindowArrays:
    """Repeat a set of generated windows cyclically up to n_windows (bench uses it to fill 4096
    windows quickly from a few hundred distinct ones; stated in bench output as `distinct`)."""
    nb = base.n_windows
    idx = np.arange(n_windows) % nb
    d = dict(base.dims)
    d["n_windows"] = n_windows
    return WindowArrays(d, {k: np.ascontiguousarray(v[idx]) for k, v in base.a.items()})


def make_fsel(n_problems, first_id=0, horizon=10, n_cand=500, n_used=0, n_cloud=150, max_features=150,
              max_cand=None, max_used=None, max_cloud=None) -> FselArrays:
    H, P = horizon, n_problems
    max_cand = max_cand or max(n_cand, 1)
    max_used = max_used or max(n_used, 1)
    max_cloud = max_cloud or max(n_cloud, 1)
    a = {
        "hor_pos": np.zeros((P, H + 1, 3)),
        "hor_quat": np.zeros((P, H + 1, 4)),
        "nr_imu": np.full(P, 20, np.int32),
        "delta_imu": np.full(P, 0.005),
        "n_cand": np.full(P, n_cand, np.int32),
        "cand_id": np.zeros((P, max_cand), np.int32),
        "cand_xy": np.zeros((P, max_cand, 2)),
        "cand_prob": np.zeros((P, max_cand)),
        "n_used": np.full(P, n_used, np.int32),
        "used_id": np.zeros((P, max_used), np.int32),
        "used_xy": np.zeros((P, max_used, 2)),
        "n_cloud": np.full(P, n_cloud, np.int32),
        "cloud_xy": np.zeros((P, max_cloud, 2)),
        "cloud_depth": np.ones((P, max_cloud)),
    }
    for p in range(P):
        rng = np.random.Generator(np.random.Philox(key=SEED_FSEL + first_id + p))
        tr = _Trajectory(rng)
        for h in range(H + 1):
            t = 0.1 * h
            a["hor_pos"][p, h] = tr.pos(t)
            a["hor_quat"][p, h] = quat_from_R(tr.R(t))

        def pix(n):
            u = rng.uniform(0, CAM["image_width"], n)
            v = rng.uniform(0, CAM["image_height"], n)
            return np.stack([(u - CAM["cx"]) / CAM["fx"], (v - CAM["cy"]) / CAM["fy"]], 1)

        ids = 1000 + np.cumsum(rng.integers(1, 4, n_used + n_cand))
        a["used_id"][p, :n_used] = ids[:n_used]
        a["used_xy"][p, :n_used] = pix(n_used)
        a["cand_id"][p, :n_cand] = ids[n_used:]
        a["cand_xy"][p, :n_cand] = pix(n_cand)
        a["cand_prob"][p, :n_cand] = rng.uniform(0.05, 1.0, n_cand).astype(np.float32).astype(np.float64)
        a["cloud_xy"][p, :n_cloud] = pix(n_cloud)
        a["cloud_depth"][p, :n_cloud] = rng.uniform(2.0, 15.0, n_cloud)
    dims = dict(n_problems=P, horizon=H, max_cand=max_cand, max_used=max_used, max_cloud=max_cloud, max_features=max_features)
    sc = dict(acc_var=ACC_N, acc_bias_var=ACC_W, q_ic=quat_from_R(RIC), t_ic=TIC, **CAM)
    return FselArrays(dims, a, sc)


class Sequence:
    """A synthetic image sequence for streaming tests (solve -> roll -> solve ...): one smooth trajectory sampled at
    10 Hz, landmarks that are born and lost along the way, 200 Hz IMU.  Only the bookkeeping the reference does on the
    host per image lives here (FeatureManager::addFeatureCheckParallax: append the new frame's observations, admit tracks
    that now pass used_num >= 2 && start_frame < WINDOW_SIZE - 2); solving, marginalizing, rolling, triangulating and the
    dead-reckoning of the newest frame are the library's (or the oracle's) job."""

    def __init__(self, sid, n_frames=24, n_landmarks=420, max_feat=150, max_samp=20):
        rng = np.random.Generator(np.random.Philox(key=SEED_WINDOW + 0x01000000 + sid))
        self.tr, self.n_frames, self.max_feat, self.max_samp = _Trajectory(rng), n_frames, max_feat, max_samp
        self.ba, self.bg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
        tk = np.arange(n_frames) * 0.1
        self.Rw, self.Pw, self.Vw = [self.tr.R(t) for t in tk], [self.tr.pos(t) for t in tk], [self.tr.vel(t) for t in tk]
        sig = 1.5 / 460.0
        self.tracks = []  # (birth frame, {frame: xy}, true inverse depth in the birth frame)
        for _ in range(n_landmarks):
            s = int(rng.integers(0, n_frames - 2))
            life = int(rng.integers(3, 14))
            xy = np.array([rng.uniform(-0.7, 0.7), rng.uniform(-0.45, 0.45)])
            depth = rng.uniform(2.0, 15.0)
            pw = self.Rw[s] @ (RIC @ (np.array([xy[0], xy[1], 1.0]) * depth) + TIC) + self.Pw[s]
            obs = {}
            for f in range(s, min(s + life, n_frames)):
                pc = RIC.T @ (self.Rw[f].T @ (pw - self.Pw[f]) - TIC)
                if pc[2] < 0.5 or abs(pc[0] / pc[2]) > 0.9 or abs(pc[1] / pc[2]) > 0.6:
                    break
                obs[f] = pc[:2] / pc[2] + rng.normal(0, sig, 2)
            if len(obs) >= 2:
                self.tracks.append((s, obs, pw))
        self.tracks.sort(key=lambda t: t[0])
        self._imu_seed = SEED_WINDOW + 0x02000000 + sid

    def imu_interval(self, j):
        """Raw samples between absolute frames j and j + 1: dt [20], acc / gyr [21, 3] (row 0 = the sample at frame j)."""
        rng = np.random.Generator(np.random.Philox(key=self._imu_seed + 1000 * j))
        ns, dt, G = 20, 0.005, np.array([0, 0, G_NORM])
        acc, gyr = np.zeros((ns + 1, 3)), np.zeros((ns + 1, 3))
        for s in range(ns + 1):
            t = 0.1 * j + s * dt
            acc[s] = self.tr.R(t).T @ (self.tr.acc(t) + G) + self.ba + rng.normal(0, ACC_N, 3)
            gyr[s] = self.tr.omega_body(t) + self.bg + rng.normal(0, GYR_N, 3)
        return np.full(ns, dt), acc, gyr

    def _admissible(self, k, li):
        """(start frame in window k, observations inside the window) of landmark li, or None if it fails the filter."""
        s, obs, _ = self.tracks[li]
        fr = [f for f in sorted(obs) if k <= f <= k + abi.WINDOW_SIZE]
        if len(fr) < 2 or fr[0] - k >= abi.WINDOW_SIZE - 2:
            return None
        return fr[0] - k, [obs[f] for f in fr]

    def _write_tracks(self, a, b, k, ids, lam):
        o = 0
        a["n_feat"][b] = len(ids)
        for e, li in enumerate(ids):
            st, ob = self._admissible(k, li) if not isinstance(li, tuple) else li
            a["feat_start"][b, e], a["feat_nobs"][b, e], a["feat_obs_begin"][b, e] = st, len(ob), o
            a["obs_xy"][b, o:o + len(ob)] = ob
            a["inv_depth"][b, e] = lam[e]
            o += len(ob)

    def first_window(self, rng_seed=0):
        """Window 0 (frames 0..10) as a one-window batch, no prior; returns (WindowArrays, landmark ids of its features)."""
        rng = np.random.Generator(np.random.Philox(key=self._imu_seed + 7 + rng_seed))
        w = make_windows(1, tracks="sparse", n_feat=1, with_prior=False, max_feat=self.max_feat, max_samp=self.max_samp)
        a = w.a
        ids = [li for li in range(len(self.tracks)) if self._admissible(0, li)][: self.max_feat]
        lam = []
        for li in ids:
            s, _, pw = self.tracks[li]
            f0 = max(s, 0)
            pc = RIC.T @ (self.Rw[f0].T @ (pw - self.Pw[f0]) - TIC)
            lam.append(1.0 / pc[2] * (1.0 + rng.normal(0, 0.10)))
        a["obs_xy"][:] = 0
        self._write_tracks(a, 0, 0, ids, lam)
        ba0, bg0 = self.ba + rng.normal(0, 0.01, 3), self.bg + rng.normal(0, 0.001, 3)
        for f in range(abi.NFRAMES):
            a["pose"][0, f, :3] = self.Pw[f] + rng.normal(0, 0.05, 3)
            a["pose"][0, f, 3:] = quat_from_R(self.Rw[f] @ _expm_so3(rng.normal(0, np.deg2rad(1.0), 3)))
            a["speedbias"][0, f] = np.concatenate([self.Vw[f] + rng.normal(0, 0.05, 3), ba0, bg0])
        for j in range(abi.WINDOW_SIZE):
            dt, acc, gyr = self.imu_interval(j)
            a["imu_n"][0, j], a["imu_dt"][0, j, :20], a["imu_acc"][0, j, :21], a["imu_gyr"][0, j, :21] = 20, dt, acc, gyr
            a["imu_lin_ba"][0, j], a["imu_lin_bg"][0, j] = ba0, bg0
        return w, ids

    def next_image(self, w, ids, k):
        """After the library rolled window k (MARGIN_OLD, shift_depth) in `w`: the host's per-image bookkeeping for window
        k + 1.  `ids` are the landmark ids of the features of window k BEFORE the roll.  Appends the observations of the new
        frame, admits the tracks that now pass the filter (inverse depth -1: to be triangulated), installs the new IMU
        interval.  Returns the landmark ids of window k + 1's features."""
        a = w.a
        n_after = int(a["n_feat"][0])
        # survivors of removeBackShiftDepth: everything except tracks that started in frame 0 with <= 2 observations
        before = [self._admissible(k, li) for li in ids]
        keep = [li for li, (st, ob) in zip(ids, before) if not (st == 0 and len(ob) <= 2)]
        assert len(keep) == n_after, (len(keep), n_after)
        lam = [float(a["inv_depth"][0, e]) for e in range(n_after)]
        rows = []
        for e, li in enumerate(keep):
            st, no, ob0 = int(a["feat_start"][0, e]), int(a["feat_nobs"][0, e]), int(a["feat_obs_begin"][0, e])
            ob = [a["obs_xy"][0, ob0 + i].copy() for i in range(no)]
            newf = k + 1 + abi.WINDOW_SIZE
            if st + no == abi.WINDOW_SIZE and newf in self.tracks[li][1]:  # tracked up to the previous newest frame
                ob.append(self.tracks[li][1][newf])
            rows.append((st, ob))
        tracked = set(ids)
        entrants = [li for li in range(len(self.tracks)) if li not in tracked and self.tracks[li][0] == k + 1 + abi.WINDOW_SIZE - 3
                    and self._admissible(k + 1, li)]
        entrants = entrants[: max(0, self.max_feat - len(keep))]
        for li in entrants:
            rows.append(self._admissible(k + 1, li))
            lam.append(-1.0)
        a["obs_xy"][:] = 0
        self._write_tracks(a, 0, k + 1, rows, lam)
        dt, acc, gyr = self.imu_interval(k + 1 + abi.WINDOW_SIZE - 1)
        a["imu_n"][0, 9], a["imu_dt"][0, 9, :20], a["imu_acc"][0, 9, :21], a["imu_gyr"][0, 9, :21] = 20, dt, acc, gyr
        return keep + entrants
