"""SURVEY 8(f)4 + B4 (ground-truth mode): the host-side format adapters.  They need no GPU (pure host bookkeeping, like
their reference counterparts), so the product functions are exercised here directly, against the oracle restatement and
an independent numpy statement of horizon_generator.cpp:73-123,169-210 and estimator_node.cpp:303-321."""
import importlib

import numpy as np
import pytest

from helpers import PKG, rel

adapters = importlib.import_module(PKG + ".adapters")
lib_m = importlib.import_module(PKG + ".lib")


@pytest.fixture(autouse=True, params=["cpu", pytest.param("gpu", marks=pytest.mark.gpu)])
def tier(request):
    """Every case of this file runs in BOTH tiers: unmarked on the CPU (the adapters are host code) and, marked `gpu`, on the
    GPU box next to the kernels they feed - there through the same libavm_hip.so the parity tests have loaded."""
    return request.param


def _gt_rows(n=700, seed=3):
    """A smooth synthetic trajectory in the EuRoC state_groundtruth_estimate0/data.csv layout, 200 Hz."""
    rng = np.random.default_rng(seed)
    t = 1403638519492829440 + 5000000 * np.arange(n, dtype=np.int64) + rng.integers(-300, 300, n)
    s = np.arange(n) * 0.005
    p = np.stack([4.4 + 0.8 * np.sin(0.7 * s), -1.6 + 0.5 * np.cos(0.9 * s), 0.6 + 0.2 * s], 1)
    ang = 0.6 * s
    axis = np.array([0.3, -0.5, 0.8]) / np.linalg.norm([0.3, -0.5, 0.8])
    q = np.concatenate([np.cos(ang / 2)[:, None], np.sin(ang / 2)[:, None] * axis], 1)  # w x y z, unit up to rounding
    q = np.round(q, 6)                                                                     # the CSV carries 6 decimals
    rows = np.concatenate([t[:, None].astype(float), np.round(p, 6), q, np.round(rng.normal(size=(n, 9)), 6)], 1)
    return rows


def _qmul(a, b):  # w x y z
    return np.array([a[0] * b[0] - a[1:] @ b[1:], *(a[0] * b[1:] + b[0] * a[1:] + np.cross(a[1:], b[1:]))])


def _qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]]) / (q @ q)


def _qrot(q, v):  # Eigen: v + w t + u x t, t = 2 u x v
    t = 2 * np.cross(q[1:], v)
    return v + q[0] * t + np.cross(q[1:], t)


class NumpyHorizon:
    def __init__(self, rows):
        self.t, self.p, self.q, self.seek = rows[:, 0] * 1e-9, rows[:, 1:4], rows[:, 4:8], 0

    def ground_truth(self, H, t0, p0, q0_xyzw, dF):
        n = len(self.t)
        ts = self.t[0] if t0 > self.t[-1] else t0
        while self.seek < n:          # while (seek < n && truth[seek++].t <= ts);
            self.seek += 1
            if not self.t[self.seek - 1] <= ts:
                break
        idx = self.seek - 1
        pos, quat = [np.asarray(p0, float)], [np.array([q0_xyzw[3], *q0_xyzw[:3]])]
        prevP, prevQ = self.p[idx], self.q[idx]
        for _ in range(H):
            nxt = self.t[idx] + dF
            while idx < n:
                idx += 1
                if not self.t[idx - 1] <= nxt:
                    break
            g_p, g_q = self.p[idx], self.q[idx]
            relQ, relP = _qmul(_qinv(prevQ), g_q), _qrot(_qinv(g_q), g_p - prevP)
            pos.append(pos[-1] + _qrot(quat[-1], relP))
            quat.append(_qmul(quat[-1], relQ))
            prevP, prevQ = g_p, g_q
        q = np.array(quat)
        return np.array(pos), np.concatenate([q[:, 1:], q[:, :1]], 1)


def test_ground_truth_horizon_matches_numpy_and_oracle(oracle, tmp_path):
    rows = _gt_rows()
    csv = tmp_path / "data.csv"
    with open(csv, "w") as f:
        f.write("#timestamp, p_RS_R_x [m], p_RS_R_y [m], p_RS_R_z [m], q_RS_w [], q_RS_x [], q_RS_y [], q_RS_z [], v..., b_w..., b_a...\n")
        for r in rows:
            f.write("%d," % int(r[0]) + ",".join("%.6f" % v for v in r[1:]) + "\n")
    hg = adapters.HorizonGenerator()
    assert hg.loadGroundTruth(str(csv)) == len(rows)
    hm = adapters.HorizonGenerator()
    hm.setGroundTruth(rows)
    og, ng = oracle.GroundTruth(rows), NumpyHorizon(rows)
    rng = np.random.default_rng(1)
    q0 = rng.normal(size=4); q0 /= np.linalg.norm(q0)
    p0 = rng.normal(size=3)
    t0 = rows[0, 0] * 1e-9
    # consecutive frames (the cursor is stateful), a time step backwards (the cursor still advances by one row) and a
    # "random first state" later than the table (falls back to the first row's time)
    for k, (tk, H, dF) in enumerate([(t0 + 0.101, 10, 0.05), (t0 + 0.151, 10, 0.05), (t0 + 0.12, 3, 0.1), (t0 + 1e6, 5, 0.05), (t0 + 0.31, 13, 0.05)]):
        gp, gq = hg.groundTruth(H, tk, p0, q0, dF)
        mp, mq = hm.groundTruth(H, tk, p0, q0, dF)
        rc, op, oq = og.horizon(H, tk, p0, q0, dF)
        npos, nq = ng.ground_truth(H, tk, p0, q0, dF)
        assert rc == 0 and hg.seek_idx == hm.seek_idx == og.seek_idx == ng.seek, k
        assert np.array_equal(gp, mp) and np.array_equal(gq, mq)          # CSV text and in-memory rows give the same table
        assert rel(gp, op) < 1e-14 and rel(gq, oq) < 1e-14
        assert rel(op, npos) < 1e-12 and rel(oq, nq) < 1e-12
        assert np.array_equal(gp[0], p0) and np.array_equal(gq[0], q0)
        p0, q0 = gp[1], gq[1] / np.linalg.norm(gq[1])
    # the relative motion of the table is reproduced: the step lengths equal the ground truth's own (6-decimal unit quaternions)
    assert abs(np.linalg.norm(np.diff(gp, axis=0), axis=1) - np.linalg.norm(np.diff(npos, axis=0), axis=1)).max() < 1e-12
    # running off the end of the table is an error, not an out-of-bounds read
    with pytest.raises(lib_m.AvmError):
        hg.groundTruth(10, rows[-5, 0] * 1e-9, p0, q0, 0.05)
    assert og.horizon(10, rows[-5, 0] * 1e-9, p0, q0, 0.05)[0] != 0
    with pytest.raises(lib_m.AvmError):
        adapters.HorizonGenerator().loadGroundTruth(str(tmp_path / "missing.csv"))


def test_pointcloud_decode_matches_numpy_and_oracle(oracle):
    rng = np.random.default_rng(5)
    n, num_cam = 180, 1
    ids = rng.permutation(4000)[:n]
    pts = np.concatenate([rng.normal(size=(n, 2)), np.ones((n, 1))], 1).astype(np.float32)
    ch = [(ids * num_cam).astype(np.float32)] + [rng.normal(size=n).astype(np.float32) for _ in range(4)] + [rng.uniform(size=n).astype(np.float32)]
    fid, cam, out = adapters.image_from_pointcloud(pts, ch, num_cam)
    rc, ofid, ocam, oout = oracle.image_from_pointcloud(pts, ch, num_cam)
    order = np.argsort(ids, kind="stable")
    exp = np.concatenate([pts[order].astype(float)] + [c[order].astype(float)[:, None] for c in ch[1:]], 1)
    assert rc == 0 and np.array_equal(fid, ids[order]) and np.array_equal(fid, ofid) and (cam == 0).all() and np.array_equal(cam, ocam)
    assert np.array_equal(out, exp) and np.array_equal(out, oout)
    # two cameras: id = v / 2, camera = v % 2, message order kept inside an id; float ids carry the +0.5 rounding
    v = np.array([7, 6, 3, 2, 11], np.float32) + np.float32(0.25)
    pts2 = np.concatenate([rng.normal(size=(5, 2)), np.ones((5, 1))], 1).astype(np.float32)
    ch2 = [v] + [np.arange(5, dtype=np.float32) + k for k in range(5)]
    fid, cam, out = adapters.image_from_pointcloud(pts2, ch2, 2)
    rc, ofid, ocam, oout = oracle.image_from_pointcloud(pts2, ch2, 2)
    assert fid.tolist() == [1, 1, 3, 3, 5] and cam.tolist() == [1, 0, 1, 0, 1] and out[:, 3].tolist() == [2.0, 3.0, 0.0, 1.0, 4.0]
    assert rc == 0 and np.array_equal(fid, ofid) and np.array_equal(cam, ocam) and np.array_equal(out, oout)
    # ROS_ASSERT(z == 1)
    pts2[3, 2] = 0.5
    with pytest.raises(lib_m.AvmError):
        adapters.image_from_pointcloud(pts2, ch2, 2)
    assert oracle.image_from_pointcloud(pts2, ch2, 2)[0] != 0
    # empty message
    fid, cam, out = adapters.image_from_pointcloud(np.zeros((0, 3), np.float32), [np.zeros(0, np.float32)] * 6)
    assert fid.size == 0 and out.shape == (0, 8)
