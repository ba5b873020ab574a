"""Generates tests/golden/*.npz: known-answer vectors from an INDEPENDENT numpy restatement of the
reference math (written from the reference sources, not from oracle/ or csrc/).

Run once in the build container:  python tests/golden/gen_golden.py
The fixtures are data (inputs + expected outputs); nothing here reads /root/reference at test time.

Sources restated (file:line in /root/reference):
  ProjectionFactor::Evaluate            vins_estimator/src/factor/projection_factor.cpp:21-121
  IntegrationBase::midPointIntegration  vins_estimator/src/factor/integration_base.h:54-128
  IntegrationBase::evaluate             vins_estimator/src/factor/integration_base.h:160-186
  CauchyLoss / Corrector                vins_estimator/src/factor/marginalization_factor.cpp:37-68
  createLinearImuMatrices               vins_estimator/src/feature_selector.cpp:531-598 with the parameter set of
                                        support_files/scripts/createMatricesLinearImuFactor.m:17-101 (delta=0.005,
                                        accVar=0.01, biasVar=1e-4, n=2) and test_ccT.m:24-36 (eigenvalues of CC^T)
  Delta_ell block structure             vins_estimator/src/feature_selector.cpp:338-359
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20260928)


def skew(v):
    return np.array([[0, -v[2], v[1]], [v[2], 0, -v[0]], [-v[1], v[0], 0]])


def q2R(q):  # q = (w, x, y, z)
    w, x, y, z = q
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def qmul(a, b):
    aw, ax, ay, az = a
    bw, bx, by, bz = b
    return np.array([aw * bw - ax * bx - ay * by - az * bz, aw * bx + ax * bw + ay * bz - az * by,
                     aw * by + ay * bw + az * bx - ax * bz, aw * bz + az * bw + ax * by - ay * bx])


def rand_q():
    q = rng.normal(size=4)
    return q / np.linalg.norm(q)


def pose7(P, q):  # parameter block layout x y z qx qy qz qw
    return np.array([P[0], P[1], P[2], q[1], q[2], q[3], q[0]])


def projection_factor(pose_i, pose_j, ex, lam, pts_i, pts_j, s):
    Pi, Qi = pose_i[:3], np.array([pose_i[6], *pose_i[3:6]])
    Pj, Qj = pose_j[:3], np.array([pose_j[6], *pose_j[3:6]])
    tic, qic = ex[:3], np.array([ex[6], *ex[3:6]])
    Ri, Rj, ric = q2R(Qi), q2R(Qj), q2R(qic)
    pci = pts_i / lam
    pimu_i = ric @ pci + tic
    pw = Ri @ pimu_i + Pi
    qinv = lambda q: np.array([q[0], -q[1], -q[2], -q[3]]) / (q @ q)
    # Qj.inverse() * (..), qic.inverse() * (..) (projection_factor.cpp:37-38): Eigen's conj / |q|^2, not R^T, for a quaternion that is
    # unit only to FP64 rounding (see imu_residual_raw); the Jacobians below DO use the transposes, as the reference does (:52-81)
    pimu_j = q2R(qinv(Qj)) @ (pw - Pj)
    pcj = q2R(qinv(qic)) @ (pimu_j - tic)
    dep = pcj[2]
    r = s * (pcj[:2] / dep - pts_j[:2])
    red = s * np.array([[1 / dep, 0, -pcj[0] / dep**2], [0, 1 / dep, -pcj[1] / dep**2]])
    Ji = red @ np.hstack([ric.T @ Rj.T, ric.T @ Rj.T @ Ri @ -skew(pimu_i)])
    Jj = red @ np.hstack([ric.T @ -Rj.T, ric.T @ skew(pimu_j)])
    tmp_r = ric.T @ Rj.T @ Ri @ ric
    Jex = red @ np.hstack([ric.T @ (Rj.T @ Ri - np.eye(3)),
                           -tmp_r @ skew(pci) + skew(tmp_r @ pci) + skew(ric.T @ (Rj.T @ (Ri @ tic + Pi - Pj) - tic))])
    Je = red @ ric.T @ Rj.T @ Ri @ ric @ pts_i * -1.0 / lam**2
    return r, Ji, Jj, Jex, Je


def cauchy_correct(r, J, a=1.0):
    s = float(r @ r)
    b, c = a * a, 1.0 / (a * a)
    rho1 = 1.0 / (1.0 + s * c)
    return np.sqrt(rho1) * r, np.sqrt(rho1) * J, 0.5 * b * np.log(1.0 + s * c)


def preintegrate(acc, gyr, dts, ba, bg, n):
    acc_n, gyr_n, acc_w, gyr_w = n
    Q = np.diag([acc_n**2] * 3 + [gyr_n**2] * 3 + [acc_n**2] * 3 + [gyr_n**2] * 3 + [acc_w**2] * 3 + [gyr_w**2] * 3)
    dp, dv, dq = np.zeros(3), np.zeros(3), np.array([1.0, 0, 0, 0])
    J, P = np.eye(15), np.zeros((15, 15))
    a0, g0 = acc[0], gyr[0]
    for k, dt in enumerate(dts):
        a1, g1 = acc[k + 1], gyr[k + 1]
        un_a0 = q2R(dq) @ (a0 - ba)
        ug = 0.5 * (g0 + g1) - bg
        rq = qmul(dq, np.array([1.0, ug[0] * dt / 2, ug[1] * dt / 2, ug[2] * dt / 2]))
        un_a1 = q2R(rq) @ (a1 - ba)
        ua = 0.5 * (un_a0 + un_a1)
        rp = dp + dv * dt + 0.5 * ua * dt * dt
        rv = dv + ua * dt
        Rd, Rr = q2R(dq), q2R(rq)
        Rw, Ra0, Ra1 = skew(ug), skew(a0 - ba), skew(a1 - ba)
        I = np.eye(3)
        F = np.zeros((15, 15))
        F[0:3, 0:3] = I
        F[0:3, 3:6] = -0.25 * Rd @ Ra0 * dt * dt + -0.25 * Rr @ Ra1 @ (I - Rw * dt) * dt * dt
        F[0:3, 6:9] = I * dt
        F[0:3, 9:12] = -0.25 * (Rd + Rr) * dt * dt
        F[0:3, 12:15] = -0.25 * Rr @ Ra1 * dt * dt * -dt
        F[3:6, 3:6] = I - Rw * dt
        F[3:6, 12:15] = -I * dt
        F[6:9, 3:6] = -0.5 * Rd @ Ra0 * dt + -0.5 * Rr @ Ra1 @ (I - Rw * dt) * dt
        F[6:9, 6:9] = I
        F[6:9, 9:12] = -0.5 * (Rd + Rr) * dt
        F[6:9, 12:15] = -0.5 * Rr @ Ra1 * dt * -dt
        F[9:12, 9:12] = I
        F[12:15, 12:15] = I
        V = np.zeros((15, 18))
        V[0:3, 0:3] = 0.25 * Rd * dt * dt
        V[0:3, 3:6] = 0.25 * -Rr @ Ra1 * dt * dt * 0.5 * dt
        V[0:3, 6:9] = 0.25 * Rr * dt * dt
        V[0:3, 9:12] = V[0:3, 3:6]
        V[3:6, 3:6] = 0.5 * I * dt
        V[3:6, 9:12] = 0.5 * I * dt
        V[6:9, 0:3] = 0.5 * Rd * dt
        V[6:9, 3:6] = 0.5 * -Rr @ Ra1 * dt * 0.5 * dt
        V[6:9, 6:9] = 0.5 * Rr * dt
        V[6:9, 9:12] = V[6:9, 3:6]
        V[9:12, 12:15] = I * dt
        V[12:15, 15:18] = I * dt
        J = F @ J
        P = F @ P @ F.T + V @ Q @ V.T
        dp, dv, dq = rp, rv, rq / np.linalg.norm(rq)
        a0, g0 = a1, g1
    return dp, dq, dv, J, P, float(np.sum(dts))


def imu_residual_raw(pre, G, pose_i, sb_i, pose_j, sb_j, lba, lbg):
    dp, dq, dv, J, P, sdt = pre
    Pi, Qi = pose_i[:3], np.array([pose_i[6], *pose_i[3:6]])
    Pj, Qj = pose_j[:3], np.array([pose_j[6], *pose_j[3:6]])
    Vi, Bai, Bgi = sb_i[:3], sb_i[3:6], sb_i[6:9]
    Vj, Baj, Bgj = sb_j[:3], sb_j[3:6], sb_j[6:9]
    dba, dbg = Bai - lba, Bgi - lbg
    th = J[3:6, 12:15] @ dbg
    cdq = qmul(dq, np.array([1.0, th[0] / 2, th[1] / 2, th[2] / 2]))
    cdv = dv + J[6:9, 9:12] @ dba + J[6:9, 12:15] @ dbg
    cdp = dp + J[0:3, 9:12] @ dba + J[0:3, 12:15] @ dbg
    conj = lambda q: np.array([q[0], -q[1], -q[2], -q[3]]) / (q @ q)
    # Qi.inverse() * v (integration_base.h:176,178): the rotation by conj(Qi) / |Qi|^2, which for a quaternion that is unit only to
    # FP64 rounding is not R(Qi)^T but 1e-16 away from it (found by the 50-digit run, gen_solve_trace_mp.py: the two readings of
    # the reference agreed to 1e-16 instead of 1e-30 until this line said what Eigen does)
    RiT = q2R(conj(Qi))
    r = np.zeros(15)
    r[0:3] = RiT @ (0.5 * G * sdt * sdt + Pj - Pi - Vi * sdt) - cdp
    r[3:6] = 2 * qmul(conj(cdq), qmul(conj(Qi), Qj))[1:]
    r[6:9] = RiT @ (G * sdt + Vj - Vi) - cdv
    r[9:12] = Baj - Bai
    r[12:15] = Bgj - Bgi
    return r


def linear_imu_matrices(Ri_list, n, delta, accVar, biasVar):
    """feature_selector.cpp:531-598 with rotations given (the MATLAB transcript propagates Rh = Rh*Rimu)."""
    Nij, Mij = np.zeros((3, 3)), np.zeros((3, 3))
    c11 = c12 = 0.0
    for i in range(n):
        jkh = n - i - 0.5
        Nij += jkh * Ri_list[i]
        Mij += Ri_list[i]
        c11 += jkh**2
        c12 += jkh
    cov = np.zeros((9, 9))
    cov[0:3, 0:3] = np.eye(3) * n * c11 * delta**4 * accVar
    cov[0:3, 3:6] = np.eye(3) * c12 * delta**3 * accVar
    cov[3:6, 0:3] = cov[0:3, 3:6].T
    cov[3:6, 3:6] = np.eye(3) * n * delta**2 * accVar
    cov[6:9, 6:9] = np.eye(3) * n * biasVar
    A = -np.eye(9)
    A[0:3, 3:6] = -np.eye(3) * n * delta
    A[0:3, 6:9] = Nij * delta**2
    A[3:6, 6:9] = Mij * delta
    return cov, A


def main():
    out = {}
    # ---- projection factors
    K = 24
    P = dict(pose_i=[], pose_j=[], ex=[], lam=[], pts_i=[], pts_j=[], r=[], Ji=[], Jj=[], Jex=[], Je=[], r_c=[], J_c=[], cost=[])
    for _ in range(K):
        pi = pose7(rng.normal(0, 1, 3), rand_q())
        pj = pose7(pi[:3] + rng.normal(0, 0.3, 3), qmul(np.array([pi[6], *pi[3:6]]), np.array([1, *rng.normal(0, 0.05, 3)]) / 1.0))
        pj[3:] /= np.linalg.norm(pj[3:])
        ex = pose7(rng.normal(0, 0.05, 3), rand_q())
        lam = rng.uniform(0.07, 0.5)
        pts_i = np.array([rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4), 1.0])
        # a consistent observation: project the landmark into camera j and add noise
        r0, *_ = projection_factor(pi, pj, ex, lam, pts_i, np.array([0, 0, 1.0]), 1.0)
        pts_j = np.array([r0[0] + rng.normal(0, 0.01), r0[1] + rng.normal(0, 0.01), 1.0])
        s = 460.0 / 1.5
        r, Ji, Jj, Jex, Je = projection_factor(pi, pj, ex, lam, pts_i, pts_j, s)
        J = np.hstack([Ji, Jj, Je[:, None]])
        rc, Jc, cost = cauchy_correct(r, J)
        for k, v in zip(P, (pi, pj, ex, lam, pts_i, pts_j, r, Ji, Jj, Jex, Je, rc, Jc, cost)):
            P[k].append(v)
    for k, v in P.items():
        out["proj_" + k] = np.array(v)
    # ---- IMU pre-integration + raw residual
    n = (0.08, 0.004, 0.00004, 2.0e-6)
    ns = 20
    acc = rng.normal(0, 0.5, (ns + 1, 3)) + np.array([0, 0, 9.8])
    gyr = rng.normal(0, 0.2, (ns + 1, 3))
    dts = np.full(ns, 0.005) + rng.uniform(-2e-4, 2e-4, ns)
    lba, lbg = rng.normal(0, 0.02, 3), rng.normal(0, 0.002, 3)
    pre = preintegrate(acc, gyr, dts, lba, lbg, n)
    out.update(imu_acc=acc, imu_gyr=gyr, imu_dt=dts, imu_lba=lba, imu_lbg=lbg, imu_noise=np.array(n),
               pre_dp=pre[0], pre_dq_wxyz=pre[1], pre_dv=pre[2], pre_J=pre[3], pre_P=pre[4], pre_sum_dt=pre[5])
    G = np.array([0, 0, 9.81007])
    pi = pose7(rng.normal(0, 1, 3), rand_q())
    pj = pose7(pi[:3] + rng.normal(0, 0.1, 3), rand_q())
    sbi = np.concatenate([rng.normal(0, 1, 3), lba + rng.normal(0, 0.01, 3), lbg + rng.normal(0, 0.001, 3)])
    sbj = sbi + rng.normal(0, 0.01, 9)
    out.update(imu_pose_i=pi, imu_pose_j=pj, imu_sb_i=sbi, imu_sb_j=sbj, imu_G=G,
               imu_r_raw=imu_residual_raw(pre, G, pi, sbi, pj, sbj, lba, lbg))
    # ---- createLinearImuMatrices with the MATLAB script's parameter set
    delta, accVar, biasVar, nI = 0.005, 0.01, 0.0001, 2
    t = delta * 2
    w = np.array([1.0, 0, 1.0]) * t  # Rj = expm(skew([1 0 1])*t), Ri = I
    th = np.linalg.norm(w) / nI
    ax = w / np.linalg.norm(w)
    Kx = skew(ax)
    Rimu = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    Rs = [np.eye(3)]
    for _ in range(nI - 1):
        Rs.append(Rs[-1] @ Rimu)
    cov, A = linear_imu_matrices(Rs, nI, delta, accVar, biasVar)
    out.update(lin_delta=delta, lin_accVar=accVar, lin_biasVar=biasVar, lin_n=nI, lin_cov=cov, lin_A=A, lin_Omega=np.linalg.inv(cov),
               lin_Rj=Rs[-1] @ Rimu, lin_eig_cct=np.sort(np.linalg.eigvalsh(cov[:6, :6])))
    np.savez(os.path.join(HERE, "factors.npz"), **out)
    print("wrote factors.npz with", len(out), "arrays")


if __name__ == "__main__":
    main()
