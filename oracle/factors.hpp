// oracle/factors.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// CPU restatement of the factor math on HP-A, following the reference line by line:
//   IntegrationBase      vins_estimator/src/factor/integration_base.h:13-186
//   IMUFactor::Evaluate  vins_estimator/src/factor/imu_factor.h:19-179
//   ProjectionFactor     vins_estimator/src/factor/projection_factor.cpp:21-121
//   MarginalizationFactor::Evaluate  vins_estimator/src/factor/marginalization_factor.cpp:333-381
//   CauchyLoss + Corrector (Ceres, not vendored) as re-implemented in-tree at
//                        vins_estimator/src/factor/marginalization_factor.cpp:37-68
// PARITY UNPINNED: the reference ships no tests / golden vectors and cannot be built here.
#pragma once
#include "linalg.hpp"

namespace avmo {

enum { O_P = 0, O_R = 3, O_V = 6, O_BA = 9, O_BG = 12 };  // parameters.h:58-65

struct ImuNoise {
  double acc_n, gyr_n, acc_w, gyr_w;
};

// integration_base.h:9-209
struct PreIntegration {
  V3 acc_0, gyr_0;
  V3 linearized_ba, linearized_bg;
  Mat jacobian, covariance, noise;
  double sum_dt = 0;
  V3 delta_p;
  Q delta_q;
  V3 delta_v;

  PreIntegration(V3 a0, V3 g0, V3 ba, V3 bg, const ImuNoise& nz)
      : acc_0(a0), gyr_0(g0), linearized_ba(ba), linearized_bg(bg), jacobian(Mat::identity(15)), covariance(15, 15), noise(18, 18) {
    // integration_base.h:21-27
    for (int i = 0; i < 3; i++) {
      noise(0 + i, 0 + i) = nz.acc_n * nz.acc_n;
      noise(3 + i, 3 + i) = nz.gyr_n * nz.gyr_n;
      noise(6 + i, 6 + i) = nz.acc_n * nz.acc_n;
      noise(9 + i, 9 + i) = nz.gyr_n * nz.gyr_n;
      noise(12 + i, 12 + i) = nz.acc_w * nz.acc_w;
      noise(15 + i, 15 + i) = nz.gyr_w * nz.gyr_w;
    }
  }

  // midPointIntegration (integration_base.h:54-128) + propagate (:130-158)
  void push_back(double _dt, V3 _acc_1, V3 _gyr_1) {
    V3 un_acc_0 = rot(delta_q, acc_0 - linearized_ba);
    V3 un_gyr = 0.5 * (gyr_0 + _gyr_1) - linearized_bg;
    Q result_delta_q = delta_q * Q(1, un_gyr.x * _dt / 2, un_gyr.y * _dt / 2, un_gyr.z * _dt / 2);
    V3 un_acc_1 = rot(result_delta_q, _acc_1 - linearized_ba);
    V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    V3 result_delta_p = delta_p + delta_v * _dt + 0.5 * un_acc * _dt * _dt;
    V3 result_delta_v = delta_v + un_acc * _dt;

    {
      V3 w_x = 0.5 * (gyr_0 + _gyr_1) - linearized_bg;
      V3 a_0_x = acc_0 - linearized_ba;
      V3 a_1_x = _acc_1 - linearized_ba;
      M3 R_w_x = skew(w_x), R_a_0_x = skew(a_0_x), R_a_1_x = skew(a_1_x);
      M3 I = M3::identity();
      M3 Rd = toR(delta_q), Rr = toR(result_delta_q);
      Mat F(15, 15);
      F.setBlock(0, 0, I);
      F.setBlock(0, 3, (-0.25 * (Rd * R_a_0_x)) * (_dt * _dt) + (-0.25 * (Rr * R_a_1_x * (I - _dt * R_w_x))) * (_dt * _dt));
      F.setBlock(0, 6, _dt * I);
      F.setBlock(0, 9, (-0.25 * (Rd + Rr)) * (_dt * _dt));
      F.setBlock(0, 12, (-0.25 * (Rr * R_a_1_x)) * (_dt * _dt * -_dt));
      F.setBlock(3, 3, I - _dt * R_w_x);
      F.setBlock(3, 12, (-1.0 * _dt) * I);
      F.setBlock(6, 3, (-0.5 * (Rd * R_a_0_x)) * _dt + (-0.5 * (Rr * R_a_1_x * (I - _dt * R_w_x))) * _dt);
      F.setBlock(6, 6, I);
      F.setBlock(6, 9, (-0.5 * (Rd + Rr)) * _dt);
      F.setBlock(6, 12, (-0.5 * (Rr * R_a_1_x)) * (_dt * -_dt));
      F.setBlock(9, 9, I);
      F.setBlock(12, 12, I);

      Mat V(15, 18);
      V.setBlock(0, 0, (0.25 * Rd) * (_dt * _dt));
      M3 v03 = (0.25 * (-Rr) * R_a_1_x) * (_dt * _dt * 0.5 * _dt);
      V.setBlock(0, 3, v03);
      V.setBlock(0, 6, (0.25 * Rr) * (_dt * _dt));
      V.setBlock(0, 9, v03);
      V.setBlock(3, 3, (0.5 * _dt) * I);
      V.setBlock(3, 9, (0.5 * _dt) * I);
      V.setBlock(6, 0, (0.5 * Rd) * _dt);
      M3 v63 = (0.5 * (-Rr) * R_a_1_x) * (_dt * 0.5 * _dt);
      V.setBlock(6, 3, v63);
      V.setBlock(6, 6, (0.5 * Rr) * _dt);
      V.setBlock(6, 9, v63);
      V.setBlock(9, 12, _dt * I);
      V.setBlock(12, 15, _dt * I);

      jacobian = matmul(F, jacobian);
      Mat FP = matmul(matmul(F, covariance), transpose(F));
      Mat VQ = matmul(matmul(V, noise), transpose(V));
      for (size_t i = 0; i < FP.a.size(); i++) covariance.a[i] = FP.a[i] + VQ.a[i];
    }
    delta_p = result_delta_p;
    delta_q = normalized(result_delta_q);  // integration_base.h:153
    delta_v = result_delta_v;
    sum_dt += _dt;
    acc_0 = _acc_1;
    gyr_0 = _gyr_1;
  }

  // integration_base.h:160-186
  void evaluate(V3 Pi, Q Qi, V3 Vi, V3 Bai, V3 Bgi, V3 Pj, Q Qj, V3 Vj, V3 Baj, V3 Bgj, V3 G, double* residuals) const {
    M3 dp_dba = jacobian.block3(O_P, O_BA), dp_dbg = jacobian.block3(O_P, O_BG);
    M3 dq_dbg = jacobian.block3(O_R, O_BG);
    M3 dv_dba = jacobian.block3(O_V, O_BA), dv_dbg = jacobian.block3(O_V, O_BG);
    V3 dba = Bai - linearized_ba, dbg = Bgi - linearized_bg;
    Q corrected_delta_q = delta_q * deltaQ(dq_dbg * dbg);
    V3 corrected_delta_v = delta_v + dv_dba * dba + dv_dbg * dbg;
    V3 corrected_delta_p = delta_p + dp_dba * dba + dp_dbg * dbg;
    V3 rp = rot(inverse(Qi), 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt) - corrected_delta_p;
    V3 rr = 2.0 * (inverse(corrected_delta_q) * (inverse(Qi) * Qj)).vec();
    V3 rv = rot(inverse(Qi), G * sum_dt + Vj - Vi) - corrected_delta_v;
    V3 rba = Baj - Bai, rbg = Bgj - Bgi;
    for (int i = 0; i < 3; i++) {
      residuals[O_P + i] = rp[i];
      residuals[O_R + i] = rr[i];
      residuals[O_V + i] = rv[i];
      residuals[O_BA + i] = rba[i];
      residuals[O_BG + i] = rbg[i];
    }
  }
};

// Utility::Qleft / Qright (utility/utility.h:46-64); positify is a no-op (:37-44)
inline void Qleft(const Q& q, double L[4][4]) {
  M3 S = skew(q.vec());
  L[0][0] = q.w, L[0][1] = -q.x, L[0][2] = -q.y, L[0][3] = -q.z;
  L[1][0] = q.x, L[2][0] = q.y, L[3][0] = q.z;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) L[1 + i][1 + j] = (i == j ? q.w : 0.0) + S(i, j);
}
inline void Qright(const Q& p, double R[4][4]) {
  M3 S = skew(p.vec());
  R[0][0] = p.w, R[0][1] = -p.x, R[0][2] = -p.y, R[0][3] = -p.z;
  R[1][0] = p.x, R[2][0] = p.y, R[3][0] = p.z;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) R[1 + i][1 + j] = (i == j ? p.w : 0.0) - S(i, j);
}
inline M3 bottomRight3(const double A[4][4]) {
  M3 r;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) r(i, j) = A[1 + i][1 + j];
  return r;
}
inline M3 QleftQright_br(const Q& a, const Q& b) {
  double L[4][4], R[4][4], P[4][4];
  Qleft(a, L);
  Qright(b, R);
  for (int i = 0; i < 4; i++)
    for (int j = 0; j < 4; j++) {
      double s = 0;
      for (int k = 0; k < 4; k++) s += L[i][k] * R[k][j];
      P[i][j] = s;
    }
  return bottomRight3(P);
}
inline M3 Qleft_br(const Q& a) {
  double L[4][4];
  Qleft(a, L);
  return bottomRight3(L);
}

inline void getPose(const double* p, V3& P, Q& q) {
  P = V3(p[0], p[1], p[2]);
  q = Q(p[6], p[3], p[4], p[5]);
}

// sqrt_info = LLT(covariance.inverse()).matrixL().transpose()  (imu_factor.h:64)
inline Mat imu_sqrt_info(const PreIntegration& pre) {
  Mat info = inverse_lu(pre.covariance);
  Mat L = info;
  llt_lower(L);
  Mat U(15, 15);
  for (int i = 0; i < 15; i++)
    for (int j = i; j < 15; j++) U(i, j) = L(j, i);
  return U;
}

// IMUFactor::Evaluate (imu_factor.h:19-179). jac[0]:15x7, jac[1]:15x9, jac[2]:15x7, jac[3]:15x9, row-major.
inline void imu_factor_evaluate(const PreIntegration& pre, const Mat& sqrt_info, V3 G, const double* pose_i, const double* sb_i,
                                const double* pose_j, const double* sb_j, double* residuals, double* jac[4]) {
  V3 Pi, Pj;
  Q Qi, Qj;
  getPose(pose_i, Pi, Qi);
  getPose(pose_j, Pj, Qj);
  V3 Vi(sb_i[0], sb_i[1], sb_i[2]), Bai(sb_i[3], sb_i[4], sb_i[5]), Bgi(sb_i[6], sb_i[7], sb_i[8]);
  V3 Vj(sb_j[0], sb_j[1], sb_j[2]), Baj(sb_j[3], sb_j[4], sb_j[5]), Bgj(sb_j[6], sb_j[7], sb_j[8]);

  double raw[15];
  pre.evaluate(Pi, Qi, Vi, Bai, Bgi, Pj, Qj, Vj, Baj, Bgj, G, raw);
  for (int i = 0; i < 15; i++) {
    double s = 0;
    for (int k = 0; k < 15; k++) s += sqrt_info(i, k) * raw[k];
    residuals[i] = s;
  }
  if (!jac) return;
  double sum_dt = pre.sum_dt;
  M3 dp_dba = pre.jacobian.block3(O_P, O_BA), dp_dbg = pre.jacobian.block3(O_P, O_BG);
  M3 dq_dbg = pre.jacobian.block3(O_R, O_BG);
  M3 dv_dba = pre.jacobian.block3(O_V, O_BA), dv_dbg = pre.jacobian.block3(O_V, O_BG);
  M3 RiT = toR(inverse(Qi));
  Q corrected_delta_q = pre.delta_q * deltaQ(dq_dbg * (Bgi - pre.linearized_bg));

  auto premul = [&](const Mat& Jm, double* out, int cols) {
    for (int i = 0; i < 15; i++)
      for (int j = 0; j < cols; j++) {
        double s = 0;
        for (int k = 0; k < 15; k++) s += sqrt_info(i, k) * Jm(k, j);
        out[i * cols + j] = s;
      }
  };
  if (jac[0]) {
    Mat J(15, 7);
    J.setBlock(O_P, O_P, -RiT);
    J.setBlock(O_P, O_R, skew(rot(inverse(Qi), 0.5 * G * sum_dt * sum_dt + Pj - Pi - Vi * sum_dt)));
    J.setBlock(O_R, O_R, -QleftQright_br(inverse(Qj) * Qi, corrected_delta_q));
    J.setBlock(O_V, O_R, skew(rot(inverse(Qi), G * sum_dt + Vj - Vi)));
    premul(J, jac[0], 7);
  }
  if (jac[1]) {
    Mat J(15, 9);
    J.setBlock(O_P, O_V - O_V, (-sum_dt) * RiT);
    J.setBlock(O_P, O_BA - O_V, -dp_dba);
    J.setBlock(O_P, O_BG - O_V, -dp_dbg);
    J.setBlock(O_R, O_BG - O_V, (-Qleft_br(inverse(Qj) * Qi * pre.delta_q)) * dq_dbg);
    J.setBlock(O_V, O_V - O_V, -RiT);
    J.setBlock(O_V, O_BA - O_V, -dv_dba);
    J.setBlock(O_V, O_BG - O_V, -dv_dbg);
    J.setBlock(O_BA, O_BA - O_V, -M3::identity());
    J.setBlock(O_BG, O_BG - O_V, -M3::identity());
    premul(J, jac[1], 9);
  }
  if (jac[2]) {
    Mat J(15, 7);
    J.setBlock(O_P, O_P, RiT);
    J.setBlock(O_R, O_R, Qleft_br(inverse(corrected_delta_q) * inverse(Qi) * Qj));
    premul(J, jac[2], 7);
  }
  if (jac[3]) {
    Mat J(15, 9);
    J.setBlock(O_V, O_V - O_V, RiT);
    J.setBlock(O_BA, O_BA - O_V, M3::identity());
    J.setBlock(O_BG, O_BG - O_V, M3::identity());
    premul(J, jac[3], 9);
  }
}

// ProjectionTdFactor::Evaluate (projection_td_factor.cpp:34-141), UNIT_SPHERE_ERROR off: ProjectionFactor with the
// observations shifted along their image velocity by the camera-IMU time offset td (and the rolling-shutter row
// time), plus the 2x1 Jacobian with respect to td.  row_i / row_j are the raw image rows (the constructor subtracts
// ROW / 2, :19-20).  jac: 2 x 20 row-major, columns pose_i 6 | pose_j 6 | ex_pose 6 | inv_depth 1 | td 1.
inline void projection_td_factor_evaluate(const double* pts_i2, const double* pts_j2, const double* vel_i2, const double* vel_j2, double td_i,
                                          double td_j, double row_i_raw, double row_j_raw, double TR, double ROW, double sqrt_info_s,
                                          const double* pose_i, const double* pose_j, const double* ex_pose, double inv_dep_i, double td,
                                          double* residuals, double* jac20) {
  V3 Pi, Pj, tic;
  Q Qi, Qj, qic;
  getPose(pose_i, Pi, Qi);
  getPose(pose_j, Pj, Qj);
  getPose(ex_pose, tic, qic);
  const V3 pts_i(pts_i2[0], pts_i2[1], 1.0), pts_j(pts_j2[0], pts_j2[1], 1.0);
  const V3 velocity_i(vel_i2[0], vel_i2[1], 0.0), velocity_j(vel_j2[0], vel_j2[1], 0.0);
  const double row_i = row_i_raw - ROW / 2, row_j = row_j_raw - ROW / 2;
  const V3 pts_i_td = pts_i - (td - td_i + TR / ROW * row_i) * velocity_i;
  const V3 pts_j_td = pts_j - (td - td_j + TR / ROW * row_j) * velocity_j;
  const V3 pts_camera_i = pts_i_td / inv_dep_i;
  const V3 pts_imu_i = rot(qic, pts_camera_i) + tic;
  const V3 pts_w = rot(Qi, pts_imu_i) + Pi;
  const V3 pts_imu_j = rot(inverse(Qj), pts_w - Pj);
  const V3 pts_camera_j = rot(inverse(qic), pts_imu_j - tic);
  const double dep_j = pts_camera_j.z;
  residuals[0] = sqrt_info_s * ((pts_camera_j.x / dep_j) - pts_j_td.x);
  residuals[1] = sqrt_info_s * ((pts_camera_j.y / dep_j) - pts_j_td.y);
  if (!jac20) return;
  const M3 Ri = toR(Qi), Rj = toR(Qj), ric = toR(qic);
  double reduce[2][3] = {{1. / dep_j, 0, -pts_camera_j.x / (dep_j * dep_j)}, {0, 1. / dep_j, -pts_camera_j.y / (dep_j * dep_j)}};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) reduce[i][j] = sqrt_info_s * reduce[i][j];
  auto red = [&](const M3& A, const M3& B, int col0) {
    for (int i = 0; i < 2; i++)
      for (int j = 0; j < 3; j++) {
        double sa = 0, sb = 0;
        for (int k = 0; k < 3; k++) sa += reduce[i][k] * A(k, j), sb += reduce[i][k] * B(k, j);
        jac20[i * 20 + col0 + j] = sa;
        jac20[i * 20 + col0 + 3 + j] = sb;
      }
  };
  auto redv = [&](V3 v, int i) { return reduce[i][0] * v.x + reduce[i][1] * v.y + reduce[i][2] * v.z; };
  const M3 ricT = transpose(ric), RjT = transpose(Rj);
  red(ricT * RjT, ricT * RjT * Ri * (-skew(pts_imu_i)), 0);
  red(ricT * (-RjT), ricT * skew(pts_imu_j), 6);
  {
    const M3 A = ricT * (RjT * Ri - M3::identity());
    const M3 tmp_r = ricT * RjT * Ri * ric;
    const M3 B = (-tmp_r) * skew(pts_camera_i) + skew(tmp_r * pts_camera_i) + skew(ricT * (RjT * (Ri * tic + Pi - Pj) - tic));
    red(A, B, 12);
  }
  const M3 chain = ricT * RjT * Ri * ric;
  const V3 vf = chain * pts_i_td, vt = chain * velocity_i;
  for (int i = 0; i < 2; i++) {
    jac20[i * 20 + 18] = redv(vf, i) * -1.0 / (inv_dep_i * inv_dep_i);
    jac20[i * 20 + 19] = redv(vt, i) / inv_dep_i * -1.0 + sqrt_info_s * velocity_j[i];
  }
}

// ProjectionFactor::Evaluate (projection_factor.cpp:21-121), UNIT_SPHERE_ERROR off.
// jac[0],jac[1],jac[2]: 2x7 row-major; jac[3]: 2x1.
inline void projection_factor_evaluate(V3 pts_i, V3 pts_j, double sqrt_info_s, const double* pose_i, const double* pose_j,
                                       const double* ex_pose, double inv_dep_i, double* residuals, double* jac[4]) {
  V3 Pi, Pj, tic;
  Q Qi, Qj, qic;
  getPose(pose_i, Pi, Qi);
  getPose(pose_j, Pj, Qj);
  getPose(ex_pose, tic, qic);
  V3 pts_camera_i = pts_i / inv_dep_i;
  V3 pts_imu_i = rot(qic, pts_camera_i) + tic;
  V3 pts_w = rot(Qi, pts_imu_i) + Pi;
  V3 pts_imu_j = rot(inverse(Qj), pts_w - Pj);
  V3 pts_camera_j = rot(inverse(qic), pts_imu_j - tic);
  double dep_j = pts_camera_j.z;
  double r0 = (pts_camera_j.x / dep_j) - pts_j.x;
  double r1 = (pts_camera_j.y / dep_j) - pts_j.y;
  residuals[0] = sqrt_info_s * r0;  // sqrt_info = s * I2 (estimator.cpp:17)
  residuals[1] = sqrt_info_s * r1;
  if (!jac) return;
  M3 Ri = toR(Qi), Rj = toR(Qj), ric = toR(qic);
  double reduce[2][3] = {{1. / dep_j, 0, -pts_camera_j.x / (dep_j * dep_j)}, {0, 1. / dep_j, -pts_camera_j.y / (dep_j * dep_j)}};
  for (int i = 0; i < 2; i++)
    for (int j = 0; j < 3; j++) reduce[i][j] = sqrt_info_s * reduce[i][j];
  auto red = [&](const M3& A, const M3& B, double* out, int stride) {
    // out(2 x 7 row-major).leftCols<6>() = reduce * [A B]; col 7 zero
    for (int i = 0; i < 2; i++) {
      for (int j = 0; j < 3; j++) {
        double sa = 0, sb = 0;
        for (int k = 0; k < 3; k++) sa += reduce[i][k] * A(k, j), sb += reduce[i][k] * B(k, j);
        out[i * stride + j] = sa;
        out[i * stride + 3 + j] = sb;
      }
      out[i * stride + 6] = 0.0;
    }
  };
  M3 ricT = transpose(ric), RjT = transpose(Rj);
  if (jac[0]) {
    M3 A = ricT * RjT;
    M3 B = ricT * RjT * Ri * (-skew(pts_imu_i));
    red(A, B, jac[0], 7);
  }
  if (jac[1]) {
    M3 A = ricT * (-RjT);
    M3 B = ricT * skew(pts_imu_j);
    red(A, B, jac[1], 7);
  }
  if (jac[2]) {
    M3 A = ricT * (RjT * Ri - M3::identity());
    M3 tmp_r = ricT * RjT * Ri * ric;
    M3 B = (-tmp_r) * skew(pts_camera_i) + skew(tmp_r * pts_camera_i) + skew(ricT * (RjT * (Ri * tic + Pi - Pj) - tic));
    red(A, B, jac[2], 7);
  }
  if (jac[3]) {
    V3 v = (ricT * RjT * Ri * ric) * pts_i;
    for (int i = 0; i < 2; i++) {
      double s = reduce[i][0] * v.x + reduce[i][1] * v.y + reduce[i][2] * v.z;
      jac[3][i] = s * -1.0 / (inv_dep_i * inv_dep_i);
    }
  }
}

// ceres::CauchyLoss(a)::Evaluate
inline void cauchy_loss(double a, double s, double rho[3]) {
  const double b = a * a, c = 1.0 / b;
  const double sum = 1.0 + s * c;
  const double inv = 1.0 / sum;
  rho[0] = b * std::log(sum);
  rho[1] = std::max(AVMO_NUM_MIN, inv);
  rho[2] = -c * (inv * inv);
}
// Ceres Corrector == marginalization_factor.cpp:37-68
struct Corrector {
  double sqrt_rho1, residual_scaling, alpha_sq_norm;
  Corrector(double sq_norm, const double rho[3]) {
    sqrt_rho1 = std::sqrt(rho[1]);
    if ((sq_norm == 0.0) || (rho[2] <= 0.0)) {
      residual_scaling = sqrt_rho1;
      alpha_sq_norm = 0.0;
    } else {
      const double D = 1.0 + 2.0 * sq_norm * rho[2] / rho[1];
      const double alpha = 1.0 - std::sqrt(D);
      residual_scaling = sqrt_rho1 / (1 - alpha);
      alpha_sq_norm = alpha / sq_norm;
    }
  }
  // J <- sqrt_rho1 * (J - alpha_sq_norm * r * (r^T J)) ; J is nres x ncols row-major
  void correctJacobian(int nres, int ncols, const double* r, double* J) const {
    if (alpha_sq_norm == 0.0) {
      for (int i = 0; i < nres * ncols; i++) J[i] *= sqrt_rho1;
      return;
    }
    for (int c = 0; c < ncols; c++) {
      double rtj = 0;
      for (int k = 0; k < nres; k++) rtj += r[k] * J[k * ncols + c];
      for (int k = 0; k < nres; k++) J[k * ncols + c] = sqrt_rho1 * (J[k * ncols + c] - alpha_sq_norm * r[k] * rtj);
    }
  }
  void correctResiduals(int nres, double* r) const {
    for (int i = 0; i < nres; i++) r[i] *= residual_scaling;
  }
};

// last_marginalization_info as plain data (marginalization_factor.h:46-72 + keep_block_*)
struct Prior {
  int n = 0;
  std::vector<int> blk_kind, blk_frame, blk_idx;  // blk_idx: local offset of the block inside dx (keep_block_idx - m)
  std::vector<std::vector<double>> x0;            // keep_block_data
  Mat J;                                          // linearized_jacobians n x n
  std::vector<double> r;                          // linearized_residuals
  static int gsize(int kind) { return kind == 1 ? 9 : (kind == 3 ? 1 : 7); }  // AVM_BLK_TD: one value
  static int lsize(int kind) { return kind == 1 ? 9 : (kind == 3 ? 1 : 6); }
};

// MarginalizationFactor::Evaluate (marginalization_factor.cpp:333-381); params[i] -> current block i
inline void prior_dx(const Prior& pr, const std::vector<const double*>& params, std::vector<double>& dx) {
  dx.assign(pr.n, 0.0);
  for (size_t i = 0; i < pr.blk_kind.size(); i++) {
    int size = Prior::gsize(pr.blk_kind[i]);
    int idx = pr.blk_idx[i];
    const double* x = params[i];
    const double* x0 = pr.x0[i].data();
    if (size != 7) {
      for (int k = 0; k < size; k++) dx[idx + k] = x[k] - x0[k];
    } else {
      for (int k = 0; k < 3; k++) dx[idx + k] = x[k] - x0[k];
      Q q0(x0[6], x0[3], x0[4], x0[5]), q(x[6], x[3], x[4], x[5]);
      Q dq = inverse(q0) * q;
      V3 v = 2.0 * dq.vec();
      if (!(dq.w >= 0)) v = 2.0 * (-dq.vec());
      for (int k = 0; k < 3; k++) dx[idx + 3 + k] = v[k];
    }
  }
}
inline void prior_residual(const Prior& pr, const std::vector<double>& dx, double* res) {
  for (int i = 0; i < pr.n; i++) {
    double s = 0;
    for (int k = 0; k < pr.n; k++) s += pr.J(i, k) * dx[k];
    res[i] = pr.r[i] + s;
  }
}

}  // namespace avmo
