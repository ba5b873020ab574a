// oracle/triangulate.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// FeatureManager::triangulate (vins_estimator/src/feature_manager.cpp:202-257): linear multi-view triangulation of
// every feature that has no depth yet, in the camera frame of its first observation.
// Eigen::JacobiSVD is restated as a one-sided (Hestenes) Jacobi SVD of the (2 nobs) x 4 matrix: like Eigen's
// two-sided Jacobi it delivers the right singular vectors to working precision; only the LAST right singular vector
// (smallest singular value) is used, and only through the ratio V[2] / V[3], so the sign ambiguity drops out.
// PARITY UNPINNED (no reference fixtures); pinned in tests/ against numpy.linalg.svd on the same matrices.
#pragma once
#include "linalg.hpp"

namespace avmo {

// right singular vector of the smallest singular value of the n x 4 matrix A (row-major), n >= 4
inline void smallest_right_singular_vector(std::vector<double> A, int n, double v[4]) {
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
    for (int p = 0; p < 3; p++)
      for (int q = p + 1; q < 4; q++) {
        double app = 0, aqq = 0, apq = 0;
        for (int i = 0; i < n; i++) {
          const double x = A[i * 4 + p], y = A[i * 4 + q];
          app += x * x, aqq += y * y, apq += x * y;
        }
        if (std::fabs(apq) <= 1e-300 || std::fabs(apq) <= 2.3e-16 * std::sqrt(app * aqq)) continue;
        rotated = true;
        const double tau = (aqq - app) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int i = 0; i < n; i++) {
          const double x = A[i * 4 + p], y = A[i * 4 + q];
          A[i * 4 + p] = c * x - s * y, A[i * 4 + q] = s * x + c * y;
        }
        for (int i = 0; i < 4; i++) {
          const double x = V[i][p], y = V[i][q];
          V[i][p] = c * x - s * y, V[i][q] = s * x + c * y;
        }
      }
    if (!rotated) break;
  }
  int best = 0;
  double bn = AVMO_NUM_INF;
  for (int j = 0; j < 4; j++) {
    double s = 0;
    for (int i = 0; i < n; i++) s += A[i * 4 + j] * A[i * 4 + j];
    if (s < bn) bn = s, best = j;
  }
  for (int i = 0; i < 4; i++) v[i] = V[i][best];
}

// depth of one feature: poses[f] = (p, q) of the body frames, (tic, qic) the camera extrinsic, obs[k] the normalized
// image points of the nobs observations starting at frame `start`.  feature_manager.cpp:209-249
inline double triangulate_feature(const double (*pose)[7], const double* ex, int start, int nobs, const double* obs_xy, double init_depth) {
  const V3 tic(ex[0], ex[1], ex[2]);
  const M3 ric = toR(Q(ex[6], ex[3], ex[4], ex[5]));
  auto Rs = [&](int f) { return toR(Q(pose[f][6], pose[f][3], pose[f][4], pose[f][5])); };
  auto Ps = [&](int f) { return V3(pose[f][0], pose[f][1], pose[f][2]); };
  const V3 t0 = Ps(start) + Rs(start) * tic;
  const M3 R0 = Rs(start) * ric;
  std::vector<double> A((size_t)2 * nobs * 4);
  for (int k = 0; k < nobs; k++) {
    const int j = start + k;
    const V3 t1 = Ps(j) + Rs(j) * tic;
    const M3 R1 = Rs(j) * ric;
    const V3 t = transpose(R0) * (t1 - t0);
    const M3 R = transpose(R0) * R1;
    const M3 Rt = transpose(R);
    const V3 mt = -(Rt * t);
    double P[3][4];
    for (int a = 0; a < 3; a++) {
      for (int b = 0; b < 3; b++) P[a][b] = Rt(a, b);
      P[a][3] = mt[a];
    }
    const V3 f = normalized(V3(obs_xy[2 * k], obs_xy[2 * k + 1], 1.0));
    for (int b = 0; b < 4; b++) {
      A[(size_t)(2 * k) * 4 + b] = f[0] * P[2][b] - f[2] * P[0][b];
      A[(size_t)(2 * k + 1) * 4 + b] = f[1] * P[2][b] - f[2] * P[1][b];
    }
  }
  double v[4];
  smallest_right_singular_vector(A, 2 * nobs, v);
  double depth = v[2] / v[3];
  if (!(depth >= 0.1)) depth = init_depth;  // `estimated_depth < 0.1` -> INIT_DEPTH (a NaN ratio also falls back)
  return depth;
}

// Estimator::processIMU, the dead-reckoning of the newest frame (estimator.cpp:100-107): world-frame midpoint
// integration of the raw samples of the last interval, starting from the state slideWindow() left in frame j
// (a copy of the previous newest frame).  Rs is a MATRIX that is multiplied by the rotation matrix of the
// unnormalized deltaQ (no re-orthonormalization); the quaternion is only formed afterwards (vector2double,
// estimator.cpp:486).  acc/gyr: row 0 = the sample the interval's IntegrationBase was constructed with (acc_0).
inline void propagate_newest_frame(double pose[7], double sb[9], int n, const double* dt, const double* acc, const double* gyr, V3 g) {
  V3 P(pose[0], pose[1], pose[2]), V(sb[0], sb[1], sb[2]);
  const V3 Ba(sb[3], sb[4], sb[5]), Bg(sb[6], sb[7], sb[8]);
  M3 R = toR(Q(pose[6], pose[3], pose[4], pose[5]));
  V3 acc0(acc[0], acc[1], acc[2]), gyr0(gyr[0], gyr[1], gyr[2]);
  for (int s = 0; s < n; s++) {
    const V3 a1(acc[3 * (s + 1)], acc[3 * (s + 1) + 1], acc[3 * (s + 1) + 2]), w1(gyr[3 * (s + 1)], gyr[3 * (s + 1) + 1], gyr[3 * (s + 1) + 2]);
    const double h = dt[s];
    const V3 un_acc_0 = R * (acc0 - Ba) - g;
    const V3 un_gyr = 0.5 * (gyr0 + w1) - Bg;
    R = R * toR(deltaQ(un_gyr * h));
    const V3 un_acc_1 = R * (a1 - Ba) - g;
    const V3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    P = P + h * V + (0.5 * h * h) * un_acc;
    V = V + h * un_acc;
    acc0 = a1, gyr0 = w1;
  }
  const Q q = fromR(R);
  pose[0] = P.x, pose[1] = P.y, pose[2] = P.z, pose[3] = q.x, pose[4] = q.y, pose[5] = q.z, pose[6] = q.w;
  sb[0] = V.x, sb[1] = V.y, sb[2] = V.z;
}

}  // namespace avmo
