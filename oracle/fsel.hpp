// oracle/fsel.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// CPU restatement of FeatureSelector::select() (vins_estimator/src/feature_selector.cpp:74-202)
// for the initialized (NON_LINEAR) branch, with HORIZON a runtime parameter:
//   calcInfoFromRobotMotion :463-527, createLinearImuMatrices :531-598, addOmegaPrior :602-609,
//   calcInfoFromFeatures :239-365, inFOV :369-376, findNNDepth :437-459 (exact 1-NN, brute force),
//   selectInformativeFeatures :613-686, sortedlogDetUB :690-728, Utility::logdet utility.h:144-167,
//   PinholeCamera::spaceToPlane / distortion camera_model/src/camera_models/PinholeCamera.cc:520-542,646-662.
// Bug-compatibility kept on purpose: q_IC applied twice in Bh, no z>0
// check before projection, std::round, UB-key collisions in the std::map, fMax = -1.0, strict >.
// PARITY UNPINNED: no reference tests/golden vectors exist for this path.
#pragma once
#include <map>

#include "../include/avm.h"
#include "linalg.hpp"

namespace avmo {

struct FselCamera {
  double fx, fy, cx, cy, k1, k2, p1, p2;
  int width, height;
  void spaceToPlane(V3 P, double& u, double& v) const {
    double xu = P.x / P.z, yu = P.y / P.z;
    double mx2 = xu * xu, my2 = yu * yu, mxy = xu * yu, rho2 = mx2 + my2;
    double rad = k1 * rho2 + k2 * rho2 * rho2;
    double dx = xu * rad + 2.0 * p1 * mxy + p2 * (rho2 + 2.0 * mx2);
    double dy = yu * rad + 2.0 * p2 * mxy + p1 * (rho2 + 2.0 * my2);
    u = fx * (xu + dx) + cx;
    v = fy * (yu + dy) + cy;
  }
  bool inFOV(double pu, double pv) const {  // feature_selector.cpp:369-376
    // NaN or out-of-range doubles convert to INT_MIN on the reference's platform (x86 cvttsd2si): outside the image.
    // Stated explicitly so that the restatement does not depend on the conversion's undefined behaviour.
    if (!(std::fabs(pu) < 2147483647.0 && std::fabs(pv) < 2147483647.0)) return false;
    int u = (int)std::round(pu), v = (int)std::round(pv);
    return (0 <= u && u < width) && (0 <= v && v < height);
  }
};

struct FselProblem {
  int H;
  std::vector<V3> pos;  // H+1
  std::vector<Q> quat;  // H+1
  int nrImu;
  double deltaImu, accVar, accBiasVar;
  Q q_IC;
  V3 t_IC;
  FselCamera cam;
  std::vector<int> cand_id;
  std::vector<double> cand_x, cand_y, cand_p;
  std::vector<int> used_id;
  std::vector<double> used_x, used_y;
  std::vector<double> cloud_x, cloud_y, cloud_d;
  int maxFeatures;
};

// createLinearImuMatrices, feature_selector.cpp:531-598
inline void createLinearImuMatrices(const Q& Qi, const Q& Qj, double nrImu, double deltaImu, double accVar, double accBiasVar,
                                    Mat& Omega, Mat& Ablk) {
  M3 Nij, Mij;
  double CCt_11 = 0, CCt_12 = 0;
  for (int i = 0; i < nrImu; ++i) {
    Q q = slerp(Qi, i / nrImu, Qj);
    double jkh = (nrImu - i - 0.5);
    M3 R = toR(q);
    Nij = Nij + jkh * R;
    Mij = Mij + R;
    CCt_11 += jkh * jkh;
    CCt_12 += jkh;
  }
  const double d2 = deltaImu * deltaImu, d3 = d2 * deltaImu, d4 = d3 * deltaImu;
  Mat cov(9, 9);
  for (int i = 0; i < 3; i++) {
    cov(i, i) = 1.0 * nrImu * CCt_11 * d4 * accVar;
    cov(i, 3 + i) = 1.0 * CCt_12 * d3 * accVar;
    cov(3 + i, i) = cov(i, 3 + i);
    cov(3 + i, 3 + i) = 1.0 * nrImu * d2 * accVar;
    cov(6 + i, 6 + i) = 1.0 * nrImu * accBiasVar;
  }
  Nij = Nij * d2;
  Mij = Mij * deltaImu;
  Ablk = Mat(9, 9);
  for (int i = 0; i < 9; i++) Ablk(i, i) = -1.0;
  for (int i = 0; i < 3; i++) Ablk(i, 3 + i) = -1.0 * nrImu * deltaImu;
  Ablk.setBlock(0, 6, Nij);
  Ablk.setBlock(3, 6, Mij);
  Omega = inverse_lu(cov);
}

// calcInfoFromRobotMotion + addOmegaPrior, feature_selector.cpp:463-527,602-609
inline Mat calcInfoFromRobotMotion(const FselProblem& p) {
  const int N = 9 * (p.H + 1);
  Mat Om(N, N);
  for (int h = 1; h <= p.H; ++h) {
    Mat W, A;
    createLinearImuMatrices(p.quat[h - 1], p.quat[h], p.nrImu, p.deltaImu, p.accVar, p.accBiasVar, W, A);
    Mat At = transpose(A);
    Mat tmp = matmul(At, W);   // At*Omega
    Mat b1 = matmul(tmp, A);   // At*Omega*A
    for (int i = 0; i < 9; i++)
      for (int j = 0; j < 9; j++) {
        Om((h - 1) * 9 + i, (h - 1) * 9 + j) += b1(i, j);
        Om((h - 1) * 9 + i, h * 9 + j) += tmp(i, j);
        Om(h * 9 + i, (h - 1) * 9 + j) += tmp(j, i);
        Om(h * 9 + i, h * 9 + j) += W(i, j);
      }
  }
  for (int i = 0; i < 9; i++) Om(i, i) += 1.0;  // addOmegaPrior
  return Om;
}

// findNNDepth, feature_selector.cpp:437-459 (nanoflann exact 1-NN, L2_Simple; first strictly smaller wins)
inline double findNNDepth(const FselProblem& p, double x, double y) {
  if (p.cloud_d.empty()) return 1.0;
  size_t best = 0;
  double bd = AVMO_NUM_MAX;
  for (size_t i = 0; i < p.cloud_d.size(); i++) {
    double dx = x - p.cloud_x[i], dy = y - p.cloud_y[i];
    double d = dx * dx + dy * dy;
    if (d < bd) bd = d, best = i;
  }
  return p.cloud_d[best];
}

// calcInfoFromFeatures, feature_selector.cpp:239-365. Returns dense Delta_ell per id (only for
// features that can be triangulated, numVisible > 1).
inline std::map<int, Mat> calcInfoFromFeatures(const FselProblem& p, const std::vector<int>& ids, const std::vector<double>& xs,
                                               const std::vector<double>& ys) {
  std::map<int, Mat> out;
  const int H = p.H, N = 9 * (H + 1);
  V3 t_WC_k1 = p.pos[1] + rot(p.quat[1], p.t_IC);
  Q q_WC_k1 = p.quat[1] * p.q_IC;
  for (size_t f = 0; f < ids.size(); f++) {
    V3 feature(xs[f], ys[f], 1.0);
    double d = findNNDepth(p, feature.x, feature.y);
    feature = normalized(feature) * d;
    V3 pell = t_WC_k1 + rot(q_WC_k1, feature);
    int numVisible = 1;
    std::vector<M3> Ch(H);
    M3 EtE;
    for (int h = 2; h <= H; ++h) {
      V3 t_WC_h = p.pos[h] + rot(p.quat[h], p.t_IC);
      Q q_WC_h = p.quat[h] * p.q_IC;
      V3 uell = normalized(rot(inverse(q_WC_h), pell - t_WC_h));
      double pu, pv;
      p.cam.spaceToPlane(uell, pu, pv);
      if (!p.cam.inFOV(pu, pv)) continue;
      M3 Bh = skew(uell) * toR(inverse(q_WC_h * p.q_IC));  // q_IC twice: bug-compatible (:304)
      Ch[h - 1] = transpose(Bh) * Bh;
      EtE = EtE + Ch[h - 1];
      ++numVisible;
    }
    if (numVisible == 1) continue;
    M3 Bh = skew(normalized(feature)) * toR(inverse(q_WC_k1 * p.q_IC));
    Ch[0] = transpose(Bh) * Bh;
    EtE = EtE + Ch[0];
    M3 W = inverse3(EtE);
    Mat D(N, N);
    for (int j = 1; j <= H; ++j)
      for (int i = j; i <= H; ++i) {
        M3 Dij = Ch[i - 1] * W * transpose(Ch[j - 1]);
        if (i == j) {
          D.setBlock(9 * i, 9 * j, Ch[i - 1] - Dij);
        } else {
          D.setBlock(9 * i, 9 * j, -Dij);
          D.setBlock(9 * j, 9 * i, -transpose(Dij));
        }
      }
    out[ids[f]] = D;
  }
  return out;
}

// Utility::logdet(M, true), utility.h:144-167 — NaN on a failed factorisation (documented)
inline double logdet_chol(const Mat& M) {
  Mat L = M;
  if (!llt_lower(L)) return AVMO_NUM_NAN;
  double ld = 0;
  for (int i = 0; i < M.r; i++) ld += std::log(L(i, i));
  return ld * 2;
}

struct FselResult {
  std::vector<int> selected;
  std::vector<double> fvalues;
  long n_logdet = 0;
};

// selectInformativeFeatures + sortedlogDetUB, feature_selector.cpp:613-728
inline FselResult fsel_select(const FselProblem& p) {
  FselResult R;
  const int N = 9 * (p.H + 1);
  Mat Omega = calcInfoFromRobotMotion(p);
  std::map<int, Mat> Delta_ells = calcInfoFromFeatures(p, p.cand_id, p.cand_x, p.cand_y);
  std::map<int, Mat> Delta_used = calcInfoFromFeatures(p, p.used_id, p.used_x, p.used_y);
  std::map<int, double> prob;
  for (size_t i = 0; i < p.cand_id.size(); i++) prob[p.cand_id[i]] = p.cand_p[i];
  int kappa = std::max(0, p.maxFeatures - (int)p.used_id.size());
  for (auto& d : Delta_used)
    for (size_t i = 0; i < Omega.a.size(); i++) Omega.a[i] += d.second.a[i];
  std::vector<int> blacklist;
  Mat OmegaS(N, N);
  for (int it = 0; it < kappa; ++it) {
    // sortedlogDetUB
    std::map<double, int, std::greater<double>> UBs;
    Mat M(N, N);
    for (size_t i = 0; i < M.a.size(); i++) M.a[i] = Omega.a[i] + OmegaS.a[i];
    for (auto& fp : Delta_ells) {
      int id = fp.first;
      if (std::find(blacklist.begin(), blacklist.end(), id) != blacklist.end()) continue;
      double pr = prob.at(id);
      double ub = 0;
      for (int i = 0; i < N; i++) ub += std::log(M(i, i) + pr * fp.second(i, i));
      UBs[ub] = id;
    }
    double fMax = -1.0;
    int lMax = -1;
    for (auto& up : UBs) {
      int id = up.second;
      double ub = up.first;
      if (ub < fMax) break;
      const Mat& D = Delta_ells.at(id);
      double pr = prob.at(id);
      Mat A(N, N);
      for (size_t i = 0; i < A.a.size(); i++) A.a[i] = Omega.a[i] + OmegaS.a[i] + pr * D.a[i];
      double fValue = logdet_chol(A);
      R.n_logdet++;
      if (fValue > fMax) fMax = fValue, lMax = id;
    }
    if (lMax > -1) {
      double pr = prob.at(lMax);
      const Mat& D = Delta_ells.at(lMax);
      for (size_t i = 0; i < OmegaS.a.size(); i++) OmegaS.a[i] += pr * D.a[i];
      blacklist.push_back(lMax);
      R.fvalues.push_back(fMax);
    }
  }
  R.selected = blacklist;
  return R;
}

// HorizonGenerator::imu (utility/horizon_generator.cpp:25-69).  States 0 and 1 are given; 2..H are propagated with a
// constant body acceleration a and angular rate w; Qimu = deltaQ(w deltaImu) is unnormalized and the attitude is never
// renormalized inside the loop.  pos/quat: [H+1][3] / [H+1][4] (x y z w).
inline void horizon_imu(int H, const double* k_pos, const double* k_quat, const double* k_ba, const double* k1_pos, const double* k1_vel,
                        const double* k1_quat, const double* a_, const double* w_, int nrImu, double deltaImu, double* pos, double* quat) {
  const V3 gravity(0, 0, -9.80665);  // state_defs.h:37-41
  const V3 Ba(k_ba[0], k_ba[1], k_ba[2]), a(a_[0], a_[1], a_[2]), w(w_[0], w_[1], w_[2]);
  for (int k = 0; k < 3; k++) pos[k] = k_pos[k], pos[3 + k] = k1_pos[k];
  for (int k = 0; k < 4; k++) quat[k] = k_quat[k], quat[4 + k] = k1_quat[k];
  const Q Qimu = deltaQ(w * deltaImu);
  V3 p(k1_pos[0], k1_pos[1], k1_pos[2]), v(k1_vel[0], k1_vel[1], k1_vel[2]);
  Q q(k1_quat[3], k1_quat[0], k1_quat[1], k1_quat[2]);
  for (int h = 2; h <= H; h++) {
    for (int i = 0; i < nrImu; i++) {
      q = q * Qimu;
      const V3 qa = rot(q, a - Ba);
      v = v + (gravity + qa) * deltaImu;
      p = p + v * deltaImu + ((0.5 * gravity) * deltaImu) * deltaImu + ((0.5 * qa) * deltaImu) * deltaImu;
    }
    pos[3 * h] = p.x, pos[3 * h + 1] = p.y, pos[3 * h + 2] = p.z;
    quat[4 * h] = q.x, quat[4 * h + 1] = q.y, quat[4 * h + 2] = q.z, quat[4 * h + 3] = q.w;
  }
}

// FeatureSelector::initKDTree, the cloud construction (feature_selector.cpp:396-419) for one window.
// Returns the number of cloud points; xy [max_cloud][2], depth [max_cloud].
inline int build_cloud(const double (*pose)[7], const double* ex, int nf, const int* start, const int* obs_begin, const double* obs_xy,
                       const double* inv_depth, const double* k1_pos, const double* k1_quat, int max_cloud, double* xy, double* depth) {
  const V3 tic(ex[0], ex[1], ex[2]);
  const Q qic(ex[6], ex[3], ex[4], ex[5]);
  const M3 ric = toR(qic);
  const Q qk1(k1_quat[3], k1_quat[0], k1_quat[1], k1_quat[2]);
  const V3 pk1(k1_pos[0], k1_pos[1], k1_pos[2]);
  int n = 0;
  for (int e = 0; e < nf && n < max_cloud; e++) {
    if (start[e] > 10 * 3.0 / 4.0) continue;         // start_frame > WINDOW_SIZE * 3.0 / 4.0
    const double est_depth = 1.0 / inv_depth[e];
    if (!(est_depth >= 0)) continue;                // solve_flag != 1 (depth < 0 -> 2; NaN never qualifies)
    const int f = start[e];
    const M3 Rs = toR(Q(pose[f][6], pose[f][3], pose[f][4], pose[f][5]));
    const V3 Ps(pose[f][0], pose[f][1], pose[f][2]);
    const V3 pts_i = V3(obs_xy[2 * obs_begin[e]], obs_xy[2 * obs_begin[e] + 1], 1.0) * est_depth;
    const V3 w_pts_i = Rs * (ric * pts_i + tic) + Ps;
    const V3 p_IL = rot(inverse(qk1), w_pts_i - pk1);
    const V3 p_CL = rot(inverse(qic), p_IL - tic);
    const V3 nip = p_CL / p_CL.z;
    xy[2 * n] = nip.x, xy[2 * n + 1] = nip.y, depth[n] = est_depth;
    n++;
  }
  return n;
}

}  // namespace avmo
