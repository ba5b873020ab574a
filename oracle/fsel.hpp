// oracle/fsel.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// CPU restatement of FeatureSelector::select() (vins_estimator/src/feature_selector.cpp:74-202)
// for the initialized (NON_LINEAR) branch, with HORIZON a runtime parameter:
//   calcInfoFromRobotMotion :463-527, createLinearImuMatrices :531-598, addOmegaPrior :602-609,
//   calcInfoFromFeatures :239-365, inFOV :369-376, findNNDepth :437-459 (the reference's kd-tree, nanoflann, restated: KdIndex),
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

// The kd-tree behind findNNDepth: nanoflann::KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<double, PointCloud>, PointCloud, 2>
// with leaf_max_size 10, as initKDTree builds it (feature_selector.cpp:424-429; vendored header
// vins_estimator/lib/nanoflann/nanoflann.hpp).  Restated because WHICH of several equidistant cloud points a query gets
// is decided by the order in which the tree's traversal meets them:
//   buildIndex :1189-1201, computeBoundingBox :1309-1337, divideTree :857-907, middleSplit_ :909-958, planeSplit :969-1005,
//   computeMinMax :835-848, findNeighbors :1222-1243, computeInitialDistances :1007-1026, searchLevel :1346-1405,
//   L2_Simple_Adaptor::evalMetric / accum_dist :432-445, KNNResultSet<double>(1)::addPoint :175-202 (strict compare).
// Pinned by the outputs of that header itself: tests/golden/nanoflann_nn.npz (tests/test_nanoflann_nn.py).
struct KdIndex {
  struct Interval { double low, high; };
  struct Node {
    int child1 = -1, child2 = -1;  // both -1: a leaf over vind[left, right)
    size_t left = 0, right = 0;
    int divfeat = 0;
    double divlow = 0, divhigh = 0;
  };
  const std::vector<double>&X, &Y;
  std::vector<size_t> vind;
  std::vector<Node> nodes;
  Interval root_bbox[2];
  int root = -1;
  static constexpr size_t LEAF_MAX = 10;  // KDTreeSingleIndexAdaptorParams(10), feature_selector.cpp:428

  double pt(size_t idx, int dim) const { return dim == 0 ? X[idx] : Y[idx]; }  // PointCloud::kdtree_get_pt, feature_selector.h:130-134

  KdIndex(const std::vector<double>& x, const std::vector<double>& y) : X(x), Y(y) {
    const size_t N = X.size();
    vind.resize(N);
    for (size_t i = 0; i < N; i++) vind[i] = i;  // init_vind
    if (N == 0) return;
    for (int i = 0; i < 2; i++) root_bbox[i].low = root_bbox[i].high = pt(0, i);  // computeBoundingBox
    for (size_t k = 1; k < N; k++)
      for (int i = 0; i < 2; i++) {
        if (pt(k, i) < root_bbox[i].low) root_bbox[i].low = pt(k, i);
        if (pt(k, i) > root_bbox[i].high) root_bbox[i].high = pt(k, i);
      }
    root = divideTree(0, N, root_bbox);
  }

  void computeMinMax(const size_t* ind, size_t count, int element, double& min_elem, double& max_elem) const {
    min_elem = max_elem = pt(ind[0], element);
    for (size_t i = 1; i < count; i++) {
      const double val = pt(ind[i], element);
      if (val < min_elem) min_elem = val;
      if (val > max_elem) max_elem = val;
    }
  }

  // on return: [0, lim1) < cutval, [lim1, lim2) == cutval, [lim2, count) > cutval
  void planeSplit(size_t* ind, size_t count, int cutfeat, double cutval, size_t& lim1, size_t& lim2) const {
    size_t left = 0, right = count - 1;
    for (;;) {
      while (left <= right && pt(ind[left], cutfeat) < cutval) ++left;
      while (right && left <= right && pt(ind[right], cutfeat) >= cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left, --right;
    }
    lim1 = left;
    right = count - 1;
    for (;;) {
      while (left <= right && pt(ind[left], cutfeat) <= cutval) ++left;
      while (right && left <= right && pt(ind[right], cutfeat) > cutval) --right;
      if (left > right || !right) break;
      std::swap(ind[left], ind[right]);
      ++left, --right;
    }
    lim2 = left;
  }

  void middleSplit(size_t* ind, size_t count, size_t& index, int& cutfeat, double& cutval, const Interval* bbox) const {
    const double EPS = 0.00001;
    double max_span = bbox[0].high - bbox[0].low;
    for (int i = 1; i < 2; i++) {
      const double span = bbox[i].high - bbox[i].low;
      if (span > max_span) max_span = span;
    }
    double max_spread = -1;
    cutfeat = 0;
    for (int i = 0; i < 2; i++) {
      const double span = bbox[i].high - bbox[i].low;
      if (span > (1 - EPS) * max_span) {
        double mn, mx;
        computeMinMax(ind, count, i, mn, mx);
        const double spread = mx - mn;
        if (spread > max_spread) cutfeat = i, max_spread = spread;
      }
    }
    const double split_val = (bbox[cutfeat].low + bbox[cutfeat].high) / 2;
    double mn, mx;
    computeMinMax(ind, count, cutfeat, mn, mx);
    cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
    size_t lim1, lim2;
    planeSplit(ind, count, cutfeat, cutval, lim1, lim2);
    index = lim1 > count / 2 ? lim1 : (lim2 < count / 2 ? lim2 : count / 2);
  }

  int divideTree(size_t left, size_t right, Interval* bbox) {
    const int me = (int)nodes.size();
    nodes.emplace_back();
    if (right - left <= LEAF_MAX) {
      nodes[me].left = left, nodes[me].right = right;
      for (int i = 0; i < 2; i++) bbox[i].low = bbox[i].high = pt(vind[left], i);
      for (size_t k = left + 1; k < right; k++)
        for (int i = 0; i < 2; i++) {
          if (bbox[i].low > pt(vind[k], i)) bbox[i].low = pt(vind[k], i);
          if (bbox[i].high < pt(vind[k], i)) bbox[i].high = pt(vind[k], i);
        }
    } else {
      size_t idx;
      int cutfeat;
      double cutval;
      middleSplit(&vind[0] + left, right - left, idx, cutfeat, cutval, bbox);
      Interval lb[2] = {bbox[0], bbox[1]}, rb[2] = {bbox[0], bbox[1]};
      lb[cutfeat].high = cutval;
      const int c1 = divideTree(left, left + idx, lb);
      rb[cutfeat].low = cutval;
      const int c2 = divideTree(left + idx, right, rb);
      Node& n = nodes[me];
      n.child1 = c1, n.child2 = c2, n.divfeat = cutfeat;
      n.divlow = lb[cutfeat].high, n.divhigh = rb[cutfeat].low;
      for (int i = 0; i < 2; i++) {
        bbox[i].low = std::min(lb[i].low, rb[i].low);
        bbox[i].high = std::max(lb[i].high, rb[i].high);
      }
    }
    return me;
  }

  // KNNResultSet<double>(1): one slot, replaced only by a strictly smaller distance
  struct Result {
    size_t index = 0;
    double dist = AVMO_NUM_MAX;
    size_t count = 0;
    void addPoint(double d, size_t i) {
      if (count == 0 || dist > d) dist = d, index = i;
      count = 1;
    }
  };

  void searchLevel(Result& res, const double* vec, int ni, double mindistsq, double* dists) const {
    const Node& node = nodes[ni];
    if (node.child1 < 0 && node.child2 < 0) {
      const double worst_dist = res.dist;  // (read once per leaf)
      for (size_t i = node.left; i < node.right; i++) {
        const size_t index = vind[i];
        double dist = 0;
        for (int d = 0; d < 2; d++) {
          const double diff = vec[d] - pt(index, d);
          dist += diff * diff;
        }
        if (dist < worst_dist) res.addPoint(dist, index);
      }
      return;
    }
    const int idx = node.divfeat;
    const double val = vec[idx];
    const double diff1 = val - node.divlow, diff2 = val - node.divhigh;
    int best, other;
    double cut_dist;
    if (diff1 + diff2 < 0)
      best = node.child1, other = node.child2, cut_dist = (val - node.divhigh) * (val - node.divhigh);
    else
      best = node.child2, other = node.child1, cut_dist = (val - node.divlow) * (val - node.divlow);
    searchLevel(res, vec, best, mindistsq, dists);
    const double dst = dists[idx];
    mindistsq = mindistsq + cut_dist - dst;
    dists[idx] = cut_dist;
    if (mindistsq * 1.0f <= res.dist) searchLevel(res, vec, other, mindistsq, dists);  // epsError = 1 + SearchParams(10).eps (0)
    dists[idx] = dst;
  }

  size_t nearest(double x, double y) const {  // findNeighbors
    const double vec[2] = {x, y};
    double dists[2] = {0, 0}, distsq = 0;
    for (int i = 0; i < 2; i++) {  // computeInitialDistances
      if (vec[i] < root_bbox[i].low) dists[i] = (vec[i] - root_bbox[i].low) * (vec[i] - root_bbox[i].low), distsq += dists[i];
      if (vec[i] > root_bbox[i].high) dists[i] = (vec[i] - root_bbox[i].high) * (vec[i] - root_bbox[i].high), distsq += dists[i];
    }
    Result res;
    searchLevel(res, vec, root, distsq, dists);
    return res.index;
  }
};

// findNNDepth, feature_selector.cpp:437-459: the depth of the cloud point nanoflann's exact 1-NN search returns
// (ret_index starts at 0 and stays there when nothing beats the initial worst distance, e.g. NaN coordinates)
inline double findNNDepth(const FselProblem& p, const KdIndex& kd, double x, double y) {
  if (p.cloud_d.empty()) return 1.0;
  return p.cloud_d[kd.nearest(x, y)];
}
inline double findNNDepth(const FselProblem& p, double x, double y) {
  KdIndex kd(p.cloud_x, p.cloud_y);
  return findNNDepth(p, kd, x, y);
}

// calcInfoFromFeatures, feature_selector.cpp:239-365. Returns dense Delta_ell per id (only for
// features that can be triangulated, numVisible > 1).
inline std::map<int, Mat> calcInfoFromFeatures(const FselProblem& p, const std::vector<int>& ids, const std::vector<double>& xs,
                                               const std::vector<double>& ys) {
  std::map<int, Mat> out;
  const int H = p.H, N = 9 * (H + 1);
  V3 t_WC_k1 = p.pos[1] + rot(p.quat[1], p.t_IC);
  Q q_WC_k1 = p.quat[1] * p.q_IC;
  const KdIndex kd(p.cloud_x, p.cloud_y);  // initKDTree (feature_selector.cpp:380-432): built once per frame
  for (size_t f = 0; f < ids.size(); f++) {
    V3 feature(xs[f], ys[f], 1.0);
    double d = findNNDepth(p, kd, feature.x, feature.y);
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
  std::vector<double> min_gap;  // (want_gap only) fMax of the round minus the best fValue among the other candidates of the round's std::map
  long n_logdet = 0;
};

// selectInformativeFeatures + sortedlogDetUB, feature_selector.cpp:613-728.
// want_gap (not in the reference; the checker of avm_fsel_out::min_gap): after the round's own loop, every candidate of the round's map that
// the lazy `ub < fMax` break left unscored is scored as well, and the runner-up's value is recorded.  Selection and fValues are untouched.
inline FselResult fsel_select(const FselProblem& p, bool want_gap = false) {
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
    if (want_gap && lMax > -1) {
      double runner = -AVMO_NUM_INF;
      for (auto& up : UBs) {
        if (up.second == lMax) continue;
        const Mat& D = Delta_ells.at(up.second);
        double pr = prob.at(up.second);
        Mat A(N, N);
        for (size_t i = 0; i < A.a.size(); i++) A.a[i] = Omega.a[i] + OmegaS.a[i] + pr * D.a[i];
        double fValue = logdet_chol(A);
        if (fValue > -1.0 && fValue > runner) runner = fValue;
      }
      R.min_gap.push_back(fMax - runner);
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
