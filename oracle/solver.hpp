// oracle/solver.hpp — TEST INFRASTRUCTURE ONLY (CPU oracle). Never linked into the product.
//
// CPU restatement of Estimator::optimization() (vins_estimator/src/estimator.cpp:661-994):
// problem assembly (:663-755), what ceres::Solve does for this problem with
// DENSE_SCHUR + DOGLEG (:794-809; Ceres is NOT vendored in /root/reference and not
// version-pinned — vins_estimator/CMakeLists.txt:23 — so this restates the published
// Ceres 1.14 algorithm: TrustRegionMinimizer, DoglegStrategy(TRADITIONAL_DOGLEG),
// SchurEliminator + dense LLT, ResidualBlock/Corrector),
// double2vector/vector2double (:477-610) and the post-solve marginalization
// (:817-990 with factor/marginalization_factor.cpp:89-319).
// PARITY UNPINNED: no reference tests/golden vectors exist for this path and the
// reference cannot be built in this image (no ROS/Ceres/Eigen/OpenCV).
#pragma once
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../include/avm.h"
#include "factors.hpp"

namespace avmo {

struct State {
  double pose[AVM_NFRAMES][7];
  double sb[AVM_NFRAMES][9];
  double ex[7];
  std::vector<double> lam;
  double td = 0.0;                          // para_Td[0][0]
  double relo[7] = {0, 0, 0, 0, 0, 0, 1};   // relo_Pose
};

struct Window {
  State x;
  int nf = 0;
  std::vector<int> start, nobs, obs_begin;
  std::vector<double> obs_xy;
  std::vector<PreIntegration> pre;
  std::vector<Mat> sqrt_info;
  bool has_prior = false;
  Prior prior;
  // optional members of the problem (avm_window_batch)
  std::vector<double> obs_aux;  // [obs slot][4] velocity.x, velocity.y, cur_td, uv.y  (estimate_td)
  int relo_n = 0;               // relocalization_info: matched features (estimator.cpp:760-792)
  bool has_relo = false;        // relocalization_info itself (relo_Pose is gauge-fixed even when no feature matched, :588-596)
  int relo_frame = 0;
  std::vector<int> relo_feat;
  std::vector<double> relo_xy;
  bool failure_occur = false;   // estimator.cpp:526-531
  double last_pose0[7] = {0, 0, 0, 0, 0, 0, 1};
};

// parameter block ids: pose f -> f ; speedbias f -> 11+f ; ex_pose -> 22 ; td -> 23 ; relo_Pose -> 24 ; feature e -> 25+e
enum { ID_SB0 = AVM_NFRAMES, ID_EX = 2 * AVM_NFRAMES, ID_TD = 2 * AVM_NFRAMES + 1, ID_RELO = 2 * AVM_NFRAMES + 2, ID_FEAT0 = 2 * AVM_NFRAMES + 3 };

struct RBlock {
  int type;  // 0 prior, 1 imu, 2 projection (aux1 < 0: relocalization factor -aux1 - 1), 3 projection with td
  int nres;
  int nb;
  int ids[16];
  int lsz[16];   // local size (0 if constant in the solve)
  int coff[16];  // column offset inside this block's row-major Jacobian
  int ncols;
  int aux0, aux1;  // imu: interval ; proj: feature, obs slot
  std::vector<double> r, J;
};

struct Problem {
  const Window* w;
  const avm_options* opt;
  int n_local;             // reduced tangent dimension
  int n_f;                 // f-block (pose/sb/[ex]) dimension
  std::vector<int> loff;   // id -> local offset, -1 constant
  std::vector<int> lsize;  // id -> local size
  std::vector<RBlock> blocks;

  void build(const Window& win, const avm_options& o) {
    w = &win;
    opt = &o;
    int nid = ID_FEAT0 + win.nf;
    loff.assign(nid, -1);
    lsize.assign(nid, 0);
    int off = 0;
    for (int f = 0; f < AVM_NFRAMES; f++) loff[f] = off, lsize[f] = 6, off += 6;
    for (int f = 0; f < AVM_NFRAMES; f++) loff[ID_SB0 + f] = off, lsize[ID_SB0 + f] = 9, off += 9;
    if (o.estimate_extrinsic) loff[ID_EX] = off, lsize[ID_EX] = 6, off += 6;
    if (o.estimate_td) loff[ID_TD] = off, lsize[ID_TD] = 1, off += 1;                 // estimator.cpp:684-688
    if (win.relo_n > 0) loff[ID_RELO] = off, lsize[ID_RELO] = 6, off += 6;           // estimator.cpp:763-764
    n_f = off;
    for (int e = 0; e < win.nf; e++) loff[ID_FEAT0 + e] = off, lsize[ID_FEAT0 + e] = 1, off += 1;
    n_local = off;
    blocks.clear();
    auto finish = [&](RBlock& b) {
      int c = 0;
      for (int k = 0; k < b.nb; k++) {
        b.lsz[k] = loff[b.ids[k]] >= 0 ? lsize[b.ids[k]] : 0;
        b.coff[k] = c;
        c += b.lsz[k];
      }
      b.ncols = c;
      b.r.assign(b.nres, 0.0);
      b.J.assign((size_t)b.nres * c, 0.0);
    };
    // prior (estimator.cpp:694-700)
    if (win.has_prior && win.prior.n > 0) {
      RBlock b;
      b.type = 0, b.nres = win.prior.n, b.nb = (int)win.prior.blk_kind.size();
      for (int k = 0; k < b.nb; k++) {
        int kind = win.prior.blk_kind[k], fr = win.prior.blk_frame[k];
        b.ids[k] = kind == AVM_BLK_POSE ? fr : (kind == AVM_BLK_SPEEDBIAS ? ID_SB0 + fr : (kind == AVM_BLK_TD ? (int)ID_TD : (int)ID_EX));
      }
      b.aux0 = b.aux1 = 0;
      finish(b);
      blocks.push_back(b);
    }
    // IMU (estimator.cpp:702-709)
    for (int i = 0; i < AVM_WINDOW_SIZE; i++) {
      if (win.pre[i].sum_dt > o.max_sum_dt) continue;
      RBlock b;
      b.type = 1, b.nres = 15, b.nb = 4;
      b.ids[0] = i, b.ids[1] = ID_SB0 + i, b.ids[2] = i + 1, b.ids[3] = ID_SB0 + i + 1;
      b.aux0 = i, b.aux1 = 0;
      finish(b);
      blocks.push_back(b);
    }
    // vision (estimator.cpp:712-755)
    for (int e = 0; e < win.nf; e++) {
      for (int t = 1; t < win.nobs[e]; t++) {
        RBlock b;
        b.type = o.estimate_td ? 3 : 2, b.nres = 2, b.nb = o.estimate_td ? 5 : 4;
        b.ids[0] = win.start[e], b.ids[1] = win.start[e] + t, b.ids[2] = ID_EX, b.ids[3] = ID_FEAT0 + e, b.ids[4] = ID_TD;
        b.aux0 = e, b.aux1 = win.obs_begin[e] + t;
        finish(b);
        blocks.push_back(b);
      }
    }
    // relocalization (estimator.cpp:760-792): plain ProjectionFactors between the start pose of a matched feature and relo_Pose
    for (int k = 0; k < win.relo_n; k++) {
      const int e = win.relo_feat[k];
      RBlock b;
      b.type = 2, b.nres = 2, b.nb = 4;
      b.ids[0] = win.start[e], b.ids[1] = ID_RELO, b.ids[2] = ID_EX, b.ids[3] = ID_FEAT0 + e;
      b.aux0 = e, b.aux1 = -(k + 1);
      finish(b);
      blocks.push_back(b);
    }
  }

  const double* param(const State& x, int id) const {
    if (id < ID_SB0) return x.pose[id];
    if (id < ID_EX) return x.sb[id - ID_SB0];
    if (id == ID_EX) return x.ex;
    if (id == ID_TD) return &x.td;
    if (id == ID_RELO) return x.relo;
    return &x.lam[id - ID_FEAT0];
  }

  // ceres ResidualBlock::Evaluate + ProgramEvaluator: cost = sum 1/2 rho(|r|^2); with jacobians the
  // blocks hold corrected residuals and corrected local Jacobians.
  // A cost-only evaluation (the candidate point) must leave the stored residuals / Jacobians of the current point alone:
  // Ceres keeps residuals_ and jacobian_ at x and evaluates the candidate into nothing (TrustRegionMinimizer::
  // ComputeCandidatePointAndEvaluateCost passes NULL for residuals), and model_cost_change of a retried step after a
  // rejection is computed from residuals_ at x.
  double evaluate(const State& x, bool want_jac) {
    const Window& win = *w;
    double cost = 0.0;
    const double sq = opt->focal_length / 1.5;
    V3 G(opt->g[0], opt->g[1], opt->g[2]);
    std::vector<double> scratch_r;
    for (auto& blk : blocks) {
      RBlock& b = blk;
      double* br = b.r.data();
      if (!want_jac) {
        scratch_r.assign(b.nres, 0.0);
        br = scratch_r.data();
      }
      if (b.type == 0) {
        std::vector<const double*> ps(b.nb);
        for (int k = 0; k < b.nb; k++) ps[k] = param(x, b.ids[k]);
        std::vector<double> dx;
        prior_dx(win.prior, ps, dx);
        prior_residual(win.prior, dx, br);
        if (want_jac) {
          for (int k = 0; k < b.nb; k++) {
            if (!b.lsz[k]) continue;
            int idx = win.prior.blk_idx[k];
            for (int i = 0; i < b.nres; i++)
              for (int c = 0; c < b.lsz[k]; c++) b.J[(size_t)i * b.ncols + b.coff[k] + c] = win.prior.J(i, idx + c);
          }
        }
        double s = 0;
        for (int i = 0; i < b.nres; i++) s += br[i] * br[i];
        cost += 0.5 * s;
      } else if (b.type == 1) {
        int i = b.aux0;
        double j0[15 * 7], j1[15 * 9], j2[15 * 7], j3[15 * 9];
        double* jac[4] = {j0, j1, j2, j3};
        imu_factor_evaluate(win.pre[i], win.sqrt_info[i], G, x.pose[i], x.sb[i], x.pose[i + 1], x.sb[i + 1], br,
                            want_jac ? jac : nullptr);
        if (want_jac) {
          const int gs[4] = {7, 9, 7, 9};
          for (int k = 0; k < 4; k++)
            for (int r = 0; r < 15; r++)
              for (int c = 0; c < b.lsz[k]; c++) b.J[(size_t)r * b.ncols + b.coff[k] + c] = jac[k][r * gs[k] + c];
        }
        double s = 0;
        for (int r = 0; r < 15; r++) s += br[r] * br[r];
        cost += 0.5 * s;
      } else {
        int e = b.aux0, slot = b.aux1, s0 = win.obs_begin[e];
        if (b.type == 3) {
          // ProjectionTdFactor (estimator.cpp:732-747): velocity / cur_td / uv.y of the two observations
          const double* ai = &win.obs_aux[4 * (size_t)s0];
          const double* aj = &win.obs_aux[4 * (size_t)slot];
          double j20[40];
          projection_td_factor_evaluate(&win.obs_xy[2 * s0], &win.obs_xy[2 * slot], ai, aj, ai[2], aj[2], ai[3], aj[3], opt->tr, opt->row, sq,
                                        x.pose[b.ids[0]], x.pose[b.ids[1]], x.ex, x.lam[e], x.td, br, want_jac ? j20 : nullptr);
          if (want_jac) {
            const int src[5] = {0, 6, 12, 18, 19};  // pose_i | pose_j | ex_pose | inv depth | td
            for (int k = 0; k < 5; k++)
              for (int r = 0; r < 2; r++)
                for (int c = 0; c < b.lsz[k]; c++) b.J[(size_t)r * b.ncols + b.coff[k] + c] = j20[r * 20 + src[k] + c];
          }
        } else {
          V3 pts_i(win.obs_xy[2 * s0], win.obs_xy[2 * s0 + 1], 1.0);
          V3 pts_j = slot >= 0 ? V3(win.obs_xy[2 * slot], win.obs_xy[2 * slot + 1], 1.0)
                               : V3(win.relo_xy[2 * (-slot - 1)], win.relo_xy[2 * (-slot - 1) + 1], 1.0);  // match_points (estimator.cpp:781)
          double j0[14], j1[14], j2[14], j3[2];
          double* jac[4] = {j0, j1, j2, j3};
          projection_factor_evaluate(pts_i, pts_j, sq, x.pose[b.ids[0]], param(x, b.ids[1]), x.ex, x.lam[e], br, want_jac ? jac : nullptr);
          if (want_jac) {
            const int gs[4] = {7, 7, 7, 1};
            for (int k = 0; k < 4; k++)
              for (int r = 0; r < 2; r++)
                for (int c = 0; c < b.lsz[k]; c++) b.J[(size_t)r * b.ncols + b.coff[k] + c] = jac[k][r * gs[k] + c];
          }
        }
        double sn = br[0] * br[0] + br[1] * br[1];
        double rho[3];
        cauchy_loss(opt->cauchy_a, sn, rho);
        cost += 0.5 * rho[0];
        if (want_jac) {
          Corrector corr(sn, rho);
          corr.correctJacobian(2, b.ncols, br, b.J.data());
          corr.correctResiduals(2, br);
        }
      }
    }
    return cost;
  }

  void gradient(std::vector<double>& g) const {  // g = J^T r (unscaled J)
    g.assign(n_local, 0.0);
    for (auto& b : blocks)
      for (int k = 0; k < b.nb; k++) {
        if (!b.lsz[k]) continue;
        int o = loff[b.ids[k]];
        for (int r = 0; r < b.nres; r++)
          for (int c = 0; c < b.lsz[k]; c++) g[o + c] += b.J[(size_t)r * b.ncols + b.coff[k] + c] * b.r[r];
      }
  }
  void sqColNorm(std::vector<double>& n2) const {
    n2.assign(n_local, 0.0);
    for (auto& b : blocks)
      for (int k = 0; k < b.nb; k++) {
        if (!b.lsz[k]) continue;
        int o = loff[b.ids[k]];
        for (int r = 0; r < b.nres; r++)
          for (int c = 0; c < b.lsz[k]; c++) {
            double v = b.J[(size_t)r * b.ncols + b.coff[k] + c];
            n2[o + c] += v * v;
          }
      }
  }
  void scaleColumns(const std::vector<double>& s) {
    for (auto& b : blocks)
      for (int k = 0; k < b.nb; k++) {
        if (!b.lsz[k]) continue;
        int o = loff[b.ids[k]];
        for (int r = 0; r < b.nres; r++)
          for (int c = 0; c < b.lsz[k]; c++) b.J[(size_t)r * b.ncols + b.coff[k] + c] *= s[o + c];
      }
  }
  // y = J x, returned per block concatenated; also returns sum over rows of y.(r + y/2)
  void rightMultiply(const std::vector<double>& x, std::vector<std::vector<double>>& y) const {
    y.resize(blocks.size());
    for (size_t bi = 0; bi < blocks.size(); bi++) {
      auto& b = blocks[bi];
      y[bi].assign(b.nres, 0.0);
      for (int k = 0; k < b.nb; k++) {
        if (!b.lsz[k]) continue;
        int o = loff[b.ids[k]];
        for (int r = 0; r < b.nres; r++) {
          double s = 0;
          for (int c = 0; c < b.lsz[k]; c++) s += b.J[(size_t)r * b.ncols + b.coff[k] + c] * x[o + c];
          y[bi][r] += s;
        }
      }
    }
  }
  void leftMultiply(std::vector<double>& g) const {  // g += J^T r with current (scaled) J
    for (auto& b : blocks)
      for (int k = 0; k < b.nb; k++) {
        if (!b.lsz[k]) continue;
        int o = loff[b.ids[k]];
        for (int r = 0; r < b.nres; r++)
          for (int c = 0; c < b.lsz[k]; c++) g[o + c] += b.J[(size_t)r * b.ncols + b.coff[k] + c] * b.r[r];
      }
  }

  // SchurComplementSolver (DENSE_SCHUR): minimise |J y - r|^2 + |D y|^2, e-blocks = inverse depths.
  // Returns false on Cholesky failure (LINEAR_SOLVER_FAILURE).
  bool schurSolve(const std::vector<double>& D, std::vector<double>& y) const {
    const int nf_ = n_f, ne = n_local - n_f;
    Mat lhs(nf_, nf_);
    std::vector<double> rhs(nf_, 0.0);
    for (int i = 0; i < nf_; i++) lhs(i, i) = D[i] * D[i];
    std::vector<double> ete(ne, 0.0), ge(ne, 0.0);
    Mat buf(ne, nf_);  // E^T F per feature (dense rows; only the touched pose columns are non-zero)
    for (int e = 0; e < ne; e++) ete[e] = D[nf_ + e] * D[nf_ + e];
    for (auto& b : blocks) {
      int ek = -1;  // index of e-block inside this residual block
      for (int k = 0; k < b.nb; k++)
        if (b.lsz[k] && loff[b.ids[k]] >= nf_) ek = k;
      // F^T F and F^T b
      for (int k = 0; k < b.nb; k++) {
        if (!b.lsz[k] || k == ek) continue;
        int ok = loff[b.ids[k]];
        for (int l = 0; l < b.nb; l++) {
          if (!b.lsz[l] || l == ek) continue;
          int ol = loff[b.ids[l]];
          for (int r = 0; r < b.nres; r++) {
            const double* row = &b.J[(size_t)r * b.ncols];
            for (int c = 0; c < b.lsz[k]; c++) {
              double v = row[b.coff[k] + c];
              if (v == 0.0) continue;
              for (int d = 0; d < b.lsz[l]; d++) lhs(ok + c, ol + d) += v * row[b.coff[l] + d];
            }
          }
        }
        for (int r = 0; r < b.nres; r++)
          for (int c = 0; c < b.lsz[k]; c++) rhs[ok + c] += b.J[(size_t)r * b.ncols + b.coff[k] + c] * b.r[r];
      }
      if (ek >= 0) {
        int e = loff[b.ids[ek]] - nf_;
        for (int r = 0; r < b.nres; r++) {
          const double* row = &b.J[(size_t)r * b.ncols];
          double je = row[b.coff[ek]];
          ete[e] += je * je;
          ge[e] += je * b.r[r];
          for (int k = 0; k < b.nb; k++) {
            if (!b.lsz[k] || k == ek) continue;
            int ok = loff[b.ids[k]];
            for (int c = 0; c < b.lsz[k]; c++) buf(e, ok + c) += je * row[b.coff[k] + c];
          }
        }
      }
    }
    // lhs -= buf^T ete^-1 buf ; rhs -= buf^T ete^-1 ge   (only pose columns can be non-zero)
    const int npose = AVM_NFRAMES * 6;
    std::vector<int> cols;
    if (fast_linalg()) {
      // (bench leg: the same updates over the dense pose range - a zero column subtracts an exact zero -, contiguous inner loop)
      for (int e = 0; e < ne; e++) {
        const double inv = 1.0 / ete[e];
        const double* be = &buf.a[(size_t)e * nf_];
        int c0 = nf_, c1 = -1;  // the feature's range of touched columns (its frames' poses; ex_pose / td / relo_Pose when they are variables)
        for (int c = 0; c < nf_; c++)
          if (be[c] != 0.0) c0 = c < c0 ? c : c0, c1 = c;
        (void)npose;
        for (int a = c0; a <= c1; a++) {
          if (be[a] == 0.0) continue;
          const double va = be[a] * inv;
          double* la = &lhs.a[(size_t)a * nf_];
          for (int c = c0; c <= c1; c++) la[c] -= va * be[c];
          rhs[a] -= va * ge[e];
        }
      }
    } else {
    for (int e = 0; e < ne; e++) {
      double inv = 1.0 / ete[e];
      cols.clear();
      for (int c = 0; c < nf_; c++)
        if (buf(e, c) != 0.0) cols.push_back(c);
      (void)npose;
      for (int a : cols) {
        double va = buf(e, a) * inv;
        for (int c : cols) lhs(a, c) -= va * buf(e, c);
        rhs[a] -= va * ge[e];
      }
    }
    }
    Mat L = lhs;
    std::vector<double> yf = rhs;
    if (fast_linalg()) {
      Mat U;
      if (!llt_lower_fast(L, U)) return false;
      llt_solve_fast(U, yf);
    } else {
      if (!llt_lower(L)) return false;
      llt_solve(L, yf);
    }
    y.assign(n_local, 0.0);
    for (int i = 0; i < nf_; i++) y[i] = yf[i];
    for (int e = 0; e < ne; e++) {
      double s = ge[e];
      for (int c = 0; c < nf_; c++)
        if (buf(e, c) != 0.0) s -= buf(e, c) * yf[c];
      y[nf_ + e] = s / ete[e];
    }
    for (int i = 0; i < n_local; i++)
      if (!std::isfinite(y[i])) return false;
    return true;
  }

  // Evaluator::Plus with PoseLocalParameterization::Plus (pose_local_parameterization.cpp:3-19)
  void plus(const State& x, const std::vector<double>& d, State& out) const {
    out = x;
    for (int f = 0; f < AVM_NFRAMES; f++) {
      int o = loff[f];
      if (o < 0) continue;
      for (int k = 0; k < 3; k++) out.pose[f][k] = x.pose[f][k] + d[o + k];
      Q q(x.pose[f][6], x.pose[f][3], x.pose[f][4], x.pose[f][5]);
      Q dq = deltaQ(V3(d[o + 3], d[o + 4], d[o + 5]));
      Q r = normalized(q * dq);
      out.pose[f][3] = r.x, out.pose[f][4] = r.y, out.pose[f][5] = r.z, out.pose[f][6] = r.w;
    }
    for (int f = 0; f < AVM_NFRAMES; f++) {
      int o = loff[ID_SB0 + f];
      if (o < 0) continue;
      for (int k = 0; k < 9; k++) out.sb[f][k] = x.sb[f][k] + d[o + k];
    }
    if (loff[ID_EX] >= 0) {
      int o = loff[ID_EX];
      for (int k = 0; k < 3; k++) out.ex[k] = x.ex[k] + d[o + k];
      Q q(x.ex[6], x.ex[3], x.ex[4], x.ex[5]);
      Q r = normalized(q * deltaQ(V3(d[o + 3], d[o + 4], d[o + 5])));
      out.ex[3] = r.x, out.ex[4] = r.y, out.ex[5] = r.z, out.ex[6] = r.w;
    }
    if (loff[ID_TD] >= 0) out.td = x.td + d[loff[ID_TD]];
    if (loff[ID_RELO] >= 0) {
      int o = loff[ID_RELO];
      for (int k = 0; k < 3; k++) out.relo[k] = x.relo[k] + d[o + k];
      Q q(x.relo[6], x.relo[3], x.relo[4], x.relo[5]);
      Q r = normalized(q * deltaQ(V3(d[o + 3], d[o + 4], d[o + 5])));
      out.relo[3] = r.x, out.relo[4] = r.y, out.relo[5] = r.z, out.relo[6] = r.w;
    }
    for (int e = 0; e < w->nf; e++) out.lam[e] = x.lam[e] + d[loff[ID_FEAT0 + e]];
  }
  // ambient-space helpers over the non-constant blocks
  template <class F>
  void forAmbient(const State& a, const State& b, F f) const {
    for (int fr = 0; fr < AVM_NFRAMES; fr++)
      for (int k = 0; k < 7; k++) f(a.pose[fr][k], b.pose[fr][k]);
    for (int fr = 0; fr < AVM_NFRAMES; fr++)
      for (int k = 0; k < 9; k++) f(a.sb[fr][k], b.sb[fr][k]);
    if (loff[ID_EX] >= 0)
      for (int k = 0; k < 7; k++) f(a.ex[k], b.ex[k]);
    if (loff[ID_TD] >= 0) f(a.td, b.td);
    if (loff[ID_RELO] >= 0)
      for (int k = 0; k < 7; k++) f(a.relo[k], b.relo[k]);
    for (int e = 0; e < w->nf; e++) f(a.lam[e], b.lam[e]);
  }
};

// ---- Ceres 1.14 TrustRegionMinimizer + DoglegStrategy(TRADITIONAL_DOGLEG) -----------------
struct SolveResult {
  avm_solve_summary sum;
  State x;
  std::vector<double> cost_after;  // the cost after every iteration in the oracle's own scalar type (the summary's trace is the ABI's FP64)
};

inline SolveResult trust_region_solve(Problem& P, const State& x0) {
  const avm_options& o = *P.opt;
  SolveResult R;
  std::memset(&R.sum, 0, sizeof R.sum);
  const int n = P.n_local;
  State x = x0, cand = x0;
  std::vector<double> gradient, scale(n, 1.0), tmp;
  double x_cost = 0, x_norm = 0, cand_cost = 0;
  double gradient_max_norm = 0;

  auto ambNorm = [&](const State& a) {
    double s = 0;
    P.forAmbient(a, a, [&](double u, double) { s += u * u; });
    return std::sqrt(s);
  };
  auto ambDiff = [&](const State& a, const State& b, bool inf) {
    double s = 0;
    P.forAmbient(a, b, [&](double u, double v) {
      double d = u - v;
      if (inf)
        s = std::max(s, std::fabs(d));
      else
        s += d * d;
    });
    return inf ? s : std::sqrt(s);
  };
  bool first = true;
  // TrustRegionMinimizer::EvaluateGradientAndJacobian
  auto evalGradJac = [&]() {
    x_cost = P.evaluate(x, true);
    P.gradient(gradient);
    if (o.jacobi_scaling) {
      if (first) {
        P.sqColNorm(tmp);
        for (int i = 0; i < n; i++) scale[i] = 1.0 / (1.0 + std::sqrt(tmp[i]));
        first = false;
      }
      P.scaleColumns(scale);
    }
    std::vector<double> neg(n);
    for (int i = 0; i < n; i++) neg[i] = -gradient[i];
    State pg;
    P.plus(x, neg, pg);
    gradient_max_norm = ambDiff(x, pg, true);
  };

  // DoglegStrategy state
  double radius = o.initial_trust_region_radius;
  double mu = 1e-8;
  const double min_mu = 1e-8, max_mu = 1.0, mu_increase = 10.0;
  double dogleg_step_norm = 0, alpha = 0;
  bool reuse = false;
  std::vector<double> diagonal(n), dgrad(n), gn(n), step(n), delta(n);

  auto computeTraditionalDogleg = [&]() {
    double gradient_norm = 0, gn_norm = 0;
    for (int i = 0; i < n; i++) gradient_norm += dgrad[i] * dgrad[i], gn_norm += gn[i] * gn[i];
    gradient_norm = std::sqrt(gradient_norm), gn_norm = std::sqrt(gn_norm);
    if (gn_norm <= radius) {
      for (int i = 0; i < n; i++) step[i] = gn[i] / diagonal[i];
      dogleg_step_norm = gn_norm;
      return;
    }
    if (gradient_norm * alpha >= radius) {
      for (int i = 0; i < n; i++) step[i] = (-(radius / gradient_norm) * dgrad[i]) / diagonal[i];
      dogleg_step_norm = radius;
      return;
    }
    double gdot = 0;
    for (int i = 0; i < n; i++) gdot += dgrad[i] * gn[i];
    const double b_dot_a = -alpha * gdot;
    const double a_squared_norm = std::pow(alpha * gradient_norm, 2.0);
    const double b_minus_a_squared_norm = a_squared_norm - 2 * b_dot_a + std::pow(gn_norm, 2);
    const double c = b_dot_a - a_squared_norm;
    const double d = std::sqrt(c * c + b_minus_a_squared_norm * (std::pow(radius, 2.0) - a_squared_norm));
    double beta = (c <= 0) ? (d - c) / b_minus_a_squared_norm : (radius * radius - a_squared_norm) / (d + c);
    double s2 = 0;
    for (int i = 0; i < n; i++) {
      step[i] = (-alpha * (1.0 - beta)) * dgrad[i] + beta * gn[i];
      s2 += step[i] * step[i];
    }
    dogleg_step_norm = std::sqrt(s2);
    for (int i = 0; i < n; i++) step[i] /= diagonal[i];
  };
  // returns 0 success, 1 LINEAR_SOLVER_FAILURE
  auto doglegComputeStep = [&]() -> int {
    if (reuse) {
      computeTraditionalDogleg();
      return 0;
    }
    reuse = true;
    P.sqColNorm(diagonal);
    for (int i = 0; i < n; i++) diagonal[i] = std::sqrt(std::min(std::max(diagonal[i], o.min_lm_diagonal), o.max_lm_diagonal));
    // ComputeGradient
    std::fill(dgrad.begin(), dgrad.end(), 0.0);
    P.leftMultiply(dgrad);
    for (int i = 0; i < n; i++) dgrad[i] /= diagonal[i];
    // ComputeCauchyPoint
    {
      std::vector<double> sg(n);
      for (int i = 0; i < n; i++) sg[i] = dgrad[i] / diagonal[i];
      std::vector<std::vector<double>> Jg;
      P.rightMultiply(sg, Jg);
      double g2 = 0, jg2 = 0;
      for (int i = 0; i < n; i++) g2 += dgrad[i] * dgrad[i];
      for (auto& v : Jg)
        for (double t : v) jg2 += t * t;
      alpha = g2 / jg2;
    }
    // ComputeGaussNewtonStep
    bool ok = false;
    while (mu < max_mu) {
      std::vector<double> lm(n);
      for (int i = 0; i < n; i++) lm[i] = diagonal[i] * std::sqrt(mu);
      if (!P.schurSolve(lm, gn)) {
        mu *= mu_increase;
        continue;
      }
      ok = true;
      break;
    }
    if (!ok) return 1;
    for (int i = 0; i < n; i++) gn[i] *= -diagonal[i];
    computeTraditionalDogleg();
    return 0;
  };

  // ---- Minimize -----------------------------------------------------------------------
  const auto t_minimize_start = std::chrono::steady_clock::now();
  x_norm = ambNorm(x);
  evalGradJac();  // IterationZero
  R.sum.initial_cost = x_cost;
  bool step_is_successful = true;  // iteration 0
  int iteration = 0;
  int num_consecutive_invalid = 0;
  int termination = AVM_TERM_NO_CONVERGENCE;
  double ref_cost = x_cost;  // TrustRegionStepEvaluator (monotonic): current == reference cost
  double tr_radius_report = radius;
  while (true) {
    // FinalizeIterationAndCheckIfMinimizerCanContinue
    if (step_is_successful && iteration > 0) R.sum.num_successful++;
    tr_radius_report = radius;
    if (iteration > 0 && iteration <= AVM_MAX_ITER_TRACE) {
      R.sum.cost_trace[iteration - 1] = x_cost;
      R.cost_after.push_back(x_cost);
      R.sum.radius_trace[iteration - 1] = tr_radius_report;
      if (step_is_successful) R.sum.accept_mask |= (1 << (iteration - 1));
    }
    // MaxSolverTimeReached comes first (trust_region_minimizer.cc, FinalizeIterationAndCheckIfMinimizerCanContinue);
    // options.max_solver_time_in_seconds = SOLVER_TIME (x 4/5 under MARGIN_OLD), estimator.cpp:803-806; 0 = no cap here
    if (o.max_solver_time_s > 0.0 &&
        std::chrono::duration<double>(std::chrono::steady_clock::now() - t_minimize_start).count() >= o.max_solver_time_s) {
      termination = AVM_TERM_NO_CONVERGENCE;
      break;
    }
    if (iteration >= o.max_num_iterations) {
      termination = AVM_TERM_NO_CONVERGENCE;
      break;
    }
    if (step_is_successful && gradient_max_norm <= o.gradient_tolerance) {
      termination = AVM_TERM_GRADIENT_TOL;
      break;
    }
    if (tr_radius_report <= o.min_trust_region_radius) {
      termination = AVM_TERM_MIN_RADIUS;
      break;
    }
    iteration++;
    step_is_successful = false;
    // ComputeTrustRegionStep
    bool step_is_valid = false;
    double model_cost_change = 0;
    int ls = doglegComputeStep();
    if (ls == 0) {
      std::vector<std::vector<double>> mr;
      P.rightMultiply(step, mr);
      double s = 0;
      for (size_t bi = 0; bi < mr.size(); bi++)
        for (int r = 0; r < P.blocks[bi].nres; r++) s += mr[bi][r] * (P.blocks[bi].r[r] + mr[bi][r] / 2.0);
      model_cost_change = -s;
      if (std::getenv("AVMO_TRACE")) {
        // the same quantity through the identity the GPU kernel uses: -step^T g - 1/2 step^T H step with H y = g - mu D^2 y
        double sg = 0, gg = 0;
        for (int i = 0; i < n; i++) sg += step[i] * dgrad[i] * diagonal[i], gg += dgrad[i] * dgrad[i];
        double q = 0;
        for (size_t bi = 0; bi < mr.size(); bi++)
          for (int r = 0; r < P.blocks[bi].nres; r++) q += mr[bi][r] * mr[bi][r];
        std::fprintf(stderr, "[oracle it %d] radius %.10g mu %.3g alpha %.10g |g/D| %.10g model_cost_change %.12g (-s^T g %.12g, 1/2 |J s|^2 %.12g) step_norm %.10g\n",
                     iteration, radius, mu, alpha, std::sqrt(gg), model_cost_change, -sg, 0.5 * q, dogleg_step_norm);
      }
      step_is_valid = model_cost_change > 0.0;
      if (step_is_valid) {
        for (int i = 0; i < n; i++) delta[i] = step[i] * scale[i];
        num_consecutive_invalid = 0;
      }
    }
    if (!step_is_valid) {
      // HandleInvalidStep
      if (++num_consecutive_invalid >= o.max_num_consecutive_invalid_steps) {
        termination = AVM_TERM_FAILURE;
        break;
      }
      mu *= mu_increase;  // DoglegStrategy::StepIsInvalid
      reuse = false;
      continue;
    }
    // ComputeCandidatePointAndEvaluateCost
    P.plus(x, delta, cand);
    cand_cost = P.evaluate(cand, false);
    // ParameterToleranceReached
    double step_norm = ambDiff(x, cand, false);
    if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
      termination = AVM_TERM_PARAMETER_TOL;
      break;
    }
    // FunctionToleranceReached
    double cost_change = x_cost - cand_cost;
    if (std::fabs(cost_change) <= o.function_tolerance * x_cost) {
      termination = AVM_TERM_FUNCTION_TOL;
      break;
    }
    // IsStepSuccessful (monotonic TrustRegionStepEvaluator)
    double relative_decrease = (ref_cost - cand_cost) / model_cost_change;
    if (std::getenv("AVMO_TRACE")) std::fprintf(stderr, "[oracle it %d] x_cost %.12g cand_cost %.12g rho %.12g\n", iteration, x_cost, cand_cost, relative_decrease);
    if (relative_decrease > o.min_relative_decrease) {
      // HandleSuccessfulStep
      x = cand;
      x_norm = ambNorm(x);
      evalGradJac();
      step_is_successful = true;
      // DoglegStrategy::StepAccepted
      if (relative_decrease < 0.25) radius *= 0.5;
      if (relative_decrease > 0.75) radius = std::max(radius, 3.0 * dogleg_step_norm);
      mu = std::max(min_mu, 2.0 * mu / mu_increase);
      reuse = false;
      ref_cost = cand_cost;
    } else {
      // HandleUnsuccessfulStep -> DoglegStrategy::StepRejected
      radius *= 0.5;
      reuse = true;
    }
  }
  R.sum.termination = termination;
  R.sum.num_iterations = iteration;
  R.sum.final_cost = x_cost;
  R.x = x;
  return R;
}

// ---- Utility::R2ypr / ypr2R (utility/utility.h:66-108), degrees ---------------------------
inline V3 R2ypr(const M3& R) {
  V3 n = R.col(0), o = R.col(1), a = R.col(2);
  double y = std::atan2(n.y, n.x);
  double p = std::atan2(-n.z, n.x * std::cos(y) + n.y * std::sin(y));
  double r = std::atan2(a.x * std::sin(y) - a.y * std::cos(y), -o.x * std::sin(y) + o.y * std::cos(y));
  return V3(y / M_PI * 180.0, p / M_PI * 180.0, r / M_PI * 180.0);
}
inline M3 ypr2R(V3 ypr) {
  double y = ypr.x / 180.0 * M_PI, p = ypr.y / 180.0 * M_PI, r = ypr.z / 180.0 * M_PI;
  M3 Rz, Ry, Rx;
  Rz(0, 0) = std::cos(y), Rz(0, 1) = -std::sin(y), Rz(1, 0) = std::sin(y), Rz(1, 1) = std::cos(y), Rz(2, 2) = 1;
  Ry(0, 0) = std::cos(p), Ry(0, 2) = std::sin(p), Ry(1, 1) = 1, Ry(2, 0) = -std::sin(p), Ry(2, 2) = std::cos(p);
  Rx(0, 0) = 1, Rx(1, 1) = std::cos(r), Rx(1, 2) = -std::sin(r), Rx(2, 1) = std::sin(r), Rx(2, 2) = std::cos(r);
  return Rz * Ry * Rx;
}

// Estimator::double2vector (estimator.cpp:521-587) followed by vector2double (:477-519):
// the state the host sees after optimization().  `before` = para_* before the solve
// (Rs[0]/Ps[0] are reconstructed from it), `sol` = para_* after ceres::Solve.
inline void gauge_fix_roundtrip(const State& before, const State& sol, State& out, const double* failure_anchor = nullptr,
                                bool relo = false) {
  M3 Rs0 = toR(Q(before.pose[0][6], before.pose[0][3], before.pose[0][4], before.pose[0][5]));
  V3 origin_R0 = R2ypr(Rs0);
  V3 origin_P0(before.pose[0][0], before.pose[0][1], before.pose[0][2]);
  if (failure_anchor) {  // failure_occur: last_R0 / last_P0 (estimator.cpp:526-531); Rs[0] itself stays what it was
    origin_R0 = R2ypr(toR(Q(failure_anchor[6], failure_anchor[3], failure_anchor[4], failure_anchor[5])));
    origin_P0 = V3(failure_anchor[0], failure_anchor[1], failure_anchor[2]);
  }
  M3 R00 = toR(Q(sol.pose[0][6], sol.pose[0][3], sol.pose[0][4], sol.pose[0][5]));
  V3 origin_R00 = R2ypr(R00);
  double y_diff = origin_R0.x - origin_R00.x;
  M3 rot_diff = ypr2R(V3(y_diff, 0, 0));
  if (std::fabs(std::fabs(origin_R0.y) - 90) < 1.0 || std::fabs(std::fabs(origin_R00.y) - 90) < 1.0)
    rot_diff = Rs0 * transpose(R00);
  out = sol;
  for (int i = 0; i < AVM_NFRAMES; i++) {
    Q q(sol.pose[i][6], sol.pose[i][3], sol.pose[i][4], sol.pose[i][5]);
    M3 Rsi = rot_diff * toR(normalized(q));
    V3 Psi = rot_diff * V3(sol.pose[i][0] - sol.pose[0][0], sol.pose[i][1] - sol.pose[0][1], sol.pose[i][2] - sol.pose[0][2]) + origin_P0;
    V3 Vsi = rot_diff * V3(sol.sb[i][0], sol.sb[i][1], sol.sb[i][2]);
    // vector2double
    Q qo = fromR(Rsi);
    out.pose[i][0] = Psi.x, out.pose[i][1] = Psi.y, out.pose[i][2] = Psi.z;
    out.pose[i][3] = qo.x, out.pose[i][4] = qo.y, out.pose[i][5] = qo.z, out.pose[i][6] = qo.w;
    out.sb[i][0] = Vsi.x, out.sb[i][1] = Vsi.y, out.sb[i][2] = Vsi.z;
  }
  {  // ric = q.toRotationMatrix() ; back: Quaterniond{ric}
    Q q(sol.ex[6], sol.ex[3], sol.ex[4], sol.ex[5]);
    Q qo = fromR(toR(q));
    out.ex[3] = qo.x, out.ex[4] = qo.y, out.ex[5] = qo.z, out.ex[6] = qo.w;
  }
  // setDepth: estimated_depth = 1/x ; getDepthVector: 1/estimated_depth
  for (size_t e = 0; e < sol.lam.size(); e++) out.lam[e] = 1.0 / (1.0 / sol.lam[e]);
  if (relo) {  // relo_r / relo_t of estimator.cpp:590-596 (the pose-graph outputs :597-604 are host arithmetic on them)
    Q q(sol.relo[6], sol.relo[3], sol.relo[4], sol.relo[5]);
    M3 relo_r = rot_diff * toR(normalized(q));
    V3 relo_t = rot_diff * V3(sol.relo[0] - sol.pose[0][0], sol.relo[1] - sol.pose[0][1], sol.relo[2] - sol.pose[0][2]) + origin_P0;
    Q qo = fromR(relo_r);
    out.relo[0] = relo_t.x, out.relo[1] = relo_t.y, out.relo[2] = relo_t.z;
    out.relo[3] = qo.x, out.relo[4] = qo.y, out.relo[5] = qo.z, out.relo[6] = qo.w;
  }
}

// ---- MarginalizationInfo (marginalization_factor.cpp:89-319) -------------------------------
// Deterministic block order (documented deviation from the address-hash order, SURVEY App.A #10):
// dropped: pose, speedbias, features in feature order; kept: poses by frame, speedbias by frame, ex_pose.
struct MFactor {
  int nres;
  std::vector<int> ids;
  std::vector<int> gs;             // global sizes
  std::vector<std::vector<double>> J;  // nres x gs row-major (full 7-col pose jacobians)
  std::vector<double> r;
  std::vector<int> drop;
};

// what the eigenvalue clamps of marginalization_factor.cpp:272,284-285 saw (diagnostics for the precision arbiter, tests/)
struct MargDiag {
  int m = 0, n = 0;
  std::vector<double> ev_mm, ev_rr;  // eigenvalues of Amm and of the Schur complement, ascending
  Mat A_rr;                          // the Schur complement A (n x n) and b before the square root
  std::vector<double> b_rr;
};

inline void marginalize(const Window& win, const State& x, const avm_options& o, Prior& out, MargDiag* diag = nullptr) {
  const int flag = o.marginalization_flag;
  std::vector<MFactor> factors;
  const double sq = o.focal_length / 1.5;
  V3 G(o.g[0], o.g[1], o.g[2]);
  auto idOfPrior = [&](int k) {
    int kind = win.prior.blk_kind[k], fr = win.prior.blk_frame[k];
    return kind == AVM_BLK_POSE ? fr : (kind == AVM_BLK_SPEEDBIAS ? ID_SB0 + fr : (kind == AVM_BLK_TD ? (int)ID_TD : (int)ID_EX));
  };
  auto paramOf = [&](int id) -> const double* {
    if (id < ID_SB0) return x.pose[id];
    if (id < ID_EX) return x.sb[id - ID_SB0];
    if (id == ID_EX) return x.ex;
    if (id == ID_TD) return &x.td;
    return &x.lam[id - ID_FEAT0];
  };
  auto gsOf = [&](int id) { return id < ID_SB0 ? 7 : (id < ID_EX ? 9 : (id == ID_EX ? 7 : 1)); };

  if (win.has_prior && win.prior.n > 0) {
    bool use = true;
    std::vector<int> drop;
    for (size_t k = 0; k < win.prior.blk_kind.size(); k++) {
      int id = idOfPrior((int)k);
      if (flag == AVM_MARGIN_OLD && (id == 0 || id == ID_SB0)) drop.push_back((int)k);
      if (flag == AVM_MARGIN_SECOND_NEW && id == AVM_WINDOW_SIZE - 1) drop.push_back((int)k);
    }
    if (flag == AVM_MARGIN_SECOND_NEW && drop.empty()) use = false;  // estimator.cpp:926-927
    if (use) {
      MFactor f;
      f.nres = win.prior.n;
      std::vector<const double*> ps;
      for (size_t k = 0; k < win.prior.blk_kind.size(); k++) {
        int id = idOfPrior((int)k);
        f.ids.push_back(id);
        f.gs.push_back(gsOf(id));
        ps.push_back(paramOf(id));
      }
      std::vector<double> dx;
      prior_dx(win.prior, ps, dx);
      f.r.assign(f.nres, 0.0);
      prior_residual(win.prior, dx, f.r.data());
      for (size_t k = 0; k < f.ids.size(); k++) {
        int gs = f.gs[k], ls = gs == 7 ? 6 : gs, idx = win.prior.blk_idx[k];
        std::vector<double> J((size_t)f.nres * gs, 0.0);
        for (int i = 0; i < f.nres; i++)
          for (int c = 0; c < ls; c++) J[(size_t)i * gs + c] = win.prior.J(i, idx + c);
        f.J.push_back(J);
      }
      f.drop = drop;
      factors.push_back(f);
    } else {
      out = Prior();  // nothing to do: the caller keeps the old prior
      out.n = -1;
      return;
    }
  } else if (flag == AVM_MARGIN_SECOND_NEW) {
    out = Prior();
    out.n = -1;
    return;
  }
  if (flag == AVM_MARGIN_OLD) {
    if (win.pre[0].sum_dt < o.max_sum_dt) {  // estimator.cpp:841
      MFactor f;
      f.nres = 15;
      f.ids = {0, ID_SB0, 1, ID_SB0 + 1};
      f.gs = {7, 9, 7, 9};
      f.J = {std::vector<double>(15 * 7), std::vector<double>(15 * 9), std::vector<double>(15 * 7), std::vector<double>(15 * 9)};
      f.r.assign(15, 0.0);
      double* jac[4] = {f.J[0].data(), f.J[1].data(), f.J[2].data(), f.J[3].data()};
      imu_factor_evaluate(win.pre[0], win.sqrt_info[0], G, x.pose[0], x.sb[0], x.pose[1], x.sb[1], f.r.data(), jac);
      f.drop = {0, 1};
      factors.push_back(f);
    }
    for (int e = 0; e < win.nf; e++) {
      if (win.start[e] != 0) continue;  // estimator.cpp:861-863
      int s0 = win.obs_begin[e];
      V3 pts_i(win.obs_xy[2 * s0], win.obs_xy[2 * s0 + 1], 1.0);
      for (int t = 1; t < win.nobs[e]; t++) {
        int slot = s0 + t;
        V3 pts_j(win.obs_xy[2 * slot], win.obs_xy[2 * slot + 1], 1.0);
        MFactor f;
        f.nres = 2;
        f.ids = {0, t, ID_EX, ID_FEAT0 + e};
        f.gs = {7, 7, 7, 1};
        f.J = {std::vector<double>(14), std::vector<double>(14), std::vector<double>(14), std::vector<double>(2)};
        f.r.assign(2, 0.0);
        if (o.estimate_td) {  // estimator.cpp:874-885: ProjectionTdFactor, para_Td kept
          f.ids.push_back(ID_TD), f.gs.push_back(1), f.J.push_back(std::vector<double>(2));
          const double* ai = &win.obs_aux[4 * (size_t)s0];
          const double* aj = &win.obs_aux[4 * (size_t)slot];
          double j20[40];
          projection_td_factor_evaluate(&win.obs_xy[2 * s0], &win.obs_xy[2 * slot], ai, aj, ai[2], aj[2], ai[3], aj[3], o.tr, o.row, sq, x.pose[0],
                                        x.pose[t], x.ex, x.lam[e], x.td, f.r.data(), j20);
          for (int r = 0; r < 2; r++) {
            for (int c = 0; c < 6; c++) f.J[0][r * 7 + c] = j20[r * 20 + c], f.J[1][r * 7 + c] = j20[r * 20 + 6 + c], f.J[2][r * 7 + c] = j20[r * 20 + 12 + c];
            f.J[3][r] = j20[r * 20 + 18], f.J[4][r] = j20[r * 20 + 19];
          }
        } else {
        double* jac[4] = {f.J[0].data(), f.J[1].data(), f.J[2].data(), f.J[3].data()};
        projection_factor_evaluate(pts_i, pts_j, sq, x.pose[0], x.pose[t], x.ex, x.lam[e], f.r.data(), jac);
        }
        // ResidualBlockInfo::Evaluate loss correction (marginalization_factor.cpp:37-68)
        double sn = f.r[0] * f.r[0] + f.r[1] * f.r[1], rho[3];
        cauchy_loss(o.cauchy_a, sn, rho);
        Corrector corr(sn, rho);
        for (size_t k = 0; k < f.J.size(); k++) corr.correctJacobian(2, f.gs[k], f.r.data(), f.J[k].data());
        corr.correctResiduals(2, f.r.data());
        f.drop = {0, 3};
        factors.push_back(f);
      }
    }
  }
  // ordering
  std::vector<int> dropped, kept;
  auto contains = [](const std::vector<int>& v, int a) { return std::find(v.begin(), v.end(), a) != v.end(); };
  for (auto& f : factors)
    for (int d : f.drop)
      if (!contains(dropped, f.ids[d])) dropped.push_back(f.ids[d]);
  for (auto& f : factors)
    for (int id : f.ids)
      if (!contains(dropped, id) && !contains(kept, id)) kept.push_back(id);
  std::sort(dropped.begin(), dropped.end());
  std::sort(kept.begin(), kept.end());
  std::vector<int> idx(ID_FEAT0 + win.nf, -1);
  int pos = 0;
  auto ls = [&](int id) { int g = gsOf(id); return g == 7 ? 6 : g; };
  for (int id : dropped) idx[id] = pos, pos += ls(id);
  const int m = pos;
  for (int id : kept) idx[id] = pos, pos += ls(id);
  const int n = pos - m;
  Mat A(pos, pos);
  std::vector<double> b(pos, 0.0);
  for (auto& f : factors) {
    for (size_t i = 0; i < f.ids.size(); i++) {
      int oi = idx[f.ids[i]], si = ls(f.ids[i]), gi = f.gs[i];
      for (size_t j = i; j < f.ids.size(); j++) {
        int oj = idx[f.ids[j]], sj = ls(f.ids[j]), gj = f.gs[j];
        for (int a = 0; a < si; a++)
          for (int c = 0; c < sj; c++) {
            double s = 0;
            for (int r = 0; r < f.nres; r++) s += f.J[i][(size_t)r * gi + a] * f.J[j][(size_t)r * gj + c];
            A(oi + a, oj + c) += s;
            if (i != j) A(oj + c, oi + a) = A(oi + a, oj + c);
          }
      }
      for (int a = 0; a < si; a++) {
        double s = 0;
        for (int r = 0; r < f.nres; r++) s += f.J[i][(size_t)r * gi + a] * f.r[r];
        b[oi + a] += s;
      }
    }
  }
  // Amm pseudo-inverse through eigen-decomposition (marginalization_factor.cpp:267-272)
  Mat Amm(m, m);
  for (int i = 0; i < m; i++)
    for (int j = 0; j < m; j++) Amm(i, j) = 0.5 * (A(i, j) + A(j, i));
  std::vector<double> ev;
  Mat V;
  eig_sym(Amm, ev, V);
  Mat Amm_inv(m, m);
  for (int k = 0; k < m; k++) {
    double inv = ev[k] > o.marg_eps ? 1.0 / ev[k] : 0.0;
    if (inv == 0.0) continue;
    for (int i = 0; i < m; i++) {
      double vi = V(i, k) * inv;
      for (int j = 0; j < m; j++) Amm_inv(i, j) += vi * V(j, k);
    }
  }
  // Schur (marginalization_factor.cpp:275-281)
  Mat Arm(n, m), Amr(m, n), Arr(n, n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < m; j++) Arm(i, j) = A(m + i, j), Amr(j, i) = A(j, m + i);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) Arr(i, j) = A(m + i, m + j);
  Mat T = matmul(Arm, Amm_inv);
  Mat TA = matmul(T, Amr);
  Mat S(n, n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) S(i, j) = Arr(i, j) - TA(i, j);
  std::vector<double> bn(n);
  for (int i = 0; i < n; i++) {
    double s = 0;
    for (int j = 0; j < m; j++) s += T(i, j) * b[j];
    bn[i] = b[m + i] - s;
  }
  // eigen square root (marginalization_factor.cpp:283-291)
  std::vector<double> ev2;
  Mat V2;
  eig_sym(S, ev2, V2);
  if (diag) diag->m = m, diag->n = n, diag->ev_mm = ev, diag->ev_rr = ev2, diag->A_rr = S, diag->b_rr = bn;
  out = Prior();
  out.n = n;
  out.J = Mat(n, n);
  out.r.assign(n, 0.0);
  for (int k = 0; k < n; k++) {
    double Sk = ev2[k] > o.marg_eps ? ev2[k] : 0.0;
    double Sinv = ev2[k] > o.marg_eps ? 1.0 / ev2[k] : 0.0;
    double ssq = std::sqrt(Sk), sinvsq = std::sqrt(Sinv);
    double vb = 0;
    for (int j = 0; j < n; j++) {
      out.J(k, j) = ssq * V2(j, k);
      vb += V2(j, k) * bn[j];
    }
    out.r[k] = sinvsq * vb;
  }
  // getParameterBlocks + addr_shift (estimator.cpp:904-916 / :960-983)
  for (int id : kept) {
    int kind = id < ID_SB0 ? AVM_BLK_POSE : (id < ID_EX ? AVM_BLK_SPEEDBIAS : (id == ID_TD ? AVM_BLK_TD : AVM_BLK_EXPOSE));
    int fr = kind == AVM_BLK_POSE ? id : (kind == AVM_BLK_SPEEDBIAS ? id - ID_SB0 : 0);
    if (kind == AVM_BLK_POSE || kind == AVM_BLK_SPEEDBIAS) {
      if (flag == AVM_MARGIN_OLD)
        fr = fr - 1;
      else if (fr == AVM_WINDOW_SIZE)
        fr = fr - 1;
    }
    out.blk_kind.push_back(kind);
    out.blk_frame.push_back(fr);
    out.blk_idx.push_back(idx[id] - m);
    const double* p = paramOf(id);
    out.x0.push_back(std::vector<double>(p, p + gsOf(id)));
  }
}

}  // namespace avmo
