// oracle/avm_oracle.cpp — TEST INFRASTRUCTURE ONLY. The CPU restatement ("oracle") of the two
// hot paths, exported with the same POD structs as include/avm.h so tests/, smoke() and
// bench.py's cpu_baseline leg can hand identical buffers to the oracle and the HIP product.
// Nothing under anticipated-vins-mono_amd/ may include, link or call this file.
//
// PARITY UNPINNED: /root/reference has no tests, fixtures or golden vectors, and cannot be
// compiled in this image (needs ROS, Ceres, Eigen, OpenCV, Boost).  Ceres and Eigen are
// un-vendored third-party dependencies; their algorithms are restated from the published
// sources (see solver.hpp / linalg.hpp headers).  Anchors that do exist are exercised in
// tests/: the finite-difference convention of ProjectionFactor::check()
// (projection_factor.cpp:123-225) and the MATLAB transcript of createLinearImuMatrices
// (support_files/scripts/createMatricesLinearImuFactor.m:17-101, test_ccT.m:24-36).
#include <atomic>
#include <chrono>
#include <thread>

#include "fsel.hpp"
#include "fsel_io.hpp"
#include "adapters.hpp"
#include "slide.hpp"
#include "solver.hpp"
#include "triangulate.hpp"
#include "window_io.hpp"

using namespace avmo;

namespace {

void store_state(const avm_window_batch& B, int w, const State& x, bool td = false, bool relo = false) {
  if (td && B.td) B.td[w] = x.td;
  if (relo && B.relo_pose) std::memcpy(B.relo_pose + (size_t)w * 7, x.relo, 7 * sizeof(double));
  for (int f = 0; f < AVM_NFRAMES; f++) {
    std::memcpy(B.pose + ((size_t)w * AVM_NFRAMES + f) * 7, x.pose[f], 7 * sizeof(double));
    std::memcpy(B.speedbias + ((size_t)w * AVM_NFRAMES + f) * 9, x.sb[f], 9 * sizeof(double));
  }
  std::memcpy(B.ex_pose + (size_t)w * 7, x.ex, 7 * sizeof(double));
  for (size_t e = 0; e < x.lam.size(); e++) B.inv_depth[(size_t)w * B.max_feat + e] = x.lam[e];
}

void store_prior(avm_prior_out& O, int w, const Prior& P) {
  O.n[w] = P.n;
  if (P.n < 0) {
    O.nblk[w] = 0;
    return;
  }
  O.nblk[w] = (int)P.blk_kind.size();
  for (size_t k = 0; k < P.blk_kind.size(); k++) {
    O.blk_kind[(size_t)w * O.max_pblk + k] = P.blk_kind[k];
    O.blk_frame[(size_t)w * O.max_pblk + k] = P.blk_frame[k];
    double* x0 = O.x0 + ((size_t)w * O.max_pblk + k) * 9;
    for (int i = 0; i < 9; i++) x0[i] = i < (int)P.x0[k].size() ? P.x0[k][i] : 0.0;
  }
  for (int i = 0; i < P.n; i++) {
    for (int j = 0; j < P.n; j++) O.J[((size_t)w * O.max_prior + i) * O.max_prior + j] = P.J(i, j);
    O.r[(size_t)w * O.max_prior + i] = P.r[i];
  }
}

template <class F>
void parallel_for(int n, int n_threads, F f) {
  if (n_threads <= 1) {
    for (int i = 0; i < n; i++) f(i);
    return;
  }
  std::atomic<int> next(0);
  std::vector<std::thread> th;
  for (int t = 0; t < n_threads; t++)
    th.emplace_back([&]() {
      for (;;) {
        int i = next.fetch_add(1);
        if (i >= n) break;
        f(i);
      }
    });
  for (auto& t : th) t.join();
}


}  // namespace

extern "C" {

int avmo_default_options(avm_options* o) {
  std::memset(o, 0, sizeof *o);
  o->max_num_iterations = 8;
  o->estimate_extrinsic = 0;
  o->estimate_td = 0;
  o->marginalization_flag = AVM_MARGIN_OLD;
  o->focal_length = 460.0;
  o->g[0] = 0, o->g[1] = 0, o->g[2] = 9.81007;
  o->acc_n = 0.08, o->gyr_n = 0.004, o->acc_w = 0.00004, o->gyr_w = 2.0e-6;
  o->cauchy_a = 1.0;
  o->max_sum_dt = 10.0;
  o->initial_trust_region_radius = 1e4;
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->jacobi_scaling = 1;
  o->marg_eps = 1e-8;
  o->max_solver_time_s = 0.0;
  o->marg_noise_rel = 1e-18;  // (the oracle's own clamp is always the reference-literal one: this field only steers the product)
  o->tr = 0.0, o->row = 480.0;  // global shutter (config/euroc/euroc_config.yaml:66), image_height
  return 0;
}

// Estimator::optimization() on host buffers.  n_threads > 1 runs independent windows on
// std::threads (each solve stays single-threaded like Ceres num_threads=1).
int avmo_window_solve_batch(const avm_options* opt, const avm_window_batch* batch, avm_prior_out* prior_out,
                            avm_solve_summary* summary, int n_threads) {
  parallel_for(batch->n_windows, n_threads, [&](int w) {
    Window W;
    load_window(*opt, *batch, w, W);
    Problem P;
    P.build(W, *opt);
    SolveResult R = trust_region_solve(P, W.x);
    State out;
    out.lam = R.x.lam;
    gauge_fix_roundtrip(W.x, R.x, out, W.failure_occur ? W.last_pose0 : nullptr, W.has_relo);
    if (summary) summary[w] = R.sum;
    if (opt->marginalization_flag != AVM_MARGIN_NONE && prior_out) {
      Prior np;
      marginalize(W, out, *opt, np);
      store_prior(*prior_out, w, np);
    }
    store_state(*batch, w, out, opt->estimate_td != 0, W.has_relo);
  });
  return 0;
}

int avmo_imu_preintegrate_batch(const avm_options* opt, const avm_window_batch* batch, double* out_delta, double* out_jacobian,
                                double* out_covariance, double* out_sum_dt, double* out_sqrt_info) {
  for (int w = 0; w < batch->n_windows; w++) {
    Window W;
    avm_window_batch B = *batch;
    int32_t zero = 0;
    (void)zero;
    load_window(*opt, B, w, W);
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
      size_t iv = (size_t)w * AVM_WINDOW_SIZE + j;
      const PreIntegration& p = W.pre[j];
      double* d = out_delta + iv * 10;
      d[0] = p.delta_p.x, d[1] = p.delta_p.y, d[2] = p.delta_p.z;
      d[3] = p.delta_q.x, d[4] = p.delta_q.y, d[5] = p.delta_q.z, d[6] = p.delta_q.w;
      d[7] = p.delta_v.x, d[8] = p.delta_v.y, d[9] = p.delta_v.z;
      std::memcpy(out_jacobian + iv * 225, p.jacobian.a.data(), 225 * sizeof(double));
      std::memcpy(out_covariance + iv * 225, p.covariance.a.data(), 225 * sizeof(double));
      out_sum_dt[iv] = p.sum_dt;
      if (out_sqrt_info) std::memcpy(out_sqrt_info + iv * 225, W.sqrt_info[j].a.data(), 225 * sizeof(double));
    }
  }
  return 0;
}

int avmo_window_eval_factors(const avm_options* opt, const avm_window_batch* batch, int apply_loss, double* proj_r, double* proj_J,
                             double* imu_r, double* imu_J, double* prior_res, double* cost) {
  const double sq = opt->focal_length / 1.5;
  V3 G(opt->g[0], opt->g[1], opt->g[2]);
  for (int w = 0; w < batch->n_windows; w++) {
    Window W;
    load_window(*opt, *batch, w, W);
    double c = 0;
    if (W.has_prior) {
      std::vector<const double*> ps;
      for (size_t k = 0; k < W.prior.blk_kind.size(); k++) {
        int kind = W.prior.blk_kind[k], fr = W.prior.blk_frame[k];
        ps.push_back(kind == AVM_BLK_POSE ? W.x.pose[fr] : (kind == AVM_BLK_SPEEDBIAS ? W.x.sb[fr] : W.x.ex));
      }
      std::vector<double> dx, res(W.prior.n);
      prior_dx(W.prior, ps, dx);
      prior_residual(W.prior, dx, res.data());
      for (int i = 0; i < W.prior.n; i++) {
        if (prior_res) prior_res[(size_t)w * batch->max_prior + i] = res[i];
        c += 0.5 * res[i] * res[i];
      }
    }
    for (int i = 0; i < AVM_WINDOW_SIZE; i++) {
      double r[15], j0[105], j1[135], j2[105], j3[135];
      double* jac[4] = {j0, j1, j2, j3};
      imu_factor_evaluate(W.pre[i], W.sqrt_info[i], G, W.x.pose[i], W.x.sb[i], W.x.pose[i + 1], W.x.sb[i + 1], r, jac);
      size_t iv = (size_t)w * AVM_WINDOW_SIZE + i;
      for (int k = 0; k < 15; k++) {
        if (imu_r) imu_r[iv * 15 + k] = r[k];
        if (W.pre[i].sum_dt <= opt->max_sum_dt) c += 0.5 * r[k] * r[k];
        if (imu_J) {
          double* row = imu_J + (iv * 15 + k) * 30;
          for (int cc = 0; cc < 6; cc++) row[cc] = j0[k * 7 + cc], row[15 + cc] = j2[k * 7 + cc];
          for (int cc = 0; cc < 9; cc++) row[6 + cc] = j1[k * 9 + cc], row[21 + cc] = j3[k * 9 + cc];
        }
      }
    }
    for (int e = 0; e < W.nf; e++) {
      int s0 = W.obs_begin[e];
      V3 pts_i(W.obs_xy[2 * s0], W.obs_xy[2 * s0 + 1], 1.0);
      for (int t = 1; t < W.nobs[e]; t++) {
        int slot = s0 + t;
        V3 pts_j(W.obs_xy[2 * slot], W.obs_xy[2 * slot + 1], 1.0);
        double r[2], j0[14], j1[14], j2[14], j3[2];
        double* jac[4] = {j0, j1, j2, j3};
        projection_factor_evaluate(pts_i, pts_j, sq, W.x.pose[W.start[e]], W.x.pose[W.start[e] + t], W.x.ex, W.x.lam[e], r, jac);
        double J[26];
        for (int k = 0; k < 2; k++) {
          for (int cc = 0; cc < 6; cc++) J[k * 13 + cc] = j0[k * 7 + cc], J[k * 13 + 6 + cc] = j1[k * 7 + cc];
          J[k * 13 + 12] = j3[k];
        }
        double sn = r[0] * r[0] + r[1] * r[1], rho[3];
        cauchy_loss(opt->cauchy_a, sn, rho);
        c += 0.5 * rho[0];
        if (apply_loss) {
          Corrector corr(sn, rho);
          corr.correctJacobian(2, 13, r, J);
          corr.correctResiduals(2, r);
        }
        size_t o = (size_t)w * batch->max_obs + slot;
        if (proj_r) proj_r[o * 2] = r[0], proj_r[o * 2 + 1] = r[1];
        if (proj_J) std::memcpy(proj_J + o * 26, J, sizeof J);
      }
    }
    if (cost) cost[w] = c;
  }
  return 0;
}

int avmo_fsel_select_batch(const avm_fsel_batch* batch, avm_fsel_out* out, int n_threads, int64_t* n_logdet) {
  std::atomic<long> total(0);
  parallel_for(batch->n_problems, n_threads, [&](int p) {
    FselProblem P;
    load_fsel(*batch, p, P);
    FselResult R = fsel_select(P, out->min_gap != nullptr);
    out->n_selected[p] = (int)R.selected.size();
    for (size_t i = 0; i < R.selected.size(); i++) {
      out->selected_ids[(size_t)p * batch->max_features + i] = R.selected[i];
      if (out->fvalues) out->fvalues[(size_t)p * batch->max_features + i] = R.fvalues[i];
      if (out->min_gap) out->min_gap[(size_t)p * batch->max_features + i] = R.min_gap[i];
    }
    total += R.n_logdet;
  });
  if (n_logdet) *n_logdet = total;
  return 0;
}

int avmo_fsel_information(const avm_fsel_batch* batch, double* omega, double* delta_cand, int32_t* cand_valid) {
  const int H = batch->horizon, N = 9 * (H + 1), H3 = 3 * H;
  for (int p = 0; p < batch->n_problems; p++) {
    FselProblem P;
    load_fsel(*batch, p, P);
    Mat Om = calcInfoFromRobotMotion(P);
    if (omega) std::memcpy(omega + (size_t)p * N * N, Om.a.data(), sizeof(double) * N * N);
    std::map<int, Mat> D = calcInfoFromFeatures(P, P.cand_id, P.cand_x, P.cand_y);
    for (size_t c = 0; c < P.cand_id.size(); c++) {
      auto it = D.find(P.cand_id[c]);
      if (cand_valid) cand_valid[(size_t)p * batch->max_cand + c] = it != D.end();
      if (!delta_cand) continue;
      double* dst = delta_cand + ((size_t)p * batch->max_cand + c) * H3 * H3;
      for (int i = 0; i < H3; i++)
        for (int j = 0; j < H3; j++)
          dst[i * H3 + j] = it == D.end() ? 0.0 : it->second(9 * (1 + i / 3) + i % 3, 9 * (1 + j / 3) + j % 3);
    }
  }
  return 0;
}

// bench leg only: the vectorisable forms of the Schur update, the Cholesky factorization and the forward substitution (linalg.hpp);
// bit-identical results, off by default
int avmo_set_fast_linalg(int on) {
  fast_linalg() = on != 0;
  return 0;
}

// findNNDepth (feature_selector.cpp:437-459) of every candidate: depth_out[P][max_cand]; pinned against the reference's own nanoflann
// (tests/golden/nanoflann_nn.npz)
int avmo_fsel_nn_depth(const avm_fsel_batch* batch, double* depth_out) {
  for (int p = 0; p < batch->n_problems; p++) {
    FselProblem P;
    load_fsel(*batch, p, P);
    const KdIndex kd(P.cloud_x, P.cloud_y);  // initKDTree, feature_selector.cpp:424-429
    for (size_t c = 0; c < P.cand_id.size(); c++) depth_out[(size_t)p * batch->max_cand + c] = findNNDepth(P, kd, P.cand_x[c], P.cand_y[c]);
  }
  return 0;
}

// createLinearImuMatrices exported for the MATLAB known-answer test; quaternions x,y,z,w
int avmo_linear_imu_matrices(const double* qi, const double* qj, double nr, double delta, double accVar, double biasVar, double* Omega,
                             double* Ablk) {
  Mat W, A;
  createLinearImuMatrices(Q(qi[3], qi[0], qi[1], qi[2]), Q(qj[3], qj[0], qj[1], qj[2]), nr, delta, accVar, biasVar, W, A);
  std::memcpy(Omega, W.a.data(), 81 * sizeof(double));
  std::memcpy(Ablk, A.a.data(), 81 * sizeof(double));
  return 0;
}

// symmetric eigen-solver exported for unit tests against numpy
int avmo_eig_sym(int n, const double* A, double* w, double* V) {
  Mat M(n, n);
  std::memcpy(M.a.data(), A, sizeof(double) * n * n);
  std::vector<double> d;
  Mat VV;
  eig_sym(M, d, VV);
  std::memcpy(w, d.data(), sizeof(double) * n);
  std::memcpy(V, VV.a.data(), sizeof(double) * n * n);
  return 0;
}

// FeatureManager::triangulate for every feature of every window whose inverse depth is <= 0 ("no depth yet":
// estimated_depth starts at -1, feature_manager.h:63); writes 1 / depth back.  Also exposes the (2 nobs) x 4 matrix
// of one feature for the numpy SVD known-answer test (A_out may be null).
int avmo_triangulate_batch(avm_window_batch* B, double init_depth, int n_threads) {
  const int nw = B->n_windows;
  std::atomic<int> next{0};
  auto work = [&]() {
    for (int w = next++; w < nw; w = next++) {
      const double(*pose)[7] = reinterpret_cast<const double(*)[7]>(B->pose + (size_t)w * AVM_NFRAMES * 7);
      for (int e = 0; e < B->n_feat[w]; e++) {
        double& lam = B->inv_depth[(size_t)w * B->max_feat + e];
        if (lam > 0) continue;
        const int start = B->feat_start[(size_t)w * B->max_feat + e], nobs = B->feat_nobs[(size_t)w * B->max_feat + e];
        const double* obs = B->obs_xy + ((size_t)w * B->max_obs + B->feat_obs_begin[(size_t)w * B->max_feat + e]) * 2;
        lam = 1.0 / triangulate_feature(pose, B->ex_pose + (size_t)w * 7, start, nobs, obs, init_depth);
      }
    }
  };
  std::vector<std::thread> th;
  for (int i = 1; i < std::max(1, n_threads); i++) th.emplace_back(work);
  work();
  for (auto& t : th) t.join();
  return 0;
}

int avmo_smallest_right_singular_vector(int n, const double* A, double* v) {
  smallest_right_singular_vector(std::vector<double>(A, A + (size_t)n * 4), n, v);
  return 0;
}

// Estimator::processIMU dead-reckoning of frame WINDOW_SIZE from the samples of the last interval
int avmo_imu_propagate_batch(avm_window_batch* B, const double* g) {
  for (int w = 0; w < B->n_windows; w++) {
    const size_t iv = (size_t)w * AVM_WINDOW_SIZE + (AVM_WINDOW_SIZE - 1);
    propagate_newest_frame(B->pose + ((size_t)w * AVM_NFRAMES + AVM_WINDOW_SIZE) * 7, B->speedbias + ((size_t)w * AVM_NFRAMES + AVM_WINDOW_SIZE) * 9,
                           B->imu_n[iv], B->imu_dt + iv * B->max_samp, B->imu_acc + iv * (B->max_samp + 1) * 3,
                           B->imu_gyr + iv * (B->max_samp + 1) * 3, V3(g[0], g[1], g[2]));
  }
  return 0;
}

int avmo_fsel_horizon_imu(const avm_fsel_horizon_in* in, double* hor_pos, double* hor_quat) {
  const int H = in->horizon;
  for (int p = 0; p < in->n_problems; p++)
    horizon_imu(H, in->k_pos + 3 * p, in->k_quat + 4 * p, in->k_ba + 3 * p, in->k1_pos + 3 * p, in->k1_vel + 3 * p, in->k1_quat + 4 * p,
                in->acc + 3 * p, in->gyr + 3 * p, in->nr_imu[p], in->delta_imu[p], hor_pos + (size_t)p * (H + 1) * 3, hor_quat + (size_t)p * (H + 1) * 4);
  return 0;
}

int avmo_projection_td_eval(const avm_td_factor_batch* f, double* residual, double* jac) {
  const double s = f->focal_length / 1.5;
  for (int i = 0; i < f->n; i++)
    projection_td_factor_evaluate(f->pts_i + 2 * i, f->pts_j + 2 * i, f->vel_i + 2 * i, f->vel_j + 2 * i, f->td_i[i], f->td_j[i], f->row_i[i], f->row_j[i],
                                  f->tr, f->row, s, f->pose_i + 7 * i, f->pose_j + 7 * i, f->ex_pose + 7 * i, f->inv_depth[i], f->td[i],
                                  residual + 2 * i, jac ? jac + 40 * (size_t)i : nullptr);
  return 0;
}

int avmo_fsel_build_cloud(const avm_window_batch* B, const double* k1_pos, const double* k1_quat, int max_cloud, int32_t* n_cloud,
                          double* cloud_xy, double* cloud_depth) {
  for (int w = 0; w < B->n_windows; w++) {
    const double(*pose)[7] = reinterpret_cast<const double(*)[7]>(B->pose + (size_t)w * AVM_NFRAMES * 7);
    n_cloud[w] = build_cloud(pose, B->ex_pose + (size_t)w * 7, B->n_feat[w], B->feat_start + (size_t)w * B->max_feat,
                             B->feat_obs_begin + (size_t)w * B->max_feat, B->obs_xy + (size_t)w * B->max_obs * 2,
                             B->inv_depth + (size_t)w * B->max_feat, k1_pos + 3 * (size_t)w, k1_quat + 4 * (size_t)w, max_cloud,
                             cloud_xy + (size_t)w * max_cloud * 2, cloud_depth + (size_t)w * max_cloud);
  }
  return 0;
}

void* avmo_gt_from_rows(const double* rows, int n) {
  GroundTruth* g = new GroundTruth;
  g->load(rows, n);
  return g;
}
void avmo_gt_free(void* g) { delete static_cast<GroundTruth*>(g); }
int avmo_gt_seek(void* g) { return static_cast<GroundTruth*>(g)->seek_idx; }
int avmo_fsel_horizon_ground_truth(void* gp, int H, double t0, const double* k_pos, const double* k_quat, double deltaFrame, double* hor_pos,
                                   double* hor_quat) {
  GroundTruth* g = static_cast<GroundTruth*>(gp);
  std::vector<V3> pos;
  std::vector<Q> quat;
  if (!g->horizon(H, t0, V3(k_pos[0], k_pos[1], k_pos[2]), Q(k_quat[3], k_quat[0], k_quat[1], k_quat[2]), deltaFrame, pos, quat)) return -1;
  for (int h = 0; h <= H; h++) {
    hor_pos[3 * h] = pos[h].x, hor_pos[3 * h + 1] = pos[h].y, hor_pos[3 * h + 2] = pos[h].z;
    hor_quat[4 * h] = quat[h].x, hor_quat[4 * h + 1] = quat[h].y, hor_quat[4 * h + 2] = quat[h].z, hor_quat[4 * h + 3] = quat[h].w;
  }
  return 0;
}
int avmo_image_from_pointcloud(int n, const float* pts, const float* const* ch, int num_cam, int32_t* fid, int32_t* cam, double* out) {
  return image_from_pointcloud(n, pts, ch, num_cam, fid, cam, out) ? 0 : -1;
}

// tables rewritten in place; the oracle repacks the observations (feat_obs_begin differs from the device's in-place roll:
// compare observations through feat_obs_begin, not raw obs_xy)
int avmo_slide_window(avm_window_batch* B, int flag, int shift_depth, double init_depth) {
  for (int w = 0; w < B->n_windows; w++) {
    int nf = B->n_feat[w];
    const bool ok = slide_window_one(
        flag, shift_depth != 0, init_depth, reinterpret_cast<double(*)[7]>(B->pose + (size_t)w * 77), reinterpret_cast<double(*)[9]>(B->speedbias + (size_t)w * 99),
        B->ex_pose + (size_t)w * 7, B->max_samp, const_cast<int32_t*>(B->imu_n) + (size_t)w * 10, const_cast<double*>(B->imu_dt) + (size_t)w * 10 * B->max_samp,
        const_cast<double*>(B->imu_acc) + (size_t)w * 10 * (B->max_samp + 1) * 3, const_cast<double*>(B->imu_gyr) + (size_t)w * 10 * (B->max_samp + 1) * 3,
        const_cast<double*>(B->imu_lin_ba) + (size_t)w * 30, const_cast<double*>(B->imu_lin_bg) + (size_t)w * 30, nf,
        const_cast<int32_t*>(B->feat_start) + (size_t)w * B->max_feat, const_cast<int32_t*>(B->feat_nobs) + (size_t)w * B->max_feat,
        const_cast<int32_t*>(B->feat_obs_begin) + (size_t)w * B->max_feat, const_cast<double*>(B->obs_xy) + (size_t)w * B->max_obs * 2,
        B->inv_depth + (size_t)w * B->max_feat);
    if (!ok) return -3;
    const_cast<int32_t*>(B->n_feat)[w] = nf;
  }
  return 0;
}

}  // extern "C"
