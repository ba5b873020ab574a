// oracle/window_io.hpp - TEST INFRASTRUCTURE ONLY (CPU oracle).  avm_window_batch (include/avm.h, FP64) -> the oracle's
// Window: the marshalling half of Estimator::optimization() (estimator.cpp:661-760: vector2double, the IntegrationBase of every
// interval from its raw samples, last_marginalization_info).  Shared by the FP64 oracle (avm_oracle.cpp) and by its
// extended-precision build (avm_truth.cpp, where the oracle's scalar type is __float128): every value crosses from the ABI's
// double to the oracle's scalar by assignment, never by memcpy.
#pragma once
#include "solver.hpp"

namespace avmo {

inline void load_window(const avm_options& o, const avm_window_batch& B, int w, Window& W) {
  for (int f = 0; f < AVM_NFRAMES; f++) {
    for (int k = 0; k < 7; k++) W.x.pose[f][k] = B.pose[((size_t)w * AVM_NFRAMES + f) * 7 + k];
    for (int k = 0; k < 9; k++) W.x.sb[f][k] = B.speedbias[((size_t)w * AVM_NFRAMES + f) * 9 + k];
  }
  for (int k = 0; k < 7; k++) W.x.ex[k] = B.ex_pose[(size_t)w * 7 + k];
  W.nf = B.n_feat[w];
  W.x.lam.assign(B.inv_depth + (size_t)w * B.max_feat, B.inv_depth + (size_t)w * B.max_feat + W.nf);
  W.start.assign(B.feat_start + (size_t)w * B.max_feat, B.feat_start + (size_t)w * B.max_feat + W.nf);
  W.nobs.assign(B.feat_nobs + (size_t)w * B.max_feat, B.feat_nobs + (size_t)w * B.max_feat + W.nf);
  W.obs_begin.assign(B.feat_obs_begin + (size_t)w * B.max_feat, B.feat_obs_begin + (size_t)w * B.max_feat + W.nf);
  W.obs_xy.assign(B.obs_xy + (size_t)w * B.max_obs * 2, B.obs_xy + ((size_t)w + 1) * B.max_obs * 2);
  ImuNoise nz{o.acc_n, o.gyr_n, o.acc_w, o.gyr_w};
  W.pre.clear();
  W.sqrt_info.clear();
  for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
    size_t iv = (size_t)w * AVM_WINDOW_SIZE + j;
    const auto* acc = B.imu_acc + iv * (B.max_samp + 1) * 3;
    const auto* gyr = B.imu_gyr + iv * (B.max_samp + 1) * 3;
    const auto* dt = B.imu_dt + iv * B.max_samp;
    const auto* lba = B.imu_lin_ba + iv * 3;
    const auto* lbg = B.imu_lin_bg + iv * 3;
    PreIntegration p(V3(acc[0], acc[1], acc[2]), V3(gyr[0], gyr[1], gyr[2]), V3(lba[0], lba[1], lba[2]), V3(lbg[0], lbg[1], lbg[2]), nz);
    int ns = B.imu_n[iv];
    for (int s = 0; s < ns; s++)
      p.push_back(dt[s], V3(acc[3 * (s + 1)], acc[3 * (s + 1) + 1], acc[3 * (s + 1) + 2]),
                  V3(gyr[3 * (s + 1)], gyr[3 * (s + 1) + 1], gyr[3 * (s + 1) + 2]));
    W.pre.push_back(p);
    W.sqrt_info.push_back(imu_sqrt_info(p));
  }
  // optional members
  W.obs_aux.clear();
  W.x.td = 0.0;
  if (o.estimate_td) {
    if (B.obs_vel_td) W.obs_aux.assign(B.obs_vel_td + (size_t)w * B.max_obs * 4, B.obs_vel_td + ((size_t)w + 1) * B.max_obs * 4);
    else W.obs_aux.assign((size_t)B.max_obs * 4, 0.0);
    if (B.td) W.x.td = B.td[w];
  }
  // relocalization_info (estimator.cpp:588-604): relo_Pose goes through double2vector's gauge fix whether or not a feature
  // matched; with no match (relo_n == 0) no factor references it and the solve leaves it where it was
  W.has_relo = B.relo_n && B.relo_feat && B.relo_xy && B.relo_pose;
  W.relo_n = W.has_relo ? std::max(B.relo_n[w], 0) : 0;
  W.relo_frame = B.relo_frame ? B.relo_frame[w] : 0;
  if (W.has_relo) {
    W.relo_feat.assign(B.relo_feat + (size_t)w * B.max_feat, B.relo_feat + (size_t)w * B.max_feat + W.relo_n);
    W.relo_xy.assign(B.relo_xy + (size_t)w * B.max_feat * 2, B.relo_xy + ((size_t)w * B.max_feat + W.relo_n) * 2);
    for (int k = 0; k < 7; k++) W.x.relo[k] = B.relo_pose[(size_t)w * 7 + k];
  }
  W.failure_occur = B.failure_occur && B.last_pose0 && B.failure_occur[w] != 0;
  if (W.failure_occur)
    for (int k = 0; k < 7; k++) W.last_pose0[k] = B.last_pose0[(size_t)w * 7 + k];
  W.has_prior = B.prior_n && B.prior_n[w] > 0;
  if (W.has_prior) {
    Prior& P = W.prior;
    P = Prior();
    P.n = B.prior_n[w];
    int nb = B.prior_nblk[w];
    int off = 0;
    for (int k = 0; k < nb; k++) {
      int kind = B.prior_blk_kind[(size_t)w * B.max_pblk + k];
      P.blk_kind.push_back(kind);
      P.blk_frame.push_back(B.prior_blk_frame[(size_t)w * B.max_pblk + k]);
      P.blk_idx.push_back(off);
      off += Prior::lsize(kind);
      const auto* x0 = B.prior_x0 + ((size_t)w * B.max_pblk + k) * 9;
      P.x0.push_back(std::vector<double>(x0, x0 + Prior::gsize(kind)));
    }
    P.J = Mat(P.n, P.n);
    for (int i = 0; i < P.n; i++)
      for (int j = 0; j < P.n; j++) P.J(i, j) = B.prior_J[((size_t)w * B.max_prior + i) * B.max_prior + j];
    P.r.assign(B.prior_r + (size_t)w * B.max_prior, B.prior_r + (size_t)w * B.max_prior + P.n);
  }
}

}  // namespace avmo
