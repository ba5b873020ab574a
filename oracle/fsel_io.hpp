// oracle/fsel_io.hpp - TEST INFRASTRUCTURE ONLY (CPU oracle).  avm_fsel_batch (include/avm.h, FP64) -> the oracle's FselProblem
// (the arguments of FeatureSelector::select, feature_selector.cpp:137-220).  Shared by the FP64 oracle (avm_oracle.cpp) and by its
// extended-precision build (avm_truth.cpp): every value crosses from the ABI's double to the oracle's scalar by assignment.
#pragma once
#include "fsel.hpp"

namespace avmo {

inline void load_fsel(const avm_fsel_batch& B, int p, FselProblem& P) {
  P.H = B.horizon;
  P.pos.resize(P.H + 1);
  P.quat.resize(P.H + 1);
  for (int h = 0; h <= P.H; h++) {
    const auto* a = B.hor_pos + ((size_t)p * (P.H + 1) + h) * 3;
    const auto* q = B.hor_quat + ((size_t)p * (P.H + 1) + h) * 4;
    P.pos[h] = V3(a[0], a[1], a[2]);
    P.quat[h] = Q(q[3], q[0], q[1], q[2]);
  }
  P.nrImu = B.nr_imu[p];
  P.deltaImu = B.delta_imu[p];
  P.accVar = B.acc_var;
  P.accBiasVar = B.acc_bias_var;
  P.q_IC = Q(B.q_ic[3], B.q_ic[0], B.q_ic[1], B.q_ic[2]);
  P.t_IC = V3(B.t_ic[0], B.t_ic[1], B.t_ic[2]);
  P.cam = FselCamera{B.fx, B.fy, B.cx, B.cy, B.k1, B.k2, B.p1, B.p2, B.image_width, B.image_height};
  int nc = B.n_cand[p], nu = B.n_used ? B.n_used[p] : 0, ncl = B.n_cloud ? B.n_cloud[p] : 0;
  P.cand_id.assign(B.cand_id + (size_t)p * B.max_cand, B.cand_id + (size_t)p * B.max_cand + nc);
  P.cand_x.resize(nc), P.cand_y.resize(nc), P.cand_p.resize(nc);
  for (int i = 0; i < nc; i++) {
    P.cand_x[i] = B.cand_xy[((size_t)p * B.max_cand + i) * 2];
    P.cand_y[i] = B.cand_xy[((size_t)p * B.max_cand + i) * 2 + 1];
    P.cand_p[i] = B.cand_prob[(size_t)p * B.max_cand + i];
  }
  P.used_id.clear(), P.used_x.clear(), P.used_y.clear();
  for (int i = 0; i < nu; i++) {
    P.used_id.push_back(B.used_id[(size_t)p * B.max_used + i]);
    P.used_x.push_back(B.used_xy[((size_t)p * B.max_used + i) * 2]);
    P.used_y.push_back(B.used_xy[((size_t)p * B.max_used + i) * 2 + 1]);
  }
  P.cloud_x.clear(), P.cloud_y.clear(), P.cloud_d.clear();
  for (int i = 0; i < ncl; i++) {
    P.cloud_x.push_back(B.cloud_xy[((size_t)p * B.max_cloud + i) * 2]);
    P.cloud_y.push_back(B.cloud_xy[((size_t)p * B.max_cloud + i) * 2 + 1]);
    P.cloud_d.push_back(B.cloud_depth[(size_t)p * B.max_cloud + i]);
  }
  P.maxFeatures = B.max_features;
}

}  // namespace avmo
