// oracle/slide.hpp - TEST INFRASTRUCTURE ONLY: CPU restatement of the window roll on the batch tables.
//   Estimator::slideWindow / slideWindowNew / slideWindowOld    vins_estimator/src/estimator.cpp:996-1107
//   FeatureManager::removeBackShiftDepth / removeBack / removeFront   vins_estimator/src/feature_manager.cpp:275-352
// "parity unpinned" (the reference cannot be built here); pinned by a list-based Python statement in tests/test_oracle.py.
#pragma once
#include <list>
#include <vector>
#include "linalg.hpp"

namespace avmo {

struct SlideFeature {  // FeaturePerId as far as the roll touches it
  int start_frame;
  std::vector<V3> feature_per_frame;  // .point
  double estimated_depth;
};

// returns false on capacity overflow (MARGIN_SECOND_NEW: interval 8 + interval 9 > max_samp)
inline bool slide_window_one(int flag, bool shift_depth, double INIT_DEPTH, double (*pose)[7], double (*sb)[9], const double* ex, int max_samp,
                             int* imu_n, double* imu_dt, double* imu_acc, double* imu_gyr, double* lin_ba, double* lin_bg, int& n_feat, int* fstart,
                             int* fnobs, int* fobs, double* obs_xy, double* inv_depth) {
  const int WINDOW_SIZE = 10, frame_count = 10;
  const int SD = max_samp, SA = (max_samp + 1) * 3;
  // ---- f_manager view of the tables
  std::list<SlideFeature> feature;
  for (int e = 0; e < n_feat; e++) {
    SlideFeature f{fstart[e], {}, 1.0 / inv_depth[e]};
    for (int k = 0; k < fnobs[e]; k++) f.feature_per_frame.push_back(V3(obs_xy[2 * (fobs[e] + k)], obs_xy[2 * (fobs[e] + k) + 1], 1.0));
    feature.push_back(f);
  }
  auto Rs = [&](int i) { return toR(Q(pose[i][6], pose[i][3], pose[i][4], pose[i][5])); };
  auto Ps = [&](int i) { return V3(pose[i][0], pose[i][1], pose[i][2]); };
  const M3 ric = toR(Q(ex[6], ex[3], ex[4], ex[5]));
  const V3 tic(ex[0], ex[1], ex[2]);
  // acc_0 / gyr_0 of the estimator: the last sample it pushed
  double acc_0[3], gyr_0[3];
  for (int k = 0; k < 3; k++) acc_0[k] = imu_acc[9 * SA + imu_n[9] * 3 + k], gyr_0[k] = imu_gyr[9 * SA + imu_n[9] * 3 + k];
  if (flag == 0 /* MARGIN_OLD */) {
    const M3 back_R0 = Rs(0);
    const V3 back_P0 = Ps(0);
    for (int i = 0; i < WINDOW_SIZE; i++) {  // the swaps of estimator.cpp:1005-1019 (what ends up in slot 10 is overwritten below)
      for (int k = 0; k < 7; k++) std::swap(pose[i][k], pose[i + 1][k]);
      for (int k = 0; k < 9; k++) std::swap(sb[i][k], sb[i + 1][k]);
    }
    for (int k = 0; k < 7; k++) pose[WINDOW_SIZE][k] = pose[WINDOW_SIZE - 1][k];
    for (int k = 0; k < 9; k++) sb[WINDOW_SIZE][k] = sb[WINDOW_SIZE - 1][k];
    for (int j = 0; j + 1 < 10; j++) {  // pre_integrations[i] <-> [i + 1] for the intervals that survive
      imu_n[j] = imu_n[j + 1];
      for (int k = 0; k < SD; k++) imu_dt[j * SD + k] = imu_dt[(j + 1) * SD + k];
      for (int k = 0; k < SA; k++) imu_acc[j * SA + k] = imu_acc[(j + 1) * SA + k], imu_gyr[j * SA + k] = imu_gyr[(j + 1) * SA + k];
      for (int k = 0; k < 3; k++) lin_ba[j * 3 + k] = lin_ba[(j + 1) * 3 + k], lin_bg[j * 3 + k] = lin_bg[(j + 1) * 3 + k];
    }
    // slideWindowOld
    if (shift_depth) {
      const M3 R0 = back_R0 * ric, R1 = Rs(0) * ric;
      const V3 P0 = back_P0 + back_R0 * tic, P1 = Ps(0) + Rs(0) * tic;
      for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
        it_next++;
        if (it->start_frame != 0) {
          it->start_frame--;
        } else {
          const V3 uv_i = it->feature_per_frame[0];
          it->feature_per_frame.erase(it->feature_per_frame.begin());
          if (it->feature_per_frame.size() < 2) {
            feature.erase(it);
            continue;
          }
          const V3 pts_i = uv_i * it->estimated_depth;
          const V3 w_pts_i = R0 * pts_i + P0;
          const V3 pts_j = transpose(R1) * (w_pts_i - P1);
          const double dep_j = pts_j.z;
          it->estimated_depth = dep_j > 0 ? dep_j : INIT_DEPTH;
        }
      }
    } else {
      for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
        it_next++;
        if (it->start_frame != 0) {
          it->start_frame--;
        } else {
          it->feature_per_frame.erase(it->feature_per_frame.begin());
          if (it->feature_per_frame.size() == 0) feature.erase(it);
        }
      }
    }
  } else {
    if (imu_n[8] + imu_n[9] > max_samp) return false;
    for (int i = 0; i < imu_n[9]; i++) {  // pre_integrations[frame_count - 1]->push_back(...)
      const int o = imu_n[8];
      imu_dt[8 * SD + o] = imu_dt[9 * SD + i];
      for (int k = 0; k < 3; k++) imu_acc[8 * SA + (o + 1) * 3 + k] = imu_acc[9 * SA + (i + 1) * 3 + k], imu_gyr[8 * SA + (o + 1) * 3 + k] = imu_gyr[9 * SA + (i + 1) * 3 + k];
      imu_n[8]++;
    }
    for (int k = 0; k < 7; k++) pose[frame_count - 1][k] = pose[frame_count][k];
    for (int k = 0; k < 9; k++) sb[frame_count - 1][k] = sb[frame_count][k];
    // slideWindowNew -> removeFront(frame_count)
    for (auto it = feature.begin(), it_next = feature.begin(); it != feature.end(); it = it_next) {
      it_next++;
      if (it->start_frame == frame_count) {
        it->start_frame--;
      } else {
        const int j = WINDOW_SIZE - 1 - it->start_frame;
        if (it->start_frame + (int)it->feature_per_frame.size() - 1 < frame_count - 1) continue;
        it->feature_per_frame.erase(it->feature_per_frame.begin() + j);
        if (it->feature_per_frame.size() == 0) feature.erase(it);
      }
    }
  }
  // new IntegrationBase{acc_0, gyr_0, Bas[WINDOW_SIZE], Bgs[WINDOW_SIZE]}
  imu_n[9] = 0;
  for (int k = 0; k < 3; k++) {
    imu_acc[9 * SA + k] = acc_0[k], imu_gyr[9 * SA + k] = gyr_0[k];
    lin_ba[27 + k] = sb[WINDOW_SIZE][3 + k], lin_bg[27 + k] = sb[WINDOW_SIZE][6 + k];
  }
  // ---- back to the tables: observations are rewritten from the start of the window's region in list order
  int e = 0, o = 0;
  for (const auto& f : feature) {
    fstart[e] = f.start_frame, fnobs[e] = (int)f.feature_per_frame.size(), fobs[e] = o, inv_depth[e] = 1.0 / f.estimated_depth;
    for (const auto& p : f.feature_per_frame) obs_xy[2 * o] = p.x, obs_xy[2 * o + 1] = p.y, o++;
    e++;
  }
  n_feat = e;
  return true;
}

}  // namespace avmo
