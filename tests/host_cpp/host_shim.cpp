// Test hooks around include/avm_host.hpp: lets pytest drive the C++ host objects (avm_host::Estimator,
// avm_host::FeatureSelector) through ctypes.  Test infrastructure only - the product is the header.
// Built by __graft_entry__.build() into tests/host_cpp/libavm_host_shim.so (links libavm_hip.so).
#include <chrono>
#include <cstring>
#include <memory>

#include "avm_host.hpp"

using namespace avm_host;

namespace {
thread_local std::string g_err;
thread_local int g_status = 0;
thread_local double g_call_ms = 0.0;  // wall time of the last optimization() / select() call itself (bench.py: latency_host_call_ms)
struct Stopwatch {
  std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~Stopwatch() { g_call_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(); }
};

struct Host {
  Context ctx;
  Estimator est;
  std::unique_ptr<FeatureSelector> sel;
  explicit Host(int device) : ctx(device, 1, 1), est(ctx) {}
};

template <class F>
int guarded(F f) {
  try {
    f();
    g_status = 0;
    return 0;
  } catch (const Error& e) {
    g_err = e.what(), g_status = e.status;
    return e.status;
  } catch (const std::exception& e) {
    g_err = e.what(), g_status = -100;
    return -100;
  }
}
}  // namespace

extern "C" {

const char* hs_last_error() { return g_err.c_str(); }
double hs_last_call_ms() { return g_call_ms; }

void* hs_create(int device) { return new Host(device); }
void hs_destroy(void* h) { delete static_cast<Host*>(h); }

// fill the Estimator's members from window w of a host batch (the inverse of Estimator::marshal); solve_flag[e] and
// feat_id[e] per table row; extra_* describe features that fail the filter of estimator.cpp:715 and must be skipped
// by the marshal (start frame, number of observations), interleaved BEFORE table row extra_before[k].
int hs_load_window(void* hp, const avm_window_batch* b, int w, const int32_t* feat_id, const int32_t* solve_flag, int n_extra,
                   const int32_t* extra_before, const int32_t* extra_start, const int32_t* extra_nobs) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] {
    Estimator& E = H.est;
    E.clearState();
    const double* pose = b->pose + (size_t)w * 77;
    const double* sb = b->speedbias + (size_t)w * 99;
    for (int i = 0; i < AVM_NFRAMES; i++) {
      E.Ps[i] = {pose[7 * i], pose[7 * i + 1], pose[7 * i + 2]};
      E.Rs[i] = Quaterniond{pose[7 * i + 3], pose[7 * i + 4], pose[7 * i + 5], pose[7 * i + 6]};
      for (int k = 0; k < 3; k++) E.Vs[i][k] = sb[9 * i + k], E.Bas[i][k] = sb[9 * i + 3 + k], E.Bgs[i][k] = sb[9 * i + 6 + k];
    }
    const double* ex = b->ex_pose + (size_t)w * 7;
    E.tic[0] = {ex[0], ex[1], ex[2]};
    E.ric[0] = Quaterniond{ex[3], ex[4], ex[5], ex[6]};
    int x = 0;
    const int nf = b->n_feat[w];
    for (int e = 0; e <= nf; e++) {
      while (x < n_extra && extra_before[x] == e) {
        FeaturePerId f(1000000 + x, extra_start[x]);
        f.feature_per_frame.resize(extra_nobs[x]);
        f.estimated_depth = 3.0, f.solve_flag = 1;
        E.f_manager.feature.push_back(f);
        x++;
      }
      if (e == nf) break;
      const size_t fe = (size_t)w * b->max_feat + e;
      FeaturePerId f(feat_id ? feat_id[e] : e, b->feat_start[fe]);
      f.estimated_depth = 1.0 / b->inv_depth[fe];
      f.solve_flag = solve_flag ? solve_flag[e] : 0;
      for (int t = 0; t < b->feat_nobs[fe]; t++) {
        const double* o = b->obs_xy + ((size_t)w * b->max_obs + b->feat_obs_begin[fe] + t) * 2;
        FeaturePerFrame pf;
        pf.point = {o[0], o[1], 1.0};
        if (b->obs_vel_td) {  // FeaturePerFrame::velocity / cur_td / uv.y of the td factor
          const double* a = b->obs_vel_td + ((size_t)w * b->max_obs + b->feat_obs_begin[fe] + t) * 4;
          pf.velocity = {a[0], a[1]}, pf.cur_td = a[2], pf.uv = {0.0, a[3]};
        }
        f.feature_per_frame.push_back(pf);
      }
      E.f_manager.feature.push_back(f);
    }
    E.td = b->td ? b->td[w] : 0.0;
    E.relocalization_info = b->relo_n && b->relo_n[w] > 0;
    E.match_points.clear();
    if (E.relocalization_info) {  // setReloFrame (estimator.cpp:1120-1141): match_points = (x, y, feature id), ascending ids
      E.relo_frame_local_index = b->relo_frame[w];
      for (int k = 0; k < b->relo_n[w]; k++) {
        const int e = b->relo_feat[(size_t)w * b->max_feat + k];
        const double* xy = b->relo_xy + ((size_t)w * b->max_feat + k) * 2;
        E.match_points.push_back(Vector3d{xy[0], xy[1], (double)(feat_id ? feat_id[e] : e)});
      }
      std::copy(b->relo_pose + (size_t)w * 7, b->relo_pose + (size_t)w * 7 + 7, E.relo_Pose);
      E.prev_relo_t = {E.relo_Pose[0] + 0.3, E.relo_Pose[1] - 0.2, E.relo_Pose[2] + 0.1};  // (any loop-frame pose: the outputs are checked against it)
      E.prev_relo_r = Quaterniond{0.0, 0.0, std::sin(0.2), std::cos(0.2)};
    }
    E.failure_occur = b->failure_occur && b->failure_occur[w] != 0;
    if (E.failure_occur) {
      const double* lp = b->last_pose0 + (size_t)w * 7;
      E.last_P0 = {lp[0], lp[1], lp[2]}, E.last_R0 = Quaterniond{lp[3], lp[4], lp[5], lp[6]};
    }
    const size_t S = b->max_samp;
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
      const size_t row0 = ((size_t)w * AVM_WINDOW_SIZE + j) * (S + 1);
      auto v3 = [](const double* p) { return Vector3d{p[0], p[1], p[2]}; };
      IntegrationBase p(v3(b->imu_acc + row0 * 3), v3(b->imu_gyr + row0 * 3), v3(b->imu_lin_ba + ((size_t)w * 10 + j) * 3),
                        v3(b->imu_lin_bg + ((size_t)w * 10 + j) * 3));
      for (int s = 0; s < b->imu_n[(size_t)w * 10 + j]; s++)
        p.push_back(b->imu_dt[((size_t)w * 10 + j) * S + s], v3(b->imu_acc + (row0 + s + 1) * 3), v3(b->imu_gyr + (row0 + s + 1) * 3));
      E.pre_integrations[j + 1] = p;
    }
    MarginalizationInfo& M = E.last_marginalization_info;
    M = MarginalizationInfo{};
    if (b->prior_n && b->prior_n[w] > 0) {
      const int mp = b->max_prior, mb = b->max_pblk;
      if (mp != WindowTables::MAX_PRIOR || mb != WindowTables::MAX_PBLK) throw Error(AVM_ERR_INVALID, "shim expects max_prior 96 / max_pblk 16");
      M.n = b->prior_n[w], M.nblk = b->prior_nblk[w];
      M.blk_kind.assign(b->prior_blk_kind + (size_t)w * mb, b->prior_blk_kind + (size_t)(w + 1) * mb);
      M.blk_frame.assign(b->prior_blk_frame + (size_t)w * mb, b->prior_blk_frame + (size_t)(w + 1) * mb);
      M.linearized_jacobians.assign(b->prior_J + (size_t)w * mp * mp, b->prior_J + (size_t)(w + 1) * mp * mp);
      M.linearized_residuals.assign(b->prior_r + (size_t)w * mp, b->prior_r + (size_t)(w + 1) * mp);
      M.keep_block_data.assign(b->prior_x0 + (size_t)w * mb * 9, b->prior_x0 + (size_t)(w + 1) * mb * 9);
    }
  });
}

// Estimator::marshal into caller arrays sized like one window of a batch with the WindowTables strides and max_samp
int hs_marshal(void* hp, int max_samp, double* pose, double* speedbias, double* ex_pose, double* inv_depth, int32_t* n_feat, int32_t* feat_start,
               int32_t* feat_nobs, int32_t* feat_obs_begin, double* obs_xy, int32_t* imu_n, double* imu_dt, double* imu_acc, double* imu_gyr,
               double* imu_lin_ba, double* imu_lin_bg, int32_t* feat_id) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] {
    WindowTables t;
    H.est.marshal(t, [](const FeaturePerId& f) { return 1.0 / f.estimated_depth; });
    if (t.max_samp > max_samp) throw Error(AVM_ERR_CAPACITY, "max_samp");
    auto cp = [](auto& v, auto* dst) { std::copy(v.begin(), v.end(), dst); };
    cp(t.pose, pose), cp(t.speedbias, speedbias), cp(t.ex_pose, ex_pose), cp(t.inv_depth, inv_depth);
    *n_feat = t.n_feat;
    cp(t.feat_start, feat_start), cp(t.feat_nobs, feat_nobs), cp(t.feat_obs_begin, feat_obs_begin), cp(t.obs_xy, obs_xy), cp(t.imu_n, imu_n);
    cp(t.feat_id, feat_id);
    const int S = t.max_samp;
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
      for (int s = 0; s < S; s++) imu_dt[j * max_samp + s] = t.imu_dt[j * S + s];
      for (int s = 0; s < (S + 1) * 3; s++)
        imu_acc[j * (max_samp + 1) * 3 + s] = t.imu_acc[j * (S + 1) * 3 + s], imu_gyr[j * (max_samp + 1) * 3 + s] = t.imu_gyr[j * (S + 1) * 3 + s];
    }
    cp(t.imu_lin_ba, imu_lin_ba), cp(t.imu_lin_bg, imu_lin_bg);
  });
}

int hs_set_flags(void* hp, int solver_flag, int marginalization_flag, int max_num_iterations) {
  Host& H = *static_cast<Host*>(hp);
  H.est.solver_flag = solver_flag ? Estimator::NON_LINEAR : Estimator::INITIAL;
  H.est.marginalization_flag = marginalization_flag ? Estimator::MARGIN_SECOND_NEW : Estimator::MARGIN_OLD;
  if (max_num_iterations > 0) H.est.options.max_num_iterations = max_num_iterations;
  return 0;
}

int hs_set_options(void* hp, const avm_options* o) {
  static_cast<Host*>(hp)->est.options = *o;
  return 0;
}

int hs_optimization(void* hp) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] {
    Stopwatch sw;
    H.est.optimization();
  });
}

int hs_triangulate(void* hp, double init_depth) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] { H.est.triangulate(init_depth); });
}

// read the members back: states, per-feature depth / solve_flag of the features that pass the filter, the prior, the summary
int hs_get_state(void* hp, double* pose, double* speedbias, double* ex_pose, int32_t* n_feat, double* est_depth, int32_t* solve_flag,
                 avm_solve_summary* summary) {
  Host& H = *static_cast<Host*>(hp);
  Estimator& E = H.est;
  for (int i = 0; i < AVM_NFRAMES; i++) {
    const double p[7] = {E.Ps[i][0], E.Ps[i][1], E.Ps[i][2], E.Rs[i].x, E.Rs[i].y, E.Rs[i].z, E.Rs[i].w};
    std::copy(p, p + 7, pose + 7 * i);
    for (int k = 0; k < 3; k++) speedbias[9 * i + k] = E.Vs[i][k], speedbias[9 * i + 3 + k] = E.Bas[i][k], speedbias[9 * i + 6 + k] = E.Bgs[i][k];
  }
  const double e[7] = {E.tic[0][0], E.tic[0][1], E.tic[0][2], E.ric[0].x, E.ric[0].y, E.ric[0].z, E.ric[0].w};
  std::copy(e, e + 7, ex_pose);
  int k = 0;
  for (auto& f : E.f_manager.feature) {
    if (!in_problem(f)) continue;
    est_depth[k] = f.estimated_depth, solve_flag[k] = f.solve_flag, k++;
  }
  *n_feat = k;
  if (summary) *summary = E.summary;
  return 0;
}

// the optional members after optimization(): td, relo_Pose, and the pose-graph outputs of double2vector (estimator.cpp:588-604)
// out[0] td | 1..7 relo_Pose | 8 drift_correct_yaw | 9..11 drift_correct_t | 12..14 relo_relative_t | 15..18 relo_relative_q | 19 relo_relative_yaw
// | 20 relocalization_info | 21 failure_occur
int hs_get_extras(void* hp, double* out) {
  const Estimator& E = static_cast<Host*>(hp)->est;
  out[0] = E.td;
  std::copy(E.relo_Pose, E.relo_Pose + 7, out + 1);
  out[8] = E.drift_correct_yaw;
  for (int k = 0; k < 3; k++) out[9 + k] = E.drift_correct_t[k], out[12 + k] = E.relo_relative_t[k];
  out[15] = E.relo_relative_q.x, out[16] = E.relo_relative_q.y, out[17] = E.relo_relative_q.z, out[18] = E.relo_relative_q.w;
  out[19] = E.relo_relative_yaw, out[20] = E.relocalization_info ? 1.0 : 0.0, out[21] = E.failure_occur ? 1.0 : 0.0;
  return 0;
}

int hs_get_prior(void* hp, int32_t* n, int32_t* nblk, int32_t* kind, int32_t* frame, double* J, double* r, double* x0) {
  const MarginalizationInfo& M = static_cast<Host*>(hp)->est.last_marginalization_info;
  *n = M.n, *nblk = M.nblk;
  if (M.n > 0) {
    std::copy(M.blk_kind.begin(), M.blk_kind.end(), kind), std::copy(M.blk_frame.begin(), M.blk_frame.end(), frame);
    std::copy(M.linearized_jacobians.begin(), M.linearized_jacobians.end(), J);
    std::copy(M.linearized_residuals.begin(), M.linearized_residuals.end(), r);
    std::copy(M.keep_block_data.begin(), M.keep_block_data.end(), x0);
  }
  return 0;
}

int hs_propagate(void* hp) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] { H.est.propagateNewestFrame(); });
}

int hs_slide_window(void* hp, double init_depth) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] { H.est.slideWindow(init_depth); });
}

// every feature of f_manager in list order: id, start_frame, number of observations, estimated_depth, first / last point
int hs_dump_features(void* hp, int cap, int32_t* id, int32_t* start, int32_t* nobs, double* depth, double* first_xy, double* last_xy) {
  Host& H = *static_cast<Host*>(hp);
  int k = 0;
  for (const auto& f : H.est.f_manager.feature) {
    if (k >= cap) return -1;
    id[k] = f.feature_id, start[k] = f.start_frame, nobs[k] = (int)f.feature_per_frame.size(), depth[k] = f.estimated_depth;
    first_xy[2 * k] = f.feature_per_frame.front().point[0], first_xy[2 * k + 1] = f.feature_per_frame.front().point[1];
    last_xy[2 * k] = f.feature_per_frame.back().point[0], last_xy[2 * k + 1] = f.feature_per_frame.back().point[1];
    k++;
  }
  return k;
}

// the raw IMU buffers of interval j (pre_integrations[j + 1]): returns the sample count
int hs_dump_imu(void* hp, int j, int cap, double* dt, double* acc, double* gyr, double* lin) {
  const IntegrationBase& p = static_cast<Host*>(hp)->est.pre_integrations[j + 1];
  const int n = (int)p.dt_buf.size();
  if (n > cap) return -1;
  for (int k = 0; k < 3; k++) acc[k] = p.linearized_acc[k], gyr[k] = p.linearized_gyr[k], lin[k] = p.linearized_ba[k], lin[3 + k] = p.linearized_bg[k];
  for (int s = 0; s < n; s++) {
    dt[s] = p.dt_buf[s];
    for (int k = 0; k < 3; k++) acc[(s + 1) * 3 + k] = p.acc_buf[s][k], gyr[(s + 1) * 3 + k] = p.gyr_buf[s][k];
  }
  return n;
}

// ---- selector ----------------------------------------------------------------------------------------------------
// cam = fx fy cx cy k1 k2 p1 p2
int hs_sel_create(void* hp, const double* cam, int width, int height, int horizon) {
  Host& H = *static_cast<Host*>(hp);
  PinholeCamera c;
  c.fx = cam[0], c.fy = cam[1], c.cx = cam[2], c.cy = cam[3], c.k1 = cam[4], c.k2 = cam[5], c.p1 = cam[6], c.p2 = cam[7];
  c.image_width = width, c.image_height = height;
  H.sel.reset(new FeatureSelector(H.est, c, horizon));
  return 0;
}

int hs_sel_set_parameters(void* hp, double accVar, double accBiasVar, int enable, int maxFeatures, int initThresh, int useGT) {
  static_cast<Host*>(hp)->sel->setParameters(accVar, accBiasVar, enable != 0, maxFeatures, initThresh, useGT != 0);
  return 0;
}

int hs_sel_set_ground_truth(void* hp, const double* rows17, int n) {
  Host& H = *static_cast<Host*>(hp);
  return guarded([&] { H.sel->setGroundTruth(rows17, n); });
}

// P3 Q4(xyzw) V3 a3 w3 Ba3
int hs_sel_set_next_state(void* hp, double stamp, const double* P, const double* Q, const double* V, const double* a, const double* w,
                          const double* Ba) {
  auto v3 = [](const double* p) { return Vector3d{p[0], p[1], p[2]}; };
  static_cast<Host*>(hp)->sel->setNextStateFromImuPropagation(stamp, v3(P), Quaterniond{Q[0], Q[1], Q[2], Q[3]}, v3(V), v3(a), v3(w), v3(Ba));
  return 0;
}

// image in: n features (ascending or not: it is a std::map), one camera each.  image out: the ids select() left in `image`.
// returns the number of selected ids (>= 0) or a negative status; *n_returned_pairs == 0 when select() returned {} (disabled).
int hs_sel_select(void* hp, int n, const int32_t* ids, const double* rows8, double stamp, int nrImu, int32_t* image_out, int32_t* n_image_out,
                  int32_t* tracked, int32_t* n_tracked, int32_t* selected, int32_t cap) {
  Host& H = *static_cast<Host*>(hp);
  int n_sel = 0;
  const int rc = guarded([&] {
    image_t image;
    for (int i = 0; i < n; i++) {
      std::array<double, 8> v;
      std::copy(rows8 + 8 * i, rows8 + 8 * i + 8, v.begin());
      image[ids[i]].emplace_back(0, v);
    }
    std::pair<std::vector<int>, std::vector<int>> ret;
    {
      Stopwatch sw;
      ret = H.sel->select(image, stamp, nrImu);
    }
    if ((int)image.size() > cap || (int)ret.first.size() > cap || (int)ret.second.size() > cap) throw Error(AVM_ERR_CAPACITY, "shim output capacity");
    int k = 0;
    for (const auto& f : image) image_out[k++] = f.first;
    *n_image_out = k;
    std::copy(ret.first.begin(), ret.first.end(), tracked);
    *n_tracked = (int)ret.first.size();
    std::copy(ret.second.begin(), ret.second.end(), selected);
    n_sel = (int)ret.second.size();
  });
  return rc < 0 ? rc : n_sel;
}

// wall time of the three device calls inside the last select(): horizon, depth cloud (incl. the marshal of the window), the greedy selection
int hs_sel_last_parts_ms(void* hp, double* out3) {
  const auto& s = *static_cast<Host*>(hp)->sel;
  out3[0] = s.lastHorizonMs_, out3[1] = s.lastCloudMs_, out3[2] = s.lastSelectMs_;
  return 0;
}

int hs_sel_gt_seek(void* hp) { return static_cast<Host*>(hp)->sel->groundTruthSeek(); }

int hs_sel_last_feature_id(void* hp) { return static_cast<Host*>(hp)->sel->lastFeatureId_; }

}  // extern "C"
