// avm_host.hpp — header-only C++17 host side above the C ABI of avm.h.
//
// The reference's two hot paths are member functions of two stateful C++ objects.  This header
// restates those objects with their member names and call surfaces on plain STL data (no ROS, Eigen,
// Ceres, OpenCV), so that a host that already drives the reference can switch the two calls
//
//     estimator.optimization();                                   vins_estimator/src/estimator.h:47
//     f_selector.select(image, header, nrImuMeasurements);        vins_estimator/src/feature_selector.h:49-50
//
// to the GPU without touching the code around them.  Everything numerical happens behind avm.h; what
// is left here is the index work the reference also does on the host: vector2double / double2vector
// (estimator.cpp:477-610, minus the gauge fix, which the device applies), the feature-list filter and
// order of estimator.cpp:712-755, the raw IMU buffers of IntegrationBase (integration_base.h:205-207),
// the prior hand-off (estimator.cpp:817-990), and the selector's bookkeeping (feature_selector.cpp:74-202,
// 208-219, 38-70).
//
// Error behaviour: the reference has none on these paths (ROS_BREAK / assert).  Every failing ABI call
// throws avm_host::Error carrying the avm_status and avm_last_error(); there is no CPU fallback.
//
// Rotations are held as quaternions (x, y, z, w) where the reference holds Matrix3d (Rs, ric): the
// arrays it hands to the solver are quaternions anyway (estimator.cpp:484-488).
#ifndef AVM_HOST_HPP_
#define AVM_HOST_HPP_

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <iterator>
#include <list>
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "avm.h"

namespace avm_host {

using Vector3d = std::array<double, 3>;
struct Quaterniond {
  double x = 0, y = 0, z = 0, w = 1;
};

namespace detail {
// the handful of rotation helpers double2vector's relocalization outputs need (utility/utility.h:66-108,124-141); unit quaternions
inline Quaterniond conj(const Quaterniond& q) { return Quaterniond{-q.x, -q.y, -q.z, q.w}; }
inline Quaterniond mul(const Quaterniond& a, const Quaterniond& b) {
  return Quaterniond{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
                     a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
inline Vector3d rotate(const Quaterniond& q, const Vector3d& v) {  // Eigen: v + w t + u x t, t = 2 u x v
  const double tx = 2 * (q.y * v[2] - q.z * v[1]), ty = 2 * (q.z * v[0] - q.x * v[2]), tz = 2 * (q.x * v[1] - q.y * v[0]);
  return Vector3d{v[0] + q.w * tx + (q.y * tz - q.z * ty), v[1] + q.w * ty + (q.z * tx - q.x * tz), v[2] + q.w * tz + (q.x * ty - q.y * tx)};
}
inline double yaw_deg(const Quaterniond& q) {  // Utility::R2ypr(R).x(): atan2(R(1,0), R(0,0)) in degrees
  const double r10 = 2 * (q.x * q.y + q.z * q.w), r00 = 1 - 2 * (q.y * q.y + q.z * q.z);
  return std::atan2(r10, r00) / M_PI * 180.0;
}
inline Vector3d rotate_yaw(double yaw_degrees, const Vector3d& v) {  // Utility::ypr2R(Vector3d(yaw, 0, 0)) * v
  const double y = yaw_degrees / 180.0 * M_PI, c = std::cos(y), s2 = std::sin(y);
  return Vector3d{c * v[0] - s2 * v[1], s2 * v[0] + c * v[1], v[2]};
}
inline double normalize_angle(double a) {  // Utility::normalizeAngle (utility.h:124-141), degrees
  const double two_pi = 360.0;
  if (a > 0) return a - two_pi * std::floor((a + 180.0) / two_pi);
  return a + two_pi * std::floor((-a + 180.0) / two_pi);
}
}  // namespace detail

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& what) : std::runtime_error(what), status(s) {}
};

// one avm_ctx per host thread (the reference calls both paths from process(), estimator_node.cpp:340,360).
// The device is opened by the first call that needs it; without a usable HIP device that call throws.
class Context {
 public:
  explicit Context(int device = 0, int max_windows = 1, int max_problems = 1) {
    cfg_.device = device, cfg_.max_windows = max_windows, cfg_.max_problems = max_problems, cfg_.abi_version = AVM_ABI_VERSION;
  }
  ~Context() {
    if (h_) avm_destroy(h_);
  }
  Context(const Context&) = delete;
  Context& operator=(const Context&) = delete;
  avm_ctx* get() {
    if (!h_) {
      const int rc = avm_create(&cfg_, &h_);
      if (rc != AVM_OK) {
        h_ = nullptr;
        throw Error(rc, "avm_create failed (" + std::to_string(rc) + "): no usable HIP device, and there is no CPU path");
      }
    }
    return h_;
  }
  void check(int rc, const char* call) const {
    if (rc != AVM_OK) throw Error(rc, std::string(call) + " failed (" + std::to_string(rc) + "): " + (h_ ? avm_last_error(h_) : ""));
  }

 private:
  avm_config cfg_{};
  avm_ctx* h_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------------
// FeatureManager read/write side used by the two paths (feature_manager.h:18-66)
// ---------------------------------------------------------------------------------------------------------
struct FeaturePerFrame {
  Vector3d point{0, 0, 1};           // normalized plane, z == 1
  std::array<double, 2> uv{0, 0};
  std::array<double, 2> velocity{0, 0};
  double cur_td = 0;
};

struct FeaturePerId {
  int feature_id = 0;
  int start_frame = 0;
  std::vector<FeaturePerFrame> feature_per_frame;
  int used_num = 0;
  double estimated_depth = -1.0;  // feature_manager.h:61
  int solve_flag = 0;             // 0 not solved yet, 1 ok, 2 failed
  FeaturePerId() = default;
  FeaturePerId(int id, int start) : feature_id(id), start_frame(start) {}
  int endFrame() const { return start_frame + (int)feature_per_frame.size() - 1; }
};

// the filter of estimator.cpp:715 / feature_manager.cpp:34,147,190,212
inline bool in_problem(const FeaturePerId& f) { return f.feature_per_frame.size() >= 2 && f.start_frame < AVM_WINDOW_SIZE - 2; }

class FeatureManager {
 public:
  std::list<FeaturePerId> feature;

  int getFeatureCount() {  // feature_manager.cpp:28-42
    int n = 0;
    for (auto& f : feature) {
      f.used_num = (int)f.feature_per_frame.size();
      if (in_problem(f)) n++;
    }
    return n;
  }
  std::vector<double> getDepthVector() {  // feature_manager.cpp:184-200: inverse depths in list order
    std::vector<double> v;
    for (auto& f : feature) {
      f.used_num = (int)f.feature_per_frame.size();
      if (in_problem(f)) v.push_back(1.0 / f.estimated_depth);
    }
    return v;
  }
  void setDepth(const double* inv_depth) {  // feature_manager.cpp:141-159
    int k = 0;
    for (auto& f : feature) {
      f.used_num = (int)f.feature_per_frame.size();
      if (!in_problem(f)) continue;
      f.estimated_depth = 1.0 / inv_depth[k++];
      f.solve_flag = f.estimated_depth < 0 ? 2 : 1;
    }
  }
  void removeFailures() {  // feature_manager.cpp:161-171
    feature.remove_if([](const FeaturePerId& f) { return f.solve_flag == 2; });
  }
};

// the raw sample buffers of IntegrationBase (integration_base.h:13-36,205-207); the integration itself runs on the device
struct IntegrationBase {
  Vector3d linearized_acc{}, linearized_gyr{}, linearized_ba{}, linearized_bg{};
  std::vector<double> dt_buf;
  std::vector<Vector3d> acc_buf, gyr_buf;
  IntegrationBase() = default;
  IntegrationBase(const Vector3d& acc_0, const Vector3d& gyr_0, const Vector3d& ba, const Vector3d& bg)
      : linearized_acc(acc_0), linearized_gyr(gyr_0), linearized_ba(ba), linearized_bg(bg) {}
  void push_back(double dt, const Vector3d& acc, const Vector3d& gyr) {
    dt_buf.push_back(dt), acc_buf.push_back(acc), gyr_buf.push_back(gyr);
  }
};

// last_marginalization_info + last_marginalization_parameter_blocks as plain data (marginalization_factor.h:49-72):
// the kept blocks are (kind, frame) pairs instead of addresses, frame already carrying the addr_shift.
struct MarginalizationInfo {
  int n = 0, nblk = 0;
  std::vector<int32_t> blk_kind, blk_frame;  // [max_pblk], nblk used
  std::vector<double> linearized_jacobians;  // [max_prior][max_prior], n x n used
  std::vector<double> linearized_residuals;  // [max_prior]
  std::vector<double> keep_block_data;       // [max_pblk][9]
  bool valid() const { return n > 0; }
};

// flat tables of one window in the layout of avm_window_batch; owns the storage the batch points into
struct WindowTables {
  static constexpr int MAX_FEAT = 150, MAX_OBS = 1650, MAX_PRIOR = 96, MAX_PBLK = 16;
  int32_t n_feat = 0, prior_n = 0, prior_nblk = 0;
  int max_samp = 1;
  std::vector<double> pose, speedbias, ex_pose, inv_depth, obs_xy, imu_dt, imu_acc, imu_gyr, imu_lin_ba, imu_lin_bg;
  std::vector<int32_t> feat_start, feat_nobs, feat_obs_begin, imu_n;
  std::vector<int32_t> feat_id;  // feature_id per table row (host side only)
  // optional members of the problem (avm_window_batch): handed over only when the matching use_* flag is set
  bool use_td = false, use_relo = false, use_failure = false;
  std::vector<double> obs_vel_td, relo_xy;
  std::vector<int32_t> relo_feat;
  double td = 0;
  int32_t relo_n = 0, relo_frame = 0, failure_occur = 0;
  std::array<double, 7> relo_pose{{0, 0, 0, 0, 0, 0, 1}}, last_pose0{{0, 0, 0, 0, 0, 0, 1}};
  avm_window_batch batch(const MarginalizationInfo* prior) {
    avm_window_batch b{};
    if (use_td) b.obs_vel_td = obs_vel_td.data(), b.td = &td;
    if (use_relo) b.relo_n = &relo_n, b.relo_frame = &relo_frame, b.relo_feat = relo_feat.data(), b.relo_xy = relo_xy.data(), b.relo_pose = relo_pose.data();
    if (use_failure) b.failure_occur = &failure_occur, b.last_pose0 = last_pose0.data();
    b.n_windows = 1, b.max_feat = MAX_FEAT, b.max_obs = MAX_OBS, b.max_samp = max_samp, b.max_prior = MAX_PRIOR, b.max_pblk = MAX_PBLK;
    b.pose = pose.data(), b.speedbias = speedbias.data(), b.ex_pose = ex_pose.data(), b.inv_depth = inv_depth.data();
    b.n_feat = &n_feat, b.feat_start = feat_start.data(), b.feat_nobs = feat_nobs.data(), b.feat_obs_begin = feat_obs_begin.data();
    b.obs_xy = obs_xy.data();
    b.imu_n = imu_n.data(), b.imu_dt = imu_dt.data(), b.imu_acc = imu_acc.data(), b.imu_gyr = imu_gyr.data();
    b.imu_lin_ba = imu_lin_ba.data(), b.imu_lin_bg = imu_lin_bg.data();
    prior_n = prior && prior->valid() ? prior->n : 0;
    prior_nblk = prior_n ? prior->nblk : 0;
    b.prior_n = &prior_n, b.prior_nblk = &prior_nblk;
    if (prior_n) {
      b.prior_blk_kind = prior->blk_kind.data(), b.prior_blk_frame = prior->blk_frame.data();
      b.prior_J = prior->linearized_jacobians.data(), b.prior_r = prior->linearized_residuals.data(), b.prior_x0 = prior->keep_block_data.data();
    } else {
      zero_i_.assign(MAX_PBLK, 0), zero_d_.assign((size_t)MAX_PRIOR * MAX_PRIOR, 0.0);
      b.prior_blk_kind = b.prior_blk_frame = zero_i_.data();
      b.prior_J = b.prior_r = b.prior_x0 = zero_d_.data();
    }
    return b;
  }

 private:
  std::vector<int32_t> zero_i_;
  std::vector<double> zero_d_;
};

// ---------------------------------------------------------------------------------------------------------
// Estimator: the members optimization() reads and writes (estimator.h:62-115), and HP-A itself
// ---------------------------------------------------------------------------------------------------------
class Estimator {
 public:
  enum SolverFlag { INITIAL, NON_LINEAR };                            // estimator.h:27-31
  enum MarginalizationFlag { MARGIN_OLD = 0, MARGIN_SECOND_NEW = 1 };  // estimator.h:33-37

  explicit Estimator(Context& ctx) : ctx_(ctx) {
    avm_default_options(&options);
    clearState();
  }

  void clearState() {  // estimator.cpp:35-83, the members of this path
    for (int i = 0; i <= AVM_WINDOW_SIZE; i++) {
      Ps[i] = Vs[i] = Bas[i] = Bgs[i] = Vector3d{0, 0, 0};
      Rs[i] = Quaterniond{};
      pre_integrations[i] = IntegrationBase{};
    }
    tic[0] = Vector3d{0, 0, 0}, ric[0] = Quaterniond{}, td = 0;
    f_manager.feature.clear();
    solver_flag = INITIAL, marginalization_flag = MARGIN_OLD;
    last_marginalization_info = MarginalizationInfo{};
    summary = avm_solve_summary{};
  }

  SolverFlag solver_flag = INITIAL;
  MarginalizationFlag marginalization_flag = MARGIN_OLD;
  Vector3d Ps[AVM_NFRAMES], Vs[AVM_NFRAMES], Bas[AVM_NFRAMES], Bgs[AVM_NFRAMES];
  Quaterniond Rs[AVM_NFRAMES];
  Vector3d tic[1];
  Quaterniond ric[1];
  double td = 0;
  FeatureManager f_manager;
  IntegrationBase pre_integrations[AVM_NFRAMES];  // [j] spans frames j-1 .. j; [0] is unused by this path (estimator.cpp:702-709)
  MarginalizationInfo last_marginalization_info;
  avm_options options;        // NUM_ITERATIONS, noise densities, G, ... (parameters.cpp); marginalization_flag is set per call
  double SOLVER_TIME = 0.0;   // parameters.cpp:97 / euroc_config.yaml:54 (0.04 s); 0 = no wall-clock cap (the default here: a cap makes
                              // results depend on the clock, and one window takes ~1 ms on the device against the reference's 32-40 ms)
  avm_solve_summary summary;  // ceres::Solver::Summary subset of the last optimization()

  double para_Pose[AVM_NFRAMES][AVM_SIZE_POSE];
  double para_SpeedBias[AVM_NFRAMES][AVM_SIZE_SPEEDBIAS];
  double para_Ex_Pose[1][AVM_SIZE_POSE];
  double para_Td[1][1] = {{0}};

  // relocalization (estimator.h:117-131; setReloFrame estimator.cpp:1120-1141): match_points[k] = (x, y, feature_id)
  bool relocalization_info = false;
  int relo_frame_local_index = 0;
  std::vector<Vector3d> match_points;
  double relo_Pose[AVM_SIZE_POSE] = {0, 0, 0, 0, 0, 0, 1};
  Vector3d prev_relo_t{0, 0, 0};
  Quaterniond prev_relo_r{};
  // ... and what double2vector leaves for the pose graph (estimator.cpp:588-604)
  double drift_correct_yaw = 0;       // drift_correct_r = ypr2R(drift_correct_yaw, 0, 0)
  Vector3d drift_correct_t{0, 0, 0}, relo_relative_t{0, 0, 0};
  Quaterniond relo_relative_q{};
  double relo_relative_yaw = 0;
  // failure_occur re-anchoring of double2vector (estimator.cpp:526-531; last_R0 / last_P0 are set by processImage :170-171,207-208)
  bool failure_occur = false;
  Quaterniond last_R0{};
  Vector3d last_P0{0, 0, 0};

  void vector2double() {  // estimator.cpp:477-519 (para_Feature is marshalled with the feature tables)
    for (int i = 0; i <= AVM_WINDOW_SIZE; i++) {
      const double p[7] = {Ps[i][0], Ps[i][1], Ps[i][2], Rs[i].x, Rs[i].y, Rs[i].z, Rs[i].w};
      std::copy(p, p + 7, para_Pose[i]);
      for (int k = 0; k < 3; k++) para_SpeedBias[i][k] = Vs[i][k], para_SpeedBias[i][3 + k] = Bas[i][k], para_SpeedBias[i][6 + k] = Bgs[i][k];
    }
    const double e[7] = {tic[0][0], tic[0][1], tic[0][2], ric[0].x, ric[0].y, ric[0].z, ric[0].w};
    std::copy(e, e + 7, para_Ex_Pose[0]);
  }

  // The loop bodies of double2vector (estimator.cpp:548-587).  The yaw / position gauge fix of :521-546 is already in the
  // arrays the device returns, so this is a copy.
  void double2vector() {
    for (int i = 0; i <= AVM_WINDOW_SIZE; i++) {
      Ps[i] = Vector3d{para_Pose[i][0], para_Pose[i][1], para_Pose[i][2]};
      Rs[i] = Quaterniond{para_Pose[i][3], para_Pose[i][4], para_Pose[i][5], para_Pose[i][6]};
      for (int k = 0; k < 3; k++) Vs[i][k] = para_SpeedBias[i][k], Bas[i][k] = para_SpeedBias[i][3 + k], Bgs[i][k] = para_SpeedBias[i][6 + k];
    }
    tic[0] = Vector3d{para_Ex_Pose[0][0], para_Ex_Pose[0][1], para_Ex_Pose[0][2]};
    ric[0] = Quaterniond{para_Ex_Pose[0][3], para_Ex_Pose[0][4], para_Ex_Pose[0][5], para_Ex_Pose[0][6]};
  }

  // para_* + f_manager.feature + pre_integrations[1..10] -> the tables of avm_window_batch.
  // inv_depth_of decides what goes into inv_depth for a feature (the solve, triangulate and the selector's cloud differ).
  template <class InvDepth>
  void marshal(WindowTables& t, InvDepth inv_depth_of) {
    vector2double();
    t.pose.assign(&para_Pose[0][0], &para_Pose[0][0] + AVM_NFRAMES * 7);
    t.speedbias.assign(&para_SpeedBias[0][0], &para_SpeedBias[0][0] + AVM_NFRAMES * 9);
    t.ex_pose.assign(&para_Ex_Pose[0][0], &para_Ex_Pose[0][0] + 7);
    t.inv_depth.assign(WindowTables::MAX_FEAT, 0.0);
    t.feat_start.assign(WindowTables::MAX_FEAT, 0), t.feat_nobs.assign(WindowTables::MAX_FEAT, 0), t.feat_obs_begin.assign(WindowTables::MAX_FEAT, 0);
    t.feat_id.clear();
    t.obs_xy.assign((size_t)WindowTables::MAX_OBS * 2, 0.0);
    if (t.use_td) t.obs_vel_td.assign((size_t)WindowTables::MAX_OBS * 4, 0.0);
    int e = 0, o = 0;
    for (auto& f : f_manager.feature) {
      f.used_num = (int)f.feature_per_frame.size();
      if (!in_problem(f)) continue;
      if (e >= WindowTables::MAX_FEAT || o + f.used_num > WindowTables::MAX_OBS)
        throw Error(AVM_ERR_CAPACITY, "more than 150 features / 1650 observations pass the filter of estimator.cpp:715");
      t.feat_start[e] = f.start_frame, t.feat_nobs[e] = f.used_num, t.feat_obs_begin[e] = o;
      t.inv_depth[e] = inv_depth_of(f);
      t.feat_id.push_back(f.feature_id);
      for (const auto& pf : f.feature_per_frame) {
        t.obs_xy[2 * o] = pf.point[0], t.obs_xy[2 * o + 1] = pf.point[1];
        if (t.use_td) {  // what ProjectionTdFactor's constructor takes per observation (estimator.cpp:734-736)
          double* a = &t.obs_vel_td[4 * (size_t)o];
          a[0] = pf.velocity[0], a[1] = pf.velocity[1], a[2] = pf.cur_td, a[3] = pf.uv[1];
        }
        o++;
      }
      e++;
    }
    t.n_feat = e;
    size_t S = 1;
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) S = std::max(S, pre_integrations[j + 1].dt_buf.size());
    fill_imu(t, S);
  }

  // The dead-reckoning of Estimator::processIMU (estimator.cpp:100-107) for the newest frame, all samples of the last
  // interval at once: Ps / Rs / Vs [WINDOW_SIZE] (holding the previous frame's values, as slideWindow() leaves them) are
  // carried through pre_integrations[WINDOW_SIZE]'s raw buffers with Bas / Bgs [WINDOW_SIZE] and options.g.
  void propagateNewestFrame() {
    WindowTables t;
    marshal(t, [](const FeaturePerId&) { return 1.0; });
    avm_window_batch b = t.batch(nullptr);
    ctx_.check(avm_imu_propagate_batch(ctx_.get(), AVM_MEM_HOST, &b, options.g), "avm_imu_propagate_batch");
    std::copy(t.pose.begin(), t.pose.end(), &para_Pose[0][0]);
    std::copy(t.speedbias.begin(), t.speedbias.end(), &para_SpeedBias[0][0]);
    double2vector();
  }

  // FeatureManager::triangulate(Ps, tic, ric) (feature_manager.cpp:202-257; call site estimator.cpp:470)
  void triangulate(double init_depth = 5.0) {
    WindowTables t;
    marshal(t, [](const FeaturePerId& f) { return f.estimated_depth > 0 ? 1.0 / f.estimated_depth : -1.0; });
    avm_window_batch b = t.batch(nullptr);
    ctx_.check(avm_triangulate_batch(ctx_.get(), AVM_MEM_HOST, &b, init_depth), "avm_triangulate_batch");
    int k = 0;
    for (auto& f : f_manager.feature) {
      if (!in_problem(f)) continue;
      if (!(f.estimated_depth > 0)) f.estimated_depth = 1.0 / t.inv_depth[k];
      k++;
    }
  }

  // HP-A (estimator.cpp:661-994): solve the window in place, leave the new prior in last_marginalization_info.
  void optimization() {
    options.marginalization_flag = marginalization_flag == MARGIN_OLD ? AVM_MARGIN_OLD : AVM_MARGIN_SECOND_NEW;
    // estimator.cpp:803-806: max_solver_time_in_seconds = SOLVER_TIME * 4 / 5 under MARGIN_OLD, SOLVER_TIME otherwise
    options.max_solver_time_s = SOLVER_TIME > 0.0 ? (marginalization_flag == MARGIN_OLD ? SOLVER_TIME * 4.0 / 5.0 : SOLVER_TIME) : 0.0;
    WindowTables t;
    t.use_td = options.estimate_td != 0, t.use_relo = relocalization_info, t.use_failure = failure_occur;
    marshal(t, [](const FeaturePerId& f) { return 1.0 / f.estimated_depth; });  // getDepthVector()
    t.td = para_Td[0][0] = td;                                                  // vector2double, estimator.cpp:517-518
    if (t.use_relo) {
      // the loop of estimator.cpp:766-790 over f_manager.feature and match_points (both ascending in feature id), resolved
      // into feature indices of the tables
      t.relo_feat.assign(WindowTables::MAX_FEAT, 0), t.relo_xy.assign((size_t)WindowTables::MAX_FEAT * 2, 0.0);
      size_t retrive_feature_index = 0;
      int feature_index = -1, k = 0;
      for (auto& it_per_id : f_manager.feature) {
        if (!in_problem(it_per_id)) continue;
        ++feature_index;
        if (it_per_id.start_frame <= relo_frame_local_index) {
          while (retrive_feature_index < match_points.size() && (int)match_points[retrive_feature_index][2] < it_per_id.feature_id) retrive_feature_index++;
          if (retrive_feature_index < match_points.size() && (int)match_points[retrive_feature_index][2] == it_per_id.feature_id) {
            t.relo_feat[k] = feature_index;
            t.relo_xy[2 * k] = match_points[retrive_feature_index][0], t.relo_xy[2 * k + 1] = match_points[retrive_feature_index][1];
            k++, retrive_feature_index++;
          }
        }
      }
      t.relo_n = k, t.relo_frame = relo_frame_local_index;
      std::copy(relo_Pose, relo_Pose + 7, t.relo_pose.begin());
      // (k == 0: no factor references relo_Pose and the solve does not move it, but double2vector still takes it through the
      //  gauge fix of the window, estimator.cpp:590-596 - the device does that whenever the relocalization arrays are there)
    }
    if (t.use_failure) {
      t.failure_occur = 1;
      t.last_pose0 = {last_P0[0], last_P0[1], last_P0[2], last_R0.x, last_R0.y, last_R0.z, last_R0.w};
    }
    avm_window_batch b = t.batch(&last_marginalization_info);

    MarginalizationInfo next;
    next.blk_kind.assign(WindowTables::MAX_PBLK, 0), next.blk_frame.assign(WindowTables::MAX_PBLK, 0);
    next.linearized_jacobians.assign((size_t)WindowTables::MAX_PRIOR * WindowTables::MAX_PRIOR, 0.0);
    next.linearized_residuals.assign(WindowTables::MAX_PRIOR, 0.0);
    next.keep_block_data.assign((size_t)WindowTables::MAX_PBLK * 9, 0.0);
    int32_t out_n = 0, out_nblk = 0;
    avm_prior_out po{};
    po.max_prior = WindowTables::MAX_PRIOR, po.max_pblk = WindowTables::MAX_PBLK;
    po.n = &out_n, po.nblk = &out_nblk, po.blk_kind = next.blk_kind.data(), po.blk_frame = next.blk_frame.data();
    po.J = next.linearized_jacobians.data(), po.r = next.linearized_residuals.data(), po.x0 = next.keep_block_data.data();

    ctx_.check(avm_window_solve_batch(ctx_.get(), &options, AVM_MEM_HOST, &b, &po, &summary), "avm_window_solve_batch");

    std::copy(t.pose.begin(), t.pose.end(), &para_Pose[0][0]);
    std::copy(t.speedbias.begin(), t.speedbias.end(), &para_SpeedBias[0][0]);
    std::copy(t.ex_pose.begin(), t.ex_pose.end(), &para_Ex_Pose[0][0]);
    double2vector();
    f_manager.setDepth(t.inv_depth.data());
    if (options.estimate_td) td = para_Td[0][0] = t.td;  // estimator.cpp:585-586
    failure_occur = false;                                 // estimator.cpp:530
    if (relocalization_info) {
      // estimator.cpp:588-604: relo_t / relo_r come back gauge-fixed from the device (also when no feature matched: the
      // loop frame then did not move in the solve, the window's yaw / origin correction applies to it all the same); the
      // relative pose of the loop frame and the drift correction are host arithmetic on them
      if (t.use_relo) std::copy(t.relo_pose.begin(), t.relo_pose.end(), relo_Pose);
      const Vector3d relo_t{relo_Pose[0], relo_Pose[1], relo_Pose[2]};
      const Quaterniond relo_r{relo_Pose[3], relo_Pose[4], relo_Pose[5], relo_Pose[6]};
      drift_correct_yaw = detail::yaw_deg(prev_relo_r) - detail::yaw_deg(relo_r);
      const Vector3d rt = detail::rotate_yaw(drift_correct_yaw, relo_t);
      drift_correct_t = {prev_relo_t[0] - rt[0], prev_relo_t[1] - rt[1], prev_relo_t[2] - rt[2]};
      const int i = relo_frame_local_index;
      relo_relative_t = detail::rotate(detail::conj(relo_r), Vector3d{Ps[i][0] - relo_t[0], Ps[i][1] - relo_t[1], Ps[i][2] - relo_t[2]});
      relo_relative_q = detail::mul(detail::conj(relo_r), Rs[i]);
      relo_relative_yaw = detail::normalize_angle(detail::yaw_deg(Rs[i]) - detail::yaw_deg(relo_r));
      relocalization_info = false;
    }
    if (out_n >= 0) {  // -1: MARGIN_SECOND_NEW had nothing to drop, the old prior stays (estimator.cpp:926-927)
      next.n = out_n, next.nblk = out_nblk;
      last_marginalization_info = std::move(next);
    }
  }

  // Estimator::slideWindow (estimator.cpp:996-1107) with slideWindowOld / slideWindowNew and the three FeatureManager
  // remove* methods (feature_manager.cpp:275-352), for frame_count == WINDOW_SIZE: the states, the IMU buffers and EVERY
  // feature of f_manager (not only the ones that pass the solve's filter) go through avm_slide_window() and come back.
  // Headers / all_image_frame bookkeeping is the caller's.
  void slideWindow(double init_depth = 5.0) {
    vector2double();
    WindowTables t;
    t.pose.assign(&para_Pose[0][0], &para_Pose[0][0] + AVM_NFRAMES * 7);
    t.speedbias.assign(&para_SpeedBias[0][0], &para_SpeedBias[0][0] + AVM_NFRAMES * 9);
    t.ex_pose.assign(&para_Ex_Pose[0][0], &para_Ex_Pose[0][0] + 7);
    const int nf = (int)f_manager.feature.size();
    size_t n_obs = 0;
    for (const auto& f : f_manager.feature) n_obs += f.feature_per_frame.size();
    const int max_feat = std::max(1, nf), max_obs = (int)std::max<size_t>(1, n_obs);
    t.inv_depth.assign(max_feat, 0.0);
    t.feat_start.assign(max_feat, 0), t.feat_nobs.assign(max_feat, 0), t.feat_obs_begin.assign(max_feat, 0);
    t.obs_xy.assign((size_t)max_obs * 2, 0.0);
    {
      int e = 0, o = 0;
      for (auto it = f_manager.feature.begin(); it != f_manager.feature.end(); ++it, ++e) {
        t.feat_start[e] = it->start_frame, t.feat_nobs[e] = (int)it->feature_per_frame.size(), t.feat_obs_begin[e] = o;
        t.inv_depth[e] = 1.0 / it->estimated_depth;
        for (const auto& pf : it->feature_per_frame) t.obs_xy[2 * o] = pf.point[0], t.obs_xy[2 * o + 1] = pf.point[1], o++;
      }
      t.n_feat = nf;
    }
    // second-new appends interval 10's samples to interval 9: leave room for them
    size_t S = 1;
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) S = std::max(S, pre_integrations[j + 1].dt_buf.size());
    S = std::max(S, pre_integrations[AVM_WINDOW_SIZE - 1].dt_buf.size() + pre_integrations[AVM_WINDOW_SIZE].dt_buf.size());
    fill_imu(t, S);
    avm_window_batch b = t.batch(nullptr);
    b.max_feat = max_feat, b.max_obs = max_obs;
    const int flag = marginalization_flag == MARGIN_OLD ? AVM_MARGIN_OLD : AVM_MARGIN_SECOND_NEW;
    ctx_.check(avm_slide_window(ctx_.get(), AVM_MEM_HOST, &b, flag, solver_flag == NON_LINEAR ? 1 : 0, init_depth), "avm_slide_window");

    std::copy(t.pose.begin(), t.pose.end(), &para_Pose[0][0]);
    std::copy(t.speedbias.begin(), t.speedbias.end(), &para_SpeedBias[0][0]);
    double2vector();
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
      const size_t row0 = (size_t)j * (S + 1);
      auto v3 = [](const double* p) { return Vector3d{p[0], p[1], p[2]}; };
      IntegrationBase p(v3(&t.imu_acc[row0 * 3]), v3(&t.imu_gyr[row0 * 3]), v3(&t.imu_lin_ba[j * 3]), v3(&t.imu_lin_bg[j * 3]));
      for (int s2 = 0; s2 < t.imu_n[j]; s2++) p.push_back(t.imu_dt[j * S + s2], v3(&t.imu_acc[(row0 + s2 + 1) * 3]), v3(&t.imu_gyr[(row0 + s2 + 1) * 3]));
      pre_integrations[j + 1] = std::move(p);
    }
    // features: the roll compacts the tables in list order and never moves an observation to another feature's slots
    std::vector<char> kept(nf, 0);
    std::vector<std::list<FeaturePerId>::iterator> all;
    for (auto it = f_manager.feature.begin(); it != f_manager.feature.end(); ++it) all.push_back(it);
    {
      std::vector<int> index_of(max_obs, -1), first_slot(nf + 1, 0);  // observation slot -> feature, feature -> first slot
      int e = 0, o = 0;
      for (auto it = f_manager.feature.begin(); it != f_manager.feature.end(); ++it, ++e) {
        first_slot[e] = o;
        for (size_t k = 0; k < it->feature_per_frame.size(); k++) index_of[o++] = e;
      }
      for (int k = 0; k < t.n_feat; k++) {
        const int ob = t.feat_obs_begin[k], e0 = index_of[ob];
        FeaturePerId& f = *all[e0];
        kept[e0] = 1;
        auto& v = f.feature_per_frame;
        if (flag == AVM_MARGIN_OLD) {
          if (ob == first_slot[e0] + 1) v.erase(v.begin());                    // lost its first observation
        } else if ((int)v.size() != t.feat_nobs[k]) {
          v.erase(v.begin() + (AVM_WINDOW_SIZE - 1 - f.start_frame));          // lost its observation in frame 9
        }
        f.start_frame = t.feat_start[k];
        f.estimated_depth = 1.0 / t.inv_depth[k];
      }
    }
    {
      int e = 0;
      for (auto it = f_manager.feature.begin(); it != f_manager.feature.end(); ++e) it = kept[e] ? std::next(it) : f_manager.feature.erase(it);
    }
  }

  Context& context() { return ctx_; }

 private:
  void fill_imu(WindowTables& t, size_t S) const {
    t.max_samp = (int)S;
    t.imu_n.assign(AVM_WINDOW_SIZE, 0);
    t.imu_dt.assign(AVM_WINDOW_SIZE * S, 0.0);
    t.imu_acc.assign(AVM_WINDOW_SIZE * (S + 1) * 3, 0.0), t.imu_gyr.assign(AVM_WINDOW_SIZE * (S + 1) * 3, 0.0);
    t.imu_lin_ba.assign(AVM_WINDOW_SIZE * 3, 0.0), t.imu_lin_bg.assign(AVM_WINDOW_SIZE * 3, 0.0);
    for (int j = 0; j < AVM_WINDOW_SIZE; j++) {
      const IntegrationBase& p = pre_integrations[j + 1];
      const size_t n = p.dt_buf.size(), row0 = (size_t)j * (S + 1);
      t.imu_n[j] = (int32_t)n;
      for (int k = 0; k < 3; k++) {
        t.imu_acc[row0 * 3 + k] = p.linearized_acc[k], t.imu_gyr[row0 * 3 + k] = p.linearized_gyr[k];  // row 0 = the constructor's sample
        t.imu_lin_ba[j * 3 + k] = p.linearized_ba[k], t.imu_lin_bg[j * 3 + k] = p.linearized_bg[k];
      }
      for (size_t s = 0; s < n; s++) {
        t.imu_dt[j * S + s] = p.dt_buf[s];
        for (int k = 0; k < 3; k++) t.imu_acc[(row0 + s + 1) * 3 + k] = p.acc_buf[s][k], t.imu_gyr[(row0 + s + 1) * 3 + k] = p.gyr_buf[s][k];
      }
    }
  }

  Context& ctx_;
};

// ---------------------------------------------------------------------------------------------------------
// FeatureSelector (feature_selector.h:36-188) and HP-B
// ---------------------------------------------------------------------------------------------------------
// x y z u v vx vy prob per camera (state_defs.h:24-33)
using image_t = std::map<int, std::vector<std::pair<int, std::array<double, 8>>>>;
enum { fPROB = 7 };

struct PinholeCamera {  // what generateCameraFromYamlFile() reads for a PINHOLE model (feature_selector.cpp:19)
  double fx = 0, fy = 0, cx = 0, cy = 0, k1 = 0, k2 = 0, p1 = 0, p2 = 0;
  int image_width = 0, image_height = 0;
};

struct HorizonState {  // state_t of state_defs.h:15-19 without the unused entries
  double timestamp = 0;
  Vector3d pos{}, vel{}, b_a{};
  Quaterniond q;
};

class FeatureSelector {
 public:
  static constexpr int HORIZON = 13;  // state_defs.h:8; a member here because the device takes it at run time

  FeatureSelector(Estimator& estimator, const PinholeCamera& camera, int horizon = HORIZON)
      : estimator_(estimator), camera_(camera), horizon_(horizon), q_IC_(estimator.ric[0]), t_IC_(estimator.tic[0]) {}

  void setParameters(double accVar, double accBiasVar, bool enable, int maxFeatures, int initThresh, bool useGT) {  // :24-34
    accVarDTime_ = accVar, accBiasVarDTime_ = accBiasVar;
    enable_ = enable, maxFeatures_ = maxFeatures, initThresh_ = initThresh, useGT_ = useGT;
  }

  // HorizonGenerator::loadGroundTruth for useGT (horizon_generator.cpp:169-196)
  void loadGroundTruth(const std::string& data_csv) {
    if (gt_) avm_gt_free(gt_);
    gt_ = avm_gt_load_csv(data_csv.c_str());
    if (!gt_) throw Error(AVM_ERR_INVALID, "cannot read ground truth " + data_csv);
  }
  void setGroundTruth(const double* rows17, int n) {
    if (gt_) avm_gt_free(gt_);
    gt_ = avm_gt_from_rows(rows17, n);
    if (!gt_) throw Error(AVM_ERR_INVALID, "bad ground-truth table");
  }
  ~FeatureSelector() {
    if (gt_) avm_gt_free(gt_);
  }
  int groundTruthSeek() const { return gt_ ? avm_gt_seek(gt_) : -1; }  // HorizonGenerator::seek_idx_
  FeatureSelector(const FeatureSelector&) = delete;
  FeatureSelector& operator=(const FeatureSelector&) = delete;

  void setNextStateFromImuPropagation(double imageTimestamp, const Vector3d& P, const Quaterniond& Q, const Vector3d& V, const Vector3d& a,
                                      const Vector3d& w, const Vector3d& Ba) {  // :38-70
    state_k_.timestamp = state_k1_.timestamp;
    state_k_.pos = estimator_.Ps[AVM_WINDOW_SIZE], state_k_.vel = estimator_.Vs[AVM_WINDOW_SIZE], state_k_.b_a = estimator_.Bas[AVM_WINDOW_SIZE];
    state_k_.q = estimator_.Rs[AVM_WINDOW_SIZE];
    state_k1_.timestamp = imageTimestamp, state_k1_.pos = P, state_k1_.vel = V, state_k1_.b_a = Ba, state_k1_.q = Q;
    ak1_ = a, wk1_ = w;
  }

  // HP-B.  `image` is replaced by the subset handed to the back end; returns {trackedFeatures_, selectedIds}.
  // header_stamp = header.stamp.toSec().
  std::pair<std::vector<int>, std::vector<int>> select(image_t& image, double header_stamp, int nrImuMeasurements) {
    if (!enable_) return {};

    // the reference latches the first stamp in a function-local static (:85); here it is per object
    if (!have_frame_time_) frameTime_k_ = header_stamp, have_frame_time_ = true;
    const double deltaF = header_stamp - frameTime_k_;
    const double deltaImu = deltaF / nrImuMeasurements;

    // new features are the ids above the largest one seen so far (:208-219)
    image_t image_new;
    {
      auto it = image.upper_bound(lastFeatureId_);
      image_new.insert(it, image.end());
      image.erase(it, image.end());
    }
    if (!image_new.empty()) lastFeatureId_ = image_new.rbegin()->first;

    image_t subset;
    for (int fid : trackedFeatures_) {
      auto f = image.find(fid);
      if (f != image.end()) subset[fid] = f->second;
    }

    const bool initialized = estimator_.solver_flag == Estimator::NON_LINEAR;
    std::vector<int> selectedIds;
    // The reference builds the horizon and the information matrices on every call (:131-143) and reads them only when it
    // selects; here the device is asked only then.  The ground-truth cursor moves on every call, as it does there.
    std::vector<double> hor_pos, hor_quat;
    const auto tm0 = std::chrono::steady_clock::now();
    if (useGT_ || initialized) generateFutureHorizon(nrImuMeasurements, deltaImu, deltaF, hor_pos, hor_quat);
    lastHorizonMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count();
    if (initialized) {
      selectedIds = selectInformativeFeatures(subset, image_new, nrImuMeasurements, deltaImu, hor_pos, hor_quat);
      for (int id : selectedIds) subset[id] = image_new.at(id);  // :677
    } else if (firstImage_) {
      subset.swap(image_new);
      for (const auto& f : subset) trackedFeatures_.push_back(f.first);
      firstImage_ = false;
    }
    if (!initialized && (int)subset.size() < initThresh_) subset.insert(image.begin(), image.end());

    image.swap(subset);
    trackedFeatures_.insert(trackedFeatures_.end(), selectedIds.begin(), selectedIds.end());
    frameTime_k_ = header_stamp;
    return std::make_pair(trackedFeatures_, selectedIds);
  }

  // B4: state_kkH[0..H] as hor_pos [H+1][3], hor_quat [H+1][4] (feature_selector.cpp:223-236)
  void generateFutureHorizon(int nrImuMeasurements, double deltaImu, double deltaFrame, std::vector<double>& hor_pos, std::vector<double>& hor_quat) {
    const int H = horizon_;
    hor_pos.assign((size_t)(H + 1) * 3, 0.0), hor_quat.assign((size_t)(H + 1) * 4, 0.0);
    const double kq[4] = {state_k_.q.x, state_k_.q.y, state_k_.q.z, state_k_.q.w};
    if (useGT_) {
      if (!gt_) throw Error(AVM_ERR_INVALID, "useGT without a ground-truth table");
      const int rc = avm_fsel_horizon_ground_truth(gt_, H, state_k_.timestamp, state_k_.pos.data(), kq, deltaFrame, hor_pos.data(), hor_quat.data());
      if (rc != AVM_OK) throw Error(rc, "avm_fsel_horizon_ground_truth: the horizon runs past the end of the table");
      return;
    }
    const double k1q[4] = {state_k1_.q.x, state_k1_.q.y, state_k1_.q.z, state_k1_.q.w};
    const int32_t nr = nrImuMeasurements;
    avm_fsel_horizon_in in{};
    in.n_problems = 1, in.horizon = H;
    in.k_pos = state_k_.pos.data(), in.k_quat = kq, in.k_ba = state_k_.b_a.data();
    in.k1_pos = state_k1_.pos.data(), in.k1_vel = state_k1_.vel.data(), in.k1_quat = k1q;
    in.acc = ak1_.data(), in.gyr = wk1_.data(), in.nr_imu = &nr, in.delta_imu = &deltaImu;
    Context& c = estimator_.context();
    c.check(avm_fsel_horizon_imu(c.get(), AVM_MEM_HOST, &in, hor_pos.data(), hor_quat.data()), "avm_fsel_horizon_imu");
  }

  std::vector<int> trackedFeatures_;  // feature_selector.h:93 (never pruned, :196)
  int lastFeatureId_ = 0;
  std::vector<double> lastF_;         // f value of each greedy round of the last select()
  double lastHorizonMs_ = 0, lastCloudMs_ = 0, lastSelectMs_ = 0;  // wall time of the three device calls of the last select() (diagnostics: bench.py)

 private:
  // :139-171 + :613-686 behind one ABI call: information of the horizon, of every candidate and tracked feature, and the
  // lazy greedy with the log-det upper bounds.  Returns the chosen ids in selection order.
  std::vector<int> selectInformativeFeatures(const image_t& subset, const image_t& image_new, int nrImuMeasurements, double deltaImu,
                                             const std::vector<double>& hor_pos, const std::vector<double>& hor_quat) {
    Context& c = estimator_.context();

    // depth cloud of initKDTree (:380-433): solve_flag != 1 is expressed as "no depth" for the device-side filter
    WindowTables t;
    estimator_.marshal(t, [](const FeaturePerId& f) { return f.solve_flag == 1 ? 1.0 / f.estimated_depth : -1.0; });
    avm_window_batch wb = t.batch(nullptr);
    const int max_cloud = WindowTables::MAX_FEAT;
    int32_t n_cloud = 0;
    std::vector<double> cloud_xy((size_t)max_cloud * 2, 0.0), cloud_depth(max_cloud, 0.0);
    const double k1q[4] = {state_k1_.q.x, state_k1_.q.y, state_k1_.q.z, state_k1_.q.w};
    const auto tm0 = std::chrono::steady_clock::now();
    c.check(avm_fsel_build_cloud(c.get(), AVM_MEM_HOST, &wb, state_k1_.pos.data(), k1q, max_cloud, &n_cloud, cloud_xy.data(), cloud_depth.data()),
            "avm_fsel_build_cloud");
    lastCloudMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm0).count();

    const int32_t n_cand = (int32_t)image_new.size(), n_used = (int32_t)subset.size();
    std::vector<int32_t> cand_id, used_id;
    std::vector<double> cand_xy, cand_prob, used_xy;
    for (const auto& f : image_new) {
      const auto& v = f.second[0].second;
      cand_id.push_back(f.first), cand_xy.push_back(v[0]), cand_xy.push_back(v[1]), cand_prob.push_back(v[fPROB]);
    }
    for (const auto& f : subset) {
      const auto& v = f.second[0].second;
      used_id.push_back(f.first), used_xy.push_back(v[0]), used_xy.push_back(v[1]);
    }
    if (cand_id.empty()) cand_id.push_back(0), cand_xy.resize(2, 0.0), cand_prob.push_back(0.0);
    if (used_id.empty()) used_id.push_back(0), used_xy.resize(2, 0.0);

    avm_fsel_batch p{};
    p.n_problems = 1, p.horizon = horizon_, p.max_features = maxFeatures_;
    p.max_cand = (int32_t)cand_id.size(), p.max_used = (int32_t)used_id.size(), p.max_cloud = max_cloud;
    const int32_t nr = nrImuMeasurements;
    p.hor_pos = hor_pos.data(), p.hor_quat = hor_quat.data(), p.nr_imu = &nr, p.delta_imu = &deltaImu;
    p.acc_var = accVarDTime_, p.acc_bias_var = accBiasVarDTime_;
    p.q_ic[0] = q_IC_.x, p.q_ic[1] = q_IC_.y, p.q_ic[2] = q_IC_.z, p.q_ic[3] = q_IC_.w;
    std::copy(t_IC_.begin(), t_IC_.end(), p.t_ic);
    p.fx = camera_.fx, p.fy = camera_.fy, p.cx = camera_.cx, p.cy = camera_.cy;
    p.k1 = camera_.k1, p.k2 = camera_.k2, p.p1 = camera_.p1, p.p2 = camera_.p2;
    p.image_width = camera_.image_width, p.image_height = camera_.image_height;
    p.n_cand = &n_cand, p.cand_id = cand_id.data(), p.cand_xy = cand_xy.data(), p.cand_prob = cand_prob.data();
    p.n_used = &n_used, p.used_id = used_id.data(), p.used_xy = used_xy.data();
    p.n_cloud = &n_cloud, p.cloud_xy = cloud_xy.data(), p.cloud_depth = cloud_depth.data();

    int32_t n_sel = 0;
    std::vector<int32_t> ids(std::max(1, maxFeatures_), 0);
    lastF_.assign(std::max(1, maxFeatures_), 0.0);
    avm_fsel_out out{&n_sel, ids.data(), lastF_.data(), nullptr};
    const auto tm1 = std::chrono::steady_clock::now();
    c.check(avm_fsel_select_batch(c.get(), AVM_MEM_HOST, &p, &out), "avm_fsel_select_batch");
    lastSelectMs_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tm1).count();
    lastF_.resize(n_sel);
    return std::vector<int>(ids.begin(), ids.begin() + n_sel);
  }

  Estimator& estimator_;
  PinholeCamera camera_;
  int horizon_;
  Quaterniond q_IC_;
  Vector3d t_IC_;
  double accVarDTime_ = 0, accBiasVarDTime_ = 0;
  bool enable_ = true, useGT_ = false, firstImage_ = true;
  int maxFeatures_ = 0, initThresh_ = 0;
  HorizonState state_k_, state_k1_;
  Vector3d ak1_{}, wk1_{};
  double frameTime_k_ = 0;
  bool have_frame_time_ = false;
  avm_gt* gt_ = nullptr;
};

}  // namespace avm_host
#endif  // AVM_HOST_HPP_
