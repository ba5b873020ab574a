// adapters.hip - host-side format adapters of SURVEY 8(f)4 and the ground-truth mode of B4 (include/avm.h).
// Pure host code (the reference does this bookkeeping on the host too); compiled into libavm_hip.so with the kernels.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include "../../include/avm.h"

struct avm_gt {
  struct Row { double t, p[3], q[4] /* w x y z */; };
  std::vector<Row> rows;
  int seek = 0;  // HorizonGenerator::seek_idx_
};

namespace {
struct Quat { double w, x, y, z; };
inline Quat qmul(const Quat& a, const Quat& b) {
  return {a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z, a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y,
          a.w * b.y + a.y * b.w + a.z * b.x - a.x * b.z, a.w * b.z + a.z * b.w + a.x * b.y - a.y * b.x};
}
inline Quat qinv(const Quat& q) {  // Eigen: conjugate / squaredNorm
  const double n2 = q.w * q.w + q.x * q.x + q.y * q.y + q.z * q.z;
  return {q.w / n2, -q.x / n2, -q.y / n2, -q.z / n2};
}
inline void qrot(const Quat& q, const double v[3], double o[3]) {  // Eigen: v + w t + u x t, t = 2 u x v
  double t[3] = {2 * (q.y * v[2] - q.z * v[1]), 2 * (q.z * v[0] - q.x * v[2]), 2 * (q.x * v[1] - q.y * v[0])};
  o[0] = v[0] + q.w * t[0] + (q.y * t[2] - q.z * t[1]);
  o[1] = v[1] + q.w * t[1] + (q.z * t[0] - q.x * t[2]);
  o[2] = v[2] + q.w * t[2] + (q.x * t[1] - q.y * t[0]);
}
}  // namespace

extern "C" {

avm_gt* avm_gt_from_rows(const double* rows, int32_t n) {
  if (!rows || n <= 0) return nullptr;
  avm_gt* g = new avm_gt;
  g->rows.resize(n);
  for (int i = 0; i < n; i++) {
    const double* r = rows + (size_t)i * 17;
    avm_gt::Row& o = g->rows[i];
    o.t = r[0] * 1e-9;
    for (int k = 0; k < 3; k++) o.p[k] = r[1 + k];
    for (int k = 0; k < 4; k++) o.q[k] = r[4 + k];
  }
  return g;
}

avm_gt* avm_gt_load_csv(const char* data_csv) {
  if (!data_csv) return nullptr;
  FILE* f = fopen(data_csv, "r");
  if (!f) return nullptr;
  std::vector<double> rows;
  char line[2048];
  bool first = true;
  while (fgets(line, sizeof line, f)) {
    if (first) { first = false; continue; }  // header
    double v[17];
    int k = 0;
    char* s = line;
    while (k < 17) {
      char* e;
      v[k] = strtod(s, &e);
      if (e == s) break;
      k++;
      s = e;
      while (*s == ',' || *s == ' ') s++;
    }
    if (k == 0) continue;             // blank line
    if (k < 17) { fclose(f); return nullptr; }
    rows.insert(rows.end(), v, v + 17);
  }
  fclose(f);
  return avm_gt_from_rows(rows.data(), (int32_t)(rows.size() / 17));
}

void avm_gt_free(avm_gt* gt) { delete gt; }
int32_t avm_gt_size(const avm_gt* gt) { return gt ? (int32_t)gt->rows.size() : 0; }
int32_t avm_gt_seek(const avm_gt* gt) { return gt ? gt->seek : 0; }

int avm_fsel_horizon_ground_truth(avm_gt* gt, int32_t horizon, double timestamp_k, const double* k_pos, const double* k_quat,
                                  double delta_frame, double* hor_pos, double* hor_quat) {
  if (!gt || gt->rows.empty() || horizon < 1 || !k_pos || !k_quat || !hor_pos || !hor_quat) return AVM_ERR_INVALID;
  const std::vector<avm_gt::Row>& T = gt->rows;
  const int n = (int)T.size();
  double timestamp = timestamp_k;
  if (timestamp > T.back().t) timestamp = T.front().t;  // "likely the first state_0 (which may have random values)"
  while (gt->seek < n && T[gt->seek++].t <= timestamp) {}
  int idx = gt->seek - 1;
  if (idx >= n) return AVM_ERR_INVALID;
  for (int k = 0; k < 3; k++) hor_pos[k] = k_pos[k];
  for (int k = 0; k < 4; k++) hor_quat[k] = k_quat[k];
  double prevP[3] = {T[idx].p[0], T[idx].p[1], T[idx].p[2]};
  Quat prevQ{T[idx].q[0], T[idx].q[1], T[idx].q[2], T[idx].q[3]};
  for (int h = 1; h <= horizon; h++) {
    // getNextFrameTruth: advances idx one PAST the first row later than the next time step
    const double next = T[idx].t + delta_frame;
    while (idx < n && T[idx++].t <= next) {}
    if (idx >= n) return AVM_ERR_INVALID;
    const avm_gt::Row& g = T[idx];
    const Quat gq{g.q[0], g.q[1], g.q[2], g.q[3]};
    const Quat relQ = qmul(qinv(prevQ), gq);
    const double dp[3] = {g.p[0] - prevP[0], g.p[1] - prevP[1], g.p[2] - prevP[2]};
    double relP[3], step[3];
    qrot(qinv(gq), dp, relP);
    const double* qp = hor_quat + 4 * (h - 1);
    const Quat qprev{qp[3], qp[0], qp[1], qp[2]};
    qrot(qprev, relP, step);
    for (int k = 0; k < 3; k++) hor_pos[3 * h + k] = hor_pos[3 * (h - 1) + k] + step[k];
    const Quat qh = qmul(qprev, relQ);
    hor_quat[4 * h] = qh.x, hor_quat[4 * h + 1] = qh.y, hor_quat[4 * h + 2] = qh.z, hor_quat[4 * h + 3] = qh.w;
    for (int k = 0; k < 3; k++) prevP[k] = g.p[k];
    prevQ = gq;
  }
  return AVM_OK;
}

int avm_image_from_pointcloud(int32_t n_points, const float* points_xyz, const float* const* channels, int32_t num_cam,
                              int32_t* feature_id, int32_t* camera_id, double* xyz_uv_velocity) {
  if (n_points < 0 || num_cam < 1 || (n_points > 0 && (!points_xyz || !channels || !feature_id || !camera_id || !xyz_uv_velocity)))
    return AVM_ERR_INVALID;
  for (int c = 0; c < 6 && n_points > 0; c++)
    if (!channels[c]) return AVM_ERR_INVALID;
  std::vector<int> fid(n_points), order(n_points);
  for (int i = 0; i < n_points; i++) {
    if (points_xyz[3 * i + 2] != 1.0f) return AVM_ERR_INVALID;
    fid[i] = (int)(channels[0][i] + 0.5);  // float + double 0.5, truncated
  }
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return fid[a] / num_cam < fid[b] / num_cam; });
  for (int o = 0; o < n_points; o++) {
    const int i = order[o];
    feature_id[o] = fid[i] / num_cam, camera_id[o] = fid[i] % num_cam;
    double* d = xyz_uv_velocity + 8 * (size_t)o;
    d[0] = points_xyz[3 * i], d[1] = points_xyz[3 * i + 1], d[2] = points_xyz[3 * i + 2];
    for (int c = 1; c < 6; c++) d[2 + c] = channels[c][i];
  }
  return AVM_OK;
}

}  // extern "C"
