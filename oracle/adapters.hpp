// oracle/adapters.hpp - TEST INFRASTRUCTURE ONLY (see avm_oracle.cpp): CPU restatement of the reference's host-side
// format adapters.  "parity unpinned" like the rest of the oracle (the reference cannot be built here); pinned by
// independent numpy statements in tests/test_oracle.py.
//   HorizonGenerator::loadGroundTruth / groundTruth / getNextFrameTruth   utility/horizon_generator.cpp:169-196, 73-123, 200-210
//   PointCloud -> image_t decode                                           estimator_node.cpp:303-321
#pragma once
#include <map>
#include <utility>
#include <vector>
#include "linalg.hpp"

namespace avmo {

struct GtRow {
  double timestamp;
  V3 p;
  Q q;
};

struct GroundTruth {
  std::vector<GtRow> truth;
  int seek_idx = 0;
  // rows [n][17]: timestamp (ns), p, q (w x y z), v, w, a   (loadGroundTruth: stod(field) * 1e-9)
  void load(const double* rows, int n) {
    truth.clear();
    for (int i = 0; i < n; i++) {
      const double* r = rows + 17 * (size_t)i;
      truth.push_back({r[0] * 1e-9, V3(r[1], r[2], r[3]), Q(r[4], r[5], r[6], r[7])});
    }
    seek_idx = 0;
  }
  // getNextFrameTruth (:200-210); false where the reference would index past the table
  bool next_frame(int& idx, double deltaFrame, GtRow& out) const {
    const double nextTimestep = truth[idx].timestamp + deltaFrame;
    while (idx < (int)truth.size() && truth[idx++].timestamp <= nextTimestep) {}
    if (idx >= (int)truth.size()) return false;
    out = truth[idx];
    return true;
  }
  // groundTruth (:73-123): pos [H+1], quat [H+1]
  bool horizon(int H, double t0, V3 p0, Q q0, double deltaFrame, std::vector<V3>& pos, std::vector<Q>& quat) {
    double timestamp = t0;
    if (timestamp > truth.back().timestamp) timestamp = truth.front().timestamp;
    while (seek_idx < (int)truth.size() && truth[seek_idx++].timestamp <= timestamp) {}
    int idx = seek_idx - 1;
    if (idx >= (int)truth.size()) return false;
    pos.assign(H + 1, V3()), quat.assign(H + 1, Q());
    pos[0] = p0, quat[0] = q0;
    V3 prevP = truth[idx].p;
    Q prevQ = truth[idx].q;
    for (int h = 1; h <= H; h++) {
      GtRow gt;
      if (!next_frame(idx, deltaFrame, gt)) return false;
      const Q relQ = inverse(prevQ) * gt.q;
      const V3 relP = rot(inverse(gt.q), gt.p - prevP);
      pos[h] = pos[h - 1] + rot(quat[h - 1], relP);
      quat[h] = quat[h - 1] * relQ;
      prevP = gt.p, prevQ = gt.q;
    }
    return true;
  }
};

// estimator_node.cpp:303-321 with image_t = std::map<int, vector<pair<int, Matrix<double,8,1>>>>, flattened in map order
inline bool image_from_pointcloud(int n, const float* pts, const float* const* ch, int num_cam, int* feature_id, int* camera_id, double* out) {
  std::map<int, std::vector<std::pair<int, std::vector<double>>>> image;
  for (int i = 0; i < n; i++) {
    const int v = ch[0][i] + 0.5;
    const int fid = v / num_cam, cam = v % num_cam;
    const double x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
    if (!(z == 1)) return false;  // ROS_ASSERT(z == 1)
    image[fid].emplace_back(cam, std::vector<double>{x, y, z, ch[1][i], ch[2][i], ch[3][i], ch[4][i], ch[5][i]});
  }
  int o = 0;
  for (const auto& kv : image)
    for (const auto& e : kv.second) {
      feature_id[o] = kv.first, camera_id[o] = e.first;
      for (int k = 0; k < 8; k++) out[8 * (size_t)o + k] = e.second[k];
      o++;
    }
  return true;
}

}  // namespace avmo
