// fsel.hip — FeatureSelector::select() (initialized branch) for a batch of frames on gfx950.
//
// reference: vins_estimator/src/feature_selector.cpp
//   calcInfoFromRobotMotion :463-527, createLinearImuMatrices :531-598, addOmegaPrior :602-609,
//   calcInfoFromFeatures :239-365 (+ inFOV :369-376, findNNDepth :437-459, PinholeCamera
//   spaceToPlane/distortion camera_model/src/camera_models/PinholeCamera.cc:520-542,646-662),
//   selectInformativeFeatures :613-686, sortedlogDetUB :690-728, Utility::logdet utility.h:144-167.
//
// MI355X mapping.  Delta_ell only touches the 3H position rows of horizon states 1..H
// (feature_selector.cpp:349-355), while Omega's other 6H+9 rows never change during a
// select() call.  So the Cholesky pivots of those rows are hoisted: setup eliminates them once
// per frame (partial Cholesky in LDS) leaving C0 = Omega_pp - Omega_pn Omega_nn^-1 Omega_np and
// logdet(Omega_nn); every candidate evaluation logdet(Omega + OmegaS + p*Delta) is then
// logdet(Omega_nn) + logdet(C + p*Delta_pp) with a 3H x 3H factorisation held entirely in
// registers (four candidates per wavefront, one per 16-lane DPP row; one launch per greedy round).
// This is the same Cholesky with the constant leading pivots factored once — the same kind of
// hoist as IMUFactor's sqrt_info.  All FP64; selection order is deterministic.
#include <algorithm>
#include <type_traits>
#include <cfloat>
#include <cstdlib>

#include "devmath.hpp"
#include "kernels.hpp"

namespace avm {

namespace {

constexpr int FS_NT = 256;
constexpr int FS_CPW = 4;  // candidates per wavefront in the Delta slices of the setup kernel
constexpr int FS_TABLES_OK = 0x7f7f7f7f;  // (what launch_validate_fsel leaves in the flag when every frame passes)
#define FS_TABLES_GUARD(A) if ((A).vflag && *(A).vflag != FS_TABLES_OK) return

struct FselDev {
  avm_fsel_batch b;  // device pointers
  int no_key_rule;   // test switch (AVM_FSEL_NO_KEY_RULE=1): skip the std::map equal-key rule of sortedlogDetUB
  int lazy_stats;    // development (AVM_FSEL_LAZY_STATS=1): workgroup 0 of fsel_solo_kernel leaves its counters and phase clocks in sync[32..]
  double lazy_tau;   // fsel_solo_kernel: a candidate is scored in a round's first pass when its gain bound reaches lazy_tau x the last winner's gain
  const int* vflag;  // result of the table validation that runs ahead on the same stream (null: already checked by the host): any
                     // value but FS_TABLES_OK means a malformed table - no kernel of the select may index with the tables then
  // work buffers
  double* C;        // [P][T*T] current reduced position information (C0 + used + OmegaS)
  double* dpp;      // [P][T]   un-reduced diagonal of the position rows (for the Hadamard bound)
  double* consts;   // [P][4]   ld_nn, Kn
  double* delta;    // [P][max_cand][T*T]
  double* delta_pk; // [P][max_cand][T(T+1)/2] the same, lower triangle by columns (entry (R, c), c <= R, at c T - c (c - 1) / 2 + R - c): what
                    // fsel_solo_kernel scores from - half the bytes per evaluation; null unless the solo form runs
  double* delta_u;  // [P][max_used][T*T]
  double* ddiag;    // [P][max_cand][T] every candidate's Delta diagonal (fsel_solo_kernel at 3 H = 39, where its LDS copy is single precision)
  int32_t* valid;   // [P][max_cand] 1 = triangulable (numVisible > 1)
  int32_t* valid_u; // [P][max_used]
  int32_t* black;   // [P][max_cand]
  double* fval;     // [P][max_cand]
  double* ub;       // [P][max_cand]
  int32_t* nsel;    // [P] number selected so far
  int32_t* done;    // [P] 1 when a round found no winner (state is then frozen)
  int32_t* live;    // [P][max_cand] indices of the candidates still in the race (valid, not yet selected), any order
  int32_t* pos;     // [P][max_cand] position of candidate l in live[]
  int32_t* nlive;   // [P]
  double* omega_out;  // optional [P][N*N] (tests)
  avm_fsel_out out;
  double* kd;         // [P][kd_stride(max_cloud)] the frame's kd-tree over its depth cloud (fsel_kdtree_kernel; searched by kd_depth)
};

AVM_DEV quat slerp_eigen(quat a, double t, quat b) {
  const double one = 1.0 - DBL_EPSILON;
  const double d = a.w * b.w + a.x * b.x + a.y * b.y + a.z * b.z;
  const double ad = fabs(d);
  double s0, s1;
  if (ad >= one) {
    s0 = 1.0 - t;
    s1 = t;
  } else {
    const double th = acos(ad), st = sin(th);
    s0 = sin((1.0 - t) * th) / st;
    s1 = sin(t * th) / st;
  }
  if (d < 0) s1 = -s1;
  return quat{s0 * a.w + s1 * b.w, s0 * a.x + s1 * b.x, s0 * a.y + s1 * b.y, s0 * a.z + s1 * b.z};
}

// Delta_ell position blocks of one feature (calcInfoFromFeatures), written as dense T x T.
// cam[h] (h = 1..H): t_WC (3), R of q_WC^-1 (9), R of (q_WC * q_IC)^-1 (9)  => 21 doubles per h
// front part: per-frame blocks C_h (Ch, 6 per h), W = (sum C_h)^-1 (Wm); false if the feature is seen in no future frame
// WAVE: called by all 64 lanes of a wavefront for the same feature - the leaves of the nearest-neighbour search are split over the lanes
//
// ---- findNNDepth (feature_selector.cpp:437-459): the reference's kd-tree, bit for bit -------------------------------------------
// The reference asks nanoflann (vendored, vins_estimator/lib/nanoflann/nanoflann.hpp; KDTreeSingleIndexAdaptor<L2_Simple_Adaptor<double>, ., 2>,
// leaf_max_size 10, feature_selector.cpp:424-429) for the 1-NN of the candidate among the window's landmarks and uses that landmark's depth.
// The search is exact, so WHICH point it returns only depends on the tree when several points are at bit-identical distances - and
// then it is the one the traversal meets first (KNNResultSet::addPoint :175-202 replaces on a strictly smaller distance only).  That
// order is part of the reference's behaviour (selected ids are compared bit-exact), so the tree is built here as nanoflann builds it:
//   divideTree :857-907 - a range of <= 10 indices is a leaf; else middleSplit_ :909-958 picks the dimension of largest spread among
//   those whose bounding-box span is within 1e-5 of the largest, cuts at the box's middle clamped to the points' range, planeSplit
//   :969-1005 partitions the index range Hoare-style (< cut | == cut | > cut) and the split position is lim1 / lim2 / count / 2;
//   divlow / divhigh of a node are the children's tightened boxes = max of the left points / min of the right points in the cut dimension;
//   searchLevel :1346-1405 - nearer child first ((val - divlow) + (val - divhigh) < 0), the other one if mindistsq <= worst, a leaf's
//   points in index-array order against the worst distance read at the leaf's entry; computeInitialDistances :1007-1026 from the root box.
// fsel_kdtree_kernel builds it with one wavefront per frame (the partitions as ballot / prefix-count permutations: Hoare's swaps pair
// the i-th misplaced index from the left with the i-th from the right, which is what the sequential loop does), kd_depth walks it
// without a stack: the state of searchLevel's recursion along the current root-to-leaf path is a bit per level (near or far child),
// and (mindistsq, dists[]) are functions of that path, recomputed on the way down - the same additions in the same order.
// Arithmetic that decides comparisons is kept un-contracted (no FMA: the reference's x86 build has none).
// Known-answer tests: tests/test_nanoflann_nn.py (16 928 queries answered by the reference's own header, 4 237 of them exact ties).
struct KdNode {   // 32 bytes
  int a, b;       // inner node: children; leaf: the range [a, b) of the permuted point arrays
  int feat, pad;  // cut dimension (0 / 1); -1: leaf
  double lo, hi;  // divlow, divhigh
};
constexpr int KD_HDR = 8;  // doubles: root box low0 high0 low1 high1 | n_nodes, max_depth (two ints) | -
__host__ __device__ constexpr size_t kd_stride(int max_cloud) { return KD_HDR + (size_t)11 * max_cloud; }  // header | 2 mc nodes (4 doubles each) | xy[mc][2] | depth[mc], in tree order
constexpr int KD_MAXW = 64;  // path words of 64 levels each beyond the first (a tree of n points is at most n - 10 deep)

// minimum / maximum over the wavefront, the result in every lane: four DPP exchange steps inside the 16-lane rows (two 32-bit moves each),
// then the four row results through SGPRs - a __shfl_xor ladder is twelve dependent ds_bpermute per double (1.5 K cycles; the tree of a
// 150-point cloud takes ~90 of these reductions)
template <int CTRL>
AVM_DEV double kd_dpp(double v) {
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
AVM_DEV double kd_lane(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
AVM_DEV double kd_wave_min(double v) {
  v = fmin(v, kd_dpp<0xB1>(v)), v = fmin(v, kd_dpp<0x4E>(v)), v = fmin(v, kd_dpp<0x141>(v)), v = fmin(v, kd_dpp<0x140>(v));
  return fmin(fmin(kd_lane(v, 0), kd_lane(v, 16)), fmin(kd_lane(v, 32), kd_lane(v, 48)));
}
AVM_DEV double kd_wave_max(double v) {
  v = fmax(v, kd_dpp<0xB1>(v)), v = fmax(v, kd_dpp<0x4E>(v)), v = fmax(v, kd_dpp<0x141>(v)), v = fmax(v, kd_dpp<0x140>(v));
  return fmax(fmax(kd_lane(v, 0), kd_lane(v, 16)), fmax(kd_lane(v, 32), kd_lane(v, 48)));
}

// initKDTree (feature_selector.cpp:380-432, the buildIndex part): one wavefront per frame.  LDS: x[mc] y[mc] (doubles), vind[mc], two
// work lists [mc] (ints), a stack of pending ranges.  A node's work depends on its own index range and the box handed down only, so
// the larger child is parked and the smaller one taken first: the stack stays below log2(n) entries whatever the tree's shape.
struct KdPending {
  int l, r, slot, depth;
  double bb[4];
};
constexpr int KD_STACK = 40;
__global__ __launch_bounds__(64) void fsel_kdtree_kernel(FselDev A) {
#pragma clang fp contract(off)
  FS_TABLES_GUARD(A);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const avm_fsel_batch& b = A.b;
  const int p = blockIdx.x, lane = threadIdx.x, mc = b.max_cloud;
  const int n = b.n_cloud ? b.n_cloud[p] : 0;
  if (n <= 0) return;  // (findNNDepth answers 1.0 without a tree)
  double* X = reinterpret_cast<double*>(smem_raw);
  double* Y = X + mc;
  KdPending* stk = reinterpret_cast<KdPending*>(Y + mc);
  int* vi = reinterpret_cast<int*>(stk + KD_STACK);
  int* lml = vi + mc;   // positions of the misplaced indices of the left part, ascending
  int* lmr = lml + mc;  // ... of the right part, ascending
  double* kd = A.kd + (size_t)p * kd_stride(mc);
  KdNode* nodes = reinterpret_cast<KdNode*>(kd + KD_HDR);
  const double* cxy = b.cloud_xy + (size_t)p * mc * 2;
  double lo0 = DBL_MAX, hi0 = -DBL_MAX, lo1 = DBL_MAX, hi1 = -DBL_MAX;
  for (int i = lane; i < n; i += 64) {
    const double x = cxy[2 * i], y = cxy[2 * i + 1];
    X[i] = x, Y[i] = y, vi[i] = i;
    lo0 = fmin(lo0, x), hi0 = fmax(hi0, x), lo1 = fmin(lo1, y), hi1 = fmax(hi1, y);
  }
  lo0 = kd_wave_min(lo0), hi0 = kd_wave_max(hi0), lo1 = kd_wave_min(lo1), hi1 = kd_wave_max(hi1);  // computeBoundingBox
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  auto sync = [&]() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto coord = [&](int pos, int dim) { return dim == 0 ? X[vi[pos]] : Y[vi[pos]]; };
  // min / max of one coordinate over the index range [l, r)
  auto minmax = [&](int l, int r, int dim, double& mn, double& mx) {
    double a = DBL_MAX, c = -DBL_MAX;
    for (int q = l + lane; q < r; q += 64) {
      const double v = coord(q, dim);
      a = fmin(a, v), c = fmax(c, v);
    }
    mn = kd_wave_min(a), mx = kd_wave_max(c);
  };
  // one pass of planeSplit over [l, l + cnt): indices whose coordinate is < cut (strict) or <= cut to the front; returns how many
  auto partition = [&](int l, int cnt, int dim, double cut, bool strict) {
    int nl = 0;
    for (int base = 0; base < cnt; base += 64) {
      const int q = base + lane;
      const double v = q < cnt ? coord(l + q, dim) : 0.0;
      nl += __popcll(__ballot(q < cnt && (strict ? v < cut : v <= cut)));
    }
    int nml = 0, nmr = 0;
    for (int base = 0; base < cnt; base += 64) {
      const int q = base + lane;
      const double v = q < cnt ? coord(l + q, dim) : 0.0;
      const bool f = q < cnt && (strict ? v < cut : v <= cut);
      const bool ml = q < cnt && !f && q < nl, mr = f && q >= nl;
      const unsigned long long bl = __ballot(ml), br = __ballot(mr), lt = (1ull << lane) - 1ull;
      if (ml) lml[nml + __popcll(bl & lt)] = q;
      if (mr) lmr[nmr + __popcll(br & lt)] = q;
      nml += __popcll(bl), nmr += __popcll(br);
    }
    sync();
    // Hoare's swaps: the i-th misplaced index from the left with the i-th from the right (nml == nmr)
    for (int i = lane; i < nml; i += 64) {
      const int qa = l + lml[i], qb = l + lmr[nml - 1 - i];
      const int t = vi[qa];
      vi[qa] = vi[qb], vi[qb] = t;
    }
    sync();
    return nl;
  };
  int nn = 1, sp = 0, maxdepth = 0;
  int l = 0, r = n, slot = 0, depth = 0;
  double bb[4] = {lo0, hi0, lo1, hi1};
  for (;;) {
    const int cnt = r - l;
    maxdepth = max(maxdepth, depth);
    if (cnt <= 10) {  // a leaf
      if (lane == 0) nodes[slot] = KdNode{l, r, -1, 0, 0.0, 0.0};
      if (sp == 0) break;
      sp--;
      l = stk[sp].l, r = stk[sp].r, slot = stk[sp].slot, depth = stk[sp].depth;
#pragma unroll
      for (int k = 0; k < 4; k++) bb[k] = stk[sp].bb[k];
      continue;
    }
    // middleSplit_
    const double EPS = 0.00001;
    const double span0 = bb[1] - bb[0], span1 = bb[3] - bb[2];
    double max_span = span0;
    if (span1 > max_span) max_span = span1;
    double max_spread = -1.0, mn = 0, mx = 0;
    int cutfeat = 0;
    if (span0 > (1 - EPS) * max_span) {
      double a, c;
      minmax(l, r, 0, a, c);
      const double spread = c - a;
      if (spread > max_spread) cutfeat = 0, max_spread = spread;
      mn = a, mx = c;
    }
    if (span1 > (1 - EPS) * max_span) {
      double a, c;
      minmax(l, r, 1, a, c);
      const double spread = c - a;
      if (spread > max_spread) cutfeat = 1, max_spread = spread, mn = a, mx = c;
    }
    if (max_spread < 0) minmax(l, r, 0, mn, mx);  // (no dimension passed the test: NaN boxes; cutfeat stays 0 like the reference's)
    const double split_val = (bb[2 * cutfeat] + bb[2 * cutfeat + 1]) / 2;
    const double cutval = split_val < mn ? mn : (split_val > mx ? mx : split_val);
    const int lim1 = partition(l, cnt, cutfeat, cutval, true);
    const int lim2 = lim1 + partition(l + lim1, cnt - lim1, cutfeat, cutval, false);
    int idx = lim1 > cnt / 2 ? lim1 : (lim2 < cnt / 2 ? lim2 : cnt / 2);
    idx = min(max(idx, 1), cnt - 1);  // (both children non-empty: holds for every finite cloud, keeps the loop finite for any other)
    double dl, dh, tmp;
    minmax(l, l + idx, cutfeat, tmp, dl);  // divlow: the left child's tightened box, high side
    minmax(l + idx, r, cutfeat, dh, tmp);  // divhigh: the right child's, low side
    const int c1 = nn, c2 = nn + 1;
    nn += 2;
    if (lane == 0) nodes[slot] = KdNode{c1, c2, cutfeat, 0, dl, dh};
    // children: (l, l + idx) with the box cut at high = cutval, (l + idx, r) with low = cutval; the smaller one now, the other parked
    const bool left_now = idx <= cnt - idx;
    if (lane == 0 && sp < KD_STACK) {
      KdPending& e = stk[sp];
      e.l = left_now ? l + idx : l, e.r = left_now ? r : l + idx, e.slot = left_now ? c2 : c1, e.depth = depth + 1;
#pragma unroll
      for (int k = 0; k < 4; k++) e.bb[k] = bb[k];
      e.bb[left_now ? 2 * cutfeat : 2 * cutfeat + 1] = cutval;
    }
    sp++;
    sync();
    if (left_now) r = l + idx, bb[2 * cutfeat + 1] = cutval, slot = c1;
    else l = l + idx, bb[2 * cutfeat] = cutval, slot = c2;
    depth++;
  }
  // header + the points and their depths in tree order (a leaf reads consecutive entries)
  if (lane == 0) {
    kd[0] = lo0, kd[1] = hi0, kd[2] = lo1, kd[3] = hi1;
    int* hi = reinterpret_cast<int*>(kd + 4);
    hi[0] = nn, hi[1] = maxdepth;
  }
  double* pxy = kd + KD_HDR + 8 * (size_t)mc;
  double* pdep = pxy + 2 * (size_t)mc;
  const double* cdep = b.cloud_depth + (size_t)p * mc;
  for (int i = lane; i < n; i += 64) {
    const int o = vi[i];
    pxy[2 * i] = X[o], pxy[2 * i + 1] = Y[o], pdep[i] = cdep[o];
  }
}
size_t fsel_kdtree_lds_bytes(int max_cloud) { return (size_t)max_cloud * (2 * sizeof(double) + 3 * sizeof(int)) + KD_STACK * sizeof(KdPending) + 16; }
// the kd-tree of every frame's depth cloud -> kd[P][kd_stride(max_cloud)] (what the setup kernel and fsel_nn_depth_kernel search)
hipError_t launch_fsel_kdtree(const FselDev& d, hipStream_t stream) {
  const avm_fsel_batch& b = d.b;
  if (b.n_problems == 0 || !b.n_cloud || b.max_cloud <= 0) return hipSuccess;
  if (b.max_cloud > FS_MAX_CLOUD) return hipErrorInvalidValue;
  const size_t lds = fsel_kdtree_lds_bytes(b.max_cloud);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fsel_kdtree_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fsel_kdtree_kernel, dim3(b.n_problems), dim3(64), lds, stream, d);
  return hipGetLastError();
}


// findNeighbors + searchLevel for ONE query by W cooperating lanes (1, 16 or 64 consecutive lanes that enter together with the same
// query): everything is uniform over the group except the leaf scan, where lane i of the group takes the leaf's i-th point.
// Returns findNNDepth's value: the depth of the point the reference's search returns (ret_index stays 0 if nothing is ever closer
// than the initial worst distance, e.g. for a NaN query: then the depth of cloud point 0), 1.0 for an empty cloud.
template <int W>
AVM_DEV double kd_depth(const avm_fsel_batch& b, const double* kdall, int p, double qx, double qy) {
#pragma clang fp contract(off)
  const int n = b.n_cloud ? b.n_cloud[p] : 0;
  if (n <= 0) return 1.0;
  const int mc = b.max_cloud, gl = threadIdx.x & (W - 1);
  const double* kd = kdall + (size_t)p * kd_stride(mc);
  const KdNode* nodes = reinterpret_cast<const KdNode*>(kd + KD_HDR);
  const double* pxy = kd + KD_HDR + 8 * (size_t)mc;
  const double* pdep = pxy + 2 * (size_t)mc;
  // computeInitialDistances
  double di0 = 0.0, di1 = 0.0, dsq = 0.0;
  if (qx < kd[0]) di0 = (qx - kd[0]) * (qx - kd[0]), dsq += di0;
  if (qx > kd[1]) di0 = (qx - kd[1]) * (qx - kd[1]), dsq += di0;
  if (qy < kd[2]) di1 = (qy - kd[2]) * (qy - kd[2]), dsq += di1;
  if (qy > kd[3]) di1 = (qy - kd[3]) * (qy - kd[3]), dsq += di1;
  double worst = DBL_MAX, ans = b.cloud_depth[(size_t)p * mc];
  // the current root-to-leaf path: bit l set = the FAR child was taken at level l (its near child is done); levels >= plen: near
  unsigned long long path0 = 0;
  unsigned long long pathx[KD_MAXW];  // levels 64 .. (touched only by trees deeper than 64: private memory); words [0, nx) are in use
  int plen = 0, nx = 0;
  auto bit = [&](int l) -> bool { return l < 64 ? (path0 >> l) & 1ull : (pathx[min(l >> 6, KD_MAXW) - 1] >> (l & 63)) & 1ull; };
  for (;;) {
    // down: along the recorded path, then near children to a leaf
    int node = 0, lvl = 0;
    double mind = dsq, d0 = di0, d1 = di1;
    KdNode nd = nodes[0];
    while (nd.feat >= 0) {
      const double val = nd.feat ? qy : qx;
      const bool near1 = (val - nd.lo) + (val - nd.hi) < 0.0;
      if (lvl < plen && bit(lvl)) {
        const double cut = near1 ? (val - nd.hi) * (val - nd.hi) : (val - nd.lo) * (val - nd.lo);
        const double dst = nd.feat ? d1 : d0;
        mind = mind + cut - dst;
        if (nd.feat) d1 = cut; else d0 = cut;
        node = near1 ? nd.b : nd.a;
      } else {
        node = near1 ? nd.a : nd.b;
      }
      lvl++;
      nd = nodes[node];
    }
    // the leaf: points in index-array order against the worst distance at entry; a strictly smaller distance replaces
    {
      const int cnt = nd.b - nd.a;
      if (W == 1) {
        const double wl = worst;
        for (int i = 0; i < cnt; i++) {
          const double dx = qx - pxy[2 * (nd.a + i)], dy = qy - pxy[2 * (nd.a + i) + 1];
          const double d = dx * dx + dy * dy;
          if (d < wl && d < worst) worst = d, ans = pdep[nd.a + i];
        }
      } else {
        const bool in = gl < cnt;
        const int q = nd.a + min(gl, max(cnt - 1, 0));
        const double dx = qx - pxy[2 * q], dy = qy - pxy[2 * q + 1];
        const double d = dx * dx + dy * dy;
        double bd = (in && d < worst) ? d : DBL_MAX;
        int bi = (in && d < worst) ? gl : 1 << 20;
#pragma unroll
        for (int o = W / 2; o > 0; o >>= 1) {
          const double od = __shfl_xor(bd, o, 64);
          const int ob = __shfl_xor(bi, o, 64);
          if (ob < (1 << 20) && (bi >= (1 << 20) || od < bd || (od == bd && ob < bi))) bd = od, bi = ob;
        }
        if (bi < (1 << 20)) worst = bd, ans = pdep[nd.a + bi];
      }
    }
    // up: the deepest level of the path whose far child is still pending and passes mindistsq <= worst (the others below it fail
    // now, which is when the recursion would test them)
    int take = -1;
    {
      int node2 = 0;
      double mind2 = dsq, e0 = di0, e1 = di1;
      for (int l2 = 0; l2 < lvl; l2++) {
        const KdNode m = nodes[node2];
        const double val = m.feat ? qy : qx;
        const bool near1 = (val - m.lo) + (val - m.hi) < 0.0;
        const double cut = near1 ? (val - m.hi) * (val - m.hi) : (val - m.lo) * (val - m.lo);
        const double mo = mind2 + cut - (m.feat ? e1 : e0);
        if (l2 < plen && bit(l2)) {
          mind2 = mo;
          if (m.feat) e1 = cut; else e0 = cut;
          node2 = near1 ? m.b : m.a;
        } else {
          if (mo <= worst) take = l2;
          node2 = near1 ? m.a : m.b;
        }
      }
    }
    if (take < 0) break;
    if (take < 64) {
      path0 = (path0 & ((1ull << take) - 1ull)) | (1ull << take);
      nx = 0;
    } else {
      const int w = min(take >> 6, KD_MAXW) - 1;
      while (nx <= w) pathx[nx++] = 0ull;
      pathx[w] = (pathx[w] & ((1ull << (take & 63)) - 1ull)) | (1ull << (take & 63));
      nx = w + 1;
    }
    plen = take + 1;
  }
  return ans;
}

template <bool WAVE>
AVM_DEV double nn_depth(const avm_fsel_batch& b, const double* kd, int p, double fx_, double fy_) {
  return kd_depth<WAVE ? 64 : 1>(b, kd, p, fx_, fy_);
}

template <bool WAVE>
AVM_DEV bool feature_front(const avm_fsel_batch& b, const double* kd, int p, const double* cam, double fx_, double fy_, int H, double* Ch /*13*6*/, double* Wm /*9*/) {
  const double dep = nn_depth<WAVE>(b, kd, p, fx_, fy_);
  const double nrm = sqrt(fx_ * fx_ + fy_ * fy_ + 1.0);
  const v3 fn = mk3(fx_ / nrm, fy_ / nrm, 1.0 / nrm);  // feature.normalized()
  const v3 feat = dep * fn;
  // pell = t_WC_k1 + q_WC_k1 * feature  (R of q_WC^-1 is the transpose of R(q_WC) for unit quaternions; the
  // oracle rotates with the quaternion itself; cam[1] block stores R(q_WC) too at +21*H.. see setup)
  const double* c1 = cam + 1 * 30;
  const v3 pell = mk3(c1[0], c1[1], c1[2]) + Rmul(c1 + 21, feat);
  int numVisible = 1;
  for (int i = 0; i < H * 6; i++) Ch[i] = 0.0;  // symmetric 3x3 per h: xx xy xz yy yz zz
  double E[6] = {0, 0, 0, 0, 0, 0};
  auto addC = [&](int hidx, v3 u, const double* Rinv2) {
    // Bh = skew(u) * Rinv2 ; C = Bh^T Bh
    double S[9], Bm[9];
    skew9(u, S);
    mat3mul(S, Rinv2, Bm);
    double* C = Ch + hidx * 6;
    C[0] = Bm[0] * Bm[0] + Bm[3] * Bm[3] + Bm[6] * Bm[6];
    C[1] = Bm[0] * Bm[1] + Bm[3] * Bm[4] + Bm[6] * Bm[7];
    C[2] = Bm[0] * Bm[2] + Bm[3] * Bm[5] + Bm[6] * Bm[8];
    C[3] = Bm[1] * Bm[1] + Bm[4] * Bm[4] + Bm[7] * Bm[7];
    C[4] = Bm[1] * Bm[2] + Bm[4] * Bm[5] + Bm[7] * Bm[8];
    C[5] = Bm[2] * Bm[2] + Bm[5] * Bm[5] + Bm[8] * Bm[8];
    for (int k = 0; k < 6; k++) E[k] += C[k];
  };
  for (int h = 2; h <= H; ++h) {
    const double* ch = cam + h * 30;
    const v3 tw = mk3(ch[0], ch[1], ch[2]);
    v3 ue = Rmul(ch + 3, pell - tw);  // q_WC_h^-1 * (pell - t_WC_h)
    const double n = sqrt(dot(ue, ue));
    ue = mk3(ue.x / n, ue.y / n, ue.z / n);
    // PinholeCamera::spaceToPlane with radial-tangential distortion
    const double xu = ue.x / ue.z, yu = ue.y / ue.z;
    const double mx2 = xu * xu, my2 = yu * yu, mxy = xu * yu, rho2 = mx2 + my2;
    const double rad = b.k1 * rho2 + b.k2 * rho2 * rho2;
    const double dxx = xu * rad + 2.0 * b.p1 * mxy + b.p2 * (rho2 + 2.0 * mx2);
    const double dyy = yu * rad + 2.0 * b.p2 * mxy + b.p1 * (rho2 + 2.0 * my2);
    const double pu = b.fx * (xu + dxx) + b.cx, pv = b.fy * (yu + dyy) + b.cy;
    const int iu = (int)round(pu), ivv = (int)round(pv);  // std::round: half away from zero
    // a NaN pixel is outside the image: the reference's double -> int conversion yields INT_MIN for it (x86 cvttsd2si),
    // v_cvt_i32_f64 would yield 0
    if (!(pu == pu && pv == pv && (0 <= iu && iu < b.image_width) && (0 <= ivv && ivv < b.image_height))) continue;
    addC(h - 1, ue, ch + 12);
    ++numVisible;
  }
  if (numVisible == 1) return false;
  addC(0, fn, c1 + 12);
  // W = EtE^-1 (cofactors / det)
  const double a00 = E[0], a01 = E[1], a02 = E[2], a11 = E[3], a12 = E[4], a22 = E[5];
  {
    const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
    const double c10 = a12 * a02 - a01 * a22, c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12;
    const double c20 = a01 * a12 - a11 * a02, c21 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
    const double det = a00 * c00 + a01 * c10 + a02 * c20;
    const double id = 1.0 / det;
    Wm[0] = id * c00, Wm[1] = id * c01, Wm[2] = id * c02, Wm[3] = id * c10, Wm[4] = id * c11, Wm[5] = id * c12, Wm[6] = id * c20,
    Wm[7] = id * c21, Wm[8] = id * c22;
  }
  return true;
}

// feature_front for FOUR candidates per wavefront at once (round 4): candidate u on the 16-lane row u, horizon frame h = 1 + (lane & 15) on
// its lanes.  The frames of a candidate are independent until E = sum_h C_h: every lane does ONE frame's projection, visibility test and C_h
// (the one-candidate form did the H of them one after the other on 64 identical lanes), the nearest cloud point is searched by the row's 16
// lanes (kd_depth<16>), C_h goes to the row's LDS record wlu[6 h' + k] (h' = h - 1; zeros for a frame that does not see the feature, as before), and E is
// summed from there IN THE SAME ORDER as feature_front sums it (frames 2 .. H, then frame 1) - the Deltas are bit-identical to the
// one-candidate form's.  Returns (to every lane of the row) whether the candidate is visible from a second frame; W at wlu[6 H ..].
AVM_DEV bool feature_front4(const avm_fsel_batch& b, const double* kd, int p, const double* cam, int k, bool have, int H, double* wlu) {
  const int lane = threadIdx.x & 63, hl = lane & 15, h = hl + 1;
  const double* xy = b.cand_xy + ((size_t)p * b.max_cand + (have ? k : 0)) * 2;
  const double fx_ = xy[0], fy_ = xy[1];
  // findNNDepth by the row's 16 lanes (kd_depth<16>: the reference's kd-tree search, a leaf's points across the lanes)
  const double dep = kd_depth<16>(b, kd, p, fx_, fy_);
  const double nrm = sqrt(fx_ * fx_ + fy_ * fy_ + 1.0);
  const v3 fn = mk3(fx_ / nrm, fy_ / nrm, 1.0 / nrm);
  const v3 feat = dep * fn;
  const double* c1 = cam + 1 * 30;
  const v3 pell = mk3(c1[0], c1[1], c1[2]) + Rmul(c1 + 21, feat);
  double C[6] = {0, 0, 0, 0, 0, 0};
  bool vis = false;
  if (h <= H) {
    const double* ch = cam + h * 30;
    v3 ue = fn;
    if (h >= 2) {
      const v3 tw = mk3(ch[0], ch[1], ch[2]);
      ue = Rmul(ch + 3, pell - tw);
      const double n = sqrt(dot(ue, ue));
      ue = mk3(ue.x / n, ue.y / n, ue.z / n);
      const double xu = ue.x / ue.z, yu = ue.y / ue.z;
      const double mx2 = xu * xu, my2 = yu * yu, mxy = xu * yu, rho2 = mx2 + my2;
      const double rad = b.k1 * rho2 + b.k2 * rho2 * rho2;
      const double dxx = xu * rad + 2.0 * b.p1 * mxy + b.p2 * (rho2 + 2.0 * mx2);
      const double dyy = yu * rad + 2.0 * b.p2 * mxy + b.p1 * (rho2 + 2.0 * my2);
      const double pu = b.fx * (xu + dxx) + b.cx, pv = b.fy * (yu + dyy) + b.cy;
      const int iu = (int)round(pu), ivv = (int)round(pv);
      vis = pu == pu && pv == pv && (0 <= iu && iu < b.image_width) && (0 <= ivv && ivv < b.image_height);
    }
    if (vis || h == 1) {
      double S[9], Bm[9];
      skew9(ue, S);
      mat3mul(S, ch + 12, Bm);
      C[0] = Bm[0] * Bm[0] + Bm[3] * Bm[3] + Bm[6] * Bm[6];
      C[1] = Bm[0] * Bm[1] + Bm[3] * Bm[4] + Bm[6] * Bm[7];
      C[2] = Bm[0] * Bm[2] + Bm[3] * Bm[5] + Bm[6] * Bm[8];
      C[3] = Bm[1] * Bm[1] + Bm[4] * Bm[4] + Bm[7] * Bm[7];
      C[4] = Bm[1] * Bm[2] + Bm[4] * Bm[5] + Bm[7] * Bm[8];
      C[5] = Bm[2] * Bm[2] + Bm[5] * Bm[5] + Bm[8] * Bm[8];
    }
#pragma unroll
    for (int q = 0; q < 6; q++) wlu[6 * (h - 1) + q] = C[q];
  }
  const unsigned long long bal = __ballot(vis);
  const bool ok = have && ((bal >> (lane & 48)) & 0xffffull) != 0;  // numVisible > 1
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // E in feature_front's order: frames 2 .. H as the loop met them (an invisible frame adds an exact zero), then frame 1
  double E[6];
#pragma unroll
  for (int q = 0; q < 6; q++) {
    double e = 0.0;
    for (int hh = 2; hh <= H; hh++) e += wlu[6 * (hh - 1) + q];
    E[q] = e + wlu[q];
  }
  const double a00 = E[0], a01 = E[1], a02 = E[2], a11 = E[3], a12 = E[4], a22 = E[5];
  const double c00 = a11 * a22 - a12 * a12, c01 = a02 * a12 - a01 * a22, c02 = a01 * a12 - a02 * a11;
  const double c10 = a12 * a02 - a01 * a22, c11 = a00 * a22 - a02 * a02, c12 = a02 * a01 - a00 * a12;
  const double c20 = a01 * a12 - a11 * a02, c21 = a01 * a02 - a00 * a12, c22 = a00 * a11 - a01 * a01;
  const double det = a00 * c00 + a01 * c10 + a02 * c20;
  const double id = 1.0 / det;
  if (hl == 0) {
    double* Wm = wlu + 6 * H;
    Wm[0] = id * c00, Wm[1] = id * c01, Wm[2] = id * c02, Wm[3] = id * c10, Wm[4] = id * c11, Wm[5] = id * c12, Wm[6] = id * c20,
    Wm[7] = id * c21, Wm[8] = id * c22;
  }
  return ok;
}

// block (i, j), 1 <= j <= i <= H, of Delta_ell = blkdiag(C_h) - [C_i W C_j^T] (and its mirror image), dense T x T
AVM_DEV void feature_pair(const double* Ch, const double* Wm, int i, int j, int T, double* out) {
  auto full = [&](int hidx, double* M) {
    const double* C = Ch + hidx * 6;
    M[0] = C[0], M[1] = C[1], M[2] = C[2], M[3] = C[1], M[4] = C[3], M[5] = C[4], M[6] = C[2], M[7] = C[4], M[8] = C[5];
  };
  double Cj[9], Ci[9], CW[9], D[9];
  full(j - 1, Cj);
  full(i - 1, Ci);
  mat3mul(Ci, Wm, CW);
  // Dij = Ci * W * Cj^T
  for (int a = 0; a < 3; a++)
    for (int c = 0; c < 3; c++) D[a * 3 + c] = CW[a * 3] * Cj[c * 3] + CW[a * 3 + 1] * Cj[c * 3 + 1] + CW[a * 3 + 2] * Cj[c * 3 + 2];
  for (int a = 0; a < 3; a++)
    for (int c = 0; c < 3; c++) {
      const int r = 3 * (i - 1) + a, q = 3 * (j - 1) + c;
      if (i == j) {
        out[r * T + q] = Ci[a * 3 + c] - D[a * 3 + c];
      } else {
        out[r * T + q] = -D[a * 3 + c];
        out[q * T + r] = -D[a * 3 + c];
      }
    }
}

// Delta_ell of one feature by one thread (the used subset; the candidates go one per wavefront, see fsel_setup_kernel)
AVM_DEV bool feature_delta(const avm_fsel_batch& b, const double* kd, int p, const double* cam, double fx_, double fy_, int H, double* out /*T*T*/) {
  double Ch[13 * 6], Wm[9];
  if (!feature_front<false>(b, kd, p, cam, fx_, fy_, H, Ch, Wm)) return false;
  for (int j = 1; j <= H; ++j)
    for (int i = j; i <= H; ++i) feature_pair(Ch, Wm, i, j, 3 * H, out);
  return true;
}

// ---- setup: Omega, partial Cholesky of the non-position rows, Delta of every feature ------
// (round 4) Two launches: slice 0 of every frame with the whole carve (N x N doubles of Omega: 104 KB at H = 10, one workgroup per CU), and
// the candidate slices (slice_base = 1, compact) with only what they touch - the camera frames, four wavefronts' C_h / W and Delta tiles,
// 35 KB: four workgroups per CU.  In one launch the candidate slices of a 256-frame batch (8192 workgroups) went through the CUs one at a
// time: 3.2 ms of its 10.4 (profiles/r04c_fsel.md).
__global__ __launch_bounds__(FS_NT) void fsel_setup_kernel(FselDev A, int slice_base, int compact) {
  FS_TABLES_GUARD(A);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  const avm_fsel_batch& b = A.b;
  const int p = blockIdx.x, t = threadIdx.x;
  const int slice = blockIdx.y + slice_base;  // 0: Omega, its partial factorization, the used features; >= 1: candidates [16 (slice-1), 16 slice)
#ifdef FS_TRACE_EVAL
  const long long ts0 = clock64();
#endif
  const int H = b.horizon, N = 9 * (H + 1), T = 3 * H;
  double* Om = lds;                 // N*N (compact: the candidate slices' share of it, see fsel_setup_lds_bytes)
  double* Wh = Om + (compact ? (FS_NT / 64) * (FS_CPW * (6 * H + 9) + T * T) : N * N);  // [H+1][81] Omega_h (h>=1)
  double* Ah = Wh + (H + 1) * 81;   // [H+1][81] Ablk_h
  double* Th = Ah + (H + 1) * 81;   // [H+1][81] At*Omega
  double* cam = compact ? Wh : Th + (H + 1) * 81;  // [H+1][30]
  double* col = cam + (H + 1) * 30; // N
  double* red = col + N;            // 64
  int* isp = reinterpret_cast<int*>(red + 64);  // N: position-row flag
  const double* hp = b.hor_pos + (size_t)p * (H + 1) * 3;
  const double* hq = b.hor_quat + (size_t)p * (H + 1) * 4;
  const quat qic{b.q_ic[3], b.q_ic[0], b.q_ic[1], b.q_ic[2]};
  if (!compact) {
    for (int i = t; i < N * N; i += FS_NT) Om[i] = 0.0;
    for (int i = t; i < N; i += FS_NT) isp[i] = (i >= 9 && (i % 9) < 3) ? 1 : 0;
  }
  // per consecutive pair: createLinearImuMatrices (only slice 0 needs them).  The nr interpolated rotations of a pair are
  // independent: one thread each first (parked in Omega's storage, re-zeroed below), then thread h sums them in the
  // reference's order.  More rotations than fit there: thread h computes them in its loop as before.
  const int nri = b.nr_imu[p];
  const bool rpar = slice == 0 && nri > 0 && (long long)H * nri * 9 <= (long long)N * N;
  if (rpar) {
    __syncthreads();
    for (int idx = t; idx < H * nri; idx += FS_NT) {
      const int h = 1 + idx / nri, i = idx % nri;
      const quat Qi{hq[(h - 1) * 4 + 3], hq[(h - 1) * 4], hq[(h - 1) * 4 + 1], hq[(h - 1) * 4 + 2]};
      const quat Qj{hq[h * 4 + 3], hq[h * 4], hq[h * 4 + 1], hq[h * 4 + 2]};
      q2R(slerp_eigen(Qi, i / (double)nri, Qj), Om + (size_t)idx * 9);
    }
    __syncthreads();
  }
  if (slice == 0 && t >= 1 && t <= H) {
    const int h = t;
    const quat Qi{hq[(h - 1) * 4 + 3], hq[(h - 1) * 4], hq[(h - 1) * 4 + 1], hq[(h - 1) * 4 + 2]};
    const quat Qj{hq[h * 4 + 3], hq[h * 4], hq[h * 4 + 1], hq[h * 4 + 2]};
    const double nr = (double)b.nr_imu[p], dI = b.delta_imu[p];
    double Nij[9], Mij[9];
    for (int k = 0; k < 9; k++) Nij[k] = 0, Mij[k] = 0;
    double c11 = 0, c12 = 0;
    for (int i = 0; i < nr; ++i) {
      double R[9];
      if (rpar) {
        for (int k = 0; k < 9; k++) R[k] = Om[((size_t)(h - 1) * nri + i) * 9 + k];
      } else {
        q2R(slerp_eigen(Qi, i / nr, Qj), R);
      }
      const double jkh = (nr - i - 0.5);
      for (int k = 0; k < 9; k++) Nij[k] += jkh * R[k], Mij[k] += R[k];
      c11 += jkh * jkh;
      c12 += jkh;
    }
    const double d2 = dI * dI, d3 = d2 * dI, d4 = d3 * dI;
    const double ca = 1.0 * nr * c11 * d4 * b.acc_var, cb = 1.0 * c12 * d3 * b.acc_var, cd = 1.0 * nr * d2 * b.acc_var,
                 cc = 1.0 * nr * b.acc_bias_var;
    // inverse of [[ca I, cb I, 0],[cb I, cd I, 0],[0,0,cc I]]
    const double det = ca * cd - cb * cb;
    double* W = Wh + h * 81;
    double* Am = Ah + h * 81;
    for (int k = 0; k < 81; k++) W[k] = 0, Am[k] = 0;
    for (int i = 0; i < 3; i++) {
      W[i * 9 + i] = cd / det, W[i * 9 + 3 + i] = -cb / det, W[(3 + i) * 9 + i] = -cb / det, W[(3 + i) * 9 + 3 + i] = ca / det;
      W[(6 + i) * 9 + 6 + i] = 1.0 / cc;
    }
    for (int i = 0; i < 9; i++) Am[i * 9 + i] = -1.0;
    for (int i = 0; i < 3; i++) Am[i * 9 + 3 + i] = -1.0 * nr * dI;
    for (int a = 0; a < 3; a++)
      for (int c = 0; c < 3; c++) Am[a * 9 + 6 + c] = Nij[a * 3 + c] * d2, Am[(3 + a) * 9 + 6 + c] = Mij[a * 3 + c] * dI;
  }
  // camera frames for calcInfoFromFeatures
  if (t >= 64 && t <= 64 + H) {
    const int h = t - 64;
    const quat q{hq[h * 4 + 3], hq[h * 4], hq[h * 4 + 1], hq[h * 4 + 2]};
    const v3 tw = mk3(hp[h * 3], hp[h * 3 + 1], hp[h * 3 + 2]) + qrot(q, mk3(b.t_ic[0], b.t_ic[1], b.t_ic[2]));
    const quat qwc = qmul(q, qic);
    double* c = cam + h * 30;
    c[0] = tw.x, c[1] = tw.y, c[2] = tw.z;
    q2R(qinv(qwc), c + 3);                 // q_WC^-1
    q2R(qinv(qmul(qwc, qic)), c + 12);     // (q_WC * q_IC)^-1 : q_IC twice, bug-compatible (:304,:321)
    q2R(qwc, c + 21);                      // q_WC (frame k+1 back-projection)
  }
  __syncthreads();
  if (slice > 0) {
    // Delta of this slice's candidates (the camera frames above are all they need), FS_CPW candidates per wavefront: every
    // lane runs the short front part (uniform), lane 0 parks C_h and W in LDS, then the H (H + 1) / 2 block pairs go one
    // per lane - the T x T block is written by 64 lanes at once instead of 900 scattered stores from one thread.
    const int lane = t & 63, wv = t >> 6;
    // (round 4: the four candidates' front parts run side by side, a frame per lane - feature_front4; then the block pairs and the
    //  stores candidate by candidate through the wavefront's one tile)
    const int WS = 6 * H + 9;                                   // a candidate's record: C_h (6 H) | W (9)
    double* wl0 = Om + (wv * FS_CPW) * WS;                      // (Omega's storage is unused in these slices)
    double* tile = Om + (FS_NT / 64) * FS_CPW * WS + wv * T * T;  // (16 (6 H + 9) + 36 H^2 <= 81 (H + 1)^2 doubles of Omega's storage)
    const int npair = H * (H + 1) / 2;
    const int k0 = ((slice - 1) * (FS_NT / 64) + wv) * FS_CPW;
    if (k0 >= b.n_cand[p]) return;  // (wave-uniform)
    const int ku = k0 + (lane >> 4);
    const bool oku = feature_front4(b, A.kd, p, cam, ku, ku < b.n_cand[p], H, wl0 + (lane >> 4) * WS);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (int u = 0; u < FS_CPW; u++) {
      const int k = k0 + u;
      if (k >= b.n_cand[p]) break;  // (wave-uniform)
      const bool ok = __shfl(oku ? 1 : 0, 16 * u, 64) != 0;
      if (ok) {
        const double* wl = wl0 + u * WS;
        double* out = A.delta + ((size_t)p * b.max_cand + k) * T * T;
        // the block pairs are put together in this wavefront's LDS tile and go out as whole rows (written pair by pair - 24-byte
        // pieces, ten to a row, from different lanes at different times - a batch's Deltas cost 6.8 x their size in write traffic)
        for (int q = lane; q < npair; q += 64) {
          int j = 1, rem = q;  // pairs in the order j = 1..H, i = j..H
          while (rem >= H - j + 1) rem -= H - j + 1, j++;
          feature_pair(wl, wl + 6 * H, j + rem, j, T, tile);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int idx = lane; idx < T * T; idx += 64) out[idx] = tile[idx];
        if (A.delta_pk) {  // ... and the lower triangle by columns, for the solo form
          double* opk = A.delta_pk + ((size_t)p * b.max_cand + k) * (T * (T + 1) / 2);
          for (int cc = 0; cc < T; cc++)
            if (cc + lane < T) opk[cc * T - cc * (cc - 1) / 2 + lane] = tile[cc * T + cc + lane];  // (row cc of the symmetric tile = column cc)
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (lane == 0) A.valid[(size_t)p * b.max_cand + k] = ok, A.black[(size_t)p * b.max_cand + k] = 0;
    }
    return;
  }
  if (rpar)
    for (int idx = t; idx < H * nri * 9; idx += FS_NT) Om[idx] = 0.0;  // (the parked rotations)
  for (int idx = t; idx < H * 81; idx += FS_NT) {  // Th = A^T W
    const int h = 1 + idx / 81, i = (idx % 81) / 9, j = idx % 9;
    double s = 0;
    for (int k = 0; k < 9; k++) s += Ah[h * 81 + k * 9 + i] * Wh[h * 81 + k * 9 + j];
    Th[h * 81 + i * 9 + j] = s;
  }
  __syncthreads();
  // assemble Omega: diagonal block d = Omega_d (pair d) + At*Omega*A (pair d+1) [+ I for d == 0]
  for (int idx = t; idx < (H + 1) * 81; idx += FS_NT) {
    const int d = idx / 81, i = (idx % 81) / 9, j = idx % 9;
    double s = 0;
    if (d >= 1) s += Wh[d * 81 + i * 9 + j];
    if (d < H) {  // At*Omega*A of pair d+1
      double s1 = 0;
      for (int k = 0; k < 9; k++) s1 += Th[(d + 1) * 81 + i * 9 + k] * Ah[(d + 1) * 81 + k * 9 + j];
      s += s1;
    }
    if (d == 0 && i == j) s += 1.0;
    Om[(d * 9 + i) * N + d * 9 + j] = s;
  }
  for (int idx = t; idx < H * 81; idx += FS_NT) {
    const int h = 1 + idx / 81, i = (idx % 81) / 9, j = idx % 9;
    const double v = Th[h * 81 + i * 9 + j];
    Om[((h - 1) * 9 + i) * N + h * 9 + j] = v;  // At*Omega
    Om[(h * 9 + j) * N + (h - 1) * 9 + i] = v;  // its transpose
  }
  __syncthreads();
  if (A.omega_out)
    for (int i = t; i < N * N; i += FS_NT) A.omega_out[(size_t)p * N * N + i] = Om[i];
  // constants of the Hadamard bound + original position diagonal
  double kn = 0;
  for (int i = t; i < N; i += FS_NT)
    if (!isp[i]) kn += log(Om[i * N + i]);
  kn = block_sum<FS_NT>(kn, red);
  if (t < T) A.dpp[(size_t)p * T + t] = Om[(9 * (1 + t / 3) + t % 3) * (N + 1)];
  __syncthreads();
  // Partial right-looking elimination of the non-position rows (ascending order), square-root free: row i loses
  // (A_ik / d_k) A_kj.  Omega is block tridiagonal, so a pivot of state s only reaches the rows of states s and s + 1 and -
  // through fill - the position rows of the states before s: at most 3 (s - 1) + 18 <= 54 rows, listed once per state.  Four
  // threads per listed row, every fourth listed column each (<= 14): all loads of a pivot are in flight at once, one trip
  // through LDS and ONE workgroup barrier per pivot - nobody writes row k while it is being read, and the multiplier is the
  // row's own column-k entry.  Column k is zeroed as it is consumed, so eliminated columns need no mask later; the pivots
  // are parked for the logarithms.  (The previous form - two threads per row over all columns, a column buffer and two
  // barriers per pivot - took 150 us of a single frame's select.)
  double* piv = col;  // (col[] has no other use any more)
#ifdef FS_TRACE_EVAL
  const long long ts1 = clock64();
#endif
  for (int st = 0; st <= H; st++) {
    const int npre = st >= 1 ? 3 * (st - 1) : 0;
    const int na = npre + 9 + (st < H ? 9 : 0);
    auto rowof = [&](int a) { return a < npre ? 9 * (1 + a / 3) + a % 3 : 9 * st + (a - npre); };
    const int a = t >> 2, q = t & 3;
    const int i = rowof(min(a, na - 1));
    const bool ipos = i >= 9 && (i % 9) < 3;
    constexpr int MC = 14;  // ceil(54 / 4)
    int cj[MC];
#pragma unroll
    for (int m = 0; m < MC; m++) cj[m] = q + 4 * m < na ? rowof(q + 4 * m) : -1;
    double* row = Om + i * N;
    for (int kk = st == 0 ? 0 : 3; kk < 9; kk++) {
      const int k = 9 * st + kk;
      const double dkk = Om[k * N + k];
      double inv = __builtin_amdgcn_rcp(dkk), e = fma(-dkk, inv, 1.0);
      inv = fma(inv, e, inv);
      e = fma(-dkk, inv, 1.0);
      inv = fma(inv, e, inv);
      const bool act = a < na && i != k && (i > k || ipos);
      // (every load of the pivot is requested before the first use: one trip through LDS)
      const double* rk = Om + k * N;
      const double xik = row[k];
      double rv[MC], xk[MC];
#pragma unroll
      for (int m = 0; m < MC; m++) rv[m] = row[max(cj[m], 0)], xk[m] = rk[max(cj[m], 0)];
      const double li = act ? xik * inv : 0.0;
      if (li != 0.0) {
#pragma unroll
        for (int m = 0; m < MC; m++)  // (an unlisted slot goes to a dump slot: no predicated LDS store)
          *(cj[m] >= 0 ? row + cj[m] : red + (t & 63)) = cj[m] == k ? 0.0 : rv[m] - li * xk[m];
      }
      __syncthreads();
      if (t == 0) piv[k] = dkk;
    }
  }
  __syncthreads();
  double ld = 0;
  for (int i = t; i < N; i += FS_NT)
    if (!isp[i]) ld += log(piv[i]);
  ld = 0.5 * block_sum<FS_NT>(ld, red);
#ifdef FS_TRACE_EVAL
  if (t == 0) A.consts[(size_t)p * 4 + 2] = (double)(clock64() - ts1), A.consts[(size_t)p * 4 + 3] = (double)(ts1 - ts0);
#endif
  double* C = A.C + (size_t)p * T * T;
  for (int idx = t; idx < T * T; idx += FS_NT) {
    const int i = idx / T, j = idx % T;
    C[idx] = Om[(9 * (1 + i / 3) + i % 3) * N + 9 * (1 + j / 3) + j % 3];
  }
  if (t == 0) {
    A.consts[(size_t)p * 4] = 2.0 * ld;
    A.consts[(size_t)p * 4 + 1] = kn;
    A.nsel[p] = 0;
    A.done[p] = 0;
  }
  // Delta of the already-used subset (the candidates are done by the other slices of the grid).  feature_delta writes
  // every entry of the T x T block unless it returns false, and an invalid feature's block is never read.
  const int nu = b.n_used ? b.n_used[p] : 0;
  for (int k = t; k < nu; k += FS_NT) {
    const double* xy = b.used_xy + ((size_t)p * b.max_used + k) * 2;
    A.valid_u[(size_t)p * b.max_used + k] = feature_delta(b, A.kd, p, cam, xy[0], xy[1], H, A.delta_u + ((size_t)p * b.max_used + k) * T * T);
  }
  __syncthreads();
  // Omega += sum of Delta_used (ascending id order = input order)
  for (int idx = t; idx < T * T; idx += FS_NT) {
    double s = C[idx], dd = 0;
    for (int u = 0; u < nu; u++)
      if (A.valid_u[(size_t)p * b.max_used + u]) {
        const double v = A.delta_u[((size_t)p * b.max_used + u) * T * T + idx];
        s += v;
        dd += v;
      }
    C[idx] = s;
    if (idx / T == idx % T) A.dpp[(size_t)p * T + idx / T] += dd;
  }
}

// ---- one greedy round: f_l = logdet(Omega + OmegaS + p_l Delta_l) for every live candidate ----
AVM_DEV double fs_readlane_d(double v, int srclane) {  // srclane must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// FOUR candidates per wavefront: candidate g lives in the 16-lane DPP row g of the wave.  The T x T matrix is cut into NB block
// rows of BS <= 16 rows (T = 30: 2 x 15, T = 39: 3 x 13); lane r of the row holds row r of EVERY block row in registers
// (block row bi: columns 0 .. (bi + 1) BS - 1), so an entry A[gk][gj] is broadcast to the whole candidate with a DPP
// row_newbcast of lane gk % BS - since round 6 as the DPP operand of the multiply-add itself (fs_fmac_bcast below: v_fmac_f64_dpp, one instruction per
// update; before: two 32-bit DPP moves feeding the updates of all block rows).  The factorization is the same right-looking, square-root-free
// LDL^T as before (column j divided by its pivot with v_rcp_f64 + two Newton steps; junk above the diagonal of the diagonal
// blocks is computed and never read), only the lanes are used four times as densely and there are no SGPR round trips:
// 1-3 DPP multiply-adds per (pivot, column) pair for four candidates instead of 2 v_readlane + 1 FMA for one.
// logdet = sum_j log(d_j) and the Hadamard bound (sortedlogDetUB) are summed in one fixed association for every candidate, so
// mirror-image candidates still get bit-identical bounds (the std::map rule of the pick depends on that).
template <int K>
AVM_DEV double fs_rowbcast_k(double v) {  // lane K of every 16-lane row -> the whole row (row_newbcast:K = dpp_ctrl 0x150 + K)
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, true);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
AVM_DEV double fs_rowbcast(double v, int k) {  // k is a compile-time constant at every call site (fully unrolled loops)
  switch (k) {
    case 0: return fs_rowbcast_k<0>(v);
    case 1: return fs_rowbcast_k<1>(v);
    case 2: return fs_rowbcast_k<2>(v);
    case 3: return fs_rowbcast_k<3>(v);
    case 4: return fs_rowbcast_k<4>(v);
    case 5: return fs_rowbcast_k<5>(v);
    case 6: return fs_rowbcast_k<6>(v);
    case 7: return fs_rowbcast_k<7>(v);
    case 8: return fs_rowbcast_k<8>(v);
    case 9: return fs_rowbcast_k<9>(v);
    case 10: return fs_rowbcast_k<10>(v);
    case 11: return fs_rowbcast_k<11>(v);
    case 12: return fs_rowbcast_k<12>(v);
    case 13: return fs_rowbcast_k<13>(v);
    case 14: return fs_rowbcast_k<14>(v);
    default: return fs_rowbcast_k<15>(v);
  }
}

// acc += (lane k of src's 16-lane row) * nmul in ONE instruction: v_fmac_f64_dpp with row_newbcast (the FP64 ALU of gfx90a+ takes a DPP operand
// of that one kind).  Round 6, scripts/ubench/dpp2.hip: 5.8 cycles an issue against 4.8 for a plain v_fmac_f64 - and it does accumulate; the round-3
// probe (scripts/ubench/dpp.hip: "assembles but does not accumulate") issued it right behind the VALU write of its source, and a DPP read needs two wait
// states there which the compiler's hazard recognizer does not add inside inline assembly.  The caller keeps those two wait states (fs_dpp_fence) between the
// last write of any operand and the first instruction of a run; inside a run of these nothing reads what a neighbour writes.
#ifndef FS_NO_FMAC_DPP
#define FS_FMAC_DPP 1
#endif
AVM_DEV void fs_dpp_fence() { asm volatile("s_nop 1"); }
template <int K>
AVM_DEV void fs_fmac_bcast(double& acc, double src, double nmul) {
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(nmul), "n"(K));
}
// compile-time loops (the lane index of a DPP operand is part of the instruction)
template <class F, int... Is>
AVM_DEV void fs_sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
AVM_DEV void fs_sfor(F&& f) {
  fs_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int FS_CPWG = (FS_NT / 64) * 4;  // candidates per workgroup of the round kernel

// State that changes from round to round exists twice (C, dpp, the live list and its inverse, nlive, fval, ub: buffer `par` at
// offset par * size): launch k reads the buffers k & 1 and writes the others, so that no workgroup of a launch reads what another
// one writes.  That lets ONE launch per round do both halves of a greedy step:
//   1. every workgroup picks the winner of the previous round for itself from the values the previous launch left (a few KB,
//      the same deterministic argmax everywhere; workgroup 0 of the problem also records it and writes the next buffers:
//      C + p Delta_winner, the live list with the winner swap-removed);
//   2. it evaluates its candidates against that next state, which it patches in on the fly (the same expressions workgroup 0
//      stores, so the values are bit-identical to the stored ones).
// Round 1 and the first half of round 2 launched a pick kernel between two evaluations: 300 dependent launches per select, and the
// gaps between them were 40 % of the time.  Now 151.
struct FselPar {
  const double *C, *dpp, *fval, *ub;
  const int32_t *live, *pos;
  double *Cn, *dppn, *fvaln, *ubn;
  int32_t *liven, *posn, *nliven;
  int nl;
};
AVM_DEV FselPar fsel_par(const FselDev& A, int p, int k) {
  const avm_fsel_batch& b = A.b;
  const int T = 3 * b.horizon, cur = k & 1, nxt = cur ^ 1;
  const size_t P = b.n_problems, mc = b.max_cand, TT = (size_t)T * T;
  FselPar r;
  r.C = A.C + (cur * P + p) * TT, r.Cn = A.C + (nxt * P + p) * TT;
  r.dpp = A.dpp + (cur * P + p) * T, r.dppn = A.dpp + (nxt * P + p) * T;
  r.fval = A.fval + (cur * P + p) * mc, r.fvaln = A.fval + (nxt * P + p) * mc;
  r.ub = A.ub + (cur * P + p) * mc, r.ubn = A.ub + (nxt * P + p) * mc;
  r.live = A.live + (cur * P + p) * mc, r.liven = A.live + (nxt * P + p) * mc;
  r.pos = A.pos + (cur * P + p) * mc, r.posn = A.pos + (nxt * P + p) * mc;
  r.nl = A.nlive[cur * P + p], r.nliven = A.nlive + nxt * P + p;
  return r;
}

// Maximum over the wavefront, in every lane: four DPP exchange steps inside the 16-lane rows (lane ^ 1, lane ^ 2, mirror of 8,
// mirror of 16 - any pairing of already-reduced groups will do for a maximum), then the four row results through SGPRs.
// (__shfl_xor is a ds_bpermute per 32 bits and step: the lexicographic argmax below took 30 of them, 2 K cycles.)
template <int CTRL>
AVM_DEV double fs_dpp_d(double v) {
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
// Sum over the 16 lanes of a DPP row, in every lane of the row: the same four exchange steps as fs_wave_max (a fixed
// association, the same for every candidate - mirror-image candidates keep bit-identical Hadamard bounds).
AVM_DEV double fs_row_sum(double v) {
  v += fs_dpp_d<0xB1>(v);   // quad_perm [1,0,3,2]
  v += fs_dpp_d<0x4E>(v);   // quad_perm [2,3,0,1]
  v += fs_dpp_d<0x141>(v);  // row_half_mirror
  v += fs_dpp_d<0x140>(v);  // row_mirror
  return v;
}
AVM_DEV double fs_wave_max(double v) {
  v = fmax(v, fs_dpp_d<0xB1>(v));   // quad_perm [1,0,3,2]
  v = fmax(v, fs_dpp_d<0x4E>(v));   // quad_perm [2,3,0,1]
  v = fmax(v, fs_dpp_d<0x141>(v));  // row_half_mirror
  v = fmax(v, fs_dpp_d<0x140>(v));  // row_mirror
  const double r0 = fs_readlane_d(v, 0), r1 = fs_readlane_d(v, 16), r2 = fs_readlane_d(v, 32), r3 = fs_readlane_d(v, 48);
  return fmax(fmax(r0, r1), fmax(r2, r3));
}
AVM_DEV int fs_wave_max(int v) {
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0xB1, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x4E, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x141, 0xf, 0xf, true));
  v = max(v, __builtin_amdgcn_update_dpp(0, v, 0x140, 0xf, 0xf, true));
  return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)),
             max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---- the round's winner (feature_selector.cpp:669-683), computed by every workgroup of the problem for itself ---------------
// returns the winner's candidate index (-1: none) to all threads; *fwin its value; *frun (when the caller asked for avm_fsel_out::min_gap)
// the largest value among the OTHER candidates of the round (-HUGE_VAL: nobody else took part)
AVM_DEV int fsel_pick_local(const FselDev& A, const FselPar& S, double* fwin, double* frun) {
  __shared__ double s_f[FS_NT / 64], s_u[FS_NT / 64];
  __shared__ int s_i[FS_NT / 64];
  __shared__ int s_win;
  const int t = threadIdx.x;
  const int32_t* live = S.live;
  const int nl = S.nl;
  auto better = [](double f, double u, int i, double f2, double u2, int i2) {
    if (i2 < 0) return false;
    if (i < 0) return true;
    return f2 > f || (f2 == f && (u2 > u || (u2 == u && i2 > i)));
  };
  // sortedlogDetUB keeps the upper bounds in a std::map<double, int> (feature_selector.cpp:724): of two live candidates
  // with BIT-IDENTICAL upper bounds only the later (higher) id survives the round, the other one is never scored.  The
  // argmax below therefore runs until its winner is not shadowed by a higher id with the same key; `shadowed` holds the
  // (at most a handful of) candidates that were ruled out this way.  One extra pass over the bounds in the usual case.
  constexpr int MAXSH = 8;
  __shared__ int s_shadow[MAXSH];
  __shared__ int s_nsh, s_hit;
  __syncthreads();  // (the shared variables may still be read by a slower wavefront of this workgroup's previous use)
  if (t == 0) s_nsh = 0;
  __syncthreads();
  // this thread's candidates (at most FS_PC of them) stay in registers for every pass of the loop below
  constexpr int FS_PC = 4;  // (candidates beyond FS_PC * FS_NT = 1024 are re-read in every pass)
  int cl[FS_PC];
  double cf[FS_PC], cu[FS_PC];
  // (values are stored by SLOT of the live list they were computed for - the list of these buffers - so the three loads are
  //  independent: one trip to memory)
#pragma unroll
  for (int q = 0; q < FS_PC; q++) {
    const int sq = min(t + q * FS_NT, max(nl - 1, 0));
    cl[q] = live[sq], cf[q] = S.fval[sq], cu[q] = S.ub[sq];
    if (t + q * FS_NT >= nl) cl[q] = -1;
  }
  double bf;
  int bi;
  for (;;) {
    // lexicographic max of (fValue, ub, id) over live candidates with fValue > fMax0 = -1.0 (NaN never wins)
    bf = -1.0;
    double bu = -DBL_MAX;
    bi = -1;
    const int nsh = s_nsh;
#pragma unroll
    for (int q = 0; q < FS_PC; q++) {
      const int l = cl[q];
      bool sh = l < 0;
      for (int qq = 0; qq < nsh; qq++) sh |= s_shadow[qq] == l;
      const double f = cf[q], u = cu[q];
      if (sh || !(f > -1.0)) continue;
      if (bi < 0 || f > bf || (f == bf && (u > bu || (u == bu && l > bi)))) bf = f, bu = u, bi = l;
    }
    for (int s = t + FS_PC * FS_NT; s < nl; s += FS_NT) {
      const int l = live[s];
      bool sh = false;
      for (int qq = 0; qq < nsh; qq++) sh |= s_shadow[qq] == l;
      const double f = S.fval[s], u = S.ub[s];
      if (sh || !(f > -1.0)) continue;
      if (bi < 0 || f > bf || (f == bf && (u > bu || (u == bu && l > bi)))) bf = f, bu = u, bi = l;
    }
    {  // the wavefront's best: three maxima in a row, each over the lanes that tie in the previous ones
      const double wf = fs_wave_max(bi >= 0 ? bf : -1.0);
      const bool tf = bi >= 0 && bf == wf;
      const double wu = fs_wave_max(tf ? bu : -DBL_MAX);
      const bool tu = tf && bu == wu;
      bi = fs_wave_max(tu ? bi : -1), bf = wf, bu = wu;
    }
    if ((t & 63) == 0) s_f[t >> 6] = bf, s_u[t >> 6] = bu, s_i[t >> 6] = bi;
    __syncthreads();
    if (t == 0) {
      for (int w = 1; w < FS_NT / 64; w++)
        if (better(bf, bu, bi, s_f[w], s_u[w], s_i[w])) bf = s_f[w], bu = s_u[w], bi = s_i[w];
      s_win = bi, s_f[0] = bf, s_u[0] = bu, s_hit = 0;
    }
    __syncthreads();
    const int cand = s_win;
    if (cand < 0) break;
    const double cuw = s_u[0];
    int hit = 0;
#pragma unroll
    for (int q = 0; q < FS_PC; q++)  // a live candidate with a higher id and the same key?
      if (cl[q] > cand && cu[q] == cuw) hit = 1;
    for (int s = t + FS_PC * FS_NT; s < nl; s += FS_NT)
      if (live[s] > cand && S.ub[s] == cuw) hit = 1;
    if (hit) s_hit = 1;
    __syncthreads();
    if (!s_hit || s_nsh >= MAXSH || A.no_key_rule) break;  // (more than MAXSH chained collisions in one round: keep the last winner)
    __syncthreads();
    if (t == 0) s_shadow[s_nsh++] = cand;
    __syncthreads();
  }
  *fwin = s_f[0];
  if (A.out.min_gap) {  // the runner-up: the best value among the others that took part (the candidates the std::map rule ruled out above did not)
    const int wl = s_win, nsh = s_nsh;
    double r2 = -HUGE_VAL;
#pragma unroll
    for (int q = 0; q < FS_PC; q++) {
      const int l = cl[q];
      bool sh = l < 0 || l == wl;
      for (int qq = 0; qq < nsh; qq++) sh |= s_shadow[qq] == l;
      if (!sh && cf[q] > -1.0) r2 = fmax(r2, cf[q]);
    }
    for (int sq = t + FS_PC * FS_NT; sq < nl; sq += FS_NT) {
      const int l = live[sq];
      bool sh = l == wl;
      for (int qq = 0; qq < nsh; qq++) sh |= s_shadow[qq] == l;
      const double f = S.fval[sq];
      if (!sh && f > -1.0) r2 = fmax(r2, f);
    }
    r2 = fs_wave_max(r2);
    __syncthreads();  // (s_u is free: every thread has read the winner's bound)
    if ((t & 63) == 0) s_u[t >> 6] = r2;
    __syncthreads();
    double rr = s_u[0];
#pragma unroll
    for (int w = 1; w < FS_NT / 64; w++) rr = fmax(rr, s_u[w]);
    *frun = rr;
  }
  return s_win;
}

// The same pick for the single-frame kernel (slot s IS candidate s; the caller hands in this thread's two candidates - index -1 =
// not in the race - with the values it has read): two workgroup barriers per pass, the shadow list in registers, and the
// lexicographic maximum of (fValue, ub, id) as three maxima in a row, each over the lanes that tie in the previous ones.
AVM_DEV int fsel_pick_frame(const FselDev& A, const int* cl, const double* cf, const double* cu, double* fwin, double* frun) {
  __shared__ double s_f[2][FS_NT / 64], s_u[2][FS_NT / 64];
  __shared__ int s_i[2][FS_NT / 64], s_h[2][FS_NT / 64];
  const int t = threadIdx.x, wv = t >> 6;
  constexpr int MAXSH = 8;
  int sh[MAXSH], nsh = 0;
#pragma unroll
  for (int qq = 0; qq < MAXSH; qq++) sh[qq] = -1;
  for (int pass = 0;; pass++) {
    const int sl = pass & 1;
    double bf = -1.0, bu = -DBL_MAX;
    int bi = -1;
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int l = cl[q];
      bool out = l < 0;
#pragma unroll
      for (int qq = 0; qq < MAXSH; qq++) out |= sh[qq] == l;
      const double f = cf[q], u = cu[q];
      if (out || !(f > -1.0)) continue;
      if (bi < 0 || f > bf || (f == bf && (u > bu || (u == bu && l > bi)))) bf = f, bu = u, bi = l;
    }
    {  // the wavefront's best (a lane without a candidate carries f = -1, which no candidate in the race has)
      const double wf = fs_wave_max(bi >= 0 ? bf : -1.0);
      const bool tf = bi >= 0 && bf == wf;
      const double wu = fs_wave_max(tf ? bu : -DBL_MAX);
      const bool tu = tf && bu == wu;
      const int wi = fs_wave_max(tu ? bi : -1);
      if ((t & 63) == 0) s_f[sl][wv] = wf, s_u[sl][wv] = wu, s_i[sl][wv] = wi;
    }
    __syncthreads();
    bf = s_f[sl][0], bu = s_u[sl][0], bi = s_i[sl][0];
#pragma unroll
    for (int w = 1; w < FS_NT / 64; w++) {
      const double f2 = s_f[sl][w], u2 = s_u[sl][w];
      const int i2 = s_i[sl][w];
      if (i2 >= 0 && (bi < 0 || f2 > bf || (f2 == bf && (u2 > bu || (u2 == bu && i2 > bi))))) bf = f2, bu = u2, bi = i2;
    }
    *fwin = bf;
    // the runner-up for avm_fsel_out::min_gap (see fsel_pick_local): one more maximum, only when it was asked for
    auto runner_up = [&](int wl) {
      if (!A.out.min_gap) return;
      double r2 = -HUGE_VAL;
#pragma unroll
      for (int q = 0; q < 2; q++) {
        const int l = cl[q];
        bool out = l < 0 || l == wl;
#pragma unroll
        for (int qq = 0; qq < MAXSH; qq++) out |= sh[qq] == l;
        if (!out && cf[q] > -1.0) r2 = fmax(r2, cf[q]);
      }
      r2 = fs_wave_max(r2);
      __syncthreads();  // (every thread has read slot sl of s_u)
      if ((t & 63) == 0) s_u[sl][wv] = r2;
      __syncthreads();
      double rr = s_u[sl][0];
#pragma unroll
      for (int w = 1; w < FS_NT / 64; w++) rr = fmax(rr, s_u[sl][w]);
      *frun = rr;
    };
    if (bi < 0 || A.no_key_rule || nsh >= MAXSH) {  // (more than MAXSH chained collisions in one round: keep the last winner)
      runner_up(bi);
      return bi;
    }
    // std::map rule (see fsel_pick_local): a live candidate with a higher id and the same key shadows the winner
    const bool hit = (cl[0] > bi && cu[0] == bu) || (cl[1] > bi && cu[1] == bu);
    const bool wh = __any(hit);
    if ((t & 63) == 0) s_h[sl][wv] = wh ? 1 : 0;
    __syncthreads();
    int any = 0;
#pragma unroll
    for (int w = 0; w < FS_NT / 64; w++) any |= s_h[sl][w];
    if (!any) {
      runner_up(bi);
      return bi;
    }
#pragma unroll
    for (int qq = 0; qq < MAXSH; qq++)
      if (qq == nsh) sh[qq] = bi;
    nsh++;
  }
}

// Natural logarithm of a positive, normal, finite double (every argument here is a pivot or a diagonal entry that has already
// passed `> 0`; anything else gives finite junk or NaN, which the callers discard): the fdlibm reduction - x = 2^k (1 + f),
// sqrt(1/2) <= 1 + f < sqrt(2), s = f / (2 + f), log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)) with the degree-14 minimax R - with
// the quotient from v_rcp_f64 + two Newton steps + a residual correction.  < 1 ulp like the library's, in 45 instead of ~80
// instructions: the evaluation takes five logarithms per candidate and round.
AVM_DEV double fs_log(double x) {
  int k = __builtin_amdgcn_frexp_exp(x);           // x = m 2^k, 1/2 <= m < 1
  double m = __builtin_amdgcn_frexp_mant(x);
  const bool lo = m < 0.70710678118654752440;
  m = lo ? m + m : m;
  k = lo ? k - 1 : k;
  const double f = m - 1.0, d = 2.0 + f;
  double r = __builtin_amdgcn_rcp(d), e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  e = fma(-d, r, 1.0);
  r = fma(r, e, r);
  double sq = f * r;
  sq = fma(fma(-d, sq, f), r, sq);                  // s = f / (2 + f)
  const double z = sq * sq, w = z * z;
  const double t1 = w * fma(w, fma(w, 1.531383769920937332e-01, 2.222219843214978396e-01), 3.999999999940941908e-01);
  const double t2 = z * fma(w, fma(w, fma(w, 1.479819860511658591e-01, 1.818357216161805012e-01), 2.857142874366239149e-01), 6.666666666666735130e-01);
  const double R = t2 + t1, hfsq = 0.5 * f * f, dk = (double)k;
  return dk * 6.93147180369123816490e-01 - ((hfsq - (sq * (hfsq + R) + dk * 1.90821492927058770002e-10)) - f);
}

// logdet(C + pr D) and the Hadamard bound for the candidate of this lane's 16-lane row (see the comment above fs_rowbcast_k):
// *ld_out = sum_j log(sqrt(d_j)) in pivot order, *ub_out = sum_i log((dpp + pr D)_ii); returns false on a non-positive pivot.
// sC / sdpp: the frame's current reduced information and position diagonal (LDS), D: the candidate's Delta (global).
// PHASED (the single-frame kernel, one wavefront per SIMD): the phases are kept apart in the schedule; the compiler's own
// interleaving of the loads, the logarithms and the elimination was measured 10 % slower there - and 6 % faster on the batched
// path, where a second wavefront fills the gaps.
#ifdef FS_TRACE_EVAL
#define FS_TK(i) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); const long long n__ = clock64(); if (tk) tk[i] += n__ - tkp; tkp = n__; __builtin_amdgcn_sched_barrier(0); }
#else
#define FS_TK(i) if (PHASED) __builtin_amdgcn_sched_barrier(0);
#endif
// (Measured and dropped: the multipliers as LDS broadcasts - a column written once, one ds_read per gk instead of two DPP moves
//  per pair, the round trip hidden by look-ahead - made the evaluation 10-17 % slower.)
// PACKED: D is the packed lower triangle (row R at R (R + 1) / 2, the single-frame kernel's LDS copy) instead of the full
// T x T block; an entry past the diagonal of a diagonal block - never used, see above - then reads into the next row.
// PACKED == 2: D is the lower triangle BY COLUMNS (FselDev::delta_pk): the 15 lanes of a candidate still read consecutive doubles.
template <int T, int BS, int NB, bool PHASED = false, int PACKED = 0>
AVM_DEV bool fsel_logdet4(const double* sC, const double* sdpp, const double* D, double pr, double* ld_out, double* ub_out, long long* tk = nullptr) {
#ifdef FS_TRACE_EVAL
  long long tkp = clock64();
#endif
  const int lane = threadIdx.x & 63;
  const int r = min(lane & 15, BS - 1);  // this lane's row inside every block row
  // m[bi][c] = (C + p Delta)[bi BS + r][c], c < (bi + 1) BS.  Both matrices are symmetric, so the entry is fetched as
  // [c][bi BS + r]: the 15 lanes of a candidate then read 15 consecutive doubles instead of 15 different cache lines
  double m[NB][T];
  double ddg[NB];  // the candidate's diagonal entries of this lane's rows (the Hadamard bound)
  if constexpr (PACKED == 2) {
    // D comes from MEMORY here (fsel_solo_kernel): a block row's entries are all requested before the first one is used.  Left to itself the
    // compiler pairs each load with its multiply-add and keeps one or two in flight - 45 dependent trips to the L2, 15.3 K of an
    // evaluation's 23.2 K cycles (round 5, profiles/r05_fsel_single_frame_floor.md).  One block row at a time (15 + 30 entries at 3 H = 30,
    // 13 + 26 + 39 at 39): every entry of the candidate at once costs registers the elimination needs (49 spilled at 30, 6 % slower at 39).
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
#pragma unroll
      for (int c = 0; c < (bi + 1) * BS; c++) {
        const int R = bi * BS + r;
        m[bi][c] = D[c <= R ? c * T - c * (c - 1) / 2 + (R - c) : R * T - R * (R - 1) / 2];  // (an entry past the diagonal is never used)
      }
      const int dgi = bi * BS + r;
      ddg[bi] = D[dgi * T - dgi * (dgi - 1) / 2];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int c = 0; c < (bi + 1) * BS; c++) m[bi][c] = sC[c * T + bi * BS + r] + pr * m[bi][c];
      __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int bi = 0; bi < NB; bi++)
#pragma unroll
      for (int c = 0; c < (bi + 1) * BS; c++) {
        const int idx = c * T + bi * BS + r, R = bi * BS + r;
        m[bi][c] = sC[idx] + pr * D[PACKED == 1 ? R * (R + 1) / 2 + c : idx];
      }
#pragma unroll
    for (int bi = 0; bi < NB; bi++) {
      const int dgi = bi * BS + r;
      ddg[bi] = D[PACKED == 1 ? dgi * (dgi + 1) / 2 + dgi : dgi * T + dgi];
    }
  }
  FS_TK(0)
  // Hadamard upper bound: sum over the rows, block row by block row, then across the 16 lanes in lane order
  double ubl = 0.0;
#pragma unroll
  for (int bi = 0; bi < NB; bi++) {
    const int dgi = bi * BS + r;
    ubl += fs_log(sdpp[dgi] + pr * ddg[bi]);
  }
  const double ubt = fs_row_sum((lane & 15) < BS ? ubl : 0.0);
  FS_TK(1)
  double dkeep[NB];  // lane j keeps the pivot of row bj BS + j
  bool bad = false;
#ifdef FS_FMAC_DPP
  // m[bi][gk] += A[gk][gj] * (-mult[bi]) with A[gk][gj] = lane k of m[bk][gj], taken by the multiply-add itself (fs_fmac_bcast): one instruction per
  // (pivot, column, block row) where it was two 32-bit DPP moves per (pivot, column) and a multiply-add per block row - the same product, the same rounding
  fs_sfor<NB>([&](auto BJ) {
    constexpr int bj = BJ;
    dkeep[bj] = 1.0;
    fs_sfor<BS>([&](auto J) {
      constexpr int j = J, gj = bj * BS + j;
      const double djj = fs_rowbcast_k<j>(m[bj][gj]);
      if (!(djj > 0.0)) bad = true;
      dkeep[bj] = (lane & 15) == j ? djj : dkeep[bj];
      double y = __builtin_amdgcn_rcp(djj), e = fma(-djj, y, 1.0);
      y = fma(y, e, y);
      e = fma(-djj, y, 1.0);
      y = fma(y, e, y);
      double nmult[NB];
#pragma unroll
      for (int bi = bj; bi < NB; bi++) nmult[bi] = -(m[bi][gj] * y);
      fs_dpp_fence();
      fs_sfor<NB - bj>([&](auto BKK) {
        constexpr int bk = bj + BKK, k0 = bk == bj ? j + 1 : 0;
        fs_sfor<BS - k0>([&](auto KK) {
          constexpr int k = k0 + KK, gk = bk * BS + k;
          fs_sfor<NB - bk>([&](auto BII) {
            constexpr int bi = bk + BII;
            fs_fmac_bcast<k>(m[bi][gk], m[bk][gj], nmult[bi]);
          });
        });
      });
      fs_dpp_fence();  // (the next pivot's broadcast reads an entry this run has written)
    });
  });
#else
#pragma unroll
  for (int bj = 0; bj < NB; bj++) {
    dkeep[bj] = 1.0;
#pragma unroll
    for (int j = 0; j < BS; j++) {
      const int gj = bj * BS + j;
      const double djj = fs_rowbcast(m[bj][gj], j);
      if (!(djj > 0.0)) bad = true;
      dkeep[bj] = (lane & 15) == j ? djj : dkeep[bj];
      double y = __builtin_amdgcn_rcp(djj), e = fma(-djj, y, 1.0);
      y = fma(y, e, y);
      e = fma(-djj, y, 1.0);
      y = fma(y, e, y);
      double mult[NB];
#pragma unroll
      for (int bi = bj; bi < NB; bi++) mult[bi] = m[bi][gj] * y;
#pragma unroll
      for (int bk = bj; bk < NB; bk++)
#pragma unroll
        for (int k = (bk == bj ? j + 1 : 0); k < BS; k++) {
          const int gk = bk * BS + k;
          const double v = fs_rowbcast(m[bk][gj], k);  // A[gk][gj]
#pragma unroll
          for (int bi = bk; bi < NB; bi++) m[bi][gk] = fma(-mult[bi], v, m[bi][gk]);
        }
    }
  }
#endif
  FS_TK(2)
  // log(sqrt(d)): per lane over its block rows, then across the candidate's lanes
  double ldl = 0;
#pragma unroll
  for (int bj = 0; bj < NB; bj++) ldl += dkeep[bj] > 0.0 ? 0.5 * fs_log(dkeep[bj]) : 0.0;
  const double ld = fs_row_sum((lane & 15) < BS ? ldl : 0.0);
  FS_TK(3)
  *ld_out = ld, *ub_out = ubt;
  return !bad;
}


// The Hadamard bound alone, for the candidate of this lane's 16-lane row: the same expression, in the same order, as the bound
// inside fsel_logdet4.  dd[d * stride]: the T diagonal entries of the candidate's Delta (fsel_solo_kernel takes every candidate's bound from
// here, scored or not, so the equal-key rule and the (fValue, bound, id) order of the pick see one function).
template <int T, int BS, int NB>
AVM_DEV double fsel_ub4(const double* sdpp, const double* dd, int stride, double pr) {
  const int lane = threadIdx.x & 63;
  const int r = min(lane & 15, BS - 1);
  double ubl = 0.0;
#pragma unroll
  for (int bi = 0; bi < NB; bi++) {
    const int dgi = bi * BS + r;
    ubl += fs_log(sdpp[dgi] + pr * dd[dgi * stride]);
  }
  return fs_row_sum((lane & 15) < BS ? ubl : 0.0);
}


AVM_DEV double fs_rsqrt(double x) {  // v_rsq_f64 + two Newton steps (about one ulp on normal positive numbers)
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * y * y);
  y = y * (1.5 - (0.5 * x) * y * y);
  return y;
}
AVM_DEV void wave_lds_sync_fs() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// ---- the evaluation on the matrix cores (round 3) ------------------------------------------------------------------------------
// logdet(C + pr D) of FOUR candidates per wavefront by a blocked Cholesky factorization with 4 x 4 tiles whose rank-4 updates run on
// v_mfma_f64_4x4x4_4b: one instruction does four INDEPENDENT 4 x 4 x 4 products - one per candidate.  Operand layout (measured,
// scripts/ubench/mfma4.hip): block b = (lane / 4) % 4 is a quad COLUMN of the wavefront, the k index is the 16-lane row:
//     A[b][i][k] at lane 16 k + 4 b + i,   B[b][k][j] at lane 16 k + 4 b + j,   D[b][i][j] at lane 16 i + 4 b + j.
// Candidate b's matrix is held as its UPPER tiles U[k][i] (k <= i) in the D layout, one double per lane and tile - lane (li = lane
// / 16, lj = lane % 4) holds entry (4 k + li, 4 i + lj).  With that choice nothing is ever transposed or moved between lanes:
//   step k:  L_kk L_kk^T = U[k][k]                     the 4 x 4 diagonal tile, gathered through LDS and factored REDUNDANTLY by the 16
//                                                      lanes of the block (no broadcast inside the pivot chain)
//            W_i = L_kk^-1 U[k][i]        (i > k)      one MFMA each: A = L_kk^-1 (every lane selects its entry), B = the tile as it is;
//                                                      W_i = L_ik^T comes out in the D layout ...
//            U[j][i] -= W_j^T W_i     (k < j <= i)     ... which is at once the A operand (read as the transpose) and the B operand
//                                                      of the trailing update: one MFMA per tile, the tile itself the accumulator.
// 36 tiles at 3 H = 30 (padded to 32 with an identity block: log 1 = 0), 28 + 84 MFMAs per evaluation at 17.7 cycles each
// (scripts/ubench/mfma4_rate.hip) against 435 (pivot, column) pairs of two 32-bit DPP moves and one or two FP64 multiply-adds each
// in the DPP formulation (fsel_logdet4 above, which stays in use where it is the faster one: see fsel_frame_kernel).
// The logarithms are spread over the block's lanes like before (four per lane and evaluation): lane (li, lj) takes pivot lj of the
// steps k = li (mod 4) and the diagonal entries 4 k + li of the Hadamard bound for k = lj (mod 4); the sums over the block's 16
// lanes use the same order in every block, so candidates with bit-identical inputs get bit-identical values wherever they sit
// (the std::map equal-key rule of sortedlogDetUB depends on that).
// sC / sdpp: the frame's current reduced information and position diagonal (LDS); D, pr: THIS LANE'S candidate (uniform over the
// block); gather: 64 doubles of LDS scratch per wavefront.  *ld_out = sum_j log(sqrt(d_j)), *ub_out = sum_i log((dpp + pr D)_ii),
// valid in every lane of the block; returns false on a non-positive pivot (block-uniform).
AVM_DEV double fs_blk_sum(double v) {  // sum over the 16 lanes {16 i + 4 b + j} of this lane's block, the same order in every lane
  v += fs_dpp_d<0xB1>(v);                       // quad_perm [1,0,3,2]: j ^ 1
  v += fs_dpp_d<0x4E>(v);                       // quad_perm [2,3,0,1]: j ^ 2
  v += __shfl_xor(v, 16, 64);                   // li ^ 1
  v += __shfl_xor(v, 32, 64);                   // li ^ 2
  return v;
}
template <int T, bool PACKED>
AVM_DEV bool fsel_logdet4m(const double* sC, const double* sdpp, const double* D, double pr, double* gather, double* ld_out, double* ub_out) {
  constexpr int NT4 = (T + 3) / 4;
  const int lane = threadIdx.x & 63, li = lane >> 4, lj = lane & 3, blk = (lane >> 2) & 3;
  auto dget = [&](int R, int Cc) {  // D[R][Cc], symmetric; PACKED: the lower triangle, row R at R (R + 1) / 2
    const int hi = max(R, Cc), lo = min(R, Cc);
    return PACKED ? D[hi * (hi + 1) / 2 + lo] : D[hi * T + lo];
  };
  // ---- tiles
  double U[NT4 * (NT4 + 1) / 2];
#pragma unroll
  for (int k = 0; k < NT4; k++)
#pragma unroll
    for (int i = k; i < NT4; i++) {
      const int R = 4 * k + li, Cc = 4 * i + lj;
      const bool in = R < T && Cc < T;
      const int Rc = min(R, T - 1), Ccc = min(Cc, T - 1);
      const double v = sC[Rc * T + Ccc] + pr * dget(Rc, Ccc);
      U[k * NT4 - k * (k - 1) / 2 + (i - k)] = in ? v : (R == Cc ? 1.0 : 0.0);
    }
  // ---- Hadamard upper bound: this lane's diagonal entries 4 k + li, k = lj (mod 4)
  double ubl = 0.0;
#pragma unroll
  for (int q = 0; q < (NT4 + 3) / 4; q++) {
    const int r = 4 * (lj + 4 * q) + li;
    const int rc = min(r, T - 1);
    const double dv = sdpp[rc] + pr * dget(rc, rc);
    ubl += (r < T && lj + 4 * q < NT4) ? fs_log(dv) : 0.0;
  }
  const double ubt = fs_blk_sum(ubl);
  // ---- factorization
  double pv[(NT4 + 3) / 4];  // the pivots this lane takes the logarithm of
#pragma unroll
  for (int q = 0; q < (NT4 + 3) / 4; q++) pv[q] = 1.0;
  bool bad = false;
  double* gb = gather + blk * 16;
#pragma unroll
  for (int k = 0; k < NT4; k++) {
    const int dk = k * NT4 - k * (k - 1) / 2;  // U[k][k]
    gb[li * 4 + lj] = U[dk];
    wave_lds_sync_fs();
    double a[4][4];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c <= r; c++) a[r][c] = gb[c * 4 + r];  // (the upper triangle holds the current values: entry (c, r), c <= r)
    wave_lds_sync_fs();  // (the next step's writes wait for these reads)
    // 4 x 4 Cholesky, replicated; pivots d_j, their reciprocal square roots, L, L^-1
    double l[4][4], m[4][4], d[4], rs[4];
    d[0] = a[0][0];
    rs[0] = fs_rsqrt(d[0]);
    l[1][0] = a[1][0] * rs[0], l[2][0] = a[2][0] * rs[0], l[3][0] = a[3][0] * rs[0];
    d[1] = fma(-l[1][0], l[1][0], a[1][1]);
    rs[1] = fs_rsqrt(d[1]);
    l[2][1] = fma(-l[2][0], l[1][0], a[2][1]) * rs[1], l[3][1] = fma(-l[3][0], l[1][0], a[3][1]) * rs[1];
    d[2] = fma(-l[2][1], l[2][1], fma(-l[2][0], l[2][0], a[2][2]));
    rs[2] = fs_rsqrt(d[2]);
    l[3][2] = fma(-l[3][1], l[2][1], fma(-l[3][0], l[2][0], a[3][2])) * rs[2];
    d[3] = fma(-l[3][2], l[3][2], fma(-l[3][1], l[3][1], fma(-l[3][0], l[3][0], a[3][3])));
    rs[3] = fs_rsqrt(d[3]);
    bad |= !(d[0] > 0.0) || !(d[1] > 0.0) || !(d[2] > 0.0) || !(d[3] > 0.0);
    m[0][0] = rs[0], m[1][1] = rs[1], m[2][2] = rs[2], m[3][3] = rs[3];
    m[1][0] = -(l[1][0] * m[0][0]) * rs[1];
    m[2][1] = -(l[2][1] * m[1][1]) * rs[2];
    m[2][0] = -fma(l[2][1], m[1][0], l[2][0] * m[0][0]) * rs[2];
    m[3][2] = -(l[3][2] * m[2][2]) * rs[3];
    m[3][1] = -fma(l[3][2], m[2][1], l[3][1] * m[1][1]) * rs[3];
    m[3][0] = -fma(l[3][2], m[2][0], fma(l[3][1], m[1][0], l[3][0] * m[0][0])) * rs[3];
    // the pivot whose logarithm this lane takes: pivot lj of the steps k = li (mod 4)
    {
      const double dsel = lj == 0 ? d[0] : (lj == 1 ? d[1] : (lj == 2 ? d[2] : d[3]));
      pv[k / 4] = (k & 3) == li ? dsel : pv[k / 4];
    }
    if (k + 1 < NT4) {
      // A operand of W = L^-1 U: A[i'][k'] = (L^-1)[i'][k'] at lane 16 k' + 4 b + i', i.e. this lane needs (L^-1)[lj][li]
      double asel = 0.0;
#pragma unroll
      for (int r = 0; r < 4; r++)
#pragma unroll
        for (int c = 0; c <= r; c++) asel = (lj == r && li == c) ? m[r][c] : asel;
      double W[NT4];
#pragma unroll
      for (int i = k + 1; i < NT4; i++) W[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(asel, U[dk + (i - k)], 0.0, 0, 0, 0);
#pragma unroll
      for (int j = k + 1; j < NT4; j++) {
        const double nw = -W[j];
#pragma unroll
        for (int i = j; i < NT4; i++) {
          const int t_ji = j * NT4 - j * (j - 1) / 2 + (i - j);
          U[t_ji] = __builtin_amdgcn_mfma_f64_4x4x4f64(nw, W[i], U[t_ji], 0, 0, 0);
        }
      }
    }
  }
  // log(sqrt(d)): this lane's pivots, then across the block
  double ldl = 0.0;
#pragma unroll
  for (int q = 0; q < (NT4 + 3) / 4; q++) ldl += (li + 4 * q < NT4 && pv[q] > 0.0) ? 0.5 * fs_log(pv[q]) : 0.0;
  *ld_out = fs_blk_sum(ldl), *ub_out = ubt;
  // (a non-positive pivot anywhere in the block's factorization: every lane of the block saw it)
  return !bad;
}

// One greedy step of workgroup `bx` of problem p: settle round k - 1, evaluate round k.  Returns true when the problem is
// finished (the same answer in every workgroup of the problem: it depends on the shared state only).
template <int T, int BS, int NB>
AVM_DEV bool fsel_round_body(const FselDev& A, int p, int k, int bx) {
  static_assert(BS * NB == T && BS <= 16, "block rows of at most 16 lanes");
  const avm_fsel_batch& b = A.b;
  const int t = threadIdx.x;
  const int kappa = max(0, b.max_features - (b.n_used ? b.n_used[p] : 0));
  if (A.done[p]) return true;  // (set by an earlier round: the state is frozen)
  const bool has_pick = k >= 1 && k <= kappa, has_eval = k < kappa;
  if (!has_pick && !has_eval) return true;
  const FselPar S = fsel_par(A, p, k);
  // ---- 1. the previous round's winner
  int win = -1;
  double fwin = 0.0, frun = -HUGE_VAL;
  if (has_pick) {
    win = fsel_pick_local(A, S, &fwin, &frun);
    if (win < 0) {
      if (bx == 0 && t == 0) A.done[p] = 1;  // lMax == -1: nothing is added; later rounds would repeat the same state
      return true;
    }
  }
  const bool won = win >= 0;
  const int wc = max(win, 0);
  const int nl = S.nl, nln = won ? nl - 1 : nl;
  const int at = won ? S.pos[wc] : -1, lastc = S.live[max(nl - 1, 0)];  // swap-remove: the last candidate takes the winner's slot
  const double prw = b.cand_prob[(size_t)p * b.max_cand + wc];
  const double* Dw = A.delta + ((size_t)p * b.max_cand + wc) * T * T;
  if (bx == 0) {  // this problem's recorder: outputs and the next buffers
    if (won && t == 0) {
      const int ks = A.nsel[p];
      A.out.selected_ids[(size_t)p * b.max_features + ks] = b.cand_id[(size_t)p * b.max_cand + win];
      if (A.out.fvalues) A.out.fvalues[(size_t)p * b.max_features + ks] = fwin;
      if (A.out.min_gap) A.out.min_gap[(size_t)p * b.max_features + ks] = fwin - frun;
      A.nsel[p] = ks + 1;
      A.out.n_selected[p] = ks + 1;
      A.black[(size_t)p * b.max_cand + win] = 1;
    }
    for (int idx = t; idx < T * T; idx += FS_NT) {
      const double c = won ? S.C[idx] + prw * Dw[idx] : S.C[idx];
      S.Cn[idx] = c;
      if (idx / T == idx % T) S.dppn[idx / T] = won ? S.dpp[idx / T] + prw * Dw[idx] : S.dpp[idx / T];
    }
    for (int s = t; s < nln; s += FS_NT) {
      const int l = s == at ? lastc : S.live[s];
      S.liven[s] = l, S.posn[l] = s;
    }
    if (t == 0) *S.nliven = nln;
  }
  if (!has_eval) return true;
  // ---- 2. this round's candidates against the state with the winner folded in: every workgroup builds it in LDS (the same
  //         expressions workgroup 0 stores), and the candidates' matrices take their C part from there
  __shared__ double sC[T * T], sdpp[T];
  for (int idx = t; idx < T * T; idx += FS_NT) {
    const double c = won ? S.C[idx] + prw * Dw[idx] : S.C[idx];
    sC[idx] = c;
    if (idx / T == idx % T) sdpp[idx / T] = won ? S.dpp[idx / T] + prw * Dw[idx] : S.dpp[idx / T];
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int g = lane >> 4;  // candidate slot of this lane
  // the candidates still in the race are kept compact (the winner is swap-removed), so late rounds do not pay for
  // the slots of the features already selected
  const int slot = (bx * (FS_NT / 64) + wv) * 4 + g;
  const bool live = slot < nln;
  if (!__any(live)) return false;  // (wave-uniform; no workgroup barrier follows in this function)
  const int sc = min(slot, max(nln - 1, 0));
  const int l = sc == at ? lastc : S.live[sc];
  const int lc = l;  // a slot past the end factors the last live candidate's matrix again and throws the result away
  const double pr = b.cand_prob[(size_t)p * b.max_cand + lc];
  const double* D = A.delta + ((size_t)p * b.max_cand + lc) * T * T;
  const double ld_nn = A.consts[(size_t)p * 4], ub_nn = A.consts[(size_t)p * 4 + 1];  // (requested before the evaluation, not after it)
  double ld, ubt;
  const bool bad = !fsel_logdet4<T, BS, NB>(sC, sdpp, D, pr, &ld, &ubt);
  if (live && (lane & 15) == 0) {
    const double f = bad ? __builtin_nan("") : (ld_nn + 2.0 * ld);
    S.fvaln[sc] = f;  // (by slot of the next live list: see fsel_pick_local)
    S.ubn[sc] = ub_nn + ubt;
  }
  return false;
}

// (amdgpu_waves_per_eu(2, 2): a batch puts two of these wavefronts on a SIMD; left to itself the scheduler trades the
//  evaluation's instruction-level parallelism for an occupancy the launch never reaches - measured 0.22 -> 0.30 ms per frame)
template <int T, int BS, int NB>
__global__ __launch_bounds__(FS_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void fsel_round_kernel(FselDev A, int k) {
  FS_TABLES_GUARD(A);
  (void)fsel_round_body<T, BS, NB>(A, blockIdx.y, k, blockIdx.x);
}

// ---- all rounds of a frame in ONE launch (a select() is 151 dependent steps) ----------------------------------------------------
// Measured on the launch-per-round path: a step is bound by its chain of dependent trips to memory (state, live list, values,
// the winner's Delta, the candidates' Delta: each a miss, because a kernel boundary invalidates the caches), not by the launch.
// Here the frame's state never leaves the compute unit: every workgroup keeps its own copy of C, of the position diagonal and
// of the candidates' alive flags in LDS and applies the same deterministic update (winner out, C += p Delta_winner) to it.  The
// only thing exchanged per round is (fValue, ub) of each workgroup's 16 candidates - a fixed assignment by candidate index, no
// live list.
// There is no barrier and no fence.  A value travels as a 16-byte record {value, round tag, check word} written with ONE store
// and read with ONE device-scope load (a single request each, never served by the vector L1; two parity buffers): a reader spins
// until the records of all candidates still in the race carry the round's tag, and then it has the values - one trip after the
// last writer, nothing to order, no cache maintenance.  Everything else the kernel reads from global memory was written before
// the launch.  The workgroups that exchange records sit on ONE XCD (a team, see fsel_frame_kernel): the records never leave
// that XCD's L2 - a trip is ~0.5 us instead of a trip across the fabric.
// A spin longer than FS_SPIN_TICKS of the 100 MHz clock raises sync[2], every participant leaves, and the host runs the call
// again one mode down (and stays there).
constexpr long long FS_SPIN_TICKS = 20 * 100000;  // 20 ms
constexpr int FS_FRAME_MAXC = 512;                // candidates of a frame on the single-launch path (32 slots of 16)
// {value, round tag, check}: `check` = the value's two halves xor-ed with the tag.  The 16 bytes travel as one request, but
// nothing in the ISA promises that a concurrent reader cannot see them half-written: a record counts as arrived only when its
// tag is the round's AND its check matches its value.
struct alignas(16) FselRec {
  double v;
  int32_t tag, chk;
};
AVM_DEV int fsel_rec_check(int lo, int hi, int tag) { return lo ^ hi ^ (tag * 0x9E3779B1); }
template <bool SC1>
AVM_DEV void fsel_rec_load2(const FselRec* pa, const FselRec* pb, FselRec* a, FselRec* b) {  // device-scope loads
  typedef int v4i __attribute__((ext_vector_type(4)));
  v4i va, vb;
  asm volatile("global_load_dwordx4 %0, %2, off sc1\n\tglobal_load_dwordx4 %1, %3, off sc1\n\ts_waitcnt vmcnt(0)"
               : "=&v"(va), "=&v"(vb)
               : "v"(pa), "v"(pb)
               : "memory");
  a->v = __hiloint2double(va[1], va[0]), a->tag = va[3] == fsel_rec_check(va[0], va[1], va[2]) ? va[2] : -1;
  b->v = __hiloint2double(vb[1], vb[0]), b->tag = vb[3] == fsel_rec_check(vb[0], vb[1], vb[2]) ? vb[2] : -1;
}
template <bool SC1>
AVM_DEV void fsel_rec_store(FselRec* p, double v, int tag) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i x = {__double2loint(v), __double2hiint(v), tag, fsel_rec_check(__double2loint(v), __double2hiint(v), tag)};
  if (SC1) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(p), "v"(x) : "memory");
}

// TEAMS: the workgroups of every XCD form a team of `nslots` (first come, first slot; the rest exit), and a team takes frames from
// a queue until it is empty - a batch of P frames runs as up to eight independent selects side by side, each inside one L2.
// A team's leader (slot 0) hands out the frame: it waits for the team to be complete (first frame) or for everybody's `done`
// (later frames: nobody may still be reading the last frame's records), takes the next frame number and publishes it as a tagged
// record; a team that does not fill up within FS_TEAM_TICKS (its XCD is busy with something else) dissolves and leaves the frames
// to the others.  The host checks that every frame was finished (sync[4]) and runs the launch-per-round path otherwise.
// !TEAMS: one team over the whole device (blockIdx.x = slot, records written through to memory), one frame: the first fallback.
constexpr long long FS_TEAM_TICKS = 2 * 100000;  // 2 ms
constexpr int FS_TEAM_HDR = 32;                   // ints per team header: [0] members [1] done [4..7] the frame assignment record
constexpr int FS_SYNC_HDR = 64;                   // ints: [2] failure [3] frame queue [4] frames finished [8..15] arrivals per XCD [16..17] evaluations executed (solo form, 64-bit) [32..] trace
constexpr int FS_MAX_TEAMS = 16;
// TPX = 2 (batches of more than eight frames, 3H <= 30): TWO teams per XCD, i.e. two wavefronts per SIMD - the second one fills the
// latency gaps of the first (a team alone is bound by dependent latencies, not by issue).  Two workgroups then share a compute
// unit's LDS, so the Delta copies are packed lower triangles (16 x 3.7 KB).
template <int T, int BS, int NB, int TPX>
AVM_DEV void fsel_frame_body(const FselDev& A, int32_t* sync, int nslots, int test_drop) {
  FS_TABLES_GUARD(A);
  constexpr bool TEAMS = TPX > 0;
  // the workgroup's 16 Delta matrices stay in LDS for the whole select: full blocks while they fit (3H <= 30: 16 x 7.2 KB), packed
  // lower triangles beyond (3H = 39: 16 x 6.2 KB; the packed indexing costs 3 % at 3H = 30) or when two workgroups share the LDS
  constexpr bool PACKD = T > 30 || TPX > 1;
  constexpr int PK = PACKD ? T * (T + 1) / 2 : T * T;
  __shared__ int s_slot, s_fail, s_frame;
  __shared__ double sC[T * T], sdpp[T];
  __shared__ int32_t s_alive[FS_FRAME_MAXC];
  extern __shared__ double s_delta[];
  // Which evaluation: the matrix-core form (fsel_logdet4m) where the instruction issue rate is the limit - two teams per XCD, i.e. two
  // wavefronts per SIMD, 3 H <= 32 - and the DPP form (fsel_logdet4) where a wavefront has its SIMD to itself and the evaluation's
  // dependent chain is what counts (a single frame: 1.38 ms against 1.46) or where the 55 tiles of 3 H = 39 do not fit the registers
  // next to everything else (90 spilled registers, 0.33 -> 0.37 ms per frame).  Measured: profiles/r03_experiments.md, 3.
  // Round 6: with the broadcast folded into the multiply-add (fs_fmac_bcast) the DPP form issues 45 % fewer instructions and wins there too - 16 frames per
  // call 0.140 -> 0.119 ms per frame, 32 frames 0.127 -> 0.104 - so the matrix-core form is only built on request (-DFS_MF).
#ifdef FS_MF
  constexpr bool MF = TPX == 2 && T <= 32;
#else
  constexpr bool MF = false;
#endif
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int g = MF ? (lane >> 2) & 3 : lane >> 4;  // this lane's candidate slot: its MFMA block (quad column) / its 16-lane DPP row
  const bool rec_lane = MF ? (lane & 0x33) == 0 : (lane & 15) == 0;  // one lane per candidate writes the records
  __shared__ double s_gather[MF ? (FS_NT / 64) * 64 : 1];
  int bx = blockIdx.x, team = 0;
  if (TEAMS) {
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(20, 0, 4)" : "=s"(xcc));  // HW_REG_XCC_ID[3:0]
    xcc &= 7;
    if (t == 0) s_slot = __hip_atomic_fetch_add(&sync[8 + xcc], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const int arrival = s_slot;  // order of arrival on this XCD: the first nslots are its first team, ...
    if (arrival >= TPX * nslots) return;
    team = xcc * TPX + arrival / nslots, bx = arrival % nslots;
    __syncthreads();
  }
  int32_t* th = sync + FS_SYNC_HDR + team * FS_TEAM_HDR;
  if (TEAMS && t == 0) __hip_atomic_fetch_add(&th[0], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (t == 0) s_fail = 0;
  FselRec* recF = reinterpret_cast<FselRec*>(sync + FS_SYNC_HDR + FS_MAX_TEAMS * FS_TEAM_HDR) + (size_t)team * 4 * FS_FRAME_MAXC;  // [2][MAXC] fValues
  FselRec* recU = recF + 2 * FS_FRAME_MAXC;                                                                      // [2][MAXC] bounds
  FselRec* assign = reinterpret_cast<FselRec*>(th + 4);
  const avm_fsel_batch& b = A.b;
  const int mc = b.max_cand, P = b.n_problems;
  const int l = (bx * (FS_NT / 64) + wv) * 4 + g;  // this block's candidate index, in every frame
  const int lc = min(l, mc - 1);
  auto give_up = [&]() { __hip_atomic_store(&sync[2], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
#ifdef FS_TRACE_EVAL
  long long tk_body = 0, tk_wait = 0, tk_pick = 0, tk_upd = 0, tk0 = clock64();
  long long tke[4] = {0, 0, 0, 0};
#define FS_SEG(acc) { const long long n__ = clock64(); acc += n__ - tk0; tk0 = n__; }
#else
#define FS_SEG(acc)
#endif
  for (int seq = 1;; seq++) {
    // ---- which frame
    int p = 0;
    if (TEAMS) {
      __syncthreads();  // (s_frame / s_fail of the previous frame have been read)
      if (t == 0) {
        const long long t0 = wall_clock64();
        if (bx == 0) {  // the leader
          int f = -2;   // (-2: the team never filled up)
          for (;;) {
            const int have = seq == 1 ? __hip_atomic_load(&th[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                                      : __hip_atomic_load(&th[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) / (seq - 1);
            if (have >= nslots) {
              f = __hip_atomic_fetch_add(&sync[3], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              if (f >= P) f = -1;
              break;
            }
            if (wall_clock64() - t0 > (seq == 1 ? FS_TEAM_TICKS : FS_SPIN_TICKS)) {
              if (seq > 1) give_up();  // (a member got lost in the middle of the batch)
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
          fsel_rec_store<false>(assign, (double)f, seq);
        }
        FselRec ra, rb;
        for (;;) {
          fsel_rec_load2<true>(assign, assign, &ra, &rb);
          if (ra.tag == seq) break;
          if (__hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > 2 * FS_SPIN_TICKS) {
            give_up();
            ra.v = -3.0;
            break;
          }
          __builtin_amdgcn_s_sleep(2);
        }
        s_frame = (int)ra.v;
      }
      __syncthreads();
      p = s_frame;
      if (p < 0) return;
    } else if (seq > 1) {
      return;
    }
    // ---- the frame's state, this workgroup's copy
    const int nc = b.n_cand[p];
    const int kappa = max(0, b.max_features - (b.n_used ? b.n_used[p] : 0));
    const size_t pc = (size_t)p * mc;
    for (int idx = t; idx < T * T; idx += FS_NT) sC[idx] = A.C[(size_t)p * T * T + idx];
    for (int idx = t; idx < T; idx += FS_NT) sdpp[idx] = A.dpp[(size_t)p * T + idx];
    for (int c = t; c < FS_FRAME_MAXC; c += FS_NT) s_alive[c] = (c < nc && A.valid[pc + min(c, mc - 1)] != 0) ? 1 : 0;
    for (int q = 0; q < FS_CPWG; q++) {
      const double* src = A.delta + (pc + min(bx * FS_CPWG + q, mc - 1)) * T * T;
      for (int idx = t; idx < T * T; idx += FS_NT) {
        const int R = idx / T, c = idx % T;
        if (!PACKD) s_delta[q * PK + idx] = src[idx];
        else if (c >= R) s_delta[q * PK + c * (c + 1) / 2 + R] = src[idx];  // (slot (c, R) <- entry [R][c]: the entry the full form reads for it)
      }
    }
    const double pr = b.cand_prob[pc + lc];
    const double* D = s_delta + (wv * 4 + g) * PK;
    const double ld_nn = A.consts[(size_t)p * 4], ub_nn = A.consts[(size_t)p * 4 + 1];  // logdet of the hoisted pivots / their share of the bound
    const int tag0 = seq << 12;  // (round tags of different frames never meet: max_features < 4096 on this path)
    __syncthreads();
    int nsel = 0;
    for (int k = 0; k <= kappa; k++) {
      // ---- 1. the previous round's winner (its values are in parity buffer (k - 1) & 1, tagged k)
      if (k >= 1) {
        const int par = (k - 1) & 1;
        int cl[2];
        double cf[2], cu[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {  // this thread's candidates: t and t + FS_NT
          const int sq = t + q * FS_NT;
          cl[q] = (sq < nc && s_alive[sq]) ? sq : -1;
          const FselRec *pf = recF + par * FS_FRAME_MAXC + sq, *pu = recU + par * FS_FRAME_MAXC + sq;
          FselRec rf, ru;
          const long long t0 = wall_clock64();
          for (;;) {
            fsel_rec_load2<true>(pf, pu, &rf, &ru);
            if (__all(cl[q] < 0 || (rf.tag == tag0 + k && ru.tag == tag0 + k))) break;
            // A record of THIS frame that already carries a later round's tag: its writer is two rounds ahead and has overwritten the
            // value this workgroup still needed.  That can only happen to a workgroup none of whose own candidates is alive (nobody
            // waits for its records, so nobody is held back by it); the value is gone - leave at once instead of spinning into the
            // time-out (the host runs the call again one mode down, avm_fsel_fallback_stats counts it).
            const bool overtaken = cl[q] >= 0 && (((rf.tag >> 12) == seq && rf.tag > tag0 + k) || ((ru.tag >> 12) == seq && ru.tag > tag0 + k));
            if (__any(overtaken) || __hip_atomic_load(&sync[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0 || wall_clock64() - t0 > FS_SPIN_TICKS) {
              give_up();
              s_fail = 1;
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
          cf[q] = rf.v, cu[q] = ru.v;
        }
        FS_SEG(tk_wait)
        double fwin, frun = -HUGE_VAL;
        const int win = fsel_pick_frame(A, cl, cf, cu, &fwin, &frun);  // (a workgroup barrier inside: s_fail is settled after it)
        FS_SEG(tk_pick)
        if (s_fail) return;
        if (win < 0) break;  // lMax == -1: nothing is added; later rounds would repeat the same state
        if (bx == 0 && t == 0) {  // this frame's recorder
          A.out.selected_ids[(size_t)p * b.max_features + nsel] = b.cand_id[pc + win];
          if (A.out.fvalues) A.out.fvalues[(size_t)p * b.max_features + nsel] = fwin;
          if (A.out.min_gap) A.out.min_gap[(size_t)p * b.max_features + nsel] = fwin - frun;
          A.out.n_selected[p] = nsel + 1;
          A.black[pc + win] = 1;
        }
        nsel++;
        const double* Dw = A.delta + (pc + win) * T * T;
        // (all of the thread's entries of the winner's Delta - and its probability - requested before the first is used: one trip to memory,
        //  not one per entry; round 5)
        constexpr int NFOLD = (T * T + FS_NT - 1) / FS_NT;
        double dwv[NFOLD];
#pragma unroll
        for (int q = 0; q < NFOLD; q++) dwv[q] = Dw[min(t + q * FS_NT, T * T - 1)];
        const double prw = b.cand_prob[pc + win];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < NFOLD; q++) {
          const int idx = t + q * FS_NT;
          if (idx < T * T) {
            sC[idx] = sC[idx] + prw * dwv[q];
            if (idx / T == idx % T) sdpp[idx / T] = sdpp[idx / T] + prw * dwv[q];
          }
        }
        if (t == 0) s_alive[win] = 0;
        __syncthreads();
        FS_SEG(tk_upd)
      }
      if (k >= kappa) break;
      // ---- 2. this round's values of this workgroup's candidates, published with tag k + 1
      const bool live = l < nc && s_alive[min(l, FS_FRAME_MAXC - 1)] != 0;
      if (__any(live)) {
        double ld, ubt;
        bool ok;
        if constexpr (MF) {
          ok = fsel_logdet4m<T, PACKD>(sC, sdpp, D, pr, s_gather + wv * 64, &ld, &ubt);
        } else {
#ifdef FS_TRACE_EVAL
          ok = fsel_logdet4<T, BS, NB, true, PACKD>(sC, sdpp, D, pr, &ld, &ubt, tke);
#else
          ok = fsel_logdet4<T, BS, NB, true, PACKD>(sC, sdpp, D, pr, &ld, &ubt);
#endif
        }
        if (live && rec_lane && l != test_drop) {  // (test_drop: a record that never arrives, tests only; -1 otherwise)
          fsel_rec_store<!TEAMS>(recF + (k & 1) * FS_FRAME_MAXC + l, ok ? (ld_nn + 2.0 * ld) : __builtin_nan(""), tag0 + k + 1);
          fsel_rec_store<!TEAMS>(recU + (k & 1) * FS_FRAME_MAXC + l, ub_nn + ubt, tag0 + k + 1);
        }
      }
      __syncthreads();  // (sC / s_alive are read by the evaluation above and written by the next round's update)
      FS_SEG(tk_body)
    }
#ifdef FS_TRACE_EVAL  // (development: cycles per phase of workgroup 0 of the team that took frame 0, printed with AVM_FSEL_TRACE=1)
    if (t == 0 && bx == 0 && p == 0) {
      long long* o = reinterpret_cast<long long*>(sync + 32);
      o[0] = tk_pick, o[1] = tk_upd, o[2] = tk_body, o[3] = (long long)A.consts[2], o[4] = tk_wait;
      o[9] = (long long)A.consts[3];
      o[5] = tke[0], o[6] = tke[1], o[7] = tke[2], o[8] = tke[3];
    }
#endif
    if (t == 0) {
      if (bx == 0) {
        A.nsel[p] = nsel;
        __hip_atomic_fetch_add(&sync[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // frames finished
      }
      if (TEAMS) __hip_atomic_fetch_add(&th[1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // done with this frame's records
    }
  }
#undef FS_SEG
}

// The kernel proper.  The instances that evaluate on the matrix cores (two teams per XCD, 3 H <= 32) are pinned to two wavefronts
// per SIMD: left alone the allocator takes 334 registers for them, one wavefront per SIMD, and the second team of an XCD never
// becomes resident (every launch then times out and falls back).  The DPP instances are left to the scheduler - the attribute
// costs them 3-4 %.
template <int T, int BS, int NB, int TPX>
__global__ __launch_bounds__(FS_NT) void fsel_frame_kernel(FselDev A, int32_t* sync, int nslots, int test_drop) {
  fsel_frame_body<T, BS, NB, TPX>(A, sync, nslots, test_drop);
}
template <int T, int BS, int NB>
__global__ __launch_bounds__(FS_NT) __attribute__((amdgpu_waves_per_eu(2, 2))) void fsel_frame_kernel_mf(FselDev A, int32_t* sync, int nslots, int test_drop) {
  fsel_frame_body<T, BS, NB, 2>(A, sync, nslots, test_drop);
}

// the compact list of the candidates that take part in the greedy rounds: the valid ones, in ascending index (= id) order
// ---- SOLO: one workgroup per frame, lazy evaluation (batches of many frames) -------------------------------------------------------
// The frame kernel above spreads ONE frame's 425 evaluations per round over a team of workgroups and pays an exchange of records per
// round; a batch larger than the number of teams queues.  Here a frame belongs to one workgroup from its first round to its last -
// no records, no waiting for anybody, hundreds of frames side by side - which only pays because a round does not have to score every
// candidate: the objective is submodular (every p Delta is positive semidefinite), so a candidate's gain f_l(S) - logdet C(S) can only
// shrink as features are added, and the gain it had when it was last scored, g_l, bounds its value now: f_l <= logdet C + g_l (Minoux'
// accelerated greedy; logdet C is the last winner's value).  Per round:
//   1. every live candidate's Hadamard bound (the std::map equal-key rule and the order of the pick need all of them);
//   2. the candidates with g_l >= lazy_tau x (the last winner's gain) are scored, four per wavefront, Delta straight from memory;
//   3. the pick among the scored ones, and its check: a candidate that was NOT scored and whose bound logdet C + g_l + margin reaches
//      the winner's value is scored after all and the pick repeated, until nobody is left.  A candidate that is never scored in a
//      round is therefore PROVEN to lose it (strictly, beyond the margin: it can neither win nor tie), which is all the reference's
//      loop (feature_selector.cpp:669-683) needs of it: ids and fValues are those of the full evaluation.  lazy_tau trades second
//      passes against scored candidates and cannot change a result.
// Measured on the bench frames (500 candidates, 150 selected, H = 10): ~40 candidates scored per round instead of 425.
constexpr int FS_SOLO_NT = 512;
static_assert(FS_SOLO_NT == FS_FRAME_MAXC, "one candidate per thread");

template <int T, int BS, int NB>
__global__ __launch_bounds__(FS_SOLO_NT) void fsel_solo_kernel(FselDev A, int32_t* sync) {
  FS_TABLES_GUARD(A);
  constexpr int NW = FS_SOLO_NT / 64, MAXC = FS_FRAME_MAXC;
  __shared__ double sC[T * T], sdpp[T];
  __shared__ double s_f[MAXC], s_u[MAXC], s_ua[MAXC], s_ue[MAXC], s_bound[MAXC], s_pr[MAXC], s_inv[T];
  __shared__ unsigned char s_alive[MAXC], s_scored[MAXC];
  __shared__ short s_list[MAXC];
  __shared__ double s_g0;
  __shared__ double s_wf[2][NW], s_wu[2][NW];
  __shared__ int s_wi[2][NW];
  // [T][MAXC]: every candidate's Delta diagonal, candidates along the lanes - what the bound estimates of every round read.  3 H = 39: 160 KB in
  // double precision, so the LDS copy is SINGLE precision there (the estimates' error bars account for it) and the exact bounds - rare - read
  // the double-precision diagonals from A.ddiag.
  using dd_t = std::conditional_t<(T > 30), float, double>;
  extern __shared__ double s_dd_raw[];
  dd_t* s_dd = reinterpret_cast<dd_t*>(s_dd_raw);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6, g = lane >> 4;
  __shared__ int s_or[2][NW];
  int orc = 0;
  // "does any thread of the workgroup say yes": one barrier (two slots: a slot is written again only after the barrier of the call between)
  auto wg_or = [&](bool v) {
    const int sl = orc++ & 1;
    const bool a = __any(v);
    if (lane == 0) s_or[sl][wv] = a ? 1 : 0;
    __syncthreads();
    int r = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) r |= s_or[sl][w];
    return r != 0;
  };
  const bool rec_lane = (lane & 15) == 0;
  const avm_fsel_batch& b = A.b;
  const int mc = b.max_cand;
  // the candidates whose thread says `mark` -> s_list (ascending) with their gain bounds beside them in s_lb; returns how many
  __shared__ int s_cnt[NW];
  __shared__ double s_lb[MAXC];
  auto build_list = [&](bool mark) {
    const unsigned long long bal = __ballot(mark);
    if (lane == 0) s_cnt[wv] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; w++) {
      const int cnt = s_cnt[w];
      base += w < wv ? cnt : 0, tot += cnt;
    }
    if (mark) {
      const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
      s_list[pos] = (short)t, s_lb[pos] = s_bound[t];
    }
    __syncthreads();
    return tot;
  };
  for (int p = blockIdx.x; p < b.n_problems; p += gridDim.x) {
    __syncthreads();
    const int nc = b.n_cand[p];
    const int kappa = max(0, b.max_features - (b.n_used ? b.n_used[p] : 0));
    const size_t pc = (size_t)p * mc;
    const double* Dp = A.delta + pc * T * T;
    const double* Dk = A.delta_pk + pc * (T * (T + 1) / 2);
    for (int idx = t; idx < T * T; idx += FS_SOLO_NT) sC[idx] = A.C[(size_t)p * T * T + idx];
    for (int idx = t; idx < T; idx += FS_SOLO_NT) sdpp[idx] = A.dpp[(size_t)p * T + idx];
    const int c = t;  // this thread's candidate
    {
      const bool ok = c < nc && A.valid[pc + min(c, mc - 1)] != 0;
      s_alive[c] = ok ? 1 : 0, s_bound[c] = HUGE_VAL, s_scored[c] = 0;
      s_pr[c] = ok ? b.cand_prob[pc + c] : 0.0;
    }
    for (int idx = t; idx < nc * T; idx += FS_SOLO_NT) {  // (one strided pass over the frame's Deltas)
      const int cc = idx / T, d = idx % T;
      const double dv = Dp[(size_t)cc * T * T + d * T + d];
      s_dd[d * MAXC + cc] = (dd_t)dv;
      if constexpr (T > 30) A.ddiag[pc * T + idx] = dv;
    }
    const double ld_nn = A.consts[(size_t)p * 4], ub_nn = A.consts[(size_t)p * 4 + 1];
    __syncthreads();
    if (wv == 0) {  // logdet of the frame's first C: the same evaluation with p = 0 - on the Delta of the first VALID candidate (the setup
                    // kernel writes delta_pk for those only: 0 x stale memory could be 0 x NaN)
      int first = -1;
      for (int c0 = 0; c0 < MAXC && first < 0; c0 += 64) {
        const unsigned long long m = __ballot(s_alive[c0 + lane] != 0);
        if (m) first = c0 + __ffsll((long long)m) - 1;
      }
      double ld0, ub0;
      const bool ok0 = fsel_logdet4<T, BS, NB, false, 2>(sC, sdpp, Dk + (size_t)max(first, 0) * (T * (T + 1) / 2), 0.0, &ld0, &ub0);
      if (lane == 0) s_g0 = ok0 ? (ld_nn + 2.0 * ld0) : __builtin_nan("");
    }
    __syncthreads();
    double G = s_g0, gprev = HUGE_VAL;
    // the EXACT Hadamard bounds of the candidates of s_list (fsel_ub4: the one function every compared bound comes from)
    auto bound_list = [&](int n) {
      for (int i0 = 0; i0 < n; i0 += NW * 4) {
        if (i0 + wv * 4 >= n) break;  // (uniform per wavefront)
        const int i = i0 + wv * 4 + g;
        const int cc = s_list[min(i, n - 1)];
        double ubt;
        if constexpr (T > 30) ubt = fsel_ub4<T, BS, NB>(sdpp, A.ddiag + (pc + cc) * T, 1, s_pr[cc]);
        else ubt = fsel_ub4<T, BS, NB>(sdpp, reinterpret_cast<const double*>(s_dd) + cc, MAXC, s_pr[cc]);
        if (i < n && rec_lane) s_u[cc] = ub_nn + ubt;
      }
    };
    // scores the candidates of s_list, four per wavefront, Delta straight from memory
    auto score_list = [&](int n) {
      for (int i0 = 0; i0 < n; i0 += NW * 4) {
        if (i0 + wv * 4 >= n) break;  // (uniform per wavefront)
        const int i = i0 + wv * 4 + g;
        const int cc = s_list[min(i, n - 1)];  // (a row without a candidate scores the list's last one again and drops the result)
        double ld, ubt;
        const bool ok = fsel_logdet4<T, BS, NB, false, 2>(sC, sdpp, Dk + (size_t)cc * (T * (T + 1) / 2), s_pr[cc], &ld, &ubt);
        if (i < n && rec_lane) {
          const double f = ok ? (ld_nn + 2.0 * ld) : __builtin_nan("");
          s_f[cc] = f, s_scored[cc] = 1;
          s_bound[cc] = ok ? f - G : HUGE_VAL;  // (a failed factorization: scored again every round)
        }
      }
      bound_list(n);
    };
    int nsel = 0;
    long long tk[6] = {0, 0, 0, 0, 0, 0}, tq[5] = {0, 0, 0, 0, 0}, tqp = 0, tkp = 0, n_scored = 0, n_second = 0, n_flag = 0, n_pass = 0;
    const bool stats = A.lazy_stats != 0 && p == 0;
#define FS_SOLO_Q(i) if (stats) { const long long n__ = clock64(); tq[i] += n__ - tqp; tqp = n__; }
    // The round's winner among the scored candidates: the lexicographic maximum of (fValue, bound, id) - feature_selector.cpp:669-683 with
    // the std::map equal-key rule of sortedlogDetUB (see fsel_pick_local): a live candidate with a higher id and a BIT-IDENTICAL bound
    // shadows the winner, scored or not.  The bounds of the unscored candidates are not computed every round.  What is: an estimate ua of
    // every live candidate's bound MINUS the part all candidates share, sum_d log1p(p Delta_dd / dpp_d), with a rigorous error bar ue
    // (a term below 0.01 by its series, remainder < x^4 / 4; above, by the single-precision logarithm, 4e-7 of the term).  Two bounds can
    // only be BIT-equal when the estimates are closer than the two error bars plus the rounding of the exact evaluation (64 T ulps of
    // the bound: 1.3e-10 on values of a few hundred at T = 30); only then the unscored candidate gets its exact bound, by the same function, to be compared.
    auto pick = [&](bool live, bool scored, double* fwin) -> int {
      constexpr int MAXSH = 8;
      int sh[MAXSH], nsh = 0;
#pragma unroll
      for (int qq = 0; qq < MAXSH; qq++) sh[qq] = -1;
      const double cf = scored ? s_f[c] : __builtin_nan("");
      if (stats) tqp = clock64();
      for (int pass = 0;; pass++) {
        const int sl = pass & 1;
        const double cu = s_u[c];  // (exact for the scored candidates and for those a previous pass has flagged)
        bool out = !scored;
#pragma unroll
        for (int qq = 0; qq < MAXSH; qq++) out |= sh[qq] == c;
        const bool in = !out && cf > -1.0;  // (NaN never wins)
        {  // the wavefront's best: three maxima in a row, each over the lanes that tie in the previous ones
          const double wf = fs_wave_max(in ? cf : -1.0);
          const bool tf = in && cf == wf;
          const double wu = fs_wave_max(tf ? cu : -DBL_MAX);
          const bool tu = tf && cu == wu;
          const int wi = fs_wave_max(tu ? c : -1);
          if (lane == 0) s_wf[sl][wv] = wf, s_wu[sl][wv] = wu, s_wi[sl][wv] = wi;
        }
        __syncthreads();
        double bf = s_wf[sl][0], bu = s_wu[sl][0];
        int bi = s_wi[sl][0];
#pragma unroll
        for (int w = 1; w < NW; w++) {
          const double f2 = s_wf[sl][w], u2 = s_wu[sl][w];
          const int i2 = s_wi[sl][w];
          if (i2 >= 0 && (bi < 0 || f2 > bf || (f2 == bf && (u2 > bu || (u2 == bu && i2 > bi))))) bf = f2, bu = u2, bi = i2;
        }
        *fwin = bf;
        FS_SOLO_Q(0)
        if (bi < 0 || A.no_key_rule || nsh >= MAXSH) return bi;  // (more than MAXSH chained collisions in one round: keep the last winner)
        const double slack = 64.0 * DBL_EPSILON * T * fmax(fabs(s_u[bi]), 1.0);  // (rounding of the two exact bounds: it grows with their size)
        const bool flag = live && !scored && c > bi && !(fabs(s_ua[c] - s_ua[bi]) > s_ue[c] + s_ue[bi] + slack);  // (an estimate that is not finite: compare the exact bounds)
        n_pass++;
        if (wg_or(flag)) {
          n_flag++;
          bound_list(build_list(flag));
          __syncthreads();
        }
        FS_SOLO_Q(1)
        const bool hit = live && c > bi && (scored || flag) && s_u[c] == bu;
        const bool anyhit = wg_or(hit);
        FS_SOLO_Q(2)
        if (!anyhit) return bi;
#pragma unroll
        for (int qq = 0; qq < MAXSH; qq++)
          if (qq == nsh) sh[qq] = bi;
        nsh++;
      }
    };
#define FS_SOLO_SEG(i) if (stats) { const long long n__ = clock64(); tk[i] += n__ - tkp; tkp = n__; }
    for (int k = 0; k < kappa; k++) {
      if (stats) tkp = clock64();
      // ---- 1. who is scored in the first pass; the estimate of every live candidate's bound
      const double th = A.lazy_tau * gprev;
      const bool live = c < nc && s_alive[c] != 0;
      if (t < T) s_inv[t] = 1.0 / sdpp[t];
      __syncthreads();
      bool mark = live && !(s_bound[c] < th);
      s_scored[c] = 0;
      int n = build_list(mark);
      // A first pass holds NW * 4 = 32 candidates (two wavefronts per SIMD: the CU's FP64 pipe is full); a 33rd costs half a pass more.  When
      // more are marked, only the 32 with the largest gain bounds are scored now - the others are exactly the ones the check of the pick
      // looks at again, and it rarely needs them (their bounds are the lowest of the marked).  Like lazy_tau: a choice of WHEN a candidate
      // is scored, never of the result.
      constexpr int CAP = NW * 4;
      if (n > CAP && gprev < HUGE_VAL) {
        const double bc = s_bound[c];
        int rank = 0;
        if (mark) {
#pragma unroll 4
          for (int j = 0; j < n; j++) {
            const int cj = s_list[j];
            const double bj = s_lb[j];
            rank += (bj > bc || (bj == bc && cj < c)) ? 1 : 0;
          }
        }
        mark = mark && rank < CAP;
        __syncthreads();  // (every reader of the first list is done)
        n = build_list(mark);
      }
      // The listed candidates' Deltas are asked for NOW (one 8-byte read per 128-byte line, a candidate per instruction: 29 lanes) and the
      // values are looked at only after the estimates below: the evaluations then find their operands in the L2 instead of waiting
      // for memory with all eight wavefronts (a scoring pass: 31 K cycles, 19 K with the operands in cache).
      constexpr int PKN = T * (T + 1) / 2, PFQ = (CAP + NW - 1) / NW, PFL = (PKN + 15) / 16;
      double pf[PFQ];
#pragma unroll
      for (int q = 0; q < PFQ; q++) {
        const int i = wv + q * NW;
        pf[q] = 0.0;
        if (n <= CAP && i < n && lane < PFL) pf[q] = Dk[(size_t)s_list[i] * PKN + min(lane * 16, PKN - 1)];
      }
      FS_SOLO_SEG(0)
      {
        double ua = 0.0, ue = 0.0;
        if (live) {
          const double prc = s_pr[c];
          const dd_t* dd = s_dd + c;
#pragma unroll
          for (int d = 0; d < T; d++) {
            const double x = (prc * dd[d * MAXC]) * s_inv[d];
            const bool small = fabs(x) <= 0.01;  // (NaN: the other branch, and the estimate is NaN - compared exactly)
            const double x2 = x * x, lg = (double)__log2f((float)(1.0 + x)) * 0.6931471805599453;
            ua += small ? x * (1.0 + x * (-0.5 + x * (1.0 / 3.0))) : lg;
            // the series' remainder is below x^4 / 4 / (1 - |x|); the other branch: 1 + x rounded to single precision (6e-8 of it) and a
            // logarithm good to two units in its last place (2.4e-7 of the result)
            ue += (small ? 0.26 * x2 * x2 + 1e-15 * fabs(x) : 1e-7 + 3e-7 * fabs(lg)) + (T > 30 ? 1.2e-7 * fabs(x) : 0.0);  // (a single-precision diagonal: 6e-8 of x)
          }
        }
        s_ua[c] = ua, s_ue[c] = ue;
      }
#pragma unroll
      for (int q = 0; q < PFQ; q++) asm volatile("" ::"v"(pf[q]));
      FS_SOLO_SEG(1)
      n_scored += n;
      // ---- 2. the scores (and the exact bounds of the scored)
      score_list(n);
      FS_SOLO_SEG(2)
      // ---- 3. the pick and its check
      int win;
      double fwin;
      for (;;) {
        __syncthreads();
        const bool scored = live && s_scored[c] != 0;
        win = pick(live, scored, &fwin);
        const double V = win >= 0 ? fwin : -1.0;  // (the reference's fMax = -1.0 when nobody has won)
        const double margin = 1e-8 * fmax(1.0, fabs(V));
        const bool need = live && !scored && !(G + s_bound[c] + margin < V);
        if (stats) tqp = clock64();
        const bool anyneed = wg_or(need);
        FS_SOLO_Q(3)
        if (!anyneed) break;
        n = build_list(need);
        FS_SOLO_SEG(3)
        n_scored += n, n_second++;
        score_list(n);
        FS_SOLO_SEG(4)
      }
      FS_SOLO_SEG(3)
      if (win < 0) break;  // lMax == -1: nothing is added; later rounds would repeat the same state
      double frun = -HUGE_VAL;
      if (A.out.min_gap) {
        // avm_fsel_out::min_gap: the winner's value minus the largest value any OTHER live candidate can have this round - its score if it
        // was scored, else its bound G + g_l, which the check above has put more than 1e-8 (relative) below the winner: exact whenever
        // the gap is smaller than that, a lower bound otherwise
        const bool scd = live && s_scored[c] != 0;
        double r2 = (live && c != win) ? (scd ? s_f[c] : G + s_bound[c]) : -HUGE_VAL;
        if (!(r2 > -1.0)) r2 = -HUGE_VAL;  // (NaN / a failed factorization never wins)
        r2 = fs_wave_max(r2);
        __syncthreads();
        if (lane == 0) s_wf[0][wv] = r2;
        __syncthreads();
        frun = s_wf[0][0];
#pragma unroll
        for (int w = 1; w < NW; w++) frun = fmax(frun, s_wf[0][w]);
        __syncthreads();
      }
      if (t == 0) {
        A.out.selected_ids[(size_t)p * b.max_features + nsel] = b.cand_id[pc + win];
        if (A.out.fvalues) A.out.fvalues[(size_t)p * b.max_features + nsel] = fwin;
        if (A.out.min_gap) A.out.min_gap[(size_t)p * b.max_features + nsel] = fwin - frun;
        A.out.n_selected[p] = nsel + 1;
        A.black[pc + win] = 1;
      }
      nsel++;
      gprev = fwin - G, G = fwin;  // the winner's value IS logdet of the next C
      const double prw = s_pr[win];
      const double* Dw = Dp + (size_t)win * T * T;
      constexpr int NFOLD = (T * T + FS_SOLO_NT - 1) / FS_SOLO_NT;  // (the thread's entries of the winner's Delta in one trip to memory)
      double dwv[NFOLD];
#pragma unroll
      for (int q = 0; q < NFOLD; q++) dwv[q] = Dw[min(t + q * FS_SOLO_NT, T * T - 1)];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < NFOLD; q++) {
        const int idx = t + q * FS_SOLO_NT;
        if (idx < T * T) {
          sC[idx] = sC[idx] + prw * dwv[q];
          if (idx / T == idx % T) sdpp[idx / T] = sdpp[idx / T] + prw * dwv[q];
        }
      }
      if (t == 0) s_alive[win] = 0;
      __syncthreads();
      FS_SOLO_SEG(5)
    }
#undef FS_SOLO_SEG
#undef FS_SOLO_Q
    if (stats && t == 0) {  // (cycles: marks + estimates, list, first-pass scores, pick + check, second-pass scores, fold; then the counters)
      long long* o = reinterpret_cast<long long*>(sync + 32);
      for (int i = 0; i < 6; i++) o[i] = tk[i];
      o[6] = n_scored, o[7] = n_second, o[8] = nsel, o[9] = n_flag, o[10] = n_pass;
      for (int i = 0; i < 4; i++) o[11 + i] = tq[i];
    }
    if (t == 0) {
      A.nsel[p] = nsel;
      __hip_atomic_fetch_add(&sync[4], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // frames finished
      // candidate evaluations this frame executed (what bench.py prices the solo form's roofline on): a 64-bit count at sync[16..17]
      __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(sync + 16), (unsigned long long)n_scored, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

__global__ __launch_bounds__(64) void fsel_live_init_kernel(FselDev A) {
  FS_TABLES_GUARD(A);
  const avm_fsel_batch& b = A.b;
  const int p = blockIdx.x, lane = threadIdx.x;
  const int nc = b.n_cand[p];
  int n = 0;
  for (int base = 0; base < nc; base += 64) {
    const int l = base + lane;
    const bool ok = l < nc && A.valid[(size_t)p * b.max_cand + l] != 0;
    const unsigned long long m = __ballot(ok);
    if (ok) {
      const int at = n + __popcll(m & ((1ull << lane) - 1));
      A.live[(size_t)p * b.max_cand + at] = l, A.pos[(size_t)p * b.max_cand + l] = at;
    }
    n += __popcll(m);
  }
  if (lane == 0) A.nlive[p] = n;
}

}  // namespace

struct FselWork {
  FselDev d;
};

size_t fsel_setup_lds_bytes(int H) {
  const int N = 9 * (H + 1);
  return sizeof(double) * ((size_t)N * N + 3 * (H + 1) * 81 + (H + 1) * 30 + N + 64) + sizeof(int) * N + 16;
}
size_t fsel_setup_lds_bytes_compact(int H) {  // the candidate slices: C_h / W and the Delta tile of four wavefronts, the camera frames
  const int T = 3 * H;
  return sizeof(double) * ((size_t)(FS_NT / 64) * (FS_CPW * (6 * H + 9) + T * T) + (H + 1) * 30) + 16;
}

// Launches setup (+ optional rounds).  All pointers in `d` are device pointers.
// frame_mode (single frames only): 0 = one launch per greedy round, 1 = fsel_frame_kernel on all XCDs, 2 = on one XCD
hipError_t launch_fsel(const avm_fsel_batch& b, const FselBuffers& w, const avm_fsel_out& out, double* omega_out, bool run_rounds,
                       int frame_mode, const int* vflag, hipStream_t stream) {
  FselDev d;
  d.b = b;
  d.vflag = vflag;
  {
    const char* nk = getenv("AVM_FSEL_NO_KEY_RULE");
    d.no_key_rule = (nk && nk[0] == '1') ? 1 : 0;
  }
  d.delta_pk = frame_mode == 3 ? w.delta_pk : nullptr;
  d.ddiag = w.ddiag;
  d.C = w.C, d.dpp = w.dpp, d.consts = w.consts, d.delta = w.delta, d.delta_u = w.delta_u, d.valid = w.valid, d.valid_u = w.valid_u;
  d.black = w.black, d.fval = w.fval, d.ub = w.ub, d.nsel = w.nsel, d.done = w.done, d.omega_out = omega_out, d.out = out;
  d.live = w.live, d.pos = w.pos, d.nlive = w.nlive;
  d.kd = w.kd;
  hipError_t e = launch_fsel_kdtree(d, stream);  // initKDTree: the setup kernel's findNNDepth walks it
  if (e != hipSuccess) return e;
  const int H = b.horizon, T = 3 * H;
  const size_t lds = fsel_setup_lds_bytes(H);
  e = hipFuncSetAttribute(reinterpret_cast<const void*>(fsel_setup_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  const int cand_per_wg = (FS_NT / 64) * FS_CPW;
  const int nslices = (b.max_cand + cand_per_wg - 1) / cand_per_wg;
  if (b.n_problems < 16) {  // few frames: one launch, slice 0 beside the candidate slices (a single frame: 1.29 ms against 1.38 in two)
    hipLaunchKernelGGL(fsel_setup_kernel, dim3(b.n_problems, 1 + nslices), dim3(FS_NT), lds, stream, d, 0, 0);
  } else {
    hipLaunchKernelGGL(fsel_setup_kernel, dim3(b.n_problems, 1), dim3(FS_NT), lds, stream, d, 0, 0);
    if (nslices > 0) hipLaunchKernelGGL(fsel_setup_kernel, dim3(b.n_problems, nslices), dim3(FS_NT), fsel_setup_lds_bytes_compact(H), stream, d, 1, 1);
  }
  if ((e = hipGetLastError()) != hipSuccess) return e;
  if (!run_rounds) return hipSuccess;
  const int per_block = FS_CPWG;  // four candidates per wavefront
  const dim3 grid((b.max_cand + per_block - 1) / per_block, b.n_problems);
  {
    const char* lt = getenv("AVM_FSEL_LAZY_TAU");  // (development: any value gives the same result, see fsel_solo_kernel)
    d.lazy_tau = lt ? atof(lt) : 0.95;
    const char* ls = getenv("AVM_FSEL_LAZY_STATS");
    d.lazy_stats = (ls && ls[0] == '1') ? 1 : 0;
  }
  if (frame_mode == 3) {  // one workgroup per frame, lazy evaluation (fsel_solo_kernel): batches of many frames
    if (b.max_cand > FS_FRAME_MAXC || T > 39) return hipErrorInvalidValue;
    if ((e = hipMemsetAsync(w.sync, 0, sizeof(int32_t) * (FS_SYNC_HDR + 64), stream)) != hipSuccess) return e;
    const size_t dl = (T > 30 ? sizeof(float) : sizeof(double)) * (size_t)FS_FRAME_MAXC * T;  // [T][512]
    static int ncu = 0;  // (one device per process: include/avm.h)
    if (ncu == 0) {
      int dev = 0, v = 0;
      ncu = (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) ? v : 256;
    }
    const int gridx = std::max(1, std::min(b.n_problems, ncu));
#define AVM_SOLO(T_, BS_, NB_)                                                                                                \
  {                                                                                                                           \
    auto kf = fsel_solo_kernel<T_, BS_, NB_>;                                                                                 \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dl)) != hipSuccess) return e; \
    hipLaunchKernelGGL(kf, dim3(gridx), dim3(FS_SOLO_NT), dl, stream, d, w.sync);                                             \
  }
    switch (T) {
      case 6: AVM_SOLO(6, 6, 1) break;
      case 9: AVM_SOLO(9, 9, 1) break;
      case 15: AVM_SOLO(15, 15, 1) break;
      case 30: AVM_SOLO(30, 15, 2) break;
      case 39: AVM_SOLO(39, 13, 3) break;
      default: return hipErrorInvalidValue;
    }
#undef AVM_SOLO
    return hipGetLastError();
  }
  if (frame_mode != 0) {  // (every frame's rounds in one launch, see fsel_frame_kernel)
    if (b.max_cand > FS_FRAME_MAXC || b.max_features >= 4096 || (frame_mode == 1 && b.n_problems != 1)) return hipErrorInvalidValue;
    // teams per XCD: 0 = one team over the whole device (a single frame), 1, or 2 when there are frames for more than eight teams
    // and two workgroups fit a compute unit's LDS (3H <= 30)
    const int tpx = frame_mode == 2 ? ((b.n_problems > 8 && T <= 30) ? 2 : 1) : 0;
    const char* td = getenv("AVM_FSEL_TEST_DROP");  // (tests: the candidate whose values never arrive -> timeout -> fallback)
    const int test_drop = td ? atoi(td) : -1;
    if ((e = hipMemsetAsync(w.sync, 0, sizeof(int32_t) * FS_SYNC_INTS, stream)) != hipSuccess) return e;
    const int ns = (int)grid.x;
#define AVM_FRAME(T_, BS_, NB_)                                                                                              \
  {                                                                                                                          \
    const size_t dl = sizeof(double) * FS_CPWG * ((T_ > 30 || tpx == 2) ? T_ * (T_ + 1) / 2 : T_ * T_);                      \
    auto kf = tpx == 2 ? (T_ <= 30 ? fsel_frame_kernel_mf<T_, BS_, NB_> : fsel_frame_kernel<T_, BS_, NB_, 1>) : tpx == 1 ? fsel_frame_kernel<T_, BS_, NB_, 1> \
                                                                                         : fsel_frame_kernel<T_, BS_, NB_, 0>; \
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(kf), hipFuncAttributeMaxDynamicSharedMemorySize, (int)dl)) != \
        hipSuccess)                                                                                                          \
      return e;                                                                                                              \
    hipLaunchKernelGGL(kf, dim3(tpx ? ns * 8 * tpx : ns), dim3(FS_NT), dl, stream, d, w.sync, ns, test_drop);                \
  }
    switch (T) {
      case 6: AVM_FRAME(6, 6, 1) break;
      case 9: AVM_FRAME(9, 9, 1) break;
      case 15: AVM_FRAME(15, 15, 1) break;
      case 30: AVM_FRAME(30, 15, 2) break;
      case 39: AVM_FRAME(39, 13, 3) break;
      default: return hipErrorInvalidValue;
    }
#undef AVM_FRAME
    return hipGetLastError();
  }
  hipLaunchKernelGGL(fsel_live_init_kernel, dim3(b.n_problems), dim3(64), 0, stream, d);
  for (int r = 0; r <= b.max_features; r++) {  // launch r: the winner of round r - 1, then the values of round r
    switch (T) {
      case 6: hipLaunchKernelGGL((fsel_round_kernel<6, 6, 1>), grid, dim3(FS_NT), 0, stream, d, r); break;
      case 9: hipLaunchKernelGGL((fsel_round_kernel<9, 9, 1>), grid, dim3(FS_NT), 0, stream, d, r); break;
      case 15: hipLaunchKernelGGL((fsel_round_kernel<15, 15, 1>), grid, dim3(FS_NT), 0, stream, d, r); break;
      case 30: hipLaunchKernelGGL((fsel_round_kernel<30, 15, 2>), grid, dim3(FS_NT), 0, stream, d, r); break;
      case 39: hipLaunchKernelGGL((fsel_round_kernel<39, 13, 3>), grid, dim3(FS_NT), 0, stream, d, r); break;
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

// ---- B4: HorizonGenerator::imu (utility/horizon_generator.cpp:25-69), one thread per frame ------------------------
__global__ __launch_bounds__(64) void fsel_horizon_imu_kernel(avm_fsel_horizon_in in, double* hor_pos, double* hor_quat) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= in.n_problems) return;
  const int H = in.horizon;
  double* pos = hor_pos + (size_t)p * (H + 1) * 3;
  double* qo = hor_quat + (size_t)p * (H + 1) * 4;
  const v3 gravity = mk3(0, 0, -9.80665);  // state_defs.h:37-41
  const v3 Ba = mk3(in.k_ba[3 * p], in.k_ba[3 * p + 1], in.k_ba[3 * p + 2]);
  const v3 a = mk3(in.acc[3 * p], in.acc[3 * p + 1], in.acc[3 * p + 2]), w = mk3(in.gyr[3 * p], in.gyr[3 * p + 1], in.gyr[3 * p + 2]);
  for (int k = 0; k < 3; k++) pos[k] = in.k_pos[3 * p + k], pos[3 + k] = in.k1_pos[3 * p + k];
  for (int k = 0; k < 4; k++) qo[k] = in.k_quat[4 * p + k], qo[4 + k] = in.k1_quat[4 * p + k];
  const double dI = in.delta_imu[p];
  const int nr = in.nr_imu[p];
  const quat Qimu = deltaQ(dI * w);  // unnormalized, and the attitude is never renormalized in the loop
  v3 pp = mk3(in.k1_pos[3 * p], in.k1_pos[3 * p + 1], in.k1_pos[3 * p + 2]), vv = mk3(in.k1_vel[3 * p], in.k1_vel[3 * p + 1], in.k1_vel[3 * p + 2]);
  quat q{in.k1_quat[4 * p + 3], in.k1_quat[4 * p], in.k1_quat[4 * p + 1], in.k1_quat[4 * p + 2]};
  for (int h = 2; h <= H; h++) {
    for (int i = 0; i < nr; i++) {
      q = qmul(q, Qimu);
      const v3 qa = qrot(q, a - Ba);
      vv = vv + dI * (gravity + qa);
      pp = pp + dI * vv + dI * (dI * (0.5 * gravity)) + dI * (dI * (0.5 * qa));
    }
    pos[3 * h] = pp.x, pos[3 * h + 1] = pp.y, pos[3 * h + 2] = pp.z;
    qo[4 * h] = q.x, qo[4 * h + 1] = q.y, qo[4 * h + 2] = q.z, qo[4 * h + 3] = q.w;
  }
}

hipError_t launch_fsel_horizon_imu(const avm_fsel_horizon_in& in, double* hor_pos, double* hor_quat, hipStream_t stream) {
  if (in.n_problems == 0) return hipSuccess;
  hipLaunchKernelGGL(fsel_horizon_imu_kernel, dim3((in.n_problems + 63) / 64), dim3(64), 0, stream, in, hor_pos, hor_quat);
  return hipGetLastError();
}

// ---- B8 (first half): the depth cloud of initKDTree() (feature_selector.cpp:396-419), one thread per window ---------
__global__ __launch_bounds__(64) void fsel_build_cloud_kernel(avm_window_batch B, const double* k1_pos, const double* k1_quat, int max_cloud,
                                                              int32_t* n_cloud, double* cloud_xy, double* cloud_depth) {
  const int w = blockIdx.x * 64 + threadIdx.x;
  if (w >= B.n_windows) return;
  const double* ex = B.ex_pose + (size_t)w * 7;
  const v3 tic = mk3(ex[0], ex[1], ex[2]);
  const quat qic{ex[6], ex[3], ex[4], ex[5]};
  double ric[9];
  q2R(qic, ric);
  const quat qk1{k1_quat[4 * w + 3], k1_quat[4 * w], k1_quat[4 * w + 1], k1_quat[4 * w + 2]};
  const v3 pk1 = mk3(k1_pos[3 * w], k1_pos[3 * w + 1], k1_pos[3 * w + 2]);
  const double* pose = B.pose + (size_t)w * NFR * 7;
  double* xy = cloud_xy + (size_t)w * max_cloud * 2;
  double* dep = cloud_depth + (size_t)w * max_cloud;
  int n = 0;
  for (int e = 0; e < B.n_feat[w] && n < max_cloud; e++) {
    const int f = B.feat_start[(size_t)w * B.max_feat + e];
    if (f > (NFR - 1) * 3.0 / 4.0) continue;
    const double est_depth = 1.0 / B.inv_depth[(size_t)w * B.max_feat + e];
    if (!(est_depth >= 0)) continue;
    double Rs[9];
    q2R(quat{pose[f * 7 + 6], pose[f * 7 + 3], pose[f * 7 + 4], pose[f * 7 + 5]}, Rs);
    const double* o = B.obs_xy + ((size_t)w * B.max_obs + B.feat_obs_begin[(size_t)w * B.max_feat + e]) * 2;
    const v3 pts_i = est_depth * mk3(o[0], o[1], 1.0);
    const v3 w_pts = Rmul(Rs, Rmul(ric, pts_i) + tic) + mk3(pose[f * 7], pose[f * 7 + 1], pose[f * 7 + 2]);
    const v3 p_IL = qrot(qinv(qk1), w_pts - pk1);
    const v3 p_CL = qrot(qinv(qic), p_IL - tic);
    xy[2 * n] = p_CL.x / p_CL.z, xy[2 * n + 1] = p_CL.y / p_CL.z, dep[n] = est_depth;
    n++;
  }
  n_cloud[w] = n;
}

// B8, second half as a parity surface: findNNDepth of every candidate, one wavefront per (frame, candidate) - the search the setup
// kernel runs inside calcInfoFromFeatures - in all three of its forms: by a whole wavefront, by a 16-lane row (the candidates' slices:
// feature_front4) and by one thread (the used features: feature_delta).  They must agree bit for bit; a disagreement is reported as NaN.
__global__ __launch_bounds__(256) void fsel_nn_depth_kernel(avm_fsel_batch b, const double* kd, double* depth_out) {
  const int p = blockIdx.y, cnd = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (cnd >= b.n_cand[p]) return;  // (wave-uniform)
  const double* xy = b.cand_xy + ((size_t)p * b.max_cand + cnd) * 2;
  const double d = kd_depth<64>(b, kd, p, xy[0], xy[1]);
  const double d16 = kd_depth<16>(b, kd, p, xy[0], xy[1]), d1 = kd_depth<1>(b, kd, p, xy[0], xy[1]);
  const bool same = __double_as_longlong(d16) == __double_as_longlong(d) && __double_as_longlong(d1) == __double_as_longlong(d);
  if ((threadIdx.x & 63) == 0) depth_out[(size_t)p * b.max_cand + cnd] = __all(same) ? d : __longlong_as_double(0x7ff8000000000000ll);
}

hipError_t launch_fsel_nn_depth(const avm_fsel_batch& b, double* kd, double* depth_out, hipStream_t stream) {
  if (b.n_problems == 0 || b.max_cand == 0) return hipSuccess;
  FselDev d{};
  d.b = b, d.kd = kd, d.vflag = nullptr;
  hipError_t e = launch_fsel_kdtree(d, stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fsel_nn_depth_kernel, dim3((b.max_cand + 3) / 4, b.n_problems), dim3(256), 0, stream, b, kd, depth_out);
  return hipGetLastError();
}
size_t fsel_kd_doubles(const avm_fsel_batch& b) { return (size_t)b.n_problems * kd_stride(b.max_cloud > 0 ? b.max_cloud : 0) + 8; }

hipError_t launch_fsel_build_cloud(const avm_window_batch& b, const double* k1_pos, const double* k1_quat, int max_cloud, int32_t* n_cloud,
                                   double* cloud_xy, double* cloud_depth, hipStream_t stream) {
  if (b.n_windows == 0) return hipSuccess;
  hipLaunchKernelGGL(fsel_build_cloud_kernel, dim3((b.n_windows + 63) / 64), dim3(64), 0, stream, b, k1_pos, k1_quat, max_cloud, n_cloud, cloud_xy,
                     cloud_depth);
  return hipGetLastError();
}

bool fsel_horizon_supported(int H) { return H == 2 || H == 3 || H == 5 || H == 10 || H == 13; }

}  // namespace avm
