// triangulate.hip — the small per-feature / per-factor kernels around the solve (also: newest-frame dead-reckoning,
// ProjectionTdFactor evaluation).  FeatureManager::triangulate (vins_estimator/src/feature_manager.cpp:202-257) on gfx950:
// linear multi-view triangulation of the features that have no depth yet, in the camera frame of their first
// observation.  One thread per (window, feature); the (2 nobs) x 4 system (nobs <= 11) lives in the thread's
// registers as four columns.  Eigen::JacobiSVD is replaced by a one-sided (Hestenes) Jacobi SVD on those columns -
// right singular vectors to working precision - and only the last right singular vector is used, through the ratio
// v[2] / v[3], so its sign does not matter.  HBM-trivial (a few hundred bytes per feature); the kernel exists so that
// the state never has to leave the device between the steps of solveOdometry().
#include "devmath.hpp"
#include "kernels.hpp"

namespace avm {

namespace {
constexpr int TRI_NT = 64;
constexpr int TRI_ROWS = 2 * NFR;  // 22
}

__global__ __launch_bounds__(TRI_NT) void triangulate_kernel(avm_window_batch B, double init_depth) {
  const long gid = (long)blockIdx.x * TRI_NT + threadIdx.x;
  const int w = (int)(gid / B.max_feat), e = (int)(gid % B.max_feat);
  if (w >= B.n_windows || e >= B.n_feat[w]) return;
  double* lam = B.inv_depth + (size_t)w * B.max_feat + e;
  if (*lam > 0.0) return;
  const int start = B.feat_start[(size_t)w * B.max_feat + e], nobs = B.feat_nobs[(size_t)w * B.max_feat + e];
  const double* obs = B.obs_xy + ((size_t)w * B.max_obs + B.feat_obs_begin[(size_t)w * B.max_feat + e]) * 2;
  const double* pose = B.pose + (size_t)w * NFR * 7;
  const double* ex = B.ex_pose + (size_t)w * 7;
  const v3 tic = mk3(ex[0], ex[1], ex[2]);
  double ric[9];
  q2R(quat{ex[6], ex[3], ex[4], ex[5]}, ric);
  auto cam = [&](int f, double* R, v3& t) {  // camera f in the world: R = Rs ric, t = Ps + Rs tic
    double Rs[9];
    q2R(quat{pose[f * 7 + 6], pose[f * 7 + 3], pose[f * 7 + 4], pose[f * 7 + 5]}, Rs);
    mat3mul(Rs, ric, R);
    t = mk3(pose[f * 7], pose[f * 7 + 1], pose[f * 7 + 2]) + Rmul(Rs, tic);
  };
  double R0[9];
  v3 t0;
  cam(start, R0, t0);
  double A[4][TRI_ROWS];
#pragma unroll
  for (int k = 0; k < NFR; k++) {
    double row0[4] = {0, 0, 0, 0}, row1[4] = {0, 0, 0, 0};
    if (k < nobs) {
      double R1[9];
      v3 t1;
      cam(start + k, R1, t1);
      // R = R0^T R1, t = R0^T (t1 - t0);  P = [R^T | -R^T t] = [R1^T R0 | -R1^T (t1 - t0)]
      const v3 d = t1 - t0;
      double P[3][4];
#pragma unroll
      for (int a = 0; a < 3; a++) {
#pragma unroll
        for (int b = 0; b < 3; b++) {
          // (R^T)(a, b) = R(b, a) = sum_m R0(m, b) R1(m, a)
          P[a][b] = R0[0 * 3 + b] * R1[0 * 3 + a] + R0[1 * 3 + b] * R1[1 * 3 + a] + R0[2 * 3 + b] * R1[2 * 3 + a];
        }
      }
      // t = R0^T d ; -R^T t
      const v3 tt = mk3(R0[0] * d.x + R0[3] * d.y + R0[6] * d.z, R0[1] * d.x + R0[4] * d.y + R0[7] * d.z, R0[2] * d.x + R0[5] * d.y + R0[8] * d.z);
#pragma unroll
      for (int a = 0; a < 3; a++) P[a][3] = -(P[a][0] * tt.x + P[a][1] * tt.y + P[a][2] * tt.z);
      const double ox = obs[2 * k], oy = obs[2 * k + 1];
      const double nrm = sqrt(ox * ox + oy * oy + 1.0);
      const double f0 = ox / nrm, f1 = oy / nrm, f2 = 1.0 / nrm;
#pragma unroll
      for (int b = 0; b < 4; b++) row0[b] = f0 * P[2][b] - f2 * P[0][b], row1[b] = f1 * P[2][b] - f2 * P[1][b];
    }
#pragma unroll
    for (int b = 0; b < 4; b++) A[b][2 * k] = row0[b], A[b][2 * k + 1] = row1[b];
  }
  // one-sided Jacobi on the four columns (rows beyond 2 nobs are zero and stay zero)
  double V[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};
  for (int sweep = 0; sweep < 60; sweep++) {
    bool rotated = false;
#pragma unroll
    for (int p = 0; p < 3; p++)
#pragma unroll
      for (int q = p + 1; q < 4; q++) {
        double app = 0, aqq = 0, apq = 0;
#pragma unroll
        for (int i = 0; i < TRI_ROWS; i++) app += A[p][i] * A[p][i], aqq += A[q][i] * A[q][i], apq += A[p][i] * A[q][i];
        if (fabs(apq) <= 1e-300 || fabs(apq) <= 2.3e-16 * sqrt(app * aqq)) continue;
        rotated = true;
        const double tau = (aqq - app) / (2.0 * apq);
        const double tn = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
        const double c = 1.0 / sqrt(1.0 + tn * tn), s = tn * c;
#pragma unroll
        for (int i = 0; i < TRI_ROWS; i++) {
          const double x = A[p][i], y = A[q][i];
          A[p][i] = c * x - s * y, A[q][i] = s * x + c * y;
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const double x = V[i][p], y = V[i][q];
          V[i][p] = c * x - s * y, V[i][q] = s * x + c * y;
        }
      }
    if (!rotated) break;
  }
  double bn = 0, v2 = 0, v3v = 0;
#pragma unroll
  for (int j = 0; j < 4; j++) {
    double s = 0;
#pragma unroll
    for (int i = 0; i < TRI_ROWS; i++) s += A[j][i] * A[j][i];
    if (j == 0 || s < bn) bn = s, v2 = V[2][j], v3v = V[3][j];
  }
  double depth = v2 / v3v;
  if (!(depth >= 0.1)) depth = init_depth;  // `estimated_depth < 0.1` -> INIT_DEPTH (a NaN ratio also falls back)
  *lam = 1.0 / depth;
}

// Estimator::processIMU dead-reckoning of the newest frame (estimator.cpp:100-107): one thread per window, sequential
// over the (<= max_samp) samples of the last interval.  Rs stays a matrix multiplied by toRotationMatrix() of the
// unnormalized deltaQ, exactly like the reference; the quaternion is formed at the end.
__global__ __launch_bounds__(TRI_NT) void imu_propagate_kernel(avm_window_batch B, double gx, double gy, double gz) {
  const int w = blockIdx.x * TRI_NT + threadIdx.x;
  if (w >= B.n_windows) return;
  const size_t iv = (size_t)w * (NFR - 1) + (NFR - 2);
  double* pose = B.pose + ((size_t)w * NFR + (NFR - 1)) * 7;
  double* sb = B.speedbias + ((size_t)w * NFR + (NFR - 1)) * 9;
  const int n = B.imu_n[iv];
  const double* dt = B.imu_dt + iv * B.max_samp;
  const double* acc = B.imu_acc + iv * (B.max_samp + 1) * 3;
  const double* gyr = B.imu_gyr + iv * (B.max_samp + 1) * 3;
  v3 P = mk3(pose[0], pose[1], pose[2]), V = mk3(sb[0], sb[1], sb[2]);
  const v3 Ba = mk3(sb[3], sb[4], sb[5]), Bg = mk3(sb[6], sb[7], sb[8]), g = mk3(gx, gy, gz);
  double R[9];
  q2R(quat{pose[6], pose[3], pose[4], pose[5]}, R);
  v3 acc0 = mk3(acc[0], acc[1], acc[2]), gyr0 = mk3(gyr[0], gyr[1], gyr[2]);
  for (int s = 0; s < n; s++) {
    const v3 a1 = mk3(acc[3 * (s + 1)], acc[3 * (s + 1) + 1], acc[3 * (s + 1) + 2]);
    const v3 w1 = mk3(gyr[3 * (s + 1)], gyr[3 * (s + 1) + 1], gyr[3 * (s + 1) + 2]);
    const double h = dt[s];
    const v3 un_acc_0 = Rmul(R, acc0 - Ba) - g;
    const v3 un_gyr = 0.5 * (gyr0 + w1) - Bg;
    double dR[9], Rn[9];
    q2R(deltaQ(h * un_gyr), dR);
    mat3mul(R, dR, Rn);
#pragma unroll
    for (int k = 0; k < 9; k++) R[k] = Rn[k];
    const v3 un_acc_1 = Rmul(R, a1 - Ba) - g;
    const v3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    P = P + h * V + (0.5 * h * h) * un_acc;
    V = V + h * un_acc;
    acc0 = a1, gyr0 = w1;
  }
  const quat q = R2q(R);
  pose[0] = P.x, pose[1] = P.y, pose[2] = P.z, pose[3] = q.x, pose[4] = q.y, pose[5] = q.z, pose[6] = q.w;
  sb[0] = V.x, sb[1] = V.y, sb[2] = V.z;
}

hipError_t launch_imu_propagate(const avm_window_batch& b, const double* g, hipStream_t stream) {
  if (b.n_windows == 0) return hipSuccess;
  hipLaunchKernelGGL(imu_propagate_kernel, dim3((b.n_windows + TRI_NT - 1) / TRI_NT), dim3(TRI_NT), 0, stream, b, g[0], g[1], g[2]);
  return hipGetLastError();
}

// ---- 8(f)2: Estimator::slideWindow (estimator.cpp:996-1107) + removeBackShiftDepth / removeBack / removeFront
// (feature_manager.cpp:275-352), in place.  One wavefront per window: the lanes move the frame / IMU arrays, lane 0 walks
// the feature list (order-preserving compaction of <= 150 entries: not worth a scan).
__global__ __launch_bounds__(64) void slide_window_kernel(avm_window_batch B, int flag, int shift_depth, double init_depth, int* err) {
  const int w = blockIdx.x, lane = threadIdx.x;
  double* pose = B.pose + (size_t)w * NFR * 7;
  double* sb = B.speedbias + (size_t)w * NFR * 9;
  int32_t* imu_n = const_cast<int32_t*>(B.imu_n) + (size_t)w * 10;
  double* dt = const_cast<double*>(B.imu_dt) + (size_t)w * 10 * B.max_samp;
  double* acc = const_cast<double*>(B.imu_acc) + (size_t)w * 10 * (B.max_samp + 1) * 3;
  double* gyr = const_cast<double*>(B.imu_gyr) + (size_t)w * 10 * (B.max_samp + 1) * 3;
  double* lba = const_cast<double*>(B.imu_lin_ba) + (size_t)w * 30;
  double* lbg = const_cast<double*>(B.imu_lin_bg) + (size_t)w * 30;
  const int SD = B.max_samp, SA = (B.max_samp + 1) * 3;
  // back_R0 / back_P0 (estimator.cpp:1001-1002)
  double back[7];
  for (int k = 0; k < 7; k++) back[k] = pose[k];
  const int n9 = imu_n[9];
  double a0[3], g0[3];  // acc_0 / gyr_0: the last sample pushed
  for (int k = 0; k < 3; k++) a0[k] = acc[9 * SA + n9 * 3 + k], g0[k] = gyr[9 * SA + n9 * 3 + k];
  __syncthreads();
  if (flag == AVM_MARGIN_OLD) {
    for (int i = 0; i < NFR - 1; i++) {
      if (lane < 7) pose[i * 7 + lane] = pose[(i + 1) * 7 + lane];
      if (lane < 9) sb[i * 9 + lane] = sb[(i + 1) * 9 + lane];
      __syncthreads();
    }
    for (int j = 0; j < 9; j++) {
      const int n = imu_n[j + 1];
      __syncthreads();
      for (int k = lane; k < n; k += 64) dt[j * SD + k] = dt[(j + 1) * SD + k];
      for (int k = lane; k < (n + 1) * 3; k += 64) acc[j * SA + k] = acc[(j + 1) * SA + k], gyr[j * SA + k] = gyr[(j + 1) * SA + k];
      if (lane < 3) lba[j * 3 + lane] = lba[(j + 1) * 3 + lane], lbg[j * 3 + lane] = lbg[(j + 1) * 3 + lane];
      if (lane == 0) imu_n[j] = n;
      __syncthreads();
    }
  } else {
    // pre_integrations[9 - 1]->push_back(...) for every sample of interval 9 (estimator.cpp:1051-1062)
    const int n8 = imu_n[8];
    if (n8 + n9 > SD) {
      if (lane == 0) *err = 1;
      return;
    }
    for (int k = lane; k < n9; k += 64) dt[8 * SD + n8 + k] = dt[9 * SD + k];
    for (int k = lane; k < n9 * 3; k += 64) acc[8 * SA + (n8 + 1) * 3 + k] = acc[9 * SA + 3 + k], gyr[8 * SA + (n8 + 1) * 3 + k] = gyr[9 * SA + 3 + k];
    if (lane < 7) pose[9 * 7 + lane] = pose[10 * 7 + lane];
    if (lane < 9) sb[9 * 9 + lane] = sb[10 * 9 + lane];
    __syncthreads();
    if (lane == 0) imu_n[8] = n8 + n9;
  }
  // the fresh pre_integrations[WINDOW_SIZE] (estimator.cpp:1027-1028, 1071-1072)
  if (lane < 3) {
    acc[9 * SA + lane] = a0[lane], gyr[9 * SA + lane] = g0[lane];
    lba[27 + lane] = sb[10 * 9 + 3 + lane], lbg[27 + lane] = sb[10 * 9 + 6 + lane];
  }
  if (lane == 0) imu_n[9] = 0;
  __syncthreads();
  if (lane != 0) return;
  // ---- feature list
  int32_t* n_feat = const_cast<int32_t*>(B.n_feat) + w;
  int32_t* fstart = const_cast<int32_t*>(B.feat_start) + (size_t)w * B.max_feat;
  int32_t* fnobs = const_cast<int32_t*>(B.feat_nobs) + (size_t)w * B.max_feat;
  int32_t* fobs = const_cast<int32_t*>(B.feat_obs_begin) + (size_t)w * B.max_feat;
  double* obs = const_cast<double*>(B.obs_xy) + (size_t)w * B.max_obs * 2;
  double* lam = B.inv_depth + (size_t)w * B.max_feat;
  const double* ex = B.ex_pose + (size_t)w * 7;
  double ric[9], Rb[9], Rn[9];
  q2R(quat{ex[6], ex[3], ex[4], ex[5]}, ric);
  const v3 tic = mk3(ex[0], ex[1], ex[2]);
  q2R(quat{back[6], back[3], back[4], back[5]}, Rb);
  q2R(quat{pose[6], pose[3], pose[4], pose[5]}, Rn);
  // R0 = back_R0 ric, P0 = back_P0 + back_R0 tic; R1 = Rs[0] ric, P1 = Ps[0] + Rs[0] tic   (estimator.cpp:1098-1103)
  const v3 P0 = mk3(back[0], back[1], back[2]) + Rmul(Rb, tic), P1 = mk3(pose[0], pose[1], pose[2]) + Rmul(Rn, tic);
  const int nf = *n_feat;
  int o = 0;
  for (int e = 0; e < nf; e++) {
    int st = fstart[e], no = fnobs[e], ob = fobs[e];
    double l = lam[e];
    bool keep = true;
    if (flag == AVM_MARGIN_OLD) {
      if (st != 0) {
        st--;
      } else {
        const double ux = obs[2 * ob], uy = obs[2 * ob + 1];
        ob++, no--;
        if (shift_depth) {
          if (no < 2) {
            keep = false;
          } else {
            const double depth = 1.0 / l;
            const v3 pts_i = depth * mk3(ux, uy, 1.0);
            const v3 w_pts = Rmul(Rb, Rmul(ric, pts_i)) + P0;
            const v3 pts_j = RTmul(ric, RTmul(Rn, w_pts - P1));
            l = 1.0 / (pts_j.z > 0 ? pts_j.z : init_depth);
          }
        } else if (no == 0) {
          keep = false;
        }
      }
    } else {
      if (st == NFR - 1) {
        st--;
      } else if (st + no - 1 >= NFR - 2) {  // endFrame() >= frame_count - 1: it has an observation in frame 9
        const int j = NFR - 2 - st;
        for (int k = j; k + 1 < no; k++) obs[2 * (ob + k)] = obs[2 * (ob + k + 1)], obs[2 * (ob + k) + 1] = obs[2 * (ob + k + 1) + 1];
        no--;
        if (no == 0) keep = false;
      }
    }
    if (keep) fstart[o] = st, fnobs[o] = no, fobs[o] = ob, lam[o] = l, o++;
  }
  *n_feat = o;
}

hipError_t launch_slide_window(const avm_window_batch& b, int flag, int shift_depth, double init_depth, int* err, hipStream_t stream) {
  if (b.n_windows == 0) return hipSuccess;
  hipLaunchKernelGGL(slide_window_kernel, dim3(b.n_windows), dim3(64), 0, stream, b, flag, shift_depth, init_depth, err);
  return hipGetLastError();
}

// ---- table validation: one wavefront per window (one thread per selector frame), the lowest failing index wins ------------------------------
__global__ __launch_bounds__(64) void validate_windows_kernel(avm_window_batch B, int what, int* first_bad) {
  const int w = blockIdx.x, lane = threadIdx.x;  // one wavefront per window, lane = feature (mod 64)
  int rule = check_window_tables(B, w, what, lane, 64);
  rule = rule ? rule : 1 << 30;
  for (int o = 32; o > 0; o >>= 1) rule = min(rule, __shfl_xor(rule, o, 64));
  if (lane == 0 && rule != 1 << 30) atomicMin(first_bad, w * 8 + rule);
  // (only for a window whose prior tables passed: the count below indexes with them)
  if (lane == 0 && rule == 1 << 30 && (what & CHK_PRIOR)) {
    const int m = window_prior_tp_misfit(B, w);
    if (m) atomicOr(first_bad + 1, m);
  }
}

__global__ __launch_bounds__(64) void validate_fsel_kernel(avm_fsel_batch b, int* first_bad) {
  const int p = blockIdx.x * 64 + threadIdx.x;
  if (p >= b.n_problems) return;
  const int rule = check_fsel_tables(b, p);
  if (rule) atomicMin(first_bad, p * 8 + rule);
}

hipError_t launch_validate_windows(const avm_window_batch& b, int what, int* first_bad, hipStream_t stream) {
  hipLaunchKernelGGL(validate_windows_kernel, dim3(b.n_windows), dim3(64), 0, stream, b, what, first_bad);
  return hipGetLastError();
}

hipError_t launch_validate_fsel(const avm_fsel_batch& b, int* first_bad, hipStream_t stream) {
  hipLaunchKernelGGL(validate_fsel_kernel, dim3((b.n_problems + 63) / 64), dim3(64), 0, stream, b, first_bad);
  return hipGetLastError();
}

// A7: ProjectionTdFactor::Evaluate (factor/projection_td_factor.cpp:34-141), one thread per factor.
__global__ __launch_bounds__(TRI_NT) void projection_td_eval_kernel(avm_td_factor_batch f, double* residual, double* jac) {
  const int i = blockIdx.x * TRI_NT + threadIdx.x;
  if (i >= f.n) return;
  const double* pi = f.pose_i + 7 * (size_t)i;
  const double* pj = f.pose_j + 7 * (size_t)i;
  const double* ex = f.ex_pose + 7 * (size_t)i;
  const v3 Pi = mk3(pi[0], pi[1], pi[2]), Pj = mk3(pj[0], pj[1], pj[2]), tic = mk3(ex[0], ex[1], ex[2]);
  const quat Qi{pi[6], pi[3], pi[4], pi[5]}, Qj{pj[6], pj[3], pj[4], pj[5]}, qic{ex[6], ex[3], ex[4], ex[5]};
  const double lam = f.inv_depth[i], td = f.td[i], s = f.focal_length / 1.5;
  const v3 pts_i = mk3(f.pts_i[2 * i], f.pts_i[2 * i + 1], 1.0), pts_j = mk3(f.pts_j[2 * i], f.pts_j[2 * i + 1], 1.0);
  const v3 vel_i = mk3(f.vel_i[2 * i], f.vel_i[2 * i + 1], 0.0), vel_j = mk3(f.vel_j[2 * i], f.vel_j[2 * i + 1], 0.0);
  const double row_i = f.row_i[i] - f.row / 2, row_j = f.row_j[i] - f.row / 2;
  const v3 pts_i_td = pts_i - (td - f.td_i[i] + f.tr / f.row * row_i) * vel_i;
  const v3 pts_j_td = pts_j - (td - f.td_j[i] + f.tr / f.row * row_j) * vel_j;
  const v3 pts_camera_i = (1.0 / lam) * pts_i_td;
  const v3 pts_imu_i = qrot(qic, pts_camera_i) + tic;
  const v3 pts_w = qrot(Qi, pts_imu_i) + Pi;
  const v3 pts_imu_j = qrot(qinv(Qj), pts_w - Pj);
  const v3 pts_camera_j = qrot(qinv(qic), pts_imu_j - tic);
  const double dep_j = pts_camera_j.z;
  residual[2 * (size_t)i] = s * ((pts_camera_j.x / dep_j) - pts_j_td.x);
  residual[2 * (size_t)i + 1] = s * ((pts_camera_j.y / dep_j) - pts_j_td.y);
  if (!jac) return;
  double* J = jac + 40 * (size_t)i;
  double Ri[9], Rj[9], ric[9];
  q2R(Qi, Ri), q2R(Qj, Rj), q2R(qic, ric);
  const double red[2][3] = {{s / dep_j, 0.0, s * (-pts_camera_j.x / (dep_j * dep_j))}, {0.0, s / dep_j, s * (-pts_camera_j.y / (dep_j * dep_j))}};
  // reduce * M for a 3x3 M (row-major) -> J(:, col0 .. col0+2)
  auto put = [&](const double* M, int col0) {
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
      for (int c = 0; c < 3; c++) J[r * 20 + col0 + c] = red[r][0] * M[c] + red[r][1] * M[3 + c] + red[r][2] * M[6 + c];
  };
  double ricT[9], RjT[9], A[9], B[9], T[9], S[9], neg[9];
#pragma unroll
  for (int r = 0; r < 3; r++)
#pragma unroll
    for (int c = 0; c < 3; c++) ricT[r * 3 + c] = ric[c * 3 + r], RjT[r * 3 + c] = Rj[c * 3 + r];
  // pose_i : [ric^T Rj^T | ric^T Rj^T Ri (-[pts_imu_i]x)]
  mat3mul(ricT, RjT, A);
  put(A, 0);
  mat3mul(A, Ri, T);
  skew9(pts_imu_i, S);
#pragma unroll
  for (int k = 0; k < 9; k++) neg[k] = -S[k];
  mat3mul(T, neg, B);
  put(B, 3);
  // pose_j : [-ric^T Rj^T | ric^T [pts_imu_j]x]
#pragma unroll
  for (int k = 0; k < 9; k++) neg[k] = -A[k];
  put(neg, 6);
  skew9(pts_imu_j, S);
  mat3mul(ricT, S, B);
  put(B, 9);
  // ex_pose
  {
    double RjTRi[9], M[9], tmp_r[9];
    mat3mul(RjT, Ri, RjTRi);
#pragma unroll
    for (int k = 0; k < 9; k++) M[k] = RjTRi[k] - ((k % 4 == 0) ? 1.0 : 0.0);
    mat3mul(ricT, M, B);
    put(B, 12);
    mat3mul(T, ric, tmp_r);  // ric^T Rj^T Ri ric
    double S1[9], S2[9], S3[9], P1[9];
    skew9(pts_camera_i, S1);
    mat3mul(tmp_r, S1, P1);
    skew9(Rmul(tmp_r, pts_camera_i), S2);
    const v3 inner = Rmul(RjT, Rmul(Ri, tic) + Pi - Pj) - tic;
    skew9(Rmul(ricT, inner), S3);
#pragma unroll
    for (int k = 0; k < 9; k++) B[k] = -P1[k] + S2[k] + S3[k];
    put(B, 15);
    // inverse depth and td
    const v3 vf = Rmul(tmp_r, pts_i_td), vt = Rmul(tmp_r, vel_i);
#pragma unroll
    for (int r = 0; r < 2; r++) {
      J[r * 20 + 18] = (red[r][0] * vf.x + red[r][1] * vf.y + red[r][2] * vf.z) * -1.0 / (lam * lam);
      J[r * 20 + 19] = (red[r][0] * vt.x + red[r][1] * vt.y + red[r][2] * vt.z) / lam * -1.0 + s * (r == 0 ? vel_j.x : vel_j.y);
    }
  }
}

hipError_t launch_projection_td_eval(const avm_td_factor_batch& f, double* residual, double* jac, hipStream_t stream) {
  if (f.n == 0) return hipSuccess;
  hipLaunchKernelGGL(projection_td_eval_kernel, dim3((f.n + TRI_NT - 1) / TRI_NT), dim3(TRI_NT), 0, stream, f, residual, jac);
  return hipGetLastError();
}

hipError_t launch_triangulate(const avm_window_batch& b, double init_depth, hipStream_t stream) {
  const long n = (long)b.n_windows * b.max_feat;
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(triangulate_kernel, dim3((unsigned)((n + TRI_NT - 1) / TRI_NT)), dim3(TRI_NT), 0, stream, b, init_depth);
  return hipGetLastError();
}

}  // namespace avm
