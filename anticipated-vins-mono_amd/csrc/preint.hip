// preint.hip — IntegrationBase on gfx950: midpoint pre-integration of raw IMU samples and
// the IMU factor's constant sqrt_info.
//   reference: vins_estimator/src/factor/integration_base.h:13-28 (ctor/noise), :54-128
//   (midPointIntegration: F, V, jacobian = F*jacobian, covariance = F P F^T + V Q V^T),
//   :130-158 (propagate), and imu_factor.h:64 (sqrt_info = LLT(cov^-1).matrixL()^T, which the
//   reference recomputes on every Evaluate; it is constant during a solve, so hoisted here).
// Mapping: one 64-lane wavefront per (window, interval); the 15x15 state lives in LDS, the
// 225 outputs of each 15x15 product are spread 4 per lane.  4 intervals per 256-thread block.
#include "devmath.hpp"
#include "kernels.hpp"

namespace avm {

namespace {
constexpr int PW = 4;  // waves per block

// every wavefront integrates its own interval: only wave-level ordering of its LDS traffic is needed
AVM_DEV void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_sched_barrier(0);  // keep the phases of a sample from being interleaved (register pressure)
}

struct PreLds {
  double Fi[256];  // F, 16 x 16 image (row/column 15 = 0); the sqrt_info phase reuses it as its 15 x 15 work matrix
  double Vi[320];  // V, 16 x 20 image (row 15, columns 18-19 = 0); reused for the inverse in the sqrt_info phase
  double m[72];    // Rd, Rr, Ra0, Ra1, IRw (I - Rw*dt), T1=Rd*Ra0, T2=Rr*Ra1, T3=T2*IRw
  double piv[16];
};
typedef double d4 __attribute__((ext_vector_type(4)));
}  // namespace

// The 15 x 15 state matrices never leave registers: with v_mfma_f64_16x16x4 the accumulator layout
// (lane l, register r) = M[(l >> 4) + 4 r][l & 15] is also the layout of a B operand (k = (l >> 4) + 4 m), so
//   jacobian   <- F * jacobian                      4 MFMAs, the result is the next B operand
//   covariance <- F * (P * F^T) + V * (Q V^T)       4 + 4 + 5 MFMAs; P is symmetric, so its accumulator registers
//                                                   double as the A operand P[l & 15][(l >> 4) + 4 m], and the
//                                                   A-layout registers of F (of V) double as the B operand F^T (V^T)
// F and V are rebuilt per sample as small LDS images (only their sample-dependent 3x3 blocks are rewritten) and
// each lane fetches its 4 + 5 operand entries from there.
__global__ __launch_bounds__(64 * PW) __attribute__((amdgpu_waves_per_eu(4, 8))) void preint_kernel(PreintArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  PreLds* all = reinterpret_cast<PreLds*>(smem_raw);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  PreLds& L = all[wv];
  const long iv = (long)blockIdx.x * PW + wv;  // interval index = w*10 + j
  if (iv >= (long)a.n_windows * 10) return;    // no block-level barriers below
  const int ns = a.imu_n[iv];

  const double* acc = a.imu_acc + iv * (a.max_samp + 1) * 3;
  const double* gyr = a.imu_gyr + iv * (a.max_samp + 1) * 3;
  const double* dts = a.imu_dt + iv * a.max_samp;
  const v3 lba = mk3(a.imu_lin_ba[iv * 3], a.imu_lin_ba[iv * 3 + 1], a.imu_lin_ba[iv * 3 + 2]);
  const v3 lbg = mk3(a.imu_lin_bg[iv * 3], a.imu_lin_bg[iv * 3 + 1], a.imu_lin_bg[iv * 3 + 2]);

  const int li = lane & 15, lk = lane >> 4;
  // static part of the images: zeros, the identity blocks of F
  for (int i = lane; i < 256; i += 64) L.Fi[i] = (i / 16 == i % 16 && i / 16 < 15) ? 1.0 : 0.0;
  for (int i = lane; i < 320; i += 64) L.Vi[i] = 0.0;
  d4 Jb = {0, 0, 0, 0}, Pb = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; r++) Jb[r] = (lk + 4 * r == li && li < 15) ? 1.0 : 0.0;
  // noise variance of V's column k = lk + 4 m (integration_base.h:21-27)
  double qn[5];
#pragma unroll
  for (int m = 0; m < 5; m++) {
    const int k = lk + 4 * m;
    qn[m] = k >= 18 ? 0.0 : (k < 3 || (k >= 6 && k < 9)) ? a.acc_n * a.acc_n : (k < 12 ? a.gyr_n * a.gyr_n : (k < 15 ? a.acc_w * a.acc_w : a.gyr_w * a.gyr_w));
  }
  // wave-uniform running state (every lane holds a copy)
  v3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
  quat dq{1, 0, 0, 0};
  v3 acc0 = mk3(acc[0], acc[1], acc[2]), gyr0 = mk3(gyr[0], gyr[1], gyr[2]);
  double sum_dt = 0;
  wsync();

  for (int s = 0; s < ns; s++) {
    const double dt = dts[s];
    const v3 acc1 = mk3(acc[3 * (s + 1)], acc[3 * (s + 1) + 1], acc[3 * (s + 1) + 2]);
    const v3 gyr1 = mk3(gyr[3 * (s + 1)], gyr[3 * (s + 1) + 1], gyr[3 * (s + 1) + 2]);
    // integration_base.h:63-69
    v3 un_acc_0 = qrot(dq, acc0 - lba);
    v3 un_gyr = 0.5 * (gyr0 + gyr1) - lbg;
    quat rq = qmul(dq, quat{1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2});
    v3 un_acc_1 = qrot(rq, acc1 - lba);
    v3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    v3 rp = dp + dt * dv + (0.5 * dt * dt) * un_acc;
    v3 rv = dv + dt * un_acc;
    if (lane == 0) {
      double Ra0[9], Ra1[9], Rw[9];
      q2R(dq, &L.m[0]);
      q2R(rq, &L.m[9]);
      skew9(acc0 - lba, Ra0);
      skew9(acc1 - lba, Ra1);
      skew9(un_gyr, Rw);
      for (int i = 0; i < 9; i++) {
        L.m[18 + i] = Ra0[i];
        L.m[27 + i] = Ra1[i];
        L.m[36 + i] = ((i % 4 == 0) ? 1.0 : 0.0) - Rw[i] * dt;
      }
    }
    wsync();
    if (lane < 18) {  // T1 = Rd * R_a_0_x (lanes 0-8), T2 = Rr * R_a_1_x (lanes 9-17)
      const int e = lane % 9, r = e / 3, c = e % 3, o = lane < 9 ? 0 : 9;
      L.m[45 + lane] = L.m[o + 3 * r] * L.m[18 + o + c] + L.m[o + 3 * r + 1] * L.m[18 + o + 3 + c] + L.m[o + 3 * r + 2] * L.m[18 + o + 6 + c];
    }
    wsync();
    if (lane < 9) {  // T3 = T2 * (I - R_w_x dt)
      const int r = lane / 3, c = lane % 3;
      L.m[63 + lane] = L.m[54 + 3 * r] * L.m[36 + c] + L.m[54 + 3 * r + 1] * L.m[36 + 3 + c] + L.m[54 + 3 * r + 2] * L.m[36 + 6 + c];
    }
    dp = rp;
    dv = rv;
    dq = qnormalized(rq);  // integration_base.h:153
    sum_dt += dt;
    acc0 = acc1;
    gyr0 = gyr1;
    wsync();
    if (lane < 9) {
      const int r = lane / 3, c = lane % 3;
      const double Rd = L.m[lane], Rr = L.m[9 + lane], T1 = L.m[45 + lane], T2 = L.m[54 + lane], T3 = L.m[63 + lane];
      const double I = (r == c) ? 1.0 : 0.0;
      const double dt2 = dt * dt;
      // F (integration_base.h:90-105)
      L.Fi[(0 + r) * 16 + 3 + c] = -0.25 * T1 * dt2 + -0.25 * T3 * dt2;
      L.Fi[(0 + r) * 16 + 6 + c] = I * dt;
      L.Fi[(0 + r) * 16 + 9 + c] = -0.25 * (Rd + Rr) * dt2;
      L.Fi[(0 + r) * 16 + 12 + c] = -0.25 * T2 * dt2 * -dt;
      L.Fi[(3 + r) * 16 + 3 + c] = L.m[36 + lane];
      L.Fi[(3 + r) * 16 + 12 + c] = -1.0 * I * dt;
      L.Fi[(6 + r) * 16 + 3 + c] = -0.5 * T1 * dt + -0.5 * T3 * dt;
      L.Fi[(6 + r) * 16 + 9 + c] = -0.5 * (Rd + Rr) * dt;
      L.Fi[(6 + r) * 16 + 12 + c] = -0.5 * T2 * dt * -dt;
      // V (integration_base.h:108-120)
      const double v03 = 0.25 * -T2 * dt2 * 0.5 * dt;
      const double v63 = 0.5 * -T2 * dt * 0.5 * dt;
      L.Vi[(0 + r) * 20 + 0 + c] = 0.25 * Rd * dt2;
      L.Vi[(0 + r) * 20 + 3 + c] = v03;
      L.Vi[(0 + r) * 20 + 6 + c] = 0.25 * Rr * dt2;
      L.Vi[(0 + r) * 20 + 9 + c] = v03;
      L.Vi[(3 + r) * 20 + 3 + c] = 0.5 * I * dt;
      L.Vi[(3 + r) * 20 + 9 + c] = 0.5 * I * dt;
      L.Vi[(6 + r) * 20 + 0 + c] = 0.5 * Rd * dt;
      L.Vi[(6 + r) * 20 + 3 + c] = v63;
      L.Vi[(6 + r) * 20 + 6 + c] = 0.5 * Rr * dt;
      L.Vi[(6 + r) * 20 + 9 + c] = v63;
      L.Vi[(9 + r) * 20 + 12 + c] = I * dt;
      L.Vi[(12 + r) * 20 + 15 + c] = I * dt;
    }
    wsync();
    double fa[4], va[5];
#pragma unroll
    for (int m = 0; m < 4; m++) fa[m] = L.Fi[li * 16 + lk + 4 * m];
#pragma unroll
    for (int m = 0; m < 5; m++) va[m] = L.Vi[li * 20 + lk + 4 * m];
    // jacobian = F * jacobian
    d4 Jn = {0, 0, 0, 0}, Z = {0, 0, 0, 0}, Pn = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < 4; m++) Jn = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], Jb[m], Jn, 0, 0, 0);
    // Z = P * F^T ; covariance = F * Z + V * (Q V^T)
#pragma unroll
    for (int m = 0; m < 4; m++) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(Pb[m], fa[m], Z, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 4; m++) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], Z[m], Pn, 0, 0, 0);
#pragma unroll
    for (int m = 0; m < 5; m++) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(va[m], va[m] * qn[m], Pn, 0, 0, 0);
    Jb = Jn;
    Pb = Pn;
    wsync();  // the images are rewritten by the next sample
  }

  // covariance to LDS for the sqrt_info phase (15 x 15, row-major)
  double* LA = L.Fi;
  double* LI = L.Vi;
  wsync();
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = lk + 4 * r;
    if (row < 15 && li < 15) LA[row * 15 + li] = Pb[r];
  }
  for (int i = lane; i < 225; i += 64) LI[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
  wsync();
  // ---- sqrt_info = LLT(P^-1).matrixL()^T : partial-pivot Gauss-Jordan then Cholesky -------
  for (int k = 0; k < 15; k++) {
    if (lane == 0) {
      int p = k;
      double best = fabs(LA[k * 15 + k]);
      for (int i = k + 1; i < 15; i++)
        if (fabs(LA[i * 15 + k]) > best) best = fabs(LA[i * 15 + k]), p = i;
      L.piv[0] = (double)p;
    }
    wsync();
    const int p = (int)L.piv[0];
    if (p != k && lane < 15) {
      double t = LA[k * 15 + lane];
      LA[k * 15 + lane] = LA[p * 15 + lane];
      LA[p * 15 + lane] = t;
      t = LI[k * 15 + lane];
      LI[k * 15 + lane] = LI[p * 15 + lane];
      LI[p * 15 + lane] = t;
    }
    wsync();
    const double piv = LA[k * 15 + k];
    double fa[4], fi[4];
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      if (i < 225) {
        const int r = i / 15, c = i % 15;
        const double f = LA[r * 15 + k] / piv;
        fa[q] = (r > k && c >= k) ? LA[i] - f * LA[k * 15 + c] : LA[i];
        fi[q] = (r > k) ? LI[i] - f * LI[k * 15 + c] : LI[i];
      }
    }
    wsync();
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      if (i < 225) LA[i] = fa[q], LI[i] = fi[q];
    }
    wsync();
  }
  for (int k = 14; k >= 0; k--) {
    const double piv = LA[k * 15 + k];
    if (lane < 15) LI[k * 15 + lane] = LI[k * 15 + lane] / piv;
    wsync();
    double fi[4];
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      if (i < 225) {
        const int r = i / 15, c = i % 15;
        fi[q] = (r < k) ? LI[i] - LA[r * 15 + k] * LI[k * 15 + c] : LI[i];
      }
    }
    wsync();
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      if (i < 225) LI[i] = fi[q];
    }
    wsync();
  }
  // lower Cholesky of Inv (column algorithm, same operation order as the oracle's llt_lower), one lane per row
  for (int k = 0; k < 15; k++) {
    if (lane == 0) {
      double x = LI[k * 15 + k];
      for (int j = 0; j < k; j++) x -= LI[k * 15 + j] * LI[k * 15 + j];
      LI[k * 15 + k] = sqrt(x);
    }
    wsync();
    if (lane > k && lane < 15) {
      double sacc = LI[lane * 15 + k];
      for (int j = 0; j < k; j++) sacc -= LI[lane * 15 + j] * LI[k * 15 + j];
      LI[lane * 15 + k] = sacc / LI[k * 15 + k];
    }
    wsync();
  }
  double* od = a.out_delta + iv * 10;
  if (lane == 0) {
    od[0] = dp.x, od[1] = dp.y, od[2] = dp.z;
    od[3] = dq.x, od[4] = dq.y, od[5] = dq.z, od[6] = dq.w;
    od[7] = dv.x, od[8] = dv.y, od[9] = dv.z;
    a.out_sum_dt[iv] = sum_dt;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = lk + 4 * r;
    if (row < 15 && li < 15) {
      a.out_jacobian[iv * 225 + row * 15 + li] = Jb[r];
      a.out_covariance[iv * 225 + row * 15 + li] = Pb[r];
    }
  }
  for (int i = lane; i < 225; i += 64) {
    const int r = i / 15, c = i % 15;
    a.out_sqrt_info[iv * 225 + i] = (c >= r) ? LI[c * 15 + r] : 0.0;  // U = L^T
  }
}

void launch_preint(const PreintArgs& a, hipStream_t stream) {
  const long n_iv = (long)a.n_windows * 10;
  const int blocks = (int)((n_iv + PW - 1) / PW);
  hipLaunchKernelGGL(preint_kernel, dim3(blocks), dim3(64 * PW), sizeof(PreLds) * PW, stream, a);
}

}  // namespace avm
