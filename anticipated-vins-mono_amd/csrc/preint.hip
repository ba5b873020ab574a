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
}

struct PreLds {
  double J[225], P[225], F[225], T[225], V[270];
  double m[72];  // Rd, Rr, Ra0, Ra1, IRw (I - Rw*dt), T1=Rd*Ra0, T2=Rr*Ra1, T3=T2*IRw
  double piv[16];  // (the Gauss-Jordan work matrices of the sqrt_info phase alias F and T, which are dead by then)
};
}  // namespace

__global__ __launch_bounds__(64 * PW) void preint_kernel(PreintArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  PreLds* all = reinterpret_cast<PreLds*>(smem_raw);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  PreLds& L = all[wv];
  const long iv = (long)blockIdx.x * PW + wv;  // interval index = w*10 + j
  const bool live = iv < (long)a.n_windows * 10;
  const long ivc = live ? iv : 0;
  const int ns = live ? a.imu_n[ivc] : 0;
  if (!live) return;  // no block-level barriers below
  const int nmax = ns;

  const double* acc = a.imu_acc + ivc * (a.max_samp + 1) * 3;
  const double* gyr = a.imu_gyr + ivc * (a.max_samp + 1) * 3;
  const double* dts = a.imu_dt + ivc * a.max_samp;
  const v3 lba = mk3(a.imu_lin_ba[ivc * 3], a.imu_lin_ba[ivc * 3 + 1], a.imu_lin_ba[ivc * 3 + 2]);
  const v3 lbg = mk3(a.imu_lin_bg[ivc * 3], a.imu_lin_bg[ivc * 3 + 1], a.imu_lin_bg[ivc * 3 + 2]);

  for (int i = lane; i < 225; i += 64) {
    L.J[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
    L.P[i] = 0.0;
  }
  // wave-uniform running state (every lane holds a copy)
  v3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
  quat dq{1, 0, 0, 0};
  v3 acc0 = mk3(acc[0], acc[1], acc[2]), gyr0 = mk3(gyr[0], gyr[1], gyr[2]);
  double sum_dt = 0;
  const double an2 = a.acc_n * a.acc_n, gn2 = a.gyr_n * a.gyr_n, aw2 = a.acc_w * a.acc_w, gw2 = a.gyr_w * a.gyr_w;
  wsync();

  for (int s = 0; s < nmax; s++) {
    const bool act = s < ns;
    double dt = 0;
    if (act) {
      dt = dts[s];
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
        mat3mul(&L.m[0], Ra0, &L.m[45]);          // T1 = Rd * R_a_0_x
        mat3mul(&L.m[9], Ra1, &L.m[54]);          // T2 = Rr * R_a_1_x
        mat3mul(&L.m[54], &L.m[36], &L.m[63]);    // T3 = T2 * (I - R_w_x dt)
      }
      dp = rp;
      dv = rv;
      dq = qnormalized(rq);  // integration_base.h:153
      sum_dt += dt;
      acc0 = acc1;
      gyr0 = gyr1;
    }
    for (int i = lane; i < 225; i += 64) L.F[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
    for (int i = lane; i < 270; i += 64) L.V[i] = 0.0;
    wsync();
    if (act && lane < 9) {
      const int r = lane / 3, c = lane % 3;
      const double Rd = L.m[lane], Rr = L.m[9 + lane], T1 = L.m[45 + lane], T2 = L.m[54 + lane], T3 = L.m[63 + lane];
      const double I = (r == c) ? 1.0 : 0.0;
      const double dt2 = dt * dt;
      // F (integration_base.h:90-105)
      L.F[(0 + r) * 15 + 3 + c] = -0.25 * T1 * dt2 + -0.25 * T3 * dt2;
      L.F[(0 + r) * 15 + 6 + c] = I * dt;
      L.F[(0 + r) * 15 + 9 + c] = -0.25 * (Rd + Rr) * dt2;
      L.F[(0 + r) * 15 + 12 + c] = -0.25 * T2 * dt2 * -dt;
      L.F[(3 + r) * 15 + 3 + c] = L.m[36 + lane];
      L.F[(3 + r) * 15 + 12 + c] = -1.0 * I * dt;
      L.F[(6 + r) * 15 + 3 + c] = -0.5 * T1 * dt + -0.5 * T3 * dt;
      L.F[(6 + r) * 15 + 9 + c] = -0.5 * (Rd + Rr) * dt;
      L.F[(6 + r) * 15 + 12 + c] = -0.5 * T2 * dt * -dt;
      // V (integration_base.h:108-120)
      const double v03 = 0.25 * -T2 * dt2 * 0.5 * dt;
      const double v63 = 0.5 * -T2 * dt * 0.5 * dt;
      L.V[(0 + r) * 18 + 0 + c] = 0.25 * Rd * dt2;
      L.V[(0 + r) * 18 + 3 + c] = v03;
      L.V[(0 + r) * 18 + 6 + c] = 0.25 * Rr * dt2;
      L.V[(0 + r) * 18 + 9 + c] = v03;
      L.V[(3 + r) * 18 + 3 + c] = 0.5 * I * dt;
      L.V[(3 + r) * 18 + 9 + c] = 0.5 * I * dt;
      L.V[(6 + r) * 18 + 0 + c] = 0.5 * Rd * dt;
      L.V[(6 + r) * 18 + 3 + c] = v63;
      L.V[(6 + r) * 18 + 6 + c] = 0.5 * Rr * dt;
      L.V[(6 + r) * 18 + 9 + c] = v63;
      L.V[(9 + r) * 18 + 12 + c] = I * dt;
      L.V[(12 + r) * 18 + 15 + c] = I * dt;
    }
    wsync();
    // T = F*J ; then J = T.  (jacobian = F * jacobian)
    double o[4];
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      o[q] = 0;
      if (i < 225) {
        const int r = i / 15, c = i % 15;
        double sacc = 0;
        for (int k = 0; k < 15; k++) sacc += L.F[r * 15 + k] * L.J[k * 15 + c];
        o[q] = sacc;
      }
    }
    wsync();
    if (act)
      for (int q = 0; q < 4; q++) {
        const int i = lane + 64 * q;
        if (i < 225) L.J[i] = o[q];
      }
    // T = F*P
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      if (i < 225) {
        const int r = i / 15, c = i % 15;
        double sacc = 0;
        for (int k = 0; k < 15; k++) sacc += L.F[r * 15 + k] * L.P[k * 15 + c];
        L.T[i] = sacc;
      }
    }
    wsync();
    // P = T*F^T + V*Q*V^T
    for (int q = 0; q < 4; q++) {
      const int i = lane + 64 * q;
      if (i < 225 && act) {
        const int r = i / 15, c = i % 15;
        double s1 = 0;
        for (int k = 0; k < 15; k++) s1 += L.T[r * 15 + k] * L.F[c * 15 + k];
        double s2 = 0;
        for (int k = 0; k < 18; k++) {
          const double nk = (k < 3) ? an2 : (k < 6) ? gn2 : (k < 9) ? an2 : (k < 12) ? gn2 : (k < 15) ? aw2 : gw2;
          s2 += (L.V[r * 18 + k] * nk) * L.V[c * 18 + k];
        }
        L.P[i] = s1 + s2;
      }
    }
    wsync();
  }

  // ---- sqrt_info = LLT(P^-1).matrixL()^T : partial-pivot Gauss-Jordan then Cholesky -------
  double* LA = L.F;
  double* LI = L.T;
  for (int i = lane; i < 225; i += 64) {
    LA[i] = L.P[i];
    LI[i] = (i / 15 == i % 15) ? 1.0 : 0.0;
  }
  wsync();
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
  if (live) {
    double* od = a.out_delta + iv * 10;
    if (lane == 0) {
      od[0] = dp.x, od[1] = dp.y, od[2] = dp.z;
      od[3] = dq.x, od[4] = dq.y, od[5] = dq.z, od[6] = dq.w;
      od[7] = dv.x, od[8] = dv.y, od[9] = dv.z;
      a.out_sum_dt[iv] = sum_dt;
    }
    for (int i = lane; i < 225; i += 64) {
      a.out_jacobian[iv * 225 + i] = L.J[i];
      a.out_covariance[iv * 225 + i] = L.P[i];
      const int r = i / 15, c = i % 15;
      a.out_sqrt_info[iv * 225 + i] = (c >= r) ? LI[c * 15 + r] : 0.0;  // U = L^T
    }
  }
}

void launch_preint(const PreintArgs& a, hipStream_t stream) {
  const long n_iv = (long)a.n_windows * 10;
  const int blocks = (int)((n_iv + PW - 1) / PW);
  hipLaunchKernelGGL(preint_kernel, dim3(blocks), dim3(64 * PW), sizeof(PreLds) * PW, stream, a);
}

}  // namespace avm
