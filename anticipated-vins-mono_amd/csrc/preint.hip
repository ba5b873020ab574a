// preint.hip — IntegrationBase on gfx950: midpoint pre-integration of raw IMU samples and
// the IMU factor's constant sqrt_info.
//   reference: vins_estimator/src/factor/integration_base.h:13-28 (ctor/noise), :54-128
//   (midPointIntegration: F, V, jacobian = F*jacobian, covariance = F P F^T + V Q V^T),
//   :130-158 (propagate), and imu_factor.h:64 (sqrt_info = LLT(cov^-1).matrixL()^T, which the
//   reference recomputes on every Evaluate; it is constant during a solve, so hoisted here).
// Mapping: one 64-lane wavefront per FOUR (window, interval) pairs, two kernels.
//  preint_kernel   the sample loop.  Its scalar part (the running delta_p / delta_q / delta_v, the rotation blocks of F and
//                  V) is lane-agnostic code: each 16-lane group runs it for its own interval, so one instruction stream
//                  serves four intervals.  The matrix products need all 64 lanes per interval: they are done for the four
//                  intervals one after the other, each with its own 15x15 state in registers and its own images of the
//                  sample-dependent rows of F and V in LDS (the other rows are synthesized in registers).
//  sqrt_info_kernel  lane = (interval, row): the 15 x 30 tableau [P | I] lives in registers, a row per lane; the pivot row
//                  of a step travels through a 30-double LDS buffer.  Partial pivoting without moving rows (every lane
//                  tracks the position its row would have after the swaps, which also settles ties the way the swapped
//                  storage would), same operations per entry as the row-swapping form.
#include "devmath.hpp"
#include "kernels.hpp"

namespace avm {

namespace {
constexpr int PW = 2;  // wavefronts per block
constexpr int PG = 4;  // intervals per wavefront

// every wavefront works on its own intervals: only wave-level ordering of its LDS traffic is needed
AVM_DEV void wsync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  __builtin_amdgcn_sched_barrier(0);  // keep the phases of a sample from being interleaved (register pressure)
}

// Row strides of the two images: 17 and 22 doubles (34 and 44 banks).  The operand reads below take, per half wavefront, rows 0..8 at
// two k offsets each: with the natural strides 16 / 20 (32 / 40 banks) rows 0, 2, 4, 6, 8 of F - and rows 0 and 8 of V - fall on the
// same banks (a five-way conflict on every operand read: 10 conflict cycles per LDS instruction in profiles/r03f.md); with 17 / 22
// the nine rows of a half wavefront hit 18 distinct bank pairs.
constexpr int FS = 17, VS = 22;
struct PreLds {
  double Fi[9 * FS];  // rows p, theta, v of F (the reference's rows 0..8); COLUMNS in the new order theta 0 | v 3 | ba 6 | bg 9 (| p 12: never read)
  double Vi[9 * VS];  // rows p, theta, v of V, noise columns 0..11 (the reference's order); the rows ba, bg are zero there
  double m[72];       // Rd, Rr, Ra0, Ra1, IRw (I - Rw*dt), T1=Rd*Ra0, T2=Rr*Ra1, T3=T2*IRw
};
typedef double d4 __attribute__((ext_vector_type(4)));

AVM_DEV double readlane_f64(double v, int srclane) {  // srclane must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}
}  // namespace

// The 15 x 15 state matrices never leave registers: with v_mfma_f64_16x16x4 the accumulator layout
// (lane l, register r) = M[(l >> 4) + 4 r][l & 15] is also the layout of a B operand (k = (l >> 4) + 4 m), so
//   jacobian   <- F * jacobian                      3 MFMAs, the result is the next B operand
//   covariance <- F * (P * F^T) + V * (Q V^T)       3 + 3 + 3 MFMAs; P is symmetric, so its accumulator registers
//                                                   double as the A operand P[l & 15][(l >> 4) + 4 m], and the
//                                                   A-layout registers of F (of V) double as the B operand F^T (V^T)
// Twelve instead of seventeen (round 5) by the order the state is held in: theta | v | ba | bg | p instead of the reference's
// p | theta | v | ba | bg (NEW index n <-> reference index n < 12 ? n + 3 : n - 12; undone by the final store).  F's p COLUMNS are
// columns of the identity (nothing depends on delta_p, integration_base.h:90-105), and they now fill the fourth k-step of a product
// together with the padding: that k-step is "add rows / columns p of the other operand", i.e. the accumulator's initial value.
// V's noise columns 12..17 (the bias random walks) only reach the rows ba, bg, as I dt (:119-120): V Q V^T is the product over
// the first twelve noise columns (three k-steps; the ba, bg rows of V are zero there) plus dt^2 sigma_w^2 on six diagonal entries.
// Every term left out is an exact zero or a product with 1.0: the results differ from the seventeen-MFMA form by the order of the sums.
__global__ __launch_bounds__(64 * PW) void preint_kernel(PreintArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  PreLds* all = reinterpret_cast<PreLds*>(smem_raw);
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const long n_iv = (long)a.n_windows * 10;
  const long iv0 = ((long)blockIdx.x * PW + wv) * PG;  // first interval of this wavefront (interval index = w*10 + j)
  if (iv0 >= n_iv) return;                             // no block-level barriers below
  PreLds* Lw = all + wv * PG;                          // the wavefront's four images
  // ---- scalar side: this lane works for interval iv0 + lk (clamped: a group past the end repeats the last interval and
  //      stores nothing)
  const bool gv = iv0 + lk < n_iv;
  const long iv = gv ? iv0 + lk : n_iv - 1;
  PreLds& L = Lw[lk];
  // (clamped to the table's stride: the pre-integration may be enqueued behind the table check whose verdict the host reads while it
  //  runs - avm_api.hip, validate_windows_begin / _end; a batch with a bad imu_n is refused either way, it must only not be read out of bounds)
  const int ns = gv ? min(max(a.imu_n[iv], 0), a.max_samp) : 0;
  int nsg[PG], ns_max = 0;
#pragma unroll
  for (int g = 0; g < PG; g++) nsg[g] = __builtin_amdgcn_readlane(ns, 16 * g), ns_max = max(ns_max, nsg[g]);

  const double* acc = a.imu_acc + iv * (a.max_samp + 1) * 3;
  const double* gyr = a.imu_gyr + iv * (a.max_samp + 1) * 3;
  const double* dts = a.imu_dt + iv * a.max_samp;
  const v3 lba = mk3(a.imu_lin_ba[iv * 3], a.imu_lin_ba[iv * 3 + 1], a.imu_lin_ba[iv * 3 + 2]);
  const v3 lbg = mk3(a.imu_lin_bg[iv * 3], a.imu_lin_bg[iv * 3 + 1], a.imu_lin_bg[iv * 3 + 2]);

  // static part of the images: zeros, the identity blocks of F's rows 0..8 (16 lanes per image)
  for (int i = li; i < 9 * FS; i += 16) L.Fi[i] = (i / FS >= 6 && i % FS == i / FS - 3) ? 1.0 : 0.0;  // F[v][v] = I; everything else is written per sample or zero
  for (int i = li; i < 9 * VS; i += 16) L.Vi[i] = 0.0;
  d4 Jb[PG], Pb[PG];
#pragma unroll
  for (int g = 0; g < PG; g++) {
    Pb[g] = d4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) Jb[g][r] = (lk + 4 * r == li && li < 15) ? 1.0 : 0.0;
  }
  // noise variance of V's column k = lk + 4 m < 12 (integration_base.h:21-27), and of the random walks on this lane's diagonal entries
  // (register r holds row lk + 4 r, column li; new rows 6..8 = ba, 9..11 = bg)
  double qn[3], qw[4];
#pragma unroll
  for (int m = 0; m < 3; m++) {
    const int k = lk + 4 * m;
    qn[m] = (k < 3 || (k >= 6 && k < 9)) ? a.acc_n * a.acc_n : a.gyr_n * a.gyr_n;
  }
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = lk + 4 * r;
    qw[r] = (row != li || row < 6 || row >= 12) ? 0.0 : (row < 9 ? a.acc_w * a.acc_w : a.gyr_w * a.gyr_w);
  }
  // running state of the lane's interval (the 16 lanes of a group hold copies)
  v3 dp = mk3(0, 0, 0), dv = mk3(0, 0, 0);
  quat dq{1, 0, 0, 0};
  v3 acc0 = mk3(acc[0], acc[1], acc[2]), gyr0 = mk3(gyr[0], gyr[1], gyr[2]);
  double sum_dt = 0;
  wsync();

  for (int s = 0; s < ns_max; s++) {
    const bool on = s < ns;              // (uniform within a 16-lane group)
    const int sc = on ? s : 0;           // clamped sample index: the loads stay inside the interval's arrays
    const double dt = dts[sc];
    const v3 acc1 = mk3(acc[3 * (sc + 1)], acc[3 * (sc + 1) + 1], acc[3 * (sc + 1) + 2]);
    const v3 gyr1 = mk3(gyr[3 * (sc + 1)], gyr[3 * (sc + 1) + 1], gyr[3 * (sc + 1) + 2]);
    // integration_base.h:63-69
    v3 un_acc_0 = qrot(dq, acc0 - lba);
    v3 un_gyr = 0.5 * (gyr0 + gyr1) - lbg;
    quat rq = qmul(dq, quat{1, un_gyr.x * dt / 2, un_gyr.y * dt / 2, un_gyr.z * dt / 2});
    v3 un_acc_1 = qrot(rq, acc1 - lba);
    v3 un_acc = 0.5 * (un_acc_0 + un_acc_1);
    v3 rp = dp + dt * dv + (0.5 * dt * dt) * un_acc;
    v3 rv = dv + dt * un_acc;
    if (li == 0 && on) {
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
    if (li < 9 && on) {  // T1 = Rd * R_a_0_x, T2 = Rr * R_a_1_x
      const int r = li / 3, c = li % 3;
#pragma unroll
      for (int o = 0; o < 18; o += 9)
        L.m[45 + o + li] = L.m[o + 3 * r] * L.m[18 + o + c] + L.m[o + 3 * r + 1] * L.m[18 + o + 3 + c] + L.m[o + 3 * r + 2] * L.m[18 + o + 6 + c];
    }
    wsync();
    if (li < 9 && on) {  // T3 = T2 * (I - R_w_x dt)
      const int r = li / 3, c = li % 3;
      L.m[63 + li] = L.m[54 + 3 * r] * L.m[36 + c] + L.m[54 + 3 * r + 1] * L.m[36 + 3 + c] + L.m[54 + 3 * r + 2] * L.m[36 + 6 + c];
    }
    if (on) {
      dp = rp;
      dv = rv;
      dq = qnormalized(rq);  // integration_base.h:153
      sum_dt += dt;
      acc0 = acc1;
      gyr0 = gyr1;
    }
    wsync();
    if (li < 9 && on) {
      const int r = li / 3, c = li % 3;
      const double Rd = L.m[li], Rr = L.m[9 + li], T1 = L.m[45 + li], T2 = L.m[54 + li], T3 = L.m[63 + li];
      const double I = (r == c) ? 1.0 : 0.0;
      const double dt2 = dt * dt;
      // F (integration_base.h:90-105)
      // (image rows: the reference's 0..8 = p, theta, v; image columns: theta 0 | v 3 | ba 6 | bg 9)
      L.Fi[(0 + r) * FS + 0 + c] = -0.25 * T1 * dt2 + -0.25 * T3 * dt2;
      L.Fi[(0 + r) * FS + 3 + c] = I * dt;
      L.Fi[(0 + r) * FS + 6 + c] = -0.25 * (Rd + Rr) * dt2;
      L.Fi[(0 + r) * FS + 9 + c] = -0.25 * T2 * dt2 * -dt;
      L.Fi[(3 + r) * FS + 0 + c] = L.m[36 + li];
      L.Fi[(3 + r) * FS + 9 + c] = -1.0 * I * dt;
      L.Fi[(6 + r) * FS + 0 + c] = -0.5 * T1 * dt + -0.5 * T3 * dt;
      L.Fi[(6 + r) * FS + 6 + c] = -0.5 * (Rd + Rr) * dt;
      L.Fi[(6 + r) * FS + 9 + c] = -0.5 * T2 * dt * -dt;
      // V (integration_base.h:108-120)
      const double v03 = 0.25 * -T2 * dt2 * 0.5 * dt;
      const double v63 = 0.5 * -T2 * dt * 0.5 * dt;
      L.Vi[(0 + r) * VS + 0 + c] = 0.25 * Rd * dt2;
      L.Vi[(0 + r) * VS + 3 + c] = v03;
      L.Vi[(0 + r) * VS + 6 + c] = 0.25 * Rr * dt2;
      L.Vi[(0 + r) * VS + 9 + c] = v03;
      L.Vi[(3 + r) * VS + 3 + c] = 0.5 * I * dt;
      L.Vi[(3 + r) * VS + 9 + c] = 0.5 * I * dt;
      L.Vi[(6 + r) * VS + 0 + c] = 0.5 * Rd * dt;
      L.Vi[(6 + r) * VS + 3 + c] = v63;
      L.Vi[(6 + r) * VS + 6 + c] = 0.5 * Rr * dt;
      L.Vi[(6 + r) * VS + 9 + c] = v63;
    }
    wsync();
    // ---- matrix side: the four intervals in turn, all 64 lanes each.  Lane li is row li of the NEW order: rows 0..5 (theta, v) are
    //      image rows 3..8, rows 12..14 (p) image rows 0..2; rows 6..11 (ba, bg) do not depend on the rotations: F's are rows of the
    //      identity, V's are zero in the first twelve noise columns (integration_base.h:102-104, 119-120)
    const bool dat = li < 6 || (li >= 12 && li < 15);
    const int lic = li < 6 ? li + 3 : (dat ? li - 12 : 0);
#pragma unroll
    for (int g = 0; g < PG; g++) {
      if (s >= nsg[g]) continue;  // (wave-uniform)
      const PreLds& G = Lw[g];
      const double dtg = readlane_f64(dt, 16 * g);
      double fa[3], va[3];
#pragma unroll
      for (int m = 0; m < 3; m++) fa[m] = G.Fi[lic * FS + lk + 4 * m];
#pragma unroll
      for (int m = 0; m < 3; m++) va[m] = G.Vi[lic * VS + lk + 4 * m];
#pragma unroll
      for (int m = 0; m < 3; m++) fa[m] = dat ? fa[m] : ((li < 12 && lk + 4 * m == li) ? 1.0 : 0.0);
#pragma unroll
      for (int m = 0; m < 3; m++) va[m] = dat ? va[m] : 0.0;
      // jacobian = F * jacobian: the k-step over the p columns of F (columns of the identity) is "+ rows p of the jacobian" (register 3 of
      // the lane groups 0..2; group 3 holds the padding row)
      d4 Jn = {0, 0, 0, lk < 3 ? Jb[g][3] : 0.0};
#pragma unroll
      for (int m = 0; m < 3; m++) Jn = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], Jb[g][m], Jn, 0, 0, 0);
      // Z = P * F^T (its columns p start as P's columns p); covariance = F * Z (rows p start as Z's rows p) + V * (Q V^T)
      const bool pc = li >= 12 && li < 15;
      d4 Z = {pc ? Pb[g][0] : 0.0, pc ? Pb[g][1] : 0.0, pc ? Pb[g][2] : 0.0, pc ? Pb[g][3] : 0.0};
#pragma unroll
      for (int m = 0; m < 3; m++) Z = __builtin_amdgcn_mfma_f64_16x16x4f64(Pb[g][m], fa[m], Z, 0, 0, 0);
      d4 Pn = {0, 0, 0, lk < 3 ? Z[3] : 0.0};
#pragma unroll
      for (int m = 0; m < 3; m++) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[m], Z[m], Pn, 0, 0, 0);
#pragma unroll
      for (int m = 0; m < 3; m++) Pn = __builtin_amdgcn_mfma_f64_16x16x4f64(va[m], va[m] * qn[m], Pn, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 4; r++) Pn[r] = fma(dtg, dtg * qw[r], Pn[r]);  // the random walks: (I dt) sigma_w^2 (I dt) on the diagonals of ba, bg
      Jb[g] = Jn;
      Pb[g] = Pn;
    }
    wsync();  // the images are rewritten by the next sample
  }

  if (gv && li == 0) {
    double* od = a.out_delta + iv * 10;
    od[0] = dp.x, od[1] = dp.y, od[2] = dp.z;
    od[3] = dq.x, od[4] = dq.y, od[5] = dq.z, od[6] = dq.w;
    od[7] = dv.x, od[8] = dv.y, od[9] = dv.z;
    a.out_sum_dt[iv] = sum_dt;
  }
#pragma unroll
  for (int g = 0; g < PG; g++) {
    if (iv0 + g >= n_iv) continue;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = lk + 4 * r;
      if (row < 15 && li < 15) {  // back to the reference's order p | theta | v | ba | bg
        const int ro = row < 12 ? row + 3 : row - 12, co = li < 12 ? li + 3 : li - 12;
        a.out_jacobian[(iv0 + g) * 225 + ro * 15 + co] = Jb[g][r];
        a.out_covariance[(iv0 + g) * 225 + ro * 15 + co] = Pb[g][r];
      }
    }
  }
}

// ---- sqrt_info = LLT(P^-1).matrixL()^T (imu_factor.h:64): Gaussian elimination with partial pivoting on [P | I], back
// substitution, then the lower Cholesky factor of the inverse (column algorithm, the oracle's llt_lower) --------------------
__global__ __launch_bounds__(64 * PW) void sqrt_info_kernel(PreintArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int li = lane & 15, lk = lane >> 4;
  const long n_iv = (long)a.n_windows * 10;
  const long iv0 = ((long)blockIdx.x * PW + wv) * PG;
  if (iv0 >= n_iv) return;
  double* buf = reinterpret_cast<double*>(smem_raw) + (wv * PG + lk) * 32;  // the group's pivot-row buffer
  const bool gv = iv0 + lk < n_iv;
  const long iv = gv ? iv0 + lk : n_iv - 1;
  const bool rowv = li < 15;
  double A[15], Iv[15];
  {
    const double* P = a.out_covariance + iv * 225 + min(li, 14) * 15;
#pragma unroll
    for (int c = 0; c < 15; c++) A[c] = P[c], Iv[c] = c == li ? 1.0 : 0.0;
  }
  int pos = li;  // where this lane's row would sit after the row swaps done so far
  // forward elimination
#pragma unroll
  for (int k = 0; k < 15; k++) {
    // pivot: among the rows at positions k..14 the largest |a_k|, the lowest position on a tie
    double best = (rowv && pos >= k) ? fabs(A[k]) : -1.0;
    int bp = pos, bl = li;
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) {
      const double ob = __shfl_xor(best, o, 64);
      const int op = __shfl_xor(bp, o, 64), ol = __shfl_xor(bl, o, 64);
      const bool take = ob > best || (ob == best && op < bp);
      best = take ? ob : best, bp = take ? op : bp, bl = take ? ol : bl;
    }
    // the swap of positions k and bp
    if (rowv) pos = li == bl ? k : (pos == k ? bp : pos);
    if (li == bl) {
#pragma unroll
      for (int c = k; c < 15; c++) buf[c] = A[c];
#pragma unroll
      for (int c = 0; c < 15; c++) buf[15 + c] = Iv[c];
    }
    wsync();
    if (rowv && pos > k) {
      const double f = A[k] / buf[k];
#pragma unroll
      for (int c = k; c < 15; c++) A[c] = A[c] - f * buf[c];
#pragma unroll
      for (int c = 0; c < 15; c++) Iv[c] = Iv[c] - f * buf[15 + c];
    }
    wsync();
  }
  // back substitution
#pragma unroll
  for (int k = 14; k >= 0; k--) {
    if (rowv && pos == k) {
      const double piv = A[k];
#pragma unroll
      for (int c = 0; c < 15; c++) Iv[c] = Iv[c] / piv, buf[c] = Iv[c];
    }
    wsync();
    if (rowv && pos < k) {
#pragma unroll
      for (int c = 0; c < 15; c++) Iv[c] = Iv[c] - A[k] * buf[c];
    }
    wsync();
  }
  // lower Cholesky of the inverse; the lane at position r holds row r, overwritten in place by row r of L
#pragma unroll
  for (int k = 0; k < 15; k++) {
    if (rowv && pos == k) {
#pragma unroll
      for (int j = 0; j < k; j++) buf[j] = Iv[j];
    }
    wsync();
    double sacc = Iv[k];
#pragma unroll
    for (int j = 0; j < k; j++) sacc -= Iv[j] * buf[j];
    if (rowv && pos == k) buf[16] = sacc;
    wsync();
    const double lkk = sqrt(buf[16]);
    if (rowv && pos >= k) Iv[k] = pos == k ? lkk : sacc / lkk;
    wsync();
  }
  // U = L^T: the lane at position c writes column c
  if (gv && rowv) {
    double* out = a.out_sqrt_info + iv * 225;
#pragma unroll
    for (int r = 0; r < 15; r++) out[r * 15 + pos] = r <= pos ? Iv[r] : 0.0;
  }
}

void launch_preint(const PreintArgs& a, hipStream_t stream) {
  const long n_iv = (long)a.n_windows * 10;
  const int blocks = (int)((n_iv + PW * PG - 1) / (PW * PG));
  hipLaunchKernelGGL(preint_kernel, dim3(blocks), dim3(64 * PW), sizeof(PreLds) * PW * PG, stream, a);
  hipLaunchKernelGGL(sqrt_info_kernel, dim3(blocks), dim3(64 * PW), sizeof(double) * 32 * PW * PG, stream, a);
}

}  // namespace avm
