// window_solve.hip — Estimator::optimization() for a batch of independent sliding windows
// on gfx950, one 512-thread workgroup per window, the whole trust-region loop on device.
//
// What it replaces (reference, all CPU):
//   problem assembly            vins_estimator/src/estimator.cpp:663-755
//   ceres::Solve (DENSE_SCHUR + DOGLEG, <= 8 iterations)          estimator.cpp:794-809
//     = per-factor Evaluate     factor/imu_factor.h:19-179, factor/projection_factor.cpp:21-121,
//                               factor/marginalization_factor.cpp:333-381
//     + CauchyLoss/Corrector    (Ceres; in-tree copy factor/marginalization_factor.cpp:37-68)
//     + Jacobi scaling, J^T J, Schur elimination of the inverse depths, dense Cholesky,
//       traditional dogleg, step acceptance                       (Ceres 1.14)
//   double2vector + vector2double gauge fix                       estimator.cpp:477-587
//
// Data placement: the reduced 165x165 system lives in LDS for the whole solve (row-packed lower
// triangle, rows padded to even length: 13778 doubles = 110 KB); states, gradient, dogleg
// vectors, per-frame rotation blocks in LDS too (~163 KB total of the 160 KiB CU budget, one
// workgroup per CU).  Per-factor Jacobian rows (28 doubles / observation) and E^T F rows are
// streamed through a per-workgroup global scratch slot that stays L2 resident.
// All arithmetic FP64.  Every reduction has a fixed order, so results are bit-reproducible
// run to run and independent of how windows are sharded over ranks.
#include <cfloat>

#include "devmath.hpp"
#include "kernels.hpp"

namespace avm {

namespace {

#define PROF_T0() long long pt__ = clock64()
#define PROF(c, k) do { if ((c).prof && threadIdx.x == 0) { long long n__ = clock64(); (c).prof[k] += n__ - pt__; pt__ = n__; } } while (0)

constexpr int NT = 512;          // threads per workgroup (8 wavefronts)
constexpr int SROWS = 13778;     // padded packed lower triangle of a 165x165 matrix
constexpr int VEC = 320;         // padded NCOL
constexpr int XN = 328;          // pose 77 | speedbias 99 | inv depth 150 (+2 pad)
constexpr int XSB = 77, XLAM = 176;
constexpr int WCH = 40;          // features per Schur staging chunk

// LDS carve (offsets in doubles)
constexpr int L_S = 0;
constexpr int L_G = L_S + SROWS;   // scaled gradient g (f | e)
constexpr int L_Y = L_G + VEC;     // Gauss-Newton solution y of (H + mu D^2) y = g
constexpr int L_DG = L_Y + VEC;    // g / D
constexpr int L_DD = L_DG + VEC;   // D
constexpr int L_SC = L_DD + VEC;   // Jacobi scaling
constexpr int L_ST = L_SC + VEC;   // trust region step (scaled space)
constexpr int L_X = L_ST + VEC;
constexpr int L_XC = L_X + XN;
constexpr int L_FR = L_XC + XN;    // [2][198]: R (11x9) then A = ric^T R^T (11x9)
constexpr int L_RIC = L_FR + 396;  // ric 9, tic 3, current ex_pose 7 (+1 pad)
constexpr int L_HEE = L_RIC + 20;  // E^T E (150) padded
constexpr int L_DXP = L_HEE + 152;
constexpr int L_RP = L_DXP + MAXPRIOR;
constexpr int L_RED = L_RP + MAXPRIOR;
constexpr int L_WCH = L_RED + 64;
constexpr int L_INT = L_WCH + WCH * NPOSE;  // int region (as doubles): 360 doubles = 720 ints
constexpr int L_SUM = L_INT + 360;  // cost_trace[16], radius_trace[16]
constexpr int L_END = L_SUM + 32;
static_assert(L_END * 8 <= 163840, "LDS budget exceeded");
// int carve (offsets in ints from L_INT)
constexpr int I_FSTART = 0, I_FNOBS = 150, I_FOBS = 300, I_PIDX = 450, I_FS = 546, I_PBLK = 560 /* kind,frame,off x16 */, I_FAIL = 620, I_END = 624;
static_assert(I_END <= 720, "int carve");

AVM_DEV int roff(int i) {
  const int q = i >> 1;
  return (i & 1) ? 2 * (q + 1) * (q + 1) : 2 * q * (q + 1);
}

struct Frames {
  const double* R;  // [11][9]
  const double* A;  // [11][9]  ric^T * R_f^T
};

// ---- projection factor (projection_factor.cpp:21-121) -----------------------------------
// out: r[2], Ji[12] (2x6), Jj[12], Je[2]; returns 1/2 rho(|r|^2) ; robustified with CauchyLoss.
template <bool WANT_J>
AVM_DEV double proj_eval(const double* x, Frames fr, const double* ric, const double* tic, double pix, double piy, double pjx,
                         double pjy, double lam, int fa, int fb, double sqi, double cauchy_a, bool apply_loss, double* r,
                         double* Ji, double* Jj, double* Je) {
  const double* Ra = fr.R + fa * 9;
  const double* Rb = fr.R + fb * 9;
  const v3 Pa = mk3(x[fa * 7], x[fa * 7 + 1], x[fa * 7 + 2]);
  const v3 Pb = mk3(x[fb * 7], x[fb * 7 + 1], x[fb * 7 + 2]);
  const v3 t = mk3(tic[0], tic[1], tic[2]);
  const v3 pci = mk3(pix / lam, piy / lam, 1.0 / lam);
  const v3 pimu_i = Rmul(ric, pci) + t;
  const v3 pw = Rmul(Ra, pimu_i) + Pa;
  const v3 pimu_j = RTmul(Rb, pw - Pb);
  const v3 pcj = RTmul(ric, pimu_j - t);
  const double dep = pcj.z;
  double r0 = sqi * (pcj.x / dep - pjx);
  double r1 = sqi * (pcj.y / dep - pjy);
  const double sn = r0 * r0 + r1 * r1;
  // ceres::CauchyLoss + Corrector: rho'' < 0 => residual and Jacobian scale by sqrt(rho')
  const double b = cauchy_a * cauchy_a, c = 1.0 / b;
  const double sum = 1.0 + sn * c;
  const double inv = 1.0 / sum;
  const double rho0 = b * log(sum);
  const double srho = apply_loss ? sqrt(fmax(DBL_MIN, inv)) : 1.0;
  r[0] = srho * r0;
  r[1] = srho * r1;
  if (WANT_J) {
    const double* Ab = fr.A + fb * 9;
    const double id = 1.0 / dep, id2 = 1.0 / (dep * dep);
    const double red[6] = {sqi * id, 0.0, sqi * (-pcj.x * id2), 0.0, sqi * id, sqi * (-pcj.y * id2)};
    double M[6], MR[6], N[6];
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) {
        M[rr * 3 + cc] = srho * (red[rr * 3] * Ab[cc] + red[rr * 3 + 1] * Ab[3 + cc] + red[rr * 3 + 2] * Ab[6 + cc]);
        N[rr * 3 + cc] = srho * (red[rr * 3] * ric[cc * 3] + red[rr * 3 + 1] * ric[cc * 3 + 1] + red[rr * 3 + 2] * ric[cc * 3 + 2]);
      }
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) MR[rr * 3 + cc] = M[rr * 3] * Ra[cc] + M[rr * 3 + 1] * Ra[3 + cc] + M[rr * 3 + 2] * Ra[6 + cc];
    const v3 u = Rmul(ric, mk3(pix, piy, 1.0));  // ric * pts_i
    const double il2 = -1.0 / (lam * lam);
#pragma unroll
    for (int rr = 0; rr < 2; rr++) {
      const v3 m = mk3(M[rr * 3], M[rr * 3 + 1], M[rr * 3 + 2]);
      const v3 mr = mk3(MR[rr * 3], MR[rr * 3 + 1], MR[rr * 3 + 2]);
      const v3 n = mk3(N[rr * 3], N[rr * 3 + 1], N[rr * 3 + 2]);
      const v3 ci = cross(pimu_i, mr);  // mr * (-skew(pts_imu_i))
      const v3 cj = cross(n, pimu_j);   // n * skew(pts_imu_j)
      Ji[rr * 6 + 0] = m.x, Ji[rr * 6 + 1] = m.y, Ji[rr * 6 + 2] = m.z;
      Ji[rr * 6 + 3] = ci.x, Ji[rr * 6 + 4] = ci.y, Ji[rr * 6 + 5] = ci.z;
      Jj[rr * 6 + 0] = -m.x, Jj[rr * 6 + 1] = -m.y, Jj[rr * 6 + 2] = -m.z;
      Jj[rr * 6 + 3] = cj.x, Jj[rr * 6 + 4] = cj.y, Jj[rr * 6 + 5] = cj.z;
      Je[rr] = dot(mr, u) * il2;
    }
  }
  return 0.5 * rho0;
}

// ---- IMU factor, raw part before sqrt_info (imu_factor.h:60-175, integration_base.h:160-186).
// One thread evaluates factor i; writes raw residual (15) and, if WANT_J, the raw 15x30
// Jacobian (pose_i 6 | sb_i 9 | pose_j 6 | sb_j 9) into stage[0..465) laid out [15][31]
// (col 0 = residual).  stage must be zeroed beforehand.
template <bool WANT_J>
AVM_DEV void imu_raw(const double* x, const double* Rfr, const avm_options& o, const double* delta, const double* pj /*15x15*/,
                     double sum_dt, const double* lba, const double* lbg, int i, double* stage) {
  const int j = i + 1;
  const v3 Pi = mk3(x[i * 7], x[i * 7 + 1], x[i * 7 + 2]), Pj = mk3(x[j * 7], x[j * 7 + 1], x[j * 7 + 2]);
  const quat Qi{x[i * 7 + 6], x[i * 7 + 3], x[i * 7 + 4], x[i * 7 + 5]}, Qj{x[j * 7 + 6], x[j * 7 + 3], x[j * 7 + 4], x[j * 7 + 5]};
  const double* si = x + XSB + i * 9;
  const double* sj = x + XSB + j * 9;
  const v3 Vi = mk3(si[0], si[1], si[2]), Bai = mk3(si[3], si[4], si[5]), Bgi = mk3(si[6], si[7], si[8]);
  const v3 Vj = mk3(sj[0], sj[1], sj[2]), Baj = mk3(sj[3], sj[4], sj[5]), Bgj = mk3(sj[6], sj[7], sj[8]);
  const v3 G = mk3(o.g[0], o.g[1], o.g[2]);
  const v3 dP = mk3(delta[0], delta[1], delta[2]), dV = mk3(delta[7], delta[8], delta[9]);
  const quat dQ{delta[6], delta[3], delta[4], delta[5]};
  auto blk = [&](int r0, int c0, double* M) {
#pragma unroll
    for (int a = 0; a < 3; a++)
#pragma unroll
      for (int b = 0; b < 3; b++) M[a * 3 + b] = pj[(r0 + a) * 15 + c0 + b];
  };
  double dp_dba[9], dp_dbg[9], dq_dbg[9], dv_dba[9], dv_dbg[9];
  blk(0, 9, dp_dba), blk(0, 12, dp_dbg), blk(3, 12, dq_dbg), blk(6, 9, dv_dba), blk(6, 12, dv_dbg);
  const v3 dba = Bai - mk3(lba[0], lba[1], lba[2]), dbg = Bgi - mk3(lbg[0], lbg[1], lbg[2]);
  const quat cdq = qmul(dQ, deltaQ(Rmul(dq_dbg, dbg)));
  const v3 cdv = dV + Rmul(dv_dba, dba) + Rmul(dv_dbg, dbg);
  const v3 cdp = dP + Rmul(dp_dba, dba) + Rmul(dp_dbg, dbg);
  const quat Qi_inv = qinv(Qi);
  const v3 tp = qrot(Qi_inv, (0.5 * sum_dt * sum_dt) * G + Pj - Pi - sum_dt * Vi);
  const v3 tv = qrot(Qi_inv, sum_dt * G + Vj - Vi);
  const v3 rp = tp - cdp;
  const quat qe = qmul(qinv(cdq), qmul(Qi_inv, Qj));
  const v3 rr = mk3(2.0 * qe.x, 2.0 * qe.y, 2.0 * qe.z);
  const v3 rv = tv - cdv;
  const v3 rba = Baj - Bai, rbg = Bgj - Bgi;
  for (int k = 0; k < 3; k++) {
    stage[(0 + k) * 31] = get(rp, k);
    stage[(3 + k) * 31] = get(rr, k);
    stage[(6 + k) * 31] = get(rv, k);
    stage[(9 + k) * 31] = get(rba, k);
    stage[(12 + k) * 31] = get(rbg, k);
  }
  if (WANT_J) {
    const double* Ri = Rfr + i * 9;  // R_i ; R_i^T = (Qi.inverse()).toRotationMatrix() for unit Qi
    auto put = [&](int r0, int c0, const double* M, double sgn) {
#pragma unroll
      for (int a = 0; a < 3; a++)
#pragma unroll
        for (int b = 0; b < 3; b++) stage[(r0 + a) * 31 + 1 + c0 + b] = sgn * M[a * 3 + b];
    };
    double RiT[9];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) RiT[a * 3 + b] = Ri[b * 3 + a];
    double M[9], M2[9];
    // pose_i (cols 0..5)
    put(0, 0, RiT, -1.0);
    skew9(tp, M);
    put(0, 3, M, 1.0);
    qleft_qright_br(qmul(qinv(Qj), Qi), cdq, M);
    put(3, 3, M, -1.0);
    skew9(tv, M);
    put(6, 3, M, 1.0);
    // speedbias_i (cols 6..14): V 6.., BA 9.., BG 12..
    for (int a = 0; a < 9; a++) M[a] = RiT[a] * sum_dt;
    put(0, 6, M, -1.0);
    put(0, 9, dp_dba, -1.0);
    put(0, 12, dp_dbg, -1.0);
    qleft_br(qmul(qmul(qinv(Qj), Qi), dQ), M);
    mat3mul(M, dq_dbg, M2);
    put(3, 12, M2, -1.0);
    put(6, 6, RiT, -1.0);
    put(6, 9, dv_dba, -1.0);
    put(6, 12, dv_dbg, -1.0);
    for (int a = 0; a < 3; a++) {
      stage[(9 + a) * 31 + 1 + 9 + a] = -1.0;
      stage[(12 + a) * 31 + 1 + 12 + a] = -1.0;
    }
    // pose_j (cols 15..20)
    put(0, 15, RiT, 1.0);
    qleft_br(qmul(qmul(qinv(cdq), Qi_inv), Qj), M);
    put(3, 18, M, 1.0);
    // speedbias_j (cols 21..29)
    put(6, 21, RiT, 1.0);
    for (int a = 0; a < 3; a++) {
      stage[(9 + a) * 31 + 1 + 24 + a] = 1.0;
      stage[(12 + a) * 31 + 1 + 27 + a] = 1.0;
    }
  }
}

// state column of IMU-factor-local column c (0..29) for factor i
AVM_DEV int imu_col(int i, int c) {
  if (c < 6) return 6 * i + c;
  if (c < 15) return SB0 + 9 * i + (c - 6);
  if (c < 21) return 6 * (i + 1) + (c - 15);
  return SB0 + 9 * (i + 1) + (c - 21);
}

// MarginalizationFactor dx of one kept block (marginalization_factor.cpp:346-363)
AVM_DEV void prior_block_dx(int kind, const double* xb, const double* x0, double* dx) {
  if (kind == AVM_BLK_SPEEDBIAS) {
    for (int k = 0; k < 9; k++) dx[k] = xb[k] - x0[k];
  } else {
    for (int k = 0; k < 3; k++) dx[k] = xb[k] - x0[k];
    const quat q0{x0[6], x0[3], x0[4], x0[5]}, q{xb[6], xb[3], xb[4], xb[5]};
    const quat d = qmul(qinv(q0), q);
    const double sg = (d.w >= 0) ? 2.0 : -2.0;
    dx[3] = sg * d.x, dx[4] = sg * d.y, dx[5] = sg * d.z;
  }
}

struct WinCtx {
  long long* prof;
  double* lds;
  int* ids;
  double* sc;   // global scratch slot
  int32_t* osf; // observation slot -> feature
  int w, nf, nobs_tot, pn, pnblk;
  const double* obs;   // [max_obs][2]
  const double *pdelta, *pjac, *psqrt, *psum;  // this window's 10 intervals
  const double *lba, *lbg;
  const double *pJ, *pr, *px0;  // prior
  int ldp;
};

// frames: R_f and A_f = ric^T R_f^T for state vector xs into frame slot `which`
AVM_DEV void build_frames(double* lds, const double* xs, int which) {
  const int t = threadIdx.x;
  double* R = lds + L_FR + which * 198;
  double* A = R + 99;
  const double* ric = lds + L_RIC;
  if (t < NFR) {
    quat q{xs[t * 7 + 6], xs[t * 7 + 3], xs[t * 7 + 4], xs[t * 7 + 5]};
    double Rm[9];
    q2R(q, Rm);
    for (int k = 0; k < 9; k++) R[t * 9 + k] = Rm[k];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) A[t * 9 + a * 3 + b] = ric[0 * 3 + a] * Rm[b * 3 + 0] + ric[1 * 3 + a] * Rm[b * 3 + 1] + ric[2 * 3 + a] * Rm[b * 3 + 2];
  }
}

// prior residual r_p = r0 + J0 * dx(xs) into lds[L_RP]; returns (to all threads) nothing; needs syncs by caller
AVM_DEV void prior_residual_dev(const WinCtx& c, const double* xs) {
  double* lds = c.lds;
  const int t = threadIdx.x;
  if (t < c.pnblk) {
    const int kind = c.ids[I_PBLK + t * 3], fr = c.ids[I_PBLK + t * 3 + 1], off = c.ids[I_PBLK + t * 3 + 2];
    // ex_pose is constant in the solve; its current value sits behind ric/tic
    const double* xb = kind == AVM_BLK_POSE ? xs + fr * 7 : (kind == AVM_BLK_SPEEDBIAS ? xs + XSB + fr * 9 : lds + L_RIC + 12);
    double dx[9];
    prior_block_dx(kind, xb, c.px0 + t * 9, dx);
    const int n = kind == AVM_BLK_SPEEDBIAS ? 9 : 6;
    for (int k = 0; k < n; k++) lds[L_DXP + off + k] = dx[k];
  }
  __syncthreads();
  // r_p[i] = r0[i] + sum_k J0[i][k] dx[k] : one wave per row group, lanes over k (k < 128 supported)
  const int lane = t & 63, wv = t >> 6;
  for (int i = wv; i < c.pn; i += NT / 64) {
    double s = 0;
    if (lane < c.pn) s = c.pJ[(size_t)i * c.ldp + lane] * lds[L_DXP + lane];
    if (lane + 64 < c.pn) s += c.pJ[(size_t)i * c.ldp + lane + 64] * lds[L_DXP + lane + 64];
    s = wave_sum(s);
    if (lane == 0) lds[L_RP + i] = c.pr[i] + s;
  }
  __syncthreads();
}

// residual-only cost at state xs (frames slot `which` must be built). Uses lds[L_S..] as IMU staging.
AVM_DEV double eval_cost(const WinCtx& c, const avm_options& o, const double* xs, int which) {
  double* lds = c.lds;
  const int t = threadIdx.x;
  Frames fr{lds + L_FR + which * 198, lds + L_FR + which * 198 + 99};
  const double sqi = o.focal_length / 1.5;
  double acc = 0;
  // IMU raw residuals by threads of the last wave (so they overlap with projection work of the others)
  for (int i = t; i < 10 * 31 * 15; i += NT) lds[L_S + i] = 0.0;
  __syncthreads();
  if (t >= NT - 64 && t < NT - 64 + 10) {
    const int i = t - (NT - 64);
    if (c.psum[i] <= o.max_sum_dt)
      imu_raw<false>(xs, fr.R, o, c.pdelta + i * 10, c.pjac + i * 225, c.psum[i], c.lba + i * 3, c.lbg + i * 3, i, lds + L_S + i * 465);
  }
  for (int s = t; s < c.nobs_tot; s += NT) {
    const int e = c.osf[s];
    const int s0 = c.ids[I_FOBS + e];
    if (s == s0) continue;
    const int fa = c.ids[I_FSTART + e], fb = fa + (s - s0);
    double r[2];
    acc += proj_eval<false>(xs, fr, lds + L_RIC, lds + L_RIC + 9, c.obs[2 * s0], c.obs[2 * s0 + 1], c.obs[2 * s], c.obs[2 * s + 1],
                            xs[XLAM + e], fa, fb, sqi, o.cauchy_a, true, r, nullptr, nullptr, nullptr);
  }
  __syncthreads();
  if (t < 150) {
    const int i = t / 15, r = t % 15;
    if (c.psum[i] <= o.max_sum_dt) {
      double s = 0;
      for (int k = r; k < 15; k++) s += c.psqrt[i * 225 + r * 15 + k] * lds[L_S + i * 465 + k * 31];
      acc += 0.5 * s * s;
    }
  }
  if (c.pn > 0) {
    prior_residual_dev(c, xs);
    if (t < c.pn) acc += 0.5 * lds[L_RP + t] * lds[L_RP + t];
  }
  return block_sum<NT>(acc, lds + L_RED);
}

// Full evaluation at lds[L_X]: fills S (unscaled H_ff), W, hee, g (unscaled) and returns the cost.
AVM_DEV double eval_jac(const WinCtx& c, const avm_options& o) {
  double* lds = c.lds;
  const int t = threadIdx.x;
  const double* xs = lds + L_X;
  PROF_T0();
  build_frames(lds, xs, 0);
  for (int i = t; i < 10 * 465; i += NT) lds[L_S + i] = 0.0;
  __syncthreads();
  Frames fr{lds + L_FR, lds + L_FR + 99};
  const double sqi = o.focal_length / 1.5;
  double acc = 0;
  double* JF = c.sc + Scratch::JF;
  // E1: IMU raw (last wave, 10 lanes) || projection factors (everyone, strided)
  if (t >= NT - 64 && t < NT - 64 + 10) {
    const int i = t - (NT - 64);
    if (c.psum[i] <= o.max_sum_dt)
      imu_raw<true>(xs, fr.R, o, c.pdelta + i * 10, c.pjac + i * 225, c.psum[i], c.lba + i * 3, c.lbg + i * 3, i, lds + L_S + i * 465);
  }
  for (int s = t; s < c.nobs_tot; s += NT) {
    const int e = c.osf[s];
    const int s0 = c.ids[I_FOBS + e];
    if (s == s0) continue;
    const int fa = c.ids[I_FSTART + e], fb = fa + (s - s0);
    double r[2], Ji[12], Jj[12], Je[2];
    acc += proj_eval<true>(xs, fr, lds + L_RIC, lds + L_RIC + 9, c.obs[2 * s0], c.obs[2 * s0 + 1], c.obs[2 * s], c.obs[2 * s + 1],
                           xs[XLAM + e], fa, fb, sqi, o.cauchy_a, true, r, Ji, Jj, Je);
    JF[0 * MAXOBS + s] = r[0];
    JF[1 * MAXOBS + s] = r[1];
#pragma unroll
    for (int k = 0; k < 12; k++) JF[(2 + k) * MAXOBS + s] = Ji[k], JF[(14 + k) * MAXOBS + s] = Jj[k];
    JF[26 * MAXOBS + s] = Je[0];
    JF[27 * MAXOBS + s] = Je[1];
  }
  __syncthreads();
  PROF(c, 0);
  // E2: IJ = sqrt_info * raw  (residual col 0 + 30 Jacobian cols), sqrt_info upper triangular
  double* IJ = c.sc + Scratch::IJ;
  for (int idx = t; idx < 10 * 465; idx += NT) {
    const int i = idx / 465, rc = idx % 465, r = rc / 31, cc = rc % 31;
    double s = 0;
    if (c.psum[i] <= o.max_sum_dt)
      for (int k = r; k < 15; k++) s += c.psqrt[i * 225 + r * 15 + k] * lds[L_S + i * 465 + k * 31 + cc];
    IJ[idx] = s;
    if (cc == 0) acc += 0.5 * s * s;
  }
  PROF(c, 1);
  // E1b: per-feature aggregates over the start pose (H_aa, w_a, h_e, g_a, g_e)
  double* FA = c.sc + Scratch::FA;
  for (int idx = t; idx < c.nf * 35; idx += NT) {
    const int e = idx / 35, q = idx % 35;
    const int s0 = c.ids[I_FOBS + e], no = c.ids[I_FNOBS + e];
    int ra, rb;  // JF rows (first residual row); second row = +6 for Ji, +1 for Je/r
    int stride_a, stride_b;
    if (q < 21) {
      int ci = 0;
      while ((ci + 1) * (ci + 2) / 2 <= q) ci++;
      const int cj = q - ci * (ci + 1) / 2;
      ra = 2 + ci, rb = 2 + cj, stride_a = 6, stride_b = 6;
    } else if (q < 27) {
      ra = 2 + (q - 21), rb = 26, stride_a = 6, stride_b = 1;
    } else if (q == 27) {
      ra = 26, rb = 26, stride_a = 1, stride_b = 1;
    } else if (q < 34) {
      ra = 2 + (q - 28), rb = 0, stride_a = 6, stride_b = 1;
    } else {
      ra = 26, rb = 0, stride_a = 1, stride_b = 1;
    }
    double s = 0;
    for (int k = 1; k < no; k++) {
      const int sl = s0 + k;
      s += JF[ra * MAXOBS + sl] * JF[rb * MAXOBS + sl] + JF[(ra + stride_a) * MAXOBS + sl] * JF[(rb + stride_b) * MAXOBS + sl];
    }
    FA[q * MAXE + e] = s;
  }
  PROF(c, 2);
  // prior residual (uses L_DXP/L_RP; includes syncs)
  if (c.pn > 0) {
    prior_residual_dev(c, xs);
    if (t < c.pn) acc += 0.5 * lds[L_RP + t] * lds[L_RP + t];
  } else {
    __syncthreads();
  }
  PROF(c, 3);
  // zero S and g
  for (int i = t; i < SROWS; i += NT) lds[L_S + i] = 0.0;
  for (int i = t; i < VEC; i += NT) lds[L_G + i] = 0.0;
  __syncthreads();
  // E4(A): pose-pose lower-triangular entries
  const int* fs = c.ids + I_FS;
  for (int idx = t; idx < NPOSE * (NPOSE + 1) / 2; idx += NT) {
    int i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
    while ((i + 1) * (i + 2) / 2 <= idx) i++;
    while (i * (i + 1) / 2 > idx) i--;
    const int j = idx - i * (i + 1) / 2;
    const int b = i / 6, ci = i % 6, a = j / 6, cj = j % 6;
    double s = 0;
    if (a < b) {
      const int d = b - a;
      for (int e = fs[a]; e < fs[a + 1]; e++) {
        if (c.ids[I_FNOBS + e] > d) {
          const int sl = c.ids[I_FOBS + e] + d;
          s += JF[(14 + ci) * MAXOBS + sl] * JF[(2 + cj) * MAXOBS + sl] + JF[(20 + ci) * MAXOBS + sl] * JF[(8 + cj) * MAXOBS + sl];
        }
      }
    } else {
      const int q = ci * (ci + 1) / 2 + cj;
      for (int e = fs[a]; e < fs[a + 1]; e++) s += FA[q * MAXE + e];
      for (int e = 0; e < fs[a]; e++) {
        const int d = a - c.ids[I_FSTART + e];
        if (c.ids[I_FNOBS + e] > d) {
          const int sl = c.ids[I_FOBS + e] + d;
          s += JF[(14 + ci) * MAXOBS + sl] * JF[(14 + cj) * MAXOBS + sl] + JF[(20 + ci) * MAXOBS + sl] * JF[(20 + cj) * MAXOBS + sl];
        }
      }
    }
    lds[L_S + roff(i) + j] = s;
  }
  PROF(c, 4);
  // E4(B): pose gradient
  if (t < NPOSE) {
    const int b = t / 6, ci = t % 6;
    double s = 0;
    for (int e = fs[b]; e < fs[b + 1]; e++) s += FA[(28 + ci) * MAXE + e];
    for (int e = 0; e < fs[b]; e++) {
      const int d = b - c.ids[I_FSTART + e];
      if (c.ids[I_FNOBS + e] > d) {
        const int sl = c.ids[I_FOBS + e] + d;
        s += JF[(14 + ci) * MAXOBS + sl] * JF[sl] + JF[(20 + ci) * MAXOBS + sl] * JF[MAXOBS + sl];
      }
    }
    lds[L_G + t] = s;
  }
  PROF(c, 5);
  // E4(C): W rows (E^T F) ; E4(D): E^T E and feature gradient
  double* W = c.sc + Scratch::W;
  for (int idx = t; idx < c.nf * NPOSE; idx += NT) {
    const int e = idx / NPOSE, cc = idx % NPOSE, bb = cc / 6, ck = cc % 6;
    const int a = c.ids[I_FSTART + e], d = bb - a;
    double s = 0;
    if (d == 0)
      s = FA[(21 + ck) * MAXE + e];
    else if (d > 0 && d < c.ids[I_FNOBS + e]) {
      const int sl = c.ids[I_FOBS + e] + d;
      s = JF[(14 + ck) * MAXOBS + sl] * JF[26 * MAXOBS + sl] + JF[(20 + ck) * MAXOBS + sl] * JF[27 * MAXOBS + sl];
    }
    W[idx] = s;
  }
  if (t < c.nf) {
    lds[L_HEE + t] = FA[27 * MAXE + t];
    lds[L_G + NF + t] = FA[34 * MAXE + t];
  }
  __syncthreads();
  PROF(c, 6);
  // E4(F): IMU J^T J and J^T r, even then odd factors
  for (int par = 0; par < 2; par++) {
    for (int idx = t; idx < 5 * 495; idx += NT) {
      const int i = 2 * (idx / 495) + par, q = idx % 495;
      if (c.psum[i] > o.max_sum_dt) continue;
      const double* Jm = IJ + i * 465;
      if (q < 465) {
        int p = (int)((sqrt(8.0 * q + 1.0) - 1.0) * 0.5);
        while ((p + 1) * (p + 2) / 2 <= q) p++;
        while (p * (p + 1) / 2 > q) p--;
        const int qq = q - p * (p + 1) / 2;
        double s = 0;
        for (int r = 0; r < 15; r++) s += Jm[r * 31 + 1 + p] * Jm[r * 31 + 1 + qq];
        const int ip = imu_col(i, p), iq = imu_col(i, qq);
        const int hi = max(ip, iq), lo = min(ip, iq);
        lds[L_S + roff(hi) + lo] += s;
      } else {
        const int p = q - 465;
        double s = 0;
        for (int r = 0; r < 15; r++) s += Jm[r * 31 + 1 + p] * Jm[r * 31];
        lds[L_G + imu_col(i, p)] += s;
      }
    }
    __syncthreads();
  }
  PROF(c, 7);
  // E4(G): prior  H += Hp (mapped), g += J0^T r_p
  if (c.pn > 0) {
    const double* HP = c.sc + Scratch::HP;
    const int* pidx = c.ids + I_PIDX;
    for (int idx = t; idx < c.pn * (c.pn + 1) / 2; idx += NT) {
      int p = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
      while ((p + 1) * (p + 2) / 2 <= idx) p++;
      while (p * (p + 1) / 2 > idx) p--;
      const int q = idx - p * (p + 1) / 2;
      const int ip = pidx[p], iq = pidx[q];
      if (ip < 0 || iq < 0) continue;
      const int hi = max(ip, iq), lo = min(ip, iq);
      lds[L_S + roff(hi) + lo] += HP[p * MAXPRIOR + q];
    }
    if (t < c.pn && pidx[t] >= 0) {
      double s = 0;
      for (int i = 0; i < c.pn; i++) s += c.pJ[(size_t)i * c.ldp + t] * lds[L_RP + i];
      lds[L_G + pidx[t]] += s;
    }
  }
  const double cost = block_sum<NT>(acc, lds + L_RED);
  __syncthreads();
  PROF(c, 8);
  return cost;
}

// || J' u ||^2 with J' the Jacobi-scaled Jacobian, u in lds[L_ST] (scaled space); JF/IJ valid for L_X
AVM_DEV double jac_times_vec_sq(const WinCtx& c, const avm_options& o) {
  double* lds = c.lds;
  const int t = threadIdx.x;
  const double* u = lds + L_ST;
  const double* scl = lds + L_SC;
  const double* JF = c.sc + Scratch::JF;
  const double* IJ = c.sc + Scratch::IJ;
  double acc = 0;
  for (int s = t; s < c.nobs_tot; s += NT) {
    const int e = c.osf[s];
    const int s0 = c.ids[I_FOBS + e];
    if (s == s0) continue;
    const int fa = c.ids[I_FSTART + e], fb = fa + (s - s0);
    double y0 = 0, y1 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double va = u[fa * 6 + k] * scl[fa * 6 + k], vb = u[fb * 6 + k] * scl[fb * 6 + k];
      y0 += JF[(2 + k) * MAXOBS + s] * va + JF[(14 + k) * MAXOBS + s] * vb;
      y1 += JF[(8 + k) * MAXOBS + s] * va + JF[(20 + k) * MAXOBS + s] * vb;
    }
    const double ve = u[NF + e] * scl[NF + e];
    y0 += JF[26 * MAXOBS + s] * ve;
    y1 += JF[27 * MAXOBS + s] * ve;
    acc += y0 * y0 + y1 * y1;
  }
  if (t < 150) {
    const int i = t / 15, r = t % 15;
    if (c.psum[i] <= o.max_sum_dt) {
      double y = 0;
      for (int p = 0; p < 30; p++) {
        const int col = imu_col(i, p);
        y += IJ[i * 465 + r * 31 + 1 + p] * (u[col] * scl[col]);
      }
      acc += y * y;
    }
  }
  if (c.pn > 0 && t >= 192 && t < 192 + c.pn) {
    const int i = t - 192;
    const int* pidx = c.ids + I_PIDX;
    double y = 0;
    for (int k = 0; k < c.pn; k++)
      if (pidx[k] >= 0) y += c.pJ[(size_t)i * c.ldp + k] * (u[pidx[k]] * scl[pidx[k]]);
    acc += y * y;
  }
  return block_sum<NT>(acc, lds + L_RED);
}

// In-place lower Cholesky of the packed NFxNF matrix in lds[L_S]; returns false on a non-positive pivot.
AVM_DEV bool cholesky_lds(double* lds) {
  double* S = lds + L_S;
  const int t = threadIdx.x;
  constexpr int NB = 8;
  volatile int& s_fail = reinterpret_cast<int*>(lds + L_INT)[I_FAIL];
  if (t == 0) s_fail = 0;
  __syncthreads();
  for (int c0 = 0; c0 < NF; c0 += NB) {
    const int nb = min(NB, NF - c0);
    // (1) panel update with the already factored columns [0, c0)
    if (c0 > 0) {
      for (int idx = t; idx < (NF - c0) * nb; idx += NT) {
        const int i = c0 + idx / nb, j = c0 + idx % nb;
        if (i < j) continue;
        const double* ri = S + roff(i);
        const double* rj = S + roff(j);
        double s = 0;
        for (int l = 0; l < c0; l++) s += ri[l] * rj[l];
        S[roff(i) + j] -= s;
      }
    }
    __syncthreads();
    // (2) factor the nb x nb diagonal block with one wavefront (lane r = row c0 + r)
    if (t < 64) {
      for (int jj = 0; jj < nb; jj++) {
        const int j = c0 + jj;
        if (t == jj) {
          double v = S[roff(j) + j];
          for (int l = c0; l < j; l++) v -= S[roff(j) + l] * S[roff(j) + l];
          if (!(v > 0.0)) s_fail = 1;
          S[roff(j) + j] = sqrt(v);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t > jj && t < nb) {
          const int i = c0 + t;
          double v = S[roff(i) + j];
          for (int l = c0; l < j; l++) v -= S[roff(i) + l] * S[roff(j) + l];
          S[roff(i) + j] = v / S[roff(j) + j];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      }
    }
    __syncthreads();
    if (s_fail) return false;
    // (3) panel solve: rows below the diagonal block
    for (int i = c0 + nb + t; i < NF; i += NT) {
      double* ri = S + roff(i);
      for (int jj = 0; jj < nb; jj++) {
        const int j = c0 + jj;
        const double* rj = S + roff(j);
        double v = ri[j];
        for (int l = c0; l < j; l++) v -= ri[l] * rj[l];
        ri[j] = v / rj[j];
      }
    }
    __syncthreads();
  }
  return true;
}

// Solve L L^T z = b in place on lds[vec..vec+NF) with the factor in lds[L_S]
AVM_DEV void chol_solve_lds(double* lds, int vec) {
  double* S = lds + L_S;
  double* b = lds + vec;
  const int t = threadIdx.x;
  constexpr int NB = 8;
  // forward
  for (int c0 = 0; c0 < NF; c0 += NB) {
    const int nb = min(NB, NF - c0);
    if (t == 0) {
      for (int jj = 0; jj < nb; jj++) {
        const int j = c0 + jj;
        double v = b[j];
        for (int l = c0; l < j; l++) v -= S[roff(j) + l] * b[l];
        b[j] = v / S[roff(j) + j];
      }
    }
    __syncthreads();
    for (int i = c0 + nb + t; i < NF; i += NT) {
      double v = b[i];
      for (int l = c0; l < c0 + nb; l++) v -= S[roff(i) + l] * b[l];
      b[i] = v;
    }
    __syncthreads();
  }
  // backward
  for (int c1 = NF; c1 > 0; c1 -= NB) {
    const int c0 = max(0, c1 - NB);
    if (t == 0) {
      for (int j = c1 - 1; j >= c0; j--) {
        double v = b[j];
        for (int i = j + 1; i < c1; i++) v -= S[roff(i) + j] * b[i];
        b[j] = v / S[roff(j) + j];
      }
    }
    __syncthreads();
    for (int j = t; j < c0; j += NT) {
      double v = b[j];
      for (int i = c0; i < c1; i++) v -= S[roff(i) + j] * b[i];
      b[j] = v;
    }
    __syncthreads();
  }
}

// Evaluator::Plus : xc = x (+) (step * scale)
AVM_DEV void state_plus(double* lds) {
  const int t = threadIdx.x;
  const double* x = lds + L_X;
  double* xc = lds + L_XC;
  const double* st = lds + L_ST;
  const double* scl = lds + L_SC;
  if (t < NFR) {
    const int o = t * 6;
    for (int k = 0; k < 3; k++) xc[t * 7 + k] = x[t * 7 + k] + st[o + k] * scl[o + k];
    quat q{x[t * 7 + 6], x[t * 7 + 3], x[t * 7 + 4], x[t * 7 + 5]};
    quat r = qnormalized(qmul(q, deltaQ(mk3(st[o + 3] * scl[o + 3], st[o + 4] * scl[o + 4], st[o + 5] * scl[o + 5]))));
    xc[t * 7 + 3] = r.x, xc[t * 7 + 4] = r.y, xc[t * 7 + 5] = r.z, xc[t * 7 + 6] = r.w;
  }
  if (t >= 64 && t < 64 + 99) {
    const int k = t - 64;
    xc[XSB + k] = x[XSB + k] + st[SB0 + k] * scl[SB0 + k];
  }
  if (t >= 192 && t < 192 + MAXE) {
    const int e = t - 192;
    xc[XLAM + e] = x[XLAM + e] + st[NF + e] * scl[NF + e];
  }
}

}  // namespace

__global__ __launch_bounds__(NT) void window_solve_kernel(SolveArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int t = threadIdx.x;
  const avm_options& o = A.opt;
  const avm_window_batch& B = A.b;

  for (int w = blockIdx.x; w < B.n_windows; w += gridDim.x) {
    WinCtx c;
    c.lds = lds, c.ids = ids;
    c.sc = A.scratch + (size_t)blockIdx.x * Scratch::TOTAL;
    c.osf = A.iscratch + (size_t)blockIdx.x * MAXOBS;
    c.w = w;
    c.prof = A.prof ? A.prof + (size_t)blockIdx.x * 32 : nullptr;
    c.nf = B.n_feat[w];
    c.obs = B.obs_xy + (size_t)w * B.max_obs * 2;
    c.pdelta = A.pre_delta + (size_t)w * 100, c.pjac = A.pre_jac + (size_t)w * 2250, c.psqrt = A.pre_sqrt + (size_t)w * 2250;
    c.psum = A.pre_sum_dt + (size_t)w * 10;
    c.lba = B.imu_lin_ba + (size_t)w * 30, c.lbg = B.imu_lin_bg + (size_t)w * 30;
    c.pn = B.prior_n ? B.prior_n[w] : 0;
    c.pnblk = c.pn > 0 ? B.prior_nblk[w] : 0;
    c.ldp = B.max_prior;
    c.pJ = B.prior_J + (size_t)w * B.max_prior * B.max_prior;
    c.pr = B.prior_r + (size_t)w * B.max_prior;
    c.px0 = B.prior_x0 + (size_t)w * B.max_pblk * 9;
    __syncthreads();
    PROF_T0();
    // ---------------- load ----------------
    for (int i = t; i < 77; i += NT) lds[L_X + i] = B.pose[(size_t)w * 77 + i];
    for (int i = t; i < 99; i += NT) lds[L_X + XSB + i] = B.speedbias[(size_t)w * 99 + i];
    for (int i = t; i < MAXE; i += NT) lds[L_X + XLAM + i] = i < c.nf ? B.inv_depth[(size_t)w * B.max_feat + i] : 1.0;
    for (int i = t; i < VEC; i += NT) lds[L_SC + i] = 1.0, lds[L_ST + i] = 0.0, lds[L_Y + i] = 0.0, lds[L_DG + i] = 0.0, lds[L_DD + i] = 1.0;
    for (int i = t; i < MAXPRIOR; i += NT) lds[L_DXP + i] = 0.0, lds[L_RP + i] = 0.0;
    if (t < c.nf) {
      ids[I_FSTART + t] = B.feat_start[(size_t)w * B.max_feat + t];
      ids[I_FNOBS + t] = B.feat_nobs[(size_t)w * B.max_feat + t];
      ids[I_FOBS + t] = B.feat_obs_begin[(size_t)w * B.max_feat + t];
    }
    if (t == 0) {
      const double* ex = B.ex_pose + (size_t)w * 7;
      double R[9];
      q2R(quat{ex[6], ex[3], ex[4], ex[5]}, R);
      for (int k = 0; k < 9; k++) lds[L_RIC + k] = R[k];
      for (int k = 0; k < 3; k++) lds[L_RIC + 9 + k] = ex[k];
    }
    __syncthreads();
    if (t < 7) lds[L_RIC + 12 + t] = B.ex_pose[(size_t)w * 7 + t];  // current ex_pose for the prior's dx
    if (t < c.nf) {
      const int s0 = ids[I_FOBS + t], no = ids[I_FNOBS + t];
      for (int k = 0; k < no; k++) c.osf[s0 + k] = t;
    }
    if (t <= NFR) {  // fs[a] = first feature with start >= a
      int cnt = 0;
      for (int e = 0; e < c.nf; e++) cnt += (ids[I_FSTART + e] < t) ? 1 : 0;
      ids[I_FS + t] = cnt;
    }
    if (t == 0) {
      int off = 0;
      for (int k = 0; k < c.pnblk; k++) {
        const int kind = B.prior_blk_kind[(size_t)w * B.max_pblk + k], fr = B.prior_blk_frame[(size_t)w * B.max_pblk + k];
        ids[I_PBLK + k * 3] = kind, ids[I_PBLK + k * 3 + 1] = fr, ids[I_PBLK + k * 3 + 2] = off;
        const int n = kind == AVM_BLK_SPEEDBIAS ? 9 : 6;
        for (int q = 0; q < n; q++) ids[I_PIDX + off + q] = kind == AVM_BLK_POSE ? fr * 6 + q : (kind == AVM_BLK_SPEEDBIAS ? SB0 + fr * 9 + q : -1);
        off += n;
      }
    }
    {
      int tot = 0;
      if (c.nf > 0) tot = B.feat_obs_begin[(size_t)w * B.max_feat + c.nf - 1] + B.feat_nobs[(size_t)w * B.max_feat + c.nf - 1];
      c.nobs_tot = tot;
    }
    __syncthreads();
    // Hp = J0^T J0 (constant during the solve: hoisted out of the per-iteration J^T J)
    if (c.pn > 0) {
      double* HP = c.sc + Scratch::HP;
      for (int idx = t; idx < c.pn * c.pn; idx += NT) {
        const int p = idx / c.pn, q = idx % c.pn;
        double s = 0;
        for (int i = 0; i < c.pn; i++) s += c.pJ[(size_t)i * c.ldp + p] * c.pJ[(size_t)i * c.ldp + q];
        HP[p * MAXPRIOR + q] = s;
      }
    }
    __syncthreads();

    PROF(c, 9);
    // ---------------- TrustRegionMinimizer ----------------
    if (t < 32) lds[L_SUM + t] = 0.0;
    int n_successful = 0, accept_mask = 0;
    double initial_cost = 0;
    double radius = o.initial_trust_region_radius, mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    bool reuse = false, first = true, have_alpha = false;
    double alpha = 0, dogleg_step_norm = 0;
    double gnorm = 0, gn_norm = 0, ytg = 0, jusq = 0;  // |g/D|, |D y|, y^T g, |J u|^2
    double k1 = 0, k2 = 0;                             // step = -(k1 * g/D^2 + k2 * y)
    double x_cost = 0, x_norm = 0, gradient_max_norm = 0;
    int iteration = 0, num_invalid = 0, termination = AVM_TERM_NO_CONVERGENCE;
    bool step_ok = true;

    auto amb_norm = [&](const double* xs) {
      double s = 0;
      for (int i = t; i < 176 + c.nf; i += NT) s += xs[i] * xs[i];
      return sqrt(block_sum<NT>(s, lds + L_RED));
    };
    // evaluate + scaling + gradient max norm at lds[L_X]
    auto evaluate_x = [&]() {
      { PROF_T0(); (void)pt__; }
      x_cost = eval_jac(c, o);
      PROF_T0();
      // Jacobi scaling from the column norms of the first Jacobian (diag of unscaled H)
      if (first) {
        if (o.jacobi_scaling) {
          if (t < NF) lds[L_SC + t] = 1.0 / (1.0 + sqrt(lds[L_S + roff(t) + t]));
          if (t >= 192 && t < 192 + c.nf) lds[L_SC + NF + t - 192] = 1.0 / (1.0 + sqrt(lds[L_HEE + t - 192]));
        }
        first = false;
      }
      // gradient_max_norm = |x - Plus(x, -g)|_inf with the unscaled gradient
      double gm = 0;
      {
        const double* x = lds + L_X;
        const double* g = lds + L_G;
        if (t < NFR) {
          for (int k = 0; k < 3; k++) gm = fmax(gm, fabs(g[t * 6 + k]));
          quat q{x[t * 7 + 6], x[t * 7 + 3], x[t * 7 + 4], x[t * 7 + 5]};
          quat r = qnormalized(qmul(q, deltaQ(mk3(-g[t * 6 + 3], -g[t * 6 + 4], -g[t * 6 + 5]))));
          gm = fmax(gm, fmax(fmax(fabs(q.x - r.x), fabs(q.y - r.y)), fmax(fabs(q.z - r.z), fabs(q.w - r.w))));
        }
        if (t >= 64 && t < 64 + 99) gm = fmax(gm, fabs(g[SB0 + t - 64]));
        if (t >= 192 && t < 192 + c.nf) gm = fmax(gm, fabs(g[NF + t - 192]));
      }
      gradient_max_norm = block_max<NT>(gm, lds + L_RED);
      __syncthreads();
      // scale: H' = S H S, W' , hee', g'
      const double* scl = lds + L_SC;
      for (int i = t; i < NF; i += NT) {
        double* ri = lds + L_S + roff(i);
        const double si = scl[i];
        for (int j = 0; j <= i; j++) ri[j] *= si * scl[j];
      }
      double* W = c.sc + Scratch::W;
      for (int idx = t; idx < c.nf * NPOSE; idx += NT) W[idx] *= scl[NF + idx / NPOSE] * scl[idx % NPOSE];
      if (t < c.nf) lds[L_HEE + t] *= scl[NF + t] * scl[NF + t];
      for (int i = t; i < NF + c.nf; i += NT) lds[L_G + i] *= scl[i];
      __syncthreads();
      PROF(c, 10);
    };

    x_norm = amb_norm(lds + L_X);
    evaluate_x();
    initial_cost = x_cost;
    double ref_cost = x_cost;

    while (true) {
      // FinalizeIterationAndCheckIfMinimizerCanContinue
      if (iteration > 0) {
        if (step_ok) n_successful++;
        if (iteration <= AVM_MAX_ITER_TRACE) {
          if (t == 0) lds[L_SUM + iteration - 1] = x_cost, lds[L_SUM + 16 + iteration - 1] = radius;
          if (step_ok) accept_mask |= 1 << (iteration - 1);
        }
      }
      if (iteration >= o.max_num_iterations) {
        termination = AVM_TERM_NO_CONVERGENCE;
        break;
      }
      if (step_ok && gradient_max_norm <= o.gradient_tolerance) {
        termination = AVM_TERM_GRADIENT_TOL;
        break;
      }
      if (radius <= o.min_trust_region_radius) {
        termination = AVM_TERM_MIN_RADIUS;
        break;
      }
      iteration++;
      step_ok = false;
      bool solver_ok = true;
      if (!reuse) {
        reuse = true;
        have_alpha = false;
        // D = sqrt(clamp(diag(J'^T J'))), g/D
        if (t < NF) lds[L_DD + t] = sqrt(fmin(fmax(lds[L_S + roff(t) + t], o.min_lm_diagonal), o.max_lm_diagonal));
        if (t >= 192 && t < 192 + c.nf) lds[L_DD + NF + t - 192] = sqrt(fmin(fmax(lds[L_HEE + t - 192], o.min_lm_diagonal), o.max_lm_diagonal));
        __syncthreads();
        double g2 = 0;
        for (int i = t; i < NF + c.nf; i += NT) {
          const double v = lds[L_G + i] / lds[L_DD + i];
          lds[L_DG + i] = v;
          g2 += v * v;
        }
        gnorm = sqrt(block_sum<NT>(g2, lds + L_RED));
        // Gauss-Newton step with mu retry (DoglegStrategy::ComputeGaussNewtonStep)
        solver_ok = false;
        bool rebuilt = true;
        while (mu < max_mu) {
          if (!rebuilt) {  // S was destroyed by a failed factorisation: rebuild the normal equations
            evaluate_x();
            rebuilt = true;
          }
          PROF_T0();
          // Schur complement on the inverse depths: S_pp -= W^T (hee + mu D_e^2)^-1 W ; rhs
          if (t < NF) lds[L_S + roff(t) + t] += mu * lds[L_DD + t] * lds[L_DD + t];
          for (int i = t; i < NF; i += NT) lds[L_Y + i] = lds[L_G + i];
          __syncthreads();
          const double* W = c.sc + Scratch::W;
          // 1/(hee + mu D_e^2) per feature (L_ST is dead here)
          if (t < c.nf) lds[L_ST + t] = 1.0 / (lds[L_HEE + t] + mu * lds[L_DD + NF + t] * lds[L_DD + NF + t]);
          double accS[5];
          int ei[5], ej[5];
          for (int q = 0; q < 5; q++) {
            accS[q] = 0;
            const int idx = t + q * NT;
            int i = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
            while ((i + 1) * (i + 2) / 2 <= idx) i++;
            while (i * (i + 1) / 2 > idx) i--;
            ei[q] = idx < NPOSE * (NPOSE + 1) / 2 ? i : -1;
            ej[q] = idx - i * (i + 1) / 2;
          }
          double accR = 0;
          for (int e0 = 0; e0 < c.nf; e0 += WCH) {
            const int ne = min(WCH, c.nf - e0);
            __syncthreads();
            for (int idx = t; idx < ne * NPOSE; idx += NT) lds[L_WCH + idx] = W[(size_t)e0 * NPOSE + idx];
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 5; q++) {
              if (ei[q] >= 0) {
                double s = 0;
                for (int e = 0; e < ne; e++) s += (lds[L_WCH + e * NPOSE + ei[q]] * lds[L_ST + e0 + e]) * lds[L_WCH + e * NPOSE + ej[q]];
                accS[q] += s;
              }
            }
            if (t < NPOSE) {
              double s = 0;
              for (int e = 0; e < ne; e++) s += (lds[L_WCH + e * NPOSE + t] * lds[L_ST + e0 + e]) * lds[L_G + NF + e0 + e];
              accR += s;
            }
          }
#pragma unroll
          for (int q = 0; q < 5; q++)
            if (ei[q] >= 0) lds[L_S + roff(ei[q]) + ej[q]] -= accS[q];
          if (t < NPOSE) lds[L_Y + t] -= accR;
          __syncthreads();
          PROF(c, 11);
          const bool ok = cholesky_lds(lds);
          PROF(c, 12);
          if (!ok) {
            mu *= mu_inc;
            rebuilt = false;
            continue;
          }
          chol_solve_lds(lds, L_Y);
          PROF(c, 13);
          // back substitution y_e = (g_e - W_e y_p) / (hee + mu D_e^2): one wavefront per feature
          {
            const int lane = t & 63, wv = t >> 6;
            for (int e = wv; e < c.nf; e += NT / 64) {
              double s = W[(size_t)e * NPOSE + lane] * lds[L_Y + lane];
              if (lane < NPOSE - 64) s += W[(size_t)e * NPOSE + 64 + lane] * lds[L_Y + 64 + lane];
              s = wave_sum(s);
              if (lane == 0) {
                const double he = lds[L_HEE + e] + mu * lds[L_DD + NF + e] * lds[L_DD + NF + e];
                lds[L_Y + NF + e] = (lds[L_G + NF + e] - s) / he;
              }
            }
          }
          __syncthreads();
          PROF(c, 14);
          double bad = 0;
          for (int i = t; i < NF + c.nf; i += NT)
            if (!isfinite(lds[L_Y + i])) bad = 1;
          if (block_max<NT>(bad, lds + L_RED) > 0) {
            mu *= mu_inc;
            rebuilt = false;
            continue;
          }
          solver_ok = true;
          break;
        }
        if (solver_ok) {
          double a1 = 0, a2 = 0;
          for (int i = t; i < NF + c.nf; i += NT) {
            const double yv = lds[L_Y + i], dv = lds[L_DD + i] * yv;
            a1 += dv * dv;
            a2 += yv * lds[L_G + i];
          }
          gn_norm = sqrt(block_sum<NT>(a1, lds + L_RED));
          ytg = block_sum<NT>(a2, lds + L_RED);
        }
      }
      bool step_is_valid = false;
      double model_cost_change = 0;
      if (solver_ok) {
        // ComputeTraditionalDoglegStep
        if (gn_norm <= radius) {
          k1 = 0, k2 = 1;
          dogleg_step_norm = gn_norm;
        } else {
          if (!have_alpha) {  // Cauchy point, needed only when the GN step leaves the trust region
            for (int i = t; i < VEC; i += NT) lds[L_ST + i] = i < NF + c.nf ? lds[L_DG + i] / lds[L_DD + i] : 0.0;
            __syncthreads();
            jusq = jac_times_vec_sq(c, o);
            alpha = gnorm * gnorm / jusq;
            have_alpha = true;
            __syncthreads();
          }
          if (gnorm * alpha >= radius) {
            k1 = radius / gnorm, k2 = 0;
            dogleg_step_norm = radius;
          } else {
            // a = -alpha g/D, b = -D y
            const double b_dot_a = alpha * ytg;  // (-alpha g/D).(-D y) = alpha g^T y
            const double a2n = (alpha * gnorm) * (alpha * gnorm);
            const double bma = a2n - 2 * b_dot_a + gn_norm * gn_norm;
            const double cc = b_dot_a - a2n;
            const double dd = sqrt(cc * cc + bma * (radius * radius - a2n));
            const double beta = (cc <= 0) ? (dd - cc) / bma : (radius * radius - a2n) / (dd + cc);
            k1 = alpha * (1.0 - beta), k2 = beta;
            double s2 = 0;
            for (int i = t; i < NF + c.nf; i += NT) {
              const double v = -k1 * lds[L_DG + i] - k2 * lds[L_DD + i] * lds[L_Y + i];
              s2 += v * v;
            }
            dogleg_step_norm = sqrt(block_sum<NT>(s2, lds + L_RED));
          }
        }
        for (int i = t; i < VEC; i += NT)
          lds[L_ST + i] = i < NF + c.nf ? -(k1 * lds[L_DG + i] / lds[L_DD + i] + k2 * lds[L_Y + i]) : 0.0;
        // model_cost_change = -step^T g - 1/2 step^T H step, with H y = g - mu D^2 y
        {
          const double utg = gnorm * gnorm;                     // u^T g, u = g/D^2
          const double yDy = gn_norm * gn_norm;                 // y^T D^2 y
          const double uHy = utg - mu * ytg;                    // u^T (g - mu D^2 y)
          const double yHy = ytg - mu * yDy;
          const double sHs = k1 * k1 * (k1 != 0 ? jusq : 0.0) + 2 * k1 * k2 * uHy + k2 * k2 * yHy;
          model_cost_change = (k1 * utg + k2 * ytg) - 0.5 * sHs;
        }
        step_is_valid = model_cost_change > 0.0;
        if (step_is_valid) num_invalid = 0;
        __syncthreads();
      }
      if (!step_is_valid) {
        if (++num_invalid >= o.max_num_consecutive_invalid_steps) {
          termination = AVM_TERM_FAILURE;
          break;
        }
        mu *= mu_inc;  // StepIsInvalid
        reuse = false;
        evaluate_x();  // S holds a Cholesky factor: rebuild the normal equations for the retry
        continue;
      }
      // candidate
      PROF_T0();
      state_plus(lds);
      __syncthreads();
      build_frames(lds, lds + L_XC, 1);
      __syncthreads();
      const double cand_cost = eval_cost(c, o, lds + L_XC, 1);
      PROF(c, 15);
      double d2 = 0;
      for (int i = t; i < 176 + c.nf; i += NT) {
        const double d = lds[L_X + i] - lds[L_XC + i];
        d2 += d * d;
      }
      const double step_norm = sqrt(block_sum<NT>(d2, lds + L_RED));
      if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
        termination = AVM_TERM_PARAMETER_TOL;
        break;
      }
      const double cost_change = x_cost - cand_cost;
      if (fabs(cost_change) <= o.function_tolerance * x_cost) {
        termination = AVM_TERM_FUNCTION_TOL;
        break;
      }
      const double rel = (ref_cost - cand_cost) / model_cost_change;
      if (rel > o.min_relative_decrease) {
        __syncthreads();
        for (int i = t; i < XN; i += NT) lds[L_X + i] = lds[L_XC + i];
        __syncthreads();
        x_norm = amb_norm(lds + L_X);
        evaluate_x();
        step_ok = true;
        if (rel < 0.25) radius *= 0.5;
        if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
        mu = fmax(min_mu, 2.0 * mu / mu_inc);
        reuse = false;
        ref_cost = cand_cost;
      } else {
        radius *= 0.5;
        reuse = true;
      }
    }
    __syncthreads();
    // ---------------- double2vector + vector2double (estimator.cpp:521-587, 477-519) ----------------
    {
      // rot_diff from yaw of frame 0 before / after ; stored in lds[L_DG..+9], origin_P0 in +9..12
      if (t == 0) {
        const double* p0 = B.pose + (size_t)w * 77;
        double Rs0[9], R00[9];
        q2R(quat{p0[6], p0[3], p0[4], p0[5]}, Rs0);
        q2R(quat{lds[L_X + 6], lds[L_X + 3], lds[L_X + 4], lds[L_X + 5]}, R00);
        auto ypr = [](const double* R, double* out) {
          const double y = atan2(R[3], R[0]);
          const double p = atan2(-R[6], R[0] * cos(y) + R[3] * sin(y));
          const double r = atan2(R[2] * sin(y) - R[5] * cos(y), -R[1] * sin(y) + R[4] * cos(y));
          out[0] = y / M_PI * 180.0, out[1] = p / M_PI * 180.0, out[2] = r / M_PI * 180.0;
        };
        double a0[3], a1[3];
        ypr(Rs0, a0);
        ypr(R00, a1);
        const double yd = (a0[0] - a1[0]) / 180.0 * M_PI;
        double rd[9] = {cos(yd), -sin(yd), 0, sin(yd), cos(yd), 0, 0, 0, 1};
        if (fabs(fabs(a0[1]) - 90) < 1.0 || fabs(fabs(a1[1]) - 90) < 1.0) {
          for (int a = 0; a < 3; a++)
            for (int b = 0; b < 3; b++) rd[a * 3 + b] = Rs0[a * 3] * R00[b * 3] + Rs0[a * 3 + 1] * R00[b * 3 + 1] + Rs0[a * 3 + 2] * R00[b * 3 + 2];
        }
        for (int k = 0; k < 9; k++) lds[L_DG + k] = rd[k];
        for (int k = 0; k < 3; k++) lds[L_DG + 9 + k] = p0[k], lds[L_DG + 12 + k] = lds[L_X + k];
      }
      __syncthreads();
      if (t < NFR) {
        const double* rd = lds + L_DG;
        const double* x = lds + L_X;
        quat q = qnormalized(quat{x[t * 7 + 6], x[t * 7 + 3], x[t * 7 + 4], x[t * 7 + 5]});
        double Rq[9], Rs[9];
        q2R(q, Rq);
        mat3mul(rd, Rq, Rs);
        const v3 P = Rmul(rd, mk3(x[t * 7] - lds[L_DG + 12], x[t * 7 + 1] - lds[L_DG + 13], x[t * 7 + 2] - lds[L_DG + 14])) +
                     mk3(lds[L_DG + 9], lds[L_DG + 10], lds[L_DG + 11]);
        const v3 V = Rmul(rd, mk3(x[XSB + t * 9], x[XSB + t * 9 + 1], x[XSB + t * 9 + 2]));
        const quat qo = R2q(Rs);
        double* po = B.pose + (size_t)w * 77 + t * 7;
        po[0] = P.x, po[1] = P.y, po[2] = P.z, po[3] = qo.x, po[4] = qo.y, po[5] = qo.z, po[6] = qo.w;
        double* so = B.speedbias + (size_t)w * 99 + t * 9;
        so[0] = V.x, so[1] = V.y, so[2] = V.z;
        for (int k = 3; k < 9; k++) so[k] = x[XSB + t * 9 + k];
      }
      if (t == 64) {
        double* ex = B.ex_pose + (size_t)w * 7;
        double R[9];
        q2R(quat{ex[6], ex[3], ex[4], ex[5]}, R);
        const quat qo = R2q(R);
        ex[3] = qo.x, ex[4] = qo.y, ex[5] = qo.z, ex[6] = qo.w;
      }
      if (t >= 128 && t < 128 + c.nf) {
        const int e = t - 128;
        B.inv_depth[(size_t)w * B.max_feat + e] = 1.0 / (1.0 / lds[L_X + XLAM + e]);
      }
    }
    if (c.prof && t == 0) c.prof[31] += 1;
    if (t == 0 && A.summary) {
      avm_solve_summary* so = A.summary + w;
      so->termination = termination;
      so->num_iterations = iteration;
      so->num_successful = n_successful;
      so->accept_mask = accept_mask;
      so->initial_cost = initial_cost;
      so->final_cost = x_cost;
      for (int k = 0; k < AVM_MAX_ITER_TRACE; k++) so->cost_trace[k] = lds[L_SUM + k], so->radius_trace[k] = lds[L_SUM + 16 + k];
    }
    __syncthreads();
  }
}

// Per-factor evaluation at the input state (no solve): parity-test surface for A5/A6/A8.
__global__ __launch_bounds__(NT) void eval_factors_kernel(EvalArgs A) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  double* lds = reinterpret_cast<double*>(smem_raw);
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int t = threadIdx.x;
  const avm_options& o = A.opt;
  const avm_window_batch& B = A.b;
  const int w = blockIdx.x;
  WinCtx c;
  c.lds = lds, c.ids = ids, c.sc = nullptr, c.osf = nullptr, c.w = w;
  c.nf = B.n_feat[w];
  c.obs = B.obs_xy + (size_t)w * B.max_obs * 2;
  c.pdelta = A.pre_delta + (size_t)w * 100, c.pjac = A.pre_jac + (size_t)w * 2250, c.psqrt = A.pre_sqrt + (size_t)w * 2250;
  c.psum = A.pre_sum_dt + (size_t)w * 10;
  c.lba = B.imu_lin_ba + (size_t)w * 30, c.lbg = B.imu_lin_bg + (size_t)w * 30;
  c.pn = B.prior_n ? B.prior_n[w] : 0;
  c.pnblk = c.pn > 0 ? B.prior_nblk[w] : 0;
  c.ldp = B.max_prior;
  c.pJ = B.prior_J + (size_t)w * B.max_prior * B.max_prior;
  c.pr = B.prior_r + (size_t)w * B.max_prior;
  c.px0 = B.prior_x0 + (size_t)w * B.max_pblk * 9;
  for (int i = t; i < 77; i += NT) lds[L_X + i] = B.pose[(size_t)w * 77 + i];
  for (int i = t; i < 99; i += NT) lds[L_X + XSB + i] = B.speedbias[(size_t)w * 99 + i];
  for (int i = t; i < MAXE; i += NT) lds[L_X + XLAM + i] = i < c.nf ? B.inv_depth[(size_t)w * B.max_feat + i] : 1.0;
  for (int i = t; i < MAXPRIOR; i += NT) lds[L_DXP + i] = 0.0, lds[L_RP + i] = 0.0;
  for (int i = t; i < 10 * 465; i += NT) lds[L_S + i] = 0.0;
  if (t < 7) lds[L_RIC + 12 + t] = B.ex_pose[(size_t)w * 7 + t];
  if (t == 0) {
    const double* ex = B.ex_pose + (size_t)w * 7;
    double R[9];
    q2R(quat{ex[6], ex[3], ex[4], ex[5]}, R);
    for (int k = 0; k < 9; k++) lds[L_RIC + k] = R[k];
    for (int k = 0; k < 3; k++) lds[L_RIC + 9 + k] = ex[k];
    int off = 0;
    for (int k = 0; k < c.pnblk; k++) {
      const int kind = B.prior_blk_kind[(size_t)w * B.max_pblk + k], fr = B.prior_blk_frame[(size_t)w * B.max_pblk + k];
      ids[I_PBLK + k * 3] = kind, ids[I_PBLK + k * 3 + 1] = fr, ids[I_PBLK + k * 3 + 2] = off;
      off += kind == AVM_BLK_SPEEDBIAS ? 9 : 6;
    }
  }
  __syncthreads();
  build_frames(lds, lds + L_X, 0);
  __syncthreads();
  Frames fr{lds + L_FR, lds + L_FR + 99};
  const double sqi = o.focal_length / 1.5;
  double acc = 0;
  if (t >= NT - 64 && t < NT - 64 + 10) {
    const int i = t - (NT - 64);
    imu_raw<true>(lds + L_X, fr.R, o, c.pdelta + i * 10, c.pjac + i * 225, c.psum[i], c.lba + i * 3, c.lbg + i * 3, i, lds + L_S + i * 465);
  }
  for (int e = 0; e < c.nf; e++) {  // thread per observation of feature e
    const int s0 = B.feat_obs_begin[(size_t)w * B.max_feat + e], no = B.feat_nobs[(size_t)w * B.max_feat + e];
    const int fa = B.feat_start[(size_t)w * B.max_feat + e];
    for (int k = 1 + t; k < no; k += NT) {
      const int s = s0 + k;
      double r[2], Ji[12], Jj[12], Je[2];
      acc += proj_eval<true>(lds + L_X, fr, lds + L_RIC, lds + L_RIC + 9, c.obs[2 * s0], c.obs[2 * s0 + 1], c.obs[2 * s], c.obs[2 * s + 1],
                             lds[L_X + XLAM + e], fa, fa + k, sqi, o.cauchy_a, A.apply_loss != 0, r, Ji, Jj, Je);
      const size_t ob = (size_t)w * B.max_obs + s;
      if (A.proj_r) A.proj_r[ob * 2] = r[0], A.proj_r[ob * 2 + 1] = r[1];
      if (A.proj_J)
        for (int rr = 0; rr < 2; rr++) {
          for (int q = 0; q < 6; q++) A.proj_J[ob * 26 + rr * 13 + q] = Ji[rr * 6 + q], A.proj_J[ob * 26 + rr * 13 + 6 + q] = Jj[rr * 6 + q];
          A.proj_J[ob * 26 + rr * 13 + 12] = Je[rr];
        }
    }
  }
  __syncthreads();
  for (int idx = t; idx < 10 * 465; idx += NT) {
    const int i = idx / 465, rc = idx % 465, r = rc / 31, cc = rc % 31;
    double s = 0;
    for (int k = r; k < 15; k++) s += c.psqrt[i * 225 + r * 15 + k] * lds[L_S + i * 465 + k * 31 + cc];
    const size_t iv = (size_t)w * 10 + i;
    if (cc == 0) {
      if (A.imu_r) A.imu_r[iv * 15 + r] = s;
      if (c.psum[i] <= o.max_sum_dt) acc += 0.5 * s * s;
    } else if (A.imu_J) {
      A.imu_J[(iv * 15 + r) * 30 + cc - 1] = s;
    }
  }
  if (c.pn > 0) {
    prior_residual_dev(c, lds + L_X);
    if (t < c.pn) {
      acc += 0.5 * lds[L_RP + t] * lds[L_RP + t];
      if (A.prior_res) A.prior_res[(size_t)w * B.max_prior + t] = lds[L_RP + t];
    }
  }
  const double cost = block_sum<NT>(acc, lds + L_RED);
  if (t == 0 && A.cost) A.cost[w] = cost;
}

int window_solve_lds_bytes() { return L_END * 8; }

hipError_t launch_window_solve(const SolveArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_solve_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = a.b.n_windows < a.n_slots ? a.b.n_windows : a.n_slots;
  hipLaunchKernelGGL(window_solve_kernel, dim3(grid), dim3(NT), L_END * 8, stream, a);
  return hipGetLastError();
}


hipError_t launch_eval_factors(const EvalArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(eval_factors_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(eval_factors_kernel, dim3(a.b.n_windows), dim3(NT), L_END * 8, stream, a);
  return hipGetLastError();
}

}  // namespace avm
