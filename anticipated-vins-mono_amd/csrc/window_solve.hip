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
#include <utility>

#include "devmath.hpp"
#include "kernels.hpp"

namespace avm {

extern __shared__ __attribute__((aligned(16))) char avm_smem[];

namespace {

// the workgroup's dynamic LDS, always reached through the shared symbol (never through a generic pointer that
// crosses a function boundary), so the compiler keeps ds_* addressing inside outlined functions
//
// None of the kernels in this file has static LDS, so the dynamic segment starts at LDS address 0 (checked once per
// workgroup by lds_base_check()).  Spelling the base as the constant instead of the symbol matters: outside a
// kernel body the symbol's address is a load from llvm.amdgcn.dynlds.offset.table, which the compiler happily
// re-issues (s_load + s_waitcnt) in front of every predicated LDS access of the outlined phases.
typedef __attribute__((address_space(3))) double lds_double_t;
typedef __attribute__((address_space(3))) char lds_char_t;
AVM_DEV double* LDS() {
  // (integer -> pointer keeps this the LDS address 0; a literal null would become the address-space's null, -1)
  return (double*)reinterpret_cast<lds_double_t*>((uintptr_t)__builtin_amdgcn_readfirstlane(0));
}
AVM_DEV void lds_base_check() {
  if ((unsigned)(uintptr_t)(lds_char_t*)avm_smem != 0u) __builtin_trap();
}

#define AVM_NOINL __device__ __noinline__
#define PROF_T0() long long pt__ = clock64()
#define PROF(c, k) do { if ((c).prof && threadIdx.x == 0) { long long n__ = clock64(); (c).prof[k] += n__ - pt__; pt__ = n__; } } while (0)
// second, independent stopwatch for the trust-region loop's own segments (slots 32..)
#define PROFQ_T0() pq__ = clock64()
#define PROFQ(c, k) do { if ((c).prof && threadIdx.x == 0) { long long n__ = clock64(); (c).prof[k] += n__ - pq__; pq__ = n__; } } while (0)

#ifdef AVM_TP
// THROUGHPUT build (window_solve_tp.o, -DAVM_TP): the same minimizer as a 256-thread workgroup (four wavefronts, one per SIMD) with at most
// 80 KB of LDS, so that TWO windows are resident per CU and the dependent chains of one overlap the other's.  What makes it fit:
//   * only the dense pose-pose rows of S (66 packed rows, 18 KB) stay in LDS as they are; the speed-bias rows are kept in their
//     structural form (per 9-row block the 18 pose and 18 speed-bias columns an IMU factor can reach, plus the prior's speed-bias x pose
//     strip), in the range the frame tasks' staging occupies during phase A;
//   * the factorization runs on REGISTER tiles distributed over the four wavefronts (chol_regs below), fed from those two forms;
//   * the frame tasks stage half a chunk (32 factors) at a time.
constexpr int NT = 256;
#else
constexpr int NT = 512;          // threads per workgroup (8 wavefronts)
#endif
constexpr int croff(int i) { return 2 * ((i >> 1) + 1) * ((i >> 1) + (i & 1)); }  // roff() at compile time
constexpr int SROWS = croff(NF + 1);  // padded packed lower triangle of the NF x NF matrix + one augmented row (the RHS): 13944 (16200)
constexpr int VEC = (NCOL + 7) & ~7;  // padded NCOL: 320 (328)
constexpr int XSB = 7 * NFRP, XLAM = XSB + 99;  // state vector: poses (relo_Pose as frame 11) | speedbias 99 | inv depth 150 [| ex_pose 7 | td]
#ifdef AVM_X
constexpr int XEX = XLAM + MAXE, XTD = XEX + 7;
constexpr int XN = (XTD + 2) & ~1;  // 342
#else
constexpr int XN = XLAM + MAXE + 2;  // 328
#endif

// LDS carve (offsets in doubles).  Everything between L_S + SPP (end of the pose-pose rows of S) and
// L_G is dead while the projection factors are being assembled, so that range doubles as the per-wave
// staging area of the MFMA X^T X products (ASM_WAVES x XSTG doubles).
constexpr int SPP = croff(NPOSE);  // packed rows 0..NPOSE-1 = the dense pose(-like) block: 2244 (3200)
constexpr int WLD = 80;
constexpr int XRS = 132;           // column stride of the frame tasks' column-major staging tile: 128 rows + 4 (bank spread)
#ifdef AVM_X
constexpr int WCH = 8;             // rows of the scratch tile at L_WCH (x 80 columns): diag-block temporaries, back-substitution vector
constexpr int XCOLS = 20;          // staged factor row: Jj(6) | Ji(6) | r | Jex(6) | Jtd
constexpr int XRS_X = 68;          // column stride of the HALF-chunk staging tile (round 5): 32 factors x 2 residual rows + 4 (bank spread)
constexpr int XSTG = XCOLS * XRS_X;
constexpr int ASM_WAVES = 7;       // wavefronts assembling projection factors (round 5: seven half-chunk tiles fit where five whole ones did; eleven
                                   // frames deal 2 2 2 2 1 1 1 instead of 3 2 2 2 2, and wavefront 7 takes the raw IMU Jacobians AND the prior)
#elif defined(AVM_TP)
constexpr int XRS_H = 68;          // column stride of the HALF-chunk staging tile: 32 factors x 2 residual rows + 4 (bank spread)
constexpr int XSTG = 13 * XRS_H;   // 884
constexpr int ASM_WAVES = 4;       // every wavefront assembles; wavefront 2 then takes the raw IMU Jacobians, wavefront 3 the prior
#else
constexpr int WCH = 32;
constexpr int XLD = 14;            // staged factor row: Jj(6) | Ji(6) | r (+1 pad)
constexpr int XSTG = 128 * XLD;    // 64 factors x 2 residual rows (>= 13 * XRS)
static_assert(13 * XRS <= XSTG, "column-major staging tile fits");
constexpr int ASM_WAVES = 6;       // wavefronts assembling projection factors (wavefront 6: the raw IMU Jacobians, 7: the prior)
#endif
#ifndef AVM_LPT_RUNW
#define AVM_LPT_RUNW 16
#endif
// positions of chol_regs' elimination order (see there): [0, 48) B, [48, 96) F, frame 5's speed-bias block, the dense columns, the right-hand side
constexpr int TP_M0 = 96, TP_P0 = 105, TP_RHS = TP_P0 + NPOSE;  // 171 (184)
constexpr int TPT = TP_RHS / 16 + 1;                            // tile columns: 11 (12)
constexpr int TP_NPOS = 16 * TPT;                               // 176 (192)
#ifdef AVM_X
constexpr int TP_NWO = 8;          // wavefronts that hold tiles of the factorization
#else
constexpr int TP_NWO = 4;
#endif
constexpr int CNB = 16;            // Cholesky panel width (pivot chain per diagonal block); trailing tiles stay 16x16
constexpr int TLAST = NF / 16;     // last 16-row tile of the packed matrix incl. the augmented row NF: 10 (11)
constexpr int FRS = 18 * NFRP;     // one frames slot: R (NFRP x 9) then A = ric^T R^T (NFRP x 9)
constexpr int L_S = 0;
#ifdef AVM_TP
// rows 0..65 of S packed as in the other builds, then the union region U: phase A: 4 staging tiles; from phase D on: the speed-bias
// rows in structural form + the prior's strip; during the factorization: diagonal patch, L_kk^-T (two buffers), the published row of W
constexpr int SBW = 36;                       // compact speed-bias row: 18 pose columns (poses i-1, i, i+1) | 18 speed-bias columns (i-1, i)
constexpr int L_U = L_S + SPP;
constexpr int L_SBC = L_U;                    // [99][SBW]
constexpr int L_STRIP = L_SBC + 99 * SBW;     // [9][66]: rows of the prior's speed-bias block x every pose column
constexpr int USZ = 99 * SBW + 9 * NPOSE + 2; // 4160; its last two doubles hold the constants 0.0 and 1.0 for chol_regs' tile load (set by schur_reduce)
constexpr int L_ZERO = L_U + USZ - 2, L_ONE = L_U + USZ - 1;
static_assert(ASM_WAVES * XSTG <= USZ, "staging fits the union region");
constexpr int TP_PS = 17;                     // row stride of the 16 x 16 blocks below: lane = row accesses of a stride-16 block put sixteen lanes on two LDS banks
constexpr int L_PATCH = L_U;                  // factorization: [2][16][TP_PS] diagonal blocks of the (up to two) pivot columns of a step in lane = row form
constexpr int L_LINV = L_PATCH + 2 * 16 * TP_PS;  // [4][16][TP_PS]: L_kk^-T (unscaled, see chol_diag_block), 1 / sqrt(pivot) of column r in the padding word of row r (tp_buf)
constexpr int TP_WSLOTS = 9;                  // tiles of a step's rows of W that exist beside the diagonal (tp_wslot: the factor is sparse in the order chol_regs eliminates in)
constexpr int L_WROW = L_LINV + 4 * 16 * TP_PS;      // [TP_WSLOTS][256]: the step's rows of W, the tiles that exist in column order, in the accumulator layout [r][lane]
constexpr int L_PARTV = L_WROW;               // back substitution (the rows of W are dead by then): [4][176] partial sums of the four wavefronts
constexpr int L_ZV = L_WROW + TP_WSLOTS * 256;  // [176] z = L^-1 b, then x, in elimination order (lds[L_Y] keeps the right-hand side until x replaces it, in the system's order)
static_assert(L_ZV + TP_NPOS <= L_ZERO && TP_NWO * TP_NPOS <= TP_WSLOTS * 256, "factorization scratch fits the union region");
static_assert(ASM_WAVES * XSTG <= USZ - 2, "staging leaves the two constants alone");
constexpr int L_Y = L_U + USZ;                // Gauss-Newton solution y; until the solve writes it: the right-hand side (row NF of the other builds)
constexpr int L_RHS = L_Y;
constexpr int WCH_TP = 224;                   // doubles: ys of back_substitute / rvb of jac_times_vec_sq (<= 150), then 64 dump slots
constexpr int L_ST = L_Y + VEC;
constexpr int L_XC = L_ST + VEC;
constexpr int L_WCH = L_XC + XN;
constexpr int L_DUMP = L_WCH + 160;
constexpr int L_G = L_WCH + WCH_TP;
constexpr int L_DD = L_G + VEC;               // (g / D is recomputed where it is needed, as in the extended build)
// Per-wavefront accumulators of the two per-feature sums that end in LDS anyway (E^T E -> lds[L_HEE], E^T r -> lds[L_G + NF]), over the range the
// Gauss-Newton step, the dogleg step and the candidate state occupy between evaluations (all three are dead or parked while eval_jac runs: the
// minimizer recomputes them, and a speculative evaluation parks y in the slot).  Wavefront 0 accumulates in the destinations themselves, wavefronts
// 1..3 in [3][2][152] here; the per-feature sums add the four in a fixed order.  Round 5: every 8 bytes per factor the frame tasks write to the slot
// cost 0.1 ms per 4096 windows (profiles/r05e_experiments.md section 10); these two were sixteen of them.
constexpr int L_ACC = L_Y, ACCW = 152;
static_assert(L_ACC + 6 * ACCW <= L_WCH, "the accumulators stay inside y | step | candidate state");
#else
constexpr int L_Y = L_S + SROWS;   // Gauss-Newton solution y of (H + mu D^2) y = g
constexpr int L_ST = L_Y + VEC;    // trust region step (scaled space)
constexpr int L_XC = L_ST + VEC;   // candidate state
constexpr int L_WCH = L_XC + XN;   // [WCH][80] scratch tile
constexpr int L_DUMP = L_WCH + 512;     // per-lane dump slots of the masked-out stores
constexpr int L_G = L_WCH + WCH * WLD;  // scaled gradient g (f | e)
#endif
#ifndef AVM_TP
// Latency and extended builds: the factorization on register tiles (chol_regs, the throughput build's; latency: on wavefronts 0..3, extended: on all
// eight) reads the packed system once; from then on the range of S is its scratch - same carve as the throughput build's union region, and the
// right-hand side is row NF of S.
constexpr int L_RHS = L_S + croff(NF);
constexpr int TP_PS = 17;
constexpr int L_PATCH = L_S;
constexpr int L_LINV = L_PATCH + 2 * 16 * TP_PS;
#ifdef AVM_X
constexpr int TP_WSLOTS = 14;
#else
constexpr int TP_WSLOTS = 9;
#endif
constexpr int L_WROW = L_LINV + 4 * 16 * TP_PS;
constexpr int L_PARTV = L_WROW;
constexpr int L_ZV = L_WROW + TP_WSLOTS * 256;
static_assert(L_ZV + TP_NPOS <= L_S + SROWS && TP_NWO * TP_NPOS <= TP_WSLOTS * 256, "factorization scratch fits the range of S");
#endif
#ifdef AVM_TP
#elif defined(AVM_X)
constexpr int L_DD = L_G + VEC;    // D   (g / D is recomputed where it is needed: no room for a fourth vector next to the 178 x 178 system)
#else
constexpr int L_DG = L_G + VEC;    // g / D
constexpr int L_DD = L_DG + VEC;   // D
#endif
constexpr int L_SC = L_DD + VEC;   // Jacobi scaling
constexpr int L_X = L_SC + VEC;
constexpr int L_FR = L_X + XN;     // [2][FRS]
#ifdef AVM_X
constexpr int L_RIC = L_FR + 2 * FRS;  // [2][12]: ric 9, tic 3 of the current point / of the candidate
constexpr int L_HEE = L_RIC + 24;      // E^T E (150) padded
#else
constexpr int L_RIC = L_FR + 2 * FRS;  // ric 9, tic 3, current ex_pose 7 (+1 pad)
constexpr int L_HEE = L_RIC + 20;      // E^T E (150) padded
#endif
constexpr int L_DXP = L_HEE + 152;
constexpr int L_RP = L_DXP + MAXPRIOR;
#ifdef AVM_X
constexpr int L_DX2 = L_DXP;       // (the extended build keeps the prior on one wavefront)
constexpr int L_RED = L_RP + MAXPRIOR;
#else
constexpr int L_DX2 = L_RP + MAXPRIOR;  // dx / J0^T r_p of the second wavefront that shares the prior
constexpr int L_RED = L_DX2 + MAXPRIOR;
#endif
constexpr int L_RED_B = L_RED + 16, L_RED_CNT = L_RED + 32;  // second value of a paired reduction; the wavefronts' reduction counters (8 ints)
constexpr int L_INT = L_RED + 36;  // int region (as doubles): 360 doubles = 720 ints
constexpr int L_SUM = L_INT + 360;  // cost_trace[16], radius_trace[16]
constexpr int L_CTX = L_SUM + 32;   // WinCtx of the window being solved (32 doubles)
constexpr int L_OPT = L_CTX + 32;   // avm_options (copied from the kernel arguments)
constexpr int L_END = L_OPT + (int)((sizeof(avm_options) + 7) / 8);
#ifdef AVM_TP
static_assert(L_END * 8 <= 81920, "two workgroups per CU: 80 KB each");
#else
static_assert(L_END * 8 <= 163840, "LDS budget exceeded");
#endif
static_assert(L_S + SPP + ASM_WAVES * XSTG <= L_G, "assembly staging overlaps live data");
// int carve (offsets in ints from L_INT)
constexpr int I_FSTART = 0, I_FNOBS = 150, I_FOBS = 300, I_PIDX = 450, I_FS = 546, I_PBLK = 560 /* kind,frame,off x16 */, I_FAIL = 620,
              I_NCOV = 624 /* [12] factors observed in frame b */, I_FRW = 636 /* [12] assembling wave of frame b */,
              I_PMASK = 648 /* [12] start frames flushed by frame b */, I_TIMEUP = 660 /* max_solver_time reached (set by thread 0) */,
              I_NRUN = 661 /* [12] distinct start frames among the factors observed in frame b */,
              I_PSB = 673 /* throughput build: frame of the prior's speed-bias block (its rows x every pose column: the strip) */,
              I_CNT = 674 /* wavefronts x rows of W published so far in this factorization (chol_regs) */,
              I_CRFIT = 675 /* latency build: the window's prior fits the sparse factorization (chol_regs), else cholesky_lds */, I_END = 676;
static_assert(I_END <= 720, "int carve");
#ifndef AVM_TP
constexpr int L_ZERO = L_INT + 340, L_ONE = L_INT + 341;  // the constants 0.0 and 1.0 of chol_regs' tile load, in the unused tail of the int carve (set by schur_reduce)
static_assert(2 * 340 >= I_END, "the constants sit behind the int carve");
#endif
typedef double d4 __attribute__((ext_vector_type(4)));

// Issue priority of the calling wavefront (throughput build only).  Two windows share every SIMD there, one wavefront each: while one
// of them streams MFMAs / factor arithmetic (the frame tasks, the Schur tiles, the trailing updates of the factorization: AVM_PRIO_BULK)
// and the other walks a dependent chain or one of the short barrier-separated vector phases of the trust-region loop (AVM_PRIO_LIGHT),
// the arbiter should hand the next free issue slot to the latter - its instructions are the window's critical path, the bulk work
// fills whatever is left.  (The pivot chains have run at priority 3 since round 4.)
#ifdef AVM_TP
#ifndef AVM_PRIO_L
#define AVM_PRIO_L 2
#endif
#ifndef AVM_PRIO_CHOL
#define AVM_PRIO_CHOL 1
#endif
#ifndef AVM_PRIO_SCHUR
#define AVM_PRIO_SCHUR 1
#endif
#define AVM_PRIO_BULK() __builtin_amdgcn_s_setprio(0)
#define AVM_PRIO_BULK_CHOL() __builtin_amdgcn_s_setprio(AVM_PRIO_CHOL)
#define AVM_PRIO_BULK_SCHUR() __builtin_amdgcn_s_setprio(AVM_PRIO_SCHUR)
#define AVM_PRIO_LIGHT() __builtin_amdgcn_s_setprio(AVM_PRIO_L)
#else
#define AVM_PRIO_BULK() ((void)0)
#define AVM_PRIO_BULK_CHOL() ((void)0)
#define AVM_PRIO_BULK_SCHUR() ((void)0)
#define AVM_PRIO_LIGHT() ((void)0)
#endif

AVM_DEV void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Workgroup reductions of the solve kernel with ONE barrier each.  The partial sums of consecutive reductions go to alternating
// halves of lds[L_RED]: a wavefront can only overwrite a half two reductions later, i.e. after a barrier that every wavefront
// reaches with its reads of that half done.  Which half is next is a counter every wavefront keeps for itself in LDS (all
// wavefronts run the same sequence of reductions, so the counters agree); red_init() zeroes it at kernel entry.
AVM_DEV int* red_counter() { return reinterpret_cast<int*>(LDS() + L_RED_CNT) + (threadIdx.x >> 6); }
AVM_DEV void red_init() {
  if ((threadIdx.x & 63) == 0) *red_counter() = 0;
}
template <class Op>
AVM_DEV double block_reduce1(double v, Op op) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int* cnt = red_counter();
  const int k = *cnt;
  double* red = LDS() + L_RED + 8 * (k & 1);
  if (lane == 0) red[wv] = v, *cnt = k + 1;
  __syncthreads();
  double s = red[0];
#pragma unroll
  for (int i = 1; i < NT / 64; i++) s = op(s, red[i]);
  return s;
}
AVM_DEV double block_sum1(double v) {
  return block_reduce1(wave_sum(v), [](double a, double b) { return a + b; });
}
AVM_DEV double block_max1(double v) {
  return block_reduce1(wave_max(v), [](double a, double b) { return fmax(a, b); });
}
// two sums at once (one barrier, both halves of the pair in the same half of lds[L_RED])
AVM_DEV void block_sum1x2(double& a, double& b) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double wa = wave_sum(a), wb = wave_sum(b);
  int* cnt = red_counter();
  const int k = *cnt;
  double* red = LDS() + L_RED + 8 * (k & 1);
  double* redb = LDS() + L_RED_B + 8 * (k & 1);
  if (lane == 0) red[wv] = wa, redb[wv] = wb, *cnt = k + 1;
  __syncthreads();
  double sa = red[0], sb = redb[0];
#pragma unroll
  for (int i = 1; i < NT / 64; i++) sa += red[i], sb += redb[i];
  a = sa, b = sb;
}

AVM_DEV int roff(int i) {  // even i = 2q: 2q(q+1); odd i = 2q+1: 2(q+1)^2 -> every row starts 16-byte aligned
  const int q = i >> 1;
  return 2 * __mul24(q + 1, q + (i & 1));  // 24-bit multiply: full rate (v_mul_lo_u32 is quarter rate)
}

#ifdef AVM_TP
// Throughput build: offset (doubles from lds[0]) of entry (r, c), c <= r < NF, of the assembled system, or -1 where the entry is
// structurally zero.  Pose rows: the packed triangle; speed-bias rows: the compact row [poses i-1, i, i+1 | speed-biases i-1, i] of
// block i, except that the pose columns of the prior's speed-bias block live in the strip (the prior couples it to every pose).
AVM_DEV int s_off(int r, int c) {
  if (r < NPOSE) return L_S + roff(r) + c;
  const int q = r - NPOSE, i = q / 9;
  if (c < NPOSE) {
    if (i == reinterpret_cast<const int*>(LDS() + L_INT)[I_PSB]) return L_STRIP + (q - 9 * i) * NPOSE + c;
    const int p = c - 6 * (i - 1);
    return (p >= 0 && p < 18) ? L_SBC + q * SBW + p : -1;
  }
  const int p = c - (NPOSE + 9 * (i - 1));
  return (p >= 0 && p < 18) ? L_SBC + q * SBW + 18 + p : -1;
}
#define S_OFF(r, c) s_off(r, c)
#else
#define S_OFF(r, c) (L_S + roff(r) + (c))
#endif

// reciprocal / reciprocal square root from the hardware estimate + two Newton steps (about one ulp; the library forms spend
// two to three times as long on range handling that the operands here - depths, squared norms >= 1 - never need)
AVM_DEV double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x), e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
// raw v_rsq_f64 + two Newton steps (the library rsqrt spends ~3x as long in range handling we do not need:
// pivots of an SPD matrix are normal positive numbers)
AVM_DEV double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * y * y);
  y = y * (1.5 - (0.5 * x) * y * y);
  return y;
}

AVM_DEV double fast_rsqrt_pe(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * y * y);
  y = y * (1.5 - (0.5 * x) * y * y);
  return y;
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
                         double* Ji, double* Jj, double* Je, double* Jex = nullptr, double* Jtd = nullptr, double vix = 0.0,
                         double viy = 0.0, double vjx = 0.0, double vjy = 0.0) {
  const double* Ra = fr.R + fa * 9;
  const double* Rb = fr.R + fb * 9;
  const v3 Pa = mk3(x[fa * 7], x[fa * 7 + 1], x[fa * 7 + 2]);
  const v3 Pb = mk3(x[fb * 7], x[fb * 7 + 1], x[fb * 7 + 2]);
  const v3 t = mk3(tic[0], tic[1], tic[2]);
  const double il = fast_rcp(lam);
  const v3 pci = mk3(pix * il, piy * il, il);
  const v3 pimu_i = Rmul(ric, pci) + t;
  const v3 pw = Rmul(Ra, pimu_i) + Pa;
  const v3 pimu_j = RTmul(Rb, pw - Pb);
  const v3 pcj = RTmul(ric, pimu_j - t);
  const double dep = pcj.z;
  const double id = fast_rcp(dep);  // one reciprocal per quantity: the quotients below are products with it
  double r0 = sqi * (pcj.x * id - pjx);
  double r1 = sqi * (pcj.y * id - pjy);
  const double sn = r0 * r0 + r1 * r1;
  // ceres::CauchyLoss + Corrector: rho'' < 0 => residual and Jacobian scale by sqrt(rho')
  const double b = cauchy_a * cauchy_a, c = fast_rcp(b);
  const double sum = 1.0 + sn * c;
  const double rho0 = b * log(sum);
  // sqrt(max(DBL_MIN, 1 / sum)), Corrector's sqrt(rho'): 1.4916681462400413e-154 = sqrt(DBL_MIN)
  const double srho = apply_loss ? fmax(fast_rsqrt_pe(sum), 1.4916681462400413e-154) : 1.0;
  r[0] = srho * r0;
  r[1] = srho * r1;
  if (WANT_J) {
    const double* Ab = fr.A + fb * 9;
    const double id2 = id * id;
    // reduce = sqrt_info [1/z 0 -x/z^2; 0 1/z -y/z^2], with the loss scaling folded in: two terms per entry
    const double rd = srho * sqi * id, rx = -(srho * sqi) * (pcj.x * id2), ry = -(srho * sqi) * (pcj.y * id2);
    const double red[6] = {sqi * id, 0.0, sqi * (-pcj.x * id2), 0.0, sqi * id, sqi * (-pcj.y * id2)};
    double M[6], MR[6], N[6];
#pragma unroll
    for (int cc = 0; cc < 3; cc++) {
      M[cc] = rd * Ab[cc] + rx * Ab[6 + cc];
      M[3 + cc] = rd * Ab[3 + cc] + ry * Ab[6 + cc];
      N[cc] = rd * ric[cc * 3] + rx * ric[cc * 3 + 2];
      N[3 + cc] = rd * ric[cc * 3 + 1] + ry * ric[cc * 3 + 2];
    }
#pragma unroll
    for (int rr = 0; rr < 2; rr++)
#pragma unroll
      for (int cc = 0; cc < 3; cc++) MR[rr * 3 + cc] = M[rr * 3] * Ra[cc] + M[rr * 3 + 1] * Ra[3 + cc] + M[rr * 3 + 2] * Ra[6 + cc];
    const v3 u = Rmul(ric, mk3(pix, piy, 1.0));  // ric * pts_i
    const double il2 = -(il * il);
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
      // ProjectionTdFactor (projection_td_factor.cpp:131-136): d r / d td = reduce ric^T Rj^T Ri ric velocity_i (-1 / lambda) +
      // sqrt_info velocity_j; (pix, piy) / (pjx, pjy) are the td-shifted observations then (the caller shifts them)
      if (Jtd) Jtd[rr] = dot(mr, Rmul(ric, mk3(vix, viy, 0.0))) * -il + (srho * sqi) * (rr == 0 ? vjx : vjy);
      if (Jex) {
        // jaco_ex (projection_factor.cpp:97-107): left = ric^T (Rj^T Ri - I); right = -tmp_r [pc_i]x + [tmp_r pc_i]x + [q]x,
        // and tmp_r pc_i + q is the point in camera j, so the two skew terms collapse to [pc_j]x
        const v3 mrr = mk3(mr.x * ric[0] + mr.y * ric[3] + mr.z * ric[6], mr.x * ric[1] + mr.y * ric[4] + mr.z * ric[7],
                           mr.x * ric[2] + mr.y * ric[5] + mr.z * ric[8]);  // (reduce ric^T Rj^T Ri ric) row
        const v3 rho = mk3(srho * red[rr * 3], srho * red[rr * 3 + 1], srho * red[rr * 3 + 2]);
        const v3 ex_r = cross(pci, mrr) + cross(rho, pcj);
        Jex[rr * 6 + 0] = mr.x - n.x, Jex[rr * 6 + 1] = mr.y - n.y, Jex[rr * 6 + 2] = mr.z - n.z;
        Jex[rr * 6 + 3] = ex_r.x, Jex[rr * 6 + 4] = ex_r.y, Jex[rr * 6 + 5] = ex_r.z;
      }
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
  } else if (kind == AVM_BLK_TD) {
    dx[0] = xb[0] - x0[0];
  } else {
    for (int k = 0; k < 3; k++) dx[k] = xb[k] - x0[k];
    const quat q0{x0[6], x0[3], x0[4], x0[5]}, q{xb[6], xb[3], xb[4], xb[5]};
    const quat d = qmul(qinv(q0), q);
    const double sg = (d.w >= 0) ? 2.0 : -2.0;
    dx[3] = sg * d.x, dx[4] = sg * d.y, dx[5] = sg * d.z;
  }
}

// Global-memory pointers carried into the outlined phases are typed with their address space: behind a struct
// reference the compiler cannot prove it and would fall back to flat_load/flat_store, which count against the LDS
// counter too (every LDS wait then also waits for HBM).
typedef double dv2 __attribute__((ext_vector_type(2)));
#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) const double gcdouble;
typedef __attribute__((address_space(1))) int32_t gint;
typedef __attribute__((address_space(1))) long long glong;
typedef __attribute__((address_space(1))) const dv2 gcdv2;
template <class T> AVM_DEV __attribute__((address_space(1))) T* as_global(T* p) { return (__attribute__((address_space(1))) T*)p; }
#else  // host pass of the same translation unit: plain pointers
typedef double gdouble;
typedef const double gcdouble;
typedef int32_t gint;
typedef long long glong;
typedef const dv2 gcdv2;
template <class T> AVM_DEV T* as_global(T* p) { return p; }
#endif

struct WinCtx {
  glong* prof;
  gdouble* sc;   // global scratch slot
  gint* osf;     // observation slot -> feature
  gint* cov;     // [11][150] features observed in frame b, in feature order
  int w, nf, nobs_tot, pn, pnblk;
  gcdouble* obs;   // [max_obs][2]
  gcdouble *pdelta, *pjac, *psqrt, *psum;  // this window's 10 intervals
  gcdouble *lba, *lbg;
  gcdouble *pJ, *pr, *px0;  // prior
  int ldp;
  // optional members of the problem (the solve reads them in the AVM_X build only, the marginalization in both)
  gcdouble* aux;      // [max_obs][4] velocity.x, velocity.y, cur_td, uv.y per observation slot (null unless estimate_td)
  gcdouble* relo_xy;  // [relo_n][2] match points
  int relo_n;         // > 0: the relocalization frame takes part (frame 11)
  int has_relo;       // relocalization_info: relo_Pose is frame 11 of the state and goes through the gauge fix, even with no match (relo_n == 0)
  int est_ex, est_td;
};

static_assert(sizeof(WinCtx) <= 32 * 8, "WinCtx outgrew its LDS slot");
// The per-window context and the options live in LDS: handed to the outlined phases by reference they would sit in
// the caller's private (scratch) memory and every field access would be a flat load from it.
AVM_DEV const WinCtx& lds_ctx() { return *reinterpret_cast<const WinCtx*>(LDS() + L_CTX); }
AVM_DEV const avm_options& lds_opt() { return *reinterpret_cast<const avm_options*>(LDS() + L_OPT); }
AVM_DEV void lds_store_ctx(const WinCtx& cl, const avm_options& ol) {  // call by all threads, then barrier
  if (threadIdx.x == 0) *reinterpret_cast<WinCtx*>(LDS() + L_CTX) = cl;
  const int nw = (int)(sizeof(avm_options) / 4);
  const int* src = reinterpret_cast<const int*>(&ol);
  int* dst = reinterpret_cast<int*>(LDS() + L_OPT);
  for (int i = threadIdx.x; i < nw; i += NT) dst[i] = src[i];
}

// frames: R_f and A_f = ric^T R_f^T for state vector xs into frame slot `which`
AVM_DEV void build_frames(int xs_off, int which) {
  double* lds = LDS();
  const double* xs = lds + xs_off;
  const int t = threadIdx.x;
  double* R = lds + L_FR + which * FRS;
  double* A = R + 9 * NFRP;
#ifdef AVM_X
  // ex_pose is part of the state here: every thread that needs ric recomputes it (thread NFRP publishes it for the factors)
  double ricv[9];
  q2R(quat{xs[XEX + 6], xs[XEX + 3], xs[XEX + 4], xs[XEX + 5]}, ricv);
  const double* ric = ricv;
  if (t == NFRP) {
    double* dst = lds + L_RIC + which * 12;
    for (int k = 0; k < 9; k++) dst[k] = ricv[k];
    for (int k = 0; k < 3; k++) dst[9 + k] = xs[XEX + k];
  }
#else
  const double* ric = lds + L_RIC;
#endif
  if (t < NFRP) {
    quat q{xs[t * 7 + 6], xs[t * 7 + 3], xs[t * 7 + 4], xs[t * 7 + 5]};
    double Rm[9];
    q2R(q, Rm);
    for (int k = 0; k < 9; k++) R[t * 9 + k] = Rm[k];
    for (int a = 0; a < 3; a++)
      for (int b = 0; b < 3; b++) A[t * 9 + a * 3 + b] = ric[0 * 3 + a] * Rm[b * 3 + 0] + ric[1 * 3 + a] * Rm[b * 3 + 1] + ric[2 * 3 + a] * Rm[b * 3 + 2];
  }
}

// ric / tic the factors of frames slot `which` are evaluated with
AVM_DEV const double* ric_of(int which) {
#ifdef AVM_X
  return LDS() + L_RIC + which * 12;
#else
  (void)which;
  return LDS() + L_RIC;
#endif
}

// the td-shifted pair of observations of one ProjectionTdFactor (projection_td_factor.cpp:50-52) and the two velocities:
// ob = {pts_i.x, pts_i.y, pts_j.x, pts_j.y}, ai / aj = {velocity.x, velocity.y, cur_td, uv.y} of the two observations
AVM_DEV void td_shift(double* ob, const double* ai, const double* aj, double td, double tr, double row) {
  const double si = td - ai[2] + tr / row * (ai[3] - row / 2), sj = td - aj[2] + tr / row * (aj[3] - row / 2);
  ob[0] -= si * ai[0], ob[1] -= si * ai[1], ob[2] -= sj * aj[0], ob[3] -= sj * aj[1];
}

#ifndef AVM_TP
// prior residual r_p = r0 + J0 * dx(xs) into lds[L_RP]; returns (to all threads) nothing; needs syncs by caller
AVM_NOINL void prior_residual_dev(const WinCtx&, int xs_off) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const double* xs = lds + xs_off;
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int t = threadIdx.x;
  // r_p[i] = r0[i] + sum_k J0[i][k] dx[k] : 4 lanes per row (k = part, part + 4, ...).  The row's entries do not depend on
  // dx: their loads are issued first, so the trip to the slot's memory overlaps the dx computation and the barrier
  const int row = t >> 2, part = t & 3;
  static_assert(NT >= 4 * MAXPRIOR, "one pass over the rows");
  double v[MAXPRIOR / 4], r0 = 0.0;
  {
    gcdouble* Jr = c.pJ + (size_t)min(row, max(c.pn - 1, 0)) * c.ldp;
#pragma unroll
    for (int j = 0; j < MAXPRIOR / 4; j++) v[j] = Jr[min(part + 4 * j, max(c.pn - 1, 0))];  // clamped, masked below
    r0 = c.pr[min(row, max(c.pn - 1, 0))];
  }
  if (t < c.pnblk) {
    const int kind = ids[I_PBLK + t * 3], fr = ids[I_PBLK + t * 3 + 1], off = ids[I_PBLK + t * 3 + 2];
#ifdef AVM_X
    const double* xb = kind == AVM_BLK_POSE ? xs + fr * 7 : (kind == AVM_BLK_SPEEDBIAS ? xs + XSB + fr * 9 : (kind == AVM_BLK_TD ? xs + XTD : xs + XEX));
#else
    // ex_pose is constant in the solve; its current value sits behind ric/tic
    const double* xb = kind == AVM_BLK_POSE ? xs + fr * 7 : (kind == AVM_BLK_SPEEDBIAS ? xs + XSB + fr * 9 : lds + L_RIC + (kind == AVM_BLK_TD ? 19 : 12));
#endif
    double dx[9];
    prior_block_dx(kind, xb, c.px0 + t * 9, dx);
    const int n = kind == AVM_BLK_SPEEDBIAS ? 9 : (kind == AVM_BLK_TD ? 1 : 6);
    for (int k = 0; k < n; k++) lds[L_DXP + off + k] = dx[k];
  }
  __syncthreads();
  {
    double s = 0;
#pragma unroll
    for (int j = 0; j < MAXPRIOR / 4; j++) s += (part + 4 * j < c.pn ? v[j] : 0.0) * lds[L_DXP + part + 4 * j];
    s += __shfl_xor(s, 1, 64);
    s += __shfl_xor(s, 2, 64);
    if (row < c.pn && part == 0) lds[L_RP + row] = r0 + s;
  }
  __syncthreads();
}

#endif  // !AVM_TP

// The prior's share of an evaluation on ONE wavefront, with wave-level synchronisation only, so that it runs beside the
// projection factors (whose wavefronts do not touch these LDS ranges) instead of in a phase of its own:
//   dx -> lds[L_DXP],  r_p = r0 + J0 dx -> lds[L_RP],  and (WANT_G) g_p = J0^T r_p -> lds[L_DXP], over dx;
// returns 1/2 |r_p|^2 on every lane.  J0 is read along its rows both times (16 lanes per row for r_p, a lane per column
// for g_p), several rows in flight; every sum has a fixed order.
// Rows [rb, re) of the prior only, dx / g_p in the buffer at buf_off: two wavefronts can share the prior, each with its own buffer;
// their costs and their g_p add up (r_p rows are disjoint).
template <bool WANT_G>
AVM_DEV double prior_wave(int xs_off, int rb, int re, int buf_off) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const double* xs = lds + xs_off;
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int lane = threadIdx.x & 63, lr = lane & 15, lg = lane >> 4;
  const int pn = c.pn, pn1 = max(pn - 1, 0);
  if (lane < c.pnblk) {
    const int kind = ids[I_PBLK + lane * 3], fr = ids[I_PBLK + lane * 3 + 1], off = ids[I_PBLK + lane * 3 + 2];
#ifdef AVM_X
    const double* xb = kind == AVM_BLK_POSE ? xs + fr * 7 : (kind == AVM_BLK_SPEEDBIAS ? xs + XSB + fr * 9 : (kind == AVM_BLK_TD ? xs + XTD : xs + XEX));
#else
    // ex_pose is constant in the solve; its current value sits behind ric/tic
    const double* xb = kind == AVM_BLK_POSE ? xs + fr * 7 : (kind == AVM_BLK_SPEEDBIAS ? xs + XSB + fr * 9 : lds + L_RIC + (kind == AVM_BLK_TD ? 19 : 12));
#endif
    double dx[9];
    prior_block_dx(kind, xb, c.px0 + lane * 9, dx);
    const int n = kind == AVM_BLK_SPEEDBIAS ? 9 : (kind == AVM_BLK_TD ? 1 : 6);
    for (int k = 0; k < n; k++) lds[buf_off + off + k] = dx[k];
  }
  wave_lds_sync();
  constexpr int NK = MAXPRIOR / 16, RU = 7;  // 6 column groups of 16; 7 x 4 rows in flight (three trips to the slot's memory for 75 rows)
  double dxv[NK];
#pragma unroll
  for (int j = 0; j < NK; j++) dxv[j] = lr + 16 * j < pn ? lds[buf_off + lr + 16 * j] : 0.0;
  double cost = 0;
  for (int r0 = rb; r0 < re; r0 += 4 * RU) {
    double v[RU][NK], rr[RU];
#pragma unroll
    for (int u = 0; u < RU; u++) {
      const int rc = min(r0 + 4 * u + lg, pn1);
      gcdouble* Jr = c.pJ + (size_t)rc * c.ldp;
#pragma unroll
      for (int j = 0; j < NK; j++) v[u][j] = Jr[min(lr + 16 * j, pn1)];  // clamped; the padding columns meet dx = 0
      rr[u] = c.pr[rc];
    }
#pragma unroll
    for (int u = 0; u < RU; u++) {
      double sacc = 0;
#pragma unroll
      for (int j = 0; j < NK; j++) sacc += v[u][j] * dxv[j];
      sacc += __shfl_xor(sacc, 8, 64);
      sacc += __shfl_xor(sacc, 4, 64);
      sacc += __shfl_xor(sacc, 2, 64);
      sacc += __shfl_xor(sacc, 1, 64);
      const int row = r0 + 4 * u + lg;
      const double rp = rr[u] + sacc;
      if (lr == 0 && row < re) {
        lds[L_RP + row] = rp;
        cost += 0.5 * rp * rp;
      }
    }
  }
  cost = wave_sum(cost);
  if (WANT_G) {
    wave_lds_sync();
    // g_p[k] = sum_i J0[i][k] r_p[i]: lane = column (k = lane, lane + 64), rows in ascending order, 19 rows in flight
    constexpr int GU = 19;  // (four trips for 75 rows)
    const int k0 = min(lane, pn1), k1 = min(lane + 64, pn1);
    double g0 = 0, g1 = 0;
    for (int i0 = rb; i0 < re; i0 += GU) {
      double a0[GU], a1[GU];
#pragma unroll
      for (int u = 0; u < GU; u++) {
        gcdouble* Jr = c.pJ + (size_t)min(i0 + u, pn1) * c.ldp;
        a0[u] = Jr[k0], a1[u] = Jr[k1];
      }
#pragma unroll
      for (int u = 0; u < GU; u++) {
        const double r = i0 + u < re ? lds[L_RP + min(i0 + u, MAXPRIOR - 1)] : 0.0;
        g0 += a0[u] * r, g1 += a1[u] * r;
      }
    }
    lds[buf_off + lane] = lane < pn ? g0 : 0.0;  // (dx lives in dxv by now)
    if (lane + 64 < MAXPRIOR) lds[buf_off + lane + 64] = lane + 64 < pn ? g1 : 0.0;
  }
  return cost;
}

// residual-only cost at state xs (frames slot `which` must be built). Uses lds[L_S..] as IMU staging.
AVM_NOINL double eval_cost(const WinCtx&, const avm_options&, int xs_off, int which) {
  const WinCtx& c = lds_ctx();
  const avm_options& o = lds_opt();
  double* lds = LDS();
  const double* xs = lds + xs_off;
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int t = threadIdx.x;
  Frames fr{lds + L_FR + which * FRS, lds + L_FR + which * FRS + 9 * NFRP};
  const double* ric = ric_of(which);
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
  if (c.nobs_tot > 0) {
    // thread per observation slot; the feature id and the two observations of every slot a thread owns are
    // fetched up front (two dependent rounds of loads in total instead of two per slot); slots past the end are
    // clamped to the last valid one, so every address stays inside the window's tables
    constexpr int NSL = (MAXOBS + NT - 1) / NT;
    int es[NSL], s0s[NSL];
    double ob[NSL][4];
#pragma unroll
    for (int u = 0; u < NSL; u++) es[u] = min(max(c.osf[min(t + u * NT, c.nobs_tot - 1)], 0), c.nf - 1);
#pragma unroll
    for (int u = 0; u < NSL; u++) {
      const int s = min(t + u * NT, c.nobs_tot - 1);
      s0s[u] = ids[I_FOBS + es[u]];
      ob[u][0] = c.obs[2 * s0s[u]], ob[u][1] = c.obs[2 * s0s[u] + 1], ob[u][2] = c.obs[2 * s], ob[u][3] = c.obs[2 * s + 1];
    }
#pragma unroll
    for (int u = 0; u < NSL; u++) {
      const int s = t + u * NT;
      const int e = es[u];
      // the observation table may have holes (avm_slide_window drops a feature's first observation in place): a slot
      // belongs to the feature the slot map names only if it lies inside that feature's range of the CURRENT table
      if (s >= c.nobs_tot || s <= s0s[u] || s >= s0s[u] + ids[I_FNOBS + e]) continue;
      const int fa = ids[I_FSTART + e], fb = fa + (s - s0s[u]);
      double r[2];
#ifdef AVM_X
      if (c.est_td) {
        double ai[4], aj[4];
#pragma unroll
        for (int k = 0; k < 4; k++) ai[k] = c.aux[4 * s0s[u] + k], aj[k] = c.aux[4 * s + k];
        td_shift(ob[u], ai, aj, xs[XTD], o.tr, o.row);
      }
#endif
      acc += proj_eval<false>(xs, fr, ric, ric + 9, ob[u][0], ob[u][1], ob[u][2], ob[u][3], xs[XLAM + e], fa, fb, sqi,
                              o.cauchy_a, true, r, nullptr, nullptr, nullptr);
    }
  }
#ifdef AVM_X
  // relocalization factors (estimator.cpp:760-792): plain ProjectionFactors against relo_Pose = frame 11
  for (int k = t; k < c.relo_n; k += NT) {
    const int e = c.cov[(NFRP - 1) * MAXE + k], s0 = ids[I_FOBS + e];
    double r[2];
    acc += proj_eval<false>(xs, fr, ric, ric + 9, c.obs[2 * s0], c.obs[2 * s0 + 1], c.relo_xy[2 * k], c.relo_xy[2 * k + 1], xs[XLAM + e],
                            ids[I_FSTART + e], NFRP - 1, sqi, o.cauchy_a, true, r, nullptr, nullptr, nullptr);
  }
#endif
  // the prior on the last wavefront (the same code, hence the same rounding, as in eval_jac)
  if (t >= NT - 64 && c.pn > 0) {
    const double pc = prior_wave<false>(xs_off, 0, c.pn, L_DXP);
    if (t == NT - 64) acc += pc;
  }
  __syncthreads();
  if (t < 150) {
    const int i = t / 15, r = t % 15;
    if (c.psum[i] <= o.max_sum_dt) {
      double ps[15], s = 0;  // (all fifteen loads in flight: from k = r every step was a trip to memory of its own)
#pragma unroll
      for (int k = 0; k < 15; k++) ps[k] = c.psqrt[i * 225 + r * 15 + k];
#pragma unroll
      for (int k = 0; k < 15; k++) s += k >= r ? ps[k] * lds[L_S + i * 465 + k * 31] : 0.0;
      acc += 0.5 * s * s;
    }
  }

  return block_sum1(acc);
}

// Prior J0^T J0 on the matrix cores (16x16 tiles, K = prior rows), marginalization-kernel variant: the tiles are
// added straight into the packed system in LDS at the
// columns pidx[] maps the prior's columns to (every lower entry is produced exactly once, so the wavefronts never
// touch the same element).  All operand loads of a tile are issued before the MFMA chain.
AVM_NOINL void prior_jtj_add_lds(gcdouble* pJ, int ldp, int pn, int s_off) {
  double* lds = LDS();
  const int* pidx = reinterpret_cast<const int*>(lds + L_INT) + I_PIDX;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ntl = (pn + 15) >> 4;
  for (int tile = wv; tile < ntl * (ntl + 1) / 2; tile += NT / 64) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ti++;
    const int tj = tile - ti * (ti + 1) / 2;
    const int ca = min(16 * ti + (lane & 15), pn - 1), cb = min(16 * tj + (lane & 15), pn - 1);
    const bool va = 16 * ti + (lane & 15) < pn, vb = 16 * tj + (lane & 15) < pn;
    double av[MAXPRIOR / 4], bv[MAXPRIOR / 4];
#pragma unroll
    for (int m = 0; m < MAXPRIOR / 4; m++) {
      const int r = 4 * m + (lane >> 4), rc = min(r, pn - 1);
      const double a = pJ[(size_t)rc * ldp + ca], b = pJ[(size_t)rc * ldp + cb];
      av[m] = (r < pn && va) ? a : 0.0;
      bv[m] = (r < pn && vb) ? b : 0.0;
    }
    d4 D = {0, 0, 0, 0};
#pragma unroll
    for (int m = 0; m < MAXPRIOR / 4; m++) D = __builtin_amdgcn_mfma_f64_16x16x4f64(av[m], bv[m], D, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gi = 16 * ti + (lane >> 4) + 4 * r, gj = 16 * tj + (lane & 15);
      if (gi < pn && gj <= gi) {
        const int ip = pidx[gi], iq = pidx[gj];
        if (ip >= 0 && iq >= 0) lds[s_off + roff(max(ip, iq)) + min(ip, iq)] += D[r];
      }
    }
  }
}

// Solve-kernel variant: lower triangle packed by idx = p (p + 1) / 2 + q into HPk, plus the destination of every
// entry inside the packed S (or -1 if the prior column is not a state of the solve) - the per-iteration add is then
// a flat gather.  All operand loads of a tile are issued before the MFMA chain.
// Layout of the solve kernel's per-factor products in the scratch slot: the FEATURE index runs fastest, so the
// frame tasks (lane = feature) write, and the per-feature sums / Schur tiles / back substitution read, whole lines:
//   Wt [NPOSE][WLE]       E^T F transposed: Wt[c][e] = (E^T F)[e][c]
//   PFt[8][NFR][WLE]      per (quantity q, observing frame b, feature e): Ji^T Je (q < 6), Je^T Je, Je^T r
constexpr int WLE = 152;
#ifdef AVM_X
constexpr int NQ = 15;      // per-factor quantities: Ji^T Je (6), Je^T Je, Je^T r, Jex^T Je (6), Jtd^T Je
constexpr int SPARTW = 69;   // per (frame b, start a): Ji^T Ji (21) | Ji^T r (6) | [Jex; Jtd]^T Ji (7 x 6)
constexpr int PARTX = 35;   // per frame b: [Jex; Jtd]^T [Jex; Jtd] lower (28) | [Jex; Jtd]^T r (7)
constexpr int PARTX0 = NFRP * NFR * SPARTW;
static_assert(PARTX0 + NFRP * PARTX <= 9600, "partial blocks fit the PART region");
#else
constexpr int NQ = 8;
constexpr int SPARTW = 27;
#endif
static_assert(NPOSE * WLE <= 80 * 152 && NQ * NFRP * WLE <= 17 * MAXOBS, "transposed layouts fit the W / PF regions");
constexpr int HPK_MAX = MAXPRIOR * (MAXPRIOR + 1) / 2;  // 4656 doubles, followed by 4656 ints (fits the [96][96] slot)
static_assert(HPK_MAX + HPK_MAX / 2 <= MAXPRIOR * MAXPRIOR, "packed Hp + destinations fit the HP scratch region");
AVM_NOINL void prior_jtj_packed(gcdouble* pJ, int ldp, int pn, gdouble* HPk, gint* dst) {
  const int* pidx = reinterpret_cast<const int*>(LDS() + L_INT) + I_PIDX;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const int ntl = (pn + 15) >> 4;
  for (int tile = wv; tile < ntl * (ntl + 1) / 2; tile += NT / 64) {
    int ti = 0;
    while ((ti + 1) * (ti + 2) / 2 <= tile) ti++;
    const int tj = tile - ti * (ti + 1) / 2;
    const int ca = 16 * ti + (lane & 15), cb = 16 * tj + (lane & 15);
    // all 48 loads in flight (clamped to a valid element, masked afterwards: a predicated load is a branch with its own
    // s_waitcnt), four independent MFMA chains
    double av[MAXPRIOR / 4], bv[MAXPRIOR / 4];
    const int cac = min(ca, pn - 1), cbc = min(cb, pn - 1);
#pragma unroll
    for (int m = 0; m < MAXPRIOR / 4; m++) {
      const int r = min(4 * m + (lane >> 4), pn - 1);
      av[m] = pJ[(size_t)r * ldp + cac];
      bv[m] = pJ[(size_t)r * ldp + cbc];
    }
    d4 Dq[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
    for (int m = 0; m < MAXPRIOR / 4; m++) {
      const bool rv = 4 * m + (lane >> 4) < pn;
      const double a = (rv && ca < pn) ? av[m] : 0.0, b = (rv && cb < pn) ? bv[m] : 0.0;
      Dq[m & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, Dq[m & 3], 0, 0, 0);
    }
    const d4 D = (Dq[0] + Dq[1]) + (Dq[2] + Dq[3]);
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gi = 16 * ti + (lane >> 4) + 4 * r, gj = 16 * tj + (lane & 15);
      if (gi < pn && gj <= gi) {
        const int idx = gi * (gi + 1) / 2 + gj;
        const int ip = pidx[gi], iq = pidx[gj];
        HPk[idx] = D[r];
#ifdef AVM_TP
        dst[idx] = (ip < 0 || iq < 0) ? -1 : s_off(max(ip, iq), min(ip, iq)) - L_S;  // (-1 also where the structural form has no slot: see tp_prior_ok)
#else
        dst[idx] = (ip < 0 || iq < 0) ? -1 : roff(max(ip, iq)) + min(ip, iq);
#endif
      }
    }
  }
}

// All projection factors observed in frame b, by one wavefront (lane = factor, 64 at a time).
// Each lane evaluates its factor, then the 2 x 13 rows [Jj | Ji | r] of the 64 factors are staged in LDS
// and X^T X is accumulated with v_mfma_f64_16x16x4: one 16x16 product gives Jj^T Jj (block b,b),
// Jj^T Ji (block b,a), Ji^T Ji (goes to block a,a), Jj^T r and Ji^T r at once — the cross-lane reduction
// is done by the matrix core.  Features are sorted by start frame, so factors with the same start frame a
// are consecutive; the B operand is masked per a-run to keep the (b,a)/(a,a) blocks separate.
// Blocks (b,b), (b,a) and g_b belong to this frame only and are written straight into LDS; the (a,a)
// contributions go to PART[b][a] in the scratch slot and are summed in a fixed order afterwards.
#ifndef AVM_X
// One wavefront takes ALL the frames assigned to it as one list of factors (frames in ascending order, each frame's factors
// in feature order), 64 at a time: a chunk may straddle two frames, so a wavefront with two frames of 150 factors runs 5
// chunks instead of 3 + 3.  The runs of the MFMA accumulation are keyed by (frame b, start frame a).
AVM_DEV double frame_task(const WinCtx&, const avm_options&, int wvi, int stage_off) {
  const WinCtx& c = lds_ctx();
  const avm_options& o = lds_opt();
  double* lds = LDS();
  double* stage = lds + stage_off;
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int lane = threadIdx.x & 63;
  Frames fr{lds + L_FR, lds + L_FR + 99};
  const double* xs = lds + L_X;
  const double sqi = o.focal_length / 1.5;
  double* W = c.sc + Scratch::W;
  double* PF = c.sc + Scratch::PF;
  double* PART0 = c.sc + Scratch::PART;
  const double* scl = lds + L_SC;
  d4 Dtot = {0, 0, 0, 0}, Drun = {0, 0, 0, 0}, Drun1 = {0, 0, 0, 0}, Drun2 = {0, 0, 0, 0}, Drun3 = {0, 0, 0, 0};
  int a_run = -1, b_run = -1, pmask = 0;
  double cost = 0;
  const int drow = lane >> 4, dcol = lane & 15;
  // end offsets of the frames in this wavefront's list (a frame of another wavefront has zero width); wave-uniform values
  // kept in scalar registers, so that locating a factor costs a few compares and no LDS traffic
  int endo[NFR];
  endo[0] = 0;
  {
    int off = 0;
#pragma unroll
    for (int bb = 1; bb < NFR; bb++) {
      off += ids[I_FRW + bb] == wvi ? ids[I_NCOV + bb] : 0;
      endo[bb] = __builtin_amdgcn_readfirstlane(off);
    }
  }
  const int ntot = endo[NFR - 1];  // factors of this wavefront's frames
  // position in the wavefront's list -> (frame, index in the frame's list); past the end: the last factor (masked by `act`)
  auto locate = [&](int idx, int& bl, int& pos) {
    const int ic = min(idx, ntot - 1);
    int start = 0;
    bl = 1;
#pragma unroll
    for (int bb = 1; bb < NFR - 1; bb++) {
      const bool past = ic >= endo[bb];
      bl += past ? 1 : 0;
      start = past ? endo[bb] : start;
    }
    pos = ic - start;
  };
  auto flush = [&]() {  // ends the run (b_run, a_run)
    if (a_run < 0) return;
    Drun = (Drun + Drun1) + (Drun2 + Drun3);
    Drun1 = Drun2 = Drun3 = d4{0, 0, 0, 0};
    double* PART = PART0 + (size_t)b_run * NFR * 27;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = drow + 4 * r;
      const double v = Drun[r];
      // (entries of S are written Jacobi-scaled: s_i s_j H_ij, with s = 1 until the first evaluation has fixed it)
      if (row < 6 && dcol >= 6 && dcol < 12)
        lds[L_S + roff(6 * b_run + row) + 6 * a_run + (dcol - 6)] = v * (scl[6 * b_run + row] * scl[6 * a_run + (dcol - 6)]);  // Jj^T Ji
      if (row >= 6 && row < 12) {
        const int i = row - 6;
        if (dcol >= 6 && dcol < 12 && dcol - 6 <= i) PART[a_run * 27 + i * (i + 1) / 2 + (dcol - 6)] = v;  // Ji^T Ji (lower)
        if (dcol == 12) PART[a_run * 27 + 21 + i] = v;                                                    // Ji^T r
      }
    }
    pmask |= 1 << a_run;
    Dtot += Drun;
    Drun = d4{0, 0, 0, 0};
    a_run = -1;
  };
  auto end_frame = [&]() {  // block (b,b) lower triangle and g_b of the frame that just ended
    flush();
    if (b_run < 0) return;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = drow + 4 * r;
      if (row < 6 && dcol <= row) lds[L_S + roff(6 * b_run + row) + 6 * b_run + dcol] = Dtot[r] * (scl[6 * b_run + row] * scl[6 * b_run + dcol]);
      if (row < 6 && dcol == 12) lds[L_G + 6 * b_run + row] = Dtot[r];
    }
    if (lane == 0) ids[I_PMASK + b_run] = pmask;
    Dtot = d4{0, 0, 0, 0};
    pmask = 0;
  };
  // inputs of a chunk (feature id, its two observations) are fetched one chunk ahead: their HBM / L2 latency hides
  // behind the stores, the staging and the MFMA chain of the chunk before
  int e_nx = 0, fa_nx = 0, b_nx = 1;
  double ob_nx[4] = {0, 0, 0, 0};
  auto fetch = [&](int chunk0) {
    int pos;
    locate(chunk0 + lane, b_nx, pos);
    e_nx = c.cov[b_nx * MAXE + pos];
    fa_nx = ids[I_FSTART + e_nx];
    const int s0 = ids[I_FOBS + e_nx], s = s0 + (b_nx - fa_nx);
    ob_nx[0] = c.obs[2 * s0], ob_nx[1] = c.obs[2 * s0 + 1], ob_nx[2] = c.obs[2 * s], ob_nx[3] = c.obs[2 * s + 1];
  };
  if (ntot > 0) fetch(0);
  for (int chunk0 = 0; chunk0 < ntot; chunk0 += 64) {
    const int idx = chunk0 + lane;
    const bool act = idx < ntot;
    const int e = e_nx, fa = fa_nx, b = b_nx;  // (inactive lanes repeat the wavefront's last factor: valid, never stored)
    const double ob0 = ob_nx[0], ob1 = ob_nx[1], ob2 = ob_nx[2], ob3 = ob_nx[3];
    if (chunk0 + 64 < ntot) fetch(chunk0 + 64);
    double r[2] = {0, 0}, Ji[12], Jj[12], Je[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 12; k++) Ji[k] = 0, Jj[k] = 0;
    if (act) {
      cost += proj_eval<true>(xs, fr, lds + L_RIC, lds + L_RIC + 9, ob0, ob1, ob2, ob3,
                              xs[XLAM + e], fa, b, sqi, o.cauchy_a, true, r, Ji, Jj, Je);
      // (round 5: Ji's translation columns are minus Jj's - set so in proj_eval -, so Ji_t^T Je is exactly -W[6 b + k][e], k < 3: those three products
      //  are not stored a second time, the per-feature sums read them out of W.  Every 8 bytes per factor written here cost 0.1 ms per 4096 windows:
      //  profiles/r05e_experiments.md section 10.)
#pragma unroll
      for (int k = 0; k < 6; k++) {
        W[(6 * b + k) * WLE + e] = Jj[k] * Je[0] + Jj[6 + k] * Je[1];
        if (k >= 3) PF[(k * NFR + b) * WLE + e] = Ji[k] * Je[0] + Ji[6 + k] * Je[1];
      }
#ifndef AVM_TP
      PF[(6 * NFR + b) * WLE + e] = Je[0] * Je[0] + Je[1] * Je[1];
      PF[(7 * NFR + b) * WLE + e] = Je[0] * r[0] + Je[1] * r[1];
#endif
    }
#ifdef AVM_TP
    {
      // E^T E and E^T r of the chunk's factors into this wavefront's accumulators (L_ACC above), frame by frame: the frames of a list are in ascending
      // order along the lanes, and a feature occurs once per frame - so the lanes of one frame never meet in an address, and a feature's terms are
      // added in the order of this wavefront's frames, always the same
      const double he = Je[0] * Je[0] + Je[1] * Je[1], ge = Je[0] * r[0] + Je[1] * r[1];
      double* a0 = wvi == 0 ? lds + L_HEE : lds + L_ACC + (2 * (wvi - 1)) * ACCW;
      double* a1 = wvi == 0 ? lds + L_G + NF : lds + L_ACC + (2 * (wvi - 1) + 1) * ACCW;
      int bb = __builtin_amdgcn_readfirstlane(b);
      for (;;) {
        if (act && b == bb) a0[e] += he, a1[e] += ge;
        wave_lds_sync();
        const unsigned long long rest = __ballot(act && b > bb);
        if (!rest) break;
        bb = __builtin_amdgcn_readlane(b, (int)__ffsll((long long)rest) - 1);
      }
    }
    // Throughput build: the staging tile holds HALF a chunk (lanes 0-31 stage and the wavefront multiplies, then lanes 32-63; the
    // scheme of marg_frame_task).  A run that straddles the two halves simply continues: the switches below only act on a new key.
    const int nact = min(64, ntot - chunk0);
    const int key = (b << 4) | fa;  // frames ascending, start frames ascending inside a frame: equal keys are consecutive
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      const int h0 = 32 * half, lim = min(nact, h0 + 32);
      if (h0 >= nact) break;  // (uniform)
      if ((lane >> 5) == half) {
        dv2* st = reinterpret_cast<dv2*>(stage) + (lane & 31);
#pragma unroll
        for (int k = 0; k < 6; k++) st[k * (XRS_H / 2)] = dv2{Jj[k], Jj[6 + k]}, st[(6 + k) * (XRS_H / 2)] = dv2{Ji[k], Ji[6 + k]};
        st[12 * (XRS_H / 2)] = dv2{r[0], r[1]};
      }
      wave_lds_sync();
      int l = h0;
      while (l < lim) {
        const int k_cur = __shfl(key, l, 64);
        const int l_end = min(l + __popcll(__ballot(act && key == k_cur && lane >= l)), lim);
        if ((k_cur >> 4) != b_run) {
          end_frame();
          b_run = k_cur >> 4;
        }
        if ((k_cur & 15) != a_run) {
          flush();
          a_run = k_cur & 15;
        }
        const int j_end = (l_end - h0 + 3) >> 2;
#pragma unroll 1
        for (int j0 = (l - h0) >> 2; j0 < j_end; j0 += 4) {
          dv2 v[4];
#pragma unroll
          for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const dv2*>(stage + min(dcol, 12) * XRS_H + 8 * min(j0 + u, 7) + 2 * drow);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int f = h0 + 4 * (j0 + u) + drow;
            const bool in = dcol < 13 && f >= l && f < l_end;
            const double a0 = in ? v[u][0] : 0.0, a1 = in ? v[u][1] : 0.0;
            if (u & 1) {
              Drun2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, Drun2, 0, 0, 0);
              Drun3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, Drun3, 0, 0, 0);
            } else {
              Drun = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, Drun, 0, 0, 0);
              Drun1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, Drun1, 0, 0, 0);
            }
          }
        }
        l = l_end;
      }
      wave_lds_sync();
    }
  }
#else
    // staged column-major, X^T[col][row], row = 2 lane + residual row: one 16-byte store per column, contiguous
    // across the lanes (a row-major [row][14] tile puts the 64 lanes of a store on 8 banks)
    {
      dv2* st = reinterpret_cast<dv2*>(stage) + lane;
#pragma unroll
      for (int k = 0; k < 6; k++) st[k * (XRS / 2)] = dv2{Jj[k], Jj[6 + k]}, st[(6 + k) * (XRS / 2)] = dv2{Ji[k], Ji[6 + k]};
      st[12 * (XRS / 2)] = dv2{r[0], r[1]};
    }
    wave_lds_sync();
    const int nact = min(64, ntot - chunk0);
    const int key = (b << 4) | fa;  // frames ascending, start frames ascending inside a frame: equal keys are consecutive
    int l = 0;
    while (l < nact) {
      const int k_cur = __shfl(key, l, 64);
      const int cnt = __popcll(__ballot(act && key == k_cur));
      const int l_end = l + cnt;
      if ((k_cur >> 4) != b_run) {
        end_frame();
        b_run = k_cur >> 4;
      }
      if ((k_cur & 15) != a_run) {
        flush();
        a_run = k_cur & 15;
      }
      // The k index of X^T X is a summation index: lane group drow takes the two rows of factor 4 j + drow for the
      // k-step pair j (one 16-byte read, conflict-free with the 132-row column stride), four pairs = eight MFMAs at a
      // time with the reads issued together, on four independent chains (an MFMA issues every 16 cycles but completes
      // after 64).  Factors outside the run are masked out by their index, so they add exact zeros.
      const int j_end = (l_end + 3) >> 2;
#pragma unroll 1
      for (int j0 = l >> 2; j0 < j_end; j0 += 4) {
        dv2 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) v[u] = *reinterpret_cast<const dv2*>(stage + min(dcol, 12) * XRS + 8 * min(j0 + u, 15) + 2 * drow);
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int f = 4 * (j0 + u) + drow;
          const bool in = dcol < 13 && f >= l && f < l_end;
          const double a0 = in ? v[u][0] : 0.0, a1 = in ? v[u][1] : 0.0;
          if (u & 1) {
            Drun2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, Drun2, 0, 0, 0);
            Drun3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, Drun3, 0, 0, 0);
          } else {
            Drun = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, Drun, 0, 0, 0);
            Drun1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, Drun1, 0, 0, 0);
          }
        }
      }
      l = l_end;
    }
    wave_lds_sync();
  }
#endif
  end_frame();
  return cost;
}
#else
// The same task with every optional member of the problem: staged row [Jj | Ji | r | Jex | Jtd] (20 columns, two 16-wide
// operand tiles), frame 11 = the relocalization frame (its "observations" are the match points, its pose relo_Pose).
// X^T X now has three tiles: D00 = [Jj Ji r]^2 as before, D10 = [Jex Jtd]^T [Jj Ji r] and D11 = [Jex Jtd]^2.
//   D00, per start-frame run a:  (b,a), (a,a) -> PART[b][a], g_a;        total: (b,b), g_b
//   D10, per run:  [Jex Jtd]^T Ji -> PART[b][a][27..68];                 total: [Jex Jtd]^T Jj -> S rows 72..78 x cols 6b.. (owned by
//        this frame), [Jex Jtd]^T r -> PARTX[b][28..34]
//   D11, total: -> PARTX[b][0..27]
// Members that are switched off (estimate_extrinsic / estimate_td == 0) stage exact zeros, so their blocks come out zero.
AVM_DEV double frame_task(const WinCtx&, const avm_options&, int b, int stage_off) {
  const WinCtx& c = lds_ctx();
  const avm_options& o = lds_opt();
  double* lds = LDS();
  double* stage = lds + stage_off;
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int lane = threadIdx.x & 63;
  const int ncov = ids[I_NCOV + b];
  const int32_t* cov = c.cov + b * MAXE;
  Frames fr{lds + L_FR, lds + L_FR + 9 * NFRP};
  const double* ric = ric_of(0);
  const double* xs = lds + L_X;
  const double sqi = o.focal_length / 1.5;
  const bool relo = b == NFRP - 1;
  const bool use_td = c.est_td && !relo;  // the relocalization factors are plain ProjectionFactors (estimator.cpp:783)
  const double exm = c.est_ex ? 1.0 : 0.0;
  const double td = xs[XTD];
  double* W = c.sc + Scratch::W;
  double* PF = c.sc + Scratch::PF;
  double* PART = c.sc + Scratch::PART + (size_t)b * NFR * SPARTW;
  double* PX = c.sc + Scratch::PART + PARTX0 + (size_t)b * PARTX;
  const double* scl = lds + L_SC;
  d4 Dtot = {0, 0, 0, 0}, D00 = {0, 0, 0, 0}, E00 = {0, 0, 0, 0}, D10 = {0, 0, 0, 0}, E10 = {0, 0, 0, 0}, D10tot = {0, 0, 0, 0},
     D11 = {0, 0, 0, 0}, E11 = {0, 0, 0, 0};
  int a_run = -1, pmask = 0;
  double cost = 0;
  const int drow = lane >> 4, dcol = lane & 15;
  auto flush = [&]() {
    if (a_run < 0) return;
    D00 += E00, D10 += E10;
    E00 = E10 = d4{0, 0, 0, 0};
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = drow + 4 * r;
      const double v = D00[r];
      if (row < 6 && dcol >= 6 && dcol < 12) lds[L_S + roff(6 * b + row) + 6 * a_run + (dcol - 6)] = v * (scl[6 * b + row] * scl[6 * a_run + (dcol - 6)]);  // Jj^T Ji (S is written Jacobi-scaled, as in the other builds)
      if (row >= 6 && row < 12) {
        const int i = row - 6;
        if (dcol >= 6 && dcol < 12 && dcol - 6 <= i) PART[a_run * SPARTW + i * (i + 1) / 2 + (dcol - 6)] = v;  // Ji^T Ji (lower)
        if (dcol == 12) PART[a_run * SPARTW + 21 + i] = v;                                                    // Ji^T r
      }
      if (row < 7 && dcol >= 6 && dcol < 12) PART[a_run * SPARTW + 27 + row * 6 + (dcol - 6)] = D10[r];         // [Jex Jtd]^T Ji
    }
    pmask |= 1 << a_run;
    Dtot += D00, D10tot += D10;
    D00 = D10 = d4{0, 0, 0, 0};
  };
  for (int chunk0 = 0; chunk0 < ncov; chunk0 += 64) {
    const int idx = chunk0 + lane;
    const bool act = idx < ncov;
    const int e = cov[min(idx, ncov - 1)];
    const int fa = ids[I_FSTART + e];
    const int s0 = ids[I_FOBS + e], s = s0 + (b - fa);
    double ob[4];
    ob[0] = c.obs[2 * s0], ob[1] = c.obs[2 * s0 + 1];
    if (relo)
      ob[2] = c.relo_xy[2 * min(idx, ncov - 1)], ob[3] = c.relo_xy[2 * min(idx, ncov - 1) + 1];
    else
      ob[2] = c.obs[2 * s], ob[3] = c.obs[2 * s + 1];
    double ai[4] = {0, 0, 0, 0}, aj[4] = {0, 0, 0, 0};
    if (use_td) {
#pragma unroll
      for (int k = 0; k < 4; k++) ai[k] = c.aux[4 * s0 + k], aj[k] = c.aux[4 * s + k];
      td_shift(ob, ai, aj, td, o.tr, o.row);
    }
    double r[2] = {0, 0}, Ji[12], Jj[12], Je[2] = {0, 0}, Jx[12], Jt[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 12; k++) Ji[k] = 0, Jj[k] = 0, Jx[k] = 0;
    if (act) {
      cost += proj_eval<true>(xs, fr, ric, ric + 9, ob[0], ob[1], ob[2], ob[3], xs[XLAM + e], fa, b, sqi, o.cauchy_a, true, r, Ji, Jj, Je, Jx,
                              Jt, ai[0], ai[1], aj[0], aj[1]);
#pragma unroll
      for (int k = 0; k < 12; k++) Jx[k] *= exm;
      if (!use_td) Jt[0] = Jt[1] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        W[(6 * b + k) * WLE + e] = Jj[k] * Je[0] + Jj[6 + k] * Je[1];
        if (k >= 3) PF[(k * NFRP + b) * WLE + e] = Ji[k] * Je[0] + Ji[6 + k] * Je[1];  // (k < 3: minus W's entry, see the base build's frame task)
        PF[((8 + k) * NFRP + b) * WLE + e] = Jx[k] * Je[0] + Jx[6 + k] * Je[1];
      }
      PF[(6 * NFRP + b) * WLE + e] = Je[0] * Je[0] + Je[1] * Je[1];
      PF[(7 * NFRP + b) * WLE + e] = Je[0] * r[0] + Je[1] * r[1];
      PF[(14 * NFRP + b) * WLE + e] = Jt[0] * Je[0] + Jt[1] * Je[1];
    }
    // The staging tile holds HALF a chunk (lanes 0-31 stage and the wavefront multiplies, then lanes 32-63: the scheme of the throughput build
    // and of marg_frame_task).  A run that straddles the two halves simply continues: the switch below only acts on a new start frame.
    const int nact = min(64, ncov - chunk0);
    const int fav = act ? fa : -1;
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      const int h0 = 32 * half, lim = min(nact, h0 + 32);
      if (h0 >= nact) break;  // (uniform)
      if ((lane >> 5) == half) {
        dv2* st = reinterpret_cast<dv2*>(stage) + (lane & 31);
#pragma unroll
        for (int k = 0; k < 6; k++) {
          st[k * (XRS_X / 2)] = dv2{Jj[k], Jj[6 + k]};
          st[(6 + k) * (XRS_X / 2)] = dv2{Ji[k], Ji[6 + k]};
          st[(13 + k) * (XRS_X / 2)] = dv2{Jx[k], Jx[6 + k]};
        }
        st[12 * (XRS_X / 2)] = dv2{r[0], r[1]};
        st[19 * (XRS_X / 2)] = dv2{Jt[0], Jt[1]};
      }
      wave_lds_sync();
      int l = h0;
      while (l < lim) {
        const int a_cur = __shfl(fav, l, 64);
        const int l_end = min(l + __popcll(__ballot(act && fa == a_cur && lane >= l)), lim);
        if (a_cur != a_run) {
          flush();
          a_run = a_cur;
        }
        const int j_end = (l_end - h0 + 3) >> 2;
#pragma unroll 1
        for (int j0 = (l - h0) >> 2; j0 < j_end; j0 += 4) {
          // D10 / D11 have seven rows ([Jex Jtd]): two four-row strips on v_mfma_f64_4x4x4 each (18 cycles of the FP64 pipe an issue against
          // 64; schur_strip4's operand layout: A = row li % 4 of the strip in every quad, B as the 16 x 16 tile takes it, D = register r of the
          // tile's accumulator for strip r) - 128 + 8 x 18 = 272 instead of 384 cycles per step: 5.78 -> 5.60 ms per 1024 windows (round 5)
          dv2 u0[4], u1[4], ua[4], ub[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int ro = 8 * min(j0 + u, 7) + 2 * drow;
            u0[u] = *reinterpret_cast<const dv2*>(stage + min(dcol, 12) * XRS_X + ro);
            u1[u] = *reinterpret_cast<const dv2*>(stage + (13 + min(dcol, 6)) * XRS_X + ro);
            ua[u] = *reinterpret_cast<const dv2*>(stage + (13 + (dcol & 3)) * XRS_X + ro);
            ub[u] = *reinterpret_cast<const dv2*>(stage + (13 + min(4 + (dcol & 3), 6)) * XRS_X + ro);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int f = h0 + 4 * (j0 + u) + drow;
            const bool in = f >= l && f < l_end;
            const double a0 = (in && dcol < 13) ? u0[u][0] : 0.0, a1 = (in && dcol < 13) ? u0[u][1] : 0.0;
            const double x0 = (in && dcol < 7) ? u1[u][0] : 0.0, x1 = (in && dcol < 7) ? u1[u][1] : 0.0;
            const double p0 = in ? ua[u][0] : 0.0, p1 = in ? ua[u][1] : 0.0;
            const double q0 = (in && (dcol & 3) < 3) ? ub[u][0] : 0.0, q1 = (in && (dcol & 3) < 3) ? ub[u][1] : 0.0;
            D00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, D00, 0, 0, 0);
            D10[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(p0, a0, D10[0], 0, 0, 0), D10[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(q0, a0, D10[1], 0, 0, 0);
            D11[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(p0, x0, D11[0], 0, 0, 0), D11[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(q0, x0, D11[1], 0, 0, 0);
            E00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, E00, 0, 0, 0);
            E10[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(p1, a1, E10[0], 0, 0, 0), E10[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(q1, a1, E10[1], 0, 0, 0);
            E11[0] = __builtin_amdgcn_mfma_f64_4x4x4f64(p1, x1, E11[0], 0, 0, 0), E11[1] = __builtin_amdgcn_mfma_f64_4x4x4f64(q1, x1, E11[1], 0, 0, 0);
          }
        }
        l = l_end;
      }
      wave_lds_sync();
    }
  }
  flush();
  D11 += E11;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int row = drow + 4 * r;
    if (row < 6 && dcol <= row) lds[L_S + roff(6 * b + row) + 6 * b + dcol] = Dtot[r] * (scl[6 * b + row] * scl[6 * b + dcol]);   // (b,b) lower
    if (row < 6 && dcol == 12) lds[L_G + 6 * b + row] = Dtot[r];                         // g_b (the gradient is scaled afterwards, as a vector)
    if (row < 7) {
      if (dcol < 6) lds[L_S + roff(XC_EX + row) + 6 * b + dcol] = D10tot[r] * (scl[XC_EX + row] * scl[6 * b + dcol]);  // ([ex td], pose b)
      if (dcol == 12) PX[28 + row] = D10tot[r];                                          // [Jex Jtd]^T r
      if (dcol <= row) PX[row * (row + 1) / 2 + dcol] = D11[r];                          // ([ex td], [ex td]) lower
    }
  }
  if (lane == 0) ids[I_PMASK + b] = pmask;
  return cost;
}
#endif

// One wavefront, one IMU factor i: J = sqrt_info * [r | J_raw] (15 x 31) and its Gram matrix on v_mfma_f64_16x16x4,
// then S += J^T J (lower), g += J^T r; returns 0.5 r^T r on lane 0 (0 elsewhere).
// The two 16-column accumulator tiles of J are, register for register, both the A operand (J^T) and the B operand
// (J) of the Gram products, so nothing moves between the two steps.  Factors sharing a frame must not run
// concurrently (the caller alternates even / odd factors).
struct ImuOperands {
  double ua[4], b0[4], b1[4];
};
// operands of factor i: clamped unconditional loads (issued for both factors of a wavefront before the first is used)
AVM_DEV void imu_factor_load(int i, ImuOperands& o) {
  const WinCtx& c = lds_ctx();
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  gcdouble* U = c.psqrt + i * 225;                 // upper triangular, zeros stored below the diagonal
  gcdouble* raw = c.sc + Scratch::IJRAW + i * 465; // [15][31]: column 0 = residual, 1..30 = Jacobian
  const int lic = min(li, 14);
  // The combined columns are taken in the order residual | pose i | pose i + 1 | speed-bias i | speed-bias i + 1, the
  // order of the state columns themselves (12 consecutive pose columns, 18 consecutive speed-bias entries), so that the scatter of
  // imu_factor_mfma needs no ordering of (row, column) and its offsets are linear in i.  raw's own order is pose i | sb i | pose i + 1 | sb i + 1.
  auto rawcol = [](int cc) { return cc <= 6 ? cc : (cc <= 12 ? cc + 9 : (cc <= 21 ? cc - 6 : cc)); };
  const int c0 = rawcol(li), c1 = rawcol(16 + lic);
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const int k = min(lk + 4 * m, 14);
    o.ua[m] = U[lic * 15 + k], o.b0[m] = raw[k * 31 + c0], o.b1[m] = raw[k * 31 + c1];
  }
#pragma unroll
  for (int m = 0; m < 4; m++) {
    const bool kv = lk + 4 * m < 15;
    o.ua[m] = (li < 15 && kv) ? o.ua[m] : 0.0;
    o.b0[m] = kv ? o.b0[m] : 0.0;
    o.b1[m] = (kv && li < 15) ? o.b1[m] : 0.0;
  }
}

AVM_DEV double imu_factor_mfma(const WinCtx&, int i, const ImuOperands& ops) {
  double* lds = LDS();
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  double ua[4], b0[4], b1[4];
#pragma unroll
  for (int m = 0; m < 4; m++) ua[m] = ops.ua[m], b0[m] = ops.b0[m], b1[m] = ops.b1[m];
  d4 D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < 4; m++) {
    D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[m], b0[m], D0, 0, 0, 0);
    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ua[m], b1[m], D1, 0, 0, 0);
  }
  d4 G00 = {0, 0, 0, 0}, G10 = {0, 0, 0, 0}, G11 = {0, 0, 0, 0};
#pragma unroll
  for (int m = 0; m < 4; m++) {
    G00 = __builtin_amdgcn_mfma_f64_16x16x4f64(D0[m], D0[m], G00, 0, 0, 0);
    G10 = __builtin_amdgcn_mfma_f64_16x16x4f64(D1[m], D0[m], G10, 0, 0, 0);
    G11 = __builtin_amdgcn_mfma_f64_16x16x4f64(D1[m], D1[m], G11, 0, 0, 0);
  }
  // scatter: combined index 0 = residual, p + 1 = local column p.  Branch-free: every lane computes the destination of
  // its (up to) 12 entries - or its private dump slot in the scratch tile - then all reads, all adds, all writes
  // (a predicated LDS read-modify-write is a branch with its own s_waitcnt; 16 of them in a row cost ~2K cycles).
  double half_rr = 0;
  // (round 5) combined index cc: 0 = residual, 1..12 = pose column 6 i + cc - 1, 13..30 = speed-bias entry 9 i + cc - 13 (rows of the compact
  // speed-bias storage, s_off): rows and columns ascend together, and everything but the row term is a constant of the lane
  {
    auto gcol = [&](int cc) { return cc <= 12 ? 6 * i + cc - 1 : SB0 + 9 * i + cc - 13; };  // state column of combined column cc >= 1
#ifdef AVM_TP
    const int psb = reinterpret_cast<const int*>(lds + L_INT)[I_PSB];
    auto dest = [&](int R, int C) {  // R >= C >= 1
      if (R <= 12) return L_S + roff(6 * i + R - 1) + 6 * i + C - 1;
      const int qq = R - 13, second = qq >= 9 ? 1 : 0;
      const int row = L_SBC + (9 * i + qq) * SBW;
      if (C > 12) return row + 18 + (C - 13) + 9 - 9 * second;
      return (i + second == psb) ? L_STRIP + (qq - 9 * second) * NPOSE + 6 * i + C - 1 : row + (C - 1) + 6 - 6 * second;
    };
#else
    auto dest = [&](int R, int C) { return L_S + roff(gcol(R)) + gcol(C); };  // R >= C >= 1: the packed triangle
#endif
    // entries of S are written Jacobi-scaled (see frame_task); the gradient is scaled afterwards, as a vector
    auto scl = [&](int g) { return lds[L_SC + g]; };
    const int dump = L_DUMP + lane;
    const double sc0 = li > 0 ? scl(gcol(li)) : 1.0, sc1 = li < 15 ? scl(gcol(16 + li)) : 1.0;
    int off[12];
    double val[12];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int R0 = lk + 4 * r, R1 = 16 + R0;  // combined rows in tile 0 / tile 1 (31 = padding)
      const double sr0 = scl(gcol(max(R0, 1))), sr1 = scl(gcol(min(R1, 30)));
      if (R0 == 0 && li == 0) half_rr = 0.5 * G00[r];
      const bool v00 = R0 > 0 && li <= R0, v10 = R1 < 31, v11 = R1 < 31 && li < 15 && 16 + li <= R1;
      off[3 * r] = !v00 ? dump : (li == 0 ? L_G + gcol(max(R0, 1)) : dest(max(R0, 1), max(li, 1)));
      val[3 * r] = G00[r] * (li == 0 ? 1.0 : sr0 * sc0);
      off[3 * r + 1] = !v10 ? dump : (li == 0 ? L_G + gcol(min(R1, 30)) : dest(min(R1, 30), max(li, 1)));
      val[3 * r + 1] = G10[r] * (li == 0 ? 1.0 : sr1 * sc0);
      off[3 * r + 2] = !v11 ? dump : dest(min(R1, 30), min(16 + li, min(R1, 30)));
      val[3 * r + 2] = G11[r] * (sr1 * sc1);
    }
    double cur[12];
#pragma unroll
    for (int q = 0; q < 12; q++) cur[q] = lds[off[q]];
#pragma unroll
    for (int q = 0; q < 12; q++) lds[off[q]] = cur[q] + val[q];
    return half_rr;
  }
}

// Full evaluation at lds[L_X]: fills S (unscaled H_ff), W, hee, g (unscaled) and returns the cost.
AVM_NOINL double eval_jac(const WinCtx&, const avm_options&) {
  const WinCtx& c = lds_ctx();
  const avm_options& o = lds_opt();
  double* lds = LDS();
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const double* xs = lds + L_X;
  constexpr int xs_off = L_X;
  PROF_T0();
  build_frames(L_X, 0);
  for (int i = t; i < SPP; i += NT) lds[L_S + i] = 0.0;
  for (int i = t; i < VEC; i += NT) lds[L_G + i] = 0.0;
#ifdef AVM_TP
  for (int i = t; i < 6 * ACCW; i += NT) lds[L_ACC + i] = 0.0;  // the frame tasks' E^T E / E^T r accumulators (wavefront 0's are lds[L_HEE], lds[L_G + NF])
  for (int i = t; i < 152; i += NT) lds[L_HEE + i] = 0.0;
#endif
  if (t < NFRP) ids[I_PMASK + t] = 0;
  double* IJR = c.sc + Scratch::IJRAW;  // (zeroed once per window: imu_raw rewrites the same entries every time)
  __syncthreads();
  Frames fr{lds + L_FR, lds + L_FR + 9 * NFRP};
  double acc = 0;
  // ---- phase A: projection factors (wavefronts 0..ASM_WAVES-1) || raw IMU Jacobians (the next wavefront) || the prior
  const long long pa__ = c.prof ? clock64() : 0;
#ifdef AVM_TP
  constexpr int IMUW = 2;  // (every wavefront assembles; wavefronts 2 and 3 get fewer frames and take the raw IMU Jacobians and the prior afterwards)
#else
  constexpr int IMUW = ASM_WAVES;
#endif
  if (wv < ASM_WAVES) {
#ifdef AVM_X
    for (int b = 1; b < NFRP; b++)
      if (ids[I_FRW + b] == wv) acc += frame_task(c, o, b, L_S + SPP + wv * XSTG);
#else
    AVM_PRIO_BULK();
    acc += frame_task(c, o, wv, L_S + SPP + wv * XSTG);  // all the frames of this wavefront as one list
    AVM_PRIO_LIGHT();
#endif
  }
#ifdef AVM_TP
  if (wv == IMUW && lane < 10) {
#else
  else if (wv == ASM_WAVES && lane < 10) {
#endif
    const int i = lane;
    if (c.psum[i] <= o.max_sum_dt)
      imu_raw<true>(xs, fr.R, o, c.pdelta + i * 10, c.pjac + i * 225, c.psum[i], c.lba + i * 3, c.lbg + i * 3, i, IJR + i * 465);
  }
  // ... || the prior (dx, residual, cost, J0^T r_p) on the last wavefront, which has the lightest load of phase A
#ifdef AVM_X
  if (wv == NT / 64 - 1 && c.pn > 0) {
    const double pc = prior_wave<true>(xs_off, 0, c.pn, L_DXP);
    if (lane == 0) acc += pc;
  }
#else
  // (the raw-IMU wavefront takes the last two fifths of the prior's rows once it is done: each of the two reads J0 along its
  //  own rows only)
  if (wv >= IMUW && c.pn > 0) {
    const int h = (3 * c.pn + 2) / 5;
    const double pc = wv == IMUW ? prior_wave<true>(xs_off, h, c.pn, L_DX2) : prior_wave<true>(xs_off, 0, h, L_DXP);
    if (lane == 0) acc += pc;
  }
#endif
  if (c.prof && lane == 0) c.prof[48 + wv] += clock64() - pa__;  // this wavefront's busy time in phase A
  __syncthreads();
  PROF(c, 0);
  // ---- phase B: per-feature sums over the start pose, diagonal blocks, pose gradient
  {
    double* W = c.sc + Scratch::W;
    const double* PF = c.sc + Scratch::PF;
    // sums over the feature's own factors: one thread per (quantity, feature), features along the lanes
    // (the W blocks of frames that do not observe a feature were zeroed once, at window load)
    // (every round's loads are requested before the first sum: one trip to the slot's memory instead of one per round)
#ifdef AVM_TP
    constexpr int NQB = 6;  // (E^T E and E^T r come out of the frame tasks' accumulators in LDS: below)
#else
    constexpr int NQB = NQ;
#endif
    constexpr int NRND = (MAXE * NQB + NT - 1) / NT;
    double pv[NRND][NFR - 1];
#ifdef AVM_X
    double prl[NRND];
#endif
#pragma unroll
    for (int u = 0; u < NRND; u++) {
      const int idx = min(t + u * NT, max(c.nf * NQB - 1, 0));
      const int q = idx / max(c.nf, 1), e = idx - q * c.nf;
      // (a window without features has no table entry to read: the clamped loads then stay at the start of the region)
      const int a = c.nf > 0 ? ids[I_FSTART + e] : 0, no = c.nf > 0 ? ids[I_FNOBS + e] : 0;
      // + k * stride : the factor observed in frame a + k.  q < 3 (Ji_t^T Je): minus the observing frame's E^T F entry, read out of W (frame_task)
      const double* P = q < 3 ? W + (6 * a + q) * WLE + e : PF + (q * NFRP + a) * WLE + e;
      const int pst = q < 3 ? 6 * WLE : WLE;
      // all (<= 10) loads in flight, clamped to the feature's last observation and masked; same pairing of the
      // partial sums as a sequential two-accumulator loop
#pragma unroll
      for (int k = 1; k < NFR; k++) pv[u][k - 1] = P[min(k, max(no - 1, 0)) * pst];
#ifdef AVM_X
      prl[u] = q < 3 ? W[(6 * (NFRP - 1) + q) * WLE + e] : PF[(q * NFRP + (NFRP - 1)) * WLE + e];
#endif
    }
#ifndef AVM_X
    // the partial (a,a) blocks of the frame tasks, summed further down, are requested now as well
#ifdef AVM_TP
    constexpr int NPR = (NFR * 27 + NT - 1) / NT;  // 297 sums on 256 threads: two rounds
    double pp[NPR][NFR - 1];
#pragma unroll
    for (int u = 0; u < NPR; u++) {
      const int tt = min(t + u * NT, NFR * 27 - 1);
      const int f = tt / 27, q = tt % 27;
#pragma unroll
      for (int b = 1; b < NFR; b++) pp[u][b - 1] = (c.sc + Scratch::PART)[((size_t)b * NFR + f) * 27 + q];  // unconditional, masked below
    }
#else
    double pp[NFR - 1];
    if (t < NFR * 27) {
      const int f = t / 27, q = t % 27;
#pragma unroll
      for (int b = 1; b < NFR; b++) pp[b - 1] = (c.sc + Scratch::PART)[((size_t)b * NFR + f) * 27 + q];  // unconditional, masked below
    }
#endif
#endif
#pragma unroll
    for (int u = 0; u < NRND; u++) {
      const int idx = t + u * NT;
      if (idx >= c.nf * NQB) break;
      const int q = idx / c.nf, e = idx - q * c.nf;
      const int a = ids[I_FSTART + e], no = ids[I_FNOBS + e];
      double s0a = 0, s1a = 0;
#pragma unroll
      for (int k = 1; k < NFR; k++) {
        const double v = k < no ? pv[u][k - 1] : 0.0;
        if (k & 1) s0a += v; else s1a += v;
      }
#ifdef AVM_X
      // + the feature's relocalization factor (frame 11; its slots were zeroed at window load for unmatched features)
      const double sraw = (s0a + s1a) + (c.relo_n > 0 ? prl[u] : 0.0);
      const double sacc = q < 3 ? -sraw : sraw;  // (the W entries are minus the products summed here: exact)
      if (q < 6)
        W[(6 * a + q) * WLE + e] = sacc;
      else if (q == 6)
        lds[L_HEE + e] = sacc;
      else if (q == 7)
        lds[L_G + NF + e] = sacc;
      else
        W[(XC_EX + (q - 8)) * WLE + e] = sacc;  // E^T F of the ex_pose (6) and td (1) columns
#else
      const double sacc = q < 3 ? -(s0a + s1a) : s0a + s1a;  // (the W entries are minus the products summed here: exact)
      if (q < 6)
        W[(6 * a + q) * WLE + e] = sacc;
      else if (q == 6)
        lds[L_HEE + e] = sacc;
      else
        lds[L_G + NF + e] = sacc;
#endif
    }
#ifdef AVM_TP
    for (int e = t; e < c.nf; e += NT) {  // the four wavefronts' accumulators, in a fixed order
      const double* ac = lds + L_ACC + e;
      lds[L_HEE + e] = (lds[L_HEE + e] + ac[0]) + (ac[2 * ACCW] + ac[4 * ACCW]);
      lds[L_G + NF + e] = (lds[L_G + NF + e] + ac[ACCW]) + (ac[3 * ACCW] + ac[5 * ACCW]);
    }
#endif
    const double* PART = c.sc + Scratch::PART;
#ifdef AVM_X
    __syncthreads();  // (the sums below add to blocks other frames' tasks have written: all of phase A is behind the barrier above)
    for (int tt = t; tt < NFR * SPARTW + PARTX; tt += NT) {
      if (tt < NFR * SPARTW) {
        const int f = tt / SPARTW, q = tt % SPARTW;
        double sacc = 0;
        double pp[NFRP - 1];
#pragma unroll
        for (int b = 1; b < NFRP; b++) pp[b - 1] = PART[((size_t)b * NFR + f) * SPARTW + q];
#pragma unroll
        for (int b = 1; b < NFRP; b++)
          if (b > f && (ids[I_PMASK + b] & (1 << f))) sacc += pp[b - 1];
        if (q < 21) {
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= q) i++;
          const int j = q - i * (i + 1) / 2;
          lds[L_S + roff(6 * f + i) + 6 * f + j] += sacc * (lds[L_SC + 6 * f + i] * lds[L_SC + 6 * f + j]);
        } else if (q < 27) {
          lds[L_G + 6 * f + (q - 21)] += sacc;
        } else {
          lds[L_S + roff(XC_EX + (q - 27) / 6) + 6 * f + (q - 27) % 6] += sacc * (lds[L_SC + XC_EX + (q - 27) / 6] * lds[L_SC + 6 * f + (q - 27) % 6]);  // ([ex td], start pose f)
        }
      } else {
        const int q = tt - NFR * SPARTW;
        double sacc = 0;
        for (int b = 1; b < NFRP; b++) sacc += PART[PARTX0 + (size_t)b * PARTX + q];  // (a frame without factors wrote zeros)
        if (q < 28) {
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= q) i++;
          lds[L_S + roff(XC_EX + i) + XC_EX + (q - i * (i + 1) / 2)] = sacc * (lds[L_SC + XC_EX + i] * lds[L_SC + XC_EX + (q - i * (i + 1) / 2)]);
        } else {
          lds[L_G + XC_EX + (q - 28)] = sacc;
        }
      }
    }
    __syncthreads();
    // members that are switched off: unit diagonal, nothing else (their rows / columns stay zero), so their step is exactly 0
    if (t < 13) {
      const int col = NFR * 6 + t;  // relo 66..71 | ex 72..77 | td 78
      const bool on = t < 6 ? c.relo_n > 0 : (t < 12 ? c.est_ex != 0 : c.est_td != 0);
      if (!on) lds[L_S + roff(col) + col] = lds[L_SC + col] * lds[L_SC + col];  // (1.0, Jacobi-scaled like every other entry)
    }
#elif defined(AVM_TP)
#pragma unroll
    for (int u = 0; u < NPR; u++) {
      const int tt = t + u * NT;
      if (tt >= NFR * 27) break;
      const int f = tt / 27, q = tt % 27;
      double sacc = 0;
#pragma unroll
      for (int b = 1; b < NFR; b++)
        if (b > f && (ids[I_PMASK + b] & (1 << f))) sacc += pp[u][b - 1];
      if (q < 21) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= q) i++;
        const int j = q - i * (i + 1) / 2;
        lds[L_S + roff(6 * f + i) + 6 * f + j] += sacc * (lds[L_SC + 6 * f + i] * lds[L_SC + 6 * f + j]);
      } else {
        lds[L_G + 6 * f + (q - 21)] += sacc;
      }
    }
#else
    if (t < NFR * 27) {
      const int f = t / 27, q = t % 27;
      double sacc = 0;
#pragma unroll
      for (int b = 1; b < NFR; b++)
        if (b > f && (ids[I_PMASK + b] & (1 << f))) sacc += pp[b - 1];
      if (q < 21) {
        int i = 0;
        while ((i + 1) * (i + 2) / 2 <= q) i++;
        const int j = q - i * (i + 1) / 2;
        lds[L_S + roff(6 * f + i) + 6 * f + j] += sacc * (lds[L_SC + 6 * f + i] * lds[L_SC + 6 * f + j]);
      } else {
        lds[L_G + 6 * f + (q - 21)] += sacc;
      }
    }
#endif
  }
  PROF(c, 1);
  // phase D's operands (sqrt_info, the raw Jacobians wave ASM_WAVES left in the slot during phase A) and phase E's packed
  // prior are fetched now: their trip to the slot's memory overlaps the zeroing and the barriers in between
#ifdef AVM_TP
  // ten factors on four wavefronts: three rounds of factors that share no frame - {0 2 4 6}, {8 1 3 5}, {7 9}
  constexpr int NIMR = 3;
  auto imu_of = [&](int rd) { return rd == 0 ? 2 * wv : (rd == 1 ? (wv == 0 ? 8 : 2 * wv - 1) : (wv == 0 ? 7 : (wv == 1 ? 9 : -1))); };
  ImuOperands io[NIMR];
#pragma unroll
  for (int rd = 0; rd < NIMR; rd++)
    if (imu_of(rd) >= 0) imu_factor_load(imu_of(rd), io[rd]);
  constexpr int NIT = 12;  // rounds fetched ahead: they cover a prior of up to 77 rows (the tail of a larger one is added straight from the slot)
#else
  ImuOperands io[2];
  if (wv < 5) imu_factor_load(2 * wv, io[0]), imu_factor_load(2 * wv + 1, io[1]);
  constexpr int NIT = (HPK_MAX + NT - 1) / NT;  // 10 rounds cover the largest prior
#endif
  const int npk = c.pn * (c.pn + 1) / 2;
  int dd[NIT];
  double hv[NIT];
  if (c.pn > 0) {
    gcdouble* HPk = c.sc + Scratch::HP;
    const gint* dst = reinterpret_cast<const gint*>(c.sc + Scratch::HP + HPK_MAX);
#pragma unroll
    for (int u = 0; u < NIT; u++) {
      const int idx = min(t + u * NT, npk - 1);
      dd[u] = dst[idx], hv[u] = HPk[idx];
    }
  }
  // rows 66.. of S (the staging area is dead now)
#ifdef AVM_TP
  for (int i = t; i < 99 * SBW + 9 * NPOSE; i += NT) lds[L_SBC + i] = 0.0;  // the speed-bias rows in structural form + the prior's strip
#else
  for (int i = SPP + t; i < SROWS; i += NT) lds[L_S + i] = 0.0;
#endif
  __syncthreads();
  PROF(c, 3);
  // ---- phase D: IMU factors on MFMA, one wavefront per factor; even factors then odd ones (neighbours share a frame)
  {
#ifdef AVM_TP
#pragma unroll
    for (int rd = 0; rd < NIMR; rd++) {
      const int i = imu_of(rd);
      if (i >= 0 && c.psum[max(i, 0)] <= o.max_sum_dt) acc += imu_factor_mfma(c, i, io[rd]);
      __syncthreads();
    }
#else
#pragma unroll
    for (int par = 0; par < 2; par++) {
      if (wv < 5) {
        const int i = 2 * wv + par;
        if (c.psum[i] <= o.max_sum_dt) acc += imu_factor_mfma(c, i, io[par]);
      }
      __syncthreads();
    }
#endif
  }
  PROF(c, 7);
  // ---- phase E: prior  H += Hp (packed values + destinations prepared once per solve), g += J0^T r_p
  if (c.pn > 0) {
    // H += Hp (packed values + destinations, fetched above), g += J0^T r_p (phase A left it in lds[L_DXP])
    {
#pragma unroll
      for (int u = 0; u < NIT; u++)
        if (t + u * NT < npk && dd[u] >= 0) lds[L_S + dd[u]] += hv[u];
#ifdef AVM_TP
      for (int idx = t + NIT * NT; idx < npk; idx += NT) {
        const int d = reinterpret_cast<const gint*>(c.sc + Scratch::HP + HPK_MAX)[idx];
        if (d >= 0) lds[L_S + d] += (c.sc + Scratch::HP)[idx];
      }
#endif
    }
    {
      const int* pidx = ids + I_PIDX;
#ifdef AVM_X
      if (t < c.pn && pidx[t] >= 0) lds[L_G + pidx[t]] += lds[L_DXP + t];
#else
      if (t < c.pn && pidx[t] >= 0) lds[L_G + pidx[t]] += lds[L_DXP + t] + lds[L_DX2 + t];
#endif
    }
  }
  const double cost = block_sum1(acc);
  __syncthreads();
  PROF(c, 8);
  return cost;
}

// || J' u ||^2 with J' the Jacobi-scaled Jacobian, u in lds[L_ST] (scaled space), at state lds[L_X].
// Only needed when the Gauss-Newton step leaves the trust region (Cauchy point), so the factors are
// simply re-evaluated here instead of keeping their Jacobians around.
AVM_NOINL double jac_times_vec_sq(const WinCtx&, const avm_options&) {
  const WinCtx& c = lds_ctx();
  const avm_options& o = lds_opt();
  double* lds = LDS();
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int t = threadIdx.x;
  const double* u = lds + L_ST;
  const double* scl = lds + L_SC;
  const double* xs = lds + L_X;
  Frames fr{lds + L_FR, lds + L_FR + 9 * NFRP};
  const double* ric = ric_of(0);
  const double sqi = o.focal_length / 1.5;
  double acc = 0;
#ifdef AVM_X
  // regular factors, then the relocalization factors (slot index >= nobs_tot: match k against frame 11)
  for (int s = t; s < c.nobs_tot + c.relo_n; s += NT) {
    const bool relo = s >= c.nobs_tot;
    const int e = relo ? c.cov[(NFRP - 1) * MAXE + (s - c.nobs_tot)] : min(max(c.osf[s], 0), c.nf - 1);
    const int s0 = ids[I_FOBS + e];
    if (!relo && (s <= s0 || s >= s0 + ids[I_FNOBS + e])) continue;  // first observation, or a hole of the table (see eval_cost)
    const int fa = ids[I_FSTART + e], fb = relo ? NFRP - 1 : fa + (s - s0);
    double ob[4] = {c.obs[2 * s0], c.obs[2 * s0 + 1], 0, 0}, ai[4] = {0, 0, 0, 0}, aj[4] = {0, 0, 0, 0};
    if (relo)
      ob[2] = c.relo_xy[2 * (s - c.nobs_tot)], ob[3] = c.relo_xy[2 * (s - c.nobs_tot) + 1];
    else
      ob[2] = c.obs[2 * s], ob[3] = c.obs[2 * s + 1];
    const bool use_td = c.est_td && !relo;
    if (use_td) {
#pragma unroll
      for (int k = 0; k < 4; k++) ai[k] = c.aux[4 * s0 + k], aj[k] = c.aux[4 * s + k];
      td_shift(ob, ai, aj, xs[XTD], o.tr, o.row);
    }
    double r[2], Ji[12], Jj[12], Je[2], Jx[12], Jt[2];
    proj_eval<true>(xs, fr, ric, ric + 9, ob[0], ob[1], ob[2], ob[3], xs[XLAM + e], fa, fb, sqi, o.cauchy_a, true, r, Ji, Jj, Je, Jx, Jt,
                    ai[0], ai[1], aj[0], aj[1]);
    double y0 = 0, y1 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double va = u[fa * 6 + k] * scl[fa * 6 + k], vb = u[fb * 6 + k] * scl[fb * 6 + k];
      const double vx = c.est_ex ? u[XC_EX + k] * scl[XC_EX + k] : 0.0;
      y0 += Ji[k] * va + Jj[k] * vb + Jx[k] * vx;
      y1 += Ji[6 + k] * va + Jj[6 + k] * vb + Jx[6 + k] * vx;
    }
    const double ve = u[NF + e] * scl[NF + e], vt = use_td ? u[XC_TD] * scl[XC_TD] : 0.0;
    y0 += Je[0] * ve + Jt[0] * vt;
    y1 += Je[1] * ve + Jt[1] * vt;
    acc += y0 * y0 + y1 * y1;
  }
#else
  for (int s = t; s < c.nobs_tot; s += NT) {
    const int e = min(max(c.osf[s], 0), c.nf - 1);
    const int s0 = ids[I_FOBS + e];
    if (s <= s0 || s >= s0 + ids[I_FNOBS + e]) continue;  // first observation, or a hole of the table (see eval_cost)
    const int fa = ids[I_FSTART + e], fb = fa + (s - s0);
    double r[2], Ji[12], Jj[12], Je[2];
    proj_eval<true>(xs, fr, ric, ric + 9, c.obs[2 * s0], c.obs[2 * s0 + 1], c.obs[2 * s], c.obs[2 * s + 1], xs[XLAM + e],
                    fa, fb, sqi, o.cauchy_a, true, r, Ji, Jj, Je);
    double y0 = 0, y1 = 0;
#pragma unroll
    for (int k = 0; k < 6; k++) {
      const double va = u[fa * 6 + k] * scl[fa * 6 + k], vb = u[fb * 6 + k] * scl[fb * 6 + k];
      y0 += Ji[k] * va + Jj[k] * vb;
      y1 += Ji[6 + k] * va + Jj[6 + k] * vb;
    }
    const double ve = u[NF + e] * scl[NF + e];
    y0 += Je[0] * ve;
    y1 += Je[1] * ve;
    acc += y0 * y0 + y1 * y1;
  }
#endif
  // IMU: y = sqrt_info * (raw_J * v), raw Jacobians of the last eval_jac are still in the scratch slot.  Two steps with one trip to
  // the slot each (thread (i, k): row k of raw_J times v, thirty loads in flight; thread (i, r): row r of sqrt_info times that) - as
  // nested loops from k = r every thread made up to fifteen trips of its own, and every row of raw_J v was computed up to 15 times
  double* rvb = lds + L_WCH;  // [150] (the factorization's scratch is dead here)
  if (t < 150) {
    const int i = t / 15, k = t % 15;
    const double* IJR = c.sc + Scratch::IJRAW + i * 465;
    double jv[30];
#pragma unroll
    for (int p = 0; p < 30; p++) jv[p] = IJR[k * 31 + 1 + p];
    double rv = 0;
#pragma unroll
    for (int p = 0; p < 30; p++) {
      const int col = imu_col(i, p);
      rv += jv[p] * (u[col] * scl[col]);
    }
    rvb[t] = rv;
  }
  // the prior's rows meanwhile: J0 row i times v, sixteen loads in flight (behind `pidx[k] >= 0` they were up to 75 trips)
#ifdef AVM_TP
  constexpr int PT0 = 160;  // (256 threads: the prior's rows sit right behind the 150 IMU rows)
#else
  constexpr int PT0 = 192;
#endif
  if (c.pn > 0 && t >= PT0 && t < PT0 + c.pn) {
    const int i = t - PT0;
    const int* pidx = ids + I_PIDX;
    const int pn1 = c.pn - 1;
    double y = 0;
    for (int k0 = 0; k0 < c.pn; k0 += 16) {
      double pj[16];
#pragma unroll
      for (int q = 0; q < 16; q++) pj[q] = c.pJ[(size_t)i * c.ldp + min(k0 + q, pn1)];
#pragma unroll
      for (int q = 0; q < 16; q++) {
        const int ix = pidx[min(k0 + q, pn1)], ic = max(ix, 0);
        y += (k0 + q <= pn1 && ix >= 0) ? pj[q] * (u[ic] * scl[ic]) : 0.0;
      }
    }
    acc += y * y;
  }
  __syncthreads();
  if (t < 150) {
    const int i = t / 15, r = t % 15;
    if (c.psum[i] <= o.max_sum_dt) {
      double ps[15];
#pragma unroll
      for (int k = 0; k < 15; k++) ps[k] = c.psqrt[i * 225 + r * 15 + k];
      double y = 0;
#pragma unroll
      for (int k = 0; k < 15; k++) y += k >= r ? ps[k] * rvb[i * 15 + k] : 0.0;
      acc += y * y;
    }
  }
  return block_sum1(acc);
}

AVM_DEV double readlane_d(double v, int srclane) {  // srclane must be wave-uniform
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, srclane);
  hi = __builtin_amdgcn_readlane(hi, srclane);
  return __hiloint2double(hi, lo);
}

// =====================================================================================================================
// Throughput build, and the latency / extended builds for every window whose prior fits (I_CRFIT): the factorization on REGISTER tiles,
// distributed over four wavefronts (the latency build's other four only keep the barriers company; the extended build, with 13 more dense
// columns = one more tile column, spreads them over all eight), in an elimination order that keeps the factor SPARSE and lets TWO pivot
// chains run at a time (round 6).
//
// Order of elimination - a nested dissection of the speed-bias chain with frame 5's block as the separator:
//     B: the speed-bias blocks of frames 10, 9, .. 6   (45 columns + 3 of padding = tile columns 0, 1, 2)
//     F: the speed-bias blocks of frames 4, 3, .. 0    (45 columns + 3 of padding = tile columns 3, 4, 5)
//     then frame 5's block, the poses and the right-hand side (9 + 66 + 1 = 76 positions = tile columns 6 .. 10).
// Speed-bias block b couples to blocks b - 1 / b + 1 and to the poses b - 1 .. b + 1 (one IMU factor each side).  Eliminated from the
// window's end, B hands each next block the poses b .. 10 as fill and nothing else; F is eliminated from the separator outwards, so
// that the prior's speed-bias block (frame 0: the one block the prior couples to EVERY pose - the strip) comes last of it and its
// dense row fills nothing.  B and F never meet (no factor joins them, and neither does the fill: what joins them is eliminated
// later), so the 16-pivot chains of tile columns t and 3 + t, t = 0, 1, 2, run at the same time on two wavefronts: a factorization is
// EIGHT chain times long (3 + 5) instead of eleven, and the chain - one wavefront, 4.5 K cycles - is what a step lasts.  Of the 66
// upper tiles of the 11 x 11 grid 47 can be nonzero and a factorization takes 91 tile updates (364 MFMAs) where the order poses |
// speed-biases takes 220 (880): with the poses first every speed-bias row fills in completely.  The padding positions are rows of the
// identity.  The rest of the kernel keeps its layout; the permutation happens when the tiles are loaded (tp_offsets) and when the
// solution is written back.
//
// The augmented system [H' + mu D^2, g'; g'^T, .] in that order is cut into 11 x 11 tiles of 16 x 16 and held as its UPPER tiles
// U(k, i), k <= i (U(k, i) = L(i, k)^T once factored), in the accumulator layout of v_mfma_f64_16x16x4: register r of lane
// (lk = lane / 16, lr = lane % 16) is entry (lk + 4 r, lr).  With the k index of a product running as lk + 4 r such a tile IS a B
// operand and, read as an A operand, its transpose (the scheme of prior_chol_kernel, prior_eig.hip), so nothing is transposed or
// moved between lanes.  The right-hand side is position TP_RHS (tile column 10, local column 11): the forward substitution rides along.
//   * which tiles exist is a compile-time table (TPP: the system's tile pattern closed under the elimination's fill); a tile outside
//     it is never loaded, solved, published or updated;
//   * ownership by tile COLUMN (tp_owner): wavefront 3 holds B's columns, wavefront 2 F's - their tiles only ever meet each other, the two
//     chains of chains run there -, the other columns are spread so that a chain's owner has little else to do: at most 13 tiles;
//   * step t (tp_step_piv: the pivot columns {t, 3 + t} for t < 3, then {t + 3}):
//              [owner of a pivot column k] 16-pivot chain on the diagonal tile (through a 2 KB LDS patch into lane = row form: the
//              square-root-free chain of chol_diag_block, L_kk^-T riding along in lanes 16..31) ...................... barrier
//              [every wavefront] W(k, i) = L_kk^-1 U(k, i) for its columns i > k, published to LDS; the owner of a pivot column q of
//              step t + 1 then updates tile (q, q) - it needs its own W(k, q) only -, stages it and starts the chain ... counted
//              [every other wavefront] waits for the four counts, then U(j, i) -= W(k, j)^T W(k, i) for its columns while the chains run
//   * backward substitution L^T x = z by the same steps, last to first: the owner of column i solves x_i from z_i minus the four
//     wavefronts' partial sums, folds x_i into element-wise accumulators E_k += U(k, i) .* x_i (k < i, no reduction), and every
//     wavefront that holds a tile of the next step's rows reduces its E over the 16-lane rows (DPP) into its partial vector: one barrier per step.
// Nothing of the factor ever goes to memory; the LDS traffic is the published rows of W (<= 18 KB per step).
static_assert(NFR == 11 && NF == NPOSE + 99 && (TPT == 11 || TPT == 12), "the elimination order below is written for eleven frames");
constexpr int TP_PAD = -1;
// (TP_M0, TP_P0, TP_RHS - first position of frame 5's block / of the dense columns (poses [, relo_Pose, ex_pose, td]) / the right-hand side - and TPT: at the LDS carve)
constexpr int TP_NBL = TP_RHS - 16 * (TPT - 1);       // state columns in the last tile column: 11 (+ the right-hand side at local column 11)
static_assert(TP_NBL >= 1 && TP_NBL < 16, "the right-hand side fits the last tile column");
// position n of the elimination order -> column of the assembled system (poses | speed-biases; NF = the right-hand side), TP_PAD for padding
__host__ __device__ constexpr int tp_perm(int n) {
  if (n < 45) return NPOSE + 9 * (10 - n / 9) + n % 9;             // B: frames 10 .. 6
  if (n < 48) return TP_PAD;
  if (n < 93) return NPOSE + 9 * (4 - (n - 48) / 9) + (n - 48) % 9;  // F: frames 4 .. 0
  if (n < TP_M0) return TP_PAD;
  if (n < TP_P0) return NPOSE + 45 + (n - TP_M0);                   // frame 5's block
  if (n < TP_RHS) return n - TP_P0;                                 // poses
  return n == TP_RHS ? NF : TP_PAD;
}

// Where the tiles are loaded from: for a pair of positions the LDS offset (in doubles) of the entry - s_off() of the two columns with the
// prior's speed-bias block at frame 0 (the host sends every other prior to the latency form), the right-hand side for position TP_RHS -,
// TP_NONE for a structural zero, TP_ONE for the diagonal of a padding position: the places of the constants 0.0 and 1.0 (with codes to be masked
// the compiler built a branch per entry: 10 K cycles per factorization).
constexpr int TP_NONE = L_ZERO, TP_ONE = L_ONE;
constexpr int tp_off_c(int Rn, int Cn) {
  const int R = tp_perm(Rn), C = tp_perm(Cn);
  if (R == TP_PAD || C == TP_PAD) return Rn == Cn ? TP_ONE : TP_NONE;
  const int hi = R > C ? R : C, lo = R > C ? C : R;
  if (hi == NF) return lo < NF ? L_RHS + lo : TP_NONE;
  if (hi < NPOSE) return L_S + croff(hi) + lo;
  const int q = hi - NPOSE, b = q / 9;
#ifdef AVM_TP
  if (lo < NPOSE) {
    if (b == 0) return L_STRIP + (q - 9 * b) * NPOSE + lo;
    const int p = lo - 6 * (b - 1);
    return p >= 0 && p < 18 ? L_SBC + q * SBW + p : TP_NONE;
  }
  const int p = lo - (NPOSE + 9 * (b - 1));
  return p >= 0 && p < 18 ? L_SBC + q * SBW + 18 + p : TP_NONE;
#else
  // (latency build: the same structure, the places are those of the packed triangle)
  const int p = lo < NPOSE ? lo - 6 * (b - 1) : lo - (NPOSE + 9 * (b - 1));
  return (lo < NPOSE && b == 0) || (p >= 0 && p < 18) ? L_S + croff(hi) + lo : TP_NONE;
#endif
}

// Tile pattern of the system in elimination order, [k][i] with k <= i: h = the assembled system can be nonzero there (tp_off_c names a place),
// nz = h closed under the fill of the tile-level elimination (which is what the scalar elimination fills, aggregated:
// tests/test_tp_pattern.py states both in numpy).
struct TpPattern {
  bool h[TPT][TPT], nz[TPT][TPT];
};
constexpr TpPattern tp_make_pattern() {
  TpPattern P{};
  for (int k = 0; k < TPT; k++)
    for (int i = k; i < TPT; i++) {
      bool any = false;
      for (int a = 0; a < 16 && !any; a++)
        for (int b = 0; b < 16 && !any; b++) any = tp_off_c(16 * k + a, 16 * i + b) != TP_NONE;
      P.h[k][i] = P.nz[k][i] = any;
    }
  for (int k = 0; k < TPT; k++)
    for (int j = k + 1; j < TPT; j++)
      if (P.nz[k][j])
        for (int i = j; i < TPT; i++)
          if (P.nz[k][i]) P.nz[j][i] = true;
  return P;
}
constexpr TpPattern TPP = tp_make_pattern();
__host__ __device__ constexpr bool tp_nz(int k, int i) { return k <= i && TPP.nz[k][i]; }
// the steps of the factorization: pivot columns {t, 3 + t} for t < 3 (B and F side by side), then one column per step
constexpr int TP_NSTEP = TPT - 3;  // 8 (9)
__host__ __device__ constexpr int tp_step_np(int t) { return t < 3 ? 2 : 1; }
__host__ __device__ constexpr int tp_step_piv(int t, int a) { return t < 3 ? (a == 0 ? t : t + 3) : t + 3; }
__host__ __device__ constexpr int tp_step_of(int k) { return k < 3 ? k : k - 3; }
__host__ __device__ constexpr int tp_slot_of(int k) { return k >= 3 && k < 6 ? 1 : 0; }           // which of its step's pivot columns (the patch it uses)
__host__ __device__ constexpr int tp_buf(int k) { return 2 * (tp_step_of(k) & 1) + tp_slot_of(k); }  // its L^-T buffer: the next step's chains write the other pair
__host__ __device__ constexpr bool tp_is_piv(int t, int q) {  // is q a pivot column of step t ?
  return t >= 0 && t < TP_NSTEP && (tp_step_piv(t, 0) == q || (tp_step_np(t) == 2 && tp_step_piv(t, 1) == q));
}
__host__ __device__ constexpr bool tp_steps_ok() {  // the two pivot columns of a step share no tile, and a column's rows all belong to earlier steps
  for (int t = 0; t < 3; t++)
    if (tp_nz(t, t + 3)) return false;
  for (int i = 0; i < TPT; i++)
    for (int k = 0; k < i; k++)
      if (tp_nz(k, i) && tp_step_of(k) >= tp_step_of(i)) return false;
  return true;
}
static_assert(tp_steps_ok(), "B and F must not meet");
__host__ __device__ constexpr int tp_owner(int i) {
#ifdef AVM_X
  // eight wavefronts: B and F as below, every later column a wavefront of its own (a chain's owner has nothing else in the rows of the step before)
  return i < 3 ? 3 : (i < 6 ? 2 : (i == 6 ? 0 : (i == 7 ? 1 : i - 4)));
#else
  // (build/dev: the assignment that leaves the owner of a step's pivot columns the least other work in the step before)
  return i < 3 ? 3 : (i < 7 ? 2 : (i < 9 ? 0 : (i == 9 ? 3 : 1)));
#endif
}
__host__ __device__ constexpr int tp_ncol(int i) {  // tiles of column i
  int n = 0;
  for (int k = 0; k <= i; k++) n += tp_nz(k, i) ? 1 : 0;
  return n;
}
__host__ __device__ constexpr int tp_idx(int wv, int k, int i) {  // index of tile (k, i) in wavefront wv's array
  int n = 0;
  for (int c = 0; c < i; c++) n += tp_owner(c) == wv ? tp_ncol(c) : 0;
  for (int q = 0; q < k; q++) n += tp_nz(q, i) ? 1 : 0;
  return n;
}
__host__ __device__ constexpr int tp_ntiles(int wv) { return tp_idx(wv, 0, TPT); }
__host__ __device__ constexpr int tp_nrow(int k) {  // tiles of row k beside the diagonal
  int n = 0;
  for (int c = k + 1; c < TPT; c++) n += tp_nz(k, c) ? 1 : 0;
  return n;
}
__host__ __device__ constexpr int tp_wslot(int k, int i) {  // slot of W(k, i) among its step's published tiles
  int n = tp_slot_of(k) == 1 ? tp_nrow(tp_step_piv(tp_step_of(k), 0)) : 0;
  for (int c = k + 1; c < i; c++) n += tp_nz(k, c) ? 1 : 0;
  return n;
}
__host__ __device__ constexpr int tp_max_wslots() {
  int m = 0;
  for (int t = 0; t < TP_NSTEP; t++) {
    int n = 0;
    for (int a = 0; a < tp_step_np(t); a++) n += tp_nrow(tp_step_piv(t, a));
    m = n > m ? n : m;
  }
  return m;
}
static_assert(tp_max_wslots() <= TP_WSLOTS, "the published rows of W fit their LDS slots");
__host__ __device__ constexpr bool tp_row_held(int wv, int k) {  // does wavefront wv hold a tile (k, i), i > k ?
  for (int i = k + 1; i < TPT; i++)
    if (tp_owner(i) == wv && tp_nz(k, i)) return true;
  return false;
}
__host__ __device__ constexpr bool tp_owns_piv(int wv, int t) {  // does wavefront wv own a pivot column of step t ?
  for (int a = 0; t >= 0 && t < TP_NSTEP && a < tp_step_np(t); a++)
    if (tp_owner(tp_step_piv(t, a)) == wv) return true;
  return false;
}
__host__ __device__ constexpr bool tp_owners_ok() {  // a wavefront runs one chain at a time
  for (int t = 0; t < 3; t++)
    if (tp_owner(tp_step_piv(t, 0)) == tp_owner(tp_step_piv(t, 1))) return false;
  return true;
}
static_assert(tp_owners_ok(), "the two chains of a step run on two wavefronts");

AVM_DEV int tp_perm_dev(int n) {
  const int m = n - 48;
  const int b = NPOSE + 9 * 10 - 9 * (n / 9) + n % 9, f = NPOSE + 9 * 4 - 9 * (m / 9) + m % 9;
  return n < 45 ? b : (n < 48 ? TP_PAD : (n < 93 ? f : (n < TP_M0 ? TP_PAD : (n < TP_P0 ? NPOSE + 45 + (n - TP_M0) : (n < TP_RHS ? n - TP_P0 : (n == TP_RHS ? NF : TP_PAD))))));
}

// The offsets as a table in the code object's constant data, evaluated at compile time ([tile][lane][register]: one 8-byte load per lane and
// tile).  The generic form - position -> column, s_off with its division and branches, an LDS read behind each - was 28 K cycles per factorization.
__host__ __device__ constexpr int tp_h_ord(int k, int i) {  // ordinal of tile (k, i) among the tiles with TPP.h, column by column
  int n = 0;
  for (int c = 0; c < TPT; c++)
    for (int q = 0; q <= c; q++) {
      if (c == i && q == k) return n;
      n += TPP.h[q][c] ? 1 : 0;
    }
  return n;
}
constexpr int TP_NH = tp_h_ord(TPT, TPT);
struct TpOffsets {
  unsigned short o[TP_NH][64][4];
};
constexpr TpOffsets tp_make_offsets() {
  TpOffsets t{};
  for (int i = 0; i < TPT; i++)
    for (int k = 0; k <= i; k++)
      if (TPP.h[k][i])
        for (int lane = 0; lane < 64; lane++)
          for (int r = 0; r < 4; r++) t.o[tp_h_ord(k, i)][lane][r] = (unsigned short)tp_off_c(16 * k + (lane >> 4) + 4 * r, 16 * i + (lane & 15));
  return t;
}
__device__ const TpOffsets tp_offsets = tp_make_offsets();

// 16-pivot chain on the diagonal block in LDS patch `patch` ([row][16], symmetric): chol_diag_block with the patch as its source and
// destination.  Leaves L~ (lower, unscaled: times sqrt(d_c) per column c, the pivot d_c on the diagonal) in the patch and
// L~^-T with 1 / sqrt(d_c) behind it in buffer `buf`.
AVM_DEV void tp_diag_chain(int nb, int patch, int buf, int stamp) {
  constexpr int NB = 16;
  double* lds = LDS();
  const int r = threadIdx.x & 63;
  __builtin_amdgcn_s_setprio(3);
  double a[NB];
  const bool idl = (r & 48) == 16;
  const int rc = min(r, nb - 1);
  double* row = lds + L_PATCH + patch * (16 * TP_PS) + (rc & 15) * TP_PS;
  {
#pragma unroll
    for (int k = 0; k < NB; k++) a[k] = row[k];
#pragma unroll
    for (int k = 0; k < NB; k++) a[k] = idl ? ((r & 15) == k ? 1.0 : 0.0) : a[k];  // lanes 16..31: the identity's rows
  }
  wave_lds_sync();  // (every lane holds its row: the stores below go to the same patch)
  double uprev = 0.0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    if (j > 0) a[j] = fma(-uprev, readlane_d(a[j - 1], j), a[j]);
    const double djj = readlane_d(a[j], j);
    double y = __builtin_amdgcn_rcp(djj), e = 0;
#define AVM_TAIL(slot)                                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  if (j > 0) {                                                                                                         \
    double sk[3];                                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 3; q++) sk[q] = readlane_d(a[j - 1], min(j + 1 + (slot) + 5 * q, NB - 1));   \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 3; q++)                                                                      \
      if (j + 1 + (slot) + 5 * q < NB) a[j + 1 + (slot) + 5 * q] = fma(-uprev, sk[q], a[j + 1 + (slot) + 5 * q]);      \
  }                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
    AVM_TAIL(0)
    e = fma(-djj, y, 1.0);
    AVM_TAIL(1)
    y = fma(y, e, y);
    AVM_TAIL(2)
    e = fma(-djj, y, 1.0);
    AVM_TAIL(3)
    y = fma(y, e, y);
    AVM_TAIL(4)
#undef AVM_TAIL
    uprev = a[j] * y;
  }
  {
    double* dst = idl ? lds + L_LINV + buf * (16 * TP_PS) + (r & 15) * TP_PS : row;
    double* dump = lds + L_DUMP + r;
    const int kmax = idl ? NB - 1 : (r < nb ? r : -1);
#pragma unroll
    for (int k = 0; k < NB; k++) *(k <= kmax ? dst + k : dump) = a[k];
  }
  // 1 / sqrt(d_c) of the block's columns beside L~^-T (every wavefront's solves scale their rows with it: computed here once, not four times
  // behind the barrier), and the verdict on the pivots
  wave_lds_sync();
  if (r < NB) {
    const double dc = lds[L_PATCH + patch * (16 * TP_PS) + min(r, nb - 1) * (TP_PS + 1)];
    if (!(dc > 0.0)) reinterpret_cast<int*>(lds + L_INT)[I_FAIL] = stamp;  // non-positive (or NaN) pivot in a pivot column of step stamp - 1
    lds[L_LINV + buf * (16 * TP_PS) + r * TP_PS + 16] = fast_rsqrt_pe(dc);
  }
#ifdef AVM_TP
  AVM_PRIO_BULK_CHOL();
#else
  __builtin_amdgcn_s_setprio(0);
#endif
}

// sum over the 16 lanes of a DPP row; the result is valid in lane 15 of every row (row_shr with bound_ctrl: a lane without a source adds 0)
AVM_DEV double tp_row_sum(double v) {
#define AVM_DPP_ADD(ctrl)                                                                      \
  {                                                                                            \
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), ctrl, 0xf, 0xf, true);    \
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), ctrl, 0xf, 0xf, true);    \
    v += __hiloint2double(hi, lo);                                                             \
  }
  AVM_DPP_ADD(0x111)
  AVM_DPP_ADD(0x112)
  AVM_DPP_ADD(0x114)
  AVM_DPP_ADD(0x118)
#undef AVM_DPP_ADD
  return v;
}

// compile-time loops: every tile index below has to be a constant, or the tile array would live in scratch memory
template <class F, int... Is>
AVM_DEV void tp_sfor_impl(F&& f, std::integer_sequence<int, Is...>) {
  (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, class F>
AVM_DEV void tp_sfor(F&& f) {
  tp_sfor_impl(f, std::make_integer_sequence<int, N>{});
}

// Factor the assembled system and solve it: (H' + mu D^2) y = g', y -> lds[L_Y .. L_Y + NF).  Returns false on a non-positive pivot
// (uniform over the workgroup).  Called by every wavefront of the workgroup, WV = the caller's wavefront; the tiles live on wavefronts 0..3
// (tp_owner) - the latency build's wavefronts 4..7 hold none and only take part in the barriers and the count.
// (tile indices and LDS slots as constants of the instantiation: left to the optimizer, one of the four wavefronts' tile arrays ended up in scratch memory)
#ifdef AVM_PROF_CHOL  // (development: where a factorization's time goes, per wavefront; slots 56.. of the profile: chain, wait b, solve, wait d, update, rest)
#define CPROF_T0() long long cp__ = clock64()
#define CPROF(slot) do { if (c.prof && lane == 0 && WV == AVM_PROF_CHOL) { long long n__ = clock64(); c.prof[56 + (slot)] += n__ - cp__; cp__ = n__; } } while (0)
#else
#define CPROF_T0() ((void)0)
#define CPROF(slot) ((void)0)
#endif
#define TPI(k, i) (std::integral_constant<int, tp_idx(WV, k, i)>::value)
#define TPW(k, i) (std::integral_constant<int, tp_wslot(k, i)>::value)
template <int WV>
AVM_NOINL bool chol_regs() {
  double* lds = LDS();
  const int lane = threadIdx.x & 63, lk = lane >> 4, lr = lane & 15;
  int* s_fail = reinterpret_cast<int*>(lds + L_INT) + I_FAIL;
  constexpr int NTL = tp_ntiles(WV) > 0 ? tp_ntiles(WV) : 1;
  d4 T[NTL];
#ifdef AVM_PROF_CHOL
  const long long cp_in__ = clock64();
#endif
  // ---- load, in elimination order (structural zeros included; a tile of the pattern the assembled system cannot reach starts as zero: it is fill).
  // Two passes, each with all its memory operations in flight: the offsets of every tile (left alone the compiler waited for one 8-byte load
  // per tile before the next: 13 trips to the L2 in a row), then the entries.
  typedef unsigned short us4 __attribute__((ext_vector_type(4)));
  us4 off[NTL];
  tp_sfor<TPT>([&](auto I) {
    constexpr int i = I;
    if constexpr (tp_owner(i) == WV) {
      tp_sfor<i + 1>([&](auto K) {
        constexpr int k = K;
        if constexpr (tp_nz(k, i) && TPP.h[k][i])
          off[TPI(k, i)] = *reinterpret_cast<const __attribute__((address_space(1))) us4*>(
              (const __attribute__((address_space(1))) unsigned short*)&tp_offsets.o[tp_h_ord(k, i)][0][0] + 4 * lane);
      });
    }
  });
  __builtin_amdgcn_sched_barrier(0);
  tp_sfor<TPT>([&](auto I) {
    constexpr int i = I;
    if constexpr (tp_owner(i) == WV) {
      tp_sfor<i + 1>([&](auto K) {
        constexpr int k = K;
        if constexpr (tp_nz(k, i)) {
          d4& t = T[TPI(k, i)];
          if constexpr (TPP.h[k][i]) {
#pragma unroll
            for (int r = 0; r < 4; r++) t[r] = lds[off[TPI(k, i)][r]];
          } else {
            t = d4{0, 0, 0, 0};
          }
        }
      });
    }
  });
  typedef __attribute__((address_space(3))) int lds_int_t;
  lds_int_t* s_cnt = reinterpret_cast<lds_int_t*>((uintptr_t)(L_INT * 8 + I_CNT * 4));
  if (threadIdx.x == 0) *s_fail = 0, *s_cnt = 0;
  const WinCtx& c = lds_ctx();
#ifdef AVM_PROF_CHOL
  if (c.prof && lane == 0 && WV == AVM_PROF_CHOL) c.prof[61] += clock64() - cp_in__;  // the load
#endif
  PROF_T0();
  __syncthreads();  // every tile is in registers: the union region becomes the factorization's scratch
  PROF(c, 4);
  CPROF_T0();
  // by the owner of pivot column k: diagonal tile (staged in its step's patch) -> chain -> L~_kk^T back into the tile, L~_kk^-T in buffer tp_buf(k)
  d4 Dlast = {0, 0, 0, 0};  // the last diagonal tile as it was before its chain (its column TP_NBL is the right-hand side)
  auto run_chain = [&](auto K) {
    constexpr int k = K;
    CPROF(4);
    d4& D = T[TPI(k, k)];
    if constexpr (tp_step_of(k) == 0) {  // (the later ones were staged by the step before)
#pragma unroll
      for (int r = 0; r < 4; r++) lds[L_PATCH + tp_slot_of(k) * (16 * TP_PS) + (lk + 4 * r) * TP_PS + lr] = D[r];
    }
    wave_lds_sync();
    tp_diag_chain(k == TPT - 1 ? TP_NBL : 16, tp_slot_of(k), tp_buf(k), tp_step_of(k) + 1);
    wave_lds_sync();
    // the diagonal tile becomes L~_kk^T (entry (a, b) = L~[b][a]); the patch is free for the next step's chain
#pragma unroll
    for (int r = 0; r < 4; r++) D[r] = lds[L_PATCH + tp_slot_of(k) * (16 * TP_PS) + lr * TP_PS + lk + 4 * r];
    CPROF(0);
  };
  tp_sfor<2>([&](auto A) {
    constexpr int k = tp_step_piv(0, A);
    if constexpr (tp_owner(k) == WV) run_chain(std::integral_constant<int, k>{});
  });
  bool failed = false;
  tp_sfor<TP_NSTEP>([&](auto TT) {
    constexpr int t = TT;
    if (failed) return;  // (uniform)
    CPROF(4);
    __syncthreads();  // (b) L~_kk^-T of this step's pivot columns are published; every wavefront is done with step t - 1
    CPROF(1);
    // (a chain stamps a non-positive pivot with its step + 1: the next chains may already run while a slow wavefront reads this, and
    //  all four have to take the same way out)
    {
      const int f = *s_fail;
      if (f != 0 && f <= t + 1) {
        failed = true;
        return;
      }
    }
    // (c) W(k, i) = L_kk^-1 U(k, i) for this wavefront's columns i > k: the final factor tiles, published for the others' updates
    tp_sfor<tp_step_np(t)>([&](auto A) {
      constexpr int k = tp_step_piv(t, A);
      constexpr int nb = k == TPT - 1 ? TP_NBL : 16;
      if constexpr (tp_row_held(WV, k) || (k == TPT - 1 && tp_owner(k) == WV)) {
        // A operand of the solves: L_kk^-1[i' = lr][k' = lk + 4 m] = L~^-T[k'][i'] / sqrt(d_i'); the row scaling is applied to the product
        double aop[4], isq4[4];
        const double* LT = lds + L_LINV + tp_buf(k) * (16 * TP_PS);
#pragma unroll
        for (int m = 0; m < 4; m++) {
          const double v = LT[(lk + 4 * m) * TP_PS + lr];
          aop[m] = (lk + 4 * m < nb && lr < nb) ? v : 0.0;
          isq4[m] = LT[min(lk + 4 * m, nb - 1) * TP_PS + 16];
        }
        if constexpr (k == TPT - 1 && tp_owner(k) == WV) {  // the last diagonal tile gives up the right-hand side: z_10 = L^-1 b
          d4 Za = {0, 0, 0, 0}, Zb = {0, 0, 0, 0};
          Za = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[0], Dlast[0], Za, 0, 0, 0);
          Zb = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[1], Dlast[1], Zb, 0, 0, 0);
          Za = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[2], Dlast[2], Za, 0, 0, 0);
          Zb = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[3], Dlast[3], Zb, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; r++)
            if (lr == TP_NBL && lk + 4 * r < TP_NBL) lds[L_ZV + 16 * k + lk + 4 * r] = (Za[r] + Zb[r]) * isq4[r];
        }
        tp_sfor<TPT - 1 - k>([&](auto II) {
          constexpr int i = k + 1 + II;
          if constexpr (tp_owner(i) == WV && tp_nz(k, i)) {
            d4& U = T[TPI(k, i)];
            d4 Wa = {0, 0, 0, 0}, Wb = {0, 0, 0, 0};
            Wa = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[0], U[0], Wa, 0, 0, 0);
            Wb = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[1], U[1], Wb, 0, 0, 0);
            Wa = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[2], U[2], Wa, 0, 0, 0);
            Wb = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[3], U[3], Wb, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
              U[r] = (Wa[r] + Wb[r]) * isq4[r];
              lds[L_WROW + TPW(k, i) * 256 + r * 64 + lane] = U[r];
            }
            if constexpr (i == TPT - 1) {  // the right-hand side column of tile column 10 is z_k
#pragma unroll
              for (int r = 0; r < 4; r++)
                if (lr == TP_NBL) lds[L_ZV + 16 * k + lk + 4 * r] = U[r];
            }
          }
        });
      }
    });
    if constexpr (t < TP_NSTEP - 1) {
      // the owner of a pivot column q of the next step needs nothing but its own W(k, q) for tile (q, q): it is updated and staged in the
      // patch before the count, while the wavefronts with more tiles in this step's rows still solve; the chain starts right behind it
      tp_sfor<tp_step_np(t + 1)>([&](auto B) {
        constexpr int q = tp_step_piv(t + 1, B);
        if constexpr (tp_owner(q) == WV) {
          d4& U = T[TPI(q, q)];
          tp_sfor<tp_step_np(t)>([&](auto A) {
            constexpr int k = tp_step_piv(t, A);
            if constexpr (tp_nz(k, q)) {
              const d4& Wd = T[TPI(k, q)];
#pragma unroll
              for (int r = 0; r < 4; r++) U = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wd[r], Wd[r], U, 0, 0, 0);
            }
          });
          if constexpr (q == TPT - 1) Dlast = U;
#pragma unroll
          for (int r = 0; r < 4; r++) lds[L_PATCH + tp_slot_of(q) * (16 * TP_PS) + (lk + 4 * r) * TP_PS + lr] = U[r];
        }
      });
      CPROF(2);
      // (d) this step's rows of W are published - counted, not a barrier: the owner of a next pivot column needs nobody's tiles for its chain
      // and does not wait (1 K cycles per step it spent at a barrier for the wavefronts with more tiles to solve); everybody else waits for all four counts
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
      if (lane == 0) __hip_atomic_fetch_add(s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      auto wait_rows = [&]() {
        while (__hip_atomic_load(s_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (NT / 64) * (t + 1)) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      };
      if constexpr (tp_owns_piv(WV, t + 1)) {
        tp_sfor<tp_step_np(t + 1)>([&](auto B) {
          constexpr int q = tp_step_piv(t + 1, B);
          if constexpr (tp_owner(q) == WV) run_chain(std::integral_constant<int, q>{});
        });
      }
      if constexpr (tp_ntiles(WV) > 0) wait_rows();
      CPROF(3);
      // (e) trailing update U(j, i) -= W(k, j)^T W(k, i), k < j <= i, over the tiles of this step's rows that exist (the next step's diagonal
      // tiles have theirs already), while the next chains run on their owners
      tp_sfor<tp_step_np(t)>([&](auto A) {
        constexpr int k = tp_step_piv(t, A);
        tp_sfor<TPT - 1 - k>([&](auto II) {
          constexpr int i = k + 1 + II;
          if constexpr (tp_owner(i) == WV && tp_nz(k, i)) {
            const d4& Wi = T[TPI(k, i)];
            tp_sfor<i - k>([&](auto JJ) {
              constexpr int j = k + 1 + JJ;
              if constexpr (tp_nz(k, j) && !(j == i && tp_is_piv(t + 1, i))) {
                static_assert(tp_nz(j, i), "the pattern is closed under the elimination's fill");
                d4 Wj;
                if constexpr (tp_owner(j) == WV) {
                  Wj = T[TPI(k, j)];
                } else {
#pragma unroll
                  for (int r = 0; r < 4; r++) Wj[r] = lds[L_WROW + TPW(k, j) * 256 + r * 64 + lane];
                }
                d4& U = T[TPI(j, i)];
#pragma unroll
                for (int r = 0; r < 4; r++) U = __builtin_amdgcn_mfma_f64_16x16x4f64(-Wj[r], Wi[r], U, 0, 0, 0);
              }
            });
          }
        });
      });
    }
  });
  if (failed) return false;
  // (every wavefront is past the last step's barrier: nobody reads a row of W any more, and the partial sums of the back substitution live there)
  if constexpr (WV < TP_NWO)
    for (int q = lane; q < TP_NPOS; q += 64) lds[L_PARTV + WV * TP_NPOS + q] = 0.0;
  __syncthreads();  // z is complete in lds[L_ZV]
  PROF(c, 5);
  if (*s_fail) return false;
  // ---- backward substitution L^T x = z by the same steps, last to first; x_i replaces z_i (elimination order) and goes to lds[L_Y] (the system's order)
  d4 E[TPT - 1];  // E[k] += U(k, i) .* x_i over this wavefront's columns i > k (element-wise: reduced once, when block k is due)
#pragma unroll
  for (int k = 0; k < TPT - 1; k++) E[k] = d4{0, 0, 0, 0};
  tp_sfor<TP_NSTEP>([&](auto TR) {
    constexpr int t = TP_NSTEP - 1 - TR;
    tp_sfor<tp_step_np(t)>([&](auto A) {
      constexpr int i = tp_step_piv(t, A);
      constexpr int nb = i == TPT - 1 ? TP_NBL : 16;
      constexpr int PB = L_PATCH + tp_slot_of(i) * (16 * TP_PS);
      if constexpr (tp_owner(i) == WV) {
        // v = z_i - the four partial sums; L~_ii back into the patch in [row][column] form; the 16-step chain of chol_solve_block
        const d4& D = T[TPI(i, i)];
#pragma unroll
        for (int r = 0; r < 4; r++) lds[PB + lr * TP_PS + lk + 4 * r] = D[r];
        const int rr = min(lr, nb - 1);
        double bv = lds[L_ZV + 16 * i + rr];
#pragma unroll
        for (int w = 0; w < TP_NWO; w++) bv -= lds[L_PARTV + w * TP_NPOS + 16 * i + rr];
        wave_lds_sync();
        double colv[16];
#pragma unroll
        for (int q = 0; q < 16; q++) colv[q] = lds[PB + q * TP_PS + rr];
        const double isq = fast_rsqrt_pe(lds[PB + rr * (TP_PS + 1)]), di2 = isq * isq;
        bv *= isq;
#pragma unroll
        for (int q = 0; q < 16; q++) colv[q] *= di2;
        double xout = 0.0;
#ifndef AVM_BS_READLANE  // (round 6, last: 2 v_readlane_b32 + v_fma_f64 per step before - bit-identical, solve 9.29 -> 9.21 ms)
        // x_jj is lane jj's bv; every lane subtracts colv[jj] x_jj - the broadcast as the multiply-add's own DPP operand (v_fmac_f64_dpp row_newbcast: no trip
        // through the scalar registers; the s_nop is the two wait states a DPP read needs behind the VALU write of the same register)
        tp_sfor<nb>([&](auto JR) {
          constexpr int jj = nb - 1 - JR;
          xout = lr == jj ? bv : xout;
          const double nc = -colv[jj];
          asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "+v"(bv) : "v"(nc), "n"(jj));
        });
#else
#pragma unroll
        for (int jj = nb - 1; jj >= 0; jj--) {
          const double xj = readlane_d(bv, jj);
          bv = fma(-colv[jj], xj, bv);
          xout = lane == jj ? xj : xout;
        }
#endif
        if (lane < nb) {
          lds[L_ZV + 16 * i + lane] = xout;
          const int col = tp_perm_dev(16 * i + lane);
          if (col != TP_PAD) lds[L_Y + col] = xout;
        }
        wave_lds_sync();
        // fold x_i into the element-wise accumulators of the blocks above (lane (lk, lr): column lr of every tile)
        const double xl = lr < nb ? lds[L_ZV + 16 * i + min(lr, nb - 1)] : 0.0;
        tp_sfor<i>([&](auto K) {
          constexpr int k = K;
          if constexpr (tp_nz(k, i)) {
            const d4& U = T[TPI(k, i)];
#pragma unroll
            for (int r = 0; r < 4; r++) E[k][r] = fma(U[r], xl, E[k][r]);
          }
        });
      }
    });
    if constexpr (t > 0) {
      // every wavefront that holds a tile of a row the next step solves: its share of that block is complete (all its columns beyond it have been
      // folded in); the others' partial sums stay the zeros they were set to
      tp_sfor<tp_step_np(t - 1)>([&](auto B) {
        constexpr int p = tp_step_piv(t - 1, B);
        if constexpr (tp_row_held(WV, p)) {
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const double sacc = tp_row_sum(E[p][r]);
            if (lr == 15) lds[L_PARTV + WV * TP_NPOS + 16 * p + lk + 4 * r] = sacc;
          }
        }
      });
      __syncthreads();
    }
  });
  __syncthreads();
  PROF(c, 6);
  return true;
}
#undef TPI
#undef TPW
// the factorization's compile-time tables of this build for the tests (tests/test_tp_pattern.py states them in numpy): T = TPT tile columns,
// out[0 .. T T) = TPP.h, [T T .. 2 T T) = TPP.nz (both [k][i]), then tp_owner [T], then tp_perm of the 16 T positions (-1: padding)
int tp_pattern_export(int* out) {
  for (int k = 0; k < TPT; k++)
    for (int i = 0; i < TPT; i++) out[k * TPT + i] = TPP.h[k][i], out[TPT * TPT + k * TPT + i] = TPP.nz[k][i];
  for (int i = 0; i < TPT; i++) out[2 * TPT * TPT + i] = tp_owner(i);
  for (int n = 0; n < 16 * TPT; n++) out[2 * TPT * TPT + TPT + n] = tp_perm(n);
  return 2 * TPT * TPT + TPT + 16 * TPT;
}
// (end of chol_regs)
#ifndef AVM_TP  // the other builds (the latency build: for a prior chol_regs' pattern does not hold): left-looking factorization of the packed system in LDS
// Scratch of the factorization inside the tile at L_WCH (dead while S is being factored): L^-T of the current and of the
// next diagonal block, and a per-lane dump slot for the masked-out stores.
constexpr int L_CLT = L_WCH /* two buffers of 256: block j's L^-T in buffer j & 1 */, L_CDUMP = L_WCH + 512;

// ---- tiles of the factorization, 16x16 on v_mfma_f64_16x16x4 --------------------------------------------------------
// Everything is unconditional (a predicated LDS access compiles to a branch with its own s_waitcnt): operand rows are
// clamped to the last valid row (the duplicates only reach outputs that are not stored), destination loads are clamped to
// a valid address and masked-out stores go to a per-lane dump slot.  The k index of a product is a summation index, so
// lane group lk takes columns 4 lk + {0..3} of a 16-column block: two 16-byte loads per operand instead of four 8-byte ones.
struct CholTile {
  double d[4];
  int o[4];  // destination offsets (doubles from lds[0]); masked-out entries point at the dump slot
};

// Factor the nb x nb diagonal block at c0 in the registers of the calling wavefront (lane = row, register = column).
//  * Select-free: lanes / columns outside the block (and the upper triangle) just carry finite junk that is never stored.
//  * Lanes 16..31 carry the rows of the identity through the same eliminations: lane 16+i ends with row i of L^-T, which the
//    MFMA panel solve multiplies the rows below with (zero extra instructions in the pivot chain).
//  * Square-root free: column j is divided by its pivot with v_rcp_f64 + two Newton steps; rows and L^-T are stored
//    unscaled (times sqrt(d_c) per column c, the pivot d_c itself on the diagonal) and the consumers apply rsqrt(d_c):
//    the panel solve (which also raises the non-positive-pivot flag) and chol_solve_lds.
//  * The wavefront is instruction-issue bound (~4.5 cycles per FP64 / v_readlane instruction, 3 instructions per
//    (pivot, column) pair), so everything else is kept out of it: no pivot bookkeeping, stores by address select, and the
//    rank-1 update of pivot j-1 is software-pipelined by hand into the latency shadows of pivot j's reciprocal chain.
AVM_NOINL void chol_diag_block(int c0, int nb, int buf) {
  constexpr int NB = CNB;
  double* S = LDS() + L_S;
  const int r = threadIdx.x & 63;
  __builtin_amdgcn_s_setprio(3);  // this wavefront is the critical path of the factorization: win issue arbitration
  double a[NB];
  const bool idl = (r & 48) == 16;
  const int rc = min(r, nb - 1);
  double* row = S + roff(c0 + rc) + c0;
  {
#pragma unroll
    for (int k = 0; k < NB; k++) a[k] = row[k];  // 16 reads in flight at immediate offsets; past the diagonal they run into the
                                                 // next packed rows (still inside the factor's LDS region + slack): junk, never stored
#pragma unroll
    for (int k = 0; k < NB; k++) a[k] = idl ? ((r & 15) == k ? 1.0 : 0.0) : a[k];  // lanes 16..31: the identity's rows
  }
  double uprev = 0.0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    if (j > 0) a[j] = fma(-uprev, readlane_d(a[j - 1], j), a[j]);
    const double djj = readlane_d(a[j], j);
    double y = __builtin_amdgcn_rcp(djj), e = 0;
#define AVM_TAIL(slot)                                                                                                 \
  __builtin_amdgcn_sched_barrier(0);                                                                                   \
  if (j > 0) {                                                                                                         \
    double sk[3];                                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 3; q++) sk[q] = readlane_d(a[j - 1], min(j + 1 + (slot) + 5 * q, NB - 1));   \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    _Pragma("unroll") for (int q = 0; q < 3; q++)                                                                      \
      if (j + 1 + (slot) + 5 * q < NB) a[j + 1 + (slot) + 5 * q] = fma(-uprev, sk[q], a[j + 1 + (slot) + 5 * q]);      \
  }                                                                                                                    \
  __builtin_amdgcn_sched_barrier(0);
    AVM_TAIL(0)
    e = fma(-djj, y, 1.0);
    AVM_TAIL(1)
    y = fma(y, e, y);
    AVM_TAIL(2)
    e = fma(-djj, y, 1.0);
    AVM_TAIL(3)
    y = fma(y, e, y);
    AVM_TAIL(4)
#undef AVM_TAIL
    uprev = a[j] * y;
  }
  {
    double* dst = idl ? LDS() + L_CLT + buf * (NB * NB) + (r & 15) * NB : row;
    double* dump = LDS() + L_CDUMP + r;
    const int kmax = idl ? NB - 1 : (r < nb ? r : -1);
#pragma unroll
    for (int k = 0; k < NB; k++) *(k <= kmax ? dst + k : dump) = a[k];
  }
  __builtin_amdgcn_s_setprio(0);
}

// U_ij = A_ij - sum_{p < j} X_ip X_jp^T: the update of tile (ti, tj) by the panels [p_begin, p_end) at once (LEFT-looking),
// accumulated in registers over the solved panels - 4 tj MFMAs on four independent chains - and ONE read-modify-write of the destination
// (right-looking costs a destination round trip per panel, and the LDS write path is the slow one: ~70 B/clk).
AVM_DEV void chol_left_tile(int ti, int tj, int p_begin, int p_end) {
  constexpr int NR = NF + 1;
  double* S = LDS() + L_S;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const dv2* pa = reinterpret_cast<const dv2*>(S + roff(min(16 * ti + lr, NR - 1)) + 4 * lk);
  const dv2* pb = reinterpret_cast<const dv2*>(S + roff(min(16 * tj + lr, NF - 1)) + 4 * lk);
  CholTile T;  // (destination part only)
  {
    const int gj = 16 * tj + lr;
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int gi = 16 * ti + lk + 4 * r;
      const bool ok = gi < NR && gj < NF && gj <= gi;
      const int gic = min(gi, NR - 1);
      const int ol = L_S + roff(gic) + min(gj, min(gic, NF - 1));
      T.d[r] = LDS()[ol];
      T.o[r] = ok ? ol : L_CDUMP + lane;
    }
  }
  d4 D0 = {0, 0, 0, 0}, D1 = {0, 0, 0, 0}, D2 = {0, 0, 0, 0}, D3 = {0, 0, 0, 0};
  const bool diag = ti == tj;
#pragma unroll 1
  for (int p = p_begin; p < p_end; p++) {
    const dv2 a0 = pa[8 * p], a1 = pa[8 * p + 1];
    dv2 b0 = a0, b1 = a1;
    if (!diag) b0 = pb[8 * p], b1 = pb[8 * p + 1];
    D0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], D0, 0, 0, 0);
    D1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], D1, 0, 0, 0);
    D2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], D2, 0, 0, 0);
    D3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], D3, 0, 0, 0);
  }
  const d4 D = (D0 + D1) + (D2 + D3);
#pragma unroll
  for (int r = 0; r < 4; r++) LDS()[T.o[r]] = T.d[r] - D[r];
}

// X_ij = (A_ij - X_{i,j-1} X_{j,j-1}^T) L_jj^-T: the tile of row block ti in block column j (c0 = 16 j, nb columns), for the
// rows >= c0 + nb.  `upd`: the tile still lacks the update of the last solved panel (j - 1); that product is computed
// TRANSPOSED - X_{j,j-1} X_{i,j-1}^T - because the accumulator layout of the transposed tile is exactly the A operand
// layout of the solve: the update costs no round trip through LDS.  bop = L_jj^-T (B operand), isq = rsqrt(d_c) of the
// lane's column (see chol_diag_block).
AVM_DEV void chol_panel_tile(int ti, int c0, int nb, const double (&bop)[CNB / 4], double isq, bool upd) {
  constexpr int NB = CNB, NR = NF + 1;
  double* lds = LDS();
  double* S = lds + L_S;
  const int lane = threadIdx.x & 63, lr = lane & 15, lk = lane >> 4;
  const int c1 = c0 + nb;
  const int row = 16 * ti + lr;
  const double* pa = S + roff(min(row, NR - 1)) + c0 + lk;
  const bool va = row < NR && row >= c1;
  double aop[NB / 4];
#pragma unroll
  for (int m = 0; m < NB / 4; m++) aop[m] = pa[4 * m];  // past-the-row reads stay inside the LDS carve and are masked below
  if (upd) {
    const dv2* pj = reinterpret_cast<const dv2*>(S + roff(min(c0 + lr, NF - 1)) + (c0 - NB) + 4 * lk);
    const dv2* pi = reinterpret_cast<const dv2*>(S + roff(min(row, NR - 1)) + (c0 - NB) + 4 * lk);
    const dv2 a0 = pj[0], a1 = pj[1], b0 = pi[0], b1 = pi[1];
    d4 C0 = {0, 0, 0, 0}, C1 = {0, 0, 0, 0}, C2 = {0, 0, 0, 0}, C3 = {0, 0, 0, 0};
    C0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[0], b0[0], C0, 0, 0, 0);
    C1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[1], b0[1], C1, 0, 0, 0);
    C2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[0], b1[0], C2, 0, 0, 0);
    C3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[1], b1[1], C3, 0, 0, 0);
    const d4 C = (C0 + C1) + (C2 + C3);  // C[m] = (X_{i,j-1} X_{j,j-1}^T)[row lr][column lk + 4 m]
#pragma unroll
    for (int m = 0; m < NB / 4; m++) aop[m] -= C[m];
  }
#pragma unroll
  for (int m = 0; m < NB / 4; m++) aop[m] = (va && lk + 4 * m < nb) ? aop[m] : 0.0;
  d4 Da = {0, 0, 0, 0}, Db = {0, 0, 0, 0};
  Da = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[0], bop[0], Da, 0, 0, 0);
  Db = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[1], bop[1], Db, 0, 0, 0);
  Da = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[2], bop[2], Da, 0, 0, 0);
  Db = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[3], bop[3], Db, 0, 0, 0);
  const d4 D = (Da + Db) * isq;
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int gi = 16 * ti + lk + 4 * r;
    lds[(gi < NR && gi >= c1 && lr < nb) ? L_S + roff(gi) + c0 + lr : L_CDUMP + lane] = D[r];
  }
}

// In-place lower Cholesky of the packed NFxNF matrix in lds[L_S]; returns false on a non-positive pivot.
// The right-hand side rides along as row NF of the packed storage, so the forward substitution L z = b happens as part of
// the factorization (z ends up in that row).  LEFT-looking by 16-column blocks, ONE workgroup barrier per block column,
// everything lagging one panel behind the diagonal.  At the barrier that opens phase j: the panels p < j are solved, the
// diagonal block j is factored (L_jj^-T in buffer j & 1), the tiles of block column j carry the updates of the panels
// p <= j - 2 and the diagonal tile (j + 1, j + 1) those of the panels p <= j - 1.  Phase j:
//   wavefront 0  (the critical path): tile (j + 1, j) = [last panel's update, solve]; diagonal tile (j + 1, j + 1) -= panel
//                j; then its 16-pivot chain (chol_diag_block).  It reads nothing the others write in this phase.
//   the helpers  (wavefronts 1-3, 5-7; wavefront 4 shares wavefront 0's SIMD and FP64 pipe and stays idle): the other
//                tiles (i, j) = [last panel's update, solve]; block column j + 1 receives the panels p < j (solved before
//                the phase began); the diagonal tile (j + 2, j + 2) receives the panels p <= j from the helper that
//                solves tile (j + 2, j).
// Every tile is read-modify-written once for all its early panels and once more, fused with its solve, for the last one.
AVM_NOINL bool cholesky_lds(long long* prof) {
  struct { long long* prof; } c{prof};
  double* lds = LDS();
  double* S = lds + L_S;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, lr = lane & 15, lk = lane >> 4;
  constexpr int NB = CNB;
  constexpr int NHELP = NT / 64 - 2;
  const int hslot = wv < 4 ? wv - 1 : wv - 5 + 3;  // helpers 1 2 3 5 6 7 -> 0..5 (wavefronts 0 and 4: not helpers)
  const bool helper = wv != 0 && wv != 4;
  int* s_fail = reinterpret_cast<int*>(lds + L_INT) + I_FAIL;
  if (t == 0) *s_fail = 0;
  PROF_T0();
  if (wv == 0) chol_diag_block(0, NB, 0);
  __syncthreads();
  PROF(c, 4);
  for (int j = 0, c0 = 0; c0 < NF; j++, c0 += NB) {
    const int nb = min(NB, NF - c0), c1 = c0 + nb;
    const int t0 = c1 >> 4;  // first tile row with rows below the block (the block's own tile row when nb < 16)
    // B operand of the solves = L_jj^-T (left in buffer j & 1 by chol_diag_block, stored times sqrt(d_c) per column: the
    // pivots sit on the diagonal of the block)
    double bop[NB / 4];
    {
      const double* LT = lds + L_CLT + (j & 1) * (NB * NB);
#pragma unroll
      for (int m = 0; m < NB / 4; m++) bop[m] = LT[(lk + 4 * m) * NB + lr];
    }
    const int cc = c0 + min(lr, nb - 1);
    const double dc = S[roff(cc) + cc];
    if (!(dc > 0.0)) *s_fail = 1;  // non-positive (or NaN) pivot: every wavefront sees the same values
    const double isq = fast_rsqrt(dc);  // applied to the product's columns: its latency hides under the loads and MFMAs
#pragma unroll
    for (int m = 0; m < NB / 4; m++) bop[m] = (lk + 4 * m < nb && lr < nb) ? bop[m] : 0.0;
    const bool last = c1 >= NF;
    if (wv == 0) {
      const long long q0 = clock64();
      chol_panel_tile(t0, c0, nb, bop, isq, j > 0 && t0 > j);
      if (!last) {
        wave_lds_sync();
        chol_left_tile(j + 1, j + 1, j, j + 1);  // (the panels before were applied a phase ago by the first helper)
        wave_lds_sync();
        chol_diag_block(c1, min(NB, NF - c1), (j + 1) & 1);
      }
      if (c.prof && t == 0) c.prof[28] += clock64() - q0;
    } else if (helper) {
      const long long q0 = clock64();
      for (int ti = t0 + 1 + hslot; ti <= TLAST; ti += NHELP) chol_panel_tile(ti, c0, nb, bop, isq, j > 0 && ti > j);
      // the diagonal tile (j + 2, j + 2) only needs its own row block's panels: the helper that has just solved tile
      // (j + 2, j) applies all of them, panel j included, so that wavefront 0 adds a single panel next phase
      if (!last && hslot == 0 && j + 2 <= TLAST) {
        wave_lds_sync();
        chol_left_tile(j + 2, j + 2, 0, j + 1);
      }
      if (!last && j > 0) {
        // tiles (j + 2 .. TLAST, j + 1) receive the panels p < j; dealt in the opposite order of the panel tiles above
        const int ntile = TLAST - (j + 1);
        for (int k = NHELP - 1 - hslot; k < ntile; k += NHELP) chol_left_tile(j + 2 + k, j + 1, 0, j);
      }
      if (c.prof && t == 64) c.prof[27] += clock64() - q0;
    }
    __syncthreads();
    PROF(c, 5);
    if (*s_fail) return false;
    if (last) break;
  }
  return true;
}

// Backward substitution L^T x = z with z in the augmented row of lds[L_S] (left there by cholesky_lds), result to
// lds[vec..vec+NF).  13.6K multiply-adds on an 11-block serial chain: all of it runs in wavefront 0 with no workgroup
// barrier (a barrier costs ~250 cycles, two per block were most of the old version's time).  Per 16-column block, last
// to first:  lane r (mod 16) holds column r of the block triangle scaled so that x_r comes straight out of v_readlane
// (x_r = d_r^-1/2 z_r - d_r^-1 sum_i raw[i][r] x_i; the diagonal blocks are stored unscaled, see chol_diag_block) and
// the 16 steps are readlane -> fma; entries at or above the diagonal are finite junk that only reaches values that
// have already been consumed.  The finished x_i stay in SGPRs and are applied to the remaining b[j], j < c0, by all
// 64 lanes (loads issued ahead of the chain).
template <int NBV>
AVM_DEV void chol_solve_block(double* S, double* b, int c0, int lane) {
  const int rr = min(lane & 15, NBV - 1);
  const double* col = S + c0 + rr;  // + roff(row): column c0+rr
  const double dr = col[roff(c0 + rr)];
  double colv[NBV];
#pragma unroll
  for (int i = 0; i < NBV; i++) colv[i] = col[roff(c0 + i)];  // uniform row offset; i < rr reads (finite) entries of the next rows
  double bv = b[c0 + rr];
  // rows of the block for all (<= 160 = 3 x 64) remaining columns: in flight during the chain (clamped addresses; a
  // segment beyond c0 is simply not stored)
  const int jc = max(c0 - 1, 0);
  double v0[3][NBV], acc[3];
#pragma unroll
  for (int sgm = 0; sgm < 3; sgm++) {
    const int j = min(64 * sgm + lane, jc);
    acc[sgm] = b[j];
#pragma unroll
    for (int i = 0; i < NBV; i++) v0[sgm][i] = S[roff(c0 + i) + j];
  }
  const double isq = fast_rsqrt(dr), di2 = isq * isq;
  bv *= isq;
#pragma unroll
  for (int i = 0; i < NBV; i++) colv[i] *= di2;
  double xs[NBV], xout = 0.0;
#pragma unroll
  for (int jj = NBV - 1; jj >= 0; jj--) {
    xs[jj] = readlane_d(bv, jj);
    bv = fma(-colv[jj], xs[jj], bv);
    xout = lane == jj ? xs[jj] : xout;
  }
  if (lane < NBV) b[c0 + lane] = xout;
#pragma unroll
  for (int sgm = 0; sgm < 3; sgm++) {
    if (64 * sgm >= c0) break;  // (uniform)
    double a0 = acc[sgm], a1 = 0.0;
#pragma unroll
    for (int i = 0; i < NBV; i++) {
      if (i & 1) a1 = fma(-v0[sgm][i], xs[i], a1); else a0 = fma(-v0[sgm][i], xs[i], a0);
    }
    if (64 * sgm + lane < c0) b[64 * sgm + lane] = a0 + a1;
  }
  wave_lds_sync();
}

AVM_NOINL void chol_solve_lds(int vec) {
  double* lds = LDS();
  double* S = lds + L_S;
  double* b = lds + vec;
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  constexpr int NB = 16;
  if (wv == 0) {
#pragma unroll
    for (int q = 0; q < 3; q++)
      if (lane + 64 * q < NF) b[lane + 64 * q] = S[roff(NF) + lane + 64 * q];
    wave_lds_sync();
    if (NF % NB) chol_solve_block<(NF % NB) ? (NF % NB) : NB>(S, b, (NF / NB) * NB, lane);
    // (unrolled: every block's row offsets become immediates of its LDS reads)
#pragma unroll
    for (int blk = NF / NB - 1; blk >= 0; blk--) chol_solve_block<NB>(S, b, blk << 4, lane);
  }
  __syncthreads();
}

#endif  // AVM_TP / LDS factorization

// One wavefront's share of the Schur update: the tiles (R, C), R in {R0, R1}, C in {C0, C1}, C <= R, of the 5x5
// grid (-1 = absent).  Every 16-column block of W is loaded once per k-step and feeds all the tiles that use it.
// SM (throughput build): the wavefront also does its share of the three-row strip under the grid (schur_strip4 below: rows 64, 65 and
// the right-hand side on v_mfma_f64_4x4x4) over column blocks whose operands it holds anyway - SM = 1: C0, C1 and R0; SM = 2: R0 and the
// strip's own diagonal block 4 - so that the strip costs two more rows of loads and no block a second time.
AVM_DEV double mfma4(double a, double b, double c) { return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0); }
template <int R0, int R1, int C0, int C1, int SM = 0>
AVM_DEV void schur_macro_tile(const WinCtx&) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const double* scl = lds + L_SC;
  gcdouble* W = c.sc + Scratch::W;  // Wt[c][e]
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  constexpr int NR = R1 >= 0 ? 2 : 1, NC = C1 >= 0 ? 2 : 1;
  static_assert(SM == 0 || (R1 < 0 && C1 >= 0 && R0 != C0), "strip modes: one row block against two column blocks");
  constexpr int NS = SM == 1 ? 3 : (SM == 2 ? 2 : 0);
  constexpr int SB[3] = {SM == 1 ? C0 : R0, SM == 1 ? C1 : 4, SM == 1 ? R0 : -1};  // the strip's column blocks
  const int ai = li & 3;
  double D4[3] = {0, 0, 0};
  constexpr int RB[2] = {R0, R1}, CB[2] = {C0, C1};
  constexpr bool SAME = R0 == C0 && R1 == C1;  // diagonal macro tile: the row blocks are the column blocks
  constexpr int KB = 8;                        // k-steps (of 4 features) per batch
  d4 D[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
  // The k index of the products is a summation index, so features may be dealt to (k-step m, lane group lk) in any
  // order: e = e0 + 8 lk + m gives every lane 8 consecutive features = 64 contiguous bytes per block of Wt.
  // Unconditional loads from clamped rows, masked afterwards (a predicated load is a branch + wait).
  struct Batch {
    double vr[2][KB], vc[2][KB], va[KB];
  };
  auto load = [&](int e0, Batch& q) {
    if (SM) {
      gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)min(64 + ai, NPOSE - 1) * WLE + e0 + 8 * lk);
#pragma unroll
      for (int m2 = 0; m2 < KB / 2; m2++) {
        const dv2 v = src[m2];
        q.va[2 * m2] = v.x, q.va[2 * m2 + 1] = v.y;
      }
    }
#pragma unroll
    for (int a = 0; a < NR; a++) {
      gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)min(16 * RB[a] + li, NPOSE - 1) * WLE + e0 + 8 * lk);
#pragma unroll
      for (int m2 = 0; m2 < KB / 2; m2++) {
        const dv2 v = src[m2];
        q.vr[a][2 * m2] = v.x, q.vr[a][2 * m2 + 1] = v.y;
      }
    }
    if (!SAME) {
#pragma unroll
      for (int b = 0; b < NC; b++) {
        gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)min(16 * CB[b] + li, NPOSE - 1) * WLE + e0 + 8 * lk);
#pragma unroll
        for (int m2 = 0; m2 < KB / 2; m2++) {
          const dv2 v = src[m2];
          q.vc[b][2 * m2] = v.x, q.vc[b][2 * m2 + 1] = v.y;
        }
      }
    }
  };
  auto multiply = [&](int e0, const Batch& q) {
    double fe[KB], xe[KB];
#pragma unroll
    for (int m = 0; m < KB; m++) {
      const int el = min(e0 + 8 * lk + m, MAXE + 1);
      fe[m] = lds[L_ST + el], xe[m] = lds[L_ST + 152 + el];
    }
#pragma unroll
    for (int m = 0; m < KB; m++) {
      const bool on = e0 + 8 * lk + m < c.nf;
      double aop[2], bop[2], wr0 = 0;
#pragma unroll
      for (int a = 0; a < NR; a++) {
        const int col = 16 * RB[a] + li;
        const double w = (on && col < NPOSE) ? q.vr[a][m] : 0.0;
        aop[a] = col == NPOSE ? xe[m] : w * fe[m];  // padded row 66: the right-hand side
        if (SAME) bop[a] = w;
        if (a == 0) wr0 = w;
      }
      if (!SAME) {
#pragma unroll
        for (int b = 0; b < NC; b++) bop[b] = (on && 16 * CB[b] + li < NPOSE) ? q.vc[b][m] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < NR; a++)
#pragma unroll
        for (int b = 0; b < NC; b++)
          if (CB[b] <= RB[a]) D[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[a], bop[b], D[a][b], 0, 0, 0);
      if (SM) {  // the strip: A = rows 64, 65 (scaled by f_e), x_e, zeros in lane li % 4 of every quad; B = the blocks as the tiles take them
        const double w4 = (on && ai < 2) ? q.va[m] : 0.0;
        const double a4 = ai == 2 ? xe[m] : w4 * fe[m];
        if (SM == 1) {
          D4[0] = mfma4(a4, bop[0], D4[0]), D4[1] = mfma4(a4, bop[1], D4[1]), D4[2] = mfma4(a4, wr0, D4[2]);
        } else {
          D4[0] = mfma4(a4, wr0, D4[0]), D4[1] = mfma4(a4, li < 2 ? w4 : 0.0, D4[1]);
        }
      }
    }
  };
  // (requesting batch n + 1 while batch n is multiplied was measured: nothing in the throughput build, slower in the other two - registers)
  for (int e0 = 0; e0 < c.nf; e0 += 4 * KB) {
    Batch q;
    load(e0, q);
    multiply(e0, q);
  }
#pragma unroll
  for (int b = 0; b < NS; b++) {  // lane (lk, li) of a strip: row 64 + lk (lk = 2: the right-hand side, 3: nothing), column 16 SB + li
    const int gi = 64 + lk, gj = 16 * SB[b] + li;
    const bool body = gi < NPOSE && gj <= gi, rhs = gi == NPOSE && gj < NPOSE;
#ifdef AVM_TP
    const int off = body ? L_S + roff(gi) + gj : (rhs ? L_RHS + gj : L_DUMP + lane);
#else
    const int off = body ? L_S + roff(gi) + gj : (rhs ? L_S + roff(NF) + gj : L_DUMP + lane);
#endif
    const double sc = (body ? scl[min(gi, NPOSE - 1)] : 1.0) * scl[min(gj, NPOSE - 1)];
    lds[off] = lds[off] - sc * D4[b];
  }
#pragma unroll
  for (int a = 0; a < NR; a++)
#pragma unroll
    for (int b = 0; b < NC; b++) {
      if (CB[b] > RB[a]) continue;
      // branch-free: destination (or this lane's dump slot in the scratch tile), all reads, then all writes
      int off[4];
      double sc[4], cur[4];
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = 16 * RB[a] + lk + 4 * r, gj = 16 * CB[b] + li;
        const bool body = gi < NPOSE && gj <= gi, rhs = gi == NPOSE && gj < NPOSE;
#ifdef AVM_TP
        off[r] = body ? L_S + roff(gi) + gj : (rhs ? L_RHS + gj : L_DUMP + (threadIdx.x & 63));
#else
        off[r] = body ? L_S + roff(gi) + gj : (rhs ? L_S + roff(NF) + gj : L_DUMP + (threadIdx.x & 63));
#endif
        sc[r] = (body ? scl[min(gi, NPOSE - 1)] : 1.0) * scl[min(gj, NPOSE - 1)];
        cur[r] = lds[off[r]];
      }
#pragma unroll
      for (int r = 0; r < 4; r++) lds[off[r]] = cur[r] - sc[r] * D[a][b][r];
    }
}

#ifndef AVM_X
// Tile row 4 of the 5 x 5 grid holds three rows: pose columns 64, 65 and the right-hand side.  As 16 x 16 tiles that is a third of the
// update's matrix instructions for 3 / 80 of its rows; v_mfma_f64_4x4x4 - four independent 4 x 4 x 4 products per instruction, a
// quarter of the FP64 pipe time (scripts/ubench/pair.hip: 18 cycles against 64) - does the same strip with the SAME B operand a
// 16 x 16 tile takes (lane 16 k + c holds W[column c][feature k]: block b = c / 4 is the quad column, fsel.hip's layout note) when all
// four blocks get the three rows (+ one of zeros) as their A: lane 16 k + c holds row c % 4.  D[i][c] comes out at lane 16 i + c.
// The strip over the column blocks C0, C1, C2 (-1 = absent), round 5.
template <int C0, int C1, int C2>
AVM_DEV void schur_strip4(const WinCtx&) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const double* scl = lds + L_SC;
  gcdouble* W = c.sc + Scratch::W;  // Wt[c][e]
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4, ai = li & 3;
  constexpr int NC = C2 >= 0 ? 3 : (C1 >= 0 ? 2 : 1);
  constexpr int CB[3] = {C0, C1, C2};
  constexpr int KB = 8;
  static_assert(NPOSE == 66, "rows 64, 65 | right-hand side | zeros");
  double D[3] = {0, 0, 0};
  for (int e0 = 0; e0 < c.nf; e0 += 4 * KB) {
    double va[KB], vc[3][KB], fe[KB], xe[KB];
    {
      gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)min(64 + ai, NPOSE - 1) * WLE + e0 + 8 * lk);
#pragma unroll
      for (int m2 = 0; m2 < KB / 2; m2++) {
        const dv2 v = src[m2];
        va[2 * m2] = v.x, va[2 * m2 + 1] = v.y;
      }
    }
#pragma unroll
    for (int b = 0; b < NC; b++) {
      if (CB[b] == 4) continue;  // (the diagonal block's columns 64, 65 are the A rows of the lanes li < 2)
      gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)(16 * CB[b] + li) * WLE + e0 + 8 * lk);
#pragma unroll
      for (int m2 = 0; m2 < KB / 2; m2++) {
        const dv2 v = src[m2];
        vc[b][2 * m2] = v.x, vc[b][2 * m2 + 1] = v.y;
      }
    }
#pragma unroll
    for (int m = 0; m < KB; m++) {
      const int el = min(e0 + 8 * lk + m, MAXE + 1);
      fe[m] = lds[L_ST + el], xe[m] = lds[L_ST + 152 + el];
    }
#pragma unroll
    for (int m = 0; m < KB; m++) {
      const bool on = e0 + 8 * lk + m < c.nf;
      const double w = (on && ai < 2) ? va[m] : 0.0;
      const double aop = ai == 2 ? xe[m] : w * fe[m];  // (x_e = 0 beyond the window's features: schur_reduce)
#pragma unroll
      for (int b = 0; b < NC; b++) {
        const double bop = CB[b] == 4 ? (li < 2 ? w : 0.0) : (on ? vc[b][m] : 0.0);
        D[b] = mfma4(aop, bop, D[b]);
      }
    }
  }
  // lane (lk, li): row 64 + lk (lk = 2: the right-hand side, 3: nothing), column 16 C + li
#pragma unroll
  for (int b = 0; b < NC; b++) {
    const int gi = 64 + lk, gj = 16 * CB[b] + li;
    const bool body = gi < NPOSE && gj <= gi, rhs = gi == NPOSE && gj < NPOSE;
#ifdef AVM_TP
    const int off = body ? L_S + roff(gi) + gj : (rhs ? L_RHS + gj : L_DUMP + lane);
#else
    const int off = body ? L_S + roff(gi) + gj : (rhs ? L_S + roff(NF) + gj : L_DUMP + lane);
#endif
    const double sc = (body ? scl[min(gi, NPOSE - 1)] : 1.0) * scl[min(gj, NPOSE - 1)];
    lds[off] = lds[off] - sc * D[b];
  }
}
#endif

// Schur complement on the inverse depths, then the right-hand side into the augmented row:
//   S_pp -= W'^T (hee' + mu D_e^2)^-1 W' ,  rhs = g'_f - W'^T (hee' + mu D_e^2)^-1 g'_e      (' = Jacobi-scaled)
// W = E^T F stays UNSCALED in the scratch slot (W'[e][c] = s_e s_c W[e][c]); the scaling is folded in here:
//   W'^T d' W' = s_i s_j sum_e W[e][i] (s_e^2 d'_e) W[e][j]
// 16x16 tiles on the matrix cores over the 66 (padded 80) pose columns, K = features.  The 15 lower tiles of the
// 5x5 grid are grouped into 6 macro tiles, one per wavefront, so a block of W is fetched once for up to four
// products; operands come straight from the L2-resident slot, a batch of 8 k-steps in flight at a time - no LDS
// staging, no barriers.  Row 66 of the padded grid carries x_e = s_e d'_e g'_e in place of a W column, so tile
// row 4 also delivers the right-hand-side update.
AVM_NOINL void schur_reduce(const WinCtx&, double mu) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const int t = threadIdx.x;
  const double* scl = lds + L_SC;
#ifdef AVM_TP
  if (t < NF) lds[s_off(t, t)] += mu * lds[L_DD + t] * lds[L_DD + t];
  for (int i = t; i < NF; i += NT) lds[L_RHS + i] = lds[L_G + i];  // the right-hand side (column NF of the register tiles; L_Y is free until the solve)
  if (t == NT - 1) lds[L_ZERO] = 0.0, lds[L_ONE] = 1.0;             // what chol_regs' tile load reads for structural zeros / the padding's diagonal
#else
  if (t < NF) lds[L_S + roff(t) + t] += mu * lds[L_DD + t] * lds[L_DD + t];
  for (int i = t; i < NF; i += NT) lds[L_S + roff(NF) + i] = lds[L_G + i];  // RHS rides along as row NF
  if (t == NT - 1) lds[L_ZERO] = 0.0, lds[L_ONE] = 1.0;  // (chol_regs' tile load, as above)
#endif
  // per feature: f_e = s_e^2 / (hee' + mu D_e^2) and x_e = s_e g'_e / (hee' + mu D_e^2)   (L_ST is dead here)
  if (t < MAXE + 2) {
    double f = 0, x = 0;
    if (t < c.nf) {
      const double d = 1.0 / (lds[L_HEE + t] + mu * lds[L_DD + NF + t] * lds[L_DD + NF + t]);
      const double se = scl[NF + t];
      f = se * se * d, x = se * d * lds[L_G + NF + t];
    }
    lds[L_ST + t] = f, lds[L_ST + 152 + t] = x;
  }
  __syncthreads();
#ifdef AVM_TP
  AVM_PRIO_BULK_SCHUR();
  switch (t >> 6) {  // four wavefronts, one per SIMD: the 10 lower tiles of the 4 x 4 grid 2 | 3 | 3 | 2, the three-row strip below them with the pairs
    case 0: schur_macro_tile<2, -1, 0, 1, 1>(c); break;  // + the strip over blocks 0, 1, 2
    case 1: schur_macro_tile<0, 1, 0, 1>(c); break;
    case 2: schur_macro_tile<2, 3, 2, 3>(c); break;
    default: schur_macro_tile<3, -1, 0, 1, 2>(c); break;  // + the strip over blocks 3, 4
  }
  AVM_PRIO_LIGHT();
  __syncthreads();
  return;
#endif
#ifdef AVM_X
  switch (t >> 6) {
    case 0: schur_macro_tile<2, 3, 0, 1>(c); break;  // 4 tiles
    case 1: schur_macro_tile<0, 1, 0, 1>(c); break;  // 3 tiles
    case 2: schur_macro_tile<2, 3, 2, 3>(c); break;  // 3 tiles
    case 3: schur_macro_tile<4, -1, 0, 1>(c); break;
    case 7: schur_macro_tile<4, -1, 2, 3>(c); break;  // (wavefronts w and w + 4 share a SIMD: 4 | 3 + 1 | 3 | 2 + 2 tiles per SIMD)
    case 5: schur_macro_tile<4, -1, 4, -1>(c); break;
    default: break;
  }
#else
  // latency build: the 10 lower tiles of the 4 x 4 grid + the three-row strip (schur_strip4) on eight wavefronts, at most three tiles' worth
  // per SIMD (wavefronts w and w + 4 share one): 2 + strip | 2 + 1 | 2 + 1 | 2 + strip
  switch (t >> 6) {
    case 0: schur_macro_tile<1, -1, 0, 1>(c); break;
    case 4: schur_strip4<0, 1, -1>(c); break;
    case 1: schur_macro_tile<2, -1, 0, 1>(c); break;
    case 5: schur_macro_tile<0, -1, 0, -1>(c); break;
    case 2: schur_macro_tile<3, -1, 0, 1>(c); break;
    case 6: schur_macro_tile<2, -1, 2, -1>(c); break;
    case 3: schur_macro_tile<3, -1, 2, 3>(c); break;
    default: schur_strip4<2, 3, 4>(c); break;
  }
#endif
  __syncthreads();
}

// back substitution y_e = (g'_e - W'_e y_p) / (hee' + mu D_e^2) with W'[e][c] = s_e s_c W[e][c] (W unscaled in the
// slot): 4 lanes per feature, every lane's loads in flight at once; returns 1 if y is not finite
AVM_NOINL double back_substitute(const WinCtx&, double mu) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const int t = threadIdx.x;
  gcdouble* W = c.sc + Scratch::W;
  const double* scl = lds + L_SC;
  double* ys = lds + L_WCH;  // s_c y_c
  constexpr int NQ4 = (NPOSE + 3) / 4;  // W rows dealt to the 4 lanes of a feature, a quarter each: 17 (20)
  if (t < NPOSE) ys[t] = scl[t] * lds[L_Y + t];
  if (t >= NPOSE && t < 4 * NQ4 + 4) ys[t] = 0.0;
  __syncthreads();
  AVM_PRIO_BULK();  // (150 independent dot products from the slot: bulk work; measured 12.54 -> 12.46 ms against leaving it at the light level)
  const int part = t & 3;
#pragma unroll
  for (int pass = 0; pass < (MAXE + NT / 4 - 1) / (NT / 4); pass++) {
    const int e = (t >> 2) + (NT / 4) * pass;
    double sacc = 0;
    if (e < c.nf) {
      gcdouble* We = W + e;  // Wt[c][e]
      double v[NQ4];
#pragma unroll
      for (int j = 0; j < NQ4; j++) v[j] = (part + 4 * j < NPOSE) ? We[(size_t)(part + 4 * j) * WLE] : 0.0;
#pragma unroll
      for (int j = 0; j < NQ4; j++) sacc += v[j] * ys[part + 4 * j];
    }
    sacc += __shfl_xor(sacc, 1, 64);
    sacc += __shfl_xor(sacc, 2, 64);
    if (e < c.nf && part == 0) {
      const double he = lds[L_HEE + e] + mu * lds[L_DD + NF + e] * lds[L_DD + NF + e];
      lds[L_Y + NF + e] = (lds[L_G + NF + e] - scl[NF + e] * sacc) / he;
    }
  }
  AVM_PRIO_LIGHT();
  __syncthreads();
  double bad = 0;
  for (int i = t; i < NF + c.nf; i += NT)
    if (!isfinite(lds[L_Y + i])) bad = 1;
  return block_max1(bad);
}

// Jacobi column scaling of the assembled system: H' = S H S, hee', g'  (W stays unscaled: see schur_reduce)
// `matrix`: also the entries of S.  In the base build that is needed once per solve, after the evaluation that fixes the
// scaling: from then on the evaluations write S scaled (frame_task, the (a,a) sums, imu_factor_mfma) and the packed prior
// in the slot is scaled in place here, once.
AVM_NOINL void scale_system(const WinCtx&, bool matrix) {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  const int t = threadIdx.x;
  const double* scl = lds + L_SC;
  if (matrix && c.pn > 0) {
    const int* pidx = reinterpret_cast<const int*>(lds + L_INT) + I_PIDX;
    gdouble* HPk = c.sc + Scratch::HP;
    const int npk = c.pn * (c.pn + 1) / 2;
    for (int idx = t; idx < npk; idx += NT) {
      int gi = (int)((__builtin_sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
      while (gi * (gi + 1) / 2 > idx) gi--;
      while ((gi + 1) * (gi + 2) / 2 <= idx) gi++;
      const int gj = idx - gi * (gi + 1) / 2;
      const int ip = pidx[gi], iq = pidx[gj];
      if (ip >= 0 && iq >= 0) HPk[idx] *= scl[ip] * scl[iq];
    }
  }
#ifdef AVM_TP
  if (matrix) {
    // once per solve: the pose rows of the packed triangle entry by entry, then the speed-bias rows in their structural form
    for (int idx = t; idx < NPOSE * NPOSE; idx += NT) {
      const int r = idx / NPOSE, cc = idx - r * NPOSE;
      if (cc <= r) lds[L_S + roff(r) + cc] *= scl[r] * scl[cc];
    }
    for (int idx = t; idx < 99 * SBW; idx += NT) {
      const int q = idx / SBW, p = idx - q * SBW, i = q / 9;
      const int cc = p < 18 ? 6 * (i - 1) + p : NPOSE + 9 * (i - 1) + (p - 18);
      if (cc >= 0 && (p >= 18 || cc < NPOSE)) lds[L_SBC + idx] *= scl[NPOSE + q] * scl[cc];
    }
    const int psb = reinterpret_cast<const int*>(lds + L_INT)[I_PSB];
    for (int idx = t; idx < 9 * NPOSE; idx += NT) {
      const int a = idx / NPOSE, cc = idx - a * NPOSE;
      lds[L_STRIP + idx] *= scl[NPOSE + 9 * psb + a] * scl[cc];
    }
  }
#else
  if (matrix)
  // 16x16 tiles of the packed lower triangle dealt to the wavefronts, 4 entries per lane and tile (the same lane <-> entry
  // map as the accumulators of the factorization): every lane has the same amount of work; three tiles per round with
  // all their loads in flight before the first store, the tile index arithmetic on the scalar unit, and entries outside
  // the matrix go to the lane's dump slot instead of a predicated store
  {
    const int lane = t & 63, lr = lane & 15, lk = lane >> 4;
    const int wvu = __builtin_amdgcn_readfirstlane(t >> 6);
    constexpr int NTR = (NF + 15) / 16, NTILE = NTR * (NTR + 1) / 2, NW = NT / 64, UN = 3;
#pragma unroll 1
    for (int base = wvu; base < NTILE; base += UN * NW) {
      int off[UN][4];
      double v[UN][4], f[UN][4];
#pragma unroll
      for (int u = 0; u < UN; u++) {
        const int tile = base + u * NW;
        const bool tv = tile < NTILE;
        const int tl = min(tile, NTILE - 1);
        int ti = 0;
        while ((ti + 1) * (ti + 2) / 2 <= tl) ti++;
        const int tj = tl - ti * (ti + 1) / 2;
        const int gj = 16 * tj + lr;
        const double sj = scl[min(gj, NF - 1)];
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int gi = 16 * ti + lk + 4 * r;
          const bool ok = tv && gi < NF && gj <= gi;
          off[u][r] = ok ? L_S + roff(gi) + gj : L_DUMP + lane;
          f[u][r] = scl[min(gi, NF - 1)] * sj;
          v[u][r] = lds[off[u][r]];
        }
      }
#pragma unroll
      for (int u = 0; u < UN; u++)
#pragma unroll
        for (int r = 0; r < 4; r++) lds[off[u][r]] = v[u][r] * f[u][r];
    }
  }
#endif
  if (t < c.nf) lds[L_HEE + t] *= scl[NF + t] * scl[NF + t];
  for (int i = t; i < NF + c.nf; i += NT) lds[L_G + i] *= scl[i];
  __syncthreads();
}

// Evaluator::Plus : xc = x (+) (step * scale)
AVM_DEV void state_plus() {
  double* lds = LDS();
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
#ifdef AVM_TP
  for (int e = t; e < MAXE; e += NT) xc[XLAM + e] = x[XLAM + e] + st[NF + e] * scl[NF + e];
#else
  if (t >= 192 && t < 192 + MAXE) {
    const int e = t - 192;
    xc[XLAM + e] = x[XLAM + e] + st[NF + e] * scl[NF + e];
  }
#endif
#ifdef AVM_X
  // relo_Pose (frame 11) / ex_pose: PoseLocalParameterization::Plus when they are variables, else carried over untouched
  const WinCtx& c = lds_ctx();
  if (t >= 384 && t < 386) {
    const bool ex = t == 385;
    const int xo = ex ? XEX : 7 * NFR, o = ex ? XC_EX : 6 * NFR;
    if (ex ? c.est_ex != 0 : c.relo_n > 0) {
      for (int k = 0; k < 3; k++) xc[xo + k] = x[xo + k] + st[o + k] * scl[o + k];
      quat q{x[xo + 6], x[xo + 3], x[xo + 4], x[xo + 5]};
      quat r = qnormalized(qmul(q, deltaQ(mk3(st[o + 3] * scl[o + 3], st[o + 4] * scl[o + 4], st[o + 5] * scl[o + 5]))));
      xc[xo + 3] = r.x, xc[xo + 4] = r.y, xc[xo + 5] = r.z, xc[xo + 6] = r.w;
    } else {
      for (int k = 0; k < 7; k++) xc[xo + k] = x[xo + k];
    }
  }
  if (t == 386) xc[XTD] = c.est_td ? x[XTD] + st[XC_TD] * scl[XC_TD] : x[XTD];
#endif
}

}  // namespace

#ifdef AVM_X
#define AVM_SOLVE_KERNEL window_solve_x_kernel
#define AVM_SOLVE_OCC
#elif defined(AVM_TP)
#define AVM_SOLVE_KERNEL window_solve_tp_kernel
#define AVM_SOLVE_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))  // two four-wavefront workgroups per CU: 256 registers each
#else
#define AVM_SOLVE_KERNEL window_solve_kernel
#define AVM_SOLVE_OCC
#endif
// rot_diff and origin_P0 of double2vector (estimator.cpp:521-546): the yaw of frame 0 before and after the solve -> lds[gf .. gf + 9) = rot_diff,
// [gf + 9 .. gf + 12) = origin_P0 (last_P0 under failure_occur, estimator.cpp:526-531: pl = last_pose0), [gf + 12 .. gf + 15) = the solved P[0].
// Called by every thread before a workgroup barrier.  A function of its own since round 6: inlined, its nine atan2 and twelve sin / cos - full
// library functions on ONE lane - sat in the kernel body with 114 scratch instructions of their own; the three Euler-angle conversions are
// independent and run on three lanes side by side, each with one sincos of its yaw.
AVM_NOINL void gauge_rot_diff(const double* p0, const double* pl, int gf) {
  double* lds = LDS();
  const int t = threadIdx.x;
  if (t >= 64) return;
  double* ang = lds + gf + 16;  // [3][3] yaw pitch roll (degrees) of: frame 0 before | the anchor (last_R0, or frame 0 before) | frame 0 after
  if (t < 3) {
    const double* src = t == 2 ? lds + L_X : (t == 1 && pl ? pl : p0);
    double R[9];
    q2R(quat{src[6], src[3], src[4], src[5]}, R);
    const double y = atan2(R[3], R[0]), sy = sin(y), cy = cos(y);
    const double p = atan2(-R[6], R[0] * cy + R[3] * sy);
    const double r = atan2(R[2] * sy - R[5] * cy, -R[1] * sy + R[4] * cy);
    ang[3 * t] = y / M_PI * 180.0, ang[3 * t + 1] = p / M_PI * 180.0, ang[3 * t + 2] = r / M_PI * 180.0;
  }
  wave_lds_sync();
  if (t == 0) {
    const double* a0 = ang + 3;  // (the anchor: what the reference's origin_R0 holds)
    const double* a1 = ang + 6;
    const double yd = (a0[0] - a1[0]) / 180.0 * M_PI;
    double rd[9] = {cos(yd), -sin(yd), 0, sin(yd), cos(yd), 0, 0, 0, 1};
    if (fabs(fabs(a0[1]) - 90) < 1.0 || fabs(fabs(a1[1]) - 90) < 1.0) {
      // (pitch-singular branch: rot_diff = Rs[0] * R(para_Pose[0])^T with Rs[0] itself - failure_occur does not change it)
      double Rs0[9], R00[9];
      q2R(quat{p0[6], p0[3], p0[4], p0[5]}, Rs0);
      q2R(quat{lds[L_X + 6], lds[L_X + 3], lds[L_X + 4], lds[L_X + 5]}, R00);
      for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) rd[a * 3 + b] = Rs0[a * 3] * R00[b * 3] + Rs0[a * 3 + 1] * R00[b * 3 + 1] + Rs0[a * 3 + 2] * R00[b * 3 + 2];
    }
    const double* P0 = pl ? pl : p0;
    for (int k = 0; k < 9; k++) lds[gf + k] = rd[k];
    for (int k = 0; k < 3; k++) lds[gf + 9 + k] = P0[k], lds[gf + 12 + k] = lds[L_X + k];
  }
}

__global__ __launch_bounds__(NT) AVM_SOLVE_OCC void AVM_SOLVE_KERNEL(SolveArgs A) {
  lds_base_check();
  red_init();
  AVM_PRIO_LIGHT();
  double* lds = LDS();
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int t = threadIdx.x;
  const avm_options& o = lds_opt();
  const avm_window_batch& B = A.b;

  for (int w = blockIdx.x; w < B.n_windows; w += gridDim.x) {
    WinCtx cl;
    cl.sc = as_global(A.scratch + (size_t)blockIdx.x * Scratch::TOTAL);
    cl.osf = as_global(A.iscratch + (size_t)blockIdx.x * ISCRATCH);
    cl.cov = cl.osf + MAXOBS;
    cl.w = w;
    cl.prof = A.prof ? as_global(A.prof + (size_t)blockIdx.x * PROF_SLOTS) : nullptr;
    cl.nf = B.n_feat[w];
    cl.obs = as_global(B.obs_xy + (size_t)w * B.max_obs * 2);
    cl.pdelta = as_global(A.pre_delta + (size_t)w * 100), cl.pjac = as_global(A.pre_jac + (size_t)w * 2250), cl.psqrt = as_global(A.pre_sqrt + (size_t)w * 2250);
    cl.psum = as_global(A.pre_sum_dt + (size_t)w * 10);
    cl.lba = as_global(B.imu_lin_ba + (size_t)w * 30), cl.lbg = as_global(B.imu_lin_bg + (size_t)w * 30);
    cl.pn = B.prior_n ? B.prior_n[w] : 0;
    cl.pnblk = cl.pn > 0 ? B.prior_nblk[w] : 0;
    cl.ldp = B.max_prior;
    cl.pJ = as_global(B.prior_J + (size_t)w * B.max_prior * B.max_prior);
    cl.pr = as_global(B.prior_r + (size_t)w * B.max_prior);
    cl.px0 = as_global(B.prior_x0 + (size_t)w * B.max_pblk * 9);
    {
      int tot = 0;
      if (cl.nf > 0) tot = B.feat_obs_begin[(size_t)w * B.max_feat + cl.nf - 1] + B.feat_nobs[(size_t)w * B.max_feat + cl.nf - 1];
      cl.nobs_tot = tot;
    }
#ifdef AVM_X
    cl.est_ex = A.opt.estimate_extrinsic != 0, cl.est_td = A.opt.estimate_td != 0;
    cl.aux = (cl.est_td && B.obs_vel_td) ? as_global(B.obs_vel_td + (size_t)w * B.max_obs * 4) : nullptr;
    if (!cl.aux) cl.est_td = 0;  // (the host refuses estimate_td without the per-observation data)
    cl.has_relo = B.relo_n && B.relo_feat && B.relo_xy && B.relo_pose;
    cl.relo_n = cl.has_relo ? min(max(B.relo_n[w], 0), cl.nf) : 0;
    cl.relo_xy = cl.relo_n > 0 ? as_global(B.relo_xy + (size_t)w * B.max_feat * 2) : nullptr;
#endif
    __syncthreads();  // the previous window's readers of the LDS context are done
    lds_store_ctx(cl, A.opt);
    const WinCtx& c = lds_ctx();
    __syncthreads();
    PROF_T0();
    long long pq__ = 0;
    (void)pq__;
    PROFQ_T0();
    const long long pw__ = clock64();
    const long long wall0 = A.time_cap_ticks > 0 ? wall_clock64() : 0;  // (only thread 0's copy is ever compared)
    // ---------------- load ----------------
    for (int i = t; i < 77; i += NT) lds[L_X + i] = B.pose[(size_t)w * 77 + i];
    for (int i = t; i < 99; i += NT) lds[L_X + XSB + i] = B.speedbias[(size_t)w * 99 + i];
    for (int i = t; i < MAXE; i += NT) lds[L_X + XLAM + i] = i < c.nf ? B.inv_depth[(size_t)w * B.max_feat + i] : 1.0;
#ifdef AVM_X
    for (int i = t; i < VEC; i += NT) lds[L_SC + i] = 1.0, lds[L_ST + i] = 0.0, lds[L_Y + i] = 0.0, lds[L_DD + i] = 1.0;
    if (t < 7) {
      lds[L_X + XEX + t] = B.ex_pose[(size_t)w * 7 + t];
      // relo_Pose is frame 11 of the state; without a relocalization frame it mirrors pose 0 (never read by a factor)
      lds[L_X + 7 * NFR + t] = c.has_relo ? B.relo_pose[(size_t)w * 7 + t] : B.pose[(size_t)w * 77 + t];
    }
    if (t == 7) lds[L_X + XTD] = (c.est_td && B.td) ? B.td[w] : 0.0;
    if (t == 8) lds[L_X + XTD + 1] = 0.0;
#elif defined(AVM_TP)
    for (int i = t; i < VEC; i += NT) lds[L_SC + i] = 1.0, lds[L_ST + i] = 0.0, lds[L_Y + i] = 0.0, lds[L_DD + i] = 1.0;
#else
    for (int i = t; i < VEC; i += NT) lds[L_SC + i] = 1.0, lds[L_ST + i] = 0.0, lds[L_Y + i] = 0.0, lds[L_DG + i] = 0.0, lds[L_DD + i] = 1.0;
#endif
    for (int i = t; i < MAXPRIOR; i += NT) lds[L_DXP + i] = 0.0, lds[L_RP + i] = 0.0;
    if (t < c.nf) {
      ids[I_FSTART + t] = B.feat_start[(size_t)w * B.max_feat + t];
      ids[I_FNOBS + t] = B.feat_nobs[(size_t)w * B.max_feat + t];
      ids[I_FOBS + t] = B.feat_obs_begin[(size_t)w * B.max_feat + t];
    }
#ifdef AVM_TP
    constexpr int PBT0 = 160;
#else
    constexpr int PBT0 = 256;
#endif
    if (t >= PBT0 && t < PBT0 + c.pnblk) {  // the prior's block table (one round trip instead of one per block)
      const int k = t - PBT0;
      ids[I_PBLK + k * 3] = B.prior_blk_kind[(size_t)w * B.max_pblk + k], ids[I_PBLK + k * 3 + 1] = B.prior_blk_frame[(size_t)w * B.max_pblk + k];
    }
#ifndef AVM_X
    if (t == 0) {
      const double* ex = B.ex_pose + (size_t)w * 7;
      double R[9];
      q2R(quat{ex[6], ex[3], ex[4], ex[5]}, R);
      for (int k = 0; k < 9; k++) lds[L_RIC + k] = R[k];
      for (int k = 0; k < 3; k++) lds[L_RIC + 9 + k] = ex[k];
    }
#endif
    __syncthreads();
    PROFQ(c, 38);
#ifndef AVM_X
    if (t < 7) lds[L_RIC + 12 + t] = B.ex_pose[(size_t)w * 7 + t];  // current ex_pose for the prior's dx
    if (t == 7) lds[L_RIC + 19] = B.td ? B.td[w] : 0.0;             // ... and para_Td (a constant here)
#endif
    if (t < c.nf) {
      const int s0 = ids[I_FOBS + t], no = ids[I_FNOBS + t];
      for (int k = 0; k < no; k++) c.osf[s0 + k] = t;
    }
    {
      // fs[a] = first feature with start >= a, and per frame the features observed in it (as imu_j) in feature order:
      // one wavefront per list, features along the lanes, positions from a ballot's prefix population count
      const int ln = t & 63;
      for (int q = t >> 6; q <= NFR; q += NT / 64) {
        int cnt = 0;
        for (int e0 = 0; e0 < c.nf; e0 += 64) cnt += __popcll(__ballot(e0 + ln < c.nf && ids[I_FSTART + min(e0 + ln, MAXE - 1)] < q));
        if (ln == 0) ids[I_FS + q] = cnt;
      }
    }
    if (t == 0) {
      int off = 0;
#ifdef AVM_TP
      int psb_ = 0;
#else
      int nsb_ = 0, sbfr_ = 0;
#endif
      for (int k = 0; k < c.pnblk; k++) {
        const int kind = ids[I_PBLK + k * 3], fr = ids[I_PBLK + k * 3 + 1];  // (loaded by 16 lanes at once above)
        ids[I_PBLK + k * 3 + 2] = off;
        const int n = kind == AVM_BLK_SPEEDBIAS ? 9 : (kind == AVM_BLK_TD ? 1 : 6);
#ifdef AVM_X
        for (int q = 0; q < n; q++)
          ids[I_PIDX + off + q] = kind == AVM_BLK_POSE ? fr * 6 + q
                                  : (kind == AVM_BLK_SPEEDBIAS ? SB0 + fr * 9 + q
                                     : (kind == AVM_BLK_TD ? (c.est_td ? XC_TD : -1) : (c.est_ex ? XC_EX + q : -1)));
#else
        for (int q = 0; q < n; q++) ids[I_PIDX + off + q] = kind == AVM_BLK_POSE ? fr * 6 + q : (kind == AVM_BLK_SPEEDBIAS ? SB0 + fr * 9 + q : -1);
#endif
#ifdef AVM_TP
        if (kind == AVM_BLK_SPEEDBIAS) psb_ = fr;  // (at most one such block: the host checks it before it chooses this kernel)
#else
        if (kind == AVM_BLK_SPEEDBIAS) nsb_++, sbfr_ |= fr;
#endif
        off += n;
      }
#ifdef AVM_TP
      ids[I_PSB] = psb_;
#else
      // the rule of window_prior_tp_misfit (kernels.hpp): chol_regs' elimination order takes a prior whose only speed-bias block is frame 0's
#ifdef AVM_NO_CR
      ids[I_CRFIT] = 0;
#else
      ids[I_CRFIT] = (nsb_ <= 1 && sbfr_ == 0) ? 1 : 0;
#endif
#endif
    }
    for (int f = 1 + (t >> 6); f < NFR; f += NT / 64) {  // features observed in frame f (as imu_j), in feature order
      const int ln = t & 63;
      int n = 0;
      unsigned am = 0;  // start frames that occur among the frame's factors (one accumulation run of the frame task each)
      for (int e0 = 0; e0 < c.nf; e0 += 64) {
        const int e = min(e0 + ln, MAXE - 1), a = ids[I_FSTART + e];
        const bool in = e0 + ln < c.nf && a < f && f < a + ids[I_FNOBS + e];
        const unsigned long long m = __ballot(in);
        if (in) c.cov[f * MAXE + n + __popcll(m & ((1ull << ln) - 1ull))] = e;
        n += __popcll(m);
#pragma unroll
        for (int aa = 0; aa < NFR - 1; aa++) am |= __any(in && a == aa) ? 1u << aa : 0u;
      }
      if (ln == 0) ids[I_NCOV + f] = n, ids[I_NRUN + f] = __popc(am);
    }
    if (t == 0) ids[I_NCOV] = 0;
#ifdef AVM_X
    // frame 11: the features matched in the relocalization frame (the host's list, in its order)
    for (int k = t; k < c.relo_n; k += NT) c.cov[(NFRP - 1) * MAXE + k] = min(max(B.relo_feat[(size_t)w * B.max_feat + k], 0), max(c.nf - 1, 0));
    if (t == 64) ids[I_NCOV + NFRP - 1] = c.relo_n;
#endif
    __syncthreads();
    PROFQ(c, 39);
#ifdef AVM_X
    if (t == 0) {  // longest-processing-time assignment of the frames to the assembling wavefronts
      int done = 0;
      ids[I_FRW] = -1;
      // (the loads in registers - constant indices only: indexed by a run-time value the array lived in private memory, 53 scratch instructions
      //  in a one-thread loop of 130 steps per window)
      int load[ASM_WAVES];
#pragma unroll
      for (int k = 0; k < ASM_WAVES; k++) load[k] = 0;
      for (int k = 1; k < NFRP; k++) {
        int bb = -1, bn = -1;
        for (int f = 1; f < NFRP; f++)
          if (!(done & (1 << f)) && ids[I_NCOV + f] > bn) bn = ids[I_NCOV + f], bb = f;
        int bw = 0, lb = load[0];
#pragma unroll
        for (int q = 1; q < ASM_WAVES; q++)
          if (load[q] < lb) lb = load[q], bw = q;
        ids[I_FRW + bb] = bw;
        const int inc = ((bn + 63) / 64) * 64 + 8;
#pragma unroll
        for (int q = 0; q < ASM_WAVES; q++) load[q] += q == bw ? inc : 0;
        done |= 1 << bb;
      }
    }
#elif defined(AVM_TP)
    if (t < 64) {
      // Longest-processing-time assignment of the frames to the four wavefronts (lane q keeps the load of wavefront q, in factors).
      // Every wavefront has a SIMD to itself within the workgroup; wavefront 2 also evaluates the raw IMU Jacobians (about two
      // chunks' worth) and two fifths of the prior's rows, wavefront 3 the other three fifths: they start with that load.
      // (Round 5: with the issue priorities those two run at the light level and weigh less than they did: 60 / 300 factors' worth,
      //  re-measured - were 88 / 380: ragged tracks 11.82 -> 11.74 ms, dense 12.54 -> 12.49.)
#ifndef AVM_TP_WIMU
#define AVM_TP_WIMU 60
#endif
#ifndef AVM_TP_WPRI
#define AVM_TP_WPRI 300
#endif
      int fc = t == 2 ? AVM_TP_WIMU + (c.pn > 0 ? 2 * AVM_TP_WPRI / 5 : 0) : (t == 3 && c.pn > 0 ? 3 * AVM_TP_WPRI / 5 : 0), done = 0;
      if (t == 0) ids[I_FRW] = -1;
      for (int k = 1; k < NFRP; k++) {
        constexpr int RUNW = AVM_LPT_RUNW;
        int bb = -1, bn = -1;
        for (int f = 1; f < NFRP; f++) {
          const int n = ids[I_NCOV + f] + RUNW * max(ids[I_NRUN + f] - 1, 0);
          if (!(done & (1 << f)) && n > bn) bn = n, bb = f;
        }
        const int own = (fc + 63) >> 6, with = (fc + bn + 63) >> 6;
        int key = (with << 16) | (own << 8) | t;
        if (t >= ASM_WAVES) key = 0x7fffffff;
#pragma unroll
        for (int o = 2; o > 0; o >>= 1) key = min(key, __shfl_xor(key, o, 64));
        const int bw = __builtin_amdgcn_readfirstlane(key) & 255;
        if (t == bw) fc += bn;
        if (t == 0) ids[I_FRW + bb] = bw;
        done |= 1 << bb;
      }
    }
#else
    if (t < 64) {
      // Longest-processing-time assignment of the frames to the assembling wavefronts, by the lanes of wavefront 0 (lane q
      // keeps the factor count of wavefront q).  A wavefront's cost is its number of 64-factor chunks over ALL its frames;
      // wavefronts w and w + 4 share a SIMD, so the quantity to keep level is the chunk count per SIMD (the raw-IMU
      // wavefront 6 weighs about two chunks on SIMD 2, the prior's wavefront 7 about one on SIMD 3).  Largest frame first,
      // to the wavefront that leaves its SIMD lowest (ties: the one with fewer chunks of its own, then the lower index).
      int fc = 0, done = 0;
      if (t == 0) ids[I_FRW] = -1;
      for (int k = 1; k < NFRP; k++) {
        // (a frame weighs its factors plus RUNW factors' worth for every accumulation run beyond the first - a run costs a flush of
        //  the partial blocks and a group of eight MFMAs however short it is: with ragged tracks a frame has up to ten runs of a
        //  handful of factors each, and by factor counts alone two wavefronts ended up with twice the others' time)
        constexpr int RUNW = AVM_LPT_RUNW;
        int bb = -1, bn = -1;
        for (int f = 1; f < NFRP; f++) {
          const int n = ids[I_NCOV + f] + RUNW * max(ids[I_NRUN + f] - 1, 0);
          if (!(done & (1 << f)) && n > bn) bn = n, bb = f;
        }
        const int own = (fc + 63) >> 6, with = (fc + bn + 63) >> 6;
        const int partner = __shfl(own, t ^ 4, 64);
        int key = ((with + partner + ((t & 3) == 2 ? 2 : ((t & 3) == 3 ? 1 : 0))) << 16) | (own << 8) | t;
        if (t >= ASM_WAVES) key = 0x7fffffff;
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) key = min(key, __shfl_xor(key, o, 64));
        const int bw = __builtin_amdgcn_readfirstlane(key) & 255;
        if (t == bw) fc += bn;
        if (t == 0) ids[I_FRW + bb] = bw;
        done |= 1 << bb;
      }
    }
#endif
    __syncthreads();
    // once per window: the structural zeros of the scratch slot (raw IMU Jacobians outside their blocks, E^T F of
    // the frames that do not observe a feature) - the evaluations only ever rewrite the same nonzero entries
    {
      gdouble* IJR = c.sc + Scratch::IJRAW;
      for (int i = t; i < 10 * 465; i += NT) IJR[i] = 0.0;
      gdouble* Wt = c.sc + Scratch::W;
      for (int idx = t; idx < c.nf * NFR; idx += NT) {
        const int f = idx / c.nf, e = idx - f * c.nf;
        const int a = ids[I_FSTART + e], no = ids[I_FNOBS + e];
        if (f < a || f >= a + no) {
#pragma unroll
          for (int q = 0; q < 6; q++) Wt[(6 * f + q) * WLE + e] = 0.0;
        }
      }
#ifdef AVM_X
      // relocalization frame: E^T F rows 66..71 and the per-factor products of frame 11 are zero except for the matched
      // features, whose entries frame task 11 rewrites at every evaluation
      gdouble* PF = c.sc + Scratch::PF;
      for (int idx = t; idx < MAXE * 6; idx += NT) Wt[(6 * NFR + idx / MAXE) * WLE + idx % MAXE] = 0.0;
      for (int idx = t; idx < MAXE * NQ; idx += NT) PF[((idx / MAXE) * NFRP + (NFRP - 1)) * WLE + idx % MAXE] = 0.0;
#endif
    }
    PROFQ(c, 40);
    // Hp = J0^T J0 (constant during the solve: hoisted out of the per-iteration J^T J)
    if (c.pn > 0) prior_jtj_packed(c.pJ, c.ldp, c.pn, c.sc + Scratch::HP, reinterpret_cast<gint*>(c.sc + Scratch::HP + HPK_MAX));
    __syncthreads();
    PROFQ(c, 41);

    PROF(c, 9);
    // ---------------- TrustRegionMinimizer ----------------
    if (t < 32) lds[L_SUM + t] = 0.0;
    int n_successful = 0, accept_mask = 0;
    double initial_cost = 0;
    double radius = o.initial_trust_region_radius, mu = 1e-8;
    const double min_mu = 1e-8, max_mu = 1.0, mu_inc = 10.0;
    bool reuse = false, first = true, have_alpha = false;
#if defined(AVM_X) || defined(AVM_TP)
    auto DG = [&](int i) { return lds[L_G + i] / lds[L_DD + i]; };  // g / D, recomputed (the same division every time)
#else
    auto DG = [&](int i) { return lds[L_DG + i]; };
#endif
    double alpha = 0, dogleg_step_norm = 0;
    double gnorm = 0, gn_norm = 0, ytg = 0, jusq = 0;  // |g/D|, |D y|, y^T g, |J u|^2
    double k1 = 0, k2 = 0;                             // step = -(k1 * g/D^2 + k2 * y)
    double x_cost = 0, x_norm = 0, gradient_max_norm = 0;
    int iteration = 0, num_invalid = 0, termination = AVM_TERM_NO_CONVERGENCE;
    bool step_ok = true;

    // squared ambient norm over the variable parameter blocks (Ceres' reduced program: constant blocks are not in it)
#ifdef AVM_X
    auto amb_sq = [&](const double* xa, const double* xb) {  // |xa - xb|^2, xb == nullptr: |xa|^2
      double s = 0;
      auto term = [&](int i) {
        const double d = xb ? xa[i] - xb[i] : xa[i];
        s += d * d;
      };
      for (int i = t; i < 7 * NFR; i += NT) term(i);                                 // poses
      if (c.relo_n > 0 && t >= 128 && t < 135) term(7 * NFR + t - 128);             // relo_Pose
      for (int i = t; i < 99 + c.nf; i += NT) term(XSB + i);                          // speed-biases, inverse depths
      if (c.est_ex && t >= 192 && t < 199) term(XEX + t - 192);
      if (c.est_td && t == 200) term(XTD);
      return block_sum1(s);
    };
    auto amb_norm = [&](const double* xs) { return sqrt(amb_sq(xs, nullptr)); };
#else
    auto amb_norm = [&](const double* xs) {
      double s = 0;
      for (int i = t; i < 176 + c.nf; i += NT) s += xs[i] * xs[i];
      return sqrt(block_sum1(s));
    };
#endif
    // evaluate + scaling + gradient max norm at lds[L_X]
    // what follows a Jacobian evaluation at lds[L_X]: scaling, gradient max norm
    auto post_evaluate = [&]() {
      PROF_T0();
      // Jacobi scaling from the column norms of the first Jacobian (diag of unscaled H)
      const bool was_first = first;
      (void)was_first;
      if (first) {
        if (o.jacobi_scaling) {
#ifdef AVM_TP
          if (t < NF) lds[L_SC + t] = 1.0 / (1.0 + sqrt(lds[s_off(t, t)]));
          for (int e = t; e < c.nf; e += NT) lds[L_SC + NF + e] = 1.0 / (1.0 + sqrt(lds[L_HEE + e]));
#else
          if (t < NF) lds[L_SC + t] = 1.0 / (1.0 + sqrt(lds[L_S + roff(t) + t]));
          if (t >= 192 && t < 192 + c.nf) lds[L_SC + NF + t - 192] = 1.0 / (1.0 + sqrt(lds[L_HEE + t - 192]));
#endif
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
#ifdef AVM_TP
        for (int e = t; e < c.nf; e += NT) gm = fmax(gm, fabs(g[NF + e]));
#else
        if (t >= 192 && t < 192 + c.nf) gm = fmax(gm, fabs(g[NF + t - 192]));
#endif
#ifdef AVM_X
        if ((t == 400 && c.relo_n > 0) || (t == 401 && c.est_ex)) {  // relo_Pose / ex_pose: pose blocks like the others
          const int xo = t == 401 ? XEX : 7 * NFR, go = t == 401 ? XC_EX : 6 * NFR;
          for (int k = 0; k < 3; k++) gm = fmax(gm, fabs(g[go + k]));
          quat q{x[xo + 6], x[xo + 3], x[xo + 4], x[xo + 5]};
          quat r = qnormalized(qmul(q, deltaQ(mk3(-g[go + 3], -g[go + 4], -g[go + 5]))));
          gm = fmax(gm, fmax(fmax(fabs(q.x - r.x), fabs(q.y - r.y)), fmax(fabs(q.z - r.z), fabs(q.w - r.w))));
        }
        if (t == 402 && c.est_td) gm = fmax(gm, fabs(g[XC_TD]));
#endif
      }
      gradient_max_norm = block_max1(gm);
      __syncthreads();
      if (c.prof && t == 0) c.prof[43] += clock64() - pt__;
      scale_system(c, was_first);
      PROF(c, 10);
    };
    auto evaluate_x = [&]() {
      x_cost = eval_jac(c, o);
      post_evaluate();
    };
    // eval_jac() stages the frame tasks' rows in the LDS range that also holds the Gauss-Newton step, the dogleg step
    // and the candidate state, so a speculative evaluation parks what a rejection needs (current point, GN step)
    // in the spare tail of the slot's prior region
    gdouble* spec_save = c.sc + Scratch::HP + HPK_MAX + HPK_MAX / 2 + 8;
    static_assert(HPK_MAX + HPK_MAX / 2 + 8 + XN + VEC <= MAXPRIOR * MAXPRIOR, "speculation backup fits the slot");
    auto spec_enter = [&]() {  // x -> backup, x <- candidate
      __syncthreads();
      for (int i = t; i < XN; i += NT) spec_save[i] = lds[L_X + i], lds[L_X + i] = lds[L_XC + i];
      for (int i = t; i < VEC; i += NT) spec_save[XN + i] = lds[L_Y + i];
      __syncthreads();
    };
    auto spec_restore = [&]() {  // x <- backup, candidate <- x
      __syncthreads();
      for (int i = t; i < XN; i += NT) {
        const double cand = lds[L_X + i];
        lds[L_X + i] = spec_save[i];
        lds[L_XC + i] = cand;
      }
      __syncthreads();
    };
    auto spec_restore_gn_step = [&]() {  // after the system at x has been rebuilt (eval_jac stages over it again)
      for (int i = t; i < VEC; i += NT) lds[L_Y + i] = spec_save[XN + i];
      __syncthreads();
    };
    // Speculation (exact: the same evaluations, fewer of them).  Ceres evaluates the cost at the candidate and, if
    // the step is accepted, evaluates residuals AND Jacobians at that same point again.  While steps keep being
    // accepted with a good model fit, the Jacobian is evaluated at the candidate right away (its cost decides the
    // step) and nothing is recomputed on acceptance; a rejected speculation pays one extra evaluation to restore
    // the system at x, and switches speculation off until a step with rho > 0.75 comes by.
    bool speculate = A.speculate != 0;

    x_norm = amb_norm(lds + L_X);
    evaluate_x();
    initial_cost = x_cost;
    double ref_cost = x_cost;

    while (true) {
      PROFQ_T0();
      // FinalizeIterationAndCheckIfMinimizerCanContinue
      if (iteration > 0) {
        if (step_ok) n_successful++;
        if (iteration <= AVM_MAX_ITER_TRACE) {
          if (t == 0) lds[L_SUM + iteration - 1] = x_cost, lds[L_SUM + 16 + iteration - 1] = radius;
          if (step_ok) accept_mask |= 1 << (iteration - 1);
        }
      }
      if (A.time_cap_ticks > 0) {
        // MaxSolverTimeReached (checked before the iteration limit, like Ceres): options.max_solver_time_in_seconds of
        // estimator.cpp:803-806.  One thread reads the clock, the verdict goes through LDS so that it is workgroup-uniform.
        if (t == 0) ids[I_TIMEUP] = wall_clock64() - wall0 >= A.time_cap_ticks;
        __syncthreads();
        if (ids[I_TIMEUP]) {
          termination = AVM_TERM_NO_CONVERGENCE;
          break;
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
#ifdef AVM_TP
        if (t < NF) lds[L_DD + t] = sqrt(fmin(fmax(lds[s_off(t, t)], o.min_lm_diagonal), o.max_lm_diagonal));
        for (int e = t; e < c.nf; e += NT) lds[L_DD + NF + e] = sqrt(fmin(fmax(lds[L_HEE + e], o.min_lm_diagonal), o.max_lm_diagonal));
#else
        if (t < NF) lds[L_DD + t] = sqrt(fmin(fmax(lds[L_S + roff(t) + t], o.min_lm_diagonal), o.max_lm_diagonal));
        if (t >= 192 && t < 192 + c.nf) lds[L_DD + NF + t - 192] = sqrt(fmin(fmax(lds[L_HEE + t - 192], o.min_lm_diagonal), o.max_lm_diagonal));
#endif
        __syncthreads();
        double g2 = 0;
        for (int i = t; i < NF + c.nf; i += NT) {
          const double v = lds[L_G + i] / lds[L_DD + i];
#if !defined(AVM_X) && !defined(AVM_TP)
          lds[L_DG + i] = v;
#endif
          g2 += v * v;
        }
        gnorm = sqrt(block_sum1(g2));
        // Gauss-Newton step with mu retry (DoglegStrategy::ComputeGaussNewtonStep)
        solver_ok = false;
        bool rebuilt = true;
        PROFQ(c, 32);
        while (mu < max_mu) {
          if (!rebuilt) {  // S was destroyed by a failed factorisation: rebuild the normal equations
            evaluate_x();
            rebuilt = true;
          }
          PROF_T0();
          schur_reduce(c, mu);
          PROF(c, 11);
#ifdef AVM_TP
          // factorization + both triangular solves on the register tiles of the four wavefronts (y -> lds[L_Y])
          bool ok;
          AVM_PRIO_BULK_CHOL();  // (its pivot chains raise themselves to 3)
          switch (__builtin_amdgcn_readfirstlane(t >> 6)) {
            case 0: ok = chol_regs<0>(); break;
            case 1: ok = chol_regs<1>(); break;
            case 2: ok = chol_regs<2>(); break;
            default: ok = chol_regs<3>(); break;
          }
          AVM_PRIO_LIGHT();
          PROF(c, 12);
          if (!ok) {
            mu *= mu_inc;
            rebuilt = false;
            continue;
          }
#else
          bool ok;
          if (ids[I_CRFIT]) {  // (uniform: the window's prior has the structure chol_regs' pattern is closed for)
            switch (__builtin_amdgcn_readfirstlane(t >> 6)) {
              case 0: ok = chol_regs<0>(); break;
              case 1: ok = chol_regs<1>(); break;
              case 2: ok = chol_regs<2>(); break;
              case 3: ok = chol_regs<3>(); break;
#ifdef AVM_X
              case 4: ok = chol_regs<4>(); break;
              case 5: ok = chol_regs<5>(); break;
              case 6: ok = chol_regs<6>(); break;
              default: ok = chol_regs<7>(); break;
#else
              default: ok = chol_regs<4>(); break;  // (wavefronts 4..7: no tiles)
#endif
            }
            PROF(c, 12);
          } else {
            ok = cholesky_lds(c.prof);
            PROF(c, 12);
            if (ok) chol_solve_lds(L_Y);
            PROF(c, 13);
          }
          if (!ok) {
            mu *= mu_inc;
            rebuilt = false;
            continue;
          }
#endif
          const double bad_y = back_substitute(c, mu);
          PROF(c, 14);
          if (bad_y > 0) {
            mu *= mu_inc;
            rebuilt = false;
            continue;
          }
          solver_ok = true;
          break;
        }
        PROFQ_T0();
        if (solver_ok) {
          double a1 = 0, a2 = 0;
          for (int i = t; i < NF + c.nf; i += NT) {
            const double yv = lds[L_Y + i], dv = lds[L_DD + i] * yv;
            a1 += dv * dv;
            a2 += yv * lds[L_G + i];
          }
          block_sum1x2(a1, a2);
          gn_norm = sqrt(a1);
          ytg = a2;
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
            for (int i = t; i < VEC; i += NT) lds[L_ST + i] = i < NF + c.nf ? DG(i) / lds[L_DD + i] : 0.0;
            __syncthreads();
            PROFQ(c, 33);
            jusq = jac_times_vec_sq(c, o);
            alpha = gnorm * gnorm / jusq;
            have_alpha = true;
            __syncthreads();
            PROFQ(c, 34);
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
              const double v = -k1 * DG(i) - k2 * lds[L_DD + i] * lds[L_Y + i];
              s2 += v * v;
            }
            dogleg_step_norm = sqrt(block_sum1(s2));
          }
        }
        for (int i = t; i < VEC; i += NT)
          lds[L_ST + i] = i < NF + c.nf ? -(k1 * DG(i) / lds[L_DD + i] + k2 * lds[L_Y + i]) : 0.0;
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
      PROFQ(c, 33);
      PROF_T0();
      state_plus();
      __syncthreads();
#ifdef AVM_X
      const double step_norm = sqrt(amb_sq(lds + L_X, lds + L_XC));
#else
      double d2 = 0;
      for (int i = t; i < 176 + c.nf; i += NT) {
        const double d = lds[L_X + i] - lds[L_XC + i];
        d2 += d * d;
      }
      const double step_norm = sqrt(block_sum1(d2));
#endif
      // the last iteration the options allow: the minimizer stops right after it (the iteration limit is checked before
      // the gradient tolerance, trust_region_minimizer.cc FinalizeIterationAndCheckIfMinimizerCanContinue), so the
      // Jacobian Ceres evaluates at the accepted point is never used: only the cost is computed there
      const bool last_iteration = iteration >= o.max_num_iterations;
      const bool spec = speculate && !last_iteration;
      double cand_cost;
      PROFQ(c, 35);
      if (spec) {
        spec_enter();  // x <- candidate; the current point and the GN step are parked in the slot
        cand_cost = eval_jac(c, o);
        PROF(c, 15);
      } else {
        build_frames(L_XC, 1);
        __syncthreads();
        cand_cost = eval_cost(c, o, L_XC, 1);
        PROF(c, 15);
      }
      PROFQ_T0();
      if (step_norm <= o.parameter_tolerance * (x_norm + o.parameter_tolerance)) {
        if (spec) spec_restore();  // the minimizer stops at the current point, not at the candidate
        termination = AVM_TERM_PARAMETER_TOL;
        break;
      }
      const double cost_change = x_cost - cand_cost;
      if (fabs(cost_change) <= o.function_tolerance * x_cost) {
        if (spec) spec_restore();
        termination = AVM_TERM_FUNCTION_TOL;
        break;
      }
      const double rel = (ref_cost - cand_cost) / model_cost_change;
      if (rel > o.min_relative_decrease) {
        if (spec) {
          // the system at the accepted point is already assembled
          x_norm = amb_norm(lds + L_X);
          x_cost = cand_cost;
          post_evaluate();
        } else {
          __syncthreads();
          for (int i = t; i < XN; i += NT) lds[L_X + i] = lds[L_XC + i];
          __syncthreads();
          x_norm = amb_norm(lds + L_X);
          if (last_iteration)
            x_cost = cand_cost;
          else
            evaluate_x();
        }
        speculate = A.speculate != 0 && rel > 0.75;
        step_ok = true;
        if (rel < 0.25) radius *= 0.5;
        if (rel > 0.75) radius = fmax(radius, 3.0 * dogleg_step_norm);
        mu = fmax(min_mu, 2.0 * mu / mu_inc);
        reuse = false;
        ref_cost = cand_cost;
      } else {
        if (spec) {
          // back to the current point: rebuild its system (g, E^T F, raw IMU Jacobians) for the retried step
          spec_restore();
          evaluate_x();
          spec_restore_gn_step();
        }
        speculate = false;
        radius *= 0.5;
        reuse = true;
      }
      PROFQ(c, 36);
    }
    __syncthreads();
    PROFQ_T0();
    // ---------------- double2vector + vector2double (estimator.cpp:521-587, 477-519) ----------------
    {
      // rot_diff from yaw of frame 0 before / after ; stored in lds[L_GF..+9], origin_P0 in +9..12
#if defined(AVM_X) || defined(AVM_TP)
      constexpr int L_GF = L_Y;  // (the Gauss-Newton step is dead after the loop)
#else
      constexpr int L_GF = L_DG;
#endif
      gauge_rot_diff(B.pose + (size_t)w * 77, (B.failure_occur && B.last_pose0 && B.failure_occur[w]) ? B.last_pose0 + (size_t)w * 7 : nullptr, L_GF);
      __syncthreads();
      if (t < NFRP) {
        const double* rd = lds + L_GF;
        const double* x = lds + L_X;
        quat q = qnormalized(quat{x[t * 7 + 6], x[t * 7 + 3], x[t * 7 + 4], x[t * 7 + 5]});
        double Rq[9], Rs[9];
        q2R(q, Rq);
        mat3mul(rd, Rq, Rs);
        const v3 P = Rmul(rd, mk3(x[t * 7] - lds[L_GF + 12], x[t * 7 + 1] - lds[L_GF + 13], x[t * 7 + 2] - lds[L_GF + 14])) +
                     mk3(lds[L_GF + 9], lds[L_GF + 10], lds[L_GF + 11]);
        const quat qo = R2q(Rs);
        if (t < NFR) {
          const v3 V = Rmul(rd, mk3(x[XSB + t * 9], x[XSB + t * 9 + 1], x[XSB + t * 9 + 2]));
          double* po = B.pose + (size_t)w * 77 + t * 7;
          po[0] = P.x, po[1] = P.y, po[2] = P.z, po[3] = qo.x, po[4] = qo.y, po[5] = qo.z, po[6] = qo.w;
          double* so = B.speedbias + (size_t)w * 99 + t * 9;
          so[0] = V.x, so[1] = V.y, so[2] = V.z;
          for (int k = 3; k < 9; k++) so[k] = x[XSB + t * 9 + k];
        }
#ifdef AVM_X
        else if (c.has_relo) {  // relo_t / relo_r of estimator.cpp:590-596 (frame 11 went through the same transformation; with
                                // no matched feature it did not move in the solve, the gauge fix applies all the same)
          double* po = B.relo_pose + (size_t)w * 7;
          po[0] = P.x, po[1] = P.y, po[2] = P.z, po[3] = qo.x, po[4] = qo.y, po[5] = qo.z, po[6] = qo.w;
        }
#endif
      }
      if (t == 64) {
        double* ex = B.ex_pose + (size_t)w * 7;
        double R[9];
#ifdef AVM_X
        const double* exs = lds + L_X + XEX;  // tic / ric come back from para_Ex_Pose (estimator.cpp:569-579)
        for (int k = 0; k < 3; k++) ex[k] = exs[k];
        q2R(quat{exs[6], exs[3], exs[4], exs[5]}, R);
#else
        q2R(quat{ex[6], ex[3], ex[4], ex[5]}, R);
#endif
        const quat qo = R2q(R);
        ex[3] = qo.x, ex[4] = qo.y, ex[5] = qo.z, ex[6] = qo.w;
      }
#ifdef AVM_X
      if (t == 65 && c.est_td && B.td) B.td[w] = lds[L_X + XTD];
#endif
#ifdef AVM_TP
      for (int e = t; e < c.nf; e += NT) B.inv_depth[(size_t)w * B.max_feat + e] = 1.0 / (1.0 / lds[L_X + XLAM + e]);
#else
      if (t >= 128 && t < 128 + c.nf) {
        const int e = t - 128;
        B.inv_depth[(size_t)w * B.max_feat + e] = 1.0 / (1.0 / lds[L_X + XLAM + e]);
      }
#endif
    }
    PROFQ(c, 37);
    if (c.prof && t == 0) c.prof[31] += 1, c.prof[42] += clock64() - pw__;
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

#if !defined(AVM_X)
// =====================================================================================
// Post-solve marginalization: MarginalizationInfo::addResidualBlockInfo / preMarginalize /
// marginalize / getParameterBlocks (vins_estimator/src/factor/marginalization_factor.cpp:89-319)
// as driven by Estimator::optimization() (estimator.cpp:817-990), one workgroup per window.
//
// Variable layout of the joint system: poses 0..65 | speed-bias 66..164 | ex_pose 165..170 (171 dims,
// packed lower triangle in LDS).  Factors: old prior, IMU factor 0, every projection factor of the
// features that start in frame 0 (with their ex_pose Jacobians) — assembled with the same MFMA X^T X
// scheme as the solve (X row = Jj | Ji | r | Jex).  The inverse depths of those features are
// eliminated first as scalar pivots (they are mutually independent; identical to the reference's joint
// eigen-pseudo-inverse of Amm whenever no eigenvalue is clamped), then pose0/speedbias0 through the
// eigen-decomposition of their 15x15 block with the reference's 1e-8 clamp, and the kept block is
// square-rooted through a second eigen-decomposition (parallel cyclic Jacobi in LDS).
// Deterministic block order (the reference's is address-hash order): kept = poses by frame,
// speed-bias by frame, ex_pose.
namespace mg {
constexpr int MXRS = 68;                              // rows per staged column: HALF a chunk (32 factors x 2 residual rows) + 4 (bank spread)
constexpr int MXSTG = 20 * MXRS;                      // column-major staging tile: Jj 0-5 | Ji 6-11 | r 12 | Jex 13-18 | Jtd 19
#ifdef AVM_TP
// THROUGHPUT form of the marginalization (marginalize_tp_kernel in window_solve_tp.o, round 5): the same phases as a 256-thread
// workgroup inside the throughput build's 80 KB of LDS, so that TWO windows are resident per CU - the kernel is a sequence of short
// latency-bound phases (62 % of its wavefront cycles waiting), and a second window fills them.  What makes it fit: the joint system
// only holds the variables a marginalization can touch - poses | speed-bias 0, 1 | ex_pose | td = 91 instead of 172 (packed 33 KB
// instead of 117): IMU factor 0 reaches speed-biases 0 and 1, the projection factors the poses and ex_pose / td, and the old prior
// whatever it kept last time, which for a prior the reference can build is a subset of these (estimator.cpp:904-916 keeps
// para_SpeedBias[1], shifted to frame 0).  A prior with a speed-bias block of a later frame takes the other kernel (the host checks:
// window_prior_fits_marg_tp).  Speed-biases 0 and 1 keep their indices (66 .. 83), so imu_col() and SB0 + 9 fr hold unchanged.
constexpr int MEX0 = 84, MTD = 90, MVARS = 91;
constexpr int MASM = 4;                               // every wavefront assembles (frames 1 8 9 | 2 7 10 | 3 6 + raw IMU, prior | 4 5 + prior)
#else
constexpr int MEX0 = 165, MTD = 171, MVARS = 172;     // 172 variables: poses | speed-biases | ex_pose | td
constexpr int MASM = 7;                               // assembling wavefronts (staging must stay below row 165: half tiles let seven fit)
#endif
constexpr int MROWS = croff(MVARS);
#ifdef AVM_TP
// LDS of the throughput form: S (4232) | EA EV EB / IMU factor rows (2048) | T (1536) | g_e (152) ... b in the scaling vector's place;
// the staging tiles of phase A lie over everything from row 66 of S to 7684, all of it written after phase A only
constexpr int M_WCH = MROWS;                          // Amm, its eigenvectors / inverse factor, Arm (n x 16); before: the IMU factor's rows
constexpr int M_GT = M_WCH + 2048;                    // T = Arm Amm^+ (n x 16)
constexpr int M_GE = M_GT + 96 * 16;                  // g_e (152)
constexpr int M_G = L_SC;                             // b over the 91 variables (the Jacobi scaling is the solve's)
static_assert(M_GE + 152 <= M_G && MVARS <= VEC && M_G + VEC <= L_X, "marg layout (throughput form)");
static_assert(L_S + SPP + MASM * MXSTG <= M_G, "marg staging must not reach b");
#else
constexpr int M_G = MROWS;                            // b over the 171 variables (176)
constexpr int M_GE = M_G + 176;                       // g_e (152)
constexpr int M_WCH = M_GE + 152;                     // [24][80] Schur staging / IMU factor rows
constexpr int M_GT = L_G;                             // T = Arm Amm^+ in the range of the solve's gradient / scaling vectors (unused here)
constexpr int MWCH = 24;
static_assert(M_WCH + MWCH * WLD <= L_G, "marg layout");
static_assert(SPP + MASM * MXSTG <= 13778, "marg staging must not reach the ex_pose rows (roff(165))");
#endif
constexpr int PARTW = 146;  // aa 21 | g_a 6 | [ex td].pose0 42 | [ex td]^2 28 | g_[ex td] 7 | [ex td].pose_b 42
constexpr int MNW = 73;     // columns of W = E^T F here: 66 pose | 6 ex_pose | 1 td
constexpr int MWS = 80;     // row stride of W[e][.]
}  // namespace mg

// column of the joint system for W column c (0..71): poses, then ex_pose
AVM_DEV int mg_col(int c) { return c < NPOSE ? c : mg::MEX0 + (c - NPOSE); }  // (td: W column 72 -> variable 171)

// One wavefront's share of the elimination of the start-0 inverse depths (marginalization): the tiles (R, C),
// R in {R0, R1}, C in {C0, C1}, C <= R, of  W^T diag(1 / E^T E) W  over the 72 (padded 80) columns of W = E^T F
// (66 pose + 6 ex_pose columns, row-major [e][72] here).  Padded row 72 carries g_e / (E^T E) in place of a W column,
// so tile row 4 also delivers the right-hand-side update.  Operands straight from the scratch slot, 8 k-steps of
// loads in flight, no staging, no barriers.
template <int R0, int R1, int C0, int C1>
AVM_DEV void marg_schur_macro_tile(int nf0) {
  using namespace mg;
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  gcdouble* W = c.sc + Scratch::W;
  const int lane = threadIdx.x & 63, li = lane & 15, lk = lane >> 4;
  constexpr int NR = R1 >= 0 ? 2 : 1, NC = C1 >= 0 ? 2 : 1;
  constexpr int RB[2] = {R0, R1}, CB[2] = {C0, C1};
  constexpr bool SAME = R0 == C0 && R1 == C1;
  constexpr int KB = 8, NW = MNW;
  d4 D[2][2] = {{{0, 0, 0, 0}, {0, 0, 0, 0}}, {{0, 0, 0, 0}, {0, 0, 0, 0}}};
  for (int e0 = 0; e0 < nf0; e0 += 4 * KB) {
    double vr[2][KB], vc[2][KB], fe[KB], xe[KB];
    // (the k index is a summation index: lane group lk takes the 8 consecutive features e0 + 8 lk .. + 7 = 64 contiguous bytes of a
    //  column of Wt, as in schur_macro_tile; rows clamped, masked afterwards; the features beyond nf0 read stale but finite entries of
    //  the region - WLE leaves room for the 8-feature granularity - and are masked out by `on`)
#pragma unroll
    for (int a = 0; a < NR; a++) {
      gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)min(16 * RB[a] + li, NW - 1) * WLE + e0 + 8 * lk);
#pragma unroll
      for (int m2 = 0; m2 < KB / 2; m2++) {
        const dv2 v = src[m2];
        vr[a][2 * m2] = v.x, vr[a][2 * m2 + 1] = v.y;
      }
    }
    if (!SAME) {
#pragma unroll
      for (int b = 0; b < NC; b++) {
        gcdv2* src = reinterpret_cast<gcdv2*>(W + (size_t)min(16 * CB[b] + li, NW - 1) * WLE + e0 + 8 * lk);
#pragma unroll
        for (int m2 = 0; m2 < KB / 2; m2++) {
          const dv2 v = src[m2];
          vc[b][2 * m2] = v.x, vc[b][2 * m2 + 1] = v.y;
        }
      }
    }
#pragma unroll
    for (int m = 0; m < KB; m++) {
      const int ec = min(e0 + 8 * lk + m, nf0 - 1);
      fe[m] = lds[L_HEE + ec], xe[m] = lds[L_HEE + ec] * lds[M_GE + ec];
    }
#pragma unroll
    for (int m = 0; m < KB; m++) {
      const bool on = e0 + 8 * lk + m < nf0;
      double aop[2], bop[2];
#pragma unroll
      for (int a = 0; a < NR; a++) {
        const int col = 16 * RB[a] + li;
        const double w = (on && col < NW) ? vr[a][m] : 0.0;
        aop[a] = col == NW ? (on ? xe[m] : 0.0) : w * fe[m];
        if (SAME) bop[a] = w;
      }
      if (!SAME) {
#pragma unroll
        for (int b = 0; b < NC; b++) bop[b] = (on && 16 * CB[b] + li < NW) ? vc[b][m] : 0.0;
      }
#pragma unroll
      for (int a = 0; a < NR; a++)
#pragma unroll
        for (int b = 0; b < NC; b++)
          if (CB[b] <= RB[a]) D[a][b] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[a], bop[b], D[a][b], 0, 0, 0);
    }
  }
#pragma unroll
  for (int a = 0; a < NR; a++)
#pragma unroll
    for (int b = 0; b < NC; b++) {
      if (CB[b] > RB[a]) continue;
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int gi = 16 * RB[a] + lk + 4 * r, gj = 16 * CB[b] + li;
        if (gi < NW && gj <= gi) {
          const int si = mg_col(gi), sj = mg_col(gj);
          lds[L_S + roff(max(si, sj)) + min(si, sj)] -= D[a][b][r];
        }
        if (gi == NW && gj < NW) lds[M_G + mg_col(gj)] -= D[a][b][r];
      }
    }
}

// ... and the two single-wavefront jobs beside the frame tasks (IMU factor 0's raw Jacobians on one lane, the old prior's residual and gradient)
AVM_NOINL void marg_imu0_raw() {
  const WinCtx& c = lds_ctx();
  double* lds = LDS();
  imu_raw<true>(lds + L_X, lds + L_FR, lds_opt(), c.pdelta, c.pjac, c.psum[0], c.lba, c.lbg, 0, c.sc + Scratch::IJRAW);
}
AVM_NOINL void marg_prior_wave(int rb, int re, int buf_off) { (void)prior_wave<true>(L_X, rb, re, buf_off); }
// Phase D of the marginalization (IMU factor 0: J = sqrt_info [r | J_raw], then J^T J and J^T r into the system) as a function of its own
AVM_NOINL void marg_imu0_gram() {
  const WinCtx& c = lds_ctx();
  using namespace mg;
  double* lds = LDS();
  const int t = threadIdx.x;
  const double* IJR = c.sc + Scratch::IJRAW;
  double* IJ = lds + M_WCH;
  for (int idx = t; idx < 465; idx += NT) {
    const int r = idx / 31, cc = idx % 31;
    // (sqrt_info is stored with zeros below its diagonal: all fifteen products, their thirty loads in flight at once - as a loop
    //  from k = r every step was a trip to the slot of its own)
    double ps[15], ij[15];
#pragma unroll
    for (int k = 0; k < 15; k++) ps[k] = c.psqrt[r * 15 + k], ij[k] = IJR[k * 31 + cc];
    double sacc = 0;
#pragma unroll
    for (int k = 0; k < 15; k++) sacc += k >= r ? ps[k] * ij[k] : 0.0;
    IJ[idx] = sacc;
  }
  __syncthreads();
  for (int q = t; q < 495; q += NT) {
    if (q < 465) {
      int p = 0;
      while ((p + 1) * (p + 2) / 2 <= q) p++;
      const int qq = q - p * (p + 1) / 2;
      double sacc = 0;
      for (int r = 0; r < 15; r++) sacc += IJ[r * 31 + 1 + p] * IJ[r * 31 + 1 + qq];
      const int ip = imu_col(0, p), iq = imu_col(0, qq);
      lds[L_S + roff(max(ip, iq)) + min(ip, iq)] += sacc;
    } else {
      const int p = q - 465;
      double sacc = 0;
      for (int r = 0; r < 15; r++) sacc += IJ[r * 31 + 1 + p] * IJ[r * 31];
      lds[M_G + imu_col(0, p)] += sacc;
    }
  }
  __syncthreads();
}
// Phase B of the marginalization (the per-feature sums) as a function of its own, like marg_schur_phase: its ten-deep load arrays are 140 registers
AVM_NOINL void marg_feature_sums(int nf0) {
  const WinCtx& c = lds_ctx();
  using namespace mg;
  double* lds = LDS();
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int t = threadIdx.x;
  double* W = c.sc + Scratch::W;
  const double* PF = c.sc + Scratch::PF;
  const double* PF2 = c.sc + Scratch::PF + 8 * (size_t)NFR * WLE;
  // (the factor of feature e observed in frame k - these features start in frame 0 - sits at [quantity][k][e])
  // the two heavy items of a feature (f == 0: its own pose block, hee, g_e;  f == 11: the ex_pose / td columns) are dealt
  // densely to the threads; the structural zeros of the frames that do not observe it follow in a loop of their own
  for (int idx = t; idx < nf0 * 2; idx += NT) {
    const int e = idx >> 1, f = (idx & 1) ? 11 : 0;
    const int no = ids[I_FNOBS + e], s0 = ids[I_FOBS + e];
    {
      const double* P = f == 0 ? PF : PF2;
      // all loads of the feature's (<= 10) factors in flight at once, clamped to its last observation and masked
      // (f == 11: the six ex_pose columns and the td column, W columns 66..72)
      double pv[7][NFR - 1];
#pragma unroll
      for (int k = 1; k < NFR; k++)
#pragma unroll
        for (int q = 0; q < 7; q++) {  // (f == 0, q < 3: Ji_t^T Je is minus the observing frame's W entry - marg_frame_task does not store it twice)
          const int kk = min(k, max(no - 1, 0));
          pv[q][k - 1] = (f == 0 && q < 3) ? W[(size_t)(6 * kk + q) * WLE + e] : P[(size_t)(min(q, (f == 0 || !c.est_td) ? 5 : 6) * NFR + kk) * WLE + e];
        }
      double sacc[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int k = 1; k < NFR; k++)
#pragma unroll
        for (int q = 0; q < 7; q++) sacc[q] += k < no ? pv[q][k - 1] : 0.0;
#pragma unroll
      for (int q = 0; q < 6; q++) W[(size_t)(6 * f + q) * WLE + e] = (f == 0 && q < 3) ? -sacc[q] : sacc[q];
      if (f == 11) W[(size_t)72 * WLE + e] = c.est_td ? sacc[6] : 0.0;
      if (f == 0) {
        double hv[2][NFR - 1];
#pragma unroll
        for (int k = 1; k < NFR; k++) {
          const int kk = min(k, max(no - 1, 0));
          hv[0][k - 1] = PF[(size_t)(6 * NFR + kk) * WLE + e], hv[1][k - 1] = PF[(size_t)(7 * NFR + kk) * WLE + e];
        }
        double he = 0, ge = 0;
#pragma unroll
        for (int k = 1; k < NFR; k++) he += k < no ? hv[0][k - 1] : 0.0, ge += k < no ? hv[1][k - 1] : 0.0;
        lds[L_HEE + e] = he;
        lds[M_GE + e] = ge;
      }
    }
  }
  for (int idx = t; idx < nf0 * (NFR - 1); idx += NT) {
    const int e = idx / (NFR - 1), f = 1 + idx % (NFR - 1);
    if (f >= ids[I_FNOBS + e]) {
#pragma unroll
      for (int q = 0; q < 6; q++) W[(size_t)(6 * f + q) * WLE + e] = 0.0;
    }
  }
}

// Phase F of the marginalization as a function of its own (round 6): inlined, its accumulators and operands pushed the kernel body's
// allocation so far that the registers holding SPILLED SGPRs were spilled themselves - every thread-range predicate of the kernel then began
// with a trip to scratch memory (106 sites, 32 of them in this phase's scatter).
AVM_NOINL void marg_schur_phase(int nf0) {
#ifdef AVM_TP
  AVM_PRIO_BULK();
  switch (threadIdx.x >> 6) {  // four wavefronts, one per SIMD: 4 | 3 + 1 | 3 | 2 + 2 tiles (as schur_reduce)
    case 0: marg_schur_macro_tile<2, 3, 0, 1>(nf0); break;
    case 1: marg_schur_macro_tile<0, 1, 0, 1>(nf0), marg_schur_macro_tile<4, -1, 4, -1>(nf0); break;
    case 2: marg_schur_macro_tile<2, 3, 2, 3>(nf0); break;
    default: marg_schur_macro_tile<4, -1, 0, 1>(nf0), marg_schur_macro_tile<4, -1, 2, 3>(nf0); break;
  }
  AVM_PRIO_LIGHT();
#else
  switch (threadIdx.x >> 6) {
    case 0: marg_schur_macro_tile<2, 3, 0, 1>(nf0); break;
    case 1: marg_schur_macro_tile<0, 1, 0, 1>(nf0); break;
    case 2: marg_schur_macro_tile<2, 3, 2, 3>(nf0); break;
    case 3: marg_schur_macro_tile<4, -1, 0, 1>(nf0); break;
    case 7: marg_schur_macro_tile<4, -1, 2, 3>(nf0); break;
    case 5: marg_schur_macro_tile<4, -1, 4, -1>(nf0); break;
    default: break;
  }
#endif
}

AVM_NOINL void marg_frame_task(const WinCtx&, const avm_options&, int b0, int b1, int stage_off) {
  // The wavefront's (at most two) frames b0 < b1 as ONE list of factors, 64 at a time: a chunk may straddle the two frames (5
  // chunks for two frames of 150 factors instead of 3 + 3), the MFMA accumulation is cut at the frame boundary.
  const WinCtx& c = lds_ctx();
  const avm_options& o = lds_opt();
  using namespace mg;
  double* lds = LDS();
  double* stage = lds + stage_off;
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  (void)ids;
  const int lane = threadIdx.x & 63;
  const int n0 = ids[I_NCOV + b0], n1 = b1 < NFR ? ids[I_NCOV + b1] : 0, ntot = n0 + n1;
  Frames fr{lds + L_FR, lds + L_FR + 99};
  const double* xs = lds + L_X;
  const double sqi = o.focal_length / 1.5;
  // FEATURE-MAJOR like the solve's slot (round 3): the lanes of a chunk are consecutive features of one frame, so W / PF / PF2 are
  // written as whole cache lines (they were [feature][column] and [quantity][observation slot]: 8-byte stores 640 and 88 bytes
  // apart, 80 K of this phase's 181 K cycles per window)
  double* W = c.sc + Scratch::W;       // Wt[MNW][WLE]: E^T F, column-major over the features
  double* PF = c.sc + Scratch::PF;     // [8][NFR][WLE] Ji^T Je (6), Je^T Je, Je^T r of the factor (feature e, frame b)
  double* PF2 = c.sc + Scratch::PF + 8 * (size_t)NFR * WLE;  // [7][NFR][WLE] Jex^T Je (6), Jtd^T Je
  const double td = lds[L_RIC + 19];   // para_Td (0 unless estimate_td)
  d4 D00 = {0, 0, 0, 0}, D10 = {0, 0, 0, 0}, D11 = {0, 0, 0, 0}, E00 = {0, 0, 0, 0}, E10 = {0, 0, 0, 0}, E11 = {0, 0, 0, 0};
  const int drow = lane >> 4, dcol = lane & 15;
  // COMPACT (no time offset in the problem: the reference's default): Jj's translation columns are minus Ji's (projection_factor.cpp:
  // 81-95: both are +-reduce ric^T Rj^T), so the staged row is [Jj_r 0-2 | Ji_t 3-5 | Ji_r 6-8 | r 9 | Jex 10-15] - ONE 16-column tile
  // and ONE X^T X product per k-step instead of three; the three Gram tiles the scatter below works on are read back out of it
  // (entries of other lanes through ds_bpermute, signs for the columns that stand for Jj_t) when a frame ends.
  const bool cp = !c.est_td;
  auto gram_get = [&](const d4& G, int Rs, int Cs) {  // entry (Rs, Cs) of a 16 x 16 accumulator tile, for every lane its own
    const int src = (Rs & 3) * 16 + Cs, q = Rs >> 2;
    const double v0 = __shfl(G[0], src, 64), v1 = __shfl(G[1], src, 64), v2 = __shfl(G[2], src, 64), v3 = __shfl(G[3], src, 64);
    return q == 0 ? v0 : (q == 1 ? v1 : (q == 2 ? v2 : v3));
  };
  auto cmap = [](int p, double& sg) {  // column p of [Jj | Ji | r] -> its column in the compact row, and its sign
    sg = p < 3 ? -1.0 : 1.0;
    return p < 3 ? 3 + p : (p < 6 ? p - 3 : (p < 9 ? p - 3 : (p < 12 ? p - 3 : 9)));
  };
  auto end_frame = [&](int b) {  // the blocks frame b owns, from the accumulators
    double* PART = c.sc + Scratch::PART + (size_t)b * PARTW;
    D00 += E00, D10 += E10, D11 += E11;
    if (cp) {
      const d4 G = D00 + D10;  // (all four chains of the one tile)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int row = drow + 4 * r;
        double sr, sc2;
        const int mr = cmap(min(row, 12), sr), mc = cmap(min(dcol, 12), sc2);
        const double g00 = gram_get(G, mr, mc), g10 = gram_get(G, 10 + min(row, 5), mc), g11 = gram_get(G, 10 + min(row, 5), 10 + min(dcol, 5));
        D00[r] = (row < 13 && dcol < 13) ? sr * sc2 * g00 : 0.0;
        D10[r] = (row < 6 && dcol < 13) ? sc2 * g10 : 0.0;
        D11[r] = (row < 6 && dcol < 6) ? g11 : 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const int row = drow + 4 * r;
      // D00: rows/cols over [Jj | Ji | r]
      if (row < 6 && dcol <= row) lds[L_S + roff(6 * b + row) + 6 * b + dcol] = D00[r];                   // (b,b)
      if (row < 6 && dcol >= 6 && dcol < 12) lds[L_S + roff(6 * b + row) + (dcol - 6)] = D00[r];          // (b,0)
      if (row < 6 && dcol == 12) lds[M_G + 6 * b + row] = D00[r];                                         // g_b
      if (row >= 6 && row < 12) {
        const int i = row - 6;
        if (dcol >= 6 && dcol < 12 && dcol - 6 <= i) PART[i * (i + 1) / 2 + (dcol - 6)] = D00[r];         // (0,0)
        if (dcol == 12) PART[21 + i] = D00[r];                                                            // g_0
      }
      // D10: rows = [Jex | Jtd] (7), cols = [Jj | Ji | r]
      if (row < 7) {
        if (dcol < 6) PART[104 + row * 6 + dcol] = D10[r];                      // ([ex td], pose b)
        if (dcol >= 6 && dcol < 12) PART[27 + row * 6 + (dcol - 6)] = D10[r];   // ([ex td], pose 0)
        if (dcol == 12) PART[97 + row] = D10[r];                                // g_[ex td]
        if (dcol <= row) PART[69 + row * (row + 1) / 2 + dcol] = D11[r];        // ([ex td], [ex td])
      }
    }
    D00 = D10 = D11 = E00 = E10 = E11 = d4{0, 0, 0, 0};
  };
  // inputs of a chunk (feature id, its two observations) are fetched one chunk ahead, as in the solve's frame task (round 5: the
  // id and then the observations were two dependent trips to memory at the top of every chunk)
  int e_nx = 0, b_nx = b0, s0_nx = 0;
  double ob_nx[4] = {0, 0, 0, 0};
  auto fetch = [&](int chunk0) {
    const int ic = min(chunk0 + lane, max(ntot - 1, 0));
    b_nx = ic < n0 ? b0 : b1;
    e_nx = c.cov[b_nx * MAXE + (ic < n0 ? ic : ic - n0)];  // (inactive lanes repeat the last factor: valid, never stored)
    s0_nx = ids[I_FOBS + e_nx];
    const int s = s0_nx + b_nx;
    ob_nx[0] = c.obs[2 * s0_nx], ob_nx[1] = c.obs[2 * s0_nx + 1], ob_nx[2] = c.obs[2 * s], ob_nx[3] = c.obs[2 * s + 1];
  };
  if (ntot > 0) fetch(0);
  for (int chunk0 = 0; chunk0 < ntot; chunk0 += 64) {
    const int idx = chunk0 + lane;
    const bool act = idx < ntot;
    const int b = b_nx, e = e_nx, s0 = s0_nx, s = s0 + b;
    const double ob0 = ob_nx[0], ob1 = ob_nx[1], ob2 = ob_nx[2], ob3 = ob_nx[3];
    if (chunk0 + 64 < ntot) fetch(chunk0 + 64);
    double r[2] = {0, 0}, Ji[12], Jj[12], Je[2] = {0, 0}, Jx[12], Jt[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 12; k++) Ji[k] = 0, Jj[k] = 0, Jx[k] = 0;
    if (act) {
      double ob[4] = {ob0, ob1, ob2, ob3}, ai[4] = {0, 0, 0, 0}, aj[4] = {0, 0, 0, 0};
      if (c.est_td) {  // ProjectionTdFactor (estimator.cpp:874-885)
#pragma unroll
        for (int k = 0; k < 4; k++) ai[k] = c.aux[4 * s0 + k], aj[k] = c.aux[4 * s + k];
        td_shift(ob, ai, aj, td, o.tr, o.row);
      }
      proj_eval<true>(xs, fr, lds + L_RIC, lds + L_RIC + 9, ob[0], ob[1], ob[2], ob[3], xs[XLAM + e], 0, b, sqi, o.cauchy_a, true, r, Ji, Jj,
                      Je, Jx, Jt, ai[0], ai[1], aj[0], aj[1]);
      if (!c.est_td) Jt[0] = Jt[1] = 0.0;
#pragma unroll
      for (int k = 0; k < 6; k++) {
        W[(size_t)(6 * b + k) * WLE + e] = Jj[k] * Je[0] + Jj[6 + k] * Je[1];
        if (k >= 3) PF[(size_t)(k * NFR + b) * WLE + e] = Ji[k] * Je[0] + Ji[6 + k] * Je[1];  // (k < 3: minus W's entry, as in the solve's frame task)
        PF2[(size_t)(k * NFR + b) * WLE + e] = Jx[k] * Je[0] + Jx[6 + k] * Je[1];
      }
      PF[(size_t)(6 * NFR + b) * WLE + e] = Je[0] * Je[0] + Je[1] * Je[1];
      PF[(size_t)(7 * NFR + b) * WLE + e] = Je[0] * r[0] + Je[1] * r[1];
      if (c.est_td) PF2[(size_t)(6 * NFR + b) * WLE + e] = Jt[0] * Je[0] + Jt[1] * Je[1];  // (without a time offset the per-feature sums take a zero instead)
    }
    // staged column-major like the solve kernel's frame tasks (Jj 0-5 | Ji 6-11 | r 12 | Jex 13-18): one 16-byte store
    // per column, contiguous across the lanes; inactive lanes stage zeros, so no row needs masking.  The tile holds half
    // a chunk: lanes 0-31 stage and the wavefront multiplies, then lanes 32-63.
    const int nact = min(64, ntot - chunk0);
#pragma unroll 1
    for (int half = 0; half < 2; half++) {
      const int nh = min(max(nact - 32 * half, 0), 32);
      if (nh == 0) break;  // (uniform)
      if ((lane >> 5) == half) {
        dv2* st = reinterpret_cast<dv2*>(stage) + (lane & 31);
        if (cp) {
#pragma unroll
          for (int k = 0; k < 3; k++) {
            st[k * (MXRS / 2)] = dv2{Jj[3 + k], Jj[9 + k]};
            st[(3 + k) * (MXRS / 2)] = dv2{Ji[k], Ji[6 + k]};
            st[(6 + k) * (MXRS / 2)] = dv2{Ji[3 + k], Ji[9 + k]};
          }
          st[9 * (MXRS / 2)] = dv2{r[0], r[1]};
#pragma unroll
          for (int k = 0; k < 6; k++) st[(10 + k) * (MXRS / 2)] = dv2{Jx[k], Jx[6 + k]};
        } else {
#pragma unroll
          for (int k = 0; k < 6; k++) {
            st[k * (MXRS / 2)] = dv2{Jj[k], Jj[6 + k]};
            st[(6 + k) * (MXRS / 2)] = dv2{Ji[k], Ji[6 + k]};
            st[(13 + k) * (MXRS / 2)] = dv2{Jx[k], Jx[6 + k]};
          }
          st[12 * (MXRS / 2)] = dv2{r[0], r[1]};
          st[19 * (MXRS / 2)] = dv2{Jt[0], Jt[1]};
        }
      }
      wave_lds_sync();
      // the factors of frame b0 in this half, then those of b1 (either may be empty)
      const int g0 = chunk0 + 32 * half;                      // list position of the half's first factor
      const int nb0 = min(max(n0 - g0, 0), nh);               // factors of b0 in the half
#pragma unroll 1
      for (int run = 0; run < 2; run++) {
        const int l = run == 0 ? 0 : nb0, l_end = run == 0 ? nb0 : nh;
        if (l_end <= l) continue;  // (uniform)
        if (run == 1 && g0 + l == n0 && n0 > 0) end_frame(b0);  // frame b1 begins exactly here: frame b0 is complete
        // lane group drow takes the two rows of factor 4 j + drow (one 16-byte read per tile), four j at a time: 24 MFMAs on
        // six independent chains; factors outside the run are masked out by their index
        const int j_end = (l_end + 3) >> 2;
        if (cp) {  // one tile: two MFMAs (the two residual rows) per k-step, eight in flight
#pragma unroll 1
          for (int j0 = l >> 2; j0 < j_end; j0 += 4) {
            dv2 u0[4];
#pragma unroll
            for (int u = 0; u < 4; u++) u0[u] = *reinterpret_cast<const dv2*>(stage + dcol * MXRS + 8 * min(j0 + u, 7) + 2 * drow);
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int f = 4 * (j0 + u) + drow;
              const bool on = f >= l && f < l_end;
              const double a0 = on ? u0[u][0] : 0.0, a1 = on ? u0[u][1] : 0.0;
              if (u & 1) {
                D10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, D10, 0, 0, 0);  // (D10 / E10: the second pair of chains of the
                E10 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, E10, 0, 0, 0);  //  same tile, folded into D00 below)
              } else {
                D00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, D00, 0, 0, 0);
                E00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, E00, 0, 0, 0);
              }
            }
          }
          continue;
        }
#pragma unroll 1
        for (int j0 = l >> 2; j0 < j_end; j0 += 4) {
          dv2 u0[4], u1[4];
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int ro = 8 * min(j0 + u, 7) + 2 * drow;
            u0[u] = *reinterpret_cast<const dv2*>(stage + min(dcol, 12) * MXRS + ro);
            u1[u] = *reinterpret_cast<const dv2*>(stage + (13 + min(dcol, 6)) * MXRS + ro);
          }
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int f = 4 * (j0 + u) + drow;
            const bool on = f >= l && f < l_end;
            const double a0 = (on && dcol < 13) ? u0[u][0] : 0.0, a1 = (on && dcol < 13) ? u0[u][1] : 0.0;
            const double x0 = (on && dcol < 7) ? u1[u][0] : 0.0, x1 = (on && dcol < 7) ? u1[u][1] : 0.0;
            D00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, a0, D00, 0, 0, 0);
            D10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, a0, D10, 0, 0, 0);
            D11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x0, x0, D11, 0, 0, 0);
            E00 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, a1, E00, 0, 0, 0);
            E10 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, a1, E10, 0, 0, 0);
            E11 = __builtin_amdgcn_mfma_f64_16x16x4f64(x1, x1, E11, 0, 0, 0);
          }
        }
      }
      wave_lds_sync();
    }
  }
  // what is still in the accumulators belongs to the last frame with factors; a frame without factors owns zeros
  if (n1 > 0) {
    end_frame(b1);
    if (n0 == 0) end_frame(b0);
  } else {
    end_frame(b0);
    if (b1 < NFR) end_frame(b1);
  }
}

// Cyclic Jacobi eigen-decomposition of the symmetric n x n matrix A (row-major, leading dimension ld) in LDS.
// Only the LOWER triangle of A is read and written.  On return the diagonal of A holds the eigenvalues and
// the columns of V the eigenvectors (A0 = V diag V^T).
// Round-robin pairing: n/2 disjoint rotations per step.  A <- J^T A J is applied as independent 2x2 blocks
// (rows of pair k1, columns of pair k2, k1 >= k2); V <- V J with threads grouped by pair so the rotation is
// loaded once for several rows.  The step is LDS-instruction bound, so every access is kept to the minimum:
// rotation table read as double2 / int2, no mirrored writes.  Two barriers per step.
template <int NTH>
AVM_NOINL int jacobi_eig_lds(int A_off, int V_off, int n, int ld, int rot_off) {
  double* A = LDS() + A_off;
  double* V = LDS() + V_off;
  double2* rcs = reinterpret_cast<double2*>(LDS() + rot_off);        // [np] (c, s)
  int2* rpq = reinterpret_cast<int2*>(LDS() + rot_off + 2 * 64);     // [np] (p, q), p < q
  double* red = LDS() + L_RED;
  constexpr bool WAVE = NTH == 64;  // a single wavefront: wave-level ordering of its LDS traffic is enough
  auto sync = [&]() {
    if (WAVE)
      wave_lds_sync();
    else
      __syncthreads();
  };
  const int t = WAVE ? (threadIdx.x & 63) : threadIdx.x;
  const int ne = (n + 1) & ~1, np = ne >> 1;
  for (int i = t; i < n * n; i += NTH) V[(i / n) * ld + i % n] = (i / n == i % n) ? 1.0 : 0.0;
  // static work assignment
  //  - blocks (k1 >= k2): up to MAXB per thread
  //  - V: thread -> pair kv = t / tpp, rows (t % tpp) + tpp * m
  constexpr int MAXB = 3, MAXR = 8;
  const int nblk = np * (np + 1) / 2;
  short bk1[MAXB], bk2[MAXB];
#pragma unroll
  for (int u = 0; u < MAXB; u++) {
    const int idx = t + u * NTH;
    bk1[u] = -1, bk2[u] = 0;
    if (idx < nblk) {
      int k1 = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
      while ((k1 + 1) * (k1 + 2) / 2 <= idx) k1++;
      while (k1 * (k1 + 1) / 2 > idx) k1--;
      bk1[u] = (short)k1, bk2[u] = (short)(idx - k1 * (k1 + 1) / 2);
    }
  }
  const int tpp = max(1, NTH / np);          // threads per pair for the V update
  const int kv = t / tpp, rv0 = t % tpp;     // pair and first row of this thread (kv >= np: idle)
  sync();
  auto Lw = [&](int i, int j) -> double& { return A[max(i, j) * ld + min(i, j)]; };
  int sweeps = 0;
  for (int sweep = 0; sweep < 20; sweep++) {
    // converged when every |a_pq| <= tol sqrt(a_pp a_qq) (relative criterion: keeps the small eigenvalues
    // accurate, which matters for the 1e-8 clamp next to eigenvalues of 1e12)
    double off = 0;
    for (int i = t; i < n * n; i += NTH) {
      const int r = i / n, q = i % n;
      if (r <= q) continue;
      const double v = fabs(A[r * ld + q]);
      const double sc = sqrt(fabs(A[r * ld + r]) * fabs(A[q * ld + q]));
      off = fmax(off, sc > 0.0 ? v / sc : (v > 0.0 ? 1.0 : 0.0));
    }
    if (WAVE) {
      off = wave_max(off);
    } else {
      off = block_max<NTH>(off, red);
    }
    if (off <= 1e-15) break;
    sweeps++;
    for (int step = 0; step < ne - 1; step++) {
      if (t < np) {
        const int a = t == 0 ? ne - 1 : (step + t) % (ne - 1);
        const int b = t == 0 ? step : (step - t + (ne - 1)) % (ne - 1);
        const int pI = min(a, b), qI = max(a, b);
        double cs = 1.0, sn = 0.0;
        if (qI < n) {
          const double apq = A[qI * ld + pI];
          if (fabs(apq) > 1e-300) {
            const double tau = (A[qI * ld + qI] - A[pI * ld + pI]) / (2.0 * apq);
            const double tt = (tau >= 0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            cs = fast_rsqrt(1.0 + tt * tt);
            sn = tt * cs;
          }
        }
        rcs[t] = double2{cs, sn};
        rpq[t] = int2{pI, qI};
      }
      sync();
#pragma unroll
      for (int u = 0; u < MAXB; u++) {
        if (bk1[u] < 0) continue;
        const int k1 = bk1[u], k2 = bk2[u];
        const int2 pq1 = rpq[k1], pq2 = rpq[k2];
        const double2 r1v = rcs[k1], r2v = rcs[k2];
        const int p1 = pq1.x, q1 = pq1.y, p2 = pq2.x, q2 = pq2.y;
        const double c1 = r1v.x, s1 = r1v.y, c2 = r2v.x, s2 = r2v.y;
        const bool r1 = q1 < n, r2 = q2 < n;  // a dummy partner (odd n) leaves its line untouched (c = 1, s = 0)
        if (k1 != k2) {
          double& e00 = Lw(p1, p2);
          const double a00 = e00, a01 = r2 ? Lw(p1, q2) : 0.0, a10 = r1 ? Lw(q1, p2) : 0.0, a11 = (r1 && r2) ? Lw(q1, q2) : 0.0;
          const double b00 = c1 * a00 - s1 * a10, b01 = c1 * a01 - s1 * a11;
          const double b10 = s1 * a00 + c1 * a10, b11 = s1 * a01 + c1 * a11;
          e00 = c2 * b00 - s2 * b01;
          if (r2) Lw(p1, q2) = s2 * b00 + c2 * b01;
          if (r1) Lw(q1, p2) = c2 * b10 - s2 * b11;
          if (r1 && r2) Lw(q1, q2) = s2 * b10 + c2 * b11;
        } else {
          // diagonal block of the pair itself: [app apq; apq aqq] -> diag(app - t apq, aqq + t apq)
          const double app = A[p1 * ld + p1];
          if (r1) {
            const double aqq = A[q1 * ld + q1], apq = A[q1 * ld + p1];
            A[p1 * ld + p1] = c1 * c1 * app - 2.0 * c1 * s1 * apq + s1 * s1 * aqq;
            A[q1 * ld + q1] = s1 * s1 * app + 2.0 * c1 * s1 * apq + c1 * c1 * aqq;
            A[q1 * ld + p1] = (c1 * c1 - s1 * s1) * apq + c1 * s1 * (app - aqq);
          }
        }
      }
      if (kv < np) {
        const int2 pq = rpq[kv];
        if (pq.y < n) {
          const double2 cs2 = rcs[kv];
#pragma unroll
          for (int m = 0; m < MAXR; m++) {
            const int i = rv0 + tpp * m;
            if (i < n) {
              const double x = V[i * ld + pq.x], y = V[i * ld + pq.y];
              V[i * ld + pq.x] = cs2.x * x - cs2.y * y;
              V[i * ld + pq.y] = cs2.y * x + cs2.x * y;
            }
          }
        }
      }
      sync();
    }
  }
  return sweeps;
}

// Fast path of the 16 x 16 pseudo-inverse of the marginalization (Amm^+ = V diag(lambda > eps ? 1 / lambda : 0) V^T,
// marginalization_factor.cpp:283-286) for the usual case that NO eigenvalue is clamped: then Amm^+ is the plain inverse,
// which one wavefront gets from the same register-resident square-root-free Cholesky as the solve's diagonal blocks
// (lanes 0..15 = rows, lanes 16..31 = rows of the identity -> L^-T), ~3K cycles instead of ~135K for the Jacobi sweeps.
// The condition is checked rigorously: lambda_min >= 1 / trace(Amm^-1), so "trace(Amm^-1) < 1 / eps" (and positive
// pivots) proves that every eigenvalue is above eps; otherwise the caller falls back to the eigen-decomposition.
// On success the result is handed over in the eigen-solver's output format: EV[i][c] = (L D^1/2)^-T rows, diag(EA) = the
// pivots d_c, so that EV diag(1 / d) EV^T = Amm^-1.  EA is left untouched on failure.  Call with one full wavefront.
AVM_NOINL bool pinv16_cholesky(double* EA, double* EV, int m, double eps) {  // (outlined: its sixteen-register row was spilled inside the kernel body)
  constexpr int NB = 16;
  const int r = threadIdx.x & 63;
  const bool idl = (r & 48) == 16;
  double a[NB];
  {
    const int rc = r & 15;
#pragma unroll
    for (int k = 0; k < NB; k++) a[k] = idl ? (rc == k ? 1.0 : 0.0) : EA[rc * NB + min(k, rc)];
  }
  double uprev = 0.0, dvec = 1.0;
#pragma unroll
  for (int j = 0; j < NB; j++) {
    if (j > 0) a[j] = fma(-uprev, readlane_d(a[j - 1], j), a[j]);
    const double djj = readlane_d(a[j], j);
    dvec = (r & 15) == j ? djj : dvec;
    double y = __builtin_amdgcn_rcp(djj), e = 0;
#define AVM_TAIL(slot)                                                                                                 \
  if (j > 0) {                                                                                                         \
    double sk[3];                                                                                                      \
    _Pragma("unroll") for (int q = 0; q < 3; q++) sk[q] = readlane_d(a[j - 1], min(j + 1 + (slot) + 5 * q, NB - 1));   \
    _Pragma("unroll") for (int q = 0; q < 3; q++)                                                                      \
      if (j + 1 + (slot) + 5 * q < NB) a[j + 1 + (slot) + 5 * q] = fma(-uprev, sk[q], a[j + 1 + (slot) + 5 * q]);      \
  }
    AVM_TAIL(0)
    e = fma(-djj, y, 1.0);
    AVM_TAIL(1)
    y = fma(y, e, y);
    AVM_TAIL(2)
    e = fma(-djj, y, 1.0);
    AVM_TAIL(3)
    y = fma(y, e, y);
    AVM_TAIL(4)
#undef AVM_TAIL
    uprev = a[j] * y;
  }
  // trace(Amm^-1) = sum_i sum_c x_i[c]^2 / d_c over the real indices; pivots must be positive
  double tr = 0.0;
  bool bad = false;
#pragma unroll
  for (int c = 0; c < NB; c++) {
    const double dc = readlane_d(dvec, c);
    if (c < m) {
      bad |= !(dc > 0.0);
      tr = fma(a[c] * a[c], 1.0 / dc, tr);
    }
  }
  tr = (idl && (r & 15) < m) ? tr : 0.0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) tr += __shfl_xor(tr, off, 64);
  const bool fast = !bad && tr * eps < 1.0;  // (NaN compares false)
  if (fast) {
    if (idl) {
#pragma unroll
      for (int c = 0; c < NB; c++) EV[(r & 15) * NB + c] = a[c];
    }
    wave_lds_sync();
    if (r < NB) EA[r * NB + r] = dvec;  // pad indices (>= m) carry pivot 1 and are masked by the consumer
  }
  return fast;
}

#ifdef AVM_TP
#define AVM_MARG_KERNEL marginalize_tp_kernel
#define AVM_MARG_OCC __attribute__((amdgpu_waves_per_eu(2, 2)))  // two four-wavefront workgroups per CU, like the solve beside it
#else
#define AVM_MARG_KERNEL marginalize_kernel
#define AVM_MARG_OCC
#endif
__global__ __launch_bounds__(NT) AVM_MARG_OCC void AVM_MARG_KERNEL(SolveArgs A, avm_prior_out PO, int* err, double* scale_out) {
  lds_base_check();
  AVM_PRIO_LIGHT();
  using namespace mg;
  double* lds = LDS();
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  const avm_options& o = lds_opt();
  const avm_window_batch& B = A.b;
  const int flag = A.opt.marginalization_flag;
  for (int w = blockIdx.x; w < B.n_windows; w += gridDim.x) {
    WinCtx cl;
    cl.prof = A.prof ? as_global(A.prof + (size_t)blockIdx.x * PROF_SLOTS) : nullptr;
    cl.sc = as_global(A.scratch + (size_t)blockIdx.x * Scratch::TOTAL);
    cl.osf = as_global(A.iscratch + (size_t)blockIdx.x * ISCRATCH);
    cl.cov = cl.osf + MAXOBS;
    cl.w = w;
    cl.nf = B.n_feat[w];
    cl.obs = as_global(B.obs_xy + (size_t)w * B.max_obs * 2);
    cl.pdelta = as_global(A.pre_delta + (size_t)w * 100), cl.pjac = as_global(A.pre_jac + (size_t)w * 2250), cl.psqrt = as_global(A.pre_sqrt + (size_t)w * 2250);
    cl.psum = as_global(A.pre_sum_dt + (size_t)w * 10);
    cl.lba = as_global(B.imu_lin_ba + (size_t)w * 30), cl.lbg = as_global(B.imu_lin_bg + (size_t)w * 30);
    cl.pn = B.prior_n ? B.prior_n[w] : 0;
    cl.pnblk = cl.pn > 0 ? B.prior_nblk[w] : 0;
    cl.ldp = B.max_prior;
    cl.pJ = as_global(B.prior_J + (size_t)w * B.max_prior * B.max_prior);
    cl.pr = as_global(B.prior_r + (size_t)w * B.max_prior);
    cl.px0 = as_global(B.prior_x0 + (size_t)w * B.max_pblk * 9);
    cl.nobs_tot = 0;
    cl.est_ex = 0, cl.est_td = (A.opt.estimate_td != 0 && B.obs_vel_td && B.td) ? 1 : 0;
    cl.aux = cl.est_td ? as_global(B.obs_vel_td + (size_t)w * B.max_obs * 4) : nullptr;
    cl.relo_n = 0, cl.has_relo = 0, cl.relo_xy = nullptr;  // (the relocalization factors take no part in the marginalization)
    __syncthreads();  // the previous window's readers of the LDS context are done
    lds_store_ctx(cl, A.opt);
    const WinCtx& c = lds_ctx();
    __syncthreads();
    PROF_T0();
    // ---- load the post-solve state and tables.  Every table entry of this thread is requested before the first one is stored (round 5:
    // written as one loop per table, each load waited for its own store - eight dependent trips to memory per window, half of this phase)
    static_assert(NT >= 160 && 99 <= NT && MAXE <= NT, "one entry of every table per thread");
    {
      const int nfl = c.nf, npb = c.pnblk;
      constexpr int PBT0 = NT >= 512 ? 256 : 160;
      const bool in_f = t < nfl, in_pb = t >= PBT0 && t < PBT0 + npb;
      const int kpb = in_pb ? t - PBT0 : 0;
      const double v_pose = B.pose[(size_t)w * 77 + min(t, 76)], v_sb = B.speedbias[(size_t)w * 99 + min(t, 98)];
      const double v_lam = in_f ? B.inv_depth[(size_t)w * B.max_feat + t] : 1.0;
      const size_t kf = (size_t)w * B.max_feat + (in_f ? t : 0);
      const int v_fs = B.feat_start[kf], v_fn = B.feat_nobs[kf], v_fo = B.feat_obs_begin[kf];
      const double v_ex = B.ex_pose[(size_t)w * 7 + min(t, 6)];
      const double v_td = (t == 7 && c.est_td) ? B.td[w] : 0.0;
      const int v_pk = in_pb ? B.prior_blk_kind[(size_t)w * B.max_pblk + kpb] : 0, v_pf = in_pb ? B.prior_blk_frame[(size_t)w * B.max_pblk + kpb] : 0;
      double ex[7] = {0, 0, 0, 0, 0, 0, 1};
      if (t == 0) {
#pragma unroll
        for (int k = 0; k < 7; k++) ex[k] = B.ex_pose[(size_t)w * 7 + k];
      }
      __builtin_amdgcn_sched_barrier(0);
      for (int i = t; i < MAXPRIOR; i += NT) lds[L_DXP + i] = 0.0, lds[L_RP + i] = 0.0;
#ifdef AVM_TP
      for (int i = t; i < MROWS; i += NT) lds[L_S + i] = 0.0;
      for (int i = t; i < VEC; i += NT) lds[M_G + i] = 0.0;
      for (int i = t; i < 152; i += NT) lds[M_GE + i] = 0.0;
#else
      for (int i = t; i < MROWS + 176 + 152; i += NT) lds[i] = 0.0;  // S, b, g_e
#endif
      for (int i = t; i < 152; i += NT) lds[L_HEE + i] = 0.0;
      if (t < 77) lds[L_X + t] = v_pose;
      if (t < 99) lds[L_X + XSB + t] = v_sb;
      if (t < MAXE) lds[L_X + XLAM + t] = v_lam;
      if (in_f) ids[I_FSTART + t] = v_fs, ids[I_FNOBS + t] = v_fn, ids[I_FOBS + t] = v_fo;
      if (t < 7) lds[L_RIC + 12 + t] = v_ex;
      if (t == 7) lds[L_RIC + 19] = v_td;  // para_Td
      if (in_pb) ids[I_PBLK + kpb * 3] = v_pk, ids[I_PBLK + kpb * 3 + 1] = v_pf;  // the prior's block table
      if (t == 0) {
        double R[9];
        q2R(quat{ex[6], ex[3], ex[4], ex[5]}, R);
        for (int k = 0; k < 9; k++) lds[L_RIC + k] = R[k];
        for (int k = 0; k < 3; k++) lds[L_RIC + 9 + k] = ex[k];
      }
    }
    __syncthreads();
    if (t == 0) {  // offsets and state columns of the prior's blocks (read after the barrier that precedes phase A)
      int off = 0;
      for (int k = 0; k < c.pnblk; k++) {
        const int kind = ids[I_PBLK + k * 3], fr = ids[I_PBLK + k * 3 + 1];
        ids[I_PBLK + k * 3 + 2] = off;
        const int n = kind == AVM_BLK_SPEEDBIAS ? 9 : (kind == AVM_BLK_TD ? 1 : 6);
        for (int q = 0; q < n; q++)
          ids[I_PIDX + off + q] = kind == AVM_BLK_POSE ? fr * 6 + q : (kind == AVM_BLK_SPEEDBIAS ? SB0 + fr * 9 + q : (kind == AVM_BLK_TD ? MTD : MEX0 + q));
        off += n;
      }
    }
    // does the prior take part?  MARGIN_SECOND_NEW needs pose[WINDOW_SIZE-1] in it (estimator.cpp:926-927)
    bool use_prior = c.pn > 0;
    bool has9 = false;
    for (int k = 0; k < c.pnblk; k++)
      if (ids[I_PBLK + k * 3] == AVM_BLK_POSE && ids[I_PBLK + k * 3 + 1] == AVM_WINDOW_SIZE - 1) has9 = true;
    if (flag == AVM_MARGIN_SECOND_NEW && !(use_prior && has9)) {
      if (t == 0) PO.n[w] = -1, PO.nblk[w] = 0;  // nothing to do: the caller keeps the old prior
      continue;
    }
    const bool imu0 = flag == AVM_MARGIN_OLD && c.psum[0] < o.max_sum_dt;  // estimator.cpp:841
    if (flag == AVM_MARGIN_OLD) {
      for (int f = 1 + wv; f < NFR; f += NT / 64) {  // start-frame-0 features observed in frame f (ballot compaction, see the solve)
        int n = 0;
        for (int e0 = 0; e0 < c.nf; e0 += 64) {
          const int e = min(e0 + lane, MAXE - 1);
          const bool in = e0 + lane < c.nf && ids[I_FSTART + e] == 0 && f < ids[I_FNOBS + e];
          const unsigned long long m = __ballot(in);
          if (in) c.cov[f * MAXE + n + __popcll(m & ((1ull << lane) - 1ull))] = e;
          n += __popcll(m);
        }
        if (lane == 0) ids[I_NCOV + f] = n;
      }
    } else if (t < NFR) {
      ids[I_NCOV + t] = 0;
    }
    if (t == 0) ids[I_NCOV] = 0;
    build_frames(L_X, 0);
    double* IJR = c.sc + Scratch::IJRAW;
    for (int i = t; i < 465; i += NT) IJR[i] = 0.0;
    __syncthreads();
    int nf0 = 0;  // features starting at frame 0 (they come first)
    for (int e0 = 0; e0 < c.nf; e0 += 64) nf0 += __popcll(__ballot(e0 + lane < c.nf && ids[I_FSTART + min(e0 + lane, MAXE - 1)] == 0));
    PROF(c, 16);
    // ---- phase A: projection factors of the start-0 features || IMU factor 0
#ifdef AVM_TP
    {
      // four wavefronts, one per SIMD: frames {1 8 9} {2 7 10} {3 6} {4 5} (a start-0 feature's track ends early or late: the factor
      // counts fall with the frame, and this deal keeps the sums level), a pair as one list of factors, the third frame after it;
      // wavefront 2 then takes IMU factor 0 and two fifths of the old prior's rows, wavefront 3 the other three fifths (each reads J0
      // along its own rows only; the partial gradients are added in phase E, as in the solve)
      static_assert(NFR == 11 && MASM == 4, "the deal below");
      const int stage = L_S + SPP + wv * MXSTG;
      AVM_PRIO_BULK();
      switch (wv) {
        case 0: marg_frame_task(c, o, 1, 8, stage), marg_frame_task(c, o, 9, NFR, stage); break;
        case 1: marg_frame_task(c, o, 2, 7, stage), marg_frame_task(c, o, 10, NFR, stage); break;
        case 2: marg_frame_task(c, o, 3, 6, stage); break;
        default: marg_frame_task(c, o, 4, 5, stage); break;
      }
      AVM_PRIO_LIGHT();
      if (wv == 2 && lane == 0 && imu0) marg_imu0_raw();
      if (wv >= 2 && use_prior) {
        const int h = (3 * c.pn + 2) / 5;
        if (wv == 2)
          marg_prior_wave(h, c.pn, L_DX2);
        else
          marg_prior_wave(0, h, L_DXP);
      }
    }
#else
    if (wv < MASM) {
      marg_frame_task(c, o, 1 + wv, 1 + wv + MASM, L_S + SPP + wv * MXSTG);  // this wavefront's (at most two) frames
      static_assert(1 + 2 * MASM >= NFR, "two frames per wavefront cover all frames");
    } else if (wv == 7) {
      if (lane == 0 && imu0) marg_imu0_raw();
      // ... and the old prior's residual and gradient (MarginalizationFactor at the current state): dx, r_p, J0^T r_p
      if (use_prior) marg_prior_wave(0, c.pn, L_DXP);
    }
#endif
    __syncthreads();
    PROF(c, 17);
    // ---- phase B: per-feature sums, PART gather
    marg_feature_sums(nf0);
    __syncthreads();  // staging dead: rows >= 66 can be cleared, then the PART sums land (incl. the ex_pose rows)
    for (int i = SPP + t; i < MROWS; i += NT) lds[L_S + i] = 0.0;
    __syncthreads();
    if (flag == AVM_MARGIN_OLD && t < PARTW) {
      const double* PART = c.sc + Scratch::PART;
      const int q = t;  // (rows MEX0 .. MEX0 + 6 = the six ex_pose variables and td)
      // (every frame's PART row was written by its frame task - zeros for a frame without factors -, so all ten loads go out at once:
      //  behind the `I_NCOV > 0` test they were ten dependent trips to the slot)
      double pv[NFR - 1];
#pragma unroll
      for (int b = 1; b < NFR; b++) pv[b - 1] = PART[(size_t)b * PARTW + q];
      if (q < 104) {
        double sacc = 0;
#pragma unroll
        for (int b = 1; b < NFR; b++) sacc += ids[I_NCOV + b] > 0 ? pv[b - 1] : 0.0;
        if (q < 21) {
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= q) i++;
          lds[L_S + roff(i) + (q - i * (i + 1) / 2)] = sacc;
        } else if (q < 27) {
          lds[M_G + (q - 21)] = sacc;
        } else if (q < 69) {
          lds[L_S + roff(MEX0 + (q - 27) / 6) + (q - 27) % 6] = sacc;
        } else if (q < 97) {
          const int k = q - 69;
          int i = 0;
          while ((i + 1) * (i + 2) / 2 <= k) i++;
          lds[L_S + roff(MEX0 + i) + MEX0 + (k - i * (i + 1) / 2)] = sacc;
        } else {
          lds[M_G + MEX0 + (q - 97)] = sacc;
        }
      } else {
        const int k = q - 104;
#pragma unroll
        for (int b = 1; b < NFR; b++)
          if (ids[I_NCOV + b] > 0) lds[L_S + roff(MEX0 + k / 6) + 6 * b + k % 6] = pv[b - 1];
      }
    }
    __syncthreads();
    PROF(c, 18);
    // ---- phase D: IMU factor 0
    if (imu0) marg_imu0_gram();
    PROF(c, 19);
    // ---- phase E: old prior (MarginalizationFactor at the current state)
    if (use_prior) {
      const int* pidx = ids + I_PIDX;
      prior_jtj_add_lds(c.pJ, c.ldp, c.pn, L_S);
#ifdef AVM_TP
      if (t < c.pn) lds[M_G + pidx[t]] += lds[L_DXP + t] + lds[L_DX2 + t];  // g += J0^T r_p (the two shares of phase A)
#else
      if (t < c.pn) lds[M_G + pidx[t]] += lds[L_DXP + t];  // g += J0^T r_p (left in lds[L_DXP] by phase A)
#endif
    }
    __syncthreads();
    PROF(c, 20);
    // ---- phase F: eliminate the start-0 inverse depths (scalar pivots)
    if (flag == AVM_MARGIN_OLD && nf0 > 0) {
      if (t < MAXE) lds[L_HEE + t] = (t < nf0 && lds[L_HEE + t] > o.marg_eps) ? 1.0 / lds[L_HEE + t] : 0.0;  // 1 / E^T E in place
      __syncthreads();
      marg_schur_phase(nf0);
    }
    __syncthreads();
    PROF(c, 21);
    // ---- phase G: dropped / kept variable lists (ints at I_FSTART.. are dead now)
    int* midx = ids + 0;       // [<=16]
    int* kidx = ids + 16;      // [<=96]
    int* kblk = ids + 120;     // [<=16] id of kept block k : pose f -> f, speedbias f -> 11+f, ex -> 22, td -> 23
    int* cnts = ids + 140;     // m, n, nblk
    __syncthreads();
    if (t == 0) {
      int present = 0;  // bit id
      for (int k = 0; k < c.pnblk; k++) {
        const int kind = ids[I_PBLK + k * 3], fr = ids[I_PBLK + k * 3 + 1];
        present |= 1 << (kind == AVM_BLK_POSE ? fr : (kind == AVM_BLK_SPEEDBIAS ? 11 + fr : (kind == AVM_BLK_TD ? 23 : 22)));
      }
      if (!use_prior) present = 0;
      int m = 0, n = 0, nb = 0;
      if (flag == AVM_MARGIN_OLD) {
        if (imu0) present |= (1 << 0) | (1 << 11) | (1 << 1) | (1 << 12);
        if (nf0 > 0) present |= (1 << 0) | (1 << 22) | (c.est_td ? 1 << 23 : 0);  // ProjectionTdFactor keeps para_Td (estimator.cpp:880-883)
        for (int b = 1; b < NFR; b++)
          if (ids[I_NCOV + b] > 0) present |= 1 << b;
        for (int q = 0; q < 6; q++) midx[m++] = q;
        for (int q = 0; q < 9; q++) midx[m++] = SB0 + q;
        present &= ~((1 << 0) | (1 << 11));
      } else {
        for (int q = 0; q < 6; q++) midx[m++] = 6 * (AVM_WINDOW_SIZE - 1) + q;
        present &= ~(1 << (AVM_WINDOW_SIZE - 1));
      }
      for (int id = 0; id < 24; id++) {
        if (!(present & (1 << id))) continue;
        const int base = id < 11 ? 6 * id : (id < 22 ? SB0 + 9 * (id - 11) : (id == 23 ? MTD : MEX0));
        const int sz = (id >= 11 && id < 22) ? 9 : (id == 23 ? 1 : 6);
        if (n + sz > MAXKEEP || n + sz > PO.max_prior || nb >= MAXPBLK || nb >= PO.max_pblk) {
          atomicMin(err, w);  // (lowest failing window) the host turns this into AVM_ERR_CAPACITY: a truncated kept set would silently lose information
          break;
        }
        kblk[nb++] = id;
        for (int q = 0; q < sz; q++) kidx[n++] = base + q;
      }
      cnts[0] = m, cnts[1] = n, cnts[2] = nb;
    }
    __syncthreads();
    const int m = cnts[0], n = cnts[1], nblk = cnts[2];
    // extract Amm (16x16 at EA), Arm (n x 16 at EB), Arr (n x n), b before the packed matrix is overwritten
    auto Sget = [&](int i, int j) { return lds[L_S + roff(max(i, j)) + min(i, j)]; };
    double* EA = lds + M_WCH;            // Amm 16 x 16, then its eigenvectors next to it
    double* EV = EA + 256;               // 16 x 16
    double* EB = EV + 256;               // Arm : n x 16   (n <= 96 -> 1536)  (M_WCH region holds 1920+; spills into the dead L_G.. vectors)
    // (the 16x16 eigen-solver keeps its rotation records at L_HEE: hee / dxp / rp are dead by now)
    double* BV = lds + L_FR + 198;       // b_m (16), b_r (96): the candidate-state frame slot is unused here
    for (int idx = t; idx < 16 * 16; idx += NT) {
      const int i = idx / 16, j = idx % 16;
      EA[idx] = (i < m && j < m) ? 0.5 * (Sget(midx[i], midx[j]) + Sget(midx[j], midx[i])) : (i == j ? 1.0 : 0.0);
    }
    for (int idx = t; idx < n * 16; idx += NT) {
      const int i = idx / 16, j = idx % 16;
      EB[idx] = j < m ? Sget(kidx[i], midx[j]) : 0.0;
    }
    if (t < 16) BV[t] = t < m ? lds[M_G + midx[t]] : 0.0;
    if (t >= 64 && t < 64 + n) BV[16 + t - 64] = lds[M_G + kidx[t - 64]];
    __syncthreads();
    PROF(c, 22);
    // pseudo-inverse of Amm: Cholesky fast path when provably no eigenvalue is clamped, else the eigen-decomposition
    if (t < 64) {
      const bool fast = pinv16_cholesky(EA, EV, m, o.marg_eps);
      if (t == 0) cnts[3] = fast ? 1 : 0;
    }
    __syncthreads();
    if (!cnts[3]) {
      if (t < 64) jacobi_eig_lds<64>(M_WCH, M_WCH + 256, 16, 16, L_HEE);  // 16 x 16: one wavefront, no block barriers
      __syncthreads();
    }
    PROF(c, 23);
    // Amm^+ = V diag(1/lambda > eps) V^T  -> EA (reuse) ; T = Arm Amm^+ ; A' = Arr - T Amr ; b' = br - T bm
    {
      // (1 / lambda once, by sixteen threads, through LDS - g_e's array is dead since phase F: every thread used to divide sixteen times)
      double* lam_inv = lds + M_GE;
      if (t < 16) lam_inv[t] = (t < m && EA[t * 16 + t] > o.marg_eps) ? 1.0 / EA[t * 16 + t] : 0.0;
      __syncthreads();
      if (t < 256) {
        const int i = t / 16, j = t % 16;
        double sacc = 0;
        for (int k = 0; k < 16; k++) sacc += EV[i * 16 + k] * lam_inv[k] * EV[j * 16 + k];
        EA[t] = (i < m && j < m) ? sacc : 0.0;
      }
      __syncthreads();
    }
    // T = Arm Amm^+ : n x 16, in the LDS range of the solve's gradient / scaling vectors (unused here; it was in the scratch slot:
    // every entry of A' then waited for 16 trips to its memory)
#ifndef AVM_TP
    static_assert(MAXKEEP * 16 <= L_X - L_G, "T fits the dead vectors");
#endif
    static_assert(MAXKEEP <= 96, "T / Arm: 96 rows");
    double* GT = lds + M_GT;
    for (int idx = t; idx < n * 16; idx += NT) {
      const int i = idx / 16, j = idx % 16;
      double sacc = 0;
      for (int k = 0; k < 16; k++) sacc += EB[i * 16 + k] * EA[k * 16 + j];
      GT[idx] = sacc;
    }
    __syncthreads();
    // A' and b' go to the output slots PO.J / PO.r; prior_eig_kernel (prior_eig.hip) turns them into
    // linearized_jacobians / linearized_residuals in place
    {
      double* oJ = PO.J + (size_t)w * PO.max_prior * PO.max_prior;
      double* orr = PO.r + (size_t)w * PO.max_prior;
      // T Amr by 16 x 16 tiles on the matrix cores (K = the 16 dropped columns): lower tiles only - the eigen-solver reads the lower
      // triangle only, as Eigen's SelfAdjointEigenSolver does - dealt to the wavefronts; operands straight from LDS (the scalar form
      // read 32 LDS words per entry: 11 K cycles per window)
      {
        const int lr = lane & 15, lk = lane >> 4, ntl = (n + 15) >> 4;
        for (int tile = wv; tile < ntl * (ntl + 1) / 2; tile += NT / 64) {
          int ti = 0;
          while ((ti + 1) * (ti + 2) / 2 <= tile) ti++;
          const int tj = tile - ti * (ti + 1) / 2;
          const int ra = min(16 * ti + lr, n - 1), rb = min(16 * tj + lr, n - 1);
          d4 D = {0, 0, 0, 0};
#pragma unroll
          for (int mq = 0; mq < 4; mq++) D = __builtin_amdgcn_mfma_f64_16x16x4f64(GT[ra * 16 + lk + 4 * mq], EB[rb * 16 + lk + 4 * mq], D, 0, 0, 0);
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int i = 16 * ti + lk + 4 * r, j = 16 * tj + lr;
            if (i < n && j <= i) {
              const double arr = Sget(kidx[i], kidx[j]), sacc = D[r];
              oJ[(size_t)i * PO.max_prior + j] = arr - sacc;
              // The magnitude the diagonal entry was formed at (|Arr_ii| + |(Arm Amm^+ Amr)_ii|: the bias rows of the kept
              // speed-bias block are differences of two numbers of size 1e10 .. 1e12) goes to the ctx's scale array:
              // prior_eig_kernel's clamp measures an eigenvalue against the rounding noise of ITS variables (prior_eig.hip).
              if (i == j) scale_out[(size_t)w * PO.max_prior + i] = fabs(arr) + fabs(sacc);
            }
          }
        }
      }
      if (t < n) {
        double sacc = 0;
        for (int k = 0; k < 16; k++) sacc += GT[t * 16 + k] * BV[k];
        orr[t] = BV[16 + t] - sacc;
      }
      PROF(c, 24);
      if (t < nblk) {
        const int id = kblk[t];
        const int kind = id < 11 ? AVM_BLK_POSE : (id < 22 ? AVM_BLK_SPEEDBIAS : (id == 23 ? AVM_BLK_TD : AVM_BLK_EXPOSE));
        int fr = id < 11 ? id : (id < 22 ? id - 11 : 0);
        if (kind == AVM_BLK_POSE || kind == AVM_BLK_SPEEDBIAS) {
          if (flag == AVM_MARGIN_OLD)
            fr -= 1;  // addr_shift, estimator.cpp:904-909
          else if (fr == AVM_WINDOW_SIZE)
            fr -= 1;  // estimator.cpp:965-969
        }
        PO.blk_kind[(size_t)w * PO.max_pblk + t] = kind;
        PO.blk_frame[(size_t)w * PO.max_pblk + t] = fr;
        double* x0 = PO.x0 + ((size_t)w * PO.max_pblk + t) * 9;
        const double* src = kind == AVM_BLK_POSE ? lds + L_X + id * 7
                            : (kind == AVM_BLK_SPEEDBIAS ? lds + L_X + XSB + (id - 11) * 9 : lds + L_RIC + (kind == AVM_BLK_TD ? 19 : 12));
        const int gs = kind == AVM_BLK_SPEEDBIAS ? 9 : (kind == AVM_BLK_TD ? 1 : 7);
        for (int q = 0; q < 9; q++) x0[q] = q < gs ? src[q] : 0.0;
      }
      if (t == 0) PO.n[w] = n, PO.nblk[w] = nblk;
    }
    __syncthreads();
    PROF(c, 26);
    if (c.prof && t == 0) c.prof[30] += 1;
  }
}

#ifndef AVM_TP
// Per-factor evaluation at the input state (no solve): parity-test surface for A5/A6/A8.
__global__ __launch_bounds__(NT) void eval_factors_kernel(EvalArgs A) {
  lds_base_check();
  double* lds = LDS();
  int* ids = reinterpret_cast<int*>(lds + L_INT);
  const int t = threadIdx.x;
  const avm_options& o = lds_opt();
  const avm_window_batch& B = A.b;
  const int w = blockIdx.x;
  WinCtx cl;
  cl.sc = nullptr, cl.osf = nullptr, cl.w = w;
  cl.nf = B.n_feat[w];
  cl.obs = as_global(B.obs_xy + (size_t)w * B.max_obs * 2);
  cl.pdelta = as_global(A.pre_delta + (size_t)w * 100), cl.pjac = as_global(A.pre_jac + (size_t)w * 2250), cl.psqrt = as_global(A.pre_sqrt + (size_t)w * 2250);
  cl.psum = as_global(A.pre_sum_dt + (size_t)w * 10);
  cl.lba = as_global(B.imu_lin_ba + (size_t)w * 30), cl.lbg = as_global(B.imu_lin_bg + (size_t)w * 30);
  cl.pn = B.prior_n ? B.prior_n[w] : 0;
  cl.pnblk = cl.pn > 0 ? B.prior_nblk[w] : 0;
  cl.ldp = B.max_prior;
  cl.pJ = as_global(B.prior_J + (size_t)w * B.max_prior * B.max_prior);
  cl.pr = as_global(B.prior_r + (size_t)w * B.max_prior);
  cl.px0 = as_global(B.prior_x0 + (size_t)w * B.max_pblk * 9);
  cl.prof = nullptr, cl.cov = nullptr, cl.nobs_tot = 0;
  lds_store_ctx(cl, A.opt);
  __syncthreads();
  const WinCtx& c = lds_ctx();
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
  build_frames(L_X, 0);
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
    prior_residual_dev(c, L_X);
    if (t < c.pn) {
      acc += 0.5 * lds[L_RP + t] * lds[L_RP + t];
      if (A.prior_res) A.prior_res[(size_t)w * B.max_prior + t] = lds[L_RP + t];
    }
  }
  const double cost = block_sum<NT>(acc, lds + L_RED);
  if (t == 0 && A.cost) A.cost[w] = cost;
}
#endif  // !AVM_TP

#endif  // !AVM_X (marginalization: base and throughput build; per-factor evaluation kernel: base build only)

#ifdef AVM_TP
int window_solve_tp_lds_bytes() { return L_END * 8; }
// the factorization's compile-time tables for the tests (tests/test_tp_pattern.py states them in numpy): out[0..120] = TPP.h, [121..241] = TPP.nz
// (both [k][i]), [242..252] = tp_owner, [253 ..] = tp_perm of the 176 positions (-1: padding)
int window_solve_tp_pattern(int* out) { return tp_pattern_export(out); }
// workgroups of the throughput kernel the runtime says a CU can hold (2 is what the kernel is built for)
int window_solve_tp_occupancy() {
  int n = 0;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(window_solve_tp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, window_solve_tp_kernel, NT, L_END * 8) != hipSuccess) return -1;
  return n;
}

// Throughput form of the solve (window_solve_tp.o): two 256-thread workgroups per CU, a.n_slots = 2 x CUs scratch slots.
hipError_t launch_window_solve_tp(const SolveArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_solve_tp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = a.b.n_windows < a.n_slots ? a.b.n_windows : a.n_slots;
  hipLaunchKernelGGL(window_solve_tp_kernel, dim3(grid), dim3(NT), L_END * 8, stream, a);
  return hipGetLastError();
}

// Throughput form of the marginalization: two 256-thread workgroups per CU, a.n_slots = 2 x CUs scratch slots (the solve's)
hipError_t launch_marginalize_tp(const SolveArgs& a, const avm_prior_out& po, int* err, double* scale, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(marginalize_tp_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = a.b.n_windows < a.n_slots ? a.b.n_windows : a.n_slots;
  hipLaunchKernelGGL(marginalize_tp_kernel, dim3(grid), dim3(NT), L_END * 8, stream, a, po, err, scale);
  return hipGetLastError();
}
#elif !defined(AVM_X)
int window_solve_lds_bytes() { return L_END * 8; }
int window_solve_pattern(int* out) { return tp_pattern_export(out); }

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

hipError_t launch_marginalize(const SolveArgs& a, const avm_prior_out& po, int* err, double* scale, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(marginalize_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = a.b.n_windows < a.n_slots ? a.b.n_windows : a.n_slots;
  hipLaunchKernelGGL(marginalize_kernel, dim3(grid), dim3(NT), L_END * 8, stream, a, po, err, scale);
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
#else
int window_solve_x_lds_bytes() { return L_END * 8; }
int window_solve_x_pattern(int* out) { return tp_pattern_export(out); }

// the solve with ex_pose / td / relo_Pose as (optional) variables: 178 x 178 reduced system
hipError_t launch_window_solve_x(const SolveArgs& a, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(window_solve_x_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, L_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = a.b.n_windows < a.n_slots ? a.b.n_windows : a.n_slots;
  hipLaunchKernelGGL(window_solve_x_kernel, dim3(grid), dim3(NT), L_END * 8, stream, a);
  return hipGetLastError();
}
#endif

}  // namespace avm
