// prior_eig.hip — second half of the marginalization (MarginalizationInfo::marginalize(),
// vins_estimator/src/factor/marginalization_factor.cpp:283-301): the symmetric eigen-decomposition of the
// reduced information matrix A' = Arr - Arm Amm^+ Amr and the "square-root" prior
//     linearized_jacobians = diag(sqrt(S))   V^T ,   linearized_residuals = diag(1/sqrt(S)) V^T b' ,
// with S the eigenvalues clamped to 0 below eps (marginalization_factor.cpp:292-299).
//
// marginalize_kernel (window_solve.hip) leaves A' (lower triangle is read) in PO.J[w] and b' in PO.r[w]; this
// kernel overwrites both in place.  One 512-thread workgroup per window, 48 KB of LDS and <= 128 VGPRs so
// TWO workgroups share a CU and fill each other's barrier / LDS latency.
//
// Method: cyclic Jacobi in the odd-even (Brent-Luk) ordering, carried out in POSITION space.  The ne indices sit
// on positions 0..ne-1; an even step rotates the pairs on positions (2k, 2k+1), an odd step those on
// (2k+1, 2k+2), and after its rotation a pair swaps positions, so ne steps visit every pair once.  Rows and
// columns are physically exchanged with the swap, which keeps every access regular:
//   * A (lower triangle, row-major by position, LDS): A <- R^T A R decomposes into independent 2x2 blocks (rows of
//     pair k1, columns of pair k2, k1 >= k2) at fixed addresses - two 16-byte reads + writes per block on even
//     steps, four 8-byte ones on odd steps, consecutive lanes on consecutive addresses;
//   * V^T lives in REGISTERS: lane k of a "V wavefront" holds, for its 19 columns, the rows on positions 2k and
//     2k+1; an even step is lane-local, an odd step moves one row to the neighbouring lane and back with DPP
//     wave shifts.
//   wavefronts 0-3 : A blocks; wavefront 0 then computes the next step's rotations from the pivots
//   wavefronts 4-7 : V^T, half of the columns while the A blocks run, the other half under the rotation
//                    computation (rotation tables are double buffered).  Two barriers per step.
// The eigenpairs come out in position order, which is as good as any: J^T J and J^T r do not depend on it.
// The rotation angle only steers convergence, so it is computed with the hardware rcp/sqrt approximations;
// (c, s) themselves are normalised to full precision (c^2 + s^2 = 1 to 1 ulp keeps V orthogonal and the
// similarity transform exact).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kernels.hpp"

namespace avm {
namespace pe {

constexpr int NT = 512;
constexpr int NMAX = MAXKEEP;         // 76: padded (even) dimension limit; kept sets of this problem have n <= 75
constexpr int NPMAX = NMAX / 2;       // 38 rotation pairs
constexpr int LD = NMAX;              // row stride of A and V^T in LDS (even: 2x2 blocks are 16-byte aligned)
constexpr int AW = 256;               // threads of the A-block wavefronts (0-3)
constexpr int MAXBLK = 3;             // ceil(38*39/2 / 256)
constexpr int VC = 19;                // columns of V^T per V wavefront (4 x 19 = 76)
constexpr int VC1 = 10;               // columns done while the A blocks run; the rest overlaps the rotation computation

// LDS carve (doubles)
constexpr int P_A = 0;                               // A by position [NMAX][LD]; reused for V^T at the end
constexpr int P_ROT = P_A + NMAX * LD;               // 2 x [NPMAX] double2 (c, s)
constexpr int P_B = P_ROT + 2 * NPMAX * 2;           // b' [NMAX]
constexpr int P_EV = P_B + NMAX;                     // eigenvalues [NMAX]
constexpr int P_FLAG = P_EV + NMAX;                  // 2 ints
constexpr int P_END = P_FLAG + 2;

__device__ __forceinline__ double nrm_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  y = y * (1.5 - (0.5 * x) * y * y);
  y = y * (1.5 - (0.5 * x) * y * y);
  return y;
}

// lane i <- lane i+1 / lane i-1 of the wavefront (lanes without a source keep their own value)
__device__ __forceinline__ int shl_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ int shr_i(int v) { return __builtin_amdgcn_update_dpp(v, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ double shl_d(double v) {
  return __hiloint2double(shl_i(__double2hiint(v)), shl_i(__double2loint(v)));
}
// lane i <- v of lane i-1; lane 0 keeps `keep`
__device__ __forceinline__ double shr_into(double keep, double v) {
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(keep), __double2hiint(v), 0x138, 0xf, 0xf, false);
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(keep), __double2loint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void prior_eig_kernel(avm_prior_out PO, int n_windows, double eps,
                                                                                              long long* prof) {
  extern __shared__ char pe_smem[];
  double* lds = reinterpret_cast<double*>(pe_smem);
  double* A = lds + P_A;
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63;
  const int w = blockIdx.x;
  if (w >= n_windows) return;
  const int n = PO.n[w];
  if (n <= 0 || n > NMAX) return;  // n == -1: MARGIN_SECOND_NEW had nothing to drop (the caller keeps the old prior)
  const long long t_start = prof ? (long long)__builtin_readcyclecounter() : 0;
  const int ne = (n + 1) & ~1, np = ne >> 1;
  double* gJ = PO.J + (size_t)w * PO.max_prior * PO.max_prior;
  double* gr = PO.r + (size_t)w * PO.max_prior;
  const int ldj = PO.max_prior;

  // ---- load: lower triangle of A' (pad row/column and the unused upper triangle = 0), b'
  for (int e = t; e < ne * ne; e += NT) {
    const int i = e / ne, j = e - i * ne;
    A[i * LD + j] = (j <= i && i < n) ? gJ[(size_t)i * ldj + j] : 0.0;
  }
  if (t < ne) lds[P_B + t] = t < n ? gr[t] : 0.0;
  int* flag = reinterpret_cast<int*>(lds + P_FLAG);  // [2] "not converged", double buffered over sweeps
  if (t < 2) flag[t] = 0;
  __syncthreads();
  int sweeps = 0;
  // Both roles run the same barrier sequence: per sweep one after the convergence test, one after the first
  // rotation set, then two per step.
  if (wv < 4) {
    // ================= wavefronts 0-3: A in LDS =================
    // static assignment of the 2x2 blocks, idx = k1 (k1 + 1) / 2 + k2 with k1 >= k2 >= 0
    short bk1[MAXBLK], bk2[MAXBLK];
    const int nblk = (np * (np + 1)) >> 1;
#pragma unroll
    for (int u = 0; u < MAXBLK; u++) {
      const int idx = t + u * AW;
      bk1[u] = -1, bk2[u] = 0;
      if (idx < nblk) {
        int k1 = (int)((sqrt(8.0 * idx + 1.0) - 1.0) * 0.5);
        while ((((k1 + 1) * (k1 + 2)) >> 1) <= idx) k1++;
        while (((k1 * (k1 + 1)) >> 1) > idx) k1--;
        bk1[u] = (short)k1, bk2[u] = (short)(idx - ((k1 * (k1 + 1)) >> 1));
      }
    }

    // wavefront 0: rotation of pair k (= lane) for the step with parity `odd` -> table[buf]
    auto make_rotation = [&](int odd, int buf) {
      const int k = lane;
      const bool have = odd ? (k < np - 1) : (k < np);
      double cs = 1.0, sn = 0.0;
      if (have) {
        const int a = 2 * k + odd;  // positions a, a + 1
        const double aaa = A[a * LD + a], abb = A[(a + 1) * LD + a + 1], aab = A[(a + 1) * LD + a];
        // below 1e-17 sqrt(aaa abb) the pivot is under the rounding noise of the diagonal: leave it
        if (aab * aab > 1e-34 * fabs(aaa * abb) && fabs(aab) > 1e-290) {
          const double d = abb - aaa;
          const double h = __builtin_amdgcn_sqrt(d * d + 4.0 * aab * aab);
          const double tt = (d >= 0 ? 2.0 : -2.0) * aab * __builtin_amdgcn_rcp(fabs(d) + h);  // tan of the rotation angle
          cs = nrm_rsqrt(1.0 + tt * tt);
          sn = tt * cs;
        }
      }
      // odd step: positions 0 and ne-1 sit out; table entry np-1 = identity
      if (k < np) reinterpret_cast<double2*>(lds + P_ROT)[buf * NPMAX + k] = double2{cs, sn};
    };

    // even step: block rows (2 k1, 2 k1 + 1) x columns (2 k2, 2 k2 + 1); both pairs swap positions afterwards.
    // All addresses are static per thread.
    int oe[MAXBLK];
#pragma unroll
    for (int u = 0; u < MAXBLK; u++) oe[u] = bk1[u] < 0 ? -1 : 2 * bk1[u] * LD + 2 * bk2[u];
    auto a_even = [&]() {
      const double2* rcs = reinterpret_cast<const double2*>(lds + P_ROT);
      double2 u0[MAXBLK], u1[MAXBLK], r1[MAXBLK], r2[MAXBLK];
#pragma unroll
      for (int u = 0; u < MAXBLK; u++) {
        if (oe[u] < 0) continue;
        r1[u] = rcs[bk1[u]], r2[u] = rcs[bk2[u]];
        u0[u] = *reinterpret_cast<const double2*>(A + oe[u]);
        u1[u] = *reinterpret_cast<const double2*>(A + oe[u] + LD);
      }
#pragma unroll
      for (int u = 0; u < MAXBLK; u++) {
        if (oe[u] < 0) continue;
        const double c1 = r1[u].x, s1 = r1[u].y, c2 = r2[u].x, s2 = r2[u].y;
        const double a00 = u0[u].x, a10 = u1[u].x, a11 = u1[u].y;
        const double a01 = bk1[u] == bk2[u] ? a10 : u0[u].y;  // diagonal block: the upper element is not stored
        const double b00 = c1 * a00 - s1 * a10, b01 = c1 * a01 - s1 * a11;
        const double b10 = s1 * a00 + c1 * a10, b11 = s1 * a01 + c1 * a11;
        const double n00 = c2 * b00 - s2 * b01, n01 = s2 * b00 + c2 * b01;
        const double n10 = c2 * b10 - s2 * b11, n11 = s2 * b10 + c2 * b11;
        *reinterpret_cast<double2*>(A + oe[u]) = double2{n11, n10};
        *reinterpret_cast<double2*>(A + oe[u] + LD) = double2{n01, n00};
      }
    };
    // odd step: block rows (2 k1 + 1, 2 k1 + 2) x columns (2 k2 + 1, 2 k2 + 2) for k1, k2 < np - 1, rows and columns
    // swap afterwards; the items with k1 = np - 1 are the two positions that sit out (ne - 1 and 0): identity row
    // rotation, only the columns swap.  Load addresses li**, store addresses of n11 / n10 / n01 / n00 = so**.
    int li00[MAXBLK], li01[MAXBLK], li10[MAXBLK], li11[MAXBLK], so11[MAXBLK], so10[MAXBLK], so01[MAXBLK], so00[MAXBLK];
#pragma unroll
    for (int u = 0; u < MAXBLK; u++) {
      li00[u] = -1, li01[u] = li10[u] = li11[u] = so11[u] = so10[u] = so01[u] = so00[u] = 0;
      if (bk1[u] < 0 || bk2[u] >= np - 1) continue;
      const int ra = 2 * bk1[u] + 1, rb = (ra + 1 == ne) ? 0 : ra + 1, ca = 2 * bk2[u] + 1, cb = ca + 1;
      // lower-triangle addresses: (r, c) -> [max][min]
      li00[u] = ra * LD + ca;                              // ra >= ca always
      li01[u] = ra >= cb ? ra * LD + cb : cb * LD + ra;    // diagonal block: same element as li10
      li10[u] = rb >= ca ? rb * LD + ca : ca * LD + rb;    // rb = 0 (sitting out): transposed
      li11[u] = rb >= cb ? rb * LD + cb : cb * LD + rb;
      const bool edge = bk1[u] == np - 1;
      so11[u] = edge ? li10[u] : li00[u];
      so10[u] = edge ? li11[u] : li01[u];
      so01[u] = edge ? li00[u] : li10[u];
      so00[u] = edge ? li01[u] : li11[u];
    }
    auto a_odd = [&]() {
      const double2* rcs = reinterpret_cast<const double2*>(lds + P_ROT) + NPMAX;
      double a00[MAXBLK], a01[MAXBLK], a10[MAXBLK], a11[MAXBLK];
      double2 r1[MAXBLK], r2[MAXBLK];
#pragma unroll
      for (int u = 0; u < MAXBLK; u++) {
        if (li00[u] < 0) continue;
        r1[u] = rcs[bk1[u]], r2[u] = rcs[bk2[u]];
        a00[u] = A[li00[u]], a01[u] = A[li01[u]], a10[u] = A[li10[u]], a11[u] = A[li11[u]];
      }
#pragma unroll
      for (int u = 0; u < MAXBLK; u++) {
        if (li00[u] < 0) continue;
        const double c1 = r1[u].x, s1 = r1[u].y, c2 = r2[u].x, s2 = r2[u].y;
        const double b00 = c1 * a00[u] - s1 * a10[u], b01 = c1 * a01[u] - s1 * a11[u];
        const double b10 = s1 * a00[u] + c1 * a10[u], b11 = s1 * a01[u] + c1 * a11[u];
        A[so10[u]] = c2 * b10 - s2 * b11;
        A[so01[u]] = s2 * b00 + c2 * b01;  // (diagonal block: same address as so10, equal up to rounding)
        A[so00[u]] = c2 * b00 - s2 * b01;
        A[so11[u]] = s2 * b10 + c2 * b11;
      }
    };

    for (int sweep = 0; sweep < 20; sweep++) {
      // converged when every |a_pq| <= 1e-15 sqrt(|a_pp a_qq|): the relative criterion keeps the small
      // eigenvalues accurate, which matters for the eps clamp next to eigenvalues of 1e12
      bool bad = false;
      for (int e = t; e < ne * ne; e += AW) {
        const int i = e / ne, j = e - i * ne;
        if (j < i) {
          const double v = A[i * LD + j];
          const double dd = fabs(A[i * LD + i] * A[j * LD + j]);
          bad |= v * v > 1e-30 * dd;
        }
      }
      if (bad) flag[sweep & 1] = 1;
      __syncthreads();
      if (!flag[sweep & 1]) break;
      if (t == 0) flag[(sweep + 1) & 1] = 0;
      sweeps++;
      if (wv == 0) make_rotation(0, 0);
      __syncthreads();
      for (int step = 0; step < ne; step += 2) {  // ne is even: (even, odd) step pairs, tables 0 / 1
        a_even();
        __syncthreads();
        if (wv == 0) make_rotation(1, 1);
        __syncthreads();
        a_odd();
        __syncthreads();
        if (wv == 0 && step + 2 < ne) make_rotation(0, 0);
        __syncthreads();
      }
    }
    if (t < ne) lds[P_EV + t] = A[t * LD + t];
    __syncthreads();
  } else {
    // ================= wavefronts 4-7: V^T in registers =================
    // lane k holds the rows on positions 2k and 2k+1, columns c0 .. c0+18
    const int c0 = (wv - 4) * VC;
    double X[VC], Y[VC];
#pragma unroll
    for (int c = 0; c < VC; c++) X[c] = (2 * lane == c0 + c) ? 1.0 : 0.0, Y[c] = (2 * lane + 1 == c0 + c) ? 1.0 : 0.0;

    auto rot_of = [&](int buf) {
      return lane < np ? reinterpret_cast<const double2*>(lds + P_ROT)[buf * NPMAX + lane] : double2{1.0, 0.0};
    };
    auto rot_odd = [&]() {
      return lane < np - 1 ? reinterpret_cast<const double2*>(lds + P_ROT)[NPMAX + lane] : double2{0.0, 1.0};
    };
    // rows a, b of V^T  ->  (c a - s b, s a + c b), then the two rows swap positions.
    // even step: a, b = this lane's X, Y
    auto v_even = [&](double2 r, int cbeg, int cend) {
      const double c = r.x, s = r.y;
#pragma unroll
      for (int q = 0; q < VC; q++) {
        if (q < cbeg || q >= cend) continue;
        const double x = X[q], y = Y[q];
        X[q] = s * x + c * y;
        Y[q] = c * x - s * y;
      }
    };
    // odd step: a = this lane's Y (position 2k+1), b = the next lane's X (position 2k+2).  Lane np-1 has no
    // partner: it gets (c, s) = (0, 1), which leaves its Y alone (s y + c x_next = y) whatever the shift brought in;
    // lane 0 has no source for the shift back and keeps its X (position 0 sits out).
    auto v_odd = [&](double2 r, int cbeg, int cend) {
      const double c = r.x, s = r.y;
#pragma unroll
      for (int q = 0; q < VC; q++) {
        if (q < cbeg || q >= cend) continue;
        const double y = Y[q], xn = shl_d(X[q]);
        const double ra = c * y - s * xn;
        Y[q] = s * y + c * xn;
        X[q] = shr_into(X[q], ra);  // row a moves to position 2k+2 = X of lane k+1
      }
    };

    for (int sweep = 0; sweep < 20; sweep++) {
      __syncthreads();
      if (!flag[sweep & 1]) break;
      sweeps++;
      __syncthreads();
      for (int step = 0; step < ne; step += 2) {
        // (register-only work: pin it between the barriers it is meant to overlap with)
        const double2 re = rot_of(0);
        v_even(re, 0, VC1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        v_even(re, VC1, VC);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        const double2 ro = rot_odd();
        v_odd(ro, 0, VC1);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        v_odd(ro, VC1, VC);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __syncthreads();  // eigenvalues are out of A: its storage now takes V^T (row = position)
    if (lane < np) {
#pragma unroll
      for (int c = 0; c < VC; c++)
        if (c0 + c < ne) A[2 * lane * LD + c0 + c] = X[c], A[(2 * lane + 1) * LD + c0 + c] = Y[c];
    }
  }
  __syncthreads();
  const double* Vt = A;
  // ---- linearized_jacobians = diag(sqrt(S)) V^T ; linearized_residuals = diag(1/sqrt(S)) V^T b'
  // An odd n carries one pad index (zero row / column of A): it only ever sees identity rotations, so its
  // eigenvector is still exactly the unit vector e_{ne-1} and every other row of V^T has an exact 0 in that column.
  // Its position is skipped: output row k = position k before the pad, k + 1 after it.
  int padpos = ne;  // no pad
  if (ne != n) {
    // the pad eigenvector is the unit vector e_{ne-1}: the position whose V^T row has 1 in column ne-1
    for (int k = 0; k < ne; k++)
      if (Vt[k * LD + ne - 1] != 0.0) padpos = k;
  }
  for (int e = t; e < n * n; e += NT) {
    const int k = e / n, j = e - k * n;
    const int pos = k < padpos ? k : k + 1;
    const double ev = lds[P_EV + pos];
    gJ[(size_t)k * ldj + j] = (ev > eps ? sqrt(ev) : 0.0) * Vt[pos * LD + j];
  }
  if (t < n) {
    const int pos = t < padpos ? t : t + 1;
    const double ev = lds[P_EV + pos];
    double vb = 0;
    for (int j = 0; j < n; j++) vb += Vt[pos * LD + j] * lds[P_B + j];
    gr[t] = (ev > eps ? sqrt(1.0 / ev) : 0.0) * vb;
  }
  if (prof && t == 0) {
    atomicAdd(reinterpret_cast<unsigned long long*>(prof + 25), (unsigned long long)((long long)__builtin_readcyclecounter() - t_start));
    atomicAdd(reinterpret_cast<unsigned long long*>(prof + 29), (unsigned long long)sweeps);
  }
}

}  // namespace pe

hipError_t launch_prior_eig(const avm_prior_out& po, int n_windows, double eps, long long* prof, hipStream_t stream) {
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(pe::prior_eig_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, pe::P_END * 8);
    if (e != hipSuccess) return e;
    attr_set = true;
    if (prof) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(pe::prior_eig_kernel), pe::NT, pe::P_END * 8);
      fprintf(stderr, "[avm] prior_eig_kernel: %d workgroups / CU (LDS %d B)\n", nb, pe::P_END * 8);
    }
  }
  hipLaunchKernelGGL(pe::prior_eig_kernel, dim3(n_windows), dim3(pe::NT), pe::P_END * 8, stream, po, n_windows, eps, prof);
  return hipGetLastError();
}

}  // namespace avm
