// prior_eig.hip — second half of the marginalization (MarginalizationInfo::marginalize(),
// vins_estimator/src/factor/marginalization_factor.cpp:283-301): the symmetric eigen-decomposition of the
// reduced information matrix A' = Arr - Arm Amm^+ Amr and the "square-root" prior
//     linearized_jacobians = diag(sqrt(S))   V^T ,   linearized_residuals = diag(1/sqrt(S)) V^T b' ,
// with S the eigenvalues clamped to 0 below eps (marginalization_factor.cpp:292-299).
//
// marginalize_kernel (window_solve.hip) leaves A' (lower triangle is read) in PO.J[w] and b' in PO.r[w]; this
// kernel overwrites both in place.  One 512-thread workgroup per window, 70 KB of LDS and <= 128 VGPRs so
// TWO workgroups share a CU and fill each other's barrier / LDS latency.
//
// Method: cyclic Jacobi in the odd-even (Brent-Luk) ordering.  The ne indices sit on positions 0..ne-1; an even
// step rotates the pairs on positions (2k, 2k+1), an odd step those on (2k+1, 2k+2), and after its rotation a pair
// swaps positions, so ne steps visit every pair once.  That ordering only ever pairs neighbours, which lets
// the eigenvector matrix live in REGISTERS: lane k of a "V wavefront" holds, for its 19 columns, the two rows
// of V^T on positions 2k and 2k+1; an even step is lane-local, an odd step moves one row to the neighbouring
// lane and back with DPP wave shifts.  Only A (packed lower triangle) stays in LDS, where A <- R^T A R
// decomposes into independent 2x2 blocks (rows of pair k1, columns of pair k2, k1 > k2).
//   wavefronts 0-3 : A blocks (four scattered 8-byte reads + writes per block); wavefront 0 then computes the
//                    next step's rotations and applies them to the pairs' own 2x2 diagonal blocks
//   wavefronts 4-7 : V^T, half of the columns while the A blocks run, the other half under the rotation
//                    computation (rotation tables are double buffered).  Two barriers per step.
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
constexpr int AW = 256;               // threads of the A-block wavefronts (0-3)
constexpr int MAXBLK = 3;             // ceil(38*37/2 / 256)
constexpr int VC = 19;                // columns of V^T per V wavefront (4 x 19 = 76)
constexpr int VC1 = 10;               // columns done while the A blocks run; the rest overlaps the rotation computation

// LDS carve (doubles)
constexpr int P_A = 0;                               // packed lower, NMAX*(NMAX+1)/2 = 2926
constexpr int P_V = 2926;                            // V^T [NMAX][NMAX] = 5776, written once at the end (coalesced output)
constexpr int P_ROT = P_V + NMAX * NMAX;             // 2 x [NPMAX] double2 (c, s)
constexpr int P_PQ = P_ROT + 2 * NPMAX * 2;          // 2 x [NPMAX] int2 (a, b): indices of the rotated pair
constexpr int P_B = P_PQ + 2 * NPMAX;                // b' [NMAX]
constexpr int P_FLAG = P_B + NMAX;                   // 2 ints
constexpr int P_END = P_FLAG + 2;

__device__ __forceinline__ int tri(int i, int j) {  // packed lower index of (max, min)
  const int a = max(i, j), b = min(i, j);
  return ((a * (a + 1)) >> 1) + b;
}

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
__device__ __forceinline__ double shr_d(double v) {
  return __hiloint2double(shr_i(__double2hiint(v)), shr_i(__double2loint(v)));
}

__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(4, 4))) void prior_eig_kernel(avm_prior_out PO, int n_windows, double eps, long long* prof) {
  extern __shared__ char pe_smem[];
  double* lds = reinterpret_cast<double*>(pe_smem);
  double* A = lds + P_A;
  double* Vt = lds + P_V;
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

  // ---- load: lower triangle of A' (pad row/column = 0), b'
  for (int e = t; e < ne * ne; e += NT) {
    const int i = e / ne, j = e - i * ne;
    if (j <= i) A[((i * (i + 1)) >> 1) + j] = (i < n) ? gJ[(size_t)i * ldj + j] : 0.0;
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
    // static assignment of the 2x2 blocks (k1 > k2)
    short bk1[MAXBLK], bk2[MAXBLK];
    const int nblk = (np * (np - 1)) >> 1;
#pragma unroll
    for (int u = 0; u < MAXBLK; u++) {
      const int idx = t + u * AW;
      bk1[u] = -1, bk2[u] = 0;
      if (idx < nblk) {
        // idx = k1 (k1 - 1) / 2 + k2 , k1 > k2 >= 0
        int k1 = (int)((sqrt(8.0 * idx + 1.0) + 1.0) * 0.5);
        while (((k1 * (k1 + 1)) >> 1) <= idx) k1++;
        while (((k1 * (k1 - 1)) >> 1) > idx) k1--;
        bk1[u] = (short)k1, bk2[u] = (short)(idx - ((k1 * (k1 - 1)) >> 1));
      }
    }
    int pa = 2 * lane, pb = 2 * lane + 1;  // index on positions 2k / 2k+1 (used by wavefront 0 only)

    // wavefront 0: rotations of the step with parity `odd` -> tables[buf]; rotates the pairs' own 2x2 diagonal
    // blocks of A in place and advances the position -> index map (pairs swap places after their rotation)
    auto make_rotation = [&](int odd, int buf) {
      const int k = lane;
      const int pan = shl_i(pa);                           // index on position 2k+2
      const int pa0 = __builtin_amdgcn_readfirstlane(pa);  // index on position 0
      // odd step: positions 0 and ne-1 sit out; they form a pseudo pair with the identity rotation so that their
      // rows / columns of A still see the other pairs' rotations through the 2x2 block scheme
      const int a = odd ? pb : pa, b = odd ? (k == np - 1 ? pa0 : pan) : pb;
      const bool have = odd ? (k < np - 1) : (k < np);
      double cs = 1.0, sn = 0.0;
      if (have) {
        const int iaa = ((a * (a + 1)) >> 1) + a, ibb = ((b * (b + 1)) >> 1) + b, iab = tri(a, b);
        const double aaa = A[iaa], abb = A[ibb], aab = A[iab];
        // below 1e-17 sqrt(aaa abb) the pivot is under the rounding noise of the diagonal: leave it
        if (aab * aab > 1e-34 * fabs(aaa * abb) && fabs(aab) > 1e-290) {
          const double d = abb - aaa;
          const double h = __builtin_amdgcn_sqrt(d * d + 4.0 * aab * aab);
          const double tt = (d >= 0 ? 2.0 : -2.0) * aab * __builtin_amdgcn_rcp(fabs(d) + h);  // tan of the rotation angle
          cs = nrm_rsqrt(1.0 + tt * tt);
          sn = tt * cs;
          const double cc = cs * cs, ss = sn * sn, sc = cs * sn;
          A[iaa] = cc * aaa - 2.0 * sc * aab + ss * abb;
          A[ibb] = ss * aaa + 2.0 * sc * aab + cc * abb;
          A[iab] = (cc - ss) * aab + sc * (aaa - abb);
        }
      }
      if (k < np) {
        reinterpret_cast<double2*>(lds + P_ROT)[buf * NPMAX + k] = double2{cs, sn};
        reinterpret_cast<int2*>(lds + P_PQ)[buf * NPMAX + k] = int2{a, b};
      }
      if (odd) {
        const int pbs = shr_i(pb);  // index on position 2k-1
        if (k >= 1 && k < np) pa = pbs;
        if (k < np - 1) pb = pan;
      } else {
        const int tmp = pa;
        pa = pb, pb = tmp;
      }
    };

    auto a_blocks = [&](int buf) {
      const double2* rcs = reinterpret_cast<const double2*>(lds + P_ROT) + buf * NPMAX;
      const int2* rpq = reinterpret_cast<const int2*>(lds + P_PQ) + buf * NPMAX;
      double a00[MAXBLK], a01[MAXBLK], a10[MAXBLK], a11[MAXBLK];
      double2 r1[MAXBLK], r2[MAXBLK];
      int i00[MAXBLK], i01[MAXBLK], i10[MAXBLK], i11[MAXBLK];
      bool act[MAXBLK];
#pragma unroll
      for (int u = 0; u < MAXBLK; u++) {
        act[u] = false;
        if (bk1[u] < 0) continue;
        r1[u] = rcs[bk1[u]], r2[u] = rcs[bk2[u]];
        if (r1[u].y == 0.0 && r2[u].y == 0.0) continue;
        const int2 pq1 = rpq[bk1[u]], pq2 = rpq[bk2[u]];
        act[u] = true;
        i00[u] = tri(pq1.x, pq2.x), i01[u] = tri(pq1.x, pq2.y), i10[u] = tri(pq1.y, pq2.x), i11[u] = tri(pq1.y, pq2.y);
        a00[u] = A[i00[u]], a01[u] = A[i01[u]], a10[u] = A[i10[u]], a11[u] = A[i11[u]];
      }
#pragma unroll
      for (int u = 0; u < MAXBLK; u++) {
        if (!act[u]) continue;
        const double c1 = r1[u].x, s1 = r1[u].y, c2 = r2[u].x, s2 = r2[u].y;
        const double b00 = c1 * a00[u] - s1 * a10[u], b01 = c1 * a01[u] - s1 * a11[u];
        const double b10 = s1 * a00[u] + c1 * a10[u], b11 = s1 * a01[u] + c1 * a11[u];
        A[i00[u]] = c2 * b00 - s2 * b01;
        A[i01[u]] = s2 * b00 + c2 * b01;
        A[i10[u]] = c2 * b10 - s2 * b11;
        A[i11[u]] = s2 * b10 + c2 * b11;
      }
    };

    for (int sweep = 0; sweep < 20; sweep++) {
      // converged when every |a_pq| <= 1e-15 sqrt(|a_pp a_qq|): the relative criterion keeps the small
      // eigenvalues accurate, which matters for the eps clamp next to eigenvalues of 1e12
      bool bad = false;
      for (int e = t; e < ne * ne; e += AW) {
        const int i = e / ne, j = e - i * ne;
        if (j < i) {
          const double v = A[((i * (i + 1)) >> 1) + j];
          const double dd = fabs(A[((i * (i + 1)) >> 1) + i] * A[((j * (j + 1)) >> 1) + j]);
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
        a_blocks(0);
        __syncthreads();
        if (wv == 0) make_rotation(1, 1);
        __syncthreads();
        a_blocks(1);
        __syncthreads();
        if (wv == 0 && step + 2 < ne) make_rotation(0, 0);
        __syncthreads();
      }
    }
    if (wv == 0 && lane < np) reinterpret_cast<int2*>(lds + P_PQ)[lane] = int2{pa, pb};
    __syncthreads();
  } else {
    // ================= wavefronts 4-7: V^T in registers =================
    // lane k holds the rows on positions 2k and 2k+1, columns c0 .. c0+18
    const int c0 = (wv - 4) * VC;
    double X[VC], Y[VC];
#pragma unroll
    for (int c = 0; c < VC; c++) X[c] = (2 * lane == c0 + c) ? 1.0 : 0.0, Y[c] = (2 * lane + 1 == c0 + c) ? 1.0 : 0.0;

    // rows a, b of V^T  ->  (c a - s b, s a + c b), then the two rows swap positions.
    // even step: a, b = this lane's X, Y
    auto rot_of = [&](int buf) {
      return lane < np ? reinterpret_cast<const double2*>(lds + P_ROT)[buf * NPMAX + lane] : double2{1.0, 0.0};
    };
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
    // odd step: a = this lane's Y (position 2k+1), b = the next lane's X (position 2k+2)
    auto v_odd = [&](double2 r, int cbeg, int cend) {
      const double c = r.x, s = r.y;
      const bool have = lane < np - 1, recv = lane >= 1 && lane < np;
#pragma unroll
      for (int q = 0; q < VC; q++) {
        if (q < cbeg || q >= cend) continue;
        const double y = Y[q], xn = shl_d(X[q]);
        const double ra = c * y - s * xn, rb = s * y + c * xn;
        Y[q] = have ? rb : y;
        const double down = shr_d(ra);  // row a moves to position 2k+2 = X of lane k+1
        X[q] = recv ? down : X[q];
      }
    };

    for (int sweep = 0; sweep < 20; sweep++) {
      __syncthreads();
      if (!flag[sweep & 1]) break;
      sweeps++;
      __syncthreads();
      for (int step = 0; step < ne; step += 2) {  // ne is even: (even, odd) step pairs, tables 0 / 1
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
        const double2 ro = rot_of(1);
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
    __syncthreads();
    // V^T to LDS by index (row a of V^T = eigenvector of eigenvalue A[a][a])
    if (lane < np) {
      const int2 ab = reinterpret_cast<const int2*>(lds + P_PQ)[lane];
#pragma unroll
      for (int c = 0; c < VC; c++)
        if (c0 + c < ne) Vt[ab.x * ne + c0 + c] = X[c], Vt[ab.y * ne + c0 + c] = Y[c];
    }
  }
  __syncthreads();
  // ---- linearized_jacobians = diag(sqrt(S)) V^T ; linearized_residuals = diag(1/sqrt(S)) V^T b'
  for (int e = t; e < n * n; e += NT) {
    const int k = e / n, j = e - k * n;
    const double ev = A[((k * (k + 1)) >> 1) + k];
    gJ[(size_t)k * ldj + j] = (ev > eps ? sqrt(ev) : 0.0) * Vt[k * ne + j];
  }
  if (t < n) {
    const double ev = A[((t * (t + 1)) >> 1) + t];
    double vb = 0;
    for (int j = 0; j < n; j++) vb += Vt[t * ne + j] * lds[P_B + j];
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
