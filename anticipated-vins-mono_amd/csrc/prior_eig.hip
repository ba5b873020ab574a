// prior_eig.hip — second half of the marginalization (MarginalizationInfo::marginalize(),
// vins_estimator/src/factor/marginalization_factor.cpp:283-301): the symmetric eigen-decomposition of the
// reduced information matrix A' = Arr - Arm Amm^+ Amr and the "square-root" prior
//     linearized_jacobians = diag(sqrt(S))   V^T ,   linearized_residuals = diag(1/sqrt(S)) V^T b' ,
// with S the eigenvalues clamped to 0 below eps (marginalization_factor.cpp:292-299).
//
// marginalize_kernel (window_solve.hip) leaves A' (lower triangle is read) in PO.J[w] and b' in PO.r[w]; this
// kernel overwrites both in place.  One 512-thread workgroup per window; everything it touches lives in
// 72 KB of LDS so TWO workgroups share a CU and hide each other's LDS / barrier latency.
//
// Method: cyclic Jacobi with the round-robin (chess tournament) ordering, ne/2 disjoint rotations per step.
//   * A is kept as a packed lower triangle.  A <- R^T A R decomposes into independent 2x2 blocks
//     (rows of pair k1, columns of pair k2, k1 > k2): four scattered 8-byte reads + writes per block.
//   * the eigenvectors are accumulated transposed (Vt[k][:] = eigenvector k), so V <- V R is two contiguous
//     rows per pair, moved as 16-byte LDS accesses.
//   * waves 0-2 own the A blocks, waves 3-7 the Vt rows; wave 0 computes the next step's rotations (and applies
//     them to the 2x2 pair-diagonal blocks) while the V waves are still finishing the current step: the rotation
//     tables are double buffered.  Two barriers per step.
//   * the rotation angle only steers convergence, so it is computed with the hardware rcp/sqrt approximations;
//     (c, s) themselves are normalised to full precision (c^2 + s^2 = 1 to 1 ulp keeps V orthogonal and the
//     similarity transform exact).
#include <hip/hip_runtime.h>
#include <cstdio>
#include "kernels.hpp"

namespace avm {
namespace pe {

constexpr int NT = 512;
constexpr int NMAX = 76;              // padded (even) dimension limit; kept sets of this problem have n <= 75
constexpr int NPMAX = NMAX / 2;       // 38 rotation pairs
constexpr int AW = 192;               // threads of the A-block waves (waves 0-2)
constexpr int VW = NT - AW;           // threads of the V waves (waves 3-7)
constexpr int MAXBLK = 4;             // ceil(38*37/2 / 192)
constexpr int MAXVU = 5;              // ceil(38*38 / 320)
constexpr int VU_PH1 = 3;             // V units done in phase 1; the rest overlaps the rotation computation

// LDS carve (doubles)
constexpr int P_A = 0;                               // packed lower, NMAX*(NMAX+1)/2 = 2926
constexpr int P_V = 2926;                            // Vt [NMAX][NMAX] = 5776 (16-byte aligned: 2926*8 = 23408)
constexpr int P_ROT = P_V + NMAX * NMAX;             // 2 x [NPMAX] double2 (c, s)
constexpr int P_PQ = P_ROT + 2 * NPMAX * 2;          // 2 x [NPMAX] int2 (p, q)
constexpr int P_B = P_PQ + 2 * NPMAX;                // b' [NMAX]
constexpr int P_RED = P_B + NMAX;                    // [16]
constexpr int P_END = P_RED + 16;

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

__global__ __launch_bounds__(NT) void prior_eig_kernel(avm_prior_out PO, int n_windows, double eps, long long* prof) {
  extern __shared__ char pe_smem[];
  double* lds = reinterpret_cast<double*>(pe_smem);
  double* A = lds + P_A;
  double* Vt = lds + P_V;
  const int t = threadIdx.x, wv = t >> 6;
  const int w = blockIdx.x;
  if (w >= n_windows) return;
  const int n = PO.n[w];
  if (n <= 0 || n > NMAX) return;  // n == -1: MARGIN_SECOND_NEW had nothing to drop (the caller keeps the old prior)
  const long long t_start = prof ? (long long)__builtin_readcyclecounter() : 0;
  const int ne = (n + 1) & ~1, np = ne >> 1, hc = ne >> 1;  // hc: 16-byte chunks per Vt row
  double* gJ = PO.J + (size_t)w * PO.max_prior * PO.max_prior;
  double* gr = PO.r + (size_t)w * PO.max_prior;
  const int ldj = PO.max_prior;

  // ---- load: lower triangle of A' (pad row/column = 0), Vt = I, b'
  for (int e = t; e < ne * ne; e += NT) {
    const int i = e / ne, j = e - i * ne;
    if (j <= i) A[((i * (i + 1)) >> 1) + j] = (i < n) ? gJ[(size_t)i * ldj + j] : 0.0;
    Vt[e] = (i == j) ? 1.0 : 0.0;
  }
  if (t < ne) lds[P_B + t] = t < n ? gr[t] : 0.0;

  // ---- static work assignment
  short bk1[MAXBLK], bk2[MAXBLK];
  const int nblk = (np * (np - 1)) >> 1;
#pragma unroll
  for (int u = 0; u < MAXBLK; u++) {
    const int idx = t + u * AW;
    bk1[u] = -1, bk2[u] = 0;
    if (t < AW && idx < nblk) {
      // idx = k1 (k1 - 1) / 2 + k2 , k1 > k2 >= 0
      int k1 = (int)((sqrt(8.0 * idx + 1.0) + 1.0) * 0.5);
      while (((k1 * (k1 + 1)) >> 1) <= idx) k1++;
      while (((k1 * (k1 - 1)) >> 1) > idx) k1--;
      bk1[u] = (short)k1, bk2[u] = (short)(idx - ((k1 * (k1 - 1)) >> 1));
    }
  }
  short vk[MAXVU], vj[MAXVU];
  const int nvu = np * hc;
#pragma unroll
  for (int u = 0; u < MAXVU; u++) {
    const int idx = (t - AW) + u * VW;
    vk[u] = -1, vj[u] = 0;
    if (t >= AW && idx < nvu) vk[u] = (short)(idx / hc), vj[u] = (short)(idx % hc);
  }
  __syncthreads();

  // rotation of pair k at step `step` -> tables[buf]; also rotates the pair's own 2x2 diagonal block in place
  auto make_rotation = [&](int step, int buf) {
    const int k = t;  // t < np
    const int a = k == 0 ? ne - 1 : (step + k) % (ne - 1);
    const int b = k == 0 ? step : (step - k + (ne - 1)) % (ne - 1);
    const int p = min(a, b), q = max(a, b);
    const int ipp = ((p * (p + 1)) >> 1) + p, iqq = ((q * (q + 1)) >> 1) + q, iqp = ((q * (q + 1)) >> 1) + p;
    const double app = A[ipp], aqq = A[iqq], apq = A[iqp];
    double cs = 1.0, sn = 0.0;
    // below 1e-17 sqrt(app aqq) the pivot is under the rounding noise of the diagonal: leave it
    if (apq * apq > 1e-34 * fabs(app * aqq) && fabs(apq) > 1e-290) {
      const double d = aqq - app;
      const double h = __builtin_amdgcn_sqrt(d * d + 4.0 * apq * apq);
      const double tt = (d >= 0 ? 2.0 : -2.0) * apq * __builtin_amdgcn_rcp(fabs(d) + h);  // tan of the rotation angle
      cs = nrm_rsqrt(1.0 + tt * tt);
      sn = tt * cs;
      const double cc = cs * cs, ss = sn * sn, sc = cs * sn;
      A[ipp] = cc * app - 2.0 * sc * apq + ss * aqq;
      A[iqq] = ss * app + 2.0 * sc * apq + cc * aqq;
      A[iqp] = (cc - ss) * apq + sc * (app - aqq);
    }
    reinterpret_cast<double2*>(lds + P_ROT)[buf * NPMAX + k] = double2{cs, sn};
    reinterpret_cast<int2*>(lds + P_PQ)[buf * NPMAX + k] = int2{p, q};
  };

  auto v_units = [&](int buf, int u0, int u1) {
    const double2* rcs = reinterpret_cast<const double2*>(lds + P_ROT) + buf * NPMAX;
    const int2* rpq = reinterpret_cast<const int2*>(lds + P_PQ) + buf * NPMAX;
    double2 x[MAXVU], y[MAXVU], r[MAXVU];
    int ox[MAXVU], oy[MAXVU];
#pragma unroll
    for (int u = 0; u < MAXVU; u++) {
      if (u < u0 || u >= u1) continue;
      r[u] = double2{1.0, 0.0};
      if (vk[u] < 0) continue;
      r[u] = rcs[vk[u]];
      if (r[u].y == 0.0) continue;
      const int2 pq = rpq[vk[u]];
      ox[u] = pq.x * ne + 2 * vj[u], oy[u] = pq.y * ne + 2 * vj[u];
      x[u] = *reinterpret_cast<const double2*>(Vt + ox[u]);
      y[u] = *reinterpret_cast<const double2*>(Vt + oy[u]);
    }
#pragma unroll
    for (int u = 0; u < MAXVU; u++) {
      if (u < u0 || u >= u1) continue;
      if (r[u].y == 0.0) continue;
      const double c = r[u].x, s = r[u].y;
      *reinterpret_cast<double2*>(Vt + ox[u]) = double2{c * x[u].x - s * y[u].x, c * x[u].y - s * y[u].y};
      *reinterpret_cast<double2*>(Vt + oy[u]) = double2{s * x[u].x + c * y[u].x, s * x[u].y + c * y[u].y};
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
      act[u] = true;
      const int2 pq1 = rpq[bk1[u]], pq2 = rpq[bk2[u]];
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

  int sweeps = 0;
  for (int sweep = 0; sweep < 20; sweep++) {
    // converged when every |a_pq| <= 1e-15 sqrt(|a_pp a_qq|): the relative criterion keeps the small eigenvalues
    // accurate, which matters for the eps clamp next to eigenvalues of 1e12
    int bad = 0;
    for (int e = t; e < ne * ne; e += NT) {
      const int i = e / ne, j = e - i * ne;
      if (j < i) {
        const double v = A[((i * (i + 1)) >> 1) + j];
        const double dd = fabs(A[((i * (i + 1)) >> 1) + i] * A[((j * (j + 1)) >> 1) + j]);
        bad |= (v * v > 1e-30 * dd) ? 1 : 0;
      }
    }
    const int any = __syncthreads_or(bad);
    if (!any) break;
    sweeps++;
    if (wv == 0 && t < np) make_rotation(0, 0);
    __syncthreads();
    for (int step = 0; step < ne - 1; step++) {
      const int buf = step & 1;
      if (t < AW)
        a_blocks(buf);
      else
        v_units(buf, 0, VU_PH1);
      __syncthreads();
      if (t >= AW)
        v_units(buf, VU_PH1, MAXVU);
      else if (t < np && step + 1 < ne - 1)
        make_rotation(step + 1, buf ^ 1);
      __syncthreads();
    }
  }

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
