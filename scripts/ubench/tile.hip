// Where does a Cholesky trailing tile spend its time?  Pieces of chol_fused_macro on a packed-triangular LDS matrix.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef double dv2 __attribute__((ext_vector_type(2)));
__device__ inline int roff(int i) { const int q = i >> 1; return 2 * __mul24(q + 1, q + (i & 1)); }
#define TIE4(v) asm volatile("" : "+v"(v))
#define REP 16
__global__ __launch_bounds__(512) void k(double* out, long long* cyc, int nw_active) {
  extern __shared__ double sm[];
  const int t = threadIdx.x, wv = t >> 6, lane = t & 63, lr = lane & 15, lk = lane >> 4;
  for (int i = t; i < 14000; i += blockDim.x) sm[i] = 1.0 + 1e-3 * (i % 97);
  __syncthreads();
  if (wv >= nw_active) return;
  double acc = 0;
  long long t0, t1;
  const int c0 = 16;
  // 0: eight ds_read_b128 of four row blocks (rows 16*ti+lr, cols c0+4lk), as in the solves
  { dv2 a[8];
    t0 = clock64();
    for (int rep = 0; rep < REP; rep++) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const dv2* pa = reinterpret_cast<const dv2*>(sm + roff(16 * (4 + q + (rep & 1)) + lr) + c0 + 4 * lk);
        a[2 * q] = pa[0], a[2 * q + 1] = pa[1];
      }
#pragma unroll
      for (int q = 0; q < 8; q++) { TIE4(a[q]); acc += a[q][0]; }
    }
    t1 = clock64(); if (t == 0) cyc[0] = t1 - t0; }
  // 1: 16 MFMAs in 8 chains, then adds, then 16 MFMAs in 4 chains (register only)
  { d4 Xa[4], Xb[4], X[4], D[4]; double b0 = 1.0 + lane, b1 = 0.5 + lane;
    t0 = clock64();
    for (int rep = 0; rep < REP; rep++) {
#pragma unroll
      for (int q = 0; q < 4; q++) Xa[q] = d4{0, 0, 0, 0}, Xb[q] = d4{0, 0, 0, 0}, D[q] = d4{0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 4; q++) { Xa[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b1, Xa[q], 0, 0, 0); Xb[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b0, Xb[q], 0, 0, 0); }
#pragma unroll
      for (int q = 0; q < 4; q++) { Xa[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(b1, b1, Xa[q], 0, 0, 0); Xb[q] = __builtin_amdgcn_mfma_f64_16x16x4f64(b0, b0, Xb[q], 0, 0, 0); }
#pragma unroll
      for (int q = 0; q < 4; q++) X[q] = Xa[q] + Xb[q];
#pragma unroll
      for (int m = 0; m < 4; m++) {
        D[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0][m], X[2][m], D[0], 0, 0, 0);
        D[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1][m], X[2][m], D[1], 0, 0, 0);
        D[2] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[1][m], X[3][m], D[2], 0, 0, 0);
        D[3] = __builtin_amdgcn_mfma_f64_16x16x4f64(X[0][m], X[3][m], D[3], 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) { TIE4(D[q]); b0 += D[q][0] * 1e-30; }
    }
    t1 = clock64(); if (t == 0) cyc[1] = t1 - t0; acc += b0; }
  // 2: destination read-modify-write of four tiles (16 ds_read_b64 + 16 ds_write_b64, rows lk+4r, cols lr)
  { t0 = clock64();
    for (int rep = 0; rep < REP; rep++) {
      double d[16]; int o[16];
#pragma unroll
      for (int q = 0; q < 4; q++)
#pragma unroll
        for (int r = 0; r < 4; r++) { o[4 * q + r] = roff(16 * (5 + (q >> 1) + (rep & 1)) + lk + 4 * r) + 16 * (2 + (q & 1)) + lr; d[4 * q + r] = sm[o[4 * q + r]]; }
#pragma unroll
      for (int q = 0; q < 16; q++) sm[o[q]] = d[q] - 1e-9;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    t1 = clock64(); if (t == 0) cyc[2] = t1 - t0; }
  out[t] = acc;
}
int main() {
  double* out; long long* cyc;
  (void)hipMalloc(&out, 8 * 512); (void)hipMalloc(&cyc, 8 * 16);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 14000 * 8);
  const char* names[] = {"8 x ds_read_b128 (4 row blocks of the packed triangle)", "32 MFMA f64 as in the macro tile (8 chains, add, 4 chains)", "dest RMW of 4 tiles (16 ds_read_b64 + 16 ds_write_b64)"};
  for (int nw : {1, 8}) {
    long long h[16] = {0};
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(512), 14000 * 8, 0, out, cyc, nw);
    (void)hipDeviceSynchronize();
    (void)hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost);
    printf("active waves %d (wave 0 timed; ticks per macro-tile-equivalent)\n", nw);
    for (int i = 0; i < 3; i++) printf("  %-60s %8.1f\n", names[i], (double)h[i] / REP);
  }
  return 0;
}
