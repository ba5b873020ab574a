// (ROUND 6: superseded by dpp2.hip.  This probe issued v_fmac_f64_dpp right behind the VALU write of its source - a DPP read needs two wait states there,
//  and the hazard recognizer does not look into inline assembly -, read stale lanes and concluded "does not accumulate"; its v_mov_b64_dpp timing had the
//  same flaw.  With the wait states v_fmac_f64_dpp accumulates and issues in 5.8 cycles.)
// v_fmac_f64_dpp / v_mov_b64_dpp with row_newbcast on gfx950: semantics and cost (the selector's round kernel broadcasts a
// lane's value to its 16-lane row with it: four 15-row matrices per wavefront).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/dpp scripts/ubench/dpp.hip && gpurun -- scripts/ubench/dpp
#include <hip/hip_runtime.h>

#include <cstdio>

template <int K>
__device__ inline void fmac_bcast(double& acc, double src, double mul) {  // acc += bcast_K(src) * mul
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(K));
}
template <int K>
__device__ inline double bcast(double v) {
  double r;
  asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(v), "n"(K));
  return r;
}
template <int K>
__device__ inline double bcast32(double v) {  // two 32-bit DPP moves (row_newbcast:K = dpp_ctrl 0x150 + K)
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + K, 0xf, 0xf, true);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + K, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void sem(double* out) {
  double a = threadIdx.x, acc = 0.5, m = 2.0;
  fmac_bcast<3>(acc, a, m);
  out[threadIdx.x] = acc;
  out[64 + threadIdx.x] = bcast<5>(a);
  out[128 + threadIdx.x] = bcast32<5>(a);
}
__global__ void cost(double* out, long long* cyc) {
  double a[8], s = 1.0 + 1e-9 * threadIdx.x, m = 1e-3;
  for (int q = 0; q < 8; q++) a[q] = q + threadIdx.x;
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) a[i & 7] = fma(bcast<7>(a[(i + 3) & 7]), m, a[i & 7]);   // v_mov_b64_dpp + v_fma_f64
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) a[i & 7] = fma(s, m, a[i & 7]);
  long long t2 = clock64();
#pragma unroll
  for (int i = 0; i < 64; i++) a[i & 7] = fma(bcast32<7>(a[(i + 3) & 7]), m, a[i & 7]);   // 2 x v_mov_b32_dpp + v_fma_f64
  long long t3 = clock64();
  double r = 0;
  for (int q = 0; q < 8; q++) r += a[q];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = t2 - t1, cyc[2] = t3 - t2;
}
int main() {
  double* d;
  long long* c;
  (void)hipMalloc(&d, 256 * 8), (void)hipMalloc(&c, 32);
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d);
  double h[192];
  (void)hipMemcpy(h, d, 192 * 8, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; i++) ok &= h[64 + i] == (i & ~15) + 5 && h[128 + i] == (i & ~15) + 5;  // (v_fmac_f64_dpp assembles but does not accumulate on gfx950: not used)
  hipLaunchKernelGGL(cost, dim3(1), dim3(64), 0, 0, d, c);
  long long hc[3];
  (void)hipMemcpy(hc, c, 24, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; i += 5) std::printf("lane %d: fmac %g  mov %g\n", i, h[i], h[64 + i]);
  std::printf("{\"row_newbcast_semantics_ok\": %d, \"mov_b64_dpp_plus_fma_ticks_per_64\": %lld, \"fma_f64_ticks_per_64\": %lld, \"two_mov_b32_dpp_plus_fma_ticks_per_64\": %lld}\n", ok, hc[0], hc[1], hc[2]);
  return 0;
}
