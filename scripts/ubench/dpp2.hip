// v_fmac_f64_dpp / v_mov_b64_dpp row_newbcast on gfx950, measured again (round 6): throughput over eight independent accumulators, with the two wait
// states a DPP read needs behind a VALU write of the same register (the compiler's hazard recognizer does not look into inline assembly: scripts/ubench/dpp.hip
// of round 3 issued the DPP instruction right behind the write of its source and read stale lanes - "does not accumulate").
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/dpp2 scripts/ubench/dpp2.hip && gpurun -- scripts/ubench/dpp2
#include <hip/hip_runtime.h>
#include <cstdio>

#define FMAC_DPP(acc, src, mul, K) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul))
#define MOV_DPP(dst, src, K) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:" #K " row_mask:0xf bank_mask:0xf" : "=v"(dst) : "v"(src))
__device__ inline double bcast32(double v) {
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x157, 0xf, 0xf, true);
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x157, 0xf, 0xf, true);
  return __hiloint2double(hi, lo);
}
__global__ void sem(double* out) {
  double a = threadIdx.x, acc = 0.5, m = 2.0;
  asm volatile("s_nop 1");
  FMAC_DPP(acc, a, m, 3);
  out[threadIdx.x] = acc;  // expect 0.5 + 2 * ((lane & ~15) + 3)
}
__global__ void cost(double* out, long long* cyc) {
  double a[8], src = 1.0 + 1e-9 * threadIdx.x, m = 1e-3, t[8];
  for (int q = 0; q < 8; q++) a[q] = q + threadIdx.x;
  asm volatile("s_nop 1");
  long long t0 = clock64();
#pragma unroll
  for (int i = 0; i < 32; i++) {
    FMAC_DPP(a[0], src, m, 7); FMAC_DPP(a[1], src, m, 7); FMAC_DPP(a[2], src, m, 7); FMAC_DPP(a[3], src, m, 7);
    FMAC_DPP(a[4], src, m, 7); FMAC_DPP(a[5], src, m, 7); FMAC_DPP(a[6], src, m, 7); FMAC_DPP(a[7], src, m, 7);
  }
  long long t1 = clock64();
#pragma unroll
  for (int i = 0; i < 32; i++) {
#pragma unroll
    for (int q = 0; q < 8; q++) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(a[q]) : "v"(src), "v"(m));
  }
  long long t2 = clock64();
#pragma unroll
  for (int i = 0; i < 32; i++) {
    MOV_DPP(t[0], src, 7); MOV_DPP(t[1], src, 7); MOV_DPP(t[2], src, 7); MOV_DPP(t[3], src, 7);
    MOV_DPP(t[4], src, 7); MOV_DPP(t[5], src, 7); MOV_DPP(t[6], src, 7); MOV_DPP(t[7], src, 7);
  }
  long long t3 = clock64();
  double u[8];
#pragma unroll
  for (int i = 0; i < 32; i++) {
#pragma unroll
    for (int q = 0; q < 8; q++) u[q] = bcast32(src + q), a[q] = fma(u[q], m, a[q]);
  }
  long long t4 = clock64();
  double r = 0;
  for (int q = 0; q < 8; q++) r += a[q] + t[q];
  out[threadIdx.x] = r;
  if (threadIdx.x == 0) cyc[0] = t1 - t0, cyc[1] = t2 - t1, cyc[2] = t3 - t2, cyc[3] = t4 - t3;
}
int main() {
  double* d;
  long long* c;
  (void)hipMalloc(&d, 256 * 8), (void)hipMalloc(&c, 64);
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, d);
  double h[64];
  (void)hipMemcpy(h, d, 64 * 8, hipMemcpyDeviceToHost);
  int ok = 1;
  for (int i = 0; i < 64; i++) ok &= h[i] == 0.5 + 2.0 * ((i & ~15) + 3);
  for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(cost, dim3(1), dim3(64), 0, 0, d, c);
  long long hc[4];
  (void)hipMemcpy(hc, c, 32, hipMemcpyDeviceToHost);
  std::printf("{\"fmac_f64_dpp_accumulates\": %d, \"per_instruction_cycles\": {\"v_fmac_f64_dpp\": %.2f, \"v_fmac_f64\": %.2f, \"v_mov_b64_dpp\": %.2f, \"two_v_mov_b32_dpp_plus_v_fma_f64 (3 instructions)\": %.2f}}\n",
              ok, hc[0] / 256.0, hc[1] / 256.0, hc[2] / 256.0, hc[3] / 256.0);
  return 0;
}
