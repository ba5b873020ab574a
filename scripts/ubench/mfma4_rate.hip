// v_mfma_f64_4x4x4_4b on gfx950: cycles per instruction, dependent (one accumulator chain) and independent (eight chains), next to
// v_fma_f64 and the 32-bit DPP move the selector's current evaluation is made of.  One wavefront per SIMD-less CU (grid 1 x 64).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mfma4_rate scripts/ubench/mfma4_rate.hip && gpurun -- scripts/ubench/mfma4_rate
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(long long* out, double* sink) {
  const int lane = threadIdx.x;
  double a = 1.0 + lane * 1e-3, b = 0.5 + lane * 1e-4;
  double c0 = 0, c1 = 0, c2 = 0, c3 = 0, c4 = 0, c5 = 0, c6 = 0, c7 = 0;
  constexpr int N = 512;
  long long t0 = clock64();
#pragma unroll 8
  for (int i = 0; i < N; i++) c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
  long long t1 = clock64();
#pragma unroll 2
  for (int i = 0; i < N / 8; i++) {
    c0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c3, 0, 0, 0);
    c4 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4, 0, 0, 0);
    c5 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c5, 0, 0, 0);
    c6 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c6, 0, 0, 0);
    c7 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c7, 0, 0, 0);
  }
  long long t2 = clock64();
  double f0 = a, f1 = b, f2 = a + 1, f3 = b + 1, f4 = a + 2, f5 = b + 2, f6 = a + 3, f7 = b + 3;
#pragma unroll 2
  for (int i = 0; i < N / 8; i++) {
    f0 = fma(f0, a, b), f1 = fma(f1, a, b), f2 = fma(f2, a, b), f3 = fma(f3, a, b);
    f4 = fma(f4, a, b), f5 = fma(f5, a, b), f6 = fma(f6, a, b), f7 = fma(f7, a, b);
  }
  long long t3 = clock64();
#pragma unroll 8
  for (int i = 0; i < N; i++) f0 = fma(f0, a, b);
  long long t4 = clock64();
  // A-operand produced by an MFMA, consumed by the next (the trsm -> syrk hand-over): latency of the dependent pair
#pragma unroll 8
  for (int i = 0; i < N; i++) c1 = __builtin_amdgcn_mfma_f64_4x4x4f64(c1, b, c2, 0, 0, 0);
  long long t5 = clock64();
  if (lane == 0) out[0] = t1 - t0, out[1] = t2 - t1, out[2] = t3 - t2, out[3] = t4 - t3, out[4] = t5 - t4, out[5] = N;
  sink[lane] = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7;
}
int main() {
  long long* o;
  double* s;
  (void)hipMalloc(&o, 64), (void)hipMalloc(&s, 64 * 8);
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s);
  long long h[8];
  (void)hipMemcpy(h, o, 48, hipMemcpyDeviceToHost);
  const double n = (double)h[5];
  std::printf("cycles per instruction (one wavefront): mfma_f64_4x4x4 dependent %.1f, 8 independent chains %.1f | v_fma_f64 8 chains %.1f, dependent %.1f | mfma with its A operand from the previous mfma %.1f\n",
              h[0] / n, h[1] / n, h[2] / n, h[3] / n, h[4] / n);
  return 0;
}
