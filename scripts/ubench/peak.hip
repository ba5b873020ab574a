// FP64 peak of one MI355X, measured: the denominator of bench.py's roofline.frac.
//   (1) v_mfma_f64_16x16x4_f64   independent accumulator chains, every CU full of wavefronts
//   (2) v_fma_f64                independent chains
// Prints one JSON line: TFLOP/s of both, the clock the runtime reports, and the datasheet value they are compared with
// (78.6 TFLOP/s FP64 vector = FP64 matrix).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/peak scripts/ubench/peak.hip && gpurun -- scripts/ubench/peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int CHAINS = 8;   // independent accumulators per wavefront
constexpr int INNER = 64;   // unrolled MFMAs / FMAs per chain per outer iteration

__global__ __launch_bounds__(256) void mfma_peak(double* out, int outer) {
  d4 acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = d4{0.0, 0.0, 0.0, 0.0};
  double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
  for (int o = 0; o < outer; o++) {
#pragma unroll
    for (int i = 0; i < INNER; i++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[c], 0, 0, 0);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void fma_peak(double* out, int outer) {
  double acc[CHAINS];
#pragma unroll
  for (int c = 0; c < CHAINS; c++) acc[c] = 1e-3 * (c + threadIdx.x);
  const double a = 1.0 - 1e-9, b = 1e-12;
  for (int o = 0; o < outer; o++) {
#pragma unroll
    for (int i = 0; i < INNER; i++)
#pragma unroll
      for (int c = 0; c < CHAINS; c++) acc[c] = fma(acc[c], a, b);
  }
  double s = 0;
#pragma unroll
  for (int c = 0; c < CHAINS; c++) s += acc[c];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
static double time_ms(K kern, dim3 grid, dim3 block, double* out, int outer) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, grid, block, 0, 0, out, 8);  // warm-up
  (void)hipDeviceSynchronize();
  float best = 1e30f;
  for (int rep = 0; rep < 5; rep++) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kern, grid, block, 0, 0, out, outer);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    best = ms < best ? ms : best;
  }
  return best;
}

int main() {
  hipDeviceProp_t p;
  if (hipGetDeviceProperties(&p, 0) != hipSuccess) {
    std::printf("{\"error\": \"no HIP device\"}\n");
    return 1;
  }
  const int cus = p.multiProcessorCount;
  const int waves_per_cu = 16;  // 4 per SIMD
  dim3 block(256), grid(cus * waves_per_cu / 4);
  double* out = nullptr;
  (void)hipMalloc(&out, sizeof(double) * grid.x * block.x);
  const int outer = 400;
  const double n_waves = (double)grid.x * 4;
  const double ms_m = time_ms(mfma_peak, grid, block, out, outer);
  const double fl_m = n_waves * (double)outer * INNER * CHAINS * (2.0 * 16 * 16 * 4);
  const double ms_f = time_ms(fma_peak, grid, block, out, outer);
  const double fl_f = n_waves * (double)outer * INNER * CHAINS * (2.0 * 64);
  std::printf(
      "{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"mfma_f64_16x16x4_tflops\": %.2f, \"mfma_ms\": %.3f, "
      "\"fma_f64_tflops\": %.2f, \"fma_ms\": %.3f, \"datasheet_fp64_tflops\": 78.6, "
      "\"mfma_cycles_per_instr_per_simd\": %.1f}\n",
      p.gcnArchName, cus, p.clockRate / 1000, fl_m / (ms_m * 1e-3) / 1e12, ms_m, fl_f / (ms_f * 1e-3) / 1e12, ms_f,
      // SIMD-cycles available / MFMA instructions issued per SIMD
      (ms_m * 1e-3 * (p.clockRate * 1e3)) / ((double)outer * INNER * CHAINS * (waves_per_cu / 4)));
  (void)hipFree(out);
  return 0;
}
