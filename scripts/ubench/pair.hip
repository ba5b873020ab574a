// Two wavefronts on one SIMD (the throughput kernels' situation: one wavefront of each of a CU's two windows): how fast does a LIGHT wavefront
// - a dependent v_fma_f64 chain, or a v_readlane -> v_fma chain like the pivot chains - get through its instructions while its partner streams
// BULK work: v_mfma_f64_16x16x4 (64 cycles of the FP64 pipe per issue), v_mfma_f64_4x4x4 (a quarter of the work per issue), independent v_fma_f64,
// or nothing; with the light wavefront at the same issue priority as the partner and above it (s_setprio).
// One workgroup of 512 threads on one CU: wavefronts w and w + 4 share SIMD w; wavefronts 0-3 are the light ones, 4-7 the bulk ones.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/pair scripts/ubench/pair.hip && gpurun -- scripts/ubench/pair
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
constexpr int NL = 2048;  // light instructions timed (the printed figure is cycles / NL: the fma and integer chains have NL dependent instructions, the
                          // readlane / LDS chains NL / 2 steps of 2 + 1 / 1 + 1 instructions, readfirstlane NL / 3 steps of 2 + 1, the reciprocal NL / 5 of 6)

template <int BULK, int LIGHT, int PRIO>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2))) void k(long long* out, double* sink, volatile int* flag) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  double a = 1.0 + lane * 1e-9, b = 1e-9 * (lane + 1);
  __shared__ int done;
  if (threadIdx.x == 0) done = 0;
  __syncthreads();
  if (wv < 4) {
    if (PRIO) __builtin_amdgcn_s_setprio(2);
    double x = a;
    const long long t0 = clock64();
    if (LIGHT == 0) {
#pragma unroll 16
      for (int i = 0; i < NL; i++) x = fma(x, a, b);  // dependent FMA chain
    } else if (LIGHT == 1) {
#pragma unroll 8
      for (int i = 0; i < NL / 2; i++) {  // v_readlane -> v_fma pairs (two instructions per step)
        const double s = __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((int)(__builtin_bit_cast(unsigned long long, x) >> 32), i & 15) << 32) |
                                                      (unsigned)__builtin_amdgcn_readlane((int)__builtin_bit_cast(unsigned long long, x), i & 15));
        x = fma(s, b, a);
      }
    }
    if (LIGHT == 2) {  // LDS broadcast read -> fma (every lane reads the same word: what the LDS-broadcast form of the pivot chain does)
      __shared__ double bc[64];
      bc[lane] = x;
#pragma unroll 8
      for (int i = 0; i < NL / 2; i++) {
        const double s = *(volatile double*)&bc[i & 15];
        x = fma(s, b, a);
        if ((i & 15) == 15) bc[lane] = x;
      }
    }
    if (LIGHT == 4) {  // v_readfirstlane pair -> fma
#pragma unroll 8
      for (int i = 0; i < NL / 3; i++) {
        const unsigned long long v = __builtin_bit_cast(unsigned long long, x);
        const unsigned hi = __builtin_amdgcn_readfirstlane((int)(v >> 32)), lo = __builtin_amdgcn_readfirstlane((int)v);
        x = fma(__builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo), b, a);
      }
    }
    if (LIGHT == 5) {  // 32-bit integer VALU chain
      int q = lane;
#pragma unroll 16
      for (int i = 0; i < NL; i++) q = q * 3 + i;
      x += q;
    }
    if (LIGHT == 6) {  // v_rcp_f64 + two Newton steps, dependent (the pivot's reciprocal)
#pragma unroll 4
      for (int i = 0; i < NL / 5; i++) {
        double y = __builtin_amdgcn_rcp(x), e = fma(-x, y, 1.0);
        y = fma(y, e, y), e = fma(-x, y, 1.0), y = fma(y, e, y);
        x = y + a;
      }
    }
    const long long t1 = clock64();
    if (lane == 0) out[wv] = t1 - t0;
    sink[threadIdx.x] = x;
    if (lane == 0) atomicAdd(&done, 1);
  } else {
    d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double m0 = 0, m1 = 0, m2 = 0, m3 = 0, f0 = a, f1 = b, f2 = a + 1, f3 = b + 1;
    long long n = 0;
    const long long t0 = clock64();
    while (__hip_atomic_load(&done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < 4) {  // until the four light wavefronts are through
      if (BULK == 1) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
          c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0), c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
          c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0), c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
        n += 16;
      } else if (BULK == 2) {
#pragma unroll
        for (int u = 0; u < 8; u++) {
          m0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m0, 0, 0, 0), m1 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m1, 0, 0, 0);
          m2 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m2, 0, 0, 0), m3 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, m3, 0, 0, 0);
        }
        n += 32;
      } else if (BULK == 3) {
#pragma unroll
        for (int u = 0; u < 16; u++) f0 = fma(f0, a, b), f1 = fma(f1, a, b), f2 = fma(f2, a, b), f3 = fma(f3, a, b);
        n += 64;
      } else {
        __builtin_amdgcn_s_sleep(8);
      }
    }
    const long long t1 = clock64();
    if (lane == 0) out[wv] = t1 - t0, out[8 + wv] = n;
    sink[threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + m0 + m1 + m2 + m3 + f0 + f1 + f2 + f3;
  }
}

template <int BULK, int LIGHT, int PRIO>
void run(const char* what, long long* o, double* s, int* f) {
  for (int r = 0; r < 2; r++) hipLaunchKernelGGL((k<BULK, LIGHT, PRIO>), dim3(1), dim3(512), 0, 0, o, s, f);
  long long h[16];
  (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  double light = 0, bulk_rate = 0;
  for (int w = 0; w < 4; w++) light += (double)h[w] / NL / 4.0, bulk_rate += h[4 + w] > 0 ? (double)h[4 + w] / (double)(h[12 + w] > 0 ? h[12 + w] : 1) / 4.0 : 0.0;
  std::printf("%-58s light: %6.1f cycles per instruction   partner: %6.1f cycles per bulk instruction\n", what, light, BULK ? bulk_rate : 0.0);
}

int main() {
  long long* o;
  double* s;
  int* f;
  (void)hipMalloc(&o, 16 * 8), (void)hipMalloc(&s, 512 * 8), (void)hipMalloc(&f, 64);
  run<0, 0, 0>("fma chain, partner idle", o, s, f);
  run<1, 0, 0>("fma chain beside v_mfma_f64_16x16x4, equal priority", o, s, f);
  run<1, 0, 1>("fma chain beside v_mfma_f64_16x16x4, light at priority 2", o, s, f);
  run<2, 0, 0>("fma chain beside v_mfma_f64_4x4x4, equal priority", o, s, f);
  run<2, 0, 1>("fma chain beside v_mfma_f64_4x4x4, light at priority 2", o, s, f);
  run<3, 0, 0>("fma chain beside independent v_fma_f64, equal priority", o, s, f);
  run<3, 0, 1>("fma chain beside independent v_fma_f64, light at priority 2", o, s, f);
  run<0, 1, 0>("readlane -> fma chain, partner idle", o, s, f);
  run<1, 1, 0>("readlane -> fma chain beside v_mfma_f64_16x16x4, equal", o, s, f);
  run<1, 1, 1>("readlane -> fma chain beside v_mfma_f64_16x16x4, priority 2", o, s, f);
  run<2, 1, 1>("readlane -> fma chain beside v_mfma_f64_4x4x4, priority 2", o, s, f);
  run<3, 1, 1>("readlane -> fma chain beside independent v_fma_f64, priority 2", o, s, f);
  run<0, 2, 0>("LDS broadcast -> fma, partner idle", o, s, f);
  run<1, 2, 0>("LDS broadcast -> fma beside v_mfma_f64_16x16x4, equal", o, s, f);
  run<1, 2, 1>("LDS broadcast -> fma beside v_mfma_f64_16x16x4, priority 2", o, s, f);
  run<0, 4, 0>("readfirstlane x2 -> fma, partner idle", o, s, f);
  run<1, 4, 0>("readfirstlane x2 -> fma beside v_mfma_f64_16x16x4, equal", o, s, f);
  run<0, 5, 0>("int mad chain, partner idle", o, s, f);
  run<1, 5, 0>("int mad chain beside v_mfma_f64_16x16x4, equal", o, s, f);
  run<1, 5, 1>("int mad chain beside v_mfma_f64_16x16x4, priority 2", o, s, f);
  run<0, 6, 0>("rcp + 2 Newton, partner idle", o, s, f);
  run<1, 6, 0>("rcp + 2 Newton beside v_mfma_f64_16x16x4, equal", o, s, f);
  run<1, 6, 1>("rcp + 2 Newton beside v_mfma_f64_16x16x4, priority 2", o, s, f);
  run<1, 0, 0>("(again) fma chain beside v_mfma_f64_16x16x4, equal priority", o, s, f);
  return 0;
}
