// Instruction cost microbenchmarks for gfx950 (single wavefront / several wavefronts of one workgroup).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ inline double readlane_d(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l); hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
#define REP 64
#define BAR() asm volatile("s_nop 0" : "+v"(x), "+v"(y))
#define TIE(v) asm volatile("" : "+v"(v))
__global__ void k(double* out, long long* cyc, int nw_active) {
  const int t = threadIdx.x, wv = t >> 6;
  __shared__ double sm[4096];
  for (int i = t; i < 4096; i += blockDim.x) sm[i] = 1.0 + 1e-3 * i;
  __syncthreads();
  double x = 1.0 + 1e-9 * t, y = 0.5 + 1e-9 * t, acc = 0;
  long long t0, t1;
  if (wv >= nw_active) return;
  // 0: dependent fma chain
  t0 = clock64(); BAR();
#pragma unroll
  for (int i = 0; i < REP; i++) x = fma(x, y, 0.25);
  BAR();
  t1 = clock64(); if (t == 0) cyc[0] = t1 - t0; acc += x;
  // 1: independent fma (8 chains)
  { double a[8]; for (int q = 0; q < 8; q++) a[q] = x + q;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) a[i & 7] = fma(a[i & 7], y, 0.25);
    for (int q = 0; q < 8; q++) TIE(a[q]);
    BAR();
    t1 = clock64(); if (t == 0) cyc[1] = t1 - t0; for (int q = 0; q < 8; q++) acc += a[q]; }
  // 2: readlane pair + fma with SGPR operand, independent accumulators
  { double a[8]; for (int q = 0; q < 8; q++) a[q] = x + q;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) a[i & 7] = fma(-y, readlane_d(x, i & 15), a[i & 7]);
    for (int q = 0; q < 8; q++) TIE(a[q]);
    BAR();
    t1 = clock64(); if (t == 0) cyc[2] = t1 - t0; for (int q = 0; q < 8; q++) acc += a[q]; }
  // 3: rcp dependent chain
  { double r = x;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) r = __builtin_amdgcn_rcp(r);
    TIE(r); BAR();
    t1 = clock64(); if (t == 0) cyc[3] = t1 - t0; acc += r; }
  // 4: rsq dependent chain
  { double r = x;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) r = __builtin_amdgcn_rsq(r);
    TIE(r); BAR();
    t1 = clock64(); if (t == 0) cyc[4] = t1 - t0; acc += r; }
  // 5: dependent MFMA f64 chain
  { d4 D = {0, 0, 0, 0};
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) D = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, D, 0, 0, 0);
    TIE(D); BAR();
    acc += D[0] + D[1] + D[2] + D[3];
    t1 = clock64(); if (t == 0) cyc[5] = t1 - t0; }
  // 6: independent MFMA f64 (4 accumulators)
  { d4 D[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) D[i & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, D[i & 3], 0, 0, 0);
    for (int q = 0; q < 4; q++) TIE(D[q]);
    BAR();
    for (int q = 0; q < 4; q++) acc += D[q][0] + D[q][3];
    t1 = clock64(); if (t == 0) cyc[6] = t1 - t0; }
  // 7: dependent LDS read chain (pointer chase)
  { int idx = t & 63;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) idx = ((int)sm[idx]) + (t & 63);
    TIE(idx); BAR();
    t1 = clock64(); if (t == 0) cyc[7] = t1 - t0; acc += idx; }
  // 8: readlane pair dependent through fma (readlane -> fma -> readlane)
  { double r = x;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) r = fma(r, readlane_d(r, i & 15), 0.25);
    TIE(r); BAR();
    t1 = clock64(); if (t == 0) cyc[8] = t1 - t0; acc += r; }
  // 9: independent v_mul_f64 x 64
  { double a[16]; for (int q = 0; q < 16; q++) a[q] = x + q;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) a[i & 15] = a[i & 15] * y;
    for (int q = 0; q < 16; q++) TIE(a[q]);
    BAR();
    t1 = clock64(); if (t == 0) cyc[9] = t1 - t0; for (int q = 0; q < 16; q++) acc += a[q]; }
  // 10: independent LDS reads b64 (64 reads, stride 1 across lanes)
  { double a = 0;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) a += sm[(t & 63) + 64 * (i & 31)];
    TIE(a); BAR();
    t1 = clock64(); if (t == 0) cyc[10] = t1 - t0; acc += a; }
  // 11: readlane pairs only (independent)
  { int accs = 0;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) accs += __builtin_amdgcn_readlane(__double2loint(x) + i, i & 15);
    asm volatile("" : "+s"(accs)); BAR();
    t1 = clock64(); if (t == 0) cyc[11] = t1 - t0; acc += accs; }
  // 12: rcp + 2 Newton dependent chain (5 ops per step)
  { double r = x;
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP / 4; i++) { double yy = __builtin_amdgcn_rcp(r); double e = fma(-r, yy, 1.0); yy = fma(yy, e, yy); e = fma(-r, yy, 1.0); r = fma(yy, e, yy) + 1.0; }
    TIE(r); BAR();
    t1 = clock64(); if (t == 0) cyc[12] = (t1 - t0) * 4; acc += r; }
  // 13: independent ds_write_b64
  { t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < REP; i++) sm[(t & 63) + 64 * (i & 31)] = x;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); BAR();
    t1 = clock64(); if (t == 0) cyc[13] = t1 - t0; }
  // 14: s_barrier (only meaningful when all 8 waves are active)
  if (nw_active == 8) {
    t0 = clock64(); BAR();
#pragma unroll
    for (int i = 0; i < 16; i++) __syncthreads();
    BAR();
    t1 = clock64(); if (t == 0) cyc[14] = (t1 - t0) * 4; }
  out[t] = acc;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, 8 * 512); hipMalloc(&cyc, 8 * 16);
  const char* names[] = {"dep fma f64", "indep fma f64 (8 chains)", "readlane pair + fma (indep)", "dep rcp f64", "dep rsq f64", "dep mfma f64 16x16x4",
                         "indep mfma f64 (4 acc)", "dep LDS read b64", "readlane->fma->readlane chain", "indep mul f64", "indep LDS read+add", "readlane b32 (indep)", "rcp+2 Newton step (per 5-op step /4)", "indep ds_write_b64", "s_barrier (x4)"};
  for (int nw : {1, 2, 5, 8}) {
    long long h[16] = {0};
    hipMemset(cyc, 0, 8 * 16);
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(512), 0, 0, out, cyc, nw);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, 8 * 16, hipMemcpyDeviceToHost);
    printf("active waves %d (wave 0 timed; clock64 ticks / op, REP=%d)\n", nw, REP);
    for (int i = 0; i < 15; i++) printf("  %-32s %7.2f\n", names[i], (double)h[i] / REP);
  }
  return 0;
}
