// v_mfma_f64_4x4x4_4b on gfx950: operand layout (the selector's evaluation could run its rank-4 tile updates on it: four
// independent 4x4x4 products per instruction).  Measured layout: block b = (lane / 4) % 4, the k index is the 16-lane ROW:
//   A[b][i][k] at lane 16 k + 4 b + i,   B[b][k][j] at lane 16 k + 4 b + j,   D[b][i][j] at lane 16 i + 4 b + j
// - a block is a quad COLUMN of the wavefront, not a 16-lane row; a tile kept in the D layout is directly a B operand and, read
// as an A operand, its transpose.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mfma4 scripts/ubench/mfma4.hip && gpurun -- scripts/ubench/mfma4
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void sem(int* out) {  // out[la * 16 + lb] = lane of block 0 that receives 1 when A is one-hot at la and B at lb (-1 none, -2 several)
  const int lane = threadIdx.x;
  for (int la = 0; la < 16; la++)
    for (int lb = 0; lb < 16; lb++) {
      const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
      const unsigned long long m = __ballot(d != 0.0);
      if (lane == 0) out[la * 16 + lb] = m == 0 ? -1 : (__popcll(m) == 1 ? __ffsll((long long)m) - 1 : -2);
    }
}
int main() {
  int* o;
  (void)hipMalloc(&o, 256 * 4);
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, o);
  int h[256];
  (void)hipMemcpy(h, o, sizeof(h), hipMemcpyDeviceToHost);
  std::printf("rows: A one-hot lane la; columns: B one-hot lane lb; entry: the D lane of block 0 that becomes 1 (. = none)\n");
  for (int la = 0; la < 16; la++) {
    for (int lb = 0; lb < 16; lb++) h[la * 16 + lb] < 0 ? std::printf("  .") : std::printf(" %2d", h[la * 16 + lb]);
    std::printf("\n");
  }
  return 0;
}
