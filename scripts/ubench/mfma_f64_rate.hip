// Micro-benchmark: sustained v_mfma_f64_16x16x4_f64 rate of the whole chip (gfx950), 1 / 2 / 4 waves per SIMD, 8 independent
// accumulator tiles per wave, operands in registers (no memory traffic): the ceiling a dense fp64 GEMM inner loop can reach.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/mfma_f64_rate scripts/ubench/mfma_f64_rate.hip && scripts/ubench/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4f64 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  v4f64 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4f64{seed, seed, seed, seed};
  double a = seed + threadIdx.x * 1e-9, b = seed * 0.5;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 1.2345) out[0] = s;
}
int main() {
  double* d; hipMalloc(&d, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wgs_per_cu = 1; wgs_per_cu <= 4; wgs_per_cu *= 2) {
    const int grid = 256 * wgs_per_cu;
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, 100, 1.0);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, d, iters, 1.0);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double flops = (double)grid * 4 /*waves*/ * iters * 32.0 * (16.0 * 16 * 4 * 2);
    printf("%d waves/SIMD: %.2f ms, %.1f TFLOP/s (fp64 MFMA 16x16x4, register operands)\n", wgs_per_cu, ms, flops / ms / 1e9);
  }
  return 0;
}
