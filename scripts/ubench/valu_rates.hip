// Micro-benchmark: issue cost of the vector instructions the fused chain kernel is made of, on gfx950.
// One workgroup of 256 / 1024 threads (1 / 4 waves per SIMD) on one CU; each wave runs `iters` x 64 independent copies of
// one instruction (8 accumulators round-robin); cycles by s_memtime (constant 100 MHz clock -> converted with
// s_memrealtime is not needed: we report the ratio to v_fma_f64).
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/valu_rates scripts/ubench/valu_rates.hip && scripts/ubench/valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
#include <string>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int OP>
__global__ void k(uint64_t* out, int iters, double seed) {
  double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7;
  double b = seed * 0.5 + 1.0, c = 1e-9;
  uint32_t i0 = threadIdx.x * 2654435761u + 1, i1 = i0 + 7, i2 = i0 + 11, i3 = i0 + 13, m = 0xD2511F53u;
  uint64_t l0 = i0, l1 = i1, l2 = i2, l3 = i3;
  int s0 = 0;
  __syncthreads();
  const uint64_t t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (OP == 0) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 1) { REP8(asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 2) { REP8(asm volatile("v_add_f64 %0, %0, %8\n v_add_f64 %1, %1, %8\n v_add_f64 %2, %2, %8\n v_add_f64 %3, %3, %8\n v_add_f64 %4, %4, %8\n v_add_f64 %5, %5, %8\n v_add_f64 %6, %6, %8\n v_add_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 3) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %6, %1\n v_mad_u64_u32 %2, vcc, %4, %7, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n v_mad_u64_u32 %0, vcc, %4, %6, %0\n v_mad_u64_u32 %1, vcc, %4, %7, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %6, %3" : "+v"(l0), "+v"(l1), "+v"(l2), "+v"(l3) : "v"(m), "v"(i0), "v"(i1), "v"(i2) : "vcc");) }
    if (OP == 4) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 5) { REP8(asm volatile("v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4\n v_mul_hi_u32 %0, %0, %4\n v_mul_hi_u32 %1, %1, %4\n v_mul_hi_u32 %2, %2, %4\n v_mul_hi_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 6) { REP8(asm volatile("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 7) { REP8(asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 8) { REP8(asm volatile("v_rsq_f64 %0, %0\n v_rsq_f64 %1, %1\n v_rsq_f64 %2, %2\n v_rsq_f64 %3, %3\n v_rsq_f64 %4, %4\n v_rsq_f64 %5, %5\n v_rsq_f64 %6, %6\n v_rsq_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 9) { REP8(asm volatile("v_readlane_b32 %0, %1, 5\n v_readlane_b32 %0, %2, 6\n v_readlane_b32 %0, %3, 7\n v_readlane_b32 %0, %4, 8\n v_readlane_b32 %0, %1, 9\n v_readlane_b32 %0, %2, 10\n v_readlane_b32 %0, %3, 11\n v_readlane_b32 %0, %4, 12" : "+s"(s0) : "v"(i0), "v"(i1), "v"(i2), "v"(i3));) }
    if (OP == 10) { REP8(asm volatile("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m) : "vcc");) }
    if (OP == 11) { REP8(asm volatile("v_cvt_f64_u32 %0, %8\n v_cvt_f64_u32 %1, %8\n v_cvt_f64_u32 %2, %8\n v_cvt_f64_u32 %3, %8\n v_cvt_f64_u32 %4, %8\n v_cvt_f64_u32 %5, %8\n v_cvt_f64_u32 %6, %8\n v_cvt_f64_u32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(i0));) }
    if (OP == 12) { REP8(asm volatile("v_ldexp_f64 %0, %0, %8\n v_ldexp_f64 %1, %1, %8\n v_ldexp_f64 %2, %2, %8\n v_ldexp_f64 %3, %3, %8\n v_ldexp_f64 %4, %4, %8\n v_ldexp_f64 %5, %5, %8\n v_ldexp_f64 %6, %6, %8\n v_ldexp_f64 %7, %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(s0));) }
    if (OP == 13) { REP8(asm volatile("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "s"(s0));) }
    if (OP == 14) { REP8(asm volatile("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 15) { REP8(asm volatile("v_sqrt_f64 %0, %0\n v_sqrt_f64 %1, %1\n v_sqrt_f64 %2, %2\n v_sqrt_f64 %3, %3\n v_sqrt_f64 %4, %4\n v_sqrt_f64 %5, %5\n v_sqrt_f64 %6, %6\n v_sqrt_f64 %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
    if (OP == 16) { REP8(asm volatile("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 17) { REP8(asm volatile("v_mov_b64 %0, %8\n v_mov_b64 %1, %8\n v_mov_b64 %2, %8\n v_mov_b64 %3, %8\n v_mov_b64 %4, %8\n v_mov_b64 %5, %8\n v_mov_b64 %6, %8\n v_mov_b64 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));) }
    if (OP == 18) { REP8(asm volatile("v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4\n v_lshl_add_u32 %0, %0, 3, %4\n v_lshl_add_u32 %1, %1, 3, %4\n v_lshl_add_u32 %2, %2, 3, %4\n v_lshl_add_u32 %3, %3, 3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }

    if (OP == 19) { REP8(asm volatile("v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc\n v_cndmask_b32 %0, %4, %5, vcc\n v_cndmask_b32 %1, %4, %5, vcc\n v_cndmask_b32 %2, %4, %5, vcc\n v_cndmask_b32 %3, %4, %5, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m), "v"(threadIdx.x) : "vcc");) }
    if (OP == 20) { REP8(asm volatile("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m) : "s20", "s21");) }
    if (OP == 21) { REP8(asm volatile("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 22) { REP8(asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %4\n v_cmp_gt_u32 vcc, %2, %4\n v_cmp_gt_u32 vcc, %3, %4\n v_cmp_gt_u32 vcc, %0, %4\n v_cmp_gt_u32 vcc, %1, %4\n v_cmp_gt_u32 vcc, %2, %4\n v_cmp_gt_u32 vcc, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m) : "vcc");) }
    if (OP == 23) { REP8(asm volatile("v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc\n v_cmp_gt_u32 vcc, %0, %4\n v_cndmask_b32 %1, %1, %4, vcc\n v_cmp_gt_u32 vcc, %2, %4\n v_cndmask_b32 %3, %3, %4, vcc" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m) : "vcc");) }
    if (OP == 24) { REP8(asm volatile("v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %0, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 25) { REP8(asm volatile("v_fma_f64 %0, %0, %8, s[20:21]\n v_fma_f64 %1, %1, %8, s[20:21]\n v_fma_f64 %2, %2, %8, s[20:21]\n v_fma_f64 %3, %3, %8, s[20:21]\n v_fma_f64 %4, %4, %8, s[20:21]\n v_fma_f64 %5, %5, %8, s[20:21]\n v_fma_f64 %6, %6, %8, s[20:21]\n v_fma_f64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b) : "s20", "s21");) }
    if (OP == 26) { REP8(asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %0, %0, %8, %9" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));) }
    if (OP == 27) { REP8(asm volatile("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : "v"(m));) }
    if (OP == 28) { REP8(asm volatile("v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s20, 4\n v_writelane_b32 %2, s20, 5\n v_writelane_b32 %3, s20, 6\n v_writelane_b32 %0, s20, 7\n v_writelane_b32 %1, s20, 8\n v_writelane_b32 %2, s20, 9\n v_writelane_b32 %3, s20, 10" : "+v"(i0), "+v"(i1), "+v"(i2), "+v"(i3) : : "s20");) }
    if (OP == 29) { REP8(asm volatile("v_cmp_o_f64 vcc, %0, %0\n v_cmp_o_f64 vcc, %1, %1\n v_cmp_o_f64 vcc, %2, %2\n v_cmp_o_f64 vcc, %3, %3\n v_cmp_o_f64 vcc, %4, %4\n v_cmp_o_f64 vcc, %5, %5\n v_cmp_o_f64 vcc, %6, %6\n v_cmp_o_f64 vcc, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");) }
  }
  const uint64_t t1 = __builtin_amdgcn_s_memtime();
  if ((threadIdx.x & 63) == 0) out[threadIdx.x >> 6] = t1 - t0;
  if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 1.2345 || i0 + i1 + i2 + i3 == 77 || l0 + l1 + l2 + l3 == 99 || s0 == 1234567) out[100] = 1;
}

template <int OP>
double run(int threads, uint64_t* d, int iters) {
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, iters, 1.000001);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(k<OP>, dim3(1), dim3(threads), 0, 0, d, iters, 1.000001);
  hipDeviceSynchronize();
  std::vector<uint64_t> h(16);
  hipMemcpy(h.data(), d, 16 * 8, hipMemcpyDeviceToHost);
  uint64_t mx = 0;
  for (int w = 0; w < threads / 64; ++w) mx = mx > h[w] ? mx : h[w];
  return (double)mx;
}

int main() {
  uint64_t* d;
  hipMalloc(&d, 128 * 8);
  const int iters = 200;
  const char* names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_mad_u64_u32", "v_mul_lo_u32", "v_mul_hi_u32", "v_xor_b32", "v_rcp_f64",
                         "v_rsq_f64", "v_readlane_b32", "v_cndmask_b32", "v_cvt_f64_u32", "v_ldexp_f64", "v_mov_b32(s)", "v_mul_u32_u24",
                         "v_sqrt_f64", "v_fma_f32", "v_mov_b64", "v_lshl_add_u32", "cndmask indep dst", "cndmask_e64 sgpr", "v_add_u32", "v_cmp_gt_u32", "cmp+cndmask", "v_mov_b32_dpp", "fma_f64 sgpr src", "fma_f64 dep chain", "v_and_b32", "v_writelane", "v_cmp_o_f64"};
  double t1[30], t4[30];
#define RUN(i) t1[i] = run<i>(256, d, iters); t4[i] = run<i>(1024, d, iters);
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12) RUN(13) RUN(14) RUN(15) RUN(16) RUN(17) RUN(18) RUN(19) RUN(20) RUN(21) RUN(22) RUN(23) RUN(24) RUN(25) RUN(26) RUN(27) RUN(28) RUN(29)
  // s_memtime counts at 100 MHz: report relative to v_fma_f64 at 4 waves per SIMD
  printf("%-16s %14s %14s %10s\n", "instruction", "ticks 1w/SIMD", "ticks 4w/SIMD", "vs fma64");
  for (int i = 0; i < 30; ++i) printf("%-16s %14.0f %14.0f %10.2f\n", names[i], t1[i], t4[i], t4[i] / t4[0]);
  printf("(each wave issues %d instructions; 4w/SIMD = 1024 threads on one CU)\n", iters * 64);
  return 0;
}
