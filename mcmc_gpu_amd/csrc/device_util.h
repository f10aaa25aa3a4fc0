// Device helpers shared by the step kernels and the proposal code (gfx950 only): buffer-descriptor loads/stores with
// 32-bit per-lane offsets, DPP reductions, exact division, compensated sums.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsm {
namespace dev {


constexpr int kNT = 1024;
constexpr int kNW = kNT / 64;
constexpr uint32_t kOOB = 0x80000000u;                 // beyond every descriptor's num_records
constexpr uint64_t kNoUpdBits = 0x7FF8DEADBEEF0001ull; // tagged quiet NaN: "update mask not set"

typedef int v2i32 __attribute__((ext_vector_type(2)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;

__device__ __forceinline__ rsrc_t make_rsrc(const void* p, uint32_t bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
template <int AUX>
__device__ __forceinline__ double ld_f64(rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, 0, AUX));
}
template <int AUX>
__device__ __forceinline__ float ld_f32(rsrc_t r, uint32_t off) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, 0, AUX));
}
__device__ __forceinline__ double2 ld_f64x2(rsrc_t r, uint32_t off, uint32_t soff = 0u) {   // soff: uniform byte offset
  const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)off, (int)soff, 0);
  double2 d;
  d.x = __builtin_bit_cast(double, v2i32{v.x, v.y});
  d.y = __builtin_bit_cast(double, v2i32{v.z, v.w});
  return d;
}
template <typename TS> struct StateIO;
template <> struct StateIO<double> {
  static __device__ __forceinline__ double load(rsrc_t r, uint32_t off) { return ld_f64<2>(r, off); }
  static __device__ __forceinline__ void store(rsrc_t r, uint32_t off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i32, v), r, (int)off, 0, 2);
  }
};
template <> struct StateIO<float> {
  static __device__ __forceinline__ double load(rsrc_t r, uint32_t off) { return (double)ld_f32<2>(r, off); }
  static __device__ __forceinline__ void store(rsrc_t r, uint32_t off, double v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)v), r, (int)off, 0, 2);
  }
};

__device__ __forceinline__ uint32_t magic_for(uint32_t d) { return (uint32_t)(0xFFFFFFFFu / d) + 1u; }

// Correctly rounded x / d from y = RN(1/d) (Markstein); see step_kernel.hip and tests/test_gpu_api.py.
__device__ __forceinline__ double exact_div(double x, double d, double y) {
  const double q0 = x * y;
  const double r = __fma_rn(-q0, d, x);
  return __fma_rn(r, y, q0);
}

__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double x) {
  const v2i32 b = __builtin_bit_cast(v2i32, x);
  v2i32 o;
  if constexpr (ROW_MASK == 0xF) {
    // every lane is written and (quad_perm / row_mirror / row_half_mirror) has a source: no "old" value to initialise -- with
    // update_dpp(0, ...) the compiler zeroes the destination pair before each of these moves
    o.x = __builtin_amdgcn_mov_dpp(b.x, CTRL, 0xF, 0xF, true);
    o.y = __builtin_amdgcn_mov_dpp(b.y, CTRL, 0xF, 0xF, true);
  } else {
    o.x = __builtin_amdgcn_update_dpp(0, b.x, CTRL, ROW_MASK, 0xF, false);
    o.y = __builtin_amdgcn_update_dpp(0, b.y, CTRL, ROW_MASK, 0xF, false);
  }
  return __builtin_bit_cast(double, o);
}
// sum over the 16 lanes of a DPP row; every lane of the row ends with the same value
__device__ __forceinline__ double row16_sum(double x) {
  x += dpp_f64<0xB1, 0xF>(x);    // quad_perm [1,0,3,2]
  x += dpp_f64<0x4E, 0xF>(x);    // quad_perm [2,3,0,1]
  x += dpp_f64<0x141, 0xF>(x);   // row_half_mirror
  x += dpp_f64<0x140, 0xF>(x);   // row_mirror
  return x;
}
// sum over the 64 lanes, returned wave-uniform
__device__ __forceinline__ double wave64_sum(double x) {
  x = row16_sum(x);
  x += dpp_f64<0x142, 0xA>(x);   // row_bcast:15 into rows 1 and 3
  x += dpp_f64<0x143, 0xC>(x);   // row_bcast:31 into rows 2 and 3
  const v2i32 b = __builtin_bit_cast(v2i32, x);
  v2i32 o;
  o.x = __builtin_amdgcn_readlane(b.x, 63);
  o.y = __builtin_amdgcn_readlane(b.y, 63);
  return __builtin_bit_cast(double, o);
}


}  // namespace dev
}  // namespace gsm
