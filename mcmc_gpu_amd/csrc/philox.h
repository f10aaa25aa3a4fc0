// Philox4x32-10 counter-based RNG (Salmon, Moraes, Dror, Shaw: "Parallel random numbers: as easy as
// 1, 2, 3", SC'11) for host and gfx950 device code, plus the uniform / Box-Muller conversions the
// proposal generator uses.  Known-answer vectors of the Random123 distribution are checked in
// tests/test_philox.py through gsm_philox_selftest().
//
// Counter layout used by the sampler (key = 64-bit chain seed):
//   ctr = { draw index, stream id, absolute step (low 32), absolute step (high 32) }
// so a (chain, step, stream, draw) tuple never repeats and a run split into segments reproduces the
// unsplit run.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define GSM_HD __host__ __device__ __forceinline__
#else
#define GSM_HD inline
#endif

namespace gsm {

struct u32x4 { uint32_t x, y, z, w; };

GSM_HD void philox_mulhilo(uint32_t a, uint32_t b, uint32_t& hi, uint32_t& lo) {
  const uint64_t p = (uint64_t)a * (uint64_t)b;
  hi = (uint32_t)(p >> 32);
  lo = (uint32_t)p;
}

GSM_HD u32x4 philox4x32_10(u32x4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    uint32_t hi0, lo0, hi1, lo1;
    philox_mulhilo(0xD2511F53u, c.x, hi0, lo0);
    philox_mulhilo(0xCD9E8D57u, c.z, hi1, lo1);
    u32x4 n;
    n.x = hi1 ^ c.y ^ k0;
    n.y = lo1;
    n.z = hi0 ^ c.w ^ k1;
    n.w = lo0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}

enum : uint32_t { kStreamScalars = 0, kStreamSpectrum = 1, kStreamNugget = 2, kStreamCholesky = 3 };

GSM_HD u32x4 philox_draw(uint64_t seed, int64_t step, uint32_t stream, uint32_t idx) {
  u32x4 c;
  c.x = idx;
  c.y = stream;
  c.z = (uint32_t)((uint64_t)step & 0xFFFFFFFFu);
  c.w = (uint32_t)((uint64_t)step >> 32);
  return philox4x32_10(c, (uint32_t)(seed & 0xFFFFFFFFu), (uint32_t)(seed >> 32));
}

// [0, 1): 53 random bits, the NumPy Generator convention ((u64 >> 11) * 2^-53)
GSM_HD double u01_from(uint32_t lo, uint32_t hi) {
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)(v >> 11) * (1.0 / 9007199254740992.0);
}
// (0, 1]: safe argument for log()
GSM_HD double u01_open0_from(uint32_t lo, uint32_t hi) {
  const uint64_t v = ((uint64_t)hi << 32) | lo;
  return (double)((v >> 11) + 1) * (1.0 / 9007199254740992.0);
}

// one 32-bit word per uniform (the spectrum's normals: four per Philox block): (0, 1] for the logarithm, [0, 1) for the angle
GSM_HD double u01_open0_from32(uint32_t w) { return ((double)w + 1.0) * (1.0 / 4294967296.0); }
GSM_HD double u01_from32(uint32_t w) { return (double)w * (1.0 / 4294967296.0); }

}  // namespace gsm
