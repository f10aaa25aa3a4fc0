// NumPy's random-number machinery on gfx950, bit for bit: what the reference's chains draw from numpy.random.Generator(PCG64)
// (gstatsMCMC/MCMC.py:483-492, :1056-1066; per-step call sequence :755, :199-207, :242, :251 on RandField.rng and :1254-1258,
// :1336 on the chain's generator).  NumPy is a third-party dependency of the reference; its published algorithms are restated
// here and in oracle/pcg64_oracle.py, which tests/test_pcg64_oracle.py pins against NumPy itself:
//   PCG64                128-bit LCG (multiplier 0x2360ed051fc65da4_4385df649fccf645), output XSL-RR 128/64; next_uint32 hands out
//                        the two halves of one 64-bit draw, the second one cached in the generator state (has_uint32, uinteger)
//   random / uniform     (next_uint64 >> 11) * 2^-53;  low + (high - low) * next_double
//   integers(low, high)  Lemire's bounded rejection on 32-bit words (range < 2^32 - 1)
//   normal               loc + scale * z, z from the 256-layer ziggurat (tables: ziggurat_tables.h), libm's log1p in the tail
//                        (its VALUE is returned: log1p_fdlibm below is glibc's algorithm operation for operation) and exp in the
//                        wedge test (a comparison only)
// One wavefront serves ONE generator stream: the LCG jumps ahead -- state_{n} = A_n * state_0 + inc * G_n with A_n = mult^n,
// G_n = 1 + mult + ... + mult^(n-1) -- so lane l decodes the draw l + 1 positions ahead and 64 normals come out of one pass
// whenever all 64 take the ziggurat's fast path (99.3 % of the draws do); a slow-path draw is resolved by the whole wave in
// uniform code and the stream is re-based behind it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gsm {
namespace pcg {

struct u128 { uint64_t lo, hi; };
#define GSM_PCG_HD __host__ __device__ inline
GSM_PCG_HD uint64_t mulhi64(uint64_t a, uint64_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __umul64hi(a, b);
#else
  return (uint64_t)(((unsigned __int128)a * b) >> 64);
#endif
}
GSM_PCG_HD u128 mul128(u128 a, u128 b) {      // low 128 bits
  u128 r;
  r.lo = a.lo * b.lo;
  r.hi = mulhi64(a.lo, b.lo) + a.lo * b.hi + a.hi * b.lo;
  return r;
}
GSM_PCG_HD u128 add128(u128 a, u128 b) {
  u128 r;
  r.lo = a.lo + b.lo;
  r.hi = a.hi + b.hi + (r.lo < a.lo ? 1u : 0u);
  return r;
}
constexpr uint64_t kMultHi = 0x2360ed051fc65da4ull, kMultLo = 0x4385df649fccf645ull;
GSM_PCG_HD uint64_t output_xsl_rr(u128 s) {
  const uint64_t x = s.hi ^ s.lo;
  const unsigned rot = (unsigned)(s.hi >> 58);
  return (x >> rot) | (x << ((64u - rot) & 63u));
}
GSM_PCG_HD double to_double(uint64_t r) { return (double)(r >> 11) * (1.0 / 9007199254740992.0); }

constexpr int kJump = 128;                  // jump table entries: (A_n, G_n) for n = 1 .. 128 at index n - 1
// host: fills tab[4 * (n - 1) ..] = A_n.lo, A_n.hi, G_n.lo, G_n.hi
inline void build_jump_table(uint64_t* tab) {
  u128 A{1, 0}, G{0, 0};
  const u128 M{kMultLo, kMultHi};
  for (int n = 1; n <= kJump; ++n) {
    G = add128(mul128(G, M), u128{1, 0});     // G_n = G_{n-1} * mult + 1
    A = mul128(A, M);
    tab[4 * (n - 1)] = A.lo; tab[4 * (n - 1) + 1] = A.hi; tab[4 * (n - 1) + 2] = G.lo; tab[4 * (n - 1) + 3] = G.hi;
  }
}

// glibc's log1p for double (sysdeps/ieee754/dbl-64/s_log1p.c: the fdlibm algorithm), operation for operation; inputs here are
// x = -u, u in [0, 1).  Restated and checked against libm in oracle/pcg64_oracle.py / tests/test_pcg64_oracle.py.
__device__ __forceinline__ double log1p_fdlibm(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01, Lp3 = 2.857142874366239149e-01,
               Lp4 = 2.222219843214978396e-01, Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  const int32_t hx = (int32_t)(__builtin_bit_cast(uint64_t, x) >> 32);
  const int32_t ax = hx & 0x7fffffff;
  int32_t k = 1, hu = 0;
  double f = 0.0, c = 0.0;
  if (hx < 0x3FDA827A) {
    if (ax >= 0x3ff00000) return (x == -1.0) ? -INFINITY : NAN;
    if (ax < 0x3e200000) {
      if (ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= (int32_t)0xbfd2bec3) { k = 0; f = x; hu = 1; }
  }
  if (hx >= 0x7ff00000) return x + x;
  if (k != 0) {
    double u;
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = (int32_t)(__builtin_bit_cast(uint64_t, u) >> 32);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      c /= u;
    } else {
      u = x;
      hu = (int32_t)(__builtin_bit_cast(uint64_t, u) >> 32);
      k = (hu >> 20) - 1023;
      c = 0.0;
    }
    hu &= 0x000fffff;
    const uint64_t ub = __builtin_bit_cast(uint64_t, u);
    if (hu < 0x6a09e) {
      u = __builtin_bit_cast(double, (ub & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32));
    } else {
      k += 1;
      u = __builtin_bit_cast(double, (ub & 0xFFFFFFFFull) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32));
      hu = (0x00100000 - hu) >> 2;
    }
    f = u - 1.0;
  }
  const double hfsq = 0.5 * f * f;
  if (hu == 0) {
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    const double R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  const double s = f / (2.0 + f);
  const double z = s * s;
  const double R1 = z * Lp1;
  const double z2 = z * z;
  const double R2 = Lp2 + z * Lp3;
  const double z4 = z2 * z2;
  const double R3 = Lp4 + z * Lp5;
  const double z6 = z4 * z2;
  const double R4 = Lp6 + z * Lp7;
  const double R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// One generator stream held by a wavefront (every lane holds the same values: uniform code).
struct Stream {
  u128 s, inc;            // LCG state (before the next draw) and increment
  uint32_t has32, cached; // pcg64_next32's cache
  const uint64_t* jump;   // LDS: kJump x (A.lo, A.hi, G.lo, G.hi)
  const uint64_t* zig;    // LDS: ki[256], wi bits[256], fi bits[256]

  // state n draws ahead of `s`, 1 <= n <= kJump
  __device__ __forceinline__ u128 ahead(int n) const {
    const u128 A{jump[4 * (n - 1)], jump[4 * (n - 1) + 1]}, G{jump[4 * (n - 1) + 2], jump[4 * (n - 1) + 3]};
    return add128(mul128(A, s), mul128(G, inc));
  }
  __device__ __forceinline__ void advance(int n) { s = ahead(n); }
  __device__ __forceinline__ uint64_t next64() {
    s = add128(mul128(u128{kMultLo, kMultHi}, s), inc);
    return output_xsl_rr(s);
  }
  __device__ __forceinline__ uint32_t next32() {
    if (has32) { has32 = 0; return cached; }
    const uint64_t r = next64();
    has32 = 1; cached = (uint32_t)(r >> 32);
    return (uint32_t)r;
  }
  __device__ __forceinline__ double next_double() { return to_double(next64()); }
  __device__ __forceinline__ double uniform(double low, double high) { return low + (high - low) * next_double(); }
  // Generator.integers(0, n, size=1)[0], 1 < n < 2^32 - 1
  __device__ __forceinline__ uint32_t bounded(uint32_t n) {
    const uint32_t rng = n - 1;
    if (rng == 0) return 0;
    const uint32_t rng_excl = rng + 1;
    uint64_t m = (uint64_t)next32() * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
      const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
      while (leftover < threshold) {
        m = (uint64_t)next32() * rng_excl;
        leftover = (uint32_t)m;
      }
    }
    return (uint32_t)(m >> 32);
  }

  // `count` values loc + scale * standard_normal() in stream order to dst[0 .. count) (dst may be nullptr: draws consumed,
  // nothing stored).  Called by all 64 lanes of the wavefront; A_l / C_l: this lane's jump constants (mult^(l+1), inc * G_(l+1)).
  __device__ __forceinline__ void normals(int count, double loc, double scale, double* dst, int lane, u128 A_l, u128 C_l) {
    const double zr = 3.6541528853610087963519472518, zinv = 0.27366123732975827203338247596;
    int rank = 0;
    while (rank < count) {
      const uint64_t raw = output_xsl_rr(add128(mul128(A_l, s), C_l));      // the draw lane + 1 positions ahead
      const int idx = (int)(raw & 0xff);
      const uint64_t r8 = raw >> 8;
      const uint64_t rabs = (r8 >> 1) & 0x000fffffffffffffull;
      double x = (double)rabs * __builtin_bit_cast(double, zig[256 + idx]);
      if (r8 & 1) x = -x;
      const int m = min(64, count - rank);
      const bool slow = !(rabs < zig[idx]) && lane < m;
      const unsigned long long sm = __ballot(slow);
      const int L = sm ? (__ffsll((long long)sm) - 1) : m;                  // draws ahead of the first slow one: all fast
      if (lane < L && dst) dst[rank + lane] = loc + scale * x;
      rank += L;
      if (L == m) { advance(m); continue; }
      // the draw at position L + 1 takes a slow path of random_standard_normal: resolved in uniform code.  The draws it
      // consumes next (positions L + 2, ...) are the ones lanes L + 1, ... hold already: read them across the wave; only past
      // lane 63 the state is jumped ahead.
      auto raw_at = [&](int pos1) -> uint64_t {                             // the draw `pos1` positions ahead of `s` (1-based)
        if (pos1 <= 64) {
          const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(uint32_t)raw, pos1 - 1);
          const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(uint32_t)(raw >> 32), pos1 - 1);
          return ((uint64_t)hi << 32) | lo;
        }
        return output_xsl_rr(ahead(pos1));
      };
      const uint64_t rawL = raw_at(L + 1);
      const int idxL = (int)(rawL & 0xff);
      const uint64_t rL = rawL >> 8;
      const uint64_t rabsL = (rL >> 1) & 0x000fffffffffffffull;
      double xL = (double)rabsL * __builtin_bit_cast(double, zig[256 + idxL]);
      if (rL & 1) xL = -xL;
      int pos = L + 1;                                                      // draws consumed so far, counted from `s`
      bool produced = false;
      double val = 0.0;
      if (idxL == 0) {
        for (;;) {
          if (pos + 2 > kJump) break;                                       // (never in practice: > 30 rejected tail pairs in a row)
          const double u1 = to_double(raw_at(pos + 1));
          const double u2 = to_double(raw_at(pos + 2));
          pos += 2;
          const double xx = -zinv * log1p_fdlibm(-u1);
          const double yy = -log1p_fdlibm(-u2);
          if (yy + yy > xx * xx) { val = ((rabsL >> 8) & 1) ? -(zr + xx) : zr + xx; produced = true; break; }
        }
        if (!produced) {                                                    // continue the tail loop from a re-based state
          advance(pos); pos = 0;
          for (;;) {
            const double u1 = next_double(), u2 = next_double();
            const double xx = -zinv * log1p_fdlibm(-u1);
            const double yy = -log1p_fdlibm(-u2);
            if (yy + yy > xx * xx) { val = ((rabsL >> 8) & 1) ? -(zr + xx) : zr + xx; produced = true; break; }
          }
        }
      } else {
        const double uu = to_double(raw_at(pos + 1));
        pos += 1;
        const double f0 = __builtin_bit_cast(double, zig[512 + idxL - 1]), f1 = __builtin_bit_cast(double, zig[512 + idxL]);
        if ((f0 - f1) * uu + f1 < exp(-0.5 * xL * xL)) { val = xL; produced = true; }
      }
      if (produced) {
        if (lane == 0 && dst) dst[rank] = loc + scale * val;
        rank += 1;
      }
      if (pos > 0) advance(pos);
    }
  }
};
}  // namespace pcg
}  // namespace gsm
