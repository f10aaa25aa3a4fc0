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

constexpr int kJump = 512;                  // jump table entries: (A_n, G_n) for n = 1 .. 512 at index n - 1 (8 windows of 64 draws)
constexpr int kJumpWave = 64;               // what a kernel with one wavefront per stream stages in LDS
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

// A tail start (idx == 0 on the slow path: about 1 normal in 3 700) at the head of the stream, in uniform code with sequential
// draws: pairs of uniforms until yy + yy > xx * xx (random_standard_normal's tail loop, libm's log1p restated).  Out of line -- one
// copy in the kernel -- and by value: a member function would take the stream's address and put its state into scratch memory.
struct TailDraw { double val; u128 s; };
__device__ __noinline__ inline TailDraw tail_normal(u128 s, u128 inc) {
  const double zr = 3.6541528853610087963519472518, zinv = 0.27366123732975827203338247596;
  const u128 M{kMultLo, kMultHi};
  s = add128(mul128(M, s), inc);
  const uint64_t raw0 = output_xsl_rr(s);
  const uint64_t rabs0 = ((raw0 >> 8) >> 1) & 0x000fffffffffffffull;
  for (;;) {
    s = add128(mul128(M, s), inc);
    const double u1 = to_double(output_xsl_rr(s));
    s = add128(mul128(M, s), inc);
    const double u2 = to_double(output_xsl_rr(s));
    const double xx = -zinv * log1p_fdlibm(-u1);
    const double yy = -log1p_fdlibm(-u2);
    if (yy + yy > xx * xx) return TailDraw{((rabs0 >> 8) & 1) ? -(zr + xx) : zr + xx, s};
  }
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

  // Generator.integers(low, high, size=1)[0]
  __device__ __forceinline__ int integers(int low, int high) { return low + (int)bounded((uint32_t)(high - low)); }
  // random_interval: uniform on [0, mx] by masked rejection on 32-bit words (Generator.shuffle's index draw)
  __device__ __forceinline__ uint32_t interval(uint32_t mx) {
    if (mx == 0) return 0;
    uint32_t mask = mx;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    uint32_t v;
    do { v = next32() & mask; } while (v > mx);
    return v;
  }

  // the state `n` draws ahead of `s`, 1 <= n <= 64, from the lanes' states of the current window (lane n - 1 holds it)
  static __device__ __forceinline__ u128 lane_state(u128 st, int n) {
    u128 r;
    r.lo = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)st.lo, n - 1) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(st.lo >> 32), n - 1) << 32);
    r.hi = (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)st.hi, n - 1) |
           ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(st.hi >> 32), n - 1) << 32);
    return r;
  }

  // `count` values loc + scale * standard_normal() in stream order to dst[0 .. count) (dst may be nullptr: draws consumed,
  // nothing stored).  Called by all 64 lanes of the wavefront; A_l / C_l: this lane's jump constants (mult^(l+1), inc * G_(l+1)).
  //
  // A window = the next 64 draws, one per lane.  random_standard_normal consumes ONE draw per normal on its fast path (99.3 %),
  // TWO on the wedge path (the draw itself and a uniform; the normal is produced or not) and 2 k + 1 in the tail (idx == 0).  So
  // inside a window the positions that START a normal follow from the lanes' own decodes: every position is a start unless it is
  // the uniform of a wedge start before it.  All starts of a window -- fast and wedge alike -- are resolved in ONE pass (the wedge
  // test in vector code under its lanes' EXEC), ranks by a prefix count over the ballot of produced normals; the window is cut
  // before a tail start (resolved in uniform code at the head of the next window) and before a wedge start in lane 63 (its
  // uniform is the next window's draw).  The state behind the consumed draws is the one a lane already holds (v_readlane).
  __device__ __forceinline__ void normals(int count, double loc, double scale, double* dst, int lane, u128 A_l, u128 C_l) {
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int rank = 0;
    while (rank < count) {
      const u128 st = add128(mul128(A_l, s), C_l);                          // the state lane + 1 draws ahead
      const uint64_t raw = output_xsl_rr(st);
      const int idx = (int)(raw & 0xff);
      const uint64_t r8 = raw >> 8;
      const uint64_t rabs = (r8 >> 1) & 0x000fffffffffffffull;
      // (double)rabs, rabs < 2^52: the integer sits in the mantissa of 2^52 + rabs (exact)
      double x = (__builtin_bit_cast(double, 0x4330000000000000ull | rabs) - 4503599627370496.0) * __builtin_bit_cast(double, zig[256 + idx]);
      if (r8 & 1) x = -x;
      const bool slow = !(rabs < zig[idx]);
      const unsigned long long slow_mask = __ballot(slow);
      const int need = count - rank;
      if (slow_mask == 0ull) {
        // ---- all 64 draws take the fast path (64 % of the windows): 64 normals, 64 draws -- or the rest of the plane ----------
        if (lane < need && dst) dst[rank + lane] = loc + scale * x;
        const int used = min(64, need);
        rank += used;
        s = lane_state(st, used);
        continue;
      }
      const unsigned long long tail_mask = __ballot(slow && idx == 0);
      if (tail_mask & 1ull) {
        const TailDraw td = tail_normal(s, inc);
        // a call returns in vector registers: back to scalar ones, or every later use of the state becomes vector code
        s.lo = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)td.s.lo) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(td.s.lo >> 32)) << 32);
        s.hi = (uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)td.s.hi) |
               ((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(td.s.hi >> 32)) << 32);
        if (lane == 0 && dst) dst[rank] = loc + scale * td.val;
        rank += 1;
        continue;
      }
      // ---- starts of the window ---------------------------------------------------------------------------------------
      unsigned long long start_mask = ~0ull;
      for (unsigned long long t = slow_mask & ~tail_mask; t; t &= t - 1) {  // wedge positions in ascending order
        const int bpos = __ffsll((long long)t) - 1;
        if (((start_mask >> bpos) & 1ull) && bpos < 63) start_mask &= ~(1ull << (bpos + 1));   // its uniform is not a start
      }
      const unsigned long long wedge63 = (slow_mask & ~tail_mask) & start_mask & (1ull << 63);
      const unsigned long long cut = (tail_mask & start_mask) | wedge63;
      const int limit = cut ? (__ffsll((long long)cut) - 1) : 64;           // >= 1: lane 0 is a start and neither cut case
      const bool is_start = ((start_mask >> lane) & 1ull) && lane < limit;
      // the uniform of a wedge start is the next lane's draw (all lanes active here: the exchange reads live lanes)
      const uint64_t raw_next = (uint64_t)(uint32_t)__shfl_down((int)(uint32_t)raw, 1, 64) |
                                ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(raw >> 32), 1, 64) << 32);
      bool produced = is_start && !slow;
      if (is_start && slow) {
        const double f0 = __builtin_bit_cast(double, zig[512 + idx - 1]), f1 = __builtin_bit_cast(double, zig[512 + idx]);
        produced = (f0 - f1) * to_double(raw_next) + f1 < exp(-0.5 * x * x);
      }
      const unsigned long long pmask = __ballot(produced);
      const int got = __popcll(pmask);
      const int my = __popcll(pmask & below);
      if (produced && my < need && dst) dst[rank + my] = loc + scale * x;
      int consumed = limit;
      if (got >= need) {                                                    // the plane ends inside (or at the end of) the window
        // lane of the need-th produced normal; the draws up to it (and its uniform, if it is a wedge start) are consumed --
        // nothing behind it, not even a rejected wedge start: NumPy stops drawing with the last normal of the array
        unsigned long long t = pmask;
        for (int k = 1; k < need; ++k) t &= t - 1;
        const int q = __ffsll((long long)t) - 1;
        consumed = q + 1 + (int)((slow_mask >> q) & 1ull);
      }
      rank += min(got, need);
      s = lane_state(st, consumed);
    }
  }

  // normals() with NW wavefronts per stream (a workgroup of NW x 64 threads; every wavefront holds the same Stream and calls this
  // together): wavefront w decodes the window of draws 64 w + 1 .. 64 w + 64 ahead of `s` ON THE ASSUMPTION that every window
  // before it is consumed whole -- true unless one of them is cut (before a tail start, before a wedge start in its last lane, at
  // the end of the plane: 2-3 % of the windows).  Each wavefront analyses its window as normals() does and publishes (draws it
  // would consume, normals it would produce); after a barrier every wavefront walks the NW entries in order -- the same uniform
  // code everywhere -- and knows the rank its normals start at, whether its window counts at all (no cut before it), and the new
  // rank; the last window that counts hands the state behind its consumed draws to the others through LDS.  A tail start at the
  // head of window 0 is resolved by all wavefronts redundantly with sequential draws (1 normal in 3 700).
  // A_wl / C_wl: jump constants of THIS lane of THIS wavefront (mult^(64 w + l + 1), inc * G_(64 w + l + 1)); xch: LDS, 2 NW + 4 words.
  template <int NW>
  __device__ __forceinline__ void normals_mw(int count, double loc, double scale, double* dst, int lane, int wave, u128 A_wl, u128 C_wl,
                                             uint64_t* xch) {
    const double zr = 3.6541528853610087963519472518, zinv = 0.27366123732975827203338247596;
    const unsigned long long below = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    int* info = (int*)xch;                      // info[2 w] = draws window w consumes if it counts whole, info[2 w + 1] = normals produced
    uint64_t* st_x = xch + NW;                  // the state behind the last counted window
    int rank = 0;
    while (rank < count) {
      const u128 st = add128(mul128(A_wl, s), C_wl);                        // the state 64 wave + lane + 1 draws ahead
      const uint64_t raw = output_xsl_rr(st);
      const int idx = (int)(raw & 0xff);
      const uint64_t r8 = raw >> 8;
      const uint64_t rabs = (r8 >> 1) & 0x000fffffffffffffull;
      double x = (double)rabs * __builtin_bit_cast(double, zig[256 + idx]);
      if (r8 & 1) x = -x;
      const bool slow = !(rabs < zig[idx]);
      const unsigned long long slow_mask = __ballot(slow), tail_mask = __ballot(slow && idx == 0);
      int limit = 0;
      unsigned long long pmask = 0ull;
      bool produced = false;
      if (!(tail_mask & 1ull)) {
        unsigned long long start_mask = ~0ull;
        for (unsigned long long t = slow_mask & ~tail_mask; t; t &= t - 1) {
          const int bpos = __ffsll((long long)t) - 1;
          if (((start_mask >> bpos) & 1ull) && bpos < 63) start_mask &= ~(1ull << (bpos + 1));
        }
        const unsigned long long wedge63 = (slow_mask & ~tail_mask) & start_mask & (1ull << 63);
        const unsigned long long cut = (tail_mask & start_mask) | wedge63;
        limit = cut ? (__ffsll((long long)cut) - 1) : 64;
        const bool is_start = ((start_mask >> lane) & 1ull) && lane < limit;
        const uint64_t raw_next = (uint64_t)(uint32_t)__shfl_down((int)(uint32_t)raw, 1, 64) |
                                  ((uint64_t)(uint32_t)__shfl_down((int)(uint32_t)(raw >> 32), 1, 64) << 32);
        produced = is_start && !slow;
        if (is_start && slow) {
          const double f0 = __builtin_bit_cast(double, zig[512 + idx - 1]), f1 = __builtin_bit_cast(double, zig[512 + idx]);
          produced = (f0 - f1) * to_double(raw_next) + f1 < exp(-0.5 * x * x);
        }
        pmask = __ballot(produced);
      }
      const int got = __popcll(pmask);
      if (lane == 0) { info[2 * wave] = limit; info[2 * wave + 1] = got; }
      __syncthreads();
      if (info[0] == 0) {
        // ---- the stream continues with a tail start: every wavefront resolves it with sequential draws -------------------
        const uint64_t raw0 = next64();
        const uint64_t rabs0 = ((raw0 >> 8) >> 1) & 0x000fffffffffffffull;
        double val;
        for (;;) {
          const double u1 = next_double(), u2 = next_double();
          const double xx = -zinv * log1p_fdlibm(-u1);
          const double yy = -log1p_fdlibm(-u2);
          if (yy + yy > xx * xx) { val = ((rabs0 >> 8) & 1) ? -(zr + xx) : zr + xx; break; }
        }
        if (wave == 0 && lane == 0 && dst) dst[rank] = loc + scale * val;
        rank += 1;
        __syncthreads();                                                    // info[] is rewritten by the next pass
        continue;
      }
      // ---- which windows count, and where their normals go (uniform code, the same in every wavefront) -------------------
      int base = rank, my_base = 0, last = 0, new_rank = rank;
      bool open = true, mine = false, ends_here = false;
#pragma unroll
      for (int u = 0; u < NW; ++u) {
        const int lim_u = info[2 * u], got_u = info[2 * u + 1];
        if (open && lim_u > 0) {
          if (u == wave) { mine = true; my_base = base; }
          last = u;
          const int need_u = count - base;
          if (got_u >= need_u) { open = false; base = count; if (u == wave) ends_here = true; }   // the plane ends inside window u
          else { base += got_u; if (lim_u < 64) open = false; }
        } else {
          open = false;
        }
      }
      new_rank = base;
      if (mine) {
        const int need = count - my_base;
        const int my = __popcll(pmask & below);
        if (produced && my < need && dst) dst[my_base + my] = loc + scale * x;
        if (wave == last) {
          int consumed = limit;
          if (ends_here) {
            unsigned long long t = pmask;
            for (int k = 1; k < need; ++k) t &= t - 1;
            const int q = __ffsll((long long)t) - 1;
            consumed = q + 1 + (int)((slow_mask >> q) & 1ull);
          }
          const u128 ns = lane_state(st, consumed);
          if (lane == 0) { st_x[0] = ns.lo; st_x[1] = ns.hi; }
        }
      }
      __syncthreads();
      s.lo = st_x[0]; s.hi = st_x[1];
      rank = new_rank;
    }
  }
};
}  // namespace pcg
}  // namespace gsm
