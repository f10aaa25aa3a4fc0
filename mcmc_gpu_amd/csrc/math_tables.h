// Table-driven log and sincos of the coefficient phase (proposal_device.h): one 16-byte table read and a short polynomial
// instead of a reciprocal / quadrant reduction and a long one.  The file compiles for the host too (tests/test_math_tables.py
// checks both functions against long double arithmetic with g++).
//
// Table (kMathTabDoubles doubles; built once on the host by build_math_tables, staged in LDS by every kernel that draws):
//   [2 i], [2 i + 1]            i < 128: 1 / c_i and log(c_i) with c_i = 1 + (i + 1) / 128, the upper edge of mantissa interval i
//   [256 + 2 j], [256 + 2 j + 1] j < 64 : sin(2 pi j / 64), cos(2 pi j / 64)
#pragma once
#include <cstdint>
#include <cstring>
#include <cmath>
#if !defined(__HIPCC__)
#define GSM_MT_FN inline
#else
#define GSM_MT_FN __host__ __device__ __forceinline__
#endif

namespace gsm {

constexpr int kMathTabDoubles = 384;

namespace mt {
GSM_MT_FN double fma_c(double a, double b, double C) {
#if defined(__HIP_DEVICE_COMPILE__)
  double r;      // constant from a scalar register pair (see fma_sc in proposal_device.h)
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(C));
  return r;
#else
  return __builtin_fma(a, b, C);
#endif
}
GSM_MT_FN uint64_t bits_of(double x) { uint64_t b; memcpy(&b, &x, 8); return b; }
GSM_MT_FN double double_of(uint64_t b) { double x; memcpy(&x, &b, 8); return x; }
}  // namespace mt

// log(x) for positive, finite, normal x.  x = 2^e m, m in [1, 2); interval i = top 7 mantissa bits; r = m / c_i - 1 in
// (-1/129, 0] (exactly m / 2 - 1 in the last interval, c = 2); log x = e ln2 + log c_i + log1p(r), log1p by its series to
// r^7 (next term < 2e-18).  Absolute error < 3e-16 + 2.5e-16 |log x|; for x -> 1 from below (e = -1, c = 2: the first two
// terms cancel exactly) the error is relative, so sqrt(-2 log u) keeps its accuracy for uniforms next to 1.
GSM_MT_FN double log_tab(double x, const double* tab) {
  const uint64_t b = mt::bits_of(x);
  const int e = (int)(b >> 52) - 1023;
  const int i = (int)(b >> 45) & 127;
  const double m = mt::double_of((b & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull);
  const double inv_c = tab[2 * i], log_c = tab[2 * i + 1];
  const double r = __builtin_fma(m, inv_c, -1.0);
  double q = mt::fma_c(r, 1.0 / 7.0, -1.0 / 6.0);
  q = mt::fma_c(r, q, 0.2);
  q = mt::fma_c(r, q, -0.25);
  q = mt::fma_c(r, q, 1.0 / 3.0);
  q = mt::fma_c(r, q, -0.5);
  const double p = __builtin_fma(r * r, q, r);
  // one constant for e ln2 and for the table's log 2: for e = -1, c = 2 the two cancel exactly.  (RN(ln 2) is off by 2.3e-17:
  // at most 1.3e-15 absolute at e = -53, 3e-17 relative.)
  return mt::fma_c((double)e, 6.93147180559945286227e-01, log_c) + p;
}

// (sin, cos)(2 pi u) for u in [0, 1): a = 64 u (exact), j = rint(a), t = (a - j) 2 pi / 64 with |t| <= 0.0491; angle
// addition with the table's (sin, cos)(2 pi j / 64) and the series of sin t (to t^7) and cos t (to t^8).  Absolute error
// < 3e-16.
GSM_MT_FN void sincos_tab(double u, const double* tab, double& s, double& c) {
  const double a = u * 64.0;
  const double n = __builtin_rint(a);
  const double t = (a - n) * 9.81747704246810387019e-02;      // 2 pi / 64
  const int j = (int)n & 63;
  const double sj = tab[256 + 2 * j], cj = tab[256 + 2 * j + 1];
  const double z = t * t;
  double ps = mt::fma_c(z, -1.0 / 5040.0, 1.0 / 120.0);
  ps = mt::fma_c(z, ps, -1.0 / 6.0);
  const double st = __builtin_fma(t * z, ps, t);
  double pc = mt::fma_c(z, 1.0 / 40320.0, -1.0 / 720.0);
  pc = mt::fma_c(z, pc, 1.0 / 24.0);
  pc = mt::fma_c(z, pc, -0.5);
  const double ct = __builtin_fma(z, pc, 1.0);
  s = __builtin_fma(sj, ct, cj * st);
  c = __builtin_fma(cj, ct, -(sj * st));
}

// host: the table, rounded from long double
inline void build_math_tables(double* tab) {
  for (int i = 0; i < 128; ++i) {
    const long double c = 1.0L + (long double)(i + 1) / 128.0L;
    tab[2 * i] = (double)(1.0L / c);
    tab[2 * i + 1] = (i == 127) ? 6.93147180559945286227e-01 : (double)logl(c);     // c = 2: the constant log_tab multiplies e by
  }
  for (int j = 0; j < 64; ++j) {
    const long double ang = 2.0L * 3.14159265358979323846264338327950288L * (long double)j / 64.0L;
    tab[256 + 2 * j] = (j % 32 == 0) ? 0.0 : (double)sinl(ang);
    tab[256 + 2 * j + 1] = (j % 32 == 16) ? 0.0 : (double)cosl(ang);
  }
}

}  // namespace gsm
