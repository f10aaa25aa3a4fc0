// Host check of mcmc_gpu_amd/csrc/math_tables.h (table-driven log / sincos of the coefficient phase) against long double
// arithmetic; built and run by tests/test_math_tables.py.
#include "math_tables.h"
#include <cstdio>
#include <random>
#include <cmath>
int main() {
  double tab[gsm::kMathTabDoubles]; gsm::build_math_tables(tab);
  std::mt19937_64 g(1);
  long double max_abs = 0, max_rel_small = 0, max_sc = 0, max_rad = 0;
  for (int it = 0; it < 3000000; ++it) {
    uint64_t w = g();
    double u = ((double)(w >> 11) + 1.0) * 0x1p-53;   // (0, 1]
    if (it % 4 == 0) { u = 1.0 - (double)(w >> 40) * 0x1p-53; if (u <= 0) u = 0.5; }  // next to 1
    if (it % 4 == 1) { int e = (int)(w % 60); u = ldexp(u, -e); if (u < 0x1p-53) u = 0x1p-53; }
    double l = gsm::log_tab(u, tab);
    long double ref = logl((long double)u);
    long double err = fabsl((long double)l - ref);
    long double bound = 3e-16L + 2.5e-16L * fabsl(ref);
    if (err / bound > max_abs) max_abs = err / bound;
    if (u < 1.0) { long double rr = fabsl(sqrtl(-2 * ref) - sqrtl(-2 * (long double)l)); if (rr > max_rad) max_rad = rr; }
    double x = (double)(g() >> 11) * 0x1p-53; double s, c; gsm::sincos_tab(x, tab, s, c);
    long double a = 2.0L * 3.14159265358979323846264338327950288L * (long double)x;
    long double e2 = fmaxl(fabsl(s - sinl(a)), fabsl(c - cosl(a))); if (e2 > max_sc) max_sc = e2;
  }
  // general positive arguments (spectral density)
  long double max_gen = 0;
  for (int it = 0; it < 1000000; ++it) { double x = ldexp(1.0 + (double)(g() >> 11) * 0x1p-53, (int)(g() % 200) - 100); long double ref = logl((long double)x); long double err = fabsl(gsm::log_tab(x, tab) - ref) / (3e-16L + 2.5e-16L * fabsl(ref)); if (err > max_gen) max_gen = err; }
  printf("log err/bound %.3Lf general %.3Lf  rad abs err %.3Le  sincos abs err %.3Le\n", max_abs, max_gen, max_rad, max_sc);
  return !(max_abs < 1 && max_gen < 1 && max_sc < 3e-16L && max_rad < 1.5e-15L);
}
