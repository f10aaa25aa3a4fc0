// Normal-score transform of the small-scale chain on the device: scikit-learn's QuantileTransformer(output_distribution=
// "normal"), one feature, as chain_sgs.run applies it to the whole map before and after every SGS block (MCMC.py:1766-1777;
// the drivers fit it with n_quantiles = 1000, smallScaleChain_multiprocessing.py:493-496).  Restated from the published
// algorithms the reference's dependencies use (neither is part of the reference tree):
//   * sklearn.preprocessing.QuantileTransformer._transform_col (scikit-learn 1.7): bounds test with BOUNDS_THRESHOLD = 1e-7,
//     numpy.interp forwards and backwards averaged (forward transform), numpy.interp (inverse), clipping of the normal scores
//     at ppf(1e-7 - spacing(1)) / ppf(1 - (1e-7 - spacing(1)));
//   * scipy.stats.norm.ppf / cdf = Cephes ndtri / ndtr (S. L. Moshier, Cephes Math Library 2.2; scipy/special/xsf/cephes/ndtr.h).
// The file compiles for the host too: tests/test_normal_score.py checks it against scipy and scikit-learn with g++.
#pragma once
#include <cmath>
#include <cstdint>
#if !defined(__HIPCC__)
#define GSM_NS_FN inline
#else
#define GSM_NS_FN __host__ __device__ __forceinline__
#endif

namespace gsm {
namespace ns {

GSM_NS_FN double polevl(double x, const double* c, int n) { double a = c[0]; for (int i = 1; i <= n; ++i) a = a * x + c[i]; return a; }
GSM_NS_FN double p1evl(double x, const double* c, int n) { double a = x + c[0]; for (int i = 1; i < n; ++i) a = a * x + c[i]; return a; }

GSM_NS_FN double erf_c(double x);
GSM_NS_FN double erfc_c(double a) {
  const double P[] = {2.46196981473530512524E-10, 5.64189564831068821977E-1, 7.46321056442269912687E0, 4.86371970985681366614E1,
                      1.96520832956077098242E2, 5.26445194995477358631E2, 9.34528527171957607540E2, 1.02755188689515710272E3,
                      5.57535335369399327526E2};
  const double Q[] = {1.32281951154744992508E1, 8.67072140885989742329E1, 3.54937778887819891062E2, 9.75708501743205489753E2,
                      1.82390916687909736289E3, 2.24633760818710981792E3, 1.65666309194161350182E3, 5.57535340817727675546E2};
  const double R[] = {5.64189583547755073984E-1, 1.27536670759978104416E0, 5.01905042251180477414E0, 6.16021097993053585195E0,
                      7.40974269950448939160E0, 2.97886665372100240670E0};
  const double S[] = {2.26052863220117276590E0, 9.39603524938001434673E0, 1.20489539808096656605E1, 1.70814450747565897222E1,
                      9.60896809063285878198E0, 3.36907645100081516050E0};
  if (std::isnan(a)) return a;
  const double x = (a < 0.0) ? -a : a;
  if (x < 1.0) return 1.0 - erf_c(a);
  double z = -a * a;
  if (z < -7.09782712893383996732E2) return (a < 0) ? 2.0 : 0.0;
  z = std::exp(z);
  double p, q;
  if (x < 8.0) { p = polevl(x, P, 8); q = p1evl(x, Q, 8); }
  else { p = polevl(x, R, 5); q = p1evl(x, S, 6); }
  double y = (z * p) / q;
  if (a < 0) y = 2.0 - y;
  if (y != 0.0) return y;
  return (a < 0) ? 2.0 : 0.0;
}
GSM_NS_FN double erf_c(double x) {
  const double T[] = {9.60497373987051638749E0, 9.00260197203842689217E1, 2.23200534594684319226E3, 7.00332514112805075473E3,
                      5.55923013010394962768E4};
  const double U[] = {3.35617141647503099647E1, 5.21357949780152679795E2, 4.59432382970980127987E3, 2.26290000613890934246E4,
                      4.92673942608635921086E4};
  if (std::isnan(x)) return x;
  if (x < 0.0) return -erf_c(-x);
  if (x > 1.0) return 1.0 - erfc_c(x);
  const double z = x * x;
  return x * polevl(z, T, 4) / p1evl(z, U, 5);
}
// scipy.special.ndtr = scipy.stats.norm.cdf
GSM_NS_FN double ndtr(double a) {
  if (std::isnan(a)) return a;
  const double x = a * 0.70710678118654752440;
  const double z = std::fabs(x);
  if (z < 1.0) return 0.5 + 0.5 * erf_c(x);
  double y = 0.5 * erfc_c(z);
  if (x > 0) y = 1.0 - y;
  return y;
}
// scipy.special.ndtri = scipy.stats.norm.ppf (Cephes ndtri.c)
GSM_NS_FN double ndtri(double y0) {
  const double P0[] = {-5.99633501014107895267E1, 9.80010754185999661536E1, -5.66762857469070293439E1, 1.39312609387279679503E1,
                       -1.23916583867381258016E0};
  const double Q0[] = {1.95448858338141759834E0, 4.67627912898881538453E0, 8.63602421390890590575E1, -2.25462687854119370527E2,
                       2.00260212380060660359E2, -8.20372256168333339912E1, 1.59056225126211695515E1, -1.18331621121330003142E0};
  const double P1[] = {4.05544892305962419923E0, 3.15251094599893866154E1, 5.71628192246421288162E1, 4.40805073893200834700E1,
                       1.46849561928858024014E1, 2.18663306850790267539E0, -1.40256079171354495875E-1, -3.50424626827848203418E-2,
                       -8.57456785154685413611E-4};
  const double Q1[] = {1.57799883256466749731E1, 4.53907635128879210584E1, 4.13172038254672030440E1, 1.50425385692907503408E1,
                       2.50464946208309415979E0, -1.42182922854787788574E-1, -3.80806407691578277194E-2, -9.33259480895457427372E-4};
  const double P2[] = {3.23774891776946035970E0, 6.91522889068984211695E0, 3.93881025292474443415E0, 1.33303460815807542389E0,
                       2.01485389549179081538E-1, 1.23716634817820021358E-2, 3.01581553508235416007E-4, 2.65806974686737550832E-6,
                       6.23974539184983293730E-9};
  const double Q2[] = {6.02427039364742014255E0, 3.67983563856160859403E0, 1.37702099489081330271E0, 2.16236993594496635890E-1,
                       1.34204006088543189037E-2, 3.28014464682127739104E-4, 2.89247864745380683936E-6, 6.79019408009981274425E-9};
  if (std::isnan(y0)) return y0;
  if (y0 == 0.0) return -INFINITY;
  if (y0 == 1.0) return INFINITY;
  if (y0 < 0.0 || y0 > 1.0) return NAN;
  bool negate = true;
  double y = y0;
  if (y > 1.0 - 0.13533528323661269189) { y = 1.0 - y; negate = false; }      // exp(-2)
  if (y > 0.13533528323661269189) {
    y = y - 0.5;
    const double y2 = y * y;
    double x = y + y * (y2 * polevl(y2, P0, 4) / p1evl(y2, Q0, 8));
    return x * 2.50662827463100050242E0;                                       // sqrt(2 pi)
  }
  double x = std::sqrt(-2.0 * std::log(y));
  const double x0 = x - std::log(x) / x;
  const double z = 1.0 / x;
  const double x1 = (x < 8.0) ? z * polevl(z, P1, 8) / p1evl(z, Q1, 8) : z * polevl(z, P2, 8) / p1evl(z, Q2, 8);
  x = x0 - x1;
  return negate ? -x : x;
}

// numpy.interp(x, xp, fp) for finite x, xp ascending (repeats allowed), n >= 1 (numpy/core/src/multiarray/compiled_base.c)
GSM_NS_FN double interp(double x, const double* xp, const double* fp, int n) {
  if (x > xp[n - 1]) return fp[n - 1];
  if (x < xp[0]) return fp[0];
  int lo = 0, hi = n;                       // j = (number of xp <= x) - 1
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (xp[mid] <= x) lo = mid + 1; else hi = mid; }
  const int j = lo - 1;
  if (j == n - 1) return fp[j];
  if (xp[j] == x) return fp[j];
  const double slope = (fp[j + 1] - fp[j]) / (xp[j + 1] - xp[j]);
  double r = slope * (x - xp[j]) + fp[j];
  if (std::isnan(r)) { r = slope * (x - xp[j + 1]) + fp[j + 1]; if (std::isnan(r) && fp[j] == fp[j + 1]) r = fp[j]; }
  return r;
}
// the same with xp' = -xp reversed, fp' = -fp reversed evaluated at -x (the backward pass of the forward transform),
// without materialising the reversed arrays: xp'[i] = -xp[n - 1 - i]
GSM_NS_FN double interp_reversed(double x, const double* xp, const double* fp, int n) {
  const double xm = -x;
  if (xm > -xp[0]) return -fp[0];
  if (xm < -xp[n - 1]) return -fp[n - 1];
  int lo = 0, hi = n;                       // j' = (number of i with -xp[n-1-i] <= xm) - 1
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (-xp[n - 1 - mid] <= xm) lo = mid + 1; else hi = mid; }
  const int j = lo - 1;
  const double xj = -xp[n - 1 - j], fj = -fp[n - 1 - j];
  if (j == n - 1) return fj;
  if (xj == xm) return fj;
  const double xj1 = -xp[n - 2 - j], fj1 = -fp[n - 2 - j];
  const double slope = (fj1 - fj) / (xj1 - xj);
  double r = slope * (xm - xj) + fj;
  if (std::isnan(r)) { r = slope * (xm - xj1) + fj1; if (std::isnan(r) && fj == fj1) r = fj; }
  return r;
}

constexpr double kBoundsThreshold = 1e-7;
// QuantileTransformer.transform of one value: quantiles q[nq] (ascending), references ref[nq] (linspace(0, 1, nq))
GSM_NS_FN double qt_forward(double x, const double* q, const double* ref, int nq, double clip_min, double clip_max) {
  if (std::isnan(x)) return x;
  const bool lower = x - kBoundsThreshold < q[0], upper = x + kBoundsThreshold > q[nq - 1];
  double p = 0.5 * (interp(x, q, ref, nq) - interp_reversed(x, q, ref, nq));
  if (upper) p = 1.0;
  if (lower) p = 0.0;
  const double z = ndtri(p);
  return std::fmin(std::fmax(z, clip_min), clip_max);
}
// QuantileTransformer.inverse_transform of one value
GSM_NS_FN double qt_inverse(double z, const double* q, const double* ref, int nq) {
  if (std::isnan(z)) return z;
  const double u = ndtr(z);
  const bool lower = u - kBoundsThreshold < 0.0, upper = u + kBoundsThreshold > 1.0;
  double x = interp(u, ref, q, nq);
  if (upper) x = q[nq - 1];
  if (lower) x = q[0];
  return x;
}

}  // namespace ns
}  // namespace gsm
