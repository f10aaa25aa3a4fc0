// Host build of mcmc_gpu_amd/csrc/normal_score.h as a tiny shared library for tests/test_normal_score.py (ctypes).
#include "normal_score.h"
extern "C" {
void ns_ndtri(const double* x, double* out, int n) { for (int i = 0; i < n; ++i) out[i] = gsm::ns::ndtri(x[i]); }
void ns_ndtr(const double* x, double* out, int n) { for (int i = 0; i < n; ++i) out[i] = gsm::ns::ndtr(x[i]); }
void ns_qt(const double* x, double* out, int n, const double* q, const double* ref, int nq, double clip_min, double clip_max, int inverse) {
  for (int i = 0; i < n; ++i) out[i] = inverse ? gsm::ns::qt_inverse(x[i], q, ref, nq) : gsm::ns::qt_forward(x[i], q, ref, nq, clip_min, clip_max);
}
}
