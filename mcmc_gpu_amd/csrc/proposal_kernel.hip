// Philox proposal generator for gfx950 (MI355X): one workgroup per (chain, step) proposal.
//
// Replaces RandField.get_rfblock + spectral_synthesis_field (reference gstatsMCMC/MCMC.py:742-778,
// :176-254), the block-centre rejection loop (MCMC.py:1253-1261) and rng.random() (MCMC.py:1336) with
// a counter-based generator.  The draws are different numbers than NumPy's PCG64 stream gives, so this
// path is checked (a) value-for-value against the CPU restatement of THIS algorithm
// (oracle/philox_oracle.py, <=1e-10) and (b) in distribution against the reference's spectral
// proposal (covariance, accept rate).
//
// Same distribution, half the work.  The reference takes Re(ifft2((N1 + i N2) * sqrt(S))) with two
// full planes of normals.  The real part of an inverse DFT is the inverse DFT of the Hermitian part of
// the spectrum, Zh[k] = sqrt(S[k]) * ((N1[k] + N1[-k])/2 + i (N2[k] - N2[-k])/2).  Its entries are
// independent over {k, -k} pairs with variance 1/2 per component (variance 1, real, where k == -k), so
// the kernel draws the half plane kx in [0, bw/2] directly -- bh*bw normals instead of 2*bh*bw -- and
// runs a complex-to-real inverse DFT: columns first (complex, bw/2+1 of them), then rows (real output).
//
// v1: both DFT stages are plain fp64 FMA loops over LDS-resident data and twiddle tables.

#include "gsm_internal.h"
#include "philox.h"
#include <math.h>

namespace gsm {

constexpr int kPBlock = 512;
constexpr int kPWaves = kPBlock / 64;

__device__ __forceinline__ uint32_t pmagic(uint32_t d) { return (uint32_t)(0xFFFFFFFFu / d) + 1u; }

__device__ __forceinline__ void normals2(uint64_t seed, int64_t step, uint32_t stream, uint32_t idx, double& g1,
                                         double& g2) {
  const u32x4 r = philox_draw(seed, step, stream, idx);
  const double u1 = u01_open0_from(r.x, r.y);
  const double u2 = u01_from(r.z, r.w);
  const double rad = sqrt(-2.0 * log(u1));
  double s, c;
  sincospi(2.0 * u2, &s, &c);
  g1 = rad * c;
  g2 = rad * s;
}

__device__ __forceinline__ double block_sum(double v, double* red, int tid) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  __syncthreads();  // red reuse
  if ((tid & 63) == 0) red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
#pragma unroll
  for (int w = 0; w < kPWaves; ++w) t += red[w];
  return t;
}

// frequency magnitude^2 helper: 2*pi*fftfreq(n, d=res)[k]
__device__ __forceinline__ double wavenumber(int k, int n, double res) {
  const int kk = (k < (n + 1) / 2) ? k : k - n;  // numpy fftfreq ordering (n even: k=n/2 -> -n/2)
  return ((double)kk / ((double)n * res)) * 2.0 * M_PI;
}

__global__ __launch_bounds__(kPBlock) void propose_kernel(const ProposeArgs a, const int cap_half) {
  extern __shared__ double plds[];
  // layout: Xr | Xi | Tr | Ti (cap_half each) | cy sy (max_bh each) | cx sx (max_bw each) | red
  double* Xr = plds;
  double* Xi = Xr + cap_half;
  double* Tr = Xi + cap_half;
  double* Ti = Tr + cap_half;
  double* cy = Ti + cap_half;
  double* sy = cy + a.B.max_bh;
  double* cx = sy + a.B.max_bh;
  double* sx = cx + a.B.max_bw;
  double* red = sx + a.B.max_bw;
  double* fieldv = Xr;  // bh*bw doubles <= 2*cap_half, overlays X once stage 1 is done

  const int tid = threadIdx.x;
  const int s = blockIdx.x, chain = blockIdx.y;
  const int64_t step = a.step0 + s;
  const uint64_t seed = a.seeds[chain];
  const int64_t rec = (int64_t)chain * a.n_steps + s;
  const gsm_rf_params& P = a.rf;

  // ---- scalar draws (every thread computes the same values) -----------------------------------
  const u32x4 d0 = philox_draw(seed, step, kStreamScalars, 0);
  const u32x4 d1 = philox_draw(seed, step, kStreamScalars, 1);
  const u32x4 d2 = philox_draw(seed, step, kStreamScalars, 2);
  const u32x4 d3 = philox_draw(seed, step, kStreamScalars, 3);
  const int si = (int)__umulhi(d3.x, (uint32_t)a.B.n_sizes);
  const double scale = (P.scale_min + (P.scale_max - P.scale_min) * u01_from(d0.x, d0.y)) / 3.0;
  const double nug = 0.0 + (P.nugget_max - 0.0) * u01_from(d0.z, d0.w);
  const double range_x = P.range_min_x + (P.range_max_x - P.range_min_x) * u01_from(d1.x, d1.y);
  const double range_y = P.isotropic ? range_x : P.range_min_y + (P.range_max_y - P.range_min_y) * u01_from(d1.z, d1.w);
  const double u_acc = u01_from(d2.x, d2.y);
  const uint64_t cw = ((uint64_t)d2.w << 32) | d2.z;
  const int cell = a.centres[(int)__umul64hi(cw, (uint64_t)a.n_centres)];
  const int bh = a.B.bh[si], bw = a.B.bw[si];
  const int ncol = bw / 2 + 1;
  if (tid == 0) {
    a.size_idx[rec] = si;
    a.centre[2 * rec] = cell / a.W;
    a.centre[2 * rec + 1] = cell - (cell / a.W) * a.W;
    a.u[rec] = u_acc;
    if (a.rf_scalars) {
      a.rf_scalars[4 * rec] = scale;
      a.rf_scalars[4 * rec + 1] = nug;
      a.rf_scalars[4 * rec + 2] = range_x;
      a.rf_scalars[4 * rec + 3] = range_y;
    }
  }

  // ---- twiddle tables of the two lengths -> LDS ------------------------------------------------
  {
    const double* ty = a.twiddle + a.tw_off[bh];
    const double* tx = a.twiddle + a.tw_off[bw];
    for (int i = tid; i < bh; i += kPBlock) { cy[i] = ty[2 * i]; sy[i] = ty[2 * i + 1]; }
    for (int i = tid; i < bw; i += kPBlock) { cx[i] = tx[2 * i]; sx[i] = tx[2 * i + 1]; }
  }

  // ---- spectral amplitude parameters (MCMC.py:209-239) ----------------------------------------
  double lx, ly;
  if (P.model == GSM_MODEL_GAUSSIAN) { lx = range_x / sqrt(3.0); ly = range_y / sqrt(3.0); }
  else if (P.model == GSM_MODEL_EXPONENTIAL) { lx = range_x / 3.0; ly = range_y / 3.0; }
  else { lx = range_x / 2.0; ly = range_y / 2.0; }
  const double aa = sqrt(lx * ly);
  const double nu = (P.smoothness != 0.0) ? P.smoothness : 1.0;
  double m_const = 0.0, m_kappa = 0.0;
  if (P.model == GSM_MODEL_MATERN) {
    m_const = (4.0 * M_PI * tgamma(nu + 1.0) * pow(2.0 * nu, nu)) / (tgamma(nu) * pow(aa, 2.0 * nu));
    m_kappa = 2.0 * nu / (aa * aa);
  }

  // ---- Hermitian half-plane coefficients --------------------------------------------------------
  const int nhalf = bh * ncol;
  const uint32_t m_ncol = pmagic((uint32_t)ncol);
  for (int i = tid; i < nhalf; i += kPBlock) {
    const int ky = (int)__umulhi((uint32_t)i, m_ncol);
    const int kx = i - ky * ncol;
    const double kxv = wavenumber(kx, bw, P.resolution), kyv = wavenumber(ky, bh, P.resolution);
    const double k = sqrt(kxv * kxv + kyv * kyv) + 1e-10;
    double Sp;
    if (P.model == GSM_MODEL_GAUSSIAN) { const double ak = aa * k; Sp = exp(-0.5 * (ak * ak)); }
    else if (P.model == GSM_MODEL_EXPONENTIAL) { const double ak = aa * k; Sp = 1.0 / pow(1.0 + ak * ak, 1.5); }
    else Sp = m_const * pow(m_kappa + 4.0 * M_PI * (k * k), -nu - 1.0);
    const double amp = sqrt(Sp);
    double g1, g2;
    normals2(seed, step, kStreamSpectrum, (uint32_t)i, g1, g2);
    double xr, xi;
    if (kx > 0 && kx < bw / 2) {
      xr = amp * (g1 * M_SQRT1_2);
      xi = amp * (g2 * M_SQRT1_2);
    } else {
      const int kyc = (ky == 0) ? 0 : bh - ky;
      double h1, h2;
      normals2(seed, step, kStreamSpectrum, (uint32_t)(kyc * ncol + kx), h1, h2);
      xr = amp * (0.5 * (g1 + h1));
      xi = amp * (0.5 * (g2 - h2));
    }
    Xr[i] = xr;
    Xi[i] = xi;
  }
  __syncthreads();

  // ---- stage 1: T[y, kx] = sum_ky X[ky, kx] * exp(+2 pi i ky y / bh) -----------------------------
  for (int o = tid; o < nhalf; o += kPBlock) {
    const int y = (int)__umulhi((uint32_t)o, m_ncol);
    const int kx = o - y * ncol;
    double tr = 0.0, ti = 0.0;
    int m = 0;
    for (int ky = 0; ky < bh; ++ky) {
      const double c = cy[m], sn = sy[m];
      const double xr = Xr[ky * ncol + kx], xi = Xi[ky * ncol + kx];
      tr += xr * c - xi * sn;
      ti += xr * sn + xi * c;
      m += y;
      if (m >= bh) m -= bh;
    }
    Tr[o] = tr;
    Ti[o] = ti;
  }
  __syncthreads();

  // ---- stage 2: field[y, x] = (T[y,0] + (-1)^x T[y,bw/2] + 2 Re sum_kx T[y,kx] e^{2 pi i kx x/bw}) / (bh bw)
  const int ncell = bh * bw;
  const uint32_t m_bw = pmagic((uint32_t)bw);
  const double inv_n = 1.0 / (double)ncell;
  for (int o = tid; o < ncell; o += kPBlock) {
    const int y = (int)__umulhi((uint32_t)o, m_bw);
    const int x = o - y * bw;
    const double* trow = Tr + y * ncol;
    const double* tirow = Ti + y * ncol;
    double acc = 0.0;
    int m = x;
    for (int kx = 1; kx < bw / 2; ++kx) {
      acc += trow[kx] * cx[m] - tirow[kx] * sx[m];
      m += x;
      if (m >= bw) m -= bw;
    }
    const double edge = trow[0] + ((x & 1) ? -trow[bw / 2] : trow[bw / 2]);
    fieldv[o] = (edge + 2.0 * acc) * inv_n;
  }
  __syncthreads();

  // ---- standardise (MCMC.py:248), scale, nugget (MCMC.py:251), edge mask (MCMC.py:778) ---------
  double part = 0.0;
  for (int o = tid; o < ncell; o += kPBlock) part += fieldv[o];
  const double mean = block_sum(part, red, tid) / (double)ncell;
  part = 0.0;
  for (int o = tid; o < ncell; o += kPBlock) { const double d = fieldv[o] - mean; part += d * d; }
  const double sd = sqrt(block_sum(part, red, tid) / (double)ncell);
  const double denom = sd + 1e-12;
  const double* mask = a.B.masks + a.B.mask_off[si];
  double* out = a.fields + rec * a.field_stride;
  const double sq_nug = sqrt(nug);
  const bool with_nugget = (P.nugget_max > 0.0);
  for (int o = tid; o < ncell; o += kPBlock) {
    double v = ((fieldv[o] - mean) / denom) * scale;
    if (with_nugget) {
      double n1, n2;
      normals2(seed, step, kStreamNugget, (uint32_t)(o >> 1), n1, n2);
      v = v + ((o & 1) ? n2 : n1) * sq_nug;
    }
    out[o] = v * mask[o];
  }
}

static size_t propose_lds_doubles(const BlockTable& B, int* cap_half_out) {
  const int cap_half = B.max_bh * (B.max_bw / 2 + 1);
  *cap_half_out = cap_half;
  return (size_t)4 * cap_half + 2 * (size_t)B.max_bh + 2 * (size_t)B.max_bw + kPWaves;
}

hipError_t launch_propose(const ProposeArgs& a, hipStream_t st) {
  int cap_half = 0;
  const size_t lds = propose_lds_doubles(a.B, &cap_half) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)propose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(propose_kernel, dim3(a.n_steps, a.n_chains), dim3(kPBlock), lds, st, a, cap_half);
  return hipGetLastError();
}

// host-visible self test of the Philox implementation (used by the C ABI below)
}  // namespace gsm

extern "C" int gsm_philox_selftest(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
  gsm::u32x4 c{ctr4[0], ctr4[1], ctr4[2], ctr4[3]};
  const gsm::u32x4 r = gsm::philox4x32_10(c, key2[0], key2[1]);
  out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
  return 0;
}
