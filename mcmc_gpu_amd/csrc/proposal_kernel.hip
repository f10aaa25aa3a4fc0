// Philox proposal generator for gfx950 (MI355X): one workgroup per (chain, step) proposal.
//
// Replaces RandField.get_rfblock + spectral_synthesis_field (reference gstatsMCMC/MCMC.py:742-778,
// :176-254), the block-centre rejection loop (MCMC.py:1253-1261) and rng.random() (MCMC.py:1336) with
// a counter-based generator.  The draws are different numbers than NumPy's PCG64 stream gives, so this
// path is checked (a) value-for-value against the CPU restatement of THIS algorithm
// (oracle/philox_oracle.py, <=1e-10 of the field scale) and (b) in distribution against the reference's
// spectral proposal (covariance, accept rate).
//
// Same distribution, half the work.  The reference takes Re(ifft2((N1 + i N2) * sqrt(S))) with two
// full planes of normals.  The real part of an inverse DFT is the inverse DFT of the Hermitian part of
// the spectrum, Zh[k] = sqrt(S[k]) * ((N1[k] + N1[-k])/2 + i (N2[k] - N2[-k])/2).  Its entries are
// independent over {k, -k} pairs with variance 1/2 per component (variance 1, real, where k == -k), so
// the kernel draws the half plane kx in [0, bw/2] directly -- bh*bw normals instead of 2*bh*bw -- and
// runs a complex-to-real inverse DFT: columns first (complex, bw/2+1 of them), then rows (real output).
//
// The two DFT stages are dense fp64 matrix products on the matrix cores (v_mfma_f64_16x16x4_f64):
//   stage 1   T^T[kx][y]  = sum_ky X[ky][kx] * exp(+2 pi i ky y / bh)      A = folded X^T (LDS), B = cos/sin table (L2)
//   stage 2   field[y][x] = sum_k (Tr^T[k][y] Gc[k][x] + Ti^T[k][y] Gs[k][x])  A = T^T (LDS), B = G tables (L2)
// with Gc = c_k cos(2 pi k x / bw), Gs = -c_k sin(...) (c_k = 1 for k in {0, bw/2}, else 2: the Hermitian half), both
// folded by the even/odd symmetry of cos/sin so that only indices <= n/2 enter the products (4x / 2x fewer flops than
// the dense DFT; see propose_kernel).  The field never touches LDS: it stays in the MFMA accumulators through the
// standardisation and goes straight to HBM.
//
// MFMA operand maps used here (16x16x4 f64): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C/D reg q: row = (lane >> 4) + 4 q, col = lane & 15.

#include "gsm_internal.h"
#include "philox.h"
#include <math.h>
#include <stdlib.h>

namespace gsm {

constexpr int kPBlock = 512;
constexpr int kPWaves = kPBlock / 64;
constexpr int kMaxT1 = 2;   // stage-1 output tiles per wave
constexpr int kMaxT2 = 2;   // stage-2 output tiles per wave
typedef double v4f64 __attribute__((ext_vector_type(4)));

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

// 2*pi*fftfreq(n, d=res)[k]
__device__ __forceinline__ double wavenumber(int k, int n, double res) {
  const int kk = (k < (n + 1) / 2) ? k : k - n;  // numpy fftfreq ordering (n even: k=n/2 -> -n/2)
  return ((double)kk / ((double)n * res)) * 2.0 * M_PI;
}

// ---------------------------------------------------------------------------------------------------
// per-proposal scalars: one thread per (chain, step).  Draw layout (stream kStreamScalars):
//   idx 0: {scale u, nugget u}   idx 1: {range_x u, range_y u}   idx 2: {accept u, centre word}   idx 3: {size word}
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void propose_scalars_kernel(const ProposeArgs a) {
  const int64_t rec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (rec >= (int64_t)a.n_chains * a.n_steps) return;
  const int chain = (int)(rec / a.n_steps), s = (int)(rec - (int64_t)chain * a.n_steps);
  const int64_t step = a.step0 + s;
  const uint64_t seed = a.seeds[chain];
  const gsm_rf_params& P = a.rf;
  const u32x4 d0 = philox_draw(seed, step, kStreamScalars, 0);
  const u32x4 d1 = philox_draw(seed, step, kStreamScalars, 1);
  const u32x4 d2 = philox_draw(seed, step, kStreamScalars, 2);
  const u32x4 d3 = philox_draw(seed, step, kStreamScalars, 3);
  PropScalars r;
  r.si = (int)__umulhi(d3.x, (uint32_t)a.B.n_sizes);
  r.scale = (P.scale_min + (P.scale_max - P.scale_min) * u01_from(d0.x, d0.y)) / 3.0;
  r.nug = 0.0 + (P.nugget_max - 0.0) * u01_from(d0.z, d0.w);
  r.range_x = P.range_min_x + (P.range_max_x - P.range_min_x) * u01_from(d1.x, d1.y);
  r.range_y = P.isotropic ? r.range_x : P.range_min_y + (P.range_max_y - P.range_min_y) * u01_from(d1.z, d1.w);
  r.u = u01_from(d2.x, d2.y);
  const uint64_t cw = ((uint64_t)d2.w << 32) | d2.z;
  const int cell = a.centres[(int)__umul64hi(cw, (uint64_t)a.n_centres)];
  r.row = cell / a.W;
  r.col = cell - r.row * a.W;
  // spectral amplitude parameters (MCMC.py:209-239)
  double lx, ly;
  if (P.model == GSM_MODEL_GAUSSIAN) { lx = r.range_x / sqrt(3.0); ly = r.range_y / sqrt(3.0); }
  else if (P.model == GSM_MODEL_EXPONENTIAL) { lx = r.range_x / 3.0; ly = r.range_y / 3.0; }
  else { lx = r.range_x / 2.0; ly = r.range_y / 2.0; }
  r.aa = sqrt(lx * ly);
  r.m_const = 0.0; r.m_kappa = 0.0;
  if (P.model == GSM_MODEL_MATERN) {
    const double nu = (P.smoothness != 0.0) ? P.smoothness : 1.0;
    r.m_const = (4.0 * M_PI * tgamma(nu + 1.0) * pow(2.0 * nu, nu)) / (tgamma(nu) * pow(r.aa, 2.0 * nu));
    r.m_kappa = 2.0 * nu / (r.aa * r.aa);
  }
  r.pad = 0;
  r.bh = a.B.bh[r.si];
  r.bw = a.B.bw[r.si];
  r.fy_off = a.fy_off[r.bh];
  r.g_off = a.g_off[r.bw];
  r.mask_off = a.B.mask_off[r.si];
  a.scalars[rec] = r;
  a.size_idx[rec] = r.si;
  a.centre[2 * rec] = r.row;
  a.centre[2 * rec + 1] = r.col;
  a.u[rec] = r.u;
  if (a.rf_scalars) {
    a.rf_scalars[4 * rec] = r.scale;
    a.rf_scalars[4 * rec + 1] = r.nug;
    a.rf_scalars[4 * rec + 2] = r.range_x;
    a.rf_scalars[4 * rec + 3] = r.range_y;
  }
}

// sqrt(S(k)) of MCMC.py:227-239, :244
__device__ __forceinline__ double spectral_amp(const gsm_rf_params& P, const PropScalars& sc, int ky, int kx, int bh, int bw) {
  const double kxv = wavenumber(kx, bw, P.resolution), kyv = wavenumber(ky, bh, P.resolution);
  const double k = sqrt(kxv * kxv + kyv * kyv) + 1e-10;
  double Sp;
  if (P.model == GSM_MODEL_GAUSSIAN) { const double ak = sc.aa * k; Sp = exp(-0.5 * (ak * ak)); }
  else if (P.model == GSM_MODEL_EXPONENTIAL) { const double ak = sc.aa * k; Sp = exp(-1.5 * log(1.0 + ak * ak)); }
  else {
    const double nu = (P.smoothness != 0.0) ? P.smoothness : 1.0;
    Sp = sc.m_const * exp((-nu - 1.0) * log(sc.m_kappa + 4.0 * M_PI * (k * k)));
  }
  return sqrt(Sp);
}

// DFT folding used below (n even, h = n/2).  With P[k] = X[k] + X[n-k], M[k] = X[k] - X[n-k] (0 < k < h; P = X, M = 0
// for k in {0, h}):   sum_k X[k] e^{+i t k y} = U[y] + i V[y],  U = sum_{k<=h} P[k] cos(t k y),  V = sum_{k<h} M[k] sin(t k y)
// and the mirrored output is  U[y] - i V[y]  at n - y.  Only k, y in [0, h] enter the products: 4x fewer flops than the
// dense complex DFT.  The real (c2r) stage folds the same way in x: field[y][x] = E + O, field[y][bw - x] = E - O.
__global__ __launch_bounds__(kPBlock, 2) void propose_kernel(const ProposeArgs a) {
  extern __shared__ double plds[];
  const int SX = a.lds_sx, ST = a.lds_st;
  double* Pr = plds;                       // 4 planes [KRmax][SX]: P re, P im, M re, M im
  double* Pi = Pr + a.lds_x_half;
  double* Mr = Pi + a.lds_x_half;
  double* Mi = Mr + a.lds_x_half;
  double* TT = plds;                       // [2 Kc][ST]  -- overlays the planes once stage 1 has consumed them
  double* red = plds + a.lds_main;         // [kPWaves]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int s = blockIdx.x, chain = blockIdx.y;
  const int64_t step = a.step0 + s;
  const uint64_t seed = a.seeds[chain];
  const int64_t rec = (int64_t)chain * a.n_steps + s;
  const gsm_rf_params& P = a.rf;
  const PropScalars sc = a.scalars[rec];
  const int bh = sc.bh, bw = sc.bw;
  const int hh = bh / 2, hw = bw / 2;
  const int ncol = hw + 1, nrow = hh + 1;
  if (a.dbg & 64) { if (tid == 0) a.fields[rec * a.field_stride] = (double)bh; return; }

  // padded GEMM dimensions (host builds the tables with the same formulas)
  const int KR = (nrow + 3) & ~3;          // stage-1 K  (ky <= hh)
  const int NR = (nrow + 15) & ~15;        // stage-1 N  (y  <= hh)
  const int M1 = (ncol + 15) & ~15;        // stage-1 M  (kx)  = stage-2 N (x <= hw)
  const int Kc = (ncol + 3) & ~3;          // stage-2 K per half (re | im rows of T^T)
  const int N1 = (bh + 15) & ~15;          // stage-2 M  (y)

  // ---- folded Hermitian half-plane coefficients -> LDS -------------------------------------------
  // one work item per (ky <= hh, kx): rows ky and bh-ky share the spectral amplitude, and on the two self-conjugate
  // columns they are a conjugate pair built from the same two draws.
  {
    const int npad = KR * M1;
    const uint32_t m_m1 = pmagic((uint32_t)M1);
    for (int i = tid; i < npad; i += kPBlock) {
      const int ky = (int)__umulhi((uint32_t)i, m_m1);
      const int kx = i - ky * M1;
      if (ky >= nrow || kx >= ncol) {
        const int o = ky * SX + kx;
        Pr[o] = 0.0; Pi[o] = 0.0; Mr[o] = 0.0; Mi[o] = 0.0;
      }
    }
    const int nitem = nrow * ncol;
    const uint32_t m_nc = pmagic((uint32_t)ncol);
    for (int i = tid; i < nitem && !(a.dbg & 32); i += kPBlock) {
      const int ky = (int)__umulhi((uint32_t)i, m_nc);
      const int kx = i - ky * ncol;
      const int kyc = bh - ky;
      const bool paired = (ky != 0) && (ky != hh);
      double amp, g1, g2, h1 = 0.0, h2 = 0.0;
      if (a.dbg & 1) { amp = 1.0; g1 = ky; g2 = kx; h1 = 1.0; h2 = 2.0; }
      else {
        amp = spectral_amp(P, sc, ky, kx, bh, bw);
        normals2(seed, step, kStreamSpectrum, (uint32_t)(ky * ncol + kx), g1, g2);
        if (paired) normals2(seed, step, kStreamSpectrum, (uint32_t)(kyc * ncol + kx), h1, h2);
      }
      double ar, ai, br = 0.0, bi = 0.0;   // X[ky], X[bh - ky]
      if (kx > 0 && kx < hw) {
        ar = amp * (g1 * M_SQRT1_2); ai = amp * (g2 * M_SQRT1_2);
        if (paired) { br = amp * (h1 * M_SQRT1_2); bi = amp * (h2 * M_SQRT1_2); }
      } else if (paired) {
        ar = amp * (0.5 * (g1 + h1)); ai = amp * (0.5 * (g2 - h2));
        br = amp * (0.5 * (h1 + g1)); bi = amp * (0.5 * (h2 - g2));
      } else {
        ar = amp * (0.5 * (g1 + g1)); ai = amp * (0.5 * (g2 - g2));
      }
      const int o = ky * SX + kx;
      Pr[o] = ar + br; Pi[o] = ai + bi;
      Mr[o] = paired ? ar - br : 0.0;
      Mi[o] = paired ? ai - bi : 0.0;
    }
  }
  __syncthreads();

  const int l15 = lane & 15, l4 = lane >> 4;

  // ---- stage 1 (MFMA): U = P^T C, V = M^T S on ky, y in [0, hh] -----------------------------------
  // results wait in registers until every wave has finished reading the planes, then overwrite them as T^T
  v4f64 ur[kMaxT1], ui[kMaxT1], vr[kMaxT1], vi[kMaxT1];
  const int n_mt = M1 >> 4, n_nt = NR >> 4;
  const int n_t1 = n_mt * n_nt;
  {
    const double* __restrict__ FC = a.tables + sc.fy_off;      // [KR][NR]
    const double* __restrict__ FS = FC + KR * NR;
#pragma unroll
    for (int j = 0; j < kMaxT1; ++j) {
      v4f64 aur = {0.0, 0.0, 0.0, 0.0}, aui = aur, avr = aur, avi = aur;
      const int t = wave + j * kPWaves;
      if (t < n_t1 && !(a.dbg & 2)) {
        const int mt = t % n_mt, nt = t / n_mt;
        const int ao = l4 * SX + 16 * mt + l15;
        const double* fc_p = FC + l4 * NR + 16 * nt + l15;
        const double* fs_p = FS + l4 * NR + 16 * nt + l15;
#pragma unroll 2
        for (int k0 = 0; k0 < KR; k0 += 4) {
          const double bc = fc_p[k0 * NR], bs = fs_p[k0 * NR];
          const int o = ao + k0 * SX;
          aur = __builtin_amdgcn_mfma_f64_16x16x4f64(Pr[o], bc, aur, 0, 0, 0);
          aui = __builtin_amdgcn_mfma_f64_16x16x4f64(Pi[o], bc, aui, 0, 0, 0);
          avr = __builtin_amdgcn_mfma_f64_16x16x4f64(Mr[o], bs, avr, 0, 0, 0);
          avi = __builtin_amdgcn_mfma_f64_16x16x4f64(Mi[o], bs, avi, 0, 0, 0);
        }
      }
      ur[j] = aur; ui[j] = aui; vr[j] = avr; vi[j] = avi;
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kMaxT1; ++j) {
    const int t = wave + j * kPWaves;
    if (t < n_t1) {
      const int mt = t % n_mt, nt = t / n_mt;
      const int y = 16 * nt + l15;
      if (y <= hh) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kx = 16 * mt + l4 + 4 * q;
          if (kx < Kc) {
            TT[kx * ST + y] = ur[j][q] - vi[j][q];
            TT[(Kc + kx) * ST + y] = ui[j][q] + vr[j][q];
            if (y > 0 && y < hh) {
              TT[kx * ST + (bh - y)] = ur[j][q] + vi[j][q];
              TT[(Kc + kx) * ST + (bh - y)] = ui[j][q] - vr[j][q];
            }
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- stage 2 (MFMA): E = Tr^T Gc, O = Ti^T Gs on x in [0, hw]; results stay in registers -----------
  v4f64 fe[kMaxT2], fo[kMaxT2];
  const int n_mt2 = N1 >> 4, n_nt2 = M1 >> 4;
  const int n_t2 = n_mt2 * n_nt2;
  {
    const double* __restrict__ GC = a.tables + sc.g_off;       // [Kc][M1]
    const double* __restrict__ GS = GC + Kc * M1;
#pragma unroll
    for (int j = 0; j < kMaxT2; ++j) {
      v4f64 ae = {0.0, 0.0, 0.0, 0.0}, ao = ae;
      const int t = wave + j * kPWaves;
      if (t < n_t2 && !(a.dbg & 4)) {
        const int mt = t % n_mt2, nt = t / n_mt2;
        const double* a_p = TT + l4 * ST + 16 * mt + l15;
        const double* gc_p = GC + l4 * M1 + 16 * nt + l15;
        const double* gs_p = GS + l4 * M1 + 16 * nt + l15;
#pragma unroll 2
        for (int k0 = 0; k0 < Kc; k0 += 4) {
          ae = __builtin_amdgcn_mfma_f64_16x16x4f64(a_p[k0 * ST], gc_p[k0 * M1], ae, 0, 0, 0);
          ao = __builtin_amdgcn_mfma_f64_16x16x4f64(a_p[(Kc + k0) * ST], gs_p[k0 * M1], ao, 0, 0, 0);
        }
      }
      fe[j] = ae; fo[j] = ao;
    }
  }

  // ---- standardise (MCMC.py:248) on the register-resident field ---------------------------------
  // lane holds, per (tile j, reg q): v1 = field[y][x] = E + O and, for 0 < x < hw, v2 = field[y][bw - x] = E - O
  const int ncell = bh * bw;
  const double inv_n = 1.0 / (double)ncell;
  double part = 0.0;
#pragma unroll
  for (int j = 0; j < kMaxT2; ++j) {
    const int t = wave + j * kPWaves;
    const int mt = t % n_mt2, nt = t / n_mt2;
    const int x = 16 * nt + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 16 * mt + l4 + 4 * q;
      const bool ok = (t < n_t2) && (y < bh) && (x <= hw);
      const bool two = ok && (x > 0) && (x < hw);
      const double e = fe[j][q], o = fo[j][q];
      const double v1 = ok ? (e + o) * inv_n : 0.0;
      const double v2 = two ? (e - o) * inv_n : 0.0;
      fe[j][q] = v1; fo[j][q] = v2;
      part += v1 + v2;
    }
  }
  const double mean = (a.dbg & 16) ? part : block_sum(part, red, tid) * inv_n;
  part = 0.0;
#pragma unroll
  for (int j = 0; j < kMaxT2; ++j) {
    const int t = wave + j * kPWaves;
    const int mt = t % n_mt2, nt = t / n_mt2;
    const int x = 16 * nt + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 16 * mt + l4 + 4 * q;
      const bool ok = (t < n_t2) && (y < bh) && (x <= hw);
      if (ok) { const double d = fe[j][q] - mean; part += d * d; }
      if (ok && x > 0 && x < hw) { const double d = fo[j][q] - mean; part += d * d; }
    }
  }
  const double sd = (a.dbg & 16) ? part : sqrt(block_sum(part, red, tid) * inv_n);
  const double gain = sc.scale / (sd + 1e-12);

  // ---- scale, nugget (MCMC.py:251), edge mask (MCMC.py:778), store ------------------------------
  const double* __restrict__ mask = a.B.masks + sc.mask_off;
  double* __restrict__ out = a.fields + rec * a.field_stride;
  const double sq_nug = sqrt(sc.nug);
  const bool with_nugget = (P.nugget_max > 0.0);
#pragma unroll
  for (int j = 0; j < kMaxT2; ++j) {
    const int t = wave + j * kPWaves;
    const int mt = t % n_mt2, nt = t / n_mt2;
    const int x = 16 * nt + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 16 * mt + l4 + 4 * q;
      if ((t < n_t2) && (y < bh) && (x <= hw) && !(a.dbg & 8)) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (half == 1 && !(x > 0 && x < hw)) continue;
          const int o = y * bw + (half ? bw - x : x);
          double v = ((half ? fo[j][q] : fe[j][q]) - mean) * gain;
          if (with_nugget) {
            double n1, n2;
            normals2(seed, step, kStreamNugget, (uint32_t)(o >> 1), n1, n2);
            v = v + ((o & 1) ? n2 : n1) * sq_nug;
          }
          out[o] = v * mask[o];
        }
      }
    }
  }
}

hipError_t launch_propose(const ProposeArgs& a_in, hipStream_t st) {
  ProposeArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("GSM_PROPOSE_DBG"); dbg = v ? atoi(v) : 0; } a.dbg = dbg; }
  const size_t lds = ((size_t)a.lds_main + kPWaves) * sizeof(double);
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)propose_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int64_t nrec = (int64_t)a.n_chains * a.n_steps;
  hipLaunchKernelGGL(propose_scalars_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, a);
  hipLaunchKernelGGL(propose_kernel, dim3(a.n_steps, a.n_chains), dim3(kPBlock), lds, st, a);
  return hipGetLastError();
}

int propose_max_tiles1_per_wave() { return kMaxT1; }
int propose_max_tiles_per_wave() { return kMaxT2; }
int propose_waves() { return kPWaves; }

}  // namespace gsm

// host-visible self test of the Philox implementation (declared in include/gsm.h)
extern "C" int gsm_philox_selftest(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
  gsm::u32x4 c{ctr4[0], ctr4[1], ctr4[2], ctr4[3]};
  const gsm::u32x4 r = gsm::philox4x32_10(c, key2[0], key2[1]);
  out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
  return 0;
}
