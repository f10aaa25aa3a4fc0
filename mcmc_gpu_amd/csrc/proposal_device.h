// Spectral proposal field of one (chain, step), as a sequence of stage functions shared by the stand-alone proposal kernel
// (512 threads, two workgroups per CU) and the fused chain kernel.  See proposal_kernel.hip for the algorithm
// (reference gstatsMCMC/MCMC.py:742-778, :176-254).
#pragma once
#ifndef PSTAMP
#define PSTAMP(slot) do {} while (0)
#endif
#include "gsm_internal.h"
#include "device_util.h"
#include "philox.h"
#include "math_tables.h"
#include <math.h>

namespace gsm {

typedef double v4f64 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t pmagic(uint32_t d) { return (uint32_t)(0xFFFFFFFFu / d) + 1u; }

// a * b + C and a * C with the constant C read from a scalar register pair.  Written as inline assembly because the
// compiler otherwise copies every fp64 literal into a VGPR pair first (two v_mov_b32 per constant, re-done at every use
// since machine LICM is off for this file): with ~35 polynomial constants per work item that was a tenth of the
// coefficient phase's vector instructions.  s_mov_b32 on the scalar unit issues beside another wave's vector instruction.
__device__ __forceinline__ double fma_sc(double a, double b, double C) {
  double r;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(C));
  return r;
}
__device__ __forceinline__ double mul_sc(double a, double C) {
  double r;
  asm("v_mul_f64 %0, %1, %2" : "=v"(r) : "v"(a), "s"(C));
  return r;
}

// exp(x) for finite x, |x| < 700 (spectral amplitudes): k = rint(x / ln 2), r = x - k ln2_hi - k ln2_lo (|r| <= 0.3466),
// degree-12 Taylor polynomial of exp(r) (truncation 1.7e-16 relative), scaled by 2^k with v_ldexp_f64 (which also takes
// care of underflow towards 0).  About half the instructions of the library routine; error < 1.5 ulp.
__device__ __forceinline__ double exp_lean(double x) {
  const double kf = __builtin_rint(mul_sc(x, 1.44269504088896338700e+00));
  const double r = fma_sc(kf, -1.90821492927058770002e-10, fma_sc(kf, -6.93147180369123816490e-01, x));
  double p = fma_sc(r, 2.08767569878680989792e-09, 2.50521083854417187751e-08);      // 1/12!, 1/11!
  p = fma_sc(r, p, 2.75573192239858906526e-07);     // 1/10!
  p = fma_sc(r, p, 2.75573192239858906526e-06);     // 1/9!
  p = fma_sc(r, p, 2.48015873015873015873e-05);     // 1/8!
  p = fma_sc(r, p, 1.98412698412698412698e-04);     // 1/7!
  p = fma_sc(r, p, 1.38888888888888888889e-03);     // 1/6!
  p = fma_sc(r, p, 8.33333333333333333333e-03);     // 1/5!
  p = fma_sc(r, p, 4.16666666666666666667e-02);     // 1/4!
  p = fma_sc(r, p, 1.66666666666666666667e-01);     // 1/3!
  p = fma_sc(r, p, 0.5);
  p = __fma_rn(r, p, 1.0);
  p = __fma_rn(r, p, 1.0);
  return ldexp(p, (int)kf);
}

// sqrt(t) for 0 <= t < 2^500 (here t = -2 log u <= 73.5): v_rsq_f64, one Goldschmidt step and one residual correction (error
// < 1 ulp) -- the library routine's range scaling, special-case selects and second correction left out.  t = 0 (u = 1,
// probability 2^-53) is raised to 1e-300: the result 1e-150 stands for 0.
__device__ __forceinline__ double sqrt_lean(double t) {
  t = fmax(t, 1e-300);
  const double y = __builtin_amdgcn_rsq(t);
  double g = t * y, h = 0.5 * y;
  const double r = __fma_rn(-h, g, 0.5);
  g = __fma_rn(g, r, g);
  h = __fma_rn(h, r, h);
  return __fma_rn(__fma_rn(-g, g, t), h, g);
}

// mt: the table of math_tables.h, in LDS
__device__ __forceinline__ void normals2(uint64_t seed, int64_t step, uint32_t stream, uint32_t idx, double& g1,
                                         double& g2, const double* mt) {
  // The 20 round keys are uniform functions of the seed: left alone, the compiler computes them once per kernel and
  // holds 20 SGPRs for ever (spilled to VGPR lanes and read back with v_readlane, a vector instruction, at every use).
  // Laundering the seed here makes them 20 scalar adds per call instead.
  uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFu), k1 = (uint32_t)(seed >> 32);
  asm volatile("" : "+s"(k0), "+s"(k1));
  const u32x4 r = philox_draw(((uint64_t)k1 << 32) | k0, step, stream, idx);
  const double u1 = u01_open0_from(r.x, r.y);
  const double u2 = u01_from(r.z, r.w);
  const double rad = sqrt_lean(-2.0 * log_tab(u1, mt));
  double s, c;
  sincos_tab(u2, mt, s, c);
  g1 = rad * c;
  g2 = rad * s;
}

// Four standard normals from ONE Philox block, a 32-bit word per uniform: (x, y) -> (g1, g2), (z, w) -> (h1, h2).  The spectrum's
// coefficients are drawn this way (coef_items: a work item needs four normals, for rows ky and bh - ky): the Philox rounds were a
// quarter of the coefficient phase's instructions and the phase 28 % of the fused kernel's.  2^-32 steps in the uniforms bound
// |normal| by sqrt(64 ln 2) = 6.66 (probability 2.7e-11 per draw beyond it with exact normals) -- each normal is one of thousands of
// coefficients of a field that is standardised afterwards.  Restated by oracle/philox_oracle.normals4.
__device__ __forceinline__ void normals4(uint64_t seed, int64_t step, uint32_t stream, uint32_t idx, double& g1, double& g2,
                                         double& h1, double& h2, const double* mt) {
  uint32_t k0 = (uint32_t)(seed & 0xFFFFFFFFu), k1 = (uint32_t)(seed >> 32);
  asm volatile("" : "+s"(k0), "+s"(k1));                       // see normals2
  const u32x4 r = philox_draw(((uint64_t)k1 << 32) | k0, step, stream, idx);
  const double ra = sqrt_lean(-2.0 * log_tab(u01_open0_from32(r.x), mt));
  const double rb = sqrt_lean(-2.0 * log_tab(u01_open0_from32(r.z), mt));
  double s, c;
  sincos_tab(u01_from32(r.y), mt, s, c);
  g1 = ra * c; g2 = ra * s;
  sincos_tab(u01_from32(r.w), mt, s, c);
  h1 = rb * c; h2 = rb * s;
}

// normals2 for per-LANE seeds (the Cholesky generator's z: every lane of a wave serves another chain)
__device__ __forceinline__ void normals2_key(uint64_t seed, int64_t step, uint32_t stream, uint32_t idx, double& g1, double& g2,
                                             const double* mt) {
  const u32x4 r = philox_draw(seed, step, stream, idx);
  const double u1 = u01_open0_from(r.x, r.y);
  const double u2 = u01_from(r.z, r.w);
  const double rad = sqrt_lean(-2.0 * log_tab(u1, mt));
  double s, c;
  sincos_tab(u2, mt, s, c);
  g1 = rad * c;
  g2 = rad * s;
}

// 2*pi*fftfreq(n, d=res)[k] with inv = 1 / (n * res): numpy.fft.fftfreq multiplies the integer frequency by that
// reciprocal too (MCMC.py:221-222), so this is the reference's value and the division happens once per proposal
__device__ __forceinline__ double wavenumber(int k, int n, double inv) {
  const int kk = (k < (n + 1) / 2) ? k : k - n;  // numpy fftfreq ordering (n even: k=n/2 -> -n/2)
  return ((double)kk * inv) * 2.0 * M_PI;
}

// sqrt(S(k)) of MCMC.py:227-239, :244.  k2 = (sqrt(kx^2 + ky^2) + 1e-10)^2 comes from a per-shape table (it depends on
// nothing that is drawn, gsm_api.hip: k2_table_kernel); the outer square root is taken in the exponent:
//   Gaussian     sqrt(exp(-(a k)^2 / 2))            = exp(-(a k)^2 / 4)
//   Exponential  sqrt((1 + (a k)^2)^-1.5)           = exp(-0.75 log(1 + (a k)^2))
//   Matern       sqrt(C (kappa + 4 pi k^2)^(-nu-1)) = exp(log(C) / 2 - (nu + 1) / 2 * log(kappa + 4 pi k^2))
// sc.m_const holds log(C) / 2 (propose_scalars_kernel).  Same values as the reference's formula to a few ulp.
__device__ __forceinline__ double spectral_amp(const gsm_rf_params& P, const PropScalars& sc, const double k2, const double* mt) {
  if (P.model == GSM_MODEL_GAUSSIAN) return exp_lean(-0.25 * ((sc.aa * sc.aa) * k2));
  if (P.model == GSM_MODEL_EXPONENTIAL) return exp_lean(-0.75 * log_tab(1.0 + (sc.aa * sc.aa) * k2, mt));
  const double nu = (P.smoothness != 0.0) ? P.smoothness : 1.0;
  return exp_lean(__fma_rn(-0.5 * (nu + 1.0), log_tab(sc.m_kappa + 4.0 * M_PI * k2, mt), sc.m_const));
}

// DFT folding used below (n even, h = n/2).  With P[k] = X[k] + X[n-k], M[k] = X[k] - X[n-k] (0 < k < h; P = X, M = 0
// for k in {0, h}):   sum_k X[k] e^{+i t k y} = U[y] + i V[y],  U = sum_{k<=h} P[k] cos(t k y),  V = sum_{k<h} M[k] sin(t k y)
// and the mirrored output is  U[y] - i V[y]  at n - y.  Only k, y in [0, h] enter the products: 4x fewer flops than the
// dense complex DFT.  The real (c2r) stage folds the same way in x: field[y][x] = E + O, field[y][bw - x] = E - O.
//
// The proposal is a sequence of stages, each a function below; a stage is executed by NW waves of the workgroup and `w`
// is the calling wave's index among them (tile / unit ownership: unit u belongs to wave u mod NW).  Callers:
//   propose_field<NT>        all NT / 64 waves run every stage in turn (stand-alone proposal kernel, gsm_spectral_from_noise)
//   chain_fused_kernel       all 16 waves run every stage in turn, with the DFT tables staged in LDS (chain_fused_kernel.hip)
// The arithmetic of every field value -- including the order of the two reductions of the standardisation -- does not
// depend on NW, so all forms produce bit-identical fields.
//
// LDS: four coefficient planes [KRmax][SX] (P re, P im, M re, M im), overlaid by T^T [2 Kc][ST] once stage 1 has consumed
// them; `red`: 32 doubles.  TABLDS: the DFT operand tables of this block shape are staged in LDS -- tabA ([cos | sin] of the
// block height, 2*KR*NR doubles) must not overlap the planes, tabG (folded c2r table of the block width, 2*Kc*M1 doubles)
// must not overlap planes, T^T or tabA and may overlap the output -- so the MFMA loops read both operands from LDS.

// NOISE_IN (gsm_spectral_from_noise, the value pin against the reference): the coefficients are not drawn but formed from
// caller-supplied white-noise planes N1, N2 of the full (bh, bw) spectrum, X[k] = amp(k) ((N1[k] + N1[-k])/2 +
// i (N2[k] - N2[-k])/2) -- the Hermitian part of the reference's (N1 + i N2) sqrt(S) (MCMC.py:242-247) -- and the nugget
// term is the caller's rng.normal(0, sqrt(nug)) plane.  Everything downstream of the coefficients is the same code.
struct NoiseIn { const double* re; const double* im; const double* nug; };

// shape-derived sizes of one proposal (the host builds the tables with the same formulas)
// t / d for a tile index t < 64 and a tile-grid dimension d (in [1, 8] for every square table) without a division (a uniform integer division costs a
// float reciprocal on the vector unit, v_readfirstlane and ~20 dependent scalar instructions -- at the head of six stages
// of every step): q = (t * (2^15 / d + 1)) >> 15, the eight multipliers packed in two 64-bit constants.
__device__ __forceinline__ uint32_t tile_magic(int d) {
  const uint64_t lo = 32769ull | (16385ull << 16) | (10923ull << 32) | (8193ull << 48);     // d = 1 .. 4
  const uint64_t hi = 6554ull | (5462ull << 16) | (4682ull << 32) | (4097ull << 48);        // d = 5 .. 8
  if (d > 8) return 32768u / (uint32_t)d + 1u;      // elongated block tables only (up to 32 tiles along one side)
  return (uint32_t)(((d <= 4) ? lo : hi) >> (16 * ((d - 1) & 3))) & 0xFFFFu;
}
__device__ __forceinline__ int tile_div(int t, uint32_t magic) { return (int)(((uint32_t)t * magic) >> 15); }

struct PropGeom {
  int bh, bw, hh, hw, ncol, nrow;
  int KR;   // stage-1 K  (ky <= hh), multiple of 4
  int NR;   // stage-1 N  (y  <= hh), multiple of 16
  int M1;   // stage-1 M  (kx)  = stage-2 N (x <= hw), multiple of 16
  int Kc;   // stage-2 K per half (re | im rows of T^T), multiple of 4
  int N1;   // stage-2 M  (y), multiple of 16
  int SX, ST;
  uint32_t q_mt, q_mt2;   // tile_magic of M1 / 16 (stage-1 tile columns) and N1 / 16 (stage-2 tile rows)
  // even block heights (every table of the reference: MCMC.py:576-579 rounds the sizes to even): stage 1 is split by the parity of ky
  int hq;   // hh / 2: the half-sums are evaluated for y in [0, hq]; hh - y mirrors them
  int NRh;  // ... padded to a multiple of 16
  int Ke;   // terms of a half-sum, padded to a multiple of 4
  // stage 2 likewise by the parity of kx -- where the width is even, the handle allows it (ProposeArgs::split2) and the pairs of
  // (x, hw - x) tile slots fit the 16 slots of a workgroup: two slots j, j + 1 of a wave hold x0 = 16 nt + l15 <= hqx and its mirror hw - x0
  bool split2;
  int hqx, M1h, Kce;     // hw / 2, (hqx + 1) padded to 16, terms of a half-sum padded to 4
};
__device__ __forceinline__ PropGeom prop_geom(const ProposeArgs& a, int bh, int bw) {
  PropGeom g;
  g.bh = bh; g.bw = bw; g.hh = bh / 2; g.hw = bw / 2;
  g.ncol = g.hw + 1; g.nrow = g.hh + 1;
  g.KR = (g.nrow + 3) & ~3; g.NR = (g.nrow + 15) & ~15;
  g.M1 = (g.ncol + 15) & ~15; g.Kc = (g.ncol + 3) & ~3;
  g.N1 = (bh + 15) & ~15;
  g.SX = a.lds_sx; g.ST = a.lds_st;
  g.q_mt = tile_magic(g.M1 >> 4); g.q_mt2 = tile_magic(g.N1 >> 4);
  g.hq = g.hh >> 1; g.NRh = (g.hq + 16) & ~15; g.Ke = (g.hq + 4) & ~3;
  g.hqx = g.hw >> 1; g.M1h = (g.hqx + 16) & ~15; g.Kce = (g.hqx + 4) & ~3;
  g.split2 = a.split2 && !(bw & 1) && (g.N1 >> 4) * (g.M1h >> 4) <= 8;
  return g;
}

// ---- folded Hermitian half-plane coefficients ------------------------------------------------------
// One work item per (ky <= hh, kx <= hw): rows ky and bh-ky share the spectral amplitude, and on the two self-conjugate
// columns they are a conjugate pair built from the same two draws.  Thread t of NTH handles items i_lo + t, + NTH, ... below
// i_hi, and -- if `pad` -- the zero padding of the [KR][M1] operand grid.  The four planes may live in LDS or in global
// memory (plane stride `plane`, row stride SX).
// FOLD_CK: the Hermitian-half factor c_kx (1 for kx in {0, bw / 2}, else 2) of the stage-2 operand table is applied here, to the
// spectral amplitude -- one multiply per work item; stage 1 is linear, so T^T comes out scaled exactly as if its rows had been
// doubled (powers of two commute with every rounding) and TABMODE 2 of dft_stage2 reads plain cos / sin.
template <int NTH, bool NOISE_IN, bool FOLD_CK = false>
__device__ __forceinline__ void coef_items(const int t, const int i_lo, const int i_hi, const bool pad, const ProposeArgs& a,
                                           const PropScalars& sc, const PropGeom& g, const uint64_t seed, const int64_t step,
                                           double* __restrict__ Pr, const int plane, const NoiseIn noise, const double* mt,
                                           double* pw_red = nullptr) {
  const gsm_rf_params& P = a.rf;
  double* __restrict__ Pi = Pr + plane;
  double* __restrict__ Mr = Pi + plane;
  double* __restrict__ Mi = Mr + plane;
  const int SX = g.SX, bh = g.bh, bw = g.bw, hh = g.hh, hw = g.hw, ncol = g.ncol, nrow = g.nrow;
  if (pad) {
    // zero padding of the [KR][M1] operand grid around the nrow x ncol coefficients: the (at most 3) rows below them, whole,
    // and the (at most 15) columns to their right -- 16 columns x NTH / 16 rows per pass, no division
    const int M1 = g.M1, KR = g.KR;
    const uint32_t m_m1 = sc.m_m1;
    for (int i = t; i < (KR - nrow) * M1; i += NTH) {
      const int dy = (int)__umulhi((uint32_t)i, m_m1);
      const int o = (nrow + dy) * SX + (i - dy * M1);
      Pr[o] = 0.0; Pi[o] = 0.0; Mr[o] = 0.0; Mi[o] = 0.0;
    }
    const int kx = ncol + (t & 15);
    if (kx < M1) {
      for (int ky = t >> 4; ky < nrow; ky += NTH / 16) {
        const int o = ky * SX + kx;
        Pr[o] = 0.0; Pi[o] = 0.0; Mr[o] = 0.0; Mi[o] = 0.0;
      }
    }
  }
  const uint32_t m_nc = sc.m_nc;
  double pw = 0.0;
  const double* __restrict__ k2tab = a.k2tab + sc.pad;          // this shape's [nrow][ncol] table (pad = its offset)
  for (int i = i_lo + t; i < i_hi && !(a.dbg & 32); i += NTH) {
    const double k2 = k2tab[i];                                 // requested first: lands under the Box-Muller code
    const int ky = (int)__umulhi((uint32_t)i, m_nc);
    const int kx = i - ky * ncol;
    const int kyc = bh - ky;
    const bool paired = (ky != 0) && (ky != hh);
    double amp, g1 = 0.0, g2 = 0.0, h1 = 0.0, h2 = 0.0;
    double ar, ai, br = 0.0, bi = 0.0;   // X[ky], X[bh - ky]
    if (NOISE_IN) {
      amp = spectral_amp(P, sc, k2, mt);
      if (FOLD_CK && kx != 0 && kx != hw) amp = 2.0 * amp;
      const int nky = (ky == 0) ? 0 : kyc, nkx = (kx == 0) ? 0 : bw - kx;      // -k modulo the block shape
      ar = amp * (0.5 * (noise.re[ky * bw + kx] + noise.re[nky * bw + nkx]));
      ai = amp * (0.5 * (noise.im[ky * bw + kx] - noise.im[nky * bw + nkx]));
      if (paired) {                                                            // row bh - ky; its partner row is ky
        br = amp * (0.5 * (noise.re[kyc * bw + kx] + noise.re[ky * bw + nkx]));
        bi = amp * (0.5 * (noise.im[kyc * bw + kx] - noise.im[ky * bw + nkx]));
      }
    } else {
      if (a.dbg & 1) { amp = 1.0; g1 = ky; g2 = kx; h1 = 1.0; h2 = 2.0; }
      else {
        // the four normals of the item -- (g1, g2) for row ky, (h1, h2) for row bh - ky -- from one Philox block at counter
        // ky * ncol + kx; evaluated unconditionally: one straight-line block, so the two Box-Muller chains interleave
        normals4(seed, step, kStreamSpectrum, (uint32_t)(ky * ncol + kx), g1, g2, h1, h2, mt);
        if (!paired) { h1 = 0.0; h2 = 0.0; }
        amp = spectral_amp(P, sc, k2, mt);
      }
      if (FOLD_CK && kx != 0 && kx != hw) amp = 2.0 * amp;
      if (kx > 0 && kx < hw) {
        ar = amp * (g1 * M_SQRT1_2); ai = amp * (g2 * M_SQRT1_2);
        if (paired) { br = amp * (h1 * M_SQRT1_2); bi = amp * (h2 * M_SQRT1_2); }
      } else if (paired) {
        ar = amp * (0.5 * (g1 + h1)); ai = amp * (0.5 * (g2 - h2));
        br = amp * (0.5 * (h1 + g1)); bi = amp * (0.5 * (h2 - g2));
      } else {
        ar = amp * (0.5 * (g1 + g1)); ai = amp * (0.5 * (g2 - g2));
      }
    }
    const int o = ky * SX + kx;
    Pr[o] = ar + br; Pi[o] = ai + bi;
    Mr[o] = paired ? ar - br : 0.0;
    Mi[o] = paired ? ai - bi : 0.0;
    // power of the item's entries of the full Hermitian spectrum, the DC term left out: columns 0 < kx < bw / 2 stand for kx and bw - kx
    // (with FOLD_CK they are stored doubled: |2 X|^2 / 2 = 2 |X|^2 exactly), rows ky and bh - ky are a and b
    if (pw_red && i != 0) {
      const bool inner = (kx != 0) && (kx != hw);          // (callers pass pw_red for even shapes only)
      const double wc = FOLD_CK ? (inner ? 0.5 : 1.0) : (inner ? 2.0 : 1.0);
      pw += wc * ((ar * ar + ai * ai) + (br * br + bi * bi));
    }
  }
  // Parseval: sum over the cells of (n (field - mean))^2 = n * sum_{k != 0} |X_k|^2 -- the standardisation's variance without a pass over the
  // field (standardise).  Partial of each wave -> pw_red[wave]; read after the barrier that ends the coefficient phase.
  if (pw_red) {
    const double ws = dev::wave64_sum(pw);
    if ((t & 63) == 0) pw_red[t >> 6] = ws;
  }
}

// DFT operand tables of this block shape: global -> LDS by LDS-DMA (no registers); whole 1 KiB pieces (128 doubles: the
// table sizes are multiples of 128 doubles), piece c by wave c mod NW of the NW issuing waves.  AUX: cache policy bits.
template <int NW, int AUX = 0>
__device__ __forceinline__ void dma_to_lds(const double* __restrict__ src, double* dst, const int n_doubles, const int w, const int lane) {
  for (int c = w; c < n_doubles / 128; c += NW)
    __builtin_amdgcn_global_load_lds(src + c * 128 + 2 * lane, (__attribute__((address_space(3))) void*)(dst + c * 128), 16, 0, AUX);
}

// ---- stage 1 (MFMA): U = P^T C, V = M^T S on ky, y in [0, hh] -----------------------------------
// Work unit = half an output tile: half 0 accumulates (Ur, Vi) = (Pr C, Mi S), which is all the real part of T^T
// needs; half 1 accumulates (Ui, Vr) = (Pi C, Mr S) for the imaginary part.  Unit u goes to wave u mod NW.  Results wait
// in registers until every wave has finished reading the planes, then overwrite them as T^T (tt_write).
// TABMODE: 0 = the [K][N] operand tables from global memory (L2), 1 = the same tables staged in LDS (tabA),
// 2 = 1-D twiddle tables in LDS (tabA: cos(2 pi m / n) at [m], sin at [kT1S + m], n = block height), the operand of
// (k, j) read at m = (k * j) mod n, advanced from one K step to the next by an add and a conditional subtract -- the
// same numbers at 1 / 30 of the LDS footprint (chain_strip_kernel: two workgroups per CU).
constexpr int kT1S = 128;    // doubles between the cos and the sin half of a 1-D table: block lengths up to 128
__device__ __forceinline__ uint32_t mod_magic(uint32_t x, uint32_t n, uint32_t magic) { return x - __umulhi(x, magic) * n; }

template <int NW, int UPW, int TABMODE>
__device__ __forceinline__ void dft_stage1(const int w, const int lane, const ProposeArgs& a, const PropScalars& sc, const PropGeom& g,
                                           const double* plds, const double* tabA, v4f64 (&uc)[UPW], v4f64 (&us)[UPW]) {
  const int SX = g.SX, KR = g.KR, NR = g.NR;
  const double* Pr = plds;
  const double* Pi = Pr + a.lds_x_half;
  const double* Mr = Pi + a.lds_x_half;
  const double* Mi = Mr + a.lds_x_half;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int n_mt = g.M1 >> 4, n_nt = NR >> 4;
  const int n_t1 = n_mt * n_nt;
  if (!(g.bh & 1)) {
    // ---- even block height n = 2 h: the sums split by the parity of ky (radix 2) ----------------------------------------------
    //   U[y] = Ue[y] + Uo[y], U[h - y] = Ue[y] - Uo[y];  V[y] = Ve[y] + Vo[y], V[h - y] = -Ve[y] + Vo[y]   (dft_tt_write)
    // with Ue / Ve over ky = 2 m and Uo / Vo over ky = 2 m + 1, each evaluated for y in [0, h / 2] only: half the products of the
    // direct sums.  A work unit is one parity of one (output tile, re | im) pair; the two parities of a pair are units j, j + 1
    // of the same wave and run through the K loop together (four independent MFMAs per K step).  Row 2 (k0 + l4) (+ 1) of a
    // plane may lie beyond its zero padding in the last K step only: that step reads a clamped row and zeroes the operand.
    const int n_pairs = 2 * n_mt * (g.NRh >> 4);
    const int Ke = g.Ke, bh = g.bh;
    const uint32_t n8 = 8u * (uint32_t)bh;
    const double* __restrict__ T2 = (TABMODE == 1) ? tabA : (TABMODE == 0 ? a.tables + sc.fy_off : nullptr);      // [cos: KR x NR][sin: KR x NR]
#pragma unroll
    for (int j = 0; j < UPW; j += 2) {
      const int pq = w + (j >> 1) * NW;
      v4f64 cE = {0.0, 0.0, 0.0, 0.0}, sE = cE, cO = cE, sO = cE;
      if ((j + 1 < UPW) && pq < n_pairs && !(a.dbg & 2)) {
        const int t = pq >> 1;
        const int nt = tile_div(t, g.q_mt), mt = t - nt * n_mt;
        const double* __restrict__ Ac = (pq & 1) ? Pi : Pr;
        const double* __restrict__ As = (pq & 1) ? Mr : Mi;
        const int col = 16 * mt + l15, y = 16 * nt + l15;
        uint32_t mE = 0u, mO = 0u, d8 = 0u;
        if (TABMODE == 2) {
          const uint32_t yy = mod_magic((uint32_t)y, (uint32_t)bh, sc.m_bh);
          mE = 8u * mod_magic((uint32_t)(2 * l4) * yy, (uint32_t)bh, sc.m_bh);
          mO = 8u * mod_magic((uint32_t)(2 * l4 + 1) * yy, (uint32_t)bh, sc.m_bh);
          d8 = 8u * mod_magic(8u * yy, (uint32_t)bh, sc.m_bh);
        }
        auto kstep = [&](const int k0, const bool last) {
          int rE = 2 * (k0 + l4), rO = rE + 1;
          const bool vE = rE < KR, vO = rO < KR;
          if (last) { rE = min(rE, KR - 1); rO = min(rO, KR - 1); }
          double aEc = Ac[rE * SX + col], aEs = As[rE * SX + col], aOc = Ac[rO * SX + col], aOs = As[rO * SX + col];
          if (last) {
            if (!vE) { aEc = 0.0; aEs = 0.0; }
            if (!vO) { aOc = 0.0; aOs = 0.0; }
          }
          double bEc, bEs, bOc, bOs;
          if (TABMODE == 2) {
            const char* tE = (const char*)tabA + mE;
            const char* tO = (const char*)tabA + mO;
            bEc = *(const double*)tE; bEs = *(const double*)(tE + 8 * kT1S);
            bOc = *(const double*)tO; bOs = *(const double*)(tO + 8 * kT1S);
            mE += d8; mE = min(mE, mE - n8);
            mO += d8; mO = min(mO, mO - n8);
          } else {
            bEc = T2[rE * NR + y]; bEs = T2[KR * NR + rE * NR + y];
            bOc = T2[rO * NR + y]; bOs = T2[KR * NR + rO * NR + y];
          }
          cE = __builtin_amdgcn_mfma_f64_16x16x4f64(aEc, bEc, cE, 0, 0, 0);
          sE = __builtin_amdgcn_mfma_f64_16x16x4f64(aEs, bEs, sE, 0, 0, 0);
          cO = __builtin_amdgcn_mfma_f64_16x16x4f64(aOc, bOc, cO, 0, 0, 0);
          sO = __builtin_amdgcn_mfma_f64_16x16x4f64(aOs, bOs, sO, 0, 0, 0);
        };
        for (int k0 = 0; k0 < Ke - 4; k0 += 4) kstep(k0, false);
        kstep(Ke - 4, true);
      }
      uc[j] = cE; us[j] = sE;
      if (j + 1 < UPW) { uc[j + 1] = cO; us[j + 1] = sO; }
    }
    return;
  }
  if constexpr (TABMODE == 2) {
    // Two units at a time: per K step the eight operands of both are requested together, then four independent MFMAs follow --
    // one exposed LDS round trip per four MFMAs instead of per two (a lone unit, the odd one out, runs two K steps per turn).
    struct Unit { int ao; uint32_t m8, d8; const double* Ac; const double* As; bool on; };
    const uint32_t n8 = 8u * (uint32_t)g.bh;
    auto setup = [&](const int j) {
      Unit x{0, 0u, 0u, Pr, Mi, false};
      const int u = w + j * NW, t = u >> 1;
      x.on = (j < UPW) && (t < n_t1) && !(a.dbg & 2);
      if (x.on) {
        const int nt = tile_div(t, g.q_mt), mt = t - nt * n_mt;
        x.ao = l4 * SX + 16 * mt + l15;
        x.Ac = (u & 1) ? Pi : Pr; x.As = (u & 1) ? Mr : Mi;
        const uint32_t yy = mod_magic((uint32_t)(16 * nt + l15), (uint32_t)g.bh, sc.m_bh);
        x.m8 = 8u * mod_magic((uint32_t)l4 * yy, (uint32_t)g.bh, sc.m_bh);       // byte index of (k, y) = 8 * ((k * y) mod bh)
        x.d8 = 8u * mod_magic(4u * yy, (uint32_t)g.bh, sc.m_bh);                 // ... stepping k by 4
      }
      return x;
    };
    auto fetch = [&](Unit& x, const int k0, double& a1, double& a2, double& b1, double& b2) {
      const char* tb = (const char*)tabA + x.m8;
      const int o = x.ao + k0 * SX;
      a1 = x.Ac[o]; a2 = x.As[o]; b1 = *(const double*)tb; b2 = *(const double*)(tb + 8 * kT1S);
      x.m8 += x.d8;
      x.m8 = min(x.m8, x.m8 - n8);            // unsigned: m8 - n8 wraps to a huge value when m8 < n8
    };
#pragma unroll
    for (int j = 0; j < UPW; j += 2) {
      Unit A = setup(j), B = setup(j + 1);
      v4f64 cA = {0.0, 0.0, 0.0, 0.0}, sA = cA, cB = cA, sB = cA;
      if (A.on && B.on) {
        {
          const v4f64 Z = {0.0, 0.0, 0.0, 0.0};
          double a1, a2, b1, b2, a3, a4, b3, b4;
          fetch(A, 0, a1, a2, b1, b2);
          fetch(B, 0, a3, a4, b3, b4);
          cA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, Z, 0, 0, 0);
          sA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, Z, 0, 0, 0);
          cB = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, Z, 0, 0, 0);
          sB = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, Z, 0, 0, 0);
        }
        for (int k0 = 4; k0 < KR; k0 += 4) {
          double a1, a2, b1, b2, a3, a4, b3, b4;
          fetch(A, k0, a1, a2, b1, b2);
          fetch(B, k0, a3, a4, b3, b4);
          cA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cA, 0, 0, 0);
          sA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, sA, 0, 0, 0);
          cB = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, cB, 0, 0, 0);
          sB = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, sB, 0, 0, 0);
        }
      } else if (A.on) {
        int k0 = 0;
        for (; k0 + 8 <= KR; k0 += 8) {
          double a1, a2, b1, b2, a3, a4, b3, b4;
          fetch(A, k0, a1, a2, b1, b2);
          fetch(A, k0 + 4, a3, a4, b3, b4);
          cA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cA, 0, 0, 0);
          sA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, sA, 0, 0, 0);
          cA = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, cA, 0, 0, 0);
          sA = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, sA, 0, 0, 0);
        }
        if (k0 < KR) {
          double a1, a2, b1, b2;
          fetch(A, k0, a1, a2, b1, b2);
          cA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, cA, 0, 0, 0);
          sA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, sA, 0, 0, 0);
        }
      }
      uc[j] = cA; us[j] = sA;
      if (j + 1 < UPW) { uc[j + 1] = cB; us[j + 1] = sB; }
    }
    return;
  }
  const double* __restrict__ FC = a.tables + sc.fy_off;      // [KR][NR]
  const double* __restrict__ FS = FC + KR * NR;
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    v4f64 ac = {0.0, 0.0, 0.0, 0.0}, as = ac;
    const int u = w + j * NW;
    const int t = u >> 1;
    if (t < n_t1 && !(a.dbg & 2)) {
      const int nt = tile_div(t, g.q_mt), mt = t - nt * n_mt;
      const int ao = l4 * SX + 16 * mt + l15;
      const double* __restrict__ Ac = (u & 1) ? Pi : Pr;
      const double* __restrict__ As = (u & 1) ? Mr : Mi;
      const double* fc_p = FC + l4 * NR + 16 * nt + l15;
      const double* fs_p = FS + l4 * NR + 16 * nt + l15;
#pragma unroll 2
      for (int k0 = 0; k0 < KR; k0 += 4) {
        double bc, bs;
        if (TABMODE == 1) {
          const int bo = (l4 + k0) * NR + 16 * nt + l15;
          bc = tabA[bo]; bs = tabA[KR * NR + bo];
        } else {
          bc = fc_p[k0 * NR]; bs = fs_p[k0 * NR];
        }
        const int o = ao + k0 * SX;
        ac = __builtin_amdgcn_mfma_f64_16x16x4f64(Ac[o], bc, ac, 0, 0, 0);
        as = __builtin_amdgcn_mfma_f64_16x16x4f64(As[o], bs, as, 0, 0, 0);
      }
    }
    uc[j] = ac; us[j] = as;
  }
}

// FOLD_CK: the rows kx of T^T are scaled by the Hermitian-half factor c_kx (1 for kx in {0, bw / 2}, else 2) here instead of
// in the stage-2 operand table (TABMODE 2 reads plain cos / sin): (2 a) b and a (2 b) are the same product, exactly.
template <int NW, int UPW, bool FOLD_CK = false>
__device__ __forceinline__ void dft_tt_write(const int w, const int lane, const PropGeom& g, double* TT, const v4f64 (&uc)[UPW],
                                             const v4f64 (&us)[UPW]) {
  const int l15 = lane & 15, l4 = lane >> 4;
  const int n_mt = g.M1 >> 4, n_nt = g.NR >> 4;
  const int n_t1 = n_mt * n_nt;
  const int ST = g.ST, Kc = g.Kc, hh = g.hh, bh = g.bh;
  if (!(bh & 1)) {
    // even block height: units j, j + 1 hold the even- and odd-ky half-sums of pair w + (j / 2) NW (dft_stage1); rows y and hh - y
    const int n_pairs = 2 * n_mt * (g.NRh >> 4);
#pragma unroll
    for (int j = 0; j + 1 < UPW; j += 2) {
      const int pq = w + (j >> 1) * NW;
      if (pq < n_pairs) {
        const int t = pq >> 1;
        const bool im = pq & 1;
        const int nt = tile_div(t, g.q_mt), mt = t - nt * n_mt;
        const int y = 16 * nt + l15;
        if (y <= g.hq) {
          const int ym = hh - y;
          double* __restrict__ Th = TT + (im ? Kc * ST : 0);   // real rows, then imaginary rows
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int kx = 16 * mt + l4 + 4 * q;
            if (kx < Kc) {
              const double U = uc[j][q] + uc[j + 1][q], V = us[j][q] + us[j + 1][q];
              const double Um = uc[j][q] - uc[j + 1][q], Vm = us[j + 1][q] - us[j][q];
              double t1 = im ? U + V : U - V, t2 = im ? U - V : U + V;
              double t3 = im ? Um + Vm : Um - Vm, t4 = im ? Um - Vm : Um + Vm;
              if (FOLD_CK && kx != 0 && kx != g.hw) { t1 = 2.0 * t1; t2 = 2.0 * t2; t3 = 2.0 * t3; t4 = 2.0 * t4; }
              Th[kx * ST + y] = t1;
              if (y > 0 && y < hh) Th[kx * ST + (bh - y)] = t2;
              if (ym != y) {
                Th[kx * ST + ym] = t3;
                if (ym > 0 && ym < hh) Th[kx * ST + (bh - ym)] = t4;
              }
            }
          }
        }
      }
    }
    return;
  }
#pragma unroll
  for (int j = 0; j < UPW; ++j) {
    const int u = w + j * NW;
    const int t = u >> 1;
    if (t < n_t1) {
      const int nt = tile_div(t, g.q_mt), mt = t - nt * n_mt;
      const int y = 16 * nt + l15;
      if (y <= hh) {
        double* __restrict__ Th = TT + ((u & 1) ? Kc * ST : 0);   // real rows, then imaginary rows
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int kx = 16 * mt + l4 + 4 * q;
          if (kx < Kc) {
            double t1 = (u & 1) ? uc[j][q] + us[j][q] : uc[j][q] - us[j][q];
            double t2 = (u & 1) ? uc[j][q] - us[j][q] : uc[j][q] + us[j][q];
            if (FOLD_CK && kx != 0 && kx != g.hw) { t1 = 2.0 * t1; t2 = 2.0 * t2; }
            Th[kx * ST + y] = t1;
            if (y > 0 && y < hh) Th[kx * ST + (bh - y)] = t2;
          }
        }
      }
    }
  }
}

// Tile slot j of wave w after stage 2: which rows and which column the lane's four values belong to.  Direct form: slot = output
// tile t = w + j NW of the (N1 / 16) x (M1 / 16) grid, x = 16 nt + l15.  Split form (g.split2): slots j, j + 1 = pair w + (j / 2) NW
// of the (N1 / 16) x (M1h / 16) grid; the even slot holds x0 = 16 nt + l15 <= hqx, the odd one its mirror hw - x0 (nothing where that is x0 itself).
struct SlotX { int mt, x; bool ok; };
__device__ __forceinline__ SlotX slot_x(const int w, const int j, const int NW, const int l15, const PropGeom& g) {
  SlotX r;
  const int n_mt2 = g.N1 >> 4;
  if (g.split2) {
    const int tp = w + (j >> 1) * NW;
    const int nt = tile_div(tp, g.q_mt2);
    r.mt = tp - nt * n_mt2;
    const int x0 = 16 * nt + l15;
    r.x = (j & 1) ? g.hw - x0 : x0;
    r.ok = (w >= 0) && tp < n_mt2 * (g.M1h >> 4) && x0 <= g.hqx && !((j & 1) && g.hw - x0 == x0);
  } else {
    const int t = w + j * NW;
    const int nt = tile_div(t, g.q_mt2);
    r.mt = t - nt * n_mt2;
    r.x = 16 * nt + l15;
    r.ok = (w >= 0) && t < n_mt2 * (g.M1 >> 4) && r.x <= g.hw;
  }
  return r;
}

// ---- stage 2 (MFMA): E = Tr^T Gc, O = Ti^T Gs on x in [0, hw]; results stay in registers -----------
// tile t of the (N1/16) x (M1/16) output grid goes to wave t mod NW
// TABMODE as in dft_stage1; 2: tabG = 1-D table of the block width, cos at [m], -sin at [kT1S + m]; T^T carries c_kx
template <int NW, int MAXT, int TABMODE>
__device__ __forceinline__ void dft_stage2(const int w, const int lane, const ProposeArgs& a, const PropScalars& sc, const PropGeom& g,
                                           const double* TT, const double* tabG, v4f64 (&fe)[MAXT], v4f64 (&fo)[MAXT]) {
  const int l15 = lane & 15, l4 = lane >> 4;
  const int ST = g.ST, Kc = g.Kc, M1 = g.M1;
  const int n_mt2 = g.N1 >> 4, n_nt2 = M1 >> 4;
  const int n_t2 = n_mt2 * n_nt2;
  if constexpr (MAXT >= 2) if (g.split2) {
    // ---- even block width, split by the parity of kx (as dft_stage1 by the parity of ky): E[x] = Ee + Eo, E[hw - x] = Ee - Eo,
    // O[x] = Oe + Oo, O[hw - x] = -Oe + Oo, the half-sums over kx = 2 m and kx = 2 m + 1 for x in [0, hw / 2] only.  Slot j of the wave
    // leaves with (E, O) of x0, slot j + 1 with (E, O) of hw - x0 (slot_x): standardise and emit_field treat them as two tiles.
    const int n_tp = n_mt2 * (g.M1h >> 4);
    const int Kce = g.Kce, bw = g.bw;
    const uint32_t n8 = 8u * (uint32_t)bw;
    const double* __restrict__ G2 = (TABMODE == 1) ? tabG : (TABMODE == 0 ? a.tables + sc.g_off : nullptr);      // [cos: Kc x M1][-sin: Kc x M1]
#pragma unroll
    for (int j = 0; j + 1 < MAXT; j += 2) {
      const int tp = w + (j >> 1) * NW;
      v4f64 eE = {0.0, 0.0, 0.0, 0.0}, oE = eE, eO = eE, oO = eE;
      if (tp < n_tp && !(a.dbg & 4)) {
        const int nt = tile_div(tp, g.q_mt2), mt = tp - nt * n_mt2;
        const int ycol = 16 * mt + l15, x = 16 * nt + l15;
        uint32_t mE = 0u, mO = 0u, d8 = 0u;
        if (TABMODE == 2) {
          const uint32_t xx = mod_magic((uint32_t)x, (uint32_t)bw, sc.m_bw);
          mE = 8u * mod_magic((uint32_t)(2 * l4) * xx, (uint32_t)bw, sc.m_bw);
          mO = 8u * mod_magic((uint32_t)(2 * l4 + 1) * xx, (uint32_t)bw, sc.m_bw);
          d8 = 8u * mod_magic(8u * xx, (uint32_t)bw, sc.m_bw);
        }
        auto kstep = [&](const int k0, const bool last) {
          int rE = 2 * (k0 + l4), rO = rE + 1;
          const bool vE = rE < Kc, vO = rO < Kc;
          if (last) { rE = min(rE, Kc - 1); rO = min(rO, Kc - 1); }
          double aEr = TT[rE * ST + ycol], aEi = TT[(Kc + rE) * ST + ycol], aOr = TT[rO * ST + ycol], aOi = TT[(Kc + rO) * ST + ycol];
          if (last) {
            if (!vE) { aEr = 0.0; aEi = 0.0; }
            if (!vO) { aOr = 0.0; aOi = 0.0; }
          }
          double bEc, bEs, bOc, bOs;
          if (TABMODE == 2) {
            const char* tE = (const char*)tabG + mE;
            const char* tO = (const char*)tabG + mO;
            bEc = *(const double*)tE; bEs = *(const double*)(tE + 8 * kT1S);
            bOc = *(const double*)tO; bOs = *(const double*)(tO + 8 * kT1S);
            mE += d8; mE = min(mE, mE - n8);
            mO += d8; mO = min(mO, mO - n8);
          } else {
            bEc = G2[rE * M1 + x]; bEs = G2[Kc * M1 + rE * M1 + x];
            bOc = G2[rO * M1 + x]; bOs = G2[Kc * M1 + rO * M1 + x];
          }
          eE = __builtin_amdgcn_mfma_f64_16x16x4f64(aEr, bEc, eE, 0, 0, 0);
          oE = __builtin_amdgcn_mfma_f64_16x16x4f64(aEi, bEs, oE, 0, 0, 0);
          eO = __builtin_amdgcn_mfma_f64_16x16x4f64(aOr, bOc, eO, 0, 0, 0);
          oO = __builtin_amdgcn_mfma_f64_16x16x4f64(aOi, bOs, oO, 0, 0, 0);
        };
        for (int k0 = 0; k0 < Kce - 4; k0 += 4) kstep(k0, false);
        kstep(Kce - 4, true);
      }
      fe[j] = eE + eO; fo[j] = oE + oO;
      fe[j + 1] = eE - eO; fo[j + 1] = oO - oE;
    }
    return;
  }
  if constexpr (TABMODE == 2) {
    // two tiles at a time, as in dft_stage1
    struct Tile { const double* a_p; uint32_t m8, d8; bool on; };
    const uint32_t n8 = 8u * (uint32_t)g.bw;
    auto setup = [&](const int j) {
      Tile x{TT, 0u, 0u, false};
      const int t = w + j * NW;
      x.on = (j < MAXT) && (t < n_t2) && !(a.dbg & 4);
      if (x.on) {
        const int nt = tile_div(t, g.q_mt2), mt = t - nt * n_mt2;
        x.a_p = TT + l4 * ST + 16 * mt + l15;
        const uint32_t xx = mod_magic((uint32_t)(16 * nt + l15), (uint32_t)g.bw, sc.m_bw);
        x.m8 = 8u * mod_magic((uint32_t)l4 * xx, (uint32_t)g.bw, sc.m_bw);
        x.d8 = 8u * mod_magic(4u * xx, (uint32_t)g.bw, sc.m_bw);
      }
      return x;
    };
    auto fetch = [&](Tile& x, const int k0, double& a1, double& a2, double& b1, double& b2) {
      const char* tb = (const char*)tabG + x.m8;
      a1 = x.a_p[k0 * ST]; a2 = x.a_p[(Kc + k0) * ST]; b1 = *(const double*)tb; b2 = *(const double*)(tb + 8 * kT1S);
      x.m8 += x.d8;
      x.m8 = min(x.m8, x.m8 - n8);
    };
#pragma unroll
    for (int j = 0; j < MAXT; j += 2) {
      Tile A = setup(j), B = setup(j + 1);
      v4f64 eA = {0.0, 0.0, 0.0, 0.0}, oA = eA, eB = eA, oB = eA;
      if (A.on && B.on) {
        {
          const v4f64 Z = {0.0, 0.0, 0.0, 0.0};
          double a1, a2, b1, b2, a3, a4, b3, b4;
          fetch(A, 0, a1, a2, b1, b2);
          fetch(B, 0, a3, a4, b3, b4);
          eA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, Z, 0, 0, 0);
          oA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, Z, 0, 0, 0);
          eB = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, Z, 0, 0, 0);
          oB = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, Z, 0, 0, 0);
        }
        for (int k0 = 4; k0 < Kc; k0 += 4) {
          double a1, a2, b1, b2, a3, a4, b3, b4;
          fetch(A, k0, a1, a2, b1, b2);
          fetch(B, k0, a3, a4, b3, b4);
          eA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, eA, 0, 0, 0);
          oA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, oA, 0, 0, 0);
          eB = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, eB, 0, 0, 0);
          oB = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, oB, 0, 0, 0);
        }
      } else if (A.on) {
        int k0 = 0;
        for (; k0 + 8 <= Kc; k0 += 8) {
          double a1, a2, b1, b2, a3, a4, b3, b4;
          fetch(A, k0, a1, a2, b1, b2);
          fetch(A, k0 + 4, a3, a4, b3, b4);
          eA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, eA, 0, 0, 0);
          oA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, oA, 0, 0, 0);
          eA = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, eA, 0, 0, 0);
          oA = __builtin_amdgcn_mfma_f64_16x16x4f64(a4, b4, oA, 0, 0, 0);
        }
        if (k0 < Kc) {
          double a1, a2, b1, b2;
          fetch(A, k0, a1, a2, b1, b2);
          eA = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, eA, 0, 0, 0);
          oA = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, oA, 0, 0, 0);
        }
      }
      fe[j] = eA; fo[j] = oA;
      if (j + 1 < MAXT) { fe[j + 1] = eB; fo[j + 1] = oB; }
    }
    return;
  }
  const double* __restrict__ GC = a.tables + sc.g_off;       // [Kc][M1]
  const double* __restrict__ GS = GC + Kc * M1;
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    v4f64 ae = {0.0, 0.0, 0.0, 0.0}, ao = ae;
    const int t = w + j * NW;
    if (t < n_t2 && !(a.dbg & 4)) {
      const int nt = tile_div(t, g.q_mt2), mt = t - nt * n_mt2;
      const double* a_p = TT + l4 * ST + 16 * mt + l15;
      const double* gc_p = GC + l4 * M1 + 16 * nt + l15;
      const double* gs_p = GS + l4 * M1 + 16 * nt + l15;
#pragma unroll 2
      for (int k0 = 0; k0 < Kc; k0 += 4) {
        double gc, gs;
        if (TABMODE == 1) {
          const int bo = (l4 + k0) * M1 + 16 * nt + l15;
          gc = tabG[bo]; gs = tabG[Kc * M1 + bo];
        } else {
          gc = gc_p[k0 * M1]; gs = gs_p[k0 * M1];
        }
        ae = __builtin_amdgcn_mfma_f64_16x16x4f64(a_p[k0 * ST], gc, ae, 0, 0, 0);
        ao = __builtin_amdgcn_mfma_f64_16x16x4f64(a_p[(Kc + k0) * ST], gs, ao, 0, 0, 0);
      }
    }
    fe[j] = ae; fo[j] = ao;
  }
}

// edge-mask values of the wave's cells (TABLDS form: issued early, in flight during stage 2 and the reductions)
template <int NW, int MAXT>
__device__ __forceinline__ void mask_prefetch(const int w, const int lane, const ProposeArgs& a, const PropScalars& sc, const PropGeom& g,
                                              double (&mreg)[MAXT][8]) {
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bh = g.bh, bw = g.bw, hw = g.hw;
  const int n_mt2 = g.N1 >> 4, n_t2 = n_mt2 * (g.M1 >> 4);
  const dev::rsrc_t r_mask = dev::make_rsrc(a.B.masks + sc.mask_off, (uint32_t)(bh * bw) * 8u);
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const int t = w + j * NW;
    const int nt = tile_div(t, g.q_mt2), mt = t - nt * n_mt2;
    const int x = 16 * nt + l15;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 16 * mt + l4 + 4 * q;
      const bool ok = (t < n_t2) && (y < bh) && (x <= hw);
      mreg[j][2 * q] = dev::ld_f64<0>(r_mask, ok ? (uint32_t)(y * bw + x) * 8u : dev::kOOB);
      mreg[j][2 * q + 1] = dev::ld_f64<0>(r_mask, (ok && x > 0 && x < hw) ? (uint32_t)(y * bw + bw - x) * 8u : dev::kOOB);
    }
  }
}

// Sum over the workgroup of per-tile partials: tile t is reduced by the wave that owns it (fixed lane order, DPP),
// then the 16 tile slots are added by a fixed DPP tree in every thread.  red: 16 doubles of LDS, not reused before the
// next barrier.  Every thread of the workgroup must call it (it contains the barrier); waves that own no tile pass w < 0
// and write nothing.  All 16 slots must be written: slot t by its owner, or by wave 0 of the owners if t >= n_tiles...
template <int NW, int MAXT>
__device__ __forceinline__ double tiles_sum(const double (&part)[MAXT], const int n_tiles, double* red, const int w, const int lane) {
  // 16 tile slots (every block table the fused kernel takes; all forms then add them in the same order: bit-identical
  // fields), or 32 for the wide instantiation of the stand-alone kernel (blocks beyond ~80 x 80)
  constexpr int SLOTS = (NW * MAXT > 16) ? 32 : 16;
  static_assert(NW * MAXT >= 16, "every tile slot must have an owner");
  if (w >= 0) {
#pragma unroll
    for (int j = 0; j < MAXT; ++j) {
      const double ws = dev::wave64_sum(part[j]);
      const int t = w + j * NW;
      if (lane == 0 && t < SLOTS) red[t] = (t < n_tiles) ? ws : 0.0;   // all slots are written
    }
  }
  // only LDS traffic has to be complete here: a bare barrier leaves the caller's global loads in flight
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  double r = dev::row16_sum(red[lane & 15]);   // fixed tree over the tile slots: the same value in every lane
  if (SLOTS == 32) r += dev::row16_sum(red[16 + (lane & 15)]);
  return r;
}

// ---- standardise (MCMC.py:248) on the register-resident field -------------------------------------
// lane holds, per (tile j, reg q): E and O with n * field[y][x] = E + O and, for 0 < x < hw, n * field[y][bw - x] = E - O
// (n = bh bw: the inverse DFT's 1 / n is still to be applied).  On return fe / fo hold n * (field - mean) for those two
// cells (the DC coefficient dc is n * mean) and the function value is gain / n with gain = scale / (sd + 1e-12): the
// standardised cell is fe * value.  The 1 / n is applied to the sum of squares and to the gain instead of to every
// cell, which leaves four fp64 operations per cell here and one in emit_field, where scaled values cost five and two;
// the deviations are still squared about the mean (fields whose spectrum is all but DC have mean^2 >> variance).
template <int NW, int MAXT>
__device__ __forceinline__ double standardise(const int w, const int lane, const PropScalars& sc, const PropGeom& g, const double dc,
                                              double* red, v4f64 (&fe)[MAXT], v4f64 (&fo)[MAXT], const double* pw_red = nullptr) {
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bh = g.bh, hw = g.hw;
  const int n_mt2 = g.N1 >> 4, n_t2 = n_mt2 * (g.M1 >> 4);
  const int ncell = bh * g.bw;
  const double inv_n = 1.0 / (double)ncell;
  if (pw_red) {
    // the variance comes from the spectrum (coef_items: pw_red[0 .. NW) = the waves' partials of sum_{k != 0} |X_k|^2): only the deviations
    // are formed here; the barrier stays (T^T -> field tile: every wave is past its stage-2 reads)
#pragma unroll
    for (int j = 0; j < MAXT; ++j)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double e = fe[j][q], o = fo[j][q];
        fe[j][q] = (e + o) - dc; fo[j][q] = (e - o) - dc;
      }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    double S = pw_red[0];
#pragma unroll
    for (int i = 1; i < NW; ++i) S += pw_red[i];
    const double sd = sqrt(S * inv_n * inv_n);      // field = ifft / n: sum (field - mean)^2 = S / n, variance S / n^2
    return sc.scale / (sd + 1e-12) * inv_n;
  }
  double part[MAXT];
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const SlotX sx = slot_x(w, j, NW, l15, g);
    const int mt = sx.mt, x = sx.x;
    const bool okx = sx.ok;
    const bool twox = okx && (x > 0) && (x < hw);
    double p = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 16 * mt + l4 + 4 * q;
      const double e = fe[j][q], o = fo[j][q];
      const double d1 = (e + o) - dc, d2 = (e - o) - dc;
      fe[j][q] = d1; fo[j][q] = d2;
      const double s1 = d1 * d1, s2 = d2 * d2;
      p += (okx && y < bh) ? s1 : 0.0;
      p += (twox && y < bh) ? s2 : 0.0;
    }
    part[j] = p;
  }
  // sum of n^2 (field - mean)^2 -> variance: twice 1 / n for the scaling, once for the mean over the cells
  // (split form: the slots in use are not the first n ones -- an unused slot's partial is zero, every slot is written)
  const double sd = sqrt(tiles_sum<NW, MAXT>(part, g.split2 ? 32 : n_t2, red, w, lane) * inv_n * inv_n * inv_n);
  return sc.scale / (sd + 1e-12) * inv_n;
}

// ---- scale, nugget (MCMC.py:251), edge mask (MCMC.py:778), store ------------------------------
// out = (t + n * sqrt(nug)) * mask with t = fe * gain, fe and gain as standardise leaves them.  Without a nugget the finished value is stored
// directly; with one, t is stored first and nugget_pass adds the nugget normals and applies the mask -- the same
// operations in the same order.  Cell (y, x) goes to out[omap(y, x)], or nowhere if omap returns a negative index.
// MASKMODE: where the edge mask of a cell comes from -- 0 the packed table in global memory, 1 `mreg` (mask_prefetch), 2 a table
// in LDS indexed by the cell's distance to the block border, m1d[min(y, bh - 1 - y, x, bw - 1 - x)] (BlockTable::mask1d).
template <int NW, int MAXT, int MASKMODE, class OMap>
__device__ __forceinline__ void emit_field(const int w, const int lane, const ProposeArgs& a, const PropScalars& sc, const PropGeom& g,
                                           const v4f64 (&fe)[MAXT], const v4f64 (&fo)[MAXT], const double (&mreg)[MAXT][8],
                                           const double gain, const bool with_nugget,
                                           double* __restrict__ out, OMap omap, const double* m1d = nullptr) {
  if (w < 0) return;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int bh = g.bh, bw = g.bw, hw = g.hw;
  const int n_mt2 = g.N1 >> 4, n_t2 = n_mt2 * (g.M1 >> 4);
  const double* __restrict__ mask = a.B.masks + sc.mask_off;
#pragma unroll
  for (int j = 0; j < MAXT; ++j) {
    const SlotX sx = slot_x(w, j, NW, l15, g);
    const int mt = sx.mt, x = sx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int y = 16 * mt + l4 + 4 * q;
      if (sx.ok && (y < bh) && !(a.dbg & 8)) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          if (half == 1 && !(x > 0 && x < hw)) continue;
          const int xx = half ? bw - x : x;
          const int o = y * bw + xx;
          const int oi = omap(y, xx);
          const double v = (half ? fo[j][q] : fe[j][q]) * gain;
          double mk;
          if (MASKMODE == 2) mk = m1d[min(min(y, bh - 1 - y), min(xx, bw - 1 - xx))];
          else mk = (MASKMODE == 1) ? mreg[j][2 * q + half] : mask[o];
          if (oi >= 0) out[oi] = with_nugget ? v : v * mk;
        }
      }
    }
  }
}

// second pass of the nugget case over cell pairs, by NTH threads (thread index t); call after a workgroup barrier
template <int NTH, bool NOISE_IN, class OMap>
__device__ __forceinline__ void nugget_pass(const int t, const ProposeArgs& a, const PropScalars& sc, const PropGeom& g,
                                            const uint64_t seed, const int64_t step, const NoiseIn noise,
                                            double* __restrict__ out, OMap omap, const double* mt) {
  const double* __restrict__ mask = a.B.masks + sc.mask_off;
  const int bw = g.bw, ncell = g.bh * g.bw;
  const double sq_nug = sqrt(sc.nug);
  for (int pr = t; 2 * pr < ncell; pr += NTH) {
    double n1, n2;
    const int o = 2 * pr;
    if (NOISE_IN) { n1 = noise.nug[o]; n2 = noise.nug[o + 1]; }
    else {
      normals2(seed, step, kStreamNugget, (uint32_t)pr, n1, n2, mt);
      n1 *= sq_nug; n2 *= sq_nug;
    }
    const int y = o / bw, x = o - y * bw;       // bw is even: the pair (o, o + 1) lies in one row
    const int o0 = omap(y, x), o1 = omap(y, x + 1);
    if (o0 >= 0) out[o0] = (out[o0] + n1) * mask[o];
    if (o1 >= 0) out[o1] = (out[o1] + n2) * mask[o + 1];
  }
}

// ---- all stages in turn, by all NT threads of the workgroup (stand-alone kernels) ------------------------------------
// plds: LDS work area of a.lds_main doubles; red: 32 + kMathTabDoubles doubles of LDS (reductions, math table).  Contains workgroup barriers: every thread must call
// it with the same (uniform) arguments.  On return other waves may still be reading `red`.
// WIDE = 2: twice the output tiles per wave (32 stage-2 tiles, 64 stage-1 units) for block tables beyond ~80 x 80.
template <int NT, bool NOISE_IN = false, int WIDE = 1, class OMap>
__device__ __forceinline__ void propose_field(const int tid, const ProposeArgs& a, const PropScalars& sc, const uint64_t seed,
                                              const int64_t step, double* plds, double* red, double* __restrict__ out, OMap omap,
                                              const NoiseIn noise = NoiseIn{nullptr, nullptr, nullptr}) {
  constexpr int NW = NT / 64, MAXT = WIDE * 16 / NW, UPW = WIDE * 32 / NW;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const PropGeom g = prop_geom(a, sc.bh, sc.bw);
  double* mt = red + 32;                      // [kMathTabDoubles] math_tables.h
  for (int i = tid; i < kMathTabDoubles; i += NT) mt[i] = a.mathtab[i];
  __syncthreads();
  // even shapes (every table of the reference) on handles of the strip family: red[16 .. 24) is free beside tiles_sum's 16 slots.  Odd sizes keep the
  // pass over the field: the half-plane's last column / row is then no self-conjugate entry and the stages weight it by convention
  double* pw_red = (a.parseval && NW == 8 && !((sc.bh | sc.bw) & 1)) ? red + 16 : nullptr;
  coef_items<NT, NOISE_IN>(tid, 0, g.nrow * g.ncol, true, a, sc, g, seed, step, plds, a.lds_x_half, noise, mt, pw_red);
  __syncthreads();
  // Mean of the field (MCMC.py:248 subtracts it): every non-DC term of the inverse DFT sums to zero over the block, so
  // mean = X[0][0] / (bh bw) with X[0][0] = Pr[0] (real: its own conjugate partner).  Read before T^T overlays the plane.
  const double dc = plds[0];
  v4f64 uc[UPW], us[UPW];
  dft_stage1<NW, UPW, false>(wave, lane, a, sc, g, plds, nullptr, uc, us);
  __syncthreads();
  dft_tt_write<NW, UPW>(wave, lane, g, plds, uc, us);
  __syncthreads();
  v4f64 fe[MAXT], fo[MAXT];
  dft_stage2<NW, MAXT, false>(wave, lane, a, sc, g, plds, nullptr, fe, fo);
  double mreg[MAXT][8];
  const double gain = standardise<NW, MAXT>(wave, lane, sc, g, dc, red, fe, fo, pw_red);
  const bool with_nugget = NOISE_IN ? (noise.nug != nullptr) : (a.rf.nugget_max > 0.0);
  emit_field<NW, MAXT, false>(wave, lane, a, sc, g, fe, fo, mreg, gain, with_nugget, out, omap);
  if (with_nugget) {
    __syncthreads();
    nugget_pass<NT, NOISE_IN>(tid, a, sc, g, seed, step, noise, out, omap, mt);
  }
}

}  // namespace gsm
