// Metropolis step kernels for gfx950 (MI355X): one workgroup per chain, the CANDIDATE bed of the proposal
// window (+1-cell halo) staged in LDS, all steps of a launch looped inside the kernel.
//
// Replaces the loop body of chain_crf.run (reference gstatsMCMC/MCMC.py:1263-1360):
//   window/clipping         MCMC.py:1266-1276
//   perturb + update mask   MCMC.py:1279-1290
//   residual on window+halo MCMC.py:1293-1315 -> Topography.get_mass_conservation_residual (Topography.py:592-600)
//   Gaussian loss           MCMC.py:1318      -> chain.loss (MCMC.py:1021-1044)
//   thickness guard         MCMC.py:1321-1329
//   accept / bookkeeping    MCMC.py:1331-1360
//
// What is carried between steps (per chain, in HBM): the bed, the "energy" e = r^2 where the cell enters the loss
// (mc_mask == 1 and r is not NaN -- nansum semantics, MCMC.py:1041) else 0, the uint32 resampled counts, and the
// compensated pair (hi, lo) with hi + lo = sum(e).  The reference carries the residual array r itself
// (MCMC.py:1308-1315, :1340) and re-reduces the whole grid every step; because the edge taper is exactly 0 on the
// block border (MCMC.py:583-623 with the driver's logistic parameters) a proposal changes r only inside its window,
// so   sum_next = sum_prev - sum_window(e_old) + sum_window(e_new)   is the same quantity.  Arithmetic per residual
// is the reference's, operation for operation (this file is built with -ffp-contract=off; the only fused op is the
// explicit fma pair of exact_div, which returns the correctly rounded quotient), so only the summation order of the
// loss differs (|rel diff| ~1e-15; test bound 1e-10) and accept decisions are identical.
//
// Per step:  A  stage bed window+halo in LDS with the perturbation already applied (f * weight where update_mask),
//               thickness guard, and sum the carried e of the window          -- 1 barrier
//            D  5-point stencil on the LDS tile -> e_new kept in registers, summed
//            R  workgroup reduction, every thread evaluates the same accept test -- 1 barrier
//            E  on accept: write bed window, e window, bump resampled           -- 1 barrier (a full fence only
//               when the next step's halo window overlaps this window; see the end of the step loop)
// HBM bytes per chain-step: read (bh+2)(bw+2) bed + bh*bw e + bh*bw f; on accept write bh*bw bed + bh*bw e and
// read-modify-write bh*bw uint32.  Static fields are shared by all chains and stay in L2 / Infinity Cache.

#include "gsm_internal.h"
#include "residual_device.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace gsm {

__device__ __forceinline__ uint32_t magic_for(uint32_t d) {
  // q = __umulhi(n, M) == n / d for n*d < 2^32 (here n < 2^16, 2 <= d < 2^8); d == 1 would need M = 2^32
  return (uint32_t)(0xFFFFFFFFu / d) + 1u;
}
__device__ __forceinline__ int div_magic(int n, uint32_t d, uint32_t M) {
  return (d == 1u) ? n : (int)__umulhi((uint32_t)n, M);   // a 2-wide block clipped at the grid edge is 1 cell wide
}

// Correctly rounded x / d from y = RN(1/d): q0 = RN(x*y), r = x - q0*d (exact in an fma), q = RN(q0 + r*y)
// (Markstein 1990).  Enabled by the host only when d's significand is not all ones and d is far from the
// exponent limits; tests/test_exact_div.py checks it against exact rational arithmetic.
__device__ __forceinline__ double exact_div(double x, double d, double y) {
  const double q0 = x * y;
  const double r = __fma_rn(-q0, d, x);
  return __fma_rn(r, y, q0);
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}

__device__ __forceinline__ void two_sum(double a, double b, double& s, double& e) {
  s = a + b;
  const double bb = s - a;
  e = (a - (s - bb)) + (b - bb);
}

// ---------------------------------------------------------------------------------------------------
// step kernel.  NT threads per workgroup, KMAX = ceil(max window cells / NT) window cells per thread.
// ---------------------------------------------------------------------------------------------------
// TS = state element type: double, or float ("fp32 state, fp64 arithmetic": the candidate bed and the new
// energies are rounded to float BEFORE they are used, so the carried sum always equals the sum of what is stored).
template <typename TS, int NT, int KMAX, bool FAST_DIV, int MINW>
__global__ __launch_bounds__(NT, MINW) void step_kernel(const StepArgs a) {
  constexpr bool F32 = sizeof(TS) == 4;
  constexpr bool DEEP = (MINW * 256 / NT) <= 1;   // one workgroup per CU: registers to spare for deep load batching
  constexpr int NW = NT / 64;
  extern __shared__ double lds[];
  double* tile = lds;
  double* red = lds + a.tile_cap;  // [NW][3]

  const StaticFields& S = a.S;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int chain = blockIdx.x;
  const int H = S.H, W = S.W;
  const size_t plane = (size_t)H * W;
  TS* __restrict__ bed = (TS*)a.beds + (size_t)chain * plane;
  TS* __restrict__ energy = (TS*)a.energy + (size_t)chain * plane;
  uint32_t* __restrict__ resamp = a.resampled + (size_t)chain * plane;
  const double2* __restrict__ svx = S.svx;
  const double2* __restrict__ svy = S.svy;
  const double2* __restrict__ dsp = S.ds;

  double s_hi = a.loss_sum[2 * chain], s_lo = a.loss_sum[2 * chain + 1];
  double loss_prev = (s_hi + s_lo) / S.two_sigma2;

  // block-size table -> LDS once; per-step scalars are fetched one step ahead (the end-of-step fence below needs the
  // next window before the step ends)
  int* tab = (int*)(red + 3 * 16);
  for (int i = tid; i < a.B.n_sizes; i += NT) { tab[2 * i] = a.B.bh[i]; tab[2 * i + 1] = a.B.bw[i]; }
  const int64_t rin0 = (int64_t)chain * a.in_stride;
  int n_si = a.size_idx[rin0], n_row = a.centre[2 * rin0], n_col = a.centre[2 * rin0 + 1];
  double n_u = a.u[rin0];
  __syncthreads();

  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t rin = rin0 + s;
    const int64_t rout = (int64_t)chain * a.rec_stride + a.rec_offset + s;
    const int si = n_si, row = n_row, col = n_col;
    const double uu = n_u;
    const bool has_next = s + 1 < a.n_steps;
    if (has_next) {
      n_si = a.size_idx[rin + 1]; n_row = a.centre[2 * rin + 2]; n_col = a.centre[2 * rin + 3]; n_u = a.u[rin + 1];
    }
    if (si < 0 || si >= a.B.n_sizes || row < 0 || row >= H || col < 0 || col >= W) {
      if (tid == 0) {
        atomicExch(a.err_flag, 1);
        a.loss[rout] = loss_prev;
        a.accept[rout] = 0;
        if (a.blocks) { a.blocks[4 * rout] = row; a.blocks[4 * rout + 1] = col; a.blocks[4 * rout + 2] = 0; a.blocks[4 * rout + 3] = 0; }
      }
      continue;  // uniform across the workgroup
    }
    const int bh = tab[2 * si], bw = tab[2 * si + 1];
    const double* __restrict__ fld = a.fields + rin * a.field_stride;

    // window, clipped to the grid, and the matching sub-block of f (MCMC.py:1266-1276)
    const int r0 = max(0, row - bh / 2), r1 = min(H, row + bh / 2);
    const int c0 = max(0, col - bw / 2), c1 = min(W, col + bw / 2);
    const int mr0 = max(bh - r1, 0), mc0 = max(bw - c1, 0);
    const int wh = r1 - r0, ww = c1 - c0;
    // halo window (MCMC.py:1293-1297)
    const int hr0 = max(0, r0 - 1), hr1 = min(H, r1 + 1);
    const int hc0 = max(0, c0 - 1), hc1 = min(W, c1 + 1);
    const int th = hr1 - hr0, tw = hc1 - hc0;
    const int ncell = th * tw, nwin = wh * ww;
    const uint32_t m_tw = magic_for((uint32_t)tw), m_ww = magic_for((uint32_t)ww);
    const int dr = r0 - hr0, dc = c0 - hc0;  // window origin inside the tile (0 or 1)

    // ---- A: candidate bed -> LDS, guard, sum of the carried energy of the window ------------------
    double acc_old = 0.0;
    int guard = 0;
    if (DEEP) {
      // deep variant (one workgroup per CU, 128-VGPR budget): every HBM-latency load of the step (bed, energy, f)
      // is issued before the first is consumed -- one exposed HBM round trip instead of one per unrolled pair
      constexpr int KT = KMAX + 1;           // tile cells per thread: (bh+2)(bw+2) <= (KMAX + 1) * NT (host-checked)
      double vb[KT], vf[KT], ve[KT];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int i = tid + k * NT;
        vb[k] = 0.0; vf[k] = 0.0; ve[k] = 0.0;
        if (i < ncell) {
          const int lr = (int)__umulhi((uint32_t)i, m_tw);
          const int lc = i - lr * tw;
          const int g = (hr0 + lr) * W + hc0 + lc;
          vb[k] = (double)__builtin_nontemporal_load(&bed[g]);
          const int wr = lr - dr, wc = lc - dc;
          if ((unsigned)wr < (unsigned)wh && (unsigned)wc < (unsigned)ww) {
            ve[k] = (double)__builtin_nontemporal_load(&energy[g]);
            vf[k] = __builtin_nontemporal_load(&fld[(mr0 + wr) * bw + mc0 + wc]);
          }
        }
      }
#pragma unroll
      for (int k = 0; k < KT; ++k) acc_old += ve[k];
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        const int i = tid + k * NT;
        if (i < ncell) {
          const int lr = (int)__umulhi((uint32_t)i, m_tw);
          const int lc = i - lr * tw;
          const int g = (hr0 + lr) * W + hc0 + lc;
          double v = vb[k];
          const int wr = lr - dr, wc = lc - dc;
          if ((unsigned)wr < (unsigned)wh && (unsigned)wc < (unsigned)ww && S.upd[g]) {
            const double pert = S.weight ? vf[k] * S.weight[g] : vf[k];
            v = v + pert;
            if (F32) v = (double)(float)v;
            if (svx[g].x - v <= 0.0) guard = 1;
          }
          tile[i] = v;
        }
      }
    } else {
#pragma unroll 2
    for (int i = tid; i < ncell; i += NT) {
      const int lr = (int)__umulhi((uint32_t)i, m_tw);
      const int lc = i - lr * tw;
      const int g = (hr0 + lr) * W + hc0 + lc;
      double v = (double)__builtin_nontemporal_load(&bed[g]);
      const int wr = lr - dr, wc = lc - dc;
      if ((unsigned)wr < (unsigned)wh && (unsigned)wc < (unsigned)ww) {
        acc_old += (double)__builtin_nontemporal_load(&energy[g]);
        if (S.upd[g]) {
          const double f = __builtin_nontemporal_load(&fld[(mr0 + wr) * bw + mc0 + wc]);
          const double pert = S.weight ? f * S.weight[g] : f;
          v = v + pert;
          if (F32) v = (double)(float)v;
          if (svx[g].x - v <= 0.0) guard = 1;
        }
      }
      tile[i] = v;
    }
    }
    __syncthreads();

    // ---- D: residual stencil on the candidate tile ---------------------------------------------------
    double e_new[KMAX];
    double acc_new = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      const int i = tid + k * NT;
      double e = 0.0;
      if (i < nwin) {
        const int wr = div_magic(i, (uint32_t)ww, m_ww);
        const int wc = i - wr * ww;
        const int r = r0 + wr, c = c0 + wc;
        const int g = r * W + c;
        if (S.mc[g] == 1) {
          const int cl = (c == 0) ? 0 : c - 1, cr = (c == W - 1) ? W - 1 : c + 1;
          const int ru = (r == 0) ? 0 : r - 1, rd = (r == H - 1) ? H - 1 : r + 1;
          const double2 xr = svx[r * W + cr], xl = svx[r * W + cl];
          const double2 yd = svy[rd * W + c], yu = svy[ru * W + c];
          const double2 dd = dsp[g];
          const int trow = (r - hr0) * tw - hc0;
          const double qxr = xr.y * (xr.x - tile[trow + cr]);
          const double qxl = xl.y * (xl.x - tile[trow + cl]);
          const double qyd = yd.y * (yd.x - tile[(rd - hr0) * tw + (c - hc0)]);
          const double qyu = yu.y * (yu.x - tile[(ru - hr0) * tw + (c - hc0)]);
          double dx, dy;
          if (FAST_DIV) {
            dx = (cr - cl == 2) ? exact_div(qxr - qxl, S.two_res, S.rcp_two_res) : exact_div(qxr - qxl, S.res, S.rcp_res);
            dy = (rd - ru == 2) ? exact_div(qyd - qyu, S.two_res, S.rcp_two_res) : exact_div(qyd - qyu, S.res, S.rcp_res);
          } else {
            dx = (qxr - qxl) / ((cr - cl == 2) ? S.two_res : S.res);
            dy = (qyd - qyu) / ((rd - ru == 2) ? S.two_res : S.res);
          }
          const double v = ((dx + dy) + dd.x) - dd.y;
          if (!isnan(v)) e = v * v;
          if (F32) e = (double)(float)e;
        }
      }
      e_new[k] = e;
      acc_new += e;
    }

    // ---- R: reduce, decide (every thread evaluates the same numbers in the same order) ----------------
    acc_old = wave_sum(acc_old);
    acc_new = wave_sum(acc_new);
    const int any_guard = __any(guard);
    if (lane == 0) {
      red[wave * 3 + 0] = acc_old;
      red[wave * 3 + 1] = acc_new;
      red[wave * 3 + 2] = any_guard ? 1.0 : 0.0;
    }
    __syncthreads();
    double so = 0.0, sn = 0.0, gd = 0.0;
#pragma unroll
    for (int w = 0; w < NW; ++w) { so += red[w * 3]; sn += red[w * 3 + 1]; gd += red[w * 3 + 2]; }
    double c_hi, c_err;
    two_sum(s_hi, sn - so, c_hi, c_err);
    const double c_lo = s_lo + c_err;
    double loss_next = (c_hi + c_lo) / S.two_sigma2;
    if (gd > 0.0) loss_next = INFINITY;
    const double p_acc = (loss_prev > loss_next) ? 1.0 : fmin(1.0, exp(loss_prev - loss_next));
    const bool acc = (uu <= p_acc);

    // ---- E: commit -------------------------------------------------------------------------------------
    if (acc) {
#pragma unroll
      for (int k = 0; k < KMAX; ++k) {
        const int i = tid + k * NT;
        if (i < nwin) {
          const int wr = div_magic(i, (uint32_t)ww, m_ww);
          const int wc = i - wr * ww;
          const int r = r0 + wr, c = c0 + wc;
          const size_t g = (size_t)r * W + c;
          __builtin_nontemporal_store((TS)e_new[k], &energy[g]);
          if (S.upd[g]) {
            __builtin_nontemporal_store((TS)tile[(r - hr0) * tw + (c - hc0)], &bed[g]);
            __builtin_nontemporal_store(__builtin_nontemporal_load(&resamp[g]) + 1u, &resamp[g]);
          }
        }
      }
      two_sum(c_hi, c_lo, s_hi, s_lo);
      loss_prev = loss_next;
    }
    if (tid == 0) {
      a.loss[rout] = loss_prev;
      a.accept[rout] = acc ? 1 : 0;
      if (a.blocks) { a.blocks[4 * rout] = row; a.blocks[4 * rout + 1] = col; a.blocks[4 * rout + 2] = bh; a.blocks[4 * rout + 3] = bw; }
    }
    // End of step.  The LDS tile / reduction scratch may be reused once every wave is here (raw barrier after the
    // wave's own LDS reads have returned).  The global stores of an accepted step must be visible to the next step's
    // loads only if the next halo window touches this window; otherwise the workgroup does not wait for them
    // (__syncthreads() would: its release fence drains vmcnt, one exposed HBM round trip per step).
    bool fence = acc;
    if (fence && has_next && (unsigned)n_si < (unsigned)a.B.n_sizes) {
      const int nbh = tab[2 * n_si], nbw = tab[2 * n_si + 1];
      const int nr0 = max(0, n_row - nbh / 2) - 1, nr1 = min(H, n_row + nbh / 2) + 1;
      const int nc0 = max(0, n_col - nbw / 2) - 1, nc1 = min(W, n_col + nbw / 2) + 1;
      fence = (nr0 < r1) && (r0 < nr1) && (nc0 < c1) && (c0 < nc1);
    }
    if (fence) __syncthreads();
    else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (tid == 0) {
    a.loss_sum[2 * chain] = s_hi;
    a.loss_sum[2 * chain + 1] = s_lo;
  }
}

size_t step_lds_bytes(int tile_cap) { return ((size_t)tile_cap + 3 * 16 + 64) * sizeof(double); }  // tile + reduction + size table (<= 64 sizes)

template <typename TS, int NT, int KMAX, int MINW>
static hipError_t launch_step_t(const StepArgs& a, hipStream_t st) {
  size_t lds = step_lds_bytes(a.tile_cap);
  { static long pad = -1; if (pad < 0) { const char* v = getenv("GSM_STEP_LDS_PAD"); pad = v ? atol(v) : 0; }
    if ((size_t)pad > lds) lds = (size_t)pad; }
  auto kfast = step_kernel<TS, NT, KMAX, true, MINW>;
  auto kslow = step_kernel<TS, NT, KMAX, false, MINW>;
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  if (a.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.n_chains), dim3(NT), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.n_chains), dim3(NT), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_step(const StepArgs& a, hipStream_t st) {
  // window cells per thread: the largest block of the table decides the instantiation
  // (window cells <= KMAX * NT and tile cells <= (KMAX + 1) * NT)
  if (a.strip) return launch_step_strip(a, st);      // pairs with chain_strip_kernel (same sums)
  const int max_win = std::max(a.B.max_bh * a.B.max_bw, a.tile_cap - 1024);
  static int variant = -1;
  if (variant < 0) { const char* v = getenv("GSM_STEP_VARIANT"); variant = v ? atoi(v) : 3; }
  // default: the flux-tile kernel (step_flux_kernel.hip); blocks beyond its two LDS tiles fall back to the bed-tile form
  if (variant == 3 && step_flux_supported(a)) return launch_step_flux(a, st);
  if (a.f32_state) {
    if (max_win <= 1024 * 7) return launch_step_t<float, 1024, 7, 8>(a, st);
    if (max_win <= 1024 * 12) return launch_step_t<float, 1024, 12, 4>(a, st);
    if (max_win <= 1024 * 20) return launch_step_t<float, 1024, 20, 4>(a, st);
    return hipErrorInvalidValue;
  }
  if (max_win <= 1024 * 7) {
    switch (variant) {
      case 0: return launch_step_t<double, 1024, 7, 4>(a, st);
      case 2: return launch_step_t<double, 512, 13, 4>(a, st);
      default: return launch_step_t<double, 1024, 7, 8>(a, st);   // <=64 VGPRs: 2 workgroups per CU (fastest measured)
    }
  }
  if (max_win <= 1024 * 12) return launch_step_t<double, 1024, 12, 4>(a, st);
  if (max_win <= 1024 * 20) return launch_step_t<double, 1024, 20, 4>(a, st);
  return hipErrorInvalidValue;  // gsm_set_blocks refuses such tables (LDS tile limit is reached first)
}

// ---------------------------------------------------------------------------------------------------
// full-grid residual, energy and loss of the current beds (MCMC.py:1189-1195)
// ---------------------------------------------------------------------------------------------------
constexpr int kBlock = 256;
constexpr int kWaves = kBlock / 64;

template <typename TS>
__global__ __launch_bounds__(kBlock) void init_loss_kernel(const StaticFields S, const TS* beds, TS* energy,
                                                           double* loss_sum, double* loss0) {
  __shared__ double red[kWaves * 2];
  const int chain = blockIdx.x, tid = threadIdx.x;
  const size_t plane = (size_t)S.H * S.W;
  const TS* bed = beds + (size_t)chain * plane;
  TS* en = energy ? energy + (size_t)chain * plane : nullptr;
  auto bed_at = [&](int rr, int cc) { return (double)bed[(size_t)rr * S.W + cc]; };
  // per-thread compensated partial
  double hi = 0.0, lo = 0.0;
  for (int g = tid; g < (int)plane; g += kBlock) {
    double e = 0.0;
    if (S.mc[g] == 1) {
      const int r = g / S.W, c = g - r * S.W;
      const double v = cell_residual(S, r, c, bed_at);
      if (!isnan(v)) e = v * v;
      if (sizeof(TS) == 4) e = (double)(float)e;
    }
    if (en) en[g] = (TS)e;
    double s, err;
    two_sum(hi, e, s, err);
    hi = s;
    lo += err;
  }
  // wave tree on (hi, lo) with two_sum at each level
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ohi = __shfl_down(hi, off, 64), olo = __shfl_down(lo, off, 64);
    double s, e;
    two_sum(hi, ohi, s, e);
    hi = s;
    lo += olo + e;
  }
  if ((tid & 63) == 0) { red[(tid >> 6) * 2] = hi; red[(tid >> 6) * 2 + 1] = lo; }
  __syncthreads();
  if (tid == 0) {
    double th = 0.0, tl = 0.0;
    for (int w = 0; w < kWaves; ++w) {
      double s, e;
      two_sum(th, red[2 * w], s, e);
      th = s;
      tl += red[2 * w + 1] + e;
    }
    double nh, nl;
    two_sum(th, tl, nh, nl);
    loss_sum[2 * chain] = nh;
    loss_sum[2 * chain + 1] = nl;
    if (loss0) loss0[chain] = (nh + nl) / S.two_sigma2;
  }
}

hipError_t launch_init_loss(const StaticFields& S, int n_chains, const void* beds, void* energy, int f32_state,
                            double* loss_sum, double* loss0, hipStream_t st) {
  if (f32_state)
    hipLaunchKernelGGL(init_loss_kernel<float>, dim3(n_chains), dim3(kBlock), 0, st, S, (const float*)beds, (float*)energy, loss_sum, loss0);
  else
    hipLaunchKernelGGL(init_loss_kernel<double>, dim3(n_chains), dim3(kBlock), 0, st, S, (const double*)beds, (double*)energy, loss_sum, loss0);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void pack_static_kernel(const StaticFields S, double2* svx, double2* svy, double2* ds) {
  const int n = S.H * S.W;
  for (int g = blockIdx.x * kBlock + threadIdx.x; g < n; g += gridDim.x * kBlock) {
    svx[g] = make_double2(S.surf[g], S.velx[g]);
    svy[g] = make_double2(S.surf[g], S.vely[g]);
    ds[g] = make_double2(S.dhdt[g], S.smb[g]);
  }
}

hipError_t launch_pack_static(const StaticFields& S, double2* svx, double2* svy, double2* ds, hipStream_t st) {
  int grid = (S.H * S.W + kBlock - 1) / kBlock;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_static_kernel, dim3(grid), dim3(kBlock), 0, st, S, svx, svy, ds);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Utilities.min_dist_from_mask (Utilities.py:21-24): distance of every cell to the nearest masked cell.
// Exact brute force: min over masked points of (dx*dx + dy*dy), then sqrt -- the arithmetic of the KD-tree
// query's final distance (no FMA: this file is built with -ffp-contract=off); points staged through LDS.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void compact_points_kernel(const double* xx, const double* yy, const uint8_t* mask,
                                                                int n, double2* pts, int* count) {
  for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
    if (mask[i]) pts[atomicAdd(count, 1)] = make_double2(xx[i], yy[i]);
}

__global__ __launch_bounds__(kBlock) void min_dist_kernel(const double* __restrict__ xx, const double* __restrict__ yy,
                                                          int n, const double2* __restrict__ pts,
                                                          const int* __restrict__ count, double* __restrict__ dist) {
  __shared__ double2 sp[kBlock];
  const int m = *count;
  const int i = blockIdx.x * kBlock + threadIdx.x;
  const double x = (i < n) ? xx[i] : 0.0, y = (i < n) ? yy[i] : 0.0;
  double best = INFINITY;
  for (int p0 = 0; p0 < m; p0 += kBlock) {
    if (p0 + (int)threadIdx.x < m) sp[threadIdx.x] = pts[p0 + threadIdx.x];
    __syncthreads();
    const int lim = min(kBlock, m - p0);
    for (int j = 0; j < lim; ++j) {
      const double dx = x - sp[j].x, dy = y - sp[j].y;
      best = fmin(best, dx * dx + dy * dy);
    }
    __syncthreads();
  }
  if (i < n) dist[i] = sqrt(best);
}

hipError_t launch_min_dist(const double* xx, const double* yy, const uint8_t* mask, int n, double2* pts, int* count,
                           double* dist, hipStream_t st) {
  hipLaunchKernelGGL(compact_points_kernel, dim3(256), dim3(kBlock), 0, st, xx, yy, mask, n, pts, count);
  hipLaunchKernelGGL(min_dist_kernel, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, st, xx, yy, n, pts, count, dist);
  return hipGetLastError();
}

__global__ __launch_bounds__(kBlock) void stream_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t n) {
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n; i += (int64_t)gridDim.x * kBlock) dst[i] = src[i];
}

hipError_t launch_stream_copy(const double* src, double* dst, int64_t n, hipStream_t st) {
  hipLaunchKernelGGL(stream_copy_kernel, dim3(2048), dim3(kBlock), 0, st, src, dst, n);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// Topography.get_mass_conservation_residual for a batch of beds (Topography.py:592-600)
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void residual_kernel(const StaticFields S, int n_chains, const double* beds,
                                                          double* out) {
  const size_t plane = (size_t)S.H * S.W;
  const size_t total = plane * (size_t)n_chains;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t chain = i / plane;
    const int g = (int)(i - chain * plane);
    const int r = g / S.W, c = g - r * S.W;
    const double* bed = beds + chain * plane;
    auto bed_at = [&](int rr, int cc) { return bed[(size_t)rr * S.W + cc]; };
    out[i] = cell_residual(S, r, c, bed_at);
  }
}

hipError_t launch_residual(const StaticFields& S, int n_chains, const double* beds, double* out, hipStream_t st) {
  const size_t total = (size_t)S.H * S.W * n_chains;
  int grid = (int)((total + kBlock - 1) / kBlock);
  if (grid > 2048) grid = 2048;
  if (grid < 1) grid = 1;
  hipLaunchKernelGGL(residual_kernel, dim3(grid), dim3(kBlock), 0, st, S, n_chains, beds, out);
  return hipGetLastError();
}

}  // namespace gsm
