// Fused many-chain Metropolis kernel, 512-thread form: TWO workgroups per CU (68 KiB of LDS each), so that the
// barrier- and latency-bound phases of one chain's step run under the VALU/MFMA-bound phases of another chain's.
// Same contract, the same arithmetic and bit-identical results as chain_fused_kernel.hip (1024 threads, one per CU);
// what differs is what lives where:
//   * one LDS region is time-shared: DFT planes -> T^T -> proposal field f, stored by the proposal's epilogue at the
//     window-tile position of each cell -> candidate-bed tile, updated in place by the thread that owns the cell;
//   * the DFT tables stay in L2 (global loads in the MFMA loops, as in the stand-alone proposal kernel);
//   * the stencil (phase D) reads the candidate bed of the four neighbours from the LDS tile and their (surf, vel) pairs
//     from L2 (the bed-tile form of step_kernel.hip): qx = velx * (surf - bed_next) is the same product the flux-tile
//     kernels store, so residuals and energies are bit-identical;
//   * the chain state is loaded after the proposal (no registers to spare under 2 x 4 MFMA accumulators); the other
//     workgroup of the CU covers the latency.
// Sums are taken in the order of the 1024-thread kernels: thread t accumulates the cells of "virtual threads" t (even k)
// and t + 512 (odd k) separately and wave w reports the partials of virtual waves w and w + 8.
#include "gsm_internal.h"
#include "device_util.h"
#include "proposal_device.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace gsm {

using namespace dev;

namespace {
constexpr int kT5 = 512;
constexpr int kW5 = kT5 / 64;

__device__ __forceinline__ PropScalars unpack_scalars5(uint32_t dw_lane) {
  auto dw = [&](int i) { return (uint32_t)__builtin_amdgcn_readlane((int)dw_lane, i); };
  auto f64 = [&](int i) { return __builtin_bit_cast(double, ((uint64_t)dw(2 * i + 1) << 32) | dw(2 * i)); };
  PropScalars r;
  r.scale = f64(0); r.nug = f64(1); r.range_x = f64(2); r.range_y = f64(3); r.u = f64(4);
  r.aa = f64(5); r.m_const = f64(6); r.m_kappa = f64(7);
  r.si = (int)dw(16); r.row = (int)dw(17); r.col = (int)dw(18); r.bh = (int)dw(19);
  r.bw = (int)dw(20); r.fy_off = (int)dw(21); r.g_off = (int)dw(22); r.pad = 0;
  r.mask_off = (int64_t)(((uint64_t)dw(25) << 32) | dw(24));
  return r;
}
}  // namespace

size_t fused512_lds_doubles(const FusedArgs& a) {
  return (size_t)std::max(std::max(a.P.lds_main, a.T.tile_cap), a.T.B.max_bh * a.T.B.max_bw) + 4 * 16 + 32;
}

// KT = tile cells per thread: (bh + 2)(bw + 2) <= KT * 512; KT even.
template <typename TS, int KT, bool FAST_DIV, int KD>
__global__ __launch_bounds__(kT5, 4) void chain_fused512_kernel(const FusedArgs fa) {
  static_assert(KT % 2 == 0, "cells of virtual threads t and t + 512 alternate");
  constexpr bool F32 = sizeof(TS) == 4;
  const StepArgs& a = fa.T;
  const ProposeArgs& pa = fa.P;
  extern __shared__ double lds[];
  double* __restrict__ tile = lds;                                  // planes / T^T / field / candidate-bed tile
  double* __restrict__ red = lds + fa.work_len;                     // [16][4]
  double* __restrict__ red2 = red + 4 * 16;                         // [32] proposal reductions

  const StaticFields& S = a.S;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x;
  const int H = S.H, W = S.W;
  const uint32_t ncells = (uint32_t)H * (uint32_t)W;
  const size_t plane = (size_t)H * W;
  const rsrc_t r_bed = make_rsrc((const TS*)a.beds + (size_t)chain * plane, ncells * (uint32_t)sizeof(TS));
  const rsrc_t r_en = make_rsrc((const TS*)a.energy + (size_t)chain * plane, ncells * (uint32_t)sizeof(TS));
  const rsrc_t r_rs = make_rsrc(a.resampled + (size_t)chain * plane, ncells * 4u);
  const rsrc_t r_sA = make_rsrc(S.sA, ncells * 16u);
  const rsrc_t r_sC = make_rsrc(S.sC, ncells * 16u);
  const rsrc_t r_vx = make_rsrc(S.svx, ncells * 16u);   // (surf, velx)
  const rsrc_t r_vy = make_rsrc(S.svy, ncells * 16u);   // (surf, vely)
  const rsrc_t r_sc = make_rsrc(pa.scalars + (size_t)chain * pa.n_steps, (uint32_t)pa.n_steps * (uint32_t)sizeof(PropScalars));
  const uint64_t seed = pa.seeds[chain];

  double s_hi = a.loss_sum[2 * chain], s_lo = a.loss_sum[2 * chain + 1];
  double loss_prev = (s_hi + s_lo) / S.two_sigma2;

  PropScalars sc_next = unpack_scalars5((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_sc, (int)(lane < 26 ? 4u * lane : kOOB), 0, 0));
  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t rout = (int64_t)chain * a.rec_stride + a.rec_offset + s;
    const PropScalars sc = sc_next;
    const uint32_t nxt_dw = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(
        r_sc, (int)((lane < 26 && s + 1 < a.n_steps) ? (uint32_t)(s + 1) * (uint32_t)sizeof(PropScalars) + 4u * lane : kOOB), 0, 0);
    const int row = sc.row, col = sc.col, bh = sc.bh, bw = sc.bw;
    const double uu = sc.u;

    // window, clipped to the grid, and the matching sub-block of f (MCMC.py:1266-1276); halo (MCMC.py:1293-1297)
    const int r0 = max(0, row - bh / 2), r1 = min(H, row + bh / 2);
    const int c0 = max(0, col - bw / 2), c1 = min(W, col + bw / 2);
    const int mr0 = max(bh - r1, 0), mc0 = max(bw - c1, 0);
    const int wh = r1 - r0, ww = c1 - c0;
    const int hr0 = max(0, r0 - 1), hr1 = min(H, r1 + 1);
    const int hc0 = max(0, c0 - 1), hc1 = min(W, c1 + 1);
    const int th = hr1 - hr0, tw = hc1 - hc0;
    const int ncell = th * tw;
    const uint32_t m_tw = magic_for((uint32_t)tw);
    const int dr = r0 - hr0, dc = c0 - hc0;

    // ---- P: proposal field -> LDS (it ends up at the start of the shared region, over T^T) -------------
    int ptid = tid;
    asm volatile("" : "+v"(ptid));
    // the epilogue stores each field cell that falls inside the grid at its position in the window tile (tile row
    // stride tw), so phase A updates the tile in place: thread t reads f at tile[i] and writes the candidate bed there
    propose_field<kT5, false, 0>(ptid, pa, sc, seed, pa.step0 + s, lds, red2, nullptr, nullptr, [] {}, lds,
                                 [=](int y, int x) {
                                   const int wr = y - mr0, wc = x - mc0;
                                   return ((unsigned)wr < (unsigned)wh && (unsigned)wc < (unsigned)ww) ? (wr + dr) * tw + wc + dc : -1;
                                 });
    // also: the stores of the previous (accepted) step have landed before this step's loads (vmcnt(0) in the barrier)
    __syncthreads();

    auto cell = [&](int k, int& i, int& lr, int& lc, uint32_t& g, bool& valid, bool& inwin) {
      i = ptid + k * kT5;
      valid = i < ncell;
      lr = (int)__umulhi((uint32_t)i, m_tw);
      lc = i - lr * tw;
      g = (uint32_t)((hr0 + lr) * W + hc0 + lc);
      inwin = valid && (unsigned)(lr - dr) < (unsigned)wh && (unsigned)(lc - dc) < (unsigned)ww;
    };

    // ---- A: candidate bed -> LDS tile, guard, carried energy of the window ---------------------------------
    uint32_t upd_bits = 0;
    double acc_old0 = 0.0, acc_old1 = 0.0;   // virtual threads t (even k) and t + 512 (odd k)
    int guard = 0;
    constexpr int KB = 4;
#pragma unroll
    for (int kb = 0; kb < KT; kb += KB) {
      double vb[KB], ve[KB];
      double2 A2[KB];
      asm volatile("" : "+v"(ptid) :: "memory");   // also a compiler barrier: the previous sub-batch is complete
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          vb[j] = StateIO<TS>::load(r_bed, valid ? g * (uint32_t)sizeof(TS) : kOOB);
          ve[j] = StateIO<TS>::load(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB);
          A2[j] = ld_f64x2(r_sA, valid ? g * 16u : kOOB);   // (wupd, surf)
        }
      }
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          const bool upd = inwin && (__builtin_bit_cast(uint64_t, A2[j].x) != kNoUpdBits);
          upd_bits |= upd ? (1u << k) : 0u;
          double v = vb[j];
          if (upd) {
            v = v + tile[i] * A2[j].x;
            if (F32) v = (double)(float)v;
          }
          if (upd && A2[j].y - v <= 0.0) guard = 1;
          if (k & 1) acc_old1 += ve[j]; else acc_old0 += ve[j];
          if (valid) tile[i] = v;
        }
      }
      asm volatile("" : "+v"(acc_old0), "+v"(acc_old1));
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    // ---- D: residual stencil: bed neighbours from the tile, their (surf, vel) from L2; an x pass (d/dx of the x flux
    // kept in e_new[]) and a y pass, so that only two resp. three 16-byte operands per cell are in flight ------------
    double e_new[KT];
    double acc_new0 = 0.0, acc_new1 = 0.0;
    constexpr int KX = (KD == 1) ? 2 : 4, KY = (KD == 1) ? 1 : 2;
#pragma unroll
    for (int kb = 0; kb < KT; kb += KX) {
      double2 xr[KX], xl[KX];
      asm volatile("" : "+v"(ptid) :: "memory");
#pragma unroll
      for (int j = 0; j < KX; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          const int c = hc0 + lc;
          const uint32_t gl = (c == 0) ? g : g - 1, gr = (c == W - 1) ? g : g + 1;
          xr[j] = ld_f64x2(r_vx, inwin ? gr * 16u : kOOB);
          xl[j] = ld_f64x2(r_vx, inwin ? gl * 16u : kOOB);
        }
      }
#pragma unroll
      for (int j = 0; j < KX; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          const int c = hc0 + lc;
          const int il = (c == 0) ? i : i - 1, ir = (c == W - 1) ? i : i + 1;
          double dx = 0.0;
          if (inwin) {
            const double qxr = xr[j].y * (xr[j].x - tile[ir]);
            const double qxl = xl[j].y * (xl[j].x - tile[il]);
            const double ddx = qxr - qxl;
            if (FAST_DIV) dx = (ir - il == 2) ? exact_div(ddx, S.two_res, S.rcp_two_res) : exact_div(ddx, S.res, S.rcp_res);
            else dx = ddx / ((ir - il == 2) ? S.two_res : S.res);
          }
          e_new[k] = dx;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int kb = 0; kb < KT; kb += KY) {
      double2 yd[KY], yu[KY], C2[KY];
      asm volatile("" : "+v"(ptid) :: "memory");
#pragma unroll
      for (int j = 0; j < KY; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          const int r = hr0 + lr;
          const uint32_t gu = (r == 0) ? g : g - (uint32_t)W, gd = (r == H - 1) ? g : g + (uint32_t)W;
          yd[j] = ld_f64x2(r_vy, inwin ? gd * 16u : kOOB);
          yu[j] = ld_f64x2(r_vy, inwin ? gu * 16u : kOOB);
          C2[j] = ld_f64x2(r_sC, inwin ? g * 16u : kOOB);
        }
      }
#pragma unroll
      for (int j = 0; j < KY; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          const int r = hr0 + lr;
          const int iu = (r == 0) ? i : i - tw, id = (r == H - 1) ? i : i + tw;
          double e = 0.0;
          if (inwin) {
            const double qyd = yd[j].y * (yd[j].x - tile[id]);
            const double qyu = yu[j].y * (yu[j].x - tile[iu]);
            const double ddy = qyd - qyu;
            double dy;
            if (FAST_DIV) dy = (id - iu == 2 * tw) ? exact_div(ddy, S.two_res, S.rcp_two_res) : exact_div(ddy, S.res, S.rcp_res);
            else dy = ddy / ((id - iu == 2 * tw) ? S.two_res : S.res);
            const double v = ((e_new[k] + dy) + C2[j].x) - C2[j].y;
            if (!isnan(v)) e = v * v;
            if (F32) e = (double)(float)e;
          }
          e_new[k] = e;
          if (k & 1) acc_new1 += e; else acc_new0 += e;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }

    sc_next = unpack_scalars5(nxt_dw);
    // ---- R: reduce in the 1024-thread kernels' order: wave w holds virtual waves w and w + 8 ------------------
    {
      const double wo0 = wave64_sum(acc_old0), wo1 = wave64_sum(acc_old1);
      const double wn0 = wave64_sum(acc_new0), wn1 = wave64_sum(acc_new1);
      const bool w_guard = __any(guard) != 0;
      if (lane == 0) {
        red[wave * 4 + 0] = wo0; red[(wave + 8) * 4 + 0] = wo1;
        red[wave * 4 + 1] = wn0; red[(wave + 8) * 4 + 1] = wn1;
        red[wave * 4 + 2] = w_guard ? 1.0 : 0.0; red[(wave + 8) * 4 + 2] = 0.0;
      }
    }
    __syncthreads();
    const int rl = (lane & 15) * 4;
    const double so = row16_sum(red[rl]);
    const double sn = row16_sum(red[rl + 1]);
    const double gd = row16_sum(red[rl + 2]);
    double c_hi, c_err;
    two_sum(s_hi, sn - so, c_hi, c_err);
    const double c_lo = s_lo + c_err;
    double loss_next = (c_hi + c_lo) / S.two_sigma2;
    if (gd > 0.0) loss_next = INFINITY;
    const double p_acc = (loss_prev > loss_next) ? 1.0 : fmin(1.0, exp(loss_prev - loss_next));
    const bool acc = (uu <= p_acc);

    // ---- E: commit -------------------------------------------------------------------------------------
    if (acc) {
      asm volatile("" : "+v"(ptid));
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cell(k, i, lr, lc, g, valid, inwin);
        const bool upd = (upd_bits >> k) & 1u;
        StateIO<TS>::store(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB, e_new[k]);
        StateIO<TS>::store(r_bed, upd ? g * (uint32_t)sizeof(TS) : kOOB, upd ? tile[i] : 0.0);
        __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_rs, (int)(upd ? g * 4u : kOOB), 0, 0);
      }
      two_sum(c_hi, c_lo, s_hi, s_lo);
      loss_prev = loss_next;
    }
    if (tid == 0) {
      a.loss[rout] = loss_prev;
      a.accept[rout] = acc ? 1 : 0;
      if (a.blocks) { a.blocks[4 * rout] = row; a.blocks[4 * rout + 1] = col; a.blocks[4 * rout + 2] = bh; a.blocks[4 * rout + 3] = bw; }
    }
    // the tile is read by phase E: it may be overwritten (next proposal's planes) only after every wave is here; the
    // barrier's vmcnt(0) also completes this step's stores before the next step's loads
    __syncthreads();
  }
  if (tid == 0) {
    a.loss_sum[2 * chain] = s_hi;
    a.loss_sum[2 * chain + 1] = s_lo;
  }
}

template <typename TS, int KT>
static hipError_t launch_fused512_t(const FusedArgs& a, hipStream_t st) {
  const size_t lds = fused512_lds_doubles(a) * sizeof(double);
  static int kd = -1;   // GSM_F512_KD=2: two cells of phase D in flight (more registers)
  if (kd < 0) { const char* v = getenv("GSM_F512_KD"); kd = (v && atoi(v) == 2) ? 2 : 1; }
  auto kfast = (kd == 2) ? chain_fused512_kernel<TS, KT, true, 2> : chain_fused512_kernel<TS, KT, true, 1>;
  auto kslow = chain_fused512_kernel<TS, KT, false, 1>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)chain_fused512_kernel<TS, KT, true, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)chain_fused512_kernel<TS, KT, true, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (a.T.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.T.n_chains), dim3(kT5), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.T.n_chains), dim3(kT5), lds, st, a);
  return hipGetLastError();
}

bool fused512_supported(const FusedArgs& a) {
  return a.T.S.sA != nullptr && a.T.S.svx != nullptr && a.T.tile_cap <= 14 * kT5 &&
         (uint64_t)a.T.S.H * a.T.S.W * 16u < 0x80000000ull && fused512_lds_doubles(a) * sizeof(double) <= 80 * 1024;
}

hipError_t launch_chain_fused512(const FusedArgs& a_in, hipStream_t st) {
  if (!fused512_supported(a_in)) return hipErrorInvalidValue;
  FusedArgs a = a_in;
  a.work_len = (int)fused512_lds_doubles(a) - 4 * 16 - 32;
  a.fld_len = 0;
  if (a.T.f32_state) {
    if (a.T.tile_cap <= 2 * kT5) return launch_fused512_t<float, 2>(a, st);
    if (a.T.tile_cap <= 6 * kT5) return launch_fused512_t<float, 6>(a, st);
    return launch_fused512_t<float, 14>(a, st);
  }
  if (a.T.tile_cap <= 2 * kT5) return launch_fused512_t<double, 2>(a, st);
  if (a.T.tile_cap <= 6 * kT5) return launch_fused512_t<double, 6>(a, st);
  return launch_fused512_t<double, 14>(a, st);
}

}  // namespace gsm
