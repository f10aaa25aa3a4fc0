// Fused many-chain Metropolis kernel for gfx950 (MI355X), Philox mode: one persistent 1024-thread workgroup per chain
// (one per CU), and per step BOTH the spectral proposal (proposal_device.h; reference gstatsMCMC/MCMC.py:742-778,
// :176-254) AND the Metropolis step (step_flux_kernel.hip; MCMC.py:1263-1360, Topography.py:592-600) inside it.
// The proposal field never leaves the CU: it goes from the MFMA accumulators to an LDS tile and is consumed there.
// Only the chain state touches HBM: read bed (window + halo) and carried energy (window); on accept write both back
// and bump resampled_times -- the algorithmic traffic of SURVEY.md section 8d plus the halo ring.
//
// Per step:
//   P0  issue the loads of the bed / energy of the window into registers (HBM latency runs under the proposal)
//   P   proposal: Philox + Box-Muller coefficients -> LDS planes, two fp64 MFMA DFT stages, standardise,
//       scale x edge mask -> LDS field tile
//   A   candidate bed = bed + f * weight (where update_mask), thickness guard, fluxes -> two LDS tiles (they overlay
//       the proposal's planes), sum of the carried energy
//   D   5-point stencil on the flux tiles -> new energies;  R  reduction + accept test;  E  commit on accept
// LDS (80 x 80 blocks): flux tiles / DFT planes 105 KiB + field tile 50 KiB + scratch < 160 KiB.
//
// The arithmetic of a step is the same, operation for operation, as gsm_propose_philox followed by gsm_run_replay
// (tests/test_gpu_philox.py: bit-identical losses, accepts and beds).
#include "gsm_internal.h"
#include "device_util.h"
#ifdef GSM_STAMPS
// absolute stamps of the proposal's internal phases go to LDS behind its reduction scratch; the kernel folds them in
#define PSTAMP(slot) do { if (threadIdx.x == 0) ((unsigned long long*)(red + 32))[slot] = __builtin_amdgcn_s_memtime(); } while (0)
#endif
#include "proposal_device.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace gsm {

using namespace dev;

#ifdef GSM_STAMPS
// diagnostic build only (GSM_STAMPS=1 at build time): per-workgroup cycle totals of the phases, thread 0
__device__ unsigned long long g_stamps_fused[4096 * 16];
#define STAMP(slot) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[slot] += t_ - st_last; st_last = t_; } } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

// work area: flux tiles, overlaid by the DFT planes + the [cos|sin] table during the proposal;
// field tile: overlaid by the c2r table until the field is written
static int fused_work_len(const FusedArgs& a) { return std::max(std::max(2 * a.T.tile_cap, a.P.lds_main), 4 * a.P.lds_x_half + a.P.tab_max); }
static int fused_fld_len(const FusedArgs& a) { return std::max(a.T.B.max_bh * a.T.B.max_bw, a.P.tab_max); }

size_t fused_lds_doubles(const FusedArgs& a) {
  return (size_t)fused_work_len(a) + (size_t)fused_fld_len(a) + 4 * kNW + 32 + 16;
}

static_assert(sizeof(PropScalars) == 104, "PropScalars layout is unpacked dword by dword below");

// per-step scalars of the proposal: lanes 0..25 each load one dword of the record (one VGPR in flight for a whole
// step), the fields are then broadcast to SGPRs with v_readlane
__device__ __forceinline__ PropScalars unpack_scalars(uint32_t dw_lane) {
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

template <typename TS, int KT, bool FAST_DIV>
__global__ __launch_bounds__(kNT, 4) void chain_fused_kernel(const FusedArgs fa) {
  constexpr bool F32 = sizeof(TS) == 4;
  const StepArgs& a = fa.T;
  const ProposeArgs& pa = fa.P;
  extern __shared__ double lds[];
  const int work_len = fa.work_len;
  double* __restrict__ qx = lds;
  double* __restrict__ qy = lds + a.tile_cap;
  double* __restrict__ fld = lds + work_len;                       // [max_bh * max_bw]
  double* __restrict__ red = fld + fa.fld_len;                     // [kNW][4]
  double* __restrict__ red2 = red + 4 * kNW;                       // [32] proposal reductions

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
  const rsrc_t r_sB = make_rsrc(S.sB, ncells * 16u);
  const rsrc_t r_sC = make_rsrc(S.sC, ncells * 16u);
  const rsrc_t r_sc = make_rsrc(pa.scalars + (size_t)chain * pa.n_steps, (uint32_t)pa.n_steps * (uint32_t)sizeof(PropScalars));
  const uint64_t seed = pa.seeds[chain];

  double s_hi = a.loss_sum[2 * chain], s_lo = a.loss_sum[2 * chain + 1];
  double loss_prev = (s_hi + s_lo) / S.two_sigma2;
  // window of the previous step if it was accepted (its stores may still be in flight), else empty.  Older stores
  // are complete: vmcnt counts in order and every thread has since waited for younger loads of its own.
  int pr0 = 0, pr1 = 0, pc0 = 0, pc1 = 0;

#ifdef GSM_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  PropScalars sc_next = unpack_scalars((uint32_t)__builtin_amdgcn_raw_buffer_load_b32(r_sc, (int)(lane < 26 ? 4u * lane : kOOB), 0, 0));
  for (int s = 0; s < a.n_steps; ++s) {
    STAMP(15);
    const int64_t rout = (int64_t)chain * a.rec_stride + a.rec_offset + s;
    const PropScalars sc = sc_next;
    // record of step s + 1: issued now, unpacked after phase D (older than every load of this step, so complete by then)
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
    const int dr = r0 - hr0, dc = c0 - hc0;  // window origin inside the tile (0 or 1)

    // stores of an earlier accepted step must have landed before this step reads an overlapping halo window
    if ((hr0 < pr1) && (pr0 < hr1) && (hc0 < pc1) && (pc0 < hc1)) __syncthreads();

    STAMP(0);
    // Geometry of the thread's tile cells t, t + 1024, ... by (magic) division; ptid is laundered per phase so that the
    // derived values are recomputed instead of being kept live across phases.  (An incremental form without the
    // multiplies measured 1.8 % slower on the same box.)
    int ptid = tid;
    asm volatile("" : "+v"(ptid));
    auto cell = [&](int k, int& i, int& lr, int& lc, uint32_t& g, bool& valid, bool& inwin) {
      i = ptid + k * kNT;
      valid = i < ncell;
      lr = (int)__umulhi((uint32_t)i, m_tw);
      lc = i - lr * tw;
      g = (uint32_t)((hr0 + lr) * W + hc0 + lc);
      inwin = valid && (unsigned)(lr - dr) < (unsigned)wh && (unsigned)(lc - dc) < (unsigned)ww;
    };
    // Cell slot k of this WAVE holds tile cells 64 * wave + 1024 * k ...: past the end of the tile for the later slots of
    // smaller blocks (on average 2.7 of the 7 slots).  Wave-uniform, so a scalar branch skips the whole slot in the
    // stencil and commit phases.
    auto slot_on = [&](int k) { return k * kNT + 64 * wave < ncell; };
    auto relaunder = [&] { asm volatile("" : "+v"(ptid)); };

    // ---- P: proposal field -> LDS; P0 (inside, after the coefficient phase): chain state of the window -> registers,
    // in flight during the two MFMA stages ---------------------------------------------------------------------
    double vb[KT], ve[KT];
    STAMP(1);
    propose_field<kNT, true, 2 * KT>(ptid, pa, sc, seed, pa.step0 + s, lds, red2, lds + 4 * pa.lds_x_half, fld,
      [&] {
#pragma unroll
        for (int k = 0; k < KT; ++k) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          vb[k] = StateIO<TS>::load(r_bed, valid ? g * (uint32_t)sizeof(TS) : kOOB);
          ve[k] = StateIO<TS>::load(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB);
        }
      },
      fld, [bw](int y, int x) { return y * bw + x; });
#ifdef GSM_STAMPS
    if (tid == 0) {
      const unsigned long long* ps = (const unsigned long long*)(red2 + 32);
      st_acc[10] += ps[10] - st_last; st_acc[11] += ps[11] - ps[10]; st_acc[12] += ps[12] - ps[11];
      st_acc[13] += ps[13] - ps[12]; st_acc[14] += ps[14] - ps[13]; st_last = ps[14];
    }
#endif
    STAMP(2);
    __syncthreads();
    STAMP(3);

    // ---- A: candidate bed, fluxes -> LDS, guard, carried energy of the window ---------------------------
    double v_new[KT];
    uint32_t upd_bits = 0;
    double acc_old = 0.0;
    int guard = 0;
    relaunder();
    // Geometry for phases A, D and E, computed once per step after the proposal and kept packed in two registers per
    // cell: gq = flat grid index, rq = tile row | tile col << 8 | valid << 16 | in-window << 17 (+1.9 % over recomputing
    // it in every phase, same box).  The arrays are laundered per phase so that only they stay live across phases.
    uint32_t gq[KT], rq[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      int i, lr, lc; uint32_t g; bool valid, inwin;
      cell(k, i, lr, lc, g, valid, inwin);
      gq[k] = g;
      rq[k] = (uint32_t)lr | ((uint32_t)lc << 8) | (valid ? 1u << 16 : 0u) | (inwin ? 1u << 17 : 0u);
    }
    auto cellq = [&](int k, int& i, int& lr, int& lc, uint32_t& g, bool& valid, bool& inwin) {
      i = tid + k * kNT; g = gq[k];
      lr = (int)(rq[k] & 0xFFu); lc = (int)((rq[k] >> 8) & 0xFFu);
      valid = (rq[k] >> 16) & 1u; inwin = (rq[k] >> 17) & 1u;
    };
    auto launderq = [&] {
#pragma unroll
      for (int k = 0; k < KT; ++k) asm volatile("" : "+v"(gq[k]), "+v"(rq[k]));
    };
    constexpr int KB = (KT > 4) ? 2 : KT;      // cells per sub-batch of phase A (2: +0.8 % over 4, same box)
#pragma unroll
    for (int kb = 0; kb < KT; kb += KB) {
      if (kb > 0 && !slot_on(kb)) break;   // this wave has no cell in this sub-batch nor in any later one
      double vf[KB];
      double2 A2[KB], B2[KB];
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cellq(k, i, lr, lc, g, valid, inwin);
          A2[j] = ld_f64x2(r_sA, valid ? g * 16u : kOOB);   // (wupd, surf)
          B2[j] = ld_f64x2(r_sB, valid ? g * 16u : kOOB);   // (velx, vely)
          vf[j] = inwin ? fld[(mr0 + lr - dr) * bw + mc0 + lc - dc] : 0.0;
        }
      }
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cellq(k, i, lr, lc, g, valid, inwin);
          const bool upd = inwin && (__builtin_bit_cast(uint64_t, A2[j].x) != kNoUpdBits);
          upd_bits |= upd ? (1u << k) : 0u;
          double v = vb[k];
          if (upd) {
            v = v + vf[j] * A2[j].x;
            if (F32) v = (double)(float)v;
          }
          const double thick = A2[j].y - v;
          if (upd && thick <= 0.0) guard = 1;
          v_new[k] = v;
          acc_old += ve[k];
          if (valid) {
            qx[i] = B2[j].x * thick;
            qy[i] = B2[j].y * thick;
          }
        }
      }
      asm volatile("" : "+v"(acc_old));
      __builtin_amdgcn_sched_barrier(0);
    }
    // (dhdt_mc, smb) of the window cells: issued before the barrier, in flight across it
    double2 C2[KT];
    launderq();
#pragma unroll
    for (int k = 0; k < KT; ++k) {
      int i, lr, lc; uint32_t g; bool valid, inwin;
      cellq(k, i, lr, lc, g, valid, inwin);
      C2[k] = ld_f64x2(r_sC, inwin ? g * 16u : kOOB);
    }
    STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS tiles complete; the loads above stay in flight
    STAMP(5);

    // ---- D: residual stencil on the flux tiles ---------------------------------------------------------
    double e_new[KT];
    double acc_new = 0.0;
    launderq();
    // interior step (a halo ring on all four sides, ~5 steps in 6): no window cell touches a grid border, every
    // difference is central.  The general form applies np.gradient's one-sided edge rules.
    const bool interior = (hr0 < r0) && (hr1 > r1) && (hc0 < c0) && (hc1 > c1);
    auto phase_d = [&](auto interior_tag) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (!slot_on(k)) { e_new[k] = 0.0; continue; }
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cellq(k, i, lr, lc, g, valid, inwin);
        double e = 0.0;
        if (inwin) {
          double dx, dy;
          if (INTERIOR) {
            const double ddx = qx[i + 1] - qx[i - 1];
            const double ddy = qy[i + tw] - qy[i - tw];
            if (FAST_DIV) { dx = exact_div(ddx, S.two_res, S.rcp_two_res); dy = exact_div(ddy, S.two_res, S.rcp_two_res); }
            else { dx = ddx / S.two_res; dy = ddy / S.two_res; }
          } else {
            const int r = hr0 + lr, c = hc0 + lc;
            const int il = (c == 0) ? i : i - 1, ir = (c == W - 1) ? i : i + 1;
            const int iu = (r == 0) ? i : i - tw, id = (r == H - 1) ? i : i + tw;
            const double ddx = qx[ir] - qx[il];
            const double ddy = qy[id] - qy[iu];
            if (FAST_DIV) {
              dx = (ir - il == 2) ? exact_div(ddx, S.two_res, S.rcp_two_res) : exact_div(ddx, S.res, S.rcp_res);
              dy = (id - iu == 2 * tw) ? exact_div(ddy, S.two_res, S.rcp_two_res) : exact_div(ddy, S.res, S.rcp_res);
            } else {
              dx = ddx / ((ir - il == 2) ? S.two_res : S.res);
              dy = ddy / ((id - iu == 2 * tw) ? S.two_res : S.res);
            }
          }
          const double v = ((dx + dy) + C2[k].x) - C2[k].y;
          if (!isnan(v)) e = v * v;
          if (F32) e = (double)(float)e;
        }
        e_new[k] = e;
        acc_new += e;
        if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    };
    if (interior) phase_d(std::true_type{}); else phase_d(std::false_type{});

    sc_next = unpack_scalars(nxt_dw);
    STAMP(6);
    // ---- R: reduce, decide (every thread evaluates the same numbers in the same order) ----------------
    {
      const double w_old = wave64_sum(acc_old);
      const double w_new = wave64_sum(acc_new);
      const bool w_guard = __any(guard) != 0;
      if (lane == 0) {
        red[wave * 4 + 0] = w_old;
        red[wave * 4 + 1] = w_new;
        red[wave * 4 + 2] = w_guard ? 1.0 : 0.0;
      }
    }
    STAMP(7);
    __syncthreads();
    STAMP(8);
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
      launderq();
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (!slot_on(k)) continue;
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cellq(k, i, lr, lc, g, valid, inwin);
        const bool upd = (upd_bits >> k) & 1u;
        StateIO<TS>::store(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB, e_new[k]);
        StateIO<TS>::store(r_bed, upd ? g * (uint32_t)sizeof(TS) : kOOB, v_new[k]);
        __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_rs, (int)(upd ? g * 4u : kOOB), 0, 0);
      }
      two_sum(c_hi, c_lo, s_hi, s_lo);
      loss_prev = loss_next;
      pr0 = r0; pr1 = r1; pc0 = c0; pc1 = c1;
    } else {
      pr0 = pr1 = pc0 = pc1 = 0;
    }
    STAMP(9);
    if (tid == 0) {
      a.loss[rout] = loss_prev;
      a.accept[rout] = acc ? 1 : 0;
      if (a.blocks) { a.blocks[4 * rout] = row; a.blocks[4 * rout + 1] = col; a.blocks[4 * rout + 2] = bh; a.blocks[4 * rout + 3] = bw; }
    }
  }
  if (tid == 0) {
    a.loss_sum[2 * chain] = s_hi;
    a.loss_sum[2 * chain + 1] = s_lo;
  }
#ifdef GSM_STAMPS
  if (tid == 0 && chain < 4096) for (int q = 0; q < 16; ++q) g_stamps_fused[chain * 16 + q] = st_acc[q];
#endif
}

int debug_read_stamps_fused(unsigned long long* out, int n_chains) {
#ifdef GSM_STAMPS
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps_fused), sizeof(unsigned long long) * 16 * (size_t)n_chains) == hipSuccess ? 0 : -3;
#else
  (void)out; (void)n_chains;
  return -4;
#endif
}

template <typename TS, int KT>
static hipError_t launch_fused_t(const FusedArgs& a, hipStream_t st) {
  const size_t lds = fused_lds_doubles(a) * sizeof(double);
  auto kfast = chain_fused_kernel<TS, KT, true>;
  auto kslow = chain_fused_kernel<TS, KT, false>;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  if (a.T.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.T.n_chains), dim3(kNT), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.T.n_chains), dim3(kNT), lds, st, a);
  return hipGetLastError();
}

bool fused_supported(const FusedArgs& a) {
  return step_flux_supported(a.T) && a.P.tab_max > 0 && fused_lds_doubles(a) * sizeof(double) <= 160 * 1024;
}

// One launch: propose_scalars_kernel for all steps must have filled a.P.scalars (n_chains x a.P.n_steps records).
hipError_t launch_chain_fused(const FusedArgs& a_in, hipStream_t st) {
  if (!fused_supported(a_in)) return hipErrorInvalidValue;
  FusedArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("GSM_PROPOSE_DBG"); dbg = v ? atoi(v) : 0; } a.P.dbg = dbg; }   // diagnostics only
  a.work_len = fused_work_len(a);
  a.fld_len = fused_fld_len(a);
  if (a.T.f32_state) {
    if (a.T.tile_cap <= 2 * kNT) return launch_fused_t<float, 2>(a, st);
    if (a.T.tile_cap <= 4 * kNT) return launch_fused_t<float, 4>(a, st);
    return launch_fused_t<float, 7>(a, st);
  }
  if (a.T.tile_cap <= 2 * kNT) return launch_fused_t<double, 2>(a, st);
  if (a.T.tile_cap <= 4 * kNT) return launch_fused_t<double, 4>(a, st);
  return launch_fused_t<double, 7>(a, st);
}

}  // namespace gsm
