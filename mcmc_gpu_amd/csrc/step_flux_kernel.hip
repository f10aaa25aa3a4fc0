// Metropolis step kernel, "flux tile" form, for gfx950 (MI355X): one 1024-thread workgroup per chain and per CU,
// all steps of a launch looped inside the kernel.  Same contract and the same arithmetic as step_kernel.hip
// (reference gstatsMCMC/MCMC.py:1263-1360, Topography.py:592-600); what differs is where the operands live:
//
//   * The mass fluxes  qx = velx * (surf - bed_next),  qy = vely * (surf - bed_next)  of the candidate bed are
//     computed ONCE per cell of the window + 1-cell halo and kept in two LDS tiles (2 x 82 x 82 fp64 = 105 KiB for
//     the 80 x 80 block).  The 5-point stencil then reads only LDS: np.gradient's central / one-sided differences of
//     the flux arrays are differences of tile neighbours.  (step_kernel.hip keeps the candidate bed in LDS and loads
//     surf/velx/vely of four neighbours per window cell: 80 B of static operands per cell against 48 B here.)
//   * The masks are folded into the static operands so that no load depends on another load:
//         sA = (wupd, surf)      wupd = crf weight (1.0 for block_type 'RF') where update_mask is set, else a tagged NaN
//         sB = (velx, vely)
//         sC = (dhdt_mc, smb)    dhdt_mc = dhdt where mc_mask == 1, else NaN  (a NaN residual is ignored by the loss:
//                                nansum semantics, MCMC.py:1041 -- the same outcome as "cell not in the mask")
//   * Every per-lane address is a 32-bit offset into a buffer descriptor; lanes outside the tile / window use an
//     out-of-range offset (the load returns 0, the store is dropped), so the phases have no divergent branches and the
//     loads of a phase issue back to back.
//   * One mapping for everything: thread t owns tile cells t, t + 1024, ...; it stages them, evaluates their residuals
//     and commits them, so the candidate bed and the new energies never leave its registers.
//   * resampled_times is bumped with a no-return integer atomic (no read round trip).
//
// Per step: A  loads + candidate bed + fluxes -> LDS, guard, sum of the carried energy      -- 1 barrier
//           D  stencil from LDS -> new energies (registers)
//           R  DPP wave reduction, 16 partials through LDS                                    -- 1 barrier
//           E  commit on accept; the stores are waited for only when the next window overlaps this one
#include "gsm_internal.h"
#include "device_util.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>

namespace gsm {

using namespace dev;

#ifdef GSM_STAMPS
// diagnostic build only (GSM_STAMPS=1 at build time): per-workgroup cycle totals of the phases, wave 0
__device__ unsigned long long g_stamps[4096 * 8];
#define STAMP(slot) do { if (tid == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[slot] += t_ - st_last; st_last = t_; } } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

size_t step_flux_lds_bytes(int tile_cap) { return ((size_t)2 * tile_cap + 4 * kNW + 64) * sizeof(double); }

// KT = tile cells per thread: (bh + 2)(bw + 2) <= KT * 1024 for every block of the table (host-checked).
template <typename TS, int KT, bool FAST_DIV>
__global__ __launch_bounds__(kNT, 4) void step_flux_kernel(const StepArgs a) {
  constexpr bool F32 = sizeof(TS) == 4;
  extern __shared__ double lds[];
  double* __restrict__ qx = lds;
  double* __restrict__ qy = lds + a.tile_cap;
  double* __restrict__ red = lds + 2 * a.tile_cap;   // [kNW][4]
  int* tab = (int*)(red + 4 * kNW);

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

  double s_hi = a.loss_sum[2 * chain], s_lo = a.loss_sum[2 * chain + 1];
  double loss_prev = (s_hi + s_lo) / S.two_sigma2;

  for (int i = tid; i < a.B.n_sizes; i += kNT) { tab[2 * i] = a.B.bh[i]; tab[2 * i + 1] = a.B.bw[i]; }
  const int64_t rin0 = (int64_t)chain * a.in_stride;
  int n_si = a.size_idx[rin0], n_row = a.centre[2 * rin0], n_col = a.centre[2 * rin0 + 1];
  double n_u = a.u[rin0];
  __syncthreads();

#ifdef GSM_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  for (int s = 0; s < a.n_steps; ++s) {
    STAMP(7);
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
    const rsrc_t r_f = make_rsrc(a.fields + rin * a.field_stride, (uint32_t)(bh * bw) * 8u);

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

    // geometry of the thread's k-th tile cell (recomputed per phase: cheaper than 2 registers per cell)
    STAMP(0);
    int ptid = tid;   // re-laundered at each phase so that the geometry is recomputed, not kept live across phases
    auto cell = [&](int k, int& i, int& lr, int& lc, uint32_t& g, bool& valid, bool& inwin) {
      i = ptid + k * kNT;
      valid = i < ncell;
      lr = (int)__umulhi((uint32_t)i, m_tw);
      lc = i - lr * tw;
      g = (uint32_t)((hr0 + lr) * W + hc0 + lc);
      inwin = valid && (unsigned)(lr - dr) < (unsigned)wh && (unsigned)(lc - dc) < (unsigned)ww;
    };

    // cell slot k of this wave is past the end of the tile for the later slots of smaller blocks: wave-uniform skip
    auto slot_on = [&](int k) { return k * kNT + 64 * wave < ncell; };

    // ---- A: loads, candidate bed, fluxes -> LDS, guard, carried energy of the window -----------------
    // in sub-batches of KB cells: all loads of a sub-batch are in flight together (14 VGPRs per cell)
    double v_new[KT];        // candidate bed of the thread's cells
    uint32_t upd_bits = 0;
    double acc_old = 0.0;
    int guard = 0;
    constexpr int KB = (KT > 4) ? 4 : KT;
#pragma unroll
    for (int kb = 0; kb < KT; kb += KB) {
      if (kb > 0 && !slot_on(kb)) break;   // no cell of this wave in this sub-batch nor in any later one
      double vb[KB], ve[KB], vf[KB];
      double2 A2[KB], B2[KB];
#pragma unroll
      for (int j = 0; j < KB; ++j) {
        const int k = kb + j;
        if (k < KT) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(k, i, lr, lc, g, valid, inwin);
          vb[j] = StateIO<TS>::load(r_bed, valid ? g * (uint32_t)sizeof(TS) : kOOB);
          ve[j] = StateIO<TS>::load(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB);
          vf[j] = ld_f64<2>(r_f, inwin ? (uint32_t)((mr0 + lr - dr) * bw + mc0 + lc - dc) * 8u : kOOB);
          A2[j] = ld_f64x2(r_sA, valid ? g * 16u : kOOB);   // (wupd, surf)
          B2[j] = ld_f64x2(r_sB, valid ? g * 16u : kOOB);   // (velx, vely)
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
            v = v + vf[j] * A2[j].x;
            if (F32) v = (double)(float)v;
          }
          const double thick = A2[j].y - v;
          if (upd && thick <= 0.0) guard = 1;
          v_new[k] = v;
          acc_old += ve[j];
          if (valid) {
            qx[i] = B2[j].x * thick;
            qy[i] = B2[j].y * thick;
          }
        }
      }
      asm volatile("" : "+v"(acc_old));   // pin the partial sum here (otherwise the adds sink to phase R and the
      __builtin_amdgcn_sched_barrier(0);   // loaded energies stay live across phase D)
    }
    STAMP(1);
    __syncthreads();
    STAMP(2);

    // ---- D: residual stencil on the flux tiles ---------------------------------------------------------
    double e_new[KT];
    double acc_new = 0.0;
    asm volatile("" : "+v"(ptid));
    {
      double2 C2[KT];   // (dhdt_mc, smb) of the window cells
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cell(k, i, lr, lc, g, valid, inwin);
        C2[k] = ld_f64x2(r_sC, inwin ? g * 16u : kOOB);
      }
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (!slot_on(k)) { e_new[k] = 0.0; continue; }
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cell(k, i, lr, lc, g, valid, inwin);
        const int r = hr0 + lr, c = hc0 + lc;
        const int il = (c == 0) ? i : i - 1, ir = (c == W - 1) ? i : i + 1;
        const int iu = (r == 0) ? i : i - tw, id = (r == H - 1) ? i : i + tw;
        double e = 0.0;
        if (inwin) {
          const double ddx = qx[ir] - qx[il];
          const double ddy = qy[id] - qy[iu];
          double dx, dy;
          if (FAST_DIV) {
            dx = (ir - il == 2) ? exact_div(ddx, S.two_res, S.rcp_two_res) : exact_div(ddx, S.res, S.rcp_res);
            dy = (id - iu == 2 * tw) ? exact_div(ddy, S.two_res, S.rcp_two_res) : exact_div(ddy, S.res, S.rcp_res);
          } else {
            dx = ddx / ((ir - il == 2) ? S.two_res : S.res);
            dy = ddy / ((id - iu == 2 * tw) ? S.two_res : S.res);
          }
          const double v = ((dx + dy) + C2[k].x) - C2[k].y;
          if (!isnan(v)) e = v * v;
          if (F32) e = (double)(float)e;
        }
        e_new[k] = e;
        acc_new += e;
        if ((k & 1) == 1) __builtin_amdgcn_sched_barrier(0);
      }
    }

    STAMP(3);
    // ---- R: reduce, decide (every thread evaluates the same numbers in the same order) ----------------
    {
      // one sum: the thread's change of energy, or +inf from a thread whose candidate grounds the ice (MCMC.py:1321-1329:
      // loss = inf); energies are never NaN (phase D), so an infinite total can only come from the guard
      double delta = acc_new - acc_old;
      if (guard) delta = INFINITY;
      const double w_delta = wave64_sum(delta);
      if (lane == 0) red[wave] = w_delta;
    }
    STAMP(4);
    __syncthreads();
    STAMP(5);
    const double sd = row16_sum(red[lane & 15]);
    double c_hi, c_err;
    two_sum(s_hi, sd, c_hi, c_err);
    const double c_lo = s_lo + c_err;
    double loss_next = (c_hi + c_lo) / S.two_sigma2;
    if (sd == INFINITY) loss_next = INFINITY;
    const double p_acc = (loss_prev > loss_next) ? 1.0 : fmin(1.0, exp(loss_prev - loss_next));
    const bool acc = (uu <= p_acc);

    // ---- E: commit -------------------------------------------------------------------------------------
    if (acc) {
      asm volatile("" : "+v"(ptid));
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (!slot_on(k)) continue;
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cell(k, i, lr, lc, g, valid, inwin);
        const bool upd = (upd_bits >> k) & 1u;
        StateIO<TS>::store(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB, e_new[k]);
        StateIO<TS>::store(r_bed, upd ? g * (uint32_t)sizeof(TS) : kOOB, v_new[k]);
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
    STAMP(6);
    // End of step.  LDS needs no barrier here: the tiles are rewritten only after every wave has passed the R barrier
    // (its stencil reads are done), and `red` only after the next A barrier.  The global stores of an accepted step must
    // be visible to the next step's loads only if the next halo window touches this window.
    if (acc && has_next && (unsigned)n_si < (unsigned)a.B.n_sizes) {
      const int nbh = tab[2 * n_si], nbw = tab[2 * n_si + 1];
      const int nr0 = max(0, n_row - nbh / 2) - 1, nr1 = min(H, n_row + nbh / 2) + 1;
      const int nc0 = max(0, n_col - nbw / 2) - 1, nc1 = min(W, n_col + nbw / 2) + 1;
      if ((nr0 < r1) && (r0 < nr1) && (nc0 < c1) && (c0 < nc1)) __syncthreads();
    }
  }
  if (tid == 0) {
    a.loss_sum[2 * chain] = s_hi;
    a.loss_sum[2 * chain + 1] = s_lo;
  }
#ifdef GSM_STAMPS
  if (tid == 0 && chain < 4096) for (int q = 0; q < 8; ++q) g_stamps[chain * 8 + q] = st_acc[q];
#endif
}

template <typename TS, int KT>
static hipError_t launch_flux_t(const StepArgs& a, hipStream_t st) {
  const size_t lds = step_flux_lds_bytes(a.tile_cap);
  auto kfast = step_flux_kernel<TS, KT, true>;
  auto kslow = step_flux_kernel<TS, KT, false>;
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  if (a.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.n_chains), dim3(kNT), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.n_chains), dim3(kNT), lds, st, a);
  return hipGetLastError();
}

bool step_flux_supported(const StepArgs& a) {
  return a.tile_cap <= 7 * kNT && step_flux_lds_bytes(a.tile_cap) <= 160 * 1024 && a.S.sA != nullptr &&
         (uint64_t)a.S.H * a.S.W * 16u < 0x80000000ull;
}

hipError_t launch_step_flux(const StepArgs& a, hipStream_t st) {
  if (!step_flux_supported(a)) return hipErrorInvalidValue;
  if (a.f32_state) {
    if (a.tile_cap <= 2 * kNT) return launch_flux_t<float, 2>(a, st);
    if (a.tile_cap <= 4 * kNT) return launch_flux_t<float, 4>(a, st);
    return launch_flux_t<float, 7>(a, st);
  }
  if (a.tile_cap <= 2 * kNT) return launch_flux_t<double, 2>(a, st);
  if (a.tile_cap <= 4 * kNT) return launch_flux_t<double, 4>(a, st);
  return launch_flux_t<double, 7>(a, st);
}

// Folded static operands of the flux kernel (see the header of this file).
__global__ __launch_bounds__(256) void pack_flux_static_kernel(const StaticFields S, double2* sA, double2* sB, double2* sC) {
  const int n = S.H * S.W;
  for (int g = blockIdx.x * 256 + threadIdx.x; g < n; g += gridDim.x * 256) {
    const double w = S.weight ? S.weight[g] : 1.0;
    sA[g] = make_double2(S.upd[g] ? w : __builtin_bit_cast(double, kNoUpdBits), S.surf[g]);
    sB[g] = make_double2(S.velx[g], S.vely[g]);
    sC[g] = make_double2(S.mc[g] == 1 ? S.dhdt[g] : __builtin_nan(""), S.smb[g]);
  }
}

int debug_read_stamps(unsigned long long* out, int n_chains) {
#ifdef GSM_STAMPS
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 8 * (size_t)n_chains) == hipSuccess ? 0 : -3;
#else
  (void)out; (void)n_chains;
  return -4;
#endif
}

hipError_t launch_pack_flux_static(const StaticFields& S, double2* sA, double2* sB, double2* sC, hipStream_t st) {
  int grid = (S.H * S.W + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(pack_flux_static_kernel, dim3(grid), dim3(256), 0, st, S, sA, sB, sC);
  return hipGetLastError();
}

}  // namespace gsm
