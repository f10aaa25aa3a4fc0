// Fused many-chain Metropolis kernel for gfx950 (MI355X), Philox mode: one persistent 1024-thread workgroup per chain
// (one per CU), and per step BOTH the spectral proposal (proposal_device.h; reference gstatsMCMC/MCMC.py:742-778,
// :176-254) AND the Metropolis step (step_flux_kernel.hip; MCMC.py:1263-1360, Topography.py:592-600) inside it.
// The proposal field never leaves the CU: it goes from the MFMA accumulators to an LDS tile and is consumed there.
// Only the chain state touches HBM: read bed (window + halo) and carried energy (window); on accept write both back
// and bump resampled_times -- the algorithmic traffic of SURVEY.md section 8d plus the halo ring.
//
// Per step:
//   P   proposal: DFT tables by LDS-DMA; Philox + Box-Muller + spectral amplitude -> folded coefficients in LDS planes
//   P0  issue the loads of the bed / energy of the window into registers (HBM latency runs under the two MFMA stages;
//       issued after the coefficient phase, whose Box-Muller code needs the registers)
//   S1/S2 two fp64 MFMA inverse-DFT stages, standardise, scale x edge mask -> LDS field tile
//   A   candidate bed = bed + f * weight (where update_mask), thickness guard, fluxes -> two LDS tiles (they overlay the
//       proposal's planes), sum of the carried energy
//   D   5-point stencil on the flux tiles -> new energies;  R  reduction + accept test;  E  commit on accept
// Measured and rejected (round 2): splitting the workgroup into matrix-core waves (DFT stages of step s)
// and coefficient waves (Philox / Box-Muller of step s + 1, handed over through an L2-resident scratch and LDS-DMA).  On
// gfx950 the fp64 MFMA and the fp64 vector FMA have the same peak and do not run side by side on one SIMD: the merged
// interval took as long as the two phases one after the other.
// LDS (80 x 80 blocks): flux tiles / DFT planes 105 KiB + field tile 50 KiB + scratch < 160 KiB.
//
// The arithmetic of a step is the same, operation for operation, as gsm_propose_philox followed by gsm_run_replay
// (tests/test_gpu_philox.py: bit-identical losses, accepts and beds).
#include "gsm_internal.h"
#include "device_util.h"
#include "proposal_device.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace gsm {

using namespace dev;

constexpr int kUPW = 32 / kNW;             // stage-1 units per wave (2 halves x at most 16 output tiles over 16 waves)

#ifdef GSM_STAMPS
// diagnostic build only (GSM_STAMPS=1 at build time): per-workgroup cycle totals of the phases, thread 0
__device__ unsigned long long g_stamps_fused[4096 * 16];
#ifndef GSM_STAMP_TID
#define GSM_STAMP_TID 0
#endif
#define STAMP(slot) do { if (tid == GSM_STAMP_TID) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
    st_acc[slot] += t_ - st_last; st_last = t_; } } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

// work area: flux tiles, overlaid by the DFT planes + the [cos|sin] table during the proposal;
// field tile: overlaid by the c2r table until the field is written
static int fused_work_len(const FusedArgs& a) { return std::max(std::max(2 * a.T.tile_cap, a.P.lds_main), 4 * a.P.lds_x_half + a.P.tab_max); }
static int fused_fld_len(const FusedArgs& a) { return std::max(a.T.B.max_bh * a.T.B.max_bw, a.P.tab_max); }

size_t fused_lds_doubles(const FusedArgs& a) {
  return (size_t)fused_work_len(a) + (size_t)fused_fld_len(a) + 4 * kNW + 32 + 16 + kMathTabDoubles;
}

static_assert(sizeof(PropScalars) == 136, "PropScalars: 136-byte records, read field by field with scalar loads");

// The per-step record (propose_scalars_kernel's output) is read through the constant address space: the address is uniform
// and the memory is never written by this kernel, so every access is a scalar load.  Each phase re-reads the few fields it
// needs through a laundered pointer instead of keeping the whole record (and everything derived from it) in SGPRs for the
// whole step -- the kernel has far more uniform values than scalar registers, and a spilled SGPR comes back through
// v_readlane, a VECTOR instruction (round 1: ~8 % of the vector instructions of a step were such reloads).
typedef const __attribute__((address_space(4))) PropScalars* crec_t;

// copy of a struct that lives in the constant address space (device pass only: the host pass never runs this code)
template <class T>
__device__ __forceinline__ T load_c(const __attribute__((address_space(4))) T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  (void)p;
  return T();
#endif
}

// window of a step, clipped to the grid (MCMC.py:1266-1276), its halo (MCMC.py:1293-1297) and the matching sub-block of f
struct Win {
  int r0, r1, c0, c1;      // window rows / cols [r0, r1) x [c0, c1)
  int mr0, mc0;            // first row / col of the block that lies inside the grid
  int wh, ww;              // window size
  int hr0, hc0, hr1, hc1;  // halo tile
  int tw, ncell;           // tile width, tile cells
  int dr, dc;              // window origin inside the tile (0 or 1)
  uint32_t m_tw;           // magic reciprocal of tw
};
// m_tw: the magic reciprocal of the tile width, from the step's record (computing it here would be a uniform 32-bit
// division: a float reciprocal on the vector unit, v_readfirstlane and ~20 dependent scalar instructions)
__device__ __forceinline__ Win make_win(int H, int W, int row, int col, int bh, int bw, uint32_t m_tw) {
  Win g;
  g.r0 = max(0, row - bh / 2); g.r1 = min(H, row + bh / 2);
  g.c0 = max(0, col - bw / 2); g.c1 = min(W, col + bw / 2);
  g.mr0 = max(bh - g.r1, 0); g.mc0 = max(bw - g.c1, 0);
  g.wh = g.r1 - g.r0; g.ww = g.c1 - g.c0;
  g.hr0 = max(0, g.r0 - 1); g.hr1 = min(H, g.r1 + 1);
  g.hc0 = max(0, g.c0 - 1); g.hc1 = min(W, g.c1 + 1);
  g.tw = g.hc1 - g.hc0;
  g.ncell = (g.hr1 - g.hr0) * g.tw;
  g.m_tw = m_tw;
  g.dr = g.r0 - g.hr0; g.dc = g.c0 - g.hc0;
  return g;
}

template <typename TS, int KT, bool FAST_DIV>
__global__ __launch_bounds__(kNT, 4) void chain_fused_kernel(const FusedArgs fa) {
  constexpr bool F32 = sizeof(TS) == 4;
  // The kernel arguments are re-read from the kernarg segment (scalar loads, scalar cache) in every phase through a
  // laundered pointer: read once, they would be loop invariants that the compiler keeps in -- and spills from -- SGPRs.
  typedef const __attribute__((address_space(4))) FusedArgs* cargs_t;
  auto kargs = [] { cargs_t p = (cargs_t)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p; };
  extern __shared__ double lds[];
  double* __restrict__ qx = lds;
  double* __restrict__ qy = lds + fa.T.tile_cap;
  double* __restrict__ fld = lds + fa.work_len;                    // [max_bh * max_bw]
  double* __restrict__ red = fld + fa.fld_len;                     // [kNW][4]
  double* __restrict__ red2 = red + 4 * kNW;                       // [32] proposal reductions
  double* __restrict__ mtab = red2 + 32 + 16;                      // [kMathTabDoubles] log / sincos table (math_tables.h)
  const int lds_xh4 = 4 * fa.P.lds_x_half;                         // the four coefficient planes; the [cos | sin] table follows

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

  const int chain = blockIdx.x;
  const int n_steps = fa.T.n_steps;
  const int gH = fa.T.S.H, gW = fa.T.S.W;
  // base pointers stay resident (a spilled SGPR comes back with one v_readlane; a reload is a scalar-memory round trip at the
  // head of the phase); the descriptors themselves (4 SGPRs each) are rebuilt where they are used
  const TS* const p_bed = (const TS*)fa.T.beds + (size_t)chain * (size_t)fa.T.S.H * (size_t)fa.T.S.W;
  const TS* const p_en = (const TS*)fa.T.energy + (size_t)chain * (size_t)fa.T.S.H * (size_t)fa.T.S.W;
  const double2* const p_st = fa.T.S.sA;
  const double two_res = fa.T.S.two_res, rcp_two_res = fa.T.S.rcp_two_res, two_sigma2 = fa.T.S.two_sigma2;
  const double rcp_two_sigma2 = fa.T.S.rcp_two_sigma2;
  auto n_cells = [&](cargs_t) { return (uint32_t)gH * (uint32_t)gW; };
  auto rsrc_bed = [&](cargs_t K) { return make_rsrc(p_bed, n_cells(K) * (uint32_t)sizeof(TS)); };
  auto rsrc_en = [&](cargs_t K) { return make_rsrc(p_en, n_cells(K) * (uint32_t)sizeof(TS)); };
  // the three packed static operands are one allocation: sA | sB | sC, selected by the scalar offset of the load
  auto rsrc_st = [&](cargs_t K) { return make_rsrc(p_st, 3u * n_cells(K) * 16u); };
  const crec_t rec0 = (crec_t)(uintptr_t)(fa.P.scalars + (size_t)chain * fa.P.n_steps);   // this chain's records, resident
  const uint64_t seed = fa.P.seeds[chain];
  for (int i = tid; i < kMathTabDoubles; i += kNT) mtab[i] = fa.P.mathtab[i];
  __syncthreads();

  double s_hi = fa.T.loss_sum[2 * chain], s_lo = fa.T.loss_sum[2 * chain + 1];
  double loss_prev = (s_hi + s_lo) / two_sigma2;
  // window of the previous step if it was accepted (its stores may still be in flight), else empty.  Older stores
  // are complete: vmcnt counts in order and every thread has since waited for younger loads of its own.
  int pr0 = 0, pr1 = 0, pc0 = 0, pc1 = 0;

#ifdef GSM_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
#endif
  const NoiseIn no_noise{nullptr, nullptr, nullptr};
  // the fields of a record the proposal stages use
  auto prop_rec = [&](crec_t r) {
    PropScalars q;
    q.scale = r->scale; q.nug = r->nug; q.aa = r->aa; q.m_const = r->m_const; q.m_kappa = r->m_kappa;
    q.bh = r->bh; q.bw = r->bw; q.fy_off = r->fy_off; q.g_off = r->g_off; q.pad = r->pad; q.mask_off = r->mask_off;
    q.m_nc = r->m_nc; q.m_m1 = r->m_m1;
    return q;
  };
  for (int s = 0; s < n_steps; ++s) {
    STAMP(15);
    // Geometry of the thread's tile cells t, t + 1024, ... by (magic) division; ptid is laundered per phase so that the
    // derived values are recomputed instead of being kept live across phases.  (An incremental form without the
    // multiplies measured 1.8 % slower on the same box.)
    int ptid = tid;
    asm volatile("" : "+v"(ptid));
    auto relaunder = [&] { asm volatile("" : "+v"(ptid)); };
    // The step's record pointer and its four window integers stay in scalar registers for the whole step (a spilled SGPR
    // costs one v_readlane; re-reading them costs two dependent scalar-memory round trips at the head of every phase).
    const crec_t rec = rec0 + s;
    const int s_row = rec->row, s_col = rec->col, s_bh = rec->bh, s_bw = rec->bw;
    const uint32_t s_mtw = rec->m_tw;
    auto win_now = [&] { return make_win(gH, gW, s_row, s_col, s_bh, s_bw, s_mtw); };
    auto cell = [&](const Win& G, int W, int k, int& i, int& lr, int& lc, uint32_t& g, bool& valid, bool& inwin) {
      i = ptid + k * kNT;
      valid = i < G.ncell;
      lr = (int)__umulhi((uint32_t)i, G.m_tw);
      lc = i - lr * G.tw;
      g = (uint32_t)((G.hr0 + lr) * W + G.hc0 + lc);
      inwin = valid && (unsigned)(lr - G.dr) < (unsigned)G.wh && (unsigned)(lc - G.dc) < (unsigned)G.ww;
    };

    // ---- P: DFT tables -> LDS by LDS-DMA (in flight during the coefficient phase), folded coefficients -> LDS planes -----
    {
      const cargs_t K = kargs();
      const ProposeArgs pa = load_c(&K->P);
      const PropScalars q = prop_rec(rec);
      const PropGeom pg = prop_geom(pa, q.bh, q.bw);
      dma_to_lds<kNW, 0>(pa.tables + q.fy_off, lds + lds_xh4, 2 * pg.KR * pg.NR, wave, lane);
      dma_to_lds<kNW, 0>(pa.tables + q.g_off, fld, 2 * pg.Kc * pg.M1, wave, lane);
      coef_items<kNT, false>(ptid, 0, pg.nrow * pg.ncol, true, pa, q, pg, seed, pa.step0 + s, lds, pa.lds_x_half, no_noise, mtab);
    }
    STAMP(10);
    // ---- P0: chain state of the window -> registers, in flight during the two MFMA stages ------------------------------
    double vb[KT], ve[KT];
    // tile row | tile col << 8 | valid << 16 | in-window << 17 of the thread's cells: computed once per step, here
    uint32_t rq[KT];
    {
      relaunder();
      const cargs_t K = kargs();
      const Win G = win_now();
      // stores of an earlier accepted step must have landed before this step reads an overlapping halo window
      if ((G.hr0 < pr1) && (pr0 < G.hr1) && (G.hc0 < pc1) && (pc0 < G.hc1)) __syncthreads();
      const rsrc_t r_bed = rsrc_bed(K);
      const rsrc_t r_en = rsrc_en(K);
      const int W = gW;
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        // a slot past the end of the tile for the whole wave: no geometry, but the two loads are still issued (out of
        // range, they return 0) so that the counted wait below holds for every wave
        if (k * kNT + 64 * wave < G.ncell) {
          int i, lr, lc; uint32_t g; bool valid, inwin;
          cell(G, W, k, i, lr, lc, g, valid, inwin);
          vb[k] = StateIO<TS>::load(r_bed, valid ? g * (uint32_t)sizeof(TS) : kOOB);
          ve[k] = StateIO<TS>::load(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB);
          rq[k] = (uint32_t)lr | ((uint32_t)lc << 8) | (valid ? 1u << 16 : 0u) | (inwin ? 1u << 17 : 0u);
        } else {
          rq[k] = 0;
          vb[k] = StateIO<TS>::load(r_bed, kOOB);
          ve[k] = StateIO<TS>::load(r_en, kOOB);
        }
      }
    }
    STAMP(0);
    // The table LDS-DMAs are older than the 2 KT state loads just issued: wait for everything but those (vmcnt counts in
    // order) and for this wave's LDS writes; then a bare barrier.  __syncthreads() would wait vmcnt(0), i.e. for HBM.
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" :: "n"(2 * KT) : "memory");
    STAMP(2);
    // mean of the field = DC coefficient / n (proposal_device.h); read before T^T overlays the plane
    const double dc0 = lds[0];
    {
      v4f64 uc[kUPW], us[kUPW];
      {
        const cargs_t K = kargs();
        const ProposeArgs pa = load_c(&K->P);
        const PropScalars q = prop_rec(rec);
        dft_stage1<kNW, kUPW, true>(wave, ptid & 63, pa, q, prop_geom(pa, q.bh, q.bw), lds, lds + lds_xh4, uc, us);
      }
      STAMP(1);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has finished reading the planes
      STAMP(13);
      {
        const cargs_t K = kargs();
        const ProposeArgs pa = load_c(&K->P);
        relaunder();
        dft_tt_write<kNW, kUPW>(wave, ptid & 63, prop_geom(pa, s_bh, s_bw), lds, uc, us);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- S2: stage 2 (output tile t on wave t), standardise, scale x edge mask -> field tile --------------------------
    {
      const cargs_t K = kargs();
      const ProposeArgs pa = load_c(&K->P);
      const PropScalars q = prop_rec(rec);
      const PropGeom pg = prop_geom(pa, q.bh, q.bw);
      const int bw = q.bw;
      v4f64 fe[1], fo[1];
      double mreg[1][8];
      relaunder();
      const int ln = ptid & 63;
      dft_stage2<kNW, 1, true>(wave, ln, pa, q, pg, lds, fld, fe, fo);
      mask_prefetch<kNW, 1>(wave, ln, pa, q, pg, mreg);
      STAMP(11);
      const double gain = standardise<kNW, 1>(wave, ln, q, pg, dc0, red2, fe, fo);     // contains a barrier
      STAMP(14);
      const bool with_nugget = pa.rf.nugget_max > 0.0;
      emit_field<kNW, 1, true>(wave, ln, pa, q, pg, fe, fo, mreg, gain, with_nugget, fld, [bw](int y, int x) { return y * bw + x; });
      if (with_nugget) {
        __syncthreads();
        relaunder();
        nugget_pass<kNT, false>(ptid, pa, q, pg, seed, pa.step0 + s, no_noise, fld, [bw](int y, int x) { return y * bw + x; }, mtab);
      }
    }
    STAMP(12);
    // static operands of the thread's first phase-A cells: requested before the barrier that phase A begins with
    // (after it every wave would wait for them at the same time)
    constexpr int KBA = (KT > 4) ? 2 : KT;
    double2 A2p[KBA], B2p[KBA];
    {
      const cargs_t K = kargs();
      const Win G = win_now();
      const rsrc_t r_st = rsrc_st(K);
      const uint32_t off_sB = n_cells(K) * 16u;
#pragma unroll
      for (int j = 0; j < KBA; ++j) {
        const uint32_t g = (uint32_t)((G.hr0 + (int)(rq[j] & 0xFFu)) * gW + G.hc0 + (int)((rq[j] >> 8) & 0xFFu));
        const bool valid = (rq[j] >> 16) & 1u;
        A2p[j] = ld_f64x2(r_st, valid ? g * 16u : kOOB, 0u);
        B2p[j] = ld_f64x2(r_st, valid ? g * 16u : kOOB, off_sB);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // field tile complete
    STAMP(3);

    // ---- A: candidate bed, fluxes -> LDS, guard ---------------------------------------------------------------------
    double v_new[KT];
    uint32_t upd_bits = 0;
    int guard = 0;
    relaunder();
    // Geometry for phases A, D and E, computed once per step after the proposal and kept packed in two registers per
    // cell: gq = flat grid index, rq = tile row | tile col << 8 | valid << 16 | in-window << 17 (+1.9 % over recomputing
    // it in every phase, same box).  The arrays are laundered per phase so that only they stay live across phases.
    uint32_t gq[KT];
    auto cellq = [&](int k, int& i, int& lr, int& lc, uint32_t& g, bool& valid, bool& inwin) {
      i = ptid + k * kNT; g = gq[k];
      lr = (int)(rq[k] & 0xFFu); lc = (int)((rq[k] >> 8) & 0xFFu);
      valid = (rq[k] >> 16) & 1u; inwin = (rq[k] >> 17) & 1u;
    };
    auto launderq = [&] {
#pragma unroll
      for (int k = 0; k < KT; ++k) asm volatile("" : "+v"(gq[k]), "+v"(rq[k]));
    };
    // Cell slot k of this WAVE holds tile cells 64 * wave + 1024 * k ...: past the end of the tile for the later slots of
    // smaller blocks (on average 2.7 of the 7 slots).  Wave-uniform, so a scalar branch skips the whole slot in the
    // stencil and commit phases.
    double2 C2[KT];
    double acc_old = 0.0;
    {
      const cargs_t K = kargs();
      const Win G = win_now();
      const int bw = s_bw, W = gW;
      const uint32_t off_sB = n_cells(K) * 16u, off_sC = 2u * off_sB;
      auto slot_on = [&](int k) { return k * kNT + 64 * wave < G.ncell; };
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        gq[k] = (uint32_t)((G.hr0 + (int)(rq[k] & 0xFFu)) * W + G.hc0 + (int)((rq[k] >> 8) & 0xFFu));
      }
      const rsrc_t r_st = rsrc_st(K);
      constexpr int KB = KBA;                    // cells per sub-batch of phase A (2: +0.8 % over 4, same box)
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
            if (kb == 0) { A2[j] = A2p[j]; B2[j] = B2p[j]; }
            else {
              A2[j] = ld_f64x2(r_st, valid ? g * 16u : kOOB, 0u);       // (wupd, surf)
              B2[j] = ld_f64x2(r_st, valid ? g * 16u : kOOB, off_sB);   // (velx, vely)
            }
            vf[j] = inwin ? fld[(G.mr0 + lr - G.dr) * bw + G.mc0 + lc - G.dc] : 0.0;
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
      launderq();
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cellq(k, i, lr, lc, g, valid, inwin);
        if (!slot_on(k)) { C2[k] = make_double2(0.0, 0.0); continue; }
        C2[k] = ld_f64x2(r_st, inwin ? g * 16u : kOOB, off_sC);
      }
    }
    STAMP(4);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // LDS tiles complete; the loads above stay in flight
    STAMP(5);

    // ---- D: residual stencil on the flux tiles ---------------------------------------------------------
    double e_new[KT];
    double acc_new = 0.0;
    launderq();
    {
      const int H = gH, W = gW;
      const Win G = win_now();
      const int tw = G.tw, hr0 = G.hr0, hc0 = G.hc0;
      auto slot_on = [&](int k) { return k * kNT + 64 * wave < G.ncell; };
      // interior step (a halo ring on all four sides, ~5 steps in 6): no window cell touches a grid border, every
      // difference is central.  The general form applies np.gradient's one-sided edge rules.
      const bool interior = (G.hr0 < G.r0) && (G.hr1 > G.r1) && (G.hc0 < G.c0) && (G.hc1 > G.c1);
      auto phase_d = [&](auto interior_tag) {
        constexpr bool INTERIOR = decltype(interior_tag)::value;
        double res = 0.0, rcp_res = 0.0;                                   // border windows only: read where they are needed
        if (!INTERIOR) { const cargs_t K = kargs(); res = K->T.S.res; rcp_res = K->T.S.rcp_res; }
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
              if (FAST_DIV) { dx = exact_div(ddx, two_res, rcp_two_res); dy = exact_div(ddy, two_res, rcp_two_res); }
              else { dx = ddx / two_res; dy = ddy / two_res; }
            } else {
              const int r = hr0 + lr, c = hc0 + lc;
              const int il = (c == 0) ? i : i - 1, ir = (c == W - 1) ? i : i + 1;
              const int iu = (r == 0) ? i : i - tw, id = (r == H - 1) ? i : i + tw;
              const double ddx = qx[ir] - qx[il];
              const double ddy = qy[id] - qy[iu];
              if (FAST_DIV) {
                dx = (ir - il == 2) ? exact_div(ddx, two_res, rcp_two_res) : exact_div(ddx, res, rcp_res);
                dy = (id - iu == 2 * tw) ? exact_div(ddy, two_res, rcp_two_res) : exact_div(ddy, res, rcp_res);
              } else {
                dx = ddx / ((ir - il == 2) ? two_res : res);
                dy = ddy / ((id - iu == 2 * tw) ? two_res : res);
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
    }
    STAMP(6);
    // ---- R: reduce, decide (every thread evaluates the same numbers in the same order) ----------------
    {
      // one sum: the thread's change of energy, or +inf from a thread whose candidate grounds the ice (MCMC.py:1321-1329:
      // loss = inf); energies are never NaN (phase D), so an infinite total can only come from the guard
      double delta = acc_new - acc_old;
      if (guard) delta = INFINITY;
      const double w_delta = wave64_sum(delta);
      if (lane == 0) red[wave] = w_delta;
    }
    STAMP(7);
    __syncthreads();
    STAMP(8);
    const double sd = row16_sum(red[lane & 15]);
    double c_hi, c_err;
    two_sum(s_hi, sd, c_hi, c_err);
    const double c_lo = s_lo + c_err;
    const cargs_t Ke = kargs();
    double loss_next = FAST_DIV ? exact_div(c_hi + c_lo, two_sigma2, rcp_two_sigma2) : (c_hi + c_lo) / two_sigma2;
    if (sd == INFINITY) loss_next = INFINITY;
    // every thread holds the same numbers: a scalar branch skips the exponential of a downhill step
    double p_acc = 1.0;
    if (!__builtin_amdgcn_readfirstlane((int)(loss_prev > loss_next))) p_acc = fmin(1.0, exp(loss_prev - loss_next));
    const bool acc = (rec->u <= p_acc);

    // ---- E: commit -------------------------------------------------------------------------------------
    if (acc) {
      launderq();
      const Win G = win_now();
      auto slot_on = [&](int k) { return k * kNT + 64 * wave < G.ncell; };
      const rsrc_t r_bed = rsrc_bed(Ke);
      const rsrc_t r_en = rsrc_en(Ke);
#pragma unroll
      for (int k = 0; k < KT; ++k) {
        if (!slot_on(k)) continue;
        int i, lr, lc; uint32_t g; bool valid, inwin;
        cellq(k, i, lr, lc, g, valid, inwin);
        const bool upd = (upd_bits >> k) & 1u;
        StateIO<TS>::store(r_en, inwin ? g * (uint32_t)sizeof(TS) : kOOB, e_new[k]);
        StateIO<TS>::store(r_bed, upd ? g * (uint32_t)sizeof(TS) : kOOB, v_new[k]);
      }
      two_sum(c_hi, c_lo, s_hi, s_lo);
      loss_prev = loss_next;
      pr0 = G.r0; pr1 = G.r1; pc0 = G.c0; pc1 = G.c1;
    } else {
      pr0 = pr1 = pc0 = pc1 = 0;
    }
    STAMP(9);
    if (tid == 0) {
      const StepArgs a = load_c(&Ke->T);
      const int64_t rout = (int64_t)chain * a.rec_stride + a.rec_offset + s;
      a.loss[rout] = loss_prev;
      a.accept[rout] = acc ? 1 : 0;
      if (a.blocks) { a.blocks[4 * rout] = s_row; a.blocks[4 * rout + 1] = s_col; a.blocks[4 * rout + 2] = s_bh; a.blocks[4 * rout + 3] = s_bw; }
    }
  }
  if (tid == 0) {
    fa.T.loss_sum[2 * chain] = s_hi;
    fa.T.loss_sum[2 * chain + 1] = s_lo;
  }
#ifdef GSM_STAMPS
  if (tid == GSM_STAMP_TID && chain < 4096) for (int q = 0; q < 16; ++q) g_stamps_fused[chain * 16 + q] = st_acc[q];
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
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  if (a.T.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.T.n_chains), dim3(kNT), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.T.n_chains), dim3(kNT), lds, st, a);
  return hipGetLastError();
}

bool fused_supported(const FusedArgs& a) {
  if (a.T.strip && a.P.tab_max > 0) return true;
  return step_flux_supported(a.T) && a.P.tab_max > 0 && fused_lds_doubles(a) * sizeof(double) <= 160 * 1024 &&
         2 * a.P.tiles1_max <= kNW * kUPW && a.P.tiles2_max <= 16;
}

// ---- resampled counts of a launch (MCMC.py:1347 resampled_times[window] += update_mask on every accepted step) ------------
// The chain kernel does not touch `resampled`: per-cell atomics on lines that live in HBM sit in the in-order vector-memory
// pipeline of the CU for a memory latency (+1.9 % without them).  The counts of a launch follow from its records and accept
// flags alone: cell (r, c) gains the number of accepted steps whose window covers it.  One workgroup per (chain, band of
// rows, tile of columns): every accepted window adds +1 / -1 at its first / past-the-last column in each of its rows of the
// band (LDS atomics), a running sum along each row turns that into counts, and the counts are added to the plane where
// update_mask is set.  Integer arithmetic: the same numbers as one atomic per accepted step and cell.
constexpr int kRsThreads = 256, kRsColTile = 1024, kRsSeg = 64;
__global__ __launch_bounds__(kRsThreads) void resampled_from_records_kernel(const FusedArgs fa, const int band_rows) {
  extern __shared__ int cnt[];                 // [band_rows][wt + 1] differences -> counts, then [band_rows][n_seg] segment totals
  const int chain = blockIdx.x, band0 = blockIdx.y * band_rows, col0 = blockIdx.z * kRsColTile;
  const int H = fa.T.S.H, W = fa.T.S.W;
  const int band1 = min(H, band0 + band_rows), col1 = min(W, col0 + kRsColTile);
  const int wt = col1 - col0, ld = wt + 1, nb = band1 - band0, n_seg = (wt + kRsSeg - 1) / kRsSeg;
  int* __restrict__ seg_tot = cnt + band_rows * (min(W, kRsColTile) + 1);
  const int tid = threadIdx.x;
  for (int i = tid; i < nb * ld; i += kRsThreads) cnt[i] = 0;
  __syncthreads();
  // one thread per step of the launch
  const PropScalars* __restrict__ recs = fa.P.scalars + (size_t)chain * fa.P.n_steps;
  const uint8_t* __restrict__ acc = fa.T.accept + (int64_t)chain * fa.T.rec_stride + fa.T.rec_offset;
  for (int s = tid; s < fa.T.n_steps; s += kRsThreads) {
    if (!acc[s]) continue;
    const int row = recs[s].row, col = recs[s].col, bh = recs[s].bh, bw = recs[s].bw;
    const int r0 = max(band0, max(0, row - bh / 2)), r1 = min(band1, min(H, row + bh / 2));   // window rows within the band
    const int c0 = max(col0, max(0, col - bw / 2)), c1 = min(col1, min(W, col + bw / 2));
    if (c0 >= c1) continue;
    for (int r = r0; r < r1; ++r) {
      atomicAdd(&cnt[(r - band0) * ld + (c0 - col0)], 1);
      atomicAdd(&cnt[(r - band0) * ld + (c1 - col0)], -1);
    }
  }
  __syncthreads();
  // running sums: one thread per (row, 64-column segment), consecutive threads on consecutive rows (row stride wt + 1 words)
  for (int t = tid; t < nb * n_seg; t += kRsThreads) {
    const int sg = t / nb, r = t - sg * nb;
    const int c_lo = sg * kRsSeg, c_hi = min(wt, c_lo + kRsSeg);
    int run = 0;
    for (int c = c_lo; c < c_hi; ++c) { run += cnt[r * ld + c]; cnt[r * ld + c] = run; }
    seg_tot[r * n_seg + sg] = run;
  }
  __syncthreads();
  uint32_t* __restrict__ plane = fa.T.resampled + (size_t)chain * H * W;
  const uint8_t* __restrict__ upd = fa.T.S.upd;
  auto count_at = [&](int r, int c) {
    int n = cnt[r * ld + c];
    for (int sg = 0; sg < c / kRsSeg; ++sg) n += seg_tot[r * n_seg + sg];
    return n;
  };
  if ((W & 3) == 0) {                          // four cells per thread and pass: 16-byte loads and stores
    const int wq = wt >> 2;
    for (int i = tid; i < nb * wq; i += kRsThreads) {
      const int r = i / wq, c = (i - r * wq) * 4;
      const size_t g = (size_t)(band0 + r) * W + col0 + c;
      const int base = count_at(r, c) - cnt[r * ld + c];        // segment offset (c .. c + 3 lie in one segment)
      const uchar4 m = *(const uchar4*)(upd + g);
      const int n0 = m.x ? base + cnt[r * ld + c] : 0, n1 = m.y ? base + cnt[r * ld + c + 1] : 0;
      const int n2 = m.z ? base + cnt[r * ld + c + 2] : 0, n3 = m.w ? base + cnt[r * ld + c + 3] : 0;
      if ((n0 | n1 | n2 | n3) == 0) continue;
      uint4 v = *(uint4*)(plane + g);
      v.x += (uint32_t)n0; v.y += (uint32_t)n1; v.z += (uint32_t)n2; v.w += (uint32_t)n3;
      *(uint4*)(plane + g) = v;
    }
  } else {
    for (int i = tid; i < nb * wt; i += kRsThreads) {
      const int r = i / wt, c = i - r * wt;
      const int n = count_at(r, c);
      const size_t g = (size_t)(band0 + r) * W + col0 + c;
      if (n != 0 && upd[g]) plane[g] += (uint32_t)n;
    }
  }
}

hipError_t launch_resampled_from_records(const FusedArgs& a, hipStream_t st) {
  const int W = a.T.S.W, H = a.T.S.H;
  const int wt = std::min(W, kRsColTile);
  // ~32 KiB of LDS per workgroup: several workgroups per CU hide each other's memory latency
  const int band_rows = std::max(1, std::min(H, (int)((32 * 1024 / sizeof(int)) / (size_t)(wt + 1))));
  const dim3 grid(a.T.n_chains, (H + band_rows - 1) / band_rows, (W + kRsColTile - 1) / kRsColTile);
  const size_t lds = (size_t)band_rows * (size_t)(wt + 1 + (wt + kRsSeg - 1) / kRsSeg) * sizeof(int);
  hipLaunchKernelGGL(resampled_from_records_kernel, grid, dim3(kRsThreads), lds, st, a, band_rows);
  return hipGetLastError();
}

// One launch: propose_scalars_kernel for all steps must have filled a.P.scalars (n_chains x a.P.n_steps records).
hipError_t launch_chain_fused(const FusedArgs& a_in, hipStream_t st) {
  if (!fused_supported(a_in)) return hipErrorInvalidValue;
  if (a_in.T.strip) {                               // two chains per CU: chain_strip_kernel.hip
    const hipError_t es = launch_chain_strip(a_in, st);
    if (es != hipSuccess) return es;
    return launch_resampled_from_records(a_in, st);
  }
  FusedArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("GSM_PROPOSE_DBG"); dbg = v ? atoi(v) : 0; } a.P.dbg = dbg; }   // diagnostics only
  a.work_len = fused_work_len(a);
  a.fld_len = fused_fld_len(a);
  hipError_t e;
  if (a.T.f32_state) {
    if (a.T.tile_cap <= 2 * kNT) e = launch_fused_t<float, 2>(a, st);
    else if (a.T.tile_cap <= 4 * kNT) e = launch_fused_t<float, 4>(a, st);
    else e = launch_fused_t<float, 7>(a, st);
  } else {
    if (a.T.tile_cap <= 2 * kNT) e = launch_fused_t<double, 2>(a, st);
    else if (a.T.tile_cap <= 4 * kNT) e = launch_fused_t<double, 4>(a, st);
    else e = launch_fused_t<double, 7>(a, st);
  }
  if (e != hipSuccess) return e;
  return launch_resampled_from_records(a, st);      // the launch's accept flags and records -> resampled counts
}

}  // namespace gsm
