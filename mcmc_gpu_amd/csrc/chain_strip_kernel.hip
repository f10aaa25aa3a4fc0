// Many-chain Metropolis kernels in "strip" form for gfx950 (MI355X): one persistent 512-thread workgroup per chain, TWO
// chains per CU (<= 80 KiB of LDS and <= 128 registers each), all steps of a launch looped inside the kernel.
//
//   chain_strip_kernel   Philox mode (gsm_run_philox): per step the spectral proposal (proposal_device.h; reference
//                        gstatsMCMC/MCMC.py:742-778, :176-254) AND the Metropolis step (strip_step.h; MCMC.py:1263-1360,
//                        Topography.py:592-600).  The proposal field goes from the MFMA accumulators to an LDS tile that
//                        overlays the DFT work area and is consumed there; only the chain state touches HBM.
//   step_strip_kernel    replay mode (gsm_run_replay) and the two-kernel pipeline: the same step with the field read from
//                        HBM.  Same functions, same order of every sum: bit-identical to the fused kernel on the same
//                        proposals (tests/test_gpu_fused.py, test_gpu_philox.py).
//
// Why two chains per CU: chain_fused_kernel (1024 threads, 159 KiB of LDS: two flux tiles or four coefficient planes plus
// the DFT tables, one workgroup per CU) leaves 38 % of its wave-cycles waiting at barriers and on memory with nothing
// else to run.  Here the step needs no LDS (strip_step.h), the DFT operands come from 1-D twiddle tables (4 KiB of LDS instead of 66), and the field tile shares the
// proposal's work area, so a second workgroup fits and fills those waits.
//
// Per step of chain_strip_kernel:
//   P    Philox + Box-Muller + spectral amplitude -> four folded coefficient planes in LDS                -- barrier
//   S1   stage-1 DFT on the matrix cores (v_mfma_f64_16x16x4_f64), results in registers               -- barrier
//        T^T overlays the planes                                                                          -- barrier
//   S2   stage-2 DFT, standardise (one barrier inside), scale x edge mask -> field tile (overlays T^T); the loads of the
//        chain state of the strip are issued                                                              -- barrier
//   A/D  strip pass (registers only)
//   R    DPP wave reduction, 8 partials through LDS                                                       -- barrier
//   E    accept test (every thread, same numbers), commit on accept
#include "gsm_internal.h"
#include "device_util.h"
#include "proposal_device.h"
#include "strip_step.h"
#include <math.h>
#include <stdlib.h>
#include <algorithm>
#include <type_traits>

namespace gsm {

using namespace dev;
using strip::kST;
using strip::kSW;
using strip::kNR;

constexpr int kSUPW = 32 / kSW;   // stage-1 units per wave
constexpr int kSMAXT = 16 / kSW;  // stage-2 tiles per wave

static int strip_tile_len(const BlockTable& B) { return (B.max_bh + 2) * (B.max_bw + 2); }   // candidate-bed tile: block + halo ring
static int strip_main_len(const FusedArgs& a) { return std::max(a.P.lds_main, strip_tile_len(a.T.B)); }
constexpr int kStripAux = 16 + 32 + 16 + kMathTabDoubles + 4 * kT1S + kMask1D;     // wave partials (+ the carried sums), proposal reductions, math table, 1-D twiddle tables, 1-D edge mask
size_t fused_strip_lds_doubles(const FusedArgs& a) { return (size_t)strip_main_len(a) + kStripAux; }

// One decision per (static fields, block table): both strip kernels or neither (the sums of a step are taken in another
// order than in the flux-tile kernels, and fused == propose + replay must hold bit for bit).  GSM_STRIP=0 switches them off.
bool strip_table_ok(const StaticFields& S, const BlockTable& B, int lds_main, int tiles1_max, int tiles2_max) {
  static int on = -1;
  if (on < 0) { const char* v = getenv("GSM_STRIP"); on = v ? atoi(v) : 1; }
  const size_t lds = ((size_t)std::max(lds_main, strip_tile_len(B)) + kStripAux) * sizeof(double);
  // The strip pass streams the static operands row by row, one row ahead: that hides an L2 hit, not a trip to the Infinity Cache
  // or HBM.  The three packed planes (48 bytes per cell) must fit an XCD's 4 MiB L2 (256 x 256: 3 MiB); on larger grids the
  // flux-tile kernels, which request a whole tile's operands at once, are faster (1024 x 1024, fp32 state: 12.8 M against 10.4 M
  // chain-steps/s; 512 x 512 behind the Cholesky generator: 2.77 M against 2.75 M).
  return on && S.sA != nullptr && (uint64_t)S.H * S.W * 48u <= (4ull << 20) && B.n_sizes <= 64 && strip::table_ok(B.max_bh, B.max_bw) &&
         B.max_bh <= kT1S && B.max_bw <= kT1S && (B.masks == nullptr || B.mask1d != nullptr) && lds <= 80 * 1024 && 2 * tiles1_max <= kSW * kSUPW && tiles2_max <= kSW * kSMAXT;
}

typedef const __attribute__((address_space(4))) PropScalars* srec_t;

template <class T>
__device__ __forceinline__ T load_cs(const __attribute__((address_space(4))) T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *p;
#else
  (void)p;
  return T();
#endif
}

// accept test of a step (MCMC.py:1331-1336) on the carried compensated sum: every thread evaluates the same numbers
template <bool FAST_DIV>
__device__ __forceinline__ bool decide(const double sd, const double s_hi, const double s_lo, const double two_sigma2, const double rcp_two_sigma2,
                                       const double loss_prev, const double u, double& c_hi, double& c_lo, double& loss_next) {
  double c_err;
  two_sum(s_hi, sd, c_hi, c_err);
  c_lo = s_lo + c_err;
  loss_next = FAST_DIV ? exact_div(c_hi + c_lo, two_sigma2, rcp_two_sigma2) : (c_hi + c_lo) / two_sigma2;
  if (sd == INFINITY) loss_next = INFINITY;
  double p_acc = 1.0;
  if (!__builtin_amdgcn_readfirstlane((int)(loss_prev > loss_next))) p_acc = fmin(1.0, exp(loss_prev - loss_next));
  return u <= p_acc;
}

// NOISE: the coefficients are formed from caller-supplied white-noise planes (the reference's own draws, advanced on the device
// by gsm_draw_pcg64) instead of Philox draws: gsm_run_noise, the 'pcg64' draw mode with synthesis and step in one kernel.
template <typename TS, bool FAST_DIV, bool NOISE>
__global__ __launch_bounds__(kST, 4) void chain_strip_kernel(const FusedArgs fa) {
  typedef const __attribute__((address_space(4))) FusedArgs* cargs_t;
  auto kargs = [] { cargs_t p = (cargs_t)__builtin_amdgcn_kernarg_segment_ptr(); asm volatile("" : "+s"(p)); return p; };
  extern __shared__ double lds[];                                  // planes -> T^T -> field tile
  double* __restrict__ red = lds + fa.work_len;                    // [16] wave partials of the step (8 used, the rest 0)
  double* __restrict__ red2 = red + 16;                            // [32] proposal reductions, [16] carried sums: s_hi, s_lo, loss_prev
  double* __restrict__ carry = red2 + 32;
  double* __restrict__ mtab = red2 + 32 + 16;                      // [kMathTabDoubles] log / sincos table (math_tables.h)
  double* __restrict__ t1 = mtab + kMathTabDoubles;                // [4][kT1S] 1-D twiddle tables of the step's block: cos, sin (height), cos, -sin (width)
  double* __restrict__ m1 = t1 + 4 * kT1S;                         // [kMask1D] edge mask of the step's block size by distance to the block border

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x;
  const int n_steps = fa.T.n_steps;
  const int gH = fa.T.S.H, gW = fa.T.S.W;
  const TS* const p_bed = (const TS*)fa.T.beds + (size_t)chain * (size_t)gH * (size_t)gW;
  const TS* const p_en = (const TS*)fa.T.energy + (size_t)chain * (size_t)gH * (size_t)gW;
  const double2* const p_st = fa.T.S.sA;
  const double two_sigma2 = fa.T.S.two_sigma2, rcp_two_sigma2 = fa.T.S.rcp_two_sigma2;
  const uint32_t n_cells = (uint32_t)gH * (uint32_t)gW;
  const srec_t rec0 = (srec_t)(uintptr_t)(fa.P.scalars + (size_t)chain * fa.P.n_steps);
  const uint64_t seed = NOISE ? 0ull : fa.P.seeds[chain];
  const int dbg = fa.P.dbg;      // diagnostics only (GSM_PROPOSE_DBG): 128 / 256 / 512 / 1024 skip phase D / phase A / the state loads / the commit
  for (int i = tid; i < kMathTabDoubles; i += kST) mtab[i] = fa.P.mathtab[i];
  if (tid < 16) red[tid] = 0.0;
  __syncthreads();

  // The carried compensated sum and the current loss are the same numbers in every thread and are needed once per step, at
  // the accept test: they live in LDS between the tests (thread 0 writes them back after an accepted step) instead of in six
  // vector registers of every thread through the whole step.
  if (tid == 0) {
    const double h0 = fa.T.loss_sum[2 * chain], l0 = fa.T.loss_sum[2 * chain + 1];
    carry[0] = h0; carry[1] = l0; carry[2] = (h0 + l0) / two_sigma2;
  }
  __syncthreads();
  int pr0 = 0, pr1 = 0, pc0 = 0, pc1 = 0;      // window of the previous step if it was accepted, else empty
  const int64_t noise0 = NOISE ? (int64_t)chain * fa.P.n_steps * fa.noise_stride : 0;     // this chain's first record's planes
  auto prop_rec = [&](srec_t r) {
    PropScalars q;
    q.scale = r->scale; q.nug = r->nug; q.aa = r->aa; q.m_const = r->m_const; q.m_kappa = r->m_kappa;
    q.bh = r->bh; q.bw = r->bw; q.fy_off = r->fy_off; q.g_off = r->g_off; q.pad = r->pad; q.mask_off = r->mask_off;
    q.m_nc = r->m_nc; q.m_m1 = r->m_m1; q.m_bh = r->m_bh; q.m_bw = r->m_bw;
    return q;
  };

  for (int s = 0; s < n_steps; ++s) {
    int ptid = tid;
    asm volatile("" : "+v"(ptid));
    auto relaunder = [&] { asm volatile("" : "+v"(ptid)); };
    const srec_t rec = rec0 + s;
    const int s_row = rec->row, s_col = rec->col, s_bh = rec->bh, s_bw = rec->bw;

    // ---- P: twiddle tables of this block shape and folded coefficients -> LDS ------------------------------------------------
    {
      const cargs_t K = kargs();
      const double* __restrict__ g1 = K->P.tab1d;
      if (tid >= kST - kMask1D) m1[tid - (kST - kMask1D)] = K->P.B.mask1d[rec->si * kMask1D + tid - (kST - kMask1D)];
      if (tid < 2 * s_bh) t1[(tid < s_bh) ? tid : kT1S + tid - s_bh] = g1[rec->t1h_off + tid];
      else if (tid >= 256 && tid < 256 + 2 * s_bw) {
        const int i = tid - 256;
        const double v = g1[rec->t1w_off + i];
        t1[2 * kT1S + ((i < s_bw) ? i : kT1S + i - s_bw)] = (i < s_bw) ? v : -v;
      }
    }
    {
      const cargs_t K = kargs();
      const ProposeArgs pa = load_cs(&K->P);
      const PropScalars q = prop_rec(rec);
      const PropGeom pg = prop_geom(pa, q.bh, q.bw);
      NoiseIn nz{nullptr, nullptr, nullptr};
      if (NOISE) {
        typedef const __attribute__((address_space(4))) FusedArgs* cfa_t;
        const cfa_t F = (cfa_t)K;
        const int64_t o = noise0 + (int64_t)s * F->noise_stride;
        nz.re = F->noise_re + o; nz.im = F->noise_im + o;
      }
      coef_items<kST, NOISE, true>(ptid, 0, pg.nrow * pg.ncol, true, pa, q, pg, seed, pa.step0 + s, lds, pa.lds_x_half, nz, mtab, (pa.parseval && !((q.bh | q.bw) & 1)) ? red2 + 16 : nullptr);
    }
    // Stores of an accepted step must have landed before a later step reads an overlapping window.  They were issued a
    // whole coefficient phase ago; the wait is free, and the barriers that follow order it across the waves.
    {
      const strip::Window G = strip::make_window(gH, gW, s_row, s_col, s_bh, s_bw);
      if ((G.r0 - 1 < pr1) && (pr0 < G.r0 + G.wh + 1) && (G.c0 - 1 < pc1) && (pc0 < G.c0 + G.ww + 1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const double dc0 = lds[0];                 // mean of the field = DC coefficient / n; read before T^T overlays the plane
    {
      v4f64 uc[kSUPW], us[kSUPW];
      {
        const cargs_t K = kargs();
        const ProposeArgs pa = load_cs(&K->P);
        const PropScalars q = prop_rec(rec);
        dft_stage1<kSW, kSUPW, 2>(wave, ptid & 63, pa, q, prop_geom(pa, q.bh, q.bw), lds, t1, uc, us);
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // every wave has finished reading the planes
      {
        const cargs_t K = kargs();
        const ProposeArgs pa = load_cs(&K->P);
        relaunder();
        if (!(dbg & 2048)) dft_tt_write<kSW, kSUPW>(wave, ptid & 63, prop_geom(pa, s_bh, s_bw), lds, uc, us);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- S2: stage 2, standardise, scale x edge mask -> field tile (over T^T: every wave is past its stage-2 reads once the
    // barrier inside standardise() has been passed) -------------------------------------------------------------------------
    {
      const cargs_t K = kargs();
      const ProposeArgs pa = load_cs(&K->P);
      const PropScalars q = prop_rec(rec);
      const PropGeom pg = prop_geom(pa, q.bh, q.bw);
      const int bw = q.bw;
      v4f64 fe[kSMAXT], fo[kSMAXT];
      const double mreg[kSMAXT][8] = {};        // not used: the edge mask comes from the 1-D table in LDS (m1)
      relaunder();
      const int ln = ptid & 63;
      dft_stage2<kSW, kSMAXT, 2>(wave, ln, pa, q, pg, lds, t1 + 2 * kT1S, fe, fo);
      double gain = 1.0;
      if (dbg & 4096) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      else gain = standardise<kSW, kSMAXT>(wave, ln, q, pg, dc0, red2, fe, fo, (pa.parseval && !((q.bh | q.bw) & 1)) ? red2 + 16 : nullptr);     // contains a barrier; even shapes: variance from the spectrum
      NoiseIn nz{nullptr, nullptr, nullptr};
      if (NOISE) {
        typedef const __attribute__((address_space(4))) FusedArgs* cfa_t;
        const cfa_t F = (cfa_t)K;
        const double* nb = F->noise_nug;
        if (nb) nz.nug = nb + noise0 + (int64_t)s * F->noise_stride;
      }
      const bool with_nugget = NOISE ? (nz.nug != nullptr) : (pa.rf.nugget_max > 0.0);
      emit_field<kSW, kSMAXT, 2>(wave, ln, pa, q, pg, fe, fo, mreg, gain, with_nugget, lds, [bw](int y, int x) { return (y + 1) * (bw + 2) + x + 1; }, m1);
      if (with_nugget) {
        __syncthreads();
        relaunder();
        nugget_pass<kST, NOISE>(ptid, pa, q, pg, seed, pa.step0 + s, nz, lds, [bw](int y, int x) { return (y + 1) * (bw + 2) + x + 1; }, mtab);
      }
    }
    // ---- the step: strip geometry, chain state -> registers, pass, reduce, decide, commit -------------------------------------
    relaunder();
    const strip::Window G = strip::make_window(gH, gW, s_row, s_col, s_bh, s_bw);
    const strip::Cfg cfg = strip::config(G.wh, G.ww);
    const strip::Lane L = strip::lane_setup(ptid & 63, wave, cfg, G, gH, gW);
    const rsrc_t r_bed = make_rsrc(p_bed, n_cells * (uint32_t)sizeof(TS));
    const rsrc_t r_en = make_rsrc(p_en, n_cells * (uint32_t)sizeof(TS));
    const rsrc_t r_st = make_rsrc(p_st, 3u * n_cells * 16u);
    uint32_t upd_bits;
    double acc_old;
    bool guard;
    {
      double vb[kNR + 2], ve[kNR];
      double2 a2[strip::kNA];
      if (dbg & 512) {
#pragma unroll
        for (int i = 0; i < kNR + 2; ++i) vb[i] = 0.0;
#pragma unroll
        for (int i = 0; i < kNR; ++i) ve[i] = 0.0;
#pragma unroll
        for (int i = 0; i < strip::kNA; ++i) a2[i] = make_double2(0.0, 0.0);
      } else if (G.interior) strip::load_state<TS, true>(L, cfg.n, gW, r_bed, r_en, r_st, vb, ve, a2);
      else strip::load_state<TS, false>(L, cfg.n, gW, r_bed, r_en, r_st, vb, ve, a2);
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");      // field tile complete; the state loads stay in flight
      const int ts = s_bw + 2;
      auto field = [&](int jj, bool) { return lds[L.tidx + jj * ts]; };      // unconditional: a lane without the cell discards the value
      if (dbg & 256) { upd_bits = 0u; acc_old = 0.0; guard = false; }
      else if (G.interior) strip::phase_a<TS, true>(L, cfg.n, gW, s_bw, r_st, field, lds, vb, ve, a2, upd_bits, acc_old, guard);
      else strip::phase_a<TS, false>(L, cfg.n, gW, s_bw, r_st, field, lds, vb, ve, a2, upd_bits, acc_old, guard);
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // candidate-bed tile complete
    double en[kNR], vn[kNR];
    {
      double acc_new;
      const cargs_t K = kargs();
      const strip::StepConsts SC{K->T.S.res, K->T.S.rcp_res, K->T.S.two_res, K->T.S.rcp_two_res};
      if (dbg & 128) {
        acc_new = 0.0;
#pragma unroll
        for (int i = 0; i < kNR; ++i) en[i] = 0.0;
      } else if (G.interior) strip::phase_d<TS, FAST_DIV, true>(L, cfg.n, gW, s_bw, r_st, n_cells * 16u, SC, lds, en, acc_new);
      else strip::phase_d<TS, FAST_DIV, false>(L, cfg.n, gW, s_bw, r_st, n_cells * 16u, SC, lds, en, acc_new);
      double delta = acc_new - acc_old;
      if (guard) delta = INFINITY;
      const double w_delta = wave64_sum(delta);
      if (lane == 0) red[wave] = w_delta;
    }
    if (dbg & 1024) {
#pragma unroll
      for (int i = 0; i < kNR; ++i) vn[i] = 0.0;
    } else strip::read_candidate(L, cfg.n, s_bw, lds, vn);
    const double s_hi = carry[0], s_lo = carry[1];
    double loss_prev = carry[2];
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const double sd = strip::waves_sum(red, lane);
    double c_hi, c_lo, loss_next;
    const bool acc = decide<FAST_DIV>(sd, s_hi, s_lo, two_sigma2, rcp_two_sigma2, loss_prev, rec->u, c_hi, c_lo, loss_next);
    if (acc && !(dbg & 1024)) {
      if (G.interior) strip::commit<TS, true, false>(L, cfg.n, gW, r_bed, r_en, r_en, vn, en, upd_bits);
      else strip::commit<TS, false, false>(L, cfg.n, gW, r_bed, r_en, r_en, vn, en, upd_bits);
      loss_prev = loss_next;
      if (tid == 0) {
        double n_hi, n_lo;
        two_sum(c_hi, c_lo, n_hi, n_lo);
        carry[0] = n_hi; carry[1] = n_lo; carry[2] = loss_next;
      }
      pr0 = G.r0; pr1 = G.r0 + G.wh; pc0 = G.c0; pc1 = G.c0 + G.ww;
    } else {
      pr0 = pr1 = pc0 = pc1 = 0;
    }
    if (tid == 0) {
      const cargs_t Ke = kargs();
      const StepArgs a = load_cs(&Ke->T);
      const int64_t rout = (int64_t)chain * a.rec_stride + a.rec_offset + s;
      a.loss[rout] = loss_prev;
      a.accept[rout] = acc ? 1 : 0;
      if (a.blocks) { a.blocks[4 * rout] = s_row; a.blocks[4 * rout + 1] = s_col; a.blocks[4 * rout + 2] = s_bh; a.blocks[4 * rout + 3] = s_bw; }
    }
  }
  if (tid == 0) {                      // thread 0 wrote the sums itself: program order, no barrier needed
    fa.T.loss_sum[2 * chain] = carry[0];
    fa.T.loss_sum[2 * chain + 1] = carry[1];
  }
}

// ---- replay: the same step with host- or kernel-supplied proposals read from HBM --------------------------------------------------
template <typename TS, bool FAST_DIV>
__global__ __launch_bounds__(kST, 4) void step_strip_kernel(const StepArgs a) {
  extern __shared__ double tile[];              // [(max_bh + 2) * (max_bw + 2)] candidate bed of the window + halo ring
  __shared__ double red[16];
  __shared__ int tab[2 * 64];
  const StaticFields& S = a.S;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chain = blockIdx.x;
  const int H = S.H, W = S.W;
  const uint32_t n_cells = (uint32_t)H * (uint32_t)W;
  const size_t plane = (size_t)H * W;
  const rsrc_t r_bed = make_rsrc((const TS*)a.beds + (size_t)chain * plane, n_cells * (uint32_t)sizeof(TS));
  const rsrc_t r_en = make_rsrc((const TS*)a.energy + (size_t)chain * plane, n_cells * (uint32_t)sizeof(TS));
  const rsrc_t r_rs = make_rsrc(a.resampled + (size_t)chain * plane, n_cells * 4u);
  const rsrc_t r_st = make_rsrc(S.sA, 3u * n_cells * 16u);
  const strip::StepConsts SC{S.res, S.rcp_res, S.two_res, S.rcp_two_res};

  double s_hi = a.loss_sum[2 * chain], s_lo = a.loss_sum[2 * chain + 1];
  double loss_prev = (s_hi + s_lo) / S.two_sigma2;
  for (int i = tid; i < a.B.n_sizes; i += kST) { tab[2 * i] = a.B.bh[i]; tab[2 * i + 1] = a.B.bw[i]; }
  if (tid < 16) red[tid] = 0.0;
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
    const rsrc_t r_f = make_rsrc(a.fields + rin * a.field_stride, (uint32_t)(bh * bw) * 8u);
    const strip::Window G = strip::make_window(H, W, row, col, bh, bw);
    const strip::Cfg cfg = strip::config(G.wh, G.ww);
    const strip::Lane L = strip::lane_setup(lane, wave, cfg, G, H, W);
    uint32_t upd_bits;
    double acc_old;
    bool guard;
    {
      double vb[kNR + 2], ve[kNR];
      double2 a2[strip::kNA];
      auto field = [&](int jj, bool in) { return ld_f64<2>(r_f, in ? (uint32_t)(L.fidx + jj * bw) * 8u : kOOB); };
      if (G.interior) {
        strip::load_state<TS, true>(L, cfg.n, W, r_bed, r_en, r_st, vb, ve, a2);
        strip::phase_a<TS, true>(L, cfg.n, W, bw, r_st, field, tile, vb, ve, a2, upd_bits, acc_old, guard);
      } else {
        strip::load_state<TS, false>(L, cfg.n, W, r_bed, r_en, r_st, vb, ve, a2);
        strip::phase_a<TS, false>(L, cfg.n, W, bw, r_st, field, tile, vb, ve, a2, upd_bits, acc_old, guard);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");        // candidate-bed tile complete
    double en[kNR], vn[kNR];
    {
      double acc_new;
      if (G.interior) strip::phase_d<TS, FAST_DIV, true>(L, cfg.n, W, bw, r_st, n_cells * 16u, SC, tile, en, acc_new);
      else strip::phase_d<TS, FAST_DIV, false>(L, cfg.n, W, bw, r_st, n_cells * 16u, SC, tile, en, acc_new);
      double delta = acc_new - acc_old;
      if (guard) delta = INFINITY;
      const double w_delta = wave64_sum(delta);
      if (lane == 0) red[wave] = w_delta;
    }
    strip::read_candidate(L, cfg.n, bw, tile, vn);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    const double sd = strip::waves_sum(red, lane);
    double c_hi, c_lo, loss_next;
    const bool acc = decide<FAST_DIV>(sd, s_hi, s_lo, S.two_sigma2, S.rcp_two_sigma2, loss_prev, uu, c_hi, c_lo, loss_next);
    if (acc) {
      if (G.interior) strip::commit<TS, true, true>(L, cfg.n, W, r_bed, r_en, r_rs, vn, en, upd_bits);
      else strip::commit<TS, false, true>(L, cfg.n, W, r_bed, r_en, r_rs, vn, en, upd_bits);
      two_sum(c_hi, c_lo, s_hi, s_lo);
      loss_prev = loss_next;
    }
    if (tid == 0) {
      a.loss[rout] = loss_prev;
      a.accept[rout] = acc ? 1 : 0;
      if (a.blocks) { a.blocks[4 * rout] = row; a.blocks[4 * rout + 1] = col; a.blocks[4 * rout + 2] = bh; a.blocks[4 * rout + 3] = bw; }
    }
    // End of step: one barrier, so that no wave writes its partial of the next step into `red` while another still reads this
    // step's; where the next halo window touches the window just written, each wave first waits for its own stores.
    if (acc && has_next && (unsigned)n_si < (unsigned)a.B.n_sizes) {
      const int nbh = tab[2 * n_si], nbw = tab[2 * n_si + 1];
      const int nr0 = max(0, n_row - nbh / 2) - 1, nr1 = min(H, n_row + nbh / 2) + 1;
      const int nc0 = max(0, n_col - nbw / 2) - 1, nc1 = min(W, n_col + nbw / 2) + 1;
      if ((nr0 < G.r0 + G.wh) && (G.r0 < nr1) && (nc0 < G.c0 + G.ww) && (G.c0 < nc1)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
  }
  if (tid == 0) {
    a.loss_sum[2 * chain] = s_hi;
    a.loss_sum[2 * chain + 1] = s_lo;
  }
}

template <typename TS>
static hipError_t launch_step_strip_t(const StepArgs& a, hipStream_t st) {
  const size_t lds = (size_t)strip_tile_len(a.B) * sizeof(double);
  auto kfast = step_strip_kernel<TS, true>;
  auto kslow = step_strip_kernel<TS, false>;
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  if (a.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.n_chains), dim3(kST), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.n_chains), dim3(kST), lds, st, a);
  return hipGetLastError();
}
hipError_t launch_step_strip(const StepArgs& a, hipStream_t st) {
  if (!a.strip) return hipErrorInvalidValue;
  return a.f32_state ? launch_step_strip_t<float>(a, st) : launch_step_strip_t<double>(a, st);
}

template <typename TS, bool NOISE>
static hipError_t launch_fused_strip_t(const FusedArgs& a, hipStream_t st) {
  const size_t lds = fused_strip_lds_doubles(a) * sizeof(double);
  auto kfast = chain_strip_kernel<TS, true, NOISE>;
  auto kslow = chain_strip_kernel<TS, false, NOISE>;
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)kfast, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)kslow, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  if (a.T.S.fast_div) hipLaunchKernelGGL(kfast, dim3(a.T.n_chains), dim3(kST), lds, st, a);
  else hipLaunchKernelGGL(kslow, dim3(a.T.n_chains), dim3(kST), lds, st, a);
  return hipGetLastError();
}

// One launch: propose_scalars_kernel for all steps must have filled a.P.scalars (n_chains x a.P.n_steps records).  The
// resampled counts of the launch are added afterwards by launch_chain_fused's post-pass (the caller's job).
hipError_t launch_chain_strip(const FusedArgs& a_in, hipStream_t st) {
  if (!a_in.T.strip || a_in.P.tab_max <= 0) return hipErrorInvalidValue;
  FusedArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("GSM_PROPOSE_DBG"); dbg = v ? atoi(v) : 0; } a.P.dbg = dbg; }   // diagnostics only
  a.work_len = strip_main_len(a);
  a.fld_len = 0;
  return a.T.f32_state ? launch_fused_strip_t<float, false>(a, st) : launch_fused_strip_t<double, false>(a, st);
}

// gsm_run_noise: launch_noise_chain_scalars must have filled a.P.scalars
hipError_t launch_chain_strip_noise(const FusedArgs& a_in, hipStream_t st) {
  if (!a_in.T.strip || a_in.P.tab_max <= 0 || !a_in.noise_re || !a_in.noise_im) return hipErrorInvalidValue;
  FusedArgs a = a_in;
  a.P.dbg = 0;
  a.work_len = strip_main_len(a);
  a.fld_len = 0;
  return a.T.f32_state ? launch_fused_strip_t<float, true>(a, st) : launch_fused_strip_t<double, true>(a, st);
}

}  // namespace gsm
