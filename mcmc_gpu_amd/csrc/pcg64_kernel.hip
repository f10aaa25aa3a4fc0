// gsm_draw_pcg64: the random draws of the reference's large-scale chain made ON THE DEVICE from NumPy's own generator streams.
//
// Per Metropolis step the reference consumes two numpy.random.Generator(PCG64) objects:
//   RandField.rng   integers(0, n_sizes, size=1) (MCMC.py:755); uniform x 3 (isotropic) or 4 (scale, nugget, ranges: :199-207);
//                   normal(size=(bh, bw)) twice (:242); normal(0, sqrt(nug), size=(bh, bw)) (:251, drawn even when nug = 0)
//   chain.rng       integers(0, H, size=1), integers(0, W, size=1) until region_mask == 1 (:1254-1258); random() (:1336)
// One 64-lane workgroup per chain walks both streams step after step (pcg64_device.h): the same numbers, bit for bit, that the
// host mirror (mcmc_gpu_amd/MCMC_gpu.py: draw_chunk) gets from NumPy -- tests/test_gpu_pcg64.py -- at ~10^7 chain-steps/s instead
// of the ~3 k per host core.  The white-noise planes go to gsm_spectral_from_noise, the scalars to gsm_run_replay.
#include "gsm_internal.h"
#include "pcg64_device.h"
#include "ziggurat_tables.h"
#include <math.h>

#ifndef GSM_PCG_DRAW_WAVES
#define GSM_PCG_DRAW_WAVES 4
#endif

namespace gsm {
static_assert(kPcgJumpWords == 4 * pcg::kJump && 64 * GSM_PCG_DRAW_WAVES <= pcg::kJump, "jump table size");

// kDrawWaves wavefronts per chain: the scalar draws of a step are made by all of them redundantly (uniform code, same state
// everywhere), the planes of normals by Stream::normals_mw -- one 64-draw window per wavefront and pass, speculating that the
// windows before it are consumed whole.  One wavefront per chain (the first version: 0.15 ms per step of 1024 chains) leaves a
// SIMD with a single latency-bound wave; four per chain fill the issue slots.
constexpr int kDrawWaves = GSM_PCG_DRAW_WAVES;
__global__ __launch_bounds__(64 * kDrawWaves) void pcg64_draw_kernel(const PcgDrawArgs a) {
  constexpr int NW = kDrawWaves;
  __shared__ uint64_t jump[4 * 64 * NW];
  __shared__ uint64_t zig[kZigTabWords];
  __shared__ uint64_t xch[NW + 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), chain = blockIdx.x;
  for (int i = tid; i < 4 * 64 * NW; i += 64 * NW) jump[i] = a.jump[i];
  for (int i = tid; i < kZigTabWords; i += 64 * NW) zig[i] = a.zig[i];
  __syncthreads();
  pcg::Stream R, C;
  uint64_t* rs = a.rf_state + 6 * (size_t)chain;
  uint64_t* cs = a.ch_state + 6 * (size_t)chain;
  R.s = pcg::u128{rs[0], rs[1]}; R.inc = pcg::u128{rs[2], rs[3]}; R.has32 = (uint32_t)rs[4]; R.cached = (uint32_t)rs[5];
  C.s = pcg::u128{cs[0], cs[1]}; C.inc = pcg::u128{cs[2], cs[3]}; C.has32 = (uint32_t)cs[4]; C.cached = (uint32_t)cs[5];
  R.jump = C.jump = jump; R.zig = C.zig = zig;
  // this lane's jump constants of the RandField stream: 64 wave + lane + 1 draws ahead
  const int ahead = 64 * wave + lane;
  const pcg::u128 A_l{jump[4 * ahead], jump[4 * ahead + 1]};
  const pcg::u128 C_l = pcg::mul128(pcg::u128{jump[4 * ahead + 2], jump[4 * ahead + 3]}, R.inc);
  const gsm_rf_params& P = a.rf;
  for (int s = 0; s < a.n_steps; ++s) {
    const int64_t rec = (int64_t)chain * a.n_steps + s;
    const int si = (int)R.bounded((uint32_t)a.n_sizes);
    const int bh = a.bh[si], bw = a.bw[si], B = bh * bw;
    const double scale = R.uniform(P.scale_min, P.scale_max) / 3.0;
    const double nug = R.uniform(0.0, P.nugget_max);
    double rx, ry;
    if (!P.isotropic) { rx = R.uniform(P.range_min_x, P.range_max_x); ry = R.uniform(P.range_min_y, P.range_max_y); }
    else { rx = R.uniform(P.range_min_x, P.range_max_x); ry = rx; }
    // the three planes of a step (MCMC.py:242 twice, :251) through ONE copy of the window loop
#pragma nounroll
    for (int pl = 0; pl < 3; ++pl) {
      double* dst = (pl == 0) ? a.noise_re + rec * a.field_stride : (pl == 1) ? a.noise_im + rec * a.field_stride
                                                                              : (a.nugget ? a.nugget + rec * a.field_stride : nullptr);
      if (NW == 1) R.normals(B, 0.0, (pl == 2) ? sqrt(nug) : 1.0, dst, lane, A_l, C_l);
      else R.template normals_mw<NW>(B, 0.0, (pl == 2) ? sqrt(nug) : 1.0, dst, lane, wave, A_l, C_l, xch);
    }
    int ix = 0, iy = 0;
    for (int tries = 0;; ++tries) {
      ix = (int)C.bounded((uint32_t)a.H);
      iy = (int)C.bounded((uint32_t)a.W);
      if (!a.region_mask || a.region_mask[ix * a.W + iy] == 1) break;
      if (tries > (1 << 20)) { if (tid == 0) atomicOr(a.err, 128); break; }
    }
    const double uu = C.next_double();
    if (tid == 0) {
      a.size_idx[rec] = si;
      a.centre[2 * rec] = ix; a.centre[2 * rec + 1] = iy;
      a.u[rec] = uu;
      a.rf_scalars[4 * rec] = scale; a.rf_scalars[4 * rec + 1] = nug; a.rf_scalars[4 * rec + 2] = rx; a.rf_scalars[4 * rec + 3] = ry;
    }
  }
  if (tid == 0) {
    rs[0] = R.s.lo; rs[1] = R.s.hi; rs[4] = R.has32; rs[5] = R.cached;
    cs[0] = C.s.lo; cs[1] = C.s.hi; cs[4] = C.has32; cs[5] = C.cached;
  }
}

hipError_t launch_pcg64_draw(const PcgDrawArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(pcg64_draw_kernel, dim3(a.n_chains), dim3(64 * kDrawWaves), 0, st, a);
  return hipGetLastError();
}

// gsm_sgs_draw_pcg64: the draws of the small-scale chain from the chain's own NumPy generator, on the device (one wavefront per
// chain, iterations one after the other: the stream is sequential).  Per iteration, in the reference's order (MCMC.py:1750-1757
// block centre with rejection on region_mask and block sizes; :128 rng.shuffle of the block's cells; :165 one rng.normal per cell
// without conditioning data, in visiting order; :1797 rng.random()).  Same outputs as sgs_draw_kernel (Philox mode).
__global__ __launch_bounds__(64) void sgs_draw_pcg64_kernel(const SgsDrawArgs a, uint64_t* states, const uint64_t* jump_g, const uint64_t* zig_g) {
  __shared__ uint64_t jump[4 * pcg::kJumpWave];
  __shared__ uint64_t zig[kZigTabWords];
  __shared__ int32_t order[1024];                // packed (row << 16 | col) of the block's cells, shuffled in place
  __shared__ double zt[1024];                    // the iteration's normals in drawing order
  const int lane = threadIdx.x, chain = blockIdx.x;
  for (int i = lane; i < 4 * pcg::kJumpWave; i += 64) jump[i] = jump_g[i];
  for (int i = lane; i < kZigTabWords; i += 64) zig[i] = zig_g[i];
  __syncthreads();
  pcg::Stream G;
  uint64_t* gs = states + 6 * (size_t)chain;
  G.s = pcg::u128{gs[0], gs[1]}; G.inc = pcg::u128{gs[2], gs[3]}; G.has32 = (uint32_t)gs[4]; G.cached = (uint32_t)gs[5];
  G.jump = jump; G.zig = zig;
  const pcg::u128 A_l{jump[4 * lane], jump[4 * lane + 1]};
  const pcg::u128 C_l = pcg::mul128(pcg::u128{jump[4 * lane + 2], jump[4 * lane + 3]}, G.inc);
  for (int j = 0; j < a.n_iters; ++j) {
    const int64_t rec = (int64_t)j * a.n_chains + chain;
    int row = 0, col = 0;
    for (int tries = 0;; ++tries) {
      row = G.integers(0, a.H); col = G.integers(0, a.W);
      if (!a.region_mask || a.region_mask[row * a.W + col] == 1) break;
      if (tries > (1 << 20)) { if (lane == 0) atomicOr(a.err, 16); break; }
    }
    const int bsx = G.integers(a.min_x, a.max_x), bsy = G.integers(a.min_y, a.max_y);
    // int(ix -/+ bs / 2) of MCMC.py:1758-1761 (truncation towards zero of a half-integer)
    const int r0 = max(0, (2 * row - bsx) / 2), r1 = min(a.H, (2 * row + bsx) / 2);
    const int c0 = max(0, (2 * col - bsy) / 2), c1 = min(a.W, (2 * col + bsy) / 2);
    const int ww = c1 - c0, n = max(0, r1 - r0) * max(0, ww);
    if (lane == 0) {
      a.win[4 * rec] = r0; a.win[4 * rec + 1] = r1; a.win[4 * rec + 2] = c0; a.win[4 * rec + 3] = c1;
      a.blk[4 * rec] = row; a.blk[4 * rec + 1] = col; a.blk[4 * rec + 2] = bsx; a.blk[4 * rec + 3] = bsy;
      a.cell_off[rec] = (int32_t)(rec * a.max_cells); a.cell_cnt[rec] = (n <= a.max_cells && n <= 1024) ? n : 0;
    }
    if ((n > a.max_cells || n > 1024) && lane == 0) atomicOr(a.err, 1);
    const int nn = (n > a.max_cells || n > 1024) ? 0 : n;
    for (int p = lane; p < nn; p += 64) order[p] = ((r0 + p / ww) << 16) | (c0 + p % ww);
    __syncthreads();
    // rng.shuffle(inds): Fisher-Yates from the back, index by masked rejection on 32-bit words
    for (int i = nn - 1; i >= 1; --i) {
      const int jx = (int)G.interval((uint32_t)i);
      if (lane == 0 && jx != i) { const int32_t t = order[jx]; order[jx] = order[i]; order[i] = t; }
    }
    __syncthreads();
    // cells in visiting order; one standard normal per cell without conditioning data, in that order
    int count = 0;
    for (int p0 = 0; p0 < nn; p0 += 64) {
      const int p = p0 + lane;
      bool free_cell = false;
      if (p < nn) {
        const int i = order[p] >> 16, jj = order[p] & 0xFFFF;
        a.cells[2 * (rec * a.max_cells + p)] = i; a.cells[2 * (rec * a.max_cells + p) + 1] = jj;
        free_cell = a.is_data[i * a.W + jj] == 0;
      }
      count += __popcll(__ballot(free_cell));
    }
    G.normals(count, 0.0, 1.0, zt, lane, A_l, C_l);
    __syncthreads();
    int seen = 0;
    for (int p0 = 0; p0 < nn; p0 += 64) {
      const int p = p0 + lane;
      bool free_cell = false;
      if (p < nn) free_cell = a.is_data[(order[p] >> 16) * a.W + (order[p] & 0xFFFF)] == 0;
      const unsigned long long m = __ballot(free_cell);
      const int my = seen + __popcll(m & ((lane == 0) ? 0ull : (~0ull >> (64 - lane))));
      if (p < nn) a.z[rec * a.max_cells + p] = free_cell ? zt[my] : 0.0;
      seen += __popcll(m);
    }
    const double uu = G.next_double();
    if (lane == 0) a.u[rec] = uu;
    __syncthreads();
  }
  if (lane == 0) { gs[0] = G.s.lo; gs[1] = G.s.hi; gs[4] = G.has32; gs[5] = G.cached; }
}

hipError_t launch_sgs_draw_pcg64(const SgsDrawArgs& a, uint64_t* states, const uint64_t* jump, const uint64_t* zig, hipStream_t st) {
  hipLaunchKernelGGL(sgs_draw_pcg64_kernel, dim3(a.n_chains), dim3(64), 0, st, a, states, jump, zig);
  return hipGetLastError();
}

// the two constant tables (host side): jump table of the LCG and the ziggurat tables
void pcg64_host_tables(uint64_t* jump_out, const uint64_t** zig_out) {
  pcg::build_jump_table(jump_out);
  *zig_out = kZigTab;
}

}  // namespace gsm
