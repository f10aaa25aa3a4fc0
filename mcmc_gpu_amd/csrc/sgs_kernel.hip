// Small-scale chain on gfx950: sequential Gaussian simulation of one block per chain, and the chain's full-grid loss.
//
// Replaces, for a batch of chains:
//   sgs                      gstatsMCMC/MCMC.py:91-173   (cell loop :135-168, radius widening :150-156)
//   neighbors (octant search) gstatsMCMC/gstatsim_custom/neighbors.py:4-64
//   ok_solve                 gstatsMCMC/gstatsim_custom/_krige.py:5-44
//   chain.loss on the proposed bed + thickness guard  MCMC.py:1781-1795 (loss :1021-1044, Topography.py:592-600)
//
// The random numbers are the caller's (replay: NumPy's PCG64 on the host, in the reference's order; Philox mode:
// sgs_draw_kernel): the order in which the block's cells are visited (rng.shuffle, MCMC.py:128) and one standard normal per
// simulated cell (rng.normal(est, sqrt(var)) = est + sqrt(var) * z, MCMC.py:165).
//
// What depends on what.  A cell's NEIGHBOUR SET and KRIGING WEIGHTS depend only on where values exist when the cell is
// visited -- everything outside the block, the block's conditioning data, and the block cells visited before it -- not on the
// values themselves.  So the simulation of a block is split in two:
//   sgs_weights_kernel   one wavefront per (chain, cell), all cells of all chains side by side: octant search, ordinary-kriging
//                        system, weights and kriging variance -> a record per cell (neighbour list with the values that are
//                        known already, weights, standard deviation);
//   sgs_sequence_kernel  one wavefront per chain walks the cells in visiting order: est = mean + sum w (v - mean),
//                        value = est + sd * z, the only sequential part (a 48-term dot product per cell).
//
// Octant search (exact for any search radius; the reference's driver uses 48 neighbours within 30 km at 500 m = 60 cells):
// every cell within +-hw of the cell that holds a value and lies closer than `radius` belongs to one of eight 45-degree
// sectors, (b pi/4, (b+1) pi/4], b = -4..3, of atan2(y0 - y, x0 - x), decided from the signs and magnitudes of the two
// coordinate differences -- exactly what numpy.arctan2 gives on the sector boundaries (multiples of pi/4 are hit only by
// cells on the axes and diagonals, where arctan2 returns the boundary value bit for bit; checked on the host).  Per sector
// the num_points / 8 nearest are kept, equidistant ones in ascending (row, column) -- numpy.argsort(kind='stable') of the
// reference's masked C-order array; the reference's default argsort is not stable, so its choice among equidistant
// candidates is implementation-defined.  The window is scanned in square rings of growing Chebyshev
// radius R.  After ring R every unscanned cell is at least (R + 1) min(|dx|, |dy|) away, so a sector is complete as soon as it
// holds num_points / 8 candidates closer than that; a sector whose remaining cells all lie outside the grid / window is
// exhausted.  The scan stops when every sector is complete or exhausted: with values on most cells that is after 4-6 rings,
// whatever the radius.  A cell without any value inside `radius` widens the search as the reference does (radius += 100 km,
// window = ceil(radius / |dx|) cells, MCMC.py:150-156) until the window covers the grid.
//
// Ordinary kriging: (n+1) x (n+1) system [Sigma 1; 1^T 0] w = [rho; 1], covariances from a table indexed by the integer lag
// between two cells (the host evaluates the reference's covariance model -- scipy's Bessel K for Matern -- once per lag),
// solved by Gauss-Jordan elimination on the diagonal (the covariance block is positive definite, the last pivot is
// -1' Sigma^-1 1) in fp64 with one row of the augmented matrix per lane, held in registers; the pivot row of a step reaches
// the other lanes through v_readlane.  The reference calls numpy.linalg.lstsq (SVD): same solution, other rounding -- the
// stated tolerance of this path.  A pivot below eps * N * max|diag| raises the "singular" flag (lstsq would truncate there).
// Limits: num_points <= 48, block <= 1024 cells.
#include "gsm_internal.h"
#include "device_util.h"
#include "residual_device.h"
#include "normal_score.h"
#include "proposal_device.h"
#include <math.h>
#ifndef GSM_SGS_LISTCAP
#define GSM_SGS_LISTCAP 80
#endif

namespace gsm {

constexpr int kSgsMaxPts = 48;
constexpr int kSgsMaxWin = 1024;
// candidates kept per sector between prunings (a scan pass appends at most 64; a list is pruned to its k8 nearest when it holds more than
// kSgsListCap - 64 = 16).  The search structure's LDS bounds the kernel's occupancy: 128 -> 80 entries, 64 rings of 16-bit certification
// counters instead of 128 of 32 bits took it from 17 KiB to 9.7 KiB per workgroup = 9 -> 16 wavefronts per CU (the register budget's
// four per SIMD): +14 % per iteration at 256 chains, the extra prunings included (same-box A/B of 128 / 112 / 96 / 88 / 80 entries)
constexpr int kSgsListCap = GSM_SGS_LISTCAP;
constexpr int kSgsCertMax = 64;           // rings with certification counters (the driver's 30 km at 500 m = 60 rings); beyond, a sector
                                          // completes by exhaustion only
constexpr uint64_t kSgsPendingTag = 0x7FF8C0DE00000000ull;   // neighbour record: NaN-boxed (visiting slot << 16 | block-local index) of a cell visited earlier
// with SgsArgs::defer the values of the other neighbours are not in the record either (the weights do not depend on them: the
// records of an iteration can then be made while the iteration before is still running):
constexpr uint64_t kSgsWindowTag = 0x7FF8C0DF00000000ull;    // | block-local index: conditioning data inside the block (sgs_sequence_kernel's overlay holds it)
constexpr uint64_t kSgsGridTag = 0x7FF8C0E000000000ull;      // | flat grid index: a cell outside the block, read from the grid by sgs_sequence_kernel

__device__ __forceinline__ double wave_sum_f64(double v) { return dev::wave64_sum(v); }     // DPP tree, wave-uniform result
// two sums over the 64 lanes at once, results wave-uniform.  The halves are folded first -- v_permlane32_swap (gfx950) exchanges the
// upper 32 lanes of `a` with the lower 32 of `b`, so a' + b' carries a[l] + a[l + 32] in lanes 0..31 and b[l] + b[l + 32] in lanes
// 32..63 -- and ONE chain of row reductions then sums both: 25 instructions instead of the 52 of two full wave reductions (the
// sequence kernel is one wavefront per chain: every instruction of the cell loop is 5-8 cycles of its critical path).
__device__ __forceinline__ void wave_sum2_f64(double& a, double& b) {
  typedef unsigned v2u32 __attribute__((ext_vector_type(2)));
  const dev::v2i32 ba = __builtin_bit_cast(dev::v2i32, a), bb = __builtin_bit_cast(dev::v2i32, b);
  const v2u32 lo = __builtin_amdgcn_permlane32_swap((unsigned)ba.x, (unsigned)bb.x, false, false);
  const v2u32 hi = __builtin_amdgcn_permlane32_swap((unsigned)ba.y, (unsigned)bb.y, false, false);
  dev::v2i32 na, nb;
  na.x = (int)lo.x; na.y = (int)hi.x; nb.x = (int)lo.y; nb.y = (int)hi.y;
  double z = __builtin_bit_cast(double, na) + __builtin_bit_cast(double, nb);
  z = dev::row16_sum(z);
  z += dev::dpp_f64<0x142, 0xA>(z);                             // row_bcast:15: rows 1 and 3 now hold the totals of their halves
  const dev::v2i32 bz = __builtin_bit_cast(dev::v2i32, z);
  dev::v2i32 oa, ob;
  oa.x = __builtin_amdgcn_readlane(bz.x, 31); oa.y = __builtin_amdgcn_readlane(bz.y, 31);
  ob.x = __builtin_amdgcn_readlane(bz.x, 63); ob.y = __builtin_amdgcn_readlane(bz.y, 63);
  a = __builtin_bit_cast(double, oa); b = __builtin_bit_cast(double, ob);
}

// sector b + 4 in 0..7 of atan2(dy, dx) in (b pi/4, (b+1) pi/4]
__device__ __forceinline__ int octant(double dy, double dx) {
  // the same case analysis as selects (no divergent branches in the search pass):
  //   dy == 0: angle pi -> b = 3 (sector 7), angle 0 -> b = -1 (sector 3)
  //   dy > 0:  dx > 0: (0, pi/4] 4 | (pi/4, pi/2) 5;   dx == 0: 5;   dx < 0: (pi/2, 3pi/4] 6 | (3pi/4, pi) 7
  //   dy < 0:  dx > 0: (-pi/4, 0) 3 | (-pi/2, -pi/4] 2;  dx == 0: 1 (-pi/2 -> (-3pi/4, -pi/2]);  dx < 0: (-3pi/4, -pi/2) 1 | (-pi, -3pi/4] 0
  const double ay = fabs(dy), ax = fabs(dx);
  const bool right = dx > 0.0, xz = dx == 0.0;
  const int s_up = right ? ((ay <= ax) ? 4 : 5) : (xz ? 5 : ((ay >= ax) ? 6 : 7));
  const int s_dn = right ? ((ay < ax) ? 3 : 2) : (xz ? 1 : ((ay > ax) ? 1 : 0));
  return (dy == 0.0) ? ((dx < 0.0) ? 7 : 3) : ((dy > 0.0) ? s_up : s_dn);
}

// ---------------------------------------------------------------------------------------------------------------------
// sgs_rank_kernel: per chain, the visiting rank of every block cell (-1: the cell holds conditioning data and is never
// simulated), so that the search of cell k knows which block cells hold a value when k is visited.  One workgroup per chain.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgs_rank_kernel(const SgsArgs a) {
  const int chain = blockIdx.x, tid = threadIdx.x;
  const int H = a.H, W = a.W;
  const int r0 = a.win[4 * chain], r1 = a.win[4 * chain + 1], c0 = a.win[4 * chain + 2], c1 = a.win[4 * chain + 3];
  const int wh = r1 - r0, ww = c1 - c0;
  int32_t* rank = a.rank + (size_t)chain * kSgsMaxWin;
  const int k_lo = a.cell_off[chain], cnt = a.cell_cnt ? a.cell_cnt[chain] : a.cell_off[chain + 1] - k_lo;
  if (r0 < 0 || c0 < 0 || r1 > H || c1 > W || wh < 0 || ww < 0 || wh * ww > kSgsMaxWin || cnt > a.max_cells || cnt < 0) {
    if (tid == 0) { atomicOr(a.err, 1); a.rank_ok[chain] = 0; }
    return;
  }
  if (tid == 0) a.rank_ok[chain] = 1;
  for (int p = tid; p < wh * ww; p += 256) rank[p] = 0x7fffffff;          // a window cell that is not listed never holds a value
  __syncthreads();
  const double* g = a.grid + (size_t)chain * H * W;
  for (int k = tid; k < cnt; k += 256) {
    const int i = a.cells[2 * (k_lo + k)], j = a.cells[2 * (k_lo + k) + 1];
    if (i < r0 || i >= r1 || j < c0 || j >= c1) { atomicOr(a.err, 2); continue; }
    const double v = a.zcond ? a.zcond[i * W + j] : g[i * W + j];
    rank[(i - r0) * ww + (j - c0)] = isnan(v) ? k : -1;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// sgs_weights_kernel: one 64-lane workgroup (one wavefront) per (cell slot, chain)
// ---------------------------------------------------------------------------------------------------------------------
struct SgsSearchLds {
  double list_d[8][kSgsListCap];
  int32_t list_g[8][kSgsListCap];
  int32_t len[8];
  int32_t cum[8];
  uint32_t cert[8][kSgsCertMax / 2];      // candidates of a sector by certification ring: 16-bit counters, two per word (a ring holds
                                          // at most 8 x 127 cells).  The kernel's occupancy is bounded by this structure's size.
  int32_t nb_g[kSgsMaxPts];
  int32_t nb_rc[kSgsMaxPts];            // (row << 16) | col
  double tmp_d[kSgsMaxPts];
  int32_t tmp_g[kSgsMaxPts];
};

// keeps the `keep` smallest (distance, cell) of sector s, in ascending order, at the head of its list; all lanes call it
__device__ __forceinline__ void sgs_prune_sector(SgsSearchLds& L, int s, int keep, int lane) {
  const int len = L.len[s];
  __syncthreads();
  for (int e = lane; e < len; e += 64) {
    const double d = L.list_d[s][e];
    const int g = L.list_g[s][e];
    int r = 0;
    for (int q = 0; q < len; ++q) {
      const double dq = L.list_d[s][q];
      const int gq = L.list_g[s][q];
      r += (dq < d || (dq == d && gq < g)) ? 1 : 0;
    }
    if (r < keep) { L.tmp_d[r] = d; L.tmp_g[r] = g; }
  }
  __syncthreads();
  const int n = min(len, keep);
  if (lane < n) { L.list_d[s][lane] = L.tmp_d[lane]; L.list_g[s][lane] = L.tmp_g[lane]; }
  if (lane == 0) L.len[s] = n;
  __syncthreads();
}

// cell `p` (0 <= p < R) of arc `g` (0..7) of the square ring of Chebyshev radius R around a cell, as (row, column) offsets: the ring's
// 8 R cells in the order top row left to right, right column downwards, bottom row right to left, left column upwards; each side
// is two arcs of R cells, the first starting at a corner (p = 0), the second at the cell on the axis
__device__ __forceinline__ void ring_cell(int R, int g, int p, int& di, int& dj) {
  const int side = g >> 1, e = ((g & 1) ? 0 : -R) + p;
  di = (side == 0) ? -R : (side == 2) ? R : (side == 1) ? e : -e;
  dj = (side == 1) ? R : (side == 3) ? -R : (side == 0) ? e : -e;
}

template <int K>
struct GjStep {
  // one Gauss-Jordan step on pivot K: r[] is this lane's row of [A | b] (columns 0..47 neighbours, 48 the Lagrange column,
  // 49 the right-hand side)
  static __device__ __forceinline__ void run(double (&r)[50], int lane, int n, bool lagr, double tol, double tol_l, double& mypiv, bool& singular) {
    if (K < n || (K == 48 && lagr)) {                            // wave-uniform: rows n..47 do not exist; no Lagrange row in simple kriging
      const dev::v2i32 pb = __builtin_bit_cast(dev::v2i32, r[K]);
      dev::v2i32 ps;
      ps.x = __builtin_amdgcn_readlane(pb.x, K);
      ps.y = __builtin_amdgcn_readlane(pb.y, K);
      const double pv = __builtin_bit_cast(double, ps);
      if (!(fabs(pv) > (K == 48 ? tol_l : tol))) singular = true;
      // 1 / pivot: hardware estimate + two Newton steps (every lane computes the same value), then the fma-corrected quotient
      double rp = __builtin_amdgcn_rcp(pv);
      rp = __fma_rn(__fma_rn(-pv, rp, 1.0), rp, rp);
      rp = __fma_rn(__fma_rn(-pv, rp, 1.0), rp, rp);
      double f = dev::exact_div(r[K], pv, rp);
      if (lane == K) { f = 0.0; mypiv = pv; }
#pragma unroll
      for (int j = K + 1; j < 50; ++j) {
        const dev::v2i32 b = __builtin_bit_cast(dev::v2i32, r[j]);
        dev::v2i32 o;
        o.x = __builtin_amdgcn_readlane(b.x, K);
        o.y = __builtin_amdgcn_readlane(b.y, K);
        r[j] = __fma_rn(-f, __builtin_bit_cast(double, o), r[j]);
      }
    }
    GjStep<K + 1>::run(r, lane, n, lagr, tol, tol_l, mypiv, singular);
  }
};
template <>
struct GjStep<49> {
  static __device__ __forceinline__ void run(double (&)[50], int, int, bool, double, double, double&, bool&) {}
};

__global__ __launch_bounds__(64) void sgs_weights_kernel(const SgsArgs a) {
  __shared__ SgsSearchLds L;
  const int slot = blockIdx.x, chain = blockIdx.y, lane = threadIdx.x;
  if (!a.rank_ok[chain]) return;
  const int k_lo = a.cell_off[chain], cnt = a.cell_cnt ? a.cell_cnt[chain] : a.cell_off[chain + 1] - k_lo;
  if (slot >= cnt) return;
  const int H = a.H, W = a.W;
  const size_t rec = (size_t)chain * a.max_cells + slot;
  const int i0 = a.cells[2 * (k_lo + slot)], j0 = a.cells[2 * (k_lo + slot) + 1];
  const int r0 = a.win[4 * chain], r1 = a.win[4 * chain + 1], c0 = a.win[4 * chain + 2], c1 = a.win[4 * chain + 3];
  const int ww = c1 - c0;
  if (i0 < r0 || i0 >= r1 || j0 < c0 || j0 >= c1) { if (lane == 0) a.rec_hdr[rec].n = -2; return; }     // flagged by sgs_rank_kernel
  const int32_t* rank = a.rank + (size_t)chain * kSgsMaxWin;
  if (rank[(i0 - r0) * ww + (j0 - c0)] != slot) { if (lane == 0) { a.rec_hdr[rec].n = -1; a.rec_hdr[rec].op = (i0 - r0) * ww + (j0 - c0); } return; }       // conditioned already (MCMC.py:141)
  const double* __restrict__ g = a.grid + (size_t)chain * H * W;
  const int k8 = a.num_points / 8;
  const double x0 = a.xs[j0], y0 = a.ys[i0];
  const double sx = a.xs[1] - a.xs[0], sy = a.ys[1] - a.ys[0];
  const double adx = fabs(sx), ady = fabs(sy), dmin = fmin(adx, ady);
  const double inv_cert = 1.0 / (dmin * (1.0 - 1e-6));          // certification ring of a distance: floor(d * inv_cert)
  const double fac_x = fmin(1.0, ady / adx), fac_y = fmin(1.0, adx / ady);
  double radius = a.radius;
  int hw = a.hw;
  int n = 0;
  for (;;) {                                                     // radius widening (MCMC.py:150-156): usually one trip
    const int ilo = max(0, i0 - hw), ihi = min(H - 1, i0 + hw), jlo = max(0, j0 - hw), jhi = min(W - 1, j0 + hw);
    // cells towards smaller / larger row and column that the window holds
    const int e_up = i0 - ilo, e_dn = ihi - i0, e_lf = j0 - jlo, e_rt = jhi - j0;
    const int r_max = max(max(e_up, e_dn), max(e_lf, e_rt));
    // sector s: extent (in cells) along its primary axis on its side.  dy = y0 - y > 0 <=> rows with smaller y.
    const int e_ypos = (sy > 0.0) ? e_up : e_dn, e_yneg = (sy > 0.0) ? e_dn : e_up;
    const int e_xpos = (sx > 0.0) ? e_lf : e_rt, e_xneg = (sx > 0.0) ? e_rt : e_lf;
    for (int q = lane; q < 8 * kSgsCertMax / 2; q += 64) (&L.cert[0][0])[q] = 0u;
    if (lane < 8) { L.len[lane] = 0; L.cum[lane] = 0; }
    __syncthreads();
    int my_ext = 0;
    double my_fac = 1.0;
    if (lane < 8) {
      const bool xprim = (lane == 3 || lane == 4 || lane == 7 || lane == 0);
      my_fac = xprim ? fac_x : fac_y;
      my_ext = (lane == 3 || lane == 4) ? e_xpos : (lane == 7 || lane == 0) ? e_xneg : (lane == 5 || lane == 6) ? e_ypos : e_yneg;
    }
    unsigned done_mask = 0;                                      // wave-uniform: sectors complete or exhausted
    int R = 0;
    bool long_list = false;
    // one candidate cell per lane: everything is computed for every lane at clamped (always valid) indices; ONE predicate guards the insertion
    auto probe = [&](int di, int dj, bool ok) {
      const int i = i0 + di, j = j0 + dj;
      ok = ok && i >= ilo && i <= ihi && j >= jlo && j <= jhi;
      const int ic = min(max(i, ilo), ihi), jc = min(max(j, jlo), jhi);
      const bool inwin = ic >= r0 && ic < r1 && jc >= c0 && jc < c1;
      const int rk = rank[inwin ? (ic - r0) * ww + (jc - c0) : 0];
      const double gv = a.defer ? 0.0 : g[ic * W + jc];          // defer: every cell outside the block holds a value (the caller's promise)
      const double ddx = x0 - a.xs[jc], ddy = y0 - a.ys[ic];
      const double d = sqrt(ddx * ddx + ddy * ddy);
      const int s = octant(ddy, ddx);
      // rk: -1 = conditioning data, < slot = simulated before this cell
      const bool ins = ok && (inwin ? rk < slot : !isnan(gv)) && d < radius && !((done_mask >> s) & 1u);
      if (ins) {
        const int pos = atomicAdd(&L.len[s], 1);
        long_list |= pos + 1 > kSgsListCap - 64;
        L.list_d[s][pos] = d; L.list_g[s][pos] = i * W + j;
        const double qf = d * inv_cert;
        if (qf < (double)kSgsCertMax) { const int qi = (int)qf; atomicAdd(&L.cert[s][qi >> 1], 1u << (16 * (qi & 1))); }
      }
      __syncthreads();
      // a list that could not take another full pass is cut back to the k8 nearest (nothing beyond them can be selected); the lane
      // whose insertion took a list over that mark knows: the eight lengths are only looked at then
      if (__ballot(long_list)) {
        for (int s = 0; s < 8; ++s)
          if (L.len[s] > kSgsListCap - 64) sgs_prune_sector(L, s, k8, lane);
        long_list = false;
      }
    };
    while (R < r_max && done_mask != 0xFFu) {
      // one pass = rings R+1 .. R_hi: the 7 x 7 window first (rings 1-3 = 48 cells), then ring by ring
      const int R_lo = R + 1;
      int R_hi;
      if (R == 0) {
        R_hi = min(3, r_max);
        const int side_w = 2 * R_hi + 1, cells_in_pass = side_w * side_w;
        for (int t0 = 0; t0 < cells_in_pass; t0 += 64) {
          const int t = t0 + lane;
          const int di = t / side_w - R_hi, dj = t % side_w - R_hi;
          probe(di, dj, t < cells_in_pass && !(di == 0 && dj == 0));
        }
      } else {
        R_hi = R_lo;
        const int cells_in_pass = 8 * R_hi;
        const float inv_side = 1.0f / (float)(2 * R_hi);
        for (int t0 = 0; t0 < cells_in_pass; t0 += 64) {
          const int t = t0 + lane;
          // t / (2 R_hi) without an integer division: (t + 1/2) / (2 R_hi) is at least 1 / (4 R_hi) away from an integer, fp32 is exact enough
          const int side = (int)(((float)t + 0.5f) * inv_side), o = t - side * 2 * R_hi;
          int di, dj;
          ring_cell(R_hi, 2 * side + (o >= R_hi ? 1 : 0), o >= R_hi ? o - R_hi : o, di, dj);
          probe(di, dj, t < cells_in_pass);
        }
      }
      R = R_hi;
      bool fin = false;
      if (lane < 8) {
        int c = L.cum[lane];
        for (int q = R_lo; q <= R && q < kSgsCertMax; ++q) c += (int)((L.cert[lane][q >> 1] >> (16 * (q & 1))) & 0xFFFFu);
        L.cum[lane] = c;
        fin = c >= k8 || (double)my_ext <= floor((double)R * my_fac + 1e-6);
      }
      done_mask |= (unsigned)(__ballot(fin) & 0xFFull);
      __syncthreads();
    }
    // selection: per sector the k8 nearest in ascending (distance, cell); sectors concatenated in angle order
    // all eight sectors at once, eight lanes each (a sector's list rarely holds more than a few dozen candidates: one sector after
    // the other left most lanes idle): lane = 8 * sector + e mod 8 ranks its candidates against the whole list of its sector
    {
      const int my_s = lane >> 3;
      int tot = 0, my_base = 0, my_len = 0;
      for (int s = 0; s < 8; ++s) {
        const int len = L.len[s];
        if (s == my_s) { my_base = tot; my_len = len; }
        tot += min(len, k8);
      }
      for (int e = lane & 7; e < my_len; e += 8) {
        const double d = L.list_d[my_s][e];
        const int gg = L.list_g[my_s][e];
        int r = 0;
        for (int q = 0; q < my_len; ++q) {
          const double dq = L.list_d[my_s][q];
          const int gq = L.list_g[my_s][q];
          r += (dq < d || (dq == d && gq < gg)) ? 1 : 0;
        }
        if (r < k8) L.nb_g[my_base + r] = gg;
      }
      n = tot;
    }
    __syncthreads();
    if (n > 0) break;
    // nothing within the radius: the reference adds 100 km and rebuilds the stencil (window = ceil(radius / |dx|) cells)
    if (ilo == 0 && jlo == 0 && ihi == H - 1 && jhi == W - 1 &&
        radius * radius > ((double)(W - 1) * adx) * ((double)(W - 1) * adx) + ((double)(H - 1) * ady) * ((double)(H - 1) * ady)) break;
    radius += 100e3;
    hw = (int)fmin(ceil(radius / adx), 1.0e6);
  }
  if (n == 0) {                                                  // no value anywhere on the grid: the reference would loop forever
    if (lane == 0) { atomicOr(a.err, 4); a.rec_hdr[rec].n = 0; a.rec_hdr[rec].op = (i0 - r0) * ww + (j0 - c0); }
    return;
  }
  if (lane < n) { const int gg = L.nb_g[lane]; const int rr = gg / W; L.nb_rc[lane] = (rr << 16) | (gg - rr * W); }
  __syncthreads();
  // ---- kriging system, one row per lane in registers: ordinary kriging [Sigma 1; 1' 0] w = [rho; 1] (_krige.py:25-37) or, with
  // a.ktype == 1, simple kriging Sigma w = rho (_krige.py:66-73: no Lagrange row / column) ------------------------------------
  const bool lagr = a.ktype == 0;
  const int mi = a.mi, mj = a.mj, lag_w = 2 * mj + 1;
  const double* __restrict__ lag = a.lag;
  double r[50];
  const int my_rc = (lane < n) ? L.nb_rc[lane] : 0;
  const int my_i = my_rc >> 16, my_j = my_rc & 0xFFFF;
  bool lag_ok = true;
  if (mi >= H - 1 && mj >= W - 1) {
    // the table spans every lag of the grid (lag_extents: grids up to 2048 x 2048 lags): no range test, and the index of the pair
    // (this lane's neighbour, neighbour j) is one subtraction -- (my_i + mi) lag_w + my_j + mj minus neighbour j's i lag_w + j
    const int my_base = (my_i + mi) * lag_w + my_j + mj;
    __syncthreads();
    if (lane < n) L.nb_rc[lane] = my_i * lag_w + my_j;           // the (row, col) pairs are in registers by now
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 48; ++j) {
      double v = 0.0;
      if (j < n) {                                               // wave-uniform
        if (lane < n) v = lag[my_base - L.nb_rc[j]];
        else if (lane == 48 && lagr) v = 1.0;
      }
      r[j] = v;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 48; ++j) {
      double v = 0.0;
      if (j < n) {                                               // wave-uniform
        if (lane < n) {
          const int rc = L.nb_rc[j];
          const int di = my_i - (rc >> 16), dj = my_j - (rc & 0xFFFF);
          if (abs(di) > mi || abs(dj) > mj) lag_ok = false; else v = lag[(di + mi) * lag_w + dj + mj];
        } else if (lane == 48 && lagr) v = 1.0;
      }
      r[j] = v;
    }
  }
  {
    double v48 = 0.0, v49 = 0.0;
    if (lane < n) {
      const int di = my_i - i0, dj = my_j - j0;
      v48 = lagr ? 1.0 : 0.0;
      if (abs(di) > mi || abs(dj) > mj) lag_ok = false; else v49 = lag[(di + mi) * lag_w + dj + mj];
    } else if (lane == 48 && lagr) v49 = 1.0;
    r[48] = v48; r[49] = v49;
  }
  if (__ballot(!lag_ok)) { if (lane == 0) { atomicOr(a.err, 64); a.rec_hdr[rec].n = 0; a.rec_hdr[rec].op = (i0 - r0) * ww + (j0 - c0); } return; }
  const double rho_l = r[49];
  const double c00 = lag[mi * lag_w + mj];
  // relative pivot test: eps * N * max|diag| for the covariance pivots (conditional variances), eps * N / max|diag| for the
  // Lagrange pivot -1' Sigma^-1 1
  const double tol = 2.220446049250313e-16 * (double)(n + 1) * fabs(c00), tol_l = 2.220446049250313e-16 * (double)(n + 1) / fabs(c00);
  double mypiv = 1.0;
  bool singular = false;
  GjStep<0>::run(r, lane, n, lagr, tol, tol_l, mypiv, singular);
  if (singular) {
    if (lane == 0) { atomicOr(a.err, 8); a.rec_hdr[rec].n = 0; a.rec_hdr[rec].op = (i0 - r0) * ww + (j0 - c0); }
    return;
  }
  const double w_l = (lane < n) ? r[49] / mypiv : 0.0;
  double var = a.sill - wave_sum_f64(w_l * rho_l);
  var = fabs(var);
  const double sw = wave_sum_f64(w_l);
  // ---- the cell's record ---------------------------------------------------------------------------------------------
  if (lane < kSgsMaxPts) {
    double2 vw = make_double2(0.0, 0.0);
    int gg = -1;
    if (lane < n) {
      gg = L.nb_g[lane];
      vw.y = w_l;
      const bool inblk = my_i >= r0 && my_i < r1 && my_j >= c0 && my_j < c1;
      const int nb_pos = (my_i - r0) * ww + (my_j - c0);
      const int nb_slot = inblk ? rank[nb_pos] : -1;
      if (nb_slot >= 0)          // simulated earlier in this block: its visiting slot and its block-local index (both < 1024)
        vw.x = __builtin_bit_cast(double, kSgsPendingTag | ((uint64_t)nb_slot << 16) | (uint64_t)nb_pos);
      else if (a.defer)
        vw.x = __builtin_bit_cast(double, inblk ? (kSgsWindowTag | (uint64_t)nb_pos) : (kSgsGridTag | (uint64_t)(uint32_t)gg));
      else
        vw.x = (inblk && a.zcond) ? a.zcond[gg] : g[gg];
    }
    // entry `lane` of 64 consecutive cell slots lies side by side: sgs_sequence_kernel reads one cell per lane
    a.rec_vw[((rec >> 6) * kSgsMaxPts + lane) * 64 + (rec & 63)] = vw;
    if (a.nbr_trace) a.nbr_trace[(size_t)(k_lo + slot) * kSgsMaxPts + lane] = gg;
  }
  if (lane == 0) {
    SgsCellHdr hd;
    hd.n = n; hd.op = (i0 - r0) * ww + (j0 - c0); hd.sdz = sqrt(var) * a.z[k_lo + slot]; hd.var = var;
    // ordinary: est = local mean + sum w (v - local mean) (_krige.py:42) -> c1 = (1 - sum w) / n multiplies sum v;
    // simple:   est = global mean + sum w (v - global mean) (_krige.py:79) -> c1 = global mean * (1 - sum w) is added as it is
    hd.c1 = lagr ? (1.0 - sw) / (double)n : a.gmean[chain] * (1.0 - sw);
    a.rec_hdr[rec] = hd;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// sgs_sequence_kernel: one workgroup of four wavefronts per chain; the block's cells in visiting order, 64 at a time, one
// cell per lane.  The value pass is a sparse unit-lower-triangular solve (I - A) v = c with A[j][k] the coefficient of the
// earlier block cell k in the estimate of cell j (w_jk + (1 - sum w_j) / n_j in ordinary kriging, w_jk in simple kriging)
// and c the part of the estimate that is known already plus sd * z.  Per chunk of 64 cells:
//   gather      (all four waves, every fourth group of four list entries each) every lane walks its own cell's neighbour list:
//               values known before the block and values of cells of EARLIER chunks (final by now, in `overlay`) are summed;
//               the coefficient of a neighbour in the SAME chunk goes to column `lane` of a 64 x 64 tile in LDS, and one bit
//               of a 64-bit mask per lane says which of the tile's entries exist;
//   sequence    (wave 0) cell kk of the chunk is final once the cells before it are: its value is broadcast (v_readlane) and
//               every lane that lists it adds coefficient * value -- two v_readlane and one fma on the chain's critical path
//               per cell, where a wave reduction over the neighbour list per cell cost 650-800 cycles.
// The records of the next chunk are requested (LDS-DMA) before the sequence part of the current one and land meanwhile.
// A cell whose kriging system failed (n == 0, error flag raised by sgs_weights_kernel) gets NaN, and so does every later
// cell of its chunk and every cell that lists one of those.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double readlane_f64(double v, int l) {
  const dev::v2i32 b = __builtin_bit_cast(dev::v2i32, v);
  dev::v2i32 o;
  o.x = __builtin_amdgcn_readlane(b.x, l); o.y = __builtin_amdgcn_readlane(b.y, l);
  return __builtin_bit_cast(double, o);
}
constexpr int kSeqWaves = 4;
__global__ __launch_bounds__(64 * kSeqWaves) void sgs_sequence_kernel(const SgsArgs a) {
  __shared__ double overlay[kSgsMaxWin];
  __shared__ double tile[65 * 64];                              // [cell of the chunk the value comes from][lane = cell that uses it]; row 64 takes the writes of absent entries
  __shared__ __attribute__((aligned(16))) double2 stage[kSgsMaxPts * 64];      // a chunk's neighbour records, [entry][cell]
  __shared__ double2 part[kSeqWaves][64];                       // per wave and cell: (sum v, sum w v) over the wave's share of the list
  __shared__ uint64_t mpart[kSeqWaves][64];
  const int chain = blockIdx.x, lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (!a.rank_ok[chain]) return;
  const int H = a.H, W = a.W;
  double* __restrict__ g = a.grid + (size_t)chain * H * W;
  const int r0 = a.win[4 * chain], r1 = a.win[4 * chain + 1], c0 = a.win[4 * chain + 2], c1 = a.win[4 * chain + 3];
  const int wh = r1 - r0, ww = c1 - c0;
  const int k_lo = a.cell_off[chain], cnt = a.cell_cnt ? a.cell_cnt[chain] : a.cell_off[chain + 1] - k_lo;
  const int np8 = (a.num_points + 7) & ~7;
  const size_t rec0 = (size_t)chain * a.max_cells;              // a multiple of 64 (sgs_fill)
  const double* __restrict__ vw_src = (const double*)(a.rec_vw + rec0 * kSgsMaxPts);
  const double4* __restrict__ hdr_src = (const double4*)(a.rec_hdr + rec0);
  static_assert(sizeof(SgsCellHdr) == 32, "a header is one double4");
  double4 hd = make_double4(0.0, 0.0, 0.0, 0.0);
  // The records of a chunk travel global -> LDS by LDS-DMA (no registers; whole 1 KiB pieces = one entry of 64 cells, piece c by
  // wave c mod 4); the headers (one per lane, the same in every wave) come through registers.
  auto request = [&](int kc) {
    hd = hdr_src[kc + lane];
    dma_to_lds<kSeqWaves, 0>(vw_src + (size_t)(kc >> 6) * kSgsMaxPts * 128, (double*)stage, np8 * 128, wave, lane);
  };
  if (cnt > 0) request(0);
  for (int p = threadIdx.x; p < wh * ww; p += 64 * kSeqWaves) {
    const int gi = (r0 + p / ww) * W + c0 + p % ww;
    overlay[p] = a.zcond ? a.zcond[gi] : g[gi];
  }
  for (int kc = 0; kc < cnt; kc += 64) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's pieces of the chunk's records and its headers have landed
    __syncthreads();                                            // ... and everybody's; the overlay holds every cell of the chunks before
    const int j = kc + lane;
    const uint64_t nop = __builtin_bit_cast(uint64_t, hd.x);
    const int n = (j < cnt) ? (int)(uint32_t)nop : -2, op = (int)(nop >> 32);
    const double sdz = hd.y, var = hd.z, c1 = hd.w;
    // ---- gather ----  (no branches; this wave's entries 4 wave + 16 k + u: all their reads first -- records from LDS, deferred values
    // from the grid, earlier cells from the overlay --, then the sums and the tile's entries)
    double sv = 0.0, swv = 0.0;
    uint64_t mask = 0;
    {
      double2 r[12];
      double gvv[12], ov[12];
#pragma unroll
      for (int q = 0; q < 12; ++q) r[q] = stage[(4 * wave + 16 * (q >> 2) + (q & 3)) * 64 + lane];
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int e = 4 * wave + 16 * (q >> 2) + (q & 3);
        const uint64_t bits = __builtin_bit_cast(uint64_t, r[q].x);
        const uint32_t hi = (uint32_t)(bits >> 32), lo = (uint32_t)bits;
        gvv[q] = g[(e < n && hi == (uint32_t)(kSgsGridTag >> 32)) ? lo : 0u];
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int e = 4 * wave + 16 * (q >> 2) + (q & 3);
        const uint64_t bits = __builtin_bit_cast(uint64_t, r[q].x);
        const uint32_t hi = (uint32_t)(bits >> 32), lo = (uint32_t)bits;
        const bool early = (e < n) && ((hi == (uint32_t)(kSgsPendingTag >> 32) && (int)(lo >> 16) < kc) || hi == (uint32_t)(kSgsWindowTag >> 32));
        ov[q] = overlay[early ? (lo & 0xFFFFu) : 0u];
      }
#pragma unroll
      for (int q = 0; q < 12; ++q) {
        const int e = 4 * wave + 16 * (q >> 2) + (q & 3);
        const uint64_t bits = __builtin_bit_cast(uint64_t, r[q].x);
        const uint32_t hi = (uint32_t)(bits >> 32), lo = (uint32_t)bits;
        const double w = r[q].y;
        const bool pend = hi == (uint32_t)(kSgsPendingTag >> 32), inwin = hi == (uint32_t)(kSgsWindowTag >> 32), ingrid = hi == (uint32_t)(kSgsGridTag >> 32);
        const int ks = (int)(lo >> 16) - kc;                     // >= 0: a cell of this chunk (visited before this one: ks < lane)
        const bool here = (e < n) && pend && ks >= 0, known = (e < n) && !(pend && ks >= 0);
        const double v = ingrid ? gvv[q] : ((pend || inwin) ? ov[q] : r[q].x);
        const double wv = w * v;
        sv += known ? v : 0.0;
        swv += known ? wv : 0.0;
        tile[here ? ks * 64 + lane : 64 * 64 + lane] = (a.ktype == 0) ? w + c1 : w;       // row 64: nobody reads it
        mask |= here ? (1ull << ks) : 0ull;
      }
    }
    part[wave][lane] = make_double2(sv, swv);
    mpart[wave][lane] = mask;
    __syncthreads();                                            // the records have been read: the next chunk's may overwrite them
    if (kc + 64 < cnt) request(kc + 64);
    if (wave != 0) continue;
    // ---- sequence (wave 0) ----
    sv = 0.0; swv = 0.0; mask = 0;
#pragma unroll
    for (int w = 0; w < kSeqWaves; ++w) { const double2 q = part[w][lane]; sv += q.x; swv += q.y; mask |= mpart[w][lane]; }
    // est = mean + sum w (v - mean) (_krige.py:42) as sum w v + sum v * (1 - sum w) / n with c1 = (1 - sum w) / n from the record;
    // simple kriging (_krige.py:79): sum w v + global mean * (1 - sum w) = c1
    // val = est + sd z is what travels: value of the cell = (known part of the estimate + sd z) + the chunk's own cells' part
    // n == 0: error flagged by sgs_weights_kernel -> NaN; a lane without a cell to simulate must hold a finite number (0 * NaN)
    double val = (n > 0) ? (swv + (a.ktype == 0 ? sv * c1 : c1)) + sdz : (n == 0 ? NAN : 0.0);
    const int kend = min(64, cnt - kc);
    for (int k8 = 0; k8 < kend; k8 += 8) {
      double t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { const double tt = tile[(k8 + u) * 64 + lane]; t[u] = ((mask >> (k8 + u)) & 1ull) ? tt : 0.0; }
#pragma unroll
      for (int u = 0; u < 8; ++u) val = __fma_rn(t[u], readlane_f64(val, k8 + u), val);     // lane k8 + u is final: every cell it lists came before
    }
    if (n == -1) {                                              // conditioned already: nothing drawn for it (MCMC.py:141)
      if (a.trace) { a.trace[3 * (k_lo + j)] = -1.0; a.trace[3 * (k_lo + j) + 1] = overlay[op]; a.trace[3 * (k_lo + j) + 2] = 0.0; }
    } else if (n >= 0) {
      overlay[op] = val;
      if (a.trace && n > 0) { a.trace[3 * (k_lo + j)] = (double)n; a.trace[3 * (k_lo + j) + 1] = val - sdz; a.trace[3 * (k_lo + j) + 2] = var; }
    }
  }
  __syncthreads();
  for (int p = threadIdx.x; p < wh * ww; p += 64 * kSeqWaves) g[(size_t)(r0 + p / ww) * W + c0 + p % ww] = overlay[p];
}

static bool sgs_args_ok(const SgsArgs& a) {
  return !(a.hw < 1 || a.num_points < 8 || a.num_points > kSgsMaxPts || a.H < 2 || a.W < 2 || a.H > 32767 || a.W > 32767);
}
// the two halves of launch_sgs_blocks: the records of every cell (visiting ranks, then search + kriging weights), and the value pass
hipError_t launch_sgs_weights(const SgsArgs& a, int launch_cells, hipStream_t st) {
  if (!sgs_args_ok(a)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sgs_rank_kernel, dim3(a.n_chains), dim3(256), 0, st, a);
  if (launch_cells > 0) hipLaunchKernelGGL(sgs_weights_kernel, dim3(launch_cells, a.n_chains), dim3(64), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_sgs_sequence(const SgsArgs& a, hipStream_t st) {
  if (!sgs_args_ok(a)) return hipErrorInvalidValue;
  hipLaunchKernelGGL(sgs_sequence_kernel, dim3(a.n_chains), dim3(64 * kSeqWaves), 0, st, a);
  return hipGetLastError();
}
hipError_t launch_sgs_blocks(const SgsArgs& a, int launch_cells, hipStream_t st) {
  hipError_t e = launch_sgs_weights(a, launch_cells, st);
  return e != hipSuccess ? e : launch_sgs_sequence(a, st);
}

// ---------------------------------------------------------------------------------------------------------------------
// loss of the proposed bed (+ trend) over the whole grid and the thickness guard, one 256-thread workgroup per chain:
//   loss = nansum(residual^2 where mc_mask == 1) / (2 sigma^2)      (MCMC.py:1021-1044 on Topography.py:592-600)
//   bad  = number of cells with guard_mask == 1 and surf - (bed + trend) <= 0   (MCMC.py:1789-1795)
// S.upd is the guard mask here (grounded_ice_mask); trend may be NULL.
// ---------------------------------------------------------------------------------------------------------------------
// Several workgroups per chain (a chain's grid is split into `parts` contiguous ranges of cells), then one thread per chain
// adds the parts in order: with one workgroup per chain a 256 x 256 grid kept 16 CUs busy for 0.19 ms per iteration.
// one part of a chain's map by the 256 threads of a workgroup; thread 0 returns the part's sum and bad-cell count
// (bed[q - off] is cell q of the chain's map: off != 0 when `bed` is a tile of the map in LDS)
__device__ __forceinline__ void sgs_loss_part(const StaticFields& S, const double* bed, const int off, const double* trend, int g_lo, int g_hi, int tid,
                                              double* red, int* redb, double& sum_out, int& bad_out) {
  auto bed_at = [&](int rr, int cc) { const int q = rr * S.W + cc; return trend ? bed[q - off] + trend[q] : bed[q - off]; };
  double hi = 0.0, lo = 0.0;
  int nbad = 0;
  for (int g = g_lo + tid; g < g_hi; g += 256) {
    const int r = g / S.W, c = g - r * S.W;
    double e = 0.0;
    if (S.mc[g] == 1) {
      const double v = cell_residual(S, r, c, bed_at);
      if (!isnan(v)) e = v * v;
    }
    const double s = hi + e, bb = s - hi;
    lo += (hi - (s - bb)) + (e - bb);
    hi = s;
    if (S.upd[g] == 1 && (S.surf[g] - bed_at(r, c)) <= 0.0) ++nbad;
  }
  double t = hi + lo;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) { t += __shfl_xor(t, off, 64); nbad += __shfl_xor(nbad, off, 64); }
  if ((tid & 63) == 0) { red[tid >> 6] = t; redb[tid >> 6] = nbad; }
  __syncthreads();
  sum_out = ((red[0] + red[1]) + red[2]) + red[3];
  bad_out = redb[0] + redb[1] + redb[2] + redb[3];
}
__global__ __launch_bounds__(256) void sgs_loss_kernel(const StaticFields S, const double* beds, const double* trend, int parts,
                                                       double* part_sum, int32_t* part_bad) {
  __shared__ double red[8];
  __shared__ int redb[4];
  const int chain = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int plane = S.H * S.W;
  const int per = (plane + parts - 1) / parts, g_lo = part * per, g_hi = min(plane, g_lo + per);
  double t; int nb;
  sgs_loss_part(S, beds + (size_t)chain * plane, 0, trend, g_lo, g_hi, tid, red, redb, t, nb);
  if (tid == 0) { part_sum[chain * parts + part] = t; part_bad[chain * parts + part] = nb; }
}
__global__ __launch_bounds__(64) void sgs_loss_finish_kernel(int n_chains, int parts, double two_sigma2, const double* __restrict__ part_sum,
                                                             const int32_t* __restrict__ part_bad, double* __restrict__ loss, int32_t* __restrict__ bad) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n_chains) return;
  double s = 0.0;
  int nb = 0;
  for (int p = 0; p < parts; ++p) { s += part_sum[c * parts + p]; nb += part_bad[c * parts + p]; }
  loss[c] = s / two_sigma2;
  bad[c] = nb;
}

// parts of a chain's map: a function of the grid alone (a chain's loss must not depend on how many chains share the handle);
// ~1024 cells each, so that a few chains still spread over the chip (4 chains at 64 x 64: +6 % per iteration against 4096)
int sgs_loss_parts(const StaticFields& S) { return std::max(1, std::min(64, (S.H * S.W + 1023) / 1024)); }

hipError_t launch_sgs_loss(const StaticFields& S, int n_chains, const double* beds, const double* trend, double* loss, int32_t* bad,
                           double* part_sum, int32_t* part_bad, hipStream_t st) {
  const int parts = sgs_loss_parts(S);
  hipLaunchKernelGGL(sgs_loss_kernel, dim3(n_chains, parts), dim3(256), 0, st, S, beds, trend, parts, part_sum, part_bad);
  hipLaunchKernelGGL(sgs_loss_finish_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, st, n_chains, parts, S.two_sigma2, part_sum, part_bad, loss, bad);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// Windowed iteration end for chains WITHOUT a normal-score transformer (gsm.h: gsm_sgs_state_init / gsm_sgs_finish).  The
// reference recomputes residual, loss and guard over the whole map every iteration (MCMC.py:1781-1795); without a
// transformer only the block changes, so only the residuals of the block and its one-cell halo change (5-point stencil).
// Carried per chain: the plane of squared residuals that enter the loss (`energy`, 0 where mc_mask != 1 or the residual is
// NaN), their compensated sum, the loss, and the number of grounded cells with non-positive thickness.
//   state[4 c .. 4 c + 3] = (sum hi, sum lo, loss = sum / (2 sigma^2), bad cells)
// One 256-thread workgroup per chain: loss / guard of the proposal from the halo window, the acceptance test, the commit.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sgs_state_init_kernel(const StaticFields S, const double* beds, const double* trend, double* energy,
                                                             double* state) {
  __shared__ double red[8];
  __shared__ int redb[4];
  const int chain = blockIdx.x, tid = threadIdx.x;
  const int plane = S.H * S.W;
  const double* bed = beds + (size_t)chain * plane;
  double* en = energy + (size_t)chain * plane;
  auto bed_at = [&](int rr, int cc) { const int q = rr * S.W + cc; return trend ? bed[q] + trend[q] : bed[q]; };
  double hi = 0.0, lo = 0.0;
  int nbad = 0;
  for (int g = tid; g < plane; g += 256) {
    const int r = g / S.W, c = g - r * S.W;
    double e = 0.0;
    if (S.mc[g] == 1) {
      const double v = cell_residual(S, r, c, bed_at);
      if (!isnan(v)) e = v * v;
    }
    en[g] = e;
    double sm, er;
    dev::two_sum(hi, e, sm, er);
    hi = sm; lo += er;
    if (S.upd[g] == 1 && (S.surf[g] - bed_at(r, c)) <= 0.0) ++nbad;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const double ohi = __shfl_xor(hi, off, 64), olo = __shfl_xor(lo, off, 64);
    double sm, er;
    dev::two_sum(hi, ohi, sm, er);
    hi = sm; lo += olo + er;
    nbad += __shfl_xor(nbad, off, 64);
  }
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = hi; red[2 * (tid >> 6) + 1] = lo; redb[tid >> 6] = nbad; }
  __syncthreads();
  if (tid == 0) {
    double th = 0.0, tl = 0.0;
    for (int w = 0; w < 4; ++w) { double sm, er; dev::two_sum(th, red[2 * w], sm, er); th = sm; tl += red[2 * w + 1] + er; }
    double sm, er;
    dev::two_sum(th, tl, sm, er);
    state[4 * chain] = sm; state[4 * chain + 1] = er; state[4 * chain + 2] = (sm + er) / S.two_sigma2;
    state[4 * chain + 3] = (double)(redb[0] + redb[1] + redb[2] + redb[3]);
  }
}

constexpr int kSgsHaloMax = 36 * 36;       // cells of block + halo held in LDS: blocks up to 34 x 34 (or e.g. 16 x 70)
__global__ __launch_bounds__(256) void sgs_finish_kernel(const StaticFields S, double* cur, double* next, const double* trend, double* energy,
                                                         double* state, const int32_t* win, const double* u, uint32_t* resampled,
                                                         uint8_t* accept, double* loss_rec, uint8_t* acc_rec, int64_t rec_stride, int32_t* err) {
  __shared__ double e_new[kSgsHaloMax];
  __shared__ double red[8];
  __shared__ int redb[8];
  __shared__ int acc_s;
  const int chain = blockIdx.x, tid = threadIdx.x;
  const int H = S.H, W = S.W;
  const size_t base = (size_t)chain * H * W;
  const int r0 = win[4 * chain], r1 = win[4 * chain + 1], c0 = win[4 * chain + 2], c1 = win[4 * chain + 3];
  const int hr0 = max(0, r0 - 1), hr1 = min(H, r1 + 1), hc0 = max(0, c0 - 1), hc1 = min(W, c1 + 1);
  const int hww = hc1 - hc0, hn = (hr1 - hr0) * hww;
  if (r0 < 0 || c0 < 0 || r1 > H || c1 > W || r1 < r0 || c1 < c0 || hn > kSgsHaloMax) {
    if (tid == 0) { atomicOr(err, 1); accept[chain] = 0; if (acc_rec) acc_rec[(int64_t)chain * rec_stride] = 0; if (loss_rec) loss_rec[(int64_t)chain * rec_stride] = state[4 * chain + 2]; }
    return;
  }
  const double* nb = next + base;
  auto next_at = [&](int rr, int cc) { const int q = rr * W + cc; return trend ? nb[q] + trend[q] : nb[q]; };
  double s_old = 0.0, s_new = 0.0;
  int bad_old = 0, bad_new = 0;
  for (int p = tid; p < hn; p += 256) {
    const int r = hr0 + p / hww, c = hc0 + p % hww, g = r * W + c;
    double e = 0.0;
    if (S.mc[g] == 1) {
      const double v = cell_residual(S, r, c, next_at);
      if (!isnan(v)) e = v * v;
    }
    e_new[p] = e;
    s_new += e;
    s_old += energy[base + g];
    if (r >= r0 && r < r1 && c >= c0 && c < c1 && S.upd[g] == 1) {
      const double t = trend ? trend[g] : 0.0;
      bad_new += (S.surf[g] - next_at(r, c)) <= 0.0;
      bad_old += (S.surf[g] - (trend ? cur[base + g] + t : cur[base + g])) <= 0.0;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    s_old += __shfl_xor(s_old, off, 64); s_new += __shfl_xor(s_new, off, 64);
    bad_old += __shfl_xor(bad_old, off, 64); bad_new += __shfl_xor(bad_new, off, 64);
  }
  if ((tid & 63) == 0) { red[2 * (tid >> 6)] = s_old; red[2 * (tid >> 6) + 1] = s_new; redb[2 * (tid >> 6)] = bad_old; redb[2 * (tid >> 6) + 1] = bad_new; }
  __syncthreads();
  if (tid == 0) {
    const double so = ((red[0] + red[2]) + red[4]) + red[6], sn = ((red[1] + red[3]) + red[5]) + red[7];
    const int bo = redb[0] + redb[2] + redb[4] + redb[6], bn = redb[1] + redb[3] + redb[5] + redb[7];
    double hi = state[4 * chain], lo = state[4 * chain + 1];
    double sm, er;
    dev::two_sum(hi, -so, sm, er); hi = sm; lo += er;
    dev::two_sum(hi, sn, sm, er); hi = sm; lo += er;
    dev::two_sum(hi, lo, sm, er); hi = sm; lo = er;
    const double bad_next = state[4 * chain + 3] - (double)bo + (double)bn;
    const double lp = state[4 * chain + 2];
    const double ln = (bad_next > 0.0) ? INFINITY : (hi + lo) / S.two_sigma2;
    bool acc = true;
    if (!(lp > ln)) {
      const double p = exp(lp - ln);
      acc = u[chain] <= ((p > 1.0) ? 1.0 : p);        // p NaN: the comparison is false (numpy.minimum propagates the NaN)
    }
    if (acc) { state[4 * chain] = hi; state[4 * chain + 1] = lo; state[4 * chain + 2] = ln; state[4 * chain + 3] = bad_next; }
    accept[chain] = acc ? 1 : 0;
    if (loss_rec) loss_rec[(int64_t)chain * rec_stride] = acc ? ln : lp;
    if (acc_rec) acc_rec[(int64_t)chain * rec_stride] = acc ? 1 : 0;
    acc_s = acc ? 1 : 0;
  }
  __syncthreads();
  const bool acc = acc_s != 0;
  if (acc) for (int p = tid; p < hn; p += 256) energy[base + (size_t)(hr0 + p / hww) * W + hc0 + p % hww] = e_new[p];
  const int ww = c1 - c0, n = (r1 - r0) * ww;
  for (int p = tid; p < n; p += 256) {
    const size_t q = base + (size_t)(r0 + p / ww) * W + c0 + p % ww;
    if (acc) { cur[q] = next[q]; resampled[q] += 1u; }
    else next[q] = cur[q];
  }
}

hipError_t launch_sgs_state_init(const StaticFields& S, int n_chains, const double* beds, const double* trend, double* energy, double* state,
                                 hipStream_t st) {
  hipLaunchKernelGGL(sgs_state_init_kernel, dim3(n_chains), dim3(256), 0, st, S, beds, trend, energy, state);
  return hipGetLastError();
}
hipError_t launch_sgs_finish(const StaticFields& S, int n_chains, double* cur, double* next, const double* trend, double* energy, double* state,
                             const int32_t* win, const double* u, uint32_t* resampled, uint8_t* accept, double* loss_rec, uint8_t* acc_rec,
                             int64_t rec_stride, int32_t* err, hipStream_t st) {
  hipLaunchKernelGGL(sgs_finish_kernel, dim3(n_chains), dim3(256), 0, st, S, cur, next, trend, energy, state, win, u, resampled, accept,
                     loss_rec, acc_rec, rec_stride, err);
  return hipGetLastError();
}

// the acceptance test of an iteration (gsm.h: gsm_sgs_decide), one thread per chain
__global__ __launch_bounds__(64) void sgs_decide_kernel(int n, const double* __restrict__ loss_next, const int32_t* __restrict__ bad,
                                                        const double* __restrict__ u, double* __restrict__ loss_prev,
                                                        uint8_t* __restrict__ accept, double* loss_rec, uint8_t* acc_rec, int64_t rec_stride) {
  const int c = blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  const double ln = (bad[c] > 0) ? INFINITY : loss_next[c];
  const double lp = loss_prev[c];
  bool acc = true;
  if (!(lp > ln)) {
    const double p = exp(lp - ln);
    acc = u[c] <= ((p > 1.0) ? 1.0 : p);        // p NaN: the comparison is false (numpy.minimum propagates the NaN)
  }
  const double l = acc ? ln : lp;
  loss_prev[c] = l;
  accept[c] = acc ? 1 : 0;
  if (loss_rec) loss_rec[(int64_t)c * rec_stride] = l;
  if (acc_rec) acc_rec[(int64_t)c * rec_stride] = acc ? 1 : 0;
}

hipError_t launch_sgs_decide(int n_chains, const double* loss_next, const int32_t* bad, const double* u, double* loss_prev,
                             uint8_t* accept, double* loss_rec, uint8_t* acc_rec, int64_t rec_stride, hipStream_t st) {
  hipLaunchKernelGGL(sgs_decide_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, st, n_chains, loss_next, bad, u, loss_prev, accept,
                     loss_rec, acc_rec, rec_stride);
  return hipGetLastError();
}

// Philox mode: the draws of one iteration of one chain (gsm.h: gsm_sgs_draw_philox), one 64-lane workgroup per (iteration, chain)
constexpr uint32_t kStreamSgs = 4;
__global__ __launch_bounds__(64) void sgs_draw_kernel(const SgsDrawArgs a) {
  __shared__ double mt[kMathTabDoubles];
  __shared__ uint32_t key[kSgsMaxWin];
  __shared__ int geo[8];
  const int chain = blockIdx.x, j = blockIdx.y, lane = threadIdx.x;
  const int64_t rec = (int64_t)j * a.n_chains + chain, it = a.iter0 + j;
  const uint64_t seed = a.seeds[chain];
  for (int i = lane; i < kMathTabDoubles; i += 64) mt[i] = a.mathtab[i];
  if (lane == 0) {
    int row = 0, col = 0;
    for (int att = 0; att < 64; ++att) {
      const u32x4 r = philox_draw(seed, it, kStreamSgs, (uint32_t)att);
      row = (int)__umulhi(r.x, (uint32_t)a.H); col = (int)__umulhi(r.y, (uint32_t)a.W);
      if (!a.region_mask || a.region_mask[row * a.W + col] == 1) break;
      if (att == 63) atomicOr(a.err, 16);
    }
    const u32x4 r = philox_draw(seed, it, kStreamSgs, 64u);
    const int bsx = a.min_x + (int)__umulhi(r.x, (uint32_t)(a.max_x - a.min_x));
    const int bsy = a.min_y + (int)__umulhi(r.y, (uint32_t)(a.max_y - a.min_y));
    // int(ix -/+ bs / 2) of MCMC.py:1758-1761 (truncation towards zero of a half-integer)
    const int r0 = max(0, (2 * row - bsx) / 2), r1 = min(a.H, (2 * row + bsx) / 2);
    const int c0 = max(0, (2 * col - bsy) / 2), c1 = min(a.W, (2 * col + bsy) / 2);
    a.win[4 * rec] = r0; a.win[4 * rec + 1] = r1; a.win[4 * rec + 2] = c0; a.win[4 * rec + 3] = c1;
    a.blk[4 * rec] = row; a.blk[4 * rec + 1] = col; a.blk[4 * rec + 2] = bsx; a.blk[4 * rec + 3] = bsy;
    a.u[rec] = u01_from(r.z, r.w);
    geo[0] = r0; geo[1] = r1; geo[2] = c0; geo[3] = c1;
  }
  __syncthreads();
  const int r0 = geo[0], r1 = geo[1], c0 = geo[2], c1 = geo[3];
  const int ww = c1 - c0, n = max(0, r1 - r0) * max(0, ww);
  if (lane == 0) { a.cell_off[rec] = (int32_t)(rec * a.max_cells); a.cell_cnt[rec] = (n <= a.max_cells && n <= kSgsMaxWin) ? n : 0; }
  if (n > a.max_cells || n > kSgsMaxWin) { if (lane == 0) atomicOr(a.err, 1); return; }
  for (int p = lane; p < n; p += 64) key[p] = philox_draw(seed, it, kStreamSgs, 128u + (uint32_t)p).x;
  __syncthreads();
  for (int p = lane; p < n; p += 64) {
    const uint32_t kp = key[p];
    int rank = 0;
    for (int q = 0; q < n; ++q) { const uint32_t kq = key[q]; rank += (kq < kp || (kq == kp && q < p)) ? 1 : 0; }
    const int i = r0 + p / ww, jj = c0 + p % ww;
    const int64_t o = rec * a.max_cells + rank;
    a.cells[2 * o] = i; a.cells[2 * o + 1] = jj;
    double g1, g2;
    normals2(seed, it, kStreamSgs, 2048u + (uint32_t)(p >> 1), g1, g2, mt);
    a.z[o] = a.is_data[i * a.W + jj] ? 0.0 : ((p & 1) ? g2 : g1);
  }
}
hipError_t launch_sgs_draw(const SgsDrawArgs& a, hipStream_t st) {
  hipLaunchKernelGGL(sgs_draw_kernel, dim3(a.n_chains, a.n_iters), dim3(64), 0, st, a);
  return hipGetLastError();
}

// normal-score transform, elementwise (gsm.h: gsm_qt_transform); the two tables are staged in LDS when they fit
constexpr int kQtLds = 4096;
__global__ __launch_bounds__(256) void qt_kernel(const double* __restrict__ quantiles, const double* __restrict__ references, int nq,
                                                 double clip_min, double clip_max, const double* x, double* out, int64_t n, int inverse) {
  __shared__ double tab[2 * kQtLds];
  const double* q = quantiles;
  const double* ref = references;
  if (nq <= kQtLds) {
    for (int i = threadIdx.x; i < nq; i += 256) { tab[i] = quantiles[i]; tab[kQtLds + i] = references[i]; }
    __syncthreads();
    q = tab; ref = tab + kQtLds;
  }
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = inverse ? ns::qt_inverse(x[i], q, ref, nq) : ns::qt_forward(x[i], q, ref, nq, clip_min, clip_max);
}
hipError_t launch_qt(const double* quantiles, const double* references, int nq, double clip_min, double clip_max, const double* x,
                     double* out, int64_t n, int inverse, hipStream_t st) {
  const int blocks = (int)std::min<int64_t>((n + 255) / 256, 4096);
  hipLaunchKernelGGL(qt_kernel, dim3(blocks), dim3(256), 0, st, quantiles, references, nq, clip_min, clip_max, x, out, n, inverse);
  return hipGetLastError();
}

// gsm_sgs_commit_map: an accepted chain takes the whole proposed plane
__global__ __launch_bounds__(256) void sgs_commit_map_kernel(int H, int W, double* cur, const double* proposed, uint32_t* resampled,
                                                             const int32_t* win, const uint8_t* accept) {
  const int chain = blockIdx.x;
  if (!accept[chain]) return;
  const size_t base = (size_t)chain * H * W;
  for (int p = threadIdx.x; p < H * W; p += 256) cur[base + p] = proposed[base + p];
  const int r0 = win[4 * chain], r1 = win[4 * chain + 1], c0 = win[4 * chain + 2], c1 = win[4 * chain + 3];
  const int ww = c1 - c0, n = (r1 - r0) * ww;
  for (int p = threadIdx.x; p < n; p += 256) resampled[base + (size_t)(r0 + p / ww) * W + c0 + p % ww] += 1u;
}
hipError_t launch_sgs_commit_map(int H, int W, int n_chains, double* cur, const double* proposed, uint32_t* resampled, const int32_t* win,
                                 const uint8_t* accept, hipStream_t st) {
  hipLaunchKernelGGL(sgs_commit_map_kernel, dim3(n_chains), dim3(256), 0, st, H, W, cur, proposed, resampled, win, accept);
  return hipGetLastError();
}

// accept[c] != 0: the block of chain c goes from `next` to `cur` and its resampled counts are bumped (MCMC.py:1803-1812);
// else the block of `next` is restored from `cur`, so that next == cur everywhere again.
__global__ __launch_bounds__(64) void sgs_commit_kernel(int H, int W, double* cur, double* next, uint32_t* resampled, const int32_t* win,
                                                        const uint8_t* accept) {
  const int chain = blockIdx.x;
  const size_t base = (size_t)chain * H * W;
  const int r0 = win[4 * chain], r1 = win[4 * chain + 1], c0 = win[4 * chain + 2], c1 = win[4 * chain + 3];
  const int ww = c1 - c0, n = (r1 - r0) * ww;
  const bool acc = accept[chain] != 0;
  for (int p = threadIdx.x; p < n; p += 64) {
    const size_t q = base + (size_t)(r0 + p / ww) * W + c0 + p % ww;
    if (acc) { cur[q] = next[q]; resampled[q] += 1u; }
    else next[q] = cur[q];
  }
}

hipError_t launch_sgs_commit(int H, int W, int n_chains, double* cur, double* next, uint32_t* resampled, const int32_t* win,
                             const uint8_t* accept, hipStream_t st) {
  hipLaunchKernelGGL(sgs_commit_kernel, dim3(n_chains), dim3(64), 0, st, H, W, cur, next, resampled, win, accept);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// The end of an iteration of gsm_sgs_iterate in ONE launch: loss parts as sgs_loss_kernel; the workgroup of a chain that
// finishes last (a ticket per chain, device-scope fences) adds the parts in order (= sgs_loss_finish_kernel), takes the
// decision (= sgs_decide_kernel) and commits (mode 1 = sgs_commit_map_kernel: `beds` is the proposed plane; mode 2 =
// sgs_commit_kernel: `beds` is `next`).  Four launches of ~5 us each at 4 chains become one.  The ticket resets itself.
// ---------------------------------------------------------------------------------------------------------------------
struct SgsTailArgs {
  const double* trend; int parts; double* part_sum; int32_t* part_bad; int32_t* ticket;
  double* loss; int32_t* bad; const double* u; double* loss_prev; uint8_t* accept; double* loss_rec; uint8_t* acc_rec; int64_t rec_stride;
  int mode; double* cur; double* beds; uint32_t* resampled; const int32_t* win;
  // QT: the two normal-score transforms of an iteration in this launch as well (see below)
  const double* qt_q; const double* qt_ref; int qt_n; double clip_min, clip_max; double* next; double* next_acc;
};
// QT (mode 1 only; tables of <= kTailQt entries, parts of <= kTailTile cells with their two halo rows): `beds` (the proposed plane) is not
// read but MADE here -- every workgroup inverse-transforms its part of `next` and one row either side into LDS (qt_kernel's arithmetic:
// the same functions on the same table values), stores its own part, and scores from LDS -- and so is the NEXT iteration's forward transform:
// the part's T(proposed) goes to `next_acc`; the chain's last workgroup copies it to `next` if the proposal is accepted (cur = proposed,
// so T(cur) = T(proposed)) and otherwise re-transforms the block window of `cur` (everything else of `next` still is T(cur)).  Two
// elementwise launches per iteration less (MCMC.py:1766, :1777 stay where they are in the arithmetic).
constexpr int kTailQt = 1024, kTailTile = 2048;
template <bool QT>
__global__ __launch_bounds__(256) void sgs_loss_tail_kernel(const StaticFields S, const SgsTailArgs a) {
  __shared__ double red[8];
  __shared__ int redb[4];
  __shared__ int s_last, s_acc;
  __shared__ double qtab[QT ? 2 * kTailQt : 1];
  __shared__ double tile[QT ? kTailTile : 1];
  const int chain = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int H = S.H, W = S.W, plane = H * W, parts = a.parts;
  const int per = (plane + parts - 1) / parts, g_lo = part * per, g_hi = min(plane, g_lo + per);
  const size_t base = (size_t)chain * plane;
  const double* bed = a.beds + base;
  int off = 0;
  if (QT) {
    for (int i = tid; i < a.qt_n; i += 256) { qtab[i] = a.qt_q[i]; qtab[kTailQt + i] = a.qt_ref[i]; }
    __syncthreads();
    const int lo_h = max(0, g_lo - W), hi_h = min(plane, g_hi + W);
    for (int idx = lo_h + tid; idx < hi_h; idx += 256) {
      const double v = ns::qt_inverse(a.next[base + idx], qtab, qtab + kTailQt, a.qt_n);
      tile[idx - lo_h] = v;
      if (idx >= g_lo && idx < g_hi) {
        a.beds[base + idx] = v;
        a.next_acc[base + idx] = ns::qt_forward(v, qtab, qtab + kTailQt, a.qt_n, a.clip_min, a.clip_max);
      }
    }
    __syncthreads();
    bed = tile; off = lo_h;
  }
  double t; int nb;
  sgs_loss_part(S, bed, off, a.trend, g_lo, g_hi, tid, red, redb, t, nb);
  if (tid == 0) {
    __hip_atomic_store(&a.part_sum[chain * parts + part], t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(&a.part_bad[chain * parts + part], nb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int ticket = __hip_atomic_fetch_add(&a.ticket[chain], 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    s_last = ticket == parts - 1;
  }
  __syncthreads();
  if (!s_last) return;
  if (tid == 0) {
    double s = 0.0;
    int nbad = 0;
    for (int p = 0; p < parts; ++p) {
      s += __hip_atomic_load(&a.part_sum[chain * parts + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      nbad += __hip_atomic_load(&a.part_bad[chain * parts + p], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    a.ticket[chain] = 0;
    const double loss = s / S.two_sigma2;
    a.loss[chain] = loss; a.bad[chain] = nbad;
    const double ln = (nbad > 0) ? INFINITY : loss;
    const double lp = a.loss_prev[chain];
    bool acc = true;
    if (!(lp > ln)) {
      const double p = exp(lp - ln);
      acc = a.u[chain] <= ((p > 1.0) ? 1.0 : p);        // p NaN: the comparison is false (numpy.minimum propagates the NaN)
    }
    const double l = acc ? ln : lp;
    a.loss_prev[chain] = l;
    a.accept[chain] = acc ? 1 : 0;
    if (a.loss_rec) a.loss_rec[(int64_t)chain * a.rec_stride] = l;
    if (a.acc_rec) a.acc_rec[(int64_t)chain * a.rec_stride] = acc ? 1 : 0;
    s_acc = acc ? 1 : 0;
  }
  __syncthreads();
  const bool acc = s_acc != 0;
  const int r0 = a.win[4 * chain], r1 = a.win[4 * chain + 1], c0 = a.win[4 * chain + 2], c1 = a.win[4 * chain + 3];
  const int ww = c1 - c0, n = (r1 - r0) * ww;
  if (QT) {
    // the other workgroups' parts of `beds` and `next_acc` were stored before their tickets (release); thread 0 took the last ticket (acquire)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (acc) {
      for (int p = tid; p < plane; p += 256) { a.cur[base + p] = a.beds[base + p]; a.next[base + p] = a.next_acc[base + p]; }
      for (int p = tid; p < n; p += 256) a.resampled[base + (size_t)(r0 + p / ww) * W + c0 + p % ww] += 1u;
    } else {
      for (int p = tid; p < n; p += 256) {
        const size_t q = base + (size_t)(r0 + p / ww) * W + c0 + p % ww;
        a.next[q] = ns::qt_forward(a.cur[q], qtab, qtab + kTailQt, a.qt_n, a.clip_min, a.clip_max);
      }
    }
  } else if (a.mode == 1) {
    if (!acc) return;
    for (int p = tid; p < plane; p += 256) a.cur[base + p] = a.beds[base + p];
    for (int p = tid; p < n; p += 256) a.resampled[base + (size_t)(r0 + p / ww) * W + c0 + p % ww] += 1u;
  } else {
    for (int p = tid; p < n; p += 256) {
      const size_t q = base + (size_t)(r0 + p / ww) * W + c0 + p % ww;
      if (acc) { a.cur[q] = a.beds[q]; a.resampled[q] += 1u; }
      else a.beds[q] = a.cur[q];
    }
  }
}
// whether the tail launch can make the two transforms of an iteration as well (sgs_loss_tail_kernel<true>)
bool sgs_tail_takes_qt(const StaticFields& S, int nq) {
  const int plane = S.H * S.W, parts = sgs_loss_parts(S), per = (plane + parts - 1) / parts;
  return nq >= 1 && nq <= kTailQt && per + 2 * S.W <= kTailTile;
}
hipError_t launch_sgs_loss_tail(const StaticFields& S, int n_chains, const double* trend, double* part_sum, int32_t* part_bad, int32_t* ticket,
                                double* loss, int32_t* bad, const double* u, double* loss_prev, uint8_t* accept, double* loss_rec,
                                uint8_t* acc_rec, int64_t rec_stride, int mode, double* cur, double* beds, uint32_t* resampled,
                                const int32_t* win, hipStream_t st, const double* qt_q, const double* qt_ref, int qt_n, double clip_min,
                                double clip_max, double* next, double* next_acc) {
  SgsTailArgs a;
  a.trend = trend; a.parts = sgs_loss_parts(S); a.part_sum = part_sum; a.part_bad = part_bad; a.ticket = ticket;
  a.loss = loss; a.bad = bad; a.u = u; a.loss_prev = loss_prev; a.accept = accept; a.loss_rec = loss_rec; a.acc_rec = acc_rec;
  a.rec_stride = rec_stride; a.mode = mode; a.cur = cur; a.beds = beds; a.resampled = resampled; a.win = win;
  a.qt_q = qt_q; a.qt_ref = qt_ref; a.qt_n = qt_n; a.clip_min = clip_min; a.clip_max = clip_max; a.next = next; a.next_acc = next_acc;
  if (qt_q) {
    if (mode != 1 || !sgs_tail_takes_qt(S, qt_n) || !next || !next_acc) return hipErrorInvalidValue;
    hipLaunchKernelGGL(sgs_loss_tail_kernel<true>, dim3(n_chains, a.parts), dim3(256), 0, st, S, a);
  } else
    hipLaunchKernelGGL(sgs_loss_tail_kernel<false>, dim3(n_chains, a.parts), dim3(256), 0, st, S, a);
  return hipGetLastError();
}

}  // namespace gsm
