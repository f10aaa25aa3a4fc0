// The Metropolis step of the large-scale chain (reference gstatsMCMC/MCMC.py:1263-1360, Topography.py:592-600) in
// "strip" form for gfx950: 512-thread workgroups (two chains per CU), the window of a step cut into strips that one
// (part of a) wavefront marches down row by row with the lanes along the columns.
//
//   phase A  every strip loads the chain state of its own cells ONCE (bed, carried energy: all loads of the step issued
//            back to back), forms the candidate bed (MCMC.py:1279-1290), checks the thickness guard (MCMC.py:1321-1329),
//            sums the carried energy, and writes the candidate bed -- plus the unchanged bed of the halo ring around the
//            window -- into an LDS tile of (bh + 2) x (bw + 2) doubles.  The tile IS the proposal's field tile: emit_field
//            wrote f there, every cell is read (f) and rewritten (bed') by its one owner, so no second buffer is needed.
//   barrier
//   phase D  every strip marches down its rows plus one halo row above and below: thickness and mass fluxes
//            qx = velx * (surf - bed'), qy = vely * (surf - bed') of a row from the tile, kept in registers for three
//            iterations: the y-difference of the 5-point stencil is between the row before and the row after in the SAME
//            lane, the x-difference comes from the neighbour lanes through DPP wave shifts (v_mov_b32 wave_shr:1 /
//            wave_shl:1).  A strip recomputes the fluxes of one halo column on either side (lanes 0 and C - 1 of its C
//            lanes) and of one halo row above and below instead of exchanging them.  New energies stay in registers.
//   R / E    reduction, accept test, commit (candidate bed back from the tile, energies from registers).
//
// Against the flux-tile kernels (step_flux_kernel.hip: two tiles of fluxes, 105 KiB, one 1024-thread workgroup per CU):
// 52 KiB of LDS shared with the proposal, and lane = column means that a row's addresses are one per-lane offset (fixed
// for the step) plus a scalar row offset: no per-cell index arithmetic, no per-cell predicates -- column predicates are lane
// masks in scalar registers for the whole step, row predicates one 32-bit compare per row.
//
// Decomposition of a wh x ww window (strip::config): C = 64, 32 or 16 lanes per strip, each owning C - 2 columns, so
// that g = ceil(ww / (C - 2)) column groups waste the fewest lanes; the 8 * 64 / C strips of the workgroup are dealt as
// g column groups x sr row strips of n = ceil(wh / sr) rows.  n <= kNR for every block of the table (host-checked).
//
// The arithmetic per cell is that of step_flux_kernel.hip, operation for operation; the sums of a step are taken in the
// order this file defines (per lane down the rows, DPP tree over the lanes, fixed tree over the 8 waves), in the fused
// chain kernel and in the replay kernel alike (chain_strip_kernel.hip): both call the functions below.
#pragma once
#include "gsm_internal.h"
#include "device_util.h"
#include <math.h>

namespace gsm {
namespace strip {

using namespace dev;

constexpr int kST = 512;          // threads per workgroup
constexpr int kSW = kST / 64;     // wavefronts per workgroup
constexpr int kNR = 16;           // owned rows per strip, at most

struct Cfg { int cs, g, sr, n; };   // log2(lanes per strip), column groups, row strips, rows per strip

// ceil(wh / sr) for the row-strip counts config() produces (8, 6, 5, 4, 2, 1), without a division
__host__ __device__ inline int rows_per_strip(int wh, int sr) {
  switch (sr) {
    case 8: return (wh + 7) >> 3;
    case 6: return (int)(((uint32_t)(wh + 5) * 10923u) >> 16);   // exact for wh < 8192
    case 5: return (int)(((uint32_t)(wh + 4) * 13108u) >> 16);
    case 4: return (wh + 3) >> 2;
    case 2: return (wh + 1) >> 1;
    default: return wh;
  }
}
__host__ __device__ inline Cfg config(int wh, int ww) {
  Cfg c;
  if (ww <= 62) { c.cs = 6; c.g = 1; c.sr = 8; }
  else if (ww <= 70) { c.cs = 4; c.g = 5; c.sr = 6; }       // 32 strips of 16 lanes: 5 x 6
  else if (ww <= 90) { c.cs = 5; c.g = 3; c.sr = 5; }       // 16 strips of 32 lanes: 3 x 5
  else if (ww <= 124) { c.cs = 6; c.g = 2; c.sr = 4; }
  else if (ww <= 248) { c.cs = 6; c.g = 4; c.sr = 2; }
  else { c.cs = 6; c.g = 8; c.sr = 1; }                     // ww <= 496
  c.n = rows_per_strip(wh, c.sr);
  return c;
}
// the largest window a block table may hold
__host__ inline bool table_ok(int max_bh, int max_bw) {
  if (max_bw > 496) return false;
  for (int ww = 1; ww <= max_bw; ++ww) if (config(max_bh, ww).n > kNR) return false;
  return true;
}

// t / d for t < 64, 1 <= d <= 8 (multipliers 2^15 / d + 1 packed in two constants)
__device__ __forceinline__ int small_div(int t, int d) {
  const uint64_t lo = 32769ull | (16385ull << 16) | (10923ull << 32) | (8193ull << 48);     // d = 1 .. 4
  const uint64_t hi = 6554ull | (5462ull << 16) | (4682ull << 32) | (4097ull << 48);        // d = 5 .. 8
  const uint32_t m = (uint32_t)(((d <= 4) ? lo : hi) >> (16 * ((d - 1) & 3))) & 0xFFFFu;
  return (int)(((uint32_t)t * m) >> 15);
}

// geometry of a step's window (MCMC.py:1266-1276) -- uniform
struct Window {
  int r0, c0, wh, ww;      // origin and size of the clipped window
  int mr0, mc0;            // first row / column of the block that lies inside the grid
  int bw;                  // block width (row stride of the proposal field)
  bool interior;           // a halo ring on all four sides lies inside the grid
};
__device__ __forceinline__ Window make_window(int H, int W, int row, int col, int bh, int bw) {
  Window g;
  const int r0 = max(0, row - bh / 2), r1 = min(H, row + bh / 2);
  const int c0 = max(0, col - bw / 2), c1 = min(W, col + bw / 2);
  g.r0 = r0; g.c0 = c0; g.wh = r1 - r0; g.ww = c1 - c0;
  g.mr0 = max(bh - r1, 0); g.mc0 = max(bw - c1, 0);
  g.bw = bw;
  g.interior = (r0 > 0) && (r1 < H) && (c0 > 0) && (c1 < W);
  return g;
}

// what a lane does in a step: fixed for the step.  The yes / no facts are bits of one register; a phase turns the ones it
// needs into lane masks (scalar registers) for its own duration.
enum : uint32_t {
  kFValid = 1u,        // the lane's column exists (strip in use, column inside the grid and within the halo of the window)
  kFColIn = 2u,        // ... and lies inside the window
  kFColOwn = 4u,       // ... and is one of the strip's own columns
  kFTopIn = 8u,        // the row of iteration 0 lies inside the window (a strip above owns it)
  kFBelowIn = 16u,     // the row of iteration rows + 1 lies inside the window (a strip below owns it)
  kFAtLeft = 32u, kFAtRight = 64u,   // border windows: the lane's column is grid column 0 / W - 1
};
struct Lane {
  uint32_t cell0;          // flat grid index of (row of iteration 0, the lane's column); wraps below zero for a border strip
  int fidx;                // index of that cell in the bh x bw proposal field (replay kernel: the field comes from HBM)
  int tidx;                // index of that cell in the LDS tile [(bh + 2)][bw + 2] (block cell (y, x) at (y + 1, x + 1))
  int rows;                // owned rows: iterations 1 .. rows (-2: the lane has nothing to do)
  uint32_t flags;
  int lo, hi;              // border windows: iterations whose row exists and is needed: lo .. hi (hi < lo: none)
  int jtop, jbot;          // border windows: iteration of grid row 0 / H - 1
};
__device__ __forceinline__ Lane lane_setup(const int lane, const int wave, const Cfg& c, const Window& g, const int H, const int W) {
  Lane L;
  const int C = 1 << c.cs;
  const int sub = lane >> c.cs, l = lane & (C - 1);
  const int sidx = wave * (64 >> c.cs) + sub;
  const int kr = small_div(sidx, c.g), kc = sidx - kr * c.g;
  const int wc = kc * (C - 2) + l - 1;                    // window column, -1 .. ww (halo columns included)
  const int gc = g.c0 + wc;
  int rows = min(c.n, g.wh - kr * c.n);
  const bool valid = (kr < c.sr) && (rows > 0) && (wc <= g.ww) && (gc >= 0) && (gc < W);
  if (!valid) rows = -2;
  const bool colin = valid && (wc >= 0) && (wc < g.ww);
  const bool colown = colin && (l >= 1) && (l <= C - 2);
  const int wr0 = kr * c.n - 1;                           // window row of iteration 0
  const int gr0 = g.r0 + wr0;
  L.rows = rows;
  L.flags = (valid ? kFValid : 0u) | (colin ? kFColIn : 0u) | (colown ? kFColOwn : 0u) | ((valid && kr > 0) ? kFTopIn : 0u) |
            ((valid && kr * c.n + rows < g.wh) ? kFBelowIn : 0u) | ((gc == 0) ? kFAtLeft : 0u) | ((gc == W - 1) ? kFAtRight : 0u);
  L.lo = max(0, -gr0);
  L.hi = valid ? min(rows + 1, H - 1 - gr0) : -1;
  L.jtop = -gr0; L.jbot = H - 1 - gr0;
  L.cell0 = (uint32_t)(gr0 * W + gc);
  L.fidx = (g.mr0 + wr0) * g.bw + g.mc0 + wc;
  L.tidx = (g.mr0 + wr0 + 1) * (g.bw + 2) + g.mc0 + wc + 1;
  return L;
}
__device__ __forceinline__ bool has(const Lane& L, const uint32_t f) { return (L.flags & f) != 0u; }

// Row predicates are one 32-bit compare each.  Left alone, the compiler computes every one of them once per step and keeps
// the lane masks (two scalar registers each, several per row) alive from the loads to the commit -- hundreds of spilled
// SGPRs.  Every phase therefore works on a copy of the lane record whose contents it cannot trace back.
__device__ __forceinline__ Lane fresh(Lane L) {
  asm volatile("" : "+v"(L.rows), "+v"(L.flags), "+v"(L.lo), "+v"(L.hi));
  return L;
}
// The same for the grid width: jj * W * bytes-per-cell are invariants of the whole launch, and the compiler would keep all
// ~50 of them in scalar registers from the first step to the last.
__device__ __forceinline__ int fresh_s(int W) { asm volatile("" : "+s"(W)); return W; }

template <typename TS> struct RowIO;
template <> struct RowIO<double> {
  static __device__ __forceinline__ double load(rsrc_t r, uint32_t off, uint32_t soff) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, (int)soff, 2));
  }
  static __device__ __forceinline__ void store(rsrc_t r, uint32_t off, uint32_t soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2i32, v), r, (int)off, (int)soff, 2);
  }
};
template <> struct RowIO<float> {
  static __device__ __forceinline__ double load(rsrc_t r, uint32_t off, uint32_t soff) {
    return (double)__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, (int)off, (int)soff, 2));
  }
  static __device__ __forceinline__ void store(rsrc_t r, uint32_t off, uint32_t soff, double v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, (float)v), r, (int)off, (int)soff, 2);
  }
};
__device__ __forceinline__ double ld_f64_s(rsrc_t r, uint32_t off, uint32_t soff) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, (int)off, (int)soff, 0));
}

// Byte offset of the lane's cell of iteration jj in a plane of `cb` bytes per cell, as (per-lane part, uniform part).
// Interior windows: the lane's part is fixed for the step and the row advances in the scalar offset of the instruction.
// Border windows: cell0 may lie one row above the grid (a negative, wrapped index), so the row term is added per lane and no
// offset is negative when it is used.
template <bool INTERIOR>
__device__ __forceinline__ void row_offsets(const Lane& L, const int jj, const int W, const uint32_t cb, uint32_t& off, uint32_t& soff) {
  if (INTERIOR) { off = L.cell0 * cb; soff = (uint32_t)(jj * W) * cb; }
  else { off = (L.cell0 + (uint32_t)(jj * W)) * cb; soff = 0u; }
}
// border windows: the row of iteration jj exists in the grid and the lane needs it
__device__ __forceinline__ bool row_exists(const Lane& L, const int jj) { return jj >= L.lo && jj <= L.hi; }
__device__ __forceinline__ bool row_own(const Lane& L, const int jj) { return (jj >= 1) && (jj <= kNR) && (jj <= L.rows); }
// phase A writes the tile cell of iteration jj: the lane's own cells (candidate bed) and the cells of the halo ring around
// the window (bed); cells inside the window that belong to another strip are that strip's to write.  w0, w1, w2: the
// answer for iteration 0, for the own rows, for iteration rows + 1 (lane masks, fixed for the step).
struct WriteMasks { bool w0, w1, w2; };
__device__ __forceinline__ WriteMasks write_masks(const Lane& L) {
  WriteMasks m;
  const bool valid = has(L, kFValid), colin = has(L, kFColIn);
  m.w0 = valid && !(has(L, kFTopIn) && colin);
  m.w1 = valid && (has(L, kFColOwn) || !colin);
  m.w2 = valid && !(has(L, kFBelowIn) && colin);
  return m;
}
template <bool INTERIOR>
__device__ __forceinline__ bool cell_written(const Lane& L, const WriteMasks& m, const int jj) {
  bool w = (jj == 0) ? m.w0 : ((jj <= L.rows && m.w1) || (jj - 1 == L.rows && m.w2));
  if (!INTERIOR) w = w && row_exists(L, jj);
  return w;
}

// ---- phase A, part 1: every load of the chain state of the step, issued back to back --------------------------------
// vb[jj] = bed of the cell of iteration jj where the lane writes it to the tile, ve[R - 1] = carried energy of the own
// cell of row R, a2[R - 1] = (wupd, surf) of the own cells of rows 1 .. kNA (a ring: part 2 requests row R + kNA when it is
// done with row R).  0 where the lane has no such cell.
constexpr int kNA = 4;
// PARTS: 1 = the bed, 2 = carried energy and (wupd, surf), 3 = both, row by row
template <typename TS, bool INTERIOR, int PARTS = 3>
__device__ __forceinline__ void load_state(const Lane& L_in, const int n, const int W_in, const rsrc_t r_bed, const rsrc_t r_en, const rsrc_t r_st,
                                           double (&vb)[kNR + 2], double (&ve)[kNR], double2 (&a2)[kNA]) {
  const Lane L = fresh(L_in);
  const int W = fresh_s(W_in);
  const WriteMasks wm = write_masks(L);
  const bool colown = has(L, kFColOwn);
  // Row by row, in the order phase A consumes them (loads return in the order they were issued).  The out-of-range offset lives
  // in a register (a literal is re-materialised before every select), the own-column offsets are selected once, and a row's
  // three loads share its compares: 5 vector instructions per row for three loads.
  uint32_t oob = kOOB;
  asm volatile("" : "+v"(oob));
  const uint32_t ownS = colown ? L.cell0 * (uint32_t)sizeof(TS) : oob, own16 = colown ? L.cell0 * 16u : oob;
#pragma unroll
  for (int jj = 0; jj < kNR + 2; ++jj) {
    if (PARTS & 1) vb[jj] = 0.0;
    if ((PARTS & 2) && jj >= 1 && jj <= kNR) ve[jj - 1] = 0.0;
    if ((PARTS & 2) && jj >= 1 && jj <= kNA) a2[jj - 1] = make_double2(0.0, 0.0);
    if (jj <= n + 1) {
      uint32_t off, soff, off16, soff16;
      row_offsets<INTERIOR>(L, jj, W, (uint32_t)sizeof(TS), off, soff);
      row_offsets<INTERIOR>(L, jj, W, 16u, off16, soff16);
      if (PARTS & 1) vb[jj] = RowIO<TS>::load(r_bed, cell_written<INTERIOR>(L, wm, jj) ? off : oob, soff);
      if ((PARTS & 2) && jj >= 1 && jj <= kNR) {
        const bool ro = row_own(L, jj);
        if (jj <= kNA) a2[jj - 1] = ld_f64x2(r_st, ro ? (INTERIOR ? own16 : (colown ? off16 : oob)) : oob, soff16);
        ve[jj - 1] = RowIO<TS>::load(r_en, ro ? (INTERIOR ? ownS : (colown ? off : oob)) : oob, soff);
      }
    }
  }
}

// ---- phase A, part 2: candidate bed -> tile, guard, sum of the carried energy -------------------------------------------------
//   field(jj, in): the proposal value at the lane's cell of iteration jj (0 where !in); read before the tile cell is rewritten
//   out: upd_bits: bit R set where the lane's own cell of row R takes the update; acc_old; guard
template <typename TS, bool INTERIOR, class Field>
__device__ __forceinline__ void phase_a(const Lane& L_in, const int n, const int W_in, const int bw, const rsrc_t r_st, Field field,
                                        double* __restrict__ tile, double (&vb)[kNR + 2], double (&ve)[kNR], double2 (&a2)[kNA],
                                        uint32_t& upd_bits, double& acc_old, bool& guard) {
  constexpr bool F32 = sizeof(TS) == 4;
  const Lane L = fresh(L_in);
  const int W = fresh_s(W_in);
  const WriteMasks wm = write_masks(L);
  const bool colown = has(L, kFColOwn);
  uint32_t oob = kOOB;
  asm volatile("" : "+v"(oob));
  const uint32_t own16 = colown ? L.cell0 * 16u : oob;
  guard = false;
  upd_bits = 0u;
  const int ts = bw + 2;
#pragma unroll
  for (int jj = 0; jj < kNR + 2; ++jj) {
    if (jj <= n + 1) {
      double v = vb[jj];
      if (jj >= 1 && jj <= kNR) {
        const bool own = row_own(L, jj) && colown;
        const double2 A2 = a2[(jj - 1) % kNA];
        if (jj + kNA <= kNR && jj + kNA <= n) {             // (wupd, surf) of row jj + kNA into the slot just read
          uint32_t off, soff;
          row_offsets<INTERIOR>(L, jj + kNA, W, 16u, off, soff);
          a2[(jj - 1) % kNA] = ld_f64x2(r_st, row_own(L, jj + kNA) ? (INTERIOR ? own16 : (colown ? off : oob)) : oob, soff);
        }
        const double f = field(jj, own);
        const bool upd = own && (__builtin_bit_cast(uint64_t, A2.x) != kNoUpdBits);
        if (upd) {
          v = v + f * A2.x;
          if (F32) v = (double)(float)v;
        }
        const double thick = A2.y - v;
        guard = guard || (upd && thick <= 0.0);
        upd_bits |= upd ? (1u << jj) : 0u;
      }
      if (cell_written<INTERIOR>(L, wm, jj)) tile[L.tidx + jj * ts] = v;
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  acc_old = 0.0;
#pragma unroll
  for (int R = 1; R <= kNR; ++R) if (R <= n) acc_old += ve[R - 1];
}

template <int CTRL>
__device__ __forceinline__ double wave_shift(double x) {
  const v2i32 b = __builtin_bit_cast(v2i32, x);
  v2i32 o;
  o.x = __builtin_amdgcn_update_dpp(0, b.x, CTRL, 0xF, 0xF, true);
  o.y = __builtin_amdgcn_update_dpp(0, b.y, CTRL, 0xF, 0xF, true);
  return __builtin_bit_cast(double, o);
}
constexpr int kWaveShr1 = 0x138;   // lane i <- lane i - 1
constexpr int kWaveShl1 = 0x130;   // lane i <- lane i + 1

struct StepConsts { double res, rcp_res, two_res, rcp_two_res; };

// ---- phase D: fluxes from the tile, residual stencil (Topography.py:592-600, np.gradient's one-sided differences at the
// grid border), new energies -> en[R - 1] of the lane's own rows; acc_new = their sum.  Every lane of the wave must be here
// (DPP).  The operands of row jj + 1 are requested before row jj is worked on.  Interior windows: every lane with a column
// reads every row up to n + 1 -- rows past its own halo row give fluxes nobody uses -- so that no load carries a per-row
// predicate: one register of per-lane offset, fixed for the step, and the row in the instruction's scalar offset.
struct RowOps { double v, surf; double2 B2, C2; };
template <typename TS, bool FAST_DIV, bool INTERIOR>
struct PhaseD {
  const Lane& L; const int n, W, ts; const rsrc_t r_st; const uint32_t off_sB; const StepConsts& K; const double* __restrict__ tile;
  const uint32_t off_v, off_o;      // interior: 16 * cell0 where the lane has a column / an own column, else out of range
  const bool colown;
  double (&en)[kNR]; double acc_new;
  double qy_m2, qy_m1, qx_m1; double2 C_m1; bool own_m1;

  __device__ __forceinline__ RowOps request(const int jj) const {
    RowOps o;
    uint32_t ov, oo, soff;
    if (INTERIOR) { ov = off_v; oo = off_o; soff = (uint32_t)(jj * W) * 16u; }
    else {
      const uint32_t off = (L.cell0 + (uint32_t)(jj * W)) * 16u;
      const bool need = row_exists(L, jj);
      ov = need ? off : kOOB; oo = (need && colown) ? off : kOOB; soff = 0u;
    }
    o.v = tile[L.tidx + jj * ts];
    o.surf = ld_f64_s(r_st, ov + 8u, soff);
    o.B2 = ld_f64x2(r_st, ov, soff + off_sB);
    o.C2 = make_double2(0.0, 0.0);
    if (jj >= 1 && jj <= kNR) o.C2 = ld_f64x2(r_st, oo, soff + 2u * off_sB);
    return o;
  }
  template <int JJ>
  __device__ __forceinline__ void row(const RowOps cur) {
    constexpr bool F32 = sizeof(TS) == 4;
    if (JJ <= n + 1) {        // uniform; nested, not a sequence: the rolling values need no copies on the way out
      RowOps nxt;
      if constexpr (JJ + 1 <= kNR + 1) nxt = request(JJ + 1);
      const double thick = cur.surf - cur.v;
      const double qx = cur.B2.x * thick, qy = cur.B2.y * thick;
      if (JJ >= 2) {
        // residual of the row of iteration R = JJ - 1: its x neighbours from the lanes beside it, its y neighbours from the
        // iterations before and after it
        constexpr int R = (JJ >= 2) ? JJ - 1 : 1;
        double qxl = wave_shift<kWaveShr1>(qx_m1), qxr = wave_shift<kWaveShl1>(qx_m1);
        double qya = qy_m2, qyb = qy;
        double dx, dy;
        if (INTERIOR) {
          const double ddx = qxr - qxl, ddy = qyb - qya;
          if (FAST_DIV) { dx = exact_div(ddx, K.two_res, K.rcp_two_res); dy = exact_div(ddy, K.two_res, K.rcp_two_res); }
          else { dx = ddx / K.two_res; dy = ddy / K.two_res; }
        } else {
          const bool atleft = has(L, kFAtLeft), atright = has(L, kFAtRight);
          const bool xedge = atleft || atright;
          const bool attop = (R == L.jtop), atbot = (R == L.jbot);
          if (atleft) qxl = qx_m1;
          if (atright) qxr = qx_m1;
          if (attop) qya = qy_m1;
          if (atbot) qyb = qy_m1;
          const double ddx = qxr - qxl, ddy = qyb - qya;
          if (FAST_DIV) {
            dx = xedge ? exact_div(ddx, K.res, K.rcp_res) : exact_div(ddx, K.two_res, K.rcp_two_res);
            dy = (attop || atbot) ? exact_div(ddy, K.res, K.rcp_res) : exact_div(ddy, K.two_res, K.rcp_two_res);
          } else {
            dx = ddx / (xedge ? K.res : K.two_res);
            dy = ddy / ((attop || atbot) ? K.res : K.two_res);
          }
        }
        const double r = ((dx + dy) + C_m1.x) - C_m1.y;
        double e = 0.0;
        if (own_m1 && !isnan(r)) e = r * r;
        if (F32) e = (double)(float)e;
        en[R - 1] = e;
        acc_new += e;
      }
      qy_m2 = qy_m1; qy_m1 = qy; qx_m1 = qx; C_m1 = cur.C2; own_m1 = row_own(L, JJ) && colown;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (JJ + 1 <= kNR + 1) row<JJ + 1>(nxt);
    }
  }
};

template <typename TS, bool FAST_DIV, bool INTERIOR>
__device__ __forceinline__ void phase_d(const Lane& L_in, const int n, const int W_in, const int bw, const rsrc_t r_st, const uint32_t off_sB,
                                        const StepConsts& K, const double* __restrict__ tile, double (&en)[kNR], double& acc_new) {
  const Lane L = fresh(L_in);
  const int W = fresh_s(W_in);
#pragma unroll
  for (int R = 0; R < kNR; ++R) en[R] = 0.0;
  const bool colown = has(L, kFColOwn);
  PhaseD<TS, FAST_DIV, INTERIOR> P{L, n, W, bw + 2, r_st, off_sB, K, tile,
                                   has(L, kFValid) ? L.cell0 * 16u : kOOB, colown ? L.cell0 * 16u : kOOB, colown,
                                   en, 0.0, 0.0, 0.0, 0.0, make_double2(0.0, 0.0), false};
  P.template row<0>(P.request(0));
  acc_new = P.acc_new;
}

// candidate bed of the lane's own cells, back from the tile (before the reduction's barrier: once a wave is past it, the next
// step's proposal may overwrite the tile).  Read for every row; only the cells that take the update are stored.
__device__ __forceinline__ void read_candidate(const Lane& L, const int n, const int bw, const double* __restrict__ tile, double (&vn)[kNR]) {
  const int ts = bw + 2;
#pragma unroll
  for (int R = 1; R <= kNR; ++R) {
    vn[R - 1] = 0.0;
    if (R <= n) vn[R - 1] = tile[L.tidx + R * ts];
  }
}

// accepted step: candidate bed (where the update mask is set) and new energies of the strip's own cells -> HBM
// (MCMC.py:1338-1347).  RS: also bump resampled_times with a no-return atomic (replay kernel; the fused kernel counts
// them afterwards from its records).
template <typename TS, bool INTERIOR, bool RS>
__device__ __forceinline__ void commit(const Lane& L_in, const int n, const int W_in, const rsrc_t r_bed, const rsrc_t r_en, const rsrc_t r_rs,
                                       const double (&vn)[kNR], const double (&en)[kNR], const uint32_t upd_bits) {
  const Lane L = fresh(L_in);
  const int W = fresh_s(W_in);
  const bool colown = has(L, kFColOwn);
  uint32_t oob = kOOB;
  asm volatile("" : "+v"(oob));
  const uint32_t ownS = colown ? L.cell0 * (uint32_t)sizeof(TS) : oob;
#pragma unroll
  for (int R = 1; R <= kNR; ++R) {
    if (R > n) continue;
    uint32_t off, soff;
    row_offsets<INTERIOR>(L, R, W, (uint32_t)sizeof(TS), off, soff);
    const bool upd = (upd_bits >> R) & 1u;
    RowIO<TS>::store(r_en, row_own(L, R) ? (INTERIOR ? ownS : (colown ? off : oob)) : oob, soff, en[R - 1]);
    RowIO<TS>::store(r_bed, upd ? off : oob, soff, vn[R - 1]);
    if (RS) {
      uint32_t off4, soff4;
      row_offsets<INTERIOR>(L, R, W, 4u, off4, soff4);
      __builtin_amdgcn_raw_ptr_buffer_atomic_add_i32(1, r_rs, (int)(upd ? off4 : kOOB), (int)soff4, 0);
    }
  }
}

// sum over the 8 waves' partials (red[0 .. 7], red[8 .. 15] = 0), the same value in every lane
__device__ __forceinline__ double waves_sum(const double* red, const int lane) { return row16_sum(red[lane & 15]); }

}  // namespace strip
}  // namespace gsm
