// Small-scale chain on gfx950: sequential Gaussian simulation of one block per chain, and the chain's full-grid loss.
//
// Replaces, for a batch of chains (one 64-lane workgroup = one wavefront per chain):
//   sgs                      gstatsMCMC/MCMC.py:91-173   (cell loop :135-168)
//   neighbors (octant search) gstatsMCMC/gstatsim_custom/neighbors.py:4-64
//   ok_solve                 gstatsMCMC/gstatsim_custom/_krige.py:5-44
//   chain.loss on the proposed bed + thickness guard  MCMC.py:1781-1795 (loss :1021-1044, Topography.py:592-600)
//
// The random numbers are the caller's (replay: NumPy's PCG64 on the host, in the reference's order): the order in which
// the block's cells are visited (rng.shuffle, MCMC.py:128) and one standard normal per simulated cell
// (rng.normal(est, sqrt(var)) = est + sqrt(var) * z, MCMC.py:165).
//
// Per cell, in the given order, if the cell is not conditioned yet:
//   1 octant search: every grid cell within +-hw of the cell that holds a value (everything outside the block does: the
//     current bed; inside the block the conditioning data and the cells simulated so far) and lies closer than `radius` is
//     put into one of eight 45-degree sectors, (b pi/4, (b+1) pi/4], b = -4..3, of atan2(y0 - y, x0 - x).  The sector is
//     decided from the signs and magnitudes of the two coordinate differences -- exactly what numpy.arctan2 gives on the
//     sector boundaries (multiples of pi/4 are hit only by cells on the axes and diagonals, where arctan2 returns the
//     boundary value bit for bit; checked on the host).  Per sector the num_points / 8 nearest, ties by window position
//     (NumPy's masked C order followed by a stable sort; numpy.argsort's default sort is stable below 17 elements).
//   2 ordinary kriging: (n+1) x (n+1) system [Sigma 1; 1^T 0] w = [rho; 1], covariances from a table indexed by the integer
//     lag between two cells (the host evaluates the reference's covariance model -- scipy's Bessel K for Matern -- once per
//     lag), solved in LDS by Gauss-Jordan elimination on the diagonal (positive-definite covariance block) in fp64 (the reference calls numpy.linalg.lstsq,
//     an SVD solve: same solution for these non-singular systems, different rounding -- the stated tolerance of this path).
//   3 value = est + sqrt(|var|) * z; the cell becomes conditioning data for the cells after it.
// Limits: hw <= 16 (search window 33 x 33), num_points <= 48, window (block) cells <= 1024.  A cell without any neighbour
// inside `radius` would make the reference grow the radius by 100 km (MCMC.py:152-156): not built, reported as an error.
#include "gsm_internal.h"
#include "device_util.h"
#include "residual_device.h"
#include "normal_score.h"
#include "proposal_device.h"
#include <math.h>

namespace gsm {

constexpr int kSgsMaxHw = 16;
constexpr int kSgsMaxCand = (2 * kSgsMaxHw + 1) * (2 * kSgsMaxHw + 1);   // 1089
constexpr int kSgsMaxPts = 48;
constexpr int kSgsStride = kSgsMaxPts + 3;                                // row stride of the augmented matrix
constexpr int kSgsMaxWin = 1024;
constexpr int kSgsCandPerLane = (kSgsMaxCand + 63) / 64;                  // 18

// Wave-wide minima on the DPP network (row steps, two row broadcasts, v_readlane of lane 63): a minimum does not depend on
// the order of its operands, and ds_bpermute shuffles (__shfl_xor) cost an LDS round trip per level -- with 66 minima per
// simulated cell (neighbour selection, pivot search) they were two thirds of the kernel's time.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_keep_i32(int x) { return __builtin_amdgcn_update_dpp(x, x, CTRL, ROW_MASK, 0xF, false); }
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_keep_f64(double x) {
  const dev::v2i32 b = __builtin_bit_cast(dev::v2i32, x);
  dev::v2i32 o;
  o.x = __builtin_amdgcn_update_dpp(b.x, b.x, CTRL, ROW_MASK, 0xF, false);
  o.y = __builtin_amdgcn_update_dpp(b.y, b.y, CTRL, ROW_MASK, 0xF, false);
  return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ double wave_min_f64(double v) {
  v = fmin(v, dpp_keep_f64<0xB1, 0xF>(v));     // quad_perm [1,0,3,2]
  v = fmin(v, dpp_keep_f64<0x4E, 0xF>(v));     // quad_perm [2,3,0,1]
  v = fmin(v, dpp_keep_f64<0x141, 0xF>(v));    // row_half_mirror
  v = fmin(v, dpp_keep_f64<0x140, 0xF>(v));    // row_mirror
  v = fmin(v, dpp_keep_f64<0x142, 0xA>(v));    // row_bcast:15 into rows 1 and 3
  v = fmin(v, dpp_keep_f64<0x143, 0xC>(v));    // row_bcast:31 into rows 2 and 3
  const dev::v2i32 b = __builtin_bit_cast(dev::v2i32, v);
  dev::v2i32 o;
  o.x = __builtin_amdgcn_readlane(b.x, 63);
  o.y = __builtin_amdgcn_readlane(b.y, 63);
  return __builtin_bit_cast(double, o);
}
__device__ __forceinline__ int wave_min_i32(int v) {
  v = min(v, dpp_keep_i32<0xB1, 0xF>(v));
  v = min(v, dpp_keep_i32<0x4E, 0xF>(v));
  v = min(v, dpp_keep_i32<0x141, 0xF>(v));
  v = min(v, dpp_keep_i32<0x140, 0xF>(v));
  v = min(v, dpp_keep_i32<0x142, 0xA>(v));
  v = min(v, dpp_keep_i32<0x143, 0xC>(v));
  return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ double wave_sum_f64(double v) { return dev::wave64_sum(v); }     // DPP tree, wave-uniform result

// sector b + 4 in 0..7 of atan2(dy, dx) in (b pi/4, (b+1) pi/4]
__device__ __forceinline__ int octant(double dy, double dx) {
  if (dy == 0.0) return (dx < 0.0) ? 7 : 3;                    // angle pi -> b = 3; angle 0 -> b = -1
  const double ay = fabs(dy), ax = fabs(dx);
  if (dy > 0.0) {
    if (dx > 0.0) return (ay <= ax) ? 4 : 5;                   // (0, pi/4] | (pi/4, pi/2)
    if (dx == 0.0) return 5;                                   // pi/2
    return (ay >= ax) ? 6 : 7;                                 // (pi/2, 3pi/4] | (3pi/4, pi)
  }
  if (dx > 0.0) return (ay < ax) ? 3 : 2;                      // (-pi/4, 0) | (-pi/2, -pi/4]
  if (dx == 0.0) return 1;                                     // -pi/2 -> (-3pi/4, -pi/2]
  return (ay > ax) ? 1 : 0;                                    // (-3pi/4, -pi/2) | (-pi, -3pi/4]
}

// One 64-lane workgroup per chain; the block's cells are simulated one after the other.  Everything a cell needs is read
// from LDS: the part of the grid its search windows can reach (block + hw cells on every side, staged once; simulated values
// are written into it as the loop goes), the lag covariance table, the candidates' values.  `staged` = 0 (a region beyond
// kSgsRegionMax cells, or a lag table beyond kSgsLagMax): the same code reads the grid / the table from global memory.
constexpr int kSgsRegionMax = 6144;                       // doubles: e.g. an 8 x 8 block with hw = 16 needs 40 x 40
constexpr int kSgsLagMax = (4 * kSgsMaxHw + 1) * (4 * kSgsMaxHw + 1);   // 4225
struct SgsLds {
  double* cand_d; double* cand_v; int8_t* cand_s; double* A; double* nb_val; int* nb_i; int* nb_j; double* region; double* lag;
};
static size_t sgs_lds_bytes() {
  return sizeof(double) * (2 * (size_t)kSgsMaxCand + (size_t)(kSgsMaxPts + 1) * kSgsStride + kSgsMaxPts + kSgsRegionMax + kSgsLagMax) +
         sizeof(int) * 2 * kSgsMaxPts + ((kSgsMaxCand + 7) & ~7);
}

__global__ __launch_bounds__(64) void sgs_blocks_kernel(const SgsArgs a) {
  extern __shared__ double sgs_lds[];
  SgsLds L;
  L.cand_d = sgs_lds; L.cand_v = L.cand_d + kSgsMaxCand; L.A = L.cand_v + kSgsMaxCand;
  L.nb_val = L.A + (kSgsMaxPts + 1) * kSgsStride; L.region = L.nb_val + kSgsMaxPts; L.lag = L.region + kSgsRegionMax;
  L.nb_i = (int*)(L.lag + kSgsLagMax); L.nb_j = L.nb_i + kSgsMaxPts; L.cand_s = (int8_t*)(L.nb_j + kSgsMaxPts);
  double* const cand_d = L.cand_d; double* const cand_v = L.cand_v; int8_t* const cand_s = L.cand_s; double* const A = L.A;
  double* const nb_val = L.nb_val; int* const nb_i = L.nb_i; int* const nb_j = L.nb_j;
  const int chain = blockIdx.x, lane = threadIdx.x;
  const int H = a.H, W = a.W;
  double* __restrict__ g = a.grid + (size_t)chain * H * W;
  const int r0 = a.win[4 * chain], r1 = a.win[4 * chain + 1], c0 = a.win[4 * chain + 2], c1 = a.win[4 * chain + 3];
  const int wh = r1 - r0, ww = c1 - c0;
  if (r0 < 0 || c0 < 0 || r1 > H || c1 > W || wh < 0 || ww < 0 || wh * ww > kSgsMaxWin) {
    if (lane == 0) atomicOr(a.err, 1);
    return;
  }
  const int hw = a.hw, k8 = a.num_points / 8;
  const int side = 2 * hw + 1, m = a.m, lag_side = 2 * m + 1;
  const uint32_t side_magic = 65536u / (uint32_t)side + 1u;      // p / side for p < 1985 (side <= 33) as a multiply
  // region of the grid the search windows of the block's cells can reach
  const int ri0 = max(0, r0 - hw), ri1 = min(H, r1 + hw), rj0 = max(0, c0 - hw), rj1 = min(W, c1 + hw);
  const int rh = ri1 - ri0, rw = rj1 - rj0;
  const bool staged = rh * rw <= kSgsRegionMax;
  // block cells: conditioning data (or NaN) now, simulated values as the loop goes; in LDS until the end
  double* const overlay = staged ? L.region : L.region;         // unstaged: the first wh * ww doubles hold the block only
  if (staged) {
    for (int p = lane; p < rh * rw; p += 64) {
      const int i = ri0 + p / rw, j = rj0 + p % rw;
      const int gi = i * W + j;
      const bool in_block = i >= r0 && i < r1 && j >= c0 && j < c1;
      L.region[p] = (in_block && a.zcond) ? a.zcond[gi] : g[gi];
    }
  } else {
    for (int p = lane; p < wh * ww; p += 64) {
      const int gi = (r0 + p / ww) * W + c0 + p % ww;
      overlay[p] = a.zcond ? a.zcond[gi] : g[gi];
    }
  }
  const bool lag_lds = lag_side * lag_side <= kSgsLagMax;
  if (lag_lds) for (int p = lane; p < lag_side * lag_side; p += 64) L.lag[p] = a.lag[p];
  const double* __restrict__ lag = lag_lds ? L.lag : a.lag;
  __syncthreads();
  auto block_index = [&](int i, int j) { return staged ? (i - ri0) * rw + (j - rj0) : (i - r0) * ww + (j - c0); };
  auto value_at = [&](int i, int j) -> double {
    if (staged) return L.region[(i - ri0) * rw + (j - rj0)];
    if (i >= r0 && i < r1 && j >= c0 && j < c1) return overlay[(i - r0) * ww + (j - c0)];
    return g[i * W + j];
  };
  auto lag_cov = [&](int di, int dj) { return lag[(di + m) * lag_side + dj + m]; };
  const int k_lo = a.cell_off[chain], k_hi = a.cell_cnt ? k_lo + a.cell_cnt[chain] : a.cell_off[chain + 1];
  for (int k = k_lo; k < k_hi; ++k) {
    const int i0 = a.cells[2 * k], j0 = a.cells[2 * k + 1];
    if (i0 < r0 || i0 >= r1 || j0 < c0 || j0 >= c1) { if (lane == 0) atomicOr(a.err, 2); continue; }
    const int op = block_index(i0, j0);
    if (!isnan(overlay[op])) {                       // conditioned already: nothing drawn for it (MCMC.py:141)
      if (a.trace && lane == 0) { a.trace[3 * k] = -1.0; a.trace[3 * k + 1] = overlay[op]; a.trace[3 * k + 2] = 0.0; }
      continue;
    }
    // ---- 1 candidates of the search window -> (distance, sector) in registers, value in LDS -----------
    // candidate p = lane + 64 t of the (2 hw + 1)^2 window, t < kSgsCandPerLane: the loops over t are unrolled, so the
    // two arrays live in registers and the 2 * num_points selection rounds below never touch LDS
    const double x0 = a.xs[j0], y0 = a.ys[i0];
    double cd[kSgsCandPerLane];
    int cs[kSgsCandPerLane];
#pragma unroll
    for (int t = 0; t < kSgsCandPerLane; ++t) {
      const int p = lane + 64 * t;
      int sec = -1;
      double d = 0.0;
      if (p < side * side) {
        const int io = (int)(((uint32_t)p * side_magic) >> 16);
        const int i = i0 - hw + io, j = j0 - hw + (p - io * side);
        double v = 0.0;
        if (i >= 0 && i < H && j >= 0 && j < W) {
          v = value_at(i, j);
          if (!isnan(v)) {
            const double dx = x0 - a.xs[j], dy = y0 - a.ys[i];
            d = sqrt(dx * dx + dy * dy);
            if (d < a.radius) sec = octant(dy, dx);
          }
        }
        cand_v[p] = v;
      }
      cd[t] = d; cs[t] = sec;
    }
    __syncthreads();
    // per sector, in ascending (distance, window position): extract up to k8 points
    int n = 0;
    for (int s = 0; s < 8; ++s) {
      double prev_d = -1.0;
      int prev_p = -1;
      for (int r = 0; r < k8; ++r) {
        double best_d = INFINITY;
        int best_p = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < kSgsCandPerLane; ++t) {
          if (64 * t >= side * side) break;                    // wave-uniform: no candidate in this slot for any lane
          const int p = lane + 64 * t;
          const double d = cd[t];
          const bool open = (cs[t] == s) && !(d < prev_d || (d == prev_d && p <= prev_p));    // not taken in an earlier round
          if (open && (d < best_d || (d == best_d && p < best_p))) { best_d = d; best_p = p; }
        }
        const double wd = wave_min_f64(best_d);
        if (wd == INFINITY) break;                                          // sector exhausted
        // the smallest window position among the lanes that hold the minimum: usually one lane (then its position is read
        // with one v_readlane), else a second reduction
        const unsigned long long tied = __ballot(best_d == wd);
        const int wp = (__popcll(tied) == 1) ? __builtin_amdgcn_readlane(best_p, __ffsll((long long)tied) - 1)
                                             : wave_min_i32((best_d == wd) ? best_p : 0x7fffffff);
        if (lane == 0) {
          const int io = (int)(((uint32_t)wp * side_magic) >> 16);
          nb_i[n] = i0 - hw + io; nb_j[n] = j0 - hw + (wp - io * side); nb_val[n] = cand_v[wp];
        }
        prev_d = wd; prev_p = wp;
        ++n;
      }
    }
    __syncthreads();
    if (n == 0) {
      if (lane == 0) { atomicOr(a.err, 4); overlay[op] = NAN; }
      continue;
    }
    // ---- 2 ordinary kriging system, Gauss-Jordan elimination on the diagonal ----------------------------
    const int N = n + 1;
    for (int e = lane; e < N * (N + 1); e += 64) {
      const int ra = e / (N + 1), cb = e % (N + 1);
      double v;
      if (cb == N) v = (ra < n) ? lag_cov(nb_i[ra] - i0, nb_j[ra] - j0) : 1.0;      // rho | 1
      else if (ra < n && cb < n) v = lag_cov(nb_i[ra] - nb_i[cb], nb_j[ra] - nb_j[cb]);
      else v = (ra == n && cb == n) ? 0.0 : 1.0;
      A[ra * kSgsStride + cb] = v;
    }
    __syncthreads();
    const double rho_l = (lane < n) ? A[lane * kSgsStride + N] : 0.0;            // kept: the elimination overwrites it
    bool singular = false;
    for (int kk = 0; kk < N; ++kk) {
      // pivot = the diagonal entry: the leading n x n block is a covariance matrix (symmetric positive definite), whose
      // elimination needs no row exchanges, and the last pivot is -1' Sigma^-1 1 < 0
      const double pv = A[kk * kSgsStride + kk];
      if (!(fabs(pv) > 0.0)) { singular = true; break; }
      // row update: lane = (row slot, column slot) of a 16 x 4 arrangement; a lane takes every fourth column (beyond kk: column
      // kk itself is not read again) of its rows: a - f * b per entry, f = A[row][kk] / pivot.  One barrier per step.
      for (int row = lane >> 2; row < N; row += 16) {
        if (row == kk) continue;
        const double f = A[row * kSgsStride + kk] / pv;
        for (int c = kk + 1 + (lane & 3); c <= N; c += 4) A[row * kSgsStride + c] -= f * A[kk * kSgsStride + c];
      }
      __syncthreads();
    }
    if (singular) {
      if (lane == 0) { atomicOr(a.err, 8); overlay[op] = NAN; }
      __syncthreads();
      continue;
    }
    const double w_l = (lane < n) ? A[lane * kSgsStride + N] / A[lane * kSgsStride + lane] : 0.0;
    const double v_l = (lane < n) ? nb_val[lane] : 0.0;
    const double local_mean = wave_sum_f64(v_l) / (double)n;
    const double est = local_mean + wave_sum_f64((lane < n) ? w_l * (v_l - local_mean) : 0.0);
    double var = a.sill - wave_sum_f64(w_l * rho_l);
    var = fabs(var);
    if (lane == 0) {
      overlay[op] = est + sqrt(var) * a.z[k];
      if (a.trace) { a.trace[3 * k] = (double)n; a.trace[3 * k + 1] = est; a.trace[3 * k + 2] = var; }
    }
    __syncthreads();
  }
  for (int p = lane; p < wh * ww; p += 64) {
    const int i = r0 + p / ww, j = c0 + p % ww;
    g[i * W + j] = overlay[block_index(i, j)];
  }
}

hipError_t launch_sgs_blocks(const SgsArgs& a, hipStream_t st) {
  if (a.hw < 1 || a.hw > kSgsMaxHw || a.num_points < 8 || a.num_points > kSgsMaxPts) return hipErrorInvalidValue;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)sgs_blocks_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  hipLaunchKernelGGL(sgs_blocks_kernel, dim3(a.n_chains), dim3(64), sgs_lds_bytes(), st, a);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------------------------
// loss of the proposed bed (+ trend) over the whole grid and the thickness guard, one 256-thread workgroup per chain:
//   loss = nansum(residual^2 where mc_mask == 1) / (2 sigma^2)      (MCMC.py:1021-1044 on Topography.py:592-600)
//   bad  = number of cells with guard_mask == 1 and surf - (bed + trend) <= 0   (MCMC.py:1789-1795)
// S.upd is the guard mask here (grounded_ice_mask); trend may be NULL.
// ---------------------------------------------------------------------------------------------------------------------
// Several workgroups per chain (a chain's grid is split into `parts` contiguous ranges of cells), then one thread per chain
// adds the parts in order: with one workgroup per chain a 256 x 256 grid kept 16 CUs busy for 0.19 ms per iteration.
__global__ __launch_bounds__(256) void sgs_loss_kernel(const StaticFields S, const double* beds, const double* trend, int parts,
                                                       double* part_sum, int32_t* part_bad) {
  __shared__ double red[8];
  __shared__ int redb[4];
  const int chain = blockIdx.x, part = blockIdx.y, tid = threadIdx.x;
  const int plane = S.H * S.W;
  const int per = (plane + parts - 1) / parts, g_lo = part * per, g_hi = min(plane, g_lo + per);
  const double* bed = beds + (size_t)chain * plane;
  auto bed_at = [&](int rr, int cc) { const int q = rr * S.W + cc; return trend ? bed[q] + trend[q] : bed[q]; };
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
  if (tid == 0) {
    part_sum[chain * parts + part] = ((red[0] + red[1]) + red[2]) + red[3];
    part_bad[chain * parts + part] = redb[0] + redb[1] + redb[2] + redb[3];
  }
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

int sgs_loss_parts(const StaticFields& S) { return std::max(1, std::min(64, (S.H * S.W + 4095) / 4096)); }

hipError_t launch_sgs_loss(const StaticFields& S, int n_chains, const double* beds, const double* trend, double* loss, int32_t* bad,
                           double* part_sum, int32_t* part_bad, hipStream_t st) {
  const int parts = sgs_loss_parts(S);
  hipLaunchKernelGGL(sgs_loss_kernel, dim3(n_chains, parts), dim3(256), 0, st, S, beds, trend, parts, part_sum, part_bad);
  hipLaunchKernelGGL(sgs_loss_finish_kernel, dim3((n_chains + 63) / 64), dim3(64), 0, st, n_chains, parts, S.two_sigma2, part_sum, part_bad, loss, bad);
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

}  // namespace gsm
