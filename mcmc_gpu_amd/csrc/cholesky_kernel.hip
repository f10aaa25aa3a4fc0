// Covariance assembly and the precomputed-factor (Cholesky "L z") proposal generator for gfx950.
//
// The reference lists a Cholesky/LU random-field generator as future work (README.md:21-23) and ships only its
// ingredients: the covariance models on normalised lag (gstatsMCMC/gstatsim_custom/covariance.py:4-28), the
// anisotropy rotation (gstatsim_custom/_krige.py:83-103) and the dense assembly (_krige.py:105-122).  This file
// builds the generator BASELINE.json's north_star names from them:
//
//   cov_assemble_kernel     Sigma[a][b] = cov(|| (coord_a - coord_b) @ R ||), coord = (j*res, i*res) of block cells
//                           -> _krige.make_sigma.  Matern takes its values from a host lag table (scipy.special.kv,
//                           as covariance.py:17-22 does); the closed-form models are evaluated here.
//   chol_*_kernel           U = chol(Sigma + jitter I)^T, once per (block size, range class): setup, not hot path.
//   cz_scalars / cz_bucket / cz_zgen   per-proposal scalars; bucket the proposals of a launch by (size, range class) in one
//                           workgroup (LDS counters); draw z ~ N(0, I) (Philox, table-driven Box-Muller).
//   cz_gemm_dma_kernel      F^T[p][n] = sum_{k <= n} Z[k][p] U[k][n] on the fp64 matrix cores
//                           (v_mfma_f64_16x16x4_f64, 64 x 128 block tiles, operands global -> LDS by LDS-DMA, triangular
//                           K range), epilogue: * scale * edge mask -> the field layout the step kernel consumes.
//
// Algorithmic flops per proposal: N^2 (N = bh*bw; N^2/2 multiply-adds), SURVEY.md section 8d.

#include "gsm_internal.h"
#include "philox.h"
#include "proposal_device.h"
#include <math.h>

namespace gsm {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kTS = 80;    // LDS row stride in doubles of a 64-wide tile (== 16 mod 32: conflict-free fragment reads)
constexpr int kTS2 = 144;  // ... of the 128-wide U tile

// ---------------------------------------------------------------------------------------------------
// covariance assembly
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double cov_norm(double h, const gsm_vario& v) {
  const double c0 = v.sill - v.nugget;
  switch (v.vtype) {
    case GSM_VTYPE_EXPONENTIAL: return c0 * exp(-3.0 * h);
    case GSM_VTYPE_GAUSSIAN: return c0 * exp(-3.0 * (h * h));
    case GSM_VTYPE_SPHERICAL: {
      // reference quirk kept: beyond the range the value is sill - 1 (covariance.py:14)
      const double c = v.sill - v.nugget - 1.5 * h + 0.5 * (h * h * h);
      return (h > 1.0) ? v.sill - 1.0 : c;
    }
    default: return 0.0;
  }
}

// lag_table (Matern only): [2*bh-1][2*bw-1] covariance at (di, dj) = (row - (bh-1), col - (bw-1))
__global__ __launch_bounds__(256) void cov_assemble_kernel(int bh, int bw, double res, gsm_vario v, double r00,
                                                           double r01, double r10, double r11,
                                                           const double* __restrict__ lag_table, double* __restrict__ sigma,
                                                           int ld) {
  const int N = bh * bw;
  const int64_t total = (int64_t)N * N;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int a = (int)(t / N), b = (int)(t - (int64_t)a * N);
    const int ia = a / bw, ja = a - ia * bw, ib = b / bw, jb = b - ib * bw;
    double c;
    if (lag_table) {
      c = lag_table[(ia - ib + bh - 1) * (2 * bw - 1) + (ja - jb + bw - 1)];
    } else {
      // (coord @ R) per point, then the difference, as squareform(pdist(coord @ R)) does
      const double xa = ja * res, ya = ia * res, xb = jb * res, yb = ib * res;
      const double ma0 = xa * r00 + ya * r10, ma1 = xa * r01 + ya * r11;
      const double mb0 = xb * r00 + yb * r10, mb1 = xb * r01 + yb * r11;
      const double d0 = ma0 - mb0, d1 = ma1 - mb1;
      c = cov_norm(sqrt(d0 * d0 + d1 * d1), v);
    }
    sigma[(int64_t)a * ld + b] = c;
  }
}

hipError_t launch_cov_assemble(int bh, int bw, double res, const gsm_vario& v, const double* lag_table, double* sigma,
                               int ld, hipStream_t st) {
  const double th = (v.azimuth / 180.0) * M_PI;
  const double c = cos(th), s = sin(th);
  // R = [[c, -s], [s, c]] @ diag(1/major, 1/minor)   (_krige.py:96-101)
  const double r00 = c * (1.0 / v.major_range), r01 = -s * (1.0 / v.minor_range);
  const double r10 = s * (1.0 / v.major_range), r11 = c * (1.0 / v.minor_range);
  hipLaunchKernelGGL(cov_assemble_kernel, dim3(2048), dim3(256), 0, st, bh, bw, res, v, r00, r01, r10, r11, lag_table,
                     sigma, ld);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------------
// proposal scalars + grouping
// ---------------------------------------------------------------------------------------------------
// draw layout (stream kStreamScalars): idx 0 {scale u, -}, idx 1 {range-class word, -}, idx 2 {accept u, centre word},
// idx 3 {size word}
__global__ __launch_bounds__(256) void cz_scalars_kernel(const ProposeArgs a, const CholArgs c) {
  const int64_t rec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (rec >= (int64_t)a.n_chains * a.n_steps) return;
  const int chain = (int)(rec / a.n_steps), s = (int)(rec - (int64_t)chain * a.n_steps);
  const int64_t step = a.step0 + s;
  const uint64_t seed = a.seeds[chain];
  const gsm_rf_params& P = a.rf;
  const u32x4 d0 = philox_draw(seed, step, kStreamScalars, 0);
  const u32x4 d1 = philox_draw(seed, step, kStreamScalars, 1);
  const u32x4 d2 = philox_draw(seed, step, kStreamScalars, 2);
  const u32x4 d3 = philox_draw(seed, step, kStreamScalars, 3);
  const int si = (int)__umulhi(d3.x, (uint32_t)a.B.n_sizes);
  const int rc = (int)__umulhi(d1.x, (uint32_t)c.n_classes);
  const double scale = (P.scale_min + (P.scale_max - P.scale_min) * u01_from(d0.x, d0.y)) / 3.0;
  const uint64_t cw = ((uint64_t)d2.w << 32) | d2.z;
  const int cell = a.centres[(int)__umul64hi(cw, (uint64_t)a.n_centres)];
  a.size_idx[rec] = si;
  a.centre[2 * rec] = cell / a.W;
  a.centre[2 * rec + 1] = cell - (cell / a.W) * a.W;
  a.u[rec] = u01_from(d2.x, d2.y);
  c.scale[rec] = scale;
  const int g = si * c.n_classes + rc;
  c.group_of[rec] = g;
  if (a.rf_scalars) {
    a.rf_scalars[4 * rec] = scale;
    a.rf_scalars[4 * rec + 1] = 0.0;
    a.rf_scalars[4 * rec + 2] = (double)rc;
    a.rf_scalars[4 * rec + 3] = (double)rc;
  }
}

// Bucketing of a launch's proposals by group (block size, range class), ONE 1024-thread workgroup: counts and cursors live in
// LDS (one global atomic per record on ~50 counters serialised per address: 0.27 + 0.16 ms per 32768 records), the exclusive
// scans of the counts (records, padded columns, 64-wide tiles) by thread 0 in between.  The order of a group's records is
// whatever the LDS atomics make it: a record's field depends on its own (seed, step) only, not on its column.
__global__ __launch_bounds__(1024) void cz_bucket_kernel(const ProposeArgs a, const CholArgs c) {
  extern __shared__ int bk[];                     // [n_groups] counts, [n_groups] cursors
  int* cnt = bk;
  int* cur = bk + c.n_groups;
  const int tid = threadIdx.x;
  const int64_t nrec = (int64_t)a.n_chains * a.n_steps;
  for (int g = tid; g < c.n_groups; g += 1024) cnt[g] = 0;
  __syncthreads();
  for (int64_t rec = tid; rec < nrec; rec += 1024) atomicAdd(&cnt[c.group_of[rec]], 1);
  __syncthreads();
  if (tid == 0) {
    int rec_off = 0, tile_off = 0, work_off = 0;
    int64_t z_off = 0;
    for (int g = 0; g < c.n_groups; ++g) {
      const int n = cnt[g];
      const int ppad = (n + 63) & ~63;
      const int si = g / c.n_classes;
      const int N = a.B.bh[si] * a.B.bw[si];
      c.counts[g] = n;
      c.rec_off[g] = rec_off;
      c.tile_off[g] = tile_off;
      c.z_off[g] = z_off;
      c.work_off[g] = work_off;
      cur[g] = rec_off;
      rec_off += n;
      tile_off += ppad >> 6;
      work_off += (ppad >> 6) * ((N + 127) >> 7);           // 64 x 128 output tiles of this group
      z_off += (int64_t)((N + 63) & ~63) * ppad;
    }
    c.rec_off[c.n_groups] = rec_off;
    c.tile_off[c.n_groups] = tile_off;
    c.work_off[c.n_groups] = work_off;
  }
  __syncthreads();
  for (int64_t rec = tid; rec < nrec; rec += 1024) c.order[atomicAdd(&cur[c.group_of[rec]], 1)] = (int)rec;
}

// the group whose [off[g], off[g + 1]) holds `tile` (off ascending, off[0] = 0): binary search, ~log2(n_groups) dependent loads
__device__ __forceinline__ int find_group(const int* __restrict__ tile_off, int n_groups, int tile) {
  int lo = 0, hi = n_groups - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tile_off[mid] <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// Z[k][p] ~ N(0,1): counter (chain seed, absolute step, stream kStreamCholesky, k >> 1); zero in the padding.  Box-Muller with
// the table-driven log / sincos and the lean sqrt of the spectral generator (normals2, proposal_device.h; pinned against NumPy
// on the same counters to 5e-15, tests/test_gpu_philox.py): half the time of libm's log / sincospi.
__global__ __launch_bounds__(256) void cz_zgen_kernel(const ProposeArgs a, const CholArgs c) {
  __shared__ double mt[kMathTabDoubles];
  for (int i = threadIdx.x; i < kMathTabDoubles; i += 256) mt[i] = a.mathtab[i];
  __syncthreads();
  const int tile = blockIdx.x;
  if (tile >= c.tile_off[c.n_groups]) return;
  const int g = find_group(c.tile_off, c.n_groups, tile);
  const int cnt = c.counts[g];
  const int ppad = (cnt + 63) & ~63;
  const int si = g / c.n_classes;
  const int N = a.B.bh[si] * a.B.bw[si];
  const int Npad = (N + 63) & ~63;
  const int p = (tile - c.tile_off[g]) * 64 + (threadIdx.x & 63);
  double* __restrict__ Z = c.zbuf + c.z_off[g];
  uint64_t seed = 0;
  int64_t step = 0;
  const bool live = p < cnt;
  if (live) {
    const int rec = c.order[c.rec_off[g] + p];
    const int chain = rec / a.n_steps;
    seed = a.seeds[chain];
    step = a.step0 + (rec - chain * a.n_steps);
  }
  // 4 k-pair lanes per proposal column; blockIdx.y strides over the pairs
  for (int kp = blockIdx.y * 4 + (threadIdx.x >> 6); kp < Npad / 2; kp += gridDim.y * 4) {
    double z0 = 0.0, z1 = 0.0;
    if (live && 2 * kp < N) {
      normals2_key(seed, step, kStreamCholesky, (uint32_t)kp, z0, z1, mt);
      if (2 * kp + 1 >= N) z1 = 0.0;
    }
    Z[(int64_t)(2 * kp) * ppad + p] = z0;
    Z[(int64_t)(2 * kp + 1) * ppad + p] = z1;
  }
}

// ---------------------------------------------------------------------------------------------------
// F^T = Z^T U on 64 (proposals) x 128 (cells) block tiles: four waves of 32 x 64 (eight accumulator tiles each; per K = 4 six
// fragment reads feed eight MFMAs), K step 16, two LDS stages, triangular K range (U[k][n] = 0 for k > n).
// The operand tiles travel global -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane): no VGPR staging and no ds_write --
// the VGPR -> LDS store path costs ~13 LDS cycles per ds_write_b128 (MI355X_MICROARCH.md, LDS) and a register-staged loop
// lost 11 % of its time to it (ablation).  An LDS-DMA instruction fills 1 KiB of
// CONTIGUOUS LDS, lane l bytes [16 l, 16 l + 16), from any 16 global bytes the lane names: the padded row layout (row stride
// 80 / 144 doubles == 16 mod 32: conflict-free fragment reads) is kept by letting every lane fetch the element pair that
// belongs at its LDS position; the lanes that land on padding re-fetch a valid pair.  28 chunks per stage, 7 per wave.
__global__ __launch_bounds__(256) void cz_gemm_dma_kernel(const ProposeArgs a, const CholArgs c) {
  constexpr int kStageA = 16 * kTS, kStageB = 16 * kTS2, kStage = kStageA + kStageB;     // 1280 + 2304 doubles = 10 + 18 chunks
  static_assert(kStageA % 128 == 0 && kStageB % 128 == 0, "whole 1 KiB chunks");
  __shared__ __attribute__((aligned(16))) double stage[2][kStage];
  const int work = blockIdx.x;
  if (work >= c.work_off[c.n_groups]) return;
  const int g = find_group(c.work_off, c.n_groups, work);
  const int cnt = c.counts[g];
  const int ppad = (cnt + 63) & ~63;
  const int si = g / c.n_classes;
  const int N = a.B.bh[si] * a.B.bw[si];
  const int Npad = (N + 63) & ~63;
  const int n_pt = ppad >> 6;
  const int idx = work - c.work_off[g];
  const int n0 = (idx / n_pt) * 128;
  const int p0 = (idx % n_pt) * 64;
  const double* __restrict__ Z = c.zbuf + c.z_off[g];        // [Npad][ppad]
  const double* __restrict__ U = c.factors[g];               // [Npad][Npad] upper triangular (= L^T), zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wp = (wave & 1) * 32, wn = (wave >> 1) * 64;     // this wave's 32 x 64 sub-tile
  const bool right = n0 + 64 < Npad;                          // the tile's right half exists (else its columns are never stored)

  // this lane's source of each of the wave's 7 chunks at k0 = 0, and the chunks' LDS offsets
  const double* src[7];
  int64_t step[7];
  int dst[7];
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int ch = wave + 4 * i;                               // 0..9: Z tile, 10..27: U tile
    if (ch < 10) {
      const int u = ch * 64 + lane, row = u / 40, cp = u - row * 40;
      src[i] = Z + (int64_t)row * ppad + p0 + 2 * (cp < 32 ? cp : 0);
      step[i] = 16 * (int64_t)ppad;
      dst[i] = ch * 128;
    } else {
      const int u = (ch - 10) * 64 + lane, row = u / 72, cp = u - row * 72;
      const int col = (cp < 64 && (right || cp < 32)) ? cp : 0;
      src[i] = U + (int64_t)row * Npad + n0 + 2 * col;
      step[i] = 16 * (int64_t)Npad;
      dst[i] = kStageA + (ch - 10) * 128;
    }
  }
  auto request = [&](int b) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      __builtin_amdgcn_global_load_lds(src[i], (__attribute__((address_space(3))) void*)(&stage[b][0] + dst[i]), 16, 0, 0);
      src[i] += step[i];
    }
  };

  v4f64 acc[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};

  const int kend = min(Npad, n0 + 128);                      // U[k][n] = 0 for k > n
  request(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  int buf = 0;
  for (int k0 = 0; k0 < kend; k0 += 16) {
    if (k0 + 16 < kend) request(buf ^ 1);                     // its last readers passed the barrier that ended the previous iteration
    const double* As = &stage[buf][0];
    const double* Bs = As + kStageA;
    if (k0 < n0 + wn + 64)                                     // beyond this wave's own diagonal U[k][n] = 0: no MFMAs (wave-uniform)
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      const double a0 = As[(kk + l4) * kTS + wp + l15], a1 = As[(kk + l4) * kTS + wp + 16 + l15];
      double b[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = Bs[(kk + l4) * kTS2 + wn + 16 * j + l15];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b[j], acc[0][j], 0, 0, 0);
        acc[1][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b[j], acc[1][j], 0, 0, 0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's chunks of the next stage have landed
    __syncthreads();
    buf ^= 1;
  }

  // epilogue: field[rec][n] = acc * scale[rec] * mask[n]
  const double* __restrict__ mask = a.B.masks + a.B.mask_off[si];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int p = p0 + wp + 16 * i + l4 + 4 * q;
      if (p < cnt) {
        const int rec = c.order[c.rec_off[g] + p];
        const double sc = c.scale[rec];
        double* __restrict__ out = a.fields + (int64_t)rec * a.field_stride;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = n0 + wn + 16 * j + l15;
          if (n < N) out[n] = (acc[i][j][q] * sc) * mask[n];
        }
      }
    }
}

// ---------------------------------------------------------------------------------------------------
// In-place blocked Cholesky, upper form: A = U^T U, A row-major [n][ld], n a multiple of 64 (setup time; one
// factor per block size and range class).  Right-looking, 64-wide panels:
//   chol_diag_kernel    U_kk = chol(A_kk) in LDS (one workgroup), lower triangle of the block zeroed
//   chol_panel_kernel   A[k, j] <- U_kk^{-T} A[k, j] for the block columns j > k (one thread per column, U_kk in LDS)
//   chol_update_kernel  A[i, j] -= U[k, i]^T U[k, j] for k < i <= j on the fp64 matrix cores (64x64 tiles, K = 64)
// A non-positive pivot sets *info (1-based pivot index) and the factorisation is abandoned by the host.
// ---------------------------------------------------------------------------------------------------
constexpr int kCB = 64;

__global__ __launch_bounds__(kCB) void chol_diag_kernel(double* __restrict__ A, int ld, int k0, double jitter, int* info) {
  __shared__ double a[kCB][kCB + 1];
  const int c = threadIdx.x;
  for (int r = 0; r < kCB; ++r) a[r][c] = A[(size_t)(k0 + r) * ld + k0 + c] + ((r == c) ? jitter : 0.0);
  __syncthreads();
  for (int s = 0; s < kCB; ++s) {
    const double piv = a[s][s];
    if (!(piv > 0.0)) { if (c == 0) atomicCAS(info, 0, k0 + s + 1); return; }   // uniform: every thread reads the same pivot
    const double d = sqrt(piv);
    __syncthreads();
    if (c >= s) a[s][c] = (c == s) ? d : a[s][c] / d;
    __syncthreads();
    if (c > s)
      for (int r = s + 1; r <= c; ++r) a[r][c] -= a[s][r] * a[s][c];
    __syncthreads();
  }
  for (int r = 0; r < kCB; ++r) A[(size_t)(k0 + r) * ld + k0 + c] = (c >= r) ? a[r][c] : 0.0;
}

__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int ld, int n, int k0) {
  __shared__ double u[kCB][kCB + 1];
  for (int t = threadIdx.x; t < kCB * kCB; t += 256) u[t / kCB][t % kCB] = A[(size_t)(k0 + t / kCB) * ld + k0 + t % kCB];
  __syncthreads();
  const int c = k0 + kCB + blockIdx.x * 256 + threadIdx.x;
  if (c >= n) return;
  double x[kCB];
#pragma unroll
  for (int r = 0; r < kCB; ++r) {
    double v = A[(size_t)(k0 + r) * ld + c];
#pragma unroll
    for (int q = 0; q < r; ++q) v -= u[q][r] * x[q];
    x[r] = v / u[r][r];
  }
#pragma unroll
  for (int r = 0; r < kCB; ++r) A[(size_t)(k0 + r) * ld + c] = x[r];
}

// one workgroup per (i, j) tile pair with k < i <= j; blockIdx.x enumerates the pairs of the trailing triangle
__global__ __launch_bounds__(256) void chol_update_kernel(double* __restrict__ A, int ld, int k0, int nt_trail) {
  __shared__ double Pi[kCB][kTS];
  __shared__ double Pj[kCB][kTS];
  // pair index -> (ti <= tj) in the trailing nt_trail x nt_trail block grid
  int p = blockIdx.x, ti = 0;
  while (p >= nt_trail - ti) { p -= nt_trail - ti; ++ti; }
  const int tj = ti + p;
  const int i0 = k0 + kCB * (1 + ti), j0 = k0 + kCB * (1 + tj);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  for (int t = tid; t < kCB * 16; t += 256) {          // 64 rows x 16 groups of 4 doubles
    const int r = t >> 4, c4 = (t & 15) * 4;
    const double* pi = A + (size_t)(k0 + r) * ld + i0 + c4;
    const double* pj = A + (size_t)(k0 + r) * ld + j0 + c4;
    *(double2*)&Pi[r][c4] = *(const double2*)pi; *(double2*)&Pi[r][c4 + 2] = *(const double2*)(pi + 2);
    *(double2*)&Pj[r][c4] = *(const double2*)pj; *(double2*)&Pj[r][c4 + 2] = *(const double2*)(pj + 2);
  }
  __syncthreads();
  const int wm = (wave & 1) * 32, wn = (wave >> 1) * 32;
  v4f64 acc[2][2];
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int b_ = 0; b_ < 2; ++b_) acc[a_][b_] = v4f64{0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int kk = 0; kk < kCB; kk += 4) {
    const double a0 = Pi[kk + l4][wm + l15], a1 = Pi[kk + l4][wm + 16 + l15];
    const double b0 = Pj[kk + l4][wn + l15], b1 = Pj[kk + l4][wn + 16 + l15];
    acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
  }
#pragma unroll
  for (int a_ = 0; a_ < 2; ++a_)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int m = i0 + wm + 16 * a_ + l4 + 4 * q;
#pragma unroll
      for (int b_ = 0; b_ < 2; ++b_) {
        const int nn = j0 + wn + 16 * b_ + l15;
        if (nn >= m) A[(size_t)m * ld + nn] -= acc[a_][b_][q];   // upper triangle only
      }
    }
}

// zero the strict lower triangle (the update touches only the upper one) -- cosmetic for callers that read U whole
__global__ __launch_bounds__(256) void chol_clear_lower_kernel(double* __restrict__ A, int ld, int n) {
  const int64_t total = (int64_t)n * n;
  for (int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x; t < total; t += (int64_t)gridDim.x * 256) {
    const int r = (int)(t / n), c = (int)(t - (int64_t)r * n);
    if (c < r) A[(size_t)r * ld + c] = 0.0;
  }
}

hipError_t launch_cholesky_upper(double* A, int n, int ld, double jitter, int* d_info, hipStream_t st) {
  const int nb = n / kCB;
  for (int k = 0; k < nb; ++k) {
    const int k0 = k * kCB;
    hipLaunchKernelGGL(chol_diag_kernel, dim3(1), dim3(kCB), 0, st, A, ld, k0, jitter, d_info);
    const int rem = n - k0 - kCB;
    if (rem > 0) {
      hipLaunchKernelGGL(chol_panel_kernel, dim3((rem + 255) / 256), dim3(256), 0, st, A, ld, n, k0);
      const int nt = rem / kCB;
      hipLaunchKernelGGL(chol_update_kernel, dim3(nt * (nt + 1) / 2), dim3(256), 0, st, A, ld, k0, nt);
    }
  }
  hipLaunchKernelGGL(chol_clear_lower_kernel, dim3(2048), dim3(256), 0, st, A, ld, n);
  return hipGetLastError();
}

hipError_t launch_propose_cholesky(const ProposeArgs& a, const CholArgs& c, hipStream_t st) {
  const int64_t nrec = (int64_t)a.n_chains * a.n_steps;
  if (c.n_groups > 4096) return hipErrorInvalidValue;            // cz_bucket_kernel keeps 2 ints per group in LDS
  const unsigned gb = (unsigned)((nrec + 255) / 256);
  hipLaunchKernelGGL(cz_scalars_kernel, dim3(gb), dim3(256), 0, st, a, c);
  hipLaunchKernelGGL(cz_bucket_kernel, dim3(1), dim3(1024), sizeof(int) * 2 * c.n_groups, st, a, c);
  const unsigned max_tiles = (unsigned)((nrec + 63) / 64 + c.n_groups);
  const int nmax = a.B.max_bh * a.B.max_bw;
  hipLaunchKernelGGL(cz_zgen_kernel, dim3(max_tiles, 16), dim3(256), 0, st, a, c);
  hipLaunchKernelGGL(cz_gemm_dma_kernel, dim3(max_tiles * (unsigned)((nmax + 127) / 128)), dim3(256), 0, st, a, c);
  return hipGetLastError();
}

}  // namespace gsm
