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
//   cz_group_* / cz_zgen    bucket the proposals of a launch by (size, range class); draw z ~ N(0, I) (Philox).
//   cz_gemm_kernel          F^T[p][n] = sum_{k <= n} Z[k][p] U[k][n] on the fp64 matrix cores
//                           (v_mfma_f64_16x16x4_f64, 64x64 block tiles staged through LDS, triangular K range),
//                           epilogue: * scale * edge mask -> the field layout the step kernel consumes.
//
// Algorithmic flops per proposal: N^2 (N = bh*bw; N^2/2 multiply-adds), SURVEY.md section 8d.

#include "gsm_internal.h"
#include "philox.h"
#include <math.h>

namespace gsm {

typedef double v4f64 __attribute__((ext_vector_type(4)));
constexpr int kTS = 80;   // LDS row stride in doubles (== 16 mod 32: conflict-free fragment reads)

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
  atomicAdd(&c.counts[g], 1);
  if (a.rf_scalars) {
    a.rf_scalars[4 * rec] = scale;
    a.rf_scalars[4 * rec + 1] = 0.0;
    a.rf_scalars[4 * rec + 2] = (double)rc;
    a.rf_scalars[4 * rec + 3] = (double)rc;
  }
}

// one thread: exclusive scans of the per-group proposal counts (records, padded columns, 64-wide tiles)
__global__ void cz_scan_kernel(const ProposeArgs a, const CholArgs c) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int rec_off = 0, tile_off = 0, work_off = 0;
  int64_t z_off = 0;
  for (int g = 0; g < c.n_groups; ++g) {
    const int cnt = c.counts[g];
    const int ppad = (cnt + 63) & ~63;
    const int si = g / c.n_classes;
    const int N = a.B.bh[si] * a.B.bw[si];
    c.rec_off[g] = rec_off;
    c.tile_off[g] = tile_off;
    c.z_off[g] = z_off;
    c.cursor[g] = 0;
    c.work_off[g] = work_off;
    rec_off += cnt;
    tile_off += ppad >> 6;
    work_off += (ppad >> 6) * ((N + 63) >> 6);            // 64 x 64 output tiles of this group
    z_off += (int64_t)((N + 63) & ~63) * ppad;
  }
  c.rec_off[c.n_groups] = rec_off;
  c.tile_off[c.n_groups] = tile_off;
  c.work_off[c.n_groups] = work_off;
}

__global__ __launch_bounds__(256) void cz_scatter_kernel(const ProposeArgs a, const CholArgs c) {
  const int64_t rec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (rec >= (int64_t)a.n_chains * a.n_steps) return;
  const int g = c.group_of[rec];
  const int pos = atomicAdd(&c.cursor[g], 1);
  c.order[c.rec_off[g] + pos] = (int)rec;
}

__device__ __forceinline__ int find_group(const int* __restrict__ tile_off, int n_groups, int tile) {
  int g = 0;
  while (g + 1 < n_groups && tile_off[g + 1] <= tile) ++g;
  return g;
}

// Z[k][p] ~ N(0,1): counter (chain seed, absolute step, stream kStreamCholesky, k >> 1); zero in the padding
__global__ __launch_bounds__(256) void cz_zgen_kernel(const ProposeArgs a, const CholArgs c) {
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
      const u32x4 r = philox_draw(seed, step, kStreamCholesky, (uint32_t)kp);
      const double u1 = u01_open0_from(r.x, r.y), u2 = u01_from(r.z, r.w);
      const double rad = sqrt(-2.0 * log(u1));
      double sn, cs;
      sincospi(2.0 * u2, &sn, &cs);
      z0 = rad * cs;
      z1 = (2 * kp + 1 < N) ? rad * sn : 0.0;
    }
    Z[(int64_t)(2 * kp) * ppad + p] = z0;
    Z[(int64_t)(2 * kp + 1) * ppad + p] = z1;
  }
}

// ---------------------------------------------------------------------------------------------------
// F^T = Z^T U  (block tile 64 proposals x 64 cells, 4 waves of 32x32, K step 16, LDS double buffer)
// ---------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void cz_gemm_kernel(const ProposeArgs a, const CholArgs c) {
  __shared__ double As[2][16][kTS];
  __shared__ double Bs[2][16][kTS];
  // Output tiles are enumerated group by group ((block size, range class): one factor U, one Z), n-tile major inside a
  // group: the workgroups in flight at any time then share one factor (<= 164 MB of its upper half) and one Z (<= 36 MB),
  // which the Infinity Cache holds -- with a (p-tile, n-tile) grid over all groups every n-tile pass re-read every group's
  // Z from HBM.
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
  const int n0 = (idx / n_pt) * 64;
  const int p0 = (idx % n_pt) * 64;
  const double* __restrict__ Z = c.zbuf + c.z_off[g];        // [Npad][ppad]
  const double* __restrict__ U = c.factors[g];               // [Npad][Npad] upper triangular (= L^T), zero padded
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int wp = (wave & 1) * 32, wn = (wave >> 1) * 32;     // this wave's 32x32 sub-tile
  const int lr = tid >> 4, lc = (tid & 15) * 4;              // staging: row 0..15, 4 doubles at column lc

  v4f64 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = v4f64{0.0, 0.0, 0.0, 0.0};

  const int kend = min(Npad, n0 + 64);                       // U[k][n] = 0 for k > n
  const double* zp = Z + (int64_t)lr * ppad + p0 + lc;
  const double* up = U + (int64_t)lr * Npad + n0 + lc;
  double2 za = *(const double2*)zp, zb = *(const double2*)(zp + 2);
  double2 ua = *(const double2*)up, ub = *(const double2*)(up + 2);
  int buf = 0;
  for (int k0 = 0; k0 < kend; k0 += 16) {
    *(double2*)&As[buf][lr][lc] = za; *(double2*)&As[buf][lr][lc + 2] = zb;
    *(double2*)&Bs[buf][lr][lc] = ua; *(double2*)&Bs[buf][lr][lc + 2] = ub;
    __syncthreads();
    if (k0 + 16 < kend) {
      const double* zn = zp + (int64_t)(k0 + 16) * ppad;
      const double* un = up + (int64_t)(k0 + 16) * Npad;
      za = *(const double2*)zn; zb = *(const double2*)(zn + 2);
      ua = *(const double2*)un; ub = *(const double2*)(un + 2);
    }
#pragma unroll
    for (int kk = 0; kk < 16; kk += 4) {
      const double a0 = As[buf][kk + l4][wp + l15], a1 = As[buf][kk + l4][wp + 16 + l15];
      const double b0 = Bs[buf][kk + l4][wn + l15], b1 = Bs[buf][kk + l4][wn + 16 + l15];
      acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
    }
    buf ^= 1;   // the other buffer is free: its readers passed the barrier of this iteration
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
        for (int j = 0; j < 2; ++j) {
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
  hipError_t e = hipMemsetAsync(c.counts, 0, sizeof(int) * c.n_groups, st);
  if (e != hipSuccess) return e;
  const unsigned gb = (unsigned)((nrec + 255) / 256);
  hipLaunchKernelGGL(cz_scalars_kernel, dim3(gb), dim3(256), 0, st, a, c);
  hipLaunchKernelGGL(cz_scan_kernel, dim3(1), dim3(64), 0, st, a, c);
  hipLaunchKernelGGL(cz_scatter_kernel, dim3(gb), dim3(256), 0, st, a, c);
  const unsigned max_tiles = (unsigned)((nrec + 63) / 64 + c.n_groups);
  const int nmax = a.B.max_bh * a.B.max_bw;
  hipLaunchKernelGGL(cz_zgen_kernel, dim3(max_tiles, 16), dim3(256), 0, st, a, c);
  hipLaunchKernelGGL(cz_gemm_kernel, dim3(max_tiles * (unsigned)((nmax + 63) / 64)), dim3(256), 0, st, a, c);
  return hipGetLastError();
}

}  // namespace gsm
