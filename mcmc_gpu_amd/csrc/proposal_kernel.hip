// Philox proposal generator for gfx950 (MI355X): one workgroup per (chain, step) proposal.
//
// Replaces RandField.get_rfblock + spectral_synthesis_field (reference gstatsMCMC/MCMC.py:742-778,
// :176-254), the block-centre rejection loop (MCMC.py:1253-1261) and rng.random() (MCMC.py:1336) with
// a counter-based generator.  The draws are different numbers than NumPy's PCG64 stream gives, so this
// path is checked (a) value-for-value against the CPU restatement of THIS algorithm
// (oracle/philox_oracle.py, <=1e-10 of the field scale) and (b) in distribution against the reference's
// spectral proposal (covariance, accept rate).
//
// Same distribution, half the work.  The reference takes Re(ifft2((N1 + i N2) * sqrt(S))) with two
// full planes of normals.  The real part of an inverse DFT is the inverse DFT of the Hermitian part of
// the spectrum, Zh[k] = sqrt(S[k]) * ((N1[k] + N1[-k])/2 + i (N2[k] - N2[-k])/2).  Its entries are
// independent over {k, -k} pairs with variance 1/2 per component (variance 1, real, where k == -k), so
// the kernel draws the half plane kx in [0, bw/2] directly -- bh*bw normals instead of 2*bh*bw -- and
// runs a complex-to-real inverse DFT: columns first (complex, bw/2+1 of them), then rows (real output).
//
// The two DFT stages are dense fp64 matrix products on the matrix cores (v_mfma_f64_16x16x4_f64):
//   stage 1   T^T[kx][y]  = sum_ky X[ky][kx] * exp(+2 pi i ky y / bh)      A = folded X^T (LDS), B = cos/sin table (L2)
//   stage 2   field[y][x] = sum_k (Tr^T[k][y] Gc[k][x] + Ti^T[k][y] Gs[k][x])  A = T^T (LDS), B = G tables (L2)
// with Gc = c_k cos(2 pi k x / bw), Gs = -c_k sin(...) (c_k = 1 for k in {0, bw/2}, else 2: the Hermitian half), both
// folded by the even/odd symmetry of cos/sin so that only indices <= n/2 enter the products (4x / 2x fewer flops than
// the dense DFT; see propose_kernel).  The field never touches LDS: it stays in the MFMA accumulators through the
// standardisation and goes straight to HBM.
//
// MFMA operand maps used here (16x16x4 f64): A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15],
// C/D reg q: row = (lane >> 4) + 4 q, col = lane & 15.

#include "gsm_internal.h"
#include "philox.h"
#include "proposal_device.h"
#include <math.h>
#include <stdlib.h>

namespace gsm {

// spectral-amplitude parameters of a proposal from its range draws (MCMC.py:209-239)
__device__ __forceinline__ void spectral_params(PropScalars& r, const gsm_rf_params& P) {
  double lx, ly;
  if (P.model == GSM_MODEL_GAUSSIAN) { lx = r.range_x / sqrt(3.0); ly = r.range_y / sqrt(3.0); }
  else if (P.model == GSM_MODEL_EXPONENTIAL) { lx = r.range_x / 3.0; ly = r.range_y / 3.0; }
  else { lx = r.range_x / 2.0; ly = r.range_y / 2.0; }
  r.aa = sqrt(lx * ly);
  r.m_const = 0.0; r.m_kappa = 0.0;
  if (P.model == GSM_MODEL_MATERN) {
    const double nu = (P.smoothness != 0.0) ? P.smoothness : 1.0;
    r.m_const = 0.5 * log((4.0 * M_PI * tgamma(nu + 1.0) * pow(2.0 * nu, nu)) / (tgamma(nu) * pow(r.aa, 2.0 * nu)));
    r.m_kappa = 2.0 * nu / (r.aa * r.aa);
  }
}

// block shape, DFT-table offsets and mask offset of size index r.si
__device__ __forceinline__ void block_shape(PropScalars& r, const ProposeArgs& a) {
  r.pad = a.k2_off[r.si];
  r.bh = a.B.bh[r.si];
  r.bw = a.B.bw[r.si];
  r.fy_off = a.fy_off[r.bh];
  r.g_off = a.g_off[r.bw];
  r.mask_off = a.B.mask_off[r.si];
  const int ncol = r.bw / 2 + 1;
  r.m_nc = pmagic((uint32_t)ncol);
  r.m_m1 = pmagic((uint32_t)((ncol + 15) & ~15));
  // halo tile of the clipped window (make_win in chain_fused_kernel.hip)
  const int c0 = max(0, r.col - r.bw / 2), c1 = min(a.W, r.col + r.bw / 2);
  r.m_tw = pmagic((uint32_t)max(1, min(a.W, c1 + 1) - max(0, c0 - 1)));
  r.reserved = 0;
  r.m_bh = pmagic((uint32_t)r.bh); r.m_bw = pmagic((uint32_t)r.bw);
  r.t1h_off = a.t1_off ? a.t1_off[r.bh] : 0;
  r.t1w_off = a.t1_off ? a.t1_off[r.bw] : 0;
}

// ---------------------------------------------------------------------------------------------------
// per-proposal scalars: one thread per (chain, step).  Draw layout (stream kStreamScalars):
//   idx 0: {scale u, nugget u}   idx 1: {range_x u, range_y u}   idx 2: {accept u, centre word}   idx 3: {size word}
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void propose_scalars_kernel(const ProposeArgs a) {
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
  PropScalars r;
  r.si = (int)__umulhi(d3.x, (uint32_t)a.B.n_sizes);
  r.scale = (P.scale_min + (P.scale_max - P.scale_min) * u01_from(d0.x, d0.y)) / 3.0;
  r.nug = 0.0 + (P.nugget_max - 0.0) * u01_from(d0.z, d0.w);
  r.range_x = P.range_min_x + (P.range_max_x - P.range_min_x) * u01_from(d1.x, d1.y);
  r.range_y = P.isotropic ? r.range_x : P.range_min_y + (P.range_max_y - P.range_min_y) * u01_from(d1.z, d1.w);
  r.u = u01_from(d2.x, d2.y);
  const uint64_t cw = ((uint64_t)d2.w << 32) | d2.z;
  const int cell = a.centres[(int)__umul64hi(cw, (uint64_t)a.n_centres)];
  r.row = cell / a.W;
  r.col = cell - r.row * a.W;
  spectral_params(r, P);
  block_shape(r, a);
  a.scalars[rec] = r;
  a.size_idx[rec] = r.si;
  a.centre[2 * rec] = r.row;
  a.centre[2 * rec + 1] = r.col;
  a.u[rec] = r.u;
  if (a.rf_scalars) {
    a.rf_scalars[4 * rec] = r.scale;
    a.rf_scalars[4 * rec + 1] = r.nug;
    a.rf_scalars[4 * rec + 2] = r.range_x;
    a.rf_scalars[4 * rec + 3] = r.range_y;
  }
}

// One workgroup of NT threads per (chain, step); body in proposal_device.h.  WIDE = 2: block tables whose largest shape
// needs more than 16 stage-2 output tiles (beyond ~80 x 80): one workgroup per CU, twice the accumulators per wave.
template <int NT, int WIDE>
__global__ __launch_bounds__(NT, (WIDE == 1) ? NT / 256 : 1) void propose_kernel(const ProposeArgs a) {
  extern __shared__ double plds[];
  double* red = plds + a.lds_main;         // [32] reductions + [kMathTabDoubles] math table
  const int s = blockIdx.x, chain = blockIdx.y;
  const int64_t rec = (int64_t)chain * a.n_steps + s;
  const PropScalars sc = a.scalars[rec];
  if (a.dbg & 64) { if (threadIdx.x == 0) a.fields[rec * a.field_stride] = (double)sc.bh; return; }
  double* __restrict__ out = a.fields + rec * a.field_stride;
  propose_field<NT, false, WIDE>((int)threadIdx.x, a, sc, a.seeds[chain], a.step0 + s, plds, red, out,
                                 [bw = sc.bw](int y, int x) { return y * bw + x; });
}

// k^2 tables: for block size s the [bh/2 + 1][bw/2 + 1] values (sqrt(kx^2 + ky^2) + 1e-10)^2 with kx, ky = 2 pi fftfreq
// (MCMC.py:221-224) -- one workgroup per block size.  They depend on the grid resolution only, which arrives with the
// random-field parameters: rebuilt when it changes.
__global__ __launch_bounds__(256) void k2_table_kernel(const BlockTable B, const int32_t* __restrict__ k2_off, double resolution,
                                                       double* __restrict__ k2tab) {
  const int si = blockIdx.x;
  const int bh = B.bh[si], bw = B.bw[si];
  const int nrow = bh / 2 + 1, ncol = bw / 2 + 1;
  const double inv_x = 1.0 / ((double)bw * resolution), inv_y = 1.0 / ((double)bh * resolution);
  double* out = k2tab + k2_off[si];
  for (int i = threadIdx.x; i < nrow * ncol; i += 256) {
    const int ky = i / ncol, kx = i - ky * ncol;
    const double kxv = wavenumber(kx, bw, inv_x), kyv = wavenumber(ky, bh, inv_y);
    const double k = sqrt(kxv * kxv + kyv * kyv) + 1e-10;
    out[i] = k * k;
  }
}

hipError_t launch_k2_tables(const BlockTable& B, const int32_t* k2_off, double resolution, double* k2tab, hipStream_t st) {
  hipLaunchKernelGGL(k2_table_kernel, dim3(B.n_sizes), dim3(256), 0, st, B, k2_off, resolution, k2tab);
  return hipGetLastError();
}

// gsm_spectral_from_noise: scalar records from caller-supplied (size index, scale, nugget, ranges) ...
__global__ __launch_bounds__(256) void noise_scalars_kernel(const ProposeArgs a, const int32_t* __restrict__ size_idx,
                                                            const double* __restrict__ rf_scalars) {
  const int rec = blockIdx.x * 256 + threadIdx.x;
  if (rec >= a.n_steps) return;
  PropScalars r;
  r.si = size_idx[rec];
  if (r.si < 0 || r.si >= a.B.n_sizes) { r.si = 0; r.bh = 0; }   // flagged by the host wrapper before the launch
  r.scale = rf_scalars[4 * rec]; r.nug = rf_scalars[4 * rec + 1];
  r.range_x = rf_scalars[4 * rec + 2]; r.range_y = rf_scalars[4 * rec + 3];
  r.u = 0.0; r.row = 0; r.col = 0;
  spectral_params(r, a.rf);
  block_shape(r, a);
  a.scalars[rec] = r;
}

// gsm_run_noise: the same plus the step's block centre and accept uniform -- the records chain_strip_kernel<NOISE> reads
__global__ __launch_bounds__(256) void noise_chain_scalars_kernel(const ProposeArgs a, const int32_t* __restrict__ size_idx,
                                                                  const int32_t* __restrict__ centre, const double* __restrict__ u,
                                                                  const double* __restrict__ rf_scalars, int32_t* __restrict__ err_flag) {
  const int64_t rec = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (rec >= (int64_t)a.n_chains * a.n_steps) return;
  PropScalars r;
  r.si = size_idx[rec]; r.row = centre[2 * rec]; r.col = centre[2 * rec + 1];
  if (r.si < 0 || r.si >= a.B.n_sizes || r.row < 0 || r.row >= a.H || r.col < 0 || r.col >= a.W) {
    atomicExch(err_flag, 1);                 // reported by the host wrapper after the launch; the step runs on a valid stand-in
    r.si = 0; r.row = 0; r.col = 0;
  }
  r.scale = rf_scalars[4 * rec]; r.nug = rf_scalars[4 * rec + 1];
  r.range_x = rf_scalars[4 * rec + 2]; r.range_y = rf_scalars[4 * rec + 3];
  r.u = u[rec];
  spectral_params(r, a.rf);
  block_shape(r, a);
  a.scalars[rec] = r;
}
hipError_t launch_noise_chain_scalars(const ProposeArgs& a, const int32_t* size_idx, const int32_t* centre, const double* u,
                                      const double* rf_scalars, int32_t* err_flag, hipStream_t st) {
  const int64_t nrec = (int64_t)a.n_chains * a.n_steps;
  hipLaunchKernelGGL(noise_chain_scalars_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, a, size_idx, centre, u, rf_scalars, err_flag);
  return hipGetLastError();
}

// ... and the synthesis itself: propose_field with the coefficients formed from the caller's noise planes
template <int WIDE>
__global__ __launch_bounds__(512, (WIDE == 1) ? 2 : 1) void spectral_from_noise_kernel(const ProposeArgs a, const double* __restrict__ noise_re,
                                                                                       const double* __restrict__ noise_im,
                                                                                       const double* __restrict__ nugget_field) {
  extern __shared__ double plds[];
  double* red = plds + a.lds_main;
  const int64_t rec = blockIdx.x;
  const PropScalars sc = a.scalars[rec];
  const NoiseIn nz{noise_re + rec * a.field_stride, noise_im + rec * a.field_stride,
                   nugget_field ? nugget_field + rec * a.field_stride : nullptr};
  propose_field<512, true, WIDE>((int)threadIdx.x, a, sc, 0, 0, plds, red, a.fields + rec * a.field_stride,
                                 [bw = sc.bw](int y, int x) { return y * bw + x; }, nz);
}

static bool wide_table(const ProposeArgs& a) { return a.tiles2_max > 16 || a.tiles1_max > 16; }

hipError_t launch_spectral_from_noise(const ProposeArgs& a, const int32_t* size_idx, const double* rf_scalars,
                                      const double* noise_re, const double* noise_im, const double* nugget_field, hipStream_t st) {
  const size_t lds = ((size_t)a.lds_main + 32 + kMathTabDoubles) * sizeof(double);
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)spectral_from_noise_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)spectral_from_noise_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  hipLaunchKernelGGL(noise_scalars_kernel, dim3((unsigned)((a.n_steps + 255) / 256)), dim3(256), 0, st, a, size_idx, rf_scalars);
  if (wide_table(a)) hipLaunchKernelGGL(spectral_from_noise_kernel<2>, dim3(a.n_steps), dim3(512), lds, st, a, noise_re, noise_im, nugget_field);
  else hipLaunchKernelGGL(spectral_from_noise_kernel<1>, dim3(a.n_steps), dim3(512), lds, st, a, noise_re, noise_im, nugget_field);
  return hipGetLastError();
}

// test hook (gsm_debug_normals): normals2 exactly as the coefficient phase calls it
__global__ __launch_bounds__(256) void debug_normals_kernel(uint64_t seed, int64_t step, uint32_t stream_id, uint32_t idx0, int n,
                                                            const double* __restrict__ mathtab, double* __restrict__ out) {
  __shared__ double mt[kMathTabDoubles];
  for (int i = threadIdx.x; i < kMathTabDoubles; i += 256) mt[i] = mathtab[i];
  __syncthreads();
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  double g1, g2;
  normals2(seed, step, stream_id, idx0 + (uint32_t)i, g1, g2, mt);
  out[2 * i] = g1; out[2 * i + 1] = g2;
}
hipError_t launch_debug_normals(uint64_t seed, int64_t step, uint32_t stream_id, uint32_t idx0, int n, const double* mathtab, double* out,
                                hipStream_t st) {
  hipLaunchKernelGGL(debug_normals_kernel, dim3((n + 255) / 256), dim3(256), 0, st, seed, step, stream_id, idx0, n, mathtab, out);
  return hipGetLastError();
}

hipError_t launch_propose(const ProposeArgs& a_in, hipStream_t st) {
  ProposeArgs a = a_in;
  { static int dbg = -1; if (dbg < 0) { const char* v = getenv("GSM_PROPOSE_DBG"); dbg = v ? atoi(v) : 0; } a.dbg = dbg; }
  const size_t lds = ((size_t)a.lds_main + 32 + kMathTabDoubles) * sizeof(double);
  static int nt = -1;   // GSM_PROPOSE_NT=1024: the fused kernel's workgroup size (tests: bit-identical fields)
  if (nt < 0) { const char* v = getenv("GSM_PROPOSE_NT"); nt = (v && atoi(v) == 1024) ? 1024 : 512; }
  static bool attr_set[kMaxDevices] = {};
  int attr_dev;
  if (attr_needed_on_this_device(attr_set, attr_dev)) {
    hipError_t e = hipFuncSetAttribute((const void*)propose_kernel<512, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)propose_kernel<1024, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)propose_kernel<512, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (e != hipSuccess) return e;
    if (attr_dev >= 0) attr_set[attr_dev] = true;
  }
  if (lds > 160 * 1024) return hipErrorInvalidValue;
  const int64_t nrec = (int64_t)a.n_chains * a.n_steps;
  hipLaunchKernelGGL(propose_scalars_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, a);
  if (wide_table(a)) hipLaunchKernelGGL((propose_kernel<512, 2>), dim3(a.n_steps, a.n_chains), dim3(512), lds, st, a);
  else if (nt == 1024) hipLaunchKernelGGL((propose_kernel<1024, 1>), dim3(a.n_steps, a.n_chains), dim3(1024), lds, st, a);
  else hipLaunchKernelGGL((propose_kernel<512, 1>), dim3(a.n_steps, a.n_chains), dim3(512), lds, st, a);
  return hipGetLastError();
}

hipError_t launch_propose_scalars(const ProposeArgs& a, hipStream_t st) {
  const int64_t nrec = (int64_t)a.n_chains * a.n_steps;
  hipLaunchKernelGGL(propose_scalars_kernel, dim3((unsigned)((nrec + 255) / 256)), dim3(256), 0, st, a);
  return hipGetLastError();
}

int propose_max_tiles1_per_wave() { return 4; }   // wide instantiation: 32 stage-1 output tiles (64 half-tile units) on 8 waves
int propose_max_tiles_per_wave() { return 4; }    // 32 stage-2 output tiles
int propose_waves() { return 8; }

}  // namespace gsm

// host-visible self test of the Philox implementation (declared in include/gsm.h)
extern "C" int gsm_philox_selftest(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
  gsm::u32x4 c{ctr4[0], ctr4[1], ctr4[2], ctr4[3]};
  const gsm::u32x4 r = gsm::philox4x32_10(c, key2[0], key2[1]);
  out4[0] = r.x; out4[1] = r.y; out4[2] = r.z; out4[3] = r.w;
  return 0;
}
