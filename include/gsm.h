/*
 * gsm.h -- C ABI of the MI355X-native many-chain geostatistical MCMC sampler (libgsm_hip.so).
 *
 * This is the drop-in boundary for ONE hot path of tylerrleee/mcmc-gpu: the large-scale-chain
 * Metropolis step.  Every entry point names the reference interface it replaces (paths relative to
 * the reference repository root).  The reference is pure Python, so its "FFI" for this path is a
 * ctypes binding; INTEGRATION.md shows the stub a maintainer adds to gstatsMCMC/MCMC_gpu.py.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - Every pointer marked [dev] is a DEVICE pointer (e.g. torch.Tensor.data_ptr() of a cuda tensor),
 *     row-major, borrowed for the duration of the call (asynchronous calls: until the stream has been
 *     synchronised).  The caller owns all state; the library owns only its handle and scratch.
 *   - Return value 0 = ok, negative = error (GSM_E_*); gsm_last_error() gives the text.  No C++
 *     exception crosses the ABI.  A handle is not thread-safe; distinct handles are independent.
 *   - "stream" is a hipStream_t passed as void* (0 = the null stream).  Kernels are enqueued on it and
 *     the call returns without synchronising unless stated.
 *   - Grids are H rows x W columns.  Static fields are fp64; the per-chain state (beds, energy) is fp64, or
 *     fp32 for a handle created with dtype 1 (then `void* beds` / `void* energy` point to float arrays).
 *     Chain c's bed starts at element c*H*W.
 *
 * Layout of per-step records, for chain c and step s of a call with n_steps steps: index c*n_steps+s.
 */
#ifndef GSM_H
#define GSM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gsm_context* gsm_handle;

enum {
  GSM_OK = 0,
  GSM_E_ARG = -1,       /* bad argument (null pointer, non-positive size, block larger than grid ...) */
  GSM_E_STATE = -2,     /* call order violated (e.g. run before set_static) */
  GSM_E_HIP = -3,       /* a HIP runtime call failed */
  GSM_E_UNSUPPORTED = -4, /* valid request this build cannot serve (dtype, block larger than the LDS tile) */
  GSM_E_DEVICE_DATA = -5  /* a kernel found an out-of-range size index / block centre in device data */
};

enum { GSM_MODEL_GAUSSIAN = 0, GSM_MODEL_EXPONENTIAL = 1, GSM_MODEL_MATERN = 2 };

/* Random-field parameters: mirrors RandField.__init__ (gstatsMCMC/MCMC.py:462-510) plus the grid
 * resolution handed to spectral_synthesis_field (MCMC.py:176, :767). */
typedef struct {
  double range_min_x, range_max_x, range_min_y, range_max_y;
  double scale_min, scale_max;
  double nugget_max;
  double smoothness;  /* Matern nu; ignored otherwise */
  double resolution;
  int32_t model;      /* GSM_MODEL_* */
  int32_t isotropic;  /* 1: one range draw shared by x and y (MCMC.py:207) */
  int32_t generator;  /* GSM_GEN_SPECTRAL (reference's generator) or GSM_GEN_CHOLESKY (precomputed factors) */
  int32_t reserved;
} gsm_rf_params;

enum { GSM_GEN_SPECTRAL = 0, GSM_GEN_CHOLESKY = 1 };
enum { GSM_VTYPE_EXPONENTIAL = 0, GSM_VTYPE_GAUSSIAN = 1, GSM_VTYPE_SPHERICAL = 2, GSM_VTYPE_MATERN = 3 };

/* Variogram of gstatsim_custom (the `vario` dict of _krige.py:83-122 / covariance.py:4-28). */
typedef struct {
  double azimuth, major_range, minor_range, sill, nugget, s;
  int32_t vtype;      /* GSM_VTYPE_* */
  int32_t reserved;
} gsm_vario;

/* Library / build identification: "gsm-hip <version> gfx950 src:<hash>", hash = first 16 hex digits of the SHA-256 of the
 * library sources at build time (the Python loader checks it against the sources it sees). */
const char* gsm_version(void);

/* Text of the last error on this handle (or of the last failed gsm_create when h is NULL). */
const char* gsm_last_error(gsm_handle h);

/* Create a sampler for n_chains chains on an H x W grid on HIP device `device`.
 * dtype: 0 = fp64 state; 1 = fp32 state with fp64 arithmetic (beds and energy are float arrays; every value is
 * rounded to float before it is used, so the carried loss sum always equals the sum of what is stored --
 * BASELINE configs[4]; the reference's torch class is all-fp32, MCMC_gpu.py:261).
 * Replaces: the per-process chain construction of lsc_run_wrapper
 * (largeScaleChain_multiprocessing_GPU.py:126-127) -- one handle holds all chains of one GPU. */
int gsm_create(gsm_handle* out, int32_t H, int32_t W, int32_t n_chains, int32_t dtype, int32_t device);
int gsm_destroy(gsm_handle h);

/* Static fields shared by all chains [dev, H*W each].  crf_weight may be NULL => block_type 'RF'
 * (MCMC.py:1279-1282).  update_mask: 1 where a proposal may change the bed AND where the thickness
 * guard and resampled_times apply (region_mask when update_in_region else grounded_ice_mask,
 * MCMC.py:1287-1290, :1323-1328, :1349-1352).  mc_mask: 1 where the residual enters the loss
 * (mc_region_mask, MCMC.py:1041).  The arrays are copied; the caller may free them afterwards.
 * Replaces: chain.__init__ / set_update_region / set_loss_type / set_crf_data_weight state
 * (MCMC.py:808-848, :849-872, :950-1018, :1124-1134). */
int gsm_set_static(gsm_handle h, const double* surf, const double* velx, const double* vely,
                   const double* dhdt, const double* smb, const double* crf_weight,
                   const uint8_t* update_mask, const uint8_t* mc_mask,
                   double resolution, double sigma_mc, void* stream);

/* Proposal block table: n_sizes (bh, bw) pairs [host] and their edge masks packed back to back [dev],
 * mask i starting at mask_offsets[i] doubles [host, n_sizes entries].  Masks are needed only by the
 * Philox proposal generator; replay mode receives fields that already carry the mask.
 * Replaces: RandField.set_block_sizes / set_weight_param (MCMC.py:524-566, :568-623). */
int gsm_set_blocks(gsm_handle h, int32_t n_sizes, const int32_t* bh, const int32_t* bw,
                   const double* edge_masks_packed, const int64_t* mask_offsets, void* stream);

/* Cells (flat index row*W+col) [dev] from which Philox mode draws block centres uniformly --
 * the cells with region_mask==1; same distribution as the rejection loop of MCMC.py:1253-1261. */
int gsm_set_centres(gsm_handle h, const int32_t* cells, int32_t n_cells, void* stream);

/* Full-grid residual and loss of every chain's current bed: initialises the carried state.
 * energy   [dev, n_chains*H*W]: r^2 where the cell enters the loss (mc_mask == 1 and r not NaN: nansum
 *           semantics of MCMC.py:1041), else 0 -- the build's counterpart of the carried residual array
 *           mc_res (MCMC.py:1189, :1308-1315, :1340);
 * loss_sum [dev, n_chains*2]: compensated (hi, lo) pair with hi + lo = sum(energy);
 * loss0    [dev, n_chains] (may be NULL): sum / (2 sigma^2) = loss_cache[0].
 * Replaces: MCMC.py:1189-1195 (Topography.get_mass_conservation_residual, Topography.py:592-600,
 * and chain.loss, MCMC.py:1021-1044). */
int gsm_init_loss(gsm_handle h, const void* beds, void* energy, double* loss_sum, double* loss0, void* stream);

/* Full-grid residual of chain beds [dev, n_chains*H*W] -> residual [dev, same shape].
 * Replaces: Topography.get_mass_conservation_residual (Topography.py:592-600). */
int gsm_residual(gsm_handle h, const double* beds, double* residual, void* stream);

/* n_steps Metropolis steps per chain with caller-supplied draws ("replay" / parity mode).
 *   size_idx [dev, n_chains*n_steps]      index into the block table
 *   centre   [dev, n_chains*n_steps*2]    (row, col) block centre
 *   u        [dev, n_chains*n_steps]      uniform draw of the accept test
 *   fields   [dev]                        masked proposal fields f (bh x bw, row-major) at
 *                                         fields + (c*n_steps+s)*field_stride doubles
 * In/out state: beds, energy, resampled (uint32 counts), loss_sum.  Outputs: loss [n_chains*n_steps]
 * (loss_cache entries), accept [n_chains*n_steps] (0/1).
 * Synchronises the stream before returning (it reports out-of-range device data as GSM_E_DEVICE_DATA).
 * Replaces: the loop body of chain_crf.run, MCMC.py:1263-1360 (torch twin MCMC_gpu.py:385-494). */
int gsm_run_replay(gsm_handle h, int32_t n_steps, void* beds, void* energy, uint32_t* resampled, double* loss_sum,
                   const int32_t* size_idx, const int32_t* centre, const double* u,
                   const double* fields, int64_t field_stride,
                   double* loss, uint8_t* accept, void* stream);

/* Philox4x32-10 proposal generator: for every chain and steps step0 .. step0+n_steps-1 draws the block
 * size, the block centre, the accept uniform and one spectral-synthesis field (already multiplied by
 * its edge mask), into the same layout gsm_run_replay consumes.  rf_scalars (optional, may be NULL)
 * [dev, n_chains*n_steps*4] receives (scale, nugget, range_x, range_y).
 * Key = seeds[c] [dev, n_chains]; counters are functions of (absolute step, draw index) only, so a run
 * split into segments reproduces the unsplit run.  Block tables up to about 110 x 110 (the four coefficient planes must fit
 * 160 KiB of LDS, at most 32 output tiles of 16 x 16 per DFT stage); larger tables return GSM_E_UNSUPPORTED.  Tables beyond
 * ~80 x 80 use a wider instantiation of the proposal kernel and, in gsm_run_philox, the two-kernel pipeline.
 * Replaces: RandField.get_rfblock + spectral_synthesis_field (MCMC.py:742-778, :176-254), the centre
 * rejection loop (MCMC.py:1253-1261) and rng.random() (MCMC.py:1336). */
int gsm_propose_philox(gsm_handle h, int32_t n_steps, int64_t step0, const uint64_t* seeds,
                       const gsm_rf_params* rf, int32_t* size_idx, int32_t* centre, double* u,
                       double* fields, int64_t field_stride, double* rf_scalars, void* stream);

/* The reference's OWN random numbers on the device ('pcg64' draw mode): every chain's two numpy.random.Generator(PCG64) streams --
 * RandField.rng and the chain's generator, SURVEY.md section 8 a13 -- advanced by n_steps Metropolis steps' worth of draws in the
 * reference's call order, bit for bit what NumPy returns on the host (PCG64 XSL-RR, the 32-bit half cache of integers(), Lemire's
 * bounded integers, uniform, the 256-layer ziggurat normal incl. libm's log1p in its tail: mcmc_gpu_amd/csrc/pcg64_device.h,
 * restated in oracle/pcg64_oracle.py and pinned there against NumPy itself).
 *   rf_state, chain_state [dev, n_chains*6] in/out   per generator: state lo, state hi, inc lo, inc hi, has_uint32, uinteger -- the
 *                                                   integers of numpy's bit_generator.state dict
 *   region_mask [dev, H*W] or NULL                   centre rejection of MCMC.py:1253-1258 (NULL: update_in_region False)
 * Outputs, record r = c * n_steps + s: size_idx[r] (RandField.rng.integers(0, n_sizes), MCMC.py:755), rf_scalars[4 r] = (scale / 3,
 * nugget, range_x, range_y) (MCMC.py:199-207), noise_re / noise_im / nugget_field at r * field_stride: the rng.normal planes of
 * MCMC.py:242 and :251, row-major (bh, bw) (nugget_field may be NULL when rf->nugget_max == 0: the draws are consumed, nothing is
 * stored), centre[2 r] (row, col) and u[r] (MCMC.py:1336).  Feed them to gsm_spectral_from_noise and gsm_run_replay: the chain
 * then follows the CPU reference on the same seeds -- accept masks identical, beds / losses to the 1e-12 x scale of the device's
 * inverse DFT against pocketfft -- without any host draw.  Asynchronous on `stream`.
 * Replaces: the numpy.random calls of RandField.get_rfblock / spectral_synthesis_field / chain_crf.run listed above. */
int gsm_draw_pcg64(gsm_handle h, int32_t n_steps, const gsm_rf_params* rf, uint64_t* rf_state, uint64_t* chain_state,
                   const uint8_t* region_mask, int32_t* size_idx, int32_t* centre, double* u, double* rf_scalars,
                   double* noise_re, double* noise_im, double* nugget_field, int64_t field_stride, void* stream);

/* The 'pcg64' draw mode with synthesis and step in ONE launch: per (chain, step) record r = c * n_steps + s of gsm_draw_pcg64's
 * outputs, the proposal field is formed from the white-noise planes exactly as gsm_spectral_from_noise forms it (the same stage
 * functions, bit-identical fields) and consumed by the Metropolis step exactly as gsm_run_replay consumes it, inside the
 * workgroup that owns the chain -- the field never goes to HBM (chain_strip_kernel.hip, two chains per CU).  Results equal
 * gsm_spectral_from_noise + gsm_run_replay on the same handle bit for bit (tests/test_gpu_pcg64.py).  Needs a block table that
 * goes to the strip kernels (gsm_strip_active), else GSM_E_UNSUPPORTED.  nugget_field may be NULL (no nugget term).  Checks the
 * size indices and centres on the device (GSM_E_DEVICE_DATA).  Synchronises `stream` before returning (error flag).
 * Replaces: RandField.get_rfblock + spectral_synthesis_field + the loop body of chain_crf.run (MCMC.py:742-778, :176-254,
 * :1247-1360) for draws that are the reference's own. */
int gsm_run_noise(gsm_handle h, int32_t n_steps, void* beds, void* energy, uint32_t* resampled, double* loss_sum,
                  const int32_t* size_idx, const int32_t* centre, const double* u, const double* rf_scalars,
                  const gsm_rf_params* rf, const double* noise_re, const double* noise_im, const double* nugget_field,
                  int64_t field_stride, double* loss, uint8_t* accept, void* stream);

/* The spectral synthesis alone, fed with CALLER-SUPPLIED white noise instead of Philox draws: the value pin of the device
 * arithmetic against the reference's.  For field r of n_fields:
 *   size_idx   [dev, n_fields]                 block-table index -> shape (bh, bw)
 *   rf_scalars [dev, n_fields*4]               (scale, nugget, range_x, range_y): `scale` is the value multiplied in at
 *                                              MCMC.py:251 (already / 3), ranges as drawn at MCMC.py:204-207
 *   noise_re, noise_im [dev]                   the two rng.normal(size=(bh, bw)) planes of MCMC.py:242, row-major at
 *                                              r*field_stride doubles
 *   nugget_field [dev] (may be NULL)           rng.normal(0, sqrt(nug), size=(bh, bw)) of MCMC.py:251, same layout
 * The library forms the Hermitian half Zh[k] = sqrt(S(k)) ((N1[k] + N1[-k])/2 + i (N2[k] - N2[-k])/2) -- Re(ifft2(Z)) is
 * the inverse DFT of the Hermitian part of Z -- and then runs EXACTLY the code of gsm_propose_philox / the fused kernel
 * on it: spectral amplitude, folded matrix-core inverse DFT, standardisation, scale, nugget, edge mask (MCMC.py:221-251,
 * :778).  fields [dev] receives f * edge_mask at r*field_stride.
 * Replaces: spectral_synthesis_field (MCMC.py:176-254) for given draws. */
int gsm_spectral_from_noise(gsm_handle h, int32_t n_fields, const int32_t* size_idx, const double* rf_scalars,
                            const gsm_rf_params* rf, const double* noise_re, const double* noise_im,
                            const double* nugget_field, double* fields, int64_t field_stride, void* stream);

/* Philox mode end to end.  Spectral generator: the fused chain kernel (per chain-step the proposal is generated and
 * consumed inside one workgroup; nothing but chain state touches HBM), launched once per segment of at most 4096 steps
 * (scratch: one 140-byte scalar record per chain and step of a SEGMENT; `batch` is not used).  Cholesky generator, block
 * tables beyond the fused kernel's LDS budget, or GSM_FUSED=0 in the environment: batches of `batch` steps, proposals of
 * batch k+1 generated on a second stream while batch k is stepped, scratch owned by the handle.
 * Both forms give bit-identical results, whatever the segment / batch size.  Outputs as gsm_run_replay plus blocks
 * [dev, n_chains*n_steps*4] = (row, col, bh, bw) (blocks_cache, MCMC.py:1264).  Synchronises the stream before returning.
 * Replaces: chain_crf.run for a whole shard of chains (MCMC.py:1137-1443) as called from
 * lsc_run_wrapper (largeScaleChain_multiprocessing_GPU.py:194-201). */
int gsm_run_philox(gsm_handle h, int32_t n_steps, int64_t step0, int32_t batch, const uint64_t* seeds,
                   const gsm_rf_params* rf, void* beds, void* energy, uint32_t* resampled, double* loss_sum,
                   double* loss, uint8_t* accept, int32_t* blocks, void* stream);

/* Select the launch structure of gsm_run_philox for the spectral generator on this handle: 1 = fused chain kernel
 * (default), 0 = two-kernel pipeline.  Identical results; kept for A/B measurements and tests.  Environment default:
 * GSM_FUSED. */
int gsm_set_fused(gsm_handle h, int32_t on);
/* Which form the last gsm_run_philox call on this handle ran: 1 = fused chain kernel, 0 = the two-kernel pipeline. */
int gsm_last_run_fused(gsm_handle h);
/* Which step kernels this handle's static fields and block table select (decided once per table, for gsm_run_replay and
 * gsm_run_philox alike, so that fused == propose + replay holds bit for bit): 1 = the strip kernels (chain_strip_kernel.hip:
 * 512-thread workgroups, two chains per CU, candidate bed in a (bh + 2) x (bw + 2) LDS tile), 0 = the flux-tile kernels
 * (1024-thread workgroups, one chain per CU: grids whose packed static operands, 48 bytes per cell, exceed an XCD's 4 MiB L2
 * -- more than 87 381 cells --, block tables whose strips would exceed 16 rows or whose proposal work area exceeds 80 KiB,
 * GSM_STRIP=0).  Needs gsm_set_static and gsm_set_blocks.  Same arithmetic per cell (MCMC.py:1279-1360,
 * Topography.py:592-600); the window sums of a step are taken in another order (loss within 1e-15 relative). */
int gsm_strip_active(gsm_handle h);

/* Average duration in milliseconds of the step kernel / the proposal kernel over the launches made
 * by the last gsm_run_philox call, measured with HIP events on the streams the kernels ran on
 * (0 when timing was not enabled with gsm_enable_timing(h, 1)). */
int gsm_enable_timing(gsm_handle h, int32_t on);
int gsm_last_timing(gsm_handle h, double* step_kernel_ms, int32_t* step_launches,
                    double* proposal_kernel_ms, int32_t* proposal_launches);

/* Dense covariance of the bh*bw cells of a block (cell a = i*bw + j at (x, y) = (j*res, i*res)):
 * sigma[a*ld + b] = cov(|| (coord_a - coord_b) @ R ||) with R the anisotropy rotation/scaling matrix.
 * lag_table [dev, (2bh-1)*(2bw-1)] is required for GSM_VTYPE_MATERN (values from scipy.special.kv, which is what
 * covariance.py:17-22 evaluates) and optional otherwise.  sigma [dev, N*ld], ld >= N = bh*bw.
 * Replaces: gstatsim_custom._krige.make_rotation_matrix / make_sigma (_krige.py:83-122) with the covariance
 * models of gstatsim_custom/covariance.py:4-28 (incl. the spherical model's `sill - 1` beyond the range). */
int gsm_cov_assemble(gsm_handle h, int32_t bh, int32_t bw, double resolution, const gsm_vario* vario,
                     const double* lag_table, double* sigma, int64_t ld, void* stream);

/* In-place Cholesky factorisation, upper form: on entry a [dev, n*ld] holds a symmetric positive-definite matrix
 * (only its upper triangle is read), on exit its upper triangle holds U with U^T U = A + jitter*I and the strict lower
 * triangle is zero.  n must be a multiple of 64 (pad with an identity block).  Setup-time helper of the Cholesky
 * proposal generator (blocked right-looking, panel updates on the fp64 matrix cores); synchronises the stream.
 * GSM_E_ARG with the failing pivot in gsm_last_error when the matrix is not positive definite. */
int gsm_cholesky_upper(gsm_handle h, double* a, int32_t n, int64_t ld, double jitter, void* stream);

/* Precomputed factors of the Cholesky proposal generator: for block size i and range class r,
 * factors[i*n_classes + r] [host array of dev pointers] is U = chol(Sigma + jitter I)^T, upper triangular,
 * row-major [Npad][Npad] with Npad = N rounded up to 64 and zero padding.  The matrices stay caller-owned.
 * With rf->generator == GSM_GEN_CHOLESKY, gsm_propose_philox / gsm_run_philox draw f = scale * (L z) * edge_mask
 * (z ~ N(0, I) from Philox, size and range class uniform) instead of the spectral field.
 * The reference has no such generator (README.md:21-23 lists it as future work); this is north_star's. */
int gsm_set_factors(gsm_handle h, int32_t n_classes, const double* const* factors, void* stream);

/* ---- small-scale chain (chain_sgs, gstatsMCMC/MCMC.py:1445-1911) ------------------------------------------------------
 * Sequential Gaussian simulation of one block per chain with ordinary kriging on octant-searched neighbours.
 *   grids    [dev, n_chains*H*W] in/out  the conditioning grid of each chain (`bed_tosim`, MCMC.py:1766-1771): values
 *                                       everywhere except NaN at the cells to simulate; on return the block's cells hold
 *                                       the simulated values (`newsim`, MCMC.py:1774)
 *   zcond    [dev, H*W] or NULL         if given, the block's cells are first set from it (the conditioning data of the
 *                                       block, NaN where there is none: z_cond_bed, MCMC.py:1771)
 *   windows  [dev, n_chains*4]          (row0, row1, col0, col1) of each chain's block; at most 1024 cells
 *   x_axis [dev, W], y_axis [dev, H]    coordinates of the columns / rows (the grid must be axis-aligned: xx[i][j] = x_axis[j])
 *   lag_cov  [dev, (2 lag_mi + 1) * (2 lag_mj + 1)]  covariance at the integer lag (di, dj), |di| <= lag_mi, |dj| <= lag_mj,
 *                                       row-major: the caller evaluates the variogram model (covariance.py:4-28 through
 *                                       make_sigma / make_rho, _krige.py:105-143) once per lag.  Two chosen neighbours are at
 *                                       most 2 hw apart: lag_mi >= min(2 hw, H - 1), lag_mj >= min(2 hw, W - 1); the whole grid
 *                                       (H - 1, W - 1) if the radius-widening fallback is to work
 *   hw                                  search half-width in cells: ceil(radius / |x_axis[1] - x_axis[0]|), the half-width of the
 *                                       reference's circle stencil (neighbors.py:66-83) -- any value >= 1 (the reference's driver:
 *                                       30 km at 500 m = 60).  The ring-by-ring search certifies a sector as complete from
 *                                       per-ring counters kept for the first 64 rings; beyond ring 64 (hw > 64 with sparse data) a
 *                                       sector completes only when its candidates are exhausted -- same result, more passes
 *   cell_off [dev, n_chains+1], cells [dev, total*2]   the (row, col) of each chain's cells in simulation order -- the
 *                                       caller's rng.shuffle (MCMC.py:128); every cell must lie inside the chain's window
 *   z        [dev, total]               one standard normal per listed cell (used only if the cell is simulated):
 *                                       rng.normal(est, sqrt(var)) = est + sqrt(var) * z (MCMC.py:165)
 *   max_cells                           upper bound of a chain's cell count (<= 1024): stride of the library's scratch records
 *   trace    [dev, total*3] or NULL     (number of neighbours, kriging estimate, kriging variance) per cell; -1 neighbours =
 *                                       cell was conditioned already
 *   nbr_trace [dev, total*48] or NULL   the chosen neighbours of every simulated cell (flat index row * W + col, sector by
 *                                       sector, ascending distance; -1 padded)
 * num_points in [8, 48] (num_points / 8 per octant; equidistant candidates in ascending (row, col): numpy.argsort(kind='stable')
 * of the reference's masked array -- the reference's default argsort is unstable, its choice there is implementation-defined).
 * A cell with no value within `radius` widens the search by 100 km steps like the reference (MCMC.py:150-156).
 * The kriging systems are solved by Gauss-Jordan elimination on the diagonal (the covariance block is symmetric positive
 * definite) where the reference calls numpy.linalg.lstsq (_krige.py:37): weights agree to about cond(Sigma) * 1e-16 relative,
 * not bit for bit.  lstsq(rcond=None) truncates singular values below eps * N * s_max; here a pivot below eps * N * max|diag|
 * ends the call with GSM_E_DEVICE_DATA ("singular kriging system") instead -- tested range: tests/test_gpu_sgs.py.
 * Ordinary kriging unless gsm_sgs_set_kriging selected simple kriging (sk_solve; chain_sgs.run itself never passes ktype,
 * MCMC.py:1599, :160-161).
 * Synchronises the stream.
 * Replaces: sgs (MCMC.py:91-173), neighbors (gstatsim_custom/neighbors.py:4-64), ok_solve (gstatsim_custom/_krige.py:5-44). */
int gsm_sgs_blocks(gsm_handle h, double* grids, const double* zcond, const int32_t* windows, const double* x_axis,
                   const double* y_axis, const double* lag_cov, int32_t lag_mi, int32_t lag_mj, int32_t hw, double radius,
                   int32_t num_points, double sill, const int32_t* cell_off, const int32_t* cells, const double* z,
                   int32_t max_cells, double* trace, int32_t* nbr_trace, void* stream);

/* gsm_sgs_blocks for batches of iterations: no trace, the chain's cells are cells[cell_off[c] .. cell_off[c] + cell_cnt[c])
 * when cell_cnt is given (else .. cell_off[c + 1]), and NO synchronisation: device-side errors accumulate in the handle and
 * are reported by gsm_sgs_check. */
int gsm_sgs_blocks_batch(gsm_handle h, double* grids, const double* zcond, const int32_t* windows, const double* x_axis,
                         const double* y_axis, const double* lag_cov, int32_t lag_mi, int32_t lag_mj, int32_t hw, double radius,
                         int32_t num_points, double sill, const int32_t* cell_off, const int32_t* cell_cnt, const int32_t* cells,
                         const double* z, int32_t max_cells, void* stream);
/* Synchronises the stream and reports what gsm_sgs_blocks_batch calls since the last check have flagged (GSM_OK if nothing). */
int gsm_sgs_check(gsm_handle h, void* stream);

/* Philox mode of the small-scale chain: the draws of n_iters iterations of every chain on the device -- what chain_sgs.run takes
 * from chain.rng per iteration (MCMC.py:1750-1765 block centre with rejection on region_mask and block sizes, :128 the visiting
 * order, :165 one normal per simulated cell, :1797 the accept uniform), from Philox4x32-10 counters (draw index, stream 4,
 * iteration), key = seeds[c]:
 *   draws 0..63: centre attempts (row = mulhi(x, H), col = mulhi(y, W)), the first with region_mask == 1 (any if NULL) is taken;
 *   draw 64: block sizes bs_x = min_x + mulhi(x, max_x - min_x), bs_y likewise (numpy's integers(low, high): high excluded), accept
 *   uniform from (z, w); draws 128 + p: visiting key of window cell p (row-major) -- cells are visited in ascending (key, p);
 *   draws 2048 + p / 2: the standard normal of cell p (first / second Box-Muller value for even / odd p).
 * Outputs, [dev], record r = j * n_chains + c for iteration iter0 + j: windows[4 r] (r0, r1, c0, c1 as MCMC.py:1758-1761),
 * blocks[4 r] (row, col, bs_x, bs_y), cell_cnt[r], cell_off[r] = r * max_cells, cells[2 (r max_cells + k)] the k-th visited cell,
 * z[r max_cells + k] its normal (0 for a conditioning cell: is_data != 0), u[r].  max_cells >= (max_x - 1) * (max_y - 1) windows fit.
 * oracle/sgs_philox_oracle.py restates it. */
int gsm_sgs_draw_philox(gsm_handle h, const uint64_t* seeds, int64_t iter0, int32_t n_iters, const uint8_t* region_mask,
                        const uint8_t* is_data, int32_t min_x, int32_t max_x, int32_t min_y, int32_t max_y, int32_t max_cells,
                        int32_t* windows, int32_t* blocks, int32_t* cell_off, int32_t* cell_cnt, int32_t* cells, double* z, double* u,
                        void* stream);

/* gsm_sgs_draw_philox's outputs from the chains' OWN NumPy generators ('pcg64' draw mode of the small-scale chain): chain_state
 * [dev, n_chains*6] (state lo, hi, inc lo, hi, has_uint32, uinteger of numpy's PCG64 bit_generator.state; in/out) is advanced by
 * n_iters iterations' worth of draws in chain_sgs.run's order -- integers(0, H), integers(0, W) until region_mask == 1 and
 * integers(min_x, max_x), integers(min_y, max_y) (MCMC.py:1750-1757), rng.shuffle of the block's (row, col) list (:128: Fisher-Yates
 * from the back, random_interval's masked rejection on 32-bit words), one rng.normal per cell without conditioning data in
 * visiting order (:165), rng.random() (:1797) -- bit for bit what NumPy returns (mcmc_gpu_amd/csrc/pcg64_device.h).  Record
 * r = j * n_chains + c for the j-th iteration; layouts as in gsm_sgs_draw_philox.  Asynchronous; device-side errors are reported
 * by gsm_sgs_check. */
int gsm_sgs_draw_pcg64(gsm_handle h, uint64_t* chain_state, int32_t n_iters, const uint8_t* region_mask, const uint8_t* is_data,
                       int32_t min_x, int32_t max_x, int32_t min_y, int32_t max_y, int32_t max_cells, int32_t* windows,
                       int32_t* blocks, int32_t* cell_off, int32_t* cell_cnt, int32_t* cells, double* z, double* u, void* stream);

/* Loss and thickness guard of proposed beds over the whole grid: loss[c] = nansum(residual(bed_c + trend)^2 where
 * mc_mask == 1) / (2 sigma^2), bad[c] = number of cells with update_mask == 1 (set it to grounded_ice_mask) and
 * surf - (bed_c + trend) <= 0.  beds [dev, n_chains*H*W], trend [dev, H*W] or NULL, loss [dev, n_chains], bad [dev, n_chains].
 * Replaces: chain_sgs.run's per-iteration loss and guard (MCMC.py:1781-1795). */
int gsm_sgs_loss(gsm_handle h, const double* beds, const double* trend, double* loss, int32_t* bad, void* stream);

/* The Metropolis decision of one small-scale iteration on the device, so that a batch of iterations needs no host round
 * trip: loss_next[c] (set to +inf where bad[c] > 0), p = 1 if loss_prev[c] > loss_next[c] else min(1, exp(loss_prev[c] -
 * loss_next[c])) (a NaN stays a rejection, as with numpy.minimum), accept[c] = u[c] <= p; an accepted chain's loss_prev[c]
 * becomes loss_next[c].  loss_rec[c * rec_stride] = loss_prev[c] after the decision and acc_rec[c * rec_stride] =
 * accept[c] (the iteration's column of the caller's [n_chains][rec_stride] record arrays; either may be NULL).  All [dev].
 * Replaces: chain_sgs.run's acceptance test (MCMC.py:1797-1812). */
int gsm_sgs_decide(gsm_handle h, const double* loss_next, const int32_t* bad, const double* u, double* loss_prev,
                   uint8_t* accept, double* loss_rec, uint8_t* acc_rec, int64_t rec_stride, void* stream);

/* Windowed end of a small-scale iteration for chains WITHOUT a normal-score transformer.  The reference recomputes residual, loss
 * and thickness guard over the whole map every iteration (MCMC.py:1781-1795); without a transformer only the block differs from the
 * current bed, so only the residuals of the block and its one-cell halo change.  Carried per chain: `energy` [dev, n_chains*H*W]
 * (squared residual where mc_mask == 1 and the residual is not NaN, else 0 -- as the large-scale step kernels carry it) and
 * state [dev, n_chains*4] = (sum of energy: hi, lo of a compensated pair; loss = sum / (2 sigma^2); number of cells with
 * update_mask == 1 -- set it to grounded_ice_mask -- and surf - (bed + trend) <= 0).
 * gsm_sgs_state_init fills both from the current beds (trend [dev, H*W] or NULL).
 * gsm_sgs_finish: loss and guard of the proposal `next` (== `cur` outside windows[c]) from the halo window, the acceptance test of
 * gsm_sgs_decide with u[c], then gsm_sgs_commit's bookkeeping (cur / next / resampled) and the update of energy / state of an
 * accepted chain.  accept [dev, n_chains]; loss_rec / acc_rec / rec_stride as in gsm_sgs_decide.  The loss equals the full-grid
 * nansum up to summation order (<= 1e-12 relative).  A window whose halo exceeds 36 x 36 cells is reported by gsm_sgs_check.
 * Replaces: chain_sgs.run's per-iteration loss, guard, acceptance test and commit (MCMC.py:1781-1812) when do_transform is False. */
int gsm_sgs_state_init(gsm_handle h, const double* beds, const double* trend, double* energy, double* state, void* stream);
int gsm_sgs_finish(gsm_handle h, double* cur, double* next, const double* trend, double* energy, double* state,
                   const int32_t* windows, const double* u, uint32_t* resampled, uint8_t* accept, double* loss_rec,
                   uint8_t* acc_rec, int64_t rec_stride, void* stream);

/* scikit-learn's QuantileTransformer(output_distribution="normal") of one feature on the device, as chain_sgs.run applies it
 * to the whole map around every SGS block (MCMC.py:1766, :1777): out[i] = transform(x[i]) (inverse == 0) or
 * inverse_transform(x[i]) (inverse != 0), i < n.  quantiles = transformer.quantiles_[:, 0], references =
 * transformer.references_, both [dev, nq] ascending; x, out [dev, n] (may alias).  NaN stays NaN.  Same arithmetic as
 * QuantileTransformer._transform_col with scipy.stats.norm.ppf / cdf (Cephes ndtri / ndtr): mcmc_gpu_amd/csrc/normal_score.h. */
int gsm_qt_transform(gsm_handle h, const double* quantiles, const double* references, int32_t nq, const double* x, double* out,
                     int64_t n, int32_t inverse, void* stream);

/* gsm_sgs_commit for a chain with a normal-score transformer: the inverse transform touches every cell of the proposed map,
 * so an accepted chain takes the WHOLE plane of `proposed` (cur[c] = proposed[c]) and the resampled counts of its window are
 * incremented (MCMC.py:1803-1812); a rejected chain keeps cur.  cur, proposed [dev, n_chains*H*W]. */
int gsm_sgs_commit_map(gsm_handle h, double* cur, const double* proposed, uint32_t* resampled, const int32_t* windows,
                       const uint8_t* accept, void* stream);

/* Accept / reject bookkeeping of one small-scale iteration: where accept[c] != 0 the block of chain c is copied from
 * `next` into `cur` and resampled counts of the block are incremented (MCMC.py:1803-1812); elsewhere the block of `next`
 * is restored from `cur`.  cur, next [dev, n_chains*H*W], resampled [dev, n_chains*H*W], windows as in gsm_sgs_blocks,
 * accept [dev, n_chains]. */
int gsm_sgs_commit(gsm_handle h, double* cur, double* next, uint32_t* resampled, const int32_t* windows,
                   const uint8_t* accept, void* stream);

/* Kriging type of the gsm_sgs_blocks* calls that follow on this handle.  GSM_KRIGING_ORDINARY (the default; what
 * chain_sgs.run uses, MCMC.py:1774 passes no ktype): ok_solve, the (n+1) x (n+1) system with the Lagrange row, estimate around
 * the LOCAL mean of the neighbours (_krige.py:5-44).  GSM_KRIGING_SIMPLE: sk_solve, Sigma w = rho, estimate
 * global_mean + sum w (v - global_mean) (_krige.py:46-81); global_mean [dev, n_chains] = the mean of each chain's conditioning
 * values BEFORE the call (MCMC.py:81, np.mean(out_grid[cond_msk])), read when the blocks are simulated.
 * Replaces: the ktype argument of MCMC.sgs (MCMC.py:91, :158-161). */
#define GSM_KRIGING_ORDINARY 0
#define GSM_KRIGING_SIMPLE 1
int gsm_sgs_set_kriging(gsm_handle h, int32_t ktype, const double* global_mean);

/* One batch of small-scale iterations in ONE call: for j < n_iters, in the order of chain_sgs.run's loop body
 * (MCMC.py:1741-1822) -- [gsm_qt_transform cur -> next] gsm_sgs_blocks_batch [gsm_sgs_finish | gsm_qt_transform next ->
 * proposed, gsm_sgs_loss, gsm_sgs_decide, gsm_sgs_commit(_map)] -- with iteration j's draws at windows + 4*n_chains*j,
 * cell_off + cell_off_stride*j, cell_cnt + n_chains*j (or NULL), u + n_chains*j and records at loss_rec + j / acc_rec + j
 * (rec_stride = n_iters).  cell_base (HOST, n_iters entries, or NULL = 0): iteration j's cells / z start at
 * cells + 2*cell_base[j] / z + cell_base[j] (the tightly packed lists of replay mode).
 * use_graph != 0 (and cell_base == NULL, stream != NULL): the launch sequence is captured into a hipGraph the second time the
 * SAME batch (every byte of the struct -- zero it before filling -- and n_iters) is submitted and replayed afterwards: one
 * launch per batch instead of up to 7 per iteration; the device buffers must then be static and refilled in place
 * (gsm_sgs_draw_philox / gsm_sgs_draw_pcg64 do).  Asynchronous; gsm_sgs_check reports a raised device flag.
 * grid_finite != 0: the caller's promise that no cell of `cur` is NaN (it stays so: simulation fills every cell of a block).
 * A cell's neighbour set and kriging weights depend on WHERE values are, not on the values; with the promise the search reads no
 * grid value, the records name their cells, and the records of iteration j + 1 are made on a second stream of the handle while
 * iteration j runs its value pass, transforms, loss and decision (the longest kernel of an iteration leaves the critical path;
 * the value pass then reads the neighbours' values from the grid).  Same numbers as without the promise.  With NaN cells in
 * `cur` the promise is false and the neighbour sets would be wrong: leave it 0.
 * gsm_sgs_graph_replays: how many batches of this handle were graph launches (diagnostics / tests). */
typedef struct gsm_sgs_batch {
  double* cur; double* next; double* proposed;            /* proposed: only with a transformer */
  const double* zcond; const double* trend;               /* trend NULL: not detrended */
  const double* qt_quantiles; const double* qt_references;
  double* energy; double* state;                          /* windowed finish only */
  const double* x_axis; const double* y_axis; const double* lag_cov;
  const int32_t* windows; const int32_t* cell_off; const int32_t* cell_cnt; const int32_t* cells; const double* z;
  const int64_t* cell_base;                               /* HOST or NULL */
  const double* u;
  uint32_t* resampled; double* loss; int32_t* bad; double* loss_prev; uint8_t* accept; double* loss_rec; uint8_t* acc_rec;
  double radius, sill;
  int64_t cell_off_stride;
  int32_t qt_n, windowed, lag_mi, lag_mj, hw, num_points, max_cells, use_graph;
  int32_t grid_finite;                                    /* the caller's promise: no NaN in cur (see below) */
} gsm_sgs_batch;
int gsm_sgs_iterate(gsm_handle h, const gsm_sgs_batch* batch, int32_t n_iters, void* stream);
int gsm_sgs_graph_replays(gsm_handle h);

/* Setup-time distance transform: dist[i] = Euclidean distance from cell i (coordinates xx[i], yy[i]) to the nearest
 * cell with mask[i] != 0, all [dev, H*W].  Exact (brute force over the masked cells, same dx*dx + dy*dy, sqrt
 * arithmetic as the KD-tree query).  Synchronises the stream.
 * Replaces: Utilities.min_dist_from_mask (Utilities.py:21-24), used by RandField.get_crf_weight /
 * chain_crf.set_crf_data_weight (MCMC.py:689-714, :1124-1134). */
int gsm_min_dist_from_mask(gsm_handle h, const double* xx, const double* yy, const uint8_t* mask, double* dist,
                           void* stream);

/* Diagnostics: stream-copy n doubles src -> dst [dev] with the step kernel's access shape (8 bytes per lane,
 * coalesced).  A known byte count for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM). */
int gsm_debug_stream_copy(const double* src, double* dst, int64_t n, void* stream);

/* Diagnostics: per-chain cycle totals of the step kernel's phases (8 x uint64 per chain, host buffer) from the last
 * launch.  Only in a library built with GSM_STAMPS=1 (a diagnostic build: the stamps cost cycles); the production build
 * returns GSM_E_UNSUPPORTED. */
int gsm_debug_stamps(uint64_t* out, int32_t n_chains);
/* Same for the fused chain kernel (16 x uint64 per chain). */
int gsm_debug_stamps_fused(uint64_t* out, int32_t n_chains);

/* Test hook: the device's standard-normal pairs (Philox4x32-10 counters (idx, stream, step), Box-Muller with the table-driven
 * log / sincos of the coefficient phase) for idx = idx0 .. idx0 + n - 1: out[2 i], out[2 i + 1] [dev].  Checked against
 * oracle/philox_oracle.normals2 (numpy log / sqrt / cos / sin). */
int gsm_debug_normals(uint64_t seed, int64_t step, uint32_t stream_id, uint32_t idx0, int32_t n, double* out, void* stream);

/* Test hook: one Philox4x32-10 block on the host (ctr[4], key[2] -> out[4]); the device generator uses
 * the same inline function.  Checked against the Random123 known-answer vectors. */
int gsm_philox_selftest(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4);

/* sizeof of a struct of this header as the library was compiled: which = 0 gsm_rf_params, 1 gsm_sgs_batch, 2 gsm_vario; -1 for any other value.
 * For bindings that restate the structs (ctypes, cgo ...): a mismatch is a layout error that would otherwise corrupt silently. */
int gsm_struct_size(int32_t which);

#ifdef __cplusplus
}
#endif
#endif /* GSM_H */
