// Internal declarations shared by the translation units of libgsm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/gsm.h"

namespace gsm {

// hipFuncSetAttribute is per device: one flag per (call site, device) -- `flags` is the call site's static array
constexpr int kMaxDevices = 64;
inline bool attr_needed_on_this_device(bool (&flags)[kMaxDevices], int& dev) {
  dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) { dev = -1; return true; }
  return !flags[dev];
}

// Static fields shared by every chain of a handle (device pointers, H*W each, row-major).
struct StaticFields {
  const double* surf;
  const double* velx;
  const double* vely;
  const double* dhdt;
  const double* smb;
  const double* weight;   // nullptr => block_type 'RF'
  const uint8_t* upd;     // update / guard / resampled mask
  const uint8_t* mc;      // loss mask
  // packed copies for the step kernel's stencil: one 16-byte load per neighbour
  const double2* svx;     // (surf, velx)
  const double2* svy;     // (surf, vely)
  const double2* ds;      // (dhdt, smb)
  // folded operands of the flux step kernel (step_flux_kernel.hip): masks encoded as NaNs
  const double2* sA;      // (crf weight where update_mask else tagged NaN, surf)
  const double2* sB;      // (velx, vely)
  const double2* sC;      // (dhdt where mc_mask == 1 else NaN, smb)
  int H, W;
  double res;             // grid spacing h
  double two_res;         // 2.0 * h  (np.gradient interior denominator)
  double rcp_res;         // RN(1/h), RN(1/(2h)): used when fast_div
  double rcp_two_res;
  int fast_div;           // 1: x/h and x/(2 sigma^2) as fma-corrected reciprocal multiplies (bit-identical to IEEE division)
  double two_sigma2;      // 2 * sigma_mc**2
  double rcp_two_sigma2;  // RN(1/(2 sigma^2))
};

struct BlockTable {
  const int32_t* bh;      // device, n_sizes
  const int32_t* bw;      // device, n_sizes
  const double* masks;    // device, packed edge masks
  const int64_t* mask_off;  // device, n_sizes
  const double* mask1d;   // device [n_sizes][64] or nullptr: the edge masks as a function of the distance to the block's border,
                          // mask[y][x] = mask1d[size][min(y, bh - 1 - y, x, bw - 1 - x)] -- set when every mask has that form
                          // (get_edge_masks, MCMC.py:583-621: a logistic function of the distance to the nearest border cell)
  int n_sizes;
  int max_bh, max_bw;
};

struct StepArgs {
  StaticFields S;
  BlockTable B;
  int n_chains, n_steps;
  int tile_cap;           // doubles available for the LDS tile
  void* beds;             // [n_chains][H][W] double, or float when f32_state
  void* energy;           // [n_chains][H][W] masked squared residual of the current bed (0 where not counted)
  int f32_state;
  uint32_t* resampled;
  double* loss_sum;       // [n_chains][2]
  const int32_t* size_idx;
  const int32_t* centre;
  const double* u;
  const double* fields;
  int64_t field_stride;
  double* loss;
  uint8_t* accept;
  int32_t* blocks;        // optional [n_chains*n_steps*4] (row, col, bh, bw) or nullptr
  int64_t rec_stride;     // records per chain in the per-step arrays (== n_steps of the whole call)
  int64_t rec_offset;     // first record of this launch within a chain's row
  int64_t in_stride;      // records per chain in the proposal arrays (size_idx, centre, u, fields)
  int32_t* err_flag;      // device int, set non-zero on bad device data
  int strip;              // 1: the strip kernels (chain_strip_kernel.hip) take this block table; decided once per table by
                          // strip_table_ok so that the fused kernel and the replay kernel of a handle always pair up
};

// per-proposal scalars written by propose_scalars_kernel and read (uniformly) by propose_kernel
struct PropScalars {
  double scale, nug, range_x, range_y, u;
  double aa, m_const, m_kappa;   // spectral-amplitude parameters (MCMC.py:209-239); m_const = log(Matern constant) / 2
  int32_t si, row, col, bh;
  int32_t bw, fy_off, g_off, pad;   // block shape and DFT-table offsets: no dependent table look-ups in propose_kernel;
                                    // pad = offset of this shape's k^2 table in ProposeArgs::k2tab
  int64_t mask_off;
  // magic reciprocals (0xFFFFFFFF / d + 1) of the three divisors of a step -- half-plane columns bw/2 + 1, operand
  // columns M1, width of the window's halo tile -- so that no workgroup derives them with a uniform division
  uint32_t m_nc, m_m1, m_tw, reserved;
  // strip kernel (chain_strip_kernel.hip): the DFT operand tables are 1-D twiddle tables in LDS, indexed by (k * j) mod n:
  // magic reciprocals of the block height / width and the offsets of their [cos | sin](2 pi m / n) tables in ProposeArgs::tab1d
  uint32_t m_bh, m_bw;
  int32_t t1h_off, t1w_off;
};

struct ProposeArgs {
  BlockTable B;
  gsm_rf_params rf;
  int H, W;
  int n_chains, n_steps;
  int64_t step0;
  const uint64_t* seeds;
  const int32_t* centres;  // flat cell ids
  int n_centres;
  int32_t* size_idx;
  int32_t* centre;
  double* u;
  double* fields;
  int64_t field_stride;
  double* rf_scalars;      // optional
  // DFT operand tables (device), built by gsm_set_blocks; see proposal_kernel.hip for the shapes
  const double* tables;
  int tables_len;          // doubles in `tables`
  int tab_max;             // largest single table pair (2*KR*NR or 2*Kc*M1 doubles) over the block table
  const int32_t* fy_off;   // indexed by block height n: offset of [cos | sin](2 pi ky y / n), each [K1][N1]
  const int32_t* g_off;    // indexed by block width n: offset of the folded c2r table G, [K2][N2]
  int lds_sx, lds_st;      // LDS row strides of X and T^T (== 16 mod 32 doubles: conflict-free fragment reads)
  int lds_x_half;          // doubles per X plane (re or im)
  int lds_tt;              // doubles of T^T
  const double* mathtab;   // device copy of the table of math_tables.h (kMathTabDoubles doubles)
  int lds_main;            // max(4 * lds_x_half, lds_tt): T^T overlays X
  int tiles1_max, tiles2_max;  // largest stage-1 / stage-2 output-tile counts over the block table
  const double* tab1d;     // per distinct block length n: cos(2 pi m / n), m < n, then sin(2 pi m / n)  (the values of `tables`)
  const int32_t* t1_off;   // indexed by the length n: offset of its pair in tab1d
  const double* k2tab;     // (sqrt(kx^2 + ky^2) + 1e-10)^2 on ky <= bh/2, kx <= bw/2, one [nrow][ncol] table per block size
  const int32_t* k2_off;   // [n_sizes] offsets into k2tab
  PropScalars* scalars;    // device scratch, n_chains * n_steps records
  int dbg;                 // diagnostics only (GSM_PROPOSE_DBG): bit0 cheap coefficients, bit1 skip stage 1, bit2 skip stage 2
  int parseval;            // the field's variance from the spectrum (Parseval) instead of from the field: same handles as split2 (the sum's order is that of
                           // 512-thread workgroups; proposal_device.h: coef_items, standardise)
  int split2;              // stage 2 of even block widths split by the parity of kx where the shape allows it (handles on the strip kernels: every kernel of
                           // such a handle holds two tile slots per wave; the flux-tile fused kernel holds one) -- proposal_device.h: prop_geom
};

// fused chain kernel (chain_fused_kernel.hip): step arguments + proposal arguments of the same call
struct FusedArgs {
  StepArgs T;
  ProposeArgs P;
  int work_len, fld_len;   // LDS region sizes in doubles, set by launch_chain_fused
  // gsm_run_noise (chain_strip_kernel<NOISE>): caller-supplied white noise instead of Philox draws -- record r's planes at
  // r * noise_stride (row-major (bh, bw)); noise_nug NULL: no nugget term
  const double* noise_re; const double* noise_im; const double* noise_nug;
  int64_t noise_stride;
};

// scratch + factor table of the Cholesky proposal generator (cholesky_kernel.hip)
struct CholArgs {
  int n_classes, n_groups;         // n_groups = n_sizes * n_classes
  const double* const* factors;    // device array [n_groups] of U matrices
  int* counts;                     // [n_groups]
  int* rec_off;                    // [n_groups + 1]
  int* tile_off;                   // [n_groups + 1]
  int* work_off;                   // [n_groups + 1] prefix of 64 x 128 output tiles (p-tiles x n-tiles) per group
  int64_t* z_off;                  // [n_groups]
  int* group_of;                   // [n_rec]
  int* order;                      // [n_rec]
  double* scale;                   // [n_rec]
  double* zbuf;                    // sum over groups of Npad * ppad doubles
};

// gsm_draw_pcg64 (pcg64_kernel.hip): NumPy's PCG64 generator streams of the reference's chains, on the device
struct PcgDrawArgs {
  int H, W, n_chains, n_steps, n_sizes;
  gsm_rf_params rf;
  const int32_t* bh; const int32_t* bw;          // block table
  uint64_t* rf_state;                            // [n_chains*6] state lo, hi, inc lo, hi, has_uint32, uinteger (in/out)
  uint64_t* ch_state;                            // same for the chain's generator
  const uint8_t* region_mask;                    // [H*W] or nullptr (update_in_region False)
  const uint64_t* jump; const uint64_t* zig;     // device copies of the constant tables
  int32_t* size_idx; int32_t* centre; double* u; double* rf_scalars;
  double* noise_re; double* noise_im; double* nugget;   // nugget may be nullptr (draws consumed, nothing stored)
  int64_t field_stride;
  int32_t* err;
};
hipError_t launch_pcg64_draw(const PcgDrawArgs& a, hipStream_t st);
constexpr int kPcgJumpWords = 4 * 512;          // = 4 * pcg::kJump (pcg64_device.h; checked there where both are visible)
void pcg64_host_tables(uint64_t* jump_out, const uint64_t** zig_out);
struct SgsDrawArgs;
hipError_t launch_sgs_draw_pcg64(const SgsDrawArgs& a, uint64_t* states, const uint64_t* jump, const uint64_t* zig, hipStream_t st);

// small-scale chain: sequential Gaussian simulation of one block per chain (sgs_kernel.hip)
struct SgsCellHdr {          // one per (chain, cell slot), written by sgs_weights_kernel
  int32_t n;                 // neighbours; -1: the cell holds conditioning data; -2: cell outside its window; 0: error
  int32_t op;                // block-local index of the cell
  double sdz;                // sqrt(|kriging variance|) * the cell's standard normal
  double var;                // |kriging variance|
  double c1;                 // (1 - sum of the kriging weights) / n
};
struct SgsArgs {
  int H, W, n_chains;
  double* grid;            // [n_chains][H*W] in/out: the conditioning grid; the window's cells are rewritten
  const double* zcond;     // [H*W] or nullptr: initial content of the window cells (conditioning data, NaN elsewhere)
  const int32_t* win;      // [n_chains*4] r0, r1, c0, c1 of the simulated block
  const double* xs;        // [W] x coordinate of every column
  const double* ys;        // [H] y coordinate of every row
  const double* lag;       // [(2 mi + 1) * (2 mj + 1)] covariance at integer lag (di, dj), |di| <= mi, |dj| <= mj
  int hw, mi, mj, num_points;
  int ktype;               // 0 ordinary kriging (ok_solve), 1 simple kriging (sk_solve) with gmean
  int defer;               // 1: every cell outside the blocks holds a value (the caller's promise); sgs_weights_kernel then reads no grid value
                           //    -- the records name the cells -- and can run while the grid of the iteration before is still being written
  const double* gmean;     // [n_chains] global mean of each chain's conditioning values (simple kriging only)
  double radius, sill;
  const int32_t* cell_off; // [n_chains+1] (or [n_chains] with cell_cnt)
  const int32_t* cell_cnt; // optional [n_chains]: number of cells of each chain
  const int32_t* cells;    // [total*2] (i, j) in simulation order
  const double* z;         // [total] standard normals
  double* trace;           // optional [total*3]: (neighbours, estimate, variance)
  int32_t* nbr_trace;      // optional [total*48]: the chosen neighbours (flat cell index, -1 padded) of every simulated cell
  int32_t* err;
  // scratch of the handle: visiting ranks of the block cells and one record per (chain, cell slot)
  int max_cells;           // record stride per chain (>= the largest cell count of the call)
  int32_t* rank;           // [n_chains][1024]
  int32_t* rank_ok;        // [n_chains]
  SgsCellHdr* rec_hdr;     // [n_chains*max_cells]
  double2* rec_vw;         // [n_chains*max_cells/64][48][64]: entry e of 64 consecutive cell slots side by side -- (the neighbour's value,
                           //  or NaN-boxed (visiting slot << 16 | block-local index) where the neighbour is a cell simulated earlier in
                           //  the block; its kriging weight)
};
hipError_t launch_sgs_blocks(const SgsArgs& a, int launch_cells, hipStream_t st);
hipError_t launch_sgs_weights(const SgsArgs& a, int launch_cells, hipStream_t st);     // ranks + records
hipError_t launch_sgs_sequence(const SgsArgs& a, hipStream_t st);                      // value pass
struct SgsDrawArgs {
  int H, W, n_chains, n_iters;
  int64_t iter0;
  const uint64_t* seeds;
  const uint8_t* region_mask;   // nullable
  const uint8_t* is_data;
  int min_x, max_x, min_y, max_y, max_cells;
  const double* mathtab;
  int32_t* win; int32_t* blk; int32_t* cell_off; int32_t* cell_cnt; int32_t* cells; double* z; double* u;
  int32_t* err;
};
hipError_t launch_sgs_draw(const SgsDrawArgs& a, hipStream_t st);
int sgs_loss_parts(const StaticFields& S);      // workgroups per chain of the loss kernel; scratch = n_chains * parts doubles + ints
hipError_t launch_sgs_loss(const StaticFields& S, int n_chains, const double* beds, const double* trend, double* loss, int32_t* bad,
                           double* part_sum, int32_t* part_bad, hipStream_t st);
hipError_t launch_sgs_loss_tail(const StaticFields& S, int n_chains, const double* trend, double* part_sum, int32_t* part_bad, int32_t* ticket,
                                double* loss, int32_t* bad, const double* u, double* loss_prev, uint8_t* accept, double* loss_rec,
                                uint8_t* acc_rec, int64_t rec_stride, int mode, double* cur, double* beds, uint32_t* resampled,
                                const int32_t* win, hipStream_t st, const double* qt_q = nullptr, const double* qt_ref = nullptr, int qt_n = 0,
                                double clip_min = 0.0, double clip_max = 0.0, double* next = nullptr, double* next_acc = nullptr);
bool sgs_tail_takes_qt(const StaticFields& S, int nq);
hipError_t launch_sgs_state_init(const StaticFields& S, int n_chains, const double* beds, const double* trend, double* energy, double* state,
                                 hipStream_t st);
hipError_t launch_sgs_finish(const StaticFields& S, int n_chains, double* cur, double* next, const double* trend, double* energy, double* state,
                             const int32_t* win, const double* u, uint32_t* resampled, uint8_t* accept, double* loss_rec, uint8_t* acc_rec,
                             int64_t rec_stride, int32_t* err, hipStream_t st);
hipError_t launch_sgs_decide(int n_chains, const double* loss_next, const int32_t* bad, const double* u, double* loss_prev,
                             uint8_t* accept, double* loss_rec, uint8_t* acc_rec, int64_t rec_stride, hipStream_t st);
hipError_t launch_qt(const double* quantiles, const double* references, int nq, double clip_min, double clip_max, const double* x,
                     double* out, int64_t n, int inverse, hipStream_t st);
hipError_t launch_sgs_commit_map(int H, int W, int n_chains, double* cur, const double* proposed, uint32_t* resampled, const int32_t* win,
                                 const uint8_t* accept, hipStream_t st);
hipError_t launch_sgs_commit(int H, int W, int n_chains, double* cur, double* next, uint32_t* resampled, const int32_t* win,
                             const uint8_t* accept, hipStream_t st);

// launchers (defined next to their kernels)
hipError_t launch_propose_cholesky(const ProposeArgs& a, const CholArgs& c, hipStream_t st);
hipError_t launch_cholesky_upper(double* A, int n, int ld, double jitter, int* d_info, hipStream_t st);
hipError_t launch_cov_assemble(int bh, int bw, double res, const gsm_vario& v, const double* lag_table, double* sigma,
                               int ld, hipStream_t st);
hipError_t launch_step(const StepArgs& a, hipStream_t st);
hipError_t launch_step_flux(const StepArgs& a, hipStream_t st);
bool step_flux_supported(const StepArgs& a);
int debug_read_stamps(unsigned long long* out, int n_chains);
int debug_read_stamps_fused(unsigned long long* out, int n_chains);
hipError_t launch_pack_flux_static(const StaticFields& S, double2* sA, double2* sB, double2* sC, hipStream_t st);
hipError_t launch_init_loss(const StaticFields& S, int n_chains, const void* beds, void* energy, int f32_state,
                            double* loss_sum, double* loss0, hipStream_t st);
hipError_t launch_pack_static(const StaticFields& S, double2* svx, double2* svy, double2* ds, hipStream_t st);
hipError_t launch_residual(const StaticFields& S, int n_chains, const double* beds, double* out, hipStream_t st);
hipError_t launch_propose(const ProposeArgs& a, hipStream_t st);
hipError_t launch_propose_scalars(const ProposeArgs& a, hipStream_t st);
hipError_t launch_debug_normals(uint64_t seed, int64_t step, uint32_t stream_id, uint32_t idx0, int n, const double* mathtab, double* out,
                                hipStream_t st);
hipError_t launch_k2_tables(const BlockTable& B, const int32_t* k2_off, double resolution, double* k2tab, hipStream_t st);
hipError_t launch_spectral_from_noise(const ProposeArgs& a, const int32_t* size_idx, const double* rf_scalars, const double* noise_re,
                                      const double* noise_im, const double* nugget_field, hipStream_t st);
hipError_t launch_chain_fused(const FusedArgs& a, hipStream_t st);
hipError_t launch_chain_strip(const FusedArgs& a, hipStream_t st);
hipError_t launch_chain_strip_noise(const FusedArgs& a, hipStream_t st);
hipError_t launch_noise_chain_scalars(const ProposeArgs& a, const int32_t* size_idx, const int32_t* centre, const double* u,
                                      const double* rf_scalars, int32_t* err_flag, hipStream_t st);
hipError_t launch_step_strip(const StepArgs& a, hipStream_t st);
hipError_t launch_resampled_from_records(const FusedArgs& a, hipStream_t st);
bool strip_table_ok(const StaticFields& S, const BlockTable& B, int lds_main, int tiles1_max, int tiles2_max);
constexpr int kMask1D = 64;    // entries per block size of BlockTable::mask1d
bool fused_supported(const FusedArgs& a);
int propose_max_tiles_per_wave();
int propose_max_tiles1_per_wave();
int propose_waves();
size_t step_lds_bytes(int tile_cap);
hipError_t launch_min_dist(const double* xx, const double* yy, const uint8_t* mask, int n, double2* pts, int* count,
                           double* dist, hipStream_t st);
hipError_t launch_stream_copy(const double* src, double* dst, int64_t n, hipStream_t st);

}  // namespace gsm
