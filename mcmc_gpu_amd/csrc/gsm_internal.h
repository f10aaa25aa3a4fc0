// Internal declarations shared by the translation units of libgsm_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>
#include "../../include/gsm.h"

namespace gsm {

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
  int H, W;
  double res;             // grid spacing h
  double two_res;         // 2.0 * h  (np.gradient interior denominator)
  double two_sigma2;      // 2 * sigma_mc**2
};

struct BlockTable {
  const int32_t* bh;      // device, n_sizes
  const int32_t* bw;      // device, n_sizes
  const double* masks;    // device, packed edge masks
  const int64_t* mask_off;  // device, n_sizes
  int n_sizes;
  int max_bh, max_bw;
};

struct StepArgs {
  StaticFields S;
  BlockTable B;
  int n_chains, n_steps;
  int tile_cap;           // doubles available for the LDS tile
  double* beds;
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
  const double* twiddle;   // device: cos/sin tables per distinct length (see proposal_kernel.hip)
  const int32_t* tw_off;   // device: offset (doubles) of the table of length n, indexed by n (0..max)
};

// launchers (defined next to their kernels)
hipError_t launch_step(const StepArgs& a, hipStream_t st);
hipError_t launch_init_loss(const StaticFields& S, int n_chains, const double* beds, double* loss_sum,
                            double* loss0, hipStream_t st);
hipError_t launch_residual(const StaticFields& S, int n_chains, const double* beds, double* out, hipStream_t st);
hipError_t launch_propose(const ProposeArgs& a, hipStream_t st);
size_t step_lds_bytes(int tile_cap);

}  // namespace gsm
