// C ABI of libgsm_hip.so (see include/gsm.h for the contract and the reference interfaces replaced).
#include "gsm_internal.h"
#include "math_tables.h"
#include "normal_score.h"
#include <math.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>

using namespace gsm;

struct gsm_context {
  int H = 0, W = 0, n_chains = 0, device = 0, f32_state = 0;
  std::string err;
  bool have_static = false, have_blocks = false, have_centres = false;
  // owned device copies
  double* d_static[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // surf velx vely dhdt smb weight
  uint8_t* d_upd = nullptr;
  uint8_t* d_mc = nullptr;
  double2 *d_svx = nullptr, *d_svy = nullptr, *d_ds = nullptr;
  double2 *d_sA = nullptr, *d_sB = nullptr, *d_sC = nullptr;
  StaticFields S{};
  int32_t *d_bh = nullptr, *d_bw = nullptr;
  int64_t* d_mask_off = nullptr;
  double* d_masks = nullptr;
  double* d_mask1d = nullptr;    // [n_sizes][kMask1D]: the masks as a function of the distance to the block border, if they are one
  double* d_tables = nullptr;
  int32_t *d_fy_off = nullptr, *d_g_off = nullptr;
  double* d_tab1d = nullptr; int32_t* d_t1_off = nullptr;   // 1-D twiddle tables of the strip kernel's DFT stages
  int lds_sx = 0, lds_st = 0, lds_x_half = 0, lds_tt = 0, prop_tiles = 0, prop_tiles1 = 0;
  int tables_len = 0, tab_max = 0;
  double* d_k2 = nullptr;        // per-size k^2 tables of the spectral amplitude (depend on rf.resolution)
  double* d_mathtab = nullptr;   // log / sincos table of the coefficient phase (math_tables.h)
  double* d_sgs_part_sum = nullptr; int32_t* d_sgs_part_bad = nullptr; int32_t* d_sgs_ticket = nullptr; size_t sgs_part_cap = 0;   // gsm_sgs_loss partial sums
  // gsm_sgs_blocks scratch: visiting ranks + one record per (chain, cell slot), see SgsArgs
  static constexpr int kSgsDepth = 8;                                            // sets of record scratch of an overlapped batch (iteration j uses set j mod depth)
  void* d_sgs_rec[kSgsDepth] = {}; size_t sgs_rec_cells[kSgsDepth] = {};
  double* d_sgs_next_acc = nullptr; size_t sgs_next_acc_cap = 0;                 // T(proposed) of every chain (sgs_loss_tail_kernel<true>)
  hipStream_t sgs_side = nullptr, sgs_side2 = nullptr; hipEvent_t sgs_ev[kSgsDepth + 2] = {};         // gsm_sgs_iterate's second stream (records of later iterations beside the current one)
  // gsm_sgs_iterate: the captured launch sequence of one batch (hipGraph), keyed by the bytes of its gsm_sgs_batch + n_iters
  int sgs_ktype = 0; const double* sgs_gmean = nullptr;        // gsm_sgs_set_kriging
  std::vector<char> sgs_graph_key; hipGraphExec_t sgs_graph_exec = nullptr; int sgs_graph_replays = 0;
  uint64_t* d_pcg_tab = nullptr;   // gsm_draw_pcg64: LCG jump table (kPcgJumpWords) + ziggurat tables (768 words)
  int32_t* d_k2_off = nullptr;
  double k2_resolution = 0.0;
  PropScalars* d_scalars[2] = {nullptr, nullptr};
  size_t scalars_cap[2] = {0, 0};
  // Cholesky generator
  int n_classes = 0;
  const double** d_factors = nullptr;
  struct CholScratch { int* ints = nullptr; int64_t* zoff = nullptr; int* per_rec = nullptr; double* scale = nullptr;
                       double* zbuf = nullptr; size_t recs = 0; int groups = 0; } chol[2];
  BlockTable B{};
  int tile_cap = 0;
  int32_t* d_centres = nullptr;
  int n_centres = 0;
  int32_t* d_err = nullptr;
  // philox-mode scratch (two buffers)
  struct Scratch {
    int32_t* size_idx = nullptr;
    int32_t* centre = nullptr;
    double* u = nullptr;
    double* fields = nullptr;
    size_t recs = 0;
  } scr[2];
  int64_t field_stride = 0;
  hipStream_t aux = nullptr;
  hipEvent_t ev_prop[2] = {nullptr, nullptr}, ev_step[2] = {nullptr, nullptr};
  // timing
  bool timing = false;
  int last_fused = 0;     // 1 when the last gsm_run_philox call went through the fused chain kernel
  int use_fused = -1;     // -1: decide from GSM_FUSED at the first gsm_run_philox; 0 / 1: set by gsm_set_fused
  double t_step_ms = 0, t_prop_ms = 0;
  int n_step_launch = 0, n_prop_launch = 0;
};

static thread_local std::string g_create_err;
static constexpr int kFusedSegment = 4096;   // steps per launch of the fused chain kernel

static int fail(gsm_handle h, int code, const std::string& msg) {
  if (h) h->err = msg; else g_create_err = msg;
  return code;
}
#define HIPCHK(h, expr)                                                                       \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess)                                                                     \
      return fail(h, GSM_E_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));           \
  } while (0)

extern "C" const char* gsm_last_error(gsm_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

extern "C" int gsm_create(gsm_handle* out, int32_t H, int32_t W, int32_t n_chains, int32_t dtype, int32_t device) {
  if (!out) return fail(nullptr, GSM_E_ARG, "gsm_create: out is NULL");
  *out = nullptr;
  if (H < 3 || W < 3 || n_chains < 1) return fail(nullptr, GSM_E_ARG, "gsm_create: need H,W >= 3 and n_chains >= 1");
  if ((int64_t)H * W > (1LL << 30)) return fail(nullptr, GSM_E_ARG, "gsm_create: grid too large");
  if (dtype != 0 && dtype != 1) return fail(nullptr, GSM_E_UNSUPPORTED, "gsm_create: dtype must be 0 (fp64 state) or 1 (fp32 state)");
  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev < 1)
    return fail(nullptr, GSM_E_HIP, std::string("gsm_create: no HIP device: ") + hipGetErrorString(e));
  if (device < 0 || device >= ndev) return fail(nullptr, GSM_E_ARG, "gsm_create: device index out of range");
  e = hipSetDevice(device);
  if (e != hipSuccess) return fail(nullptr, GSM_E_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  gsm_context* c = new gsm_context();
  c->H = H; c->W = W; c->n_chains = n_chains; c->device = device; c->f32_state = dtype;
  e = hipMalloc(&c->d_err, sizeof(int32_t));
  if (e == hipSuccess) e = hipMemset(c->d_err, 0, sizeof(int32_t));
  if (e != hipSuccess) { delete c; return fail(nullptr, GSM_E_HIP, std::string("hipMalloc: ") + hipGetErrorString(e)); }
  *out = c;
  return GSM_OK;
}

static void free_scratch(gsm_context* c) {
  for (auto& s : c->scr) {
    if (s.size_idx) hipFree(s.size_idx);
    if (s.centre) hipFree(s.centre);
    if (s.u) hipFree(s.u);
    if (s.fields) hipFree(s.fields);
    s = gsm_context::Scratch();
  }
}

extern "C" int gsm_destroy(gsm_handle h) {
  if (!h) return GSM_OK;
  hipSetDevice(h->device);
  for (auto& p : h->d_static) if (p) hipFree(p);
  if (h->d_upd) hipFree(h->d_upd);
  if (h->d_mc) hipFree(h->d_mc);
  if (h->d_svx) hipFree(h->d_svx);
  if (h->d_svy) hipFree(h->d_svy);
  if (h->d_ds) hipFree(h->d_ds);
  if (h->d_tab1d) hipFree(h->d_tab1d);
  if (h->d_t1_off) hipFree(h->d_t1_off);
  if (h->d_sA) hipFree(h->d_sA);      // d_sB, d_sC point into the same allocation
  if (h->d_bh) hipFree(h->d_bh);
  if (h->d_bw) hipFree(h->d_bw);
  if (h->d_mask_off) hipFree(h->d_mask_off);
  if (h->d_masks) hipFree(h->d_masks);
  if (h->d_mask1d) hipFree(h->d_mask1d);
  if (h->d_tables) hipFree(h->d_tables);
  if (h->d_fy_off) hipFree(h->d_fy_off);
  if (h->d_g_off) hipFree(h->d_g_off);
  for (auto& p : h->d_scalars) if (p) hipFree(p);
  if (h->d_k2) hipFree(h->d_k2);
  if (h->d_mathtab) hipFree(h->d_mathtab);
  if (h->d_sgs_part_sum) { hipFree(h->d_sgs_part_sum); hipFree(h->d_sgs_part_bad); hipFree(h->d_sgs_ticket); }
  for (void* q : h->d_sgs_rec) if (q) hipFree(q);
  for (hipEvent_t e : h->sgs_ev) if (e) hipEventDestroy(e);
  if (h->d_sgs_next_acc) hipFree(h->d_sgs_next_acc);
  if (h->sgs_side) hipStreamDestroy(h->sgs_side);
  if (h->sgs_side2) hipStreamDestroy(h->sgs_side2);
  if (h->sgs_graph_exec) hipGraphExecDestroy(h->sgs_graph_exec);
  if (h->d_pcg_tab) hipFree(h->d_pcg_tab);
  if (h->d_k2_off) hipFree(h->d_k2_off);
  if (h->d_factors) hipFree(h->d_factors);
  for (auto& c : h->chol) { if (c.ints) hipFree(c.ints); if (c.zoff) hipFree(c.zoff); if (c.per_rec) hipFree(c.per_rec);
                            if (c.scale) hipFree(c.scale); if (c.zbuf) hipFree(c.zbuf); }
  if (h->d_centres) hipFree(h->d_centres);
  if (h->d_err) hipFree(h->d_err);
  free_scratch(h);
  for (int i = 0; i < 2; ++i) {
    if (h->ev_prop[i]) hipEventDestroy(h->ev_prop[i]);
    if (h->ev_step[i]) hipEventDestroy(h->ev_step[i]);
  }
  if (h->aux) hipStreamDestroy(h->aux);
  delete h;
  return GSM_OK;
}

template <class T>
static hipError_t dup_device(T** dst, const T* src, size_t n, hipStream_t st) {
  if (*dst) { hipFree(*dst); *dst = nullptr; }
  hipError_t e = hipMalloc(dst, n * sizeof(T));
  if (e != hipSuccess) return e;
  return hipMemcpyAsync(*dst, src, n * sizeof(T), hipMemcpyDefault, st);
}

extern "C" int gsm_set_static(gsm_handle h, const double* surf, const double* velx, const double* vely,
                              const double* dhdt, const double* smb, const double* crf_weight,
                              const uint8_t* update_mask, const uint8_t* mc_mask, double resolution,
                              double sigma_mc, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!surf || !velx || !vely || !dhdt || !smb || !update_mask || !mc_mask)
    return fail(h, GSM_E_ARG, "gsm_set_static: NULL field");
  if (!(resolution > 0.0) || !(sigma_mc > 0.0)) return fail(h, GSM_E_ARG, "gsm_set_static: resolution and sigma_mc must be > 0");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  const size_t n = (size_t)h->H * h->W;
  const double* src[6] = {surf, velx, vely, dhdt, smb, crf_weight};
  for (int i = 0; i < 6; ++i) {
    if (!src[i]) { if (h->d_static[i]) { hipFree(h->d_static[i]); h->d_static[i] = nullptr; } continue; }
    HIPCHK(h, dup_device(&h->d_static[i], src[i], n, st));
  }
  HIPCHK(h, dup_device(&h->d_upd, update_mask, n, st));
  HIPCHK(h, dup_device(&h->d_mc, mc_mask, n, st));
  StaticFields& S = h->S;
  S.surf = h->d_static[0]; S.velx = h->d_static[1]; S.vely = h->d_static[2];
  S.dhdt = h->d_static[3]; S.smb = h->d_static[4]; S.weight = h->d_static[5];
  S.upd = h->d_upd; S.mc = h->d_mc;
  S.H = h->H; S.W = h->W;
  S.res = resolution;
  S.two_res = 2.0 * resolution;
  S.rcp_res = 1.0 / S.res;
  S.rcp_two_res = 1.0 / S.two_res;
  S.two_sigma2 = 2 * (sigma_mc * sigma_mc);
  S.rcp_two_sigma2 = 1.0 / S.two_sigma2;
  // exact_div() needs a correctly rounded reciprocal of a divisor whose significand is not all ones and
  // quotients far from the exponent limits; anything else takes the IEEE division path.
  {
    auto divisor_ok = [](double d) {
      uint64_t bits;
      memcpy(&bits, &d, sizeof(bits));
      const bool all_ones = (bits & 0xFFFFFFFFFFFFFull) == 0xFFFFFFFFFFFFFull;
      return !all_ones && d > 1e-100 && d < 1e100;
    };
    S.fast_div = (divisor_ok(S.res) && divisor_ok(S.two_sigma2)) ? 1 : 0;
  }
  if (!h->d_svx) {
    HIPCHK(h, hipMalloc(&h->d_svx, n * sizeof(double2)));
    HIPCHK(h, hipMalloc(&h->d_svy, n * sizeof(double2)));
    HIPCHK(h, hipMalloc(&h->d_ds, n * sizeof(double2)));
    HIPCHK(h, hipMalloc(&h->d_sA, 3 * n * sizeof(double2)));      // sA | sB | sC in one allocation (one buffer descriptor)
    h->d_sB = h->d_sA + n;
    h->d_sC = h->d_sA + 2 * n;
  }
  HIPCHK(h, launch_pack_static(S, h->d_svx, h->d_svy, h->d_ds, st));
  HIPCHK(h, launch_pack_flux_static(S, h->d_sA, h->d_sB, h->d_sC, st));
  S.svx = h->d_svx; S.svy = h->d_svy; S.ds = h->d_ds;
  S.sA = h->d_sA; S.sB = h->d_sB; S.sC = h->d_sC;
  HIPCHK(h, hipStreamSynchronize(st));
  h->have_static = true;
  return GSM_OK;
}

extern "C" int gsm_set_blocks(gsm_handle h, int32_t n_sizes, const int32_t* bh, const int32_t* bw,
                              const double* edge_masks_packed, const int64_t* mask_offsets, void* stream) {
  if (!h) return GSM_E_ARG;
  if (n_sizes < 1 || !bh || !bw) return fail(h, GSM_E_ARG, "gsm_set_blocks: empty block table");
  if (n_sizes > 64) return fail(h, GSM_E_UNSUPPORTED, "gsm_set_blocks: at most 64 block sizes");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  int max_bh = 0, max_bw = 0, cap = 0;
  int64_t mask_total = 0;
  for (int i = 0; i < n_sizes; ++i) {
    if (bh[i] < 2 || bw[i] < 2 || (bh[i] & 1) || (bw[i] & 1))
      return fail(h, GSM_E_ARG, "gsm_set_blocks: block sizes must be even and >= 2 (RandField.get_block_sizes makes them even)");
    if (bh[i] > h->H || bw[i] > h->W)
      return fail(h, GSM_E_ARG, "gsm_set_blocks: block larger than the grid (the reference's slices mismatch there)");
    max_bh = std::max(max_bh, bh[i]);
    max_bw = std::max(max_bw, bw[i]);
    cap = std::max(cap, (bh[i] + 2) * (bw[i] + 2));
    if (mask_offsets) mask_total = std::max<int64_t>(mask_total, mask_offsets[i] + (int64_t)bh[i] * bw[i]);
  }
  if (step_lds_bytes(cap) > 160 * 1024)
    return fail(h, GSM_E_UNSUPPORTED, "gsm_set_blocks: (bh+2)*(bw+2) window does not fit the 160 KiB LDS tile");
  HIPCHK(h, dup_device(&h->d_bh, bh, (size_t)n_sizes, st));
  HIPCHK(h, dup_device(&h->d_bw, bw, (size_t)n_sizes, st));
  if (edge_masks_packed && mask_offsets) {
    HIPCHK(h, dup_device(&h->d_mask_off, mask_offsets, (size_t)n_sizes, st));
    HIPCHK(h, dup_device(&h->d_masks, edge_masks_packed, (size_t)mask_total, st));
    // The reference's edge masks are a function of the distance to the nearest border cell of the block (get_edge_masks,
    // MCMC.py:583-621): mask[y][x] = T[min(y, bh - 1 - y, x, bw - 1 - x)].  Checked value for value here; if every mask of the
    // table has that form, the strip kernel reads the 64-entry T from LDS instead of 8 bytes per cell from a 51 KB table.
    if (h->d_mask1d) { hipFree(h->d_mask1d); h->d_mask1d = nullptr; }
    {
      std::vector<double> hm((size_t)mask_total);
      HIPCHK(h, hipMemcpyAsync(hm.data(), edge_masks_packed, sizeof(double) * (size_t)mask_total, hipMemcpyDefault, st));
      HIPCHK(h, hipStreamSynchronize(st));
      std::vector<double> t1((size_t)n_sizes * kMask1D, 0.0);
      bool one_d = true;
      for (int i = 0; i < n_sizes && one_d; ++i) {
        const double* m = hm.data() + mask_offsets[i];
        const int H = bh[i], W = bw[i];
        if ((std::min(H, W) - 1) / 2 >= kMask1D) { one_d = false; break; }
        std::vector<char> have(kMask1D, 0);
        for (int y = 0; y < H && one_d; ++y)
          for (int x = 0; x < W; ++x) {
            const int d = std::min(std::min(y, H - 1 - y), std::min(x, W - 1 - x));
            const double v = m[(size_t)y * W + x];
            double& t = t1[(size_t)i * kMask1D + d];
            if (!have[d]) { have[d] = 1; t = v; }
            else if (memcmp(&t, &v, sizeof(double)) != 0) { one_d = false; break; }
          }
      }
      if (one_d) HIPCHK(h, dup_device(&h->d_mask1d, t1.data(), t1.size(), st));
      HIPCHK(h, hipStreamSynchronize(st));
    }
  } else {
    if (h->d_masks) { hipFree(h->d_masks); h->d_masks = nullptr; }
    if (h->d_mask_off) { hipFree(h->d_mask_off); h->d_mask_off = nullptr; }
    if (h->d_mask1d) { hipFree(h->d_mask1d); h->d_mask1d = nullptr; }
  }
  // DFT operand tables of the proposal kernel, one set per distinct block height / width, folded to indices <= n/2
  // and zero padded to the MFMA tile grid (dimension formulas mirror propose_kernel):
  //   height n: FC[ky][y] = cos(2 pi ky y / n), FS = sin(...), ky, y <= n/2, each [KR = ceil4(n/2+1)][NR = ceil16(n/2+1)]
  //   width  n: GC[k][x]  = c_k cos(2 pi k x / n), GS = -c_k sin(...), k, x <= n/2 (c_k = 1 for k in {0, n/2}, else 2),
  //             each [Kc = ceil4(n/2+1)][M1 = ceil16(n/2+1)]
  const int max_len = std::max(max_bh, max_bw);
  std::vector<int32_t> fy_off(max_len + 1, -1), g_off(max_len + 1, -1);
  std::vector<double> tb;
  int krmax = 0, n1max = 0, m1max = 0, kcmax = 0, tiles_max = 0;
  h->prop_tiles1 = 0;
  for (int i = 0; i < n_sizes; ++i) {
    const int n = bh[i];
    const int nrow = n / 2 + 1;
    const int KR = (nrow + 3) & ~3, NR = (nrow + 15) & ~15, N1 = (n + 15) & ~15;
    krmax = std::max(krmax, KR); n1max = std::max(n1max, N1);
    if (fy_off[n] < 0) {
      fy_off[n] = (int32_t)tb.size();
      tb.resize(tb.size() + (size_t)2 * KR * NR, 0.0);
      double* FC = tb.data() + fy_off[n];
      double* FS = FC + (size_t)KR * NR;
      for (int ky = 0; ky < nrow; ++ky)
        for (int y = 0; y < nrow; ++y) {
          const double ang = 2.0 * M_PI * (double)(((int64_t)ky * y) % n) / (double)n;
          FC[ky * NR + y] = cos(ang);
          FS[ky * NR + y] = sin(ang);
        }
    }
    const int w = bw[i];
    const int ncol = w / 2 + 1, Kc = (ncol + 3) & ~3, M1 = (ncol + 15) & ~15;
    m1max = std::max(m1max, M1); kcmax = std::max(kcmax, Kc);
    tiles_max = std::max(tiles_max, (N1 / 16) * (M1 / 16));
    // stage-1 work units = 2 x this: (output tile, re | im) of the direct sums (odd heights), twice that over the half range of y for
    // the parity-split sums of even heights (proposal_device.h: dft_stage1)
    const int NRh = ((n / 2) / 2 + 16) & ~15;
    h->prop_tiles1 = std::max(h->prop_tiles1, (n & 1) ? (M1 / 16) * (NR / 16) : 2 * (M1 / 16) * (NRh / 16));
    if (g_off[w] < 0) {
      g_off[w] = (int32_t)tb.size();
      tb.resize(tb.size() + (size_t)2 * Kc * M1, 0.0);
      double* GC = tb.data() + g_off[w];
      double* GS = GC + (size_t)Kc * M1;
      for (int k = 0; k < ncol; ++k) {
        const double ck = (k == 0 || k == w / 2) ? 1.0 : 2.0;
        for (int x = 0; x < ncol; ++x) {
          const double ang = 2.0 * M_PI * (double)(((int64_t)k * x) % w) / (double)w;
          GC[k * M1 + x] = ck * cos(ang);
          GS[k * M1 + x] = -ck * sin(ang);
        }
      }
    }
  }
  auto stride16mod32 = [](int v) { int s = v; while ((s & 31) != 16) ++s; return s; };
  h->lds_sx = stride16mod32(m1max);
  h->lds_st = stride16mod32(n1max);
  h->lds_x_half = (krmax * h->lds_sx + 127) & ~127;   // one of the four folded coefficient planes; whole 1 KiB LDS-DMA pieces
  h->lds_tt = 2 * kcmax * h->lds_st;
  h->prop_tiles = tiles_max;
  h->tables_len = (int)tb.size();
  h->tab_max = 0;
  for (int i = 0; i < n_sizes; ++i) {
    const int nrow = bh[i] / 2 + 1, ncol = bw[i] / 2 + 1;
    h->tab_max = std::max(h->tab_max, 2 * ((nrow + 3) & ~3) * ((nrow + 15) & ~15));
    h->tab_max = std::max(h->tab_max, 2 * ((ncol + 3) & ~3) * ((ncol + 15) & ~15));
  }
  {
    std::vector<int32_t> k2_off((size_t)n_sizes);
    int32_t tot = 0;
    for (int i = 0; i < n_sizes; ++i) { k2_off[i] = tot; tot += (bh[i] / 2 + 1) * (bw[i] / 2 + 1); }
    HIPCHK(h, dup_device(&h->d_k2_off, k2_off.data(), k2_off.size(), st));
    if (h->d_k2) { hipFree(h->d_k2); h->d_k2 = nullptr; }
    HIPCHK(h, hipMalloc(&h->d_k2, sizeof(double) * (size_t)tot));
    h->k2_resolution = 0.0;
  }
  HIPCHK(h, dup_device(&h->d_tables, tb.data(), tb.size(), st));
  HIPCHK(h, dup_device(&h->d_fy_off, fy_off.data(), fy_off.size(), st));
  HIPCHK(h, dup_device(&h->d_g_off, g_off.data(), g_off.size(), st));
  {
    // 1-D twiddle tables: for every distinct block length n the values cos / sin(2 pi m / n), m < n -- the numbers the 2-D
    // tables above hold at (k, j) with m = (k * j) mod n (same expression, same libm calls: bit-identical operands)
    std::vector<int32_t> t1_off(max_len + 1, 0);
    std::vector<double> t1;
    std::vector<char> seen(max_len + 1, 0);
    for (int i = 0; i < n_sizes; ++i)
      for (int n : {bh[i], bw[i]}) {
        if (seen[n]) continue;
        seen[n] = 1;
        t1_off[n] = (int32_t)t1.size();
        for (int m = 0; m < n; ++m) t1.push_back(cos(2.0 * M_PI * (double)m / (double)n));
        for (int m = 0; m < n; ++m) t1.push_back(sin(2.0 * M_PI * (double)m / (double)n));
      }
    HIPCHK(h, dup_device(&h->d_tab1d, t1.data(), t1.size(), st));
    HIPCHK(h, dup_device(&h->d_t1_off, t1_off.data(), t1_off.size(), st));
    HIPCHK(h, hipStreamSynchronize(st));       // the host vectors end with this block
  }
  HIPCHK(h, hipStreamSynchronize(st));
  h->B.bh = h->d_bh; h->B.bw = h->d_bw; h->B.masks = h->d_masks; h->B.mask_off = h->d_mask_off; h->B.mask1d = h->d_mask1d;
  h->B.n_sizes = n_sizes; h->B.max_bh = max_bh; h->B.max_bw = max_bw;
  h->tile_cap = cap;
  h->field_stride = (int64_t)max_bh * max_bw;
  h->have_blocks = true;
  free_scratch(h);
  return GSM_OK;
}

extern "C" int gsm_set_centres(gsm_handle h, const int32_t* cells, int32_t n_cells, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!cells || n_cells < 1) return fail(h, GSM_E_ARG, "gsm_set_centres: empty centre list");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, dup_device(&h->d_centres, cells, (size_t)n_cells, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->n_centres = n_cells;
  h->have_centres = true;
  return GSM_OK;
}

extern "C" int gsm_init_loss(gsm_handle h, const void* beds, void* energy, double* loss_sum, double* loss0, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_init_loss: call gsm_set_static first");
  if (!beds || !energy || !loss_sum) return fail(h, GSM_E_ARG, "gsm_init_loss: NULL pointer");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_init_loss(h->S, h->n_chains, beds, energy, h->f32_state, loss_sum, loss0, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_residual(gsm_handle h, const double* beds, double* residual, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_residual: call gsm_set_static first");
  if (!beds || !residual) return fail(h, GSM_E_ARG, "gsm_residual: NULL pointer");
  if (h->f32_state) return fail(h, GSM_E_UNSUPPORTED, "gsm_residual: fp64 beds only (create the handle with dtype 0)");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_residual(h->S, h->n_chains, beds, residual, (hipStream_t)stream));
  return GSM_OK;
}

// 1 when this handle's static fields and block table go to the strip kernels (chain_strip_kernel.hip)
static int strip_for(gsm_handle h) {
  return (h->have_static && h->have_blocks &&
          strip_table_ok(h->S, h->B, std::max(4 * h->lds_x_half, h->lds_tt), h->prop_tiles1, h->prop_tiles)) ? 1 : 0;
}

extern "C" int gsm_strip_active(gsm_handle h) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static || !h->have_blocks) return fail(h, GSM_E_STATE, "gsm_strip_active: call gsm_set_static and gsm_set_blocks first");
  return strip_for(h);
}

static int check_device_flag(gsm_handle h, hipStream_t st, const char* who) {
  int32_t flag = 0;
  HIPCHK(h, hipMemcpyAsync(&flag, h->d_err, sizeof(flag), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (flag) {
    hipMemsetAsync(h->d_err, 0, sizeof(int32_t), st);
    hipStreamSynchronize(st);
    return fail(h, GSM_E_DEVICE_DATA, std::string(who) + ": size index or block centre out of range in device data (those steps were skipped)");
  }
  return GSM_OK;
}

extern "C" int gsm_run_replay(gsm_handle h, int32_t n_steps, void* beds, void* energy, uint32_t* resampled, double* loss_sum,
                              const int32_t* size_idx, const int32_t* centre, const double* u,
                              const double* fields, int64_t field_stride, double* loss, uint8_t* accept,
                              void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static || !h->have_blocks) return fail(h, GSM_E_STATE, "gsm_run_replay: call gsm_set_static and gsm_set_blocks first");
  if (n_steps < 0) return fail(h, GSM_E_ARG, "gsm_run_replay: n_steps < 0");
  if (n_steps == 0) return GSM_OK;
  if (!beds || !energy || !resampled || !loss_sum || !size_idx || !centre || !u || !fields || !loss || !accept)
    return fail(h, GSM_E_ARG, "gsm_run_replay: NULL pointer");
  if (field_stride < (int64_t)h->B.max_bh * h->B.max_bw)
    return fail(h, GSM_E_ARG, "gsm_run_replay: field_stride smaller than the largest block");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  StepArgs a{};
  a.S = h->S; a.B = h->B;
  a.n_chains = h->n_chains; a.n_steps = n_steps; a.tile_cap = h->tile_cap; a.strip = strip_for(h);
  a.beds = beds; a.energy = energy; a.f32_state = h->f32_state; a.resampled = resampled; a.loss_sum = loss_sum;
  a.size_idx = size_idx; a.centre = centre; a.u = u; a.fields = fields; a.field_stride = field_stride;
  a.loss = loss; a.accept = accept; a.blocks = nullptr;
  a.rec_stride = n_steps; a.rec_offset = 0; a.in_stride = n_steps;
  a.err_flag = h->d_err;
  HIPCHK(h, launch_step(a, st));
  return check_device_flag(h, st, "gsm_run_replay");
}

static int check_propose_ready(gsm_handle h, const gsm_rf_params* rf, const char* who) {
  if (!h->have_blocks || !h->d_masks) return fail(h, GSM_E_STATE, std::string(who) + ": call gsm_set_blocks with edge masks first");
  if (!h->have_centres) return fail(h, GSM_E_STATE, std::string(who) + ": call gsm_set_centres first");
  if (!rf) return fail(h, GSM_E_ARG, std::string(who) + ": rf is NULL");
  if (rf->generator == GSM_GEN_CHOLESKY) {
    if (!h->d_factors) return fail(h, GSM_E_STATE, std::string(who) + ": call gsm_set_factors first");
    return GSM_OK;
  }
  if (rf->generator != GSM_GEN_SPECTRAL) return fail(h, GSM_E_ARG, std::string(who) + ": unknown generator");
  if (rf->model < 0 || rf->model > 2) return fail(h, GSM_E_ARG, std::string(who) + ": unknown covariance model");
  if (!(rf->resolution > 0.0)) return fail(h, GSM_E_ARG, std::string(who) + ": rf.resolution must be > 0");
  if (rf->model == GSM_MODEL_MATERN && !(rf->smoothness > 0.0))
    return fail(h, GSM_E_ARG, std::string(who) + ": Matern needs smoothness > 0");
  const size_t lds = ((size_t)std::max(4 * h->lds_x_half, h->lds_tt) + 64 + kMathTabDoubles) * 8;
  if (lds > 160 * 1024 || h->prop_tiles > propose_max_tiles_per_wave() * propose_waves() ||
      h->prop_tiles1 > propose_max_tiles1_per_wave() * propose_waves())
    return fail(h, GSM_E_UNSUPPORTED, std::string(who) + ": block too large for the proposal kernel (LDS / accumulator tiles)");
  return GSM_OK;
}

// k^2 tables of the spectral amplitude for this resolution (built on first use, rebuilt when the resolution changes)
static int ensure_k2(gsm_handle h, const gsm_rf_params* rf, hipStream_t st) {
  if (!h->d_mathtab) {      // once per handle: the table of math_tables.h (Box-Muller of both generators)
    double tab[kMathTabDoubles];
    build_math_tables(tab);
    HIPCHK(h, hipMalloc(&h->d_mathtab, sizeof(tab)));
    HIPCHK(h, hipMemcpy(h->d_mathtab, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  if (rf->generator != GSM_GEN_SPECTRAL || h->k2_resolution == rf->resolution) return GSM_OK;
  HIPCHK(h, launch_k2_tables(h->B, h->d_k2_off, rf->resolution, h->d_k2, st));
  h->k2_resolution = rf->resolution;
  return GSM_OK;
}

static int ensure_scalars(gsm_handle h, int slot, size_t recs) {
  if (h->scalars_cap[slot] >= recs) return GSM_OK;
  if (h->d_scalars[slot]) { hipFree(h->d_scalars[slot]); h->d_scalars[slot] = nullptr; h->scalars_cap[slot] = 0; }
  HIPCHK(h, hipMalloc(&h->d_scalars[slot], recs * sizeof(PropScalars)));
  h->scalars_cap[slot] = recs;
  return GSM_OK;
}

static int ensure_chol(gsm_handle h, int slot, size_t recs, CholArgs* out) {
  auto& c = h->chol[slot];
  const int groups = h->B.n_sizes * h->n_classes;
  if (c.recs < recs || c.groups != groups) {
    if (c.ints) { hipFree(c.ints); hipFree(c.zoff); hipFree(c.per_rec); hipFree(c.scale); hipFree(c.zbuf); c = gsm_context::CholScratch(); }
    const size_t nmax_pad = (size_t)((h->B.max_bh * h->B.max_bw + 63) & ~63);
    HIPCHK(h, hipMalloc(&c.ints, sizeof(int) * (size_t)(6 * groups + 4)));
    HIPCHK(h, hipMalloc(&c.zoff, sizeof(int64_t) * (size_t)groups));
    HIPCHK(h, hipMalloc(&c.per_rec, sizeof(int) * 2 * recs));
    HIPCHK(h, hipMalloc(&c.scale, sizeof(double) * recs));
    HIPCHK(h, hipMalloc(&c.zbuf, sizeof(double) * nmax_pad * (recs + (size_t)64 * groups)));
    c.recs = recs; c.groups = groups;
  }
  out->n_classes = h->n_classes; out->n_groups = groups; out->factors = h->d_factors;
  out->counts = c.ints; out->rec_off = c.ints + groups; out->tile_off = c.ints + 2 * groups + 1;
  out->work_off = c.ints + 4 * groups + 2; out->z_off = c.zoff;
  out->group_of = c.per_rec; out->order = c.per_rec + recs; out->scale = c.scale; out->zbuf = c.zbuf;
  return GSM_OK;
}

static ProposeArgs make_propose(gsm_handle h, const gsm_rf_params* rf, int n_steps, int64_t step0, const uint64_t* seeds) {
  ProposeArgs p{};
  p.B = h->B; p.rf = *rf; p.H = h->H; p.W = h->W;
  p.n_chains = h->n_chains; p.n_steps = n_steps; p.step0 = step0; p.seeds = seeds;
  p.centres = h->d_centres; p.n_centres = h->n_centres;
  p.tables = h->d_tables; p.tables_len = h->tables_len; p.tab_max = h->tab_max; p.fy_off = h->d_fy_off; p.g_off = h->d_g_off;
  p.lds_sx = h->lds_sx; p.lds_st = h->lds_st; p.lds_x_half = h->lds_x_half; p.lds_tt = h->lds_tt;
  p.k2tab = h->d_k2; p.k2_off = h->d_k2_off; p.mathtab = h->d_mathtab;
  p.tab1d = h->d_tab1d; p.t1_off = h->d_t1_off;
  p.lds_main = std::max(4 * h->lds_x_half, h->lds_tt);
  p.tiles1_max = h->prop_tiles1; p.tiles2_max = h->prop_tiles;
  // stage 2 split by the parity of kx: on handles whose kernels hold two tile slots per wave (the strip kernels and the stand-alone proposal
  // kernel beside them); GSM_SPLIT2=0 keeps the direct sums (tests/test_gpu_strip.py compares the step kernels of the two families on equal fields)
  { static int on = -1; if (on < 0) { const char* v = getenv("GSM_SPLIT2"); on = v ? atoi(v) : 1; } p.split2 = (on && strip_for(h)) ? 1 : 0; p.parseval = p.split2; }
  return p;
}

extern "C" int gsm_propose_philox(gsm_handle h, int32_t n_steps, int64_t step0, const uint64_t* seeds,
                                  const gsm_rf_params* rf, int32_t* size_idx, int32_t* centre, double* u,
                                  double* fields, int64_t field_stride, double* rf_scalars, void* stream) {
  if (!h) return GSM_E_ARG;
  int rc = check_propose_ready(h, rf, "gsm_propose_philox");
  if (rc) return rc;
  if (n_steps < 1 || n_steps > 65535) return fail(h, GSM_E_ARG, "gsm_propose_philox: n_steps must be in [1, 65535]");
  if (!seeds || !size_idx || !centre || !u || !fields) return fail(h, GSM_E_ARG, "gsm_propose_philox: NULL pointer");
  if (field_stride < (int64_t)h->B.max_bh * h->B.max_bw) return fail(h, GSM_E_ARG, "gsm_propose_philox: field_stride too small");
  HIPCHK(h, hipSetDevice(h->device));
  { int rc2 = ensure_scalars(h, 0, (size_t)h->n_chains * n_steps); if (rc2) return rc2; }
  { int rc2 = ensure_k2(h, rf, (hipStream_t)stream); if (rc2) return rc2; }
  ProposeArgs p = make_propose(h, rf, n_steps, step0, seeds);
  p.size_idx = size_idx; p.centre = centre; p.u = u; p.fields = fields; p.field_stride = field_stride;
  p.rf_scalars = rf_scalars; p.scalars = h->d_scalars[0];
  if (rf->generator == GSM_GEN_CHOLESKY) {
    CholArgs c{};
    int rc2 = ensure_chol(h, 0, (size_t)h->n_chains * n_steps, &c);
    if (rc2) return rc2;
    HIPCHK(h, launch_propose_cholesky(p, c, (hipStream_t)stream));
  } else {
    HIPCHK(h, launch_propose(p, (hipStream_t)stream));
  }
  return GSM_OK;
}

extern "C" int gsm_spectral_from_noise(gsm_handle h, int32_t n_fields, const int32_t* size_idx, const double* rf_scalars,
                                       const gsm_rf_params* rf, const double* noise_re, const double* noise_im,
                                       const double* nugget_field, double* fields, int64_t field_stride, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_blocks || !h->d_masks) return fail(h, GSM_E_STATE, "gsm_spectral_from_noise: call gsm_set_blocks with edge masks first");
  if (!rf) return fail(h, GSM_E_ARG, "gsm_spectral_from_noise: rf is NULL");
  gsm_rf_params rfs = *rf;
  rfs.generator = GSM_GEN_SPECTRAL;
  const bool had_centres = h->have_centres;
  h->have_centres = true;                      // no centre is drawn here
  int rc = check_propose_ready(h, &rfs, "gsm_spectral_from_noise");
  h->have_centres = had_centres;
  if (rc) return rc;
  if (n_fields < 1 || n_fields > (1 << 20)) return fail(h, GSM_E_ARG, "gsm_spectral_from_noise: n_fields must be in [1, 2^20]");
  if (!size_idx || !rf_scalars || !noise_re || !noise_im || !fields) return fail(h, GSM_E_ARG, "gsm_spectral_from_noise: NULL pointer");
  if (field_stride < (int64_t)h->B.max_bh * h->B.max_bw) return fail(h, GSM_E_ARG, "gsm_spectral_from_noise: field_stride too small");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  std::vector<int32_t> si((size_t)n_fields);
  HIPCHK(h, hipMemcpyAsync(si.data(), size_idx, sizeof(int32_t) * (size_t)n_fields, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  for (int32_t v : si)
    if (v < 0 || v >= h->B.n_sizes) return fail(h, GSM_E_DEVICE_DATA, "gsm_spectral_from_noise: size index out of range");
  { int rc2 = ensure_scalars(h, 0, (size_t)n_fields); if (rc2) return rc2; }
  { int rc2 = ensure_k2(h, &rfs, st); if (rc2) return rc2; }
  ProposeArgs p = make_propose(h, &rfs, n_fields, 0, nullptr);
  p.n_chains = 1;
  p.fields = fields; p.field_stride = field_stride; p.scalars = h->d_scalars[0];
  HIPCHK(h, launch_spectral_from_noise(p, size_idx, rf_scalars, noise_re, noise_im, nugget_field, st));
  return GSM_OK;
}

extern "C" int gsm_run_noise(gsm_handle h, int32_t n_steps, void* beds, void* energy, uint32_t* resampled, double* loss_sum,
                             const int32_t* size_idx, const int32_t* centre, const double* u, const double* rf_scalars,
                             const gsm_rf_params* rf, const double* noise_re, const double* noise_im, const double* nugget_field,
                             int64_t field_stride, double* loss, uint8_t* accept, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_run_noise: call gsm_set_static first");
  if (!h->have_blocks || !h->d_masks) return fail(h, GSM_E_STATE, "gsm_run_noise: call gsm_set_blocks with edge masks first");
  if (!rf) return fail(h, GSM_E_ARG, "gsm_run_noise: rf is NULL");
  gsm_rf_params rfs = *rf;
  rfs.generator = GSM_GEN_SPECTRAL;
  const bool had_centres = h->have_centres;
  h->have_centres = true;                      // the centres arrive with the draws
  int rc = check_propose_ready(h, &rfs, "gsm_run_noise");
  h->have_centres = had_centres;
  if (rc) return rc;
  if (!strip_for(h)) return fail(h, GSM_E_UNSUPPORTED, "gsm_run_noise: this block table does not go to the strip kernels (gsm_strip_active); "
                                                       "use gsm_spectral_from_noise + gsm_run_replay");
  if (n_steps < 0 || n_steps > 65535) return fail(h, GSM_E_ARG, "gsm_run_noise: n_steps must be in [0, 65535]");
  if (n_steps == 0) return GSM_OK;
  if (!beds || !energy || !resampled || !loss_sum || !size_idx || !centre || !u || !rf_scalars || !noise_re || !noise_im || !loss || !accept)
    return fail(h, GSM_E_ARG, "gsm_run_noise: NULL pointer");
  if (field_stride < (int64_t)h->B.max_bh * h->B.max_bw) return fail(h, GSM_E_ARG, "gsm_run_noise: field_stride smaller than the largest block");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  { int rc2 = ensure_scalars(h, 0, (size_t)h->n_chains * n_steps); if (rc2) return rc2; }
  { int rc2 = ensure_k2(h, &rfs, st); if (rc2) return rc2; }
  FusedArgs fa{};
  StepArgs& a = fa.T;
  a.S = h->S; a.B = h->B;
  a.n_chains = h->n_chains; a.n_steps = n_steps; a.tile_cap = h->tile_cap; a.strip = 1;
  a.beds = beds; a.energy = energy; a.f32_state = h->f32_state; a.resampled = resampled; a.loss_sum = loss_sum;
  a.loss = loss; a.accept = accept; a.blocks = nullptr;
  a.rec_stride = n_steps; a.rec_offset = 0; a.in_stride = n_steps;
  a.err_flag = h->d_err;
  fa.P = make_propose(h, &rfs, n_steps, 0, nullptr);
  fa.P.scalars = h->d_scalars[0];
  fa.noise_re = noise_re; fa.noise_im = noise_im; fa.noise_nug = nugget_field; fa.noise_stride = field_stride;
  HIPCHK(h, launch_noise_chain_scalars(fa.P, size_idx, centre, u, rf_scalars, h->d_err, st));
  HIPCHK(h, launch_chain_strip_noise(fa, st));
  HIPCHK(h, launch_resampled_from_records(fa, st));
  return check_device_flag(h, st, "gsm_run_noise");
}

extern "C" int gsm_last_run_fused(gsm_handle h) { return h ? h->last_fused : GSM_E_ARG; }

extern "C" int gsm_set_fused(gsm_handle h, int32_t on) {
  if (!h) return GSM_E_ARG;
  h->use_fused = on ? 1 : 0;
  return GSM_OK;
}

extern "C" int gsm_enable_timing(gsm_handle h, int32_t on) {
  if (!h) return GSM_E_ARG;
  h->timing = on != 0;
  return GSM_OK;
}

extern "C" int gsm_last_timing(gsm_handle h, double* step_ms, int32_t* step_launches, double* prop_ms, int32_t* prop_launches) {
  if (!h) return GSM_E_ARG;
  if (step_ms) *step_ms = h->n_step_launch ? h->t_step_ms / h->n_step_launch : 0.0;
  if (step_launches) *step_launches = h->n_step_launch;
  if (prop_ms) *prop_ms = h->n_prop_launch ? h->t_prop_ms / h->n_prop_launch : 0.0;
  if (prop_launches) *prop_launches = h->n_prop_launch;
  return GSM_OK;
}

extern "C" int gsm_run_philox(gsm_handle h, int32_t n_steps, int64_t step0, int32_t batch, const uint64_t* seeds,
                              const gsm_rf_params* rf, void* beds, void* energy, uint32_t* resampled, double* loss_sum,
                              double* loss, uint8_t* accept, int32_t* blocks, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_run_philox: call gsm_set_static first");
  int rc = check_propose_ready(h, rf, "gsm_run_philox");
  if (rc) return rc;
  if (n_steps < 0) return fail(h, GSM_E_ARG, "gsm_run_philox: n_steps < 0");
  if (n_steps == 0) return GSM_OK;
  if (batch < 1 || batch > 65535) return fail(h, GSM_E_ARG, "gsm_run_philox: batch must be in [1, 65535]");
  if (!seeds || !beds || !energy || !resampled || !loss_sum || !loss || !accept) return fail(h, GSM_E_ARG, "gsm_run_philox: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  if (batch > n_steps) batch = n_steps;
  { int rc2 = ensure_k2(h, rf, st); if (rc2) return rc2; }
  // Spectral generator: one fused launch (chain_fused_kernel.hip) -- proposals are generated and consumed on the CU,
  // no field scratch, no second stream.  GSM_FUSED=0 keeps the two-kernel pipeline (also used by the Cholesky generator
  // and by block tables beyond the fused kernel's LDS budget).
  if (h->use_fused < 0) { const char* v = getenv("GSM_FUSED"); h->use_fused = v ? atoi(v) : 1; }
  h->last_fused = 0;
  if (h->use_fused && rf->generator == GSM_GEN_SPECTRAL) {
    // Segments of at most kFusedSegment steps: the per-(chain, step) scalar records (120 + 20 bytes) are sized by the
    // segment, not by the call, and a long call is a sequence of bounded launches on the caller's stream.  Counters are
    // functions of the absolute step, so the split is invisible in the results (test_fused_internal_segments...).
    int seg_cap = kFusedSegment;
    if (const char* v = getenv("GSM_FUSED_SEGMENT")) { const int q = atoi(v); if (q >= 1) seg_cap = q; }
    const int seg_max = std::min(n_steps, seg_cap);
    FusedArgs fa{};
    StepArgs& a = fa.T;
    a.S = h->S; a.B = h->B;
    a.n_chains = h->n_chains; a.n_steps = seg_max; a.tile_cap = h->tile_cap; a.strip = strip_for(h);
    a.beds = beds; a.energy = energy; a.f32_state = h->f32_state; a.resampled = resampled; a.loss_sum = loss_sum;
    a.loss = loss; a.accept = accept; a.blocks = blocks;
    a.rec_stride = n_steps; a.rec_offset = 0; a.in_stride = seg_max;
    a.err_flag = h->d_err;
    fa.P = make_propose(h, rf, seg_max, step0, seeds);
    if (fused_supported(fa)) {
      h->last_fused = 1;
      const size_t recs1 = (size_t)h->n_chains * seg_max;
      { int rc2 = ensure_scalars(h, 0, recs1); if (rc2) return rc2; }
      // the scalars kernel also writes (size_idx, centre, u) records: give it the scalar-sized scratch of slot 1
      auto& sc = h->scr[1];
      if (sc.recs < recs1 || sc.fields) {
        if (sc.size_idx) { hipFree(sc.size_idx); hipFree(sc.centre); hipFree(sc.u); if (sc.fields) hipFree(sc.fields); sc = gsm_context::Scratch(); }
        HIPCHK(h, hipMalloc(&sc.size_idx, recs1 * sizeof(int32_t)));
        HIPCHK(h, hipMalloc(&sc.centre, recs1 * 2 * sizeof(int32_t)));
        HIPCHK(h, hipMalloc(&sc.u, recs1 * sizeof(double)));
        sc.recs = recs1;
      }
      const int n_seg = (n_steps + seg_max - 1) / seg_max;
      std::vector<hipEvent_t> tev;
      if (h->timing) { tev.resize((size_t)2 * n_seg); for (auto& e : tev) HIPCHK(h, hipEventCreate(&e)); }
      for (int k = 0; k < n_seg; ++k) {
        const int off = k * seg_max;
        const int ns = std::min(seg_max, n_steps - off);
        a.n_steps = ns; a.in_stride = ns; a.rec_offset = off;
        fa.P = make_propose(h, rf, ns, step0 + off, seeds);
        fa.P.scalars = h->d_scalars[0];
        fa.P.size_idx = sc.size_idx; fa.P.centre = sc.centre; fa.P.u = sc.u;
        HIPCHK(h, launch_propose_scalars(fa.P, st));
        if (h->timing) HIPCHK(h, hipEventRecord(tev[2 * k], st));
        HIPCHK(h, launch_chain_fused(fa, st));
        if (h->timing) HIPCHK(h, hipEventRecord(tev[2 * k + 1], st));
      }
      rc = check_device_flag(h, st, "gsm_run_philox");
      if (h->timing) {
        h->t_step_ms = h->t_prop_ms = 0; h->n_step_launch = h->n_prop_launch = 0;
        for (int k = 0; k < n_seg; ++k) {
          float ms = 0;
          if (hipEventElapsedTime(&ms, tev[2 * k], tev[2 * k + 1]) == hipSuccess) { h->t_step_ms += ms; h->n_step_launch++; }
        }
        for (auto& e : tev) hipEventDestroy(e);
      }
      return rc;
    }
  }
  // scratch
  const size_t recs = (size_t)h->n_chains * batch;
  for (auto& s : h->scr) {
    if (s.recs >= recs && s.fields) continue;
    if (s.size_idx) { hipFree(s.size_idx); hipFree(s.centre); hipFree(s.u); if (s.fields) hipFree(s.fields); s = gsm_context::Scratch(); }
    HIPCHK(h, hipMalloc(&s.size_idx, recs * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&s.centre, recs * 2 * sizeof(int32_t)));
    HIPCHK(h, hipMalloc(&s.u, recs * sizeof(double)));
    HIPCHK(h, hipMalloc(&s.fields, recs * (size_t)h->field_stride * sizeof(double)));
    s.recs = recs;
  }
  for (int i = 0; i < 2; ++i) { int rc2 = ensure_scalars(h, i, recs); if (rc2) return rc2; }
  if (!h->aux) HIPCHK(h, hipStreamCreateWithFlags(&h->aux, hipStreamNonBlocking));
  for (int i = 0; i < 2; ++i) {
    if (!h->ev_prop[i]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_prop[i], hipEventDisableTiming));
    if (!h->ev_step[i]) HIPCHK(h, hipEventCreateWithFlags(&h->ev_step[i], hipEventDisableTiming));
  }
  const int n_batches = (n_steps + batch - 1) / batch;
  std::vector<hipEvent_t> tev;  // timing events: (prop start, prop stop, step start, step stop) per batch
  if (h->timing) {
    tev.resize((size_t)n_batches * 4);
    for (auto& e : tev) HIPCHK(h, hipEventCreate(&e));
  }
  // order the aux stream behind everything already queued on the caller's stream (seeds upload etc.)
  HIPCHK(h, hipEventRecord(h->ev_step[0], st));
  HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_step[0], 0));

  auto issue_propose = [&](int k) -> int {
    const int nb = std::min(batch, n_steps - k * batch);
    auto& s = h->scr[k & 1];
    if (k >= 2) HIPCHK(h, hipStreamWaitEvent(h->aux, h->ev_step[k & 1], 0));  // buffer free again
    ProposeArgs p = make_propose(h, rf, nb, step0 + (int64_t)k * batch, seeds);
    p.size_idx = s.size_idx; p.centre = s.centre; p.u = s.u; p.fields = s.fields; p.field_stride = h->field_stride;
    p.rf_scalars = nullptr; p.scalars = h->d_scalars[k & 1];
    if (h->timing) HIPCHK(h, hipEventRecord(tev[4 * k], h->aux));
    if (rf->generator == GSM_GEN_CHOLESKY) {
      CholArgs c{};
      int rc2 = ensure_chol(h, k & 1, recs, &c);
      if (rc2) return rc2;
      HIPCHK(h, launch_propose_cholesky(p, c, h->aux));
    } else
    HIPCHK(h, launch_propose(p, h->aux));
    if (h->timing) HIPCHK(h, hipEventRecord(tev[4 * k + 1], h->aux));
    HIPCHK(h, hipEventRecord(h->ev_prop[k & 1], h->aux));
    return GSM_OK;
  };

  rc = issue_propose(0);
  if (rc) return rc;
  for (int k = 0; k < n_batches; ++k) {
    if (k + 1 < n_batches) { rc = issue_propose(k + 1); if (rc) return rc; }
    const int nb = std::min(batch, n_steps - k * batch);
    auto& s = h->scr[k & 1];
    HIPCHK(h, hipStreamWaitEvent(st, h->ev_prop[k & 1], 0));
    StepArgs a{};
    a.S = h->S; a.B = h->B;
    a.n_chains = h->n_chains; a.n_steps = nb; a.tile_cap = h->tile_cap; a.strip = strip_for(h);
    a.beds = beds; a.energy = energy; a.f32_state = h->f32_state; a.resampled = resampled; a.loss_sum = loss_sum;
    a.size_idx = s.size_idx; a.centre = s.centre; a.u = s.u; a.fields = s.fields; a.field_stride = h->field_stride;
    a.loss = loss; a.accept = accept; a.blocks = blocks;
    a.rec_stride = n_steps; a.rec_offset = (int64_t)k * batch; a.in_stride = nb;
    a.err_flag = h->d_err;
    if (h->timing) HIPCHK(h, hipEventRecord(tev[4 * k + 2], st));
    HIPCHK(h, launch_step(a, st));
    if (h->timing) HIPCHK(h, hipEventRecord(tev[4 * k + 3], st));
    HIPCHK(h, hipEventRecord(h->ev_step[k & 1], st));
  }
  rc = check_device_flag(h, st, "gsm_run_philox");
  HIPCHK(h, hipStreamSynchronize(h->aux));
  if (h->timing) {
    h->t_step_ms = h->t_prop_ms = 0;
    h->n_step_launch = h->n_prop_launch = 0;
    for (int k = 0; k < n_batches; ++k) {
      float ms = 0;
      if (hipEventElapsedTime(&ms, tev[4 * k], tev[4 * k + 1]) == hipSuccess) { h->t_prop_ms += ms; h->n_prop_launch++; }
      if (hipEventElapsedTime(&ms, tev[4 * k + 2], tev[4 * k + 3]) == hipSuccess) { h->t_step_ms += ms; h->n_step_launch++; }
    }
    for (auto& e : tev) hipEventDestroy(e);
  }
  return rc;
}

extern "C" int gsm_debug_stamps(uint64_t* out, int32_t n_chains) {
  if (!out || n_chains < 1 || n_chains > 4096) return GSM_E_ARG;
  return debug_read_stamps((unsigned long long*)out, n_chains);
}

extern "C" int gsm_debug_stamps_fused(uint64_t* out, int32_t n_chains) {
  if (!out || n_chains < 1 || n_chains > 4096) return GSM_E_ARG;
  return debug_read_stamps_fused((unsigned long long*)out, n_chains);
}

extern "C" int gsm_debug_normals(uint64_t seed, int64_t step, uint32_t stream_id, uint32_t idx0, int32_t n, double* out, void* stream) {
  if (!out || n < 1) return GSM_E_ARG;
  double tab[kMathTabDoubles];
  build_math_tables(tab);
  double* d_tab = nullptr;
  if (hipMalloc(&d_tab, sizeof(tab)) != hipSuccess) return GSM_E_HIP;
  int rc = GSM_OK;
  if (hipMemcpy(d_tab, tab, sizeof(tab), hipMemcpyHostToDevice) != hipSuccess ||
      launch_debug_normals(seed, step, stream_id, idx0, n, d_tab, out, (hipStream_t)stream) != hipSuccess ||
      hipStreamSynchronize((hipStream_t)stream) != hipSuccess) rc = GSM_E_HIP;
  hipFree(d_tab);
  return rc;
}

extern "C" int gsm_debug_stream_copy(const double* src, double* dst, int64_t n, void* stream) {
  if (!src || !dst || n < 0) return GSM_E_ARG;
  return launch_stream_copy(src, dst, n, (hipStream_t)stream) == hipSuccess ? GSM_OK : GSM_E_HIP;
}

extern "C" int gsm_cov_assemble(gsm_handle h, int32_t bh, int32_t bw, double resolution, const gsm_vario* vario,
                                const double* lag_table, double* sigma, int64_t ld, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!vario || !sigma || bh < 1 || bw < 1 || ld < (int64_t)bh * bw || ld > 0x7fffffff)
    return fail(h, GSM_E_ARG, "gsm_cov_assemble: bad argument");
  if (vario->vtype < 0 || vario->vtype > 3) return fail(h, GSM_E_ARG, "gsm_cov_assemble: unknown vtype");
  if (vario->vtype == GSM_VTYPE_MATERN && !lag_table)
    return fail(h, GSM_E_ARG, "gsm_cov_assemble: the Matern model needs the host-computed lag table (scipy.special.kv)");
  if (!(vario->major_range > 0.0) || !(vario->minor_range > 0.0) || !(resolution > 0.0))
    return fail(h, GSM_E_ARG, "gsm_cov_assemble: ranges and resolution must be > 0");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_cov_assemble(bh, bw, resolution, *vario, lag_table, sigma, (int)ld, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_set_factors(gsm_handle h, int32_t n_classes, const double* const* factors, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_blocks) return fail(h, GSM_E_STATE, "gsm_set_factors: call gsm_set_blocks first");
  if (n_classes < 1 || !factors) return fail(h, GSM_E_ARG, "gsm_set_factors: bad argument");
  const int groups = h->B.n_sizes * n_classes;
  for (int g = 0; g < groups; ++g)
    if (!factors[g]) return fail(h, GSM_E_ARG, "gsm_set_factors: NULL factor");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  if (h->d_factors) { hipFree(h->d_factors); h->d_factors = nullptr; }
  HIPCHK(h, hipMalloc(&h->d_factors, sizeof(double*) * groups));
  HIPCHK(h, hipMemcpyAsync(h->d_factors, factors, sizeof(double*) * groups, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipStreamSynchronize(st));
  h->n_classes = n_classes;
  return GSM_OK;
}

static int ensure_pcg_tables(gsm_handle h) {
  if (h->d_pcg_tab) return GSM_OK;
  std::vector<uint64_t> tab(kPcgJumpWords + 768);
  const uint64_t* zig = nullptr;
  pcg64_host_tables(tab.data(), &zig);
  memcpy(tab.data() + kPcgJumpWords, zig, 768 * sizeof(uint64_t));
  HIPCHK(h, hipMalloc(&h->d_pcg_tab, tab.size() * sizeof(uint64_t)));
  HIPCHK(h, hipMemcpy(h->d_pcg_tab, tab.data(), tab.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
  return GSM_OK;
}

extern "C" int gsm_draw_pcg64(gsm_handle h, int32_t n_steps, const gsm_rf_params* rf, uint64_t* rf_state, uint64_t* chain_state,
                              const uint8_t* region_mask, int32_t* size_idx, int32_t* centre, double* u, double* rf_scalars,
                              double* noise_re, double* noise_im, double* nugget_field, int64_t field_stride, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_blocks) return fail(h, GSM_E_STATE, "gsm_draw_pcg64: call gsm_set_blocks first");
  if (!rf || !rf_state || !chain_state || !size_idx || !centre || !u || !rf_scalars || !noise_re || !noise_im)
    return fail(h, GSM_E_ARG, "gsm_draw_pcg64: NULL pointer");
  if (n_steps < 1) return fail(h, GSM_E_ARG, "gsm_draw_pcg64: n_steps must be >= 1");
  if (field_stride < (int64_t)h->B.max_bh * h->B.max_bw) return fail(h, GSM_E_ARG, "gsm_draw_pcg64: field_stride too small");
  if (rf->nugget_max > 0.0 && !nugget_field) return fail(h, GSM_E_ARG, "gsm_draw_pcg64: nugget_max > 0 needs nugget_field");
  if (h->B.n_sizes < 1 || (int64_t)h->H >= 0xFFFFFFFFll) return fail(h, GSM_E_ARG, "gsm_draw_pcg64: bad block table / grid");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = ensure_pcg_tables(h); if (rc) return rc; }
  PcgDrawArgs a{};
  a.H = h->H; a.W = h->W; a.n_chains = h->n_chains; a.n_steps = n_steps; a.n_sizes = h->B.n_sizes; a.rf = *rf;
  a.bh = h->B.bh; a.bw = h->B.bw; a.rf_state = rf_state; a.ch_state = chain_state; a.region_mask = region_mask;
  a.jump = h->d_pcg_tab; a.zig = h->d_pcg_tab + kPcgJumpWords;
  a.size_idx = size_idx; a.centre = centre; a.u = u; a.rf_scalars = rf_scalars;
  a.noise_re = noise_re; a.noise_im = noise_im; a.nugget = (rf->nugget_max > 0.0) ? nugget_field : nullptr; a.field_stride = field_stride;
  a.err = h->d_err;
  HIPCHK(h, launch_pcg64_draw(a, st));
  return GSM_OK;          // asynchronous: a chain that finds no centre inside region_mask raises the handle's device flag,
                          // reported by the next gsm_run_replay (which would also reject the out-of-range record)
}

static int sgs_fill(gsm_handle h, SgsArgs& a, double* grids, const double* zcond, const int32_t* windows, const double* x_axis,
                    const double* y_axis, const double* lag_cov, int32_t lag_mi, int32_t lag_mj, int32_t hw, double radius,
                    int32_t num_points, double sill, const int32_t* cell_off, const int32_t* cells, const double* z,
                    int32_t max_cells, const char* who, int parity = 0) {
  if (!grids || !windows || !x_axis || !y_axis || !lag_cov || !cell_off || !cells || !z) return fail(h, GSM_E_ARG, std::string(who) + ": NULL pointer");
  if (hw < 1) return fail(h, GSM_E_ARG, std::string(who) + ": search half-width (ceil(radius / grid spacing)) must be >= 1 cell");
  if (num_points < 8 || num_points > 48) return fail(h, GSM_E_UNSUPPORTED, std::string(who) + ": num_points must be in [8, 48]");
  if (!(radius > 0.0)) return fail(h, GSM_E_ARG, std::string(who) + ": radius must be > 0");
  if (lag_mi < 0 || lag_mj < 0) return fail(h, GSM_E_ARG, std::string(who) + ": lag table extents must be >= 0");
  if (max_cells < 1 || max_cells > 1024) return fail(h, GSM_E_ARG, std::string(who) + ": max_cells must be in [1, 1024]");
  // cells travel packed as (row << 16 | col) in an int32 and are unpacked with an arithmetic shift: rows up to 32767
  if (h->H < 2 || h->W < 2 || h->H > 32767 || h->W > 32767) return fail(h, GSM_E_UNSUPPORTED, std::string(who) + ": grid sides must be in [2, 32767]");
  // scratch: per (chain, slot) 48 x (value, weight) and a header; then ranks [n][1024] i32, rank_ok [n] i32
  max_cells = (max_cells + 63) & ~63;                        // record stride: whole 64-cell chunks (sgs_sequence_kernel: one cell per lane)
  const size_t n = (size_t)h->n_chains, cells_cap = n * (size_t)max_cells;
  if (h->sgs_rec_cells[parity] < cells_cap) {
    // a captured batch (gsm_sgs_iterate's hipGraph) holds the old scratch pointers: it must not be replayed
    if (h->sgs_graph_exec) { hipGraphExecDestroy(h->sgs_graph_exec); h->sgs_graph_exec = nullptr; }
    h->sgs_graph_key.clear();
    if (h->d_sgs_rec[parity]) { hipFree(h->d_sgs_rec[parity]); h->d_sgs_rec[parity] = nullptr; h->sgs_rec_cells[parity] = 0; }
    const size_t bytes = n * 1024 * 4 + n * 4 + 64 + cells_cap * (sizeof(SgsCellHdr) + 48 * sizeof(double2));
    hipError_t e = hipMalloc(&h->d_sgs_rec[parity], bytes);
    if (e != hipSuccess) return fail(h, GSM_E_HIP, std::string(who) + ": " + hipGetErrorString(e));
    h->sgs_rec_cells[parity] = cells_cap;
  }
  char* p = (char*)h->d_sgs_rec[parity];
  const size_t cap = h->sgs_rec_cells[parity];
  a.rec_vw = (double2*)p; p += cap * 48 * sizeof(double2);
  a.rec_hdr = (SgsCellHdr*)p; p += cap * sizeof(SgsCellHdr);
  a.rank = (int32_t*)p; p += n * 1024 * 4;
  a.rank_ok = (int32_t*)p;
  a.H = h->H; a.W = h->W; a.n_chains = h->n_chains;
  a.grid = grids; a.zcond = zcond; a.win = windows; a.xs = x_axis; a.ys = y_axis; a.lag = lag_cov;
  a.hw = hw; a.mi = lag_mi; a.mj = lag_mj; a.num_points = num_points; a.radius = radius; a.sill = sill;
  a.cell_off = cell_off; a.cells = cells; a.z = z; a.err = h->d_err; a.max_cells = max_cells;
  a.ktype = h->sgs_ktype; a.gmean = h->sgs_gmean; a.defer = 0;
  return GSM_OK;
}

static int sgs_report(gsm_handle h, hipStream_t st, const char* who) {
  int32_t flag = 0;
  HIPCHK(h, hipMemcpyAsync(&flag, h->d_err, sizeof(flag), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (!flag) return GSM_OK;
  hipMemsetAsync(h->d_err, 0, sizeof(int32_t), st);
  hipStreamSynchronize(st);
  const std::string w(who);
  if (flag & 4) return fail(h, GSM_E_DEVICE_DATA, w + ": a cell to simulate has no conditioning value anywhere on the grid (the reference "
                                                   "would widen its search radius for ever, MCMC.py:150-156)");
  if (flag & 8) return fail(h, GSM_E_DEVICE_DATA, w + ": singular kriging system (a pivot below eps * N * max|diag|: numpy.linalg.lstsq "
                                                   "would truncate singular values there, _krige.py:37)");
  if (flag & 16) return fail(h, GSM_E_DEVICE_DATA, w + ": no block centre inside the region mask after 64 attempts");
  if (flag & 64) return fail(h, GSM_E_ARG, w + ": the lag covariance table does not reach the lag between two chosen neighbours "
                                            "(lag_mi / lag_mj must cover 2 * hw, or the whole grid when the search radius is widened)");
  return fail(h, GSM_E_DEVICE_DATA, w + ": window outside the grid / larger than 1024 cells, more cells than max_cells, or a listed cell outside its window");
}

extern "C" int gsm_sgs_blocks(gsm_handle h, double* grids, const double* zcond, const int32_t* windows, const double* x_axis,
                              const double* y_axis, const double* lag_cov, int32_t lag_mi, int32_t lag_mj, int32_t hw, double radius,
                              int32_t num_points, double sill, const int32_t* cell_off, const int32_t* cells, const double* z,
                              int32_t max_cells, double* trace, int32_t* nbr_trace, void* stream) {
  if (!h) return GSM_E_ARG;
  SgsArgs a{};
  HIPCHK(h, hipSetDevice(h->device));
  int rc = sgs_fill(h, a, grids, zcond, windows, x_axis, y_axis, lag_cov, lag_mi, lag_mj, hw, radius, num_points, sill, cell_off, cells, z,
                    max_cells, "gsm_sgs_blocks");
  if (rc) return rc;
  a.cell_cnt = nullptr; a.trace = trace; a.nbr_trace = nbr_trace;
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, launch_sgs_blocks(a, a.max_cells, st));
  return sgs_report(h, st, "gsm_sgs_blocks");
}

extern "C" int gsm_sgs_blocks_batch(gsm_handle h, double* grids, const double* zcond, const int32_t* windows, const double* x_axis,
                                    const double* y_axis, const double* lag_cov, int32_t lag_mi, int32_t lag_mj, int32_t hw, double radius,
                                    int32_t num_points, double sill, const int32_t* cell_off, const int32_t* cell_cnt, const int32_t* cells,
                                    const double* z, int32_t max_cells, void* stream) {
  if (!h) return GSM_E_ARG;
  SgsArgs a{};
  HIPCHK(h, hipSetDevice(h->device));
  int rc = sgs_fill(h, a, grids, zcond, windows, x_axis, y_axis, lag_cov, lag_mi, lag_mj, hw, radius, num_points, sill, cell_off, cells, z,
                    max_cells, "gsm_sgs_blocks_batch");
  if (rc) return rc;
  a.cell_cnt = cell_cnt; a.trace = nullptr; a.nbr_trace = nullptr;
  HIPCHK(h, launch_sgs_blocks(a, a.max_cells, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_set_kriging(gsm_handle h, int32_t ktype, const double* global_mean) {
  if (!h) return GSM_E_ARG;
  if (ktype != GSM_KRIGING_ORDINARY && ktype != GSM_KRIGING_SIMPLE) return fail(h, GSM_E_ARG, "gsm_sgs_set_kriging: ktype must be GSM_KRIGING_ORDINARY or GSM_KRIGING_SIMPLE");
  if (ktype == GSM_KRIGING_SIMPLE && !global_mean) return fail(h, GSM_E_ARG, "gsm_sgs_set_kriging: simple kriging needs the global mean of every chain");
  h->sgs_ktype = ktype; h->sgs_gmean = ktype == GSM_KRIGING_SIMPLE ? global_mean : nullptr;
  h->sgs_graph_key.clear();                     // a captured batch baked the old choice into its launches
  if (h->sgs_graph_exec) { hipGraphExecDestroy(h->sgs_graph_exec); h->sgs_graph_exec = nullptr; }
  return GSM_OK;
}

extern "C" int gsm_sgs_check(gsm_handle h, void* stream) {
  if (!h) return GSM_E_ARG;
  HIPCHK(h, hipSetDevice(h->device));
  return sgs_report(h, (hipStream_t)stream, "gsm_sgs_check");
}

extern "C" int gsm_sgs_draw_philox(gsm_handle h, const uint64_t* seeds, int64_t iter0, int32_t n_iters, const uint8_t* region_mask,
                                   const uint8_t* is_data, int32_t min_x, int32_t max_x, int32_t min_y, int32_t max_y, int32_t max_cells,
                                   int32_t* windows, int32_t* blocks, int32_t* cell_off, int32_t* cell_cnt, int32_t* cells, double* z,
                                   double* u, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!seeds || !is_data || !windows || !blocks || !cell_off || !cell_cnt || !cells || !z || !u)
    return fail(h, GSM_E_ARG, "gsm_sgs_draw_philox: NULL pointer");
  if (n_iters < 1 || n_iters > 65535 || iter0 < 0) return fail(h, GSM_E_ARG, "gsm_sgs_draw_philox: n_iters must be in [1, 65535], iter0 >= 0");
  if (min_x < 1 || max_x <= min_x || min_y < 1 || max_y <= min_y) return fail(h, GSM_E_ARG, "gsm_sgs_draw_philox: block size ranges must be 1 <= min < max");
  if (max_cells < (max_x - 1) * (max_y - 1) || max_cells > 1024) return fail(h, GSM_E_ARG, "gsm_sgs_draw_philox: max_cells must hold the largest block and be <= 1024");
  if ((int64_t)n_iters * h->n_chains * max_cells >= (1LL << 31)) return fail(h, GSM_E_ARG, "gsm_sgs_draw_philox: n_iters * n_chains * max_cells must stay below 2^31 (32-bit cell offsets)");
  HIPCHK(h, hipSetDevice(h->device));
  if (!h->d_mathtab) {
    double tab[kMathTabDoubles];
    build_math_tables(tab);
    HIPCHK(h, hipMalloc(&h->d_mathtab, sizeof(tab)));
    HIPCHK(h, hipMemcpy(h->d_mathtab, tab, sizeof(tab), hipMemcpyHostToDevice));
  }
  SgsDrawArgs a{};
  a.H = h->H; a.W = h->W; a.n_chains = h->n_chains; a.n_iters = n_iters; a.iter0 = iter0; a.seeds = seeds;
  a.region_mask = region_mask; a.is_data = is_data; a.min_x = min_x; a.max_x = max_x; a.min_y = min_y; a.max_y = max_y;
  a.max_cells = max_cells; a.mathtab = h->d_mathtab;
  a.win = windows; a.blk = blocks; a.cell_off = cell_off; a.cell_cnt = cell_cnt; a.cells = cells; a.z = z; a.u = u; a.err = h->d_err;
  HIPCHK(h, launch_sgs_draw(a, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_draw_pcg64(gsm_handle h, uint64_t* chain_state, int32_t n_iters, const uint8_t* region_mask,
                                  const uint8_t* is_data, int32_t min_x, int32_t max_x, int32_t min_y, int32_t max_y, int32_t max_cells,
                                  int32_t* windows, int32_t* blocks, int32_t* cell_off, int32_t* cell_cnt, int32_t* cells, double* z,
                                  double* u, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!chain_state || !is_data || !windows || !blocks || !cell_off || !cell_cnt || !cells || !z || !u)
    return fail(h, GSM_E_ARG, "gsm_sgs_draw_pcg64: NULL pointer");
  if (n_iters < 1 || n_iters > 65535) return fail(h, GSM_E_ARG, "gsm_sgs_draw_pcg64: n_iters must be in [1, 65535]");
  if (min_x < 1 || max_x <= min_x || min_y < 1 || max_y <= min_y) return fail(h, GSM_E_ARG, "gsm_sgs_draw_pcg64: block size ranges must be 1 <= min < max");
  if (max_cells < (max_x - 1) * (max_y - 1) || max_cells > 1024) return fail(h, GSM_E_ARG, "gsm_sgs_draw_pcg64: max_cells must hold the largest block and be <= 1024");
  if (h->H > 32767 || h->W > 32767) return fail(h, GSM_E_UNSUPPORTED, "gsm_sgs_draw_pcg64: grid sides up to 32767 (cells are packed as row << 16 | col)");
  if ((int64_t)n_iters * h->n_chains * max_cells >= (1LL << 31)) return fail(h, GSM_E_ARG, "gsm_sgs_draw_pcg64: n_iters * n_chains * max_cells must stay below 2^31 (32-bit cell offsets)");
  HIPCHK(h, hipSetDevice(h->device));
  { int rc = ensure_pcg_tables(h); if (rc) return rc; }
  SgsDrawArgs a{};
  a.H = h->H; a.W = h->W; a.n_chains = h->n_chains; a.n_iters = n_iters; a.iter0 = 0; a.seeds = nullptr;
  a.region_mask = region_mask; a.is_data = is_data; a.min_x = min_x; a.max_x = max_x; a.min_y = min_y; a.max_y = max_y;
  a.max_cells = max_cells; a.mathtab = nullptr;
  a.win = windows; a.blk = blocks; a.cell_off = cell_off; a.cell_cnt = cell_cnt; a.cells = cells; a.z = z; a.u = u; a.err = h->d_err;
  HIPCHK(h, launch_sgs_draw_pcg64(a, chain_state, h->d_pcg_tab, h->d_pcg_tab + kPcgJumpWords, (hipStream_t)stream));
  return GSM_OK;
}

// scratch of the loss kernels: a partial sum and bad-cell count per (chain, part), and per chain the ticket of sgs_loss_tail_kernel
// (zero between launches: the kernel resets it)
static int sgs_parts_ensure(gsm_handle h) {
  const size_t need = (size_t)h->n_chains * sgs_loss_parts(h->S);
  if (h->sgs_part_cap >= need) return GSM_OK;
  if (h->sgs_graph_exec) { hipGraphExecDestroy(h->sgs_graph_exec); h->sgs_graph_exec = nullptr; }     // captured with the old scratch
  h->sgs_graph_key.clear();
  if (h->d_sgs_part_sum) { hipFree(h->d_sgs_part_sum); hipFree(h->d_sgs_part_bad); hipFree(h->d_sgs_ticket); h->d_sgs_part_sum = nullptr; h->d_sgs_part_bad = nullptr; h->d_sgs_ticket = nullptr; h->sgs_part_cap = 0; }
  HIPCHK(h, hipMalloc(&h->d_sgs_part_sum, need * sizeof(double)));
  HIPCHK(h, hipMalloc(&h->d_sgs_part_bad, need * sizeof(int32_t)));
  HIPCHK(h, hipMalloc(&h->d_sgs_ticket, (size_t)h->n_chains * sizeof(int32_t)));
  HIPCHK(h, hipMemset(h->d_sgs_ticket, 0, (size_t)h->n_chains * sizeof(int32_t)));
  h->sgs_part_cap = need;
  return GSM_OK;
}

extern "C" int gsm_sgs_loss(gsm_handle h, const double* beds, const double* trend, double* loss, int32_t* bad, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_sgs_loss: call gsm_set_static first");
  if (h->f32_state) return fail(h, GSM_E_UNSUPPORTED, "gsm_sgs_loss: fp64 beds only");
  if (!beds || !loss || !bad) return fail(h, GSM_E_ARG, "gsm_sgs_loss: NULL pointer");
  HIPCHK(h, hipSetDevice(h->device));
  if (int rc = sgs_parts_ensure(h)) return rc;
  HIPCHK(h, launch_sgs_loss(h->S, h->n_chains, beds, trend, loss, bad, h->d_sgs_part_sum, h->d_sgs_part_bad, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_decide(gsm_handle h, const double* loss_next, const int32_t* bad, const double* u, double* loss_prev,
                              uint8_t* accept, double* loss_rec, uint8_t* acc_rec, int64_t rec_stride, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!loss_next || !bad || !u || !loss_prev || !accept) return fail(h, GSM_E_ARG, "gsm_sgs_decide: NULL pointer");
  if ((loss_rec || acc_rec) && rec_stride < 1) return fail(h, GSM_E_ARG, "gsm_sgs_decide: rec_stride must be >= 1");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_sgs_decide(h->n_chains, loss_next, bad, u, loss_prev, accept, loss_rec, acc_rec, rec_stride, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_state_init(gsm_handle h, const double* beds, const double* trend, double* energy, double* state, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_sgs_state_init: call gsm_set_static first");
  if (h->f32_state) return fail(h, GSM_E_UNSUPPORTED, "gsm_sgs_state_init: fp64 beds only");
  if (!beds || !energy || !state) return fail(h, GSM_E_ARG, "gsm_sgs_state_init: NULL pointer");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_sgs_state_init(h->S, h->n_chains, beds, trend, energy, state, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_finish(gsm_handle h, double* cur, double* next, const double* trend, double* energy, double* state,
                              const int32_t* windows, const double* u, uint32_t* resampled, uint8_t* accept, double* loss_rec,
                              uint8_t* acc_rec, int64_t rec_stride, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_sgs_finish: call gsm_set_static first");
  if (!cur || !next || !energy || !state || !windows || !u || !resampled || !accept) return fail(h, GSM_E_ARG, "gsm_sgs_finish: NULL pointer");
  if ((loss_rec || acc_rec) && rec_stride < 1) return fail(h, GSM_E_ARG, "gsm_sgs_finish: rec_stride must be >= 1");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_sgs_finish(h->S, h->n_chains, cur, next, trend, energy, state, windows, u, resampled, accept, loss_rec, acc_rec, rec_stride,
                              h->d_err, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_qt_transform(gsm_handle h, const double* quantiles, const double* references, int32_t nq, const double* x,
                                double* out, int64_t n, int32_t inverse, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!quantiles || !references || !x || !out || nq < 1 || n < 0) return fail(h, GSM_E_ARG, "gsm_qt_transform: bad argument");
  if (n == 0) return GSM_OK;
  // sklearn clips the scores at ppf(1e-7 - spacing(1)) and ppf(1 - (1e-7 - spacing(1))) (QuantileTransformer._transform_col)
  const double lo = 1e-7 - 2.220446049250313e-16;
  const double clip_min = ns::ndtri(lo), clip_max = ns::ndtri(1.0 - lo);
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_qt(quantiles, references, nq, clip_min, clip_max, x, out, n, inverse, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_commit_map(gsm_handle h, double* cur, const double* proposed, uint32_t* resampled, const int32_t* windows,
                                  const uint8_t* accept, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!cur || !proposed || !resampled || !windows || !accept) return fail(h, GSM_E_ARG, "gsm_sgs_commit_map: NULL pointer");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_sgs_commit_map(h->H, h->W, h->n_chains, cur, proposed, resampled, windows, accept, (hipStream_t)stream));
  return GSM_OK;
}

extern "C" int gsm_sgs_commit(gsm_handle h, double* cur, double* next, uint32_t* resampled, const int32_t* windows,
                              const uint8_t* accept, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!cur || !next || !resampled || !windows || !accept) return fail(h, GSM_E_ARG, "gsm_sgs_commit: NULL pointer");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, launch_sgs_commit(h->H, h->W, h->n_chains, cur, next, resampled, windows, accept, (hipStream_t)stream));
  return GSM_OK;
}

// ---- one batch of small-scale iterations in ONE call (and, when the buffers are static, one hipGraph launch) ----------------
static int sgs_issue(gsm_handle h, const gsm_sgs_batch* b, int32_t n_iters, void* st) {
  const int64_t n = h->n_chains;
  const bool qt = b->qt_n > 0;
  const int64_t map = n * (int64_t)h->H * h->W;
  hipStream_t main_st = (hipStream_t)st;
  // The kriging weights of an iteration do not depend on the values of the grid, only on where values are: when the caller
  // promises that every cell holds one (grid_finite), the records of the iterations ahead (sgs_rank_kernel, sgs_weights_kernel) are made on a
  // second stream while the current iteration runs its value pass, transforms, loss and decision -- the longest kernel of an iteration
  // leaves the critical path.  `depth` sets of record scratch (iteration j uses set j mod depth; <= 512 MiB in all): the second stream
  // runs up to depth - 1 iterations ahead and waits for the main stream only every depth / 2 iterations -- a wait between two kernels of
  // a stream costs ~10 us even when it is satisfied (rocprofv3 timeline of the two-set version), more than a quarter of the kernel it precedes.
  const bool overlap = b->grid_finite != 0 && n_iters > 1;
  int depth = 1;
  if (overlap) {
    const size_t set_bytes = (size_t)n * (size_t)((b->max_cells + 63) & ~63) * (sizeof(SgsCellHdr) + 48 * sizeof(double2)) + (size_t)n * 4100;
    depth = (int)std::min<size_t>(gsm_context::kSgsDepth, std::max<size_t>(2, ((size_t)512 << 20) / std::max<size_t>(set_bytes, 1)));
    depth = std::min(depth, (int)n_iters);
    if (depth >= 4) depth &= ~1;                                      // an even depth: the waits fall every depth / 2 iterations
  }
  if (overlap && !h->sgs_side) {
    HIPCHK(h, hipStreamCreateWithFlags(&h->sgs_side, hipStreamNonBlocking));
    HIPCHK(h, hipStreamCreateWithFlags(&h->sgs_side2, hipStreamNonBlocking));
    for (hipEvent_t& e : h->sgs_ev) HIPCHK(h, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  // two record streams, even and odd iterations: with few chains a launch of sgs_weights_kernel leaves most of the chip idle
  hipStream_t rec_st[2] = {h->sgs_side, depth >= 4 ? h->sgs_side2 : h->sgs_side};
  bool must_wait[2] = {false, false};                         // the stream has not yet been told of the main stream's latest sync point
  auto fill = [&](int32_t j, SgsArgs& a) -> int {
    const int64_t base = b->cell_base ? b->cell_base[j] : 0;
    int rc = sgs_fill(h, a, b->next, b->zcond, b->windows + 4 * n * j, b->x_axis, b->y_axis, b->lag_cov, b->lag_mi, b->lag_mj, b->hw, b->radius,
                      b->num_points, b->sill, b->cell_off + b->cell_off_stride * j, b->cells + 2 * base, b->z + base, b->max_cells,
                      "gsm_sgs_iterate", overlap ? (j % depth) : 0);
    if (rc) return rc;
    a.cell_cnt = b->cell_cnt ? b->cell_cnt + n * j : nullptr; a.trace = nullptr; a.nbr_trace = nullptr;
    a.defer = overlap ? 1 : 0;
    return GSM_OK;
  };
  hipEvent_t* ev_w = h->sgs_ev;                               // [depth] the records of set s are complete
  hipEvent_t ev_fork = h->sgs_ev[gsm_context::kSgsDepth];    // the draws are there
  hipEvent_t ev_seq = h->sgs_ev[gsm_context::kSgsDepth + 1]; // the main stream has finished the value pass of some iteration
  std::vector<SgsArgs> args(overlap ? n_iters : 1);
  int rc;
  const int half = std::max(1, depth / 2);
  int32_t issued = 0;                                         // iterations whose records have been enqueued on the second stream
  auto enqueue_records = [&](int32_t upto) -> int {          // records of iterations issued .. upto - 1
    for (; issued < upto; ++issued) {
      const int q = issued & 1;
      if (must_wait[q]) { HIPCHK(h, hipStreamWaitEvent(rec_st[q], ev_seq, 0)); must_wait[q] = false; if (rec_st[0] == rec_st[1]) must_wait[q ^ 1] = false; }
      HIPCHK(h, launch_sgs_weights(args[issued], args[issued].max_cells, rec_st[q]));
      HIPCHK(h, hipEventRecord(ev_w[issued % depth], rec_st[q]));
    }
    return GSM_OK;
  };
  if (overlap) {
    for (int32_t j = 0; j < n_iters; ++j)
      if ((rc = fill(j, args[j]))) return rc;                 // every set of scratch exists before anything is enqueued
    HIPCHK(h, hipEventRecord(ev_fork, main_st));              // fork: whatever made the draws is on the main stream
    HIPCHK(h, hipStreamWaitEvent(rec_st[0], ev_fork, 0));
    if (rec_st[1] != rec_st[0] && n_iters > 1) HIPCHK(h, hipStreamWaitEvent(rec_st[1], ev_fork, 0));
    if ((rc = enqueue_records(std::min<int32_t>(n_iters, depth - 1 > 0 ? depth - 1 : 1)))) return rc;
  }
  // with a transformer: both transforms of an iteration inside the tail launch where its tables and the map's parts fit
  // (sgs_loss_tail_kernel<true>); the forward transform of the batch's first iteration is the stand-alone launch
  // Measured (same box): 16 chains +4.8 %, 32 chains +9.5 %, 64 chains +4.1 %, 256 chains +2.9 %, 8 chains -7 %, 4 chains -19 % (a thread of the tail launch then makes ~8 transforms one after
  // the other where the stand-alone launches make one per thread: with a few chains latency is what counts).  GSM_SGS_TAIL_QT=0 / 1 forces.
  const char* tail_env = getenv("GSM_SGS_TAIL_QT");
  const bool tail_qt_env = tail_env ? tail_env[0] != '0' : h->n_chains >= 12;
  const bool tail_qt = qt && !b->windowed && tail_qt_env && h->have_static && sgs_tail_takes_qt(h->S, b->qt_n);
  double qt_clip_min = 0.0, qt_clip_max = 0.0;
  if (tail_qt) {
    const double lo = 1e-7 - 2.220446049250313e-16;               // as gsm_qt_transform
    qt_clip_min = ns::ndtri(lo); qt_clip_max = ns::ndtri(1.0 - lo);
    if (h->sgs_next_acc_cap < (size_t)map) {
      if (h->sgs_graph_exec) { hipGraphExecDestroy(h->sgs_graph_exec); h->sgs_graph_exec = nullptr; }
      h->sgs_graph_key.clear();
      if (h->d_sgs_next_acc) { hipFree(h->d_sgs_next_acc); h->d_sgs_next_acc = nullptr; h->sgs_next_acc_cap = 0; }
      HIPCHK(h, hipMalloc(&h->d_sgs_next_acc, (size_t)map * sizeof(double)));
      h->sgs_next_acc_cap = (size_t)map;
    }
  }
  for (int32_t j = 0; j < n_iters; ++j) {
    const int32_t* win = b->windows + 4 * n * j;
    const double* u = b->u + n * j;
    if (qt && (!tail_qt || j == 0) && (rc = gsm_qt_transform(h, b->qt_quantiles, b->qt_references, b->qt_n, b->cur, b->next, map, 0, st))) return rc;   // MCMC.py:1766
    if (overlap) {
      HIPCHK(h, hipStreamWaitEvent(main_st, ev_w[j % depth], 0));
      HIPCHK(h, launch_sgs_sequence(args[j], main_st));
      // set j mod depth is free again once this value pass is over: every `half` iterations the second stream is told so and
      // takes the next `half` iterations' records (it then runs between depth - half and depth - 1 iterations ahead)
      if ((j + 1) % half == 0 && issued < n_iters) {
        HIPCHK(h, hipEventRecord(ev_seq, main_st));
        must_wait[0] = must_wait[1] = true;                    // (a stream waits when it next gets work: no wait is left dangling in a capture)
        if ((rc = enqueue_records(std::min<int32_t>(n_iters, j + depth)))) return rc;
      }
    } else {
      if ((rc = fill(j, args[0]))) return rc;
      HIPCHK(h, launch_sgs_blocks(args[0], args[0].max_cells, main_st));
    }
    if (b->windowed) {
      if ((rc = gsm_sgs_finish(h, b->cur, b->next, b->trend, b->energy, b->state, win, u, b->resampled, b->accept,
                               b->loss_rec ? b->loss_rec + j : nullptr, b->acc_rec ? b->acc_rec + j : nullptr, n_iters, st))) return rc;
      continue;
    }
    if (qt && !tail_qt && (rc = gsm_qt_transform(h, b->qt_quantiles, b->qt_references, b->qt_n, b->next, b->proposed, map, 1, st))) return rc;  // MCMC.py:1777
    // loss of the proposal, decision and commit (gsm_sgs_loss, gsm_sgs_decide, gsm_sgs_commit_map / gsm_sgs_commit) in one launch
    if (!h->have_static) return fail(h, GSM_E_STATE, "gsm_sgs_iterate: call gsm_set_static first");
    if (h->f32_state) return fail(h, GSM_E_UNSUPPORTED, "gsm_sgs_iterate: fp64 beds only");
    if ((rc = sgs_parts_ensure(h))) return rc;
    HIPCHK(h, launch_sgs_loss_tail(h->S, h->n_chains, b->trend, h->d_sgs_part_sum, h->d_sgs_part_bad, h->d_sgs_ticket, b->loss, b->bad, u,
                                   b->loss_prev, b->accept, b->loss_rec ? b->loss_rec + j : nullptr, b->acc_rec ? b->acc_rec + j : nullptr,
                                   n_iters, qt ? 1 : 2, b->cur, qt ? b->proposed : b->next, b->resampled, win, main_st,
                                   tail_qt ? b->qt_quantiles : nullptr, b->qt_references, b->qt_n, qt_clip_min, qt_clip_max, b->next, h->d_sgs_next_acc));
  }
  return GSM_OK;
}

extern "C" int gsm_sgs_iterate(gsm_handle h, const gsm_sgs_batch* b, int32_t n_iters, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!b || n_iters < 1) return fail(h, GSM_E_ARG, "gsm_sgs_iterate: NULL batch / n_iters < 1");
  if (!b->cur || !b->next || !b->windows || !b->cell_off || !b->cells || !b->z || !b->u || !b->resampled || !b->accept)
    return fail(h, GSM_E_ARG, "gsm_sgs_iterate: NULL pointer");
  if (b->qt_n < 0 || (b->qt_n > 0 && (!b->qt_quantiles || !b->qt_references || !b->proposed)))
    return fail(h, GSM_E_ARG, "gsm_sgs_iterate: a transformer needs qt_quantiles, qt_references and the `proposed` planes");
  if (b->windowed && (b->qt_n > 0 || !b->energy || !b->state))
    return fail(h, GSM_E_ARG, "gsm_sgs_iterate: the windowed finish needs energy / state and no transformer");
  if (!b->windowed && (!b->loss || !b->bad || !b->loss_prev)) return fail(h, GSM_E_ARG, "gsm_sgs_iterate: loss / bad / loss_prev are NULL");
  if (b->cell_off_stride < h->n_chains) return fail(h, GSM_E_ARG, "gsm_sgs_iterate: cell_off_stride must be >= n_chains");
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipSetDevice(h->device));
  // Graph replay needs every pointer of the batch to be the same as at capture, a capturable (non-NULL) stream, and all lazy
  // allocations of the launch path made: the first call with a given batch runs eagerly, the second captures, later ones replay.
  const bool graphable = b->use_graph && st != nullptr && !b->cell_base;
  if (!graphable) return sgs_issue(h, b, n_iters, stream);
  std::vector<char> key(sizeof(gsm_sgs_batch) + sizeof(int32_t));
  memcpy(key.data(), b, sizeof(gsm_sgs_batch));
  memcpy(key.data() + sizeof(gsm_sgs_batch), &n_iters, sizeof(int32_t));
  if (key != h->sgs_graph_key) {
    if (h->sgs_graph_exec) { hipGraphExecDestroy(h->sgs_graph_exec); h->sgs_graph_exec = nullptr; }
    h->sgs_graph_key = key;
    return sgs_issue(h, b, n_iters, stream);
  }
  if (!h->sgs_graph_exec) {
    HIPCHK(h, hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    int rc = sgs_issue(h, b, n_iters, stream);
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture(st, &g);
    if (rc || e != hipSuccess) {
      if (g) hipGraphDestroy(g);
      h->sgs_graph_key.clear();
      return rc ? rc : fail(h, GSM_E_HIP, std::string("gsm_sgs_iterate: hipStreamEndCapture: ") + hipGetErrorString(e));
    }
    e = hipGraphInstantiate(&h->sgs_graph_exec, g, nullptr, nullptr, 0);
    hipGraphDestroy(g);
    if (e != hipSuccess) { h->sgs_graph_exec = nullptr; h->sgs_graph_key.clear(); return fail(h, GSM_E_HIP, std::string("gsm_sgs_iterate: hipGraphInstantiate: ") + hipGetErrorString(e)); }
  }
  HIPCHK(h, hipGraphLaunch(h->sgs_graph_exec, st));
  ++h->sgs_graph_replays;
  return GSM_OK;
}

extern "C" int gsm_sgs_graph_replays(gsm_handle h) { return h ? h->sgs_graph_replays : GSM_E_ARG; }

extern "C" int gsm_struct_size(int32_t which) {
  return which == 0 ? (int)sizeof(gsm_rf_params) : which == 1 ? (int)sizeof(gsm_sgs_batch) : which == 2 ? (int)sizeof(gsm_vario) : -1;
}

extern "C" int gsm_min_dist_from_mask(gsm_handle h, const double* xx, const double* yy, const uint8_t* mask,
                                      double* dist, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!xx || !yy || !mask || !dist) return fail(h, GSM_E_ARG, "gsm_min_dist_from_mask: NULL pointer");
  HIPCHK(h, hipSetDevice(h->device));
  const int n = h->H * h->W;
  hipStream_t st = (hipStream_t)stream;
  double2* pts = nullptr;
  int* count = nullptr;
  HIPCHK(h, hipMalloc(&pts, sizeof(double2) * (size_t)n));
  HIPCHK(h, hipMalloc(&count, sizeof(int)));
  HIPCHK(h, hipMemsetAsync(count, 0, sizeof(int), st));
  hipError_t e = launch_min_dist(xx, yy, mask, n, pts, count, dist, st);
  int m = 0;
  if (e == hipSuccess) e = hipMemcpyAsync(&m, count, sizeof(int), hipMemcpyDeviceToHost, st);
  if (e == hipSuccess) e = hipStreamSynchronize(st);
  hipFree(pts);
  hipFree(count);
  if (e != hipSuccess) return fail(h, GSM_E_HIP, std::string("gsm_min_dist_from_mask: ") + hipGetErrorString(e));
  if (m == 0) return fail(h, GSM_E_ARG, "gsm_min_dist_from_mask: mask selects no cell");
  return GSM_OK;
}

extern "C" int gsm_cholesky_upper(gsm_handle h, double* a, int32_t n, int64_t ld, double jitter, void* stream) {
  if (!h) return GSM_E_ARG;
  if (!a || n < 64 || (n % 64) != 0 || ld < n || ld > 0x7fffffff)
    return fail(h, GSM_E_ARG, "gsm_cholesky_upper: n must be a positive multiple of 64 and ld >= n");
  HIPCHK(h, hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  HIPCHK(h, hipMemsetAsync(h->d_err, 0, sizeof(int32_t), st));
  HIPCHK(h, launch_cholesky_upper(a, n, (int)ld, jitter, h->d_err, st));
  int32_t info = 0;
  HIPCHK(h, hipMemcpyAsync(&info, h->d_err, sizeof(info), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (info != 0) {
    hipMemsetAsync(h->d_err, 0, sizeof(int32_t), st);
    hipStreamSynchronize(st);
    return fail(h, GSM_E_ARG, "gsm_cholesky_upper: matrix not positive definite at pivot " + std::to_string(info) +
                              " (raise the jitter: the Gaussian covariance is numerically singular, SURVEY.md section 7)");
  }
  return GSM_OK;
}
