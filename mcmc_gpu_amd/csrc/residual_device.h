// Full-grid mass-conservation residual of one cell from plain arrays, shared by the init / residual kernels
// (step_kernel.hip) and the small-scale chain's loss kernel (sgs_kernel.hip).
#pragma once
#include "gsm_internal.h"

namespace gsm {

// Residual of cell (r, c) of the full grid from plain arrays (init / residual kernels).
// np.gradient semantics: interior (f[i+1]-f[i-1])/(2.0*h); first/last (f[1]-f[0])/h, (f[-1]-f[-2])/h.
template <class BedAt>
__device__ __forceinline__ double cell_residual(const StaticFields& S, int r, int c, BedAt bed_at) {
  const int W = S.W, H = S.H;
  const int cl = (c == 0) ? 0 : c - 1;
  const int cr = (c == W - 1) ? W - 1 : c + 1;
  const int ru = (r == 0) ? 0 : r - 1;
  const int rd = (r == H - 1) ? H - 1 : r + 1;
  const double denx = (cr - cl == 2) ? S.two_res : S.res;
  const double deny = (rd - ru == 2) ? S.two_res : S.res;
  const int gl = r * W + cl, gr = r * W + cr, gu = ru * W + c, gd = rd * W + c, g = r * W + c;
  const double qxr = S.velx[gr] * (S.surf[gr] - bed_at(r, cr));
  const double qxl = S.velx[gl] * (S.surf[gl] - bed_at(r, cl));
  const double qyd = S.vely[gd] * (S.surf[gd] - bed_at(rd, c));
  const double qyu = S.vely[gu] * (S.surf[gu] - bed_at(ru, c));
  const double dx = (qxr - qxl) / denx;
  const double dy = (qyd - qyu) / deny;
  return ((dx + dy) + S.dhdt[g]) - S.smb[g];
}

}  // namespace gsm
