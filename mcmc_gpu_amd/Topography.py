"""Mass-conservation residual on the MI355X (reference gstatsMCMC/Topography.py:592-612).

Only the likelihood stencil of the hot path lives here; the reference's data loaders / gridding helpers
(Topography.py:36-571) are out of scope.
"""
from __future__ import annotations

import numpy as np


def _residual_device(bed, surf, velx, vely, dhdt, smb, resolution):
    import torch
    from .engine import GsmEngine
    is_t = isinstance(bed, torch.Tensor)
    b = bed.detach().cpu().numpy() if is_t else np.asarray(bed, dtype=np.float64)
    batched = b.ndim == 3
    b3 = b if batched else b[None]
    H, W = b3.shape[1:]
    to_np = lambda a: np.array(np.broadcast_to(np.asarray(a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else a,
                                                          dtype=np.float64), (H, W)))
    eng = GsmEngine(H, W, b3.shape[0])
    try:
        ones = np.ones((H, W), dtype=np.uint8)
        eng.set_static(to_np(surf), to_np(velx), to_np(vely), to_np(dhdt), to_np(smb), None, ones, ones,
                       float(resolution), 1.0)
        r = eng.residual(np.ascontiguousarray(b3, dtype=np.float64))
        return (r if batched else r[0]) if is_t else (r.cpu().numpy() if batched else r[0].cpu().numpy())
    finally:
        eng.close()


def get_mass_conservation_residual(bed, surf, velx, vely, dhdt, smb, resolution):
    """d/dx(velx*(surf-bed)) + d/dy(vely*(surf-bed)) + dhdt - smb with np.gradient's second-order interior /
    first-order edge differences (Topography.py:592-600), evaluated by the HIP residual kernel in fp64.
    `bed` may be (H, W) or a batch (n, H, W); returns a NumPy array of the same shape."""
    return _residual_device(bed, surf, velx, vely, dhdt, smb, resolution)


def get_mass_conservation_residual_tensor(bed, surf, velx, vely, dhdt, smb, resolution):
    """torch twin (Topography.py:602-612): accepts tensors, returns a cuda float64 tensor."""
    return _residual_device(bed, surf, velx, vely, dhdt, smb, resolution)


def get_highvel_boundary(velx, vely, velmag_threshold, grounded_ice_mask, ocean_mask, distance_max, xx, yy,
                         smooth_mode=10):
    """High-velocity region mask, smoothed and grown outward by `distance_max` (reference Topography.py:543-571).

    Same steps as the reference: threshold on the speed of grounded ice (plus ocean), PIL mode filter, then every
    grounded cell closer than `distance_max` to the filtered region.  The reference finds that distance with an
    O((H*W)^2) Python double loop (:564-566); here it is one call of the exact distance transform
    (`min_dist_from_mask`: HIP kernel on a GPU machine, KD-tree otherwise) -- SURVEY.md section 8f rank 2."""
    from PIL import Image, ImageFilter
    from .MCMC_gpu import min_dist_from_mask
    grounded = np.asarray(grounded_ice_mask)
    mask = (grounded) & (np.sqrt(velx ** 2 + vely ** 2) >= velmag_threshold)
    mask = mask | ocean_mask
    image = Image.fromarray((mask * 255).astype(np.uint8)).filter(ImageFilter.ModeFilter(size=smooth_mode))
    mask_mat = np.array(np.array(image) / 255, dtype=int)
    hard = (mask_mat == 1) & (grounded == 1)
    if not hard.any():
        return np.zeros(np.shape(xx), dtype=bool) & grounded
    dist = min_dist_from_mask(np.asarray(xx, dtype=np.float64), np.asarray(yy, dtype=np.float64), hard)
    return (dist < distance_max) & grounded
