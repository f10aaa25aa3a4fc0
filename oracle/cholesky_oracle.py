"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the build's precomputed-factor ("L z") proposal generator.

The reference has no such generator (README.md:21-23: future work); it is assembled from the reference's covariance
ingredients, which mcmc_oracle.cov_matrix restates and fixture F6 pins bit-for-bit to
gstatsim_custom._krige.make_sigma.  Parity of the generator itself is therefore pinned by this file alone
("parity unpinned" by reference vectors: none exist)."""
from __future__ import annotations

import math

import numpy as np

import mcmc_oracle as orc
import philox_oracle as po

STREAM_CHOLESKY = 3


def block_coords(bh, bw, res):
    jj, ii = np.meshgrid(np.arange(bw), np.arange(bh))
    return np.stack([jj.ravel() * res, ii.ravel() * res], axis=1)


def proposal(seed, step, rf, pairs, masks, centres, W, resolution, varios, jitter=1e-8, factors=None):
    """One proposal of cholesky_kernel.hip.  varios: list (one per range class) of gstatsim vario dicts.
    factors: optional dict {(si, rc): L} cache."""
    n_sizes, n_classes = pairs.shape[1], len(varios)
    d = po.draw(seed, step, po.STREAM_SCALARS, np.arange(4))
    si = int((int(d[3, 0]) * n_sizes) >> 32)
    rc = int((int(d[1, 0]) * n_classes) >> 32)
    scale = (rf.scale_min + (rf.scale_max - rf.scale_min) * float(po.u01(d[0, 0], d[0, 1]))) / 3.0
    u_acc = float(po.u01(d[2, 0], d[2, 1]))
    cw = (int(d[2, 3]) << 32) | int(d[2, 2])
    cell = int(centres[(cw * len(centres)) >> 64])
    bw, bh = int(pairs[0, si]), int(pairs[1, si])
    N = bh * bw
    key = (si, rc)
    if factors is not None and key in factors:
        L = factors[key]
    else:
        sig = orc.cov_matrix(block_coords(bh, bw, resolution), varios[rc])
        L = np.linalg.cholesky(sig + jitter * varios[rc]["sill"] * np.eye(N))
        if factors is not None:
            factors[key] = L
    kp = np.arange((N + 1) // 2)
    g1, g2 = po.normals2(seed, step, STREAM_CHOLESKY, kp)
    z = np.empty(2 * len(kp))
    z[0::2], z[1::2] = g1, g2
    z = z[:N]
    fld = (L @ z).reshape(bh, bw) * scale
    return dict(size_idx=si, range_class=rc, centre=(cell // W, cell % W), u=u_acc, scale=scale, z=z,
                field=fld * masks[si])
