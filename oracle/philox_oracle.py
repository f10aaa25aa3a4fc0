"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the build's Philox proposal generator.

The reference has no counter-based generator: it draws from NumPy's PCG64
(gstatsMCMC/MCMC.py:483-492).  The device generator (mcmc_gpu_amd/csrc/proposal_kernel.hip) is new, so
its parity is pinned in two ways:
  * value-for-value against THIS file (same counters, same formulas, NumPy arithmetic and
    numpy.fft.irfft2 instead of the device DFT), tolerance 1e-10 relative to the field scale;
  * in distribution against the reference's spectral proposal restated in mcmc_oracle.spectral_field
    (gstatsMCMC/MCMC.py:176-254), which the golden vectors pin to the reference.
Philox4x32-10 itself is checked against the Random123 known-answer vectors (tests/test_philox.py).
"""
from __future__ import annotations

import math

import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK32 = np.uint64(0xFFFFFFFF)
STREAM_SCALARS, STREAM_SPECTRUM, STREAM_NUGGET = 0, 1, 2


def philox4x32_10(ctr, key):
    """ctr: (..., 4) uint32-valued array, key: (2,) ints -> (..., 4) uint32 array."""
    c = np.asarray(ctr, dtype=np.uint64) & MASK32
    x, y, z, w = c[..., 0], c[..., 1], c[..., 2], c[..., 3]
    k0, k1 = int(key[0]) & 0xFFFFFFFF, int(key[1]) & 0xFFFFFFFF
    for _ in range(10):
        p0 = M0 * x
        p1 = M1 * z
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK32
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK32
        x, y, z, w = (hi1 ^ y ^ np.uint64(k0)), lo1, (hi0 ^ w ^ np.uint64(k1)), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return np.stack([x, y, z, w], axis=-1).astype(np.uint32)


def draw(seed, step, stream, idx):
    """Counter layout of csrc/philox.h: {draw index, stream id, step lo, step hi}, key = seed."""
    idx = np.asarray(idx, dtype=np.uint64)
    ctr = np.empty(idx.shape + (4,), dtype=np.uint64)
    ctr[..., 0] = idx
    ctr[..., 1] = stream
    ctr[..., 2] = int(step) & 0xFFFFFFFF
    ctr[..., 3] = (int(step) >> 32) & 0xFFFFFFFF
    seed = int(seed)
    return philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))


def u01(lo, hi):
    v = (np.asarray(hi, dtype=np.uint64) << np.uint64(32)) | np.asarray(lo, dtype=np.uint64)
    return (v >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def u01_open0(lo, hi):
    v = (np.asarray(hi, dtype=np.uint64) << np.uint64(32)) | np.asarray(lo, dtype=np.uint64)
    return ((v >> np.uint64(11)) + np.uint64(1)).astype(np.float64) * (1.0 / 9007199254740992.0)


def normals2(seed, step, stream, idx):
    r = draw(seed, step, stream, idx)
    u1 = u01_open0(r[..., 0], r[..., 1])
    u2 = u01(r[..., 2], r[..., 3])
    rad = np.sqrt(-2.0 * np.log(u1))
    ang = 2.0 * np.pi * u2
    return rad * np.cos(ang), rad * np.sin(ang)


def normals4(seed, step, stream, idx):
    """Four normals from one Philox block, one 32-bit word per uniform (proposal_device.h: normals4): (x, y) -> (g1, g2),
    (z, w) -> (h1, h2); u for the logarithm in (0, 1], for the angle in [0, 1)."""
    r = draw(seed, step, stream, idx).astype(np.float64)
    two32 = 1.0 / 4294967296.0
    ra = np.sqrt(-2.0 * np.log((r[..., 0] + 1.0) * two32))
    rb = np.sqrt(-2.0 * np.log((r[..., 2] + 1.0) * two32))
    aa, ab = 2.0 * np.pi * (r[..., 1] * two32), 2.0 * np.pi * (r[..., 3] * two32)
    return ra * np.cos(aa), ra * np.sin(aa), rb * np.cos(ab), rb * np.sin(ab)


def wavenumber(n, res):
    k = np.arange(n)
    kk = np.where(k < (n + 1) // 2, k, k - n)
    return (kk * (1.0 / (n * res))) * 2.0 * np.pi      # = 2 pi numpy.fft.fftfreq(n, d=res): integer frequency x reciprocal


def amplitude_half(bh, bw, res, model, range_x, range_y, nu):
    """sqrt(S) on ky in [0,bh), kx in [0, bw/2] (formulas of MCMC.py:209-239)."""
    if model == "Gaussian":
        lx, ly = range_x / math.sqrt(3.0), range_y / math.sqrt(3.0)
    elif model == "Exponential":
        lx, ly = range_x / 3.0, range_y / 3.0
    else:
        lx, ly = range_x / 2.0, range_y / 2.0
    a = math.sqrt(lx * ly)
    kx = wavenumber(bw, res)[: bw // 2 + 1]
    ky = wavenumber(bh, res)
    k = np.sqrt(kx[None, :] ** 2 + ky[:, None] ** 2) + 1e-10
    if model == "Gaussian":
        S = np.exp(-0.5 * (a * k) ** 2)
    elif model == "Exponential":
        S = 1.0 / (1.0 + (a * k) ** 2) ** 1.5
    else:
        nu = nu or 1.0
        const = (4.0 * math.pi * math.gamma(nu + 1.0) * (2.0 * nu) ** nu) / (math.gamma(nu) * a ** (2.0 * nu))
        kappa = 2.0 * nu / (a * a)
        S = const * (kappa + 4.0 * math.pi * k * k) ** (-nu - 1.0)
    return np.sqrt(S)


def proposal(seed, step, rf, pairs, masks, centres, W, resolution):
    """One proposal of the device generator.  rf: object with RandField attributes; pairs (2,n) with
    row 0 widths / row 1 heights; masks: list of edge masks; centres: flat ids of region cells.
    Returns dict(size_idx, centre=(row, col), u, scale, nug, range_x, range_y, field=(bh,bw) masked)."""
    n_sizes = pairs.shape[1]
    d = draw(seed, step, STREAM_SCALARS, np.arange(4))
    si = int((int(d[3, 0]) * n_sizes) >> 32)
    scale = (rf.scale_min + (rf.scale_max - rf.scale_min) * float(u01(d[0, 0], d[0, 1]))) / 3.0
    nug = 0.0 + (rf.nugget_max - 0.0) * float(u01(d[0, 2], d[0, 3]))
    range_x = rf.range_min_x + (rf.range_max_x - rf.range_min_x) * float(u01(d[1, 0], d[1, 1]))
    range_y = range_x if rf.isotropic else rf.range_min_y + (rf.range_max_y - rf.range_min_y) * float(u01(d[1, 2], d[1, 3]))
    u_acc = float(u01(d[2, 0], d[2, 1]))
    cw = (int(d[2, 3]) << 32) | int(d[2, 2])
    cell = int(centres[(cw * len(centres)) >> 64])
    bw, bh = int(pairs[0, si]), int(pairs[1, si])
    ncol = bw // 2 + 1

    amp = amplitude_half(bh, bw, resolution, rf.model_name, range_x, range_y, rf.smoothness)
    # rows ky <= bh / 2 draw: the block at counter ky * ncol + kx holds the normals of row ky (first pair) and of row bh - ky (second
    # pair; unused for the two rows that are their own partners)
    hh = bh // 2
    idx = np.arange((hh + 1) * ncol).reshape(hh + 1, ncol)
    a1, a2, b1, b2 = normals4(seed, step, STREAM_SPECTRUM, idx)
    g1 = np.zeros((bh, ncol)); g2 = np.zeros((bh, ncol))
    g1[: hh + 1], g2[: hh + 1] = a1, a2
    for ky in range(1, hh):
        g1[bh - ky], g2[bh - ky] = b1[ky], b2[ky]
    X = amp * (g1 * math.sqrt(0.5)) + 1j * (amp * (g2 * math.sqrt(0.5)))
    kyc = (bh - np.arange(bh)) % bh
    for kx in (0, bw // 2):
        X[:, kx] = amp[:, kx] * (0.5 * (g1[:, kx] + g1[kyc, kx])) + 1j * (amp[:, kx] * (0.5 * (g2[:, kx] - g2[kyc, kx])))
    fld = np.fft.irfft2(X, s=(bh, bw))
    # standardisation (MCMC.py:248).  The device takes the mean from the DC coefficient -- every other term of the inverse
    # DFT sums to zero over the block, so mean(fld) == X[0, 0].real / (bh * bw) up to rounding -- and the population std
    # as sqrt(mean((fld - mean)^2)) like numpy.
    mean = float(X[0, 0].real) / (bh * bw)
    raw_mean = mean
    raw_std = float(np.sqrt(np.mean((fld - mean) ** 2)))
    fld = (fld - mean) / (raw_std + 1e-12)
    fld = fld * scale
    if rf.nugget_max > 0.0:
        o = np.arange(bh * bw)
        n1, n2 = normals2(seed, step, STREAM_NUGGET, o >> 1)
        fld = fld + (np.where(o & 1, n2, n1) * math.sqrt(nug)).reshape(bh, bw)
    return dict(size_idx=si, centre=(cell // W, cell % W), u=u_acc, scale=scale, nug=nug, range_x=range_x,
                range_y=range_y, field=fld * masks[si], raw_mean=raw_mean, raw_std=raw_std)


def field_atol(e):
    """Absolute tolerance for comparing a device field with proposal()'s.  1e-10 of the field scale, plus the
    amplification of DFT rounding noise by the standardisation: when the spectrum is numerically all-DC
    (Gaussian model, range >> block: S underflows for every k != 0) the raw field is a constant plus ~1e-16
    relative rounding noise, and (field-mean)/(std+1e-12) blows that noise up to ~1e-6*scale -- in the
    reference too (MCMC.py:248).  Both sides are then noise of that size, not signal."""
    return e["scale"] * (1e-10 + 8e-15 * abs(e["raw_mean"]) / (e["raw_std"] + 1e-12))
