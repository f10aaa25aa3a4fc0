"""Precomputed-factor ("L z") proposal generator: host-side setup.

The reference ships the ingredients only -- covariance models on normalised lag
(gstatsim_custom/covariance.py:4-28), rotation matrix and dense assembly (gstatsim_custom/_krige.py:83-122) -- and
names the Cholesky/LU generator as future work (README.md:21-23).  Here the covariance of every block size and range
class is assembled on the device (gsm_cov_assemble), factorised once (setup: torch.linalg.cholesky on the GPU), and
handed to libgsm_hip (gsm_set_factors); the per-step work -- z draws and the batched L z product on the fp64 matrix
cores -- is cholesky_kernel.hip.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from ._lib import VTYPE_IDS, Vario
from .engine import _ptr

MODEL_TO_VTYPE = {"Exponential": "exponential", "Gaussian": "gaussian", "Matern": "matern", "Spherical": "spherical"}


def make_vario(vtype, major_range, minor_range, azimuth=0.0, sill=1.0, nugget=0.0, s=None) -> dict:
    """The `vario` dict of gstatsim_custom (keys as in _krige.py:83-122)."""
    d = dict(azimuth=float(azimuth), nugget=float(nugget), major_range=float(major_range),
             minor_range=float(minor_range), sill=float(sill), vtype=str(vtype))
    if s is not None:
        d["s"] = float(s)
    return d


def vario_struct(v: dict) -> Vario:
    out = Vario()
    out.azimuth, out.major_range, out.minor_range = v["azimuth"], v["major_range"], v["minor_range"]
    out.sill, out.nugget, out.s = v["sill"], v["nugget"], float(v.get("s") or 0.0)
    out.vtype = VTYPE_IDS[v["vtype"].lower()]
    return out


def rotation_matrix(v: dict) -> np.ndarray:
    th = (v["azimuth"] / 180.0) * np.pi
    return np.dot(np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]),
                  np.array([[1 / v["major_range"], 0], [0, 1 / v["minor_range"]]]))


def matern_lag_table(bh, bw, res, v: dict) -> np.ndarray:
    """Matern covariance at every distinct lag (di, dj) of a bh x bw block: (2bh-1, 2bw-1) values with
    scipy.special.kv -- the function covariance.py:17-22 evaluates on all N^2 pairs (19 s at N = 6400);
    a regular grid has only (2bh-1)(2bw-1) distinct lags."""
    from scipy.special import gamma, kv
    R = rotation_matrix(v)
    di = np.arange(-(bh - 1), bh)[:, None] * res
    dj = np.arange(-(bw - 1), bw)[None, :] * res
    m0 = dj * R[0, 0] + di * R[1, 0]
    m1 = dj * R[0, 1] + di * R[1, 1]
    h = np.sqrt(m0 * m0 + m1 * m1)
    s = v["s"]
    sc = 0.45246434 * np.exp(-0.70449189 * s) + 1.7863836
    hh = np.where(h == 0.0, 1e-8, h)
    c = (v["sill"] - v["nugget"]) * 2 / gamma(s) * np.power(sc * hh * np.sqrt(s), s) * kv(s, 2 * sc * hh * np.sqrt(s))
    return np.where(np.isnan(c), v["sill"] - v["nugget"], c)


def cov_assemble(eng, bh, bw, res, v: dict, ld=None) -> torch.Tensor:
    """Dense covariance (N, ld) of a block on the device (gsm_cov_assemble)."""
    N = bh * bw
    ld = N if ld is None else int(ld)
    sigma = torch.zeros((N, ld), dtype=torch.float64, device=eng.dev)
    table = None
    if v["vtype"].lower() == "matern":
        table = torch.as_tensor(np.ascontiguousarray(matern_lag_table(bh, bw, res, v))).to(eng.dev)
    vs = vario_struct(v)
    with torch.cuda.device(eng.dev):
        eng._check(eng.lib.gsm_cov_assemble(eng.h, int(bh), int(bw), float(res), C.byref(vs), _ptr(table), _ptr(sigma),
                                            ld, eng._stream()))
    return sigma


def factor_upper_padded(eng, sigma: torch.Tensor, jitter: float, use_torch: bool = False) -> torch.Tensor:
    """U = chol(Sigma + jitter I)^T zero-padded to [Npad, Npad], Npad = N rounded up to 64 (setup time).
    Default: the library's own blocked Cholesky (gsm_cholesky_upper); use_torch=True cross-checks with
    torch.linalg.cholesky."""
    N = sigma.shape[0]
    Np = (N + 63) // 64 * 64
    if use_torch:
        A = sigma[:, :N].clone()
        A.diagonal().add_(jitter)
        U = torch.zeros((Np, Np), dtype=torch.float64, device=sigma.device)
        U[:N, :N] = torch.linalg.cholesky(A).T
        return U.contiguous()
    U = torch.zeros((Np, Np), dtype=torch.float64, device=sigma.device)
    U[:N, :N] = sigma[:, :N]
    if Np > N:
        idx = torch.arange(N, Np, device=sigma.device)
        U[idx, idx] = 1.0 - jitter               # identity block in the padding (factor = 1, removed below)
    with torch.cuda.device(eng.dev):
        eng._check(eng.lib.gsm_cholesky_upper(eng.h, _ptr(U), Np, Np, float(jitter), eng._stream()))
    if Np > N:
        U[idx, idx] = 0.0
    return U


def class_ranges(rf, n_classes):
    """Mid-points of n_classes equal range bins of the RandField ranges (x -> major, y -> minor)."""
    out = []
    for r in range(n_classes):
        t = (r + 0.5) / n_classes
        rx = rf.range_min_x + t * (rf.range_max_x - rf.range_min_x)
        ry = rx if rf.isotropic else rf.range_min_y + t * (rf.range_max_y - rf.range_min_y)
        out.append((rx, ry))
    return out


def class_varios(rf, n_classes):
    return [make_vario(MODEL_TO_VTYPE[rf.model_name], rx, ry, s=rf.smoothness) for rx, ry in class_ranges(rf, n_classes)]


def build_factors(eng, rf, n_classes=1, jitter=1e-8):
    """Assemble + factorise every (block size, range class) and register the factors with the engine.
    Returns the list of U tensors (the caller keeps them alive) in the order gsm_set_factors expects."""
    varios = class_varios(rf, n_classes)
    factors = []
    for i in range(eng.n_sizes):
        bh, bw = int(eng.bh[i]), int(eng.bw[i])
        for v in varios:
            sigma = cov_assemble(eng, bh, bw, rf.resolution, v)
            factors.append(factor_upper_padded(eng, sigma, jitter * v["sill"]))
            del sigma
    ptrs = (C.c_void_p * len(factors))(*[f.data_ptr() for f in factors])
    with torch.cuda.device(eng.dev):
        eng._check(eng.lib.gsm_set_factors(eng.h, int(n_classes), ptrs, eng._stream()))
    eng._factors = factors
    eng.n_classes = n_classes
    return factors
