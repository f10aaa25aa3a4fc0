"""Deterministic synthetic glacier problem of SURVEY.md section 8d (the reference's real inputs --
DenmanDataGridded.csv, sgs_bed_*.txt, data_weight*.txt -- are git-ignored upstream and absent).
Used by bench.py and the examples; builds the template chain + RandField through the public API."""
from __future__ import annotations

import contextlib
import io

import numpy as np

from . import MCMC_gpu


def synthetic_problem(H, W=None, res=500.0):
    W = H if W is None else W
    xx, yy = np.meshgrid(np.arange(W) * res, np.arange(H) * res)
    Lx, Ly = W * res, H * res
    g = np.random.default_rng(1234)
    surf = 2000.0 + 200.0 * np.sin(2 * np.pi * xx / Lx) * np.cos(2 * np.pi * yy / Ly)
    thick = 1000.0 + 150.0 * np.cos(4 * np.pi * xx / Lx) + 100.0 * np.sin(2 * np.pi * yy / Ly)
    bed = surf - thick + g.normal(0, 5, (H, W))
    velx = 200.0 + 50.0 * np.sin(2 * np.pi * yy / Ly)
    vely = 50.0 * np.cos(2 * np.pi * xx / Lx)
    dhdt = g.normal(0, 0.1, (H, W))
    smb = np.full((H, W), 0.2)
    data_mask = np.zeros((H, W), dtype=bool)
    data_mask[::16, :] = True
    region = np.zeros((H, W), dtype=int)
    region[H // 8: 7 * H // 8, W // 8: 7 * W // 8] = 1
    return dict(xx=xx, yy=yy, bed=bed, surf=surf, velx=velx, vely=vely, dhdt=dhdt, smb=smb,
                cond_bed=np.where(data_mask, bed, np.nan), data_mask=data_mask,
                grounded_ice_mask=np.ones((H, W), dtype=bool), region_mask=region, resolution=res)


def initial_beds(prob, n_chains, first=0):
    """Chain 0 starts from `bed`; chain i > 0 from bed + N(0, 5) drawn with default_rng(10000 + i)."""
    H, W = prob["bed"].shape
    out = np.empty((n_chains, H, W))
    for k in range(n_chains):
        i = first + k
        out[k] = prob["bed"] if i == 0 else prob["bed"] + np.random.default_rng(10_000 + i).normal(0, 5, (H, W))
    return out


def template(H, W=None, sigma_mc=5.0, block_range=None, model="Matern", smoothness=0.9125):
    """(problem, chain, RandField) set up with the driver's parameters
    (largeScaleChain_multiprocessing_GPU.py:556-608; nu from T3_LargeScaleChain.ipynb cell 14)."""
    prob = synthetic_problem(H, W)
    with contextlib.redirect_stdout(io.StringIO()):
        ch = MCMC_gpu.chain_crf_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"],
                                    prob["dhdt"], prob["smb"], prob["cond_bed"], prob["data_mask"],
                                    prob["grounded_ice_mask"], prob["resolution"])
        ch.set_update_region(True, prob["region_mask"])
        ch.set_loss_type(sigma_mc=sigma_mc, massConvInRegion=True)
        rf = MCMC_gpu.RandField(10e3, 50e3, 10e3, 50e3, 50, 150, 0, model, True,
                                smoothness=smoothness if model == "Matern" else None)
        lo, hi = block_range or ((8, 16) if min(prob["bed"].shape) < 128 else (50, 80))
        rf.set_block_sizes(lo, hi, lo, hi)
        rf.set_weight_param(2, 0, 6, 1, 49900.0, prob["resolution"])
        rf.set_generation_method(True)
        ch.set_crf_data_weight(rf)
        ch.set_update_type('CRF_weight')
    return prob, ch, rf
