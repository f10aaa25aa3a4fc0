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


DRIVER_V1_P = (9932.545836561178, 1.021964658501033, 1.2259010610301213, 0.0)   # T4_SmallScaleChain.ipynb cell 20: range, sill, smoothness, nugget


def sgs_template(H, transform=True, light=False):
    """Small-scale chain (chain_sgs_gpu) on the synthetic problem, configured like the reference's small-scale driver
    (smallScaleChain_multiprocessing.py:470-560): radar-like conditioning lines, smooth trend (Gaussian filter of the initial
    bed, sigma 10 cells), QuantileTransformer(1000) fitted on all of (bed - trend), Matern variogram with the tutorial's
    fitted parameters, set_sgs_param(48, 30e3) (search half-width 60 cells at 500 m), blocks 5-20, sigma_mc = 5.
    light=True: the small test configuration of rounds 1-2 (exponential variogram, 16 neighbours within 4 km, blocks 3-8,
    sigma_mc 60, denser data).  transform=False: no normal-score transform / trend.  Returns (problem dict, chain)."""
    from . import sgs
    prob = synthetic_problem(H)
    data_mask = np.zeros((H, H), dtype=bool)
    data_mask[::4 if light else 8, :] = True
    data_mask[:, ::8 if light else 16] = True
    cond = np.where(data_mask, prob["bed"], np.nan)
    region = np.zeros((H, H), dtype=int)
    region[H // 8: 7 * H // 8, H // 8: 7 * H // 8] = 1
    prob.update(data_mask=data_mask, cond_bed=cond, region_mask=region)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           cond, data_mask, np.ones((H, H), dtype=int), prob["resolution"])
    ch.set_update_region(True, region)
    ch.set_loss_type(sigma_mc=60.0 if light else 5.0, massConvInRegion=True)
    nst, trend = None, None
    if transform:
        try:                        # as the driver does (:485-493): smooth trend, transformer fitted on all of (bed - trend)
            from scipy.ndimage import gaussian_filter
            from sklearn.preprocessing import QuantileTransformer
            trend = gaussian_filter(prob["bed"], sigma=10)
            nst = QuantileTransformer(n_quantiles=1000, output_distribution="normal", random_state=0,
                                      subsample=None).fit((prob["bed"] - trend).reshape(-1, 1))
        except ImportError:
            nst, trend = None, None
    ch.set_normal_transformation(nst, do_transform=nst is not None)
    ch.set_trend(trend, detrend_map=trend is not None)
    if light:
        sill = 1.0 if nst is not None else float(np.var(prob["bed"]))      # normal scores have unit variance
        ch.set_variogram("Exponential", 6000.0, sill, 0.0, isotropic=True)
        ch.set_sgs_param(16, 4000.0)
        ch.set_block_sizes(3, 8, 3, 8)
    else:
        sill = DRIVER_V1_P[1] if nst is not None else float(np.var(prob["bed"])) * DRIVER_V1_P[1]
        ch.set_variogram("Matern", DRIVER_V1_P[0], sill, DRIVER_V1_P[3], isotropic=True, vario_smoothness=DRIVER_V1_P[2])
        ch.set_sgs_param(48, 30e3, sgs_rand_dropout_on=False)
        ch.set_block_sizes(5, 20, 5, 20)
    return prob, ch
