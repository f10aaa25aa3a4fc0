"""Shared setup of the small-scale-chain tests: the synthetic problem of golden F10 (oracle/make_fixtures.sgs_problem), the
oracle configuration and the product chain for one of its two variants."""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import mcmc_oracle as orc  # noqa: E402
import sgs_oracle as so  # noqa: E402

GOLD = ROOT / "tests" / "golden" / "f10_sgs_chain32.npz"
VARIANTS = {"a": ("Exponential", None, False, False), "b": ("Matern", 1.5, True, True)}


def problem(H):
    prob = orc.synthetic_problem(H, res=500.0)
    data_mask = np.zeros((H, H), dtype=bool)
    data_mask[::4, :] = True
    data_mask[:, ::8] = True
    prob["data_mask"] = data_mask
    prob["cond_bed"] = np.where(data_mask, prob["bed"], np.nan)
    region = np.zeros((H, H), dtype=int)
    region[H // 8: 7 * H // 8, H // 8: 7 * H // 8] = 1
    prob["region_mask"] = region
    prob["trend"] = prob["surf"] - 1000.0 - 150.0 * np.cos(4 * np.pi * prob["xx"] / (H * 500.0))
    return prob


def setup(tag):
    """(golden arrays, problem, oracle SgsConfig, product chain_sgs_gpu) of F10 variant `tag`."""
    from mcmc_gpu_amd import sgs
    g = np.load(GOLD, allow_pickle=False)
    vtype, smooth, use_trend, use_nst = VARIANTS[tag]
    H = int(g["H"])
    prob = problem(H)
    trend = prob["trend"] if use_trend else None
    nst = None
    if use_nst:
        from sklearn.preprocessing import QuantileTransformer
        data = (prob["cond_bed"] - trend)[prob["data_mask"]].reshape(-1, 1)
        nst = QuantileTransformer(n_quantiles=200, output_distribution="normal", random_state=152).fit(data)
    sill, seed, sigma = float(g[f"{tag}_sill"]), int(g[f"{tag}_seed"]), float(g[f"{tag}_sigma_mc"])
    rr, rad, npts = float(g["range"]), float(g["radius"]), int(g["num_points"])
    grounded = np.ones((H, H), dtype=int)
    cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                       prob["cond_bed"], prob["data_mask"], grounded, prob["region_mask"], prob["resolution"], sigma,
                       [0, 0.0, rr, rr, sill, vtype, smooth], [npts, rad, False, 0], 3, 8, 3, 8, trend=trend, nst_trans=nst)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           prob["cond_bed"], prob["data_mask"], grounded, prob["resolution"])
    ch.set_update_region(True, prob["region_mask"])
    ch.set_loss_type(sigma_mc=sigma, massConvInRegion=True)
    ch.set_normal_transformation(nst, do_transform=use_nst)
    ch.set_trend(trend, detrend_map=use_trend)
    ch.set_variogram(vtype, rr, sill, 0.0, isotropic=True, vario_smoothness=smooth)
    ch.set_sgs_param(npts, rad)
    ch.set_block_sizes(3, 8, 3, 8)
    ch.set_random_generator(rng_seed=seed)
    return g, prob, cfg, ch
