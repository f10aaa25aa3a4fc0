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


# ---- golden F11: the reference DRIVER's own small-scale configuration (smallScaleChain_multiprocessing.py:489-556,
# T4_SmallScaleChain.ipynb cells 11-38): blocks 5-20, Matern variogram with the tutorial's fitted parameters, detrended with a
# Gaussian filter of the initial bed (sigma = 10 cells), QuantileTransformer(1000) fitted on the whole detrended map,
# set_sgs_param(48, 30e3) at 500 m spacing (search half-width 60 cells), sigma_mc = 5 ---------------------------------------
GOLD11 = ROOT / "tests" / "golden" / "f11_sgs_driver_config64.npz"
DRIVER_V1_P = [9932.545836561178, 1.021964658501033, 1.2259010610301213, 0]     # T4 notebook cell 20: range, sill, smoothness, nugget
DRIVER_SGS = (48, 30e3)
DRIVER_BLOCKS = (5, 20, 5, 20)


TIE_FREE_DY = 503.7     # row spacing of the tie-free variants: no two cells of a search window are equidistant from a third


def driver_problem(H=64, dy=500.0):
    """Synthetic stand-in for the driver's private CSV: the standard grid, radar-like conditioning lines (every 8th row, every
    16th column), interior update region.  dy != 500: the rows are dy apart in the COORDINATES the kriging sees (yy), which
    makes the octant search tie-free; the flux stencil keeps `resolution` = 500 as in the driver."""
    prob = orc.synthetic_problem(H, res=500.0)
    if dy != 500.0:
        prob["yy"] = np.ascontiguousarray(np.broadcast_to((np.arange(H) * dy)[:, None], (H, H)))
    data_mask = np.zeros((H, H), dtype=bool)
    data_mask[::8, :] = True
    data_mask[:, ::16] = True
    prob["data_mask"] = data_mask
    prob["cond_bed"] = np.where(data_mask, prob["bed"], np.nan)
    region = np.zeros((H, H), dtype=int)
    region[H // 8: 7 * H // 8, H // 8: 7 * H // 8] = 1
    prob["region_mask"] = region
    return prob


def driver_trend_and_transformer(prob):
    """trend = gaussian_filter(initial_bed, 10) (driver :489); nst = QuantileTransformer(1000, normal, subsample=None,
    random_state=0) fitted on (initial_bed - trend) of the whole map (driver :493-496)."""
    import scipy.ndimage
    from sklearn.preprocessing import QuantileTransformer
    trend = scipy.ndimage.gaussian_filter(prob["bed"], sigma=10)
    nst = QuantileTransformer(n_quantiles=1000, output_distribution="normal", subsample=None,
                              random_state=0).fit((prob["bed"] - trend).reshape((-1, 1)))
    return trend, nst


def driver_cfg(prob, trend, nst, vario_param=None, sgs_param=None, blocks=None, sigma=5.0):
    vp = vario_param or [0, DRIVER_V1_P[3], DRIVER_V1_P[0], DRIVER_V1_P[0], DRIVER_V1_P[1], "Matern", DRIVER_V1_P[2]]
    sp = sgs_param or [DRIVER_SGS[0], DRIVER_SGS[1], False, 0]
    b = blocks or DRIVER_BLOCKS
    H = prob["xx"].shape[0]
    return so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                        prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["region_mask"], prob["resolution"],
                        sigma, vp, sp, b[0], b[1], b[2], b[3], trend=trend, nst_trans=nst)


def driver_chain(prob, trend, nst, seed, vario_param=None, sgs_param=None, blocks=None, sigma=5.0, cls=None):
    """The chain object (product class by default; the fixture generator passes the reference's chain_sgs) set up with the
    driver's calls in the driver's order (smallScaleChain_multiprocessing.py:530-560)."""
    if cls is None:
        from mcmc_gpu_amd import sgs
        cls = sgs.chain_sgs_gpu
    vp = vario_param or [0, DRIVER_V1_P[3], DRIVER_V1_P[0], DRIVER_V1_P[0], DRIVER_V1_P[1], "Matern", DRIVER_V1_P[2]]
    sp = sgs_param or [DRIVER_SGS[0], DRIVER_SGS[1], False, 0]
    b = blocks or DRIVER_BLOCKS
    H = prob["xx"].shape[0]
    ch = cls(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
             prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["resolution"])
    ch.set_update_region(True, prob["region_mask"])
    ch.set_loss_type(sigma_mc=sigma, massConvInRegion=True)
    ch.set_block_sizes(*b)
    ch.set_normal_transformation(nst, do_transform=nst is not None)
    ch.set_trend(trend=trend, detrend_map=trend is not None)
    iso = vp[2] == vp[3]
    ch.set_variogram(vp[5], vp[2] if iso else [vp[2], vp[3]], vp[4], vp[1], isotropic=iso, vario_smoothness=vp[6],
                     vario_azimuth=None if iso else vp[0])
    ch.set_sgs_param(sp[0], sp[1], sgs_rand_dropout_on=False)
    ch.set_random_generator(rng_seed=seed)
    return ch


# ---- golden F12: the reference's module-level MCMC.sgs with ordinary and simple kriging (oracle/make_fixtures_r3b.py) -------------
GOLD12 = ROOT / "tests" / "golden" / "f12_sgs_function_ok_sk.npz"
GOLD13 = GOLD12.parent / "f13_sgs_driver_deep64.npz"
F12_CASES = ("ok", "sk", "skm")


def f12_case(tag):
    """(xx, yy, grid, variogram, keyword arguments of MCMC.sgs, seed) of F12 case `tag`: the standardised synthetic bed of a
    48 x 48 tie-free grid with a 13 x 16 block of NaN cells crossed by two conditioning lines."""
    H = 48
    prob = orc.synthetic_problem(H, res=500.0)
    xx = np.ascontiguousarray(prob["xx"], dtype=np.float64)
    yy = np.ascontiguousarray(np.broadcast_to((np.arange(H) * TIE_FREE_DY)[:, None], (H, H)), dtype=np.float64)
    grid = (prob["bed"] - prob["bed"].mean()) / prob["bed"].std()
    hole = np.zeros((H, H), bool)
    hole[17:30, 9:25] = True
    hole[22, :] = False                     # conditioning lines through the block
    hole[:, 14] = False
    grid = np.where(hole, np.nan, grid)
    if tag == "ok":
        vario = dict(azimuth=0, nugget=0, major_range=9000.0, minor_range=9000.0, sill=1.0, vtype="Exponential")
        return xx, yy, grid, vario, dict(radius=10e3), 1201
    if tag == "sk":
        vario = dict(azimuth=0, nugget=0, major_range=11000.0, minor_range=11000.0, sill=1.0, vtype="Matern", s=1.3)
        return xx, yy, grid, vario, dict(radius=12e3, num_points=32, ktype="sk"), 1202
    if tag == "skm":
        vario = dict(azimuth=35.0, nugget=0.05, major_range=14000.0, minor_range=7000.0, sill=1.3, vtype="Spherical")
        sim_mask = np.zeros((H, H), bool)
        sim_mask[17:30, 9:19] = True        # the rest of the NaN block is not simulated
        return xx, yy, grid, vario, dict(radius=8e3, num_points=16, ktype="sk", sim_mask=sim_mask), 1203
    raise KeyError(tag)
