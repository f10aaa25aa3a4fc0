"""Host-side mirror of the reference's SMALL-scale chain (gstatsMCMC/MCMC.py namespace):

    chain_sgs                      gstatsMCMC/MCMC.py:1445-1911   -> chain_sgs_gpu
    init_msc_chain_by_instance     gstatsMCMC/MCMC.py:402-431
    sgs / neighbors / ok_solve / sk_solve   MCMC.py:91-173, gstatsim_custom/neighbors.py:4-64, _krige.py:5-81  -> sgs, gsm_sgs_blocks (HIP)

Same class / setter names, argument meaning and return tuple as the reference.  Per iteration the host draws what the
reference draws from the chain's NumPy generator, in its order (block centre by rejection, block sizes, the shuffle of
the block's cells, one normal per simulated cell, the accept uniform -- none of which depends on the chain's state) and
the device does the work: the sequential Gaussian simulation of the block (octant search + ordinary kriging per cell),
the full-grid mass-conservation loss and thickness guard of the proposed bed, and the commit.  All chains of a call
share one libgsm_hip handle (run_many_sgs); chain_sgs_gpu.run is the one-chain case.

Normal-score transform (do_transform): the transformer is a caller-supplied object (scikit-learn's QuantileTransformer
in the reference's drivers).  Its transform / inverse_transform are called on the host once per iteration on the whole
map, exactly where the reference calls them (MCMC.py:1766, :1777); simulation and loss still run on the device.

Numerics: the kriging systems are solved by pivoted elimination on the device where the reference calls
numpy.linalg.lstsq, so simulated values agree with the CPU chain to ~1e-9 of the bed's scale, not bit for bit; accept
decisions are identical unless an accept uniform falls within that distance of its threshold.
Kept reference behaviour: the loop runs n_iter times and overwrites the record of the initial
state at index 0.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
import time
from copy import deepcopy

import numpy as np

__all__ = ["chain_sgs_gpu", "init_msc_chain_by_instance", "run_many_sgs", "sgs", "cov_norm", "lag_cov_table"]


def cov_norm(h, vtype, sill, nugget, s=None):
    """Covariance models of gstatsim_custom on normalised lag (covariance.py:4-28), incl. the spherical model's
    `sill - 1` beyond the range and Matern's h == 0 -> 1e-8 substitution (the input is not mutated here)."""
    vtype = vtype.lower()
    if vtype == "exponential":
        return (sill - nugget) * np.exp(-3 * h)
    if vtype == "gaussian":
        return (sill - nugget) * np.exp(-3 * np.square(h))
    if vtype == "spherical":
        c = sill - nugget - 1.5 * h + 0.5 * np.power(h, 3)
        return np.where(h > 1, sill - 1, c)
    if vtype == "matern":
        from scipy.special import gamma, kv
        sc = 0.45246434 * np.exp(-0.70449189 * s) + 1.7863836
        hh = np.where(h == 0.0, 1e-8, h)
        c = (sill - nugget) * 2 / gamma(s) * np.power(sc * hh * np.sqrt(s), s) * kv(s, 2 * sc * hh * np.sqrt(s))
        return np.where(np.isnan(c), sill - nugget, c)
    raise ValueError("vtype must be Exponential, Gaussian, Spherical or Matern")


def rotation_matrix(v: dict) -> np.ndarray:
    """Anisotropy rotation x scaling (make_rotation_matrix, _krige.py:83-103)."""
    th = (v["azimuth"] / 180.0) * np.pi
    return np.dot(np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]]),
                  np.array([[1 / v["major_range"], 0], [0, 1 / v["minor_range"]]]))


def lag_cov_table(vario: dict, hw: int, dx: float, dy: float, mi: int | None = None, mj: int | None = None) -> np.ndarray:
    """Covariance at every integer lag (di, dj), |di| <= mi, |dj| <= mj (default 2 hw: two neighbours of one cell), between
    two cells of an axis-aligned grid with column spacing dx and row spacing dy (signed): what make_sigma / make_rho
    (_krige.py:105-143) evaluate pair by pair."""
    mi = 2 * int(hw) if mi is None else int(mi)
    mj = 2 * int(hw) if mj is None else int(mj)
    R = rotation_matrix(vario)
    di = np.arange(-mi, mi + 1)[:, None] * dy
    dj = np.arange(-mj, mj + 1)[None, :] * dx
    m0 = dj * R[0, 0] + di * R[1, 0]
    m1 = dj * R[0, 1] + di * R[1, 1]
    h = np.sqrt(m0 * m0 + m1 * m1)
    return np.ascontiguousarray(cov_norm(h, vario["vtype"], vario["sill"], vario["nugget"], vario.get("s")))


def lag_extents(hw: int, H: int, W: int) -> tuple[int, int]:
    """Extents of the lag table handed to gsm_sgs_blocks: the whole grid while that stays small (4 M lags = 32 MiB), so that
    the radius-widening fallback (MCMC.py:150-156) finds every lag; else what a search window needs."""
    if (2 * H - 1) * (2 * W - 1) <= (1 << 22):
        return H - 1, W - 1
    return min(2 * int(hw), H - 1), min(2 * int(hw), W - 1)


def _axes(xx, yy):
    xs, ys = np.ascontiguousarray(xx[0, :], dtype=np.float64), np.ascontiguousarray(yy[:, 0], dtype=np.float64)
    if not (np.array_equal(xx, np.broadcast_to(xs[None, :], xx.shape)) and np.array_equal(yy, np.broadcast_to(ys[:, None], yy.shape))):
        raise NotImplementedError("the device SGS needs an axis-aligned grid (xx[i, j] = x[j], yy[i, j] = y[i])")
    dx, dy = xs[1] - xs[0], ys[1] - ys[0]
    if not (np.allclose(np.diff(xs), dx, rtol=1e-9, atol=0) and np.allclose(np.diff(ys), dy, rtol=1e-9, atol=0)):
        raise NotImplementedError("the device SGS needs uniform grid spacing")
    return xs, ys, float(dx), float(dy)


def sgs(xx, yy, grid, variogram, radius=100e3, num_points=20, ktype='ok', sim_mask=None, quiet=False, stencil=None, rcond=None,
        seed=None, device=None):
    """Sequential Gaussian simulation with ordinary ('ok') or simple ('sk') kriging -- the reference's module-level MCMC.sgs
    (MCMC.py:91-173 with _preprocess :42-88), same arguments and return value, executed by gsm_sgs_blocks.  The generator is
    consumed exactly as the reference consumes it: one shuffle of the cells of sim_mask, then one normal per simulated cell.
    Limits of the device path (NotImplementedError otherwise): the NaN cells to simulate lie within one window of at most 1024
    cells (the small-scale chain's blocks; MCMC.py:1762-1774), the circular search stencil and lstsq's default rcond,
    scalar variogram parameters, 8 <= num_points <= 48, an axis-aligned uniform grid."""
    import torch
    from .engine import GsmEngine
    for name, a in (("xx", xx), ("yy", yy), ("grid", grid)):
        if not isinstance(a, np.ndarray) or a.ndim != 2:
            raise ValueError(f"{name} must be a 2D NumPy array")                  # _sanity_checks, interpolate.py:282-298
    if xx.shape != yy.shape or xx.shape != grid.shape:
        raise ValueError("xx, yy, and grid must have same shape")
    for key in ("major_range", "minor_range", "azimuth", "sill", "nugget", "vtype"):
        if key not in variogram:
            raise ValueError(f"Missing variogram parameter {key}")
    if variogram["vtype"].lower() == "matern" and "s" not in variogram:
        raise ValueError("Missing variogram parameter s for Matern covariance")
    if ktype not in ("ok", "sk"):
        raise ValueError("ktype must be 'ok' or 'sk'")
    if stencil is not None or rcond is not None:
        raise NotImplementedError("the device SGS searches the circular stencil and solves with lstsq's default cut-off (rcond=None)")
    if any(not isinstance(variogram[k], (int, float, np.integer, np.floating)) for k in variogram if k != "vtype"):
        raise NotImplementedError("the device SGS takes scalar variogram parameters (one covariance table per call)")
    if not 8 <= int(num_points) <= 48:
        raise NotImplementedError("the device SGS takes 8 <= num_points <= 48")
    if seed is None:                                                                  # utilities.get_random_generator, :50-70
        rng = np.random.default_rng()
    elif isinstance(seed, int):
        rng = np.random.default_rng(seed=seed)
    elif isinstance(seed, np.random.Generator):
        rng = seed
    else:
        raise ValueError("Seed should be an integer, a NumPy random Generator, or None")
    grid = np.asarray(grid, dtype=np.float64)
    H, W = grid.shape
    cond_msk = ~np.isnan(grid)
    if sim_mask is None:
        sim_mask = np.full(xx.shape, True)
    ii, jj = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    inds = np.array([ii[sim_mask].flatten(), jj[sim_mask].flatten()]).T
    global_mean = np.mean(grid[cond_msk])                                          # MCMC.py:81
    rng.shuffle(inds)                                                                 # MCMC.py:128
    need = ~cond_msk[inds[:, 0], inds[:, 1]] if inds.shape[0] else np.zeros(0, bool)
    todo = inds[need]
    out = grid.copy()
    if todo.shape[0] == 0:
        return out
    r0, r1, c0, c1 = int(todo[:, 0].min()), int(todo[:, 0].max()) + 1, int(todo[:, 1].min()), int(todo[:, 1].max()) + 1
    if (r1 - r0) * (c1 - c0) > 1024:
        raise NotImplementedError("the device SGS simulates one block: the cells to simulate must fit a window of at most 1024 cells")
    # listed cells: the cells to simulate in visiting order, then every other window cell that holds a value (conditioning data:
    # never simulated, seen by the search from the start); a window cell that stays NaN is not listed
    inside = np.zeros((H, W), bool); inside[r0:r1, c0:c1] = True
    todo_m = np.zeros((H, W), bool); todo_m[todo[:, 0], todo[:, 1]] = True
    rest = np.argwhere(inside & cond_msk & ~todo_m)
    cells = np.ascontiguousarray(np.concatenate([todo, rest]), dtype=np.int32)
    z = np.zeros(cells.shape[0])
    z[:todo.shape[0]] = rng.standard_normal(todo.shape[0])                           # rng.normal(est, sd, 1) = est + sd * normal, :165
    xs, ys, dx, dy = _axes(np.asarray(xx, dtype=np.float64), np.asarray(yy, dtype=np.float64))
    vario = {k: (variogram[k] if k == "vtype" else float(variogram[k])) for k in variogram}
    hw = int(math.ceil(float(radius) / abs(dx)))
    eng = GsmEngine(H, W, 1, device)
    try:
        dev, lib, h = eng.dev, eng.lib, eng.h
        f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        i32 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.int32)).to(dev)
        mi, mj = lag_extents(hw, H, W)
        d_grid, d_xs, d_ys, d_lag = f64(grid[None]), f64(xs), f64(ys), f64(lag_cov_table(vario, hw, dx, dy, mi, mj))
        d_win, d_off, d_cells, d_z = i32([[r0, r1, c0, c1]]), i32([0, cells.shape[0]]), i32(cells), f64(z)
        d_gm = f64([global_mean])
        with torch.cuda.device(dev):
            eng._check(lib.gsm_sgs_set_kriging(h, 1 if ktype == "sk" else 0, _ptr(d_gm)))
            eng._check(lib.gsm_sgs_blocks(h, _ptr(d_grid), None, _ptr(d_win), _ptr(d_xs), _ptr(d_ys), _ptr(d_lag), mi, mj, hw, float(radius),
                                          int(num_points), float(vario["sill"]), _ptr(d_off), _ptr(d_cells), _ptr(d_z), int(cells.shape[0]),
                                          None, None, eng._stream()))
        out = d_grid[0].cpu().numpy()
    finally:
        eng.close()
    return out


class chain_sgs_gpu:
    """Small-scale (SGS block) Metropolis chain executed on the MI355X (reference chain_sgs, MCMC.py:1445-1911)."""

    def __init__(self, xx, yy, initial_bed, surf, velx, vely, dhdt, smb, cond_bed, data_mask, grounded_ice_mask, resolution):
        self.xx, self.yy = xx, yy
        self.initial_bed = initial_bed
        self.surf, self.velx, self.vely, self.dhdt, self.smb = surf, velx, vely, dhdt, smb
        self.cond_bed = cond_bed
        self.data_mask = data_mask
        self.grounded_ice_mask = grounded_ice_mask
        self.resolution = resolution
        self.loss_function_list = []
        self.sample_loc = None
        shp = initial_bed.shape
        if any(a.shape != shp for a in (surf, velx, vely, dhdt, smb, cond_bed, data_mask)):
            raise Exception('the shape of bed, surf, velx, vely, dhdt, smb, radar_bed, data_mask need to be same')
        self.do_transform = False
        self.nst_trans = None
        self.trend = None
        self.detrend_map = False

    # ---- setters shared with the large-scale chain (MCMC.py:849-872, :950-1018) ---------------------------------------
    def set_update_region(self, update_in_region, region_mask=[]):
        self.update_in_region = update_in_region
        if update_in_region is False:
            self.region_mask = np.full(self.xx.shape, 1)
        else:
            if np.shape(region_mask) != self.xx.shape:
                raise ValueError('the region_mask input is invalid. It has to be a 2D numpy array with the shape of the map')
            self.region_mask = region_mask

    def set_loss_type(self, sigma_mc=-1, massConvInRegion=True):
        self.mc_region_mask = self.region_mask if massConvInRegion else np.full(self.xx.shape, 1)
        self.sigma_mc = sigma_mc

    def set_sample_points_locations(self, loc):
        self.sample_loc = loc

    # ---- chain_sgs setters (MCMC.py:1466-1598) ------------------------------------------------------------------------
    def set_normal_transformation(self, nst_trans, do_transform=True):
        self.do_transform = do_transform
        self.nst_trans = nst_trans if do_transform else None

    def set_trend(self, trend=None, detrend_map=True):
        if detrend_map == True:  # noqa: E712
            if trend is None or len(trend) != len(self.xx) or trend.shape != self.xx.shape:
                raise ValueError('if detrend_map is set to True, then the trend of the topography, which is a 2D numpy array, must be provided')
            self.trend = trend
        else:
            self.trend = None
        self.detrend_map = detrend_map

    def set_variogram(self, vario_type, vario_range, vario_sill, vario_nugget, isotropic=True, vario_smoothness=None,
                      vario_azimuth=None):
        if vario_type in ('Gaussian', 'Exponential', 'Spherical'):
            pass
        elif vario_type == 'Matern':
            if (vario_smoothness is None) or (vario_smoothness <= 0):
                raise ValueError('vario_smoothness argument should be a positive float when the vario_type is Matern')
        else:
            raise ValueError('vario_type argument should be one of the following: Gaussian, Exponential, Spherical, or Matern')
        self.vario_type = vario_type
        if isotropic:
            self.vario_param = [0, vario_nugget, vario_range, vario_range, vario_sill, vario_type, vario_smoothness]
        else:
            if len(vario_range) != 2:
                raise ValueError("vario_range need to be a list with two floats to specifying for major range and minor range of the variogram when isotropic is set to False")
            self.vario_param = [vario_azimuth, vario_nugget, vario_range[0], vario_range[1], vario_sill, vario_type, vario_smoothness]

    def set_sgs_param(self, sgs_num_nearest_neighbors, sgs_searching_radius, sgs_rand_dropout_on=False, dropout_rate=0):
        if sgs_rand_dropout_on == False:  # noqa: E712
            dropout_rate = 0
        self.sgs_param = [sgs_num_nearest_neighbors, sgs_searching_radius, sgs_rand_dropout_on, dropout_rate]

    def set_block_sizes(self, block_min_x, block_max_x, block_min_y, block_max_y):
        self.block_min_x, self.block_min_y = block_min_x, block_min_y
        self.block_max_x, self.block_max_y = block_max_x, block_max_y

    def set_random_generator(self, rng_seed=None):
        if rng_seed is None:
            rng = np.random.default_rng()
        elif isinstance(rng_seed, (int, np.integer)):
            rng = np.random.default_rng(seed=int(rng_seed))
            self.rng_seed = int(rng_seed)
        elif isinstance(rng_seed, np.random.Generator):
            rng = rng_seed
        else:
            raise ValueError('Seed should be an integer, a NumPy random Generator, or None')
        self.rng = rng

    def set_rng_mode(self, mode):
        """'replay' (default): the draws come from chain.rng in the reference's order (MCMC.py:1750-1797) -- accept masks and
        beds follow the reference on the same seed.  'philox': the draws are made on the device from Philox4x32-10 counters
        keyed by the chain's seed (gsm_sgs_draw_philox) -- no host work per iteration; a chain of its own definition, restated
        by oracle/sgs_philox_oracle.py.  'pcg64': the draws of 'replay' -- chain.rng's own NumPy PCG64 stream, bit for bit -- made on
        the device (gsm_sgs_draw_pcg64): the reference's chain on the same seed without host work per iteration."""
        if mode not in ('replay', 'philox', 'pcg64'):
            raise ValueError("rng mode must be 'replay', 'pcg64' or 'philox'")
        self.rng_mode = mode
        self.philox_iter = 0

    def _philox_seed(self):
        seed = getattr(self, 'rng_seed', None)
        if seed is None:
            seed = int(self.rng.bit_generator.seed_seq.entropy) if hasattr(self.rng.bit_generator, 'seed_seq') else 0
        return int(seed) & 0xFFFFFFFFFFFFFFFF

    def loss(self, massConvResidual, dataDiff):
        loss_mc = np.nansum(np.square(massConvResidual[self.mc_region_mask == 1])) / (2 * self.sigma_mc ** 2)
        return loss_mc + 0, loss_mc, 0

    def _vario(self):
        vp = self.vario_param
        v = dict(azimuth=vp[0], nugget=vp[1], major_range=vp[2], minor_range=vp[3], sill=vp[4], vtype=vp[5])
        if vp[5] == 'Matern':
            v['s'] = vp[6]
        return v

    # ---- host draws of one iteration (MCMC.py:1747-1760, sgs :128, :165, run :1799) -------------------------------------
    def _draw_iteration(self, rng, cond_is_data):
        H, W = self.xx.shape
        while True:
            ix = rng.integers(low=0, high=H, size=1)[0]
            iy = rng.integers(low=0, high=W, size=1)[0]
            if self.region_mask[ix, iy] == 1:
                break
        bsx = rng.integers(low=self.block_min_x, high=self.block_max_x, size=1)[0]
        bsy = rng.integers(low=self.block_min_y, high=self.block_max_y, size=1)[0]
        r0 = max(0, int(ix - bsx / 2)); r1 = min(H, int(ix + bsx / 2))
        c0 = max(0, int(iy - bsy / 2)); c1 = min(W, int(iy + bsy / 2))
        ii, jj = np.meshgrid(np.arange(r0, r1), np.arange(c0, c1), indexing='ij')
        inds = np.array([ii.flatten(), jj.flatten()]).T
        rng.shuffle(inds)
        need = ~cond_is_data[inds[:, 0], inds[:, 1]] if inds.shape[0] else np.zeros(0, bool)
        z = np.zeros(inds.shape[0])
        if need.any():
            z[need] = rng.standard_normal(int(need.sum()))     # rng.normal(est, sd, 1) = est + sd * standard normal
        u = rng.random()
        return (ix, iy, bsx, bsy), (r0, r1, c0, c1), np.ascontiguousarray(inds, dtype=np.int32), z, u

    def run(self, n_iter, only_save_last_bed=False, info_per_iter=100, plot=True, progress_bar=True):
        """n_iter SGS-block Metropolis iterations from self.initial_bed; returns the reference's tuple (bed or bed_cache,
        loss_mc_cache, loss_data_cache, loss_cache, step_cache, resampled_times, blocks_cache[, sample_values])."""
        if not hasattr(self, 'rng'):
            self.set_random_generator(getattr(self, 'rng_seed', None))
        mode = getattr(self, 'rng_mode', 'replay')
        philox = mode == 'philox'
        out, _ = run_many_sgs(self, [self.initial_bed], [self.rng], n_iter, only_save_last_bed=only_save_last_bed,
                              info_per_iter=info_per_iter, progress_bar=progress_bar,
                              philox_seeds=[self._philox_seed()] if philox else None, philox_iter0=getattr(self, 'philox_iter', 0),
                              pcg64=(mode == 'pcg64'))
        if philox:
            self.philox_iter += int(n_iter)
        return out[0]


LAST_GRAPH_REPLAYS = 0      # batches of the last device-draw run_many_sgs call that were hipGraph launches (gsm_sgs_iterate)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def run_many_sgs(chain, initial_beds, rngs, n_iter, only_save_last_bed=True, info_per_iter=100, progress_bar=None, device=None,
                 philox_seeds=None, philox_iter0=0, pcg64=False):
    """n small-scale chains of one template (same static fields, variogram, block sizes) in ONE handle.  rngs: one NumPy
    Generator per chain (consumed exactly as chain_sgs.run consumes chain.rng).  philox_seeds (one 64-bit key per chain): Philox
    mode -- the draws of iterations philox_iter0 .. are made on the device and rngs are not touched.  pcg64=True: the draws of
    replay mode (rngs' own PCG64 streams, bit for bit) are made on the device and the generators are left where NumPy would leave
    them.
    Returns (list of result tuples, rngs)."""
    import torch
    from .engine import GsmEngine
    H, W = chain.xx.shape
    n = len(initial_beds)
    if len(rngs) != n:
        raise ValueError('need one random generator per chain')
    n_iter = int(n_iter)
    xs, ys, dx, dy = _axes(np.asarray(chain.xx, dtype=np.float64), np.asarray(chain.yy, dtype=np.float64))
    rad, npts = float(chain.sgs_param[1]), int(chain.sgs_param[0])
    hw = int(math.ceil(rad / abs(dx)))
    vario = chain._vario()
    detrend = bool(chain.detrend_map)
    trend = np.asarray(chain.trend, dtype=np.float64) if detrend else None
    nst = chain.nst_trans if chain.do_transform else None
    cond_c = np.asarray(chain.cond_bed, dtype=np.float64) - trend if detrend else np.asarray(chain.cond_bed, dtype=np.float64).copy()
    z_cond = nst.transform(cond_c.reshape(-1, 1)).reshape(H, W) if nst is not None else cond_c
    cond_is_data = ~np.isnan(z_cond)
    # scikit-learn's QuantileTransformer with normal output and one feature (what the reference's drivers attach,
    # smallScaleChain_multiprocessing.py:493-496) runs on the device (gsm_qt_transform); any other transformer object is
    # called on the host once per iteration, where the reference calls it
    dev_qt = (nst is not None and type(nst).__name__ == 'QuantileTransformer' and getattr(nst, 'output_distribution', None) == 'normal'
              and getattr(nst, 'quantiles_', None) is not None and nst.quantiles_.ndim == 2 and nst.quantiles_.shape[1] == 1
              and os.environ.get('GSM_SGS_HOST_TRANSFORM', '0') != '1')
    host_nst = nst if (nst is not None and not dev_qt) else None
    track = chain.sample_loc is not None
    keep_all = not only_save_last_bed

    eng = GsmEngine(H, W, n, device)
    try:
        dev = eng.dev
        eng.set_static(chain.surf, chain.velx, chain.vely, chain.dhdt, chain.smb, None, chain.grounded_ice_mask,
                       chain.mc_region_mask, chain.resolution, chain.sigma_mc)
        f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
        d_xs, d_ys = f64(xs), f64(ys)
        lag_mi, lag_mj = lag_extents(hw, H, W)
        d_lag = f64(lag_cov_table(vario, hw, dx, dy, lag_mi, lag_mj))
        max_cells = min(1024, max(1, (int(chain.block_max_x) - 1) * (int(chain.block_max_y) - 1)))
        d_zcond = f64(z_cond)
        d_trend = f64(trend) if detrend else None
        bed_c = np.stack([np.asarray(b, dtype=np.float64) - trend if detrend else np.asarray(b, dtype=np.float64) for b in initial_beds])
        cur = f64(bed_c)
        nxt = cur.clone()
        if dev_qt:
            d_q = f64(nst.quantiles_[:, 0]); d_ref = f64(nst.references_); nq = int(d_q.numel())
            prop = cur.clone()                                   # proposed beds in data space (inverse transform of nxt)
        def qt(src, dst, inverse):
            eng._check(lib.gsm_qt_transform(h, _ptr(d_q), _ptr(d_ref), nq, _ptr(src), _ptr(dst), int(src.numel()), int(inverse), eng._stream()))
        resampled = torch.zeros((n, H, W), dtype=torch.int32, device=dev)
        d_loss = torch.empty(n, dtype=torch.float64, device=dev)
        d_bad = torch.empty(n, dtype=torch.int32, device=dev)
        lib, h = eng.lib, eng.h

        def loss_of(t):
            eng._check(lib.gsm_sgs_loss(h, _ptr(t), _ptr(d_trend), _ptr(d_loss), _ptr(d_bad), eng._stream()))
            return d_loss.cpu().numpy().copy(), d_bad.cpu().numpy().copy()

        loss_prev, _ = loss_of(cur)
        # no transformer: only the block and its one-cell halo change per iteration -> carried squared residuals, windowed loss,
        # and loss / guard / acceptance test / commit in ONE launch (gsm_sgs_finish); the reference recomputes the whole map
        # (gsm_sgs_finish keeps block + halo in LDS: 36 x 36 cells; a longer, thinner block takes the whole-map path)
        windowed = (nst is None and os.environ.get('GSM_SGS_WINDOWED', '1') != '0' and
                    (int(chain.block_max_x) + 1) * (int(chain.block_max_y) + 1) <= 1296)
        if windowed:
            d_energy = torch.empty((n, H, W), dtype=torch.float64, device=dev)
            d_state = torch.empty((n, 4), dtype=torch.float64, device=dev)
            eng._check(lib.gsm_sgs_state_init(h, _ptr(cur), _ptr(d_trend), _ptr(d_energy), _ptr(d_state), eng._stream()))
        loss_cache = np.zeros((n, n_iter)); step_cache = np.zeros((n, n_iter)); blocks_cache = np.full((n, n_iter, 4), np.nan)
        loss_cache[:, 0] = loss_prev
        if keep_all:
            bed_cache = np.zeros((n, n_iter, H, W))
            bed_cache[:, 0] = bed_c
        if track:
            loc = np.asarray(chain.sample_loc)
            ij = np.zeros(loc.shape, dtype=np.int64)
            for k in range(loc.shape[0]):
                i_, j_ = np.where((chain.xx == loc[k, 0]) & (chain.yy == loc[k, 1]))
                ij[k] = [int(i_[0]), int(j_[0])]
            sample_values = np.zeros((n, ij.shape[0], n_iter))
            for c in range(n):
                sample_values[c, :, 0] = np.asarray(initial_beds[c])[ij[:, 0], ij[:, 1]]
        t0 = time.time()
        # Without a normal-score transformer and without per-iteration bed records nothing of an iteration has to come back
        # to the host before the next one: the draws do not depend on the chain state (chain_sgs.run consumes chain.rng in
        # the same order whatever is accepted), so a batch of iterations is drawn ahead, uploaded once, and simulated /
        # scored / decided (gsm_sgs_decide) / committed on the device back to back.
        philox = philox_seeds is not None
        if pcg64 and philox:
            raise ValueError("choose one of philox_seeds / pcg64")
        from .engine import GsmEngine
        if pcg64:
            d_gen = torch.as_tensor(GsmEngine.pack_pcg64_states(list(rngs)).view(np.int64)).to(dev)
        philox = philox or pcg64          # both draw on the device: same loop below
        # iterations per gsm_sgs_iterate call: a batch ends with a host round trip (device flag, record download) and restarts the pipeline
        # of records made ahead -- with few chains 128 instead of 32 iterations per batch is +7 % (4 chains: 66.6 -> 71.0 k chain-iterations/s);
        # with the chip full it changes nothing and the draw buffers grow with batch x chains
        batch = int(os.environ.get('GSM_SGS_BATCH', '128' if n <= 64 else '32')) if (host_nst is None and not keep_all and not track) else 1
        if philox:
            if not pcg64:
                if len(philox_seeds) != n:
                    raise ValueError('need one Philox seed per chain')
                d_seeds = torch.as_tensor(np.asarray([int(x) & 0xFFFFFFFFFFFFFFFF for x in philox_seeds], dtype=np.uint64).view(np.int64)).to(dev)
            d_region = torch.as_tensor(np.ascontiguousarray(chain.region_mask == 1, dtype=np.uint8)).to(dev) if chain.update_in_region else None
            d_isdata = torch.as_tensor(np.ascontiguousarray(cond_is_data, dtype=np.uint8)).to(dev)
        from ._lib import SgsBatch

        def make_batch(d_win, d_off, off_stride, d_cnt, d_cells, d_z, d_us, d_lrec, d_arec, cell_base=None, use_graph=False):
            """gsm_sgs_batch (include/gsm.h) of one batch of iterations: the loop body of chain_sgs.run (MCMC.py:1741-1822) is issued
            by ONE gsm_sgs_iterate call -- and, with static buffers and use_graph, replayed as one hipGraph launch."""
            b = SgsBatch()
            pv = lambda t: t.data_ptr() if t is not None else None
            b.cur, b.next, b.proposed = pv(cur), pv(nxt), pv(prop) if dev_qt else None
            b.zcond, b.trend = pv(d_zcond), pv(d_trend)
            if dev_qt:
                b.qt_quantiles, b.qt_references, b.qt_n = pv(d_q), pv(d_ref), nq
            if windowed:
                b.energy, b.state, b.windowed = pv(d_energy), pv(d_state), 1
            b.x_axis, b.y_axis, b.lag_cov = pv(d_xs), pv(d_ys), pv(d_lag)
            b.windows, b.cell_off, b.cell_cnt, b.cells, b.z, b.u = pv(d_win), pv(d_off), pv(d_cnt), pv(d_cells), pv(d_z), pv(d_us)
            b.cell_off_stride = off_stride
            if cell_base is not None:
                b.cell_base = cell_base.ctypes.data
            b.resampled, b.loss, b.bad, b.loss_prev, b.accept = pv(resampled), pv(d_loss), pv(d_bad), pv(d_lprev), pv(d_acc)
            b.loss_rec, b.acc_rec = pv(d_lrec), pv(d_arec)
            b.radius, b.sill = rad, float(vario["sill"])
            b.lag_mi, b.lag_mj, b.hw, b.num_points, b.max_cells, b.use_graph = lag_mi, lag_mj, hw, npts, max_cells, int(use_graph)
            b.grid_finite = int(grid_finite)
            return b

        # no NaN in the beds (and none can appear: every cell of a block is simulated): gsm_sgs_iterate may then make the records of
        # iteration j + 1 while iteration j is still running (include/gsm.h: grid_finite).  GSM_SGS_OVERLAP=0 turns that off.
        grid_finite = os.environ.get('GSM_SGS_OVERLAP', '1') != '0' and bool(torch.isfinite(cur).all())
        it_done = 0
        if philox and n_iter > 0:
            # device draws refill the SAME buffers batch after batch: the launch sequence of a full batch CAN be a hipGraph
            # (captured on a side stream -- the legacy default stream cannot be captured)
            # GSM_SGS_GRAPH=1.  Off by default: measured, the replay of a captured batch is no faster than its launches (the queue never runs
            # dry), and capturing + instantiating the 256 nodes of a batch costs about 35 ms -- a third of a 100-iteration run of 256 chains
            use_graph = batch > 1 and os.environ.get('GSM_SGS_GRAPH', '0') != '0'
            i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
            kmax = min(batch, n_iter)
            # two sets of draw buffers: the draws of batch b + 1 (they depend on the generators only, never on the chains' state) are
            # made on their own stream while batch b iterates.  A captured batch needs ONE set of static buffers: no second set then.
            # Only for few chains (4 chains, pcg64 mode: 38.7 -> 56.2 k chain-iterations/s): with the chip full (256 chains) the draw
            # kernel fits into the gap where the host downloads a batch's records, and drawing ahead measured 10 % slower (same box)
            ahead = os.environ.get('GSM_SGS_DRAW_AHEAD', '1' if n <= 64 else '0') != '0'
            n_sets = 2 if (ahead and not use_graph) else 1
            sets = []
            for _ in range(n_sets):
                sets.append(dict(win=i32(kmax * n * 4), blk=i32(kmax * n * 4), off=i32(kmax * n), cnt=i32(kmax * n),
                                 cells=i32(kmax * n * max_cells, 2), z=torch.empty(kmax * n * max_cells, dtype=torch.float64, device=dev),
                                 us=torch.empty(kmax * n, dtype=torch.float64, device=dev)))
            b_lrec = torch.empty(n * kmax, dtype=torch.float64, device=dev); b_arec = torch.empty(n * kmax, dtype=torch.uint8, device=dev)
            d_lprev = f64(loss_prev); d_acc = torch.empty(n, dtype=torch.uint8, device=dev)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            draw_st = torch.cuda.Stream(dev) if n_sets == 2 else side
            draw_st.wait_stream(torch.cuda.current_stream(dev))

            def views(bs, kb):
                return (bs['win'][:kb * n * 4].view(kb, n, 4), bs['blk'][:kb * n * 4].view(kb, n, 4), bs['off'][:kb * n].view(kb, n),
                        bs['cnt'][:kb * n].view(kb, n), bs['us'][:kb * n].view(kb, n))

            def draw(bs, it0, kb):
                """the draws of iterations it0 .. it0 + kb - 1 into the buffer set bs, on the draw stream"""
                d_win, d_blk, d_off, d_cnt, d_us = views(bs, kb)
                with torch.cuda.device(dev), torch.cuda.stream(draw_st):
                    args = (kb, _ptr(d_region), _ptr(d_isdata), int(chain.block_min_x), int(chain.block_max_x), int(chain.block_min_y),
                            int(chain.block_max_y), max_cells, _ptr(d_win), _ptr(d_blk), _ptr(d_off), _ptr(d_cnt), _ptr(bs['cells']), _ptr(bs['z']),
                            _ptr(d_us), draw_st.cuda_stream)
                    if pcg64:
                        eng._check(lib.gsm_sgs_draw_pcg64(h, _ptr(d_gen), *args))
                    else:
                        eng._check(lib.gsm_sgs_draw_philox(h, _ptr(d_seeds), int(philox_iter0) + it0, *args))
            if host_nst is None:
                draw(sets[0], 0, min(batch, n_iter))
        # (a transformer of another kind than scikit-learn's is called on the host once per iteration: the loop further down)
        n_batch = 0
        while philox and host_nst is None and it_done < n_iter:
            kb = min(batch, n_iter - it_done)
            bs = sets[n_batch % n_sets]
            d_win, d_blk, d_off, d_cnt, d_us = views(bs, kb)
            d_lrec, d_arec = b_lrec[:n * kb].view(n, kb), b_arec[:n * kb].view(n, kb)
            side.wait_stream(draw_st)                                  # this batch's draws
            nxt_kb = min(batch, n_iter - it_done - kb)
            if nxt_kb > 0 and n_sets == 2:
                draw(sets[(n_batch + 1) % 2], it_done + kb, nxt_kb)   # the other set: the batch that used it is over (its records were downloaded)
            with torch.cuda.device(dev), torch.cuda.stream(side):
                bt = make_batch(d_win, d_off, n, d_cnt, bs['cells'], bs['z'], d_us, d_lrec, d_arec, use_graph=use_graph)
                eng._check(lib.gsm_sgs_iterate(h, C.byref(bt), kb, side.cuda_stream))
                eng._check(lib.gsm_sgs_check(h, side.cuda_stream))
                lrec_h, arec_h, blk_h = d_lrec.cpu().numpy(), d_arec.cpu().numpy(), d_blk.cpu().numpy()
            if nxt_kb > 0 and n_sets == 1:
                draw(sets[0], it_done + kb, nxt_kb)
            n_batch += 1
            loss_cache[:, it_done:it_done + kb] = lrec_h
            step_cache[:, it_done:it_done + kb] = arec_h
            blocks_cache[:, it_done:it_done + kb] = blk_h.transpose(1, 0, 2)
            if keep_all or track:                      # per-iteration bed records (chain_sgs.run, MCMC.py:1814-1822): kb == 1 here
                bed_c = cur.cpu().numpy()
                if keep_all:
                    bed_cache[:, it_done] = bed_c + trend if detrend else bed_c
                if track:
                    for c in range(n):
                        sample_values[c, :, it_done] = bed_c[c][ij[:, 0], ij[:, 1]]
            it_done += kb
            if it_done >= n_iter:
                torch.cuda.current_stream(dev).wait_stream(side)
                torch.cuda.current_stream(dev).wait_stream(draw_st)
                global LAST_GRAPH_REPLAYS
                LAST_GRAPH_REPLAYS = int(lib.gsm_sgs_graph_replays(h))
            if progress_bar is not None:
                el = time.time() - t0
                print(f"Chain {getattr(chain, 'chain_id', 0)} ({str(getattr(chain, 'seed', 'Unknown'))[:6]}): "
                      f"{100 * (it_done - 1) / max(n_iter - 1, 1):3.0f}% | it/s: {it_done / max(el, 1e-9):7.2f} | n: {n_iter} | "
                      f"loss: {loss_cache[0, it_done - 1]:.3e} | acc: {step_cache[0, :it_done].sum() / it_done:.4f}", file=sys.stdout, flush=True)
        # Replay batches: the host draws of batch b + 1 are made while the device works on batch b (its launches are
        # asynchronous; gsm_sgs_check and the record download of batch b come after the draws of b + 1).
        def host_draws(it0, kb):
            wins = np.empty((kb, n, 4), np.int32); offs = np.zeros((kb, n + 1), np.int32); us = np.empty((kb, n))
            cells, zs, bases = [], [], np.zeros(kb + 1, np.int64)
            for c in range(n):                     # per chain in iteration order: each chain owns its generator
                for j in range(kb):
                    blk, win, inds, z, us[j, c] = chain._draw_iteration(rngs[c], cond_is_data)
                    blocks_cache[c, it0 + j] = blk
                    wins[j, c] = win
                    cells.append((j, c, inds)); zs.append((j, c, z))
            cells.sort(key=lambda t: (t[0], t[1])); zs.sort(key=lambda t: (t[0], t[1]))
            k = 0
            for j in range(kb):
                for c in range(n):
                    offs[j, c + 1] = offs[j, c] + cells[k][2].shape[0]; k += 1
                bases[j + 1] = bases[j] + offs[j, n]
            tot = int(bases[kb])
            cells_all = np.ascontiguousarray(np.concatenate([t[2] for t in cells]) if tot else np.zeros((1, 2), np.int32))
            z_all = np.concatenate([t[2] for t in zs]) if tot else np.zeros(1)
            return dict(it0=it0, kb=kb, wins=wins, offs=offs, us=us, bases=bases, cells=cells_all, z=z_all)

        def launch(d):
            kb, bases = d['kb'], d['bases']
            d_win = torch.as_tensor(d['wins']).to(dev); d_off = torch.as_tensor(d['offs']).to(dev); d_us = torch.as_tensor(d['us']).to(dev)
            d_cells = torch.as_tensor(d['cells']).to(dev); d_z = torch.as_tensor(d['z']).to(dev)
            d_lrec = torch.empty((n, kb), dtype=torch.float64, device=dev); d_arec = torch.empty((n, kb), dtype=torch.uint8, device=dev)
            with torch.cuda.device(dev):
                bt = make_batch(d_win, d_off, n + 1, None, d_cells, d_z, d_us, d_lrec, d_arec, cell_base=np.ascontiguousarray(bases, dtype=np.int64))
                eng._check(lib.gsm_sgs_iterate(h, C.byref(bt), kb, eng._stream()))
            d['keep'] = (d_win, d_off, d_us, d_cells, d_z)      # alive until the batch is finished
            d['lrec'], d['arec'] = d_lrec, d_arec

        def finish(d):
            with torch.cuda.device(dev):
                eng._check(lib.gsm_sgs_check(h, eng._stream()))
            it0, kb = d['it0'], d['kb']
            loss_cache[:, it0:it0 + kb] = d['lrec'].cpu().numpy()
            step_cache[:, it0:it0 + kb] = d['arec'].cpu().numpy()
            if progress_bar is not None:
                done = it0 + kb
                el = time.time() - t0
                print(f"Chain {getattr(chain, 'chain_id', 0)} ({str(getattr(chain, 'seed', 'Unknown'))[:6]}): "
                      f"{100 * (done - 1) / max(n_iter - 1, 1):3.0f}% | it/s: {done / max(el, 1e-9):7.2f} | n: {n_iter} | "
                      f"loss: {loss_cache[0, done - 1]:.3e} | acc: {step_cache[0, :done].sum() / done:.4f}", file=sys.stdout, flush=True)

        if batch > 1 and it_done < n_iter:
            d_lprev = f64(loss_prev); d_acc = torch.empty(n, dtype=torch.uint8, device=dev)
            running = host_draws(0, min(batch, n_iter))
            launch(running)
            it_done = running['kb']
            while it_done < n_iter:
                nxt_draws = host_draws(it_done, min(batch, n_iter - it_done))      # overlaps the device work of `running`
                finish(running)
                launch(nxt_draws)
                running = nxt_draws
                it_done += running['kb']
            finish(running)
        for it in range(it_done, n_iter):
            if philox:
                # device draws of ONE iteration (a host-side transformer sits between the draws and the simulation)
                bs = sets[0]
                d_win, d_blk1 = bs['win'][:n * 4].view(n, 4), bs['blk'][:n * 4].view(n, 4)
                d_off, d_cnt, d_us1, d_cells, d_z = bs['off'][:n], bs['cnt'][:n], bs['us'][:n], bs['cells'], bs['z']
                with torch.cuda.device(dev):
                    if pcg64:
                        eng._check(lib.gsm_sgs_draw_pcg64(h, _ptr(d_gen), 1, _ptr(d_region), _ptr(d_isdata),
                                                          int(chain.block_min_x), int(chain.block_max_x), int(chain.block_min_y), int(chain.block_max_y),
                                                          max_cells, _ptr(d_win), _ptr(d_blk1), _ptr(d_off), _ptr(d_cnt), _ptr(d_cells), _ptr(d_z),
                                                          _ptr(d_us1), eng._stream()))
                    else:
                        eng._check(lib.gsm_sgs_draw_philox(h, _ptr(d_seeds), int(philox_iter0) + it, 1, _ptr(d_region), _ptr(d_isdata),
                                                           int(chain.block_min_x), int(chain.block_max_x), int(chain.block_min_y), int(chain.block_max_y),
                                                           max_cells, _ptr(d_win), _ptr(d_blk1), _ptr(d_off), _ptr(d_cnt), _ptr(d_cells), _ptr(d_z),
                                                           _ptr(d_us1), eng._stream()))
                wins, us = d_win.cpu().numpy(), d_us1.cpu().numpy()
                blocks_cache[:, it] = d_blk1.cpu().numpy()
            else:
                wins = np.empty((n, 4), np.int32); offs = np.zeros(n + 1, np.int32); us = np.empty(n)
                cells, zs = [], []
                for c in range(n):
                    blk, win, inds, z, us[c] = chain._draw_iteration(rngs[c], cond_is_data)
                    blocks_cache[c, it] = blk
                    wins[c] = win
                    cells.append(inds); zs.append(z)
                    offs[c + 1] = offs[c] + inds.shape[0]
                d_win = torch.as_tensor(wins).to(dev)
                d_off = torch.as_tensor(offs).to(dev)
                d_cnt = None
                d_cells = torch.as_tensor(np.ascontiguousarray(np.concatenate(cells) if offs[-1] else np.zeros((1, 2), np.int32))).to(dev)
                d_z = torch.as_tensor(np.concatenate(zs) if offs[-1] else np.zeros(1)).to(dev)
            if dev_qt:
                qt(cur, nxt, 0)
            elif host_nst is not None:
                # the caller's transformer on the whole map, where the reference calls it (MCMC.py:1766)
                nxt.copy_(f64(np.stack([nst.transform(bed_c[c].reshape(-1, 1)).reshape(H, W) for c in range(n)])))
            with torch.cuda.device(dev):
                eng._check(lib.gsm_sgs_blocks_batch(h, _ptr(nxt), _ptr(d_zcond), _ptr(d_win), _ptr(d_xs), _ptr(d_ys), _ptr(d_lag), lag_mi, lag_mj, hw,
                                                    rad, npts, float(vario["sill"]), _ptr(d_off), _ptr(d_cnt), _ptr(d_cells), _ptr(d_z), max_cells,
                                                    eng._stream()))
                eng._check(lib.gsm_sgs_check(h, eng._stream()))
            if dev_qt:
                qt(nxt, prop, 1)
                loss_next, bad = loss_of(prop)
            elif host_nst is not None:
                newsim = nxt.cpu().numpy()
                bed_next = np.stack([nst.inverse_transform(newsim[c].reshape(-1, 1)).reshape(H, W) for c in range(n)])   # MCMC.py:1777
                d_next = f64(bed_next)
                loss_next, bad = loss_of(d_next)
            else:
                loss_next, bad = loss_of(nxt)
            loss_next = np.where(bad > 0, np.inf, loss_next)
            with np.errstate(over='ignore', invalid='ignore'):
                p_acc = np.where(loss_prev > loss_next, 1.0, np.minimum(1.0, np.exp(loss_prev - loss_next)))
            acc = us <= p_acc
            d_acc = torch.as_tensor(acc.astype(np.uint8)).to(dev)
            if dev_qt:
                with torch.cuda.device(dev):
                    eng._check(lib.gsm_sgs_commit_map(h, _ptr(cur), _ptr(prop), _ptr(resampled), _ptr(d_win), _ptr(d_acc), eng._stream()))
            elif host_nst is not None:
                for c in np.flatnonzero(acc):
                    bed_c[c] = bed_next[c]
                    r0, r1, c0, c1 = wins[c]
                    resampled[c, r0:r1, c0:c1] += 1
            else:
                with torch.cuda.device(dev):
                    eng._check(lib.gsm_sgs_commit(h, _ptr(cur), _ptr(nxt), _ptr(resampled), _ptr(d_win), _ptr(d_acc), eng._stream()))
            loss_prev = np.where(acc, loss_next, loss_prev)
            loss_cache[:, it] = loss_prev
            step_cache[:, it] = acc
            if keep_all or track:
                if host_nst is None:
                    bed_c = cur.cpu().numpy()
                if keep_all:
                    bed_cache[:, it] = bed_c + trend if detrend else bed_c
                if track:
                    for c in range(n):
                        sample_values[c, :, it] = bed_c[c][ij[:, 0], ij[:, 1]]
            if progress_bar is not None and (it % max(int(info_per_iter), 1) == 0 or it == n_iter - 1):
                el = time.time() - t0
                print(f"Chain {getattr(chain, 'chain_id', 0)} ({str(getattr(chain, 'seed', 'Unknown'))[:6]}): "
                      f"{100 * it / max(n_iter - 1, 1):3.0f}% | it/s: {(it + 1) / max(el, 1e-9):7.2f} | n: {n_iter} | "
                      f"loss: {loss_cache[0, it]:.3e} | acc: {step_cache[0, :it + 1].sum() / (it + 1):.4f}", file=sys.stdout, flush=True)
        if host_nst is None:
            bed_c = cur.cpu().numpy()
        res = resampled.cpu().numpy().astype(np.float64)
        if pcg64:                                  # the generators continue where the device left them, as after NumPy calls
            for g, st in zip(rngs, GsmEngine.unpack_pcg64_states(d_gen.cpu().numpy().view(np.uint64))):
                g.bit_generator.state = st
    finally:
        eng.close()
    out = []
    for c in range(n):
        last = bed_c[c] + trend if detrend else bed_c[c]
        tup = (bed_cache[c] if keep_all else last, loss_cache[c].copy(), np.zeros(n_iter), loss_cache[c], step_cache[c], res[c],
               blocks_cache[c])
        out.append(tup + (sample_values[c],) if track else tup)
    return out, rngs


def init_msc_chain_by_instance(param_dict):
    """Rebuild a small-scale chain from a copy of another one's __dict__ (+ 'rng_seed', 'initial_bed') (MCMC.py:402-431)."""
    p = param_dict
    ch = chain_sgs_gpu(p['xx'], p['yy'], p['initial_bed'], p['surf'], p['velx'], p['vely'], p['dhdt'], p['smb'], p['cond_bed'],
                       p['data_mask'], p['grounded_ice_mask'], p['resolution'])
    ch.update_in_region = p['update_in_region']
    ch.region_mask = p['region_mask']
    ch.sigma_mc = p['sigma_mc']
    ch.mc_region_mask = p['mc_region_mask']
    ch.block_min_x, ch.block_min_y = p['block_min_x'], p['block_min_y']
    ch.block_max_x, ch.block_max_y = p['block_max_x'], p['block_max_y']
    ch.do_transform = p['do_transform']
    ch.nst_trans = deepcopy(p['nst_trans'])
    ch.trend = p['trend']
    ch.detrend_map = p['detrend_map']
    ch.vario_type = p['vario_type']
    ch.vario_param = deepcopy(p['vario_param'])
    ch.sgs_param = deepcopy(p['sgs_param'])
    ch.rng = np.random.default_rng(seed=p['rng_seed'])
    ch.rng_seed = p['rng_seed']
    ch.sample_loc = deepcopy(p['sample_loc'])
    return ch
