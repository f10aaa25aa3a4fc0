"""TEST INFRASTRUCTURE ONLY -- CPU oracle for the large-scale-chain Metropolis step.

Clean-room NumPy restatement of the reference hot path (all citations are relative to
/root/reference/):

    gstatsMCMC/MCMC.py:176-254    spectral_synthesis_field     -> spectral_field()
    gstatsMCMC/MCMC.py:568-623    get_block_sizes/get_edge_masks -> block_pairs(), edge_masks()
    gstatsMCMC/MCMC.py:689-714    get_crf_weight               -> crf_weight()
    gstatsMCMC/MCMC.py:742-778    RandField.get_rfblock        -> OracleRandField.get_rfblock()
    gstatsMCMC/Topography.py:592-600 get_mass_conservation_residual -> mc_residual()
    gstatsMCMC/MCMC.py:1021-1044  chain.loss                   -> gaussian_loss()
    gstatsMCMC/MCMC.py:1137-1443  chain_crf.run                -> run_chain()
    gstatsMCMC/gstatsim_custom/covariance.py:4-28, _krige.py:83-122 -> cov_matrix()

Pinning: ``oracle/make_fixtures.py`` runs the imported reference and this file on the same
seeded inputs in the build container and asserts bit-identity of every output before it
writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` re-checks this file against
those vectors on every run (CPU suite).  The reference has no tests or golden vectors of
its own, so the committed vectors are the pin.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product package (``mcmc_gpu_amd``) never does.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field as dc_field

import numpy as np

# --------------------------------------------------------------------------------------
# synthetic inputs -- deterministic
# --------------------------------------------------------------------------------------


def synthetic_problem(H: int, W: int | None = None, res: float = 500.0) -> dict:
    """Deterministic synthetic glacier grid of SURVEY.md section 8d."""
    W = H if W is None else W
    x = np.arange(W) * res
    y = np.arange(H) * res
    xx, yy = np.meshgrid(x, y)
    Lx, Ly = W * res, H * res
    rng = np.random.default_rng(1234)
    surf = 2000.0 + 200.0 * np.sin(2 * np.pi * xx / Lx) * np.cos(2 * np.pi * yy / Ly)
    thick = 1000.0 + 150.0 * np.cos(4 * np.pi * xx / Lx) + 100.0 * np.sin(2 * np.pi * yy / Ly)
    bed = surf - thick + rng.normal(0, 5, (H, W))
    velx = 200.0 + 50.0 * np.sin(2 * np.pi * yy / Ly)
    vely = 50.0 * np.cos(2 * np.pi * xx / Lx)
    dhdt = rng.normal(0, 0.1, (H, W))
    smb = np.full((H, W), 0.2)
    data_mask = np.zeros((H, W), dtype=bool)
    data_mask[::16, :] = True
    cond_bed = np.where(data_mask, bed, np.nan)
    grounded = np.ones((H, W), dtype=bool)
    region = np.zeros((H, W), dtype=int)
    region[H // 8: 7 * H // 8, W // 8: 7 * W // 8] = 1
    return dict(xx=xx, yy=yy, bed=bed, surf=surf, velx=velx, vely=vely, dhdt=dhdt, smb=smb,
                cond_bed=cond_bed, data_mask=data_mask, grounded_ice_mask=grounded,
                region_mask=region, resolution=res)


def chain_initial_bed(problem: dict, i: int) -> np.ndarray:
    """Initial bed of chain i: chain 0 = bed, others perturbed."""
    if i == 0:
        return problem["bed"].copy()
    H, W = problem["bed"].shape
    return problem["bed"] + np.random.default_rng(10_000 + i).normal(0, 5, (H, W))


# --------------------------------------------------------------------------------------
# setup-time helpers
# --------------------------------------------------------------------------------------


def block_pairs(min_x, max_x, min_y, max_y, steps=5) -> np.ndarray:
    """(2, steps*steps) int array; row 0 = widths, row 1 = heights, all even (MCMC.py:568-581)."""
    w = np.linspace(min_x, max_x, steps, dtype=int)
    h = np.linspace(min_y, max_y, steps, dtype=int)
    ww, hh = np.meshgrid(w, h)
    return np.array([(ww // 2 * 2).ravel(), (hh // 2 * 2).ravel()])


def _logistic(d_rescaled, lp):
    L, x0, k, off = lp
    return L / (1 + np.exp(-k * (d_rescaled - x0))) - off


def edge_masks(pairs, logistic_param, max_dist, res) -> list:
    """Logistic taper of the distance to the block border, one (bh, bw) array per size
    (MCMC.py:583-623).  The KD-tree distance of the reference to the nearest border cell is
    res*min(i, bh-1-i, j, bw-1-j) exactly (one coordinate difference is always zero)."""
    out = []
    for n in range(pairs.shape[1]):
        bw, bh = int(pairs[0, n]), int(pairs[1, n])
        jj, ii = np.meshgrid(np.arange(bw), np.arange(bh))
        d_i = np.minimum(ii, bh - 1 - ii) * res
        d_j = np.minimum(jj, bw - 1 - jj) * res
        dist = np.sqrt(np.minimum(d_i, d_j) ** 2)
        resc = np.where(dist > max_dist, 1, dist / max_dist)
        out.append(_logistic(resc, logistic_param))
    return out


def nearest_dist(xx, yy, mask) -> np.ndarray:
    """Euclidean distance of every cell to the nearest True cell of ``mask``
    (Utilities.py:21-24); brute force in row chunks, same sqrt(dx^2+dy^2) arithmetic."""
    px, py = xx[mask], yy[mask]
    fx, fy = xx.ravel(), yy.ravel()
    out = np.empty(fx.shape)
    step = max(1, int(4_000_000 // max(1, px.size)))
    for s in range(0, fx.size, step):
        dx = fx[s:s + step, None] - px[None, :]
        dy = fy[s:s + step, None] - py[None, :]
        out[s:s + step] = np.sqrt((dx * dx + dy * dy).min(axis=1))
    return out.reshape(xx.shape)


def crf_weight(xx, yy, data_mask, logistic_param, max_dist) -> np.ndarray:
    """Data-conditioning weight, 0 on data cells (MCMC.py:689-714)."""
    dist = nearest_dist(xx, yy, data_mask == 1)
    resc = np.where(dist > max_dist, 1, dist / max_dist)
    logi = _logistic(resc, logistic_param)
    return logi - np.min(logi)


# --------------------------------------------------------------------------------------
# proposal: spectral synthesis (MCMC.py:176-254) and the RandField wrapper (MCMC.py:742-778)
# --------------------------------------------------------------------------------------


@dataclass
class RFParams:
    range_min_x: float
    range_max_x: float
    range_min_y: float
    range_max_y: float
    scale_min: float
    scale_max: float
    nugget_max: float
    model_name: str
    isotropic: bool
    smoothness: float | None = None


def spectral_amplitude(shape, res, model_name, range_x, range_y, smoothness=None) -> np.ndarray:
    """sqrt(S(k)) on the fftfreq grid (MCMC.py:209-239, :244)."""
    ny, nx = shape
    if model_name == "Gaussian":
        lx, ly = range_x / np.sqrt(3), range_y / np.sqrt(3)
    elif model_name == "Exponential":
        lx, ly = range_x / 3.0, range_y / 3.0
    else:
        lx, ly = range_x / 2.0, range_y / 2.0
    kx = np.fft.fftfreq(nx, d=res) * 2 * np.pi
    ky = np.fft.fftfreq(ny, d=res) * 2 * np.pi
    kyv, kxv = np.meshgrid(ky, kx, indexing="ij")
    k = np.sqrt(kxv ** 2 + kyv ** 2) + 1e-10
    a = np.sqrt(lx * ly)
    if model_name == "Gaussian":
        S = np.exp(-0.5 * (a * k) ** 2)
    elif model_name == "Exponential":
        S = 1.0 / (1.0 + (a * k) ** 2) ** 1.5
    else:
        nu = smoothness or 1.0
        const = (4 * np.pi * math.gamma(nu + 1) * (2 * nu) ** nu) / (math.gamma(nu) * a ** (2 * nu))
        kappa = 2 * nu / (a ** 2)
        S = const * ((kappa + 4 * np.pi * k ** 2) ** (-nu - 1))
    return np.sqrt(S)


def spectral_draws(rng: np.random.Generator, p: RFParams, shape) -> dict:
    """The random numbers of one spectral_synthesis_field call, consumed in the reference's order (MCMC.py:200-207, :242,
    :251): scale, nugget, range(s), two normal planes, the nugget plane."""
    ny, nx = shape
    scale = rng.uniform(p.scale_min, p.scale_max) / 3.0
    nug = rng.uniform(0.0, p.nugget_max)
    if not p.isotropic:
        range_x = rng.uniform(p.range_min_x, p.range_max_x)
        range_y = rng.uniform(p.range_min_y, p.range_max_y)
    else:
        range_x = range_y = rng.uniform(p.range_min_x, p.range_max_x)
    n_re = rng.normal(size=(ny, nx))
    n_im = rng.normal(size=(ny, nx))
    n_nug = rng.normal(0, np.sqrt(nug), size=(ny, nx))
    return dict(scale=scale, nug=nug, range_x=range_x, range_y=range_y, n_re=n_re, n_im=n_im, n_nug=n_nug)


def spectral_from_draws(d: dict, p: RFParams, shape, res) -> np.ndarray:
    """The arithmetic of MCMC.py:209-251 on given draws."""
    amp = spectral_amplitude(shape, res, p.model_name, d["range_x"], d["range_y"], p.smoothness)
    noise = d["n_re"] + 1j * d["n_im"]
    fld = np.fft.ifft2(noise * amp).real
    fld = (fld - np.mean(fld)) / (np.std(fld) + 1e-12)
    return fld * d["scale"] + d["n_nug"]


def spectral_field(rng: np.random.Generator, p: RFParams, shape, res, trace: dict | None = None):
    """One spectral-synthesis realisation; draw order of MCMC.py:200-251."""
    d = spectral_draws(rng, p, shape)
    fld = spectral_from_draws(d, p, shape, res)
    if trace is not None:
        trace.update(scale=d["scale"], nug=d["nug"], range_x=d["range_x"], range_y=d["range_y"])
    return fld


class OracleRandField:
    """State needed by get_rfblock: params, size table, edge masks and its own Generator."""

    def __init__(self, params: RFParams, seed, pairs, masks, resolution):
        self.p = params
        self.rng = np.random.default_rng(seed=seed) if not isinstance(seed, np.random.Generator) else seed
        self.pairs = pairs
        self.edge_masks = masks
        self.resolution = resolution

    def get_rfblock(self, trace: dict | None = None):
        idx = self.rng.integers(low=0, high=self.pairs.shape[1], size=1)[0]
        bw, bh = int(self.pairs[0, idx]), int(self.pairs[1, idx])
        while True:
            f = spectral_field(self.rng, self.p, (bh, bw), self.resolution, trace)
            if np.isnan(f).sum() == 0:
                break
        if trace is not None:
            trace["size_idx"] = int(idx)
        return f * self.edge_masks[idx]


# --------------------------------------------------------------------------------------
# likelihood pieces
# --------------------------------------------------------------------------------------


def _grad_uniform(f, h, axis):
    """np.gradient(f, h, axis=axis) restated: central differences divided by 2.0*h in the
    interior, one-sided first-order differences divided by h on the two edges."""
    f = np.asarray(f, dtype=float)
    out = np.empty_like(f)
    n = f.shape[axis]
    if n < 2:
        raise ValueError("need at least 2 samples along the axis")

    def sl(a, b=None, step=None):
        s = [slice(None)] * f.ndim
        s[axis] = slice(a, b, step)
        return tuple(s)

    out[sl(1, -1)] = (f[sl(2, None)] - f[sl(None, -2)]) / (2.0 * h)
    out[sl(0, 1)] = (f[sl(1, 2)] - f[sl(0, 1)]) / h
    out[sl(-1, None)] = (f[sl(-1, None)] - f[sl(-2, -1)]) / h
    return out


def mc_residual(bed, surf, velx, vely, dhdt, smb, resolution):
    """d/dx(velx*H) + d/dy(vely*H) + dhdt - smb with H = surf - bed (Topography.py:592-600)."""
    thick = surf - bed
    dx = _grad_uniform(velx * thick, resolution, 1)
    dy = _grad_uniform(vely * thick, resolution, 0)
    return dx + dy + dhdt - smb


def gaussian_loss(mc_res, mc_region_mask, sigma_mc, state_f32=False):
    """(total, loss_mc, loss_data) with loss_data == 0 (MCMC.py:1021-1044)."""
    sq = np.square(mc_res[mc_region_mask == 1])
    if state_f32:
        sq = sq.astype(np.float32).astype(np.float64)
    loss_mc = np.nansum(sq) / (2 * sigma_mc ** 2)
    return loss_mc + 0, loss_mc, 0


# --------------------------------------------------------------------------------------
# the chain
# --------------------------------------------------------------------------------------


@dataclass
class ChainConfig:
    surf: np.ndarray
    velx: np.ndarray
    vely: np.ndarray
    dhdt: np.ndarray
    smb: np.ndarray
    region_mask: np.ndarray        # block centres + update mask when update_in_region
    grounded_ice_mask: np.ndarray  # update mask otherwise
    mc_region_mask: np.ndarray     # where the residual enters the loss
    crf_data_weight: np.ndarray | None
    resolution: float
    sigma_mc: float
    update_in_region: bool = True
    block_type: str = "CRF_weight"
    # build-defined mixed-precision mode (BASELINE configs[4]; not in the reference, whose torch class is all-fp32,
    # MCMC_gpu.py:261): bed and per-cell squared residuals are stored as float32, arithmetic stays float64
    state_f32: bool = False


@dataclass
class StepTrace:
    """Per-step record of every random draw, for replaying the identical chain on the GPU."""
    size_idx: list = dc_field(default_factory=list)
    centre: list = dc_field(default_factory=list)     # (row, col)
    u: list = dc_field(default_factory=list)
    fields: list = dc_field(default_factory=list)     # masked fields f (bh, bw)
    rf_scalars: list = dc_field(default_factory=list)  # (scale, nug, range_x, range_y)
    tries: list = dc_field(default_factory=list)


def window_bounds(row, col, bh, bw, H, W):
    """Clipped window and the matching sub-slice of f (MCMC.py:1266-1276)."""
    r0 = max(0, int(row - bh / 2))
    r1 = min(H, int(row + bh / 2))
    c0 = max(0, int(col - bw / 2))
    c1 = min(W, int(col + bw / 2))
    mr0 = max(bh - r1, 0)
    mr1 = min(H - r0, bh)
    mc0 = max(bw - c1, 0)
    mc1 = min(W - c0, bw)
    return r0, r1, c0, c1, mr0, mr1, mc0, mc1


def mh_step(cfg: ChainConfig, bed_c, mc_res, loss_prev, f, row, col, u):
    """One Metropolis step given all draws (MCMC.py:1263-1360).
    Returns (accepted, bed_c, mc_res, loss_now, (r0, r1, c0, c1))."""
    H, W = bed_c.shape
    bh, bw = f.shape
    r0, r1, c0, c1, mr0, mr1, mc0, mc1 = window_bounds(row, col, bh, bw, H, W)
    if cfg.block_type == "CRF_weight":
        pert = f[mr0:mr1, mc0:mc1] * cfg.crf_data_weight[r0:r1, c0:c1]
    else:
        pert = f[mr0:mr1, mc0:mc1]
    bed_next = bed_c.copy()
    bed_next[r0:r1, c0:c1] = bed_next[r0:r1, c0:c1] + pert
    upd_mask = cfg.region_mask if cfg.update_in_region else cfg.grounded_ice_mask
    bed_next = np.where(upd_mask, bed_next, bed_c)
    if cfg.state_f32:
        bed_next = bed_next.astype(np.float32).astype(np.float64)

    hr0, hr1 = max(0, r0 - 1), min(H, r1 + 1)
    hc0, hc1 = max(0, c0 - 1), min(W, c1 + 1)
    loc = mc_residual(bed_next[hr0:hr1, hc0:hc1], cfg.surf[hr0:hr1, hc0:hc1],
                      cfg.velx[hr0:hr1, hc0:hc1], cfg.vely[hr0:hr1, hc0:hc1],
                      cfg.dhdt[hr0:hr1, hc0:hc1], cfg.smb[hr0:hr1, hc0:hc1], cfg.resolution)
    cand = mc_res.copy()
    cand[r0:r1, c0:c1] = loc[r0 - hr0: r0 - hr0 + (r1 - r0), c0 - hc0: c0 - hc0 + (c1 - c0)]
    loss_next = gaussian_loss(cand, cfg.mc_region_mask, cfg.sigma_mc, cfg.state_f32)[0]

    thick = cfg.surf[r0:r1, c0:c1] - bed_next[r0:r1, c0:c1]
    if np.sum((thick <= 0)[upd_mask[r0:r1, c0:c1] == 1]) > 0:
        loss_next = np.inf
    if loss_prev > loss_next:
        p_acc = 1
    else:
        p_acc = min(1, np.exp(loss_prev - loss_next))
    if u <= p_acc:
        return True, bed_next, cand, loss_next, (r0, r1, c0, c1)
    return False, bed_c, mc_res, loss_prev, (r0, r1, c0, c1)


def run_chain(cfg: ChainConfig, initial_bed, n_iter, rf: OracleRandField, rng: np.random.Generator,
              record: bool = False, only_save_last_bed: bool = True):
    """chain_crf.run restated (MCMC.py:1137-1443).  Returns the 7-tuple (+ StepTrace when
    ``record``).  Entry 0 of every cache is the initial state; n_iter-1 proposals follow."""
    H, W = initial_bed.shape
    loss_mc_cache = np.zeros(n_iter)
    loss_data_cache = np.zeros(n_iter)
    loss_cache = np.zeros(n_iter)
    step_cache = np.zeros(n_iter)
    blocks_cache = np.full((n_iter, 4), np.nan)
    resampled = np.zeros((H, W))
    bed_cache = None if only_save_last_bed else np.zeros((n_iter, H, W))
    trace = StepTrace() if record else None

    bed_c = initial_bed.astype(np.float32).astype(np.float64) if cfg.state_f32 else initial_bed
    mc_res = mc_residual(bed_c, cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.resolution)
    loss_prev = gaussian_loss(mc_res, cfg.mc_region_mask, cfg.sigma_mc, cfg.state_f32)[0]
    loss_cache[0] = loss_mc_cache[0] = loss_prev
    if bed_cache is not None:
        bed_cache[0] = bed_c
    upd_mask = cfg.region_mask if cfg.update_in_region else cfg.grounded_ice_mask

    for i in range(1, n_iter):
        tr = {} if record else None
        f = rf.get_rfblock(tr)
        bh, bw = f.shape
        tries = 0
        if cfg.update_in_region:
            while True:
                row = rng.integers(low=0, high=H, size=1)[0]
                col = rng.integers(low=0, high=W, size=1)[0]
                tries += 1
                if cfg.region_mask[row, col] == 1:
                    break
        else:
            row = rng.integers(low=0, high=H, size=1)[0]
            col = rng.integers(low=0, high=W, size=1)[0]
            tries = 1
        blocks_cache[i, :] = [row, col, bh, bw]
        u = rng.random()
        acc, bed_c, mc_res, loss_prev, (r0, r1, c0, c1) = mh_step(cfg, bed_c, mc_res, loss_prev, f, row, col, u)
        loss_cache[i] = loss_mc_cache[i] = loss_prev
        step_cache[i] = acc
        if acc:
            resampled[r0:r1, c0:c1] += upd_mask[r0:r1, c0:c1]
        if bed_cache is not None:
            bed_cache[i] = bed_c
        if record:
            trace.size_idx.append(tr["size_idx"])
            trace.centre.append((int(row), int(col)))
            trace.u.append(float(u))
            trace.fields.append(f)
            trace.rf_scalars.append((tr["scale"], tr["nug"], tr["range_x"], tr["range_y"]))
            trace.tries.append(tries)
    out = (bed_c if only_save_last_bed else bed_cache, loss_mc_cache, loss_data_cache, loss_cache,
           step_cache, resampled, blocks_cache)
    return out + (trace,) if record else out


# --------------------------------------------------------------------------------------
# standard setups used by fixtures, tests and the CPU baseline
# --------------------------------------------------------------------------------------


def standard_rf_params(model="Matern", isotropic=True, nugget_max=0.0, smoothness=0.9125) -> RFParams:
    """Driver values (largeScaleChain_multiprocessing_GPU.py:567-579; nu from T3 cell 14)."""
    return RFParams(10e3, 50e3, 10e3, 50e3, 50, 150, nugget_max, model, isotropic,
                    smoothness if model == "Matern" else None)


def standard_setup(H, W=None, block_min=None, block_max=None, sigma_mc=5.0, block_type="CRF_weight",
                   update_in_region=True, rf_params: RFParams | None = None):
    """Problem + ChainConfig + (pairs, masks) of SURVEY.md section 8d."""
    prob = synthetic_problem(H, W)
    res = prob["resolution"]
    if block_min is None:
        block_min, block_max = (8, 16) if min(prob["bed"].shape) < 128 else (50, 80)
    pairs = block_pairs(block_min, block_max, block_min, block_max)
    lp = [2, 0, 6, 1]
    max_dist = 49900.0
    masks = edge_masks(pairs, lp, max_dist, res)
    w = crf_weight(prob["xx"], prob["yy"], prob["data_mask"], lp, max_dist)
    region = prob["region_mask"]
    cfg = ChainConfig(prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                      region if update_in_region else np.full(region.shape, 1), prob["grounded_ice_mask"],
                      region if update_in_region else np.full(region.shape, 1),
                      w, res, sigma_mc, update_in_region, block_type)
    return prob, cfg, pairs, masks, (rf_params or standard_rf_params())


def run_standard_chain(H, n_iter, chain_index=0, record=False, **kw):
    prob, cfg, pairs, masks, rfp = standard_setup(H, **kw)
    seed = 7 + chain_index
    rf = OracleRandField(rfp, seed, pairs, masks, prob["resolution"])
    rng = np.random.default_rng(seed=seed)
    return run_chain(cfg, chain_initial_bed(prob, chain_index), n_iter, rf, rng, record=record)


# --------------------------------------------------------------------------------------
# covariance assembly (gstatsim_custom/covariance.py:4-28, _krige.py:83-122)
# --------------------------------------------------------------------------------------


def rotation_matrix(azimuth, major_range, minor_range):
    th = (azimuth / 180.0) * np.pi
    rot = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    return np.dot(rot, np.array([[1 / major_range, 0], [0, 1 / minor_range]]))


def cov_norm(h, vtype, sill, nugget, s=None):
    """Covariance on normalised lag h (covariance.py:4-28); h is not mutated here."""
    vtype = vtype.lower()
    if vtype == "exponential":
        return (sill - nugget) * np.exp(-3 * h)
    if vtype == "gaussian":
        return (sill - nugget) * np.exp(-3 * np.square(h))
    if vtype == "spherical":
        c = sill - nugget - 1.5 * h + 0.5 * np.power(h, 3)
        c = np.where(h > 1, sill - 1, c)   # reference quirk, covariance.py:14
        return c
    if vtype == "matern":
        from scipy.special import kv, gamma
        sc = 0.45246434 * np.exp(-0.70449189 * s) + 1.7863836
        hh = np.where(h == 0.0, 1e-8, h)
        c = (sill - nugget) * 2 / gamma(s) * np.power(sc * hh * np.sqrt(s), s) * kv(s, 2 * sc * hh * np.sqrt(s))
        return np.where(np.isnan(c), sill - nugget, c)
    raise ValueError(vtype)


def cov_matrix(coord, vario):
    """Dense covariance of N points (N,2) (_krige.py:105-122)."""
    R = rotation_matrix(vario["azimuth"], vario["major_range"], vario["minor_range"])
    m = coord @ R
    d = m[:, None, :] - m[None, :, :]
    h = np.sqrt((d * d).sum(axis=2))
    return cov_norm(h, vario["vtype"], vario["sill"], vario["nugget"], vario.get("s"))
