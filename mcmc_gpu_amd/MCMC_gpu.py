"""Host-side mirror of the reference's large-scale-chain interface (gstatsMCMC/MCMC_gpu.py namespace).

Same class / function names, argument meaning, return tuples and exceptions as the reference, so that
`from mcmc_gpu_amd import MCMC_gpu` can stand where `from gstatsMCMC import MCMC_gpu` stands in
largeScaleChain_multiprocessing_GPU.py:

    RandField                      gstatsMCMC/MCMC.py:433-778
    chain_crf_gpu                  gstatsMCMC/MCMC_gpu.py:149-582 (and chain / chain_crf, MCMC.py:780-1443)
    init_lsc_chain_by_instance     gstatsMCMC/MCMC.py:359-377   (returns the GPU class: fixes SURVEY section 9-1)
    initiate_RF_by_instance        gstatsMCMC/MCMC.py:381-398

The Metropolis loop itself never runs here: `run` hands whole segments of steps to libgsm_hip.so.
Two draw modes:
  'replay'  (default) -- the draws come from the two NumPy Generators exactly as in the reference (RandField.rng
            for the proposal, chain.rng for centre and accept uniform, MCMC.py:742-778, :1253-1261, :1336), are
            uploaded in chunks, and the device performs the steps.  Results equal the CPU chain_crf.run on the
            same seeds (accept masks bit-exact, loss <= 1e-10 rel).  This mirrors the reference's own GPU class,
            which also draws on the host (MCMC_gpu.py:368-369, :468).
  'philox'  -- everything on the device with the counter-based generator (throughput mode, many chains).
"""
from __future__ import annotations

import math
import os
import sys
import time
from copy import deepcopy

import numpy as np

from . import Topography  # noqa: F401  (re-exported like the reference namespace does)

__all__ = ["RandField", "chain_crf_gpu", "init_lsc_chain_by_instance", "initiate_RF_by_instance",
           "spectral_synthesis_field", "run_many", "run_many_replay", "draw_chunk", "min_dist_from_mask"]


def __getattr__(name):
    """The small-scale chain lives in mcmc_gpu_amd.sgs; it is reachable from this namespace like in the reference's
    (gstatsMCMC.MCMC holds both chains), imported on first use."""
    if name in ("chain_sgs_gpu", "init_msc_chain_by_instance", "run_many_sgs"):
        from . import sgs
        return getattr(sgs, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


def min_dist_from_mask(xx, yy, mask, device=None):
    """Distance from every cell to the nearest True cell of `mask` (Utilities.py:21-24) -- SETUP-time code, run once per
    problem, not part of the sampler's hot path.
    device=True: the exact brute-force HIP kernel (gsm_min_dist_from_mask); raises without a GPU.
    device=False: the host KD-tree query, which is literally what the reference calls at this point (scipy, not the oracle).
    device=None (default): the HIP kernel when a GPU is visible, else the host KD-tree -- so that chain / RandField objects
    can be SET UP (and pickled, and handed to workers) on a machine without a GPU, as the reference's driver does before it
    starts its pool.  Both give the same values bit for bit (test_min_dist_device_equals_kdtree).  The sampler itself
    (chain_crf_gpu.run, run_many) has no host fallback."""
    if device is None:
        import torch
        device = torch.cuda.is_available()
    if device:
        from .engine import GsmEngine
        eng = GsmEngine(xx.shape[0], xx.shape[1], 1)      # raises RuntimeError without a GPU
        try:
            return eng.min_dist_from_mask(xx, yy, mask)
        finally:
            eng.close()
    from scipy.spatial import cKDTree
    pts = np.array([xx[mask], yy[mask]]).T
    if pts.shape[0] == 0:
        raise ValueError("mask selects no cell")
    return cKDTree(pts).query(np.array([xx.ravel(), yy.ravel()]).T)[0].reshape(xx.shape)


_KGRID = {}


def _wavenumber_grid(ny, nx, res):
    """(k, 4 pi k^2) of MCMC.py:221-224, :239 for a block shape -- the same operations in the same order as the reference,
    computed once per shape instead of once per proposal (they depend on nothing that is drawn)."""
    key = (ny, nx, float(res))
    g = _KGRID.get(key)
    if g is None:
        kyv, kxv = np.meshgrid(np.fft.fftfreq(ny, d=res) * 2 * np.pi, np.fft.fftfreq(nx, d=res) * 2 * np.pi, indexing="ij")
        k = np.sqrt(kxv ** 2 + kyv ** 2) + 1e-10
        g = _KGRID[key] = (k, 4 * np.pi * k ** 2)
    return g


def spectral_synthesis_field(RF, shape, res=1.0):
    """FFT spectral-synthesis realisation with the reference's draw order (MCMC.py:176-254): scale, nugget,
    range(s), two planes of normals, nugget normals.  Host generator of 'replay' mode."""
    ny, nx = shape
    rng = RF.rng
    scale = rng.uniform(RF.scale_min, RF.scale_max) / 3.0
    nug = rng.uniform(0.0, RF.nugget_max)
    if RF.isotropic:
        rx = ry = rng.uniform(RF.range_min_x, RF.range_max_x)
    else:
        rx = rng.uniform(RF.range_min_x, RF.range_max_x)
        ry = rng.uniform(RF.range_min_y, RF.range_max_y)
    div = {"Gaussian": np.sqrt(3), "Exponential": 3.0}.get(RF.model_name, 2.0)
    a = np.sqrt((rx / div) * (ry / div))
    k, k2 = _wavenumber_grid(ny, nx, res)
    if RF.model_name == "Gaussian":
        S = np.exp(-0.5 * (a * k) ** 2)
    elif RF.model_name == "Exponential":
        S = 1.0 / (1.0 + (a * k) ** 2) ** 1.5
    else:
        nu = RF.smoothness or 1.0
        S = ((4 * np.pi * math.gamma(nu + 1) * (2 * nu) ** nu) / (math.gamma(nu) * a ** (2 * nu))) * \
            ((2 * nu / (a ** 2) + k2) ** (-nu - 1))
    white = rng.normal(size=(ny, nx)) + 1j * rng.normal(size=(ny, nx))
    fld = np.fft.ifft2(white * np.sqrt(S)).real
    fld = (fld - np.mean(fld)) / (np.std(fld) + 1e-12)
    return fld * scale + rng.normal(0, np.sqrt(nug), size=(ny, nx))


class RandField:
    """Proposal description: covariance-model ranges, block-size table, edge taper (MCMC.py:433-778)."""

    def __init__(self, range_min_x, range_max_x, range_min_y, range_max_y, scale_min, scale_max, nugget_max,
                 model_name, isotropic, smoothness=None, rng_seed=None):
        if rng_seed is None:
            rng = np.random.default_rng()
        elif isinstance(rng_seed, (int, np.integer)):
            rng = np.random.default_rng(seed=int(rng_seed))
        elif isinstance(rng_seed, np.random.Generator):
            rng = rng_seed
        else:
            raise ValueError('Seed should be an integer, a NumPy random Generator, or None')
        self.rng = rng
        if (range_max_x < range_min_x) or (range_max_y < range_min_y):
            print('the maximum range must be greater to equal to the minimum range')
        self.range_max_x, self.range_max_y = range_max_x, range_max_y
        self.range_min_x, self.range_min_y = range_min_x, range_min_y
        self.scale_min, self.scale_max = scale_min, scale_max
        self.nugget_max = nugget_max
        if model_name not in ('Gaussian', 'Exponential', 'Matern'):
            raise Exception('please put in a valid model_name, including Gaussian, Exponential, and Matern')
        if model_name == 'Matern' and smoothness is None:
            raise Exception('a smoothness value must be defined if model name is Matern')
        self.smoothness = smoothness
        self.model_name = model_name
        self.isotropic = isotropic

    def set_generation_method(self, spectral):
        self.spectral = spectral

    def set_block_sizes(self, min_block_x, max_block_x, min_block_y, max_block_y, steps=5):
        self.min_block_x, self.min_block_y = min_block_x, min_block_y
        self.max_block_x, self.max_block_y = max_block_x, max_block_y
        self.steps = steps
        self.pairs = self.get_block_sizes()

    def set_weight_param(self, logis_func_L, logis_func_x0, logis_func_k, logis_func_offset, max_dist, resolution):
        if not hasattr(self, 'pairs'):
            raise Exception('It seems like the set_block_sizes has not been called yet before calling set_weight_param')
        self.logistic_param = [logis_func_L, logis_func_x0, logis_func_k, logis_func_offset]
        self.max_dist = max_dist
        self.resolution = resolution
        self.edge_masks = self.get_edge_masks()

    def get_block_sizes(self):
        width = np.linspace(self.min_block_x, self.max_block_x, self.steps, dtype=int)
        height = np.linspace(self.min_block_y, self.max_block_y, self.steps, dtype=int)
        w, h = np.meshgrid(width, height)
        return np.array([(w // 2 * 2).flatten(), (h // 2 * 2).flatten()])

    def _logistic(self, dist):
        L, x0, k, off = self.logistic_param
        resc = np.where(dist > self.max_dist, 1, dist / self.max_dist)
        return resc, L / (1 + np.exp(-k * (resc - x0))) - off

    def get_edge_masks(self):
        if not hasattr(self, 'pairs'):
            raise Exception('It seems like the set_block_sizes has not been called yet before calling get_edge_mask')
        out = []
        for n in range(self.pairs.shape[1]):
            bw, bh = int(self.pairs[0, n]), int(self.pairs[1, n])
            jj, ii = np.meshgrid(np.arange(bw), np.arange(bh))
            # distance to the nearest border cell: the nearest one is straight up/down/left/right
            dist = np.minimum(np.minimum(ii, bh - 1 - ii), np.minimum(jj, bw - 1 - jj)) * self.resolution
            out.append(self._logistic(np.sqrt(dist ** 2))[1])
        return out

    def get_crf_weight(self, xx, yy, cond_data_mask):
        dist = min_dist_from_mask(xx, yy, cond_data_mask == 1)
        return self.get_crf_weight_from_dist(xx, yy, dist)

    def get_crf_weight_from_dist(self, xx, yy, dist):
        resc, logi = self._logistic(dist)
        return logi - np.min(logi), dist, resc, logi

    def get_random_field(self, X, Y, n=1):
        raise NotImplementedError("the gstools randomisation-method generator (MCMC.py:625-687) is not available "
                                  "in this build; use set_generation_method(True) (spectral synthesis)")

    def get_rfblock(self):
        """One masked proposal block with the reference's draw order (MCMC.py:742-778)."""
        idx = self.rng.integers(low=0, high=self.pairs.shape[1], size=1)[0]
        bw, bh = int(self.pairs[0, idx]), int(self.pairs[1, idx])
        if not getattr(self, 'spectral', False):
            return self.get_random_field(None, None)
        while True:
            f = spectral_synthesis_field(self, (bh, bw), res=self.resolution)
            if np.sum(np.isnan(f)) == 0:
                break
            print('f have nan')
        self._last_size_idx = int(idx)
        return f * self.edge_masks[idx]


def usable_cpus(cap=16):
    """CPUs this process may really use: the affinity mask, the cgroup CPU quota (containers report every host core in
    the mask), and a cap -- the default size of the host draw pool (the reference uses physical cores - 1,
    largeScaleChain_multiprocessing_GPU.py:472)."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, cap))


def draw_chunk(RF, rng, n, H, W, update_in_region, region_mask):
    """n Metropolis steps' worth of host draws in the reference's per-generator order: RF.rng gives the proposal block
    (MCMC.py:742-778 -> :176-254), `rng` (the chain's generator) the block centre -- rejection on region_mask when
    update_in_region (MCMC.py:1253-1261) -- and the accept uniform (MCMC.py:1336), which the reference always draws.
    None of this depends on the chain's state, so draws can be made ahead of the device (and in other processes)."""
    size_idx = np.empty(n, dtype=np.int32)
    centre = np.empty((n, 2), dtype=np.int32)
    u = np.empty(n)
    fields = []
    for s in range(n):
        f = RF.get_rfblock()
        fields.append(f)
        size_idx[s] = RF._last_size_idx
        if update_in_region:
            while True:
                ix = rng.integers(low=0, high=H, size=1)[0]
                iy = rng.integers(low=0, high=W, size=1)[0]
                if region_mask[ix, iy] == 1:
                    break
        else:
            ix = rng.integers(low=0, high=H, size=1)[0]
            iy = rng.integers(low=0, high=W, size=1)[0]
        centre[s] = (ix, iy)
        u[s] = rng.random()
    return size_idx, centre, u, fields


# ---- host processes that draw replay-mode proposals ahead of the device (run_many_replay) ------------------------------
class DrawWorkers:
    """n child processes (mcmc_gpu_amd/_draw_worker.py), each owning the NumPy generators of a fixed subset of chains."""

    def __init__(self, n_workers, n_chains, rf_param, H, W, update_in_region, region_mask, shm_names, shm_shape,
                 rf_states, chain_states):
        import os
        import subprocess
        from pathlib import Path
        from . import _draw_worker as proto
        self.proto = proto
        self.procs, self.fin, self.fout = [], [], []
        env = dict(os.environ)
        root = str(Path(__file__).resolve().parent.parent)
        env['PYTHONPATH'] = root + (os.pathsep + env['PYTHONPATH'] if env.get('PYTHONPATH') else '')
        env.setdefault('OMP_NUM_THREADS', '1'); env.setdefault('OPENBLAS_NUM_THREADS', '1')
        n_workers = max(1, min(int(n_workers), n_chains))
        try:
            for w in range(n_workers):
                p2c_r, p2c_w = os.pipe()
                c2p_r, c2p_w = os.pipe()
                self.procs.append(subprocess.Popen([sys.executable, '-m', 'mcmc_gpu_amd._draw_worker', str(p2c_r), str(c2p_w)],
                                                   pass_fds=(p2c_r, c2p_w), env=env, stdin=subprocess.DEVNULL))
                os.close(p2c_r); os.close(c2p_w)
                self.fout.append(os.fdopen(p2c_w, 'wb')); self.fin.append(os.fdopen(c2p_r, 'rb'))
            for w in range(n_workers):       # all workers are starting concurrently; now hand each its chains
                slots = list(range(w, n_chains, n_workers))
                proto.send(self.fout[w], dict(rf_param=rf_param, H=H, W=W, update_in_region=update_in_region, region_mask=region_mask,
                                              shm_names=shm_names, shm_shape=shm_shape, slots=slots,
                                              rf_states=[rf_states[c] for c in slots], chain_states=[chain_states[c] for c in slots]))
            for w in range(n_workers):
                if proto.recv(self.fin[w]) != 'ready':
                    raise RuntimeError('draw worker did not start')
        except BaseException:
            self.close(kill=True)
            raise

    def request(self, which, n):
        for f in self.fout:
            self.proto.send(f, ('draw', which, n))

    def collect(self):
        out = []
        for f in self.fin:
            out.extend(self.proto.recv(f))
        return out

    def states(self):
        for f in self.fout:
            self.proto.send(f, ('states',))
        st = {}
        for f in self.fin:
            st.update(self.proto.recv(f))
        return st

    def close(self, kill=False):
        for f in self.fout:
            try:
                if not kill:
                    self.proto.send(f, ('quit',))
                f.close()
            except Exception:
                pass
        for p in self.procs:
            try:
                if kill:
                    p.kill()
                p.wait(timeout=10)
            except Exception:
                p.kill()
        for f in self.fin:
            try:
                f.close()
            except Exception:
                pass
        self.procs, self.fin, self.fout = [], [], []


class chain_crf_gpu:
    """Large-scale random-field Metropolis chain executed on the MI355X (reference: chain / chain_crf /
    chain_crf_gpu, MCMC.py:780-1443, MCMC_gpu.py:149-582).  All state lives in plain attributes so the
    object can be rebuilt from a deep copy of its __dict__ (largeScaleChain_multiprocessing_GPU.py:56-75)."""

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
        self.rng_mode = 'replay'
        self.replay_chunk = 256
        self.philox_step = 0
        self.philox_batch = 8
        self.state_dtype = 'f64'   # 'f32': float32 bed/energy storage with float64 arithmetic (BASELINE configs[4])

    # ---- setters (MCMC.py:849-872, :950-1018, :1046-1081, :1098-1134) ---------------------------
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

    def loss(self, massConvResidual, dataDiff):
        loss_mc = np.nansum(np.square(massConvResidual[self.mc_region_mask == 1])) / (2 * self.sigma_mc ** 2)
        return loss_mc + 0, loss_mc, 0

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

    def set_sample_points_locations(self, loc):
        self.sample_loc = loc

    def set_update_type(self, block_type):
        if block_type not in ('CRF_weight', 'CRF_rbf', 'RF'):
            raise ValueError('The block_type argument should be one of the following: CRF_weight, CRF_rbf, RF')
        self.block_type = block_type

    def set_crf_data_weight(self, RF):
        self.crf_data_weight = RF.get_crf_weight(self.xx, self.yy, self.data_mask)[0]

    def set_rng_mode(self, mode, philox_batch=None):
        """'replay': NumPy draws and NumPy spectral synthesis on the host, the device does the step -- the reference-identical
        chain, bit for bit.  'pcg64': the SAME NumPy generator streams advanced on the device (gsm_draw_pcg64) and the device's
        spectral synthesis: the reference's draws, block records and accept decisions on the same seeds without host work per
        step (beds / losses to the accuracy of the device's inverse DFT).  'philox': counter-based device draws, a chain of its
        own definition (throughput mode)."""
        if mode not in ('replay', 'philox', 'pcg64'):
            raise ValueError("mode must be 'replay', 'pcg64' or 'philox'")
        self.rng_mode = mode
        if philox_batch:
            self.philox_batch = int(philox_batch)

    # ---- engine plumbing ---------------------------------------------------------------------------
    def _make_engine(self, RF, n_chains, device=None):
        from .engine import GsmEngine
        H, W = self.xx.shape
        eng = GsmEngine(H, W, n_chains, device, state_dtype=getattr(self, 'state_dtype', 'f64'))
        upd = self.region_mask if self.update_in_region else self.grounded_ice_mask
        weight = self.crf_data_weight if self.block_type == 'CRF_weight' else None
        eng.set_static(self.surf, self.velx, self.vely, self.dhdt, self.smb, weight, upd, self.mc_region_mask,
                       self.resolution, self.sigma_mc)
        eng.set_blocks(RF.pairs, RF.edge_masks)
        if self.update_in_region:
            eng.set_centres(self.region_mask)
        else:
            eng.set_centres(np.ones(self.xx.shape, dtype=np.uint8))
        return eng

    def _philox_seed(self):
        seed = getattr(self, 'rng_seed', None)
        if seed is None:
            seed = int(self.rng.bit_generator.seed_seq.entropy) if hasattr(self.rng.bit_generator, 'seed_seq') else 0
        return int(seed) & 0xFFFFFFFFFFFFFFFF

    def _draw_chunk(self, RF, n):
        """n steps of host draws in the reference's per-generator order."""
        H, W = self.xx.shape
        return draw_chunk(RF, self.rng, n, H, W, self.update_in_region, self.region_mask)

    def _sample_indices(self):
        loc = np.asarray(self.sample_loc)
        ij = np.zeros(loc.shape, dtype=np.int64)
        for k in range(loc.shape[0]):
            i, j = np.where((self.xx == loc[k, 0]) & (self.yy == loc[k, 1]))
            ij[k] = [int(i[0]), int(j[0])]
        return ij

    # ---- the chain ----------------------------------------------------------------------------------
    def run(self, n_iter, RF, only_save_last_bed=False, info_per_iter=1000, plot=True, progress_bar=True):
        """n_iter-1 Metropolis proposals from self.initial_bed; returns the reference's tuple
        (bed or bed_cache, loss_mc_cache, loss_data_cache, loss_cache, step_cache, resampled_times,
        blocks_cache[, sample_values]) as NumPy arrays (MCMC.py:1137-1443)."""
        if not isinstance(RF, RandField):
            raise TypeError('The arugment "RF" has to be an object of the class RandField')
        if not getattr(RF, 'spectral', False):
            raise NotImplementedError('only the spectral-synthesis generator (set_generation_method(True)) is built')
        if not hasattr(self, 'rng'):
            self.set_random_generator(getattr(self, 'rng_seed', None))
        H, W = self.xx.shape
        n_iter = int(n_iter)
        if n_iter < 1:
            raise ValueError('n_iter must be >= 1')
        loss_cache = np.zeros(n_iter)
        step_cache = np.zeros(n_iter)
        blocks_cache = np.full((n_iter, 4), np.nan)
        keep_all = not only_save_last_bed
        track = self.sample_loc is not None
        if keep_all:
            bed_cache = np.zeros((n_iter, H, W))
        if track:
            ij = self._sample_indices()
            sample_values = np.zeros((ij.shape[0], n_iter))

        eng = self._make_engine(RF, 1)
        try:
            bed0 = np.asarray(self.initial_bed, dtype=np.float64)
            loss_cache[0] = eng.set_state(bed0[None])[0]
            if keep_all:
                bed_cache[0] = bed0
            if track:
                sample_values[:, 0] = bed0[ij[:, 0], ij[:, 1]]
            per_step = keep_all or track
            chunk = 1 if per_step else (self.replay_chunk if self.rng_mode in ('replay', 'pcg64') else max(n_iter - 1, 1))
            if self.rng_mode == 'pcg64':
                import ctypes as C
                import torch
                from .engine import GsmEngine, _ptr
                if RF.rng is self.rng or RF.rng.bit_generator is self.rng.bit_generator:
                    # the device advances the two streams independently; one shared generator interleaves its draws between the
                    # proposal and the chain (set_random_generator's docstring) -- only the host-drawn replay mode follows that
                    raise ValueError("'pcg64' mode needs separate generators for the RandField and the chain (a shared generator "
                                     "interleaves the two draw sequences: use the 'replay' mode for that)")
                p64 = eng.rf_struct(RF)
                d_rf = torch.as_tensor(GsmEngine.pack_pcg64_states([RF.rng]).view(np.int64)).to(eng.dev)
                d_ch = torch.as_tensor(GsmEngine.pack_pcg64_states([self.rng]).view(np.int64)).to(eng.dev)
                d_reg = (torch.as_tensor(np.ascontiguousarray(np.asarray(self.region_mask) == 1, dtype=np.uint8)).to(eng.dev)
                         if self.update_in_region else None)
            t0 = time.time()
            done = 1
            next_info = info_per_iter
            while done < n_iter:
                n = min(chunk, n_iter - done)
                if self.rng_mode == 'pcg64':
                    d = eng.draw_pcg64(n, p64, d_rf, d_ch, d_reg)
                    fl = torch.zeros((1, n, eng.field_stride), dtype=torch.float64, device=eng.dev)
                    l_b = torch.empty((1, n), dtype=torch.float64, device=eng.dev)
                    a_b = torch.empty((1, n), dtype=torch.uint8, device=eng.dev)
                    with torch.cuda.device(eng.dev):
                        eng._check(eng.lib.gsm_spectral_from_noise(eng.h, n, _ptr(d['size_idx']), _ptr(d['rf_scalars']), C.byref(p64),
                                                                   _ptr(d['noise_re']), _ptr(d['noise_im']), _ptr(d['nugget']), _ptr(fl),
                                                                   eng.field_stride, eng._stream()))
                        eng._check(eng.lib.gsm_run_replay(eng.h, n, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum),
                                                          _ptr(d['size_idx']), _ptr(d['centre']), _ptr(d['u']), _ptr(fl), eng.field_stride,
                                                          _ptr(l_b), _ptr(a_b), eng._stream()))
                    loss, acc = l_b.cpu().numpy(), a_b.cpu().numpy()
                    si = d['size_idx'][0].cpu().numpy()
                    blocks_cache[done:done + n, 0:2] = d['centre'][0].cpu().numpy()
                    blocks_cache[done:done + n, 2] = eng.bh[si]
                    blocks_cache[done:done + n, 3] = eng.bw[si]
                elif self.rng_mode == 'replay':
                    si, ce, u, fields = self._draw_chunk(RF, n)
                    loss, acc = eng.run_replay(si[None], ce[None], u[None], eng.pack_fields([fields]))
                    blocks_cache[done:done + n, 0:2] = ce
                    blocks_cache[done:done + n, 2] = eng.bh[si]
                    blocks_cache[done:done + n, 3] = eng.bw[si]
                else:
                    loss, acc, blk = eng.run_philox(n, self.philox_step, [self._philox_seed()], RF, batch=self.philox_batch)
                    self.philox_step += n
                    blocks_cache[done:done + n] = blk[0]
                loss_cache[done:done + n] = loss[0]
                step_cache[done:done + n] = acc[0]
                if per_step:
                    b = eng.beds[0].double().cpu().numpy()
                    if keep_all:
                        bed_cache[done] = b
                    if track:
                        sample_values[:, done] = b[ij[:, 0], ij[:, 1]]
                done += n
                if progress_bar is not None and (done >= next_info or done == n_iter):
                    next_info = done + info_per_iter
                    el = time.time() - t0
                    print(f"Chain {getattr(self, 'chain_id', 0)} ({str(getattr(self, 'seed', 'Unknown'))[:6]}): "
                          f"{100 * (done - 1) / max(n_iter - 1, 1):3.0f}% | it/s: {(done - 1) / max(el, 1e-9):8.1f} | "
                          f"n: {n_iter} | loss: {loss_cache[done - 1]:.3e} | acc: {step_cache[:done].sum() / done:.4f}",
                          file=sys.stdout, flush=True)
            bed_c = eng.beds[0].double().cpu().numpy()
            resampled = eng.resampled[0].cpu().numpy().astype(np.float64)
            if self.rng_mode == 'pcg64':           # the generators continue where the device left them, as after NumPy calls
                RF.rng.bit_generator.state = GsmEngine.unpack_pcg64_states(d_rf.cpu().numpy().view(np.uint64))[0]
                self.rng.bit_generator.state = GsmEngine.unpack_pcg64_states(d_ch.cpu().numpy().view(np.uint64))[0]
        finally:
            eng.close()
        loss_mc_cache = loss_cache.copy()
        loss_data_cache = np.zeros(n_iter)
        out = (bed_cache if keep_all else bed_c, loss_mc_cache, loss_data_cache, loss_cache, step_cache, resampled, blocks_cache)
        return out + (sample_values,) if track else out


def run_many(chain, RF, initial_beds, seeds, n_iter, batch=8, device=None, step0=0, return_device=False):
    """Batched entry the reference lacks: n independent Philox-mode chains of one template (same static fields,
    same RandField) on one GPU.  Returns a list of per-chain 7-tuples shaped like chain.run(..., only_save_last_bed=True)
    (what Pool.starmap returns in largeScaleChain_mp, largeScaleChain_multiprocessing_GPU.py:84-85)."""
    if not isinstance(RF, RandField):
        raise TypeError('The arugment "RF" has to be an object of the class RandField')
    beds = np.asarray(initial_beds, dtype=np.float64)
    n_chains = beds.shape[0]
    if len(seeds) != n_chains:
        raise ValueError('need one seed per chain')
    eng = chain._make_engine(RF, n_chains, device)
    try:
        loss0 = eng.set_state(beds)
        n_steps = int(n_iter) - 1
        if n_steps > 0:
            loss, acc, blk = eng.run_philox(n_steps, step0, [int(s) & 0xFFFFFFFFFFFFFFFF for s in seeds], RF, batch=batch)
        else:
            loss = np.zeros((n_chains, 0)); acc = np.zeros((n_chains, 0), np.uint8); blk = np.zeros((n_chains, 0, 4), np.int32)
        if return_device:
            return eng, loss0, loss, acc, blk
        beds_out = eng.beds.double().cpu().numpy()
        res = eng.resampled.cpu().numpy().astype(np.float64)
    finally:
        if not return_device:
            eng.close()
    out = []
    for c in range(n_chains):
        lc = np.concatenate([[loss0[c]], loss[c]])
        sc = np.concatenate([[0.0], acc[c].astype(np.float64)])
        bc = np.vstack([np.full((1, 4), np.nan), blk[c].astype(np.float64)])
        out.append((beds_out[c], lc.copy(), np.zeros(int(n_iter)), lc, sc, res[c], bc))
    return out


def run_many_replay(chain, RF, initial_beds, rf_states, chain_states, n_iter, chunk=None, n_workers=None, device=None,
                    progress=False):
    """Replay mode for MANY chains of one template in ONE handle: what the reference's pool computes with one chain per
    process (largeScaleChain_multiprocessing_GPU.py:84-85).  Chain c draws from its own two NumPy generators
    (rf_states[c], chain_states[c]: `Generator.bit_generator.state` dicts) exactly as chain_crf.run does; the draws for
    chunk k+1 are made by a pool of host processes (they do not depend on the chains' state) while the device steps
    chunk k; the fields travel through shared memory.  Returns (list of chain.run(..., only_save_last_bed=True) tuples,
    final rf states, final chain states) -- results equal n independent chain_crf_gpu.run calls."""
    from multiprocessing import shared_memory
    import torch
    if not isinstance(RF, RandField):
        raise TypeError('The arugment "RF" has to be an object of the class RandField')
    if not getattr(RF, 'spectral', False):
        raise NotImplementedError('only the spectral-synthesis generator (set_generation_method(True)) is built')
    beds = np.asarray(initial_beds, dtype=np.float64)
    n_chains = beds.shape[0]
    if len(rf_states) != n_chains or len(chain_states) != n_chains:
        raise ValueError('need one RandField and one chain generator state per chain')
    H, W = chain.xx.shape
    n_iter = int(n_iter)
    n_steps = n_iter - 1
    eng = chain._make_engine(RF, n_chains, device)
    shms = []
    pool = None
    try:
        loss0 = eng.set_state(beds)
        loss = np.zeros((n_chains, n_steps)); acc = np.zeros((n_chains, n_steps), np.uint8)
        blocks = np.zeros((n_chains, n_steps, 4))
        rf_states, chain_states = list(rf_states), list(chain_states)
        if n_steps > 0:
            stride = eng.field_stride
            if chunk is None:   # two shared buffers of at most ~256 MiB each
                chunk = int(max(4, min(256, (256 << 20) // (n_chains * stride * 8))))
            chunk = max(1, min(int(chunk), n_steps))
            shape = (n_chains, chunk, stride)
            shms = [shared_memory.SharedMemory(create=True, size=int(np.prod(shape)) * 8) for _ in range(2)]
            bufs = [np.ndarray(shape, dtype=np.float64, buffer=m.buf) for m in shms]
            if n_workers is None:
                n_workers = max(1, usable_cpus() - 1)
            n_workers = max(1, min(int(n_workers), n_chains))
            rf_param = {k: v for k, v in RF.__dict__.items() if k not in ('rng', '_last_size_idx')}
            pool = DrawWorkers(n_workers, n_chains, rf_param, H, W, chain.update_in_region, np.asarray(chain.region_mask),
                               [m.name for m in shms], shape, rf_states, chain_states)
            n_chunks = (n_steps + chunk - 1) // chunk
            pool.request(0, min(chunk, n_steps))
            t0 = time.time()
            t_wait = t_dev = 0.0
            for k in range(n_chunks):
                n = min(chunk, n_steps - k * chunk)
                si = np.empty((n_chains, n), np.int32); ce = np.empty((n_chains, n, 2), np.int32); u = np.empty((n_chains, n))
                tw = time.time()
                drawn = pool.collect()
                t_wait += time.time() - tw
                for slot, a, b, c_ in drawn:
                    si[slot], ce[slot], u[slot] = a, b, c_
                if k + 1 < n_chunks:                   # drawn on the host cores while the device steps chunk k
                    pool.request((k + 1) & 1, min(chunk, n_steps - (k + 1) * chunk))
                fields = torch.from_numpy(bufs[k & 1])
                if n < chunk:
                    fields = fields[:, :n].contiguous()
                lo = k * chunk
                tw = time.time()
                loss[:, lo:lo + n], acc[:, lo:lo + n] = eng.run_replay(si, ce, u, fields.to(eng.dev))
                t_dev += time.time() - tw
                blocks[:, lo:lo + n, 0:2] = ce
                blocks[:, lo:lo + n, 2] = eng.bh[si]
                blocks[:, lo:lo + n, 3] = eng.bw[si]
                if progress:
                    done = lo + n
                    print(f"{n_chains} chains: {100 * done / n_steps:3.0f}% | chain-it/s: {n_chains * done / max(time.time() - t0, 1e-9):9.1f} | "
                          f"acc: {acc[:, :done].mean():.4f} | waited for draws {t_wait:.2f} s, upload + device {t_dev:.2f} s",
                          file=sys.stdout, flush=True)
            st = pool.states()
            rf_states = [st[c][0] for c in range(n_chains)]
            chain_states = [st[c][1] for c in range(n_chains)]
            pool.close()
            pool = None
        beds_out = eng.beds.double().cpu().numpy()
        res = eng.resampled.cpu().numpy().astype(np.float64)
    finally:
        eng.close()
        if pool is not None:
            pool.close(kill=True)
        for m in shms:
            m.close()
            m.unlink()
    out = []
    for c in range(n_chains):
        lc = np.concatenate([[loss0[c]], loss[c]])
        sc = np.concatenate([[0.0], acc[c].astype(np.float64)])
        bc = np.vstack([np.full((1, 4), np.nan), blocks[c]])
        out.append((beds_out[c], lc.copy(), np.zeros(n_iter), lc, sc, res[c], bc))
    return out, rf_states, chain_states


def run_many_pcg64(chain, RF, initial_beds, rf_states, chain_states, n_iter, batch=None, device=None, progress=False, fused=None,
                   timing=None):
    """The 'pcg64' draw mode: run_many_replay with NO host draws.  Chain c's two NumPy generators (rf_states[c], chain_states[c]:
    `Generator.bit_generator.state` dicts of PCG64 generators) are advanced ON THE DEVICE, bit for bit as NumPy advances them
    (gsm_draw_pcg64: PCG64, the 32-bit half cache of integers(), Lemire's bounded integers, uniform, the ziggurat normal), in
    the reference's call order; the white-noise planes go through the device spectral synthesis (gsm_spectral_from_noise,
    value-pinned to the reference's spectral_synthesis_field at 1e-12 x scale) and the fields through gsm_run_replay.  Same
    return value as run_many_replay.  Against the CPU reference on the same seeds: block records, the draws and the final
    generator states are identical; accept masks are identical unless an accept uniform falls within ~1e-12 of its threshold;
    beds and losses agree to the accuracy of the device's inverse DFT against pocketfft (tests/test_gpu_pcg64.py).
    fused (default: whenever the block table goes to the strip kernels): synthesis and step in ONE kernel per batch
    (gsm_run_noise: the field never leaves the CU) -- bit-identical to the two calls.
    timing: a dict that receives 'loop_seconds' -- the draw / synthesis / step batches alone, chain state resident in HBM
    (synchronised before and after) -- and 'fused'."""
    import ctypes as C
    import torch
    from .engine import GsmEngine, _ptr
    if not isinstance(RF, RandField):
        raise TypeError('The arugment "RF" has to be an object of the class RandField')
    if not getattr(RF, 'spectral', False):
        raise NotImplementedError('only the spectral-synthesis generator (set_generation_method(True)) is built')
    beds = np.asarray(initial_beds, dtype=np.float64)
    n_chains = beds.shape[0]
    if len(rf_states) != n_chains or len(chain_states) != n_chains:
        raise ValueError('need one RandField and one chain generator state per chain')
    n_iter = int(n_iter)
    n_steps = n_iter - 1
    eng = chain._make_engine(RF, n_chains, device)
    try:
        dev = eng.dev
        loss0 = eng.set_state(beds)
        p = eng.rf_struct(RF)
        d_rf = torch.as_tensor(GsmEngine.pack_pcg64_states(list(rf_states)).view(np.int64)).to(dev)
        d_ch = torch.as_tensor(GsmEngine.pack_pcg64_states(list(chain_states)).view(np.int64)).to(dev)
        d_reg = (torch.as_tensor(np.ascontiguousarray(np.asarray(chain.region_mask) == 1, dtype=np.uint8)).to(dev)
                 if chain.update_in_region else None)
        stride = eng.field_stride
        if fused is None:
            fused = eng.strip_active() and os.environ.get("GSM_PCG64_FUSED", "1") != "0"
        elif fused and not eng.strip_active():
            raise NotImplementedError("run_many_pcg64(fused=True): this block table does not go to the strip kernels")
        if batch is None:
            # Two sets of three noise planes (+ the fields of a batch without the fused kernel): at most ~48 GiB of the 288.  Long
            # batches matter: the draw kernel and the chain kernel are persistent workgroups that fill the chip one after the
            # other (4.2 M chain-steps/s at 46 steps per batch, 4.8 M at 128; 1024 chains at 256 x 256, state resident).
            batch = int(max(1, min(256, (48 << 30) // ((6 if fused else 7) * n_chains * stride * 8))))
        batch = max(1, min(int(batch), max(n_steps, 1)))
        n_al = max(n_steps, 1)
        loss = torch.empty((n_chains, n_al), dtype=torch.float64, device=dev)
        acc = torch.empty((n_chains, n_al), dtype=torch.uint8, device=dev)
        si_all = torch.empty((n_chains, n_al), dtype=torch.int32, device=dev)
        blocks = np.zeros((n_chains, n_steps, 4))
        ce_all = torch.empty((n_chains, n_al, 2), dtype=torch.int32, device=dev)
        fields = None if fused else torch.zeros((n_chains, batch, stride), dtype=torch.float64, device=dev)
        nug = p.nugget_max > 0.0
        bufs = [eng.alloc_pcg64_buffers(batch, nug), eng.alloc_pcg64_buffers(batch, nug)]
        # Two streams: the draws of batch k + 1 (one wavefront per chain: latency-bound, leaves most of the chip idle) run beside the
        # spectral synthesis and the steps of batch k.  Events order the reuse of the two draw buffers.
        s_draw, s_step = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
        cur = torch.cuda.current_stream(dev)
        s_draw.wait_stream(cur); s_step.wait_stream(cur)
        ev_drawn = [torch.cuda.Event(), torch.cuda.Event()]
        ev_free = [None, None]
        sizes = [min(batch, n_steps - lo) for lo in range(0, n_steps, batch)]

        def draw(k):
            with torch.cuda.stream(s_draw):
                if ev_free[k & 1] is not None:
                    s_draw.wait_event(ev_free[k & 1])
                n = sizes[k]
                d = eng.draw_pcg64(n, p, d_rf, d_ch, d_reg, buffers=bufs[k & 1] if n == batch else None)
                ev_drawn[k & 1].record(s_draw)
            return d

        torch.cuda.synchronize(dev)
        t0 = time.time()
        done = 0
        nxt = draw(0) if sizes else None
        for k, n in enumerate(sizes):
            d = nxt
            nxt = draw(k + 1) if k + 1 < len(sizes) else None
            with torch.cuda.stream(s_step):
                s_step.wait_event(ev_drawn[k & 1])
                l_b = torch.empty((n_chains, n), dtype=torch.float64, device=dev)
                a_b = torch.empty((n_chains, n), dtype=torch.uint8, device=dev)
                if fused:
                    eng._check(eng.lib.gsm_run_noise(eng.h, n, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum),
                                                     _ptr(d['size_idx']), _ptr(d['centre']), _ptr(d['u']), _ptr(d['rf_scalars']), C.byref(p),
                                                     _ptr(d['noise_re']), _ptr(d['noise_im']), _ptr(d['nugget']), stride,
                                                     _ptr(l_b), _ptr(a_b), eng._stream()))
                else:
                    fl = fields if n == batch else torch.zeros((n_chains, n, stride), dtype=torch.float64, device=dev)
                    eng._check(eng.lib.gsm_spectral_from_noise(eng.h, n_chains * n, _ptr(d['size_idx']), _ptr(d['rf_scalars']), C.byref(p),
                                                               _ptr(d['noise_re']), _ptr(d['noise_im']), _ptr(d['nugget']), _ptr(fl), stride,
                                                               eng._stream()))
                    eng._check(eng.lib.gsm_run_replay(eng.h, n, _ptr(eng.beds), _ptr(eng.energy), _ptr(eng.resampled), _ptr(eng.loss_sum),
                                                      _ptr(d['size_idx']), _ptr(d['centre']), _ptr(d['u']), _ptr(fl), stride,
                                                      _ptr(l_b), _ptr(a_b), eng._stream()))
                loss[:, done:done + n] = l_b
                acc[:, done:done + n] = a_b
                si_all[:, done:done + n] = d['size_idx']
                ce_all[:, done:done + n] = d['centre']
                ev_free[k & 1] = torch.cuda.Event()
                ev_free[k & 1].record(s_step)
            done += n
            if progress:
                s_step.synchronize()
                print(f"{n_chains} chains: {100 * done / n_steps:3.0f}% | chain-it/s: {n_chains * done / max(time.time() - t0, 1e-9):9.1f}",
                      file=sys.stdout, flush=True)
        cur.wait_stream(s_draw); cur.wait_stream(s_step)
        torch.cuda.synchronize(dev)
        if timing is not None:
            timing['loop_seconds'] = time.time() - t0
            timing['fused'] = bool(fused)
            timing['batch'] = batch
        si = si_all[:, :n_steps].cpu().numpy()
        blocks[:, :, 0:2] = ce_all[:, :n_steps].cpu().numpy()
        blocks[:, :, 2] = eng.bh[si]
        blocks[:, :, 3] = eng.bw[si]
        loss_h = loss[:, :n_steps].cpu().numpy()
        acc_h = acc[:, :n_steps].cpu().numpy()
        rf_out = GsmEngine.unpack_pcg64_states(d_rf.cpu().numpy().view(np.uint64))
        ch_out = GsmEngine.unpack_pcg64_states(d_ch.cpu().numpy().view(np.uint64))
        beds_out = eng.beds.double().cpu().numpy()
        res = eng.resampled.cpu().numpy().astype(np.float64)
    finally:
        eng.close()
    out = []
    for c in range(n_chains):
        lc = np.concatenate([[loss0[c]], loss_h[c]])
        sc = np.concatenate([[0.0], acc_h[c].astype(np.float64)])
        bc = np.vstack([np.full((1, 4), np.nan), blocks[c]])
        out.append((beds_out[c], lc.copy(), np.zeros(n_iter), lc, sc, res[c], bc))
    return out, rf_out, ch_out


def init_lsc_chain_by_instance(param_dict):
    """Rebuild a chain from a copy of another chain's __dict__ (+ 'rng_seed') (MCMC.py:359-377)."""
    p = param_dict
    ch = chain_crf_gpu(p['xx'], p['yy'], p['initial_bed'], p['surf'], p['velx'], p['vely'], p['dhdt'], p['smb'],
                       p['cond_bed'], p['data_mask'], p['grounded_ice_mask'], p['resolution'])
    ch.update_in_region = p['update_in_region']
    ch.region_mask = p['region_mask']
    ch.sigma_mc = p['sigma_mc']
    ch.block_type = p['block_type']
    ch.crf_data_weight = p['crf_data_weight']
    ch.rng = np.random.default_rng(seed=p['rng_seed'])
    ch.rng_seed = p['rng_seed']
    ch.mc_region_mask = p['mc_region_mask']
    ch.sample_loc = deepcopy(p['sample_loc'])
    for k in ('rng_mode', 'replay_chunk', 'philox_step', 'philox_batch', 'state_dtype'):
        if k in p:
            setattr(ch, k, p[k])
    return ch


def initiate_RF_by_instance(param_dict):
    """Rebuild a RandField from a copy of another one's __dict__ (+ 'rng_seed') (MCMC.py:381-398)."""
    p = param_dict
    rf = RandField(p['range_min_x'], p['range_max_x'], p['range_min_y'], p['range_max_y'], p['scale_min'],
                   p['scale_max'], p['nugget_max'], p['model_name'], p['isotropic'], smoothness=p['smoothness'])
    rf.spectral = p['spectral']
    rf.min_block_x, rf.min_block_y = p['min_block_x'], p['min_block_y']
    rf.max_block_x, rf.max_block_y = p['max_block_x'], p['max_block_y']
    rf.steps = p['steps']
    rf.pairs = p['pairs']
    rf.logistic_param = p['logistic_param']
    rf.max_dist = p['max_dist']
    rf.resolution = p['resolution']
    rf.edge_masks = p['edge_masks']
    rf.rng = np.random.default_rng(seed=p['rng_seed'])
    return rf
