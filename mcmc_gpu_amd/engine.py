"""GsmEngine: owns one libgsm_hip handle and the torch device tensors of a shard of chains.

PyTorch is plumbing here -- device memory, streams and (in parallel.py) torch.distributed.  All compute
is in the HIP library behind the C ABI of include/gsm.h.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import GsmError, RfParams, MODEL_IDS


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def binary_mask(mask, name):
    """Masks of this path are 0/1 (bool or numeric).  The reference mixes `mask == 1` and truthiness
    tests (MCMC.py:1257, :1288, :1328); they agree only for 0/1 masks, so anything else is refused."""
    m = np.asarray(mask)
    if m.dtype != bool:
        if np.isnan(m.astype(float)).any() or not np.isin(m, (0, 1)).all():
            raise ValueError(f"{name} must contain only 0/1 (or bool) values")
    return np.ascontiguousarray(m.astype(np.uint8))


class GsmEngine:
    def __init__(self, H: int, W: int, n_chains: int, device: int | None = None, state_dtype="f64"):
        if not torch.cuda.is_available():
            raise RuntimeError("no ROCm/HIP device visible to torch: the sampler has no CPU fallback")
        self.lib = _lib.load()
        self.H, self.W, self.n_chains = int(H), int(W), int(n_chains)
        self.device_index = torch.cuda.current_device() if device is None else int(device)
        self.dev = torch.device("cuda", self.device_index)
        h = C.c_void_p()
        if state_dtype not in ("f64", "f32"):
            raise ValueError("state_dtype must be 'f64' or 'f32'")
        self.state_dtype = torch.float64 if state_dtype == "f64" else torch.float32
        rc = self.lib.gsm_create(C.byref(h), self.H, self.W, self.n_chains, 0 if state_dtype == "f64" else 1,
                                 self.device_index)
        if rc != 0:
            raise GsmError(rc, self.lib.gsm_last_error(None).decode())
        self.h = h
        self.beds = self.energy = self.resampled = self.loss_sum = None
        self.field_stride = 0
        self.n_sizes = 0

    # ------------------------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None):
            self.lib.gsm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc):
        if rc != 0:
            raise GsmError(rc, self.lib.gsm_last_error(self.h).decode())

    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.dev).cuda_stream)

    def _f64(self, a):
        return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.dev)

    # ------------------------------------------------------------------------------------------
    def set_static(self, surf, velx, vely, dhdt, smb, crf_weight, update_mask, mc_mask, resolution, sigma_mc):
        shp = (self.H, self.W)
        arrs = []
        for a in (surf, velx, vely, dhdt, smb):
            a = np.asarray(a)
            if a.shape != shp:
                raise ValueError(f"static field of shape {a.shape}, grid is {shp}")
            arrs.append(self._f64(a))
        w = None
        if crf_weight is not None:
            if np.asarray(crf_weight).shape != shp:
                raise ValueError("crf_data_weight has the wrong shape")
            w = self._f64(crf_weight)
        um = torch.as_tensor(binary_mask(update_mask, "update mask")).to(self.dev)
        mm = torch.as_tensor(binary_mask(mc_mask, "mc_region_mask")).to(self.dev)
        if um.shape != shp or mm.shape != shp:
            raise ValueError("mask has the wrong shape")
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_set_static(self.h, *[_ptr(t) for t in arrs], _ptr(w), _ptr(um), _ptr(mm),
                                                float(resolution), float(sigma_mc), self._stream()))

    def set_blocks(self, pairs, edge_masks=None):
        """pairs: (2, n) int array, row 0 widths, row 1 heights (RandField.pairs, MCMC.py:576-579)."""
        pairs = np.asarray(pairs)
        n = pairs.shape[1]
        bw = (C.c_int32 * n)(*[int(v) for v in pairs[0]])
        bh = (C.c_int32 * n)(*[int(v) for v in pairs[1]])
        packed = offs = None
        if edge_masks is not None:
            o, chunks, tot = [], [], 0
            for i in range(n):
                m = np.ascontiguousarray(edge_masks[i], dtype=np.float64)
                if m.shape != (int(pairs[1, i]), int(pairs[0, i])):
                    raise ValueError("edge mask shape does not match its block size")
                o.append(tot)
                chunks.append(m.ravel())
                tot += m.size
            packed = self._f64(np.concatenate(chunks))
            offs = (C.c_int64 * n)(*o)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_set_blocks(self.h, n, bh, bw, _ptr(packed), offs, self._stream()))
        self.bh = np.array(pairs[1], dtype=np.int64)
        self.bw = np.array(pairs[0], dtype=np.int64)
        self.n_sizes = n
        self.field_stride = int(self.bh.max() * self.bw.max())

    def set_centres(self, region_mask):
        cells = np.flatnonzero(np.asarray(region_mask).ravel() == 1).astype(np.int32)
        if cells.size == 0:
            raise ValueError("region_mask has no cell equal to 1")
        t = torch.as_tensor(cells).to(self.dev)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_set_centres(self.h, _ptr(t), int(cells.size), self._stream()))

    def set_state(self, beds, resampled=None):
        """beds: (n_chains, H, W) array or cuda tensor.  Returns loss_cache[0] per chain (numpy)."""
        if isinstance(beds, torch.Tensor):
            b = beds.to(device=self.dev, dtype=self.state_dtype).contiguous()
            if b.data_ptr() == beds.data_ptr():
                b = b.clone()          # the engine updates its state in place: never alias the caller's tensor
        else:
            b = self._f64(beds).to(self.state_dtype)
        if tuple(b.shape) != (self.n_chains, self.H, self.W):
            raise ValueError(f"beds must have shape {(self.n_chains, self.H, self.W)}, got {tuple(b.shape)}")
        self.beds = b
        if resampled is None:
            self.resampled = torch.zeros((self.n_chains, self.H, self.W), dtype=torch.int32, device=self.dev)
        else:
            self.resampled = torch.as_tensor(resampled).to(device=self.dev, dtype=torch.int32).contiguous()
        self.loss_sum = torch.zeros((self.n_chains, 2), dtype=torch.float64, device=self.dev)
        self.energy = torch.empty((self.n_chains, self.H, self.W), dtype=self.state_dtype, device=self.dev)
        loss0 = torch.zeros(self.n_chains, dtype=torch.float64, device=self.dev)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_init_loss(self.h, _ptr(self.beds), _ptr(self.energy), _ptr(self.loss_sum), _ptr(loss0),
                                               self._stream()))
        return loss0.cpu().numpy()

    def min_dist_from_mask(self, xx, yy, mask):
        """Distance of every cell to the nearest masked cell on the device (Utilities.py:21-24)."""
        m = np.ascontiguousarray(np.asarray(mask) != 0).astype(np.uint8)
        if m.shape != (self.H, self.W):
            raise ValueError("mask has the wrong shape")
        dx, dy, dm = self._f64(xx), self._f64(yy), torch.as_tensor(m).to(self.dev)
        out = torch.empty((self.H, self.W), dtype=torch.float64, device=self.dev)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_min_dist_from_mask(self.h, _ptr(dx), _ptr(dy), _ptr(dm), _ptr(out), self._stream()))
        return out.cpu().numpy()

    def residual(self, beds):
        b = beds.to(device=self.dev, dtype=torch.float64).contiguous() if isinstance(beds, torch.Tensor) else self._f64(beds)
        if tuple(b.shape) != (self.n_chains, self.H, self.W):
            raise ValueError("beds has the wrong shape")
        out = torch.empty_like(b)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_residual(self.h, _ptr(b), _ptr(out), self._stream()))
        return out

    # ------------------------------------------------------------------------------------------
    def pack_fields(self, fields):
        """fields[c][s] = masked proposal (bh, bw) -> (n_chains, n_steps, field_stride) float64."""
        n_steps = len(fields[0])
        out = np.zeros((self.n_chains, n_steps, self.field_stride))
        for c in range(self.n_chains):
            for s in range(n_steps):
                f = np.asarray(fields[c][s], dtype=np.float64)
                out[c, s, :f.size] = f.ravel()
        return out

    def run_replay(self, size_idx, centre, u, fields_packed):
        """size_idx (n_chains, n_steps) int, centre (n_chains, n_steps, 2) int, u (n_chains, n_steps),
        fields_packed (n_chains, n_steps, field_stride).  Returns (loss, accept) as numpy arrays."""
        size_idx = np.ascontiguousarray(size_idx, dtype=np.int32)
        n_steps = size_idx.shape[1]
        centre = np.ascontiguousarray(centre, dtype=np.int32)
        u = np.ascontiguousarray(u, dtype=np.float64)
        if size_idx.shape != (self.n_chains, n_steps) or centre.shape != (self.n_chains, n_steps, 2) or \
                u.shape != (self.n_chains, n_steps):
            raise ValueError("replay arrays have inconsistent shapes")
        if (size_idx < 0).any() or (size_idx >= self.n_sizes).any():
            raise ValueError("size_idx out of range")
        if (centre < 0).any() or (centre[..., 0] >= self.H).any() or (centre[..., 1] >= self.W).any():
            raise ValueError("block centre outside the grid")
        if isinstance(fields_packed, torch.Tensor):
            f = fields_packed.to(device=self.dev, dtype=torch.float64).contiguous()
        else:
            f = self._f64(fields_packed)
        if tuple(f.shape) != (self.n_chains, n_steps, self.field_stride):
            raise ValueError("fields_packed has the wrong shape")
        if self.beds is None:
            raise RuntimeError("set_state() first")
        d_si = torch.as_tensor(size_idx).to(self.dev)
        d_c = torch.as_tensor(centre).to(self.dev)
        d_u = torch.as_tensor(u).to(self.dev)
        loss = torch.empty((self.n_chains, n_steps), dtype=torch.float64, device=self.dev)
        acc = torch.empty((self.n_chains, n_steps), dtype=torch.uint8, device=self.dev)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_run_replay(self.h, n_steps, _ptr(self.beds), _ptr(self.energy), _ptr(self.resampled), _ptr(self.loss_sum),
                                                _ptr(d_si), _ptr(d_c), _ptr(d_u), _ptr(f), self.field_stride,
                                                _ptr(loss), _ptr(acc), self._stream()))
        return loss.cpu().numpy(), acc.cpu().numpy()

    # ------------------------------------------------------------------------------------------
    @staticmethod
    def rf_struct(rf, resolution=None) -> RfParams:
        """RfParams from any object with the RandField attributes (MCMC.py:497-510)."""
        p = RfParams()
        p.range_min_x, p.range_max_x = float(rf.range_min_x), float(rf.range_max_x)
        p.range_min_y, p.range_max_y = float(rf.range_min_y), float(rf.range_max_y)
        p.scale_min, p.scale_max = float(rf.scale_min), float(rf.scale_max)
        p.nugget_max = float(rf.nugget_max)
        p.smoothness = float(rf.smoothness) if rf.smoothness else 0.0
        p.resolution = float(resolution if resolution is not None else rf.resolution)
        p.model = MODEL_IDS[rf.model_name]
        p.isotropic = 1 if rf.isotropic else 0
        p.generator = 1 if getattr(rf, "generator", "spectral") == "cholesky" else 0
        return p

    def _seeds(self, seeds):
        s = np.ascontiguousarray(np.asarray(seeds, dtype=np.uint64))
        if s.shape != (self.n_chains,):
            raise ValueError("need one seed per chain")
        return torch.as_tensor(s.view(np.int64)).to(self.dev)

    def propose_philox(self, n_steps, step0, seeds, rf, want_scalars=True):
        d_seeds = self._seeds(seeds)
        n = self.n_chains * n_steps
        si = torch.empty(n, dtype=torch.int32, device=self.dev)
        ce = torch.empty(n * 2, dtype=torch.int32, device=self.dev)
        u = torch.empty(n, dtype=torch.float64, device=self.dev)
        fl = torch.zeros((self.n_chains, n_steps, self.field_stride), dtype=torch.float64, device=self.dev)
        sc = torch.empty(n * 4, dtype=torch.float64, device=self.dev) if want_scalars else None
        p = rf if isinstance(rf, RfParams) else self.rf_struct(rf)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_propose_philox(self.h, int(n_steps), int(step0), _ptr(d_seeds), C.byref(p),
                                                    _ptr(si), _ptr(ce), _ptr(u), _ptr(fl), self.field_stride,
                                                    _ptr(sc), self._stream()))
            torch.cuda.synchronize(self.dev)
        return dict(size_idx=si.view(self.n_chains, n_steps), centre=ce.view(self.n_chains, n_steps, 2),
                    u=u.view(self.n_chains, n_steps), fields=fl,
                    rf_scalars=None if sc is None else sc.view(self.n_chains, n_steps, 4))

    def spectral_from_noise(self, size_idx, rf_scalars, rf, noise_re, noise_im, nugget_fields=None):
        """The device spectral synthesis on caller-supplied white noise (gsm_spectral_from_noise): for field r,
        size_idx[r] picks the block shape, rf_scalars[r] = (scale, nugget, range_x, range_y) as drawn at
        MCMC.py:200-207 (scale already / 3), noise_re[r] / noise_im[r] are the two (bh, bw) normal planes of MCMC.py:242,
        nugget_fields[r] the rng.normal(0, sqrt(nug)) plane of MCMC.py:251 (or None).  Returns the masked fields, a list
        of (bh, bw) arrays."""
        size_idx = np.ascontiguousarray(size_idx, dtype=np.int32)
        n = size_idx.shape[0]
        sc = np.ascontiguousarray(rf_scalars, dtype=np.float64)
        if sc.shape != (n, 4):
            raise ValueError("rf_scalars must have shape (n, 4)")

        def pack(planes):
            out = np.zeros((n, self.field_stride))
            for r, a in enumerate(planes):
                a = np.asarray(a, dtype=np.float64)
                if a.shape != (int(self.bh[size_idx[r]]), int(self.bw[size_idx[r]])):
                    raise ValueError("noise plane shape does not match its block size")
                out[r, :a.size] = a.ravel()
            return self._f64(out)

        d_re, d_im = pack(noise_re), pack(noise_im)
        d_ng = pack(nugget_fields) if nugget_fields is not None else None
        d_si, d_sc = torch.as_tensor(size_idx).to(self.dev), self._f64(sc)
        out = torch.zeros((n, self.field_stride), dtype=torch.float64, device=self.dev)
        p = rf if isinstance(rf, RfParams) else self.rf_struct(rf)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_spectral_from_noise(self.h, n, _ptr(d_si), _ptr(d_sc), C.byref(p), _ptr(d_re), _ptr(d_im),
                                                         _ptr(d_ng), _ptr(out), self.field_stride, self._stream()))
            torch.cuda.synchronize(self.dev)
        h = out.cpu().numpy()
        return [h[r, :int(self.bh[size_idx[r]] * self.bw[size_idx[r]])].reshape(int(self.bh[size_idx[r]]), int(self.bw[size_idx[r]]))
                for r in range(n)]

    # ---- 'pcg64' draw mode: NumPy's generator streams on the device (gsm_draw_pcg64) ------------------------------------
    @staticmethod
    def pack_pcg64_states(gens):
        """[n, 6] uint64 from numpy Generators / bit_generator.state dicts: state lo, hi, inc lo, hi, has_uint32, uinteger."""
        out = np.zeros((len(gens), 6), dtype=np.uint64)
        m = (1 << 64) - 1
        for i, g in enumerate(gens):
            st = g if isinstance(g, dict) else g.bit_generator.state
            if st.get("bit_generator") != "PCG64":
                raise ValueError("the pcg64 draw mode needs numpy.random.Generator(PCG64) generators (numpy.random.default_rng)")
            s, inc = int(st["state"]["state"]), int(st["state"]["inc"])
            out[i] = [s & m, s >> 64, inc & m, inc >> 64, int(st["has_uint32"]), int(st["uinteger"])]
        return out

    @staticmethod
    def unpack_pcg64_states(words):
        """numpy bit_generator.state dicts from [n, 6] uint64 (assign to generator.bit_generator.state)."""
        return [{"bit_generator": "PCG64", "state": {"state": int(w[0]) | (int(w[1]) << 64), "inc": int(w[2]) | (int(w[3]) << 64)},
                 "has_uint32": int(w[4]), "uinteger": int(w[5])} for w in np.asarray(words, dtype=np.uint64)]

    def alloc_pcg64_buffers(self, n_steps, nugget):
        """Caller-owned outputs of draw_pcg64 for batches of n_steps (reused from batch to batch: only the bh * bw leading doubles of
        a record's planes are written and read)."""
        n = self.n_chains * n_steps
        shape = (self.n_chains, n_steps, self.field_stride)
        z = lambda: torch.zeros(shape, dtype=torch.float64, device=self.dev)
        return dict(size_idx=torch.empty(n, dtype=torch.int32, device=self.dev), centre=torch.empty(n * 2, dtype=torch.int32, device=self.dev),
                    u=torch.empty(n, dtype=torch.float64, device=self.dev), rf_scalars=torch.empty(n * 4, dtype=torch.float64, device=self.dev),
                    noise_re=z(), noise_im=z(), nugget=z() if nugget else None, n_steps=n_steps)

    def draw_pcg64(self, n_steps, rf, d_rf_state, d_chain_state, d_region_mask=None, buffers=None):
        """n_steps Metropolis steps' worth of the reference's NumPy draws for every chain, on the device (asynchronous on the
        current stream).  d_rf_state / d_chain_state: int64 device tensors [n_chains, 6] (pack_pcg64_states), advanced in place.
        Returns device tensors size_idx [n, s], centre [n, s, 2], u [n, s], rf_scalars [n, s, 4], noise_re / noise_im
        [n, s, field_stride] and nugget (or None); `buffers` (alloc_pcg64_buffers of the same n_steps) are used if given."""
        p = rf if isinstance(rf, RfParams) else self.rf_struct(rf)
        b = buffers if (buffers is not None and buffers["n_steps"] == n_steps) else self.alloc_pcg64_buffers(n_steps, p.nugget_max > 0.0)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_draw_pcg64(self.h, int(n_steps), C.byref(p), _ptr(d_rf_state), _ptr(d_chain_state),
                                                _ptr(d_region_mask), _ptr(b["size_idx"]), _ptr(b["centre"]), _ptr(b["u"]), _ptr(b["rf_scalars"]),
                                                _ptr(b["noise_re"]), _ptr(b["noise_im"]), _ptr(b["nugget"]), self.field_stride, self._stream()))
        return dict(size_idx=b["size_idx"].view(self.n_chains, n_steps), centre=b["centre"].view(self.n_chains, n_steps, 2),
                    u=b["u"].view(self.n_chains, n_steps), rf_scalars=b["rf_scalars"].view(self.n_chains, n_steps, 4),
                    noise_re=b["noise_re"], noise_im=b["noise_im"], nugget=b["nugget"])

    def enable_timing(self, on=True):
        self._check(self.lib.gsm_enable_timing(self.h, 1 if on else 0))

    def last_timing(self):
        a, b = C.c_double(), C.c_double()
        na, nb = C.c_int32(), C.c_int32()
        self._check(self.lib.gsm_last_timing(self.h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return dict(step_ms=a.value, step_launches=na.value, proposal_ms=b.value, proposal_launches=nb.value)

    def set_fused(self, on: bool):
        """Philox mode, spectral generator: fused chain kernel (default) or the two-kernel pipeline (same results)."""
        self._check(self.lib.gsm_set_fused(self.h, int(bool(on))))

    def last_run_fused(self) -> int:
        return int(self.lib.gsm_last_run_fused(self.h))

    def strip_active(self) -> bool:
        """True when this handle's tables go to the strip kernels (two chains per CU), False for the flux-tile kernels."""
        rc = int(self.lib.gsm_strip_active(self.h))
        if rc < 0:
            self._check(rc)
        return bool(rc)

    def run_philox(self, n_steps, step0, seeds, rf, batch=8, out=None, to_host=True):
        """n_steps Metropolis steps for every chain with on-device proposals.
        Returns (loss, accept, blocks): (n_chains, n_steps), (n_chains, n_steps), (n_chains, n_steps, 4)."""
        if self.beds is None:
            raise RuntimeError("set_state() first")
        d_seeds = self._seeds(seeds)
        if out is None:
            loss = torch.empty((self.n_chains, n_steps), dtype=torch.float64, device=self.dev)
            acc = torch.empty((self.n_chains, n_steps), dtype=torch.uint8, device=self.dev)
            blocks = torch.empty((self.n_chains, n_steps, 4), dtype=torch.int32, device=self.dev)
        else:
            loss, acc, blocks = out
        p = rf if isinstance(rf, RfParams) else self.rf_struct(rf)
        with torch.cuda.device(self.dev):
            self._check(self.lib.gsm_run_philox(self.h, int(n_steps), int(step0), int(batch), _ptr(d_seeds), C.byref(p),
                                                _ptr(self.beds), _ptr(self.energy), _ptr(self.resampled), _ptr(self.loss_sum),
                                                _ptr(loss), _ptr(acc), _ptr(blocks), self._stream()))
        if to_host:
            return loss.cpu().numpy(), acc.cpu().numpy(), blocks.cpu().numpy()
        return loss, acc, blocks
