"""TEST INFRASTRUCTURE ONLY (imported by tests/): the draws of the small-scale chain's Philox mode, restated on the CPU.

In Philox mode (mcmc_gpu_amd/sgs.py: chain_sgs_gpu.set_rng_mode('philox'), include/gsm.h: gsm_sgs_draw_philox) the device makes
the draws chain_sgs.run takes from chain.rng (MCMC.py:1750-1765 block centre and sizes, :128 the visiting order, :165 one normal
per simulated cell, :1797 the accept uniform) from Philox4x32-10 counters (draw index, stream 4, iteration), key = the chain's
seed.  `PhiloxSgsRng` serves exactly those values through the few numpy.random.Generator methods oracle/sgs_oracle.py calls, in
the order it calls them, so that sgs_oracle.run_chain_sgs(cfg, bed, n_iter, PhiloxSgsRng(...)) is the CPU restatement of a
Philox-mode chain (the chain's arithmetic itself is pinned to the reference by golden F10 through sgs_oracle).

  draws 0..63        centre attempts: row = (x * H) >> 32, col = (y * W) >> 32
  draw  64           block sizes min + ((x|y) * (max - min)) >> 32, accept uniform u01(z, w)
  draws 128 + p      visiting key (x) of window cell p (row-major); cells are visited in ascending (key, p)
  draws 2048 + p/2   standard normal of cell p: first / second Box-Muller value for even / odd p
"""
import numpy as np

import philox_oracle as po

STREAM_SGS = 4


class PhiloxSgsRng:
    def __init__(self, seed, is_data, first_iteration=0):
        self.seed = int(seed)
        self.is_data = np.asarray(is_data, dtype=bool)
        self.it = int(first_iteration)
        self._reset()

    def _reset(self):
        self.n_centre_calls = 0
        self.n_size_calls = 0
        self.visit = None          # (cells in visiting order [(i, j)], their row-major index p)
        self.k = 0

    def _draw(self, idx):
        return po.draw(self.seed, self.it, STREAM_SGS, np.asarray([idx], dtype=np.uint32))[0]

    def integers(self, low=0, high=None, size=1):
        if low == 0:               # centre attempts: alternately a row (high = H) and a column (high = W)
            att, comp = divmod(self.n_centre_calls, 2)
            self.n_centre_calls += 1
            if att >= 64:
                raise RuntimeError("no block centre inside the region after 64 attempts")
            v = (int(self._draw(att)[comp]) * int(high)) >> 32
        else:                      # block sizes: numpy's integers(low, high) excludes high
            comp = self.n_size_calls
            self.n_size_calls += 1
            v = int(low) + ((int(self._draw(64)[comp]) * (int(high) - int(low))) >> 32)
        return np.array([v])

    def shuffle(self, inds):
        n = inds.shape[0]
        keys = po.draw(self.seed, self.it, STREAM_SGS, (128 + np.arange(n)).astype(np.uint32))[:, 0].astype(np.uint64)
        order = np.lexsort((np.arange(n), keys))          # ascending (key, p)
        cells = inds[order].copy()
        inds[:] = cells
        self.visit = (cells, order)
        self.k = 0

    def normal(self, loc=0.0, scale=1.0, size=None):
        cells, order = self.visit
        while self.is_data[cells[self.k, 0], cells[self.k, 1]]:     # conditioning cells of the block draw nothing
            self.k += 1
        p = int(order[self.k])
        self.k += 1
        g1, g2 = po.normals2(self.seed, self.it, STREAM_SGS, np.asarray([2048 + (p >> 1)], dtype=np.uint32))
        z = float(g2[0] if (p & 1) else g1[0])
        return np.array([loc + scale * z])

    def random(self):
        d = self._draw(64)
        u = float(po.u01(d[2], d[3]))
        self.it += 1
        self._reset()
        return u
