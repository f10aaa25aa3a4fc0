"""-m gpu: the strip kernels (chain_strip_kernel.hip: 512-thread workgroups, two chains per CU) against the flux-tile kernels
(chain_fused_kernel / step_flux_kernel, GSM_STRIP=0) on the same proposals.  Per-cell arithmetic is the same, operation for
operation (reference gstatsMCMC/MCMC.py:1279-1360, Topography.py:592-600): beds, energies, accept masks, blocks and resampled
counts must be identical; the loss differs by the order of the window sums only (tolerance 1e-12 relative, stated here).
The proposal fields of both runs are made equal with GSM_SPLIT2=0 (handles on the strip kernels otherwise split stage 2 of the
inverse DFT by the parity of kx, which the one-slot-per-wave flux-tile fused kernel cannot: last-bit differences in the fields)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import mcmc_oracle as orc
from gpu_common import make_engine

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

_CHILD = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/oracle'); sys.path.insert(0, {root!r} + '/tests')
import mcmc_oracle as orc
from gpu_common import make_engine
H, n, state = {H}, {n}, {state!r}
eng, prob, cfg, pairs, masks, rfp = make_engine(H, 3, state_dtype=state)
rfp = orc.standard_rf_params(); rfp.resolution = prob["resolution"]
eng.set_centres(np.ones_like(cfg.region_mask))
beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(3)])
eng.set_state(beds0)
loss, acc, blk = eng.run_philox(n, 5, [11, 12, 13], rfp, batch=n)
np.savez({out!r}, strip=int(eng.strip_active()), loss=loss, acc=acc, blk=blk, beds=eng.beds.cpu().numpy(),
         energy=eng.energy.cpu().numpy(), res=eng.resampled.cpu().numpy())
"""


def _run_child(tmp_path, H, n, state, strip):
    out = str(tmp_path / f"r_{H}_{state}_{strip}.npz")
    env = dict(os.environ, GSM_STRIP=str(strip), GSM_SPLIT2="0")     # equal fields for both families: the flux-tile fused kernel has no split stage 2
    r = subprocess.run([sys.executable, "-c", _CHILD.format(root=str(ROOT), H=H, n=n, state=state, out=out)], env=env,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.load(out)


@pytest.mark.parametrize("H,n,state", [(64, 80, "f64"), (256, 60, "f64"), (64, 40, "f32")])
def test_strip_kernels_equal_flux_tile_kernels(tmp_path, H, n, state):
    """GSM_STRIP is read once per process: one child per setting.  256: blocks 50-80 (16-, 32- and 64-lane strips, windows
    clipped on every side of the grid since centres may lie anywhere)."""
    a = _run_child(tmp_path, H, n, state, 1)
    b = _run_child(tmp_path, H, n, state, 0)
    assert int(a["strip"]) == 1 and int(b["strip"]) == 0          # the paths under test really ran
    for k in ("acc", "blk", "beds", "energy", "res"):
        assert np.array_equal(a[k], b[k]), k
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=1e-12)
    assert 0.2 < a["acc"].mean() < 1.0


def test_standard_tables_run_the_strip_kernels():
    for H in (64, 256):
        eng, *_ = make_engine(H, 2)
        assert eng.strip_active()
        eng.close()


_CHILD_WIDE = r"""
import sys, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {root!r} + '/oracle'); sys.path.insert(0, {root!r} + '/tests')
import mcmc_oracle as orc
from mcmc_gpu_amd.engine import GsmEngine
H, W, n = {H}, {W}, {n}
prob = orc.synthetic_problem(H, W)
res = prob["resolution"]
pairs = orc.block_pairs({bw0}, {bw1}, {bh0}, {bh1})          # (bw, bh) columns, MCMC.py:576-579
lp = [2, 0, 6, 1]
masks = orc.edge_masks(pairs, lp, 49900.0, res)
w = orc.crf_weight(prob["xx"], prob["yy"], prob["data_mask"], lp, 49900.0)
ones = np.full(prob["region_mask"].shape, 1)
eng = GsmEngine(H, W, 3)
eng.set_static(prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], w, ones, ones, res, 5.0)
eng.set_blocks(pairs, masks)
eng.set_centres(ones)                                          # whole-map updates: windows clipped on every side
rfp = orc.standard_rf_params(); rfp.resolution = res
eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(3)]))
loss, acc, blk = eng.run_philox(n, 3, [21, 22, 23], rfp, batch=n)
np.savez({out!r}, strip=int(eng.strip_active()), loss=loss, acc=acc, blk=blk, beds=eng.beds.cpu().numpy(),
         energy=eng.energy.cpu().numpy(), res=eng.resampled.cpu().numpy())
"""


def test_strip_kernels_equal_flux_tile_kernels_wide_blocks_on_an_oblong_grid(tmp_path):
    """Blocks 20 - 110 cells wide and 20 - 50 tall on a 128 x 192 grid, whole-map updates: every strip decomposition up to two
    64-lane column groups (strip::config: 64-, 16-, 32- and 2 x 64-lane strips), H != W, most windows clipped."""
    outs = {}
    for strip, split2 in ((1, "0"), (0, "0"), (1, "1")):
        out = str(tmp_path / f"wide_{strip}_{split2}.npz")
        code = _CHILD_WIDE.format(root=str(ROOT), H=128, W=192, n=80, bw0=20, bw1=110, bh0=20, bh1=50, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, GSM_STRIP=str(strip), GSM_SPLIT2=split2), capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        outs[(strip, split2)] = np.load(out)
    a, b, c = outs[(1, "0")], outs[(0, "0")], outs[(1, "1")]
    # stage 2 split by the parity of kx (the default on strip handles) against the direct sums: the same chain to rounding
    assert np.array_equal(c["acc"], a["acc"]) and np.array_equal(c["blk"], a["blk"])
    np.testing.assert_allclose(c["beds"], a["beds"], rtol=0, atol=1e-9)
    np.testing.assert_allclose(c["loss"], a["loss"], rtol=1e-10)
    assert int(a["strip"]) == 1 and int(b["strip"]) == 0
    widths = set(int(x) for x in a["blk"][..., 3].ravel())
    assert max(widths) > 90 and min(widths) <= 62                 # the two-group 64-lane strips and the one-group ones both ran
    for k in ("acc", "blk", "beds", "energy", "res"):
        assert np.array_equal(a[k], b[k]), k
    np.testing.assert_allclose(a["loss"], b["loss"], rtol=1e-12)
