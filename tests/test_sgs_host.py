"""CPU suite: host logic of the small-scale chain mirror (mcmc_gpu_amd/sgs.py) -- the per-iteration draws consume the
chain's NumPy generator exactly as the reference's chain_sgs.run / sgs do (golden F10 holds the reference's final generator
state, block list and the order of the simulated cells), the covariance lag table equals the reference's pairwise
covariance, and the C-ABI entries exist."""
import json

import numpy as np
import pytest

import sgs_common as sc
from mcmc_gpu_amd import _lib, sgs


@pytest.mark.parametrize("tag", ["a", "b"])
def test_host_draws_consume_the_generator_like_the_reference(tag):
    g, prob, cfg, ch = sc.setup(tag)
    n_iter = int(g["n_iter"])
    nst = ch.nst_trans
    cond_c = prob["cond_bed"] - prob["trend"] if ch.detrend_map else prob["cond_bed"]
    z_cond = nst.transform(cond_c.reshape(-1, 1)).reshape(cond_c.shape) if nst is not None else cond_c
    is_data = ~np.isnan(z_cond)
    rng = ch.rng
    blocks, order = [], []
    for _ in range(n_iter):
        blk, win, inds, z, u = ch._draw_iteration(rng, is_data)
        blocks.append(blk)
        need = ~is_data[inds[:, 0], inds[:, 1]]
        assert np.all(z[~need] == 0.0) and 0.0 <= u < 1.0
        order.extend(map(tuple, inds[need]))
        r0, r1, c0, c1 = win
        assert sorted(map(tuple, inds)) == [(i, j) for i in range(r0, r1) for j in range(c0, c1)]
    assert rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))          # same number and kind of draws
    assert np.array_equal(np.array(blocks, dtype=float), g[f"{tag}_blocks"])
    assert len(order) == int(g[f"{tag}_n_sim"])
    assert np.array_equal(np.array(order[:40], dtype=float), g[f"{tag}_trace_head"][:, :2])   # simulation order of the cells


def test_lag_table_equals_pairwise_covariance():
    """lag_cov_table(di, dj) == make_sigma's value for two cells that far apart (oracle restatement pinned by golden F6),
    anisotropic + rotated, all four models."""
    import mcmc_oracle as orc
    dx, dy, hw = 500.0, -400.0, 3
    for vt, extra in (("Exponential", {}), ("Gaussian", {}), ("Spherical", {}), ("Matern", {"s": 1.5})):
        v = dict(azimuth=30.0, nugget=0.1, major_range=4000.0, minor_range=2500.0, sill=1.3, vtype=vt, **extra)
        tab = sgs.lag_cov_table(v, hw, dx, dy)
        assert tab.shape == (4 * hw + 1, 4 * hw + 1)
        pts = [(0, 0), (2, 5), (6, 1), (3, 3)]
        coord = np.array([[j * dx, i * dy] for i, j in pts])
        sig = orc.cov_matrix(coord, v)
        for a, (ia, ja) in enumerate(pts):
            for b, (ib, jb) in enumerate(pts):
                np.testing.assert_allclose(tab[ia - ib + 2 * hw, ja - jb + 2 * hw], sig[a, b], rtol=1e-12, atol=1e-15)


def test_sgs_entries_are_exported_and_setters_validate():
    names = _lib.declared_symbols()
    assert {"gsm_sgs_blocks", "gsm_sgs_loss", "gsm_sgs_commit"} <= set(names)
    lib = _lib.load()
    for n in ("gsm_sgs_blocks", "gsm_sgs_loss", "gsm_sgs_commit"):
        assert hasattr(lib, n)
    g, prob, cfg, ch = sc.setup("a")
    with pytest.raises(ValueError):
        ch.set_variogram("Matern", 1000.0, 1.0, 0.0)             # smoothness missing (MCMC.py:1521-1523)
    with pytest.raises(ValueError):
        ch.set_variogram("Cubic", 1000.0, 1.0, 0.0)
    with pytest.raises(ValueError):
        ch.set_trend(None, detrend_map=True)
    with pytest.raises(ValueError):
        ch.set_update_region(True, np.ones((3, 3)))
    p = dict(ch.__dict__, rng_seed=5, initial_bed=prob["bed"])
    ch2 = sgs.init_msc_chain_by_instance(p)
    assert ch2.vario_param == ch.vario_param and ch2.sgs_param == ch.sgs_param and ch2.rng_seed == 5
    import torch
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError):
            ch.run(3, only_save_last_bed=True, plot=False, progress_bar=None)      # no CPU fallback


def test_sgs_function_argument_errors_are_the_references():
    """mcmc_gpu_amd.sgs.sgs mirrors MCMC.sgs (MCMC.py:91): the reference's _sanity_checks / get_random_generator errors
    (interpolate.py:282-330, utilities.py:62-69) are raised before anything touches the device; what the device path does not
    cover is a NotImplementedError, never a silent CPU computation."""
    xx, yy, grid, vario, kw, seed = sc.f12_case("sk")
    with pytest.raises(ValueError):
        sgs.sgs(xx, yy, grid[0], vario, **kw)                                  # grid must be 2D
    with pytest.raises(ValueError):
        sgs.sgs(xx, yy[:-1], grid, vario, **kw)                                # same shapes
    with pytest.raises(ValueError):
        sgs.sgs(xx, yy, grid, {k: v for k, v in vario.items() if k != "sill"}, **kw)
    with pytest.raises(ValueError):
        sgs.sgs(xx, yy, grid, vario, **dict(kw, ktype="uk"))
    with pytest.raises(ValueError):
        sgs.sgs(xx, yy, grid, vario, seed="seed", **kw)
    with pytest.raises(NotImplementedError):
        sgs.sgs(xx, yy, grid, vario, stencil=np.ones((3, 3), bool), **kw)
    with pytest.raises(NotImplementedError):
        sgs.sgs(xx, yy, grid, vario, **dict(kw, num_points=64))
    with pytest.raises(NotImplementedError):
        sgs.sgs(xx, yy, np.full(grid.shape, np.nan), vario, seed=1, **kw)      # 2304 cells to simulate: not one block
    full = np.where(np.isnan(grid), 0.0, grid)
    assert np.array_equal(sgs.sgs(xx, yy, full, vario, seed=1, **kw), full)    # nothing to simulate: the grid comes back
