"""-m gpu: the small-scale chain at the reference DRIVER's own parameters -- set_sgs_param(48, 30e3) at 500 m spacing (search
half-width 60 cells), blocks 5-20, Matern variogram, QuantileTransformer(1000), detrending (smallScaleChain_multiprocessing.py:
489-556) -- against golden F11 (oracle/make_fixtures_r3.py, the imported reference), and the properties of the octant search that
hold whatever the sort order of equidistant candidates.

Tolerances.  Index work (accept masks, blocks, resampled counts, generator state, neighbour counts) must be identical.  Values:
the device solves the 49 x 49 kriging systems by Gauss-Jordan elimination, the reference by numpy.linalg.lstsq (SVD); with 48
neighbours a few cells apart and the Matern(1.23) model the systems have condition numbers around 1e6-1e8, so weights agree to
~1e-9 relative and the simulated values, which feed on each other cell after cell and pass through the inverse normal-score
transform, to the tolerances asserted below."""
import json

import numpy as np
import pytest

import sgs_common as sc
import sgs_oracle as so
from test_oracle_sgs_golden import f11_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tag,mode", [("a", "replay"), ("t", "replay"), ("b", "replay"), ("c", "replay"), ("a", "pcg64"), ("b", "pcg64")])
def test_chain_sgs_gpu_equals_reference_at_driver_config(tag, mode):
    """mode 'replay': the chain's NumPy generator is consumed on the host; 'pcg64': the same stream is advanced on the device
    (gsm_sgs_draw_pcg64) -- same fixture, same assertions, the final generator state included."""
    g, prob, trend, nst, vp, sp, sigma, stable = f11_case(tag)
    ch = sc.driver_chain(prob, trend, nst, int(g[f"{tag}_seed"]), vp, sp, None, sigma)
    ch.set_rng_mode(mode)
    n_iter = int(g[f"{tag}_n_iter"])
    out = ch.run(n_iter, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert np.array_equal(out[6], g[f"{tag}_blocks"]), "blocks"
    assert np.array_equal(out[4], g[f"{tag}_steps"]), "accept mask differs from the reference"
    assert np.array_equal(out[5], g[f"{tag}_resampled"]), "resampled counts"
    assert ch.rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))
    np.testing.assert_allclose(out[3], g[f"{tag}_loss"], rtol=1e-8)
    np.testing.assert_allclose(out[0], g[f"{tag}_bed"], rtol=0, atol=1e-6)
    assert out[4].sum() >= 2
    if tag == "a":
        # for information: the unmodified reference on the fixture container's CPU (unstable argsort among equidistant
        # candidates, about every fifth cell) stays within a metre and took the same decisions
        assert np.array_equal(out[4], g["a_native_steps"])
        assert np.abs(out[0] - g["a_native_bed"]).max() < 2.0


@pytest.mark.parametrize("mode", ["replay", "pcg64"])
def test_chain_sgs_gpu_equals_reference_over_a_deep_driver_run(mode):
    """Golden F13 (oracle/make_fixtures_r4.py): the UNMODIFIED reference at the driver's parameters on the tie-free geometry with
    sigma_mc = 30 -- 160 iterations, 68 of them accepted, so the values the device simulates (Gauss-Jordan where the reference calls
    lstsq) pass through the inverse transform and the commit and condition later iterations 68 times.  Index work identical (accept
    mask, blocks, resampled counts, generator state); values to the tolerances of the F11 test, which hold over this depth too:
    loss 1e-8 relative, bed 1e-6 m."""
    from test_oracle_sgs_golden import f13_case
    g, prob, trend, nst, sigma = f13_case()
    ch = sc.driver_chain(prob, trend, nst, int(g["d_seed"]), None, None, None, sigma)
    ch.set_rng_mode(mode)
    out = ch.run(int(g["d_n_iter"]), only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert np.array_equal(out[6], g["d_blocks"]), "blocks"
    assert np.array_equal(out[4], g["d_steps"]), "accept mask differs from the reference"
    assert np.array_equal(out[5], g["d_resampled"]), "resampled counts"
    assert ch.rng.bit_generator.state == json.loads(str(g["d_rng_state"]))
    np.testing.assert_allclose(out[3], g["d_loss"], rtol=1e-8)
    np.testing.assert_allclose(out[0], g["d_bed"], rtol=0, atol=1e-6)
    assert out[4].sum() >= 30


def _one_iteration_inputs(ch, prob, rngs, cond_is_data):
    n = len(rngs)
    wins = np.empty((n, 4), np.int32); offs = np.zeros(n + 1, np.int32)
    cells, zs = [], []
    for c in range(n):
        _, win, inds, z, _ = ch._draw_iteration(rngs[c], cond_is_data)
        wins[c] = win; cells.append(inds); zs.append(z); offs[c + 1] = offs[c] + inds.shape[0]
    return wins, offs, np.ascontiguousarray(np.concatenate(cells)), np.concatenate(zs)


@pytest.mark.parametrize("npts,rad,blocks,n_iter", [(16, 4000.0, (3, 8, 3, 8), 2000), (48, 30e3, (5, 20, 5, 20), 120)])
def test_device_neighbour_sets_are_valid_octant_searches(npts, rad, blocks, n_iter):
    """The property that holds whatever breaks ties (neighbors.py:50-60): for every simulated cell of a long run (8 chains, square
    500 m grid, so equidistant candidates abound) the device's neighbour set holds, per 45-degree sector, min(k8, available)
    cells, all with a value when the cell is visited and closer than the radius, and no unselected candidate of the sector is
    closer than a selected one.  Uses gsm_sgs_blocks' neighbour trace."""
    import torch
    from mcmc_gpu_amd import sgs
    from mcmc_gpu_amd.engine import GsmEngine, _ptr
    H, n = 32, 8
    prob = sc.problem(H)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["resolution"])
    ch.set_update_region(True, prob["region_mask"]); ch.set_block_sizes(*blocks)
    sill = float(np.var(prob["bed"]))
    vario = dict(azimuth=0, nugget=0.0, major_range=6000.0, minor_range=6000.0, sill=sill, vtype="Exponential")
    xs, ys, dx, dy = sgs._axes(prob["xx"], prob["yy"])
    hw = int(np.ceil(rad / abs(dx)))
    k8 = npts // 8
    cond = prob["cond_bed"]
    is_data = ~np.isnan(cond)
    eng = GsmEngine(H, H, n)
    dev = eng.dev
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    mi, mj = sgs.lag_extents(hw, H, H)
    d_lag, d_xs, d_ys, d_zc = f64(sgs.lag_cov_table(vario, hw, dx, dy, mi, mj)), f64(xs), f64(ys), f64(cond)
    beds = np.stack([prob["bed"] + np.random.default_rng(70 + c).normal(0, 3, (H, H)) for c in range(n)])
    rngs = [np.random.default_rng(500 + c) for c in range(n)]
    ii, jj = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
    max_cells = (blocks[1] - 1) * (blocks[3] - 1)
    checked = tied_cut = 0
    try:
        for it in range(n_iter):
            wins, offs, cells, z = _one_iteration_inputs(ch, prob, rngs, is_data)
            tot = int(offs[-1])
            grid = f64(beds)
            nbr = torch.full((tot, 48), -2, dtype=torch.int32, device=dev)
            tr = torch.zeros((tot, 3), dtype=torch.float64, device=dev)
            keep = (torch.as_tensor(wins).to(dev), torch.as_tensor(offs).to(dev), torch.as_tensor(cells).to(dev), f64(z))
            eng._check(eng.lib.gsm_sgs_blocks(eng.h, _ptr(grid), _ptr(d_zc), _ptr(keep[0]), _ptr(d_xs), _ptr(d_ys), _ptr(d_lag), mi, mj, hw,
                                              float(rad), npts, sill, _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), max_cells, _ptr(tr),
                                              _ptr(nbr), eng._stream()))
            nb, t = nbr.cpu().numpy(), tr.cpu().numpy()
            for c in range(n):
                r0, r1, c0, c1 = wins[c]
                has = np.ones((H, H), dtype=bool)
                has[r0:r1, c0:c1] = is_data[r0:r1, c0:c1]
                for k in range(offs[c], offs[c + 1]):
                    i0, j0 = cells[k]
                    if t[k, 0] < 0:
                        assert is_data[i0, j0]
                        continue
                    assert not has[i0, j0]
                    sel = nb[k][nb[k] >= 0]
                    assert sel.size == int(t[k, 0]) and np.unique(sel).size == sel.size
                    ilo, ihi, jlo, jhi = max(0, i0 - hw), min(H, i0 + hw + 1), max(0, j0 - hw), min(H, j0 + hw + 1)
                    w_has = has[ilo:ihi, jlo:jhi]
                    ddx = prob["xx"][i0, j0] - prob["xx"][ilo:ihi, jlo:jhi]; ddy = prob["yy"][i0, j0] - prob["yy"][ilo:ihi, jlo:jhi]
                    d = np.sqrt(ddx ** 2 + ddy ** 2)
                    ang = np.arctan2(ddy, ddx)
                    flat = (ii[ilo:ihi, jlo:jhi] * H + jj[ilo:ihi, jlo:jhi])
                    chosen = np.isin(flat, sel)
                    assert chosen.sum() == sel.size, "a neighbour outside the search window"
                    assert (w_has & (d < rad))[chosen].all(), "a neighbour without a value or beyond the radius"
                    pos = 0
                    for b in range(-4, 4):
                        m = (d < rad) & (ang > b / 4 * np.pi) & (ang <= (b + 1) / 4 * np.pi) & w_has
                        want = min(k8, int(m.sum()))
                        got = chosen & m
                        assert got.sum() == want, "sector count"
                        seg = sel[pos:pos + want]; pos += want                       # sectors concatenated in angle order ...
                        assert np.isin(seg, flat[got]).all()
                        dseg = np.array([d[flat == s_][0] for s_ in seg])
                        assert (np.diff(dseg) >= 0).all(), "not nearest first"      # ... nearest first
                        if want and (m & ~chosen).any():
                            assert d[got].max() <= d[m & ~chosen].min(), "an unselected candidate is closer than a selected one"
                            tied_cut += int(d[got].max() == d[m & ~chosen].min())
                    checked += 1
                    has[i0, j0] = True
    finally:
        eng.close()
    assert checked > 1000 and tied_cut > 0          # the run did meet cuts between equidistant candidates


@pytest.mark.parametrize("vtype,vrange", [("Gaussian", 700.0), ("Gaussian", 1000.0), ("Gaussian", 1500.0), ("Gaussian", 2200.0),
                                          ("Gaussian", 3000.0), ("Gaussian", 9000.0), ("Spherical", 9000.0), ("Matern", 9000.0)])
def test_kriging_solve_against_lstsq_over_condition_numbers(vtype, vrange):
    """One block (48 neighbours, 30 km) per variogram: device estimates vs the oracle's numpy.linalg.lstsq (_krige.py:36-38).
    Gaussian model with a growing range at 500 m spacing (the reference's `nugget` only scales the covariance, it does not
    regularise the diagonal: covariance.py:8-10): the condition number of the 49 x 49 systems climbs from ~1e1 to beyond 1e16.
    The device either agrees with lstsq to cond * 1e-15 (stated per case) or refuses with 'singular kriging system' where a
    pivot falls below eps * N * max|diag| -- the regime where lstsq(rcond=None) starts to truncate singular values and the two
    methods solve different problems.  Recorded range: include/gsm.h (gsm_sgs_blocks)."""
    import torch
    from mcmc_gpu_amd import sgs
    from mcmc_gpu_amd._lib import GsmError
    from mcmc_gpu_amd.engine import GsmEngine, _ptr
    H = 48
    prob = sc.driver_problem(H, dy=sc.TIE_FREE_DY)
    sill = 30.0 if vtype != "Spherical" else 40.0
    vp = [0, 0.0, vrange, vrange, sill, vtype, 1.2259 if vtype == "Matern" else None]
    cfg = sc.driver_cfg(prob, None, None, vp, [48, 30e3, False, 0], (9, 12, 9, 12), 40.0)
    ch = sc.driver_chain(prob, None, None, 41, vp, [48, 30e3, False, 0], (9, 12, 9, 12), 40.0)
    rng_o = np.random.default_rng(41)
    trace = []
    conds = []
    orig = np.linalg.lstsq
    def spy(a, b, rcond=None):
        conds.append(np.linalg.cond(a))
        return orig(a, b, rcond=rcond)
    np.linalg.lstsq = spy
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            so.run_chain_sgs(cfg, prob["bed"], 1, rng_o, trace=trace)
    finally:
        np.linalg.lstsq = orig
    exp = np.array(trace)
    is_data = ~np.isnan(prob["cond_bed"])
    _, win, inds, z, _ = ch._draw_iteration(np.random.default_rng(41), is_data)
    xs, ys, dx, dy = sgs._axes(prob["xx"], prob["yy"])
    hw = int(np.ceil(30e3 / abs(dx)))
    vario = ch._vario()
    eng = GsmEngine(H, H, 1)
    dev = eng.dev
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    mi, mj = sgs.lag_extents(hw, H, H)
    grid = f64(prob["bed"][None])
    tr = torch.zeros((inds.shape[0], 3), dtype=torch.float64, device=dev)
    keep = (f64(prob["cond_bed"]), torch.as_tensor(np.array([win], np.int32)).to(dev), f64(xs), f64(ys),
            f64(sgs.lag_cov_table(vario, hw, dx, dy, mi, mj)), torch.as_tensor(np.array([0, inds.shape[0]], np.int32)).to(dev),
            torch.as_tensor(inds).to(dev), f64(z))
    cmax = float(np.max(conds))
    try:
        eng._check(eng.lib.gsm_sgs_blocks(eng.h, _ptr(grid), _ptr(keep[0]), _ptr(keep[1]), _ptr(keep[2]), _ptr(keep[3]), _ptr(keep[4]), mi, mj,
                                          hw, 30e3, 48, float(vario["sill"]), _ptr(keep[5]), _ptr(keep[6]), _ptr(keep[7]), int(inds.shape[0]),
                                          _ptr(tr), None, eng._stream()))
    except GsmError as e:
        assert "singular kriging system" in str(e)
        assert cmax > 1e12, f"refused a well-conditioned system (cond {cmax:.2e})"
        print(f"\n{vtype} range {vrange:g} m: cond up to {cmax:.2e}: refused (singular kriging system)")
        return
    finally:
        eng.close()
    t = tr.cpu().numpy()
    sim = t[t[:, 0] >= 0]
    assert np.array_equal(sim[:, 0], exp[:, 2])
    scale = np.abs(exp[:, 3]).max()
    err = np.abs(sim[:, 1] - exp[:, 3]).max() / scale
    print(f"\n{vtype} range {vrange:g} m: cond up to {cmax:.2e}: max |est - lstsq| / scale = {err:.2e}")
    assert err < max(1e-11, cmax * 2e-15), (err, cmax)
