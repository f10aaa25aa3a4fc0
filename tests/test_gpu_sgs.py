"""-m gpu: the small-scale chain on the device against the reference through golden F10 and the oracle.

Tolerance.  The device solves every ordinary-kriging system by pivoted elimination, the reference by numpy.linalg.lstsq
(SVD): kriging estimates agree to ~1e-10 relative and the simulated beds, which feed on each other cell after cell, to
1e-7 m on beds of order 1e3 m (asserted: 1e-7 absolute); losses to 1e-9 relative.  Accept masks, blocks and resampled
counts must be identical.  Both F10 variants: (a) exponential variogram, no transform / trend; (b) Matern variogram,
detrended, normal-score transform (scikit-learn's QuantileTransformer: on the device, gsm_qt_transform)."""
import json

import numpy as np
import pytest

import sgs_common as sc
import sgs_oracle as so

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _device_tie_rule():
    """Where the oracle runs beside the device it uses the device's rule for equidistant candidates (ascending (distance, row,
    column) = numpy.argsort(kind='stable') of the reference's masked array); the reference's own default sort is unstable there
    (DESIGN.md section 8).  Fixture comparisons (F10, F11) do not run the oracle."""
    so.STABLE_TIES = True
    yield
    so.STABLE_TIES = False


def test_sgs_one_block_equals_oracle_trace():
    """gsm_sgs_blocks alone on the first block of F10a: neighbour counts exact, estimates / variances to 1e-9 relative."""
    import ctypes as C
    import torch
    from mcmc_gpu_amd import sgs
    from mcmc_gpu_amd.engine import GsmEngine, _ptr
    g, prob, cfg, ch = sc.setup("a")
    H = int(g["H"])
    rng = np.random.default_rng(seed=int(g["a_seed"]))
    trace = []
    so.run_chain_sgs(cfg, prob["bed"], 1, rng, trace=trace)           # the oracle's first iteration
    is_data = ~np.isnan(prob["cond_bed"])
    blk, win, inds, z, u = ch._draw_iteration(np.random.default_rng(seed=int(g["a_seed"])), is_data)
    eng = GsmEngine(H, H, 1)
    dev = eng.dev
    f64 = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(dev)
    xs, ys, dx, dy = sgs._axes(prob["xx"], prob["yy"])
    hw = int(np.ceil(float(g["radius"]) / abs(dx)))
    vario = ch._vario()
    grid = f64(prob["bed"][None])
    tr = torch.zeros((inds.shape[0], 3), dtype=torch.float64, device=dev)
    d = dict(zc=f64(prob["cond_bed"]), win=torch.as_tensor(np.array([win], np.int32)).to(dev), xs=f64(xs), ys=f64(ys),
             lag=f64(sgs.lag_cov_table(vario, hw, dx, dy)), off=torch.as_tensor(np.array([0, inds.shape[0]], np.int32)).to(dev),
             cells=torch.as_tensor(inds).to(dev), z=f64(z))
    eng._check(eng.lib.gsm_sgs_blocks(eng.h, _ptr(grid), _ptr(d["zc"]), _ptr(d["win"]), _ptr(d["xs"]), _ptr(d["ys"]), _ptr(d["lag"]),
                                      2 * hw, 2 * hw, hw, float(g["radius"]), int(g["num_points"]), float(vario["sill"]), _ptr(d["off"]),
                                      _ptr(d["cells"]), _ptr(d["z"]), int(inds.shape[0]), _ptr(tr), None, eng._stream()))
    t = tr.cpu().numpy()
    sim = t[t[:, 0] >= 0]
    exp = np.array(trace)
    assert sim.shape[0] == exp.shape[0] > 0
    assert np.array_equal(sim[:, 0], exp[:, 2])                       # neighbour counts
    np.testing.assert_allclose(sim[:, 1], exp[:, 3], rtol=1e-9)       # kriging estimates
    np.testing.assert_allclose(sim[:, 2], exp[:, 4], rtol=1e-7, atol=1e-9 * float(vario["sill"]))
    # cells outside the block untouched, data cells of the block keep their value
    out = grid[0].cpu().numpy()
    r0, r1, c0, c1 = win
    mask = np.ones((H, H), bool); mask[r0:r1, c0:c1] = False
    assert np.array_equal(out[mask], prob["bed"][mask])
    blockd = is_data[r0:r1, c0:c1]
    assert np.array_equal(out[r0:r1, c0:c1][blockd], prob["cond_bed"][r0:r1, c0:c1][blockd])
    assert not np.isnan(out).any()
    eng.close()


@pytest.mark.parametrize("tag", ["a", "b"])
def test_chain_sgs_gpu_equals_reference_fixture(tag):
    g, prob, cfg, ch = sc.setup(tag)
    n_iter = int(g["n_iter"])
    out = ch.run(n_iter, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert len(out) == 7
    assert np.array_equal(out[4], g[f"{tag}_steps"]), "accept mask differs from the reference"
    assert np.array_equal(out[6], g[f"{tag}_blocks"])
    assert np.array_equal(out[5], g[f"{tag}_resampled"])
    np.testing.assert_allclose(out[3], g[f"{tag}_loss"], rtol=1e-9)
    np.testing.assert_allclose(out[0], g[f"{tag}_bed"], rtol=0, atol=1e-7)
    assert np.array_equal(out[1], out[3]) and not out[2].any()
    assert ch.rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))
    assert out[4].sum() >= 2


def test_small_scale_driver_at_the_reference_drivers_parameters(tmp_path):
    """BASELINE configs[0] literally: smallScaleChain, 64x64 grid, 4 chains, at the parameters of the reference's own driver
    (smallScaleChain_multiprocessing.py:470-560 through synthetic.sgs_template: set_sgs_param(48, 30e3) at 500 m = 60-cell search
    half-width, blocks 5-20, Matern, QuantileTransformer(1000), trend, sigma_mc 5) through smallScaleChain_mp / msc_run_wrapper
    (:211-399) with their text checkpoints: all four chains in one handle == each chain alone, file for file; a second segment
    resumes from the files; the device-drawn 'pcg64' mode writes the same files as the host-drawn replay mode."""
    from copy import deepcopy
    from mcmc_gpu_amd import driver, synthetic
    prob, ch = synthetic.sgs_template(64)
    assert ch.sgs_param[0] == 48 and ch.sgs_param[1] == 30e3 and ch.nst_trans is not None and ch.detrend_map
    seeds = [901, 902, 903, 904]
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(4)]
    n_it = 1000
    res = driver.smallScaleChain_mp(4, 3, ch, beds, seeds, 123456789, [n_it] * 4, output_path=str(tmp_path / "all"))
    base = tmp_path / "all" / "LargeScaleChain" / "123456" / "SmallScaleChain"
    keys = ("bed", "loss_mc", "loss_data", "loss", "steps", "resampled_times", "blocks_used")
    assert sorted(p.name for p in (base / "902").iterdir()) == sorted(f"{k}_1k.txt" for k in keys)
    cp = deepcopy(ch.__dict__); cp["rng_seed"] = 902; cp["initial_bed"] = beds[1]
    alone = driver.msc_run_wrapper(cp, dict(n_iter=n_it, only_save_last_bed=True, info_per_iter=10, plot=False, progress_bar=False,
                                            chain_id=1, tqdm_position=3, ssc_seed=902, lsc_seed=123456789,
                                            output_path=str(tmp_path / "alone")))
    for a, b in zip(res[1], alone):
        assert np.array_equal(a, b, equal_nan=True)                   # same device arithmetic: bit-equal
    for k in keys:
        assert np.array_equal(np.loadtxt(base / "902" / f"{k}_1k.txt"), np.loadtxt(tmp_path / "alone" / "902" / f"{k}_1k.txt")), k
    assert 0.005 < np.mean([r[4].mean() for r in res]) < 0.5         # sigma_mc 5 accepts a few per cent on the synthetic problem
    # the same chains with the generator advanced on the device: same files
    res_p = driver.smallScaleChain_mp(4, 3, ch, beds, seeds, 123456789, [n_it] * 4, output_path=str(tmp_path / "pcg"), mode='pcg64')
    for a, b in zip(res, res_p):
        assert np.array_equal(a[4], b[4]) and np.array_equal(a[6], b[6], equal_nan=True) and np.array_equal(a[0], b[0])
    # second segment resumes from the text files
    driver.smallScaleChain_mp(4, 3, ch, beds, seeds, 123456789, [n_it] * 4, output_path=str(tmp_path / "all"))
    assert sorted(p.name for p in (base / "903").iterdir()) == sorted(f"{k}_2k.txt" for k in keys)
    assert np.loadtxt(base / "903" / "loss_2k.txt").shape == (2000,)


def test_small_scale_driver_light_configuration(tmp_path):
    """The LIGHT test configuration of rounds 1-2 (16 neighbours within 4 km, blocks 3-8, no transformer) through the
    smallScaleChain_mp counterpart: all four chains in one handle == each chain alone through msc_run_wrapper (text files
    included), and a second segment resumes.  (The reference driver's own parameters: the test above.)"""
    from copy import deepcopy
    from mcmc_gpu_amd import driver, sgs
    prob = sc.problem(64)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           prob["cond_bed"], prob["data_mask"], np.ones((64, 64), dtype=int), prob["resolution"])
    ch.set_update_region(True, prob["region_mask"])
    ch.set_loss_type(sigma_mc=60.0, massConvInRegion=True)
    ch.set_normal_transformation(None, do_transform=False)
    ch.set_trend(None, detrend_map=False)
    ch.set_variogram("Exponential", 6000.0, float(np.var(prob["bed"])), 0.0, isotropic=True)
    ch.set_sgs_param(16, 4000.0)
    ch.set_block_sizes(3, 8, 3, 8)
    seeds = [901, 902, 903, 904]
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(4)]
    res = driver.smallScaleChain_mp(4, 3, ch, beds, seeds, 123456789, [1000] * 4, output_path=str(tmp_path / "all"))
    base = tmp_path / "all" / "LargeScaleChain" / "123456" / "SmallScaleChain"
    names = sorted(p.name for p in (base / "902").iterdir())
    assert names == sorted(f"{k}_1k.txt" for k in ("bed", "loss_mc", "loss_data", "loss", "steps", "resampled_times", "blocks_used"))
    cp = deepcopy(ch.__dict__); cp["rng_seed"] = 902; cp["initial_bed"] = beds[1]
    alone = driver.msc_run_wrapper(cp, dict(n_iter=1000, only_save_last_bed=True, info_per_iter=10, plot=False, progress_bar=False,
                                            chain_id=1, tqdm_position=3, ssc_seed=902, lsc_seed=123456789,
                                            output_path=str(tmp_path / "alone")))
    for a, b in zip(res[1], alone):
        assert np.array_equal(a, b, equal_nan=True)                   # same device arithmetic: bit-equal
    assert np.array_equal(np.loadtxt(base / "902" / "bed_1k.txt"), np.loadtxt(tmp_path / "alone" / "902" / "bed_1k.txt"))
    assert 0.05 < np.mean([r[4].mean() for r in res]) < 0.95
    # second segment resumes from the text files
    driver.smallScaleChain_mp(4, 3, ch, beds, seeds, 123456789, [1000] * 4, output_path=str(tmp_path / "all"))
    assert sorted(p.name for p in (base / "903").iterdir()) == sorted(
        f"{k}_2k.txt" for k in ("bed", "loss_mc", "loss_data", "loss", "steps", "resampled_times", "blocks_used"))
    assert np.loadtxt(base / "903" / "loss_2k.txt").shape == (2000,)


@pytest.mark.parametrize("vtype,smooth,aniso,npts,rad", [("Gaussian", None, True, 24, 3000.0), ("Spherical", None, True, 16, 4000.0),
                                                         ("Matern", 0.9, False, 8, 2500.0)])
def test_chain_sgs_gpu_equals_oracle_on_more_variograms(vtype, smooth, aniso, npts, rad):
    """Beyond the two F10 variants: anisotropic variograms with a rotation (the lag table's orientation), the spherical and
    Gaussian models, other neighbour counts / radii, whole-map update region (blocks clipped at the grid border), a
    non-square grid with descending y -- against the oracle (which golden F10 pins to the reference) on the same generator."""
    from mcmc_gpu_amd import sgs
    H, W, n_iter = 28, 36, 40
    prob = sc.orc.synthetic_problem(H, W, res=400.0)
    prob["yy"] = prob["yy"][::-1].copy()                                # north-up grid: y decreases with the row index
    data_mask = np.zeros((H, W), dtype=bool); data_mask[::5, :] = True; data_mask[:, ::7] = True
    cond = np.where(data_mask, prob["bed"], np.nan)
    region = np.ones((H, W), dtype=int)
    sill = float(np.var(prob["bed"]))
    grounded = np.ones((H, W), dtype=int)
    if aniso:
        vp = [35.0, 0.05 * sill, 5000.0, 2600.0, sill, vtype, smooth]
    else:
        vp = [0, 0.0, 4000.0, 4000.0, sill, vtype, smooth]
    cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], cond, data_mask,
                       grounded, region, prob["resolution"], 40.0, vp, [npts, rad, False, 0], 3, 8, 3, 8)
    rng_o = np.random.default_rng(seed=77)
    ref = so.run_chain_sgs(cfg, prob["bed"], n_iter, rng_o)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           cond, data_mask, grounded, prob["resolution"])
    ch.set_update_region(False)
    ch.set_loss_type(sigma_mc=40.0, massConvInRegion=True)
    ch.set_normal_transformation(None, do_transform=False)
    ch.set_trend(None, detrend_map=False)
    if aniso:
        ch.set_variogram(vtype, [5000.0, 2600.0], sill, 0.05 * sill, isotropic=False, vario_smoothness=smooth, vario_azimuth=35.0)
    else:
        ch.set_variogram(vtype, 4000.0, sill, 0.0, isotropic=True, vario_smoothness=smooth)
    ch.set_sgs_param(npts, rad)
    ch.set_block_sizes(3, 8, 3, 8)
    ch.set_random_generator(rng_seed=77)
    out = ch.run(n_iter, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert np.array_equal(out[4], ref[4]) and np.array_equal(out[6], ref[6]) and np.array_equal(out[5], ref[5])
    # The Gaussian model's kriging matrices are close to singular (condition numbers beyond 1e10): the reference's SVD solve
    # and the device's pivoted elimination then agree to ~1e-9 relative instead of 1e-12 -- 1e-5 m on these beds.
    np.testing.assert_allclose(out[3], ref[3], rtol=1e-7 if vtype == "Gaussian" else 1e-9)
    np.testing.assert_allclose(out[0], ref[0], rtol=0, atol=1e-5 if vtype == "Gaussian" else 1e-7)
    assert ch.rng.bit_generator.state == rng_o.bit_generator.state
    blk = ref[6]
    assert ((blk[:, 0] - blk[:, 2] / 2 < 0) | (blk[:, 1] - blk[:, 3] / 2 < 0) | (blk[:, 0] + blk[:, 2] / 2 > H) |
            (blk[:, 1] + blk[:, 3] / 2 > W)).any(), "no block was clipped at the border"


def test_batched_iterations_equal_one_by_one(monkeypatch):
    """run_many_sgs draws a batch of iterations ahead and decides on the device (gsm_sgs_finish: windowed loss, no transformer
    here); GSM_SGS_BATCH=1 is the iteration-by-iteration path with the full-grid loss and the decision on the host.  Same chains,
    same generators: identical beds, accept masks, counts; losses equal up to summation order."""
    from mcmc_gpu_amd import sgs
    H = 32
    prob = sc.problem(H)
    def make():
        ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                               prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["resolution"])
        ch.set_update_region(True, prob["region_mask"]); ch.set_loss_type(sigma_mc=60.0, massConvInRegion=True)
        ch.set_normal_transformation(None, do_transform=False); ch.set_trend(None, detrend_map=False)
        ch.set_variogram("Exponential", 6000.0, float(np.var(prob["bed"])), 0.0, isotropic=True)
        ch.set_sgs_param(16, 4000.0); ch.set_block_sizes(3, 8, 3, 8)
        return ch
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(3)]
    res = []
    for batch in ("32", "1"):
        monkeypatch.setenv("GSM_SGS_BATCH", batch)
        rngs = [np.random.default_rng(900 + i) for i in range(3)]
        out, rngs = sgs.run_many_sgs(make(), beds, rngs, 75)
        res.append((out, [r.bit_generator.state for r in rngs]))
    (a, sa), (b, sb) = res
    assert sa == sb
    for x, y in zip(a, b):
        for k in (0, 4, 5, 6):
            assert np.array_equal(x[k], y[k]), k
        np.testing.assert_allclose(x[3], y[3], rtol=1e-12)
    assert 0.05 < np.mean([x[4].mean() for x in a]) < 0.95


def test_device_quantile_transformer_equals_sklearn():
    """gsm_qt_transform (mcmc_gpu_amd/csrc/normal_score.h on the device) against scikit-learn's QuantileTransformer itself:
    forward (normal scores) within 1e-12, inverse within 1e-9 m on bed-like data, NaNs kept, out-of-range values clipped."""
    import ctypes as C
    import torch
    from sklearn.preprocessing import QuantileTransformer
    from mcmc_gpu_amd.engine import GsmEngine
    rng = np.random.default_rng(5)
    data = rng.normal(-300.0, 120.0, 6000)
    data[:1000] = np.round(data[:1000] / 40.0) * 40.0
    eng = GsmEngine(8, 8, 1, None)
    try:
        for nq in (1000, 64, 5000):
            qt = QuantileTransformer(n_quantiles=nq, output_distribution="normal", subsample=None).fit(data.reshape(-1, 1))
            q = torch.as_tensor(np.ascontiguousarray(qt.quantiles_[:, 0])).to(eng.dev); ref = torch.as_tensor(np.ascontiguousarray(qt.references_)).to(eng.dev)
            x = np.concatenate([rng.normal(-300.0, 220.0, 50000), data[:3000], [qt.quantiles_[0, 0] - 5, qt.quantiles_[-1, 0] + 5, np.nan]])
            d_x = torch.as_tensor(x).to(eng.dev); d_out = torch.empty_like(d_x)
            vp = lambda t: C.c_void_p(t.data_ptr())
            eng._check(eng.lib.gsm_qt_transform(eng.h, vp(q), vp(ref), int(q.numel()), vp(d_x), vp(d_out), x.size, 0, eng._stream()))
            want = qt.transform(x.reshape(-1, 1))[:, 0]
            got = d_out.cpu().numpy()
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12, equal_nan=True)
            eng._check(eng.lib.gsm_qt_transform(eng.h, vp(q), vp(ref), int(q.numel()), vp(d_out), vp(d_x), x.size, 1, eng._stream()))
            want_i = qt.inverse_transform(want.reshape(-1, 1))[:, 0]
            np.testing.assert_allclose(d_x.cpu().numpy(), want_i, rtol=0, atol=1e-9, equal_nan=True)
    finally:
        eng.close()


@pytest.mark.parametrize("transform,dy,n_iter", [(False, sc.TIE_FREE_DY, 1000), (True, sc.TIE_FREE_DY, 300), (False, 500.0, 300)])
def test_philox_mode_chain_equals_its_oracle(transform, dy, n_iter):
    """Philox mode of the small-scale chain (draws on the device, gsm_sgs_draw_philox; the whole chain without host work per
    iteration) against oracle/sgs_oracle.py driven by oracle/sgs_philox_oracle.PhiloxSgsRng: blocks, accept masks and
    resampled counts identical over the whole run, losses and beds to the tolerance of the kriging solve; two run() calls
    continue the counters.  dy = 503.7: a geometry without equidistant candidates (the oracle's sort order is then irrelevant);
    dy = 500: the square grid, where sectors do hold equidistant candidates at their cut -- the oracle sorts stably there, the
    device's rule (autouse fixture above)."""
    import sgs_philox_oracle as spo
    from mcmc_gpu_amd import sgs
    H = 32
    prob = sc.problem(H)
    if dy != 500.0:
        prob["yy"] = np.ascontiguousarray(np.broadcast_to((np.arange(H) * dy)[:, None], (H, H)))
    sill = float(np.var(prob["bed"]))
    nst = None
    if transform:
        from sklearn.preprocessing import QuantileTransformer
        nst = QuantileTransformer(n_quantiles=300, output_distribution="normal", random_state=1, subsample=None).fit(prob["bed"].reshape(-1, 1))
        sill = 1.0
    grounded = np.ones((H, H), dtype=int)
    cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], prob["cond_bed"],
                       prob["data_mask"], grounded, prob["region_mask"], prob["resolution"], 60.0,
                       [0, 0.0, 6000.0, 6000.0, sill, "Exponential", None], [16, 4000.0, False, 0], 3, 8, 3, 8, nst_trans=nst)
    seed = 2 ** 40 + 77
    so.TIE_LOG = []
    try:
        ref = so.run_chain_sgs(cfg, prob["bed"], n_iter, spo.PhiloxSgsRng(seed, ~np.isnan(prob["cond_bed"])))
        tied = len(so.TIE_LOG)
    finally:
        so.TIE_LOG = None
    assert (tied == 0) if dy != 500.0 else (tied > 0), f"{tied} sector cuts between equidistant candidates"
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           prob["cond_bed"], prob["data_mask"], grounded, prob["resolution"])
    ch.set_update_region(True, prob["region_mask"]); ch.set_loss_type(sigma_mc=60.0, massConvInRegion=True)
    ch.set_normal_transformation(nst, do_transform=transform); ch.set_trend(None, detrend_map=False)
    ch.set_variogram("Exponential", 6000.0, sill, 0.0, isotropic=True); ch.set_sgs_param(16, 4000.0); ch.set_block_sizes(3, 8, 3, 8)
    ch.set_random_generator(rng_seed=seed)
    ch.set_rng_mode('philox')
    out = ch.run(n_iter, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert np.array_equal(out[6], ref[6]), "blocks"
    assert np.array_equal(out[4], ref[4]), "accept masks"
    assert np.array_equal(out[5], ref[5]), "resampled counts"
    np.testing.assert_allclose(out[3], ref[3], rtol=1e-9)
    np.testing.assert_allclose(out[0], ref[0], rtol=0, atol=1e-7 if not transform else 1e-6)
    assert 0.05 < out[4].mean() < 0.98
    # a second segment continues the iteration counter: equal to one run of both lengths
    ch2 = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                            prob["cond_bed"], prob["data_mask"], grounded, prob["resolution"])
    ch2.__dict__.update({k: v for k, v in ch.__dict__.items() if k not in ("initial_bed",)})
    ch2.initial_bed = prob["bed"]; ch2.philox_iter = 0
    n_a = n_iter * 5 // 8
    a = ch2.run(n_a, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    ch2.initial_bed = a[0]
    b = ch2.run(n_iter - n_a, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert np.array_equal(np.concatenate([a[4], b[4]]), out[4]) and np.array_equal(np.concatenate([a[6], b[6]]), out[6])
    np.testing.assert_allclose(b[0], out[0], rtol=0, atol=1e-9)


def test_small_scale_driver_philox_mode_segments(tmp_path):
    """smallScaleChain_mp(mode='philox'): two 1000-iteration segments (the second resumes from the seed folders' files and
    continues the Philox iteration counter) equal one 2000-iteration call."""
    from mcmc_gpu_amd import driver, sgs, synthetic
    prob, ch = synthetic.sgs_template(32, transform=False, light=True)
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(3)]
    seeds = [811, 822, 833]
    one = driver.smallScaleChain_mp(3, 1, ch, beds, seeds, 5, [2000] * 3, output_path=str(tmp_path / "one"), mode='philox')
    driver.smallScaleChain_mp(3, 1, ch, beds, seeds, 5, [1000] * 3, output_path=str(tmp_path / "two"), mode='philox')
    two = driver.smallScaleChain_mp(3, 1, ch, beds, seeds, 5, [1000] * 3, output_path=str(tmp_path / "two"), mode='philox')
    for c in range(3):
        assert np.array_equal(one[c][4][1000:], two[c][4]) and np.array_equal(one[c][6][1000:], two[c][6])
        np.testing.assert_allclose(one[c][0], two[c][0], rtol=0, atol=1e-6)     # the bed passes through a text file between segments
    f = tmp_path / "two" / "LargeScaleChain" / "5" / "SmallScaleChain" / "811"
    assert np.loadtxt(f / "steps_2k.txt").shape == (2000,)
    # segments that are not multiples of 1000: the resume continues the Philox counters at the exact iteration count (the
    # length of the stored records), not at the file label's floor(count / 1000) * 1000
    one = driver.smallScaleChain_mp(3, 1, ch, beds, seeds, 5, [3000] * 3, output_path=str(tmp_path / "one15"), mode='philox')
    driver.smallScaleChain_mp(3, 1, ch, beds, seeds, 5, [1500] * 3, output_path=str(tmp_path / "two15"), mode='philox')
    two = driver.smallScaleChain_mp(3, 1, ch, beds, seeds, 5, [1500] * 3, output_path=str(tmp_path / "two15"), mode='philox')
    for c in range(3):
        assert np.array_equal(one[c][4][1500:], two[c][4]) and np.array_equal(one[c][6][1500:], two[c][6])
    # file label as the reference computes it ((1000 + 1500) // 1000 = 2: its label arithmetic restarts from k * 1000); the record holds all 3000
    assert np.loadtxt(tmp_path / "two15" / "LargeScaleChain" / "5" / "SmallScaleChain" / "811" / "steps_2k.txt").shape == (3000,)


def test_small_scale_driver_starts_its_own_ranks(tmp_path, monkeypatch):
    """smallScaleChain_mp(n_gpus=2) without a launcher: two self-started ranks (sharing the box's one GPU, gloo instead of RCCL)
    take contiguous shards of the chains and the results are gathered -- equal to the single-process run in both draw modes."""
    from mcmc_gpu_amd import driver, synthetic
    monkeypatch.setenv("GSM_DIST_BACKEND", "gloo")
    prob, ch = synthetic.sgs_template(32, transform=False, light=True)
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(5)]
    seeds = [911, 922, 933, 944, 955]
    for mode in ("philox", "replay"):
        one = driver.smallScaleChain_mp(5, 1, ch, beds, seeds, 5, [1000] * 5, output_path=str(tmp_path / f"one_{mode}"), mode=mode, n_gpus=1)
        two = driver.smallScaleChain_mp(5, 1, ch, beds, seeds, 5, [1000] * 5, output_path=str(tmp_path / f"two_{mode}"), mode=mode, n_gpus=2)
        assert len(two) == 5
        for a, b in zip(one, two):
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True)


def test_windowed_iteration_end_equals_the_full_grid_path(monkeypatch):
    """Without a transformer gsm_sgs_finish scores a proposal from the block and its one-cell halo (carried squared residuals)
    and decides / commits in the same launch; GSM_SGS_WINDOWED=0 is the full-grid loss + guard + decide + commit sequence that
    follows the reference's whole-map recomputation (MCMC.py:1781-1812).  Same accept masks, beds and counts; losses equal up to
    summation order.  With and without a trend; a block at the grid border; a proposal that grounds the ice is refused."""
    from mcmc_gpu_amd import sgs, synthetic
    for use_trend in (False, True):
        res = []
        for flag in ("1", "0"):
            monkeypatch.setenv("GSM_SGS_WINDOWED", flag)
            prob, ch = synthetic.sgs_template(48, transform=False, light=True)
            ch.set_update_region(False)
            ch.set_loss_type(sigma_mc=60.0, massConvInRegion=False)
            if use_trend:
                ch.set_trend(prob["surf"] - 1000.0, detrend_map=True)
            beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(6)]
            surf = prob["surf"].copy()
            surf[20:28, 20:28] = np.max(beds, axis=0)[20:28, 20:28] + 1.5     # thin ice: proposals there often ground it (guard)
            ch.surf = surf
            out, _ = sgs.run_many_sgs(ch, beds, [np.random.default_rng(i) for i in range(6)], 400, philox_seeds=[300 + i for i in range(6)])
            res.append(out)
        for a, b in zip(*res):
            assert np.array_equal(a[4], b[4]) and np.array_equal(a[6], b[6]) and np.array_equal(a[5], b[5])
            assert np.array_equal(a[0], b[0])
            np.testing.assert_allclose(a[3], b[3], rtol=1e-12)
        acc = np.mean([o[4].mean() for o in res[0]])
        assert 0.05 < acc < 0.95


def test_philox_mode_keeps_per_iteration_records():
    """chain_sgs.run's full signature in Philox mode (MCMC.py:1599, :1814-1829): only_save_last_bed=False returns the bed of every
    iteration, sample locations their values -- the same chain as the last-bed-only run."""
    from mcmc_gpu_amd import synthetic
    n_iter = 60
    outs = []
    for keep in (False, True):
        prob, ch = synthetic.sgs_template(32, transform=True, light=True)
        ch.set_random_generator(rng_seed=4242)
        ch.set_rng_mode('philox')
        if keep:
            loc = np.array([[prob["xx"][5, 7], prob["yy"][5, 7]], [prob["xx"][20, 11], prob["yy"][20, 11]]])
            ch.set_sample_points_locations(loc)
        outs.append(ch.run(n_iter, only_save_last_bed=not keep, info_per_iter=10 ** 9, plot=False, progress_bar=None))
    last, full = outs
    assert len(full) == 8 and full[0].shape == (n_iter, 32, 32)
    assert np.array_equal(full[4], last[4]) and np.array_equal(full[6], last[6]) and np.array_equal(full[3], last[3])
    assert np.array_equal(full[0][-1], last[0])
    tr = ch.trend if ch.detrend_map else np.zeros((32, 32))
    np.testing.assert_allclose(full[7][0, 1:], full[0][1:, 5, 7] - tr[5, 7], rtol=0, atol=1e-9)      # sample values are detrended beds
    np.testing.assert_allclose(full[7][1, 1:], full[0][1:, 20, 11] - tr[20, 11], rtol=0, atol=1e-9)
    moved = np.abs(np.diff(full[0], axis=0)).max(axis=(1, 2)) > 0          # the bed changes exactly on accepted iterations
    assert np.array_equal(moved, full[4][1:].astype(bool))


@pytest.mark.parametrize("transform,pcg64", [(True, False), (False, False), (True, True)])
def test_graph_replay_of_a_batch_equals_the_eager_launches(monkeypatch, transform, pcg64):
    """gsm_sgs_iterate issues the loop body of chain_sgs.run (MCMC.py:1741-1822) for a batch of iterations; with device draws the
    buffers are static, so the second full batch is captured into a hipGraph and later ones are replays.  GSM_SGS_GRAPH=0 issues
    the same launches one by one: every record must be identical bit for bit."""
    from mcmc_gpu_amd import sgs, synthetic
    res, replays = [], []
    monkeypatch.setenv("GSM_SGS_BATCH", "32")          # (few chains default to batches of 128: too few full batches in 200 iterations)
    for flag in ("1", "0"):
        monkeypatch.setenv("GSM_SGS_GRAPH", flag)
        prob, ch = synthetic.sgs_template(32, transform=transform, light=True)
        beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(5)]
        rngs = [np.random.default_rng(70 + i) for i in range(5)]
        out, rngs = sgs.run_many_sgs(ch, beds, rngs, 200, philox_seeds=None if pcg64 else [11 + i for i in range(5)], pcg64=pcg64)
        res.append((out, [r.bit_generator.state for r in rngs]))
        replays.append(sgs.LAST_GRAPH_REPLAYS)
    assert replays[0] >= 4 and replays[1] == 0, replays          # 200 iterations = 6 full batches of 32 (eager, capture + 5 launches) + 8
    (a, sa), (b, sb) = res
    assert sa == sb
    for x, y in zip(a, b):
        for k in (0, 3, 4, 5, 6):
            assert np.array_equal(np.asarray(x[k], dtype=float), np.asarray(y[k], dtype=float), equal_nan=True), k


@pytest.mark.parametrize("transform,pcg64,light", [(True, False, False), (False, False, True), (True, True, False)])
def test_records_made_ahead_and_draws_made_ahead_change_nothing(monkeypatch, transform, pcg64, light):
    """With no NaN in the beds gsm_sgs_iterate makes the records of iteration j + 1 (ranks, search, kriging weights) on a second
    stream while iteration j runs -- the records then name their cells and the value pass reads the values from the grid (gsm.h:
    grid_finite) --, and run_many_sgs draws batch b + 1 while batch b iterates.  Neither changes a number: every record and the
    generators' final states are identical bit for bit to the launches in the reference's order (MCMC.py:1741-1822)."""
    from mcmc_gpu_amd import sgs, synthetic
    res = []
    for overlap, ahead, tail_qt in (("1", "1", "1"), ("0", "0", "0"), ("1", "0", "0"), ("0", "1", "1")):
        monkeypatch.setenv("GSM_SGS_OVERLAP", overlap)
        monkeypatch.setenv("GSM_SGS_DRAW_AHEAD", ahead)
        monkeypatch.setenv("GSM_SGS_TAIL_QT", tail_qt)         # both transforms of an iteration inside the tail launch, or stand-alone
        prob, ch = synthetic.sgs_template(48 if not light else 32, transform=transform, light=light)
        beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(3)]
        rngs = [np.random.default_rng(170 + i) for i in range(3)]
        out, rngs = sgs.run_many_sgs(ch, beds, rngs, 100, philox_seeds=None if pcg64 else [21 + i for i in range(3)], pcg64=pcg64)
        res.append((out, [r.bit_generator.state for r in rngs]))
    for other, so in res[1:]:
        assert so == res[0][1]
        for x, y in zip(res[0][0], other):
            for k in (0, 3, 4, 5, 6):
                assert np.array_equal(np.asarray(x[k], dtype=float), np.asarray(y[k], dtype=float), equal_nan=True), k
    assert sum(float(np.asarray(o[4]).sum()) for o in res[0][0]) > 0      # some proposals were accepted: the commit path ran


@pytest.mark.parametrize("tag", ["ok", "sk", "skm"])
def test_sgs_function_ok_and_sk_equal_reference_fixture(tag):
    """Golden F12: the reference's module-level MCMC.sgs (MCMC.py:91-173) with ordinary and simple kriging (_krige.py:5-44, :46-81;
    gsm_sgs_set_kriging) on a tie-free grid.  Same generator consumption (final state identical), same NaN cells, simulated values
    to 1e-8 absolute on standardised values (lstsq vs elimination, values feeding on each other cell after cell)."""
    from mcmc_gpu_amd import sgs
    g = np.load(sc.GOLD12, allow_pickle=False)
    xx, yy, grid, vario, kw, seed = sc.f12_case(tag)
    rng = np.random.default_rng(seed)
    out = sgs.sgs(xx, yy, grid.copy(), dict(vario), seed=rng, **kw)
    exp = g[f"{tag}_out"]
    assert rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))
    assert np.array_equal(np.isnan(out), np.isnan(exp))
    keep = ~np.isnan(grid)
    assert np.array_equal(out[keep], grid[keep])                              # conditioning values untouched
    new = np.isnan(grid) & ~np.isnan(exp)
    assert new.sum() == int(g[f"{tag}_n_sim"])
    np.testing.assert_allclose(out[new], exp[new], rtol=0, atol=1e-8)
    if tag != "ok":                                                           # simple != ordinary kriging on this problem
        other = sgs.sgs(xx, yy, grid.copy(), dict(vario), seed=np.random.default_rng(seed), **dict(kw, ktype="ok"))
        assert np.abs(other[new] - out[new]).max() > 1e-3


def _max_block_setup(block_min, block_max):
    from mcmc_gpu_amd import sgs
    H = 64
    prob = sc.orc.synthetic_problem(H, res=500.0)
    prob["yy"] = np.ascontiguousarray(np.broadcast_to((np.arange(H) * sc.TIE_FREE_DY)[:, None], (H, H)))
    data_mask = np.zeros((H, H), dtype=bool); data_mask[::8, :] = True; data_mask[:, ::16] = True
    cond = np.where(data_mask, prob["bed"], np.nan)
    grounded = np.ones((H, H), dtype=int)
    sill = float(np.var(prob["bed"]))
    region = np.zeros((H, H), dtype=int); region[17:47, 17:47] = 1            # block centres: no block of <= 32 cells is clipped
    vp = [0, 0.0, 6000.0, 6000.0, sill, "Exponential", None]
    cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], cond, data_mask,
                       grounded, region, prob["resolution"], 5000.0, vp, [16, 4000.0, False, 0], block_min, block_max, block_min, block_max)
    ch = sgs.chain_sgs_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                           cond, data_mask, grounded, prob["resolution"])
    ch.set_update_region(True, region)
    ch.set_loss_type(sigma_mc=5000.0, massConvInRegion=True)       # flat posterior: proposals are accepted, their values recorded
    ch.set_normal_transformation(None, do_transform=False)
    ch.set_trend(None, detrend_map=False)
    ch.set_variogram("Exponential", 6000.0, sill, 0.0, isotropic=True)
    ch.set_sgs_param(16, 4000.0)
    ch.set_block_sizes(block_min, block_max, block_min, block_max)
    return prob, cfg, ch


@pytest.mark.parametrize("mode", ["replay", "pcg64"])
def test_largest_blocks_the_device_takes_equal_the_oracle(mode):
    """Block size 32 in both directions (rng.integers(32, 33)): every window is 32 x 32 = 1024 cells, the most one block simulation
    holds (window and record buffers of sgs_kernel.hip) -- against the oracle on the same generator, with host draws and with
    the generator advanced on the device."""
    prob, cfg, ch = _max_block_setup(32, 33)
    n_iter = 4
    rng_o = np.random.default_rng(seed=314)
    ref = so.run_chain_sgs(cfg, prob["bed"], n_iter, rng_o)
    ch.set_random_generator(rng_seed=314)
    ch.set_rng_mode(mode)
    out = ch.run(n_iter, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert np.array_equal(out[6], ref[6]) and np.array_equal(out[4], ref[4]) and np.array_equal(out[5], ref[5])
    assert np.all(ref[6][1:, 2] * ref[6][1:, 3] == 1024) and ref[5].max() >= 1
    np.testing.assert_allclose(out[3], ref[3], rtol=1e-9)
    np.testing.assert_allclose(out[0], ref[0], rtol=0, atol=1e-7)
    assert ch.rng.bit_generator.state == rng_o.bit_generator.state


def test_blocks_beyond_the_device_limit_are_refused():
    """A block of more than 1024 cells does not fit one block simulation: the call ends with the library's error, not with a
    wrong bed (block sizes up to 40 -> windows up to 39 x 39)."""
    from mcmc_gpu_amd._lib import GsmError
    prob, cfg, ch = _max_block_setup(36, 41)
    ch.set_random_generator(rng_seed=1)
    with pytest.raises(GsmError):
        ch.run(3, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)


class _WrappedTransformer:
    """A normal-score transformer that is NOT scikit-learn's class (chain_sgs accepts any object with transform /
    inverse_transform, MCMC.py:1766, :1777): the product then calls it on the host once per iteration."""

    def __init__(self, inner):
        self.inner = inner

    def transform(self, x):
        return self.inner.transform(x)

    def inverse_transform(self, x):
        return self.inner.inverse_transform(x)


@pytest.mark.parametrize("mode", ["replay", "pcg64", "philox"])
def test_transformer_of_another_class_runs_on_the_host_in_every_draw_mode(mode):
    """The same chain with scikit-learn's QuantileTransformer (on the device: gsm_qt_transform) and with that transformer behind a
    wrapper class (called on the host where the reference calls it, the draws still made where the mode makes them): identical
    blocks, accept masks, counts and generator state; beds and losses to the device transform's agreement with scikit-learn."""
    from mcmc_gpu_amd import synthetic
    outs, states = [], []
    for wrap in (False, True):
        prob, ch = synthetic.sgs_template(32, transform=True, light=True)
        if wrap:
            ch.set_normal_transformation(_WrappedTransformer(ch.nst_trans), do_transform=True)
        ch.set_random_generator(rng_seed=2024)
        ch.set_rng_mode(mode)
        outs.append(ch.run(40, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None))
        states.append(ch.rng.bit_generator.state)
    a, b = outs
    assert np.array_equal(a[6], b[6], equal_nan=True) and np.array_equal(a[4], b[4]) and np.array_equal(a[5], b[5])
    assert states[0] == states[1]
    np.testing.assert_allclose(a[3], b[3], rtol=1e-9)
    np.testing.assert_allclose(a[0], b[0], rtol=0, atol=1e-7)
    assert 0.05 < a[4].mean() < 0.95


def test_checkpoint_labels_follow_the_reference_when_n_iter_is_not_a_multiple_of_1000(tmp_path):
    """Two 1500-iteration segments: the reference restarts its label arithmetic from k * 1000 (smallScaleChain_multiprocessing.py:
    322-383), so the files are *_1k.txt, then *_2k.txt ((1000 + 1500) // 1000) -- not 3k -- while the stored records hold all 3000
    iterations (and a Philox-mode resume continues its counters at 3000)."""
    from mcmc_gpu_amd import driver, synthetic
    prob, ch = synthetic.sgs_template(64, transform=False, light=True)
    seeds, beds = [911], [prob["bed"]]
    for _ in range(2):
        driver.smallScaleChain_mp(1, 1, ch, beds, seeds, 123456789, [1500], output_path=str(tmp_path))
    folder = tmp_path / "LargeScaleChain" / "123456" / "SmallScaleChain" / "911"
    assert sorted(p.name for p in folder.iterdir()) == sorted(
        f"{k}_2k.txt" for k in ("bed", "loss_mc", "loss_data", "loss", "steps", "resampled_times", "blocks_used"))
    assert np.loadtxt(folder / "loss_2k.txt").shape == (3000,)
    assert driver._msc_load_previous(folder)["cumulative"] == 3000


def test_elongated_blocks_without_a_transformer_take_the_whole_map_path(monkeypatch):
    """A block table up to 8 x 128 cells is legal (<= 1024 cells) but its block + halo, 10 x 130 = 1300 cells, exceeds the 1296-cell
    LDS tile of the windowed iteration end (gsm_sgs_finish), which used to abort such chains with error flag 1: the chain now takes
    the whole-map loss / decide / commit path for such a table -- the same results as with the windowed path switched off by hand."""
    from mcmc_gpu_amd import synthetic
    outs = []
    for windowed in ("1", "0"):
        monkeypatch.setenv("GSM_SGS_WINDOWED", windowed)
        prob, ch = synthetic.sgs_template(160, transform=False, light=True)
        ch.set_block_sizes(2, 9, 100, 129)
        ch.set_random_generator(rng_seed=77)
        outs.append(ch.run(25, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None))
    a, b = outs
    assert np.isfinite(a[3]).all() and np.nanmax(a[6][1:, 2:4]) > 100          # long blocks were drawn
    for k in (0, 3, 4, 5):
        assert np.array_equal(a[k], b[k]), k
