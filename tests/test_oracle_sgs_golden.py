"""not gpu: the small-scale-chain restatement (oracle/sgs_oracle.py) against golden F10, generated from the imported
reference with bit-identity asserted (oracle/make_fixtures.py: make_f10_sgs).  Groundwork for SURVEY.md section 8f
rank 3 -- there is no HIP path for chain_sgs yet; this pins the CPU checker the HIP path will be tested against."""
import hashlib
import json
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import mcmc_oracle as orc  # noqa: E402
import sgs_oracle as so  # noqa: E402

GOLD = ROOT / "tests" / "golden" / "f10_sgs_chain32.npz"


def _problem(H):
    """Same synthetic inputs as make_fixtures.sgs_problem (kept in step by the fixture values themselves)."""
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


@pytest.mark.parametrize("tag,vtype,smooth,use_trend,use_nst", [("a", "Exponential", None, False, False),
                                                                ("b", "Matern", 1.5, True, True)])
def test_sgs_oracle_reproduces_reference_fixture(tag, vtype, smooth, use_trend, use_nst):
    g = np.load(GOLD, allow_pickle=False)
    H, n_iter = int(g["H"]), int(g["n_iter"])
    prob = _problem(H)
    trend = prob["trend"] if use_trend else None
    nst = None
    if use_nst:
        from sklearn.preprocessing import QuantileTransformer
        data = (prob["cond_bed"] - trend)[prob["data_mask"]].reshape(-1, 1)
        nst = QuantileTransformer(n_quantiles=200, output_distribution="normal", random_state=152).fit(data)
    sill, seed, sigma = float(g[f"{tag}_sill"]), int(g[f"{tag}_seed"]), float(g[f"{tag}_sigma_mc"])
    rr, rad, npts = float(g["range"]), float(g["radius"]), int(g["num_points"])
    cfg = so.SgsConfig(prob["xx"], prob["yy"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"],
                       prob["cond_bed"], prob["data_mask"], np.ones((H, H), dtype=int), prob["region_mask"],
                       prob["resolution"], sigma, [0, 0.0, rr, rr, sill, vtype, smooth], [npts, rad, False, 0],
                       3, 8, 3, 8, trend=trend, nst_trans=nst)
    rng = np.random.default_rng(seed=seed)
    trace = []
    out = so.run_chain_sgs(cfg, prob["bed"], n_iter, rng, trace=trace)
    assert np.array_equal(out[0], g[f"{tag}_bed"])
    assert np.array_equal(out[3], g[f"{tag}_loss"])
    assert np.array_equal(out[4], g[f"{tag}_steps"])
    assert np.array_equal(out[5], g[f"{tag}_resampled"])
    assert np.array_equal(out[6], g[f"{tag}_blocks"])
    assert rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))
    tr = np.array(trace)
    assert len(trace) == int(g[f"{tag}_n_sim"])
    assert np.array_equal(tr[:40], g[f"{tag}_trace_head"])
    assert hashlib.sha256(np.ascontiguousarray(tr).tobytes()).hexdigest() == str(g[f"{tag}_trace_sha"])


def test_octant_search_properties():
    """neighbors(): at most num_points // 8 points per 45-degree sector, all within the radius, nearest first."""
    H = 24
    prob = _problem(H)
    grid = prob["cond_bed"].copy()
    cond = ~np.isnan(grid)
    ii, jj = np.meshgrid(np.arange(H), np.arange(H), indexing="ij")
    stencil, _, _ = so.make_circle_stencil(prob["xx"][0, :], 3000.0)
    pts = so.neighbors(10, 13, ii, jj, prob["xx"], prob["yy"], grid, cond, 3000.0, 16, stencil=stencil)
    assert 0 < pts.shape[0] <= 16
    d = np.hypot(pts[:, 0] - prob["xx"][10, 13], pts[:, 1] - prob["yy"][10, 13])
    assert (d < 3000.0).all()
    ang = np.arctan2(prob["yy"][10, 13] - pts[:, 1], prob["xx"][10, 13] - pts[:, 0])
    sector = np.ceil(ang / (np.pi / 4)).astype(int)
    for s_ in np.unique(sector):
        ds = d[sector == s_]
        assert ds.size <= 2 and (np.diff(ds) >= 0).all()


# ---- golden F11: the reference driver's own small-scale configuration (oracle/make_fixtures_r3.py) ---------------------------
import sgs_common as sc  # noqa: E402

F11_CASES = {
    # tag: (tie-free geometry, trend, transformer, vario_param, sgs_param, sigma, stable sort)
    "a": (False, True, True, None, None, 5.0, True),
    "t": (True, True, True, None, None, 5.0, False),
    "b": (True, False, False, [0, 0.0, 8000.0, 8000.0, 30.0, "Exponential", None], [16, 1200.0, False, 0], 40.0, False),
    "c": (True, True, False, [35.0, 2.0, 9000.0, 5000.0, 40.0, "Spherical", None], [48, 30e3, False, 0], 40.0, False),
}


def f11_case(tag):
    """(golden arrays, problem, trend, transformer, vario_param, sgs_param, sigma, stable) of F11 case `tag`."""
    g = np.load(sc.GOLD11, allow_pickle=False)
    tie_free, use_trend, use_nst, vp, sp, sigma, stable = F11_CASES[tag]
    prob = sc.driver_problem(int(g["H"]), dy=float(g["tie_free_dy"]) if tie_free else 500.0)
    trend, nst = sc.driver_trend_and_transformer(prob)
    assert hashlib.sha256(np.ascontiguousarray(trend).tobytes()).hexdigest() == str(g["trend_sha"]), "scipy's gaussian_filter differs here"
    assert hashlib.sha256(np.ascontiguousarray(nst.quantiles_).tobytes()).hexdigest() == str(g["quantiles_sha"]), "sklearn's quantiles differ here"
    return g, prob, (trend if use_trend else None), (nst if use_nst else None), vp, sp, sigma, stable


@pytest.mark.parametrize("tag", ["a", "t", "b", "c"])
def test_sgs_oracle_reproduces_driver_config_fixture(tag):
    """F11: 48 neighbours within 30 km at 500 m (search half-width 60 cells), blocks 5-20, Matern, QuantileTransformer(1000),
    trend (case a: square grid, argsort forced stable in the reference; t: tie-free geometry, unmodified reference), the
    radius-widening fallback (b) and the spherical far field with 48 neighbours (c)."""
    g, prob, trend, nst, vp, sp, sigma, stable = f11_case(tag)
    cfg = sc.driver_cfg(prob, trend, nst, vp, sp, None, sigma)
    rng = np.random.default_rng(seed=int(g[f"{tag}_seed"]))
    trace = []
    so.STABLE_TIES, so.TIE_LOG = stable, []
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = so.run_chain_sgs(cfg, prob["bed"], int(g[f"{tag}_n_iter"]), rng, trace=trace)
        tied = len(so.TIE_LOG)
    finally:
        so.STABLE_TIES, so.TIE_LOG = False, None
    assert tied == int(g[f"{tag}_tied_cuts"])
    assert (tied == 0) or stable
    for k, name in ((0, "bed"), (3, "loss"), (4, "steps"), (5, "resampled"), (6, "blocks")):
        assert np.array_equal(out[k], g[f"{tag}_{name}"], equal_nan=True), name
    assert rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))
    tr = np.array(trace)
    assert len(trace) == int(g[f"{tag}_n_sim"])
    assert hashlib.sha256(np.ascontiguousarray(tr).tobytes()).hexdigest() == str(g[f"{tag}_trace_sha"])
    if tag == "b":
        assert tr[:, 2].min() >= 1                         # the widened search found something for every cell


def f13_case():
    """Golden F13 (oracle/make_fixtures_r4.py): the driver configuration on the tie-free geometry, sigma_mc 30, 160 iterations."""
    g = np.load(sc.GOLD13, allow_pickle=False)
    prob = sc.driver_problem(int(g["H"]), dy=float(g["tie_free_dy"]))
    trend, nst = sc.driver_trend_and_transformer(sc.driver_problem(int(g["H"])))
    assert hashlib.sha256(np.ascontiguousarray(trend).tobytes()).hexdigest() == str(g["trend_sha"]), "scipy's gaussian_filter differs here"
    assert hashlib.sha256(np.ascontiguousarray(nst.quantiles_).tobytes()).hexdigest() == str(g["quantiles_sha"]), "sklearn's quantiles differ here"
    return g, prob, trend, nst, float(g["sigma_mc"])


def test_sgs_oracle_reproduces_deep_driver_fixture():
    """F13: the UNMODIFIED reference at the driver's parameters (48 neighbours / 30 km, blocks 5-20, Matern, QuantileTransformer(1000),
    trend) over 160 iterations with 68 accepted -- the accepted path (inverse transform, commit, the next iteration conditioning on
    the committed values) is walked 68 times, against twice in F11's cases a / t."""
    g, prob, trend, nst, sigma = f13_case()
    cfg = sc.driver_cfg(prob, trend, nst, None, None, None, sigma)
    rng = np.random.default_rng(seed=int(g["d_seed"]))
    trace = []
    so.STABLE_TIES, so.TIE_LOG = False, []
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            out = so.run_chain_sgs(cfg, prob["bed"], int(g["d_n_iter"]), rng, trace=trace)
        tied = len(so.TIE_LOG)
    finally:
        so.STABLE_TIES, so.TIE_LOG = False, None
    assert tied == 0 == int(g["d_tied_cuts"])
    for k, name in ((0, "bed"), (3, "loss"), (4, "steps"), (5, "resampled"), (6, "blocks")):
        assert np.array_equal(out[k], g[f"d_{name}"], equal_nan=True), name
    assert rng.bit_generator.state == json.loads(str(g["d_rng_state"]))
    assert int(out[4].sum()) >= 30
    assert hashlib.sha256(np.ascontiguousarray(np.array(trace)).tobytes()).hexdigest() == str(g["d_trace_sha"])


@pytest.mark.parametrize("tag", ["ok", "sk", "skm"])
def test_sgs_function_oracle_reproduces_reference_fixture_ok_and_sk(tag):
    """Golden F12 (oracle/make_fixtures_r3b.py): the reference's module-level MCMC.sgs with ktype 'ok' and 'sk' (_krige.py:5-81)
    on a tie-free grid; the restatement must give the same grid bit for bit and leave the generator in the same state."""
    import sgs_common as sc
    g = np.load(sc.GOLD12, allow_pickle=False)
    xx, yy, grid, vario, kw, seed = sc.f12_case(tag)
    rng = np.random.default_rng(seed)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        out = so.sgs(xx, yy, grid.copy(), dict(vario), rng=rng, **kw)
    assert np.array_equal(out, g[f"{tag}_out"], equal_nan=True)
    assert rng.bit_generator.state == json.loads(str(g[f"{tag}_rng_state"]))
    assert int(np.isnan(grid).sum() - np.isnan(out).sum()) == int(g[f"{tag}_n_sim"]) > 100
