"""CPU: mcmc_gpu_amd/csrc/normal_score.h (the device's QuantileTransformer / ndtri / ndtr restatement, compiled for the host
with g++) against scipy.special and scikit-learn themselves."""
import ctypes as C
import shutil
import subprocess
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sp = pytest.importorskip("scipy.special")
skp = pytest.importorskip("sklearn.preprocessing")


@pytest.fixture(scope="module")
def lib(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not found")
    so = tmp_path_factory.mktemp("ns") / "libns.so"
    subprocess.run(["g++", "-O2", "-ffp-contract=off", "-shared", "-fPIC", "-I", str(ROOT / "mcmc_gpu_amd" / "csrc"), "-o", str(so),
                    str(ROOT / "tests" / "native" / "normal_score_host.cpp")], check=True)
    L = C.CDLL(str(so))
    dp = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
    L.ns_ndtri.argtypes = [dp, dp, C.c_int]; L.ns_ndtr.argtypes = [dp, dp, C.c_int]
    L.ns_qt.argtypes = [dp, dp, C.c_int, dp, dp, C.c_int, C.c_double, C.c_double, C.c_int]
    return L


def test_ndtri_and_ndtr_equal_scipy(lib):
    rng = np.random.default_rng(0)
    p = np.concatenate([rng.random(200000), 10.0 ** rng.uniform(-300, 0, 50000), 1.0 - 10.0 ** rng.uniform(-16, 0, 50000),
                        [0.0, 1.0, 0.5, 0.13533528323661269, 1 - 0.13533528323661269, 1e-7, 1 - 1e-7]])
    out = np.empty_like(p); lib.ns_ndtri(p, out, p.size)
    ref = sp.ndtri(p)
    fin = np.isfinite(ref)
    assert np.array_equal(np.isfinite(out), fin) and np.array_equal(out[~fin], ref[~fin])
    assert np.abs(out[fin] - ref[fin]).max() <= 4e-16 * np.maximum(1.0, np.abs(ref[fin])).max()
    assert np.max(np.abs(out[fin] - ref[fin]) / np.maximum(np.abs(ref[fin]), 1e-300)) < 1e-14
    x = np.concatenate([rng.normal(0, 3, 200000), rng.uniform(-40, 40, 50000), [0.0, -1.0, 1.0, np.sqrt(2), -np.sqrt(2), 8 * np.sqrt(2)]])
    out = np.empty_like(x); lib.ns_ndtr(x, out, x.size)
    ref = sp.ndtr(x)
    assert np.max(np.abs(out - ref) / np.maximum(ref, 1e-300)) < 1e-14


@pytest.mark.parametrize("nq,repeats", [(1000, False), (500, True), (17, False)])
def test_quantile_transformer_equals_sklearn(lib, nq, repeats):
    rng = np.random.default_rng(3)
    data = rng.normal(-300.0, 120.0, 5000)
    if repeats:
        data[:1500] = np.round(data[:1500] / 50.0) * 50.0          # repeated values -> repeated quantiles
    qt = skp.QuantileTransformer(n_quantiles=nq, output_distribution="normal", subsample=None).fit(data.reshape(-1, 1))
    q = np.ascontiguousarray(qt.quantiles_[:, 0]); ref = np.ascontiguousarray(qt.references_)
    from scipy import stats
    cmin = stats.norm.ppf(1e-7 - np.spacing(1)); cmax = stats.norm.ppf(1 - (1e-7 - np.spacing(1)))
    x = np.concatenate([rng.normal(-300.0, 200.0, 20000), data[:2000], q, [q[0] - 1.0, q[-1] + 1.0, q[0], q[-1], np.nan]])
    want = qt.transform(x.reshape(-1, 1))[:, 0]
    got = np.empty_like(x); lib.ns_qt(np.ascontiguousarray(x), got, x.size, q, ref, q.size, cmin, cmax, 0)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-13, equal_nan=True)
    z = np.concatenate([rng.normal(0, 1.5, 20000), want[~np.isnan(want)][:3000], [-6.0, 6.0, cmin, cmax, 0.0, np.nan]])
    want_i = qt.inverse_transform(z.reshape(-1, 1))[:, 0]
    got_i = np.empty_like(z); lib.ns_qt(np.ascontiguousarray(z), got_i, z.size, q, ref, q.size, cmin, cmax, 1)
    np.testing.assert_allclose(got_i, want_i, rtol=0, atol=1e-10, equal_nan=True)
