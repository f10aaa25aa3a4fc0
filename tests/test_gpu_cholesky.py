"""-m gpu: covariance assembly and the precomputed-factor proposal generator (BASELINE configs[3] path)."""
import numpy as np
import pytest

import cholesky_oracle as co
import mcmc_oracle as orc
from gpu_common import make_engine
from mcmc_gpu_amd import cholesky as chol

pytestmark = pytest.mark.gpu


def test_cov_assemble_matches_reference_make_sigma(golden_dir):
    """Fixture F6: gstatsim_custom._krige.make_sigma on a 6x5 block, anisotropic, every covariance model."""
    g = np.load(golden_dir / "f6_covariance.npz")
    eng, *_ = make_engine(64, 1)
    for vt, extra in (("Exponential", {}), ("Gaussian", {}), ("Spherical", {}), ("Matern", {"s": 0.9125})):
        v = chol.make_vario(vt, 4000.0, 2500.0, azimuth=30.0, **extra)
        sig = chol.cov_assemble(eng, 6, 5, 500.0, v).cpu().numpy()
        ref = g[f"sigma_{vt.lower()}"]
        # fp64 tolerance: the device forms (coord @ R) and the lag with its own rounding; 1e-12 of the sill
        np.testing.assert_allclose(sig, ref, rtol=0, atol=1e-12)
        assert np.array_equal(sig, sig.T) or np.allclose(sig, sig.T, atol=1e-15)
    eng.close()


@pytest.mark.parametrize("model,iso,ncls", [("Exponential", True, 1), ("Matern", False, 3)])
def test_cholesky_proposals_match_oracle(model, iso, ncls):
    rfp = orc.RFParams(10e3, 50e3, 12e3, 40e3, 50, 150, 0.0, model, iso, 0.9125 if model == "Matern" else None)
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 3, rf_params=rfp)
    rfp.resolution = prob["resolution"]
    rfp.generator = "cholesky"
    factors = chol.build_factors(eng, rfp, n_classes=ncls, jitter=1e-8)
    varios = chol.class_varios(rfp, ncls)
    seeds = [7, 2 ** 40 + 12345, 99]
    out = eng.propose_philox(6, 500, seeds, rfp)
    centres = np.flatnonzero(cfg.region_mask.ravel() == 1)
    cache = {}
    seen = set()
    for c in range(3):
        for s in range(6):
            e = co.proposal(seeds[c], 500 + s, rfp, pairs, masks, centres, 64, prob["resolution"], varios, 1e-8, cache)
            assert int(out["size_idx"][c, s]) == e["size_idx"]
            assert tuple(out["centre"][c, s].tolist()) == e["centre"]
            assert float(out["u"][c, s]) == e["u"]
            assert int(out["rf_scalars"][c, s, 2]) == e["range_class"]
            seen.add(e["range_class"])
            bh, bw = e["field"].shape
            f = out["fields"][c, s, : bh * bw].cpu().numpy().reshape(bh, bw)
            # (a) against the oracle's own factor of the reference-pinned covariance: Cholesky of a matrix with
            #     condition number ~1e5 amplifies the 1e-16 assembly differences -> 1e-8 of the scale
            np.testing.assert_allclose(f, e["field"], rtol=0, atol=1e-8 * e["scale"])
            # (b) the matrix-core product itself, against the SAME device factor: tight
            g = e["size_idx"] * ncls + e["range_class"]
            U = factors[g].cpu().numpy()[: bh * bw, : bh * bw]
            exact = ((U.T @ e["z"]).reshape(bh, bw) * e["scale"]) * masks[e["size_idx"]]
            np.testing.assert_allclose(f, exact, rtol=0, atol=1e-12 * e["scale"])
    assert len(seen) == ncls or ncls == 1
    eng.close()


def test_cholesky_chain_runs_and_is_batch_independent():
    rfp = orc.standard_rf_params(model="Exponential")
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 4, rf_params=rfp)
    rfp.resolution = prob["resolution"]
    rfp.generator = "cholesky"
    chol.build_factors(eng, rfp, n_classes=2)
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(4)])
    seeds = [31, 32, 33, 34]
    eng.set_state(beds0)
    la, aa, ba = eng.run_philox(40, 0, seeds, rfp, batch=8)
    bed_a = eng.beds.cpu().numpy().copy()
    eng.set_state(beds0)
    lb, ab, bb = eng.run_philox(40, 0, seeds, rfp, batch=40)
    assert np.array_equal(aa, ab) and np.array_equal(la, lb) and np.array_equal(ba, bb)
    assert np.array_equal(bed_a, eng.beds.cpu().numpy())
    assert 0.2 < aa.mean() <= 1.0
    # the same proposals replayed through the step kernel give the same chain
    eng.set_state(beds0)
    p = eng.propose_philox(40, 0, seeds, rfp)
    lc, ac = eng.run_replay(p["size_idx"].cpu().numpy(), p["centre"].cpu().numpy(), p["u"].cpu().numpy(), p["fields"])
    assert np.array_equal(aa, ac) and np.array_equal(la, lc)
    eng.close()


def test_library_cholesky_matches_lapack():
    """gsm_cholesky_upper (blocked, MFMA trailing updates) against torch.linalg.cholesky / numpy on the covariance of
    a 50x56 block (N = 2800, padded to 2816) and on a small well-conditioned matrix; non-PD input is reported."""
    import torch
    from mcmc_gpu_amd._lib import GsmError
    eng, prob, *_ = make_engine(64, 1)
    v = chol.make_vario("Exponential", 20e3, 15e3, azimuth=20.0)
    sigma = chol.cov_assemble(eng, 50, 56, 500.0, v)
    U = chol.factor_upper_padded(eng, sigma, 1e-8)
    Ut = chol.factor_upper_padded(eng, sigma, 1e-8, use_torch=True)
    N = 2800
    assert U.shape == (2816, 2816) and torch.count_nonzero(U[N:, :]) == 0 and torch.count_nonzero(U[:, N:]) == 0
    assert torch.count_nonzero(torch.tril(U, -1)) == 0
    A = sigma + 1e-8 * torch.eye(N, dtype=torch.float64, device=sigma.device)
    rec = U[:N, :N].T @ U[:N, :N]
    assert float((rec - A).abs().max()) < 1e-12
    assert float((U - Ut).abs().max()) < 1e-9          # cond(Sigma) ~ 1e4: factors agree far below the proposal scale
    # Matern, through the whole build_factors path already covered by test_cholesky_proposals_match_oracle
    g = np.random.default_rng(1)
    M = g.normal(size=(128, 128)); S = M @ M.T + 128 * np.eye(128)
    t = torch.as_tensor(S).cuda()
    Us = chol.factor_upper_padded(eng, t, 0.0).cpu().numpy()
    np.testing.assert_allclose(Us, np.linalg.cholesky(S).T, rtol=0, atol=1e-12)
    bad = torch.as_tensor(S - 400 * np.eye(128)).cuda()
    with pytest.raises(GsmError, match="positive definite"):
        chol.factor_upper_padded(eng, bad, 0.0)
    eng.close()
