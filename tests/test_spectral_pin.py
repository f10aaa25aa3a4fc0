"""Value pin of the DEVICE spectral synthesis against the reference's arithmetic (gstatsMCMC/MCMC.py:221-251).

gsm_spectral_from_noise runs the proposal code of the fused chain kernel / gsm_propose_philox (spectral amplitude, folded
matrix-core inverse DFT of the Hermitian half plane, standardisation, scale, nugget, edge mask) on caller-supplied
white noise.  The noise is regenerated here from the fixture's seed with NumPy's Generator in the reference's draw
order; the expected fields are the imported reference's own outputs:
  F3   tests/golden/f3_fields64.npz        first five proposals of the standard 64x64 chain (blocks 8-16, masked)
  F3b  tests/golden/f3b_spectral_blocks.npz  50-80 cell blocks, three models, anisotropy, nugget (oracle/make_fixtures_r2.py)
Tolerance: 1e-12 x scale (fp64; the device sums the DFT in another order than pocketfft and evaluates S(k) with its own
exp/log).  CPU part: the oracle reproduces F3b bit for bit, and the half-plane identity the device relies on holds."""
import numpy as np
import pytest

import mcmc_oracle as orc


def _case(g, i):
    q = g[f"params{i}"]
    model = str(g[f"model{i}"])
    p = orc.RFParams(q[0], q[1], q[2], q[3], q[4], q[5], q[6], model, bool(q[8]), float(q[7]) if q[7] else None)
    return p, (int(q[9]), int(q[10])), int(q[11])


def test_oracle_reproduces_f3b_bitwise(golden_dir):
    g = np.load(golden_dir / "f3b_spectral_blocks.npz")
    for i in range(int(g["n_cases"])):
        p, shape, seed = _case(g, i)
        d = orc.spectral_draws(np.random.default_rng(seed=seed), p, shape)
        assert np.array_equal(np.array([d["scale"], d["nug"], d["range_x"], d["range_y"]]), g[f"scalars{i}"])
        assert np.array_equal(orc.spectral_from_draws(d, p, shape, float(g["resolution"])), g[f"field{i}"])


def test_hermitian_half_plane_identity(golden_dir):
    """What the device computes, in NumPy: X[k] = amp(k) ((N1[k] + N1[-k])/2 + i (N2[k] - N2[-k])/2) on kx <= bw/2, then a
    complex-to-real inverse DFT -- equals Re(ifft2((N1 + i N2) amp)) of the reference to rounding."""
    g = np.load(golden_dir / "f3b_spectral_blocks.npz")
    for i in range(int(g["n_cases"])):
        p, (bh, bw), seed = _case(g, i)
        d = orc.spectral_draws(np.random.default_rng(seed=seed), p, (bh, bw))
        amp = orc.spectral_amplitude((bh, bw), float(g["resolution"]), p.model_name, d["range_x"], d["range_y"], p.smoothness)
        neg = lambda a: np.roll(a[::-1, ::-1], (1, 1), axis=(0, 1))          # a[-ky, -kx]
        X = amp * (0.5 * (d["n_re"] + neg(d["n_re"])) + 0.5j * (d["n_im"] - neg(d["n_im"])))
        fld = np.fft.irfft2(X[:, : bw // 2 + 1], s=(bh, bw))
        fld = (fld - fld.mean()) / (fld.std() + 1e-12) * d["scale"] + d["n_nug"]
        np.testing.assert_allclose(fld, g[f"field{i}"], rtol=0, atol=1e-12 * d["scale"])


@pytest.mark.gpu
def test_device_spectral_synthesis_equals_reference_on_blocks_50_80(golden_dir):
    from gpu_common import make_engine
    g = np.load(golden_dir / "f3b_spectral_blocks.npz")
    eng, prob, cfg, pairs, masks, _ = make_engine(256, 1)
    res = float(g["resolution"])
    assert res == prob["resolution"]
    for i in range(int(g["n_cases"])):
        p, (bh, bw), seed = _case(g, i)
        idx = int(np.flatnonzero((pairs[0] == bw) & (pairs[1] == bh))[0])
        d = orc.spectral_draws(np.random.default_rng(seed=seed), p, (bh, bw))
        p.resolution = res
        out = eng.spectral_from_noise([idx], [[d["scale"], d["nug"], d["range_x"], d["range_y"]]], p, [d["n_re"]], [d["n_im"]],
                                      [d["n_nug"]] if p.nugget_max > 0 else None)[0]
        exp = g[f"field{i}"] * masks[idx]
        err = np.abs(out - exp).max()
        assert err <= 1e-12 * d["scale"], f"case {i} ({p.model_name}, {bh}x{bw}): max error {err:.3e}, scale {d['scale']:.3f}"
        assert np.abs(exp).max() > 0.5 * d["scale"]          # not a comparison of zeros
    eng.close()


@pytest.mark.gpu
def test_device_spectral_synthesis_equals_golden_f3(golden_dir):
    """The first five proposals of the standard 64x64 chain (seed 7): RandField.rng draws the size index, then the
    spectral draws (MCMC.py:755, :200-251); golden F3 holds the reference's masked fields."""
    from gpu_common import make_engine
    g = np.load(golden_dir / "f3_fields64.npz")
    eng, prob, cfg, pairs, masks, rfp = make_engine(64, 1)
    rng = np.random.default_rng(seed=7)
    rfp.resolution = prob["resolution"]
    idxs, scal, n1, n2 = [], [], [], []
    for i in range(5):
        idx = int(rng.integers(low=0, high=pairs.shape[1], size=1)[0])
        bw, bh = int(pairs[0, idx]), int(pairs[1, idx])
        d = orc.spectral_draws(rng, rfp, (bh, bw))
        idxs.append(idx); scal.append([d["scale"], d["nug"], d["range_x"], d["range_y"]]); n1.append(d["n_re"]); n2.append(d["n_im"])
    outs = eng.spectral_from_noise(idxs, scal, rfp, n1, n2, None)
    for i in range(5):
        exp = g[f"field{i}"]
        assert outs[i].shape == exp.shape
        np.testing.assert_allclose(outs[i], exp, rtol=0, atol=1e-12 * scal[i][0])
    eng.close()
