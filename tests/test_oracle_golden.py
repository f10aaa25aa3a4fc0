"""CPU suite: the oracle (oracle/mcmc_oracle.py) against the committed golden vectors.

The vectors were produced by oracle/make_fixtures.py from the imported reference
(/root/reference, build container only) after asserting bit-identity with the oracle; here the
oracle alone is re-run and must still reproduce them bit-for-bit (integer/index/accept data) and
exactly (fp64 -- same NumPy operation sequence)."""
import hashlib
import json

import numpy as np
import pytest

import mcmc_oracle as orc


def _sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _check_chain(g, out):
    assert np.array_equal(out[0], g["bed"])
    assert np.array_equal(out[3], g["loss"])
    assert np.array_equal(out[4], g["steps"])
    assert np.array_equal(out[5], g["resampled"])
    assert np.array_equal(out[6], g["blocks"], equal_nan=True)


def test_f1_standard_chain(golden_dir):
    g = np.load(golden_dir / "f1_chain64_standard.npz")
    out = orc.run_standard_chain(64, int(g["n_iter"]), chain_index=0, record=True)
    _check_chain(g, out)
    tr = out[7]
    assert np.array_equal(np.array(tr.size_idx), g["size_idx"])
    assert np.array_equal(np.array(tr.centre), g["centre"])
    assert np.array_equal(np.array(tr.u), g["u"])
    assert np.array_equal(np.array(tr.rf_scalars), g["rf_scalars"])
    # SURVEY.md 8c smoke anchors
    assert out[3][:4].tolist() == [3443.491974414064, 3444.826032000513, 3444.998304407799, 3444.9840819946035]
    assert out[6][1:4].tolist() == [[37, 49, 16, 14], [18, 55, 14, 8], [31, 52, 8, 8]]
    f3 = np.load(golden_dir / "f3_fields64.npz")
    for i in range(5):
        assert np.array_equal(tr.fields[i], f3[f"field{i}"])
    assert [_sha(f) for f in tr.fields] == f3["field_sha"].tolist()


def test_f1b_second_chain(golden_dir):
    g = np.load(golden_dir / "f1b_chain64_seed10.npz")
    _check_chain(g, orc.run_standard_chain(64, int(g["n_iter"]), chain_index=int(g["chain_index"])))


def test_f2_variant_chain(golden_dir):
    g = np.load(golden_dir / "f2_chain64_variant.npz")
    rfp = orc.RFParams(8e3, 30e3, 12e3, 40e3, 30, 90, 4.0, "Gaussian", False, None)
    out = orc.run_standard_chain(64, int(g["n_iter"]), chain_index=1, block_type="RF",
                                 update_in_region=False, rf_params=rfp)
    _check_chain(g, out)


def test_f2e_exponential_chain(golden_dir):
    g = np.load(golden_dir / "f2e_chain64_exponential.npz")
    rfp = orc.RFParams(10e3, 50e3, 10e3, 50e3, 50, 150, 0.0, "Exponential", True, None)
    _check_chain(g, orc.run_standard_chain(64, int(g["n_iter"]), chain_index=2, rf_params=rfp))


def test_f4_setup(golden_dir):
    g = np.load(golden_dir / "f4_setup64.npz")
    prob, cfg, pairs, masks, _ = orc.standard_setup(64)
    assert np.array_equal(pairs, g["pairs"])
    assert np.array_equal(cfg.crf_data_weight, g["crf_weight"])
    for i, m in enumerate(masks):
        assert np.array_equal(m, g[f"mask{i}"])
        # the taper is exactly zero on the block border: the carried residual stays exact
        assert (m[0] == 0).all() and (m[-1] == 0).all() and (m[:, 0] == 0).all() and (m[:, -1] == 0).all()
    assert (cfg.crf_data_weight[prob["data_mask"]] == 0).all()


def test_f5_residual(golden_dir):
    g = np.load(golden_dir / "f5_residual.npz")
    r = orc.mc_residual(g["bed"], g["surf"], g["velx"], g["vely"], g["dhdt"], g["smb"], float(g["resolution"]))
    assert np.array_equal(r, g["residual"], equal_nan=True)
    assert np.isnan(r).sum() > 0
    s = (slice(0, 2), slice(0, 2))
    thin = orc.mc_residual(g["bed"][s], g["surf"][s], g["velx"][s], g["vely"][s], g["dhdt"][s], g["smb"][s],
                           float(g["resolution"]))
    assert np.array_equal(thin, g["residual_thin"], equal_nan=True)
    # restated gradient == numpy's
    assert np.array_equal(orc._grad_uniform(g["surf"], 3.0, 0), np.gradient(g["surf"], 3.0, axis=0))
    assert np.array_equal(orc._grad_uniform(g["surf"], 3.0, 1), np.gradient(g["surf"], 3.0, axis=1))


def test_f6_covariance(golden_dir):
    g = np.load(golden_dir / "f6_covariance.npz")
    for vt, extra in (("Exponential", {}), ("Gaussian", {}), ("Spherical", {}), ("Matern", {"s": 0.9125})):
        vario = dict(azimuth=30.0, nugget=0.0, major_range=4000.0, minor_range=2500.0, sill=1.0, vtype=vt, **extra)
        sig = orc.cov_matrix(g["coord"], vario)
        assert np.array_equal(sig, g[f"sigma_{vt.lower()}"])
        if vt != "Spherical":
            L = np.linalg.cholesky(sig + 1e-10 * np.eye(sig.shape[0]))
            np.testing.assert_allclose(L @ g["z"], g[f"draw_{vt.lower()}"], rtol=1e-12, atol=1e-13)


def test_f7_two_segment_semantics(golden_dir):
    """Segment concat semantics of the reference wrapper (largeScaleChain_multiprocessing_GPU.py:213-244):
    2 x run(1000) from the same seed, the second resuming from the first's bed and RNG states."""
    g = np.load(golden_dir / "f7_wrapper_two_segments.npz")
    seed = int(g["seed"])
    prob, cfg, pairs, masks, rfp = orc.standard_setup(64)
    rf = orc.OracleRandField(rfp, seed, pairs, masks, prob["resolution"])
    rng = np.random.default_rng(seed=seed)
    bed = prob["bed"].copy()
    loss, steps, blocks, resampled = [], [], [], np.zeros_like(bed)
    for _ in range(2):
        out = orc.run_chain(cfg, bed, 1000, rf, rng)
        bed = out[0]
        loss.append(out[3]); steps.append(out[4]); blocks.append(out[6]); resampled = resampled + out[5]
    assert np.array_equal(bed, g["bed_2k"])
    assert np.array_equal(np.concatenate(loss), g["res_loss"])
    assert np.array_equal(np.concatenate(steps), g["res_steps"])
    assert np.array_equal(np.vstack(blocks), g["res_blocks_used"], equal_nan=True)
    assert np.array_equal(resampled, g["res_resampled_times"])
    assert int(g["current_iter"]) == 2000
    assert sorted(g["files"].tolist()) == ["RNGState_RandField.txt", "RNGState_chain.txt", "bed_1k.npy",
                                           "bed_2k.npy", "current_iter.txt", "results_2k.npz"]
    assert json.loads(str(g["rng_state_chain"])) == rng.bit_generator.state
    assert json.loads(str(g["rng_state_randfield"])) == rf.rng.bit_generator.state


def test_f8_256_anchor(golden_dir):
    g = np.load(golden_dir / "f8_chain256_anchor.npz")
    out = orc.run_standard_chain(256, int(g["n_iter"]), chain_index=0)
    assert np.array_equal(out[3], g["loss"])
    assert np.array_equal(out[4], g["steps"])
    assert np.array_equal(out[6], g["blocks"], equal_nan=True)
    assert _sha(out[0]) == str(g["bed_sha"])
    assert _sha(out[5]) == str(g["resampled_sha"])
    assert abs(out[3][0] - 4.936735e3) < 1e-2
