"""-m gpu: the reference-compatible Python surface (mcmc_gpu_amd.MCMC_gpu / driver / Topography) on the device,
against golden fixtures generated from the reference itself and against the oracle."""
import json
from copy import deepcopy

import numpy as np
import pytest

import mcmc_oracle as orc
from mcmc_gpu_amd import MCMC_gpu, Topography, driver, synthetic

pytestmark = pytest.mark.gpu
RTOL = 1e-10


def _rehydrate(ch, rf, seed, bed):
    cp = deepcopy(ch.__dict__); cp["rng_seed"] = seed; cp["initial_bed"] = bed
    rp = deepcopy(rf.__dict__); rp["rng_seed"] = seed
    return MCMC_gpu.init_lsc_chain_by_instance(cp), MCMC_gpu.initiate_RF_by_instance(rp), cp, rp


def test_run_replay_equals_reference_chain(golden_dir):
    """chain_crf_gpu.run on seeds of fixture F1 (reference chain_crf.run, 300 iterations)."""
    g = np.load(golden_dir / "f1_chain64_standard.npz")
    prob, ch, rf = synthetic.template(64)
    c, r, _, _ = _rehydrate(ch, rf, 7, prob["bed"].copy())
    c.replay_chunk = 64
    out = c.run(300, r, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    bed, loss_mc, loss_data, loss, steps, resampled, blocks = out
    assert np.array_equal(bed, g["bed"])
    assert np.array_equal(steps, g["steps"])
    assert np.array_equal(blocks, g["blocks"], equal_nan=True)
    assert np.array_equal(resampled, g["resampled"])
    np.testing.assert_allclose(loss, g["loss"], rtol=RTOL)
    assert np.array_equal(loss_mc, loss) and not loss_data.any()
    assert all(isinstance(a, np.ndarray) and a.dtype == np.float64 for a in out)
    # generators advanced exactly as in the reference
    o = orc.run_standard_chain(64, 300, record=False)
    assert np.array_equal(o[0], bed)


def test_run_keeps_every_bed_and_sample_points():
    prob, ch, rf = synthetic.template(64)
    c, r, _, _ = _rehydrate(ch, rf, 9, prob["bed"].copy())
    loc = np.array([[prob["xx"][20, 30], prob["yy"][20, 30]], [prob["xx"][40, 12], prob["yy"][40, 12]]])
    c.set_sample_points_locations(loc)
    out = c.run(25, r, only_save_last_bed=False, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert len(out) == 8
    bed_cache, *_rest, sample_values = out
    assert bed_cache.shape == (25, 64, 64) and sample_values.shape == (2, 25)
    assert np.array_equal(bed_cache[0], prob["bed"])
    assert np.array_equal(sample_values[0], bed_cache[:, 20, 30]) and np.array_equal(sample_values[1], bed_cache[:, 40, 12])
    # same chain through the oracle with every bed kept
    p2, cfg, pairs, masks, rfp = orc.standard_setup(64)
    o = orc.run_chain(cfg, prob["bed"].copy(), 25, orc.OracleRandField(rfp, 9, pairs, masks, 500.0),
                      np.random.default_rng(9), only_save_last_bed=False)
    assert np.array_equal(bed_cache, o[0])
    assert np.array_equal(out[4], o[4])


def test_wrapper_two_segments_files_equal_reference(golden_dir, tmp_path):
    """lsc_run_wrapper twice (resume from files) == the reference wrapper's files and contents (fixture F7)."""
    g = np.load(golden_dir / "f7_wrapper_two_segments.npz")
    seed = int(g["seed"])
    prob, ch, rf = synthetic.template(64)
    outdir = tmp_path / "LargeScaleChain"
    (outdir / str(seed)[:6]).mkdir(parents=True)
    for _ in range(2):
        cp = deepcopy(ch.__dict__); cp["rng_seed"] = seed; cp["initial_bed"] = prob["bed"].copy()
        rp = deepcopy(rf.__dict__); rp["rng_seed"] = seed
        runp = dict(n_iter=1000, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=False,
                    chain_id=0, tqdm_position=1, seed=seed, output_path=str(outdir))
        driver.lsc_run_wrapper(cp, rp, runp)
    folder = outdir / str(seed)[:6]
    files = sorted(p.name for p in folder.iterdir())
    assert [f for f in files if f != "RNGState_philox.txt"] == sorted(g["files"].tolist())
    assert np.array_equal(np.load(folder / "bed_2k.npy"), g["bed_2k"])
    with np.load(folder / "results_2k.npz") as r:
        assert sorted(r.files) == ["blocks_used", "loss", "loss_data", "loss_mc", "resampled_times", "steps"]
        assert np.array_equal(r["steps"], g["res_steps"])
        assert np.array_equal(r["blocks_used"], g["res_blocks_used"], equal_nan=True)
        assert np.array_equal(r["resampled_times"], g["res_resampled_times"])
        np.testing.assert_allclose(r["loss"], g["res_loss"], rtol=RTOL)
        np.testing.assert_allclose(r["loss_mc"], g["res_loss_mc"], rtol=RTOL)
    assert int(np.loadtxt(folder / "current_iter.txt")) == 2000
    assert json.load(open(folder / "RNGState_chain.txt")) == json.loads(str(g["rng_state_chain"]))
    assert json.load(open(folder / "RNGState_RandField.txt")) == json.loads(str(g["rng_state_randfield"]))


def test_philox_accept_rate_parity_with_cpu_path():
    """Philox-mode chains vs the oracle's (reference-identical) CPU chains on the same problem: pooled accept
    rate within 0.02 and mean loss trajectory within MC error.  Different random numbers, same sampler."""
    prob, ch, rf = synthetic.template(64)
    n_chains, n_iter = 48, 401
    beds = synthetic.initial_beds(prob, n_chains)
    res = MCMC_gpu.run_many(ch, rf, beds, [1000 + i for i in range(n_chains)], n_iter, batch=16)
    acc_gpu = np.mean([r[4][1:].mean() for r in res])
    loss_gpu = np.mean([r[3][-1] for r in res])
    p2, cfg, pairs, masks, rfp = orc.standard_setup(64)
    acc_cpu, loss_cpu = [], []
    for c in range(12):
        o = orc.run_chain(cfg, orc.chain_initial_bed(p2, c), n_iter, orc.OracleRandField(rfp, 50 + c, pairs, masks, 500.0),
                          np.random.default_rng(50 + c))
        acc_cpu.append(o[4][1:].mean()); loss_cpu.append(o[3][-1])
    assert abs(acc_gpu - np.mean(acc_cpu)) < 0.02, (acc_gpu, np.mean(acc_cpu))
    # chain 0 starts from the unperturbed bed (lower loss): compare the perturbed-start chains
    lg = np.mean([r[3][-1] for r in res[1:13]]); lc = np.mean(loss_cpu[1:])
    assert abs(lg - lc) < 0.05 * lc, (lg, lc)
    # run_many result layout == chain.run(only_save_last_bed=True)
    bed, loss_mc, loss_data, loss, steps, resampled, blocks = res[3]
    assert bed.shape == (64, 64) and loss.shape == (n_iter,) and blocks.shape == (n_iter, 4)
    assert steps[0] == 0 and np.isnan(blocks[0]).all() and not np.isnan(blocks[1:]).any()
    assert resampled.sum() > 0 and set(np.unique(steps)) <= {0.0, 1.0}


def test_philox_run_is_reproducible_and_segmentable():
    prob, ch, rf = synthetic.template(64)
    beds = synthetic.initial_beds(prob, 3)
    a = MCMC_gpu.run_many(ch, rf, beds, [5, 6, 7], 101, batch=8)
    b = MCMC_gpu.run_many(ch, rf, beds, [5, 6, 7], 101, batch=32)
    for ra, rb in zip(a, b):
        assert all(np.array_equal(x, y, equal_nan=True) for x, y in zip(ra, rb))
    # two segments of 51 iterations (50 proposals each) == one of 101
    s1 = MCMC_gpu.run_many(ch, rf, beds, [5, 6, 7], 51, batch=8, step0=0)
    s2 = MCMC_gpu.run_many(ch, rf, np.stack([r[0] for r in s1]), [5, 6, 7], 51, batch=8, step0=50)
    for c in range(3):
        assert np.array_equal(s2[c][0], a[c][0])
        assert np.array_equal(np.concatenate([s1[c][4], s2[c][4][1:]]), a[c][4])


def test_largeScaleChain_mp_philox_writes_reference_layout(tmp_path):
    prob, ch, rf = synthetic.template(64)
    ch.set_rng_mode("philox")
    seeds = [111111, 222222, 333333]
    beds = list(synthetic.initial_beds(prob, 3))
    res = driver.largeScaleChain_mp(3, 7, ch, rf, beds, seeds, [1000] * 3, output_path=str(tmp_path))
    res2 = driver.largeScaleChain_mp(3, 7, ch, rf, beds, seeds, [1000] * 3, output_path=str(tmp_path))
    assert len(res) == 3 and len(res[0]) == 7
    for s, r2 in zip(seeds, res2):
        folder = tmp_path / "LargeScaleChain" / str(s)[:6]
        assert int(np.loadtxt(folder / "current_iter.txt")) == 2000
        assert np.array_equal(np.load(folder / "bed_2k.npy"), r2[0])
        with np.load(folder / "results_2k.npz") as r:
            assert r["loss"].shape == (2000,) and r["blocks_used"].shape == (2000, 4)
        st = json.load(open(folder / "RNGState_philox.txt"))
        assert st == {"key": s, "step": 1998}
    # one unsplit philox run of the same total length gives the same final beds
    one = MCMC_gpu.run_many(ch, rf, np.stack(beds), seeds, 1999, batch=8)
    for a, b in zip(one, res2):
        assert np.array_equal(a[0], b[0])


def test_topography_residual_on_device(golden_dir):
    g = np.load(golden_dir / "f5_residual.npz")
    r = Topography.get_mass_conservation_residual(g["bed"], g["surf"], g["velx"], g["vely"], g["dhdt"], g["smb"],
                                                  float(g["resolution"]))
    assert isinstance(r, np.ndarray) and np.array_equal(r, g["residual"], equal_nan=True)
    import torch
    t = Topography.get_mass_conservation_residual_tensor(torch.tensor(g["bed"]), torch.tensor(g["surf"]),
                                                         torch.tensor(g["velx"]), torch.tensor(g["vely"]),
                                                         torch.tensor(g["dhdt"]), torch.tensor(g["smb"]),
                                                         float(g["resolution"]))
    assert t.is_cuda and np.array_equal(t.cpu().numpy(), g["residual"], equal_nan=True)


def test_min_dist_device_equals_kdtree():
    """gsm_min_dist_from_mask == scipy KD-tree distances (what Utilities.min_dist_from_mask returns), bit for bit,
    and the crf weight built from it equals the oracle's (fixture F4 pins that to the reference)."""
    prob, ch, rf = synthetic.template(64)
    d_dev = MCMC_gpu.min_dist_from_mask(prob["xx"], prob["yy"], prob["data_mask"], device=True)
    d_host = MCMC_gpu.min_dist_from_mask(prob["xx"], prob["yy"], prob["data_mask"], device=False)
    assert np.array_equal(d_dev, d_host)
    g = np.random.default_rng(0)
    mask = g.random((96, 130)) < 0.01
    xx, yy = np.meshgrid(np.arange(130) * 437.5 + 1e6, np.arange(96) * 437.5 - 2e5)
    assert np.array_equal(MCMC_gpu.min_dist_from_mask(xx, yy, mask, True), MCMC_gpu.min_dist_from_mask(xx, yy, mask, False))
    p2, cfg, *_ = orc.standard_setup(64)
    assert np.array_equal(ch.crf_data_weight, cfg.crf_data_weight)     # template() ran the device path on this box
    with pytest.raises(Exception):
        MCMC_gpu.min_dist_from_mask(xx, yy, np.zeros_like(mask), True)


def test_largeScaleChain_mp_replay_four_chains_equals_cpu_pool(tmp_path):
    """BASELINE configs[0] plumbing: 64x64 grid, 4 chains through largeScaleChain_mp in replay mode == what the
    reference's process pool computes on the CPU (oracle chains on the same seeds), incl. the checkpoint files."""
    prob, ch, rf = synthetic.template(64)
    seeds = [101, 202, 303, 404]
    beds = list(synthetic.initial_beds(prob, 4))
    res = driver.largeScaleChain_mp(4, 7, ch, rf, beds, seeds, [200] * 4, output_path=str(tmp_path))
    p2, cfg, pairs, masks, rfp = orc.standard_setup(64)
    for i, s in enumerate(seeds):
        o = orc.run_chain(cfg, beds[i].copy(), 200, orc.OracleRandField(rfp, s, pairs, masks, 500.0), np.random.default_rng(s))
        assert np.array_equal(res[i][0], o[0]) and np.array_equal(res[i][4], o[4])
        np.testing.assert_allclose(res[i][3], o[3], rtol=RTOL)
        assert np.array_equal(res[i][6], o[6], equal_nan=True) and np.array_equal(res[i][5], o[5])
        folder = tmp_path / "LargeScaleChain" / str(s)[:6]
        assert sorted(p.name for p in folder.iterdir()) == ["RNGState_RandField.txt", "RNGState_chain.txt",
                                                            "RNGState_philox.txt", "bed_0k.npy", "current_iter.txt",
                                                            "results_0k.npz"]
        assert np.array_equal(np.load(folder / "bed_0k.npy"), o[0])


def test_ieee_division_path_when_resolution_has_all_ones_significand():
    """exact_div is switched off by the host for a divisor whose significand is all ones; the IEEE-division
    instantiation of the step kernel must give the same parity."""
    from mcmc_gpu_amd.engine import GsmEngine
    res = float(np.nextafter(512.0, 0.0))          # 0x1.fffffffffffffp+8
    prob, cfg, pairs, masks, rfp = orc.standard_setup(64)
    cfg.resolution = res
    masks = orc.edge_masks(pairs, [2, 0, 6, 1], 49900.0, res)
    eng = GsmEngine(64, 64, 1)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.region_mask,
                   cfg.mc_region_mask, res, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    rf = orc.OracleRandField(rfp, 13, pairs, masks, res)
    out = orc.run_chain(cfg, prob["bed"].copy(), 150, rf, np.random.default_rng(13), record=True)
    tr = out[7]
    eng.set_state(prob["bed"][None])
    loss, acc = eng.run_replay(np.array([tr.size_idx]), np.array([tr.centre]), np.array([tr.u]), eng.pack_fields([tr.fields]))
    assert np.array_equal(acc[0], out[4][1:].astype(np.uint8))
    np.testing.assert_allclose(loss[0], out[3][1:], rtol=RTOL)
    assert np.array_equal(eng.beds[0].cpu().numpy(), out[0])
    eng.close()


def test_philox_single_chain_run_equals_batched_run(tmp_path):
    """chain.run(rng_mode='philox') (one chain per handle, the per-chain wrapper path taken when the chains of a
    launch have different lengths) gives the same chain as the batched run_many on the same key and counters."""
    prob, ch, rf = synthetic.template(64)
    ch.set_rng_mode("philox", philox_batch=8)
    seeds = [515151, 626262]
    beds = list(synthetic.initial_beds(prob, 2))
    res = driver.largeScaleChain_mp(2, 7, ch, rf, beds, seeds, [120, 90], output_path=str(tmp_path))
    many = MCMC_gpu.run_many(ch, rf, np.stack(beds), seeds, 120, batch=8)
    assert np.array_equal(res[0][0], many[0][0]) and np.array_equal(res[0][4], many[0][4])
    np.testing.assert_array_equal(res[0][3], many[0][3])
    # the shorter chain is a prefix of the same Philox sequence
    assert np.array_equal(res[1][4], many[1][4][:90]) and np.array_equal(res[1][6], many[1][6][:90], equal_nan=True)
    st = json.load(open(tmp_path / "LargeScaleChain" / "626262" / "RNGState_philox.txt"))
    assert st == {"key": 626262, "step": 89}
    # keep every bed in philox mode
    c2, r2, _, _ = _rehydrate(ch, rf, 515151, beds[0].copy())
    c2.set_rng_mode("philox")
    out = c2.run(20, r2, only_save_last_bed=False, plot=False, progress_bar=None)
    assert out[0].shape == (20, 64, 64) and np.array_equal(out[0][-1], MCMC_gpu.run_many(ch, rf, beds[0][None], [515151], 20)[0][0])


def test_highvel_boundary_on_device(golden_dir):
    g = np.load(golden_dir / "f9_highvel_boundary.npz")
    m = Topography.get_highvel_boundary(g["velx"], g["vely"], float(g["threshold"]), g["grounded"], g["ocean"],
                                        float(g["distance_max"]), g["xx"], g["yy"], smooth_mode=int(g["smooth_mode"]))
    assert np.array_equal(m, g["mask_final"])


def test_batched_replay_two_segments_files_equal_reference(golden_dir, tmp_path):
    """largeScaleChain_mp in its default (replay) mode -- all chains of the rank in one handle, draws made by the host
    pool -- twice in a row (resume from the seed folders): the folder of the F7 seed equals the reference wrapper's
    files, and every chain equals its own single-chain lsc_run_wrapper run."""
    g = np.load(golden_dir / "f7_wrapper_two_segments.npz")
    seed = int(g["seed"])
    prob, ch, rf = synthetic.template(64)
    seeds = [seed, 777001, 777002]
    beds = [prob["bed"].copy() for _ in seeds]
    for _ in range(2):
        res = driver.largeScaleChain_mp(3, 3, ch, rf, beds, seeds, [1000] * 3, output_path=str(tmp_path), n_gpus=1)
    folder = tmp_path / "LargeScaleChain" / str(seed)[:6]
    files = sorted(p.name for p in folder.iterdir())
    assert [f for f in files if f != "RNGState_philox.txt"] == sorted(g["files"].tolist())
    assert np.array_equal(np.load(folder / "bed_2k.npy"), g["bed_2k"])
    with np.load(folder / "results_2k.npz") as r:
        assert np.array_equal(r["steps"], g["res_steps"])
        assert np.array_equal(r["blocks_used"], g["res_blocks_used"], equal_nan=True)
        assert np.array_equal(r["resampled_times"], g["res_resampled_times"])
        np.testing.assert_allclose(r["loss"], g["res_loss"], rtol=RTOL)
    assert int(np.loadtxt(folder / "current_iter.txt")) == 2000
    assert json.load(open(folder / "RNGState_chain.txt")) == json.loads(str(g["rng_state_chain"]))
    assert json.load(open(folder / "RNGState_RandField.txt")) == json.loads(str(g["rng_state_randfield"]))
    # one of the other chains against the single-chain wrapper, second segment included
    alone = tmp_path / "alone" / "LargeScaleChain"
    (alone / "777002").mkdir(parents=True)
    for _ in range(2):
        cp = deepcopy(ch.__dict__); cp["rng_seed"] = 777002; cp["initial_bed"] = prob["bed"].copy()
        rp = deepcopy(rf.__dict__); rp["rng_seed"] = 777002
        out = driver.lsc_run_wrapper(cp, rp, dict(n_iter=1000, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False,
                                                  progress_bar=False, chain_id=2, tqdm_position=1, seed=777002, output_path=str(alone)))
    for a, b in zip(res[2], out):
        assert np.array_equal(a, b, equal_nan=True)
    assert np.array_equal(np.load(tmp_path / "LargeScaleChain" / "777002" / "bed_2k.npy"), np.load(alone / "777002" / "bed_2k.npy"))


def test_largeScaleChain_mp_starts_its_own_ranks(tmp_path, monkeypatch):
    """n_gpus=2 without a launcher: the driver spawns two ranks itself (here they share the box's one GPU, gloo instead of
    RCCL), shards the chains, gathers -- same results as the single-process run, in both draw modes."""
    monkeypatch.setenv("GSM_DIST_BACKEND", "gloo")
    prob, ch, rf = synthetic.template(64)
    seeds = [31, 32, 33, 34, 35]
    beds = list(synthetic.initial_beds(prob, 5))
    for mode in ("philox", "replay"):
        one = driver.largeScaleChain_mp(5, 2, ch, rf, beds, seeds, [120] * 5, output_path=str(tmp_path / f"one_{mode}"), mode=mode, n_gpus=1)
        two = driver.largeScaleChain_mp(5, 2, ch, rf, beds, seeds, [120] * 5, output_path=str(tmp_path / f"two_{mode}"), mode=mode, n_gpus=2)
        assert len(two) == 5
        for a, b in zip(one, two):
            for x, y in zip(a, b):
                assert np.array_equal(np.asarray(x, dtype=float), np.asarray(y, dtype=float), equal_nan=True)
        for s in seeds:
            assert np.array_equal(np.load(tmp_path / f"one_{mode}" / "LargeScaleChain" / str(s) / "bed_0k.npy"),
                                  np.load(tmp_path / f"two_{mode}" / "LargeScaleChain" / str(s) / "bed_0k.npy"))
