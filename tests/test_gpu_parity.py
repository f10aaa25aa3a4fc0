"""-m gpu: the HIP path, through the C ABI, against the oracle on identical draws (replay mode).

Bars: accept masks, block records, resampled counts and final beds bit-exact; loss within 1e-10
relative (north_star: "log-likelihood within 1e-10 fp64") -- the kernel sums the window's squared
residuals in a different order than numpy's full-grid pairwise nansum."""
import numpy as np
import pytest

import mcmc_oracle as orc
from gpu_common import make_engine, oracle_chains, replay_inputs

pytestmark = pytest.mark.gpu
LOSS_RTOL = 1e-10


def _check(eng, outs, loss0, loss, acc):
    for c, o in enumerate(outs):
        assert abs(loss0[c] - o[3][0]) <= LOSS_RTOL * abs(o[3][0])
        assert np.array_equal(acc[c], o[4][1:].astype(np.uint8)), f"accept mask differs, chain {c}"
        np.testing.assert_allclose(loss[c], o[3][1:], rtol=LOSS_RTOL, atol=0)
        assert np.array_equal(eng.beds[c].cpu().numpy(), o[0]), f"final bed differs, chain {c}"
        assert np.array_equal(eng.resampled[c].cpu().numpy().astype(np.float64), o[5])


def test_replay_standard_64(golden_dir):
    eng, prob, cfg, pairs, masks, rfp = make_engine(64, 4)
    outs = oracle_chains(prob, cfg, pairs, masks, rfp, 4, 300)
    loss0 = eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(4)]))
    loss, acc = eng.run_replay(*replay_inputs(eng, outs))
    _check(eng, outs, loss0, loss, acc)
    # chain 0 is golden fixture F1 (pinned to the reference itself)
    g = np.load(golden_dir / "f1_chain64_standard.npz")
    assert np.array_equal(acc[0], g["steps"][1:].astype(np.uint8))
    np.testing.assert_allclose(loss[0], g["loss"][1:], rtol=LOSS_RTOL)
    assert np.array_equal(eng.beds[0].cpu().numpy(), g["bed"])
    assert np.array_equal(eng.resampled[0].cpu().numpy(), g["resampled"].astype(np.int32))
    eng.close()


def test_replay_variant_rf_whole_map_nugget(golden_dir):
    """block_type 'RF', update_in_region False (blocks clipped at the grid edges), anisotropic Gaussian
    model with nugget: golden fixture F2."""
    rfp = orc.RFParams(8e3, 30e3, 12e3, 40e3, 30, 90, 4.0, "Gaussian", False, None)
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 3, block_type="RF", update_in_region=False, rf_params=rfp)
    outs = oracle_chains(prob, cfg, pairs, masks, rfp, 3, 300, seed0=7)
    loss0 = eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(3)]))
    loss, acc = eng.run_replay(*replay_inputs(eng, outs))
    _check(eng, outs, loss0, loss, acc)
    g = np.load(golden_dir / "f2_chain64_variant.npz")   # chain_index 1, seed 8
    assert np.array_equal(acc[1], g["steps"][1:].astype(np.uint8))
    assert np.array_equal(eng.beds[1].cpu().numpy(), g["bed"])
    # some windows were clipped by the grid edge
    b = outs[1][6][1:]
    assert ((b[:, 0] - b[:, 2] / 2 < 0) | (b[:, 1] - b[:, 3] / 2 < 0)).any()
    eng.close()


def test_replay_256_full_size_blocks(golden_dir):
    """256x256 grid, 50-80 cell blocks (the headline geometry): golden fixture F8 + 2 more chains."""
    eng, prob, cfg, pairs, masks, rfp = make_engine(256, 3)
    outs = oracle_chains(prob, cfg, pairs, masks, rfp, 3, 120)
    loss0 = eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(3)]))
    loss, acc = eng.run_replay(*replay_inputs(eng, outs))
    _check(eng, outs, loss0, loss, acc)
    g = np.load(golden_dir / "f8_chain256_anchor.npz")
    assert np.array_equal(acc[0], g["steps"][1:].astype(np.uint8))
    np.testing.assert_allclose(loss[0], g["loss"][1:], rtol=LOSS_RTOL)
    assert np.array_equal(eng.beds[0].cpu().numpy()[128], g["bed_row128"])
    eng.close()


def test_replay_in_segments_equals_one_call():
    """Two replay calls of 100 steps == one call of 200 steps (state carried in beds/resampled/loss_sum)."""
    eng, prob, cfg, pairs, masks, rfp = make_engine(64, 2)
    outs = oracle_chains(prob, cfg, pairs, masks, rfp, 2, 201)
    si, ce, u, fl = replay_inputs(eng, outs)
    eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(2)]))
    l1, a1 = eng.run_replay(si[:, :100], ce[:, :100], u[:, :100], fl[:, :100])
    l2, a2 = eng.run_replay(si[:, 100:], ce[:, 100:], u[:, 100:], fl[:, 100:])
    for c, o in enumerate(outs):
        assert np.array_equal(np.concatenate([a1[c], a2[c]]), o[4][1:].astype(np.uint8))
        np.testing.assert_allclose(np.concatenate([l1[c], l2[c]]), o[3][1:], rtol=LOSS_RTOL)
        assert np.array_equal(eng.beds[c].cpu().numpy(), o[0])
    eng.close()


def test_thickness_guard_rejects():
    """A proposal that lifts the bed to/above the surface gets loss = inf and is rejected even when the
    residual loss does not change (MCMC.py:1321-1329).  Zero velocities make the loss independent of the bed,
    so only the guard can reject (u is tiny)."""
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, pairs, masks, rfp = orc.standard_setup(64)
    cfg.velx = np.zeros_like(cfg.velx)
    cfg.vely = np.zeros_like(cfg.vely)
    cfg.block_type = "RF"
    eng = GsmEngine(64, 64, 1)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, None, cfg.region_mask, cfg.mc_region_mask,
                   cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    bed0 = orc.chain_initial_bed(prob, 0)
    loss0 = eng.set_state(bed0[None])
    bh, bw = int(pairs[1, 0]), int(pairs[0, 0])
    thick = (cfg.surf - bed0)[32, 32]
    f_ok = np.zeros((bh, bw)); f_ok[bh // 2, bw // 2] = thick - 1e-9     # 1 nm of ice left: allowed
    f_bad = np.zeros((bh, bw)); f_bad[bh // 2, bw // 2] = thick          # thickness == 0: guard
    fields = eng.pack_fields([[f_bad, f_ok, f_bad]])
    u = np.array([[1e-300, 1e-300, 0.999]])
    loss, acc = eng.run_replay(np.zeros((1, 3), int), np.array([[[32, 32]] * 3]), u, fields)
    # oracle on the same three steps
    mc = orc.mc_residual(bed0, cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.resolution)
    lp = orc.gaussian_loss(mc, cfg.mc_region_mask, cfg.sigma_mc)[0]
    bed, exp_acc = bed0, []
    for f, uu in zip((f_bad, f_ok, f_bad), u[0]):
        a, bed, mc, lp, _ = orc.mh_step(cfg, bed, mc, lp, f, 32, 32, uu)
        exp_acc.append(a)
    assert exp_acc == [False, True, False]
    assert acc[0].tolist() == [0, 1, 0]
    assert np.array_equal(eng.beds[0].cpu().numpy(), bed)
    np.testing.assert_allclose(loss[0], [lp] * 3, rtol=1e-12)
    eng.close()


def test_residual_and_init_loss_with_nans(golden_dir):
    """Full-grid residual kernel vs fixture F5 (NaNs in bed and velx) and nansum semantics of the loss."""
    from mcmc_gpu_amd.engine import GsmEngine
    g = np.load(golden_dir / "f5_residual.npz")
    H, W = g["bed"].shape
    eng = GsmEngine(H, W, 2)
    ones = np.ones((H, W), dtype=np.uint8)
    eng.set_static(g["surf"], g["velx"], g["vely"], g["dhdt"], g["smb"], None, ones, ones, float(g["resolution"]), 3.0)
    beds = np.stack([g["bed"], g["bed"] + 1.5])
    r = eng.residual(beds).cpu().numpy()
    assert np.array_equal(r[0], g["residual"], equal_nan=True)
    exp1 = orc.mc_residual(beds[1], g["surf"], g["velx"], g["vely"], g["dhdt"], g["smb"], float(g["resolution"]))
    assert np.array_equal(r[1], exp1, equal_nan=True)
    loss0 = eng.set_state(beds)
    for c, rr in enumerate((g["residual"], exp1)):
        exp = orc.gaussian_loss(rr, ones, 3.0)[0]
        assert abs(loss0[c] - exp) <= 1e-12 * abs(exp)
    eng.close()


def test_bad_device_data_is_reported():
    from mcmc_gpu_amd._lib import GsmError
    eng, prob, cfg, pairs, masks, rfp = make_engine(64, 1)
    eng.set_state(orc.chain_initial_bed(prob, 0)[None])
    fields = eng.pack_fields([[np.zeros((int(pairs[1, 0]), int(pairs[0, 0])))]])
    with pytest.raises(ValueError):
        eng.run_replay(np.array([[99]]), np.array([[[3, 3]]]), np.array([[0.5]]), fields)
    with pytest.raises(ValueError):
        eng.run_replay(np.array([[0]]), np.array([[[64, 3]]]), np.array([[0.5]]), fields)
    eng.close()


def test_f32_state_mode_matches_its_oracle_and_tracks_fp64():
    """dtype 1: bed and energies stored as float32, arithmetic in fp64 (BASELINE configs[4]).  Bit-exact accept
    masks and beds against the oracle's emulation of the same rounding; against the fp64 chain the accept masks agree
    and the loss differs by float32 storage noise only (the reference's own all-fp32 twin: <= 2.8e-7)."""
    eng, prob, cfg, pairs, masks, rfp = make_engine(64, 3, state_dtype="f32")
    outs32 = oracle_chains(prob, cfg, pairs, masks, rfp, 3, 300, state_f32=True)
    loss0 = eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(3)]))
    loss, acc = eng.run_replay(*replay_inputs(eng, outs32))
    assert eng.beds.dtype.itemsize == 4 and eng.energy.dtype.itemsize == 4
    for c, o in enumerate(outs32):
        assert abs(loss0[c] - o[3][0]) <= 1e-10 * abs(o[3][0])
        assert np.array_equal(acc[c], o[4][1:].astype(np.uint8))
        np.testing.assert_allclose(loss[c], o[3][1:], rtol=1e-10)
        assert np.array_equal(eng.beds[c].cpu().numpy().astype(np.float64), o[0])
    outs64 = oracle_chains(prob, cfg, pairs, masks, rfp, 3, 300, state_f32=False)
    for c in range(3):
        assert (outs64[c][4] != outs32[c][4]).sum() <= 1
        np.testing.assert_allclose(outs32[c][3][:50], outs64[c][3][:50], rtol=1e-6)
    cfg.state_f32 = False
    eng.close()


def test_nan_cells_follow_nansum_semantics():
    """Real grids carry NaNs.  NaN residuals are ignored by the loss (nansum, MCMC.py:1041), NaN thickness never trips
    the guard (`nan <= 0` is False, MCMC.py:1328): a bed with NaN holes and a NaN velocity patch must replay exactly."""
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, pairs, masks, rfp = orc.standard_setup(64)
    cfg.velx = cfg.velx.copy(); cfg.velx[30:33, 40:44] = np.nan
    cfg.dhdt = cfg.dhdt.copy(); cfg.dhdt[12, 12] = np.nan
    bed0 = orc.chain_initial_bed(prob, 0)
    bed0[20:22, 20:25] = np.nan
    bed0[45, 50] = np.nan
    eng = GsmEngine(64, 64, 1)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.region_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    rf = orc.OracleRandField(rfp, 21, pairs, masks, 500.0)
    out = orc.run_chain(cfg, bed0.copy(), 400, rf, np.random.default_rng(21), record=True)
    tr = out[7]
    loss0 = eng.set_state(bed0[None])
    loss, acc = eng.run_replay(np.array([tr.size_idx]), np.array([tr.centre]), np.array([tr.u]), eng.pack_fields([tr.fields]))
    assert np.isfinite(out[3]).all() and 0.3 < out[4].mean() < 1.0
    assert abs(loss0[0] - out[3][0]) <= LOSS_RTOL * out[3][0]
    assert np.array_equal(acc[0], out[4][1:].astype(np.uint8))
    np.testing.assert_allclose(loss[0], out[3][1:], rtol=LOSS_RTOL)
    assert np.array_equal(eng.beds[0].cpu().numpy(), out[0], equal_nan=True)
    assert np.isnan(eng.beds[0].cpu().numpy()).sum() >= 11
    eng.close()


def test_tiny_grid():
    """8 x 10 grid with 2-4 cell blocks: every window touches an edge or sits one cell from it."""
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, _, _, rfp = orc.standard_setup(8, 10, block_min=2, block_max=4, update_in_region=False)
    pairs = orc.block_pairs(2, 4, 2, 4, steps=2)
    masks = orc.edge_masks(pairs, [2, 0, 6, 1], 49900.0, 500.0)
    # a 2-cell block has only border cells: its taper is all zeros -> give the masks some interior weight instead
    masks = [np.ones_like(m) * 0.25 for m in masks]
    eng = GsmEngine(8, 10, 1)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.grounded_ice_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    rf = orc.OracleRandField(rfp, 4, pairs, masks, 500.0)
    out = orc.run_chain(cfg, prob["bed"].copy(), 200, rf, np.random.default_rng(4), record=True)
    tr = out[7]
    eng.set_state(prob["bed"][None])
    loss, acc = eng.run_replay(np.array([tr.size_idx]), np.array([tr.centre]), np.array([tr.u]), eng.pack_fields([tr.fields]))
    # with a non-zero taper on the block border the carried residual of the REFERENCE goes stale outside the window
    # (SURVEY a8 holds only for border-zero masks); the device recomputes nothing outside the window either, and both
    # sum the same carried values, so they still agree
    assert np.array_equal(acc[0], out[4][1:].astype(np.uint8))
    np.testing.assert_allclose(loss[0], out[3][1:], rtol=LOSS_RTOL)
    assert np.array_equal(eng.beds[0].cpu().numpy(), out[0])
    eng.close()
