"""-m gpu: BASELINE's full size (256x256 grid x 1024 chains, fp64, Philox mode) through size-independent
properties -- the oracle cannot run this many chain-steps, the invariants below need no oracle:

  P1  the carried energy array equals a from-scratch recompute of the final beds, BIT FOR BIT, for every chain
      (the window-only residual update is exact: what the reference's carried mc_res also satisfies)
  P2  the compensated carried sum equals the sum of the carried energy (<= 1e-12 rel), and the last loss_cache entry
      equals the recomputed loss (<= 1e-10 rel, north_star's bound)
  P3  resampled counts == number of accepted proposals covering each cell inside the update mask, exactly
  P4  rejected steps repeat the previous loss exactly; cells outside the update mask never change
  P5  the fused chain kernel and the two-kernel pipeline (another batch size) give identical results on a 64-chain slice,
      and each path is asserted to have run (gsm_last_run_fused)
and non-square / edge-clipping geometry on a smaller ragged grid."""
import numpy as np
import pytest
import torch

import mcmc_oracle as orc
from mcmc_gpu_amd import synthetic

pytestmark = pytest.mark.gpu


def test_full_size_invariants():
    from gpu_common import check_chain_invariants
    H, n_chains, n_steps = 256, 1024, 96
    prob, ch, rf = synthetic.template(H)
    eng = ch._make_engine(rf, n_chains, 0)
    beds0 = synthetic.initial_beds(prob, n_chains)
    loss0 = eng.set_state(beds0)
    d_beds0 = eng.beds.clone()
    seeds = list(range(500, 500 + n_chains))
    loss, acc, blk = eng.run_philox(n_steps, 0, seeds, rf, batch=32)
    assert eng.last_run_fused() == 1
    assert 0.45 < acc.mean() < 0.65
    check_chain_invariants(eng, prob["region_mask"], d_beds0, loss0, loss, acc, blk, (0, 1, 511, 1023))
    eng.close()
    # P5: the two launch structures (fused chain kernel / two-kernel pipeline with another batch size) on a 64-chain slice
    eng = ch._make_engine(rf, 64, 0)
    res = []
    for fused, batch in ((1, 32), (0, 8)):
        eng.set_fused(fused)
        eng.set_state(beds0[:64])
        out = eng.run_philox(n_steps, 0, seeds[:64], rf, batch=batch)
        assert eng.last_run_fused() == fused
        res.append(out + (eng.beds.cpu().numpy().copy(), eng.resampled.cpu().numpy().copy(), eng.energy.cpu().numpy().copy()))
    for x, y in zip(*res):
        assert np.array_equal(x, y)
    assert np.array_equal(res[0][0], loss[:64]) and np.array_equal(res[0][1], acc[:64]) and np.array_equal(res[0][2], blk[:64])
    eng.close()


def test_ragged_grid_all_edges_clipped():
    """48 x 80 grid (H != W), whole-map updates so that blocks hang over all four edges and the corners; replay
    against the oracle on identical draws."""
    from gpu_common import oracle_chains, replay_inputs
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, pairs, masks, rfp = orc.standard_setup(48, 80, block_type="CRF_weight", update_in_region=False)
    eng = GsmEngine(48, 80, 2)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.grounded_ice_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    outs = oracle_chains(prob, cfg, pairs, masks, rfp, 2, 400)
    loss0 = eng.set_state(np.stack([orc.chain_initial_bed(prob, c) for c in range(2)]))
    loss, acc = eng.run_replay(*replay_inputs(eng, outs))
    clipped = set()
    for c, o in enumerate(outs):
        assert abs(loss0[c] - o[3][0]) <= 1e-10 * abs(o[3][0])
        assert np.array_equal(acc[c], o[4][1:].astype(np.uint8))
        np.testing.assert_allclose(loss[c], o[3][1:], rtol=1e-10)
        assert np.array_equal(eng.beds[c].cpu().numpy(), o[0])
        assert np.array_equal(eng.resampled[c].cpu().numpy().astype(float), o[5])
        b = o[6][1:]
        clipped |= {"top"} if (b[:, 0] - b[:, 2] / 2 < 0).any() else set()
        clipped |= {"bottom"} if (b[:, 0] + b[:, 2] / 2 > 48).any() else set()
        clipped |= {"left"} if (b[:, 1] - b[:, 3] / 2 < 0).any() else set()
        clipped |= {"right"} if (b[:, 1] + b[:, 3] / 2 > 80).any() else set()
    assert clipped == {"top", "bottom", "left", "right"}
    eng.close()


def test_block_as_large_as_the_grid_and_single_chain():
    """One 16x16 block size on a 16x16 grid: every window is clipped to the whole map or less."""
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, _, _, rfp = orc.standard_setup(16, 16, block_min=16, block_max=16, update_in_region=False)
    pairs = orc.block_pairs(16, 16, 16, 16, steps=1)
    masks = orc.edge_masks(pairs, [2, 0, 6, 1], 49900.0, 500.0)
    eng = GsmEngine(16, 16, 1)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.grounded_ice_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    rf = orc.OracleRandField(rfp, 3, pairs, masks, 500.0)
    out = orc.run_chain(cfg, prob["bed"].copy(), 120, rf, np.random.default_rng(3), record=True)
    tr = out[7]
    eng.set_state(prob["bed"][None])
    loss, acc = eng.run_replay(np.array([tr.size_idx]), np.array([tr.centre]), np.array([tr.u]), eng.pack_fields([tr.fields]))
    assert np.array_equal(acc[0], out[4][1:].astype(np.uint8))
    np.testing.assert_allclose(loss[0], out[3][1:], rtol=1e-10)
    assert np.array_equal(eng.beds[0].cpu().numpy(), out[0])
    # a block larger than the grid is refused, like odd sizes
    from mcmc_gpu_amd._lib import GsmError
    with pytest.raises(GsmError):
        eng.set_blocks(np.array([[18], [16]]), None)
    with pytest.raises(GsmError):
        eng.set_blocks(np.array([[7], [8]]), None)
    eng.close()


def test_large_blocks_use_the_wider_instantiation():
    """Blocks of 96-110 cells on a 128x128 grid: the step kernel's 12-cells-per-thread instantiation (one workgroup per
    CU, deep load batching) against the oracle in replay mode; in Philox mode the wide instantiation of the proposal kernel
    (32 output tiles per DFT stage) against the Philox restatement, and gsm_run_philox (two-kernel pipeline: the fused kernel
    does not take such tables) == propose + replay.  Tables whose coefficient planes exceed the LDS are refused cleanly."""
    import philox_oracle as po
    from gpu_common import oracle_chains, replay_inputs
    from mcmc_gpu_amd._lib import GsmError
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, pairs, masks, rfp = orc.standard_setup(128, 128, block_min=96, block_max=110)
    assert pairs.max() == 110
    eng = GsmEngine(128, 128, 2)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.region_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    eng.set_centres(cfg.region_mask)
    outs = oracle_chains(prob, cfg, pairs, masks, rfp, 2, 60)
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(2)])
    loss0 = eng.set_state(beds0)
    loss, acc = eng.run_replay(*replay_inputs(eng, outs))
    for c, o in enumerate(outs):
        assert np.array_equal(acc[c], o[4][1:].astype(np.uint8))
        np.testing.assert_allclose(loss[c], o[3][1:], rtol=1e-10)
        assert np.array_equal(eng.beds[c].cpu().numpy(), o[0])
        assert np.array_equal(eng.resampled[c].cpu().numpy().astype(float), o[5])
    # Philox mode on the same table
    rfp.resolution = prob["resolution"]
    seeds = [5, 6]
    p = eng.propose_philox(12, 40, seeds, rfp)
    centres = np.flatnonzero(cfg.region_mask.ravel() == 1)
    shapes = set()
    for c in range(2):
        for s in range(12):
            e = po.proposal(seeds[c], 40 + s, rfp, pairs, masks, centres, 128, prob["resolution"])
            assert int(p["size_idx"][c, s]) == e["size_idx"] and tuple(p["centre"][c, s].tolist()) == e["centre"]
            bh, bw = e["field"].shape
            shapes.add((bh, bw))
            f = p["fields"][c, s, : bh * bw].cpu().numpy().reshape(bh, bw)
            np.testing.assert_allclose(f, e["field"], rtol=0, atol=po.field_atol(e))
    assert max(bh * bw for bh, bw in shapes) >= 104 * 104
    eng.set_state(beds0)
    la, aa, ba = eng.run_philox(12, 40, seeds, rfp, batch=5)
    assert eng.last_run_fused() == 0
    bed_a = eng.beds.cpu().numpy().copy()
    eng.set_state(beds0)
    lr, ar = eng.run_replay(p["size_idx"].cpu().numpy(), p["centre"].cpu().numpy(), p["u"].cpu().numpy(), p["fields"])
    assert np.array_equal(aa, ar) and np.array_equal(la, lr) and np.array_equal(bed_a, eng.beds.cpu().numpy())
    eng.close()
    # 122-126-cell blocks: four coefficient planes of 64 x 80 doubles do not fit 160 KiB
    prob, cfg, pairs, masks, rfp = orc.standard_setup(128, 128, block_min=122, block_max=126)
    eng = GsmEngine(128, 128, 1)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.region_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    rfp.resolution = prob["resolution"]
    try:
        eng.set_blocks(pairs, masks)
        eng.set_centres(cfg.region_mask)
        with pytest.raises(GsmError, match="too large"):
            eng.propose_philox(2, 0, [1], rfp)
    except GsmError as e:           # or already refused by gsm_set_blocks (window larger than the LDS tile)
        assert "LDS" in str(e)
    eng.close()
