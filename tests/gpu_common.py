"""Helpers shared by the -m gpu tests: build an engine for the oracle's standard setups."""
import numpy as np

import mcmc_oracle as orc


def make_engine(H, n_chains, block_type="CRF_weight", update_in_region=True, rf_params=None, W=None, state_dtype="f64"):
    from mcmc_gpu_amd.engine import GsmEngine
    prob, cfg, pairs, masks, rfp = orc.standard_setup(H, W, block_type=block_type,
                                                      update_in_region=update_in_region, rf_params=rf_params)
    Hh, Ww = prob["bed"].shape
    eng = GsmEngine(Hh, Ww, n_chains, state_dtype=state_dtype)
    upd = cfg.region_mask if update_in_region else cfg.grounded_ice_mask
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb,
                   cfg.crf_data_weight if block_type == "CRF_weight" else None,
                   upd, cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    eng.set_centres(cfg.region_mask)
    return eng, prob, cfg, pairs, masks, rfp


def oracle_chains(prob, cfg, pairs, masks, rfp, n_chains, n_iter, seed0=7, state_f32=False):
    cfg.state_f32 = state_f32
    outs = []
    for c in range(n_chains):
        rf = orc.OracleRandField(rfp, seed0 + c, pairs, masks, prob["resolution"])
        rng = np.random.default_rng(seed=seed0 + c)
        outs.append(orc.run_chain(cfg, orc.chain_initial_bed(prob, c), n_iter, rf, rng, record=True))
    return outs


def replay_inputs(eng, outs):
    n_chains, n_steps = len(outs), len(outs[0][7].u)
    size_idx = np.array([o[7].size_idx for o in outs])
    centre = np.array([o[7].centre for o in outs])
    u = np.array([o[7].u for o in outs])
    fields = eng.pack_fields([o[7].fields for o in outs])
    return size_idx, centre, u, fields


def windows_count(blocks, acc, H, W):
    cnt = np.zeros((H, W), dtype=np.int64)
    for (row, col, bh, bw), a in zip(blocks, acc):
        if a:
            r0, r1 = max(0, row - bh // 2), min(H, row + bh // 2)
            c0, c1 = max(0, col - bw // 2), min(W, col + bw // 2)
            cnt[r0:r1, c0:c1] += 1
    return cnt


def check_chain_invariants(eng, region_mask, beds0, loss0, loss, acc, blk, sample_chains):
    """Size-independent properties of a finished Philox-mode run (no oracle needed), see test_gpu_fullsize.py:
    P1 carried energy == from-scratch recompute of the final beds, bit for bit, every chain;
    P2 compensated carried sum == sum(energy) (1e-12), last loss == recomputed loss (1e-10);
    P3 resampled == accepted windows covering each cell inside the update mask (sample of chains);
    P4 rejected steps repeat the previous loss exactly, cells outside the update mask never change.
    beds0: device tensor of the initial beds (state dtype).  Leaves the engine state recomputed from its final beds."""
    import torch
    H, W = eng.H, eng.W
    beds = eng.beds.clone(); energy = eng.energy.clone(); res = eng.resampled.clone(); lsum = eng.loss_sum.clone()
    loss_re = eng.set_state(beds, resampled=res)
    assert torch.equal(eng.energy, energy), "carried energy differs from a full recompute"
    s_carried = (lsum[:, 0] + lsum[:, 1]).cpu().numpy()
    s_energy = energy.sum(dim=(1, 2), dtype=torch.float64).cpu().numpy()
    np.testing.assert_allclose(s_carried, s_energy, rtol=1e-12)
    np.testing.assert_allclose(loss[:, -1], loss_re, rtol=1e-10)
    prev = np.concatenate([loss0[:, None], loss[:, :-1]], axis=1)
    assert np.array_equal(loss[acc == 0], prev[acc == 0])
    assert (loss[acc == 1] != prev[acc == 1]).mean() > 0.99
    outside = torch.as_tensor(np.asarray(region_mask) == 0, device=beds.device)
    assert torch.equal(beds[:, outside], beds0[:, outside])
    assert not torch.equal(beds, beds0)
    resh = res[list(sample_chains)].cpu().numpy()
    for k, c in enumerate(sample_chains):
        exp = windows_count(blk[c], acc[c], H, W) * (np.asarray(region_mask) == 1)
        assert np.array_equal(resh[k], exp)
    return beds, res
