"""-m gpu: the fused chain kernel (gsm_run_philox, spectral generator) against the two-kernel pipeline and against
gsm_propose_philox + gsm_run_replay on the same Philox counters.  All three must agree bit for bit: the fused kernel
generates every proposal inside the workgroup that consumes it, with the same arithmetic."""
import numpy as np
import pytest

import mcmc_oracle as orc
from gpu_common import make_engine

pytestmark = pytest.mark.gpu


def _run_both(eng, beds0, n, step0, seeds, rfp, batch):
    res = []
    for fused in (True, False):
        eng.set_fused(fused)
        eng.set_state(beds0)
        loss, acc, blk = eng.run_philox(n, step0, seeds, rfp, batch=batch)
        assert eng.last_run_fused() == fused      # the path under test really ran
        res.append((loss, acc, blk, eng.beds.cpu().numpy().copy(), eng.resampled.cpu().numpy().copy(),
                    eng.energy.cpu().numpy().copy(), eng.loss_sum.cpu().numpy().copy()))
    return res


@pytest.mark.parametrize("model,iso,nug,block_type,in_region", [
    ("Matern", True, 0.0, "CRF_weight", True),
    ("Gaussian", False, 4.0, "RF", False),          # nugget pass, anisotropic ranges, no weight, grounded-ice mask
    ("Exponential", True, 0.0, "CRF_weight", True),
])
def test_fused_equals_two_kernel_pipeline(model, iso, nug, block_type, in_region):
    rfp = orc.RFParams(10e3, 50e3, 12e3, 40e3, 50, 150, nug, model, iso, 0.9125 if model == "Matern" else None)
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 5, block_type=block_type, update_in_region=in_region, rf_params=rfp)
    rfp.resolution = prob["resolution"]
    seeds = [3, 2 ** 41 + 5, 17, 18, 19]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(5)])
    a, b = _run_both(eng, beds0, 61, 123456789012, seeds, rfp, batch=16)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert 0.2 < a[1].mean() <= 1.0
    # resampled counts = accepted proposals summed over their (masked) windows; loss trace consistent with the carried sum
    np.testing.assert_allclose(a[0][:, -1], a[6].sum(axis=1) / (2 * cfg.sigma_mc ** 2), rtol=1e-14)
    eng.close()


def test_fused_256_headline_blocks_equal_propose_then_replay():
    """Blocks 50-80 on a 256 grid (the 7-cells-per-thread instantiation, windows clipped at the grid border included)."""
    rfp = orc.standard_rf_params()
    eng, prob, cfg, pairs, masks, _ = make_engine(256, 3)
    rfp.resolution = prob["resolution"]
    # allow centres anywhere so that windows get clipped on all four sides
    eng.set_centres(np.ones_like(cfg.region_mask))
    seeds = [31, 32, 33]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(3)])
    n = 48
    eng.set_fused(True)
    eng.set_state(beds0)
    lossA, accA, blkA = eng.run_philox(n, 0, seeds, rfp, batch=n)
    assert eng.last_run_fused()
    bedA = eng.beds.cpu().numpy().copy()
    resA = eng.resampled.cpu().numpy().copy()
    eng.set_state(beds0)
    p = eng.propose_philox(n, 0, seeds, rfp)
    lossC, accC = eng.run_replay(p["size_idx"].cpu().numpy(), p["centre"].cpu().numpy(), p["u"].cpu().numpy(), p["fields"])
    assert np.array_equal(accA, accC) and np.array_equal(lossA, lossC)
    assert np.array_equal(bedA, eng.beds.cpu().numpy())
    assert np.array_equal(resA, eng.resampled.cpu().numpy())
    # some window of the run must have been clipped, and some must have been interior
    row, col, bh, bw = (blkA[..., i] for i in range(4))
    clipped = (row - bh // 2 < 0) | (row + bh // 2 > 256) | (col - bw // 2 < 0) | (col + bw // 2 > 256)
    assert clipped.any() and (~clipped).any()
    eng.close()


def test_fused_fp32_state_equals_two_kernel_pipeline():
    rfp = orc.standard_rf_params()
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 4, state_dtype="f32")
    rfp.resolution = prob["resolution"]
    seeds = [41, 42, 43, 44]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(4)])
    a, b = _run_both(eng, beds0, 40, 7, seeds, rfp, batch=8)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    assert a[3].dtype == np.float32
    eng.close()


def test_fused_segments_reproduce_unsplit_run():
    """Counters depend on (absolute step, draw) only: 2 segments == 1 run, chain state carried in HBM between launches."""
    rfp = orc.standard_rf_params()
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 3)
    rfp.resolution = prob["resolution"]
    seeds = [51, 52, 53]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(3)])
    eng.set_state(beds0)
    loss, acc, blk = eng.run_philox(50, 1000, seeds, rfp)
    assert eng.last_run_fused()
    bed = eng.beds.cpu().numpy().copy()
    eng.set_state(beds0)
    l1, a1, b1 = eng.run_philox(23, 1000, seeds, rfp)
    l2, a2, b2 = eng.run_philox(27, 1023, seeds, rfp)
    assert np.array_equal(np.concatenate([a1, a2], axis=1), acc)
    assert np.array_equal(np.concatenate([l1, l2], axis=1), loss)
    assert np.array_equal(np.concatenate([b1, b2], axis=1), blk)
    assert np.array_equal(bed, eng.beds.cpu().numpy())
    eng.close()


def test_fused_nan_fields_and_thickness_guard_equal_two_kernel_pipeline():
    """NaN holes in the bed, a NaN velocity patch, a NaN dhdt cell (nansum semantics, MCMC.py:1041, :1328) and a surface
    lowered so far in one corner that proposals there trip the thickness guard (loss = inf, MCMC.py:1321-1329): the fused
    kernel's NaN-coded operands must give what the mask-byte kernels give (which replay tests pin to the oracle)."""
    from mcmc_gpu_amd.engine import GsmEngine
    rfp = orc.standard_rf_params()
    prob, cfg, pairs, masks, _ = orc.standard_setup(64)
    cfg.velx = cfg.velx.copy(); cfg.velx[30:33, 40:44] = np.nan
    cfg.dhdt = cfg.dhdt.copy(); cfg.dhdt[12, 12] = np.nan
    cfg.surf = cfg.surf.copy(); cfg.surf[8:30, 8:30] = prob["bed"][8:30, 8:30] + 6.0      # 6 m of ice over beds that start 0-15 m off: guard country
    eng = GsmEngine(64, 64, 6)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.region_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    eng.set_centres(cfg.region_mask)
    rfp.resolution = prob["resolution"]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(6)])
    beds0[:, 50, 20:23] = np.nan
    seeds = [71 + c for c in range(6)]
    a, b = _run_both(eng, beds0, 80, 0, seeds, rfp, batch=16)
    for x, y in zip(a, b):
        assert np.array_equal(x, y, equal_nan=True)
    loss, acc, blk = a[0], a[1], a[2]
    assert np.isfinite(loss).all()
    # The guard, deterministically: regenerate the very fields the fused kernel drew (gsm_propose_philox, same counters),
    # walk every chain on the host with the device's accept flags, and require that EVERY proposal whose candidate bed
    # leaves a cell of its window with surf - bed_next <= 0 (inside the update mask, MCMC.py:1321-1329) was rejected.
    eng.set_state(beds0)
    p = eng.propose_philox(80, 0, seeds, rfp)
    fields = p["fields"].cpu().numpy()
    w = cfg.crf_data_weight
    upd = cfg.region_mask == 1
    tripped = accepted_in_corner = 0
    for c in range(6):
        bed = beds0[c].copy()
        for s in range(80):
            row, col, bh, bw = (int(v) for v in blk[c, s])
            r0, r1 = max(0, row - bh // 2), min(64, row + bh // 2)
            c0, c1 = max(0, col - bw // 2), min(64, col + bw // 2)
            mr0, mc0 = max(bh - r1, 0), max(bw - c1, 0)
            f = fields[c, s, : bh * bw].reshape(bh, bw)[mr0:mr0 + r1 - r0, mc0:mc0 + c1 - c0]
            nxt = bed.copy()
            win = nxt[r0:r1, c0:c1]
            m = upd[r0:r1, c0:c1]
            win[m] = (win + f * w[r0:r1, c0:c1])[m]
            bad = ((cfg.surf[r0:r1, c0:c1] - win) <= 0) & m
            if bad.any():
                tripped += 1
                assert acc[c, s] == 0, f"chain {c} step {s}: thickness guard tripped but the proposal was accepted"
            if acc[c, s]:
                bed = nxt
                accepted_in_corner += int(12 <= row < 26 and 12 <= col < 26)
        assert np.array_equal(bed, a[3][c], equal_nan=True), "host walk of the accepted proposals does not reproduce the device bed"
    assert tripped >= 20, f"only {tripped} proposals tripped the guard: the test does not exercise it"
    eng.close()


def test_fused_internal_segments_are_invisible(monkeypatch):
    """gsm_run_philox cuts a long call into launches of at most 4096 steps (scratch sized by the segment);
    GSM_FUSED_SEGMENT lowers the cap: 7-step segments (a ragged last one) must reproduce the single-launch run exactly."""
    rfp = orc.standard_rf_params()
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 3)
    rfp.resolution = prob["resolution"]
    seeds = [81, 82, 83]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(3)])
    res = []
    for cap in (None, "7"):
        if cap:
            monkeypatch.setenv("GSM_FUSED_SEGMENT", cap)
        eng.set_state(beds0)
        loss, acc, blk = eng.run_philox(45, 2 ** 33 + 5, seeds, rfp)
        assert eng.last_run_fused() == 1
        res.append((loss, acc, blk, eng.beds.cpu().numpy().copy(), eng.resampled.cpu().numpy().copy(), eng.loss_sum.cpu().numpy().copy()))
    monkeypatch.delenv("GSM_FUSED_SEGMENT")
    for x, y in zip(*res):
        assert np.array_equal(x, y)
    eng.enable_timing(True)
    monkeypatch.setenv("GSM_FUSED_SEGMENT", "10")
    eng.run_philox(45, 0, seeds, rfp)
    assert eng.last_timing()["step_launches"] == 5
    eng.close()
