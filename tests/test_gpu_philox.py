"""-m gpu: the Philox proposal generator against its CPU restatement (oracle/philox_oracle.py)."""
import numpy as np
import pytest

import mcmc_oracle as orc
import philox_oracle as po
from gpu_common import make_engine

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("model,iso,nug", [("Matern", True, 0.0), ("Gaussian", False, 4.0), ("Exponential", True, 0.0)])
def test_proposals_match_oracle(model, iso, nug):
    rfp = orc.RFParams(10e3, 50e3, 12e3, 40e3, 50, 150, nug, model, iso, 0.9125 if model == "Matern" else None)
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 3, rf_params=rfp)
    seeds = [7, 2 ** 40 + 12345, 99]
    rfp.resolution = prob["resolution"]
    out = eng.propose_philox(5, 1000, seeds, rfp)
    centres = np.flatnonzero(cfg.region_mask.ravel() == 1)
    for c in range(3):
        for s in range(5):
            e = po.proposal(seeds[c], 1000 + s, rfp, pairs, masks, centres, 64, prob["resolution"])
            assert int(out["size_idx"][c, s]) == e["size_idx"]
            assert tuple(out["centre"][c, s].tolist()) == e["centre"]
            assert float(out["u"][c, s]) == e["u"]
            sc = out["rf_scalars"][c, s].cpu().numpy()
            np.testing.assert_allclose(sc, [e["scale"], e["nug"], e["range_x"], e["range_y"]], rtol=1e-15)
            bh, bw = e["field"].shape
            f = out["fields"][c, s, : bh * bw].cpu().numpy().reshape(bh, bw)
            np.testing.assert_allclose(f, e["field"], rtol=0, atol=po.field_atol(e))
    eng.close()


def test_proposals_256_blocks_match_oracle():
    rfp = orc.standard_rf_params()
    eng, prob, cfg, pairs, masks, _ = make_engine(256, 2)
    rfp.resolution = prob["resolution"]
    seeds = [11, 12]
    out = eng.propose_philox(6, 0, seeds, rfp)
    centres = np.flatnonzero(cfg.region_mask.ravel() == 1)
    for c in range(2):
        for s in range(6):
            e = po.proposal(seeds[c], s, rfp, pairs, masks, centres, 256, prob["resolution"])
            assert int(out["size_idx"][c, s]) == e["size_idx"]
            bh, bw = e["field"].shape
            f = out["fields"][c, s, : bh * bw].cpu().numpy().reshape(bh, bw)
            np.testing.assert_allclose(f, e["field"], rtol=0, atol=po.field_atol(e))
    eng.close()


def test_run_philox_equals_propose_then_replay():
    """gsm_run_philox (batched, two streams) == gsm_propose_philox + gsm_run_replay on the same counters,
    and is independent of the batch size."""
    rfp = orc.standard_rf_params()
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 4)
    rfp.resolution = prob["resolution"]
    seeds = [21, 22, 23, 24]
    beds0 = np.stack([orc.chain_initial_bed(prob, c) for c in range(4)])
    n = 37
    eng.set_state(beds0)
    lossA, accA, blkA = eng.run_philox(n, 5, seeds, rfp, batch=8)
    bedA = eng.beds.cpu().numpy().copy()
    resA = eng.resampled.cpu().numpy().copy()
    eng.set_state(beds0)
    lossB, accB, blkB = eng.run_philox(n, 5, seeds, rfp, batch=37)
    assert np.array_equal(accA, accB) and np.array_equal(lossA, lossB) and np.array_equal(blkA, blkB)
    assert np.array_equal(bedA, eng.beds.cpu().numpy())
    eng.set_state(beds0)
    p = eng.propose_philox(n, 5, seeds, rfp)
    lossC, accC = eng.run_replay(p["size_idx"].cpu().numpy(), p["centre"].cpu().numpy(), p["u"].cpu().numpy(), p["fields"])
    assert np.array_equal(accA, accC) and np.array_equal(lossA, lossC)
    assert np.array_equal(bedA, eng.beds.cpu().numpy())
    assert np.array_equal(resA, eng.resampled.cpu().numpy())
    assert np.array_equal(blkA[..., 0], p["centre"].cpu().numpy()[..., 0])
    assert 0.3 < accA.mean() <= 1.0
    eng.close()


def test_tiny_and_odd_shaped_blocks_match_oracle():
    """Block table 2..12 cells (incl. 2x2, 2x12, 12x2): the smallest DFT sizes and the most padded MFMA tiles."""
    rfp = orc.standard_rf_params()
    prob, cfg, _, _, _ = orc.standard_setup(64)
    pairs = orc.block_pairs(2, 12, 2, 12, steps=3)
    masks = [np.full((int(pairs[1, i]), int(pairs[0, i])), 0.5) for i in range(pairs.shape[1])]
    from mcmc_gpu_amd.engine import GsmEngine
    eng = GsmEngine(64, 64, 2)
    eng.set_static(cfg.surf, cfg.velx, cfg.vely, cfg.dhdt, cfg.smb, cfg.crf_data_weight, cfg.region_mask,
                   cfg.mc_region_mask, cfg.resolution, cfg.sigma_mc)
    eng.set_blocks(pairs, masks)
    eng.set_centres(cfg.region_mask)
    rfp.resolution = 500.0
    seeds = [3, 4]
    out = eng.propose_philox(24, 0, seeds, rfp)
    centres = np.flatnonzero(cfg.region_mask.ravel() == 1)
    shapes = set()
    for c in range(2):
        for s in range(24):
            e = po.proposal(seeds[c], s, rfp, pairs, masks, centres, 64, 500.0)
            assert int(out["size_idx"][c, s]) == e["size_idx"]
            bh, bw = e["field"].shape
            shapes.add((bh, bw))
            f = out["fields"][c, s, : bh * bw].cpu().numpy().reshape(bh, bw)
            np.testing.assert_allclose(f, e["field"], rtol=0, atol=po.field_atol(e))
    assert len(shapes) >= 6 and (2, 2) in shapes or len(shapes) >= 6
    eng.close()


def test_device_normals_match_oracle():
    """gsm_debug_normals: the device's Box-Muller pairs (table-driven log / sincos, lean sqrt) against numpy's on the same
    Philox counters: absolute error < 5e-15 over 2e5 pairs (|normal| <= 8.6), including both streams."""
    import ctypes as C
    import torch
    from mcmc_gpu_amd import _lib
    lib = _lib.load()
    n = 200000
    out = torch.empty(2 * n, dtype=torch.float64, device="cuda:0")
    for seed, step, stream, idx0 in ((7, 0, po.STREAM_SPECTRUM, 0), (2 ** 40 + 12345, 10 ** 9 + 7, po.STREAM_NUGGET, 4000000000)):
        rc = lib.gsm_debug_normals(C.c_uint64(seed), step, stream, idx0, n, C.c_void_p(out.data_ptr()), None)
        assert rc == 0
        got = out.cpu().numpy().reshape(n, 2)
        idx = (np.arange(n, dtype=np.uint64) + np.uint64(idx0)).astype(np.uint32)
        g1, g2 = po.normals2(seed, step, stream, idx)
        err = max(np.abs(got[:, 0] - g1).max(), np.abs(got[:, 1] - g2).max())
        assert err < 5e-15, err
        assert abs(got.mean()) < 0.01 and abs(got.std() - 1.0) < 0.01
