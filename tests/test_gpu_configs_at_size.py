"""-m gpu: the BASELINE configs that round 1 only exercised at reduced size, at their size (one GPU's share):

  configs[3]  512x512 grid, blocks 50-80, precomputed-Cholesky generator, 2 range classes (N = bh*bw up to 6400: the
              many-tile triangular regime of cz_gemm_dma_kernel), 256 chains x 32 steps with value checks and
              1024 chains x 64 steps (the bench's launch) with the invariants
  configs[4]  1024x1024 grid, fp32 state / fp64 arithmetic, 512 chains (one GPU's shard of the 4096), 32 steps

The oracle cannot run these sizes in seconds; checked instead: (a) proposal fields of a sample of (size, class) buckets
against U^T z * scale * mask formed with torch.matmul from the SAME device factor and the Philox restatement's z
(1e-12 x scale), (b) run == propose + replay bit for bit, (c) the size-independent invariants P1-P4 of
test_gpu_fullsize.py."""
import numpy as np
import pytest
import torch

import philox_oracle as po
from gpu_common import check_chain_invariants
from mcmc_gpu_amd import cholesky as chol, synthetic

pytestmark = pytest.mark.gpu


def _device_initial_beds(prob, n_chains, dev, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    bed = torch.as_tensor(prob["bed"], device=dev)
    return bed[None] + 5.0 * torch.randn((n_chains,) + tuple(bed.shape), dtype=torch.float64, device=dev, generator=g)


def test_config3_cholesky_512_grid_at_size():
    H, n_chains, n_steps, ncls = 512, 256, 32, 2
    prob, ch, rf = synthetic.template(H)
    rf.generator = "cholesky"
    eng = ch._make_engine(rf, n_chains, 0)
    factors = chol.build_factors(eng, rf, n_classes=ncls)
    assert max(f.shape[0] for f in factors) == 6400 and len(factors) == 25 * ncls
    seeds = [9000 + 7 * c for c in range(n_chains)]
    step0 = 12345
    beds0 = _device_initial_beds(prob, n_chains, eng.dev, 3)
    # (a) the proposals themselves
    p = eng.propose_philox(n_steps, step0, seeds, rf)
    si_h, rc_h = p["size_idx"].cpu().numpy(), p["rf_scalars"][..., 2].cpu().numpy().astype(int)
    seen, checked = set(), 0
    for c in range(n_chains):
        for s in range(n_steps):
            key = (int(si_h[c, s]), int(rc_h[c, s]))
            if key in seen or checked >= 14:
                continue
            seen.add(key); checked += 1
            d = po.draw(seeds[c], step0 + s, po.STREAM_SCALARS, np.arange(4))
            si = int((int(d[3, 0]) * eng.n_sizes) >> 32); rc = int((int(d[1, 0]) * ncls) >> 32)
            assert (si, rc) == key
            scale = (rf.scale_min + (rf.scale_max - rf.scale_min) * float(po.u01(d[0, 0], d[0, 1]))) / 3.0
            bh, bw = int(eng.bh[si]), int(eng.bw[si]); N = bh * bw
            g1, g2 = po.normals2(seeds[c], step0 + s, 3, np.arange((N + 1) // 2))
            z = np.empty(2 * len(g1)); z[0::2], z[1::2] = g1, g2
            U = factors[si * ncls + rc][:N, :N]
            exact = (torch.matmul(U.T, torch.as_tensor(z[:N], device=eng.dev)).reshape(bh, bw) * scale
                     * torch.as_tensor(rf.edge_masks[si], device=eng.dev))
            got = p["fields"][c, s, :N].reshape(bh, bw)
            err = float((got - exact).abs().max())
            assert err <= 1e-12 * scale, f"bucket {key} N={N}: max error {err:.3e} (scale {scale:.2f})"
            assert float(exact.abs().max()) > 0.2 * scale
    assert checked >= 10 and {k[1] for k in seen} == {0, 1} and max(int(eng.bh[k[0]]) * int(eng.bw[k[0]]) for k in seen) >= 5600
    # (b) run_philox (pipelined two-stream batches) == propose + replay
    loss0 = eng.set_state(beds0)
    loss, acc, blk = eng.run_philox(n_steps, step0, seeds, rf, batch=16)
    assert eng.last_run_fused() == 0
    beds_a = eng.beds.clone()
    eng.set_state(beds0)
    loss_r, acc_r = eng.run_replay(si_h, p["centre"].cpu().numpy(), p["u"].cpu().numpy(), p["fields"])
    assert np.array_equal(acc, acc_r) and np.array_equal(loss, loss_r) and torch.equal(beds_a, eng.beds)
    assert 0.4 < acc.mean() < 0.75
    del p
    # (c) invariants on the run_philox result
    eng.set_state(beds0)
    loss, acc, blk = eng.run_philox(n_steps, step0, seeds, rf, batch=16)
    check_chain_invariants(eng, prob["region_mask"], beds0, loss0, loss, acc, blk, (0, 1, 128, 255))
    eng.close()


def test_config3_cholesky_512_grid_1024_chains():
    """BASELINE configs[3] at its full chain count (512 x 512 grid x 1024 chains, the bench's batch of 64 steps per launch:
    65536 proposals bucketed into 50 (size, class) groups of ~1300): the size-independent invariants P1-P4 on the result."""
    H, n_chains, n_steps, ncls = 512, 1024, 64, 2
    prob, ch, rf = synthetic.template(H)
    rf.generator = "cholesky"
    eng = ch._make_engine(rf, n_chains, 0)
    chol.build_factors(eng, rf, n_classes=ncls)
    seeds = [31000 + 3 * c for c in range(n_chains)]
    beds0 = _device_initial_beds(prob, n_chains, eng.dev, 5)
    loss0 = eng.set_state(beds0)
    loss, acc, blk = eng.run_philox(n_steps, 777, seeds, rf, batch=64)
    assert eng.last_run_fused() == 0
    assert 0.4 < acc.mean() < 0.75
    check_chain_invariants(eng, prob["region_mask"], beds0, loss0, loss, acc, blk, (0, 1, 511, 1023))
    eng.close()


def test_config4_f32_state_1024_grid_at_size():
    H, n_chains, n_steps = 1024, 512, 32
    prob, ch, rf = synthetic.template(H)
    ch.state_dtype = "f32"
    eng = ch._make_engine(rf, n_chains, 0)
    beds0 = _device_initial_beds(prob, n_chains, eng.dev, 4).float()
    loss0 = eng.set_state(beds0)
    assert eng.beds.dtype == torch.float32 and eng.energy.dtype == torch.float32
    d_beds0 = eng.beds.clone()
    seeds = list(range(70000, 70000 + n_chains))
    loss, acc, blk = eng.run_philox(n_steps, 0, seeds, rf)
    assert eng.last_run_fused() == 1
    assert 0.4 < acc.mean() < 0.75
    check_chain_invariants(eng, prob["region_mask"], d_beds0, loss0, loss, acc, blk, (0, 255, 511))
    # the two-kernel pipeline on a slice gives the same chain (same arithmetic, same rounding to float)
    eng.close()
    eng = ch._make_engine(rf, 16, 0)
    res = []
    for fused in (1, 0):
        eng.set_fused(fused)
        eng.set_state(beds0[:16])
        out = eng.run_philox(n_steps, 0, seeds[:16], rf, batch=8)
        assert eng.last_run_fused() == fused
        res.append(out + (eng.beds.cpu().numpy().copy(),))
    for x, y in zip(*res):
        assert np.array_equal(x, y)
    assert np.array_equal(res[0][1], acc[:16]) and np.array_equal(res[0][0], loss[:16])
    eng.close()
