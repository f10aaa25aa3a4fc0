"""-m gpu: gsm_draw_pcg64 -- NumPy's PCG64 generator streams of the reference's chains advanced on the device -- against NumPy
itself on the host, in the reference's call order (MCMC.py:755, :199-207, :242, :251; :1254-1258, :1336): every draw bit for
bit, the final generator states included."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _host_draws(rf, gen_rf, gen_ch, n_steps, H, W, region_mask):
    out = dict(size_idx=[], centre=[], u=[], sc=[], re=[], im=[], ng=[])
    for _ in range(n_steps):
        i = int(gen_rf.integers(low=0, high=rf.pairs.shape[1], size=1)[0])
        bw, bh = int(rf.pairs[0, i]), int(rf.pairs[1, i])
        scale = gen_rf.uniform(rf.scale_min, rf.scale_max) / 3.0
        nug = gen_rf.uniform(0.0, rf.nugget_max)
        if not rf.isotropic:
            rx = gen_rf.uniform(rf.range_min_x, rf.range_max_x); ry = gen_rf.uniform(rf.range_min_y, rf.range_max_y)
        else:
            rx = ry = gen_rf.uniform(rf.range_min_x, rf.range_max_x)
        re = gen_rf.normal(size=(bh, bw)); im = gen_rf.normal(size=(bh, bw))
        ng = gen_rf.normal(0, np.sqrt(nug), size=(bh, bw))
        while True:
            ix = int(gen_ch.integers(low=0, high=H, size=1)[0]); iy = int(gen_ch.integers(low=0, high=W, size=1)[0])
            if region_mask is None or region_mask[ix, iy] == 1:
                break
        out["size_idx"].append(i); out["centre"].append((ix, iy)); out["u"].append(gen_ch.random())
        out["sc"].append((scale, nug, rx, ry)); out["re"].append(re); out["im"].append(im); out["ng"].append(ng)
    return out


@pytest.mark.parametrize("H,blocks,isotropic,nugget_max,in_region", [(64, (8, 16), True, 0.0, True), (96, (20, 40), False, 4.0, True),
                                                                      (256, (50, 80), True, 0.0, False)])
def test_device_draws_equal_numpy(H, blocks, isotropic, nugget_max, in_region):
    from mcmc_gpu_amd import MCMC_gpu, synthetic
    from mcmc_gpu_amd.engine import GsmEngine
    prob, ch, _ = synthetic.template(H)
    rf = MCMC_gpu.RandField(10e3, 50e3, 12e3, 40e3, 50, 150, nugget_max, "Matern", isotropic, smoothness=0.9125)
    rf.set_block_sizes(blocks[0], blocks[1], blocks[0], blocks[1])
    rf.set_weight_param(2, 0, 6, 1, 49900.0, prob["resolution"])
    rf.set_generation_method(True)
    n_chains, n_steps = 5, 7 if H < 256 else 4
    eng = ch._make_engine(rf, n_chains, 0)
    seeds = [3, 7, 2 ** 40 + 11, 123456789, 987654321987654321]
    gens_rf = [np.random.default_rng(seed=s) for s in seeds]
    gens_ch = [np.random.default_rng(seed=s) for s in seeds]
    for g in gens_ch[:2]:
        g.integers(low=0, high=9, size=1)                      # start with a cached 32-bit half in some generators
    region = prob["region_mask"] if in_region else None
    d_rf = torch.as_tensor(GsmEngine.pack_pcg64_states(gens_rf).view(np.int64)).to(eng.dev)
    d_ch = torch.as_tensor(GsmEngine.pack_pcg64_states(gens_ch).view(np.int64)).to(eng.dev)
    d_reg = torch.as_tensor(np.ascontiguousarray(region == 1, dtype=np.uint8)).to(eng.dev) if in_region else None
    d = eng.draw_pcg64(n_steps, rf, d_rf, d_ch, d_reg)
    torch.cuda.synchronize()
    st_rf = GsmEngine.unpack_pcg64_states(d_rf.cpu().numpy().view(np.uint64))
    st_ch = GsmEngine.unpack_pcg64_states(d_ch.cpu().numpy().view(np.uint64))
    re, im = d["noise_re"].cpu().numpy(), d["noise_im"].cpu().numpy()
    ng = d["nugget"].cpu().numpy() if d["nugget"] is not None else None
    assert (ng is None) == (nugget_max == 0.0)
    for c in range(n_chains):
        h = _host_draws(rf, gens_rf[c], gens_ch[c], n_steps, H, H, region)
        assert np.array_equal(d["size_idx"][c].cpu().numpy(), h["size_idx"])
        assert np.array_equal(d["centre"][c].cpu().numpy(), np.array(h["centre"]))
        assert np.array_equal(d["u"][c].cpu().numpy(), np.array(h["u"]))
        assert np.array_equal(d["rf_scalars"][c].cpu().numpy(), np.array(h["sc"]))
        for s in range(n_steps):
            B = h["re"][s].size
            assert np.array_equal(re[c, s, :B], h["re"][s].ravel()), (c, s, "real plane")
            assert np.array_equal(im[c, s, :B], h["im"][s].ravel()), (c, s, "imaginary plane")
            if ng is not None:
                assert np.array_equal(ng[c, s, :B], h["ng"][s].ravel()), (c, s, "nugget plane")
        assert st_rf[c] == gens_rf[c].bit_generator.state, "RandField generator state"
        assert st_ch[c] == gens_ch[c].bit_generator.state, "chain generator state"
    eng.close()


def _rehydrate(ch, rf, seed, bed):
    from copy import deepcopy
    from mcmc_gpu_amd import MCMC_gpu
    cp = deepcopy(ch.__dict__); cp["rng_seed"] = seed; cp["initial_bed"] = bed
    rp = deepcopy(rf.__dict__); rp["rng_seed"] = seed
    return MCMC_gpu.init_lsc_chain_by_instance(cp), MCMC_gpu.initiate_RF_by_instance(rp)


def test_pcg64_mode_follows_the_reference_chain(golden_dir):
    """chain_crf_gpu.run in 'pcg64' mode on the seeds of golden F1 (the imported reference's chain_crf.run, 300 iterations): the
    draws are NumPy's, made on the device, so block records, accept mask, resampled counts and the final generator states are
    the reference's; the bed and the loss follow to the accuracy of the device's inverse DFT against pocketfft (1e-12 x the
    field scale per proposal, summed over the accepted steps)."""
    from mcmc_gpu_amd import synthetic
    g = np.load(golden_dir / "f1_chain64_standard.npz")
    prob, ch, rf = synthetic.template(64)
    c, r = _rehydrate(ch, rf, 7, prob["bed"].copy())
    c.set_rng_mode('pcg64')
    c.replay_chunk = 64
    out = c.run(300, r, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    bed, loss_mc, loss_data, loss, steps, resampled, blocks = out
    assert np.array_equal(steps, g["steps"]), "accept mask differs from the reference"
    assert np.array_equal(blocks, g["blocks"], equal_nan=True)
    assert np.array_equal(resampled, g["resampled"])
    np.testing.assert_allclose(loss, g["loss"], rtol=1e-10)
    np.testing.assert_allclose(bed, g["bed"], rtol=0, atol=1e-9)
    # the generators end where the host-drawn replay chain leaves them
    c2, r2 = _rehydrate(ch, rf, 7, prob["bed"].copy())
    c2.replay_chunk = 64
    c2.run(300, r2, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
    assert c.rng.bit_generator.state == c2.rng.bit_generator.state and r.rng.bit_generator.state == r2.rng.bit_generator.state


def test_pcg64_batches_equal_host_replay_at_headline_geometry():
    """run_many_pcg64 against run_many_replay (host NumPy draws + NumPy FFT, the bit-exact parity mode): 24 chains x 120 steps at
    256 x 256 with blocks 50-80 -- identical accept masks, blocks and generator states, beds within 1e-9 m."""
    from mcmc_gpu_amd import MCMC_gpu, synthetic
    prob, ch, rf = synthetic.template(256)
    n = 24
    beds = np.stack(list(synthetic.initial_beds(prob, n)))
    st = [np.random.default_rng(seed=900 + i).bit_generator.state for i in range(n)]
    a, rf_a, ch_a = MCMC_gpu.run_many_pcg64(ch, rf, beds, st, st, 121, batch=32)
    b, rf_b, ch_b = MCMC_gpu.run_many_replay(ch, rf, beds, st, st, 121, n_workers=8)
    assert rf_a == rf_b and ch_a == ch_b
    for x, y in zip(a, b):
        assert np.array_equal(x[4], y[4]) and np.array_equal(x[6], y[6], equal_nan=True) and np.array_equal(x[5], y[5])
        np.testing.assert_allclose(x[3], y[3], rtol=1e-10)
        np.testing.assert_allclose(x[0], y[0], rtol=0, atol=1e-9)
    assert 0.3 < np.mean([x[4][1:].mean() for x in a]) < 0.8


@pytest.mark.parametrize("H,blocks,isotropic,nugget_max", [(256, (50, 80), True, 0.0), (96, (20, 40), False, 4.0)])
def test_fused_synthesis_and_step_equal_the_two_calls(H, blocks, isotropic, nugget_max):
    """gsm_run_noise (synthesis + step in one kernel, the field never leaving the CU) against gsm_spectral_from_noise +
    gsm_run_replay on the same device draws: every output bit for bit (beds, losses, accept masks, resampled counts, generator
    states) -- the same stage functions form the field, the same strip functions take the step."""
    from mcmc_gpu_amd import MCMC_gpu, synthetic
    prob, ch, _ = synthetic.template(H)
    rf = MCMC_gpu.RandField(10e3, 50e3, 12e3, 40e3, 50, 150, nugget_max, "Matern", isotropic, smoothness=0.9125)
    rf.set_block_sizes(blocks[0], blocks[1], blocks[0], blocks[1])
    rf.set_weight_param(2, 0, 6, 1, 49900.0, prob["resolution"])
    rf.set_generation_method(True)
    ch.set_crf_data_weight(rf)
    n = 6
    beds = np.stack(list(synthetic.initial_beds(prob, n)))
    st = [np.random.default_rng(seed=500 + i).bit_generator.state for i in range(n)]
    a, rf_a, ch_a = MCMC_gpu.run_many_pcg64(ch, rf, beds, st, st, 81, batch=32, fused=True)
    b, rf_b, ch_b = MCMC_gpu.run_many_pcg64(ch, rf, beds, st, st, 81, batch=32, fused=False)
    assert rf_a == rf_b and ch_a == ch_b
    for x, y in zip(a, b):
        for k in (0, 3, 4, 5):
            assert np.array_equal(x[k], y[k]), k
        assert np.array_equal(x[6], y[6], equal_nan=True)
    assert 0.2 < np.mean([x[4][1:].mean() for x in a]) < 0.95


def test_small_scale_device_draws_equal_numpy():
    """gsm_sgs_draw_pcg64 against the host mirror's NumPy calls (chain_sgs_gpu._draw_iteration: centre, block sizes, rng.shuffle,
    one normal per cell without data, rng.random -- MCMC.py:1750-1757, :128, :165, :1797): windows, visiting orders, normals,
    uniforms and the final generator states, bit for bit, blocks up to 19 x 19."""
    import ctypes as C
    from mcmc_gpu_amd import synthetic
    from mcmc_gpu_amd.engine import GsmEngine, _ptr
    H, n, kb = 64, 6, 9
    prob, ch = synthetic.sgs_template(H, transform=False)
    is_data = ~np.isnan(prob["cond_bed"])
    gens = [np.random.default_rng(seed=77 + c) for c in range(n)]
    gens[1].integers(low=0, high=5, size=1)                   # a cached 32-bit half at the start
    eng = GsmEngine(H, H, n)
    dev = eng.dev
    d_gen = torch.as_tensor(GsmEngine.pack_pcg64_states(gens).view(np.int64)).to(dev)
    max_cells = (ch.block_max_x - 1) * (ch.block_max_y - 1)
    i32 = lambda *shape: torch.empty(shape, dtype=torch.int32, device=dev)
    d_win, d_blk, d_off, d_cnt = i32(kb, n, 4), i32(kb, n, 4), i32(kb, n), i32(kb, n)
    d_cells = i32(kb * n * max_cells, 2); d_z = torch.zeros(kb * n * max_cells, dtype=torch.float64, device=dev)
    d_u = torch.empty((kb, n), dtype=torch.float64, device=dev)
    d_reg = torch.as_tensor(np.ascontiguousarray(prob["region_mask"] == 1, dtype=np.uint8)).to(dev)
    d_isd = torch.as_tensor(np.ascontiguousarray(is_data, dtype=np.uint8)).to(dev)
    eng._check(eng.lib.gsm_sgs_draw_pcg64(eng.h, _ptr(d_gen), kb, _ptr(d_reg), _ptr(d_isd), ch.block_min_x, ch.block_max_x, ch.block_min_y,
                                          ch.block_max_y, max_cells, _ptr(d_win), _ptr(d_blk), _ptr(d_off), _ptr(d_cnt), _ptr(d_cells), _ptr(d_z),
                                          _ptr(d_u), eng._stream()))
    eng._check(eng.lib.gsm_sgs_check(eng.h, eng._stream()))
    win, blk, cnt = d_win.cpu().numpy(), d_blk.cpu().numpy(), d_cnt.cpu().numpy()
    cells, z, u = d_cells.cpu().numpy().reshape(kb, n, max_cells, 2), d_z.cpu().numpy().reshape(kb, n, max_cells), d_u.cpu().numpy()
    states = GsmEngine.unpack_pcg64_states(d_gen.cpu().numpy().view(np.uint64))
    for c in range(n):
        for j in range(kb):
            b, w, inds, zz, uu = ch._draw_iteration(gens[c], is_data)
            assert tuple(blk[j, c]) == tuple(int(v) for v in b) and tuple(win[j, c]) == tuple(w)
            assert cnt[j, c] == inds.shape[0]
            assert np.array_equal(cells[j, c, :cnt[j, c]], inds)
            assert np.array_equal(z[j, c, :cnt[j, c]], zz)
            assert u[j, c] == uu
        assert states[c] == gens[c].bit_generator.state
    eng.close()


def test_pcg64_mode_refuses_a_generator_shared_by_randfield_and_chain():
    """The reference lets one Generator serve both the RandField and the chain (draws interleave); the device advances two
    independent streams, so 'pcg64' mode must refuse that set-up instead of silently diverging (replay mode handles it)."""
    from mcmc_gpu_amd import synthetic
    prob, ch, rf = synthetic.template(64)
    c, r = _rehydrate(ch, rf, 7, prob["bed"].copy())
    r.rng = c.rng
    c.set_rng_mode('pcg64')
    with pytest.raises(ValueError, match="separate generators"):
        c.run(5, r, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False, progress_bar=None)
