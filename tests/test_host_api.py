"""CPU suite: host logic of the product package, the C-ABI library's exports, Philox and exact-division checks.
No compute call is made on a GPU here."""
import ctypes
import json
import subprocess
import sys
from fractions import Fraction

import numpy as np
import pytest

import mcmc_oracle as orc
import philox_oracle as po
from mcmc_gpu_amd import MCMC_gpu, _lib, parallel, synthetic


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    names = _lib.declared_symbols()
    assert {"gsm_create", "gsm_destroy", "gsm_set_static", "gsm_set_blocks", "gsm_set_centres", "gsm_init_loss",
            "gsm_residual", "gsm_run_replay", "gsm_propose_philox", "gsm_run_philox", "gsm_last_error",
            "gsm_version", "gsm_enable_timing", "gsm_last_timing", "gsm_philox_selftest"} <= set(names)
    for n in names:
        assert hasattr(lib, n), n
    v = lib.gsm_version().decode()
    assert " gfx950 src:" in v and v.endswith(_lib.source_hash())      # the binary is the one built from these sources
    assert "gsm_spectral_from_noise" in names
    # the shared object carries a gfx950 code object
    out = subprocess.run(["strings", "-n", "6", str(_lib.LIB_PATH)], capture_output=True, text=True).stdout
    assert "gfx950" in out


def test_struct_restatements_have_the_librarys_layout():
    """The ctypes mirrors of include/gsm.h's structs against sizeof in the compiled library (gsm_struct_size): a field added on one side
    only would shift every later field silently.  (load() makes the same check and refuses a mismatching library.)"""
    import ctypes as C
    from mcmc_gpu_amd import _lib
    lib = _lib.load()
    assert lib.gsm_struct_size(0) == C.sizeof(_lib.RfParams)
    assert lib.gsm_struct_size(1) == C.sizeof(_lib.SgsBatch)
    assert lib.gsm_struct_size(2) == C.sizeof(_lib.Vario)
    assert lib.gsm_struct_size(99) == -1


def test_create_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    lib = _lib.load()
    h = ctypes.c_void_p()
    rc = lib.gsm_create(ctypes.byref(h), 64, 64, 1, 0, 0)
    assert rc == -3 and not h.value
    assert b"HIP" in lib.gsm_last_error(None) or b"device" in lib.gsm_last_error(None)
    from mcmc_gpu_amd.engine import GsmEngine
    with pytest.raises(RuntimeError):
        GsmEngine(64, 64, 1)
    prob, ch, rf = synthetic.template(64)
    ch.set_random_generator(7)
    with pytest.raises(RuntimeError):
        ch.run(5, rf, only_save_last_bed=True, plot=False, progress_bar=False)


def test_philox_known_answer_vectors():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        assert tuple(_lib.philox4x32_10(ctr, key)) == exp                       # csrc/philox.h (host build)
        assert tuple(int(v) for v in po.philox4x32_10(np.array(ctr), key)) == exp  # oracle
    # library and oracle agree on random counters
    g = np.random.default_rng(3)
    for _ in range(50):
        ctr = g.integers(0, 2 ** 32, 4); key = g.integers(0, 2 ** 32, 2)
        assert _lib.philox4x32_10(ctr, key) == [int(v) for v in po.philox4x32_10(ctr, key)]


def test_exact_div_is_correctly_rounded():
    """step_kernel.hip's exact_div(x, d, RN(1/d)) == RN(x/d): checked with exact rationals (fma emulated exactly)."""
    g = np.random.default_rng(0)
    dens = [500.0, 1000.0, 437.5, 875.0, 250.0, 100.0, 3.0, 0.1] + list(g.uniform(1, 5000, 12))
    n = bad = 0
    for d in dens:
        y = float(Fraction(1) / Fraction(d))
        xs = np.concatenate([g.normal(0, 1e5, 300), g.normal(0, 1, 150), 10.0 ** g.uniform(-20, 20, 100)])
        for x in map(float, xs):
            q0 = float(Fraction(x) * Fraction(y))
            r = float(Fraction(x) - Fraction(q0) * Fraction(d))
            q1 = float(Fraction(q0) + Fraction(r) * Fraction(y))
            n += 1
            bad += q1 != float(Fraction(x) / Fraction(d))
    assert n > 10000 and bad == 0


def test_randfield_setup_and_draws_equal_the_oracle():
    prob, ch, rf = synthetic.template(64)
    p2, cfg, pairs, masks, rfp = orc.standard_setup(64)
    assert np.array_equal(rf.pairs, pairs)
    assert all(np.array_equal(a, b) for a, b in zip(rf.edge_masks, masks))
    assert np.array_equal(ch.crf_data_weight, cfg.crf_data_weight)
    for k in ("surf", "velx", "vely", "dhdt", "smb", "bed", "region_mask"):
        assert np.array_equal(prob[k], p2[k])
    rf.rng = np.random.default_rng(11)
    o = orc.OracleRandField(rfp, 11, pairs, masks, 500.0)
    for _ in range(25):
        assert np.array_equal(rf.get_rfblock(), o.get_rfblock())
    assert rf.rng.bit_generator.state == o.rng.bit_generator.state
    # anisotropic Gaussian with nugget, Exponential
    for model, iso, nug in (("Gaussian", False, 4.0), ("Exponential", True, 0.0)):
        r2 = MCMC_gpu.RandField(8e3, 30e3, 12e3, 40e3, 30, 90, nug, model, iso, rng_seed=5)
        r2.set_block_sizes(8, 16, 8, 16); r2.set_weight_param(2, 0, 6, 1, 49900.0, 500.0); r2.set_generation_method(True)
        o2 = orc.OracleRandField(orc.RFParams(8e3, 30e3, 12e3, 40e3, 30, 90, nug, model, iso, None), 5, pairs, masks, 500.0)
        for _ in range(10):
            assert np.array_equal(r2.get_rfblock(), o2.get_rfblock())


def test_host_draw_chunk_equals_oracle_trace(golden_dir):
    """chain_crf_gpu._draw_chunk consumes the two generators exactly like the reference loop (fixture F1)."""
    g = np.load(golden_dir / "f1_chain64_standard.npz")
    prob, ch, rf = synthetic.template(64)
    cp = dict(ch.__dict__); cp["rng_seed"] = 7; cp["initial_bed"] = prob["bed"]
    rp = dict(rf.__dict__); rp["rng_seed"] = 7
    c = MCMC_gpu.init_lsc_chain_by_instance(cp)
    r = MCMC_gpu.initiate_RF_by_instance(rp)
    assert isinstance(c, MCMC_gpu.chain_crf_gpu)
    si, ce, u, fields = c._draw_chunk(r, 299)
    assert np.array_equal(si, g["size_idx"]) and np.array_equal(ce, g["centre"]) and np.array_equal(u, g["u"])
    f3 = np.load(golden_dir / "f3_fields64.npz")
    for i in range(5):
        assert np.array_equal(fields[i], f3[f"field{i}"])


def test_reference_error_behaviour():
    prob, ch, rf = synthetic.template(64)
    with pytest.raises(Exception, match="shape of bed"):
        MCMC_gpu.chain_crf_gpu(prob["xx"], prob["yy"], prob["bed"], prob["surf"][:10], prob["velx"], prob["vely"],
                               prob["dhdt"], prob["smb"], prob["cond_bed"], prob["data_mask"],
                               prob["grounded_ice_mask"], 500.0)
    with pytest.raises(ValueError, match="region_mask"):
        ch.set_update_region(True, np.zeros((3, 3)))
    with pytest.raises(ValueError, match="block_type"):
        ch.set_update_type("nope")
    with pytest.raises(TypeError, match="RandField"):
        ch.run(5, object(), plot=False, progress_bar=False)
    with pytest.raises(Exception, match="valid model_name"):
        MCMC_gpu.RandField(1, 2, 1, 2, 1, 2, 0, "Cubic", True)
    with pytest.raises(Exception, match="smoothness"):
        MCMC_gpu.RandField(1, 2, 1, 2, 1, 2, 0, "Matern", True)
    with pytest.raises(ValueError, match="Seed"):
        MCMC_gpu.RandField(1, 2, 1, 2, 1, 2, 0, "Gaussian", True, rng_seed="x")
    r = MCMC_gpu.RandField(1, 2, 1, 2, 1, 2, 0, "Gaussian", True)
    with pytest.raises(Exception, match="set_block_sizes"):
        r.set_weight_param(2, 0, 6, 1, 1.0, 1.0)
    rf.set_generation_method(False)
    with pytest.raises(NotImplementedError):
        rf.get_rfblock()
    # loss() host helper == oracle
    mc = orc.mc_residual(prob["bed"], prob["surf"], prob["velx"], prob["vely"], prob["dhdt"], prob["smb"], 500.0)
    assert ch.loss(mc, None)[0] == orc.gaussian_loss(mc, ch.mc_region_mask, 5.0)[0]


def test_masks_must_be_binary():
    from mcmc_gpu_amd.engine import binary_mask
    assert binary_mask(np.array([[True, False]]), "m").dtype == np.uint8
    assert binary_mask(np.array([[1.0, 0.0]]), "m").tolist() == [[1, 0]]
    with pytest.raises(ValueError):
        binary_mask(np.array([[2, 0]]), "m")
    with pytest.raises(ValueError):
        binary_mask(np.array([[np.nan, 0]]), "m")


def test_shard_bounds_partition():
    for n, w in ((1024, 8), (10, 4), (3, 8), (8192, 8), (7, 1)):
        cover = []
        for r in range(w):
            lo, hi = parallel.shard_bounds(n, w, r)
            cover += list(range(lo, hi))
            assert 0 <= hi - lo <= -(-n // w)
        assert cover == list(range(n))
    with pytest.raises(ValueError):
        parallel.shard_bounds(4, 2, 2)


def test_bench_algorithmic_bytes_formula():
    sys.path.insert(0, str(_lib.PKG_DIR.parent))
    import bench
    blocks = np.array([[[128, 128, 80, 80], [0, 0, 50, 56]]])     # unclipped 6400 cells; corner: 25 x 28 cells
    acc = np.array([[1, 0]])
    b = bench.algorithmic_bytes(blocks, acc, 256, 256)
    assert b == (16 * 6400 + 16 * 6400 + 8 * 6400) + 16 * (25 * 28)


def test_philox_oracle_proposal_distribution_matches_reference_spectral():
    """The Hermitian half-plane construction of the Philox generator has the same field distribution as the
    reference's Re(ifft2((N1 + i N2) sqrt(S))) (mcmc_oracle.spectral_field, pinned to the reference by F1/F3):
    compare the ensemble covariance at a few lags on a fixed block and range."""
    rfp = orc.RFParams(20e3, 20e3, 20e3, 20e3, 90, 90, 0.0, "Matern", True, 0.9125)   # fixed range and scale
    pairs = np.array([[16], [12]]); masks = [np.ones((12, 16))]
    rfp.resolution = 500.0
    n = 3000
    A = np.stack([po.proposal(99, s, rfp, pairs, masks, np.array([0]), 64, 500.0)["field"] for s in range(n)])
    g = np.random.default_rng(5)
    B = np.stack([orc.spectral_field(g, rfp, (12, 16), 500.0) for _ in range(n)])
    for lag in ((0, 0), (0, 1), (1, 0), (2, 3), (5, 7)):
        ca = np.mean(A[:, :12 - lag[0], :16 - lag[1]] * A[:, lag[0]:, lag[1]:])
        cb = np.mean(B[:, :12 - lag[0], :16 - lag[1]] * B[:, lag[0]:, lag[1]:])
        assert abs(ca - cb) < 0.04 * 900.0, (lag, ca, cb)      # variance is scale^2 = 900; MC error ~1-2 %
    assert abs(A.mean()) < 1.0 and abs(B.mean()) < 1.0


def test_highvel_boundary_equals_reference(golden_dir):
    """Topography.get_highvel_boundary (host path: KD-tree distance) against fixture F9 = the reference's
    O(N^2) double loop on the same inputs."""
    from mcmc_gpu_amd import Topography
    g = np.load(golden_dir / "f9_highvel_boundary.npz")
    m = Topography.get_highvel_boundary(g["velx"], g["vely"], float(g["threshold"]), g["grounded"], g["ocean"],
                                        float(g["distance_max"]), g["xx"], g["yy"], smooth_mode=int(g["smooth_mode"]))
    assert m.dtype == bool and np.array_equal(m, g["mask_final"])
    assert 0 < m.sum() < m.size


def test_draw_workers_equal_serial_draws():
    """run_many_replay's host draw workers (child processes owning the generators of their chains, fields through shared
    memory, two chunks into alternating buffers) produce exactly the draws chain_crf_gpu._draw_chunk makes in-process --
    the reference's per-generator order -- and hand back the generators' final states."""
    from multiprocessing import shared_memory
    prob, ch, rf = synthetic.template(64)
    H, W = prob["bed"].shape
    seeds, n, chunks = [11, 12, 13], 5, 3
    stride = int(rf.pairs[0].max() * rf.pairs[1].max())
    shape = (len(seeds), n, stride)
    shms = [shared_memory.SharedMemory(create=True, size=int(np.prod(shape)) * 8) for _ in range(2)]
    pool = None
    try:
        bufs = [np.ndarray(shape, dtype=np.float64, buffer=m.buf) for m in shms]
        rf_param = {k: v for k, v in rf.__dict__.items() if k not in ("rng", "_last_size_idx")}
        st = [np.random.default_rng(seed=s).bit_generator.state for s in seeds]
        pool = MCMC_gpu.DrawWorkers(2, len(seeds), rf_param, H, W, ch.update_in_region, np.asarray(ch.region_mask),
                                    [m.name for m in shms], shape, st, st)
        got = [[] for _ in seeds]
        pool.request(0, n)
        for k in range(chunks):
            res = pool.collect()
            for slot, si, ce, u in res:
                got[slot].append((si, ce, u, bufs[k & 1][slot].copy()))
            if k + 1 < chunks:
                pool.request((k + 1) & 1, n)
        states = pool.states()
        pool.close(); pool = None
        for c, s in enumerate(seeds):
            rf_c = MCMC_gpu.initiate_RF_by_instance(dict(rf_param, rng_seed=s))
            rng = np.random.default_rng(seed=s)
            for k in range(chunks):
                si, ce, u, fields = MCMC_gpu.draw_chunk(rf_c, rng, n, H, W, ch.update_in_region, ch.region_mask)
                assert np.array_equal(si, got[c][k][0]) and np.array_equal(ce, got[c][k][1]) and np.array_equal(u, got[c][k][2])
                for j, f in enumerate(fields):
                    assert np.array_equal(got[c][k][3][j, :f.size], f.ravel())
            assert states[c] == (rf_c.rng.bit_generator.state, rng.bit_generator.state)
    finally:
        if pool is not None:
            pool.close(kill=True)
        for m in shms:
            m.close()
            m.unlink()
