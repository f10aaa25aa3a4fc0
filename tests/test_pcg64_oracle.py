"""not gpu: oracle/pcg64_oracle.py (the CPU restatement of NumPy's PCG64 generator methods the reference's chains consume) against
NumPy itself, bit for bit and state for state; the generated ziggurat table header; the fdlibm log1p restatement against libm."""
import math
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "oracle"))
import pcg64_oracle as po  # noqa: E402


def _pair(seed):
    g = np.random.default_rng(seed=seed)
    return g, po.Pcg64.from_numpy(g)


@pytest.mark.parametrize("seed", [0, 7, 12345678901234567890])
def test_bit_generator_and_scalar_methods_equal_numpy(seed):
    g, m = _pair(seed)
    raw = g.bit_generator.random_raw(1000)
    assert [int(x) for x in raw] == [m.next_uint64() for _ in range(1000)]
    # the reference's per-step call sequence (MCMC.py:755, :199-207, :242, :251; :1254-1258, :1336), 32-bit halves cached across calls
    for _ in range(300):
        assert int(g.integers(low=0, high=25, size=1)[0]) == m.integers(0, 25)
        assert g.uniform(50.0, 150.0) == m.uniform(50.0, 150.0)
        assert g.uniform(0.0, 0.0) == m.uniform(0.0, 0.0)
        assert g.uniform(10e3, 50e3) == m.uniform(10e3, 50e3)
        a = g.normal(size=(3, 5))
        assert np.array_equal(a, np.array([m.normal() for _ in range(15)]).reshape(3, 5))
        b = g.normal(0, math.sqrt(2.5), size=(2, 2))
        assert np.array_equal(b, np.array([m.normal(0, math.sqrt(2.5)) for _ in range(4)]).reshape(2, 2))
        assert int(g.integers(low=0, high=256, size=1)[0]) == m.integers(0, 256)
        assert int(g.integers(low=0, high=199, size=1)[0]) == m.integers(0, 199)
        assert g.random() == m.random()
        assert g.bit_generator.state == m.numpy_state()


def test_standard_normal_stream_incl_wedge_and_tail():
    g, m = _pair(2024)
    n = 300_000
    ref = g.standard_normal(n)
    mine = np.array([m.standard_normal() for _ in range(n)])
    assert np.array_equal(ref, mine)
    assert g.bit_generator.state == m.numpy_state()
    assert m.n_raw > n * 1.005                   # the slow paths were exercised (~1.4 % extra raw draws)
    assert np.abs(mine).max() > po.ZIG_R         # ... the tail too


def test_log1p_restatement_equals_libm_on_the_tail_inputs():
    rng = np.random.default_rng(5)
    u = np.concatenate([rng.random(400_000), rng.random(100_000) * 1e-3, 1.0 - rng.random(100_000) * 1e-6,
                        [0.0, 2.0 ** -53, 2.0 ** -30, 2.0 ** -29, 0.2928, 0.2929, 0.29289321881345254, 0.5, 1 - 2.0 ** -53]])
    # libm's scalar log1p is what numpy's ziggurat calls (npy_log1p); numpy.log1p the ufunc may run a vector library instead
    ref = np.array([math.log1p(-float(v)) for v in u])
    mine = np.array([po.log1p_fdlibm(-float(v)) for v in u])
    assert np.array_equal(ref, mine), int((ref != mine).sum())


def test_shuffle_and_offset_integers_equal_numpy():
    """The small-scale chain's calls (MCMC.py:1750-1757 block sizes with low > 0, :128 rng.shuffle of the block's (row, col) list)."""
    g, m = _pair(99)
    for n in (1, 2, 3, 17, 64, 255, 361, 1000):
        for _ in range(5):
            a = np.arange(2 * n).reshape(n, 2)
            b = a.copy()
            g.shuffle(a)
            m.shuffle(b)
            assert np.array_equal(a, b)
            assert int(g.integers(low=5, high=20, size=1)[0]) == m.integers(5, 20)
            assert g.random() == m.random()
            assert g.bit_generator.state == m.numpy_state()


def test_state_packing_round_trip_and_mode_names():
    """Host plumbing of the 'pcg64' draw mode: numpy state dict <-> the six 64-bit words handed to gsm_draw_pcg64; mode names."""
    sys.path.insert(0, str(ROOT))
    from mcmc_gpu_amd.engine import GsmEngine
    from mcmc_gpu_amd import MCMC_gpu, sgs, synthetic
    gens = [np.random.default_rng(seed=s) for s in (1, 2 ** 63 + 5)]
    gens[1].integers(low=0, high=7, size=1)
    words = GsmEngine.pack_pcg64_states(gens)
    assert words.shape == (2, 6) and words.dtype == np.uint64 and words[1, 4] == 1
    back = GsmEngine.unpack_pcg64_states(words)
    assert back == [g.bit_generator.state for g in gens]
    m = po.Pcg64(int(words[1, 0]) | (int(words[1, 1]) << 64), int(words[1, 2]) | (int(words[1, 3]) << 64), int(words[1, 4]), int(words[1, 5]))
    assert m.integers(0, 100) == int(gens[1].integers(low=0, high=100, size=1)[0])
    with pytest.raises(ValueError):
        GsmEngine.pack_pcg64_states([np.random.Generator(np.random.MT19937(3))])
    prob, ch, rf = synthetic.template(64)
    ch.set_rng_mode('pcg64'); assert ch.rng_mode == 'pcg64'
    with pytest.raises(ValueError):
        ch.set_rng_mode('numpy')
    _, s_ch = synthetic.sgs_template(32, transform=False)
    s_ch.set_rng_mode('pcg64'); assert s_ch.rng_mode == 'pcg64'
