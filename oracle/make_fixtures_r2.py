"""TEST INFRASTRUCTURE ONLY -- round-2 golden vectors (build container only; needs /root/reference).

F3b: spectral_synthesis_field of the imported reference (gstatsMCMC/MCMC.py:176-254) at the headline block sizes
(50-80 cells), all three covariance models, isotropic / anisotropic ranges, with and without nugget.  The fixture holds
the generator seed, the parameters and the reference's OUTPUT field only; the noise is regenerated in the test from the
seed with NumPy's Generator in the documented draw order (oracle/mcmc_oracle.spectral_draws), so the planes themselves
need not be stored.  Before writing, the oracle restatement is asserted bit-identical to the reference on every case.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fixtures_r2.py
"""
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import mcmc_oracle as orc  # noqa: E402
import ref_loader  # noqa: E402

GOLD = HERE.parent / "tests" / "golden"

# (model, isotropic, nugget_max, smoothness, (bh, bw), seed)
CASES = [
    ("Matern", True, 0.0, 0.9125, (80, 80), 101),
    ("Matern", True, 0.0, 0.9125, (50, 56), 102),
    ("Gaussian", False, 4.0, None, (64, 72), 103),
    ("Exponential", True, 0.0, None, (72, 50), 104),
    ("Matern", False, 2.0, 1.5, (56, 80), 105),
    ("Exponential", False, 1.0, None, (80, 64), 106),
]
RES = 500.0


def main():
    M, _, _, _ = ref_loader.load_reference()
    out = {"n_cases": len(CASES), "resolution": RES}
    for i, (model, iso, nug, nu, shape, seed) in enumerate(CASES):
        p = orc.RFParams(10e3, 50e3, 12e3, 40e3, 50, 150, nug, model, iso, nu)
        rf = M.RandField(p.range_min_x, p.range_max_x, p.range_min_y, p.range_max_y, p.scale_min, p.scale_max,
                         p.nugget_max, p.model_name, p.isotropic, smoothness=p.smoothness, rng_seed=seed)
        ref = M.spectral_synthesis_field(rf, shape, res=RES)
        g = np.random.default_rng(seed=seed)
        d = orc.spectral_draws(g, p, shape)
        mine = orc.spectral_from_draws(d, p, shape, RES)
        if not np.array_equal(ref, mine):
            raise AssertionError(f"F3b case {i}: oracle differs from the reference")
        assert g.bit_generator.state == rf.rng.bit_generator.state, f"F3b case {i}: draw count differs"
        out[f"field{i}"] = ref
        out[f"params{i}"] = np.array([p.range_min_x, p.range_max_x, p.range_min_y, p.range_max_y, p.scale_min, p.scale_max,
                                      p.nugget_max, nu if nu else 0.0, 1.0 if iso else 0.0, shape[0], shape[1], seed])
        out[f"model{i}"] = model
        out[f"scalars{i}"] = np.array([d["scale"], d["nug"], d["range_x"], d["range_y"]])
        print(f"F3b case {i}: {model:12s} iso={iso} nug={nug} {shape} seed {seed}: oracle == reference, "
              f"field std {ref.std():.3f}")
    np.savez_compressed(GOLD / "f3b_spectral_blocks.npz", **out)
    print("written", GOLD / "f3b_spectral_blocks.npz", (GOLD / "f3b_spectral_blocks.npz").stat().st_size, "B")


if __name__ == "__main__":
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    main()
