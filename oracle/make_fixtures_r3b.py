"""TEST INFRASTRUCTURE ONLY -- golden F12 (build container only; needs /root/reference).

F12: the reference's module-level MCMC.sgs (gstatsMCMC/MCMC.py:91-173) called directly, with ordinary AND simple kriging
(ktype='ok' / 'sk': _krige.py:5-44 / :46-81) -- chain_sgs.run never passes ktype, so F10 / F11 only cover 'ok'.  A 48 x 48
grid with rows 503.7 m apart (no equidistant neighbours: the unmodified reference is the pin, tests/sgs_common.TIE_FREE_DY),
values = the standardised synthetic bed, one block of NaN cells crossed by conditioning lines:
  ok   default num_points (20), exponential variogram, sim_mask None (the shuffle runs over all 2304 cells, MCMC.py:71, :128)
  sk   simple kriging, Matern variogram, 32 neighbours
  skm  simple kriging, anisotropic spherical variogram with sill 1.3, sim_mask = part of the block (the other NaN cells stay NaN
       and are never neighbours)
Before writing, oracle/sgs_oracle.sgs is asserted bit-identical to the reference on the output grid and the final generator
state of every case.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fixtures_r3b.py
"""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent / "tests"))
import ref_loader  # noqa: E402
import sgs_oracle as so  # noqa: E402
import sgs_common as sc  # noqa: E402
from make_fixtures import quiet, warnings_off  # noqa: E402

GOLD = HERE.parent / "tests" / "golden"


def main():
    M, _, _, _ = ref_loader.load_reference()
    out = {}
    for tag in sc.F12_CASES:
        xx, yy, grid, vario, kw, seed = sc.f12_case(tag)
        rng_r, rng_o = np.random.default_rng(seed), np.random.default_rng(seed)
        with quiet(), warnings_off():
            ref = M.sgs(xx, yy, grid.copy(), dict(vario), seed=rng_r, quiet=True, **kw)
        so.TIE_LOG = []
        with warnings_off():
            mine = so.sgs(xx, yy, grid.copy(), dict(vario), rng=rng_o, **kw)
        ties, so.TIE_LOG = len(so.TIE_LOG), None
        assert ties == 0, f"F12{tag}: the geometry is not tie-free"
        assert np.array_equal(ref, mine, equal_nan=True), f"F12{tag}: oracle != reference"
        assert rng_r.bit_generator.state == rng_o.bit_generator.state, f"F12{tag}: generator state differs"
        n_sim = int(np.isnan(grid).sum() - np.isnan(ref).sum())
        print(f"F12{tag}: oracle == reference; {n_sim} cells simulated, {int(np.isnan(ref).sum())} stay NaN")
        out[f"{tag}_out"] = ref
        out[f"{tag}_rng_state"] = json.dumps(rng_r.bit_generator.state)
        out[f"{tag}_n_sim"] = n_sim
    np.savez_compressed(GOLD / "f12_sgs_function_ok_sk.npz", **out)
    print("written", GOLD / "f12_sgs_function_ok_sk.npz", (GOLD / "f12_sgs_function_ok_sk.npz").stat().st_size, "B")


if __name__ == "__main__":
    main()
