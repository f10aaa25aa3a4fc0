"""TEST INFRASTRUCTURE ONLY -- round-4 golden vector (build container only; needs /root/reference).

F13: a DEEP run of the imported reference's chain_sgs.run (gstatsMCMC/MCMC.py:1599-1911) at the parameters of the reference's own
small-scale driver (smallScaleChain_multiprocessing.py:489-556: set_sgs_param(48, 30e3) at 500 m = 60-cell search half-width,
blocks 5-20, Matern variogram, QuantileTransformer(1000) on the detrended map, Gaussian-filter trend) -- golden F11's cases a / t
hold two accepted iterations each (sigma_mc = 5 accepts 2.7 % on the synthetic problem), so the path where simulated values go
through the inverse transform and the commit back into the chain was pinned twice per case.  Case d: the same configuration on the
tie-free geometry (rows 503.7 m apart: the UNMODIFIED reference is the pin), sigma_mc = 30 (accept rate ~25 %), 160 iterations,
at least 30 accepted.  oracle/sgs_oracle.py is asserted bit-identical to the reference on every output and on the final generator
state before the file is written.

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_fixtures_r4.py
"""
import os
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parent / "tests"))
import ref_loader  # noqa: E402
import sgs_common as sc  # noqa: E402
from make_fixtures import sha  # noqa: E402
from make_fixtures_r3 import run_case  # noqa: E402

GOLD = HERE.parent / "tests" / "golden"


def main():
    M, _, _, C = ref_loader.load_reference()
    H = 64
    out = {"H": H, "v1_p": np.array(sc.DRIVER_V1_P, dtype=np.float64), "tie_free_dy": sc.TIE_FREE_DY, "sigma_mc": 30.0}
    trend, nst = sc.driver_trend_and_transformer(sc.driver_problem(H))
    out.update({"trend_sha": sha(trend), "quantiles_sha": sha(nst.quantiles_)})
    probt = sc.driver_problem(H, dy=sc.TIE_FREE_DY)
    out.update(run_case(M, C, "d", probt, trend, nst, n_iter=160, stable=False, sigma=30.0, first_seed=400, min_accepts=30))
    acc = int(out["d_steps"].sum())
    assert acc >= 30, acc
    np.savez_compressed(GOLD / "f13_sgs_driver_deep64.npz", **out)
    print("written", GOLD / "f13_sgs_driver_deep64.npz", (GOLD / "f13_sgs_driver_deep64.npz").stat().st_size, "B;", acc, "accepted iterations")


if __name__ == "__main__":
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    sys.dont_write_bytecode = True
    main()
