import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np
import mcmc_oracle as orc, philox_oracle as po
from gpu_common import make_engine
for model, iso, nug in [("Gaussian", True, 0.0), ("Gaussian", False, 0.0), ("Gaussian", True, 4.0), ("Matern", False, 0.0), ("Matern", True, 4.0)]:
    rfp = orc.RFParams(10e3, 50e3, 12e3, 40e3, 50, 150, nug, model, iso, 0.9125 if model == "Matern" else None)
    eng, prob, cfg, pairs, masks, _ = make_engine(64, 1, rf_params=rfp)
    rfp.resolution = prob["resolution"]
    out = eng.propose_philox(8, 1000, [7], rfp)
    centres = np.flatnonzero(cfg.region_mask.ravel() == 1)
    worst = 0
    for s in range(8):
        e = po.proposal(7, 1000 + s, rfp, pairs, masks, centres, 64, prob["resolution"])
        bh, bw = e["field"].shape
        f = out["fields"][0, s, : bh * bw].cpu().numpy().reshape(bh, bw)
        d = np.abs(f - e["field"])
        worst = max(worst, d.max() / e["scale"])
        if d.max() > 1e-9:
            i = np.unravel_index(d.argmax(), d.shape)
            print("   step", s, "shape", (bh, bw), "max", d.max(), "at", i, "val", e["field"][i], "nviol", (d > 1e-9).sum())
    print(model, iso, nug, "worst rel-to-scale", worst)
    eng.close()
