"""Soak of the small-scale chain in Philox mode: 64 chains x 5000 iterations without and with the normal-score transformer; beds finite, every\nchain resampled.  (The loss of a smooth initial bed RISES towards the level of the SGS realisations: the prior adds small-scale roughness.)"""
import sys
sys.path.insert(0, '.')
import numpy as np, time
from mcmc_gpu_amd import sgs, synthetic
for transform in (False, True):
    prob, ch = synthetic.sgs_template(64, transform=transform)
    n = 64
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(n)]
    t0 = time.time()
    out, _ = sgs.run_many_sgs(ch, beds, [np.random.default_rng(i) for i in range(n)], 5000, philox_seeds=[100 + i for i in range(n)])
    dt = time.time() - t0
    acc = np.mean([o[4].mean() for o in out]); fin = all(np.isfinite(o[0]).all() for o in out)
    l0 = np.mean([o[3][0] for o in out]); l1 = np.mean([o[3][-1] for o in out])
    res_ok = all(o[5].sum() > 0 for o in out)
    print(f"transform={transform}: {n} chains x 5000 iterations in {dt:.1f} s ({n*5000/dt:.0f} chain-it/s), accept {acc:.3f}, beds finite {fin}, loss {l0:.1f} -> {l1:.1f}, resampled ok {res_ok}")
    assert fin and res_ok
