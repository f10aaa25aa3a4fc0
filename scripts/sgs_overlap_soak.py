"""Soak of gsm_sgs_iterate's record streams: long runs with the records made ahead (two record streams, eight sets of scratch, draws
made ahead) against the same runs with every launch in the reference's order on one stream -- every output must be identical bit for bit.
    python scripts/sgs_overlap_soak.py [--chains 4] [--iters 4000] [--pcg64]"""
import argparse, os, sys
sys.path.insert(0, '.'); sys.path.insert(0, 'oracle'); sys.path.insert(0, 'tests')
import numpy as np

ap = argparse.ArgumentParser()
ap.add_argument('--chains', type=int, default=4); ap.add_argument('--iters', type=int, default=4000)
ap.add_argument('--grid', type=int, default=64); ap.add_argument('--pcg64', action='store_true')
a = ap.parse_args()
from mcmc_gpu_amd import sgs, synthetic
res = []
for overlap in ('1', '0'):
    os.environ['GSM_SGS_OVERLAP'] = overlap; os.environ['GSM_SGS_DRAW_AHEAD'] = overlap; os.environ['GSM_SGS_TAIL_QT'] = overlap
    prob, ch = synthetic.sgs_template(a.grid)
    beds = [prob["bed"] + np.random.default_rng(40 + i).normal(0, 3, prob["bed"].shape) for i in range(a.chains)]
    rngs = [np.random.default_rng(900 + i) for i in range(a.chains)]
    out, rngs = sgs.run_many_sgs(ch, beds, rngs, a.iters, philox_seeds=None if a.pcg64 else [7000 + i for i in range(a.chains)], pcg64=a.pcg64)
    res.append((out, [r.bit_generator.state for r in rngs]))
(x, sx), (y, sy) = res
same = sx == sy
for p, q in zip(x, y):
    for k in (0, 3, 4, 5, 6):
        same = same and np.array_equal(np.asarray(p[k], dtype=float), np.asarray(q[k], dtype=float), equal_nan=True)
acc = float(np.mean([o[4].mean() for o in x]))
print(f"{a.chains} chains x {a.iters} iterations ({'pcg64' if a.pcg64 else 'philox'}), accept {acc:.3f}: records made ahead == one stream: {same}")
sys.exit(0 if same else 1)
