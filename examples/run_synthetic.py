#!/usr/bin/env python3
"""End-to-end example on the synthetic 64x64 problem: the reference's driver workflow with mcmc_gpu_amd.

    python examples/run_synthetic.py [output_dir]

1. builds the template chain + RandField through the reference's setters,
2. runs 4 chains x 2 segments in replay mode (NumPy draws, results identical to the CPU reference) through
   largeScaleChain_mp -- checkpoint files land in <output_dir>/LargeScaleChain/<seed>/,
3. runs 256 chains in Philox mode (device draws) and prints the pooled accept rate and posterior-mean bed.
"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from mcmc_gpu_amd import MCMC_gpu, driver, synthetic  # noqa: E402

out = Path(sys.argv[1] if len(sys.argv) > 1 else "./example_output")
prob, chain, rf = synthetic.template(64)

seeds = [1111, 2222, 3333, 4444]
beds = list(synthetic.initial_beds(prob, 4))
for segment in range(2):                                   # second call resumes from the files of the first
    res = driver.largeScaleChain_mp(4, None, chain, rf, beds, seeds, [1000] * 4, output_path=str(out))
print("replay mode:", [f"{r[4].mean():.3f}" for r in res], "accept rates;",
      sorted(p.name for p in (out / "LargeScaleChain" / "1111").iterdir()))

chain.set_rng_mode("philox")
n = 256
res = MCMC_gpu.run_many(chain, rf, synthetic.initial_beds(prob, n), list(range(100, 100 + n)), 2001, batch=32)
acc = np.mean([r[4][1:].mean() for r in res])
post_mean = np.mean([r[0] for r in res], axis=0)
print(f"philox mode: {n} chains x 2000 steps, pooled accept rate {acc:.3f}, "
      f"loss {np.mean([r[3][0] for r in res]):.1f} -> {np.mean([r[3][-1] for r in res]):.1f}, "
      f"posterior-mean bed range [{post_mean.min():.1f}, {post_mean.max():.1f}] m")
