import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from mcmc_gpu_amd import synthetic
prob, ch, rf = synthetic.template(256)
eng = ch._make_engine(rf, 1024, 0)
beds0 = synthetic.initial_beds(prob, 1024)
seeds = list(range(7, 7 + 1024))
p = eng.rf_struct(rf)
res = {}
for mode in (1, 2):
    eng.set_fused(mode)
    eng.set_state(beds0)
    eng.enable_timing(True)
    eng.run_philox(64, 0, seeds, p, batch=32)
    loss, acc, blk = eng.run_philox(256, 64, seeds, p, batch=32)
    tm = eng.last_timing()
    res[mode] = (loss, acc, eng.beds.cpu().numpy().copy())
    print(f"mode {mode}: last_run_fused={eng.last_run_fused()} kernel {tm['step_ms']:.2f} ms for 256 steps -> {1024*256/tm['step_ms']/1e3:.2f} M chain-steps/s, accept {acc.mean():.4f}")
print("bit-identical:", all(np.array_equal(a, b) for a, b in zip(res[1], res[2])))
