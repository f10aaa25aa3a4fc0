"""'pcg64' draw mode (the reference's NumPy generator streams advanced on the device): rate of MCMC_gpu.run_many_pcg64 alone -- no
checkpoint files -- at the headline geometry, beside the host-drawn replay mode.   python scripts/pcg64_bench.py [--chains 1024] [--iters 513]"""
import argparse, sys, time
sys.path.insert(0, '.')
import numpy as np
from mcmc_gpu_amd import MCMC_gpu, synthetic

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=1024); ap.add_argument('--iters', type=int, default=513)
    ap.add_argument('--grid', type=int, default=256); ap.add_argument('--batch', type=int, default=0)
    ap.add_argument('--replay', action='store_true', help='also time run_many_replay (host draws) on min(chains, 64) chains')
    a = ap.parse_args()
    prob, ch, rf = synthetic.template(a.grid)
    beds = np.stack(list(synthetic.initial_beds(prob, a.chains)))
    st = [np.random.default_rng(seed=900 + i).bit_generator.state for i in range(a.chains)]
    MCMC_gpu.run_many_pcg64(ch, rf, beds[:8], st[:8], st[:8], 9)                    # warm-up
    t0 = time.time()
    tm = {}
    out, _, _ = MCMC_gpu.run_many_pcg64(ch, rf, beds, st, st, a.iters, batch=a.batch or None, timing=tm)
    dt = time.time() - t0
    acc = np.mean([o[4][1:].mean() for o in out])
    print(f"pcg64 mode: {a.chains} chains x {a.iters - 1} steps at {a.grid}^2: {dt:.2f} s = {a.chains * (a.iters - 1) / dt / 1e6:.3f} M chain-steps/s "
          f"(incl. engine setup and result download); state resident in HBM: {tm['loop_seconds']:.3f} s = "
          f"{a.chains * (a.iters - 1) / tm['loop_seconds'] / 1e6:.3f} M chain-steps/s (fused {tm['fused']}, batch {tm['batch']}), accept {acc:.3f}")
    if a.replay:
        n = min(a.chains, 64)
        t0 = time.time()
        MCMC_gpu.run_many_replay(ch, rf, beds[:n], st[:n], st[:n], a.iters)
        dt = time.time() - t0
        print(f"replay mode (host NumPy draws, {n} chains): {n * (a.iters - 1) / dt / 1e3:.1f} k chain-steps/s")
