"""Wall time of the drop-in's DEFAULT (replay) mode for many chains: largeScaleChain_mp with all chains of the rank in
one handle + host draw pool, against the round-1 behaviour (one chain after another through lsc_run_wrapper).
    python scripts/replay_batch_bench.py [--chains 64] [--iters 1000] [--grid 256] [--serial-chains 4]"""
import argparse, sys, tempfile, time
sys.path.insert(0, '.')
import numpy as np
from copy import deepcopy
from mcmc_gpu_amd import driver, synthetic

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument('--chains', type=int, default=64); ap.add_argument('--iters', type=int, default=1000)
    ap.add_argument('--grid', type=int, default=256); ap.add_argument('--serial-chains', type=int, default=4)
    ap.add_argument('--workers', type=int, default=0)
    ap.add_argument('--mode', default='replay', help="'replay' (host NumPy draws) or 'pcg64' (the same generator streams on the device)")
    a = ap.parse_args()
    prob, ch, rf = synthetic.template(a.grid)
    seeds = [5000 + i for i in range(a.chains)]
    beds = list(synthetic.initial_beds(prob, a.chains))
    with tempfile.TemporaryDirectory() as td:
        t0 = time.time()
        res = driver.largeScaleChain_mp(a.chains, a.workers, ch, rf, beds, seeds, [a.iters] * a.chains, output_path=td + '/b', n_gpus=1, mode=a.mode)
        t_batched = time.time() - t0
        t0 = time.time()
        ser = []
        for i in range(a.serial_chains):
            cp = deepcopy(ch.__dict__); cp['rng_seed'] = seeds[i]; cp['initial_bed'] = beds[i]
            rp = deepcopy(rf.__dict__); rp['rng_seed'] = seeds[i]
            ser.append(driver.lsc_run_wrapper(cp, rp, dict(n_iter=a.iters, only_save_last_bed=True, info_per_iter=10 ** 9, plot=False,
                                                           progress_bar=False, chain_id=i, tqdm_position=1, seed=seeds[i], output_path=td + '/s')))
        t_serial = (time.time() - t0) * a.chains / a.serial_chains
    same = all(np.array_equal(x, y, equal_nan=True) for i in range(a.serial_chains) for x, y in zip(res[i], ser[i]))
    if a.mode == 'pcg64':
        same = all(np.array_equal(res[i][4], ser[i][4]) and np.array_equal(res[i][6], ser[i][6], equal_nan=True) and
                   np.allclose(res[i][0], ser[i][0], rtol=0, atol=1e-8) for i in range(a.serial_chains))
    how = 'accept masks / blocks identical, beds 1e-8 m' if a.mode == 'pcg64' else 'bit for bit'
    print(f"{a.chains} chains x {a.iters} iterations, {a.grid}^2: batched {a.mode} {t_batched:.2f} s "
          f"({a.chains * (a.iters - 1) / t_batched:.0f} chain-steps/s); one chain after another {t_serial:.1f} s "
          f"(extrapolated from {a.serial_chains} chains) -> {t_serial / t_batched:.1f}x; equal on those chains ({how}): {same}")
