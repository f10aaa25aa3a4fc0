"""Counterparts of the reference's multi-chain driver functions (largeScaleChain_multiprocessing_GPU.py):

    largeScaleChain_mp(n_chains, n_workers, largeScaleChain, rf, initial_beds, rng_seeds, n_iters, output_path)   :22-104
    lsc_run_wrapper(param_chain, param_rf, param_run)                                                            :106-246

Same arguments, same return value (a list of per-chain result tuples, what Pool.starmap returns), same per-seed
checkpoint files under  <output_path>/LargeScaleChain/<str(seed)[:6]>/ :

    bed_{k}k.npy            last bed after k*1000 cumulative iterations                      (:227)
    results_{k}k.npz        loss_mc, loss_data, loss, steps, resampled_times, blocks_used    (:229-237), concatenated
                            over segments (:213-220); the previous results file and current_iter.txt are deleted (:240-242)
    current_iter.txt        cumulative iteration count                                       (:244)
    RNGState_RandField.txt, RNGState_chain.txt   JSON of Generator.bit_generator.state      (:207-210)
    RNGState_philox.txt     {"key", "step"} of the counter-based generator (this build only)

Where the reference maps chains to OS processes (one chain per pool worker), this maps them to GPUs: all chains of a
rank run in one libgsm_hip handle; with torch.distributed initialised the chains are sharded contiguously over the
ranks (no communication inside the step loop) and the results are all-gathered once at the end over RCCL.
`n_workers` is accepted for signature compatibility and ignored.
"""
from __future__ import annotations

import contextlib
import io
import json
import time
from copy import deepcopy
from pathlib import Path

import numpy as np

from . import MCMC_gpu, parallel


def _seed_folder(output_path, seed):
    return Path(output_path) / f'{str(seed)[:6]}'


def _load_previous(seed_folder):
    """Resume state of a seed folder (reference :141-186) or None."""
    marker = seed_folder / 'current_iter.txt'
    if not marker.exists():
        return None
    cumulative = int(np.loadtxt(marker))
    k = int(cumulative / 1000)
    with np.load(seed_folder / f'results_{k}k.npz') as r:
        prev = {key: r[key] for key in ('loss_mc', 'loss_data', 'loss', 'steps', 'resampled_times', 'blocks_used')}
    st = dict(cumulative=cumulative, bed=np.load(seed_folder / f'bed_{k}k.npy'), results=prev,
              delete=[seed_folder / f'results_{k}k.npz', marker])
    for name, key in (('RNGState_RandField.txt', 'rf_state'), ('RNGState_chain.txt', 'chain_state'),
                      ('RNGState_philox.txt', 'philox')):
        f = seed_folder / name
        if f.exists():
            with open(f, 'r') as fh:
                st[key] = json.load(fh)
    return st


def _save_segment(seed_folder, result, n_iter, prev, rf_state, chain_state, philox_state):
    """Write the checkpoint files of one finished segment (reference :203-244)."""
    seed_folder.mkdir(parents=True, exist_ok=True)
    beds, loss_mc, loss_data, loss, steps, resampled, blocks = result[:7]
    with open(seed_folder / 'RNGState_RandField.txt', 'w') as fh:
        json.dump(rf_state, fh)
    with open(seed_folder / 'RNGState_chain.txt', 'w') as fh:
        json.dump(chain_state, fh)
    with open(seed_folder / 'RNGState_philox.txt', 'w') as fh:
        json.dump(philox_state, fh)
    cumulative = 0
    if prev is not None:
        p = prev['results']
        loss_mc = np.concatenate([p['loss_mc'], loss_mc])
        loss_data = np.concatenate([p['loss_data'], loss_data])
        loss = np.concatenate([p['loss'], loss])
        steps = np.concatenate([p['steps'], steps])
        resampled = p['resampled_times'] + resampled
        blocks = np.vstack([p['blocks_used'], blocks])
        cumulative = prev['cumulative']
    cumulative += n_iter
    label = f'{cumulative // 1000}k'
    np.save(seed_folder / f'bed_{label}.npy', beds)
    np.savez_compressed(seed_folder / f'results_{label}.npz', loss_mc=loss_mc, loss_data=loss_data, loss=loss,
                        steps=steps, resampled_times=resampled, blocks_used=blocks)
    if prev is not None:
        for f in prev['delete']:
            if f.exists() and f.name != f'results_{label}.npz':
                f.unlink()
    np.savetxt(seed_folder / 'current_iter.txt', [cumulative], fmt='%d')


def lsc_run_wrapper(param_chain, param_rf, param_run):
    """Rebuild one chain + RandField from parameter dicts, resume from its seed folder if present, run one segment
    on the GPU, write the checkpoint files, return chain.run's tuple (reference :106-246)."""
    with contextlib.redirect_stdout(io.StringIO()):
        chain = MCMC_gpu.init_lsc_chain_by_instance(param_chain)
        rf1 = MCMC_gpu.initiate_RF_by_instance(param_rf)
    output_path = param_run.get('output_path', './Data/LargeScaleChain')
    seed = param_run['seed']
    n_iter = param_run['n_iter']
    folder = _seed_folder(output_path, seed)
    prev = _load_previous(folder)
    if prev is not None:
        chain.initial_bed = prev['bed']
        if 'rf_state' in prev:
            rf1.rng.bit_generator.state = prev['rf_state']
        if 'chain_state' in prev:
            chain.rng.bit_generator.state = prev['chain_state']
        if 'philox' in prev:
            chain.philox_step = int(prev['philox']['step'])
    chain.chain_id = param_run.get('chain_id', 'Unknown')
    chain.tqdm_position = param_run.get('tqdm_position', 0)
    chain.seed = param_run.get('seed', 'Unknown')
    result = chain.run(n_iter=n_iter, RF=rf1, only_save_last_bed=param_run['only_save_last_bed'],
                       info_per_iter=param_run['info_per_iter'], plot=param_run['plot'],
                       progress_bar=None if not param_run.get('verbose', False) else param_run['progress_bar'])
    _save_segment(folder, result, n_iter, prev, rf1.rng.bit_generator.state, chain.rng.bit_generator.state,
                  {'key': chain._philox_seed(), 'step': int(chain.philox_step)})
    return result


def _make_params(largeScaleChain, rf, i, initial_beds, rng_seeds, n_iters, output_path):
    chain_param = deepcopy(largeScaleChain.__dict__)
    chain_param['rng_seed'] = rng_seeds[i]
    chain_param['initial_bed'] = initial_beds[i]
    rf_param = deepcopy(rf.__dict__)
    rf_param['rng_seed'] = rng_seeds[i]
    run_param = dict(n_iter=n_iters[i], only_save_last_bed=True, info_per_iter=1000, plot=False, progress_bar=False,
                     chain_id=i, tqdm_position=i + 1, seed=rng_seeds[i],
                     output_path=str(Path(output_path) / 'LargeScaleChain'))
    return chain_param, rf_param, run_param


def largeScaleChain_mp(n_chains, n_workers, largeScaleChain, rf, initial_beds, rng_seeds, n_iters,
                       output_path='./Data/output', mode=None, batch=8, gather=True):
    """Run n_chains large-scale chains and return the list of their result tuples (reference :22-104).

    mode 'replay' (default when largeScaleChain.rng_mode == 'replay'): every chain draws from its own NumPy
    generators exactly as the reference's pool workers do, so results and checkpoint files equal the CPU driver's.
    mode 'philox': all chains of this rank advance together in one handle with device-generated proposals."""
    tic = time.time()
    mode = mode or getattr(largeScaleChain, 'rng_mode', 'replay')
    rank, _, world = parallel.dist_env()
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not sharded:
        rank, world = 0, 1
    lo, hi = parallel.shard_bounds(n_chains, world, rank)
    base = Path(output_path) / 'LargeScaleChain'
    local = []
    same_len = len(set(int(n_iters[i]) for i in range(lo, hi))) <= 1
    if mode == 'philox' and same_len and hi > lo:
        n_iter = int(n_iters[lo])
        prevs = [_load_previous(_seed_folder(base, rng_seeds[i])) for i in range(lo, hi)]
        steps0 = {int(p['philox']['step']) if (p and 'philox' in p) else 0 for p in prevs}
        if len(steps0) == 1:
            step0 = steps0.pop()
            beds = np.stack([np.asarray(p['bed'] if p else initial_beds[i], dtype=np.float64)
                             for p, i in zip(prevs, range(lo, hi))])
            seeds = [int(rng_seeds[i]) for i in range(lo, hi)]
            local = MCMC_gpu.run_many(largeScaleChain, rf, beds, seeds, n_iter, batch=batch, step0=step0)
            for k, i in enumerate(range(lo, hi)):
                g = np.random.default_rng(seed=rng_seeds[i]).bit_generator.state
                p = prevs[k]
                _save_segment(_seed_folder(base, rng_seeds[i]), local[k], n_iter, p,
                              p.get('rf_state', g) if p else g, p.get('chain_state', g) if p else g,
                              {'key': seeds[k] & 0xFFFFFFFFFFFFFFFF, 'step': step0 + n_iter - 1})
    if not local:
        for i in range(lo, hi):
            cp, rp, runp = _make_params(largeScaleChain, rf, i, initial_beds, rng_seeds, n_iters, output_path)
            cp['rng_mode'] = mode
            _seed_folder(base, rng_seeds[i]).mkdir(parents=True, exist_ok=True)
            local.append(lsc_run_wrapper(cp, rp, runp))
    result = local
    if sharded and gather:
        result = _gather_results(local, n_chains, lo, hi)
    if rank == 0:
        print(f'Completed in {time.time() - tic:.2f} seconds')
    return result


def _gather_results(local, n_chains, lo, hi):
    """All-gather per-chain result tuples (equal n_iter on every chain) so every rank returns the full list."""
    import torch
    dev = torch.device('cuda', torch.cuda.current_device()) if torch.cuda.is_available() else torch.device('cpu')
    out_cols = []
    for col in range(7):
        t = torch.as_tensor(np.stack([np.asarray(r[col], dtype=np.float64) for r in local])).to(dev)
        out_cols.append(parallel.all_gather_chains(t, n_chains).cpu().numpy())
    return [tuple(c[i] for c in out_cols) for i in range(n_chains)]
