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
    label_count = 0
    if prev is not None:
        p = prev['results']
        loss_mc = np.concatenate([p['loss_mc'], loss_mc])
        loss_data = np.concatenate([p['loss_data'], loss_data])
        loss = np.concatenate([p['loss'], loss])
        steps = np.concatenate([p['steps'], steps])
        resampled = p['resampled_times'] + resampled
        blocks = np.vstack([p['blocks_used'], blocks])
        cumulative = prev['cumulative']
        label_count = prev.get('label_count', cumulative)
    cumulative += n_iter
    label = f'{(label_count + n_iter) // 1000}k'
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


def _run_shard(lo, hi, largeScaleChain, rf, initial_beds, rng_seeds, n_iters, output_path, mode, batch, n_workers):
    """The chains [lo, hi) of this rank: one libgsm_hip handle for all of them when they share n_iter (and, in Philox
    mode, the Philox step); otherwise chain by chain through lsc_run_wrapper."""
    base = Path(output_path) / 'LargeScaleChain'
    idx = list(range(lo, hi))
    same_len = len(set(int(n_iters[i]) for i in idx)) <= 1
    if same_len and idx:
        n_iter = int(n_iters[lo])
        prevs = [_load_previous(_seed_folder(base, rng_seeds[i])) for i in idx]
        beds = np.stack([np.asarray(p['bed'] if p else initial_beds[i], dtype=np.float64) for p, i in zip(prevs, idx)])
        fresh = [np.random.default_rng(seed=rng_seeds[i]).bit_generator.state for i in idx]
        rf_st = [p.get('rf_state', g) if p else g for p, g in zip(prevs, fresh)]
        ch_st = [p.get('chain_state', g) if p else g for p, g in zip(prevs, fresh)]
        steps0 = [int(p['philox']['step']) if (p and 'philox' in p) else 0 for p in prevs]
        seeds = [int(rng_seeds[i]) for i in idx]
        if mode == 'philox' and len(set(steps0)) == 1:
            local = MCMC_gpu.run_many(largeScaleChain, rf, beds, seeds, n_iter, batch=batch, step0=steps0[0])
            steps1 = [steps0[0] + n_iter - 1] * len(idx)
        elif mode == 'replay':
            local, rf_st, ch_st = MCMC_gpu.run_many_replay(largeScaleChain, rf, beds, rf_st, ch_st, n_iter,
                                                           n_workers=n_workers if n_workers and n_workers > 0 else None)
            steps1 = steps0
        elif mode == 'pcg64':
            local, rf_st, ch_st = MCMC_gpu.run_many_pcg64(largeScaleChain, rf, beds, rf_st, ch_st, n_iter)
            steps1 = steps0
        else:
            local = None
        if local is not None:
            for k, i in enumerate(idx):
                _save_segment(_seed_folder(base, rng_seeds[i]), local[k], n_iter, prevs[k], rf_st[k], ch_st[k],
                              {'key': seeds[k] & 0xFFFFFFFFFFFFFFFF, 'step': steps1[k]})
            return local
    local = []
    for i in idx:
        cp, rp, runp = _make_params(largeScaleChain, rf, i, initial_beds, rng_seeds, n_iters, output_path)
        cp['rng_mode'] = mode
        _seed_folder(base, rng_seeds[i]).mkdir(parents=True, exist_ok=True)
        local.append(lsc_run_wrapper(cp, rp, runp))
    return local


def _rank_main(rank, world, port, backend, payload_path, result_path, which='largeScaleChain_mp'):
    """Body of one self-started rank (a fresh process): rendezvous on 127.0.0.1, run the shard, all-gather, rank 0 stores
    the full result list for the parent."""
    import os
    import pickle
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR='127.0.0.1',
                      MASTER_PORT=str(port), GSM_DIST_BACKEND=backend)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    import torch
    import torch.distributed as dist
    if 'OMP_NUM_THREADS' not in os.environ:
        torch.set_num_threads(1)          # as torchrun does for its ranks
    parallel.init_distributed(backend)
    if torch.cuda.is_available():
        torch.cuda.set_device(rank if torch.cuda.device_count() > rank else 0)
    with open(payload_path, 'rb') as fh:
        kw = pickle.load(fh)
    res = globals()[which](n_gpus=1, **kw)           # inside an initialised group: runs this rank's shard, gathers
    if rank == 0:
        with open(result_path, 'wb') as fh:
            pickle.dump(res, fh, protocol=4)
    parallel.barrier()
    dist.destroy_process_group()


def _self_launch(n_gpus, kw, which='largeScaleChain_mp'):
    """Start n_gpus ranks of this driver (fresh processes: 'spawn'), one per GPU, as the reference's driver starts its own
    pool workers (largeScaleChain_multiprocessing_GPU.py:47, :84-85); return rank 0's gathered result list."""
    import os
    import pickle
    import socket
    import tempfile
    import torch
    import torch.multiprocessing as tmp_mp
    backend = os.environ.get('GSM_DIST_BACKEND') or ('nccl' if torch.cuda.device_count() >= n_gpus and torch.cuda.device_count() > 0 else 'gloo')
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    with tempfile.TemporaryDirectory(prefix='gsm_launch_') as td:
        payload, result = os.path.join(td, 'payload.pkl'), os.path.join(td, 'result.pkl')
        with open(payload, 'wb') as fh:
            pickle.dump(kw, fh, protocol=4)
        tmp_mp.spawn(_rank_main, args=(n_gpus, port, backend, payload, result, which), nprocs=n_gpus, join=True)   # raises if a rank fails
        with open(result, 'rb') as fh:
            return pickle.load(fh)


def largeScaleChain_mp(n_chains, n_workers, largeScaleChain, rf, initial_beds, rng_seeds, n_iters,
                       output_path='./Data/output', mode=None, batch=8, gather=True, n_gpus=None):
    """Run n_chains large-scale chains and return the list of their result tuples (reference :22-104).

    mode 'replay' (default when largeScaleChain.rng_mode == 'replay'): every chain draws from its own NumPy
    generators exactly as the reference's pool workers do, so results and checkpoint files equal the CPU driver's; all
    chains of a rank share one handle, and `n_workers` host processes (the reference's argument; <= 0 or None: physical
    cores - 1) draw the proposals of the next chunk while the device steps the current one.
    mode 'pcg64': the same NumPy generator streams advanced on the device (gsm_draw_pcg64) with the device's spectral synthesis:
    the reference's draws, block records, accept decisions and generator-state files on the same seeds, no host draw per step
    (beds / losses to the accuracy of the device's inverse DFT against pocketfft).
    mode 'philox': all chains of this rank advance together in one handle with device-generated proposals.

    n_gpus: None = every visible GPU.  With more than one and no torch.distributed group in this process, the function
    starts its own ranks (one fresh process per GPU, RCCL) -- the caller brings no launcher, like the reference's driver;
    under torchrun (or inside such a rank) the initialised group is used and this process runs its shard."""
    tic = time.time()
    mode = mode or getattr(largeScaleChain, 'rng_mode', 'replay')
    import os
    import torch.distributed as dist
    sharded = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if not sharded and 'RANK' not in os.environ:
        import torch
        if n_gpus is None:
            n_gpus = max(1, torch.cuda.device_count())
        n_gpus = max(1, min(int(n_gpus), int(n_chains)))
        if n_gpus > 1:
            res = _self_launch(n_gpus, dict(n_chains=n_chains, n_workers=n_workers, largeScaleChain=largeScaleChain, rf=rf,
                                            initial_beds=initial_beds, rng_seeds=rng_seeds, n_iters=n_iters,
                                            output_path=output_path, mode=mode, batch=batch, gather=True))
            print(f'Completed in {time.time() - tic:.2f} seconds')
            return res
    rank, world = (dist.get_rank(), dist.get_world_size()) if sharded else (0, 1)
    lo, hi = parallel.shard_bounds(n_chains, world, rank)
    workers = n_workers
    if workers and workers > 0 and world > 1:
        workers = max(1, workers // world)
    result = _run_shard(lo, hi, largeScaleChain, rf, initial_beds, rng_seeds, n_iters, output_path, mode, batch, workers)
    if sharded and gather:
        result = _gather_results(result, n_chains, lo, hi)
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


# ---- small-scale chain driver (smallScaleChain_multiprocessing.py:211-399) ---------------------------------------------
_MSC_FILES = ('loss_mc', 'loss_data', 'loss', 'steps', 'resampled_times', 'blocks_used')


def _msc_load_previous(seed_folder):
    """Resume state of a small-scale seed folder (reference :299-341).  The bed_{k}k.txt file name carries floor(count / 1000)
    as in the reference; the exact iteration count -- what a Philox-mode resume continues its counters from -- is the length of
    the stored loss record (the reference restarts its label arithmetic from k * 1000, which is the same number whenever
    n_iter is a multiple of 1000, as in its drivers)."""
    beds = list(seed_folder.glob('bed_*.txt'))
    if not beds:
        return None
    k = int(beds[0].stem.split('_')[1].replace('k', ''))
    prev = {key: np.loadtxt(seed_folder / f'{key}_{k}k.txt') for key in _MSC_FILES}
    # label_count: what the reference continues its FILE LABELS from (k * 1000, :322-383); cumulative: the exact count
    return dict(cumulative=int(np.atleast_1d(prev['loss']).shape[0]), label_count=1000 * k, bed=np.loadtxt(beds[0]), results=prev,
                delete=[seed_folder / f'bed_{k}k.txt'] + [seed_folder / f'{key}_{k}k.txt' for key in _MSC_FILES])


def _msc_save(seed_folder, result, n_iter, prev):
    """Text checkpoint files of one finished small-scale segment (reference :363-397)."""
    seed_folder.mkdir(parents=True, exist_ok=True)
    beds, loss_mc, loss_data, loss, steps, resampled, blocks = result[:7]
    cumulative = 0
    label_count = 0
    if prev is not None:
        p = prev['results']
        loss_mc = np.concatenate([np.atleast_1d(p['loss_mc']), loss_mc])
        loss_data = np.concatenate([np.atleast_1d(p['loss_data']), loss_data])
        loss = np.concatenate([np.atleast_1d(p['loss']), loss])
        steps = np.concatenate([np.atleast_1d(p['steps']), steps])
        resampled = p['resampled_times'] + resampled
        blocks = np.vstack([p['blocks_used'], blocks])
        cumulative = prev['cumulative']
        label_count = prev.get('label_count', cumulative)
    cumulative += n_iter
    label = f'{(label_count + n_iter) // 1000}k'
    for key, arr in zip(('bed',) + _MSC_FILES, (beds, loss_mc, loss_data, loss, steps, resampled, blocks)):
        np.savetxt(seed_folder / f'{key}_{label}.txt', arr)
    if prev is not None:
        keep = {seed_folder / f'{key}_{label}.txt' for key in ('bed',) + _MSC_FILES}
        for f in prev['delete']:
            if f.exists() and f not in keep:
                f.unlink()


def msc_run_wrapper(param_chain, param_run):
    """Rebuild one small-scale chain from its parameter dict, resume from its seed folder if present, run one segment on
    the GPU, write the text files, return chain.run's tuple (reference :277-399)."""
    from . import sgs
    with contextlib.redirect_stdout(io.StringIO()):
        chain = sgs.init_msc_chain_by_instance(param_chain)
    output_path = param_run.get('output_path', './Data/LargeScaleChain/' + str(param_run['lsc_seed'])[:6] + '/SmallScaleChain')
    seed = param_run['ssc_seed']
    n_iter = param_run['n_iter']
    folder = Path(output_path) / f'{str(seed)[:6]}'
    prev = _msc_load_previous(folder)
    if prev is not None:
        chain.initial_bed = prev['bed']
    chain.chain_id = param_run.get('chain_id', 'Unknown')
    chain.tqdm_position = param_run.get('tqdm_position', 0)
    chain.seed = param_run.get('ssc_seed', 'Unkown')
    result = chain.run(n_iter=n_iter, only_save_last_bed=param_run['only_save_last_bed'], info_per_iter=param_run['info_per_iter'],
                       plot=param_run['plot'], progress_bar=None if not param_run.get('verbose', False) else param_run['progress_bar'])
    _msc_save(folder, result, n_iter, prev)
    return result


def smallScaleChain_mp(n_chains, n_workers, smallScaleChain, initial_beds, ssc_rng_seeds, lsc_rng_seed, n_iters,
                       output_path='./Data/output', mode=None, n_gpus=None):
    """Run n_chains small-scale chains and return the list of their result tuples (reference :211-274); files under
    <output_path>/LargeScaleChain/<lsc seed>/SmallScaleChain/<ssc seed>/ as the reference writes them.  Chains that share
    n_iter run together in one libgsm_hip handle (one workgroup per chain in every launch); `n_workers` is accepted for
    signature compatibility (the reference's pool size) and not used.

    mode 'replay' (default, or smallScaleChain.rng_mode): each chain draws from numpy.random.default_rng(its seed) as the
    reference's workers do -- results and files follow the CPU driver.  mode 'philox': the draws are made on the device (Philox
    counters keyed by the chain's seed, continuing at the iteration count of the seed folder's checkpoint): no host work per
    iteration.  mode 'pcg64': the draws of 'replay' (each chain's numpy.random.default_rng(its seed) stream, bit for bit) made on the
    device: the reference's chains and files without host work per iteration."""
    from . import sgs
    mode_eff = mode or getattr(smallScaleChain, 'rng_mode', 'replay')
    philox = mode_eff == 'philox'
    if mode not in (None, 'replay', 'philox', 'pcg64'):
        raise ValueError("mode must be 'replay', 'pcg64' or 'philox'")
    # n_gpus as in largeScaleChain_mp: the chains are independent, so they are sharded contiguously over the ranks (one per
    # GPU, started here when the caller brought no launcher) and the result tuples are gathered at the end
    import os
    import torch.distributed as dist
    sharded = n_gpus != -1 and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if n_gpus != -1 and not sharded and 'RANK' not in os.environ:
        import torch
        if n_gpus is None:
            n_gpus = max(1, torch.cuda.device_count())
        n_gpus = max(1, min(int(n_gpus), int(n_chains)))
        if n_gpus > 1:
            return _self_launch(n_gpus, dict(n_chains=n_chains, n_workers=n_workers, smallScaleChain=smallScaleChain, initial_beds=initial_beds,
                                             ssc_rng_seeds=ssc_rng_seeds, lsc_rng_seed=lsc_rng_seed, n_iters=n_iters,
                                             output_path=output_path, mode=mode), which='smallScaleChain_mp')
    if sharded:
        rank, world = dist.get_rank(), dist.get_world_size()
        lo, hi = parallel.shard_bounds(n_chains, world, rank)
        local = smallScaleChain_mp(hi - lo, n_workers, smallScaleChain, list(initial_beds[lo:hi]), list(ssc_rng_seeds[lo:hi]), lsc_rng_seed,
                                   list(n_iters[lo:hi]), output_path=output_path, mode=mode, n_gpus=-1) if hi > lo else []
        gathered = [None] * world
        dist.all_gather_object(gathered, local)
        return [r for part in gathered for r in part]
    tic = time.time()
    base = Path(output_path) / 'LargeScaleChain' / str(lsc_rng_seed)[:6] / 'SmallScaleChain'
    if len(set(int(v) for v in n_iters[:n_chains])) == 1 and n_chains > 0:
        n_iter = int(n_iters[0])
        folders = [base / f'{str(ssc_rng_seeds[i])[:6]}' for i in range(n_chains)]
        prevs = [_msc_load_previous(f) for f in folders]
        beds = [p['bed'] if p else initial_beds[i] for i, p in enumerate(prevs)]
        rngs = [np.random.default_rng(seed=ssc_rng_seeds[i]) for i in range(n_chains)]
        starts = set(p['cumulative'] if p else 0 for p in prevs)
        if philox and len(starts) != 1:
            raise ValueError('Philox mode runs the chains of a call in lock-step: their seed folders must hold the same iteration count')
        result, _ = sgs.run_many_sgs(smallScaleChain, beds, rngs, n_iter, only_save_last_bed=True, info_per_iter=10, progress_bar=None,
                                     philox_seeds=[int(v) for v in ssc_rng_seeds[:n_chains]] if philox else None,
                                     philox_iter0=starts.pop() if philox else 0, pcg64=(mode_eff == 'pcg64'))
        for i in range(n_chains):
            _msc_save(folders[i], result[i], n_iter, prevs[i])
    else:
        if philox:
            raise ValueError('Philox mode needs the same n_iter for every chain of a call')
        result = []
        for i in range(n_chains):
            cp = deepcopy(smallScaleChain.__dict__)
            cp['rng_seed'] = ssc_rng_seeds[i]
            cp['initial_bed'] = initial_beds[i]
            runp = dict(n_iter=n_iters[i], only_save_last_bed=True, info_per_iter=10, plot=False, progress_bar=False, chain_id=i,
                        tqdm_position=i + 2, ssc_seed=ssc_rng_seeds[i], lsc_seed=lsc_rng_seed, output_path=str(base))
            result.append(msc_run_wrapper(cp, runp))
    print(f'Completed in {time.time() - tic} seconds')
    return result
