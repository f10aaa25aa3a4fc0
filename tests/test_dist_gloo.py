"""CPU suite: the N>1 host path with torch.distributed (gloo, world_size 2): contiguous chain shards, the
segment-end all-gather of per-chain results (equal and ragged shards), the posterior-mean all-reduce and the
max-over-ranks timing reduction used by bench.py.  On the GPU node the same code runs over RCCL (backend "nccl")."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parent.parent


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n_chains, q):
    sys.path.insert(0, str(ROOT))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from mcmc_gpu_amd import parallel, driver
    r, lr, w = parallel.init_distributed("gloo")
    assert (r, w) == (rank, world)
    lo, hi = parallel.shard_bounds(n_chains, world, rank)
    H = W = 6
    # per-chain payloads whose content encodes the global chain index
    beds = torch.stack([torch.full((H, W), float(c)) for c in range(lo, hi)]) if hi > lo else torch.zeros((0, H, W))
    loss = torch.stack([torch.arange(5, dtype=torch.float64) + 100 * c for c in range(lo, hi)]) if hi > lo else torch.zeros((0, 5), dtype=torch.float64)
    g_beds = parallel.all_gather_chains(beds.double(), n_chains)
    g_loss = parallel.all_gather_chains(loss, n_chains)
    mean = parallel.all_reduce_mean_field(beds.double().sum(dim=0), n_chains)
    tmax = parallel.max_over_ranks(1.0 + rank, torch.device("cpu"))
    # driver-level gather of result tuples
    local = [(np.full((H, W), float(c)), np.arange(5.) + c, np.zeros(5), np.arange(5.) + c, np.ones(5) * (c % 2),
              np.full((H, W), 2.0 * c), np.full((5, 4), float(c))) for c in range(lo, hi)]
    res = driver._gather_results(local, n_chains, lo, hi)
    parallel.barrier()
    q.put((rank, g_beds.numpy(), g_loss.numpy(), mean.numpy(), tmax, [float(t[0][0, 0]) for t in res],
           [float(t[5][0, 0]) for t in res]))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_chains", [8, 7])
def test_gather_and_reduce_world2(n_chains):
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_chains, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    exp_beds = np.stack([np.full((6, 6), float(c)) for c in range(n_chains)])
    exp_loss = np.stack([np.arange(5.) + 100 * c for c in range(n_chains)])
    for rank, g_beds, g_loss, mean, tmax, first, res5 in outs:
        assert np.array_equal(g_beds, exp_beds)
        assert np.array_equal(g_loss, exp_loss)
        assert np.allclose(mean, exp_beds.mean(axis=0))
        assert tmax == 2.0
        assert first == [float(c) for c in range(n_chains)]
        assert res5 == [2.0 * c for c in range(n_chains)]
