"""Chain sharding across the GPUs of one node: one process per GPU, torch.distributed over RCCL/xGMI.

The reference's only parallelism is one chain per OS process (mp.Pool.starmap,
largeScaleChain_multiprocessing_GPU.py:84-85); chains never communicate.  Here every rank owns one contiguous shard of the chains -- floor(n_chains / world)
of them, the first n_chains % world ranks one more (shard_bounds) -- the step loop runs with no collective at all, and the
per-chain results are all-gathered once per segment, after the loop.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def dist_env():
    """(rank, local_rank, world_size) from the torchrun environment (1 process => (0, 0, 1))."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def init_distributed(backend: str | None = None):
    """Initialise torch.distributed when launched with WORLD_SIZE > 1.  backend "nccl" is RCCL on ROCm."""
    rank, local_rank, world = dist_env()
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = os.environ.get("GSM_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, local_rank, world


def shard_bounds(n_chains: int, world: int, rank: int):
    """[lo, hi) of the contiguous shard of `rank`; the first n_chains % world ranks hold one more chain."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    q, r = divmod(n_chains, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _via_host(t: torch.Tensor) -> bool:
    """gloo (CPU rehearsal of the multi-rank path) moves device tensors through the host; RCCL does not."""
    return t.is_cuda and dist.get_backend() == "gloo"


def all_gather_chains(local: torch.Tensor, n_chains: int) -> torch.Tensor:
    """All-gather per-chain rows (dim 0 = local chains, ragged across ranks allowed) into the full
    (n_chains, ...) tensor on every rank.  One collective; equal shards use all_gather_into_tensor."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    if _via_host(local):
        return all_gather_chains(local.cpu(), n_chains).to(local.device)
    world, rank = dist.get_world_size(), dist.get_rank()
    sizes = [shard_bounds(n_chains, world, r)[1] - shard_bounds(n_chains, world, r)[0] for r in range(world)]
    if local.shape[0] != sizes[rank]:
        raise ValueError(f"rank {rank} holds {local.shape[0]} chains, expected {sizes[rank]}")
    local = local.contiguous()
    if len(set(sizes)) == 1:
        out = torch.empty((n_chains,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local)
        return out
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[: local.shape[0]] = local
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad)
    return torch.cat([p[:s] for p, s in zip(parts, sizes)], dim=0)


def all_reduce_mean_field(local_sum: torch.Tensor, n_chains: int) -> torch.Tensor:
    """Posterior-mean field from per-rank sums over local chains: one all-reduce of an (H, W) tensor."""
    t = local_sum.clone()
    if dist.is_initialized() and dist.get_world_size() > 1:
        if _via_host(t):
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM)
            t = h.to(t.device)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t / float(n_chains)


def barrier():
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()


def max_over_ranks(value: float, device) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    t = torch.tensor([value], dtype=torch.float64, device="cpu" if dist.get_backend() == "gloo" else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
