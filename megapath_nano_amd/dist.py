"""One process per GPU: torch.distributed plumbing (backend "nccl" is RCCL on ROCm; "gloo" for the CPU tests).

Reads shard across ranks with no data-path collective; the only exchange is the sum of the small per-name /
per-species int64 counters around the reassignment pass."""
import os

import numpy as np


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK/WORLD_SIZE/MASTER_* if WORLD_SIZE > 1.  Returns (rank, world, local_rank)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        import torch
        if backend is None:
            backend = 'nccl' if torch.cuda.is_available() else 'gloo'
        if backend == 'nccl' and os.environ.get('MPN_SINGLE_DEVICE') != '1':
            torch.cuda.set_device(local)
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if not dist.is_initialized():
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def make_allreduce(device=None):
    """-> callable(np.ndarray[int64]) summing in place over all ranks (identity when not distributed)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return None
    import torch

    def allreduce(arr):
        t = torch.from_numpy(arr)
        if device is not None:
            t = t.to(device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        if device is not None:
            arr[:] = t.cpu().numpy()
    return allreduce


def shard_bounds(lengths, world):
    """Contiguous read ranges with balanced total bases (equal sum of bp, not equal count).
    -> list of (lo, hi) per rank."""
    lengths = np.asarray(lengths, dtype=np.int64)
    n = len(lengths)
    if n == 0:
        return [(0, 0)] * world
    csum = np.cumsum(lengths)
    total = int(csum[-1])
    cuts = [0]
    for r in range(1, world):
        cuts.append(int(np.searchsorted(csum, total * r / world, side='left')) + 1 if total else 0)
    cuts.append(n)
    cuts = [min(max(c, 0), n) for c in cuts]
    for i in range(1, len(cuts)):
        cuts[i] = max(cuts[i], cuts[i - 1])
    return [(cuts[r], cuts[r + 1]) for r in range(world)]


def barrier():
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        dist.barrier()
