"""Block sharding over the GPUs of one node.

The reference codes the blocks of a cloud serially in one process (src/model_types.py:192-212); blocks are
independent (no cross-block context), so here every rank (one process per GPU, torch.distributed over
RCCL/xGMI) codes a contiguous range of the Morton-ordered block list with replicated weights, and ONE gather
at the end brings the per-block (threshold index, strings) to rank 0, which assembles the same file a
single-GPU run writes.  The payload is tiny (tens of KB per cloud) so the collective is latency-bound:
one all_gather of the byte counts + one padded all_gather of the bytes.
"""
import pickle

import numpy as np
import torch


def shard_range(n_items, rank, world):
    """Contiguous, balanced range [lo, hi) of rank `rank` (the first n % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def _dist():
    import torch.distributed as dist
    return dist if dist.is_available() and dist.is_initialized() else None


def world_info():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d is not None else (0, 1)


def gather_objects(local_obj, device=None):
    """Gathers one picklable object per rank to EVERY rank (list ordered by rank) with two collectives.
    `device`: where the staging tensors live (cuda for the nccl/RCCL backend, cpu for gloo)."""
    d = _dist()
    if d is None:
        return [local_obj]
    world = d.get_world_size()
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device()) if d.get_backend() == 'nccl' else torch.device('cpu')
    payload = np.frombuffer(pickle.dumps(local_obj, protocol=4), np.uint8)
    n = torch.tensor([payload.size], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    d.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    buf = torch.zeros(max(sizes), dtype=torch.uint8, device=device)
    buf[:payload.size] = torch.from_numpy(payload.copy()).to(device)
    bufs = [torch.zeros(max(sizes), dtype=torch.uint8, device=device) for _ in range(world)]
    d.all_gather(bufs, buf)
    return [pickle.loads(bufs[r][:sizes[r]].cpu().numpy().tobytes()) for r in range(world)]
