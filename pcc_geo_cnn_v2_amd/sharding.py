"""Block sharding over the GPUs of one node.

The reference codes the blocks of a cloud serially in one process (src/model_types.py:192-212); blocks are
independent (no cross-block context), so here every rank (one process per GPU, torch.distributed over
RCCL/xGMI) codes a contiguous range of the Morton-ordered block list with replicated weights.  What crosses
ranks at the end is small and typed -- no pickled Python objects, so a C-ABI caller can reproduce it:

  1. `all_gather_rows`: one int64 row per block (len_y, len_z, threshold indices, candidate point counts) to every rank;
  2. `gather_bytes`: one padded uint8 `gather` of the concatenated strings to rank 0 (tens of KB per cloud), which
     assembles the same file a single-GPU run writes;
  3. the D1/D2 numbers of `select_best_per_opt_metric` (src/model_types.py:128-176) from per-rank partial sums
     (`sharded_metrics`): one `all_reduce(MIN)` over the original points + one `all_reduce(SUM)` of four scalars;
  4. only when the caller wants the reconstruction on rank 0 (`--dec_files`, `--debug`): `gather_rows` of the decoded
     float32 points.
Everything is latency-bound except (3)'s MIN over N_A int64 keys (8 MB per million input points).
"""
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


def _device(d, device=None):
    if device is not None:
        return device
    return torch.device('cuda', torch.cuda.current_device()) if d.get_backend() == 'nccl' else torch.device('cpu')


def all_gather_rows(rows, device=None):
    """rows: (n_local, k) int64/float array, k equal on all ranks.  Returns the rank-ordered concatenation on EVERY rank
    (two collectives: row counts, then padded rows)."""
    d = _dist()
    rows = np.ascontiguousarray(rows)
    if d is None:
        return rows
    world, dev = d.get_world_size(), _device(d, device)
    t = torch.from_numpy(rows)
    cnt = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    d.all_gather(cnt, torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev))
    cnt = [int(c.item()) for c in cnt]
    pad = torch.zeros((max(cnt + [1]),) + tuple(rows.shape[1:]), dtype=t.dtype, device=dev)
    pad[:rows.shape[0]] = t.to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)]
    d.all_gather(bufs, pad)
    return np.concatenate([bufs[r][:cnt[r]].cpu().numpy() for r in range(world)], 0)


def gather_rows(rows, device=None, dst=0):
    """Like all_gather_rows, but only rank `dst` receives the concatenation (others get None): one small all_gather of the
    row counts + one padded `gather`."""
    d = _dist()
    rows = np.ascontiguousarray(rows)
    if d is None:
        return rows
    world, rank, dev = d.get_world_size(), d.get_rank(), _device(d, device)
    t = torch.from_numpy(rows)
    cnt = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    d.all_gather(cnt, torch.tensor([rows.shape[0]], dtype=torch.int64, device=dev))
    cnt = [int(c.item()) for c in cnt]
    pad = torch.zeros((max(cnt + [1]),) + tuple(rows.shape[1:]), dtype=t.dtype, device=dev)
    pad[:rows.shape[0]] = t.to(dev)
    bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
    d.gather(pad, bufs, dst=dst)
    if rank != dst:
        return None
    return np.concatenate([bufs[r][:cnt[r]].cpu().numpy() for r in range(world)], 0)


def gather_bytes(payload, device=None, dst=0):
    """One byte string per rank -> list of byte strings (rank order) on rank `dst`, None elsewhere."""
    d = _dist()
    if d is None:
        return [bytes(payload)]
    world = d.get_world_size()
    arr = np.frombuffer(bytes(payload), np.uint8)
    sizes = all_gather_rows(np.array([[arr.size]], np.int64), device)[:, 0]
    flat = gather_rows(arr, device, dst)
    if flat is None:
        return None
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [flat[off[r]:off[r + 1]].tobytes() for r in range(world)]


def all_reduce(arr, op, device=None):
    """In-place-style all_reduce of a numpy array ('min' | 'sum'); returns the reduced array on every rank."""
    d = _dist()
    arr = np.ascontiguousarray(arr)
    if d is None:
        return arr
    t = torch.from_numpy(arr.copy()).to(_device(d, device))
    d.all_reduce(t, op=d.ReduceOp.MIN if op == 'min' else d.ReduceOp.SUM)
    return t.cpu().numpy()


def sharded_metrics(p1, p2_local, r, p1_n=None, t1=None, device=None):
    """utils.pc_metric.compute_metrics(p1, p2, r, p1_n) (src/utils/pc_metric.py:76-138) where the decoded cloud p2 is
    the union of every rank's `p2_local` and the original cloud p1 (+ normals) is replicated.  Exact for D1: squared
    distances between integer points are integers, A->B is a MIN over ranks, B->A a SUM.  For D2 the nearest decoded
    point of an original point is taken from the lowest rank among equidistant candidates (a single process takes the
    KD-tree's pick): sums can differ from the single-process value only through such cross-shard ties.
    Returns None when the decoded cloud is empty on every rank (the caller substitutes -inf like model_types.py:150)."""
    from scipy.spatial import cKDTree
    from .utils.pc_metric import psnr, sum_d2
    rank, world = world_info()
    p1 = np.asarray(p1, np.float64)
    p2 = np.asarray(p2_local, np.float64).reshape(-1, 3)
    if t1 is None:
        t1 = cKDTree(p1, balanced_tree=False)
    BIG = np.iinfo(np.int64).max
    if len(p2):
        t2 = cKDTree(p2, balanced_tree=False)
        _, idx2 = t2.query(p1, workers=-1 if len(p1) > 200000 else 1)
        d2 = np.rint(np.sum((p1 - p2[idx2]) ** 2, axis=1)).astype(np.int64)
        key = d2 * world + rank
        _, idx1 = t1.query(p2, workers=-1 if len(p2) > 200000 else 1)
        sum_ba = float(np.sum((p2 - p1[idx1]) ** 2))
    else:
        idx2 = np.zeros(len(p1), np.int64)
        key = np.full(len(p1), BIG, np.int64)
        idx1 = np.zeros(0, np.int64)
        sum_ba = 0.0
    key = all_reduce(key, 'min', device)
    if key.size and key[0] == BIG:
        return None
    owner, d2min = key % world, key // world
    mine = owner == rank
    part = np.zeros(4, np.float64)                      # d1_sum_BA, n_B, d2_sum_AB, d2_sum_BA
    part[0], part[1] = sum_ba, len(p2)
    if p1_n is not None and len(p2):
        # assign_attr (pc_metric.py:8-25) restricted to this rank's decoded points
        counts = np.zeros(len(p2))
        attr = np.zeros((len(p2), p1_n.shape[1]))
        np.add.at(counts, idx2[mine], 1)
        np.add.at(attr, idx2[mine], p1_n[mine])
        empty = counts == 0
        attr[empty] += p1_n[idx1[empty]]
        counts[empty] += 1
        p2_n = attr / counts[:, None]
        part[2] = sum_d2(p1[mine], p2[idx2[mine]], p2_n[idx2[mine]])
        part[3] = sum_d2(p2, p1[idx1], p1_n[idx1])
    part = all_reduce(part, 'sum', device)
    n1, n2 = p1.shape[0], part[1]
    max_energy = 3 * r * r
    s_ab, s_ba = np.float64(np.sum(d2min)), np.float64(part[0])      # numpy scalars: x / 0 -> inf like in compute_metrics
    m_ab, m_ba = s_ab / n1, s_ba / n2
    metrics = {
        'd1_sum_AB': s_ab, 'd1_sum_BA': s_ba, 'd1_sum_max': max(s_ab, s_ba), 'd1_sum_mean': (s_ab + s_ba) / 2,
        'd1_mse_AB': m_ab, 'd1_mse_BA': m_ba, 'd1_mse': max(m_ab, m_ba), 'd1_psnr_AB': psnr(m_ab, max_energy),
        'd1_psnr_BA': psnr(m_ba, max_energy), 'd1_psnr': min(psnr(m_ab, max_energy), psnr(m_ba, max_energy))}
    if p1_n is not None:
        s_ab, s_ba = np.float64(part[2]), np.float64(part[3])
        m_ab, m_ba = s_ab / n1, s_ba / n2
        metrics.update({
            'd2_sum_AB': s_ab, 'd2_sum_BA': s_ba, 'd2_sum_max': max(s_ab, s_ba), 'd2_sum_mean': (s_ab + s_ba) / 2,
            'd2_mse_AB': m_ab, 'd2_mse_BA': m_ba, 'd2_mse': max(m_ab, m_ba), 'd2_psnr_AB': psnr(m_ab, max_energy),
            'd2_psnr_BA': psnr(m_ba, max_energy), 'd2_psnr': min(psnr(m_ab, max_energy), psnr(m_ba, max_energy))})
    return metrics
