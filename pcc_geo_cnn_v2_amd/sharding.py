"""Block sharding over the GPUs of one node.

The reference codes the blocks of a cloud serially in one process (src/model_types.py:192-212); blocks are
independent (no cross-block context), so here every rank (one process per GPU, torch.distributed over
RCCL/xGMI) codes a contiguous range of the Morton-ordered block list with replicated weights.  What crosses
ranks at the end is small and typed -- no pickled Python objects, so a C-ABI caller can reproduce it.  The shard sizes
are a pure function of (n_blocks, world) (`shard_range`), so no size exchange is needed:

  encoder (compress_blocks), TWO collectives per cloud (+1 per selected candidate with --dec_files / --debug; round 4: 4):
  1. ONE `all_gather` of fixed-width int64 rows: per block (string lengths, threshold indices, candidate point counts) --
     every rank derives every other rank's payload sizes from it -- and, riding as extra rows, the keys `d2 * world + rank` of
     ALL candidates x original points (`PiggybackGroup.claim`): every rank takes the MIN over ranks itself and so knows which
     rank holds the nearest decoded point of every original point (the D1/D2 numbers of `select_best_per_opt_metric`,
     src/model_types.py:128-176, are tallied by the owner: `cloud_metrics_batch(..., partial=True)`);
  2. ONE padded uint8 `all_gather` of (concatenated strings + the rank's partial tallies, candidates x 5 doubles): rank 0 assembles
     the same file a single-GPU run writes, every rank sums the tallies in rank order (the selection is replicated);
  3. only when the caller wants the reconstruction on rank 0 (`--dec_files`, `--debug`): one `gather` of the selected
     candidate's decoded float32 points.
  The keys are 8 B per input point and candidate and an all_gather moves them `world` times: above PCC_KEY_GATHER_MAX_BYTES (64 MB:
  a million points, one candidate, eight ranks) the MIN stays ONE `all_reduce` (`RankGroup.claim`), the tallies ride in the row
  all_gather and the strings go to rank 0 with one `gather`: three collectives.
  decoder (decompress_blocks), 2 collectives: one `all_gather` of the per-block point counts, one `gather` of the points.
Everything is latency-bound except (3)'s MIN over N_A int64 keys per candidate (8 MB per million input points).
`all_gather_rows` / `gather_rows` / `gather_bytes` without `counts` (ragged input of unknown size) prepend one small
`all_gather` of the sizes; the codec paths always pass `counts`.
"""
import numpy as np
import torch


# Optional accounting of the collectives (bench.py --workload configs2): STATS = {'calls', 'seconds', 'bytes'} while enabled.  `seconds` is
# host wall time from the call to the result on the host (the codec paths consume every collective on the host), `bytes` what this rank sent.
STATS = None


def stats_begin():
    global STATS
    STATS = {'calls': 0, 'seconds': 0.0, 'bytes': 0}


def stats_end():
    global STATS
    out, STATS = STATS, None
    return out


class _Timed:
    def __init__(self, nbytes):
        self.nbytes = int(nbytes)

    def __enter__(self):
        if STATS is not None:
            import time
            self.t0 = time.perf_counter()

    def __exit__(self, *exc):
        if STATS is not None:
            import time
            STATS['calls'] += 1
            STATS['seconds'] += time.perf_counter() - self.t0
            STATS['bytes'] += self.nbytes


def shard_range(n_items, rank, world):
    """Contiguous, balanced range [lo, hi) of rank `rank` (the first n % world ranks get one more)."""
    q, r = divmod(n_items, world)
    lo = rank * q + min(rank, r)
    return lo, lo + q + (1 if rank < r else 0)


def shard_sizes(n_items, world):
    return [hi - lo for lo, hi in (shard_range(n_items, r, world) for r in range(world))]


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


def _row_counts(d, n_local, counts, dev):
    """Rows held by every rank: given by the caller (no communication) or exchanged with one small all_gather."""
    world = d.get_world_size()
    if counts is not None:
        counts = [int(c) for c in counts]
        assert len(counts) == world and counts[d.get_rank()] == n_local, (counts, n_local)
        return counts
    cnt = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    d.all_gather(cnt, torch.tensor([n_local], dtype=torch.int64, device=dev))
    return [int(c.item()) for c in cnt]


def _padded(rows, cnt, dev):
    pad = torch.zeros((max(cnt + [1]),) + tuple(rows.shape[1:]), dtype=torch.from_numpy(rows).dtype, device=dev)
    pad[:rows.shape[0]] = torch.from_numpy(rows).to(dev)
    return pad


def all_gather_rows(rows, device=None, counts=None):
    """rows: (n_local, k) array, k and dtype equal on all ranks.  Returns the rank-ordered concatenation on EVERY rank: one
    padded `all_gather` (plus one for the row counts when `counts` -- rows per rank -- is not given)."""
    d = _dist()
    rows = np.ascontiguousarray(rows)
    if d is None:
        return rows
    world, dev = d.get_world_size(), _device(d, device)
    cnt = _row_counts(d, rows.shape[0], counts, dev)
    pad = _padded(rows, cnt, dev)
    with _Timed(pad.numel() * pad.element_size()):
        bufs = [torch.zeros_like(pad) for _ in range(world)]
        d.all_gather(bufs, pad)
        return np.concatenate([bufs[r][:cnt[r]].cpu().numpy() for r in range(world)], 0)


def gather_rows(rows, device=None, dst=0, counts=None):
    """Like all_gather_rows, but only rank `dst` receives the concatenation (others get None): one padded `gather`."""
    d = _dist()
    rows = np.ascontiguousarray(rows)
    if d is None:
        return rows
    world, rank, dev = d.get_world_size(), d.get_rank(), _device(d, device)
    cnt = _row_counts(d, rows.shape[0], counts, dev)
    pad = _padded(rows, cnt, dev)
    with _Timed(pad.numel() * pad.element_size()):
        bufs = [torch.zeros_like(pad) for _ in range(world)] if rank == dst else None
        d.gather(pad, bufs, dst=dst)
        if rank != dst:
            return None
        return np.concatenate([bufs[r][:cnt[r]].cpu().numpy() for r in range(world)], 0)


def gather_bytes(payload, device=None, dst=0, counts=None):
    """One byte string per rank -> list of byte strings (rank order) on rank `dst`, None elsewhere.  `counts`: the payload
    size of every rank when all ranks already know them (then this is a single `gather`)."""
    d = _dist()
    if d is None:
        return [bytes(payload)]
    world = d.get_world_size()
    arr = np.frombuffer(bytes(payload), np.uint8)
    sizes = _row_counts(d, arr.size, counts, _device(d, device))
    flat = gather_rows(arr, device, dst, counts=sizes)
    if flat is None:
        return None
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [flat[off[r]:off[r + 1]].tobytes() for r in range(world)]


def all_gather_bytes(payload, device=None, counts=None):
    """One byte string per rank -> list of byte strings (rank order) on EVERY rank: one padded uint8 `all_gather` (`counts`: the payload
    size of every rank, known to all)."""
    d = _dist()
    if d is None:
        return [bytes(payload)]
    arr = np.frombuffer(bytes(payload), np.uint8)
    sizes = _row_counts(d, arr.size, counts, _device(d, device))
    flat = all_gather_rows(arr, device, counts=sizes)
    off = np.concatenate([[0], np.cumsum(sizes)])
    return [flat[off[r]:off[r + 1]].tobytes() for r in range(d.get_world_size())]


def all_reduce(arr, op, device=None):
    """all_reduce of a numpy array ('min' | 'sum'); returns the reduced array on every rank."""
    d = _dist()
    arr = np.ascontiguousarray(arr)
    if d is None:
        return arr
    with _Timed(arr.nbytes):
        t = torch.from_numpy(arr.copy()).to(_device(d, device))
        d.all_reduce(t, op=d.ReduceOp.MIN if op == 'min' else d.ReduceOp.SUM)
        return t.cpu().numpy()


class RankGroup:
    """The communicator `utils.pc_metric.cloud_metrics_batch` uses under torch.distributed (the single-process one is
    pc_metric.SingleProcess): the original cloud is replicated, every rank holds the decoded points of its own blocks.

    claim(): an original point belongs to the rank that holds its nearest decoded point.  Squared distances between integer
    points are integers, so `d2 * world + rank` is an exact int64 key and one all_reduce(MIN) yields both the global
    minimum and its owner -- the lowest rank among equidistant candidates (a single process takes the KD-tree's pick, so D2
    can differ from the single-process value through such cross-shard ties; D1 cannot)."""

    def __init__(self, device=None):
        self.rank, self.world = world_info()
        self.device = device

    def claim(self, sq_dist_ab, have_points):
        n_cand = len(sq_dist_ab)
        big = np.iinfo(np.int64).max
        keys = np.stack([np.rint(d).astype(np.int64) * self.world + self.rank if h else np.full(len(d), big, np.int64)
                         for d, h in zip(sq_dist_ab, have_points)]) if n_cand else np.zeros((0, 0), np.int64)
        keys = all_reduce(keys, 'min', self.device)
        return [None if (keys.shape[1] and keys[m, 0] == big) or not keys.shape[1] else keys[m] % self.world == self.rank
                for m in range(n_cand)]

    def total(self, tallies):
        return all_reduce(tallies, 'sum', self.device)


class PiggybackGroup(RankGroup):
    """RankGroup whose claim() does not spend a collective of its own: the int64 keys `d2 * world + rank` of all candidates ride as extra
    rows in an all_gather the caller needs anyway (the per-block row table of the sharded encoder), and every rank takes the MIN over
    ranks itself -- the "single gather" of SURVEY.md 8e.  Moves the keys `world` times (an all_reduce moves them once): the caller uses it
    when 8 B x candidates x original points x world stays below a bound, else RankGroup.  After claim(): `.table` = the caller's rows
    of all ranks in rank order."""

    def __init__(self, rows, per_rank_rows, device=None):
        super().__init__(device)
        self.rows, self.per_rank_rows, self.table = np.ascontiguousarray(rows, np.int64), [int(n) for n in per_rank_rows], None

    def claim(self, sq_dist_ab, have_points):
        n_cand = len(sq_dist_ab)
        big = np.iinfo(np.int64).max
        keys = np.stack([np.rint(d).astype(np.int64) * self.world + self.rank if h else np.full(len(d), big, np.int64)
                         for d, h in zip(sq_dist_ab, have_points)]) if n_cand else np.zeros((0, 0), np.int64)
        width = self.rows.shape[1]
        K = -(-keys.size // width)
        send = np.zeros((self.rows.shape[0] + K, width), np.int64)
        send[:self.rows.shape[0]] = self.rows
        send[self.rows.shape[0]:].reshape(-1)[:keys.size] = keys.reshape(-1)
        got = all_gather_rows(send, self.device, counts=[n + K for n in self.per_rank_rows])
        ends = np.cumsum([n + K for n in self.per_rank_rows])
        self.table = np.concatenate([got[e - n - K:e - K] for e, n in zip(ends, self.per_rank_rows)], 0)
        best = None
        for e in ends:
            k = got[e - K:e].reshape(-1)[:keys.size].reshape(keys.shape)
            best = k if best is None else np.minimum(best, k)
        if best is None:
            best = keys
        return [None if (best.shape[1] and best[m, 0] == big) or not best.shape[1] else best[m] % self.world == self.rank
                for m in range(n_cand)]


def sharded_metrics(p1, p2_local, r, p1_n=None, t1=None, device=None):
    """compute_metrics(p1, union over ranks of p2_local, r, p1_n) on every rank (None when the union is empty)."""
    from .utils.pc_metric import cloud_metrics_batch
    return cloud_metrics_batch(p1, [p2_local], r, p1_n, t1, RankGroup(device))[0]
