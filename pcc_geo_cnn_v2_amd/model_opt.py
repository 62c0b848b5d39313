"""Per-block threshold search -- same decisions as /root/reference/src/model_opt.py:9-77 (pinned by the reference-generated
fixtures tests/golden/model_opt.npz and model_opt_d2.npz and by the known answers of src/test_model_opt.py).

Formulation used here (host AND GPU path): the decoded set at threshold index t is the level set B_t = {v : x_hat(v) >
thresholds[t]}.  A block is reduced to a TABLE of per-threshold tallies (utils.pc_metric: |B_t| and the D1 / D2 sums in both
directions) for the leading run of non-empty level sets, plus the tally of the single rounded mean point (the reference's
failure guard, model_opt.py:59-68).  `select_thresholds_from_stats` then makes every decision (argmin per metric, max_delta
eligibility, guard) on whole columns of `pc_metric.metrics_table`.  Where the tallies come from is interchangeable:

  * `host_threshold_stats`: KD-trees, used only to obtain nearest-neighbour indices.  The tree over the original points is the
    same for every threshold, so the B->A neighbours of all voxels are queried ONCE (on B_0, the largest level set) and sliced
    per threshold; one tree per level set remains for A->B (which neighbour scipy's tree returns among equidistant ones decides
    the D2 numbers, so that tree must be the one the reference builds: same points, same order, balanced_tree=False);
  * `ops.d1_threshold_stats` (csrc/threshold_search.hip): exact squared Euclidean distance transforms on the GPU -- D1 only;
  * both: d1_* columns from the GPU, d2_* columns from the host pool (`HostSearchPool`, groups=('d2',)).

The comparisons `x_hat > t` are done in float32 like the reference under its pinned numpy 1.18.
"""
import logging

import os

import numpy as np
from scipy.spatial import cKDTree

from .utils import pc_metric as PM
from .utils.pc_metric import validate_opt_metrics

logger = logging.getLogger(__name__)


def _gt(x_hat, t):
    x_hat = np.asarray(x_hat)
    if x_hat.dtype == np.float32:
        return x_hat > np.float32(t)
    return x_hat > t


def _level_subsets(x_hat, thresholds):
    """(cand, [sel_0, sel_1, ...]): `cand` = float32 coordinates of the voxels above the smallest threshold in np.argwhere
    (lexicographic) order -- the grid is scanned once -- and sel_t = indices into `cand` of the level set {x_hat >
    thresholds[t]}, for t = 0, 1, ... up to (excluding) the first empty one.  Every level set is an order-preserving subset
    of `cand`, whatever the order of the thresholds."""
    x_hat = np.asarray(x_hat)
    thresholds = np.asarray(thresholds)
    if len(thresholds) == 0:
        return np.zeros((0, x_hat.ndim), 'float32'), []
    cand = np.argwhere(_gt(x_hat, thresholds.min()))
    vals = x_hat[tuple(cand.T)]
    subsets = []
    for thr in thresholds:
        sel = np.flatnonzero(_gt(vals, thr))
        if len(sel) == 0:
            break
        subsets.append(sel)
    return cand.astype('float32'), subsets


def level_sets(x_hat, thresholds):
    """[(t, points float32 (n, ndim))] for t = 0, 1, ... while the level set {x_hat > thresholds[t]} is non-empty."""
    cand, subsets = _level_subsets(x_hat, thresholds)
    return [(t, cand[sel]) for t, sel in enumerate(subsets)]


def ratio_eligible(n_b, n_a, max_delta):
    """model_opt.py:16-17: a decoded set is eligible when 1/max_delta < |B| / |A| < max_delta."""
    ratio = np.asarray(n_b) / n_a
    return ((1 / max_delta) < ratio) & (ratio < max_delta)


def build_points_threshold(x_hat, thresholds, len_block, max_delta=np.inf):
    """[(t, points)] of the eligible level sets (the reference's helper, model_opt.py:9-18; known answers in
    src/test_model_opt.py:12-36)."""
    return [(t, pts) for t, pts in level_sets(x_hat, thresholds) if ratio_eligible(len(pts), len_block, max_delta)]


def metric_names(opt_metrics, max_deltas):
    return [f'{m}_{d}' for d in max_deltas for m in opt_metrics]


def host_threshold_stats(block, x_hat, thresholds, normals=None):
    """Tallies of every non-empty leading level set of one block against the original points, with KD-trees.
    Returns (tallies float64[T, 5], mean_tally float64[5]); T = number of leading non-empty level sets."""
    a = block[:, :3]
    tree_a = cKDTree(a, balanced_tree=False)
    cand, subsets = _level_subsets(x_hat, thresholds)
    tallies = np.zeros((len(subsets), 5), np.float64)
    cand_to_a = PM.nearest(tree_a, cand) if subsets else None       # B->A: one query serves every threshold
    for t, sel in enumerate(subsets):
        b = cand[sel]
        to_b = PM.nearest(cKDTree(b, balanced_tree=False), a)       # A->B: the reference's tree over B_t
        tallies[t] = PM.pair_tally(a, b, to_b, cand_to_a[sel], normals)
    mean_point = np.round(np.mean(a, axis=0))[np.newaxis, :]
    mean_tally = PM.pair_tally(a, mean_point, np.zeros(len(a), np.int64), PM.nearest(tree_a, mean_point), normals)
    return tallies, mean_tally


def host_threshold_stats_pruned(block, x_hat, thresholds, normals, d1_gpu, resolution, opt_metrics, max_deltas):
    """host_threshold_stats for the d2_* metrics with KD-tree work ONLY where it can still change a decision (round 6, VERDICT r05 item 2: the
    reference's decisions -- /root/reference/src/model_opt.py:33-73 with the neighbour picks of src/utils/pc_metric.py:76-131 -- at a fraction
    of the queries).  Exact and free for EVERY threshold: |B_t| and the D1 sums (`d1_gpu`: the GPU's integer distance transforms,
    float64[T, 5] with the N_B / D1_AB / D1_BA slots filled).  Bounds of the D2 sums that hold for ANY pick among equidistant neighbours
    and any transferred normal (a mean of original normals; Cauchy-Schwarz term by term), c = max |n|^2:

        A->B (one tree PER threshold):      0 <= D2_AB(t) <= c D1_AB(t)
        B->A (one tree for all, but one query per decoded voxel -- the far, low-level voxels are the slow ones): with the voxels of the
        level set Q = B_q queried and p(v) their exact plane errors,
            t >= q:  D2_BA(t) = sum_{v in B_t} p(v)                                       (exact: B_t is a subset of Q)
            t <  q:  sum_Q p <= D2_BA(t) <= sum_Q p + c (D1_BA(t) - D1_BA(q))             (the voxels outside Q: 0 <= term <= c |gap|^2)

    Every d2 metric is monotone in both sums, so the reference's metric table evaluated at both ends brackets it: [L_m(t), U_m(t)].
    Branch and bound per (max_delta pool, metric m): best = min over the pool of U_m (exact values as they arrive: L = U); a threshold with
    L_m(t) > best cannot be the (first) argmin; of the others the one with the smallest estimate L + rho (U - L) is evaluated next --
    exactly as the unpruned search does it (Q is first extended down to t if need be; the reference's tree over B_t, queried with the
    original points) -- until none is left.  Every threshold whose true value equals the minimum has L <= best and is evaluated, so
    `first minimum` picks the same index; the rest keeps its upper end (strictly above the minimum: never selected).  An original point
    that is itself in B_t is its own nearest decoded point (distance 0, B_t has no duplicates): only the others are asked of the tree.
    Thresholds must be non-decreasing for the level arithmetic (the reference's np.linspace is); otherwise everything is evaluated.
    Returns (tallies float64[T, 5], mean_tally float64[5], number of thresholds evaluated exactly)."""
    thr = np.asarray(thresholds)
    if len(thr) and np.any(np.diff(thr) < 0):
        tallies, mean_tally = host_threshold_stats(block, x_hat, thresholds, normals)
        return tallies, mean_tally, len(tallies)
    a = block[:, :3]
    n_a = len(a)
    tree_a = cKDTree(a, balanced_tree=False)
    mean_point = np.round(np.mean(a, axis=0))[np.newaxis, :]
    mean_tally = PM.pair_tally(a, mean_point, np.zeros(n_a, np.int64), PM.nearest(tree_a, mean_point), normals)
    d1_gpu = np.asarray(d1_gpu, np.float64).reshape(-1, 5)
    T = len(d1_gpu)
    x_hat = np.asarray(x_hat)
    cand_vox = np.argwhere(_gt(x_hat, thr.min())) if len(thr) else np.zeros((0, x_hat.ndim), np.int64)
    vals = x_hat[tuple(cand_vox.T)]
    # level(v) = number of thresholds below x_hat(v), compared like _gt: B_t = {level > t}, in np.argwhere order like the reference's sets
    level = np.searchsorted(thr.astype(np.float32) if x_hat.dtype == np.float32 else thr, vals, side='left')
    assert (int(level.max()) if len(level) else 0) == T, 'host and GPU disagree on the number of non-empty level sets'
    if T == 0:
        return np.zeros((0, 5), np.float64), mean_tally, 0
    n_b = np.cumsum(np.bincount(level, minlength=T + 1)[::-1])[::-1][1:T + 1]       # |B_t| = #{level > t}
    assert np.array_equal(n_b, d1_gpu[:, PM.N_B]), 'host and GPU level sets differ'
    cand = cand_vox.astype('float32')
    c = float((np.asarray(normals, np.float64) ** 2).sum(axis=1).max()) if n_a else 1.0
    # slack of every comparison between a bracket end and an exact value: the exact rows are summed in the dtype numpy promotes block and
    # voxel coordinates to, like the reference (float32 PLY points: pairwise float32 sums), the bracket ends in float64
    wide = np.result_type(block.dtype, np.float32) == np.float64
    tol = 1e-9 if wide else 1e-5

    # B -> A, lazily: cand_to_a / plane error p(v) of the voxels of Q = B_q
    cand_to_a = np.full(len(cand), -1, np.int64)
    p_ba = np.zeros(len(cand), np.float64)
    q = [T]                                                                         # nothing queried yet

    def extend(t):
        if t >= q[0]:
            return
        new = np.flatnonzero((level > t) & (level <= q[0]))
        to_a = PM.nearest(tree_a, cand[new])
        cand_to_a[new] = to_a
        p_ba[new] = ((cand[new] - a[to_a]) * normals[to_a]).sum(axis=1) ** 2
        q[0] = t

    lo, hi, exact = d1_gpu.copy(), d1_gpu.copy(), np.zeros(T, bool)

    def refresh_ba():
        known = np.cumsum(np.bincount(level, weights=p_ba, minlength=T + 1)[::-1])[::-1][1:T + 1]   # sum of p over (B_t and Q)
        free = ~exact
        lo[free, PM.D2_BA] = known[free]
        slack = np.where(np.arange(T) < q[0], c * (d1_gpu[:, PM.D1_BA] - d1_gpu[min(q[0], T - 1), PM.D1_BA]) * (1 + tol), 0.0)
        hi[free, PM.D2_BA] = (known + np.maximum(slack, 0.0))[free]

    # start with the level sets up to ~4 |A| voxels: near the original surface, the cheap queries
    small = np.flatnonzero(n_b <= 4 * n_a + 1024)
    extend(int(small[0]) if len(small) else T - 1)
    lo[:, PM.D2_AB] = 0.0
    hi[:, PM.D2_AB] = c * d1_gpu[:, PM.D1_AB] * (1 + tol) + 1e-300
    refresh_ba()

    # where (if anywhere) each original point sits in `cand`: its own voxel, when x_hat is above the smallest threshold there
    a_vox = np.asarray(a).astype(np.int64)
    own = np.full(n_a, -1, np.int64)
    if len(cand) and np.array_equal(a_vox, a):
        inside = ((a_vox >= 0) & (a_vox < np.asarray(x_hat.shape))).all(axis=1)
        slot_of = np.full(x_hat.size, -1, np.int64)
        slot_of[np.ravel_multi_index(tuple(cand_vox.T), x_hat.shape)] = np.arange(len(cand))
        own[inside] = slot_of[np.ravel_multi_index(tuple(a_vox[inside].T), x_hat.shape)]

    def evaluate(t):
        if t < q[0]:
            extend(t)
            refresh_ba()
        member = np.zeros(len(cand) + 1, bool)                                      # (+1: slot -1 of the points without a voxel in cand)
        member[:-1] = level > t
        sel = np.flatnonzero(member)
        b = cand[sel]
        at_home = member[own]
        to_b = np.empty(n_a, np.int64)
        if at_home.any():
            to_b[at_home] = (np.cumsum(member[:-1]) - 1)[own[at_home]]              # position of the point's own voxel inside B_t
        away = ~at_home
        if away.any():
            to_b[away] = PM.nearest(cKDTree(b, balanced_tree=False), a[away])       # A->B: the reference's tree over B_t
        row = PM.pair_tally(a, b, to_b, cand_to_a[sel], normals)
        assert np.array_equal(row[:3], d1_gpu[t, :3]) if wide else np.allclose(row[:3], d1_gpu[t, :3], rtol=tol, atol=0), \
            'host KD-tree and GPU distance-transform D1 sums differ'
        lo[t], hi[t], exact[t] = row, row, True

    d2_metrics = [m for m in opt_metrics if m.startswith('d2_')]
    everything = np.arange(T)
    rho, seen = 0.25, 0
    for max_delta in max_deltas:
        pool = everything
        if max_delta is not None:
            ok = everything[ratio_eligible(n_b, n_a, max_delta)]
            pool = ok if len(ok) else everything
        for m in d2_metrics:
            while True:
                ends = PM.metrics_table(n_a, lo[pool], resolution - 1, ['d2'])[m], PM.metrics_table(n_a, hi[pool], resolution - 1, ['d2'])[m]
                L, U = np.minimum(*ends), np.maximum(*ends)
                todo = ~exact[pool]
                if np.isfinite(L).all() and np.isfinite(U).all():
                    todo &= L * (1 - tol) <= U.min() * (1 + tol)
                open_ = np.flatnonzero(todo)                                        # (a degenerate table: no pruning)
                if len(open_) == 0:
                    break
                k = open_[np.argmin((L + rho * (U - L))[open_])]
                evaluate(int(pool[k]))
                value = PM.metrics_table(n_a, lo[pool[k]], resolution - 1, ['d2'])[m]
                if np.isfinite(U[k]) and U[k] > L[k]:                                # where in its bracket the exact value fell
                    rho, seen = (rho * (seen + 1) + float((value - L[k]) / (U[k] - L[k]))) / (seen + 2), seen + 1
    return hi, mean_tally, int(exact.sum())


def mean_point_d1_tally(block):
    """D1 tally of the rounded mean point without a KD-tree (which nearest original point is taken does not matter for D1)."""
    a = np.asarray(block)[:, :3].astype(np.float64)
    sq = PM.squared_norms(a - np.round(np.mean(a, axis=0)))
    tally = np.zeros(5, np.float64)
    tally[PM.N_B], tally[PM.D1_AB], tally[PM.D1_BA] = 1, sq.sum(), sq.min()
    return tally


def select_thresholds_from_stats(n_a, tallies, mean_tally, n_thresholds, resolution, opt_metrics, max_deltas):
    """Every decision of the reference's search (model_opt.py:33-73) from the per-threshold tallies of one block: for each
    max_delta the eligible thresholds (all of them when none is eligible or max_delta is None), for each metric the first
    minimum over those, replaced by the 'emit nothing' index n_thresholds - 1 when the single mean point scores better."""
    names = metric_names(opt_metrics, max_deltas)
    nothing = n_thresholds - 1
    tallies = np.asarray(tallies, np.float64).reshape(-1, 5)
    if len(tallies) == 0:
        return names, [nothing] * len(opt_metrics)          # (sic) not x len(max_deltas): model_opt.py:35-36
    groups = sorted({m.split('_', 1)[0] for m in opt_metrics})
    table = PM.metrics_table(n_a, tallies, resolution - 1, groups)
    guard = PM.metrics_table(n_a, mean_tally, resolution - 1, groups)
    everything = np.arange(len(tallies))
    best = []
    for max_delta in max_deltas:
        pool = everything
        if max_delta is not None:
            ok = everything[ratio_eligible(tallies[:, PM.N_B], n_a, max_delta)]
            pool = ok if len(ok) else everything
        for m in opt_metrics:
            column = table[m][pool]
            k = int(np.argmin(column))
            best.append(nothing if column[k] > guard[m] else int(pool[k]))
    return names, best


def compute_optimal_thresholds(block, x_hat, thresholds, resolution, normals=None, opt_metrics=['d1_mse'],
                               max_deltas=[np.inf], fixed_threshold=False):
    """The reference's entry point (model_opt.py:21): (names '{metric}_{max_delta}', threshold index per name)."""
    validate_opt_metrics(opt_metrics, with_normals=normals is not None)
    assert len(max_deltas) > 0
    if fixed_threshold:
        return metric_names(opt_metrics, max_deltas), [len(thresholds) // 2] * (len(max_deltas) * len(opt_metrics))
    tallies, mean_tally = host_threshold_stats(block, x_hat, thresholds, normals)
    return select_thresholds_from_stats(len(block), tallies, mean_tally, len(thresholds), resolution, opt_metrics, max_deltas)


class HostSearchPool:
    """Persistent pool of worker PROCESSES (pcc_geo_cnn_v2_amd.search_worker) for the host KD-tree part of the threshold
    search: the blocks of a cloud are independent, the reference searches them one after the other (model_types.py:192-212).
    Subprocesses with pipes instead of multiprocessing: no fork of a process that holds a HIP context, no re-import of the
    caller's __main__.  Jobs go through one queue (`submit` returns a future), so the encoder can hand over the blocks of many
    chunks while the GPU works on the next ones; a worker that dies is replaced and the failure reported with its exit status."""

    def __init__(self, n_workers):
        import queue
        import threading
        self.procs = [self._spawn() for _ in range(max(1, int(n_workers)))]
        self._q = queue.Queue()
        self._threads = [threading.Thread(target=self._drive, args=(w,), daemon=True) for w in range(len(self.procs))]
        for t in self._threads:
            t.start()

    @staticmethod
    def _spawn():
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1')
        return subprocess.Popen([sys.executable, '-m', 'pcc_geo_cnn_v2_amd.search_worker'], stdin=subprocess.PIPE,
                                stdout=subprocess.PIPE, env=env)

    def _exchange(self, w, job):
        import pickle
        import struct
        p = self.procs[w]
        try:
            data = pickle.dumps(job, protocol=4)
            p.stdin.write(struct.pack('<Q', len(data)) + data)
            p.stdin.flush()
            hdr = p.stdout.read(8)
            body = p.stdout.read(struct.unpack('<Q', hdr)[0]) if len(hdr) == 8 else b''
            if len(hdr) < 8 or len(body) < struct.unpack('<Q', hdr)[0]:
                raise EOFError('short read')
        except (EOFError, BrokenPipeError, OSError, ValueError) as e:
            p.kill()
            code = p.wait()
            self.procs[w] = self._spawn()       # the pool stays usable
            raise RuntimeError(f'threshold search worker {w} died (exit status {code}, {e}); it has been restarted') from e
        return pickle.loads(body)

    def _drive(self, w):
        while True:
            item = self._q.get()
            if item is None:
                return
            job, fut = item
            try:
                status, a, b = self._exchange(w, job)
                if status != 'ok':
                    raise AssertionError(f'threshold search worker: {a}')
                fut.set_result((a, b))
            except BaseException as e:      # delivered to whoever waits for this job
                fut.set_exception(e)

    def submit(self, job):
        """job: ('decide', block, x_hat, thresholds, resolution, with_normals, opt_metrics, max_deltas) -> (names, best)
                ('tally', block, x_hat, thresholds, with_normals)                                      -> (tallies, mean_tally)
           (a bare 7-tuple is a 'decide' job).  Returns a concurrent.futures.Future."""
        from concurrent.futures import Future
        assert self.procs, 'pool is closed'
        fut = Future()
        self._q.put((job, fut))
        return fut

    def map(self, jobs):
        """Results of `jobs` in order (first failure re-raised)."""
        futs = [self.submit(j) for j in jobs]
        return [f.result() for f in futs]

    def close(self):
        for _ in getattr(self, '_threads', []):
            self._q.put(None)
        for t in getattr(self, '_threads', []):
            t.join(timeout=5)
        self._threads = []
        for p in self.procs:
            try:
                p.stdin.close()
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.procs = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def gpu_search_supported(opt_metrics, dhw):
    """The GPU search (csrc/threshold_search.hip) covers grids up to 128^3: the d1_* metrics by exact distance transforms, the
    d2_* metrics (round 4) by nearest-index transforms with a stated tie rule (d2_on_gpu)."""
    return max(dhw) <= 128


D2_SEARCH = None      # 'kdtree' | 'gpu': set by the CLIs' --d2_search; None = the environment (PCC_D2_GPU=1 -> 'gpu') or 'kdtree'
_d2_logged = False


def d2_on_gpu(mode=None):
    """Where the d2_* statistics of the adaptive search come from.  DEFAULT (round 5): scipy KD-trees in the host worker pool
    -- the reference's own neighbour picks (pc_metric.py:114: among equidistant nearest neighbours it takes whatever the tree traversal
    returns), hence the reference's decisions (pinned by tests/golden/model_opt_d2.npz).  OPT-IN (--d2_search gpu / PCC_D2_GPU=1): the
    nearest-index transforms of csrc/threshold_search.hip, 8-18x faster per cloud, ties to the lowest (x, y, z).  The two agree exactly
    where no tie occurs (tests/golden/model_opt_d2_tiefree.npz); on voxelised surfaces ties are the rule: measured (tools/d2_tie_table.py,
    DESIGN_HISTORY.md 3.8) the d2_mse decision differs on 82-89 % of the blocks of a 1024^3 cloud and the D2 PSNR of the d2-optimised stream,
    evaluated with the reference's metric, drops by 0.2-1.2 dB -- the search then minimises a function other than the one it is judged by.
    d1_* metrics never depend on the pick and always come from the GPU."""
    global _d2_logged
    # precedence: the caller's model (`model.d2_search`, set by the CLI's --d2_search), the module default (tools, tests), the environment
    if mode is None and D2_SEARCH is None and os.environ.get('PCC_D2_HOST') is not None and os.environ.get('PCC_D2_GPU') is not None:
        logger.warning('PCC_D2_HOST and PCC_D2_GPU are both set: PCC_D2_HOST wins (kdtree)')
    mode = mode or D2_SEARCH or ('kdtree' if os.environ.get('PCC_D2_HOST') is not None else 'gpu' if os.environ.get('PCC_D2_GPU') is not None else 'kdtree')
    if not _d2_logged:
        _d2_logged = True
        if mode != 'gpu':
            logger.info('d2_* threshold search: statistics from scipy KD-trees in the host worker pool (the reference\'s neighbour picks; 8-18x slower per cloud than '
                        '--d2_search gpu, whose decisions differ on most blocks of a voxelised surface: DESIGN_HISTORY.md 3.8)')
    if mode == 'gpu' and _d2_logged != 'gpu':
        _d2_logged = 'gpu'
        logger.warning('d2_* threshold search on the GPU (opt-in): equidistant nearest neighbours resolve to the lowest (x, y, z), not to '
                       'scipy\'s KD-tree pick as in the reference; decisions differ on most blocks of a voxelised surface (DESIGN_HISTORY.md 3.8)')
    return mode == 'gpu'


def d1_tallies_gpu(ctx, blocks, x_hat, thresholds):
    """Per-block D1 tallies from exact distance transforms on the GPU (csrc/threshold_search.hip).  blocks: list of
    (n_i, >=3) arrays; x_hat: (B,D,H,W) float32 device tensor (clipped inside, like model_types.py:202).
    Returns [float64[T_i, 5]] -- the d2 slots are zero."""
    import torch
    from . import ops
    # C-contiguous (n,3): np.argwhere-style inputs are transposed views and would otherwise stay Fortran-ordered
    pts = np.ascontiguousarray(np.concatenate([np.asarray(b)[:, :3] for b in blocks]).astype(np.uint32).astype(np.int32))
    bof = np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(blocks)])
    thr = torch.from_numpy(np.asarray(thresholds).astype(np.float32)).to(ctx.device)
    s_ab, s_ba, n_b, tcount = ops.d1_threshold_stats(ctx, x_hat, thr, torch.from_numpy(pts).to(ctx.device),
                                                     torch.from_numpy(bof).to(ctx.device), clip=True)
    out = []
    for i in range(len(blocks)):
        T = int(tcount[i])
        t = np.zeros((T, 5), np.float64)
        t[:, PM.N_B], t[:, PM.D1_AB], t[:, PM.D1_BA] = n_b[i][:T], s_ab[i][:T], s_ba[i][:T]
        out.append(t)
    return out


def d12_tallies_gpu(ctx, blocks, x_hat, thresholds):
    """Per-block D1 AND D2 tallies on the GPU.  blocks: list of (n_i, 6) arrays, xyz + normals (the last three columns, as
    model_types.get_normals_if takes them).  Returns [float64[T_i, 5]].  Ties between equidistant nearest neighbours go to the
    lowest (x, y, z) -- the reference takes scipy's pick (pc_metric.py:114); D1 does not depend on the pick."""
    import torch
    from . import ops
    xyz = np.ascontiguousarray(np.concatenate([np.asarray(b)[:, :3] for b in blocks]).astype(np.uint32).astype(np.int32))
    nrm = np.ascontiguousarray(np.concatenate([np.asarray(b)[:, np.asarray(b).shape[1] - 3:] for b in blocks]).astype(np.float32))
    sizes = np.array([len(b) for b in blocks], np.int64)
    bof = np.repeat(np.arange(len(blocks), dtype=np.int32), sizes)
    start = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dev = ctx.device
    thr = torch.from_numpy(np.asarray(thresholds).astype(np.float32)).to(dev)
    s_ab, s_ba, n_b, tcount, d2_ab, d2_ba = ops.d12_threshold_stats(
        ctx, x_hat, thr, torch.from_numpy(xyz).to(dev), torch.from_numpy(bof).to(dev), torch.from_numpy(start).to(dev),
        torch.from_numpy(nrm).to(dev), clip=True)
    out = []
    for i in range(len(blocks)):
        T = int(tcount[i])
        t = np.zeros((T, 5), np.float64)
        t[:, PM.N_B], t[:, PM.D1_AB], t[:, PM.D1_BA] = n_b[i][:T], s_ab[i][:T], s_ba[i][:T]
        t[:, PM.D2_AB], t[:, PM.D2_BA] = d2_ab[i][:T], d2_ba[i][:T]
        out.append(t)
    return out


def mean_point_tally(block, with_normals):
    """Tally of the rounded mean point (the guard of model_opt.py:59-68) without a KD-tree: its nearest original point; among
    equidistant ones the lowest (x, y, z) in lexicographic order -- the tie rule of the GPU path (csrc/threshold_search.hip), whatever
    order the block's points are stored in (a plain argmin is that rule only for sorted blocks)."""
    blk = np.asarray(block)
    if not with_normals:
        return mean_point_d1_tally(blk)
    a = blk[:, :3]
    mean_point = np.round(np.mean(a, axis=0))[np.newaxis, :]
    d = PM.squared_norms(a - mean_point)
    near = np.flatnonzero(d == d.min())
    to_a = np.array([int(near[np.lexsort((a[near, 2], a[near, 1], a[near, 0]))[0]])], np.int64)
    return PM.pair_tally(a, mean_point, np.zeros(len(a), np.int64), to_a, blk[:, blk.shape[1] - 3:])


def decide_from_tallies(blocks, d1, n_thresholds, resolution, opt_metrics, max_deltas, d2_stats=None, gpu_d2=False):
    """Decisions of a chunk from the GPU's D1 tallies (d1_tallies_gpu), merged with the host pool's (tallies, mean_tally) per
    block when d2_* metrics are requested: their D2 slots go into the GPU's table; the D1 slots of both sources are the same
    integers, which is asserted.  Returns (names, [best thresholds per block])."""
    validate_opt_metrics(opt_metrics, with_normals=d2_stats is not None or gpu_d2)
    names, best = metric_names(list(opt_metrics), list(max_deltas)), []
    for i, blk in enumerate(blocks):
        # gpu_d2: the tallies of d12_tallies_gpu already hold their D2 slots; the guard point follows the same tie rule
        tallies, mean_tally = d1[i], mean_point_tally(blk, gpu_d2)
        if d2_stats is not None:
            host_t, host_mean = d2_stats[i]
            assert len(host_t) == len(tallies), 'host and GPU disagree on the number of non-empty level sets'
            assert np.array_equal(host_t[:, :3], tallies[:, :3]), 'host KD-tree and GPU distance-transform D1 sums differ'
            tallies[:, [PM.D2_AB, PM.D2_BA]] = host_t[:, [PM.D2_AB, PM.D2_BA]]
            mean_tally[[PM.D2_AB, PM.D2_BA]] = host_mean[[PM.D2_AB, PM.D2_BA]]
        names, bt = select_thresholds_from_stats(len(blk), tallies, mean_tally, n_thresholds, resolution, list(opt_metrics), list(max_deltas))
        best.append(bt)
    return names, best


def compute_optimal_thresholds_gpu(ctx, blocks, x_hat, thresholds, resolution, opt_metrics=('d1_mse',),
                                   max_deltas=(np.inf,), d2_stats=None):
    """compute_optimal_thresholds for a batch of blocks with the KD-tree work of the d1_* metrics replaced by exact distance
    transforms on the GPU.  `d2_stats`: per block (tallies, mean_tally) from the host pool when d2_* metrics are requested -- or
    a callable returning that list, evaluated after the GPU work has been issued.  Returns (names, [best thresholds per block])."""
    d1 = d1_tallies_gpu(ctx, blocks, x_hat, thresholds)
    if callable(d2_stats):
        d2_stats = d2_stats()
    return decide_from_tallies(blocks, d1, len(thresholds), resolution, opt_metrics, max_deltas, d2_stats)
