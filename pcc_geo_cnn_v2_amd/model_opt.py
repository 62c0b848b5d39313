"""Per-block threshold search -- same decisions as /root/reference/src/model_opt.py:9-77.

Host (numpy + KD-tree) restatement used when `fixed_threshold=False`.  The comparisons
`x_hat > t` are done in float32 like the reference under its pinned numpy 1.18.
"""
import logging

import numpy as np
from scipy.spatial import cKDTree

from .utils.pc_metric import compute_metrics, validate_opt_metrics

logger = logging.getLogger(__name__)


def _gt(x_hat, t):
    x_hat = np.asarray(x_hat)
    if x_hat.dtype == np.float32:
        return x_hat > np.float32(t)
    return x_hat > t


def build_points_threshold(x_hat, thresholds, len_block, max_delta=np.inf):
    pa_list = []
    for i, t in enumerate(thresholds):
        pa = np.argwhere(_gt(x_hat, t)).astype('float32')
        if len(pa) == 0:
            break
        len_ratio = len(pa) / len_block
        if (1 / max_delta) < len_ratio < max_delta:
            pa_list.append((i, pa))
    return pa_list


def compute_optimal_thresholds(block, x_hat, thresholds, resolution, normals=None, opt_metrics=['d1_mse'],
                               max_deltas=[np.inf], fixed_threshold=False):
    validate_opt_metrics(opt_metrics, with_normals=normals is not None)
    assert len(max_deltas) > 0
    best_thresholds = []
    ret_opt_metrics = [f'{opt_metric}_{max_delta}' for max_delta in max_deltas for opt_metric in opt_metrics]
    if fixed_threshold:
        half_thr = len(thresholds) // 2
        return ret_opt_metrics, [half_thr] * len(max_deltas) * len(opt_metrics)

    pa_list = build_points_threshold(x_hat, thresholds, len(block))
    max_threshold_idx = len(thresholds) - 1
    if len(pa_list) == 0:
        return ret_opt_metrics, [max_threshold_idx] * len(opt_metrics)

    t1 = cKDTree(block[:, :3], balanced_tree=False)
    pa_metrics = [compute_metrics(block[:, :3], pa, resolution - 1, p1_n=normals, t1=t1) for _, pa in pa_list]
    mean_point = np.round(np.mean(block[:, :3], axis=0))[np.newaxis, :]
    mean_metrics = compute_metrics(block[:, :3], mean_point, resolution - 1, p1_n=normals, t1=t1)

    for max_delta in max_deltas:
        cur_pa_list, cur_pa_metrics = pa_list, pa_metrics
        if max_delta is not None:
            filt = build_points_threshold(x_hat, thresholds, len(block), max_delta)
            if len(filt) > 0:
                cur_pa_list = filt
                # NOTE: the reference indexes pa_metrics with the THRESHOLD index (model_opt.py:46-47),
                # which equals the list position because pa_list has no gaps before its first empty set.
                cur_pa_metrics = [pa_metrics[i] for i, _ in filt]
        for opt_metric in opt_metrics:
            best_threshold_idx = int(np.argmin([x[opt_metric] for x in cur_pa_metrics]))
            cur_best_metric = cur_pa_metrics[best_threshold_idx][opt_metric]
            # failure case: a single mean point beats the network output -> output no points (:59-68)
            if cur_best_metric > mean_metrics[opt_metric]:
                final_idx = max_threshold_idx
            else:
                final_idx = cur_pa_list[best_threshold_idx][0]
            best_thresholds.append(final_idx)
    assert len(ret_opt_metrics) == len(best_thresholds)
    return ret_opt_metrics, best_thresholds


class HostSearchPool:
    """Persistent pool of worker PROCESSES (pcc_geo_cnn_v2_amd.search_worker) for the host KD-tree threshold search: the blocks
    of a chunk are independent, the reference searches them one after the other (model_types.py:192-212).  Subprocesses
    with pipes instead of multiprocessing: no fork of a process that holds a HIP context, no re-import of the caller's
    __main__.  Decisions are those of compute_optimal_thresholds (same code, same scipy)."""

    def __init__(self, n_workers):
        import os
        import subprocess
        import sys
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''), OMP_NUM_THREADS='1')
        self.procs = [subprocess.Popen([sys.executable, '-m', 'pcc_geo_cnn_v2_amd.search_worker'], stdin=subprocess.PIPE,
                                       stdout=subprocess.PIPE, env=env) for _ in range(max(1, int(n_workers)))]

    def map(self, jobs):
        """jobs: list of (block, x_hat, thresholds, resolution, with_normals, opt_metrics, max_deltas).  Returns
        [(names, best)] in order.  Each worker handles jobs i, i + W, i + 2W, ... through its own pipe."""
        import pickle
        import struct
        from concurrent.futures import ThreadPoolExecutor
        W = len(self.procs)
        out = [None] * len(jobs)

        def drive(w):
            p = self.procs[w]
            for i in range(w, len(jobs), W):
                data = pickle.dumps(jobs[i], protocol=4)
                p.stdin.write(struct.pack('<Q', len(data)) + data)
                p.stdin.flush()
                n = struct.unpack('<Q', p.stdout.read(8))[0]
                status, a, b = pickle.loads(p.stdout.read(n))
                if status != 'ok':
                    raise AssertionError(f'threshold search worker: {a}')
                out[i] = (a, b)

        with ThreadPoolExecutor(max_workers=W) as ex:
            list(ex.map(drive, range(min(W, len(jobs)))))
        return out

    def close(self):
        for p in self.procs:
            try:
                p.stdin.close()
                p.wait(timeout=5)
            except Exception:
                p.kill()
        self.procs = []

    def __del__(self):
        self.close()


def select_thresholds_from_stats(block, s_ab, s_ba, n_b, tcount, n_thresholds, resolution, opt_metrics, max_deltas):
    """The decision logic of compute_optimal_thresholds (model_opt.py:33-73) applied to exact per-threshold D1 sums
    (integers) of one block, as produced by ops.d1_threshold_stats.  Only d1_* metrics (no normals)."""
    ret_opt_metrics = [f'{opt_metric}_{max_delta}' for max_delta in max_deltas for opt_metric in opt_metrics]
    max_threshold_idx = n_thresholds - 1
    T = int(tcount)                                   # thresholds 0..T-1 have a non-empty decoded set (pa_list)
    if T == 0:
        return ret_opt_metrics, [max_threshold_idx] * len(opt_metrics)
    nA = len(block)
    sab, sba, nb = s_ab[:T].astype(np.float64), s_ba[:T].astype(np.float64), n_b[:T].astype(np.float64)
    met = {'d1_sum_AB': sab, 'd1_sum_BA': sba, 'd1_sum_max': np.maximum(sab, sba), 'd1_sum_mean': (sab + sba) / 2,
           'd1_mse_AB': sab / nA, 'd1_mse_BA': sba / nb}
    met['d1_mse'] = np.maximum(met['d1_mse_AB'], met['d1_mse_BA'])
    # single mean point (failure guard, :59-68)
    pts = np.asarray(block)[:, :3].astype(np.float64)
    mp = np.round(np.mean(pts, axis=0))
    d = np.sum((pts - mp) ** 2, axis=1)
    m_ab, m_ba = float(np.sum(d)), float(np.min(d))
    mean_met = {'d1_sum_AB': m_ab, 'd1_sum_BA': m_ba, 'd1_sum_max': max(m_ab, m_ba), 'd1_sum_mean': (m_ab + m_ba) / 2,
                'd1_mse_AB': m_ab / nA, 'd1_mse_BA': m_ba / 1.0}
    mean_met['d1_mse'] = max(mean_met['d1_mse_AB'], mean_met['d1_mse_BA'])
    best_thresholds = []
    for max_delta in max_deltas:
        idxs = np.arange(T)
        if max_delta is not None:
            ratio = nb / nA
            elig = idxs[((1 / max_delta) < ratio) & (ratio < max_delta)]
            if len(elig) > 0:
                idxs = elig
        for opt_metric in opt_metrics:
            vals = met[opt_metric][idxs]
            j = int(np.argmin(vals))
            if vals[j] > mean_met[opt_metric]:
                best_thresholds.append(max_threshold_idx)
            else:
                best_thresholds.append(int(idxs[j]))
    return ret_opt_metrics, best_thresholds


def gpu_search_supported(opt_metrics, normals, dhw):
    return normals is None and all(m.startswith('d1_') for m in opt_metrics) and max(dhw) <= 128


def compute_optimal_thresholds_gpu(ctx, blocks, x_hat, thresholds, resolution, opt_metrics=('d1_mse',),
                                   max_deltas=(np.inf,)):
    """compute_optimal_thresholds for a batch of blocks with all KD-tree work replaced by exact distance transforms on
    the GPU.  blocks: list of (n_i, >=3) arrays; x_hat: (B,D,H,W) float32 device tensor (clipped inside, like
    model_types.py:202).  Returns (ret_opt_metrics, [best thresholds per block])."""
    import torch
    from . import ops
    from .utils.pc_metric import validate_opt_metrics
    validate_opt_metrics(opt_metrics, with_normals=False)
    B = len(blocks)
    # C-contiguous (n,3): np.argwhere-style inputs are transposed views and would otherwise stay Fortran-ordered
    pts = np.ascontiguousarray(np.concatenate([np.asarray(b)[:, :3] for b in blocks]).astype(np.uint32).astype(np.int32))
    bof = np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(blocks)])
    thr = torch.from_numpy(np.asarray(thresholds).astype(np.float32)).to(ctx.device)
    s_ab, s_ba, n_b, tcount = ops.d1_threshold_stats(ctx, x_hat, thr, torch.from_numpy(pts).to(ctx.device),
                                                     torch.from_numpy(bof).to(ctx.device), clip=True)
    names, best = None, []
    for i in range(B):
        names, bt = select_thresholds_from_stats(blocks[i], s_ab[i], s_ba[i], n_b[i], tcount[i], len(thresholds),
                                                 resolution, list(opt_metrics), list(max_deltas))
        best.append(bt)
    return names, best
