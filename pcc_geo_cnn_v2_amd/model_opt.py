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
