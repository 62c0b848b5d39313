"""D1 / D2 point-cloud metrics -- same numbers as /root/reference/src/utils/pc_metric.py:8-138
(two KD-trees, nearest-neighbour residuals).  `n_jobs=-1` became `workers=-1` in scipy >= 1.6 and
the numba loop of assign_attr (:8-25) is vectorised with np.add.at.
"""
import numpy as np
from scipy.spatial import cKDTree


def assign_attr(attr1, idx1, idx2):
    """Transfers attributes attr1 from x1 to x2.  idx1: (N2,) nearest neighbours of x2 in x1;
    idx2: (N1,) nearest neighbours of x1 in x2."""
    counts = np.zeros(idx1.shape[0])
    attr_sums = np.zeros((idx1.shape[0], attr1.shape[1]))
    np.add.at(counts, idx2, 1)
    np.add.at(attr_sums, idx2, attr1)
    empty = counts == 0
    attr_sums[empty] += attr1[idx1[empty]]
    counts[empty] += 1
    return attr_sums / counts[:, None]


def d1_res(x, y):
    return np.sum((x - y) ** 2, axis=1)


def sum_d1(x, y):
    return np.sum(d1_res(x, y))


def sum_d2(x, y, n):
    return np.sum(np.sum((x - y) * n, axis=1) ** 2)


def psnr(x, max_energy):
    with np.errstate(divide='ignore'):
        return 10 * np.log10(max_energy / x)


# No PSNR as minimizing MSE is equivalent
avail_opt_metrics = [y for x in zip(*[(f'd1_{x}', f'd2_{x}') for x in ['sum_AB', 'sum_BA', 'sum_max', 'sum_mean',
                                                                       'mse_AB', 'mse_BA', 'mse']]) for y in x]


def validate_opt_metrics(opt_metrics, with_normals=False):
    for opt_metric in opt_metrics:
        assert opt_metric in avail_opt_metrics, f'{opt_metric} not found in {avail_opt_metrics}'
        if not with_normals:
            assert not opt_metric.startswith('d2'), f'{opt_metric} not available without normals'


def compute_metrics(p1, p2, r, p1_n=None, t1=None):
    if t1 is None:
        t1 = cKDTree(p1, balanced_tree=False)
    t2 = cKDTree(p2, balanced_tree=False)
    # the reference asks for all cores (n_jobs=-1, pc_metric.py:80-81); for the small per-block queries of the
    # threshold search the thread start-up dominates on many-core hosts, so only large clouds go parallel
    workers = -1 if max(len(p1), len(p2)) > 200000 else 1
    _, idx2 = t2.query(p1, workers=workers)
    _, idx1 = t1.query(p2, workers=workers)

    max_energy = 3 * r * r
    p1_ngb = p2[idx2]
    p2_ngb = p1[idx1]
    d1_sum_AB = sum_d1(p1, p1_ngb)
    d1_sum_BA = sum_d1(p2, p2_ngb)
    d1_mse_AB = d1_sum_AB / p1.shape[0]
    d1_mse_BA = d1_sum_BA / p2.shape[0]
    d1_psnr_AB = psnr(d1_mse_AB, max_energy)
    d1_psnr_BA = psnr(d1_mse_BA, max_energy)
    metrics = {
        'd1_sum_AB': d1_sum_AB, 'd1_sum_BA': d1_sum_BA, 'd1_sum_max': max(d1_sum_AB, d1_sum_BA),
        'd1_sum_mean': (d1_sum_AB + d1_sum_BA) / 2, 'd1_mse_AB': d1_mse_AB, 'd1_mse_BA': d1_mse_BA,
        'd1_mse': max(d1_mse_AB, d1_mse_BA), 'd1_psnr_AB': d1_psnr_AB, 'd1_psnr_BA': d1_psnr_BA,
        'd1_psnr': min(d1_psnr_AB, d1_psnr_BA)}
    if p1_n is not None:
        p2_n = assign_attr(p1_n, idx1, idx2)
        p1_ngb_n = p2_n[idx2]
        p2_ngb_n = p1_n[idx1]
        d2_sum_AB = sum_d2(p1, p1_ngb, p1_ngb_n)
        d2_sum_BA = sum_d2(p2, p2_ngb, p2_ngb_n)
        d2_mse_AB = d2_sum_AB / p1.shape[0]
        d2_mse_BA = d2_sum_BA / p2.shape[0]
        d2_psnr_AB = psnr(d2_mse_AB, max_energy)
        d2_psnr_BA = psnr(d2_mse_BA, max_energy)
        metrics.update({
            'd2_sum_AB': d2_sum_AB, 'd2_sum_BA': d2_sum_BA, 'd2_sum_max': max(d2_sum_AB, d2_sum_BA),
            'd2_sum_mean': (d2_sum_AB + d2_sum_BA) / 2, 'd2_mse_AB': d2_mse_AB, 'd2_mse_BA': d2_mse_BA,
            'd2_mse': max(d2_mse_AB, d2_mse_BA), 'd2_psnr_AB': d2_psnr_AB, 'd2_psnr_BA': d2_psnr_BA,
            'd2_psnr': min(d2_psnr_AB, d2_psnr_BA)})
    return metrics
