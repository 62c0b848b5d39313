"""GPU: exact adaptive-threshold statistics (distance transforms) vs the host KD-tree path, whose results are
pinned by the reference fixtures (tests/golden/model_opt.npz)."""
import os

import numpy as np
import pytest
import torch
from scipy.ndimage import gaussian_filter
from scipy.spatial import cKDTree

from pcc_geo_cnn_v2_amd import model_opt, ops

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _case(rng, R, npts, sharp):
    block = np.unique(rng.integers(0, R, (npts, 3)), axis=0).astype(np.float64)
    dense = np.zeros((R, R, R), np.float32)
    dense[tuple(block.astype(int).T)] = 1
    x_hat = (gaussian_filter(dense, sharp) * 2.5 + rng.normal(0, 0.02, dense.shape)).astype(np.float32)
    return block, x_hat


def _host_stats(block, x_hat, thresholds):
    xh = np.clip(x_hat, 0, 1)
    t1 = cKDTree(block)
    out = []
    for t in thresholds:
        pa = np.argwhere(xh > np.float32(t)).astype(np.float64)
        if len(pa) == 0:
            break
        d_ab, _ = cKDTree(pa).query(block)
        d_ba, _ = t1.query(pa)
        out.append((int(round(np.sum(d_ab ** 2))), int(round(np.sum(d_ba ** 2))), len(pa)))
    return out


@pytest.mark.parametrize('R', [16, 32, 64])
def test_stats_match_kdtree_exactly(ctx, R):
    rng = np.random.default_rng(R)
    thresholds = np.linspace(0, 1.0, 256)
    blocks, xs = zip(*[_case(rng, R, n, s) for n, s in [(40, 0.7), (400, 1.0), (3, 0.5), (900, 1.5)]])
    x_hat = torch.from_numpy(np.stack(xs)).to(ctx.device)
    pts = np.concatenate(blocks).astype(np.int32)
    bof = np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(blocks)])
    s_ab, s_ba, n_b, tcount = ops.d1_threshold_stats(ctx, x_hat, torch.from_numpy(thresholds.astype(np.float32)).to(ctx.device),
                                                     torch.from_numpy(pts).to(ctx.device), torch.from_numpy(bof).to(ctx.device))
    for i, (block, xh) in enumerate(zip(blocks, xs)):
        ref = _host_stats(block, xh, thresholds)
        assert tcount[i] == len(ref)
        for t, (ab, ba, n) in enumerate(ref):
            assert (s_ab[i, t], s_ba[i, t], n_b[i, t]) == (ab, ba, n), (i, t)


def test_decisions_match_host_search_and_reference_fixtures(ctx):
    g = np.load(os.path.join(G, 'model_opt.npz'))
    thresholds = np.linspace(0, 1.0, 256)
    blocks = [g[f'm{i}_block'] for i in range(int(g['n_cases'][0]))]
    xs = np.stack([g[f'm{i}_x_hat'] for i in range(len(blocks))])
    names, best = model_opt.compute_optimal_thresholds_gpu(ctx, blocks, torch.from_numpy(xs).to(ctx.device), thresholds, 64,
                                                           ['d1_mse', 'd1_sum_mean'], [np.inf])
    for i in range(len(blocks)):
        assert names == list(g[f'm{i}_names_fixed0']) and best[i] == list(g[f'm{i}_best_fixed0'])   # == the reference's own answers
    # more metrics and max_delta filters against the host restatement
    rng = np.random.default_rng(1)
    blocks, xs = zip(*[_case(rng, 32, n, s) for n, s in [(300, 0.8), (50, 1.2), (1500, 1.0)]])
    mets, deltas = ['d1_mse', 'd1_sum_AB', 'd1_sum_BA', 'd1_mse_BA', 'd1_sum_max'], [np.inf, 2.0, 1.2]
    names, best = model_opt.compute_optimal_thresholds_gpu(ctx, list(blocks), torch.from_numpy(np.stack(xs)).to(ctx.device),
                                                           thresholds, 64, mets, deltas)
    for b, xh, bt in zip(blocks, xs, best):
        hn, hb = model_opt.compute_optimal_thresholds(b, np.clip(xh, 0, 1), thresholds, 64, opt_metrics=mets, max_deltas=deltas)
        assert hn == names and hb == bt


def test_empty_and_degenerate_blocks(ctx):
    thresholds = np.linspace(0, 1.0, 256)
    block = np.array([[3, 4, 5], [3, 4, 6]], float)
    xs = np.zeros((2, 16, 16, 16), np.float32)          # block 0: nothing above threshold 0 -> [255]
    xs[1, 8, 8, 8] = 0.9                                 # block 1: one far voxel -> mean point wins -> 255
    names, best = model_opt.compute_optimal_thresholds_gpu(ctx, [block, block], torch.from_numpy(xs).to(ctx.device), thresholds, 16)
    for b, xh, bt in zip([block, block], xs, best):
        assert model_opt.compute_optimal_thresholds(b, xh, thresholds, 16, opt_metrics=['d1_mse'], max_deltas=[np.inf])[1] == bt
    assert best[0] == [255]


def test_fortran_ordered_blocks(ctx):
    """np.argwhere returns transposed views; a concatenation of those stays column-major unless forced.  The host
    wrappers must hand the kernels row-major (n,3) points regardless."""
    rng = np.random.default_rng(7)
    thresholds = np.linspace(0, 1.0, 256)
    blocks, xs = [], []
    for n, s in [(600, 0.9), (60, 1.1)]:
        b, xh = _case(rng, 32, n, s)
        dense = np.zeros((32, 32, 32), np.uint8)
        dense[tuple(b.astype(int).T)] = 1
        bf = np.asfortranarray(np.argwhere(dense).astype(np.float64))
        assert not bf.flags['C_CONTIGUOUS']
        blocks.append(bf)
        xs.append(xh)
    names, best = model_opt.compute_optimal_thresholds_gpu(ctx, blocks, torch.from_numpy(np.stack(xs)).to(ctx.device),
                                                           thresholds, 64, ['d1_mse', 'd1_sum_mean'], [np.inf, 1.5])
    for b, xh, bt in zip(blocks, xs, best):
        hn, hb = model_opt.compute_optimal_thresholds(np.ascontiguousarray(b), np.clip(xh, 0, 1), thresholds, 64,
                                                      opt_metrics=['d1_mse', 'd1_sum_mean'], max_deltas=[np.inf, 1.5])
        assert hn == names and hb == bt
