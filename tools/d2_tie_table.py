"""GPU box: what the GPU D2 tie rule (lowest (x, y, z) among equidistant nearest neighbours, csrc/threshold_search.hip) changes against the
reference's picks (scipy KD-tree traversal, /root/reference/src/utils/pc_metric.py:109-131) -- VERDICT r04 item 5.

 (1) tests/golden/model_opt_d2.npz (four voxelised-shell blocks, every level set tied) and model_opt_d2_tiefree.npz (six blocks without
     ties): per (block, metric_maxdelta) the threshold index of the REFERENCE (fixture), of the GPU search, and of the host KD-tree path
     (--d2_search kdtree / PCC_D2_HOST=1);
 (2) two 1024^3 level-4 clouds through compress_blocks with ['d1_mse', 'd2_mse'] and normals under both searches: blocks whose d2
     decision differs, and the D1 / D2 PSNR of the decoded cloud each search ends up with (the stream length does not depend on the
     threshold index: one byte per block either way).
 Prints markdown (DESIGN_HISTORY.md 3.8); the counts are asserted by tests/test_threshold_search_gpu.py."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops, model_opt
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
ctx = ops.get_context(torch.device('cuda', 0))
thr = np.linspace(0, 1.0, 256)

print('### (1) fixtures: threshold index per (block, metric): reference / GPU search / host KD-tree search\n')
for fn in ('model_opt_d2.npz', 'model_opt_d2_tiefree.npz'):
    g = np.load(os.path.join(G, fn), allow_pickle=True)
    mets, deltas = [str(m) for m in g['opt_metrics']], [float(d) for d in g['max_deltas']]
    names = [str(n) for n in g['s0_names']]
    print(f'`{fn}`\n\n| block | ' + ' | '.join(names) + ' | d2 decisions GPU != reference |\n|---|' + '---|' * (len(names) + 1))
    tot = totd2 = 0
    for i in range(int(g['n_cases'][0])):
        blk, xh = g[f's{i}_block'], g[f's{i}_x_hat']
        ref = [int(b) for b in g[f's{i}_best']]
        tallies = model_opt.d12_tallies_gpu(ctx, [blk], torch.from_numpy(xh[None]).to(ctx.device), thr)[0]
        _, gpu = model_opt.decide_from_tallies([blk], [tallies], len(thr), 64, mets, deltas, gpu_d2=True)
        _, host = model_opt.compute_optimal_thresholds(blk, xh, thr, 64, normals=blk[:, 3:6], opt_metrics=mets, max_deltas=deltas)
        d = sum(1 for n, a, b in zip(names, gpu[0], ref) if n.startswith('d2_') and a != b)
        tot += d; totd2 += sum(n.startswith('d2_') for n in names)
        assert all(a == b for n, a, b in zip(names, gpu[0], ref) if n.startswith('d1_')), 'a d1 decision differs'
        print(f'| {i} ({len(blk)} pts{", float32" if blk.dtype != np.float64 else ""}) | ' + ' | '.join(f'{r} / {a} / {h}' for r, a, h in zip(ref, gpu[0], host)) + f' | {d} |')
    print(f'\n{tot} of {totd2} d2 decisions differ from the reference in `{fn}`.\n')

print('### (2) whole clouds: compress_blocks(opt_metrics = [d1_mse, d2_mse], normals) under both searches\n')
R, level, res = 1024, 4, 64
model = ModelConfigType['c3p'].build(batch_size=32); model.compress([1, 1, res, res, res])
model.set_weights(bench.synthetic_weights(model))
# The network here has synthetic weights: its x_hat is unrelated to the input and the mean-point guard would fire on every block (index
# 255 everywhere: nothing to compare).  What the SEARCH sees is therefore replaced by the output of a plausible decoder -- the input
# occupancy blurred (sigma 0.8) x 2.2 + N(0, 0.03), clipped, the recipe of the golden fixtures -- while the bitstream stays the network's.
_orig_encode = model._encode_batch


def _plausible_x_hat(x):
    k = torch.exp(-torch.arange(-2, 3, device=x.device, dtype=torch.float32) ** 2 / (2 * 0.8 ** 2)); k /= k.sum()
    v = x[:, None]
    for ax in range(3):
        shape = [1, 1, 1, 1, 1]; shape[2 + ax] = 5
        pad = [0, 0, 0, 0, 0, 0]; pad[2 * (2 - ax)] = pad[2 * (2 - ax) + 1] = 2
        v = torch.nn.functional.conv3d(torch.nn.functional.pad(v, pad), k.reshape(shape))
    g = torch.Generator(device=x.device).manual_seed(int(x.sum().item()) & 0xffff)
    return (v[:, 0] * 2.2 + 0.03 * torch.randn(x.shape, device=x.device, generator=g)).clamp_(0, 1).contiguous()


def _encode_with_plausible_x_hat(ctx_, x, debug=False, thr=None, slot=0):
    enc = _orig_encode(ctx_, x, debug, thr=thr, slot=slot)
    enc['x_hat'] = _plausible_x_hat(x)
    return enc


model._encode_batch = _encode_with_plausible_x_hat


def cloud(kind):
    rng = np.random.default_rng(0)
    u = rng.standard_normal((3_000_000 if kind == 'smooth' else 1_500_000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    rad = 200 if kind == 'smooth' else 200 + 6 * np.sin(9 * u[:, :1]) * np.cos(7 * u[:, 1:2]) + rng.normal(0, 0.6, (len(u), 1))
    pts, first = np.unique(np.round(u * rad + np.array([512, 500, 520])).astype(np.int64), axis=0, return_index=True)
    return np.hstack([pts.astype(np.float64), u[first]])


print('| cloud | blocks | search | s / cloud | blocks whose d2_mse decision differs from the KD-tree search | d1_psnr (d1-optimised stream) | d2_psnr (d2-optimised stream) |\n|---|---|---|---|---|---|---|')
for kind in ('smooth', 'rough'):
    c = cloud(kind)
    blocks, binstr = partition_octree(c, [0, 0, 0], [R] * 3, level)
    res_by = {}
    for mode in ('kdtree', 'gpu', 'kdtree', 'gpu', 'kdtree-unpruned'):          # (first pair = warm-up of pool and kernels)
        model_opt.D2_SEARCH = mode.split('-')[0]
        os.environ.pop('PCC_D2_NO_PRUNE', None)
        if mode.endswith('unpruned'): os.environ['PCC_D2_NO_PRUNE'] = '1'
        model.search_trees_built = model.search_trees_total = 0
        torch.cuda.synchronize(); t0 = time.perf_counter()
        data, meta, _ = model.compress_blocks(ctx, blocks, binstr, c, R, level, with_normals=True, opt_metrics=['d1_mse', 'd2_mse'],
                                              max_deltas=[np.inf], need_points=False)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        res_by[mode] = (dt, data, meta, model.search_trees_built, model.search_trees_total)
    thr_of = lambda data: [[t for _, t in d] for d in data]
    tk, tg = thr_of(res_by['kdtree'][1]), thr_of(res_by['gpu'][1])
    os.environ.pop('PCC_D2_NO_PRUNE', None)
    assert thr_of(res_by['kdtree-unpruned'][1]) == tk, 'the bound-pruned search took a different decision'
    for mode in ('kdtree-unpruned', 'kdtree', 'gpu'):
        dt, data, meta, built, total = res_by[mode]
        m1, m2 = meta[0]['metrics'], meta[-1]['metrics']
        differ = sum(a != b for a, b in zip(tk[-1], tg[-1]))
        hist = np.bincount(np.clip(np.array(thr_of(data)[-1]), 0, 255) // 32, minlength=8)
        label = mode + (f' (bound-pruned: {built} of {total} A->B trees built)' if mode == 'kdtree' else '')
        print(f'| {kind} shell, {len(c)} points | {len(blocks)} | {label} | {dt:.2f} | {differ if mode == "gpu" else "-"} | {m1["d1_psnr"]:.4f} | {m2.get("d2_psnr", float("nan")):.4f} (d2 indexes by 32s: {hist.tolist()}) |')
    print(f'| | | | | d1_mse decisions differing: {sum(a != b for a, b in zip(tk[0], tg[0]))}; bytes {sum(len(s) for ss, _ in res_by["kdtree"][1][0] for s in ss)} vs {sum(len(s) for ss, _ in res_by["gpu"][1][0] for s in ss)} | | |')
model_opt.D2_SEARCH = None
