"""GPU box: seconds per cloud of the adaptive threshold search on the 190-block stand-in for longdress (thin shell @1024^3, level 4),
c3p, with normals -- the reference's experiment setting opt_metrics ['d1_mse', 'd2_mse'] (ev_experiment.yml:47), and d1 only.
  round 2: any normals -> every metric on the host KD-tree pool ('decide' jobs);
  round 3: d1_* from the GPU distance transforms, only the D2 tallies from the host pool ('tally' jobs, B->A neighbours queried once);
  round 4: d2_* on the GPU as well (nearest-index transforms, stated tie rule);
  round 5: that is the OPT-IN (--d2_search gpu / PCC_D2_GPU=1); the default is the round-3 dispatch again (tools/d2_tie_table.py: why)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops, model_opt
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree
ctx = ops.get_context(torch.device('cuda', 0))
R, level, res = 1024, 4, 64
rng = np.random.default_rng(0)
u = rng.standard_normal((3_000_000, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
pts, first = np.unique(np.round(u * 200 + np.array([512, 500, 520])).astype(np.int64), axis=0, return_index=True)
cloud = np.hstack([pts.astype(np.float64), u[first]])
blocks, binstr = partition_octree(cloud, [0, 0, 0], [R] * 3, level)
model = ModelConfigType['c3p'].build(batch_size=32); model.compress([1, 1, res, res, res])
model.set_weights(bench.synthetic_weights(model))
print(f'{len(pts)} points, {len(blocks)} blocks, host cores {os.cpu_count()}, usable {len(os.sched_getaffinity(0))}')
def run(mets, host_only=False):
    if host_only:
        keep = model_opt.gpu_search_supported
        import pcc_geo_cnn_v2_amd.model_types as MTY
        MTY.gpu_search_supported = lambda *a: False
    torch.cuda.synchronize(); t0 = time.perf_counter()
    out = model.encode_block_range(ctx, blocks, R, with_normals=True, opt_metrics=mets, max_deltas=[np.inf])
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    if host_only:
        MTY.gpu_search_supported = keep
    return dt, out[1]
run(['d1_mse'])                                      # warm-up (kernels)
run(['d1_mse', 'd2_mse'])
for mets in (['d1_mse'], ['d1_mse', 'd2_mse']):
    model.host_search_jobs = 0
    model_opt.D2_SEARCH = 'gpu'
    t_new, thr_new = run(mets)
    jobs_new = model.host_search_jobs
    model_opt.D2_SEARCH = 'kdtree'                   # default dispatch: D2 tallies from the host KD-tree pool
    run(mets) if mets[-1].startswith('d2') and 'warm' not in globals() else None
    globals()['warm'] = True
    t_r3, thr_r3 = run(mets)
    model_opt.D2_SEARCH = None
    same_d1 = [a[0] for a in thr_new] == [a[0] for a in thr_r3]
    differ = sum(a != b for a, b in zip(thr_new, thr_r3))
    print(f'{mets}: round 4 (all on the GPU, {jobs_new} host jobs) {t_new:.2f} s / cloud; round-3 dispatch (D2 on the host KD-tree pool) {t_r3:.2f} s / cloud, '
          f'{t_r3 / t_new:.1f}x; d1 decisions equal: {same_d1}; blocks whose d2 decision differs (tie rule): {differ} of {len(thr_new)}')

# round 6: the default 'kdtree' search with and without the bound pruning of the host pool (model_opt.host_threshold_stats_pruned)
model_opt.D2_SEARCH = 'kdtree'
for tag, env in (('pruned (default)', None), ('PCC_D2_NO_PRUNE=1', '1')):
    if env: os.environ['PCC_D2_NO_PRUNE'] = env
    else: os.environ.pop('PCC_D2_NO_PRUNE', None)
    model.search_trees_built = model.search_trees_total = 0
    run(['d1_mse', 'd2_mse'])
    t, thr_ = run(['d1_mse', 'd2_mse'])
    print(f"kdtree search, {tag}: {t:.2f} s / cloud; A->B trees built {getattr(model, 'search_trees_built', 0)} of {getattr(model, 'search_trees_total', 0)} (two runs)")
    globals()['thr_' + ('p' if env is None else 'f')] = thr_
print('decisions equal:', thr_p == thr_f)
