"""Where compress_blocks spends a cloud (fixed threshold / adaptive d1): encode_block_range vs select_best_per_opt_metric pieces."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops, model_types as MT
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree, departition_octree
from pcc_geo_cnn_v2_amd.utils import pc_metric
from scipy.spatial import cKDTree
ctx = ops.get_context(torch.device('cuda', 0))
R, level, res = 1024, 4, 64
pts = bench.standin_cloud()
blocks, binstr = partition_octree(pts, [0, 0, 0], [R] * 3, level)
m = ModelConfigType['c3p'].build(batch_size=32); m.compress([1, 1, res, res, res]); m.set_weights(bench.synthetic_weights(m))
print('usable cores', ops.usable_cores(), 'visible', os.cpu_count())
for fixed in (True, False):
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        s, thr, xh, names, dbg = m.encode_block_range(ctx, blocks, R, False, ('d1_mse',), (np.inf,), fixed, False)
        torch.cuda.synchronize(); t1 = time.perf_counter()
        thr_l = list(zip(*thr)); xh_l = list(zip(*xh))
        placed = departition_octree(xh_l[0], binstr, [0, 0, 0], [R] * 3, level); cloud = np.vstack(placed)
        t2 = time.perf_counter()
        ta = cKDTree(pts); t3 = time.perf_counter()
        tb = cKDTree(cloud, balanced_tree=False); t4 = time.perf_counter()
        i1 = pc_metric.nearest(tb, pts); t5 = time.perf_counter()
        i2 = pc_metric.nearest(ta, cloud); t6 = time.perf_counter()
        md = MT.select_best_per_opt_metric(binstr, xh_l, level, names, pts, R, False); t7 = time.perf_counter()
        print(f'fixed={fixed} rep{rep}: encode_block_range {t1-t0:.3f}  departition+vstack {t2-t1:.3f}  tree(A, balanced) {t3-t2:.3f}  tree(B) {t4-t3:.3f}  A->B {t5-t4:.3f}  B->A {t6-t5:.3f}  '
              f'select_best total {t7-t6:.3f}  decoded points {len(cloud)}')
