# adaptive d1 search of one 190-block cloud: wall clock of encode_block_range and the GPU time of its kernels (kernel trace)
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/r06srch; rm -rf $OUT; mkdir -p $OUT; export TMPDIR=/tmp; cd /tmp
cat > $OUT/run.py <<'PY'
import os, sys, time
sys.path.insert(0, os.environ['GRAFT_REPO_ROOT'])
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
from pcc_geo_cnn_v2_amd.utils.octree_coding import partition_octree
ctx = ops.get_context(torch.device('cuda', 0))
R, level, res = 1024, 4, 64
pts = bench.standin_cloud()
blocks, binstr = partition_octree(pts, [0, 0, 0], [R] * 3, level)
m = ModelConfigType['c3p'].build(batch_size=32); m.compress([1, 1, res, res, res]); m.set_weights(bench.synthetic_weights(m))
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.encode_block_range(ctx, blocks, R, False, ('d1_mse',), (np.inf,), False, False)
    torch.cuda.synchronize(); print('adaptive d1 encode_block_range', round(time.perf_counter() - t0, 3), 's', len(blocks), 'blocks')
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    m.encode_block_range(ctx, blocks, R, False, ('d1_mse',), (np.inf,), True, False)
    torch.cuda.synchronize(); print('fixed threshold encode_block_range', round(time.perf_counter() - t0, 3), 's')
PY
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -o t -- python $OUT/run.py 2>&1 | grep "encode_block_range"
f=$(find $OUT/t -name "t_kernel_stats.csv" | head -1)
head -24 $f | cut -c1-110,150-230
rm -rf $OUT/t
