"""Fixed cost of a launch of the 16-channel Winograd kernel: 256 workgroups (16 blocks x 4 x 4 tiles), z extent varied."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(0)
layer = ops.ConvLayer((rng.standard_normal((3, 3, 3, 16, 16)) / 20).astype(np.float32), np.zeros(16, np.float32), 1, True, True)
for B in (16, 32):
    for D in (8, 16, 32, 64):
        x = torch.randn((B, D, 64, 64, 16), device=ctx.device)
        out = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_AUTO)
        ts = []
        for _ in range(6):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_AUTO, out=out)
            e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 100)
        print(f'B={B} D={D}: min {min(ts):.1f} us  median {sorted(ts)[3]:.1f} us')
