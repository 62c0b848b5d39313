"""Time the fp16-storage k3 s1 layer: python tools/bench_f16.py C N D [res] [o32]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops
C, N, D = [int(v) for v in sys.argv[1:4]]
res, o32 = 'res' in sys.argv, 'o32' in sys.argv
ctx = ops.Context(0)
rng = np.random.default_rng(0)
layer = ops.ConvLayer((rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32), rng.standard_normal(C).astype(np.float32), 1, True, True)
x = torch.randn((N, D, D, D, C), device=ctx.device).half()
r = torch.randn((N, D, D, D, C), device=ctx.device).half() if res else None
f = lambda: ops.conv3d_fp16_storage(ctx, x, layer, r, out16=not o32)
f(); torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): f()
    e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 10)
nb = N * D ** 3 * C * 2 * (3 if res else 2) + (N * D ** 3 * C * 2 if o32 else 0)
print(f'f16 C={C} N={N} D={D} res={res} o32={o32}: min {min(ts)*1e3:.1f} us median {sorted(ts)[2]*1e3:.1f} us  {nb/sorted(ts)[2]/1e9:.2f} TB/s algorithmic')
