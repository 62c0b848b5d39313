"""Time one conv layer shape: python tools/bench_one.py B D cin cout k s tr [res]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
B, D, cin, cout, k, s, tr = [int(v) for v in sys.argv[1:8]]
IMPL = int(os.environ.get("PCC_BENCH_IMPL", L.PCC_IMPL_MFMA))
ctx = ops.Context(0)
rng = np.random.default_rng(0)
wshape = (k, k, k, cout, cin) if tr else (k, k, k, cin, cout)
layer = ops.ConvLayer((rng.standard_normal(wshape) / np.sqrt(k ** 3 * cin)).astype(np.float32), rng.standard_normal(cout).astype(np.float32), s, bool(tr), True)
x = torch.randn((B, D, D, D, cin), device=ctx.device)
res = torch.randn(ops.conv_out_shape(layer, x.shape), device=ctx.device) if len(sys.argv) > 8 else None
FL = int(os.environ.get('PCC_BENCH_FLAGS', '0'), 0)
out = ops.conv3d(ctx, x, layer, residual=res, impl=IMPL, flags=FL)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3d(ctx, x, layer, residual=res, impl=IMPL, out=out, flags=FL)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
ms = min(ts)
od = D * s if tr else D // s
macs = B * (D ** 3 if tr else od ** 3) * k ** 3 * cin * cout
print(f'impl={IMPL} B={B} D={D} {cin}->{cout} k{k} s{s} tr{tr}: min {ms*1000:.1f} us median {sorted(ts)[2]*1000:.1f} us  {2*macs/ms/1e9:.1f} TFLOP/s ({100*2*macs/ms/1e9/157.3:.1f}%)')
