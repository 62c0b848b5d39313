"""Per-layer timing of the c3p conv stack on one MI355X (HIP events on the launch stream)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
IMPL = int(os.environ.get("PCC_BENCH_IMPL", L.PCC_IMPL_AUTO))
ctx = ops.Context(0)
rng = np.random.default_rng(0)
# (name, D, cin, cout, k, s, tr)
layers = [('A 1->16 s2', 64, 1, 16, 3, 2, 0), ('A 16->16', 32, 16, 16, 3, 1, 0), ('A 16->32 s2', 32, 16, 32, 3, 2, 0),
          ('A 32->32', 16, 32, 32, 3, 1, 0), ('A 32->64 s2', 16, 32, 64, 3, 2, 0), ('A 64->64', 8, 64, 64, 3, 1, 0),
          ('HA 64->64 s2', 8, 64, 64, 3, 2, 0), ('H 64->64 @4', 4, 64, 64, 3, 1, 0), ('HS T2 64->64', 4, 64, 64, 3, 2, 1),
          ('S T2 64->64 8->16', 8, 64, 64, 3, 2, 1), ('S T 64->64 @16', 16, 64, 64, 3, 1, 1),
          ('S T2 64->32 16->32', 16, 64, 32, 3, 2, 1), ('S T 32->32 @32', 32, 32, 32, 3, 1, 1),
          ('S T2 32->16 32->64', 32, 32, 16, 3, 2, 1), ('S T 16->16 @64', 64, 16, 16, 3, 1, 1),
          ('S T 16->1 @64', 64, 16, 1, 3, 1, 1)]
tot = 0
for name, D, cin, cout, k, s, tr in layers:
    wshape = (k, k, k, cout, cin) if tr else (k, k, k, cin, cout)
    w = (rng.standard_normal(wshape) / np.sqrt(k ** 3 * cin)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(cout).astype(np.float32), s, bool(tr), True)
    x = torch.randn((B, D, D, D, cin), device=ctx.device)
    out = ops.conv3d(ctx, x, layer, impl=IMPL)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 10
    e0.record()
    for _ in range(n):
        ops.conv3d(ctx, x, layer, impl=IMPL, out=out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    if tr and s == 2:
        macs = B * D ** 3 * k ** 3 * cin * cout
    elif tr:
        macs = B * D ** 3 * k ** 3 * cin * cout
    else:
        macs = B * (D // s) ** 3 * k ** 3 * cin * cout
    tf = 2 * macs / ms / 1e9
    print(f'{name:24s} B={B} {ms*1000:9.1f} us  {tf:7.1f} TFLOP/s  ({100*tf/157.3:5.1f}% of fp32 MFMA peak)')
