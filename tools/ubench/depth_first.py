"""Layer-by-layer over the whole batch vs depth-first over half batches for the full-resolution tail of the c3p synthesis
(32 -> 16 stride-2 transposed, 16 -> 16, 16 -> 16 + residual, 16 -> 1 @64^3): with 32 blocks every activation is 537 MB, twice the
256 MB memory-side cache; with 16 blocks a producer's output is still cached when its consumer reads it."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(0)
mk = lambda cin, cout, s, relu=True: ops.ConvLayer((rng.standard_normal((3, 3, 3, cout, cin)) / np.sqrt(27 * cin)).astype(np.float32),
                                                   rng.standard_normal(cout).astype(np.float32), s, True, relu)
l_up, l_a, l_b, l_out = mk(32, 16, 2), mk(16, 16, 1), mk(16, 16, 1), mk(16, 1, 1)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
x = torch.randn((B, 32, 32, 32, 32), device=ctx.device)
t1 = torch.empty((B, 64, 64, 64, 16), device=ctx.device); t2 = torch.empty_like(t1); t3 = torch.empty_like(t1)
o = torch.empty((B, 64, 64, 64, 1), device=ctx.device)


def tail(lo, hi):
    ops.conv3d(ctx, x[lo:hi], l_up, impl=L.PCC_IMPL_AUTO, out=t1[lo:hi])
    ops.conv3d(ctx, t1[lo:hi], l_a, impl=L.PCC_IMPL_AUTO, out=t2[lo:hi])
    ops.conv3d(ctx, t2[lo:hi], l_b, residual=t1[lo:hi], impl=L.PCC_IMPL_AUTO, out=t3[lo:hi])
    ops.conv3d(ctx, t3[lo:hi], l_out, impl=L.PCC_IMPL_AUTO, out=o[lo:hi])


def run(parts):
    ts = []
    for _ in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for p in range(parts):
            tail(p * B // parts, (p + 1) * B // parts)
        e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[0], ts[len(ts) // 2]


tail(0, B); ref = o.clone()
for parts in (1, 2, 4, 1, 2, 4):
    mn, md = run(parts)
    print(f'{parts} part(s) of {B // parts} blocks: min {mn:.0f} us, median {md:.0f} us; same bits: {bool(torch.equal(o, ref))}')
