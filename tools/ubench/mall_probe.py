"""Does the memory-side cache (MALL / Infinity Cache, 256 MB) carry a producer's output to its consumer?  Times the 16 -> 1 last
layer (HBM-bound, 537 MB input) right after (a) a copy that just wrote its input, (b) a 1 GB fill that flushed the cache, and
(c) a copy that wrote only the second half of the input last."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(0)
layer = ops.ConvLayer((rng.standard_normal((3, 3, 3, 1, 16)) / 20).astype(np.float32), np.zeros(1, np.float32), 1, True, True)
x = torch.randn((32, 64, 64, 64, 16), device=ctx.device)
y = torch.randn_like(x)
junk = torch.empty((256 * 1024 * 1024,), dtype=torch.float32, device=ctx.device)
out = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_AUTO)


def timed(prep):
    ts = []
    for _ in range(12):
        prep()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_AUTO, out=out); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return f'min {ts[0]:.1f} median {ts[len(ts) // 2]:.1f} us'


print('input just written (whole tensor)        :', timed(lambda: x.copy_(y)))
print('cache flushed by a 1 GB fill             :', timed(lambda: junk.zero_()))
print('second half of the blocks written last   :', timed(lambda: (x[:16].copy_(y[:16]), x[16:].copy_(y[16:]))))
print('nothing in between (back to back)        :', timed(lambda: None))


# round 5: does the ORDER matter?  The consumer walks the blocks 0 .. 31 (workgroup index order).  If the producer wrote block 31 last, what
# the 256 MB cache still holds (the last-written half) is read LAST, after the consumer's own traffic has pushed it out; if the producer
# wrote block 0 last, the consumer starts on cached data.
def write_order(order, step=4):
    def prep():
        for n0 in order:
            x[n0:n0 + step].copy_(y[n0:n0 + step])
    return prep


print('written in blocks of 4, ascending  (0 .. 31) :', timed(write_order(range(0, 32, 4))))
print('written in blocks of 4, descending (31 .. 0) :', timed(write_order(range(28, -1, -4))))
print('written in blocks of 4, ascending  (0 .. 31) :', timed(write_order(range(0, 32, 4))))
print('written in blocks of 4, descending (31 .. 0) :', timed(write_order(range(28, -1, -4))))
