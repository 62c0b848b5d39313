"""GPU box: time a stride-2 transposed layer in the fp16 mode with fp16 hand-over (conv_tr2m_f16.hip vs the tiled conv_tr2g_kernel<F16>):
python tools/bench_tr2_f16.py B D cin cout   (PCC_NO_TR2M=1: the tiled kernel)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops
B, D, cin, cout = [int(v) for v in sys.argv[1:5]]
ctx = ops.Context(0)
rng = np.random.default_rng(0)
layer = ops.ConvLayer((rng.standard_normal((3, 3, 3, cout, cin)) / np.sqrt(27 * cin)).astype(np.float32), rng.standard_normal(cout).astype(np.float32), 2, True, True)
x = torch.randn((B, D, D, D, cin), device=ctx.device)
ops.conv3d_fp16_storage(ctx, x, layer, None, in16=False, out16=True)
torch.cuda.synchronize()
ts = []
for rep in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.conv3d_fp16_storage(ctx, x, layer, None, in16=False, out16=True)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
byt = B * D ** 3 * cin * 4 + B * (2 * D) ** 3 * cout * 2
print(f'fp16 mode B={B} D={D} {cin}->{cout} k3 s2 transposed: min {min(ts)*1000:.1f} us median {sorted(ts)[2]*1000:.1f} us  {byt/min(ts)/1e9:.2f} TB/s of algorithmic bytes (numerics switches 0x{ctx.numerics()[1]:x})')
