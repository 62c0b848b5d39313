"""Live HIP-event times of one synthesis layer inside the streaming pipeline (encoder and decoder instances alternate)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops, _lib as L
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
dev = torch.device('cuda', 0); ctx = ops.get_context(dev)
model = ModelConfigType['c3p'].build(batch_size=32); model.compress([1, 1, 64, 64, 64])
model.set_weights(bench.synthetic_weights(model))
x = bench.synthetic_blocks(32, dev, 0)
def run(steps):
    for _ in model.roundtrip_stream(ctx, (x for _ in range(steps))): pass
run(3); torch.cuda.synchronize()
for layer in [int(v) for v in sys.argv[1:]] or [1]:
    ops.profile_select(ctx, L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2, layer)
    run(12); torch.cuda.synchronize()
    t = np.array(ops.profile_read(ctx)) * 1e3
    print(f'layer {layer}: n={len(t)} even-idx mean {t[0::2].mean():.1f} us odd-idx mean {t[1::2].mean():.1f} us  min {t.min():.1f} max {t.max():.1f}', np.round(t[:12], 1))
