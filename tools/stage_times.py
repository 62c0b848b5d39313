"""CPU wall-clock per pipeline stage of roundtrip_stream (no extra syncs)."""
import sys, os, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
dev = torch.device('cuda', 0); ctx = ops.get_context(dev)
model = ModelConfigType['c3p'].build(batch_size=32); model.compress([1, 1, 64, 64, 64])
model.set_weights(bench.synthetic_weights(model))
x = bench.synthetic_blocks(32, dev, 0)
T = collections.defaultdict(float)
def timed(name, fn):
    def w(*a, **k):
        t = time.perf_counter(); r = fn(*a, **k); T[name] += time.perf_counter() - t; return r
    return w
model._encode_batch = timed('enc_enqueue', model._encode_batch)
model._decode_phase_a = timed('dec_phase_a(zdec+enqueue HS)', model._decode_phase_a)
model._decode_phase_b = timed('dec_phase_b(wait idx+ydec+enqueue S)', model._decode_phase_b)
model._extract_points = timed('extract_points enqueue', model._extract_points)
model._gather_points = timed('gather_points(sync)', model._gather_points)
orig_re, orig_rd = ops.range_encode_batch, ops.range_decode_batch
ops.range_encode_batch = timed('  range_encode_batch', orig_re)
ops.range_decode_batch = timed('  range_decode_batch', orig_rd)
import pcc_geo_cnn_v2_amd.model_types as MTY
def run(steps):
    for _ in model.roundtrip_stream(ctx, (x for _ in range(steps))): pass
run(3); torch.cuda.synchronize(); T.clear()
steps = 10
t0 = time.perf_counter(); run(steps); torch.cuda.synchronize(); el = time.perf_counter() - t0
print(f'{1e3*el/steps:.2f} ms/step')
for k, v in sorted(T.items(), key=lambda kv: -kv[1]): print(f'  {1e3*v/steps:7.3f} ms/step  {k}')
# finish() closure time = total - others: measure separately
