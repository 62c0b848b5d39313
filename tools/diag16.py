import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType
import _stagecheck as SC
from test_codec_gpu import make_blocks, scaled_weights
from oracle import torch_oracle as T, oracle as O
ctx = ops.Context(0)
res = int(sys.argv[1]) if len(sys.argv) > 1 else 64
model = ModelConfigType['c3p'].build(batch_size=2, precision='fp16')
model.compress([1, 1, res, res, res]); model.set_weights(scaled_weights(model, 2.2))
blocks = make_blocks(1, res, seed=2)
om = SC.oracle_model(model, 'c3p')
x = model._voxelize(ctx, blocks, (res,) * 3)
enc = model._encode_batch(model._ctx(ctx), x, debug=True); strings = enc['finish'](); torch.cuda.synchronize()
g = enc['debug'][0]
P = om['params']
for name, prefix, inp, out in [('AnalysisTransformProgressiveV2','analysis', x[0:1].cpu().numpy()[...,None], g['y']), ('HyperAnalysisTransform','hyper_analysis', g['y'], g['z']),
                               ('HyperSynthesisTransform','hyper_synthesis', g['z_hat'], g['sigma_hat']), ('SynthesisTransformProgressiveV2','synthesis', g['y_hat'], g['x_hat'])]:
    a16 = T.run_transform_fp16(name, 64, P, prefix, inp); a32 = T.run_transform(name, 64, P, prefix, inp)
    m = np.abs(a32).max()
    print(f'{prefix:16s} max {m:9.3f}  |gpu-fp16or| {np.abs(out-a16).max()/ (1+m):.2e}  |gpu-fp32or| {np.abs(out-a32).max()/(1+m):.2e}  |fp16or-fp32or| {np.abs(a16-a32).max()/(1+m):.2e}')
# partial synthesis: layer by layer using ops with the GPU? use per-layer forward of the model's transform in fp16 mode
