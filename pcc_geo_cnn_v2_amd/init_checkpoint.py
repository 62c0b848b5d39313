"""Writes a checkpoint directory (`model.npz`) with seeded synthetic weights for a model config -- the
trained checkpoints of the reference are not available offline (README.md:20-25).

  python -m pcc_geo_cnn_v2_amd.init_checkpoint --model_config c3p --checkpoint_dir /tmp/ckpt [--seed 42]
"""
import argparse
import os
import sys

import numpy as np


def make_synthetic_weights(model_config, seed=42, gain_analysis=1.35, gain_synthesis=1.8, final_bias=0.0,
                           eb_init_scale=0.2):
    """Host-only (no GPU needed): builds the transforms to learn the layer shapes and draws the weights."""
    from . import model_transforms as MT
    from .entropy_models import EntropyBottleneck, GaussianConditional, scale_table
    from .model_configs import ModelConfigType
    cfg = ModelConfigType[model_config].value
    p = cfg.model_params
    F = p['num_filters']
    rng = np.random.default_rng(seed)
    w = {}
    names = [('analysis', p['analysis_transform_type'], 1)]
    if 'hyper_analysis_transform_type' in p:
        names += [('hyper_analysis', p['hyper_analysis_transform_type'], F),
                  ('hyper_synthesis', p['hyper_synthesis_transform_type'], F)]
    names.append(('synthesis', p['synthesis_transform_type'], F))
    brng = np.random.default_rng(seed + 1)
    for prefix, ttype, cin in names:
        tr = MT.init_transform(ttype.value(F, data_format='channels_first'), cin, rng)
        for k, v in MT.get_weights(tr, prefix).items():
            if k.endswith('/kernel'):
                v = (v * (gain_analysis if prefix != 'synthesis' else gain_synthesis)).astype(np.float32)
            else:
                v = brng.normal(0, 0.05, v.shape).astype(np.float32)
            w[k] = v
    last = max(int(k.split('/')[1]) for k in w if k.startswith('synthesis/'))
    w[f'synthesis/{last}/bias'] = np.array([final_bias], np.float32)
    eb = EntropyBottleneck(F, params=EntropyBottleneck.init_params(F, init_scale=eb_init_scale, seed=seed))
    for k, v in eb.params.items():
        w[f'entropy_bottleneck/{k}'] = v
    w.update({'entropy_bottleneck/quantized_cdf': eb.quantized_cdf, 'entropy_bottleneck/cdf_length': eb.cdf_length,
              'entropy_bottleneck/offset': eb.offset})
    if 'hyper_analysis_transform_type' in p:
        gc = GaussianConditional(scale_table())
        w.update({'gaussian_conditional/quantized_cdf': gc.quantized_cdf, 'gaussian_conditional/cdf_length': gc.cdf_length,
                  'gaussian_conditional/offset': gc.offset})
    return w


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model_config', required=True)
    ap.add_argument('--checkpoint_dir', required=True)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--gain_analysis', type=float, default=1.35)
    ap.add_argument('--gain_synthesis', type=float, default=1.8)
    ap.add_argument('--final_bias', type=float, default=0.0)
    a = ap.parse_args()
    w = make_synthetic_weights(a.model_config, a.seed, a.gain_analysis, a.gain_synthesis, a.final_bias)
    os.makedirs(a.checkpoint_dir, exist_ok=True)
    np.savez(os.path.join(a.checkpoint_dir, 'model.npz'), **w)
    print(f'wrote {len(w)} arrays to {a.checkpoint_dir}/model.npz')


if __name__ == '__main__':
    sys.exit(main())
