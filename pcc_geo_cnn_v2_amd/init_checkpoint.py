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


# ---------------------------------------------------------------------------------------------------------------------
# Designed (not trained) c3p weight sets: rate points for RD plumbing without checkpoints (BASELINE.json configs[3])
# ---------------------------------------------------------------------------------------------------------------------
CELL_BITS = [(0, 2), (1, 2), (2, 2), (0, 1), (1, 1), (2, 1)]      # refinement order: (axis, bit) of the voxel coordinate in its 8^3 cell


def cell_shape(level):
    """Edge lengths (x, y, z) of the cells of rate point `level` (0..6): 8^3, 4x8x8, 4x4x8, 4^3, 2x4x4, 2x2x4, 2^3."""
    size = [8, 8, 8]
    for a, _ in CELL_BITS[:level]:
        size[a] //= 2
    return tuple(size)


def make_cell_codec_weights(level, sigma=None, seed=42):
    """A c3p weight set whose codec is known in closed form: the latent y holds, per 8^3 cell of the block, the NUMBER OF
    OCCUPIED VOXELS of each sub-cell of size cell_shape(level) (2^level of the 64 channels are used), coded with a constant
    Gaussian scale; the decoder switches on every voxel of every non-empty sub-cell (x_hat = 0.6 * count > thresholds[128]).
    Finer cells cost more bits and hug the surface more tightly: 7 rate points with a monotone rate-distortion behaviour,
    stand-ins for the lambda sweep of /root/reference/src/ev_experiment.yml:25-29 (trained checkpoints are not available).
    All weights are 0 / 1 / 0.6, so every layer is exact in fp32: encoder, decoder and oracle agree bit for bit and the decoded
    point set can be predicted with numpy (tests/test_rd_sweep_gpu.py).
    Layers: AnalysisBlock / SynthesisBlock keep only their strided conv (the two residual-branch convs are zero, so the block
    returns tensor1 + ReLU(0)); strided convs sum or separate the two children per axis (SAME padding puts taps 0, 1 on the
    children, tap 2 on the neighbour cell: unused)."""
    assert 0 <= level <= 6
    from .entropy_models import EntropyBottleneck, GaussianConditional, scale_table
    F = 64
    kept = CELL_BITS[:level]
    ax2 = [a for a, b in kept if b == 2]            # axes whose bit 2 (the 4-voxel half of the 8-cell) is resolved
    ax1 = [a for a, b in kept if b == 1]            # axes whose bit 1 is resolved
    n1 = len(ax1)

    def idx(child, axes):                           # channel index of a child (cx, cy, cz) over the resolved axes
        return sum(child[a] << j for j, a in enumerate(axes))

    w = {}
    zeros = lambda *sh: np.zeros(sh, np.float32)
    children = [(cx, cy, cz) for cx in (0, 1) for cy in (0, 1) for cz in (0, 1)]
    # ---- analysis: blocks f = 16, 32, 64 (convs 0-2, 3-5, 6-8) + conv 9.  Forward kernels (kd,kh,kw,Cin,Cout).
    a_shapes = [(1, 16), (16, 16), (16, 16), (16, 32), (32, 32), (32, 32), (32, 64), (64, 64), (64, 64), (64, 64)]
    for i, (ci, co) in enumerate(a_shapes):
        w[f'analysis/{i}/kernel'] = zeros(3, 3, 3, ci, co)
        if i < 9:
            w[f'analysis/{i}/bias'] = zeros(co)
    for c in children:
        w['analysis/0/kernel'][c[0], c[1], c[2], 0, 0] = 1                          # bit 0: always summed (2^3 counts)
        w['analysis/3/kernel'][c[0], c[1], c[2], 0, idx(c, ax1)] = 1                # bit 1
        for i1 in range(1 << n1):
            w['analysis/6/kernel'][c[0], c[1], c[2], i1, (idx(c, ax2) << n1) + i1] = 1   # bit 2
    w['analysis/9/kernel'][1, 1, 1] = np.eye(64, dtype=np.float32)                  # identity (centre tap of SAME k3 s1)
    # ---- synthesis: blocks f = 64, 32, 16 + ConvT 16 -> 1.  Transposed kernels (kd,kh,kw,Cout,Cin).
    s_shapes = [(64, 64), (64, 64), (64, 64), (32, 64), (32, 32), (32, 32), (16, 32), (16, 16), (16, 16), (1, 16)]
    for i, (co, ci) in enumerate(s_shapes):
        w[f'synthesis/{i}/kernel'] = zeros(3, 3, 3, co, ci)
        w[f'synthesis/{i}/bias'] = zeros(co)
    for c in children:
        for i1 in range(1 << n1):
            w['synthesis/0/kernel'][c[0], c[1], c[2], i1, (idx(c, ax2) << n1) + i1] = 1
        w['synthesis/3/kernel'][c[0], c[1], c[2], 0, idx(c, ax1)] = 1
        w['synthesis/6/kernel'][c[0], c[1], c[2], 0, 0] = 1
    w['synthesis/9/kernel'][1, 1, 1, 0, 0] = 0.6
    # ---- hyperprior: z = 0, sigma_hat = one constant for the 2^level used channels
    for prefix, tr, nb in (('hyper_analysis', False, (True, True, False)), ('hyper_synthesis', True, (True, True, True))):
        for i in range(3):
            w[f'{prefix}/{i}/kernel'] = zeros(3, 3, 3, F, F)
            if nb[i]:
                w[f'{prefix}/{i}/bias'] = zeros(F)
    if sigma is None:
        cx, cy, cz = cell_shape(level)
        sigma = max(1.0, cx * cy * cz / 16.0)       # counts range up to the cell volume
    w['hyper_synthesis/2/bias'][:1 << level] = np.float32(sigma)    # the unused channels keep sigma_hat = 0 -> table[0]: ~0 bits
    eb = EntropyBottleneck(F, params=EntropyBottleneck.init_params(F, init_scale=0.2, seed=seed))
    for k, v in eb.params.items():
        w[f'entropy_bottleneck/{k}'] = v
    w.update({'entropy_bottleneck/quantized_cdf': eb.quantized_cdf, 'entropy_bottleneck/cdf_length': eb.cdf_length,
              'entropy_bottleneck/offset': eb.offset})
    gc = GaussianConditional(scale_table())
    w.update({'gaussian_conditional/quantized_cdf': gc.quantized_cdf, 'gaussian_conditional/cdf_length': gc.cdf_length,
              'gaussian_conditional/offset': gc.offset})
    return w


def cell_codec_expected_points(points, level, block=64):
    """The point set the cell codec of `level` must decode for an input cloud of integer points: every voxel of every
    non-empty cell.  Returns an (n, 3) int64 array sorted lexicographically."""
    size = np.array(cell_shape(level), np.int64)
    cells = np.unique(np.asarray(points)[:, :3].astype(np.int64) // size, axis=0)
    off = np.stack(np.meshgrid(*[np.arange(s) for s in size], indexing='ij'), -1).reshape(-1, 3)
    out = (cells[:, None, :] * size + off[None]).reshape(-1, 3)
    return out[np.lexsort((out[:, 2], out[:, 1], out[:, 0]))]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--model_config', required=True)
    ap.add_argument('--checkpoint_dir', required=True)
    ap.add_argument('--seed', type=int, default=42)
    ap.add_argument('--gain_analysis', type=float, default=1.35)
    ap.add_argument('--gain_synthesis', type=float, default=1.8)
    ap.add_argument('--final_bias', type=float, default=0.0)
    ap.add_argument('--cell_level', type=int, default=None,
                    help='c3p only: the designed "occupied cell" codec of make_cell_codec_weights, rate point 0 (8^3 cells) .. 6 (2^3 cells)')
    a = ap.parse_args()
    if a.cell_level is not None:
        assert a.model_config == 'c3p', '--cell_level is defined for the c3p graph'
        w = make_cell_codec_weights(a.cell_level, seed=a.seed)
    else:
        w = make_synthetic_weights(a.model_config, a.seed, a.gain_analysis, a.gain_synthesis, a.final_bias)
    os.makedirs(a.checkpoint_dir, exist_ok=True)
    np.savez(os.path.join(a.checkpoint_dir, 'model.npz'), **w)
    print(f'wrote {len(w)} arrays to {a.checkpoint_dir}/model.npz')


if __name__ == '__main__':
    sys.exit(main())
