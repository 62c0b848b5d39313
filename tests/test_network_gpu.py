"""GPU: the batched-graph entry points of the C ABI (pcc_network_forward_*, pcc_codec_*; include/pcc_geo.h) are
bit-identical to the same layers issued one by one through pcc_conv3d, and one encode+decode step needs <= 8 ABI calls."""
import os

import numpy as np
import pytest
import torch

from pcc_geo_cnn_v2_amd import _lib as L
from pcc_geo_cnn_v2_amd import model_transforms as MT
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_configs import ModelConfigType

pytestmark = pytest.mark.gpu

CASES = [('AnalysisTransformV1', 32, 1, 32), ('SynthesisTransformV1', 32, 32, 4), ('AnalysisTransformV2', 32, 1, 32),
         ('SynthesisTransformV2', 32, 32, 4), ('AnalysisTransformProgressiveV2', 64, 1, 64),
         ('SynthesisTransformProgressiveV2', 64, 64, 8), ('HyperAnalysisTransform', 64, 64, 8), ('HyperSynthesisTransform', 64, 64, 4)]


@pytest.mark.parametrize('name,F,cin,res', CASES)
def test_network_forward_is_bit_identical_to_the_layerwise_path(ctx, monkeypatch, name, F, cin, res):
    """/root/reference/src/model_transforms.py:41-158: one Keras layer(tensor) call per transform."""
    tr = MT.TransformType[name].value(F, data_format='channels_last')
    MT.init_transform(tr, cin, np.random.default_rng(3))
    for c in tr.conv_layers():          # non-zero biases
        if c.use_bias:
            c.set_weights(c.layer.kernel, np.random.default_rng(c.filters).normal(0, 0.1, c.filters).astype(np.float32))
    x = torch.randn((3, res, res, res, cin), generator=torch.Generator().manual_seed(1)).to(ctx.device)
    assert tr.network() is not None
    a = tr.forward_ndhwc(ctx, x)
    a2 = tr.forward_ndhwc(ctx, x)
    monkeypatch.setenv('PCC_LAYERWISE', '1')
    assert tr.network() is None
    b = tr.forward_ndhwc(ctx, x)
    torch.cuda.synchronize()
    assert a.shape == b.shape and torch.equal(a, b) and torch.equal(a, a2)
    if name.startswith('Synthesis'):    # encoder flavour: clip fused into the last layer
        monkeypatch.delenv('PCC_LAYERWISE')
        c = tr.forward_ndhwc(ctx, x, final_flags=L.PCC_CONV_CLIP01)
        assert torch.equal(c, a.clamp(0, 1))


@pytest.mark.parametrize('name,F,cin,res', [('AnalysisTransformProgressiveV2', 64, 1, 128), ('SynthesisTransformProgressiveV2', 64, 64, 16),
                                            ('AnalysisTransformProgressiveV2', 64, 1, 64), ('SynthesisTransformProgressiveV2', 64, 64, 8),
                                            ('AnalysisTransformV2', 32, 1, 64), ('SynthesisTransformV2', 32, 32, 8)])
def test_fp16_mode_network_matches_the_fp16_restatement(ctx, name, F, cin, res):
    """fp16 mode (PCC_CONV_F16, BASELINE.json configs[4]): inside the residual blocks of BOTH transform families whose grids are
    multiples of 16 the mid-block tensors are fp16 in HBM (csrc/network.hip plans it: OUT16 on the stride-2 layer, conv_f16.hip
    on the two k3 stride-1 layers).  The whole stack is compared with the CPU restatement of exactly that graph
    (oracle/torch_oracle.run_transform_fp16: the same fp16 roundings, fp32 accumulation) within 2e-3 * (1 + max|ref|) -- what is
    left are one-ulp flips of stored fp16 intermediates, see tests/test_codec_gpu.py -- and is bit-deterministic."""
    from oracle import torch_oracle as T
    tr = MT.TransformType[name].value(F, data_format='channels_last')
    MT.init_transform(tr, cin, np.random.default_rng(5))
    g = torch.Generator().manual_seed(2)
    x = (torch.rand((2, res, res, res, cin), generator=g) < 0.1).float() if cin == 1 else torch.randn((2, res, res, res, cin), generator=g)
    x = x.to(ctx.device)
    ref32 = tr.forward_ndhwc(ctx, x)
    v = ctx.view(L.PCC_CONV_F16)
    a = tr.forward_ndhwc(v, x)
    b = tr.forward_ndhwc(v, x)
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert not torch.equal(a, ref32)          # the mode really is a different arithmetic
    ref16 = T.run_transform_fp16(name, F, MT.get_weights(tr, 't'), 't', x[:1].cpu().numpy())
    err = np.abs(a[:1].cpu().numpy() - ref16).max()
    assert err <= 2e-3 * (1 + np.abs(ref16).max()), err


def test_family_entry_points_reject_other_families(ctx):
    tr = MT.TransformType['HyperAnalysisTransform'].value(64, data_format='channels_last')
    MT.init_transform(tr, 64, np.random.default_rng(0))
    net = tr.network()
    x = torch.zeros((1, 8, 8, 8, 64), device=ctx.device)
    y = torch.empty((1, 4, 4, 4, 64), device=ctx.device)
    ws = ctx.workspace(L.lib().pcc_network_workspace_bytes(net.transform, 64, 1, 8, 8, 8))
    args = (ctx.handle, net.transform, 64, net.blob(ctx).data_ptr(), x.data_ptr(), 1, 8, 8, 8, y.data_ptr(), ws.data_ptr(), ws.numel(), 0, 0, None)
    assert L.lib().pcc_network_forward_hyper_a(*args) == 0
    assert L.lib().pcc_network_forward_synthesis(*args) == -1 and b'synthesis' in L.lib().pcc_last_error()
    small = (ctx.handle, net.transform, 64, net.blob(ctx).data_ptr(), x.data_ptr(), 1, 8, 8, 8, y.data_ptr(), ws.data_ptr(), 16, 0, 0, None)
    assert L.lib().pcc_network_forward(*small) == -1 and b'workspace' in L.lib().pcc_last_error()


@pytest.mark.parametrize('cfg,res', [('c1', 32), ('c3p', 32), ('c3p', 64)])
def test_codec_phase_calls_match_the_layerwise_graph(ctx, monkeypatch, cfg, res):
    """pcc_codec_encode / _decode_hyper / _decode_main (src/model_types.py:283-309, 371-411) == the per-layer graph, tensor by
    tensor and byte by byte."""
    from test_codec_gpu import make_blocks, scaled_weights
    m = ModelConfigType[cfg].build(batch_size=4)
    m.compress([1, 1, res, res, res])
    m.set_weights(scaled_weights(m, 2.2))
    x = m._voxelize(ctx, make_blocks(4, res, seed=2), (res,) * 3)
    thr = m._thr_tensor(ctx, [128] * 4)

    def run():
        enc = m._encode_batch(ctx, x, debug=True, thr=thr)
        strings = enc['finish']()
        st = m._decode_phase_a(ctx, strings, (res,) * 3)
        dec = m._decode_phase_b(ctx, st, (res,) * 3, True, thr=thr)
        torch.cuda.synchronize()
        return enc, strings, dec

    assert m._codec(ctx) is not None
    e1, s1, d1 = run()
    monkeypatch.setenv('PCC_LAYERWISE', '1')
    assert m._codec(ctx) is None
    e2, s2, d2 = run()
    assert s1 == s2
    for b in range(4):
        for k in e2['debug'][b]:
            assert np.array_equal(e1['debug'][b][k], e2['debug'][b][k]), k
        for k in d2['debug'][b]:
            assert np.array_equal(d1['debug'][b][k], d2['debug'][b][k]), k
        assert np.array_equal(e1['debug'][b]['x_hat'], d1['debug'][b]['x_hat'])      # encoder == decoder, bit for bit
    assert torch.equal(e1['counts'], e2['counts']) and torch.equal(d1['counts'], d2['counts'])
    for b in range(4):
        n = int(e1['counts'][b])
        assert n > 0 and torch.equal(e1['xyz'][b, :n], e2['xyz'][b, :n]) and torch.equal(d1['xyz'][b, :n], d2['xyz'][b, :n])


def test_one_encode_decode_step_is_at_most_eight_abi_calls(ctx):
    """the hot path crosses the C ABI once per graph phase, not once per layer."""
    m = ModelConfigType['c3p'].build(batch_size=4)
    m.compress([1, 1, 64, 64, 64])
    x = (torch.rand((4, 64, 64, 64), generator=torch.Generator().manual_seed(0)) < 0.03).float().to(ctx.device)
    list(m.roundtrip_stream(ctx, [x]))          # warm-up: weight upload, pinned buffers
    lib, calls = L.lib(), []
    hot = [n for n in L.EXPORTS if n.startswith(('pcc_conv3d', 'pcc_quantize', 'pcc_dequantize', 'pcc_scale_to_index', 'pcc_threshold_compact',
                                                 'pcc_voxelize', 'pcc_range_', 'pcc_network_forward', 'pcc_codec_encode', 'pcc_codec_decode',
                                                 'pcc_weights_'))]
    orig = {n: getattr(lib, n) for n in hot}
    try:
        for n in hot:
            def make(n=n):
                f = orig[n]
                def w(*a):
                    calls.append(n)
                    return f(*a)
                return w
            setattr(lib, n, make())
        out = list(m.roundtrip_stream(ctx, [x]))
    finally:
        for n in hot:
            setattr(lib, n, orig[n])
    assert len(out) == 1 and len(out[0][0]) == 4
    # (the range coder is entered through its narrow-array entry points: int16 symbols / uint8 rows come off PCIe)
    sfx = '' if os.environ.get('PCC_WIDE_SYMBOLS') else '_n'       # (A/B switch: int32 symbols take the 32-bit coder entry points)
    assert sorted(calls) == sorted(['pcc_codec_encode', 'pcc_range_encode_batch' + sfx, 'pcc_range_encode_batch' + sfx, 'pcc_range_decode_batch' + sfx,
                                    'pcc_codec_decode_hyper', 'pcc_range_decode_batch' + sfx, 'pcc_codec_decode_main']), calls
    assert len(calls) <= 8


def test_live_profile_events(ctx):
    m = ModelConfigType['c3p'].build(batch_size=2)
    m.compress([1, 1, 32, 32, 32])
    x = (torch.rand((2, 32, 32, 32), generator=torch.Generator().manual_seed(0)) < 0.03).float().to(ctx.device)
    ops.profile_select(ctx, L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2, 8)
    list(m.roundtrip_stream(ctx, [x, x, x]))
    ms = ops.profile_read(ctx)
    ops.profile_select(ctx, -1, -1)
    assert len(ms) == 6 and all(0 < v < 50 for v in ms)          # encoder + decoder synthesis of three chunks
    assert ops.profile_read(ctx) == []
