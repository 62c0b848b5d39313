"""GPU: element-wise / compaction / reduction kernels vs the oracle (bit-exact for integer results), and
the reference's transform shape tests (src/test_model_transforms.py) against the HIP-backed classes."""
import numpy as np
import pytest
import torch
from numpy.testing import assert_array_equal

from pcc_geo_cnn_v2_amd import _lib as L
from pcc_geo_cnn_v2_amd import ops
from pcc_geo_cnn_v2_amd.model_transforms import (AnalysisBlock, AnalysisTransformProgressiveV2, AnalysisTransformV1,
                                                 AnalysisTransformV2, HyperAnalysisTransform, HyperSynthesisTransform,
                                                 SynthesisBlock, SynthesisTransformProgressiveV2, SynthesisTransformV1,
                                                 SynthesisTransformV2)
from pcc_geo_cnn_v2_amd.utils.focal_loss import focal_loss

pytestmark = pytest.mark.gpu


class TestModelTransforms:
    """mirror of src/test_model_transforms.py:14-73 (zeros input, channels_last, shapes only)"""
    data_format = 'channels_last'

    def run_layer_test(self, ctx, layer, x):
        return layer(torch.from_numpy(x).to(ctx.device)).cpu().numpy()

    @pytest.fixture(autouse=True)
    def _inputs(self):
        self.x = np.zeros((1, 8, 8, 8, 1), np.float32)
        self.y = np.zeros((1, 1, 1, 1, 1), np.float32)

    def test_analysis_transform_v1(self, ctx):
        assert self.run_layer_test(ctx, AnalysisTransformV1(1, data_format=self.data_format), self.x).shape == (1, 1, 1, 1, 1)

    def test_synthesis_transform_v1(self, ctx):
        assert self.run_layer_test(ctx, SynthesisTransformV1(2, data_format=self.data_format), self.y).shape == (1, 8, 8, 8, 1)

    def test_analysis_block(self, ctx):
        assert self.run_layer_test(ctx, AnalysisBlock(1, data_format=self.data_format), self.x).shape == (1, 4, 4, 4, 1)
        x = self.run_layer_test(ctx, AnalysisBlock(1, data_format=self.data_format, residual_mode='concat'), self.x)
        assert x.shape == (1, 4, 4, 4, 2)

    def test_synthesis_block(self, ctx):
        assert self.run_layer_test(ctx, SynthesisBlock(1, data_format=self.data_format), self.y).shape == (1, 2, 2, 2, 1)
        y = self.run_layer_test(ctx, SynthesisBlock(1, data_format=self.data_format, residual_mode='concat'), self.y)
        assert y.shape == (1, 2, 2, 2, 2)

    def test_analysis_transform_v2(self, ctx):
        assert self.run_layer_test(ctx, AnalysisTransformV2(2, data_format=self.data_format), self.x).shape == (1, 1, 1, 1, 2)
        x = self.run_layer_test(ctx, AnalysisTransformV2(2, data_format=self.data_format, residual_mode='concat'), self.x)
        assert x.shape == (1, 1, 1, 1, 2)

    def test_synthesis_transform_v2(self, ctx):
        assert self.run_layer_test(ctx, SynthesisTransformV2(2, data_format=self.data_format), self.y).shape == (1, 8, 8, 8, 1)
        y = self.run_layer_test(ctx, SynthesisTransformV2(2, data_format=self.data_format, residual_mode='concat'), self.y)
        assert y.shape == (1, 8, 8, 8, 1)

    def test_analysis_transform_progressive_v2(self, ctx):
        assert self.run_layer_test(ctx, AnalysisTransformProgressiveV2(4, data_format=self.data_format), self.x).shape == (1, 1, 1, 1, 4)

    def test_synthesis_transform_progressive_v2(self, ctx):
        assert self.run_layer_test(ctx, SynthesisTransformProgressiveV2(4, data_format=self.data_format), self.y).shape == (1, 8, 8, 8, 1)

    def test_hyper_analysis_transform(self, ctx):
        assert self.run_layer_test(ctx, HyperAnalysisTransform(1, data_format=self.data_format), self.x).shape == (1, 4, 4, 4, 1)

    def test_hyper_synthesis_transform(self, ctx):
        assert self.run_layer_test(ctx, HyperSynthesisTransform(1, data_format=self.data_format), self.y).shape == (1, 2, 2, 2, 1)


def test_channels_first_equals_channels_last(ctx):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, 3, 8, 8, 8)).astype(np.float32)
    a = AnalysisBlock(4, data_format='channels_first')
    b = AnalysisBlock(4, data_format='channels_last')
    ya = a(torch.from_numpy(x).to(ctx.device))
    for la, lb in zip(a.conv_layers(), b.conv_layers()):
        lb.set_weights(la.layer.kernel, la.layer.bias)
    yb = b(torch.from_numpy(x.transpose(0, 2, 3, 4, 1).copy()).to(ctx.device))
    assert ya.shape == (2, 4, 4, 4, 4)
    assert torch.equal(ya.permute(0, 2, 3, 4, 1), yb)


def test_concat_residual_values(ctx, oracle):
    rng = np.random.default_rng(1)
    blk = AnalysisBlock(3, data_format='channels_last', residual_mode='concat')
    x = rng.standard_normal((1, 6, 6, 6, 2)).astype(np.float32)
    y = blk(torch.from_numpy(x).to(ctx.device)).cpu().numpy()
    ls = [c.layer for c in blk.conv_layers()]
    t1 = oracle.conv3d(x, ls[0].kernel, ls[0].bias, 2, True)
    t = oracle.conv3d(oracle.conv3d(t1, ls[1].kernel, ls[1].bias, 1, True), ls[2].kernel, ls[2].bias, 1, True)
    assert np.abs(y - np.concatenate([t, t1], -1)).max() < 1e-4     # tf.concat((tensor, tensor1)), model_transforms.py:38


@pytest.mark.parametrize('mode', [L.PCC_ROUND_FLOOR_HALF, L.PCC_ROUND_HALF_EVEN])
def test_quantize_bit_exact(ctx, oracle, mode):
    rng = np.random.default_rng(0)
    v = (rng.standard_normal((3, 4, 4, 4, 16)) * 3).astype(np.float32)
    v.ravel()[:64] = np.arange(-32, 32) * 0.5            # exact .5 boundaries
    v.ravel()[64] = 0.49999997
    med = rng.normal(0, 0.3, 16).astype(np.float32)
    for m in (None, med):
        sym, deq = ops.quantize(ctx, torch.from_numpy(v).to(ctx.device), None if m is None else torch.from_numpy(m).to(ctx.device), mode)
        osym, odeq = oracle.quantize(v, m, mode)
        assert_array_equal(sym.cpu().numpy(), osym)
        assert_array_equal(deq.cpu().numpy(), odeq)
        d2 = ops.dequantize(ctx, sym, None if m is None else torch.from_numpy(m).to(ctx.device))
        assert_array_equal(d2.cpu().numpy(), odeq)


def test_scale_to_index_bit_exact(ctx, oracle):
    tab = oracle.scale_table().astype(np.float32)
    rng = np.random.default_rng(1)
    s = np.exp(rng.uniform(np.log(0.01), np.log(600), 20000)).astype(np.float32)
    s[:64] = tab                                           # exact table values
    s[64:128] = np.nextafter(tab, np.float32(1e9))
    s[128] = 0.0
    idx = ops.scale_to_index(ctx, torch.from_numpy(s).to(ctx.device), torch.from_numpy(tab).to(ctx.device))
    assert_array_equal(idx.cpu().numpy(), oracle.scale_index(s, tab))


@pytest.mark.parametrize('shape', [(3, 16, 16, 16), (2, 64, 64, 64), (2, 5, 7, 3), (1, 128, 128, 128)])
def test_threshold_compact_bit_exact(ctx, oracle, shape):
    rng = np.random.default_rng(2)
    x = rng.random(shape).astype(np.float32) * 1.4 - 0.2
    thr = np.float32(np.linspace(0, 1.0, 256)[[128, 3, 255][:shape[0]] if shape[0] <= 3 else 128])
    thr = np.broadcast_to(thr, (shape[0],)).astype(np.float32).copy()
    x[0].ravel()[:50] = thr[0]                                    # values equal to the threshold are NOT selected
    for clip in (False, True):
        xyz, cnt = ops.threshold_compact(ctx, torch.from_numpy(x).to(ctx.device), torch.from_numpy(thr).to(ctx.device), clip=clip)
        cnt = cnt.cpu().numpy()
        for b in range(shape[0]):
            ref = oracle.threshold_argwhere(oracle.clip01(x[b]) if clip else x[b], thr[b])
            assert cnt[b] == len(ref)
            assert_array_equal(xyz[b, :cnt[b]].cpu().numpy(), ref)  # same points, same (x,y,z) lexicographic order
    # empty and full
    xyz, cnt = ops.threshold_compact(ctx, torch.zeros((1, 8, 8, 8), device=ctx.device), torch.tensor([0.5], device=ctx.device))
    assert int(cnt[0]) == 0
    xyz, cnt = ops.threshold_compact(ctx, torch.ones((1, 8, 8, 8), device=ctx.device), torch.tensor([0.5], device=ctx.device), cap=100)
    assert int(cnt[0]) == 512                                       # count is exact even when cap truncates the list


def test_voxelize_matches_sparse_to_dense(ctx):
    from pcc_geo_cnn_v2_amd.model_types import sparse_to_dense
    rng = np.random.default_rng(3)
    blocks = [rng.integers(0, 16, (n, 3)).astype(np.float64) for n in (50, 1, 300)]
    pts = np.concatenate(blocks).astype(np.int32)
    bof = np.concatenate([np.full(len(b), i, np.int32) for i, b in enumerate(blocks)])
    d = ops.voxelize(ctx, torch.from_numpy(pts).to(ctx.device), torch.from_numpy(bof).to(ctx.device), 3, 16, 16, 16).cpu().numpy()
    for i, b in enumerate(blocks):
        assert_array_equal(d[i], sparse_to_dense(b, (1, 1, 16, 16, 16), 'channels_first')[0, 0])


def test_focal_loss_matches_oracle_and_is_deterministic(ctx, oracle):
    rng = np.random.default_rng(4)
    n = 32 * 64 ** 3 // 8
    yt = (rng.random(n) < 0.02).astype(np.float32)
    yp = (rng.random(n) * 1.2 - 0.1).astype(np.float32)
    a, b = torch.from_numpy(yt).to(ctx.device), torch.from_numpy(yp).to(ctx.device)
    for gamma, alpha in [(2, 0.9), (2, 0.75)]:
        got = float(focal_loss(ctx, a, b, gamma, alpha))
        ref = oracle.focal_loss(yt, yp, gamma, alpha)
        assert abs(got - ref) <= 2e-5 * abs(ref)                     # stated fp32 tolerance of the reduction
        assert float(focal_loss(ctx, a, b, gamma, alpha)) == got     # fixed reduction order
