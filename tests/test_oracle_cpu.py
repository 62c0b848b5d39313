"""CPU: the oracle's two independent restatements agree with each other, honour the reference's shape
contract, and the product's host-side table builders agree with the oracle's."""
import numpy as np
import pytest

from oracle import torch_oracle as T
from pcc_geo_cnn_v2_amd import entropy_models as EM


@pytest.mark.parametrize('D,k,s,ci,co', [(8, 3, 1, 4, 5), (7, 3, 2, 3, 4), (8, 5, 2, 2, 3), (9, 9, 2, 1, 2), (6, 3, 2, 16, 16)])
def test_c_loops_match_torch_restatement(oracle, D, k, s, ci, co):
    rng = np.random.default_rng(0)
    x = rng.standard_normal((2, D, D + 1, D + 2, ci)).astype(np.float32)
    w = rng.standard_normal((k, k, k, ci, co)).astype(np.float32)
    b = rng.standard_normal(co).astype(np.float32)
    a, t = oracle.conv3d(x, w, b, s, True), T.conv3d(x, w, b, s, True).numpy()
    assert a.shape == t.shape and np.abs(a - t).max() <= 1e-5 * (1 + np.abs(a).max()) * k ** 3 / 27
    wt = rng.standard_normal((k, k, k, co, ci)).astype(np.float32)
    a, t = oracle.conv3d_transpose(x, wt, b, s, False), T.conv3d_transpose(x, wt, b, s, False).numpy()
    assert a.shape == t.shape and np.abs(a - t).max() <= 1e-5 * (1 + np.abs(a).max()) * k ** 3 / 27


@pytest.mark.parametrize('n,k,s', [(8, 3, 2), (8, 5, 2), (16, 9, 2), (6, 3, 1)])
def test_transposed_conv_is_adjoint_of_same_conv(oracle, n, k, s):
    """Conv3DTranspose(SAME) == adjoint of Conv3D(SAME) on an input of size n*s."""
    rng = np.random.default_rng(1)
    ci, co = 2, 3
    x = rng.standard_normal((1, n, n, n, ci)).astype(np.float32)
    w = rng.standard_normal((k, k, k, ci, co)).astype(np.float32)
    y = rng.standard_normal((1, n // s, n // s, n // s, co)).astype(np.float32)
    lhs = (oracle.conv3d(x, w, None, s).astype(np.float64) * y).sum()
    rhs = (x.astype(np.float64) * oracle.conv3d_transpose(y, w, None, s)).sum()
    assert abs(lhs - rhs) <= 1e-4 * (1 + abs(lhs))


def _params(oracle, name, F, cin, rng):
    p = {}
    for i, (kind, cout, k, s, bias, relu, res) in enumerate(oracle.transform_layers(name, F)):
        shape = (k, k, k, cout, cin) if kind == 'convT' else (k, k, k, cin, cout)
        p[f't/{i}/kernel'] = rng.standard_normal(shape).astype(np.float32) * 0.1
        if bias:
            p[f't/{i}/bias'] = rng.standard_normal(cout).astype(np.float32) * 0.1
        cin = cout
    return p


def test_shape_contract_of_reference_tests(oracle):
    """src/test_model_transforms.py:27-73: /8 and x8 for transforms, /2 and x2 for hyper transforms."""
    rng = np.random.default_rng(0)
    x8, y1 = np.zeros((1, 8, 8, 8, 1), np.float32), np.zeros((1, 1, 1, 1, 1), np.float32)
    for name, F in [('AnalysisTransformV1', 1), ('AnalysisTransformV2', 2), ('AnalysisTransformProgressiveV2', 4)]:
        assert oracle.run_transform(name, F, _params(oracle, name, F, 1, rng), 't', x8).shape == (1, 1, 1, 1, F)
    for name, F in [('SynthesisTransformV1', 2), ('SynthesisTransformV2', 2), ('SynthesisTransformProgressiveV2', 4)]:
        assert oracle.run_transform(name, F, _params(oracle, name, F, 1, rng), 't', y1).shape == (1, 8, 8, 8, 1)
    assert oracle.run_transform('HyperAnalysisTransform', 1, _params(oracle, 'HyperAnalysisTransform', 1, 1, rng), 't', x8).shape == (1, 4, 4, 4, 1)
    assert oracle.run_transform('HyperSynthesisTransform', 1, _params(oracle, 'HyperSynthesisTransform', 1, 1, rng), 't', y1).shape == (1, 2, 2, 2, 1)


def test_focal_loss_oracle_vs_numpy(oracle):
    """src/utils/focal_loss.py:5-12 written out in numpy float64."""
    rng = np.random.default_rng(3)
    yt = (rng.random(5000) < 0.1).astype(np.float32)
    yp = rng.random(5000).astype(np.float32) * 1.2 - 0.1
    for gamma, alpha in [(2, 0.9), (2, 0.75), (1.5, 0.5)]:
        pt1 = np.clip(np.where(yt == 1, yp, 1.0), 1e-3, .999).astype(np.float64)
        pt0 = np.clip(np.where(yt == 0, yp, 0.0), 1e-3, .999).astype(np.float64)
        ref = -np.sum(alpha * (1 - pt1) ** gamma * np.log(pt1)) - np.sum((1 - alpha) * pt0 ** gamma * np.log(1 - pt0))
        assert abs(oracle.focal_loss(yt, yp, gamma, alpha) - ref) <= 1e-5 * abs(ref)


def test_quantize_and_index_rules(oracle):
    v = np.array([[-1.5, -0.5, 0.5, 1.5, 2.5, 0.49999997, -0.50000006, 0.3]], np.float32)
    sym, deq = oracle.quantize(v, None, 0)
    assert sym.tolist() == [[-1, 0, 1, 2, 3, 1, -1, 0]]          # floor(v + 0.5) in float32 (tfc 1.3)
    sym, _ = oracle.quantize(v, None, 1)
    assert sym.tolist() == [[-2, -0, 0, 2, 2, 0, -1, 0]]         # round half to even (tf.round)
    med = np.array([0.25] * 8, np.float32)
    sym, deq = oracle.quantize(v, med, 0)
    assert np.array_equal(deq, sym.astype(np.float32) + med)
    tab = oracle.scale_table().astype(np.float32)
    s = np.array([0.0, 0.05, tab[0], np.nextafter(tab[0], np.float32(1)), tab[5], tab[62], tab[63], 1e9], np.float32)
    assert oracle.scale_index(s, tab).tolist() == [0, 0, 0, 1, 5, 62, 63, 63]


def test_gaussian_tables_product_vs_oracle(oracle):
    tab = EM.scale_table()
    gc = EM.GaussianConditional(tab)
    cdf, size, off = oracle.gaussian_tables(tab)
    assert np.array_equal(gc.quantized_cdf, cdf) and np.array_equal(gc.cdf_length, size) and np.array_equal(gc.offset, off)
    assert gc.quantized_cdf.shape == (64, 1481) and off[0] == -1 and off[-1] == -739   # tail_mass 2**-8 (tfc 1.3 default)
    for r in range(64):
        row = cdf[r, :size[r]]
        assert row[0] == 0 and row[-1] == 65536 and np.all(np.diff(row) >= 1)
    wide = EM.GaussianConditional(tab, tail_mass=1e-9)                                    # SURVEY.md's recollection
    assert wide.quantized_cdf.shape == (64, 3133)


def test_factorized_tables_product_vs_oracle(oracle):
    for scale in (10, 0.2):
        p = EM.EntropyBottleneck.init_params(8, init_scale=scale, seed=5)
        eb = EM.EntropyBottleneck(8, params=p)
        o = oracle.factorized_tables(dict(matrices=[p[f'matrix_{i}'] for i in range(4)], biases=[p[f'bias_{i}'] for i in range(4)],
                                          factors=[p[f'factor_{i}'] for i in range(3)], quantiles=p['quantiles']))
        assert np.array_equal(eb.quantized_cdf, o['cdf']) and np.array_equal(eb.cdf_length, o['cdf_size'])
        assert np.array_equal(eb.offset, o['offset']) and np.array_equal(eb.medians, o['medians'])
    assert eb.cdf_length.tolist() == [5] * 8 and eb.offset.tolist() == [-1] * 8


def test_oracle_block_roundtrip_small(oracle):
    """compress_block -> decompress_block of the oracle itself (c2 at 16^3: V2 wiring, k9/k5 layers)."""
    from pcc_geo_cnn_v2_amd.init_checkpoint import make_synthetic_weights
    w = make_synthetic_weights('c2', seed=1, gain_analysis=2.0, gain_synthesis=1.5)
    om = dict(config='c2', params=w, round_mode=0, scale_table=oracle.scale_table().astype(np.float32),
              eb=dict(cdf=w['entropy_bottleneck/quantized_cdf'], cdf_size=w['entropy_bottleneck/cdf_length'],
                      offset=w['entropy_bottleneck/offset'], medians=w['entropy_bottleneck/quantiles'][:, 0, 1]),
              gc=(w['gaussian_conditional/quantized_cdf'], w['gaussian_conditional/cdf_length'], w['gaussian_conditional/offset']))
    x = (np.random.default_rng(0).random((1, 16, 16, 16, 1)) < 0.1).astype(np.float32)
    strings, x_hat, dbg = oracle.compress_block(om, x)
    x_dec, ddbg = oracle.decompress_block(om, strings, (16, 16, 16))
    assert np.array_equal(dbg['y_hat'], ddbg['y_hat']) and np.array_equal(x_hat, x_dec)
    assert len(strings) == 2 and x_hat.shape == (16, 16, 16)


@pytest.mark.parametrize('name,data_format', [('c1', 'channels_first'), ('c3p', 'channels_first'), ('c3p', 'channels_last')])
def test_stagecheck_accepts_an_independent_restatement(oracle, name, data_format):
    """The staged parity checker used by the GPU tests (tests/_stagecheck.py), run on CPU: the oneDNN restatement plays the
    part of the implementation under test and must pass every stage against the C loops -- and a corrupted string, a flipped
    symbol or a wrong stream order must fail."""
    import _stagecheck as SC
    from oracle import torch_oracle as T
    from pcc_geo_cnn_v2_amd.init_checkpoint import make_synthetic_weights
    w = make_synthetic_weights(name, seed=3, gain_analysis=2.0, gain_synthesis=1.5)
    om = dict(config=name, params=w, round_mode=0, data_format=data_format,
              eb=dict(cdf=w['entropy_bottleneck/quantized_cdf'], cdf_size=w['entropy_bottleneck/cdf_length'],
                      offset=w['entropy_bottleneck/offset'], medians=w['entropy_bottleneck/quantiles'][:, 0, 1]))
    if name != 'c1':
        om['scale_table'] = oracle.scale_table().astype(np.float32)
        om['gc'] = (w['gaussian_conditional/quantized_cdf'], w['gaussian_conditional/cdf_length'], w['gaussian_conditional/offset'])
    dense = (np.random.default_rng(1).random((16, 16, 16)) < 0.1).astype(np.float32)
    strings, _, g = oracle.compress_block(om, dense[None, ..., None], run=T.run_transform)
    info = SC.check_block(oracle, om, dense, g, strings)
    assert info['sym_flips'] <= 4
    bad = list(strings)
    bad[0] = bad[0][:-1] + bytes([bad[0][-1] ^ 1])
    with pytest.raises(AssertionError):
        SC.check_block(oracle, om, dense, g, tuple(bad))
    g2 = dict(g)
    g2['symbols'] = g['symbols'].copy()
    g2['symbols'].ravel()[5] += 1
    with pytest.raises(AssertionError):
        SC.check_block(oracle, om, dense, g2, strings)
    other = dict(om, data_format='channels_last' if data_format == 'channels_first' else 'channels_first')
    with pytest.raises(AssertionError):
        SC.check_block(oracle, other, dense, g, strings)
