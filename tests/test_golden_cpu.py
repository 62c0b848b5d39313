"""CPU: host modules and the oracle against the golden fixtures captured from the reference's own
importable modules (tests/golden/make_golden.py) and against the known answers of the reference's tests."""
import io
import os

import numpy as np
import pytest
from numpy.testing import assert_array_equal

from pcc_geo_cnn_v2_amd import model_opt, model_syntax
from pcc_geo_cnn_v2_amd.utils import octree_coding, pc_io, pc_metric

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def _syntax_case(g, i):
    lens, cat, ns = g[f'c{i}_strlens'], g[f'c{i}_strcat'].tobytes(), int(g[f'c{i}_nstr'][0])
    strs, p = [], 0
    for l in lens:
        strs.append(cat[p:p + l])
        p += l
    blocks = [(strs[j * ns:(j + 1) * ns], int(t)) for j, t in enumerate(g[f'c{i}_thr'])]
    return list(g[f'c{i}_binstr']), blocks, int(g[f'c{i}_res_level'][0]), int(g[f'c{i}_res_level'][1])


def test_container_bytes_match_reference():
    g = np.load(os.path.join(G, 'model_syntax.npz'))
    for i in range(int(g['n_cases'][0])):
        binstr, blocks, res, level = _syntax_case(g, i)
        data = model_syntax.save_compressed_file(binstr, blocks, res, level)
        assert data == g[f'c{i}_bytes'].tobytes()
        r, l, b, bl = model_syntax.load_compressed_file(io.BytesIO(g[f'c{i}_bytes'].tobytes()))
        assert (r, l) == (res, level) and list(b) == binstr
        for (s0, t0), (s1, t1) in zip(blocks, bl):
            assert list(s0) == list(s1) and t0 == t1
    # the example bytes quoted in SURVEY.md row X
    assert model_syntax.save_compressed_file([3, 129], [([b'a'], 1)], 1024, 4)[:10] == bytes.fromhex('00040401000102000381')


def test_model_syntax_save_load_like_reference_test():
    """mirror of src/test_model_syntax.py:8-30"""
    binstr = [1, 2, 3]
    data_b_list = [[[b'abc', b'efg'], 35], [[b'xyz', b'uvw'], 7]]
    c = model_syntax.save_compressed_file(binstr, data_b_list, 512, 4)
    resolution_dec, level_dec, binstr_dec, blocks_dec = model_syntax.load_compressed_file(io.BytesIO(c))
    assert_array_equal(binstr, binstr_dec)
    for data, data_dec in zip(data_b_list, blocks_dec):
        assert_array_equal(data[0], data_dec[0])
        assert data[1] == data_dec[1]
    assert level_dec == 4 and resolution_dec == 512
    overflow = [[[b'abc'] * (2 ** 16 + 1), 35]]
    c = model_syntax.save_compressed_file(binstr, overflow, 512, 4)          # wraps like numpy 1.18 ...
    with pytest.raises(AssertionError):
        model_syntax.load_compressed_file(io.BytesIO(c))                      # ... and surfaces at load
    c = model_syntax.save_compressed_file(binstr, overflow, 512, -1)
    with pytest.raises(AssertionError):
        model_syntax.load_compressed_file(io.BytesIO(c))
    with pytest.raises(AssertionError):                                        # strict mode: raise at save
        model_syntax.save_compressed_file(binstr, overflow, 512, 4, strict=True)
    with pytest.raises(AssertionError):
        model_syntax.save_compressed_file(binstr, data_b_list, 70000, 4, strict=True)


def test_octree_partition_matches_reference():
    g = np.load(os.path.join(G, 'octree_coding.npz'))
    for i in range(int(g['n_cases'][0])):
        res, level = (int(v) for v in g[f'o{i}_spec'])
        pts = g[f'o{i}_points']
        blocks, binstr = octree_coding.partition_octree(pts, [0, 0, 0], [res] * 3, level)
        assert binstr == list(g[f'o{i}_binstr'])
        assert [len(b) for b in blocks] == list(g[f'o{i}_block_len'])
        assert_array_equal(np.vstack(blocks), g[f'o{i}_blocks_cat'])
        assert all(b.dtype == np.float64 for b in blocks)
        dep = octree_coding.departition_octree(blocks, np.array(binstr, np.uint8), [0, 0, 0], [res] * 3, level)
        assert_array_equal(np.vstack(dep), g[f'o{i}_depart_cat'])
        # float32 decoded blocks + int64 origins promote to float64, like the reference
        dep32 = octree_coding.departition_octree([b[:, :3].astype(np.float32) for b in blocks], binstr, [0, 0, 0], [res] * 3, level)
        assert dep32[0].dtype == np.float64


def test_octree_edge_cases():
    pts = np.array([[0, 0, 0], [63, 63, 63], [1, 0, 0]], float)
    blocks, binstr = octree_coding.partition_octree(pts, [0, 0, 0], [64] * 3, 0)
    assert binstr is None and len(blocks) == 1
    blocks, binstr = octree_coding.partition_octree(np.zeros((0, 3)), [0, 0, 0], [64] * 3, 3)
    assert binstr is None
    # level 1 (the reference's departition raises IndexError here, octree_coding.py:164; ours works)
    blocks, binstr = octree_coding.partition_octree(pts, [0, 0, 0], [64] * 3, 1)
    assert binstr == [0b10000001] and [len(b) for b in blocks] == [2, 1]
    dep = octree_coding.departition_octree(blocks, binstr, [0, 0, 0], [64] * 3, 1)
    assert_array_equal(np.vstack(dep), [[0, 0, 0], [1, 0, 0], [63, 63, 63]])
    # Morton order has x as the least significant axis
    pts = np.array([[0, 0, 64], [0, 64, 0], [64, 0, 0], [0, 0, 0], [64, 64, 0]], float)
    blocks, _ = octree_coding.partition_octree(pts, [0, 0, 0], [128] * 3, 1)
    origins = octree_coding.block_origins([0b00011111], [0, 0, 0], [128] * 3, 1)
    assert [tuple(o) for o in origins] == [(0, 0, 0), (64, 0, 0), (0, 64, 0), (64, 64, 0), (0, 0, 64)]


def test_threshold_search_and_metrics_match_reference():
    g = np.load(os.path.join(G, 'model_opt.npz'))
    thr = np.linspace(0, 1.0, 256)
    for i in range(int(g['n_cases'][0])):
        block, xh = g[f'm{i}_block'], g[f'm{i}_x_hat']
        for fixed in (0, 1):
            names, best = model_opt.compute_optimal_thresholds(block, xh, thr, 64, opt_metrics=['d1_mse', 'd1_sum_mean'],
                                                               max_deltas=[np.inf], fixed_threshold=bool(fixed))
            assert names == list(g[f'm{i}_names_fixed{fixed}']) and best == list(g[f'm{i}_best_fixed{fixed}'])
        met = pc_metric.compute_metrics(block, g[f'm{i}_pa100'], 63)
        np.testing.assert_allclose([met[k] for k in g[f'm{i}_metric_keys']], g[f'm{i}_metric_vals'], rtol=1e-12)
        pal = model_opt.build_points_threshold(xh, thr, len(block), max_delta=2.0)
        assert [j for j, _ in pal] == list(g[f'm{i}_bpt_idx']) and [len(p) for _, p in pal] == list(g[f'm{i}_bpt_len'])
    met = pc_metric.compute_metrics(g['d2_block'], g['d2_p2'], 63, p1_n=g['d2_normals'])
    np.testing.assert_allclose([met[k] for k in g['d2_metric_keys']], g['d2_metric_vals'], rtol=1e-9)


def test_model_opt_known_answers_like_reference_test(oracle):
    """mirror of src/test_model_opt.py:12-49, for the host module AND the oracle's argwhere"""
    x_hat = np.array([[0, 2, 4, 6], [2, 4, 6, 0]])
    thresholds = np.array([1, 3, 5, 7])
    expected = [[[0, 1], [0, 2], [0, 3], [1, 0], [1, 1], [1, 2]], [[0, 2], [0, 3], [1, 1], [1, 2]], [[0, 3], [1, 2]]]
    pa_list = model_opt.build_points_threshold(x_hat, thresholds, 2)
    assert [i for i, _ in pa_list] == [0, 1, 2]
    for (_, pa), e in zip(pa_list, expected):
        assert_array_equal(pa, e)
    assert [i for i, _ in model_opt.build_points_threshold(x_hat, thresholds, 2, max_delta=2.5)] == [1, 2]
    assert [i for i, _ in model_opt.build_points_threshold(x_hat, thresholds, 2, max_delta=2)] == [2]
    for t, e in zip(thresholds, expected):  # oracle (3-D): same index lists with a leading singleton axis
        got = oracle.threshold_argwhere(x_hat[None].astype(np.float32), t)
        assert_array_equal(got[:, 1:], e)
    block, xh = np.array([[0, 0]]), np.array([[0, 1]])
    r = np.sqrt(2)
    assert model_opt.compute_optimal_thresholds(block, xh, np.array([0, 1.5, 3.0]), r, opt_metrics=['d1_mse'],
                                                max_deltas=[np.inf]) == (['d1_mse_inf'], [2])
    assert model_opt.compute_optimal_thresholds(block, xh, np.array([0, 1.5, 3.0]), r, opt_metrics=['d1_mse'],
                                                max_deltas=[np.inf], fixed_threshold=True) == (['d1_mse_inf'], [1])
    assert model_opt.compute_optimal_thresholds(block, xh, np.array([0, 1.5, 3.0, 4.5, 6.0]), r, opt_metrics=['d1_mse'],
                                                max_deltas=[np.inf], fixed_threshold=True) == (['d1_mse_inf'], [2])


def test_threshold_compare_is_float32(oracle):
    """x_hat == float32(128/255) is NOT above thresholds[128] in float32 arithmetic."""
    thr64 = np.linspace(0, 1.0, 256)[128]
    x = np.full((1, 1, 4), np.float32(thr64), np.float32)
    x[0, 0, 1] = np.nextafter(np.float32(thr64), np.float32(2))
    assert float(np.float32(thr64)) > thr64           # the float32 rounding of the threshold is above the float64 value
    assert len(oracle.threshold_argwhere(x, np.float32(thr64))) == 1
    assert len(np.argwhere(model_opt._gt(x, thr64))) == 1


def test_ply_roundtrip(tmp_path):
    pts = np.random.default_rng(0).integers(0, 64, (100, 3)).astype(np.float32)
    for text in (False, True):
        p = str(tmp_path / f'a{int(text)}.ply')
        pc_io.write_ply(p, pc_io.pa_to_df(pts), as_text=text)
        assert_array_equal(pc_io.load_pc(p), pts)
    n = np.random.default_rng(1).normal(size=(100, 3)).astype(np.float32)
    import pandas as pd
    df = pd.DataFrame({'x': pts[:, 0], 'y': pts[:, 1], 'z': pts[:, 2], 'nx': n[:, 0], 'ny': n[:, 1], 'nz': n[:, 2]})
    p = str(tmp_path / 'n.ply')
    pc_io.write_ply(p, df)
    assert_array_equal(pc_io.load_normals(p), n)
    pmin, pmax, shape = pc_io.get_shape_data(64, 'channels_first')
    assert list(shape) == [1, 64, 64, 64] and list(pc_io.get_shape_data(64, 'channels_last')[2]) == [64, 64, 64, 1]


def test_tf_checkpoint_reader_against_hand_assembled_bundle(tmp_path):
    """pcc_geo_cnn_v2_amd/tf_checkpoint.py against tests/golden/tf_bundle/: a TensorBundle assembled byte by byte from the
    published format constants by tests/golden/make_tf_bundle.py (no code shared with the reader or with the test-side writer
    tests/tf_bundle_writer.py): two data shards, a snappy-compressed index block (literal, copy-1, copy-2 elements),
    prefix-compressed keys, masked CRC-32C everywhere."""
    import shutil
    from pcc_geo_cnn_v2_amd import tf_checkpoint as T
    src = os.path.join(G, 'tf_bundle')
    prefix = T.latest_checkpoint(src)
    assert os.path.basename(prefix) == 'model.ckpt-4242'
    header, entries = T.read_index(prefix, verify=True)
    assert header['num_shards'] == 2 and {e['shard_id'] for e in entries.values()} == {0, 1}
    got = T.load_checkpoint(prefix, verify=True)
    exp = np.load(os.path.join(src, 'expected.npz'))
    assert sorted(got) == sorted(k.replace('|', '/') for k in exp.files) and len(got) == 12
    for k in exp.files:
        a, b = got[k.replace('|', '/')], exp[k]
        assert a.dtype == b.dtype and a.shape == b.shape and np.array_equal(a, b), k
    assert got['global_step'] == 4242 and got['unused/empty'].shape == (0, 3)
    # checksums are really checked: flip one bit of the compressed index block / of a tensor
    dst = str(tmp_path / 'b')
    shutil.copytree(src, dst)
    p = os.path.join(dst, 'model.ckpt-4242')
    raw = bytearray(open(p + '.index', 'rb').read())
    raw[300] ^= 0x10
    open(p + '.index', 'wb').write(bytes(raw))
    with pytest.raises((AssertionError, ValueError, IndexError)):
        T.load_checkpoint(p, verify=True)
    shutil.copy(os.path.join(src, 'model.ckpt-4242.index'), p + '.index')
    raw = bytearray(open(p + '.data-00001-of-00002', 'rb').read())
    raw[7] ^= 1
    open(p + '.data-00001-of-00002', 'wb').write(bytes(raw))
    with pytest.raises(AssertionError):
        T.load_checkpoint(p, verify=True)
    T.load_checkpoint(p, verify=False)


def test_host_search_pool_matches_reference_answers():
    """model_opt.HostSearchPool (worker processes for the KD-tree threshold search used when normals / d2_* metrics are requested)
    gives the reference's own answers on the fixtures and, with normals, the answers of the in-process call."""
    f = np.load(os.path.join(G, 'model_opt.npz'), allow_pickle=True)
    thr = np.linspace(0, 1.0, 256)
    rng = np.random.default_rng(0)
    jobs = [(f[f'm{i}_block'], f[f'm{i}_x_hat'], thr, 64, False, ['d1_mse', 'd1_sum_mean'], [np.inf]) for i in range(4)]
    blk = np.hstack([f['m0_block'], rng.normal(size=f['m0_block'].shape)])
    jobs.append((blk, f['m0_x_hat'], thr, 64, True, ['d1_mse', 'd2_mse'], [np.inf, 2.0]))
    pool = model_opt.HostSearchPool(3)
    try:
        out = pool.map(jobs)
        bad = pool.map([(f['m0_block'], f['m0_x_hat'], thr, 64, False, ['d2_mse'], [np.inf])])      # d2 without normals
    except AssertionError as e:
        assert 'not available without normals' in str(e)
        bad = None
    finally:
        pool.close()
    assert bad is None
    for i in range(4):
        assert out[i][1] == [int(v) for v in f[f'm{i}_best_fixed0']] and list(out[i][0]) == list(f[f'm{i}_names_fixed0'])
    names, best = model_opt.compute_optimal_thresholds(blk, f['m0_x_hat'], thr, 64, normals=blk[:, 3:6], opt_metrics=['d1_mse', 'd2_mse'],
                                                       max_deltas=[np.inf, 2.0])
    assert out[4] == (names, [int(b) for b in best]) and len(best) == 4
