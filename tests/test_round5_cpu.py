"""CPU: round-5 host logic -- the codec-numerics tag of the CLI streams, the numerics switch table of the ABI, coder-wrapper validation,
the trace summariser."""
import gzip
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_tagged_gzip_is_a_plain_gzip_file_for_the_reference_reader(tmp_path):
    """compress_octree writes <target> through model_syntax.write_tagged_gzip: `gzip.open(...).read()` -- the reference's reader,
    /root/reference/src/decompress_octree.py:61 -- returns the container bytes unchanged, the tag sits in the member header's FCOMMENT;
    a file written by `gzip.open(..., 'wb')` (the reference's writer, compress_octree.py:112) carries no tag."""
    from pcc_geo_cnn_v2_amd import model_syntax as MS
    rng = np.random.default_rng(0)
    for n in (0, 1, 70000):
        payload = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
        p = str(tmp_path / f'a{n}.ply.bin')
        MS.write_tagged_gzip(p, payload, 'pcc_geo_cnn_v2_amd/k5/sw0000/fp32')
        with gzip.open(p, 'rb') as fh:
            assert fh.read() == payload
        assert MS.read_gzip_tag(p) == 'pcc_geo_cnn_v2_amd/k5/sw0000/fp32'
    q = str(tmp_path / 'ref.ply.bin')
    with gzip.open(q, 'wb') as fh:
        fh.write(b'xyz')
    assert MS.read_gzip_tag(q) is None
    open(str(tmp_path / 'junk'), 'wb').write(b'not gzip')
    assert MS.read_gzip_tag(str(tmp_path / 'junk')) is None


def test_decoder_refuses_a_stream_of_other_numerics():
    from pcc_geo_cnn_v2_amd import model_syntax as MS
    mine = 'pcc_geo_cnn_v2_amd/k5/sw0000/fp32'
    MS.check_numerics_tag(mine, mine)
    MS.check_numerics_tag(None, mine)                       # the reference's own files: nothing to compare
    for other in ('pcc_geo_cnn_v2_amd/k4/sw0000/fp32', 'pcc_geo_cnn_v2_amd/k5/sw0001/fp32', 'pcc_geo_cnn_v2_amd/k5/sw0000/fp16'):
        with pytest.raises(RuntimeError, match='codec numerics'):
            MS.check_numerics_tag(other, mine)
        MS.check_numerics_tag(other, mine, ignore=True)


def test_numerics_switch_table_matches_the_header():
    """_lib.PCC_NUM (what ops.Context.set_numerics takes) == the PCC_NUM_* defines of include/pcc_geo.h, bit for bit; every switch has an
    environment variable of the same name read in ONE place (csrc/ctx.hip), and no other translation unit calls getenv for a switch."""
    from pcc_geo_cnn_v2_amd import _lib as L
    hdr = open(os.path.join(ROOT, 'include', 'pcc_geo.h')).read()
    defs = {m.group(1).lower(): int(m.group(2), 16) for m in re.finditer(r'#define PCC_NUM_([A-Z0-9_]+) (0x[0-9a-fA-F]+)', hdr)}
    assert defs == L.PCC_NUM and len(defs) >= 15
    assert len(set(defs.values())) == len(defs) and all(v & (v - 1) == 0 for v in defs.values())
    ctx_src = open(os.path.join(ROOT, 'pcc_geo_cnn_v2_amd', 'csrc', 'ctx.hip')).read()
    for name in defs:
        env = 'PCC_' + name.upper()
        env = {'PCC_SPLIT_MFMA16': 'PCC_SPLIT_MFMA', 'PCC_SPLIT_MFMA32': 'PCC_SPLIT_MFMA', 'PCC_SPLIT_TILE8': 'PCC_SPLIT_TILE'}.get(env, env)
        assert f'"{env}"' in ctx_src, env
    csrc = os.path.join(ROOT, 'pcc_geo_cnn_v2_amd', 'csrc')
    for fn in os.listdir(csrc):
        if fn.endswith(('.hip', '.cpp', '.h')) and fn != 'ctx.hip':
            for m in re.finditer(r'getenv\("(PCC_[A-Z0-9_]+)"\)', open(os.path.join(csrc, fn)).read()):
                # (PCC_EDT_OLD selects between two kernels that produce the same integers -- tests/test_round6_gpu.py -- it is not a numerics switch)
                    assert m.group(1) in ('PCC_NO_THR_FUSE', 'PCC_EDT_OLD'), f'{fn} reads {m.group(1)} per call: numerics switches live in pcc_ctx'
    assert int(re.search(r'#define PCC_KERNEL_FAMILY (\d+)', hdr).group(1)) >= 5


def test_range_decode_batch_rejects_mismatched_index_shapes(oracle):
    """the 2-D fast path of ops.range_decode_batch took any index_list -- a list of per-stream arrays was flattened and every
    stream decoded with stream 0's rows."""
    from pcc_geo_cnn_v2_amd import ops
    from test_abi_cpu import _tables

    class gc:
        table = ops.HostCdfTable(*_tables(oracle)[1])
    rng = np.random.default_rng(3)
    S, n = 3, 40
    idx = rng.integers(0, 64, (S, n)).astype(np.int32)
    sym = rng.integers(-2, 3, (S, n)).astype(np.int32)
    strings = ops.range_encode_batch(gc.table, [s for s in sym], [i for i in idx])
    out = np.empty((S, n), np.int32)
    ops.range_decode_batch(gc.table, strings, [n] * S, idx, out=out)
    assert np.array_equal(out, sym)
    with pytest.raises(AssertionError):
        ops.range_decode_batch(gc.table, strings, [n] * S, [i for i in idx], out=out)          # list + 2-D out
    with pytest.raises(AssertionError):
        ops.range_decode_batch(gc.table, strings, [n] * S, idx[:, :n - 1].copy(), out=out)       # wrong row length
    with pytest.raises(AssertionError):
        ops.range_decode_batch(gc.table, strings, [n] * S, idx[0, :n - 1].copy(), out=out)       # shared 1-D index of the wrong size
    same = ops.range_encode_batch(gc.table, [s for s in sym], [idx[0]] * S)
    ops.range_decode_batch(gc.table, same, [n] * S, idx[0].copy(), out=out)                       # one shared row
    assert np.array_equal(out, sym)


def test_trace_summary_reports_the_steady_state_not_the_first_launch(tmp_path):
    """one 28.8 ms first launch among 64 of 0.36 ms made profiles/ print 808 us for the dominant kernel.  The summariser prints
    median and trimmed mean, flags the row, and divides by the step count it is given."""
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    import summarize_trace as ST
    p = tmp_path / 't.csv'
    rows = ['Kernel_Name,Grid_Size_X,Start_Timestamp,End_Timestamp']
    t = 0
    for i in range(64):
        d = 28_825_472 if i == 0 else 363_000 + 100 * (i % 7)
        rows.append(f'"void k<true>(A)",131072,{t},{t + d}')
        t += d + 1000
    for i in range(16):
        rows.append(f'small,512,{t},{t + 3000}')
        t += 4000
    p.write_text('\n'.join(rows) + '\n')
    text, st = ST.table(ST.load(str(p)), 16)
    k = st[('void k<true>(A)', '131072')]
    assert k['outlier'] and 362_000 < k['median'] < 364_000 and 362_000 < k['trimmed'] < 364_000 and k['avg'] > 800_000
    assert not st[('small', '512')]['outlier']
    assert 'max > 10 x median' in text and '1.46 ms/step' in text          # (64 x 363.3 us + 16 x 3 us) / 16 steps


def test_tie_free_d2_fixture_pins_the_host_search_to_the_reference():
    """tests/golden/model_opt_d2_tiefree.npz (make_golden.py --round5-only: the REFERENCE's compute_optimal_thresholds and
    compute_metrics on six sparse blocks whose level sets -- all of them -- and mean-point query have unique nearest neighbours in both
    directions).  Without ties the reference's d2_* numbers do not depend on scipy's traversal: the host restatement must give the
    same decisions for every (metric, max_delta) and the same metric values at EVERY level set; so must the brute-force
    lowest-(x,y,z) restatement the GPU kernel is tested against (oracle.search_tallies_lowest_index)."""
    from oracle import oracle as O
    from pcc_geo_cnn_v2_amd import model_opt
    from pcc_geo_cnn_v2_amd.utils import pc_metric as PM
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'model_opt_d2_tiefree.npz'))
    thr = np.linspace(0, 1.0, 256)
    mets, deltas = [str(m) for m in g['opt_metrics']], [float(d) for d in g['max_deltas']]
    n = int(g['n_cases'][0])
    assert n >= 6
    seen = set()
    for i in range(n):
        blk, xh = g[f's{i}_block'], g[f's{i}_x_hat']
        assert all(O.tie_free(blk, xh, thr))
        names, best = model_opt.compute_optimal_thresholds(blk, xh, thr, 64, normals=blk[:, 3:6], opt_metrics=mets, max_deltas=deltas)
        assert names == [str(x) for x in g[f's{i}_names']] and best == [int(b) for b in g[f's{i}_best']], i
        seen.update(best)
        keys, want = [str(k) for k in g[f's{i}_keys']], g[f's{i}_vals']
        tallies, mean_tally = model_opt.host_threshold_stats(blk, xh, thr, blk[:, 3:6])
        brute = O.search_tallies_lowest_index(blk, xh, thr)
        assert tallies.shape == brute.shape == (want.shape[0], 5)
        # (the brute-force restatement rounds the normals to float32 like the GPU path: 1e-6; the host path computes in the block's dtype)
        assert np.array_equal(tallies[:, :3], brute[:, :3]) and np.allclose(tallies[:, 3:], brute[:, 3:], rtol=1e-6, atol=1e-300)
        for src, rtol in ((tallies, 1e-6 if blk.dtype != np.float64 else 1e-12), (brute, 1e-6)):
            table = PM.metrics_table(len(blk), src, 63)
            got = np.array([[table[k][t] for k in keys] for t in range(len(src))])
            np.testing.assert_allclose(got, want, rtol=rtol)
        # the same decisions from the table form, with the lexicographic mean-point guard of the GPU path
        assert model_opt.select_thresholds_from_stats(len(blk), brute, model_opt.mean_point_tally(blk, True), 256, 64, mets, deltas)[1] == best
    assert 255 in seen and 0 in seen and len(seen) >= 8          # guards fire, d1 and d2 disagree: the fixture decides something


def test_mean_point_guard_takes_the_lowest_xyz_whatever_the_storage_order():
    """mean_point_tally used the lowest ARRAY index among equidistant original points; the GPU path's rule is the lowest
    (x, y, z).  A block stored in another order must give the same tally."""
    from pcc_geo_cnn_v2_amd import model_opt
    a = np.array([[0, 0, 0], [2, 0, 0], [0, 2, 0], [2, 2, 0], [1, 1, 4]], np.float64)       # mean (1, 1, 0.8) -> (1, 1, 1): four equidistant points
    nrm = np.array([[1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 0], [0, 1, 1]], np.float64)
    blk = np.hstack([a, nrm])
    want = model_opt.mean_point_tally(blk, True)
    for perm in ([3, 2, 1, 0, 4], [4, 1, 3, 0, 2], [2, 4, 0, 3, 1]):
        assert np.array_equal(model_opt.mean_point_tally(blk[perm], True), want), perm
    d = (np.array([1.0, 1.0, 1.0]) - a[0]) @ nrm[0]                                          # the neighbour is (0, 0, 0): lowest (x, y, z)
    assert np.isclose(want[4], d * d)
