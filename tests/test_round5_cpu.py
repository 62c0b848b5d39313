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
                assert m.group(1) in ('PCC_NO_THR_FUSE',), f'{fn} reads {m.group(1)} per call: numerics switches live in pcc_ctx'
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
