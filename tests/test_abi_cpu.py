"""CPU-only: the C-ABI library loads and exports every symbol include/pcc_geo.h declares; the host
range coder and CDF quantiser (product code) agree with the oracle's restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pcc_geo_cnn_v2_amd import _lib as L
from pcc_geo_cnn_v2_amd import ops

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, 'include', 'pcc_geo.h')).read()
    declared = set(re.findall(r'\b(pcc_[a-z0-9_]+)\s*\(', hdr))
    declared -= {'pcc_last_error'} - {'pcc_last_error'}  # keep all
    lib = C.CDLL(L.LIB_PATH)
    missing = [s for s in sorted(declared) if not hasattr(lib, s)]
    assert not missing, f'symbols declared in pcc_geo.h but not exported: {missing}'
    assert set(L.EXPORTS) <= declared
    assert L.lib().pcc_abi_version() == L.ABI_VERSION == 4


def test_network_layer_tables_match_the_oracle_restatement(oracle):
    """The C++ layer stacks behind pcc_network_forward (csrc/network.hip) against the oracle's independent restatement of
    /root/reference/src/model_transforms.py:41-158, and the packed weight blob against the per-layer packer (host only)."""
    names = ['AnalysisTransformV1', 'SynthesisTransformV1', 'AnalysisTransformV2', 'SynthesisTransformV2',
             'AnalysisTransformProgressiveV2', 'SynthesisTransformProgressiveV2', 'HyperAnalysisTransform', 'HyperSynthesisTransform']
    lib = L.lib()
    rng = np.random.default_rng(0)
    for tid, name in enumerate(names):
        F = 64 if 'Progressive' in name or 'Hyper' in name else 32
        ref = oracle.transform_layers(name, F)
        assert lib.pcc_network_num_layers(tid, F) == len(ref)
        cin = 1 if name.startswith('Analysis') else F
        d, role = L.ConvDesc(), C.c_int32()
        kernels, biases = [], []
        for i, (kind, cout, k, s, bias, relu, res) in enumerate(ref):
            assert lib.pcc_network_layer(tid, F, i, C.byref(d), C.byref(role)) == 0
            assert (d.Cin, d.Cout, d.k, d.stride, d.transposed) == (cin, cout, k, s, int(kind == 'convT')), (name, i)
            assert bool(d.flags & L.PCC_CONV_BIAS) == bias and bool(d.flags & L.PCC_CONV_RELU) == relu
            assert role.value == {None: 0, 'save': 1, 'add': 2}[res]
            shape = (k, k, k, cout, cin) if kind == 'convT' else (k, k, k, cin, cout)
            kernels.append(rng.standard_normal(shape).astype(np.float32))
            biases.append(rng.standard_normal(cout).astype(np.float32) if bias else None)
            cin = cout
        n = lib.pcc_weights_blob_floats(tid, F)
        blob = np.full(n, np.nan, np.float32)
        ks = (C.c_void_p * len(ref))(*[a.ctypes.data for a in kernels])
        bs = (C.c_void_p * len(ref))(*[None if b is None else b.ctypes.data for b in biases])
        assert lib.pcc_weights_pack(tid, F, ks, bs, blob.ctypes.data_as(C.c_void_p)) == 0
        assert not np.isnan(blob).any()
        # every layer's Keras kernel, its fragment image (== pcc_conv_pack_weights) and its bias appear in the blob, in order
        pos, cin = 0, (1 if name.startswith('Analysis') else F)
        al = lambda v: (v + 63) // 64 * 64
        for i, (kind, cout, k, s, bias, relu, res) in enumerate(ref):
            w = kernels[i].ravel()
            assert np.array_equal(blob[pos:pos + w.size], w)
            pos += al(w.size)
            dd = L.ConvDesc(1, 64, 64, 64, cin, cout, k, s, int(kind == 'convT'), 0, 0, 0, 0)
            npk = lib.pcc_conv_packed_floats(C.byref(dd))
            if npk:
                pk = np.empty(npk, np.float32)
                assert lib.pcc_conv_pack_weights(C.byref(dd), kernels[i].ctypes.data_as(C.c_void_p), pk.ctypes.data_as(C.c_void_p)) == 0
                assert np.array_equal(blob[pos:pos + npk], pk)
            pos += al(npk)
            if bias:
                assert np.array_equal(blob[pos:pos + cout], biases[i])
                pos += al(cout)
            cin = cout
        assert pos == n
    assert lib.pcc_network_num_layers(8, 32) < 0 and lib.pcc_weights_blob_floats(99, 32) == 0


def test_ctx_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    with pytest.raises(L.PccError):
        ops.Context(0)
    h = C.c_void_p()
    rc = L.lib().pcc_ctx_create(0, C.byref(h))
    assert rc < 0 and len(L.lib().pcc_last_error()) > 0


def _tables(oracle):
    tab = oracle.scale_table()
    return tab, oracle.gaussian_tables(tab)


def test_range_coder_matches_oracle_bytes_and_roundtrips(oracle):
    tab, (cdf, size, off) = _tables(oracle)
    table = ops.HostCdfTable(cdf, size, off)
    rng = np.random.default_rng(0)
    datas, idxs = [], []
    for s in range(13):
        n = int(rng.integers(0, 6000))
        idx = rng.integers(0, 64, n).astype(np.int32)
        scale = rng.choice([0.3, 1.0, 5.0], n)  # 5.0 forces overflow (escape) symbols
        datas.append(np.rint(rng.standard_normal(n) * tab[idx] * scale).astype(np.int32))
        idxs.append(idx)
    datas.append(np.array([2 ** 20, -2 ** 20, 0, 7], np.int32))  # huge escapes
    idxs.append(np.array([0, 63, 5, 5], np.int32))
    strings = ops.range_encode_batch(table, datas, idxs, n_threads=4)
    for d, i, s in zip(datas, idxs, strings):
        assert s == oracle.range_encode(d, i, cdf, size, off)
    dec = ops.range_decode_batch(table, strings, [d.size for d in datas], idxs, n_threads=3)
    for d, o in zip(datas, dec):
        assert np.array_equal(d, o)
    # oracle decodes the product's strings too
    for d, i, s in zip(datas, idxs, strings):
        assert np.array_equal(oracle.range_decode(s, i, cdf, size, off), d)


def test_range_coder_channel_mode_and_empty(oracle):
    rng = np.random.default_rng(1)
    Cn = 8
    pmf = rng.random((Cn, 21)).astype(np.float32)
    pmf /= pmf.sum(1, keepdims=True) * 1.01
    cdf = np.zeros((Cn, 23), np.int32)
    for c in range(Cn):
        cdf[c, :23] = oracle.pmf_to_quantized_cdf(np.concatenate([pmf[c], [0.0099]]).astype(np.float32))
    size = np.full(Cn, 23, np.int32)
    off = np.full(Cn, -10, np.int32)
    table = ops.HostCdfTable(cdf, size, off)
    data = rng.integers(-14, 15, (4 * 4 * 4, Cn)).astype(np.int32)
    (s,) = ops.range_encode_batch(table, [data], None, index_mod=Cn)
    ch = np.broadcast_to(np.arange(Cn, dtype=np.int32), data.shape)
    assert s == oracle.range_encode(data, ch, cdf, size, off)
    (d,) = ops.range_decode_batch(table, [s], [data.size], None, index_mod=Cn)
    assert np.array_equal(d.reshape(data.shape), data)
    assert ops.range_encode_batch(table, [np.zeros(0, np.int32)], None, index_mod=Cn) == [b'']
    assert ops.range_encode_batch(table, []) == []


def test_range_decoder_rejects_bad_row():
    cdf = np.array([[0, 1 << 15, 1 << 16]], np.int32)
    table = ops.HostCdfTable(cdf, [3], [0])
    with pytest.raises(AssertionError):
        ops.range_encode_batch(table, [np.zeros(3, np.int32)], [np.array([0, 1, 0], np.int32)])


def test_pmf_to_quantized_cdf_matches_oracle(oracle):
    rng = np.random.default_rng(2)
    for n in (2, 5, 64, 1479):
        for trial in range(5):
            p = rng.random(n).astype(np.float32) ** 4
            p /= p.sum() * rng.choice([0.97, 1.0, 1.04])
            a = ops.pmf_to_quantized_cdf(p)
            b = oracle.pmf_to_quantized_cdf(p)
            assert a[0] == 0 and a[-1] == 1 << 16 and np.all(np.diff(a) >= 1)
            assert np.array_equal(a, b)


def test_ctypes_structs_match_the_header_as_compiled_by_a_c_compiler(tmp_path):
    """include/pcc_geo.h is a C header: a C compiler must accept it, and the sizes / field offsets it gives the ABI structs must be
    the ones the ctypes mirror (pcc_geo_cnn_v2_amd/_lib.py) uses -- a silent mismatch would shift every pointer behind it."""
    import shutil
    import subprocess
    cc = shutil.which('gcc') or shutil.which('cc')
    if cc is None:
        pytest.skip('no C compiler')
    structs = {'pcc_conv_desc': L.ConvDesc, 'pcc_cdf_table': L.CdfTable, 'pcc_codec_desc': L.CodecDesc, 'pcc_symbol_io': L.SymbolSink}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{os.path.join(ROOT, "include", "pcc_geo.h")}"', 'int main(void) {']
    for name, cls in structs.items():
        lines.append(f'  printf("{name} %zu", sizeof({name}));')
        for field, _ in cls._fields_:
            lines.append(f'  printf(" %zu", offsetof({name}, {field}));')
        lines.append('  printf("\\n");')
    lines += ['  printf("abi %d\\n", PCC_ABI_VERSION);', '  return 0;', '}']
    src = tmp_path / 'layout.c'
    src.write_text('\n'.join(lines))
    exe = tmp_path / 'layout'
    subprocess.run([cc, '-std=c99', '-Wall', '-Werror', '-o', str(exe), str(src)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    for line in out[:-1]:
        name, size, *offs = line.split()
        cls = structs[name]
        assert int(size) == C.sizeof(cls), f'{name}: header {size} bytes, ctypes {C.sizeof(cls)}'
        assert [int(o) for o in offs] == [getattr(cls, f).offset for f, _ in cls._fields_], f'{name}: field offsets differ'
    assert out[-1] == f'abi {L.ABI_VERSION}'
