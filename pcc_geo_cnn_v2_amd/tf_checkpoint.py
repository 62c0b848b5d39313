"""TF1 checkpoint (TensorBundle V2: `<prefix>.index` + `<prefix>.data-00000-of-0000N`) importer -- SURVEY.md §8f row 1.

The reference restores its weights with `tf.train.Saver().restore(sess, tf.train.latest_checkpoint(dir))`
(/root/reference/src/compress_octree.py:82,90-92; decompress_octree.py:41,53-55).  This module reads such a
checkpoint WITHOUT TensorFlow and converts it to this package's `model.npz`.

Format (restated from the published TensorFlow / LevelDB sources; TensorFlow is not installable here and the reference
ships no checkpoint, so the reader is exercised against an independent writer in tests/ -- "format parity unpinned"):
  * `.index` is a LevelDB table ("SSTable"): data blocks, meta-index block, index block, 48-byte footer
    (two varint64 block handles, padding, magic 0xdb4775248b80fb57).  A block is a run of prefix-compressed entries
    (varint32 shared, non_shared, value_len; key delta; value), a uint32 restart array and its length; every block is
    followed by a 1-byte compression type (0 none, 1 snappy) and a masked crc32c.
  * key "" -> BundleHeaderProto {1: num_shards, 2: endianness, 3: version}; every other key is a variable name ->
    BundleEntryProto {1: dtype, 2: TensorShapeProto{2: dim{1: size}}, 3: shard_id, 4: offset, 5: size, 6: crc32c}.
  * tensors are raw little-endian bytes at [offset, offset+size) of shard `shard_id`.

Variable names are Keras auto-names whose scoping cannot be verified here, so the conversion is
STRUCTURAL: conv layers are recognised by their leaf name `conv3d[_N]` / `conv3d_transpose[_N]`, ordered by N (Keras
numbers layers in construction order: analysis, synthesis, hyper-analysis, hyper-synthesis -- model_types.py:371-376),
optimizer slots are dropped, and every tensor shape is checked against the model it is loaded into.
"""
import argparse
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}


# ------------------------------------------------------------------------------------------------ primitives
def _varint(buf, pos):
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7


def _proto_fields(buf):
    """Yields (field_number, wire_type, value) of one protobuf message (varint / 64-bit / bytes / 32-bit)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fn, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from('<Q', buf, pos)[0]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + ln])
            pos += ln
        elif wt == 5:
            v = struct.unpack_from('<I', buf, pos)[0]
            pos += 4
        else:
            raise ValueError(f'unsupported protobuf wire type {wt}')
        yield fn, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _snappy_decompress(buf):
    """Raw snappy block format (LevelDB compression type 1)."""
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], 'little')
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 2], 'little')
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], 'little')
            pos += 4
        for _ in range(ln):
            out.append(out[-off])
    assert len(out) == n, 'corrupt snappy block'
    return bytes(out)


_CRC_TABLE = None


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), table driven; used only when `verify=True`."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tab = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ (0x82F63B78 if c & 1 else 0)
            tab.append(c)
        _CRC_TABLE = tab
    c = crc ^ 0xFFFFFFFF
    tab = _CRC_TABLE
    for b in data:
        c = tab[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------ table reader
def _read_block(data, offset, size, verify=False):
    raw = data[offset:offset + size]
    ctype = data[offset + size]
    if verify:
        stored = struct.unpack_from('<I', data, offset + size + 1)[0]
        assert stored == mask_crc(crc32c(data[offset:offset + size + 1])), 'index block checksum mismatch'
    if ctype == 1:
        raw = _snappy_decompress(raw)
    elif ctype != 0:
        raise ValueError(f'unknown block compression type {ctype}')
    return raw


def _block_entries(block):
    n_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * n_restarts
    pos, key = 0, b''
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=False):
    """All (key, value) pairs of a LevelDB table file, in key order."""
    data = open(path, 'rb').read()
    assert len(data) >= 48, f'{path}: too short for a table'
    footer = data[-48:]
    assert struct.unpack_from('<Q', footer, 40)[0] == TABLE_MAGIC, f'{path}: not a TensorBundle index (bad magic)'
    pos = 0
    _, pos = _varint(footer, pos)          # meta-index handle (unused)
    _, pos = _varint(footer, pos)
    ioff, pos = _varint(footer, pos)
    isize, pos = _varint(footer, pos)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        boff, p = _varint(handle, 0)
        bsize, p = _varint(handle, p)
        out.extend(_block_entries(_read_block(data, boff, bsize, verify)))
    return out


def read_index(prefix, verify=False):
    """-> (header dict, {name: dict(dtype, shape, shard_id, offset, size, crc32c)})"""
    header, entries = {}, {}
    for key, val in read_table(prefix + '.index', verify):
        f = {}
        shape = []
        for fn, wt, v in _proto_fields(val):
            if key == b'':
                header[{1: 'num_shards', 2: 'endianness', 3: 'version'}.get(fn, fn)] = v
            elif fn == 2 and wt == 2:
                for fn2, wt2, dim in _proto_fields(v):
                    if fn2 == 2 and wt2 == 2:
                        size = 0
                        for fn3, _, v3 in _proto_fields(dim):
                            if fn3 == 1:
                                size = _signed64(v3)
                        shape.append(size)
            else:
                f[fn] = v
        if key != b'':
            entries[key.decode()] = dict(dtype=f.get(1, 0), shape=tuple(shape), shard_id=f.get(3, 0), offset=f.get(4, 0),
                                         size=f.get(5, 0), crc32c=f.get(6, 0), sliced=7 in f)
    header.setdefault('num_shards', 1)
    assert header.get('endianness', 0) == 0, 'big-endian bundles are not supported'
    return header, entries


def load_checkpoint(prefix, verify=False):
    """{variable name: np.ndarray} of every tensor in the bundle."""
    header, entries = read_index(prefix, verify)
    shards = {}
    out = {}
    for name, e in entries.items():
        assert not e['sliced'], f'{name}: partitioned (sliced) variables are not supported'
        assert e['dtype'] in DTYPES, f'{name}: unsupported dtype enum {e["dtype"]}'
        sid = e['shard_id']
        if sid not in shards:
            shards[sid] = np.memmap(f'{prefix}.data-{sid:05d}-of-{header["num_shards"]:05d}', dtype=np.uint8, mode='r')
        raw = shards[sid][e['offset']:e['offset'] + e['size']]
        if verify:
            assert mask_crc(crc32c(bytes(raw))) == e['crc32c'], f'{name}: tensor checksum mismatch'
        dt = np.dtype(DTYPES[e['dtype']])
        assert e['size'] == int(np.prod(e['shape'], dtype=np.int64)) * dt.itemsize, f'{name}: size does not match shape'
        out[name] = np.frombuffer(bytes(raw), dtype=dt).reshape(e['shape']).copy()
    return out


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: parses the `checkpoint` state file; falls back to the newest *.index."""
    state = os.path.join(checkpoint_dir, 'checkpoint')
    if os.path.exists(state):
        m = re.search(r'model_checkpoint_path:\s*"([^"]+)"', open(state).read())
        if m:
            p = m.group(1)
            return p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)
    idx = sorted((f for f in os.listdir(checkpoint_dir) if f.endswith('.index')),
                 key=lambda f: os.path.getmtime(os.path.join(checkpoint_dir, f)))
    return os.path.join(checkpoint_dir, idx[-1][:-len('.index')]) if idx else None


# ------------------------------------------------------------------------------------------------ conversion
_SLOT = re.compile(r'(/Adam(_\d+)?$)|(^beta[12]_power)|(^global_step$)|(/ExponentialMovingAverage$)|(/Momentum$)|(/RMSProp(_\d+)?$)')
_CONV = re.compile(r'(^|/)(conv3d(_transpose)?)(_(\d+))?/(kernel|bias)$')
_EB = re.compile(r'(^|/)entropy_bottleneck(_\d+)?/(matrix_\d+|bias_\d+|factor_\d+|quantiles|quantized_cdf|cdf_length)$')
_GC = re.compile(r'(^|/)gaussian_conditional(_\d+)?/(quantized_cdf|cdf_length)$')


def convert_variables(variables, model):
    """TF variable dict -> this package's parameter dict for `model.set_weights` (model: a built CompressionModelV1/V2).
    Raises AssertionError with the offending names when the checkpoint does not fit the model."""
    convs = {False: {}, True: {}}          # transposed? -> {N: {'kernel':…, 'bias':…}}
    eb, gc, ignored = {}, {}, []
    for name, arr in variables.items():
        if _SLOT.search(name):
            continue
        m = _CONV.search(name)
        if m:
            convs[m.group(3) is not None].setdefault(int(m.group(5) or 0), {})[m.group(6)] = arr
            continue
        m = _EB.search(name)
        if m:
            eb[m.group(3)] = arr
            continue
        m = _GC.search(name)
        if m:
            gc[m.group(3)] = arr
            continue
        ignored.append(name)
    fwd = [convs[False][k] for k in sorted(convs[False])]
    tr = [convs[True][k] for k in sorted(convs[True])]
    out = {}
    want = {k: v for k, v in model.get_weights().items()}
    # construction order of the reference: analysis, synthesis, hyper_analysis, hyper_synthesis (model_types.py:371-376)
    pools = {'analysis': fwd, 'hyper_analysis': fwd, 'synthesis': tr, 'hyper_synthesis': tr}
    taken = {id(fwd): 0, id(tr): 0}
    for prefix in ('analysis', 'synthesis', 'hyper_analysis', 'hyper_synthesis'):
        n = len({k.split('/')[1] for k in want if k.startswith(prefix + '/')})
        pool = pools[prefix]
        for i in range(n):
            j = taken[id(pool)]
            assert j < len(pool), f'checkpoint has too few {"conv3d_transpose" if pool is tr else "conv3d"} layers for {prefix}/{i}'
            layer = pool[j]
            taken[id(pool)] += 1
            for part in ('kernel', 'bias'):
                key = f'{prefix}/{i}/{part}'
                if key in want:
                    assert part in layer, f'{key}: the checkpoint layer #{j} has no {part}'
                    assert tuple(layer[part].shape) == tuple(want[key].shape), \
                        f'{key}: checkpoint shape {layer[part].shape} != model shape {want[key].shape}'
                    out[key] = np.ascontiguousarray(layer[part], np.float32)
                else:
                    assert part not in layer, f'{prefix}/{i}: unexpected {part} in the checkpoint (layer order mismatch?)'
    # a decoder-only model (model.decompress(), model_types.py:297-309,393-411) has no analysis / hyper-analysis transform:
    # the checkpoint's conv3d* layers are simply not needed; the transposed pool keeps its order (synthesis, hyper-synthesis)
    has_analysis = any(k.startswith('analysis/') for k in want)
    assert (taken[id(fwd)] == len(fwd) or not has_analysis) and taken[id(tr)] == len(tr), \
        f'unused conv layers in the checkpoint: {len(fwd) - taken[id(fwd)]} conv3d, {len(tr) - taken[id(tr)]} conv3d_transpose'
    # factorized prior: parameters as stored; the integer tables of the checkpoint win over recomputed ones (bit-exact rate)
    for k, v in eb.items():
        if k in ('quantized_cdf', 'cdf_length'):
            continue
        key = f'entropy_bottleneck/{k}'
        assert key in want and tuple(v.shape) == tuple(want[key].shape), f'{key}: shape {v.shape} does not fit the model'
        out[key] = np.ascontiguousarray(v, np.float32)
    missing = [k for k in want if k.startswith('entropy_bottleneck/') and k.split('/')[1] not in ('quantized_cdf', 'cdf_length', 'offset') and k not in out]
    assert not missing, f'entropy bottleneck parameters missing from the checkpoint: {missing}'
    if 'quantized_cdf' in eb and 'cdf_length' in eb:
        q = out['entropy_bottleneck/quantiles']
        minima = np.maximum(np.ceil(q[:, 0, 1] - q[:, 0, 0]).astype(np.int32), 0)      # entropy_models.EntropyBottleneck._build
        out['entropy_bottleneck/quantized_cdf'] = eb['quantized_cdf'].astype(np.int32)
        out['entropy_bottleneck/cdf_length'] = eb['cdf_length'].astype(np.int32)
        out['entropy_bottleneck/offset'] = (-minima).astype(np.int32)
    if 'quantized_cdf' in gc and 'gaussian_conditional/quantized_cdf' in want:
        # Tables as stored (bit-exact rate).  Their width depends on the tail_mass the checkpoint was built with (tfc default
        # 2**-8 -> 64 x 1481; 1e-9 -> 64 x 3133), so any width is accepted: a row holds pmf_length = 2*center + 1
        # symbols plus the overflow bin, cdf_length = pmf_length + 2, and the offset is -center (patch_gaussian_conditional.py:63,96,118)
        cl = gc['cdf_length'].astype(np.int32)
        assert gc['quantized_cdf'].shape[0] == want['gaussian_conditional/quantized_cdf'].shape[0] == len(cl), \
            f'gaussian_conditional: {gc["quantized_cdf"].shape[0]} table rows in the checkpoint, model has {want["gaussian_conditional/quantized_cdf"].shape[0]} scales'
        assert np.all((cl - 3) % 2 == 0) and gc['quantized_cdf'].shape[1] >= int(cl.max()), 'gaussian_conditional: inconsistent cdf_length'
        out['gaussian_conditional/quantized_cdf'] = gc['quantized_cdf'].astype(np.int32)
        out['gaussian_conditional/cdf_length'] = cl
        out['gaussian_conditional/offset'] = (-((cl - 3) // 2)).astype(np.int32)
    return out, ignored


def import_checkpoint(checkpoint_dir, model, verify=False):
    prefix = latest_checkpoint(checkpoint_dir)
    assert prefix is not None and os.path.exists(prefix + '.index'), f'Checkpoint {checkpoint_dir} was not found'
    params, ignored = convert_variables(load_checkpoint(prefix, verify), model)
    model.set_weights(params)
    return prefix, ignored


def main():
    from .model_configs import ModelConfigType
    p = argparse.ArgumentParser(prog='import_tf_checkpoint', description='TF1 checkpoint -> model.npz (no TensorFlow needed)')
    p.add_argument('--checkpoint_dir', required=True, help='directory holding `checkpoint`, *.index, *.data-*')
    p.add_argument('--model_config', required=True, help='c1 | c2 | c3 | c3p (and aliases)')
    p.add_argument('--output_dir', required=True)
    p.add_argument('--resolution', type=int, default=64)
    p.add_argument('--verify', action='store_true', help='check every crc32c')
    args = p.parse_args()
    model = ModelConfigType[args.model_config].build()
    model.compress([1] + ([args.resolution] * 3 + [1] if model.data_format == 'channels_last' else [1] + [args.resolution] * 3))
    prefix, ignored = import_checkpoint(args.checkpoint_dir, model, args.verify)
    model.save_checkpoint(args.output_dir)
    print(f'imported {prefix} -> {args.output_dir}/model.npz' + (f' (ignored: {ignored})' if ignored else ''))


if __name__ == '__main__':
    main()
