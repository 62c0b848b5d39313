#!/usr/bin/env python
"""Hand-assembles a small TensorFlow V2 checkpoint (TensorBundle) byte by byte from the published format constants and
writes it, with the tensors it encodes, under tests/golden/tf_bundle/.  It shares NO code with the reader under test
(pcc_geo_cnn_v2_amd/tf_checkpoint.py) or with tests/tf_bundle_writer.py: every constant below is spelled out here.

Published layout (tensorflow/core/util/tensor_bundle, tensorflow/core/lib/io/{table,block,format}, leveldb table_format.md,
snappy format_description.txt):

  <prefix>.index : LevelDB-style table
      data block(s)   = entries [varint32 shared][varint32 non_shared][varint32 value_len][key suffix][value] ...,
                        then uint32le restart offsets, then uint32le restart count
      block trailer   = 1 byte compression type (0 = none, 1 = snappy) + uint32le masked CRC-32C of (block bytes + type byte)
                        masked = rotr(crc, 15) + 0xa282ead8  (mod 2^32)
      metaindex block = an empty block; index block = one entry per data block: key >= last key of the block,
                        value = BlockHandle (varint64 offset, varint64 size of the block WITHOUT its 5-byte trailer)
      footer (48 B)   = metaindex handle, index handle, zero padding to 40 bytes, magic 0xdb4775248b80fb57 (uint64le)
      key ""          -> BundleHeaderProto  { 1: num_shards (varint)  2: endianness (0 = little)  3: VersionDef { 1: producer } }
      key <name>      -> BundleEntryProto   { 1: dtype  2: TensorShapeProto { 2: Dim { 1: size } ... }  3: shard_id  4: offset
                                              5: size  6: crc32c (fixed32, masked CRC-32C of the tensor bytes) }
                         dtype enum: DT_FLOAT = 1, DT_INT32 = 3, DT_INT64 = 9
  <prefix>.data-0000S-of-0000N : the raw little-endian tensor bytes of shard S, back to back
  checkpoint      : text proto, model_checkpoint_path: "<basename>"

What the fixture exercises: two data shards, three data blocks in the index (one stored snappy-compressed with literal,
1-byte-offset copy and 2-byte-offset copy elements), prefix-compressed keys with a restart interval of 2, a scalar, a 5-D
kernel, int32 / int64 tensors, an optimizer slot and a step counter (which the importer must drop), zero-size dimension.
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'tf_bundle')
PREFIX = 'model.ckpt-4242'


def varint(n):
    b = bytearray()
    while n >= 0x80:
        b.append((n & 0x7F) | 0x80)
        n >>= 7
    b.append(n)
    return bytes(b)


def crc32c_bitwise(data):
    """CRC-32C (Castagnoli): reflected polynomial 0x82F63B78, init / xorout 0xFFFFFFFF, one bit at a time."""
    crc = 0xFFFFFFFF
    for byte in data:
        crc ^= byte
        for _ in range(8):
            crc = (crc >> 1) ^ 0x82F63B78 if crc & 1 else crc >> 1
    return crc ^ 0xFFFFFFFF


def masked(crc):
    return (((crc >> 15) | ((crc << 17) & 0xFFFFFFFF)) + 0xA282EAD8) & 0xFFFFFFFF


def pb_varint_field(num, value):
    return varint((num << 3) | 0) + varint(value)


def pb_bytes_field(num, payload):
    return varint((num << 3) | 2) + varint(len(payload)) + payload


def pb_fixed32_field(num, value):
    return varint((num << 3) | 5) + struct.pack('<I', value)


def shape_proto(shape):
    out = b''
    for d in shape:
        out += pb_bytes_field(2, pb_varint_field(1, d))
    return out


def entry_proto(dtype_enum, shape, shard, offset, size, crc):
    msg = pb_varint_field(1, dtype_enum) + pb_bytes_field(2, shape_proto(shape))
    if shard:
        msg += pb_varint_field(3, shard)          # proto3: zero-valued fields are omitted on the wire
    if offset:
        msg += pb_varint_field(4, offset)
    msg += pb_varint_field(5, size) if size else b''
    return msg + pb_fixed32_field(6, crc)


def block(entries, restart_every):
    body, restarts, prev = bytearray(), [], b''
    for i, (key, value) in enumerate(entries):
        if i % restart_every == 0:
            restarts.append(len(body))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(key)) and prev[shared] == key[shared]:
                shared += 1
        body += varint(shared) + varint(len(key) - shared) + varint(len(value)) + key[shared:] + value
        prev = key
    for r in restarts:
        body += struct.pack('<I', r)
    body += struct.pack('<I', len(restarts))
    return bytes(body)


def snappy_handmade(raw):
    """Snappy-compress `raw` with the three element kinds the fixture wants to exercise.  Strategy: literal for the first
    bytes, then greedy longest-match search (window = everything before), emitting copy-1 (len 4..11, offset < 2048) or
    copy-2 (len <= 64) elements, literals otherwise."""
    out = bytearray(varint(len(raw)))
    lit = bytearray()

    def flush():
        nonlocal lit
        while lit:
            chunk, lit = lit[:60], lit[60:]
            out.append((len(chunk) - 1) << 2)      # literal with length <= 60 encoded in the tag
            out.extend(chunk)

    i = 0
    while i < len(raw):
        best_len, best_off = 0, 0
        if i >= 4:
            for j in range(max(0, i - 60000), i):
                ln = 0
                while i + ln < len(raw) and ln < 64 and raw[j + ln] == raw[i + ln]:
                    ln += 1                        # (overlapping copies are legal in snappy: j + ln may run past i)
                if ln > best_len:
                    best_len, best_off = ln, i - j
        if best_len >= 4:
            flush()
            if best_len <= 11 and best_off < 2048:
                out.append(0b01 | ((best_len - 4) << 2) | ((best_off >> 8) << 5))
                out.append(best_off & 0xFF)
            else:
                out.append(0b10 | ((best_len - 1) << 2))
                out.extend(struct.pack('<H', best_off))
            i += best_len
        else:
            lit.append(raw[i])
            i += 1
    flush()
    return bytes(out)


def main():
    os.makedirs(OUT, exist_ok=True)
    rng = np.random.default_rng(4242)
    # name -> (array, shard).  Keras-style names in the reference's construction order + Saver extras.
    tensors = [
        ('analysis_transform_v1/conv3d/bias', rng.standard_normal(4).astype('<f4'), 0),
        ('analysis_transform_v1/conv3d/kernel', rng.standard_normal((3, 3, 3, 1, 4)).astype('<f4'), 0),
        ('analysis_transform_v1/conv3d/kernel/Adam', np.zeros((3, 3, 3, 1, 4), '<f4'), 1),
        ('analysis_transform_v1/conv3d_1/kernel', (np.arange(3 * 3 * 3 * 4 * 2, dtype='<f4') / 7).reshape(3, 3, 3, 4, 2), 1),
        ('entropy_bottleneck/cdf_length', np.array([5, 7], '<i4'), 0),
        ('entropy_bottleneck/quantized_cdf', np.array([[0, 100, 60000, 65536, 0, 0, 0], [0, 1, 2, 3, 4, 5, 65536]], '<i4'), 1),
        ('entropy_bottleneck/quantiles', np.array([[[-1.5, 0.25, 2.0]], [[-3.0, 0.0, 3.0]]], '<f4'), 0),
        ('global_step', np.array(4242, '<i8'), 0),
        ('synthesis_transform_v1/conv3d_transpose/bias', np.array([0.47], '<f4'), 1),
        ('synthesis_transform_v1/conv3d_transpose/kernel', rng.standard_normal((3, 3, 3, 1, 2)).astype('<f4'), 1),
        ('unused/empty', np.zeros((0, 3), '<f4'), 0),
        ('unused/scalar', np.array(2.5, '<f4'), 1),
    ]
    tensors.sort(key=lambda t: t[0].encode())          # table keys are sorted bytewise
    dtype_enum = {'<f4': 1, '<i4': 3, '<i8': 9}
    shards = [bytearray(), bytearray()]
    kv = [(b'', pb_varint_field(1, 2) + pb_bytes_field(3, pb_varint_field(1, 1)))]   # header: num_shards = 2, little endian (0, omitted), version {producer: 1}
    for name, arr, shard in tensors:
        raw = arr.tobytes()
        off = len(shards[shard])
        shards[shard] += raw
        kv.append((name.encode(), entry_proto(dtype_enum[arr.dtype.str], arr.shape, shard, off, len(raw), masked(crc32c_bitwise(raw)))))
    # three data blocks: 5 + 4 + 4 entries; the middle one snappy-compressed
    groups = [kv[:5], kv[5:9], kv[9:]]
    file_bytes, index_entries = bytearray(), []
    for gi, g in enumerate(groups):
        raw = block(g, restart_every=2)
        ctype = 1 if gi == 1 else 0
        stored = snappy_handmade(raw) if ctype else raw
        if ctype:
            kinds = set()
            p = len(varint(len(raw)))
            # (self-check of the hand-made stream: all three element kinds present)
            q = p
            while q < len(stored):
                t = stored[q] & 3
                kinds.add(t)
                if t == 0:
                    q += 1 + (stored[q] >> 2) + 1
                elif t == 1:
                    q += 2
                else:
                    q += 3
            assert kinds == {0, 1, 2}, kinds
        handle = varint(len(file_bytes)) + varint(len(stored))
        file_bytes += stored + bytes([ctype]) + struct.pack('<I', masked(crc32c_bitwise(stored + bytes([ctype]))))
        index_entries.append((g[-1][0] + b'\x00' if gi < 2 else g[-1][0], handle))      # separator >= last key of the block
    meta = block([], restart_every=16)
    meta_handle = varint(len(file_bytes)) + varint(len(meta))
    file_bytes += meta + b'\x00' + struct.pack('<I', masked(crc32c_bitwise(meta + b'\x00')))
    idx = block(index_entries, restart_every=1)
    idx_handle = varint(len(file_bytes)) + varint(len(idx))
    file_bytes += idx + b'\x00' + struct.pack('<I', masked(crc32c_bitwise(idx + b'\x00')))
    footer = meta_handle + idx_handle
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xDB4775248B80FB57)
    file_bytes += footer
    with open(os.path.join(OUT, PREFIX + '.index'), 'wb') as f:
        f.write(bytes(file_bytes))
    for s in range(2):
        with open(os.path.join(OUT, f'{PREFIX}.data-{s:05d}-of-00002'), 'wb') as f:
            f.write(bytes(shards[s]))
    with open(os.path.join(OUT, 'checkpoint'), 'w') as f:
        f.write(f'model_checkpoint_path: "{PREFIX}"\nall_model_checkpoint_paths: "{PREFIX}"\n')
    np.savez(os.path.join(OUT, 'expected.npz'), **{name.replace('/', '|'): arr for name, arr, _ in tensors})
    print(f'wrote {len(file_bytes)} index bytes, shards {[len(s) for s in shards]} bytes, {len(tensors)} tensors')


if __name__ == '__main__':
    main()
