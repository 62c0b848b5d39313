"""Test helper: writes a TensorFlow V2 checkpoint (TensorBundle: LevelDB table index + raw data shard) from the published
format description, independently of pcc_geo_cnn_v2_amd/tf_checkpoint.py's reader (no shared code)."""
import struct

import numpy as np

_DT = {np.dtype(np.float32): 1, np.dtype(np.int32): 3, np.dtype(np.int64): 9, np.dtype(np.float64): 2}


def _vi(n):
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _crc32c(data):
    c = 0xFFFFFFFF
    for b in data:
        c ^= b
        for _ in range(8):
            c = (c >> 1) ^ (0x82F63B78 & -(c & 1))
    return c ^ 0xFFFFFFFF


def _mask(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


def _block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b''
    for i, (k, v) in enumerate(entries):
        shared = 0
        if i % restart_interval == 0:
            restarts.append(len(out))
        else:
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _vi(shared) + _vi(len(k) - shared) + _vi(len(v)) + k[shared:] + v
        prev = k
    if not restarts:
        restarts = [0]
    for r in restarts:
        out += struct.pack('<I', r)
    out += struct.pack('<I', len(restarts))
    return bytes(out)


def write_bundle(prefix, variables, entries_per_block=5, checksums=True):
    names = sorted(variables, key=lambda s: s.encode())
    data, meta = bytearray(), []
    for n in names:
        a = np.asarray(variables[n], order="C")        # (ascontiguousarray would turn 0-d into 1-d)
        raw = a.tobytes()
        shape = b''.join(b'\x12' + _vi(len(d)) + d for d in (b'\x08' + _vi(int(s)) for s in a.shape))
        e = b'\x08' + _vi(_DT[a.dtype]) + b'\x12' + _vi(len(shape)) + shape
        if len(data):
            e += b'\x20' + _vi(len(data))
        e += b'\x28' + _vi(len(raw))
        e += b'\x35' + struct.pack('<I', _mask(_crc32c(raw)) if checksums else 0)
        meta.append((n.encode(), e))
        data += raw
    open(prefix + '.data-00000-of-00001', 'wb').write(bytes(data))
    header = b'\x08\x01' + b'\x1a\x02\x08\x01'          # num_shards = 1, version {producer = 1}
    entries = [(b'', header)] + meta
    f = bytearray()
    index = []

    def put(block):
        off = len(f)
        f.extend(block)
        f.extend(b'\x00' + struct.pack('<I', _mask(_crc32c(block + b'\x00'))))
        return _vi(off) + _vi(len(block))

    for i in range(0, len(entries), entries_per_block):
        chunk = entries[i:i + entries_per_block]
        h = put(_block(chunk))
        index.append((chunk[-1][0] + b'\x00', h))         # any separator >= last key of the block
    mh = put(_block([]))
    ih = put(_block(index, restart_interval=1))
    footer = mh + ih
    footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57)
    f.extend(footer)
    open(prefix + '.index', 'wb').write(bytes(f))
