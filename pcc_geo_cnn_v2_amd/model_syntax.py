"""`.ply.bin` container -- mirrors /root/reference/src/model_syntax.py:4-58 byte for byte.

Layout (little endian): u16 resolution, u8 level, u16 n_blocks, u8 n_strings, u16 n_binstr,
u8[n_binstr] binstr, then per block: u8 threshold index, per string: u16 length + bytes.
The whole file is gzip'd by the CLI (src/compress_octree.py:112).

Out-of-range header fields: under the reference's pinned numpy 1.18 the casts in `to_bytes` wrap
silently, its asserts (model_syntax.py:7-8) are vacuous, and the damage only surfaces as an
AssertionError in `load_compressed_file` (pinned by src/test_model_syntax.py:22-30).  `strict=False`
(default) reproduces exactly that; `strict=True` (used by our CLI) raises AssertionError at save time.
"""
import logging

import numpy as np

logger = logging.getLogger(__name__)


def to_bytes(x, dtype, strict=False):
    iinfo = np.iinfo(dtype)
    x64 = np.array(x, dtype=np.int64)
    bad = np.any(x64 > iinfo.max) or np.any(x64 < iinfo.min)
    if bad:
        assert not strict, f'Overflow/underflow {x} {iinfo}'
        logger.warning('value %s does not fit %s: wrapping like numpy 1.18 did for the reference', x, dtype)
    return (x64 & ((1 << iinfo.bits) - 1)).astype(np.uint64).astype(dtype).tobytes()


def scalar_to_bytes(x, dtype, strict=False):
    return to_bytes([x], dtype, strict)


def read_from_buffer(f, n, dtype):
    return np.frombuffer(f.read(int(np.dtype(dtype).itemsize * n)), dtype=dtype)


def save_compressed_file(binstr, data_b_list, resolution, octree_level, strict=False):
    """Saves an octree partitioned point cloud and its partition bitstreams as an unified bitstream"""
    ret = [scalar_to_bytes(resolution, np.uint16, strict), scalar_to_bytes(octree_level, np.uint8, strict),
           scalar_to_bytes(len(data_b_list), np.uint16, strict),
           scalar_to_bytes(len(data_b_list[0][0]), np.uint8, strict),
           scalar_to_bytes(len(binstr), np.uint16, strict), to_bytes(binstr, np.uint8, strict)]
    for strings, best_threshold_idx in data_b_list:
        ret.append(scalar_to_bytes(best_threshold_idx, np.uint8, strict))
        for s in strings:
            ret.append(scalar_to_bytes(len(s), np.uint16, strict))
            ret.append(bytes(s))
    return b''.join(ret)


def load_compressed_file(f):
    """Loads an octree partitioned point cloud unified bitstream"""
    blocks = []
    resolution = read_from_buffer(f, 1, np.uint16)[0]
    level = read_from_buffer(f, 1, np.uint8)[0]
    n_blocks = read_from_buffer(f, 1, np.uint16)[0]
    n_strings = read_from_buffer(f, 1, np.uint8)[0]
    n_binstr = read_from_buffer(f, 1, np.uint16)[0]
    binstr = read_from_buffer(f, n_binstr, np.uint8)
    for _ in range(n_blocks):
        best_threshold_idx = read_from_buffer(f, 1, np.uint8)[0]
        strings = []
        for _i in range(n_strings):
            n_bytes = read_from_buffer(f, 1, np.uint16)[0]
            strings.append(f.read(int(n_bytes)))
        blocks.append((strings, best_threshold_idx))
    file_end = f.read()
    assert file_end == b'', f'File not read completely file_end {file_end[:64]}'
    return resolution, level, binstr, blocks
