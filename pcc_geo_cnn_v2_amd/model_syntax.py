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
import struct

import numpy as np

logger = logging.getLogger(__name__)


_HEADER = struct.Struct('<HBHBH')        # resolution, octree level, n_blocks, strings per block, len(binstr)
_U16 = struct.Struct('<H')


def _field(value, bits, strict, what):
    """An unsigned `bits`-wide field.  A value that does not fit wraps modulo 2**bits -- what the reference's numpy 1.18 casts
    did silently -- or, with strict=True, is an AssertionError here at save time."""
    value = int(value)
    if not 0 <= value < (1 << bits):
        assert not strict, f'Overflow/underflow: {what} = {value} does not fit {bits} bits'
        logger.warning('%s = %s does not fit %d bits: wrapping like numpy 1.18 did for the reference', what, value, bits)
        value &= (1 << bits) - 1
    return value


def to_bytes(x, dtype, strict=False):
    """Little-endian bytes of a sequence of unsigned integers of numpy `dtype` (uint8 / uint16), see `_field`."""
    width = np.dtype(dtype).itemsize
    return b''.join(_field(v, 8 * width, strict, 'value').to_bytes(width, 'little') for v in np.asarray(x).reshape(-1).tolist())


def scalar_to_bytes(x, dtype, strict=False):
    return to_bytes([x], dtype, strict)


def save_compressed_file(binstr, data_b_list, resolution, octree_level, strict=False):
    """Serialises an octree-partitioned cloud: the partition bits `binstr` and, per block, (strings, threshold index) -- byte
    for byte the file of the reference's writer (model_syntax.py:20-35; pinned by tests/golden/model_syntax.npz)."""
    out = bytearray(_HEADER.pack(_field(resolution, 16, strict, 'resolution'), _field(octree_level, 8, strict, 'octree level'),
                                 _field(len(data_b_list), 16, strict, 'number of blocks'),
                                 _field(len(data_b_list[0][0]), 8, strict, 'strings per block'),
                                 _field(len(binstr), 16, strict, 'len(binstr)')))
    out += bytes(_field(b, 8, strict, 'binstr entry') for b in binstr)
    for strings, threshold_idx in data_b_list:
        out.append(_field(threshold_idx, 8, strict, 'threshold index'))
        for payload in strings:
            out += _U16.pack(_field(len(payload), 16, strict, 'string length'))
            out += payload
    return bytes(out)


class _Cursor:
    """Bounds-checked reader over the whole container held in memory."""

    def __init__(self, raw):
        self.raw, self.pos = memoryview(raw), 0

    def take(self, n):
        end = self.pos + n
        if end > len(self.raw):
            # the reference indexes an empty np.frombuffer result here (model_syntax.py:41-52): IndexError
            raise IndexError(f'compressed file truncated: need {end} bytes, have {len(self.raw)}')
        chunk, self.pos = self.raw[self.pos:end], end
        return chunk

    def unpack(self, fmt):
        return fmt.unpack(self.take(fmt.size))


def load_compressed_file(f):
    """Parses the container written by save_compressed_file (layout in the module docstring; the reference's reader is
    model_syntax.py:38-58).  `f`: binary file object.  Returns (resolution, level, binstr uint8 array, [(strings, threshold
    index)]); bytes left over after the last block are an AssertionError like in the reference."""
    cur = _Cursor(f.read())
    resolution, level, n_blocks, per_block, n_binstr = cur.unpack(_HEADER)
    binstr = np.frombuffer(cur.take(n_binstr), dtype=np.uint8)
    blocks = []
    while len(blocks) < n_blocks:
        threshold_idx = cur.take(1)[0]
        blocks.append(([bytes(cur.take(cur.unpack(_U16)[0])) for _ in range(per_block)], threshold_idx))
    rest = bytes(cur.raw[cur.pos:])
    assert not rest, f'File not read completely file_end {rest[:64]}'
    return resolution, level, binstr, blocks


# ---- the gzip wrapper of the CLIs (src/compress_octree.py:112, src/decompress_octree.py:61) with a codec-numerics tag ----------------
# The container above is the reference's, byte for byte, and stays that way.  What it cannot say is which kernel family computed
# sigma-hat on the encoder (include/pcc_geo.h, "codec numerics"): a decoder on another family flips scale indexes and the range decoder
# desynchronises silently.  The tag rides in the gzip member header's FCOMMENT field (RFC 1952): `gzip.open` -- the reference's reader --
# skips it, so a tagged file is still a valid input for the reference, and a file written by the reference simply has no tag.
def write_tagged_gzip(path, payload, tag):
    import zlib
    comment = tag.encode('latin-1')
    assert b'\0' not in comment
    deflate = zlib.compressobj(9, zlib.DEFLATED, -15)
    body = deflate.compress(payload) + deflate.flush()
    with open(path, 'wb') as fh:
        fh.write(b'\x1f\x8b\x08\x10' + struct.pack('<IBB', 0, 2, 255) + comment + b'\0')      # FLG = FCOMMENT, MTIME 0, XFL 2, OS unknown
        fh.write(body)
        fh.write(struct.pack('<II', zlib.crc32(payload) & 0xffffffff, len(payload) & 0xffffffff))


def read_gzip_tag(path):
    """The FCOMMENT of the first gzip member, or None (no comment / not a gzip file)."""
    with open(path, 'rb') as fh:
        head = fh.read(10)
        if len(head) < 10 or head[:3] != b'\x1f\x8b\x08':
            return None
        flg = head[3]
        if flg & 0x04:                                  # FEXTRA
            n, = struct.unpack('<H', fh.read(2))
            fh.read(n)

        def cstring():
            out = bytearray()
            while True:
                c = fh.read(1)
                if not c or c == b'\0':
                    return bytes(out)
                out += c
        if flg & 0x08:                                  # FNAME
            cstring()
        return cstring().decode('latin-1') if flg & 0x10 else None


def check_numerics_tag(tag, expected, ignore=False):
    """Decoder side.  No tag: a stream of the reference or of a build before the tag existed -- nothing to compare, logged.  Another
    tag: refuse (or warn with ignore=True) -- the decoded cloud could be silent garbage."""
    if tag is None or not tag.startswith('pcc_geo_cnn_v2_amd/'):
        logger.warning('the stream carries no codec-numerics tag: it was written by another build; decoding with %s', expected)
        return
    if tag != expected:
        msg = (f'the stream was encoded with codec numerics {tag}, this decoder computes {expected}: sigma-hat would differ in its last '
               'bits, scale indexes flip and the range decoder desynchronises (set the same PCC_* switches / --precision, or pass '
               '--ignore_numerics_tag to try anyway)')
        if not ignore:
            raise RuntimeError(msg)
        logger.warning(msg)
