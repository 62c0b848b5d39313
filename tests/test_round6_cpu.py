"""Round 6, CPU: host packing of the two-piece fp16 weight images and the new export."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_piece_fp16_winograd_image_reconstructs_the_weights_to_22_bits():
    """conv_wino_f16s.hip's host packing (pcc_wino_f16s_pack): U = (h + l) / su with h, l fp16 and su a power of two, lane-contiguous
    [cout group][cin group][lane][slot q = 3 py + 2 - dz][px][4 h | 4 l] + 8 B pad (776 B per lane), for 16, 32 and 64 channels."""
    from pcc_geo_cnn_v2_amd import _lib as L
    lib = L.lib()
    for C in (16, 32, 64):
        d = L.ConvDesc(N=1, D=16, H=16, W=16, Cin=C, Cout=C, k=3, stride=1, transposed=0, flags=0)
        n = lib.pcc_conv_packed_floats(ctypes.byref(d))
        rng = np.random.default_rng(3)
        w = (rng.standard_normal((3, 3, 3, C, C)) * np.exp(rng.uniform(-6, 2, (3, 3, 3, C, C)))).astype(np.float32)
        pk = np.zeros(n, np.float32)
        assert lib.pcc_conv_pack_weights(ctypes.byref(d), w.ctypes.data_as(ctypes.c_void_p), pk.ctypes.data_as(ctypes.c_void_p)) == 0
        G = C // 16
        u32 = pk[27 * C * C:27 * C * C + G * G * 48 * 64 * 4].reshape(G, G, 3, 4, 4, 64, 4)           # [cig][cog][dz][py][px][lane][c]
        per = 64 * 776 // 4
        tail = n - 64
        su = float(pk[tail])
        assert su > 0 and np.log2(su) == np.round(np.log2(su)) and 2.0 ** 13 <= np.abs(u32).max() * su < 2.0 ** 14
        img = pk[tail - G * G * per:tail].view(np.uint8).reshape(G, G, 64, 776)                       # [cog][cig][lane][bytes]
        assert not img[..., 768:].any()
        for cog in range(G):
            for cig in range(G):
                rows = img[cog, cig, :, :768].copy().view(np.float16).reshape(64, 12, 4, 2, 4)          # [lane][q][px][piece][c]
                for q in range(12):
                    py, dz = q // 3, 2 - q % 3
                    h, l = rows[:, q, :, 0, :].astype(np.float64), rows[:, q, :, 1, :].astype(np.float64)
                    ref = u32[cig, cog, dz, py].transpose(1, 0, 2).astype(np.float64) * su                # [lane][px][c]
                    assert np.array_equal(h, ref.astype(np.float16).astype(np.float64))               # h = fp16_rn(su U)
                    assert np.array_equal(l, (ref - h).astype(np.float16).astype(np.float64))          # l = fp16_rn(su U - h)
                    assert np.abs(h + l - ref).max() <= 2.0 ** -22 * np.abs(ref).max() + 2.0 ** -25


def test_header_declares_and_library_exports_the_kernel_family_query():
    from pcc_geo_cnn_v2_amd import _lib as L
    hdr = open(os.path.join(ROOT, 'include', 'pcc_geo.h')).read()
    assert 'int pcc_conv_kernel_family(pcc_ctx* ctx, const pcc_conv_desc* d, char* buf, int32_t cap);' in hdr
    assert hasattr(L.lib(), 'pcc_conv_kernel_family')
    assert '#define PCC_NUM_NO_F16S 0x4000' in hdr and L.PCC_NUM['no_f16s'] == 0x4000
