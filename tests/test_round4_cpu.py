"""Round 4, CPU side: provenance of profiled numbers, the host-budget fields of the bench line, the split-weight image."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_profiled_traffic_of_the_dominant_kernel_is_tied_to_the_source_it_was_measured_on():
    """bench.py no longer prints a constant as a measurement: `roofline.traffic` of the live line is null and
    `traffic_profiled` carries the file, the hash of the kernel source the counters were taken on and a `stale` flag.  This test
    fails when profiles/dominant_kernel_traffic.json is older than the kernel: re-run tools/profile_round.sh + summarize_round.py."""
    import bench
    t = bench.traffic_provenance()
    assert t is not None and t['file'] == 'profiles/dominant_kernel_traffic.json'
    assert t['kernel_source'] and t['kernel_source_sha1_at_measurement'] and t['kernel_source_sha1_now']
    assert 0.9 < t['traffic_over_algorithmic'] < 1.5 and t['hbm_bytes_per_launch'] > 1e9
    assert not t['stale'], (f"profiles/dominant_kernel_traffic.json was measured on {t['kernel_source']} {t['kernel_source_sha1_at_measurement'][:10]}, "
                            f"the tree holds {t['kernel_source_sha1_now'][:10]}: regenerate the profile")
    import subprocess
    blob = subprocess.run(['git', 'hash-object', os.path.join(ROOT, 'pcc_geo_cnn_v2_amd', 'csrc', t['kernel_source'])], capture_output=True, text=True)
    if blob.returncode == 0:
        assert blob.stdout.strip() == t['kernel_source_sha1_now']          # (the same hash git prints)


def test_split_weight_image_reconstructs_the_fp32_winograd_weights_exactly():
    """conv_wino_bf16.hip's host packing: every fp32 U = h + m + l exactly (three bf16 pieces, round-to-nearest residuals), laid out
    as the two MFMA operands [Uh | Um] and [Ul | Uh] per (slot, point, lane)."""
    import ctypes
    from pcc_geo_cnn_v2_amd import _lib as L
    lib = L.lib()
    C = 32
    d = L.ConvDesc(N=1, D=16, H=16, W=16, Cin=C, Cout=C, k=3, stride=1, transposed=0, flags=0)
    n = lib.pcc_conv_packed_floats(ctypes.byref(d))
    rng = np.random.default_rng(3)
    w = (rng.standard_normal((3, 3, 3, C, C)) * np.exp(rng.uniform(-20, 5, (3, 3, 3, C, C)))).astype(np.float32)      # wide exponent range
    pk = np.zeros(n, np.float32)
    assert lib.pcc_conv_pack_weights(ctypes.byref(d), w.ctypes.data_as(ctypes.c_void_p), pk.ctypes.data_as(ctypes.c_void_p)) == 0
    G = C // 16
    u32 = pk[27 * C * C:27 * C * C + G * G * 48 * 64 * 4].reshape(G * G, 3, 4, 4, 64, 4)           # [pair][dz][py][px][lane][c]
    # image order: Keras-layout taps | fp32 Winograd U | fp16 fragments (conv_f16.hip: C/16 * 9 * 3 KB for C = 32) | split-bf16 U | conv_split images
    ub0 = 27 * C * C + G * G * 48 * 64 * 4 + (C // 16) * 9 * 3 * 1024 // 4
    ub = pk[ub0:ub0 + G * G * 48 * 2 * 64 * 4].view(np.uint16).reshape(G * G, 12, 4, 2, 64, 8)      # [pair][slot q][px][operand][lane][8 bf16]

    def f32(h):
        return (h.astype(np.uint32) << 16).view(np.float32)
    for q in range(12):
        py, dz = q // 3, 2 - q % 3
        a1, a2 = ub[:, q, :, 0], ub[:, q, :, 1]                        # [pair][px][lane][8]
        uh, um, ul, uh2 = f32(a1[..., :4]), f32(a1[..., 4:]), f32(a2[..., :4]), f32(a2[..., 4:])
        assert np.array_equal(uh, uh2)
        ref = u32[:, dz, py]                                           # [pair][px][lane][4]
        assert np.array_equal((uh.astype(np.float64) + um + ul).astype(np.float32), ref)
        assert np.all(np.abs(um) <= np.abs(uh) * 2.0 ** -8 + 1e-45) and np.all(np.abs(ul) <= np.abs(uh) * 2.0 ** -16 + 1e-45)
