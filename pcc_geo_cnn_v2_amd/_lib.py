"""ctypes binding of libpcc_geo_hip.so (the C ABI declared in include/pcc_geo.h).

The library is built in-tree by `make -C pcc_geo_cnn_v2_amd/csrc` (see __graft_entry__.build()).
There is NO CPU fallback: if the shared object is missing, or no gfx950 device is visible when a
context is requested, the product path raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('PCC_GEO_LIB', os.path.join(_HERE, 'libpcc_geo_hip.so'))   # override: A/B builds

PCC_CONV_BIAS, PCC_CONV_RELU, PCC_CONV_ADD, PCC_CONV_CLIP01, PCC_CONV_F16 = 1, 2, 4, 8, 16
PCC_CONV_IN16, PCC_CONV_OUT16, PCC_CONV_RES16 = 32, 64, 128          # fp16 storage inside the fp16 mode
PCC_IMPL_AUTO, PCC_IMPL_GENERIC, PCC_IMPL_MFMA, PCC_IMPL_WINOGRAD, PCC_IMPL_SPLIT = 0, 1, 2, 3, 4
PCC_ROUND_FLOOR_HALF, PCC_ROUND_HALF_EVEN = 0, 1

EXPORTS = [
    'pcc_abi_version', 'pcc_last_error', 'pcc_ctx_create', 'pcc_ctx_destroy', 'pcc_ctx_num_cu', 'pcc_ctx_get_numerics', 'pcc_ctx_set_numerics',
    'pcc_conv_out_dims', 'pcc_conv_mfma_supported', 'pcc_conv_packed_floats', 'pcc_conv_pack_weights', 'pcc_conv_kernel_family',
    'pcc_conv3d', 'pcc_quantize', 'pcc_dequantize', 'pcc_scale_to_index', 'pcc_threshold_compact',
    'pcc_threshold_scratch_ints', 'pcc_voxelize', 'pcc_focal_loss', 'pcc_focal_scratch_floats',
    'pcc_symbols_tiles', 'pcc_symbols_pack', 'pcc_symbols_unpack',
    'pcc_range_encode_batch', 'pcc_range_decode_batch', 'pcc_range_encode_batch_n', 'pcc_range_decode_batch_n', 'pcc_pmf_to_quantized_cdf',
    'pcc_d1_search_workspace_bytes', 'pcc_d1_threshold_stats', 'pcc_d12_search_workspace_bytes', 'pcc_d12_threshold_stats', 'pcc_octree_bucket',
    'pcc_network_num_layers', 'pcc_network_layer', 'pcc_weights_blob_floats', 'pcc_weights_pack', 'pcc_weights_upload',
    'pcc_network_workspace_bytes', 'pcc_network_out_dims', 'pcc_network_forward', 'pcc_network_forward_analysis',
    'pcc_network_forward_synthesis', 'pcc_network_forward_hyper_a', 'pcc_network_forward_hyper_s',
    'pcc_codec_workspace_bytes', 'pcc_codec_encode', 'pcc_codec_decode_hyper', 'pcc_codec_decode_main',
    'pcc_profile_select', 'pcc_profile_read',
]
ABI_VERSION = 4
# include/pcc_geo.h "codec numerics": switches that select the kernel family of a layer (state of the context, recorded beside every stream)
PCC_NUM = dict(no_split=0x1, no_split_direct=0x2, no_split_tr2=0x4, no_winograd=0x8, no_winograd32=0x10, no_winograd64=0x20,
               wino_per_group=0x40, no_tr2m=0x80, tr2m=0x100, tr2_old=0x200, split_mfma16=0x400, split_mfma32=0x800, split_tile8=0x1000,
               p16=0x2000, no_f16s=0x4000, cout1_t16=0x8000)
PCC_ERR_ARG, PCC_ERR_HIP, PCC_ERR_NOGPU, PCC_ERR_SPACE, PCC_ERR_CORRUPT = -1, -2, -3, -4, -5      # include/pcc_geo.h
(PCC_NET_ANALYSIS_V1, PCC_NET_SYNTHESIS_V1, PCC_NET_ANALYSIS_V2, PCC_NET_SYNTHESIS_V2, PCC_NET_ANALYSIS_PROGRESSIVE_V2,
 PCC_NET_SYNTHESIS_PROGRESSIVE_V2, PCC_NET_HYPER_ANALYSIS, PCC_NET_HYPER_SYNTHESIS) = range(8)


class ConvDesc(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ('N', 'D', 'H', 'W', 'Cin', 'Cout', 'k', 'stride', 'transposed',
                                         'flags', 'impl', 'out_cstride', 'out_coffset')]


class CdfTable(C.Structure):
    _fields_ = [('cdf', C.POINTER(C.c_int32)), ('cdf_size', C.POINTER(C.c_int32)), ('offset', C.POINTER(C.c_int32)),
                ('rows', C.c_int32), ('cdf_stride', C.c_int32), ('precision', C.c_int32),
                ('overflow_width', C.c_int32)]


class CodecDesc(C.Structure):
    _fields_ = [('version', C.c_int32), ('filters', C.c_int32), ('analysis', C.c_int32), ('synthesis', C.c_int32),
                ('w_analysis', C.c_void_p), ('w_synthesis', C.c_void_p), ('w_hyper_analysis', C.c_void_p),
                ('w_hyper_synthesis', C.c_void_p), ('medians', C.c_void_p), ('scale_table', C.c_void_p),
                ('scale_levels', C.c_int32), ('round_mode', C.c_int32)]


class SymbolSink(C.Structure):
    _fields_ = [('zsym', C.c_void_p), ('ysym', C.c_void_p), ('idx', C.c_void_p), ('zsym_tile_max', C.c_void_p),
                ('ysym_tile_max', C.c_void_p), ('sym_bytes', C.c_int32), ('idx_bytes', C.c_int32), ('channels_first', C.c_int32)]


class PccError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (loudly failing if it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Importing torch first makes
    # our library bind to that already-loaded runtime, so the process has ONE HIP runtime and our kernels
    # share torch's device context, streams and allocations.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise PccError(f'{LIB_PATH} is missing: build it with `make -C pcc_geo_cnn_v2_amd/csrc` '
                       '(python -c "import __graft_entry__ as g; g.build()"). There is no CPU fallback.')
    L = C.CDLL(LIB_PATH)
    vp, i32, sz = C.c_void_p, C.c_int32, C.c_size_t
    L.pcc_abi_version.restype = C.c_int
    L.pcc_last_error.restype = C.c_char_p
    L.pcc_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.pcc_ctx_destroy.argtypes = [vp]
    L.pcc_ctx_num_cu.argtypes = [vp]
    L.pcc_ctx_get_numerics.argtypes = [vp, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.pcc_ctx_set_numerics.argtypes = [vp, C.c_uint32]
    L.pcc_conv_out_dims.argtypes = [C.POINTER(ConvDesc), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    L.pcc_conv_mfma_supported.argtypes = [C.POINTER(ConvDesc)]
    L.pcc_conv_packed_floats.argtypes = [C.POINTER(ConvDesc)]
    L.pcc_conv_packed_floats.restype = sz
    L.pcc_conv_kernel_family.argtypes = [vp, C.POINTER(ConvDesc), C.c_char_p, i32]
    L.pcc_conv_pack_weights.argtypes = [C.POINTER(ConvDesc), vp, vp]
    L.pcc_conv3d.argtypes = [vp, C.POINTER(ConvDesc), vp, vp, vp, vp, vp, vp, vp]
    L.pcc_quantize.argtypes = [vp, vp, vp, vp, vp, sz, i32, i32, vp]
    L.pcc_dequantize.argtypes = [vp, vp, vp, vp, sz, i32, vp]
    L.pcc_scale_to_index.argtypes = [vp, vp, vp, i32, vp, sz, vp]
    L.pcc_threshold_compact.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, vp, vp, C.c_int64, vp, vp]
    L.pcc_threshold_scratch_ints.argtypes = [i32, i32, i32, i32]
    L.pcc_threshold_scratch_ints.restype = sz
    L.pcc_voxelize.argtypes = [vp, vp, vp, C.c_int64, i32, i32, i32, i32, vp, vp]
    L.pcc_focal_loss.argtypes = [vp, vp, vp, sz, C.c_float, C.c_float, vp, vp, vp]
    L.pcc_focal_scratch_floats.restype = sz
    L.pcc_range_encode_batch.argtypes = [C.POINTER(CdfTable), i32, vp, vp, i32, vp, vp, vp, vp, i32]
    L.pcc_range_decode_batch.argtypes = [C.POINTER(CdfTable), i32, vp, vp, vp, i32, vp, vp, i32]
    L.pcc_pmf_to_quantized_cdf.argtypes = [vp, i32, i32, vp]
    L.pcc_d1_search_workspace_bytes.argtypes = [i32, i32, i32, i32]
    L.pcc_d1_search_workspace_bytes.restype = sz
    L.pcc_d1_threshold_stats.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.pcc_d12_search_workspace_bytes.argtypes = [i32, i32, i32, i32, C.c_int64]
    L.pcc_d12_search_workspace_bytes.restype = sz
    L.pcc_d12_threshold_stats.argtypes = [vp, vp, i32, i32, i32, i32, vp, i32, i32, vp, vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.pcc_octree_bucket.argtypes = [vp, C.c_int64, i32, i32, i32, vp, vp]
    L.pcc_octree_bucket.restype = C.c_int64
    L.pcc_network_num_layers.argtypes = [i32, i32]
    L.pcc_network_layer.argtypes = [i32, i32, i32, C.POINTER(ConvDesc), C.POINTER(i32)]
    L.pcc_weights_blob_floats.argtypes = [i32, i32]
    L.pcc_weights_blob_floats.restype = sz
    L.pcc_weights_pack.argtypes = [i32, i32, vp, vp, vp]
    L.pcc_weights_upload.argtypes = [vp, i32, i32, vp, vp, vp, vp]
    L.pcc_network_workspace_bytes.argtypes = [i32, i32, i32, i32, i32, i32]
    L.pcc_network_workspace_bytes.restype = sz
    L.pcc_network_out_dims.argtypes = [i32, i32, i32, i32, i32, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.POINTER(i32)]
    for fn in (L.pcc_network_forward, L.pcc_network_forward_analysis, L.pcc_network_forward_synthesis,
               L.pcc_network_forward_hyper_a, L.pcc_network_forward_hyper_s):
        fn.argtypes = [vp, i32, i32, vp, vp, i32, i32, i32, i32, vp, vp, sz, i32, i32, vp]
    L.pcc_codec_workspace_bytes.argtypes = [C.POINTER(CodecDesc), i32, i32, i32, i32]
    L.pcc_codec_workspace_bytes.restype = sz
    L.pcc_codec_encode.argtypes = [vp, C.POINTER(CodecDesc), vp, i32, i32, i32, i32] + [vp] * 9 + [vp, vp, vp, C.c_int64, vp] + \
        [vp, sz, i32, i32, C.POINTER(SymbolSink), vp, vp]
    L.pcc_symbols_tiles.argtypes = [i32, C.c_int64, i32]
    L.pcc_symbols_tiles.restype = sz
    L.pcc_symbols_pack.argtypes = [vp, vp, i32, C.c_int64, i32, i32, vp, i32, vp, vp]
    L.pcc_symbols_unpack.argtypes = [vp, vp, i32, i32, C.c_int64, i32, i32, vp, vp]
    L.pcc_codec_decode_hyper.argtypes = [vp, C.POINTER(CodecDesc), vp, i32, i32, i32, i32, vp, vp, vp, vp, sz, i32, C.POINTER(SymbolSink), vp]
    L.pcc_codec_decode_main.argtypes = [vp, C.POINTER(CodecDesc), vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, C.c_int64, vp,
                                        vp, sz, i32, C.POINTER(SymbolSink), vp]
    L.pcc_profile_select.argtypes = [vp, i32, i32]
    L.pcc_profile_read.argtypes = [vp, vp, i32, C.POINTER(i32)]
    for name in EXPORTS:
        getattr(L, name)          # AttributeError here = the shared object is older than the header
    if L.pcc_abi_version() != ABI_VERSION:
        raise PccError(f'{LIB_PATH} has ABI version {L.pcc_abi_version()}, this package needs {ABI_VERSION}: rebuild it')
    _lib = L
    return L


def check(rc, what=''):
    """Map a negative status to the exception classes the reference raises at the same spot."""
    if rc is not None and rc < 0:
        msg = lib().pcc_last_error().decode('utf-8', 'replace')
        if rc == -1:
            raise AssertionError(f'{what}: {msg}')
        raise PccError(f'{what}: {msg} (status {rc})')
    return rc
