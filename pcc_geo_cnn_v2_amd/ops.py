"""Thin torch-facing wrappers over the C ABI (include/pcc_geo.h).

PyTorch-ROCm is plumbing only: it owns device memory (tensors) and streams; every operator below is
a hand-written HIP kernel (or the host range coder) inside libpcc_geo_hip.so.
"""
import contextlib
import ctypes as C
import os

import numpy as np
import torch

from . import _lib as L


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class Context:
    """One per (process, GPU).  Replaces the tf.Session of src/compress_octree.py:84-92."""

    def __init__(self, device=0):
        if not torch.cuda.is_available():
            raise L.PccError('no ROCm device visible to PyTorch: the MI355X path cannot run (no CPU fallback)')
        self.device = torch.device('cuda', device)
        h = C.c_void_p()
        L.check(L.lib().pcc_ctx_create(device, C.byref(h)), 'pcc_ctx_create')
        self.handle = h
        self.num_cu = L.lib().pcc_ctx_num_cu(h)
        self.numerics_at_creation = self.numerics()[1]       # the PCC_* environment switches as pcc_ctx_create read them

    @property
    def stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def workspace(self, nbytes):
        """Activations workspace of the batched-graph entry points: one caller-owned device buffer per context, grown on
        demand.  Calls on a context are ordered on one stream, so consecutive transforms may reuse it."""
        ws = getattr(self, '_ws', None)
        if ws is None or ws.numel() < nbytes:
            self._ws = ws = torch.empty((int(nbytes),), dtype=torch.uint8, device=self.device)
        return ws

    conv_flags = 0      # extra pcc_conv_desc.flags of every conv issued through this context (0 = the reference's fp32)

    # ---- codec numerics (include/pcc_geo.h): which kernel family computes a layer is state of the context, read once from the PCC_*
    #      environment switches at creation; encoder and decoder must agree on it (sigma-hat selects the entropy coder's rows)
    def numerics(self):
        """(kernel family of this build, PCC_NUM_* switches in effect)."""
        fam, sw = C.c_uint32(), C.c_uint32()
        L.check(L.lib().pcc_ctx_get_numerics(self.handle, C.byref(fam), C.byref(sw)), 'pcc_ctx_get_numerics')
        return int(fam.value), int(sw.value)

    def numerics_tag(self, precision='fp32'):
        """What the CLIs record beside a stream (gzip header comment and `.enc.metric.json` key `codec_numerics`) and the decoder compares.
        Switches that are documented AND tested as bit-identical (P16, COUT1_T16: launch geometry only) are masked out: they must not
        make a decoder refuse a stream."""
        fam, sw = self.numerics()
        sw &= ~(L.PCC_NUM['p16'] | L.PCC_NUM['cout1_t16'])
        return f'pcc_geo_cnn_v2_amd/k{fam}/sw{sw:04x}/{precision}'

    def set_numerics(self, **switches):
        """Replace named switches (L.PCC_NUM keys -> bool); returns the previous word.  Tests / A/B runs only: never between an
        encode and the decode of its stream."""
        _, old = self.numerics()
        new = old
        for k, v in switches.items():
            new = (new | L.PCC_NUM[k]) if v else (new & ~L.PCC_NUM[k])
        L.check(L.lib().pcc_ctx_set_numerics(self.handle, new), 'pcc_ctx_set_numerics')
        return old

    @contextlib.contextmanager
    def numerics_override(self, **switches):
        old = self.set_numerics(**switches)
        try:
            yield self
        finally:
            L.check(L.lib().pcc_ctx_set_numerics(self.handle, old), 'pcc_ctx_set_numerics')

    def view(self, conv_flags):
        """The same context (handle, device, stream, workspace) with other default conv flags -- how a model in the fp16 mode
        (PCC_CONV_F16) uses a context without changing it for anybody else."""
        if not conv_flags:
            return self
        views = self.__dict__.setdefault('_views', {})
        if conv_flags not in views:
            views[conv_flags] = _ContextView(self, conv_flags)
        return views[conv_flags]

    def close(self):
        if self.handle is not None:
            L.lib().pcc_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _ContextView:
    def __init__(self, base, conv_flags):
        self._base, self.conv_flags = base, conv_flags

    def __getattr__(self, name):          # handle, device, num_cu, stream, workspace, ...
        return getattr(self._base, name)

    def view(self, conv_flags):
        return self._base.view(conv_flags)


_CONTEXTS = {}


def get_context(device=None):
    """Process-wide context cache: one pcc_ctx per GPU."""
    if device is None:
        device = torch.device('cuda', torch.cuda.current_device() if torch.cuda.is_available() else 0)
    device = torch.device(device)
    if device.type != 'cuda':
        raise L.PccError(f'tensor is on {device}: the codec operators only run on the MI355X (no CPU fallback)')
    idx = device.index if device.index is not None else torch.cuda.current_device()
    if idx not in _CONTEXTS:
        _CONTEXTS[idx] = Context(idx)
    return _CONTEXTS[idx]


class ConvLayer:
    """Weights of one Conv3D / Conv3DTranspose (Keras layouts) + their device images."""

    def __init__(self, kernel, bias, stride, transposed, relu):
        kernel = np.ascontiguousarray(kernel, np.float32)
        self.k = int(kernel.shape[0])
        assert kernel.shape[:3] == (self.k,) * 3
        self.transposed = bool(transposed)
        if transposed:
            self.cout, self.cin = int(kernel.shape[3]), int(kernel.shape[4])
        else:
            self.cin, self.cout = int(kernel.shape[3]), int(kernel.shape[4])
        self.kernel = kernel
        self.bias = None if bias is None else np.ascontiguousarray(bias, np.float32)
        self.stride = int(stride)
        self.relu = bool(relu)
        self._dev = {}

    def desc(self, N, D, H, W, flags=0, impl=L.PCC_IMPL_AUTO, out_cstride=0, out_coffset=0):
        f = flags | (L.PCC_CONV_BIAS if self.bias is not None else 0) | (L.PCC_CONV_RELU if self.relu else 0)
        return L.ConvDesc(N, D, H, W, self.cin, self.cout, self.k, self.stride, int(self.transposed), f, impl,
                          out_cstride, out_coffset)

    def device_images(self, ctx, d):
        key = ctx.device.index
        if key not in self._dev:
            self._dev[key] = dict(w=torch.from_numpy(self.kernel).to(ctx.device),
                                  b=None if self.bias is None else torch.from_numpy(self.bias).to(ctx.device),
                                  pk=None)
        im = self._dev[key]
        if im['pk'] is None and L.lib().pcc_conv_mfma_supported(C.byref(d)) == 1:
            n = L.lib().pcc_conv_packed_floats(C.byref(d))
            pk = np.empty(n, np.float32)
            L.check(L.lib().pcc_conv_pack_weights(C.byref(d), self.kernel.ctypes.data_as(C.c_void_p),
                                                  pk.ctypes.data_as(C.c_void_p)), 'pcc_conv_pack_weights')
            im['pk'] = torch.from_numpy(pk).to(ctx.device)
        return im


def conv_out_shape(layer, x_shape):
    N, D, H, W, _ = x_shape
    s = layer.stride
    if layer.transposed:
        return (N, D * s, H * s, W * s, layer.cout)
    o = lambda n: -(-n // s)
    return (N, o(D), o(H), o(W), layer.cout)


# Optional live profiling of selected conv launches with HIP events on the launch stream (bench.py):
# PROFILE = {'match': callable(layer, x_shape) -> bool, 'events': [(start, end), ...]}
PROFILE = None


def conv3d(ctx, x, layer, residual=None, flags=0, impl=L.PCC_IMPL_AUTO, out=None, out_coffset=0):
    """x: (N,D,H,W,Cin) float32 contiguous on ctx.device.  Returns (N,OD,OH,OW,Cout)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.device == ctx.device and x.dim() == 5
    N, D, H, W, Cin = x.shape
    assert Cin == layer.cin, f'expected {layer.cin} input channels, got {Cin}'
    oshape = conv_out_shape(layer, x.shape)
    ocs = 0
    if out is None:
        out = torch.empty(oshape, dtype=torch.float32, device=ctx.device)
    else:
        assert out.is_contiguous() and tuple(out.shape[:4]) == tuple(oshape[:4])
        ocs = out.shape[4]
    if residual is not None:
        flags |= L.PCC_CONV_ADD
        assert residual.is_contiguous() and tuple(residual.shape) == tuple(oshape)
    flags |= getattr(ctx, 'conv_flags', 0)
    d = layer.desc(N, D, H, W, flags, impl, ocs, out_coffset)
    im = layer.device_images(ctx, d)
    prof = PROFILE is not None and PROFILE['match'](layer, x.shape)
    if prof:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream(ctx.device))
    rc = L.lib().pcc_conv3d(ctx.handle, C.byref(d), _ptr(x), _ptr(im['w']), _ptr(im['pk']), _ptr(im['b']),
                            _ptr(residual), _ptr(out), ctx.stream)
    L.check(rc, 'pcc_conv3d')
    if prof:
        e1.record(torch.cuda.current_stream(ctx.device))
        PROFILE['events'].append((e0, e1))
    return out


class NetworkWeights:
    """The weights of one whole transform (src/model_transforms.py:41-158) as ONE packed device blob per GPU
    (pcc_weights_upload): Keras-layout kernel + MFMA/Winograd fragment image + bias of every conv layer."""

    def __init__(self, transform_id, filters, conv_layers):
        self.transform, self.filters = int(transform_id), int(filters)
        n = L.lib().pcc_network_num_layers(self.transform, self.filters)
        L.check(n, 'pcc_network_num_layers')
        assert n == len(conv_layers), f'transform {transform_id}: {n} layers in the library, {len(conv_layers)} in the model'
        d, role = L.ConvDesc(), C.c_int32()
        for i, cl in enumerate(conv_layers):     # the model's layers must be the reference stack the library restates
            L.check(L.lib().pcc_network_layer(self.transform, self.filters, i, C.byref(d), C.byref(role)), 'pcc_network_layer')
            got = (cl.cin, cl.cout, cl.k, cl.stride, int(cl.transposed), cl.bias is not None, cl.relu)
            want = (d.Cin, d.Cout, d.k, d.stride, d.transposed, bool(d.flags & L.PCC_CONV_BIAS), bool(d.flags & L.PCC_CONV_RELU))
            assert got == want, f'layer {i} of transform {transform_id}: model {got} != library {want}'
        self.layers = list(conv_layers)
        self._blob = {}

    def blob(self, ctx):
        key = ctx.device.index
        if key not in self._blob:
            n = len(self.layers)
            ks = (C.c_void_p * n)(*[l.kernel.ctypes.data for l in self.layers])
            bs = (C.c_void_p * n)(*[None if l.bias is None else l.bias.ctypes.data for l in self.layers])
            dev = torch.empty((L.lib().pcc_weights_blob_floats(self.transform, self.filters),), dtype=torch.float32, device=ctx.device)
            L.check(L.lib().pcc_weights_upload(ctx.handle, self.transform, self.filters, ks, bs, _ptr(dev), ctx.stream),
                    'pcc_weights_upload')
            self._blob[key] = dev
        return self._blob[key]


_FAMILY = {0: 'analysis', 2: 'analysis', 4: 'analysis', 1: 'synthesis', 3: 'synthesis', 5: 'synthesis', 6: 'hyper_a', 7: 'hyper_s'}


def network_forward(ctx, net, x, final_flags=0):
    """y = transform(x) in ONE ABI call (pcc_network_forward_{analysis,synthesis,hyper_a,hyper_s}).  x: (N,D,H,W,Cin)."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.device == ctx.device and x.dim() == 5
    N, D, H, W, _ = x.shape
    od, oh, ow, oc = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    L.check(L.lib().pcc_network_out_dims(net.transform, net.filters, D, H, W, C.byref(od), C.byref(oh), C.byref(ow), C.byref(oc)),
            'pcc_network_out_dims')
    y = torch.empty((N, od.value, oh.value, ow.value, oc.value), dtype=torch.float32, device=ctx.device)
    nb = L.lib().pcc_network_workspace_bytes(net.transform, net.filters, N, D, H, W)
    ws = ctx.workspace(nb)
    fn = getattr(L.lib(), 'pcc_network_forward_' + _FAMILY[net.transform])
    L.check(fn(ctx.handle, net.transform, net.filters, _ptr(net.blob(ctx)), _ptr(x), N, D, H, W, _ptr(y), _ptr(ws), ws.numel(),
               getattr(ctx, 'conv_flags', 0), final_flags, ctx.stream), 'pcc_network_forward')
    return y


def codec_desc(ctx, version, filters, nets, medians=None, scale_table=None, round_mode=L.PCC_ROUND_FLOOR_HALF):
    """pcc_codec_desc of a model: nets = dict(analysis=, synthesis=, hyper_analysis=, hyper_synthesis=) of NetworkWeights
    (None where absent); medians / scale_table: device tensors.  Returns (desc, keepalive)."""
    d = L.CodecDesc()
    d.version, d.filters, d.round_mode = version, filters, round_mode
    d.analysis = nets['analysis'].transform if nets.get('analysis') is not None else -1
    d.synthesis = nets['synthesis'].transform
    keep = []
    for name in ('analysis', 'synthesis', 'hyper_analysis', 'hyper_synthesis'):
        net = nets.get(name)
        blob = net.blob(ctx) if net is not None else None
        keep.append(blob)
        setattr(d, 'w_' + name, None if blob is None else blob.data_ptr())
    d.medians = None if medians is None else medians.data_ptr()
    d.scale_table = None if scale_table is None else scale_table.data_ptr()
    d.scale_levels = 0 if scale_table is None else scale_table.numel()
    keep += [medians, scale_table]
    return d, keep


def usable_cores():
    """Cores this process may use: the affinity mask capped by the cgroup CPU quota (cgroup v2 cpu.max, v1 cfs_quota_us)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        try:
            quota = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            period = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if quota > 0:
                n = min(n, max(1, quota // period))
        except (OSError, ValueError):
            pass
    return n


_ITEM = {torch.uint8: 1, torch.int16: 2, torch.int32: 4}


class SymbolStaging:
    """What leaves the device for the host coder after one encode of B blocks, as ONE buffer: z symbols, y symbols, CDF-row
    indexes (stream order, narrow integers) and the per-tile max|symbol| of both symbol tensors.  `dev` (device uint8) is
    filled by the library (pcc_symbol_io / pcc_symbols_pack), `host` (pinned uint8) receives it in one copy; the
    attributes zsym / ysym / idx / ztm / ytm are views of `host`.  z pieces are absent for a version-1 codec."""

    def __init__(self, device, B, stream_shape_y, stream_shape_z, F, sym_dtype, idx_dtype, channels_first):
        self.sym_dtype, self.idx_dtype, self.channels_first = sym_dtype, idx_dtype, bool(channels_first)
        vy = int(np.prod(stream_shape_y[1:])) // F
        vz = int(np.prod(stream_shape_z[1:])) // F if stream_shape_z is not None else 0
        self.vy, self.vz, self.B, self.F = vy, vz, B, F
        pieces = [('ysym', stream_shape_y, sym_dtype)]
        if stream_shape_z is not None:
            pieces += [('zsym', stream_shape_z, sym_dtype), ('idx', stream_shape_y, idx_dtype)]
        pieces.append(('ytm', (L.lib().pcc_symbols_tiles(B, vy, F),), torch.int32))
        if stream_shape_z is not None:
            pieces.append(('ztm', (L.lib().pcc_symbols_tiles(B, vz, F),), torch.int32))
        off, self.layout = 0, {}
        for name, shape, dt in pieces:
            nbytes = int(np.prod(shape)) * _ITEM[dt]
            self.layout[name] = (off, nbytes, tuple(shape), dt)
            off += (nbytes + 15) // 16 * 16
        self.nbytes = off
        self.dev = torch.empty((off,), dtype=torch.uint8, device=device)
        self.host = torch.empty((off,), dtype=torch.uint8, pin_memory=True)
        for name, (o, nb, shape, dt) in self.layout.items():
            setattr(self, name, self.host[o:o + nb].view(dt).reshape(shape))

    def dev_ptr(self, name):
        return self.dev.data_ptr() + self.layout[name][0] if name in self.layout else None

    def sink(self):
        k = L.SymbolSink()
        k.zsym, k.ysym, k.idx = self.dev_ptr('zsym'), self.dev_ptr('ysym'), self.dev_ptr('idx')
        k.zsym_tile_max, k.ysym_tile_max = self.dev_ptr('ztm'), self.dev_ptr('ytm')
        k.sym_bytes, k.idx_bytes, k.channels_first = _ITEM[self.sym_dtype], _ITEM[self.idx_dtype], int(self.channels_first)
        return k

    def pack(self, ctx, ysym, zsym=None, idx=None):
        """the same packing from Python (the per-layer path, which has no pcc_codec_encode call to do it)"""
        symbols_pack(ctx, ysym, self.channels_first, self.dev_ptr('ysym'), _ITEM[self.sym_dtype], self.dev_ptr('ytm'))
        if zsym is not None:
            symbols_pack(ctx, zsym, self.channels_first, self.dev_ptr('zsym'), _ITEM[self.sym_dtype], self.dev_ptr('ztm'))
            symbols_pack(ctx, idx, self.channels_first, self.dev_ptr('idx'), _ITEM[self.idx_dtype], None)

    def copy_out(self):
        """device -> pinned host on the current stream (one copy)"""
        self.host.copy_(self.dev, non_blocking=True)


def symbols_pack(ctx, src, channels_first, dst_ptr, dst_bytes, tile_max_ptr=None):
    """(N, ..., C) int32 device tensor -> stream order, dst_bytes-wide integers at the device address dst_ptr."""
    assert src.dtype == torch.int32 and src.is_contiguous() and src.device == ctx.device
    N, Cc = src.shape[0], src.shape[-1]
    L.check(L.lib().pcc_symbols_pack(ctx.handle, _ptr(src), N, src[0].numel() // Cc, Cc, int(bool(channels_first)),
                                     C.c_void_p(dst_ptr), dst_bytes, None if tile_max_ptr is None else C.c_void_p(tile_max_ptr),
                                     ctx.stream), 'pcc_symbols_pack')


def symbols_unpack(ctx, src, ndhwc_shape, channels_first):
    """stream-order device tensor (uint8 / int16 / int32) of N blocks -> (N,D,H,W,C) int32 device tensor."""
    assert src.is_contiguous() and src.device == ctx.device and src.dtype in _ITEM
    out = torch.empty(tuple(ndhwc_shape), dtype=torch.int32, device=ctx.device)
    N, Cc = out.shape[0], out.shape[-1]
    assert src.numel() == out.numel()
    L.check(L.lib().pcc_symbols_unpack(ctx.handle, _ptr(src), _ITEM[src.dtype], N, out[0].numel() // Cc, Cc,
                                       int(bool(channels_first)), _ptr(out), ctx.stream), 'pcc_symbols_unpack')
    return out


def codec_encode(ctx, desc, x, thr=None, cap=None, symbols_ready=None, staging=None, scratch=None):
    """The GPU part of compress() (src/model_types.py:289-293 / :379-388) for a batch of blocks in ONE ABI call.
    x: (N,D,H,W) float32.  Returns dict of device tensors (NDHWC); with `thr` (N,) float32 also the encoder-side point
    lists xyz / counts of the clipped x_hat (fixed-threshold policy).  symbols_ready: a torch.cuda.Event that has been
    recorded once (so that its handle exists); the library re-records it when the symbols are final.  staging: a
    SymbolStaging whose device buffer the library fills (stream order, narrow integers) before that event."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4 and x.device == ctx.device
    N, D, H, W = x.shape
    F, dev = desc.filters, ctx.device
    f32 = lambda *sh: torch.empty(sh, dtype=torch.float32, device=dev)
    i32 = lambda *sh: torch.empty(sh, dtype=torch.int32, device=dev)
    ys, zs = (N, D // 8, H // 8, W // 8, F), (N, D // 16, H // 16, W // 16, F)
    t = dict(y=f32(*ys), symbols=i32(*ys), y_hat=f32(*ys), x_hat=f32(N, D, H, W))
    if desc.version == 2:
        t.update(z=f32(*zs), z_symbols=i32(*zs), z_hat=f32(*zs), sigma_hat=f32(*ys), indexes=i32(*ys))
    xyz = counts = None
    cap = D * H * W if cap is None else int(cap)
    if thr is not None:
        assert thr.dtype == torch.float32 and thr.numel() == N
        xyz, counts = torch.empty((N, cap, 3), dtype=torch.float32, device=dev), torch.empty((N,), dtype=torch.int32, device=dev)
        n_scratch = L.lib().pcc_threshold_scratch_ints(N, D, H, W)
        if scratch is None:
            scratch = torch.empty((n_scratch,), dtype=torch.int32, device=dev)
        assert scratch.dtype == torch.int32 and scratch.numel() >= n_scratch and scratch.device == dev
        t.update(xyz=xyz, counts=counts)
    ws = ctx.workspace(L.lib().pcc_codec_workspace_bytes(C.byref(desc), N, D, H, W))
    L.check(L.lib().pcc_codec_encode(ctx.handle, C.byref(desc), _ptr(x), N, D, H, W, _ptr(t['y']), _ptr(t.get('z')),
                                     _ptr(t.get('z_symbols')), _ptr(t.get('z_hat')), _ptr(t.get('sigma_hat')),
                                     _ptr(t.get('indexes')), _ptr(t['symbols']), _ptr(t['y_hat']), _ptr(t['x_hat']), _ptr(thr),
                                     _ptr(xyz), _ptr(counts), cap, _ptr(scratch), _ptr(ws), ws.numel(),
                                     getattr(ctx, 'conv_flags', 0), 0, None if staging is None else C.byref(staging.sink()),
                                     None if symbols_ready is None else C.c_void_p(symbols_ready.cuda_event), ctx.stream),
            'pcc_codec_encode')
    return t


def _packed_io(packed, channels_first, idx_out=None):
    """pcc_symbol_io for the decoder calls: `packed` = stream-order symbols on the device (int16 / int32) as the host->device
    copy delivered them, idx_out = device tensor (uint8 / int32) that receives the packed CDF-row indexes."""
    k = L.SymbolSink()
    k.sym_bytes = _ITEM[packed.dtype]
    k.idx_bytes = 1 if idx_out is None else _ITEM[idx_out.dtype]
    k.channels_first = int(bool(channels_first))
    k.idx = None if idx_out is None else idx_out.data_ptr()
    return k


def codec_decode_hyper(ctx, desc, zsym, dhw, packed=None, channels_first=True, idx_packed=None):
    """z symbols (N,D/16,H/16,W/16,F) int32 -> z_hat, sigma_hat, indexes (src/model_types.py:403-406), one ABI call.
    packed: instead of zsym, the stream-order symbols (N, ...) int16 / int32 on the device -- the library unpacks them (the
    int32 tensor comes back as 'z_symbols'); idx_packed: device tensor (uint8 / int32, stream order) that receives the indexes."""
    (D, H, W), F, dev = dhw, desc.filters, ctx.device
    N = (zsym if packed is None else packed).shape[0]
    zs, ys = (N, D // 16, H // 16, W // 16, F), (N, D // 8, H // 8, W // 8, F)
    io = None
    if packed is not None or idx_packed is not None:
        src = packed if packed is not None else torch.empty((0,), dtype=torch.int16)
        io = _packed_io(src, channels_first, idx_packed)
        if packed is not None:
            assert packed.is_contiguous() and packed.device == dev and packed.numel() == int(np.prod(zs))
            io.zsym = packed.data_ptr()
            zsym = torch.empty(zs, dtype=torch.int32, device=dev)
        assert idx_packed is None or (idx_packed.is_contiguous() and idx_packed.numel() == int(np.prod(ys)))
    assert zsym.dtype == torch.int32 and zsym.is_contiguous() and tuple(zsym.shape) == zs
    t = dict(z_hat=torch.empty(zs, dtype=torch.float32, device=dev), sigma_hat=torch.empty(ys, dtype=torch.float32, device=dev),
             indexes=torch.empty(ys, dtype=torch.int32, device=dev), z_symbols=zsym)
    ws = ctx.workspace(L.lib().pcc_codec_workspace_bytes(C.byref(desc), N, D, H, W))
    L.check(L.lib().pcc_codec_decode_hyper(ctx.handle, C.byref(desc), _ptr(zsym), N, D, H, W, _ptr(t['z_hat']), _ptr(t['sigma_hat']),
                                           _ptr(t['indexes']), _ptr(ws), ws.numel(), getattr(ctx, 'conv_flags', 0),
                                           None if io is None else C.byref(io), ctx.stream),
            'pcc_codec_decode_hyper')
    return t


def codec_decode_main(ctx, desc, ysym, dhw, thr=None, cap=None, packed=None, channels_first=True, scratch=None):
    """y symbols -> y_hat -> x_hat (+ thresholding and compaction when `thr` (N,) float32 is given), one ABI call
    (src/model_types.py:305-307 / :407-408, :232-234).  packed: instead of ysym, the stream-order symbols on the device (see
    codec_decode_hyper); the int32 tensor comes back as 'symbols'."""
    (D, H, W), F, dev = dhw, desc.filters, ctx.device
    N = (ysym if packed is None else packed).shape[0]
    ys = (N, D // 8, H // 8, W // 8, F)
    io = None
    if packed is not None:
        assert packed.is_contiguous() and packed.device == dev and packed.numel() == int(np.prod(ys))
        io = _packed_io(packed, channels_first)
        io.ysym = packed.data_ptr()
        ysym = torch.empty(ys, dtype=torch.int32, device=dev)
    assert ysym.dtype == torch.int32 and ysym.is_contiguous() and tuple(ysym.shape) == ys
    t = dict(y_hat=torch.empty(ys, dtype=torch.float32, device=dev), x_hat=torch.empty((N, D, H, W), dtype=torch.float32, device=dev),
             symbols=ysym)
    xyz = counts = None
    cap = D * H * W if cap is None else int(cap)
    if thr is not None:
        assert thr.dtype == torch.float32 and thr.numel() == N
        xyz = torch.empty((N, cap, 3), dtype=torch.float32, device=dev)
        counts = torch.empty((N,), dtype=torch.int32, device=dev)
        n_scratch = L.lib().pcc_threshold_scratch_ints(N, D, H, W)
        if scratch is None:
            scratch = torch.empty((n_scratch,), dtype=torch.int32, device=dev)
        assert scratch.dtype == torch.int32 and scratch.numel() >= n_scratch and scratch.device == dev
        t.update(xyz=xyz, counts=counts)
    ws = ctx.workspace(L.lib().pcc_codec_workspace_bytes(C.byref(desc), N, D, H, W))
    L.check(L.lib().pcc_codec_decode_main(ctx.handle, C.byref(desc), _ptr(ysym), N, D, H, W, _ptr(t['y_hat']), _ptr(t['x_hat']),
                                          _ptr(thr), _ptr(xyz), _ptr(counts), cap, _ptr(scratch), _ptr(ws), ws.numel(),
                                          getattr(ctx, 'conv_flags', 0), None if io is None else C.byref(io), ctx.stream),
            'pcc_codec_decode_main')
    return t


def profile_select(ctx, transform, layer, stride=1):
    """Live HIP-event timing of one layer of one transform inside the graph calls; stride > 1 times every stride-th call only."""
    L.check(L.lib().pcc_profile_select(ctx.handle, transform, layer | (int(stride) << 16) if transform >= 0 else layer), 'pcc_profile_select')


def profile_read(ctx, cap=8192):
    ms, n = (C.c_float * cap)(), C.c_int32()
    L.check(L.lib().pcc_profile_read(ctx.handle, ms, cap, C.byref(n)), 'pcc_profile_read')
    return [ms[i] for i in range(n.value)]


def conv3d_fp16_storage(ctx, x, layer, residual=None, in16=None, out16=True, flags=0):
    """One layer of the fp16 mode with fp16 tensors in HBM (PCC_CONV_IN16 / OUT16 / RES16, include/pcc_geo.h): x fp16 (k3
    stride-1 layers, Cin = Cout in {16, 32, 64}) or fp32 (k3 stride-2 transposed layers, out16 only).  pcc_network_forward chains
    these itself in the fp16 mode; this wrapper exists for tests and for callers that chain layers by hand."""
    in16 = (x.dtype == torch.float16) if in16 is None else in16
    assert x.is_contiguous() and x.device == ctx.device and x.dim() == 5 and x.dtype == (torch.float16 if in16 else torch.float32)
    N, D, H, W, Cin = x.shape
    assert Cin == layer.cin
    oshape = conv_out_shape(layer, x.shape)
    out = torch.empty(oshape, dtype=torch.float16 if out16 else torch.float32, device=ctx.device)
    f = flags | L.PCC_CONV_F16 | (L.PCC_CONV_IN16 if in16 else 0) | (L.PCC_CONV_OUT16 if out16 else 0)
    if residual is not None:
        assert in16 and residual.dtype == torch.float16 and residual.is_contiguous() and tuple(residual.shape) == tuple(oshape)
        f |= L.PCC_CONV_ADD | L.PCC_CONV_RES16
    d = layer.desc(N, D, H, W, f)
    im = layer.device_images(ctx, d)
    L.check(L.lib().pcc_conv3d(ctx.handle, C.byref(d), _ptr(x), _ptr(im['w']), _ptr(im['pk']), _ptr(im['b']), _ptr(residual),
                               _ptr(out), ctx.stream), 'pcc_conv3d')
    return out


def mfma_supported(layer, x_shape):
    N, D, H, W, _ = x_shape
    d = layer.desc(N, D, H, W)
    return L.lib().pcc_conv_mfma_supported(C.byref(d)) == 1


def quantize(ctx, v, medians=None, mode=L.PCC_ROUND_FLOOR_HALF, want_sym=True, want_deq=True):
    assert v.dtype == torch.float32 and v.is_contiguous()
    Cn = v.shape[-1]
    sym = torch.empty(v.shape, dtype=torch.int32, device=v.device) if want_sym else None
    deq = torch.empty_like(v) if want_deq else None
    L.check(L.lib().pcc_quantize(ctx.handle, _ptr(v), _ptr(medians), _ptr(sym), _ptr(deq), v.numel(), Cn, mode,
                                 ctx.stream), 'pcc_quantize')
    return sym, deq


def dequantize(ctx, sym, medians=None):
    assert sym.dtype == torch.int32 and sym.is_contiguous()
    deq = torch.empty(sym.shape, dtype=torch.float32, device=sym.device)
    L.check(L.lib().pcc_dequantize(ctx.handle, _ptr(sym), _ptr(medians), _ptr(deq), sym.numel(), sym.shape[-1],
                                   ctx.stream), 'pcc_dequantize')
    return deq


def scale_to_index(ctx, sigma, table):
    assert sigma.dtype == torch.float32 and sigma.is_contiguous() and table.dtype == torch.float32
    idx = torch.empty(sigma.shape, dtype=torch.int32, device=sigma.device)
    L.check(L.lib().pcc_scale_to_index(ctx.handle, _ptr(sigma), _ptr(table), table.numel(), _ptr(idx), sigma.numel(),
                                       ctx.stream), 'pcc_scale_to_index')
    return idx


def threshold_compact(ctx, x, thr, clip=False, cap=None):
    """x: (B,D,H,W) float32; thr: (B,) float32 device tensor.  Returns (xyz (B,cap,3), counts (B,))."""
    assert x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 4
    B, D, H, W = x.shape
    cap = D * H * W if cap is None else int(cap)
    xyz = torch.empty((B, cap, 3), dtype=torch.float32, device=x.device)
    counts = torch.empty((B,), dtype=torch.int32, device=x.device)
    scratch = torch.empty((L.lib().pcc_threshold_scratch_ints(B, D, H, W),), dtype=torch.int32, device=x.device)
    L.check(L.lib().pcc_threshold_compact(ctx.handle, _ptr(x), B, D, H, W, _ptr(thr), int(clip), _ptr(xyz),
                                          _ptr(counts), cap, _ptr(scratch), ctx.stream), 'pcc_threshold_compact')
    return xyz, counts


def voxelize(ctx, pts, block_of, B, D, H, W):
    """pts (n,3) int32, block_of (n,) int32 (device) -> dense (B,D,H,W) float32 of {0,1}."""
    dense = torch.zeros((B, D, H, W), dtype=torch.float32, device=ctx.device)
    if pts.numel():
        assert pts.dtype == torch.int32 and pts.is_contiguous() and block_of.dtype == torch.int32
        L.check(L.lib().pcc_voxelize(ctx.handle, _ptr(pts), _ptr(block_of), pts.shape[0], B, D, H, W, _ptr(dense),
                                     ctx.stream), 'pcc_voxelize')
    return dense


def focal_loss(ctx, y_true, y_pred, gamma=2.0, alpha=0.9):
    """src/utils/focal_loss.py:5-12 -> 0-d float32 tensor on the device."""
    assert y_true.numel() == y_pred.numel() and y_true.is_contiguous() and y_pred.is_contiguous()
    out = torch.empty((1,), dtype=torch.float32, device=y_pred.device)
    scratch = torch.empty((L.lib().pcc_focal_scratch_floats(),), dtype=torch.float32, device=y_pred.device)
    L.check(L.lib().pcc_focal_loss(ctx.handle, _ptr(y_true), _ptr(y_pred), y_true.numel(), gamma, alpha, _ptr(out),
                                   _ptr(scratch), ctx.stream), 'pcc_focal_loss')
    return out[0]


def d1_threshold_stats(ctx, x_hat, thr, pts, block_of, clip=True):
    """Exact D1 sums for every threshold of every block (see include/pcc_geo.h).  x_hat (B,D,H,W) float32,
    thr (T<=256,) float32, pts (n,3) int32 grouped by block, block_of (n,) int32 -- all on the device.
    Returns int64 numpy arrays s_ab (B,256), s_ba (B,256), n_b (B,256) indexed by threshold, and tcount (B,)."""
    assert x_hat.dtype == torch.float32 and x_hat.is_contiguous() and x_hat.dim() == 4
    assert pts.dtype == torch.int32 and pts.is_contiguous() and block_of.dtype == torch.int32 and block_of.is_contiguous()
    assert thr.dtype == torch.float32 and thr.is_contiguous()
    B, D, H, W = x_hat.shape
    ws = torch.empty((L.lib().pcc_d1_search_workspace_bytes(B, D, H, W),), dtype=torch.uint8, device=x_hat.device)
    s_ab = torch.empty((B, 256), dtype=torch.int64, device=x_hat.device)
    hsum, hcnt = torch.empty_like(s_ab), torch.empty_like(s_ab)
    tcount = torch.empty((B,), dtype=torch.int32, device=x_hat.device)
    L.check(L.lib().pcc_d1_threshold_stats(ctx.handle, _ptr(x_hat), B, D, H, W, _ptr(thr), thr.numel(), int(clip),
                                           _ptr(pts), _ptr(block_of), pts.shape[0], _ptr(ws), _ptr(s_ab), _ptr(hsum),
                                           _ptr(hcnt), _ptr(tcount), ctx.stream), 'pcc_d1_threshold_stats')
    hs, hc = hsum.cpu().numpy(), hcnt.cpu().numpy()
    # suffix sums over levels k > t
    s_ba = np.concatenate([np.cumsum(hs[:, ::-1], 1)[:, ::-1][:, 1:], np.zeros((B, 1), np.int64)], 1)
    n_b = np.concatenate([np.cumsum(hc[:, ::-1], 1)[:, ::-1][:, 1:], np.zeros((B, 1), np.int64)], 1)
    return s_ab.cpu().numpy(), s_ba, n_b, tcount.cpu().numpy()


def d12_threshold_stats(ctx, x_hat, thr, pts, block_of, block_start, normals, clip=True):
    """d1_threshold_stats plus the D2 sums of every threshold (include/pcc_geo.h: pcc_d12_threshold_stats).  normals (n,3) float32,
    block_start (B+1,) int32 -- on the device.  Returns (s_ab, s_ba, n_b, tcount, d2_ab, d2_ba); d2_* float64 (B,256)."""
    assert x_hat.dtype == torch.float32 and x_hat.is_contiguous() and x_hat.dim() == 4
    assert pts.dtype == torch.int32 and pts.is_contiguous() and block_of.dtype == torch.int32 and block_of.is_contiguous()
    assert normals.dtype == torch.float32 and normals.is_contiguous() and normals.shape == pts.shape
    assert block_start.dtype == torch.int32 and block_start.is_contiguous() and thr.dtype == torch.float32 and thr.is_contiguous()
    B, D, H, W = x_hat.shape
    assert block_start.numel() == B + 1
    dev = x_hat.device
    ws = torch.empty((L.lib().pcc_d1_search_workspace_bytes(B, D, H, W),), dtype=torch.uint8, device=dev)
    ws2 = torch.empty((L.lib().pcc_d12_search_workspace_bytes(B, D, H, W, pts.shape[0]),), dtype=torch.uint8, device=dev)
    s_ab = torch.empty((B, 256), dtype=torch.int64, device=dev)
    hsum, hcnt = torch.empty_like(s_ab), torch.empty_like(s_ab)
    d2_ab = torch.empty((B, 256), dtype=torch.float64, device=dev)
    d2_ba = torch.empty_like(d2_ab)
    tcount = torch.empty((B,), dtype=torch.int32, device=dev)
    L.check(L.lib().pcc_d12_threshold_stats(ctx.handle, _ptr(x_hat), B, D, H, W, _ptr(thr), thr.numel(), int(clip), _ptr(pts), _ptr(block_of),
                                            _ptr(block_start), pts.shape[0], _ptr(normals), _ptr(ws), _ptr(ws2), _ptr(s_ab), _ptr(hsum), _ptr(hcnt),
                                            _ptr(tcount), _ptr(d2_ab), _ptr(d2_ba), ctx.stream), 'pcc_d12_threshold_stats')
    hs, hc = hsum.cpu().numpy(), hcnt.cpu().numpy()
    s_ba = np.concatenate([np.cumsum(hs[:, ::-1], 1)[:, ::-1][:, 1:], np.zeros((B, 1), np.int64)], 1)
    n_b = np.concatenate([np.cumsum(hc[:, ::-1], 1)[:, ::-1][:, 1:], np.zeros((B, 1), np.int64)], 1)
    return s_ab.cpu().numpy(), s_ba, n_b, tcount.cpu().numpy(), d2_ab.cpu().numpy(), d2_ba.cpu().numpy()


# ---------------------------------------------------------------------------------------------
# host range coder
# ---------------------------------------------------------------------------------------------
class HostCdfTable:
    """Quantised CDF table (rows, stride) + per-row size/offset, as the reference's
    `quantized_cdf` / `cdf_length` / `offset` (src/utils/patch_gaussian_conditional.py:91-97,118)."""

    def __init__(self, cdf, cdf_size, offset, precision=16, overflow_width=4):
        self.cdf = np.ascontiguousarray(cdf, np.int32)
        self.cdf_size = np.ascontiguousarray(cdf_size, np.int32)
        self.offset = np.ascontiguousarray(offset, np.int32)
        assert self.cdf.ndim == 2 and len(self.cdf_size) == len(self.offset) == self.cdf.shape[0]
        self.struct = L.CdfTable(self.cdf.ctypes.data_as(C.POINTER(C.c_int32)),
                                 self.cdf_size.ctypes.data_as(C.POINTER(C.c_int32)),
                                 self.offset.ctypes.data_as(C.POINTER(C.c_int32)),
                                 self.cdf.shape[0], self.cdf.shape[1], precision, overflow_width)


def _np_i32(a):
    if isinstance(a, torch.Tensor):
        a = a.numpy()
    a = np.ascontiguousarray(a, np.int32).reshape(-1)
    return a


def _np_host(a, allowed):
    """flat contiguous host array; keeps a dtype in `allowed` (narrow staging buffers), anything else becomes int32"""
    if isinstance(a, torch.Tensor):
        a = a.numpy()
    a = np.asarray(a)
    return np.ascontiguousarray(a if a.dtype in allowed else a.astype(np.int32)).reshape(-1)


def _uniform_dtype(arrs, allowed):
    """one dtype for all streams of a call (the ABI takes one element size per call)"""
    if all(a.dtype == arrs[0].dtype for a in arrs) and arrs[0].dtype in allowed:
        return arrs, arrs[0].dtype.itemsize
    return [np.ascontiguousarray(a, np.int32) for a in arrs], 4


_SYM, _ROW = (np.dtype(np.int16), np.dtype(np.int32)), (np.dtype(np.uint8), np.dtype(np.int32))


def _rows_2d(x, allowed):
    """x: a 2-D (streams, symbols) host array / CPU tensor whose dtype the ABI takes as it is -> C-contiguous numpy 2-D, else None."""
    if isinstance(x, torch.Tensor):
        x = x.numpy()
    if isinstance(x, np.ndarray) and x.ndim >= 2 and x.dtype in allowed:
        return np.ascontiguousarray(x.reshape(x.shape[0], -1))
    return None


def _row_ptrs(a2d):
    """uint64[streams]: the address of every row of a C-contiguous 2-D array -- what the ABI's `const void* const*` wants, without a
    Python loop over the streams (the per-stream ctypes bookkeeping was 0.4 ms per coder call: 1.6 ms of GIL-holding time per step)."""
    return np.uint64(a2d.ctypes.data) + np.arange(a2d.shape[0], dtype=np.uint64) * np.uint64(a2d.strides[0])


def _pp(ptrs):
    return ptrs.ctypes.data_as(C.POINTER(C.c_void_p))


def _sz(a):
    return a.ctypes.data_as(C.POINTER(C.c_size_t))


def range_encode_batch(table, data_list, index_list=None, index_mod=0, n_threads=0):
    """data_list: per-stream symbol arrays, int32 or int16 -- a list, or ONE 2-D (streams, symbols) array / CPU tensor (fast path: no
    per-stream Python work); index_list: per-stream CDF rows, int32 or uint8: a list, a 2-D array, or a single 1-D array shared by all
    streams.  Returns list of bytes."""
    d2 = _rows_2d(data_list, _SYM)
    if d2 is not None:
        S, n_sym = d2.shape
        if S == 0:
            return []
        if index_list is None:
            ip, ib, keep = None, 4, None
        else:
            i2 = _rows_2d(index_list, _ROW)
            if i2 is not None:
                assert i2.shape == d2.shape
                keep, ib = i2, i2.dtype.itemsize
                ip = _row_ptrs(i2)
            else:                                   # one row vector for every stream
                keep = _np_host(index_list, _ROW)
                assert keep.size == n_sym
                ib, ip = keep.dtype.itemsize, np.full(S, keep.ctypes.data, np.uint64)
        cap = n_sym * 8 + 64
        outs = np.empty((S, cap), np.uint8)
        n = np.full(S, n_sym, np.uint64)
        caps = np.full(S, cap, np.uint64)
        olen = np.zeros(S, np.uint64)
        dp, op = _row_ptrs(d2), _row_ptrs(outs)
        L.check(L.lib().pcc_range_encode_batch_n(C.byref(table.struct), S, _pp(dp), d2.dtype.itemsize, None if ip is None else _pp(ip), ib, index_mod,
                                                 _sz(n), _pp(op), _sz(caps), _sz(olen), n_threads), 'pcc_range_encode_batch_n')
        del keep
        return [outs[s_, :int(olen[s_])].tobytes() for s_ in range(S)]
    S = len(data_list)
    if S == 0:
        return []
    data, db = _uniform_dtype([_np_host(d, _SYM) for d in data_list], _SYM)
    idx, ib = (None, 4) if index_list is None else _uniform_dtype([_np_host(i, _ROW) for i in index_list], _ROW)
    n = (C.c_size_t * S)(*[d.size for d in data])
    caps = [d.size * 8 + 64 for d in data]
    outs = [np.empty(c, np.uint8) for c in caps]
    dp = (C.c_void_p * S)(*[d.ctypes.data for d in data])
    ip = None if idx is None else (C.c_void_p * S)(*[i.ctypes.data for i in idx])
    op = (C.c_void_p * S)(*[o.ctypes.data for o in outs])
    cap = (C.c_size_t * S)(*caps)
    olen = (C.c_size_t * S)()
    if db == 4 and ib == 4:
        L.check(L.lib().pcc_range_encode_batch(C.byref(table.struct), S, dp, ip, index_mod, n, op, cap, olen, n_threads),
                'pcc_range_encode_batch')
    else:
        L.check(L.lib().pcc_range_encode_batch_n(C.byref(table.struct), S, dp, db, ip, ib, index_mod, n, op, cap, olen, n_threads),
                'pcc_range_encode_batch_n')
    return [outs[s][:olen[s]].tobytes() for s in range(S)]


def range_decode_batch(table, strings, n_list, index_list=None, index_mod=0, n_threads=0, out=None):
    """strings: list of bytes; n_list: symbols per stream; index_list: int32 or uint8 CDF rows (list, 2-D array, or one shared 1-D
    array).  Returns list of int32 numpy arrays, or fills the provided `out` -- a list of arrays, or ONE 2-D (streams, symbols) array /
    CPU tensor (fast path); int32, or int16: then a symbol that does not fit raises OverflowError and the caller decodes into int32."""
    S = len(strings)
    if S == 0:
        return []
    o2 = _rows_2d(out, _SYM) if out is not None and not isinstance(out, (list, tuple)) else None
    if o2 is not None:
        n_sym = o2.shape[1]
        assert o2.shape[0] == S and all(int(k) == n_sym for k in n_list)
        assert o2.ctypes.data == (out.numpy() if isinstance(out, torch.Tensor) else out).ctypes.data, 'out must be C-contiguous (it is filled in place)'
        if index_list is None:
            ip, ib, keep = None, 4, None
        else:
            # one row per stream (2-D) or ONE row shared by all streams (1-D); a list of per-stream arrays belongs to the legacy path
            # below (flattened here it would decode every stream with stream 0's rows)
            assert not isinstance(index_list, (list, tuple)), 'range_decode_batch: a 2-D `out` takes a 2-D (or one shared 1-D) index array'
            i2 = _rows_2d(index_list, _ROW)
            if i2 is not None:
                assert i2.shape == o2.shape, f'range_decode_batch: index rows {i2.shape} for symbols {o2.shape}'
                keep, ib, ip = i2, i2.dtype.itemsize, _row_ptrs(i2)
            else:
                keep = _np_host(index_list, _ROW)
                assert keep.ndim == 1 and keep.size == n_sym, f'range_decode_batch: shared index of {keep.size} rows for {n_sym} symbols'
                ib, ip = keep.dtype.itemsize, np.full(S, keep.ctypes.data, np.uint64)
        lens = np.fromiter((len(s_) for s_ in strings), np.uint64, S)
        blob = np.frombuffer(b''.join(strings) + b'\0', np.uint8)            # all strings in one buffer: pointers by offset
        sp = np.uint64(blob.ctypes.data) + np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.uint64)
        n = np.full(S, n_sym, np.uint64)
        rc = L.lib().pcc_range_decode_batch_n(C.byref(table.struct), S, _pp(sp), _sz(lens), None if ip is None else _pp(ip), ib, index_mod, _sz(n),
                                              _pp(_row_ptrs(o2)), o2.dtype.itemsize, n_threads)
        del keep, blob
        if rc == L.PCC_ERR_SPACE and o2.dtype.itemsize == 2:
            raise OverflowError('a decoded symbol does not fit int16')
        L.check(rc, 'pcc_range_decode_batch_n')
        return out
    bufs = [np.frombuffer(s, np.uint8) if len(s) else np.zeros(1, np.uint8) for s in strings]
    idx, ib = (None, 4) if index_list is None else _uniform_dtype([_np_host(i, _ROW) for i in index_list], _ROW)
    outs = [np.empty(int(k), np.int32) for k in n_list] if out is None else out
    ob = outs[0].dtype.itemsize
    assert all(o.dtype == outs[0].dtype and o.flags['C_CONTIGUOUS'] for o in outs) and outs[0].dtype in _SYM
    sp = (C.c_void_p * S)(*[b.ctypes.data for b in bufs])
    sl = (C.c_size_t * S)(*[len(s) for s in strings])
    ip = None if idx is None else (C.c_void_p * S)(*[i.ctypes.data for i in idx])
    n = (C.c_size_t * S)(*[int(k) for k in n_list])
    op = (C.c_void_p * S)(*[o.ctypes.data for o in outs])
    if ob == 4 and ib == 4:
        L.check(L.lib().pcc_range_decode_batch(C.byref(table.struct), S, sp, sl, ip, index_mod, n, op, n_threads),
                'pcc_range_decode_batch')
    else:
        rc = L.lib().pcc_range_decode_batch_n(C.byref(table.struct), S, sp, sl, ip, ib, index_mod, n, op, ob, n_threads)
        if rc == L.PCC_ERR_SPACE and ob == 2:
            raise OverflowError('a decoded symbol does not fit int16')
        L.check(rc, 'pcc_range_decode_batch_n')
    return outs


def pmf_to_quantized_cdf(pmf, precision=16):
    pmf = np.ascontiguousarray(pmf, np.float32)
    cdf = np.zeros(pmf.size + 1, np.int32)
    L.check(L.lib().pcc_pmf_to_quantized_cdf(pmf.ctypes.data_as(C.c_void_p), pmf.size, precision,
                                             cdf.ctypes.data_as(C.c_void_p)), 'pcc_pmf_to_quantized_cdf')
    return cdf
