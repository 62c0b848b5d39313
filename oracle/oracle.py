"""CPU ORACLE for the pcc_geo_cnn_v2 64^3-block encode+decode path  --  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (pcc_geo_cnn_v2_amd) must never import anything under oracle/.

It wraps oracle/libpcc_oracle.so (plain C, built by oracle/Makefile) and restates, in numpy, the
graph wiring of the reference:

  * layer lists .............. /root/reference/src/model_transforms.py:41-158
  * compress / decompress ..... /root/reference/src/model_types.py:283-309 (V1), 371-411 (V2)
  * Gaussian tables / indexes . /root/reference/src/utils/patch_gaussian_conditional.py:49-125
  * factorized prior .......... tensorflow-compression==1.3 `EntropyBottleneck` (requirements.txt:7;
                                source NOT under /root/reference -> restated from the published
                                algorithm, **parity unpinned**)

PARITY STATUS: conv / entropy numerics are **parity unpinned** (TensorFlow 1.15 and tfc 1.3 cannot
be installed here, and the reference's tests pin only shapes).  Pinned pieces: the shape contract of
src/test_model_transforms.py:27-73, the threshold/argwhere known answers of
src/test_model_opt.py:12-49, container + octree fixtures generated from the importable reference
modules (tests/golden/make_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
u8p = C.POINTER(C.c_uint8)


def build():
    """Compile the C restatement (gcc).  Building the checker is not using it."""
    subprocess.check_call(['make', '-s', '-C', _HERE])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, 'libpcc_oracle.so')
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        for name in ('pcc_oracle_conv3d', 'pcc_oracle_conv3d_transpose'):
            fn = getattr(L, name)
            fn.restype = C.c_int
            fn.argtypes = [f32p, f32p, f32p, f32p] + [C.c_int] * 9
        L.pcc_oracle_focal_loss.restype = C.c_double
        L.pcc_oracle_focal_loss.argtypes = [f32p, f32p, C.c_size_t, C.c_float, C.c_float]
        L.pcc_oracle_threshold_argwhere.restype = C.c_long
        L.pcc_oracle_threshold_argwhere.argtypes = [f32p, C.c_int, C.c_int, C.c_int, C.c_float, f32p, C.c_long]
        L.pcc_oracle_clip01.restype = None
        L.pcc_oracle_clip01.argtypes = [f32p, C.c_size_t]
        L.pcc_oracle_quantize.restype = None
        L.pcc_oracle_quantize.argtypes = [f32p, f32p, i32p, f32p, C.c_size_t, C.c_int, C.c_int]
        L.pcc_oracle_scale_index.restype = None
        L.pcc_oracle_scale_index.argtypes = [f32p, f32p, C.c_int, i32p, C.c_size_t]
        L.pcc_oracle_range_encode.restype = C.c_long
        L.pcc_oracle_range_encode.argtypes = [i32p, i32p, C.c_size_t, i32p, C.c_int, i32p, i32p, C.c_int, C.c_int, u8p, C.c_size_t]
        L.pcc_oracle_range_decode.restype = C.c_int
        L.pcc_oracle_range_decode.argtypes = [u8p, C.c_size_t, i32p, C.c_size_t, i32p, C.c_int, i32p, i32p, C.c_int, C.c_int, i32p]
        L.pcc_oracle_pmf_to_quantized_cdf.restype = C.c_int
        L.pcc_oracle_pmf_to_quantized_cdf.argtypes = [f32p, C.c_int, C.c_int, i32p]
        _LIB = L
    return _LIB


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(i32p)


# ---------------------------------------------------------------------------------------------
# operators
# ---------------------------------------------------------------------------------------------
def conv3d(x, w, b=None, stride=1, relu=False):
    """x (N,D,H,W,Cin), w (k,k,k,Cin,Cout) -> (N,ceil(D/s),...,Cout).  TF padding='same'."""
    x, xp = _f(x)
    w, wp = _f(w)
    N, D, H, W, Cin = x.shape
    k, Cout = w.shape[0], w.shape[4]
    assert w.shape == (k, k, k, Cin, Cout)
    o = lambda n: -(-n // stride)
    out = np.empty((N, o(D), o(H), o(W), Cout), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    r = lib().pcc_oracle_conv3d(xp, wp, bp, out.ctypes.data_as(f32p), N, D, H, W, Cin, Cout, k, stride, int(relu))
    assert r == 0
    return out


def conv3d_transpose(x, w, b=None, stride=1, relu=False):
    """x (N,D,H,W,Cin), w (k,k,k,Cout,Cin) -> (N,D*s,H*s,W*s,Cout).  TF padding='same'."""
    x, xp = _f(x)
    w, wp = _f(w)
    N, D, H, W, Cin = x.shape
    k, Cout = w.shape[0], w.shape[3]
    assert w.shape == (k, k, k, Cout, Cin)
    out = np.empty((N, D * stride, H * stride, W * stride, Cout), np.float32)
    bp = None
    if b is not None:
        b, bp = _f(b)
    r = lib().pcc_oracle_conv3d_transpose(xp, wp, bp, out.ctypes.data_as(f32p), N, D, H, W, Cin, Cout, k, stride, int(relu))
    assert r == 0
    return out


def focal_loss(y_true, y_pred, gamma=2.0, alpha=0.9):
    """/root/reference/src/utils/focal_loss.py:5-12"""
    yt, ytp = _f(y_true)
    yp, ypp = _f(y_pred)
    assert yt.size == yp.size
    return lib().pcc_oracle_focal_loss(ytp, ypp, yt.size, gamma, alpha)


def threshold_argwhere(x_hat, thr):
    """np.argwhere(x_hat > thr).astype(float32) with the compare done in float32
    (/root/reference/src/model_types.py:209,233-234; numpy-1.18 value-based casting)."""
    x, xp = _f(x_hat)
    D, H, W = x.shape
    out = np.empty((x.size, 3), np.float32)
    n = lib().pcc_oracle_threshold_argwhere(xp, D, H, W, np.float32(thr), out.ctypes.data_as(f32p), x.size)
    return out[:n].copy()


def clip01(x):
    x = np.array(x, dtype=np.float32, copy=True)
    lib().pcc_oracle_clip01(x.ctypes.data_as(f32p), x.size)
    return x


def quantize(v, medians=None, mode=0):
    """returns (symbols int32, dequantised float32); channels-last, medians per channel."""
    v, vp = _f(v)
    Cn = v.shape[-1]
    mp = None
    if medians is not None:
        medians, mp = _f(medians)
        assert medians.size == Cn
    sym = np.empty(v.shape, np.int32)
    deq = np.empty(v.shape, np.float32)
    lib().pcc_oracle_quantize(vp, mp, sym.ctypes.data_as(i32p), deq.ctypes.data_as(f32p), v.size, Cn, mode)
    return sym, deq


def scale_index(sigma, table):
    s, sp = _f(sigma)
    t, tp = _f(table)
    idx = np.empty(s.shape, np.int32)
    lib().pcc_oracle_scale_index(sp, tp, t.size, idx.ctypes.data_as(i32p), s.size)
    return idx


def range_encode(data, index, cdf, cdf_size, offset, precision=16, overflow_width=4):
    d, dp = _i(np.ravel(data))
    ix, ixp = _i(np.ravel(index))
    assert d.size == ix.size
    cdf, cp = _i(cdf)
    cs, csp = _i(cdf_size)
    of, ofp = _i(offset)
    cap = d.size * 8 + 64
    out = np.empty(cap, np.uint8)
    n = lib().pcc_oracle_range_encode(dp, ixp, d.size, cp, cdf.shape[1], csp, ofp, precision, overflow_width,
                                      out.ctypes.data_as(u8p), cap)
    assert n >= 0
    return out[:n].tobytes()


def range_decode(string, index, cdf, cdf_size, offset, precision=16, overflow_width=4):
    s = np.frombuffer(string, np.uint8).copy() if len(string) else np.zeros(1, np.uint8)
    ix, ixp = _i(np.ravel(index))
    cdf, cp = _i(cdf)
    cs, csp = _i(cdf_size)
    of, ofp = _i(offset)
    out = np.empty(ix.size, np.int32)
    lib().pcc_oracle_range_decode(s.ctypes.data_as(u8p), len(string), ixp, ix.size, cp, cdf.shape[1], csp, ofp,
                                  precision, overflow_width, out.ctypes.data_as(i32p))
    return out.reshape(np.shape(index))


def pmf_to_quantized_cdf(pmf, precision=16):
    p, pp = _f(pmf)
    cdf = np.zeros(p.size + 1, np.int32)
    r = lib().pcc_oracle_pmf_to_quantized_cdf(pp, p.size, precision, cdf.ctypes.data_as(i32p))
    assert r == 0
    return cdf


# ---------------------------------------------------------------------------------------------
# transforms: layer lists restated from /root/reference/src/model_transforms.py
# a layer = (kind, cout, k, stride, bias, relu, res) with res in {None,'save','add'}
# ---------------------------------------------------------------------------------------------
def _block(kind, f):
    # AnalysisBlock :62-70 / SynthesisBlock :73-81 + ResidualLayer.call :30-38 (mode 'add'):
    # t1 = L0(x); t = L2(L1(t1)); return t1 + t   (every conv has bias+ReLU; no ReLU after the add)
    return [(kind, f, 3, 2, True, True, 'save'), (kind, f, 3, 1, True, True, None), (kind, f, 3, 1, True, True, 'add')]


def transform_layers(name, filters):
    F = filters
    if name == 'AnalysisTransformV1':  # :41-48
        return [('conv', F, 9, 2, True, True, None), ('conv', F, 5, 2, True, True, None),
                ('conv', F, 5, 2, False, False, None)]
    if name == 'SynthesisTransformV1':  # :51-59  (final activation is ReLU, :58)
        return [('convT', F, 5, 2, True, True, None), ('convT', F, 5, 2, True, True, None),
                ('convT', 1, 9, 2, True, True, None)]
    if name == 'AnalysisTransformV2':  # :84-95
        return _block('conv', F // 2) + _block('conv', F) + _block('conv', F) + [('conv', F, 3, 1, False, False, None)]
    if name == 'SynthesisTransformV2':  # :98-109
        return _block('convT', F) + _block('convT', F) + _block('convT', F // 2) + [('convT', 1, 3, 1, True, True, None)]
    if name == 'AnalysisTransformProgressiveV2':  # :112-123
        return _block('conv', F // 4) + _block('conv', F // 2) + _block('conv', F) + [('conv', F, 3, 1, False, False, None)]
    if name == 'SynthesisTransformProgressiveV2':  # :126-137
        return _block('convT', F) + _block('convT', F // 2) + _block('convT', F // 4) + [('convT', 1, 3, 1, True, True, None)]
    if name == 'HyperAnalysisTransform':  # :140-147
        return [('conv', F, 3, 1, True, True, None), ('conv', F, 3, 2, True, True, None),
                ('conv', F, 3, 1, False, False, None)]
    if name == 'HyperSynthesisTransform':  # :150-158 (all three bias+ReLU)
        return [('convT', F, 3, 1, True, True, None), ('convT', F, 3, 2, True, True, None),
                ('convT', F, 3, 1, True, True, None)]
    raise KeyError(name)


def run_transform(name, filters, params, prefix, x):
    """params: dict '<prefix>/<i>/kernel' (Keras layout) and '<prefix>/<i>/bias'.  x NDHWC."""
    t1 = None
    for i, (kind, cout, k, s, bias, relu, res) in enumerate(transform_layers(name, filters)):
        w = params[f'{prefix}/{i}/kernel']
        b = params.get(f'{prefix}/{i}/bias') if bias else None
        y = (conv3d if kind == 'conv' else conv3d_transpose)(x, w, b, stride=s, relu=relu)
        if res == 'save':
            t1 = y
        elif res == 'add':
            y = (t1 + y).astype(np.float32)
        x = y
    return x


# model configs, /root/reference/src/model_configs.py:16-42
CONFIGS = {
    'c1': dict(v=1, F=32, a='AnalysisTransformV1', s='SynthesisTransformV1'),
    'c2': dict(v=2, F=32, a='AnalysisTransformV1', s='SynthesisTransformV1'),
    'c3': dict(v=2, F=32, a='AnalysisTransformV2', s='SynthesisTransformV2'),
    'c3p': dict(v=2, F=64, a='AnalysisTransformProgressiveV2', s='SynthesisTransformProgressiveV2'),
}


# ---------------------------------------------------------------------------------------------
# entropy tables
# ---------------------------------------------------------------------------------------------
def scale_table(scales_min=0.11, scales_max=256, scales_levels=64):
    # /root/reference/src/model_types.py:318,324
    return np.exp(np.linspace(np.log(scales_min), np.log(scales_max), scales_levels))


def gaussian_tables(table, tail_mass=2 ** -8, precision=16):
    """/root/reference/src/utils/patch_gaussian_conditional.py:62-100,118.
    tail_mass: tfc 1.3 `EntropyModel.__init__(tail_mass=2**-8, ...)` default (not visible in the
    reference; the call at model_types.py:385 passes none)."""
    from scipy.special import erfc
    from scipy.stats import norm
    table = np.asarray(table, np.float64)
    multiplier = -norm.ppf(tail_mass / 2)
    pmf_center = np.ceil(table * multiplier).astype(int)
    pmf_length = 2 * pmf_center + 1
    max_length = int(np.max(pmf_length))
    samples = np.abs(np.arange(max_length, dtype=int) - pmf_center[:, None]).astype(np.float32)
    sc = table.astype(np.float32)[:, None]
    cum = lambda x: (np.float32(0.5) * erfc(np.float32(-(2 ** -0.5)) * x.astype(np.float32))).astype(np.float32)
    upper = cum((np.float32(.5) - samples) / sc)
    lower = cum((np.float32(-.5) - samples) / sc)
    pmf = (upper - lower).astype(np.float32)
    tail = (2 * lower[:, :1]).astype(np.float32)
    cdf = np.zeros((len(table), max_length + 2), np.int32)
    for i in range(len(table)):
        prob = np.concatenate([pmf[i, :pmf_length[i]], tail[i]])
        cdf[i, :pmf_length[i] + 2] = pmf_to_quantized_cdf(prob, precision)
    return cdf, (pmf_length + 2).astype(np.int32), (-pmf_center).astype(np.int32)


def _softplus(x):
    return np.logaddexp(0, x)


def factorized_logits_cumulative(eb, x):
    """tfc 1.3 EntropyBottleneck._logits_cumulative.  eb: dict matrices[i] (C,f_{i+1},f_i),
    biases[i] (C,f_{i+1},1), factors[i] (C,f_{i+1},1); x (C,1,n) float32."""
    logits = x.astype(np.float32)
    n = len(eb['matrices'])
    for i in range(n):
        m = _softplus(eb['matrices'][i].astype(np.float32)).astype(np.float32)
        logits = np.matmul(m, logits).astype(np.float32)
        logits = logits + eb['biases'][i].astype(np.float32)
        if i < len(eb['factors']):
            f = np.tanh(eb['factors'][i].astype(np.float32))
            logits = logits + f * np.tanh(logits)
        logits = logits.astype(np.float32)
    return logits


def factorized_tables(eb, precision=16):
    """tfc 1.3 EntropyBottleneck.build: quantiles (C,1,3) -> medians/minima/maxima -> pmf -> cdf."""
    q = eb['quantiles'].astype(np.float32)
    medians = q[:, 0, 1]
    minima = np.maximum(np.ceil(medians - q[:, 0, 0]).astype(np.int32), 0)
    maxima = np.maximum(np.ceil(q[:, 0, 2] - medians).astype(np.int32), 0)
    pmf_start = medians - minima.astype(np.float32)
    pmf_length = maxima + minima + 1
    max_length = int(pmf_length.max())
    samples = np.arange(max_length, dtype=np.float32)[None, None, :] + pmf_start[:, None, None]
    half = np.float32(.5)
    lower = factorized_logits_cumulative(eb, samples - half)
    upper = factorized_logits_cumulative(eb, samples + half)
    sig = lambda t: (1.0 / (1.0 + np.exp(-t.astype(np.float64)))).astype(np.float32)
    sign = -np.sign(lower + upper)
    pmf = np.abs(sig(sign * upper) - sig(sign * lower))[:, 0, :]
    tail = (sig(lower[:, 0, :1]) + sig(-upper[:, 0, -1:])).astype(np.float32)
    Cn = len(medians)
    cdf = np.zeros((Cn, max_length + 2), np.int32)
    for c in range(Cn):
        prob = np.concatenate([pmf[c, :pmf_length[c]], tail[c]]).astype(np.float32)
        cdf[c, :pmf_length[c] + 2] = pmf_to_quantized_cdf(prob, precision)
    return dict(cdf=cdf, cdf_size=(pmf_length + 2).astype(np.int32), offset=(-minima).astype(np.int32),
                medians=medians.astype(np.float32))


# ---------------------------------------------------------------------------------------------
# whole-block pipelines, /root/reference/src/model_types.py:283-309 (V1) and :371-411 (V2)
# model = dict(config=<name>, params=<weights>, eb=<factorized_tables dict>, gc=(cdf,cdf_size,offset),
#              scale_table=<float32 table>, round_mode=0|1)
# ---------------------------------------------------------------------------------------------
def to_stream(a, model):
    """(1,D,H,W,C) tensor -> the flattening order of its range-coded stream.  tfc 1.3 codes a batch item's tensor in its
    memory order: (C,D,H,W) under the reference's default data_format='channels_first' (model_types.py:180,254,377),
    (D,H,W,C) under 'channels_last' (the default of this restatement)."""
    a = np.asarray(a)
    return np.ascontiguousarray(np.moveaxis(a, -1, 1)) if model.get('data_format') == 'channels_first' else a


def from_stream(flat, shape_ndhwc, model):
    """inverse of to_stream for a decoded flat symbol array."""
    N, D, H, W, C = shape_ndhwc
    if model.get('data_format') == 'channels_first':
        return np.ascontiguousarray(np.moveaxis(np.reshape(flat, (N, C, D, H, W)), 1, -1))
    return np.reshape(flat, shape_ndhwc)


def _channel_rows(shape_ndhwc, model):
    F = shape_ndhwc[-1]
    return to_stream(np.broadcast_to(np.arange(F, dtype=np.int32), shape_ndhwc), model)


def compress_block(model, x, run=None):
    """x: (1,D,H,W,1) float32 occupancy.  Returns (strings, x_hat (D,H,W) unclipped, debug).
    `run` replaces run_transform (e.g. torch_oracle.run_transform for 64^3 blocks)."""
    run = run or run_transform
    cfg = CONFIGS[model['config']]
    P, F, rm = model['params'], cfg['F'], model.get('round_mode', 0)
    eb = model['eb']
    y = np.asarray(run(cfg['a'], F, P, 'analysis', x), np.float32)
    dbg = {'y': y}
    if cfg['v'] == 1:
        sym, y_hat = quantize(y, eb['medians'], rm)
        y_string = range_encode(to_stream(sym, model), _channel_rows(y.shape, model), eb['cdf'], eb['cdf_size'], eb['offset'])
        strings = (y_string,)
        dbg.update(symbols=sym)
    else:
        z = np.asarray(run('HyperAnalysisTransform', F, P, 'hyper_analysis', y), np.float32)
        zsym, z_hat = quantize(z, eb['medians'], rm)
        z_string = range_encode(to_stream(zsym, model), _channel_rows(z.shape, model), eb['cdf'], eb['cdf_size'], eb['offset'])
        sigma = np.asarray(run('HyperSynthesisTransform', F, P, 'hyper_synthesis', z_hat), np.float32)
        idx = scale_index(sigma, model['scale_table'])
        ysym, y_hat = quantize(y, None, rm)
        gcdf, gsize, goff = model['gc']
        y_string = range_encode(to_stream(ysym, model), to_stream(idx, model), gcdf, gsize, goff)
        strings = (y_string, z_string)
        dbg.update(z=z, z_hat=z_hat, sigma_hat=sigma, indexes=idx, symbols=ysym, z_symbols=zsym)
    x_hat = np.asarray(run(cfg['s'], F, P, 'synthesis', y_hat), np.float32)
    dbg.update(y_hat=y_hat, x_hat=x_hat)
    return strings, x_hat[0, :, :, :, 0], dbg


def decompress_entropy_only(model, strings, x_shape, indexes):
    """Range-decode the strings of one block with known indexes (no transforms): what the reference's compress graph does
    with the strings it has just produced (model_types.py:292,383,387).  Returns the decoded symbol arrays (NDHWC)."""
    cfg = CONFIGS[model['config']]
    F, eb, xs = cfg['F'], model['eb'], np.asarray(x_shape)
    if cfg['v'] == 1:
        yshape = (1,) + tuple(xs // 8) + (F,)
        return (from_stream(range_decode(strings[0], _channel_rows(yshape, model), eb['cdf'], eb['cdf_size'], eb['offset']), yshape, model),)
    zshape, yshape = (1,) + tuple(xs // 16) + (F,), (1,) + tuple(xs // 8) + (F,)
    zsym = from_stream(range_decode(strings[1], _channel_rows(zshape, model), eb['cdf'], eb['cdf_size'], eb['offset']), zshape, model)
    gcdf, gsize, goff = model['gc']
    idx = np.asarray(indexes, np.int32).reshape(yshape)
    ysym = from_stream(range_decode(strings[0], to_stream(idx, model), gcdf, gsize, goff), yshape, model)
    return ysym, zsym


def decompress_block(model, strings, x_shape, run=None, indexes=None):
    """x_shape: (D,H,W).  Returns x_hat (D,H,W) float32 (unclipped), debug.
    `indexes` (NDHWC int32), when given, replaces the oracle's own scale indexes for the y stream: a decoder whose
    sigma_hat differs in the last bit on a table boundary cannot parse the stream at all, so parity tests that compare
    against an fp32-noise-different implementation hand over the encoder's indexes after checking them (tests/_stagecheck.py)."""
    run = run or run_transform
    cfg = CONFIGS[model['config']]
    P, F = model['params'], cfg['F']
    eb = model['eb']
    xs = np.asarray(x_shape)
    if cfg['v'] == 1:
        yshape = (1,) + tuple(xs // 8) + (F,)
        sym = from_stream(range_decode(strings[0], _channel_rows(yshape, model), eb['cdf'], eb['cdf_size'], eb['offset']), yshape, model)
        y_hat = (sym.astype(np.float32) + eb['medians']).astype(np.float32)
        dbg = dict(symbols=sym)
    else:
        zshape = (1,) + tuple(xs // 16) + (F,)
        yshape = (1,) + tuple(xs // 8) + (F,)
        zsym = from_stream(range_decode(strings[1], _channel_rows(zshape, model), eb['cdf'], eb['cdf_size'], eb['offset']), zshape, model)
        z_hat = (zsym.astype(np.float32) + eb['medians']).astype(np.float32)
        sigma = np.asarray(run('HyperSynthesisTransform', F, P, 'hyper_synthesis', z_hat), np.float32)
        own_idx = scale_index(sigma, model['scale_table'])
        idx = own_idx if indexes is None else np.asarray(indexes, np.int32).reshape(yshape)
        gcdf, gsize, goff = model['gc']
        ysym = from_stream(range_decode(strings[0], to_stream(idx, model), gcdf, gsize, goff), yshape, model)
        y_hat = ysym.astype(np.float32)
        dbg = dict(z_symbols=zsym, z_hat=z_hat, sigma_hat=sigma, indexes=idx, own_indexes=own_idx, symbols=ysym)
    x_hat = np.asarray(run(cfg['s'], F, P, 'synthesis', y_hat), np.float32)
    dbg.update(y_hat=y_hat, x_hat=x_hat)
    return x_hat[0, :, :, :, 0], dbg


# ---------------------------------------------------------------------------------------------------------------------
# D1 / D2 statistics of the adaptive threshold search with a STATED tie rule (round 4)
# ---------------------------------------------------------------------------------------------------------------------
def search_tallies_lowest_index(block, x_hat, thresholds):
    """Brute-force restatement of the per-threshold statistics of /root/reference/src/model_opt.py:33-56 with
    /root/reference/src/utils/pc_metric.py:76-131, for one block: rows (|B_t|, d1_sum_AB, d1_sum_BA, d2_sum_AB, d2_sum_BA) of
    the leading non-empty level sets B_t = argwhere(clip(x_hat) > thr[t]).  Where the reference takes whatever neighbour
    scipy's KD-tree returns among equidistant ones (pc_metric.py:114), this restatement -- like csrc/threshold_search.hip -- takes
    the one with the lowest (x, y, z) in lexicographic order.  block: (n, 6) = xyz + normals.  O(n |B_t|) memory: small cases."""
    a = np.asarray(block)[:, :3].astype(np.float64)
    n_a = np.asarray(block)[:, 3:6].astype(np.float32).astype(np.float64)       # (float32 normals, promoted like the product path)
    xh = np.clip(np.asarray(x_hat, np.float32), 0.0, 1.0)
    a_order = np.lexsort((a[:, 2], a[:, 1], a[:, 0]))                          # original points in (x, y, z) order
    rows = []
    for t in thresholds:
        b = np.argwhere(xh > np.float32(t)).astype(np.float64)                 # lexicographic (x, y, z) order
        if len(b) == 0:
            break
        d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)                     # exact integers in float64
        to_b = d.argmin(axis=1)                                                # first minimum = lowest (x, y, z)
        to_a = a_order[d[a_order].argmin(axis=0)]
        cnt = np.bincount(to_b, minlength=len(b)).astype(np.float64)
        acc = np.zeros((len(b), 3))
        np.add.at(acc, to_b, n_a)                                              # ascending point order, like pc_metric.py:16-18
        orphan = cnt == 0
        acc[orphan] = n_a[to_a[orphan]]
        cnt[orphan] = 1
        n_b = acc / cnt[:, None]
        gap_ab, gap_ba = a - b[to_b], b - a[to_a]
        rows.append((len(b), d.min(axis=1).sum(), d.min(axis=0).sum(),
                     (((gap_ab * n_b[to_b]).sum(1)) ** 2).sum(), (((gap_ba * n_a[to_a]).sum(1)) ** 2).sum()))
    return np.array(rows, np.float64).reshape(-1, 5)


def tie_free(block, x_hat, thresholds):
    """Per leading non-empty level set: True when every original point has ONE nearest decoded point and every decoded point ONE
    nearest original point -- the only case in which the reference's D2 (pc_metric.py:109-131) is defined without a tie rule."""
    a = np.asarray(block)[:, :3].astype(np.float64)
    xh = np.clip(np.asarray(x_hat, np.float32), 0.0, 1.0)
    out = []
    for t in thresholds:
        b = np.argwhere(xh > np.float32(t)).astype(np.float64)
        if len(b) == 0:
            break
        d = ((a[:, None, :] - b[None, :, :]) ** 2).sum(-1)
        out.append(bool(((d == d.min(axis=1, keepdims=True)).sum(axis=1) == 1).all() and ((d == d.min(axis=0, keepdims=True)).sum(axis=0) == 1).all()))
    return out
