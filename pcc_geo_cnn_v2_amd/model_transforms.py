"""Analysis / synthesis / hyper transforms -- the layer stacks of
/root/reference/src/model_transforms.py:11-169 (same class names, constructor arguments and call
convention), executed by the HIP kernels of libpcc_geo_hip.so instead of Keras/TensorFlow.

A layer is called on a torch tensor living on the GPU, `channels_last` (N,D,H,W,C) or
`channels_first` (N,C,D,H,W); internally everything is NDHWC (for C = 1, the codec's input/output,
the two layouts are the same bytes).
"""
from enum import Enum

import numpy as np
import torch

from . import _lib as L
from . import ops

relu = 'relu'  # stands for tf.nn.relu in the reference's signatures


def normalize_data_format(data_format):
    if data_format is None:
        return 'channels_last'  # Keras default image_data_format()
    assert data_format in ('channels_first', 'channels_last')
    return data_format


def get_channel_axis(data_format):
    return 1 if data_format == 'channels_first' else -1


def _is_relu(activation):
    if activation is None:
        return False
    if activation == 'relu' or getattr(activation, '__name__', '') == 'relu':
        return True
    raise ValueError(f'unsupported activation {activation!r}: the codec path uses ReLU or none')


def _to_ndhwc(t, data_format):
    if data_format == 'channels_first':
        return t.permute(0, 2, 3, 4, 1).contiguous()
    return t.contiguous()


def _from_ndhwc(t, data_format):
    if data_format == 'channels_first':
        return t.permute(0, 4, 1, 2, 3).contiguous()
    return t


class Layer:
    data_format = 'channels_last'

    def __call__(self, tensor, **kwargs):
        ctx = ops.get_context(tensor.device)
        x = _to_ndhwc(tensor.to(torch.float32), self.data_format)
        return _from_ndhwc(self.forward_ndhwc(ctx, x), self.data_format)

    def conv_layers(self):
        return []


class _ConvBase(Layer):
    transposed = False

    def __init__(self, filters, kernel_size, strides=(1, 1, 1), padding='valid', use_bias=True, activation=None,
                 data_format=None, **kwargs):
        ks = (kernel_size,) * 3 if isinstance(kernel_size, int) else tuple(kernel_size)
        st = (strides,) * 3 if isinstance(strides, int) else tuple(strides)
        assert ks[0] == ks[1] == ks[2] and st[0] == st[1] == st[2], 'cubic kernels / isotropic strides only'
        assert padding == 'same', "the codec only uses padding='same'"
        self.filters, self.k, self.stride = int(filters), int(ks[0]), int(st[0])
        self.use_bias, self.relu = bool(use_bias), _is_relu(activation)
        self.data_format = normalize_data_format(data_format)
        self.layer = None
        self.impl = L.PCC_IMPL_AUTO

    def build(self, cin, rng=None):
        """Keras default initialisers: glorot_uniform kernel, zero bias."""
        rng = rng if rng is not None else np.random.default_rng(42)
        rf = self.k ** 3
        limit = np.sqrt(6.0 / (rf * cin + rf * self.filters))
        shape = (self.k,) * 3 + ((self.filters, cin) if self.transposed else (cin, self.filters))
        kernel = rng.uniform(-limit, limit, shape).astype(np.float32)
        self.set_weights(kernel, np.zeros(self.filters, np.float32) if self.use_bias else None)

    def set_weights(self, kernel, bias=None):
        assert (bias is not None) == self.use_bias, 'bias presence does not match use_bias'
        self.layer = ops.ConvLayer(kernel, bias, self.stride, self.transposed, self.relu)
        assert self.layer.cout == self.filters and self.layer.k == self.k

    def forward_ndhwc(self, ctx, x, residual=None, flags=0, out=None, out_coffset=0):
        if self.layer is None:
            self.build(x.shape[-1])
        return ops.conv3d(ctx, x, self.layer, residual=residual, flags=flags, impl=self.impl, out=out,
                          out_coffset=out_coffset)

    def conv_layers(self):
        return [self]


class Conv3D(_ConvBase):
    transposed = False


class Conv3DTranspose(_ConvBase):
    transposed = True


class SequentialLayer(Layer):
    NET_ID = None        # PCC_NET_* id of the reference transform this class restates (set on the eight transforms)

    def __init__(self, layers, *args, **kwargs):
        self._layers = layers
        if layers:
            self.data_format = layers[0].data_format
        self._net = None

    @property
    def net_filters(self):
        # `filters` of the constructor = output channels of the last conv for the analysis / hyper transforms, input channels
        # of the first conv for the synthesis transforms
        convs = self.conv_layers()
        return convs[-1].filters if convs[-1].filters > 1 else convs[0].layer.cin

    def network(self):
        """ops.NetworkWeights of this transform (rebuilt when a layer's weights were replaced), or None when the stack is
        not the plain reference transform (no NET_ID, 'concat' residuals, a per-layer impl override, missing weights)."""
        import os
        if self.NET_ID is None or os.environ.get('PCC_LAYERWISE'):
            return None
        convs = self.conv_layers()
        if any(c.layer is None or c.impl != L.PCC_IMPL_AUTO for c in convs):
            return None
        if any(isinstance(l, ResidualLayer) and l.residual_mode != 'add' for l in self._layers):
            return None
        key = tuple(id(c.layer) for c in convs)
        if self._net is None or self._net[0] != key:
            self._net = (key, ops.NetworkWeights(self.NET_ID, self.net_filters, [c.layer for c in convs]))
        return self._net[1]

    def forward_ndhwc(self, ctx, x, final_flags=0):
        net = self.network() if not (final_flags & ~L.PCC_CONV_CLIP01) else None
        if net is not None:
            return ops.network_forward(ctx, net, x, final_flags)       # the whole transform in one ABI call
        for i, layer in enumerate(self._layers):
            if final_flags and i == len(self._layers) - 1:
                x = layer.forward_ndhwc(ctx, x, flags=final_flags)
            else:
                x = layer.forward_ndhwc(ctx, x)
        return x

    def conv_layers(self):
        return [c for layer in self._layers for c in layer.conv_layers()]


class ResidualLayer(Layer):
    def __init__(self, layers, residual_mode='add', data_format=None, *args, **kwargs):
        assert residual_mode in ('add', 'concat')
        self._layers = layers
        self.residual_mode = residual_mode
        self.data_format = normalize_data_format(data_format)

    def forward_ndhwc(self, ctx, x, flags=0):
        t1 = self._layers[0].forward_ndhwc(ctx, x)
        t = t1
        for layer in self._layers[1:-1]:
            t = layer.forward_ndhwc(ctx, t)
        last = self._layers[-1]
        if self.residual_mode == 'add':
            # tensor1 + tensor, fused into the epilogue of the last conv (ReLU is applied before the add)
            return last.forward_ndhwc(ctx, t, residual=t1, flags=flags)
        # tf.concat((tensor, tensor1), channel_axis): the last conv writes channels [0,F), tensor1 follows
        F = last.filters
        out = torch.empty(tuple(t1.shape[:4]) + (F + t1.shape[4],), dtype=torch.float32, device=x.device)
        last.forward_ndhwc(ctx, t, out=out, out_coffset=0)
        out[..., F:] = t1
        return out

    def conv_layers(self):
        return [c for layer in self._layers for c in layer.conv_layers()]


# The conv stacks of /root/reference/src/model_transforms.py as DATA, one row per conv in Keras construction order:
# (transposed, output channels ('F' = the constructor's `filters`), cubic kernel size (None = the constructor's), stride, use_bias, activated)
_STACKS = {
    'AnalysisTransformV1': [(0, 'F', 9, 2, 1, 1), (0, 'F', 5, 2, 1, 1), (0, 'F', 5, 2, 0, 0)],                 # :41-48
    'SynthesisTransformV1': [(1, 'F', 5, 2, 1, 1), (1, 'F', 5, 2, 1, 1), (1, 1, 9, 2, 1, 1)],                   # :51-59 (final activation: ReLU)
    'AnalysisBlock': [(0, 'F', None, 'S', 1, 1), (0, 'F', None, 1, 1, 1), (0, 'F', None, 1, 1, 1)],             # :62-70 ('S' = the block's strides)
    'SynthesisBlock': [(1, 'F', None, 'S', 1, 1), (1, 'F', None, 1, 1, 1), (1, 'F', None, 1, 1, 1)],            # :73-81
    'HyperAnalysisTransform': [(0, 'F', None, 1, 1, 1), (0, 'F', None, 2, 1, 1), (0, 'F', None, 1, 0, 0)],      # :140-147
    'HyperSynthesisTransform': [(1, 'F', None, 1, 1, 1), (1, 'F', None, 2, 1, 1), (1, 'F', None, 1, 1, 1)],     # :150-158
}


def _stack(name, filters, data_format, activation, kernel_size=None, strides=None):
    """The Conv3D / Conv3DTranspose objects of one row set of _STACKS."""
    convs = []
    for transposed, cout, k, stride, bias, act in _STACKS[name]:
        ks = kernel_size if k is None else (k, k, k)
        st = strides if stride == 'S' else (stride,) * 3
        convs.append((Conv3DTranspose if transposed else Conv3D)(filters if cout == 'F' else cout, ks, strides=st, padding='same',
                                                                 data_format=data_format, use_bias=bool(bias), activation=activation if act else None))
    return convs


class AnalysisTransformV1(SequentialLayer):
    NET_ID = L.PCC_NET_ANALYSIS_V1

    def __init__(self, filters, data_format=None, activation=relu, *args, **kwargs):
        super().__init__(_stack('AnalysisTransformV1', filters, normalize_data_format(data_format), activation), *args, **kwargs)


class SynthesisTransformV1(SequentialLayer):
    NET_ID = L.PCC_NET_SYNTHESIS_V1

    def __init__(self, filters, data_format=None, activation=relu, *args, **kwargs):
        super().__init__(_stack('SynthesisTransformV1', filters, normalize_data_format(data_format), activation), *args, **kwargs)


class AnalysisBlock(ResidualLayer):
    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), strides=(2, 2, 2), activation=relu, *args, **kwargs):
        data_format = normalize_data_format(data_format)
        super().__init__(_stack('AnalysisBlock', filters, data_format, activation, kernel_size, strides), *args, data_format=data_format, **kwargs)


class SynthesisBlock(ResidualLayer):
    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), strides=(2, 2, 2), activation=relu, *args, **kwargs):
        data_format = normalize_data_format(data_format)
        super().__init__(_stack('SynthesisBlock', filters, data_format, activation, kernel_size, strides), *args, data_format=data_format, **kwargs)


def _v2(block, first, out_layer, fs, data_format, kernel_size, activation, residual_mode):
    params = {'kernel_size': kernel_size, 'activation': activation, 'data_format': data_format,
              'residual_mode': residual_mode}
    return [block(f, **params) for f in fs] + [out_layer]


class AnalysisTransformV2(SequentialLayer):
    NET_ID = L.PCC_NET_ANALYSIS_V2

    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), activation=relu, residual_mode='add',
                 *args, **kwargs):
        data_format = normalize_data_format(data_format)
        last = Conv3D(filters, kernel_size, padding='same', use_bias=False, activation=None, data_format=data_format)
        super().__init__(_v2(AnalysisBlock, None, last, [filters // 2, filters, filters], data_format, kernel_size,
                             activation, residual_mode), *args, **kwargs)


class SynthesisTransformV2(SequentialLayer):
    NET_ID = L.PCC_NET_SYNTHESIS_V2

    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), activation=relu, residual_mode='add',
                 *args, **kwargs):
        data_format = normalize_data_format(data_format)
        last = Conv3DTranspose(1, kernel_size, padding='same', use_bias=True, activation=activation,
                               data_format=data_format)
        super().__init__(_v2(SynthesisBlock, None, last, [filters, filters, filters // 2], data_format, kernel_size,
                             activation, residual_mode), *args, **kwargs)


class AnalysisTransformProgressiveV2(SequentialLayer):
    NET_ID = L.PCC_NET_ANALYSIS_PROGRESSIVE_V2

    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), activation=relu, residual_mode='add',
                 *args, **kwargs):
        data_format = normalize_data_format(data_format)
        last = Conv3D(filters, kernel_size, padding='same', use_bias=False, activation=None, data_format=data_format)
        super().__init__(_v2(AnalysisBlock, None, last, [filters // 4, filters // 2, filters], data_format,
                             kernel_size, activation, residual_mode), *args, **kwargs)


class SynthesisTransformProgressiveV2(SequentialLayer):
    NET_ID = L.PCC_NET_SYNTHESIS_PROGRESSIVE_V2

    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), activation=relu, residual_mode='add',
                 *args, **kwargs):
        data_format = normalize_data_format(data_format)
        last = Conv3DTranspose(1, kernel_size, padding='same', use_bias=True, activation=activation,
                               data_format=data_format)
        super().__init__(_v2(SynthesisBlock, None, last, [filters, filters // 2, filters // 4], data_format,
                             kernel_size, activation, residual_mode), *args, **kwargs)


class HyperAnalysisTransform(SequentialLayer):
    NET_ID = L.PCC_NET_HYPER_ANALYSIS

    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), activation=relu, *args, **kwargs):
        super().__init__(_stack('HyperAnalysisTransform', filters, normalize_data_format(data_format), activation, kernel_size), *args, **kwargs)


class HyperSynthesisTransform(SequentialLayer):
    NET_ID = L.PCC_NET_HYPER_SYNTHESIS

    def __init__(self, filters, data_format=None, kernel_size=(3, 3, 3), activation=relu, *args, **kwargs):
        super().__init__(_stack('HyperSynthesisTransform', filters, normalize_data_format(data_format), activation, kernel_size), *args, **kwargs)


class TransformType(Enum):
    AnalysisTransformV1 = AnalysisTransformV1
    AnalysisTransformV2 = AnalysisTransformV2
    AnalysisTransformProgressiveV2 = AnalysisTransformProgressiveV2
    SynthesisTransformV1 = SynthesisTransformV1
    SynthesisTransformV2 = SynthesisTransformV2
    SynthesisTransformProgressiveV2 = SynthesisTransformProgressiveV2
    HyperAnalysisTransform = HyperAnalysisTransform
    HyperSynthesisTransform = HyperSynthesisTransform


def input_channels(transform, in_channels):
    """Input channel count of every conv of `transform` in conv_layers() order."""
    res = []

    def rec(layer, c):
        if isinstance(layer, _ConvBase):
            res.append(c)
            return layer.filters
        if isinstance(layer, ResidualLayer):
            c1 = rec(layer._layers[0], c)
            ct = c1
            for sub in layer._layers[1:]:
                ct = rec(sub, ct)
            return ct if layer.residual_mode == 'add' else ct + c1
        for sub in layer._layers:
            c = rec(sub, c)
        return c

    rec(transform, in_channels)
    return res


def init_transform(transform, in_channels, rng):
    for conv, cin in zip(transform.conv_layers(), input_channels(transform, in_channels)):
        conv.build(cin, rng)
    return transform


def get_weights(transform, prefix):
    out = {}
    for i, conv in enumerate(transform.conv_layers()):
        assert conv.layer is not None, 'transform has no weights yet'
        out[f'{prefix}/{i}/kernel'] = conv.layer.kernel
        if conv.use_bias:
            out[f'{prefix}/{i}/bias'] = conv.layer.bias
    return out


def set_weights(transform, prefix, params):
    for i, conv in enumerate(transform.conv_layers()):
        conv.set_weights(params[f'{prefix}/{i}/kernel'], params.get(f'{prefix}/{i}/bias') if conv.use_bias else None)
