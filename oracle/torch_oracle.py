"""Second, independent CPU restatement of the conv stack with PyTorch-CPU (oneDNN)  --  ORACLE /
TEST INFRASTRUCTURE ONLY (same import rules as oracle/oracle.py).

Purpose: (1) cross-check the naive C loops of pcc_oracle.c (two independent restatements of the TF
`SAME` rules must agree), (2) serve as the timed CPU baseline in bench.py (`cpu_baseline.kind =
"port"`): the reference's TF-1.15 Eigen/MKL CPU path cannot run here (requirements.txt:6-7 absent),
so its batch-1 per-block loop (/root/reference/src/model_types.py:192-198,224-230) is timed through
oneDNN convs instead.  **parity unpinned** (see oracle/oracle.py header).
"""
import numpy as np
import torch
import torch.nn.functional as F


def _same_pad(n, k, s):
    out = -(-n // s)
    tot = max((out - 1) * s + k - n, 0)
    return tot // 2, tot - tot // 2


def conv3d(x, w, b=None, stride=1, relu=False):
    """x (N,D,H,W,Cin) numpy/torch, w (k,k,k,Cin,Cout).  Explicit asymmetric SAME padding."""
    x = torch.as_tensor(x, dtype=torch.float32).permute(0, 4, 1, 2, 3)
    w = torch.as_tensor(w, dtype=torch.float32).permute(4, 3, 0, 1, 2)  # (Cout,Cin,kd,kh,kw)
    k = w.shape[2]
    pads = [_same_pad(n, k, stride) for n in x.shape[2:]]
    x = F.pad(x, (pads[2][0], pads[2][1], pads[1][0], pads[1][1], pads[0][0], pads[0][1]))
    y = F.conv3d(x, w, None if b is None else torch.as_tensor(b, dtype=torch.float32), stride=stride)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def conv3d_transpose(x, w, b=None, stride=1, relu=False):
    """x (N,D,H,W,Cin), w (k,k,k,Cout,Cin).  Full transposed conv, then crop [low, low + n*s)."""
    x = torch.as_tensor(x, dtype=torch.float32).permute(0, 4, 1, 2, 3)
    w = torch.as_tensor(w, dtype=torch.float32).permute(4, 3, 0, 1, 2)  # (Cin,Cout,kd,kh,kw)
    k = w.shape[2]
    y = F.conv_transpose3d(x, w, None, stride=stride)  # size (n-1)*s + k
    sl = []
    for n in x.shape[2:]:
        low, _ = _same_pad(n * stride, k, stride)
        sl.append(slice(low, low + n * stride))
    y = y[:, :, sl[0], sl[1], sl[2]]
    if b is not None:
        y = y + torch.as_tensor(b, dtype=torch.float32).view(1, -1, 1, 1, 1)
    if relu:
        y = F.relu(y)
    return y.permute(0, 2, 3, 4, 1).contiguous()


def run_transform(name, filters, params, prefix, x):
    from . import oracle as O  # layer lists are shared with the C-backed restatement
    t1 = None
    for i, (kind, cout, k, s, bias, relu, res) in enumerate(O.transform_layers(name, filters)):
        w = params[f'{prefix}/{i}/kernel']
        b = params.get(f'{prefix}/{i}/bias') if bias else None
        y = (conv3d if kind == 'conv' else conv3d_transpose)(x, w, b, stride=s, relu=relu)
        if res == 'save':
            t1 = y
        elif res == 'add':
            y = t1 + y
        x = y
    return x.numpy() if isinstance(x, torch.Tensor) else x


def _h(t):
    """round to fp16 (RTNE, what v_cvt_pk_f16_f32 does) and back"""
    return torch.as_tensor(t, dtype=torch.float32).to(torch.float16).to(torch.float32)


def run_transform_fp16(name, filters, params, prefix, x):
    """The fp16 mode of the BUILD (BASELINE.json configs[4]; the reference has no such mode) restated on the CPU: the same layer
    lists, every operand rounded to fp16 exactly where pcc_geo_cnn_v2_amd/csrc/network.hip + conv_f16.hip + the PCC_CONV_F16
    kernels of conv_mfma.hip round it, products accumulated in fp32 (oneDNN: another summation order than the MFMA chains):

      * a block (stride-2 layer with res = 'save' and 16 / 32 / 64 output channels on a grid of 16-multiples, followed by its two
        k3 stride-1 layers) keeps its two intermediate tensors in fp16: layer 0 rounds its output (after bias + ReLU), layer 1
        reads and writes fp16, layer 2 reads fp16, adds the fp16 residual in fp32 and writes fp32 -- or fp16 when its only consumer
        is the final 16 -> 1 layer;
      * 64-channel layers of a block: the products of input channels 0..31 are rounded to fp16 before those of 32..63 are added;
      * every other layer rounds both operands to fp16 at the matrix instruction and writes fp32; the Cin = 1 first layer computes
        in fp32 (its input is 0 / 1).
    Test infrastructure for tests/test_codec_gpu.py (stage check of the fp16 graph)."""
    from . import oracle as O
    layers = O.transform_layers(name, filters)
    x = torch.as_tensor(x, dtype=torch.float32)
    t1, left, final_in16 = None, 0, False
    for i, (kind, cout, k, s, bias, relu, res) in enumerate(layers):
        fn = conv3d if kind == 'conv' else conv3d_transpose
        w = torch.as_tensor(params[f'{prefix}/{i}/kernel'], dtype=torch.float32)
        b = params.get(f'{prefix}/{i}/bias') if bias else None
        cin = x.shape[-1]
        H, W = x.shape[2], x.shape[3]
        oH, oW = (2 * H, 2 * W) if kind == 'convT' else (H // 2, W // 2)
        last = i + 1 == len(layers)
        starts = (res == 'save' and s == 2 and k == 3 and cout in (16, 32, 64) and oH % 16 == 0 and oW % 16 == 0 and H % 2 == 0
                  and W % 2 == 0 and i + 2 < len(layers) and layers[i + 1][6] is None and layers[i + 2][6] == 'add'
                  and layers[i + 1][2:4] == (3, 1) and layers[i + 2][2:4] == (3, 1))
        in16 = out16 = False
        if starts:
            out16, left = True, 2
        elif left == 2:
            in16, out16, left = True, True, 1
        elif left == 1:
            in16, left = True, 0
            if i + 2 == len(layers) and layers[i + 1][0] == 'convT' and layers[i + 1][1:4] == (1, 3, 1) and cout == 16:
                out16 = final_in16 = True
        elif last and final_in16:
            in16 = True
        if in16 and cout == 64 and not last:        # conv_f16 C = 64: two input halves, fp16 partial sums in between
            if kind == 'conv':
                part = fn(x[..., :32], _h(w[..., :32, :]), None, s, False)
                y = _h(part) + fn(x[..., 32:], _h(w[..., 32:, :]), None, s, False)
            else:
                part = fn(x[..., :32], _h(w[..., :32]), None, s, False)
                y = _h(part) + fn(x[..., 32:], _h(w[..., 32:]), None, s, False)
            if b is not None:
                y = y + torch.as_tensor(b, dtype=torch.float32)
            if relu:
                y = F.relu(y)
        elif cin == 1:
            y = fn(x, w, b, s, relu)                 # conv_cin1_kernel: fp32 arithmetic
        else:
            y = fn(x if in16 else _h(x), _h(w), b, s, relu)
        if res == 'save':
            t1 = _h(y) if out16 else y               # the residual is what was stored
        elif res == 'add':
            y = t1 + y
        x = _h(y) if out16 else y
    return x.numpy()


def codec_block_roundtrip(model, x):
    """One block through the compress graph + the decompress graph + fixed-threshold extraction,
    batch 1, exactly the unit of work of SURVEY.md §8d.  Conv stacks via oneDNN, entropy coding and
    thresholding via the C oracle.  Returns (strings, n_points_enc, n_points_dec)."""
    from . import oracle as O
    thr = np.float32(np.linspace(0, 1.0, 256)[128])
    with torch.no_grad():
        # ---- compress graph (model_types.py:379-389 / :289-294); the reference's encoder also range-DEcodes the strings
        #      it has just written (:292, :383, :387) -- timed here as an extra decode of both streams
        strings, x_hat, dbg = O.compress_block(model, x, run=run_transform)
        O.decompress_entropy_only(model, strings, x.shape[1:4], dbg.get('indexes'))
        n_enc = len(O.threshold_argwhere(O.clip01(x_hat), thr))
        # ---- decompress graph (model_types.py:403-408 / :305-307)
        x_hat, _ = O.decompress_block(model, strings, x.shape[1:4], run=run_transform)
        n_dec = len(O.threshold_argwhere(x_hat, thr))
    return strings, n_enc, n_dec
