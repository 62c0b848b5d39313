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
