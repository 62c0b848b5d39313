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
    return x


def codec_block_roundtrip(model, x):
    """One block through the compress graph + the decompress graph + fixed-threshold extraction,
    batch 1, exactly the unit of work of SURVEY.md §8d.  Conv stacks via oneDNN, entropy coding and
    thresholding via the C oracle.  Returns (strings, n_points_enc, n_points_dec)."""
    from . import oracle as O
    cfg = O.CONFIGS[model['config']]
    P, Fn, rm = model['params'], cfg['F'], model.get('round_mode', 0)
    eb = model['eb']
    thr = np.float32(np.linspace(0, 1.0, 256)[128])
    with torch.no_grad():
        # ---- compress graph (model_types.py:379-389 / :289-294)
        y = run_transform(cfg['a'], Fn, P, 'analysis', x)
        if cfg['v'] == 1:
            ch = np.broadcast_to(np.arange(Fn, dtype=np.int32), tuple(y.shape))
            sym, y_hat = O.quantize(y.numpy(), eb['medians'], rm)
            strings = (O.range_encode(sym, ch, eb['cdf'], eb['cdf_size'], eb['offset']),)
            _ = O.range_decode(strings[0], ch, eb['cdf'], eb['cdf_size'], eb['offset'])
        else:
            z = run_transform('HyperAnalysisTransform', Fn, P, 'hyper_analysis', y)
            ch = np.broadcast_to(np.arange(Fn, dtype=np.int32), tuple(z.shape))
            zsym, z_hat = O.quantize(z.numpy(), eb['medians'], rm)
            z_string = O.range_encode(zsym, ch, eb['cdf'], eb['cdf_size'], eb['offset'])
            _ = O.range_decode(z_string, ch, eb['cdf'], eb['cdf_size'], eb['offset'])
            sigma = run_transform('HyperSynthesisTransform', Fn, P, 'hyper_synthesis', z_hat)
            idx = O.scale_index(sigma.numpy(), model['scale_table'])
            ysym, y_hat = O.quantize(y.numpy(), None, rm)
            y_string = O.range_encode(ysym, idx, *model['gc'])
            _ = O.range_decode(y_string, idx, *model['gc'])
            strings = (y_string, z_string)
        x_hat = run_transform(cfg['s'], Fn, P, 'synthesis', y_hat)[0, :, :, :, 0].numpy()
        n_enc = len(O.threshold_argwhere(O.clip01(x_hat), thr))
        # ---- decompress graph (model_types.py:403-408 / :305-307)
        xs = np.array(x.shape[1:4])
        if cfg['v'] == 1:
            yshape = (1,) + tuple(xs // 8) + (Fn,)
            ch = np.broadcast_to(np.arange(Fn, dtype=np.int32), yshape)
            sym = O.range_decode(strings[0], ch, eb['cdf'], eb['cdf_size'], eb['offset'])
            y_hat = sym.astype(np.float32) + eb['medians']
        else:
            zshape = (1,) + tuple(xs // 16) + (Fn,)
            ch = np.broadcast_to(np.arange(Fn, dtype=np.int32), zshape)
            zsym = O.range_decode(strings[1], ch, eb['cdf'], eb['cdf_size'], eb['offset'])
            z_hat = zsym.astype(np.float32) + eb['medians']
            sigma = run_transform('HyperSynthesisTransform', Fn, P, 'hyper_synthesis', z_hat)
            idx = O.scale_index(sigma.numpy(), model['scale_table'])
            y_hat = O.range_decode(strings[0], idx, *model['gc']).astype(np.float32)
        x_hat = run_transform(cfg['s'], Fn, P, 'synthesis', y_hat)[0, :, :, :, 0].numpy()
        n_dec = len(O.threshold_argwhere(x_hat, thr))
    return strings, n_enc, n_dec
