"""Randomised parity sweep of the split-bf16 kernels (round 4) against the exact-fp32 direct MFMA kernels: conv16_wino_bf16_kernel
(16-channel Winograd), conv_k3s1_split_kernel (PCC_IMPL_SPLIT, 32 / 64 channels, 16- and 8-wide rows), conv_tr2m_bf16_kernel (32 -> 16 stride-2
transposed), conv_tr2_split_kernel (64 -> 32 / 64 -> 64 stride-2 transposed);
odd depths, partial tiles, every epilogue flag, repeat launches bit-identical.   python tools/stress_split.py [seed] [cases]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = n = 0
worst = {}
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    kind = str(rng.choice(['wino16', 'split', 'split8', 'tr2m', 'tr2s']))
    N = int(rng.choice([1, 2, 3, 5, 8, 17]))
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32]))
    bias = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2))
    if kind == 'tr2m':
        cin, cout, s, tr, res, clip = 32, 16, 2, True, False, False
        H = 16 * int(rng.integers(1, 3)); W = 16 * int(rng.integers(1, 3)); D = min(D, 16)
        impl_a, impl_b = L.PCC_IMPL_AUTO, None            # B: the fp32 march (PCC_NO_SPLIT_TR2=1)
    elif kind == 'tr2s':
        cin, cout, s, tr, res, clip = 64, int(rng.choice([32, 64])), 2, True, False, False
        W = int(rng.choice([8, 16, 32])); H = int(rng.choice([1, 3, 8, 9, 16, 17])); D = min(D, 16)
        impl_a, impl_b = L.PCC_IMPL_AUTO, None            # B: conv_tr2g_kernel / the fp32 march (PCC_NO_SPLIT_TR2=1)
    elif kind == 'split8':
        cin = cout = 64
        s, tr, res, clip = 1, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), False
        W = 8; H = int(rng.choice([1, 3, 4, 7, 8, 9, 16])); D = min(D, 17)
        impl_a, impl_b = L.PCC_IMPL_SPLIT, L.PCC_IMPL_MFMA
    else:
        cin = cout = 16 if kind == 'wino16' else int(rng.choice([32, 64]))
        s, tr, res, clip = 1, bool(rng.integers(0, 2)), bool(rng.integers(0, 2)), bool(rng.integers(0, 4) == 0) and kind == 'wino16'
        W = 16 * int(rng.integers(1, 4))
        H = 16 * int(rng.integers(1, 4)) if kind == 'wino16' else int(rng.choice([3, 4, 7, 8, 16, 20, 32]))
        impl_a, impl_b = (L.PCC_IMPL_WINOGRAD if kind == 'wino16' else L.PCC_IMPL_SPLIT), L.PCC_IMPL_MFMA
    if N * D * H * W * max(cin, cout) * (8 if s == 2 else 1) > 2.0e8: N = max(1, int(2.0e8 // (D * H * W * max(cin, cout) * (8 if s == 2 else 1))))
    wshape = (3, 3, 3, cout, cin) if tr else (3, 3, 3, cin, cout)
    scale = float(2.0 ** rng.integers(-12, 12))          # operands far from 1: the split keeps fp32's exponent range
    w = (rng.standard_normal(wshape) / np.sqrt(27 * cin)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(cout).astype(np.float32) * np.float32(scale) if bias else None, s, tr, relu)
    x = torch.randn((N, D, H, W, cin), device=ctx.device) * scale
    oshape = ops.conv_out_shape(layer, x.shape)
    r = torch.randn(oshape, device=ctx.device) * scale if res else None
    fl = L.PCC_CONV_CLIP01 if clip else 0
    a = ops.conv3d(ctx, x, layer, residual=r, impl=impl_a, flags=fl)
    a2 = ops.conv3d(ctx, x, layer, residual=r, impl=impl_a, flags=fl)
    if impl_b is None:
        with ctx.numerics_override(no_split_tr2=True):
            d = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_AUTO, flags=fl)
    else:
        d = ops.conv3d(ctx, x, layer, residual=r, impl=impl_b, flags=fl)
    err = (a - d).abs().max().item(); ref = d.abs().max().item()
    tol = 2e-5 * (scale + ref)          # (clipped outputs are O(1) but carry the rounding of pre-clip values of size `scale`)
    ok = err <= tol and torch.equal(a, a2) and bool(torch.isfinite(a).all())
    worst[kind] = max(worst.get(kind, 0.0), err / (scale + ref))
    n += 1; bad += (not ok)
    if not ok: print('FAIL', kind, dict(N=N, D=D, H=H, W=W, cin=cin, tr=tr, bias=bias, relu=relu, res=res, clip=clip, scale=scale), err, ref)
print(f'{n} cases, {bad} failures; worst error / (scale + max|ref|) per kernel: ' + ', '.join(f'{k} {v:.2e}' for k, v in sorted(worst.items())))
