"""Randomised parity sweep: Winograd path (and fp16 mode) vs the direct MFMA kernels over many shapes / grid splits."""
import sys, os, itertools
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    ch = int(rng.choice([16, 32, 64]))
    N = int(rng.choice([1, 2, 3, 5, 8, 17, 33]))
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 33, 64]))
    H = 16 * int(rng.integers(1, 4)); W = 16 * int(rng.integers(1, 4))
    if N * D * H * W * ch > 2.2e8: N = max(1, int(2.2e8 // (D * H * W * ch)))
    tr = bool(rng.integers(0, 2)); bias = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2)); res = bool(rng.integers(0, 2)); clip = bool(rng.integers(0, 4) == 0)
    w = (rng.standard_normal((3, 3, 3, ch, ch)) / np.sqrt(27 * ch)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(ch).astype(np.float32) if bias else None, 1, tr, relu)
    x = torch.randn((N, D, H, W, ch), device=ctx.device)
    r = torch.randn((N, D, H, W, ch), device=ctx.device) if res else None
    fl = L.PCC_CONV_CLIP01 if clip else 0
    a = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_WINOGRAD, flags=fl)
    d = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_MFMA, flags=fl)
    a2 = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_WINOGRAD, flags=fl)
    err = (a - d).abs().max().item(); ref = d.abs().max().item()
    ok = err <= 2e-5 * (1 + ref) and torch.equal(a, a2) and bool(torch.isfinite(a).all())
    n += 1; bad += (not ok)
    if not ok: print('FAIL', dict(ch=ch, N=N, D=D, H=H, W=W, tr=tr, bias=bias, relu=relu, res=res, clip=clip), err, ref)
print(f'{n} cases, {bad} failures')
