"""Randomised parity sweep: stride-2 transposed convs (conv_tr2g_kernel / conv_tr2_kernel) and the 16->1 layer (conv_cout1_mfma_kernel)
against the generic reference-order kernel over odd sizes, tile overhangs and every epilogue flag."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    kind = int(rng.integers(0, 5))
    cin, cout, s = [(64, 64, 2), (64, 32, 2), (32, 16, 2), (32, 32, 2), (16, 1, 1)][kind]
    N = int(rng.choice([1, 2, 3, 5, 9]))
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32]))
    H = int(rng.choice([4, 6, 8, 12, 16, 20, 24, 32, 40])); W = int(rng.choice([4, 8, 12, 16, 24, 32, 48]))
    if kind == 4: H, W = int(rng.choice([8, 16, 24, 33, 40, 64])), int(rng.choice([8, 16, 20, 32, 48, 64]))
    while N * D * H * W * max(cin, cout * s ** 3) > 1.2e8 and N > 1: N -= 1
    if N * D * H * W * max(cin, cout * s ** 3) > 1.2e8: D = max(1, D // 4)
    bias = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2)); res = bool(rng.integers(0, 4) == 0); clip = bool(rng.integers(0, 4) == 0)
    w = (rng.standard_normal((3, 3, 3, cout, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(cout).astype(np.float32) if bias else None, s, True, relu)
    x = torch.randn((N, D, H, W, cin), device=ctx.device)
    r = torch.randn(ops.conv_out_shape(layer, x.shape), device=ctx.device) if res else None
    fl = L.PCC_CONV_CLIP01 if clip else 0
    a = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_AUTO, flags=fl)
    a2 = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_AUTO, flags=fl)
    d = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_GENERIC, flags=fl)
    err = (a - d).abs().max().item(); ref = d.abs().max().item()
    ok = err <= 2e-5 * (1 + ref) and torch.equal(a, a2) and bool(torch.isfinite(a).all())
    n += 1; bad += (not ok)
    if not ok: print('FAIL', dict(cin=cin, cout=cout, N=N, D=D, H=H, W=W, bias=bias, relu=relu, res=res, clip=clip), err, ref)
print(f'{n} cases, {bad} failures')
