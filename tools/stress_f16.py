"""Randomised parity sweep of conv_f16_kernel (fp16 storage, C in {16, 32, 64}) against the fp32 direct kernel on the same
fp16-rounded operands: only the accumulation order and the output rounding differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
bad = n = 0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 60):
    C = int(rng.choice([16, 32, 64]))
    N = int(rng.choice([1, 2, 3, 5, 8, 17]))
    D = int(rng.choice([1, 2, 3, 4, 5, 7, 8, 9, 16, 17, 31, 32, 64]))
    H = 16 * int(rng.integers(1, 4)); W = 16 * int(rng.integers(1, 4))
    while N * D * H * W * C > 1.5e8 and N > 1: N -= 1
    tr = bool(rng.integers(0, 2)); relu = bool(rng.integers(0, 2)); res = bool(rng.integers(0, 2)); out16 = bool(rng.integers(0, 2))
    w = torch.from_numpy((rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32)).half().float().numpy()
    b = rng.standard_normal(C).astype(np.float32)
    layer = ops.ConvLayer(w, b, 1, tr, relu)
    x = torch.randn((N, D, H, W, C), device=ctx.device).half()
    r = torch.randn((N, D, H, W, C), device=ctx.device).half() if res else None
    a = ops.conv3d_fp16_storage(ctx, x, layer, r, out16=out16)
    a2 = ops.conv3d_fp16_storage(ctx, x, layer, r, out16=out16)
    d = ops.conv3d(ctx, x.float(), layer, residual=None if r is None else r.float(), impl=L.PCC_IMPL_MFMA)
    err = (a.float() - d).abs().max().item(); ref = d.abs().max().item()
    tol = (6e-4 if (out16 or C == 64) else 1e-5) * (1 + ref)
    ok = err <= tol and torch.equal(a, a2) and bool(torch.isfinite(a.float()).all())
    n += 1; bad += (not ok)
    if not ok: print('FAIL', dict(C=C, N=N, D=D, H=H, W=W, tr=tr, relu=relu, res=res, out16=out16), err, ref)
print(f'{n} cases, {bad} failures')
