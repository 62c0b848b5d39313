"""Debug helper for conv_wino_bf16.hip: split-bf16 Winograd kernel vs the direct fp32 MFMA kernel on structured operands.
   python tools/debug_wino_bf16.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
ctx = ops.Context(0)
rng = np.random.default_rng(0)
N, D, C = 2, 8, 16
def run(name, w, x, bias=None):
    layer = ops.ConvLayer(w.astype(np.float32), None if bias is None else bias.astype(np.float32), 1, False, False)
    xt = torch.from_numpy(x.astype(np.float32)).to(ctx.device)
    ref = ops.conv3d(ctx, xt, layer, impl=L.PCC_IMPL_MFMA).cpu().numpy()
    got = ops.conv3d(ctx, xt, layer, impl=L.PCC_IMPL_WINOGRAD).cpu().numpy()
    with ctx.numerics_override(no_split=True):
        f32 = ops.conv3d(ctx, xt, layer, impl=L.PCC_IMPL_WINOGRAD).cpu().numpy()
    e = np.abs(got - ref); e32 = np.abs(f32 - ref)
    print(f'{name}: max|ref| {np.abs(ref).max():.4g}  err split {e.max():.3e}  err fp32-wino {e32.max():.3e}  nan {np.isnan(got).sum()}')
    if e.max() > 1e-3 * (1 + np.abs(ref).max()):
        print('   err by cout      ', np.array2string(e.max(axis=(0, 1, 2, 3)), precision=2))
        print('   err by z         ', np.array2string(e.max(axis=(0, 2, 3, 4)), precision=2))
        print('   err by (y&1,x&1) ', [float(e[:, :, py::2, px::2].max()) for py in (0, 1) for px in (0, 1)])
        print('   err by batch     ', np.array2string(e.max(axis=(1, 2, 3, 4)), precision=2))
        i = np.unravel_index(np.argmax(e), e.shape); print('   worst', i, got[i], ref[i])
xi = rng.integers(-3, 4, (N, D, 16, 16, C)).astype(np.float64)
xr = rng.standard_normal((N, D, 16, 16, C))
wi = rng.integers(-2, 3, (3, 3, 3, C, C)).astype(np.float64)
wr = rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)
wc = np.zeros((3, 3, 3, C, C)); wc[1, 1, 1] = np.eye(C)
run('identity centre tap, int x ', wc, xi)
run('identity centre tap, real x', wc, xr)
wperm = np.zeros((3, 3, 3, C, C)); wperm[1, 1, 1] = np.roll(np.eye(C), 1, axis=1)
run('channel roll, int x        ', wperm, xi)
run('int w, int x               ', wi, xi)
run('int w, real x              ', wi, xr)
run('real w, int x              ', wr, xi)
run('real w, real x             ', wr, xr)
run('real w, real x, bias       ', wr, xr, rng.standard_normal(C))
