import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch
from pcc_geo_cnn_v2_amd import ops, _lib as L
from oracle import torch_oracle as T
ctx = ops.Context(0)
rng = np.random.default_rng(0)
for (C, N, D, H, W, tr, res, out16) in [(16,2,5,16,16,True,False,True),(16,1,9,32,48,False,True,True),(16,2,6,16,32,True,True,False),
                                         (32,1,5,16,16,True,False,True),(32,2,7,32,16,False,True,False),(16,3,32,32,32,True,True,True),(32,2,32,32,32,True,True,True)]:
    w = (rng.standard_normal((3,3,3,C,C))/np.sqrt(27*C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    layer = ops.ConvLayer(w, b, 1, tr, True)
    x = torch.from_numpy(rng.standard_normal((N,D,H,W,C)).astype(np.float32)).half()
    r = torch.from_numpy(rng.standard_normal((N,D,H,W,C)).astype(np.float32)).half() if res else None
    got = ops.conv3d_fp16_storage(ctx, x.to(ctx.device), layer, None if r is None else r.to(ctx.device), out16=out16)
    torch.cuda.synchronize()
    wq = torch.from_numpy(w).half().float().numpy()
    ref = (T.conv3d_transpose if tr else T.conv3d)(x.float(), wq, b, 1, True)
    if res: ref = ref + r.float()
    err = (got.float().cpu() - ref).abs().max().item()
    tol = 4e-3*(1+ref.abs().max().item())
    # against fp16-rounded operands the only error left is accumulation order + output rounding
    print(C, (N,D,H,W), 'tr' if tr else 'fw', 'res' if res else '', 'o16' if out16 else 'o32', 'err %.2e tol %.2e'%(err,tol), 'OK' if err<=tol else 'FAIL', got.dtype)

# ---- timing at the bench geometry (batch 32): fp16-storage kernel vs the fp32 Winograd path
for (C, N, D) in [(16, 32, 64), (32, 32, 32), (16, 8, 128)]:
    w = (rng.standard_normal((3,3,3,C,C))/np.sqrt(27*C)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(C).astype(np.float32), 1, True, True)
    xh = torch.randn((N,D,D,D,C), device=ctx.device).half(); rh = torch.randn((N,D,D,D,C), device=ctx.device).half()
    xf = xh.float(); rf = rh.float()
    def timeit(fn):
        fn(); torch.cuda.synchronize(); ts=[]
        for _ in range(5):
            e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True); e0.record()
            for _ in range(10): fn()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1)/10)
        return sorted(ts)[2]*1e3
    t16 = timeit(lambda: ops.conv3d_fp16_storage(ctx, xh, layer, rh, out16=True))
    t16o32 = timeit(lambda: ops.conv3d_fp16_storage(ctx, xh, layer, rh, out16=False))
    out = torch.empty_like(xf)
    t32 = timeit(lambda: ops.conv3d(ctx, xf, layer, residual=rf, out=out))
    nv = N*D**3
    print(f'C={C} N={N} D={D}: fp16 storage {t16:.1f} us ({nv*C*2*3/t16/1e6:.2f} TB/s algorithmic), fp16-in fp32-out {t16o32:.1f} us, fp32 Winograd {t32:.1f} us -> {t32/t16:.2f}x')
