"""Error model of the two-piece fp16 operand split (round 6) for the Winograd F(2x2,3x3)+z layer.

x = h + l,  h = fp16_rn(s x),  l = fp16_rn(s x - h)  (11 + 11 significand bits, s = a power of two per block / per layer);
all four product terms are kept (two K = 32 MFMAs: [Uh|Uh].[Vh|Vl] and [Ul|Ul].[Vh|Vl]); products of fp16 pieces are exact in
fp32, accumulation is modelled in fp64 (the fp32 accumulation error is the same for every operand form and is measured on the GPU).
Prints max |err| / (1 + max |ref|) against the fp64 direct convolution for: fp32 operands, 3-piece bf16 (six terms), 2-piece fp16.
"""
import numpy as np

rng = np.random.default_rng(0)
G = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], np.float64)
Bt = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], np.float64)
At = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], np.float64)


def bf16_rn(x):
    b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7fff + ((b >> 16) & 1)) & 0xffff0000
    return b.astype(np.uint32).view(np.float32).astype(np.float64)


def split_bf16(x):
    x = x.astype(np.float32).astype(np.float64)
    h = bf16_rn(x); m = bf16_rn(x - h); l = bf16_rn(x - h - m)
    return h, m, l


def split_f16(x, s):
    x = x.astype(np.float32).astype(np.float64) * s
    h = x.astype(np.float16).astype(np.float64)
    l = (x - h).astype(np.float16).astype(np.float64)
    return h / s, l / s


def run(D=10, HW=16, C=16, scale_in=1.0, relu=True, s_act=None):
    x = rng.standard_normal((D + 2, HW + 2, HW + 2, C)) * scale_in
    if relu:
        x = np.maximum(x, 0)
    x[0] = x[-1] = 0; x[:, 0] = x[:, -1] = 0; x[:, :, 0] = x[:, :, -1] = 0
    x = x.astype(np.float32).astype(np.float64)
    w = (rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32).astype(np.float64)
    # reference: direct conv in fp64
    ref = np.zeros((D, HW, HW, C))
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                ref += x[dz:dz + D, dy:dy + HW, dx:dx + HW] @ w[dz, dy, dx]
    U = np.einsum('py,qx,zyxio->zpqio', G, G, w)            # [dz][py][px][ci][co], computed in double, stored fp32 / pieces
    T = HW // 2
    # V[z][ty][tx][py][px][c]
    patches = np.zeros((D + 2, T, T, 4, 4, C))
    for ty in range(T):
        for tx in range(T):
            patches[:, ty, tx] = x[:, 2 * ty:2 * ty + 4, 2 * tx:2 * tx + 4]
    V = np.einsum('py,qx,ztuyxc->ztupqc', Bt, Bt, patches)
    V32 = V.astype(np.float32).astype(np.float64)           # the kernel computes V in fp32 (exact here up to fp32 rounding of sums)

    def wino(prod):
        M = np.zeros((D, T, T, 4, 4, C))
        for dz in range(3):
            M += prod(U[dz], V32[dz:dz + D])
        Y = np.einsum('ap,bq,ztupqo->ztuabo', At, At, M)
        out = np.zeros((D, HW, HW, C))
        for a in range(2):
            for b in range(2):
                out[:, a::2, b::2] = Y[:, :, :, a, b]
        return out

    ein = lambda u, v: np.einsum('pqio,ztupqi->ztupqo', u, v)
    res = {}
    U32 = U.astype(np.float32).astype(np.float64)
    res['fp32 operands'] = wino(lambda u, v: ein(u.astype(np.float32).astype(np.float64), v))
    def p_bf16(u, v):
        uh, um, ul = split_bf16(u); vh, vm, vl = split_bf16(v)
        return ein(uh, vh) + ein(um, vm) + ein(uh, vl) + ein(um, vh) + ein(ul, vh) + ein(uh, vm)
    res['bf16 x3, six terms'] = wino(p_bf16)
    umax = np.abs(U).max(); vmax = np.abs(V32).max()
    su = 2.0 ** (13 - np.ceil(np.log2(umax)))
    sv = s_act if s_act is not None else 2.0 ** (14 - np.ceil(np.log2(vmax)))
    def p_f16(terms):
        def f(u, v):
            uh, ul = split_f16(u, su); vh, vl = split_f16(v, sv)
            r = ein(uh, vh) + ein(uh, vl) + ein(ul, vh)
            return r + ein(ul, vl) if terms == 4 else r
        return f
    res['fp16 x2, hh+hl+lh (prescaled)'] = wino(p_f16(3))
    res['fp16 x2, four terms (prescaled)'] = wino(p_f16(4))
    sv_keep, su_keep = sv, su
    sv = 1.0; su = 1.0
    res['fp16 x2, four terms, s = 1'] = wino(p_f16(4))
    den = 1 + np.abs(ref).max()
    print(f'-- D={D} HW={HW} C={C} input scale {scale_in:g}: max|ref| {np.abs(ref).max():.3g}, su 2^{int(np.log2(su_keep))}, sv 2^{int(np.log2(sv_keep))}')
    for k, v in res.items():
        print(f'   {k:36s} max|err|/(1+max|ref|) = {np.abs(v - ref).max() / den:.3e}   rel-to-max|ref| {np.abs(v - ref).max() / np.abs(ref).max():.3e}')


if __name__ == '__main__':
    run()
    run(scale_in=30.0)
    run(scale_in=1e-3)
    run(scale_in=1e-6)
