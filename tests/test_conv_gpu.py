"""GPU parity: HIP conv kernels (generic + MFMA) through the C ABI vs the CPU oracle."""
import ctypes

import numpy as np
import pytest
import torch

from pcc_geo_cnn_v2_amd import _lib as L
from pcc_geo_cnn_v2_amd import ops

pytestmark = pytest.mark.gpu

# stated fp32 tolerance for one conv layer: |gpu - oracle(double acc)| <= TOL * (1 + max|ref|)
TOL = 2e-5


def _run(ctx, O, N, D, H, W, cin, cout, k, s, tr, bias, relu, res, impl, seed=0, clip=False, flags=0, tol=None):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((N, D, H, W, cin)).astype(np.float32)
    x[rng.random(x.shape) < 0.3] = 0
    wshape = (k, k, k, cout, cin) if tr else (k, k, k, cin, cout)
    w = (rng.standard_normal(wshape) / np.sqrt(k ** 3 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    layer = ops.ConvLayer(w, b, s, tr, relu)
    ref = (O.conv3d_transpose if tr else O.conv3d)(x, w, b, s, relu)
    r = None
    if res:
        r = rng.standard_normal(ref.shape).astype(np.float32)
        ref = ref + r
    if clip:
        ref = np.clip(ref, 0, 1)
    xt = torch.from_numpy(x).to(ctx.device)
    rt = None if r is None else torch.from_numpy(r).to(ctx.device)
    if impl == L.PCC_IMPL_MFMA and not ops.mfma_supported(layer, x.shape):
        pytest.fail(f'MFMA path does not cover cin={cin} cout={cout} k={k} s={s} tr={tr} dims={(D, H, W)}')
    out = ops.conv3d(ctx, xt, layer, residual=rt, impl=impl, flags=(L.PCC_CONV_CLIP01 if clip else 0) | flags)
    torch.cuda.synchronize()
    got = out.cpu().numpy()
    assert got.shape == ref.shape
    err = np.abs(got - ref).max()
    bound = (TOL if tol is None else tol) * (1 + np.abs(ref).max())
    assert err <= bound, f'max err {err} > {bound}; first bad {np.argwhere(np.abs(got - ref) > bound)[:5]}'
    return got


GENERIC_CASES = [
    # N, D, H, W, cin, cout, k, s, tr
    (2, 8, 8, 8, 1, 1, 9, 2, False), (1, 8, 8, 8, 1, 2, 5, 2, False), (1, 7, 9, 6, 3, 5, 3, 1, False),
    (1, 7, 9, 6, 3, 5, 3, 2, False), (2, 1, 1, 1, 2, 2, 5, 2, True), (1, 4, 5, 3, 2, 1, 9, 2, True),
    (1, 5, 4, 6, 4, 3, 3, 1, True), (1, 5, 4, 6, 4, 3, 3, 2, True), (1, 8, 8, 8, 16, 16, 3, 1, False),
]


@pytest.mark.parametrize('case', GENERIC_CASES)
def test_generic_conv_matches_oracle(ctx, oracle, case):
    N, D, H, W, cin, cout, k, s, tr = case
    _run(ctx, oracle, N, D, H, W, cin, cout, k, s, tr, True, True, True, L.PCC_IMPL_GENERIC)
    _run(ctx, oracle, N, D, H, W, cin, cout, k, s, tr, False, False, False, L.PCC_IMPL_GENERIC, seed=1)


# every (cin, cout, k, stride, transposed) the c1/c2/c3/c3p graphs use, at the three row widths
MFMA_CASES = []
for dims in [(8, 16, 16), (6, 8, 8), (4, 4, 4), (4, 8, 32)]:
    for (cin, cout) in [(16, 16), (32, 32), (64, 64)]:
        MFMA_CASES.append((1,) + dims + (cin, cout, 3, 1, False))
        MFMA_CASES.append((1,) + dims + (cin, cout, 3, 1, True))
for dims in [(8, 16, 32), (4, 8, 16), (8, 8, 8)]:
    for (cin, cout) in [(16, 32), (32, 64), (64, 64), (32, 32)]:
        MFMA_CASES.append((1,) + dims + (cin, cout, 3, 2, False))
    MFMA_CASES.append((1,) + dims + (32, 32, 5, 2, False))
for dims in [(4, 8, 16), (3, 8, 8), (4, 4, 4), (2, 4, 32)]:
    for (cin, cout) in [(64, 64), (64, 32), (32, 16), (32, 32)]:
        MFMA_CASES.append((1,) + dims + (cin, cout, 3, 2, True))
    MFMA_CASES.append((1,) + dims + (32, 32, 5, 2, True))
for dims in [(8, 16, 32), (16, 16, 64)]:
    MFMA_CASES += [(2,) + dims + (1, 16, 3, 2, False), (1,) + dims + (1, 32, 9, 2, False),
                   (1,) + dims + (1, 16, 9, 2, False), (1,) + dims + (1, 32, 3, 2, False)]
for dims in [(4, 8, 8), (6, 10, 16)]:
    MFMA_CASES += [(2,) + dims + (16, 1, 3, 1, True), (1,) + dims + (32, 1, 3, 1, True), (1,) + dims + (32, 1, 9, 2, True)]


@pytest.mark.parametrize('case', MFMA_CASES)
def test_mfma_conv_matches_oracle(ctx, oracle, case):
    N, D, H, W, cin, cout, k, s, tr = case
    _run(ctx, oracle, N, D, H, W, cin, cout, k, s, tr, True, True, cout > 1, L.PCC_IMPL_MFMA, seed=3)


WINO_CASES = [
    # N, D, H, W, transposed, bias, relu, residual
    (2, 6, 16, 16, False, True, True, True), (1, 5, 32, 16, True, True, True, True), (3, 1, 16, 32, False, False, False, False),
    (1, 2, 16, 16, True, True, False, False), (1, 32, 16, 16, True, True, True, True),     # last: z split over workgroups
    (2, 9, 48, 32, False, False, True, True),
]


@pytest.mark.parametrize('ch', [16, 32, 64])
@pytest.mark.parametrize('case', WINO_CASES)
def test_winograd_conv_matches_oracle(ctx, oracle, case, ch):
    """conv_wino.hip: F(2x2,3x3) in x-y + direct z taps, same stated tolerance as the direct kernels.
    32 -> 32 / 64 -> 64 run as g x g sub-convolutions of 16 channels with in-place partial sums."""
    N, D, H, W, tr, bias, relu, res = case
    _run(ctx, oracle, N, D, H, W, ch, ch, 3, 1, tr, bias, relu, res, L.PCC_IMPL_WINOGRAD, seed=21)


def test_winograd_is_deterministic_batch_invariant_and_close_to_direct(ctx):
    rng = np.random.default_rng(12)
    w = (rng.standard_normal((3, 3, 3, 16, 16)) / 20).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(16).astype(np.float32), 1, True, True)
    x = torch.from_numpy(rng.standard_normal((5, 12, 32, 32, 16)).astype(np.float32)).to(ctx.device)
    a = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_WINOGRAD)
    b = ops.conv3d(ctx, x[3:4].contiguous(), layer, impl=L.PCC_IMPL_WINOGRAD)      # different z split / grid
    c = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_WINOGRAD)
    d = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA)
    e = ops.conv3d(ctx, x, layer)                                                  # AUTO picks Winograd here
    torch.cuda.synchronize()
    assert torch.equal(a, c) and torch.equal(a[3:4], b) and torch.equal(a, e)
    assert (a - d).abs().max().item() <= TOL * (1 + d.abs().max().item())
    with pytest.raises(AssertionError):
        ops.conv3d(ctx, x[:, :, :8, :8].contiguous(), layer, impl=L.PCC_IMPL_WINOGRAD)   # H, W not multiples of 16


@pytest.mark.parametrize('N,ch,D', [(8, 16, 64), (32, 16, 64), (8, 32, 32), (16, 64, 16)])
def test_winograd_at_the_real_launch_geometry_matches_oracle(ctx, oracle, N, ch, D):
    """The k3 stride-1 Conv3DTranspose layers of the c3p synthesis transform (/root/reference/src/model_transforms.py:78-80
    through :126-137) at the launch geometry of bench.py -- 64^3 x 16 ch (the dominant layer; N = 32 is exactly the bench launch:
    512 workgroups, 2 GiB of tensors), 32^3 x 32 ch, 16^3 x 64 ch -- bias + ReLU + residual, against the oneDNN restatement
    oracle/torch_oracle.py (the naive C loops would need minutes), which the test first pins to the C oracle on a slab."""
    from oracle import torch_oracle as T
    rng = np.random.default_rng(33)
    w = (rng.standard_normal((3, 3, 3, ch, ch)) / np.sqrt(27 * ch)).astype(np.float32)
    b = rng.standard_normal(ch).astype(np.float32)
    slab = rng.standard_normal((1, 3, 16, 16, ch)).astype(np.float32)
    a, c = T.conv3d_transpose(slab, w, b, 1, True).numpy(), oracle.conv3d_transpose(slab, w, b, 1, True)
    assert np.abs(a - c).max() <= 1e-5 * (1 + np.abs(c).max())          # the two restatements agree
    layer = ops.ConvLayer(w, b, 1, True, True)
    g = torch.Generator(device='cpu').manual_seed(5)
    worst = 0.0
    x = torch.randn((N, D, D, D, ch), generator=g)
    r = torch.randn((N, D, D, D, ch), generator=g)
    got = ops.conv3d(ctx, x.to(ctx.device), layer, residual=r.to(ctx.device), impl=L.PCC_IMPL_WINOGRAD)
    torch.cuda.synchronize()
    got = got.cpu()
    # 16-channel layers take the split-bf16 kernel (conv_wino_bf16.hip) by default: its error gate is 8e-6 (twice
    # the exact-fp32 Winograd kernel's 2.9-3.8e-6), and the exact-fp32 kernel (PCC_NO_SPLIT=1) is measured beside it
    import os
    got32 = None
    if ch == 16:
        with ctx.numerics_override(no_split=True):
            got32 = ops.conv3d(ctx, x.to(ctx.device), layer, residual=r.to(ctx.device), impl=L.PCC_IMPL_WINOGRAD).cpu()
    worst32, rel = 0.0, 0.0
    for n0 in range(0, N, 4):                      # the oracle in batches of 4 blocks (bounded host memory)
        ref = T.conv3d_transpose(x[n0:n0 + 4], w, b, 1, True) + r[n0:n0 + 4]
        err = (got[n0:n0 + 4] - ref).abs().max().item()
        assert err <= TOL * (1 + ref.abs().max().item()), (n0, err)
        if ch == 16:
            assert err <= 8e-6 * (1 + ref.abs().max().item()), (n0, err)
            worst32 = max(worst32, (got32[n0:n0 + 4] - ref).abs().max().item())
        worst = max(worst, err); rel = max(rel, err / (1 + ref.abs().max().item()))
    print(f'winograd {ch}ch @{D}^3 x{N}: max abs err {worst:.2e} ({rel:.2e} of 1 + max|ref|)' + (f'; exact-fp32 MFMA kernel {worst32:.2e}' if ch == 16 else ''))


@pytest.mark.parametrize('name,N,D,cin,cout,stride,res', [
    ('64->64 @16^3 direct split', 32, 16, 64, 64, 1, True), ('64->64 @8^3 direct split, 8-wide rows', 32, 8, 64, 64, 1, False),
    ('32->32 @16^3 direct split (32x32x16)', 32, 16, 32, 32, 1, True), ('64->32 stride-2 transposed split', 32, 16, 64, 32, 2, False),
    ('64->64 stride-2 transposed split @8^3', 32, 8, 64, 64, 2, False), ('32->16 stride-2 transposed march', 32, 32, 32, 16, 2, False)])
def test_split_layers_at_the_bench_launch_geometry_match_oracle(ctx, oracle, name, N, D, cin, cout, stride, res):
    """The layers of the c3p graph that round 4 moved onto the bf16 pipe besides the 16-channel Winograd ones, AUTO dispatch, at the batch
    of bench.py (the grids differ from the small parity cases: several rounds of workgroups, XCD remapping, z splits): every block against
    the oneDNN restatement (pinned to the C oracle on a slab) at the tolerance of the exact-fp32 kernels, bit-equal to a one-block launch,
    and not the exact-fp32 kernel's bits (the dispatch really takes the split kernel)."""
    from oracle import torch_oracle as T
    import os
    rng = np.random.default_rng(91)
    w = (rng.standard_normal((3, 3, 3, cout, cin)) / np.sqrt(27 * cin)).astype(np.float32)      # Conv3DTranspose layout
    b = rng.standard_normal(cout).astype(np.float32)
    slab = rng.standard_normal((1, 3, 8, 16, cin)).astype(np.float32)
    a, c = T.conv3d_transpose(slab, w, b, stride, True).numpy(), oracle.conv3d_transpose(slab, w, b, stride, True)
    assert np.abs(a - c).max() <= 1e-5 * (1 + np.abs(c).max())
    layer = ops.ConvLayer(w, b, stride, True, True)
    g = torch.Generator(device='cpu').manual_seed(6)
    x = torch.randn((N, D, D, D, cin), generator=g)
    O = D * stride
    r = torch.randn((N, O, O, O, cout), generator=g) if res else None
    xd, rd = x.to(ctx.device), None if r is None else r.to(ctx.device)
    got = ops.conv3d(ctx, xd, layer, residual=rd)
    one = ops.conv3d(ctx, xd[N - 1:].contiguous(), layer, residual=None if rd is None else rd[N - 1:].contiguous())
    assert torch.equal(got[N - 1:], one), 'the result depends on the batch / launch geometry'
    with ctx.numerics_override(no_split=True):
        exact = ops.conv3d(ctx, xd, layer, residual=rd)
    assert not torch.equal(got, exact), 'AUTO did not take a split kernel'
    got, exact = got.cpu(), exact.cpu()
    worst = worst32 = 0.0
    for n0 in range(0, N, 4):
        ref = T.conv3d_transpose(x[n0:n0 + 4], w, b, stride, True)
        if r is not None:
            ref = ref + r[n0:n0 + 4]
        scale = 1 + ref.abs().max().item()
        worst = max(worst, (got[n0:n0 + 4] - ref).abs().max().item() / scale)
        worst32 = max(worst32, (exact[n0:n0 + 4] - ref).abs().max().item() / scale)
    print(f'{name} x{N}: max err / (1 + max|ref|) split {worst:.2e}, exact-fp32 kernel {worst32:.2e}')
    assert worst <= TOL and worst32 <= TOL


def test_split_bf16_winograd_covers_the_fp32_exponent_range_and_is_deterministic(ctx, oracle, monkeypatch):
    """conv_wino_bf16.hip: three bf16 pieces per fp32 operand keep the full fp32 exponent range (unlike fp16 pieces): operands scaled
    by 2^-60 ... 2^+40 give the scaled result of the unscaled launch bit for bit (powers of two commute with every rounding in the
    path); the split path agrees with the exact-fp32 MFMA Winograd kernel to the tolerance both are held to against the oracle, is
    bit-deterministic, batch / launch-geometry invariant, and propagates a NaN only into the outputs its voxel reaches."""
    rng = np.random.default_rng(77)
    N, D, H, W, C = 3, 12, 32, 16, 16
    w = (rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((N, D, H, W, C)).astype(np.float32)).to(ctx.device)
    layer = ops.ConvLayer(w, b, 1, False, False)
    a = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_WINOGRAD)
    assert torch.equal(a, ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_WINOGRAD))
    assert torch.equal(a[1:2], ops.conv3d(ctx, x[1:2].contiguous(), layer, impl=L.PCC_IMPL_WINOGRAD))          # other grid, other z split
    ref = oracle.conv3d(x.cpu().numpy(), w, b, 1, False)
    ctx.set_numerics(no_split=True)
    f32 = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_WINOGRAD)
    ctx.set_numerics(no_split=False)
    bound = 8e-6 * (1 + np.abs(ref).max())
    assert np.abs(a.cpu().numpy() - ref).max() <= bound and np.abs(f32.cpu().numpy() - ref).max() <= bound
    assert not torch.equal(a, f32)                 # (two kernels, two summation orders: the env switch really switches)
    nobias = ops.ConvLayer(w, None, 1, False, False)
    base = ops.conv3d(ctx, x, nobias, impl=L.PCC_IMPL_WINOGRAD)
    for e in (-60, -20, 40):
        sx = ops.conv3d(ctx, x * (2.0 ** e), nobias, impl=L.PCC_IMPL_WINOGRAD)
        assert torch.equal(sx, base * (2.0 ** e)), e
    xn = x.clone(); xn[0, 5, 7, 9, 3] = float('nan')
    an = ops.conv3d(ctx, xn, layer, impl=L.PCC_IMPL_WINOGRAD)
    bad = torch.isnan(an).nonzero().cpu().numpy()
    assert len(bad) and (bad[:, 0] == 0).all() and (np.abs(bad[:, 1] - 5) <= 1).all() and (np.abs(bad[:, 2] - 7) <= 2).all() and (np.abs(bad[:, 3] - 9) <= 2).all()


def test_winograd_concat_offset(ctx, oracle):
    rng = np.random.default_rng(4)
    x = rng.standard_normal((1, 4, 16, 16, 16)).astype(np.float32)
    w = (rng.standard_normal((3, 3, 3, 16, 16)) / 20).astype(np.float32)
    layer = ops.ConvLayer(w, None, 1, False, False)
    out = torch.zeros((1, 4, 16, 16, 32), device=ctx.device)
    ops.conv3d(ctx, torch.from_numpy(x).to(ctx.device), layer, out=out, out_coffset=12, impl=L.PCC_IMPL_WINOGRAD)
    ref = oracle.conv3d(x, w)
    got = out.cpu().numpy()
    assert np.abs(got[..., 12:28] - ref).max() < 1e-4
    assert np.all(got[..., :12] == 0) and np.all(got[..., 28:] == 0)


TR2_SPLIT_CASES = [
    # N, D, H, W, cout, bias, relu    (Cin = 64; W = 8: rows of two 8-voxel lines; odd D / H: partial tiles; W = 32: two x tiles)
    (2, 16, 16, 16, 32, True, True), (1, 5, 7, 16, 32, False, False), (3, 8, 8, 8, 64, True, True), (1, 3, 11, 8, 64, True, False),
    (2, 4, 8, 32, 32, True, True), (1, 6, 16, 16, 64, False, True), (2, 7, 9, 8, 32, True, True),
]


@pytest.mark.parametrize('case', TR2_SPLIT_CASES)
def test_stride2_transposed_split_bf16_conv_matches_oracle(ctx, oracle, monkeypatch, case):
    """conv_tr2_split_kernel (conv_split.hip): Conv3DTranspose k3 stride 2, 64 -> 32 / 64 -> 64 (/root/reference/src/model_transforms.py:78
    inside :126-137) with three-piece bf16 operands -- against the C oracle at the tolerance of the exact-fp32 kernels; AUTO selects it,
    PCC_NO_SPLIT_TR2=1 gives the fp32 kernel (other bits, same tolerance); bit-deterministic, independent of batch."""
    N, D, H, W, cout, bias, relu = case
    got = _run(ctx, oracle, N, D, H, W, 64, cout, 3, 2, True, bias, relu, False, L.PCC_IMPL_AUTO, seed=53)
    rng = np.random.default_rng(53)
    x = torch.from_numpy(rng.standard_normal((N, D, H, W, 64)).astype(np.float32)).to(ctx.device)
    w = (rng.standard_normal((3, 3, 3, cout, 64)) / np.sqrt(27 * 64)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(cout).astype(np.float32) if bias else None, 2, True, relu)
    a = ops.conv3d(ctx, x, layer)
    assert torch.equal(a, ops.conv3d(ctx, x, layer)) and torch.equal(a[N - 1:], ops.conv3d(ctx, x[N - 1:].contiguous(), layer))
    ctx.set_numerics(no_split_tr2=True)
    b = ops.conv3d(ctx, x, layer)
    assert got.shape == (N, 2 * D, 2 * H, 2 * W, cout)
    assert not torch.equal(a, b), 'AUTO did not select the split kernel'
    assert (a - b).abs().max().item() <= 2e-5 * (1 + b.abs().max().item())
    out = torch.zeros((N, 2 * D, 2 * H, 2 * W, cout + 16), device=ctx.device)      # concat offset / channel stride
    ctx.set_numerics(no_split_tr2=False)
    ops.conv3d(ctx, x, layer, out=out, out_coffset=8)
    assert torch.equal(out[..., 8:8 + cout], a) and not out[..., :8].any() and not out[..., 8 + cout:].any()


TR2M_CASES = [
    # N, D, H, W, cin, cout, bias, relu   (H, W multiples of 16: conv_tr2m.hip; D = 1, odd D, z-split slabs, several x-y tiles)
    (1, 1, 16, 16, 32, 16, True, True), (2, 5, 16, 32, 32, 16, True, False), (1, 9, 32, 16, 64, 32, False, True),
    (3, 8, 16, 16, 64, 32, True, True), (1, 16, 48, 32, 32, 16, True, True), (2, 32, 16, 16, 32, 16, False, False),
    (1, 4, 32, 32, 64, 32, True, True),
]


@pytest.mark.parametrize('case', TR2M_CASES)
def test_marching_stride2_transposed_conv_matches_oracle(ctx, oracle, monkeypatch, case):
    """conv_tr2m.hip (round 3): Conv3DTranspose k3 stride 2 (/root/reference/src/model_transforms.py:78) marching along z with
    the accumulators of three output planes live -- against the C oracle (TF SAME crop 0 low / 1 high); bit-deterministic and
    independent of batch / z split; within the tolerance of the tiled conv_tr2g_kernel (same taps, another summation order)."""
    N, D, H, W, cin, cout, bias, relu = case
    ctx.set_numerics(tr2m=True)                     # also where AUTO would prefer the tiled kernel (64 -> 32 on short slabs)
    got = _run(ctx, oracle, N, D, H, W, cin, cout, 3, 2, True, bias, relu, False, L.PCC_IMPL_MFMA, seed=41)
    rng = np.random.default_rng(41)
    x = torch.from_numpy(rng.standard_normal((N, D, H, W, cin)).astype(np.float32)).to(ctx.device)
    w = (rng.standard_normal((3, 3, 3, cout, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(cout).astype(np.float32) if bias else None, 2, True, relu)
    a = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA)
    a2 = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA)
    one = ops.conv3d(ctx, x[N - 1:].contiguous(), layer, impl=L.PCC_IMPL_MFMA)       # other grid / z split
    ctx.set_numerics(tr2m=False)
    ctx.set_numerics(no_tr2m=True)
    b = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA)
    torch.cuda.synchronize()
    assert got.shape == (N, 2 * D, 2 * H, 2 * W, cout)
    assert torch.equal(a, a2) and torch.equal(a[N - 1:], one)
    assert not torch.equal(a, b) or D * H * W <= 256, 'PCC_TR2M did not select the marching kernel'
    assert (a - b).abs().max().item() <= 2e-5 * (1 + b.abs().max().item())


def test_marching_stride2_concat_offset_and_real_geometry(ctx):
    """channel-offset output (ocs / oco) and the bench's launch: 32 -> 16 @32^3 -> 64^3 x 32 blocks (256 workgroups, two 16-plane
    slabs per column) against the generic reference-order kernel."""
    rng = np.random.default_rng(8)
    w = (rng.standard_normal((3, 3, 3, 16, 32)) / np.sqrt(27 * 32)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(16).astype(np.float32), 2, True, True)
    x = torch.randn((1, 3, 16, 16, 32), generator=torch.Generator().manual_seed(1)).to(ctx.device)
    out = torch.zeros((1, 6, 32, 32, 24), device=ctx.device)
    ops.conv3d(ctx, x, layer, out=out, out_coffset=4)
    ref = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_GENERIC)
    assert (out[..., 4:20] - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())
    assert torch.all(out[..., :4] == 0) and torch.all(out[..., 20:] == 0)
    x = torch.randn((32, 32, 32, 32, 32), generator=torch.Generator().manual_seed(2)).to(ctx.device)
    a = ops.conv3d(ctx, x, layer)
    for n0 in (0, 13, 31):
        ref = ops.conv3d(ctx, x[n0:n0 + 1].contiguous(), layer, impl=L.PCC_IMPL_GENERIC)
        assert (a[n0:n0 + 1] - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())


@pytest.mark.parametrize('N,D,HW', [(32, 64, 64), (128, 33, 32), (64, 32, 32)])
def test_last_layer_32x32_columns_match_reference_order_kernel(ctx, monkeypatch, N, D, HW):
    """conv_cout1_mfma_kernel<T = 32> (round 3): Conv3DTranspose 16 -> 1 k3 s1 + bias + ReLU (the last synthesis layer,
    /root/reference/src/model_transforms.py:137) on 32 x 32 columns with z slabs -- the bench launch (64^3 x 32: 256 workgroups of
    two 32-plane slabs), an odd depth (slabs of 17 and 16 planes) -- against the generic reference-order kernel and the 16 x 16
    column variant; deterministic."""
    rng = np.random.default_rng(D)
    w = (rng.standard_normal((3, 3, 3, 1, 16)) / np.sqrt(27 * 16)).astype(np.float32)
    layer = ops.ConvLayer(w, np.array([0.1], np.float32), 1, True, True)
    x = torch.randn((N, D, HW, HW, 16), generator=torch.Generator().manual_seed(D)).to(ctx.device)
    a = ops.conv3d(ctx, x, layer)
    a2 = ops.conv3d(ctx, x, layer)
    ctx.set_numerics(cout1_t16=True)
    b = ops.conv3d(ctx, x, layer)
    ctx.set_numerics(cout1_t16=False)
    torch.cuda.synchronize()
    # same bits as the 16 x 16 columns (per output: channels -> (ky, kx) -> kz, whatever the column / slab geometry): encoder
    # and decoder may chunk differently and still have to agree on x_hat
    assert torch.equal(a, a2) and torch.equal(a, b)
    for n0 in (0, N // 2, N - 1):
        ref = ops.conv3d(ctx, x[n0:n0 + 1].contiguous(), layer, impl=L.PCC_IMPL_GENERIC)
        assert (a[n0:n0 + 1] - ref).abs().max().item() <= 2e-5 * (1 + ref.abs().max().item())


# stated tolerance of the fp16-MFMA mode (BASELINE.json configs[4]): operands rounded to fp16 (2^-11 relative), fp32
# accumulation over <= 27*64 products -> |err| <= 4e-3 * (1 + max|ref|) against the double-accumulation oracle
TOL_F16 = 4e-3
F16_CASES = [c for c in MFMA_CASES if c[4] % 16 == 0 and c[5] % 16 == 0]


@pytest.mark.parametrize('case', F16_CASES)
def test_fp16_mfma_conv_matches_oracle(ctx, oracle, case):
    N, D, H, W, cin, cout, k, s, tr = case
    got = _run(ctx, oracle, N, D, H, W, cin, cout, k, s, tr, True, True, True, L.PCC_IMPL_AUTO, seed=31, flags=L.PCC_CONV_F16, tol=TOL_F16)
    ref32 = _run(ctx, oracle, N, D, H, W, cin, cout, k, s, tr, True, True, True, L.PCC_IMPL_MFMA, seed=31)
    assert not np.array_equal(got, ref32), 'PCC_CONV_F16 did not select the fp16 kernels'


def test_mfma_batch_partial_tiles_and_plain(ctx, oracle):
    # dims that are not multiples of the tile (partial tiles), batch > 1, no bias/relu/residual
    _run(ctx, oracle, 3, 5, 10, 16, 16, 16, 3, 1, False, False, False, False, L.PCC_IMPL_MFMA, seed=5)
    _run(ctx, oracle, 2, 3, 5, 16, 32, 16, 3, 2, True, False, False, False, L.PCC_IMPL_MFMA, seed=6)
    _run(ctx, oracle, 2, 6, 10, 32, 16, 32, 3, 2, False, True, False, False, L.PCC_IMPL_MFMA, seed=7)
    _run(ctx, oracle, 1, 6, 10, 16, 16, 1, 3, 1, True, True, True, False, L.PCC_IMPL_MFMA, seed=8, clip=True)


def test_mfma_is_bit_deterministic_and_batch_invariant(ctx):
    """sigma_hat must not depend on batch position / launch geometry."""
    rng = np.random.default_rng(11)
    w = (rng.standard_normal((3, 3, 3, 64, 64)) / 40).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(64).astype(np.float32), 1, True, True)
    x = torch.from_numpy(rng.standard_normal((5, 8, 8, 8, 64)).astype(np.float32)).to(ctx.device)
    a = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA)
    b = ops.conv3d(ctx, x[3:4].contiguous(), layer, impl=L.PCC_IMPL_MFMA)
    c = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA)
    torch.cuda.synchronize()
    assert torch.equal(a, c)
    assert torch.equal(a[3:4], b)


def test_unknown_flag_bits_are_rejected(ctx):
    layer = ops.ConvLayer(np.zeros((3, 3, 3, 16, 16), np.float32), None, 1, False, False)
    x = torch.zeros((1, 4, 16, 16, 16), device=ctx.device)
    with pytest.raises(AssertionError):
        ops.conv3d(ctx, x, layer, flags=0x40000000)
    with pytest.raises(AssertionError):
        ops.conv3d(ctx, x, layer, flags=1 << 9)


def test_concat_mode_channel_offset(ctx, oracle):
    rng = np.random.default_rng(2)
    x = rng.standard_normal((1, 4, 4, 4, 2)).astype(np.float32)
    w = rng.standard_normal((3, 3, 3, 2, 3)).astype(np.float32)
    layer = ops.ConvLayer(w, None, 1, False, False)
    out = torch.zeros((1, 4, 4, 4, 8), device=ctx.device)
    ops.conv3d(ctx, torch.from_numpy(x).to(ctx.device), layer, out=out, out_coffset=4)
    ref = oracle.conv3d(x, w)
    got = out.cpu().numpy()
    assert np.abs(got[..., 4:7] - ref).max() < 1e-4
    assert np.all(got[..., :4] == 0) and np.all(got[..., 7] == 0)


@pytest.mark.parametrize('C,N,D,H,W,tr,res,relu', [(32, 3, 32, 32, 32, True, True, True), (64, 5, 16, 16, 16, True, True, True), (64, 2, 16, 32, 16, False, False, False),
                                                     (32, 1, 6, 16, 48, False, True, True), (64, 1, 7, 16, 16, True, True, True), (32, 2, 64, 16, 16, True, False, True),
                                                     (64, 32, 16, 16, 16, True, True, True)])
def test_winograd_cin_groups_inside_the_march(ctx, monkeypatch, C, N, D, H, W, tr, res, relu):
    """conv16_wino_cin_kernel (round 3): the cin groups of a 32- / 64-channel k3 stride-1 layer (/root/reference/src/
    model_transforms.py:62-81) inside ONE z march with the accumulators live across them -- no partial sums through `out`.
    Bit-deterministic and batch invariant; against the per-group launches (PCC_WINO_PER_GROUP: every group's accumulator is
    reduced separately and the reduced values are added, a different rounding order) and against the direct kernel within the
    Winograd tolerance."""
    rng = np.random.default_rng(C + D)
    w = (rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(C).astype(np.float32), 1, tr, relu)
    x = torch.randn((N, D, H, W, C), generator=torch.Generator().manual_seed(D)).to(ctx.device)
    r = torch.randn((N, D, H, W, C), generator=torch.Generator().manual_seed(D + 1)).to(ctx.device) if res else None
    a = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_WINOGRAD)
    a2 = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_WINOGRAD)
    one = ops.conv3d(ctx, x[N - 1:].contiguous(), layer, residual=None if r is None else r[N - 1:].contiguous(), impl=L.PCC_IMPL_WINOGRAD)
    ctx.set_numerics(wino_per_group=True)
    b = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_WINOGRAD)
    ctx.set_numerics(wino_per_group=False)
    d = ops.conv3d(ctx, x, layer, residual=r, impl=L.PCC_IMPL_MFMA)
    torch.cuda.synchronize()
    assert torch.equal(a, a2) and torch.equal(a[N - 1:], one)          # deterministic; independent of batch / z split
    scale = 1 + d.abs().max().item()
    assert (a - b).abs().max().item() <= 2e-5 * scale and (a - d).abs().max().item() <= 2e-5 * scale


F16S_CASES = [  # C, N, D, H, W, transposed, residual, out16
    (16, 2, 5, 16, 16, True, False, True), (16, 1, 9, 32, 48, False, True, True), (16, 2, 6, 16, 32, True, True, False),
    (32, 1, 5, 16, 16, True, False, True), (32, 2, 7, 32, 16, False, True, False), (16, 3, 32, 32, 32, True, True, True),
    (32, 2, 32, 32, 32, True, True, True), (16, 1, 1, 16, 16, False, False, True), (32, 1, 2, 48, 32, True, True, True),
    # C = 64: two input-half launches of the 32-channel kernel, fp16 partial sums between them
    (64, 1, 5, 16, 16, True, False, True), (64, 2, 16, 16, 16, True, True, False), (64, 1, 7, 32, 16, False, True, True),
    (64, 2, 32, 32, 32, True, True, False), (64, 1, 1, 16, 32, False, False, False),
]


@pytest.mark.parametrize('case', F16S_CASES)
def test_fp16_storage_conv_matches_oracle(ctx, case):
    """conv_f16.hip (PCC_CONV_IN16 / OUT16 / RES16: fp16 activations in HBM, v_mfma_f32_16x16x32_f16, fp32 accumulate) -- the
    k3 stride-1 layers of the blocks of /root/reference/src/model_transforms.py:62-81 in the fp16 mode (BASELINE.json configs[4]).
    Against the oneDNN restatement on the SAME fp16-rounded operands the only error left is the accumulation order (fp32 output:
    1e-5) plus the output rounding (fp16 output, and the fp16 partial sums of C = 64: 2^-11 relative); against unrounded operands
    the stated fp16 tolerance 4e-3."""
    from oracle import torch_oracle as T
    C, N, D, H, W, tr, res, out16 = case
    rng = np.random.default_rng(C + N + D)
    w = (rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    layer = ops.ConvLayer(w, b, 1, tr, True)
    x32 = rng.standard_normal((N, D, H, W, C)).astype(np.float32)
    r32 = rng.standard_normal((N, D, H, W, C)).astype(np.float32)
    x, r = torch.from_numpy(x32).half(), (torch.from_numpy(r32).half() if res else None)
    got = ops.conv3d_fp16_storage(ctx, x.to(ctx.device), layer, None if r is None else r.to(ctx.device), out16=out16)
    got2 = ops.conv3d_fp16_storage(ctx, x.to(ctx.device), layer, None if r is None else r.to(ctx.device), out16=out16)
    torch.cuda.synchronize()
    assert got.dtype == (torch.float16 if out16 else torch.float32) and torch.equal(got, got2)      # deterministic
    conv = T.conv3d_transpose if tr else T.conv3d
    ref_q = conv(x.float(), torch.from_numpy(w).half().float().numpy(), b, 1, True) + (r.float() if res else 0)
    ref = conv(x32, w, b, 1, True) + (torch.from_numpy(r32) if res else 0)
    g = got.float().cpu()
    scale = 1 + ref.abs().max().item()
    assert (g - ref_q).abs().max().item() <= (6e-4 if out16 or C == 64 else 1e-5) * scale
    assert (g - ref).abs().max().item() <= TOL_F16 * scale


@pytest.mark.parametrize('cin,cout,tr,D', [(1, 16, False, 32), (1, 32, False, 32), (16, 32, False, 32), (32, 64, False, 16), (64, 64, False, 8),
                                           (32, 16, True, 16), (64, 32, True, 8), (64, 64, True, 4), (32, 32, True, 4)])
def test_fp16_handover_of_the_stride2_layers(ctx, cin, cout, tr, D):
    """PCC_CONV_OUT16 on the first (stride-2) layer of an AnalysisBlock / SynthesisBlock (/root/reference/src/model_transforms.py:
    62-81): the fp16 tensor is the fp16 rounding of what the same kernel stores in fp32."""
    rng = np.random.default_rng(cin + cout)
    w = (rng.standard_normal((3, 3, 3, cout, cin) if tr else (3, 3, 3, cin, cout)) / np.sqrt(27 * cin)).astype(np.float32)
    layer = ops.ConvLayer(w, rng.standard_normal(cout).astype(np.float32), 2, tr, True)
    x = torch.randn((2, D, D, D, cin), generator=torch.Generator().manual_seed(D)).to(ctx.device)
    ref = ops.conv3d(ctx, x, layer, flags=L.PCC_CONV_F16)
    got = ops.conv3d_fp16_storage(ctx, x, layer, None, in16=False, out16=True)
    torch.cuda.synchronize()
    assert got.dtype == torch.float16
    if tr and (cin, cout) in ((32, 16), (64, 32)) and D % 16 == 0:
        # round 5: with the fp16 hand-over these layers march along z on v_mfma_f32_16x16x32_f16 (conv_tr2m_f16.hip): the same fp16
        # operands, another fp32 summation order than the tiled kernel that stores fp32 -> equal up to that and one output rounding
        assert (got.float() - ref).abs().max().item() <= 6e-4 * (1 + ref.abs().max().item())
    else:
        assert torch.equal(got, ref.half())


def test_fp16_storage_flags_are_checked(ctx):
    rng = np.random.default_rng(0)
    layer = ops.ConvLayer((rng.standard_normal((3, 3, 3, 16, 16)) / 20).astype(np.float32), None, 1, False, False)
    x = torch.zeros((1, 4, 16, 16, 16), dtype=torch.float16, device=ctx.device)
    d = layer.desc(1, 4, 16, 16, L.PCC_CONV_IN16)                 # fp16 storage outside the fp16 mode
    im = layer.device_images(ctx, d)
    out = torch.empty((1, 4, 16, 16, 16), device=ctx.device)
    args = lambda dd: (ctx.handle, ctypes.byref(dd), x.data_ptr(), im['w'].data_ptr(), im['pk'].data_ptr(), None, None, out.data_ptr(), None)
    assert L.lib().pcc_conv3d(*args(d)) == -1 and b'fp16 mode' in L.lib().pcc_last_error()
    d = layer.desc(1, 4, 24, 16, L.PCC_CONV_IN16 | L.PCC_CONV_F16)   # H not a multiple of 16
    assert L.lib().pcc_conv3d(*args(d)) == -1


SPLIT_CASES = [
    # N, D, H, W, C, tr, bias, relu, res
    (2, 8, 16, 16, 32, False, True, True, True), (1, 6, 8, 32, 32, True, False, False, False), (3, 5, 24, 16, 64, True, True, True, True),
    (1, 16, 16, 16, 64, False, True, False, False), (2, 3, 7, 16, 32, False, True, True, True),
    # 8-wide grids (64 channels only: the 8^3 layers of analysis block 3 and the hyper transforms): an MFMA row = 2 lines of 8 voxels
    (2, 8, 8, 8, 64, False, True, True, True), (1, 5, 12, 8, 64, True, True, False, False), (3, 3, 5, 8, 64, True, False, True, True),
]


@pytest.mark.parametrize('case', SPLIT_CASES)
def test_direct_split_bf16_conv_matches_oracle(ctx, oracle, case):
    """conv_split.hip (direct k3 stride-1 convolution, split-bf16 operands on the bf16 MFMA pipe, Cin = Cout in {32, 64}) against the
    C oracle: partial tiles in z and y, every epilogue flag, forward and (flipped) transposed layers; the same tolerance as the
    exact-fp32 MFMA kernels."""
    N, D, H, W, C, tr, bias, relu, res = case
    _run(ctx, oracle, N, D, H, W, C, C, 3, 1, tr, bias, relu, res, L.PCC_IMPL_SPLIT, seed=31)


def test_direct_split_bf16_conv_is_deterministic_geometry_invariant_and_as_accurate_as_fp32(ctx, oracle):
    rng = np.random.default_rng(41)
    C = 32
    w = (rng.standard_normal((3, 3, 3, C, C)) / np.sqrt(27 * C)).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    x = torch.from_numpy(rng.standard_normal((3, 10, 16, 32, C)).astype(np.float32)).to(ctx.device)
    layer = ops.ConvLayer(w, b, 1, False, True)
    a = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_SPLIT)
    assert torch.equal(a, ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_SPLIT))
    assert torch.equal(a[2:3], ops.conv3d(ctx, x[2:3].contiguous(), layer, impl=L.PCC_IMPL_SPLIT))
    assert torch.equal(a[:, 2:8], ops.conv3d(ctx, x[:, 1:9].contiguous(), layer, impl=L.PCC_IMPL_SPLIT)[:, 1:7])      # other tile origin in z
    ref = oracle.conv3d(x.cpu().numpy(), w, b, 1, True)
    direct = ops.conv3d(ctx, x, layer, impl=L.PCC_IMPL_MFMA).cpu().numpy()
    e_split, e_f32 = np.abs(a.cpu().numpy() - ref).max(), np.abs(direct - ref).max()
    print(f'direct split-bf16 max err {e_split:.2e}, exact-fp32 direct kernel {e_f32:.2e}')
    assert e_split <= 8e-6 * (1 + np.abs(ref).max()) and e_split <= 3 * e_f32 + 1e-6
    nobias = ops.ConvLayer(w, None, 1, False, False)
    base = ops.conv3d(ctx, x, nobias, impl=L.PCC_IMPL_SPLIT)
    for e in (-50, 30):
        assert torch.equal(ops.conv3d(ctx, x * (2.0 ** e), nobias, impl=L.PCC_IMPL_SPLIT), base * (2.0 ** e)), e


def test_kernel_family_switches_are_context_state_not_per_call_environment(ctx, monkeypatch):
    """the switches that select the kernel family of a layer were read inconsistently (some per call, some cached in statics).
    They are one word on the context now: read from the PCC_* environment once at pcc_ctx_create, changed only through
    pcc_ctx_set_numerics; flipping the environment afterwards changes nothing; unknown bits are rejected."""
    fam, sw = ctx.numerics()
    assert fam >= 5 and sw == ctx.numerics_at_creation
    assert ctx.numerics_tag() == f'pcc_geo_cnn_v2_amd/k{fam}/sw{sw:04x}/fp32'
    rng = np.random.default_rng(7)
    w = (rng.standard_normal((3, 3, 3, 16, 16)) / np.sqrt(27 * 16)).astype(np.float32)
    layer = ops.ConvLayer(w, None, 1, False, False)
    x = torch.from_numpy(rng.standard_normal((1, 16, 16, 16, 16)).astype(np.float32)).to(ctx.device)
    a = ops.conv3d(ctx, x, layer)
    monkeypatch.setenv('PCC_NO_SPLIT', '1')                  # too late for this context: no effect
    assert torch.equal(a, ops.conv3d(ctx, x, layer)) and ctx.numerics()[1] == sw
    with ctx.numerics_override(no_split=True):
        assert ctx.numerics()[1] == sw | L.PCC_NUM['no_split'] and ctx.numerics_tag().endswith(f'/sw{sw | 1:04x}/fp32')
        b = ops.conv3d(ctx, x, layer)
    assert ctx.numerics()[1] == sw and not torch.equal(a, b) and (a - b).abs().max().item() < 1e-5
    fresh = ops.Context(0)                                    # a context created NOW sees the variable
    assert fresh.numerics()[1] & L.PCC_NUM['no_split'] and torch.equal(ops.conv3d(fresh, x, layer), b)
    fresh.close()
    import ctypes as C
    assert L.lib().pcc_ctx_set_numerics(ctx.handle, C.c_uint32(1 << 20)) == L.PCC_ERR_ARG


TR2M_F16_CASES = [
    # N, D, H, W, cin, cout, bias, relu   (H, W multiples of 16; D = 1, odd D, z-split slabs, several x-y tiles, both cout tiles of 64 -> 32)
    (1, 1, 16, 16, 32, 16, True, True), (2, 5, 16, 32, 32, 16, True, False), (1, 9, 32, 16, 64, 32, False, True),
    (3, 8, 16, 16, 64, 32, True, True), (1, 16, 48, 32, 32, 16, True, True), (2, 32, 16, 16, 32, 16, False, False),
    (1, 4, 32, 32, 64, 32, True, True), (8, 16, 32, 32, 32, 16, True, True),
]


@pytest.mark.parametrize('case', TR2M_F16_CASES)
def test_marching_stride2_transposed_conv_in_the_fp16_mode_matches_oracle(ctx, case):
    """conv_tr2m_f16.hip (round 5): Conv3DTranspose k3 stride 2, 32 -> 16 / 64 -> 32 (/root/reference/src/model_transforms.py:78 inside
    :126-137) in the fp16 mode of BASELINE.json configs[4] -- fp32 input, operands rounded to fp16 (RNE) at the matrix instruction
    (v_mfma_f32_16x16x32_f16), fp32 accumulation, fp16 hand-over.  Against the oneDNN restatement on the SAME fp16-rounded operands only
    the accumulation order and the output rounding are left (6e-4 of the scale, like conv_f16.hip's fp16 outputs); against unrounded
    operands the stated fp16 tolerance; bit-deterministic, independent of batch / z split; AUTO takes it (PCC_NO_TR2M: the tiled kernel,
    other bits, same tolerance); output channel stride / offset honoured."""
    from oracle import torch_oracle as T
    N, D, H, W, cin, cout, bias, relu = case
    rng = np.random.default_rng(cin + D + H)
    w = (rng.standard_normal((3, 3, 3, cout, cin)) / np.sqrt(27 * cin)).astype(np.float32)
    b = rng.standard_normal(cout).astype(np.float32) if bias else None
    layer = ops.ConvLayer(w, b, 2, True, relu)
    x32 = rng.standard_normal((N, D, H, W, cin)).astype(np.float32)
    x = torch.from_numpy(x32).to(ctx.device)
    got = ops.conv3d_fp16_storage(ctx, x, layer, None, in16=False, out16=True)
    got2 = ops.conv3d_fp16_storage(ctx, x, layer, None, in16=False, out16=True)
    one = ops.conv3d_fp16_storage(ctx, x[N - 1:].contiguous(), layer, None, in16=False, out16=True)
    with ctx.numerics_override(no_tr2m=True):
        tiled = ops.conv3d_fp16_storage(ctx, x, layer, None, in16=False, out16=True)
    torch.cuda.synchronize()
    assert got.dtype == torch.float16 and got.shape == (N, 2 * D, 2 * H, 2 * W, cout)
    assert torch.equal(got, got2) and torch.equal(got[N - 1:], one)
    xq, wq = torch.from_numpy(x32).half().float(), torch.from_numpy(w).half().float().numpy()
    ref_q = T.conv3d_transpose(xq, wq, b, 2, relu)
    ref = T.conv3d_transpose(x32, w, b, 2, relu)
    g = got.float().cpu()
    scale = 1 + ref.abs().max().item()
    assert (g - ref_q).abs().max().item() <= 6e-4 * scale, (g - ref_q).abs().max().item() / scale
    assert (g - ref).abs().max().item() <= TOL_F16 * scale
    assert (g - tiled.float().cpu()).abs().max().item() <= 1.2e-3 * scale
    assert not torch.equal(got, tiled) or D * H * W <= 256, 'AUTO did not take the marching fp16 kernel'
