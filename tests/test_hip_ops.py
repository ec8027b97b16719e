"""-m gpu: every HIP operator, called through the C ABI, against torch CPU fp32 (primitives of the
oracle).  fp32 path: tight tolerance (exact-fp32 MFMA, different summation
order); bf16 path: inputs pre-rounded to bf16 on the CPU side, tolerance = bf16 output rounding."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from hip_helpers import (DEV, conv_desc, from_nhwc, lib, rounded, sync, tdtype, to_nhwc, tol, w_krsc)

pytestmark = pytest.mark.gpu
CODES = [0, 1]


def _conv_case(code, n, h, w, cin, cout, k, stride, pad, seed=0, relu=0, affine=False, residual=False):
    L = lib()
    g = torch.Generator().manual_seed(seed)
    x = rounded(torch.randn(n, cin, h, w, generator=g), code)
    wt = rounded(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, code)
    ref = F.conv2d(x, wt, stride=stride, padding=pad)
    scale = shift = res = None
    if affine:
        scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
        ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    if residual:
        res = rounded(torch.randn(ref.shape, generator=g), code)
        ref = ref + res
    if relu:
        ref = ref.relu()
    d = conv_desc(L, code, n, h, w, cin, cout, k, stride, pad, relu=relu)
    xd, wd = to_nhwc(x, code), w_krsc(wt, code)
    y = torch.full((n, ref.shape[2], ref.shape[3], cout), float("nan"), device=DEV, dtype=tdtype(code))
    sc = scale.to(DEV) if affine else None
    sh = shift.to(DEV) if affine else None
    rd = to_nhwc(res, code) if residual else None
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wd), L.ptr(sc), L.ptr(sh), L.ptr(rd), L.ptr(y), None, None))
    sync()
    got = from_nhwc(y)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, ref, **tol(code, ref.abs().max().item())), (got - ref).abs().max()


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [
    (2, 16, 16, 64, 64, 3, 1, 1),     # layer1-like
    (2, 16, 16, 64, 128, 3, 2, 1),    # stride-2 3x3
    (2, 16, 16, 64, 128, 1, 2, 0),    # 1x1 stride-2 shortcut
    (1, 32, 32, 32, 16, 3, 1, 1),     # small cout (decoder tail)
    (2, 32, 48, 16, 16, 3, 1, 1),     # cin 16 (half-empty bf16 chunk), non-square
    (3, 8, 8, 128, 128, 3, 1, 1),     # 8-wide tile path
    (2, 2, 2, 256, 64, 3, 1, 1),      # tiny spatial
    (1, 4, 4, 128, 256, 3, 2, 1),     # stride 2 onto 2x2
    (1, 24, 40, 96, 32, 3, 1, 1),     # ragged tiles (24x40 not multiples of the 8x16 tile)
])
def test_conv_fwd_matches_torch(code, shape):
    _conv_case(code, *shape)


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [
    (2, 32, 32, 64, 128, 3, 2, 1),    # 16x16 outputs: two 8x16 tiles per image
    (1, 40, 72, 64, 64, 3, 2, 1),     # ragged: 20x36 outputs
    (3, 16, 32, 96, 64, 3, 2, 1),     # 8x16 outputs, 3 chunks
])
def test_conv_stride2_wide_tiles(code, shape):
    """The 8 x 16-output tiles the stride-2 3x3 layers take in large launches (forced here by dropping the workgroup floor)."""
    L = lib()
    old = L.lib.vs_get_option(b"conv_min_wgs")
    L.set_option("conv_min_wgs", 1)
    try:
        _conv_case(code, *shape)
        _conv_case(code, *shape, relu=True, affine=True)
    finally:
        L.set_option("conv_min_wgs", old)


@pytest.mark.parametrize("code", CODES)
def test_conv_epilogue_affine_residual_relu(code):
    _conv_case(code, 2, 16, 16, 64, 64, 3, 1, 1, seed=3, relu=1, affine=True, residual=True)
    _conv_case(code, 1, 8, 8, 32, 32, 3, 1, 1, seed=4, relu=0, affine=True, residual=False)


@pytest.fixture
def force_direct_kernel():
    """Route every eligible layer (3x3 stride 1, Cin within one chunk, Cout <= 16) through the direct shallow-layer
    kernel, with short row chunks so strips start and end inside the image."""
    L = lib()
    old = (L.lib.vs_get_option(b"conv_direct_min_px"), L.lib.vs_get_option(b"conv_direct_rows"))
    L.set_option("conv_direct_min_px", 1)
    L.set_option("conv_direct_rows", 6)
    yield
    L.set_option("conv_direct_min_px", old[0])
    L.set_option("conv_direct_rows", old[1])


@pytest.mark.parametrize("code", CODES)
def test_conv_direct_kernel(code, force_direct_kernel):
    _conv_case(code, 2, 32, 48, 16, 16, 3, 1, 1, seed=11)                      # 12 tiles on 3 workgroups
    _conv_case(code, 3, 40, 24, 16, 12, 3, 1, 1, seed=12)                      # ragged strips, 12 couts
    _conv_case(code, 2, 32, 32, 16, 16, 3, 1, 1, seed=13, relu=1, affine=True, residual=True)
    if code == 1:
        _conv_case(code, 2, 32, 32, 32, 16, 3, 1, 1, seed=14)                  # full 32-channel chunk (bf16 only)
        _conv_case(code, 1, 48, 32, 24, 16, 3, 1, 1, seed=15)                  # channel tail inside the chunk


@pytest.mark.parametrize("code", [1, 2])
@pytest.mark.parametrize("case", [(2, 64, 96, 32, 1, 16, 16, 1), (3, 40, 44, 16, 0, 16, 16, 1), (1, 34, 30, 32, 1, 16, 12, 2), (2, 32, 32, 24, 0, 8, 16, 0),
                                  (1, 70, 16, 32, 1, 16, 16, 1)])
def test_direct_pair_equals_two_launches_bit_for_bit(code, case, force_direct_kernel):
    """conv_direct_pair_kernel - smp's last decoder block (Conv2dReLU(up(x)) -> Conv2dReLU, BatchNorm folded) as ONE evaluation-mode
    launch whose intermediate tensor never exists - against the two strip-kernel launches it replaces: EVERY output bit equal (same
    products, same order, the tensor in between rounded to the storage type on its way through LDS), and against torch CPU within
    the bf16 / fp16 tolerance.  Strip widths that do not divide the image (14-column strips), heights that do not divide the row
    chunks (force_direct_kernel: 6 rows), with and without the x2 upsampling of the source, 8 - 16 channels in between, ReLU /
    swish / no activation."""
    L = lib()
    n, h, w, c0, up, cmid, cout, relu = case
    g = torch.Generator().manual_seed(71 + h)
    x0 = rounded(torch.randn(n, c0, h >> up, w >> up, generator=g), code)
    w1 = rounded(torch.randn(cmid, c0, 3, 3, generator=g) / (c0 * 9) ** 0.5, code)
    w2 = rounded(torch.randn(cout, cmid, 3, 3, generator=g) / (cmid * 9) ** 0.5, code)
    s1, b1 = torch.rand(cmid, generator=g) + 0.5, torch.randn(cmid, generator=g) * 0.2
    s2, b2 = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.2
    act = {0: lambda t: t, 1: F.relu, 2: lambda t: t * torch.sigmoid(t)}[relu]
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up else x0
    mid = rounded(act(F.conv2d(xin, w1, padding=1) * s1[None, :, None, None] + b1[None, :, None, None]), code)
    ref = act(F.conv2d(mid, w2, padding=1) * s2[None, :, None, None] + b2[None, :, None, None])
    d1 = conv_desc(L, code, n, h, w, c0, cmid, 3, 1, 1, up0=up, relu=relu)
    d2 = conv_desc(L, code, n, h, w, cmid, cout, 3, 1, 1, relu=relu)
    assert L.lib.vs_conv2d_pair_ok(d1, d2) == 1 and L.lib.vs_conv2d_variant(d1) % 10 == 4 and L.lib.vs_conv2d_variant(d2) % 10 == 4
    x0d, w1d, w2d = to_nhwc(x0, code), w_krsc(w1, code), w_krsc(w2, code)
    s1d, b1d, s2d, b2d = s1.to(DEV), b1.to(DEV), s2.to(DEV), b2.to(DEV)
    midd = torch.full((n, h, w, cmid), float("nan"), device=DEV, dtype=tdtype(code))
    two = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
    one = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(d1, L.ptr(x0d), None, L.ptr(w1d), L.ptr(s1d), L.ptr(b1d), None, L.ptr(midd), None, None))
    L.check(L.lib.vs_conv2d_fwd(d2, L.ptr(midd), None, L.ptr(w2d), L.ptr(s2d), L.ptr(b2d), None, L.ptr(two), None, None))
    L.check(L.lib.vs_conv2d_pair_fwd(d1, d2, L.ptr(x0d), L.ptr(w1d), L.ptr(s1d), L.ptr(b1d), L.ptr(w2d), L.ptr(s2d), L.ptr(b2d), L.ptr(one), None))
    sync()
    assert torch.isfinite(one.float()).all()
    assert torch.equal(one.view(torch.int16), two.view(torch.int16)), (one.float() - two.float()).abs().max()
    assert torch.allclose(from_nhwc(one), ref, **tol(code, ref.abs().max().item())), (from_nhwc(one) - ref).abs().max()


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("cout", [16, 8])
def test_conv_direct_kernel_pooled_data_gradient(code, cout, force_direct_kernel):
    """The data gradient through nearest x2 upsampling on the strip kernel (MODE 1): a stride-1 convolution whose output is summed over
    2 x 2 pixel blocks and stored at half resolution (F.interpolate's backward), 16 and 8 output channels.  Against autograd."""
    import ctypes as C
    L = lib()
    g = torch.Generator().manual_seed(60 + cout)
    n, h, w, cmid = 2, 32, 48, 16
    x0 = rounded(torch.randn(n, cout, h // 2, w // 2, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(cmid, cout, 3, 3, generator=g) / (cout * 9) ** 0.5, code)
    y = F.conv2d(F.interpolate(x0, scale_factor=2, mode="nearest"), wt, padding=1)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    d = conv_desc(L, code, n, h, w, cmid, cout, 3, 1, 1)
    t = L.ConvTrain()
    t.pool0 = 1
    assert L.lib.vs_conv2d_train_variant(d, t) == 16294
    _, wtr = _prep_weights(L, code, wt)
    dyd = to_nhwc(dy, code)
    dx0 = torch.full((n, h // 2, w // 2, cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_train(d, L.ptr(dyd), None, L.ptr(wtr), None, L.ptr(dx0), None, C.byref(t), None))
    sync()
    assert torch.allclose(from_nhwc(dx0), x0.grad, **tol(code, x0.grad.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
def test_conv_direct_kernel_upsampled_input_and_head(code, force_direct_kernel):
    L = lib()
    g = torch.Generator().manual_seed(16)
    n, h, w, c0, cout = 2, 32, 32, 16, 16
    x0 = rounded(torch.randn(n, c0, h // 2, w // 2, generator=g), code)
    wt = rounded(torch.randn(cout, c0, 3, 3, generator=g) / 12, code)
    ref = F.conv2d(F.interpolate(x0, scale_factor=2, mode="nearest"), wt, padding=1)
    d = conv_desc(L, code, n, h, w, c0, cout, 3, 1, 1, up0=1)
    y = torch.empty((n, h, w, cout), device=DEV, dtype=tdtype(code))
    x0d, wd = to_nhwc(x0, code), w_krsc(wt, code)
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x0d), None, L.ptr(wd), None, None, None, L.ptr(y), None, None))
    sync()
    assert torch.allclose(from_nhwc(y), ref, **tol(code, ref.abs().max().item()))
    # segmentation head: bias, fp32 NCHW output, 3 classes
    x = rounded(torch.randn(n, 16, h, w, generator=g), code)
    wh = rounded(torch.randn(3, 16, 3, 3, generator=g) / 12, code)
    b = torch.randn(3, generator=g)
    refh = F.conv2d(x, wh, b, padding=1)
    dh = conv_desc(L, code, n, h, w, 16, 3, 3, 1, 1, out_f32=3)
    yh = torch.full((n, 3, h, w), float("nan"), device=DEV)
    xd, whd, bd = to_nhwc(x, code), w_krsc(wh, code), b.to(DEV)
    L.check(L.lib.vs_conv2d_fwd(dh, L.ptr(xd), None, L.ptr(whd), None, L.ptr(bd), None, L.ptr(yh), None, None))
    sync()
    assert torch.allclose(yh.cpu(), refh, rtol=1e-4, atol=1e-4 if code == 0 else 2e-2)


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 30, 40, 3), (1, 33, 17, 1), (3, 8, 50, 4), (2, 5, 16, 2)])
def test_conv_head_kernel_ragged_shapes(code, shape, force_direct_kernel):
    """conv_head_kernel (four output rows per MFMA tile): heights that are not multiples of the 4-row groups or the strip
    height, widths that are not multiples of 16, 1 - 4 classes, bias; fp32 NCHW logits against torch."""
    L = lib()
    n, h, w, k = shape
    g = torch.Generator().manual_seed(100 + h)
    x = rounded(torch.randn(n, 16, h, w, generator=g), code)
    wh = rounded(torch.randn(k, 16, 3, 3, generator=g) / 12, code)
    b = torch.randn(k, generator=g)
    ref = F.conv2d(x, wh, b, padding=1)
    d = conv_desc(L, code, n, h, w, 16, k, 3, 1, 1, out_f32=3)
    y = torch.full((n, k, h, w), float("nan"), device=DEV)
    xd, wd, bd = to_nhwc(x, code), w_krsc(wh, code), b.to(DEV)
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wd), None, L.ptr(bd), None, L.ptr(y), None, None))
    sync()
    assert torch.isfinite(y).all()
    assert torch.allclose(y.cpu(), ref, rtol=1e-4, atol=1e-4 if code == 0 else 2e-2), (y.cpu() - ref).abs().max()


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("k", [3, 1])
def test_conv_zero_stuffed_source_in_the_loader(code, k):
    """up0 = 2: the source is read at the even positions and as zeros elsewhere (the data gradient of a stride-2 conv is a
    stride-1 conv over the zero-stuffed output gradient, which is never materialised)."""
    L = lib()
    g = torch.Generator().manual_seed(40 + k)
    n, h, w, c0, cout = 2, 24, 32, 64, 64
    x0 = rounded(torch.randn(n, c0, h // 2, w // 2, generator=g), code)
    wt = rounded(torch.randn(cout, c0, k, k, generator=g) / 12, code)
    stuffed = torch.zeros(n, c0, h, w)
    stuffed[:, :, ::2, ::2] = x0
    ref = F.conv2d(stuffed, wt, padding=k // 2)
    d = conv_desc(L, code, n, h, w, c0, cout, k, 1, k // 2, up0=2)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
    xd, wd = to_nhwc(x0, code), w_krsc(wt, code)      # kept alive until the launch has run
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wd), None, None, None, L.ptr(y), None, None))
    sync()
    assert torch.allclose(from_nhwc(y), ref, **tol(code, ref.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
def test_conv_upsample_concat_never_materialised(code):
    """Decoder block input: cat(F.interpolate(x, 2, 'nearest'), skip) folded into the patch loader."""
    L = lib()
    g = torch.Generator().manual_seed(5)
    n, h, w, c0, c1, cout = 2, 16, 16, 64, 32, 32
    x0 = rounded(torch.randn(n, c0, h // 2, w // 2, generator=g), code)
    x1 = rounded(torch.randn(n, c1, h, w, generator=g), code)
    wt = rounded(torch.randn(cout, c0 + c1, 3, 3, generator=g) / 30, code)
    ref = F.conv2d(torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1), wt, padding=1)
    d = conv_desc(L, code, n, h, w, c0, cout, 3, 1, 1, c1=c1, up0=1)
    y = torch.empty((n, h, w, cout), device=DEV, dtype=tdtype(code))
    x0d, x1d, wd = to_nhwc(x0, code), to_nhwc(x1, code), w_krsc(wt, code)  # keep alive until the kernel ran
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x0d), L.ptr(x1d), L.ptr(wd), None, None, None, L.ptr(y), None, None))
    sync()
    assert torch.allclose(from_nhwc(y), ref, **tol(code, ref.abs().max().item()))
    # no skip (last decoder block)
    ref2 = F.conv2d(F.interpolate(x0, scale_factor=2, mode="nearest"), wt[:, :c0], padding=1)
    d2 = conv_desc(L, code, n, h, w, c0, cout, 3, 1, 1, up0=1)
    wd2 = w_krsc(wt[:, :c0].contiguous(), code)
    L.check(L.lib.vs_conv2d_fwd(d2, L.ptr(x0d), None, L.ptr(wd2), None, None, None, L.ptr(y), None, None))
    sync()
    assert torch.allclose(from_nhwc(y), ref2, **tol(code, ref2.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("classes", [1, 2, 3, 4, 7])
def test_head_conv_bias_fp32_nchw(code, classes):
    L = lib()
    g = torch.Generator().manual_seed(6)
    n, h, w = 2, 32, 32
    x = rounded(torch.randn(n, 16, h, w, generator=g), code)
    wt = rounded(torch.randn(classes, 16, 3, 3, generator=g) / 12, code)
    b = torch.randn(classes, generator=g)
    ref = F.conv2d(x, wt, b, padding=1)
    d = conv_desc(L, code, n, h, w, 16, classes, 3, 1, 1, out_f32=3)
    y = torch.full((n, classes, h, w), float("nan"), device=DEV)
    xd, wd, bd = to_nhwc(x, code), w_krsc(wt, code), b.to(DEV)
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wd), None, L.ptr(bd), None, L.ptr(y), None, None))
    sync()
    assert torch.allclose(y.cpu(), ref, rtol=1e-4, atol=1e-4 if code == 0 else 2e-2)


def _prep_weights(L, code, w_oihw):
    cout, cin, k, _ = w_oihw.shape
    wf = w_oihw.permute(0, 2, 3, 1).contiguous().to(DEV)
    wc = torch.empty(wf.shape, device=DEV, dtype=tdtype(code))
    wt = torch.empty((cin, k, k, cout), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_weights_prepare(code, L.ptr(wf), L.ptr(wc), L.ptr(wt), cout, k * k, cin, None))
    return wc, wt


@pytest.mark.parametrize("code", CODES)
def test_weights_prepare_layout(code):
    L = lib()
    w = rounded(torch.randn(40, 24, 3, 3), code)
    wc, wt = _prep_weights(L, code, w)
    sync()
    assert torch.equal(wc.float().cpu(), w.permute(0, 2, 3, 1))
    assert torch.equal(wt.float().cpu(), torch.flip(w, dims=(2, 3)).permute(1, 2, 3, 0))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64, 3, 1, 1), (2, 16, 16, 32, 128, 3, 2, 1),
                                   (2, 16, 16, 64, 128, 1, 2, 0), (1, 32, 32, 16, 16, 3, 1, 1)])
def test_dgrad_and_wgrad_match_autograd(code, shape):
    L = lib()
    n, h, w, cin, cout, k, stride, pad = shape
    g = torch.Generator().manual_seed(7)
    x = rounded(torch.randn(n, cin, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5, code).requires_grad_()
    y = F.conv2d(x, wt, stride=stride, padding=pad)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    ho, wo = y.shape[2:]
    # ---- wgrad ----
    d = conv_desc(L, code, n, h, w, cin, cout, k, stride, pad)
    ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((cout, k, k, cin), float("nan"), device=DEV)
    xd, dyd0 = to_nhwc(x.detach(), code), to_nhwc(dy, code)
    L.check(L.lib.vs_conv2d_wgrad(d, L.ptr(xd), None, L.ptr(dyd0), L.ptr(dw), L.ptr(ws), ws_bytes, None))
    sync()
    ref_dw = wt.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())
    # ---- dgrad = stride-1 conv of the (zero-stuffed) dy with flipped/transposed weights ----
    _, wtr = _prep_weights(L, code, wt.detach())
    dyd = to_nhwc(dy, code)
    if stride == 2:
        zs = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
        L.check(L.lib.vs_zero_stuff2x(code, L.ptr(dyd), L.ptr(zs), n, ho, wo, cout, None))
        dyd = zs
    dd = conv_desc(L, code, n, h, w, cout, cin, k, 1, pad)
    dx = torch.full((n, h, w, cin), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(dd, L.ptr(dyd), None, L.ptr(wtr), None, None, None, L.ptr(dx), None, None))
    sync()
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 128, 4, 1), (2, 16, 24, 64, 8, 1), (1, 8, 8, 64, 16, 2), (3, 8, 8, 96, 32, 1),
                                   (2, 32, 32, 128, 4, 2), (1, 20, 12, 32, 8, 1),
                                   (32, 64, 64, 128, 4, 2)])      # a training batch: enough workgroups for the wide stride-2 tiles
def test_grouped_conv_fwd_dgrad_wgrad(code, shape):
    """nn.Conv2d(c, c, 3, stride, 1, groups=c/cg) - the 3x3 convolution of a ResNeXt bottleneck (torchvision resnext50_32x4d:
    cg = 4 / 8 / 16 / 32 channels per group) - forward, data gradient and weight gradient against torch CPU.  The kernels
    run on 32-channel super-groups with the groups as diagonal blocks (vs_weights_prepare_grouped)."""
    L = lib()
    n, h, w, c, cg, stride = shape
    groups = c // cg
    g = torch.Generator().manual_seed(11)
    x = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(c, cg, 3, 3, generator=g) / (cg * 9) ** 0.5, code).requires_grad_()
    y = F.conv2d(x, wt, stride=stride, padding=1, groups=groups)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    ho, wo = y.shape[2:]
    w32 = wt.detach().permute(0, 2, 3, 1).contiguous().to(DEV)                # fp32 master layout [cout][taps][cg]
    wc = torch.full((c, 9, 32), float("nan"), device=DEV, dtype=tdtype(code))
    wtr = torch.full((c, 9, 32), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_weights_prepare_grouped(code, L.ptr(w32), L.ptr(wc), L.ptr(wtr), c, 9, cg, None))
    sync()
    # the expanded copy: the group's cg x cg block inside its super-group's 32 x 32 slab, zeros elsewhere
    wcf = wc.float().cpu()
    for o in (0, cg, c - 1):
        sub = (o % 32) // cg
        assert torch.equal(wcf[o, :, sub * cg:(sub + 1) * cg], wt.detach()[o].permute(1, 2, 0).reshape(9, cg))
        rest = wcf[o].clone()
        rest[:, sub * cg:(sub + 1) * cg] = 0
        assert not rest.any()
    d = conv_desc(L, code, n, h, w, c, c, 3, stride, 1, groups=groups)
    xd = to_nhwc(x.detach(), code)
    yd = torch.full((n, ho, wo, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wc), None, None, None, L.ptr(yd), None, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    # weight gradient, in the parameter's own [cout][taps][cg] layout
    ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((c, 3, 3, cg), float("nan"), device=DEV)
    dyd = to_nhwc(dy, code)
    L.check(L.lib.vs_conv2d_wgrad(d, L.ptr(xd), None, L.ptr(dyd), L.ptr(dw), L.ptr(ws), ws_bytes, None))
    sync()
    ref_dw = wt.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())
    # data gradient: stride-1 grouped conv of the (zero-stuffed) dy with the flipped / transposed copy
    if stride == 2:
        zs = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
        L.check(L.lib.vs_zero_stuff2x(code, L.ptr(dyd), L.ptr(zs), n, ho, wo, c, None))
        dyd = zs
    dd = conv_desc(L, code, n, h, w, c, c, 3, 1, 1, groups=groups)
    dx = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(dd, L.ptr(dyd), None, L.ptr(wtr), None, None, None, L.ptr(dx), None, None))
    sync()
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))


@pytest.mark.parametrize("case", [(24, 128, 128, 64, 0, 64, "res"), (17, 112, 144, 64, 0, 64, "affine"), (36, 64, 64, 128, 0, 128, "res"),
                                  (64, 32, 32, 256, 0, 256, "plain"), (36, 64, 64, 256, 128, 128, "affine"), (17, 128, 128, 128, 64, 64, "affine"),
                                  (4, 256, 256, 64, 64, 32, "affine"), (36, 96, 80, 64, 0, 32, "swish"), (128, 16, 16, 512, 0, 512, "res")])
@pytest.mark.parametrize("code", [1, 2])
def test_persistent_stream_conv_equals_the_tile_kernel_bit_for_bit(case, code):
    """conv_stream_kernel (csrc/conv_stream.h) - the persistent LDS-DMA form that serves the evaluation-mode 3x3 layers of large
    launches (prediction batches) - against conv_igemm_kernel on the same operands: EVERY output bit equal (same accumulation
    order, same epilogue arithmetic), so a slice's prediction cannot depend on which kernel its batch size selected; and
    against torch CPU within the bf16 tolerance.  Covers resident (64 input channels) and streamed weights, residual + ReLU,
    folded-BatchNorm scale / shift, swish, the decoder form (x2-upsampled tensor + skip tensor), ragged image sizes, 32- and
    64-wide cout tiles, and grids that do not divide by the workgroup count."""
    L = lib()
    n, h, w, c0, c1, cout, kind = case           # code 1: bf16; 2: fp16 (the inference precision runs on the same kernel)
    g = torch.Generator().manual_seed(23)
    up = 1 if c1 else 0
    x0 = rounded(torch.randn(n, c0, h >> up, w >> up, generator=g), code)
    x1 = rounded(torch.randn(n, c1, h, w, generator=g), code) if c1 else None
    xin = x0 if not c1 else torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1)
    wt = rounded(torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5, code)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    res = rounded(torch.randn(n, cout, h, w, generator=g), code) if kind == "res" else None
    ref = F.conv2d(xin, wt, padding=1)
    if kind != "plain":
        ref = ref * scale[None, :, None, None] + shift[None, :, None, None]
    if res is not None:
        ref = ref + res
    ref = ref * torch.sigmoid(ref) if kind == "swish" else (F.relu(ref) if kind != "plain" else ref)
    relu = 2 if kind == "swish" else (0 if kind == "plain" else 1)
    d = conv_desc(L, code, n, h, w, c0, cout, 3, 1, 1, c1=c1, up0=up, relu=relu)
    x0d, x1d, wd = to_nhwc(x0, code), (to_nhwc(x1, code) if c1 else None), w_krsc(wt, code)
    scd, shd = (scale.to(DEV), shift.to(DEV)) if kind != "plain" else (None, None)
    rd = to_nhwc(res, code) if res is not None else None
    outs = {}
    try:
        for stream in (1, 0):
            L.set_option("conv_stream", stream)
            yd = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
            L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x0d), L.ptr(x1d) if c1 else None, L.ptr(wd), L.ptr(scd) if scd is not None else None,
                                        L.ptr(shd) if shd is not None else None, L.ptr(rd) if rd is not None else None, L.ptr(yd), None, None))
            sync()
            outs[stream] = yd
            if stream:
                assert L.lib.vs_conv2d_variant(d) % 10 == 7, "the persistent kernel was not selected for this launch"
    finally:
        L.set_option("conv_stream", 1)
    assert torch.equal(outs[1].view(torch.int16), outs[0].view(torch.int16))
    assert torch.allclose(from_nhwc(outs[1]), ref, **tol(code, ref.abs().max().item()))


@pytest.mark.parametrize("case", [(2, 32, 32, 64, 64, 3, 1, 1, 0, 1), (3, 20, 28, 40, 48, 3, 1, 1, 0, 1), (2, 32, 32, 64, 128, 3, 2, 1, 0, 1),
                                  (2, 16, 16, 96, 32, 1, 1, 0, 0, 1), (1, 64, 64, 16, 16, 3, 1, 1, 0, 1), (2, 16, 16, 128, 64, 3, 1, 2, 0, 2),
                                  (2, 8, 8, 256, 128, 3, 1, 1, 64, 1), (8, 64, 64, 64, 64, 3, 1, 1, 0, 1), (2, 128, 128, 32, 16, 3, 1, 1, 0, 1)])
def test_fp16_conv_forward_with_eval_epilogue(case):
    """VS_F16 (BASELINE configs[4] names fp16; inference only): the implicit-GEMM kernels on v_mfma_f32_16x16x32_f16 with the
    evaluation-mode epilogue (folded BatchNorm scale / shift + ReLU), fp16 storage, fp32 accumulation - every tile family
    (64 / 128 / 256-pixel tiles, stride 2, 1x1, dilated, decoder form with x2 upsampling + concatenation, direct shallow-layer
    kernel, ragged channel counts) against torch CPU on the fp16-rounded operands.  Training epilogues are refused."""
    L = lib()
    code = 2
    n, h, w, cin, cout, k, stride, pad, c1, dil = case
    g = torch.Generator().manual_seed(17)
    up = 1 if c1 else 0
    x0 = rounded(torch.randn(n, cin, h >> up, w >> up, generator=g), code)
    x1 = rounded(torch.randn(n, c1, h, w, generator=g), code) if c1 else None
    xin = x0 if not c1 else torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1)
    wt = rounded(torch.randn(cout, cin + c1, k, k, generator=g) / ((cin + c1) * k * k) ** 0.5, code)
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g) * 0.1
    ref = F.relu(F.conv2d(xin, wt, stride=stride, padding=pad, dilation=dil) * scale[None, :, None, None] + shift[None, :, None, None])
    ho, wo = ref.shape[2:]
    d = conv_desc(L, code, n, h, w, cin, cout, k, stride, pad, c1=c1, up0=up, relu=1, dilation=dil if dil > 1 else 0)
    yd = torch.full((n, ho, wo, cout), float("nan"), device=DEV, dtype=torch.float16)
    x0d, x1d, wd, scd, shd = to_nhwc(x0, code), (to_nhwc(x1, code) if c1 else None), w_krsc(wt, code), scale.to(DEV), shift.to(DEV)
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(x0d), L.ptr(x1d) if c1 else None, L.ptr(wd), L.ptr(scd), L.ptr(shd), None, L.ptr(yd), None, None))
    sync()
    assert torch.allclose(from_nhwc(yd), ref, **tol(code, ref.abs().max().item()))
    # no training forms in fp16: the weight gradient is refused with an error, not computed in another precision
    ws = torch.empty(max(L.lib.vs_conv2d_wgrad_workspace(d), 16), dtype=torch.uint8, device=DEV)
    dw = torch.zeros(cout, k, k, cin + c1, device=DEV)
    rc = L.lib.vs_conv2d_wgrad(d, L.ptr(x0d), L.ptr(x1d) if c1 else None, L.ptr(yd), L.ptr(dw), L.ptr(ws), ws.numel(), None)
    assert rc != 0


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 512, 16, 2), (2, 8, 8, 1024, 32, 2), (1, 16, 24, 1024, 32, 4), (2, 32, 32, 512, 16, 2),
                                   (3, 20, 12, 64, 8, 2), (8, 32, 32, 1024, 32, 2)])
def test_grouped_dilated_conv_fwd_dgrad_wgrad(code, shape):
    """nn.Conv2d(c, c, 3, 1, padding=d, dilation=d, groups=32): what smp's replace_strides_with_dilation makes of the 3x3
    convolutions of resnext50_32x4d's layer3 / layer4 under DeepLabV3 (rates 2 and 4), DeepLabV3+ and PAN (layer4, rate 2) -
    forward, data gradient and weight gradient against torch CPU (the super-group kernels with the dilated patch)."""
    L = lib()
    n, h, w, c, cg, dil = shape
    groups = c // cg
    g = torch.Generator().manual_seed(13)
    x = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(c, cg, 3, 3, generator=g) / (cg * 9) ** 0.5, code).requires_grad_()
    y = F.conv2d(x, wt, stride=1, padding=dil, dilation=dil, groups=groups)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    w32 = wt.detach().permute(0, 2, 3, 1).contiguous().to(DEV)
    wc = torch.full((c, 9, 32), float("nan"), device=DEV, dtype=tdtype(code))
    wtr = torch.full((c, 9, 32), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_weights_prepare_grouped(code, L.ptr(w32), L.ptr(wc), L.ptr(wtr), c, 9, cg, None))
    d = conv_desc(L, code, n, h, w, c, c, 3, 1, dil, groups=groups, dilation=dil)
    xd = to_nhwc(x.detach(), code)
    yd = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wc), None, None, None, L.ptr(yd), None, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dw = torch.full((c, 3, 3, cg), float("nan"), device=DEV)
    dyd = to_nhwc(dy, code)
    L.check(L.lib.vs_conv2d_wgrad(d, L.ptr(xd), None, L.ptr(dyd), L.ptr(dw), L.ptr(ws), ws_bytes, None))
    sync()
    ref_dw = wt.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())
    dx = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(dyd), None, L.ptr(wtr), None, None, None, L.ptr(dx), None, None))
    sync()
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 128, 128), (1, 16, 24, 16, 16), (2, 4, 4, 64, 32), (1, 32, 32, 32, 32)])
def test_conv_transpose_4x4_stride2_as_conv3x3_plus_pixel_shuffle(code, shape):
    """nn.ConvTranspose2d(cin, cout, kernel_size=4, stride=2, padding=1) - smp Linknet's TransposeX2 - through the C ABI: weight
    expansion (vs_convt_weights_prepare), 3x3 convolution onto 4 * cout channels, vs_depth_to_space2 (+ bias); backward:
    vs_colsum (bias), vs_space_to_depth2, the 3x3 form's data and weight gradients, vs_convt_wgrad_gather - against autograd."""
    L = lib()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(5)
    x = rounded(torch.randn(n, cin, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(cin, cout, 4, 4, generator=g) / (cin * 4) ** 0.5, code).requires_grad_()
    bias = torch.randn(cout, generator=g).requires_grad_()
    y = F.conv_transpose2d(x, wt, bias, stride=2, padding=1)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    wc = torch.full((4 * cout, 9, cin), float("nan"), device=DEV, dtype=tdtype(code))
    wtr = torch.full((cin, 9, 4 * cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_convt_weights_prepare(code, L.ptr(wt.detach().contiguous().to(DEV)), L.ptr(wc), L.ptr(wtr), cin, cout, None))
    d = conv_desc(L, code, n, h, w, cin, 4 * cout, 3, 1, 1)
    xd = to_nhwc(x.detach(), code)
    zeff = torch.full((n, h, w, 4 * cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wc), None, None, None, L.ptr(zeff), None, None))
    yd = torch.full((n, 2 * h, 2 * w, cout), float("nan"), device=DEV, dtype=tdtype(code))
    bd = bias.detach().to(DEV)
    L.check(L.lib.vs_depth_to_space2(code, L.ptr(zeff), L.ptr(yd), n, h, w, cout, L.ptr(bd), None, None, 0, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    # backward
    dyd = to_nhwc(dy, code)
    ws_b = L.lib.vs_colsum_workspace(cout)
    wsb = torch.empty(ws_b, dtype=torch.uint8, device=DEV)
    db = torch.full((cout,), float("nan"), device=DEV)
    L.check(L.lib.vs_colsum(code, L.ptr(dyd), n * 4 * h * w, cout, L.ptr(db), L.ptr(wsb), ws_b, None))
    dzeff = torch.full((n, h, w, 4 * cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_space_to_depth2(code, L.ptr(dyd), L.ptr(dzeff), n, h, w, cout, None))
    ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=DEV)
    dense = torch.full((4 * cout, 9, cin), float("nan"), device=DEV)
    L.check(L.lib.vs_conv2d_wgrad(d, L.ptr(xd), None, L.ptr(dzeff), L.ptr(dense), L.ptr(ws), ws_bytes, None))
    dw = torch.full((cin, cout, 4, 4), float("nan"), device=DEV)
    L.check(L.lib.vs_convt_wgrad_gather(L.ptr(dense), L.ptr(dw), cin, cout, None))
    dd = conv_desc(L, code, n, h, w, 4 * cout, cin, 3, 1, 1)
    dx = torch.full((n, h, w, cin), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(dd, L.ptr(dzeff), None, L.ptr(wtr), None, None, None, L.ptr(dx), None, None))
    sync()
    assert torch.allclose(db.cpu(), bias.grad, rtol=1e-4, atol=1e-4 * bias.grad.abs().max().item())
    assert torch.allclose(dw.cpu(), wt.grad, rtol=1e-3, atol=1e-3 * wt.grad.abs().max().item())
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    # the shuffle and its inverse are exact permutations
    back = torch.empty_like(zeff)
    L.check(L.lib.vs_depth_to_space2(code, L.ptr(zeff), L.ptr(yd), n, h, w, cout, None, None, None, 0, None))
    L.check(L.lib.vs_space_to_depth2(code, L.ptr(yd), L.ptr(back), n, h, w, cout, None))
    sync()
    assert torch.equal(back, zeff)


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 64), (2, 8, 8, 128, 256), (1, 16, 24, 32, 32), (3, 4, 4, 256, 64), (32, 16, 16, 256, 512)])
def test_dilated_conv_fwd_dgrad_wgrad(code, shape):
    """nn.Conv2d(cin, cout, 3, stride 1, padding 2, dilation 2): what smp's replace_strides_with_dilation makes of ResNet's
    last stage for DeepLabV3+ (output stride 16) - forward, data gradient (the same kernel on the flipped / transposed copy)
    and weight gradient against torch CPU."""
    L = lib()
    n, h, w, cin, cout = shape
    g = torch.Generator().manual_seed(13)
    x = rounded(torch.randn(n, cin, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, code).requires_grad_()
    y = F.conv2d(x, wt, stride=1, padding=2, dilation=2)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    d = conv_desc(L, code, n, h, w, cin, cout, 3, 1, 2, dilation=2)
    xd, wd = to_nhwc(x.detach(), code), w_krsc(wt.detach(), code)
    yd = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(xd), None, L.ptr(wd), None, None, None, L.ptr(yd), None, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
    ws = torch.empty(max(ws_bytes, 16), dtype=torch.uint8, device=DEV)
    dw = torch.full((cout, 3, 3, cin), float("nan"), device=DEV)
    dyd = to_nhwc(dy, code)
    L.check(L.lib.vs_conv2d_wgrad(d, L.ptr(xd), None, L.ptr(dyd), L.ptr(dw), L.ptr(ws), ws_bytes, None))
    sync()
    ref_dw = wt.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())
    _, wtr = _prep_weights(L, code, wt.detach())
    dd = conv_desc(L, code, n, h, w, cout, cin, 3, 1, 2, dilation=2)
    dx = torch.full((n, h, w, cin), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(dd, L.ptr(dyd), None, L.ptr(wtr), None, None, None, L.ptr(dx), None, None))
    sync()
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
def test_wgrad_through_upsample_concat_and_split_dgrad(code):
    L = lib()
    g = torch.Generator().manual_seed(8)
    n, h, w, c0, c1, cout = 2, 16, 16, 64, 64, 32
    x0 = rounded(torch.randn(n, c0, h // 2, w // 2, generator=g), code).requires_grad_()
    x1 = rounded(torch.randn(n, c1, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(cout, c0 + c1, 3, 3, generator=g) / 30, code).requires_grad_()
    y = F.conv2d(torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1), wt, padding=1)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    d = conv_desc(L, code, n, h, w, c0, cout, 3, 1, 1, c1=c1, up0=1)
    ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
    dw = torch.empty((cout, 3, 3, c0 + c1), device=DEV)
    x0d, x1d, dyd = to_nhwc(x0.detach(), code), to_nhwc(x1.detach(), code), to_nhwc(dy, code)
    L.check(L.lib.vs_conv2d_wgrad(d, L.ptr(x0d), L.ptr(x1d), L.ptr(dyd), L.ptr(dw), L.ptr(ws), ws_bytes, None))
    sync()
    ref_dw = wt.grad.permute(0, 2, 3, 1)
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())
    # dgrad split across the concat: channels < c0 -> full-res "dup" (then 2x2 summed), >= c0 -> dskip
    _, wtr = _prep_weights(L, code, wt.detach())
    dd = conv_desc(L, code, n, h, w, cout, c0 + c1, 3, 1, 1, split_c=c0)
    dup = torch.empty((n, h, w, c0), device=DEV, dtype=tdtype(code))
    dskip = torch.empty((n, h, w, c1), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_conv2d_fwd(dd, L.ptr(dyd), None, L.ptr(wtr), None, None, None, L.ptr(dup), L.ptr(dskip), None))
    dx0 = torch.empty((n, h // 2, w // 2, c0), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_upsample2x_bwd(code, L.ptr(dup), L.ptr(dx0), n, h // 2, w // 2, c0, None))
    sync()
    assert torch.allclose(from_nhwc(dskip), x1.grad, **tol(code, x1.grad.abs().max().item()))
    assert torch.allclose(from_nhwc(dx0), x0.grad, **tol(code, 2 * x0.grad.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
def test_stem_fwd_and_wgrad(code):
    L = lib()
    g = torch.Generator().manual_seed(9)
    n, h, w = 2, 64, 96
    x = torch.randn(n, 1, h, w, generator=g).requires_grad_(False)
    wt0 = torch.randn(64, 1, 7, 7, generator=g) / 7
    # the bf16 path rounds the image and the weights to bf16 inside the kernels (bf16 MFMA): the reference sees the same
    # values; the device still receives the fp32 originals
    wt = rounded(wt0, code).requires_grad_()
    y = F.conv2d(rounded(x, code), wt, stride=2, padding=3)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    out = torch.empty((n, h // 2, w // 2, 64), device=DEV, dtype=tdtype(code))
    xd, wd, dyd = x.to(DEV), wt0.reshape(64, 49).to(DEV), to_nhwc(dy, code)
    L.check(L.lib.vs_stem_fwd(code, L.ptr(xd), L.ptr(wd), None, None, 0, L.ptr(out), n, h, w, None))
    sync()
    assert torch.allclose(from_nhwc(out), y.detach(), **tol(code, y.abs().max().item()))
    sc, sh = torch.rand(64) + 0.5, torch.randn(64)
    scd, shd = sc.to(DEV), sh.to(DEV)
    L.check(L.lib.vs_stem_fwd(code, L.ptr(xd), L.ptr(wd), L.ptr(scd), L.ptr(shd), 1, L.ptr(out), n, h, w, None))
    sync()
    ref = (y.detach() * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1)).relu()
    assert torch.allclose(from_nhwc(out), ref, **tol(code, ref.abs().max().item()))
    wsb = L.lib.vs_stem_wgrad_workspace(n, h, w)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dw = torch.full((64, 49), float("nan"), device=DEV)
    L.check(L.lib.vs_stem_wgrad(code, L.ptr(xd), L.ptr(dyd), L.ptr(dw), L.ptr(ws), wsb, n, h, w, None))
    sync()
    ref_dw = wt.grad.reshape(64, 49)
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item())


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("c,rows_shape", [(64, (2, 16, 16)), (16, (2, 32, 32)), (512, (3, 2, 2)), (128, (1, 8, 8)),
                                          (48, (2, 16, 16)), (304, (1, 8, 12))])      # channel counts that do not divide the block
def test_batchnorm_train_fwd_bwd(code, c, rows_shape):
    L = lib()
    g = torch.Generator().manual_seed(10)
    n, h, w = rows_shape
    x = rounded(torch.randn(n, c, h, w, generator=g) * 2 + 0.5, code).requires_grad_()
    res = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_()
    beta = torch.randn(c, generator=g).requires_grad_()
    rm, rv = torch.randn(c, generator=g), torch.rand(c, generator=g) + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    y = (F.batch_norm(x, rm_ref, rv_ref, gamma, beta, training=True, momentum=0.1, eps=1e-5) + res).relu()
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    rows = n * h * w
    xd, resd = to_nhwc(x.detach(), code), to_nhwc(res.detach(), code)
    wsb = L.lib.vs_bn_workspace(rows, c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    mean, invstd = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    rmd, rvd = rm.to(DEV), rv.to(DEV)
    L.check(L.lib.vs_bn_stats(code, L.ptr(xd), rows, c, 1e-5, 0.1, L.ptr(mean), L.ptr(invstd), L.ptr(rmd), L.ptr(rvd), L.ptr(ws), wsb, None))
    yd = torch.empty_like(xd)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    L.check(L.lib.vs_bn_apply(code, L.ptr(xd), L.ptr(mean), L.ptr(invstd), L.ptr(gd), L.ptr(bd), L.ptr(resd), 1, L.ptr(yd), rows, c, None))
    sync()
    xr = x.detach()
    assert torch.allclose(mean.cpu(), xr.mean((0, 2, 3)), atol=1e-5, rtol=1e-5)
    assert torch.allclose(invstd.cpu(), 1 / torch.sqrt(xr.var((0, 2, 3), unbiased=False) + 1e-5), rtol=1e-4)
    assert torch.allclose(rmd.cpu(), rm_ref, atol=1e-5, rtol=1e-5) and torch.allclose(rvd.cpu(), rv_ref, atol=1e-5, rtol=1e-4)
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    dx, dres = torch.empty_like(xd), torch.empty_like(xd)
    dgamma, dbeta = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    yin = to_nhwc(y.detach(), code)
    dyd = to_nhwc(dy, code)
    L.check(L.lib.vs_bn_bwd(code, L.ptr(dyd), L.ptr(yin), L.ptr(xd), L.ptr(mean), L.ptr(invstd), L.ptr(gd), 1,
                            L.ptr(dx), L.ptr(dres), L.ptr(dgamma), L.ptr(dbeta), rows, c, L.ptr(ws), wsb, None))
    sync()
    t = tol(code, x.grad.abs().max().item())
    assert torch.allclose(from_nhwc(dres), res.grad, **tol(code, res.grad.abs().max().item()))
    assert torch.allclose(dbeta.cpu(), beta.grad, rtol=1e-3, atol=1e-3 * beta.grad.abs().max().item())
    assert torch.allclose(dgamma.cpu(), gamma.grad, rtol=2e-3, atol=2e-3 * gamma.grad.abs().max().item() + (0 if code == 0 else 0.05))
    assert torch.allclose(from_nhwc(dx), x.grad, **t)


def test_bn_fold_matches_eval_batchnorm():
    L = lib()
    c = 96
    gamma, beta, rm, rv = torch.rand(c) + 0.5, torch.randn(c), torch.randn(c), torch.rand(c) + 0.1
    sc, sh = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    gd, bd, rmd, rvd = gamma.to(DEV), beta.to(DEV), rm.to(DEV), rv.to(DEV)
    L.check(L.lib.vs_bn_fold(L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), 1e-5, L.ptr(sc), L.ptr(sh), c, None))
    sync()
    x = torch.randn(2, c, 4, 4)
    ref = F.batch_norm(x, rm, rv, gamma, beta, training=False, eps=1e-5)
    assert torch.allclose(x * sc.cpu().view(1, -1, 1, 1) + sh.cpu().view(1, -1, 1, 1), ref, atol=1e-5, rtol=1e-5)


@pytest.mark.parametrize("code", CODES)
def test_maxpool_fwd_bwd_and_helpers(code):
    L = lib()
    g = torch.Generator().manual_seed(11)
    n, c, h, w = 2, 64, 16, 24
    x = rounded(torch.randn(n, c, h, w, generator=g), code).relu().requires_grad_()  # ReLU output: many exact ties at 0
    y = F.max_pool2d(x, 3, 2, 1)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    xd = to_nhwc(x.detach(), code)
    yd = torch.empty((n, h // 2, w // 2, c), device=DEV, dtype=tdtype(code))
    idx = torch.empty((n, h // 2, w // 2, c), device=DEV, dtype=torch.uint8)
    L.check(L.lib.vs_maxpool_fwd(code, L.ptr(xd), L.ptr(yd), L.ptr(idx), n, h, w, c, None))
    base = rounded(torch.randn(n, c, h, w, generator=g), code)
    dx, dyd = to_nhwc(base, code), to_nhwc(dy, code)
    L.check(L.lib.vs_maxpool_bwd(code, L.ptr(dyd), L.ptr(idx), L.ptr(dx), 1, n, h, w, c, None))
    dx0 = torch.empty_like(dx)
    L.check(L.lib.vs_maxpool_bwd(code, L.ptr(dyd), L.ptr(idx), L.ptr(dx0), 0, n, h, w, c, None))
    sync()
    assert torch.equal(from_nhwc(yd), y.detach())
    # gradient routed to the first maximum of each window, exactly like torch
    assert torch.allclose(from_nhwc(dx0), x.grad, **tol(code, 4.0))
    assert torch.allclose(from_nhwc(dx), rounded(x.grad + base, code), **tol(code, 4.0))


def test_adamw_matches_torch():
    L = lib()
    g = torch.Generator().manual_seed(12)
    n = 10007
    p0 = torch.randn(n, generator=g)
    p_ref = p0.clone().requires_grad_()
    opt = torch.optim.AdamW([p_ref], lr=1e-3)
    p, m, v = p0.to(DEV), torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
    mask = (torch.rand(n, generator=g) > 0.3).to(torch.uint8)
    for step in range(1, 4):
        grad = torch.randn(n, generator=g)
        lr, b1 = 1e-3 * step, 0.95 - 0.03 * step
        opt.param_groups[0]["lr"], opt.param_groups[0]["betas"] = lr, (b1, 0.999)
        p_ref.grad = grad.clone()
        opt.step()
        gd = grad.to(DEV)
        L.check(L.lib.vs_adamw_step(L.ptr(p), L.ptr(gd), L.ptr(m), L.ptr(v), None, n, lr, b1, 0.999, 1e-8, 0.01, step, None))
        sync()
        assert torch.allclose(p.cpu(), p_ref.detach(), rtol=1e-5, atol=1e-6)
    # masked elements stay frozen
    before = p.clone()
    ones, maskd = torch.ones(n, device=DEV), mask.to(DEV)
    L.check(L.lib.vs_adamw_step(L.ptr(p), L.ptr(ones), L.ptr(m), L.ptr(v), L.ptr(maskd), n, 1e-2, 0.9, 0.999, 1e-8, 0.01, 4, None))
    sync()
    frozen = mask == 0
    assert torch.equal(p.cpu()[frozen], before.cpu()[frozen]) and not torch.equal(p.cpu()[~frozen], before.cpu()[~frozen])


@pytest.mark.parametrize("classes,tdtype", [(2, torch.uint8), (4, torch.float32), (1, torch.uint8)])
def test_fused_dice_loss_matches_reference_formula(classes, tdtype):
    """vs_dice_loss_fwd/_bwd against the torch restatement of DiceLoss(normalization='none') (pinned to the reference by G6)."""
    from oracle import predictor_numpy as P
    from volume_segmantics_amd.data.losses import HipDiceLoss
    g = torch.Generator().manual_seed(20)
    x = torch.randn(3, classes, 40, 56, generator=g, requires_grad=True)
    lab = torch.randint(0, max(classes, 2), (3, 40, 56), generator=g)
    t = torch.nn.functional.one_hot(lab, max(classes, 2)).permute(0, 3, 1, 2)[:, :classes].contiguous()
    ref = P.dice_loss_none(x, t.float())
    (ref * 1.7).backward()
    xd = x.detach().to(DEV).requires_grad_()
    loss = HipDiceLoss()(xd, t.to(DEV, tdtype))
    (loss * 1.7).backward()
    sync()
    assert abs(loss.item() - ref.item()) < 1e-6
    assert torch.allclose(xd.grad.cpu(), x.grad, rtol=1e-4, atol=1e-9)


@pytest.mark.parametrize("classes", [1, 2, 4])
@pytest.mark.parametrize("tdtype", [torch.uint8, torch.float32])
def test_fused_segmentation_losses_match_the_reference_classes(golden, classes, tdtype):
    """vs_seg_loss_fwd/_bwd (BCE-Dice, BCE, cross entropy, generalised Dice) against g10: value and gradient computed by the
    reference's own BCEDiceLoss / GeneralizedDiceLoss and the torch modules it instantiates (oracle/gen_goldens.py:gen_g10)."""
    from volume_segmantics_amd.data.losses import HipSegLoss
    g = golden("g10_losses.npz")
    x0, t = torch.tensor(g[f"k{classes}__logits"]), torch.tensor(g[f"k{classes}__targets"])
    for name in ("BCEDiceLoss", "BCELoss", "GeneralizedDiceLoss", "CrossEntropyLoss"):
        if f"k{classes}__{name}__loss" not in g.files:
            continue
        crit = HipSegLoss(name, 0.75, 0.25)
        xd = x0.to(DEV).requires_grad_()
        loss = crit(xd, t.to(DEV, tdtype))
        (loss * 1.3).backward()
        sync()
        ref_l, ref_g = float(g[f"k{classes}__{name}__loss"]), torch.tensor(g[f"k{classes}__{name}__grad"])
        assert abs(loss.item() - ref_l) < 2e-6 * max(1.0, abs(ref_l)), (name, loss.item(), ref_l)
        err = (xd.grad.cpu() - ref_g).abs().max().item()
        assert err < 2e-5 * ref_g.abs().max().item() + 1e-10, (name, err, ref_g.abs().max().item())
        # the CPU route of the same module (torch restatement) agrees too
        xc = x0.clone().requires_grad_()
        lc = crit(xc, t if tdtype == torch.uint8 else t.float())
        assert abs(lc.item() - ref_l) < 2e-6 * max(1.0, abs(ref_l)), name


@pytest.mark.parametrize("classes", [1, 2, 4, 7])
def test_mean_iou_matches_oracle(classes):
    """vs_mean_iou vs the oracle's restatement of MeanIoU.__call__ (pinned by golden g6 in the CPU suite)."""
    from oracle import predictor_numpy as P
    from volume_segmantics_amd.data.losses import MeanIoU
    g = torch.Generator().manual_seed(31 + classes)
    n, h, w = 3, 40, 56
    logits = torch.randn(n, classes, h, w, generator=g)
    logits[0, :, :4] = 0.25                      # ties: the first maximum must win
    probs = torch.softmax(logits, 1) if classes > 1 else torch.sigmoid(logits)
    lab = torch.randint(0, max(classes, 2), (n, h, w), generator=g)
    onehot = torch.nn.functional.one_hot(lab, max(classes, 2)).permute(0, 3, 1, 2)[:, :classes].contiguous()
    for tt in (onehot.to(torch.uint8), onehot.float()):
        ref = P.mean_iou(probs, tt)
        got = MeanIoU()(probs.to(DEV), tt.to(DEV))
        assert got.is_cuda and abs(got.item() - ref.item()) < 1e-6, (got.item(), ref.item())
        got5 = MeanIoU()(probs.to(DEV).unsqueeze(2), tt.to(DEV).unsqueeze(2))     # the trainer's 5-D form
        assert abs(got5.item() - ref.item()) < 1e-6
        if classes > 1:
            gl = MeanIoU().from_logits(logits.to(DEV), tt.to(DEV))                # softmax formed inside the kernel
            assert abs(gl.item() - ref.item()) < 1e-6


def test_onehot_matches_reference_prepare_training_batch():
    from volume_segmantics_amd.utilities.base_data_utils import prepare_training_batch
    g = torch.Generator().manual_seed(41)
    for k, shape in ((2, (3, 64, 64)), (4, (2, 40, 56)), (7, (1, 33, 31))):
        img = torch.randn(shape[0], 1, *shape[1:], generator=g)
        lab = torch.randint(0, k, shape, generator=g, dtype=torch.uint8)
        ref = torch.nn.functional.one_hot(lab.long(), k).permute(0, 3, 1, 2).to(torch.uint8)
        x, t = prepare_training_batch((img, lab), DEV, k)
        assert t.is_cuda and t.dtype == torch.uint8 and torch.equal(t.cpu(), ref) and torch.equal(x.cpu(), img)


# ---- smp.FPN's streaming operators (csrc/fpn.hip) ---------------------------------------------------------------------------
@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 128, 32), (3, 8, 12, 64, 32), (1, 32, 32, 256, 32), (2, 5, 7, 32, 8)])
def test_group_norm_relu_fwd_bwd(code, shape):
    """nn.GroupNorm(groups, c) + ReLU (smp Conv3x3GNReLU) on NHWC tensors against torch CPU: output, input gradient (the ReLU
    mask recomputed from x), dgamma, dbeta."""
    L = lib()
    n, h, w, c, G = shape
    g = torch.Generator().manual_seed(3)
    x = rounded(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3, code).requires_grad_()
    gamma = (torch.rand(c, generator=g) + 0.5).requires_grad_()
    beta = (torch.randn(c, generator=g) * 0.3).requires_grad_()
    y = F.relu(F.group_norm(x, G, gamma, beta, eps=1e-5))
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    xd = to_nhwc(x.detach(), code)
    yd = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    stats = torch.full((n, G, 2), float("nan"), device=DEV)
    wsb = L.lib.vs_gn_bwd_workspace(n, c, G)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    gd, bd = gamma.detach().to(DEV), beta.detach().to(DEV)
    L.check(L.lib.vs_gn_fwd(code, L.ptr(xd), L.ptr(gd), L.ptr(bd), 1, L.ptr(yd), L.ptr(stats), n, h * w, c, G, 1e-5, L.ptr(ws), wsb, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    xg = x.detach().view(n, G, -1)
    assert torch.allclose(stats[..., 0].cpu(), xg.mean(-1), atol=1e-5)
    assert torch.allclose(stats[..., 1].cpu(), (xg.var(-1, unbiased=False) + 1e-5).rsqrt(), rtol=1e-4)
    dyd = to_nhwc(dy, code)
    dx = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    dg, db = torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
    L.check(L.lib.vs_gn_bwd(code, L.ptr(dyd), L.ptr(xd), L.ptr(stats), L.ptr(gd), L.ptr(bd), 1, L.ptr(dx), L.ptr(dg), L.ptr(db), n, h * w,
                            c, G, L.ptr(ws), wsb, None))
    sync()
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    assert torch.allclose(dg.cpu(), gamma.grad, rtol=1e-3, atol=1e-3 * gamma.grad.abs().max().item())
    assert torch.allclose(db.cpu(), beta.grad, rtol=1e-3, atol=1e-3 * beta.grad.abs().max().item())


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 8, 8, 16, 2), (1, 5, 9, 8, 2), (2, 4, 6, 32, 4), (1, 1, 3, 8, 2)])
def test_bilinear_upsampling_align_corners_fwd_bwd(code, shape):
    """F.interpolate(scale_factor=f, mode="bilinear", align_corners=True) (Conv3x3GNReLU's upsampling; f = 4:
    nn.UpsamplingBilinear2d of the head) and its adjoint, NHWC tensors and fp32 NCHW planes, against torch CPU."""
    L = lib()
    n, h, w, c, f = shape
    g = torch.Generator().manual_seed(9)
    x = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    y = F.interpolate(x, scale_factor=f, mode="bilinear", align_corners=True)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    xd, dyd = to_nhwc(x.detach(), code), to_nhwc(dy, code)
    yd = torch.full((n, h * f, w * f, c), float("nan"), device=DEV, dtype=tdtype(code))
    dx = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_bilinear_up(code, L.ptr(xd), L.ptr(yd), n, h, w, c, f, None))
    L.check(L.lib.vs_bilinear_up_bwd(code, L.ptr(dyd), L.ptr(dx), n, h, w, c, f, 0, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, 1.0))
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    acc = dx.clone()
    L.check(L.lib.vs_bilinear_up_bwd(code, L.ptr(dyd), L.ptr(acc), n, h, w, c, f, 1, None))     # accumulate: twice the gradient
    sync()
    assert torch.allclose(acc.float(), 2 * dx.float(), **tol(code, x.grad.abs().max().item()))
    if code == 0:
        xp, dyp = x.detach().contiguous().to(DEV), dy.contiguous().to(DEV)
        yp = torch.full((n, c, h * f, w * f), float("nan"), device=DEV)
        dxp = torch.full((n, c, h, w), float("nan"), device=DEV)
        L.check(L.lib.vs_bilinear_up_planes(L.ptr(xp), L.ptr(yp), n * c, h, w, f, None))
        L.check(L.lib.vs_bilinear_up_planes_bwd(L.ptr(dyp), L.ptr(dxp), n * c, h, w, f, None))
        sync()
        assert torch.allclose(yp.cpu(), y.detach(), rtol=1e-5, atol=1e-5)
        assert torch.allclose(dxp.cpu(), x.grad, rtol=1e-4, atol=1e-5 * x.grad.abs().max().item())


@pytest.mark.parametrize("code", CODES)
def test_fpn_upsample_add_and_dropout2d(code):
    """FPNBlock's F.interpolate(x, 2, "nearest") + skip, and nn.Dropout2d: whole (sample, channel) planes zeroed with
    probability p, the rest scaled by 1 / (1 - p); a new mask per counter value, the same mask for the same (seed, counter)."""
    L = lib()
    n, h, w, c = 3, 6, 10, 32
    g = torch.Generator().manual_seed(1)
    x, s = rounded(torch.randn(n, c, h, w, generator=g), code), rounded(torch.randn(n, c, 2 * h, 2 * w, generator=g), code)
    yd = torch.full((n, 2 * h, 2 * w, c), float("nan"), device=DEV, dtype=tdtype(code))
    xd, sd = to_nhwc(x, code), to_nhwc(s, code)
    L.check(L.lib.vs_upsample2x_add(code, L.ptr(xd), L.ptr(sd), L.ptr(yd), n, h, w, c, None))
    sync()
    assert torch.allclose(from_nhwc(yd), F.interpolate(x, scale_factor=2, mode="nearest") + s, **tol(code, 4.0))
    nn_, cc, p = 64, 128, 0.2
    counter = torch.tensor([5], dtype=torch.int64, device=DEV)
    masks = []
    for bias in (0, 0, 1):
        m = torch.full((nn_, cc), float("nan"), device=DEV)
        L.check(L.lib.vs_dropout2d_mask(L.ptr(m), nn_, cc, p, 1234, L.ptr(counter), bias, None))
        masks.append(m)
    sync()
    assert torch.equal(masks[0], masks[1]) and not torch.equal(masks[0], masks[2])
    vals = set(masks[0].unique().tolist())
    assert vals == {0.0, 1.25}
    frac = (masks[0] == 0).float().mean().item()
    assert abs(frac - p) < 0.02, frac                                    # 8192 draws: sigma = 0.0044
    xs = rounded(torch.randn(nn_, cc, 3, 5, generator=g), code)
    out = torch.full((nn_, 3, 5, cc), float("nan"), device=DEV, dtype=tdtype(code))
    xsd = to_nhwc(xs, code)
    L.check(L.lib.vs_channel_scale(code, L.ptr(xsd), L.ptr(masks[0]), L.ptr(out), nn_, 15, cc, None))
    sync()
    assert torch.allclose(from_nhwc(out), xs * masks[0].cpu()[:, :, None, None], **tol(code, 4.0))


def test_rccl_communicator_through_the_c_abi_single_rank():
    """vs_comm_* (csrc/comm.hip): RCCL opened lazily, a one-rank communicator on this GPU; with one rank every collective is
    the identity - which pins the plumbing (unique id, init on the current device, datatypes, in-place calls on the caller's
    stream, destroy).  N > 1 needs as many GPUs: the driver's multi-GPU run is the first time more than one rank exists."""
    from volume_segmantics_amd.dist import VsComm
    comm = VsComm(torch.device(DEV))
    assert (comm.rank, comm.size) == (0, 1)
    L = lib()
    assert L.lib.vs_comm_size(comm.handle) == 1 and L.lib.vs_comm_rank(comm.handle) == 0
    g = torch.randn(1 << 20, device=DEV)
    ref = g.clone()
    comm.allreduce_sum_(g)
    keys = torch.randint(0, 2 ** 31 - 1, (1 << 20,), device=DEV, dtype=torch.int32)
    kref = keys.clone()
    comm.allreduce_max_keys_(keys)
    out = torch.zeros_like(keys)
    comm.reduce_scatter_max_keys(keys, out)
    gathered = torch.zeros(4096, dtype=torch.uint8, device=DEV)
    part = torch.arange(4096, device=DEV).to(torch.uint8)
    comm.allgather(part, gathered)
    comm.broadcast_(g)
    sync()
    assert torch.equal(g, ref) and torch.equal(keys, kref) and torch.equal(out, kref) and torch.equal(gathered, part)
    comm.close()


# ---- the non-dense operators of smp's DeepLabV3+ decoder (csrc/dwconv.hip) ---------------------------------------------------
@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 1), (1, 16, 24, 304, 1), (2, 16, 16, 512, 12), (2, 8, 8, 256, 24), (1, 8, 8, 32, 36)])
def test_depthwise_dilated_conv_fwd_dgrad_wgrad(code, shape):
    """nn.Conv2d(c, c, 3, padding=d, dilation=d, groups=c, bias=False) - the depthwise half of smp's SeparableConv2d (ASPP rates 12 /
    24 / 36, and rate 1 in the decoder; 304 = 256 + 48 channels after the concat) - against torch CPU."""
    L = lib()
    n, h, w, c, d = shape
    g = torch.Generator().manual_seed(17)
    x = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    wt = (torch.randn(c, 1, 3, 3, generator=g) / 3).requires_grad_()
    y = F.conv2d(x, wt, padding=d, dilation=d, groups=c)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    xd, dyd, wd = to_nhwc(x.detach(), code), to_nhwc(dy, code), wt.detach().reshape(c, 9).contiguous().to(DEV)
    yd = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    dx = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_dwconv3x3(code, L.ptr(xd), L.ptr(wd), L.ptr(yd), n, h, w, c, d, 0, None))
    L.check(L.lib.vs_dwconv3x3(code, L.ptr(dyd), L.ptr(wd), L.ptr(dx), n, h, w, c, d, 1, None))
    wsb = L.lib.vs_dwconv3x3_wgrad_workspace(c)
    ws = torch.empty(wsb, dtype=torch.uint8, device=DEV)
    dw = torch.full((c, 9), float("nan"), device=DEV)
    L.check(L.lib.vs_dwconv3x3_wgrad(code, L.ptr(xd), L.ptr(dyd), L.ptr(dw), n, h, w, c, d, L.ptr(ws), wsb, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    ref = wt.grad.reshape(c, 9)
    assert torch.allclose(dw.cpu(), ref, rtol=1e-3, atol=1e-3 * ref.abs().max().item())


@pytest.mark.parametrize("code", CODES)
def test_spatial_sum_broadcast_and_elementwise_dropout(code):
    """nn.AdaptiveAvgPool2d(1) and the broadcast back over the map (ASPPPooling; each other's gradients), and element-wise
    nn.Dropout(0.5) whose mask is a pure function of (seed, counter, element): the same call on the gradient is the backward."""
    L = lib()
    n, h, w, c = 3, 5, 7, 512
    g = torch.Generator().manual_seed(2)
    x = rounded(torch.randn(n, c, h, w, generator=g), code)
    xd = to_nhwc(x, code)
    pooled = torch.full((n, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_spatial_sum(code, L.ptr(xd), L.ptr(pooled), n, h * w, c, 1.0 / (h * w), None))
    back = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_broadcast_rows(code, L.ptr(pooled), L.ptr(back), n, h * w, c, 1.0, 0, None))
    acc = xd.clone()
    L.check(L.lib.vs_broadcast_rows(code, L.ptr(pooled), L.ptr(acc), n, h * w, c, 2.0, 1, None))
    sync()
    ref = x.mean(dim=(2, 3))
    assert torch.allclose(pooled.float().cpu(), ref, **tol(code, 1.0))
    assert torch.allclose(from_nhwc(back), pooled.float().cpu()[:, :, None, None].expand(n, c, h, w), **tol(code, 1.0))
    assert torch.allclose(from_nhwc(acc), x + 2 * pooled.float().cpu()[:, :, None, None], **tol(code, 4.0))
    counter = torch.tensor([3], dtype=torch.int64, device=DEV)
    big = torch.ones(1 << 20, device=DEV, dtype=tdtype(code))
    outs = []
    for bias in (0, 0, 1):
        o = torch.empty_like(big)
        L.check(L.lib.vs_dropout(code, L.ptr(big), L.ptr(o), big.numel(), 0.5, 77, L.ptr(counter), bias, None))
        outs.append(o)
    sync()
    assert torch.equal(outs[0], outs[1]) and not torch.equal(outs[0], outs[2])
    assert set(outs[0].float().unique().tolist()) == {0.0, 2.0}
    assert abs((outs[0] == 0).float().mean().item() - 0.5) < 0.005


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 16, 16, 64, 32, 12), (1, 32, 32, 32, 64, 24), (2, 8, 12, 64, 32, 36), (1, 20, 20, 32, 32, 3)])
def test_dense_dilated_conv_through_column_form(code, shape):
    """nn.Conv2d(cin, cout, 3, padding=r, dilation=r) at a large rate r (DeepLabV3's ASPPConv, 12 / 24 / 36): vs_dilated_im2col and
    the 1x1 convolution over 9 cin channels with the SAME weight memory ([cout][tap][cin]) - against torch CPU; the column form
    itself is exact, and its adjoint (inverse = 1, with and without accumulation) reproduces autograd's input gradient."""
    L = lib()
    n, h, w, cin, cout, r = shape
    g = torch.Generator().manual_seed(21)
    x = rounded(torch.randn(n, cin, h, w, generator=g), code).requires_grad_()
    wt = rounded(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, code)
    ref = F.conv2d(x, wt, padding=r, dilation=r)
    dy = rounded(torch.randn(ref.shape, generator=g), code)
    ref.backward(dy)
    xd = to_nhwc(x.detach(), code)
    col = torch.full((n, h, w, 9, cin), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_dilated_im2col(code, L.ptr(xd), L.ptr(col), n, h, w, cin, r, 0, 0, None))
    d = conv_desc(L, code, n, h, w, 9 * cin, cout, 1, 1, 0)
    y = torch.full((n, h, w, cout), float("nan"), device=DEV, dtype=tdtype(code))
    wd = w_krsc(wt, code)
    L.check(L.lib.vs_conv2d_fwd(d, L.ptr(col), None, L.ptr(wd), None, None, None, L.ptr(y), None, None))
    sync()
    xp = F.pad(x.detach(), (r, r, r, r))
    want = torch.stack([xp[:, :, kh * r:kh * r + h, kw * r:kw * r + w] for kh in range(3) for kw in range(3)], 1)   # [n][9][c][h][w]
    assert torch.equal(col.float().cpu(), want.permute(0, 3, 4, 1, 2))
    assert torch.allclose(from_nhwc(y), ref.detach(), **tol(code, ref.abs().max().item()))
    # the adjoint: the column form of the input gradient (dy through the [cout][9 cin] matrix, on the host) scattered back
    dcol = rounded(torch.einsum("nohw,ockl->nhwklc", dy, wt).reshape(n, h, w, 9, cin), code)
    want_dx = torch.zeros(n, cin, h, w)
    dpad = torch.zeros(n, cin, h + 2 * r, w + 2 * r)
    for kh in range(3):
        for kw in range(3):
            dpad[:, :, kh * r:kh * r + h, kw * r:kw * r + w] += dcol[:, :, :, kh * 3 + kw, :].permute(0, 3, 1, 2)
    want_dx = dpad[:, :, r:r + h, r:r + w]
    dcd = dcol.to(DEV, tdtype(code)).contiguous()
    dx = torch.full((n, h, w, cin), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_dilated_im2col(code, L.ptr(dcd), L.ptr(dx), n, h, w, cin, r, 1, 0, None))
    twice = dx.clone()
    L.check(L.lib.vs_dilated_im2col(code, L.ptr(dcd), L.ptr(twice), n, h, w, cin, r, 1, 1, None))
    sync()
    t = tol(code, want_dx.abs().max().item())
    assert torch.allclose(from_nhwc(dx), want_dx, **t)
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item() * 2))
    assert torch.allclose(from_nhwc(twice), 2 * want_dx, **tol(code, 2 * want_dx.abs().max().item()))


# ---- smp.MAnet's attention operators (csrc/manet.hip) --------------------------------------------------------------------------
@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 4, 4, 64, 512), (1, 8, 8, 64, 64), (3, 2, 3, 16, 32)])
def test_pab_attention_fwd_bwd(code, shape):
    """smp's PAB (after its three convolutions): sp = softmax over ALL hw x hw entries of center^T top, out = sp bottom, then
    `reshape(b, C, h, w)` of the [b][hw][C] product WITHOUT a transpose, added to x - restated in torch on NCHW tensors exactly
    as decoders/manet/decoder.py writes it, forward and (through autograd) backward."""
    L = lib()
    n, h, w, K, C = shape
    hw = h * w
    g = torch.Generator().manual_seed(31)
    top = rounded(torch.randn(n, K, h, w, generator=g) * 0.5, code).requires_grad_()
    center = rounded(torch.randn(n, K, h, w, generator=g) * 0.5, code).requires_grad_()
    bottom = rounded(torch.randn(n, C, h, w, generator=g), code).requires_grad_()
    x = rounded(torch.randn(n, C, h, w, generator=g), code)
    xt, xc, xb = top.flatten(2), center.flatten(2).transpose(1, 2), bottom.flatten(2).transpose(1, 2)
    sp = torch.softmax(torch.matmul(xc, xt).view(n, -1), dim=1).view(n, hw, hw)
    y = x + torch.matmul(sp, xb).reshape(n, C, h, w)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    td, cd, bd, xd, dyd = (to_nhwc(t_.detach(), code) for t_ in (top, center, bottom, x, dy))
    yd = torch.full((n, h, w, C), float("nan"), device=DEV, dtype=tdtype(code))
    spd = torch.full((n, hw, hw), float("nan"), device=DEV)
    sb = L.lib.vs_pab_scratch_bytes(n, hw, C)
    scr = torch.empty(sb, dtype=torch.uint8, device=DEV)
    L.check(L.lib.vs_pab_attention_fwd(code, L.ptr(td), L.ptr(cd), L.ptr(bd), L.ptr(xd), L.ptr(yd), L.ptr(spd), L.ptr(scr), n, hw, K, C, None))
    dt_, dc_, db_ = (torch.full(t_.shape, float("nan"), device=DEV, dtype=tdtype(code)) for t_ in (td, cd, bd))
    L.check(L.lib.vs_pab_attention_bwd(code, L.ptr(dyd), L.ptr(td), L.ptr(cd), L.ptr(bd), L.ptr(spd), L.ptr(dt_), L.ptr(dc_), L.ptr(db_), L.ptr(scr),
                                       n, hw, K, C, None))
    sync()
    assert torch.allclose(spd.cpu(), sp.detach(), rtol=1e-3, atol=1e-6)
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    for got, ref in ((dt_, top.grad), (dc_, center.grad), (db_, bottom.grad)):
        assert torch.allclose(from_nhwc(got), ref, **tol(code, ref.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("swish", [0, 1])
@pytest.mark.parametrize("shape", [(4, 256, 16), (3, 64, 4), (2, 128, 8), (2, 2688, 112)])
def test_se_gate_and_channel_gate_fwd_bwd(code, shape, swish):
    """MFAB's squeeze-excitation: AdaptiveAvgPool2d(1) -> Conv1x1(C, C/16) -> ReLU -> Conv1x1(C/16, C) -> Sigmoid, the gate multiplied
    onto the feature map - the pooled part (vs_spatial_sum is tested above) through vs_se_gate_*, the product and its two
    gradients through vs_channel_gate / vs_channel_dot, against autograd."""
    L = lib()
    n, C, R = shape
    h, w = 5, 6
    g = torch.Generator().manual_seed(41)
    x = rounded(torch.randn(n, C, h, w, generator=g), code).requires_grad_()
    p = rounded(torch.randn(n, C, generator=g), code).requires_grad_()
    w1 = (torch.randn(R, C, generator=g) / C ** 0.5).requires_grad_()
    b1 = (torch.randn(R, generator=g) * 0.1).requires_grad_()
    w2 = (torch.randn(C, R, generator=g) / R ** 0.5).requires_grad_()
    b2 = (torch.randn(C, generator=g) * 0.1).requires_grad_()
    hidden = F.linear(p, w1, b1)      # swish: efficientnet-pytorch's MBConvBlock (_se_reduce, swish, _se_expand, sigmoid)
    a = torch.sigmoid(F.linear(hidden * torch.sigmoid(hidden) if swish else F.relu(hidden), w2, b2))
    y = x * a[:, :, None, None]
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    dev = lambda t_: t_.detach().contiguous().to(DEV)
    pd = dev(p).to(tdtype(code))
    ad = torch.full((n, C), float("nan"), device=DEV, dtype=tdtype(code))
    hid = torch.full((n, R), float("nan"), device=DEV)
    w1d, b1d, w2d, b2d = dev(w1), dev(b1), dev(w2), dev(b2)
    L.check(L.lib.vs_se_gate_fwd(code, L.ptr(pd), L.ptr(w1d), L.ptr(b1d), L.ptr(w2d), L.ptr(b2d), L.ptr(ad), L.ptr(hid), n, C, R, swish, None))
    xd, dyd = to_nhwc(x.detach(), code), to_nhwc(dy, code)
    yd = torch.full((n, h, w, C), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_channel_gate(code, L.ptr(xd), L.ptr(ad), L.ptr(yd), n, h * w, C, None))
    dx = torch.full_like(yd, float("nan"))
    L.check(L.lib.vs_channel_gate(code, L.ptr(dyd), L.ptr(ad), L.ptr(dx), n, h * w, C, None))
    da = torch.full((n, C), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_channel_dot(code, L.ptr(xd), L.ptr(dyd), L.ptr(da), n, h * w, C, None))
    dp = torch.full((n, C), float("nan"), device=DEV, dtype=tdtype(code))
    dw1, db1 = torch.full((R, C), float("nan"), device=DEV), torch.full((R,), float("nan"), device=DEV)
    dw2, db2 = torch.full((C, R), float("nan"), device=DEV), torch.full((C,), float("nan"), device=DEV)
    scr = torch.empty(L.lib.vs_se_gate_scratch_floats(n, C, R), device=DEV)
    L.check(L.lib.vs_se_gate_bwd(code, L.ptr(da), L.ptr(ad), L.ptr(pd), L.ptr(hid), L.ptr(w1d), L.ptr(w2d), L.ptr(dp), L.ptr(dw1), L.ptr(db1),
                                 L.ptr(dw2), L.ptr(db2), L.ptr(scr), n, C, R, swish, None))
    sync()
    t = tol(code, 1.0)
    assert torch.allclose(ad.float().cpu(), a.detach(), **t)
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    loose = dict(rtol=5e-2, atol=5e-2 * p.grad.abs().max().item()) if code else dict(rtol=1e-3, atol=1e-4 * p.grad.abs().max().item())
    assert torch.allclose(dp.float().cpu(), p.grad, **loose)
    for got, ref in ((dw1, w1.grad), (db1, b1.grad), (dw2, w2.grad), (db2, b2.grad)):
        lim = (5e-2 if code else 1e-3) * ref.abs().max().item()
        assert torch.allclose(got.cpu(), ref, rtol=5e-2 if code else 1e-3, atol=lim), (got.cpu() - ref).abs().max()


# ---- smp.PAN's Feature Pyramid Attention (csrc/pan.hip) -------------------------------------------------------------------------
@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(4, 16, 16, 64), (2, 8, 16, 32)])
def test_fpa_pyramid_fwd_bwd(code, shape):
    """The single-channel pyramid of smp's FPABlock, restated in torch as decoders/pan/decoder.py writes it (MaxPool2d(2, 2), ConvBnRelu
    with 7x7 / 5x5 / 3x3 kernels - biased convolution, train-mode BatchNorm, ReLU -, bilinear align_corners upsamplings, the two
    additions), from the bottleneck feature to the attention plane and the final x * mid + b1: vs_maxpool2x2, vs_conv_to_plane,
    vs_fpa_pyramid_fwd / _bwd (one workgroup), vs_fpa_combine - outputs, running statistics, every parameter gradient and the
    input gradient against autograd."""
    import ctypes
    L = lib()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(51)
    torch.manual_seed(51)

    def cbr(i, o, k):
        m = torch.nn.Sequential(torch.nn.Conv2d(i, o, k, padding=k // 2), torch.nn.BatchNorm2d(o), torch.nn.ReLU())
        with torch.no_grad():
            m[1].weight.copy_(torch.rand(o) + 0.5); m[1].bias.copy_(torch.randn(o) * 0.2)
            if i == 1:
                m[0].weight.mul_(3.0)
        return m
    down1, down2, d31, d32, conv2, conv1 = cbr(c, 1, 7), cbr(1, 1, 5), cbr(1, 1, 3), cbr(1, 1, 3), cbr(1, 1, 5), cbr(1, 1, 7)
    layers = [down1, down2, d31, d32, conv2, conv1]
    x = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    mid = rounded(torch.randn(n, 32, h, w, generator=g), code).requires_grad_()
    b1 = rounded(torch.randn(n, 32, generator=g), code).requires_grad_()
    up = lambda t_, s_: F.interpolate(t_, size=s_, mode="bilinear", align_corners=True)
    pool = torch.nn.MaxPool2d(2, 2)
    x1 = down1(pool(x)); x2 = down2(pool(x1)); x3 = d32(d31(pool(x2)))
    s = conv2(x2) + up(x3, (h // 4, w // 4))
    s = up(s, (h // 2, w // 2)) + conv1(x1)
    plane = up(s, (h, w))
    out = plane * mid + b1[:, :, None, None]
    dy = rounded(torch.randn(out.shape, generator=g), code)
    out.backward(dy)
    # ---- device ----
    xd, midd, dyd = to_nhwc(x.detach(), code), to_nhwc(mid.detach(), code), to_nhwc(dy, code)
    b1d = b1.detach().to(DEV, tdtype(code)).contiguous()
    pooled = torch.full((n, h // 2, w // 2, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_maxpool2x2(code, L.ptr(xd), L.ptr(pooled), n, h, w, c, None))
    arena = torch.zeros(L.lib.vs_fpa_arena_floats(n, h, w), device=DEV)
    w7 = down1[0].weight.detach().permute(0, 2, 3, 1).reshape(49, c).contiguous().to(DEV)
    b7 = down1[0].bias.detach().to(DEV)
    L.check(L.lib.vs_conv_to_plane(code, L.ptr(pooled), L.ptr(w7), L.ptr(b7), L.ptr(arena), n, h // 2, w // 2, c, 7, None))
    keep, ptrs = [], []
    for i, m in enumerate(layers):
        ts = [m[0].weight.detach().reshape(-1) if i else torch.zeros(1), m[0].bias.detach() if i else torch.zeros(1), m[1].weight.detach(),
              m[1].bias.detach(), torch.zeros(1), torch.ones(1)]
        for t_ in ts:
            d_ = t_.clone().contiguous().to(DEV); keep.append(d_); ptrs.append(L.ptr(d_))
    params = (ctypes.c_void_p * 36)(*ptrs)
    planed = torch.full((n, h, w), float("nan"), device=DEV)
    L.check(L.lib.vs_fpa_pyramid_fwd(L.ptr(arena), L.ptr(planed), params, n, h, w, 1, None))
    outd = torch.full((n, h, w, 32), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_fpa_combine(code, L.ptr(planed), L.ptr(midd), L.ptr(b1d), L.ptr(outd), n, h * w, 32, None))
    dmid = torch.full_like(outd, float("nan"))
    dplane = torch.full((n, h, w), float("nan"), device=DEV)
    L.check(L.lib.vs_fpa_combine_bwd(code, L.ptr(dyd), L.ptr(planed), L.ptr(midd), L.ptr(dmid), L.ptr(dplane), n, h * w, 32, None))
    gkeep, gptrs = [], []
    for i, m in enumerate(layers):
        for numel in (m[0].weight.numel() if i else 1, 1, 1, 1):
            d_ = torch.full((numel,), float("nan"), device=DEV); gkeep.append(d_); gptrs.append(L.ptr(d_))
    grads = (ctypes.c_void_p * 24)(*gptrs)
    L.check(L.lib.vs_fpa_pyramid_bwd(L.ptr(arena), L.ptr(dplane), params, grads, n, h, w, None))
    dz1 = arena[L.lib.vs_fpa_dz1_offset(n, h, w):][: n * (h // 2) * (w // 2)]
    dpool = torch.full_like(pooled, float("nan"))
    dw7, db7 = torch.full((49, c), float("nan"), device=DEV), torch.full((1,), float("nan"), device=DEV)
    L.check(L.lib.vs_conv_to_plane_bwd(code, L.ptr(pooled), L.ptr(w7), L.ptr(dz1), L.ptr(dpool), L.ptr(dw7), L.ptr(db7), n, h // 2, w // 2, c, 7, None))
    dx = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_maxpool2x2_bwd(code, L.ptr(xd), L.ptr(dpool), L.ptr(dx), n, h, w, c, 0, None))
    sync()
    tl = dict(rtol=2e-3, atol=2e-3) if code == 0 else dict(rtol=3e-2, atol=3e-2)
    assert torch.allclose(planed.cpu(), plane.detach()[:, 0], **tl), (planed.cpu() - plane.detach()[:, 0]).abs().max()
    assert torch.allclose(from_nhwc(outd), out.detach(), **tol(code, out.abs().max().item() * 4))
    assert torch.allclose(from_nhwc(dmid), mid.grad, **tol(code, mid.grad.abs().max().item() * 4))
    for i, m in enumerate(layers):        # running statistics and parameter gradients
        assert torch.allclose(keep[6 * i + 4].cpu(), m[1].running_mean, rtol=1e-3, atol=1e-4), i
        assert torch.allclose(keep[6 * i + 5].cpu(), m[1].running_var, rtol=1e-3, atol=1e-4), i
        refs = [m[0].weight.grad.reshape(-1) if i else None, m[0].bias.grad if i else None, m[1].weight.grad, m[1].bias.grad]
        for j, r in enumerate(refs):
            if r is None:
                continue
            lim = (3e-3 if code == 0 else 6e-2) * max(r.abs().max().item(), 1e-3)
            if j == 1:      # a bias in front of a train-mode BatchNorm: its gradient is zero in exact arithmetic, noise in fp32
                lim = 1e-3
            assert torch.allclose(gkeep[4 * i + j].cpu(), r, rtol=3e-3 if code == 0 else 6e-2, atol=lim), (i, j, gkeep[4 * i + j].cpu(), r)
    r7 = down1[0].weight.grad.permute(0, 2, 3, 1).reshape(49, c)
    lim = (3e-3 if code == 0 else 6e-2)
    assert torch.allclose(dw7.cpu(), r7, rtol=lim, atol=lim * r7.abs().max().item())
    assert torch.allclose(from_nhwc(dx), x.grad, rtol=lim, atol=lim * x.grad.abs().max().item())


# ---- smp's EfficientNet encoders (csrc/effnet.hip) -----------------------------------------------------------------------------------
@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 6, 5, 48), (3, 4, 4, 144), (2, 3, 3, 2688)])
def test_bn2_any_channel_count_swish(code, act, shape):
    """nn.BatchNorm2d(c, eps=1e-3, momentum=0.01) + nothing / ReLU / swish for channel counts the ResNet kernels do not take (48, 144,
    2688 = EfficientNet-b4's widest expansion): batch statistics, running statistics, output, evaluation from the running statistics,
    and the backward pass (dx, dgamma, dbeta) against autograd."""
    L = lib()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(51)
    x = rounded(torch.randn(n, c, h, w, generator=g) * 1.5 + 0.3, code).requires_grad_()
    gamma = (1 + 0.2 * torch.randn(c, generator=g)).requires_grad_()
    beta = (0.2 * torch.randn(c, generator=g)).requires_grad_()
    rm0, rv0 = torch.randn(c, generator=g) * 0.1, 1 + 0.1 * torch.rand(c, generator=g)
    rm, rv = rm0.clone(), rv0.clone()
    f = {0: lambda v: v, 1: F.relu, 2: lambda v: v * torch.sigmoid(v)}[act]
    y = f(F.batch_norm(x, rm, rv, gamma, beta, True, 0.01, 1e-3))
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    y_eval = f(F.batch_norm(x.detach(), rm0, rv0, gamma.detach(), beta.detach(), False, 0.01, 1e-3))
    rows = n * h * w
    xd, dyd = to_nhwc(x.detach(), code), to_nhwc(dy, code)
    dev = lambda t_: t_.detach().contiguous().to(DEV)
    gd, bd, rmd, rvd = dev(gamma), dev(beta), dev(rm0), dev(rv0)
    rme, rve = dev(rm0), dev(rv0)
    mean, invstd = torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
    wsb = L.lib.vs_bn2_workspace(c)
    ws = torch.empty(wsb // 4, device=DEV)
    L.check(L.lib.vs_bn2_stats(code, L.ptr(xd), rows, c, 1e-3, 0.01, L.ptr(mean), L.ptr(invstd), L.ptr(rmd), L.ptr(rvd), L.ptr(ws), wsb, None))
    yd = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_bn2_apply(code, L.ptr(xd), L.ptr(mean), L.ptr(invstd), L.ptr(gd), L.ptr(bd), act, -1.0, L.ptr(yd), rows, c, None))
    ye = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_bn2_apply(code, L.ptr(xd), L.ptr(rme), L.ptr(rve), L.ptr(gd), L.ptr(bd), act, 1e-3, L.ptr(ye), rows, c, None))
    dx = torch.full_like(xd, float("nan"))
    dg, db = torch.full((c,), float("nan"), device=DEV), torch.full((c,), float("nan"), device=DEV)
    L.check(L.lib.vs_bn2_bwd(code, L.ptr(dyd), L.ptr(xd), L.ptr(mean), L.ptr(invstd), L.ptr(gd), L.ptr(bd), act, L.ptr(dx), L.ptr(dg), L.ptr(db),
                             rows, c, L.ptr(ws), wsb, None))
    sync()
    assert torch.allclose(rmd.cpu(), rm, rtol=1e-5, atol=1e-6) and torch.allclose(rvd.cpu(), rv, rtol=1e-5, atol=1e-6)
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    assert torch.allclose(from_nhwc(ye), y_eval, **tol(code, y_eval.abs().max().item()))
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item() * (4 if code else 1)))
    lim = (3e-2 if code else 1e-4)
    assert torch.allclose(dg.cpu(), gamma.grad, rtol=lim, atol=lim * gamma.grad.abs().max().item())
    assert torch.allclose(db.cpu(), beta.grad, rtol=lim, atol=lim * beta.grad.abs().max().item())


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("geom", [(3, 1, 1, 1), (3, 2, 0, 1), (5, 1, 2, 1), (5, 2, 1, 1), (3, 1, 2, 2), (5, 1, 4, 2)])   # kernel, stride, front padding, dilation
@pytest.mark.parametrize("shape", [(2, 12, 16, 48), (1, 8, 8, 528), (2, 7, 9, 144)])
def test_dwconv2d_static_same_padding(code, geom, shape):
    """efficientnet-pytorch's depthwise Conv2dStaticSamePadding (kernel 3 / 5, stride 1 / 2; at stride 2 the padding is (0, 1) resp.
    (1, 2): TF's "same" for an even nominal image size) - forward, data gradient (with and without accumulation) and weight gradient
    against F.conv2d on the explicitly padded input + autograd; 528 channels = two slabs (64 + 2 vectors)."""
    L = lib()
    k, s, lo, dil = geom          # dilation 2 with padding (k // 2) * 2: a stage after smp's replace_strides_with_dilation (DeepLabV3+)
    n, h, w, c = shape
    g = torch.Generator().manual_seed(61)
    ho, wo = -(-h // s), -(-w // s)
    hi_h, hi_w = max((ho - 1) * s + (k - 1) * dil + 1 - h, 0) - lo, max((wo - 1) * s + (k - 1) * dil + 1 - w, 0) - lo
    x = rounded(torch.randn(n, c, h, w, generator=g), code).requires_grad_()
    wt = (torch.randn(c, 1, k, k, generator=g) / k).requires_grad_()
    y = F.conv2d(F.pad(x, (lo, hi_w, lo, hi_h)), wt, stride=s, groups=c, dilation=dil)
    assert y.shape[2:] == (ho, wo)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    xd, dyd = to_nhwc(x.detach(), code), to_nhwc(dy, code)
    wd = wt.detach().reshape(c, k * k).contiguous().to(DEV)
    yd = torch.full((n, ho, wo, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_dwconv2d(code, L.ptr(xd), L.ptr(wd), L.ptr(yd), n, h, w, c, k, s, lo, dil, ho, wo, 0, None))
    dx = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_dwconv2d_bwd_data(code, L.ptr(dyd), L.ptr(wd), L.ptr(dx), n, h, w, c, k, s, lo, dil, ho, wo, 0, None))
    twice = dx.clone()
    L.check(L.lib.vs_dwconv2d_bwd_data(code, L.ptr(dyd), L.ptr(wd), L.ptr(twice), n, h, w, c, k, s, lo, dil, ho, wo, 1, None))
    wsb = L.lib.vs_dwconv2d_wgrad_workspace(c, k)
    ws = torch.empty(wsb // 4, device=DEV)
    dw = torch.full((c, k * k), float("nan"), device=DEV)
    L.check(L.lib.vs_dwconv2d_wgrad(code, L.ptr(xd), L.ptr(dyd), L.ptr(dw), n, h, w, c, k, s, lo, dil, ho, wo, 0, L.ptr(ws), wsb, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    assert torch.allclose(from_nhwc(twice), 2 * x.grad, **tol(code, 2 * x.grad.abs().max().item()))
    ref = wt.grad.reshape(c, k * k)
    assert torch.allclose(dw.cpu(), ref, rtol=1e-3, atol=1e-4 * ref.abs().max().item())


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("cout", [40, 48])
def test_efficientnet_stem_on_a_single_channel(code, cout):
    """smp's EfficientNet stem on greyscale slices: nn.Conv2d(1, cout, 3, stride=2, bias=False) behind static same padding (0, 1), the fp32
    slices broadcast over the output channels (vs_dwconv2d with x_single_channel = 1) - forward and weight gradient."""
    L = lib()
    n, h, w = 2, 16, 24
    g = torch.Generator().manual_seed(71)
    x = torch.randn(n, 1, h, w, generator=g)
    wt = (torch.randn(cout, 1, 3, 3, generator=g) / 3).requires_grad_()
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), wt, stride=2)
    dy = rounded(torch.randn(y.shape, generator=g), code)
    y.backward(dy)
    xd = x[:, 0].contiguous().to(DEV)
    wd = wt.detach().reshape(cout, 9).contiguous().to(DEV)
    yd = torch.full((n, h // 2, w // 2, cout), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_dwconv2d(code, L.ptr(xd), L.ptr(wd), L.ptr(yd), n, h, w, cout, 3, 2, 0, 1, h // 2, w // 2, 1, None))
    wsb = L.lib.vs_dwconv2d_wgrad_workspace(cout, 3)
    ws = torch.empty(wsb // 4, device=DEV)
    dw = torch.full((cout, 9), float("nan"), device=DEV)
    dyd = to_nhwc(dy, code)
    L.check(L.lib.vs_dwconv2d_wgrad(code, L.ptr(xd), L.ptr(dyd), L.ptr(dw), n, h, w, cout, 3, 2, 0, 1, h // 2, w // 2, 1, L.ptr(ws), wsb, None))
    sync()
    assert torch.allclose(from_nhwc(yd), y.detach(), **tol(code, y.abs().max().item()))
    ref = wt.grad.reshape(cout, 9)
    assert torch.allclose(dw.cpu(), ref, rtol=1e-3, atol=1e-4 * ref.abs().max().item())


@pytest.mark.parametrize("code", CODES)
def test_sample_scale_add_and_rowsum(code):
    """drop_connect + residual sum in one sweep (mask per sample), and the per-sample sums / dot products for channel counts the fast
    kernels do not take (144 = 18 vectors)."""
    L = lib()
    n, h, w, c = 3, 5, 4, 144
    g = torch.Generator().manual_seed(81)
    x, skip = rounded(torch.randn(n, c, h, w, generator=g), code), rounded(torch.randn(n, c, h, w, generator=g), code)
    mask = torch.tensor([0.0, 1.25, 1.25])
    xd, sd, md = to_nhwc(x, code), to_nhwc(skip, code), mask.to(DEV)
    y = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_sample_scale_add(code, L.ptr(xd), L.ptr(md), L.ptr(sd), L.ptr(y), n, h * w * c, None))
    y2 = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_sample_scale_add(code, L.ptr(xd), L.ptr(md), None, L.ptr(y2), n, h * w * c, None))
    y3 = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_sample_scale_add(code, L.ptr(xd), None, L.ptr(sd), L.ptr(y3), n, h * w * c, None))
    mean = torch.full((n, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_spatial_sum(code, L.ptr(xd), L.ptr(mean), n, h * w, c, 1.0 / (h * w), None))
    dot = torch.full((n, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_channel_dot(code, L.ptr(xd), L.ptr(sd), L.ptr(dot), n, h * w, c, None))
    sync()
    m4 = mask[:, None, None, None]
    assert torch.allclose(from_nhwc(y), x * m4 + skip, **tol(code, 4.0))
    assert torch.allclose(from_nhwc(y2), x * m4, **tol(code, 4.0))
    assert torch.allclose(from_nhwc(y3), x + skip, **tol(code, 4.0))
    assert torch.allclose(mean.float().cpu(), x.mean((2, 3)), **tol(code, 1.0))
    assert torch.allclose(dot.float().cpu(), (x * skip).sum((2, 3)), **tol(code, 10.0))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(2, 96, 80, 144), (1, 64, 64, 2688), (3, 40, 40, 48)])
def test_sample_rowsum_split_over_workgroups(code, shape):
    """vs_sample_rowsum_ws: the per-sample means / dot products of prediction-sized maps with a sample's rows spread over up to 64
    workgroups and a fixed-order finish - against torch, and bit-identical between two calls."""
    L = lib()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(91)
    x, y = rounded(torch.randn(n, c, h, w, generator=g), code), rounded(torch.randn(n, c, h, w, generator=g), code)
    xd, yd = to_nhwc(x, code), to_nhwc(y, code)
    wsb = L.lib.vs_sample_rowsum_workspace(n, c)
    ws = torch.empty(wsb // 4, device=DEV)
    outs = []
    for _ in range(2):
        mean = torch.full((n, c), float("nan"), device=DEV, dtype=tdtype(code))
        dot = torch.full((n, c), float("nan"), device=DEV, dtype=tdtype(code))
        L.check(L.lib.vs_sample_rowsum_ws(code, L.ptr(xd), None, L.ptr(mean), n, h * w, c, 1.0 / (h * w), L.ptr(ws), wsb, None))
        L.check(L.lib.vs_sample_rowsum_ws(code, L.ptr(xd), L.ptr(yd), L.ptr(dot), n, h * w, c, 1.0, L.ptr(ws), wsb, None))
        sync()
        outs.append((mean.clone(), dot.clone()))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.allclose(outs[0][0].float().cpu(), x.mean((2, 3)), **tol(code, 1.0))
    ref = (x * y).sum((2, 3))
    assert torch.allclose(outs[0][1].float().cpu(), ref, **tol(code, ref.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(700, 448), (96, 136), (2048, 304), (33, 8), (5000, 2688)])
def test_colsum_any_channel_count(code, shape):
    """vs_colsum (bias gradients: the column sums of a gradient map) for channel counts beyond 8 .. 128 / multiples of 256 - the
    EfficientNet feature widths 448 / 136 that smp.MAnet's biased PAB convolutions see, and slabs with a narrower tail."""
    L = lib()
    rows, c = shape
    g = torch.Generator().manual_seed(17)
    x = rounded(torch.randn(rows, c, generator=g), code)
    xd = x.to(DEV, tdtype(code)).contiguous()
    wsb = L.lib.vs_colsum_workspace(c)
    ws = torch.empty(wsb // 4, device=DEV)
    out = torch.full((c,), float("nan"), device=DEV)
    L.check(L.lib.vs_colsum(code, L.ptr(xd), rows, c, L.ptr(out), L.ptr(ws), wsb, None))
    sync()
    ref = x.double().sum(0).float()
    assert torch.allclose(out.cpu(), ref, rtol=1e-4, atol=1e-3 * rows ** 0.5)


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("geom", [(3, 1, 1, 1), (5, 2, 1, 1), (5, 1, 4, 2)])
def test_dwconv2d_affine_eval_form(code, geom):
    """Evaluation-mode MBConv middle: depthwise convolution + BatchNorm (running statistics folded by vs_bn_fold) + swish in one sweep
    (vs_dwconv2d_affine) against F.conv2d + F.batch_norm(training=False) + swish."""
    L = lib()
    k, s, lo, dil = geom
    n, h, w, c = 2, 12, 16, 144
    g = torch.Generator().manual_seed(63)
    ho, wo = -(-h // s), -(-w // s)
    hi_h, hi_w = max((ho - 1) * s + (k - 1) * dil + 1 - h, 0) - lo, max((wo - 1) * s + (k - 1) * dil + 1 - w, 0) - lo
    x = rounded(torch.randn(n, c, h, w, generator=g), code)
    wt = torch.randn(c, 1, k, k, generator=g) / k
    gamma, beta = 1 + 0.2 * torch.randn(c, generator=g), 0.2 * torch.randn(c, generator=g)
    rm, rv = 0.1 * torch.randn(c, generator=g), 0.75 + 0.5 * torch.rand(c, generator=g)
    z = F.conv2d(F.pad(x, (lo, hi_w, lo, hi_h)), wt, stride=s, groups=c, dilation=dil)
    v = F.batch_norm(z, rm, rv, gamma, beta, False, 0.01, 1e-3)
    ref = v * torch.sigmoid(v)
    dev = lambda t_: t_.contiguous().to(DEV)
    scale, shift = torch.empty(c, device=DEV), torch.empty(c, device=DEV)
    gd, bd, rmd, rvd = dev(gamma), dev(beta), dev(rm), dev(rv)
    L.check(L.lib.vs_bn_fold(L.ptr(gd), L.ptr(bd), L.ptr(rmd), L.ptr(rvd), 1e-3, L.ptr(scale), L.ptr(shift), c, None))
    xd, wd = to_nhwc(x, code), dev(wt.reshape(c, k * k))
    yd = torch.full((n, ho, wo, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_dwconv2d_affine(code, L.ptr(xd), L.ptr(wd), L.ptr(scale), L.ptr(shift), 2, L.ptr(yd), n, h, w, c, k, s, lo, dil, ho, wo, None))
    sync()
    assert torch.allclose(from_nhwc(yd), ref, **tol(code, ref.abs().max().item()))


@pytest.mark.parametrize("code", CODES)
@pytest.mark.parametrize("shape", [(3, 6, 5, 64), (2, 40, 48, 128), (4, 1, 1, 512)])
def test_radix2_softmax_and_gated_sum_fwd_bwd(code, shape):
    """timm's SplitAttnConv2d tail (radix 2, cardinality 1): RadixSoftmax on the attention logits [n][2 c] and the attention-weighted sum
    of the two splits - restated in torch exactly as split_attn.py writes it, forward and (autograd) backward: the data gradient, the
    attention's gradient (vs_sample_rowsum_b: sums of x * dout with dout repeated per split) and the softmax's."""
    L = lib()
    n, h, w, c = shape
    g = torch.Generator().manual_seed(101)
    x = rounded(torch.randn(n, 2 * c, h, w, generator=g), code).requires_grad_()
    z = rounded(torch.randn(n, 2 * c, generator=g), code).requires_grad_()
    att = torch.softmax(z.view(n, 1, 2, -1).transpose(1, 2), dim=1).reshape(n, -1)
    out = (x.reshape(n, 2, c, h, w) * att.reshape(n, 2, c, 1, 1)).sum(1)
    dout = rounded(torch.randn(out.shape, generator=g), code)
    out.backward(dout)
    dev = lambda t_: t_.detach().contiguous().to(DEV, tdtype(code))
    xd, zd, dd = to_nhwc(x.detach(), code), dev(z), to_nhwc(dout, code)
    ad = torch.full((n, 2 * c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_radix2_softmax(code, L.ptr(zd), L.ptr(ad), n, c, None))
    od = torch.full((n, h, w, c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_radix2_gated_sum(code, L.ptr(xd), L.ptr(ad), L.ptr(od), n, h * w, c, None))
    dx = torch.full_like(xd, float("nan"))
    L.check(L.lib.vs_radix2_gated_sum_bwd(code, L.ptr(dd), L.ptr(ad), L.ptr(dx), n, h * w, c, None))
    wsb = L.lib.vs_sample_rowsum_workspace(n, 2 * c)
    ws = torch.empty(wsb // 4, device=DEV)
    da = torch.full((n, 2 * c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_sample_rowsum_b(code, L.ptr(xd), L.ptr(dd), c, L.ptr(da), n, h * w, 2 * c, L.ptr(ws), wsb, None))
    dz = torch.full((n, 2 * c), float("nan"), device=DEV, dtype=tdtype(code))
    L.check(L.lib.vs_radix2_softmax_bwd(code, L.ptr(da), L.ptr(ad), L.ptr(dz), n, c, None))
    sync()
    assert torch.allclose(ad.float().cpu(), att.detach(), **tol(code, 1.0))
    assert torch.allclose(from_nhwc(od), out.detach(), **tol(code, out.abs().max().item()))
    assert torch.allclose(from_nhwc(dx), x.grad, **tol(code, x.grad.abs().max().item()))
    lim = 5e-2 if code else 1e-4
    assert torch.allclose(dz.float().cpu(), z.grad, rtol=lim, atol=lim * z.grad.abs().max().item())
