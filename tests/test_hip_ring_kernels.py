"""-m gpu: the kernels bench.py's batch-32 training step actually launches, each against torch CPU fp32, at launch sizes that
SELECT them.  `ring_mode` (csrc/conv_igemm.hip) hands a stride-1 3x3 layer to ring::conv_ring_kernel only from 224 (64-cout tiles)
/ 128 (32-cout tiles) workgroups up, so the small shapes of test_hip_ops.py never reach it; here every case runs 32 images and
asserts through vs_conv2d_train_variant (kind 6 = LDS-DMA ring) that the ring kernel is what ran.  The launches carry the
TRAINING epilogues the network plan gives them (vs_conv2d_train, include/volseg_hip.h): statistics of the fp32 accumulators in
fixed-point bins or partial rows, the BatchNorm-backward first sweep (ReLU mask + sum g / sum g xhat) in the data-gradient
epilogue, the split + 2x2-pooled data gradient of a decoder concatenation, the zero-stuffed source of a stride-2 layer's data
gradient, and normalise-on-load.  Reference call sites: the model's forward and loss.backward() of _train_one_batch
(volume_segmantics/model/operations/vol_seg_2d_trainer.py:424, 429)."""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from hip_helpers import DEV, conv_desc, from_nhwc, lib, rounded, sync, to_nhwc, tol, w_krsc

pytestmark = pytest.mark.gpu
BF = 1          # VS_BF16: the ring kernels are the bf16 training kernels
N = 32          # BASELINE configs[1]'s batch

# (h, w, c0, c1, up0, cout, expected variant code, where the batch-32 step launches it)
MODE1, MODE2, MODE3 = 64296, 32296, 32296    # cout tile * 1000 + 2 * 100 + 9 * 10 + 6 (ring); mode 3 = pairs of 8 x 8 images, 32 couts
FWD_CASES = [
    pytest.param((32, 32, 128, 0, 0, 128, MODE1), id="layer2-128to128@32-mode1"),
    pytest.param((16, 16, 256, 0, 0, 256, MODE2), id="layer3-256to256@16-mode2"),
    pytest.param((16, 16, 512, 256, 1, 256, MODE2), id="dec0.conv1-up512+256to256@16-mode2"),
    pytest.param((32, 32, 256, 128, 1, 128, MODE1), id="dec1.conv1-up256+128to128@32-mode1"),
    pytest.param((8, 8, 512, 0, 0, 512, MODE3), id="layer4-512to512@8-mode3"),
]


def _ptr(t):
    return None if t is None else t.data_ptr()


def _inputs(g, h, w, c0, c1, up0, cout):
    x0 = rounded(torch.randn(N, c0, h >> (1 if up0 == 1 else 0), w >> (1 if up0 == 1 else 0), generator=g), BF)
    x1 = rounded(torch.randn(N, c1, h, w, generator=g), BF) if c1 else None
    xin = F.interpolate(x0, scale_factor=2, mode="nearest") if up0 == 1 else x0
    if c1:
        xin = torch.cat([xin, x1], 1)
    wt = rounded(torch.randn(cout, c0 + c1, 3, 3, generator=g) / ((c0 + c1) * 9) ** 0.5, BF)
    return x0, x1, xin, wt


def _assert_ring(L, d, t, code):
    got = L.lib.vs_conv2d_train_variant(d, t)
    assert got == code, f"this launch selects kernel variant {got}, not the ring kernel {code} the batch-32 step runs"


@pytest.mark.parametrize("case", FWD_CASES)
@pytest.mark.parametrize("stats", ["bins", "rows"])
def test_ring_forward_with_batch_statistics(case, stats):
    """Training forward: z = conv(x, w) stored bf16, and the per-channel sum / sum of squares of the fp32 accumulators (what
    F.batch_norm(training=True) reduces) from the epilogue - as 64-bit fixed-point bins (the default of the step) and as fp32 partial
    rows.  Against F.conv2d on the same bf16-rounded operands."""
    L = lib()
    h, w, c0, c1, up0, cout, code = case
    g = torch.Generator().manual_seed(31)
    x0, x1, xin, wt = _inputs(g, h, w, c0, c1, up0, cout)
    ref = F.conv2d(xin, wt, padding=1)
    d = conv_desc(L, BF, N, h, w, c0, cout, 3, 1, 1, c1=c1, up0=up0)
    t = L.ConvTrain()
    if stats == "bins":
        nb = 16
        bins = torch.zeros((nb, 2, cout), dtype=torch.int64, device=DEV)
        t.stats_bins, t.stats_nb = bins.data_ptr(), nb
    else:
        rows = L.lib.vs_conv2d_stat_rows(d, t)
        assert rows > 0
        part = torch.full((rows, 2, cout), float("nan"), device=DEV)
        t.stats_partial = part.data_ptr()
    _assert_ring(L, d, t, code)
    x0d, x1d, wd = to_nhwc(x0, BF), (to_nhwc(x1, BF) if c1 else None), w_krsc(wt, BF)
    y = torch.full((N, h, w, cout), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.vs_conv2d_train(d, _ptr(x0d), _ptr(x1d), _ptr(wd), None, _ptr(y), None, C.byref(t), None))
    sync()
    got = from_nhwc(y)
    assert torch.isfinite(got).all()
    assert torch.allclose(got, ref, **tol(BF, ref.abs().max().item())), (got - ref).abs().max()
    if stats == "bins":
        tot = bins.sum(0).cpu().double()
        s1, s2 = tot[0] / L.lib.vs_stat_scale(0), tot[1] / L.lib.vs_stat_scale(1)
    else:
        tot = part.double().sum(0).cpu()
        s1, s2 = tot[0], tot[1]
    r1, r2 = ref.double().sum((0, 2, 3)), (ref.double() ** 2).sum((0, 2, 3))
    rows_per_c = N * h * w
    # sums of fp32 accumulators in another order: a few 1e-7 relative per term; the bins add 2^-24 / 2^-16 per tile
    assert torch.allclose(s1, r1, rtol=1e-4, atol=1e-4 * (r2.max().item() * rows_per_c) ** 0.5), (s1 - r1).abs().max()
    assert torch.allclose(s2, r2, rtol=1e-4), ((s2 - r2) / r2).abs().max()
    mean, var = s1 / rows_per_c, s2 / rows_per_c - (s1 / rows_per_c) ** 2
    rmean, rvar = ref.double().mean((0, 2, 3)), ref.double().var((0, 2, 3), unbiased=False)
    assert torch.allclose(mean, rmean, atol=1e-5) and torch.allclose(var, rvar, rtol=1e-4)


def _prep(L, w_oihw):
    cout, cin, k, _ = w_oihw.shape
    wf = w_oihw.permute(0, 2, 3, 1).contiguous().to(DEV)
    wtr = torch.empty((cin, k, k, cout), device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.vs_weights_prepare(BF, L.ptr(wf), None, L.ptr(wtr), cout, k * k, cin, None))
    return wtr


DGRAD_CASES = [
    pytest.param((32, 32, 128, MODE1), id="layer2-mode1"),
    pytest.param((16, 16, 256, MODE2), id="layer3-mode2"),
    pytest.param((8, 8, 512, MODE3), id="layer4-mode3"),
]


@pytest.mark.parametrize("case", DGRAD_CASES)
@pytest.mark.parametrize("mask", ["from_activation", "recomputed", "none"])
@pytest.mark.parametrize("earlier", [False, True], ids=["sole_gradient", "plus_earlier_contributions"])
def test_ring_dgrad_with_batchnorm_backward_epilogue(case, mask, earlier):
    """The data gradient that COMPLETES the gradient of a conv + BN (+ ReLU) unit's activation, as the backward pass launches it: a
    stride-1 convolution of dy with the flipped weights whose epilogue adds the contributions that arrived earlier (`residual`),
    applies the unit's ReLU mask - from the saved activation (units with a residual input) or recomputed from the pre-norm tensor -
    stores the masked gradient g and writes per-tile partials of sum g and sum g * xhat (BatchNorm backward's first sweep, which
    then never runs).  Against autograd's conv_transpose and the BatchNorm-backward sums in fp64."""
    L = lib()
    h, w, c, code = case
    g = torch.Generator().manual_seed(37)
    dy = rounded(torch.randn(N, c, h, w, generator=g), BF)
    wt = rounded(torch.randn(c, c, 3, 3, generator=g) / (c * 9) ** 0.5, BF)
    z = rounded(torch.randn(N, c, h, w, generator=g) * 1.5 + 0.2, BF)             # the unit's pre-norm output
    mean, invstd = z.mean((0, 2, 3)), 1 / torch.sqrt(z.var((0, 2, 3), unbiased=False) + 1e-5)
    gamma, beta = torch.rand(c, generator=g) + 0.5, torch.randn(c, generator=g) * 0.3
    bc = lambda v: v.view(1, -1, 1, 1)
    pre = (z - bc(mean)) * bc(invstd * gamma) + bc(beta)
    if mask == "recomputed":      # keep the recomputed pre-activation clear of zero: one fp32 contraction must not flip a mask bit
        near = pre.abs() < 1e-3
        z = torch.where(near, rounded(z + 0.5, BF), z)
        pre = (z - bc(mean)) * bc(invstd * gamma) + bc(beta)
        assert not (pre.abs() < 1e-5).any()
    y = rounded(F.relu(pre), BF)
    if mask == "from_activation":   # a residual unit: the activation also holds the shortcut, the mask is y > 0
        y = rounded(F.relu(pre + torch.randn(pre.shape, generator=g)), BF)
    prev = rounded(torch.randn(N, c, h, w, generator=g), BF) if earlier else None
    full = F.conv_transpose2d(dy, wt, padding=1) + (prev if earlier else 0)
    m = torch.ones_like(full, dtype=torch.bool) if mask == "none" else ((y > 0) if mask == "from_activation" else (pre > 0))
    ref_g = full * m
    d = conv_desc(L, BF, N, h, w, c, c, 3, 1, 1)
    t = L.ConvTrain()
    zd, yd = to_nhwc(z, BF), to_nhwc(y, BF)
    md, isd, gd, bd = mean.to(DEV), invstd.to(DEV), gamma.to(DEV), beta.to(DEV)
    t.bz, t.bmean, t.binvstd = zd.data_ptr(), md.data_ptr(), isd.data_ptr()
    t.brelu = 0 if mask == "none" else 1
    if mask == "from_activation":
        t.by = yd.data_ptr()
    if mask == "recomputed":
        t.bgamma, t.bbeta = gd.data_ptr(), bd.data_ptr()
    rows = L.lib.vs_conv2d_stat_rows(d, t)
    part = torch.full((rows, 2, c), float("nan"), device=DEV)
    t.bstats_partial = part.data_ptr()
    _assert_ring(L, d, t, code)
    dyd, wtr = to_nhwc(dy, BF), _prep(L, wt)
    pd = to_nhwc(prev, BF) if earlier else None
    out = torch.full((N, h, w, c), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.vs_conv2d_train(d, _ptr(dyd), None, _ptr(wtr), _ptr(pd), _ptr(out), None, C.byref(t), None))
    sync()
    got = from_nhwc(out)
    assert torch.isfinite(got).all()
    assert (got[~m] == 0).all(), "masked-off elements must be exactly zero"
    assert torch.allclose(got, ref_g, **tol(BF, ref_g.abs().max().item())), (got - ref_g).abs().max()
    # the statistics see exactly what was stored: compare with sums over the device's own bf16 g
    xhat = ((z - bc(mean)) * bc(invstd)).double()
    s = part.double().sum(0).cpu()
    r1, r2 = got.double().sum((0, 2, 3)), (got.double() * xhat).sum((0, 2, 3))
    scale = (got.double() ** 2).sum((0, 2, 3)).sqrt().max().item()
    assert torch.allclose(s[0], r1, rtol=1e-4, atol=1e-4 * scale), (s[0] - r1).abs().max()
    assert torch.allclose(s[1], r2, rtol=1e-4, atol=1e-4 * scale * 3), (s[1] - r2).abs().max()


@pytest.mark.parametrize("case", [pytest.param((16, 16, 256, 512, 256, MODE1), id="dec0.conv1-256to512+256@16"),
                                  pytest.param((32, 32, 128, 256, 128, MODE1), id="dec1.conv1-128to256+128@32"),
                                  pytest.param((16, 16, 256, 480, 32, MODE2), id="split-at-480-32couts@16")])
def test_ring_dgrad_split_through_the_concat_and_pooled_through_the_upsampling(case):
    """Data gradient of a decoder block's first convolution (input = cat(nearest-x2 upsampling of the deeper tensor, skip)): ONE launch
    whose couts below split_c are summed over 2 x 2 pixel blocks and stored at half resolution (F.interpolate's backward) and whose
    couts from split_c up go to the skip tensor's gradient.  Against autograd."""
    L = lib()
    h, w, cmid, c_up, c_skip, code = case
    g = torch.Generator().manual_seed(41)
    x0 = rounded(torch.randn(N, c_up, h // 2, w // 2, generator=g), BF).requires_grad_()
    x1 = rounded(torch.randn(N, c_skip, h, w, generator=g), BF).requires_grad_()
    wt = rounded(torch.randn(cmid, c_up + c_skip, 3, 3, generator=g) / ((c_up + c_skip) * 9) ** 0.5, BF)
    yy = F.conv2d(torch.cat([F.interpolate(x0, scale_factor=2, mode="nearest"), x1], 1), wt, padding=1)
    dy = rounded(torch.randn(yy.shape, generator=g), BF)
    yy.backward(dy)
    d = conv_desc(L, BF, N, h, w, cmid, c_up + c_skip, 3, 1, 1, split_c=c_up)
    t = L.ConvTrain()
    t.pool0 = 1
    _assert_ring(L, d, t, code)
    dyd, wtr = to_nhwc(dy, BF), _prep(L, wt)
    dx0 = torch.full((N, h // 2, w // 2, c_up), float("nan"), device=DEV, dtype=torch.bfloat16)
    dskip = torch.full((N, h, w, c_skip), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.vs_conv2d_train(d, _ptr(dyd), None, _ptr(wtr), None, _ptr(dx0), _ptr(dskip), C.byref(t), None))
    sync()
    assert torch.allclose(from_nhwc(dskip), x1.grad, **tol(BF, x1.grad.abs().max().item()))
    assert torch.allclose(from_nhwc(dx0), x0.grad, **tol(BF, x0.grad.abs().max().item()))


@pytest.mark.parametrize("case", [pytest.param((32, 32, 256, 128, MODE1), id="layer3.0.conv1-dgrad-256@16to128@32"),
                                  pytest.param((16, 16, 512, 256, MODE2), id="layer4.0.conv1-dgrad-512@8to256@16")])
@pytest.mark.parametrize("earlier", [False, True])
def test_ring_dgrad_of_a_stride2_layer_reads_the_zero_stuffed_gradient(case, earlier):
    """Data gradient of a stride-2 3x3 convolution = stride-1 convolution, flipped weights, over the output gradient with zeros
    between its elements; the loader reads dy at the even positions and never materialises the stuffed tensor (up0 = 2).  With and
    without the earlier contributions to the same gradient added in the epilogue.  Against autograd."""
    L = lib()
    h, w, cout, cin, code = case              # the forward layer: cin @ h x w -> cout @ h/2 x w/2
    g = torch.Generator().manual_seed(43)
    x = rounded(torch.randn(N, cin, h, w, generator=g), BF).requires_grad_()
    wt = rounded(torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5, BF)
    yy = F.conv2d(x, wt, stride=2, padding=1)
    dy = rounded(torch.randn(yy.shape, generator=g), BF)
    yy.backward(dy)
    prev = rounded(torch.randn(N, cin, h, w, generator=g), BF) if earlier else None
    ref = x.grad + (prev if earlier else 0)
    d = conv_desc(L, BF, N, h, w, cout, cin, 3, 1, 1, up0=2)
    t = L.ConvTrain()
    _assert_ring(L, d, t, code)
    dyd, wtr = to_nhwc(dy, BF), _prep(L, wt)
    pd = to_nhwc(prev, BF) if earlier else None
    dx = torch.full((N, h, w, cin), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.vs_conv2d_train(d, _ptr(dyd), None, _ptr(wtr), _ptr(pd), _ptr(dx), None, C.byref(t), None))
    sync()
    assert torch.allclose(from_nhwc(dx), ref, **tol(BF, ref.abs().max().item())), (from_nhwc(dx) - ref).abs().max()


@pytest.mark.parametrize("case", [pytest.param((32, 32, 128, 128, MODE1), id="128to128@32-mode1"),
                                  pytest.param((16, 16, 256, 256, MODE2), id="256to256@16-mode2"),
                                  pytest.param((8, 8, 512, 512, MODE3), id="512to512@8-mode3")])
def test_ring_forward_normalising_its_input_on_load(case):
    """Normalise-on-load inside the ring kernel (conv -> BN -> ReLU -> conv pairs): the source is the producer's PRE-norm tensor and
    its statistics bins; the launch finalises mean / invstd / running statistics, convolves relu(bn(z)) rounded to bf16 - the
    bits the normalisation sweep would have stored - and leaves that activation behind for the weight gradient.  Against
    F.batch_norm + F.relu + F.conv2d."""
    L = lib()
    h, w, c0, cout, code = case
    g = torch.Generator().manual_seed(47)
    z0 = rounded(torch.randn(N, c0, h, w, generator=g) * 1.3 + 0.1, BF)
    wt = rounded(torch.randn(cout, c0, 3, 3, generator=g) / (c0 * 9) ** 0.5, BF)
    gamma, beta = torch.rand(c0, generator=g) + 0.5, torch.randn(c0, generator=g) * 0.3
    rm, rv = torch.randn(c0, generator=g) * 0.1, torch.rand(c0, generator=g) + 0.5
    rows = N * h * w
    s1, s2 = z0.double().sum((0, 2, 3)), (z0.double() ** 2).sum((0, 2, 3))
    nb = 16
    bins = torch.zeros((nb, 2, c0), dtype=torch.int64)
    bins[3, 0] = torch.round(s1 * L.lib.vs_stat_scale(0)).long()       # any row: the consumer sums them all
    bins[5, 1] = torch.round(s2 * L.lib.vs_stat_scale(1)).long()
    mean = bins[:, 0].sum(0).double() / L.lib.vs_stat_scale(0) / rows
    var = bins[:, 1].sum(0).double() / L.lib.vs_stat_scale(1) / rows - mean ** 2
    invstd = 1 / torch.sqrt(var + 1e-5)
    bc = lambda v: v.view(1, -1, 1, 1).float()
    act = rounded(F.relu((z0 - bc(mean)) * bc(invstd * gamma.double()) + bc(beta)), BF)
    ref = F.conv2d(act, wt, padding=1)
    d = conv_desc(L, BF, N, h, w, c0, cout, 3, 1, 1)
    t = L.ConvTrain()
    binsd, gd, bd, rmd, rvd = bins.to(DEV), gamma.to(DEV), beta.to(DEV), rm.clone().to(DEV), rv.clone().to(DEV)
    md, isd = torch.full((c0,), float("nan"), device=DEV), torch.full((c0,), float("nan"), device=DEV)
    yd = torch.full((N, h, w, c0), float("nan"), device=DEV, dtype=torch.bfloat16)
    t.nl_bins, t.nl_nb, t.nl_rows, t.nl_eps, t.nl_mom = binsd.data_ptr(), nb, rows, 1e-5, 0.1
    t.nl_mean, t.nl_invstd, t.nl_rm, t.nl_rv = md.data_ptr(), isd.data_ptr(), rmd.data_ptr(), rvd.data_ptr()
    t.nl_gamma, t.nl_beta, t.nl_y = gd.data_ptr(), bd.data_ptr(), yd.data_ptr()
    _assert_ring(L, d, t, code)
    z0d, wd = to_nhwc(z0, BF), w_krsc(wt, BF)
    out = torch.full((N, h, w, cout), float("nan"), device=DEV, dtype=torch.bfloat16)
    L.check(L.lib.vs_conv2d_train(d, _ptr(z0d), None, _ptr(wd), None, _ptr(out), None, C.byref(t), None))
    sync()
    assert torch.allclose(md.cpu().double(), mean, atol=1e-6) and torch.allclose(isd.cpu().double(), invstd, rtol=1e-5)
    got_act = from_nhwc(yd)
    # one bf16 ulp where the fp32 affine rounds the other way
    assert torch.allclose(got_act, act, rtol=1e-2, atol=1e-6), (got_act - act).abs().max()
    assert (got_act != act).float().mean().item() < 0.02
    assert torch.allclose(from_nhwc(out), ref, **tol(BF, ref.abs().max().item())), (from_nhwc(out) - ref).abs().max()
    assert torch.allclose(rmd.cpu().double(), 0.9 * rm.double() + 0.1 * mean, atol=1e-5)
    assert torch.allclose(rvd.cpu().double(), 0.9 * rv.double() + 0.1 * var * rows / (rows - 1), rtol=1e-4)


@pytest.mark.parametrize("case", [pytest.param((32, 32, 128, 0, 0, 128), id="layer2-128to128@32"),
                                  pytest.param((16, 16, 256, 0, 0, 256), id="layer3-256to256@16"),
                                  pytest.param((16, 16, 512, 256, 1, 256), id="dec0.conv1-up512+256to256@16"),
                                  pytest.param((8, 8, 512, 0, 0, 512), id="layer4-512to512@8-two-images-per-tile"),
                                  pytest.param((64, 64, 64, 0, 0, 64), id="layer1-64to64@64")])
def test_ring_weight_gradient_at_32_images(case):
    """ring::conv_wgrad_ring_kernel - the weight gradients of the stride-1 3x3 layers in the batch-32 step (30 of its 46 weight-gradient
    launches) - at 32 images against autograd (loss.backward(), vol_seg_2d_trainer.py:429); the same launch twice gives the same bits
    (slabs summed in a fixed order), and the register-staged kernel (`wgrad_ring` 0) agrees to fp32 summation order."""
    L = lib()
    h, w, c0, c1, up0, cout = case
    g = torch.Generator().manual_seed(53)
    x0, x1, xin, wt = _inputs(g, h, w, c0, c1, up0, cout)
    wt = wt.requires_grad_()
    y = F.conv2d(xin, wt, padding=1)
    dy = rounded(torch.randn(y.shape, generator=g), BF)
    y.backward(dy)
    ref = wt.grad.permute(0, 2, 3, 1)
    d = conv_desc(L, BF, N, h, w, c0, cout, 3, 1, 1, c1=c1, up0=up0)
    x0d, x1d, dyd = to_nhwc(x0, BF), (to_nhwc(x1, BF) if c1 else None), to_nhwc(dy, BF)
    outs = []
    old = L.lib.vs_get_option(b"wgrad_ring")
    try:
        for ring in (1, 1, 0):
            L.set_option("wgrad_ring", ring)
            ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
            ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
            dw = torch.full((cout, 3, 3, c0 + c1), float("nan"), device=DEV)
            L.check(L.lib.vs_conv2d_wgrad(d, _ptr(x0d), _ptr(x1d), _ptr(dyd), _ptr(dw), _ptr(ws), ws_bytes, None))
            sync()
            outs.append(dw)
    finally:
        L.set_option("wgrad_ring", old)
    base = outs[0]
    scale = ref.abs().max().item()
    assert torch.isfinite(base).all()
    assert torch.allclose(base.cpu(), ref, rtol=1e-3, atol=1e-3 * scale), (base.cpu() - ref).abs().max()
    assert torch.equal(outs[1], base)
    assert torch.allclose(outs[2], base, rtol=1e-5, atol=1e-5 * scale), (outs[2] - base).abs().max()


@pytest.mark.parametrize("case", [pytest.param((3, 128, 128, 0), id="3x128x128"), pytest.param((2, 24, 256, 0), id="2x24x256-ragged-row-ranges"),
                                  pytest.param((1, 6, 512, 0), id="1x6x512"), pytest.param((32, 256, 256, 0), id="32x256x256-the-step's-launch"),
                                  pytest.param((3, 128, 128, 1), id="up-3x128x128"), pytest.param((2, 28, 256, 1), id="up-2x28x256-ragged-row-ranges"),
                                  pytest.param((1, 4, 512, 1), id="up-1x4x512"), pytest.param((32, 256, 256, 1), id="up-32x256x256-the-step's-launch")])
def test_row_streaming_weight_gradient_of_the_full_resolution_layers(case):
    """rows::conv_wgrad_rows16_kernel / conv_wgrad_rows_up32_kernel - the weight gradients of the 16-cout 3x3 layers at full
    resolution (the last decoder block's two convolutions, the first behind the nearest upsampling of its 32-channel input, and the
    segmentation head in the batch-32 step) - against autograd (loss.backward(), vol_seg_2d_trainer.py:429): row ranges that cross
    image boundaries, heights that are no power of two, every row width the kernels are built for; the same launch twice gives the
    same bits."""
    L = lib()
    n, h, w, up = case
    cin = 32 if up else 16
    g = torch.Generator().manual_seed(59)
    x = rounded(torch.randn(n, cin, h >> up, w >> up, generator=g), BF)
    wt = rounded(torch.randn(16, cin, 3, 3, generator=g) / 12.0, BF).requires_grad_()
    dy = rounded(torch.randn(n, 16, h, w, generator=g), BF)
    xin = F.interpolate(x, scale_factor=2, mode="nearest") if up else x
    F.conv2d(xin, wt, padding=1).backward(dy)
    ref = wt.grad.permute(0, 2, 3, 1)
    d = conv_desc(L, BF, n, h, w, cin, 16, 3, 1, 1, up0=up)
    xd, dyd = to_nhwc(x, BF), to_nhwc(dy, BF)
    outs = []
    for _ in range(2):
        ws_bytes = L.lib.vs_conv2d_wgrad_workspace(d)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        dw = torch.full((16, 3, 3, cin), float("nan"), device=DEV)
        L.check(L.lib.vs_conv2d_wgrad(d, _ptr(xd), None, _ptr(dyd), _ptr(dw), _ptr(ws), ws_bytes, None))
        sync()
        outs.append(dw)
    scale = ref.abs().max().item()
    assert torch.isfinite(outs[0]).all()
    assert torch.allclose(outs[0].cpu(), ref, rtol=1e-3, atol=1e-3 * scale), (outs[0].cpu() - ref).abs().max()
    assert torch.equal(outs[0], outs[1])


@pytest.mark.parametrize("case", [pytest.param((3, 2, 40, 128), id="3x2cls-40x128"), pytest.param((2, 4, 30, 256), id="2x4cls-30x256-two-k-steps"),
                                  pytest.param((1, 7, 6, 512), id="1x7cls-6x512"), pytest.param((32, 2, 256, 256), id="32x2cls-256x256-the-step's-launch")])
def test_head_backward_from_the_loss_gradient_planes(case):
    """head_dgrad_planes_kernel and the planes form of rows::conv_wgrad_rows16_kernel: the segmentation head's data and weight gradients
    read dLoss / dlogits as autograd hands it over (fp32 NCHW planes) - no 16-channel NHWC copy in between.  Against autograd of
    F.conv2d (loss.backward(), vol_seg_2d_trainer.py:429) with the gradient rounded to bf16 as the copy would have rounded it; 2 / 4 / 7
    classes (one and two K = 32 MFMA steps), ragged heights, every row width; the same launches twice give the same bits."""
    L = lib()
    n, k, h, w = case
    g = torch.Generator().manual_seed(67)
    x = rounded(torch.randn(n, 16, h, w, generator=g), BF).requires_grad_()
    wt = rounded(torch.randn(k, 16, 3, 3, generator=g) / 12.0, BF).requires_grad_()
    dl = torch.randn(n, k, h, w, generator=g) * 0.1
    F.conv2d(x, wt, padding=1).backward(rounded(dl, BF))
    ref_dx, ref_dw = x.grad, wt.grad.permute(0, 2, 3, 1)
    xd, wd, dld = to_nhwc(x.detach(), BF), w_krsc(wt.detach(), BF), dl.to(DEV).contiguous()
    ws_bytes = L.lib.vs_head_wgrad_planes_workspace(BF, n, k, h, w)
    assert ws_bytes > 0
    outs = []
    for _ in range(2):
        dx = torch.full((n, h, w, 16), float("nan"), device=DEV, dtype=torch.bfloat16)
        dw = torch.full((k, 3, 3, 16), float("nan"), device=DEV)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=DEV)
        L.check(L.lib.vs_head_dgrad_planes(BF, _ptr(dld), _ptr(wd), _ptr(dx), n, k, h, w, 16, None))
        L.check(L.lib.vs_head_wgrad_planes(BF, _ptr(xd), _ptr(dld), _ptr(dw), _ptr(ws), ws_bytes, n, k, h, w, None))
        sync()
        outs.append((dx, dw))
    dx, dw = outs[0]
    assert torch.isfinite(dx.float()).all() and torch.isfinite(dw).all()
    assert torch.allclose(from_nhwc(dx), ref_dx, **tol(BF, ref_dx.abs().max().item())), (from_nhwc(dx) - ref_dx).abs().max()
    assert torch.allclose(dw.cpu(), ref_dw, rtol=1e-3, atol=1e-3 * ref_dw.abs().max().item()), (dw.cpu() - ref_dw).abs().max()
    assert torch.equal(outs[1][0], dx) and torch.equal(outs[1][1], dw)
