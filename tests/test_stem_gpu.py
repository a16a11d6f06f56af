"""Stem tail (bn1 -> ReLU -> MaxPool2d(3,2,1), /root/reference/Quadtree_from scratch/models.py:224-226 via
torchvision resnet18) through the C ABI, forward and backward, against torch autograd on the CPU.

The backward never materialises d(loss)/d(relu output): BatchNorm sums come from the pooled side
(qt_stem_bn_bwd_sums), the data gradient from the gathering apply kernel (qt_stem_bn_bwd_apply);
the three-pass form (qt_stem_pool_bwd + qt_bn_bwd_reduce + qt_bn_bwd_apply) and the position-side
reduction (qt_stem_bn_bwd_reduce) must agree with it.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from _util import pkg, rel_err

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs the MI355X")
    return torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-5), (torch.bfloat16, 1.5e-2)])
def test_stem_pool_and_fused_backward(dt, tol):
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, C, H, P = 3, 64, 112, 56
    eps = 1e-5
    g = torch.Generator().manual_seed(21)
    y = (torch.randn(B, C, H, H, generator=g) * 1.5 + 0.2).to(dt).float()
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.3
    dp = torch.randn(B, C, P, P, generator=g).to(dt).float()

    # ---- reference: train-mode BatchNorm (batch statistics) -> ReLU -> max pool, autograd backward ----
    yr = y.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    out = F.max_pool2d(F.relu(F.batch_norm(yr, None, None, gr, br, True, 0.1, eps)), 3, 2, 1)
    out.backward(dp)
    mean = y.mean(dim=(0, 2, 3))
    var = y.var(dim=(0, 2, 3), unbiased=False)
    invstd = (var + eps).rsqrt()
    scale = gamma * invstd
    shift = beta - mean * scale

    f32 = dict(device=dev, dtype=torch.float32)
    yd = nhwc(y).to(dev, dt)
    dpd = nhwc(dp).to(dev, dt)
    sc, sh, mu, isd, gam = (t.to(**f32) for t in (scale, shift, mean, invstd, gamma))
    pooled = torch.empty(B, P, P, C, device=dev, dtype=dt)
    ymax = torch.empty_like(pooled)
    argmax = torch.empty(B, P, P, C, device=dev, dtype=torch.uint8)
    st = L.stream_ptr()
    qdt = L.qt_dtype(dt)
    L.check(lib.qt_stem_pool(qdt, L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(pooled), L.ptr(argmax), L.ptr(ymax), B, st),
            "qt_stem_pool")
    assert rel_err(pooled.float().cpu().permute(0, 3, 1, 2), out.detach()) <= (1e-6 if dt == torch.float32 else 4e-3)
    # y_at_max really is the conv1 output at the recorded tap
    am = argmax.cpu().long()
    ypad = F.pad(nhwc(y), (0, 0, 1, 1, 1, 1))  # [B][114][114][C]
    ph = torch.arange(P).view(1, P, 1, 1)
    pw = torch.arange(P).view(1, 1, P, 1)
    hh = (2 * ph + am // 3).expand(B, P, P, C)
    ww = (2 * pw + am % 3).expand(B, P, P, C)
    bb = torch.arange(B).view(B, 1, 1, 1).expand(B, P, P, C)
    cc = torch.arange(C).view(1, 1, 1, C).expand(B, P, P, C)
    assert torch.equal(ymax.float().cpu(), ypad[bb, hh, ww, cc])

    M = B * H * H

    def finalize(partial, rows):
        dgam = torch.empty(C, **f32)
        dbet = torch.empty(C, **f32)
        coef = torch.empty(3, C, **f32)
        L.check(lib.qt_bn_bwd_finalize(L.ptr(partial), rows, C, ctypes.c_longlong(M), L.ptr(gam), L.ptr(isd), L.ptr(dgam),
                                       L.ptr(dbet), 0, L.ptr(coef), st), "qt_bn_bwd_finalize")
        return dgam, dbet, coef

    # pooled-side sums
    rows = lib.qt_stem_bn_bwd_sums_rows(B)
    assert rows > 0
    part = torch.full((lib.qt_stats_capacity_rows(rows), 2, C), float("nan"), **f32)
    L.check(lib.qt_stem_bn_bwd_sums(qdt, L.ptr(dpd), L.ptr(ymax), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(isd), L.ptr(part),
                                    B, st), "qt_stem_bn_bwd_sums")
    dgam, dbet, coef = finalize(part, rows)
    # position-side sums (the older fused reduction) agree
    rows2 = lib.qt_stem_bn_bwd_rows(B)
    part2 = torch.full((lib.qt_stats_capacity_rows(rows2), 2, C), float("nan"), **f32)
    L.check(lib.qt_stem_bn_bwd_reduce(qdt, L.ptr(dpd), L.ptr(argmax), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mu),
                                      L.ptr(isd), L.ptr(part2), B, st), "qt_stem_bn_bwd_reduce")
    dgam2, dbet2, _ = finalize(part2, rows2)
    torch.cuda.synchronize()
    assert rel_err(dgam.cpu(), dgam2.cpu()) <= 1e-5 and rel_err(dbet.cpu(), dbet2.cpu()) <= 1e-5
    assert rel_err(dgam.cpu(), gr.grad) <= tol and rel_err(dbet.cpu(), br.grad) <= tol

    # gathering apply: d(loss)/d(conv1 output)
    dy = torch.empty(B, H, H, C, device=dev, dtype=dt)
    L.check(lib.qt_stem_bn_bwd_apply(qdt, L.ptr(dpd), L.ptr(argmax), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(isd),
                                     L.ptr(coef), L.ptr(dy), B, st), "qt_stem_bn_bwd_apply")
    # three-pass form
    gfull = torch.empty(B, H, H, C, device=dev, dtype=dt)
    L.check(lib.qt_stem_pool_bwd(qdt, L.ptr(dpd), L.ptr(argmax), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(gfull), B, st),
            "qt_stem_pool_bwd")
    dy3 = torch.empty_like(dy)
    L.check(lib.qt_bn_bwd_apply(qdt, L.ptr(gfull), None, L.ptr(yd), L.ptr(mu), L.ptr(isd), L.ptr(coef), L.ptr(dy3), None,
                                ctypes.c_longlong(M), C, st), "qt_bn_bwd_apply")
    torch.cuda.synchronize()
    ref = yr.grad
    assert rel_err(dy.float().cpu().permute(0, 3, 1, 2), ref) <= tol
    assert rel_err(dy3.float().cpu().permute(0, 3, 1, 2), ref) <= tol
    if dt == torch.float32:
        assert rel_err(dy.cpu(), dy3.cpu()) <= 1e-6
    # one-launch stem backward (bf16): the same d(loss)/d(conv1 output), computed tile by tile in LDS and contracted with the
    # packed input at once, against qt_stem_bn_bwd_apply + qt_conv2d_wgrad on the materialised map
    image = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    xpad = torch.empty(B, 230, 232, 4, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_input(qdt, L.ptr(image), L.ptr(xpad), B, st), "qt_pack_stem_input")
    d = L.ConvDesc()
    d.dtype = qdt; d.mode = L.QT_CONV_FWD; d.batch = B
    d.in_h, d.in_w, d.out_h, d.out_w = 230, 232, 112, 112
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = 32, 64, 7, 1, 2, 0
    d.src_pix_stride, d.src_row_stride, d.src_img_stride = 4, 232 * 4, 230 * 232 * 4
    dw_ref = torch.zeros(64, 7, 32, **f32)
    L.check(lib.qt_conv2d_wgrad(ctypes.byref(d), L.ptr(dy), L.ptr(xpad), L.ptr(dw_ref), st), "qt_conv2d_wgrad")
    dw = torch.zeros(64, 7, 32, **f32)
    rc = lib.qt_stem_bn_bwd_wgrad(qdt, L.ptr(dpd), L.ptr(argmax), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(mu), L.ptr(isd),
                                  L.ptr(coef), L.ptr(xpad), L.ptr(dw), B, st)
    torch.cuda.synchronize()
    if dt == torch.float32:
        assert rc != 0                                        # (the f32 build keeps the two-kernel form)
    else:
        assert rc == 0, L.last_error()
        assert rel_err(dw.cpu(), dw_ref.cpu()) <= 2e-3        # same bf16 gradient values, another summation order


def _stem_conv(L, dt, image, w, taps, want_stats, scale=None, shift=None, relu=0):
    """conv1 through the C ABI: pack input and weights, run the packed-stem descriptor."""
    dev = image.device
    lib = L.lib()
    B = image.shape[0]
    st = L.stream_ptr()
    qdt = L.qt_dtype(dt)
    xpad = torch.empty(B, 230, 232, 4, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_input(qdt, L.ptr(image), L.ptr(xpad), B, st), "qt_pack_stem_input")
    wp = torch.empty(64, taps, 32, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_weight(qdt, L.ptr(w), L.ptr(wp), taps, st), "qt_pack_stem_weight")
    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = qdt, L.QT_CONV_FWD, B
    d.in_h, d.in_w, d.out_h, d.out_w = 230, 232, 112, 112
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = 32, 64, taps, 1, 2, 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = 230 * 232 * 4, 232 * 4, 4
    d.relu = relu
    y = torch.empty(B, 112, 112, 64, device=dev, dtype=dt)
    stats = None
    if want_stats:
        rows = lib.qt_conv2d_stats_rows(ctypes.byref(d))
        stats = torch.full((rows, 2, 64), float("nan"), dtype=torch.float32, device=dev)
    io = L.ConvIO(L.ptr(xpad), L.ptr(wp), L.ptr(y), L.ptr(scale), L.ptr(shift), None, None, L.ptr(stats))
    L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), st), "qt_conv2d_igemm")
    torch.cuda.synchronize()
    return y, stats


@pytest.mark.parametrize("B", [1, 3, 20])   # 28, 84, 560 tiles: fewer / more tiles than workgroups
def test_stem_conv_dedicated_kernel(B):
    """csrc/conv_stem.hip (bf16) against F.conv2d and against the generic implicit GEMM."""
    dev = _dev()
    L = pkg("_lib")
    dt = torch.bfloat16
    g = torch.Generator().manual_seed(22 + B)
    image = torch.randn(B, 3, 224, 224, generator=g).to(dt).float()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(dt).float()
    ref = F.conv2d(image, w, None, 2, 3)                      # f32 math on bf16-representable operands
    scale = (torch.rand(64, generator=g) + 0.5).to(dev)
    shift = torch.randn(64, generator=g).to(dev)
    imd, wd = image.to(dev), w.to(dev)
    try:
        L.lib().qt_set_stem_conv(1)
        y, stats = _stem_conv(L, dt, imd, wd, 8, True)
        y_act, _ = _stem_conv(L, dt, imd, wd, 8, False, scale, shift, relu=1)
        L.lib().qt_set_stem_conv(0)
        y_gen, stats_gen = _stem_conv(L, dt, imd, wd, 8, True)
    finally:
        L.lib().qt_set_stem_conv(-1)
    got = y.float().cpu().permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= 4e-3                           # one bf16 rounding of the output
    assert torch.equal(y, y_gen) or rel_err(y.float().cpu(), y_gen.float().cpu()) <= 4e-3
    # statistics come from the f32 accumulators: sum and sum of squares per channel
    s = stats.sum(dim=0).cpu().double()
    sg = stats_gen.sum(dim=0).cpu().double()
    assert rel_err(s[0], ref.double().sum(dim=(0, 2, 3))) <= 1e-4
    assert rel_err(s[1], (ref.double() ** 2).sum(dim=(0, 2, 3))) <= 1e-5
    assert rel_err(s, sg) <= 1e-4
    act = F.relu(ref * scale.cpu().view(1, -1, 1, 1) + shift.cpu().view(1, -1, 1, 1))
    assert rel_err(y_act.float().cpu().permute(0, 3, 1, 2), act) <= 4e-3


@pytest.mark.parametrize("B", [1, 3, 11])   # 28 / 84 / 308 tiles: fewer and more tiles than workgroups
def test_stem_conv_pool_fused_eval_kernel(B):
    """qt_stem_conv_pool (conv1 + folded BatchNorm + ReLU + MaxPool2d(3,2,1), eval forward) against torch on the CPU and
    against the two-kernel form (qt_conv2d_igemm with the affine epilogue + qt_stem_pool)."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    dt = torch.bfloat16
    qdt = L.qt_dtype(dt)
    g = torch.Generator().manual_seed(120 + B)
    image = torch.randn(B, 3, 224, 224, generator=g).to(dt).float()
    w = (torch.randn(64, 3, 7, 7, generator=g) * 0.1).to(dt).float()
    scale = torch.rand(64, generator=g) + 0.5
    shift = torch.randn(64, generator=g) * 0.5
    ref = F.max_pool2d(F.relu(F.conv2d(image, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), 3, 2, 1)
    st = L.stream_ptr()
    imd, wd, sc, sh = image.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    xpad = torch.empty(B, 230, 232, 4, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_input(qdt, L.ptr(imd), L.ptr(xpad), B, st), "qt_pack_stem_input")
    wp = torch.empty(64, 8, 32, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_weight(qdt, L.ptr(wd), L.ptr(wp), 8, st), "qt_pack_stem_weight")
    pooled = torch.full((B, 56, 56, 64), float("nan"), device=dev, dtype=dt)
    L.check(lib.qt_stem_conv_pool(qdt, L.ptr(xpad), L.ptr(wp), 8, L.ptr(sc), L.ptr(sh), L.ptr(pooled), B, st), "qt_stem_conv_pool")
    # two-kernel form
    y, _ = _stem_conv(L, dt, imd, wd, 8, False, sc, sh, relu=1)
    ones, zeros = torch.ones(64, device=dev), torch.zeros(64, device=dev)
    pooled2 = torch.empty(B, 56, 56, 64, device=dev, dtype=dt)
    L.check(lib.qt_stem_pool(qdt, L.ptr(y), L.ptr(ones), L.ptr(zeros), L.ptr(pooled2), None, None, B, st), "qt_stem_pool")
    # the form that reads the f32 NCHW image itself (no packed copy): images that are not bf16-exact, so that the in-kernel
    # rounding is exercised, against qt_pack_stem_input + qt_stem_conv_pool on the same images
    image2 = torch.randn(B, 3, 224, 224, generator=g).to(dev)
    xpad2 = torch.empty(B, 230, 232, 4, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_input(qdt, L.ptr(image2), L.ptr(xpad2), B, st), "qt_pack_stem_input")
    pooled3 = torch.full((B, 56, 56, 64), float("nan"), device=dev, dtype=dt)
    L.check(lib.qt_stem_conv_pool(qdt, L.ptr(xpad2), L.ptr(wp), 8, L.ptr(sc), L.ptr(sh), L.ptr(pooled3), B, st), "qt_stem_conv_pool")
    pooled4 = torch.full((B, 56, 56, 64), float("nan"), device=dev, dtype=dt)
    L.check(lib.qt_stem_conv_pool_nchw(qdt, L.ptr(image2), L.ptr(wp), 8, L.ptr(sc), L.ptr(sh), L.ptr(pooled4), B, st),
            "qt_stem_conv_pool_nchw")
    torch.cuda.synchronize()
    assert torch.equal(pooled, pooled2)                       # same MFMA order, same rounding: bit-identical
    assert rel_err(pooled.float().cpu().permute(0, 3, 1, 2), ref) <= 4e-3
    assert not torch.isnan(pooled4.float()).any()
    assert torch.equal(pooled3, pooled4)


def test_stem_conv_pool_negative_zero_is_not_a_maximum():
    """The fused kernel pools the ReLU tile with a packed SIGNED 16-bit maximum on the raw bf16 words: a -0 (zero input,
    negative scale, zero shift give -0 conv outputs) must not beat a +0 or a small positive value."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    dt = torch.bfloat16
    qdt = L.qt_dtype(dt)
    B = 1
    image = torch.zeros(B, 3, 224, 224)
    image[0, :, 100:110, 100:110] = 1.0                       # a patch of positive responses in a field of -0
    w = torch.full((64, 3, 7, 7), 0.01).to(dt).float()
    scale = torch.full((64,), -1.0)
    scale[::2] = 1.0
    shift = torch.zeros(64)
    ref = F.max_pool2d(F.relu(F.conv2d(image, w, None, 2, 3) * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)), 3, 2, 1)
    st = L.stream_ptr()
    imd, wd, sc, sh = image.to(dev), w.to(dev), scale.to(dev), shift.to(dev)
    xpad = torch.empty(B, 230, 232, 4, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_input(qdt, L.ptr(imd), L.ptr(xpad), B, st), "qt_pack_stem_input")
    wp = torch.empty(64, 8, 32, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_weight(qdt, L.ptr(wd), L.ptr(wp), 8, st), "qt_pack_stem_weight")
    pooled = torch.full((B, 56, 56, 64), float("nan"), device=dev, dtype=dt)
    L.check(lib.qt_stem_conv_pool(qdt, L.ptr(xpad), L.ptr(wp), 8, L.ptr(sc), L.ptr(sh), L.ptr(pooled), B, st), "qt_stem_conv_pool")
    torch.cuda.synchronize()
    got = pooled.float().cpu().permute(0, 3, 1, 2)
    assert torch.isfinite(got).all() and (got >= 0).all()
    assert (got[:, 1::2] == 0).all()                           # negative scale: relu(-x) = 0 everywhere, never a stray -0 win
    assert rel_err(got, ref) <= 4e-3
    assert (pooled.view(torch.int16) >= 0).all()               # no sign bit in the result: +0, not -0


@pytest.mark.parametrize("B", [1, 3, 9])   # 56 / 168 / 504 tiles: fewer and more tiles than workgroups
def test_stem_weight_gradient_from_raw_rows(B):
    """conv1's weight gradient (conv2d backward-weight inside loss.backward(), /root/reference/Quadtree_from
    scratch/Quadtree_train.py:65, layer built at models.py:222-223) through qt_conv2d_wgrad on the packed stem descriptor
    (bf16: csrc/conv_wgrad.hip::stem_wgrad_rows_kernel) against torch on the CPU."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    dt = torch.bfloat16
    qdt = L.qt_dtype(dt)
    g = torch.Generator().manual_seed(300 + B)
    image = torch.randn(B, 3, 224, 224, generator=g).to(dt).float()
    dy = (torch.randn(B, 64, 112, 112, generator=g) * 0.1).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(image, (64, 3, 7, 7), dy, stride=2, padding=3)
    st = L.stream_ptr()
    imd = image.to(dev)
    xpad = torch.empty(B, 230, 232, 4, device=dev, dtype=dt)
    L.check(lib.qt_pack_stem_input(qdt, L.ptr(imd), L.ptr(xpad), B, st), "qt_pack_stem_input")
    dyd = nhwc(dy).to(dev, dt).contiguous()
    d = L.ConvDesc()
    d.dtype = qdt; d.mode = L.QT_CONV_FWD; d.batch = B
    d.in_h, d.in_w, d.out_h, d.out_w = 230, 232, 112, 112
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = 32, 64, 7, 1, 2, 0
    d.src_pix_stride, d.src_row_stride, d.src_img_stride = 4, 232 * 4, 230 * 232 * 4
    dw = torch.zeros(64, 7, 32, device=dev)
    L.check(lib.qt_conv2d_wgrad(ctypes.byref(d), L.ptr(dyd), L.ptr(xpad), L.ptr(dw), st), "qt_conv2d_wgrad")
    grad = torch.empty(64, 3, 7, 7, device=dev)
    L.check(lib.qt_unpack_stem_wgrad(L.ptr(dw), L.ptr(grad), 0, st), "qt_unpack_stem_wgrad")
    torch.cuda.synchronize()
    assert rel_err(grad.cpu(), ref) <= 2e-3
    # the pad columns of the packed layout (kw = 7, channel 3) receive no gradient from real data: channel 3 of the packed
    # input is zero
    assert float(dw.view(64, 7, 8, 4)[..., 3].abs().max()) == 0.0
