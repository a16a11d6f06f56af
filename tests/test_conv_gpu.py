"""GPU parity of the implicit-GEMM conv kernel (through the C ABI) against torch CPU fp32.

Tolerances: f32 build 2e-5 of max|ref| (exact-f32 MFMA, different summation
order); bf16 build 1.5e-2 (inputs are pre-rounded to bf16 on both sides, so the
difference is the bf16 rounding of the output only).
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from _util import pkg, rel_err

pytestmark = pytest.mark.gpu

TOL = {torch.float32: 2e-5, torch.bfloat16: 1.5e-2}


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def run_conv(L, dt, x_nhwc, w_krsc, B, in_hw, out_hw, k_per_tap, n_out, kh, kw, stride, pad,
             mode, quad=0, relu=0, scale=None, shift=None, residual=None, relu_mask=None,
             want_stats=False, strides=None, m_rows=None):
    dev = x_nhwc.device
    d = L.ConvDesc()
    d.dtype = L.qt_dtype(dt)
    d.mode = mode
    d.batch = B
    d.in_h, d.in_w = in_hw
    d.out_h, d.out_w = out_hw
    d.k_per_tap, d.n_out = k_per_tap, n_out
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    if strides is None:
        c = x_nhwc.shape[-1]
        strides = (x_nhwc.shape[1] * x_nhwc.shape[2] * c, x_nhwc.shape[2] * c, c)
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = strides
    d.quad, d.relu = quad, relu
    if m_rows is None:
        m_rows = B * (4 if (quad and mode == L.QT_CONV_FWD) else 1) * out_hw[0] * out_hw[1]
    y = torch.empty(m_rows, n_out, dtype=dt, device=dev)
    stats = None
    if want_stats:
        rows = L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
        stats = torch.zeros(rows, 2, n_out, dtype=torch.float32, device=dev)
    io = L.ConvIO(L.ptr(x_nhwc), L.ptr(w_krsc), L.ptr(y), L.ptr(scale), L.ptr(shift),
                  L.ptr(residual), L.ptr(relu_mask), L.ptr(stats))
    L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
    torch.cuda.synchronize()
    return y, stats


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    # B, Cin, Cout, H, k, stride, pad
    (2, 64, 64, 56, 3, 1, 1),    # patch kernel (conv_patch.hip), BN=64
    (3, 128, 128, 28, 3, 1, 1),  # patch kernel, BN=128, two channel chunks
    (1, 64, 128, 56, 3, 1, 1),   # patch kernel, widest LDS footprint
    (5, 128, 64, 28, 3, 1, 1),   # patch kernel, ragged last tile
    (3, 64, 128, 56, 3, 2, 1),
    (2, 64, 128, 56, 1, 2, 0),
    (1, 256, 512, 14, 3, 2, 1),
    (1, 512, 512, 7, 3, 1, 1),   # M = 49: ragged tile
    (130, 128, 128, 28, 3, 1, 1),  # >= 100 k pixels, N = 128: single-buffered tiles, half-tile epilogue, ragged last tile
])
def test_conv_fwd_epilogue(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, k, s, p = cfg
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dt).float()
    Ho = (H + 2 * p - k) // s + 1
    res = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    raw = F.conv2d(x, w, None, s, p)
    ref = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)

    xd = nhwc(x).to(dev, dt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    # raw + stats
    y, stats = run_conv(L, dt, xd, wd, B, (H, H), (Ho, Ho), Cin, Cout, k, k, s, p, L.QT_CONV_FWD,
                        want_stats=True)
    got = y.float().cpu().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, raw) <= TOL[dt]
    ssum = stats.sum(0).cpu()
    assert rel_err(ssum[0], raw.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(ssum[1], (raw * raw).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    # fused epilogue
    y, _ = run_conv(L, dt, xd, wd, B, (H, H), (Ho, Ho), Cin, Cout, k, k, s, p, L.QT_CONV_FWD,
                    relu=1, scale=scale.to(dev), shift=shift.to(dev),
                    residual=nhwc(res).to(dev, dt).view(-1, Cout))
    got = y.float().cpu().view(B, Ho, Ho, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 56, 3, 1, 1),    # patch kernel
    (3, 128, 128, 28, 3, 1, 1),  # patch kernel
    (2, 64, 128, 56, 3, 2, 1),
    (2, 64, 128, 56, 1, 2, 0),
    (1, 256, 512, 14, 3, 2, 1),
    (129, 128, 128, 28, 3, 1, 1),  # >= 100 k pixels, 128 channels: single-buffered tiles (bf16)
])
def test_conv_dgrad(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, k, s, p = cfg
    g = torch.Generator().manual_seed(2)
    Ho = (H + 2 * p - k) // s + 1
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cout * k * k)) ** 0.5).to(dt).float()
    dy = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    other = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    act = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    dx = torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, s, p)
    ref = (dx + other) * (act > 0)

    dyd = nhwc(dy).to(dev, dt)
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, dt)  # [Cin][kh][kw][Cout]
    y, _ = run_conv(L, dt, dyd, wt, B, (Ho, Ho), (H, H), Cout, Cin, k, k, s, p, L.QT_CONV_DGRAD,
                    residual=nhwc(other).to(dev, dt).view(-1, Cin),
                    relu_mask=nhwc(act).to(dev, dt).view(-1, Cin))
    got = y.float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_quadrant_conv_fwd_and_dgrad(dt):
    """Quadrant split with zero halo at the seam
    (/root/reference/Quadtree_from scratch/models.py:277-287)."""
    dev = _dev()
    L = pkg("_lib")
    B, C, N = 3, 256, 128
    g = torch.Generator().manual_seed(3)
    base = torch.randn(B, C, 14, 14, generator=g).to(dt).float()
    w = (torch.randn(N, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).to(dt).float()
    bias = torch.randn(N, generator=g) * 0.1
    quads = [base[:, :, :7, :7], base[:, :, :7, 7:], base[:, :, 7:, :7], base[:, :, 7:, 7:]]
    ref = torch.stack([F.relu(F.conv2d(q, w, bias, 1, 1)) for q in quads], 1)  # [B,4,N,7,7]
    xd = nhwc(base).to(dev, dt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    y, _ = run_conv(L, dt, xd, wd, B, (7, 7), (7, 7), C, N, 3, 3, 1, 1, L.QT_CONV_FWD, quad=1, relu=1,
                    shift=bias.to(dev), strides=(14 * 14 * C, 14 * C, C))
    got = y.float().cpu().view(B, 4, 7, 7, N).permute(0, 1, 4, 2, 3)
    assert rel_err(got, ref) <= TOL[dt]

    # dgrad: per-quadrant gradient images -> un-split map
    dyq = torch.randn(B, 4, N, 7, 7, generator=g).to(dt).float()
    dbase = torch.zeros(B, C, 14, 14)
    sl = [(slice(0, 7), slice(0, 7)), (slice(0, 7), slice(7, 14)), (slice(7, 14), slice(0, 7)), (slice(7, 14), slice(7, 14))]
    for q in range(4):
        dbase[:, :, sl[q][0], sl[q][1]] = torch.nn.grad.conv2d_input((B, C, 7, 7), w, dyq[:, q], 1, 1)
    dyd = dyq.permute(0, 1, 3, 4, 2).contiguous().to(dev, dt)  # [B,4,7,7,N]
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, dt)
    y, _ = run_conv(L, dt, dyd, wt, B, (7, 7), (14, 14), N, C, 3, 3, 1, 1, L.QT_CONV_DGRAD, quad=1,
                    strides=(49 * N, 7 * N, N))
    got = y.float().cpu().view(B, 14, 14, C).permute(0, 3, 1, 2)
    assert rel_err(got, dbase) <= TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_patch_kernel_matches_generic_path(dt):
    """The layer-1 ring kernel (conv_patch.hip) behind the same entry point, next to shapes it leaves to the other kernels."""
    _dev()
    L = pkg("_lib")
    lib = L.lib()
    lib.qt_set_patch_conv(1)
    try:
        # (26, 64, 64, 56): 342 tiles > 256 workgroups, so the persistent ring kernel (bf16) walks
        # several tiles per workgroup and its sliding window wraps
        for cfg in [(2, 64, 64, 56, 3, 1, 1), (26, 64, 64, 56, 3, 1, 1), (3, 128, 128, 28, 3, 1, 1),
                    (1, 64, 128, 56, 3, 1, 1), (5, 128, 64, 28, 3, 1, 1)]:
            test_conv_fwd_epilogue(dt, cfg)
        for cfg in [(2, 64, 64, 56, 3, 1, 1), (26, 64, 64, 56, 3, 1, 1), (3, 128, 128, 28, 3, 1, 1)]:
            test_conv_dgrad(dt, cfg)
    finally:
        lib.qt_set_patch_conv(2)


def test_conv_rejects_bad_args():
    _dev()
    L = pkg("_lib")
    d = L.ConvDesc()
    d.dtype = 7
    io = L.ConvIO()
    st = L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), None)
    assert st == -1 and b"dtype" in L.lib().qt_last_error()


def run_wgrad(L, dt, dy_nhwc, x_nhwc, B, in_hw, out_hw, k_per_tap, n_out, kh, kw, stride, pad, quad=0, strides=None,
              workspace=False):
    dev = x_nhwc.device
    d = L.ConvDesc()
    d.dtype = L.qt_dtype(dt)
    d.mode = L.QT_CONV_FWD
    d.batch = B
    d.in_h, d.in_w = in_hw
    d.out_h, d.out_w = out_hw
    d.k_per_tap, d.n_out = k_per_tap, n_out
    d.kh, d.kw, d.stride, d.pad = kh, kw, stride, pad
    if strides is None:
        c = x_nhwc.shape[-1]
        strides = (x_nhwc.shape[1] * x_nhwc.shape[2] * c, x_nhwc.shape[2] * c, c)
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = strides
    d.quad = quad
    dw = torch.zeros(n_out, kh * kw, k_per_tap, dtype=torch.float32, device=dev)
    if workspace:
        L.lib().qt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
        nbytes = L.lib().qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        assert nbytes > 0, "this shape was expected to use the partial-filter workspace"
        ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=dev)  # must be fully overwritten
        L.check(L.lib().qt_conv2d_wgrad_ws(ctypes.byref(d), L.ptr(dy_nhwc), L.ptr(x_nhwc), L.ptr(dw), L.ptr(ws),
                                           ctypes.c_size_t(nbytes), L.stream_ptr()), "qt_conv2d_wgrad_ws")
    else:
        L.check(L.lib().qt_conv2d_wgrad(ctypes.byref(d), L.ptr(dy_nhwc), L.ptr(x_nhwc), L.ptr(dw), L.stream_ptr()),
                "qt_conv2d_wgrad")
    torch.cuda.synchronize()
    return dw


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [
    (2, 64, 64, 56, 3, 1, 1),
    (3, 64, 128, 56, 3, 2, 1),
    (2, 64, 128, 56, 1, 2, 0),
    (2, 128, 128, 28, 3, 1, 1),
    (1, 256, 512, 14, 3, 2, 1),
    (5, 512, 512, 7, 3, 1, 1),
    (37, 5376, 2688, 1, 1, 1, 0),   # classifier.0 as a 1x1 conv on 1x1 images, ragged batch
])
def test_conv_wgrad(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, k, s, p = cfg
    g = torch.Generator().manual_seed(4)
    Ho = (H + 2 * p - k) // s + 1
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    dy = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, s, p)
    dw = run_wgrad(L, dt, nhwc(dy).to(dev, dt), nhwc(x).to(dev, dt), B, (H, H), (Ho, Ho), Cin, Cout, k, k, s, p)
    got = dw.cpu().view(Cout, k, k, Cin).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= 3e-5  # f32 accumulation of exactly representable products


@pytest.mark.parametrize("cfg", [
    (3, 64, 64, 56),      # layer1
    (5, 128, 128, 28),    # layer2: four channel tiles share a range of positions
    (9, 64, 128, 28),     # ragged batch, rectangular channel tiles
    (2, 256, 256, 14),    # layer3: 16 tiles, 30 % padding overhead
    (300, 64, 64, 14),    # more ranges than workgroups' worth of images: ranges cut through images
    (6, 512, 512, 7),     # layer4: 9x9 padded positions, a 32-position step spans three padded rows
])
@pytest.mark.parametrize("variant", [3, 0, 2])
def test_conv_wgrad_streaming_kernel(cfg, variant):
    """3x3 / stride 1 weight gradient with all nine taps accumulated by one workgroup
    (csrc/conv_wgrad_patch.hip: tile-resident kernel = 3, ring kernel with one / two wave groups = 0 / 2) against
    torch.nn.grad.conv2d_weight and against the generic kernel."""
    dev = _dev()
    L = pkg("_lib")
    L.lib().qt_set_wgrad_patch_variant(variant)
    dt = torch.bfloat16
    B, Cin, Cout, H = cfg
    g = torch.Generator().manual_seed(14)
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    dy = torch.randn(B, Cout, H, H, generator=g).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, 1, 1)
    xd, dyd = nhwc(x).to(dev, dt), nhwc(dy).to(dev, dt)
    try:
        L.lib().qt_set_wgrad_patch_min_width(7)
        dw = run_wgrad(L, dt, dyd, xd, B, (H, H), (H, H), Cin, Cout, 3, 3, 1, 1)
        dw_ws = run_wgrad(L, dt, dyd, xd, B, (H, H), (H, H), Cin, Cout, 3, 3, 1, 1, workspace=True)
        dw_ws2 = run_wgrad(L, dt, dyd, xd, B, (H, H), (H, H), Cin, Cout, 3, 3, 1, 1, workspace=True)
        L.lib().qt_set_wgrad_patch_min_width(0)
        dw_generic = run_wgrad(L, dt, dyd, xd, B, (H, H), (H, H), Cin, Cout, 3, 3, 1, 1)
    finally:
        L.lib().qt_set_wgrad_patch_min_width(-1)
    # .grad written directly in OIHW by the partial-filter sum (NaN-filled destination: fully overwritten)
    lib = L.lib()
    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
    d.in_h = d.in_w = d.out_h = d.out_w = H
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = Cin, Cout, 3, 3, 1, 1
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * Cin, H * Cin, Cin
    lib.qt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    g_oihw = torch.full((Cout, Cin, 3, 3), float("nan"), dtype=torch.float32, device=dev)
    try:
        lib.qt_set_wgrad_patch_min_width(7)
        nbytes = lib.qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        wsb = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        L.check(lib.qt_conv2d_wgrad_oihw(ctypes.byref(d), L.ptr(dyd), L.ptr(xd), L.ptr(g_oihw), L.ptr(wsb),
                                         ctypes.c_size_t(nbytes), L.stream_ptr()), "qt_conv2d_wgrad_oihw")
    finally:
        lib.qt_set_wgrad_patch_min_width(-1)
        lib.qt_set_wgrad_patch_variant(-1)
    torch.cuda.synchronize()
    assert rel_err(g_oihw.cpu(), ref) <= 3e-5
    assert torch.equal(g_oihw, dw_ws.view(Cout, 3, 3, Cin).permute(0, 3, 1, 2))
    for name, t in (("atomic", dw), ("workspace", dw_ws)):
        got = t.cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        assert rel_err(got, ref) <= 3e-5, name
    assert rel_err(dw.cpu(), dw_generic.cpu()) <= 3e-5
    assert torch.equal(dw_ws, dw_ws2)  # partial filters are summed in a fixed order: bit-reproducible


@pytest.mark.parametrize("cfg", [
    (7, 64, 64, 12, 20),    # rectangular maps, ranges end inside a tile
    (3, 128, 64, 9, 31),    # odd width: a DMA unit of 8 rows straddles padded rows and images
    (1, 64, 64, 7, 7),      # one image: fewer positions than one tile, most workgroups exit at once
    (33, 64, 128, 30, 8),
])
def test_conv_wgrad_tile_kernel_shapes(cfg):
    """Tile-resident weight-gradient kernel on map sizes the models do not use (offset table + 8-row DMA units)."""
    dev = _dev()
    L = pkg("_lib")
    dt = torch.bfloat16
    B, Cin, Cout, H, W = cfg
    g = torch.Generator().manual_seed(16)
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).float()
    dy = torch.randn(B, Cout, H, W, generator=g).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, 3, 3), dy, 1, 1)
    xd, dyd = nhwc(x).to(dev, dt), nhwc(dy).to(dev, dt)
    try:
        L.lib().qt_set_wgrad_patch_min_width(7)
        L.lib().qt_set_wgrad_patch_variant(3)
        dw = run_wgrad(L, dt, dyd, xd, B, (H, W), (H, W), Cin, Cout, 3, 3, 1, 1)
        dw_ws = run_wgrad(L, dt, dyd, xd, B, (H, W), (H, W), Cin, Cout, 3, 3, 1, 1, workspace=True)
    finally:
        L.lib().qt_set_wgrad_patch_min_width(-1)
        L.lib().qt_set_wgrad_patch_variant(-1)
    for name, t in (("atomic", dw), ("workspace", dw_ws)):
        got = t.cpu().view(Cout, 3, 3, Cin).permute(0, 3, 1, 2)
        assert rel_err(got, ref) <= 3e-5, name


@pytest.mark.parametrize("cfg", [
    (3, 64, 128, 56, 56, 3),     # layer2.0.conv1
    (2, 64, 128, 56, 56, 1),     # layer2.0.downsample.0
    (5, 128, 256, 28, 28, 3),    # layer3.0.conv1: 8 channel tiles share a range of positions
    (5, 128, 256, 28, 28, 1),
    (7, 256, 512, 14, 14, 3),    # layer4.0.conv1: 8x8 padded positions, a tile spans an image
    (7, 256, 512, 14, 14, 1),
    (33, 64, 64, 24, 24, 3),     # ragged batch against the ranges
    (4, 64, 128, 20, 12, 3),     # rectangular map: row pitch != column count
    (300, 64, 64, 8, 8, 3),      # more images than ranges: ranges cut through images, 5x5 padded positions per image
    (1, 64, 64, 16, 16, 1),      # one image: most workgroups exit at once
])
def test_conv_wgrad_stride2_on_parity_planes(cfg):
    """conv_wgrad_s2.hip: weight gradient of the stride-2 convolutions of a transition block (3x3 / 2 pad 1 and the 1x1 / 2
    downsample) against torch.nn.grad.conv2d_weight; written to OIHW through NaN-filled scratch, bit-reproducible, and in
    agreement with the generic kernel it replaces."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    dt = torch.bfloat16
    B, Cin, Cout, H, W, k = cfg
    pad = 1 if k == 3 else 0
    Ho, Wo = H // 2, W // 2
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, W, generator=g).to(dt).float()
    dy = torch.randn(B, Cout, Ho, Wo, generator=g).to(dt).float()
    ref = torch.nn.grad.conv2d_weight(x, (Cout, Cin, k, k), dy, 2, pad)
    xd, dyd = nhwc(x).to(dev, dt), nhwc(dy).to(dev, dt)
    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
    d.in_h, d.in_w, d.out_h, d.out_w = H, W, Ho, Wo
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = Cin, Cout, k, k, 2, pad
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * W * Cin, W * Cin, Cin
    lib.qt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    nbytes = lib.qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
    assert nbytes > 0
    outs = []
    for _ in range(2):
        wsb = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=dev)
        g_oihw = torch.full((Cout, Cin, k, k), float("nan"), dtype=torch.float32, device=dev)
        L.check(lib.qt_conv2d_wgrad_oihw(ctypes.byref(d), L.ptr(dyd), L.ptr(xd), L.ptr(g_oihw), L.ptr(wsb),
                                         ctypes.c_size_t(nbytes), L.stream_ptr()), "qt_conv2d_wgrad_oihw")
        torch.cuda.synchronize()
        outs.append(g_oihw.cpu())
    assert rel_err(outs[0], ref) <= 3e-5
    assert torch.equal(outs[0], outs[1])
    try:   # the generic kernel behind the same descriptor
        lib.qt_set_wgrad_s2(0)
        assert lib.qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d)) == 0
        dw = run_wgrad(L, dt, dyd, xd, B, (H, W), (Ho, Wo), Cin, Cout, k, k, 2, pad)
    finally:
        lib.qt_set_wgrad_s2(-1)
    assert rel_err(dw.cpu().view(Cout, k, k, Cin).permute(0, 3, 1, 2), outs[0]) <= 3e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(256, 2688, 5376), (37, 2688, 5376), (64, 128, 192), (4096, 128, 64)])
def test_linear_wgrad_is_written_not_accumulated(dt, cfg):
    """qt_linear_wgrad: nn.Linear backward-weight into a NaN-filled destination (plain stores where a tile has one row range --
    classifier.0 at batch 256 -- zero + atomics for the long-row case), against dy^T x in f64."""
    dev = _dev()
    L = pkg("_lib")
    rows, out, inn = cfg
    g = torch.Generator().manual_seed(23)
    dy = torch.randn(rows, out, generator=g).to(dt).float()
    x = torch.randn(rows, inn, generator=g).to(dt).float()
    ref = (dy.double().t() @ x.double()).float()
    dw = torch.full((out, inn), float("nan"), dtype=torch.float32, device=dev)
    L.lib().qt_linear_wgrad.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    dyd, xd = dy.to(dev, dt), x.to(dev, dt)
    for _ in range(2):   # (a second call over the first result: nothing is accumulated)
        L.check(L.lib().qt_linear_wgrad(L.qt_dtype(dt), L.ptr(dyd), L.ptr(xd), L.ptr(dw), rows, out, inn,
                                        L.stream_ptr()), "qt_linear_wgrad")
    torch.cuda.synchronize()
    assert rel_err(dw.cpu(), ref) <= 3e-5


def test_conv_wgrad_streaming_kernel_channel_slice():
    """X is a 64-channel slice of a wider NHWC tensor (explicit strides), as the plan's views are."""
    dev = _dev()
    L = pkg("_lib")
    dt = torch.bfloat16
    B, Cw, C, N, H = 2, 192, 64, 64, 56
    g = torch.Generator().manual_seed(15)
    wide = torch.randn(B, H, H, Cw, generator=g).to(dt)
    dy = torch.randn(B, N, H, H, generator=g).to(dt).float()
    xs = wide[..., 64:128].float().permute(0, 3, 1, 2).contiguous()
    ref = torch.nn.grad.conv2d_weight(xs, (N, C, 3, 3), dy, 1, 1)
    wd = wide.to(dev)
    try:
        L.lib().qt_set_wgrad_patch_min_width(14)
        dw = run_wgrad(L, dt, nhwc(dy).to(dev, dt), wd.view(-1)[64:], B, (H, H), (H, H), C, N, 3, 3, 1, 1,
                       strides=(H * H * Cw, H * Cw, Cw))
    finally:
        L.lib().qt_set_wgrad_patch_min_width(-1)
    got = dw.cpu().view(N, 3, 3, C).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= 3e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_quadrant_wgrad(dt):
    dev = _dev()
    L = pkg("_lib")
    B, C, N = 3, 256, 128
    g = torch.Generator().manual_seed(5)
    base = torch.randn(B, C, 14, 14, generator=g).to(dt).float()
    dyq = torch.randn(B, 4, N, 7, 7, generator=g).to(dt).float()
    quads = [base[:, :, :7, :7], base[:, :, :7, 7:], base[:, :, 7:, :7], base[:, :, 7:, 7:]]
    ref = sum(torch.nn.grad.conv2d_weight(quads[q].contiguous(), (N, C, 3, 3), dyq[:, q].contiguous(), 1, 1)
              for q in range(4))
    dyd = dyq.permute(0, 1, 3, 4, 2).contiguous().to(dev, dt)
    dw = run_wgrad(L, dt, dyd, nhwc(base).to(dev, dt), B, (7, 7), (7, 7), C, N, 3, 3, 1, 1, quad=1,
                   strides=(14 * 14 * C, 14 * C, C))
    got = dw.cpu().view(N, 3, 3, C).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= 3e-5


@pytest.mark.parametrize("cfg", [(3, 256, 128, 7, 2), (37, 256, 128, 7, 2), (5, 128, 128, 14, 2), (9, 128, 64, 7, 4)])
def test_region_wgrad_on_the_tile_kernel(cfg):
    """The region heads' weight gradient (quadrant conv of QuadtreeCNN, Quadtree_from scratch/models.py:234-238,284-287;
    quadrant / sub-quadrant convs of AttentionHierarchicalCNN, :62-78) on the tile-resident streaming kernel: every region sits
    behind its own pad row and column on the padded grid, so the zero halo at the seams is the ordinary padding.  Written to
    OIHW through NaN-filled scratch, bit-reproducible, against per-region torch.nn.grad.conv2d_weight and the generic kernel."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    dt = torch.bfloat16
    B, C, N, R, S = cfg
    H = S * R
    g = torch.Generator().manual_seed(29)
    base = torch.randn(B, C, H, H, generator=g).to(dt).float()
    dyq = torch.randn(B, S * S, N, R, R, generator=g).to(dt).float()
    ref = sum(torch.nn.grad.conv2d_weight(base[:, :, (q // S) * R:(q // S + 1) * R, (q % S) * R:(q % S + 1) * R].contiguous(),
                                          (N, C, 3, 3), dyq[:, q].contiguous(), 1, 1) for q in range(S * S))
    dyd = dyq.permute(0, 1, 3, 4, 2).contiguous().to(dev, dt)
    xd = nhwc(base).to(dev, dt)
    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
    d.in_h = d.in_w = d.out_h = d.out_w = R
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = C, N, 3, 3, 1, 1
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * C, H * C, C
    d.quad = S
    lib.qt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
    nbytes = lib.qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
    assert nbytes > 0
    outs = []
    for _ in range(2):
        wsb = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=dev)
        g_oihw = torch.full((N, C, 3, 3), float("nan"), dtype=torch.float32, device=dev)
        L.check(lib.qt_conv2d_wgrad_oihw(ctypes.byref(d), L.ptr(dyd), L.ptr(xd), L.ptr(g_oihw), L.ptr(wsb),
                                         ctypes.c_size_t(nbytes), L.stream_ptr()), "qt_conv2d_wgrad_oihw")
        torch.cuda.synchronize()
        outs.append(g_oihw.cpu())
    assert rel_err(outs[0], ref) <= 3e-5
    assert torch.equal(outs[0], outs[1])
    try:   # the generic kernel behind the same descriptor
        lib.qt_set_wgrad_patch_min_width(0)
        dw = run_wgrad(L, dt, dyd, xd, B, (R, R), (R, R), C, N, 3, 3, 1, 1, quad=S, strides=(H * H * C, H * C, C))
    finally:
        lib.qt_set_wgrad_patch_min_width(-1)
    assert rel_err(dw.cpu().view(N, 3, 3, C).permute(0, 3, 1, 2), outs[0]) <= 3e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 64, 128, 56), (3, 128, 256, 28), (5, 256, 512, 14), (1, 64, 64, 10),
                                 # >= 16 images: the four-tap instantiation of the patch-resident kernel (csrc/conv_pt.hip)
                                 (16, 64, 128, 56), (18, 128, 256, 28), (20, 256, 512, 14)])
def test_stride2_dgrad_merged_classes(dt, cfg):
    """Data gradient of a 3x3 stride-2 conv as ONE 2x2-tap launch over the gradient map (qt_conv_desc.dst_merge,
    qt_pack_dgrad_s2_merged): values (+ residual, ReLU mask) against torch.nn.grad.conv2d_input, and the
    BatchNorm-backward link sums (4 partial rows per pixel tile) against the sums of the written gradient."""
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H = cfg
    Ho = H // 2
    g = torch.Generator().manual_seed(7)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cout * 9)) ** 0.5).to(dt).float()
    dy = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    other = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    act = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    ybn = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    mean = torch.randn(Cin, generator=g) * 0.1
    invstd = torch.rand(Cin, generator=g) + 0.5
    ref = (torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, 2, 1) + other) * (act > 0)

    lib = L.lib()
    wd = torch.full((16 * Cout * Cin,), float("nan"), dtype=dt, device=dev)   # the packer zeroes the unused slots
    L.check(lib.qt_pack_dgrad_s2_merged(L.qt_dtype(dt), L.ptr(w.to(dev)), L.ptr(wd), Cout, Cin, L.stream_ptr()),
            "qt_pack_dgrad_s2_merged")
    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
    d.in_h = d.in_w = Ho
    d.out_h = d.out_w = Ho
    d.k_per_tap, d.n_out = Cout, 4 * Cin
    d.kh, d.kw, d.stride, d.pad = 2, 2, 1, 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = Ho * Ho * Cout, Ho * Cout, Cout
    d.dst_sub, d.dst_h, d.dst_w, d.dst_off_h, d.dst_off_w, d.dst_merge = 2, H, H, 0, 0, Cin
    rows = lib.qt_conv2d_stats_rows(ctypes.byref(d))
    assert rows > 0 and rows % 4 == 0
    part = torch.full((rows, 2, Cin), float("nan"), dtype=torch.float32, device=dev)
    out = torch.full((B * H * H, Cin), float("nan"), dtype=dt, device=dev)   # every pixel belongs to one class
    dyd = nhwc(dy).to(dev, dt)
    res = nhwc(other).to(dev, dt).view(-1, Cin)
    msk = nhwc(act).to(dev, dt).view(-1, Cin)
    yb = nhwc(ybn).to(dev, dt).view(-1, Cin)
    md, isd = mean.to(dev), invstd.to(dev)
    io = L.ConvIO(L.ptr(dyd), L.ptr(wd), L.ptr(out), None, None, L.ptr(res), L.ptr(msk), None,
                  L.ptr(yb), L.ptr(md), L.ptr(isd), L.ptr(part), None, None, None, None)
    L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
    torch.cuda.synchronize()
    got = out.float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= TOL[dt]
    sums = part.sum(0).cpu()
    xhat = (ybn - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
    assert rel_err(sums[0], got.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(sums[1], (got * xhat).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    # the generic tile behind the same descriptor (qt_set_pt_conv(0)): same products in another order, residual only on class
    # (0,0) (dst_merge_res0: the sparse downsample-gradient map of the plan)
    if B >= 16:
        for res0 in (0, 1):
            d.dst_merge_res0 = res0
            outs = []
            for on in (1, 0):
                lib.qt_set_pt_conv(on)
                try:
                    o2 = torch.full((B * H * H, Cin), float("nan"), dtype=dt, device=dev)
                    rows2 = lib.qt_conv2d_stats_rows(ctypes.byref(d))
                    p2 = torch.zeros(rows2, 2, Cin, dtype=torch.float32, device=dev)
                    io2 = L.ConvIO(L.ptr(dyd), L.ptr(wd), L.ptr(o2), None, None, L.ptr(res), L.ptr(msk), None,
                                   L.ptr(yb), L.ptr(md), L.ptr(isd), L.ptr(p2), None, None, None, None)
                    L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io2), L.stream_ptr()), "qt_conv2d_igemm")
                    torch.cuda.synchronize()
                    outs.append((o2.float().cpu(), p2.sum(0).cpu()))
                finally:
                    lib.qt_set_pt_conv(-1)
            assert rel_err(outs[0][0], outs[1][0]) <= TOL[dt], res0
            assert rel_err(outs[0][1], outs[1][1]) <= 1e-3 + TOL[dt], res0
        d.dst_merge_res0 = 0
    # bad shapes are refused
    d.n_out = 2 * Cin
    assert lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()) != 0


class PackItem(ctypes.Structure):   # qt_pack_item
    _fields_ = [("w_oihw", ctypes.c_void_p), ("w_fwd", ctypes.c_void_p), ("w_dgrad", ctypes.c_void_p),
                ("O", ctypes.c_int), ("I", ctypes.c_int), ("k", ctypes.c_int), ("stride2_dgrad", ctypes.c_int)]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 64, 128, 56), (3, 128, 256, 28), (17, 64, 128, 56), (1, 64, 64, 12)])
def test_stride2_dgrad_merged_with_the_downsample_as_fifth_tap_slot(dt, cfg):
    """qt_conv_desc.dst_merge_extra: the data gradients of a transition block's conv1 (3x3 / 2) and of its downsample
    (1x1 / 2; torchvision BasicBlock) in ONE launch -- the downsample's gradient map is read through a
    fifth tap slot that only reaches parity class (0,0).  Operand packed by qt_pack_weights_batched (stride2_dgrad 3 + 4).
    Values (+ residual, ReLU mask) against torch.nn.grad.conv2d_input of both convolutions, link sums against the
    written gradient."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, Cin, Cout, H = cfg
    Ho = H // 2
    g = torch.Generator().manual_seed(11)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cout * 9)) ** 0.5).to(dt).float()
    wd = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cout) ** 0.5).to(dt).float()
    dy = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    dyd = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    other = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    act = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    ybn = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    mean = torch.randn(Cin, generator=g) * 0.1
    invstd = torch.rand(Cin, generator=g) + 0.5
    both = torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, 2, 1) + torch.nn.grad.conv2d_input((B, Cin, H, H), wd, dyd, 2, 0)

    esz = 4 if dt == torch.float32 else 2
    op = torch.zeros(20 * Cout * Cin, dtype=dt, device=dev)       # [4 Cin][5 slots][Cout]; unused slots stay zero
    wdev, wddev = w.to(dev).contiguous(), wd.to(dev).contiguous()
    items = (PackItem * 2)(PackItem(wdev.data_ptr(), None, op.data_ptr(), Cout, Cin, 3, 3),
                           PackItem(wddev.data_ptr(), None, op.data_ptr(), Cout, Cin, 1, 4))
    lib.qt_pack_weights_batched.argtypes = [ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    L.check(lib.qt_pack_weights_batched(L.qt_dtype(dt), items, 2, L.stream_ptr()), "qt_pack_weights_batched")
    # both gradient maps in ONE allocation (any two device tensors would do: the kernel takes their distance)
    maps = torch.empty(2, B, Ho, Ho, Cout, dtype=dt, device=dev)
    maps[0] = nhwc(dy).to(dev, dt)
    maps[1] = nhwc(dyd).to(dev, dt)

    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
    d.in_h = d.in_w = Ho
    d.out_h = d.out_w = Ho
    d.k_per_tap, d.n_out = Cout, 4 * Cin
    d.kh, d.kw, d.stride, d.pad = 2, 2, 1, 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = Ho * Ho * Cout, Ho * Cout, Cout
    d.dst_sub, d.dst_h, d.dst_w, d.dst_off_h, d.dst_off_w, d.dst_merge = 2, H, H, 0, 0, Cin
    d.dst_merge_extra = 1
    rows = lib.qt_conv2d_stats_rows(ctypes.byref(d))
    assert rows > 0 and rows % 4 == 0
    res = nhwc(other).to(dev, dt).view(-1, Cin)
    msk = nhwc(act).to(dev, dt).view(-1, Cin)
    yb = nhwc(ybn).to(dev, dt).view(-1, Cin)
    md, isd = mean.to(dev), invstd.to(dev)
    for with_ops in (False, True):
        part = torch.full((rows, 2, Cin), float("nan"), dtype=torch.float32, device=dev)
        out = torch.full((B * H * H, Cin), float("nan"), dtype=dt, device=dev)
        io = L.ConvIO(maps[0].data_ptr(), L.ptr(op), L.ptr(out), None, None, L.ptr(res) if with_ops else None,
                      L.ptr(msk) if with_ops else None, None,
                      L.ptr(yb), L.ptr(md), L.ptr(isd), L.ptr(part), None, None, None, None)
        io.extra_src = maps[1].data_ptr()
        L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
        torch.cuda.synchronize()
        got = out.float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2)
        ref = (both + other) * (act > 0) if with_ops else both
        assert rel_err(got, ref) <= TOL[dt], with_ops
        sums = part.sum(0).cpu()
        xhat = (ybn - mean.view(1, -1, 1, 1)) * invstd.view(1, -1, 1, 1)
        assert rel_err(sums[0], got.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
        assert rel_err(sums[1], (got * xhat).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    # the slot needs its second map
    io.extra_src = None
    assert lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()) != 0


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(2, 64, 128, 56, 3), (2, 128, 256, 28, 1), (3, 256, 512, 14, 3)])
def test_stride2_dgrad_by_parity_classes(dt, cfg):
    """Data gradient of a stride-2 conv as four stride-1 gathers with a strided destination."""
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, k = cfg
    p = 1 if k == 3 else 0
    Ho = (H + 2 * p - k) // 2 + 1
    g = torch.Generator().manual_seed(6)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cout * k * k)) ** 0.5).to(dt).float()
    dy = torch.randn(B, Cout, Ho, Ho, generator=g).to(dt).float()
    other = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    act = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    ref = (torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, 2, p) + other) * (act > 0)

    lib = L.lib()
    offs, khs, kws = (ctypes.c_longlong * 4)(), (ctypes.c_int * 4)(), (ctypes.c_int * 4)()
    wd = torch.empty(Cout * Cin * k * k, dtype=dt, device=dev)
    wsrc = w.to(dev)
    L.check(lib.qt_pack_dgrad_s2(L.qt_dtype(dt), L.ptr(wsrc), L.ptr(wd), Cout, Cin, k, offs, khs, kws, L.stream_ptr()),
            "qt_pack_dgrad_s2")
    dyd = nhwc(dy).to(dev, dt)
    out = torch.zeros(B * H * H, Cin, dtype=dt, device=dev)
    res = nhwc(other).to(dev, dt).view(-1, Cin)
    msk = nhwc(act).to(dev, dt).view(-1, Cin)
    esz = 2 if dt == torch.bfloat16 else 4
    for cls in range(4):
        if khs[cls] * kws[cls] == 0:
            continue
        d = L.ConvDesc()
        d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
        d.in_h = d.in_w = Ho
        d.out_h = d.out_w = H // 2
        d.k_per_tap, d.n_out = Cout, Cin
        d.kh, d.kw, d.stride, d.pad = khs[cls], kws[cls], 1, 0
        d.src_img_stride, d.src_row_stride, d.src_pix_stride = Ho * Ho * Cout, Ho * Cout, Cout
        d.dst_sub, d.dst_h, d.dst_w, d.dst_off_h, d.dst_off_w = 2, H, H, cls >> 1, cls & 1
        io = L.ConvIO(L.ptr(dyd), ctypes.c_void_p(wd.data_ptr() + offs[cls] * esz), L.ptr(out), None, None,
                      L.ptr(res), L.ptr(msk), None)
        L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
    torch.cuda.synchronize()
    got = out.float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2)
    if k == 1:  # pixels no tap reaches were never written: only class (0,0) is defined
        got, ref = got[:, :, ::2, ::2], ref[:, :, ::2, ::2]
    assert rel_err(got, ref) <= TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_batched_weight_pack_matches_per_layer_packing(dt):
    """qt_pack_weights_batched (one launch, LDS tile transposes) must reproduce qt_pack_conv_weight and
    the parity-class / merged layouts of qt_pack_dgrad_s2 / qt_pack_dgrad_s2_merged bit for bit."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()

    class Item(ctypes.Structure):
        _fields_ = [("w", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("dgrad", ctypes.c_void_p),
                    ("O", ctypes.c_int), ("I", ctypes.c_int), ("k", ctypes.c_int), ("s2", ctypes.c_int)]

    g = torch.Generator().manual_seed(31)
    # (O, I, k, stride2_dgrad): 1 = parity-class layout, 2 = merged layout (3x3 only; zero tap slots zeroed beforehand)
    shapes = [(64, 64, 3, 0), (128, 64, 3, 1), (128, 64, 1, 1), (512, 256, 3, 1), (128, 192, 1, 0), (2688, 5376, 1, 0),
              (128, 64, 3, 2), (512, 256, 3, 2)]
    qdt = L.qt_dtype(dt)
    st = L.stream_ptr()
    items = (Item * len(shapes))()
    keep, want = [], []
    for j, (O, I, k, s2) in enumerate(shapes):
        w = torch.randn(O, I, k, k, generator=g).to(dev)
        fwd = torch.zeros(O * k * k * I, dtype=dt, device=dev)
        dg = torch.zeros((16 if s2 == 2 else k * k) * O * I, dtype=dt, device=dev)
        rf, rd = torch.zeros_like(fwd), torch.zeros_like(dg)
        if s2 == 2:
            L.check(lib.qt_pack_conv_weight(qdt, L.ptr(w), L.ptr(rf), None, O, I, k, k, st), "qt_pack_conv_weight")
            L.check(lib.qt_pack_dgrad_s2_merged(qdt, L.ptr(w), L.ptr(rd), O, I, st), "qt_pack_dgrad_s2_merged")
        elif s2:
            L.check(lib.qt_pack_conv_weight(qdt, L.ptr(w), L.ptr(rf), None, O, I, k, k, st), "qt_pack_conv_weight")
            L.check(lib.qt_pack_dgrad_s2(qdt, L.ptr(w), L.ptr(rd), O, I, k, None, None, None, st), "qt_pack_dgrad_s2")
        else:
            L.check(lib.qt_pack_conv_weight(qdt, L.ptr(w), L.ptr(rf), L.ptr(rd), O, I, k, k, st), "qt_pack_conv_weight")
        items[j] = Item(w.data_ptr(), fwd.data_ptr(), dg.data_ptr(), O, I, k, s2)
        keep.append((w, fwd, dg))
        want.append((rf, rd))
    L.check(lib.qt_pack_weights_batched(qdt, items, len(shapes), st), "qt_pack_weights_batched")
    torch.cuda.synchronize()
    for (w, fwd, dg), (rf, rd), shp in zip(keep, want, shapes):
        assert torch.equal(fwd, rf), shp
        assert torch.equal(dg, rd), shp
    # bad shapes are refused, not mangled
    bad = (Item * 1)(Item(keep[0][0].data_ptr(), keep[0][1].data_ptr(), None, 48, 64, 3, 0))
    assert lib.qt_pack_weights_batched(qdt, bad, 1, st) == -1 and b"multiples of" in lib.qt_last_error()


@pytest.mark.parametrize("cfg", [
    (256, 2688, 5376, True, 1),    # classifier.0 forward
    (256, 5376, 2688, False, 0),   # its backward-input product
    (37, 2688, 5376, True, 1),     # ragged batch
    (5, 64, 64, True, 0),          # one tile, one K-step
    (200, 192, 448, False, 1),     # seven K-steps, not divisible into 256/3 slices
])
def test_linear_splitk_bf16(cfg):
    """qt_linear_bf16 (csrc/linear.hip) against F.linear in f32 on bf16-representable operands."""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    M, N, K, with_bias, relu = cfg
    g = torch.Generator().manual_seed(41)
    x = torch.randn(M, K, generator=g).to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) * 0.05).to(torch.bfloat16)
    b = torch.randn(N, generator=g) if with_bias else None
    ref = torch.nn.functional.linear(x.float(), w.float(), b)
    if relu:
        ref = torch.relu(ref)
    lib.qt_linear_workspace_bytes.restype = ctypes.c_size_t
    nbytes = lib.qt_linear_workspace_bytes(M, N, K)
    assert nbytes >= M * N * 4
    ws = torch.full((nbytes // 4,), float("nan"), dtype=torch.float32, device=dev)
    y = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    xd, wd = x.to(dev), w.to(dev)
    bd = b.to(dev) if b is not None else None
    L.check(lib.qt_linear_bf16(L.ptr(xd), L.ptr(wd), L.ptr(bd), relu, L.ptr(y), M, N, K, L.ptr(ws), ctypes.c_size_t(nbytes),
                               L.stream_ptr()), "qt_linear_bf16")
    torch.cuda.synchronize()
    assert rel_err(y.float().cpu(), ref) <= 4e-3   # one bf16 rounding of the output
    # uncovered shapes are reported, not mangled
    assert lib.qt_linear_bf16(L.ptr(xd), L.ptr(wd), L.ptr(bd), relu, L.ptr(y), 300, N, K, L.ptr(ws), ctypes.c_size_t(nbytes),
                              L.stream_ptr()) == -3   # QT_ERR_UNSUPPORTED
