"""GPU parity of the 8-wave ping-pong convolution kernel (csrc/conv_pp.hip) through the C ABI.

Two oracles per case:
* torch CPU fp32 conv2d / conv2d_input on the same (bf16-pre-rounded) operands, tolerances as tests/test_conv_gpu.py;
* the generic implicit-GEMM kernel behind the same entry point (qt_set_pp_conv(0)): both kernels issue the same MFMAs in
  the same K order, so the written tensors must be BIT-IDENTICAL, with every epilogue option (scale / shift / residual /
  ReLU / ReLU mask / BatchNorm-backward links), ragged pixel tiles, a channel count that is not a multiple of the tile,
  the quadrant region mode and the strided destination mapping of the stride-2 data gradients.  Per-tile statistics are
  summed in a different (fixed) order: compared after the sum over tiles, to f32 rounding.
"""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from _util import pkg, rel_err
from test_conv_gpu import TOL, nhwc, run_conv

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class _pp:
    """with _pp(L, on): ... runs the entry point with the ping-pong kernel on / off, restores the default after."""

    def __init__(self, L, on):
        self.lib, self.on = L.lib(), on

    def __enter__(self):
        self.lib.qt_set_pp_conv(1 if self.on else 0)

    def __exit__(self, *exc):
        self.lib.qt_set_pp_conv(-1)


FWD_CASES = [
    # B, Cin, Cout, H, k       pixels M (tile 256)                  what it covers
    (11, 128, 128, 28, 3),   # 8,624   ragged last tile, layer2 shape
    (45, 256, 256, 14, 3),   # 8,820   two channel tiles, layer3 shape
    (170, 512, 512, 7, 3),   # 8,330   four channel tiles, 72 K-tiles, layer4 shape
    (12, 128, 192, 28, 3),   # 9,408   192 channels: the second channel tile is half empty
    (44, 1024, 128, 14, 1),  # 8,624   1x1, 16 K-tiles
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", FWD_CASES)
def test_pp_forward_all_epilogues(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, k = cfg
    p = k // 2
    g = torch.Generator().manual_seed(11)
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * (2.0 / (Cin * k * k)) ** 0.5).to(dt).float()
    res = torch.randn(B, Cout, H, H, generator=g).to(dt).float()
    scale = torch.rand(Cout, generator=g) + 0.5
    shift = torch.randn(Cout, generator=g) * 0.1
    raw = F.conv2d(x, w, None, 1, p)
    ref = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    xd = nhwc(x).to(dev, dt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    resd = nhwc(res).to(dev, dt).view(-1, Cout)
    out = {}
    for on in (True, False):
        with _pp(L, on):
            y0, st = run_conv(L, dt, xd, wd, B, (H, H), (H, H), Cin, Cout, k, k, 1, p, L.QT_CONV_FWD, want_stats=True)
            if on:   # the statistics rows follow the kernel choice: one row per 256-pixel tile
                assert st.shape[0] == (B * H * H + 255) // 256
            y1, _ = run_conv(L, dt, xd, wd, B, (H, H), (H, H), Cin, Cout, k, k, 1, p, L.QT_CONV_FWD, relu=1,
                             scale=scale.to(dev), shift=shift.to(dev), residual=resd)
        out[on] = (y0, st.sum(0), y1)
    got = out[True][0].float().cpu().view(B, H, H, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, raw) <= TOL[dt]
    got = out[True][2].float().cpu().view(B, H, H, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= TOL[dt]
    ssum = out[True][1].cpu()
    assert rel_err(ssum[0], raw.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(ssum[1], (raw * raw).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert torch.equal(out[True][0], out[False][0])
    assert torch.equal(out[True][2], out[False][2])
    assert rel_err(out[True][1].cpu(), out[False][1].cpu()) <= 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(11, 128, 128, 28), (45, 256, 256, 14), (170, 512, 512, 7), (12, 192, 128, 28)])
def test_pp_dgrad_with_mask_residual_and_bn_links(dt, cfg):
    """Stride-1 data gradient with everything the backward chain fuses into it: + residual gradient, * (act > 0), and the
    BatchNorm-backward sums  sum g, sum g * xhat  of up to two BatchNorms that consume g (csrc/plan.hip links)."""
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H = cfg
    g = torch.Generator().manual_seed(12)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cout * 9)) ** 0.5).to(dt).float()
    dy = torch.randn(B, Cout, H, H, generator=g).to(dt).float()
    other = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    act = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    ys = [torch.randn(B, Cin, H, H, generator=g).to(dt).float() for _ in range(2)]
    mus = [torch.randn(Cin, generator=g) * 0.2 for _ in range(2)]
    iss = [torch.rand(Cin, generator=g) + 0.5 for _ in range(2)]
    dx = torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, 1, 1)
    ref = (dx + other) * (act > 0)
    dyd = nhwc(dy).to(dev, dt)
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, dt)  # [Cin][kh][kw][Cout]
    resd, mskd = nhwc(other).to(dev, dt).view(-1, Cin), nhwc(act).to(dev, dt).view(-1, Cin)
    ysd = [nhwc(t).to(dev, dt).view(-1, Cin) for t in ys]
    musd, issd = [t.to(dev) for t in mus], [t.to(dev) for t in iss]

    def run(nlinks):
        d = L.ConvDesc()
        d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_DGRAD; d.batch = B
        d.in_h = d.in_w = H; d.out_h = d.out_w = H
        d.k_per_tap, d.n_out = Cout, Cin
        d.kh = d.kw = 3; d.stride = 1; d.pad = 1
        d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * Cout, H * Cout, Cout
        y = torch.empty(B * H * H, Cin, dtype=dt, device=dev)
        io = L.ConvIO(L.ptr(dyd), L.ptr(wt), L.ptr(y), None, None, L.ptr(resd), L.ptr(mskd), None)
        rows = L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
        parts = [torch.zeros(rows, 2, Cin, device=dev) for _ in range(2)]
        if nlinks >= 1:
            io.bn0_y, io.bn0_mean, io.bn0_invstd, io.bn0_partial = (ysd[0].data_ptr(), musd[0].data_ptr(),
                                                                    issd[0].data_ptr(), parts[0].data_ptr())
        if nlinks >= 2:
            io.bn1_y, io.bn1_mean, io.bn1_invstd, io.bn1_partial = (ysd[1].data_ptr(), musd[1].data_ptr(),
                                                                    issd[1].data_ptr(), parts[1].data_ptr())
        L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
        torch.cuda.synchronize()
        return y, [p.sum(0).cpu() for p in parts]

    for nlinks in (0, 1, 2):
        with _pp(L, True):
            y_pp, s_pp = run(nlinks)
        with _pp(L, False):
            y_gen, s_gen = run(nlinks)
        assert torch.equal(y_pp, y_gen), nlinks
        got = y_pp.float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2)
        assert rel_err(got, ref) <= TOL[dt]
        gq = got.double()   # the sums are taken over the value actually written (rounded to the activation type)
        for k in range(nlinks):
            xhat = (ys[k].double() - mus[k].double().view(1, -1, 1, 1)) * iss[k].double().view(1, -1, 1, 1)
            assert rel_err(s_pp[k][0], gq.sum((0, 2, 3))) <= 2e-4
            assert rel_err(s_pp[k][1], (gq * xhat).sum((0, 2, 3))) <= 2e-4
            assert rel_err(s_pp[k], s_gen[k]) <= 1e-5


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pp_quadrant_mode_and_strided_destination(dt):
    dev = _dev()
    L = pkg("_lib")
    # (a) quadrant forward at a batch that reaches the ping-pong kernel: 42 images x 4 regions x 49 pixels = 8,232
    B, C, N = 42, 256, 128
    g = torch.Generator().manual_seed(13)
    base = torch.randn(B, C, 14, 14, generator=g).to(dt).float()
    w = (torch.randn(N, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).to(dt).float()
    bias = torch.randn(N, generator=g) * 0.1
    quads = [base[:, :, :7, :7], base[:, :, :7, 7:], base[:, :, 7:, :7], base[:, :, 7:, 7:]]
    ref = torch.stack([F.relu(F.conv2d(q, w, bias, 1, 1)) for q in quads], 1)  # [B,4,N,7,7]
    xd = nhwc(base).to(dev, dt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    ys = {}
    for on in (True, False):
        with _pp(L, on):
            ys[on], _ = run_conv(L, dt, xd, wd, B, (7, 7), (7, 7), C, N, 3, 3, 1, 1, L.QT_CONV_FWD, quad=1, relu=1,
                                 shift=bias.to(dev), strides=(14 * 14 * C, 14 * C, C))
    assert torch.equal(ys[True], ys[False])
    got = ys[True].float().cpu().view(B, 4, 7, 7, N).permute(0, 1, 4, 2, 3)
    assert rel_err(got, ref) <= TOL[dt]
    # quadrant data gradient onto the un-split map: 42 x 196 = 8,232 pixels
    dyq = torch.randn(B, 4, N, 7, 7, generator=g).to(dt).float()
    dbase = torch.zeros(B, C, 14, 14)
    sl = [(slice(0, 7), slice(0, 7)), (slice(0, 7), slice(7, 14)), (slice(7, 14), slice(0, 7)), (slice(7, 14), slice(7, 14))]
    for q in range(4):
        dbase[:, :, sl[q][0], sl[q][1]] = torch.nn.grad.conv2d_input((B, C, 7, 7), w, dyq[:, q], 1, 1)
    dyd = dyq.permute(0, 1, 3, 4, 2).contiguous().to(dev, dt)
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, dt)
    for on in (True, False):
        with _pp(L, on):
            ys[on], _ = run_conv(L, dt, dyd, wt, B, (7, 7), (14, 14), N, C, 3, 3, 1, 1, L.QT_CONV_DGRAD, quad=1,
                                 strides=(49 * N, 7 * N, N))
    assert torch.equal(ys[True], ys[False])
    assert rel_err(ys[True].float().cpu().view(B, 14, 14, C).permute(0, 3, 1, 2), dbase) <= TOL[dt]

    # (b) one parity class of a stride-2 data gradient (csrc/plan.hip::dgrad): a 2x2-tap stride-1 pad-0 gather whose
    # row m = (img, oh, ow) lands on pixel (2*oh + 1, 2*ow + 1) of a 28x28 destination
    Bc, K, Nc, Hs = 44, 256, 128, 15   # source 15x15 -> 14x14 outputs -> 44 * 196 = 8,624 rows
    src = torch.randn(Bc, K, Hs, Hs, generator=g).to(dt).float()
    wc = (torch.randn(Nc, K, 2, 2, generator=g) * (2.0 / (K * 4)) ** 0.5).to(dt).float()
    refc = F.conv2d(src, wc, None, 1, 0)   # [Bc, Nc, 14, 14]
    srcd = nhwc(src).to(dev, dt)
    wcd = wc.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    for on in (True, False):
        with _pp(L, on):
            d = L.ConvDesc()
            d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_FWD; d.batch = Bc
            d.in_h = d.in_w = Hs; d.out_h = d.out_w = 14
            d.k_per_tap, d.n_out = K, Nc
            d.kh = d.kw = 2; d.stride = 1; d.pad = 0
            d.src_img_stride, d.src_row_stride, d.src_pix_stride = Hs * Hs * K, Hs * K, K
            d.dst_sub, d.dst_h, d.dst_w, d.dst_off_h, d.dst_off_w = 2, 28, 28, 1, 1
            y = torch.full((Bc * 28 * 28, Nc), 7.0, dtype=dt, device=dev)
            io = L.ConvIO(L.ptr(srcd), L.ptr(wcd), L.ptr(y), None, None, None, None, None)
            L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
            torch.cuda.synchronize()
            ys[on] = y
    assert torch.equal(ys[True], ys[False])
    full = ys[True].float().cpu().view(Bc, 28, 28, Nc)
    assert rel_err(full[:, 1::2, 1::2].permute(0, 3, 1, 2), refc) <= TOL[dt]
    assert bool((full[:, 0::2] == 7.0).all()) and bool((full[:, :, 0::2] == 7.0).all())   # other classes untouched


def test_pp_is_the_kernel_that_runs_and_thin_problems_keep_the_generic_one():
    """qt_conv2d_stats_rows mirrors the dispatch: 256-pixel tiles exactly where the ping-pong kernel is eligible."""
    _dev()
    L = pkg("_lib")

    def rows(B, H, cin, cout, k, stride=1, dt=torch.bfloat16):
        d = L.ConvDesc()
        d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_FWD; d.batch = B
        d.in_h = d.in_w = H; d.out_h = d.out_w = H // stride
        d.k_per_tap, d.n_out = cin, cout
        d.kh = d.kw = k; d.stride = stride; d.pad = k // 2
        return L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
    L.lib().qt_set_pp_conv(-1)
    assert rows(256, 28, 128, 128, 3) == 256 * 784 // 256          # layer2
    assert rows(256, 14, 256, 256, 3) == 196                        # layer3
    assert rows(256, 7, 512, 512, 3) == 49                          # layer4
    assert rows(4, 28, 128, 128, 3) == (4 * 784 + 127) // 128       # thin: generic 128-pixel tiles
    assert rows(256, 28, 64, 128, 1, stride=2) == (256 * 196 + 127) // 128   # stride 2: generic
    L.lib().qt_set_pp_conv(0)
    assert rows(256, 14, 256, 256, 3) == 392
    L.lib().qt_set_pp_conv(-1)
