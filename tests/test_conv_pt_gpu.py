"""GPU parity of the patch-resident ping-pong convolution kernel (csrc/conv_pt.hip) through the C ABI.

Two oracles per case:
* torch CPU fp32 conv2d / conv2d_input on the same (bf16-pre-rounded) operands, tolerances as tests/test_conv_gpu.py;
* the generic implicit-GEMM kernel behind the same entry point (qt_set_pt_conv(0)).  The two kernels sum the same
  products in a different K order (chunk-major / tap-minor against tap-major), so the comparison is to rounding, with
  every epilogue option (scale / shift / residual / ReLU / ReLU mask / BatchNorm-backward links), a single channel
  chunk (only the "last chunk" code path), several channel tiles, a ragged last pixel tile (7x7: four images per tile)
  and the full benchmark batch.
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


class _pt:
    """with _pt(L, on): ... runs the entry point with the patch-resident kernel on / off, restores the default after."""

    def __init__(self, L, on):
        self.lib, self.on = L.lib(), on

    def __enter__(self):
        self.lib.qt_set_pt_conv(1 if self.on else 0)

    def __exit__(self, *exc):
        self.lib.qt_set_pt_conv(-1)


def tiles(B, H):
    return {28: 4 * B, 14: B, 7: (B + 3) // 4}[H]


class _maxwg:
    """with _maxwg(L, n): caps the persistent grid at n workgroups (0 = one per CU): small batches then walk SEVERAL items
    per workgroup -- the K-tile stream crosses item boundaries, the epilogue runs between two K-tiles, the branch-free
    DMA slots behind the last item go through out-of-range offsets / a zero-record resource."""

    def __init__(self, L, n):
        self.lib, self.n = L.lib(), n

    def __enter__(self):
        self.lib.qt_set_pt_conv_max_workgroups(self.n)

    def __exit__(self, *exc):
        self.lib.qt_set_pt_conv_max_workgroups(0)


FWD_CASES = [
    # B, Cin, Cout, H        pixel tiles            what it covers
    (18, 128, 128, 28),    # 72    layer2 shape: quarter-image tiles, halo rows are real pixels of the same image
    (20, 256, 256, 14),    # 20    layer3 shape: one image per tile, two channel tiles
    (22, 512, 512, 7),     # 6     layer4 shape: four images per tile, the last tile holds two; four channel tiles
    (17, 64, 128, 28),     # 68    one channel chunk: only the last-chunk code path runs
    (16, 128, 384, 14),    # 16    three channel tiles
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", FWD_CASES)
def test_pt_forward_all_epilogues(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H = cfg
    k, p = 3, 1
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
        with _pt(L, on):
            y0, st = run_conv(L, dt, xd, wd, B, (H, H), (H, H), Cin, Cout, k, k, 1, p, L.QT_CONV_FWD, want_stats=True)
            if on:   # the statistics rows follow the kernel choice: one row per 196-pixel tile and wave row
                assert st.shape[0] == 2 * tiles(B, H)
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
    assert rel_err(out[True][0].float().cpu(), out[False][0].float().cpu()) <= TOL[dt]
    assert rel_err(out[True][2].float().cpu(), out[False][2].float().cpu()) <= TOL[dt]
    assert rel_err(out[True][1].cpu(), out[False][1].cpu()) <= 1e-4


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(18, 128, 128, 28), (20, 256, 256, 14), (22, 512, 512, 7), (17, 128, 64, 28)])
def test_pt_dgrad_with_mask_residual_and_bn_links(dt, cfg):
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
        with _pt(L, True):
            y_pt, s_pt = run(nlinks)
        with _pt(L, False):
            y_gen, s_gen = run(nlinks)
        assert rel_err(y_pt.float().cpu(), y_gen.float().cpu()) <= TOL[dt], nlinks
        got = y_pt.float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2)
        assert rel_err(got, ref) <= TOL[dt]
        gq = got.double()   # the sums are taken over the value actually written (rounded to the activation type)
        for k in range(nlinks):
            xhat = (ys[k].double() - mus[k].double().view(1, -1, 1, 1)) * iss[k].double().view(1, -1, 1, 1)
            assert rel_err(s_pt[k][0], gq.sum((0, 2, 3))) <= (2e-4 if dt == torch.float32 else 5e-3)  # (sums of the f32 value before its bf16 rounding)
            assert rel_err(s_pt[k][1], (gq * xhat).sum((0, 2, 3))) <= (2e-4 if dt == torch.float32 else 5e-3)
            assert rel_err(s_pt[k], s_gen[k]) <= (1e-4 if dt == torch.float32 else 5e-3)


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
def test_pt_quadrant_conv_forward_and_data_gradient(dt):
    """The quadrant head (Quadtree_from scratch/models.py:277-287: conv 256 -> 128 on the four 7x7 quadrants of the
    14x14 map, zero padding AT the seam) through the stacked-7x7 geometry: forward reads the un-split map, the data
    gradient scatters the four per-quadrant images back onto it."""
    dev = _dev()
    L = pkg("_lib")
    B, C, N = 19, 256, 128
    g = torch.Generator().manual_seed(3)
    base = torch.randn(B, C, 14, 14, generator=g).to(dt).float()
    w = (torch.randn(N, C, 3, 3, generator=g) * (2.0 / (C * 9)) ** 0.5).to(dt).float()
    bias = torch.randn(N, generator=g) * 0.1
    quads = [base[:, :, :7, :7], base[:, :, :7, 7:], base[:, :, 7:, :7], base[:, :, 7:, 7:]]
    ref = torch.stack([F.relu(F.conv2d(qd, w, bias, 1, 1)) for qd in quads], 1)  # [B,4,N,7,7]
    xd = nhwc(base).to(dev, dt)
    wd = w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    ys = {}
    for on in (True, False):
        with _pt(L, on):
            ys[on], _ = run_conv(L, dt, xd, wd, B, (7, 7), (7, 7), C, N, 3, 3, 1, 1, L.QT_CONV_FWD, quad=1, relu=1,
                                 shift=bias.to(dev), strides=(14 * 14 * C, 14 * C, C))
    got = ys[True].float().cpu().view(B, 4, 7, 7, N).permute(0, 1, 4, 2, 3)
    assert rel_err(got, ref) <= TOL[dt]
    assert rel_err(ys[True].float().cpu(), ys[False].float().cpu()) <= TOL[dt]
    dyq = torch.randn(B, 4, N, 7, 7, generator=g).to(dt).float()
    dbase = torch.zeros(B, C, 14, 14)
    sl = [(slice(0, 7), slice(0, 7)), (slice(0, 7), slice(7, 14)), (slice(7, 14), slice(0, 7)), (slice(7, 14), slice(7, 14))]
    for k in range(4):
        dbase[:, :, sl[k][0], sl[k][1]] = torch.nn.grad.conv2d_input((B, C, 7, 7), w, dyq[:, k], 1, 1)
    dyd = dyq.permute(0, 1, 3, 4, 2).contiguous().to(dev, dt)
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, dt)
    res = torch.randn(B, C, 14, 14, generator=g).to(dt).float()
    for on in (True, False):
        with _pt(L, on):
            ys[on], _ = run_conv(L, dt, dyd, wt, B, (7, 7), (14, 14), N, C, 3, 3, 1, 1, L.QT_CONV_DGRAD, quad=1,
                                 strides=(49 * N, 7 * N, N), residual=nhwc(res).to(dev, dt).view(-1, C))
    got = ys[True].float().cpu().view(B, 14, 14, C).permute(0, 3, 1, 2)
    assert rel_err(got, dbase + res) <= TOL[dt]
    assert rel_err(ys[True].float().cpu(), ys[False].float().cpu()) <= TOL[dt]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(18, 128, 128, 28, 7), (17, 64, 128, 28, 5), (20, 128, 256, 28, 16), (22, 512, 256, 7, 3),
                                 (16, 128, 128, 14, 4)])
def test_pt_persistent_walk_over_several_items(dt, cfg):
    """Several (channel tile, pixel tile) items per workgroup (the benchmark's 28x28 stage walks four): forward with
    statistics + scale / shift / residual / ReLU and the data gradient with residual, mask and two BatchNorm links
    against torch and against the same launch with one item per workgroup -- bit for bit (the walk changes no sum)."""
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, maxwg = cfg
    g = torch.Generator().manual_seed(31)
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).float()
    res = torch.randn(B, Cout, H, H, generator=g).to(dt).float()
    scale, shift = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    raw = F.conv2d(x, w, None, 1, 1)
    ref = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    xd, wd = nhwc(x).to(dev, dt), w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    resd = nhwc(res).to(dev, dt).view(-1, Cout)
    outs = {}
    for n in (maxwg, 1 << 20):
        with _maxwg(L, n):
            y0, st = run_conv(L, dt, xd, wd, B, (H, H), (H, H), Cin, Cout, 3, 3, 1, 1, L.QT_CONV_FWD, want_stats=True)
            y1, _ = run_conv(L, dt, xd, wd, B, (H, H), (H, H), Cin, Cout, 3, 3, 1, 1, L.QT_CONV_FWD, relu=1,
                             scale=scale.to(dev), shift=shift.to(dev), residual=resd)
        outs[n] = (y0, st, y1)
    a, b = outs[maxwg], outs[1 << 20]
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    got = a[0].float().cpu().view(B, H, H, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, raw) <= TOL[dt]
    got = a[2].float().cpu().view(B, H, H, Cout).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= TOL[dt]
    assert rel_err(a[1].sum(0)[0].cpu(), raw.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(a[1].sum(0)[1].cpu(), (raw * raw).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    # data gradient: y [B,Cout,H,H] -> dx [B,Cin,H,H]
    dy = torch.randn(B, Cout, H, H, generator=g).to(dt).float()
    other, act = torch.randn(B, Cin, H, H, generator=g).to(dt).float(), torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    if Cin % 128:
        return   # (a 64-channel data gradient takes the generic kernel)
    dxr = (torch.nn.grad.conv2d_input((B, Cin, H, H), w, dy, 1, 1) + other) * (act > 0)
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, dt)
    dyd = nhwc(dy).to(dev, dt)
    od, ad = nhwc(other).to(dev, dt).view(-1, Cin), nhwc(act).to(dev, dt).view(-1, Cin)
    douts = {}
    for n in (maxwg, 1 << 20):
        with _maxwg(L, n):
            douts[n], _ = run_conv(L, dt, dyd, wt, B, (H, H), (H, H), Cout, Cin, 3, 3, 1, 1, L.QT_CONV_DGRAD, residual=od,
                                   relu_mask=ad)
    assert torch.equal(douts[maxwg], douts[1 << 20])
    assert rel_err(douts[maxwg].float().cpu().view(B, H, H, Cin).permute(0, 3, 1, 2), dxr) <= TOL[dt]


def test_pt_full_benchmark_batch_matches_generic_kernel():
    """B = 256, bf16, the three stage shapes of the benchmark: every CU holds a tile (1024 / 512 / 256 workgroups);
    forward with statistics and the data gradient with mask + residual against the generic kernel."""
    dev = _dev()
    L = pkg("_lib")
    dt = torch.bfloat16
    B = 256
    g = torch.Generator(device=dev).manual_seed(5)
    for (H, C) in ((28, 128), (14, 256), (7, 512)):
        x = torch.randn(B, H, H, C, device=dev, generator=g).to(dt)
        w = (torch.randn(C, 3, 3, C, device=dev, generator=g) * (2.0 / (C * 9)) ** 0.5).to(dt)
        res = torch.randn(B * H * H, C, device=dev, generator=g).to(dt)
        msk = torch.randn(B * H * H, C, device=dev, generator=g).to(dt)
        out = {}
        for on in (True, False):
            with _pt(L, on):
                y, st = run_conv(L, dt, x, w, B, (H, H), (H, H), C, C, 3, 3, 1, 1, L.QT_CONV_FWD, want_stats=True)
                dx, _ = run_conv(L, dt, x, w, B, (H, H), (H, H), C, C, 3, 3, 1, 1, L.QT_CONV_DGRAD, residual=res,
                                 relu_mask=msk)
            out[on] = (y.float(), st.sum(0), dx.float())
        assert out[True][1].shape == out[False][1].shape
        for k in (0, 2):
            a, b = out[True][k], out[False][k]
            assert float((a - b).abs().max() / b.abs().max()) <= TOL[dt], (H, k)
        assert rel_err(out[True][1].cpu(), out[False][1].cpu()) <= 1e-4


def test_pt_is_chosen_exactly_where_it_is_eligible():
    """qt_conv2d_stats_rows mirrors the dispatch: two rows per 196-pixel tile exactly where the kernel is eligible."""
    _dev()
    L = pkg("_lib")

    def rows(B, H, cin, cout, k=3, stride=1, quad=0, dt=torch.bfloat16):
        d = L.ConvDesc()
        d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_FWD; d.batch = B
        d.in_h = d.in_w = H; d.out_h = d.out_w = H // stride
        d.k_per_tap, d.n_out = cin, cout
        d.kh = d.kw = k; d.stride = stride; d.pad = k // 2; d.quad = quad
        return L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
    L.lib().qt_set_pt_conv(-1)
    # (two partial rows per tile: one per wave row, no cross-wave reduction in the kernel)
    assert rows(256, 28, 128, 128) == 2 * 1024                       # layer2: quarter images
    assert rows(256, 14, 256, 256) == 2 * 256                        # layer3: images
    assert rows(256, 7, 512, 512) == 2 * 64                          # layer4: four images
    assert rows(30, 7, 512, 512) == 2 * 8                            # ragged: 7 full tiles + 2 images
    assert rows(4, 28, 128, 128) == (4 * 784 + 127) // 128           # few images: generic 128-pixel tiles
    assert rows(256, 28, 64, 128, k=1, stride=2) == (256 * 196 + 127) // 128   # 1x1 stride 2: generic
    assert rows(256, 28, 128, 64) == (256 * 784 + 127) // 128        # 64 output channels: generic
    assert rows(64, 7, 256, 128, quad=1) == 2 * 64                   # quadrant mode: the four quadrants of a map per tile
    assert rows(3, 7, 256, 128, quad=1) == (3 * 4 * 49 + 127) // 128 # 12 region images: generic
    L.lib().qt_set_pt_conv(0)
    assert rows(256, 14, 256, 256) == 392
    L.lib().qt_set_pt_conv(-1)
