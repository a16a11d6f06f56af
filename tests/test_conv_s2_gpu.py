"""GPU parity of the fused stride-2 transition kernel (csrc/conv_s2.hip: conv1 3x3/2 of layerN.0 + its 1x1/2 downsample in
one launch) through the C ABI (qt_conv_s2_pair).

Oracles per case:
* torch CPU fp32 conv2d on the same (bf16-pre-rounded) operands, both outputs, raw and with the eval epilogues
  (scale / shift / ReLU on conv1, scale / shift on the downsample), and the per-channel BatchNorm sums;
* the generic implicit GEMM (qt_conv2d_igemm, two launches) the pair replaces: same products, another K order.
Cases cover the three ResNet-18 transitions (quarter-image tiles with real halo rows, one image per tile, four stacked
7x7 images), one and several channel chunks, several channel tiles, batches that leave the last workgroup short, and
SEVERAL ITEMS PER WORKGROUP (qt_set_conv_s2_max_workgroups caps the persistent grid: the K-tile stream crosses item
boundaries, the branch-free DMA slots after the last item run through the zero-record resource).
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


def run_pair(L, dt, xd, wc, wd_, B, H, Cin, Cout, relu_conv=0, sc=None, sh=None, sd=None, shd=None, want_stats=False):
    dev = xd.device
    d = L.ConvS2Desc(L.qt_dtype(dt), B, H, H, Cin, Cout, relu_conv, 0)
    assert L.lib().qt_conv_s2_pair_supported(ctypes.byref(d)) == 1
    OH = H // 2
    y = torch.full((B * OH * OH, Cout), float("nan"), dtype=dt, device=dev)
    yd = torch.full((B * OH * OH, Cout), float("nan"), dtype=dt, device=dev)
    rows = L.lib().qt_conv_s2_pair_stats_rows(ctypes.byref(d))
    st = torch.full((rows, 2, Cout), float("nan"), device=dev) if want_stats else None
    std = torch.full((rows, 2, Cout), float("nan"), device=dev) if want_stats else None
    io = L.ConvS2IO(L.ptr(xd), L.ptr(wc), L.ptr(wd_), L.ptr(y), L.ptr(yd), L.ptr(sc), L.ptr(sh), L.ptr(sd), L.ptr(shd),
                    L.ptr(st), L.ptr(std))
    L.check(L.lib().qt_conv_s2_pair(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv_s2_pair")
    torch.cuda.synchronize()
    return y, yd, st, std


CASES = [
    # B, Cin, Cout, H_in, max workgroups (0 = one per CU)
    (16, 64, 128, 56, 0),      # layer2.0: 64 quarter-image tiles, one chunk (bf16) / two (f32)
    (18, 64, 128, 56, 7),      # ... 72 items on 7 workgroups: 11 per workgroup, the last one short (6)
    (16, 128, 256, 28, 0),     # layer3.0: one image per tile, two channel tiles
    (20, 128, 256, 28, 8),     # ... 40 items on 8 workgroups (5 each: channel tiles of a pixel tile split over workgroups)
    (17, 128, 256, 28, 17),    # ... 34 items, 2 per workgroup: both channel tiles of a pixel tile in one workgroup
    (16, 256, 512, 14, 0),     # layer4.0: four stacked images per tile, four channel tiles, four chunks
    (24, 256, 512, 14, 5),     # ... 24 items on 5 workgroups
    (16, 128, 128, 56, 3),     # two chunks at 28x28 outputs, many items per workgroup
]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", CASES)
def test_s2_pair_matches_torch_and_generic(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    B, Cin, Cout, H, maxwg = cfg
    OH = H // 2
    g = torch.Generator().manual_seed(21)
    x = torch.randn(B, Cin, H, H, generator=g).to(dt).float()
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cin * 9)) ** 0.5).to(dt).float()
    wds = (torch.randn(Cout, Cin, 1, 1, generator=g) * (2.0 / Cin) ** 0.5).to(dt).float()
    sc, sh = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    sd, shd = torch.rand(Cout, generator=g) + 0.5, torch.randn(Cout, generator=g) * 0.1
    raw = F.conv2d(x, w, None, 2, 1)
    rawd = F.conv2d(x, wds, None, 2, 0)
    ref = F.relu(raw * sc.view(1, -1, 1, 1) + sh.view(1, -1, 1, 1))
    refd = rawd * sd.view(1, -1, 1, 1) + shd.view(1, -1, 1, 1)
    xd = nhwc(x).to(dev, dt)
    wc = w.permute(0, 2, 3, 1).contiguous().to(dev, dt)
    wdd = wds.view(Cout, Cin).contiguous().to(dev, dt)
    L.lib().qt_set_conv_s2_max_workgroups(maxwg)
    try:
        y0, yd0, st, std = run_pair(L, dt, xd, wc, wdd, B, H, Cin, Cout, want_stats=True)
        y1, yd1, _, _ = run_pair(L, dt, xd, wc, wdd, B, H, Cin, Cout, relu_conv=1, sc=sc.to(dev), sh=sh.to(dev),
                                 sd=sd.to(dev), shd=shd.to(dev))
    finally:
        L.lib().qt_set_conv_s2_max_workgroups(0)

    def back(t):
        return t.float().cpu().view(B, OH, OH, Cout).permute(0, 3, 1, 2)
    assert rel_err(back(y0), raw) <= TOL[dt]
    assert rel_err(back(yd0), rawd) <= TOL[dt]
    assert rel_err(back(y1), ref) <= TOL[dt]
    assert rel_err(back(yd1), refd) <= TOL[dt]
    # BatchNorm partial sums of the RAW outputs: every row written (no NaN left), totals = per-channel sums
    assert torch.isfinite(st).all() and torch.isfinite(std).all()
    s, sdn = st.sum(0).cpu(), std.sum(0).cpu()
    assert rel_err(s[0], raw.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(s[1], (raw * raw).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(sdn[0], rawd.sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    assert rel_err(sdn[1], (rawd * rawd).sum((0, 2, 3))) <= 1e-3 + TOL[dt]
    # the two generic launches the pair replaces
    g0, _ = run_conv(L, dt, xd, wc, B, (H, H), (OH, OH), Cin, Cout, 3, 3, 2, 1, L.QT_CONV_FWD)
    g1, _ = run_conv(L, dt, xd, wdd, B, (H, H), (OH, OH), Cin, Cout, 1, 1, 2, 0, L.QT_CONV_FWD)
    assert rel_err(y0.float().cpu(), g0.float().cpu()) <= TOL[dt]
    assert rel_err(yd0.float().cpu(), g1.float().cpu()) <= TOL[dt]


def test_s2_pair_is_deterministic_and_rejects_what_it_does_not_cover():
    dev = _dev()
    L = pkg("_lib")
    dt = torch.bfloat16
    B, Cin, Cout, H = 32, 64, 128, 56
    g = torch.Generator().manual_seed(22)
    xd = torch.randn(B, H, H, Cin, generator=g).to(dev, dt)
    wc = (torch.randn(Cout, 3, 3, Cin, generator=g) * 0.05).to(dev, dt)
    wdd = (torch.randn(Cout, Cin, generator=g) * 0.1).to(dev, dt)
    a = run_pair(L, dt, xd, wc, wdd, B, H, Cin, Cout, want_stats=True)
    b = run_pair(L, dt, xd, wc, wdd, B, H, Cin, Cout, want_stats=True)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    lib = L.lib()
    for bad in (L.ConvS2Desc(1, 8, 56, 56, 64, 128, 0, 0),      # batch < 16
                L.ConvS2Desc(1, 18, 14, 14, 256, 512, 0, 0),    # 7x7 outputs: batch must be a multiple of 4
                L.ConvS2Desc(1, 16, 56, 56, 64, 64, 0, 0),      # 64 output channels
                L.ConvS2Desc(1, 16, 112, 112, 64, 128, 0, 0),   # 56x56 outputs
                L.ConvS2Desc(1, 16, 56, 56, 48, 128, 0, 0)):    # ragged channel chunk
        assert lib.qt_conv_s2_pair_supported(ctypes.byref(bad)) == 0
        io = L.ConvS2IO(L.ptr(xd), L.ptr(wc), L.ptr(wdd), L.ptr(a[0]), L.ptr(a[1]))
        assert lib.qt_conv_s2_pair(ctypes.byref(bad), ctypes.byref(io), L.stream_ptr()) != 0
    lib.qt_set_conv_s2(0)
    try:
        assert lib.qt_conv_s2_pair_supported(ctypes.byref(L.ConvS2Desc(1, B, H, H, Cin, Cout, 0, 0))) == 0
    finally:
        lib.qt_set_conv_s2(-1)
