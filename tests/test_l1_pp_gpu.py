"""GPU parity of the ping-pong cut of the persistent layer1 kernel (csrc/conv_patch.hip::conv_l1_pp_kernel: 56x56 maps,
64 -> 64 channels, bf16) through qt_conv2d_igemm: forward (raw + BatchNorm sums; scale / shift / residual / ReLU) and
data gradient (residual; ReLU mask + one BatchNorm link; residual + mask + link; nothing) against torch CPU fp32 and --
bit for bit -- against the round-1 ring kernel (qt_set_l1_pingpong(0)), at batches where a workgroup walks one tile, several
tiles with a wrapping window (26 images: 342 tiles on 256 workgroups) and many (70 images: 920 tiles)."""
import ctypes

import pytest
import torch
import torch.nn.functional as F

from _util import pkg, rel_err
from test_conv_gpu import TOL, nhwc, run_conv

pytestmark = pytest.mark.gpu
DT = torch.bfloat16
H = C = 56
CH = 64


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


def _both(L, fn):
    out = {}
    for pp in (1, 0):
        L.lib().qt_set_l1_pingpong(pp)
        try:
            out[pp] = fn()
        finally:
            L.lib().qt_set_l1_pingpong(-1)
    return out[1], out[0]


@pytest.mark.parametrize("B", [2, 26, 70])
def test_l1_pp_forward(B):
    dev = _dev()
    L = pkg("_lib")
    g = torch.Generator().manual_seed(41)
    x = torch.randn(B, CH, H, H, generator=g).to(DT).float()
    w = (torch.randn(CH, CH, 3, 3, generator=g) * (2.0 / (CH * 9)) ** 0.5).to(DT).float()
    res = torch.randn(B, CH, H, H, generator=g).to(DT).float()
    scale, shift = torch.rand(CH, generator=g) + 0.5, torch.randn(CH, generator=g) * 0.1
    raw = F.conv2d(x, w, None, 1, 1)
    ref = F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1) + res)
    xd, wd = nhwc(x).to(dev, DT), w.permute(0, 2, 3, 1).contiguous().to(dev, DT)
    resd = nhwc(res).to(dev, DT).view(-1, CH)

    def run():
        y0, st = run_conv(L, DT, xd, wd, B, (H, H), (H, H), CH, CH, 3, 3, 1, 1, L.QT_CONV_FWD, want_stats=True)
        y1, _ = run_conv(L, DT, xd, wd, B, (H, H), (H, H), CH, CH, 3, 3, 1, 1, L.QT_CONV_FWD, relu=1, scale=scale.to(dev),
                         shift=shift.to(dev), residual=resd)
        y2, _ = run_conv(L, DT, xd, wd, B, (H, H), (H, H), CH, CH, 3, 3, 1, 1, L.QT_CONV_FWD, relu=1, scale=scale.to(dev),
                         shift=shift.to(dev))
        return y0, st, y1, y2
    a, b = _both(L, run)
    for u, v in zip(a, b):
        assert torch.equal(u, v)                       # same sums in the same order as the ring kernel
    back = lambda t: t.float().cpu().view(B, H, H, CH).permute(0, 3, 1, 2)
    assert rel_err(back(a[0]), raw) <= TOL[DT]
    assert rel_err(back(a[2]), ref) <= TOL[DT]
    assert rel_err(back(a[3]), F.relu(raw * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1))) <= TOL[DT]
    s = a[1].sum(0).cpu()
    assert rel_err(s[0], raw.sum((0, 2, 3))) <= 1e-3 + TOL[DT]
    assert rel_err(s[1], (raw * raw).sum((0, 2, 3))) <= 1e-3 + TOL[DT]


@pytest.mark.parametrize("B", [2, 26, 70])
@pytest.mark.parametrize("ops", ["none", "res", "mask+link", "res+mask+link"])
def test_l1_pp_data_gradient(B, ops):
    dev = _dev()
    L = pkg("_lib")
    g = torch.Generator().manual_seed(42)
    w = (torch.randn(CH, CH, 3, 3, generator=g) * (2.0 / (CH * 9)) ** 0.5).to(DT).float()
    dy = torch.randn(B, CH, H, H, generator=g).to(DT).float()
    other = torch.randn(B, CH, H, H, generator=g).to(DT).float()
    act = torch.randn(B, CH, H, H, generator=g).to(DT).float()
    ysv = torch.randn(B, CH, H, H, generator=g).to(DT).float()
    mu, isd = torch.randn(CH, generator=g) * 0.2, torch.rand(CH, generator=g) + 0.5
    dx = torch.nn.grad.conv2d_input((B, CH, H, H), w, dy, 1, 1)
    use_res, use_msk, use_link = "res" in ops, "mask" in ops, "link" in ops
    ref = dx + (other if use_res else 0)
    if use_msk:
        ref = ref * (act > 0)
    dyd = nhwc(dy).to(dev, DT)
    wt = w.permute(1, 2, 3, 0).contiguous().to(dev, DT)
    resd, mskd, yd = (nhwc(t).to(dev, DT).view(-1, CH) for t in (other, act, ysv))
    mud, isdd = mu.to(dev), isd.to(dev)

    def run():
        d = L.ConvDesc()
        d.dtype = L.qt_dtype(DT); d.mode = L.QT_CONV_DGRAD; d.batch = B
        d.in_h = d.in_w = H; d.out_h = d.out_w = H
        d.k_per_tap, d.n_out = CH, CH
        d.kh = d.kw = 3; d.stride = 1; d.pad = 1
        d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * CH, H * CH, CH
        y = torch.full((B * H * H, CH), float("nan"), dtype=DT, device=dev)
        io = L.ConvIO(L.ptr(dyd), L.ptr(wt), L.ptr(y), None, None, L.ptr(resd) if use_res else None,
                      L.ptr(mskd) if use_msk else None, None)
        rows = L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
        part = torch.zeros(rows, 2, CH, device=dev)
        if use_link:
            io.bn0_y, io.bn0_mean, io.bn0_invstd, io.bn0_partial = yd.data_ptr(), mud.data_ptr(), isdd.data_ptr(), part.data_ptr()
        L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
        torch.cuda.synchronize()
        return y, part
    a, b = _both(L, run)
    assert torch.equal(a[0], b[0]) and rel_err(a[1].sum(0).cpu(), b[1].sum(0).cpu()) <= 1e-5
    got = a[0].float().cpu().view(B, H, H, CH).permute(0, 3, 1, 2)
    assert rel_err(got, ref) <= TOL[DT]
    if use_link:
        gq = got.double()
        xhat = (ysv.double() - mu.double().view(1, -1, 1, 1)) * isd.double().view(1, -1, 1, 1)
        s = a[1].sum(0).cpu()
        assert rel_err(s[0], gq.sum((0, 2, 3))) <= 5e-3
        assert rel_err(s[1], (gq * xhat).sum((0, 2, 3))) <= 5e-3
