"""The ReLU mask of the data-gradient epilogues as one bit per element (qt_conv_io.relu_mask_bits, written by
qt_bn_act_mask): every kernel family that takes a `relu_mask` tensor must give bit-identical results with the packed form.

Reference behaviour: ReLU backward passes the gradient where the activation is > 0 (torch.relu autograd inside
`loss.backward()`, /root/reference/Quadtree_from scratch/Quadtree_train.py:65; the activations are those of torchvision's
BasicBlock as built by /root/reference/Quadtree_from scratch/models.py:222-243)."""
import ctypes

import pytest
import torch

from _util import pkg

pytestmark = pytest.mark.gpu


def _dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def pack_bits(mask_bool):
    """[M][C] bool -> [M][C/8] uint8, bit (c & 7) of byte c / 8"""
    M, C = mask_bool.shape
    w = (1 << torch.arange(8, device=mask_bool.device, dtype=torch.int32)).view(1, 1, 8)
    return (mask_bool.view(M, C // 8, 8).to(torch.int32) * w).sum(-1).to(torch.uint8).contiguous()


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", [(3, 64, 56, False), (17, 128, 28, True), (1, 256, 14, True)])
def test_bn_act_mask_writes_the_sign_of_its_output(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, C, H, with_res = cfg
    g = torch.Generator().manual_seed(5)
    M = B * H * H
    y = torch.randn(M, C, generator=g).to(dev, dt)
    res = torch.randn(M, C, generator=g).to(dev, dt) if with_res else None
    sc = (torch.rand(C, generator=g) + 0.5).to(dev)
    sh = (torch.randn(C, generator=g) * 0.3).to(dev)
    out0 = torch.empty_like(y)
    out1 = torch.empty_like(y)
    bits = torch.full((M, C // 8), 0xAA, dtype=torch.uint8, device=dev)
    st = L.stream_ptr()
    L.check(lib.qt_bn_act(L.qt_dtype(dt), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(res), None, None, 1, L.ptr(out0),
                          ctypes.c_longlong(M), C, st), "qt_bn_act")
    L.check(lib.qt_bn_act_mask(L.qt_dtype(dt), L.ptr(y), L.ptr(sc), L.ptr(sh), L.ptr(res), None, None, 1, L.ptr(out1),
                               L.ptr(bits), ctypes.c_longlong(M), C, st), "qt_bn_act_mask")
    torch.cuda.synchronize()
    assert torch.equal(out0, out1)
    # a positive f32 value never rounds to a bf16 zero (same exponent range), so the bit is the sign of the stored value
    assert torch.equal(bits, pack_bits(out1.float() > 0))
    ref = torch.relu(y.float() * sc + sh + (res.float() if with_res else 0))
    assert torch.allclose(out1.float(), ref.to(dt).float(), rtol=1e-2 if dt == torch.bfloat16 else 1e-6, atol=1e-6)


def _dgrad_desc(L, dt, B, H, Cin, Cout):
    d = L.ConvDesc()
    d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_DGRAD; d.batch = B
    d.in_h = d.in_w = H; d.out_h = d.out_w = H
    d.k_per_tap, d.n_out = Cout, Cin
    d.kh = d.kw = 3; d.stride = 1; d.pad = 1
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * Cout, H * Cout, Cout
    return d


# (batch, channels, H, which kernel the shape takes with the defaults)
STRIDE1 = [(16, 64, 56, "conv_l1_ring"), (2, 64, 56, "conv_patch / generic"), (17, 128, 28, "conv_pt rows"),
           (16, 256, 14, "conv_pt 256-channel tile"), (20, 512, 7, "conv_pt stacked 7x7"), (3, 128, 28, "generic tile")]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", STRIDE1, ids=[c[3] for c in STRIDE1])
def test_stride1_dgrad_bits_equal_tensor_mask(dt, cfg):
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, C, H, _ = cfg
    g = torch.Generator().manual_seed(21)
    w = (torch.randn(C, 9, C, generator=g) * (2.0 / (C * 9)) ** 0.5).to(dev, dt)
    dy = torch.randn(B * H * H, C, generator=g).to(dev, dt)
    res = torch.randn(B * H * H, C, generator=g).to(dev, dt)
    act = torch.relu(torch.randn(B * H * H, C, generator=g)).to(dev, dt)
    by = torch.randn(B * H * H, C, generator=g).to(dev, dt)
    mu = (torch.randn(C, generator=g) * 0.2).to(dev)
    isd = (torch.rand(C, generator=g) + 0.5).to(dev)
    bits = pack_bits(act.float() > 0)
    d = _dgrad_desc(L, dt, B, H, C, C)
    rows = lib.qt_conv2d_stats_rows(ctypes.byref(d))
    outs = []
    for use_bits in (False, True):
        o = torch.full((B * H * H, C), float("nan"), dtype=dt, device=dev)
        part = torch.zeros(rows, 2, C, device=dev)
        io = L.ConvIO(L.ptr(dy), L.ptr(w), L.ptr(o), None, None, L.ptr(res), None if use_bits else L.ptr(act), None,
                      L.ptr(by), L.ptr(mu), L.ptr(isd), L.ptr(part), None, None, None, None,
                      L.ptr(bits) if use_bits else None)
        L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
        torch.cuda.synchronize()
        outs.append((o, part))
    assert not torch.isnan(outs[0][0].float()).any()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
    # and the mask does something: masked-out positions are exact zeros
    assert (outs[1][0].float()[act.float() <= 0] == 0).all()
    # both forms at once are refused
    io = L.ConvIO(L.ptr(dy), L.ptr(w), L.ptr(outs[0][0]), None, None, None, L.ptr(act), None, None, None, None, None, None,
                  None, None, None, L.ptr(bits))
    assert lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()) != 0


MERGED = [(2, 64, 128, 56, "generic"), (16, 128, 256, 28, "generic, full tiles"), (20, 256, 512, 14, "conv_pt four-tap")]


@pytest.mark.parametrize("dt", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cfg", MERGED, ids=[c[4] for c in MERGED])
def test_merged_stride2_dgrad_bits_equal_tensor_mask(dt, cfg):
    """the 2x2-tap gather with four parity classes (qt_conv_desc.dst_merge): the mask is read at the scattered destination"""
    dev = _dev()
    L = pkg("_lib")
    lib = L.lib()
    B, Cin, Cout, H, _ = cfg
    Ho = H // 2
    g = torch.Generator().manual_seed(33)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) * (2.0 / (Cout * 9)) ** 0.5)
    dy = torch.randn(B * Ho * Ho, Cout, generator=g).to(dev, dt)
    res = torch.randn(B * H * H, Cin, generator=g).to(dev, dt)
    act = torch.relu(torch.randn(B * H * H, Cin, generator=g)).to(dev, dt)
    by = torch.randn(B * H * H, Cin, generator=g).to(dev, dt)
    mu = (torch.randn(Cin, generator=g) * 0.2).to(dev)
    isd = (torch.rand(Cin, generator=g) + 0.5).to(dev)
    bits = pack_bits(act.float() > 0)
    wd = torch.empty(16 * Cout * Cin, dtype=dt, device=dev)
    w_dev = w.to(dev).contiguous()
    L.check(lib.qt_pack_dgrad_s2_merged(L.qt_dtype(dt), L.ptr(w_dev), L.ptr(wd), Cout, Cin, L.stream_ptr()), "pack")
    d = L.ConvDesc()
    d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_FWD; d.batch = B
    d.in_h = d.in_w = Ho; d.out_h = d.out_w = Ho
    d.k_per_tap, d.n_out = Cout, 4 * Cin
    d.kh = d.kw = 2; d.stride = 1; d.pad = 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = Ho * Ho * Cout, Ho * Cout, Cout
    d.dst_sub = 2; d.dst_h = d.dst_w = H; d.dst_merge = Cin
    rows = lib.qt_conv2d_stats_rows(ctypes.byref(d))
    outs = []
    for use_bits in (False, True):
        o = torch.full((B * H * H, Cin), float("nan"), dtype=dt, device=dev)
        part = torch.zeros(rows, 2, Cin, device=dev)
        io = L.ConvIO(L.ptr(dy), L.ptr(wd), L.ptr(o), None, None, L.ptr(res), None if use_bits else L.ptr(act), None,
                      L.ptr(by), L.ptr(mu), L.ptr(isd), L.ptr(part), None, None, None, None,
                      L.ptr(bits) if use_bits else None)
        L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()), "qt_conv2d_igemm")
        torch.cuda.synchronize()
        outs.append((o, part))
    assert not torch.isnan(outs[0][0].float()).any()
    assert torch.equal(outs[0][0], outs[1][0])
    assert torch.equal(outs[0][1], outs[1][1])
