#!/usr/bin/env python3
"""Developer aid (GPU box): the six stride-2 weight gradients of the ResNet-18 transition blocks alone, B = 256, bf16:
conv_wgrad_s2.hip (+ its partial-filter sum) against the generic kernel (+ its zero fill), microseconds per call."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg  # noqa: E402

L = pkg("_lib")
lib = L.lib()
dev = torch.device("cuda:0")
dt = torch.bfloat16
B = int(os.environ.get("B", "256"))
lib.qt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return 1e3 * a.elapsed_time(b) / n


for Cin, Cout, H, k in ((64, 128, 56, 3), (128, 256, 28, 3), (256, 512, 14, 3), (64, 128, 56, 1), (128, 256, 28, 1), (256, 512, 14, 1)):
    Ho = H // 2
    x = torch.randn(B, H, H, Cin, device=dev).to(dt)
    dy = torch.randn(B, Ho, Ho, Cout, device=dev).to(dt)
    d = L.ConvDesc()
    d.dtype, d.mode, d.batch = L.qt_dtype(dt), L.QT_CONV_FWD, B
    d.in_h = d.in_w = H
    d.out_h = d.out_w = Ho
    d.k_per_tap, d.n_out, d.kh, d.kw, d.stride, d.pad = Cin, Cout, k, k, 2, 1 if k == 3 else 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = H * H * Cin, H * Cin, Cin
    g = torch.empty(Cout, Cin, k, k, dtype=torch.float32, device=dev)
    res = {}
    for on in (1, 0):
        lib.qt_set_wgrad_s2(on)
        nbytes = lib.qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
        if nbytes:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            fn = lambda: L.check(lib.qt_conv2d_wgrad_oihw(ctypes.byref(d), L.ptr(dy), L.ptr(x), L.ptr(g), L.ptr(ws),
                                                          ctypes.c_size_t(nbytes), L.stream_ptr()), "wgrad_oihw")
        else:
            dw = torch.empty(Cout, k * k, Cin, dtype=torch.float32, device=dev)

            def fn():
                dw.zero_()
                L.check(lib.qt_conv2d_wgrad(ctypes.byref(d), L.ptr(dy), L.ptr(x), L.ptr(dw), L.stream_ptr()), "wgrad")
        res[on] = timed(fn)
    lib.qt_set_wgrad_s2(-1)
    gf = 2.0 * B * Ho * Ho * Cin * Cout * k * k / 1e9
    print(f"{Cin:4d}->{Cout:4d} {H}x{H} k{k}: parity-plane kernel {res[1]:7.1f} us ({gf / res[1] * 1e3:6.0f} TFLOP/s)   generic {res[0]:7.1f} us")
