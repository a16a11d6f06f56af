"""Developer aid: time single launches of the MFMA kernels through the C ABI (GPU box)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg
L = pkg("_lib")
dev = torch.device("cuda:0")

def desc(dt, mode, B, H, Cin, Cout, k, s, p):
    Ho = (H + 2 * p - k) // s + 1
    d = L.ConvDesc(); d.dtype = L.qt_dtype(dt); d.mode = mode; d.batch = B
    d.kh = d.kw = k; d.stride = s; d.pad = p
    if mode == L.QT_CONV_FWD:
        d.in_h = d.in_w = H; d.out_h = d.out_w = Ho; d.k_per_tap = Cin; d.n_out = Cout
        d.src_pix_stride = Cin; d.src_row_stride = H * Cin; d.src_img_stride = H * H * Cin
    else:
        d.in_h = d.in_w = Ho; d.out_h = d.out_w = H; d.k_per_tap = Cout; d.n_out = Cin
        d.src_pix_stride = Cout; d.src_row_stride = Ho * Cout; d.src_img_stride = Ho * Ho * Cout
    return d, Ho

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3  # us

def bench(kind, B, H, Cin, Cout, k, s, p, dt=torch.bfloat16, workspace=False):
    st = L.stream_ptr()
    if kind == "wgrad":
        d, Ho = desc(dt, L.QT_CONV_FWD, B, H, Cin, Cout, k, s, p)
        x = torch.randn(B, H, H, Cin, device=dev).to(dt); dy = torch.randn(B, Ho, Ho, Cout, device=dev).to(dt)
        dw = torch.zeros(Cout, k * k, Cin, device=dev)
        fn = lambda: L.check(L.lib().qt_conv2d_wgrad(ctypes.byref(d), L.ptr(dy), L.ptr(x), L.ptr(dw), st))
        if workspace:
            L.lib().qt_conv2d_wgrad_workspace_bytes.restype = ctypes.c_size_t
            nb = L.lib().qt_conv2d_wgrad_workspace_bytes(ctypes.byref(d))
            wsb = torch.empty(max(nb, 16), dtype=torch.uint8, device=dev)
            fn = lambda: L.check(L.lib().qt_conv2d_wgrad_ws(ctypes.byref(d), L.ptr(dy), L.ptr(x), L.ptr(dw), L.ptr(wsb),
                                                          ctypes.c_size_t(nb), st))
            kind = "wgradW"
    else:
        mode = L.QT_CONV_FWD if kind == "fwd" else L.QT_CONV_DGRAD
        d, Ho = desc(dt, mode, B, H, Cin, Cout, k, s, p)
        if kind == "fwd":
            x = torch.randn(B, H, H, Cin, device=dev).to(dt); w = torch.randn(Cout, k * k, Cin, device=dev).to(dt)
            y = torch.empty(B * Ho * Ho, Cout, device=dev, dtype=dt)
        else:
            x = torch.randn(B, Ho, Ho, Cout, device=dev).to(dt); w = torch.randn(Cin, k * k, Cout, device=dev).to(dt)
            y = torch.empty(B * H * H, Cin, device=dev, dtype=dt)
        io = L.ConvIO(L.ptr(x), L.ptr(w), L.ptr(y), None, None, None, None, None)
        fn = lambda: L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), st))
    us = timeit(fn)
    Ho = (H + 2 * p - k) // s + 1
    fl = 2.0 * B * Ho * Ho * Cout * Cin * k * k
    print(f"{kind:6s} B{B} H{H} {Cin:4d}->{Cout:4d} k{k} s{s}: {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s", flush=True)

def bench_s2(B, H, Cin, Cout, dt=torch.bfloat16, eval_epi=True, stats=False):
    """the fused stride-2 transition (csrc/conv_s2.hip) against the two generic launches it replaces"""
    st = L.stream_ptr()
    Ho = H // 2
    x = torch.randn(B, H, H, Cin, device=dev).to(dt)
    wc = (torch.randn(Cout, 9, Cin, device=dev) * 0.05).to(dt); wd = (torch.randn(Cout, Cin, device=dev) * 0.1).to(dt)
    y = torch.empty(B * Ho * Ho, Cout, device=dev, dtype=dt); yd = torch.empty_like(y)
    sc = torch.rand(Cout, device=dev) + 0.5; sh = torch.randn(Cout, device=dev) * 0.1
    d = L.ConvS2Desc(L.qt_dtype(dt), B, H, H, Cin, Cout, 1 if eval_epi else 0, 0)
    rows = L.lib().qt_conv_s2_pair_stats_rows(ctypes.byref(d))
    s1 = torch.zeros(rows, 2, Cout, device=dev); s2 = torch.zeros(rows, 2, Cout, device=dev)
    io = L.ConvS2IO(L.ptr(x), L.ptr(wc), L.ptr(wd), L.ptr(y), L.ptr(yd), L.ptr(sc) if eval_epi else None,
                    L.ptr(sh) if eval_epi else None, L.ptr(sc) if eval_epi else None, L.ptr(sh) if eval_epi else None,
                    L.ptr(s1) if stats else None, L.ptr(s2) if stats else None)
    us = timeit(lambda: L.check(L.lib().qt_conv_s2_pair(ctypes.byref(d), ctypes.byref(io), st)))
    d3, _ = desc(dt, L.QT_CONV_FWD, B, H, Cin, Cout, 3, 2, 1)
    d1, _ = desc(dt, L.QT_CONV_FWD, B, H, Cin, Cout, 1, 2, 0)
    d3.relu = 1 if eval_epi else 0
    io3 = L.ConvIO(L.ptr(x), L.ptr(wc), L.ptr(y), L.ptr(sc) if eval_epi else None, L.ptr(sh) if eval_epi else None, None, None, None)
    io1 = L.ConvIO(L.ptr(x), L.ptr(wd), L.ptr(yd), L.ptr(sc) if eval_epi else None, L.ptr(sh) if eval_epi else None, None, None, None)
    u3 = timeit(lambda: L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d3), ctypes.byref(io3), st)))
    u1 = timeit(lambda: L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d1), ctypes.byref(io1), st)))
    fl = 2.0 * B * Ho * Ho * Cout * Cin * 10
    mb = (x.numel() + 2 * y.numel()) * x.element_size() / 1e6
    print(f"s2pair B{B} H{H} {Cin:4d}->{Cout:4d} {'eval' if eval_epi else ('stats' if stats else 'raw')}: {us:7.1f} us "
          f"{fl / us / 1e6:7.1f} TFLOP/s {mb / us * 1e-0:6.2f} TB/s(alg {mb:.0f} MB) | generic 3x3 {u3:6.1f} + 1x1 {u1:6.1f} us", flush=True)


if __name__ == "__main__":
    B = 256
    which = sys.argv[1] if len(sys.argv) > 1 else "all"
    if which == "s2":
        for ev, stt in ((True, False), (False, True)):
            bench_s2(B, 56, 64, 128, eval_epi=ev, stats=stt)
            bench_s2(B, 28, 128, 256, eval_epi=ev, stats=stt)
            bench_s2(B, 14, 256, 512, eval_epi=ev, stats=stt)
        sys.exit(0)
    if which in ("all", "ksweep"):
        for cin in (64, 128, 256, 512):
            bench("fwd", B, 28, cin, 128, 3, 1, 1)
    if which in ("all", "layers"):
        for kind in ("fwd", "dgrad", "wgrad"):
            bench(kind, B, 56, 64, 64, 3, 1, 1)
            bench(kind, B, 28, 128, 128, 3, 1, 1)
            bench(kind, B, 14, 256, 256, 3, 1, 1)
            bench(kind, B, 7, 512, 512, 3, 1, 1)
            bench(kind, B, 56, 64, 128, 3, 2, 1)
            bench(kind, B, 56, 64, 128, 1, 2, 0)
    if which == "wgrad":
        for mw in (0, 14):   # 0: generic kernel, 14: streaming kernel wherever eligible
            L.lib().qt_set_wgrad_patch_min_width(mw)
            print("qt_set_wgrad_patch_min_width", mw, flush=True)
            bench("wgrad", B, 56, 64, 64, 3, 1, 1)
            bench("wgrad", B, 28, 128, 128, 3, 1, 1)
            bench("wgrad", B, 14, 256, 256, 3, 1, 1)
    if which == "wgrad1":   # streaming kernel only (QTCNN_WP_* env experiments)
        L.lib().qt_set_wgrad_patch_min_width(0)
        bench("wgrad", B, 7, 512, 512, 3, 1, 1)
        L.lib().qt_set_wgrad_patch_min_width(7)
        for w in (False, True):
            bench("wgrad", B, 7, 512, 512, 3, 1, 1, workspace=w)
            bench("wgrad", B, 56, 64, 64, 3, 1, 1, workspace=w)
            bench("wgrad", B, 28, 128, 128, 3, 1, 1, workspace=w)
            bench("wgrad", B, 14, 256, 256, 3, 1, 1, workspace=w)
    if which == "wgp":   # streaming weight-gradient kernel alone, partial-filter workspace (QTCNN_WP_VARIANT / PMC runs)
        L.lib().qt_set_wgrad_patch_min_width(7)
        bench("wgrad", B, 56, 64, 64, 3, 1, 1, workspace=True)
        bench("wgrad", B, 28, 128, 128, 3, 1, 1, workspace=True)
        bench("wgrad", B, 14, 256, 256, 3, 1, 1, workspace=True)
        bench("wgrad", B, 7, 512, 512, 3, 1, 1, workspace=True)
    if which == "igemm":   # forward / dgrad of the 3x3 stride-1 layers (QTCNN_IGEMM_VARIANT experiments)
        for kind in ("fwd", "dgrad"):
            bench(kind, B, 28, 128, 128, 3, 1, 1)
            bench(kind, B, 14, 256, 256, 3, 1, 1)
            bench(kind, B, 7, 512, 512, 3, 1, 1)
    if which == "l1generic":   # generic 128x64 tile on the layer1 shape (occupancy experiment)
        L.lib().qt_set_patch_conv(0)
        bench("fwd", B, 56, 64, 64, 3, 1, 1)
        bench("dgrad", B, 56, 64, 64, 3, 1, 1)
    if which == "wgrad4":   # generic weight-gradient kernel on the layer4 / stride-2 shapes (QTCNN_WGRAD_SB experiments)
        bench("wgrad", B, 7, 512, 512, 3, 1, 1)
        bench("wgrad", B, 14, 256, 512, 3, 2, 1)
        bench("wgrad", B, 28, 128, 256, 3, 2, 1)
        bench("wgrad", B, 14, 256, 128, 3, 1, 1)


def bench_ring(label, mode, residual=False, mask=False, link=False, H=56, C=64):
    """3x3 stride-1 C->C conv through qt_conv2d_igemm with the epilogue options the plan uses."""
    dt = torch.bfloat16
    d, Ho = desc(dt, mode, B, H, C, C, 3, 1, 1)
    st = L.stream_ptr()
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    w = torch.randn(C, 9, C, device=dev).to(dt)
    y = torch.empty(B * H * H, C, device=dev, dtype=dt)
    sc = torch.rand(C, device=dev) + 0.5
    sh = torch.randn(C, device=dev)
    res = torch.randn(B * H * H, C, device=dev).to(dt) if residual else None
    msk = torch.randn(B * H * H, C, device=dev).to(dt) if mask else None
    io = L.ConvIO(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(sc) if residual else None, L.ptr(sh) if residual else None,
                  L.ptr(res), L.ptr(msk), None)
    keep = []
    if link:
        by = torch.randn(B * H * H, C, device=dev).to(dt)
        mu, isd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        rows = L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
        part = torch.zeros(rows + 64, 2, C, device=dev)
        io.bn0_y, io.bn0_mean, io.bn0_invstd, io.bn0_partial = by.data_ptr(), mu.data_ptr(), isd.data_ptr(), part.data_ptr()
        keep += [by, mu, isd, part]
    d.relu = 1 if residual else 0
    fn = lambda: L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), st))
    us = timeit(fn)
    print(f"ring {label:34s}: {us:8.1f} us", flush=True)


if __name__ == "__main__" and which == "ring":
    bench_ring("fwd plain", L.QT_CONV_FWD)
    bench_ring("fwd + scale/shift + residual + relu", L.QT_CONV_FWD, residual=True)
    bench_ring("dgrad plain", L.QT_CONV_DGRAD)
    bench_ring("dgrad + relu mask", L.QT_CONV_DGRAD, mask=True)
    bench_ring("dgrad + relu mask + bn link", L.QT_CONV_DGRAD, mask=True, link=True)
    bench_ring("dgrad + residual + mask + bn link", L.QT_CONV_DGRAD, residual=True, mask=True, link=True)

if __name__ == "__main__" and which == "epi":   # generic implicit GEMM, layer2 / layer3 shapes, with epilogue operands
    for (H, C) in ((28, 128), (14, 256)):
        print("H", H, "C", C)
        bench_ring("fwd plain", L.QT_CONV_FWD, H=H, C=C)
        bench_ring("fwd + scale/shift + residual + relu", L.QT_CONV_FWD, residual=True, H=H, C=C)
        bench_ring("dgrad plain", L.QT_CONV_DGRAD, H=H, C=C)
        bench_ring("dgrad + relu mask + bn link", L.QT_CONV_DGRAD, mask=True, link=True, H=H, C=C)
        bench_ring("dgrad + residual + mask + bn link", L.QT_CONV_DGRAD, residual=True, mask=True, link=True, H=H, C=C)

if __name__ == "__main__" and which == "pt":   # patch-resident ping-pong kernel (conv_pt.hip) against the generic one, same process, interleaved rounds
    shapes = [("fwd", 28, 128, 128), ("fwd", 14, 256, 256), ("fwd", 7, 512, 512),
              ("dgrad", 28, 128, 128), ("dgrad", 14, 256, 256), ("dgrad", 7, 512, 512)]
    import io as _io, contextlib
    for rnd in range(2):
        for on in (1, 0):
            L.lib().qt_set_pt_conv(on)
            print(f"-- round {rnd} qt_set_pt_conv({on})", flush=True)
            for kind, H, ci, co in shapes:
                bench(kind, B, H, ci, co, 3, 1, 1)
    L.lib().qt_set_pt_conv(1)
    for (H, C) in ((28, 128), (14, 256), (7, 512)):
        print("epilogues, pt kernel, H", H, "C", C)
        bench_ring("fwd plain", L.QT_CONV_FWD, H=H, C=C)
        bench_ring("fwd + scale/shift + residual + relu", L.QT_CONV_FWD, residual=True, H=H, C=C)
        bench_ring("dgrad + relu mask + bn link", L.QT_CONV_DGRAD, mask=True, link=True, H=H, C=C)
        bench_ring("dgrad + residual + mask + bn link", L.QT_CONV_DGRAD, residual=True, mask=True, link=True, H=H, C=C)

if __name__ == "__main__" and which == "pt1":   # few launches of the three stage shapes, patch-resident kernel on (PMC runs)
    L.lib().qt_set_pt_conv(1)
    for H, C in ((28, 128), (14, 256), (7, 512)):
        dt = torch.bfloat16
        d, Ho = desc(dt, L.QT_CONV_FWD, B, H, C, C, 3, 1, 1)
        x = torch.randn(B, H, H, C, device=dev).to(dt); w = torch.randn(C, 9, C, device=dev).to(dt)
        y = torch.empty(B * H * H, C, device=dev, dtype=dt)
        io = L.ConvIO(L.ptr(x), L.ptr(w), L.ptr(y), None, None, None, None, None)
        for _ in range(5):
            L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()))
        torch.cuda.synchronize()

if __name__ == "__main__" and which == "pt3":   # forward of the three stage shapes with the patch-resident kernel (QTCNN_PT_DBG experiments)
    L.lib().qt_set_pt_conv(1)
    for _ in range(2):
        bench("fwd", B, 28, 128, 128, 3, 1, 1)
        bench("fwd", B, 14, 256, 256, 3, 1, 1)
        bench("fwd", B, 7, 512, 512, 3, 1, 1)
if __name__ == "__main__" and which == "wg3":   # tile-resident weight-gradient kernel on the three NT = 8 shapes (the NT = 8 instantiation)
    L.lib().qt_set_wgrad_patch_min_width(7)
    bench("wgrad", B, 28, 128, 128, 3, 1, 1, workspace=True)
    bench("wgrad", B, 14, 256, 256, 3, 1, 1, workspace=True)
    bench("wgrad", B, 7, 512, 512, 3, 1, 1, workspace=True)
if __name__ == "__main__" and which == "trans":   # stride-2 transition convs and their 1x1 downsamples, alone (forward / dgrad / wgrad)
    for kind in ("fwd", "dgrad", "wgrad"):
        bench(kind, B, 56, 64, 128, 3, 2, 1)
        bench(kind, B, 56, 64, 128, 1, 2, 0)
        bench(kind, B, 28, 128, 256, 3, 2, 1)
        bench(kind, B, 28, 128, 256, 1, 2, 0)
        bench(kind, B, 14, 256, 512, 3, 2, 1)
        bench(kind, B, 14, 256, 512, 1, 2, 0)
