"""Developer aid: where the generic implicit-GEMM tile spends its time on the stride-2 data gradients of the backward chain
(s_memrealtime stamps per workgroup: qt_set_igemm_prof).
Needs the experiment build: bash scripts/prof_build.sh, then QTCNN_LIB_PATH=<pkg>/libqtcnn_prof.so (the production library has no stamps)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg
L = pkg("_lib")
lib = L.lib()
dev = torch.device("cuda:0")
dt = torch.bfloat16
B = 256

def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3

def report(label, fn, nwg_max=8192):
    us = timeit(fn)
    prof = torch.zeros(nwg_max, 4, dtype=torch.int64, device=dev)
    lib.qt_set_igemm_prof(ctypes.c_void_p(prof.data_ptr()))
    fn(); torch.cuda.synchronize()
    lib.qt_set_igemm_prof(None)
    p = prof.cpu().double() * 0.01
    p = p[p[:, 0] > 0]
    t0 = p[:, 0].min()
    print(f"{label:40s} {us:6.1f} us | {len(p):5d} workgroups | entry spread {(p[:,0]-t0).max():6.1f} | addressing {(p[:,1]-p[:,0]).mean():5.2f}"
          f" | K loop {(p[:,2]-p[:,1]).mean():5.2f} | epilogue {(p[:,3]-p[:,2]).mean():5.2f} | workgroup {(p[:,3]-p[:,0]).mean():5.2f} | last end {(p[:,3].max()-t0):6.1f}", flush=True)

def merged(Cin, Cout, H, links=1, res=True, bits=True):
    Ho = H // 2
    g = torch.Generator().manual_seed(1)
    w = torch.randn(Cout, Cin, 3, 3, generator=g).to(dev)
    wd = torch.empty(16 * Cout * Cin, dtype=dt, device=dev)
    L.check(lib.qt_pack_dgrad_s2_merged(L.qt_dtype(dt), L.ptr(w), L.ptr(wd), Cout, Cin, L.stream_ptr()), "pack")
    dy = torch.randn(B * Ho * Ho, Cout, device=dev).to(dt)
    o = torch.empty(B * H * H, Cin, dtype=dt, device=dev)
    r = torch.randn(B * H * H, Cin, device=dev).to(dt)
    by = [torch.randn(B * H * H, Cin, device=dev).to(dt) for _ in range(2)]
    mu, isd = torch.randn(Cin, device=dev), torch.rand(Cin, device=dev) + 0.5
    mb = torch.randint(0, 255, (B * H * H, Cin // 8), dtype=torch.uint8, device=dev)
    d = L.ConvDesc()
    d.dtype = L.qt_dtype(dt); d.mode = L.QT_CONV_FWD; d.batch = B
    d.in_h = d.in_w = Ho; d.out_h = d.out_w = Ho
    d.k_per_tap, d.n_out = Cout, 4 * Cin
    d.kh = d.kw = 2; d.stride = 1; d.pad = 0
    d.src_img_stride, d.src_row_stride, d.src_pix_stride = Ho * Ho * Cout, Ho * Cout, Cout
    d.dst_sub = 2; d.dst_h = d.dst_w = H; d.dst_merge = Cin; d.dst_merge_res0 = 1
    rows = lib.qt_conv2d_stats_rows(ctypes.byref(d))
    parts = [torch.zeros(rows + 64, 2, Cin, device=dev) for _ in range(2)]
    io = L.ConvIO(L.ptr(dy), L.ptr(wd), L.ptr(o), None, None, L.ptr(r) if res else None, None, None)
    if links >= 1:
        io.bn0_y, io.bn0_mean, io.bn0_invstd, io.bn0_partial = by[0].data_ptr(), mu.data_ptr(), isd.data_ptr(), parts[0].data_ptr()
    if links >= 2:
        io.bn1_y, io.bn1_mean, io.bn1_invstd, io.bn1_partial = by[1].data_ptr(), mu.data_ptr(), isd.data_ptr(), parts[1].data_ptr()
    if bits: io.relu_mask_bits = mb.data_ptr()
    keep = (w, wd, dy, o, r, by, mu, isd, mb, parts)
    return (lambda: L.check(lib.qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), L.stream_ptr()))), keep

lib.qt_set_pt_conv(0)   # (the 7x7 map would take conv_pt)
for Cin, Cout, H in ((64, 128, 56), (128, 256, 28), (256, 512, 14)):
    fn, keep = merged(Cin, Cout, H, links=0, res=False, bits=False)
    report(f"merged dgrad {H}x{H}x{Cin} plain", fn)
    fn, keep = merged(Cin, Cout, H, links=1)
    report(f"merged dgrad {H}x{H}x{Cin} res0+bits+link", fn)
    fn, keep = merged(Cin, Cout, H, links=2)
    report(f"merged dgrad {H}x{H}x{Cin} res0+bits+2 links", fn)
