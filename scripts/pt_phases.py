"""Developer aid: where a conv_pt launch spends its time (s_memrealtime stamps per workgroup: qt_set_pt_prof).
Needs the experiment build: bash scripts/prof_build.sh, then QTCNN_LIB_PATH=<pkg>/libqtcnn_prof.so (the production library has no stamps)."""
import ctypes, os, sys
sys.argv = [sys.argv[0], "none"]
import importlib.util
import torch
here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("bc", os.path.join(here, "bench_conv.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc); bc.B = 256
L = bc.L
dev = bc.dev
prof = torch.zeros(1024, 32, dtype=torch.int64, device=dev)

def run(label, mode, H, C, **kw):
    dt = torch.bfloat16
    B = 256
    d, Ho = bc.desc(dt, mode, B, H, C, C, 3, 1, 1)
    st = L.stream_ptr()
    x = torch.randn(B, H, H, C, device=dev).to(dt)
    w = torch.randn(C, 9, C, device=dev).to(dt)
    y = torch.empty(B * H * H, C, device=dev, dtype=dt)
    res = torch.randn(B * H * H, C, device=dev).to(dt) if kw.get("residual") else None
    sc = torch.rand(C, device=dev) + 0.5
    io = L.ConvIO(L.ptr(x), L.ptr(w), L.ptr(y), L.ptr(sc) if kw.get("affine") else None, L.ptr(sc) if kw.get("affine") else None,
                  L.ptr(res), None, None)
    keep = []
    if kw.get("link"):
        by = torch.randn(B * H * H, C, device=dev).to(dt)
        mu, isd = torch.randn(C, device=dev), torch.rand(C, device=dev) + 0.5
        rows = L.lib().qt_conv2d_stats_rows(ctypes.byref(d))
        part = torch.zeros(rows + 64, 2, C, device=dev)
        io.bn0_y, io.bn0_mean, io.bn0_invstd, io.bn0_partial = by.data_ptr(), mu.data_ptr(), isd.data_ptr(), part.data_ptr()
        bits = torch.randint(0, 255, (B * H * H, C // 8), dtype=torch.uint8, device=dev)
        io.relu_mask_bits = bits.data_ptr()
        keep += [by, mu, isd, part, bits]
    d.relu = 1 if kw.get("affine") else 0
    fn = lambda: L.check(L.lib().qt_conv2d_igemm(ctypes.byref(d), ctypes.byref(io), st))
    us = bc.timeit(fn)
    L.lib().qt_set_pt_prof(ctypes.c_void_p(prof.data_ptr()))
    prof.zero_()
    fn(); torch.cuda.synchronize()
    L.lib().qt_set_pt_prof(None)
    p = prof.cpu().double() * 0.01  # us
    used = p[:, 31] > 0
    p = p[used]
    t0 = p[:, 31].min()
    entry = p[:, 31] - t0
    n_items = int(((p[:, :31] > 0).sum(1).max().item() - 1) // 3)
    line = f"{label:44s} {us:6.1f} us | entry spread {entry.max():5.1f} | prologue {(p[:,0]-p[:,31]).mean():5.2f}"
    for k in range(n_items):
        prev = p[:, 0] if k == 0 else p[:, 3 * k]
        line += f" | item{k}: K {(p[:,1+3*k]-prev).mean():5.2f} wait {(p[:,2+3*k]-p[:,1+3*k]).mean():4.2f} epi {(p[:,3+3*k]-p[:,2+3*k]).mean():5.2f}"
    line += f" | last end {(p[:, 3 * n_items].max() - t0):6.1f}"
    print(line, flush=True)

for H, C in ((28, 128), (14, 256), (7, 512)):
    run(f"fwd plain {H}x{H}x{C}", L.QT_CONV_FWD, H, C)
    run(f"fwd affine+res+relu {H}x{H}x{C}", L.QT_CONV_FWD, H, C, affine=True, residual=True)
    run(f"dgrad plain {H}x{H}x{C}", L.QT_CONV_DGRAD, H, C)
    run(f"dgrad bits+link {H}x{H}x{C}", L.QT_CONV_DGRAD, H, C, link=True)
    run(f"dgrad res+bits+link {H}x{H}x{C}", L.QT_CONV_DGRAD, H, C, link=True, residual=True)
