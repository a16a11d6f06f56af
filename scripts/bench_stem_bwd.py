"""Developer aid: the one-launch stem backward (qt_stem_bn_bwd_wgrad) alone, B = 256 (QTCNN_STEM_BWD_FUSED=1|2 picks the form)."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from _util import pkg
L = pkg("_lib"); lib = L.lib()
dev = torch.device("cuda:0"); dt = torch.bfloat16; B = 256; qdt = L.qt_dtype(dt)
img = torch.randn(B, 3, 224, 224, device=dev)
xpad = torch.zeros(B, 230, 232, 4, dtype=dt, device=dev)
st = L.stream_ptr()
L.check(lib.qt_pack_stem_input(qdt, L.ptr(img), L.ptr(xpad), B, st), "pack")
y = torch.randn(B, 112, 112, 64, device=dev).to(dt)
dp = torch.randn(B, 56, 56, 64, device=dev).to(dt)
am = torch.randint(0, 9, (B, 56, 56, 64), dtype=torch.uint8, device=dev)
v = [torch.rand(64, device=dev) + 0.5 for _ in range(4)]
coef = torch.rand(3, 64, device=dev)
dw = torch.zeros(64, 7, 32, device=dev)
fn = lambda: L.check(lib.qt_stem_bn_bwd_wgrad(qdt, L.ptr(dp), L.ptr(am), L.ptr(y), L.ptr(v[0]), L.ptr(v[1]), L.ptr(v[2]), L.ptr(v[3]),
                                              L.ptr(coef), L.ptr(xpad), L.ptr(dw), B, st), "fused")
for _ in range(3): fn()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): fn()
b.record(); torch.cuda.synchronize()
print("QTCNN_STEM_BWD_FUSED=%s: %.1f us per launch" % (os.environ.get("QTCNN_STEM_BWD_FUSED", "default"), a.elapsed_time(b) / 20 * 1e3))
