"""Developer aid: what the fused epilogues (scale/shift + residual + ReLU; ReLU mask + BatchNorm links) cost on the 3x3 layers
of layer2-4, launched alone through the C ABI (GPU box)."""
import sys, os
sys.argv = [sys.argv[0], "none"]
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import importlib.util
spec = importlib.util.spec_from_file_location("bc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "bench_conv.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc); bc.B = 256
for H, C in ((28, 128), (14, 256), (7, 512)):
    print("H", H, "C", C)
    bc.bench_ring("fwd plain", bc.L.QT_CONV_FWD, H=H, C=C)
    bc.bench_ring("fwd + scale/shift + residual + relu", bc.L.QT_CONV_FWD, residual=True, H=H, C=C)
    bc.bench_ring("dgrad plain", bc.L.QT_CONV_DGRAD, H=H, C=C)
    bc.bench_ring("dgrad + relu mask + bn link", bc.L.QT_CONV_DGRAD, mask=True, link=True, H=H, C=C)
    bc.bench_ring("dgrad + residual + mask + bn link", bc.L.QT_CONV_DGRAD, residual=True, mask=True, link=True, H=H, C=C)
