"""Developer aid: which epilogue operand of a data-gradient launch costs what (3x3 layers of layer1-4, alone, through the C ABI)."""
import sys, os
sys.argv = [sys.argv[0], "none"]
import importlib.util
here = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("bc", os.path.join(here, "bench_conv.py"))
bc = importlib.util.module_from_spec(spec); spec.loader.exec_module(bc); bc.B = 256
for H, C in ((56, 64), (28, 128), (14, 256), (7, 512)):
    print("H", H, "C", C)
    for _ in range(2):
        bc.bench_ring("dgrad plain", bc.L.QT_CONV_DGRAD, H=H, C=C)
        bc.bench_ring("dgrad + link", bc.L.QT_CONV_DGRAD, link=True, H=H, C=C)
        bc.bench_ring("dgrad + mask + link", bc.L.QT_CONV_DGRAD, mask=True, link=True, H=H, C=C)
        bc.bench_ring("dgrad + residual + link", bc.L.QT_CONV_DGRAD, residual=True, link=True, H=H, C=C)
        bc.bench_ring("dgrad + residual + mask + link", bc.L.QT_CONV_DGRAD, residual=True, mask=True, link=True, H=H, C=C)
