import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_conv as b
kind, H, cin, cout = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
if len(sys.argv) > 5 and sys.argv[5] == "patch":
    b.L.lib().qt_set_patch_conv(1)
for _ in range(3):
    b.bench(kind, 256, H, cin, cout, 3, 1, 1)
