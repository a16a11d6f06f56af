#!/usr/bin/env python3
"""Developer aid: timeline of the LAST eval forward in a rocprofv3 kernel trace (gpurun_out/<name>/trace/*/*_kernel_trace.csv):
each kernel with queue, start offset, duration and the idle gap in front of it on its queue."""
import csv
import glob
import os
import re
import sys

name = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = glob.glob(os.path.join(root, "gpurun_out", name, "trace", "*", "*_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
starts = [i for i, r in enumerate(rows) if "conv_stem_pool" in r["Kernel_Name"] or "stem_conv_pool" in r["Kernel_Name"]]
lo = starts[-1]
while lo > 0 and rows[lo]["s"] - rows[lo - 1]["e"] < 20000 and "linear" not in rows[lo - 1]["Kernel_Name"] and "gemm_small_wave" not in rows[lo - 1]["Kernel_Name"]:
    lo -= 1
step = rows[lo:]
t0 = step[0]["s"]
last = {}
tot = 0
for r in step:
    n = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"])
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)[:70]
    q = r["Queue_Id"]
    gap = (r["s"] - last[q]) / 1e3 if q in last else 0.0
    last[q] = r["e"]
    tot += r["e"] - r["s"]
    print(f"q{q} {(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f}  gap {gap:6.1f}  {n}")
print(f"wall {(max(r['e'] for r in step) - t0) / 1e3:.1f} us, kernel sum {tot / 1e3:.1f} us, {len(step)} launches")
