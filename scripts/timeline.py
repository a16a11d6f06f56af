#!/usr/bin/env python3
"""Developer aid: per-queue timeline of the LAST training step in a rocprofv3 kernel trace
(gpurun_out/<name>/trace/*/*_kernel_trace.csv).  Prints, in start order, each kernel with its
queue, start offset and duration, then the busy time per queue and the wall time of the step."""
import csv
import glob
import os
import re
import sys

name = sys.argv[1] if len(sys.argv) > 1 else "tl"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
f = glob.glob(os.path.join(root, "gpurun_out", name, "trace", "*", "*_kernel_trace.csv"))[0]
rows = [r for r in csv.DictReader(open(f))]
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
# a step starts at the stem input pack
starts = [i for i, r in enumerate(rows) if "pack_stem_input" in r["Kernel_Name"]]
lo = starts[-1]
step = rows[lo:]
if len(starts) > 1:
    prev = rows[starts[-2]:lo]
    print(f"previous step wall: {(rows[lo]['s'] - prev[0]['s']) / 1e3:.1f} us")
t0 = step[0]["s"]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return n[:60]


busy = {}
for r in step:
    q = r["Queue_Id"]
    busy[q] = busy.get(q, 0) + r["e"] - r["s"]
    if "-v" in sys.argv:
        print(f"q{q} {(r['s'] - t0) / 1e3:9.1f} {(r['e'] - r['s']) / 1e3:8.1f}  {short(r['Kernel_Name'])}  grid={r['Grid_Size_X']}")
print("busy per queue (us):", {q: round(b / 1e3, 1) for q, b in busy.items()})
print(f"last step wall (to last kernel end): {(max(r['e'] for r in step) - t0) / 1e3:.1f} us")
