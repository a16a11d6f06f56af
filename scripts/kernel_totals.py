#!/usr/bin/env python3
"""Developer aid: per-kernel totals of a rocprofv3 kernel trace directory (gpurun_out/<dir>), per step (5 steps traced by
scripts/collect_profiles.sh's command: 2 warm-up + 3 timed)."""
import collections
import csv
import glob
import os
import re
import sys

steps = 5
for d in sys.argv[1:]:
    f = max(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True), key=os.path.getmtime)   # (the latest run in the directory)
    agg = collections.defaultdict(lambda: [0, 0])
    for r in csv.DictReader(open(f)):
        n = re.sub(r"\(anonymous namespace\)::|_ZN12_GLOBAL__N_1\d+", "", r["Kernel_Name"])
        n = re.sub(r"\(.*", "", n)
        agg[n][0] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        agg[n][1] += 1
    print(d)
    tot = sum(v[0] for v in agg.values())
    for n, (t, c) in sorted(agg.items(), key=lambda x: -x[1][0])[: int(40)]:
        print(f"  {t / steps / 1000:9.1f} us/step  {c / steps:5.1f} x {t / c / 1000:8.1f} us  {n[:110]}")
    print(f"  total {tot / steps / 1000:.1f} us/step")
