#!/usr/bin/env python3
"""Developer aid: aggregate the output of `scripts/timeline.py <name> -v` per (queue, kernel)."""
import collections
import re
import sys

agg = collections.defaultdict(lambda: [0, 0.0])
for line in open(sys.argv[1]):
    m = re.match(r"q(\d)\s+([\d.]+)\s+([\d.]+)\s+(.*?)\s+grid=", line)
    if not m:
        continue
    key = (m.group(1), m.group(4)[:45])
    agg[key][0] += 1
    agg[key][1] += float(m.group(3))
for (q, n), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(f"q{q} {t:8.1f} us {c:3d}x  {n}")
