#!/bin/bash
# Run ON THE GPU BOX (through gpurun): kernel trace of scripts/bench_conv.py <mode>, median / min duration per kernel.
# Usage: scripts/trace_conv.sh <outdir under gpurun_out> <bench_conv mode>
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-trace_conv}
MODE=${2:-wgp}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/t" -- python3 $R/scripts/bench_conv.py $MODE > "$OUT/t.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/t/*/*_kernel_trace.csv")[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items()):
    if "at::native" in k: continue
    v.sort()
    print(f"{k:92s} n={len(v):3d} med={v[len(v)//2]:7.1f} min={v[0]:7.1f} us")
PY
grep -v amdgpu.ids "$OUT/t.log" || true
