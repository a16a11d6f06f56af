#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root.  Three separate rocprofv3 passes of the
# same bench command, as MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE do not fit
# one pass; PMC passes carry no trace domain besides --kernel-trace):
#   1. --kernel-trace --stats          -> per-kernel durations
#   2. --pmc FETCH_SIZE                -> read bytes  (x2 on gfx950 for wide coalesced reads)
#   3. --pmc WRITE_SIZE                -> written bytes
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-profile}
# QT_PROFILE_ARGS: extra bench.py arguments (e.g. "--model quadtree3d"); default = the headline QuadtreeCNN train step
CMD="python3 $R/bench.py ${QT_PROFILE_ARGS} --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$OUT/fetch" -- $CMD > "$OUT/fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$OUT/write" -- $CMD > "$OUT/write.log" 2>&1
echo "profiles written under $OUT"
