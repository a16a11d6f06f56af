#!/bin/bash
# Run ON THE GPU BOX (through gpurun) from the repo root: one rocprofv3 PMC pass (no trace domain besides --kernel-trace) of
# the bench command with the MFMA-busy counter, for the train step and for the eval forward.  SQ_VALU_MFMA_BUSY_CYCLES sums
# the busy cycles of the matrix pipes of all SIMDs; scripts/summarize_mfma_busy.py turns it into executed MFMA work per kernel family.
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-mfma}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d "$OUT/train" -- \
    python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 > "$OUT/train.log" 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d "$OUT/eval" -- \
    python3 $R/bench.py --forward-only --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 > "$OUT/eval.log" 2>&1
echo "PMC passes written under $OUT"
