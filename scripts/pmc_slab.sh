#!/bin/bash
# Developer aid (GPU box): PMC passes of the Quadtree3DCNN eval forward for the slab-resident conv3d_block2 kernel
# (csrc/conv3d_slab.hip): matrix-pipe busy, LDS activity / bank conflicts / waits.
#   gpurun -- 'bash scripts/pmc_slab.sh tag'  ->  gpurun_out/<tag>/{a,b}/...counter_collection.csv
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${1:-pmcslab}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d "$OUT/a" -- \
    python3 $R/bench.py --model quadtree3d --forward-only --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$OUT/a.log" 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d "$OUT/b" -- \
    python3 $R/bench.py --model quadtree3d --forward-only --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 > "$OUT/b.log" 2>&1
echo done
