#!/bin/bash
# Developer aid (GPU box): rocprofv3 kernel stats of the Quadtree3DCNN train step and eval forward (BASELINE config 4)
#   gpurun -- 'bash scripts/trace_q3d.sh r04q'  -> gpurun_out/<tag>_train/..., gpurun_out/<tag>_eval/...
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-q3d}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_train -o t -- python3 $R/bench.py --model quadtree3d --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 > $R/gpurun_out/${TAG}_train.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${TAG}_eval -o e -- python3 $R/bench.py --model quadtree3d --forward-only --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/${TAG}_eval.log 2>&1
echo traced
