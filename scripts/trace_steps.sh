#!/bin/bash
# Developer aid (GPU box): rocprofv3 kernel traces (timestamps) of the headline train step and of the eval forward
#   gpurun -- 'bash scripts/trace_steps.sh r04t'   -> gpurun_out/<tag>_train/..., gpurun_out/<tag>_eval/...
# then: python scripts/timeline.py <tag>_train -v ; python scripts/timeline_eval.py <tag>_eval
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-trace}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_train/trace -- python3 $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 > $R/gpurun_out/${TAG}_train.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/${TAG}_eval/trace -- python3 $R/bench.py --forward-only --steps 3 --warmup 2 --no-cpu-baseline --profile-steps 0 > $R/gpurun_out/${TAG}_eval.log 2>&1
echo traced
