#!/bin/bash
# Developer aid: the round's evidence in one call on a GPU box -- full GPU test suite, kernel stats + PMC passes of the headline
# command (collect_profiles.sh / collect_mfma_busy.sh), plain bench lines for the summaries, the Quadtree3DCNN kernel stats and
# the single-stream trace behind profiles/rNN_serial_kernel_costs.txt.  Outputs under gpurun_out/; summarise locally with
# scripts/summarize_profiles.py / summarize_mfma_busy.py / kernel_totals.py and copy into profiles/.
#   gpurun --timeout 1200 -- 'bash scripts/round_evidence.sh r04 tests'      (the full GPU suite: ~10 minutes)
#   gpurun --timeout 1200 -- 'bash scripts/round_evidence.sh r04 profiles'   (everything else)
set -o pipefail
R=${1:-r04}
WHAT=${2:-all}
cd "${GRAFT_REPO_ROOT:-.}"
mkdir -p gpurun_out
if [ "$WHAT" != profiles ]; then
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/${R}_gpu_tests.txt 2>&1; tail -3 gpurun_out/${R}_gpu_tests.txt
  grep -q " passed" gpurun_out/${R}_gpu_tests.txt || exit 1
  if grep -q "failed\|error" gpurun_out/${R}_gpu_tests.txt; then exit 1; fi
  [ "$WHAT" = tests ] && exit 0
fi
bash scripts/collect_profiles.sh ${R}p && bash scripts/collect_mfma_busy.sh ${R}m || exit 1
python bench.py --no-cpu-baseline --profile-steps 0 > gpurun_out/${R}_plain.json || exit 1
python bench.py --forward-only --no-cpu-baseline --profile-steps 0 > gpurun_out/${R}_plain_eval.json || exit 1
export TMPDIR=/tmp
# Quadtree3DCNN (BASELINE config 4): kernel stats + FETCH_SIZE / WRITE_SIZE passes (summarize_profiles.py <tag>q3p <round> quadtree3d)
QT_PROFILE_ARGS="--model quadtree3d" bash scripts/collect_profiles.sh ${R}q3p || exit 1
export QTCNN_SIDE_STREAM=0
timeout -k 10 240 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/${R}ser -o ser -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-forward-leg --profile-steps 0 > gpurun_out/${R}ser.log 2>&1 || exit 1
unset QTCNN_SIDE_STREAM
# LAST: the exact command the driver runs, on the tree as it stands.  Summarise + commit the profiles, then run
#   gpurun -- 'python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json'
# once more on the committed tree and commit NOTHING that bench.py reads after it (tests/test_bench_gpu.py runs the same
# command; tests/test_bench_cpu.py checks every committed summary against the reader).
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/${R}_driver_command.json || exit 1
python -c "import json,sys; r=json.load(open('gpurun_out/${R}_driver_command.json')); assert r['roofline'] and r['cpu_baseline'] and r['forward'], r" || exit 1
echo done
