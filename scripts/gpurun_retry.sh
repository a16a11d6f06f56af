#!/bin/bash
# Developer aid (this container): gpurun, re-submitted only while the pod's GPU slots are busy (exit code 3 / "transient":
# nothing ran, nothing was charged).  A command that ran -- whatever its result -- is never repeated.
#   bash scripts/gpurun_retry.sh 900 'python -m pytest tests -m gpu -q'
T=$1; shift
for attempt in 1 2 3 4 5 6 7 8; do
  /usr/local/graft/bin/gpurun --timeout "$T" -- "$@"
  rc=$?
  if [ $rc -ne 3 ]; then exit $rc; fi
  sleep 120
done
exit 3
