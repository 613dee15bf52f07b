#!/bin/bash
# tools/gpu_retry.sh <timeout_s> '<command>' : tools/gpu.sh, retried while the pod's GPU slots are busy (exit code 3)
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()" || exit 1
for i in $(seq 1 30); do
  /usr/local/graft/bin/gpurun --timeout "$1" -- "$2"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 60
done
exit 3
