#!/bin/bash
# usage: tools/gpu_retry.sh <timeout_s> <logfile> '<command>'   - retries while the pod's GPU slots are busy (exit code 3)
t=$1; log=$2; shift 2
for i in $(seq 1 40); do
  /usr/local/graft/bin/gpurun --timeout "$t" -- "$@" > "$log" 2>&1
  rc=$?
  if [ $rc -ne 3 ]; then echo "rc=$rc" >> "$log"; exit $rc; fi
  sleep 45
done
echo "rc=3 (gave up)" >> "$log"
exit 3
