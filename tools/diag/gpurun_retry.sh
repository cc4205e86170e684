#!/bin/bash
# gpurun with retries while the pod's GPU slots are busy (exit code 3: nothing charged). usage: gpurun_retry.sh <timeout s> '<command>'
for i in $(seq 1 12); do
  gpurun --timeout "$1" -- "$2"; rc=$?
  [ $rc -ne 3 ] && exit $rc
  sleep 45
done
exit 3
