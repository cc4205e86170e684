#!/bin/bash
# GPU box: loops the stress slice to reproduce the round-2 SIGSEGV (GPUTEST_r02.json), with a native backtrace on a crash.
# usage: hunt_crash.sh <loops> [extra env assignments...]     logs under gpurun_out/hunt/
cd "$(dirname "$0")/../.." || exit 1
out=gpurun_out/hunt; mkdir -p $out
loops=${1:-10}; shift
for e in "$@"; do export "$e"; done
fails=0
for i in $(seq 1 $loops); do
  LD_PRELOAD=$PWD/tools/diag/libsegv_bt.so timeout 900 python3 tools/stress_parity.py ${STRESS_N:-120} 1000 > $out/run_$i.log 2>&1
  rc=$?
  echo "loop $i rc $rc $(tail -1 $out/run_$i.log | cut -c1-120)"
  if [ $rc -ne 0 ]; then fails=$((fails+1)); cp $out/run_$i.log $out/FAIL_$i.log; fi
done
echo "hunt: $fails failures of $loops ($*)"
