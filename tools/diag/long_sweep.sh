#!/bin/bash
# a longer pass of the randomised GPU-vs-oracle sweep than tools/stability_round.sh makes: 30 000 scenes + 10 000 with random switches
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  echo "# HEAD ${HEAD_STAMP:-unknown}: long sweep"
  export LD_PRELOAD=$PWD/tools/diag/libsegv_bt.so
  t0=$SECONDS; timeout 1500 python3 tools/stress_parity.py 30000 70000 2>&1 | tail -1; echo "rc $? $((SECONDS - t0))s: tools/stress_parity.py 30000 70000"
  t0=$SECONDS; STRESS_SWITCHES=1 timeout 1200 python3 tools/stress_parity.py 10000 110000 2>&1 | tail -1; echo "rc $? $((SECONDS - t0))s: STRESS_SWITCHES=1 tools/stress_parity.py 10000 110000"
  t0=$SECONDS; STRESS_DEEP=1 timeout 900 python3 tools/stress_parity.py 4000 130000 2>&1 | tail -1; echo "rc $? $((SECONDS - t0))s: STRESS_DEEP=1 tools/stress_parity.py 4000 130000"
} > gpurun_out/r04_long_sweep.log 2>&1
cat gpurun_out/r04_long_sweep.log
