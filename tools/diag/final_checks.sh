#!/bin/bash
# the GPU suite three times over on the round's final commit, then the heavier slices of the randomised sweep with fresh seeds
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  echo "# HEAD ${HEAD_STAMP:-unknown}"
  for i in 1 2 3; do python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | grep -aE "passed|failed" | tail -1; done
  python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
  export LD_PRELOAD=$PWD/tools/diag/libsegv_bt.so
  run() { local t0=$SECONDS; "$@" > gpurun_out/_fin.tmp 2>&1; echo "rc $? $((SECONDS - t0))s: $* :: $(tail -1 gpurun_out/_fin.tmp | cut -c1-160)"; }
  STRESS_BIG=1 run timeout 1200 python3 tools/stress_parity.py 3000 150000
  STRESS_MANY=1 run timeout 1200 python3 tools/stress_parity.py 600 160000
  STRESS_FULL=1 run timeout 1200 python3 tools/stress_parity.py 800 170000
  STRESS_FULLGI=1 run timeout 1500 python3 tools/stress_parity.py 40 180000
  run timeout 900 python3 tools/stress_sharded.py 2000 190000
  run timeout 900 python3 tools/stress_host.py commits 1500 200000
  run timeout 900 python3 tools/stress_host.py bands 1500 210000
} > gpurun_out/r04_final_checks.log 2>&1
cat gpurun_out/r04_final_checks.log
