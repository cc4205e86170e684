#!/bin/bash
# One GPU-box pass of the randomised sweeps on the round's final kernels -> gpurun_out/<tag>_stability.log (copied to profiles/).
#   gpurun --timeout 2400 -- "HEAD_STAMP=$(git rev-parse --short HEAD) bash tools/stability_round.sh r04"
tag=${1:-r04}
off=${SEED_OFF:-0}   # added to every sweep's first seed: a later round of the same tag runs fresh scenes
cd "$(dirname "$0")/.." || exit 1
out=gpurun_out/${tag}_stability.log
mkdir -p gpurun_out
{
  echo "# HEAD ${HEAD_STAMP:-unknown}: randomised GPU-vs-oracle sweeps (tools/stress_*.py), native backtrace preload"
  export LD_PRELOAD=$PWD/tools/diag/libsegv_bt.so
  run() { local t0=$SECONDS; "$@" > gpurun_out/_stab.tmp 2>&1; local rc=$?; echo "rc $rc ${SECONDS}s+$((SECONDS - t0)): $* :: $(tail -1 gpurun_out/_stab.tmp | cut -c1-160)"; }
  run timeout 900 python3 tools/stress_parity.py 4000 $((71000 + off))
  STRESS_SWITCHES=1 run timeout 900 python3 tools/stress_parity.py 4000 $((75000 + off))
  DUST_HIP_RAY_STREAM=1 run timeout 900 python3 tools/stress_parity.py 3000 $((81000 + off))
  DUST_HIP_RAY_STREAM=1 STRESS_MANY=1 run timeout 900 python3 tools/stress_parity.py 100 $((84000 + off))
  DUST_HIP_RAY_STREAM=1 STRESS_DEEP=1 run timeout 900 python3 tools/stress_parity.py 800 $((85000 + off))
  STRESS_BIG=1 run timeout 900 python3 tools/stress_parity.py 600 $((79000 + off))
  STRESS_MANY=1 run timeout 900 python3 tools/stress_parity.py 150 $((79600 + off))
  STRESS_FULL=1 run timeout 900 python3 tools/stress_parity.py 200 $((79800 + off))
  STRESS_DEEP=1 run timeout 900 python3 tools/stress_parity.py 1500 $((90000 + off))
  STRESS_FULLGI=1 run timeout 1200 python3 tools/stress_parity.py 12 $((93000 + off))
  run timeout 900 python3 tools/stress_sharded.py 300 $((91500 + off))
  run timeout 900 python3 tools/stress_host.py bands 300 $((92000 + off))
  run timeout 900 python3 tools/stress_host.py commits 600 $((92300 + off))
  run timeout 900 python3 tools/stress_host.py threads 20 $((92600 + off))
  run timeout 900 python3 tools/stress_host.py schedule 40 $((92700 + off))
  run timeout 900 python3 tools/stress_host.py frames 400 $((92800 + off))
  run timeout 600 python3 tools/stress_edits.py
  run timeout 600 python3 tools/stress_denoise.py
  run timeout 600 python3 tools/stress_tonemap.py
} > "$out" 2>&1
cat "$out"
