#!/bin/bash
# GI frame against the surfel pass's share of the workgroup slots (DUST_HIP_SIDE_SHARE; "f" = the calibrated formula).
# usage: SHARES="f 30 36 42" share_sweep.sh [bench.py arguments]
cd $GRAFT_REPO_ROOT
for r in 1 2; do for sh in ${SHARES:-f 30 36 42 48 54}; do
  v=""; [ "$sh" != f ] && v="DUST_HIP_SIDE_SHARE=$sh"
  env $v python bench.py --workload gi --steps 60 --warmup 5 --no-cpu-baseline "$@" 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('share $sh', j['ms_per_step'], j['roofline']['kernels_ms'])"
done; done
