#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by a few percent, so compare only within a call):
#   gpurun -- 'bash tools/ab.sh dust_amd/libdust_hip_base.so dust_amd/libdust_hip.so'
# Alternates the builds for ROUNDS rounds and prints each run's kernel times (ms) for both bench workloads.
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for lib in "$@"; do
    for wl in ${WORKLOADS:-primary_ao gi}; do
      DUST_HIP_LIB=$PWD/$lib python bench.py --no-cpu-baseline --steps 40 --warmup 5 --workload $wl 2>/dev/null |
        python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', '$wl', j['ms_per_step'], j['roofline']['kernels_ms'])"
    done
  done
done
