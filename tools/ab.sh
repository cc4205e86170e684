#!/bin/bash
# A/B of library builds inside ONE gpurun call (boxes differ by a few percent, so compare only within a call):
#   gpurun -- 'bash tools/ab.sh dust_amd/libdust_hip.so dust_amd/libdust_hip_x.so[:ENV=VAL,ENV2=VAL2]'
# Alternates the builds for ROUNDS rounds and prints each run's kernel times (ms) for the bench workloads.
ROUNDS=${ROUNDS:-3}
for r in $(seq $ROUNDS); do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""
    if [ "$lib" != "$spec" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    for wl in ${WORKLOADS:-primary_ao}; do
      env $envs DUST_HIP_LIB=$PWD/$lib python3 bench.py --no-cpu-baseline --steps ${STEPS:-60} --warmup 5 --workload $wl 2>/dev/null |
        python3 -c "import sys,json; j=json.loads(sys.stdin.read()); print('$spec', '$wl', j['ms_per_step'], j['roofline']['kernels_ms'])"
    done
  done
done
