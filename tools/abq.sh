#!/bin/bash
# like ab.sh, on tools/quick_time.py (no ray-count bookkeeping: ablation switches that change the rays are fine)
#   abq.sh lib[:ENV=VAL,...] ...      ROUNDS=n  ARGS="--gi"
ROUNDS=${ROUNDS:-2}
for r in $(seq $ROUNDS); do
  for spec in "$@"; do
    lib=${spec%%:*}; envs=""
    if [ "$lib" != "$spec" ]; then envs=$(echo "${spec#*:}" | tr ',' ' '); fi
    env $envs DUST_HIP_LIB=$PWD/$lib python3 tools/quick_time.py --label "$spec" $ARGS 2>&1 | tail -1
  done
done
