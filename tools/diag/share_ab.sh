#!/bin/bash
# two library builds on the GI workloads (WLS: ';'-separated bench.py workload arguments), alternating. usage: [ROUNDS=3] share_ab.sh libA libB
cd $GRAFT_REPO_ROOT
IFS=';' read -ra wls <<< "${WLS:-gi;gi --width 3840 --height 2160;deep --steps 20}"
for r in $(seq ${ROUNDS:-3}); do for lib in "$@"; do
  for wl in "${wls[@]}"; do
    DUST_HIP_LIB=$PWD/$lib python bench.py --workload $wl --warmup 5 --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', '$wl', j['ms_per_step'], j['roofline']['kernels_ms'])"
  done
done; done
