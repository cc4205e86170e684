#!/bin/bash
# two library builds on the GI workloads (1080p, 4K, deep tree), alternating. usage: share_ab.sh libA libB
cd $GRAFT_REPO_ROOT
for r in 1 2 3; do for lib in "$@"; do
  for wl in "gi" "gi --width 3840 --height 2160" "deep --steps 20"; do
    DUST_HIP_LIB=$PWD/$lib python bench.py --workload $wl --warmup 5 --no-cpu-baseline 2>/dev/null |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', '$wl', j['ms_per_step'], j['roofline']['kernels_ms'])"
  done
done; done
