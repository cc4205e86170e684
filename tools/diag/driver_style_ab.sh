#!/bin/bash
# the driver's command line (bench.py --gpus 1 --steps 20 --warmup 5, headline only) under library builds, alternating. usage: driver_style_ab.sh libA libB ...
cd $GRAFT_REPO_ROOT
for r in $(seq ${ROUNDS:-4}); do for lib in "$@"; do
  DUST_HIP_LIB=$PWD/$lib python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-extra-curves 2>/dev/null |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$lib', j['ms_per_step'], j['roofline']['kernel_ms'], j['config'].get('untimed_steps_before_timing'))"
done; done
