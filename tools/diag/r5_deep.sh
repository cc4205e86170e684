#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" || exit 1
for cfg in "$@"; do
  env $cfg DUST_HIP_NO_SIDE_STREAM=1 timeout 600 python bench.py --workload deep --no-cpu-baseline --steps 20 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('%-70s %.4f ms/step  %s' % ('$cfg', j['ms_per_step'], json.dumps(j['roofline'].get('kernels_ms'))))"
done
