#!/bin/bash
# env-only sweeps of the GI frame in place; prints ms/step and the final gather / surfel pass times (HIP events) per configuration
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" || exit 1
for cfg in "$@"; do
  env $cfg DUST_HIP_NO_SIDE_STREAM=1 timeout 300 python bench.py --workload gi --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
k = j.get('kernels_ms') or j.get('config', {}).get('kernels_ms') or {}
print('%-70s %.4f ms/step  %s' % ('$cfg', j['ms_per_step'], json.dumps(j['roofline'].get('kernels_ms'))))"
done
