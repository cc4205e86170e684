#!/bin/bash
# the headline line N times (ms/step, kernel ms, roofline fraction), then optionally the moving view
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
cd "$R" || exit 1
n=${1:-3}
for i in $(seq 1 $n); do
  timeout 300 python bench.py --no-cpu-baseline --no-extra-curves ${EXTRA_ARGS} 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['ms_per_step'], j['roofline']['kernel_ms'], round(j['roofline']['frac'],4))"
done
