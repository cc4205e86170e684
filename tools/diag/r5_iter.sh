#!/bin/bash
# round 5 iteration pass: GI parity tests, the section profile of the stream kernels, GI frame in place with per-kernel times
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out
mkdir -p "$out"
cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_gi.py -x -q -p no:cacheprovider > "$out/r5_gi_tests.log" 2>&1
tail -3 "$out/r5_gi_tests.log"
timeout 600 python tools/kernel_sections.py > "$out/r5_sections.txt" 2>&1
sed -n '/k_final_gather/,$p' "$out/r5_sections.txt"
for cfg in "$@"; do
  echo "== $cfg"
  env $cfg DUST_HIP_NO_SIDE_STREAM=1 timeout 300 python bench.py --workload gi --no-cpu-baseline --steps 40 2>/dev/null | python -c "
import sys, json
j = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(j['ms_per_step'], json.dumps(j.get('passes_ms', j.get('config', {}).get('passes_ms', ''))))"
done
cd /tmp
rm -rf "$out/prof_r5a"
DUST_HIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_r5a" -o bench_gi -- python "$R/bench.py" --workload gi --steps 20 --warmup 3 --no-cpu-baseline > "$out/r5_bench_gi_prof.log" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_r5a" -name 'bench_gi_results.db') --json "$out/r5_kernel_stats_gi.json" > "$out/r5_kernel_stats_gi.txt" 2>&1
head -14 "$out/r5_kernel_stats_gi.txt" | cut -c1-120
rm -rf "$out/prof_r5a"
