#!/bin/bash
# round 5, first GPU pass of the ray-stream GI kernels: GI parity tests, then the GI frame in place, stream against packet kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out
mkdir -p "$out"
cd "$R" || exit 1
timeout 900 python -m pytest tests/test_gpu_gi.py -x -q -p no:cacheprovider > "$out/r5_gi_tests.log" 2>&1
tail -15 "$out/r5_gi_tests.log"
for v in stream packet; do
  if [ $v = packet ]; then export DUST_HIP_PACKET_GI=1; else unset DUST_HIP_PACKET_GI; fi
  DUST_HIP_NO_SIDE_STREAM=1 timeout 300 python bench.py --workload gi --no-cpu-baseline --steps 40 > "$out/r5_bench_gi_inplace_$v.log" 2>&1
  tail -1 "$out/r5_bench_gi_inplace_$v.log" | cut -c1-400
  timeout 300 python bench.py --workload gi --no-cpu-baseline --steps 40 > "$out/r5_bench_gi_$v.log" 2>&1
  tail -1 "$out/r5_bench_gi_$v.log" | cut -c1-400
done
unset DUST_HIP_PACKET_GI
cd /tmp
rm -rf "$out/prof_r5a"
DUST_HIP_NO_SIDE_STREAM=1 timeout 600 rocprofv3 --kernel-trace --stats -d "$out/prof_r5a" -o bench_gi -- python "$R/bench.py" --workload gi --steps 20 --warmup 3 --no-cpu-baseline > "$out/r5_bench_gi_prof.log" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_r5a" -name 'bench_gi_results.db') --json "$out/r5_kernel_stats_gi.json" > "$out/r5_kernel_stats_gi.txt" 2>&1
cat "$out/r5_kernel_stats_gi.txt"
rm -rf "$out/prof_r5a"
