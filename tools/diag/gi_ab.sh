#!/bin/bash
# A/B of library builds on the GI frame (arguments: the builds); step times (side stream and in place) + rocprofv3 per-kernel averages in place
R=$GRAFT_REPO_ROOT; cd $R; out=$R/gpurun_out
export ROUNDS=${ROUNDS:-2} WORKLOADS=gi STEPS=60
echo "== side stream"; bash tools/ab.sh "$@"
echo "== in place"; DUST_HIP_NO_SIDE_STREAM=1 bash tools/ab.sh "$@"
cd /tmp; export TMPDIR=/tmp
for spec in "$@"; do
  tag=$(basename $spec .so)
  rm -rf $out/s6prof_$tag
  DUST_HIP_NO_SIDE_STREAM=1 DUST_HIP_LIB=$R/$spec rocprofv3 --kernel-trace --stats -d $out/s6prof_$tag -o gi -- python $R/bench.py --workload gi --steps 30 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
  echo "== rocprof in place: $tag"
  python $R/profiles/summarize_rocprof.py $(find $out/s6prof_$tag -name 'gi_results.db') 2>&1 | cut -c1-110 | head -16
  rm -rf $out/s6prof_$tag
done
