cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_comm.py tests/test_gpu_gi_sharded.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06d_gpu_tests.log 2>&1; tail -4 gpurun_out/r06d_gpu_tests.log
DUST_BENCH_EMULATE_BAND=all/8 python bench.py --steps 200 --no-cpu-baseline --no-extra-curves --frames-in-flight 4 > gpurun_out/r06d_bands_all8.log 2>&1
tail -c 3000 gpurun_out/r06d_bands_all8.log
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r06d
DUST_BENCH_GI_ORDERED=1 DUST_HIP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06d -o gi_ordered -- python $GRAFT_REPO_ROOT/bench.py --workload gi --steps 40 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06d_gi_ordered_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(find gpurun_out/prof_r06d -name 'gi_ordered_results.db') --json gpurun_out/r06d_kernel_stats_gi_ordered.json > gpurun_out/r06d_kernel_stats_gi_ordered.txt 2>&1
head -24 gpurun_out/r06d_kernel_stats_gi_ordered.txt
rm -rf gpurun_out/prof_r06d
