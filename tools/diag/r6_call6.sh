cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_configs.py tests/test_gpu_gi.py tests/test_gpu_gi_sharded.py tests/test_gpu_comm.py tests/test_gpu_fullsize.py tests/test_gpu_ray_stream.py tests/test_gpu_lifetime.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06f_gpu_tests.log 2>&1; tail -4 gpurun_out/r06f_gpu_tests.log
STRESS_DEEP=1 timeout 600 python tools/stress_parity.py 300 61000 > gpurun_out/r06f_stress_deep.log 2>&1; tail -3 gpurun_out/r06f_stress_deep.log
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print('$1', j['ms_per_step'], j['roofline']['kernels_ms'], c.get('frames_in_flight'), c.get('emulated_band'))"; }
{
for r in 1 2; do
DUST_HIP_PACKET_PIXELS=1 python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_packets
python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_streams
done
DUST_BENCH_GI_ORDERED=1 DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --steps 60 --no-cpu-baseline 2>/dev/null | one gi_ordered_inplace
for r in 0 3; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload gi --width 3840 --height 2160 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | one gi_4k_$r; done
for r in 0 4; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload deep --steps 30 --no-cpu-baseline 2>&1 | tail -1 | one deep_$r; done
} > gpurun_out/r06f_bench.log 2>&1
cat gpurun_out/r06f_bench.log
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r06f
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06f -o deep -- python $GRAFT_REPO_ROOT/bench.py --workload deep --steps 10 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06f_deep_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(find gpurun_out/prof_r06f -name 'deep_results.db') --json gpurun_out/r06f_kernel_stats_deep.json > gpurun_out/r06f_kernel_stats_deep.txt 2>&1
head -24 gpurun_out/r06f_kernel_stats_deep.txt
rm -rf gpurun_out/prof_r06f
