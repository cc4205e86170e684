cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_configs.py tests/test_gpu_gi.py tests/test_gpu_ray_stream.py tests/test_gpu_fullsize.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06h_gpu_tests.log 2>&1; tail -4 gpurun_out/r06h_gpu_tests.log
STRESS_DEEP=1 timeout 900 python tools/stress_parity.py 600 61000 > gpurun_out/r06h_stress_deep.log 2>&1; tail -3 gpurun_out/r06h_stress_deep.log
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print('$1', j['ms_per_step'], j['roofline']['kernels_ms'], c.get('frames_in_flight'), c.get('emulated_band'))"; }
{
for r in 1 2 3; do
DUST_HIP_LIB=$PWD/dust_amd/libdust_hip_nopf.so python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_noprefetch
python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_prefetch
done
DUST_HIP_LIB=$PWD/dust_amd/libdust_hip_nopf.so DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_noprefetch_inplace
DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_prefetch_inplace
DUST_BENCH_GI_ORDERED=1 DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --steps 60 --no-cpu-baseline 2>/dev/null | one gi_ordered_inplace
} > gpurun_out/r06h_bench.log 2>&1
cat gpurun_out/r06h_bench.log
