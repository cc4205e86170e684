cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06e_gpu_tests.log 2>&1; tail -4 gpurun_out/r06e_gpu_tests.log
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print('$1', j['ms_per_step'], j['roofline']['kernels_ms'], c.get('frames_in_flight'), c.get('emulated_band'))"; }
{
for r in 1 2; do
DUST_HIP_LIB=$PWD/dust_amd/libdust_hip_nopf.so python bench.py --props 4000 --steps 60 --no-cpu-baseline --no-extra-curves 2>/dev/null | one props_nopf
python bench.py --props 4000 --steps 60 --no-cpu-baseline --no-extra-curves 2>/dev/null | one props_prefetch
done
DUST_BENCH_GI_ORDERED=1 DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --steps 60 --no-cpu-baseline 2>/dev/null | one gi_ordered_inplace
DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --steps 60 --no-cpu-baseline 2>/dev/null | one gi_racy_inplace
for r in 0 3 7; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload gi --steps 100 --no-cpu-baseline 2>&1 | tail -1 | one gi_1080p_$r; done
for r in 0 3 7; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload gi --width 3840 --height 2160 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | one gi_4k_$r; done
for r in 0 4; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload deep --steps 30 --no-cpu-baseline 2>&1 | tail -1 | one deep_$r; done
} > gpurun_out/r06e_bands.log 2>&1
cat gpurun_out/r06e_bands.log
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r06e
DUST_BENCH_EMULATE_BAND=3/8 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06e -o gi_band -- python $GRAFT_REPO_ROOT/bench.py --workload gi --width 3840 --height 2160 --steps 40 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06e_gi_band_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(find gpurun_out/prof_r06e -name 'gi_band_results.db') --json gpurun_out/r06e_kernel_stats_gi_band_4k.json > gpurun_out/r06e_kernel_stats_gi_band_4k.txt 2>&1
head -24 gpurun_out/r06e_kernel_stats_gi_band_4k.txt
rm -rf gpurun_out/prof_r06e
