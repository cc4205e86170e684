cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests/test_gpu_comm.py tests/test_gpu_edge_cases.py tests/test_gpu_gi_sharded.py -m gpu -x -q -p no:cacheprovider > gpurun_out/r06c_gpu_tests.log 2>&1; tail -4 gpurun_out/r06c_gpu_tests.log
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print('$1', j['ms_per_step'], j['roofline']['kernels_ms'], c.get('frames_in_flight'), c.get('emulated_band'), c.get('band_steps_ms'), c.get('band_balance'))"; }
{
DUST_BENCH_EMULATE_BAND=all/8 python bench.py --steps 200 --no-cpu-baseline --no-extra-curves --frames-in-flight 4 2>&1 | one bands_all8
DUST_BENCH_EMULATE_BAND=all/8 python bench.py --steps 200 --no-cpu-baseline --no-extra-curves --frames-in-flight 4 --band-rebalance 0 2>&1 | one bands_all8_nobalance
for r in 0 2 3 5 7; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload gi --steps 100 --no-cpu-baseline 2>&1 | tail -1 | one gi_1080p_$r; done
python bench.py --workload gi --steps 60 --no-cpu-baseline 2>/dev/null | one gi_1080p_full
for r in 0 3 7; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload gi --width 3840 --height 2160 --steps 60 --no-cpu-baseline 2>&1 | tail -1 | one gi_4k_$r; done
python bench.py --workload gi --width 3840 --height 2160 --steps 40 --no-cpu-baseline 2>/dev/null | one gi_4k_full
for r in 0 4; do DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload deep --steps 30 --no-cpu-baseline 2>&1 | tail -1 | one deep_$r; done
python bench.py --workload deep --steps 30 --no-cpu-baseline 2>/dev/null | one deep_full
} > gpurun_out/r06c_bands.log 2>&1
cat gpurun_out/r06c_bands.log
{
python bench.py --steps 60 --no-cpu-baseline --no-extra-curves --width 3840 --height 2160 --frames-in-flight 2 --in-flight-slots share 2>/dev/null | one 4k_fif2_share
DUST_HIP_NO_WIDE_FUSED=1 python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots all 2>/dev/null | one fif2_all_narrow
DUST_HIP_NO_WIDE_FUSED=1 python bench.py --steps 60 --no-cpu-baseline --no-extra-curves --width 3840 --height 2160 --frames-in-flight 2 --in-flight-slots all 2>/dev/null | one 4k_fif2_all_narrow
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots share 2>/dev/null | one fif2_share
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 4 --in-flight-slots share 2>/dev/null | one fif4_share
} > gpurun_out/r06c_in_flight.log 2>&1
cat gpurun_out/r06c_in_flight.log
cd /tmp; export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_r06c
DUST_BENCH_EMULATE_BAND=3/8 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r06c -o gi_band -- python $GRAFT_REPO_ROOT/bench.py --workload gi --steps 40 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r06c_gi_band_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(find gpurun_out/prof_r06c -name 'gi_band_results.db') --json gpurun_out/r06c_kernel_stats_gi_band.json > gpurun_out/r06c_kernel_stats_gi_band.txt 2>&1
head -40 gpurun_out/r06c_kernel_stats_gi_band.txt
rm -rf gpurun_out/prof_r06c
