cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06b_gpu_tests.log 2>&1; tail -4 gpurun_out/r06b_gpu_tests.log
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$1', j['ms_per_step'], j['value'], j['roofline']['kernels_ms'], j['config'].get('frames_in_flight'))"; }
{
for r in 1 2; do
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves 2>/dev/null | one fif1
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots all 2>/dev/null | one fif2_all
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 3 --in-flight-slots all 2>/dev/null | one fif3_all
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots share 2>/dev/null | one fif2_share
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 3 --in-flight-slots share 2>/dev/null | one fif3_share
done
python bench.py --steps 60 --no-cpu-baseline --no-extra-curves --width 3840 --height 2160 2>/dev/null | one 4k_fif1
python bench.py --steps 60 --no-cpu-baseline --no-extra-curves --width 3840 --height 2160 --frames-in-flight 2 2>/dev/null | one 4k_fif2_all
} > gpurun_out/r06b_in_flight.log 2>&1
cat gpurun_out/r06b_in_flight.log
{
for sh in 0 27 30 33 36 39 42; do
  for r in 1 2; do DUST_HIP_SIDE_SHARE=$sh python bench.py --workload gi --steps 60 --no-cpu-baseline 2>/dev/null | one share$sh; done
done
} > gpurun_out/r06b_share_sweep.log 2>&1
cat gpurun_out/r06b_share_sweep.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r06b_bench_driver_style.log 2> gpurun_out/r06b_bench.err
python -c "
import json
j=json.loads(open('gpurun_out/r06b_bench_driver_style.log').read().strip().splitlines()[-1])
print(j['value'], j['ms_per_step'], j['roofline']['frac'])
for k,v in j['curves'].items():
    print(k, v.get('ms_per_step'), v.get('value'), (v.get('roofline') or {}).get('frac'), v.get('error'))
"
