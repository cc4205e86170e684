cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
one() { python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print('$1', j['ms_per_step'], j['roofline']['kernels_ms'], c.get('frames_in_flight'))"; }
{
for r in 1 2 3; do
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves 2>/dev/null | one fif1
python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots share 2>/dev/null | one fif2_share
DUST_HIP_WIDE_SHARE=1 python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots share 2>/dev/null | one fif2_share_wide
DUST_HIP_IN_FLIGHT_OVERSUB=25 python bench.py --steps 100 --no-cpu-baseline --no-extra-curves --frames-in-flight 2 --in-flight-slots share 2>/dev/null | one fif2_share_over25
done
} > gpurun_out/r06i_in_flight.log 2>&1
cat gpurun_out/r06i_in_flight.log
HEAD_STAMP=$1 bash tools/stability_round.sh r06 > /dev/null 2>&1
cat gpurun_out/r06_stability.log
