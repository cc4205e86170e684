cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
{
for s in 61048 61283; do
 for e in "X=1" "DUST_HIP_PACKET_PIXELS=1" "DUST_HIP_PACKET_PIXELS=1 DUST_HIP_PACKET_GI=1" "DUST_HIP_PACKET_GI=1"; do
  echo "== seed $s env $e"; env $e STRESS_DEEP=1 timeout 300 python tools/stress_parity.py 1 $s $((s-61000)) 2>&1 | tail -2
 done
done
} > gpurun_out/r06g_deep_seeds.log 2>&1
cat gpurun_out/r06g_deep_seeds.log
