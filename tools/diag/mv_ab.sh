#!/bin/bash
# moving-view A/B: env settings given as arguments "K=V,K2=V2" each; prints still ms/step, moving ms/step, moving kernel, still-same-views
for spec in "$@"; do
  envs=$(echo "$spec" | tr ',' ' ')
  [ "$spec" = "-" ] && envs=""
  env $envs python bench.py --camera orbit --steps ${STEPS:-240} --no-cpu-baseline 2>/dev/null | python -c "
import sys,json; j=json.loads(sys.stdin.read()); m=j['curves']['moving']; print('$spec', 'still', j['curves']['strong']['ms_per_step'], 'moving', m['ms_per_step'], m['kernels_ms']['k_primary_ao'], 'same views still', m['still_same_views']['mean_ms_per_step'], 'ratio', m['still_same_views']['moving_over_still'])"
done
