#!/bin/bash
# the walk-only probe builds against the shipped k_primary, primary pass alone (one gpurun call: compare within it)
L=$PWD/dust_amd
for r in 1 2; do
  DUST_HIP_NO_FUSE=1 python3 tools/diag/walk_probe.py $@
  for v in wp4:512 wp5:640 wp6:768 wp8:1024 ws4:512 ws5:640 ws6:768; do
    DUST_HIP_LIB=$L/libdust_hip_${v%%:*}.so DUST_HIP_NO_FUSE=1 DUST_HIP_BLOCK=${v##*:} python3 tools/diag/walk_probe.py $@
  done
done
