#!/bin/bash
# session-6: headline A/B (primary_ao) of two builds, then the GI A/B of s6_ab2.sh, then the surfel item probe on the second build
R=$GRAFT_REPO_ROOT; cd $R
echo "== headline"; ROUNDS=${ROUNDS:-3} WORKLOADS=primary_ao STEPS=100 bash tools/ab.sh "$1" "$2"
shift 2
bash tools/diag/s6_ab2.sh "$@"
cd $R; echo "== surfel items: ${@: -1}"; DUST_HIP_LIB=$R/${@: -1} python tools/diag/surfel_items.py 2>&1 | tail -n 6
