#!/bin/bash
# SQ counter passes over the GI frame (in place): per-kernel instruction counts and wait cycles of the stream and packet kernels
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/../.." && pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out
mkdir -p "$out"
cd /tmp || exit 1
rm -rf "$out/pmc_r5"
groups=("SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE"
        "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM")
for v in stream packet; do
  if [ $v = packet ]; then export DUST_HIP_PACKET_GI=1; else unset DUST_HIP_PACKET_GI; fi
  i=0
  for c in "${groups[@]}"; do
    i=$((i + 1))
    DUST_HIP_NO_SIDE_STREAM=1 timeout 300 rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_r5" -o ${v}_$i -- \
        python "$R/bench.py" --workload gi --steps 4 --warmup 2 --no-cpu-baseline --no-extra-curves > "$out/pmc_${v}_$i.log" 2>&1
  done
  python "$R/profiles/summarize_pmc.py" $(find "$out/pmc_r5" -name "${v}_*_results.db" | sort) | grep -E "k_ray_walk<., 0>|k_final_gather<0>|k_surfel_trace<0>|k_primary_ao<0>" > "$out/r5_pmc_$v.txt" 2>&1
done
cat "$out/r5_pmc_stream.txt" "$out/r5_pmc_packet.txt" | cut -c1-130
rm -rf "$out/pmc_r5"
