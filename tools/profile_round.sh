#!/bin/bash
# One GPU-box pass that produces everything profiles/ keeps for a round:
#   python tools/kernel_sections.py --build && make -s -C dust_amd/csrc VARIANT=wt EXTRA=-DDUST_WAVE_TIMES && gpurun --timeout 2400 -- "HEAD_STAMP=$(git rev-parse --short HEAD) bash tools/profile_round.sh r03"
# writes gpurun_out/<tag>_{gpu_tests.log,bench*.log,kernel_stats*.{txt,json},pmc*.txt,sections*.txt,tile_costs.txt,denoise_kernels.txt}.
# Counter passes run separately from the kernel-trace/stats pass (one counter group per run).
tag=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
export TMPDIR=/tmp
out=$R/gpurun_out
mkdir -p "$out"
cd "$R" || exit 1

{ echo "# HEAD ${HEAD_STAMP:-unknown}"; python -m pytest tests -m gpu -x -q -p no:cacheprovider; } > "$out/${tag}_gpu_tests.log" 2>&1
tail -2 "$out/${tag}_gpu_tests.log"

python bench.py --no-extra-curves > "$out/${tag}_bench.log" 2> "$out/${tag}_bench.err"   # the headline: eight frames per launch (dust_hip_render_frames)
tail -1 "$out/${tag}_bench.log" | cut -c1-600
# round 6, second half: frames per launch -- 1 (rounds 1-5's headline: a launch per frame), 2, 3 (the reference's frames in flight), 4
for k in 1 2 3 4; do
  python bench.py --frames-per-launch $k --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_fpl$k.log" 2>> "$out/${tag}_bench.err"
done
python bench.py --width 3840 --height 2160 --steps 64 --frames-per-launch 1 --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_4k_fpl1.log" 2>> "$out/${tag}_bench.err"
python bench.py --workload gi --no-cpu-baseline > "$out/${tag}_bench_gi.log" 2>> "$out/${tag}_bench.err"
tail -1 "$out/${tag}_bench_gi.log" | cut -c1-600
python bench.py --workload deep --steps 30 > "$out/${tag}_bench_deep.log" 2>> "$out/${tag}_bench.err"
tail -1 "$out/${tag}_bench_deep.log" | cut -c1-600
python bench.py --gpus 1 --steps 20 --warmup 5 > "$out/${tag}_bench_driver_style.log" 2>> "$out/${tag}_bench.err"   # the driver's command line: the headline + curves.moving / primary_ao_4k / gi_1080p / deep
python bench.py --camera orbit --steps 240 --no-cpu-baseline > "$out/${tag}_bench_moving.log" 2>> "$out/${tag}_bench.err"   # the moving view as the headline
python bench.py --workload teapot_cpu > "$out/${tag}_bench_teapot_cpu.log" 2>> "$out/${tag}_bench.err"
python bench.py --workload gi --width 3840 --height 2160 --steps 40 --no-cpu-baseline > "$out/${tag}_bench_gi_4k.log" 2>> "$out/${tag}_bench.err"
DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --no-cpu-baseline > "$out/${tag}_bench_gi_inplace.log" 2>> "$out/${tag}_bench.err"
# round 6: two whole frames in flight on one GPU (each on half of the slots / every launch asking for all of them), the deterministic apply on one GPU
python bench.py --frames-in-flight 2 --frames-per-launch 1 --in-flight-slots share --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_pipelined.log" 2>> "$out/${tag}_bench.err"
python bench.py --frames-in-flight 2 --frames-per-launch 1 --in-flight-slots all --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_pipelined_all.log" 2>> "$out/${tag}_bench.err"
DUST_BENCH_GI_ORDERED=1 DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --no-cpu-baseline > "$out/${tag}_bench_gi_ordered_inplace.log" 2>> "$out/${tag}_bench.err"
python bench.py --width 3840 --height 2160 --steps 64 --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_4k.log" 2>> "$out/${tag}_bench.err"
# round 5: thousands of instances (the packet cull's 64-wide hierarchy against every box for every packet), the GI passes as ray streams
# (opt-in on the castle, the default for the deep tree's gather), N-rank denoise on one GPU
python bench.py --props 4000 --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_props.log" 2>> "$out/${tag}_bench.err"
DUST_HIP_FLAT_CULL=1 python bench.py --props 4000 --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_props_flat_cull.log" 2>> "$out/${tag}_bench.err"
DUST_HIP_RAY_STREAM=1 DUST_HIP_NO_SIDE_STREAM=1 python bench.py --workload gi --no-cpu-baseline > "$out/${tag}_bench_gi_stream_inplace.log" 2>> "$out/${tag}_bench.err"
DUST_HIP_RAY_STREAM=1 python bench.py --workload deep --steps 30 --no-cpu-baseline > "$out/${tag}_bench_deep_stream.log" 2>> "$out/${tag}_bench.err"
DUST_HIP_PACKET_GI=1 python bench.py --workload deep --steps 30 --no-cpu-baseline > "$out/${tag}_bench_deep_packet.log" 2>> "$out/${tag}_bench.err"
python bench.py --denoise --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_denoise.log" 2>> "$out/${tag}_bench.err"
# what ONE rank of an N-GPU row-band run does between collectives (four frames in flight, launches on a quarter of the slots each, 32 slots reserved):
# every band of N = 8 while the cuts are balanced on measured band steps (round 6), the timed region on the SLOWEST band; N = 2 and 4 likewise
: > "$out/${tag}_bench_bands_emulated.log"
for n in 2 4 8; do
  DUST_BENCH_EMULATE_BAND=all/$n python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-curves --frames-in-flight 4 --frames-per-launch 1 2>> "$out/${tag}_bench.err" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); c=j['config']; print(json.dumps({'bands': $n, 'slowest_band': c['emulated_band'], 'ms_per_step_slowest': j['ms_per_step'], 'band_steps_ms': c['band_steps_ms'], 'band_balance': c['band_balance'], 'frames_in_flight': c['frames_in_flight']}))" >> "$out/${tag}_bench_bands_emulated.log"
done
# ... and of an N = 8 GI job (round 6: pixel passes on the band, the exchange's export / import, 1/8 of the surfel trace, ordering + ordered apply replicated; one frame in flight)
: > "$out/${tag}_bench_gi_bands_emulated.log"
for spec in "gi 1920 1080 100" "gi 3840 2160 60" "deep 1920 1080 30"; do
  set -- $spec
  python bench.py --workload $1 --width $2 --height $3 --steps $4 --no-cpu-baseline 2>> "$out/${tag}_bench.err" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'workload': '$1', 'frame': [$2, $3], 'band': 'whole frame, one GPU', 'ms_per_step': j['ms_per_step'], 'kernels_ms': j['roofline']['kernels_ms']}))" >> "$out/${tag}_bench_gi_bands_emulated.log"
  for r in 0 1 2 3 4 5 6 7; do
    DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --workload $1 --width $2 --height $3 --steps $4 --no-cpu-baseline 2>> "$out/${tag}_bench.err" | tail -1 |
      python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'workload': '$1', 'frame': [$2, $3], 'band': '$r/8', 'ms_per_step': j['ms_per_step'], 'kernels_ms': j['roofline']['kernels_ms']}))" >> "$out/${tag}_bench_gi_bands_emulated.log"
  done
done

# a 1/8 band launched ALONE, one frame in flight: what single-frame strong scaling on 8 GPUs would get from the kernel
: > "$out/${tag}_bench_band_alone.log"
for r in 0 1 2 3 4 5 6 7; do
  DUST_HIP_RESERVE_BLOCKS=32 DUST_BENCH_EMULATE_BAND=$r/8 python bench.py --gpus 1 --steps 200 --warmup 5 --no-cpu-baseline --no-extra-curves --frames-in-flight 1 2>> "$out/${tag}_bench.err" |
    python -c "import sys,json; j=json.loads(sys.stdin.read()); print(json.dumps({'band': '$r/8', 'ms_per_step': j['ms_per_step'], 'kernel_ms': j['roofline']['kernel_ms'], 'frames_in_flight': j['config']['frames_in_flight']}))" >> "$out/${tag}_bench_band_alone.log"
done

cd /tmp || exit 1
rm -rf "$out/prof_$tag" "$out/pmc_$tag"
rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o bench -- \
    python "$R/bench.py" --steps 24 --warmup 3 --no-cpu-baseline --no-extra-curves > "$out/${tag}_bench_prof.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o bench_gi -- \
    python "$R/bench.py" --workload gi --steps 20 --warmup 3 --no-cpu-baseline > "$out/${tag}_bench_gi_prof.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o bench_deep -- \
    python "$R/bench.py" --workload deep --steps 10 --warmup 2 --no-cpu-baseline > "$out/${tag}_bench_deep_prof.log" 2>&1
DUST_HIP_RAY_STREAM=1 DUST_HIP_NO_SIDE_STREAM=1 rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o bench_gi_stream -- \
    python "$R/bench.py" --workload gi --steps 20 --warmup 3 --no-cpu-baseline > "$out/${tag}_bench_gi_stream_prof.log" 2>&1
DUST_BENCH_EMULATE_BAND=3/8 rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o bench_gi_band -- \
    python "$R/bench.py" --workload gi --width 3840 --height 2160 --steps 20 --warmup 3 --no-cpu-baseline > "$out/${tag}_bench_gi_band_prof.log" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_$tag" -name 'bench_gi_band_results.db') \
    --json "$out/${tag}_kernel_stats_gi_band_4k.json" > "$out/${tag}_kernel_stats_gi_band_4k.txt" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_$tag" -name 'bench_gi_stream_results.db') \
    --json "$out/${tag}_kernel_stats_gi_stream.json" > "$out/${tag}_kernel_stats_gi_stream.txt" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_$tag" -name 'bench_deep_results.db') \
    --json "$out/${tag}_kernel_stats_deep.json" > "$out/${tag}_kernel_stats_deep.txt" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_$tag" -name 'bench_results.db') \
    --json "$out/${tag}_kernel_stats.json" > "$out/${tag}_kernel_stats.txt" 2>&1
python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_$tag" -name 'bench_gi_results.db') \
    --json "$out/${tag}_kernel_stats_gi.json" > "$out/${tag}_kernel_stats_gi.txt" 2>&1
head -12 "$out/${tag}_kernel_stats.txt"

groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"
        "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY"
        "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE")
for wl in primary_ao single gi deep; do
  for c in "${groups[@]}"; do
    n=$(echo $c | cut -d' ' -f1)
    if [ $wl = deep ] && [ $n != FETCH_SIZE ] && [ $n != WRITE_SIZE ]; then continue; fi
    if [ $wl = single ] && [ $n != FETCH_SIZE ] && [ $n != WRITE_SIZE ] && [ $n != TCC_HIT_sum ]; then continue; fi
    if [ $wl = single ]; then wa="--workload primary_ao --frames-per-launch 1"; else wa="--workload $wl"; fi   # (single: k_primary_ao<0>, a launch per frame)
    st=4; if [ $wl = primary_ao ]; then st=8; fi   # (whole launches of eight frames only: the counters are means per launch)
    rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_$tag" -o ${wl}_$n -- \
        python "$R/bench.py" $wa --steps $st --warmup 2 --no-cpu-baseline --no-extra-curves > "$out/pmc_${wl}_$n.log" 2>&1
  done
done
python "$R/profiles/summarize_pmc.py" $(find "$out/pmc_$tag" -name 'primary_ao_*_results.db' | sort) > "$out/${tag}_pmc.txt" 2>&1
python "$R/profiles/summarize_pmc.py" $(find "$out/pmc_$tag" -name 'single_*_results.db' | sort) > "$out/${tag}_pmc_single.txt" 2>&1
python "$R/profiles/summarize_pmc.py" $(find "$out/pmc_$tag" -name 'gi_*_results.db' | sort) > "$out/${tag}_pmc_gi.txt" 2>&1
python "$R/profiles/summarize_pmc.py" $(find "$out/pmc_$tag" -name 'deep_*_results.db' | sort) > "$out/${tag}_pmc_deep.txt" 2>&1
rocprofv3 --kernel-trace --stats -d "$out/prof_$tag" -o denoise -- \
    python "$R/tools/denoise_timing.py" > "$out/${tag}_denoise_timing.log" 2>&1
{ grep "GI frame" "$out/${tag}_denoise_timing.log"; python "$R/profiles/summarize_rocprof.py" $(find "$out/prof_$tag" -name 'denoise_results.db') | grep -i "denoise\|kernel  "; } > "$out/${tag}_denoise_kernels.txt" 2>&1
python "$R/tools/kernel_sections.py" > "$out/${tag}_sections.txt" 2>&1
python "$R/tools/kernel_sections.py" --deep > "$out/${tag}_sections_deep.txt" 2>&1
DUST_HIP_RAY_STREAM=1 python "$R/tools/kernel_sections.py" > "$out/${tag}_sections_stream.txt" 2>&1
python "$R/tools/kernel_sections.py" --props 4000 > "$out/${tag}_sections_props.txt" 2>&1
# SQ counters (instructions, lane activity, waits) of the stream and the packet GI kernels side by side
for v in stream packet; do
  if [ $v = stream ]; then export DUST_HIP_RAY_STREAM=1; else unset DUST_HIP_RAY_STREAM; fi
  i=0
  for c in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_SMEM SQ_ACTIVE_INST_LDS SQ_THREAD_CYCLES_VALU GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM"; do
    i=$((i + 1))
    DUST_HIP_NO_SIDE_STREAM=1 rocprofv3 --pmc $c --kernel-trace -d "$out/pmc_$tag" -o sq_${v}_$i -- \
        python "$R/bench.py" --workload gi --steps 4 --warmup 2 --no-cpu-baseline --no-extra-curves > "$out/pmc_sq_${v}_$i.log" 2>&1
  done
  python "$R/profiles/summarize_pmc.py" $(find "$out/pmc_$tag" -name "sq_${v}_*_results.db" | sort) | grep -E "k_ray_walk<., 0>|k_gather_rays|k_surfel_rays|k_final_gather<0>|k_surfel_trace<0>|k_primary_ao<0>" > "$out/${tag}_sq_$v.txt" 2>&1
done
unset DUST_HIP_RAY_STREAM
python "$R/tools/tile_costs.py" > "$out/${tag}_tile_costs.txt" 2>&1
python "$R/tools/diag/surfel_items.py" 40 > "$out/${tag}_surfel_items.txt" 2>&1
DUST_HIP_LIB=$R/dust_amd/libdust_hip_wt.so python "$R/tools/wave_times.py" 300 > "$out/${tag}_wave_times.txt" 2>&1
DUST_HIP_LIB=$R/dust_amd/libdust_hip_wt.so python "$R/tools/wave_times.py" 320 --frames-per-launch 8 > "$out/${tag}_wave_times_8_per_launch.txt" 2>&1
DUST_HIP_EQUAL_BANDS=1 DUST_HIP_LIB=$R/dust_amd/libdust_hip_wt.so python "$R/tools/wave_times.py" 300 > "$out/${tag}_wave_times_equal_bands.txt" 2>&1
wc -l "$out/${tag}_pmc.txt" "$out/${tag}_pmc_gi.txt"
rm -rf "$out/prof_$tag" "$out/pmc_$tag"
