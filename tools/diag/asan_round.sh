#!/bin/bash
# (tests that start torch on the GPU fail under the preloaded sanitizer runtime -- torch cannot dlopen libcaffe2_nvrtc there --: an artefact of the harness)
# host code under AddressSanitizer (tools/diag/build_asan.sh): the suites that drive this round's new host paths
cd "$(dirname "$0")/../.." || exit 1
mkdir -p gpurun_out
{
  echo "# HEAD ${HEAD_STAMP:-unknown}: ASan/UBSan build of the host code (kernels unchanged)"
  bash tools/diag/with_asan.sh python -m pytest tests/test_gpu_comm.py tests/test_gpu_lifetime.py tests/test_gpu_parity.py tests/test_gpu_edge_cases.py tests/test_gpu_gi.py tests/test_gpu_gi_sharded.py tests/test_configs.py tests/test_gpu_batch.py -q -p no:cacheprovider 2>&1 | grep -aE "passed|failed|^FAILED|ERROR|AddressSanitizer|runtime error|SUMMARY|libcaffe2" | sort | uniq -c | tail -30
  for mode in "bands 60 1" "commits 80 2" "threads 8 3" "schedule 10 4" "frames 120 5"; do
    bash tools/diag/with_asan.sh timeout 600 python3 tools/stress_host.py $mode 2>&1 | grep -aE "cases|AddressSanitizer|runtime error|SUMMARY" | tail -3
  done
} > gpurun_out/${ASAN_TAG:-r05}_asan.log 2>&1
cat gpurun_out/${ASAN_TAG:-r05}_asan.log
