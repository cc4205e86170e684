#!/bin/bash
# usage: with_asan.sh <command...>   runs it against the sanitizer builds of tools/diag/build_asan.sh
root="$(cd "$(dirname "$0")/../.." && pwd)"
rt=$(ls -d /opt/rocm/lib/llvm/lib/clang/*/lib/linux)/libclang_rt.asan-x86_64.so
ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0:abort_on_error=0 LD_PRELOAD=$rt DUST_HIP_LIB=$root/dust_amd/_asan/libdust_hip.so \
  DUST_ORACLE_LIB=$root/dust_amd/_asan/liboracle.so "$@"
