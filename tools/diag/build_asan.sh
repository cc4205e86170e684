#!/bin/bash
# Sanitizer builds of the product's HOST code (kernels unchanged) and of the oracle, for crash hunting:
#   dust_amd/_asan/libdust_hip.so, dust_amd/_asan/liboracle.so   (run with tools/diag/with_asan.sh <command>)
set -e
cd "$(dirname "$0")/../.."
mkdir -p dust_amd/_asan
/opt/rocm/lib/llvm/bin/clang -fsanitize=address,undefined -shared-libsan -O1 -g -fPIC -shared -std=c11 -ffp-contract=off oracle/*.c -lm -o dust_amd/_asan/liboracle.so
cd dust_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -ffp-contract=off -fsanitize=address -fno-gpu-sanitize -shared-libsan -I . \
  kernels.hip gi.hip radix.hip edit.hip denoise.hip comm.hip capi.cpp vdb.cpp vox.cpp png.cpp sky.cpp -lz -ldl -o ../_asan/libdust_hip.so
