#!/bin/bash
# Builds the library of a git revision (default HEAD) as dust_amd/libdust_hip_base.so, for A/B runs (tools/ab.sh) against the
# working tree's build. Delete it afterwards: it travels with every gpurun push.
set -e
cd "$(dirname "$0")/.."
rev=${1:-HEAD}
tmp=$(mktemp -d)
git archive "$rev" dust_amd/csrc include | tar -x -C "$tmp"
make -s -j8 -C "$tmp/dust_amd/csrc" LIB="$PWD/dust_amd/libdust_hip_base.so" 2>/dev/null || {
  # older revisions have no Makefile: one hipcc call
  (cd "$tmp/dust_amd/csrc" && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -I . \
     kernels.hip radix.hip edit.hip denoise.hip capi.cpp vdb.cpp vox.cpp png.cpp sky.cpp -lz -o "$OLDPWD/dust_amd/libdust_hip_base.so")
}
rm -rf "$tmp"
ls -la dust_amd/libdust_hip_base.so
