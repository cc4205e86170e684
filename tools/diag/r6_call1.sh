cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q -p no:cacheprovider > gpurun_out/r06a_gpu_tests.log 2>&1; tail -4 gpurun_out/r06a_gpu_tests.log
S=DUST_HIP_NO_SIDE_STREAM=1
ROUNDS=2 WORKLOADS=gi STEPS=60 bash tools/ab.sh dust_amd/libdust_hip_base.so:$S dust_amd/libdust_hip_plane.so:$S dust_amd/libdust_hip_t16.so:$S dust_amd/libdust_hip.so:$S dust_amd/libdust_hip_t64.so:$S > gpurun_out/r06a_ab_gi_inplace.log 2>&1
cat gpurun_out/r06a_ab_gi_inplace.log
for v in _base _plane "" _t64; do echo "== lib$v"; DUST_HIP_LIB=$PWD/dust_amd/libdust_hip$v.so python tools/diag/surfel_items.py 40 2>&1 | tail -8; done > gpurun_out/r06a_surfel_items.log 2>&1
cat gpurun_out/r06a_surfel_items.log
ROUNDS=2 WORKLOADS=gi STEPS=60 bash tools/ab.sh dust_amd/libdust_hip_base.so dust_amd/libdust_hip.so > gpurun_out/r06a_ab_gi.log 2>&1
cat gpurun_out/r06a_ab_gi.log
