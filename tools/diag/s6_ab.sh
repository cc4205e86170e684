cd $GRAFT_REPO_ROOT
export ROUNDS=2 WORKLOADS=gi STEPS=60
echo "== side stream"; bash tools/ab.sh dust_amd/libdust_hip.so dust_amd/libdust_hip_scan.so dust_amd/libdust_hip_sunkey.so dust_amd/libdust_hip_both.so
echo "== in place"; DUST_HIP_NO_SIDE_STREAM=1 bash tools/ab.sh dust_amd/libdust_hip.so dust_amd/libdust_hip_scan.so dust_amd/libdust_hip_sunkey.so dust_amd/libdust_hip_both.so
echo "== parity (both)"; DUST_HIP_LIB=$PWD/dust_amd/libdust_hip_both.so timeout 600 python -m pytest tests/test_gpu_gi.py tests/test_gpu_fullsize.py -m gpu -x -q 2>&1 | tail -3
