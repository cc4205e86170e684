"""dust_amd -- MI355X-native ray-tracing hot path of the Dust voxel engine behind a C ABI.

The product is dust_amd/libdust_hip.so (HIP kernels + C ABI, sources in dust_amd/csrc, header in
include/dust_hip.h). This package only loads it (dust_amd._lib), wraps handles for tests and the
bench (dust_amd.api) and generates synthetic stand-in assets (dust_amd.synth).
"""
__all__ = ["api", "synth", "_lib"]
