// denoise.hpp -- interface between the host runtime (capi.cpp) and the spatiotemporal accumulation kernels (denoise.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "dust_dev.h"

namespace dust {

struct DenoiseArgs {
  // this frame's G-buffer (standard.rs:974-1050)
  const uint16_t* illuminance;  // YCoCg + hit distance, 4 halves / px (the noisy signal; nrd.rs IN_DIFF_RADIANCE_HITDIST)
  uint16_t* denoised;           // same layout (OUT_DIFF_RADIANCE_HITDIST); primary misses already hold the sky
  const uint32_t* normal;       // IN_NORMAL_ROUGHNESS
  const float* depth;           // IN_VIEWZ: the primary ray's t (hit.rchit:81), +inf on a miss
  const uint16_t* motion;       // IN_MV, world space (nrd.rs:763)
  const uint32_t* voxel_id;     // low 16 bits: instance
  // history: previous frame in, this frame out (two sets used in turn)
  const float* hist_in_accum;   // rgb radiance + accumulated frame count (16 B / px)
  const uint32_t* hist_in_geo;  // what the taps are tested against, ONE 16-byte record / px: {depth bits, packed normal, instance, 0}
  float* hist_out_accum;
  uint32_t* hist_out_geo;
  DevCamera cam, prev;
  uint32_t have_history, width, height, frame_index;
  float aspect;
  float max_frames;     // ReblurSettings::maxAccumulatedFrameNum (NRD default 30)
  float disocclusion;   // CommonSettings::disocclusion_threshold (NRD default 0.01)
  float antilag_sigma;  // ReblurAntilagSettings::luminance_sigma_scale (nrd.rs:777: 2.0)
  float antilag_power;  // ReblurAntilagSettings::luminance_antilag_power (nrd.rs:778: 0.8)
  float max_radius;     // ReblurSettings::blurRadius in pixels (NRD default 15)
};

hipError_t launch_denoise(const DenoiseArgs& a, hipStream_t s);

}  // namespace dust
