// png.hpp -- PNG / APNG -> sliced image array, the host-side counterpart of the reference's PngLoader
// (crates/rhyolite_bevy/src/loaders/png.rs:70-200), which feeds the six spatiotemporal blue-noise textures
// (crates/render/src/noise.rs:16-29: 128 x 128 x 64 APNGs) to the shaders.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace dust::png {

struct ImageArray {
  uint32_t width = 0, height = 0, layers = 0;
  uint32_t channels = 0;           // as stored for the GPU: 1 (grey), 2 (grey + alpha), 4 (RGB padded with 0, or RGBA)
  uint32_t bytes_per_channel = 0;  // 1 or 2 (16-bit samples stay big-endian, as the reference copies them)
  std::vector<uint8_t> texels;     // layers x height x width x channels x bytes_per_channel
};

// Throws dust::vox::ParseError (unsupported = true for what the reference rejects or cannot express:
// indexed colour, 1/2/4-bit samples, interlacing, frames that do not cover the whole image).
ImageArray load(const uint8_t* bytes, size_t n);

}  // namespace dust::png
