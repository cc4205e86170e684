// sky.hpp -- Sunlight::bake (product host code): the Hosek-Wilkie sky / sun state the shading kernels evaluate.
//
// Stands in for crates/render/src/pipeline/sky.rs:90-268. The reference embeds the model's coefficient tables with
// include_bytes! (dataset.bin: 1200 x vec3, datasetSolar.bin: 1806 x vec3); this library does not ship them -- the host
// hands their bytes over once (dust_sky_dataset_create) and bakes as often as the sun moves.
#pragma once
#include <cstddef>
#include <cstdint>
#include <vector>

namespace dust::sky {

struct Dataset {
  // sky.rs:25-64. All vec3 = (X, Y, Z) channels.
  std::vector<float> config;  // [albedo 0/1][turbidity 10][coefficient 9][control point 6][3]
  std::vector<float> rad;     // [albedo 0/1][turbidity 10][control point 6][3]
  std::vector<float> solar;   // [turbidity 10][piece 45][order 4][3]
  float ld[6][3];             // limb darkening coefficients
};
constexpr size_t kDatasetBytes = 1200 * 12, kSolarBytes = 1806 * 12;

// false when the sizes are not those of the reference's files
bool load_dataset(const uint8_t* dataset, size_t n_dataset, const uint8_t* solar, size_t n_solar, Dataset& out);
// Sunlight::bake (sky.rs:90-132). turbidity in [1, 10], direction a unit vector with y > 0 (above the horizon).
bool bake(const Dataset& d, float turbidity, const float albedo[3], const float direction[3], float out[56]);

}  // namespace dust::sky
