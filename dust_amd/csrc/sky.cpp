// sky.cpp -- see sky.hpp. Single precision throughout, operation for operation as sky.rs evaluates it (the state feeds
// fp16 radiance planes compared at 1e-3, but the baked floats themselves are compared bit for bit with the fixtures).
#include "sky.hpp"

#include <cmath>
#include <cstring>

namespace dust::sky {
namespace {

struct V3 { float x, y, z; };
inline V3 operator*(float s, V3 v) { return {s * v.x, s * v.y, s * v.z}; }
inline V3 operator*(V3 a, V3 b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 at(const float* p) { return {p[0], p[1], p[2]}; }

// f32::powi with a constant exponent: square-and-multiply from the low bit, as LLVM expands the intrinsic
// (and compiler-rt's __powisf2 evaluates it): x^4 = (x^2)^2, x^5 = x * (x^2)^2 -- not x*x*x*x*x.
inline float powi(float a, int b) {
  float r = 1.0f;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return r;
}

// sky.rs:135-143: quintic Bezier over six control points
V3 coefficient(const float* m, float e) {
  const float rev = 1.0f - e;
  V3 r = powi(rev, 5) * at(m);
  r = r + ((5.0f * powi(rev, 4)) * e) * at(m + 3);
  r = r + ((10.0f * powi(rev, 3)) * powi(e, 2)) * at(m + 6);
  r = r + ((10.0f * powi(rev, 2)) * powi(e, 3)) * at(m + 9);
  r = r + ((5.0f * rev) * powi(e, 4)) * at(m + 12);
  r = r + powi(e, 5) * at(m + 15);
  return r;
}

// the four-corner blend shared by cook_radiance_config (sky.rs:145-179) and cook_config (sky.rs:181-227);
// low / high: control points of albedo 0 / 1, `stride` floats from one turbidity to the next
V3 blend(const float* low, const float* high, size_t stride, float turbidity, V3 albedo, float elevation) {
  const int it = static_cast<int>(turbidity);
  const float rem = turbidity - static_cast<float>(it);
  const float e = std::pow(elevation / 1.57079632679489661923f, 1.0f / 3.0f);
  const V3 one_minus = {1.0f - albedo.x, 1.0f - albedo.y, 1.0f - albedo.z};
  V3 res = ((1.0f - rem) * one_minus) * coefficient(low + size_t(it - 1) * stride, e);
  res = res + ((1.0f - rem) * albedo) * coefficient(high + size_t(it - 1) * stride, e);
  if (it < 10) {
    res = res + (rem * one_minus) * coefficient(low + size_t(it) * stride, e);
    res = res + (rem * albedo) * coefficient(high + size_t(it) * stride, e);
  }
  return res;
}

// sky.rs:229-254
V3 solar_internal(const Dataset& d, uint32_t turbidity, float elevation) {
  const uint32_t pieces = 45, order = 4;
  uint32_t pos = static_cast<uint32_t>(std::pow(2.0f * elevation / 3.14159265358979323846f, 1.0f / 3.0f) * float(pieces));
  if (pos > pieces - 1) pos = pieces - 1;
  const float break_x = powi(float(pos) / float(pieces), 3) * 1.57079632679489661923f;
  const float x = elevation - break_x;
  float x_exp = 1.0f;
  V3 res = {0.0f, 0.0f, 0.0f};
  const float* coefs = d.solar.data() + size_t(order * pieces * turbidity + order * pos) * 3;
  for (int k = int(order) - 1; k >= 0; --k) {
    res = res + x_exp * at(coefs + k * 3);
    x_exp *= x;
  }
  return res;
}

}  // namespace

bool load_dataset(const uint8_t* dataset, size_t n_dataset, const uint8_t* solar, size_t n_solar, Dataset& out) {
  if (n_dataset != kDatasetBytes || n_solar != kSolarBytes) return false;
  out.config.resize(1080 * 3);
  out.rad.resize(120 * 3);
  out.solar.resize(1800 * 3);
  std::memcpy(out.config.data(), dataset, 1080 * 12);             // sky.rs:38-47
  std::memcpy(out.rad.data(), dataset + 1080 * 12, 120 * 12);     // sky.rs:49-54
  std::memcpy(out.solar.data(), solar, 1800 * 12);                // sky.rs:60-61
  std::memcpy(out.ld, solar + 1800 * 12, 6 * 12);                 // sky.rs:62-63
  return true;
}

bool bake(const Dataset& d, float turbidity, const float albedo[3], const float direction[3], float out[56]) {
  if (!(turbidity >= 1.0f && turbidity <= 10.0f)) return false;  // sky.rs:256 assert!
  if (!(direction[1] > 0.0f && direction[1] <= 1.0f)) return false;  // the cube root of a negative elevation is NaN
  const V3 alb = at(albedo);
  const float elevation = std::asin(direction[1]);
  // cook_config: [albedo][turbidity][coefficient i][6][3] -> per coefficient, stride between turbidities = 9 * 6 * 3
  for (int i = 0; i < 9; ++i) {
    const V3 c = blend(d.config.data() + size_t(i) * 18, d.config.data() + 540 * 3 + size_t(i) * 18, 9 * 18, turbidity, alb, elevation);
    out[0 + i] = c.x; out[16 + i] = c.y; out[32 + i] = c.z;
  }
  const V3 rad = blend(d.rad.data(), d.rad.data() + 60 * 3, 18, turbidity, alb, elevation);
  out[9] = rad.x; out[25] = rad.y; out[41] = rad.z;
  for (int c = 0; c < 3; ++c)
    for (int k = 0; k < 6; ++k) out[c * 16 + 10 + k] = d.ld[k][c];  // ld_coefficient0, 1, then the vec4 of 2..5
  out[48] = direction[0]; out[49] = direction[1]; out[50] = direction[2]; out[51] = 0.0f;
  // arhosekskymodel_solar_direct_radiance_xyz (sky.rs:255-268)
  uint32_t turb_low = static_cast<uint32_t>(turbidity) - 1;
  float turb_frac = turbidity - float(turb_low + 1);
  if (turb_low == 9) { turb_low = 8; turb_frac = 1.0f; }
  const V3 sol = (1.0f - turb_frac) * solar_internal(d, turb_low, elevation) + turb_frac * solar_internal(d, turb_low + 1, elevation);
  out[52] = sol.x; out[53] = sol.y; out[54] = sol.z;
  out[55] = (0.51f * (3.14159265358979323846f / 180.0f)) / 2.0f;
  return true;
}

}  // namespace dust::sky
