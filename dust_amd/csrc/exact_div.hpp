// exact_div.hpp -- IEEE-exact fp32 quotients in 3-4 instructions where the divisor's reciprocal can be shared or folded
// (device code only; tests/cpp/division_identity_test.c checks the identities on whole numerator sets).
#pragma once

namespace dust {

// a / b, correctly rounded, given y = RN(1 / b) (an IEEE division done once per instance and axis): Markstein's
// sequence q0 = RN(a y), r = a - b q0 (exact in an FMA), q = RN(q0 + r y) yields RN(a / b) whenever nothing under- or
// overflows. b == 0 (y infinite) takes q0 = a * (+-inf), which is what a / (+-0) is, NaN for a == 0 included.
// Four instructions against the ten of the hardware division sequence, bit for bit the same quotient; direction
// components in the denormal range (1 / b overflowing) are the one input class where it would differ.
__device__ __forceinline__ float div_by(float a, float b, float y, bool y_inf) {
  const float q0 = a * y;
  const float r = __builtin_fmaf(-b, q0, a);
  const float q = __builtin_fmaf(r, y, q0);
  return y_inf ? q0 : q;
}
// The same sequence wherever the shaders divide: by a constant (y folds at compile time), or several numerators by one
// divisor (one IEEE reciprocal instead of a division each). y zero, infinite or NaN (a divisor that is infinite, zero
// or NaN) takes a * y, which is a / b in those cases too.
__device__ __forceinline__ float div_const(float a, float c) {  // c: a literal
  const float y = 1.0f / c;
  const float q0 = a * y;
  return __builtin_fmaf(__builtin_fmaf(-c, q0, a), y, q0);
}
__device__ __forceinline__ bool recip_special(float y) { return __builtin_amdgcn_classf(y, 0x267); }  // NaN, +-inf, +-0

}  // namespace dust
