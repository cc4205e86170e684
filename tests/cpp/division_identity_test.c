/* The brick slab test divides six numerators by three divisors per brick. The kernels (kernels.hip, div_by) replace
 * each IEEE division a / b by Markstein's sequence on y = RN(1 / b):  q0 = RN(a y), r = a - b q0 (one FMA, exact),
 * q = RN(q0 + r y), with q0 itself when y is infinite (b == +-0). This program checks on the CPU, where both sides are
 * IEEE binary32 exactly as on the GPU, that the sequence returns the bits of a / b: random significands and exponents,
 * weighted towards the significands that are hard for reciprocal-based division (all ones, all zeros, neighbours).
 * usage: division_identity_test [samples]   -> prints "bad 0" and exits 0 when every quotient matches. */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }
static uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static uint64_t next(uint64_t* s) { *s ^= *s << 13; *s ^= *s >> 7; *s ^= *s << 17; return *s; }
static float div_by(float a, float b, float y) {
  const float q0 = a * y;
  const float r = fmaf(-b, q0, a);
  const float q = fmaf(r, y, q0);
  return isinf(y) ? q0 : q;
}
static int same(float x, float y) { return f2u(x) == f2u(y) || (x != x && y != y); }

int main(int argc, char** argv) {
  const long long n = argc > 1 ? atoll(argv[1]) : 20000000LL;
  static const uint32_t hard[] = {0x7FFFFFu, 0u, 0x7FFFFEu, 1u, 0x400000u, 0x3FFFFFu};
  uint64_t s = 0x9E3779B97F4A7C15ull;
  long long bad = 0;
  for (long long i = 0; i < n; ++i) {
    const uint64_t r = next(&s);
    uint32_t mb = (uint32_t)(r & 0x7FFFFFu), ma = (uint32_t)((r >> 23) & 0x7FFFFFu);
    const unsigned sb = (unsigned)((r >> 46) & 15u), sa = (unsigned)((r >> 50) & 15u);
    if (sb < 6) mb = hard[sb];
    if (sa < 6) ma = hard[sa];
    const int eb = 127 - 20 + (int)((r >> 54) % 41u), ea = 127 - 20 + (int)((r >> 58) % 41u);  /* 2^-20 .. 2^20 */
    const float b = u2f(((uint32_t)((r >> 62) & 1u) << 31) | ((uint32_t)eb << 23) | mb);
    const float a = u2f(((uint32_t)((r >> 63) & 1u) << 31) | ((uint32_t)ea << 23) | ma);
    volatile float y = 1.0f / b, want = a / b;
    if (!same(div_by(a, b, y), want)) {
      if (bad < 5) printf("a=%a b=%a got %a want %a\n", a, b, div_by(a, b, y), (float)want);
      ++bad;
    }
  }
  static const float num[] = {0.0f, -0.0f, 1.0f, -1.0f, 4.0f, 3.5f, -2.25f}, zero[] = {0.0f, -0.0f};
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 2; ++j) {
      volatile float y = 1.0f / zero[j], want = num[i] / zero[j];
      if (!same(div_by(num[i], zero[j], y), want)) { printf("%g / %g wrong\n", num[i], zero[j]); ++bad; }
    }
  /* The shaders' other divisions take the same sequence (div_const, camera_ray_dir in kernels.hip); their numerators come
   * from small sets, checked here one by one: bytes / 255, 10-bit fields / 1023, material ids / 3, and the pixel centres
   * (px + 0.5) / size for every frame size up to 8192. */
  {
    static const float cs[] = {255.0f, 1023.0f, 3.0f};
    static const int lim[] = {256, 1024, 256};
    for (int c = 0; c < 3; ++c)
      for (int x = 0; x < lim[c]; ++x) {
        volatile float y = 1.0f / cs[c], want = (float)x / cs[c];
        if (!same(div_by((float)x, cs[c], y), want)) { printf("%d / %g wrong\n", x, cs[c]); ++bad; }
      }
    for (int w = 1; w <= 8192; ++w) {
      volatile float y = 1.0f / (float)w;
      for (int px = 0; px < w; ++px) {
        volatile float want = ((float)px + 0.5f) / (float)w;
        if (!same(div_by((float)px + 0.5f, (float)w, y), want)) { if (bad < 5) printf("(%d + 0.5) / %d wrong\n", px, w); ++bad; }
      }
    }
  }
  /* The brick DDA's axis predicates (kernels.hip, brick_intersect): step(tMax.xyz, tMax.zxy) * step(tMax.xyz, tMax.yzx),
   * i.e. !(z < x) & !(y < x) per component, against the form the kernel evaluates, !(min3(x, y, z) < x) with a NaN-ignoring
   * minimum: every combination of NaN, infinities, signed zeros and finite values. */
  {
    static const float v[] = {NAN, -INFINITY, -1.0f, -0.0f, 0.0f, 1.0f, 2.0f, INFINITY};
    for (int i = 0; i < 8; ++i)
      for (int j = 0; j < 8; ++j)
        for (int k = 0; k < 8; ++k) {
          const float x = v[i], y = v[j], z = v[k];
          const int bx = !(z < x) & !(y < x), by = !(x < y) & !(z < y), bz = !(y < z) & !(x < z);
          const float hd = fminf(fminf(x, y), z);
          if (bx != !(hd < x) || by != !(hd < y) || bz != !(hd < z)) { printf("axis predicates differ at %g %g %g\n", x, y, z); ++bad; }
        }
  }
  printf("samples %lld bad %lld\n", n, bad);
  return bad == 0 ? 0 : 1;
}
