/*
 * shade.c -- oracle restatement of Dust's ray-tracing shaders on the CPU (TEST INFRASTRUCTURE ONLY).
 *
 * Follows assets/shaders/{primary,final_gather,headers}/ * of the reference; every function cites the
 * lines it restates. fp32 throughout, no FMA contraction (build with -ffp-contract=off), GLSL
 * min/max are IEEE minNum/maxNum (fminf/fmaxf): the reference's default sun direction has x == 0,
 * so its own DDA only terminates on hardware with that behaviour (hit.rint:87-96, sky.rs:20).
 *
 * PARITY UNPINNED vs real Vulkan output (no reference test exists for any shader); see oracle.h.
 *
 * Closest-hit semantics (the Vulkan traversal the reference delegates to):
 * over ALL bricks of ALL instances, a brick reports (t, attr) through the intersection routine,
 * the report is accepted iff tmin <= t <= current tmax, acceptance shrinks tmax. Equal-t reports
 * from two bricks are order-dependent in Vulkan; here the tie is broken towards the lower
 * (instance, block) pair so that every traversal order gives the same answer.
 */
#include "oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline float gsign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
static inline float gstep(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
static inline float gclamp(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
static inline int f2i_sat(float f) { /* float->int as AMD v_cvt_i32_f32: NaN -> 0, saturating */
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}
static inline float dot3(v3 a, v3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static inline v3 normalize3(v3 v) {
  float l = sqrtf(dot3(v, v));
  return V3(v.x / l, v.y / l, v.z / l);
}

/* ------------------------------------------------------------------ formats */
uint16_t orc_f32_to_f16(float f) { /* round-to-nearest-even, IEEE binary16 */
  uint32_t x;
  memcpy(&x, &f, 4);
  uint32_t sign = (x >> 16) & 0x8000u;
  uint32_t ax = x & 0x7FFFFFFFu;
  if (ax >= 0x7F800000u) return (uint16_t)(sign | (ax > 0x7F800000u ? 0x7E00u : 0x7C00u));
  if (ax >= 0x477FF000u) return (uint16_t)(sign | 0x7C00u); /* >= 65520 rounds to inf */
  if (ax < 0x33000001u) return (uint16_t)sign;              /* < 2^-25 (or == 2^-25 ties to even 0) */
  int e = (int)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7FFFFFu) | 0x800000u;
  if (e < -14) { /* subnormal half */
    int shift = -14 - e + 13;
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (q & 1))) q += 1;
    return (uint16_t)(sign | q);
  }
  uint32_t q = ((uint32_t)(e + 15) << 10) | ((m >> 13) & 0x3FFu);
  uint32_t rem = m & 0x1FFFu;
  if (rem > 0x1000u || (rem == 0x1000u && (q & 1))) q += 1;
  return (uint16_t)(sign | q);
}
float orc_f16_to_f32(uint16_t h) {
  uint32_t sign = (uint32_t)(h & 0x8000u) << 16, e = (h >> 10) & 0x1Fu, m = h & 0x3FFu, x;
  if (e == 0) {
    if (m == 0) x = sign;
    else {
      int sh = 0;
      while (!(m & 0x400u)) { m <<= 1; ++sh; }
      x = sign | ((uint32_t)(127 - 15 - sh + 1) << 23) | ((m & 0x3FFu) << 13);
    }
  } else if (e == 31) x = sign | 0x7F800000u | (m << 13);
  else x = sign | ((e - 15 + 127) << 23) | (m << 13);
  float f;
  memcpy(&f, &x, 4);
  return f;
}
static uint32_t unorm(float v, float scale) { /* float -> UNORM, round to nearest even (Vulkan spec 3.9.x) */
  if (!(v > 0.0f)) return 0; /* NaN and negatives -> 0 */
  if (v >= 1.0f) return (uint32_t)scale;
  return (uint32_t)rintf(v * scale);
}
/* VK_FORMAT_A2B10G10R10_UNORM_PACK32: r bits 0-9, g 10-19, b 20-29, a 30-31 (standard.rs:974-1050) */
uint32_t orc_pack_rgb10a2(const float v[4]) {
  return unorm(v[0], 1023.0f) | (unorm(v[1], 1023.0f) << 10) | (unorm(v[2], 1023.0f) << 20) | (unorm(v[3], 3.0f) << 30);
}
void orc_unpack_rgb10a2(uint32_t p, float v[4]) {
  v[0] = (float)(p & 1023u) / 1023.0f;
  v[1] = (float)((p >> 10) & 1023u) / 1023.0f;
  v[2] = (float)((p >> 20) & 1023u) / 1023.0f;
  v[3] = (float)(p >> 30) / 3.0f;
}

/* ------------------------------------------------------------------ headers/normal.glsl */
uint32_t orc_normal2faceid(const float n[3]) { /* normal.glsl:9-18 */
  float s = gclamp((n[0] + n[1]) + n[2], 0.0f, 1.0f);
  uint32_t face = (uint32_t)(uint8_t)rintf(s);
  uint32_t index = (uint32_t)(uint8_t)rintf(fabsf(n[2])) * 4u + (uint32_t)(uint8_t)rintf(fabsf(n[1])) * 2u;
  return (face + index) & 0xFFu;
}
static v3 faceid2normal(uint32_t face) { /* normal.glsl:20-26 */
  float s = (float)(face & 1u) * 2.0f - 1.0f;
  v3 n = V3(0, 0, 0);
  uint32_t a = (face & 0xFFu) >> 1;
  if (a == 0) n.x = s; else if (a == 1) n.y = s; else if (a == 2) n.z = s;
  return n;
}
static v3 cubed_normalize(v3 d) { /* normal.glsl:39-43 */
  v3 a = V3(fabsf(d.x), fabsf(d.y), fabsf(d.z));
  float mx = fmaxf(a.x, fmaxf(a.y, a.z));
  return V3(gsign(d.x) * gstep(mx, a.x), gsign(d.y) * gstep(mx, a.y), gsign(d.z) * gstep(mx, a.z));
}
void orc_cubed_normalize(const float d[3], float out[3]) {
  v3 r = cubed_normalize(V3(d[0], d[1], d[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
static v3 rotate_by_normal(v3 n, v3 t) { /* normal.glsl:31-37 */
  float qx = -n.y, qy = n.x, qz = 0.0f, qw = 1.0f + n.z;
  float l = sqrtf(((qx * qx + qy * qy) + qz * qz) + qw * qw);
  qx /= l; qy /= l; qz /= l; qw /= l;
  if (n.z < -0.99999f) { qx = -1.0f; qy = 0.0f; qz = 0.0f; qw = 0.0f; }
  v3 q = V3(qx, qy, qz);
  float two_dot = 2.0f * dot3(q, t);
  float k = qw * qw - dot3(q, q);
  v3 c = V3(q.y * t.z - t.y * q.z, q.z * t.x - t.z * q.x, q.x * t.y - t.x * q.y);
  float tw = 2.0f * qw;
  return V3((two_dot * q.x + k * t.x) + tw * c.x, (two_dot * q.y + k * t.y) + tw * c.y,
            (two_dot * q.z + k * t.z) + tw * c.z);
}
void orc_rotate_by_normal(const float n[3], const float v[3], float out[3]) {
  v3 r = rotate_by_normal(V3(n[0], n[1], n[2]), V3(v[0], v[1], v[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------ headers/nrd.glsl */
static void nrd_encode_unit_vector(v3 v, float out[2]) { /* nrd.glsl:2-10, bSigned = false */
  float s = (fabsf(v.x) + fabsf(v.y)) + fabsf(v.z);
  v.x /= s; v.y /= s; v.z /= s;
  float wx = (1.0f - fabsf(v.y)) * (gstep(0.0f, v.x) * 2.0f - 1.0f);
  float wy = (1.0f - fabsf(v.x)) * (gstep(0.0f, v.y) * 2.0f - 1.0f);
  float ex = v.z >= 0.0f ? v.x : wx, ey = v.z >= 0.0f ? v.y : wy;
  out[0] = ex * 0.5f + 0.5f;
  out[1] = ey * 0.5f + 0.5f;
}
void orc_nrd_pack_normal(const float n[3], float roughness, float material_id, float out[4]) { /* nrd.glsl:25-52 */
  nrd_encode_unit_vector(V3(n[0], n[1], n[2]), out);
  out[2] = roughness;
  out[3] = gclamp(material_id / 3.0f, 0.0f, 1.0f);
}
void orc_nrd_unpack_normal(const float p[4], float out_n[3]) { /* nrd.glsl:54-94 */
  float px = p[0] * 2.0f - 1.0f, py = p[1] * 2.0f - 1.0f;
  v3 n = V3(px, py, (1.0f - fabsf(px)) - fabsf(py));
  float t = gclamp(-n.z, 0.0f, 1.0f);
  n.x -= t * (gstep(0.0f, n.x) * 2.0f - 1.0f);
  n.y -= t * (gstep(0.0f, n.y) * 2.0f - 1.0f);
  n = normalize3(n);
  out_n[0] = n.x; out_n[1] = n.y; out_n[2] = n.z;
}
static void pack_radiance(v3 r, float hitdist, uint16_t out[4]) { /* nrd.glsl:97-147 + RGBA16F store */
  if (hitdist != 0.0f) hitdist = fmaxf(hitdist, 1e-7f);
  float Y = (r.x * 0.25f + r.y * 0.5f) + r.z * 0.25f;
  float Co = (r.x * 0.5f + r.y * 0.0f) + r.z * -0.5f;
  float Cg = (r.x * -0.25f + r.y * 0.5f) + r.z * -0.25f;
  out[0] = orc_f32_to_f16(Y); out[1] = orc_f32_to_f16(Co); out[2] = orc_f32_to_f16(Cg); out[3] = orc_f32_to_f16(hitdist);
}
static void unpack_radiance(const uint16_t in[4], v3* rgb, float* w) { /* nrd.glsl:107-125 */
  float Y = orc_f16_to_f32(in[0]), Co = orc_f16_to_f32(in[1]), Cg = orc_f16_to_f32(in[2]);
  float t = Y - Cg;
  rgb->y = fmaxf(Y + Cg, 0.0f);
  rgb->x = fmaxf(t + Co, 0.0f);
  rgb->z = fmaxf(t - Co, 0.0f);
  *w = orc_f16_to_f32(in[3]);
}

/* ------------------------------------------------------------------ headers/color.glsl (GLSL mat3 ctor is column-major) */
static v3 mat3_mul(const float m[9], v3 v) {
  return V3((m[0] * v.x + m[3] * v.y) + m[6] * v.z, (m[1] * v.x + m[4] * v.y) + m[7] * v.z,
            (m[2] * v.x + m[5] * v.y) + m[8] * v.z);
}
static const float M_XYZ2ACEScg[9] = {1.6410228f, -0.66366285f, 0.011721907f, -0.32480323f, 1.6153315f,
                                      -0.0082844375f, -0.23642465f, 0.016756356f, 0.9883947f}; /* color.glsl:24-31 */
static const float M_ACEScg2XYZ[9] = {0.66245437f, 0.2722288f, -0.0055746622f, 0.13400422f, 0.6740818f,
                                      0.00406073f, 0.15618773f, 0.05368953f, 1.0103393f}; /* color.glsl:32-39 */

/* ------------------------------------------------------------------ headers/sky.glsl */
typedef struct { float cfg[9], radiance, ld0, ld1, ld2[4]; } SkyChan;
static void sky_chan(const OrcSky* s, int c, SkyChan* o) { /* layout.playout:35-51 */
  const float* p = s->v + c * 16;
  memcpy(o->cfg, p, 9 * sizeof(float));
  o->radiance = p[9]; o->ld0 = p[10]; o->ld1 = p[11];
  memcpy(o->ld2, p + 12, 4 * sizeof(float));
}
static float sky_internal(const float c[9], float cos_theta, float gamma, float cos_gamma) { /* sky.glsl:1-15 */
  float expM = expf(c[4] * gamma);
  float rayM = cos_gamma * cos_gamma;
  float mieM = (1.0f + rayM) / powf((1.0f + c[8] * c[8]) - (2.0f * c[8]) * cos_gamma, 1.5f);
  float zenith = sqrtf(cos_theta);
  return (1.0f + c[0] * expf(c[1] / (cos_theta + 0.01f))) *
         ((((c[2] + c[3] * expM) + c[5] * rayM) + c[6] * mieM) + c[7] * zenith);
}
static v3 sky_radiance(const OrcSky* s, v3 dir) { /* sky.glsl:18-79 */
  if (s->v[49] <= 0.0f) return V3(0, 0, 0);
  v3 sd = V3(s->v[48], s->v[49], s->v[50]);
  float cos_theta = gclamp(dir.y, 0.0f, 1.0f);
  float cos_gamma = dot3(dir, sd);
  float gamma = acosf(cos_gamma);
  SkyChan r, g, b;
  sky_chan(s, 0, &r); sky_chan(s, 1, &g); sky_chan(s, 2, &b);
  float x = sky_internal(r.cfg, cos_theta, gamma, cos_gamma) * r.radiance;
  float y = sky_internal(g.cfg, cos_theta, gamma, cos_gamma) * g.radiance;
  float z = sky_internal(b.cfg, cos_theta, gamma, cos_gamma) * b.radiance;
  return mat3_mul(M_XYZ2ACEScg, V3(x * 683.0f, y * 683.0f, z * 683.0f));
}
static v3 sun_radiance(const OrcSky* s, v3 dir) { /* sky.glsl:81-113 */
  v3 sd = V3(s->v[48], s->v[49], s->v[50]);
  float cos_gamma = dot3(dir, sd);
  if (cos_gamma < 0.0f || dir.y < 0.0f) return V3(0, 0, 0);
  float sol_rad_sin = sinf(s->v[55]);
  float ar2 = 1.0f / (sol_rad_sin * sol_rad_sin);
  float singamma = 1.0f - (cos_gamma * cos_gamma);
  float sc2 = 1.0f - (ar2 * singamma) * singamma;
  if (sc2 <= 0.0f) return V3(0, 0, 0);
  float sc = sqrtf(sc2);
  SkyChan c[3];
  for (int i = 0; i < 3; ++i) sky_chan(s, i, &c[i]);
  v3 dark = V3(c[0].ld0, c[1].ld0, c[2].ld0);
  dark.x += c[0].ld1 * sc; dark.y += c[1].ld1 * sc; dark.z += c[2].ld1 * sc;
  float cur = sc;
  for (int i = 0; i < 4; ++i) {
    cur *= sc;
    dark.x += c[0].ld2[i] * cur; dark.y += c[1].ld2[i] * cur; dark.z += c[2].ld2[i] * cur;
  }
  return mat3_mul(M_XYZ2ACEScg, V3(s->v[52] * dark.x, s->v[53] * dark.y, s->v[54] * dark.z));
}
void orc_sky_radiance(const OrcSky* s, const float d[3], float out[3]) {
  v3 r = sky_radiance(s, V3(d[0], d[1], d[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
void orc_sun_radiance(const OrcSky* s, const float d[3], float out[3]) {
  v3 r = sun_radiance(s, V3(d[0], d[1], d[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* ------------------------------------------------------------------ headers/spatial_hash.glsl (codecs + hashes) */
uint32_t orc_pcg(uint32_t v) { /* spatial_hash.glsl:105-111 */
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
uint32_t orc_xxhash32(uint32_t p) { /* spatial_hash.glsl:115-126 */
  const uint32_t P2 = 2246822519u, P3 = 3266489917u, P4 = 668265263u, P5 = 374761393u;
  uint32_t h = p + P5;
  h = P4 * ((h << 17) | (h >> 15));
  h = P2 * (h ^ (h >> 15));
  h = P3 * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
uint32_t orc_logluv_encode(const float rgb[3]) { /* spatial_hash.glsl:28-60 */
  v3 XYZ = mat3_mul(M_ACEScg2XYZ, V3(rgb[0], rgb[1], rgb[2]));
  float logY = 409.6f * (log2f(XYZ.y) + 20.0f);
  float cl = gclamp(logY, 0.0f, 16383.0f);
  uint32_t Le = (cl != cl) ? 0u : (uint32_t)cl;
  if (Le == 0) return 0;
  float invDenom = 1.0f / ((-2.0f * XYZ.x + 12.0f * XYZ.y) + 3.0f * ((XYZ.x + XYZ.y) + XYZ.z));
  float u = (4.0f * XYZ.x) * invDenom, v = (9.0f * XYZ.y) * invDenom;
  float cu = gclamp(820.0f * u, 0.0f, 511.0f), cv = gclamp(820.0f * v, 0.0f, 511.0f);
  uint32_t ue = (cu != cu) ? 0u : (uint32_t)cu, ve = (cv != cv) ? 0u : (uint32_t)cv;
  return (Le << 18) | (ue << 9) | ve;
}
void orc_logluv_decode(uint32_t p, float rgb[3]) { /* spatial_hash.glsl:64-93 */
  uint32_t Le = p >> 18;
  if (Le == 0) { rgb[0] = rgb[1] = rgb[2] = 0.0f; return; }
  float logY = ((float)Le + 0.5f) / 409.6f - 20.0f;
  float Y = powf(2.0f, logY);
  float u = ((float)((p >> 9) & 0x1FFu) + 0.5f) / 820.0f, v = ((float)(p & 0x1FFu) + 0.5f) / 820.0f;
  float invDenom = 1.0f / ((6.0f * u - 16.0f * v) + 12.0f);
  float x = (9.0f * u) * invDenom, y = (4.0f * v) * invDenom;
  float s = Y / y;
  v3 r = mat3_mul(M_XYZ2ACEScg, V3(s * x, Y, s * ((1.0f - x) - y)));
  rgb[0] = fmaxf(r.x, 0.0f); rgb[1] = fmaxf(r.y, 0.0f); rgb[2] = fmaxf(r.z, 0.0f);
}

/* ------------------------------------------------------------------ intersection routines */
#define ORC_DDA_MAX_ITERS 64 /* the reference loop is unbounded; a NaN ray would hang the GPU there */

static int grid_clear(uint32_t m1, uint32_t m2, uint32_t hit) { /* GridCheck, hit.rint:13-15 (u32vec2 path) */
  /* hit can leave 0..63 only after the exit test failed to fire; shifts are taken mod 32 as the hardware does */
  return ((hit < 32u) ? (m1 & (1u << (hit & 31u))) : (m2 & (1u << ((hit - 32u) & 31u)))) == 0;
}
static uint32_t encode_index(int px, int py, int pz) { /* hit.rint:30-32 on u8vec3 */
  uint8_t x = (uint8_t)(int8_t)px, y = (uint8_t)(int8_t)py, z = (uint8_t)(int8_t)pz;
  return (uint32_t)(uint8_t)((uint8_t)(x << 4) | (uint8_t)(y << 2) | z);
}
static void intersect_aabb04(v3 o, v3 d, float* t_min, float* t_max) { /* hit.rint:20-28 with box [0,4]^3 */
  float ax = (0.0f - o.x) / d.x, ay = (0.0f - o.y) / d.y, az = (0.0f - o.z) / d.z;
  float bx = (4.0f - o.x) / d.x, by = (4.0f - o.y) / d.y, bz = (4.0f - o.z) / d.z;
  float t1x = fminf(ax, bx), t1y = fminf(ay, by), t1z = fminf(az, bz);
  float t2x = fmaxf(ax, bx), t2y = fmaxf(ay, by), t2z = fmaxf(az, bz);
  *t_min = fmaxf(fmaxf(t1x, t1y), t1z);
  *t_max = fminf(fminf(t2x, t2y), t2z);
}

/* primary/hit.rint:43-131 (kind 0) and final_gather/ambient_occlusion.rint:46-134 (kind 1).
 * o: brick-local origin (objOrigin - block.position), d: object ray direction.
 * Returns 1 if reportIntersectionEXT is reached. */
int orc_dda(int kind, const float o_[3], const float d_[3], uint32_t m1, uint32_t m2, float tmin, float* t_out,
            uint32_t* voxel_out, int* hitkind_out) {
  v3 o = V3(o_[0], o_[1], o_[2]), d = V3(d_[0], d_[1], d_[2]);
  float t0, t1;
  intersect_aabb04(o, d, &t0, &t1);
  if (t0 >= t1) return 0;
  if (t1 <= 0.0f) return 0;
  if (kind == 1) { /* ambient_occlusion.rint:62-73 */
    if (t0 <= 8.0f && 8.0f <= t1) {
      if (!(m1 == 0 && m2 == 0)) {
        *t_out = t0; *voxel_out = 0xFF; *hitkind_out = 1;
        return 1;
      }
      return 0;
    }
  }
  float hd = fmaxf(t0, tmin);
  v3 p = V3(o.x + d.x * hd, o.y + d.y * hd, o.z + d.z * hd);
  int px = f2i_sat(floorf(p.x)), py = f2i_sat(floorf(p.y)), pz = f2i_sat(floorf(p.z));
  px = px < 0 ? 0 : (px > 3 ? 3 : px);
  py = py < 0 ? 0 : (py > 3 ? 3 : py);
  pz = pz < 0 ? 0 : (pz > 3 ? 3 : pz);
  v3 st = V3(gsign(d.x), gsign(d.y), gsign(d.z));
  v3 tc = V3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  v3 tb = V3(tc.x * o.x, tc.y * o.y, tc.z * o.z);
  v3 tm = V3(((float)px + fmaxf(st.x, 0.0f)) * tc.x - tb.x, ((float)py + fmaxf(st.y, 0.0f)) * tc.y - tb.y,
             ((float)pz + fmaxf(st.z, 0.0f)) * tc.z - tb.z);
  v3 td = V3((1.0f * tc.x) * st.x, (1.0f * tc.y) * st.y, (1.0f * tc.z) * st.z);
  uint32_t hit = encode_index(px, py, pz);
  int iters = 0;
  while (grid_clear(m1, m2, hit)) {
    if (++iters > ORC_DDA_MAX_ITERS) return 0;
    float cx = gstep(tm.x, tm.z) * gstep(tm.x, tm.y);
    float cy = gstep(tm.y, tm.x) * gstep(tm.y, tm.z);
    float cz = gstep(tm.z, tm.y) * gstep(tm.z, tm.x);
    px = (int8_t)(px + (int8_t)f2i_sat(st.x * cx));
    py = (int8_t)(py + (int8_t)f2i_sat(st.y * cy));
    pz = (int8_t)(pz + (int8_t)f2i_sat(st.z * cz));
    hd = fminf(fminf(tm.x, tm.y), tm.z);
    if (hd + 0.001f >= t1) return 0;
    tm.x += td.x * cx; tm.y += td.y * cy; tm.z += td.z * cz;
    hit = encode_index(px, py, pz);
  }
  *t_out = hd / 1.0f;
  *voxel_out = hit;
  *hitkind_out = 0;
  return 1;
}

/* final_gather/rough.rint:42-59 */
int orc_dda_rough(const float o_[3], const float d_[3], uint32_t m1, uint32_t m2, float* t_out) {
  float t0, t1;
  intersect_aabb04(V3(o_[0], o_[1], o_[2]), V3(d_[0], d_[1], d_[2]), &t0, &t1);
  if (t0 >= t1) return 0;
  if (m1 == 0 && m2 == 0) return 0;
  *t_out = t0;
  return 1;
}

/* ------------------------------------------------------------------ scene */
typedef struct SceneModel {
  const OrcBlock* blocks;
  uint32_t n_blocks;
  const uint8_t* materials;
  uint64_t n_materials;
  const uint8_t* palette;
  uint32_t extent;
  /* hierarchical mode: sorted keys of occupied cells per level (cell log2 sizes, coarse -> fine) */
  int n_lv;
  uint32_t lv_log2[4];
  uint64_t* lv_keys[4];
  uint32_t lv_n[4];
  uint64_t* brick_keys; /* sorted (key, block index) */
  uint32_t* brick_idx;
  float bmin[3], bmax[3]; /* tight object-space bounds of the bricks */
} SceneModel;

typedef struct SceneInst {
  OrcInstance in;
  float w2o[12]; /* inverse of obj_to_world, 3x4 row-major */
  float wmin[3], wmax[3];
} SceneInst;

struct OrcScene {
  SceneModel* models; uint32_t n_models, cap_models;
  SceneInst* insts; uint32_t n_insts, cap_insts;
};

OrcScene* orc_scene_new(void) { return (OrcScene*)calloc(1, sizeof(OrcScene)); }
void orc_scene_free(OrcScene* s) {
  if (!s) return;
  for (uint32_t i = 0; i < s->n_models; ++i) {
    for (int l = 0; l < 4; ++l) free(s->models[i].lv_keys[l]);
    free(s->models[i].brick_keys); free(s->models[i].brick_idx);
  }
  free(s->models); free(s->insts); free(s);
}
uint32_t orc_scene_add_model(OrcScene* s, const OrcBlock* blocks, uint32_t n_blocks, const uint8_t* materials,
                             uint64_t n_materials, const uint8_t* palette, uint32_t extent) {
  if (s->n_models == s->cap_models) {
    s->cap_models = s->cap_models ? s->cap_models * 2 : 8;
    s->models = (SceneModel*)realloc(s->models, s->cap_models * sizeof(SceneModel));
  }
  SceneModel* m = &s->models[s->n_models];
  memset(m, 0, sizeof(*m));
  m->blocks = blocks; m->n_blocks = n_blocks; m->materials = materials; m->n_materials = n_materials;
  m->palette = palette; m->extent = extent;
  return s->n_models++;
}

/* inverse of a 3x4 affine transform, evaluated in double and rounded once (the product computes its own) */
static void invert_affine(const float m[12], float out[12]) {
  double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  double det = a * A + b * B + c * C;
  double id = 1.0 / det;
  double r[9] = {A * id, -(b * i - c * h) * id, (b * f - c * e) * id,
                 B * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                 C * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  double tx = m[3], ty = m[7], tz = m[11];
  for (int k = 0; k < 3; ++k) {
    out[k * 4 + 0] = (float)r[k * 3 + 0]; out[k * 4 + 1] = (float)r[k * 3 + 1]; out[k * 4 + 2] = (float)r[k * 3 + 2];
    out[k * 4 + 3] = (float)(-(r[k * 3 + 0] * tx + r[k * 3 + 1] * ty + r[k * 3 + 2] * tz));
  }
}
uint32_t orc_scene_add_instance(OrcScene* s, const OrcInstance* in) {
  if (s->n_insts == s->cap_insts) {
    s->cap_insts = s->cap_insts ? s->cap_insts * 2 : 16;
    s->insts = (SceneInst*)realloc(s->insts, s->cap_insts * sizeof(SceneInst));
  }
  SceneInst* si = &s->insts[s->n_insts];
  si->in = *in;
  invert_affine(in->obj_to_world, si->w2o);
  return s->n_insts++;
}

static int cmp_u64(const void* a, const void* b) {
  uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : (x > y ? 1 : 0);
}
static uint64_t cell_key(uint32_t x, uint32_t y, uint32_t z) { return ((uint64_t)x << 42) | ((uint64_t)y << 21) | z; }
typedef struct { uint64_t key; uint32_t idx; } KeyIdx;
static int cmp_keyidx(const void* a, const void* b) {
  const KeyIdx* x = (const KeyIdx*)a; const KeyIdx* y = (const KeyIdx*)b;
  return x->key < y->key ? -1 : (x->key > y->key ? 1 : 0);
}

void orc_scene_commit(OrcScene* s) {
  for (uint32_t mi = 0; mi < s->n_models; ++mi) {
    SceneModel* m = &s->models[mi];
    if (m->brick_keys) continue;
    /* cell levels above the 4^3 bricks: 16 (and 256 for 4096^3 trees), as hierarchy!(4,2,2) / (4,4,2,2) */
    m->n_lv = 0;
    if (m->extent > 256) m->lv_log2[m->n_lv++] = 8;
    m->lv_log2[m->n_lv++] = 4;
    KeyIdx* ki = (KeyIdx*)malloc((m->n_blocks ? m->n_blocks : 1) * sizeof(KeyIdx));
    for (int a = 0; a < 3; ++a) { m->bmin[a] = 1e30f; m->bmax[a] = -1e30f; }
    for (uint32_t i = 0; i < m->n_blocks; ++i) {
      const OrcBlock* b = &m->blocks[i];
      ki[i].key = cell_key(b->x >> 2, b->y >> 2, b->z >> 2);
      ki[i].idx = i;
      float p[3] = {(float)b->x, (float)b->y, (float)b->z};
      for (int a = 0; a < 3; ++a) {
        if (p[a] < m->bmin[a]) m->bmin[a] = p[a];
        if (p[a] + 4.0f > m->bmax[a]) m->bmax[a] = p[a] + 4.0f;
      }
    }
    qsort(ki, m->n_blocks, sizeof(KeyIdx), cmp_keyidx);
    m->brick_keys = (uint64_t*)malloc((m->n_blocks ? m->n_blocks : 1) * 8);
    m->brick_idx = (uint32_t*)malloc((m->n_blocks ? m->n_blocks : 1) * 4);
    for (uint32_t i = 0; i < m->n_blocks; ++i) { m->brick_keys[i] = ki[i].key; m->brick_idx[i] = ki[i].idx; }
    free(ki);
    for (int l = 0; l < m->n_lv; ++l) {
      uint64_t* keys = (uint64_t*)malloc((m->n_blocks ? m->n_blocks : 1) * 8);
      uint32_t sh = m->lv_log2[l];
      for (uint32_t i = 0; i < m->n_blocks; ++i) {
        const OrcBlock* b = &m->blocks[i];
        keys[i] = cell_key(b->x >> sh, b->y >> sh, b->z >> sh);
      }
      qsort(keys, m->n_blocks, 8, cmp_u64);
      uint32_t n = 0;
      for (uint32_t i = 0; i < m->n_blocks; ++i)
        if (n == 0 || keys[n - 1] != keys[i]) keys[n++] = keys[i];
      m->lv_keys[l] = keys; m->lv_n[l] = n;
    }
  }
  for (uint32_t ii = 0; ii < s->n_insts; ++ii) {
    SceneInst* si = &s->insts[ii];
    const SceneModel* m = &s->models[si->in.model];
    for (int a = 0; a < 3; ++a) { si->wmin[a] = 1e30f; si->wmax[a] = -1e30f; }
    for (int c = 0; c < 8; ++c) {
      double p[3] = {(c & 1) ? m->bmax[0] : m->bmin[0], (c & 2) ? m->bmax[1] : m->bmin[1], (c & 4) ? m->bmax[2] : m->bmin[2]};
      for (int a = 0; a < 3; ++a) {
        const float* r = si->in.obj_to_world + a * 4;
        double w = (double)r[0] * p[0] + (double)r[1] * p[1] + (double)r[2] * p[2] + (double)r[3];
        double pad = 1e-4 * (fabs(w) + 1.0);
        if ((float)(w - pad) < si->wmin[a]) si->wmin[a] = (float)(w - pad);
        if ((float)(w + pad) > si->wmax[a]) si->wmax[a] = (float)(w + pad);
      }
    }
  }
}

static int find_key(const uint64_t* keys, uint32_t n, uint64_t k) {
  uint32_t lo = 0, hi = n;
  while (lo < hi) {
    uint32_t mid = (lo + hi) >> 1;
    if (keys[mid] < k) lo = mid + 1; else hi = mid;
  }
  return (lo < n && keys[lo] == k) ? (int)lo : -1;
}

/* ------------------------------------------------------------------ tracing */
typedef struct Hit {
  float t;
  uint32_t inst, block, voxel;
  int found;
} Hit;

typedef struct RayCtx {
  int raytype; /* 0 primary dda, 1 ao dda, 2/3 rough */
  int any_hit;
  float tmin, tmax;
  Hit best;
  OrcRayStats* st;
} RayCtx;

static inline v3 xform_point(const float m[12], v3 p) {
  return V3(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
            ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
static inline v3 xform_dir(const float m[12], v3 d) {
  return V3((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z,
            (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
}

/* run the ray type's intersection routine on one brick and apply Vulkan's accept rule */
static int test_brick(RayCtx* rc, const SceneModel* m, uint32_t inst, uint32_t bi, v3 o, v3 d) {
  const OrcBlock* b = &m->blocks[bi];
  float ol[3] = {o.x - (float)b->x, o.y - (float)b->y, o.z - (float)b->z}; /* hit.rint:137-140 */
  float dl[3] = {d.x, d.y, d.z};
  uint32_t m1 = (uint32_t)b->mask, m2 = (uint32_t)(b->mask >> 32);
  float t; uint32_t vox = 0; int hk = 0;
  if (rc->st) rc->st->bricks_tested += 1;
  int rep = rc->raytype <= 1 ? orc_dda(rc->raytype, ol, dl, m1, m2, rc->tmin, &t, &vox, &hk)
                             : orc_dda_rough(ol, dl, m1, m2, &t);
  if (!rep) return 0;
  float cur = rc->best.found ? rc->best.t : rc->tmax;
  if (!(t >= rc->tmin && t <= cur)) return 0;
  if (rc->best.found && t == rc->best.t) { /* deterministic tie-break, see file header */
    if (inst > rc->best.inst || (inst == rc->best.inst && bi >= rc->best.block)) return 0;
  }
  rc->best.found = 1; rc->best.t = t; rc->best.inst = inst; rc->best.block = bi; rc->best.voxel = vox;
  return 1;
}

/* conservative slab test of a ray against a box; returns 0 when the (dilated) interval is empty */
static int slab_box(v3 o, v3 d, const float lo[3], const float hi[3], float* t_enter, float* t_exit) {
  float te = -INFINITY, tx = INFINITY;
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  for (int a = 0; a < 3; ++a) {
    if (dd[a] != 0.0f) {
      float inv = 1.0f / dd[a];
      float t0 = (lo[a] - oo[a]) * inv, t1 = (hi[a] - oo[a]) * inv;
      te = fmaxf(te, fminf(t0, t1));
      tx = fminf(tx, fmaxf(t0, t1));
    } else if (oo[a] < lo[a] - 1e-3f || oo[a] > hi[a] + 1e-3f) {
      return 0;
    }
  }
  *t_enter = te; *t_exit = tx;
  float slack = 1e-5f * (fabsf(te) + fabsf(tx)) + 1e-5f;
  return !(te > tx + slack) && !(tx + slack < 0.0f);
}

/* deepest occupied cell containing voxel ijk: returns block index or -1, *cell_log2 = size of the empty cell.
 * last16: key of the 16-cell whose mid node the caller already holds (the HIP kernel keeps that node in
 * registers), so that the descent statistics count each mid node once per visit run. */
static int find_brick(const SceneModel* m, const int ijk[3], uint32_t* cell_log2, OrcRayStats* st, uint64_t* last16) {
  uint64_t key16 = cell_key((uint32_t)ijk[0] >> 4, (uint32_t)ijk[1] >> 4, (uint32_t)ijk[2] >> 4);
  int cached = last16 && *last16 == key16;
  for (int l = 0; l < m->n_lv; ++l) {
    uint32_t sh = m->lv_log2[l];
    if (find_key(m->lv_keys[l], m->lv_n[l], cell_key((uint32_t)ijk[0] >> sh, (uint32_t)ijk[1] >> sh, (uint32_t)ijk[2] >> sh)) < 0) {
      *cell_log2 = sh;
      return -1;
    }
    if (st && !cached) st->upper_descents += 1;
  }
  if (last16) *last16 = key16;
  int k = find_key(m->brick_keys, m->n_blocks, cell_key((uint32_t)ijk[0] >> 2, (uint32_t)ijk[1] >> 2, (uint32_t)ijk[2] >> 2));
  *cell_log2 = 2;
  if (k < 0) return -1;
  if (st) st->mid_descents += 1;
  return (int)m->brick_idx[k];
}

/* Hierarchical traversal of one instance (object space). Visits, front to back, a SUPERSET of the
 * bricks whose intersection routine can report an accepted hit: cells are walked with exit planes
 * recomputed from integer cell coordinates (no accumulated error), and whenever the walk passes
 * within delta of a brick-grid edge or corner every brick around it is tested as well. */
static void trace_instance_hier(RayCtx* rc, const SceneModel* m, uint32_t inst, v3 o, v3 d) {
  float te, tx;
  if (!slab_box(o, d, m->bmin, m->bmax, &te, &tx)) return;
  const int E = (int)m->extent;
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  float inv[3];
  for (int a = 0; a < 3; ++a) inv[a] = 1.0f / dd[a];
  float t = fmaxf(te, 0.0f);
  if (rc->raytype >= 2) t = fmaxf(t, rc->tmin * (1.0f - 1e-6f));
  int ijk[3];
  int stepped[3] = {0, 0, 0};
  uint64_t last16 = ~(uint64_t)0;
  int blo[3], bhi[3]; /* voxel range that holds bricks (tight bounds, multiples of 4) */
  for (int a = 0; a < 3; ++a) {
    blo[a] = (int)m->bmin[a];
    bhi[a] = (int)m->bmax[a] - 1;
    /* the cell the ray is moving into: floor for d >= 0, ceil - 1 for d < 0 (differs only on a cell plane) */
    float p = oo[a] + dd[a] * t;
    int v = f2i_sat(dd[a] < 0.0f ? ceilf(p) - 1.0f : floorf(p));
    ijk[a] = v < 0 ? 0 : (v > E - 1 ? E - 1 : v);
  }
  for (int guard = 0; guard < 100000; ++guard) {
    float limit = rc->best.found ? rc->best.t : rc->tmax;
    if (t * (1.0f - 2e-6f) > limit) return;
    if (rc->any_hit && rc->best.found) return;
    uint32_t cl;
    int bi = find_brick(m, ijk, &cl, rc->st, &last16);
    if (bi >= 0) test_brick(rc, m, inst, (uint32_t)bi, o, d);
    /* neighbours around a brick-grid edge/corner the ray passes within delta of */
    int near_dir[3] = {0, 0, 0};
    int n_near = 0, n_stepped = 0, n_near_unstepped = 0, n_stepped_near = 0;
    for (int a = 0; a < 3; ++a) {
      float p = oo[a] + dd[a] * t;
      float delta = 1e-6f * ((fabsf(oo[a]) + fabsf(p)) + 16.0f);
      float q = p - (float)(ijk[a] & ~3);
      int b0 = ijk[a] & ~3;
      /* a plane only matters if bricks can exist on its far side (tight bounds) */
      if (stepped[a]) {
        n_stepped++;
        if (dd[a] > 0.0f) { if (b0 - 1 >= blo[a]) near_dir[a] = -1; } else if (b0 + 4 <= bhi[a]) near_dir[a] = 1;
        if (near_dir[a]) n_stepped_near++;
      }
      else if (q <= delta) { if (b0 - 1 >= blo[a]) { near_dir[a] = -1; n_near_unstepped++; } }
      else if (q >= 4.0f - delta) { if (b0 + 4 <= bhi[a]) { near_dir[a] = 1; n_near_unstepped++; } }
      if (near_dir[a]) n_near++;
    }
    if (n_near_unstepped > 0 || n_stepped_near > 1) {
      for (int sub = 1; sub < 8; ++sub) {
        int ok = 1, only_stepped = 1, all_stepped_in = 1, nj[3];
        for (int a = 0; a < 3; ++a) {
          nj[a] = ijk[a];
          if (sub & (1 << a)) {
            if (!near_dir[a]) { ok = 0; break; }
            if (!stepped[a]) only_stepped = 0;
            nj[a] = near_dir[a] < 0 ? (ijk[a] & ~3) - 1 : (ijk[a] & ~3) + 4;
          } else if (stepped[a]) all_stepped_in = 0;
        }
        if (!ok) continue;
        if (only_stepped && all_stepped_in && n_stepped > 0) continue; /* the cell we came from */
        uint32_t cl2;
        int nb = find_brick(m, nj, &cl2, NULL, &last16);
        if (nb >= 0) test_brick(rc, m, inst, (uint32_t)nb, o, d);
      }
    }
    /* leave the cell of size 2^cl that contains ijk */
    int S = 1 << cl;
    int c[3] = {ijk[0] & ~(S - 1), ijk[1] & ~(S - 1), ijk[2] & ~(S - 1)};
    float ta[3], tn = INFINITY;
    for (int a = 0; a < 3; ++a) {
      if (dd[a] != 0.0f) {
        float plane = (float)(dd[a] > 0.0f ? c[a] + S : c[a]);
        ta[a] = (plane - oo[a]) * inv[a];
      } else ta[a] = INFINITY;
      tn = fminf(tn, ta[a]);
    }
    if (!(tn < INFINITY)) return;
    for (int a = 0; a < 3; ++a) {
      if (ta[a] == tn) {
        stepped[a] = 1;
        ijk[a] = dd[a] > 0.0f ? c[a] + S : c[a] - 1;
        if (ijk[a] < 0 || ijk[a] >= E) return;
      } else {
        stepped[a] = 0;
        int v = f2i_sat(floorf(oo[a] + dd[a] * tn));
        ijk[a] = v < c[a] ? c[a] : (v > c[a] + S - 1 ? c[a] + S - 1 : v);
      }
    }
    t = fmaxf(t, tn);
    if (t * (1.0f - 2e-6f) > tx * (1.0f + 1e-5f) + 1e-5f) return;
  }
}

int orc_trace(const OrcScene* s, int mode, int raytype, int any_hit, const float o_[3], const float d_[3], float tmin,
              float tmax, float* t, uint32_t* inst, uint32_t* block, uint32_t* voxel, OrcRayStats* st) {
  RayCtx rc;
  memset(&rc, 0, sizeof(rc));
  rc.raytype = raytype; rc.any_hit = any_hit; rc.tmin = tmin; rc.tmax = tmax; rc.st = st;
  v3 o = V3(o_[0], o_[1], o_[2]), d = V3(d_[0], d_[1], d_[2]);
  if (st) st->rays += 1;
  for (uint32_t ii = 0; ii < s->n_insts; ++ii) {
    const SceneInst* si = &s->insts[ii];
    const SceneModel* m = &s->models[si->in.model];
    if (mode == ORC_MODE_HIER) {
      float te, tx;
      if (!slab_box(o, d, si->wmin, si->wmax, &te, &tx)) continue;
      float limit = rc.best.found ? rc.best.t : tmax;
      if (te * (1.0f - 2e-6f) > limit) continue;
    }
    if (st) st->instances_tested += 1;
    v3 oo = xform_point(si->w2o, o), od = xform_dir(si->w2o, d);
    if (mode == ORC_MODE_BRUTE) {
      for (uint32_t bi = 0; bi < m->n_blocks; ++bi) {
        test_brick(&rc, m, ii, bi, oo, od);
        if (any_hit && rc.best.found) break;
      }
    } else {
      trace_instance_hier(&rc, m, ii, oo, od);
    }
    if (any_hit && rc.best.found) break;
  }
  if (!rc.best.found) return 0;
  if (st) st->hits += 1;
  *t = rc.best.t; *inst = rc.best.inst; *block = rc.best.block; *voxel = rc.best.voxel;
  return 1;
}

/* ------------------------------------------------------------------ camera */
void orc_camera_ray_dir(const OrcCamera* c, uint32_t px, uint32_t py, uint32_t w, uint32_t h, float out[3]) {
  /* camera.glsl:4-16 */
  float nx = ((float)px + 0.5f) / (float)w, ny = ((float)py + 0.5f) / (float)h;
  float cx = 2.0f * nx - 1.0f, cy = 2.0f * ny - 1.0f;
  cy *= -1.0f;
  cx *= (float)w / (float)h;
  cx *= c->tan_half_fov; cy *= c->tan_half_fov;
  float cz = -1.0f;
  out[0] = (c->col0[0] * cx + c->col1[0] * cy) + c->col2[0] * cz;
  out[1] = (c->col0[1] * cx + c->col1[1] * cy) + c->col2[1] * cz;
  out[2] = (c->col0[2] * cx + c->col1[2] * cy) + c->col2[2] * cz;
}

/* ------------------------------------------------------------------ primary pass */
static void put_half4(uint16_t* plane, size_t pix, float a, float b, float c, float d) {
  plane[pix * 4 + 0] = orc_f32_to_f16(a); plane[pix * 4 + 1] = orc_f32_to_f16(b);
  plane[pix * 4 + 2] = orc_f32_to_f16(c); plane[pix * 4 + 3] = orc_f32_to_f16(d);
}

void orc_pass_primary(const OrcScene* s, int mode, const OrcCamera* cam, const OrcSky* sky, OrcGBuffer* g, uint32_t y0,
                      uint32_t y1, OrcRayStats* st) {
  const uint32_t W = g->width, H = g->height;
  for (uint32_t py = y0; py < y1 && py < H; ++py)
    for (uint32_t px = 0; px < W; ++px) {
      size_t pix = (size_t)py * W + px;
      float o[3] = {cam->pos[0], cam->pos[1], cam->pos[2]}, d[3];
      orc_camera_ray_dir(cam, px, py, W, H, d);
      float t; uint32_t inst, block, voxel;
      /* primary.rgen:8-22: tmin = near, tmax = far, ray type 0 */
      if (!orc_trace(s, mode, 0, 0, o, d, cam->near_, cam->far_, &t, &inst, &block, &voxel, st)) {
        /* primary/miss.rmiss:7-17 */
        v3 dir = normalize3(V3(d[0], d[1], d[2]));
        v3 a = sky_radiance(sky, dir), b = sun_radiance(sky, dir);
        v3 c = V3((a.x + b.x) / 3.14f, (a.y + b.y) / 3.14f, (a.z + b.z) / 3.14f);
        pack_radiance(c, 100000.0f, g->denoised + pix * 4);
        g->albedo[pix] = 0xFFFFFFFFu;
        g->depth[pix] = INFINITY;
        put_half4(g->motion, pix, 0, 0, 0, 0);
        continue;
      }
      /* primary/hit.rchit:16-95 */
      const SceneInst* si = &s->insts[inst];
      const SceneModel* m = &s->models[si->in.model];
      const OrcBlock* b = &m->blocks[block];
      v3 wo = V3(o[0], o[1], o[2]), wd = V3(d[0], d[1], d[2]);
      v3 oo = xform_point(si->w2o, wo), od = xform_dir(si->w2o, wd);
      v3 hpo = V3(t * od.x + oo.x, t * od.y + oo.y, t * od.z + oo.z);
      v3 off = V3((float)(voxel >> 4), (float)((voxel >> 2) & 3u), (float)(voxel & 3u));
      v3 ctr = V3(((float)b->x + off.x) + 0.5f, ((float)b->y + off.y) + 0.5f, ((float)b->z + off.z) + 0.5f);
      v3 no = cubed_normalize(V3(hpo.x - ctr.x, hpo.y - ctr.y, hpo.z - ctr.z));
      v3 nw = xform_dir(si->in.obj_to_world, no);
      put_half4(g->illuminance, pix, 0, 0, 0, 0);
      uint32_t m1 = (uint32_t)b->mask, m2 = (uint32_t)(b->mask >> 32);
      uint32_t ma = voxel < 32u ? (m1 & ((1u << (voxel & 31u)) - 1u)) : m1;
      uint32_t mb = voxel >= 32u ? (m2 & ((1u << ((voxel - 32u) & 31u)) - 1u)) : 0u;
      uint32_t voff = (uint32_t)__builtin_popcount(ma) + (uint32_t)__builtin_popcount(mb);
      uint8_t pal = m->materials[b->material_ptr + voff];
      const uint8_t* col = m->palette + (size_t)pal * 4;
      float alb[4] = {(float)col[0] / 255.0f, (float)col[1] / 255.0f, (float)col[2] / 255.0f, 1.0f};
      g->albedo[pix] = orc_pack_rgb10a2(alb);
      g->depth[pix] = t;
      float nwa[3] = {nw.x, nw.y, nw.z}, pk[4];
      orc_nrd_pack_normal(nwa, 1.0f, (float)pal, pk);
      g->normal[pix] = orc_pack_rgb10a2(pk);
      g->voxel_id[pix] = (voxel << 24) | (inst & 0xFFFFu) | ((uint32_t)pal << 16);
      v3 hpw = V3(t * wd.x + wo.x, t * wd.y + wo.y, t * wd.z + wo.z);
      v3 hpm = xform_point(si->w2o, hpw);
      const float* P = si->in.prev_obj_to_world; /* column-major mat4 */
      float hx = ((P[0] * hpm.x + P[4] * hpm.y) + P[8] * hpm.z) + P[12];
      float hy = ((P[1] * hpm.x + P[5] * hpm.y) + P[9] * hpm.z) + P[13];
      float hz = ((P[2] * hpm.x + P[6] * hpm.y) + P[10] * hpm.z) + P[14];
      float hw = ((P[3] * hpm.x + P[7] * hpm.y) + P[11] * hpm.z) + P[15];
      put_half4(g->motion, pix, hx / hw - hpw.x, hy / hw - hpw.y, hz / hw - hpw.z, 0.0f);
    }
}

/* ------------------------------------------------------------------ ambient occlusion + sun shadow pass */
void orc_pass_ao(const OrcScene* s, int mode, const OrcCamera* cam, const OrcSky* sky, OrcGBuffer* g, const uint8_t* noise5,
                 uint32_t rnd, uint32_t y0, uint32_t y1, OrcRayStats* st_sun, OrcRayStats* st_ao) {
  const uint32_t W = g->width, H = g->height;
  for (uint32_t py = y0; py < y1 && py < H; ++py)
    for (uint32_t px = 0; px < W; ++px) {
      size_t pix = (size_t)py * W + px;
      /* ambient_occlusion.rgen:14-66 */
      float hitT = g->depth[pix];
      if (hitT == INFINITY) continue;
      float pk[4], nw[3];
      orc_unpack_rgb10a2(g->normal[pix], pk);
      orc_nrd_unpack_normal(pk, nw);
      float d[3];
      orc_camera_ray_dir(cam, px, py, W, H, d);
      float loc[3];
      for (int a = 0; a < 3; ++a) loc[a] = (hitT * d[a] + cam->pos[a]) + nw[a] * 0.01f;
      v3 inval; float inw;
      unpack_radiance(g->illuminance + pix * 4, &inval, &inw);
      uint32_t nx = (px + 7u + rnd) % 128u, ny = (py + 183u + rnd) % 128u;
      const uint8_t* tex = noise5 + ((size_t)ny * 128 + nx) * 4;
      v3 ns = V3((float)tex[0] / 255.0f * 2.0f - 1.0f, (float)tex[1] / 255.0f * 2.0f - 1.0f,
                 (float)tex[2] / 255.0f * 2.0f - 1.0f);
      v3 n = V3(nw[0], nw[1], nw[2]);
      ns = rotate_by_normal(n, ns);
      v3 payload = inval;
      v3 sun = V3(sky->v[48], sky->v[49], sky->v[50]);
      float t; uint32_t inst, block, voxel;
      if (dot3(sun, n) > 0.0f) {
        v3 sd = normalize3(sun);
        float sdir[3] = {sd.x, sd.y, sd.z};
        if (!orc_trace(s, mode, 1, 1, loc, sdir, 0.1f, 10000.0f, &t, &inst, &block, &voxel, st_sun)) {
          /* final_gather/nee.rmiss:11-22 */
          v3 sr = sun_radiance(sky, normalize3(sd));
          float k = 1.0f - cosf(sky->v[55]);
          float dn = dot3(n, sd);
          payload.x += (sr.x * k) * dn; payload.y += (sr.y * k) * dn; payload.z += (sr.z * k) * dn;
        }
      }
      v3 ad = normalize3(ns);
      float adir[3] = {ad.x, ad.y, ad.z};
      if (orc_trace(s, mode, 1, 0, loc, adir, 0.1f, 8.0f, &t, &inst, &block, &voxel, st_ao))
        pack_radiance(payload, t, g->illuminance + pix * 4); /* ambient_occlusion.rchit:10-13 */
      else
        pack_radiance(payload, 0.0f, g->illuminance + pix * 4); /* ambient_occlusion.rmiss:10-13 */
    }
}

/* ================================================================== hash-fed GI: final gather + surfel passes
 * The reference updates the spatial hash and the surfel pool with racy read-modify-writes, so its
 * result depends on GPU scheduling. The restatement fixes an order (documented deviation, DESIGN.md "GI order"):
 *   final gather: pixels in row-major order (the highest pixel index that enqueues wins a surfel slot);
 *   surfel pass : phase 1 -- every surfel traces its rays and reads the hash AS IT WAS AT THE START OF THE PASS
 *                            (Get still stamps last_accessed_frame, a write of the same value from everyone);
 *                 phase 2 -- the SpatialHashInsert calls and pool replacements are applied in surfel-index order; of the frame's
 *                            inserts of ONE key the last ORC_APPLY_KEEP (surfel_apply: the concurrent invocations' lost updates).
 */
typedef struct OrcHashEntry { uint32_t fingerprint, radiance; uint16_t last_accessed_frame, sample_count; } OrcHashEntry;
typedef struct OrcSurfel { float pos[3]; uint32_t direction; } OrcSurfel;
struct OrcGI {
  uint32_t capacity, pool_size;
  OrcHashEntry* hash; /* capacity + 2: probes run up to 2 past the end (spatial_hash.glsl:154-158) */
  OrcSurfel* pool;
};

OrcGI* orc_gi_new(uint32_t capacity, uint32_t pool_size) {
  OrcGI* g = (OrcGI*)calloc(1, sizeof(OrcGI));
  g->capacity = capacity; g->pool_size = pool_size;
  g->hash = (OrcHashEntry*)calloc((size_t)capacity + 2, sizeof(OrcHashEntry)); /* standard.rs:348-358 relies on zeros */
  g->pool = (OrcSurfel*)malloc((size_t)pool_size * sizeof(OrcSurfel));
  memset(g->pool, 0xFF, (size_t)pool_size * sizeof(OrcSurfel)); /* fill_buffer(u32::MAX), standard.rs:345-347 */
  return g;
}
void orc_gi_free(OrcGI* g) { if (g) { free(g->hash); free(g->pool); free(g); } }
void* orc_gi_hash_ptr(OrcGI* g) { return g->hash; }
void* orc_gi_pool_ptr(OrcGI* g) { return g->pool; }

typedef struct { int32_t x, y, z; uint32_t dir; } HashKey;
static uint32_t key_fingerprint(HashKey k) { /* spatial_hash.glsl:128-135 */
  uint32_t h = orc_xxhash32((uint32_t)k.x);
  h = orc_xxhash32((uint32_t)k.y + h);
  h = orc_xxhash32((uint32_t)k.z + h);
  h = orc_xxhash32(k.dir + h);
  return h > 1u ? h : 1u;
}
static uint32_t key_location(HashKey k, uint32_t capacity) { /* spatial_hash.glsl:136-142 */
  uint32_t h = orc_pcg((uint32_t)k.x);
  h = orc_pcg((uint32_t)k.y + h);
  h = orc_pcg((uint32_t)k.z + h);
  h = orc_pcg(k.dir + h);
  return h % capacity;
}
uint32_t orc_hash_fingerprint(const int32_t pos[3], uint32_t dir) { HashKey k = {pos[0], pos[1], pos[2], dir}; return key_fingerprint(k); }
uint32_t orc_hash_location(const int32_t pos[3], uint32_t dir, uint32_t capacity) { HashKey k = {pos[0], pos[1], pos[2], dir}; return key_location(k, capacity); }

static void hash_insert(OrcGI* g, HashKey key, v3 value, uint32_t frame_index) { /* spatial_hash.glsl:147-195 */
  uint32_t fp = key_fingerprint(key), loc = key_location(key, g->capacity);
  uint32_t i_min = 0, min_frame = 0;
  for (uint32_t i = 0; i < 3; ++i) {
    OrcHashEntry* e = &g->hash[loc + i];
    uint32_t cur = e->fingerprint; /* atomicCompSwap(fp, 0, new) */
    if (cur == 0) e->fingerprint = fp;
    uint32_t cur_frame = e->last_accessed_frame;
    if (i == 0 || cur_frame < min_frame) { i_min = i; min_frame = cur_frame; }
    if (cur == fp || cur == 0) {
      float rad[3] = {0, 0, 0};
      uint32_t count = 0;
      if (cur == fp) { count = e->sample_count; orc_logluv_decode(e->radiance, rad); }
      if (count > 403u) count = 403u; /* min(count, MAX_SAMPLE_COUNT - 1) */
      uint32_t next = count + 1;
      float a = 1.0f / (float)next; /* mix(x, y, a) = x*(1-a) + y*a */
      float out[3] = {rad[0] * (1.0f - a) + value.x * a, rad[1] * (1.0f - a) + value.y * a, rad[2] * (1.0f - a) + value.z * a};
      e->radiance = orc_logluv_encode(out);
      e->last_accessed_frame = (uint16_t)frame_index;
      e->sample_count = (uint16_t)next;
      return;
    }
  }
  OrcHashEntry* e = &g->hash[loc + i_min]; /* evict the least recently accessed of the three */
  float v[3] = {value.x, value.y, value.z};
  e->fingerprint = fp;
  e->radiance = orc_logluv_encode(v);
  e->last_accessed_frame = (uint16_t)frame_index;
  e->sample_count = 1;
}
static int hash_get(OrcGI* g, HashKey key, uint32_t frame_index, v3* value, uint32_t* count) { /* spatial_hash.glsl:200-219 */
  uint32_t fp = key_fingerprint(key), loc = key_location(key, g->capacity);
  *value = V3(0, 0, 0); *count = 0;
  for (uint32_t i = 0; i < 3; ++i) {
    OrcHashEntry* e = &g->hash[loc + i];
    if (e->fingerprint == 0) return 0;
    if (e->fingerprint == fp) {
      e->last_accessed_frame = (uint16_t)frame_index;
      float rad[3];
      orc_logluv_decode(e->radiance, rad);
      *value = V3(rad[0], rad[1], rad[2]);
      *count = e->sample_count;
      return 1;
    }
  }
  return 0;
}
void orc_hash_insert(OrcGI* g, const int32_t pos[3], uint32_t dir, const float value[3], uint32_t frame_index) {
  HashKey k = {pos[0], pos[1], pos[2], dir};
  hash_insert(g, k, V3(value[0], value[1], value[2]), frame_index);
}
int orc_hash_get(OrcGI* g, const int32_t pos[3], uint32_t dir, uint32_t frame_index, float value[3], uint32_t* count) {
  HashKey k = {pos[0], pos[1], pos[2], dir};
  v3 v;
  int f = hash_get(g, k, frame_index, &v, count);
  value[0] = v.x; value[1] = v.y; value[2] = v.z;
  return f;
}

static const float M_sRGB2ACEScg[9] = {0.6031065f, 0.07011794f, 0.022178888f, 0.32633433f, 0.9199162f,
                                       0.11607823f, 0.047995567f, 0.012763573f, 0.94101846f}; /* color.glsl:8-15 */
static const float M_ACEScg2sRGB[9] = {1.7312546f, -0.131619f, -0.024568284f, -0.6040432f, 1.1348418f,
                                       -0.12575036f, -0.08010775f, -0.008679431f, 1.0656371f}; /* color.glsl:16-23 */
static float srgb_to_linear(float c) { /* color.glsl:1-5 */
  return c < 0.04045f ? c / 12.92f : powf(fabsf(c + 0.055f) / 1.055f, 2.4f);
}
/* final_gather.rchit:68-80 / surfel.rchit:60-71 */
static v3 modulate_by_avg_albedo(v3 radiance, uint32_t packed) {
  v3 alb = V3(srgb_to_linear((float)((packed >> 22) & 1023u) / 1023.0f), srgb_to_linear((float)((packed >> 12) & 1023u) / 1023.0f),
              srgb_to_linear((float)((packed >> 2) & 1023u) / 1023.0f));
  v3 s = mat3_mul(M_ACEScg2sRGB, radiance);
  return mat3_mul(M_sRGB2ACEScg, V3(s.x * alb.x, s.y * alb.y, s.z * alb.z));
}

/* the world-space surfel (brick centre + face) of a rough hit: final_gather.rchit:35-45, surfel.rchit:35-45 */
static void brick_surfel(const OrcScene* s, uint32_t inst, uint32_t block, float t, v3 o, v3 d, HashKey* key, OrcSurfel* sf,
                         uint32_t* avg_albedo) {
  const SceneInst* si = &s->insts[inst];
  const SceneModel* m = &s->models[si->in.model];
  const OrcBlock* b = &m->blocks[block];
  v3 ctr = V3((float)b->x + 2.0f, (float)b->y + 2.0f, (float)b->z + 2.0f);
  v3 oo = xform_point(si->w2o, o), od = xform_dir(si->w2o, d);
  v3 hpo = V3(t * od.x + oo.x, t * od.y + oo.y, t * od.z + oo.z);
  v3 nw = cubed_normalize(xform_dir(si->in.obj_to_world, V3(hpo.x - ctr.x, hpo.y - ctr.y, hpo.z - ctr.z)));
  v3 cw = xform_point(si->in.obj_to_world, ctr);
  float nwa[3] = {nw.x, nw.y, nw.z};
  uint32_t face = orc_normal2faceid(nwa);
  key->x = f2i_sat(cw.x / 4.0f); key->y = f2i_sat(cw.y / 4.0f); key->z = f2i_sat(cw.z / 4.0f);
  key->dir = face;
  sf->pos[0] = cw.x; sf->pos[1] = cw.y; sf->pos[2] = cw.z; sf->direction = face;
  *avg_albedo = b->avg_albedo;
}

/* final_gather.rgen:14-52 + rough.rint + final_gather.rchit:35-91 + final_gather.rmiss:12-24 */
/* The pass over rows [y0, y1). log == NULL: surfel enqueues go straight into the pool (the serial pass, row-major: the last pixel that
 * aliases a slot wins). log != NULL: they are appended to it instead, in pixel order -- the threaded pass applies the threads' logs
 * one after the other, band by band, which is the same row-major order. */
typedef struct { uint32_t slot; OrcSurfel sf; } PoolWrite;
typedef struct { PoolWrite* w; size_t n, cap; } PoolLog;
static void final_gather_rows(const OrcScene* s, int mode, const OrcCamera* cam, const OrcSky* sky, OrcGBuffer* g, const uint8_t* noise0,
                              const uint8_t* noise5, uint32_t rnd, uint32_t frame_index, OrcGI* gi, uint32_t y0, uint32_t y1,
                              OrcRayStats* st, PoolLog* log) {
  const uint32_t W = g->width, H = g->height;
  for (uint32_t py = y0; py < y1 && py < H; ++py)
    for (uint32_t px = 0; px < W; ++px) {
      size_t pix = (size_t)py * W + px;
      float hitT = g->depth[pix];
      if (hitT == INFINITY) continue;
      v3 inval; float inw;
      unpack_radiance(g->illuminance + pix * 4, &inval, &inw);
      if (inw > 0.0f) continue; /* resolved by the AO pass */
      float pk[4], nw[3], d[3];
      orc_unpack_rgb10a2(g->normal[pix], pk);
      orc_nrd_unpack_normal(pk, nw);
      orc_camera_ray_dir(cam, px, py, W, H, d);
      float loc[3];
      for (int a = 0; a < 3; ++a) loc[a] = (hitT * d[a] + cam->pos[a]) + nw[a] * 0.01f;
      uint32_t nx = (px + 7u + rnd) % 128u, ny = (py + 183u + rnd) % 128u;
      const uint8_t* tex = noise5 + ((size_t)ny * 128 + nx) * 4;
      v3 ns = V3((float)tex[0] / 255.0f * 2.0f - 1.0f, (float)tex[1] / 255.0f * 2.0f - 1.0f, (float)tex[2] / 255.0f * 2.0f - 1.0f);
      v3 ad = normalize3(rotate_by_normal(V3(nw[0], nw[1], nw[2]), ns));
      float adir[3] = {ad.x, ad.y, ad.z};
      float t; uint32_t inst, block, voxel;
      if (!orc_trace(s, mode, 2, 0, loc, adir, 8.0f, cam->far_, &t, &inst, &block, &voxel, st)) {
        v3 sk = sky_radiance(sky, normalize3(ad));
        pack_radiance(V3(inval.x + sk.x, inval.y + sk.y, inval.z + sk.z), 0.0f, g->illuminance + pix * 4);
        continue;
      }
      HashKey key; OrcSurfel sf; uint32_t alb;
      brick_surfel(s, inst, block, t, V3(loc[0], loc[1], loc[2]), ad, &key, &sf, &alb);
      v3 rad; uint32_t count;
      hash_get(gi, key, frame_index, &rad, &count);
      float prob = 1.0f / (float)(count + 2u);
      float noise = (float)noise0[((size_t)((py + 21u + rnd) % 128u)) * 128 + ((px + 34u + rnd) % 128u)] / 255.0f;
      if (noise > prob) { /* final_gather.rchit:52-63 */
        const uint32_t slot = (px + py * W) % gi->pool_size;
        if (!log) gi->pool[slot] = sf;
        else {
          if (log->n == log->cap) { log->cap = log->cap ? log->cap * 2 : 4096; log->w = (PoolWrite*)realloc(log->w, log->cap * sizeof(PoolWrite)); }
          log->w[log->n].slot = slot; log->w[log->n].sf = sf; log->n += 1;
        }
      }
      rad = modulate_by_avg_albedo(rad, alb);
      pack_radiance(V3(inval.x + rad.x, inval.y + rad.y, inval.z + rad.z), t, g->illuminance + pix * 4);
    }
}

void orc_pass_final_gather(const OrcScene* s, int mode, const OrcCamera* cam, const OrcSky* sky, OrcGBuffer* g, const uint8_t* noise0,
                           const uint8_t* noise5, uint32_t rnd, uint32_t frame_index, OrcGI* gi, uint32_t y0, uint32_t y1,
                           OrcRayStats* st) {
  final_gather_rows(s, mode, cam, sky, g, noise0, noise5, rnd, frame_index, gi, y0, y1, st, NULL);
}

/* The same pass on n_threads host threads (row bands), with the serial pass's result: a pixel only writes its own texel; the hash is
 * only READ, apart from last_accessed_frame stamps, which all carry this frame's index (concurrent stores of one value); the surfel
 * enqueues -- the one order-dependent effect -- are logged per band and applied band after band. */
typedef struct {
  const OrcScene* s; int mode; const OrcCamera* cam; const OrcSky* sky; OrcGBuffer* g; const uint8_t *noise0, *noise5;
  uint32_t rnd, frame_index; OrcGI* gi; uint32_t y0, y1; OrcRayStats st; PoolLog log;
} FgJob;
static void* fg_thread(void* p) {
  FgJob* j = (FgJob*)p;
  final_gather_rows(j->s, j->mode, j->cam, j->sky, j->g, j->noise0, j->noise5, j->rnd, j->frame_index, j->gi, j->y0, j->y1, &j->st, &j->log);
  return NULL;
}
static void add_ray_stats(OrcRayStats* d, const OrcRayStats* a) {
  d->rays += a->rays; d->instances_tested += a->instances_tested; d->upper_descents += a->upper_descents;
  d->mid_descents += a->mid_descents; d->bricks_tested += a->bricks_tested; d->hits += a->hits;
}
void orc_pass_final_gather_mt(const OrcScene* s, int mode, const OrcCamera* cam, const OrcSky* sky, OrcGBuffer* g, const uint8_t* noise0,
                              const uint8_t* noise5, uint32_t rnd, uint32_t frame_index, OrcGI* gi, uint32_t y0, uint32_t y1,
                              uint32_t n_threads, OrcRayStats* st) {
  if (y1 > g->height) y1 = g->height;
  if (n_threads < 1) n_threads = 1;
  if (y1 > y0 && n_threads > y1 - y0) n_threads = y1 - y0;
  FgJob* jobs = (FgJob*)calloc(n_threads, sizeof(FgJob));
  pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
  for (uint32_t t = 0; t < n_threads; ++t) {
    FgJob* j = &jobs[t];
    j->s = s; j->mode = mode; j->cam = cam; j->sky = sky; j->g = g; j->noise0 = noise0; j->noise5 = noise5; j->rnd = rnd;
    j->frame_index = frame_index; j->gi = gi;
    j->y0 = y0 + (uint32_t)(((uint64_t)(y1 - y0) * t) / n_threads);
    j->y1 = y0 + (uint32_t)(((uint64_t)(y1 - y0) * (t + 1)) / n_threads);
    pthread_create(&th[t], NULL, fg_thread, j);
  }
  for (uint32_t t = 0; t < n_threads; ++t) {
    pthread_join(th[t], NULL);
    for (size_t k = 0; k < jobs[t].log.n; ++k) gi->pool[jobs[t].log.w[k].slot] = jobs[t].log.w[k].sf; /* bands in order = row-major order */
    free(jobs[t].log.w);
    if (st) add_ray_stats(st, &jobs[t].st);
  }
  free(th); free(jobs);
}

/* surfel.rgen:12-67 + rough.rint + surfel.rchit:35-102 + surfel.rmiss:14-26 + surfel/nee.rmiss:15-27 */
typedef struct { int kind; HashKey key; v3 value; int replace; OrcSurfel repl; } SurfelReq;
/* phase 1 for surfels [i0, i1): traces, READS the hash as it stands (plus frame stamps) and fills in req[i] */
static void surfel_trace_range(const OrcScene* s, int mode, const OrcSky* sky, const uint8_t* noise0, const uint8_t* noise5, uint32_t rnd,
                               uint32_t frame_index, OrcGI* gi, SurfelReq* req, uint32_t i0, uint32_t i1, OrcRayStats* st_sun, OrcRayStats* st_cos) {
  for (uint32_t i = i0; i < i1; ++i) { /* phase 1 */
    OrcSurfel e = gi->pool[i];
    if (e.direction >= 6u) continue;
    v3 n = faceid2normal(e.direction);
    uint32_t ny0 = i / 128u, nx0 = i - ny0 * 128u;
    float org[3] = {e.pos[0] + 2.01f * n.x, e.pos[1] + 2.01f * n.y, e.pos[2] + 2.01f * n.z};
    const uint8_t* tex = noise5 + ((size_t)((ny0 + 47u + rnd) % 128u) * 128 + ((nx0 + 16u + rnd) % 128u)) * 4;
    v3 ns = V3((float)tex[0] / 255.0f * 2.0f - 1.0f, (float)tex[1] / 255.0f * 2.0f - 1.0f, (float)tex[2] / 255.0f * 2.0f - 1.0f);
    ns = rotate_by_normal(n, ns);
    v3 sun = V3(sky->v[48], sky->v[49], sky->v[50]);
    v3 payload = V3(0, 0, 0);
    float t; uint32_t inst, block, voxel;
    if (dot3(sun, n) > 0.0f) {
      v3 sd = normalize3(sun);
      float sdir[3] = {sd.x, sd.y, sd.z};
      if (!orc_trace(s, mode, 3, 1, org, sdir, 0.1f, 10000.0f, &t, &inst, &block, &voxel, st_sun)) {
        v3 sr = sun_radiance(sky, normalize3(sd));
        float k = 1.0f - cosf(sky->v[55]);
        float dn = dot3(n, sd);
        payload = V3((sr.x * k) * dn, (sr.y * k) * dn, (sr.z * k) * dn);
      }
    }
    HashKey skey = {f2i_sat(e.pos[0] / 4.0f), f2i_sat(e.pos[1] / 4.0f), f2i_sat(e.pos[2] / 4.0f), e.direction & 0xFFu};
    v3 cd = normalize3(ns);
    float cdir[3] = {cd.x, cd.y, cd.z};
    if (!orc_trace(s, mode, 3, 0, org, cdir, 0.1f, 10000.0f, &t, &inst, &block, &voxel, st_cos)) {
      v3 sk = sky_radiance(sky, normalize3(cd));
      req[i].kind = 1; req[i].key = skey; req[i].value = V3(sk.x + payload.x, sk.y + payload.y, sk.z + payload.z);
      continue;
    }
    HashKey key; OrcSurfel sf; uint32_t alb;
    brick_surfel(s, inst, block, t, V3(org[0], org[1], org[2]), cd, &key, &sf, &alb);
    v3 rad; uint32_t count = 0;
    int found = hash_get(gi, key, frame_index, &rad, &count);
    float rnd0 = (float)noise0[((size_t)((ny0 + 40u + rnd) % 128u)) * 128 + ((nx0 + 114u + rnd) % 128u)] / 255.0f;
    if (found) {
      rad = modulate_by_avg_albedo(rad, alb);
      req[i].kind = 1; req[i].key = skey; req[i].value = V3(rad.x + payload.x, rad.y + payload.y, rad.z + payload.z);
    } else {
      float prob = 1.0f / (float)(count + 2u);
      if (rnd0 > prob) { req[i].replace = 1; req[i].repl = sf; }
    }
  }
}
/* Which of a frame's SpatialHashInsert calls land. The reference's invocations run concurrently and only the fingerprint is claimed
 * atomically (spatial_hash.glsl:147-195): of the invocations that insert ONE key together, those that read the entry before the
 * others' stores see the same old state, and the last store wins. The defined order (DESIGN.md "GI order"; the HIP path's
 * k_surfel_apply_mark states the same rule): the frame's insert requests are sorted by (hash location, surfel index); a request is
 * SUPERSEDED -- not applied -- when the request ORC_APPLY_KEEP places further on in that order has the same location and the same key
 * (of a run of requests with one key, the last ORC_APPLY_KEEP stay); the others are applied in surfel-index order. */
#define ORC_APPLY_KEEP 8u
typedef struct { uint32_t loc, idx; } ApplyOrder;
static int apply_order_cmp(const void* a, const void* b) {
  const ApplyOrder *x = (const ApplyOrder*)a, *y = (const ApplyOrder*)b;
  if (x->loc != y->loc) return x->loc < y->loc ? -1 : 1;
  return x->idx < y->idx ? -1 : (x->idx > y->idx ? 1 : 0);
}
static void surfel_apply(OrcGI* gi, const SurfelReq* req, uint32_t frame_index) {
  const uint32_t N = gi->pool_size;
  ApplyOrder* ord = (ApplyOrder*)malloc((size_t)N * sizeof(ApplyOrder));
  uint8_t* dead = (uint8_t*)calloc(N, 1);
  uint32_t n = 0;
  for (uint32_t i = 0; i < N; ++i)
    if (req[i].kind == 1) { ord[n].loc = key_location(req[i].key, gi->capacity); ord[n].idx = i; ++n; }
  qsort(ord, n, sizeof(ApplyOrder), apply_order_cmp);
  for (uint32_t p = 0; p + ORC_APPLY_KEEP < n; ++p) {
    const ApplyOrder *x = &ord[p], *y = &ord[p + ORC_APPLY_KEEP];
    const HashKey *kx = &req[x->idx].key, *ky = &req[y->idx].key;
    if (x->loc == y->loc && kx->x == ky->x && kx->y == ky->y && kx->z == ky->z && kx->dir == ky->dir) dead[x->idx] = 1;
  }
  for (uint32_t i = 0; i < N; ++i) { /* phase 2: in surfel order */
    if (req[i].kind == 1 && !dead[i]) hash_insert(gi, req[i].key, req[i].value, frame_index);
    if (req[i].replace) gi->pool[i % N] = req[i].repl;
  }
  free(ord); free(dead);
}
void orc_pass_surfel(const OrcScene* s, int mode, const OrcSky* sky, const uint8_t* noise0, const uint8_t* noise5, uint32_t rnd,
                     uint32_t frame_index, OrcGI* gi, OrcRayStats* st_sun, OrcRayStats* st_cos) {
  SurfelReq* req = (SurfelReq*)calloc(gi->pool_size, sizeof(SurfelReq));
  surfel_trace_range(s, mode, sky, noise0, noise5, rnd, frame_index, gi, req, 0, gi->pool_size, st_sun, st_cos);
  surfel_apply(gi, req, frame_index);
  free(req);
}
/* phase 1 on n_threads host threads (it writes nothing but its own request and frame stamps), phase 2 as above */
typedef struct {
  const OrcScene* s; int mode; const OrcSky* sky; const uint8_t *noise0, *noise5; uint32_t rnd, frame_index; OrcGI* gi; SurfelReq* req;
  uint32_t i0, i1; OrcRayStats st_sun, st_cos;
} SfJob;
static void* sf_thread(void* p) {
  SfJob* j = (SfJob*)p;
  surfel_trace_range(j->s, j->mode, j->sky, j->noise0, j->noise5, j->rnd, j->frame_index, j->gi, j->req, j->i0, j->i1, &j->st_sun, &j->st_cos);
  return NULL;
}
void orc_pass_surfel_mt(const OrcScene* s, int mode, const OrcSky* sky, const uint8_t* noise0, const uint8_t* noise5, uint32_t rnd,
                        uint32_t frame_index, OrcGI* gi, uint32_t n_threads, OrcRayStats* st_sun, OrcRayStats* st_cos) {
  const uint32_t N = gi->pool_size;
  if (n_threads < 1) n_threads = 1;
  if (n_threads > N) n_threads = N;
  SurfelReq* req = (SurfelReq*)calloc(N, sizeof(SurfelReq));
  SfJob* jobs = (SfJob*)calloc(n_threads, sizeof(SfJob));
  pthread_t* th = (pthread_t*)calloc(n_threads, sizeof(pthread_t));
  for (uint32_t t = 0; t < n_threads; ++t) {
    SfJob* j = &jobs[t];
    j->s = s; j->mode = mode; j->sky = sky; j->noise0 = noise0; j->noise5 = noise5; j->rnd = rnd; j->frame_index = frame_index; j->gi = gi; j->req = req;
    j->i0 = (uint32_t)(((uint64_t)N * t) / n_threads); j->i1 = (uint32_t)(((uint64_t)N * (t + 1)) / n_threads);
    pthread_create(&th[t], NULL, sf_thread, j);
  }
  for (uint32_t t = 0; t < n_threads; ++t) {
    pthread_join(th[t], NULL);
    if (st_sun) add_ray_stats(st_sun, &jobs[t].st_sun);
    if (st_cos) add_ray_stats(st_cos, &jobs[t].st_cos);
  }
  surfel_apply(gi, req, frame_index);
  free(th); free(jobs); free(req);
}


/* ================================================================== auto exposure + tone map
 * auto_exposure.comp:20-74, auto_exposure_avg.comp:19-53, tone_map.comp:39-220 */
static uint32_t color_to_bin(v3 c, float min_log, float log_range) { /* auto_exposure.comp:20-35 */
  float lum = (c.x * 0.299f + c.y * 0.587f) + c.z * 0.114f;
  if (lum < 0.005f) return 0;
  float logLum = gclamp((log2f(lum) - min_log) * (1.0f / log_range), 0.0f, 1.0f);
  return (uint32_t)(logLum * 254.0f + 1.0f);
}
void orc_exposure_histogram(const uint16_t* illuminance, uint32_t w, uint32_t h, float min_log, float log_range, uint32_t hist[256]) {
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    v3 c; float ww;
    unpack_radiance(illuminance + i * 4, &c, &ww);
    hist[color_to_bin(c, min_log, log_range)] += 1;
  }
}
float orc_exposure_average(uint32_t hist[256], uint32_t w, uint32_t h, float min_log, float log_range, float time_coeff, float avg) {
  uint32_t sum = 0; /* uint arithmetic, as histogramShared[] (auto_exposure_avg.comp:24-34) */
  for (uint32_t i = 0; i < 256; ++i) { sum += hist[i] * i; hist[i] = 0; }
  float num = fmaxf((float)(w * h), 1.0f);
  float weighted_log_avg = ((float)sum / num) - 1.0f;
  float weighted_avg_lum = exp2f(((weighted_log_avg / 254.0f) * log_range) + min_log);
  return avg + (weighted_avg_lum - avg) * time_coeff;
}
static v3 rrt_odt_fit(v3 v) { /* tone_map.comp:39-43 */
  v3 a = V3(v.x * (v.x + 0.0245786f) - 0.000090537f, v.y * (v.y + 0.0245786f) - 0.000090537f, v.z * (v.z + 0.0245786f) - 0.000090537f);
  v3 b = V3(v.x * (0.983729f * v.x + 0.4329510f) + 0.238081f, v.y * (0.983729f * v.y + 0.4329510f) + 0.238081f,
            v.z * (0.983729f * v.z + 0.4329510f) + 0.238081f);
  return V3(a.x / b.x, a.y / b.y, a.z / b.z);
}
static v3 aces_fitted(v3 c) { /* tone_map.comp:45-71; `v *= M` is v * M: dot with the columns */
  v3 r = V3((c.x * 0.59719f + c.y * 0.35458f) + c.z * 0.04823f, (c.x * 0.07600f + c.y * 0.90834f) + c.z * 0.01566f,
            (c.x * 0.02840f + c.y * 0.13383f) + c.z * 0.83777f);
  r = rrt_odt_fit(r);
  return V3((r.x * 1.60475f + r.y * -0.53108f) + r.z * -0.07367f, (r.x * -0.10208f + r.y * 1.10813f) + r.z * -0.00605f,
            (r.x * -0.00327f + r.y * -0.07276f) + r.z * 1.07602f);
}
static float oetf(uint32_t tf, float c) { /* tone_map.comp:72-181 */
  switch (tf) {
    case 1: return c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    case 2: return c <= -0.0031308f ? -1.055f * powf(-c, 1.0f / 2.4f) + 0.055f : (c <= 0.0031308f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f);
    case 3: return powf(c / 52.37f, 1.0f / 2.6f);
    case 4: return c < 0.0030186f ? 12.92f * c : 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
    case 5: return c < 0.0181f ? 4.5f * c : 1.0993f * powf(c, 0.45f) - (1.0993f - 1.0f);
    case 6: {
      const float m1 = 2610.0f / 16384.0f, m2 = (2523.0f / 4096.0f) * 128.0f, c2 = (2413.0f / 4096.0f) * 32.0f, c3 = (2392.0f / 4096.0f) * 32.0f;
      const float c1 = c3 - c2 + 1.0f;
      float Lm = powf(c, m1);
      return powf((c1 + c2 * Lm) / (1.0f + c3 * Lm), m2);
    }
    case 7: return c < (1.0f / 12.0f) ? sqrtf(3.0f * c) : 0.17883277f * logf(12.0f * c - (1.0f - 4.0f * 0.17883277f)) + 0.55991073f;
    case 8: return powf(c, 256.0f / 563.0f);
    default: return c;
  }
}
/* tone_map.comp:184-220. conv: the nine COLOR_SPACE_CONVERSION_* constants (column-major mat3) */
void orc_tone_map(const uint16_t* src, const uint32_t* albedo, uint32_t w, uint32_t h, float avg, const float conv[9], uint32_t tf,
                  uint16_t* dst) {
  for (size_t i = 0; i < (size_t)w * h; ++i) {
    v3 c; float ww, al[4];
    unpack_radiance(src + i * 4, &c, &ww);
    orc_unpack_rgb10a2(albedo[i], al);
    v3 alb = V3(srgb_to_linear(al[0]), srgb_to_linear(al[1]), srgb_to_linear(al[2]));
    float exposure = 1.0f / (9.6f * avg);
    exposure *= 9.6f;
    v3 s = mat3_mul(M_ACEScg2sRGB, c);
    v3 m = mat3_mul(M_sRGB2ACEScg, V3(s.x * alb.x, s.y * alb.y, s.z * alb.z));
    m = V3(m.x * exposure, m.y * exposure, m.z * exposure);
    m = mat3_mul(conv, m);
    m = aces_fitted(m);
    dst[i * 4 + 0] = orc_f32_to_f16(oetf(tf, m.x)); dst[i * 4 + 1] = orc_f32_to_f16(oetf(tf, m.y));
    dst[i * 4 + 2] = orc_f32_to_f16(oetf(tf, m.z)); dst[i * 4 + 3] = orc_f32_to_f16(1.0f);
  }
}
