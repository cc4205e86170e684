// traverse.hpp -- the device code every traversal kernel is made of (included by kernels.hip and gi.hip, one copy per
// translation unit): storage formats, the intersection shaders' arithmetic, the hierarchical walk, the packet cull, the
// persistent work distribution, the spatial hash. Kernels live in kernels.hip (primary / AO / tone map / tile order) and
// gi.hip (final gather, surfel pass).
//
// What the reference does with four vkCmdTraceRaysKHR calls over a driver BVH of per-leaf AABBs
// (crates/render/src/pipeline/standard.rs:477-725, assets/shaders/{primary,final_gather,surfel})
// is done here by walking the VDB hierarchy directly:
//   * persistent workgroups, one 64-lane wavefront per 8x8 pixel packet, packets pulled from
//     per-XCD-region atomic counters (block b runs on XCD b%8, so a region stays in one L2) through a
//     small LDS queue per workgroup; the launch descriptor is read in place from the kernarg segment;
//   * the root node (4096-bit child mask + rank prefix) of every model is staged in LDS once per
//     workgroup; mid nodes and brick masks come from HBM/L2 (16 B and 8 B loads);
//   * the packet's rays are bounded once (DPP wave reductions) and tested against all instance boxes
//     64 at a time (__ballot compaction into a per-wave LDS candidate list, rank-sorted front to back);
//   * incoherent rays are regrouped before they are traced: gather rays by direction octant inside
//     32x32 pixel tiles (k_gather_order), surfels by position (k_surfel_keys + a radix sort); and their
//     lanes visit instances independently (a uniform box scan leaves each lane its own candidate mask);
//   * per ray, a hierarchical DDA over 16^3 / 4^3 cells finds candidate bricks front to back; the
//     brick test itself is the reference's intersection shader arithmetic, bit for bit
//     (primary/hit.rint:43-131, final_gather/ambient_occlusion.rint:46-134, rough.rint:42-59).
// Built with -ffp-contract=off: the brick test and the G-buffer maths must round exactly like the
// oracle's. No MFMA: this is pointer chasing, not a contraction.
#pragma once
#include <hip/hip_runtime.h>

#define DUST_DEVICE_ADDRESS_SPACES 1
#include "dust_dev.h"
#include "exact_div.hpp"

#ifdef DUST_PLAIN_STORES
#define DUST_NT_STORE(v, p) (*(p) = (v))
#else
#define DUST_NT_STORE(v, p) __builtin_nontemporal_store(v, p)
#endif

namespace dust {

extern __shared__ __attribute__((aligned(16))) unsigned char g_lds[];

// Section timers for tools/kernel_sections.py: compiled in only with -DDUST_PROFILE (a separate, never shipped
// .so). PROF_ENTER/PROF_LEAVE add -/+ s_memtime to a per-wave LDS bucket (fire-and-forget ds_add by the first
// active lane), so a bucket ends up holding the wave's inclusive cycles in that section.
// -DDUST_WAVE_TIMES (never shipped; tools/wave_times.py): every wave of the fused kernel writes when it started, when the staging
// barrier let it go and when it ran out of tiles -- on the shader clock (s_memtime) and on the constant 100 MHz wall clock
#ifdef DUST_WAVE_TIMES
static __device__ unsigned long long g_wave_times[8192][12];
static __device__ unsigned long long g_launch_clock[256][3];  // per launch (frame_index & 255): wave 0's shader ticks, wall ticks (10 ns), frame index
#endif
#ifdef DUST_PROFILE
constexpr int kProfBuckets = 24;
__shared__ unsigned long long g_prof[16][kProfBuckets];
static __device__ unsigned long long g_prof_out[kProfBuckets];  // (one per translation unit: dust_hip_profile_read adds them up)
__device__ __forceinline__ void prof_mark(int idx, bool leave) {
  const unsigned long long now = __builtin_amdgcn_s_memtime();
  const unsigned long long ex = __ballot(1);
  if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)ex) - 1))
    atomicAdd(&g_prof[threadIdx.x >> 6][idx], leave ? now : (0ull - now));
}
__device__ __forceinline__ void prof_count(int idx, unsigned long long n) {
  const unsigned long long ex = __ballot(1);
  if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)ex) - 1)) atomicAdd(&g_prof[threadIdx.x >> 6][idx], n);
}
#define PROF_ENTER(i) prof_mark(i, false)
#define PROF_LEAVE(i) prof_mark(i, true)
#define PROF_COUNT(i, n) prof_count(i, n)
#define PROF_COUNT_LANES(i, pred) prof_count(i, (unsigned long long)__popcll(__ballot(pred)))
#else
#define PROF_COUNT_LANES(i, pred)
#define PROF_ENTER(i)
#define PROF_LEAVE(i)
#define PROF_COUNT(i, n)
#endif
// -DDUST_TRACE_DEBUG (never shipped): printf what the final gather does for ONE pixel, DUST_HIP_DEBUG = (pixel index + 1) << 12
#ifdef DUST_TRACE_DEBUG
__shared__ unsigned long long g_dbg_mask[16];  // per wave: lanes being traced verbosely
#define DBG_LANE() ((g_dbg_mask[threadIdx.x >> 6] >> (threadIdx.x & 63u)) & 1ull)
#define DBG_PRINT(...) do { if (DBG_LANE()) printf(__VA_ARGS__); } while (0)
#else
#define DBG_PRINT(...)
#endif
// Kernel variants are selected by one template integer: bit 0 = the counting build (DUST_PASS_COUNT_STATS), bit 1 = the
// scene holds a 4096^3 model (hierarchy (4,4,2,2)): its 16-cell lookups then go through the per-cell table (DevModel::l2_cells)
// and the walk carries the 16-cell's child mask. Scenes without such a model run MODE 0 / 1: exactly the two-level code.
#define COUNT ((MODE & 1) != 0)
#define DEEP ((MODE & 2) != 0)
#define LARGE ((MODE & 4) != 0)  // bit 2: a scene beyond kFlatCullMax instances -- the packet cull's 64-wide hierarchy is compiled into these variants only
                                 // (in the others it cost the fused kernel 0.8 %: registers and code it never runs)
enum { P_TOTAL = 0, P_GRAB, P_CULL, P_TRACE_RAY, P_INSTANCE, P_FIND, P_BRICK, P_SCREEN, P_ADVANCE, P_STAGE, P_SHADE,
       P_N_TRACES, P_N_CAND, P_N_CAND_ITER, P_N_VISITS, P_N_STEPS };
enum { P_N_NEIGHBOUR_CALLS = 10 };  // (reuses the unused P_SHADE bucket)
enum { P_L_TRIPS = 16, P_L_BRICK, P_L_EMPTY4, P_L_EMPTY16, P_SETUP = 20, P_PRIMARY_SHADE = 21, P_AO_SETUP = 22, P_CAND = 23 };  // lane-level trip outcomes  // the P_N_* buckets count events, not cycles

// Earned priorities (round 4): per wave, when its tile began (shader clock) and the issue priority it runs at. A tile's position in the
// cost order gives it a priority to start with (packet_of_tile); a tile that turns out longer than that promised -- a view that moves
// hands out a few frames old order, a first frame has none -- earns the priority as it goes: the walk looks at the clock every 16th
// trip of a visit. 60 k / 120 k / 200 k cycles (the median tile is 49 k, the heaviest 467 k): -0.8 % on the still castle, -2 % on the
// moving one, -2 % on the deep tree; 100/180/280 k, 40/80/140 k and 30/60/100 k measured the same.
#ifndef DUST_DYN_T1
#define DUST_DYN_T1 60000u
#define DUST_DYN_T2 120000u
#define DUST_DYN_T3 200000u
#endif
__shared__ uint32_t g_tile_start[16], g_tile_prio[16];
namespace {

// launch descriptor, models and instances live in the constant address space (see dust_dev.h)
typedef const DUST_CONST_AS FrameArgs& ArgsRef;
typedef const DUST_CONST_AS DevModel& ModelRef;
typedef const DUST_CONST_AS DevInstance& InstanceRef;
// The launch descriptor travels by value in the kernel-argument segment (about 800 of the 4096 bytes it may hold) and is
// read in place through the segment's own constant-address-space pointer: uniform fields are s_loads, nothing is copied,
// and the host side needs no staging buffer, copy or event per launch. Every kernel's only parameter is the descriptor.
__device__ __forceinline__ const DUST_CONST_AS FrameArgs& launch_args() {
  return *(const DUST_CONST_AS FrameArgs*)__builtin_amdgcn_kernarg_segment_ptr();
}
// The same descriptor through a pointer the optimiser has lost track of: fields read through the result are loaded
// again (one s_load each, at the point of use) instead of being carried in SGPRs -- or spilled to VGPR lanes -- across
// whatever came before.
__device__ __forceinline__ const DUST_CONST_AS FrameArgs& reload_args(const DUST_CONST_AS FrameArgs& a) {
  const DUST_CONST_AS FrameArgs* q = &a;
  asm volatile("" : "+s"(q));
  return *q;
}
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float gsign(float x) { return (float)((x > 0.0f) - (x < 0.0f)); }
__device__ __forceinline__ float gstep(float edge, float x) { return x < edge ? 0.0f : 1.0f; }
__device__ __forceinline__ float gclamp(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
__device__ __forceinline__ float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ V3 div3(V3 v, float b) {  // v / b
  const float y = 1.0f / b;
  const bool sp = recip_special(y);
  return mk(div_by(v.x, b, y, sp), div_by(v.y, b, y, sp), div_by(v.z, b, y, sp));
}
__device__ __forceinline__ V3 normalize3(V3 v) { return div3(v, sqrtf(dot3(v, v))); }
__device__ __forceinline__ int f2i_clamp(float f, int lo, int hi) {  // clamp(int(floor-ed f)) with NaN -> lo side of 0
  float c = fminf(fmaxf(f, (float)lo), (float)hi);                   // fmaxf(NaN, lo) == lo
  return (int)c;
}
__device__ __forceinline__ int f2i_trunc(float f) {  // ivec3(float): toward zero; NaN -> 0, saturating (v_cvt_i32_f32)
  if (f != f) return 0;
  if (f >= 2147483648.0f) return 2147483647;
  if (f <= -2147483648.0f) return (-2147483647 - 1);
  return (int)f;
}
__device__ __forceinline__ V3 xform_point(DUST_RO(float) m, V3 p) {
  return mk(((m[0] * p.x + m[1] * p.y) + m[2] * p.z) + m[3], ((m[4] * p.x + m[5] * p.y) + m[6] * p.z) + m[7],
            ((m[8] * p.x + m[9] * p.y) + m[10] * p.z) + m[11]);
}
__device__ __forceinline__ V3 xform_dir(DUST_RO(float) m, V3 d) {
  return mk((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z,
            (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
}

// ------------------------------------------------------------------ storage formats (standard.rs:974-1050)
__device__ __forceinline__ uint16_t f2h(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }  // RNE
__device__ __forceinline__ float h2f(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint32_t unorm(float v, float scale) {
  if (!(v > 0.0f)) return 0;
  if (v >= 1.0f) return (uint32_t)scale;
  return (uint32_t)rintf(v * scale);
}
__device__ __forceinline__ uint32_t pack_rgb10a2(float r, float g, float b, float a) {
  return unorm(r, 1023.0f) | (unorm(g, 1023.0f) << 10) | (unorm(b, 1023.0f) << 20) | (unorm(a, 3.0f) << 30);
}
__device__ __forceinline__ u32x2 pack_half4(float a, float b, float c, float d) {
  u32x2 v;
  v.x = (uint32_t)f2h(a) | ((uint32_t)f2h(b) << 16);
  v.y = (uint32_t)f2h(c) | ((uint32_t)f2h(d) << 16);
  return v;
}
__device__ __forceinline__ void store_half4(DUST_RW(uint16_t) plane, size_t pix, float a, float b, float c, float d) {
  DUST_NT_STORE(pack_half4(a, b, c, d), (DUST_GLOBAL_AS u32x2*)(plane + pix * 4));  // written once, read by a later pass: do not displace the scene in L2
}

// ------------------------------------------------------------------ headers/normal.glsl, nrd.glsl
__device__ __forceinline__ V3 cubed_normalize(V3 d) {  // normal.glsl:39-43
  V3 a = mk(fabsf(d.x), fabsf(d.y), fabsf(d.z));
  float mx = fmaxf(a.x, fmaxf(a.y, a.z));
  return mk(gsign(d.x) * gstep(mx, a.x), gsign(d.y) * gstep(mx, a.y), gsign(d.z) * gstep(mx, a.z));
}
__device__ __forceinline__ V3 rotate_by_normal(V3 n, V3 t) {  // normal.glsl:31-37
  float qx = -n.y, qy = n.x, qz = 0.0f, qw = 1.0f + n.z;
  float l = sqrtf(((qx * qx + qy * qy) + qz * qz) + qw * qw);
  {
    const float y = 1.0f / l;
    const bool sp = recip_special(y);
    qx = div_by(qx, l, y, sp); qy = div_by(qy, l, y, sp); qz = div_by(qz, l, y, sp); qw = div_by(qw, l, y, sp);
  }
  if (n.z < -0.99999f) { qx = -1.0f; qy = 0.0f; qz = 0.0f; qw = 0.0f; }
  V3 q = mk(qx, qy, qz);
  float two_dot = 2.0f * dot3(q, t);
  float k = qw * qw - dot3(q, q);
  V3 c = mk(q.y * t.z - t.y * q.z, q.z * t.x - t.z * q.x, q.x * t.y - t.x * q.y);
  float tw = 2.0f * qw;
  return mk((two_dot * q.x + k * t.x) + tw * c.x, (two_dot * q.y + k * t.y) + tw * c.y,
            (two_dot * q.z + k * t.z) + tw * c.z);
}
__device__ __forceinline__ uint32_t nrd_pack_normal(V3 v, float roughness, float material_id) {  // nrd.glsl:2-10,25-52
  float s = (fabsf(v.x) + fabsf(v.y)) + fabsf(v.z);
  v = div3(v, s);
  float wx = (1.0f - fabsf(v.y)) * (gstep(0.0f, v.x) * 2.0f - 1.0f);
  float wy = (1.0f - fabsf(v.x)) * (gstep(0.0f, v.y) * 2.0f - 1.0f);
  float ex = v.z >= 0.0f ? v.x : wx, ey = v.z >= 0.0f ? v.y : wy;
  return pack_rgb10a2(ex * 0.5f + 0.5f, ey * 0.5f + 0.5f, roughness, gclamp(div_const(material_id, 3.0f), 0.0f, 1.0f));
}
__device__ __forceinline__ V3 nrd_unpack_normal(uint32_t p) {  // nrd.glsl:54-94 on an A2B10G10R10 texel
  float p0 = div_const((float)(p & 1023u), 1023.0f), p1 = div_const((float)((p >> 10) & 1023u), 1023.0f);
  float px = p0 * 2.0f - 1.0f, py = p1 * 2.0f - 1.0f;
  V3 n = mk(px, py, (1.0f - fabsf(px)) - fabsf(py));
  float t = gclamp(-n.z, 0.0f, 1.0f);
  n.x -= t * (gstep(0.0f, n.x) * 2.0f - 1.0f);
  n.y -= t * (gstep(0.0f, n.y) * 2.0f - 1.0f);
  return normalize3(n);
}
__device__ __forceinline__ u32x2 pack_radiance(V3 r, float hitdist) {  // nrd.glsl:127-147, as four fp16 values
  if (hitdist != 0.0f) hitdist = fmaxf(hitdist, 1e-7f);
  float Y = (r.x * 0.25f + r.y * 0.5f) + r.z * 0.25f;
  float Co = (r.x * 0.5f + r.y * 0.0f) + r.z * -0.5f;
  float Cg = (r.x * -0.25f + r.y * 0.5f) + r.z * -0.25f;
  return pack_half4(Y, Co, Cg, hitdist);
}
__device__ __forceinline__ void store_radiance(DUST_RW(uint16_t) plane, size_t pix, V3 r, float hitdist) {
  DUST_NT_STORE(pack_radiance(r, hitdist), (DUST_GLOBAL_AS u32x2*)(plane + pix * 4));
}
// the same texel from a lane whose neighbours hold unrelated pixels (regrouped gather rays): an ordinary store, so that the
// 8-byte pieces of a line meet in L2 -- a tile's packets run at about the same time on one XCD -- instead of going out one by one
__device__ __forceinline__ void store_radiance_scattered(DUST_RW(uint16_t) plane, size_t pix, V3 r, float hitdist) {
#ifdef DUST_NT_GATHER
  DUST_NT_STORE(pack_radiance(r, hitdist), (DUST_GLOBAL_AS u32x2*)(plane + pix * 4));
#else
  *(DUST_GLOBAL_AS u32x2*)(plane + pix * 4) = pack_radiance(r, hitdist);
#endif
}
__device__ __forceinline__ V3 decode_radiance(u32x2 v, float& w);
__device__ __forceinline__ V3 load_radiance(DUST_RW(uint16_t) plane, size_t pix, float& w) {  // a G-buffer plane
  return decode_radiance(*(const DUST_GLOBAL_AS u32x2*)(plane + pix * 4), w);
}
__device__ __forceinline__ V3 load_radiance(const uint16_t* plane, size_t pix, float& w) {     // a kernel argument
  return decode_radiance(*reinterpret_cast<const u32x2*>(plane + pix * 4), w);
}
__device__ __forceinline__ V3 decode_radiance(u32x2 v, float& w) {  // nrd.glsl:107-125
  float Y = h2f((uint16_t)v.x), Co = h2f((uint16_t)(v.x >> 16)), Cg = h2f((uint16_t)v.y);
  w = h2f((uint16_t)(v.y >> 16));
  float t = Y - Cg;
  return mk(fmaxf(t + Co, 0.0f), fmaxf(Y + Cg, 0.0f), fmaxf(t - Co, 0.0f));
}

// ------------------------------------------------------------------ headers/color.glsl, sky.glsl
__device__ __forceinline__ V3 xyz_to_acescg(V3 v) {  // color.glsl:24-31 (column-major mat3)
  return mk((1.6410228f * v.x + -0.32480323f * v.y) + -0.23642465f * v.z,
            (-0.66366285f * v.x + 1.6153315f * v.y) + 0.016756356f * v.z,
            (0.011721907f * v.x + -0.0082844375f * v.y) + 0.9883947f * v.z);
}
// The sky model is radiance (fp16 planes, 1e-3 parity tolerance), not geometry: it runs on the hardware's 1-ulp
// transcendentals (v_exp_f32, v_rcp_f32, v_sqrt_f32) instead of the correctly rounded library routines, and
// pow(x, 1.5) is x * sqrt(x). About a fifth of the instructions of the libm version per sky evaluation.
__device__ __forceinline__ float fast_div(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
__device__ float sky_internal(DUST_RO(float) c, float cos_theta, float gamma, float cos_gamma) {  // sky.glsl:1-15
  float expM = __expf(c[4] * gamma);
  float rayM = cos_gamma * cos_gamma;
  float base = (1.0f + c[8] * c[8]) - (2.0f * c[8]) * cos_gamma;
  float mieM = fast_div(1.0f + rayM, base * __builtin_amdgcn_sqrtf(base));
  float zenith = __builtin_amdgcn_sqrtf(cos_theta);
  return (1.0f + c[0] * __expf(fast_div(c[1], cos_theta + 0.01f))) *
         ((((c[2] + c[3] * expM) + c[5] * rayM) + c[6] * mieM) + c[7] * zenith);
}
__device__ V3 sky_radiance(DUST_RO(float) s, V3 dir) {  // sky.glsl:18-79
  if (s[49] <= 0.0f) return mk(0, 0, 0);
  float cos_theta = gclamp(dir.y, 0.0f, 1.0f);
  float cos_gamma = dot3(dir, mk(s[48], s[49], s[50]));
  float gamma = acosf(cos_gamma);
  float x = sky_internal(s, cos_theta, gamma, cos_gamma) * s[9];
  float y = sky_internal(s + 16, cos_theta, gamma, cos_gamma) * s[25];
  float z = sky_internal(s + 32, cos_theta, gamma, cos_gamma) * s[41];
  return xyz_to_acescg(mk(x * 683.0f, y * 683.0f, z * 683.0f));
}
__device__ V3 sun_radiance(DUST_RO(float) s, V3 dir) {  // sky.glsl:81-113
  float cos_gamma = dot3(dir, mk(s[48], s[49], s[50]));
  if (cos_gamma < 0.0f || dir.y < 0.0f) return mk(0, 0, 0);
  float sol_rad_sin = sinf(s[55]);
  float ar2 = 1.0f / (sol_rad_sin * sol_rad_sin);
  float singamma = 1.0f - (cos_gamma * cos_gamma);
  float sc2 = 1.0f - (ar2 * singamma) * singamma;
  if (sc2 <= 0.0f) return mk(0, 0, 0);
  float sc = sqrtf(sc2);
  V3 dark = mk(s[10], s[26], s[42]);
  dark.x += s[11] * sc; dark.y += s[27] * sc; dark.z += s[43] * sc;
  float cur = sc;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    cur *= sc;
    dark.x += s[12 + i] * cur; dark.y += s[28 + i] * cur; dark.z += s[44 + i] * cur;
  }
  return xyz_to_acescg(mk(s[52] * dark.x, s[53] * dark.y, s[54] * dark.z));
}

// ------------------------------------------------------------------ the intersection shaders
constexpr int kDdaMaxIters = 64;  // the reference loop is unbounded; a NaN ray would hang the GPU there

__device__ __forceinline__ bool grid_clear(uint32_t m1, uint32_t m2, uint32_t hit) {  // hit.rint:13-15
  return ((hit < 32u) ? (m1 & (1u << (hit & 31u))) : (m2 & (1u << ((hit - 32u) & 31u)))) == 0;
}
__device__ __forceinline__ uint32_t encode_index(int px, int py, int pz) {  // hit.rint:30-32 on u8vec3
  return (((uint32_t)px << 4) | ((uint32_t)py << 2) | ((uint32_t)pz & 0xFFu)) & 0xFFu;
}
__device__ __forceinline__ void intersect_aabb04(V3 o, V3 d, V3 rd, float& t_min, float& t_max) {  // hit.rint:20-28, rd = 1 / d
  const bool ix = fabsf(rd.x) == INFINITY, iy = fabsf(rd.y) == INFINITY, iz = fabsf(rd.z) == INFINITY;
  float ax = div_by(0.0f - o.x, d.x, rd.x, ix), ay = div_by(0.0f - o.y, d.y, rd.y, iy), az = div_by(0.0f - o.z, d.z, rd.z, iz);
  float bx = div_by(4.0f - o.x, d.x, rd.x, ix), by = div_by(4.0f - o.y, d.y, rd.y, iy), bz = div_by(4.0f - o.z, d.z, rd.z, iz);
  t_min = fmaxf(fmaxf(fminf(ax, bx), fminf(ay, by)), fminf(az, bz));
  t_max = fminf(fminf(fmaxf(ax, bx), fmaxf(ay, by)), fmaxf(az, bz));
}

// RT 0: primary/hit.rint:43-131. RT 1: ambient_occlusion.rint:46-134. RT 2,3: rough.rint:42-59.
// o is brick-local (objOrigin - block.position). Returns true when reportIntersectionEXT is reached.
template <int RT>
__device__ bool brick_intersect(V3 o, V3 d, V3 tc, uint32_t m1, uint32_t m2, float tmin, float& t_out, uint32_t& voxel) {
  float t0, t1;
  intersect_aabb04(o, d, tc, t0, t1);
  if (t0 >= t1) return false;
  if (RT >= 2) {
    if (m1 == 0 && m2 == 0) return false;
    t_out = t0;
    voxel = 0;
    return true;
  }
  if (t1 <= 0.0f) return false;
  if (RT == 1) {
    if (t0 <= 8.0f && 8.0f <= t1) {
      if (!(m1 == 0 && m2 == 0)) { t_out = t0; voxel = 0xFF; return true; }
      return false;
    }
  }
  float hd = fmaxf(t0, tmin);
  int px = f2i_clamp(floorf(o.x + d.x * hd), 0, 3), py = f2i_clamp(floorf(o.y + d.y * hd), 0, 3),
      pz = f2i_clamp(floorf(o.z + d.z * hd), 0, 3);
  V3 st = mk(gsign(d.x), gsign(d.y), gsign(d.z));
  // tc = 1.0 / dir (hit.rint:88), hoisted: dir is the same for every brick of an instance
  V3 tb = mk(tc.x * o.x, tc.y * o.y, tc.z * o.z);
  V3 tm = mk(((float)px + fmaxf(st.x, 0.0f)) * tc.x - tb.x, ((float)py + fmaxf(st.y, 0.0f)) * tc.y - tb.y,
             ((float)pz + fmaxf(st.z, 0.0f)) * tc.z - tb.z);
  V3 td = mk((1.0f * tc.x) * st.x, (1.0f * tc.y) * st.y, (1.0f * tc.z) * st.z);
  const int sx = (int)st.x, sy = (int)st.y, sz = (int)st.z;
  uint32_t hit = encode_index(px, py, pz);
  int iters = 0;
  while (grid_clear(m1, m2, hit)) {
    if (++iters > kDdaMaxIters) return false;
    // hit.rint:103-117. The shader multiplies by comp = step(tMax.xyz, tMax.zxy) * step(tMax.xyz, tMax.yzx), a 0/1
    // vector (ties and NaNs set more than one component: step(edge, x) is 1 unless x < edge); selecting on the same
    // predicates gives the same position and the same tMax -- x * 1 is x, tMax + x * 0 is tMax (up to the sign of a
    // zero tMax, and except for direction components in the denormal range, where x * 0 is NaN) -- in half the VALU work
    // step(tMax.xyz, tMax.zxy) * step(tMax.xyz, tMax.yzx) component by component is "not (the smallest of the three,
    // NaNs ignored, is below this one)": one v_min3 (needed for the hit distance anyway) and three compares
    // (tests/cpp/division_identity_test.c walks every combination of NaN, infinities, zeros and finite values)
    hd = fminf(fminf(tm.x, tm.y), tm.z);
    const bool bx = !(hd < tm.x), by = !(hd < tm.y), bz = !(hd < tm.z);
    px += bx ? sx : 0;
    py += by ? sy : 0;
    pz += bz ? sz : 0;
    if (hd + 0.001f >= t1) return false;
    tm.x = bx ? tm.x + td.x : tm.x;
    tm.y = by ? tm.y + td.y : tm.y;
    tm.z = bz ? tm.z + td.z : tm.z;
    hit = encode_index(px, py, pz);
  }
  t_out = hd / 1.0f;
  voxel = hit;
  return true;
}

// ------------------------------------------------------------------ traversal
struct Hit {
  float t;
  uint32_t inst, block, voxel;
  bool found;
};
struct LaneStats {
  uint32_t rays, instances_tested, upper_descents, mid_descents, bricks_tested, hits;
};
struct MidCache {  // the 16-cell the ray was last in and its mid-node index (saves the LDS root lookup)
  int key;
  uint32_t mid;
  uint64_t mask4;  // DEEP variants only: which 4^3 cells of that 16-cell hold a brick (all ones for two-level models)
};
__device__ __forceinline__ void midcache_reset(MidCache& mc) { mc.key = -1; mc.mid = 0; mc.mask4 = 0; }

// Ray against an axis-aligned box, conservative (slack on both ends). inv = 1/d per component (IEEE divide). A zero
// direction component constrains nothing along t and instead requires the origin to lie in the slab (1e-3 margin);
// it is handled with selects, not branches: this runs once per (ray, candidate instance) in a wave-uniform loop, and
// divergent branches there cost more exec-mask bookkeeping than the arithmetic they skip.
template <class PLo, class PHi>
__device__ __forceinline__ bool slab_box(V3 o, V3 d, V3 inv_d, PLo lo, PHi hi, float& te, float& tx) {
  te = -INFINITY; tx = INFINITY;
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, iv[3] = {inv_d.x, inv_d.y, inv_d.z};
  bool ok = true;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float l = lo[a], h = hi[a];
    const float t0 = (l - oo[a]) * iv[a], t1 = (h - oo[a]) * iv[a];
    const bool nz = dd[a] != 0.0f;
    te = fmaxf(te, nz ? fminf(t0, t1) : -INFINITY);
    tx = fminf(tx, nz ? fmaxf(t0, t1) : INFINITY);
    ok = ok && (nz || !(oo[a] < l - 1e-3f || oo[a] > h + 1e-3f));
  }
  const float slack = 1e-5f * (fabsf(te) + fabsf(tx)) + 1e-5f;
  return ok && !(te > tx + slack) && !(tx + slack < 0.0f);
}
// the same test when no ray of the wave has a zero direction component (the caller checked): 8 operations per axis
template <class PLo, class PHi>
__device__ __forceinline__ bool slab_box_nonzero(V3 o, V3 inv_d, PLo lo, PHi hi, float& te, float& tx) {
  const float t0x = (lo[0] - o.x) * inv_d.x, t1x = (hi[0] - o.x) * inv_d.x;
  const float t0y = (lo[1] - o.y) * inv_d.y, t1y = (hi[1] - o.y) * inv_d.y;
  const float t0z = (lo[2] - o.z) * inv_d.z, t1z = (hi[2] - o.z) * inv_d.z;
  te = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
  tx = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
  const float slack = 1e-5f * (fabsf(te) + fabsf(tx)) + 1e-5f;
  return !(te > tx + slack) && !(tx + slack < 0.0f);
}

// N16 lookup: bit test + rank. Root nodes staged in LDS are read with ds_read, the rest from memory.
__device__ __forceinline__ bool n16_child(DUST_RO(uint8_t) node, int lds_slot, uint32_t idx, uint32_t& child) {
  uint32_t w = idx >> 6, bit = idx & 63u;
  uint64_t word;
  uint32_t pre;
  if (lds_slot >= 0) {
    word = reinterpret_cast<const uint64_t*>(g_lds + (uint32_t)lds_slot * kN16LdsBytes)[w];
    if (!((word >> bit) & 1ull)) return false;
    pre = reinterpret_cast<const uint16_t*>(g_lds + (uint32_t)lds_slot * kN16LdsBytes + 512)[w];
  } else {
    word = ((DUST_RO(uint64_t))node)[w];
    if (!((word >> bit) & 1ull)) return false;
    pre = ((DUST_RO(uint16_t))(node + 512))[w] + *(DUST_RO(uint32_t))(node + 640);
  }
  child = pre + (uint32_t)__popcll(word & ((1ull << bit) - 1ull));
  return true;
}

#ifndef DUST_TRANSPOSE_RAYS
#define DUST_TRANSPOSE_RAYS 32   // rays of an incoherent packet still wanting a candidate batch at or below which the box scan runs lane = box (trace_ray); 0: never
#endif
constexpr uint32_t kDirectCell = 0x100u;  // find_brick's cell_log2 flag: a 16-cell whose bricks the walk tests one by one
#ifndef DUST_DIRECT_BRICKS
#define DUST_DIRECT_BRICKS 4
#endif
constexpr uint32_t kDirectBricks = DUST_DIRECT_BRICKS;  // ... when it holds at most this many
// ... and when the packet's rays are neighbours (ray types 0 and 1: camera, sun and AO rays), which then meet their sparse cells on
// the same trips: 2.30 -> 1.91 ms for the 4096^3 tree's primary + AO kernel. The gather and surfel rays of a packet each meet
// theirs on a trip of their own, every one of which then lasts as long as the longer direct test: 1.84 -> 2.13 ms. Not for them.
#ifndef DUST_WHOLE_CELLS_MAX_RT
#define DUST_WHOLE_CELLS_MAX_RT 1
#endif
template <int RT> constexpr bool kWholeCells = RT <= DUST_WHOLE_CELLS_MAX_RT;
// Deepest occupied cell containing voxel (x,y,z). Returns the brick's 64-bit occupancy (0 = no brick),
// cell_log2 = size of the cell that was found empty (2 when a brick exists), key = mid_index*64 + child bit,
// which orders bricks exactly like the block index does (both are depth-first).
// One dependent memory access per call: root in LDS -> mid index -> dense_mask[mid*64 + bit].
// ray (DEEP variants): the object-space ray o + t d with inv_d = 1 / d, for the occupied-box test of a 16-cell
template <int MODE, class Model>
__device__ __forceinline__ uint64_t find_brick(const Model& m, int x, int y, int z, uint32_t& cell_log2, uint32_t& key,
                                               MidCache& mc, LaneStats& st, bool count, V3 ray_o, V3 ray_d, V3 ray_inv_d, bool whole_cells = false,
                                               bool zero_axis = true) {
  // zero_axis (wave-uniform): some ray of the wave may have a zero direction component -- the occupied-box test then takes the general slab test
  // count: the call comes from the walk itself (its traversal is tallied), not from a neighbour visit.
  // whole_cells: hand a sparse 16-cell back whole (below) -- the walks of the camera, sun and AO rays ask for that
  const int k16 = ((x >> 4) << 16) | ((y >> 4) << 8) | (z >> 4);
  if (k16 != mc.key) {
    uint32_t mid_index;
    if (!DEEP || m.n_levels == 2) {  // (kernels of scenes without a 4096^3 model hold no three-level code at all)
      uint32_t idx = ((uint32_t)(x >> 4) << 8) | ((uint32_t)(y >> 4) << 4) | (uint32_t)(z >> 4);
      if (!n16_child(m.root, m.lds_slot, idx, mid_index)) { cell_log2 = 4; return 0; }
      if (COUNT && count) st.upper_descents += 1;
    } else {
      uint32_t idx = ((uint32_t)(x >> 8) << 8) | ((uint32_t)(y >> 8) << 4) | (uint32_t)(z >> 8);
      uint32_t l2;
      if (!n16_child(m.root, m.lds_slot, idx, l2)) { cell_log2 = 8; return 0; }
      if (COUNT && count) st.upper_descents += 1;
      uint32_t idx2 = ((uint32_t)((x >> 4) & 15) << 8) | ((uint32_t)((y >> 4) & 15) << 4) | (uint32_t)((z >> 4) & 15);
      if (DEEP) {
        // one 16-byte load instead of mask word -> rank prefix + base -> (per 4-cell) brick mask: the cell's mid index and
        // its child mask arrive together, and the walk then crosses the 16-cell's empty 4-cells without touching memory
        // (Round 6 requested the NEXT 16-cell's record a trip ahead -- which cell the ray enters after this one is geometry, not data --:
        //  k_primary_ao<2> 2.52 -> 3.26 ms. The deep walk is bound by issue at 128 registers, not by this load's latency; docs/EXPERIMENTS.md.)
        const u32x4 cell = *(DUST_RO(u32x4))(m.l2_cells + ((size_t)l2 * 4096u + idx2));
        if (cell.x == 0xFFFFFFFFu) { cell_log2 = 4; return 0; }
        if (COUNT && count) st.upper_descents += 1;
        // At low occupancy a 16-cell holds one or two bricks in its 64 places. If the ray misses the box of the occupied
        // 4-cells -- grown by 0.05 voxel, far more than the walk's own tolerance delta <= 1e-2, so nothing the conservative
        // walk or its neighbour visits could test lies outside --, the whole 16-cell is crossed in one step like an empty one.
        {
          const float bx = (float)(x & ~15), by = (float)(y & ~15), bz = (float)(z & ~15);
          const float lo[3] = {bx + 4.0f * (float)(cell.y & 3u) - 0.05f, by + 4.0f * (float)((cell.y >> 2) & 3u) - 0.05f,
                               bz + 4.0f * (float)((cell.y >> 4) & 3u) - 0.05f};
          const float hi[3] = {bx + 4.0f * (float)(((cell.y >> 6) & 3u) + 1u) + 0.05f, by + 4.0f * (float)(((cell.y >> 8) & 3u) + 1u) + 0.05f,
                               bz + 4.0f * (float)(((cell.y >> 10) & 3u) + 1u) + 0.05f};
          float te, tx;
          if (!(zero_axis ? slab_box(ray_o, ray_d, ray_inv_d, lo, hi, te, tx) : slab_box_nonzero(ray_o, ray_inv_d, lo, hi, te, tx))) { cell_log2 = 4; return 0; }
        }
        mid_index = cell.x;
#ifndef DUST_NO_DIRECT_CELLS
        // A 16-cell with a handful of bricks (at 1 % occupancy: one in 72 % of the occupied cells, two in 22 %) is not walked
        // 4-cell by 4-cell: the walk gets the whole cell back (kDirectCell), tests each of its bricks whose grown box the ray
        // meets (test_cell_bricks) and leaves the 16-cell in one step.
        // (key = the cell's mid node, return value = its child mask; the walk's cache is left as it is)
        const uint64_t cm = ((uint64_t)cell.w << 32) | cell.z;
        if (whole_cells && __popcll(cm) <= (int)kDirectBricks) { key = mid_index; cell_log2 = 4u | kDirectCell; return cm; }
#endif
        mc.mask4 = ((uint64_t)cell.w << 32) | cell.z;
      } else {
        if (!n16_child(m.l2 + (size_t)l2 * kN16Bytes, -1, idx2, mid_index)) { cell_log2 = 4; return 0; }
        if (COUNT && count) st.upper_descents += 1;
      }
    }
    if (DEEP && m.n_levels == 2) mc.mask4 = ~0ull;  // a two-level model in a scene that also holds a deep one: every step looks its cell up
    mc.key = k16; mc.mid = mid_index;
  }
  const uint32_t bit = ((uint32_t)((x >> 2) & 3) << 4) | ((uint32_t)((y >> 2) & 3) << 2) | (uint32_t)((z >> 2) & 3);
  key = mc.mid * 64u + bit;
  cell_log2 = 2;
  if (DEEP && !((mc.mask4 >> bit) & 1ull)) {  // empty 4-cell, known without a load
#ifndef DUST_NO_OCTANT_SKIP
    // ... and if the seven 4-cells that share its octant of the 16-cell are empty as well (bits {0,1} x {0,4} x {0,16} above the
    // octant's corner: at 1 % occupancy a 16-cell holds one or two bricks, so 92 % of the octants are), the walk leaves the
    // 8-cell in one step. Two shifts and a compare on a mask that is in registers already.
    if (!(mc.mask4 & (0x0000000000330033ull << (bit & 0x2Au)))) cell_log2 = 3;
#endif
    return 0;
  }
  const uint64_t mask = m.dense_mask[key];
  if (COUNT && count && mask != 0) st.mid_descents += 1;
  return mask;
}

// block index (gl_PrimitiveID) of a brick key: first_block of its mid node + rank of the child bit
__device__ __forceinline__ uint32_t resolve_block(ModelRef m, uint32_t key) {
  const u32x4 n = *(DUST_RO(u32x4))(m.mid + (key >> 6));
  const uint64_t mm = ((uint64_t)n.y << 32) | n.x;
  return n.z + (uint32_t)__popcll(mm & ((1ull << (key & 63u)) - 1ull));
}

// the 24-byte Block record as three 8-byte loads
__device__ __forceinline__ DustHipBlock load_block(DUST_RO(DustHipBlock) p) {
  DUST_RO(u32x2) q = (DUST_RO(u32x2))p;
  const u32x2 q0 = q[0], q1 = q[1], q2 = q[2];
  DustHipBlock b;
  b.x = (uint16_t)q0.x; b.y = (uint16_t)(q0.x >> 16); b.z = (uint16_t)q0.y; b.w = (uint16_t)(q0.y >> 16);
  b.mask = ((uint64_t)q1.y << 32) | q1.x;
  b.material_ptr = q2.x; b.avg_albedo = q2.y;
  return b;
}

// run the ray type's intersection routine on one brick and apply Vulkan's accept rule
// (tmin <= t <= current tmax; equal t: lower (instance, block) wins -- see oracle/shade.c header)
template <int RT, int MODE>
__device__ __forceinline__ void test_brick(uint64_t mask, uint32_t inst, uint32_t key, int bx, int by, int bz, V3 o,
                                           V3 d, V3 inv_d, float tmin, float tmax, Hit& best, LaneStats& st) {
  V3 ol = mk(o.x - (float)bx, o.y - (float)by, o.z - (float)bz);  // hit.rint:137-140
  float t;
  uint32_t vox;
  if (COUNT) st.bricks_tested += 1;
  if (!brick_intersect<RT>(ol, d, inv_d, (uint32_t)mask, (uint32_t)(mask >> 32), tmin, t, vox)) return;
  DBG_PRINT("    brick inst %u key %u at %d,%d,%d: t=%.9g (best found=%d t=%.9g inst=%u blk=%u)\n", inst, key, bx, by, bz, t, (int)best.found, best.t, best.inst, best.block);
  const float cur = best.found ? best.t : tmax;
  if (!(t >= tmin && t <= cur)) return;
  if (best.found && t == best.t) {
    if (inst > best.inst || (inst == best.inst && key >= best.block)) return;
  }
  best.found = true; best.t = t; best.inst = inst; best.block = key; best.voxel = vox;
}

// DEEP variants: every brick of the sparse 16-cell around ijk (mid node `mid`, child mask `child_mask`) that the ray can touch.
// The 0.05-voxel growth of the brick's box is far more than the walk's tolerance delta <= 1e-2 (same argument as the
// occupied-box test in find_brick), so this is a superset of what the 4-cell walk and its neighbour visits would have tested
// inside the cell; bricks that start beyond the best hit so far are skipped before their mask is loaded.
template <int RT, int MODE>
__device__ __forceinline__ void test_cell_bricks(ModelRef m, uint32_t inst, uint32_t mid, uint64_t child_mask, int x, int y, int z, V3 o, V3 d, V3 inv_d,
                                                 float tmin, float tmax, Hit& best, LaneStats& st, bool zero_axis) {
  uint64_t mm = child_mask;
  const int gx = x & ~15, gy = y & ~15, gz = z & ~15;
  while (mm != 0) {
    const uint32_t bit = (uint32_t)__builtin_ctzll(mm);
    mm &= mm - 1ull;
    const int bx = gx + (int)((bit >> 4) & 3u) * 4, by = gy + (int)((bit >> 2) & 3u) * 4, bz = gz + (int)(bit & 3u) * 4;
    const float lo[3] = {(float)bx - 0.05f, (float)by - 0.05f, (float)bz - 0.05f};
    const float hi[3] = {(float)bx + 4.05f, (float)by + 4.05f, (float)bz + 4.05f};
    float te, tx;
    if (!(zero_axis ? slab_box(o, d, inv_d, lo, hi, te, tx) : slab_box_nonzero(o, inv_d, lo, hi, te, tx))) continue;
    if (te * (1.0f - 2e-6f) > (best.found ? best.t : tmax)) continue;
    const uint32_t key = mid * 64u + bit;
    const uint64_t mask = m.dense_mask[key];
    if (mask == 0) continue;
    if (COUNT) st.mid_descents += 1;
    test_brick<RT, MODE>(mask, inst, key, bx, by, bz, o, d, inv_d, tmin, tmax, best, st);
  }
}

// The cold part of the conservative walk (see trace_instance): the entry point of the cell at ijk may lie within delta
// of further brick planes; decide exactly which, and test every brick around that edge / corner. A real call, not
// inlined: the hot loop then carries neither this code's registers nor a second copy of the lookup and brick test
// (inlined, the same code cost the fused kernel 15 %). State goes in and out by value so nothing of the caller's has
// its address taken.
// Everything crosses the call in registers: scalars in, one 8-word vector out
// {t, inst, block, voxel, found, mc.key, mc.mid, bricks_tested}. (Structs by value go through the stack here, and a build
// that passed Hit that way resolved equal-t ties between overlapping instances differently from the inlined code.)
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
template <int RT, int MODE>
__device__ __attribute__((noinline)) u32x8 visit_neighbours(const DUST_CONST_AS DevModel* mp, uint32_t inst, float ox, float oy, float oz,
                                                           float dx, float dy, float dz, float ix, float iy, float iz,
                                                           float tmin, float tmax, float t, int i0, int i1, int i2, uint32_t stepped,
                                                           float best_t, uint32_t best_inst, uint32_t best_block, uint32_t best_voxel,
                                                           uint32_t best_found, int mc_key, uint32_t mc_mid, uint32_t cell_log2,
                                                           uint32_t mc_mask_lo, uint32_t mc_mask_hi) {
  ModelRef m = *mp;
  const V3 o = mk(ox, oy, oz), d = mk(dx, dy, dz), inv_d = mk(ix, iy, iz);
  Hit best;
  best.t = best_t; best.inst = best_inst; best.block = best_block; best.voxel = best_voxel; best.found = best_found != 0;
  MidCache mc;
  // DEEP: the caller's 16-cell comes along with its child mask, so a neighbour inside the same 16-cell (three brick planes in
  // four are) is known to be empty without a lookup; the caller keeps its own cache whatever this call moves on to
  mc.key = mc_key; mc.mid = mc_mid; mc.mask4 = DEEP ? ((uint64_t)mc_mask_hi << 32) | mc_mask_lo : 0ull;
  LaneStats st = {0, 0, 0, 0, 0, 0};
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  const int ijk[3] = {i0, i1, i2};
  uint32_t near_neg = 0, near_pos = 0;  // bit a: entry point within delta of the cell's low / high plane on axis a
  uint32_t unstepped_near = 0;
  // DEEP: the walk's cell may be a whole 16-cell that holds nothing untested (empty, missed by more than any delta, or tested brick
  // by brick). Bricks INSIDE it need no visit; a neighbour cell lies outside it iff one of its axes crosses a face of the 16-cell --
  // a stepped axis (the ray came in through that face) or an unstepped one whose near plane is a multiple of 16. The planes
  // themselves are still brick planes (multiples of 4): an entry point on the 16-cell's z face within delta of x = 900 has the
  // brick across BOTH planes to test, although x = 900 is no face of the 16-cell (tools/stress_parity.py STRESS_DEEP, seed 20833).
  const bool whole16 = DEEP && cell_log2 >= 4u;
  uint32_t leaves16 = stepped;  // axes on which the other side is outside the 16-cell
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int blo = (int)m.bmin[a], bhi = (int)m.bmax[a] - 1;  // voxel range that holds bricks (tight bounds, multiples of 4)
    const float p = oo[a] + dd[a] * t;
    const float delta = 1e-6f * ((fabsf(oo[a]) + fabsf(p)) + 16.0f);
    const int b0 = ijk[a] & ~3;
    const float q = p - (float)b0;
    // a plane only matters if bricks can exist on its far side
    if (stepped & (1u << a)) { if (dd[a] > 0.0f) { if (b0 - 1 >= blo) near_neg |= 1u << a; } else if (b0 + 4 <= bhi) near_pos |= 1u << a; }
    else if (q <= delta) { if (b0 - 1 >= blo) { near_neg |= 1u << a; unstepped_near |= 1u << a; if ((b0 & 15) == 0) leaves16 |= 1u << a; } }
    else if (q >= 4.0f - delta) { if (b0 + 4 <= bhi) { near_pos |= 1u << a; unstepped_near |= 1u << a; if (((b0 + 4) & 15) == 0) leaves16 |= 1u << a; } }
  }
  const uint32_t nearm = near_neg | near_pos;
  if (unstepped_near != 0 || __popc(stepped & nearm) > 1) {
#pragma unroll 1
    for (uint32_t sub = 1; sub < 8; ++sub) {  // one visit per non-empty subset of the near axes
      if ((sub & ~nearm) != 0 || (stepped != 0 && sub == stepped)) continue;  // sub == stepped: the cell we came from
      if (whole16 && (sub & leaves16) == 0) continue;                         // a cell of the 16-cell the walk is in: nothing untested there
      int c[3] = {ijk[0], ijk[1], ijk[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a)
        if (sub & (1u << a)) c[a] = (near_neg & (1u << a)) ? (ijk[a] & ~3) - 1 : (ijk[a] & ~3) + 4;
      uint32_t cl2, key;
      const uint64_t mask = find_brick<MODE>(m, c[0], c[1], c[2], cl2, key, mc, st, false, o, d, inv_d);
      if (mask != 0) test_brick<RT, MODE>(mask, inst, key, c[0] & ~3, c[1] & ~3, c[2] & ~3, o, d, inv_d, tmin, tmax, best, st);
    }
  }
  u32x8 out;
  out[0] = __float_as_uint(best.t); out[1] = best.inst; out[2] = best.block; out[3] = best.voxel; out[4] = best.found ? 1u : 0u;
  out[5] = DEEP ? (uint32_t)mc_key : (uint32_t)mc.key; out[6] = DEEP ? mc_mid : mc.mid; out[7] = st.bricks_tested;
  return out;
}

// A ray with a direction component that is exactly zero whose origin lies exactly ON a brick plane of that axis (o_a a multiple of 4 in
// object space) hits NOTHING in the instance: for every brick the local coordinate o_a - b_a is 0, 4 or outside [0, 4], and the
// intersection shaders' slab arithmetic (hit.rint:20-28, intersect_aabb04) then yields {NaN, +-inf} or two infinities of one sign for that
// axis -- minNum / maxNum drop the NaN, t_min = +inf or t_max = -inf, and `t_min >= t_max` rejects the brick (the oracle's brute force over
// every brick runs the same arithmetic and finds the same nothing). Only a ray STRICTLY inside a slab it never leaves can hit a brick of it.
// The case is not exotic: the reference's default sun has x == 0 exactly (pipeline/sky.rs:20), a surfel's sun ray starts at its brick's
// centre + 2.01 n (surfel.rgen:27), and an instance whose lattice is offset by 2 (mod 4) from the surfel's own has a brick plane exactly
// there: the conservative walk then called visit_neighbours on 55-89 % of its trips -- for bricks on either side that cannot be hit --,
// and those sun items were the surfel trace's longest (docs/EXPERIMENTS.md, rounds 5 and 6). The incoherent ray types only: a camera or
// AO ray's origin is not on the lattice.
template <int RT>
__device__ __forceinline__ bool in_brick_plane(V3 o, V3 d) {
#ifdef DUST_NO_PLANE_EXIT
  return false;
#else
  if (RT < 2) return false;
  const float qx = o.x * 0.25f, qy = o.y * 0.25f, qz = o.z * 0.25f;  // (exact: a power of two)
  return (d.x == 0.0f && qx == rintf(qx)) || (d.y == 0.0f && qy == rintf(qy)) || (d.z == 0.0f && qz == rintf(qz));
#endif
}

// Hierarchical traversal of one instance in object space. Visits, front to back, a SUPERSET of the
// bricks whose intersection routine can report an accepted hit: exit planes are recomputed from
// integer cell coordinates at every step (no accumulated error), and whenever the walk passes within
// delta of a brick-grid edge or corner every brick around it is tested too (DESIGN.md "Conservative walk").
// One loop iteration handles one cell; when its entry point lies within delta of other brick planes, the bricks across
// those planes (one per non-empty subset of the near axes) are tested by visit_neighbours before the walk advances.
template <int RT, int MODE>
__device__ void trace_instance(ModelRef m, uint32_t inst, V3 o, V3 d, float tmin, float tmax, bool any_hit,
                               Hit& best, LaneStats& st) {
  PROF_ENTER(P_SETUP);
  const V3 inv_d = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  float te, tx;
  const bool in_bounds = slab_box(o, d, inv_d, m.bmin, m.bmax, te, tx);
  PROF_LEAVE(P_SETUP);
  if (!in_bounds) return;
  if (in_brick_plane<RT>(o, d)) return;
  PROF_ENTER(P_CAND);
  const int E = (int)m.extent;
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z}, inv[3] = {inv_d.x, inv_d.y, inv_d.z};
  float t = fmaxf(te, 0.0f);
  if (RT >= 2) t = fmaxf(t, tmin * (1.0f - 1e-6f));
  int ijk[3];
  // Near-plane screen (see the loop): |p/4 - rint(p/4)| <= near_tol flags an entry point that may lie within
  // delta = 1e-6 (|o_a| + |p_a| + 16) of a brick plane. One tolerance for the whole visit: 3e-7 (20 % above delta / 4,
  // which covers evaluating p at the step's exit time instead of the clamped t) times the largest |o_a| + |p_a| the
  // walk can meet (p is linear in t, so the ends of [te, tx] bound it).
  float reach = 0.0f;
  bool screen = false;  // does the CURRENT cell's entry point need the exact near-plane test?
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // the cell the ray is moving into: floor for d >= 0, ceil - 1 for d < 0 (differs only on a cell plane), kept inside
    // the tight bounds: the start point is the origin inside them or the entry point on them, and an entry point that
    // rounding left a hair outside would otherwise start the walk one (empty) cell early, next to the plane, every time
    const float p = oo[a] + dd[a] * t;
    ijk[a] = f2i_clamp(dd[a] < 0.0f ? ceilf(p) - 1.0f : floorf(p), (int)m.bmin[a], (int)m.bmax[a] - 1);
    reach = fmaxf(reach, fabsf(oo[a]) + fmaxf(fabsf(p), fabsf(oo[a] + dd[a] * tx)));
  }
  const float near_tol = 3.0e-7f * (reach + 16.0f);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    // The first cell gets the exact test's own shape with the looser tolerance, because it can tell what the cheap
    // distance-to-a-multiple-of-4 cannot: a walk that starts on the model's bounds (every visit from outside does:
    // the bounds are brick planes) is "near a plane" there by construction, but no brick exists beyond it, and that
    // is not worth a call. (A plane only matters if bricks can exist on its far side.)
    const int b0 = ijk[a] & ~3, blo = (int)m.bmin[a], bhi = (int)m.bmax[a] - 1;
    const float q = (oo[a] + dd[a] * t) - (float)b0;
    screen = screen | ((q <= 4.0f * near_tol) & (b0 - 1 >= blo)) | ((q >= 4.0f - 4.0f * near_tol) & (b0 + 4 <= bhi));
  }
  uint32_t stepped = 0;   // bit a: axis a crossed a plane on the last step
  uint32_t cl_main = 2;
  MidCache mc;
  midcache_reset(mc);
  const bool zero_axis = DEEP && __any(d.x == 0.0f || d.y == 0.0f || d.z == 0.0f);  // (of the lanes in this visit)
  bool prev_whole = false;  // DEEP: the cell the walk has just left was a whole 16-cell (or larger) with nothing untested in it
  const float tx_stop = tx * (1.0f + 1e-5f) + 1e-5f;
  PROF_LEAVE(P_CAND);
  for (int guard = 0; guard < 200000; ++guard) {
    PROF_COUNT(P_N_STEPS, 1);
    // a tile that turns out to be a long one earns its priority as it goes (g_tile_start above); the camera, sun and AO rays only:
    // the GI kernels measured no gain
    if (RT <= 1 && (guard & 15) == 15) {
      const uint32_t w = threadIdx.x >> 6;
      const uint32_t el = (uint32_t)__builtin_amdgcn_s_memtime() - g_tile_start[w];
      const uint32_t earned = el > DUST_DYN_T3 ? 3u : (el > DUST_DYN_T2 ? 2u : (el > DUST_DYN_T1 ? 1u : 0u));
      const uint32_t have = (uint32_t)__builtin_amdgcn_readfirstlane((int)g_tile_prio[w]);
      if ((uint32_t)__builtin_amdgcn_readfirstlane((int)earned) > have) {
        if (earned == 3u) __builtin_amdgcn_s_setprio(3); else if (earned == 2u) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(1);
        if ((threadIdx.x & 63u) == (uint32_t)(__ffsll((long long)__ballot(1)) - 1)) g_tile_prio[w] = earned;
      }
    }
    {
      const float limit = best.found ? best.t : tmax;
      if (t * (1.0f - 2e-6f) > limit) return;
      if (any_hit && best.found) return;
    }
    // The cell's lookup issues the one dependent memory access of the step (the brick mask). The step out of the cell
    // needs none of it -- only the size of the cell, which the root lookup in LDS already decided -- so it is worked out
    // HERE, while that load is in flight, into next-cell temporaries; the brick test and the neighbour visit then run on the
    // current cell's state, and the temporaries are committed afterwards. Same operations in the same per-cell order as
    // "test, visit neighbours, advance"; the load's latency is covered by ~100 instructions of the wave's own arithmetic.
    uint32_t key;
    PROF_ENTER(P_FIND);
    uint64_t mask = find_brick<MODE>(m, ijk[0], ijk[1], ijk[2], cl_main, key, mc, st, true, o, d, inv_d, kWholeCells<RT>, zero_axis);
    bool direct = false;
    uint32_t cell_mid = 0;
    uint64_t cell_mask = 0;
    if (DEEP && (cl_main & kDirectCell)) { cl_main = 4; direct = true; cell_mid = key; cell_mask = mask; mask = 0; }
    PROF_LEAVE(P_FIND);
    // DEEP, screen raised, and the cell is a whole 16-cell with nothing untested in it (empty, missed, or about to be tested brick
    // by brick): the bricks inside it need no neighbour visit -- but a brick ACROSS the face the ray came in through does, whatever
    // plane its other axes are near (round 2's kernels looked again at 16-plane granularity only and lost one such brick in 4 000
    // random deep scenes: tools/stress_parity.py STRESS_DEEP=1, seed 20833). That brick lies in the 16-cell the walk has just
    // left: if that was itself a whole cell with nothing untested (prev_whole, below) there is nothing to do; else its child mask
    // is what the cache holds, and the call is made only if the mask has a brick there -- or for the rarer shapes (ties, two
    // near planes, a near 16-plane).
    if (DEEP && __builtin_expect(screen, 0) && cl_main >= 4u) {
      bool near16 = false, across_needed = true, sided = true;
      uint32_t near4 = 0;
      int c[3] = {ijk[0], ijk[1], ijk[2]};
#pragma unroll
      for (int a = 0; a < 3; ++a) {
        if (stepped & (1u << a)) {
          c[a] = dd[a] > 0.0f ? (ijk[a] & ~15) - 1 : (ijk[a] & ~15) + 16;  // back across the face
        } else {
          const float pa = oo[a] + dd[a] * t;
          const float r4 = pa * 0.25f;
          if (fabsf(r4 - rintf(r4)) <= near_tol) {
            near4 += 1u;
            const int b0 = ijk[a] & ~3;
            const float q = pa - (float)b0;
            // (a near plane that is a 16-cell's face has neighbours outside the cell on its own: the exact code looks)
            if (q <= 8.0f * near_tol) { c[a] = b0 - 1; near16 = near16 | ((b0 & 15) == 0); }
            else if (q >= 4.0f - 8.0f * near_tol) { c[a] = b0 + 4; near16 = near16 | (((b0 + 4) & 15) == 0); }
            else sided = false;  // (the integer cell and the point disagree about the side: let the exact code look)
          }
        }
      }
      if (stepped == 0u || near4 == 0u) across_needed = false;  // nothing lies across an entered face and near another plane
      else if (__popc(stepped) == 1 && near4 == 1u && sided) {
        const int kd = ((c[0] >> 4) << 16) | ((c[1] >> 4) << 8) | (c[2] >> 4);
        const uint32_t bd = ((uint32_t)((c[0] >> 2) & 3) << 4) | ((uint32_t)((c[1] >> 2) & 3) << 2) | (uint32_t)((c[2] >> 2) & 3);
        if (prev_whole || (kd == mc.key && !((mc.mask4 >> bd) & 1ull))) across_needed = false;
      }
      screen = (__popc(stepped) > 1) | near16 | !sided | across_needed;
    }
    // leave the cell of size 2^cl_main that contains ijk
    PROF_ENTER(P_ADVANCE);
    const int S = 1 << cl_main;
    float ta[3], tn = INFINITY;
    int cc[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      cc[a] = ijk[a] & ~(S - 1);
      if (dd[a] != 0.0f) {
        const float plane = (float)(dd[a] > 0.0f ? cc[a] + S : cc[a]);
        ta[a] = (plane - oo[a]) * inv[a];
      } else {
        ta[a] = INFINITY;
      }
      tn = fminf(tn, ta[a]);
    }
    const bool stuck = !(tn < INFINITY);
    uint32_t next_stepped = 0;
    int next_ijk[3];
    bool outside = false, next_screen = false;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (ta[a] == tn) {
        next_stepped |= 1u << a;
        next_ijk[a] = dd[a] > 0.0f ? cc[a] + S : cc[a] - 1;
        if (next_ijk[a] < 0 || next_ijk[a] >= E) outside = true;
      } else {
        const float p = oo[a] + dd[a] * tn;  // the next cell's entry point on an axis that does not cross a plane
        next_ijk[a] = f2i_clamp(floorf(p), cc[a], cc[a] + S - 1);
        const float r = p * 0.25f;
        next_screen = next_screen | (fabsf(r - rintf(r)) <= near_tol);
      }
    }
    next_screen = next_screen | (__popc(next_stepped) > 1);
    PROF_LEAVE(P_ADVANCE);
    {
      const bool have = mask != 0;
      PROF_COUNT_LANES(P_L_TRIPS, true);
      PROF_COUNT_LANES(P_L_BRICK, have);
      PROF_COUNT_LANES(P_L_EMPTY4, !have && cl_main == 2);
      PROF_COUNT_LANES(P_L_EMPTY16, !have && cl_main > 2);
      PROF_ENTER(P_BRICK);
      if (have) test_brick<RT, MODE>(mask, inst, key, ijk[0] & ~3, ijk[1] & ~3, ijk[2] & ~3, o, d, inv_d, tmin, tmax, best, st);
      if (DEEP && direct) test_cell_bricks<RT, MODE>(m, inst, cell_mid, cell_mask, ijk[0], ijk[1], ijk[2], o, d, inv_d, tmin, tmax, best, st, zero_axis);
      PROF_LEAVE(P_BRICK);
    }
    // Is the entry point within delta of further brick planes? `screen` (worked out when the walk stepped into this
    // cell, from the entry point that step computed anyway) is a cheap superset of that: the distance of p to the
    // nearest multiple of 4 on the axes that did not step, or an exact tie on exit. Almost every step skips the call.
    PROF_ENTER(P_SCREEN);
    if (__builtin_expect(screen, 0)) {
      PROF_COUNT(P_N_NEIGHBOUR_CALLS, 1);
      const u32x8 nv = visit_neighbours<RT, MODE>(&m, inst, o.x, o.y, o.z, d.x, d.y, d.z, inv_d.x, inv_d.y, inv_d.z, tmin, tmax, t,
                                                   ijk[0], ijk[1], ijk[2], stepped, best.t, best.inst, best.block, best.voxel,
                                                   best.found ? 1u : 0u, mc.key, mc.mid, cl_main,
                                                   DEEP ? (uint32_t)mc.mask4 : 0u, DEEP ? (uint32_t)(mc.mask4 >> 32) : 0u);
      best.t = __uint_as_float(nv[0]); best.inst = nv[1]; best.block = nv[2]; best.voxel = nv[3]; best.found = nv[4] != 0;
      mc.key = (int)nv[5]; mc.mid = nv[6];
      if (COUNT) st.bricks_tested += nv[7];
    }
    PROF_LEAVE(P_SCREEN);
    if (stuck || outside) return;
    // DEEP: was the cell left behind a whole 16-cell (or larger) with nothing untested in it -- empty; missed by 0.05 voxel, so no
    // brick of it comes within delta of the ray anywhere; or tested whole, so every brick the ray can touch has had its test?
    // Then the next cell's neighbour across the entered face, which lies inside it, needs no visit (the inline test above).
    if (DEEP) prev_whole = cl_main >= 4u;
    ijk[0] = next_ijk[0]; ijk[1] = next_ijk[1]; ijk[2] = next_ijk[2];
    stepped = next_stepped;
    screen = next_screen;
    t = fmaxf(t, tn);
    if (t * (1.0f - 2e-6f) > tx_stop) return;
  }
}

// ------------------------------------------------------------------ packet-level instance culling
struct Range3 { float lo[3], hi[3]; };  // per-axis interval of a per-lane vector over the wave's active rays
// Wave-wide min/max of a per-lane vector on the VALU cross-lane paths (no LDS round trips): v_min/v_max_f32 with a DPP
// source, four steps inside each 16-lane row (quad swaps, half mirror, mirror), then row_bcast:15 and row_bcast:31 carry
// the row results up so that lane 63 holds the wave's; one v_readlane each picks them up. Three chains run interleaved:
// a DPP read needs two wait states after the VALU write of its source, and the two instructions in between provide
// them. 7 instructions per reduction; written through __builtin_amdgcn_update_dpp + fminf the compiler spent a move, a
// canonicalisation and a nop on every step (24). Call with all 64 lanes executing.
#define DUST_DPP_STEP(op, ctrl)  op " %0, %0, %0 " ctrl "\n" op " %1, %1, %1 " ctrl "\n" op " %2, %2, %2 " ctrl "\n"
#define DUST_DPP_REDUCE(op, x, y, z)                                         \
  asm("s_nop 1\n"                                                           \
      DUST_DPP_STEP(op, "quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")   \
      DUST_DPP_STEP(op, "quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf")   \
      DUST_DPP_STEP(op, "row_half_mirror row_mask:0xf bank_mask:0xf")       \
      DUST_DPP_STEP(op, "row_mirror row_mask:0xf bank_mask:0xf")            \
      DUST_DPP_STEP(op, "row_bcast:15 row_mask:0xa bank_mask:0xf")          \
      DUST_DPP_STEP(op, "row_bcast:31 row_mask:0xc bank_mask:0xf")          \
      : "+v"(x), "+v"(y), "+v"(z))
__device__ __forceinline__ Range3 wave_range(bool active, V3 v) {
  auto top = [](float x) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), 63)); };
  Range3 r;
  {
    float l0 = active ? v.x : INFINITY, l1 = active ? v.y : INFINITY, l2 = active ? v.z : INFINITY;
    DUST_DPP_REDUCE("v_min_f32_dpp", l0, l1, l2);
    r.lo[0] = top(l0); r.lo[1] = top(l1); r.lo[2] = top(l2);
  }
  {
    float h0 = active ? v.x : -INFINITY, h1 = active ? v.y : -INFINITY, h2 = active ? v.z : -INFINITY;
    DUST_DPP_REDUCE("v_max_f32_dpp", h0, h1, h2);
    r.hi[0] = top(h0); r.hi[1] = top(h1); r.hi[2] = top(h2);
  }
  return r;
}
#undef DUST_DPP_REDUCE
#undef DUST_DPP_STEP
__device__ __forceinline__ Range3 point_range(V3 v) {  // every ray shares v (the camera position, the sun direction)
  Range3 r;
  r.lo[0] = r.hi[0] = v.x; r.lo[1] = r.hi[1] = v.y; r.lo[2] = r.hi[2] = v.z;
  return r;
}

// Tests all instance boxes, 64 at a time, against the bundle of rays {o in org, d in dir, 0 <= t <= tmax} and
// compacts the survivors into `cand`, sorted by the earliest time any ray of the bundle can enter them.
// Returns the number of survivors; a count above kMaxCand means "list overflowed, walk every instance".
// Per axis:  exists o, d:  lo <= o + d t <= hi   <=>   org.lo + dir.lo t <= hi  and  org.hi + dir.hi t >= lo   (t >= 0).
// The interval ends are the same in every lane, so each case split below is a select on precomputed per-packet
// values (reciprocals included): no division and no branch inside the loop.
template <int MODE>
__device__ uint32_t cull_instances(ArgsRef a, bool any_active, const Range3& org, const Range3& dir, float tmax, uint32_t* cand) {
  PROF_ENTER(P_CULL);
  if (!any_active) { PROF_LEAVE(P_CULL); return 0; }
  // r1/r2: reciprocal of the direction interval's ends (0 where the end is 0); s*: which bound the quotient feeds
  float r1[3], r2[3];
  bool up1[3], lo1[3], z1[3], lo2[3], up2[3], z2[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float d1 = dir.lo[k], d2 = dir.hi[k];
    z1[k] = d1 == 0.0f; up1[k] = d1 > 0.0f; lo1[k] = d1 < 0.0f;
    z2[k] = d2 == 0.0f; lo2[k] = d2 > 0.0f; up2[k] = d2 < 0.0f;
    r1[k] = z1[k] ? 0.0f : __builtin_amdgcn_rcpf(d1);  // 1 ulp: the interval ends carry 1e-5 of slack, and the stored
    r2[k] = z2[k] ? 0.0f : __builtin_amdgcn_rcpf(d2);  // entry time is rounded down by 2^-7
  }
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t n_inst = a.n_instances;
  const bool in_lds = a.n_lds_boxes != 0;
  const f32x4* lbox = reinterpret_cast<const f32x4*>(g_lds + a.n_lds_models * kN16LdsBytes + (blockDim.x >> 6) * (kMaxCand * 8u + 8u) + 16u);
  uint32_t n = 0;
  // one box against the bundle: passes, and the earliest time any ray of the bundle can enter it
  auto test = [&](f32x4 blo, f32x4 bhi, bool valid, float& t_lo) {
    const float wlo[3] = {blo.x, blo.y, blo.z}, whi[3] = {bhi.x, bhi.y, bhi.z};
    float t_hi = tmax;
    t_lo = 0.0f;
    bool pass = valid;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const float c1 = whi[k] - org.lo[k], c2 = wlo[k] - org.hi[k];
      const float q1 = c1 * r1[k], q2 = c2 * r2[k];
      t_hi = fminf(t_hi, fminf(up1[k] ? q1 : INFINITY, up2[k] ? q2 : INFINITY));
      t_lo = fmaxf(t_lo, fmaxf(lo1[k] ? q1 : 0.0f, lo2[k] ? q2 : 0.0f));
      pass = pass & !(z1[k] & (c1 < 0.0f)) & !(z2[k] & (c2 > 0.0f));
    }
    return pass & !(t_lo > t_hi * (1.0f + 1e-5f) + 1e-4f);
  };
  // one word per candidate: earliest entry of any ray of the packet (upper 16 bits of the float, i.e. rounded DOWN: stays
  // conservative, and absorbs the reciprocal's rounding) above the 16-bit instance id -- unsigned compare orders by entry
  // time, then id
  auto append = [&](bool pass, float t_lo, uint32_t id) {
    const uint64_t bal = __ballot(pass);
    if (pass) {
      const uint32_t pos = n + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
      if (pos < kMaxCand) cand[pos] = (__float_as_uint(t_lo) & 0xFFFF0000u) | (id & 0xFFFFu);
    }
    n += (uint32_t)__popcll(bal);
  };
  if (!LARGE) {  // a few hundred instances at most: every box, 64 at a time
    for (uint32_t base = 0; base < n_inst; base += 64) {
      const uint32_t i = base + lane;
      const uint32_t ic = i < n_inst ? i : n_inst - 1u;  // clamp instead of branching around the loads
      f32x4 blo, bhi;
      if (in_lds) { blo = lbox[ic * 2u]; bhi = lbox[ic * 2u + 1u]; }
      else { blo = *(DUST_RO(f32x4))(&a.boxes[ic].lo[0]); bhi = *(DUST_RO(f32x4))(&a.boxes[ic].hi[0]); }
      const float wlo[3] = {blo.x, blo.y, blo.z}, whi[3] = {bhi.x, bhi.y, bhi.z};
      float t_lo = 0.0f, t_hi = tmax;
      bool pass = i < n_inst;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float c1 = whi[k] - org.lo[k], c2 = wlo[k] - org.hi[k];
        const float q1 = c1 * r1[k], q2 = c2 * r2[k];
        t_hi = fminf(t_hi, fminf(up1[k] ? q1 : INFINITY, up2[k] ? q2 : INFINITY));
        t_lo = fmaxf(t_lo, fmaxf(lo1[k] ? q1 : 0.0f, lo2[k] ? q2 : 0.0f));
        pass = pass & !(z1[k] & (c1 < 0.0f)) & !(z2[k] & (c2 > 0.0f));
      }
      pass = pass & !(t_lo > t_hi * (1.0f + 1e-5f) + 1e-4f);
      const uint64_t bal = __ballot(pass);
      if (pass) {
        const uint32_t pos = n + (uint32_t)__popcll(bal & ((1ull << lane) - 1ull));
        if (pos < kMaxCand) cand[pos] = (__float_as_uint(t_lo) & 0xFFFF0000u) | (i & 0xFFFFu);
      }
      n += (uint32_t)__popcll(bal);
    }
  } else {
    // Thousands of instances (the reference's TLAS, accel_struct/tlas.rs:79-117, holds one entry per entity): a 64-wide hierarchy.
    // dust_hip_scene_commit orders the instances along a space-filling curve and boxes every 64 consecutive ones (a GROUP); the
    // bundle is tested against the group boxes, 64 at a time, and only the groups it meets have their 64 instance boxes looked at
    // (one coalesced 2 KB read each: a slot's record carries the instance's id). The groups that passed are also left as a bit
    // mask in the list's sort staging area: if the list overflows, trace_ray walks those groups instead of every instance.
    for (uint32_t gbase = 0; gbase < a.n_groups; gbase += 64) {
      const uint32_t g = gbase + lane;
      const uint32_t gc = g < a.n_groups ? g : a.n_groups - 1u;
      f32x4 glo, ghi;
      if (in_lds) { glo = lbox[gc * 2u]; ghi = lbox[gc * 2u + 1u]; }
      else { glo = *(DUST_RO(f32x4))(&a.gboxes[gc].lo[0]); ghi = *(DUST_RO(f32x4))(&a.gboxes[gc].hi[0]); }
      float t_g;
      uint64_t groups = __ballot(test(glo, ghi, g < a.n_groups, t_g));
      if (lane == 0) { cand[kMaxCand + (gbase >> 5)] = (uint32_t)groups; cand[kMaxCand + (gbase >> 5) + 1u] = (uint32_t)(groups >> 32); }
      // (Round 6 requested the NEXT group's boxes before testing the current ones -- a software-pipelined loop, one memory round trip hidden per
      //  group --: the castle + 4 000 props 0.4352 / 0.4352 -> 0.4423 / 0.4452 ms. The loop is not waiting for these loads; the extra registers cost more.)
      while (groups != 0ull) {
        const uint32_t gi = gbase + (uint32_t)__builtin_ctzll(groups);
        groups &= groups - 1ull;
        const uint32_t slot = gi * 64u + lane;
        const uint32_t sc = slot < n_inst ? slot : n_inst - 1u;
        const f32x4 blo = *(DUST_RO(f32x4))(&a.sboxes[sc].lo[0]), bhi = *(DUST_RO(f32x4))(&a.sboxes[sc].hi[0]);
        float t_lo;
        const bool pass = test(blo, bhi, slot < n_inst, t_lo);
        append(pass, t_lo, __float_as_uint(blo.w));
      }
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  // Front-to-back order: rank the survivors by the packet's earliest entry time (ties by instance id), so that rays
  // which hit a near instance skip the far ones (te > best.t). Results do not depend on the order (deterministic
  // tie-break in test_brick); only the amount of work does.
  if (n > 1 && n <= kMaxCand) {  // front to back: rank = number of smaller keys (keys are unique)
    for (uint32_t base = 0; base < n; base += 64u) {
      const uint32_t me = base + lane;
      const uint32_t c = me < n ? cand[me] : 0xFFFFFFFFu;
      uint32_t rank = 0;
      for (uint32_t k = 0; k < n; ++k) rank += cand[k] < c ? 1u : 0u;  // LDS broadcast reads
      if (me < n) cand[kMaxCand + rank] = c;  // sorted copy staged behind the list (later rounds still read the original)
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (uint32_t i = lane; i < n; i += 64u) cand[i] = cand[kMaxCand + i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  PROF_LEAVE(P_CULL);
  return n;
}

// any_hit: gl_RayFlagsTerminateOnFirstHitEXT | SkipClosestHitShader (the sun shadow rays)
template <int RT, int MODE>
__device__ void trace_ray(ArgsRef a_in, bool active, V3 o, V3 d, float tmin, float tmax, bool any_hit,
                          const uint32_t* cand, uint32_t ncand, Hit& best, LaneStats& st) {
  PROF_ENTER(P_TRACE_RAY);
  best.found = false;
  best.t = tmax; best.inst = 0; best.block = 0; best.voxel = 0;
  if (COUNT && active) st.rays += 1;
  ncand = (uint32_t)__builtin_amdgcn_readfirstlane((int)ncand);  // popcounts of ballots: uniform, but only we know
  PROF_COUNT(P_N_TRACES, 1);
  PROF_COUNT(P_N_CAND, ncand);
  const bool all = ncand > kMaxCand || (a_in.debug & 4u);  // debug bit 4: ignore the list, walk every instance in index order
  const bool by_groups = LARGE && ncand > kMaxCand && !(a_in.debug & 4u);
  const uint32_t n = all ? a_in.n_instances : ncand;
  // world-space reciprocals feed only the conservative box tests below (1e-5 slack): v_rcp_f32's 1 ulp is enough.
  // (trace_instance keeps IEEE divisions: its reciprocals are the intersection shader's `1.0 / dir`.)
  const V3 inv_d = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
  const bool zero_axis = __any(active && (d.x == 0.0f || d.y == 0.0f || d.z == 0.0f));  // e.g. the default sun (x == 0)
  // when the ray leaves the union of all instance boxes (a ray that hits nothing never "settles" on a hit: this is what
  // lets a packet of sky-bound rays stop walking the candidate list)
  float t_scene = INFINITY;
  if (RT >= 2 && n > 8u) {  // the incoherent ray types; coherent packets have lists of one or two
    float te_s, tx_s;
    const bool in = slab_box(o, d, inv_d, a_in.world_min, a_in.world_max, te_s, tx_s);
    t_scene = in ? tx_s * (1.0f + 1e-5f) + 1e-3f : -1.0f;
  }
  if (RT >= 2 && !all && !(a_in.debug & 8u)) {
    // Incoherent packets (gather and surfel rays). The rays of such a packet spread over several instances, and a
    // wave-uniform walk (below) leaves most lanes idle in each visit. Here the list is taken 32 candidates at a time:
    // a uniform scan (scalar box loads, one slab test per ray and candidate) leaves every lane with the bit mask of the
    // boxes ITS ray meets, then each lane pops its own bits front to back and the wave traverses up to 64 different
    // instances at once -- instance and model records come through vector loads there.
    for (uint32_t base = 0; base < n; base += 32u) {
      const uint32_t cnt = n - base < 32u ? n - base : 32u;
      ArgsRef a = reload_args(a_in);
      uint64_t need;  // the rays this batch can still matter to
      {
        const uint32_t c0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)cand[base]);
        const float t_lo = __uint_as_float(c0 & 0xFFFF0000u) * (1.0f - 1e-5f) - 1e-4f;
        const bool settled = !active || (best.found && (any_hit || best.t < t_lo)) || t_scene < t_lo;
        need = __ballot(!settled);
        if (need == 0ull) break;  // sorted by earliest entry: no later candidate matters either
      }
      uint32_t mask = 0;
#if DUST_TRANSPOSE_RAYS > 0
      // The scan below costs one trip per CANDIDATE whatever the number of rays that still want the batch -- and from the second batch on most
      // rays of a packet are settled (a whole-sphere packet of cosine rays keeps 128 of the castle's 157 boxes: four batches, 121 trips for
      // 3.7 visits per ray; a sun item has a third of its lanes active to begin with). With few rays left the scan is turned round: a lane
      // holds a BOX (lane k and lane 32 + k: candidate k of the batch), the rays are taken two at a time -- lanes 0-31 test the first, lanes
      // 32-63 the second: six cross-lane reads bring the ray over --, and one ballot is the two rays' masks. One trip per two rays.
      // (settled rays get no mask: the pop loop below would drop theirs at its first candidate anyway)
      if ((uint32_t)__popcll(need) <= (uint32_t)DUST_TRANSPOSE_RAYS) {
        PROF_COUNT(P_N_CAND_ITER, (__popcll(need) + 1) >> 1);
        const uint32_t lane = threadIdx.x & 63u, k = lane & 31u, half = lane >> 5;
        const bool have = k < cnt;
        const uint32_t ii = cand[base + (have ? k : 0u)] & 0xFFFFu;
        f32x4 blo, bhi;
        if (!LARGE && a.n_lds_boxes != 0) {  // (a large scene's LDS holds its group boxes)
          const f32x4* lbox = reinterpret_cast<const f32x4*>(g_lds + a.n_lds_models * kN16LdsBytes + (blockDim.x >> 6) * (kMaxCand * 8u + 8u) + 16u);
          blo = lbox[ii * 2u]; bhi = lbox[ii * 2u + 1u];
        } else { blo = *(DUST_RO(f32x4))(&a.boxes[ii].lo[0]); bhi = *(DUST_RO(f32x4))(&a.boxes[ii].hi[0]); }
        const float lo[3] = {blo.x, blo.y, blo.z}, hi[3] = {bhi.x, bhi.y, bhi.z};
        for (uint64_t rest = need; rest != 0ull;) {
          const uint32_t ra = (uint32_t)__builtin_ctzll(rest);
          rest &= rest - 1ull;
          const bool two = rest != 0ull;
          const uint32_t rb = two ? (uint32_t)__builtin_ctzll(rest) : ra;
          rest &= rest - 1ull;   // (0 & anything = 0 when there was no second ray)
          const int src = (int)(half ? rb : ra);
          const V3 ro = mk(__shfl(o.x, src), __shfl(o.y, src), __shfl(o.z, src));
          const V3 ri = mk(__shfl(inv_d.x, src), __shfl(inv_d.y, src), __shfl(inv_d.z, src));
          float te, tx;
          bool box;
          if (zero_axis) box = slab_box(ro, mk(__shfl(d.x, src), __shfl(d.y, src), __shfl(d.z, src)), ri, lo, hi, te, tx);
          else box = slab_box_nonzero(ro, ri, lo, hi, te, tx);
          const uint64_t bal = __ballot(box && have && (half == 0u || two));
          if (lane == ra) mask = (uint32_t)bal;
          if (two && lane == rb) mask = (uint32_t)(bal >> 32);
        }
      } else
#endif
      for (uint32_t k = 0; k < cnt; ++k) {
        PROF_COUNT(P_N_CAND_ITER, 1);
        const uint32_t ii = (uint32_t)__builtin_amdgcn_readfirstlane((int)cand[base + k]) & 0xFFFFu;
        const DUST_CONST_AS DevBox& bx = a.boxes[ii];
        float lo[3], hi[3];
#pragma unroll
        for (int q = 0; q < 3; ++q) { lo[q] = bx.lo[q]; hi[q] = bx.hi[q]; }
        float te, tx;
        const bool box = zero_axis ? slab_box(o, d, inv_d, lo, hi, te, tx) : slab_box_nonzero(o, inv_d, lo, hi, te, tx);
        mask |= box ? 1u << k : 0u;
      }
      if (!active) mask = 0;
      while (__any(mask != 0)) {
        uint32_t mine = 0xFFFFFFFFu;
        if (mask != 0) {
          const uint32_t c = cand[base + (uint32_t)__builtin_ctz(mask)];
          mask &= mask - 1u;
          const float t_lo = __uint_as_float(c & 0xFFFF0000u) * (1.0f - 1e-5f) - 1e-4f;
          if ((best.found && (any_hit || best.t < t_lo)) || t_scene < t_lo) mask = 0;
          else mine = c & 0xFFFFu;
        }
        PROF_COUNT(P_N_VISITS, 1);
        if (mine != 0xFFFFFFFFu) {
          if (COUNT) st.instances_tested += 1;
          const DUST_CONST_AS DevVisit& v = a.visits[mine];
          PROF_ENTER(P_INSTANCE);
          trace_instance<RT, MODE>(v.m, mine, xform_point(v.w2o, o), xform_dir(v.w2o, d), tmin, tmax, any_hit, best, st);
          PROF_LEAVE(P_INSTANCE);
        }
      }
    }
    if (COUNT && best.found) st.hits += 1;
    PROF_LEAVE(P_TRACE_RAY);
    return;
  }
  for (uint32_t ci = 0; ci < n; ++ci) {  // wave-uniform loop
    ArgsRef a = reload_args(a_in);  // per candidate: the table pointers are loaded again rather than carried through the visit
    uint32_t ii;
    float lo[3], hi[3];
    if (all) {  // more instances than the list holds: walk every instance box in index order
      ii = ci;
      if (by_groups) {  // ... a large scene's in slot order, the groups the bundle's cull let through only (their mask sits behind the list)
        if ((ci & 63u) == 0u) {
          const uint32_t gbits = (uint32_t)__builtin_amdgcn_readfirstlane((int)cand[kMaxCand + (ci >> 11)]);
          if (!((gbits >> ((ci >> 6) & 31u)) & 1u)) { ci += 63u; continue; }
        }
        ii = __float_as_uint(a.sboxes[ci].pad0);
      }
    } else {
      const uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cand[ci]);  // same address in every lane: LDS broadcast
      ii = c & 0xFFFFu;
      // the list is sorted by earliest possible entry: once every ray of the packet has a hit in front of this
      // candidate's earliest entry, no later candidate can matter either
      const float t_lo = __uint_as_float(c & 0xFFFF0000u);
      const bool settled = !active || (best.found && (any_hit || best.t < t_lo * (1.0f - 1e-5f) - 1e-4f)) ||
                           t_scene < t_lo * (1.0f - 1e-5f) - 1e-4f;
      if (__all(settled)) break;
    }
    ii = (uint32_t)__builtin_amdgcn_readfirstlane((int)ii);  // wave-uniform by construction: say so, so that the
    // box, transform and model come through scalar loads of ONE record, all issued here: the object-space ray is worked
    // out before the box test decides whether any lane needs it (18 operations, pinned below so that they are not sunk
    // behind the branch again), which puts the three loads in flight together instead of one round trip after another
    const DUST_CONST_AS DevVisit& v = a.visits[ii];
#pragma unroll
    for (int k = 0; k < 3; ++k) { lo[k] = v.lo[k]; hi[k] = v.hi[k]; }
    const V3 oo = xform_point(v.w2o, o), od = xform_dir(v.w2o, d);
    const float b0 = v.m.bmin[0], b1 = v.m.bmin[1], b2 = v.m.bmin[2], b3 = v.m.bmax[0], b4 = v.m.bmax[1], b5 = v.m.bmax[2];
    bool go = active && !(any_hit && best.found);
    float te, tx;
    bool box;
    if (zero_axis) box = slab_box(o, d, inv_d, lo, hi, te, tx);
    else box = slab_box_nonzero(o, inv_d, lo, hi, te, tx);
    const float limit = best.found ? best.t : tmax;
    go = go & box & !(te * (1.0f - 2e-6f) > limit);
    asm volatile("" ::"v"(oo.x), "v"(oo.y), "v"(oo.z), "v"(od.x), "v"(od.y), "v"(od.z), "s"(b0), "s"(b1), "s"(b2), "s"(b3), "s"(b4), "s"(b5));
    PROF_COUNT(P_N_CAND_ITER, 1);
    if (!__any(go)) continue;
    PROF_COUNT(P_N_VISITS, 1);
    if (go) {
      if (COUNT) st.instances_tested += 1;
      PROF_ENTER(P_INSTANCE);
      trace_instance<RT, MODE>(v.m, ii, oo, od, tmin, tmax, any_hit, best, st);
      PROF_LEAVE(P_INSTANCE);
    }
  }
  if (COUNT && best.found) st.hits += 1;
  PROF_LEAVE(P_TRACE_RAY);
}

// ------------------------------------------------------------------ one ray per lane: a visit as a state a lane carries (k_ray_walk, gi.hip)
// walk_begin + walk_step are trace_instance's prologue and loop body, verbatim: what a ray computes is what trace_ray / trace_instance
// compute for it; only which rays share a wavefront when changes -- never a result.
struct WalkState {          // one lane's visit of one instance
  V3 o, d, inv;             // object-space ray, inv = 1 / d (IEEE division: the intersection shader's reciprocal)
  float t, tx_stop, near_tol;
  int ijk[3];
  uint32_t stepped, cl_main, steps;
  bool screen;
  bool prev_whole;          // DEEP: the cell the walk has just left was a whole 16-cell (or larger) with nothing untested in it (trace_instance)
  MidCache mc;
  uint32_t inst;
  // what a step of a two-level model reads of its record, taken along when the visit starts (a lane's model is its own: read
  // from the record at every step, these would be vector loads in front of the step's one dependent access)
  int32_t lds_slot;
  uint32_t extent;
  DUST_RO(uint8_t) root;
  DUST_RO(uint64_t) dense_mask;
};
struct ModelLite {  // the two-level part of a DevModel, as find_brick reads it
  DUST_RO(uint8_t) root;
  DUST_RO(uint64_t) dense_mask;
  int32_t lds_slot;
  static constexpr uint32_t n_levels = 2;
  DUST_RO(uint8_t) l2;
  DUST_RO(DevL2Cell) l2_cells;
};
// trace_instance's prologue: false when the ray misses the model's bounds
// (Model: a DevModel record, or any view with its bmin / bmax / lds_slot / extent / root / dense_mask members: EnterView)
struct EnterView {
  float bmin[3], bmax[3];
  int32_t lds_slot;
  uint32_t extent;
  DUST_RO(uint8_t) root;
  DUST_RO(uint64_t) dense_mask;
};
template <int RT, int MODE, class Model>
__device__ __forceinline__ bool walk_begin(WalkState& w, const Model& m, uint32_t inst, V3 o, V3 d, float tmin) {
  w.o = o; w.d = d; w.inst = inst;
  w.lds_slot = m.lds_slot; w.extent = m.extent; w.root = m.root; w.dense_mask = m.dense_mask;
  w.inv = mk(1.0f / d.x, 1.0f / d.y, 1.0f / d.z);
  float te, tx;
  if (!slab_box(o, d, w.inv, m.bmin, m.bmax, te, tx)) return false;
  if (in_brick_plane<RT>(o, d)) return false;
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  float t = fmaxf(te, 0.0f);
  if (RT >= 2) t = fmaxf(t, tmin * (1.0f - 1e-6f));
  float reach = 0.0f;
  bool screen = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const float p = oo[a] + dd[a] * t;
    w.ijk[a] = f2i_clamp(dd[a] < 0.0f ? ceilf(p) - 1.0f : floorf(p), (int)m.bmin[a], (int)m.bmax[a] - 1);
    reach = fmaxf(reach, fabsf(oo[a]) + fmaxf(fabsf(p), fabsf(oo[a] + dd[a] * tx)));
  }
  w.near_tol = 3.0e-7f * (reach + 16.0f);
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    const int b0 = w.ijk[a] & ~3, blo = (int)m.bmin[a], bhi = (int)m.bmax[a] - 1;
    const float q = (oo[a] + dd[a] * t) - (float)b0;
    screen = screen | ((q <= 4.0f * w.near_tol) & (b0 - 1 >= blo)) | ((q >= 4.0f - 4.0f * w.near_tol) & (b0 + 4 <= bhi));
  }
  w.t = t; w.screen = screen; w.stepped = 0; w.cl_main = 2; w.steps = 0; w.prev_whole = false;
  midcache_reset(w.mc);
  w.tx_stop = tx * (1.0f + 1e-5f) + 1e-5f;
  return true;
}
// trace_instance's loop body: one cell. Returns true when the visit is over.
template <int RT, int MODE>
__device__ __forceinline__ bool walk_step(WalkState& w, const DUST_CONST_AS DevModel* mp, float tmin, float tmax, bool any_hit, Hit& best, LaneStats& st) {
  {
    const float limit = best.found ? best.t : tmax;
    if (w.t * (1.0f - 2e-6f) > limit) return true;
    if (any_hit && best.found) return true;
    if (++w.steps > 200000u) return true;
  }
  const float oo[3] = {w.o.x, w.o.y, w.o.z}, dd[3] = {w.d.x, w.d.y, w.d.z}, inv[3] = {w.inv.x, w.inv.y, w.inv.z};
  const int E = (int)w.extent;
  uint32_t key;
  uint64_t mask;
  if (DEEP) {
    mask = find_brick<MODE>(*mp, w.ijk[0], w.ijk[1], w.ijk[2], w.cl_main, key, w.mc, st, true, w.o, w.d, w.inv, kWholeCells<RT>);
    if (w.cl_main & kDirectCell) {  // (key, mask) = the cell's mid node and child mask
      w.cl_main = 4;
      test_cell_bricks<RT, MODE>(*mp, w.inst, key, mask, w.ijk[0], w.ijk[1], w.ijk[2], w.o, w.d, w.inv, tmin, tmax, best, st, true);
      mask = 0;
    }
  } else {
    ModelLite lm;
    lm.root = w.root; lm.dense_mask = w.dense_mask; lm.lds_slot = w.lds_slot; lm.l2 = nullptr; lm.l2_cells = nullptr;
    mask = find_brick<MODE>(lm, w.ijk[0], w.ijk[1], w.ijk[2], w.cl_main, key, w.mc, st, true, w.o, w.d, w.inv);
  }
  // (trace_instance's refinement of `screen` for a whole 16-cell with nothing untested in it, verbatim: only a brick ACROSS the entered face
  //  can need the neighbour visit -- without it the camera, sun and AO rays of a 4096^3 tree, which take sparse cells whole, made the call on
  //  most of the trips whose entry point lies near a brick plane)
  if (DEEP && __builtin_expect(w.screen, 0) && w.cl_main >= 4u) {
    bool near16 = false, across_needed = true, sided = true;
    uint32_t near4 = 0;
    int c[3] = {w.ijk[0], w.ijk[1], w.ijk[2]};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
      if (w.stepped & (1u << a)) {
        c[a] = dd[a] > 0.0f ? (w.ijk[a] & ~15) - 1 : (w.ijk[a] & ~15) + 16;  // back across the face
      } else {
        const float pa = oo[a] + dd[a] * w.t;
        const float r4 = pa * 0.25f;
        if (fabsf(r4 - rintf(r4)) <= w.near_tol) {
          near4 += 1u;
          const int b0 = w.ijk[a] & ~3;
          const float q = pa - (float)b0;
          if (q <= 8.0f * w.near_tol) { c[a] = b0 - 1; near16 = near16 | ((b0 & 15) == 0); }
          else if (q >= 4.0f - 8.0f * w.near_tol) { c[a] = b0 + 4; near16 = near16 | (((b0 + 4) & 15) == 0); }
          else sided = false;
        }
      }
    }
    if (w.stepped == 0u || near4 == 0u) across_needed = false;
    else if (__popc(w.stepped) == 1 && near4 == 1u && sided) {
      const int kd = ((c[0] >> 4) << 16) | ((c[1] >> 4) << 8) | (c[2] >> 4);
      const uint32_t bd = ((uint32_t)((c[0] >> 2) & 3) << 4) | ((uint32_t)((c[1] >> 2) & 3) << 2) | (uint32_t)((c[2] >> 2) & 3);
      if (w.prev_whole || (kd == w.mc.key && !((w.mc.mask4 >> bd) & 1ull))) across_needed = false;
    }
    w.screen = (__popc(w.stepped) > 1) | near16 | !sided | across_needed;
  }
  const int S = 1 << w.cl_main;
  float ta[3], tn = INFINITY;
  int cc[3];
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    cc[a] = w.ijk[a] & ~(S - 1);
    if (dd[a] != 0.0f) {
      const float plane = (float)(dd[a] > 0.0f ? cc[a] + S : cc[a]);
      ta[a] = (plane - oo[a]) * inv[a];
    } else {
      ta[a] = INFINITY;
    }
    tn = fminf(tn, ta[a]);
  }
  const bool stuck = !(tn < INFINITY);
  uint32_t next_stepped = 0;
  int next_ijk[3];
  bool outside = false, next_screen = false;
#pragma unroll
  for (int a = 0; a < 3; ++a) {
    if (ta[a] == tn) {
      next_stepped |= 1u << a;
      next_ijk[a] = dd[a] > 0.0f ? cc[a] + S : cc[a] - 1;
      if (next_ijk[a] < 0 || next_ijk[a] >= E) outside = true;
    } else {
      const float p = oo[a] + dd[a] * tn;
      next_ijk[a] = f2i_clamp(floorf(p), cc[a], cc[a] + S - 1);
      const float r = p * 0.25f;
      next_screen = next_screen | (fabsf(r - rintf(r)) <= w.near_tol);
    }
  }
  next_screen = next_screen | (__popc(next_stepped) > 1);
  if (mask != 0) test_brick<RT, MODE>(mask, w.inst, key, w.ijk[0] & ~3, w.ijk[1] & ~3, w.ijk[2] & ~3, w.o, w.d, w.inv, tmin, tmax, best, st);
  if (__builtin_expect(w.screen, 0)) {
    const u32x8 nv = visit_neighbours<RT, MODE>(mp, w.inst, w.o.x, w.o.y, w.o.z, w.d.x, w.d.y, w.d.z, w.inv.x, w.inv.y, w.inv.z, tmin, tmax, w.t,
                                                 w.ijk[0], w.ijk[1], w.ijk[2], w.stepped, best.t, best.inst, best.block, best.voxel,
                                                 best.found ? 1u : 0u, w.mc.key, w.mc.mid, w.cl_main,
                                                 DEEP ? (uint32_t)w.mc.mask4 : 0u, DEEP ? (uint32_t)(w.mc.mask4 >> 32) : 0u);
    best.t = __uint_as_float(nv[0]); best.inst = nv[1]; best.block = nv[2]; best.voxel = nv[3]; best.found = nv[4] != 0;
    w.mc.key = (int)nv[5]; w.mc.mid = nv[6];
    if (COUNT) st.bricks_tested += nv[7];
  }
  if (stuck || outside) return true;
  if (DEEP) w.prev_whole = w.cl_main >= 4u;
  w.ijk[0] = next_ijk[0]; w.ijk[1] = next_ijk[1]; w.ijk[2] = next_ijk[2];
  w.stepped = next_stepped;
  w.screen = next_screen;
  w.t = fmaxf(w.t, tn);
  return w.t * (1.0f - 2e-6f) > w.tx_stop;
}

// ------------------------------------------------------------------ work distribution
struct Packet {
  uint32_t px, py;
  bool valid;
};

// Work distribution. The tile list is cut into 8 contiguous bands, one per XCD (block b runs on XCD b % 8, so a band
// stays in one L2). Each band's tiles are handed out in the band's ORDER (k_tile_order: most expensive first, by what the
// pass's previous launch measured; screen order when nothing was measured yet):
//   * a wave's first tile is DEALT, not grabbed (`static_rounds` = 1: position i of the band for its wave i -- no 512-deep
//     queue on the counter at kernel start);
//   * everything else comes from the band's atomic counter through the workgroup's LDS queue (below): whoever is done first
//     takes more. A workgroup drains its own band first, then helps the others. Each counter owns a 256-byte line (sharing
//     one line across XCDs serialised every grab: 0.77 ms -> 0.39 ms per pass when they were separated).
// Dealing MORE rounds (round r: position r W + i, back and forth over the cost-sorted band -- the classic longest-first deal, no
// atomic and no queue for three tiles in four) was built in round 3 and lost badly: 4 / 6 / 7 dealt rounds of the castle's 7.9
// per wave took 0.273 / 0.283 / 0.309 ms against 0.241 grabbed. Last frame's cost classes predict a tile to a quarter octave
// and say nothing about which waves will share a SIMD; the grab corrects both as it goes. DUST_HIP_STATIC_ROUNDS overrides.
// Measured and rejected for the grabbed part: one device atomic per tile and wave with the NEXT ticket requested before
// tracing the current packet (+7 %: returns are in order, so the first load of the packet waits for the atomic anyway),
// per-wave chunks of 2 tiles (+13 %), static striding without an order, 4 sub-queues per band (+2.5 %), queue batches of 8
// (even) and 16 (+5 %: tail).
struct WorkCursor {
  uint32_t round;       // dealt rounds taken so far
  uint32_t frame;       // k_primary_ao_batch: which frame of the launch the wave is handing itself tiles of (0 everywhere else: folded away)
  uint32_t grab;        // ... and how many tickets a refill of that frame's queue takes (everywhere else: kGrabBatch)
  uint32_t tries;       // ... and how many bands a workgroup tries before it gives the frame up (everywhere else: kRegions)
};
__device__ __forceinline__ uint32_t band_static_tickets(uint32_t band) {  // waves whose own band this is
  return ((gridDim.x + 7u - band) >> 3) * (blockDim.x >> 6);
}
__device__ __forceinline__ WorkCursor cursor_begin() {
  WorkCursor w;
  w.round = 0;
  w.frame = 0;
  w.grab = 0;   // (only a batched launch reads these two)
  w.tries = 0;
  return w;
}
// After the dealt rounds a workgroup's waves share a small queue in LDS: {next, end} in one 64-bit word, taken from
// with ds_add_rtn_u64. The wave that finds it exactly empty refills it with kGrabBatch consecutive tiles -- one device-scope
// atomic on the band's counter per batch instead of one per tile (a ~2 us round trip on which each wave used to spend 16 %
// of its time) -- and the others retry; neighbouring tiles run at the same time on the same CU. -1.5 % to -3 % per kernel.
#ifndef DUST_GRAB_BATCH
#define DUST_GRAB_BATCH 4
#endif
constexpr uint32_t kGrabBatch = DUST_GRAB_BATCH;
// A launch of several frames (k_primary_ao_batch) refills with more tickets at a time in every frame but its last: between two frames of a launch
// nothing waits for the frame's last tiles, so the larger batch's cost -- a coarser tail -- is not paid, and its gain -- a device atomic per 16
// tiles instead of per 4 -- is (8 frames per launch: 1.6815 ms with 4 everywhere, 1.6745 with 8, 1.6735 with 16; with 2: 1.7018; 16 and 4 in the
// last frame: 1.6653). Where a band holds only a round or two of tiles for its workgroups (a 1/8 row band: 510 tiles for 60 workgroups) 16 at a
// time would leave half of them without any: FrameArgs::batch_grab, worked out per frame by launch_primary_ao_batch.
constexpr uint32_t kBatchGrabMax = 16;
constexpr uint32_t kQueueDone = 0x80000000u;  // {end = 0, next >= kQueueDone}: no tiles left anywhere
__device__ __forceinline__ unsigned long long* block_queue(ArgsRef a) {  // behind the per-wave candidate lists, zeroed by stage_roots
  return reinterpret_cast<unsigned long long*>(g_lds + a.n_lds_models * kN16LdsBytes + (blockDim.x >> 6) * (kMaxCand * 8u));
}
__device__ __forceinline__ u32x4* lds_boxes(ArgsRef a) {  // behind the queue and the per-wave tile accounts (16-byte aligned: every part before it is)
  return reinterpret_cast<u32x4*>(g_lds + a.n_lds_models * kN16LdsBytes + (blockDim.x >> 6) * (kMaxCand * 8u + 8u) + 16u);
}
// A launch of several frames (k_primary_ao_batch) has a tile queue {next, end} + band_try PER FRAME: a wave that has seen frame f's queue done moves
// on while its neighbours are still on frame f's last tiles, so the two queues are live at once. Frame 0's is block_queue; the others' stand behind
// the boxes, at the very end of the launch's LDS (lds_bytes + 16 per further frame).
__device__ __forceinline__ unsigned long long* frame_queue(ArgsRef a, uint32_t frame) {
  if (frame == 0u) return block_queue(a);
  return reinterpret_cast<unsigned long long*>(g_lds + a.batch_queue_base) + (frame - 1u) * 2u;
}
// The tile a wave is working on and when it started, in the wave's LDS slot behind the workgroup's queue: next_packet closes
// the previous tile's account (cycles -> a.tile_cost) when the wave comes back for more. No register is carried for it.
__device__ __forceinline__ void account_tile(ArgsRef a, uint32_t next_tile) {
  if (!a.tile_cost) return;
  if ((threadIdx.x & 63u) == 0) {
    uint32_t* slot = reinterpret_cast<uint32_t*>(block_queue(a) + 2) + (threadIdx.x >> 6) * 2u;
    const uint32_t now = (uint32_t)__builtin_amdgcn_s_memtime(), prev = slot[0];
    if (prev != 0xFFFFFFFFu) a.tile_cost[prev] = now - slot[1];
    slot[0] = next_tile; slot[1] = now;
  }
}
// `ticket` = band * tiles_per_band + pos: position `pos` of the band's order
// first tile and tile count of band b: the launch's equal split, or the cost-balanced cuts that came with the order
__device__ __forceinline__ void band_range(ArgsRef a, uint32_t b, uint32_t& lo, uint32_t& n) {
  if (a.band_cuts) {
    lo = a.band_cuts[b];
    n = a.band_cuts[b + 1u] - lo;
  } else {
    const uint32_t total = a.tiles_x * a.tiles_y, per = a.tiles_per_band;
    lo = b * per < total ? b * per : total;
    n = total - lo < per ? total - lo : per;
  }
}
__device__ __forceinline__ void packet_of_tile(ArgsRef a, uint32_t ticket, uint32_t pos, uint32_t per, Packet& p) {
  const uint32_t lane = threadIdx.x & 63u;
  uint32_t tile = ticket;
  if (a.tile_order) {
    // With the tiles handed out longest first the launch is as long as its most expensive tile takes (it starts at once and
    // ends last: 545 k of the fused kernel's 570 k cycles on the castle), and that tile takes as long as it does because its
    // wave shares a SIMD with three others. The position in the band's order says how expensive the tile was last time:
    // the few at the front get the arbiter's priority, so the critical path runs at nearly a lone wave's speed while the
    // waves that give way have slack.
    // (round 4: without these priorities the kernel is 8-10 % slower; six other gradings and static per-slot priorities: no better)
    const uint32_t rank = pos < (per >> 5) ? 3u : (pos < (per >> 3) ? 2u : (pos < (per >> 1) ? 1u : 0u));
    const uint32_t prio = rank > a.prio_floor ? rank : a.prio_floor;
    if (prio == 3u) __builtin_amdgcn_s_setprio(3);
    else if (prio == 2u) __builtin_amdgcn_s_setprio(2);
    else if (prio == 1u) __builtin_amdgcn_s_setprio(1);
    else __builtin_amdgcn_s_setprio(0);
    if (lane == 0) g_tile_prio[threadIdx.x >> 6] = prio;
    tile = a.tile_order[ticket];  // ticket -> tile, most expensive tiles of the band first
  }
  else if (lane == 0) g_tile_prio[threadIdx.x >> 6] = 0u;  // (no order yet: every tile starts at 0 and earns what it needs)
  if (lane == 0) g_tile_start[threadIdx.x >> 6] = (uint32_t)__builtin_amdgcn_s_memtime();
  account_tile(a, tile);
  // tile / tiles_x: through the multiplier where that is exact (with_schedule); a one-row list of work items has quotient 0
  const uint32_t ty = a.tiles_x_magic ? __umulhi(tile, a.tiles_x_magic) : (a.tiles_y > 1u ? tile / a.tiles_x : 0u), tx = tile - ty * a.tiles_x;
  p.px = tx * kTileW + (lane % kTileW);
  p.py = a.row_begin + ty * kTileH + (lane / kTileW);
  p.valid = p.px < a.width && p.py < a.row_end;
}
template <bool BATCH>   // BATCH: a launch of several frames (k_primary_ao_batch) -- the queue is frame w.frame's; else the workgroup's one queue
__device__ __forceinline__ bool next_packet_of(ArgsRef a, WorkCursor& w, Packet& p) {
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t own = blockIdx.x & 7u;
  const uint32_t grab = BATCH ? w.grab : kGrabBatch;
  PROF_ENTER(P_GRAB);
  if (w.round < a.static_rounds) {  // a dealt tile
    const uint32_t W = band_static_tickets(own);
    const uint32_t i = (uint32_t)__builtin_amdgcn_readfirstlane((int)((blockIdx.x >> 3) * (blockDim.x >> 6) + (threadIdx.x >> 6)));
    const uint32_t r = w.round;
    w.round = r + 1u;
    const uint32_t pos = r * W + ((r & 1u) ? W - 1u - i : i);
    uint32_t blo, bn;
    band_range(a, own, blo, bn);
    if (pos < bn) { packet_of_tile(a, blo + pos, pos, bn, p); PROF_LEAVE(P_GRAB); return true; }
    w.round = a.static_rounds;  // (a band shorter than the deal: on to the queue, which is empty for it too)
  }
  unsigned long long* q = BATCH ? frame_queue(a, w.frame) : block_queue(a);
  volatile unsigned long long* qv = q;
  volatile uint32_t* band_try = reinterpret_cast<volatile uint32_t*>(q + 1);  // bands this workgroup has given up on
  for (;;) {
    unsigned long long old = 0;
    if (lane == 0) old = atomicAdd(q, 1ull);
    const uint32_t next = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)old);
    const uint32_t end = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(old >> 32));
    if (next < end) {  // (the queue holds tickets of ONE band: the high word's band)
      const uint32_t band = (own + (uint32_t)__builtin_amdgcn_readfirstlane((int)*band_try)) & 7u;
      uint32_t blo, bn;
      band_range(a, band, blo, bn);
      packet_of_tile(a, next, next - blo, bn, p);
      PROF_LEAVE(P_GRAB);
      return true;
    }
    if (end == 0u && next >= kQueueDone) { account_tile(a, 0xFFFFFFFFu); PROF_LEAVE(P_GRAB); return false; }
    if (next != end) { __builtin_amdgcn_s_sleep(4); continue; }  // another wave is refilling
    // exactly empty: this wave refills. Own band first, then the others' (a band stays in one XCD's L2 while it lasts).
    uint32_t bt = (uint32_t)__builtin_amdgcn_readfirstlane((int)*band_try);
    for (;;) {
      if (bt >= (BATCH ? w.tries : kRegions)) {
        if (lane == 0) *qv = (unsigned long long)kQueueDone;
        account_tile(a, 0xFFFFFFFFu);
        PROF_LEAVE(P_GRAB);
        return false;
      }
      const uint32_t band = (own + bt) & 7u;
      uint32_t blo, bn;
      band_range(a, band, blo, bn);
      uint32_t k = 0;
      if (lane == 0) k = a.static_rounds * band_static_tickets(band) + atomicAdd((uint32_t*)&a.work_counters[band * kCounterStride], grab);
      k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
      if (k < bn) {
        const uint32_t lo = blo + k, hi = blo + (k + grab < bn ? k + grab : bn);
        if (lane == 0) {
          *band_try = bt;
          *qv = ((unsigned long long)hi << 32) | (unsigned long long)(lo + 1u);  // one 8-byte LDS store: the batch goes live
        }
        packet_of_tile(a, lo, k, bn, p);
        PROF_LEAVE(P_GRAB);
        return true;
      }
      bt += 1;  // band exhausted
    }
  }
}

__device__ __forceinline__ bool next_packet(ArgsRef a, WorkCursor& w, Packet& p) { return next_packet_of<false>(a, w, p); }

__device__ __forceinline__ void prof_begin() {
#ifdef DUST_PROFILE
  if ((threadIdx.x & 63u) == 0)
    for (int i = 0; i < kProfBuckets; ++i) g_prof[threadIdx.x >> 6][i] = 0;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  PROF_ENTER(P_TOTAL);
  PROF_ENTER(P_STAGE);
#endif
}
__device__ __forceinline__ void prof_end() {
#ifdef DUST_PROFILE
  PROF_LEAVE(P_TOTAL);
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  if ((threadIdx.x & 63u) == 0)
    for (int i = 0; i < kProfBuckets; ++i) atomicAdd(&g_prof_out[i], g_prof[threadIdx.x >> 6][i]);
#endif
}
template <bool BATCH>
__device__ __forceinline__ void stage_roots_of(ArgsRef a) {
  prof_begin();
#ifdef DUST_TRACE_DEBUG
  if (threadIdx.x < 16) g_dbg_mask[threadIdx.x] = 0;
#endif
  if (blockIdx.x == 0 && threadIdx.x < kRegions) a.next_work_counters[threadIdx.x * kCounterStride] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a.started_word)  // (dust_dev.h: the frame's first launch tells the host it is running)
    __hip_atomic_store((uint32_t*)a.started_word, a.started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (threadIdx.x < 4) reinterpret_cast<uint32_t*>(block_queue(a))[threadIdx.x] = 0u;  // {next, end} = {0, 0}: empty; band_try = 0
  if (BATCH && a.batch_frames > 1u && threadIdx.x < 4u * (a.batch_frames - 1u)) reinterpret_cast<uint32_t*>(frame_queue(a, 1u))[threadIdx.x] = 0u;  // the further frames' queues
  if (threadIdx.x < (blockDim.x >> 6) * 2u) reinterpret_cast<uint32_t*>(block_queue(a) + 2)[threadIdx.x] = 0xFFFFFFFFu;  // per-wave tile accounts: none open
  // root masks + rank prefixes of the first n_lds_models models -> LDS: one coalesced 16 B-per-lane copy of the
  // scene's packed root table
  const uint32_t n16 = a.n_lds_models * (kN16LdsBytes / 16u);
  DUST_RO(u32x4) src = (DUST_RO(u32x4))a.root_table;
  {  // four 16-byte loads in flight per lane before the first is stored (this loop is the launch's first ~3 us: nothing but latency)
    u32x4* dst = reinterpret_cast<u32x4*>(g_lds);
    const uint32_t step = blockDim.x;
    uint32_t i = threadIdx.x;
    for (; i + 3u * step < n16; i += 4u * step) {
      const u32x4 v0 = src[i], v1 = src[i + step], v2 = src[i + 2u * step], v3 = src[i + 3u * step];
      dst[i] = v0; dst[i + step] = v1; dst[i + 2u * step] = v2; dst[i + 3u * step] = v3;
    }
    for (; i < n16; i += step) dst[i] = src[i];
  }
  {  // the instance boxes the packet cull streams through, when they fit as well
    DUST_RO(u32x4) bsrc = (DUST_RO(u32x4))(a.n_groups ? a.gboxes : a.boxes);  // (n_lds_boxes of them: instance boxes, or the group boxes of a large scene)
    u32x4* bdst = lds_boxes(a);
    for (uint32_t i = threadIdx.x; i < a.n_lds_boxes * 2u; i += blockDim.x) bdst[i] = bsrc[i];
  }
  __syncthreads();
  PROF_LEAVE(P_STAGE);
}
__device__ __forceinline__ void stage_roots(ArgsRef a) { stage_roots_of<false>(a); }
__device__ __forceinline__ uint32_t* wave_cand_list(ArgsRef a) {  // kMaxCand entries + kMaxCand of sort staging
  return reinterpret_cast<uint32_t*>(g_lds + a.n_lds_models * kN16LdsBytes) + (threadIdx.x >> 6) * (kMaxCand * 2);
}

__device__ __forceinline__ void add_stats(LaneStats& d, const LaneStats& s) {
  d.rays += s.rays; d.instances_tested += s.instances_tested; d.upper_descents += s.upper_descents;
  d.mid_descents += s.mid_descents; d.bricks_tested += s.bricks_tested; d.hits += s.hits;
}
template <int MODE>
__device__ __forceinline__ void flush_stats(ArgsRef a, int slot, const LaneStats& st) {
  if (!COUNT) return;
  atomicAdd((unsigned long long*)&a.stats[slot].rays, (unsigned long long)st.rays);
  atomicAdd((unsigned long long*)&a.stats[slot].instances_tested, (unsigned long long)st.instances_tested);
  atomicAdd((unsigned long long*)&a.stats[slot].upper_descents, (unsigned long long)st.upper_descents);
  atomicAdd((unsigned long long*)&a.stats[slot].mid_descents, (unsigned long long)st.mid_descents);
  atomicAdd((unsigned long long*)&a.stats[slot].bricks_tested, (unsigned long long)st.bricks_tested);
  atomicAdd((unsigned long long*)&a.stats[slot].hits, (unsigned long long)st.hits);
}

__device__ __forceinline__ V3 camera_ray_dir(ArgsRef a, uint32_t px, uint32_t py) {
  // camera.glsl:4-16. The divisions by the frame size go through the reciprocals the host put in the launch descriptor
  // (div_by: same quotients), and width / height is a launch constant.
  const DUST_CONST_AS DevCamera& c = a.cam;
  float nx = div_by((float)px + 0.5f, (float)a.width, a.inv_width, false), ny = div_by((float)py + 0.5f, (float)a.height, a.inv_height, false);
  float cx = 2.0f * nx - 1.0f, cy = 2.0f * ny - 1.0f;
  cy *= -1.0f;
  cx *= a.aspect;
  cx *= c.tan_half_fov; cy *= c.tan_half_fov;
  const float cz = -1.0f;
  return mk((c.col0[0] * cx + c.col1[0] * cy) + c.col2[0] * cz, (c.col0[1] * cx + c.col1[1] * cy) + c.col2[1] * cz,
            (c.col0[2] * cx + c.col1[2] * cy) + c.col2[2] * cz);
}

}  // namespace

// ==================================================================== spatial hash (headers/spatial_hash.glsl)
namespace {
__device__ __forceinline__ uint32_t pcg(uint32_t v) {  // spatial_hash.glsl:105-111
  uint32_t state = v * 747796405u + 2891336453u;
  uint32_t word = ((state >> ((state >> 28u) + 4u)) ^ state) * 277803737u;
  return (word >> 22u) ^ word;
}
__device__ __forceinline__ uint32_t xxhash32(uint32_t p) {  // spatial_hash.glsl:115-126
  uint32_t h = p + 374761393u;
  h = 668265263u * ((h << 17) | (h >> 15));
  h = 2246822519u * (h ^ (h >> 15));
  h = 3266489917u * (h ^ (h >> 13));
  return h ^ (h >> 16);
}
struct HashKey { int x, y, z; uint32_t dir; };
__device__ __forceinline__ uint32_t key_fingerprint(HashKey k) {  // spatial_hash.glsl:128-135
  uint32_t h = xxhash32((uint32_t)k.x);
  h = xxhash32((uint32_t)k.y + h);
  h = xxhash32((uint32_t)k.z + h);
  h = xxhash32(k.dir + h);
  return h > 1u ? h : 1u;
}
__device__ __forceinline__ uint32_t key_location(HashKey k, uint32_t capacity) {  // spatial_hash.glsl:136-142
  uint32_t h = pcg((uint32_t)k.x);
  h = pcg((uint32_t)k.y + h);
  h = pcg((uint32_t)k.z + h);
  h = pcg(k.dir + h);
  return h % capacity;
}
__device__ __forceinline__ V3 acescg_to_xyz(V3 v) {  // spatial_hash.glsl:12-19
  return mk((0.66245437f * v.x + 0.13400422f * v.y) + 0.15618773f * v.z, (0.2722288f * v.x + 0.6740818f * v.y) + 0.05368953f * v.z,
            (-0.0055746622f * v.x + 0.00406073f * v.y) + 1.0103393f * v.z);
}
__device__ uint32_t logluv_encode(V3 rgb) {  // spatial_hash.glsl:28-60
  const V3 XYZ = acescg_to_xyz(rgb);
  const float logY = 409.6f * (__builtin_amdgcn_logf(XYZ.y) + 20.0f);  // v_log_f32 (log2, 1 ulp): radiance, not geometry
  const float cl = gclamp(logY, 0.0f, 16383.0f);
  const uint32_t Le = (cl != cl) ? 0u : (uint32_t)cl;
  if (Le == 0) return 0;
  const float invDenom = 1.0f / ((-2.0f * XYZ.x + 12.0f * XYZ.y) + 3.0f * ((XYZ.x + XYZ.y) + XYZ.z));
  const float u = (4.0f * XYZ.x) * invDenom, v = (9.0f * XYZ.y) * invDenom;
  const float cu = gclamp(820.0f * u, 0.0f, 511.0f), cv = gclamp(820.0f * v, 0.0f, 511.0f);
  const uint32_t ue = (cu != cu) ? 0u : (uint32_t)cu, ve = (cv != cv) ? 0u : (uint32_t)cv;
  return (Le << 18) | (ue << 9) | ve;
}
__device__ V3 logluv_decode(uint32_t p) {  // spatial_hash.glsl:64-93
  const uint32_t Le = p >> 18;
  if (Le == 0) return mk(0, 0, 0);
  const float logY = div_const((float)Le + 0.5f, 409.6f) - 20.0f;
  const float Y = __builtin_amdgcn_exp2f(logY);  // v_exp_f32
  const float u = div_const((float)((p >> 9) & 0x1FFu) + 0.5f, 820.0f), v = div_const((float)(p & 0x1FFu) + 0.5f, 820.0f);
  const float invDenom = 1.0f / ((6.0f * u - 16.0f * v) + 12.0f);
  const float x = (9.0f * u) * invDenom, y = (4.0f * v) * invDenom;
  const float s = Y / y;
  const V3 r = xyz_to_acescg(mk(s * x, Y, s * ((1.0f - x) - y)));
  return mk(fmaxf(r.x, 0.0f), fmaxf(r.y, 0.0f), fmaxf(r.z, 0.0f));
}
// SpatialHashGet (spatial_hash.glsl:200-219): stamps last_accessed_frame of the entry it finds
// entry: 1 + index of the entry that was found (and stamped), 0 when there is none
__device__ bool hash_get(const DUST_CONST_AS DevGI& gi, HashKey key, uint32_t frame_index, V3& value, uint32_t& count, uint32_t& entry) {
  const uint32_t fp = key_fingerprint(key), loc = key_location(key, gi.hash_capacity);
  value = mk(0, 0, 0);
  count = 0;
  entry = 0;
  for (uint32_t i = 0; i < 3; ++i) {
    uint32_t* e = gi.hash + (size_t)(loc + i) * 3;
    const uint32_t cur = e[0];
    if (cur == 0) return false;
    if (cur == fp) {
      reinterpret_cast<uint16_t*>(e)[4] = (uint16_t)frame_index;  // every reader stores the same value
      value = logluv_decode(e[1]);
      count = e[2] >> 16;
      entry = loc + i + 1u;
      return true;
    }
  }
  return false;
}
// SpatialHashInsert (spatial_hash.glsl:147-195) over an accessor to the three entries of the probe window: the entries
// in memory (the shader's own form: the fingerprint is claimed with an atomic compare-and-swap), or a copy of the window
// held in registers (HashWindow: the deterministic apply runs a whole cluster of requests on it between one load and one
// store). One body, so both forms take the same decisions in the same order.
struct HashMemory {
  uint32_t* base;  // first entry of the window
  // the probe's radiance and meta words are requested WITH the compare-and-swap of its fingerprint, not behind it (three dependent round
  // trips to a random place of a 384 MB table per probe -> one). As racy as the shader's reads after its atomicCompSwap
  // (spatial_hash.glsl:147-195): another invocation may write the entry at any time either way.
  uint32_t rad_c, meta_c;
  __device__ __forceinline__ uint32_t claim(uint32_t i, uint32_t fp) {
    typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
    const u32x2_a4 rm = *reinterpret_cast<const volatile u32x2_a4*>(&base[i * 3 + 1]);
    const uint32_t old = atomicCAS(&base[i * 3], 0u, fp);
    rad_c = rm.x; meta_c = rm.y;
    return old;
  }
  __device__ __forceinline__ uint32_t meta(uint32_t) const { return meta_c; }
  __device__ __forceinline__ uint32_t radiance(uint32_t) const { return rad_c; }
  __device__ __forceinline__ void set(uint32_t i, uint32_t rad, uint32_t meta_) { base[i * 3 + 1] = rad; base[i * 3 + 2] = meta_; }
  __device__ __forceinline__ void set_fingerprint(uint32_t i, uint32_t fp) { base[i * 3] = fp; }
};
struct HashWindow {
  uint32_t w[9];
  __device__ __forceinline__ uint32_t claim(uint32_t i, uint32_t fp) {
    const uint32_t old = w[i * 3];
    if (old == 0u) w[i * 3] = fp;
    return old;
  }
  __device__ __forceinline__ uint32_t meta(uint32_t i) const { return w[i * 3 + 2]; }
  __device__ __forceinline__ uint32_t radiance(uint32_t i) const { return w[i * 3 + 1]; }
  __device__ __forceinline__ void set(uint32_t i, uint32_t rad, uint32_t meta_) { w[i * 3 + 1] = rad; w[i * 3 + 2] = meta_; }
  __device__ __forceinline__ void set_fingerprint(uint32_t i, uint32_t fp) { w[i * 3] = fp; }
};
template <class Entries>
__device__ __forceinline__ void hash_insert_window(Entries& e, uint32_t fp, V3 value, uint32_t frame_index) {
  uint32_t min_frame = 0;
  bool evict[3] = {true, false, false};  // which probe is the least recently accessed so far (the first of equals)
#pragma unroll
  for (uint32_t i = 0; i < 3; ++i) {
    const uint32_t cur = e.claim(i, fp);
    const uint32_t w2 = e.meta(i);
    const uint32_t cur_frame = w2 & 0xFFFFu;
    if (i == 0 || cur_frame < min_frame) {
      min_frame = cur_frame;
#pragma unroll
      for (uint32_t k = 0; k < 3; ++k) evict[k] = k == i;
    }
    if (cur == fp || cur == 0) {
      V3 rad = mk(0, 0, 0);
      uint32_t count = 0;
      if (cur == fp) { count = w2 >> 16; rad = logluv_decode(e.radiance(i)); }
      count = count < 403u ? count : 403u;
      const uint32_t next = count + 1;
      const float al = 1.0f / (float)next;
      const V3 out = mk(rad.x * (1.0f - al) + value.x * al, rad.y * (1.0f - al) + value.y * al, rad.z * (1.0f - al) + value.z * al);
      e.set(i, logluv_encode(out), (frame_index & 0xFFFFu) | (next << 16));
      return;
    }
  }
  const uint32_t rad = logluv_encode(value), meta = (frame_index & 0xFFFFu) | (1u << 16);
#pragma unroll
  for (uint32_t i = 0; i < 3; ++i)  // evict the least recently accessed of the three probes
    if (evict[i]) { e.set_fingerprint(i, fp); e.set(i, rad, meta); }
}
__device__ void hash_insert(const DUST_CONST_AS DevGI& gi, HashKey key, V3 value, uint32_t frame_index) {
  HashMemory m;
  m.base = gi.hash + (size_t)key_location(key, gi.hash_capacity) * 3;
  hash_insert_window(m, key_fingerprint(key), value, frame_index);
}
__device__ __forceinline__ float srgb_to_linear(float c) {  // color.glsl:1-5
  // pow(x, 2.4) as exp2(2.4 log2 x) on the hardware transcendentals: radiance (1e-3 tolerance), a fifth of the libm routine
  return c < 0.04045f ? div_const(c, 12.92f) : __builtin_amdgcn_exp2f(2.4f * __builtin_amdgcn_logf(div_const(fabsf(c + 0.055f), 1.055f)));
}
__device__ V3 modulate_by_avg_albedo(V3 r, uint32_t packed) {  // final_gather.rchit:68-80, surfel.rchit:60-71
  const V3 alb = mk(srgb_to_linear(div_const((float)((packed >> 22) & 1023u), 1023.0f)), srgb_to_linear(div_const((float)((packed >> 12) & 1023u), 1023.0f)),
                    srgb_to_linear(div_const((float)((packed >> 2) & 1023u), 1023.0f)));
  const V3 s = mk((1.7312546f * r.x + -0.6040432f * r.y) + -0.08010775f * r.z, (-0.131619f * r.x + 1.1348418f * r.y) + -0.008679431f * r.z,
                  (-0.024568284f * r.x + -0.12575036f * r.y) + 1.0656371f * r.z);  // ACEScg -> sRGB, color.glsl:16-23
  const V3 m = mk(s.x * alb.x, s.y * alb.y, s.z * alb.z);
  return mk((0.6031065f * m.x + 0.32633433f * m.y) + 0.047995567f * m.z, (0.07011794f * m.x + 0.9199162f * m.y) + 0.012763573f * m.z,
            (0.022178888f * m.x + 0.11607823f * m.y) + 0.94101846f * m.z);  // sRGB -> ACEScg, color.glsl:8-15
}
__device__ __forceinline__ uint32_t normal2faceid(V3 n) {  // normal.glsl:9-18
  const float s = gclamp((n.x + n.y) + n.z, 0.0f, 1.0f);
  return ((uint32_t)rintf(s) + (uint32_t)rintf(fabsf(n.z)) * 4u + (uint32_t)rintf(fabsf(n.y)) * 2u) & 0xFFu;
}
__device__ __forceinline__ V3 faceid2normal(uint32_t face) {  // normal.glsl:20-26
  const float s = (float)(face & 1u) * 2.0f - 1.0f;
  const uint32_t ax = (face & 0xFFu) >> 1;
  return mk(ax == 0 ? s : 0.0f, ax == 1 ? s : 0.0f, ax == 2 ? s : 0.0f);
}
// world-space surfel (brick centre + face) and hash key of a rough hit: final_gather.rchit:35-45, surfel.rchit:35-45
__device__ void brick_surfel(ArgsRef a, const Hit& h, V3 o, V3 d, HashKey& key, DevSurfel& sf, uint32_t& avg_albedo) {
  InstanceRef in = a.instances[h.inst];
  ModelRef m = a.visits[h.inst].m;  // (by the instance's index: not a round trip behind `in`, as a.models[in.model] is)
  const DustHipBlock b = load_block(m.blocks + resolve_block(m, h.block));
  const V3 ctr = mk((float)b.x + 2.0f, (float)b.y + 2.0f, (float)b.z + 2.0f);
  const V3 oo = xform_point(in.w2o, o), od = xform_dir(in.w2o, d);
  const V3 hpo = mk(h.t * od.x + oo.x, h.t * od.y + oo.y, h.t * od.z + oo.z);
  const V3 nw = cubed_normalize(xform_dir(in.o2w, mk(hpo.x - ctr.x, hpo.y - ctr.y, hpo.z - ctr.z)));
  const V3 cw = xform_point(in.o2w, ctr);
  const uint32_t face = normal2faceid(nw);
  key.x = f2i_trunc(cw.x / 4.0f); key.y = f2i_trunc(cw.y / 4.0f); key.z = f2i_trunc(cw.z / 4.0f);
  key.dir = face;
  sf.x = cw.x; sf.y = cw.y; sf.z = cw.z; sf.direction = face;
  avg_albedo = b.avg_albedo;
}
// ------------------------------------------------------------------ primary shading (hit.rchit:16-95 + miss.rmiss:7-17) for the lanes of one 8x8 pixel block
// Returns, per lane, what the later passes read back from the G-buffer: the hit distance (INFINITY on a miss) and the packed normal texel.
// store_illuminance: hit.rchit:57 zeroes img_illuminance; the fused kernel skips that store because the ambient occlusion pass
// overwrites the texel of every hit pixel anyway.
template <int MODE>
__device__ __forceinline__ void primary_shade(ArgsRef a, const Packet& p, V3 o, V3 d, const Hit& h, bool store_illuminance,
                                              float& hitT, uint32_t& normal_packed) {
  hitT = INFINITY;
  normal_packed = 0;
  if (!p.valid) return;
  const size_t pix = (size_t)p.py * a.width + p.px;
  if (!h.found) {
    const V3 dir = normalize3(d);
    const V3 s0 = sky_radiance(a.sky, dir), s1 = sun_radiance(a.sky, dir);
    store_radiance(a.g.denoised, pix, mk(div_const(s0.x + s1.x, 3.14f), div_const(s0.y + s1.y, 3.14f), div_const(s0.z + s1.z, 3.14f)), 100000.0f);
    DUST_NT_STORE(0xFFFFFFFFu, &a.g.albedo[pix]);
    DUST_NT_STORE(INFINITY, &a.g.depth[pix]);
    store_half4(a.g.motion, pix, 0.0f, 0.0f, 0.0f, 0.0f);
    return;
  }
  // Hit pixels. The instance and model records are read through SCALAR loads: the pixels of an 8x8 packet hit one or two
  // instances, so the hit lanes are taken one distinct instance at a time (the first remaining lane's, wave-uniform through
  // readfirstlane) -- 40 matrix floats and the model's pointers arrive in SGPRs instead of 50 VGPRs per lane.
  bool todo = true;
  while (todo) {
    const uint32_t cur = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.inst);
    if (h.inst != cur) continue;
    todo = false;
    InstanceRef in = a.instances[cur];
    ModelRef m = a.visits[cur].m;  // the instance's model record by the instance's index (a.models[in.model] is a round trip behind `in`)
    const uint32_t block = resolve_block(m, h.block);
    const DustHipBlock b = load_block(m.blocks + block);
    const V3 oo = xform_point(in.w2o, o), od = xform_dir(in.w2o, d);
    const V3 hpo = mk(h.t * od.x + oo.x, h.t * od.y + oo.y, h.t * od.z + oo.z);
    const V3 off = mk((float)(h.voxel >> 4), (float)((h.voxel >> 2) & 3u), (float)(h.voxel & 3u));
    const V3 ctr = mk(((float)b.x + off.x) + 0.5f, ((float)b.y + off.y) + 0.5f, ((float)b.z + off.z) + 0.5f);
    const V3 no = cubed_normalize(mk(hpo.x - ctr.x, hpo.y - ctr.y, hpo.z - ctr.z));
    const V3 nw = xform_dir(in.o2w, no);
    if (store_illuminance) store_half4(a.g.illuminance, pix, 0.0f, 0.0f, 0.0f, 0.0f);
    const uint32_t m1 = (uint32_t)b.mask, m2 = (uint32_t)(b.mask >> 32);
    const uint32_t ma = h.voxel < 32u ? (m1 & ((1u << (h.voxel & 31u)) - 1u)) : m1;
    const uint32_t mb = h.voxel >= 32u ? (m2 & ((1u << ((h.voxel - 32u) & 31u)) - 1u)) : 0u;
    const uint32_t voff = (uint32_t)__popc(ma) + (uint32_t)__popc(mb);
    const uint32_t pal = m.materials[b.material_ptr + voff];
    const uint32_t col = m.palette[pal];
    DUST_NT_STORE(pack_rgb10a2(div_const((float)(col & 255u), 255.0f), div_const((float)((col >> 8) & 255u), 255.0f),
                                             div_const((float)((col >> 16) & 255u), 255.0f), 1.0f), &a.g.albedo[pix]);
    DUST_NT_STORE(h.t, &a.g.depth[pix]);
    hitT = h.t;
    normal_packed = nrd_pack_normal(nw, 1.0f, (float)pal);
    DUST_NT_STORE(normal_packed, &a.g.normal[pix]);
    DUST_NT_STORE((h.voxel << 24) | (h.inst & 0xFFFFu) | (pal << 16), &a.g.voxel_id[pix]);
    const V3 hpw = mk(h.t * d.x + o.x, h.t * d.y + o.y, h.t * d.z + o.z);
    const V3 hpm = xform_point(in.w2o, hpw);
    DUST_RO(float) P = in.prev;
    const float hx = ((P[0] * hpm.x + P[4] * hpm.y) + P[8] * hpm.z) + P[12];
    const float hy = ((P[1] * hpm.x + P[5] * hpm.y) + P[9] * hpm.z) + P[13];
    const float hz = ((P[2] * hpm.x + P[6] * hpm.y) + P[10] * hpm.z) + P[14];
    const float hw = ((P[3] * hpm.x + P[7] * hpm.y) + P[11] * hpm.z) + P[15];
    const V3 hp = div3(mk(hx, hy, hz), hw);
    store_half4(a.g.motion, pix, hp.x - hpw.x, hp.y - hpw.y, hp.z - hpw.z, 0.0f);
  }
}

}  // namespace

// the launch's hand-out schedule (next_packet): tiles per band, the exact-quotient multiplier for tile / tiles_x, and how many
// rounds are dealt -- all but roughly the last quarter of a band's tiles, which the waves grab as they finish
static FrameArgs with_schedule(const FrameArgs& in, uint32_t grid, uint32_t block) {
  FrameArgs a = in;
  const uint32_t total = a.tiles_x * a.tiles_y;
  a.tiles_per_band = (total + kRegions - 1u) / kRegions;
  // floor(2^32 / d) + 1 gives the exact quotient for n * d < 2^32; launches beyond that are one-row lists (tiles_y == 1: quotient 0)
  // (and not for one COLUMN of tiles: 2^32 / 1 + 1 does not fit the multiplier -- as 1 it sent every tile of a frame up to 8 pixels
  // wide to the first tile row; tools/stress_host.py bands, seed 111)
  const bool exact = a.tiles_y > 1u && a.tiles_x > 1u && (unsigned long long)total * a.tiles_x < (1ull << 32);
  a.tiles_x_magic = exact ? (uint32_t)((1ull << 32) / a.tiles_x) + 1u : 0u;  // 0: the kernel divides (frames beyond ~11K x 11K) or has one row
  const uint32_t waves = ((grid + kRegions - 1u) / kRegions) * (block / 64u);  // the fullest band's
  const uint32_t rounds = waves ? a.tiles_per_band / waves : 0u;
  a.static_rounds = rounds >= 1u ? 1u : 0u;  // (see next_packet: dealing more than the first round was measured and lost)
  if (a.static_rounds_request != 0xFFFFFFFFu) a.static_rounds = a.static_rounds_request;
  return a;
}
// kernel<MODE>: bit 0 = counting build, bit 1 = DEEP (the scene holds a 4096^3 model), bit 2 = LARGE (more than kFlatCullMax instances)
#define DUST_LAUNCH_MODE(kernel, count, a_in)                                                       \
  do {                                                                                              \
    const FrameArgs a = with_schedule(a_in, grid, block);                                           \
    switch (((count) ? 1 : 0) | ((a).deep ? 2 : 0) | ((a).n_groups ? 4 : 0)) {                      \
      case 0: hipLaunchKernelGGL(kernel<0>, dim3(grid), dim3(block), lds, s, a); break;             \
      case 1: hipLaunchKernelGGL(kernel<1>, dim3(grid), dim3(block), lds, s, a); break;             \
      case 2: hipLaunchKernelGGL(kernel<2>, dim3(grid), dim3(block), lds, s, a); break;             \
      case 3: hipLaunchKernelGGL(kernel<3>, dim3(grid), dim3(block), lds, s, a); break;             \
      case 4: hipLaunchKernelGGL(kernel<4>, dim3(grid), dim3(block), lds, s, a); break;             \
      case 5: hipLaunchKernelGGL(kernel<5>, dim3(grid), dim3(block), lds, s, a); break;             \
      case 6: hipLaunchKernelGGL(kernel<6>, dim3(grid), dim3(block), lds, s, a); break;             \
      default: hipLaunchKernelGGL(kernel<7>, dim3(grid), dim3(block), lds, s, a); break;            \
    }                                                                                               \
  } while (0)
static size_t lds_bytes(const FrameArgs& a, uint32_t block) {
  return (size_t)a.n_lds_models * kN16LdsBytes + (size_t)(block / 64u) * (kMaxCand * 8u + 8u) + 16u + (size_t)a.n_lds_boxes * 32u;  // roots, candidate lists + tile accounts, tile queue, boxes
}

}  // namespace dust
