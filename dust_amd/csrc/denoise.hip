// denoise.hip -- spatiotemporal radiance accumulation (DUST_PASS_DENOISE).
//
// The reference hands the noisy per-frame illuminance to NVIDIA's NRD (ReBLUR diffuse) through nrd-sys 0.2.0
// (crates/render/src/pipeline/nrd.rs:272-617): a closed third-party SDK whose arithmetic is not in the reference tree. This
// is NOT that arithmetic; it is a native filter of the same shape, fed by the same inputs the reference feeds NRD
// (nrd.rs:355-372 / examples/castle.rs:199-207: world-space motion, packed normal + roughness, view-z = the primary ray's t,
// YCoCg radiance + hit distance) and steered by the same knobs the reference sets (nrd.rs:768-785: antilag
// luminance_sigma_scale 2.0, luminance_antilag_power 0.8; NRD defaults: 30 accumulated frames, 1 % disocclusion threshold):
//   k_denoise_temporal  per pixel: reproject the surface point into the previous frame (world-space motion vector +
//                       previous camera), fetch the accumulated history bilinearly from the taps that pass the disocclusion
//                       tests (same instance, normals agree, the tap's surface point lies on this pixel's plane within
//                       1 % of the distance), pull a history that left the current neighbourhood's luminance range back
//                       towards it (antilag), blend 1/N of the new sample in; write history (radiance, frame count, depth,
//                       normal, instance) for the next frame
//   k_denoise_spatial   per pixel: edge-aware blur of the accumulated radiance over a rotated 8-tap disc whose radius shrinks
//                       with the frame count and with short hit distances (contact detail); writes img_illuminance_denoised
// Streaming kernels, HBM-bound: one 16x16 pixel tile per 256-thread workgroup (the taps of neighbouring pixels share cache
// lines in both directions), every record moved with one 8- or 16-byte access, and the temporal pass's 5x5 luminance window
// read from a 20x20 tile staged in LDS (1.6 loads per pixel instead of 50). Parity: oracle/denoise.c restates this
// filter; property tests pin what any such filter must do (static view == running mean, no ghost behind a moving instance).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "denoise.hpp"
#include "exact_div.hpp"

namespace dust {
namespace {

// the spatial pass's 8-tap disc (golden-angle spiral, radius sqrt((k + 0.5) / 8), weight exp(-2 r^2)) and its 16 rotations
// ((k + 0.5) 2 pi / 16): literal tables so that the restatement in oracle/denoise.c picks exactly the same taps
__device__ static const float kDenoiseDisc[8][3] = {
    {0.25f, 0.0f, 0.882496893f}, {-0.319290102f, 0.292495877f, 0.687289298f}, {0.0488724671f, -0.55687654f, 0.535261452f},
    {0.402444482f, 0.524917543f, 0.416862011f}, {-0.738535106f, -0.130636469f, 0.324652463f}, {0.699604928f, -0.445031404f, 0.252839595f},
    {-0.234004155f, 0.870483816f, 0.196911678f}, {-0.4462713f, -0.859268248f, 0.153354973f}};
__device__ static const float kDenoiseRotation[16][2] = {
    {0.980785251f, 0.195090324f}, {0.831469595f, 0.555570245f}, {0.555570245f, 0.831469595f}, {0.195090324f, 0.980785251f},
    {-0.195090324f, 0.980785251f}, {-0.555570245f, 0.831469595f}, {-0.831469595f, 0.555570245f}, {-0.980785251f, 0.195090324f},
    {-0.980785251f, -0.195090324f}, {-0.831469595f, -0.555570245f}, {-0.555570245f, -0.831469595f}, {-0.195090324f, -0.980785251f},
    {0.195090324f, -0.980785251f}, {0.555570245f, -0.831469595f}, {0.831469595f, -0.555570245f}, {0.980785251f, -0.195090324f}};

struct F3 { float x, y, z; };
__device__ __forceinline__ F3 f3(float x, float y, float z) { F3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ float dot(F3 a, F3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
__device__ __forceinline__ float half_to_float(uint16_t h) { return (float)__builtin_bit_cast(_Float16, h); }
__device__ __forceinline__ uint16_t float_to_half(float f) { return __builtin_bit_cast(uint16_t, (_Float16)f); }
__device__ __forceinline__ float clampf(float x, float a, float b) { return fminf(fmaxf(x, a), b); }
__device__ __forceinline__ float stepf(float edge, float x) { return x < edge ? 0.0f : 1.0f; }

// The divisions below are IEEE quotients (oracle/denoise.c divides), taken through exact_div.hpp: by a constant, by the frame
// size (one reciprocal per thread), by a vector's length (one reciprocal per vector) -- 3-4 instructions instead of 10 each,
// the same bits; the filter's taps are arithmetic-bound otherwise (8 divisions per tap).
struct FrameRecip { float w, h, inv_w, inv_h; };
__device__ __forceinline__ FrameRecip frame_recip(uint32_t w, uint32_t h) {
  FrameRecip r;
  r.w = (float)w; r.h = (float)h; r.inv_w = 1.0f / r.w; r.inv_h = 1.0f / r.h;
  return r;
}
// camera.glsl:4-16 for an arbitrary (current or previous) camera
__device__ __forceinline__ F3 ray_dir(const DevCamera& c, float aspect, const FrameRecip& fr, float px, float py) {
  float cx = 2.0f * div_by(px + 0.5f, fr.w, fr.inv_w, false) - 1.0f, cy = 2.0f * div_by(py + 0.5f, fr.h, fr.inv_h, false) - 1.0f;
  cy *= -1.0f;
  cx *= aspect;
  cx *= c.tan_half_fov; cy *= c.tan_half_fov;
  return f3((c.col0[0] * cx + c.col1[0] * cy) - c.col2[0], (c.col0[1] * cx + c.col1[1] * cy) - c.col2[1],
            (c.col0[2] * cx + c.col1[2] * cy) - c.col2[2]);
}
// nrd.glsl:54-94 on an A2B10G10R10 texel
__device__ __forceinline__ F3 unpack_normal(uint32_t p) {
  const float px = div_const((float)(p & 1023u), 1023.0f) * 2.0f - 1.0f, py = div_const((float)((p >> 10) & 1023u), 1023.0f) * 2.0f - 1.0f;
  F3 n = f3(px, py, (1.0f - fabsf(px)) - fabsf(py));
  const float t = clampf(-n.z, 0.0f, 1.0f);
  n.x -= t * (stepf(0.0f, n.x) * 2.0f - 1.0f);
  n.y -= t * (stepf(0.0f, n.y) * 2.0f - 1.0f);
  const float l = sqrtf(dot(n, n));
  const float y = 1.0f / l;
  const bool sp = recip_special(y);
  return f3(div_by(n.x, l, y, sp), div_by(n.y, l, y, sp), div_by(n.z, l, y, sp));
}
// nrd.glsl:107-125: YCoCg + hit distance in four halves -> linear radiance
__device__ __forceinline__ F3 unpack_radiance(const uint16_t* p, float& hitdist) {
  const float Y = half_to_float(p[0]), Co = half_to_float(p[1]), Cg = half_to_float(p[2]);
  hitdist = half_to_float(p[3]);
  const float t = Y - Cg;
  return f3(fmaxf(t + Co, 0.0f), fmaxf(Y + Cg, 0.0f), fmaxf(t - Co, 0.0f));
}
__device__ __forceinline__ float luminance(F3 c) { return (c.x * 0.25f + c.y * 0.5f) + c.z * 0.25f; }  // the Y of nrd.glsl:97-105
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// the same four halves as ONE 8-byte load
__device__ __forceinline__ F3 unpack_radiance(u32x2 q, float& hitdist) {
  const float Y = half_to_float((uint16_t)q.x), Co = half_to_float((uint16_t)(q.x >> 16)), Cg = half_to_float((uint16_t)q.y);
  hitdist = half_to_float((uint16_t)(q.y >> 16));
  const float t = Y - Cg;
  return f3(fmaxf(t + Co, 0.0f), fmaxf(Y + Cg, 0.0f), fmaxf(t - Co, 0.0f));
}
constexpr int kTile = 16, kHalo = 2, kLumW = kTile + 2 * kHalo;  // the antilag window is 5x5

}  // namespace

__global__ void __launch_bounds__(256) k_denoise_temporal(DenoiseArgs a) {
  __shared__ float lum[kLumW * kLumW];  // luminance of the tile and a 2-pixel rim, -1 = no surface there (or outside the frame)
  const int bx = (int)blockIdx.x * kTile, by = (int)blockIdx.y * kTile;
  if (a.antilag_power > 0.0f) {
    for (int k = (int)threadIdx.x; k < kLumW * kLumW; k += 256) {
      const int xi = bx - kHalo + k % kLumW, yi = by - kHalo + k / kLumW;
      float y = -1.0f;
      if (xi >= 0 && yi >= 0 && xi < (int)a.width && yi < (int)a.height) {
        const size_t j = (size_t)yi * a.width + (size_t)xi;
        if (!(a.depth[j] == INFINITY)) {
          float hd;
          y = luminance(unpack_radiance(reinterpret_cast<const u32x2*>(a.illuminance)[j], hd));
        }
      }
      lum[k] = y;
    }
    __syncthreads();
  }
  const uint32_t px = (uint32_t)bx + (threadIdx.x & 15u), py = (uint32_t)by + (threadIdx.x >> 4);
  if (px >= a.width || py >= a.height) return;
  const uint32_t i = py * a.width + px;
  const float t = a.depth[i];
  f32x4* out = reinterpret_cast<f32x4*>(a.hist_out_accum) + i;
  u32x4* geo = reinterpret_cast<u32x4*>(a.hist_out_geo) + i;
  if (t == INFINITY) {  // primary miss: the sky went straight to the denoised target (miss.rmiss:13); no history here
    *out = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    *geo = u32x4{__float_as_uint(t), 0u, 0xFFFFFFFFu, 0u};
    return;
  }
  const FrameRecip fr = frame_recip(a.width, a.height);
  const uint32_t npk = a.normal[i], id = a.voxel_id[i] & 0xFFFFu;
  *geo = u32x4{__float_as_uint(t), npk, id, 0u};
  const F3 n = unpack_normal(npk);
  float hitdist;
  const F3 cur = unpack_radiance(reinterpret_cast<const u32x2*>(a.illuminance)[i], hitdist);
  // ---- reprojection
  F3 hist = f3(0, 0, 0);
  float hist_n = 0.0f;
  if (a.have_history) {
    const F3 d = ray_dir(a.cam, a.aspect, fr, (float)px, (float)py);
    const F3 x = f3(a.cam.pos[0] + t * d.x, a.cam.pos[1] + t * d.y, a.cam.pos[2] + t * d.z);
    const u32x2 mv = reinterpret_cast<const u32x2*>(a.motion)[i];  // hit.rchit:83-94: where this point was in the previous frame, minus where it is
    const F3 xp = f3(x.x + half_to_float((uint16_t)mv.x), x.y + half_to_float((uint16_t)(mv.x >> 16)), x.z + half_to_float((uint16_t)mv.y));
    const F3 rel = f3(xp.x - a.prev.pos[0], xp.y - a.prev.pos[1], xp.z - a.prev.pos[2]);
    const float vx = dot(f3(a.prev.col0[0], a.prev.col0[1], a.prev.col0[2]), rel), vy = dot(f3(a.prev.col1[0], a.prev.col1[1], a.prev.col1[2]), rel),
                vz = dot(f3(a.prev.col2[0], a.prev.col2[1], a.prev.col2[2]), rel);
    if (vz < -1e-6f) {  // in front of the previous camera (it looks down its -z)
      const float tp = -vz;  // the previous primary ray's parameter of the point: the rays' local z component is -1
      const float u = (vx / tp) / (a.aspect * a.prev.tan_half_fov), v = (vy / tp) / a.prev.tan_half_fov;
      float fx = (u * 0.5f + 0.5f) * (float)a.width - 0.5f, fy = (-v * 0.5f + 0.5f) * (float)a.height - 0.5f;
      // a point that has not moved on screen lands on its pixel centre up to rounding: snap within 1/512 of a pixel, so that
      // a static view accumulates pixel by pixel (exactly the running mean) instead of bleeding 1e-4 of its neighbours in
      if (fabsf(fx - rintf(fx)) < 0.001953125f) fx = rintf(fx);
      if (fabsf(fy - rintf(fy)) < 0.001953125f) fy = rintf(fy);
      const float x0 = floorf(fx), y0 = floorf(fy);
      const float ax = fx - x0, ay = fy - y0;
      float sum_w = 0.0f;
      F3 acc = f3(0, 0, 0);
      float acc_n = 0.0f;
      // the four taps' records are requested together (clamped addresses for taps that fall off the frame), then tested
      uint32_t jj[4];
      u32x4 g[4];
      f32x4 hh[4];
      bool in_frame[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xi = x0 + (float)(k & 1), yi = y0 + (float)(k >> 1);
        in_frame[k] = xi >= 0.0f && yi >= 0.0f && xi < (float)a.width && yi < (float)a.height;
        jj[k] = in_frame[k] ? (uint32_t)yi * a.width + (uint32_t)xi : i;
        g[k] = reinterpret_cast<const u32x4*>(a.hist_in_geo)[jj[k]];
        hh[k] = reinterpret_cast<const f32x4*>(a.hist_in_accum)[jj[k]];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float xi = x0 + (float)(k & 1), yi = y0 + (float)(k >> 1);
        if (!in_frame[k]) continue;
        const float th = __uint_as_float(g[k].x);
        if (th == INFINITY) continue;
        if (g[k].z != id) continue;                                 // another instance was there
        if (dot(n, unpack_normal(g[k].y)) < 0.9f) continue;         // another face
        const F3 dh = ray_dir(a.prev, a.aspect, fr, xi, yi);
        const F3 xh = f3(a.prev.pos[0] + th * dh.x, a.prev.pos[1] + th * dh.y, a.prev.pos[2] + th * dh.z);
        const float off = dot(n, f3(xh.x - xp.x, xh.y - xp.y, xh.z - xp.z));  // the tap's surface point against this pixel's plane
        if (fabsf(off) > a.disocclusion * tp * sqrtf(dot(dh, dh))) continue;
        const float w = ((k & 1) ? ax : 1.0f - ax) * ((k >> 1) ? ay : 1.0f - ay);
        const f32x4 h = hh[k];
        acc.x += w * h.x; acc.y += w * h.y; acc.z += w * h.z;
        acc_n += w * h.w;
        sum_w += w;
      }
      if (sum_w > 1e-3f) {
        hist = f3(acc.x / sum_w, acc.y / sum_w, acc.z / sum_w);
        hist_n = acc_n / sum_w;
      }
    }
  }
  // ---- antilag: the history's luminance against the current frame's 5x5 neighbourhood (mean +- sigma_scale sigma)
  if (hist_n > 0.0f && a.antilag_power > 0.0f) {
    float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
    const float* win = lum + (threadIdx.x >> 4) * kLumW + (threadIdx.x & 15u);  // the window's top-left corner in the staged tile
#pragma unroll
    for (int dy = 0; dy < 5; ++dy)
#pragma unroll
      for (int dx = 0; dx < 5; ++dx) {  // row by row, left to right: the order the sums are defined in
        const float y = win[dy * kLumW + dx];
        if (y == -1.0f) continue;
        s1 += y; s2 += y * y; cnt += 1.0f;
      }
    const float mean = s1 / cnt;
    const float sigma = sqrtf(fmaxf(s2 / cnt - mean * mean, 0.0f));
    const float yh = luminance(hist);
    const float yc = clampf(yh, mean - a.antilag_sigma * sigma, mean + a.antilag_sigma * sigma);
    if (yc != yh && yh > 1e-12f) {
      const float pull = a.antilag_power * (yc / yh - 1.0f) + 1.0f;  // mix(1, yc / yh, power)
      hist = f3(hist.x * pull, hist.y * pull, hist.z * pull);
      hist_n = hist_n * (1.0f - a.antilag_power * fminf(1.0f, fabsf(yh - yc) / yh));  // and trust the history less
    }
  }
  const float nn = fminf(hist_n + 1.0f, a.max_frames);
  const float al = 1.0f / nn;
  *out = f32x4{hist.x * (1.0f - al) + cur.x * al, hist.y * (1.0f - al) + cur.y * al, hist.z * (1.0f - al) + cur.z * al, nn};
}

__global__ void __launch_bounds__(256) k_denoise_spatial(DenoiseArgs a) {
  const uint32_t px = blockIdx.x * kTile + (threadIdx.x & 15u), py = blockIdx.y * kTile + (threadIdx.x >> 4);
  if (px >= a.width || py >= a.height) return;
  const uint32_t i = py * a.width + px;
  const float t = a.depth[i];
  if (t == INFINITY) return;  // miss.rmiss:13 wrote the sky there
  const f32x4 cc = reinterpret_cast<const f32x4*>(a.hist_out_accum)[i];
  const float c[4] = {cc.x, cc.y, cc.z, cc.w};
  const float hitdist = half_to_float((uint16_t)(reinterpret_cast<const u32x2*>(a.illuminance)[i].y >> 16));
  F3 sum = f3(c[0], c[1], c[2]);
  float wsum = 1.0f;
  const float radius = fminf(a.max_radius, a.max_radius * (0.25f + 0.75f * (hitdist / (hitdist + 8.0f))) / sqrtf(c[3]));
  if (radius >= 0.5f) {
    const FrameRecip fr = frame_recip(a.width, a.height);
    const F3 n = unpack_normal(a.normal[i]);
    const uint32_t id = a.voxel_id[i] & 0xFFFFu;
    const F3 d = ray_dir(a.cam, a.aspect, fr, (float)px, (float)py);
    const F3 x = f3(a.cam.pos[0] + t * d.x, a.cam.pos[1] + t * d.y, a.cam.pos[2] + t * d.z);
    const float plane_tol = a.disocclusion * t * sqrtf(dot(d, d));
    uint32_t h = (px * 0x9E3779B1u) ^ (py * 0x85EBCA77u) ^ (a.frame_index * 0xC2B2AE3Du);  // which of 16 disc rotations
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
    const float rc = kDenoiseRotation[h & 15u][0], rs = kDenoiseRotation[h & 15u][1];
#pragma unroll
    for (int k = 0; k < 8; ++k) {  // (all eight records requested up front, then tested, measured slower: the early-outs save more arithmetic than the loads cost)
      const float ox = (kDenoiseDisc[k][0] * rc - kDenoiseDisc[k][1] * rs) * radius, oy = (kDenoiseDisc[k][0] * rs + kDenoiseDisc[k][1] * rc) * radius;
      const int xi = (int)px + (int)rintf(ox), yi = (int)py + (int)rintf(oy);
      if (xi < 0 || yi < 0 || xi >= (int)a.width || yi >= (int)a.height) continue;
      const size_t j = (size_t)yi * a.width + (size_t)xi;
      const u32x4 gj = reinterpret_cast<const u32x4*>(a.hist_out_geo)[j];  // this frame's, as the temporal pass just wrote it
      const float tj = __uint_as_float(gj.x);
      if (tj == INFINITY || gj.z != id) continue;
      const float nd = dot(n, unpack_normal(gj.y));
      if (nd < 0.9f) continue;
      const F3 dj = ray_dir(a.cam, a.aspect, fr, (float)xi, (float)yi);
      const F3 xj = f3(a.cam.pos[0] + tj * dj.x, a.cam.pos[1] + tj * dj.y, a.cam.pos[2] + tj * dj.z);
      const float off = fabsf(dot(n, f3(xj.x - x.x, xj.y - x.y, xj.z - x.z)));
      if (off > plane_tol) continue;
      const f32x4 cj = reinterpret_cast<const f32x4*>(a.hist_out_accum)[j];
      const float w = kDenoiseDisc[k][2] * (1.0f - off / plane_tol);
      sum.x += w * cj.x; sum.y += w * cj.y; sum.z += w * cj.z;
      wsum += w;
    }
  }
  const F3 r = f3(sum.x / wsum, sum.y / wsum, sum.z / wsum);
  // REBLUR_FrontEnd_PackRadianceAndNormHitDist layout (nrd.glsl:127-147), what tone mapping reads
  float hd = hitdist;
  if (hd != 0.0f) hd = fmaxf(hd, 1e-7f);
  const uint32_t o0 = float_to_half((r.x * 0.25f + r.y * 0.5f) + r.z * 0.25f), o1 = float_to_half((r.x * 0.5f + r.y * 0.0f) + r.z * -0.5f);
  const uint32_t o2 = float_to_half((r.x * -0.25f + r.y * 0.5f) + r.z * -0.25f), o3 = float_to_half(hd);
  reinterpret_cast<u32x2*>(a.denoised)[i] = u32x2{o0 | (o1 << 16), o2 | (o3 << 16)};
}

hipError_t launch_denoise(const DenoiseArgs& a, hipStream_t s) {
  const dim3 grid((a.width + kTile - 1) / kTile, (a.height + kTile - 1) / kTile);
  hipLaunchKernelGGL(k_denoise_temporal, grid, dim3(256), 0, s, a);
  hipLaunchKernelGGL(k_denoise_spatial, grid, dim3(256), 0, s, a);
  return hipGetLastError();
}

}  // namespace dust
