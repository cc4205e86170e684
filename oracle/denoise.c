/*
 * denoise.c -- CPU restatement of the product's spatiotemporal accumulation filter (TEST INFRASTRUCTURE ONLY).
 *
 * PARITY UNPINNED against the reference: the reference's denoiser is NVIDIA NRD (ReBLUR) behind nrd-sys 0.2.0
 * (crates/render/src/pipeline/nrd.rs:272-617), a closed SDK whose arithmetic is not in the reference tree. What this file
 * restates is the filter dust_amd/csrc/denoise.hip defines -- same inputs (nrd.rs:355-372: motion, normal + roughness,
 * view-z, radiance + hit distance), same knobs (nrd.rs:768-785 and NRD's defaults) -- written here as plain scalar loops from
 * the filter's description (DESIGN.md "Denoiser"), so that the kernels have a second implementation to agree with; the
 * property tests in tests/test_gpu_denoise.py pin what any such filter must do.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

typedef struct { float x, y, z; } d3;
static d3 D3(float x, float y, float z) { d3 r = {x, y, z}; return r; }
static float ddot(d3 a, d3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
static float dclamp(float x, float a, float b) { return fminf(fmaxf(x, a), b); }

static d3 cam_dir(const OrcCamera* c, float aspect, uint32_t w, uint32_t h, float px, float py) { /* camera.glsl:4-16 */
  float cx = 2.0f * ((px + 0.5f) / (float)w) - 1.0f, cy = 2.0f * ((py + 0.5f) / (float)h) - 1.0f;
  cy *= -1.0f;
  cx *= aspect;
  cx *= c->tan_half_fov; cy *= c->tan_half_fov;
  return D3((c->col0[0] * cx + c->col1[0] * cy) - c->col2[0], (c->col0[1] * cx + c->col1[1] * cy) - c->col2[1],
            (c->col0[2] * cx + c->col1[2] * cy) - c->col2[2]);
}
static d3 texel_normal(uint32_t p) { /* nrd.glsl:54-94 on an A2B10G10R10 texel */
  float v4[4], n[3];
  v4[0] = (float)(p & 1023u) / 1023.0f; v4[1] = (float)((p >> 10) & 1023u) / 1023.0f; v4[2] = 0.0f; v4[3] = 0.0f;
  orc_nrd_unpack_normal(v4, n);
  return D3(n[0], n[1], n[2]);
}
static d3 texel_radiance(const uint16_t* p, float* hitdist) { /* nrd.glsl:107-125 */
  const float Y = orc_f16_to_f32(p[0]), Co = orc_f16_to_f32(p[1]), Cg = orc_f16_to_f32(p[2]);
  *hitdist = orc_f16_to_f32(p[3]);
  const float t = Y - Cg;
  return D3(fmaxf(t + Co, 0.0f), fmaxf(Y + Cg, 0.0f), fmaxf(t - Co, 0.0f));
}
static float lum(d3 c) { return (c.x * 0.25f + c.y * 0.5f) + c.z * 0.25f; }

/* the spatial pass's taps: golden-angle spiral of 8 points, radius sqrt((k + 0.5) / 8), weight exp(-2 r^2); 16 rotations */
static const float kDisc[8][3] = {
    {0.25f, 0.0f, 0.882496893f}, {-0.319290102f, 0.292495877f, 0.687289298f}, {0.0488724671f, -0.55687654f, 0.535261452f},
    {0.402444482f, 0.524917543f, 0.416862011f}, {-0.738535106f, -0.130636469f, 0.324652463f}, {0.699604928f, -0.445031404f, 0.252839595f},
    {-0.234004155f, 0.870483816f, 0.196911678f}, {-0.4462713f, -0.859268248f, 0.153354973f}};
static const float kRot[16][2] = {
    {0.980785251f, 0.195090324f}, {0.831469595f, 0.555570245f}, {0.555570245f, 0.831469595f}, {0.195090324f, 0.980785251f},
    {-0.195090324f, 0.980785251f}, {-0.555570245f, 0.831469595f}, {-0.831469595f, 0.555570245f}, {-0.980785251f, 0.195090324f},
    {-0.980785251f, -0.195090324f}, {-0.831469595f, -0.555570245f}, {-0.555570245f, -0.831469595f}, {-0.195090324f, -0.980785251f},
    {0.195090324f, -0.980785251f}, {0.555570245f, -0.831469595f}, {0.831469595f, -0.555570245f}, {0.980785251f, -0.195090324f}};

void orc_denoise(const OrcDenoise* f) {
  const uint32_t W = f->width, H = f->height;
  const float aspect = (float)W / (float)H;
  /* ---- temporal pass */
  for (uint32_t py = 0; py < H; ++py)
    for (uint32_t px = 0; px < W; ++px) {
      const size_t i = (size_t)py * W + px;
      const float t = f->depth[i];
      float* out = f->hist_out_accum + i * 4;
      f->hist_out_depth[i] = t;
      if (t == INFINITY) {
        out[0] = out[1] = out[2] = out[3] = 0.0f;
        f->hist_out_normal[i] = 0u;
        f->hist_out_id[i] = 0xFFFFFFFFu;
        continue;
      }
      const uint32_t npk = f->normal[i], id = f->voxel_id[i] & 0xFFFFu;
      f->hist_out_normal[i] = npk;
      f->hist_out_id[i] = id;
      const d3 n = texel_normal(npk);
      float hitdist;
      const d3 cur = texel_radiance(f->illuminance + i * 4, &hitdist);
      d3 hist = D3(0, 0, 0);
      float hist_n = 0.0f;
      if (f->have_history) {
        const d3 d = cam_dir(&f->cam, aspect, W, H, (float)px, (float)py);
        const d3 x = D3(f->cam.pos[0] + t * d.x, f->cam.pos[1] + t * d.y, f->cam.pos[2] + t * d.z);
        const uint16_t* mv = f->motion + i * 4;
        const d3 xp = D3(x.x + orc_f16_to_f32(mv[0]), x.y + orc_f16_to_f32(mv[1]), x.z + orc_f16_to_f32(mv[2]));
        const d3 rel = D3(xp.x - f->prev.pos[0], xp.y - f->prev.pos[1], xp.z - f->prev.pos[2]);
        const float vx = ddot(D3(f->prev.col0[0], f->prev.col0[1], f->prev.col0[2]), rel);
        const float vy = ddot(D3(f->prev.col1[0], f->prev.col1[1], f->prev.col1[2]), rel);
        const float vz = ddot(D3(f->prev.col2[0], f->prev.col2[1], f->prev.col2[2]), rel);
        if (vz < -1e-6f) {
          const float tp = -vz;
          const float u = (vx / tp) / (aspect * f->prev.tan_half_fov), v = (vy / tp) / f->prev.tan_half_fov;
          float fx = (u * 0.5f + 0.5f) * (float)W - 0.5f, fy = (-v * 0.5f + 0.5f) * (float)H - 0.5f;
          if (fabsf(fx - rintf(fx)) < 0.001953125f) fx = rintf(fx);
          if (fabsf(fy - rintf(fy)) < 0.001953125f) fy = rintf(fy);
          const float x0 = floorf(fx), y0 = floorf(fy), ax = fx - x0, ay = fy - y0;
          float sum_w = 0.0f, acc_n = 0.0f;
          d3 acc = D3(0, 0, 0);
          for (int k = 0; k < 4; ++k) {
            const float xi = x0 + (float)(k & 1), yi = y0 + (float)(k >> 1);
            if (!(xi >= 0.0f && yi >= 0.0f && xi < (float)W && yi < (float)H)) continue;
            const size_t j = (size_t)yi * W + (size_t)xi;
            const float th = f->hist_in_depth[j];
            if (th == INFINITY) continue;
            if (f->hist_in_id[j] != id) continue;
            if (ddot(n, texel_normal(f->hist_in_normal[j])) < 0.9f) continue;
            const d3 dh = cam_dir(&f->prev, aspect, W, H, xi, yi);
            const d3 xh = D3(f->prev.pos[0] + th * dh.x, f->prev.pos[1] + th * dh.y, f->prev.pos[2] + th * dh.z);
            const float off = ddot(n, D3(xh.x - xp.x, xh.y - xp.y, xh.z - xp.z));
            if (fabsf(off) > f->disocclusion_threshold * tp * sqrtf(ddot(dh, dh))) continue;
            const float w = ((k & 1) ? ax : 1.0f - ax) * ((k >> 1) ? ay : 1.0f - ay);
            const float* h = f->hist_in_accum + j * 4;
            acc.x += w * h[0]; acc.y += w * h[1]; acc.z += w * h[2];
            acc_n += w * h[3];
            sum_w += w;
          }
          if (sum_w > 1e-3f) {
            hist = D3(acc.x / sum_w, acc.y / sum_w, acc.z / sum_w);
            hist_n = acc_n / sum_w;
          }
        }
      }
      if (hist_n > 0.0f && f->antilag_power > 0.0f) {
        float s1 = 0.0f, s2 = 0.0f, cnt = 0.0f;
        for (int dy = -2; dy <= 2; ++dy)
          for (int dx = -2; dx <= 2; ++dx) {
            const int xi = (int)px + dx, yi = (int)py + dy;
            if (xi < 0 || yi < 0 || xi >= (int)W || yi >= (int)H) continue;
            const size_t j = (size_t)yi * W + (size_t)xi;
            if (f->depth[j] == INFINITY) continue;
            float hd;
            const float y = lum(texel_radiance(f->illuminance + j * 4, &hd));
            s1 += y; s2 += y * y; cnt += 1.0f;
          }
        const float mean = s1 / cnt;
        const float sigma = sqrtf(fmaxf(s2 / cnt - mean * mean, 0.0f));
        const float yh = lum(hist);
        const float yc = dclamp(yh, mean - f->antilag_sigma_scale * sigma, mean + f->antilag_sigma_scale * sigma);
        if (yc != yh && yh > 1e-12f) {
          const float pull = f->antilag_power * (yc / yh - 1.0f) + 1.0f;
          hist = D3(hist.x * pull, hist.y * pull, hist.z * pull);
          hist_n = hist_n * (1.0f - f->antilag_power * fminf(1.0f, fabsf(yh - yc) / yh));
        }
      }
      const float nn = fminf(hist_n + 1.0f, (float)f->max_accumulated_frames);
      const float al = 1.0f / nn;
      out[0] = hist.x * (1.0f - al) + cur.x * al;
      out[1] = hist.y * (1.0f - al) + cur.y * al;
      out[2] = hist.z * (1.0f - al) + cur.z * al;
      out[3] = nn;
    }
  /* ---- spatial pass */
  for (uint32_t py = 0; py < H; ++py)
    for (uint32_t px = 0; px < W; ++px) {
      const size_t i = (size_t)py * W + px;
      const float t = f->depth[i];
      if (t == INFINITY) continue;
      const float* c = f->hist_out_accum + i * 4;
      const float hitdist = orc_f16_to_f32(f->illuminance[i * 4 + 3]);
      d3 sum = D3(c[0], c[1], c[2]);
      float wsum = 1.0f;
      const float radius = fminf(f->max_blur_radius, f->max_blur_radius * (0.25f + 0.75f * (hitdist / (hitdist + 8.0f))) / sqrtf(c[3]));
      if (radius >= 0.5f) {
        const d3 n = texel_normal(f->normal[i]);
        const uint32_t id = f->voxel_id[i] & 0xFFFFu;
        const d3 d = cam_dir(&f->cam, aspect, W, H, (float)px, (float)py);
        const d3 x = D3(f->cam.pos[0] + t * d.x, f->cam.pos[1] + t * d.y, f->cam.pos[2] + t * d.z);
        const float plane_tol = f->disocclusion_threshold * t * sqrtf(ddot(d, d));
        uint32_t h = (px * 0x9E3779B1u) ^ (py * 0x85EBCA77u) ^ (f->frame_index * 0xC2B2AE3Du);
        h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
        const float rc = kRot[h & 15u][0], rs = kRot[h & 15u][1];
        for (int k = 0; k < 8; ++k) {
          const float ox = (kDisc[k][0] * rc - kDisc[k][1] * rs) * radius, oy = (kDisc[k][0] * rs + kDisc[k][1] * rc) * radius;
          const int xi = (int)px + (int)rintf(ox), yi = (int)py + (int)rintf(oy);
          if (xi < 0 || yi < 0 || xi >= (int)W || yi >= (int)H) continue;
          const size_t j = (size_t)yi * W + (size_t)xi;
          const float tj = f->depth[j];
          if (tj == INFINITY || (f->voxel_id[j] & 0xFFFFu) != id) continue;
          if (ddot(n, texel_normal(f->normal[j])) < 0.9f) continue;
          const d3 dj = cam_dir(&f->cam, aspect, W, H, (float)xi, (float)yi);
          const d3 xj = D3(f->cam.pos[0] + tj * dj.x, f->cam.pos[1] + tj * dj.y, f->cam.pos[2] + tj * dj.z);
          const float off = fabsf(ddot(n, D3(xj.x - x.x, xj.y - x.y, xj.z - x.z)));
          if (off > plane_tol) continue;
          const float* cj = f->hist_out_accum + j * 4;
          const float w = kDisc[k][2] * (1.0f - off / plane_tol);
          sum.x += w * cj[0]; sum.y += w * cj[1]; sum.z += w * cj[2];
          wsum += w;
        }
      }
      const d3 r = D3(sum.x / wsum, sum.y / wsum, sum.z / wsum);
      float hd = hitdist;
      if (hd != 0.0f) hd = fmaxf(hd, 1e-7f);
      uint16_t* o = f->denoised + i * 4;
      o[0] = orc_f32_to_f16((r.x * 0.25f + r.y * 0.5f) + r.z * 0.25f);
      o[1] = orc_f32_to_f16((r.x * 0.5f + r.y * 0.0f) + r.z * -0.5f);
      o[2] = orc_f32_to_f16((r.x * -0.25f + r.y * 0.5f) + r.z * -0.25f);
      o[3] = orc_f32_to_f16(hd);
    }
}
