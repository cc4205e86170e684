// gi.hip -- CDNA4 (gfx950) kernels of the hash-fed global illumination passes: final gather (final_gather.rgen/.rchit/.rmiss,
// rough.rint) and surfel pass (surfel/*.rgen/.rchit/.rmiss, spatial_hash.glsl). Traversal and hash code: traverse.hpp.
#include "traverse.hpp"

namespace dust {

// ==================================================================== final gather
// final_gather.rgen:14-44: is this pixel's gather ray live, and where does it start and point?
// The four texels a live pixel needs are fetched in ONE round trip (gather_fetch), then decoded (gather_decode): depth -> radiance ->
// normal + noise read where they are used were three dependent round trips behind branches the compiler does not hoist loads over
// (k_final_gather 0.225 -> 0.217 ms); a pixel that turns out not to be live has read 16 bytes it did not need.
struct GatherTexels {
  float hitT;
  u32x2 rad;
  uint32_t npk, tex;
};
__device__ __forceinline__ GatherTexels gather_fetch(ArgsRef a, uint32_t px, uint32_t py, bool valid) {
  const size_t pix = valid ? (size_t)py * a.width + px : 0;
  GatherTexels t;
  t.hitT = a.g.depth[pix];
  t.rad = *(const DUST_GLOBAL_AS u32x2*)(a.g.illuminance + pix * 4);
  t.npk = a.g.normal[pix];
  t.tex = ((DUST_RO(uint32_t))a.noise5)[((py + 183u + a.rand) % 128u) * 128u + ((px + 7u + a.rand) % 128u)];
  return t;
}
__device__ __forceinline__ bool gather_decode(ArgsRef a, uint32_t px, uint32_t py, bool valid, const GatherTexels& t, V3& inval, V3& loc, V3& ad) {
  const float hitT = valid ? t.hitT : INFINITY;
  bool live = valid && !(hitT == INFINITY);
  inval = mk(0, 0, 0); loc = mk(0, 0, 0); ad = mk(0, 0, 1);
  if (live) {
    float w;
    inval = decode_radiance(t.rad, w);
    if (w > 0.0f) live = false;  // resolved by the ambient occlusion pass
  }
  if (live) {
    const V3 n = nrd_unpack_normal(t.npk);
    const V3 d = camera_ray_dir(a, px, py);
    loc = mk((hitT * d.x + a.cam.pos[0]) + n.x * 0.01f, (hitT * d.y + a.cam.pos[1]) + n.y * 0.01f,
             (hitT * d.z + a.cam.pos[2]) + n.z * 0.01f);
    const V3 ns = mk(div_const((float)(t.tex & 255u), 255.0f) * 2.0f - 1.0f, div_const((float)((t.tex >> 8) & 255u), 255.0f) * 2.0f - 1.0f,
                     div_const((float)((t.tex >> 16) & 255u), 255.0f) * 2.0f - 1.0f);
    ad = normalize3(rotate_by_normal(n, ns));
  }
  return live;
}
__device__ __forceinline__ bool gather_ray(ArgsRef a, uint32_t px, uint32_t py, bool valid, V3& inval, V3& loc, V3& ad) {
  const GatherTexels t = gather_fetch(a, px, py, valid);
  asm volatile("" ::"v"(t.hitT), "v"(t.rad.x), "v"(t.rad.y), "v"(t.npk), "v"(t.tex));  // (issued here, not sunk into the branches of the decode)
  return gather_decode(a, px, py, valid, t, inval, loc, ad);
}

// Regrouping pre-pass. Gather rays leave neighbouring pixels in unrelated directions, so an 8x8 pixel packet bounds
// nothing by direction and most of its lanes idle through every instance visit. One workgroup per 64x64 pixel tile (four
// pixels per thread) orders the tile's LIVE pixels by direction bin -- the octant their ray points into times the order of
// its components' magnitudes, 48 bins -- with a stable counting sort on ballots (deterministic); k_final_gather then takes
// 64 consecutive entries as a packet: same neighbourhood, similar directions, no dead lanes. Measured on the castle, final
// gather kernel: 32x32 tiles and 8 octants (round 1) 0.304 ms, 24 bins 0.281, 48 bins 0.276, 96 bins 0.273 (but the frame no
// faster); 48 bins on 64x32 tiles 0.267, 64x64 0.256 (96 bins there: 0.250, frame 0.3 % faster), 128x64 0.255 (frame no faster): more rays per tile make a packet's 64
// entries fall into fewer bins, until the spread of their origins costs as much.
// Every pixel's ray, hit and stores are exactly what they were: only the lane a pixel rides in changes.
// position of direction bin (octant * 6 + dominant axis * 2 + which of the other two is larger) along that path (k_gather_order)
__constant__ uint8_t kBinPath[48] = {2, 1, 3, 4, 0, 5, 9, 10, 8, 7, 11, 6, 22, 21, 23, 18, 20, 19, 14, 13, 15, 16, 12, 17,
                                     45, 46, 44, 43, 47, 42, 38, 37, 39, 40, 36, 41, 25, 26, 24, 29, 27, 28, 33, 34, 32, 31, 35, 30};
constexpr uint32_t kOrderTile = 32, kOrderTileW = 64, kOrderTileH = 64, kOrderThreads = kOrderTile * kOrderTile, kOrderSlots = kOrderTileW * kOrderTileH;
__global__ void __launch_bounds__(kOrderThreads) k_gather_order(const FrameArgs) {
  ArgsRef a = launch_args();
  constexpr uint32_t kWaves = kOrderSlots / 64;   // "virtual" waves: the tile's 32x32 quarter h is waves 16 h .. 16 h + 15 of the order
  constexpr uint32_t kBins = 48;
  constexpr uint32_t kCounters = kBins * kWaves;
  __shared__ uint32_t cnt[kCounters];
  __shared__ uint32_t bin_base[kBins];
  __shared__ uint32_t grand_total;
  const uint32_t tile = blockIdx.x, tx = tile % a.gi.order_tiles_x, ty = tile / a.gi.order_tiles_x;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  __shared__ uint8_t bin_path[kBins];  // kBinPath out of LDS: a memory round trip per pixel otherwise
  for (uint32_t i = threadIdx.x; i < kCounters; i += kOrderThreads) cnt[i] = 0u;
  if (threadIdx.x < kBins) bin_path[threadIdx.x] = kBinPath[threadIdx.x];
  if (threadIdx.x == 0) grand_total = 0u;
  __syncthreads();
  constexpr uint32_t kSub = kOrderSlots / kOrderThreads;
  uint32_t keys[kSub], below[kSub], pix[kSub];
  bool lives[kSub];
  GatherTexels texels[kSub];
  constexpr uint32_t kSubX = kOrderTileW / kOrderTile;
#pragma unroll
  for (uint32_t h = 0; h < kSub; ++h) {  // the texels of the thread's four pixels in one round trip
    const uint32_t px = tx * kOrderTileW + (h % kSubX) * 32u + (threadIdx.x % kOrderTile), py = a.row_begin + ty * kOrderTileH + (h / kSubX) * 32u + (threadIdx.x / kOrderTile);
    texels[h] = gather_fetch(a, px, py, px < a.width && py < a.row_end);
  }
#pragma unroll
  for (uint32_t h = 0; h < kSub; ++h) asm volatile("" ::"v"(texels[h].hitT), "v"(texels[h].rad.x), "v"(texels[h].rad.y), "v"(texels[h].npk), "v"(texels[h].tex));
#pragma unroll
  for (uint32_t h = 0; h < kSub; ++h) {
    const uint32_t px = tx * kOrderTileW + (h % kSubX) * 32u + (threadIdx.x % kOrderTile), py = a.row_begin + ty * kOrderTileH + (h / kSubX) * 32u + (threadIdx.x / kOrderTile);
    V3 inval, loc, ad;
    const bool live = gather_decode(a, px, py, px < a.width && py < a.row_end, texels[h], inval, loc, ad);
    const float ax = fabsf(ad.x), ay = fabsf(ad.y), az = fabsf(ad.z);
    const uint32_t dom = ax >= ay && ax >= az ? 0u : (ay >= az ? 1u : 2u);
    const uint32_t sec = dom == 0u ? (ay >= az ? 0u : 1u) : (dom == 1u ? (ax >= az ? 0u : 1u) : (ax >= ay ? 0u : 1u));
    // the bins in the order of a path over the sphere on which consecutive bins are neighbours: inside an octant the six orders of the
    // components' magnitudes by adjacent swaps; from one octant to the next ONE sign flips, that of the smallest component (kBinPath).
    // A packet that straddles two bins -- most do: 4096 entries over 48 bins -- then bounds two adjacent cones instead of two unrelated
    // ones, whose direction intervals straddle zero on two or three axes and leave the packet's cull nothing to reject
    // (k_final_gather 0.2356 -> 0.2274 ms)
    const uint32_t raw = ((ad.x < 0.0f ? 1u : 0u) | (ad.y < 0.0f ? 2u : 0u) | (ad.z < 0.0f ? 4u : 0u)) * 6u + dom * 2u + sec;
    const uint32_t key = live ? (uint32_t)bin_path[raw] : kBins;
    uint64_t peers = ~0ull;
#pragma unroll
    for (uint32_t bit = 0; bit < 6; ++bit) {
      const bool one = (key >> bit) & 1u;
      const uint64_t m = __ballot(one);
      peers &= one ? m : ~m;
    }
    keys[h] = key; lives[h] = live; pix[h] = py * a.width + px;
    below[h] = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
    if (key < kBins && below[h] == 0) cnt[key * kWaves + h * 16u + wave] = (uint32_t)__popcll(peers);
  }
  __syncthreads();
  // a bin's counters are one per lane of a wavefront (64 virtual waves): each wave scans whole bins with shuffles, then the first
  // wave scans the bins' totals -- two barriers, whatever the number of bins
  static_assert(kWaves == 64, "one counter per lane");
  for (uint32_t b = wave; b < kBins; b += kOrderThreads / 64u) {
    const uint32_t v = cnt[b * kWaves + lane];
    uint32_t inc = v;
#pragma unroll
    for (uint32_t d = 1; d < 64; d <<= 1) {
      const uint32_t up = __shfl_up(inc, d);
      if (lane >= d) inc += up;
    }
    cnt[b * kWaves + lane] = inc - v;
    if (lane == 63) bin_base[b] = inc;
  }
  __syncthreads();
  if (wave == 0) {
    uint32_t carry = 0;
    for (uint32_t base = 0; base < kBins; base += 64u) {
      const uint32_t v = base + lane < kBins ? bin_base[base + lane] : 0u;
      uint32_t inc = v;
#pragma unroll
      for (uint32_t d = 1; d < 64; d <<= 1) {
        const uint32_t up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
      }
      if (base + lane < kBins) bin_base[base + lane] = carry + inc - v;
      carry += (uint32_t)__shfl((int)inc, 63);
    }
    if (lane == 0) grand_total = carry;
  }
  __syncthreads();
#pragma unroll
  for (uint32_t h = 0; h < kSub; ++h)
    if (lives[h]) a.gi.order[(size_t)tile * kOrderSlots + bin_base[keys[h]] + cnt[keys[h] * kWaves + h * 16u + wave] + below[h]] = pix[h];
  if (threadIdx.x == 0) a.gi.order_count[tile] = grand_total;
}

// final_gather.rchit:35-91 / final_gather.rmiss:12-24 for one finished gather ray of pixel (px, py)
__device__ __forceinline__ void gather_shade(ArgsRef ar, uint32_t px, uint32_t py, V3 inval, V3 loc, V3 ad, const Hit& h) {
  const size_t pix = (size_t)py * ar.width + px;
  if (!h.found) {
    const V3 sk = sky_radiance(ar.sky, normalize3(ad));
    store_radiance_scattered(ar.g.illuminance, pix, mk(inval.x + sk.x, inval.y + sk.y, inval.z + sk.z), 0.0f);
    return;
  }
  HashKey key;
  DevSurfel sf;
  uint32_t alb;
  brick_surfel(ar, h, loc, ad, key, sf, alb);
  V3 rad;
  uint32_t count;
  uint32_t entry;
  hash_get(ar.gi, key, ar.frame_index, rad, count, entry);
  if (ar.gi.touched) ar.gi.touched[px + py * ar.width] = entry;  // multi-GPU: the other ranks repeat this stamp
  const float prob = 1.0f / (float)(count + 2u);
  const float noise = div_const((float)ar.noise0[((py + 21u + ar.rand) % 128u) * 128u + ((px + 34u + ar.rand) % 128u)], 255.0f);
  if (noise > prob) {  // final_gather.rchit:52-63; the highest pixel index wins the slot (k_surfel_commit)
    const uint32_t index = px + py * ar.width;
    ar.gi.pixel_surfel[index] = sf;
    atomicMax(&ar.gi.slot_owner[index % ar.gi.pool_size], index + 1u);
  }
  rad = modulate_by_avg_albedo(rad, alb);
  store_radiance_scattered(ar.g.illuminance, pix, mk(inval.x + rad.x, inval.y + rad.y, inval.z + rad.z), h.t);
}
// one packet of gather rays, start to finish, with a cull of its own (final_gather.rgen:14-52 + rough.rint)
template <int MODE>
__device__ __forceinline__ void gather_packet(ArgsRef a0, uint32_t px, uint32_t py, bool valid, uint32_t* cand, LaneStats& st) {
  ArgsRef a = reload_args(a0);
  V3 inval, loc, ad;
  const bool live = gather_ray(a, px, py, valid, inval, loc, ad);
#ifdef DUST_TRACE_DEBUG
  {
    const size_t pix = valid ? (size_t)py * a.width + px : 0;
    const unsigned long long m = __ballot(valid && (uint32_t)pix + 1u == (a.debug >> 12));
    if ((threadIdx.x & 63u) == 0) g_dbg_mask[threadIdx.x >> 6] = m;
    __builtin_amdgcn_wave_barrier();
    DBG_PRINT("FG pixel %u,%u live=%d o=%.9g,%.9g,%.9g d=%.9g,%.9g,%.9g\n", px, py, (int)live, loc.x, loc.y, loc.z, ad.x, ad.y, ad.z);
  }
#endif
  Hit h;
  const uint32_t ncand = cull_instances<MODE>(a, __any(live), wave_range(live, loc), wave_range(live, ad), a.cam.far_, cand);
  trace_ray<2, MODE>(a, live, loc, ad, 8.0f, a.cam.far_, false, cand, ncand, h, st);
  __builtin_amdgcn_wave_barrier();
  if (live) gather_shade(reload_args(a0), px, py, inval, loc, ad, h);
}

// final_gather.rgen:14-52 + rough.rint + final_gather.rchit:35-91 + final_gather.rmiss:12-24, a packet at a time
template <int MODE>
__global__ void __launch_bounds__(512, 4) k_final_gather(const FrameArgs) {
  ArgsRef a0 = launch_args();
  stage_roots(a0);
  uint32_t* cand = wave_cand_list(a0);
  LaneStats st = {0, 0, 0, 0, 0, 0};
  WorkCursor wc = cursor_begin();
  Packet p;
  while (next_packet(a0, wc, p)) {
    ArgsRef a = reload_args(a0);  // per packet: nothing of the descriptor rides in SGPRs from one packet to the next
    if (a.gi.order) {  // regrouped: packet id -> 64 entries of one tile's octant-ordered pixel list
      const uint32_t id = p.px / kTileW, tile = id / (kOrderSlots / 64u), idx = (id % (kOrderSlots / 64u)) * 64u + (threadIdx.x & 63u);
      const uint32_t n = a.gi.order_count[tile];
      if ((idx & ~63u) >= n) continue;  // this tile has fewer live pixels
      p.valid = idx < n;
      const uint32_t pixel = p.valid ? a.gi.order[(size_t)tile * kOrderSlots + idx] : 0u;
      p.px = pixel % a.width;
      p.py = pixel / a.width;
    }
    gather_packet<MODE>(a0, p.px, p.py, p.valid, cand, st);
  }
  prof_end();
  flush_stats<MODE>(a0, 0, st);
}

// final_gather.rchit:35-91 / final_gather.rmiss:12-24 as a pass of its own over the hit records k_ray_walk<2> left (DevGI::fg_hits): a
// pixel per thread in pixel order, at full occupancy (the hash probe is a dependent chain instance -> block -> 36 random bytes of a 384 MB
// table). A pixel is live exactly when the ray-making kernel found it live (gather_ray reads the same G-buffer texels: nothing has written
// them in between).
__global__ void __launch_bounds__(256) k_final_gather_shade(const FrameArgs) {
  ArgsRef a = launch_args();
  const uint32_t rows = a.row_end - a.row_begin;
  const size_t n = (size_t)rows * a.width;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t py = a.row_begin + (uint32_t)(i / a.width), px = (uint32_t)(i % a.width);
    V3 inval, loc, ad;
    if (!gather_ray(a, px, py, true, inval, loc, ad)) continue;
    const u32x4 rec = *reinterpret_cast<const u32x4*>(&a.gi.fg_hits[(size_t)py * a.width + px]);
    Hit h;
    h.t = __uint_as_float(rec.x); h.inst = rec.y; h.block = rec.z; h.voxel = 0; h.found = rec.w != 0u;
    gather_shade(a, px, py, inval, loc, ad, h);
  }
}

// the surfel each slot's winning pixel enqueued -> surfel pool; clears the owner table for the next frame
__global__ void k_surfel_commit(const FrameArgs) {
  ArgsRef a = launch_args();
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < a.gi.pool_size; s += gridDim.x * blockDim.x) {
    const uint32_t o = a.gi.slot_owner[s];
    if (o != 0u) {
      a.gi.pool[s] = a.gi.pixel_surfel[o - 1u];
      a.gi.slot_owner[s] = 0u;
    }
  }
}

// ==================================================================== multi-GPU exchange of the final gather's side effects
// (dust_hip.h, dust_hip_pipeline_gi_exchange). slot_owner holds the all-reduced (MAX) owners when these run.
// export: this rank's share of the winning surfels -- the slots whose winning pixel lies in rows [row_begin, row_end)
__global__ void k_gi_export(const FrameArgs) {
  ArgsRef a = launch_args();
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < a.gi.pool_size; s += gridDim.x * blockDim.x) {
    const uint32_t o = a.gi.slot_owner[s];
    DevSurfel v;
    v.x = v.y = v.z = 0.0f; v.direction = 0u;
    if (o != 0u) {
      const uint32_t row = (o - 1u) / a.width;
      if (row >= a.row_begin && row < a.row_end) v = a.gi.pixel_surfel[o - 1u];
    }
    a.gi.merged[s] = v;
  }
}
// import: repeat the last_accessed_frame stamps of the other bands' final gather, commit the merged winners
__global__ void k_gi_import(const FrameArgs) {
  ArgsRef a = launch_args();
  const uint32_t n_px = a.width * a.height;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_px; i += gridDim.x * blockDim.x) {
    const uint32_t row = i / a.width;
    if (row >= a.row_begin && row < a.row_end) continue;  // this rank's own final gather already stamped those
    const uint32_t e = a.gi.touched[i];
    if (e != 0u) reinterpret_cast<uint16_t*>(a.gi.hash + (size_t)(e - 1u) * 3)[4] = (uint16_t)a.frame_index;
  }
  for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < a.gi.pool_size; s += gridDim.x * blockDim.x) {
    if (a.gi.slot_owner[s] != 0u) {
      a.gi.pool[s] = a.gi.merged[s];
      a.gi.slot_owner[s] = 0u;
    }
  }
}

// ==================================================================== surfel pass, phase 0: order the pool by position
// Consecutive pool slots hold the hit points of unrelated final-gather rays, scattered over the whole scene: a packet of
// 64 of them bounds nothing and walks a dozen instances per ray. Sorting the slots by a 16-bit space-filling-curve key of their
// position (32 x 32 x 64 cells over the scene's bounds; dead slots last) makes a packet's origins neighbours, so the packet culling works again.
// Only the grouping into packets changes: every surfel still computes and writes exactly what it did, at its own slot.
__device__ __forceinline__ uint32_t spread10(uint32_t v) {  // 10 bits -> every third bit
  v &= 1023u;
  v = (v | (v << 16)) & 0x030000FFu;
  v = (v | (v << 8)) & 0x0300F00Fu;
  v = (v | (v << 4)) & 0x030C30C3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}
__global__ void k_surfel_keys(const FrameArgs) {
  ArgsRef a = launch_args();
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < a.gi.pool_size; i += gridDim.x * blockDim.x) {
    const DevSurfel e = a.gi.pool[i];
    uint32_t key = 0xFFFFu;  // dead slots sort last
    if (e.direction < 6u) {
      const float q[3] = {e.x, e.y, e.z};
      uint32_t c[3];
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const float span = fmaxf(a.world_max[k] - a.world_min[k], 1.0f);
        const float f = fminf(fmaxf((q[k] - a.world_min[k]) * (1024.0f / span), 0.0f), 1023.0f);  // NaN -> 0
        c[k] = (uint32_t)f;
      }
      // 16 bits (two 8-bit digit passes of radix.hip; a finer order costs more in the sort than it saves in the trace): the index of the
      // surfel's cell along a HILBERT curve through 32^3 cells over the scene's bounds (Skilling's transform, five levels), and one more
      // split along z. Consecutive cells of that curve are always neighbours; along the Z-order of rounds 2-4 a run of 64 surfels that
      // crosses a block boundary jumps across the scene, and one stray origin opens the packet's box and with it the cull (surfel pass
      // 0.306 -> 0.292 ms; six levels, 18-bit keys: 0.309)
      {
        uint32_t X[3] = {c[0] >> 5, c[1] >> 5, c[2] >> 5};
        for (uint32_t Q = 16u; Q > 1u; Q >>= 1) {
          const uint32_t P = Q - 1u;
#pragma unroll
          for (int i = 0; i < 3; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
          }
        }
        X[1] ^= X[0]; X[2] ^= X[1];
        uint32_t t = 0;
        for (uint32_t Q = 16u; Q > 1u; Q >>= 1) if (X[2] & Q) t ^= Q - 1u;
        X[0] ^= t; X[1] ^= t; X[2] ^= t;
        key = (((spread10(X[0]) << 2) | (spread10(X[1]) << 1) | spread10(X[2])) << 1) | ((c[2] >> 4) & 1u);
      }
      key = key < 0xFFFEu ? key : 0xFFFEu;
    }
    a.gi.sort_keys[i] = key;
    a.gi.sort_vals[i] = i;
  }
}

// ==================================================================== surfel pass, phase 1: trace + read the hash
// surfel.rgen:12-67 + rough.rint + surfel.rchit:35-102 + surfel.rmiss:14-26 + surfel/nee.rmiss:15-27
template <int MODE>
__global__ void __launch_bounds__(512, 4) k_surfel_trace(const FrameArgs) {
  ArgsRef a0 = launch_args();
  stage_roots(a0);
  uint32_t* cand = wave_cand_list(a0);
  LaneStats st_sun = {0, 0, 0, 0, 0, 0}, st_cos = {0, 0, 0, 0, 0, 0};
  WorkCursor wc = cursor_begin();
  Packet p;
  const V3 sun = mk(a0.sky[48], a0.sky[49], a0.sky[50]);
  // Work items: 64 consecutive surfels x one ray kind. The closest-hit cosine rays of every group come first (the long
  // items), then the any-hit sun rays: with both rays of a group in one item the pool is only 1.3 items per resident wave and
  // the kernel lasts as long as the waves that drew two. What a lit surfel receives from the sun goes through its own array
  // and is added when the request is applied (surfel.rmiss:14-26 / surfel.rchit:35-102 add it to the same value there).
#ifdef DUST_PROFILE
  // profile builds, DUST_HIP_DEBUG = (cycles / 1024) << 12: only the work items that took at least that long stay in the section buckets
  // (what is different about the items the kernel's length hangs on, tools/kernel_sections.py --heavy). Lane i keeps bucket i as it was
  // when the item began and puts it back if the item turns out short; P_TOTAL then counts the kept items' cycles.
  unsigned long long pf_snap = 0, pf_t0 = 0, pf_kept = 0;
  bool pf_open = false;
  const unsigned long long pf_min = (unsigned long long)(a0.debug >> 12) << 10;
  auto pf_item = [&](bool last) {
    if (pf_min == 0ull) return;
    const unsigned long long now = __builtin_amdgcn_s_memtime();
    const uint32_t w = threadIdx.x >> 6, l = threadIdx.x & 63u;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (pf_open) {
      if (now - pf_t0 < pf_min) { if (l < (uint32_t)kProfBuckets) g_prof[w][l] = pf_snap; }
      else pf_kept += now - pf_t0;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (last && l == 0u) g_prof[w][P_TOTAL] = pf_kept - now;  // (prof_end adds the clock: total = the kept items' cycles)
    if (l < (uint32_t)kProfBuckets) pf_snap = g_prof[w][l];
    pf_t0 = now; pf_open = true;
  };
#endif
  while (next_packet(a0, wc, p)) {  // tiles_x = 2 * ceil(pool_size / 64), tiles_y = 1
#ifdef DUST_PROFILE
    pf_item(false);
#endif
    ArgsRef a = reload_args(a0);  // per packet: nothing of the descriptor rides in SGPRs from one packet to the next
    // (a rank of a sharded trace owns the groups [sf_group_begin, + sf_group_count) of the ordered pool; unsharded: all of them)
    const uint32_t groups = a.sf_stage_req ? a.sf_group_count : (a.gi.pool_size + 63u) / 64u;
    const uint32_t item = p.px / kTileW;
    const bool sun_item = item >= groups;
    const uint32_t slot = ((a.sf_stage_req ? a.sf_group_begin : 0u) + (sun_item ? item - groups : item)) * 64u + (threadIdx.x & 63u);
    const uint32_t i = (a.gi.perm && slot < a.gi.pool_size) ? a.gi.perm[slot] : slot;  // position order, or pool order
    const bool in_range = i < a.gi.pool_size;
    DevSurfel e;
    e.x = e.y = e.z = 0.0f; e.direction = 0xFFFFFFFFu;
    if (in_range) e = a.gi.pool[i];
    const bool live = in_range && e.direction < 6u;
    const V3 n = faceid2normal(live ? e.direction : 0u);
    const V3 org = mk(e.x + 2.01f * n.x, e.y + 2.01f * n.y, e.z + 2.01f * n.z);
    const uint32_t ny0 = i / 128u, nx0 = i - ny0 * 128u;
    const V3 sd = mk(a.sun_dir[0], a.sun_dir[1], a.sun_dir[2]);
    V3 dir = sd;
    bool act = live && dot3(sun, n) > 0.0f;
    if (!sun_item) {
      act = live;
      dir = mk(0, 0, 1);
      if (live) {
        const uint32_t tex = ((DUST_RO(uint32_t))a.noise5)[((ny0 + 47u + a.rand) % 128u) * 128u + ((nx0 + 16u + a.rand) % 128u)];
        const V3 ns = mk(div_const((float)(tex & 255u), 255.0f) * 2.0f - 1.0f, div_const((float)((tex >> 8) & 255u), 255.0f) * 2.0f - 1.0f,
                         div_const((float)((tex >> 16) & 255u), 255.0f) * 2.0f - 1.0f);
        dir = normalize3(rotate_by_normal(n, ns));
      }
    }
    Hit h;
    {
      const Range3 orgs = wave_range(live, org);
      const uint32_t ncand = cull_instances<MODE>(a, __any(act), orgs, sun_item ? point_range(sd) : wave_range(live, dir), 10000.0f, cand);
      LaneStats cur = {0, 0, 0, 0, 0, 0};
      trace_ray<3, MODE>(a, act, org, dir, 0.1f, 10000.0f, sun_item, cand, ncand, h, cur);
      if (COUNT) add_stats(sun_item ? st_sun : st_cos, cur);
      __builtin_amdgcn_wave_barrier();
#ifdef DUST_SURFEL_DEBUG  // (never shipped) every surfel ray with its result, for tools/diag/deep_mismatch.py to hand to the oracle one by one
      if (act) printf("SF %u %d %.9g %.9g %.9g %.9g %.9g %.9g %d %.9g %u %u\n", i, (int)sun_item, org.x, org.y, org.z, dir.x, dir.y, dir.z, (int)h.found, h.t, h.inst, h.block);
#endif
    }
    ArgsRef ar = reload_args(a0);
    if (sun_item) {  // surfel/nee.rmiss:15-27
      f32x4 pay = {0.0f, 0.0f, 0.0f, 0.0f};
      if (act && !h.found) {
        const float dn = dot3(n, sd);
        pay.x = ar.sun_term[0] * dn; pay.y = ar.sun_term[1] * dn; pay.z = ar.sun_term[2] * dn;
      }
      if (ar.sf_stage_req) { if (slot < ar.gi.pool_size) reinterpret_cast<f32x4*>(ar.sf_stage_sun)[slot] = pay; }
      else if (in_range) reinterpret_cast<f32x4*>(ar.gi.sun_payload)[i] = pay;
      continue;
    }
    const V3 cd = dir;
    DevHashRequest rq;
    rq.kx = rq.ky = rq.kz = 0; rq.dir_flags = 0; rq.vx = rq.vy = rq.vz = 0.0f; rq.stamped = 0;
    DevSurfel repl;
    repl.x = repl.y = repl.z = 0.0f; repl.direction = 0xFFFFFFFFu;
    if (live) {
      rq.kx = f2i_trunc(e.x / 4.0f); rq.ky = f2i_trunc(e.y / 4.0f); rq.kz = f2i_trunc(e.z / 4.0f);
      rq.dir_flags = e.direction & 0xFFu;
      if (!h.found) {  // surfel.rmiss:14-26
        const V3 sk = sky_radiance(ar.sky, normalize3(cd));
        rq.vx = sk.x; rq.vy = sk.y; rq.vz = sk.z;
        rq.dir_flags |= 0x100u;
      } else {         // surfel.rchit:35-102
        HashKey key;
        DevSurfel sf;
        uint32_t alb;
        brick_surfel(ar, h, org, cd, key, sf, alb);
        V3 rad;
        uint32_t count = 0;
        uint32_t entry;
        const bool found = hash_get(ar.gi, key, ar.frame_index, rad, count, entry);
        rq.stamped = entry;
        const float rnd0 = div_const((float)ar.noise0[((ny0 + 40u + ar.rand) % 128u) * 128u + ((nx0 + 114u + ar.rand) % 128u)], 255.0f);
        if (found) {
          rad = modulate_by_avg_albedo(rad, alb);
          rq.vx = rad.x; rq.vy = rad.y; rq.vz = rad.z;
          rq.dir_flags |= 0x100u;
        } else if (rnd0 > 1.0f / (float)(count + 2u)) {
          repl = sf;
        }
      }
    }
    if (ar.sf_stage_req) {   // slot order: a rank's records are one contiguous run (dead and out-of-range slots carry "nothing to do")
      if (slot < ar.gi.pool_size) { ar.sf_stage_req[slot] = rq; ar.sf_stage_repl[slot] = repl; }
    } else if (in_range) {
      ar.gi.requests[i] = rq;
      ar.gi.replacement[i] = repl;
    }
  }
#ifdef DUST_PROFILE
  pf_item(true);
#endif
  prof_end();
  flush_stats<MODE>(a0, 0, st_sun);
  flush_stats<MODE>(a0, 1, st_cos);
}

// ==================================================================== surfel pass, sharded trace: slot-ordered staging -> the apply's arrays
// After the all-gather every rank holds every rank's records in SLOT order (FrameArgs::sf_stage_*). One thread per slot moves its three
// records to the surfel's own index -- where the unsharded trace puts them and the apply kernels read them -- and repeats the hash stamp
// the surfel's SpatialHashGet made on the rank that traced it (`stamped`; surfel.rchit:47 -> spatial_hash.glsl:200-219: the low half of
// the entry's third word = frame_index): a 16-bit store of a value every rank agrees on, complete before the apply reads any entry.
__global__ void k_surfel_unstage(const FrameArgs) {
  ArgsRef a = launch_args();
  for (uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x; slot < a.gi.pool_size; slot += gridDim.x * blockDim.x) {
    const uint32_t i = a.gi.perm ? a.gi.perm[slot] : slot;
    if (i >= a.gi.pool_size) continue;
    const DevHashRequest rq = a.sf_stage_req[slot];
    a.gi.requests[i] = rq;
    a.gi.replacement[i] = a.sf_stage_repl[slot];
    reinterpret_cast<f32x4*>(a.gi.sun_payload)[i] = reinterpret_cast<const f32x4*>(a.sf_stage_sun)[slot];
    if (rq.stamped != 0u) reinterpret_cast<uint16_t*>(a.gi.hash + (size_t)(rq.stamped - 1u) * 3)[4] = (uint16_t)a.frame_index;
  }
}

// ==================================================================== surfel pass, phase 2: apply in surfel order
// Deterministic mode: one wavefront scans the requests 64 at a time and lane 0 applies them in index order.
__global__ void __launch_bounds__(64) k_surfel_apply_ordered(const FrameArgs) {
  ArgsRef a = launch_args();
  const uint32_t lane = threadIdx.x;
  for (uint32_t base = 0; base < a.gi.pool_size; base += 64u) {
    const uint32_t i = base + lane;
    bool work = false;
    if (i < a.gi.pool_size) work = ((a.gi.requests[i].dir_flags & 0x100u) && !(a.apply_dead && a.apply_dead[i])) || a.gi.replacement[i].direction != 0xFFFFFFFFu;
    uint64_t mask = __ballot(work);
    if (lane == 0) {
      while (mask) {
        const uint32_t j = base + (uint32_t)__ffsll((long long)mask) - 1u;
        mask &= mask - 1ull;
        const DevHashRequest rq = a.gi.requests[j];
        if ((rq.dir_flags & 0x100u) && !(a.apply_dead && a.apply_dead[j])) {
          HashKey k;
          k.x = rq.kx; k.y = rq.ky; k.z = rq.kz; k.dir = rq.dir_flags & 0xFFu;
          const f32x4 sp = reinterpret_cast<const f32x4*>(a.gi.sun_payload)[j];  // radiance + sun term, as the shaders add them
          hash_insert(a.gi, k, mk(rq.vx + sp.x, rq.vy + sp.y, rq.vz + sp.z), a.frame_index);
        }
        const DevSurfel r = a.gi.replacement[j];
        if (r.direction != 0xFFFFFFFFu) a.gi.pool[j % a.gi.pool_size] = r;
      }
    }
  }
}
// Deterministic mode at full width. A SpatialHashInsert touches the three entries of its probe window and nothing else, so
// two requests only interact when their windows overlap -- hash locations at most 2 apart. Requests sorted by location
// (radix.hip; stable, so equal locations stay in surfel order) therefore fall into CLUSTERS, maximal runs whose consecutive
// locations differ by <= 2; different clusters touch disjoint entries and commute. One thread per cluster applies its
// requests in surfel-index order: the hash ends up exactly as the serial loop above leaves it, at the speed of the racy kernel.
__global__ void k_surfel_apply_keys(const FrameArgs) {  // hash location of every insert request -> sort keys
  ArgsRef a = launch_args();
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < a.gi.pool_size; j += gridDim.x * blockDim.x) {
    const DevHashRequest rq = a.gi.requests[j];
    uint32_t loc = a.gi.hash_capacity;  // "no insert": sorts behind every real location
    if (rq.dir_flags & 0x100u) {
      HashKey k;
      k.x = rq.kx; k.y = rq.ky; k.z = rq.kz; k.dir = rq.dir_flags & 0xFFu;
      loc = key_location(k, a.gi.hash_capacity);
    }
    a.gi.sort_keys[j] = loc;
    a.gi.sort_vals[j] = j;
    const DevSurfel r = a.gi.replacement[j];  // slot replacements are independent of each other and of the hash
    if (r.direction != 0xFFFFFFFFu) a.gi.pool[j] = r;
  }
}
// Which requests a frame applies. Thousands of pixels' gather rays end on the same brick face, so thousands of surfels carry the SAME hash
// key, and the serial definition -- every request, one after the other -- is a dependent chain of decode / blend / encode per key:
// 0.27-0.43 ms for the castle's hottest faces (one thread, ~0.25 us per insert, 1 300 inserts), on EVERY rank of an N-GPU job, which
// needs this mode to keep its replicated hashes identical. The reference's own shader does not apply them all either: invocations that
// run together read the same entry and the last store wins (spatial_hash.glsl:147-195 claims only the fingerprint atomically). The
// defined order here: sort the frame's requests by (hash location, surfel index); a request is SUPERSEDED when the request kApplyKeep
// places further on has the same location and the same key -- of a run of one key, the last kApplyKeep (8) stay --; what stays is applied
// in surfel-index order. A legal outcome of the race, the same on every rank and in the oracle (oracle/shade.c, surfel_apply), and
// the chains are short: clusters 0.27 ms -> see docs/EXPERIMENTS.md round 6.
// One thread per sorted position: bit i of apply_alive, byte vals[i] of apply_dead (every surfel index occurs once in vals).
__global__ void __launch_bounds__(256) k_surfel_apply_mark(const FrameArgs) {
  ArgsRef a = launch_args();
  const uint32_t n = a.gi.pool_size, none = a.gi.hash_capacity;
  const uint32_t rounded = (n + 63u) & ~63u;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < rounded; i += gridDim.x * blockDim.x) {
    bool alive = false, cluster_start = true, run_start = true;   // (positions behind the end: boundaries, so that every scan stops there)
    if (i < n) {
      const uint32_t loc = a.gi.apply_keys[i], j = a.gi.apply_vals[i];
      const uint32_t before = i > 0u ? a.gi.apply_keys[i - 1u] : 0u;
      cluster_start = i == 0u || loc == none || loc - before > 2u;
      run_start = i == 0u || loc != before;
      alive = loc != none;
      if (alive && i + kApplyKeep < n && a.gi.apply_keys[i + kApplyKeep] == loc) {
        const u32x4 k0 = *reinterpret_cast<const u32x4*>(&a.gi.requests[j]);
        const u32x4 k1 = *reinterpret_cast<const u32x4*>(&a.gi.requests[a.gi.apply_vals[i + kApplyKeep]]);
        if (k0.x == k1.x && k0.y == k1.y && k0.z == k1.z && ((k0.w ^ k1.w) & 0xFFu) == 0u) alive = false;
      }
      a.apply_dead[j] = alive ? 0 : 1;
    }
    const uint64_t bal = __ballot(alive), bc = __ballot(cluster_start), br = __ballot(run_start);
    if ((threadIdx.x & 63u) == 0) { a.apply_alive[i >> 6] = bal; a.apply_starts[i >> 6] = bc; a.apply_starts[a.apply_words + (i >> 6)] = br; }
  }
}
// first applied position in [pos, end), or end
__device__ __forceinline__ uint32_t next_alive(const unsigned long long* alive, uint32_t pos, uint32_t end) {
  while (pos < end) {
    const unsigned long long w = alive[pos >> 6] >> (pos & 63u);
    if (w != 0ull) { pos += (uint32_t)__builtin_ctzll(w); return pos < end ? pos : end; }
    pos = (pos | 63u) + 1u;
  }
  return end;
}
__shared__ uint32_t g_apply_slab[256][25];  // k_surfel_apply_clusters: a thread's staged probe windows (8 entries, padded to 25 words)
__global__ void __launch_bounds__(256) k_surfel_apply_clusters(const FrameArgs) {
  ArgsRef a = launch_args();
  const uint32_t n = a.gi.pool_size, none = a.gi.hash_capacity;
  const unsigned long long* alive = a.apply_alive;
  const unsigned long long* cstart = a.apply_starts;                  // cluster boundaries, run boundaries (k_surfel_apply_mark): a cluster of a
  const unsigned long long* rstart = a.apply_starts + a.apply_words;  // thousand requests ends twenty words on, not a thousand dependent loads on
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!((cstart[i >> 6] >> (i & 63u)) & 1ull)) continue;  // not the first request of its cluster
    const uint32_t loc = a.gi.apply_keys[i];
    if (loc == none) continue;
    // (clusters are cut on ALL requests, superseded ones included: a superset of what interacts; positions behind n are boundaries)
    const uint32_t end = next_alive(cstart, i + 1u, (n + 63u) & ~63u);
    auto request = [&](uint32_t j, uint32_t& fp, V3& value) {
      const DevHashRequest rq = a.gi.requests[j];
      HashKey key;
      key.x = rq.kx; key.y = rq.ky; key.z = rq.kz; key.dir = rq.dir_flags & 0xFFu;
      const f32x4 sp = reinterpret_cast<const f32x4*>(a.gi.sun_payload)[j];  // radiance + sun term, as the shaders add them
      fp = key_fingerprint(key);
      value = mk(rq.vx + sp.x, rq.vy + sp.y, rq.vz + sp.z);
    };
    if (a.gi.apply_keys[end - 1] == loc) {
      // Every request of the cluster probes the same window (all but a handful of clusters: surfels of one brick face
      // share a key). The sort is stable, so they already stand in surfel order: load the window once, run them on the
      // register copy, store it once -- a chain of ALU work instead of several dependent memory round trips per request.
      HashWindow win;
      uint32_t* base = a.gi.hash + (size_t)loc * 3;
#pragma unroll
      for (int k = 0; k < 9; ++k) win.w[k] = base[k];
      // eight applied requests at a time: their (independent) loads are in flight together, then the inserts run as one chain
      // of arithmetic on the register window
      constexpr uint32_t kBatch = 8;
      for (uint32_t pos = next_alive(alive, i, end); pos < end;) {
        uint32_t at[kBatch], cnt = 0;
        while (cnt < kBatch && pos < end) { at[cnt++] = pos; pos = next_alive(alive, pos + 1u, end); }
        uint32_t fp[kBatch];
        V3 value[kBatch];
#pragma unroll
        for (uint32_t b = 0; b < kBatch; ++b) request(a.gi.apply_vals[at[b < cnt ? b : cnt - 1u]], fp[b], value[b]);
#pragma unroll
        for (uint32_t b = 0; b < kBatch; ++b)
          if (b < cnt) hash_insert_window(win, fp[b], value[b], a.frame_index);
      }
#pragma unroll
      for (int k = 0; k < 9; ++k) base[k] = win.w[k];
      continue;
    }
    // Windows at different offsets (a few dozen clusters per pass, but some are long: two popular brick faces whose
    // locations happen to lie within two entries of each other). Up to kRuns locations spanning up to kSpan entries: the union
    // of the windows is staged in this thread's LDS slab, the runs (each already in surfel order) are merged by surfel index.
    constexpr uint32_t kRuns = 4, kSpan = 8;
    uint32_t run_at[kRuns], run_end[kRuns], run_loc[kRuns], n_runs = 0;
    bool fits = true;
    for (uint32_t k = i; k < end;) {
      const uint32_t l = a.gi.apply_keys[k];
      const uint32_t e2 = next_alive(rstart, k + 1u, end);   // where the run of this location ends
      if (n_runs < kRuns) { run_at[n_runs] = next_alive(alive, k, e2); run_end[n_runs] = e2; run_loc[n_runs] = l; }
      else fits = false;
      ++n_runs;
      k = e2;
    }
    const uint32_t last_loc = a.gi.apply_keys[end - 1];
    fits = fits && last_loc - loc + 3u <= kSpan;
    if (fits) {
      uint32_t* slab = &g_apply_slab[threadIdx.x][0];
      const uint32_t words = (last_loc - loc + 3u) * 3u;
      uint32_t* base = a.gi.hash + (size_t)loc * 3;
      for (uint32_t k = 0; k < words; ++k) slab[k] = base[k];
      uint32_t head[kRuns];
#pragma unroll
      for (uint32_t r = 0; r < kRuns; ++r) head[r] = (r < n_runs && run_at[r] < run_end[r]) ? a.gi.apply_vals[run_at[r]] : 0xFFFFFFFFu;
      for (;;) {
        uint32_t best = 0;
#pragma unroll
        for (uint32_t r = 1; r < kRuns; ++r) best = head[r] < head[best] ? r : best;
        if (head[best] == 0xFFFFFFFFu) break;
        uint32_t j = 0, l = 0;
#pragma unroll
        for (uint32_t r = 0; r < kRuns; ++r)
          if (r == best) {
            j = head[r]; l = run_loc[r];
            run_at[r] = next_alive(alive, run_at[r] + 1u, run_end[r]);
            head[r] = run_at[r] < run_end[r] ? a.gi.apply_vals[run_at[r]] : 0xFFFFFFFFu;
          }
        uint32_t fp;
        V3 value;
        request(j, fp, value);
        HashMemory m;  // the same accessor: "memory" is the slab
        m.base = slab + (l - loc) * 3u;
        hash_insert_window(m, fp, value, a.frame_index);
      }
      for (uint32_t k = 0; k < words; ++k) base[k] = slab[k];
      continue;
    }
    // anything wider: through memory, in ascending surfel index found by repeated minimum over the applied requests
    uint32_t last = 0;
    for (uint32_t done = 0;; ++done) {
      uint32_t j = 0xFFFFFFFFu, at = i;
      for (uint32_t k = next_alive(alive, i, end); k < end; k = next_alive(alive, k + 1u, end)) {
        const uint32_t v = a.gi.apply_vals[k];
        if ((done == 0 || v > last) && v < j) { j = v; at = k; }
      }
      if (j == 0xFFFFFFFFu) break;
      last = j;
      uint32_t fp;
      V3 value;
      request(j, fp, value);
      HashMemory m;
      m.base = a.gi.hash + (size_t)a.gi.apply_keys[at] * 3;
      hash_insert_window(m, fp, value, a.frame_index);
    }
  }
}
// Throughput mode: every surfel applies its own insert concurrently, as the reference's shaders do (racy by design,
// spatial_hash.glsl:147-195 only claims the fingerprint atomically); results are statistically, not bitwise, repeatable.
__global__ void k_surfel_apply_racy(const FrameArgs) {
  ArgsRef a = launch_args();
  for (uint32_t j = blockIdx.x * blockDim.x + threadIdx.x; j < a.gi.pool_size; j += gridDim.x * blockDim.x) {
    const DevHashRequest rq = a.gi.requests[j];
    if (rq.dir_flags & 0x100u) {
      HashKey k;
      k.x = rq.kx; k.y = rq.ky; k.z = rq.kz; k.dir = rq.dir_flags & 0xFFu;
      const f32x4 sp = reinterpret_cast<const f32x4*>(a.gi.sun_payload)[j];
      hash_insert(a.gi, k, mk(rq.vx + sp.x, rq.vy + sp.y, rq.vz + sp.z), a.frame_index);
    }
    const DevSurfel r = a.gi.replacement[j];
    if (r.direction != 0xFFFFFFFFu) a.gi.pool[j] = r;
  }
}

// ==================================================================== ray streams: bin, then walk one ray per lane
// The incoherent passes -- final gather and surfel rays: neighbouring rays point anywhere -- as the reference's hardware runs
// them: every ray on its own (final_gather.rgen:14-52, surfel.rgen:12-67 launch one invocation per ray; the TLAS of
// accel_struct/tlas.rs:37-117 finds each ray its instances). Three kernels per pass instead of one:
//   1. a ray-making + BINNING kernel, a thread per pixel / surfel at full occupancy (k_gather_rays, k_surfel_rays): the thread
//      makes its ray and walks the top-level grid (DevGrid, staged in LDS) along all of it, listing the instances whose world box
//      the ray meets -- front to back, up to seven (DevRay::cand). A ray that meets none has its miss recorded at once; the
//      others are written to the pass's stream, compacted inside their group (a 16 x 16 pixel tile / 256 surfels): no atomics,
//      the same order in every run;
//   2. k_ray_walk, persistent: a wavefront is 64 LANES that each carry one ray of the stream through its candidates: enter the
//      next one (walk_begin), walk it (walk_step: trace_instance's loop body, verbatim), until the candidates are used up or
//      the ray is settled; the hit record is stored and the lane takes the stream's next ray. A trip of the wave's loop steps
//      every walking lane; when enough lanes are NOT walking they are set up together. This is north_star's "compaction of
//      active rays": a lane never waits for the longest ray of a packet, only for the phase its neighbours are in;
//   3. a shading kernel over the hit records, a thread per pixel / surfel at full occupancy (k_final_gather_shade,
//      k_surfel_shade): the hash lookups with their dependent chain instance -> block -> hash probe, and all stores, coalesced.
// What a ray computes is what trace_ray / trace_instance compute for it: the same brick tests on a superset of the bricks
// that can be accepted, the same tie rule. Only who shares a wavefront with whom changes -- never a result.
// (Round 5 first built 1 + 2 as ONE kernel -- a lane walked the grid itself between two instances and stopped at its hit --: it
// ran at 15 % lane activity, three times the instructions of the packet kernels. docs/EXPERIMENTS.md, round 5.)
//
// Top-level walk. A ray steps through the grid's cells (one axis per step, exit planes from integer cell coordinates). The
// instances of a cell are taken in list order; an instance is skipped when the PREVIOUS cell of the path lies inside the block
// of cells the instance is listed in: it was dealt with there. (The cells of a block that lie on a monotone path are
// consecutive, so "listed in the previous cell" is the same as "listed in any earlier cell"; the block rides in the box
// record's spare words.) Boxes are grown by kGridMargin of the scene's size when they are listed (capi.cpp, build_grid): far
// more than the rounding of the cell steps, so a ray that grazes a cell the steps skipped meets no box listed only there.
enum : uint32_t { RS_EMPTY = 0, RS_FETCH, RS_NEXT, RS_TOP, RS_BEGIN, RS_WALK, RS_DONE };
#ifndef DUST_STREAM_CHUNK
#define DUST_STREAM_CHUNK 64  // rays a wave takes from its band's counter at a time
#endif
constexpr uint32_t kStreamChunk = DUST_STREAM_CHUNK;
constexpr uint32_t kNoCell = 0xFFFFFFFFu;
constexpr uint32_t kMaxRayCand = 7;      // DevRay::cand[0..6]; cand[7] = how many | kCandOverflow
constexpr uint32_t kCandOverflow = 0x8000u;

// Cell coordinates travel as one word with a guard bit above every 8-bit field: x | y << 9 | z << 18, guards at bits 8, 17, 26.
// "p inside the block [lo, hi]" is then two subtractions: ((p | G) - lo) keeps a field's guard bit iff p >= lo there (no borrow
// leaves a field: 256 + p - lo fits its nine bits), likewise ((hi | G) - p).
constexpr uint32_t kCellGuard = (1u << 8) | (1u << 17) | (1u << 26);
struct TopState {
  uint32_t cell, prev;  // packed as above; prev = the path's previous cell (kNoCell: none)
  uint32_t cur, end;    // what is left of the cell's instance list (indices into DevGrid::items)
  float t_end;          // where the ray leaves the grid or its tmax
};
// where the top-level data is read from: LDS sections behind `base` (offsets of a DevStreamLds), or memory
struct TopSource {
  const unsigned char* base;
  uint32_t cells, items, boxes;  // byte offsets, 0xFFFFFFFF: not staged
};
__device__ __forceinline__ uint32_t grid_index(const DUST_CONST_AS DevGrid& g, uint32_t c) {
  return ((c >> 18) * g.dim[1] + ((c >> 9) & 255u)) * g.dim[0] + (c & 255u);
}
__device__ __forceinline__ void open_cell(ArgsRef a, const TopSource& src, uint32_t c, TopState& ts) {
  const uint32_t idx = grid_index(a.grid, c);
  const uint32_t packed = src.cells != 0xFFFFFFFFu ? reinterpret_cast<const uint32_t*>(src.base + src.cells)[idx] : a.grid.cells[idx];
  ts.cur = packed & ((1u << kGridItemBits) - 1u);
  ts.end = ts.cur + (packed >> kGridItemBits);
}
// the ray's first cell; false: the ray misses the grid (no instance can be hit)
__device__ __forceinline__ bool top_begin(ArgsRef a, const TopSource& src, V3 o, V3 d, V3 inv, float tmin, float tmax, TopState& ts) {
  const DUST_CONST_AS DevGrid& g = a.grid;
  float te, tx;
  if (!slab_box(o, d, inv, g.lo, g.hi, te, tx)) return false;
  const float t0 = fmaxf(fmaxf(te, tmin * (1.0f - 1e-5f)), 0.0f);
  const float t1 = fminf(tx, tmax);
  ts.t_end = t1 * (1.0f + 1e-5f) + 1e-3f;
  if (!(t0 <= ts.t_end)) return false;  // (NaN rays end here too)
  const float p[3] = {o.x + d.x * t0, o.y + d.y * t0, o.z + d.z * t0};
  uint32_t c = 0;
#pragma unroll
  for (int k = 0; k < 3; ++k) c |= (uint32_t)f2i_clamp(floorf((p[k] - g.lo[k]) * g.inv_cell[k]), 0, (int)g.dim[k] - 1) << (9 * k);
  ts.cell = c;
  ts.prev = kNoCell;
  open_cell(a, src, c, ts);
  return true;
}
// About `budget` grid steps / box tests. Returns RS_BEGIN with `inst` = an instance whose box the ray meets in front of `limit`
// (the ray's hit so far, or its tmax), RS_DONE when the ray is over -- out of the grid or of [tmin, tmax], or with a hit in front of
// the current cell's exit: every instance not yet looked at is listed only in cells beyond it --, RS_TOP when the budget ran
// out first. Two loops in turn, so that a wave's lanes share the code they run: (A) step from cell to cell until one lists
// something, (B) test what the cell lists.
// zero_axis (wave-uniform): some ray of the wave has a zero direction component -- the box tests then take the general slab test
__device__ __forceinline__ uint32_t top_next(ArgsRef a, const TopSource& src, V3 o, V3 d, V3 inv, float limit, bool found, TopState& ts, uint32_t& inst,
                                             uint32_t budget, bool zero_axis) {
  const DUST_CONST_AS DevGrid& g = a.grid;
  const bool lds_items = src.items != 0xFFFFFFFFu, lds_boxes_ = src.boxes != 0xFFFFFFFFu;
  for (uint32_t it = 0; it < budget;) {
    while (ts.cur >= ts.end) {  // (A) on to the next cell: the exit planes from the integer cell coordinates, one axis per step (a tie takes
      it += 1u;                 //     the lower axis now, the other one on the next step, at the same t). Selects only: no branch inside
      PROF_COUNT_LANES(P_L_EMPTY4, true);
      const uint32_t c0 = ts.cell & 255u, c1 = (ts.cell >> 9) & 255u, c2 = (ts.cell >> 18) & 255u;
      const bool p0 = d.x > 0.0f, p1 = d.y > 0.0f, p2 = d.z > 0.0f;
      const float q0 = (g.lo[0] + (float)(c0 + (p0 ? 1u : 0u)) * g.cell[0] - o.x) * inv.x;
      const float q1 = (g.lo[1] + (float)(c1 + (p1 ? 1u : 0u)) * g.cell[1] - o.y) * inv.y;
      const float q2 = (g.lo[2] + (float)(c2 + (p2 ? 1u : 0u)) * g.cell[2] - o.z) * inv.z;
      const float t0 = d.x != 0.0f ? q0 : INFINITY, t1 = d.y != 0.0f ? q1 : INFINITY, t2 = d.z != 0.0f ? q2 : INFINITY;
      const float tn = fminf(fminf(t0, t1), t2);
      const bool a0 = t0 <= t1 && t0 <= t2, a1 = !a0 && t1 <= t2;  // the stepping axis: 0, else 1, else 2
      const uint32_t step = a0 ? 1u : (a1 ? 1u << 9 : 1u << 18);
      const uint32_t ca = a0 ? c0 : (a1 ? c1 : c2), da = a0 ? g.dim[0] : (a1 ? g.dim[1] : g.dim[2]);
      const bool up = a0 ? p0 : (a1 ? p1 : p2);
      const bool edge = up ? ca + 1u >= da : ca == 0u;
      // over: no axis moves (a zero or NaN direction), what is left lies behind the hit, the ray's end, the grid's edge
      if (!(tn < INFINITY) || (found && limit < tn * (1.0f - 1e-5f) - 1e-4f) || tn > ts.t_end || edge) return RS_DONE;
      ts.prev = ts.cell;
      ts.cell = up ? ts.cell + step : ts.cell - step;
      open_cell(a, src, ts.cell, ts);
      if (it >= budget) return RS_TOP;
    }
    while (ts.cur < ts.end) {  // (B) the cell's instances
      it += 1u;
      PROF_COUNT_LANES(P_L_BRICK, true);
      const uint32_t ii = lds_items ? reinterpret_cast<const uint16_t*>(src.base + src.items)[ts.cur] : a.grid.items[ts.cur];
      ts.cur += 1u;
      f32x4 blo, bhi;
      if (lds_boxes_) { const f32x4* lb = reinterpret_cast<const f32x4*>(src.base + src.boxes); blo = lb[ii * 2u]; bhi = lb[ii * 2u + 1u]; }
      else { blo = *(DUST_RO(f32x4))(&a.boxes[ii].lo[0]); bhi = *(DUST_RO(f32x4))(&a.boxes[ii].hi[0]); }
      // listed in the cell the ray came from: dealt with there
      const uint32_t rl = __float_as_uint(blo.w), rh = __float_as_uint(bhi.w);
      const bool seen = ts.prev != kNoCell && ((((ts.prev | kCellGuard) - rl) & ((rh | kCellGuard) - ts.prev)) & kCellGuard) == kCellGuard;
      const float lo[3] = {blo.x, blo.y, blo.z}, hi[3] = {bhi.x, bhi.y, bhi.z};
      float te, tx;
      const bool box = zero_axis ? slab_box(o, d, inv, lo, hi, te, tx) : slab_box_nonzero(o, inv, lo, hi, te, tx);
      if (!seen && box && !(te * (1.0f - 2e-6f) > limit)) { inst = ii; return RS_BEGIN; }
      if (it >= budget) break;
    }
  }
  return RS_TOP;
}
// a ray-making workgroup's LDS: whatever of the top-level data fits (FrameArgs::sl_bin), from offset 0
__device__ __forceinline__ void copy16(unsigned char* dst, const DUST_CONST_AS void* src, uint32_t bytes) {  // bytes: a multiple of 16 (the image's sections are padded)
  DUST_RO(u32x4) s4 = (DUST_RO(u32x4))src;
  u32x4* d4 = reinterpret_cast<u32x4*>(dst);
  const uint32_t n = bytes / 16u, step = blockDim.x;
  uint32_t i = threadIdx.x;
  for (; i + 3u * step < n; i += 4u * step) {
    const u32x4 v0 = s4[i], v1 = s4[i + step], v2 = s4[i + 2u * step], v3 = s4[i + 3u * step];
    d4[i] = v0; d4[i + step] = v1; d4[i + 2u * step] = v2; d4[i + 3u * step] = v3;
  }
  for (; i < n; i += step) d4[i] = s4[i];
}
__device__ __forceinline__ TopSource stage_bin(ArgsRef a) {
  const uint32_t n_cells = a.grid.dim[0] * a.grid.dim[1] * a.grid.dim[2];
  if (a.sl_bin.cells != 0xFFFFFFFFu) copy16(g_lds + a.sl_bin.cells, a.grid.cells, (n_cells * 4u + 15u) & ~15u);
  if (a.sl_bin.items != 0xFFFFFFFFu) copy16(g_lds + a.sl_bin.items, a.grid.items, (a.grid.n_items * 2u + 15u) & ~15u);
  if (a.sl_bin.boxes != 0xFFFFFFFFu) copy16(g_lds + a.sl_bin.boxes, a.boxes, a.n_instances * 32u);
  __syncthreads();
  TopSource src;
  src.base = g_lds; src.cells = a.sl_bin.cells; src.items = a.sl_bin.items; src.boxes = a.sl_bin.boxes;
  return src;
}
// The whole top-level walk of one ray: the instances whose box it meets in [tmin, tmax], in the order the walk finds them (front to
// back by cell), as DevRay::cand -- eight 16-bit words: up to seven instance ids, then the count (| kCandOverflow when there
// are more: k_ray_walk then walks the grid itself). Returns the count.
__device__ __forceinline__ uint32_t bin_ray(ArgsRef a, const TopSource& src, bool live, V3 o, V3 d, float tmin, float tmax, u32x4& cand) {
  uint32_t w[4] = {0u, 0u, 0u, 0u};
  uint32_t n = 0, over = 0;
  const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));  // (conservative box tests: 1 ulp is enough)
  const bool zero_axis = __any(live && (d.x == 0.0f || d.y == 0.0f || d.z == 0.0f));
  TopState ts;
  if (live && top_begin(a, src, o, d, inv, tmin, tmax, ts)) {
    for (uint32_t guard = 0; guard < 4096u; ++guard) {  // (a path has at most 3 x 256 cells; the bound is a fuse)
      uint32_t inst = 0;
      if (top_next(a, src, o, d, inv, tmax, false, ts, inst, 0x7FFFFFFFu, zero_axis) != RS_BEGIN) break;
      if (n == kMaxRayCand) { over = kCandOverflow; break; }
      const uint32_t v = inst << ((n & 1u) * 16u);
#pragma unroll
      for (uint32_t k = 0; k < 4; ++k) w[k] |= (n >> 1) == k ? v : 0u;
      n += 1u;
    }
  }
  w[3] |= (n | over) << 16;
  cand.x = w[0]; cand.y = w[1]; cand.z = w[2]; cand.w = w[3];
  return n;
}

// RT 2: gather rays (rough.rint, closest hit), RT 3: surfel rays (closest hit, or any hit where DevRay::flags bit 0 is set).
// (Round 6 also ran the PIXEL passes of a 4096^3 scene through this kernel -- camera rays as RT 0, the AO pass's sun and AO rays as RT 1, with
// ray-making and shading kernels around them --: 4.32 ms against the fused packet kernel's 2.53, removed; docs/EXPERIMENTS.md.)
// Statistics slots (counting build): rays without the any-hit flag -> stats[0] for RT 2 / stats[1] for RT 3, any-hit rays -> stats[0];
// the rays that met no box never reach this kernel: the ray-making kernels counted them (gi.unbinned) and block 0 adds them here.
template <int RT, int MODE>
__global__ void __launch_bounds__(1024, 4) k_ray_walk(const FrameArgs) {
  ArgsRef a0 = launch_args();
  // the workgroup's LDS: the roots (as stage_roots), behind them the enter records when they fit (FrameArgs::sl_walk)
  prof_begin();
  if (blockIdx.x == 0 && threadIdx.x < kRegions) a0.next_work_counters[threadIdx.x * kCounterStride] = 0u;
  if (blockIdx.x == 0 && threadIdx.x == 0 && a0.started_word)  // (dust_dev.h: a frame's first traversal launch tells the host it is running)
    __hip_atomic_store((uint32_t*)a0.started_word, a0.started_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  copy16(g_lds, a0.root_table, a0.n_lds_models * kN16LdsBytes);
  if (a0.sl_walk.enters != 0xFFFFFFFFu) copy16(g_lds + a0.n_lds_models * kN16LdsBytes + a0.sl_walk.enters, a0.enters, a0.n_instances * (uint32_t)sizeof(DevEnter));
  __syncthreads();
  PROF_LEAVE(P_STAGE);
  const uint32_t lane = threadIdx.x & 63u;
  const uint64_t lower = (1ull << lane) - 1ull;
  const float tmin = a0.stream.ray_tmin, tmax = a0.stream.ray_tmax;
  // The stream's groups (tiles of the frame / runs of the pool) are cut into chunks of about kStreamChunk rays; the chunks form
  // eight bands, one per XCD (block b runs on XCD b % 8: a band's rays -- neighbours in the frame or the pool -- stay in one L2).
  // A wave takes a chunk at a time from its band's counter, then helps the other bands.
  const uint32_t sub = a0.stream.group_rays / kStreamChunk;        // chunks per group
  const uint32_t chunks = a0.stream.n_groups * sub;
  const uint32_t per = (chunks + kRegions - 1u) / kRegions;   // chunks per band
  const uint32_t own = blockIdx.x & 7u;
  uint32_t win_next = 0, win_end = 0, band_try = 0;
  bool dry = chunks == 0u;
  uint32_t state = RS_EMPTY;
  V3 o = mk(0, 0, 0), d = mk(0, 0, 1);
  uint32_t rid = 0, rflags = 0, pend = 0, ci = 0;
  u32x4 cand = {0u, 0u, 0u, 0u};
  Hit best;
  best.found = false; best.t = tmax; best.inst = 0; best.block = 0; best.voxel = 0;
  TopState ts;
  ts.cell = 0; ts.prev = kNoCell; ts.cur = ts.end = 0; ts.t_end = 0.0f;
  WalkState w;
  w.o = w.d = w.inv = mk(0, 0, 0); w.t = w.tx_stop = w.near_tol = 0.0f; w.ijk[0] = w.ijk[1] = w.ijk[2] = 0;
  w.stepped = 0; w.cl_main = 2; w.steps = 0; w.screen = false; w.prev_whole = false; midcache_reset(w.mc); w.inst = 0;
  w.lds_slot = -1; w.extent = 0; w.root = nullptr; w.dense_mask = nullptr;
  LaneStats cur = {0, 0, 0, 0, 0, 0}, st_closest = {0, 0, 0, 0, 0, 0}, st_any = {0, 0, 0, 0, 0, 0};
  TopSource memsrc;  // (a ray with more candidates than its record holds walks the grid itself, out of memory: rare)
  memsrc.base = g_lds; memsrc.cells = memsrc.items = memsrc.boxes = 0xFFFFFFFFu;
  u32x4 f0 = {0u, 0u, 0u, 0u}, f1 = {0u, 0u, 0u, 0u}, f2 = {0u, 0u, 0u, 0u};  // a ray on its way into the lane (RS_FETCH)
  for (uint32_t trip = 0; trip < (1u << 26); ++trip) {  // (the bound is a fuse: every phase below makes progress)
    PROF_COUNT(P_CAND, 1);
    // ---- empty lanes ask for the stream's next rays, in lane order: the loads are issued here and land while the others walk
    {
      const uint64_t b_empty = __ballot(state == RS_EMPTY);
      if (b_empty != 0ull && !dry) {
        ArgsRef a = reload_args(a0);
        PROF_ENTER(P_GRAB);
        PROF_COUNT(P_N_TRACES, 1);
        if (win_next == win_end) {  // the wave's chunk is used up: the next one of this band, or of the next band that has any
          dry = true;
          while (band_try < kRegions) {
            const uint32_t band = (own + band_try) & 7u;
            uint32_t k = 0;
            if (lane == 0) k = atomicAdd((uint32_t*)&a.work_counters[band * kCounterStride], 1u);
            k = (uint32_t)__builtin_amdgcn_readfirstlane((int)k);
            const uint32_t c = band * per + k;
            if (k >= per || c >= chunks) { band_try += 1u; continue; }
            // chunk c = part c % sub of group c / sub: the group's rays in `sub` equal parts
            const uint32_t grp = c / sub, part = c - grp * sub;
            const uint32_t cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)a.stream.group_count[grp]);
            const uint32_t q = (cnt + sub - 1u) / sub;
            const uint32_t lo = grp * a.stream.group_rays + part * q, hi = grp * a.stream.group_rays + min(cnt, (part + 1u) * q);
            if (lo < hi) { win_next = lo; win_end = hi; dry = false; break; }
          }
        }
        if (!dry) {
          const uint32_t rank = (uint32_t)__popcll(b_empty & lower);
          if (state == RS_EMPTY && win_next + rank < win_end) {
            const u32x4* r = reinterpret_cast<const u32x4*>(a.stream.rays) + (size_t)(win_next + rank) * 3u;
            f0 = r[0]; f1 = r[1]; f2 = r[2];
            state = RS_FETCH;
            PROF_COUNT_LANES(P_N_CAND, true);
          }
          win_next = min(win_end, win_next + (uint32_t)__popcll(b_empty));
        }
        PROF_LEAVE(P_GRAB);
      }
    }
    if (state == RS_WALK) {
      PROF_ENTER(P_INSTANCE);
      PROF_COUNT(P_N_STEPS, 1);
      PROF_COUNT_LANES(P_L_TRIPS, true);
      const DUST_CONST_AS DevVisit& v = a0.visits[w.inst];
      if (walk_step<RT, MODE>(w, &v.m, tmin, tmax, (rflags & 1u) != 0u, best, cur)) state = RS_NEXT;
      PROF_LEAVE(P_INSTANCE);
    }
    const uint32_t n_walk = (uint32_t)__popcll(__ballot(state == RS_WALK));
    if (n_walk > 64u - a0.stream_refill) continue;
    ArgsRef a = reload_args(a0);
    if (state == RS_FETCH) {  // the ray has arrived
      o = mk(__uint_as_float(f0.x), __uint_as_float(f0.y), __uint_as_float(f0.z));
      d = mk(__uint_as_float(f1.x), __uint_as_float(f1.y), __uint_as_float(f1.z));
      rid = f0.w; rflags = f1.w; cand = f2; ci = 0;
      best.found = false; best.t = tmax; best.inst = 0; best.block = 0; best.voxel = 0;
      if (COUNT) { cur.rays = 1; cur.instances_tested = cur.upper_descents = cur.mid_descents = cur.bricks_tested = cur.hits = 0; }
      state = RS_NEXT;
    }
    // ---- the ray's next candidate, or its end
    if (state == RS_NEXT) {
      const uint32_t n = (cand.w >> 16) & 0xFFu;
      if ((rflags & 1u) && best.found) state = RS_DONE;  // an any-hit ray is settled by its first hit
      else if (ci < n) {
        const uint32_t word = (ci >> 1) == 0u ? cand.x : ((ci >> 1) == 1u ? cand.y : ((ci >> 1) == 2u ? cand.z : cand.w));
        pend = (word >> ((ci & 1u) * 16u)) & 0xFFFFu;
        ci += 1u;
        state = RS_BEGIN;
      } else if ((cand.w >> 16) & kCandOverflow) {  // more instances than the record holds: from here the lane walks the grid itself
        cand.w &= 0xFFFFu;                          // (from the start: the instances it meets again give the same hits)
        const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
        state = top_begin(a, memsrc, o, d, inv, tmin, tmax, ts) ? RS_TOP : RS_DONE;
        rflags |= 2u;
      } else state = (rflags & 2u) ? RS_TOP : RS_DONE;
    }
    if (state == RS_TOP) {
      PROF_ENTER(P_CULL);
      PROF_COUNT(P_N_CAND_ITER, 1);
      PROF_COUNT_LANES(P_AO_SETUP, true);
      const V3 inv = mk(__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y), __builtin_amdgcn_rcpf(d.z));
      const bool zero_axis = __any(d.x == 0.0f || d.y == 0.0f || d.z == 0.0f);  // (of the lanes in this phase)
      state = top_next(a, memsrc, o, d, inv, best.found ? best.t : tmax, best.found, ts, pend, a.stream_top_iters, zero_axis);
      PROF_LEAVE(P_CULL);
    }
    if (state == RS_DONE) {  // the ray's hit record; the lane is free
      u32x4 rec;
      rec.x = __float_as_uint(best.t); rec.y = best.inst; rec.z = best.block; rec.w = best.found ? 1u : 0u;
      reinterpret_cast<u32x4*>(a.stream.ray_hits)[rid] = rec;
      if (COUNT) {
        if (best.found) cur.hits = 1;
        if (rflags & 1u) add_stats(st_any, cur); else add_stats(st_closest, cur);
      }
      state = RS_EMPTY;
    }
    // ---- the instance set-ups, together
    if (state == RS_BEGIN) {
      PROF_ENTER(P_SETUP);
      PROF_COUNT(P_N_VISITS, 1);
      PROF_COUNT_LANES(P_PRIMARY_SHADE, true);
      if (COUNT) cur.instances_tested += 1;
      // the instance's enter record: five 16-byte reads, from LDS where it is staged
      u32x4 e0, e1, e2, e3, e4;
      if (a.sl_walk.enters != 0xFFFFFFFFu) {
        const u32x4* le = reinterpret_cast<const u32x4*>(g_lds + a.n_lds_models * kN16LdsBytes + a.sl_walk.enters) + pend * 5u;
        e0 = le[0]; e1 = le[1]; e2 = le[2]; e3 = le[3]; e4 = le[4];
      } else {
        DUST_RO(u32x4) ge = (DUST_RO(u32x4))(a.enters + pend);
        e0 = ge[0]; e1 = ge[1]; e2 = ge[2]; e3 = ge[3]; e4 = ge[4];
      }
      const float m[12] = {__uint_as_float(e0.x), __uint_as_float(e0.y), __uint_as_float(e0.z), __uint_as_float(e0.w),
                           __uint_as_float(e1.x), __uint_as_float(e1.y), __uint_as_float(e1.z), __uint_as_float(e1.w),
                           __uint_as_float(e2.x), __uint_as_float(e2.y), __uint_as_float(e2.z), __uint_as_float(e2.w)};
      EnterView ev;
      ev.bmin[0] = (float)(e3.x & 0xFFFFu); ev.bmin[1] = (float)(e3.x >> 16); ev.bmin[2] = (float)(e3.y & 0xFFFFu);
      ev.bmax[0] = (float)(e3.y >> 16); ev.bmax[1] = (float)(e3.z & 0xFFFFu); ev.bmax[2] = (float)(e3.z >> 16);
      const uint32_t slot = (e3.w >> 16) & 255u;
      ev.lds_slot = slot == 255u ? -1 : (int32_t)slot;
      ev.extent = 1u << (e3.w >> 24);
      ev.root = (DUST_RO(uint8_t))(((uint64_t)e4.y << 32) | e4.x);
      ev.dense_mask = (DUST_RO(uint64_t))(((uint64_t)e4.w << 32) | e4.z);
      const V3 oo = mk(((m[0] * o.x + m[1] * o.y) + m[2] * o.z) + m[3], ((m[4] * o.x + m[5] * o.y) + m[6] * o.z) + m[7],
                       ((m[8] * o.x + m[9] * o.y) + m[10] * o.z) + m[11]);
      const V3 od = mk((m[0] * d.x + m[1] * d.y) + m[2] * d.z, (m[4] * d.x + m[5] * d.y) + m[6] * d.z, (m[8] * d.x + m[9] * d.y) + m[10] * d.z);
      state = walk_begin<RT, MODE>(w, ev, pend, oo, od, tmin) ? RS_WALK : RS_NEXT;
      PROF_LEAVE(P_SETUP);
    }
    if (dry && !__any(state != RS_EMPTY)) break;
  }
  prof_end();
  if (COUNT && blockIdx.x == 0 && threadIdx.x == 0) {  // the rays that met no instance box (counted by the ray-making kernel): traced, missed
    if (RT == 2) st_closest.rays += a0.stream.unbinned[0];
    else { st_any.rays += a0.stream.unbinned[1]; st_closest.rays += a0.stream.unbinned[0]; }
  }
  if (RT == 2) flush_stats<MODE>(a0, 0, st_closest);
  else { flush_stats<MODE>(a0, 0, st_any); flush_stats<MODE>(a0, 1, st_closest); }
}

// A workgroup's place in its group of the stream: `first` / `second` say which of its two possible rays a thread has. Returns the
// thread's first position inside the group (thread order: wave ballots and a scan over the waves' totals); thread 0 writes the
// group's count. Every thread of the workgroup calls it.
__device__ __forceinline__ uint32_t group_reserve(uint32_t* group_count, bool first, bool second) {
  __shared__ uint32_t wave_total[16];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  const uint64_t lower = (1ull << lane) - 1ull;
  const uint64_t b1 = __ballot(first), b2 = __ballot(second);
  if (lane == 0) wave_total[wave] = (uint32_t)(__popcll(b1) + __popcll(b2));
  __syncthreads();
  uint32_t before = 0, all = 0;
  for (uint32_t i = 0; i < n_waves; ++i) { const uint32_t c = wave_total[i]; if (i < wave) before += c; all += c; }
  if (threadIdx.x == 0) *group_count = all;
  return before + (uint32_t)(__popcll(b1 & lower) + __popcll(b2 & lower));
}
__device__ __forceinline__ void put_ray(ArgsRef a, uint32_t at, V3 o, uint32_t id, V3 d, uint32_t flags, u32x4 cand) {
  u32x4 r0, r1;
  r0.x = __float_as_uint(o.x); r0.y = __float_as_uint(o.y); r0.z = __float_as_uint(o.z); r0.w = id;
  r1.x = __float_as_uint(d.x); r1.y = __float_as_uint(d.y); r1.z = __float_as_uint(d.z); r1.w = flags;
  u32x4* r = reinterpret_cast<u32x4*>(a.stream.rays) + (size_t)at * 3u;
  r[0] = r0; r[1] = r1; r[2] = cand;
}
__device__ __forceinline__ void put_miss(ArgsRef a, uint32_t id) {  // the hit record of a ray that meets no instance box
  u32x4 rec;
  rec.x = __float_as_uint(a.stream.ray_tmax); rec.y = 0u; rec.z = 0u; rec.w = 0u;
  reinterpret_cast<u32x4*>(a.stream.ray_hits)[id] = rec;
}
// (counting build of the frame only: how many rays the binning settled itself; one atomic per workgroup)
__device__ __forceinline__ void count_unbinned(ArgsRef a, uint32_t which, bool mine) {
  if (!a.stream.count_unbinned) return;
  const uint32_t n = (uint32_t)__popcll(__ballot(mine));
  if ((threadIdx.x & 63u) == 0 && n) atomicAdd(&a.stream.unbinned[which], n);
}

// final_gather.rgen:14-44 for every pixel of the band, 16 x 16 pixel tile by tile: the pixel's gather ray, binned
__global__ void __launch_bounds__(256) k_gather_rays(const FrameArgs) {
  ArgsRef a = launch_args();
  const TopSource src = stage_bin(a);
  const uint32_t tiles_x = (a.width + 15u) / 16u;
  const uint32_t ty = blockIdx.x / tiles_x, tx = blockIdx.x - ty * tiles_x;
  const uint32_t px = tx * 16u + (threadIdx.x & 15u), py = a.row_begin + ty * 16u + (threadIdx.x >> 4);
  V3 inval, loc, ad;
  const bool live = gather_ray(a, px, py, px < a.width && py < a.row_end, inval, loc, ad);
  u32x4 cand;
  const bool walk = bin_ray(a, src, live, loc, ad, a.stream.ray_tmin, a.stream.ray_tmax, cand) != 0u;
  const uint32_t at = blockIdx.x * 256u + group_reserve(a.stream.group_count + blockIdx.x, walk, false);
  if (walk) put_ray(a, at, loc, py * a.width + px, ad, 0u, cand);
  else if (live) put_miss(a, py * a.width + px);
  count_unbinned(a, 0, live && !walk);
}

// surfel.rgen:12-67: where surfel i's two rays start and where they point. Returns false for a dead slot.
struct SurfelRays { V3 n, org, cos_dir; bool lit; };
__device__ __forceinline__ bool surfel_rays(ArgsRef a, uint32_t i, const DevSurfel& e, SurfelRays& r) {
  const bool live = e.direction < 6u;
  r.n = faceid2normal(live ? e.direction : 0u);
  r.org = mk(e.x + 2.01f * r.n.x, e.y + 2.01f * r.n.y, e.z + 2.01f * r.n.z);
  r.cos_dir = mk(0, 0, 1);
  r.lit = live && dot3(mk(a.sky[48], a.sky[49], a.sky[50]), r.n) > 0.0f;
  if (live) {
    const uint32_t ny0 = i / 128u, nx0 = i - ny0 * 128u;
    const uint32_t tex = ((DUST_RO(uint32_t))a.noise5)[((ny0 + 47u + a.rand) % 128u) * 128u + ((nx0 + 16u + a.rand) % 128u)];
    const V3 ns = mk(div_const((float)(tex & 255u), 255.0f) * 2.0f - 1.0f, div_const((float)((tex >> 8) & 255u), 255.0f) * 2.0f - 1.0f,
                     div_const((float)((tex >> 16) & 255u), 255.0f) * 2.0f - 1.0f);
    r.cos_dir = normalize3(rotate_by_normal(r.n, ns));
  }
  return live;
}
// the surfel pass's rays, in the pool's position order (gi.perm) or pool order, binned: per live surfel the cosine ray (id 2 i), and the
// sun ray (id 2 i + 1, any-hit) where the sun is above the surfel's face
__global__ void __launch_bounds__(256) k_surfel_rays(const FrameArgs) {
  ArgsRef a = launch_args();
  const TopSource src = stage_bin(a);
  const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = (a.gi.perm && slot < a.gi.pool_size) ? a.gi.perm[slot] : slot;
  const bool in_range = slot < a.gi.pool_size && i < a.gi.pool_size;
  DevSurfel e;
  e.x = e.y = e.z = 0.0f; e.direction = 0xFFFFFFFFu;
  if (in_range) e = a.gi.pool[i];
  SurfelRays r;
  const bool live = surfel_rays(a, i, e, r) && in_range;
  const bool lit = live && r.lit;
  const V3 sd = mk(a.sun_dir[0], a.sun_dir[1], a.sun_dir[2]);
  u32x4 cand_cos, cand_sun;
  const bool walk_cos = bin_ray(a, src, live, r.org, r.cos_dir, a.stream.ray_tmin, a.stream.ray_tmax, cand_cos) != 0u;
  const bool walk_sun = bin_ray(a, src, lit, r.org, sd, a.stream.ray_tmin, a.stream.ray_tmax, cand_sun) != 0u;
  const uint32_t at = blockIdx.x * 512u + group_reserve(a.stream.group_count + blockIdx.x, walk_cos, walk_sun);
  if (walk_cos) put_ray(a, at, r.org, 2u * i, r.cos_dir, 0u, cand_cos);
  else if (live) put_miss(a, 2u * i);
  if (walk_sun) put_ray(a, at + (walk_cos ? 1u : 0u), r.org, 2u * i + 1u, sd, 1u, cand_sun);
  else if (lit) put_miss(a, 2u * i + 1u);
  count_unbinned(a, 0, live && !walk_cos);
  count_unbinned(a, 1, lit && !walk_sun);
}
// surfel.rchit:35-102 + surfel.rmiss:14-26 + surfel/nee.rmiss:15-27 over the hit records: a thread per surfel, in pool order --
// what k_surfel_trace does behind its trace, with the hash probes of a whole workgroup in flight together
__global__ void __launch_bounds__(256) k_surfel_shade(const FrameArgs) {
  ArgsRef a = launch_args();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.gi.pool_size) return;
  const DevSurfel e = a.gi.pool[i];
  SurfelRays r;
  const bool live = surfel_rays(a, i, e, r);
  const V3 sd = mk(a.sun_dir[0], a.sun_dir[1], a.sun_dir[2]);
  f32x4 pay = {0.0f, 0.0f, 0.0f, 0.0f};
  DevHashRequest rq;
  rq.kx = rq.kz = rq.ky = 0; rq.dir_flags = 0; rq.vx = rq.vy = rq.vz = 0.0f; rq.stamped = 0;
  DevSurfel repl;
  repl.x = repl.y = repl.z = 0.0f; repl.direction = 0xFFFFFFFFu;
  if (live) {
    if (r.lit) {  // surfel/nee.rmiss:15-27
      const u32x4 hs = reinterpret_cast<const u32x4*>(a.stream.ray_hits)[2u * i + 1u];
      if (hs.w == 0u) {
        const float dn = dot3(r.n, sd);
        pay.x = a.sun_term[0] * dn; pay.y = a.sun_term[1] * dn; pay.z = a.sun_term[2] * dn;
      }
    }
    const u32x4 hc = reinterpret_cast<const u32x4*>(a.stream.ray_hits)[2u * i];
    Hit h;
    h.t = __uint_as_float(hc.x); h.inst = hc.y; h.block = hc.z; h.voxel = 0; h.found = hc.w != 0u;
    rq.kx = f2i_trunc(e.x / 4.0f); rq.ky = f2i_trunc(e.y / 4.0f); rq.kz = f2i_trunc(e.z / 4.0f);
    rq.dir_flags = e.direction & 0xFFu;
    if (!h.found) {  // surfel.rmiss:14-26
      const V3 sk = sky_radiance(a.sky, normalize3(r.cos_dir));
      rq.vx = sk.x; rq.vy = sk.y; rq.vz = sk.z;
      rq.dir_flags |= 0x100u;
    } else {         // surfel.rchit:35-102
      HashKey key;
      DevSurfel sf;
      uint32_t alb;
      brick_surfel(a, h, r.org, r.cos_dir, key, sf, alb);
      V3 rad;
      uint32_t count = 0, entry;
      const bool found = hash_get(a.gi, key, a.frame_index, rad, count, entry);
      rq.stamped = entry;
      const uint32_t ny0 = i / 128u, nx0 = i - ny0 * 128u;
      const float rnd0 = div_const((float)a.noise0[((ny0 + 40u + a.rand) % 128u) * 128u + ((nx0 + 114u + a.rand) % 128u)], 255.0f);
      if (found) {
        rad = modulate_by_avg_albedo(rad, alb);
        rq.vx = rad.x; rq.vy = rad.y; rq.vz = rad.z;
        rq.dir_flags |= 0x100u;
      } else if (rnd0 > 1.0f / (float)(count + 2u)) {
        repl = sf;
      }
    }
  }
  reinterpret_cast<f32x4*>(a.gi.sun_payload)[i] = pay;
  a.gi.requests[i] = rq;
  a.gi.replacement[i] = repl;
}

// a thread per pool slot for the pass's small kernels (latency chains of dependent loads: every slot's chain in flight at once -- 512 or 1024
// workgroups left a 345 600-slot pool two or three chains per thread, one behind the other); grid-stride loops take what a cap leaves
static inline uint32_t pool_grid(const FrameArgs& a) {
  const uint32_t g = (a.gi.pool_size + 255u) / 256u;
  return g < 1u ? 1u : (g > 8192u ? 8192u : g);
}

hipError_t launch_gi_export(const FrameArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_gi_export, dim3(512), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_gi_import(const FrameArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_gi_import, dim3(1024), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_gather_order(const FrameArgs& a, uint32_t n_tiles, hipStream_t s) {
  hipLaunchKernelGGL(k_gather_order, dim3(n_tiles), dim3(kOrderThreads), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_final_gather(const FrameArgs& a_in, uint32_t grid, uint32_t block, bool count, bool commit, hipStream_t s) {
  const size_t lds = lds_bytes(a_in, block);
  DUST_LAUNCH_MODE(k_final_gather, count, a_in);
  if (commit) hipLaunchKernelGGL(k_surfel_commit, dim3(pool_grid(a_in)), dim3(256), 0, s, a_in);
  return hipGetLastError();
}
hipError_t launch_final_gather_shade(const FrameArgs& a, bool commit, hipStream_t s) {
  hipLaunchKernelGGL(k_final_gather_shade, dim3(4096), dim3(256), 0, s, a);
  if (commit) hipLaunchKernelGGL(k_surfel_commit, dim3(pool_grid(a)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_surfel_keys(const FrameArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_surfel_keys, dim3(pool_grid(a)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_surfel_trace(const FrameArgs& a_in, uint32_t grid, uint32_t block, bool count, hipStream_t s) {
  const size_t lds = lds_bytes(a_in, block);
  DUST_LAUNCH_MODE(k_surfel_trace, count, a_in);
  return hipGetLastError();
}
// mode 0: concurrent (racy, as the reference); 1: serial in surfel order (one wavefront); 2: keys for the clustered apply;
// 3: the clustered apply itself (after the sort and the marks); 4: the marks (after the sort)
hipError_t launch_surfel_apply(const FrameArgs& a, int mode, hipStream_t s) {
  if (mode == 0) hipLaunchKernelGGL(k_surfel_apply_racy, dim3(pool_grid(a)), dim3(256), 0, s, a);
  else if (mode == 1) hipLaunchKernelGGL(k_surfel_apply_ordered, dim3(1), dim3(64), 0, s, a);
  else if (mode == 2) hipLaunchKernelGGL(k_surfel_apply_keys, dim3(pool_grid(a)), dim3(256), 0, s, a);
  else if (mode == 4) hipLaunchKernelGGL(k_surfel_apply_mark, dim3(pool_grid(a)), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_surfel_apply_clusters, dim3(1024), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_surfel_unstage(const FrameArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_surfel_unstage, dim3(pool_grid(a)), dim3(256), 0, s, a);
  return hipGetLastError();
}
hipError_t launch_gather_rays(const FrameArgs& a, hipStream_t s) {
  const uint32_t tiles = ((a.width + 15u) / 16u) * ((a.row_end - a.row_begin + 15u) / 16u);
  hipLaunchKernelGGL(k_gather_rays, dim3(tiles), dim3(256), a.sl_bin.total, s, a);
  return hipGetLastError();
}
hipError_t launch_surfel_rays(const FrameArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_surfel_rays, dim3((a.gi.pool_size + 255u) / 256u), dim3(256), a.sl_bin.total, s, a);
  return hipGetLastError();
}
hipError_t launch_surfel_shade(const FrameArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_surfel_shade, dim3((a.gi.pool_size + 255u) / 256u), dim3(256), 0, s, a);
  return hipGetLastError();
}
// rt 2: gather rays, 3: surfel rays; the stream is a.stream.rays / group_count, the hit records go to a.stream.ray_hits
hipError_t launch_ray_walk(const FrameArgs& a_in, int rt, uint32_t grid, uint32_t block, bool count, hipStream_t s) {
  const size_t lds = (size_t)a_in.n_lds_models * kN16LdsBytes + a_in.sl_walk.total;
  const FrameArgs a = with_schedule(a_in, grid, block);
  const int mode = (count ? 1 : 0) | (a.deep ? 2 : 0);
#define DUST_STREAM_CASE(RT_, M_) hipLaunchKernelGGL((k_ray_walk<RT_, M_>), dim3(grid), dim3(block), lds, s, a)
  if (rt == 2) {
    switch (mode) { case 0: DUST_STREAM_CASE(2, 0); break; case 1: DUST_STREAM_CASE(2, 1); break; case 2: DUST_STREAM_CASE(2, 2); break; default: DUST_STREAM_CASE(2, 3); break; }
  } else {
    switch (mode) { case 0: DUST_STREAM_CASE(3, 0); break; case 1: DUST_STREAM_CASE(3, 1); break; case 2: DUST_STREAM_CASE(3, 2); break; default: DUST_STREAM_CASE(3, 3); break; }
  }
#undef DUST_STREAM_CASE
  return hipGetLastError();
}
hipError_t configure_gi_kernels(size_t max_lds) {  // (max_lds: what configure_kernels left after the build's static LDS)
  const void* fns[] = {
      (const void*)k_final_gather<0>, (const void*)k_final_gather<1>, (const void*)k_final_gather<2>, (const void*)k_final_gather<3>, (const void*)k_final_gather<4>, (const void*)k_final_gather<5>, (const void*)k_final_gather<6>, (const void*)k_final_gather<7>,
      (const void*)k_surfel_trace<0>, (const void*)k_surfel_trace<1>, (const void*)k_surfel_trace<2>, (const void*)k_surfel_trace<3>, (const void*)k_surfel_trace<4>, (const void*)k_surfel_trace<5>, (const void*)k_surfel_trace<6>, (const void*)k_surfel_trace<7>,
      (const void*)k_ray_walk<2, 0>, (const void*)k_ray_walk<2, 1>, (const void*)k_ray_walk<2, 2>, (const void*)k_ray_walk<2, 3>,
      (const void*)k_ray_walk<3, 0>, (const void*)k_ray_walk<3, 1>, (const void*)k_ray_walk<3, 2>, (const void*)k_ray_walk<3, 3>,
      (const void*)k_gather_rays, (const void*)k_surfel_rays};
  for (const void* f : fns) {
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_lds);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace dust

#ifdef DUST_PROFILE
extern "C" int dust_gi_profile_read(unsigned long long* out, int n) {  // profiling build only: this translation unit's section buckets, read and cleared
  unsigned long long h[dust::kProfBuckets] = {};
  if (hipMemcpyFromSymbol(h, HIP_SYMBOL(dust::g_prof_out), sizeof h) != hipSuccess) return -1;
  for (int i = 0; i < n && i < dust::kProfBuckets; ++i) out[i] = h[i];
  unsigned long long z[dust::kProfBuckets] = {};
  return hipMemcpyToSymbol(HIP_SYMBOL(dust::g_prof_out), z, sizeof z) == hipSuccess ? 0 : -1;
}
#endif
