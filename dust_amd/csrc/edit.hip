// edit.hip -- device-side voxel edits and tree (re)build for hierarchy!(4,2,2) models.
//
// The reference's VoxGeometry::set/get (crates/vox/src/geometry.rs:180-185) edit the CPU tree only -- the GPU buffers of a
// loaded model never change -- and Tree::set_value (crates/vdb/src/tree.rs:83-85) is a pointer-chasing insert per voxel.
// Here an edited model keeps its voxels as a dense grid ON THE DEVICE (one byte per voxel, palette index + 1, laid out
// brick-major in Tree::iter_leaf order: 16 MiB for a 256^3 model, created from the model's own blocks the first time it is
// edited); an edit batch is a scatter of bytes into that grid followed by a full rebuild of every array the traversal and
// shading kernels read, with data-parallel scans instead of per-voxel inserts:
//   k_edit_brick_masks   one thread per 4^3 brick of the lattice: occupancy mask + voxel count, in both orders that matter
//                        (iter_leaf order for block indices, the collector's block-major order for material_ptr)
//   scan                 exclusive prefix sums of the two 262 144-entry tables
//   k_edit_emit_blocks   Block records (position, mask, material_ptr, avg_albedo) + the compacted material stream
//   k_edit_upper_levels  root mask / rank prefixes, mid nodes, dense_mask, tight bounds, counts (one workgroup)
// The result is, array for array, what dust_hip_model_create builds on the host from the same voxels
// (loader.rs:244-274 + collector.rs + geometry.rs:68-128 + capi.cpp build_hierarchy): tests compare them byte for byte.
// avg_albedo's linear->sRGB curve is a table the HOST evaluates (the same powf the host-side flatten calls), indexed by
// (voxel count, colour sum): the device only adds and looks up, so not one bit depends on a device transcendental.
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "edit.hpp"

namespace dust {

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

namespace {

// brick code in Tree::iter_leaf order: root child index (x>>4)<<8 | (y>>4)<<4 | (z>>4) (node/internal.rs:78-81), then the
// mid node's child bit ((x>>2)&3)<<4 | ((y>>2)&3)<<2 | ((z>>2)&3); bx, by, bz are brick coordinates (voxel >> 2)
__device__ __forceinline__ uint32_t leaf_code(uint32_t bx, uint32_t by, uint32_t bz) {
  return ((((bx >> 2) << 8) | ((by >> 2) << 4) | (bz >> 2)) << 6) | ((bx & 3u) << 4) | ((by & 3u) << 2) | (bz & 3u);
}
__device__ __forceinline__ void leaf_decode(uint32_t code, uint32_t& bx, uint32_t& by, uint32_t& bz) {
  const uint32_t r = code >> 6, c = code & 63u;
  bx = ((r >> 8) << 2) | (c >> 4);
  by = (((r >> 4) & 15u) << 2) | ((c >> 2) & 3u);
  bz = ((r & 15u) << 2) | (c & 3u);
}
// collector.rs:25-27: block_index = bx + 64 by + 4096 bz -- the order the material stream is compacted in
__device__ __forceinline__ uint32_t major_code(uint32_t bx, uint32_t by, uint32_t bz) { return bx + 64u * by + 4096u * bz; }

}  // namespace

// first edit of a model: its blocks + material stream -> the dense grid (which the caller zeroed)
__global__ void k_edit_expand(EditArgs e, const DustHipBlock* blocks, const uint8_t* materials, uint32_t n_blocks) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if ((i >> 6) >= n_blocks) return;
  const DustHipBlock b = blocks[i >> 6];
  const uint32_t bit = i & 63u;
  if (!((b.mask >> bit) & 1ull)) return;
  const uint32_t rank = (uint32_t)__popcll(b.mask & ((1ull << bit) - 1ull));
  e.grid[(size_t)leaf_code(b.x >> 2, b.y >> 2, b.z >> 2) * 64 + bit] = (uint8_t)(materials[b.material_ptr + rank] + 1u);
}

// VoxGeometry::set: value >= 0 -> Some(true) with that palette index, value < 0 -> None
__global__ void k_edit_apply(EditArgs e) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e.n_edits) return;
  const uint32_t x = e.xyz[i * 3], y = e.xyz[i * 3 + 1], z = e.xyz[i * 3 + 2];
  const int32_t v = e.values[i];
  e.grid[(size_t)leaf_code(x >> 2, y >> 2, z >> 2) * 64 + (((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u))] = v < 0 ? 0u : (uint8_t)(v + 1);
}
// VoxGeometry::get
__global__ void k_edit_read(EditArgs e) {
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= e.n_edits) return;
  const uint32_t x = e.xyz[i * 3], y = e.xyz[i * 3 + 1], z = e.xyz[i * 3 + 2];
  e.values_out[i] = (int32_t)e.grid[(size_t)leaf_code(x >> 2, y >> 2, z >> 2) * 64 + (((x & 3u) << 4) | ((y & 3u) << 2) | (z & 3u))] - 1;
}

__global__ void __launch_bounds__(256) k_edit_brick_masks(EditArgs e) {
  const uint32_t code = blockIdx.x * blockDim.x + threadIdx.x;  // iter_leaf order
  const u32x4_t* cells = reinterpret_cast<const u32x4_t*>(e.grid + (size_t)code * 64);
  uint64_t mask = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4_t v = cells[q];
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 16; ++k)
      if ((w[k >> 2] >> ((k & 3) * 8)) & 0xFFu) mask |= 1ull << (q * 16 + k);
  }
  e.brick_mask[code] = mask;
  e.flag_leaf[code] = mask != 0 ? 1u : 0u;
  uint32_t bx, by, bz;
  leaf_decode(code, bx, by, bz);
  e.count_major[major_code(bx, by, bz)] = (uint32_t)__popcll(mask);
}

// exclusive scan of kLattice counters, 1024 per workgroup: (1) local scan + block sum, (2) scan of the 256 sums, (3) add back.
// `which`: 0 = flag_leaf, 1 = count_major
__global__ void __launch_bounds__(256) k_edit_scan_local(EditArgs e) {
  __shared__ uint32_t sums[256];
  const uint32_t which = blockIdx.y;
  uint32_t* v = which ? e.count_major : e.flag_leaf;
  const uint32_t base = blockIdx.x * 1024u + threadIdx.x * 4u;
  const uint32_t a0 = v[base], a1 = v[base + 1], a2 = v[base + 2], a3 = v[base + 3];
  const uint32_t s = a0 + a1 + a2 + a3;
  sums[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 256u; d <<= 1) {
    const uint32_t x = threadIdx.x >= d ? sums[threadIdx.x - d] : 0u;
    __syncthreads();
    sums[threadIdx.x] += x;
    __syncthreads();
  }
  const uint32_t before = sums[threadIdx.x] - s;
  v[base] = before; v[base + 1] = before + a0; v[base + 2] = before + a0 + a1; v[base + 3] = before + a0 + a1 + a2;
  if (threadIdx.x == 255) e.scan_tmp[which * 256 + blockIdx.x] = sums[255];
}
__global__ void __launch_bounds__(256) k_edit_scan_sums(EditArgs e) {
  __shared__ uint32_t sums[256];
  uint32_t* t = e.scan_tmp + blockIdx.x * 256;
  const uint32_t s = t[threadIdx.x];
  sums[threadIdx.x] = s;
  __syncthreads();
  for (uint32_t d = 1; d < 256u; d <<= 1) {
    const uint32_t x = threadIdx.x >= d ? sums[threadIdx.x - d] : 0u;
    __syncthreads();
    sums[threadIdx.x] += x;
    __syncthreads();
  }
  t[threadIdx.x] = sums[threadIdx.x] - s;
  if (threadIdx.x == 255) {
    if (blockIdx.x == 0) e.header->n_blocks = sums[255];
    else e.header->n_materials = sums[255];
  }
}

// Block records + material stream (geometry.rs:68-128, collector.rs:50-60): one thread per brick of the lattice
__global__ void __launch_bounds__(256) k_edit_emit_blocks(EditArgs e) {
  const uint32_t code = blockIdx.x * blockDim.x + threadIdx.x;
  const uint64_t mask = e.brick_mask[code];
  if (mask == 0) return;
  uint32_t bx, by, bz;
  leaf_decode(code, bx, by, bz);
  const uint32_t major = major_code(bx, by, bz);
  const uint32_t index = e.flag_leaf[code] + e.scan_tmp[code >> 10];                   // exclusive scans: local + block base
  const uint32_t ptr = e.count_major[major] + e.scan_tmp[256 + (major >> 10)];
  const uint8_t* cells = e.grid + (size_t)code * 64;
  uint32_t sum[4] = {0, 0, 0, 0}, n = 0;
  for (uint64_t m = mask; m; m &= m - 1ull) {  // ascending bit index: the collector's order inside a block
    const uint32_t bit = (uint32_t)__builtin_ctzll(m);
    const uint32_t pal = (uint32_t)cells[bit] - 1u;
    e.materials[ptr + n] = (uint8_t)pal;
    const uint32_t c = e.palette[pal];
    sum[0] += c & 255u; sum[1] += (c >> 8) & 255u; sum[2] += (c >> 16) & 255u; sum[3] += c >> 24;
    ++n;
  }
  const uint16_t* row = e.srgb_lut + (size_t)(n - 1u) * kSrgbRow;
  const float alpha = (float)sum[3] / ((float)n * 255.0f);  // geometry.rs:97,109: no transfer curve on alpha
  DustHipBlock b;
  b.x = (uint16_t)(bx * 4u); b.y = (uint16_t)(by * 4u); b.z = (uint16_t)(bz * 4u); b.w = 0;
  b.mask = mask;
  b.material_ptr = ptr;
  b.avg_albedo = ((uint32_t)row[sum[0]] << 22) | ((uint32_t)row[sum[1]] << 12) | ((uint32_t)row[sum[2]] << 2) | (uint32_t)(alpha * 3.0f);
  e.blocks[index] = b;
}

// root node, mid nodes, dense_mask, bounds: one workgroup of 1024 threads, four root cells each (capi.cpp build_hierarchy)
__global__ void __launch_bounds__(1024) k_edit_upper_levels(EditArgs e) {
  __shared__ uint32_t part[1024];
  __shared__ uint64_t root_mask[64];
  __shared__ int lo[3], hi[3];
  if (threadIdx.x < 64) root_mask[threadIdx.x] = 0;
  if (threadIdx.x < 3) { lo[threadIdx.x] = 1 << 30; hi[threadIdx.x] = -1; }
  __syncthreads();
  uint64_t child[4];
  uint32_t mine = 0;
  for (int k = 0; k < 4; ++k) {
    const uint32_t cell = threadIdx.x * 4u + k;  // root child index (x>>4)<<8 | (y>>4)<<4 | (z>>4)
    uint64_t m = 0;
    for (uint32_t c = 0; c < 64; ++c) m |= (uint64_t)(e.brick_mask[cell * 64u + c] != 0) << c;
    child[k] = m;
    if (m) {
      mine += 1;
      atomicOr((unsigned long long*)&root_mask[cell >> 6], 1ull << (cell & 63u));
      for (uint64_t mm = m; mm; mm &= mm - 1ull) {  // tight bounds over the bricks
        uint32_t bx, by, bz;
        leaf_decode(cell * 64u + (uint32_t)__builtin_ctzll(mm), bx, by, bz);
        atomicMin(&lo[0], (int)bx * 4); atomicMin(&lo[1], (int)by * 4); atomicMin(&lo[2], (int)bz * 4);
        atomicMax(&hi[0], (int)bx * 4 + 4); atomicMax(&hi[1], (int)by * 4 + 4); atomicMax(&hi[2], (int)bz * 4 + 4);
      }
    }
  }
  part[threadIdx.x] = mine;
  __syncthreads();
  for (uint32_t d = 1; d < 1024u; d <<= 1) {
    const uint32_t x = threadIdx.x >= d ? part[threadIdx.x - d] : 0u;
    __syncthreads();
    part[threadIdx.x] += x;
    __syncthreads();
  }
  uint32_t mi = part[threadIdx.x] - mine;  // mid nodes are stored in root-child order == depth first
  for (int k = 0; k < 4; ++k) {
    if (!child[k]) continue;
    const uint32_t cell = threadIdx.x * 4u + k;
    const uint32_t first_code = cell * 64u + (uint32_t)__builtin_ctzll(child[k]);
    DevN4 n;
    n.mask_lo = (uint32_t)child[k]; n.mask_hi = (uint32_t)(child[k] >> 32);
    n.first_block = e.flag_leaf[first_code] + e.scan_tmp[first_code >> 10];
    n.pad = 0;
    e.mid[mi] = n;
    for (uint32_t c = 0; c < 64; ++c) e.dense_mask[(size_t)mi * 64 + c] = e.brick_mask[cell * 64u + c];
    ++mi;
  }
  if (threadIdx.x < 64) {  // the N16 node: 64 mask words, 64 u16 rank prefixes, child_base 0
    reinterpret_cast<uint64_t*>(e.root)[threadIdx.x] = root_mask[threadIdx.x];
    uint32_t run = 0;
    for (uint32_t w = 0; w < threadIdx.x; ++w) run += (uint32_t)__popcll(root_mask[w]);
    reinterpret_cast<uint16_t*>(e.root + 512)[threadIdx.x] = (uint16_t)run;
  }
  if (threadIdx.x == 0) {
    *reinterpret_cast<uint32_t*>(e.root + 640) = 0u;
    e.header->n_mid = part[1023];
    const bool any = part[1023] != 0;
    for (int a = 0; a < 3; ++a) { e.header->bmin[a] = any ? (float)lo[a] : 0.0f; e.header->bmax[a] = any ? (float)hi[a] : 0.0f; }
  }
}

// ------------------------------------------------------------------ launchers (capi.cpp)
hipError_t launch_edit_expand(const EditArgs& e, const DustHipBlock* blocks, const uint8_t* materials, uint32_t n_blocks, hipStream_t s) {
  if (n_blocks) hipLaunchKernelGGL(k_edit_expand, dim3((n_blocks * 64u + 255u) / 256u), dim3(256), 0, s, e, blocks, materials, n_blocks);
  return hipGetLastError();
}
hipError_t launch_edit_apply(const EditArgs& e, bool read, hipStream_t s) {
  if (e.n_edits == 0) return hipSuccess;
  if (read) hipLaunchKernelGGL(k_edit_read, dim3((e.n_edits + 255u) / 256u), dim3(256), 0, s, e);
  else hipLaunchKernelGGL(k_edit_apply, dim3((e.n_edits + 255u) / 256u), dim3(256), 0, s, e);
  return hipGetLastError();
}
hipError_t launch_edit_rebuild(const EditArgs& e, hipStream_t s) {
  hipLaunchKernelGGL(k_edit_brick_masks, dim3(kLattice / 256), dim3(256), 0, s, e);
  hipLaunchKernelGGL(k_edit_scan_local, dim3(256, 2), dim3(256), 0, s, e);
  hipLaunchKernelGGL(k_edit_scan_sums, dim3(2), dim3(256), 0, s, e);
  hipLaunchKernelGGL(k_edit_emit_blocks, dim3(kLattice / 256), dim3(256), 0, s, e);
  hipLaunchKernelGGL(k_edit_upper_levels, dim3(1), dim3(1024), 0, s, e);
  return hipGetLastError();
}

}  // namespace dust
