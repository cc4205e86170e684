/*
 * flatten.c -- oracle restatement of the .vox model -> GPU block flattening (TEST INFRASTRUCTURE ONLY).
 *
 * Follows crates/vox/src/loader.rs:238-308 (load_model), collector.rs:2-88 (ModelIndexCollector),
 * geometry.rs:55-179 (VoxGeometry::from_tree). Parity vs the real reference output is UNPINNED
 * (the reference holds no test for the loader); the vdb calls underneath are pinned (vdb.c).
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* geometry.rs:99-105 */
static float linear2srgb(float c) {
  if (c <= 0.0031308f) return 12.92f * c;
  return 1.055f * powf(c, 1.0f / 2.4f) - 0.055f;
}

OrcModel* orc_model_build(const uint8_t* xyzi, size_t n_voxels, const uint32_t size[3],
                          const uint8_t* palette_rgba256, const uint32_t* log2s, int nlevels) {
  OrcTree* tree = orc_tree_new(log2s, nlevels);
  if (!tree) return NULL;
  /* collector.rs:9-21: dense 256^3 grid in block-major order + per-block counts */
  uint8_t* grid = (uint8_t*)calloc(256u * 256u * 256u, 1);
  uint32_t* block_counts = (uint32_t*)calloc(64u * 64u * 64u, sizeof(uint32_t));
  size_t count = 0;
  for (size_t i = 0; i < n_voxels; ++i) {
    /* loader.rs:247-253: MagicaVoxel (x,y,z) -> engine (x, z, size.y - 1 - y) */
    uint8_t vx = xyzi[i * 4 + 0];
    uint8_t vy = xyzi[i * 4 + 2];
    uint8_t vz = (uint8_t)(size[1] - (uint32_t)xyzi[i * 4 + 1] - 1);
    uint8_t vi = xyzi[i * 4 + 3];
    orc_tree_set(tree, vx, vy, vz, 1); /* loader.rs:259 */
    /* collector.rs:23-34 (duplicate XYZI entries are counted twice, as in the reference) */
    count += 1;
    size_t block_index = (size_t)(vx >> 2) + (size_t)(vy >> 2) * 64 + (size_t)(vz >> 2) * 64 * 64;
    block_counts[block_index] += 1;
    unsigned index = (vz & 3u) | ((vy & 3u) << 2) | ((vx & 3u) << 4);
    grid[block_index * 64 + index] = (uint8_t)(vi + 1);
  }
  /* collector.rs:76-87: exclusive prefix sum over blocks */
  uint32_t sum = 0;
  for (size_t i = 0; i < 64u * 64u * 64u; ++i) {
    uint32_t v = block_counts[i];
    block_counts[i] = sum;
    sum += v;
  }
  /* loader.rs:265-272: leaf.material_ptr = running_sum[block] in iter_leaf order */
  size_t n_leaf = orc_tree_iter_leaf(tree, NULL, NULL, NULL, 0);
  uint32_t* leaf_xyz = (uint32_t*)malloc((n_leaf ? n_leaf : 1) * 3 * sizeof(uint32_t));
  uint64_t* leaf_mask = (uint64_t*)malloc((n_leaf ? n_leaf : 1) * sizeof(uint64_t));
  uint32_t* leaf_ptr = (uint32_t*)malloc((n_leaf ? n_leaf : 1) * sizeof(uint32_t));
  orc_tree_iter_leaf(tree, leaf_xyz, leaf_mask, NULL, n_leaf);
  for (size_t i = 0; i < n_leaf; ++i) {
    size_t bi = (size_t)(leaf_xyz[i * 3] >> 2) + (size_t)(leaf_xyz[i * 3 + 1] >> 2) * 64 +
                (size_t)(leaf_xyz[i * 3 + 2] >> 2) * 64 * 64;
    leaf_ptr[i] = block_counts[bi];
  }
  orc_tree_set_leaf_material_ptrs(tree, leaf_ptr, n_leaf);

  OrcModel* m = (OrcModel*)calloc(1, sizeof(OrcModel));
  /* collector.rs:50-60: compaction of the dense grid, zero-based indices; the iterator's len() is
   * `count` (collector.rs:66-70) but it yields one item per non-zero grid cell */
  m->materials = (uint8_t*)malloc(count ? count : 1);
  uint64_t nm = 0;
  for (size_t i = 0; i < 256u * 256u * 256u; ++i)
    if (grid[i]) {
      if (nm < count) m->materials[nm] = (uint8_t)(grid[i] - 1);
      ++nm;
    }
  m->n_materials = nm;
  memcpy(m->palette, palette_rgba256, 255 * 4); /* loader.rs:214-218 */
  uint32_t ext_log2 = 0;
  for (int i = 0; i < nlevels; ++i) ext_log2 += log2s[i];
  m->extent = 1u << ext_log2;

  /* geometry.rs:68-128 */
  m->n_blocks = (uint32_t)n_leaf;
  m->blocks = (OrcBlock*)calloc(n_leaf ? n_leaf : 1, sizeof(OrcBlock));
  for (size_t i = 0; i < n_leaf; ++i) {
    uint64_t mask = leaf_mask[i];
    uint32_t nvox = (uint32_t)__builtin_popcountll(mask);
    uint32_t cr = 0, cg = 0, cb = 0, ca = 0;
    for (uint32_t k = 0; k < nvox; ++k) {
      uint8_t pi = m->materials[leaf_ptr[i] + k];
      const uint8_t* c = palette_rgba256 + (size_t)pi * 4;
      cr += c[0]; cg += c[1]; cb += c[2]; ca += c[3];
    }
    float denom = (float)nvox * 255.0f;
    float fr = (float)cr / denom, fg = (float)cg / denom, fb = (float)cb / denom, fa = (float)ca / denom;
    fr = linear2srgb(fr); fg = linear2srgb(fg); fb = linear2srgb(fb);
    uint32_t r = (uint32_t)(fr * 1023.0f), g = (uint32_t)(fg * 1023.0f), b = (uint32_t)(fb * 1023.0f);
    uint32_t a = (uint32_t)(fa * 3.0f);
    OrcBlock* blk = &m->blocks[i];
    blk->x = (uint16_t)leaf_xyz[i * 3];
    blk->y = (uint16_t)leaf_xyz[i * 3 + 1];
    blk->z = (uint16_t)leaf_xyz[i * 3 + 2];
    blk->w = 0;
    blk->mask = mask;
    blk->material_ptr = leaf_ptr[i];
    blk->avg_albedo = (r << 22) | (g << 12) | (b << 2) | a;
  }
  free(grid); free(block_counts); free(leaf_xyz); free(leaf_mask); free(leaf_ptr);
  orc_tree_free(tree);
  return m;
}

void orc_model_free(OrcModel* m) {
  if (!m) return;
  free(m->blocks);
  free(m->materials);
  free(m);
}
