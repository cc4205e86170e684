// vdb.cpp -- host-side sparse voxel tree builder (product code); see vdb.hpp.
#include "vdb.hpp"

#include <algorithm>
#include <stdexcept>

namespace dust::vdb {

Tree::Tree(const uint32_t* fanout_log2, int n_levels) : n_levels_(n_levels) {
  if (n_levels < 1 || n_levels > kMaxLevels) throw std::invalid_argument("hierarchy must have 1..8 levels");
  uint32_t ext = 0;
  for (int L = 0; L < n_levels; ++L) {  // level 0 = leaf, root = n_levels-1
    Level& lv = lv_[L];
    lv.fanout_log2 = fanout_log2[n_levels - 1 - L];
    if (lv.fanout_log2 < 1 || lv.fanout_log2 > 6) throw std::invalid_argument("fan-out log2 must be 1..6");
    ext += lv.fanout_log2;
    if (ext > 31) throw std::invalid_argument("tree extent exceeds 2^31");
    lv.extent_log2 = ext;
    lv.size = 1u << (3 * lv.fanout_log2);
    lv.mask_words = (lv.size + 63) / 64;
    lv.bytes = L == 0 ? size_t(lv.mask_words) * 16 + 8 : size_t(lv.mask_words) * 8 + size_t(lv.size) * 4;
  }
  pools_.reserve(n_levels);
  for (int L = 0; L + 1 < n_levels; ++L) pools_.emplace_back(lv_[L].bytes, 10);  // tree.rs:36
  const size_t root_words = (lv_[n_levels - 1].bytes + 7) / 8;
  root_.reset(new uint64_t[root_words]());
}

uint32_t Tree::meta_mask() const {
  uint32_t m = 0;
  for (int L = 0; L < n_levels_; ++L) m |= 1u << (lv_[L].extent_log2 - 1);
  return m;
}

// InternalNode::set / LeafNode::set (node/internal.rs:97-130, node/leaf.rs:92-108), iteratively.
bool Tree::set_from(int level, uint32_t id, uint32_t x, uint32_t y, uint32_t z, int value, uint32_t* path) {
  while (level > 0) {
    if (value < 0) return false;  // clearing through an internal node is todo!() in the reference
    const Level& L = lv_[level];
    const uint32_t cl = lv_[level - 1].extent_log2;
    const uint32_t idx = child_index(L.fanout_log2, x >> cl, y >> cl, z >> cl);
    uint64_t* n = node(level, id);
    uint32_t* ch = children(level, n);
    if (!bit_get(n, idx)) {
      bit_set(n, idx, true);
      ch[idx] = pools_[level - 1].alloc();
    }
    const uint32_t cm = (1u << cl) - 1;
    x &= cm; y &= cm; z &= cm;
    id = ch[idx];
    --level;
    if (path) path[level] = id;
  }
  const Level& L = lv_[0];
  uint64_t* n = node(0, id);
  const uint32_t idx = child_index(L.fanout_log2, x, y, z);
  if (value >= 0) {
    bit_set(n, idx, true);
    bit_set(n + L.mask_words, idx, value != 0);
  } else {
    bit_set(n, idx, false);
  }
  return true;
}

// InternalNode::get / LeafNode::get (node/internal.rs:77-95, node/leaf.rs:80-91)
int Tree::get_from(int level, uint32_t id, uint32_t x, uint32_t y, uint32_t z, uint32_t* path) const {
  while (level > 0) {
    const Level& L = lv_[level];
    const uint32_t cl = lv_[level - 1].extent_log2;
    const uint32_t idx = child_index(L.fanout_log2, x >> cl, y >> cl, z >> cl);
    const uint64_t* n = node(level, id);
    if (!bit_get(n, idx)) return -1;
    const uint32_t cm = (1u << cl) - 1;
    x &= cm; y &= cm; z &= cm;
    id = children(level, const_cast<uint64_t*>(n))[idx];
    --level;
    if (path) path[level] = id;
  }
  const Level& L = lv_[0];
  const uint64_t* n = node(0, id);
  const uint32_t idx = child_index(L.fanout_log2, x, y, z);
  if (!bit_get(n, idx)) return -1;
  return bit_get(n + L.mask_words, idx) ? 1 : 0;
}

// lowest_common_ancestor_level (accessor.rs:15-30)
uint32_t Tree::lca_level(const uint32_t a[3], const uint32_t b[3], uint32_t mask, uint32_t root_level) {
  uint32_t parent_index = 0xFFFFFFFFu;
  for (int i = 0; i < 3; ++i) {
    const uint32_t diff = a[i] ^ b[i];
    const uint32_t lz = diff ? uint32_t(__builtin_clz(diff)) : 32u;
    const uint32_t last_set_bit = 1u << (31 - std::min(lz, 31u));
    parent_index = std::min(parent_index, uint32_t(__builtin_popcount(mask & ~(last_set_bit - 1))));
  }
  return root_level + 1 - parent_index;
}

// Accessor::get (accessor.rs:37-57)
int Tree::Accessor::get(uint32_t x, uint32_t y, uint32_t z) {
  const uint32_t c[3] = {x, y, z};
  const uint32_t root = uint32_t(tree_.root_level());
  const uint32_t lca = lca_level(last_, c, tree_.meta_mask(), root);
  last_[0] = x; last_[1] = y; last_[2] = z;
  if (lca >= root) return tree_.get_from(int(root), 0, x, y, z, path_);
  const uint32_t em = (1u << tree_.lv_[lca].extent_log2) - 1;
  return tree_.get_from(int(lca), path_[lca], x & em, y & em, z & em, path_);
}

}  // namespace dust::vdb
