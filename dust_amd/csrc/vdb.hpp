// vdb.hpp -- host-side sparse voxel tree builder (product code).
//
// Mirrors the interface of dust_vdb (reference crates/vdb/src): Tree::{new,set_value,get_value,iter,
// iter_leaf,accessor}, Pool, BitMask. The reference fixes the hierarchy at compile time with
// `hierarchy!(4,2,2)` (crates/vdb/src/node/mod.rs:112-123); the C ABI has to accept it at run time, so
// here a tree is described by a small table of per-level fan-outs and every node is a flat array of
// 64-bit words inside a chunked arena:
//   internal node: [child mask words][child index u32 x fan-out^3]      (node/internal.rs:22-32)
//   leaf node    : [occupancy words][active words][material_ptr u32]     (node/leaf.rs:13-25)
// Node handles are u32 arena indices, allocated in first-touch order exactly as Pool::alloc does
// (pool.rs:57-86), because iteration order and the accessor's cached path depend on nothing else.
#pragma once
#include <cstdint>
#include <cstring>
#include <memory>
#include <type_traits>
#include <vector>

namespace dust::vdb {

// BitMask::iter_set_bits (bitmask.rs:63-124): calls f(index) for every set bit, ascending.
template <class F>
inline void for_each_set_bit(const uint64_t* words, size_t n_words, F&& f) {
  for (size_t w = 0; w < n_words; ++w)
    for (uint64_t s = words[w]; s; s &= s - 1) f(static_cast<uint32_t>(w * 64 + __builtin_ctzll(s)));
}
inline bool bit_get(const uint64_t* words, size_t i) { return (words[i >> 6] >> (i & 63)) & 1; }
inline void bit_set(uint64_t* words, size_t i, bool v) {
  if (v) words[i >> 6] |= uint64_t(1) << (i & 63);
  else words[i >> 6] &= ~(uint64_t(1) << (i & 63));
}

// Pool (pool.rs:3-176): fixed-size items in chunks of 2^chunk_log2, LIFO free list, zeroed on alloc.
class Pool {
 public:
  Pool(size_t item_bytes, unsigned chunk_log2)
      : stride_((item_bytes + 7) / 8), chunk_log2_(chunk_log2) {}
  uint32_t alloc() {
    ++count_;
    uint32_t id;
    if (free_head_ != kNone) {
      id = free_head_;
      free_head_ = static_cast<uint32_t>(item(id)[0]);
    } else {
      id = top_++;
      if ((id >> chunk_log2_) >= chunks_.size())
        chunks_.emplace_back(new uint64_t[stride_ << chunk_log2_]());
    }
    std::memset(item(id), 0, stride_ * 8);
    return id;
  }
  void free(uint32_t id) {
    --count_;
    std::memset(item(id), 0, stride_ * 8);
    item(id)[0] = free_head_;
    free_head_ = id;
  }
  uint64_t* item(uint32_t id) { return chunks_[id >> chunk_log2_].get() + size_t(id & ((1u << chunk_log2_) - 1)) * stride_; }
  const uint64_t* item(uint32_t id) const { return const_cast<Pool*>(this)->item(id); }
  size_t num_chunks() const { return chunks_.size(); }
  uint32_t count() const { return count_; }
  bool owns(uint32_t id) const { return id < top_; }  // handed out by alloc() at some point (the C ABI's range check)

 private:
  static constexpr uint32_t kNone = 0xFFFFFFFFu;
  size_t stride_;  // words per item
  unsigned chunk_log2_;
  uint32_t free_head_ = kNone, top_ = 0, count_ = 0;
  std::vector<std::unique_ptr<uint64_t[]>> chunks_;
};

struct LeafRef {
  uint32_t origin[3];
  uint64_t occupancy;   // first 64 bits (the whole mask for 4^3 leaves)
  uint32_t* material_ptr;
};

class Tree {
 public:
  static constexpr int kMaxLevels = 8;
  // fan-out log2 per level, root first (hierarchy!(4,2,2) -> {4,2,2})
  Tree(const uint32_t* fanout_log2, int n_levels);

  int root_level() const { return n_levels_ - 1; }
  uint32_t extent_log2() const { return lv_[n_levels_ - 1].extent_log2; }
  uint32_t meta_mask() const;  // TreeMeta::META_MASK (tree.rs:154-167)

  // value: 1 Some(true), 0 Some(false); returns false for None on an internal path (reference: todo!())
  bool set(uint32_t x, uint32_t y, uint32_t z, int value) { return set_from(root_level(), 0, x, y, z, value, nullptr); }
  int get(uint32_t x, uint32_t y, uint32_t z) const { return get_from(root_level(), 0, x, y, z, nullptr); }

  template <class F> void for_each_voxel(F&& f) const { const_cast<Tree*>(this)->walk(root_level(), 0, 0, 0, 0, false, f); }
  // f(LeafRef&): depth-first, ascending child bits == Tree::iter_leaf order (tree.rs:106-124)
  template <class F> void for_each_leaf(F&& f) { walk(root_level(), 0, 0, 0, 0, true, f); }
  template <class F> void for_each_leaf(F&& f) const { const_cast<Tree*>(this)->walk(root_level(), 0, 0, 0, 0, true, f); }

  // Accessor (accessor.rs:5-57)
  class Accessor {
   public:
    explicit Accessor(const Tree& t) : tree_(t) { last_[0] = last_[1] = last_[2] = 0xFFFFFFFFu; }
    int get(uint32_t x, uint32_t y, uint32_t z);
    const Tree& tree() const { return tree_; }
   private:
    const Tree& tree_;
    uint32_t path_[kMaxLevels] = {};
    uint32_t last_[3];
  };
  static uint32_t lca_level(const uint32_t a[3], const uint32_t b[3], uint32_t mask, uint32_t root_level);

 private:
  struct Level {
    uint32_t fanout_log2, extent_log2, size, mask_words;
    size_t bytes;
  };
  uint64_t* node(int level, uint32_t id) { return level == root_level() ? root_.get() : pools_[level].item(id); }
  const uint64_t* node(int level, uint32_t id) const { return const_cast<Tree*>(this)->node(level, id); }
  uint32_t* children(int level, uint64_t* n) const { return reinterpret_cast<uint32_t*>(n + lv_[level].mask_words); }
  static uint32_t child_index(uint32_t f, uint32_t x, uint32_t y, uint32_t z) { return (x << (2 * f)) | (y << f) | z; }
  bool set_from(int level, uint32_t id, uint32_t x, uint32_t y, uint32_t z, int value, uint32_t* path);
  int get_from(int level, uint32_t id, uint32_t x, uint32_t y, uint32_t z, uint32_t* path) const;

  template <class F>
  void walk(int level, uint32_t id, uint32_t ox, uint32_t oy, uint32_t oz, bool leaves, F& f) {
    const Level& L = lv_[level];
    uint64_t* n = node(level, id);
    const uint32_t fo = L.fanout_log2, lo = (1u << fo) - 1;
    if (level == 0) {
      if constexpr (std::is_invocable_v<F&, LeafRef&>) {
        LeafRef r{{ox, oy, oz}, n[0], reinterpret_cast<uint32_t*>(n + 2 * L.mask_words)};
        f(r);
      } else {
        for_each_set_bit(n, L.mask_words, [&](uint32_t i) { f(ox + (i >> (2 * fo)), oy + ((i >> fo) & lo), oz + (i & lo)); });
      }
      (void)leaves;
      return;
    }
    const uint32_t cext = 1u << lv_[level - 1].extent_log2;
    uint32_t* ch = children(level, n);
    for_each_set_bit(n, L.mask_words, [&](uint32_t i) {
      walk(level - 1, ch[i], ox + (i >> (2 * fo)) * cext, oy + ((i >> fo) & lo) * cext, oz + (i & lo) * cext, leaves, f);
    });
  }

  int n_levels_;
  Level lv_[kMaxLevels];
  std::vector<Pool> pools_;  // [level], levels below the root (tree.rs:12,34-38); chunk = 1024 nodes
  std::unique_ptr<uint64_t[]> root_;
};

}  // namespace dust::vdb
