/*
 * vdb.c -- oracle restatement of crates/vdb (TEST INFRASTRUCTURE ONLY, see oracle.h).
 *
 * Follows: bitmask.rs:3-124, pool.rs:3-176, tree.rs:7-180, node/internal.rs:22-333,
 * node/leaf.rs:13-216, accessor.rs:5-139. The reference fixes the hierarchy at compile time
 * (`hierarchy!(4,2,2)`); here it is a run-time table of per-level fan-out log2s.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

/* ---------------------------------------------------------------- BitMask (bitmask.rs) */
/* bitmask.rs:52-61: word = index / 64, bit = index % 64, LSB first */
void orc_bitmask_set(uint64_t* w, size_t index, int val) {
  size_t i = index / 64, j = index - i * 64;
  if (val)
    w[i] |= (uint64_t)1 << j;
  else
    w[i] &= ~((uint64_t)1 << j);
}
int orc_bitmask_get(const uint64_t* w, size_t index) {
  size_t i = index / 64, j = index - i * 64;
  return (int)((w[i] >> j) & 1);
}
/* bitmask.rs:100-124 SetBitIterator: ascending within a word, words ascending */
size_t orc_bitmask_iter(const uint64_t* w, size_t nwords, uint32_t* out, size_t cap) {
  size_t n = 0;
  for (size_t i = 0; i < nwords; ++i) {
    uint64_t state = w[i];
    while (state) {
      uint64_t t = state & (~state + 1);
      unsigned r = (unsigned)__builtin_ctzll(state);
      if (n < cap) out[n] = (uint32_t)(i * 64 + r);
      ++n;
      state ^= t;
    }
  }
  return n;
}

/* ---------------------------------------------------------------- Pool (pool.rs) */
struct OrcPool {
  size_t item_size;
  uint32_t head; /* freelist head, UINT32_MAX = empty (pool.rs:46) */
  uint32_t top;
  unsigned chunk_log2;
  uint8_t** chunks;
  size_t n_chunks, cap_chunks;
  uint32_t count;
};

OrcPool* orc_pool_new(size_t item_size, unsigned chunk_size_log2) {
  OrcPool* p = (OrcPool*)calloc(1, sizeof(OrcPool));
  p->item_size = item_size;
  p->head = UINT32_MAX;
  p->chunk_log2 = chunk_size_log2;
  return p;
}
void orc_pool_free_pool(OrcPool* p) {
  if (!p) return;
  for (size_t i = 0; i < p->n_chunks; ++i) free(p->chunks[i]);
  free(p->chunks);
  free(p);
}
static uint8_t* pool_get(const OrcPool* p, uint32_t ptr) { /* pool.rs:108-116 */
  size_t chunk = (size_t)ptr >> p->chunk_log2;
  size_t item = (size_t)ptr & (((size_t)1 << p->chunk_log2) - 1);
  return p->chunks[chunk] + item * p->item_size;
}
/* pool.rs:57-86: bump allocation from zeroed chunks, LIFO freelist threaded through the first 4 bytes */
uint32_t orc_pool_alloc(OrcPool* p) {
  p->count += 1;
  if (p->head == UINT32_MAX) {
    uint32_t top = p->top;
    size_t chunk = (size_t)top >> p->chunk_log2;
    if (chunk >= p->n_chunks) {
      if (p->n_chunks == p->cap_chunks) {
        p->cap_chunks = p->cap_chunks ? p->cap_chunks * 2 : 8;
        p->chunks = (uint8_t**)realloc(p->chunks, p->cap_chunks * sizeof(uint8_t*));
      }
      p->chunks[p->n_chunks++] = (uint8_t*)calloc((size_t)1 << p->chunk_log2, p->item_size);
    }
    p->top += 1;
    /* Pool::alloc::<T> writes T::default() over the slot (pool.rs:57-63) */
    memset(pool_get(p, top), 0, p->item_size);
    return top;
  }
  uint32_t head = p->head;
  uint8_t* loc = pool_get(p, head);
  uint32_t next;
  memcpy(&next, loc, 4);
  p->head = next;
  memset(loc, 0, p->item_size);
  return head;
}
void orc_pool_free(OrcPool* p, uint32_t index) { /* pool.rs:87-102 */
  p->count -= 1;
  uint8_t* loc = pool_get(p, index);
  memset(loc, 0, p->item_size);
  memcpy(loc, &p->head, 4);
  p->head = index;
}
size_t orc_pool_num_chunks(const OrcPool* p) { return p->n_chunks; }
uint32_t orc_pool_count(const OrcPool* p) { return p->count; }

/* ---------------------------------------------------------------- Tree */
#define ORC_MAX_LEVELS 8

typedef struct LevelMeta {
  uint32_t fanout_log2;  /* per axis */
  uint32_t extent_log2;  /* per axis, of one node of this level */
  uint32_t size;         /* children (or voxels) per node */
  uint32_t nwords;       /* mask words */
  size_t node_bytes;
} LevelMeta;

struct OrcTree {
  int nlevels; /* root level = nlevels-1, leaf level = 0 */
  LevelMeta meta[ORC_MAX_LEVELS];
  OrcPool* pool[ORC_MAX_LEVELS]; /* [level], level < root level (tree.rs:12, :34-38) */
  uint8_t* root;                 /* the root node is owned by the tree (tree.rs:11) */
};

/* internal node memory: mask words, then child_ptrs u32[size] (internal.rs:22-32) */
static uint64_t* in_mask(uint8_t* n) { return (uint64_t*)n; }
static uint32_t* in_ptrs(const LevelMeta* m, uint8_t* n) { return (uint32_t*)(n + (size_t)m->nwords * 8); }
/* leaf memory: occupancy words, active words, material_ptr (leaf.rs:13-25) */
static uint64_t* lf_occ(uint8_t* n) { return (uint64_t*)n; }
static uint64_t* lf_act(const LevelMeta* m, uint8_t* n) { return (uint64_t*)(n + (size_t)m->nwords * 8); }
static uint32_t* lf_matptr(const LevelMeta* m, uint8_t* n) { return (uint32_t*)(n + (size_t)m->nwords * 16); }

OrcTree* orc_tree_new(const uint32_t* log2s, int nlevels) {
  if (nlevels < 1 || nlevels > ORC_MAX_LEVELS) return NULL;
  OrcTree* t = (OrcTree*)calloc(1, sizeof(OrcTree));
  t->nlevels = nlevels;
  uint32_t ext = 0;
  for (int L = 0; L < nlevels; ++L) {
    LevelMeta* m = &t->meta[L];
    m->fanout_log2 = log2s[nlevels - 1 - L];
    ext += m->fanout_log2;
    m->extent_log2 = ext;
    m->size = 1u << (3 * m->fanout_log2);
    m->nwords = (m->size + 63) / 64;
    if (L == 0)
      m->node_bytes = (size_t)m->nwords * 16 + 8; /* occupancy, active, material_ptr (+pad) */
    else
      m->node_bytes = (size_t)m->nwords * 8 + (size_t)m->size * 4;
  }
  for (int L = 0; L < nlevels - 1; ++L) t->pool[L] = orc_pool_new(t->meta[L].node_bytes, 10); /* tree.rs:36 */
  t->root = (uint8_t*)calloc(1, t->meta[nlevels - 1].node_bytes);
  return t;
}
void orc_tree_free(OrcTree* t) {
  if (!t) return;
  for (int L = 0; L < t->nlevels - 1; ++L) orc_pool_free_pool(t->pool[L]);
  free(t->root);
  free(t);
}
uint32_t orc_tree_root_level(const OrcTree* t) { return (uint32_t)(t->nlevels - 1); }

static uint8_t* node_at(const OrcTree* t, int level, uint32_t ptr) {
  if (level == t->nlevels - 1) return t->root;
  return pool_get(t->pool[level], ptr);
}

/* index of a child inside a node: internal.rs:78-81 / leaf.rs:81-83; x is the slowest axis */
static uint32_t child_index(uint32_t f, uint32_t x, uint32_t y, uint32_t z) { return (x << (2 * f)) | (y << f) | z; }

/* InternalNode::set -> LeafNode::set (internal.rs:97-130, leaf.rs:92-108). path: optional cached ptrs. */
static int tree_set_from(OrcTree* t, int level, uint32_t ptr, uint32_t x, uint32_t y, uint32_t z, int value,
                         uint32_t* path) {
  for (;;) {
    const LevelMeta* m = &t->meta[level];
    uint8_t* n = node_at(t, level, ptr);
    if (level == 0) {
      uint32_t idx = child_index(m->fanout_log2, x, y, z);
      if (value >= 0) {
        orc_bitmask_set(lf_occ(n), idx, 1);
        orc_bitmask_set(lf_act(m, n), idx, value);
      } else {
        orc_bitmask_set(lf_occ(n), idx, 0);
      }
      return 0;
    }
    uint32_t cl = t->meta[level - 1].extent_log2;
    uint32_t idx = child_index(m->fanout_log2, x >> cl, y >> cl, z >> cl);
    if (value < 0) return -2; /* internal.rs:121-124: clearing is todo!() in the reference */
    if (!orc_bitmask_get(in_mask(n), idx)) {
      orc_bitmask_set(in_mask(n), idx, 1);
      uint32_t np = orc_pool_alloc(t->pool[level - 1]);
      n = node_at(t, level, ptr); /* pool growth never moves nodes, but be explicit */
      in_ptrs(m, n)[idx] = np;
    }
    uint32_t cm = (1u << cl) - 1;
    x &= cm; y &= cm; z &= cm;
    ptr = in_ptrs(m, n)[idx];
    level -= 1;
    if (path) path[level] = ptr; /* get_in_pools/set_in_pools record cached_path[LEVEL] (internal.rs:138-140) */
  }
}

static int tree_get_from(const OrcTree* t, int level, uint32_t ptr, uint32_t x, uint32_t y, uint32_t z,
                         uint32_t* path) {
  for (;;) {
    const LevelMeta* m = &t->meta[level];
    uint8_t* n = node_at(t, level, ptr);
    if (level == 0) { /* leaf.rs:80-91 */
      uint32_t idx = child_index(m->fanout_log2, x, y, z);
      if (!orc_bitmask_get(lf_occ(n), idx)) return -1;
      return orc_bitmask_get(lf_act(m, n), idx);
    }
    uint32_t cl = t->meta[level - 1].extent_log2; /* internal.rs:77-95 */
    uint32_t idx = child_index(m->fanout_log2, x >> cl, y >> cl, z >> cl);
    if (!orc_bitmask_get(in_mask(n), idx)) return -1;
    uint32_t cm = (1u << cl) - 1;
    x &= cm; y &= cm; z &= cm;
    ptr = in_ptrs(m, n)[idx];
    level -= 1;
    if (path) path[level] = ptr;
  }
}

int orc_tree_set(OrcTree* t, uint32_t x, uint32_t y, uint32_t z, int value) { /* tree.rs:83-85 */
  return tree_set_from(t, t->nlevels - 1, 0, x, y, z, value, NULL);
}
int orc_tree_get(const OrcTree* t, uint32_t x, uint32_t y, uint32_t z) { /* tree.rs:78-80 */
  return tree_get_from(t, t->nlevels - 1, 0, x, y, z, NULL);
}

/* depth-first, ascending set bits at each level (internal.rs:253-287, leaf.rs:194-205) */
typedef struct IterCtx {
  const OrcTree* t;
  uint32_t* xyz;
  uint64_t* mask;
  uint32_t* matptr;
  size_t cap, n;
  int leaves; /* 0: voxels, 1: leaves */
  const uint32_t* set_ptrs; /* when non-NULL: write material_ptr instead of reading */
  size_t set_n;
} IterCtx;

static void iter_rec(IterCtx* c, int level, uint32_t ptr, uint32_t ox, uint32_t oy, uint32_t oz) {
  const OrcTree* t = c->t;
  const LevelMeta* m = &t->meta[level];
  uint8_t* n = node_at(t, level, ptr);
  uint32_t f = m->fanout_log2;
  if (level == 0) {
    if (c->leaves) { /* leaf.rs:156-164: once((offset, leaf)) */
      if (c->set_ptrs) {
        if (c->n < c->set_n) *lf_matptr(m, n) = c->set_ptrs[c->n];
      } else if (c->n < c->cap) {
        if (c->xyz) { c->xyz[c->n * 3] = ox; c->xyz[c->n * 3 + 1] = oy; c->xyz[c->n * 3 + 2] = oz; }
        if (c->mask) c->mask[c->n] = lf_occ(n)[0];
        if (c->matptr) c->matptr[c->n] = *lf_matptr(m, n);
      }
      c->n += 1;
      return;
    }
    for (uint32_t w = 0; w < m->nwords; ++w) {
      uint64_t s = lf_occ(n)[w];
      while (s) {
        uint32_t idx = w * 64 + (uint32_t)__builtin_ctzll(s);
        s &= s - 1;
        if (c->n < c->cap && c->xyz) {
          c->xyz[c->n * 3] = ox + (idx >> (2 * f));
          c->xyz[c->n * 3 + 1] = oy + ((idx >> f) & ((1u << f) - 1));
          c->xyz[c->n * 3 + 2] = oz + (idx & ((1u << f) - 1));
        }
        c->n += 1;
      }
    }
    return;
  }
  uint32_t cext = 1u << t->meta[level - 1].extent_log2;
  for (uint32_t w = 0; w < m->nwords; ++w) {
    uint64_t s = in_mask(n)[w];
    while (s) {
      uint32_t idx = w * 64 + (uint32_t)__builtin_ctzll(s);
      s &= s - 1;
      uint32_t cx = idx >> (2 * f), cy = (idx >> f) & ((1u << f) - 1), cz = idx & ((1u << f) - 1);
      iter_rec(c, level - 1, in_ptrs(m, n)[idx], ox + cx * cext, oy + cy * cext, oz + cz * cext);
    }
  }
}

size_t orc_tree_iter(const OrcTree* t, uint32_t* xyz, size_t cap) { /* tree.rs:102-104 */
  IterCtx c = {t, xyz, NULL, NULL, cap, 0, 0, NULL, 0};
  iter_rec(&c, t->nlevels - 1, 0, 0, 0, 0);
  return c.n;
}
size_t orc_tree_iter_leaf(const OrcTree* t, uint32_t* xyz, uint64_t* mask, uint32_t* material_ptr, size_t cap) {
  IterCtx c = {t, xyz, mask, material_ptr, cap, 0, 1, NULL, 0}; /* tree.rs:106-113 */
  iter_rec(&c, t->nlevels - 1, 0, 0, 0, 0);
  return c.n;
}
void orc_tree_set_leaf_material_ptrs(OrcTree* t, const uint32_t* ptrs, size_t n) { /* tree.rs:115-124 */
  IterCtx c = {t, NULL, NULL, NULL, 0, 0, 1, ptrs, n};
  iter_rec(&c, t->nlevels - 1, 0, 0, 0, 0);
}

uint32_t orc_tree_meta_mask(const OrcTree* t) { /* tree.rs:154-167 */
  uint32_t mask = 0;
  for (int L = 0; L < t->nlevels; ++L) mask |= 1u << (t->meta[L].extent_log2 - 1);
  return mask;
}

/* accessor.rs:15-30 */
static uint32_t lzcnt32(uint32_t v) { return v ? (uint32_t)__builtin_clz(v) : 32u; }
static uint32_t min_u32(uint32_t a, uint32_t b) { return a < b ? a : b; }
uint32_t orc_lca_level(const uint32_t a[3], const uint32_t b[3], uint32_t mask, uint32_t root_level) {
  uint32_t parent_index = 0xFFFFFFFFu;
  for (int i = 0; i < 3; ++i) {
    uint32_t diff = a[i] ^ b[i];
    uint32_t last_set_bit = 1u << (31 - min_u32(lzcnt32(diff), 31));
    uint32_t result = mask & ~(last_set_bit - 1);
    parent_index = min_u32(parent_index, (uint32_t)__builtin_popcount(result));
  }
  return root_level + 1 - parent_index;
}

struct OrcAccessor { /* accessor.rs:5-12 */
  const OrcTree* tree;
  uint32_t ptrs[ORC_MAX_LEVELS];
  uint32_t last[3];
};
OrcAccessor* orc_accessor_new(const OrcTree* t) { /* accessor.rs:125-131 */
  OrcAccessor* a = (OrcAccessor*)calloc(1, sizeof(OrcAccessor));
  a->tree = t;
  a->last[0] = a->last[1] = a->last[2] = UINT32_MAX;
  return a;
}
void orc_accessor_free(OrcAccessor* a) { free(a); }
int orc_accessor_get(OrcAccessor* a, uint32_t x, uint32_t y, uint32_t z) { /* accessor.rs:37-57 */
  const OrcTree* t = a->tree;
  uint32_t c[3] = {x, y, z};
  uint32_t root_level = (uint32_t)(t->nlevels - 1);
  uint32_t lca = orc_lca_level(a->last, c, orc_tree_meta_mask(t), root_level);
  a->last[0] = x; a->last[1] = y; a->last[2] = z;
  if (lca >= root_level) return tree_get_from(t, (int)root_level, 0, x, y, z, a->ptrs);
  uint32_t em = (1u << t->meta[lca].extent_log2) - 1;
  return tree_get_from(t, (int)lca, a->ptrs[lca], x & em, y & em, z & em, a->ptrs);
}
