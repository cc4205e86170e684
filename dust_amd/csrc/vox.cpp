// vox.cpp -- MagicaVoxel .vox loader and model flattening (product host code); see vox.hpp.
#include "vox.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <thread>

#include "vdb.hpp"

namespace dust::vox {
namespace {

// ------------------------------------------------------------------ chunk reader
struct Reader {
  const uint8_t* p;
  size_t n, pos = 0;
  bool has(size_t k) const { return pos + k <= n; }
  uint32_t u32() {
    if (!has(4)) throw ParseError{"unexpected end of file"};
    uint32_t v;
    std::memcpy(&v, p + pos, 4);
    pos += 4;
    return v;
  }
  int32_t i32() { return static_cast<int32_t>(u32()); }
  std::string str() {
    const uint32_t len = u32();
    if (!has(len)) throw ParseError{"string runs past end of chunk"};
    std::string s(reinterpret_cast<const char*>(p + pos), len);
    pos += len;
    return s;
  }
  std::map<std::string, std::string> dict() {
    std::map<std::string, std::string> d;
    const uint32_t k = u32();
    for (uint32_t i = 0; i < k; ++i) {
      std::string key = str();
      d[key] = str();
    }
    return d;
  }
};

struct Rotation {  // the "_r" byte of an nTRN frame: a signed permutation matrix, row-major
  int m[3][3];
  static Rotation identity() { return from_byte(0b0000100); }
  static Rotation from_byte(uint8_t b) {
    Rotation r{};
    const int i0 = b & 3, i1 = (b >> 2) & 3;
    if (i0 > 2 || i1 > 2 || i0 == i1) throw ParseError{"invalid rotation byte in nTRN frame"};
    const int i2 = 3 - i0 - i1;
    r.m[0][i0] = (b & 0x10) ? -1 : 1;
    r.m[1][i1] = (b & 0x20) ? -1 : 1;
    r.m[2][i2] = (b & 0x40) ? -1 : 1;
    return r;
  }
};

struct Keyframe {  // one frame dictionary of an nTRN node: {_t, _r, _f}
  uint32_t frame = 0;  // "_f": the animation frame this key applies from (0 when absent)
  int32_t t[3] = {0, 0, 0};
  Rotation rot = Rotation::identity();
};
struct ShapeModel {  // one (model id, attributes) pair of an nSHP node; "_f": the frame the model is shown from
  uint32_t model = 0, frame = 0;
};
struct Node {
  enum Kind { kNone, kTransform, kGroup, kShape } kind = kNone;
  // transform
  uint32_t child = 0;
  std::vector<Keyframe> frames;
  // group
  std::vector<uint32_t> children;
  // shape
  std::vector<ShapeModel> models;
};
// The entry in force at animation frame `frame`: the one with the largest "_f" <= frame, or the first when the
// animation starts later. (MagicaVoxel holds a key until the next one.) The reference stops at unimplemented!()
// for nodes with more than one entry (loader.rs:103-105,149-151); with one entry this is that entry, whatever the frame.
template <class T>
const T& in_force(const std::vector<T>& v, uint32_t frame) {
  const T* best = &v[0];
  bool found = v[0].frame <= frame;
  for (const T& e : v)
    if (e.frame <= frame && (!found || e.frame >= best->frame)) { best = &e; found = true; }
  return *best;
}

// 4x4 affine, row-major 3x4 kept as doubles while composing
struct Affine {
  double m[3][4];
  static Affine identity() {
    Affine a{};
    for (int i = 0; i < 3; ++i) a.m[i][i] = 1.0;
    return a;
  }
  Affine operator*(const Affine& b) const {
    Affine r{};
    for (int i = 0; i < 3; ++i) {
      for (int j = 0; j < 4; ++j) {
        double s = 0.0;
        for (int k = 0; k < 3; ++k) s += m[i][k] * b.m[k][j];
        r.m[i][j] = s + (j == 3 ? m[i][3] : 0.0);
      }
    }
    return r;
  }
};

// SceneGraphTraverser::to_transform (loader.rs:178-204), written with matrices instead of
// quaternion+scale: file axes (x,y,z) map to engine axes (x, z, -y); a signed permutation M in file
// axes becomes P M P^-1, dot_vox splits it into rotation * uniform scale s = det(M) = +-1.
Affine to_transform(const int32_t t[3], const Rotation& rot, const uint32_t size[3]) {
  static const int P[3][3] = {{1, 0, 0}, {0, 0, 1}, {0, -1, 0}};  // engine = P * file
  int PM[3][3] = {}, Me[3][3] = {};
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) PM[i][j] += P[i][k] * rot.m[k][j];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) Me[i][j] += PM[i][k] * P[j][k];  // * P^T == P^-1
  const int det = Me[0][0] * (Me[1][1] * Me[2][2] - Me[1][2] * Me[2][1]) -
                  Me[0][1] * (Me[1][0] * Me[2][2] - Me[1][2] * Me[2][0]) +
                  Me[0][2] * (Me[1][0] * Me[2][1] - Me[1][1] * Me[2][0]);
  const double te[3] = {double(t[0]), double(t[2]), -double(t[1])};
  const double half[3] = {size[0] / 2.0, size[2] / 2.0, size[1] / 2.0};
  const double off[3] = {size[0] % 2 == 0 ? 0.0 : 0.5, size[2] % 2 == 0 ? 0.0 : 0.5, size[1] % 2 == 0 ? 0.0 : -0.5};
  Affine a{};
  for (int i = 0; i < 3; ++i) {
    double c = 0.0, o = 0.0;
    for (int k = 0; k < 3; ++k) {
      a.m[i][k] = Me[i][k];
      c += Me[i][k] * half[k];       // center * scale
      o += det * Me[i][k] * off[k];  // quat * offset (quat = M / s)
    }
    a.m[i][3] = te[i] - c + o;
  }
  return a;
}

struct Walker {
  const std::map<uint32_t, Node>& nodes;
  Scene& scene;
  uint32_t frame = 0;    // animation frame the scene is instantiated at
  uint32_t visits = 0;   // a file is untrusted input: a graph whose groups list the same child over and over is a DAG that
                         // unfolds exponentially, so the walk as a whole is bounded, not just its depth
  // traverse_recursive (loader.rs:87-176)
  void walk(uint32_t id, const Affine& parent, int32_t tx, int32_t ty, int32_t tz, const Rotation& rot, int depth) {
    if (depth > 256) throw ParseError{"scene graph too deep"};
    if (++visits > (1u << 20)) throw ParseError{"scene graph unfolds into more than 2^20 nodes"};
    auto it = nodes.find(id);
    if (it == nodes.end()) throw ParseError{"scene graph references a missing node"};
    const Node& n = it->second;
    switch (n.kind) {
      case Node::kTransform: {
        if (n.frames.empty()) throw ParseError{"transform node without a frame"};
        const Keyframe& k = in_force(n.frames, frame);  // one frame: loader.rs:107-116; several: loader.rs:103-105 is unimplemented!()
        // translation accumulates, rotation is replaced (loader.rs:117-121)
        walk(n.child, parent, tx + k.t[0], ty + k.t[1], tz + k.t[2], k.rot, depth + 1);
        break;
      }
      case Node::kGroup: {
        const int32_t t[3] = {tx, ty, tz};
        const uint32_t zero[3] = {0, 0, 0};
        const Affine g = parent * to_transform(t, rot, zero);  // loader.rs:127-131
        for (uint32_t c : n.children) walk(c, g, 0, 0, 0, Rotation::identity(), depth + 1);
        break;
      }
      case Node::kShape: {
        if (n.models.empty()) throw ParseError{"shape node without a model"};
        const uint32_t mid = in_force(n.models, frame).model;  // one model: loader.rs:152; several: loader.rs:149-151 is unimplemented!()
        if (scene.instances.size() >= 65535) throw ParseError{"more than 65535 instances"};
        if (mid >= scene.models.size()) throw ParseError{"shape references a missing model"};
        Model& m = scene.models[mid];
        if (m.xyzi.empty()) return;  // loader.rs:154-156
        const int32_t t[3] = {tx, ty, tz};
        const Affine a = parent * to_transform(t, rot, m.size);
        DustVoxInstance inst{};
        inst.model = mid;
        for (int i = 0; i < 3; ++i)
          for (int j = 0; j < 4; ++j) inst.obj_to_world[i * 4 + j] = static_cast<float>(a.m[i][j]);
        scene.instances.push_back(inst);
        m.used = true;
        break;
      }
      default: throw ParseError{"scene graph references an unknown node kind"};
    }
  }
};

// the palette MagicaVoxel (and dot_vox's DEFAULT_PALETTE) uses when a file has no RGBA chunk:
// entry i of this table is the colour of file colour index i+1.
void default_palette(uint8_t out[256 * 4]) {
  size_t k = 0;
  static const uint8_t lv[6] = {0xFF, 0xCC, 0x99, 0x66, 0x33, 0x00};
  for (int r = 0; r < 6; ++r)
    for (int g = 0; g < 6; ++g)
      for (int b = 0; b < 6; ++b) {
        if (r == 5 && g == 5 && b == 5) continue;
        out[k * 4 + 0] = lv[r]; out[k * 4 + 1] = lv[g]; out[k * 4 + 2] = lv[b]; out[k * 4 + 3] = 0xFF;
        ++k;
      }
  static const uint8_t ramp[10] = {0xEE, 0xDD, 0xBB, 0xAA, 0x88, 0x77, 0x55, 0x44, 0x22, 0x11};
  for (int c = 0; c < 4; ++c)  // red, green, blue, grey ramps
    for (int i = 0; i < 10; ++i) {
      const uint8_t v = ramp[i];
      out[k * 4 + 0] = (c == 0 || c == 3) ? v : 0;
      out[k * 4 + 1] = (c == 1 || c == 3) ? v : 0;
      out[k * 4 + 2] = (c == 2 || c == 3) ? v : 0;
      out[k * 4 + 3] = 0xFF;
      ++k;
    }
  for (; k < 256; ++k) out[k * 4 + 0] = out[k * 4 + 1] = out[k * 4 + 2] = out[k * 4 + 3] = 0;
}

float linear2srgb(float c) {  // geometry.rs:99-105
  if (c <= 0.0031308f) return 12.92f * c;
  return 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f;
}

}  // namespace

void flatten_model(const uint8_t* xyzi, size_t n_voxels, const uint32_t size[3], const uint8_t* palette,
                   std::vector<DustHipBlock>& blocks, std::vector<uint8_t>& materials) {
  static const uint32_t kHierarchy[3] = {4, 2, 2};  // crates/vox/src/lib.rs:19
  vdb::Tree tree(kHierarchy, 3);
  // ModelIndexCollector (collector.rs:2-88), kept sparse: one 64-byte tile per touched 4^3 block instead
  // of the reference's dense 16 MiB grid. block index = bx + 64*by + 4096*bz, cell = z | y<<2 | x<<4.
  std::vector<int32_t> tile_of(64 * 64 * 64, -1);
  struct Tile { uint8_t v[64]; uint32_t count; };
  std::vector<Tile> tiles;
  for (size_t i = 0; i < n_voxels; ++i) {
    const uint8_t x = xyzi[i * 4], y = xyzi[i * 4 + 2];
    const uint8_t z = static_cast<uint8_t>(size[1] - uint32_t(xyzi[i * 4 + 1]) - 1);  // loader.rs:248-253
    tree.set(x, y, z, 1);
    const uint32_t b = uint32_t(x >> 2) + uint32_t(y >> 2) * 64 + uint32_t(z >> 2) * 4096;
    int32_t& slot = tile_of[b];
    if (slot < 0) {
      slot = static_cast<int32_t>(tiles.size());
      tiles.emplace_back();
      std::memset(&tiles.back(), 0, sizeof(Tile));
    }
    Tile& t = tiles[slot];
    t.count += 1;  // duplicates are counted every time, as the reference does (collector.rs:23-34)
    t.v[(z & 3) | ((y & 3) << 2) | ((x & 3) << 4)] = static_cast<uint8_t>(xyzi[i * 4 + 3] + 1);
  }
  // exclusive prefix sum in block order + compaction (collector.rs:50-60,76-87)
  std::vector<uint32_t> start(tiles.size());
  materials.clear();
  materials.reserve(n_voxels);
  uint32_t running = 0;
  for (uint32_t b = 0; b < 64 * 64 * 64; ++b) {
    const int32_t slot = tile_of[b];
    if (slot < 0) continue;
    start[slot] = running;
    running += tiles[slot].count;
    for (int c = 0; c < 64; ++c)
      if (tiles[slot].v[c]) materials.push_back(static_cast<uint8_t>(tiles[slot].v[c] - 1));
  }
  // VoxGeometry::from_tree (geometry.rs:68-128), with material_ptr patched in first (loader.rs:265-272)
  blocks.clear();
  tree.for_each_leaf([&](vdb::LeafRef& leaf) {
    const uint32_t b = (leaf.origin[0] >> 2) + (leaf.origin[1] >> 2) * 64 + (leaf.origin[2] >> 2) * 4096;
    *leaf.material_ptr = start[tile_of[b]];
    const uint32_t n = uint32_t(__builtin_popcountll(leaf.occupancy));
    uint32_t sum[4] = {0, 0, 0, 0};
    for (uint32_t i = 0; i < n; ++i) {
      const size_t at = size_t(*leaf.material_ptr) + i;
      if (at >= materials.size()) throw ParseError{"material index out of range (duplicate XYZI entries)"};
      const uint8_t* c = palette + size_t(materials[at]) * 4;
      for (int k = 0; k < 4; ++k) sum[k] += c[k];
    }
    const float denom = float(n) * 255.0f;
    float col[4];
    for (int k = 0; k < 4; ++k) col[k] = float(sum[k]) / denom;
    for (int k = 0; k < 3; ++k) col[k] = linear2srgb(col[k]);
    const uint32_t r = uint32_t(col[0] * 1023.0f), g = uint32_t(col[1] * 1023.0f), bl = uint32_t(col[2] * 1023.0f);
    const uint32_t a = uint32_t(col[3] * 3.0f);
    DustHipBlock blk{};
    blk.x = uint16_t(leaf.origin[0]); blk.y = uint16_t(leaf.origin[1]); blk.z = uint16_t(leaf.origin[2]);
    blk.w = 0;
    blk.mask = leaf.occupancy;
    blk.material_ptr = *leaf.material_ptr;
    blk.avg_albedo = (r << 22) | (g << 12) | (bl << 2) | a;
    blocks.push_back(blk);
  });
}

Scene load(const uint8_t* bytes, size_t n, uint32_t frame) {
  Reader rd{bytes, n};
  if (!rd.has(8) || std::memcmp(bytes, "VOX ", 4) != 0) throw ParseError{"Not a valid MagicaVoxel .vox file"};
  rd.pos = 4;
  const uint32_t version = rd.u32();
  if (version != 150 && version != 200) throw ParseError{"Unknown .vox version"};
  if (!rd.has(12) || std::memcmp(bytes + rd.pos, "MAIN", 4) != 0) throw ParseError{"missing MAIN chunk"};
  rd.pos += 4;
  const uint32_t main_content = rd.u32();
  const uint32_t main_children = rd.u32();
  if (!rd.has(size_t(main_content) + main_children)) throw ParseError{"MAIN chunk runs past end of file"};
  rd.pos += main_content;
  const size_t end = rd.pos + main_children;

  Scene scene;
  bool have_palette = false;
  std::map<uint32_t, Node> nodes;
  uint32_t pending_size[3] = {0, 0, 0};
  bool have_size = false;
  while (rd.pos + 12 <= end) {
    char id[5] = {0};
    std::memcpy(id, bytes + rd.pos, 4);
    rd.pos += 4;
    const uint32_t content = rd.u32(), children = rd.u32();
    if (rd.pos + size_t(content) + children > end) throw ParseError{"chunk runs past end of MAIN"};
    Reader c{bytes + rd.pos, content};
    if (!std::strcmp(id, "SIZE")) {
      pending_size[0] = c.u32(); pending_size[1] = c.u32(); pending_size[2] = c.u32();
      have_size = true;
    } else if (!std::strcmp(id, "XYZI")) {
      if (!have_size) throw ParseError{"XYZI chunk without SIZE"};
      const uint32_t k = c.u32();
      if (!c.has(size_t(k) * 4)) throw ParseError{"XYZI chunk truncated"};
      Model m;
      std::memcpy(m.size, pending_size, sizeof(m.size));
      m.xyzi.assign(c.p + c.pos, c.p + c.pos + size_t(k) * 4);
      for (size_t v = 0; v < k; ++v) {  // dot_vox: i = index.saturating_sub(1) (file indices are 1-based; 0 is not a colour)
        const uint8_t ci = m.xyzi[v * 4 + 3];
        m.xyzi[v * 4 + 3] = static_cast<uint8_t>(ci ? ci - 1 : 0);
      }
      scene.models.push_back(std::move(m));
      have_size = false;
    } else if (!std::strcmp(id, "RGBA")) {
      if (!c.has(256 * 4)) throw ParseError{"RGBA chunk truncated"};
      std::memcpy(scene.palette, c.p, 256 * 4);
      have_palette = true;
    } else if (!std::strcmp(id, "nTRN")) {
      Node nd;
      nd.kind = Node::kTransform;
      const uint32_t node_id = c.u32();
      c.dict();
      nd.child = c.u32();
      c.i32();  // reserved
      c.i32();  // layer
      const uint32_t n_frames = c.u32();
      if (n_frames > content / 4) throw ParseError{"nTRN frame count exceeds the chunk"};
      for (uint32_t f = 0; f < n_frames; ++f) {
        auto d = c.dict();
        Keyframe k;
        auto t = d.find("_t");
        if (t != d.end()) {
          long a = 0, b = 0, e = 0;
          if (std::sscanf(t->second.c_str(), "%ld %ld %ld", &a, &b, &e) != 3) throw ParseError{"bad _t in nTRN frame"};
          k.t[0] = int32_t(a); k.t[1] = int32_t(b); k.t[2] = int32_t(e);
        }
        auto r = d.find("_r");
        if (r != d.end()) k.rot = Rotation::from_byte(static_cast<uint8_t>(std::strtoul(r->second.c_str(), nullptr, 10)));
        auto fi = d.find("_f");
        if (fi != d.end()) k.frame = uint32_t(std::strtoul(fi->second.c_str(), nullptr, 10));
        nd.frames.push_back(k);
      }
      nodes[node_id] = std::move(nd);
    } else if (!std::strcmp(id, "nGRP")) {
      Node nd;
      nd.kind = Node::kGroup;
      const uint32_t node_id = c.u32();
      c.dict();
      const uint32_t k = c.u32();
      if (k > content / 4) throw ParseError{"nGRP child count exceeds the chunk"};
      for (uint32_t i = 0; i < k; ++i) nd.children.push_back(c.u32());
      nodes[node_id] = std::move(nd);
    } else if (!std::strcmp(id, "nSHP")) {
      Node nd;
      nd.kind = Node::kShape;
      const uint32_t node_id = c.u32();
      c.dict();
      const uint32_t k = c.u32();
      if (k > content / 8) throw ParseError{"nSHP model count exceeds the chunk"};
      for (uint32_t i = 0; i < k; ++i) {
        ShapeModel sm;
        sm.model = c.u32();
        auto d = c.dict();
        auto fi = d.find("_f");
        if (fi != d.end()) sm.frame = uint32_t(std::strtoul(fi->second.c_str(), nullptr, 10));
        nd.models.push_back(sm);
      }
      nodes[node_id] = std::move(nd);
    }  // PACK, MATL, LAYR, rOBJ, rCAM, NOTE, IMAP ...: not consumed by the reference loader
    rd.pos += size_t(content) + children;
  }
  if (!have_palette) default_palette(scene.palette);

  if (nodes.empty()) {
    // files without a scene graph: exactly one model at identity (loader.rs:68-84)
    if (scene.models.size() != 1) throw ParseError{"file without scene graph must hold exactly one model"};
    if (!scene.models[0].xyzi.empty()) {
      DustVoxInstance inst{};
      inst.model = 0;
      inst.obj_to_world[0] = inst.obj_to_world[5] = inst.obj_to_world[10] = 1.0f;
      scene.instances.push_back(inst);
      scene.models[0].used = true;
    }
  } else {
    Walker w{nodes, scene, frame};
    w.walk(0, Affine::identity(), 0, 0, 0, Rotation::identity(), 0);
  }

  for (const Model& m : scene.models)
    if (m.used && (m.size[0] > 256 || m.size[1] > 256 || m.size[2] > 256))
      throw ParseError{"model larger than 256^3"};  // loader.rs:365 assert!

  // one model per thread (loader.rs:360-372)
  std::vector<uint32_t> todo;
  for (uint32_t i = 0; i < scene.models.size(); ++i)
    if (scene.models[i].used) todo.push_back(i);
  const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  const unsigned nt = std::min<unsigned>(hw, unsigned(todo.size()));
  std::vector<std::thread> pool;
  std::vector<std::string> errors(nt);
  for (unsigned t = 0; t < nt; ++t)
    pool.emplace_back([&, t] {
      try {
        for (size_t k = t; k < todo.size(); k += nt) {
          Model& m = scene.models[todo[k]];
          flatten_model(m.xyzi.data(), m.xyzi.size() / 4, m.size, scene.palette, m.blocks, m.materials);
        }
      } catch (const ParseError& e) {
        errors[t] = e.what;
      }
    });
  for (auto& th : pool) th.join();
  for (const auto& e : errors)
    if (!e.empty()) throw ParseError{e};
  return scene;
}

}  // namespace dust::vox
