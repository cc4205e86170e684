// vox.hpp -- MagicaVoxel .vox loader and model flattening (product host code).
//
// Stands in for VoxLoader::load (reference crates/vox/src/loader.rs:322-415) minus the Vulkan uploads:
// chunk parser (dot_vox 5.1.1 in the reference; written here from the public file-format description),
// scene-graph walk -> instance transforms (loader.rs:60-204), per-model tree build + palette-index
// collector (loader.rs:238-288, collector.rs:2-88), VoxGeometry::from_tree (geometry.rs:55-179).
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/dust_hip.h"

namespace dust::vox {

struct ParseError { std::string what; bool unsupported = false; };

struct Model {
  uint32_t size[3] = {0, 0, 0};  // file axes
  std::vector<uint8_t> xyzi;     // 4 bytes per voxel, i already 0-based (dot_vox convention)
  bool used = false;
  std::vector<DustHipBlock> blocks;
  std::vector<uint8_t> materials;
};

struct Scene {
  std::vector<Model> models;
  uint8_t palette[256 * 4];
  std::vector<DustVoxInstance> instances;
};

// Throws ParseError. Builds blocks/materials for every model an instance references,
// one thread per model as the reference does with rayon (loader.rs:360-372).
// frame: the animation frame at which multi-frame transform nodes and multi-model shape nodes are instantiated (the
// reference stops at unimplemented!() for both, loader.rs:103-105,149-151; files without animation ignore it).
Scene load(const uint8_t* bytes, size_t n, uint32_t frame = 0);

// load_model + from_tree for one model (hierarchy!(4,2,2), crates/vox/src/lib.rs:19-20).
void flatten_model(const uint8_t* xyzi, size_t n_voxels, const uint32_t size[3], const uint8_t* palette_rgba256,
                   std::vector<DustHipBlock>& blocks, std::vector<uint8_t>& materials);

}  // namespace dust::vox
