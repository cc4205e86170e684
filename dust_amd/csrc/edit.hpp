// edit.hpp -- interface between the host runtime (capi.cpp) and the device-side tree rebuild (edit.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "dust_dev.h"

namespace dust {

constexpr uint32_t kLattice = 64 * 64 * 64;  // bricks of a 256^3 model
constexpr uint32_t kSrgbRow = 64 * 255 + 1;  // colour sums 0 .. 64 * 255 per voxel count

struct EditHeader {  // what the host reads back after a rebuild
  uint32_t n_blocks, n_mid;
  unsigned long long n_materials;
  float bmin[3], bmax[3];
};

struct EditArgs {
  uint8_t* grid;             // kLattice * 64 bytes: palette index + 1 per voxel, [iter_leaf brick code][bit x<<4 | y<<2 | z]
  uint64_t* brick_mask;      // kLattice, iter_leaf order
  uint32_t* flag_leaf;       // kLattice (+ scan): brick non-empty, iter_leaf order   -> block index
  uint32_t* count_major;     // kLattice (+ scan): voxels per brick, collector order  -> material_ptr
  uint32_t* scan_tmp;        // 2 * 256 block sums
  DustHipBlock* blocks;      // capacity kLattice
  uint8_t* materials;        // capacity kLattice * 64
  const uint32_t* palette;   // RGBA8 x 256
  const uint16_t* srgb_lut;  // [64][kSrgbRow]: trunc(linear2srgb(sum / (n * 255)) * 1023), evaluated on the host
  uint8_t* root;             // one N16 node (kN16Bytes)
  DevN4* mid;                // capacity 4096
  uint64_t* dense_mask;      // capacity 4096 * 64
  EditHeader* header;
  // edit batch
  const uint32_t* xyz;
  const int32_t* values;
  int32_t* values_out;
  uint32_t n_edits;
};

hipError_t launch_edit_expand(const EditArgs& e, const DustHipBlock* blocks, const uint8_t* materials, uint32_t n_blocks, hipStream_t s);
hipError_t launch_edit_apply(const EditArgs& e, bool read, hipStream_t s);
hipError_t launch_edit_rebuild(const EditArgs& e, hipStream_t s);

}  // namespace dust
