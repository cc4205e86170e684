// dust_dev.h -- plain structs shared by the host runtime (capi.cpp) and the HIP kernels (kernels.hip).
//
// HBM layout of one model (DESIGN.md "Data layout"):
//   root      : one N16 node   [64 x u64 child mask][64 x u16 rank prefix][u32 child_base][pad]   656 B
//   l2        : N16 nodes, only for 4096^3 trees (hierarchy (4,4,2,2)), same 656 B layout
//   mid       : N4 nodes       {u64 child mask, u32 first_block, u32 pad}                         16 B
//   dense_mask: u64 occupancy per (mid node, child bit), 0 where there is no brick     64 x 8 B per mid node
//   blocks    : the reference's 24-byte Block records, block order (shading only)
//   materials : u8 palette index per solid voxel; palette: 255 x RGBA8
// A child "pointer" is child_base + prefix[word] + popcount(mask[word] & below(bit)): nodes of one
// level are stored in depth-first order, which is also Tree::iter_leaf order, so the block index the
// traversal arrives at IS the reference's gl_PrimitiveID.
#pragma once
#include <stdint.h>

#include "../../include/dust_hip.h"

// Address spaces. The kernels read their launch descriptor from the kernel-argument segment; every pointer inside it
// would be a generic ("flat") pointer to the compiler: flat_load for everything, no scalar loads (it cannot prove the scene is
// not written by the G-buffer stores). In the device pass the read-only scene pointers are therefore declared in the
// constant address space (uniform index -> s_load into SGPRs, divergent index -> global_load off a scalar base) and
// the written planes in the global one (global_store). Same 64-bit representation: the host (and every translation
// unit other than kernels.hip, which opts in) sees plain pointers.
#if defined(__HIP_DEVICE_COMPILE__) && defined(DUST_DEVICE_ADDRESS_SPACES)
#define DUST_CONST_AS __attribute__((address_space(4)))
#define DUST_GLOBAL_AS __attribute__((address_space(1)))
#else
#define DUST_CONST_AS
#define DUST_GLOBAL_AS
#endif
#define DUST_RO(T) const DUST_CONST_AS T*
#define DUST_RW(T) DUST_GLOBAL_AS T*

namespace dust {

constexpr uint32_t kN16Bytes = 656;      // 512 mask + 128 prefix + 4 base + 12 pad
constexpr uint32_t kN16LdsBytes = 640;   // mask + prefix only (root child_base is 0)
#ifndef DUST_TILE_W
#define DUST_TILE_W 8
#define DUST_TILE_H 8
#endif
constexpr uint32_t kTileW = DUST_TILE_W, kTileH = DUST_TILE_H;  // pixels of one ray packet (kTileW * kTileH == 64 lanes)
constexpr uint32_t kRegions = 8;          // work bands = XCDs (finer sub-queues per band measured 2.5 % slower: more empty-queue probes at the tail)
constexpr uint32_t kCounterStride = 64;  // u32s between the per-region work counters (256 B: no two share a cache line)
constexpr uint32_t kFlatCullMax = 256;   // instances up to which the packet cull tests every box (four rounds of 64); above: cull_instances' hierarchy
constexpr uint32_t kMaxCand = 160;       // per-wave candidate list capacity: one u32 {entry time hi16 | id16} each, twice (sort staging)
constexpr uint32_t kSurfelPoolSize = 720 * 480;  // surfel.glsl:2, standard.rs:338
constexpr uint32_t kSpatialHashCapacity = 32u * 1024u * 1024u;  // spatial_hash.glsl:1
constexpr uint32_t kMaxBatch = 8;   // frames ONE persistent launch of the fused pixel passes can carry (dust_hip_render_frames; BatchArgs)
constexpr uint32_t kApplyKeep = 8;  // DUST_PASS_GI_ORDERED: inserts of ONE key that a frame applies (the last kApplyKeep in surfel order; gi.hip, k_surfel_apply_mark)

struct DevN4 {
  uint32_t mask_lo, mask_hi;
  uint32_t first_block;
  uint32_t pad;
};

struct DevL2Cell {  // one 16^3 cell of a level-2 node, laid out for ONE 16-byte load: the mid node it holds and which of its 4^3 cells hold a brick
  uint32_t mid;         // index into mid / dense_mask, 0xFFFFFFFF = empty 16-cell
  uint32_t bounds;      // box of the occupied 4-cells in child coordinates 0..3: lo.x | lo.y<<2 | lo.z<<4 | hi.x<<6 | hi.y<<8 | hi.z<<10
  uint64_t child_mask;  // the mid node's 64-bit child mask
};

struct DevModel {
  DUST_RO(uint8_t) root;          // N16
  DUST_RO(uint8_t) l2;            // N16[] or null
  DUST_RO(DevN4) mid;
  DUST_RO(uint64_t) dense_mask;   // [mid node][64 child bits] -> brick occupancy, 0 = no brick
  DUST_RO(DustHipBlock) blocks;
  DUST_RO(uint8_t) materials;
  DUST_RO(uint32_t) palette;      // RGBA8 packed little-endian, 255 entries (+1 pad)
  float bmin[3], bmax[3];       // tight object-space bounds of the bricks
  uint32_t extent;              // 256 or 4096
  uint32_t n_levels;            // internal levels: 2 (root,mid) or 3 (root,l2,mid)
  uint32_t n_blocks;
  int32_t lds_slot;             // slot of the staged root in LDS, -1 = read it from HBM/L2
  DUST_RO(DevL2Cell) l2_cells;  // 4096^3 trees: [level-2 node][4096 cells], what the DEEP kernel variants look 16-cells up in
                                // (268 MB when every 256-cell is occupied: sized for 288 GB of HBM, and within reach of the
                                // 256 MB Infinity Cache); null for 256^3 trees
};

struct DevInstance {
  float w2o[12];   // world -> object, 3x4 row-major
  float o2w[12];   // object -> world, 3x4 row-major (VkTransformMatrixKHR)
  float prev[16];  // previous-frame object -> world, column-major mat4 (`instances[]`, layout.playout:72)
  float wmin[3], wmax[3];  // conservative world-space bounds
  uint32_t model;
  uint32_t pad;
};

struct DevVisit {  // what a ray needs to test and enter an instance, in one record: box, transform and model arrive together
  float lo[3], pad0;   // world bounds, as in DevBox
  float hi[3], pad1;
  float w2o[12];
  DevModel m;
};

struct DevBox {  // an instance's world bounds, 32 bytes: {lo.xyz, cells_lo, hi.xyz, cells_hi}
  float lo[3], pad0;   // pad0 / pad1 (as bit patterns): the block of top-level grid cells the box is listed in, low and high corner,
  float hi[3], pad1;   // x | y << 9 | z << 18 (DevGrid); the packet cull ignores them
};

// The top-level structure over the instances (what the reference hands to the driver as a TLAS, accel_struct/tlas.rs:37-117):
// a uniform grid over the union of the instances' world boxes; a cell lists the instances whose (slightly grown) box overlaps it.
// Built on the host by dust_hip_scene_commit, part of the scene image. The per-ray walks of the incoherent passes (gi.hip,
// k_ray_stream) step through it front to back; up to 256 cells per axis.
struct DevGrid {
  DUST_RO(uint32_t) cells;   // per cell {first item : 20 bits, items : 12 bits}; cell (x, y, z) is entry (z * dim[1] + y) * dim[0] + x
  DUST_RO(uint16_t) items;   // instance ids, cell after cell, ascending inside a cell
  float lo[3], hi[3];        // the grid's world box (hi = lo + dim * cell)
  float cell[3], inv_cell[3];
  uint32_t dim[3];
  uint32_t n_items;
};
constexpr uint32_t kGridItemBits = 20, kGridMaxCellItems = 4095;

// What a ray needs to enter an instance (walk_begin), 80 bytes = five 16-byte accesses: world -> object, the model's tight
// bounds (multiples of 4 up to 4096: exact in 16 bits), where its root is staged, and the two arrays a step reads.
struct DevEnter {
  float w2o[12];
  uint16_t bmin[3], bmax[3];
  uint16_t model;
  uint8_t lds_slot;      // 255: the root is read from memory
  uint8_t extent_log2;   // 8 or 12
  DUST_RO(uint8_t) root;
  DUST_RO(uint64_t) dense_mask;
};
// Where a workgroup of the ray-stream kernels keeps top-level data in LDS, as byte offsets (0xFFFFFFFF: the section is not
// staged -- it did not fit, or the kernel does not read it -- and comes from memory): grid cells, grid items, instance boxes
// (32 B each) for the ray-making kernels, from offset 0; enter records for k_ray_walk, behind its staged roots
struct DevStreamLds { uint32_t boxes, cells, items, enters, total; };

// One ray of a ray stream (gi.hip): 48 bytes, written by a ray-making kernel -- which also walks the top-level grid and lists the
// instances whose box the ray meets --, walked by k_ray_walk, whose hit record goes to ray_hits[id].
// flags bit 0: any-hit (terminate on first hit: the surfel pass's sun rays). cand[0..6]: instance ids in the order the grid walk
// met them (front to back by cell), cand[7]: how many, | 0x8000 when there are more than seven
struct DevRay { float ox, oy, oz; uint32_t id; float dx, dy, dz; uint32_t flags; uint16_t cand[8]; };

struct DevCamera {
  float col0[3], col1[3], col2[3], pos[3];
  float tan_half_fov, far_, near_;
};

struct DevStats {
  unsigned long long rays, instances_tested, upper_descents, mid_descents, bricks_tested, hits;
};

struct DevGBuffer {
  DUST_RW(uint16_t) illuminance;  // 4 halves / px
  DUST_RW(uint16_t) denoised;    // 4 halves / px
  DUST_RW(uint32_t) albedo;
  DUST_RW(uint32_t) normal;
  DUST_RW(float) depth;
  DUST_RW(uint16_t) motion;      // 4 halves / px
  DUST_RW(uint32_t) voxel_id;
  DUST_RW(float) accum;          // 4 floats / px
};

// SpatialHashEntry, 12 bytes, scalar layout (layout.playout:13-18): {u32 fingerprint, u32 LogLuv radiance,
// u16 last_accessed_frame, u16 sample_count}; SurfelEntry, 16 bytes (layout.playout:1-4): {vec3 position, u32 direction}
struct DevSurfel { float x, y, z; uint32_t direction; };
struct DevHashRequest {  // one SpatialHashInsert call recorded by the surfel pass, applied afterwards in surfel order
  int32_t kx, ky, kz;
  uint32_t dir_flags;    // bits 0-7 face id, bit 8: insert valid
  float vx, vy, vz;
  uint32_t stamped;      // 1 + index of the hash entry the surfel's SpatialHashGet found and stamped (surfel.rchit:47), 0 = none: what the other
                         // ranks of a sharded trace repeat (k_surfel_unstage) so that every rank's hash stays the single-GPU one
};
struct DevGatherHit { float t; uint32_t inst, block, found; };  // what a gather ray found: the hit record the shading kernel takes up
struct DevGI {
  uint32_t* hash;           // (capacity + 2) x 3 words
  uint32_t hash_capacity;
  DevSurfel* pool;          // pool_size entries
  uint32_t pool_size;
  uint32_t* slot_owner;     // pool_size: 1 + highest pixel index that enqueued into the slot this frame, 0 = none
  DevSurfel* pixel_surfel;  // width*height: the surfel each pixel wants to enqueue
  DevHashRequest* requests; // pool_size
  DevSurfel* replacement;   // pool_size: direction == 0xFFFFFFFF means "keep"
  float* sun_payload;       // pool_size x 4: what the surfel's sun ray adds to the value of its request (zero when shadowed)
  const uint32_t* perm;     // surfel indices ordered by position (k_surfel_keys + radix sort), or null = pool order
  uint32_t* sort_keys;      // pool_size keys / indices a radix sort consumes: k_surfel_keys fills them with 16-bit position keys,
  uint32_t* sort_vals;      //   k_surfel_apply_keys with the hash location of each insert request (the deterministic apply)
  const uint32_t* apply_keys;  // the insert requests ordered by hash location (pool_size = "no insert", sorted last) ...
  const uint32_t* apply_vals;  // ... and the surfel index each one belongs to
  uint32_t* order;          // final gather: per 32x32-pixel tile, its live pixels grouped by ray-direction octant (1024 slots per tile)
  uint32_t* order_count;    // live pixels per tile; null order = plain 8x8 pixel packets
  uint32_t order_tiles_x;   // 32x32 tiles per row
  uint32_t* touched;        // multi-GPU: per pixel, 1 + index of the hash entry its final gather stamped (null otherwise)
  DevSurfel* merged;        // multi-GPU: per slot, the winning surfel after the exchange
  DevGatherHit* fg_hits;    // per pixel: k_final_gather only TRACES and leaves its hits here, k_final_gather_shade does the hash lookups and
                            // stores afterwards (null: the gather kernel shades its own rays)
};

struct DevStream {
  // ray streams (k_gather_rays / k_surfel_rays -> k_ray_walk -> k_final_gather_shade / k_surfel_shade)
  DevRay* rays;             // the pass's rays in GROUPS: group g (a 16 x 16 pixel tile / 256 consecutive surfels) owns entries
  uint32_t* group_count;    //   [g * group_rays, g * group_rays + group_count[g]): its live rays, compacted inside the group
  uint32_t n_groups, group_rays;  // (no atomics, and the same order in every run)
  DevGatherHit* ray_hits;   // [ray id]: pixel index for gather rays (== fg_hits), 2 * surfel + kind for surfel rays
  float ray_tmin, ray_tmax; // gl_RayTminEXT / gl_RayTmaxEXT of the pass (final_gather.rgen:47, surfel.rgen:33-62)
  uint32_t* unbinned;       // [2]: rays the ray-making kernel settled itself (no instance box on their way), closest-hit / any-hit;
  uint32_t count_unbinned;  //   kept only in a counting frame (count_unbinned != 0), for the pass statistics
};

struct FrameArgs {
  DUST_RO(DevModel) models;
  DUST_RO(DevInstance) instances;
  uint32_t n_models, n_instances;
  uint32_t n_lds_models;      // roots staged in LDS: models[i].lds_slot == i for i < n_lds_models
  uint32_t n_lds_boxes;       // n_instances if the instance boxes are staged in LDS too (32 B each, behind the queue), else 0
  DUST_RO(uint8_t) root_table;  // n_lds_models x kN16LdsBytes, packed copy of those roots (mask + prefix)
  DUST_RO(DevBox) boxes;        // n_instances world boxes (copy of DevInstance::wmin/wmax, packed)
  DUST_RO(DevVisit) visits;     // n_instances {world -> object, model record}
  float world_min[3], world_max[3];  // union of the instances' world boxes
  DevCamera cam;
  float sky[56];
  float sun_dir[3];           // normalize(sky[48..50]): the sun shadow rays' direction
  float sun_term[3];          // sun_radiance(sun_dir) * (1 - cos(solar radius)): what a lit point receives per unit cos(theta)
                              // (nee.rmiss:11-22 evaluates it per ray; it is a constant of the frame, worked out once on the host)
  DevGBuffer g;
  uint32_t width, height;
  float inv_width, inv_height, aspect;  // RN(1 / width), RN(1 / height), RN(width / height): camera.glsl's divisions, done once
  uint32_t row_begin, row_end;
  uint32_t tiles_x, tiles_y, tile_row0;  // 8x8 pixel tiles covering [row_begin,row_end)
  uint32_t tiles_per_band;    // ceil(tiles / 8): the launch's hand-out schedule, filled in at launch (kernels.hip, with_schedule)
  uint32_t tiles_x_magic;     // floor(2^32 / tiles_x) + 1: __umulhi(tile, magic) == tile / tiles_x
  uint32_t static_rounds;     // rounds of a band's order that are dealt to its waves; the rest is grabbed
  uint32_t static_rounds_request;  // DUST_HIP_STATIC_ROUNDS (0xFFFFFFFF: the default of with_schedule)
  // Cost-ordered work distribution: every wave records the cycles it spent on each tile (tile_cost); before the next launch
  // of the same pass k_tile_order turns that into tile_order -- per XCD band, most expensive first --, which maps ticket ->
  // tile. Both null on the first frame of a pass (identity order) or when switched off.
  DUST_RO(uint32_t) tile_order;
  DUST_RO(uint32_t) band_cuts;        // with tile_order: kRegions + 1 tile indices, band b = [cuts[b], cuts[b + 1]): contiguous bands of about
                                      // EQUAL MEASURED COST instead of equal tile counts (k_tile_order). Null: tiles_per_band each
  DUST_RW(uint32_t) tile_cost;
  DUST_RW(uint32_t) work_counters;    // 8 per-band tile counters, kCounterStride apart, zero at launch
  DUST_RW(uint32_t) next_work_counters;  // the set the next launch of this pass kind uses: this launch zeroes it
  DUST_RO(uint8_t) noise0;     // 128*128 R8 slice for this frame, or null
  DUST_RO(uint8_t) noise5;     // 128*128 RGBA8 slice for this frame, or null
  uint32_t rand, frame_index;
  DUST_RW(DevStats) stats;           // [2]: per pass kind, only written by the counting build
  uint32_t accum_count;       // frames already in `accum`
  // hash-fed GI (final gather + surfel passes)
  DevGI gi;
  uint32_t deep;              // the scene holds a 4096^3 model with its per-cell table: launch the DEEP kernel variants
  uint32_t prio_floor;        // lowest issue priority this launch's waves run at (the surfel pass on the second stream, beside the next frame's kernels: 3)
  uint32_t debug;             // DUST_HIP_DEBUG ablation bits (1: skip tracing, 2: skip culling, 4: walk every instance in
                              // index order instead of the packet's sorted candidate list, 8: gather / surfel rays take the
                              // wave-uniform candidate walk of the coherent ray types); 0 in production
  // ---- round 5, appended (the fields above keep the offsets the packet kernels were tuned with: their scalar loads come in aligned runs)
  DUST_RO(DevEnter) enters;     // n_instances compact enter records (the ray streams' instance set-up)
  // Large scenes (more than kFlatCullMax instances; n_groups != 0): the packet cull's 64-wide hierarchy. The instances in the order
  // of a space-filling curve through their boxes' centres, every 64 consecutive ones a group:
  DUST_RO(DevBox) gboxes;       //   n_groups group boxes (the union of the group's instance boxes)
  DUST_RO(DevBox) sboxes;       //   n_instances instance boxes in that order; pad0 (as a bit pattern) = the instance's id
  uint32_t n_groups;
  DevStreamLds sl_bin, sl_walk; // what the ray-making kernels / k_ray_walk stage in LDS
  uint32_t stream_refill;       // k_ray_walk: lanes not walking at which a wave leaves the walk to set the others up (DUST_HIP_STREAM_REFILL)
  uint32_t stream_top_iters;    // ... and the grid steps + box tests per phase of a lane that walks the grid itself (candidate overflow)
  DevGrid grid;               // top-level structure over the instances (per-ray walks of the GI passes)
  DevStream stream;           // the pass's ray stream (DUST_HIP_RAY_STREAM)
  // A frame's FIRST traversal launch on the context's stream says that it has started: workgroup 0 writes `started_seq` into a word of
  // pinned host memory (null: this launch says nothing). Launches of one stream run one after the other, so the host then knows that
  // every earlier frame of the stream is done -- which is how dust_hip_scene_commit recycles a scene image without waiting for the
  // whole queue when the host runs a ring of commits ahead (capi.cpp).
  DUST_RW(uint32_t) started_word;
  uint32_t started_seq;
  // ---- round 6, appended: the surfel TRACE sharded over the ranks of an N-GPU job (DustHipFrameParams::surfel_rank / surfel_world).
  // A rank traces the groups [sf_group_begin, sf_group_begin + sf_group_count) of the position-ordered pool (64 slots each) and leaves its
  // records in SLOT order in three staging arrays -- contiguous per rank, so one all-gather per array completes them on every rank --;
  // k_surfel_unstage then moves them to where the apply reads them (by surfel index) and repeats the trace's hash stamps.
  // sf_stage_req null: the unsharded pass (records go straight to gi.requests / replacement / sun_payload, by surfel index).
  DevHashRequest* sf_stage_req;
  DevSurfel* sf_stage_repl;
  float* sf_stage_sun;
  uint32_t sf_group_begin, sf_group_count;
  // the deterministic apply's "superseded" marks (k_surfel_apply_mark): bit i of apply_alive = the request at position i of the
  // location-sorted order is applied; apply_dead[j] = surfel j's request is not (the same fact by surfel index, for the serial loop)
  unsigned long long* apply_alive;
  uint8_t* apply_dead;
  unsigned long long* apply_starts;   // [0, words): bit i = position i begins a CLUSTER (requests whose probe windows may overlap); [words, 2 words): bit i =
                                      // position i begins a RUN of one location. A thread finds where its cluster and its runs end by scanning words, not keys
  uint32_t apply_words;
  // ---- round 6, second half: several frames in ONE persistent launch (dust_hip_render_frames, k_primary_ao_batch). Set in the FIRST descriptor of
  // a BatchArgs only: how many of its descriptors are frames of this launch. 0 / 1 everywhere else.
  uint32_t batch_frames;
  uint32_t batch_grab;        // every descriptor of a BatchArgs: tickets a refill of this frame's queue takes (launch_primary_ao_batch: up to 16 where a band
                              // holds many rounds of tiles for its workgroups, 4 in the launch's last frame and in small frames)
  uint32_t batch_queue_base;  // every descriptor of a BatchArgs: LDS byte offset of frame 1's tile queue (frame f: + 16 (f - 1)) -- the end of frame 0's LDS layout, which a
                              // frame that reads another scene image (n_lds_boxes = 0: its boxes come from memory) cannot work out from its own fields
};

// The kernel argument of k_primary_ao_batch: up to kMaxBatch whole launch descriptors, one per frame, side by side in the kernel-argument segment
// (8.5 KB; the runtime takes 16 KB, probed on an MI355X). The frames share scene, frame size, rows and launch geometry -- what is staged in LDS and
// how the waves are numbered --; camera, sky, noise slices, planes, tile order / costs / cuts and work counters are each frame's own. A wave that
// finds frame f without tiles goes on to frame f + 1: ONE launch tail, one staging and one inter-launch gap for batch_frames frames.
struct BatchArgs {
  FrameArgs f[kMaxBatch];
};

}  // namespace dust
