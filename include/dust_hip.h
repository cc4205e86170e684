/*
 * dust_hip.h -- C ABI of the MI355X-native replacement for Dust's ray-tracing hot path.
 *
 * Every entry point names the reference interface it stands in for (paths relative to the
 * dust-engine/dust checkout). Plain pointers and sizes only; no C++ or torch types cross this
 * boundary; no exception or panic crosses it either: every function returns a DustStatus and
 * dust_hip_last_error() holds the message of the calling thread's last failure.
 *
 * Ownership: handles are owned by the library and released by the matching *_destroy; input
 * arrays are borrowed for the duration of the call only (the library copies what it keeps).
 * Device-side handles (context, model, scene, pipeline) are reference-counted inside the library
 * the way the reference's are Arc / Handle<T> (Handle<VoxGeometry>, Arc<PipelineLayout> ...):
 * *_destroy gives up the caller's reference, and an object lives until its last user is gone, so
 * the destroy calls may come in ANY order (a garbage-collected host may finalise a context before
 * the models made from it).
 * Threading: a context and everything created from it is externally synchronised (one thread
 * at a time), matching the reference's single render system per frame (examples/castle.rs:139-236).
 */
#ifndef DUST_HIP_H
#define DUST_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum DustStatus {
  DUST_OK = 0,
  DUST_ERR_INVALID_ARGUMENT = -1,
  DUST_ERR_NO_DEVICE = -2,      /* no HIP device / HIP runtime failure at context creation */
  DUST_ERR_HIP = -3,            /* a HIP call or kernel launch failed; see dust_hip_last_error() */
  DUST_ERR_OUT_OF_MEMORY = -4,
  DUST_ERR_PARSE = -5,          /* VoxLoadingError::ParseError (crates/vox/src/loader.rs:311-317) */
  DUST_ERR_UNSUPPORTED = -6,    /* paths the reference leaves todo!() (node/internal.rs:121-124) / image kinds its loader rejects */
  DUST_ERR_NOT_READY = -7       /* StandardPipeline::render returning None (standard.rs:254-266) */
} DustStatus;

const char* dust_hip_last_error(void);
/* number of HIP devices visible; 0 when there is no GPU (never fails) */
int dust_hip_device_count(void);

/* ===================================================================== vdb tree builder (host)
 * Replaces dust_vdb::Tree<hierarchy!(...)> (crates/vdb/src/tree.rs:7-124) and Accessor
 * (accessor.rs:5-139). The hierarchy is given at run time as per-level fan-out log2s, root first:
 * hierarchy!(4,2,2) == {4,2,2}. */
typedef struct DustVdbTree DustVdbTree;
typedef struct DustVdbAccessor DustVdbAccessor;

DustStatus dust_vdb_tree_create(const uint32_t* fanout_log2, uint32_t n_levels, DustVdbTree** out); /* Tree::new, tree.rs:30-48 */
void dust_vdb_tree_destroy(DustVdbTree*);
/* Tree::set_value (tree.rs:83-85). value: 1 = Some(true), 0 = Some(false), -1 = None.
 * None returns DUST_ERR_UNSUPPORTED where the reference is todo!() (node/internal.rs:121-124). */
DustStatus dust_vdb_tree_set(DustVdbTree*, uint32_t x, uint32_t y, uint32_t z, int32_t value);
/* Tree::get_value (tree.rs:78-80). *value: 1 Some(true), 0 Some(false), -1 None */
DustStatus dust_vdb_tree_get(const DustVdbTree*, uint32_t x, uint32_t y, uint32_t z, int32_t* value);
/* Tree::iter (tree.rs:102-104): writes up to cap (x,y,z) triples in iteration order, *count = total */
DustStatus dust_vdb_tree_iter(const DustVdbTree*, uint32_t* xyz, size_t cap, size_t* count);
/* Tree::iter_leaf (tree.rs:106-113): leaf origins, 64-bit occupancy and material_ptr per leaf */
DustStatus dust_vdb_tree_iter_leaf(const DustVdbTree*, uint32_t* xyz, uint64_t* occupancy, uint32_t* material_ptr,
                                   size_t cap, size_t* count);
/* TreeMeta::META_MASK (tree.rs:154-167) and ROOT::LEVEL */
DustStatus dust_vdb_tree_meta(const DustVdbTree*, uint32_t* meta_mask, uint32_t* root_level);
/* lowest_common_ancestor_level (accessor.rs:15-30) */
uint32_t dust_vdb_lca_level(const uint32_t a[3], const uint32_t b[3], uint32_t meta_mask, uint32_t root_level);
/* Tree::accessor + Accessor::get (accessor.rs:37-57,125-131) */
DustStatus dust_vdb_accessor_create(const DustVdbTree*, DustVdbAccessor** out);
void dust_vdb_accessor_destroy(DustVdbAccessor*);
DustStatus dust_vdb_accessor_get(DustVdbAccessor*, uint32_t x, uint32_t y, uint32_t z, int32_t* value);

/* BitMask / Pool of crates/vdb (bitmask.rs:3-124, pool.rs:3-176), exposed for the reference's doctests */
typedef struct DustVdbPool DustVdbPool;
DustStatus dust_vdb_pool_create(size_t item_size, uint32_t chunk_size_log2, DustVdbPool** out); /* Pool::new */
void dust_vdb_pool_destroy(DustVdbPool*);
uint32_t dust_vdb_pool_alloc(DustVdbPool*);            /* Pool::alloc */
void dust_vdb_pool_free(DustVdbPool*, uint32_t index); /* Pool::free */
size_t dust_vdb_pool_num_chunks(const DustVdbPool*);   /* Pool::num_chunks */
/* BitMask::set / iter_set_bits over n_words 64-bit words */
void dust_vdb_bitmask_set(uint64_t* words, size_t index, int32_t value);
size_t dust_vdb_bitmask_iter_set_bits(const uint64_t* words, size_t n_words, uint32_t* out, size_t cap);

/* ===================================================================== .vox loader (host)
 * Replaces VoxLoader::load up to the point where it uploads (crates/vox/src/loader.rs:322-415):
 * file parse (dot_vox 5.1.1 in the reference), scene-graph walk (loader.rs:60-204), per-model
 * tree build + palette-index collector (loader.rs:238-288, collector.rs:2-88) and
 * VoxGeometry::from_tree (geometry.rs:55-179). */
typedef struct DustVoxScene DustVoxScene;

/* GPUVoxNode / shader `Block`, 24 bytes (geometry.rs:40-49, assets/shaders/headers/sbt.glsl:1-19) */
typedef struct DustHipBlock {
  uint16_t x, y, z, w;
  uint64_t mask;
  uint32_t material_ptr;
  uint32_t avg_albedo; /* R10 G10 B10 A2, sRGB-encoded */
} DustHipBlock;

typedef struct DustVoxModelInfo {
  uint32_t size[3]; /* model.size in file axes */
  uint32_t n_voxels;
  uint32_t n_blocks;      /* leaves of the tree == primitives of the reference BLAS */
  uint64_t n_materials;   /* bytes in the material buffer */
  uint32_t used;          /* referenced by at least one instance (loader.rs:360-372 only loads those) */
} DustVoxModelInfo;

typedef struct DustVoxInstance {
  uint32_t model;
  float obj_to_world[12]; /* 3x4 row-major, the TLAS instance transform (accel_struct/tlas.rs:99-105) */
} DustVoxInstance;

DustStatus dust_vox_load(const uint8_t* bytes, size_t n_bytes, DustVoxScene** out);
/* The same at animation frame `frame`. Where the reference stops at unimplemented!() -- transform nodes with several
 * frames (loader.rs:103-105) and shape nodes with several models (loader.rs:149-151), i.e. MagicaVoxel animations -- the
 * entry in force at `frame` is used: the one with the largest "_f" attribute <= frame, or the first one before the
 * animation starts. dust_vox_load is frame 0; files without animation give the same scene at every frame. */
DustStatus dust_vox_load_frame(const uint8_t* bytes, size_t n_bytes, uint32_t frame, DustVoxScene** out);
void dust_vox_scene_destroy(DustVoxScene*);
DustStatus dust_vox_scene_counts(const DustVoxScene*, uint32_t* n_models, uint32_t* n_instances);
DustStatus dust_vox_scene_model_info(const DustVoxScene*, uint32_t model, DustVoxModelInfo* out);
/* pointers stay valid until the scene is destroyed */
DustStatus dust_vox_scene_model_data(const DustVoxScene*, uint32_t model, const DustHipBlock** blocks,
                                     const uint8_t** materials);
DustStatus dust_vox_scene_palette(const DustVoxScene*, const uint8_t** rgba255x4); /* load_palette, loader.rs:208-236 */
DustStatus dust_vox_scene_instances(const DustVoxScene*, DustVoxInstance* out, uint32_t cap);

/* load_model + from_tree on caller-provided voxels (file axes, i = 0-based palette index).
 * Outputs are malloc'ed by the library; release with dust_vox_free(). */
DustStatus dust_vox_flatten_model(const uint8_t* xyzi, size_t n_voxels, const uint32_t size[3],
                                  const uint8_t* palette_rgba256, DustHipBlock** blocks, uint32_t* n_blocks,
                                  uint8_t** materials, uint64_t* n_materials);
void dust_vox_free(void*);

/* ---- PNG / APNG -> sliced image array ----
 * PngLoader::load (crates/rhyolite_bevy/src/loaders/png.rs:70-200), the loader behind the six spatiotemporal blue-noise
 * textures (crates/render/src/noise.rs:16-29, 128 x 128 x 64-frame APNGs): every animation frame becomes one layer;
 * grey and grey+alpha keep 1 / 2 channels, RGB is widened to RGBA with a zero fourth byte, 16-bit samples stay
 * big-endian. DUST_ERR_UNSUPPORTED for indexed colour, sub-byte samples, interlacing and partial frames.
 * The result feeds dust_hip_pipeline_set_noise directly (texture 0: 1 channel, texture 5: 4 channels). */
typedef struct DustPngInfo {
  uint32_t width, height, layers;
  uint32_t channels;           /* 1, 2 or 4 */
  uint32_t bytes_per_channel;  /* 1 or 2 */
} DustPngInfo;
DustStatus dust_png_load_array(const uint8_t* bytes, size_t n_bytes, DustPngInfo* info, uint8_t** texels /* dust_vox_free */);

/* ===================================================================== sky bake (host)
 * Sunlight::bake (crates/render/src/pipeline/sky.rs:90-268): Hosek-Wilkie sky + solar-disc state for a sun direction,
 * turbidity (1..10) and ground albedo -- the 56 floats DustHipSky carries to the shaders. The reference embeds the
 * model's tables with include_bytes! (sky.rs:34-63: dataset.bin, 1200 x vec3 = 14400 bytes; datasetSolar.bin,
 * 1806 x vec3 = 21672 bytes); this library does not contain them: the host passes the two files' bytes once.
 * (A host without the tables can use the pre-baked sweep under dust_amd/data/, see INTEGRATION.md.) */
typedef struct DustSkyDataset DustSkyDataset;
/* SkyModelState as Sunlight::bake() produces it, 56 floats (pipeline/sky.rs:66-132; layout.playout:35-51) */
typedef struct DustHipSky { float state[56]; } DustHipSky;
DustStatus dust_sky_dataset_create(const uint8_t* dataset_bin, size_t n_dataset, const uint8_t* dataset_solar_bin, size_t n_solar,
                                   DustSkyDataset** out);
void dust_sky_dataset_destroy(DustSkyDataset*);
/* direction: unit vector from eye to sun, y up, y > 0; albedo: ground albedo per XYZ channel (Sunlight::albedo) */
DustStatus dust_sky_bake(const DustSkyDataset*, float turbidity, const float albedo[3], const float direction[3], DustHipSky* out);

/* ===================================================================== device side
 * Replaces the Vulkan objects the render plugin owns (crates/render/src/lib.rs:58-134): device,
 * BLAS/TLAS stores, SBT, the four ray-tracing pipelines and their persistent buffers. */
typedef struct DustHipContext DustHipContext;
typedef struct DustHipModel DustHipModel;
typedef struct DustHipScene DustHipScene;
typedef struct DustHipPipeline DustHipPipeline;

typedef struct DustHipConfig {
  uint32_t struct_size;     /* sizeof(DustHipConfig) */
  int32_t device;           /* HIP device ordinal; -1 = current device */
  void* stream;             /* hipStream_t to launch on; NULL = a non-blocking stream owned by the context (note that the
                               legacy default stream IS the null handle: to share a stream with other code, create one) */
  uint32_t lds_root_bytes;  /* LDS budget for staged root nodes per workgroup; 0 = default (64 KiB) */
  uint32_t flags;           /* DUST_HIP_CONTEXT_* */
} DustHipConfig;
#define DUST_HIP_CONTEXT_TIMING 1u /* record hipEvents around every pass (dust_hip_pipeline_pass_stats) */
#define DUST_HIP_CONTEXT_TIMING_SPARSE 2u /* with TIMING: around the launches of every 4th frame only (dust_hip_pipeline_kernel_times then
                                             averages over those): an event record costs the stream ~6 us, 5 % of a 0.23 ms frame */

/* RenderPlugin::build -> device creation (crates/render/src/lib.rs:58-134) */
DustStatus dust_hip_context_create(const DustHipConfig*, DustHipContext** out);
void dust_hip_context_destroy(DustHipContext*);
DustStatus dust_hip_sync(DustHipContext*); /* waits for everything submitted on the context's stream */

/* Geometry + Material registration: VoxGeometry (AABB buffer + Block buffer, vox/src/geometry.rs:129-165),
 * PaletteMaterial parameters {geometry_ptr, material_ptr, palette_ptr} (vox/src/material.rs:30-41,106-119),
 * and the BLAS build (render/src/accel_struct/blas.rs:125-230), which here is the device-side VDB hierarchy.
 * tree_extent_log2: 8 for hierarchy!(4,2,2), 12 for (4,4,2,2). */
DustStatus dust_hip_model_create(DustHipContext*, const DustHipBlock* blocks, uint32_t n_blocks,
                                 const uint8_t* materials, uint64_t n_materials, const uint8_t* palette_rgba255x4,
                                 uint32_t tree_extent_log2, DustHipModel** out);
void dust_hip_model_destroy(DustHipModel*);
/* VoxGeometry::set / VoxGeometry::get (vox/src/geometry.rs:180-185; Tree::set_value / get_value, vdb/src/tree.rs:78-85) on the
 * DEVICE copy of a hierarchy (4,2,2) model. The reference's set only touches the CPU tree -- a loaded model's GPU buffers never
 * change -- so an edited voxel needs what that call does not carry, a material: values[i] >= 0 is Some(true) with palette
 * index values[i] (0..254), values[i] < 0 is None (the voxel is removed; the reference's todo!() for clearing through internal
 * nodes, node/internal.rs:121-124, does not apply: nothing is pointer-chased here). xyz: n coordinate triples in tree axes
 * (what Tree::set_value takes), < 256 each; a voxel named twice takes its last value. The first edit of a model moves it into
 * an editable form (a dense voxel grid on the device, ~44 MB); every batch then rewrites root, mid nodes, brick masks, Block
 * records (material_ptr, avg_albedo) and the material stream on the GPU with scans over the brick lattice -- afterwards the
 * device arrays are byte for byte what dust_hip_model_create builds from the same voxels. Asynchronous edits are not
 * offered: the call returns when the model is rebuilt (it reads sizes and bounds back). Scenes that instance the model must
 * be committed again (dust_hip_scene_commit) before they render: bounds and the staged root may have changed.
 * DUST_ERR_UNSUPPORTED for 4096^3 models. */
DustStatus dust_hip_model_set_voxels(DustHipModel*, const uint32_t* xyz, const int32_t* values, uint32_t n);
DustStatus dust_hip_model_get_voxels(DustHipModel*, const uint32_t* xyz, int32_t* values /* palette index, or -1 = None */, uint32_t n);
/* current size of a model's Block array and material stream, and a synchronous copy of both to the host */
DustStatus dust_hip_model_info(const DustHipModel*, uint32_t* n_blocks, uint64_t* n_materials);
DustStatus dust_hip_model_read(const DustHipModel*, DustHipBlock* blocks, uint32_t block_capacity, uint8_t* materials, uint64_t material_capacity);
/* Lifetimes: a scene keeps the models it instances alive, every model / scene / pipeline its context; destroying a scene or
 * pipeline first waits for the frames in flight on the context's stream. After *_destroy the CALLER must not use the handle
 * again, whatever else still holds the object. Calls on one context are not thread-safe against each other;
 * dust_hip_last_error() is per thread. */

/* TLASStore (render/src/accel_struct/tlas.rs:28-180) + the prev-frame transform vec (standard.rs:845-878) */
DustStatus dust_hip_scene_create(DustHipContext*, DustHipScene** out);
void dust_hip_scene_destroy(DustHipScene*);
/* tlas_system push (tlas.rs:79-128): returns gl_InstanceID == push order */
DustStatus dust_hip_scene_add_instance(DustHipScene*, const DustHipModel*, const float obj_to_world[12],
                                       const float prev_obj_to_world_mat4[16], uint32_t* instance_id);
DustStatus dust_hip_scene_set_transform(DustHipScene*, uint32_t instance_id, const float obj_to_world[12],
                                        const float prev_obj_to_world_mat4[16]);
/* TLAS build (tlas.rs:43-64, rebuilt inside the frame's command stream whenever an instance moved): must be called after
 * add / set_transform (and after editing an instanced model) before rendering. Asynchronous: the records of the instances
 * that changed are re-derived on the host and one stream-ordered copy from pinned memory carries them to the device behind
 * the frame in flight -- no allocation and no wait unless instances were added since the last commit. */
DustStatus dust_hip_scene_commit(DustHipScene*);
/* Host only (no device, no context): the two top-level structures dust_hip_scene_commit builds over the instances' world boxes --
 * what the reference hands to the driver as a TLAS (accel_struct/tlas.rs:37-117). boxes: n x {lo[3], hi[3]}.
 *   grid:  info->dim cells of info->cell from info->lo; cells[c] = first item | items << 20 for cell c = (z * dim[1] + y) * dim[0] + x;
 *          items: instance ids; ranges[2 i], ranges[2 i + 1]: the block of cells instance i is listed in, x | y << 9 | z << 18;
 *   slot_order: the instances along a Morton curve (the packet cull groups 64 consecutive ones when n > 256: info->n_groups).
 * Any output pointer may be NULL (info alone gives the sizes). For tests and tools. */
typedef struct DustTopLevelInfo {
  uint32_t struct_size;
  uint32_t dim[3];
  float lo[3], cell[3];
  uint32_t n_cells, n_items, n_groups;
} DustTopLevelInfo;
DustStatus dust_hip_top_level_build(const float* boxes, uint32_t n, DustTopLevelInfo* info, uint32_t* cells, size_t cells_capacity, uint16_t* items,
                                size_t items_capacity, uint32_t* ranges, uint32_t* slot_order);

/* the members of CameraSettings the shaders read (standard.rs:277-302,813-827; layout.playout:20-33) */
typedef struct DustHipCamera {
  float view_col0[3], view_col1[3], view_col2[3]; /* camera_view_col0..2 */
  float position[3];
  float tan_half_fov, far_, near_;
} DustHipCamera;


/* GBuffer planes (standard.rs:881-917; formats :974-1050) */
typedef enum DustHipPlane {
  DUST_PLANE_ILLUMINANCE = 0, /* RGBA16F, 8 B/px  (img_illuminance) */
  DUST_PLANE_DENOISED = 1,    /* RGBA16F, 8 B/px  (img_illuminance_denoised) */
  DUST_PLANE_ALBEDO = 2,      /* A2B10G10R10, 4 B/px */
  DUST_PLANE_NORMAL = 3,      /* A2B10G10R10, 4 B/px */
  DUST_PLANE_DEPTH = 4,       /* R32F, 4 B/px */
  DUST_PLANE_MOTION = 5,      /* RGBA16F, 8 B/px */
  DUST_PLANE_VOXEL_ID = 6,    /* R32UI, 4 B/px */
  DUST_PLANE_ACCUM = 7,       /* RGBA32F, 16 B/px: accumulated unpacked illuminance + frame count (DUST_PASS_ACCUMULATE: plain
                                 N-frame mean; DUST_PASS_DENOISE: the reprojected temporal accumulation, which IS the filter's history:
                                 the plane's device pointer then alternates between two buffers from frame to frame -- query it
                                 per frame, or bind the plane -- and a frame may ask for one of the two passes, not both) */
  DUST_PLANE_OUTPUT = 8,      /* RGBA16F, 8 B/px: tone-mapped display image (ToneMappingPipeline's dst) */
  DUST_PLANE_COUNT = 9
} DustHipPlane;

/* ray types (StandardPipeline::*_RAYTYPE, standard.rs:223-226) double as pass bits */
#define DUST_PASS_PRIMARY (1u << 0)            /* standard.rs:477-490 */
#define DUST_PASS_AMBIENT_OCCLUSION (1u << 1)  /* standard.rs:564-577 (sun shadow + AO ray) */
#define DUST_PASS_FINAL_GATHER (1u << 2)       /* standard.rs:627-640 */
#define DUST_PASS_SURFEL (1u << 3)             /* standard.rs:712-725 */
#define DUST_PASS_ACCUMULATE (1u << 4)         /* stands in for NRDPipeline::render (nrd.rs:272-617) */
#define DUST_PASS_DENOISE (1u << 5)            /* NRDPipeline::render (nrd.rs:272-617) as a native spatiotemporal filter: temporal
                                                  reprojection through the motion plane with disocclusion tests and antilag, then an
                                                  edge-aware blur; writes img_illuminance_denoised (and DUST_PLANE_ACCUM: the temporal
                                                  accumulation + frame count). Whole frames only. Settings: dust_hip_pipeline_set_denoiser */
#define DUST_PASS_COUNT_STATS (1u << 16)       /* run the counting build of the kernels (slower) */
#define DUST_PASS_GI_ORDERED (1u << 17)        /* apply the surfel pass's hash inserts in surfel-index order (bitwise
                                                  repeatable, serial); default: concurrently, as the reference's racy
                                                  shaders do (spatial_hash.glsl:147-195), statistically repeatable */
#define DUST_PASS_GI_SHARDED (1u << 18)        /* multi-GPU GI (see dust_hip_pipeline_gi_exchange): the final gather may
                                                  run on a row band; it records which hash entries it stamped and
                                                  leaves the surfel enqueues uncommitted for the exchange */

typedef struct DustHipFrameParams {
  uint32_t struct_size;
  uint32_t passes;       /* DUST_PASS_* */
  uint32_t frame_index;  /* push constant frame_index (standard.rs:252,457-463) */
  uint32_t rand;         /* push constant rand (standard.rs:449-456) */
  uint32_t row_begin, row_end; /* rows of the frame this call renders (multi-GPU bands); 0,0 = all */
  /* ---- appended in round 6 (a caller whose struct_size ends at row_end gets zeroes: the whole pool on this device) */
  uint32_t surfel_rank, surfel_world; /* DUST_PASS_SURFEL | DUST_PASS_GI_SHARDED with surfel_world >= 1: this call runs the pool's ordering and
                                  TRACES only rank surfel_rank's share of the position-ordered pool (see dust_hip_gi_surfel_exchange_run,
                                  which completes the pass); surfel_world == 0: the whole pass, as before */
} DustHipFrameParams;

typedef struct DustHipPassStats {
  float ms;                 /* kernel time from HIP events on the launch stream (needs DUST_HIP_CONTEXT_TIMING) */
  uint64_t rays;            /* rays issued (needs DUST_PASS_COUNT_STATS, else 0) */
  uint64_t instances_tested;
  uint64_t upper_descents;
  uint64_t mid_descents;
  uint64_t bricks_tested;
  uint64_t hits;
} DustHipPassStats;

/* StandardPipeline::new + use_gbuffer (standard.rs:90-173, :940-1065): persistent buffers and G-buffer */
DustStatus dust_hip_pipeline_create(DustHipContext*, uint32_t width, uint32_t height, DustHipPipeline** out);
void dust_hip_pipeline_destroy(DustHipPipeline*);
/* BlueNoise (noise.rs:7-56): texture 0 (scalar, R8) or 5 (unitvec3_cosine, RGBA8), 128 x 128 x layers */
DustStatus dust_hip_pipeline_set_noise(DustHipPipeline*, uint32_t texture, const uint8_t* texels, uint32_t layers);
/* StandardPipeline::render (standard.rs:228-810). Asynchronous on the context's stream.
 * DUST_ERR_NOT_READY while a noise texture a requested pass samples has not been set.
 * Every pass of the frame is enqueued before the call returns (nothing is kept back for a later call). The surfel pass
 * (DUST_PASS_SURFEL) only has to be complete before the NEXT frame's final gather reads the spatial hash, and it is latency-bound:
 * it is enqueued on a second stream the context owns, behind this frame's final gather, so that the next frame's primary / AO
 * kernels run beside it. It reads the scene and writes only the library-owned GI buffers; every library call that conflicts with
 * it (the next final gather, dust_hip_scene_commit, model edits, the GI state accessors) waits for it on the device, and
 * dust_hip_sync and every synchronous read-back wait for both streams -- a caller that orders its own work on the context's
 * stream (e.g. a collective that reads a bound plane) needs nothing more. DustHipPipelineConfig.side_stream = DUST_SIDE_STREAM_OFF keeps
 * the pass in place. */
DustStatus dust_hip_render_frame(DustHipPipeline*, const DustHipScene*, const DustHipCamera*, const DustHipSky*,
                                 const DustHipFrameParams*);
/* What the host does to the scene before a frame of dust_hip_render_frames: the transforms of the entities that moved since the previous frame
 * (tlas_system pushes them every frame, accel_struct/tlas.rs:79-128; castle.rs:287-291 moves the teapot). n == 0: the scene as the previous
 * frame left it. */
typedef struct DustHipFrameMoves {
  uint32_t n;                      /* instances moved before this frame */
  const uint32_t* instance_ids;    /* n ids (dust_hip_scene_add_instance) */
  const float* obj_to_world;       /* n x 12, as dust_hip_scene_set_transform takes them */
  const float* prev_obj_to_world;  /* n x 16 (the previous frame's transforms, for the motion vectors), or NULL */
} DustHipFrameMoves;
/* Frames in flight (rhyolite_bevy/src/lib.rs:58 `max_frame_in_flight: 3`; StandardPipeline::render is called once per frame and the frames
 * overlap on the device): n_frames frames in one call -- frame i with cameras[i], skies[i], params[i] into pipelines[i], after moves[i] (if
 * moves != NULL) have been applied to the scene and committed -- with exactly the results of, for every i in order,
 * dust_hip_scene_set_transform x moves[i].n + dust_hip_scene_commit + dust_hip_render_frame. Frames that qualify share ONE persistent launch
 * (up to 8 per launch, more are split): a wavefront that finds frame i without tiles goes straight on to frame i + 1, so the launch's tail,
 * the staging of the roots and the gap between launches are paid once for all of its frames (1080p primary + AO: 0.22 ms per frame alone,
 * 0.209 at four per launch, 0.2065 at eight). Qualifying: passes == DUST_PASS_PRIMARY | DUST_PASS_AMBIENT_OCCLUSION for every frame, distinct pipelines of
 * the scene's context with one frame size, one row band and the same DustHipPipelineConfig, no surfel pass outstanding. Every frame reads the
 * scene as ITS moves left it (the scene's ring of device images holds as many states as a launch has frames). Anything else (GI passes: a
 * frame's gather reads the hash its predecessor's surfel pass wrote) is enqueued frame after frame as dust_hip_render_frame would. Every
 * frame's arguments are checked before the scene is touched and the first frame is enqueued; params[i].struct_size must be
 * sizeof(DustHipFrameParams). With DUST_HIP_CONTEXT_TIMING the launch's time is reported by pipelines[0] (pass 0), once for all frames of
 * the launch. After the call the scene is as the last frame saw it. */
DustStatus dust_hip_render_frames(uint32_t n_frames, DustHipPipeline* const* pipelines, DustHipScene*, const DustHipCamera* cameras,
                                  const DustHipSky* skies, const DustHipFrameParams* params, const DustHipFrameMoves* moves);
/* pass: 0 primary, 1 AO-pass sun-shadow rays, 2 AO rays, 3 final gather, 4 surfel sun rays, 5 surfel cosine rays.
 * When primary and AO passes are requested together they run as ONE fused kernel: its time is reported under
 * pass 0 and passes 1-2 report ms = 0 (set DUST_HIP_NO_FUSE=1 to launch them separately). */
DustStatus dust_hip_pipeline_pass_stats(DustHipPipeline*, uint32_t pass, DustHipPassStats* out);
/* Kernel time over a run of frames (DUST_HIP_CONTEXT_TIMING): per pass kind -- 0 primary (or the fused primary + AO kernel), 1 AO,
 * 2 final gather (+ commit; not the regrouping pre-pass), 3 surfel pass (keys, sort, trace, apply) -- the summed HIP-event durations of its launches
 * since the last call with mark != 0 (at most the 256 most recent ones), and how many launches that was. Waits for the stream.
 * Nothing is synchronised per frame: the pairs are recorded into a ring on the launch stream and read here. */
DustStatus dust_hip_pipeline_kernel_times(DustHipPipeline*, int mark, float ms_sum[4], uint32_t launches[4]);
/* Work distribution feedback. The traversal kernels are persistent launches whose tiles (8 x 8 pixel packets; 64-entry chunks of
 * the regrouped gather and surfel lists) cost very different amounts; a launch records the shader-clock cycles each tile took,
 * and the next launch of the same pass hands its tiles out most expensive first (per XCD band), so that the launch does not end on a
 * few late, slow tiles. While camera, scene, sun and row band stay as they were the order is kept and re-measured every 8th launch at first, then every 16th, 32nd, 64th
 * only. The order never changes a result. This reads the map of the pass's last MEASURED launch (a profiling heat map):
 * pass_kind 0 primary (or fused primary + AO), 1 AO, 2 final gather, 3 surfel trace; cycles may be NULL to query the grid only
 * (tiles_x x tiles_y, 0 x 0 before the first launch). DUST_HIP_NO_TILE_ORDER=1 switches the feedback off. */
DustStatus dust_hip_pipeline_tile_costs(DustHipPipeline*, uint32_t pass_kind, uint32_t* cycles, uint32_t capacity, uint32_t* tiles_x,
                                        uint32_t* tiles_y);
DustStatus dust_hip_pipeline_plane_device_ptr(DustHipPipeline*, DustHipPlane, void** ptr, size_t* bytes);
/* Render target binding (the reference binds its G-buffer images per frame, standard.rs:974-1050): redirects one plane to
 * caller-owned device memory of at least the plane's size (16-byte aligned), or back to the pipeline's own storage with a
 * null pointer. Frames enqueued afterwards read and write the plane there -- e.g. alternate two illuminance buffers so that
 * frame k can be sent to another GPU while frame k+1 renders, without a copy. The memory must stay valid until those frames
 * have completed; ordering against the caller's own use is the caller's (same stream, or events). */
DustStatus dust_hip_pipeline_bind_plane(DustHipPipeline*, DustHipPlane, void* device_ptr, size_t bytes);
/* synchronous device-to-host copy of one plane */
DustStatus dust_hip_pipeline_read_plane(DustHipPipeline*, DustHipPlane, void* dst, size_t dst_bytes);
/* Persistent GI buffers (standard.rs:334-358): (re)allocates and resets the spatial hash (SpatialHashCapacity,
 * spatial_hash.glsl:1, default 32 Mi entries) and the surfel pool (SurfelPoolSize, surfel.glsl:2, default 345600).
 * Called implicitly with the defaults by the first frame that runs a GI pass. */
DustStatus dust_hip_pipeline_configure_gi(DustHipPipeline*, uint32_t hash_capacity, uint32_t surfel_pool_size);
/* synchronous copy of GI state to the host: which = 0 spatial hash ((capacity+2) x 12 B), 1 surfel pool (16 B each);
 * and its inverse, which restores a saved state into a pipeline configured with the same capacity and pool size (checkpoint /
 * resume of a converged hash: the reference keeps its hash for the life of the process, standard.rs:334-358) */
DustStatus dust_hip_pipeline_read_gi(DustHipPipeline*, uint32_t which, void* dst, size_t dst_bytes);
DustStatus dust_hip_pipeline_write_gi(DustHipPipeline*, uint32_t which, const void* src, size_t src_bytes);
/* Multi-GPU GI: every GPU keeps an identical spatial hash and surfel pool, the pixel passes run on row bands and the
 * (small) surfel pass is replicated. The reference has no multi-device path; the merge rule is this library's defined
 * order (final_gather.rchit:52-63 leaves the winner among the pixels aliasing a slot to a race): the highest pixel
 * index wins a surfel slot, as on one GPU. Per frame, on every rank:
 *   1. dust_hip_render_frame(PRIMARY | AMBIENT_OCCLUSION | FINAL_GATHER | DUST_PASS_GI_SHARDED, own row band)
 *   2. all-reduce MAX  of slot_owner (u32 x pool_size)                       -- the caller's collective (RCCL)
 *      all-gather      of the bands of `touched` (u32 per pixel, band r at row r * band_rows)
 *   3. dust_hip_gi_export(own band): merged[s] = the enqueued surfel if the winning pixel of slot s is in this band, else 0
 *   4. all-reduce SUM  of merged as i32 (16 B x pool_size; exactly one rank contributes per slot)
 *   5. dust_hip_gi_import(own band, frame_index): stamps last_accessed_frame of the entries the OTHER bands' final
 *      gather read, commits the winning surfels to the pool, clears slot_owner
 *   6. dust_hip_render_frame(SURFEL [| ACCUMULATE] | DUST_PASS_GI_SHARDED, rows as in 1 for ACCUMULATE)
 * With DUST_PASS_GI_ORDERED in step 6 every rank's hash and pool stay bit-identical to the single-GPU run.
 * Step 6 replicates the whole surfel pass on every rank -- 0.23 ms of the castle's 0.65 ms GI frame, which caps 8 ranks at 1.3-1.7 x.
 * Round 6 shards its TRACE (85 % of the pass) as well; the trace only reads the hash (identical on every rank after step 5):
 *   6a. dust_hip_render_frame(SURFEL | GI_ORDERED | GI_SHARDED [| ACCUMULATE], surfel_rank = r, surfel_world = N): orders the pool by
 *       position (replicated: 50 us) and traces the slots [r S, (r + 1) S) of that order, S = ceil(groups / N) x 64 -- the records
 *       {request 32 B, replacement 16 B, sun payload 16 B} go to staging arrays in SLOT order, a rank's share one contiguous run
 *   6b. dust_hip_gi_surfel_exchange_run: all-gather of the three arrays (64 B per slot: 22 MB for 345 600 slots), then on every rank
 *       the records move to their surfels, the trace's hash stamps are repeated, and the ordered apply runs (replicated: 30 us)
 * -- bit-identical to the single-device ordered run again (tests/test_gpu_comm.py, test_gpu_gi_sharded.py).
 * dust_hip_pipeline_gi_exchange allocates (once) and returns the three device buffers the collectives run on;
 * padded_rows >= height is the row count of `touched` (world_size x band_rows).
 * A rank whose band lies past the end of the frame (small frames on many GPUs) skips steps 1 and the ACCUMULATE of 6 but still takes
 * part in every collective: it exports and imports the EMPTY range (height, height) -- its `merged` must be zeroed by the export,
 * or it would add the previous frame's all-reduced sum to this frame's. */
typedef struct DustHipGiExchange {
  uint32_t struct_size;
  uint32_t pool_size;      /* surfel slots */
  uint32_t width;          /* pixels per row of `touched` */
  uint32_t touched_rows;   /* rows of `touched` */
  void* slot_owner;        /* u32[pool_size]: 1 + highest pixel index that enqueued into the slot this frame, 0 = none */
  void* touched;           /* u32[touched_rows * width]: 1 + index of the hash entry the pixel's final gather stamped, 0 = none */
  void* merged;            /* 16 B x pool_size: SurfelEntry per slot */
} DustHipGiExchange;
DustStatus dust_hip_pipeline_gi_exchange(DustHipPipeline*, uint32_t padded_rows, DustHipGiExchange* out);
DustStatus dust_hip_gi_export(DustHipPipeline*, uint32_t row_begin, uint32_t row_end);
DustStatus dust_hip_gi_import(DustHipPipeline*, uint32_t row_begin, uint32_t row_end, uint32_t frame_index);
/* AutoExposurePipeline::render + ToneMappingPipeline::render (pipeline/auto_exposure.rs:96-248, tone_mapping.rs:76-200;
 * auto_exposure.comp, auto_exposure_avg.comp, tone_map.comp): 256-bin log-luminance histogram of the denoised radiance
 * (after DUST_PASS_ACCUMULATE: the N-frame mean), exponential adaptation of the average, then
 * radiance x albedo / avg -> display primaries -> ACES fit -> transfer function into DUST_PLANE_OUTPUT. */
typedef struct DustHipToneMapParams {
  uint32_t struct_size;
  uint32_t transfer_function;      /* ColorSpaceTransferFunction 0..8 (rhyolite utils/format.rs:683-693): 0 linear, 1 sRGB, ... */
  float color_space_conversion[9]; /* COLOR_SPACE_CONVERSION_0..8, column-major: scene (ACES AP1) -> display primaries */
  float min_log_luminance, max_log_luminance, time_coefficient; /* ExposureSettings (auto_exposure.rs:225-248): -6, 8.5, 0.2 */
} DustHipToneMapParams;
DustStatus dust_hip_tone_map(DustHipPipeline*, const DustHipToneMapParams*);
/* reads (and optionally first overwrites) the adapted average luminance the tone mapper divides by */
DustStatus dust_hip_pipeline_exposure(DustHipPipeline*, float* avg_luminance, const float* set_to);
/* ReblurSettings + CommonSettings as far as this filter has the knob (nrd.rs:693-785; defaults = what the reference sets or
 * NRD's own defaults): the reference's denoiser is NVIDIA NRD, a closed SDK -- DUST_PASS_DENOISE is a native filter of the same
 * shape, not its arithmetic. */
typedef struct DustHipDenoiseParams {
  uint32_t struct_size;
  uint32_t max_accumulated_frames; /* ReblurSettings::maxAccumulatedFrameNum, 30 */
  float disocclusion_threshold;    /* CommonSettings::disocclusion_threshold, 0.01: plane distance / view distance */
  float antilag_sigma_scale;       /* ReblurAntilagSettings::luminance_sigma_scale, 2.0 (nrd.rs:777) */
  float antilag_power;             /* ReblurAntilagSettings::luminance_antilag_power, 0.8 (nrd.rs:778); 0 = no antilag */
  float max_blur_radius;           /* ReblurSettings::blurRadius, 15 pixels; 0 = temporal accumulation only */
} DustHipDenoiseParams;
DustStatus dust_hip_pipeline_set_denoiser(DustHipPipeline*, const DustHipDenoiseParams*);
/* DenoiserEvent::Restart (nrd.rs:749-755): discard the history; the next DUST_PASS_DENOISE frame starts a new accumulation */
DustStatus dust_hip_pipeline_restart_denoiser(DustHipPipeline*);
/* zero every plane, the accumulation count and the denoiser history */
DustStatus dust_hip_pipeline_clear(DustHipPipeline*);
/* How many frames the caller keeps in flight on this device, each on a pipeline and a context (stream) of its own -- the reference's
 * host runs up to three (rhyolite_bevy/src/lib.rs:58). The traversal kernels are persistent launches that take every workgroup slot
 * they are given and hold it until the launch's last tile is done; with n > 1 a launch of this pipeline's pixel passes starts about
 * 1/n of the slots, so that n frames' launches run side by side instead of each waiting behind the others' stragglers (row bands
 * of one frame on 8 GPUs, four in flight: 0.045 -> 0.037 ms per band). 1 (the default): a launch may take the whole device. 1..16. */
DustStatus dust_hip_pipeline_set_frames_in_flight(DustHipPipeline*, uint32_t n);
/* Everything that decides WHICH kernels a pipeline's frames run and on how much of the device -- what the reference's plugin takes as
 * its settings (RenderPlugin, crates/render/src/lib.rs:35-56; the frames in flight of rhyolite_bevy/src/lib.rs:58). Defaults (a
 * zeroed struct but for struct_size) are the production configuration; nothing here is read from the environment. May be called
 * between any two frames: launches enqueued afterwards follow it. */
#define DUST_GI_PATH_AUTO 0u     /* packets of 64 rays (k_final_gather, k_surfel_trace); the final gather of a scene that holds a 4096^3
                                    tree as a ray stream (one ray per lane, lanes refilled: DESIGN.md section 4) */
#define DUST_GI_PATH_PACKETS 1u  /* packets everywhere */
#define DUST_GI_PATH_STREAMS 2u  /* both GI passes as ray streams (measured slower on scenes of many instances) */
#define DUST_SIDE_STREAM_AUTO 0u /* the surfel pass on the context's second stream, beside the next frame's primary / AO kernel */
#define DUST_SIDE_STREAM_OFF 1u  /* in place, on the context's stream */
#define DUST_IN_FLIGHT_SHARE 0u  /* n frames in flight: every launch takes 1/n of the workgroup slots (row bands of an N-GPU frame) */
#define DUST_IN_FLIGHT_ALL 1u    /* every launch asks for ALL slots: the next frame's workgroups start on the CUs the previous frame's tail
                                    has left (whole frames on one GPU) */
#define DUST_RESERVE_AUTO 0xFFFFFFFFu
typedef struct DustHipPipelineConfig {
  uint32_t struct_size;
  uint32_t reserve_blocks;    /* workgroup slots (multiples of 8) the persistent traversal launches leave empty so that another queue's
                                 kernels -- RCCL's send / receive -- can become resident beside them (they hold every VGPR of the SIMDs
                                 they run on). DUST_RESERVE_AUTO (the default of dust_hip_pipeline_create): 0 until the pipeline takes part
                                 in a collective of a communicator with world > 1, 32 from then on. A value the device cannot spare (it keeps 8 slots) is ignored */
  uint32_t gi_path;           /* DUST_GI_PATH_* */
  uint32_t side_stream;       /* DUST_SIDE_STREAM_* */
  uint32_t side_share;        /* percent (5..90) of the slots the surfel pass takes on the second stream; 0 = calibrated from one timed frame */
  uint32_t frames_in_flight;  /* 1..16, as dust_hip_pipeline_set_frames_in_flight; 0 = leave as it is */
  uint32_t in_flight_slots;   /* DUST_IN_FLIGHT_* */
} DustHipPipelineConfig;
DustStatus dust_hip_pipeline_configure(DustHipPipeline*, const DustHipPipelineConfig*);
DustStatus dust_hip_pipeline_get_config(const DustHipPipeline*, DustHipPipelineConfig* out /* struct_size set by the caller */);

/* ===================================================================== multi-GPU: one process per GPU, RCCL over xGMI
 * The reference renders on one device; its plugin entry is where devices would be selected (crates/render/src/lib.rs:58-134).
 * north_star's partition: the pixels of a frame shard across the GPUs of a node as row bands (DustHipFrameParams.row_begin /
 * row_end; the scene is replicated), and the finished bands are gathered onto one GPU. A DustHipComm is this process's end of that:
 * an RCCL communicator bound to a context. librccl is opened on first use -- a single-GPU host never loads it. */
typedef struct DustHipComm DustHipComm;
#define DUST_HIP_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 makes the id and the host carries it to the other processes (any out-of-band way: a file, a socket, MPI) */
DustStatus dust_hip_comm_unique_id(uint8_t id[DUST_HIP_COMM_ID_BYTES]);
/* ncclCommInitRank on the context's device. Collective: every rank of the job calls it with the same id and world. */
DustStatus dust_hip_comm_create(DustHipContext*, uint32_t rank, uint32_t world, const uint8_t id[DUST_HIP_COMM_ID_BYTES], DustHipComm** out);
/* A LOOPBACK group: `world` ranks on ONE context (one device, one stream), out[0..world). The same entry points take these handles;
 * a collective is carried out -- with device copies and small reduction kernels, on the context's stream -- by the call that
 * completes it, i.e. when the group's last rank has made it, so every rank must make a call before any rank makes the next.
 * What a one-GPU box, the C++ host mirror and the tests drive the multi-GPU protocol with (each rank with a pipeline of its own). */
DustStatus dust_hip_comm_create_local(DustHipContext*, uint32_t world, DustHipComm** out);
void dust_hip_comm_destroy(DustHipComm*);
DustStatus dust_hip_comm_info(const DustHipComm*, uint32_t* rank, uint32_t* world, uint32_t* is_local);
/* Framebuffer gather. cuts: world + 1 row indices, 0 ... height, the same on every rank: rank r rendered rows [cuts[r], cuts[r+1]) of
 * `plane` (into the plane's current storage: the pipeline's own, or what dust_hip_pipeline_bind_plane gave it). Those rows travel to
 * rank `root` -- grouped ncclSend / ncclRecv, each peer over its own xGMI link -- into `dst` there (device memory of at least the
 * plane's size; NULL = the root pipeline's own plane, whose remaining rows are then filled in). Asynchronous: ordered behind
 * everything enqueued on the context's stream so far, carried out on the communicator's OWN stream, so that the next band frame
 * (rendering into another bound target) overlaps the transfer. *ticket (may be NULL) numbers the gather, from 1, per communicator.
 * Before a source target or `dst` is used again: dust_hip_comm_wait with that ticket. */
DustStatus dust_hip_gather_bands(DustHipPipeline*, DustHipComm*, DustHipPlane plane, const uint32_t* cuts, uint32_t root, void* dst, size_t dst_bytes,
                                 uint64_t* ticket);
/* The same for SEVERAL planes at once (plane_mask bit i = DustHipPlane i), each into the root pipeline's own plane: one RCCL group for
 * every plane and peer, one ticket. What a frame needs on the root before a pass that reads across rows -- the denoiser, which stands
 * in for the reference's NRD dispatch (crates/render/src/pipeline/nrd.rs:272-617; examples/castle.rs:190-231 runs render -> NRD -> tone
 * map in that order): gather illuminance | depth | normal | motion | voxel id (| denoised | albedo for the tone map), then
 * dust_hip_render_frame(root pipeline, passes = DUST_PASS_DENOISE) on the whole frame. */
DustStatus dust_hip_gather_planes(DustHipPipeline*, DustHipComm*, uint32_t plane_mask, const uint32_t* cuts, uint32_t root, uint64_t* ticket);
/* the context's stream waits (on the device) for gather `ticket` and every earlier one (0: for all enqueued so far) -- a host that
 * alternates two render targets waits for the gather that last read a target, not for the one still reading the other /
 * the host waits for every gather and for the context */
DustStatus dust_hip_comm_wait(DustHipComm*, uint64_t ticket);
DustStatus dust_hip_comm_sync(DustHipComm*);
/* Steps 2-5 of the multi-GPU GI protocol above (dust_hip_pipeline_gi_exchange), enqueued on the context's stream: all-reduce MAX of
 * slot_owner, all-gather of the `touched` bands (band r at row r * band_rows; every band is band_rows rows, the last may be
 * shorter in the frame), dust_hip_gi_export(row_begin, row_end), all-reduce SUM of `merged`, dust_hip_gi_import(..., frame_index).
 * A rank whose band lies past the end of the frame passes (height, height). */
DustStatus dust_hip_gi_exchange_run(DustHipPipeline*, DustHipComm*, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t frame_index);
/* Step 6b of the protocol: completes a surfel pass whose trace was sharded (dust_hip_render_frame with surfel_world >= 1, the same
 * world as the communicator's), on the context's stream: all-gather of the staged records (three ncclAllGather in one group; a loopback
 * group copies), then -- replicated on every rank -- records to their surfels + the trace's hash stamps, and the apply of the hash inserts
 * in surfel order (the deterministic apply of DUST_PASS_GI_ORDERED, whatever the frame asked for: every rank must end with the same hash).
 * A NULL communicator completes the pass with no exchange (a world of one; one emulated rank, whose peers' records are whatever the
 * staging arrays hold). DUST_ERR_NOT_READY without a sharded trace pending on the pipeline. */
DustStatus dust_hip_gi_surfel_exchange_run(DustHipPipeline*, DustHipComm*, uint32_t frame_index);

/* Device function evaluation: runs ONE of the device functions the traversal / shading kernels are built from on n
 * independent inputs (host arrays in, host arrays out, synchronous). The reference has no counterpart -- its shaders are
 * only reachable through vkCmdTraceRaysKHR -- this is how vectors of (ray, brick mask) -> (t, voxel) pairs and codec
 * round trips are checked against the device code directly (tests/golden/). Each input / output is a row of 32-bit
 * words (floats by bit pattern):
 *   fn  function (reference)                                          in words                          out words
 *   0   primary/hit.rint dda()            :43-131                      o[3] d[3] tmin mask_lo mask_hi    reported t voxel
 *   1   final_gather/ambient_occlusion.rint dda() :46-134              (same)                            (same; voxel 0xFF = threshold hit)
 *   2   final_gather/rough.rint dda()     :42-59                       (same)                            (same)
 *   3   EncodeRGBToLogLuv  (spatial_hash.glsl:28-60)                   rgb[3]                            packed
 *   4   DecodeLogLuvToRGB  (spatial_hash.glsl:64-93)                   packed                            rgb[3]
 *   5   NRD_FrontEnd_PackNormalAndRoughness -> A2B10G10R10 (nrd.glsl:25-52), roughness 1   n[3] materialID   texel
 *   6   NRD_FrontEnd_UnpackNormalAndRoughness (nrd.glsl:54-94)         texel                             n[3]
 *   7   REBLUR_FrontEnd_PackRadianceAndNormHitDist -> RGBA16F (nrd.glsl:127-147)   rgb[3] hitdist         2 words (4 halves)
 *   8   REBLUR_BackEnd_UnpackRadianceAndNormHitDist (nrd.glsl:107-125) 2 words                           rgb[3] hitdist
 *   9   float4 -> A2B10G10R10_UNORM                                    v[4]                              texel
 *   10  CubedNormalize + normal2FaceID (normal.glsl:9-18,39-43)        d[3]                              n[3] face
 *   11  rotateVectorByNormal (normal.glsl:31-37)                       n[3] target[3]                    v[3]
 *   12  (not per row) the surfel pass's stable radix sort of the n rows by key    key value              key value
 *   13  (not per row) the cost-ordered hand-out's sorter: the n rows are tile costs in tile order, 8 bands     cycles     tile index
 *   14  (not per row) the same with the bands cut at equal measured cost (what a frame uses), n >= 9           cycles     tile index, cut (the 9 band cuts in rows 0..8) */
DustStatus dust_hip_device_eval(DustHipContext*, uint32_t fn, const uint32_t* in, uint32_t in_words, uint32_t* out,
                                uint32_t out_words, uint32_t n);

#ifdef __cplusplus
}
#endif
#endif
