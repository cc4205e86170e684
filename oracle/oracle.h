/*
 * oracle.h -- CPU restatement of the Dust ray/path-tracing hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C restatement of the reference's algorithm, each function
 * citing the reference file:line it follows (paths relative to the reference checkout).
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (dust_amd/csrc, libdust_hip.so) never links, includes or calls anything here.
 *
 * Parity status (see DESIGN.md "Oracle"):
 *   - vdb tree / bitmask / pool / accessor: PINNED by the reference's own doctests and unit tests
 *     (crates/vdb/src/tree.rs:15-25, :87-101; bitmask.rs:81-90; pool.rs:22-42; accessor.rs:148-195).
 *   - loader flatten, DDA, shading, sky evaluation, spatial hash: the reference holds no test or
 *     golden vector for them and cannot be built here (Rust nightly + Vulkan RT + shaderc absent),
 *     so for those rows PARITY WITH THE REAL VULKAN OUTPUT IS UNPINNED; parity is defined against
 *     this restatement.
 *   - sky bake: pinned against fixtures generated in-container from the reference's own
 *     dataset.bin / datasetSolar.bin (tests/golden/make_sky_fixtures.py).
 */
#ifndef DUST_ORACLE_H
#define DUST_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ vdb */
typedef struct OrcTree OrcTree;
typedef struct OrcPool OrcPool;

/* BitMask<SIZE> (crates/vdb/src/bitmask.rs:3-124) */
void orc_bitmask_set(uint64_t* words, size_t index, int val);
int orc_bitmask_get(const uint64_t* words, size_t index);
/* iter_set_bits: writes ascending set-bit indices, returns count */
size_t orc_bitmask_iter(const uint64_t* words, size_t nwords, uint32_t* out, size_t cap);

/* Pool (crates/vdb/src/pool.rs:3-176) */
OrcPool* orc_pool_new(size_t item_size, unsigned chunk_size_log2);
void orc_pool_free_pool(OrcPool*);
uint32_t orc_pool_alloc(OrcPool*);
void orc_pool_free(OrcPool*, uint32_t index);
size_t orc_pool_num_chunks(const OrcPool*);
uint32_t orc_pool_count(const OrcPool*);

/* Tree<hierarchy!(l0, l1, ...)> (crates/vdb/src/tree.rs:7-124); log2s from root to leaf. */
OrcTree* orc_tree_new(const uint32_t* log2s, int nlevels);
void orc_tree_free(OrcTree*);
/* value: -1 = None (clear: reference is todo!() for internal nodes, we return -2), 0 = Some(false), 1 = Some(true) */
int orc_tree_set(OrcTree*, uint32_t x, uint32_t y, uint32_t z, int value);
/* returns -1 None, 0 Some(false), 1 Some(true) */
int orc_tree_get(const OrcTree*, uint32_t x, uint32_t y, uint32_t z);
/* Tree::iter(): voxel coordinates in iteration order; returns count (writes up to cap triples) */
size_t orc_tree_iter(const OrcTree*, uint32_t* xyz, size_t cap);
/* Tree::iter_leaf(): per leaf origin xyz, occupancy mask (first 64 bits), material_ptr; returns count */
size_t orc_tree_iter_leaf(const OrcTree*, uint32_t* xyz, uint64_t* mask, uint32_t* material_ptr, size_t cap);
/* set leaf.material_ptr in iter_leaf order (loader.rs:265-272 does this through iter_leaf_mut) */
void orc_tree_set_leaf_material_ptrs(OrcTree*, const uint32_t* ptrs, size_t n);
/* TreeMeta::META_MASK (tree.rs:154-167), same value on all three axes */
uint32_t orc_tree_meta_mask(const OrcTree*);
uint32_t orc_tree_root_level(const OrcTree*);
/* accessor.rs:15-30 */
uint32_t orc_lca_level(const uint32_t a[3], const uint32_t b[3], uint32_t mask, uint32_t root_level);
/* Accessor (accessor.rs:5-57): create, get */
typedef struct OrcAccessor OrcAccessor;
OrcAccessor* orc_accessor_new(const OrcTree*);
void orc_accessor_free(OrcAccessor*);
int orc_accessor_get(OrcAccessor*, uint32_t x, uint32_t y, uint32_t z);

/* ------------------------------------------------------------------ loader flatten */
/* GPUVoxNode / Block, 24 bytes (crates/vox/src/geometry.rs:40-49; assets/shaders/headers/sbt.glsl:1-19) */
typedef struct OrcBlock {
  uint16_t x, y, z, w;
  uint64_t mask;
  uint32_t material_ptr;
  uint32_t avg_albedo;
} OrcBlock;

typedef struct OrcModel {
  OrcBlock* blocks;
  uint32_t n_blocks;
  uint8_t* materials;
  uint64_t n_materials;
  uint8_t palette[255 * 4];
  uint32_t extent; /* tree extent per axis (256 for hierarchy!(4,2,2)) */
} OrcModel;

/* load_model (crates/vox/src/loader.rs:238-308) + ModelIndexCollector (collector.rs:2-88) +
 * VoxGeometry::from_tree (geometry.rs:55-179).
 * xyzi: n_voxels * 4 bytes in MagicaVoxel file axes, i = dot_vox 0-based palette index.
 * size: model.size (x,y,z in file axes). palette: 256 RGBA (dot_vox palette), first 255 used. */
OrcModel* orc_model_build(const uint8_t* xyzi, size_t n_voxels, const uint32_t size[3],
                          const uint8_t* palette_rgba256, const uint32_t* log2s, int nlevels);
void orc_model_free(OrcModel*);

/* ------------------------------------------------------------------ scene + shading */
typedef struct OrcInstance {
  uint32_t model;
  float obj_to_world[12]; /* 3x4 row-major, VkTransformMatrixKHR (accel_struct/tlas.rs:99-105) */
  float prev_obj_to_world[16]; /* Mat4 column-major, instances[] buffer (standard.rs:845-878) */
} OrcInstance;

typedef struct OrcCamera { /* the members of CameraSettings the shaders read (layout.playout:20-33) */
  float col0[3], col1[3], col2[3];
  float pos[3];
  float tan_half_fov, far_, near_;
} OrcCamera;

typedef struct OrcScene OrcScene;
OrcScene* orc_scene_new(void);
void orc_scene_free(OrcScene*);
/* model data is borrowed (caller keeps it alive) */
uint32_t orc_scene_add_model(OrcScene*, const OrcBlock* blocks, uint32_t n_blocks, const uint8_t* materials,
                             uint64_t n_materials, const uint8_t* palette255x4, uint32_t extent);
uint32_t orc_scene_add_instance(OrcScene*, const OrcInstance*);
/* build the traversal hierarchy used by the hierarchical mode (mode 1) */
void orc_scene_commit(OrcScene*);

/* the 56-float SkyModelState (pipeline/sky.rs:66-85, layout.playout:35-51) */
typedef struct OrcSky { float v[56]; } OrcSky;

/* G-buffer planes, w*h each (standard.rs:881-917, :974-1050). Storage emulates the image formats:
 * illuminance / denoised / motion: RGBA16F as 4 x uint16 half bits; albedo / normal: packed A2B10G10R10;
 * depth: f32; voxel_id: u32. */
typedef struct OrcGBuffer {
  uint32_t width, height;
  uint16_t* illuminance; /* 4 halves per pixel */
  uint16_t* denoised;    /* 4 halves per pixel */
  uint32_t* albedo;
  uint32_t* normal;
  float* depth;
  uint16_t* motion; /* 4 halves per pixel */
  uint32_t* voxel_id;
} OrcGBuffer;

typedef struct OrcRayStats { /* algorithmic-bytes accounting, SURVEY 8(d) */
  uint64_t rays;
  uint64_t instances_tested;
  uint64_t upper_descents; /* root + upper-internal children descended */
  uint64_t mid_descents;   /* 4^3 internal children descended (= bricks looked up) */
  uint64_t bricks_tested;
  uint64_t hits;
} OrcRayStats;

enum { ORC_MODE_BRUTE = 0, ORC_MODE_HIER = 1 };

/* single-ray entry points (used by the unit tests) */
/* primary/hit.rint dda(): returns 1 if reportIntersectionEXT was called; t, voxel out. kind: 0 primary, 1 ao */
int orc_dda(int kind, const float o[3], const float d[3], uint32_t mask_lo, uint32_t mask_hi, float tmin,
            float* t_out, uint32_t* voxel_out, int* hitkind_out);
/* rough.rint */
int orc_dda_rough(const float o[3], const float d[3], uint32_t mask_lo, uint32_t mask_hi, float* t_out);
/* camera.glsl:1-16 */
void orc_camera_ray_dir(const OrcCamera*, uint32_t px, uint32_t py, uint32_t w, uint32_t h, float out[3]);

/* trace one ray: raytype 0 primary dda, 1 ao dda, 2/3 rough; any_hit: terminate on first accepted hit.
 * returns 1 on hit; outputs t, instance, block (primitive id), voxel id. */
int orc_trace(const OrcScene*, int mode, int raytype, int any_hit, const float o[3], const float d[3], float tmin,
              float tmax, float* t, uint32_t* inst, uint32_t* block, uint32_t* voxel, OrcRayStats* stats);

/* passes over rows [y0, y1) of the frame */
void orc_pass_primary(const OrcScene*, int mode, const OrcCamera*, const OrcSky*, OrcGBuffer*, uint32_t y0,
                      uint32_t y1, OrcRayStats* stats);
/* ambient_occlusion.rgen (+rint/rchit/rmiss, nee.rmiss). noise5: 128*128 RGBA8 slice (unitvec3_cosine) */
void orc_pass_ao(const OrcScene*, int mode, const OrcCamera*, const OrcSky*, OrcGBuffer*, const uint8_t* noise5,
                 uint32_t rand, uint32_t y0, uint32_t y1, OrcRayStats* stats_sun, OrcRayStats* stats_ao);

/* hash-fed GI state: spatial hash (spatial_hash.glsl:1-220) + surfel pool (layout.playout:1-4, standard.rs:334-358) */
typedef struct OrcGI OrcGI;
OrcGI* orc_gi_new(uint32_t hash_capacity, uint32_t surfel_pool_size);
void orc_gi_free(OrcGI*);
void* orc_gi_hash_ptr(OrcGI*);  /* (capacity + 2) x {u32 fingerprint, u32 LogLuv, u16 last_frame, u16 count} */
void* orc_gi_pool_ptr(OrcGI*);  /* pool_size x {vec3 position, u32 direction} */
uint32_t orc_hash_fingerprint(const int32_t pos[3], uint32_t dir);
uint32_t orc_hash_location(const int32_t pos[3], uint32_t dir, uint32_t capacity);
void orc_hash_insert(OrcGI*, const int32_t pos[3], uint32_t dir, const float value[3], uint32_t frame_index);
int orc_hash_get(OrcGI*, const int32_t pos[3], uint32_t dir, uint32_t frame_index, float value[3], uint32_t* count);
/* final_gather.rgen/.rchit/.rmiss + rough.rint; noise0: 128*128 R8 slice, noise5: 128*128 RGBA8 slice */
void orc_pass_final_gather(const OrcScene*, int mode, const OrcCamera*, const OrcSky*, OrcGBuffer*, const uint8_t* noise0,
                           const uint8_t* noise5, uint32_t rand, uint32_t frame_index, OrcGI*, uint32_t y0, uint32_t y1,
                           OrcRayStats* stats);
/* the same pass on n_threads host threads, with the serial pass's result (surfel enqueues are logged per row band and applied in
 * row-major order afterwards); for frames at the reference's size */
void orc_pass_final_gather_mt(const OrcScene*, int mode, const OrcCamera*, const OrcSky*, OrcGBuffer*, const uint8_t* noise0,
                              const uint8_t* noise5, uint32_t rand, uint32_t frame_index, OrcGI*, uint32_t y0, uint32_t y1,
                              uint32_t n_threads, OrcRayStats* stats);
/* surfel.rgen/.rchit/.rmiss + surfel/nee.rmiss */
void orc_pass_surfel(const OrcScene*, int mode, const OrcSky*, const uint8_t* noise0, const uint8_t* noise5, uint32_t rand,
                     uint32_t frame_index, OrcGI*, OrcRayStats* stats_sun, OrcRayStats* stats_cos);

/* phase 1 (trace + hash reads) on n_threads host threads, phase 2 (inserts in surfel order) as in the serial pass: same result */
void orc_pass_surfel_mt(const OrcScene*, int mode, const OrcSky*, const uint8_t* noise0, const uint8_t* noise5, uint32_t rand,
                        uint32_t frame_index, OrcGI*, uint32_t n_threads, OrcRayStats* stats_sun, OrcRayStats* stats_cos);

/* auto exposure + tone map (auto_exposure.comp, auto_exposure_avg.comp, tone_map.comp) */
void orc_exposure_histogram(const uint16_t* illuminance, uint32_t w, uint32_t h, float min_log, float log_range, uint32_t hist[256]);
float orc_exposure_average(uint32_t hist[256], uint32_t w, uint32_t h, float min_log, float log_range, float time_coeff, float avg);
void orc_tone_map(const uint16_t* src, const uint32_t* albedo, uint32_t w, uint32_t h, float avg, const float conv[9], uint32_t tf,
                  uint16_t* dst);

/* The product's spatiotemporal accumulation filter (dust_amd/csrc/denoise.hip), restated in denoise.c. PARITY UNPINNED
 * against the reference: its denoiser is NVIDIA NRD, closed (nrd.rs:272-617). One frame: the temporal pass fills hist_out_*
 * from hist_in_* and the current planes, the spatial pass writes `denoised` for hit pixels. */
typedef struct OrcDenoise {
  uint32_t width, height, frame_index, have_history;
  OrcCamera cam, prev;
  uint32_t max_accumulated_frames;
  float disocclusion_threshold, antilag_sigma_scale, antilag_power, max_blur_radius;
  const uint16_t* illuminance; /* 4 halves / px */
  uint16_t* denoised;          /* 4 halves / px */
  const uint32_t* normal;
  const float* depth;
  const uint16_t* motion;      /* 4 halves / px */
  const uint32_t* voxel_id;
  const float* hist_in_accum;  /* rgb + frame count */
  const float* hist_in_depth;
  const uint32_t* hist_in_normal;
  const uint32_t* hist_in_id;
  float* hist_out_accum;
  float* hist_out_depth;
  uint32_t* hist_out_normal;
  uint32_t* hist_out_id;
} OrcDenoise;
void orc_denoise(const OrcDenoise*);

/* encodings (headers/nrd.glsl, color.glsl, spatial_hash.glsl, normal.glsl) for unit tests */
uint32_t orc_pack_rgb10a2(const float v[4]);
void orc_unpack_rgb10a2(uint32_t p, float v[4]);
uint16_t orc_f32_to_f16(float);
float orc_f16_to_f32(uint16_t);
void orc_nrd_pack_normal(const float n[3], float roughness, float material_id, float out[4]);
void orc_nrd_unpack_normal(const float p[4], float out_n[3]);
uint32_t orc_normal2faceid(const float n[3]);
void orc_cubed_normalize(const float d[3], float out[3]);
void orc_rotate_by_normal(const float n[3], const float v[3], float out[3]);
uint32_t orc_logluv_encode(const float rgb[3]);
void orc_logluv_decode(uint32_t p, float rgb[3]);
uint32_t orc_pcg(uint32_t);
uint32_t orc_xxhash32(uint32_t);
void orc_sky_radiance(const OrcSky*, const float dir[3], float out[3]);
void orc_sun_radiance(const OrcSky*, const float dir[3], float out[3]);

#ifdef __cplusplus
}
#endif
#endif
