// capi.cpp -- the extern "C" boundary (include/dust_hip.h) and the host runtime behind it:
// device-side VDB hierarchy build, instance records, persistent pipeline buffers, pass scheduling.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <memory>
#include <new>
#include <string>
#include <vector>

#include "capi_internal.hpp"
#include "dust_dev.h"
#include "vdb.hpp"
#include "png.hpp"
#include "vox.hpp"
#include "sky.hpp"
#include "edit.hpp"
#include "denoise.hpp"
#include <unordered_set>

namespace dust {
hipError_t launch_primary(const FrameArgs& a, uint32_t grid, uint32_t block, bool count, hipStream_t);
hipError_t launch_ambient_occlusion(const FrameArgs& a, uint32_t grid, uint32_t block, bool count, hipStream_t);
hipError_t launch_primary_ao(const FrameArgs& a, uint32_t grid, uint32_t block, bool count, hipStream_t);
hipError_t launch_primary_ao_batch(const FrameArgs* frames, uint32_t n, uint32_t grid, uint32_t block, hipStream_t);
hipError_t launch_final_gather(const FrameArgs& a, uint32_t grid, uint32_t block, bool count, bool commit, hipStream_t);
hipError_t launch_final_gather_shade(const FrameArgs& a, bool commit, hipStream_t);
hipError_t launch_gather_order(const FrameArgs& a, uint32_t n_tiles, hipStream_t);
hipError_t launch_gi_export(const FrameArgs& a, hipStream_t);
hipError_t launch_gi_import(const FrameArgs& a, hipStream_t);
hipError_t launch_surfel_keys(const FrameArgs& a, hipStream_t);
size_t radix_sort_scratch_bytes(uint32_t n);
hipError_t radix_sort_pairs(void* scratch, uint32_t* keys_a, uint32_t* vals_a, uint32_t* keys_b, uint32_t* vals_b, uint32_t n,
                            uint32_t key_bits, bool* in_b, hipStream_t s);
hipError_t launch_surfel_trace(const FrameArgs& a, uint32_t grid, uint32_t block, bool count, hipStream_t);
hipError_t launch_surfel_apply(const FrameArgs& a, int mode, hipStream_t);
hipError_t launch_surfel_unstage(const FrameArgs& a, hipStream_t);
hipError_t launch_gather_rays(const FrameArgs& a, hipStream_t);
hipError_t launch_surfel_rays(const FrameArgs& a, hipStream_t);
hipError_t launch_surfel_shade(const FrameArgs& a, hipStream_t);
hipError_t launch_ray_walk(const FrameArgs& a, int rt, uint32_t grid, uint32_t block, bool count, hipStream_t);
hipError_t launch_accumulate(const FrameArgs& a, hipStream_t);
hipError_t launch_tone_map(const uint16_t* src, const uint32_t* albedo, uint16_t* dst, uint32_t n_pixels, uint32_t* hist, float* avg,
                           float min_log, float log_range, float time_coeff, const float conv[9], uint32_t tf, hipStream_t s);
hipError_t configure_kernels(size_t max_lds);
constexpr uint32_t kTileOrderMaxBand = 65536;  // (kernels.hip)
hipError_t launch_tile_order(const uint32_t* cost, uint32_t* order, uint32_t* cuts, bool reuse_cuts, uint32_t total, uint32_t per, hipStream_t s);
hipError_t launch_cost_blend(const uint32_t* raw, uint32_t* smooth, uint32_t total, uint32_t keep_shift, hipStream_t s);
hipError_t launch_cost_dilate(const uint32_t* in, uint32_t* out, uint32_t tiles_x, uint32_t tiles_y, hipStream_t s);
hipError_t launch_device_eval(uint32_t fn, const uint32_t* in, uint32_t in_words, uint32_t* out, uint32_t out_words, uint32_t n, hipStream_t s);
}  // namespace dust

namespace {
// sky.glsl:81-113 (sun disc radiance, limb darkening) at the sun's own direction, times the solid-angle factor of
// nee.rmiss:11-22, in single precision like the shader. The kernels read the result as a launch constant.
void sun_constants(const float* s, float* dir, float* term) {
  const float len = std::sqrt((s[48] * s[48] + s[49] * s[49]) + s[50] * s[50]);
  const float d[3] = {s[48] / len, s[49] / len, s[50] / len};
  for (int k = 0; k < 3; ++k) { dir[k] = d[k]; term[k] = 0.0f; }
  const float len2 = std::sqrt((d[0] * d[0] + d[1] * d[1]) + d[2] * d[2]);  // the shader normalises the direction again
  const float e[3] = {d[0] / len2, d[1] / len2, d[2] / len2};
  const float cos_gamma = (e[0] * s[48] + e[1] * s[49]) + e[2] * s[50];
  if (!(cos_gamma >= 0.0f) || e[1] < 0.0f) return;
  const float sol_rad_sin = std::sin(s[55]);
  const float ar2 = 1.0f / (sol_rad_sin * sol_rad_sin);
  const float singamma = 1.0f - cos_gamma * cos_gamma;
  const float sc2 = 1.0f - (ar2 * singamma) * singamma;
  if (!(sc2 > 0.0f)) return;
  const float sc = std::sqrt(sc2);
  float dark[3] = {s[10] + s[11] * sc, s[26] + s[27] * sc, s[42] + s[43] * sc};
  float cur = sc;
  for (int i = 0; i < 4; ++i) {
    cur *= sc;
    dark[0] += s[12 + i] * cur; dark[1] += s[28 + i] * cur; dark[2] += s[44 + i] * cur;
  }
  const float v[3] = {s[52] * dark[0], s[53] * dark[1], s[54] * dark[2]};
  const float kk = 1.0f - std::cos(s[55]);
  term[0] = ((1.6410228f * v[0] + -0.32480323f * v[1]) + -0.23642465f * v[2]) * kk;   // color.glsl:24-31
  term[1] = ((-0.66366285f * v[0] + 1.6153315f * v[1]) + 0.016756356f * v[2]) * kk;
  term[2] = ((0.011721907f * v[0] + -0.0082844375f * v[1]) + 0.9883947f * v[2]) * kk;
}
}  // namespace

namespace {

thread_local std::string g_last_error;

DustStatus fail(DustStatus s, const std::string& msg) {
  g_last_error = msg;
  return s;
}
DustStatus hip_fail(hipError_t e, const char* what) {
  return fail(DUST_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                   \
  do {                                                  \
    hipError_t e_ = (expr);                             \
    if (e_ != hipSuccess) return hip_fail(e_, #expr);   \
  } while (0)

// struct_size is the caller's sizeof of a versioned struct: at least the layout this build knows (a newer caller may pass
// more; the known prefix is what is read)
template <class T>
bool struct_ok(const T* s) { return s->struct_size >= sizeof(T); }
#define STRUCT_TRY(ptr, name) \
  do { if (!struct_ok(ptr)) return fail(DUST_ERR_INVALID_ARGUMENT, name ".struct_size is smaller than this library's " name); } while (0)

template <class F>
DustStatus guarded(F&& f) {  // nothing may unwind across the C boundary
  try {
    return f();
  } catch (const dust::vox::ParseError& e) {
    return fail(e.unsupported ? DUST_ERR_UNSUPPORTED : DUST_ERR_PARSE, e.what);
  } catch (const std::bad_alloc&) {
    return fail(DUST_ERR_OUT_OF_MEMORY, "host allocation failed");
  } catch (const std::exception& e) {
    return fail(DUST_ERR_INVALID_ARGUMENT, e.what());
  } catch (...) {
    return fail(DUST_ERR_INVALID_ARGUMENT, "unknown error");
  }
}

struct DeviceBuffer {
  void* p = nullptr;
  size_t bytes = 0;
  DeviceBuffer() = default;
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  ~DeviceBuffer() { release(); }
  void release() { if (p) (void)hipFree(p); p = nullptr; bytes = 0; }
  hipError_t alloc(size_t n) {
    if (p) { (void)hipFree(p); p = nullptr; }
    bytes = n;
    return hipMalloc(&p, n ? n : 16);
  }
  // Host -> device on the CONTEXT's stream, then wait: a blocking hipMemcpy is a null-stream operation, which a
  // hipStreamNonBlocking stream is not ordered against (and from pageable memory it may return before the DMA has landed).
  hipError_t upload(const void* src, size_t n, hipStream_t st) {
    hipError_t e = alloc(n);
    if (e != hipSuccess || n == 0) return e;
    e = hipMemcpyAsync(p, src, n, hipMemcpyHostToDevice, st);
    if (e != hipSuccess) return e;
    return hipStreamSynchronize(st);
  }
};

// Copies between host and device go through the context's stream and wait for it: the blocking hipMemcpy / hipMemset are
// null-stream operations, and the context's stream is created hipStreamNonBlocking, i.e. NOT ordered against those.
hipError_t copy_wait(void* dst, const void* src, size_t n, hipMemcpyKind kind, hipStream_t st) {
  if (n == 0) return hipSuccess;
  const hipError_t e = hipMemcpyAsync(dst, src, n, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

// Lifetimes. Every handle of the device side is reference-counted inside the library: a model, scene or pipeline keeps its
// context alive, a scene keeps the models it instances alive. dust_hip_*_destroy gives up the CALLER's reference; the object
// (and its device memory) goes when the last user does. So handles may be destroyed in any order -- a garbage collector
// finalising a context before its models (Python's cycle collector does exactly that, in creation order) is fine.
struct RefCounted {
  std::atomic<uint32_t> refs{1};
};
template <class T> T* retain(T* o) { if (o) o->refs.fetch_add(1, std::memory_order_relaxed); return o; }
// (release() per type below: what dies with the last reference differs)

}  // namespace

struct DustVdbTree { dust::vdb::Tree tree; DustVdbTree(const uint32_t* f, int n) : tree(f, n) {} };
struct DustVdbAccessor { dust::vdb::Tree::Accessor acc; explicit DustVdbAccessor(const dust::vdb::Tree& t) : acc(t) {} };
struct DustVdbPool { dust::vdb::Pool pool; DustVdbPool(size_t b, unsigned c) : pool(b, c) {} };
struct DustVoxScene { dust::vox::Scene scene; };
struct DustSkyDataset { dust::sky::Dataset data; };

struct DustHipContext : RefCounted {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  uint32_t lds_root_bytes = 64 * 1024;
  bool timing = false;
  uint32_t timing_stride = 1;  // DUST_HIP_CONTEXT_TIMING_SPARSE: event pairs around the launches of every 4th frame only
  int num_cus = 256;
  size_t max_lds = 64 * 1024;
  DeviceBuffer srgb_lut;  // edit.hip: avg_albedo's linear->sRGB curve per (voxel count, colour sum), built on first use
  uint64_t sync_epoch = 1;  // bumped whenever the library has waited for the stream: what was enqueued before is done
  // The surfel pass of a frame runs on a second stream of the context (run_surfel_pass): it is launched in its own frame, behind
  // that frame's final gather, and only has to be complete before the NEXT final gather reads the hash -- so the next frame's
  // primary / AO kernel runs beside it, each side on its share of the workgroup slots. Nothing is kept back:
  // what conflicts with it on the main stream (the next gather, a scene commit, anything that touches the GI state) waits
  // for `ev_side_done` first (join_side); every wait for the context covers both streams (sync_stream).
  hipStream_t side = nullptr;
  hipEvent_t ev_fork = nullptr, ev_side_done = nullptr;
  bool side_busy = false;
  std::vector<hipStream_t> extra_streams;  // communicators' gather streams (comm.hip): they read pipelines' planes, so every wait for the context covers them
  hipStream_t copy = nullptr;  // scene commits upload on a stream of their own (the copy engine), beside the frame in flight -- never between two frames
  // Which frames of the stream have STARTED (FrameArgs::started_word): a word of pinned host memory that the first traversal launch of
  // every frame writes its sequence number into. `frame_seq` counts those launches as they are enqueued. 0 / null: not available.
  volatile uint32_t* started = nullptr;
  uint32_t frame_seq = 0;
};
// wait for everything enqueued on the context's stream (and remember that we did: scene commits recycle their pinned staging
// slots by this, without an event per commit)
static hipError_t sync_stream(DustHipContext* c) {
  hipError_t e = hipStreamSynchronize(c->stream);
  if (e == hipSuccess && c->side) { e = hipStreamSynchronize(c->side); c->side_busy = false; }
  for (hipStream_t x : c->extra_streams)
    if (e == hipSuccess) e = hipStreamSynchronize(x);
  if (e == hipSuccess) ++c->sync_epoch;
  return e;
}
// main stream: wait (on the device) for the side stream's pass, if one may still be running
static hipError_t join_side(DustHipContext* c) {
  if (!c->side_busy) return hipSuccess;
  c->side_busy = false;
  return hipStreamWaitEvent(c->stream, c->ev_side_done, 0);
}
// side stream: everything enqueued on the main stream so far comes first
static hipError_t fork_side(DustHipContext* c) {
  hipError_t e = hipSuccess;
  if (!c->side) {
    e = hipStreamCreateWithFlags(&c->side, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&c->ev_side_done, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  e = hipEventRecord(c->ev_fork, c->stream);
  return e != hipSuccess ? e : hipStreamWaitEvent(c->side, c->ev_fork, 0);
}
static void release(DustHipContext* c) {
  if (!c || c->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  (void)hipSetDevice(c->device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->side) { (void)hipStreamSynchronize(c->side); (void)hipStreamDestroy(c->side); }
  if (c->copy) { (void)hipStreamSynchronize(c->copy); (void)hipStreamDestroy(c->copy); }
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_side_done) (void)hipEventDestroy(c->ev_side_done);
  c->srgb_lut.release();
  if (c->started) (void)hipHostFree(const_cast<uint32_t*>(c->started));
  if (c->own_stream && c->stream) (void)hipStreamDestroy(c->stream);
  delete c;
}

// device-side voxel edits (edit.hip): the dense voxel grid and the scratch tables of the rebuild, created on a model's first edit
struct EditState {
  DeviceBuffer grid, brick_mask, flag_leaf, count_major, scan_tmp, header, xyz, values;
  uint32_t batch_capacity = 0;
};

struct DustHipModel : RefCounted {
  DustHipContext* ctx = nullptr;  // retained
  DeviceBuffer root, l2, l2_cells, mid, dense_mask, blocks, materials, palette;
  std::vector<uint8_t> host_root;  // 640 B: mask + prefix, what the kernels stage in LDS
  dust::DevModel dev{};
  uint32_t id = 0;
  uint64_t n_materials = 0;
  uint32_t generation = 0;  // bumped by every edit: scenes record it at commit and refuse to render a stale copy
  bool has_material_255 = false;  // the edit grid stores palette index + 1 in a byte: such a model cannot become editable
  std::unique_ptr<EditState> edit;
};
static void release(const DustHipModel* cm) {
  DustHipModel* m = const_cast<DustHipModel*>(cm);
  if (!m || m->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  DustHipContext* c = m->ctx;
  (void)hipSetDevice(c->device);
  (void)sync_stream(c);  // launches that read the arrays are done before they go
  delete m;
  release(c);
}

struct HostInstance {
  const DustHipModel* model;  // retained
  float o2w[12];
  float prev[16];
};

// Where a committed scene lives on the device: ONE allocation, the arrays at offsets inside it -- what depends on the models
// first, what depends on the instance transforms behind it. A commit fills a pinned host image of the same layout and sends
// it (all of it after a structural change, the transform-dependent tail otherwise) with one asynchronous copy on the context's
// stream (tlas.rs:37-65 rebuilds the TLAS inside the frame's command stream the same way): no allocation, no wait, and the
// kernels' pointers stay what they were until instances are added.
struct SceneLayout {
  size_t models = 0, root_table = 0, instances = 0, boxes = 0, visits = 0, enters = 0, gboxes = 0, sboxes = 0, grid_cells = 0, grid_items = 0, total = 0;
  size_t cap_cells = 0, cap_items = 0;  // entries the two grid sections hold (a commit that needs more lays the image out again)
  static SceneLayout make(size_t n_inst, size_t n_models, size_t n_roots, size_t n_cells, size_t n_items) {
    SceneLayout l;
    auto place = [&l](size_t bytes) { const size_t at = l.total; l.total = (l.total + bytes + 255) & ~size_t(255); return at; };
    l.models = place(n_models * sizeof(dust::DevModel));
    l.root_table = place(n_roots * dust::kN16LdsBytes);
    l.instances = place(n_inst * sizeof(dust::DevInstance));
    l.boxes = place((n_inst + 1) * sizeof(dust::DevBox));
    l.visits = place((n_inst + 1) * sizeof(dust::DevVisit));
    l.enters = place((n_inst + 1) * sizeof(dust::DevEnter));
    l.gboxes = place(((n_inst + 63) / 64 + 1) * sizeof(dust::DevBox));  // the packet cull's hierarchy (scenes beyond kFlatCullMax instances)
    l.sboxes = place((n_inst + 1) * sizeof(dust::DevBox));
    // the top-level grid last, with room to spare: its size follows the instances' positions, not only their number
    l.cap_cells = n_cells + n_cells / 2 + 64;
    l.cap_items = n_items + n_items / 2 + 256;
    l.grid_cells = place((l.cap_cells + 4) * sizeof(uint32_t));
    l.grid_items = place((l.cap_items + 8) * sizeof(uint16_t));
    return l;
  }
};

struct DustHipScene : RefCounted {
  DustHipContext* ctx = nullptr;  // retained
  std::vector<HostInstance> instances;
  std::vector<uint8_t> dirty;                // per instance: transform changed since the last commit
  std::vector<const DustHipModel*> models;   // distinct models, index == DevModel slot (kept alive through `instances`)
  std::vector<uint32_t> model_generation;    // their edit generations when the scene was committed
  std::vector<uint32_t> instance_slot;       // per instance: its model's slot
  bool structure_dirty = true;               // instances were added (or a model edited): slots, roots and capacity are re-derived
  // The device image is a RING of kImages copies, each with a pinned host twin. A commit writes the whole image into the next
  // slot -- on the context's copy stream, waited for by the host, so nothing is enqueued between two frames on the launch stream
  // (one stream-ordered copy per frame used to cost a moving scene ~25 us of a 230 us frame: wait for the frame, copy, start the
  // next) -- and frames enqueued from then on read that slot. A slot is rewritten kImages commits later: the frames that read it
  // are done if the library has waited for the streams since they were enqueued (a frame loop does, to read its result or pace
  // itself); otherwise the host is kImages commits ahead of the GPU and waits here (the reference's host runs <= 3 frames ahead).
#ifndef DUST_SCENE_IMAGES
#define DUST_SCENE_IMAGES 16   // (8 until dust_hip_render_frames took moves: a launch of eight frames, each with an image of its own, left the host no image to
#endif                        //  prepare the next launch in while that one ran -- 0.2457 ms per frame of a moving view against 0.2257 with 16; an image is ~150 KB for the castle)
  static constexpr int kImages = DUST_SCENE_IMAGES;
  struct Slot {
    DeviceBuffer dev;
    void* host = nullptr;
    mutable uint64_t epoch = 0;  // the context's sync_epoch when a frame reading the slot was last enqueued
    mutable uint32_t last_seq = 0;  // ... and that frame's start sequence number (DustHipContext::frame_seq), 0 = it has none (no traversal launch, or no word)
  } slots[kImages];
  int current = -1;           // the slot frames read
  uint32_t next_slot = 0;
  SceneLayout layout;
  size_t image_capacity = 0;  // bytes per slot
  std::vector<uint8_t> master;   // host master copy of the image (dirty instances are re-derived in place)
  float world_min[3] = {0, 0, 0}, world_max[3] = {0, 0, 0};  // union of the instances' world boxes
  // the top-level grid over the instance boxes (dust_dev.h DevGrid; rebuilt by every commit): its header, and the two arrays
  // that are copied into the image
  dust::DevGrid grid{};
  bool grid_valid = true;            // false: some cell would list more instances than a cell word counts (the ray streams then stay off)
  std::vector<uint32_t> grid_cells;
  std::vector<uint16_t> grid_items;
  std::vector<uint32_t> slot_order;  // large scenes: the instances along a space-filling curve (made by a structural commit; a moved instance keeps its slot)
  uint32_t n_groups = 0;             // ... and how many groups of 64 consecutive slots (0: the cull tests every box)
  std::vector<float> world_boxes;  // per instance {lo[3], hi[3]}: what derive_instance writes into the image, kept for the grid
  uint32_t n_lds_models = 0;
  uint64_t revision = 0;  // bumped by every commit (what the cost-ordered hand-out keys its view on)
  bool committed = false;
  const uint8_t* dev(size_t off) const { return static_cast<const uint8_t*>(slots[current].dev.p) + off; }
  void touch() const { slots[current].epoch = ctx->sync_epoch; slots[current].last_seq = 0; }  // a frame reading the current slot is being enqueued
  void free_images() {  // (the caller has waited for the streams)
    for (Slot& sl : slots) {
      if (sl.host) { (void)hipHostFree(sl.host); sl.host = nullptr; }
      sl.dev.release();
      sl.epoch = 0; sl.last_seq = 0;
    }
    current = -1;
    image_capacity = 0;
  }
};
static void release(const DustHipScene* cs) {
  DustHipScene* s = const_cast<DustHipScene*>(cs);
  if (!s || s->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
  DustHipContext* c = s->ctx;
  (void)hipSetDevice(c->device);
  (void)sync_stream(c);  // both streams: the surfel pass reads the scene image on the second one
  s->free_images();
  for (HostInstance& hi : s->instances) release(hi.model);
  delete s;
  release(c);
}

// What decides which kernels a pipeline's frames run. Two kinds, kept apart:
//  * PRODUCTION knobs arrive through the C ABI (DustHipPipelineConfig, dust_hip_pipeline_configure): slots left free for other queues'
//    kernels, which form the GI passes take, the surfel pass's second stream and its share, how several frames in flight split the slots;
//  * DIAGNOSTIC switches (ablations, A/B runs, the stress drivers' random draws) come from the environment, read ONCE when a pipeline
//    is created through the one lookup below. None of them is needed for any production frame; DESIGN.md section 9 lists them.
static const char* diag_env(const char* name) {   // "NO_FUSE" -> $DUST_HIP_NO_FUSE (the library's only environment lookup besides DUST_HIP_DEBUG-class reads here)
  char full[64];
  std::snprintf(full, sizeof full, "DUST_HIP_%s", name);
  return std::getenv(full);
}
struct Tuning {
  // ---- production (DustHipPipelineConfig)
  uint32_t reserve_blocks = 0xFFFFFFFFu;  // workgroup slots left free for another queue's kernels; 0xFFFFFFFF = auto (32 once the pipeline has taken part in a collective of world > 1)
  uint32_t gi_path = DUST_GI_PATH_AUTO;   // packets / ray streams (auto: packets, except the final gather of a scene with a 4096^3 tree)
  bool no_side_stream = false;            // the surfel pass on the main stream, in place
  uint32_t side_share = 0;                // percent of the workgroup slots the surfel pass takes on the second stream (0: calibrated)
  uint32_t in_flight_slots = DUST_IN_FLIGHT_SHARE;  // frames in flight: every launch on 1/n of the slots, or every launch asking for all of them
  // ---- diagnostics (environment)
  uint32_t debug = 0;           // DEBUG ablation bits (FrameArgs::debug)
  uint32_t block = 512;         // BLOCK: threads per workgroup, whole wavefronts, <= the kernels' launch bounds
  uint32_t blocks_per_cu = 2;   // BLOCKS_PER_CU
  bool no_fuse = false;         // NO_FUSE: primary and AO passes as two launches (the reference's shape)
  bool no_gather_order = false; // NO_GATHER_ORDER: plain 8x8 pixel packets in the final gather
  bool no_surfel_sort = false;  // NO_SURFEL_SORT: trace the surfel pool in pool order
  bool no_tile_order = false;   // NO_TILE_ORDER: hand tiles out in screen order, not most expensive first
  bool equal_bands = false;     // EQUAL_BANDS: bands of equal tile count (round 3) instead of equal measured cost
  bool no_lds_boxes = false;    // NO_LDS_BOXES: the packet cull reads the instance boxes from memory
  uint32_t static_rounds = 0xFFFFFFFFu;  // STATIC_ROUNDS: dealt rounds of the hand-out (default: one, kernels.hip with_schedule)
  uint32_t still_refresh_max = 64;  // cap of the launches between two re-measurements of a view that stands still
  uint32_t cost_keep_shift = 1; // a tile's cost estimate moves 1 / 2^k of the way to each new measurement
  bool wide_fused = true;       // NO_WIDE_FUSED: the fused kernel always as two 512-thread workgroups per CU
  bool dilate = true;           // NO_DILATE: a moving view's order from the tiles' own costs only
  bool force_moving = false;    // FORCE_MOVING: treat every view as a moving one
  uint32_t cuts_reuse = 4;      // re-orderings of a moving view that keep one set of band cuts
  uint32_t moving_refresh = 4;  // launches between two re-orderings of a view that moves (order_tiles)
  uint32_t side_prio = 3;       // issue priority floor of the surfel pass on the second stream
  uint32_t stream_refill = 16;  // STREAM_REFILL, STREAM_TOP_ITERS: FrameArgs::stream_refill / stream_top_iters
  uint32_t stream_top_iters = 8;
  uint32_t in_flight_oversub = 0;  // IN_FLIGHT_OVERSUB (percent)
  bool wide_share = false;         // WIDE_SHARE: two frames in flight on half of the slots each as 1024-thread workgroups, one per CU (experiment)
  bool no_stream_lds = false;   // NO_STREAM_LDS: the ray streams read grid, boxes and enter records from memory
  bool packet_gi() const { return gi_path != DUST_GI_PATH_STREAMS; }    // the GI passes a packet of 64 rays at a time (k_final_gather, k_surfel_trace)
  bool packet_only() const { return gi_path == DUST_GI_PATH_PACKETS; }  // ... even where the streams are the default
  static uint32_t num(const char* name, uint32_t dflt) {
    const char* e = diag_env(name);
    return e ? uint32_t(std::strtoul(e, nullptr, 10)) : dflt;
  }
  static bool flag(const char* name) { return diag_env(name) != nullptr; }
  static Tuning from_environment() {
    Tuning t;
    t.debug = num("DEBUG", 0);
#ifndef DUST_MAX_BLOCK
#define DUST_MAX_BLOCK 512u   // the kernels' launch bounds (an experiment build may raise both)
#endif
    t.block = std::min(DUST_MAX_BLOCK, std::max(128u, num("BLOCK", 512) & ~127u));  // <= the kernels' launch bounds (512); an even number of
                                                                                    // waves keeps the LDS areas behind the per-wave lists 16-byte aligned
    t.blocks_per_cu = std::max(1u, num("BLOCKS_PER_CU", 2));
    t.no_fuse = flag("NO_FUSE");
    t.no_gather_order = flag("NO_GATHER_ORDER");
    t.no_surfel_sort = flag("NO_SURFEL_SORT");
    t.no_tile_order = flag("NO_TILE_ORDER");
    t.equal_bands = flag("EQUAL_BANDS");
    t.no_lds_boxes = flag("NO_LDS_BOXES");
    t.no_stream_lds = flag("NO_STREAM_LDS");
    t.stream_refill = std::min(64u, std::max(1u, num("STREAM_REFILL", 16)));
    t.stream_top_iters = std::max(1u, num("STREAM_TOP_ITERS", 8));
    t.static_rounds = num("STATIC_ROUNDS", 0xFFFFFFFFu);
    t.dilate = !flag("NO_DILATE");
    t.wide_fused = !flag("NO_WIDE_FUSED") && !flag("BLOCK");
    t.force_moving = flag("FORCE_MOVING");
    t.in_flight_oversub = std::min(100u, num("IN_FLIGHT_OVERSUB", 0));
    t.wide_share = flag("WIDE_SHARE");
    return t;
  }
};

struct DustHipPipeline {
  DustHipContext* ctx = nullptr;  // retained
  Tuning tune;
  uint32_t frames_in_flight = 1;  // dust_hip_pipeline_set_frames_in_flight
  bool in_collective = false;     // the pipeline has taken part in a collective of a communicator with world > 1 (comm.hip): DUST_RESERVE_AUTO then leaves 32 slots free
  uint32_t width = 0, height = 0;
  DeviceBuffer planes[DUST_PLANE_COUNT];
  void* bound[DUST_PLANE_COUNT] = {};  // caller-owned storage a plane was redirected to (dust_hip_pipeline_bind_plane), or null
  void* plane(int i) const { return bound[i] ? bound[i] : planes[i].p; }
  DeviceBuffer noise0, noise5, counters, stats;
  uint32_t counter_parity[4] = {0, 0, 0, 0};  // per pass kind: which of its two counter sets the next launch uses
  // per pass kind: cycles each tile took in the pass's last launch and the hand-out order made from them (k_tile_order);
  // valid for the tile grid they were recorded on
  struct TileHistory {
    DeviceBuffer cost, order;
    DeviceBuffer spread;     // a moving view: each tile's estimate or its dearest neighbour's (k_cost_dilate)
    DeviceBuffer smooth;     // running mean of the measurements of each tile: what the order is made from (k_cost_blend)
    DeviceBuffer cuts;       // kRegions + 1 tile indices: the cost-balanced bands order[] was made for (FrameArgs::band_cuts)
    uint32_t tiles_x = 0, tiles_y = 0, capacity = 0, age = 0;
    uint32_t refresh = 8;    // launches between two measurements of a view that stands still (kOrderRefresh, doubling up to kOrderRefreshMax)
    uint64_t view = 0;       // view_key() of the launch the costs / the order were taken under
    bool recorded = false;   // cost[] holds the previous launch's measurements
    bool ordered = false;    // order[] is a valid permutation of this tile grid
    bool measured = false;   // cost[] holds a launch's measurements (maybe not the last launch's)
    bool moving = false;     // the previous launch's view differed from the one before it
    uint32_t cuts_age = 0;   // re-orderings since cuts[] was worked out
  } tile_history[4];
  uint64_t view_key = 0;     // this frame's camera + scene revision + sun + row band
  DeviceBuffer exposure;  // Histogram {u32 histogram[256]; f32 avg} (auto_exposure.playout)
  // hash-fed GI state (standard.rs:334-358): spatial hash, surfel pool, per-frame scratch
  DeviceBuffer gi_hash, gi_pool, gi_owner, gi_pixel_surfel, gi_requests, gi_replacement, gi_sun_payload;
  DeviceBuffer gi_apply_alive, gi_apply_dead;  // the deterministic apply's marks (k_surfel_apply_mark)
  DeviceBuffer gi_sort_keys[2], gi_sort_vals[2], gi_sort_scratch;  // radix sort ping-pong (position order of the pool, then the apply order)
  DeviceBuffer gi_fg_hits;  // per pixel: the hit record of its gather ray (k_ray_stream / k_final_gather -> k_final_gather_shade)
  // ray streams (gi.hip): the compacted rays of the two GI passes, the surfel rays' hit records, and per pass two ray counters used in turn
  DeviceBuffer gi_rays_fg, gi_rays_sf, gi_hits_sf, gi_groups_fg, gi_groups_sf, gi_unbinned;
  DeviceBuffer gi_order, gi_order_count;  // final gather: live pixels of each 64x64 tile grouped by ray direction bin
  DeviceBuffer gi_touched, gi_merged;  // multi-GPU exchange buffers (dust_hip_pipeline_gi_exchange)
  // sharded surfel trace (DustHipFrameParams::surfel_world >= 1): the records in slot order (made on first use), and what the pending
  // pass's second half (dust_hip_gi_surfel_exchange_run) needs of its first
  DeviceBuffer gi_stage_req, gi_stage_repl, gi_stage_sun;
  struct { bool pending = false; const uint32_t* perm = nullptr; uint32_t rank = 0, world = 0, slots_per_rank = 0; } sf_shard;
  uint32_t gi_touched_rows = 0;
  uint32_t gi_capacity = 0, gi_pool_size = 0;
  uint32_t noise0_layers = 0, noise5_layers = 0;
  uint32_t accum_count = 0;
  // DUST_PASS_DENOISE: two history sets used in turn {rgb + frame count (16 B); depth, normal, instance as one 16 B record}, the camera
  // of the frame that wrote the current one, and the filter's settings
  DeviceBuffer hist_accum[2], hist_geo[2];
  uint32_t hist_parity = 0;
  bool have_history = false;
  bool hist_traded = true;  // the radiance history is DUST_PLANE_ACCUM's own buffer (not a caller-bound one)
  dust::DevCamera prev_cam{};
  DustHipDenoiseParams denoise{sizeof(DustHipDenoiseParams), 30, 0.01f, 2.0f, 0.8f, 15.0f};
  // HIP-event pairs around the launches of the four pass kinds (primary or fused, AO, final gather, surfel pass), a ring per kind:
  // the last pair is what dust_hip_pipeline_pass_stats reports, the pairs since the last mark what dust_hip_pipeline_kernel_times sums
  static constexpr uint32_t kEvRing = 256;
  std::vector<hipEvent_t> ev_ring[4][2];
  uint32_t ev_head[4] = {0, 0, 0, 0}, ev_mark[4] = {0, 0, 0, 0};
  hipEvent_t ev_begin(int kind) { return ev_ring[kind][0][ev_head[kind]++ % kEvRing]; }
  hipEvent_t ev_end(int kind) { return ev_ring[kind][1][(ev_head[kind] - 1u) % kEvRing]; }
  bool ev_valid[4] = {false, false, false, false};  // primary, ao
  bool stats_valid = false;
  bool fused_last = false;  // the last frame ran primary + AO as one kernel: its time is reported under pass 0
  uint32_t frame_counter = 0;
  bool timed_frame = false;  // this frame's launches are bracketed by event pairs (see dust_hip_render_frame)
  dust::DevStats* host_stats = nullptr;  // pinned, 8 records: where the counting build's statistics land
  // How the workgroup slots are split while a surfel pass runs beside the next frame's primary / AO kernels: the pipeline's first
  // GI frame runs its surfel pass in place and is timed (P: primary / AO kernels, Q: the pass); once those events have completed --
  // looked at without waiting, some frames later -- the pass's share is 100 Q / (Q + 1.05 P) - 3, which is where the measured optima
  // of three workloads lie (castle 1080p 50 %, 4K 20 %, the 4096^3 tree 25 %). Until then: a guess from the ray counts.
  struct { hipEvent_t p0 = nullptr, p1 = nullptr, q0 = nullptr, q1 = nullptr; int state = 0; uint32_t share = 0; uint32_t gi_frames = 0; } side_cal;
};

static const size_t kPlaneBytesPerPixel[DUST_PLANE_COUNT] = {8, 8, 4, 4, 4, 8, 4, 16, 8};
// ------------------------------------------------------------------ device hierarchy build
namespace {

struct N16Builder {
  std::vector<uint8_t> bytes;  // kN16Bytes per node
  uint8_t* node(size_t i) { return bytes.data() + i * dust::kN16Bytes; }
  size_t add() {
    bytes.resize(bytes.size() + dust::kN16Bytes, 0);
    return bytes.size() / dust::kN16Bytes - 1;
  }
  void finish(size_t i, uint32_t child_base) {  // rank prefix per 64-bit word + base of the first child
    uint64_t* mask = reinterpret_cast<uint64_t*>(node(i));
    uint16_t* pre = reinterpret_cast<uint16_t*>(node(i) + 512);
    uint32_t run = 0;
    for (int w = 0; w < 64; ++w) {
      pre[w] = static_cast<uint16_t>(run);
      run += static_cast<uint32_t>(__builtin_popcountll(mask[w]));
    }
    std::memcpy(node(i) + 640, &child_base, 4);
  }
};

// Builds root / l2 / mid arrays from blocks given in Tree::iter_leaf order (depth-first, ascending bits).
DustStatus build_hierarchy(const DustHipBlock* blocks, uint32_t n, uint32_t extent_log2, N16Builder& root,
                           N16Builder& l2, std::vector<dust::DevN4>& mid, std::vector<uint64_t>& dense_mask,
                           float bmin[3], float bmax[3]) {
  const bool deep = extent_log2 == 12;
  const uint32_t extent = 1u << extent_log2;
  root.add();
  uint64_t prev_key = 0;
  int64_t cur_l2 = -1, cur_mid = -1;
  uint32_t cur_l2_cell = 0xFFFFFFFFu, cur_mid_cell = 0xFFFFFFFFu;
  for (int a = 0; a < 3; ++a) { bmin[a] = 1e30f; bmax[a] = -1e30f; }
  auto idx16 = [](uint32_t x, uint32_t y, uint32_t z) { return (x << 8) | (y << 4) | z; };
  for (uint32_t i = 0; i < n; ++i) {
    const DustHipBlock& b = blocks[i];
    if ((b.x & 3) || (b.y & 3) || (b.z & 3) || b.x >= extent || b.y >= extent || b.z >= extent)
      return fail(DUST_ERR_INVALID_ARGUMENT, "block position is not a 4-aligned coordinate inside the tree extent");
    if (b.mask == 0) return fail(DUST_ERR_INVALID_ARGUMENT, "block with empty occupancy mask");
    // depth-first order key: per level, x slowest (node/internal.rs:78-81)
    uint64_t key = 0;
    const uint32_t shifts_deep[3] = {8, 4, 2}, bits_deep[3] = {4, 4, 2};
    const uint32_t shifts_std[2] = {4, 2}, bits_std[2] = {4, 2};
    const uint32_t* sh = deep ? shifts_deep : shifts_std;
    const uint32_t* bt = deep ? bits_deep : bits_std;
    const int nl = deep ? 3 : 2;
    for (int l = 0; l < nl; ++l) {
      const uint32_t m = (1u << bt[l]) - 1;
      key = (key << (3 * bt[l])) | (uint64_t(((b.x >> sh[l]) & m)) << (2 * bt[l])) | (uint64_t((b.y >> sh[l]) & m) << bt[l]) |
            uint64_t((b.z >> sh[l]) & m);
    }
    if (i > 0 && key <= prev_key)
      return fail(DUST_ERR_INVALID_ARGUMENT, "blocks are not in Tree::iter_leaf order (depth-first, ascending child bits)");
    prev_key = key;
    const float p[3] = {float(b.x), float(b.y), float(b.z)};
    for (int a = 0; a < 3; ++a) {
      bmin[a] = std::min(bmin[a], p[a]);
      bmax[a] = std::max(bmax[a], p[a] + 4.0f);
    }
    // descend, creating nodes on first touch (children of a node are contiguous because of the order)
    size_t n16 = 0;           // node holding the 16-cell bit
    N16Builder* holder = &root;
    if (deep) {
      const uint32_t cell = idx16(b.x >> 8, b.y >> 8, b.z >> 8);
      if (cell != cur_l2_cell) {
        cur_l2 = int64_t(l2.add());
        cur_l2_cell = cell;
        dust::vdb::bit_set(reinterpret_cast<uint64_t*>(root.node(0)), cell, true);
        cur_mid_cell = 0xFFFFFFFFu;
      }
      holder = &l2;
      n16 = size_t(cur_l2);
    }
    const uint32_t cell16 = idx16((b.x >> 4) & 15, (b.y >> 4) & 15, (b.z >> 4) & 15);
    const uint32_t mid_cell_key = deep ? uint32_t(cur_l2) * 4096u + cell16 : cell16;
    if (mid_cell_key != cur_mid_cell) {
      dust::vdb::bit_set(reinterpret_cast<uint64_t*>(holder->node(n16)), cell16, true);
      dust::DevN4 nd{0, 0, i, 0};
      mid.push_back(nd);
      cur_mid = int64_t(mid.size()) - 1;
      cur_mid_cell = mid_cell_key;
    }
    const uint32_t bit = (((b.x >> 2) & 3) << 4) | (((b.y >> 2) & 3) << 2) | ((b.z >> 2) & 3);
    if (bit < 32) mid[size_t(cur_mid)].mask_lo |= 1u << bit;
    else mid[size_t(cur_mid)].mask_hi |= 1u << (bit - 32);
    if (dense_mask.size() < mid.size() * 64) dense_mask.resize(mid.size() * 64, 0);
    dense_mask[size_t(cur_mid) * 64 + bit] = b.mask;
  }
  // prefixes and child bases: children were appended in order, so base = running count
  if (deep) {
    root.finish(0, 0);
    uint32_t run = 0;
    for (size_t i = 0; i < l2.bytes.size() / dust::kN16Bytes; ++i) {
      l2.finish(i, run);
      const uint64_t* mask = reinterpret_cast<const uint64_t*>(l2.node(i));
      for (int w = 0; w < 64; ++w) run += uint32_t(__builtin_popcountll(mask[w]));
    }
  } else {
    root.finish(0, 0);
  }
  if (n == 0) { for (int a = 0; a < 3; ++a) { bmin[a] = 0.0f; bmax[a] = 0.0f; } }
  return DUST_OK;
}

// inverse of a 3x4 affine transform, evaluated in double and rounded once
void invert_affine(const float m[12], float out[12]) {
  const double a = m[0], b = m[1], c = m[2], d = m[4], e = m[5], f = m[6], g = m[8], h = m[9], i = m[10];
  const double A = e * i - f * h, B = -(d * i - f * g), C = d * h - e * g;
  const double det = a * A + b * B + c * C;
  const double id = 1.0 / det;
  const double r[9] = {A * id, -(b * i - c * h) * id, (b * f - c * e) * id,
                       B * id, (a * i - c * g) * id, -(a * f - c * d) * id,
                       C * id, -(a * h - b * g) * id, (a * e - b * d) * id};
  const double tx = m[3], ty = m[7], tz = m[11];
  for (int k = 0; k < 3; ++k) {
    out[k * 4 + 0] = float(r[k * 3 + 0]); out[k * 4 + 1] = float(r[k * 3 + 1]); out[k * 4 + 2] = float(r[k * 3 + 2]);
    out[k * 4 + 3] = float(-(r[k * 3 + 0] * tx + r[k * 3 + 1] * ty + r[k * 3 + 2] * tz));
  }
}

}  // namespace

extern "C" {

const char* dust_hip_last_error(void) { return g_last_error.c_str(); }

int dust_hip_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ===================================================================== vdb
DustStatus dust_vdb_tree_create(const uint32_t* fanout_log2, uint32_t n_levels, DustVdbTree** out) {
  if (!fanout_log2 || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&] { *out = new DustVdbTree(fanout_log2, int(n_levels)); return DUST_OK; });
}
void dust_vdb_tree_destroy(DustVdbTree* t) { delete t; }
DustStatus dust_vdb_tree_set(DustVdbTree* t, uint32_t x, uint32_t y, uint32_t z, int32_t value) {
  if (!t) return fail(DUST_ERR_INVALID_ARGUMENT, "null tree");
  const uint32_t e = 1u << t->tree.extent_log2();
  if (x >= e || y >= e || z >= e) return fail(DUST_ERR_INVALID_ARGUMENT, "coordinate outside the tree extent");
  return guarded([&] {
    if (!t->tree.set(x, y, z, value < 0 ? -1 : (value ? 1 : 0)))
      return fail(DUST_ERR_UNSUPPORTED, "clearing voxels through internal nodes is todo!() in the reference");
    return DUST_OK;
  });
}
DustStatus dust_vdb_tree_get(const DustVdbTree* t, uint32_t x, uint32_t y, uint32_t z, int32_t* value) {
  if (!t || !value) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  const uint32_t e = 1u << t->tree.extent_log2();
  if (x >= e || y >= e || z >= e) return fail(DUST_ERR_INVALID_ARGUMENT, "coordinate outside the tree extent");
  *value = t->tree.get(x, y, z);
  return DUST_OK;
}
DustStatus dust_vdb_tree_iter(const DustVdbTree* t, uint32_t* xyz, size_t cap, size_t* count) {
  if (!t || !count) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  size_t n = 0;
  t->tree.for_each_voxel([&](uint32_t x, uint32_t y, uint32_t z) {
    if (xyz && n < cap) { xyz[n * 3] = x; xyz[n * 3 + 1] = y; xyz[n * 3 + 2] = z; }
    ++n;
  });
  *count = n;
  return DUST_OK;
}
DustStatus dust_vdb_tree_iter_leaf(const DustVdbTree* t, uint32_t* xyz, uint64_t* occupancy, uint32_t* material_ptr,
                                   size_t cap, size_t* count) {
  if (!t || !count) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  size_t n = 0;
  t->tree.for_each_leaf([&](dust::vdb::LeafRef& l) {
    if (n < cap) {
      if (xyz) { xyz[n * 3] = l.origin[0]; xyz[n * 3 + 1] = l.origin[1]; xyz[n * 3 + 2] = l.origin[2]; }
      if (occupancy) occupancy[n] = l.occupancy;
      if (material_ptr) material_ptr[n] = *l.material_ptr;
    }
    ++n;
  });
  *count = n;
  return DUST_OK;
}
DustStatus dust_vdb_tree_meta(const DustVdbTree* t, uint32_t* meta_mask, uint32_t* root_level) {
  if (!t) return fail(DUST_ERR_INVALID_ARGUMENT, "null tree");
  if (meta_mask) *meta_mask = t->tree.meta_mask();
  if (root_level) *root_level = uint32_t(t->tree.root_level());
  return DUST_OK;
}
uint32_t dust_vdb_lca_level(const uint32_t a[3], const uint32_t b[3], uint32_t meta_mask, uint32_t root_level) {
  if (!a || !b) { (void)fail(DUST_ERR_INVALID_ARGUMENT, "null coordinates"); return 0xFFFFFFFFu; }
  return dust::vdb::Tree::lca_level(a, b, meta_mask, root_level);
}
DustStatus dust_vdb_accessor_create(const DustVdbTree* t, DustVdbAccessor** out) {
  if (!t || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&] { *out = new DustVdbAccessor(t->tree); return DUST_OK; });
}
void dust_vdb_accessor_destroy(DustVdbAccessor* a) { delete a; }
DustStatus dust_vdb_accessor_get(DustVdbAccessor* a, uint32_t x, uint32_t y, uint32_t z, int32_t* value) {
  if (!a || !value) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  const uint32_t e = 1u << a->acc.tree().extent_log2();
  if (x >= e || y >= e || z >= e) return fail(DUST_ERR_INVALID_ARGUMENT, "coordinate outside the tree extent");
  *value = a->acc.get(x, y, z);
  return DUST_OK;
}
DustStatus dust_vdb_pool_create(size_t item_size, uint32_t chunk_size_log2, DustVdbPool** out) {
  if (!out || item_size < 4 || chunk_size_log2 > 24) return fail(DUST_ERR_INVALID_ARGUMENT, "bad pool parameters");
  return guarded([&] { *out = new DustVdbPool(item_size, chunk_size_log2); return DUST_OK; });
}
void dust_vdb_pool_destroy(DustVdbPool* p) { delete p; }
uint32_t dust_vdb_pool_alloc(DustVdbPool* p) {
  if (!p) { (void)fail(DUST_ERR_INVALID_ARGUMENT, "null pool"); return 0xFFFFFFFFu; }
  try { return p->pool.alloc(); } catch (...) { (void)fail(DUST_ERR_OUT_OF_MEMORY, "pool chunk allocation failed"); return 0xFFFFFFFFu; }
}
void dust_vdb_pool_free(DustVdbPool* p, uint32_t index) {
  if (!p || !p->pool.owns(index)) { (void)fail(DUST_ERR_INVALID_ARGUMENT, "index was not allocated from this pool"); return; }
  p->pool.free(index);
}
size_t dust_vdb_pool_num_chunks(const DustVdbPool* p) { return p ? p->pool.num_chunks() : 0; }
void dust_vdb_bitmask_set(uint64_t* words, size_t index, int32_t value) {
  if (!words) { (void)fail(DUST_ERR_INVALID_ARGUMENT, "null bit mask"); return; }
  dust::vdb::bit_set(words, index, value != 0);
}
size_t dust_vdb_bitmask_iter_set_bits(const uint64_t* words, size_t n_words, uint32_t* out, size_t cap) {
  if (!words && n_words) { (void)fail(DUST_ERR_INVALID_ARGUMENT, "null bit mask"); return 0; }
  size_t n = 0;
  dust::vdb::for_each_set_bit(words, n_words, [&](uint32_t i) {
    if (out && n < cap) out[n] = i;
    ++n;
  });
  return n;
}

// ===================================================================== vox
DustStatus dust_vox_load_frame(const uint8_t* bytes, size_t n_bytes, uint32_t frame, DustVoxScene** out) {
  if (!bytes || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&] {
    std::unique_ptr<DustVoxScene> s(new DustVoxScene);
    s->scene = dust::vox::load(bytes, n_bytes, frame);
    *out = s.release();
    return DUST_OK;
  });
}
DustStatus dust_vox_load(const uint8_t* bytes, size_t n_bytes, DustVoxScene** out) { return dust_vox_load_frame(bytes, n_bytes, 0, out); }
void dust_vox_scene_destroy(DustVoxScene* s) { delete s; }
DustStatus dust_png_load_array(const uint8_t* bytes, size_t n_bytes, DustPngInfo* info, uint8_t** texels) {
  if (!bytes || !info || !texels) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&] {
    const dust::png::ImageArray img = dust::png::load(bytes, n_bytes);
    uint8_t* out = static_cast<uint8_t*>(std::malloc(img.texels.size() ? img.texels.size() : 1));
    if (!out) throw std::bad_alloc();
    std::memcpy(out, img.texels.data(), img.texels.size());
    info->width = img.width; info->height = img.height; info->layers = img.layers;
    info->channels = img.channels; info->bytes_per_channel = img.bytes_per_channel;
    *texels = out;
    return DUST_OK;
  });
}
DustStatus dust_vox_scene_counts(const DustVoxScene* s, uint32_t* n_models, uint32_t* n_instances) {
  if (!s) return fail(DUST_ERR_INVALID_ARGUMENT, "null scene");
  if (n_models) *n_models = uint32_t(s->scene.models.size());
  if (n_instances) *n_instances = uint32_t(s->scene.instances.size());
  return DUST_OK;
}
DustStatus dust_vox_scene_model_info(const DustVoxScene* s, uint32_t model, DustVoxModelInfo* out) {
  if (!s || !out || model >= s->scene.models.size()) return fail(DUST_ERR_INVALID_ARGUMENT, "bad model index");
  const auto& m = s->scene.models[model];
  std::memcpy(out->size, m.size, sizeof(out->size));
  out->n_voxels = uint32_t(m.xyzi.size() / 4);
  out->n_blocks = uint32_t(m.blocks.size());
  out->n_materials = m.materials.size();
  out->used = m.used ? 1u : 0u;
  return DUST_OK;
}
DustStatus dust_vox_scene_model_data(const DustVoxScene* s, uint32_t model, const DustHipBlock** blocks,
                                     const uint8_t** materials) {
  if (!s || model >= s->scene.models.size()) return fail(DUST_ERR_INVALID_ARGUMENT, "bad model index");
  const auto& m = s->scene.models[model];
  if (blocks) *blocks = m.blocks.data();
  if (materials) *materials = m.materials.data();
  return DUST_OK;
}
DustStatus dust_vox_scene_palette(const DustVoxScene* s, const uint8_t** rgba) {
  if (!s || !rgba) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  *rgba = s->scene.palette;
  return DUST_OK;
}
DustStatus dust_vox_scene_instances(const DustVoxScene* s, DustVoxInstance* out, uint32_t cap) {
  if (!s || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  const size_t n = std::min<size_t>(cap, s->scene.instances.size());
  std::memcpy(out, s->scene.instances.data(), n * sizeof(DustVoxInstance));
  return DUST_OK;
}
DustStatus dust_vox_flatten_model(const uint8_t* xyzi, size_t n_voxels, const uint32_t size[3], const uint8_t* palette,
                                  DustHipBlock** blocks, uint32_t* n_blocks, uint8_t** materials,
                                  uint64_t* n_materials) {
  if ((!xyzi && n_voxels) || !size || !palette || !blocks || !n_blocks || !materials || !n_materials)
    return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  if (size[0] > 256 || size[1] > 256 || size[2] > 256) return fail(DUST_ERR_INVALID_ARGUMENT, "model larger than 256^3");
  for (size_t i = 0; i < n_voxels; ++i)
    if (xyzi[i * 4] >= size[0] || xyzi[i * 4 + 1] >= size[1] || xyzi[i * 4 + 2] >= size[2])
      return fail(DUST_ERR_INVALID_ARGUMENT, "voxel outside model size");
  return guarded([&] {
    std::vector<DustHipBlock> b;
    std::vector<uint8_t> m;
    dust::vox::flatten_model(xyzi, n_voxels, size, palette, b, m);
    *blocks = static_cast<DustHipBlock*>(std::malloc(std::max<size_t>(1, b.size()) * sizeof(DustHipBlock)));
    *materials = static_cast<uint8_t*>(std::malloc(std::max<size_t>(1, m.size())));
    if (!*blocks || !*materials) return fail(DUST_ERR_OUT_OF_MEMORY, "host allocation failed");
    std::memcpy(*blocks, b.data(), b.size() * sizeof(DustHipBlock));
    std::memcpy(*materials, m.data(), m.size());
    *n_blocks = uint32_t(b.size());
    *n_materials = m.size();
    return DUST_OK;
  });
}
void dust_vox_free(void* p) { std::free(p); }

// ===================================================================== sky
DustStatus dust_sky_dataset_create(const uint8_t* dataset, size_t n_dataset, const uint8_t* solar, size_t n_solar, DustSkyDataset** out) {
  if (!dataset || !solar || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&] {
    std::unique_ptr<DustSkyDataset> d(new DustSkyDataset);
    if (!dust::sky::load_dataset(dataset, n_dataset, solar, n_solar, d->data))
      return fail(DUST_ERR_INVALID_ARGUMENT, "sky tables must be 14400 bytes (dataset.bin) and 21672 bytes (datasetSolar.bin)");
    *out = d.release();
    return DUST_OK;
  });
}
void dust_sky_dataset_destroy(DustSkyDataset* d) { delete d; }
DustStatus dust_sky_bake(const DustSkyDataset* d, float turbidity, const float albedo[3], const float direction[3], DustHipSky* out) {
  if (!d || !albedo || !direction || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  if (!dust::sky::bake(d->data, turbidity, albedo, direction, out->state))
    return fail(DUST_ERR_INVALID_ARGUMENT, "turbidity must lie in [1, 10] and the sun above the horizon (0 < direction.y <= 1)");
  return DUST_OK;
}

// ===================================================================== device side
DustStatus dust_hip_context_create(const DustHipConfig* cfg, DustHipContext** out) {
  if (!out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  if (cfg) STRUCT_TRY(cfg, "DustHipConfig");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0) return fail(DUST_ERR_NO_DEVICE, "no HIP device visible: the MI355X path has no CPU fallback");
  std::unique_ptr<DustHipContext> c(new DustHipContext);
  int dev = cfg ? cfg->device : -1;
  if (dev < 0) HIP_TRY(hipGetDevice(&dev));
  if (dev >= n) return fail(DUST_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  HIP_TRY(hipSetDevice(dev));
  c->device = dev;
  if (cfg && cfg->stream) c->stream = static_cast<hipStream_t>(cfg->stream);
  else { HIP_TRY(hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)); c->own_stream = true; }
  if (cfg && cfg->lds_root_bytes) c->lds_root_bytes = cfg->lds_root_bytes;
  c->timing = cfg && (cfg->flags & DUST_HIP_CONTEXT_TIMING);
  c->timing_stride = (cfg && (cfg->flags & DUST_HIP_CONTEXT_TIMING_SPARSE)) ? 4u : 1u;
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->max_lds = (prop.sharedMemPerBlock ? prop.sharedMemPerBlock : 64 * 1024) - 256;  // dynamic LDS a launch may ask for: the kernels hold 128 bytes of static LDS (earned priorities)
  if (const char* env = diag_env("LDS_ROOT_BYTES")) c->lds_root_bytes = uint32_t(std::strtoul(env, nullptr, 10));  // (diagnostic: DustHipConfig::lds_root_bytes is the knob)
  // what a 512-thread workgroup needs besides the staged roots: 8 candidate lists, the tile queue, and the static
  // buckets of the profiling / debug builds (kernels.hip lds_bytes(), configure_kernels())
  const size_t reserve = size_t(8) * (dust::kMaxCand * 8 + 8) + 16 + 4096;
  const size_t cap = c->max_lds > reserve ? c->max_lds - reserve : 0;
  if (c->lds_root_bytes > cap) c->lds_root_bytes = uint32_t(cap);
  HIP_TRY(dust::configure_kernels(c->max_lds));
  {  // (a context without the word works as before: commits that find the host a ring ahead wait for the whole stream)
    void* w = nullptr;
    if (!diag_env("NO_START_WORD") && hipHostMalloc(&w, 64, hipHostMallocCoherent | hipHostMallocMapped) == hipSuccess && w) {
      std::memset(w, 0, 64);
      c->started = static_cast<volatile uint32_t*>(w);
    } else (void)hipGetLastError();
  }
  *out = c.release();
  return DUST_OK;
}
void dust_hip_context_destroy(DustHipContext* c) { release(c); }  // (models, scenes and pipelines made from it keep it alive)
DustStatus dust_hip_sync(DustHipContext* c) {
  if (!c) return fail(DUST_ERR_INVALID_ARGUMENT, "null context");
  HIP_TRY(hipSetDevice(c->device));
  HIP_TRY(sync_stream(c));
  return DUST_OK;
}

DustStatus dust_hip_model_create(DustHipContext* ctx, const DustHipBlock* blocks, uint32_t n_blocks,
                                 const uint8_t* materials, uint64_t n_materials, const uint8_t* palette,
                                 uint32_t tree_extent_log2, DustHipModel** out) {
  if (!ctx || !out || (!blocks && n_blocks) || (!materials && n_materials) || !palette)
    return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  if (tree_extent_log2 != 8 && tree_extent_log2 != 12)
    return fail(DUST_ERR_INVALID_ARGUMENT, "tree_extent_log2 must be 8 (hierarchy 4,2,2) or 12 (hierarchy 4,4,2,2)");
  return guarded([&]() -> DustStatus {
    N16Builder root, l2;
    std::vector<dust::DevN4> mid;
    std::vector<uint64_t> dense_mask;
    float bmin[3], bmax[3];
    DustStatus s = build_hierarchy(blocks, n_blocks, tree_extent_log2, root, l2, mid, dense_mask, bmin, bmax);
    if (s != DUST_OK) return s;
    for (uint32_t i = 0; i < n_blocks; ++i) {
      const uint64_t need = uint64_t(blocks[i].material_ptr) + uint64_t(__builtin_popcountll(blocks[i].mask));
      if (need > n_materials) return fail(DUST_ERR_INVALID_ARGUMENT, "block material_ptr runs past the material buffer");
    }
    HIP_TRY(hipSetDevice(ctx->device));
    // (owned through the reference count from here on: an early return releases it, and with it the context reference)
    struct Drop { DustHipModel* m; ~Drop() { release(m); } } owner{new DustHipModel};
    DustHipModel* m = owner.m;
    m->ctx = retain(ctx);
    const hipStream_t up = ctx->stream;
    HIP_TRY(m->root.upload(root.bytes.data(), root.bytes.size(), up));
    m->host_root.assign(root.bytes.begin(), root.bytes.begin() + dust::kN16LdsBytes);
    HIP_TRY(m->l2.upload(l2.bytes.data(), l2.bytes.size(), up));
    if (tree_extent_log2 == 12) {  // the per-cell table the DEEP kernel variants look 16-cells up in: {mid index, child mask} per cell
      const size_t n_l2 = l2.bytes.size() / dust::kN16Bytes;
      std::vector<dust::DevL2Cell> cells(n_l2 * 4096, dust::DevL2Cell{0xFFFFFFFFu, 0u, 0ull});
      for (size_t i = 0; i < n_l2; ++i) {
        const uint64_t* mask = reinterpret_cast<const uint64_t*>(l2.node(i));
        uint32_t base;
        std::memcpy(&base, l2.node(i) + 640, 4);
        uint32_t run = base;  // children of a node are contiguous, in ascending bit order
        for (uint32_t w = 0; w < 64; ++w)
          for (uint64_t bits = mask[w]; bits; bits &= bits - 1) {
            dust::DevL2Cell& c = cells[i * 4096 + w * 64 + uint32_t(__builtin_ctzll(bits))];
            c.mid = run;
            c.child_mask = (uint64_t(mid[run].mask_hi) << 32) | mid[run].mask_lo;
            uint32_t lo[3] = {3, 3, 3}, hi[3] = {0, 0, 0};
            for (uint64_t cm = c.child_mask; cm; cm &= cm - 1) {
              const uint32_t b = uint32_t(__builtin_ctzll(cm)), xyz[3] = {b >> 4, (b >> 2) & 3u, b & 3u};
              for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], xyz[k]); hi[k] = std::max(hi[k], xyz[k]); }
            }
            c.bounds = lo[0] | (lo[1] << 2) | (lo[2] << 4) | (hi[0] << 6) | (hi[1] << 8) | (hi[2] << 10);
            ++run;
          }
      }
      HIP_TRY(m->l2_cells.upload(cells.data(), cells.size() * sizeof(dust::DevL2Cell), up));
    }
    HIP_TRY(m->mid.upload(mid.data(), mid.size() * sizeof(dust::DevN4), up));
    HIP_TRY(m->dense_mask.upload(dense_mask.data(), dense_mask.size() * 8, up));
    HIP_TRY(m->blocks.upload(blocks, size_t(n_blocks) * sizeof(DustHipBlock), up));
    HIP_TRY(m->materials.upload(materials, size_t(n_materials), up));
    m->has_material_255 = n_materials && std::memchr(materials, 255, size_t(n_materials)) != nullptr;
    uint32_t pal[256];
    std::memset(pal, 0, sizeof(pal));
    std::memcpy(pal, palette, 255 * 4);  // loader.rs:214-218: entries 0..254
    HIP_TRY(m->palette.upload(pal, sizeof(pal), up));
    dust::DevModel& d = m->dev;
    d.root = static_cast<const uint8_t*>(m->root.p);
    d.l2 = tree_extent_log2 == 12 ? static_cast<const uint8_t*>(m->l2.p) : nullptr;
    d.l2_cells = tree_extent_log2 == 12 ? static_cast<const dust::DevL2Cell*>(m->l2_cells.p) : nullptr;
    d.mid = static_cast<const dust::DevN4*>(m->mid.p);
    d.dense_mask = static_cast<const uint64_t*>(m->dense_mask.p);
    d.blocks = static_cast<const DustHipBlock*>(m->blocks.p);
    d.materials = static_cast<const uint8_t*>(m->materials.p);
    d.palette = static_cast<const uint32_t*>(m->palette.p);
    std::memcpy(d.bmin, bmin, sizeof(bmin));
    std::memcpy(d.bmax, bmax, sizeof(bmax));
    d.extent = 1u << tree_extent_log2;
    d.n_levels = tree_extent_log2 == 12 ? 3 : 2;
    d.n_blocks = n_blocks;
    d.lds_slot = -1;
    m->n_materials = n_materials;
    *out = retain(m);  // the caller's reference (the guard drops the builder's)
    return DUST_OK;
  });
}
void dust_hip_model_destroy(DustHipModel* m) { release(m); }  // (a scene that instances it keeps it alive)

// ---------------------------------------------------------------- device-side edits (edit.hip)
namespace {
float linear2srgb_host(float c) { return c <= 0.0031308f ? 12.92f * c : 1.055f * std::pow(c, 1.0f / 2.4f) - 0.055f; }  // geometry.rs:99-105

DustStatus ensure_srgb_lut(DustHipContext* ctx) {
  if (ctx->srgb_lut.p) return DUST_OK;
  std::vector<uint16_t> lut(size_t(64) * dust::kSrgbRow, 0);
  for (uint32_t n = 1; n <= 64; ++n) {
    const float denom = float(n) * 255.0f;
    for (uint32_t sum = 0; sum <= n * 255u; ++sum)
      lut[size_t(n - 1) * dust::kSrgbRow + sum] = uint16_t(uint32_t(linear2srgb_host(float(sum) / denom) * 1023.0f));
  }
  HIP_TRY(ctx->srgb_lut.upload(lut.data(), lut.size() * 2, ctx->stream));
  return DUST_OK;
}

dust::EditArgs edit_args(DustHipModel* m, EditState& st) {
  dust::EditArgs e{};
  e.grid = static_cast<uint8_t*>(st.grid.p);
  e.brick_mask = static_cast<uint64_t*>(st.brick_mask.p);
  e.flag_leaf = static_cast<uint32_t*>(st.flag_leaf.p);
  e.count_major = static_cast<uint32_t*>(st.count_major.p);
  e.scan_tmp = static_cast<uint32_t*>(st.scan_tmp.p);
  e.blocks = static_cast<DustHipBlock*>(m->blocks.p);
  e.materials = static_cast<uint8_t*>(m->materials.p);
  e.palette = static_cast<const uint32_t*>(m->palette.p);
  e.srgb_lut = static_cast<const uint16_t*>(m->ctx->srgb_lut.p);
  e.root = static_cast<uint8_t*>(m->root.p);
  e.mid = static_cast<dust::DevN4*>(m->mid.p);
  e.dense_mask = static_cast<uint64_t*>(m->dense_mask.p);
  e.header = static_cast<dust::EditHeader*>(st.header.p);
  return e;
}

// run the rebuild kernels and bring the model record up to date (sizes, bounds, the root the scene stages in LDS)
DustStatus rebuild_and_refresh(DustHipModel* m, EditState& es) {
  hipStream_t st = m->ctx->stream;
  HIP_TRY(dust::launch_edit_rebuild(edit_args(m, es), st));
  dust::EditHeader h{};
  HIP_TRY(hipMemcpyAsync(&h, es.header.p, sizeof(h), hipMemcpyDeviceToHost, st));
  m->host_root.resize(dust::kN16LdsBytes);
  HIP_TRY(hipMemcpyAsync(m->host_root.data(), m->root.p, dust::kN16LdsBytes, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  m->dev.n_blocks = h.n_blocks;
  m->n_materials = h.n_materials;
  std::memcpy(m->dev.bmin, h.bmin, sizeof(h.bmin));
  std::memcpy(m->dev.bmax, h.bmax, sizeof(h.bmax));
  m->generation += 1;
  return DUST_OK;
}

// first edit: move the model into full-capacity buffers and expand its voxels into the dense grid. The model becomes
// editable (m->edit set) only when every step has succeeded: a failure leaves it exactly as it was.
DustStatus make_editable(DustHipModel* m) {
  if (m->edit) return DUST_OK;
  if (m->dev.extent != 256) return fail(DUST_ERR_UNSUPPORTED, "device-side edits cover hierarchy (4,2,2) models (256^3); rebuild larger trees with dust_hip_model_create");
  if (m->has_material_255) return fail(DUST_ERR_UNSUPPORTED, "the model holds material byte 255 (the edit grid stores palette index + 1 in a byte; dust_hip_model_set_voxels takes 0..254)");
  DustStatus s = ensure_srgb_lut(m->ctx);
  if (s != DUST_OK) return s;
  hipStream_t st = m->ctx->stream;
  std::unique_ptr<EditState> e(new EditState);
  const size_t L = dust::kLattice;
  HIP_TRY(e->grid.alloc(L * 64));
  HIP_TRY(hipMemsetAsync(e->grid.p, 0, L * 64, st));
  HIP_TRY(e->brick_mask.alloc(L * 8));
  HIP_TRY(e->flag_leaf.alloc(L * 4));
  HIP_TRY(e->count_major.alloc(L * 4));
  HIP_TRY(e->scan_tmp.alloc(512 * 4));
  HIP_TRY(e->header.alloc(sizeof(dust::EditHeader)));
  DeviceBuffer blocks, materials, mid, dense_mask;
  HIP_TRY(blocks.alloc(L * sizeof(DustHipBlock)));
  HIP_TRY(materials.alloc(L * 64));
  HIP_TRY(mid.alloc(4096 * sizeof(dust::DevN4)));
  HIP_TRY(dense_mask.alloc(size_t(4096) * 64 * 8));
  dust::EditArgs a = edit_args(m, *e);  // (expand only writes the grid)
  HIP_TRY(dust::launch_edit_expand(a, static_cast<const DustHipBlock*>(m->blocks.p), static_cast<const uint8_t*>(m->materials.p), m->dev.n_blocks, st));
  HIP_TRY(sync_stream(m->ctx));  // every launch that reads the old arrays is done (both streams of the context)
  auto swap_all = [&] {
    std::swap(m->blocks.p, blocks.p); std::swap(m->blocks.bytes, blocks.bytes);
    std::swap(m->materials.p, materials.p); std::swap(m->materials.bytes, materials.bytes);
    std::swap(m->mid.p, mid.p); std::swap(m->mid.bytes, mid.bytes);
    std::swap(m->dense_mask.p, dense_mask.p); std::swap(m->dense_mask.bytes, dense_mask.bytes);
    m->dev.mid = static_cast<const dust::DevN4*>(m->mid.p);
    m->dev.dense_mask = static_cast<const uint64_t*>(m->dense_mask.p);
    m->dev.blocks = static_cast<const DustHipBlock*>(m->blocks.p);
    m->dev.materials = static_cast<const uint8_t*>(m->materials.p);
  };
  swap_all();
  // the same voxels, now in the full-capacity arrays (bumps the generation: scenes holding the old addresses commit again)
  s = rebuild_and_refresh(m, *e);
  if (s != DUST_OK) { swap_all(); return s; }  // back to the tightly sized originals, untouched
  m->edit = std::move(e);
  return DUST_OK;
}

DustStatus upload_batch(DustHipModel* m, const uint32_t* xyz, const int32_t* values, uint32_t n, bool with_values) {
  EditState& e = *m->edit;
  if (n > e.batch_capacity) {
    const uint32_t cap = std::max(n, 1024u);
    HIP_TRY(sync_stream(m->ctx));
    HIP_TRY(e.xyz.alloc(size_t(cap) * 12));
    HIP_TRY(e.values.alloc(size_t(cap) * 4));
    e.batch_capacity = cap;
  }
  HIP_TRY(hipMemcpyAsync(e.xyz.p, xyz, size_t(n) * 12, hipMemcpyHostToDevice, m->ctx->stream));
  if (with_values) HIP_TRY(hipMemcpyAsync(e.values.p, values, size_t(n) * 4, hipMemcpyHostToDevice, m->ctx->stream));
  return DUST_OK;
}
}  // namespace

DustStatus dust_hip_model_set_voxels(DustHipModel* m, const uint32_t* xyz, const int32_t* values, uint32_t n) {
  if (!m || (n && (!xyz || !values))) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  for (uint32_t i = 0; i < n; ++i) {
    if (xyz[i * 3] >= m->dev.extent || xyz[i * 3 + 1] >= m->dev.extent || xyz[i * 3 + 2] >= m->dev.extent)
      return fail(DUST_ERR_INVALID_ARGUMENT, "voxel coordinate outside the tree extent");
    if (values[i] > 254) return fail(DUST_ERR_INVALID_ARGUMENT, "palette index must be 0..254 (or negative to clear the voxel)");
  }
  return guarded([&]() -> DustStatus {
    HIP_TRY(hipSetDevice(m->ctx->device));
    HIP_TRY(join_side(m->ctx));  // (a surfel pass on the second stream still traces the model as it is)
    DustStatus s = make_editable(m);
    if (s != DUST_OK || n == 0) return s;
    // a voxel named more than once takes its LAST value (what a sequence of set calls would leave): keep the last entry
    std::vector<uint32_t> ux;
    std::vector<int32_t> uv;
    std::unordered_set<uint32_t> seen;
    ux.reserve(size_t(n) * 3); uv.reserve(n);
    for (uint32_t k = n; k-- > 0;) {
      const uint32_t key = (xyz[k * 3] << 16) | (xyz[k * 3 + 1] << 8) | xyz[k * 3 + 2];
      if (!seen.insert(key).second) continue;
      ux.push_back(xyz[k * 3]); ux.push_back(xyz[k * 3 + 1]); ux.push_back(xyz[k * 3 + 2]);
      uv.push_back(values[k]);
    }
    const uint32_t un = uint32_t(uv.size());
    s = upload_batch(m, ux.data(), uv.data(), un, true);
    if (s != DUST_OK) return s;
    dust::EditArgs a = edit_args(m, *m->edit);
    a.xyz = static_cast<const uint32_t*>(m->edit->xyz.p);
    a.values = static_cast<const int32_t*>(m->edit->values.p);
    a.n_edits = un;
    HIP_TRY(dust::launch_edit_apply(a, false, m->ctx->stream));
    return rebuild_and_refresh(m, *m->edit);  // synchronises: the host vectors above stay alive until the copies are done
  });
}

DustStatus dust_hip_model_get_voxels(DustHipModel* m, const uint32_t* xyz, int32_t* values, uint32_t n) {
  if (!m || (n && (!xyz || !values))) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  for (uint32_t i = 0; i < n; ++i)
    if (xyz[i * 3] >= m->dev.extent || xyz[i * 3 + 1] >= m->dev.extent || xyz[i * 3 + 2] >= m->dev.extent)
      return fail(DUST_ERR_INVALID_ARGUMENT, "voxel coordinate outside the tree extent");
  return guarded([&]() -> DustStatus {
    HIP_TRY(hipSetDevice(m->ctx->device));
    HIP_TRY(join_side(m->ctx));
    DustStatus s = make_editable(m);
    if (s != DUST_OK || n == 0) return s;
    s = upload_batch(m, xyz, nullptr, n, false);
    if (s != DUST_OK) return s;
    dust::EditArgs a = edit_args(m, *m->edit);
    a.xyz = static_cast<const uint32_t*>(m->edit->xyz.p);
    a.values_out = static_cast<int32_t*>(m->edit->values.p);
    a.n_edits = n;
    HIP_TRY(dust::launch_edit_apply(a, true, m->ctx->stream));
    HIP_TRY(hipMemcpyAsync(values, m->edit->values.p, size_t(n) * 4, hipMemcpyDeviceToHost, m->ctx->stream));
    HIP_TRY(sync_stream(m->ctx));
    return DUST_OK;
  });
}

DustStatus dust_hip_model_info(const DustHipModel* m, uint32_t* n_blocks, uint64_t* n_materials) {
  if (!m) return fail(DUST_ERR_INVALID_ARGUMENT, "null model");
  if (n_blocks) *n_blocks = m->dev.n_blocks;
  if (n_materials) *n_materials = m->n_materials;
  return DUST_OK;
}

DustStatus dust_hip_model_read(const DustHipModel* m, DustHipBlock* blocks, uint32_t block_capacity, uint8_t* materials, uint64_t material_capacity) {
  if (!m) return fail(DUST_ERR_INVALID_ARGUMENT, "null model");
  if ((blocks && block_capacity < m->dev.n_blocks) || (materials && material_capacity < m->n_materials))
    return fail(DUST_ERR_INVALID_ARGUMENT, "destination too small (see dust_hip_model_info)");
  HIP_TRY(hipSetDevice(m->ctx->device));
  const hipStream_t st = m->ctx->stream;
  if (blocks && m->dev.n_blocks) HIP_TRY(hipMemcpyAsync(blocks, m->blocks.p, size_t(m->dev.n_blocks) * sizeof(DustHipBlock), hipMemcpyDeviceToHost, st));
  if (materials && m->n_materials) HIP_TRY(hipMemcpyAsync(materials, m->materials.p, size_t(m->n_materials), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  return DUST_OK;
}

DustStatus dust_hip_scene_create(DustHipContext* ctx, DustHipScene** out) {
  if (!ctx || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  return guarded([&] {
    DustHipScene* s = new DustHipScene;
    s->ctx = retain(ctx);
    *out = s;
    return DUST_OK;
  });
}
void dust_hip_scene_destroy(DustHipScene* s) { release(s); }

static DustStatus check_affine(const float m[12]) {
  for (int i = 0; i < 12; ++i)
    if (!std::isfinite(m[i])) return fail(DUST_ERR_INVALID_ARGUMENT, "non-finite instance transform");
  const double det = double(m[0]) * (double(m[5]) * m[10] - double(m[6]) * m[9]) -
                     double(m[1]) * (double(m[4]) * m[10] - double(m[6]) * m[8]) +
                     double(m[2]) * (double(m[4]) * m[9] - double(m[5]) * m[8]);
  if (!(std::fabs(det) > 1e-20)) return fail(DUST_ERR_INVALID_ARGUMENT, "singular instance transform");
  return DUST_OK;
}

DustStatus dust_hip_scene_add_instance(DustHipScene* s, const DustHipModel* model, const float o2w[12],
                                       const float prev[16], uint32_t* instance_id) {
  if (!s || !model || !o2w) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  if (model->ctx != s->ctx) return fail(DUST_ERR_INVALID_ARGUMENT, "model belongs to another context");
  if (s->instances.size() >= 65535) return fail(DUST_ERR_INVALID_ARGUMENT, "too many instances (voxel_id holds 16 bits)");
  DustStatus st = check_affine(o2w);
  if (st != DUST_OK) return st;
  return guarded([&] {
    HostInstance hi;
    hi.model = model;
    std::memcpy(hi.o2w, o2w, sizeof(hi.o2w));
    if (prev) std::memcpy(hi.prev, prev, sizeof(hi.prev));
    else {  // first frame: previous transform == current (standard.rs:856-878), column-major mat4
      for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 4; ++r) hi.prev[c * 4 + r] = r < 3 ? o2w[r * 4 + c] : (c == 3 ? 1.0f : 0.0f);
    }
    s->instances.reserve(s->instances.size() + 1);
    s->dirty.reserve(s->dirty.size() + 1);
    if (instance_id) *instance_id = uint32_t(s->instances.size());
    s->instances.push_back(hi);
    s->dirty.push_back(1);
    retain(const_cast<DustHipModel*>(model));  // the scene keeps what it instances alive
    s->structure_dirty = true;
    s->committed = false;
    return DUST_OK;
  });
}
DustStatus dust_hip_scene_set_transform(DustHipScene* s, uint32_t id, const float o2w[12], const float prev[16]) {
  if (!s || !o2w || id >= s->instances.size()) return fail(DUST_ERR_INVALID_ARGUMENT, "bad instance id");
  DustStatus st = check_affine(o2w);
  if (st != DUST_OK) return st;
  HostInstance& hi = s->instances[id];
  std::memcpy(hi.o2w, o2w, sizeof(hi.o2w));
  if (prev) std::memcpy(hi.prev, prev, sizeof(hi.prev));
  s->dirty[id] = 1;
  s->committed = false;
  return DUST_OK;
}

namespace {
// conservative world box of instance i: the eight corners of its model's tight bounds, each padded by 1e-4 of its size
void world_box(const HostInstance& hi, float wmin[3], float wmax[3]) {
  const dust::DevModel& m = hi.model->dev;
  for (int a = 0; a < 3; ++a) { wmin[a] = 1e30f; wmax[a] = -1e30f; }
  for (int c = 0; c < 8; ++c) {
    const double p[3] = {(c & 1) ? m.bmax[0] : m.bmin[0], (c & 2) ? m.bmax[1] : m.bmin[1], (c & 4) ? m.bmax[2] : m.bmin[2]};
    for (int a = 0; a < 3; ++a) {
      const float* r = hi.o2w + a * 4;
      const double w = double(r[0]) * p[0] + double(r[1]) * p[1] + double(r[2]) * p[2] + double(r[3]);
      const double pad = 1e-4 * (std::fabs(w) + 1.0);
      wmin[a] = std::min(wmin[a], float(w - pad));
      wmax[a] = std::max(wmax[a], float(w + pad));
    }
  }
}
// the device records of instance i, re-derived in the host master image (instance, box, visit)
void derive_instance(DustHipScene* s, size_t i) {
  const HostInstance& hi = s->instances[i];
  uint8_t* img = s->master.data();
  dust::DevInstance& d = reinterpret_cast<dust::DevInstance*>(img + s->layout.instances)[i];
  std::memcpy(d.o2w, hi.o2w, sizeof(d.o2w));
  std::memcpy(d.prev, hi.prev, sizeof(d.prev));
  invert_affine(hi.o2w, d.w2o);
  d.model = s->instance_slot[i];
  d.pad = 0;
  world_box(hi, d.wmin, d.wmax);
  // the world box again, packed 32 bytes apiece: what the packet culling streams through (coalesced) and the candidate
  // loop reads with one scalar load; and the flattened visit record (box, world -> object, model)
  dust::DevBox& bx = reinterpret_cast<dust::DevBox*>(img + s->layout.boxes)[i];
  dust::DevVisit& v = reinterpret_cast<dust::DevVisit*>(img + s->layout.visits)[i];
  for (int a = 0; a < 3; ++a) { bx.lo[a] = v.lo[a] = d.wmin[a]; bx.hi[a] = v.hi[a] = d.wmax[a]; }
  bx.pad0 = bx.pad1 = v.pad0 = v.pad1 = 0.0f;
  std::memcpy(v.w2o, d.w2o, sizeof(v.w2o));
  v.m = reinterpret_cast<const dust::DevModel*>(img + s->layout.models)[d.model];
  // and what the ray streams' instance set-up reads, in 80 bytes (the model's bounds are multiples of 4 up to 4096: exact in 16 bits)
  dust::DevEnter& e = reinterpret_cast<dust::DevEnter*>(img + s->layout.enters)[i];
  std::memcpy(e.w2o, d.w2o, sizeof(e.w2o));
  for (int a = 0; a < 3; ++a) { e.bmin[a] = uint16_t(v.m.bmin[a]); e.bmax[a] = uint16_t(v.m.bmax[a]); }
  e.model = uint16_t(d.model);
  e.lds_slot = v.m.lds_slot >= 0 && v.m.lds_slot < 255 ? uint8_t(v.m.lds_slot) : uint8_t(255);
  e.extent_log2 = v.m.extent > 256u ? 12 : 8;
  e.root = v.m.root;
  e.dense_mask = v.m.dense_mask;
}

// The top-level grid over the instances' world boxes (DevGrid; tlas.rs:37-65 rebuilds the TLAS every frame the same way).
// About `density` cells per instance (DUST_HIP_GRID_DENSITY, default 12; at most 256 per axis, 2^18 in all), cubes as nearly
// as the scene's proportions allow. An instance is listed in every cell its box, grown by kGridMargin of the scene's size,
// overlaps: the margin is what lets the per-ray walk (gi.hip, top_next) trust its single-precision cell steps.
// boxes: n x {lo[3], hi[3]}. ranges: per instance the block of cells it is listed in, {lo, hi} as x | y << 9 | z << 18.
constexpr double kGridMargin = 2e-5;
void build_grid(DustHipScene* s, const std::vector<float>& boxes, std::vector<uint32_t>& ranges) {
  const size_t n = boxes.size() / 6;
  dust::DevGrid& g = s->grid;
  double ext[3], big = 0.0;
  for (int a = 0; a < 3; ++a) big = std::max(big, double(s->world_max[a]) - double(s->world_min[a]));
  const double margin = kGridMargin * big + 0.01;
  for (int a = 0; a < 3; ++a) {
    g.lo[a] = float(double(s->world_min[a]) - 2.0 * margin);
    ext[a] = std::max(double(s->world_max[a]) + 2.0 * margin - double(g.lo[a]), 1e-3 * big + 1.0);
  }
  const char* density_env = diag_env("GRID_DENSITY");  // (per commit: ~100 ns, and tests vary it)
  const double density0 = density_env ? std::max(0.001, std::atof(density_env)) : 12.0;
  ranges.resize(n * 2);
  std::vector<uint32_t> count;
  uint32_t last_dim[3] = {0, 0, 0}, halvings = 0;
  bool force_one = false;
  for (double density = density0;; density *= 0.5) {
    const double target = std::min(262144.0, std::max(1.0, density * double(std::max<size_t>(n, 1))));
    const double edge = std::cbrt(ext[0] * ext[1] * ext[2] / target);
    for (int a = 0; a < 3; ++a) {
      g.dim[a] = force_one ? 1u : uint32_t(std::min(256.0, std::max(1.0, std::floor(ext[a] / edge + 0.5))));
      g.cell[a] = float(ext[a] / double(g.dim[a]));
      g.inv_cell[a] = float(double(g.dim[a]) / ext[a]);
      g.hi[a] = float(double(g.lo[a]) + ext[a]);
    }
    const size_t n_cells = size_t(g.dim[0]) * g.dim[1] * g.dim[2];
    count.assign(n_cells, 0u);
    auto cell_of = [&](double w, int a) {
      const double c = std::floor((w - double(g.lo[a])) / ext[a] * double(g.dim[a]));
      return uint32_t(std::min(double(g.dim[a] - 1), std::max(0.0, c)));
    };
    size_t total = 0;
    uint32_t most = 0;
    for (size_t i = 0; i < n; ++i) {
      uint32_t lo[3], hi[3];
      for (int a = 0; a < 3; ++a) { lo[a] = cell_of(double(boxes[i * 6 + a]) - margin, a); hi[a] = cell_of(double(boxes[i * 6 + 3 + a]) + margin, a); }
      ranges[i * 2] = lo[0] | (lo[1] << 9) | (lo[2] << 18);
      ranges[i * 2 + 1] = hi[0] | (hi[1] << 9) | (hi[2] << 18);
      for (uint32_t z = lo[2]; z <= hi[2]; ++z)
        for (uint32_t y = lo[1]; y <= hi[1]; ++y)
          for (uint32_t x = lo[0]; x <= hi[0]; ++x) most = std::max(most, ++count[(size_t(z) * g.dim[1] + y) * g.dim[0] + x]);
      total += size_t(hi[0] - lo[0] + 1) * (hi[1] - lo[1] + 1) * (hi[2] - lo[2] + 1);
    }
    // (coarser until the item array fits the packed cell word's 20 index bits: never, for scenes of any sane shape. More than 4095 boxes over
    //  ONE cell cannot be listed at any resolution that helps: the grid is then marked unusable and the single-ray paths are not taken)
    s->grid_valid = most <= dust::kGridMaxCellItems;
    if (total < (size_t(1) << dust::kGridItemBits) || n_cells == 1) break;
    // (a fuse: an elongated scene whose rounded dims stop shrinking -- density x n <= 1 clamps the target to one cell, which ext / cbrt(V)
    //  never reaches for aspects above ~2 -- or 40 halvings: ONE cell then; should even that list 2^20 items or more -- 2^20 boxes do not
    //  exist, 65 535 instances at most --, the grid is marked unusable like a cell of more than 4095)
    const bool stuck = g.dim[0] == last_dim[0] && g.dim[1] == last_dim[1] && g.dim[2] == last_dim[2];
    for (int a = 0; a < 3; ++a) last_dim[a] = g.dim[a];
    if (force_one) { s->grid_valid = false; break; }
    if (stuck || ++halvings >= 40) force_one = true;
  }
  const size_t n_cells = count.size();
  s->grid_cells.assign(n_cells, 0u);
  std::vector<uint32_t> at(n_cells);
  uint32_t run = 0;
  for (size_t c = 0; c < n_cells; ++c) {
    at[c] = run;
    s->grid_cells[c] = run | (std::min(count[c], dust::kGridMaxCellItems) << dust::kGridItemBits);
    run += count[c];
  }
  g.n_items = run;
  s->grid_items.assign(run, 0);
  for (size_t i = 0; i < n; ++i) {  // ascending instance order inside every cell
    const uint32_t rl = ranges[i * 2], rh = ranges[i * 2 + 1];
    for (uint32_t z = rl >> 18; z <= (rh >> 18); ++z)
      for (uint32_t y = (rl >> 9) & 255u; y <= ((rh >> 9) & 255u); ++y)
        for (uint32_t x = rl & 255u; x <= (rh & 255u); ++x) s->grid_items[at[(size_t(z) * g.dim[1] + y) * g.dim[0] + x]++] = uint16_t(i);
  }
}
// the instances in the order of a Hilbert curve through their boxes' centres (10 bits per axis over the scene's bounds; Skilling's
// transform): the packet cull's groups are runs of 64 consecutive slots, and consecutive cells of this curve are always neighbours
// (along a Z-order a run that crosses a block boundary jumps across the scene, and the group's box with it)
void order_slots(DustHipScene* s) {
  const size_t n = s->world_boxes.size() / 6;
  std::vector<std::pair<uint32_t, uint32_t>> keyed(n);
  auto spread = [](uint32_t v) { v &= 1023u; v = (v | (v << 16)) & 0x030000FFu; v = (v | (v << 8)) & 0x0300F00Fu; v = (v | (v << 4)) & 0x030C30C3u; return (v | (v << 2)) & 0x09249249u; };
  for (size_t i = 0; i < n; ++i) {
    uint32_t c[3];
    for (int a = 0; a < 3; ++a) {
      const double span = std::max(1e-6, double(s->world_max[a]) - double(s->world_min[a]));
      const double mid = 0.5 * (double(s->world_boxes[i * 6 + a]) + double(s->world_boxes[i * 6 + 3 + a]));
      c[a] = uint32_t(std::min(1023.0, std::max(0.0, (mid - double(s->world_min[a])) / span * 1024.0)));
    }
    uint32_t X[3] = {c[0], c[1], c[2]};
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) {
      const uint32_t P = Q - 1u;
      for (int a = 0; a < 3; ++a) {
        if (X[a] & Q) X[0] ^= P;
        else { const uint32_t t = (X[0] ^ X[a]) & P; X[0] ^= t; X[a] ^= t; }
      }
    }
    X[1] ^= X[0]; X[2] ^= X[1];
    uint32_t t = 0;
    for (uint32_t Q = 512u; Q > 1u; Q >>= 1) if (X[2] & Q) t ^= Q - 1u;
    for (uint32_t& x : X) x ^= t;
    keyed[i] = {(spread(X[0]) << 2) | (spread(X[1]) << 1) | spread(X[2]), uint32_t(i)};
  }
  std::sort(keyed.begin(), keyed.end());
  s->slot_order.resize(n);
  for (size_t k = 0; k < n; ++k) s->slot_order[k] = keyed[k].second;
}
}  // namespace

DustStatus dust_hip_scene_commit(DustHipScene* s) {
  if (!s) return fail(DUST_ERR_INVALID_ARGUMENT, "null scene");
  return guarded([&]() -> DustStatus {
    HIP_TRY(hipSetDevice(s->ctx->device));
    const size_t n = s->instances.size();
    // a model edited since the last commit changes its record (bounds, sizes, maybe addresses): everything is derived again
    for (size_t i = 0; i < s->models.size() && !s->structure_dirty; ++i)
      if (s->models[i]->generation != s->model_generation[i]) s->structure_dirty = true;
    bool full = s->structure_dirty;
    if (full) {
      s->committed = false;  // (until the new image is up: what follows replaces the layout the current one was made with)
      s->models.clear();
      s->instance_slot.resize(n);
      for (size_t i = 0; i < n; ++i) {
        const DustHipModel* m = s->instances[i].model;
        auto it = std::find(s->models.begin(), s->models.end(), m);
        s->instance_slot[i] = uint32_t(it - s->models.begin());
        if (it == s->models.end()) s->models.push_back(m);
      }
      // roots of the first models go to LDS, as many as the budget holds
      s->n_lds_models = std::min<uint32_t>(uint32_t(s->models.size()), s->ctx->lds_root_bytes / dust::kN16LdsBytes);
    }
    // the instances' world boxes (those that moved, or all), the scene's bounds, and the top-level grid over them: the grid's
    // size decides the image's layout
    s->world_boxes.resize(n * 6);
    for (size_t i = 0; i < n; ++i)
      if (full || s->dirty[i]) world_box(s->instances[i], &s->world_boxes[i * 6], &s->world_boxes[i * 6 + 3]);
    for (int a = 0; a < 3; ++a) { s->world_min[a] = 1e30f; s->world_max[a] = -1e30f; }
    for (size_t i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) { s->world_min[a] = std::min(s->world_min[a], s->world_boxes[i * 6 + a]); s->world_max[a] = std::max(s->world_max[a], s->world_boxes[i * 6 + 3 + a]); }
    if (n == 0) for (int a = 0; a < 3; ++a) s->world_min[a] = s->world_max[a] = 0.0f;
    std::vector<uint32_t> ranges;
    build_grid(s, s->world_boxes, ranges);
    const size_t n_cells = s->grid_cells.size(), n_items = s->grid_items.size();
    if (!full && (n_cells > s->layout.cap_cells || n_items > s->layout.cap_items)) full = true;  // the grid outgrew its sections
    if (full) {
      s->committed = false;
      s->structure_dirty = true;
      s->layout = SceneLayout::make(n, s->models.size(), s->n_lds_models, n_cells, n_items);
      if (s->layout.total > s->image_capacity || s->current < 0) {
        // grow (rare: instances were added). Launches that read the old images are done before they go; the new ones are
        // allocated into locals first, so a failed allocation leaves the scene as it was (and the next commit tries again).
        HIP_TRY(sync_stream(s->ctx));
        const size_t cap = s->layout.total + s->layout.total / 2 + 4096;
        DustHipScene::Slot fresh[DustHipScene::kImages];
        hipError_t ge = hipSuccess;
        for (DustHipScene::Slot& sl : fresh) {
          if (ge == hipSuccess) ge = sl.dev.alloc(cap);
          if (ge == hipSuccess) ge = hipHostMalloc(&sl.host, cap, hipHostMallocDefault);
        }
        if (ge != hipSuccess) {
          for (DustHipScene::Slot& sl : fresh) { if (sl.host) (void)hipHostFree(sl.host); sl.dev.release(); }
          s->structure_dirty = true;  // (the layout above is not the images': the next commit starts over)
          s->committed = false;
          return hip_fail(ge, "scene image allocation");
        }
        s->committed = false;  // no current image until the upload below has succeeded (a frame must not index slot -1)
        s->structure_dirty = true;
        s->free_images();
        for (int i = 0; i < DustHipScene::kImages; ++i) {
          s->slots[i].dev.p = fresh[i].dev.p; s->slots[i].dev.bytes = fresh[i].dev.bytes;
          fresh[i].dev.p = nullptr; fresh[i].dev.bytes = 0;  // (ownership moved: the local's destructor frees nothing)
          s->slots[i].host = fresh[i].host; s->slots[i].epoch = 0;
        }
        s->image_capacity = cap;
        s->next_slot = 0;
      }
      s->master.assign(s->layout.total, 0);
      uint8_t* img = s->master.data();
      dust::DevModel* dm = reinterpret_cast<dust::DevModel*>(img + s->layout.models);
      for (size_t i = 0; i < s->models.size(); ++i) {
        dm[i] = s->models[i]->dev;
        dm[i].lds_slot = i < s->n_lds_models ? int32_t(i) : -1;
      }
      for (uint32_t i = 0; i < s->n_lds_models; ++i)
        std::memcpy(img + s->layout.root_table + size_t(i) * dust::kN16LdsBytes, s->models[i]->host_root.data(), dust::kN16LdsBytes);
      s->model_generation.clear();
      for (const DustHipModel* m : s->models) s->model_generation.push_back(m->generation);
    }
    uint8_t* img = s->master.data();
    for (size_t i = 0; i < n; ++i)
      if (full || s->dirty[i]) { derive_instance(s, i); s->dirty[i] = 0; }
    // (the record behind the last instance stays zero: the cull reads boxes 64 at a time)
    // every instance's block of grid cells into the spare words of its box record (the grid is new: so are the blocks), the grid behind the records
    {
      dust::DevBox* bx = reinterpret_cast<dust::DevBox*>(img + s->layout.boxes);
      dust::DevVisit* vs = reinterpret_cast<dust::DevVisit*>(img + s->layout.visits);
      for (size_t i = 0; i < n; ++i) {
        std::memcpy(&bx[i].pad0, &ranges[i * 2], 4); std::memcpy(&bx[i].pad1, &ranges[i * 2 + 1], 4);
        vs[i].pad0 = bx[i].pad0; vs[i].pad1 = bx[i].pad1;
      }
      std::memcpy(img + s->layout.grid_cells, s->grid_cells.data(), s->grid_cells.size() * sizeof(uint32_t));
      // the packet cull's 64-wide hierarchy (kernels: cull_instances): slots along a Morton curve through the boxes' centres -- ordered
      // by structural commits, refitted by every commit --, a box per 64 consecutive slots
      s->n_groups = n > dust::kFlatCullMax && !diag_env("FLAT_CULL") ? uint32_t((n + 63) / 64) : 0u;  // (DUST_HIP_FLAT_CULL: every box for every packet, for A/B runs)
      if (s->n_groups) {
        if (full || s->slot_order.size() != n) {
          order_slots(s);
        }
        dust::DevBox* sb = reinterpret_cast<dust::DevBox*>(img + s->layout.sboxes);
        dust::DevBox* gb = reinterpret_cast<dust::DevBox*>(img + s->layout.gboxes);
        for (uint32_t g = 0; g < s->n_groups; ++g) {
          dust::DevBox u;
          for (int a = 0; a < 3; ++a) { u.lo[a] = 1e30f; u.hi[a] = -1e30f; }
          u.pad0 = u.pad1 = 0.0f;
          for (size_t k = size_t(g) * 64; k < std::min(n, size_t(g + 1) * 64); ++k) {
            const uint32_t i = s->slot_order[k];
            sb[k] = bx[i];
            std::memcpy(&sb[k].pad0, &i, 4);
            for (int a = 0; a < 3; ++a) { u.lo[a] = std::min(u.lo[a], bx[i].lo[a]); u.hi[a] = std::max(u.hi[a], bx[i].hi[a]); }
          }
          gb[g] = u;
        }
      } else {
        s->slot_order.clear();
      }
      if (n_items) std::memcpy(img + s->layout.grid_items, s->grid_items.data(), n_items * sizeof(uint16_t));
    }
    // upload: the whole image into the next slot of the ring, on the copy stream, and wait for it here (a ~100 KB copy: ~20 us of host
    // time, none of the launch stream's); frames in flight keep reading the slot they were enqueued with
    if (!s->ctx->copy) HIP_TRY(hipStreamCreateWithFlags(&s->ctx->copy, hipStreamNonBlocking));
    const int slot = int(s->next_slot++ % DustHipScene::kImages);
    DustHipScene::Slot& sl = s->slots[slot];
    if (sl.epoch == s->ctx->sync_epoch) {  // nobody has waited since a frame last read this slot: the host is a ring ahead
      // The frame AFTER that one has started => that one is done (one stream, launches in order; only the plain case: nothing outstanding on the
      // side stream or on communicators' streams). The GPU keeps the rest of the ring to work on meanwhile:
      // waiting for the whole stream instead left it idle for the ~50 us the host needs to enqueue again, every 8th frame (2 % of a moving view).
      DustHipContext* c = s->ctx;
      bool waited = false;
      const uint32_t need = sl.last_seq + (c->side ? 2u : 1u);  // (a surfel pass on the side stream is joined in the course of the NEXT frame: one more)
      if (c->started && sl.last_seq != 0 && !c->side_busy && c->extra_streams.empty() && int32_t(c->frame_seq - need) >= 0) {
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t spin = 0;; ++spin) {
          if (int32_t(*c->started - need) >= 0) { waited = true; break; }
          if ((spin & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;  // (something else holds the queue: wait for all of it)
        }
      }
      if (!waited) HIP_TRY(sync_stream(c));
    }
    const size_t used = s->layout.grid_items + n_items * sizeof(uint16_t);  // (the sections' spare room is not sent)
    std::memcpy(sl.host, img, used);
    HIP_TRY(hipMemcpyAsync(sl.dev.p, sl.host, used, hipMemcpyHostToDevice, s->ctx->copy));
    HIP_TRY(hipStreamSynchronize(s->ctx->copy));
    s->current = slot;
    ++s->revision;
    s->structure_dirty = false;
    s->committed = true;
    return DUST_OK;
  });
}

DustStatus dust_hip_top_level_build(const float* boxes, uint32_t n, DustTopLevelInfo* info, uint32_t* cells, size_t cells_capacity, uint16_t* items,
                                size_t items_capacity, uint32_t* ranges, uint32_t* slot_order) {
  if (!boxes || !info || n == 0 || n > 65535) return fail(DUST_ERR_INVALID_ARGUMENT, "bad top-level build arguments");
  STRUCT_TRY(info, "DustTopLevelInfo");
  return guarded([&]() -> DustStatus {
    DustHipScene s;   // (a bare scene record: no context, no device: only what build_grid / order_slots read and write)
    s.world_boxes.assign(boxes, boxes + size_t(n) * 6);
    for (int a = 0; a < 3; ++a) { s.world_min[a] = 1e30f; s.world_max[a] = -1e30f; }
    for (uint32_t i = 0; i < n; ++i)
      for (int a = 0; a < 3; ++a) {
        if (!(boxes[i * 6 + a] <= boxes[i * 6 + 3 + a])) return fail(DUST_ERR_INVALID_ARGUMENT, "a box with lo > hi (or NaN)");
        s.world_min[a] = std::min(s.world_min[a], boxes[i * 6 + a]); s.world_max[a] = std::max(s.world_max[a], boxes[i * 6 + 3 + a]);
      }
    std::vector<uint32_t> rg;
    build_grid(&s, s.world_boxes, rg);
    if (!s.grid_valid) return fail(DUST_ERR_UNSUPPORTED, "more than 4095 boxes over one grid cell: no grid lists them (a scene renders by the packet kernels then)");
    order_slots(&s);
    for (int a = 0; a < 3; ++a) { info->dim[a] = s.grid.dim[a]; info->lo[a] = s.grid.lo[a]; info->cell[a] = s.grid.cell[a]; }
    info->n_cells = uint32_t(s.grid_cells.size());
    info->n_items = uint32_t(s.grid_items.size());
    info->n_groups = n > dust::kFlatCullMax ? (n + 63) / 64 : 0;
    if (cells) { if (cells_capacity < s.grid_cells.size()) return fail(DUST_ERR_INVALID_ARGUMENT, "cells buffer too small"); std::memcpy(cells, s.grid_cells.data(), s.grid_cells.size() * 4); }
    if (items) { if (items_capacity < s.grid_items.size()) return fail(DUST_ERR_INVALID_ARGUMENT, "items buffer too small"); std::memcpy(items, s.grid_items.data(), s.grid_items.size() * 2); }
    if (ranges) std::memcpy(ranges, rg.data(), size_t(n) * 8);
    if (slot_order) std::memcpy(slot_order, s.slot_order.data(), size_t(n) * 4);
    return DUST_OK;
  });
}

static void destroy_pipeline(DustHipPipeline* p) {
  if (!p) return;
  DustHipContext* c = p->ctx;
  (void)hipSetDevice(c->device);
  (void)sync_stream(c);  // both streams: the surfel pass writes the GI buffers on the second one
  for (auto& kind : p->ev_ring)
    for (auto& side : kind)
      for (auto& e : side) if (e) (void)hipEventDestroy(e);
  if (p->host_stats) (void)hipHostFree(p->host_stats);
  for (hipEvent_t e : {p->side_cal.p0, p->side_cal.p1, p->side_cal.q0, p->side_cal.q1}) if (e) (void)hipEventDestroy(e);
  delete p;
  release(c);
}
DustStatus dust_hip_pipeline_create(DustHipContext* ctx, uint32_t width, uint32_t height, DustHipPipeline** out) {
  if (!ctx || !out || width == 0 || height == 0 || width > 16384 || height > 16384)
    return fail(DUST_ERR_INVALID_ARGUMENT, "bad pipeline size");
  return guarded([&]() -> DustStatus {
    HIP_TRY(hipSetDevice(ctx->device));
    struct Drop { DustHipPipeline* p; ~Drop() { destroy_pipeline(p); } } owner{new DustHipPipeline};
    DustHipPipeline* p = owner.p;
    p->ctx = retain(ctx);
    p->tune = Tuning::from_environment();
    p->width = width; p->height = height;
    const hipStream_t st = ctx->stream;  // (every fill below is ordered on the stream the frames run on)
    const size_t px = size_t(width) * height;
    for (int i = 0; i < DUST_PLANE_COUNT; ++i) {
      HIP_TRY(p->planes[i].alloc(px * kPlaneBytesPerPixel[i]));
      HIP_TRY(hipMemsetAsync(p->planes[i].p, 0, px * kPlaneBytesPerPixel[i], st));
    }
    HIP_TRY(p->counters.alloc(8 * dust::kRegions * dust::kCounterStride * sizeof(uint32_t)));
    HIP_TRY(hipMemsetAsync(p->counters.p, 0, 8 * dust::kRegions * dust::kCounterStride * sizeof(uint32_t), st));
    HIP_TRY(p->stats.alloc(8 * sizeof(dust::DevStats)));
    HIP_TRY(hipMemsetAsync(p->stats.p, 0, 8 * sizeof(dust::DevStats), st));
    HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&p->host_stats), 8 * sizeof(dust::DevStats), hipHostMallocDefault));
    std::memset(p->host_stats, 0, 8 * sizeof(dust::DevStats));
    HIP_TRY(p->exposure.alloc(257 * 4));
    HIP_TRY(hipMemsetAsync(p->exposure.p, 0, 257 * 4, st));  // auto_exposure.rs:117: fill_buffer(0)
    // timing only (nothing waits on them for visibility): without the system-scope fence a record does not flush L2 between passes
    if (ctx->timing)
      for (auto& kind : p->ev_ring)
        for (auto& side : kind) {
          side.assign(DustHipPipeline::kEvRing, nullptr);
          for (auto& e : side) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableSystemFence));
        }
    HIP_TRY(hipStreamSynchronize(st));
    owner.p = nullptr;
    *out = p;
    return DUST_OK;
  });
}
void dust_hip_pipeline_destroy(DustHipPipeline* p) { destroy_pipeline(p); }
DustStatus dust_hip_pipeline_set_noise(DustHipPipeline* p, uint32_t texture, const uint8_t* texels, uint32_t layers) {
  if (!p || !texels || layers == 0 || (texture != 0 && texture != 5))
    return fail(DUST_ERR_INVALID_ARGUMENT, "noise texture must be 0 (scalar R8) or 5 (unitvec3_cosine RGBA8)");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));  // (frames that read the old texture are done before it is replaced)
  const size_t bytes = size_t(128) * 128 * layers * (texture == 0 ? 1 : 4);
  if (texture == 0) { HIP_TRY(p->noise0.upload(texels, bytes, p->ctx->stream)); p->noise0_layers = layers; }
  else { HIP_TRY(p->noise5.upload(texels, bytes, p->ctx->stream)); p->noise5_layers = layers; }
  return DUST_OK;
}

extern "C" DustStatus dust_hip_pipeline_configure_gi(DustHipPipeline* p, uint32_t hash_capacity, uint32_t surfel_pool_size);
// copies one launch descriptor into the next ring slot (pinned host -> device, on the launch stream)

// Work counters without a memset per launch: every pass kind owns two sets; a launch pulls tiles from one and its
// first workgroup zeroes the other, which is the set the next launch of that kind (stream-ordered behind it) will use.
// Cost-ordered hand-out for the launch about to be made (kernels.hip, k_tile_order): orders the tiles by what the pass's
// previous launch measured, if that was on the same tile grid, and has this launch measure again.
// While the view stands still (same camera, scene revision, sun and rows) the costs do too: the order is kept and re-measured
// only every kOrderRefresh launches, which takes k_tile_order (~8 us) and the cost recording out of most frames; a moving view
// measures and re-orders on every launch.
// An order that a re-measurement of the same view has just confirmed is trusted for twice as long, up to 64 launches (the GI
// kernels' costs drift as the hash fills: they keep being looked at).
constexpr uint32_t kOrderRefresh = 8, kOrderRefreshMax = 64;
static DustStatus order_tiles(DustHipPipeline* p, uint32_t kind, dust::FrameArgs& a, hipStream_t st) {
  a.tile_order = nullptr; a.tile_cost = nullptr; a.band_cuts = nullptr;
  if (p->tune.no_tile_order) return DUST_OK;
  DustHipPipeline::TileHistory& h = p->tile_history[kind];
  const uint32_t total = a.tiles_x * a.tiles_y;
  const uint32_t per_band = (total + dust::kRegions - 1) / dust::kRegions;
  if (per_band > dust::kTileOrderMaxBand) return DUST_OK;  // beyond 8K: screen order
  if (total > h.capacity) {
    HIP_TRY(hipStreamSynchronize(st));
    HIP_TRY(h.cost.alloc(size_t(total) * 4)); HIP_TRY(h.order.alloc(size_t(total) * 4)); HIP_TRY(h.smooth.alloc(size_t(total) * 4)); HIP_TRY(h.spread.alloc(size_t(total) * 4));
    if (!h.cuts.p) HIP_TRY(h.cuts.alloc(size_t(dust::kRegions + 1) * 4));
    h.capacity = total; h.tiles_x = h.tiles_y = 0;
  }
  if (h.tiles_x != a.tiles_x || h.tiles_y != a.tiles_y) {  // a new grid: tiles nobody has timed count as free
    h.recorded = false; h.ordered = false; h.measured = false; h.tiles_x = a.tiles_x; h.tiles_y = a.tiles_y;
    HIP_TRY(hipMemsetAsync(h.cost.p, 0, size_t(total) * 4, st));
    HIP_TRY(hipMemsetAsync(h.smooth.p, 0, size_t(total) * 4, st));
  }
  if (h.recorded) {
    // (the cost-balanced cuts drift slowly: a view that moves keeps them for kCutsReuse re-orderings -- the scan for them is the longer half of the sorter)
    const bool reuse = h.ordered && h.cuts_age + 1 < p->tune.cuts_reuse && h.moving;
    HIP_TRY(dust::launch_cost_blend(static_cast<const uint32_t*>(h.cost.p), static_cast<uint32_t*>(h.smooth.p), total, p->tune.cost_keep_shift, st));
    const bool spread = h.moving && p->tune.dilate && a.tiles_y > 1u;
    if (spread) HIP_TRY(dust::launch_cost_dilate(static_cast<const uint32_t*>(h.smooth.p), static_cast<uint32_t*>(h.spread.p), a.tiles_x, a.tiles_y, st));
    HIP_TRY(dust::launch_tile_order(static_cast<const uint32_t*>(spread ? h.spread.p : h.smooth.p), static_cast<uint32_t*>(h.order.p),
                                    p->tune.equal_bands ? nullptr : static_cast<uint32_t*>(h.cuts.p), reuse, total, per_band, st));
    h.cuts_age = reuse ? h.cuts_age + 1 : 0;
    h.recorded = false; h.ordered = true; h.age = 0;
  } else if (h.ordered) {
    ++h.age;
  }
  if (h.ordered) {
    a.tile_order = static_cast<const uint32_t*>(h.order.p);
    if (!p->tune.equal_bands) a.band_cuts = static_cast<const uint32_t*>(h.cuts.p);
  }
  const bool still = h.ordered && h.view == p->view_key && !p->tune.force_moving;
  if (!still) h.refresh = kOrderRefresh;
  // A view that moves: tile costs shift by a fraction of a tile per frame, so the order of a few frames ago is still a good one -- it is
  // re-measured (and the next launch re-ordered) every kMovingRefresh launches, not every launch: the sorter is a launch of its own
  // between two frames (~10 us of a 230 us frame). The first launch after a standstill (a cut, a teleport) is measured at once.
  const bool jumped = !still && !h.moving;
  const uint32_t period = still ? std::min(h.refresh, p->tune.still_refresh_max) : p->tune.moving_refresh;
  if (!h.ordered || jumped || h.age + 1 >= period) {  // measure this launch (each traced tile overwrites its cost): the next one re-orders
    a.tile_cost = static_cast<uint32_t*>(h.cost.p);
    h.recorded = true; h.measured = true;
    if (still) h.refresh = std::min(kOrderRefreshMax, h.refresh * 2u);
  }
  h.moving = !still && h.measured && h.view != 0 && (h.view != p->view_key || p->tune.force_moving);  // (the first launch of a view that stands still is not a moving one;
                                                                                                     //  DUST_HIP_FORCE_MOVING: a still view on a moving view's schedule)
  h.view = p->view_key;
  return DUST_OK;
}
static void take_counters(DustHipPipeline* p, uint32_t kind, dust::FrameArgs& a) {
  uint32_t* base = static_cast<uint32_t*>(p->counters.p) + size_t(kind) * 2 * dust::kRegions * dust::kCounterStride;
  const uint32_t par = p->counter_parity[kind];
  a.work_counters = base + size_t(par) * dust::kRegions * dust::kCounterStride;
  a.next_work_counters = base + size_t(par ^ 1u) * dust::kRegions * dust::kCounterStride;
  p->counter_parity[kind] = par ^ 1u;
}

// The ray stream of pass kind `kind` (0: final gather, 1: surfel pass) for the launch about to be made: its buffers, this launch's
// ray counter (zero: the previous launch of the kind zeroed it) and the one the next launch will use.
static void stream_args(DustHipPipeline* p, int kind, dust::FrameArgs& a, float tmin, float tmax) {
  a.stream.unbinned = static_cast<uint32_t*>(p->gi_unbinned.p) + kind * 2;
  a.stream.rays = static_cast<dust::DevRay*>(kind == 0 ? p->gi_rays_fg.p : p->gi_rays_sf.p);
  a.stream.group_count = static_cast<uint32_t*>(kind == 0 ? p->gi_groups_fg.p : p->gi_groups_sf.p);
  if (kind == 0) {  // a group = a 16 x 16 pixel tile of the band (k_gather_rays)
    a.stream.n_groups = ((p->width + 15u) / 16u) * ((a.row_end - a.row_begin + 15u) / 16u);
    a.stream.group_rays = 256;
  } else {          // a group = 256 consecutive surfels of the (ordered) pool, two rays each (k_surfel_rays)
    a.stream.n_groups = (p->gi_pool_size + 255u) / 256u;
    a.stream.group_rays = 512;
  }
  a.stream.ray_hits = static_cast<dust::DevGatherHit*>(kind == 0 ? p->gi_fg_hits.p : p->gi_hits_sf.p);
  a.stream.ray_tmin = tmin; a.stream.ray_tmax = tmax;
  a.tile_order = nullptr; a.tile_cost = nullptr; a.band_cuts = nullptr;  // (the stream hands out rays, not tiles)
  a.tiles_x = a.tiles_y = 1;
}
// The ray streams' buffers (gi.hip), made when a frame first takes a stream path -- the packet kernels, the default everywhere but the
// final gather of a scene with a 4096^3 tree, never read them (130 MB at 1080p, 530 MB at 4K): a ray per pixel / two per surfel at most,
// the hit records, the group counters. Waits for the streams once (an allocation is not stream-ordered).
static DustStatus ensure_stream_buffers(DustHipPipeline* p, bool gather, bool surfel) {
  const bool need_fg = gather && !p->gi_rays_fg.p, need_sf = surfel && !p->gi_rays_sf.p;
  if (!need_fg && !need_sf && p->gi_unbinned.p) return DUST_OK;
  HIP_TRY(sync_stream(p->ctx));
  if (!p->gi_unbinned.p) {
    HIP_TRY(p->gi_unbinned.alloc(4 * 4));
    HIP_TRY(hipMemsetAsync(p->gi_unbinned.p, 0, 4 * 4, p->ctx->stream));
  }
  if (need_fg) {
    const size_t tiles = size_t((p->width + 15) / 16) * ((p->height + 15) / 16);
    HIP_TRY(p->gi_rays_fg.alloc(tiles * 256 * sizeof(dust::DevRay)));
    HIP_TRY(p->gi_groups_fg.alloc(tiles * 4));
    HIP_TRY(p->gi_fg_hits.alloc(size_t(p->width) * p->height * sizeof(dust::DevGatherHit)));
  }
  if (need_sf) {
    const size_t runs = (size_t(p->gi_pool_size) + 255) / 256;
    HIP_TRY(p->gi_rays_sf.alloc(runs * 512 * sizeof(dust::DevRay)));
    HIP_TRY(p->gi_groups_sf.alloc(runs * 4));
    HIP_TRY(p->gi_hits_sf.alloc(size_t(p->gi_pool_size) * 2 * sizeof(dust::DevGatherHit)));
  }
  return DUST_OK;
}
// The surfel pass of one frame (surfel.rgen + the spatial hash update) on stream `st`, on at most `resident` workgroup slots.
// the recorded hash inserts, applied in surfel-index order (DUST_PASS_GI_ORDERED): in parallel over independent probe-window clusters (the serial
// one-wavefront loop it is checked against stays reachable through DUST_HIP_DEBUG bit 16)
static DustStatus apply_ordered(DustHipPipeline* p, dust::FrameArgs& b, hipStream_t st) {
  uint32_t* sk[2] = {static_cast<uint32_t*>(p->gi_sort_keys[0].p), static_cast<uint32_t*>(p->gi_sort_keys[1].p)};
  uint32_t* sv[2] = {static_cast<uint32_t*>(p->gi_sort_vals[0].p), static_cast<uint32_t*>(p->gi_sort_vals[1].p)};
  // keys -> sort by hash location -> marks (which requests the frame applies: k_surfel_apply_mark) -> the apply, parallel over clusters
  // or (DUST_HIP_DEBUG bit 16) the serial loop in surfel order it is checked against
  b.gi.sort_keys = sk[0];
  b.gi.sort_vals = sv[0];
  b.apply_alive = static_cast<unsigned long long*>(p->gi_apply_alive.p);
  b.apply_dead = static_cast<uint8_t*>(p->gi_apply_dead.p);
  b.apply_words = (p->gi_pool_size + 63u) / 64u;
  b.apply_starts = b.apply_alive + b.apply_words + 1;
  HIP_TRY(dust::launch_surfel_apply(b, 2, st));
  uint32_t bits = 1;
  while ((1ull << bits) <= uint64_t(p->gi_capacity)) ++bits;  // locations 0 .. capacity (capacity itself = "no insert")
  bool in_b = false;
  HIP_TRY(dust::radix_sort_pairs(p->gi_sort_scratch.p, sk[0], sv[0], sk[1], sv[1], p->gi_pool_size, bits, &in_b, st));
  b.gi.apply_keys = sk[in_b ? 1 : 0];
  b.gi.apply_vals = sv[in_b ? 1 : 0];
  HIP_TRY(dust::launch_surfel_apply(b, 4, st));
  HIP_TRY(dust::launch_surfel_apply(b, (p->tune.debug & 16u) ? 1 : 3, st));
  return DUST_OK;
}
// shard_world >= 1: the trace of rank shard_rank's share of the ordered pool only, records staged in slot order, nothing applied
// (dust_hip_gi_surfel_exchange_run completes the pass)
static DustStatus run_surfel_pass(DustHipPipeline* p, const dust::FrameArgs& a, uint32_t passes, bool count, hipStream_t st, uint32_t resident,
                                  uint32_t shard_rank = 0, uint32_t shard_world = 0) {
  DustHipContext* ctx = p->ctx;
  const Tuning& tune = p->tune;
  const uint32_t block = tune.block;
    dust::FrameArgs b = a;  // 64 consecutive surfels x one ray kind per wavefront: one row of "tiles", cosine items then sun items
    // on the second stream the pass is the longer of the two sides that share the SIMDs: its waves win the issue arbitration
    if (st != ctx->stream) b.prio_floor = tune.side_prio;
    b.tiles_x = 2 * ((p->gi_pool_size + 63) / 64);
    b.tiles_y = 1;
    b.stats = static_cast<dust::DevStats*>(p->stats.p) + 4;
    uint32_t* sk[2] = {static_cast<uint32_t*>(p->gi_sort_keys[0].p), static_cast<uint32_t*>(p->gi_sort_keys[1].p)};
    uint32_t* sv[2] = {static_cast<uint32_t*>(p->gi_sort_vals[0].p), static_cast<uint32_t*>(p->gi_sort_vals[1].p)};
    b.gi.sort_keys = sk[0];
    b.gi.sort_vals = sv[0];
    if (p->timed_frame) HIP_TRY(hipEventRecord(p->ev_begin(3), st));
    if (!tune.no_surfel_sort) {  // phase 0: 16-bit space-filling-curve keys + radix sort -> gi.perm
      HIP_TRY(dust::launch_surfel_keys(b, st));
      bool in_b = false;
      HIP_TRY(dust::radix_sort_pairs(p->gi_sort_scratch.p, sk[0], sv[0], sk[1], sv[1], p->gi_pool_size, 16, &in_b, st));
      b.gi.perm = sv[in_b ? 1 : 0];
    }
    const bool staged = shard_world >= 1;
    if (staged) {
      const uint32_t groups = (p->gi_pool_size + 63u) / 64u, per = (groups + shard_world - 1u) / shard_world;
      if (!p->gi_stage_req.p) {   // room for any world up to 64: a rank's share ends on a group boundary
        const size_t cap = size_t(groups + 64u) * 64u;
        HIP_TRY(hipStreamSynchronize(st));
        HIP_TRY(p->gi_stage_req.alloc(cap * sizeof(dust::DevHashRequest)));
        HIP_TRY(p->gi_stage_repl.alloc(cap * 16));
        HIP_TRY(p->gi_stage_sun.alloc(cap * 16));
        HIP_TRY(hipMemsetAsync(p->gi_stage_req.p, 0, cap * sizeof(dust::DevHashRequest), st));
        HIP_TRY(hipMemsetAsync(p->gi_stage_repl.p, 0xFF, cap * 16, st));   // direction 0xFFFFFFFF: "keep"
        HIP_TRY(hipMemsetAsync(p->gi_stage_sun.p, 0, cap * 16, st));
      }
      b.sf_stage_req = static_cast<dust::DevHashRequest*>(p->gi_stage_req.p);
      b.sf_stage_repl = static_cast<dust::DevSurfel*>(p->gi_stage_repl.p);
      b.sf_stage_sun = static_cast<float*>(p->gi_stage_sun.p);
      b.sf_group_begin = std::min(groups, shard_rank * per);
      b.sf_group_count = std::min(per, groups - b.sf_group_begin);
      b.tiles_x = 2 * b.sf_group_count;   // (a rank past the end of the pool traces nothing)
      p->sf_shard.pending = true; p->sf_shard.perm = b.gi.perm; p->sf_shard.rank = shard_rank; p->sf_shard.world = shard_world;
      p->sf_shard.slots_per_rank = per * 64u;
    }
    if (tune.packet_gi() || !b.grid.cells || staged) {   // (a sharded trace runs as packets: the stream's shading kernel writes by surfel index)
      if (b.tiles_x) {   // (a rank past the end of the pool traces nothing)
        take_counters(p, 3, b);
        { DustStatus os = order_tiles(p, 3, b, st); if (os != DUST_OK) return os; }
        const uint32_t sgrid = std::max(8u, std::min<uint32_t>(resident, (b.tiles_x + 7) / 8));
        HIP_TRY(dust::launch_surfel_trace(b, sgrid, block, count, st));
      }
    } else {
      // phase 1 as a ray stream (gi.hip): the pool's rays, compacted -> one ray per lane, lanes refilled -> the hash lookups over the hit records
      stream_args(p, 1, b, 0.1f, 10000.0f);  // surfel.rgen:33-62
      take_counters(p, 3, b);
      b.stream.count_unbinned = count ? 1u : 0u;
      if (count) HIP_TRY(hipMemsetAsync(b.stream.unbinned, 0, 2 * 4, st));
      HIP_TRY(dust::launch_surfel_rays(b, st));
      // (one 1024-thread workgroup per CU: sixteen waves share one staged copy of the top-level data)
      const uint32_t want = (p->gi_pool_size * 2u + 1023u) / 1024u;
      const uint32_t sgrid = std::max(8u, std::min<uint32_t>((resident * block / 1024u) & ~7u, (want + 7u) & ~7u));
      HIP_TRY(dust::launch_ray_walk(b, 3, sgrid, 1024, count, st));
      HIP_TRY(dust::launch_surfel_shade(b, st));
    }
    // phase 2: apply the recorded inserts. Default: concurrently, like the reference's shaders. DUST_PASS_GI_ORDERED: the
    // result of applying them in surfel-index order (apply_ordered). A sharded trace stops here: its records are not complete yet.
    if (staged) {
    } else if (!(passes & DUST_PASS_GI_ORDERED)) {
      HIP_TRY(dust::launch_surfel_apply(b, 0, st));
    } else {
      DustStatus as = apply_ordered(p, b, st);
      if (as != DUST_OK) return as;
    }
    if (p->timed_frame) { HIP_TRY(hipEventRecord(p->ev_end(3), st)); p->ev_valid[3] = true; }
  return DUST_OK;
}
// Several frames in one persistent launch (dust_hip_render_frames): the frames are PREPARED in order -- the scene as each of them sees it, its
// descriptor, work counters, tile order -- and left in `frames[]`; the preparation of the last one launches k_primary_ao_batch over all of them.
struct BatchJoin {
  dust::FrameArgs frames[dust::kMaxBatch];
  uint32_t image_of[dust::kMaxBatch] = {};   // which of the scene's ring of device images each frame reads (the scene may be committed between two frames of a launch)
  DustHipPipeline* timer = nullptr;          // frame 0's pipeline: its HIP-event pair brackets the launch
  uint32_t n = 0;      // frames of the launch
  uint32_t slot = 0;   // the frame being prepared
};
// Follower: frames 0 .. n - 2 of a launch, prepared in order; Lead: frame n - 1 -- prepared last, launches all of them
enum class FrameRole { Single, Follower, Lead };
// every argument check of a frame, before anything is enqueued or changed (a frame that has started is finished); fp_copy: the caller's
// parameters widened to this library's struct
static DustStatus check_frame(DustHipPipeline* p, const DustHipScene* s, const DustHipCamera* cam, const DustHipSky* sky,
                              const DustHipFrameParams* fp_in, DustHipFrameParams& fp_copy) {
  if (!p || !s || !cam || !sky || !fp_in) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  // (round 6 appended surfel_rank / surfel_world: a caller compiled against the struct that ends at row_end gets zeroes for them)
  if (fp_in->struct_size < offsetof(DustHipFrameParams, surfel_rank)) return fail(DUST_ERR_INVALID_ARGUMENT, "DustHipFrameParams.struct_size is smaller than this library's DustHipFrameParams");
  fp_copy = DustHipFrameParams{};
  std::memcpy(&fp_copy, fp_in, std::min<size_t>(fp_in->struct_size, sizeof fp_copy));
  const DustHipFrameParams* fp = &fp_copy;
  if (p->ctx != s->ctx) return fail(DUST_ERR_INVALID_ARGUMENT, "pipeline and scene belong to different contexts");
  if (!s->committed) return fail(DUST_ERR_NOT_READY, "scene has uncommitted changes (call dust_hip_scene_commit)");
  for (size_t i = 0; i < s->models.size(); ++i)
    if (s->models[i]->generation != s->model_generation[i])
      return fail(DUST_ERR_NOT_READY, "a model of the scene was edited after the last dust_hip_scene_commit");
  const uint32_t need5 = DUST_PASS_AMBIENT_OCCLUSION | DUST_PASS_FINAL_GATHER | DUST_PASS_SURFEL;
  if ((fp->passes & need5) && !p->noise5.p)
    return fail(DUST_ERR_NOT_READY, "blue-noise texture 5 (unitvec3_cosine) not loaded");  // standard.rs:254
  if ((fp->passes & (DUST_PASS_FINAL_GATHER | DUST_PASS_SURFEL)) && !p->noise0.p)
    return fail(DUST_ERR_NOT_READY, "blue-noise texture 0 (scalar) not loaded");
  const bool sharded = (fp->passes & DUST_PASS_GI_SHARDED) != 0;
  // (argument checks all come before the first launch: a frame that has started is finished)
  if ((fp->passes & DUST_PASS_ACCUMULATE) && (fp->passes & DUST_PASS_DENOISE))
    return fail(DUST_ERR_INVALID_ARGUMENT, "DUST_PASS_ACCUMULATE and DUST_PASS_DENOISE both keep their running result in DUST_PLANE_ACCUM: one per frame");
  if ((fp->passes & DUST_PASS_DENOISE) && (fp->row_begin != 0 || (fp->row_end != 0 && fp->row_end != p->height)))
    return fail(DUST_ERR_UNSUPPORTED, "DUST_PASS_DENOISE reprojects and blurs across rows: run it on the whole (gathered) frame");
  if (!sharded && (fp->passes & (DUST_PASS_FINAL_GATHER | DUST_PASS_SURFEL)) && (fp->row_begin != 0 || (fp->row_end != 0 && fp->row_end != p->height)))
    return fail(DUST_ERR_UNSUPPORTED, "a GI pass on a row band needs DUST_PASS_GI_SHARDED and the exchange of dust_hip_pipeline_gi_exchange");
  if (sharded && (fp->passes & DUST_PASS_FINAL_GATHER) && (fp->passes & DUST_PASS_SURFEL))
    return fail(DUST_ERR_INVALID_ARGUMENT, "DUST_PASS_GI_SHARDED: the surfel pass runs after the exchange, in its own call");
  if (fp->surfel_world != 0 && (fp->passes & DUST_PASS_SURFEL) && (!sharded || fp->surfel_world > 64 || fp->surfel_rank >= fp->surfel_world))
    return fail(DUST_ERR_INVALID_ARGUMENT, "a sharded surfel trace wants DUST_PASS_GI_SHARDED, surfel_world <= 64 and surfel_rank < surfel_world");
  if (p->sf_shard.pending && (fp->passes & (DUST_PASS_FINAL_GATHER | DUST_PASS_SURFEL)))
    return fail(DUST_ERR_NOT_READY, "a sharded surfel trace is pending on this pipeline: dust_hip_gi_surfel_exchange_run completes it before the next GI pass");
  if (sharded && (fp->passes & DUST_PASS_FINAL_GATHER) && !p->gi_touched.p)
    return fail(DUST_ERR_NOT_READY, "DUST_PASS_GI_SHARDED: call dust_hip_pipeline_gi_exchange first");
  {
    const uint32_t re = fp->row_end ? fp->row_end : p->height;
    if (fp->row_begin >= re || re > p->height) return fail(DUST_ERR_INVALID_ARGUMENT, "bad row range");
  }
  return DUST_OK;
}
static DustStatus render_frame_impl(DustHipPipeline* p, const DustHipScene* s, const DustHipCamera* cam, const DustHipSky* sky,
                                    const DustHipFrameParams* fp_in, FrameRole role, BatchJoin* join) {
  DustHipFrameParams fp_copy{};
  { DustStatus cs = check_frame(p, s, cam, sky, fp_in, fp_copy); if (cs != DUST_OK) return cs; }
  const DustHipFrameParams* fp = &fp_copy;
  const bool sharded = (fp->passes & DUST_PASS_GI_SHARDED) != 0;
  if ((fp->passes & (DUST_PASS_FINAL_GATHER | DUST_PASS_SURFEL)) && !p->gi_hash.p) {
    DustStatus gs = dust_hip_pipeline_configure_gi(p, dust::kSpatialHashCapacity, dust::kSurfelPoolSize);
    if (gs != DUST_OK) return gs;
  }
  DustHipContext* ctx = p->ctx;
  HIP_TRY(hipSetDevice(ctx->device));
  dust::FrameArgs a{};
  s->touch();
  a.models = reinterpret_cast<const dust::DevModel*>(s->dev(s->layout.models));
  a.instances = reinterpret_cast<const dust::DevInstance*>(s->dev(s->layout.instances));
  a.n_models = uint32_t(s->models.size());
  a.n_instances = uint32_t(s->instances.size());
  a.n_lds_models = s->n_lds_models;
  a.root_table = s->dev(s->layout.root_table);
  a.boxes = reinterpret_cast<const dust::DevBox*>(s->dev(s->layout.boxes));
  a.visits = reinterpret_cast<const dust::DevVisit*>(s->dev(s->layout.visits));
  a.grid = s->grid;
  // (a grid that could not list every box -- build_grid -- is not handed to the kernels at all: the packet kernels, which never read it, run instead)
  a.grid.cells = s->grid_valid ? reinterpret_cast<const uint32_t*>(s->dev(s->layout.grid_cells)) : nullptr;
  a.grid.items = s->grid_valid ? reinterpret_cast<const uint16_t*>(s->dev(s->layout.grid_items)) : nullptr;
  a.enters = reinterpret_cast<const dust::DevEnter*>(s->dev(s->layout.enters));
  a.gboxes = reinterpret_cast<const dust::DevBox*>(s->dev(s->layout.gboxes));
  a.sboxes = reinterpret_cast<const dust::DevBox*>(s->dev(s->layout.sboxes));
  a.n_groups = s->n_groups;
  a.stream_refill = p->tune.stream_refill; a.stream_top_iters = p->tune.stream_top_iters;
  {  // what the ray-stream kernels stage in LDS, as far as it goes. The ray-making kernels (256 threads, many workgroups per CU): grid cells,
     // items and instance boxes within 40 KB; k_ray_walk (one 1024-thread workgroup per CU): the enter records behind its roots.
    auto layout = [&](size_t budget, bool bin) {
      dust::DevStreamLds l;
      size_t at = 0;
      auto place = [&](size_t bytes) -> uint32_t {
        bytes = (bytes + 15) & ~size_t(15);
        if (p->tune.no_stream_lds || at + bytes > budget) return 0xFFFFFFFFu;
        const uint32_t off = uint32_t(at);
        at += bytes;
        return off;
      };
      const size_t n_cells = size_t(a.grid.dim[0]) * a.grid.dim[1] * a.grid.dim[2];
      l.cells = bin ? place(n_cells * 4) : 0xFFFFFFFFu;
      l.items = bin ? place(size_t(a.grid.n_items) * 2) : 0xFFFFFFFFu;
      l.boxes = bin ? place(size_t(a.n_instances) * 32) : 0xFFFFFFFFu;
      l.enters = bin ? 0xFFFFFFFFu : place(size_t(a.n_instances) * sizeof(dust::DevEnter));
      l.total = uint32_t(at);
      return l;
    };
    a.sl_bin = layout(40 * 1024, true);
    a.sl_walk = layout(p->ctx->max_lds - std::min<size_t>(p->ctx->max_lds, size_t(a.n_lds_models) * dust::kN16LdsBytes), false);
  }
  for (int k = 0; k < 3; ++k) { a.world_min[k] = s->world_min[k]; a.world_max[k] = s->world_max[k]; }
  std::memcpy(a.cam.col0, cam->view_col0, 12); std::memcpy(a.cam.col1, cam->view_col1, 12);
  std::memcpy(a.cam.col2, cam->view_col2, 12); std::memcpy(a.cam.pos, cam->position, 12);
  a.cam.tan_half_fov = cam->tan_half_fov; a.cam.far_ = cam->far_; a.cam.near_ = cam->near_;
  std::memcpy(a.sky, sky->state, sizeof(a.sky));
  sun_constants(a.sky, a.sun_dir, a.sun_term);
  a.g.illuminance = static_cast<uint16_t*>(p->plane(DUST_PLANE_ILLUMINANCE));
  a.g.denoised = static_cast<uint16_t*>(p->plane(DUST_PLANE_DENOISED));
  a.g.albedo = static_cast<uint32_t*>(p->plane(DUST_PLANE_ALBEDO));
  a.g.normal = static_cast<uint32_t*>(p->plane(DUST_PLANE_NORMAL));
  a.g.depth = static_cast<float*>(p->plane(DUST_PLANE_DEPTH));
  a.g.motion = static_cast<uint16_t*>(p->plane(DUST_PLANE_MOTION));
  a.g.voxel_id = static_cast<uint32_t*>(p->plane(DUST_PLANE_VOXEL_ID));
  a.g.accum = static_cast<float*>(p->plane(DUST_PLANE_ACCUM));
  a.width = p->width; a.height = p->height;
  a.inv_width = 1.0f / float(p->width); a.inv_height = 1.0f / float(p->height); a.aspect = float(p->width) / float(p->height);
  a.row_begin = fp->row_begin;
  a.row_end = fp->row_end ? fp->row_end : p->height;
  if (a.row_begin >= a.row_end || a.row_end > p->height) return fail(DUST_ERR_INVALID_ARGUMENT, "bad row range");
  a.tiles_x = (p->width + dust::kTileW - 1) / dust::kTileW;
  a.tiles_y = (a.row_end - a.row_begin + dust::kTileH - 1) / dust::kTileH;
  a.rand = fp->rand; a.frame_index = fp->frame_index;
  if (p->noise0.p) a.noise0 = static_cast<const uint8_t*>(p->noise0.p) + size_t(fp->frame_index % p->noise0_layers) * 128 * 128;
  if (p->noise5.p) a.noise5 = static_cast<const uint8_t*>(p->noise5.p) + size_t(fp->frame_index % p->noise5_layers) * 128 * 128 * 4;  // noise.rs:50
  a.stats = static_cast<dust::DevStats*>(p->stats.p);
  a.accum_count = p->accum_count;
  const Tuning& tune = p->tune;
  a.debug = tune.debug;
  a.static_rounds_request = tune.static_rounds;
  for (const DustHipModel* m : s->models) a.deep |= m->dev.n_levels == 3 ? 1u : 0u;
  {  // which GI passes of this frame run as ray streams (decided once, here: the launches below ask the same questions)
    const bool fg_stream = (fp->passes & DUST_PASS_FINAL_GATHER) && a.grid.cells &&
                           (!tune.packet_gi() || (a.deep && !tune.packet_only() && !(tune.debug & 12u) && !tune.no_gather_order));
    const bool sf_stream = (fp->passes & DUST_PASS_SURFEL) && a.grid.cells && !tune.packet_gi();
    if (fg_stream || sf_stream) { DustStatus es = ensure_stream_buffers(p, fg_stream, sf_stream); if (es != DUST_OK) return es; }
  }
  const bool count = fp->passes & DUST_PASS_COUNT_STATS;
  const uint32_t block = tune.block;
  uint32_t bpc = tune.blocks_per_cu;
  size_t lds = size_t(a.n_lds_models) * dust::kN16LdsBytes + (block / 64) * (dust::kMaxCand * 8 + 8) + 16;
  if (lds > ctx->max_lds) return fail(DUST_ERR_INVALID_ARGUMENT, "staged roots and candidate lists exceed the device's LDS");
  // the instance boxes ride along when the workgroups of a CU still fit side by side (the packet cull reads all of them, per packet)
  a.n_lds_boxes = 0;
  const uint32_t cull_boxes = a.n_groups ? a.n_groups : a.n_instances;  // (a large scene stages the boxes of its groups of 64)
  if (!tune.no_lds_boxes && (lds + size_t(cull_boxes) * 32) * bpc <= 160 * 1024 && lds + size_t(cull_boxes) * 32 <= ctx->max_lds) {
    a.n_lds_boxes = cull_boxes;
    lds += size_t(cull_boxes) * 32;
  }
  while (bpc > 1 && lds * bpc > 160 * 1024) --bpc;
  const uint32_t total_tiles = a.tiles_x * a.tiles_y;
  // Workgroups per persistent launch: every slot of every CU, minus what the caller asks to be left free. The traversal
  // kernels hold all VGPRs of the SIMDs they run on, so a kernel of another queue (an RCCL send/receive moving the previous
  // frame to another GPU) can only become resident next to them where a workgroup slot was left empty.
  uint32_t resident = uint32_t(ctx->num_cus) * bpc;
  // (DUST_RESERVE_AUTO: nothing until the pipeline has been seen in a collective of world > 1 -- comm.hip says so --, 32 from then on: without
  //  them RCCL's kernels wait 36 us - 0.2 ms behind a persistent launch, with them under 10 us; they cost the traversal 5 %)
  const uint32_t reserve_blocks = (tune.reserve_blocks == DUST_RESERVE_AUTO ? (p->in_collective ? 32u : 0u) : tune.reserve_blocks) & ~7u;
  if (reserve_blocks && reserve_blocks + 8u <= resident) resident -= reserve_blocks;
  hipStream_t st = ctx->stream;
  p->stats_valid = false;
  // An event pair around a launch costs the stream ~6 us per record (a marker packet the next dispatch waits behind): 5 % of a
  // 0.23 ms frame. A context that only wants averages over a run of frames (bench.py) times every 4th frame's launches.
  // (the further frames of a batched launch have no launch of their own to time: frame 0's pair brackets the launch of all of them)
  // -- at the same rate PER FRAME as single launches are: a launch of n frames counts as n of the stride
  const uint32_t stride = role != FrameRole::Single ? std::max(1u, ctx->timing_stride / join->n) : ctx->timing_stride;
  const bool times_launch = role == FrameRole::Single || join->slot == 0;   // (a launch of several frames: frame 0's pipeline)
  p->timed_frame = times_launch && ctx->timing && (p->frame_counter++ % stride) == 0;
  // (a frame that is not timed has no times: dust_hip_pipeline_pass_stats must not hand out an earlier frame's)
  if (!p->timed_frame) for (bool& v : p->ev_valid) v = false;
  // while a surfel pass may be running on the second stream, the primary / AO kernels leave it its share of the slots (persistent
  // launches hold what they get: whichever came first would otherwise own the GPU until it is done)
  // The share: the pass's rays against the pixel passes' (pool x 18 surfel-ray costs to 3 rays per pixel, which puts the castle at
  // 50 % at 1080p and 20 % at 4K -- where round 2's feedback loop settled, without its calibration frame, event probes and waits)
  uint32_t share = tune.side_share;
  bool calibrate = false;
  if (!share) {
    auto& cal = p->side_cal;
    if (cal.state == 1 && hipEventQuery(cal.q1) == hipSuccess) {
      float P = 0.0f, Q = 0.0f;
      if (hipEventElapsedTime(&P, cal.p0, cal.p1) == hipSuccess && hipEventElapsedTime(&Q, cal.q0, cal.q1) == hipSuccess && P > 0.0f && Q > 0.0f)
        {
        // (a scene of many instances: the pass in place is as long as its longest items, 0.19 ms for 0.10 ms of work per wave, so it needs
        // fewer slots than its time says -- share sweeps of round 5's last session: 39 % at 1080p and 16 % at 4K where 1.05 gave 47 and 21,
        // GI frame 0.691 -> 0.667 ms. The 4096^3 tree's pass, one instance and long walks, is as long as its work: 1.05 stays there.)
        const float k = a.deep ? 1.05f : 1.4f;
        cal.share = uint32_t(std::min(65.0f, std::max(10.0f, 100.0f * Q / (Q + k * P) - 3.0f)));
      }
      cal.state = 2;
    }
    (void)hipGetLastError();  // (hipEventQuery's "not ready" is not an error of this call)
    const bool gi_frame = (fp->passes & DUST_PASS_PRIMARY) && (fp->passes & DUST_PASS_SURFEL);
    // (not the pipeline's first GI frames: they touch the hash and the pool for the first time -- 400 MB of first-touch page faults inside the
    //  timed pass; one deep-tree run in five calibrated a share half as large again from it: 4.66 ms per frame against 4.27-4.30)
    if (gi_frame && cal.state == 0) ++cal.gi_frames;
    calibrate = cal.state == 0 && cal.gi_frames >= 3u && gi_frame && !tune.no_side_stream && !count && !sharded && !(tune.debug & 16u);
    if (calibrate) HIP_TRY(join_side(ctx));   // (the timed frame runs its passes one after the other, behind the previous frame's surfel pass)
    if (calibrate && !cal.p0)
      for (hipEvent_t* e : {&cal.p0, &cal.p1, &cal.q0, &cal.q1}) HIP_TRY(hipEventCreateWithFlags(e, hipEventDisableSystemFence));
    if (cal.share) share = cal.share;
    else {
      const double surfel = double(p->gi_pool_size) * 18.0, pixel = 3.0 * double(p->width) * double(a.row_end - a.row_begin);
      share = uint32_t(std::min(60.0, std::max(15.0, 100.0 * surfel / (surfel + pixel))));
    }
  }
  const uint32_t side_slots = ctx->side_busy ? std::max(8u, (resident * share / 100u) & ~7u) : 0u;
  const uint32_t main_resident = std::max(8u, resident - std::min(resident - 8u, side_slots));
  // (a caller with several frames in flight, each on a pipeline of its own: this launch takes its share of the slots -- whole
  // rounds over the 8 XCDs -- and leaves the rest to the others, dust_hip_pipeline_set_frames_in_flight)
  // (DUST_IN_FLIGHT_ALL: every launch asks for all of them -- whole frames one behind the other on two or three streams: the next frame's
  //  workgroups become resident on the CUs the previous frame's last tiles have left)
  const bool share_slots = p->frames_in_flight > 1 && tune.in_flight_slots == DUST_IN_FLIGHT_SHARE;
  // (diagnostic IN_FLIGHT_OVERSUB = percent: each of the n launches asks for that much more than its 1/n -- the extra workgroups wait for a slot)
  const uint32_t frame_slots = share_slots ? std::min(main_resident, std::max(8u, ((main_resident / p->frames_in_flight) * (100u + tune.in_flight_oversub) / 100u) & ~7u)) : main_resident;
  const uint32_t grid = std::max(8u, std::min<uint32_t>(frame_slots, (total_tiles + 7) / 8));
  {  // FNV-1a over what decides a tile's cost
    uint64_t k = 1469598103934665603ull;
    auto mix = [&k](const void* data, size_t n) { const uint8_t* b = static_cast<const uint8_t*>(data); for (size_t i = 0; i < n; ++i) { k ^= b[i]; k *= 1099511628211ull; } };
    mix(cam, sizeof *cam); mix(&s, sizeof s); mix(&s->revision, sizeof s->revision); mix(sky->state, sizeof sky->state);
    mix(&a.row_begin, sizeof a.row_begin); mix(&a.row_end, sizeof a.row_end);
    p->view_key = k;
  }
  a.gi.hash = static_cast<uint32_t*>(p->gi_hash.p);
  a.gi.hash_capacity = p->gi_capacity;
  a.gi.pool = static_cast<dust::DevSurfel*>(p->gi_pool.p);
  a.gi.pool_size = p->gi_pool_size;
  a.gi.slot_owner = static_cast<uint32_t*>(p->gi_owner.p);
  a.gi.pixel_surfel = static_cast<dust::DevSurfel*>(p->gi_pixel_surfel.p);
  a.gi.requests = static_cast<dust::DevHashRequest*>(p->gi_requests.p);
  a.gi.replacement = static_cast<dust::DevSurfel*>(p->gi_replacement.p);
  a.gi.sun_payload = static_cast<float*>(p->gi_sun_payload.p);
  if (count) HIP_TRY(hipMemsetAsync(p->stats.p, 0, 8 * sizeof(dust::DevStats), st));
  // primary + AO in one launch unless told otherwise (DUST_HIP_NO_FUSE=1 keeps the reference's one-launch-per-pass shape)
  const bool fuse = (fp->passes & DUST_PASS_PRIMARY) && (fp->passes & DUST_PASS_AMBIENT_OCCLUSION) && !tune.no_fuse;
  p->fused_last = fuse;
  // the frame's first traversal launch tells the host that the frame has started (DustHipContext::started; dust_hip_scene_commit)
  bool start_said = false;
  auto say_start = [&](dust::FrameArgs& x) {
    if (start_said || !ctx->started) return;
    start_said = true;
    x.started_word = const_cast<uint32_t*>(ctx->started);
    x.started_seq = ++ctx->frame_seq;
    s->slots[s->current].last_seq = x.started_seq;
  };
  if (calibrate) HIP_TRY(hipEventRecord(p->side_cal.p0, st));
  if (role != FrameRole::Single && (!fuse || count)) return fail(DUST_ERR_INVALID_ARGUMENT, "a batched frame must be a fused primary + AO frame");  // (dust_hip_render_frames checks)
  if (fuse) {
    take_counters(p, 0, a);
    { DustStatus os = order_tiles(p, 0, a, st); if (os != DUST_OK) return os; }
    a.stats = static_cast<dust::DevStats*>(p->stats.p);
    if (role != FrameRole::Single) {   // a frame of a launch of several: prepared; the last one launches them all
      join->frames[join->slot] = a;
      join->image_of[join->slot] = s->current;
      if (join->slot == 0) join->timer = p;
      if (role == FrameRole::Follower) return DUST_OK;
    }
    DustHipPipeline* tp = role == FrameRole::Lead ? join->timer : p;   // whose event pair brackets the launch
    if (tp->timed_frame) HIP_TRY(hipEventRecord(tp->ev_begin(0), st));
    // One 1024-thread workgroup per CU when the kernel has the device to itself (no surfel pass beside it, one frame in flight,
    // no slots reserved, the default block size): the roots are staged once per CU and sixteen waves share a tile queue
    uint32_t fblock = block, fgrid = grid;
    const size_t batch_lds = role == FrameRole::Lead ? 16u * (join->n - 1u) : 0u;   // a tile queue per further frame
    if (lds + batch_lds > ctx->max_lds) return fail(DUST_ERR_INVALID_ARGUMENT, "staged roots and candidate lists exceed the device's LDS");
    const size_t lds_wide = size_t(a.n_lds_models) * dust::kN16LdsBytes + 16u * (dust::kMaxCand * 8 + 8) + 16 + size_t(a.n_lds_boxes) * 32 + batch_lds;
    if (tune.wide_fused && block == 512 && bpc == 2 && !ctx->side_busy && !share_slots && !reserve_blocks && lds_wide <= ctx->max_lds &&
        grid == resident) {
      fblock = 1024;
      fgrid = std::max(8u, std::min<uint32_t>(uint32_t(ctx->num_cus), (total_tiles + 7) / 8));
    } else if (tune.wide_fused && tune.wide_share && block == 512 && bpc == 2 && !ctx->side_busy && share_slots && p->frames_in_flight == 2 && !reserve_blocks &&
               lds_wide <= ctx->max_lds && grid == frame_slots && frame_slots * 2u == resident) {
      // two whole frames in flight, each on half of the slots: half of the CUs each, one 1024-thread workgroup per CU
      fblock = 1024;
      fgrid = std::max(8u, (frame_slots / 2u) & ~7u);
    }
    if (role == FrameRole::Lead) {
      say_start(join->frames[0]);   // (the kernel's first descriptor says it)
      for (uint32_t i = 0; i < join->n; ++i) {
        // Every scene image a frame of the launch reads is in use from NOW until the launch is done. dust_hip_scene_commit recycles an image by
        // `epoch` (has anybody waited for the stream since a frame reading it was enqueued?) and `last_seq`: both were stamped when the frame
        // was PREPARED (touch()), and a commit between then and now may have waited for the stream -- the image would pass for idle
        // (found by tools/stress_host.py frames: 17 to 19 frames per call with moves, the third launch's commits landing on the second's images).
        s->slots[join->image_of[i]].epoch = ctx->sync_epoch;
        s->slots[join->image_of[i]].last_seq = join->frames[0].started_seq;
        // the boxes staged in LDS are frame 0's image's: a frame of another image (an instance moved in between) reads its own from memory
        if (join->image_of[i] != join->image_of[0]) join->frames[i].n_lds_boxes = 0;
      }
      if ((tune.debug & 32u) || dust::launch_primary_ao_batch(join->frames, join->n, fgrid, fblock, st) != hipSuccess) {   // (DUST_HIP_DEBUG bit 32: as if refused)
        // (the launch carries 8.5 KB of kernel arguments -- probed on this runtime, which takes 16 KB. Should a runtime refuse it: the prepared
        //  frames one launch each, the same results)
        (void)hipGetLastError();
        for (uint32_t i = 0; i < join->n; ++i) HIP_TRY(dust::launch_primary_ao(join->frames[i], fgrid, fblock, false, st));
      }
    } else {
      say_start(a);
      HIP_TRY(dust::launch_primary_ao(a, fgrid, fblock, count, st));
    }
    a.started_word = nullptr;
    if (tp->timed_frame) { HIP_TRY(hipEventRecord(tp->ev_end(0), st)); tp->ev_valid[0] = true; tp->ev_valid[1] = false; }
  }
  if (!fuse && (fp->passes & DUST_PASS_PRIMARY)) {
    take_counters(p, 0, a);
    { DustStatus os = order_tiles(p, 0, a, st); if (os != DUST_OK) return os; }
    a.stats = static_cast<dust::DevStats*>(p->stats.p);
    if (p->timed_frame) HIP_TRY(hipEventRecord(p->ev_begin(0), st));
    say_start(a);
    HIP_TRY(dust::launch_primary(a, grid, block, count, st));
    a.started_word = nullptr;
    if (p->timed_frame) { HIP_TRY(hipEventRecord(p->ev_end(0), st)); p->ev_valid[0] = true; }
  }
  if (!fuse && (fp->passes & DUST_PASS_AMBIENT_OCCLUSION)) {
    take_counters(p, 1, a);
    { DustStatus os = order_tiles(p, 1, a, st); if (os != DUST_OK) return os; }
    a.stats = static_cast<dust::DevStats*>(p->stats.p) + 1;
    if (p->timed_frame) HIP_TRY(hipEventRecord(p->ev_begin(1), st));
    say_start(a);
    HIP_TRY(dust::launch_ambient_occlusion(a, grid, block, count, st));
    a.started_word = nullptr;
    if (p->timed_frame) { HIP_TRY(hipEventRecord(p->ev_end(1), st)); p->ev_valid[1] = true; }
  }
  if (calibrate) HIP_TRY(hipEventRecord(p->side_cal.p1, st));
  if (fp->passes & DUST_PASS_FINAL_GATHER) {
    if (sharded) {  // pixels that stamp nothing must read 0 after the all-gather
      a.gi.touched = static_cast<uint32_t*>(p->gi_touched.p);
      a.gi.merged = static_cast<dust::DevSurfel*>(p->gi_merged.p);
      HIP_TRY(hipMemsetAsync(a.gi.touched + size_t(a.row_begin) * p->width, 0, size_t(a.row_end - a.row_begin) * p->width * 4, st));
    }
    a.stats = static_cast<dust::DevStats*>(p->stats.p) + 3;
    // (a 4096^3 tree: long walks through one instance -- the one workload where a lane of its own per ray pays: 1.66 against 1.82 ms)
    if (a.grid.cells && (!tune.packet_gi() || (a.deep && !tune.packet_only() && !(tune.debug & 12u) && !tune.no_gather_order))) {
      // The pass as a ray stream (gi.hip): make and bin the band's gather rays (a thread per pixel) -> walk them one per lane, lanes refilled
      // (k_ray_walk) -> shade the hit records (a thread per pixel). Behind the previous frame's surfel pass, like the packet kernel: rays and
      // hit records touch no GI state and COULD run beside that pass, but two persistent launches sharing the slots both get slower
      // (the 4096^3 tree's GI frame: 5.40 ms beside it, 4.56 behind it).
      HIP_TRY(join_side(ctx));
      dust::FrameArgs g = a;
      stream_args(p, 0, g, 8.0f, a.cam.far_);  // final_gather.rgen:47-50
      g.gi.fg_hits = g.stream.ray_hits;
      take_counters(p, 2, g);
      g.stream.count_unbinned = count ? 1u : 0u;
      if (count) HIP_TRY(hipMemsetAsync(g.stream.unbinned, 0, 2 * 4, st));
      if (p->timed_frame) HIP_TRY(hipEventRecord(p->ev_begin(2), st));
      HIP_TRY(dust::launch_gather_rays(g, st));
      const uint32_t want = uint32_t((size_t(p->width) * (a.row_end - a.row_begin) + 1023u) / 1024u);
      const uint32_t slots = resident;
      const uint32_t ggrid = std::max(8u, std::min<uint32_t>((slots * block / 1024u) & ~7u, (want + 7u) & ~7u));
      HIP_TRY(dust::launch_ray_walk(g, 2, ggrid, 1024, count, st));
      dust::FrameArgs sh = a;   // (pixel order over the band)
      sh.gi.fg_hits = g.gi.fg_hits;
      HIP_TRY(dust::launch_final_gather_shade(sh, !sharded, st));
      if (p->timed_frame) { HIP_TRY(hipEventRecord(p->ev_end(2), st)); p->ev_valid[2] = true; }
    } else {
    dust::FrameArgs g = a;
    uint32_t ggrid = std::max(8u, std::min<uint32_t>(resident, (total_tiles + 7) / 8));
    if (!tune.no_gather_order) {  // pre-pass: regroup the band's live pixels by ray-direction octant
      const uint32_t otx = (p->width + 63) / 64, oty = (a.row_end - a.row_begin + 63) / 64;
      g.gi.order = static_cast<uint32_t*>(p->gi_order.p);
      g.gi.order_count = static_cast<uint32_t*>(p->gi_order_count.p);
      g.gi.order_tiles_x = otx;
      HIP_TRY(dust::launch_gather_order(g, otx * oty, st));
      // work items: 64 packets of 64 per tile, the empty ones skipped by the kernel
      g.tiles_x = otx * oty * 64u;
      g.tiles_y = 1;
      ggrid = std::max(8u, std::min<uint32_t>(resident, (g.tiles_x + 7) / 8));
    }
    // (Round 4 also built the gather as a trace-only kernel beside the previous frame's surfel pass + a shading pass over hit records, and
    //  round 3 as refilled ray lanes inside the packet kernel: both measured slower and were removed in round 6 -- docs/EXPERIMENTS.md.)
    HIP_TRY(join_side(ctx));  // the previous frame's surfel pass has written the hash and the pool this gather reads
    take_counters(p, 2, g);
    { DustStatus os = order_tiles(p, 2, g, st); if (os != DUST_OK) return os; }
    if (p->timed_frame) HIP_TRY(hipEventRecord(p->ev_begin(2), st));  // (behind the regrouping pre-pass: the gather kernel)
    HIP_TRY(dust::launch_final_gather(g, ggrid, block, count, !sharded, st));
    if (p->timed_frame) { HIP_TRY(hipEventRecord(p->ev_end(2), st)); p->ev_valid[2] = true; }
      }
  }
  if (fp->passes & DUST_PASS_SURFEL) {
    // On the context's second stream, behind this frame's final gather (see DustHipContext::side): the pass is a handful of
    // latency-bound launches around a trace that is as long as its longest ray, and nothing of THIS frame waits for it.
    const bool aside = !tune.no_side_stream && !count && !sharded && !(tune.debug & 16u) && !calibrate;
    if (aside) HIP_TRY(fork_side(ctx));
    else HIP_TRY(join_side(ctx));
    if (calibrate) HIP_TRY(hipEventRecord(p->side_cal.q0, st));
    // beside the next frame's kernels it takes a share of the workgroup slots (both are persistent launches: with all slots
    // taken by the first, the second would simply run after it)
    const uint32_t side_resident = std::max(8u, (resident * share / 100u) & ~7u);
    DustStatus rs = run_surfel_pass(p, a, fp->passes, count, aside ? ctx->side : st, aside ? side_resident : resident, fp->surfel_rank, fp->surfel_world);
    // (whatever of the pass was enqueued -- all of it, or what came before a failed launch -- is waited for by the next user of the GI state)
    if (aside) { ctx->side_busy = true; const hipError_t re = hipEventRecord(ctx->ev_side_done, ctx->side); if (rs == DUST_OK && re != hipSuccess) rs = hip_fail(re, "hipEventRecord(ev_side_done)"); }
    if (rs != DUST_OK) return rs;
    if (calibrate) { HIP_TRY(hipEventRecord(p->side_cal.q1, st)); p->side_cal.state = 1; }
  }
  if (fp->passes & DUST_PASS_ACCUMULATE) {
    p->have_history = false;  // the plane now holds an N-frame mean, not the denoiser's history
    HIP_TRY(dust::launch_accumulate(a, st));
    p->accum_count += 1;
  }
  if (fp->passes & DUST_PASS_DENOISE) {
    const size_t px = size_t(p->width) * p->height;
    if (!p->hist_accum[0].p) {
      for (int k = 0; k < 2; ++k) {
        HIP_TRY(p->hist_accum[k].alloc(px * 16)); HIP_TRY(p->hist_geo[k].alloc(px * 16));
      }
      p->have_history = false;
    }
    dust::DenoiseArgs d{};
    d.illuminance = a.g.illuminance; d.denoised = a.g.denoised; d.normal = a.g.normal; d.depth = a.g.depth;
    d.motion = a.g.motion; d.voxel_id = a.g.voxel_id;
    const uint32_t in = p->hist_parity, out = in ^ 1u;
    // The radiance history lives in DUST_PLANE_ACCUM itself (it is what that plane shows while the denoiser runs): this frame
    // reads the plane's buffer and writes the spare one, and the two then trade places -- no copy. A caller-bound plane cannot
    // trade: it takes part as one side of an ordinary pair and receives a copy.
    const bool trade = p->bound[DUST_PLANE_ACCUM] == nullptr;
    d.hist_in_accum = static_cast<const float*>(trade ? p->planes[DUST_PLANE_ACCUM].p : p->hist_accum[in].p);
    d.hist_out_accum = static_cast<float*>(trade ? p->hist_accum[0].p : p->hist_accum[out].p);
    d.hist_in_geo = static_cast<const uint32_t*>(p->hist_geo[in].p);
    d.hist_out_geo = static_cast<uint32_t*>(p->hist_geo[out].p);
    if (trade != p->hist_traded) p->have_history = false;  // the plane was bound or released since the last frame: start over
    p->hist_traded = trade;
    d.cam = a.cam; d.prev = p->prev_cam;
    d.have_history = p->have_history ? 1u : 0u;
    d.width = p->width; d.height = p->height; d.frame_index = fp->frame_index;
    d.aspect = a.aspect;
    d.max_frames = float(std::max(1u, p->denoise.max_accumulated_frames));
    d.disocclusion = p->denoise.disocclusion_threshold;
    d.antilag_sigma = p->denoise.antilag_sigma_scale;
    d.antilag_power = p->denoise.antilag_power;
    d.max_radius = p->denoise.max_blur_radius;
    HIP_TRY(dust::launch_denoise(d, st));
    // DUST_PLANE_ACCUM shows the temporal accumulation (rgb + frame count) of the frame just filtered
    if (trade) { std::swap(p->planes[DUST_PLANE_ACCUM].p, p->hist_accum[0].p); std::swap(p->planes[DUST_PLANE_ACCUM].bytes, p->hist_accum[0].bytes); }
    else HIP_TRY(hipMemcpyAsync(p->plane(DUST_PLANE_ACCUM), d.hist_out_accum, px * 16, hipMemcpyDeviceToDevice, st));
    p->hist_parity = out;
    p->have_history = true;
    p->prev_cam = a.cam;
  }
  if (count) {
    HIP_TRY(hipMemcpyAsync(p->host_stats, p->stats.p, 8 * sizeof(dust::DevStats), hipMemcpyDeviceToHost, st));
    p->stats_valid = true;
  }
  return DUST_OK;
}

DustStatus dust_hip_render_frame(DustHipPipeline* p, const DustHipScene* s, const DustHipCamera* cam,
                                 const DustHipSky* sky, const DustHipFrameParams* fp_in) {
  return render_frame_impl(p, s, cam, sky, fp_in, FrameRole::Single, nullptr);
}
// The frames [0, n) can share one launch: fused primary + AO frames and nothing else, of distinct pipelines of the scene's context with one
// frame size, one row band and the same launch-shaping settings, and no surfel pass outstanding beside them.
static bool batchable(uint32_t n, DustHipPipeline* const* pipes, const DustHipScene* s, const DustHipFrameParams* fps) {
  if (n < 2 || n > dust::kMaxBatch) return false;
  const DustHipPipeline* p0 = pipes[0];
  if (p0->ctx != s->ctx || p0->ctx->side_busy) return false;
  const uint32_t want = DUST_PASS_PRIMARY | DUST_PASS_AMBIENT_OCCLUSION;
  for (uint32_t i = 0; i < n; ++i) {
    const DustHipPipeline* p = pipes[i];
    const Tuning& t = p->tune; const Tuning& t0 = p0->tune;
    if (p->ctx != p0->ctx || p->width != p0->width || p->height != p0->height) return false;
    if (fps[i].passes != want || fps[i].row_begin != fps[0].row_begin || fps[i].row_end != fps[0].row_end) return false;
    if (t.no_fuse || t.block != t0.block || t.blocks_per_cu != t0.blocks_per_cu || t.no_lds_boxes != t0.no_lds_boxes || t.wide_fused != t0.wide_fused ||
        t.debug != t0.debug || t.static_rounds != t0.static_rounds || t.reserve_blocks != t0.reserve_blocks || p->in_collective != p0->in_collective ||
        p->frames_in_flight != p0->frames_in_flight || (p->frames_in_flight > 1 && t.in_flight_slots != t0.in_flight_slots))
      return false;
    for (uint32_t j = 0; j < i; ++j)
      if (pipes[j] == p) return false;   // (the same pipeline twice: the second frame overwrites the first -- in sequence)
  }
  return true;
}
// what the host does to the scene before frame i (the reference's tlas_system pushes the moved entities every frame, tlas.rs:79-128)
static DustStatus apply_moves(DustHipScene* s, const DustHipFrameMoves& m) {
  if (!m.n) return DUST_OK;
  for (uint32_t j = 0; j < m.n; ++j) {
    DustStatus ms = dust_hip_scene_set_transform(s, m.instance_ids[j], m.obj_to_world + size_t(j) * 12, m.prev_obj_to_world ? m.prev_obj_to_world + size_t(j) * 16 : nullptr);
    if (ms != DUST_OK) return ms;
  }
  return dust_hip_scene_commit(s);
}
DustStatus dust_hip_render_frames(uint32_t n_frames, DustHipPipeline* const* pipelines, DustHipScene* s, const DustHipCamera* cameras,
                                  const DustHipSky* skies, const DustHipFrameParams* params, const DustHipFrameMoves* moves) {
  if (!pipelines || !s || !cameras || !skies || !params) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  if (!n_frames) return DUST_OK;
  return guarded([&]() -> DustStatus {
  // every frame's arguments are checked before the first one is enqueued (and before the scene is touched)
  std::vector<DustHipFrameParams> fps(n_frames);
  for (uint32_t i = 0; i < n_frames; ++i) {
    // (params[] is an array of THIS library's struct: every element must say so, or the stride is not ours)
    if (params[i].struct_size != sizeof(DustHipFrameParams)) return fail(DUST_ERR_INVALID_ARGUMENT, "dust_hip_render_frames: params[i].struct_size must be sizeof(DustHipFrameParams)");
    DustStatus cs = check_frame(pipelines[i], s, &cameras[i], &skies[i], &params[i], fps[i]);
    if (cs != DUST_OK) return cs;
    if (moves && moves[i].n) {
      if (!moves[i].instance_ids || !moves[i].obj_to_world) return fail(DUST_ERR_INVALID_ARGUMENT, "dust_hip_render_frames: moves[i] without instance ids or transforms");
      for (uint32_t j = 0; j < moves[i].n; ++j)
        if (moves[i].instance_ids[j] >= s->instances.size()) return fail(DUST_ERR_INVALID_ARGUMENT, "dust_hip_render_frames: moves[i] names an instance the scene does not have");
    }
  }
  for (uint32_t at = 0; at < n_frames;) {
    uint32_t n = std::min<uint32_t>(dust::kMaxBatch, n_frames - at);
    while (n >= 2 && !batchable(n, pipelines + at, s, fps.data() + at)) --n;   // the longest run from here that can share a launch
    if (n < 2) {   // one frame, or a frame that cannot share a launch with its successor: in sequence, the same results
      if (moves) { DustStatus ms = apply_moves(s, moves[at]); if (ms != DUST_OK) return ms; }
      DustStatus rs = render_frame_impl(pipelines[at], s, &cameras[at], &skies[at], &params[at], FrameRole::Single, nullptr);
      if (rs != DUST_OK) return rs;
      at += 1;
      continue;
    }
    // in frame order: the scene as frame i sees it (its moves committed: a scene image of its own in the ring), then frame i's descriptor against it;
    // the last frame's preparation launches them all. At most kMaxBatch commits lie between two launches -- the ring holds as many images --, so no
    // commit of this run lands on an image one of its own frames still waits to read.
    static_assert(dust::kMaxBatch <= DustHipScene::kImages, "a launch's frames must fit the scene's ring of images");
    BatchJoin join;
    join.n = n;
    for (uint32_t i = 0; i < n; ++i) {
      DustStatus rs = moves ? apply_moves(s, moves[at + i]) : DUST_OK;
      join.slot = i;
      if (rs == DUST_OK) rs = render_frame_impl(pipelines[at + i], s, &cameras[at + i], &skies[at + i], &params[at + i], i + 1 == n ? FrameRole::Lead : FrameRole::Follower, &join);
      if (rs != DUST_OK) {
        // (a HIP failure or a commit that could not grow the scene: the frames prepared so far are never launched -- their pipelines took a set of
        //  work counters for nothing, and no launch zeroed the other one: give it back, or their next launch would pull tiles from a set that an
        //  earlier launch has counted up)
        for (uint32_t j = 0; j < i; ++j) pipelines[at + j]->counter_parity[0] ^= 1u;
        return rs;
      }
    }
    at += n;
  }
  return DUST_OK;
  });
}

DustStatus dust_hip_pipeline_pass_stats(DustHipPipeline* p, uint32_t pass, DustHipPassStats* out) {
  if (!p || !out || pass > 5) return fail(DUST_ERR_INVALID_ARGUMENT, "bad pass index");
  std::memset(out, 0, sizeof(*out));
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));
  // pass 0: primary kernel; 1, 2: the two ray classes of the AO kernel (share its time); 3: final gather (+ surfel
  // commit); 4, 5: the two ray classes of the surfel pass (trace + apply kernels)
  const int kernel = pass == 0 ? 0 : (pass <= 2 ? 1 : (pass == 3 ? 2 : 3));
  if (kernel >= 0 && p->ctx->timing && p->ev_valid[kernel]) {
    float ms = 0.0f;
    const uint32_t slot = (p->ev_head[kernel] - 1u) % DustHipPipeline::kEvRing;
    HIP_TRY(hipEventElapsedTime(&ms, p->ev_ring[kernel][0][slot], p->ev_ring[kernel][1][slot]));
    out->ms = ms;
  }
  if (p->stats_valid) {
    const dust::DevStats& s = p->host_stats[pass];
    out->rays = s.rays; out->instances_tested = s.instances_tested; out->upper_descents = s.upper_descents;
    out->mid_descents = s.mid_descents; out->bricks_tested = s.bricks_tested; out->hits = s.hits;
  }
  return DUST_OK;
}
DustStatus dust_hip_pipeline_kernel_times(DustHipPipeline* p, int mark, float ms_sum[4], uint32_t launches[4]) {
  if (!p) return fail(DUST_ERR_INVALID_ARGUMENT, "null pipeline");
  if (mark && !ms_sum && !launches) {  // "from here": no wait, nothing read (a timed region starts right behind it)
    for (int k = 0; k < 4; ++k) p->ev_mark[k] = p->ev_head[k];
    return DUST_OK;
  }
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));
  for (int k = 0; k < 4; ++k) {
    double sum = 0.0;
    uint32_t n = 0;
    if (p->ctx->timing) {
      const uint32_t head = p->ev_head[k];
      uint32_t from = p->ev_mark[k];
      if (head - from > DustHipPipeline::kEvRing) from = head - DustHipPipeline::kEvRing;  // older pairs have been recorded over
      for (uint32_t i = from; i != head; ++i) {
        const uint32_t slot = i % DustHipPipeline::kEvRing;
        float ms = 0.0f;
        HIP_TRY(hipEventElapsedTime(&ms, p->ev_ring[k][0][slot], p->ev_ring[k][1][slot]));
        sum += ms; ++n;
      }
      if (mark) p->ev_mark[k] = head;
    }
    if (ms_sum) ms_sum[k] = float(sum);
    if (launches) launches[k] = n;
  }
  return DUST_OK;
}
DustStatus dust_hip_pipeline_plane_device_ptr(DustHipPipeline* p, DustHipPlane plane, void** ptr, size_t* bytes) {
  if (!p || int(plane) < 0 || plane >= DUST_PLANE_COUNT) return fail(DUST_ERR_INVALID_ARGUMENT, "bad plane");
  if (ptr) *ptr = p->plane(plane);
  if (bytes) *bytes = p->planes[plane].bytes;
  return DUST_OK;
}
DustStatus dust_hip_pipeline_bind_plane(DustHipPipeline* p, DustHipPlane plane, void* device_ptr, size_t bytes) {
  if (!p || int(plane) < 0 || plane >= DUST_PLANE_COUNT) return fail(DUST_ERR_INVALID_ARGUMENT, "bad plane");
  if (device_ptr && bytes < p->planes[plane].bytes) return fail(DUST_ERR_INVALID_ARGUMENT, "bound storage is smaller than the plane");
  if (device_ptr && (reinterpret_cast<uintptr_t>(device_ptr) & 15u)) return fail(DUST_ERR_INVALID_ARGUMENT, "bound storage must be 16-byte aligned");
  p->bound[plane] = device_ptr;  // launches enqueued from now on use it; the caller orders its own use of the memory on the stream
  return DUST_OK;
}
DustStatus dust_hip_pipeline_read_plane(DustHipPipeline* p, DustHipPlane plane, void* dst, size_t dst_bytes) {
  if (!p || !dst || int(plane) < 0 || plane >= DUST_PLANE_COUNT) return fail(DUST_ERR_INVALID_ARGUMENT, "bad plane");
  if (dst_bytes < p->planes[plane].bytes) return fail(DUST_ERR_INVALID_ARGUMENT, "destination too small");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));
  HIP_TRY(copy_wait(dst, p->plane(plane), p->planes[plane].bytes, hipMemcpyDeviceToHost, p->ctx->stream));
  ++p->ctx->sync_epoch;  // (copy_wait waited for the stream)
  return DUST_OK;
}
DustStatus dust_hip_pipeline_configure_gi(DustHipPipeline* p, uint32_t hash_capacity, uint32_t surfel_pool_size) {
  if (!p || hash_capacity < 4 || surfel_pool_size == 0) return fail(DUST_ERR_INVALID_ARGUMENT, "bad GI configuration");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));
  const size_t hash_bytes = (size_t(hash_capacity) + 2) * 12;  // probes run up to 2 past the end (spatial_hash.glsl:154-158)
  HIP_TRY(p->gi_hash.alloc(hash_bytes));
  HIP_TRY(hipMemsetAsync(p->gi_hash.p, 0, hash_bytes, p->ctx->stream));             // standard.rs:348-358 relies on a zeroed allocation
  HIP_TRY(p->gi_pool.alloc(size_t(surfel_pool_size) * 16));
  HIP_TRY(hipMemsetAsync(p->gi_pool.p, 0xFF, size_t(surfel_pool_size) * 16, p->ctx->stream));  // fill_buffer(u32::MAX), standard.rs:345-347
  HIP_TRY(p->gi_owner.alloc(size_t(surfel_pool_size) * 4));
  HIP_TRY(hipMemsetAsync(p->gi_owner.p, 0, size_t(surfel_pool_size) * 4, p->ctx->stream));
  HIP_TRY(p->gi_pixel_surfel.alloc(size_t(p->width) * p->height * 16));
  {
    const size_t tiles = size_t((p->width + 63) / 64) * ((p->height + 63) / 64 + 1);
    HIP_TRY(p->gi_order.alloc(tiles * 4096 * 4));
    HIP_TRY(p->gi_order_count.alloc(tiles * 4));
  }
  HIP_TRY(p->gi_requests.alloc(size_t(surfel_pool_size) * sizeof(dust::DevHashRequest)));
  HIP_TRY(p->gi_replacement.alloc(size_t(surfel_pool_size) * 16));
  HIP_TRY(p->gi_sun_payload.alloc(size_t(surfel_pool_size) * 16));
  for (DeviceBuffer* b : {&p->gi_sort_keys[0], &p->gi_sort_keys[1], &p->gi_sort_vals[0], &p->gi_sort_vals[1]}) HIP_TRY(b->alloc(size_t(surfel_pool_size) * 4));
  HIP_TRY(p->gi_sort_scratch.alloc(dust::radix_sort_scratch_bytes(surfel_pool_size)));
  HIP_TRY(p->gi_apply_alive.alloc(((size_t(surfel_pool_size) + 63) / 64 + 1) * 8 * 3));   // alive, cluster starts, run starts
  HIP_TRY(p->gi_apply_dead.alloc(size_t(surfel_pool_size)));
  HIP_TRY(hipMemsetAsync(p->gi_apply_dead.p, 0, size_t(surfel_pool_size), p->ctx->stream));
  // (the ray streams' buffers -- 48 B per pixel and more -- are made by the first frame that takes a stream path: ensure_stream_buffers)
  for (DeviceBuffer* b : {&p->gi_rays_fg, &p->gi_groups_fg, &p->gi_fg_hits, &p->gi_rays_sf, &p->gi_groups_sf, &p->gi_unbinned, &p->gi_hits_sf}) b->release();
  p->gi_capacity = hash_capacity;
  p->gi_pool_size = surfel_pool_size;
  p->gi_touched_rows = 0;  // the exchange buffers follow the pool size: dust_hip_pipeline_gi_exchange re-creates them
  p->gi_touched.release();
  p->gi_merged.release();
  return DUST_OK;
}
DustStatus dust_hip_pipeline_gi_exchange(DustHipPipeline* p, uint32_t padded_rows, DustHipGiExchange* out) {
  if (!p || !out || padded_rows < p->height) return fail(DUST_ERR_INVALID_ARGUMENT, "padded_rows must cover the frame");
  STRUCT_TRY(out, "DustHipGiExchange");
  HIP_TRY(hipSetDevice(p->ctx->device));
  if (!p->gi_hash.p) {
    DustStatus gs = dust_hip_pipeline_configure_gi(p, dust::kSpatialHashCapacity, dust::kSurfelPoolSize);
    if (gs != DUST_OK) return gs;
  }
  if (!p->gi_touched.p || p->gi_touched_rows != padded_rows) {
    HIP_TRY(sync_stream(p->ctx));
    HIP_TRY(p->gi_touched.alloc(size_t(padded_rows) * p->width * 4));
    HIP_TRY(hipMemsetAsync(p->gi_touched.p, 0, size_t(padded_rows) * p->width * 4, p->ctx->stream));
    HIP_TRY(p->gi_merged.alloc(size_t(p->gi_pool_size) * 16));
    HIP_TRY(hipMemsetAsync(p->gi_merged.p, 0, size_t(p->gi_pool_size) * 16, p->ctx->stream));
    p->gi_touched_rows = padded_rows;
  }
  out->pool_size = p->gi_pool_size;
  out->width = p->width;
  out->touched_rows = padded_rows;
  out->slot_owner = p->gi_owner.p;
  out->touched = p->gi_touched.p;
  out->merged = p->gi_merged.p;
  return DUST_OK;
}
static DustStatus gi_exchange_launch(DustHipPipeline* p, uint32_t row_begin, uint32_t row_end, uint32_t frame_index, bool import) {
  if (!p || !p->gi_touched.p) return fail(DUST_ERR_NOT_READY, "call dust_hip_pipeline_gi_exchange first");
  // an EMPTY range is fine: a rank whose band lies past the end of the frame still repeats the others' stamps and commits the
  // winners (import), and must contribute zeroes -- not last frame's all-reduced sum -- to the merge (export)
  if (row_begin > row_end || row_end > p->height) return fail(DUST_ERR_INVALID_ARGUMENT, "bad row range");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(join_side(p->ctx));
  dust::FrameArgs a{};
  a.width = p->width; a.height = p->height;
  a.inv_width = 1.0f / float(p->width); a.inv_height = 1.0f / float(p->height); a.aspect = float(p->width) / float(p->height);
  a.row_begin = row_begin; a.row_end = row_end;
  a.frame_index = frame_index;
  a.gi.hash = static_cast<uint32_t*>(p->gi_hash.p);
  a.gi.hash_capacity = p->gi_capacity;
  a.gi.pool = static_cast<dust::DevSurfel*>(p->gi_pool.p);
  a.gi.pool_size = p->gi_pool_size;
  a.gi.slot_owner = static_cast<uint32_t*>(p->gi_owner.p);
  a.gi.pixel_surfel = static_cast<dust::DevSurfel*>(p->gi_pixel_surfel.p);
  a.gi.touched = static_cast<uint32_t*>(p->gi_touched.p);
  a.gi.merged = static_cast<dust::DevSurfel*>(p->gi_merged.p);
  HIP_TRY(import ? dust::launch_gi_import(a, p->ctx->stream) : dust::launch_gi_export(a, p->ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_gi_export(DustHipPipeline* p, uint32_t row_begin, uint32_t row_end) {
  return gi_exchange_launch(p, row_begin, row_end, 0, false);
}
DustStatus dust_hip_gi_import(DustHipPipeline* p, uint32_t row_begin, uint32_t row_end, uint32_t frame_index) {
  return gi_exchange_launch(p, row_begin, row_end, frame_index, true);
}
DustStatus dust_hip_pipeline_read_gi(DustHipPipeline* p, uint32_t which, void* dst, size_t dst_bytes) {
  if (!p || !dst || which > 1 || !p->gi_hash.p) return fail(DUST_ERR_INVALID_ARGUMENT, "GI state not configured");
  const DeviceBuffer& b = which == 0 ? p->gi_hash : p->gi_pool;
  if (dst_bytes < b.bytes) return fail(DUST_ERR_INVALID_ARGUMENT, "destination too small");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(join_side(p->ctx));
  HIP_TRY(sync_stream(p->ctx));
  HIP_TRY(copy_wait(dst, b.p, b.bytes, hipMemcpyDeviceToHost, p->ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_pipeline_write_gi(DustHipPipeline* p, uint32_t which, const void* src, size_t src_bytes) {
  if (!p || !src || which > 1 || !p->gi_hash.p) return fail(DUST_ERR_INVALID_ARGUMENT, "GI state not configured");
  const DeviceBuffer& b = which == 0 ? p->gi_hash : p->gi_pool;
  if (src_bytes != b.bytes) return fail(DUST_ERR_INVALID_ARGUMENT, "saved GI state does not match the configured capacity / pool size");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(join_side(p->ctx));
  HIP_TRY(copy_wait(b.p, src, b.bytes, hipMemcpyHostToDevice, p->ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_tone_map(DustHipPipeline* p, const DustHipToneMapParams* tp) {
  if (!p || !tp) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  STRUCT_TRY(tp, "DustHipToneMapParams");
  if (tp->transfer_function > 8) return fail(DUST_ERR_INVALID_ARGUMENT, "transfer function must be 0..8");
  if (!(tp->max_log_luminance > tp->min_log_luminance)) return fail(DUST_ERR_INVALID_ARGUMENT, "empty luminance range");
  HIP_TRY(hipSetDevice(p->ctx->device));
  uint32_t* hist = static_cast<uint32_t*>(p->exposure.p);
  HIP_TRY(dust::launch_tone_map(static_cast<const uint16_t*>(p->plane(DUST_PLANE_DENOISED)),
                                static_cast<const uint32_t*>(p->plane(DUST_PLANE_ALBEDO)),
                                static_cast<uint16_t*>(p->plane(DUST_PLANE_OUTPUT)), p->width * p->height, hist,
                                reinterpret_cast<float*>(hist + 256), tp->min_log_luminance,
                                tp->max_log_luminance - tp->min_log_luminance, tp->time_coefficient, tp->color_space_conversion,
                                tp->transfer_function, p->ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_pipeline_exposure(DustHipPipeline* p, float* avg_luminance, const float* set_to) {
  if (!p) return fail(DUST_ERR_INVALID_ARGUMENT, "null pipeline");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));
  float* avg = reinterpret_cast<float*>(static_cast<uint32_t*>(p->exposure.p) + 256);
  if (set_to) HIP_TRY(copy_wait(avg, set_to, 4, hipMemcpyHostToDevice, p->ctx->stream));
  if (avg_luminance) HIP_TRY(copy_wait(avg_luminance, avg, 4, hipMemcpyDeviceToHost, p->ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_device_eval(DustHipContext* ctx, uint32_t fn, const uint32_t* in, uint32_t in_words, uint32_t* out,
                                uint32_t out_words, uint32_t n) {
  static const uint32_t kWords[15][2] = {{9, 3}, {9, 3}, {9, 3}, {3, 1}, {1, 3}, {4, 1}, {1, 3}, {4, 2}, {2, 4}, {4, 1}, {3, 4}, {6, 3}, {2, 2}, {1, 1}, {1, 2}};
  if (!ctx || !in || !out || fn >= 15) return fail(DUST_ERR_INVALID_ARGUMENT, "bad device function");
  if (in_words != kWords[fn][0] || out_words != kWords[fn][1]) return fail(DUST_ERR_INVALID_ARGUMENT, "row width does not match the function");
  if (n == 0) return DUST_OK;
  HIP_TRY(hipSetDevice(ctx->device));
  if (fn == 12) {  // the surfel pass's radix sort on caller-given (key, value) rows, all 32 key bits
    std::vector<uint32_t> k(n), v(n);
    for (uint32_t i = 0; i < n; ++i) { k[i] = in[size_t(i) * 2]; v[i] = in[size_t(i) * 2 + 1]; }
    DeviceBuffer ka, va, kb, vb, scratch;
    HIP_TRY(ka.upload(k.data(), size_t(n) * 4, ctx->stream)); HIP_TRY(va.upload(v.data(), size_t(n) * 4, ctx->stream));
    HIP_TRY(kb.alloc(size_t(n) * 4)); HIP_TRY(vb.alloc(size_t(n) * 4));
    HIP_TRY(scratch.alloc(dust::radix_sort_scratch_bytes(n)));
    bool in_b = false;
    HIP_TRY(dust::radix_sort_pairs(scratch.p, static_cast<uint32_t*>(ka.p), static_cast<uint32_t*>(va.p), static_cast<uint32_t*>(kb.p),
                                   static_cast<uint32_t*>(vb.p), n, in_words == 2 ? 32u : 32u, &in_b, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    HIP_TRY(copy_wait(k.data(), in_b ? kb.p : ka.p, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(copy_wait(v.data(), in_b ? vb.p : va.p, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
    for (uint32_t i = 0; i < n; ++i) { out[size_t(i) * 2] = k[i]; out[size_t(i) * 2 + 1] = v[i]; }
    return DUST_OK;
  }
  if (fn == 13 || fn == 14) {  // the cost-ordered hand-out's sorter (k_tile_order) on caller-given tile costs: rows in = cycles, rows out = tile order
    // fn 14: with cost-balanced bands; out rows are then {order, cut}: the kRegions + 1 cuts in the second word of the first rows (n >= 9)
    if (fn == 14 && (out_words != 2 || n < dust::kRegions + 1)) return fail(DUST_ERR_INVALID_ARGUMENT, "fn 14 wants 2 output words and at least 9 rows");
    DeviceBuffer cost, order, cuts;
    HIP_TRY(cost.upload(in, size_t(n) * 4, ctx->stream));
    HIP_TRY(order.alloc(size_t(n) * 4));
    HIP_TRY(cuts.alloc(size_t(dust::kRegions + 1) * 4));
    HIP_TRY(hipMemsetAsync(order.p, 0xFF, size_t(n) * 4, ctx->stream));
    HIP_TRY(dust::launch_tile_order(static_cast<const uint32_t*>(cost.p), static_cast<uint32_t*>(order.p), fn == 14 ? static_cast<uint32_t*>(cuts.p) : nullptr, false,
                                    n, (n + dust::kRegions - 1) / dust::kRegions, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (fn == 13) { HIP_TRY(copy_wait(out, order.p, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream)); return DUST_OK; }
    std::vector<uint32_t> o(n), c(dust::kRegions + 1);
    HIP_TRY(copy_wait(o.data(), order.p, size_t(n) * 4, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(copy_wait(c.data(), cuts.p, c.size() * 4, hipMemcpyDeviceToHost, ctx->stream));
    for (uint32_t i = 0; i < n; ++i) { out[size_t(i) * 2] = o[i]; out[size_t(i) * 2 + 1] = i < c.size() ? c[i] : 0u; }
    return DUST_OK;
  }
  DeviceBuffer din, dout;
  HIP_TRY(din.upload(in, size_t(n) * in_words * 4, ctx->stream));
  HIP_TRY(dout.alloc(size_t(n) * out_words * 4));
  HIP_TRY(hipMemsetAsync(dout.p, 0, size_t(n) * out_words * 4, ctx->stream));
  HIP_TRY(dust::launch_device_eval(fn, static_cast<const uint32_t*>(din.p), in_words, static_cast<uint32_t*>(dout.p), out_words, n, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  HIP_TRY(copy_wait(out, dout.p, size_t(n) * out_words * 4, hipMemcpyDeviceToHost, ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_pipeline_tile_costs(DustHipPipeline* p, uint32_t pass_kind, uint32_t* cycles, uint32_t capacity, uint32_t* tiles_x, uint32_t* tiles_y) {
  if (!p || pass_kind > 3) return fail(DUST_ERR_INVALID_ARGUMENT, "bad pass kind");
  const DustHipPipeline::TileHistory& h = p->tile_history[pass_kind];
  if (tiles_x) *tiles_x = h.measured ? h.tiles_x : 0;
  if (tiles_y) *tiles_y = h.measured ? h.tiles_y : 0;
  if (!cycles || !h.measured) return DUST_OK;
  if (capacity < h.tiles_x * h.tiles_y) return fail(DUST_ERR_INVALID_ARGUMENT, "destination too small");
  HIP_TRY(hipSetDevice(p->ctx->device));
  HIP_TRY(sync_stream(p->ctx));
  HIP_TRY(copy_wait(cycles, h.cost.p, size_t(h.tiles_x) * h.tiles_y * 4, hipMemcpyDeviceToHost, p->ctx->stream));
  return DUST_OK;
}
DustStatus dust_hip_pipeline_set_denoiser(DustHipPipeline* p, const DustHipDenoiseParams* dp) {
  if (!p || !dp) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  STRUCT_TRY(dp, "DustHipDenoiseParams");
  if (dp->max_accumulated_frames == 0 || !(dp->disocclusion_threshold > 0.0f) || !(dp->antilag_sigma_scale >= 0.0f) ||
      !(dp->antilag_power >= 0.0f && dp->antilag_power <= 1.0f) || !(dp->max_blur_radius >= 0.0f && dp->max_blur_radius <= 64.0f))
    return fail(DUST_ERR_INVALID_ARGUMENT, "denoiser settings out of range");
  p->denoise = *dp;
  p->denoise.struct_size = sizeof(DustHipDenoiseParams);
  return DUST_OK;
}
DustStatus dust_hip_pipeline_restart_denoiser(DustHipPipeline* p) {
  if (!p) return fail(DUST_ERR_INVALID_ARGUMENT, "null pipeline");
  p->have_history = false;  // DenoiserEvent::Restart (nrd.rs:749-755): the next frame starts a new accumulation
  return DUST_OK;
}
DustStatus dust_hip_pipeline_set_frames_in_flight(DustHipPipeline* p, uint32_t n) {
  if (!p) return fail(DUST_ERR_INVALID_ARGUMENT, "null pipeline");
  if (n < 1 || n > 16) return fail(DUST_ERR_INVALID_ARGUMENT, "frames in flight: 1..16");
  p->frames_in_flight = n;
  return DUST_OK;
}
DustStatus dust_hip_pipeline_configure(DustHipPipeline* p, const DustHipPipelineConfig* cfg) {
  if (!p || !cfg) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  STRUCT_TRY(cfg, "DustHipPipelineConfig");
  if (cfg->gi_path > DUST_GI_PATH_STREAMS || cfg->side_stream > DUST_SIDE_STREAM_OFF || cfg->in_flight_slots > DUST_IN_FLIGHT_ALL ||
      cfg->frames_in_flight > 16 || (cfg->side_share != 0 && (cfg->side_share < 5 || cfg->side_share > 90)))
    return fail(DUST_ERR_INVALID_ARGUMENT, "DustHipPipelineConfig: a field is out of range");
  Tuning& t = p->tune;
  t.reserve_blocks = cfg->reserve_blocks == DUST_RESERVE_AUTO ? DUST_RESERVE_AUTO : (cfg->reserve_blocks & ~7u);  // whole rounds over the 8 XCDs
  t.gi_path = cfg->gi_path;
  t.no_side_stream = cfg->side_stream == DUST_SIDE_STREAM_OFF;
  t.side_share = cfg->side_share;
  t.in_flight_slots = cfg->in_flight_slots;
  if (cfg->frames_in_flight) p->frames_in_flight = cfg->frames_in_flight;
  return DUST_OK;
}
DustStatus dust_hip_pipeline_get_config(const DustHipPipeline* p, DustHipPipelineConfig* out) {
  if (!p || !out) return fail(DUST_ERR_INVALID_ARGUMENT, "null argument");
  STRUCT_TRY(out, "DustHipPipelineConfig");
  const Tuning& t = p->tune;
  out->reserve_blocks = t.reserve_blocks;
  out->gi_path = t.gi_path;
  out->side_stream = t.no_side_stream ? DUST_SIDE_STREAM_OFF : DUST_SIDE_STREAM_AUTO;
  out->side_share = t.side_share;
  out->frames_in_flight = p->frames_in_flight;
  out->in_flight_slots = t.in_flight_slots;
  return DUST_OK;
}
DustStatus dust_hip_pipeline_clear(DustHipPipeline* p) {
  if (!p) return fail(DUST_ERR_INVALID_ARGUMENT, "null pipeline");
  HIP_TRY(hipSetDevice(p->ctx->device));
  for (int i = 0; i < DUST_PLANE_COUNT; ++i) HIP_TRY(hipMemsetAsync(p->plane(i), 0, p->planes[i].bytes, p->ctx->stream));
  p->have_history = false;
  p->accum_count = 0;
  return DUST_OK;
}

}  // extern "C"

// ---- what comm.hip needs of the handles (capi_internal.hpp)
namespace dust_internal {
DustStatus set_error(DustStatus status, const std::string& message) { return fail(status, message); }
hipStream_t context_stream(DustHipContext* c) { return c->stream; }
int context_device(DustHipContext* c) { return c->device; }
void context_retain(DustHipContext* c) { retain(c); }
void context_release(DustHipContext* c) { release(c); }
DustHipContext* pipeline_context(DustHipPipeline* p) { return p->ctx; }
void pipeline_size(DustHipPipeline* p, uint32_t* width, uint32_t* height) { *width = p->width; *height = p->height; }
DustStatus gi_exchange_view(DustHipPipeline* p, uint32_t padded_rows, DustHipGiExchange* out) {
  if (!p->gi_touched.p || p->gi_touched_rows != padded_rows)
    return fail(DUST_ERR_NOT_READY, "the GI exchange buffers were not prepared for this world x band_rows: call dust_hip_pipeline_gi_exchange(p, world * band_rows) "
                                    "BEFORE the frame's final gather (a later call would re-create them and drop the frame's stamps)");
  out->pool_size = p->gi_pool_size;
  out->width = p->width;
  out->touched_rows = padded_rows;
  out->slot_owner = p->gi_owner.p;
  out->touched = p->gi_touched.p;
  out->merged = p->gi_merged.p;
  return DUST_OK;
}
void pipeline_note_collective(DustHipPipeline* p) { p->in_collective = true; }
DustStatus surfel_stage_view(DustHipPipeline* p, uint32_t world, SurfelStage* out) {
  if (!p->sf_shard.pending || !p->gi_stage_req.p) return fail(DUST_ERR_NOT_READY, "no sharded surfel trace is pending on this pipeline (dust_hip_render_frame with surfel_world >= 1 first)");
  if (world != 0 && p->sf_shard.world != world) return fail(DUST_ERR_INVALID_ARGUMENT, "the pending surfel trace was sharded for another world size than the communicator's");
  out->req = p->gi_stage_req.p; out->repl = p->gi_stage_repl.p; out->sun = p->gi_stage_sun.p;
  out->slots_per_rank = p->sf_shard.slots_per_rank; out->rank = p->sf_shard.rank; out->pool_size = p->gi_pool_size;
  return DUST_OK;
}
// the second half of a sharded surfel pass, on the context's stream behind the all-gather: records to their surfels + the trace's hash stamps, ordered apply
DustStatus surfel_finish(DustHipPipeline* p, uint32_t frame_index) {
  HIP_TRY(hipSetDevice(p->ctx->device));
  dust::FrameArgs a{};
  a.frame_index = frame_index;
  a.gi.hash = static_cast<uint32_t*>(p->gi_hash.p);
  a.gi.hash_capacity = p->gi_capacity;
  a.gi.pool = static_cast<dust::DevSurfel*>(p->gi_pool.p);
  a.gi.pool_size = p->gi_pool_size;
  a.gi.requests = static_cast<dust::DevHashRequest*>(p->gi_requests.p);
  a.gi.replacement = static_cast<dust::DevSurfel*>(p->gi_replacement.p);
  a.gi.sun_payload = static_cast<float*>(p->gi_sun_payload.p);
  a.gi.perm = p->sf_shard.perm;
  a.sf_stage_req = static_cast<dust::DevHashRequest*>(p->gi_stage_req.p);
  a.sf_stage_repl = static_cast<dust::DevSurfel*>(p->gi_stage_repl.p);
  a.sf_stage_sun = static_cast<float*>(p->gi_stage_sun.p);
  const hipStream_t st = p->ctx->stream;
  HIP_TRY(dust::launch_surfel_unstage(a, st));
  p->sf_shard.pending = false;
  return apply_ordered(p, a, st);
}
void context_add_stream(DustHipContext* c, hipStream_t s) { c->extra_streams.push_back(s); }
void context_remove_stream(DustHipContext* c, hipStream_t s) {
  c->extra_streams.erase(std::remove(c->extra_streams.begin(), c->extra_streams.end(), s), c->extra_streams.end());
}
}  // namespace dust_internal
