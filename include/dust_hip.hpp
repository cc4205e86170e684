// dust_hip.hpp -- header-only C++ mirror of the reference's plugin/operator surface over the C ABI.
//
// Same names, argument meaning and error behaviour as the Rust items they stand for, so that host code
// (and the parity tests) read like the reference's own:
//   dust::Tree                 dust_vdb::Tree<hierarchy!(..)>       crates/vdb/src/tree.rs:7-124
//   dust::Tree::Accessor       dust_vdb::Accessor                   crates/vdb/src/accessor.rs:5-57
//   dust::VoxLoader::load      VoxLoader::load                      crates/vox/src/loader.rs:322-415
//   dust::VoxGeometry          VoxGeometry (+ PaletteMaterial)      crates/vox/src/geometry.rs:30-179, material.rs:9-120
//   dust::PinholeProjection    PinholeProjection                    crates/render/src/projection.rs:3-29
//   dust::Sunlight             Sunlight + bake()                    crates/render/src/pipeline/sky.rs:6-23,90-132
//   dust::ReblurSettings       ReblurSettings (the knobs this filter has)   crates/render/src/pipeline/nrd.rs:693-785
//   dust::RenderContext        what RenderPlugin::build sets up     crates/render/src/lib.rs:58-134
//   dust::Scene                TLASStore + instance vec             crates/render/src/accel_struct/tlas.rs:28-180
//   dust::StandardPipeline     StandardPipeline + GBuffer           crates/render/src/pipeline/standard.rs:51-60,222-240,881-917
// Rust `Option`/`Result` become std::optional / dust::Error exceptions; nothing throws across the C ABI itself.
#pragma once
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "dust_hip.h"

namespace dust {

struct Error : std::runtime_error {
  DustStatus status;
  Error(DustStatus s, const std::string& m) : std::runtime_error(m), status(s) {}
};
inline void check(DustStatus s) {
  if (s != DUST_OK) throw Error(s, dust_hip_last_error());
}

using UVec3 = std::array<uint32_t, 3>;

// ----------------------------------------------------------------------------- dust_vdb
class Tree {
 public:
  // hierarchy!(4,2,2) -> Tree({4,2,2})
  explicit Tree(std::vector<uint32_t> fanout_log2) { check(dust_vdb_tree_create(fanout_log2.data(), uint32_t(fanout_log2.size()), &h_)); }
  ~Tree() { dust_vdb_tree_destroy(h_); }
  Tree(const Tree&) = delete;
  Tree& operator=(const Tree&) = delete;

  void set_value(UVec3 c, std::optional<bool> v) { check(dust_vdb_tree_set(h_, c[0], c[1], c[2], v ? (*v ? 1 : 0) : -1)); }
  std::optional<bool> get_value(UVec3 c) const {
    int32_t v;
    check(dust_vdb_tree_get(h_, c[0], c[1], c[2], &v));
    return v < 0 ? std::nullopt : std::optional<bool>(v != 0);
  }
  std::vector<UVec3> iter() const {
    size_t n = 0;
    check(dust_vdb_tree_iter(h_, nullptr, 0, &n));
    std::vector<UVec3> out(n);
    check(dust_vdb_tree_iter(h_, n ? out[0].data() : nullptr, n, &n));
    return out;
  }
  struct Leaf { UVec3 origin; uint64_t occupancy; uint32_t material_ptr; };
  std::vector<Leaf> iter_leaf() const {
    size_t n = 0;
    check(dust_vdb_tree_iter_leaf(h_, nullptr, nullptr, nullptr, 0, &n));
    std::vector<uint32_t> xyz(n * 3 + 1), mp(n + 1);
    std::vector<uint64_t> occ(n + 1);
    check(dust_vdb_tree_iter_leaf(h_, xyz.data(), occ.data(), mp.data(), n, &n));
    std::vector<Leaf> out(n);
    for (size_t i = 0; i < n; ++i) out[i] = Leaf{{xyz[i * 3], xyz[i * 3 + 1], xyz[i * 3 + 2]}, occ[i], mp[i]};
    return out;
  }
  uint32_t meta_mask() const { uint32_t m; check(dust_vdb_tree_meta(h_, &m, nullptr)); return m; }
  uint32_t root_level() const { uint32_t l; check(dust_vdb_tree_meta(h_, nullptr, &l)); return l; }

  class Accessor {
   public:
    explicit Accessor(const Tree& t) { check(dust_vdb_accessor_create(t.h_, &h_)); }
    ~Accessor() { dust_vdb_accessor_destroy(h_); }
    Accessor(const Accessor&) = delete;
    std::optional<bool> get(UVec3 c) {
      int32_t v;
      check(dust_vdb_accessor_get(h_, c[0], c[1], c[2], &v));
      return v < 0 ? std::nullopt : std::optional<bool>(v != 0);
    }
   private:
    DustVdbAccessor* h_ = nullptr;
  };
  Accessor accessor() const { return Accessor(*this); }

 private:
  DustVdbTree* h_ = nullptr;
};

// ----------------------------------------------------------------------------- dust_render context
class RenderContext {
 public:
  explicit RenderContext(int device = -1, bool timing = false, void* stream = nullptr) {
    DustHipConfig cfg{};
    cfg.struct_size = sizeof(cfg);
    cfg.device = device;
    cfg.stream = stream;
    cfg.flags = timing ? DUST_HIP_CONTEXT_TIMING : 0;
    check(dust_hip_context_create(&cfg, &h_));
  }
  ~RenderContext() { dust_hip_context_destroy(h_); }
  RenderContext(const RenderContext&) = delete;
  void sync() { check(dust_hip_sync(h_)); }
  DustHipContext* raw() const { return h_; }
 private:
  DustHipContext* h_ = nullptr;
};

// ----------------------------------------------------------------------------- dust_vox
// VoxGeometry + PaletteMaterial of one model, resident on the device.
class VoxGeometry {
 public:
  VoxGeometry(RenderContext& ctx, const DustHipBlock* blocks, uint32_t n_blocks, const uint8_t* materials,
              uint64_t n_materials, const uint8_t* palette_rgba, uint32_t tree_extent_log2 = 8)
      : num_blocks(n_blocks) {
    check(dust_hip_model_create(ctx.raw(), blocks, n_blocks, materials, n_materials, palette_rgba, tree_extent_log2, &h_));
  }
  ~VoxGeometry() { dust_hip_model_destroy(h_); }
  VoxGeometry(const VoxGeometry&) = delete;
  DustHipModel* raw() const { return h_; }
  // VoxGeometry::set / get (geometry.rs:180-185) -- on the device copy, batched; Some(true) carries the palette index the
  // voxel is shaded with (the reference's call edits the CPU tree only and has no material to give). Scenes that instance the
  // geometry must commit() again before they render.
  void set(const std::vector<UVec3>& coords, const std::vector<std::optional<uint8_t>>& palette_index) {
    std::vector<uint32_t> xyz;
    std::vector<int32_t> val;
    for (size_t i = 0; i < coords.size(); ++i) {
      xyz.insert(xyz.end(), coords[i].begin(), coords[i].end());
      val.push_back(palette_index[i] ? int32_t(*palette_index[i]) : -1);
    }
    check(dust_hip_model_set_voxels(h_, xyz.data(), val.data(), uint32_t(val.size())));
    uint64_t nm = 0;
    check(dust_hip_model_info(h_, &num_blocks, &nm));
  }
  void set(UVec3 c, std::optional<uint8_t> palette_index) { set(std::vector<UVec3>{c}, {palette_index}); }
  std::optional<uint8_t> get(UVec3 c) {
    int32_t v = -1;
    check(dust_hip_model_get_voxels(h_, c.data(), &v, 1));
    return v < 0 ? std::nullopt : std::optional<uint8_t>(uint8_t(v));
  }
  uint32_t num_blocks;
 private:
  DustHipModel* h_ = nullptr;
};

// What VoxLoader::load returns: models (geometry + material) and the entities that instance them.
struct VoxScene {
  std::vector<std::unique_ptr<VoxGeometry>> geometries;   // indexed by model id; null for unused models
  std::vector<DustVoxInstance> instances;                 // VoxBundle transform + model id
  std::array<uint8_t, 1024> palette{};
};

// PngLoader (rhyolite_bevy/src/loaders/png.rs:70-200): a PNG / APNG as a sliced image array, one layer per frame
struct SlicedImageArray {
  DustPngInfo info{};
  std::vector<uint8_t> texels;  // layers x height x width x channels x bytes_per_channel
};
struct PngLoader {
  static SlicedImageArray load(const uint8_t* bytes, size_t n) {
    SlicedImageArray a;
    uint8_t* p = nullptr;
    check(dust_png_load_array(bytes, n, &a.info, &p));
    const size_t total = size_t(a.info.layers) * a.info.height * a.info.width * a.info.channels * a.info.bytes_per_channel;
    a.texels.assign(p, p + total);
    dust_vox_free(p);
    return a;
  }
};

class VoxLoader {
 public:
  explicit VoxLoader(RenderContext& ctx) : ctx_(ctx) {}
  static std::vector<std::string> extensions() { return {"vox"}; }  // loader.rs:417-419
  // throws dust::Error{DUST_ERR_PARSE | DUST_ERR_UNSUPPORTED} like VoxLoadingError / unimplemented!()
  // frame: the animation frame MagicaVoxel keyframes (multi-frame nTRN, multi-model nSHP) are instantiated at; the reference
  // stops at unimplemented!() for those files (loader.rs:103-105,149-151)
  VoxScene load(const uint8_t* bytes, size_t n, uint32_t frame = 0) {
    DustVoxScene* s = nullptr;
    check(dust_vox_load_frame(bytes, n, frame, &s));
    struct Guard { DustVoxScene* s; ~Guard() { dust_vox_scene_destroy(s); } } g{s};
    VoxScene out;
    uint32_t nm = 0, ni = 0;
    check(dust_vox_scene_counts(s, &nm, &ni));
    const uint8_t* pal = nullptr;
    check(dust_vox_scene_palette(s, &pal));
    std::memcpy(out.palette.data(), pal, 1024);
    out.geometries.resize(nm);
    for (uint32_t m = 0; m < nm; ++m) {
      DustVoxModelInfo info{};
      check(dust_vox_scene_model_info(s, m, &info));
      if (!info.used) continue;
      const DustHipBlock* blocks = nullptr;
      const uint8_t* mats = nullptr;
      check(dust_vox_scene_model_data(s, m, &blocks, &mats));
      out.geometries[m] = std::make_unique<VoxGeometry>(ctx_, blocks, info.n_blocks, mats, info.n_materials, pal);
    }
    out.instances.resize(ni);
    if (ni) check(dust_vox_scene_instances(s, out.instances.data(), ni));
    return out;
  }
 private:
  RenderContext& ctx_;
};

// ----------------------------------------------------------------------------- scene (TLAS)
class Scene {
 public:
  explicit Scene(RenderContext& ctx) { check(dust_hip_scene_create(ctx.raw(), &h_)); }
  ~Scene() { dust_hip_scene_destroy(h_); }
  Scene(const Scene&) = delete;
  // spawn(VoxBundle{transform, geometry, material}) -> gl_InstanceID
  uint32_t spawn(const VoxGeometry& g, const float obj_to_world_3x4[12], const float* prev_mat4 = nullptr) {
    uint32_t id = 0;
    check(dust_hip_scene_add_instance(h_, g.raw(), obj_to_world_3x4, prev_mat4, &id));
    return id;
  }
  void spawn_scene(const VoxScene& s) {
    for (const auto& i : s.instances) spawn(*s.geometries[i.model], i.obj_to_world);
  }
  void set_transform(uint32_t id, const float obj_to_world_3x4[12], const float* prev_mat4 = nullptr) {
    check(dust_hip_scene_set_transform(h_, id, obj_to_world_3x4, prev_mat4));
  }
  void commit() { check(dust_hip_scene_commit(h_)); }
  DustHipScene* raw() const { return h_; }
 private:
  DustHipScene* h_ = nullptr;
};

// ----------------------------------------------------------------------------- camera
struct PinholeProjection {  // projection.rs:3-29
  float fov = 0.78539816339744830962f;
  float near = 0.1f;
  float far = 10000.0f;
};
// rotation columns of a camera at `eye` looking at `target` (Transform::looking_at, -z forward)
inline std::array<float, 9> look_at_rotation(const double eye[3], const double target[3], const double up_[3]) {
  auto norm = [](double v[3]) { double l = std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); for (int i = 0; i < 3; ++i) v[i] /= l; };
  double back[3] = {eye[0] - target[0], eye[1] - target[1], eye[2] - target[2]};
  norm(back);
  double right[3] = {up_[1] * back[2] - up_[2] * back[1], up_[2] * back[0] - up_[0] * back[2], up_[0] * back[1] - up_[1] * back[0]};
  norm(right);
  double up[3] = {back[1] * right[2] - back[2] * right[1], back[2] * right[0] - back[0] * right[2], back[0] * right[1] - back[1] * right[0]};
  return {float(right[0]), float(right[1]), float(right[2]), float(up[0]), float(up[1]), float(up[2]),
          float(back[0]), float(back[1]), float(back[2])};
}
inline DustHipCamera make_camera(const float eye[3], const std::array<float, 9>& rot_cols, const PinholeProjection& p) {
  DustHipCamera c{};
  std::memcpy(c.view_col0, &rot_cols[0], 12);
  std::memcpy(c.view_col1, &rot_cols[3], 12);
  std::memcpy(c.view_col2, &rot_cols[6], 12);
  std::memcpy(c.position, eye, 12);
  c.tan_half_fov = std::tan(p.fov / 2.0f);  // standard.rs:298
  c.far_ = p.far;
  c.near_ = p.near;
  return c;
}

// ----------------------------------------------------------------------------- Sunlight
// The Hosek-Wilkie tables Sunlight::bake reads, as the bytes of the reference's dataset.bin / datasetSolar.bin (sky.rs:34-63)
class SkyDataset {
 public:
  SkyDataset(const uint8_t* dataset_bin, size_t n, const uint8_t* dataset_solar_bin, size_t m) { check(dust_sky_dataset_create(dataset_bin, n, dataset_solar_bin, m, &h_)); }
  ~SkyDataset() { dust_sky_dataset_destroy(h_); }
  SkyDataset(const SkyDataset&) = delete;
  const DustSkyDataset* raw() const { return h_; }
 private:
  DustSkyDataset* h_ = nullptr;
};
struct Sunlight {  // sky.rs:6-23
  float turbidity = 1.0f;
  std::array<float, 3> albedo{0.2f, 0.2f, 0.2f};
  std::array<float, 3> direction{0.0f, 0.80114365f, -0.5984721f};
  DustHipSky bake(const SkyDataset& d) const {  // sky.rs:90-132
    DustHipSky s{};
    check(dust_sky_bake(d.raw(), turbidity, albedo.data(), direction.data(), &s));
    return s;
  }
};
struct ReblurSettings {  // nrd.rs:768-785 + NRD defaults, as far as DUST_PASS_DENOISE has the knob
  uint32_t max_accumulated_frame_num = 30;
  float disocclusion_threshold = 0.01f;
  float luminance_sigma_scale = 2.0f, luminance_antilag_power = 0.8f;
  float blur_radius = 15.0f;
};

// ----------------------------------------------------------------------------- StandardPipeline
class StandardPipeline {
 public:
  static constexpr uint32_t PRIMARY_RAYTYPE = 0;            // standard.rs:223-226
  static constexpr uint32_t AMBIENT_OCCLUSION_RAYTYPE = 1;
  static constexpr uint32_t FINAL_GATHER_RAYTYPE = 2;
  static constexpr uint32_t SURFEL_RAYTYPE = 3;
  static constexpr uint32_t num_raytypes() { return 4; }

  StandardPipeline(RenderContext& ctx, uint32_t width, uint32_t height) : width_(width), height_(height) {
    check(dust_hip_pipeline_create(ctx.raw(), width, height, &h_));
  }
  ~StandardPipeline() { dust_hip_pipeline_destroy(h_); }
  StandardPipeline(const StandardPipeline&) = delete;

  void set_blue_noise(uint32_t texture, const uint8_t* texels, uint32_t layers) {
    check(dust_hip_pipeline_set_noise(h_, texture, texels, layers));
  }
  // StandardPipeline::render: returns false ("try next frame") where the reference returns None (standard.rs:254-266)
  bool render(const Scene& scene, const DustHipCamera& camera, const DustHipSky& sunlight_baked, uint32_t passes,
              uint32_t frame_index, uint32_t rand, uint32_t row_begin = 0, uint32_t row_end = 0) {
    DustHipFrameParams fp{};
    fp.struct_size = sizeof(fp);
    fp.passes = passes; fp.frame_index = frame_index; fp.rand = rand; fp.row_begin = row_begin; fp.row_end = row_end;
    const DustStatus s = dust_hip_render_frame(h_, scene.raw(), &camera, &sunlight_baked, &fp);
    if (s == DUST_ERR_NOT_READY) return false;
    check(s);
    return true;
  }
  // Frames in flight (rhyolite_bevy/src/lib.rs:58): n frames, frame i into pipelines[i], as n render() calls in that order; primary + AO frames of
  // distinct pipelines of one context share one persistent launch (dust_hip_render_frames). false where any of them is not ready.
  static bool render_frames(StandardPipeline* const* pipelines, uint32_t n, Scene& scene, const DustHipCamera* cameras, const DustHipSky* skies,
                            uint32_t passes, const uint32_t* frame_indices, const uint32_t* rands, uint32_t row_begin = 0, uint32_t row_end = 0,
                            const DustHipFrameMoves* moves = nullptr) {   // moves[i]: the entities that moved before frame i (tlas_system's per-frame push), or null
    std::vector<DustHipPipeline*> hs(n);
    std::vector<DustHipFrameParams> fps(n);
    for (uint32_t i = 0; i < n; ++i) {
      hs[i] = pipelines[i]->h_;
      fps[i] = DustHipFrameParams{};
      fps[i].struct_size = sizeof(DustHipFrameParams);
      fps[i].passes = passes; fps[i].frame_index = frame_indices[i]; fps[i].rand = rands[i]; fps[i].row_begin = row_begin; fps[i].row_end = row_end;
    }
    const DustStatus s = dust_hip_render_frames(n, hs.data(), scene.raw(), cameras, skies, fps.data(), moves);
    if (s == DUST_ERR_NOT_READY) return false;
    check(s);
    return true;
  }
  // NRDPipeline settings / DenoiserEvent::Restart for DUST_PASS_DENOISE
  void set_denoiser(const ReblurSettings& r) {
    DustHipDenoiseParams dp{sizeof(DustHipDenoiseParams), r.max_accumulated_frame_num, r.disocclusion_threshold, r.luminance_sigma_scale,
                            r.luminance_antilag_power, r.blur_radius};
    check(dust_hip_pipeline_set_denoiser(h_, &dp));
  }
  void restart_denoiser() { check(dust_hip_pipeline_restart_denoiser(h_)); }
  // the host's frames in flight (rhyolite_bevy/src/lib.rs:58): launches then share the device instead of queueing
  void set_frames_in_flight(uint32_t n) { check(dust_hip_pipeline_set_frames_in_flight(h_, n)); }
  // the plugin's settings for this pipeline (RenderPlugin, crates/render/src/lib.rs:35-56): read, change the fields of interest, write back
  DustHipPipelineConfig config() const {
    DustHipPipelineConfig c{};
    c.struct_size = sizeof c;
    check(dust_hip_pipeline_get_config(h_, &c));
    return c;
  }
  void configure(const DustHipPipelineConfig& c) { check(dust_hip_pipeline_configure(h_, &c)); }
  void bind_plane(DustHipPlane plane, void* device_ptr, size_t bytes) { check(dust_hip_pipeline_bind_plane(h_, plane, device_ptr, bytes)); }
  template <class T>
  std::vector<T> read_plane(DustHipPlane plane) {
    size_t bytes = 0;
    void* p = nullptr;
    check(dust_hip_pipeline_plane_device_ptr(h_, plane, &p, &bytes));
    std::vector<T> out(bytes / sizeof(T));
    check(dust_hip_pipeline_read_plane(h_, plane, out.data(), bytes));
    return out;
  }
  uint32_t width() const { return width_; }
  uint32_t height() const { return height_; }
  DustHipPipeline* raw() const { return h_; }
 private:
  DustHipPipeline* h_ = nullptr;
  uint32_t width_, height_;
};


// One rank's end of the multi-GPU partition: an RCCL communicator on the context's device -- or, on one device, the
// ranks of a loopback group. The reference's plugin selects ONE device (crates/render/src/lib.rs:58-134); a host that runs one
// process per GPU makes one of these next to its RenderContext.
class DeviceComm {
 public:
  static std::vector<uint8_t> unique_id() {  // on rank 0; carried to the other processes by the host
    std::vector<uint8_t> id(DUST_HIP_COMM_ID_BYTES);
    check(dust_hip_comm_unique_id(id.data()));
    return id;
  }
  DeviceComm(RenderContext& ctx, uint32_t rank, uint32_t world, const std::vector<uint8_t>& id) { check(dust_hip_comm_create(ctx.raw(), rank, world, id.data(), &h_)); }
  explicit DeviceComm(DustHipComm* adopted) : h_(adopted) {}
  // `world` ranks on one device: collectives complete when the group's last rank has made the call
  static std::vector<std::unique_ptr<DeviceComm>> loopback(RenderContext& ctx, uint32_t world) {
    std::vector<DustHipComm*> raw(world, nullptr);
    check(dust_hip_comm_create_local(ctx.raw(), world, raw.data()));
    std::vector<std::unique_ptr<DeviceComm>> out;
    for (DustHipComm* c : raw) out.emplace_back(new DeviceComm(c));
    return out;
  }
  ~DeviceComm() { dust_hip_comm_destroy(h_); }
  DeviceComm(const DeviceComm&) = delete;
  // rows [cuts[r], cuts[r+1]) of `plane` from every rank r to `root` (its own plane when dst is null)
  // (the library reads cuts[0..world]: a shorter vector would be an out-of-bounds host read)
  void check_cuts(const std::vector<uint32_t>& cuts) const {
    uint32_t world = 0;
    check(dust_hip_comm_info(h_, nullptr, &world, nullptr));
    if (cuts.size() != size_t(world) + 1) throw Error(DUST_ERR_INVALID_ARGUMENT, "band cuts: want world + 1 row indices");
  }
  uint64_t gather_bands(StandardPipeline& p, DustHipPlane plane, const std::vector<uint32_t>& cuts, uint32_t root, void* dst = nullptr, size_t dst_bytes = 0) {
    check_cuts(cuts);
    uint64_t ticket = 0;
    check(dust_hip_gather_bands(p.raw(), h_, plane, cuts.data(), root, dst, dst_bytes, &ticket));
    return ticket;
  }
  // several planes (bit i of the mask = DustHipPlane i) in one collective, each into the root pipeline's own plane
  uint64_t gather_planes(StandardPipeline& p, uint32_t plane_mask, const std::vector<uint32_t>& cuts, uint32_t root) {
    check_cuts(cuts);
    uint64_t ticket = 0;
    check(dust_hip_gather_planes(p.raw(), h_, plane_mask, cuts.data(), root, &ticket));
    return ticket;
  }
  void gi_exchange(StandardPipeline& p, uint32_t row_begin, uint32_t row_end, uint32_t band_rows, uint32_t frame_index) {
    check(dust_hip_gi_exchange_run(p.raw(), h_, row_begin, row_end, band_rows, frame_index));
  }
  void gi_surfel_exchange(StandardPipeline& p, uint32_t frame_index) { check(dust_hip_gi_surfel_exchange_run(p.raw(), h_, frame_index)); }
  void wait(uint64_t ticket = 0) { check(dust_hip_comm_wait(h_, ticket)); }
  void sync() { check(dust_hip_comm_sync(h_)); }
  DustHipComm* raw() const { return h_; }
 private:
  DustHipComm* h_ = nullptr;
};

}  // namespace dust
