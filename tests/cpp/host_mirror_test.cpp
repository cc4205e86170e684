// host_mirror_test.cpp -- drives include/dust_hip.hpp the way the reference's doctests and examples/castle.rs
// drive the Rust crates. Modes:
//   cpu                      the crates/vdb doctests + a .vox load, no GPU needed
//   gpu <file.vox> <w> <h> <noise5.bin> <sky.bin> <out_prefix>   offline frame like examples/castle.rs:105-236
//   commit <file.vox> <w> <h> <noise5.bin> <sky.bin> <frames>   teapot_move_system (castle.rs:287-291): one instance moves every
//                            frame, the TLAS is rebuilt inside the frame; prints the host time of set_transform + commit
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <algorithm>

#include "dust_hip.hpp"

#define EXPECT(c) do { if (!(c)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static std::vector<uint8_t> slurp(const char* path) {
  std::ifstream f(path, std::ios::binary);
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int cpu_tests() {
  using dust::Tree;
  {  // crates/vdb/src/tree.rs:15-25
    Tree tree({2, 2});
    tree.set_value({0, 4, 0}, true);
    tree.set_value({0, 2, 2}, false);
    EXPECT(tree.get_value({0, 4, 0}) == std::optional<bool>(true));
    EXPECT(tree.get_value({0, 3, 0}) == std::nullopt);
    EXPECT(tree.get_value({0, 2, 2}) == std::optional<bool>(false));
  }
  {  // crates/vdb/src/tree.rs:87-101
    Tree tree({4, 2});
    tree.set_value({0, 1, 2}, true);
    tree.set_value({63, 1, 3}, true);
    tree.set_value({63, 63, 63}, true);
    auto it = tree.iter();
    EXPECT(it.size() == 3);
    EXPECT((it[0] == dust::UVec3{0, 1, 2}) && (it[1] == dust::UVec3{63, 1, 3}) && (it[2] == dust::UVec3{63, 63, 63}));
  }
  {  // crates/vdb/src/accessor.rs:148-170
    Tree tree({2, 4, 2});
    EXPECT(tree.meta_mask() == 0b10100010u);
    const uint32_t a[3] = {0, 0, 0}, b[3] = {255, 255, 255};
    EXPECT(dust_vdb_lca_level(a, b, tree.meta_mask(), tree.root_level()) == 2);
    tree.set_value({17, 200, 3}, true);
    auto acc = tree.accessor();
    EXPECT(acc.get({17, 200, 3}) == std::optional<bool>(true));
  }
  {  // internal.rs:121-124: clearing is todo!() -> DUST_ERR_UNSUPPORTED, reported as an exception, never a crash
    Tree tree({4, 2, 2});
    bool threw = false;
    try { tree.set_value({1, 1, 1}, std::nullopt); } catch (const dust::Error& e) { threw = e.status == DUST_ERR_UNSUPPORTED; }
    EXPECT(threw);
  }
  {  // a malformed file is a ParseError (loader.rs:311-317)
    const uint8_t junk[16] = {'N', 'O', 'P', 'E'};
    DustVoxScene* s = nullptr;
    EXPECT(dust_vox_load(junk, sizeof(junk), &s) == DUST_ERR_PARSE);
  }
  {  // PngLoader: a file that is not a PNG is a parse error, not a crash
    bool threw = false;
    const uint8_t junk[16] = {'N', 'O', 'P', 'E'};
    try { (void)dust::PngLoader::load(junk, sizeof(junk)); } catch (const dust::Error& e) { threw = e.status == DUST_ERR_PARSE; }
    EXPECT(threw);
  }
  {  // Sunlight::bake needs the model's tables: sizes other than the reference's files are refused, not read out of bounds
    bool threw = false;
    std::vector<uint8_t> tiny(100);
    try { dust::SkyDataset d(tiny.data(), tiny.size(), tiny.data(), tiny.size()); } catch (const dust::Error& e) { threw = e.status == DUST_ERR_INVALID_ARGUMENT; }
    EXPECT(threw);
    std::vector<float> ds(1200 * 3, 0.5f), sol(1806 * 3, 100.0f);
    dust::SkyDataset d(reinterpret_cast<const uint8_t*>(ds.data()), ds.size() * 4, reinterpret_cast<const uint8_t*>(sol.data()), sol.size() * 4);
    const DustHipSky sky = dust::Sunlight{}.bake(d);
    EXPECT(sky.state[49] == 0.80114365f && sky.state[51] == 0.0f && sky.state[55] > 0.004f && sky.state[55] < 0.005f);
    dust::Sunlight below;
    below.direction = {0.0f, -0.5f, 0.8660254f};
    threw = false;
    try { (void)below.bake(d); } catch (const dust::Error&) { threw = true; }
    EXPECT(threw);
  }
  std::puts("cpu ok");
  return 0;
}

static int gpu_frame(int argc, char** argv) {
  if (argc < 8) { std::fprintf(stderr, "usage: gpu file.vox w h noise5.bin sky.bin out_prefix\n"); return 2; }
  const auto vox = slurp(argv[2]);
  const uint32_t w = uint32_t(std::atoi(argv[3])), h = uint32_t(std::atoi(argv[4]));
  const auto noise5 = slurp(argv[5]);
  const auto skyb = slurp(argv[6]);
  EXPECT(skyb.size() == sizeof(DustHipSky));
  DustHipSky sky;
  std::memcpy(&sky, skyb.data(), sizeof(sky));

  dust::RenderContext ctx(0);                       // add_plugins(dust_render::RenderPlugin)
  dust::VoxLoader loader(ctx);                      // add_plugins(dust_vox::VoxPlugin)
  dust::VoxScene assets = loader.load(vox.data(), vox.size());   // asset_server.load("castle.vox")
  dust::Scene scene(ctx);
  scene.spawn_scene(assets);                        // commands.spawn(SceneBundle{..})
  scene.commit();
  dust::StandardPipeline pipeline(ctx, w, h);
  pipeline.set_blue_noise(5, noise5.data(), uint32_t(noise5.size() / (128 * 128 * 4)));
  const double eye[3] = {122.0 * 0.15, 300.61 * 0.15, 54.45 * 0.15}, target[3] = {0, 0, 0}, up[3] = {0, 1, 0};
  const float eyef[3] = {float(eye[0]), float(eye[1]), float(eye[2])};
  const DustHipCamera cam = dust::make_camera(eyef, dust::look_at_rotation(eye, target, up), dust::PinholeProjection{});
  if (std::string(argv[1]) == "bands") {
    // the multi-GPU partition through the C ABI, 8 emulated ranks on this one device: every rank renders its row band into a pipeline
    // of its own, dust_hip_gather_bands (a loopback group: device copies) assembles the planes in rank 0's pipeline -- `pipeline`
    const uint32_t world = 8;
    auto comms = dust::DeviceComm::loopback(ctx, world);
    std::vector<uint32_t> cuts(world + 1);
    const uint32_t per = ((h + world - 1) / world + 7) / 8 * 8;
    for (uint32_t r = 0; r <= world; ++r) cuts[r] = std::min(h, r * per);
    std::vector<std::unique_ptr<dust::StandardPipeline>> ranks;
    for (uint32_t r = 1; r < world; ++r) {
      ranks.emplace_back(new dust::StandardPipeline(ctx, w, h));
      ranks.back()->set_blue_noise(5, noise5.data(), uint32_t(noise5.size() / (128 * 128 * 4)));
    }
    auto pipe_of = [&](uint32_t r) -> dust::StandardPipeline& { return r == 0 ? pipeline : *ranks[r - 1]; };
    for (uint32_t r = 0; r < world; ++r)
      if (cuts[r] < cuts[r + 1]) EXPECT(pipe_of(r).render(scene, cam, sky, DUST_PASS_PRIMARY | DUST_PASS_AMBIENT_OCCLUSION, 1, 4242, cuts[r], cuts[r + 1]));
    // depth on its own (dust_hip_gather_bands), the other two planes in ONE collective (dust_hip_gather_planes: what a root that
    // goes on to denoise the frame asks for)
    for (uint32_t r = 0; r < world; ++r) comms[r]->gather_bands(pipe_of(r), DUST_PLANE_DEPTH, cuts, 0);
    comms[0]->wait();
    for (uint32_t r = 0; r < world; ++r) comms[r]->gather_planes(pipe_of(r), (1u << DUST_PLANE_ILLUMINANCE) | (1u << DUST_PLANE_VOXEL_ID), cuts, 0);
    comms[0]->wait();
    comms[0]->sync();
  } else if (std::string(argv[1]) == "frames") {
    // frames in flight (rhyolite_bevy/src/lib.rs:58): three frames handed over in ONE call, one persistent launch (dust_hip_render_frames) --
    // frame 0 into `pipeline` (what is written out and compared), frames 1 and 2 into pipelines of their own with their own index and rand
    std::vector<std::unique_ptr<dust::StandardPipeline>> more;
    for (int i = 0; i < 2; ++i) {
      more.emplace_back(new dust::StandardPipeline(ctx, w, h));
      more.back()->set_blue_noise(5, noise5.data(), uint32_t(noise5.size() / (128 * 128 * 4)));
    }
    dust::StandardPipeline* pipes[3] = {&pipeline, more[0].get(), more[1].get()};
    const DustHipCamera cams[3] = {cam, cam, cam};
    const DustHipSky skies[3] = {sky, sky, sky};
    const uint32_t idx[3] = {1, 2, 3}, rnd[3] = {4242, 17, 99};
    EXPECT(dust::StandardPipeline::render_frames(pipes, 3, scene, cams, skies, DUST_PASS_PRIMARY | DUST_PASS_AMBIENT_OCCLUSION, idx, rnd));
    ctx.sync();
    const auto d1 = more[0]->read_plane<float>(DUST_PLANE_DEPTH), d0 = pipeline.read_plane<float>(DUST_PLANE_DEPTH);
    EXPECT(d1.size() == d0.size() && std::memcmp(d1.data(), d0.data(), d0.size() * 4) == 0);   // same camera: same depth, whatever the frame's noise
  } else {
    const bool ok = pipeline.render(scene, cam, sky, DUST_PASS_PRIMARY | DUST_PASS_AMBIENT_OCCLUSION, 1, 4242);
    EXPECT(ok);
  }
  ctx.sync();
  const auto depth = pipeline.read_plane<float>(DUST_PLANE_DEPTH);
  const auto ill = pipeline.read_plane<uint16_t>(DUST_PLANE_ILLUMINANCE);
  const auto vid = pipeline.read_plane<uint32_t>(DUST_PLANE_VOXEL_ID);
  std::ofstream(std::string(argv[7]) + ".depth", std::ios::binary).write(reinterpret_cast<const char*>(depth.data()), depth.size() * 4);
  std::ofstream(std::string(argv[7]) + ".ill", std::ios::binary).write(reinterpret_cast<const char*>(ill.data()), ill.size() * 2);
  std::ofstream(std::string(argv[7]) + ".vid", std::ios::binary).write(reinterpret_cast<const char*>(vid.data()), vid.size() * 4);
  std::puts("gpu ok");
  return 0;
}

// A frame loop with one moving instance: set_transform + commit + render per frame, nothing waits. What is timed is the host
// side of dust_hip_scene_set_transform + dust_hip_scene_commit (the reference rebuilds its TLAS inside the frame's command
// stream, tlas.rs:37-65; here: re-derive one instance record, one copy into pinned staging, one hipMemcpyAsync + event).
static int commit_loop(int argc, char** argv) {
  if (argc < 8) { std::fprintf(stderr, "usage: commit file.vox w h noise5.bin sky.bin frames\n"); return 2; }
  const auto vox = slurp(argv[2]);
  const uint32_t w = uint32_t(std::atoi(argv[3])), h = uint32_t(std::atoi(argv[4]));
  const auto noise5 = slurp(argv[5]);
  const auto skyb = slurp(argv[6]);
  const int frames = std::atoi(argv[7]);
  EXPECT(skyb.size() == sizeof(DustHipSky));
  DustHipSky sky;
  std::memcpy(&sky, skyb.data(), sizeof(sky));
  dust::RenderContext ctx(0);
  dust::VoxLoader loader(ctx);
  dust::VoxScene assets = loader.load(vox.data(), vox.size());
  dust::Scene scene(ctx);
  scene.spawn_scene(assets);
  scene.commit();
  dust::StandardPipeline pipeline(ctx, w, h);
  pipeline.set_blue_noise(5, noise5.data(), uint32_t(noise5.size() / (128 * 128 * 4)));
  const double eye[3] = {122.0, 300.61, 54.45}, target[3] = {0, 0, 0}, up[3] = {0, 1, 0};
  const float eyef[3] = {float(eye[0]), float(eye[1]), float(eye[2])};
  const DustHipCamera cam = dust::make_camera(eyef, dust::look_at_rotation(eye, target, up), dust::PinholeProjection{});
  float base[12];
  std::memcpy(base, assets.instances[0].obj_to_world, sizeof base);
  double commit_s = 0.0, frame_s = 0.0;
  const auto t_all = std::chrono::steady_clock::now();
  for (int f = 0; f < frames; ++f) {
    // frame f goes to the GPU (~0.25 ms of work) ...
    const auto t1 = std::chrono::steady_clock::now();
    EXPECT(pipeline.render(scene, cam, sky, DUST_PASS_PRIMARY | DUST_PASS_AMBIENT_OCCLUSION, uint32_t(1 + f), 4242u + uint32_t(f)));
    const auto t2 = std::chrono::steady_clock::now();
    // ... and while it runs, the host moves the instance for frame f + 1 and commits: this must not wait for the frame
    float m[12];
    std::memcpy(m, base, sizeof m);
    m[7] = base[7] + 20.0f * std::sin(0.1f * float(f));  // up and down, like the teapot
    const auto t3 = std::chrono::steady_clock::now();
    scene.set_transform(0, m);
    scene.commit();
    const auto t4 = std::chrono::steady_clock::now();
    commit_s += std::chrono::duration<double>(t4 - t3).count();
    frame_s += std::chrono::duration<double>(t2 - t1).count();
    if ((f & 1) == 1) ctx.sync();  // pace the loop (two frames in flight at most): an unpaced host outruns the GPU and then waits for queue space
  }
  ctx.sync();
  const double wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
  std::printf("commit loop: %d frames, %zu instances; set_transform + commit %.2f us per frame (host), render call %.2f us (host), %.3f ms per frame wall\n",
              frames, assets.instances.size(), 1e6 * commit_s / frames, 1e6 * frame_s / frames, 1e3 * wall / frames);
  return 0;
}

int main(int argc, char** argv) {
  try {
    if (argc >= 2 && (std::string(argv[1]) == "gpu" || std::string(argv[1]) == "bands" || std::string(argv[1]) == "frames")) return gpu_frame(argc, argv);
    if (argc >= 2 && std::string(argv[1]) == "commit") return commit_loop(argc, argv);
    return cpu_tests();
  } catch (const dust::Error& e) {
    std::fprintf(stderr, "dust::Error %d: %s\n", int(e.status), e.what());
    return 3;
  }
}
