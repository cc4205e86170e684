"""The BASELINE.json configurations other than the bench one, as parity-test cases (sized to run in seconds):
config 1 (teapot, 256x256, CPU) and config 5 (procedural 4096^3 deep tree, hierarchy (4,4,2,2))."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth


def teapot_desc(n=64):
    desc = P.SceneDesc.from_vox(synth.teapot_scene(n))
    # examples/castle.rs:287-291 at t = 0: the teapot floats at (0, 200, 0)
    m = desc.instances[0][1].reshape(3, 4).copy()
    m[:, 3] += np.array([0.0, 200.0, 0.0], np.float32)
    desc.instances[0] = (desc.instances[0][0], m.reshape(12))
    return desc


def test_config1_teapot_256_cpu():
    """BASELINE config 1: teapot stand-in, 256x256, one primary ray per pixel, CPU traversal only."""
    desc = teapot_desc(64)
    s = P.oracle_scene(desc)
    cam = P.camera_for((60.0, 250.0, 70.0), target=(0.0, 200.0, 0.0))
    w = h = 256
    a = P.render_oracle(s, cam, P.sky_state(), w, h, L.PASS_PRIMARY, mode=O.ORC_MODE_HIER)
    hit = np.isfinite(a.depth)
    assert 0.05 < hit.mean() < 0.9
    # the semantic definition (closest over all bricks) on a centre crop
    b = P.render_oracle(s, cam, P.sky_state(), w, h, L.PASS_PRIMARY, mode=O.ORC_MODE_BRUTE, rows=(96, 160))
    for name in ("albedo", "normal", "depth", "motion", "voxel_id", "denoised"):
        assert getattr(a, name)[96:160].tobytes() == getattr(b, name)[96:160].tobytes(), name


@pytest.mark.gpu
def test_config1_teapot_gpu_parity():
    desc = teapot_desc(96)
    ctx = api.Context(device=0)
    cam = P.camera_for((60.0, 250.0, 70.0), target=(0.0, 200.0, 0.0))
    noise5 = synth.stbn_unitvec3_cosine(layers=2)
    pipe = api.StandardPipeline(ctx, 256, 256)
    pipe.set_noise(5, noise5)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    pipe.render(P.hip_scene(ctx, desc), cam, sky, passes, frame_index=1, rand=99)
    g = P.render_oracle(P.oracle_scene(desc), cam, sky, 256, 256, passes, noise5[1], 99)
    P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(pipe)))


def deep_desc(occupancy):
    blocks, mats = synth.procedural_deep_blocks(occupancy=occupancy, sample=True)
    return blocks, mats, synth.make_palette(5)


def test_config5_deep_tree_oracle_modes_agree():
    blocks, mats, pal = deep_desc(3e-6)
    assert 1500 < len(blocks) < 6000
    s = O.Scene()
    s.add_model(blocks, mats, pal, extent=4096)
    s.add_instance(0, np.eye(3, 4, dtype=np.float32).reshape(12))
    s.commit()
    rng = np.random.default_rng(1)
    hits = 0
    for i in range(300):
        o = rng.uniform(-500, 4600, 3)
        tgt = blocks[int(rng.integers(0, len(blocks)))]
        d = np.array([tgt["x"], tgt["y"], tgt["z"]], np.float64) + rng.uniform(0, 4, 3) - o
        if i % 3 == 0:
            d /= np.linalg.norm(d)
        a = s.trace(O.ORC_MODE_BRUTE, 0, 0, o, d, 0.1, 10000.0)
        b = s.trace(O.ORC_MODE_HIER, 0, 0, o, d, 0.1, 10000.0)
        assert a == b, (i, o, d, a, b)
        hits += a is not None
    assert hits > 100


@pytest.mark.gpu
def test_config5_deep_tree_gpu_parity():
    """hierarchy (4,4,2,2): root 16^3 in LDS, 16^3 level-2 nodes and 4^3 mid nodes in memory."""
    blocks, mats, pal = deep_desc(1e-4)
    assert len(blocks) > 50000
    ctx = api.Context(device=0)
    model = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    scene = api.Scene(ctx)
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = (-2048.0, -2048.0, -2048.0)
    scene.add_instance(model, xf.reshape(12))
    scene.commit()
    os_ = O.Scene()
    os_.add_model(blocks, mats, pal, extent=4096)
    os_.add_instance(0, xf.reshape(12))
    os_.commit()
    noise5 = synth.stbn_unitvec3_cosine(layers=2)
    sky = P.sky_state()
    for eye in ((2600.0, 1900.0, 2300.0), (300.0, 200.0, -150.0)):   # outside on the bounding sphere, and inside the volume
        cam = P.camera_for(eye)
        pipe = api.StandardPipeline(ctx, 160, 100)
        pipe.set_noise(5, noise5)
        passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
        pipe.render(scene, cam, sky, passes, frame_index=1, rand=5)
        g = P.render_oracle(os_, cam, sky, 160, 100, passes, noise5[1], 5)
        P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(pipe)))
        assert np.isfinite(g.depth).mean() > 0.05


@pytest.mark.gpu
def test_config5_deep_tree_sparse_and_dense_cells():
    blocks, mats, pal = P.clustered_deep_model()
    ctx = api.Context(device=0)
    model = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    scene = api.Scene(ctx)
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = (-2048.0, -2048.0, -2048.0)
    scene.add_instance(model, xf.reshape(12))
    scene.commit()
    os_ = O.Scene()
    os_.add_model(blocks, mats, pal, extent=4096)
    os_.add_instance(0, xf.reshape(12))
    os_.commit()
    noise5 = synth.stbn_unitvec3_cosine(layers=2)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    for eye in ((300.0, 200.0, -150.0), (-620.0, 90.0, 40.0), (16.0, 16.0, 700.0)):   # inside the cluster; outside; on a cell plane
        cam = P.camera_for(eye)
        pipe = api.StandardPipeline(ctx, 160, 100)
        pipe.set_noise(5, noise5)
        pipe.render(scene, cam, sky, passes, frame_index=1, rand=5)
        g = P.render_oracle(os_, cam, sky, 160, 100, passes, noise5[1], 5)
        P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(pipe)))
        assert np.isfinite(g.depth).mean() > 0.3


@pytest.mark.gpu
def test_config5_deep_tree_at_frame_size():
    """Config 5's shape at scale: a 4096^3 model with ~1 M random bricks (three-level hierarchy, level-2 nodes and brick
    masks from memory), 1920 x 1080, primary + AO. The oracle runs over the host's cores; the GI passes ride along for two
    frames on the GPU alone, where the regrouped / sorted packets must agree with plain pixel and pool order."""
    import os
    import threading
    blocks, mats, pal = deep_desc(1e-3)
    assert len(blocks) > 900_000
    ctx = api.Context(device=0)
    model = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    scene = api.Scene(ctx)
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = (-2048.0, -2048.0, -2048.0)
    scene.add_instance(model, xf.reshape(12))
    scene.commit()
    os_ = O.Scene()
    os_.add_model(blocks, mats, pal, extent=4096)
    os_.add_instance(0, xf.reshape(12))
    os_.commit()
    w, h = 1920, 1080
    n0, n5 = synth.stbn_scalar(layers=2), synth.stbn_unitvec3_cosine(layers=2)
    sky, cam = P.sky_state(), P.camera_for((300.0, 200.0, -150.0))   # inside the volume
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.render(scene, cam, sky, passes, frame_index=1, rand=5)
    hip = P.read_hip_gbuffer(pipe)
    g = O.GBuffer(w, h)
    n = max(1, min(os.cpu_count() or 1, h // 4))
    cuts = [h * i // n for i in range(n + 1)]
    th = [threading.Thread(target=P.render_oracle, args=(os_, cam, sky, w, h, passes, n5[1], 5),
                           kwargs={"rows": (cuts[i], cuts[i + 1]), "g": g}) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    res = P.compare_gbuffers(g, hip)
    P.assert_parity(res)
    assert res["illuminance_rel_l2"] <= 1e-3, res
    assert np.isfinite(g.depth).mean() > 0.3
    # GI on the deep tree: grouping-independent
    gi_passes = passes | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    states = []
    for env in ({}, {"DUST_HIP_NO_GATHER_ORDER": "1", "DUST_HIP_NO_SURFEL_SORT": "1"}):
        for k in ("DUST_HIP_NO_GATHER_ORDER", "DUST_HIP_NO_SURFEL_SORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        p2 = api.StandardPipeline(ctx, 480, 270)
        p2.set_noise(0, n0)
        p2.set_noise(5, n5)
        p2.configure_gi(1 << 16, 4096)
        for f in (1, 2):
            p2.render(scene, cam, sky, gi_passes, frame_index=f, rand=synth.frame_rand(5, f))
        hsh, pool = p2.read_gi()
        states.append((hsh, pool.view(np.uint32).copy(), p2.read_plane(L.PLANE_ILLUMINANCE)))
    for k in ("DUST_HIP_NO_GATHER_ORDER", "DUST_HIP_NO_SURFEL_SORT"):
        os.environ.pop(k, None)
    for x, y in zip(*states):
        assert np.array_equal(x, y)


@pytest.mark.gpu
def test_config5_deep_tree_full_occupancy():
    """BASELINE configs[4] at its stated size: 4096^3, 1 % of the brick lattice occupied (~10.7 M bricks, ~340 M voxels),
    1920 x 1080. Primary + AO against the oracle over the host's cores; then two GI frames at full frame size whose
    result must not depend on how rays are grouped into wavefronts (octant-ordered gather packets and position-sorted
    surfels against plain pixel / pool order)."""
    import os
    import threading
    blocks, mats, pal = deep_desc(1e-2)
    assert len(blocks) > 10_000_000
    ctx = api.Context(device=0)
    model = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    scene = api.Scene(ctx)
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = (-2048.0, -2048.0, -2048.0)
    scene.add_instance(model, xf.reshape(12))
    scene.commit()
    os_ = O.Scene()
    os_.add_model(blocks, mats, pal, extent=4096)
    os_.add_instance(0, xf.reshape(12))
    os_.commit()
    w, h = 1920, 1080
    n0, n5 = synth.stbn_scalar(layers=2), synth.stbn_unitvec3_cosine(layers=2)
    sky, cam = P.sky_state(), P.camera_for((300.0, 200.0, -150.0))   # inside the volume
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.render(scene, cam, sky, passes, frame_index=1, rand=5)
    hip = P.read_hip_gbuffer(pipe)
    g = O.GBuffer(w, h)
    n = max(1, min(os.cpu_count() or 1, h // 4))
    cuts = [h * i // n for i in range(n + 1)]
    th = [threading.Thread(target=P.render_oracle, args=(os_, cam, sky, w, h, passes, n5[1], 5),
                           kwargs={"rows": (cuts[i], cuts[i + 1]), "g": g}) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    res = P.compare_gbuffers(g, hip)
    P.assert_parity(res)
    assert res["illuminance_rel_l2"] <= 1e-3, res
    assert np.isfinite(g.depth).mean() > 0.9  # at 1 % occupancy nothing sees the sky from inside
    gi_passes = passes | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    states = []
    for env in ({}, {"DUST_HIP_NO_GATHER_ORDER": "1", "DUST_HIP_NO_SURFEL_SORT": "1"}):
        for k in ("DUST_HIP_NO_GATHER_ORDER", "DUST_HIP_NO_SURFEL_SORT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        p2 = api.StandardPipeline(ctx, w, h)
        p2.set_noise(0, n0)
        p2.set_noise(5, n5)
        p2.configure_gi(1 << 22, 65536)
        for f in (1, 2, 3):
            p2.render(scene, cam, sky, gi_passes, frame_index=f, rand=synth.frame_rand(5, f))
        hsh, pool = p2.read_gi()
        states.append((hsh, pool.view(np.uint32).copy(), p2.read_plane(L.PLANE_ILLUMINANCE)))
    for k in ("DUST_HIP_NO_GATHER_ORDER", "DUST_HIP_NO_SURFEL_SORT"):
        os.environ.pop(k, None)
    # radiance only enters the hash where a surfel's ray reaches the sky; from inside a volume this dense few do
    assert int((states[0][0][:, 0] != 0).sum()) > 30 and int((states[0][1].reshape(-1, 4)[:, 3] < 6).sum()) > 10_000
    for x, y in zip(*states):
        assert np.array_equal(x, y)
