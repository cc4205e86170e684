"""Thousands of instances (the reference's TLAS holds one entry per entity, accel_struct/tlas.rs:79-117; a .vox scene graph spawns
one entity per shape node, crates/vox/src/loader.rs:60-176): beyond kFlatCullMax = 256 instances the packet cull goes through the
64-wide hierarchy dust_hip_scene_commit builds (groups of 64 instances along a Morton curve), and the ray streams through the top-level
grid. All five passes against the oracle, which tests every instance box for every ray."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth
from test_gpu_gi import compare_gi

pytestmark = pytest.mark.gpu


def scattered_scene(n_instances, seed=11, n_models=6, span=(520.0, 120.0, 520.0)):
    """Small random models scattered over a wide, flat region with arbitrary rotations about the vertical axis (and a few mirrors):
    every instance's world box is a loose fit of its own."""
    rng = np.random.default_rng(seed)
    pal = synth.make_palette(seed)
    models = []
    for _ in range(n_models):
        sz = tuple(int(v) for v in rng.integers(8, 21, 3))
        models.append(api.flatten_model(P.random_model(rng, sz), sz, pal))
    instances = []
    for i in range(n_instances):
        ang = float(rng.uniform(0.0, 2.0 * np.pi))
        c, s = np.cos(ang), np.sin(ang)
        m = np.zeros((3, 4), np.float32)
        m[:, :3] = np.array([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], np.float32)
        if i % 17 == 0:
            m[:, 0] *= -1.0     # a mirror
        m[:, 3] = (rng.uniform(-0.5, 0.5, 3) * np.array(span)).astype(np.float32)
        instances.append((int(rng.integers(0, n_models)), m.reshape(12)))
    return P.SceneDesc(models, pal, instances)


@pytest.mark.parametrize("stream", [False, True])
def test_4096_instances_match_oracle(monkeypatch, stream):
    if stream:
        monkeypatch.setenv("DUST_HIP_RAY_STREAM", "1")
    else:
        monkeypatch.delenv("DUST_HIP_RAY_STREAM", raising=False)
    desc = scattered_scene(4096)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    oscene = P.oracle_scene(desc)
    sky, cam = P.sky_state(), P.camera_for((180.0, 90.0, 260.0))
    w, h = 192, 108
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 14, 2048)
    gi = O.GI(1 << 14, 2048)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    for f in range(1, 4):
        rnd = synth.frame_rand(9, f)
        pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED | (L.PASS_COUNT_STATS if f == 3 else 0), frame_index=f, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f, gi_threads=8)
        hip = P.read_hip_gbuffer(pipe)
        P.assert_parity(P.compare_gbuffers(g, hip))
        used, valid = compare_gi(gi, pipe)
    ids = hip["voxel_id"][np.isfinite(hip["depth"])] & 0xFFFF
    assert len(np.unique(ids)) > 200          # hundreds of different instances on screen
    assert used > 50 and valid > 50
    st = pipe.pass_stats(0)
    assert st.rays == w * h and st.instances_tested < 40 * st.rays   # (nowhere near 4096 box visits per ray)
    monkeypatch.delenv("DUST_HIP_RAY_STREAM", raising=False)


def test_moved_instances_keep_their_slots():
    """A transform commit refits the groups (and rebuilds the grid) without re-ordering the slots: frames before and after moving a
    tenth of 1024 instances far away, against the oracle."""
    desc = scattered_scene(1024, seed=5, span=(300.0, 80.0, 300.0))
    ctx = api.Context(device=0)
    models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
    scene, oscene = api.Scene(ctx), O.Scene()
    for b, m in desc.models:
        oscene.add_model(b, m, desc.palette)
    ids = []
    for mid, t in desc.instances:
        ids.append(scene.add_instance(models[mid], t))
        oscene.add_instance(mid, t)
    scene.commit()
    oscene.commit()
    sky, cam = P.sky_state(), P.camera_for((120.0, 70.0, 160.0))
    w, h = 160, 96
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(5, n5)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    rng = np.random.default_rng(3)
    for step in range(3):
        if step:
            moved = rng.choice(len(ids), len(ids) // 10, replace=False)
            oscene = O.Scene()
            for b, m in desc.models:
                oscene.add_model(b, m, desc.palette)
            for k in moved:
                mid, t = desc.instances[k]
                t = t.copy().reshape(3, 4)
                t[:, 3] += rng.uniform(-150.0, 150.0, 3).astype(np.float32)
                desc.instances[k] = (mid, t.reshape(12))
                scene.set_transform(ids[k], desc.instances[k][1])
            for mid, t in desc.instances:
                oscene.add_instance(mid, t)
            oscene.commit()
            scene.commit()
        rnd = synth.frame_rand(3, step + 1)
        pipe.render(scene, cam, sky, passes, frame_index=step + 1, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[(step + 1) % 4], rnd)
        cmp = P.compare_gbuffers(g, P.read_hip_gbuffer(pipe))
        for k in ("depth", "voxel_id", "normal", "albedo"):
            assert cmp[k] == 0, (step, cmp)


def test_more_boxes_over_one_cell_than_a_cell_lists(monkeypatch):
    """4200 instances stacked on (nearly) one spot: no grid resolution keeps a cell's list inside the 12 count bits of its cell word, so the
    commit marks the grid unusable and a scene asked to render by ray streams renders by the packet kernels -- same results, nothing dropped."""
    monkeypatch.setenv("DUST_HIP_RAY_STREAM", "1")
    desc = scattered_scene(4200, seed=23, n_models=3, span=(6.0, 3.0, 6.0))
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    oscene = P.oracle_scene(desc)
    sky, cam = P.sky_state(), P.camera_for((40.0, 25.0, 60.0))
    w, h = 48, 32
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 12, 512)
    gi = O.GI(1 << 12, 512)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    for f in range(1, 3):
        rnd = synth.frame_rand(4, f)
        pipe.render(scene, cam, sky, passes, frame_index=f, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f, gi_threads=8)
        P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(pipe)))
        compare_gi(gi, pipe)
    assert np.isfinite(P.read_hip_gbuffer(pipe)["depth"]).sum() > 50


@pytest.mark.parametrize("n,stream", [(20000, False), (20000, True), (65535, False)])
def test_as_many_instances_as_the_api_admits(monkeypatch, n, stream):
    """The instance id is 16 bits wide (dust_hip_scene_add_instance refuses the 65 536th): 20 000 and 65 535 instances -- 313 and 1 024
    groups, several 64-wide rounds of group boxes per cull, an overflowing list where the props bunch up -- against the oracle."""
    if stream:
        monkeypatch.setenv("DUST_HIP_RAY_STREAM", "1")
    else:
        monkeypatch.delenv("DUST_HIP_RAY_STREAM", raising=False)
    desc = scattered_scene(n, seed=31, n_models=4, span=(1500.0, 60.0, 1500.0))
    for k in range(0, 400):      # a heap in the middle: more than kMaxCand boxes over one packet
        mid, t = desc.instances[k]
        t = t.copy().reshape(3, 4)
        t[:, 3] *= np.float32(0.02)
        desc.instances[k] = (mid, t.reshape(12))
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    oscene = P.oracle_scene(desc)
    sky = P.sky_state()
    w, h = (96, 54) if n < 30000 else (64, 36)   # (the oracle tests every box for every ray)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 13, 1024)
    gi = O.GI(1 << 13, 1024)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    for f, eye in ((1, (60.0, 45.0, 90.0)), (2, (700.0, 200.0, 500.0))):
        cam = P.camera_for(eye)
        rnd = synth.frame_rand(6, f)
        pipe.render(scene, cam, sky, passes, frame_index=f, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f, gi_threads=8)
        hip = P.read_hip_gbuffer(pipe)
        P.assert_parity(P.compare_gbuffers(g, hip))
        compare_gi(gi, pipe)
        assert np.isfinite(hip["depth"]).sum() > 200
    if n == 65535:
        with pytest.raises(L.DustError):
            scene.add_instance(scene._models[0], desc.instances[0][1])
