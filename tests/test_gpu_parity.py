"""Parity tests proper: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.
Bit-exact for depth / ids / packed planes; <= 1e-3 relative L2 (north_star tolerance) for radiance, which
passes through exp/pow/acos whose last-ulp behaviour differs between glibc and the device maths library."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    return api.Context(device=0)


@pytest.fixture(scope="module")
def noise5():
    return synth.stbn_unitvec3_cosine(layers=4)


def render_both(ctx, desc, cam, w, h, passes, noise5, frame_index=1, rand=777, sky_name="default", count=False):
    sky = P.sky_state(sky_name)
    scene = P.hip_scene(ctx, desc)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(5, noise5)
    pipe.render(scene, cam, sky, passes | (L.PASS_COUNT_STATS if count else 0), frame_index=frame_index, rand=rand)
    ctx.sync()
    hip = P.read_hip_gbuffer(pipe)
    stats = [O.OrcRayStats(), O.OrcRayStats(), O.OrcRayStats()]
    g = P.render_oracle(P.oracle_scene(desc), cam, sky, w, h, passes, noise5[frame_index % len(noise5)], rand, stats=stats)
    return g, hip, pipe, stats


@pytest.mark.parametrize("seed,eye", [(1, (90.0, 70.0, 110.0)), (2, (-120.0, 40.0, 30.0)), (3, (10.0, 150.0, -20.0)),
                                      (4, (20.0, 10.0, 15.0))])
def test_primary_parity(ctx, noise5, seed, eye):
    desc = P.small_scene(seed=seed)
    g, hip, _, _ = render_both(ctx, desc, P.camera_for(eye), 160, 96, L.PASS_PRIMARY, noise5)
    res = P.compare_gbuffers(g, hip)
    P.assert_parity(res)
    assert np.isfinite(g.depth).sum() > 200


@pytest.mark.parametrize("seed,eye,sky", [(1, (90.0, 70.0, 110.0), "default"), (5, (-60.0, 90.0, 75.0), "low_sun"),
                                          (6, (30.0, 40.0, -95.0), "hazy_noon")])
def test_primary_and_ao_parity(ctx, noise5, seed, eye, sky):
    desc = P.small_scene(seed=seed)
    g, hip, _, _ = render_both(ctx, desc, P.camera_for(eye), 128, 80, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, noise5,
                               frame_index=2, rand=synth.frame_rand(1, 2), sky_name=sky)
    res = P.compare_gbuffers(g, hip)
    P.assert_parity(res)
    ill = P.half_to_float(hip["illuminance"])
    hit = np.isfinite(g.depth)
    assert (ill[hit][:, 3] > 0).any() and (ill[hit][:, 3] == 0).any()  # both AO outcomes occur


def test_axis_aligned_lattice_camera(ctx, noise5):
    """Eye on the voxel lattice looking straight down an axis: rays run along brick faces and through
    edges/corners, the case the conservative walk exists for."""
    desc = P.small_scene(seed=9, n_models=2, n_instances=3)
    rot_cols = np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32).T  # looking along -y
    cam = api.make_camera((8.0, 160.0, 12.0), rot_cols, api.PinholeProjection())
    g, hip, _, _ = render_both(ctx, desc, cam, 128, 128, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, noise5)
    P.assert_parity(P.compare_gbuffers(g, hip))


def test_empty_and_offscreen_scenes(ctx, noise5):
    desc = P.small_scene(seed=2, n_models=1, n_instances=1)
    # camera looking away: every pixel misses
    cam = api.make_camera((300.0, 300.0, 300.0), api.look_at_rotation((300.0, 300.0, 300.0), (600.0, 600.0, 600.0)),
                          api.PinholeProjection())
    g, hip, _, _ = render_both(ctx, desc, cam, 64, 40, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, noise5)
    assert not np.isfinite(g.depth).any()
    P.assert_parity(P.compare_gbuffers(g, hip))
    # ragged frame size (not a multiple of the 8x8 packet)
    g, hip, _, _ = render_both(ctx, desc, P.camera_for((70.0, 60.0, 90.0)), 61, 37, L.PASS_PRIMARY, noise5)
    P.assert_parity(P.compare_gbuffers(g, hip))


def test_castle_standin_parity_and_stats(ctx, noise5):
    data, info = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    s = 0.15
    eye = (122.0 * s, 300.61 * s, 54.45 * s)  # examples/castle.rs:126 scaled with the scene
    g, hip, pipe, ostats = render_both(ctx, desc, P.camera_for(eye), 192, 108, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION,
                                       noise5, count=True)
    P.assert_parity(P.compare_gbuffers(g, hip))
    assert np.isfinite(g.depth).mean() > 0.5
    # algorithmic-bytes accounting: the counting build of the kernels issues the same rays and finds the same hits as
    # the oracle; it visits instances front to back per packet (the oracle walks them in index order), so closest-hit
    # rays do less traversal work than the oracle counts and any-hit rays about the same (a different first hit).
    for i in range(3):
        st = pipe.pass_stats(i)
        o = ostats[i]
        assert st.rays == o.rays and st.hits == o.hits, (i, st.rays, o.rays, st.hits, o.hits)
        assert st.hits <= st.bricks_tested <= o.bricks_tested * 1.02 + 8, (i, st.hits, st.bricks_tested, o.bricks_tested)
        assert st.mid_descents <= o.mid_descents * 1.02 + 8 and st.upper_descents <= o.upper_descents * 1.02 + 8
        assert st.instances_tested <= o.instances_tested * 1.02 + 8


def test_row_bands_equal_full_frame(ctx, noise5):
    """Multi-GPU sharding renders row bands; the union of bands must equal the full frame bit for bit."""
    desc = P.small_scene(seed=4)
    sky = P.sky_state()
    cam = P.camera_for((90.0, 70.0, 110.0))
    scene = P.hip_scene(ctx, desc)
    w, h = 120, 75
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    full = api.StandardPipeline(ctx, w, h)
    full.set_noise(5, noise5)
    full.render(scene, cam, sky, passes, frame_index=3, rand=5)
    ref = P.read_hip_gbuffer(full)
    banded = api.StandardPipeline(ctx, w, h)
    banded.set_noise(5, noise5)
    for r0, r1 in ((0, 19), (19, 38), (38, 57), (57, 75)):
        banded.render(scene, cam, sky, passes, frame_index=3, rand=5, rows=(r0, r1))
    got = P.read_hip_gbuffer(banded)
    for k in ref:
        assert ref[k].tobytes() == got[k].tobytes(), k


def test_fused_and_separate_launches_agree(ctx, noise5, monkeypatch):
    """primary + AO run as one fused kernel by default; launched separately (as the reference's two trace calls)
    they must leave bit-identical planes."""
    desc = P.small_scene(seed=11)
    sky = P.sky_state()
    cam = P.camera_for((60.0, 90.0, 100.0))
    scene = P.hip_scene(ctx, desc)
    outs = []
    for no_fuse in (False, True):
        if no_fuse:
            monkeypatch.setenv("DUST_HIP_NO_FUSE", "1")
        else:
            monkeypatch.delenv("DUST_HIP_NO_FUSE", raising=False)
        pipe = api.StandardPipeline(ctx, 150, 90)
        pipe.set_noise(5, noise5)
        pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, frame_index=2, rand=31)
        outs.append(P.read_hip_gbuffer(pipe))
    monkeypatch.delenv("DUST_HIP_NO_FUSE", raising=False)
    for k in outs[0]:
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k


def test_tile_order_does_not_change_results(ctx, noise5, monkeypatch):
    """Cost-ordered hand-out (k_tile_order): frames rendered with the measured order -- first launch in screen order, then
    ordered, re-measured, kept on a still view, and again after the camera moves -- are bit-identical to screen-order
    launches (DUST_HIP_NO_TILE_ORDER), and the heat map covers the tile grid."""
    desc = P.small_scene(seed=13)
    sky = P.sky_state()
    cams = [P.camera_for((60.0, 90.0, 100.0)), P.camera_for((-40.0, 70.0, 120.0))]
    scene = P.hip_scene(ctx, desc)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    runs = []
    for off in (False, True):
        if off:
            monkeypatch.setenv("DUST_HIP_NO_TILE_ORDER", "1")
        else:
            monkeypatch.delenv("DUST_HIP_NO_TILE_ORDER", raising=False)
        pipe = api.StandardPipeline(ctx, 203, 131)
        pipe.set_noise(5, noise5)
        frames = []
        for f in range(12):  # 10 frames on one view (crosses the re-measure interval), then a moved camera
            pipe.render(scene, cams[0 if f < 10 else 1], sky, passes, frame_index=f + 1, rand=7 + f)
            if f in (0, 1, 2, 9, 10, 11):
                frames.append(P.read_hip_gbuffer(pipe))
        if not off:
            costs = pipe.tile_costs(0)
            assert costs.shape == ((131 + 7) // 8, (203 + 7) // 8) and costs.max() > 0
        runs.append(frames)
    monkeypatch.delenv("DUST_HIP_NO_TILE_ORDER", raising=False)
    for a, b in zip(*runs):
        for k in a:
            assert a[k].tobytes() == b[k].tobytes(), k


def test_repeatable(ctx, noise5):
    desc = P.small_scene(seed=8)
    sky = P.sky_state()
    cam = P.camera_for((50.0, 80.0, 120.0))
    scene = P.hip_scene(ctx, desc)
    pipe = api.StandardPipeline(ctx, 200, 120)
    pipe.set_noise(5, noise5)
    outs = []
    for _ in range(3):
        pipe.clear()
        pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, frame_index=1, rand=1)
        outs.append(P.read_hip_gbuffer(pipe))
    for k in outs[0]:
        assert outs[0][k].tobytes() == outs[1][k].tobytes() == outs[2][k].tobytes(), k


def test_api_errors(ctx, noise5):
    desc = P.small_scene(seed=1, n_models=1, n_instances=1)
    scene = P.hip_scene(ctx, desc)
    pipe = api.StandardPipeline(ctx, 32, 32)
    with pytest.raises(L.DustError) as e:  # noise not loaded -> render is "not ready" (standard.rs:254)
        pipe.render(scene, P.camera_for((90.0, 70.0, 110.0)), P.sky_state(), L.PASS_AMBIENT_OCCLUSION)
    assert e.value.status == L.ERR_NOT_READY
    b, m = desc.models[0]
    with pytest.raises(L.DustError):  # blocks out of Tree::iter_leaf order
        api.Model(ctx, b[::-1].copy(), m, desc.palette)
    with pytest.raises(L.DustError):  # singular transform
        api.Scene(ctx).add_instance(api.Model(ctx, b, m, desc.palette), np.zeros(12, np.float32))


def test_candidate_list_and_index_order_agree(ctx, noise5, monkeypatch):
    """The packet's sorted candidate list (culling, front-to-back order, early exit) against the plain walk over every
    instance in index order (DUST_HIP_DEBUG bit 4): every plane identical on the castle, whose instances overlap."""
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    scene = P.hip_scene(ctx, desc)
    s = 0.15
    cam, sky = P.camera_for((122.0 * s, 300.61 * s, 54.45 * s)), P.sky_state()
    outs = []
    for dbg in (None, "4"):
        if dbg:
            monkeypatch.setenv("DUST_HIP_DEBUG", dbg)
        else:
            monkeypatch.delenv("DUST_HIP_DEBUG", raising=False)
        pipe = api.StandardPipeline(ctx, 384, 216)
        pipe.set_noise(5, noise5)
        pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, frame_index=3, rand=99)
        outs.append(P.read_hip_gbuffer(pipe))
    monkeypatch.delenv("DUST_HIP_DEBUG", raising=False)
    for k in outs[0]:
        assert outs[0][k].tobytes() == outs[1][k].tobytes(), k


def test_moving_instance_motion_vectors(ctx, noise5):
    """teapot_move_system (examples/castle.rs:287-291): an instance moves between frames; the motion plane is measured
    against the previous frame's object-to-world matrix (standard.rs:845-878), bit for bit like the oracle's."""
    desc = P.small_scene(seed=21, n_models=2, n_instances=3, size=(32, 32, 32))
    sky, cam = P.sky_state(), P.camera_for((70.0, 55.0, 80.0))
    models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
    scene = api.Scene(ctx)
    ids = [scene.add_instance(models[mid], t) for mid, t in desc.instances]
    scene.commit()
    w, h = 160, 100
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(5, noise5)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION

    def mat4_cols(t12):  # 3x4 row-major -> column-major mat4
        m4 = np.eye(4, dtype=np.float32)
        m4[:3, :] = np.asarray(t12, np.float32).reshape(3, 4)
        return m4.T.reshape(16).copy()

    cur = [np.asarray(t, np.float32).reshape(12).copy() for _, t in desc.instances]
    for frame in range(1, 4):
        prev = [mat4_cols(t) for t in cur]
        cur[0] = cur[0].copy()
        cur[0][7] += 2.5                       # the first instance rises 2.5 units per frame
        cur[1] = cur[1].copy()
        cur[1][3] -= 0.75                      # the second slides along x
        for j in (0, 1):
            scene.set_transform(ids[j], cur[j], prev[j])
        scene.commit()
        os_ = O.Scene()
        for b, m in desc.models:
            os_.add_model(b, m, desc.palette)
        for j, (mid, _) in enumerate(desc.instances):
            os_.add_instance(mid, cur[j], prev[j] if j < 2 else None)
        os_.commit()
        pipe.render(scene, cam, sky, passes, frame_index=frame, rand=frame * 13)
        g = P.render_oracle(os_, cam, sky, w, h, passes, noise5[frame % len(noise5)], frame * 13)
        hip = P.read_hip_gbuffer(pipe)
        P.assert_parity(P.compare_gbuffers(g, hip))
        hit = np.isfinite(g.depth)
        assert (P.half_to_float(hip["motion"])[hit][:, :3] != 0).any()   # something actually moved


def test_launch_shape_does_not_change_results(ctx, noise5, monkeypatch):
    """Workgroup size, workgroups per CU and slots left free for other queues (DUST_HIP_BLOCK / _BLOCKS_PER_CU /
    _RESERVE_BLOCKS) decide which wavefront traces which tile and through which LDS queue -- never what a pixel gets."""
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    scene = P.hip_scene(ctx, desc)
    s = 0.15
    cam, sky = P.camera_for((122.0 * s, 300.61 * s, 54.45 * s)), P.sky_state()
    keys = ("DUST_HIP_BLOCK", "DUST_HIP_BLOCKS_PER_CU", "DUST_HIP_RESERVE_BLOCKS")
    outs = []
    for env in ({}, {"DUST_HIP_BLOCK": "256"}, {"DUST_HIP_BLOCK": "64", "DUST_HIP_BLOCKS_PER_CU": "1"}, {"DUST_HIP_RESERVE_BLOCKS": "64"},
                {"DUST_HIP_RESERVE_BLOCKS": "100000"}):
        for k in keys:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pipe = api.StandardPipeline(ctx, 384, 216)
        pipe.set_noise(5, noise5)
        pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, frame_index=3, rand=99)
        outs.append(P.read_hip_gbuffer(pipe))
    for k in keys:
        monkeypatch.delenv(k, raising=False)
    for other in outs[1:]:
        for k in outs[0]:
            assert outs[0][k].tobytes() == other[k].tobytes(), k


def test_bound_plane_equals_own_storage():
    """dust_hip_pipeline_bind_plane: a frame rendered into caller-owned storage (here: a second pipeline's depth-sized
    scratch is not needed -- a raw hipMalloc through torch) carries the same bits as the pipeline's own plane, the other
    planes are untouched by the redirection, and unbinding restores the pipeline's storage."""
    import torch
    desc = P.small_scene(seed=21, n_models=3, n_instances=6, size=(40, 36, 28))
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky, cam = P.sky_state(), P.camera_for((60.0, 40.0, 55.0))
    n5 = synth.stbn_unitvec3_cosine(layers=2)
    W, H = 120, 72
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION

    def frame(bind):
        pipe = api.StandardPipeline(ctx, W, H)
        pipe.set_noise(5, n5)
        target = torch.zeros((H, W, 4), dtype=torch.float16, device="cuda")
        if bind:
            pipe.bind_plane(L.PLANE_ILLUMINANCE, target.data_ptr(), target.numel() * 2)
        pipe.render(scene, cam, sky, passes, frame_index=1, rand=synth.frame_rand(3, 1))
        ctx.sync()
        ill = pipe.read_plane(L.PLANE_ILLUMINANCE)
        out = (ill, pipe.read_plane(L.PLANE_DEPTH), target.cpu().numpy().view(np.uint16).copy())
        if bind:
            pipe.bind_plane(L.PLANE_ILLUMINANCE, 0, 0)
            assert not pipe.read_plane(L.PLANE_ILLUMINANCE).any()   # the pipeline's own (never written) storage is back
        return out

    own_ill, own_depth, untouched = frame(False)
    b_ill, b_depth, b_target = frame(True)
    assert own_ill.view(np.uint16).any() and not untouched.any()
    assert np.array_equal(own_ill.view(np.uint16), b_ill.view(np.uint16))
    assert np.array_equal(own_ill.view(np.uint16), b_target.reshape(own_ill.view(np.uint16).shape))
    assert np.array_equal(own_depth.view(np.uint32), b_depth.view(np.uint32))
    import pytest
    pipe = api.StandardPipeline(ctx, W, H)
    with pytest.raises(L.DustError):
        pipe.bind_plane(L.PLANE_ILLUMINANCE, torch.zeros(16, device="cuda").data_ptr(), 64)   # too small


def test_moving_view_hand_out_never_changes_a_result():
    """A view that moves: the cost-balanced bands, the order re-made every fourth launch from running-mean costs spread over 3 x 3 tiles,
    set_transform + commit into the ring of scene images every frame -- all of it decides which wave traces which tile when, never a
    texel: every frame of a moving sequence equals the same frame rendered by a pipeline and a scene that have no history."""
    W, H = 264, 152
    ctx = api.Context(device=0)
    desc = P.small_scene(seed=21, n_models=3, n_instances=7)
    scene = P.hip_scene(ctx, desc)
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    pipe = api.StandardPipeline(ctx, W, H)
    pipe.set_noise(5, n5)
    moved = 3
    base = np.asarray(desc.instances[moved][1], np.float32).reshape(3, 4)
    for f in range(1, 22):
        eye = (90.0 * np.cos(0.05 * f), 60.0 + f, -80.0 * np.sin(0.05 * f) - 40.0)
        cam = P.camera_for(eye)
        xf = base.copy()
        xf[:, 3] += np.array([2.0 * f, 0.0, -1.5 * f], np.float32)
        scene.set_transform(moved, xf.reshape(12))
        scene.commit()
        pipe.render(scene, cam, sky, passes, f, synth.frame_rand(3, f))
        if f % 4 == 1 or f > 17:
            got = P.read_hip_gbuffer(pipe)
            d2 = P.SceneDesc(desc.models, desc.palette, [(m, (xf.reshape(12) if i == moved else t)) for i, (m, t) in enumerate(desc.instances)])
            fresh_scene = P.hip_scene(ctx, d2)
            fresh = api.StandardPipeline(ctx, W, H)
            fresh.set_noise(5, n5)
            fresh.render(fresh_scene, cam, sky, passes, f, synth.frame_rand(3, f))
            want = P.read_hip_gbuffer(fresh)
            hit = np.isfinite(want["depth"])
            assert hit.any() and not hit.all()
            for k in want:
                if k == "motion":
                    continue   # (the moved instance's motion vectors are measured against its previous transform, which the fresh scene does not have)
                a, b = want[k], got[k]
                if k in ("illuminance", "normal", "voxel_id"):   # planes only hit pixels write (hit.rchit:57-94): a pixel that misses keeps what an earlier frame left
                    a, b = a[hit], b[hit]
                elif k == "denoised":                             # ... and this one only the pixels that miss (miss.rmiss:13)
                    a, b = a[~hit], b[~hit]
                assert a.tobytes() == b.tobytes(), (f, k)
