"""Edges of the boundary that round 3's host runtime added or rewrote: empty scenes, frames smaller than a ray packet, a scene
that grows after its first commit (the device image is reallocated), two pipelines sharing one context's second stream, a
model edited while a surfel pass is still running on it. Each is checked against the plain way of getting the same result."""
import numpy as np
import pytest

import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth

pytestmark = pytest.mark.gpu

PA = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
GI = PA | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED


def _pipe(ctx, w, h, gi=False):
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(5, synth.stbn_unitvec3_cosine(layers=4))
    if gi:
        pipe.set_noise(0, synth.stbn_scalar(layers=4))
        pipe.configure_gi(4093, 777)
    return pipe


def _planes(pipe):
    return [pipe.read_plane(pl) for pl in (L.PLANE_DEPTH, L.PLANE_VOXEL_ID, L.PLANE_ALBEDO, L.PLANE_NORMAL, L.PLANE_ILLUMINANCE, L.PLANE_DENOISED)]


def test_empty_scene_renders_sky():
    ctx = api.Context(device=0)
    scene = api.Scene(ctx)
    scene.commit()
    pipe = _pipe(ctx, 40, 24, gi=True)
    cam, sky = P.camera_for((30.0, 20.0, -40.0)), P.sky_state()
    for f in (1, 2):
        pipe.render(scene, cam, sky, GI, frame_index=f, rand=5 + f)
    depth = pipe.read_plane(L.PLANE_DEPTH)
    assert np.isinf(depth).all()
    assert (pipe.read_plane(L.PLANE_ALBEDO) == 0xFFFFFFFF).all()          # miss.rmiss
    oscene = P.oracle_scene(P.SceneDesc([], synth.make_palette(1), []))
    g = P.render_oracle(oscene, cam, sky, 40, 24, PA, synth.stbn_unitvec3_cosine(layers=4)[2 % 4], 7)
    res = P.compare_gbuffers(g, P.read_hip_gbuffer(pipe))
    P.assert_parity(res)


@pytest.mark.parametrize("w,h", [(1, 1), (7, 3), (9, 17), (8, 137), (5, 40)])   # (the last two: ONE column of tiles, several rows of them)
def test_frames_smaller_than_a_packet(w, h):
    desc = P.small_scene(seed=4, n_models=2, n_instances=4)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    pipe = _pipe(ctx, w, h)
    cam, sky = P.camera_for((80.0, 50.0, -70.0)), P.sky_state()
    pipe.render(scene, cam, sky, PA, frame_index=1, rand=3)
    g = P.render_oracle(P.oracle_scene(desc), cam, sky, w, h, PA, synth.stbn_unitvec3_cosine(layers=4)[1], 3)
    P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(pipe)))


def test_scene_that_grows_after_its_first_commit():
    """Instances added after a commit (the image is reallocated, the kernels' pointers move) with frames in flight: the frame
    equals a scene built in one go."""
    desc = P.small_scene(seed=9, n_models=3, n_instances=40)
    ctx = api.Context(device=0)
    cam, sky = P.camera_for((120.0, 70.0, -110.0)), P.sky_state()
    want_scene = P.hip_scene(ctx, desc)
    want_pipe = _pipe(ctx, 96, 64)
    want_pipe.render(want_scene, cam, sky, PA, frame_index=1, rand=11)
    want = _planes(want_pipe)
    models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
    scene = api.Scene(ctx)
    pipe = _pipe(ctx, 96, 64)
    for k, (mid, t) in enumerate(desc.instances):
        scene.add_instance(models[mid], t)
        if k in (0, 3, 17, 18, 39):           # commits (and frames) at several sizes on the way
            scene.commit()
            if k == 39:
                pipe.clear()                  # (planes a hit pixel does not write keep what earlier, smaller scenes left there)
            pipe.render(scene, cam, sky, PA, frame_index=1, rand=11)
    for x, y in zip(want, _planes(pipe)):
        assert np.array_equal(x, y)


def test_two_pipelines_share_the_contexts_second_stream():
    """Two pipelines of one context run GI frames in turn: each pipeline's surfel pass goes to the context's second stream, and the
    other pipeline's gather waits for it as it would for its own. Results equal two contexts of their own."""
    desc = P.small_scene(seed=6, n_models=3, n_instances=7)
    cam_a, cam_b, sky = P.camera_for((90.0, 60.0, -80.0)), P.camera_for((-70.0, 40.0, 95.0)), P.sky_state()

    def alone(cam, w, h):
        ctx = api.Context(device=0)
        scene = P.hip_scene(ctx, desc)
        pipe = _pipe(ctx, w, h, gi=True)
        for f in range(1, 5):
            pipe.render(scene, cam, sky, GI, frame_index=f, rand=synth.frame_rand(2, f))
        h_, s_ = pipe.read_gi()
        return [h_, s_.view(np.uint32).copy()] + _planes(pipe)

    want_a, want_b = alone(cam_a, 96, 64), alone(cam_b, 80, 48)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    pa, pb = _pipe(ctx, 96, 64, gi=True), _pipe(ctx, 80, 48, gi=True)
    for f in range(1, 5):
        pa.render(scene, cam_a, sky, GI, frame_index=f, rand=synth.frame_rand(2, f))
        pb.render(scene, cam_b, sky, GI, frame_index=f, rand=synth.frame_rand(2, f))
    for pipe, want in ((pa, want_a), (pb, want_b)):
        h_, s_ = pipe.read_gi()
        for x, y in zip(want, [h_, s_.view(np.uint32).copy()] + _planes(pipe)):
            assert np.array_equal(x, y)


def test_model_edit_behind_a_surfel_pass_in_flight():
    """dust_hip_model_set_voxels right after a GI frame: the frame's surfel pass may still be tracing the model on the second
    stream. The edit waits for it (on the device); the next frames equal the same sequence with a full wait in between."""
    desc = P.small_scene(seed=8, n_models=2, n_instances=5)
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    xyz = np.array([[3, 4, 5], [3, 4, 6], [10, 2, 9], [0, 0, 0]], np.uint32)
    vals = np.array([5, 7, -1, 9], np.int32)

    def run(wait):
        ctx = api.Context(device=0)
        models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
        scene = api.Scene(ctx)
        for mid, t in desc.instances:
            scene.add_instance(models[mid], t)
        scene.commit()
        pipe = _pipe(ctx, 96, 64, gi=True)
        for f in (1, 2):
            pipe.render(scene, cam, sky, GI, frame_index=f, rand=synth.frame_rand(4, f))
        if wait:
            ctx.sync()
        models[0].set_voxels(xyz, vals)
        scene.commit()
        for f in (3, 4):
            pipe.render(scene, cam, sky, GI, frame_index=f, rand=synth.frame_rand(4, f))
        h_, s_ = pipe.read_gi()
        return [h_, s_.view(np.uint32).copy()] + _planes(pipe)

    for x, y in zip(run(True), run(False)):
        assert np.array_equal(x, y)


def test_frames_in_flight_changes_no_result():
    """dust_hip_pipeline_set_frames_in_flight only sizes the launches: two pipelines of two contexts rendering side by side on a
    quarter of the slots each produce the planes of a launch that had the device to itself; out-of-range values are refused."""
    desc = P.small_scene(seed=11, n_models=3, n_instances=9)
    cam, sky = P.camera_for((100.0, 70.0, -90.0)), P.sky_state()
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    pipe = _pipe(ctx, 200, 120)
    pipe.render(scene, cam, sky, PA, frame_index=1, rand=21)
    want = _planes(pipe)
    ctxs = [api.Context(device=0) for _ in range(2)]
    scenes = [P.hip_scene(c, desc) for c in ctxs]
    pipes = [_pipe(c, 200, 120) for c in ctxs]
    for p_ in pipes:
        p_.set_frames_in_flight(4)
    for f in range(3):
        for p_, s_ in zip(pipes, scenes):
            p_.render(s_, cam, sky, PA, frame_index=1, rand=21)
    for p_ in pipes:
        for x, y in zip(want, _planes(p_)):
            assert np.array_equal(x, y)
    for bad in (0, 17):
        with pytest.raises(Exception):
            pipe.set_frames_in_flight(bad)


def test_degenerate_cameras_and_transforms_neither_hang_nor_fault():
    """A host bug upstream (NaN from a look-at at the pole, an uninitialised projection, a collapsed scale) must not take the device
    down: frames from NaN / infinite / zero / enormous cameras come back at once -- nothing visible --, singular and non-finite
    instance transforms are refused at dust_hip_scene_add_instance, and a good frame renders afterwards as before."""
    import time
    desc = P.small_scene(seed=2, n_models=1, n_instances=2)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky, base = P.sky_state(), P.camera_for((60.0, 50.0, 70.0))
    good = _pipe(ctx, 64, 32, gi=True)
    good.render(scene, base, sky, GI, frame_index=1, rand=5)
    want = _planes(good)

    def cam_with(**kw):
        c = L.Camera.from_buffer_copy(base)
        for k, v in kw.items():
            f = getattr(c, k)
            if hasattr(f, "__len__"):
                for i in range(len(f)):
                    f[i] = v
            else:
                setattr(c, k, v)
        return c

    nan, inf = float("nan"), float("inf")
    t0 = time.time()
    for kw in (dict(position=nan), dict(position=inf), dict(position=1e30), dict(view_col0=nan), dict(view_col0=0.0, view_col1=0.0, view_col2=0.0),
               dict(view_col2=inf), dict(tan_half_fov=0.0), dict(tan_half_fov=nan), dict(tan_half_fov=1e30), dict(near_=1e4, far_=0.1), dict(near_=-5.0)):
        pipe = _pipe(ctx, 64, 32, gi=True)
        for f in (1, 2):
            pipe.render(scene, cam_with(**kw), sky, GI, frame_index=f, rand=5 + f)
        pipe.read_plane(L.PLANE_DEPTH)
    assert time.time() - t0 < 20.0
    model = api.Model(ctx, desc.models[0][0], desc.models[0][1], desc.palette)
    for mat in (np.zeros((3, 4)), np.full((3, 4), np.nan), np.array([[1, 0, 0, np.inf], [0, 1, 0, 0], [0, 0, 1, 0]]), np.ones((3, 4)), np.eye(3, 4) * 1e-30):
        s2 = api.Scene(ctx)
        with pytest.raises(L.DustError):
            s2.add_instance(model, np.asarray(mat, np.float32).reshape(12))
    again = _pipe(ctx, 64, 32, gi=True)
    again.render(scene, base, sky, GI, frame_index=1, rand=5)
    for x, y in zip(want, _planes(again)):
        assert np.array_equal(x, y)


def test_pipeline_config_round_trips_and_changes_no_result(monkeypatch):
    """DustHipPipelineConfig (round 6): the production knobs arrive through the C ABI, not the environment. What is set is what is read
    back; out-of-range fields are refused and leave the pipeline as it was; and none of them changes a result -- GI frames with both
    passes as ray streams, with slots reserved, with the surfel pass in place and with every launch asking for all slots while
    'three frames are in flight' give the planes, hash and pool of the default configuration, bit for bit."""
    for k in ("DUST_HIP_RAY_STREAM", "DUST_HIP_PACKET_GI", "DUST_HIP_NO_SIDE_STREAM", "DUST_HIP_SIDE_SHARE", "DUST_HIP_RESERVE_BLOCKS"):
        monkeypatch.delenv(k, raising=False)
    desc = P.small_scene(seed=21, n_models=3, n_instances=7)
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    pipe = _pipe(ctx, 136, 72, gi=True)
    assert pipe.get_config() == {"reserve_blocks": "auto", "gi_path": "auto", "side_stream": "auto", "side_share": 0, "frames_in_flight": 1,
                                 "in_flight_slots": "share"}
    pipe.configure(reserve_blocks=40, gi_path="streams", side_stream="off", side_share=30, frames_in_flight=3, in_flight_slots="all")
    assert pipe.get_config() == {"reserve_blocks": 40, "gi_path": "streams", "side_stream": "off", "side_share": 30, "frames_in_flight": 3,
                                 "in_flight_slots": "all"}
    for bad in ({"gi_path": 3}, {"side_stream": 2}, {"side_share": 3}, {"side_share": 95}, {"frames_in_flight": 17}, {"in_flight_slots": 2}):
        with pytest.raises(L.DustError):
            pipe.configure(**bad)
    assert pipe.get_config()["gi_path"] == "streams" and pipe.get_config()["reserve_blocks"] == 40
    small = L.PipelineConfig(4)   # a struct_size below this library's: refused, nothing read
    assert ctx._lib.dust_hip_pipeline_configure(pipe._h, small) == L.ERR_INVALID_ARGUMENT
    outs = []
    for cfg in ({}, {"gi_path": "streams"}, {"gi_path": "packets", "reserve_blocks": 32, "side_stream": "off"},
                {"frames_in_flight": 3, "in_flight_slots": "all", "side_share": 25}, {"frames_in_flight": 2, "in_flight_slots": "share"},
                {"reserve_blocks": 100000}):   # (more slots than the device has: the launch keeps what it needs)
        p_ = _pipe(ctx, 136, 72, gi=True)
        p_.configure(**cfg)
        for f in range(1, 5):
            p_.render(scene, cam, sky, GI, frame_index=f, rand=synth.frame_rand(4, f))
        outs.append((_planes(p_), p_.read_gi()))
    assert int((outs[0][1][0][:, 0] != 0).sum()) > 20
    for planes, (h, sp) in outs[1:]:
        for x, y in zip(outs[0][0], planes):
            assert np.array_equal(x, y)
        assert np.array_equal(outs[0][1][0], h) and outs[0][1][1].tobytes() == sp.tobytes()
