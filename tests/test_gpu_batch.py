"""Several frames in ONE persistent launch (dust_hip_render_frames, k_primary_ao_batch): the reference keeps up to three frames in
flight (rhyolite_bevy/src/lib.rs:58) and StandardPipeline::render (standard.rs:228-810) is called once per frame; here the frames of a
call share a launch -- a wavefront that finds frame i without tiles goes on to frame i + 1. Every plane of every frame must hold the bits
the same frame rendered alone holds, against the oracle and against dust_hip_render_frame, for every kernel variant (plain, DEEP, LARGE)."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth
from test_gpu_many_instances import scattered_scene

pytestmark = pytest.mark.gpu

PAO = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
PLANES = [pid for _, pid in P.PLANES]


def _pipes(ctx, n, w, h, n5, n0=None):
    out = []
    for _ in range(n):
        p = api.StandardPipeline(ctx, w, h)
        p.set_noise(5, n5)
        if n0 is not None:
            p.set_noise(0, n0)
        out.append(p)
    return out


def _cams(n, eye=(90.0, 60.0, -80.0)):
    # a camera per frame: the frames of a launch need not share a view
    return [P.camera_for((eye[0] + 3.0 * i, eye[1] - 2.0 * i, eye[2] + 1.5 * i)) for i in range(n)]


def _planes(pipe):
    return [pipe.read_plane(pl) for pl in PLANES]


def _same(a, b, what):
    for pl, x, y in zip(PLANES, a, b):
        assert np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"{what}: plane {pl} differs in {int(np.count_nonzero(x != y))} values"


@pytest.mark.parametrize("n", [2, 3, 5, 8, 11])
def test_batched_frames_equal_single_frames_and_the_oracle(n):
    """n frames (11: a launch of 8 and one of 3), each with its own camera, frame index and rand, at a frame size that leaves ragged tiles:
    == the same frames one dust_hip_render_frame at a time, bit for bit on every plane; and == the oracle."""
    w, h = 203, 117
    desc = P.small_scene(seed=5, n_models=3, n_instances=7)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    oscene = P.oracle_scene(desc)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    cams = _cams(n)
    idx = [7 + i for i in range(n)]
    rnd = [synth.frame_rand(3, f) for f in idx]
    batch = _pipes(ctx, n, w, h, n5)
    api.StandardPipeline.render_frames(batch, scene, cams, sky, PAO, idx, rnd)
    single = _pipes(ctx, n, w, h, n5)
    for i in range(n):
        single[i].render(scene, cams[i], sky, PAO, frame_index=idx[i], rand=rnd[i])
    for i in range(n):
        _same(_planes(batch[i]), _planes(single[i]), f"frame {i} of {n}")
    for i in (0, n - 1):
        g = P.render_oracle(oscene, cams[i], sky, w, h, PAO, n5[idx[i] % 4], rnd[i])
        P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(batch[i])))


def test_batches_in_a_row_keep_their_counters_and_orders():
    """Twelve launches of three frames on the same three pipelines: the work counters alternate per pipeline (a launch zeroes the set the
    next one uses -- for every frame it carries), the tile costs are measured and the cost order comes in on the way. The last launch's
    planes == the same frames alone; a single-frame call in between uses the same counters."""
    w, h = 320, 200
    desc = P.small_scene(seed=8, n_models=4, n_instances=9)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    cams = _cams(3)
    batch = _pipes(ctx, 3, w, h, n5)
    f = 1
    for k in range(12):
        idx = [f, f + 1, f + 2]
        api.StandardPipeline.render_frames(batch, scene, cams, sky, PAO, idx, [synth.frame_rand(5, v) for v in idx])
        f += 3
        if k == 5:   # one of the pipelines renders a frame of its own between two launches
            batch[1].render(scene, cams[1], sky, PAO, frame_index=999, rand=17)
    last = [f - 3, f - 2, f - 1]
    single = _pipes(ctx, 3, w, h, n5)
    for i in range(3):
        single[i].render(scene, cams[i], sky, PAO, frame_index=last[i], rand=synth.frame_rand(5, last[i]))
        _same(_planes(batch[i]), _planes(single[i]), f"frame {i} of the twelfth launch")
    costs = batch[2].tile_costs(0)   # a follower's tiles were timed in its own cost map
    assert costs is not None and costs.shape == ((h + 7) // 8, (w + 7) // 8) and costs.any()


def test_batched_row_bands():
    """a launch of four frames of ONE row band (what a rank of an N-GPU job renders): rows outside the band keep what they held"""
    w, h = 256, 144
    desc = P.small_scene(seed=2, n_models=3, n_instances=6)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    cams = _cams(4)
    rows = (40, 104)
    batch = _pipes(ctx, 4, w, h, n5)
    single = _pipes(ctx, 4, w, h, n5)
    idx = [3, 4, 5, 6]
    api.StandardPipeline.render_frames(batch, scene, cams, sky, PAO, idx, idx, rows=rows)
    for i in range(4):
        single[i].render(scene, cams[i], sky, PAO, frame_index=idx[i], rand=idx[i], rows=rows)
        _same(_planes(batch[i]), _planes(single[i]), f"band frame {i}")
    assert not batch[0].read_plane(L.PLANE_DEPTH)[:40].any() and batch[0].read_plane(L.PLANE_DEPTH)[40:104].any()


def test_frames_that_cannot_share_a_launch_run_in_sequence():
    """GI frames (a frame's gather reads the hash its predecessor's surfel pass wrote), the same pipeline twice, pipelines of two frame sizes:
    dust_hip_render_frames enqueues them one after the other -- the results of that many dust_hip_render_frame calls."""
    w, h = 128, 80
    desc = P.small_scene(seed=4, n_models=3, n_instances=6)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky = P.sky_state()
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    cam = P.camera_for((90.0, 60.0, -80.0))
    gi = PAO | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    a, b = _pipes(ctx, 2, w, h, n5, n0)
    for p in (a, b):
        p.configure_gi(4093, 777)
    # three GI frames of ONE pipeline in one call == three calls
    api.StandardPipeline.render_frames([a, a, a], scene, cam, sky, gi, [1, 2, 3], [11, 12, 13])
    for f in (1, 2, 3):
        b.render(scene, cam, sky, gi, frame_index=f, rand=10 + f)
    _same(_planes(a), _planes(b), "GI frames in one call")
    ha, hb = a.read_gi(), b.read_gi()
    assert np.array_equal(ha[0], hb[0]) and np.array_equal(ha[1].view(np.uint32), hb[1].view(np.uint32))
    # the same pipeline twice with primary + AO frames: the second frame is what the planes hold
    c, d = _pipes(ctx, 2, w, h, n5)
    api.StandardPipeline.render_frames([c, c], scene, cam, sky, PAO, [5, 6], [1, 2])
    d.render(scene, cam, sky, PAO, frame_index=6, rand=2)
    _same(_planes(c), _planes(d), "one pipeline twice")
    # two frame sizes
    e = _pipes(ctx, 1, w, h, n5)[0]
    f_ = _pipes(ctx, 1, 96, 64, n5)[0]
    api.StandardPipeline.render_frames([e, f_], scene, cam, sky, PAO, [5, 6], [1, 2])
    g = _pipes(ctx, 1, 96, 64, n5)[0]
    g.render(scene, cam, sky, PAO, frame_index=6, rand=2)
    _same(_planes(f_), _planes(g), "two frame sizes")


def test_a_bad_frame_is_refused_before_any_frame_is_enqueued():
    w, h = 64, 48
    desc = P.small_scene(seed=4, n_models=2, n_instances=3)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=2)
    cam = P.camera_for((90.0, 60.0, -80.0))
    a = _pipes(ctx, 1, w, h, n5)[0]
    bad = api.StandardPipeline(ctx, w, h)   # no noise texture: DUST_ERR_NOT_READY (standard.rs:254)
    with pytest.raises(L.DustError) as e:
        api.StandardPipeline.render_frames([a, bad], scene, cam, sky, PAO, [1, 2], [1, 2])
    assert e.value.status == L.ERR_NOT_READY
    ctx.sync()
    assert not a.read_plane(L.PLANE_DEPTH).any()   # the first frame was not rendered either


def test_batched_frames_of_a_deep_tree():
    """the DEEP kernel variant (a 4096^3 model: hierarchy (4,4,2,2)), three frames in one launch == alone == oracle"""
    from test_configs import deep_desc
    blocks, mats, pal = deep_desc(1e-4)
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
    n5 = synth.stbn_unitvec3_cosine(layers=2)
    sky = P.sky_state()
    cams = [P.camera_for((2600.0, 1900.0, 2300.0)), P.camera_for((300.0, 200.0, -150.0)), P.camera_for((-900.0, 400.0, 700.0))]
    w, h = 160, 100
    batch = _pipes(ctx, 3, w, h, n5)
    api.StandardPipeline.render_frames(batch, scene, cams, sky, PAO, [1, 2, 3], [5, 6, 7])
    single = _pipes(ctx, 3, w, h, n5)
    for i in range(3):
        single[i].render(scene, cams[i], sky, PAO, frame_index=1 + i, rand=5 + i)
        _same(_planes(batch[i]), _planes(single[i]), f"deep frame {i}")
    g = P.render_oracle(os_, cams[1], sky, w, h, PAO, n5[2 % 2], 6)
    P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(batch[1])))


def test_batched_frames_of_a_large_scene():
    """the LARGE kernel variant (more than 256 instances: the cull's 64-wide hierarchy), four frames in one launch == alone == oracle"""
    desc = scattered_scene(1500)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    oscene = P.oracle_scene(desc)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    cams = _cams(4, eye=(180.0, 90.0, 260.0))
    w, h = 192, 108
    batch = _pipes(ctx, 4, w, h, n5)
    idx = [2, 3, 4, 5]
    api.StandardPipeline.render_frames(batch, scene, cams, sky, PAO, idx, [9, 8, 7, 6])
    single = _pipes(ctx, 4, w, h, n5)
    for i in range(4):
        single[i].render(scene, cams[i], sky, PAO, frame_index=idx[i], rand=9 - i)
        _same(_planes(batch[i]), _planes(single[i]), f"large-scene frame {i}")
    g = P.render_oracle(oscene, cams[3], sky, w, h, PAO, n5[idx[3] % 4], 6)
    P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(batch[3])))


def test_batched_frames_at_full_size_and_the_launch_is_timed_once():
    """1920 x 1080 on the castle stand-in (BASELINE configs[1]), launches of four frames, twenty times over (the cost order settles):
    the last launch's frames == the same frames alone; with timing on, pipelines[0] reports one launch per call and the others none."""
    from dust_amd import scenes as S
    W, H = 1920, 1080
    ctx = api.Context(device=0, timing=True)
    data, info = synth.castle_scene()
    desc = S.SceneDesc.from_vox(data)
    scene = S.hip_scene(ctx, desc)
    sky = S.sky_state()
    n5 = synth.stbn_unitvec3_cosine()
    eye = (122.0, 300.61, 54.45)
    cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
    batch = _pipes(ctx, 4, W, H, n5)
    f = 1
    for _ in range(20):
        idx = [f, f + 1, f + 2, f + 3]
        api.StandardPipeline.render_frames(batch, scene, cam, sky, PAO, idx, [synth.frame_rand(1, v) for v in idx])
        f += 4
    ms, n = batch[0].kernel_times(mark=True)
    assert n[0] == 20 and ms[0] > 0.0
    assert batch[1].kernel_times(mark=True)[1][0] == 0
    last = [f - 4, f - 3, f - 2, f - 1]
    alone = _pipes(ctx, 1, W, H, n5)[0]
    for i in range(4):
        alone.render(scene, cam, sky, PAO, frame_index=last[i], rand=synth.frame_rand(1, last[i]))
        _same(_planes(batch[i]), _planes(alone), f"1080p frame {i}")
    print(f"k_primary_ao_batch, 4 frames of 1080p: {ms[0] / n[0]:.4f} ms per launch = {ms[0] / n[0] / 4:.4f} ms per frame")


def test_settings_that_do_not_shape_a_launch_do_not_split_it():
    """in_flight_slots only matters with several frames in flight: pipelines that differ in it (and nothing else) still share a launch --
    seen in the timing: the first pipeline's pair brackets the one launch, the other has none. A pipeline that leaves slots free for another
    queue's kernels (reserve_blocks) has another launch geometry: frames in sequence, a launch timed on each."""
    w, h = 128, 80
    desc = P.small_scene(seed=4, n_models=3, n_instances=6)
    ctx = api.Context(device=0, timing=True)
    scene = P.hip_scene(ctx, desc)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    cam = P.camera_for((90.0, 60.0, -80.0))
    a, b = _pipes(ctx, 2, w, h, n5)
    a.configure(frames_in_flight=1, in_flight_slots="all")
    for p in (a, b):
        p.mark_kernel_times()
    api.StandardPipeline.render_frames([a, b], scene, cam, sky, PAO, [1, 2], [3, 4])
    assert a.kernel_times()[1][0] == 1 and b.kernel_times()[1][0] == 0
    b.configure(reserve_blocks=32)
    api.StandardPipeline.render_frames([a, b], scene, cam, sky, PAO, [3, 4], [5, 6])
    assert a.kernel_times()[1][0] == 1 and b.kernel_times()[1][0] == 1
    c = _pipes(ctx, 1, w, h, n5)[0]
    c.render(scene, cam, sky, PAO, frame_index=4, rand=6)
    _same(_planes(b), _planes(c), "the frame of the pipeline with reserved slots")


def _mat4(o2w):
    m = np.eye(4, dtype=np.float32)
    m[:3, :] = np.asarray(o2w, np.float32).reshape(3, 4)
    return np.ascontiguousarray(m.T).reshape(16)


@pytest.mark.parametrize("n, calls", [(4, 5), (8, 3), (11, 2), (19, 3)])
def test_frames_of_a_moving_scene_share_a_launch(n, calls):
    """castle.rs:287-291 moves an entity every frame and tlas.rs:37-65 rebuilds the TLAS in that frame's command stream: here every frame of a
    call comes with its moves (dust_hip_render_frames applies and commits them before the frame is prepared -- a scene image of its own in the
    scene's ring of 16) and the frames still share launches. Against set_transform + commit + render_frame per frame on a twin scene: every plane,
    motion vectors included, bit for bit; several calls in a row with nothing waited for in between (the ring comes round: a commit must not
    land on an image a frame that has not started yet still reads); and against the oracle."""
    w, h = 200, 120
    desc = P.small_scene(seed=6, n_models=3, n_instances=7)
    ctx = api.Context(device=0)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    scene_a, scene_b = P.hip_scene(ctx, desc), P.hip_scene(ctx, desc)
    batch, single = _pipes(ctx, n, w, h, n5), _pipes(ctx, n, w, h, n5)
    rng = np.random.default_rng(5)
    cur = [np.array(t, np.float32).reshape(3, 4).copy() for _, t in desc.instances]
    f = 1
    for _ in range(calls):
        cams = _cams(n, eye=(90.0 + f, 60.0, -80.0))
        idx = [f + i for i in range(n)]
        rnd = [synth.frame_rand(2, v) for v in idx]
        moves = []
        for i in range(n):
            ms = []
            for j in rng.choice(len(cur), size=int(rng.integers(0, 3)), replace=False):   # no, one or two instances move before this frame
                prev = _mat4(cur[j])
                cur[j] = cur[j].copy()
                cur[j][:, 3] += rng.uniform(-3.0, 3.0, 3).astype(np.float32)
                ms.append((int(j), cur[j].reshape(12).copy(), prev))
            moves.append(ms)
        api.StandardPipeline.render_frames(batch, scene_a, cams, sky, PAO, idx, rnd, moves=moves)
        for i in range(n):
            for j, xf, prev in moves[i]:
                scene_b.set_transform(j, xf, prev)
            if moves[i]:
                scene_b.commit()
            single[i].render(scene_b, cams[i], sky, PAO, frame_index=idx[i], rand=rnd[i])
        f += n
    for i in range(n):
        _same(_planes(batch[i]), _planes(single[i]), f"moving frame {i} of {n}")
    # the last frame against the oracle's scene in its final state
    final = P.SceneDesc(desc.models, desc.palette, [(m, cur[k].reshape(12)) for k, (m, _) in enumerate(desc.instances)])
    g = P.render_oracle(P.oracle_scene(final), cams[n - 1], sky, w, h, PAO, n5[idx[n - 1] % 4], rnd[n - 1])
    res = P.compare_gbuffers(g, P.read_hip_gbuffer(batch[n - 1]))
    res["motion"] = 0   # (the oracle scene built at rest has no previous transforms: motion vectors are compared against the twin above)
    P.assert_parity(res)


def test_a_refused_batched_launch_falls_back_to_a_launch_per_frame(monkeypatch):
    """The batched launch carries 8.5 KB of kernel arguments (the runtime takes 16 KB: probed). Should a runtime refuse it, the prepared frames
    are launched one by one -- DUST_HIP_DEBUG bit 32 takes that way: the same planes, moves included."""
    monkeypatch.setenv("DUST_HIP_DEBUG", "32")
    w, h = 160, 96
    desc = P.small_scene(seed=9, n_models=3, n_instances=6)
    ctx = api.Context(device=0)
    sky = P.sky_state()
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    scene_a, scene_b = P.hip_scene(ctx, desc), P.hip_scene(ctx, desc)
    batch = _pipes(ctx, 4, w, h, n5)
    cams = _cams(4)
    xf = np.array(desc.instances[1][1], np.float32).reshape(3, 4).copy()
    xf[:, 3] += 5.0
    moves = [[], [(1, xf.reshape(12), _mat4(desc.instances[1][1]))], [], []]
    api.StandardPipeline.render_frames(batch, scene_a, cams, sky, PAO, [1, 2, 3, 4], [5, 6, 7, 8], moves=moves)
    monkeypatch.delenv("DUST_HIP_DEBUG")
    single = _pipes(ctx, 4, w, h, n5)
    for i in range(4):
        for j, x, pv in moves[i]:
            scene_b.set_transform(j, x, pv)
            scene_b.commit()
        single[i].render(scene_b, cams[i], sky, PAO, frame_index=1 + i, rand=5 + i)
        _same(_planes(batch[i]), _planes(single[i]), f"fallback frame {i}")
