"""Hash-fed GI passes (final gather + surfel pass + spatial hash) on the GPU against the oracle, frame after
frame, with the deterministic apply order (DUST_PASS_GI_ORDERED). Integer state (fingerprints, counts, LRU stamps,
surfel pool) is bit-exact; LogLuv radiance words go through log2/pow on both sides and may differ by a quantisation
step, illuminance by <= 1e-3 relative L2 (north_star tolerance)."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth

pytestmark = pytest.mark.gpu


def compare_gi(gi, pipe):
    oh, op = gi.hash(), gi.pool()
    hh, hp = pipe.read_gi()
    assert np.array_equal(oh["fingerprint"], hh[:, 0]), "fingerprints differ"
    assert np.array_equal(oh["last_accessed_frame"], hh[:, 2] & 0xFFFF), "LRU stamps differ"
    assert np.array_equal(oh["sample_count"], hh[:, 2] >> 16), "sample counts differ"
    used = oh["fingerprint"] != 0
    le_o, le_h = (oh["radiance"][used] >> 18).astype(np.int64), (hh[:, 1][used] >> 18).astype(np.int64)
    assert np.abs(le_o - le_h).max(initial=0) <= 2, "LogLuv luminance differs by more than 2 steps (0.35 %)"
    uv_o = np.stack([(oh["radiance"][used] >> 9) & 511, oh["radiance"][used] & 511]).astype(np.int64)
    uv_h = np.stack([(hh[:, 1][used] >> 9) & 511, hh[:, 1][used] & 511]).astype(np.int64)
    assert np.abs(uv_o - uv_h).max(initial=0) <= 1
    assert np.array_equal(op["direction"], hp["direction"]), "surfel pool faces differ"
    v = op["direction"] < 6
    assert np.array_equal(op["pos"][v].view(np.uint32), hp["pos"][v].view(np.uint32)), "surfel pool positions differ"
    return int(used.sum()), int(v.sum())


@pytest.mark.parametrize("capacity,pool", [(1 << 14, 2048), (97, 777), (16, 333)])
def test_gi_sequence_matches_oracle(capacity, pool):
    desc = P.small_scene(seed=5, n_models=2, n_instances=4, size=(28, 28, 28))
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    oscene = P.oracle_scene(desc)
    sky = P.sky_state()
    cam = P.camera_for((80.0, 60.0, 90.0))
    w, h = 96, 64
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(capacity, pool)
    gi = O.GI(capacity, pool)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    used = valid = 0
    for f in range(1, 6):
        rnd = synth.frame_rand(1, f)
        pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f)
        hip = P.read_hip_gbuffer(pipe)
        P.assert_parity(P.compare_gbuffers(g, hip))
        used, valid = compare_gi(gi, pipe)
    assert valid > 20
    assert used > 20 or (capacity == 16 and used >= 12)   # the 16-entry table lives on probing + LRU eviction


def test_gi_racy_mode_is_statistically_close():
    """Default (concurrent) apply, as the reference's shaders do it: same keys get claimed, radiance agrees on average."""
    desc = P.small_scene(seed=6, n_models=2, n_instances=4, size=(28, 28, 28))
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky = P.sky_state()
    cam = P.camera_for((80.0, 60.0, 90.0))
    w, h = 96, 64
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    outs = []
    for flags in (L.PASS_GI_ORDERED, 0):
        pipe = api.StandardPipeline(ctx, w, h)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(1 << 16, 4096)
        for f in range(1, 5):
            pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | flags,
                        frame_index=f, rand=synth.frame_rand(1, f))
        outs.append((pipe.read_gi(), P.half_to_float(pipe.read_plane(L.PLANE_ILLUMINANCE))))
    (h_ord, _), ill_ord = outs[0]
    (h_racy, _), ill_racy = outs[1]
    fp_o, fp_r = set(h_ord[:, 0][h_ord[:, 0] != 0].tolist()), set(h_racy[:, 0][h_racy[:, 0] != 0].tolist())
    assert len(fp_o & fp_r) >= 0.9 * len(fp_o)
    a, b = ill_ord[..., :3], ill_racy[..., :3]
    assert abs(a.mean() - b.mean()) <= 0.05 * abs(a.mean()) + 1e-6


def test_surfel_position_sort_only_regroups(monkeypatch):
    """The surfel pass traces the pool in position order (k_surfel_keys + radix sort); that changes which surfels share a
    wavefront, never what a surfel computes: with the ordered apply the GI state must equal the pool-order run bit for bit."""
    desc = P.small_scene(seed=9, n_models=3, n_instances=6, size=(28, 28, 28))
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    sky, cam = P.sky_state(), P.camera_for((80.0, 60.0, 90.0))
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    states = []
    for no_sort in (False, True):
        if no_sort:
            monkeypatch.setenv("DUST_HIP_NO_SURFEL_SORT", "1")
        else:
            monkeypatch.delenv("DUST_HIP_NO_SURFEL_SORT", raising=False)
        pipe = api.StandardPipeline(ctx, 128, 80)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(1 << 14, 4096)
        for f in range(1, 5):
            pipe.render(scene, cam, sky, passes, frame_index=f, rand=synth.frame_rand(2, f))
        h, s = pipe.read_gi()
        states.append((h, s.view(np.uint32).copy(), pipe.read_plane(L.PLANE_ILLUMINANCE)))
    monkeypatch.delenv("DUST_HIP_NO_SURFEL_SORT", raising=False)
    assert (states[0][0][:, 0] != 0).sum() > 50
    for x, y in zip(states[0], states[1]):
        assert np.array_equal(x, y)


_SWITCHES = ("DUST_HIP_DEBUG", "DUST_HIP_NO_GATHER_ORDER", "DUST_HIP_NO_SURFEL_SORT", "DUST_HIP_NO_TILE_ORDER", "DUST_HIP_NO_LDS_BOXES",
             "DUST_HIP_BLOCK", "DUST_HIP_BLOCKS_PER_CU", "DUST_HIP_RAY_STREAM", "DUST_HIP_NO_STREAM_LDS",
             "DUST_HIP_STREAM_REFILL")


def _castle_gi_states(monkeypatch, settings, frames=3):
    """GI state + radiance plane after `frames` frames of the small castle (overlapping, lattice-aligned instances:
    equal-t ties between bricks of different instances are the rule there, not the exception)."""
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    s = 0.15
    sky, cam = P.sky_state(), P.camera_for((122.0 * s, 300.61 * s, 54.45 * s))
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    out = []
    for env in settings:
        for k in _SWITCHES:
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        pipe = api.StandardPipeline(ctx, 192, 104)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(1 << 14, 776)
        for f in range(1, frames + 1):
            pipe.render(scene, cam, sky, passes, frame_index=f, rand=synth.frame_rand(7, f))
        h, sp = pipe.read_gi()
        out.append((h, sp.view(np.uint32).copy(), pipe.read_plane(L.PLANE_ILLUMINANCE)))
    for k in _SWITCHES:
        monkeypatch.delenv(k, raising=False)
    return desc, cam, sky, n0, n5, out


def test_gi_does_not_depend_on_visiting_order_or_grouping(monkeypatch):
    """Closest hit with the lower (instance, block) on equal t is a property of the ray, not of the order instances are
    visited in (sorted candidate list vs index order, DUST_HIP_DEBUG bit 4; every lane on its own instance vs the whole
    wave on one, bit 8) nor of which rays share a wavefront
    (octant-ordered gather packets, position-ordered surfels), which wave traces which tile when (cost-ordered hand-out), where
    the cull reads its boxes from, the launch shape, or whether the GI rays run a packet at a time or as ray streams
    (gi_path = streams, spelled DUST_HIP_RAY_STREAM for the Python shim: every ray finds its instances in the top-level grid and is
    walked on a lane of its own).
    Caught a build whose out-of-line neighbour visit passed the
    hit record through the stack and then resolved such ties differently."""
    _, _, _, _, _, st = _castle_gi_states(monkeypatch, [{}, {"DUST_HIP_DEBUG": "4"}, {"DUST_HIP_DEBUG": "8"}, {"DUST_HIP_NO_GATHER_ORDER": "1"},
                                                       {"DUST_HIP_NO_GATHER_ORDER": "1", "DUST_HIP_NO_SURFEL_SORT": "1", "DUST_HIP_DEBUG": "4"},
                                                       {"DUST_HIP_NO_TILE_ORDER": "1", "DUST_HIP_NO_LDS_BOXES": "1"},
                                                       {"DUST_HIP_BLOCK": "256", "DUST_HIP_BLOCKS_PER_CU": "1"},
                                                       # the passes as ray streams (gi.hip: binned per ray over the top-level grid, one ray per lane):
                                                       # top-level data in LDS / in memory, lanes refilled one by one / only when all are done
                                                       {"DUST_HIP_RAY_STREAM": "1"}, {"DUST_HIP_RAY_STREAM": "1", "DUST_HIP_NO_STREAM_LDS": "1", "DUST_HIP_STREAM_REFILL": "1"},
                                                       {"DUST_HIP_RAY_STREAM": "1", "DUST_HIP_STREAM_REFILL": "64", "DUST_HIP_NO_SURFEL_SORT": "1"}],
                                          frames=5)
    assert (st[0][0][:, 0] != 0).sum() > 50
    for other in st[1:]:
        for x, y in zip(st[0], other):
            assert np.array_equal(x, y)


def test_gi_does_not_depend_on_which_roots_fit_in_lds():
    """Root nodes staged in LDS or read from memory is a per-model property; with per-lane instance visits the lanes of
    one wavefront mix both. Same planes and GI state when only the first 20 of the castle's roots are staged."""
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    s = 0.15
    sky, cam = P.sky_state(), P.camera_for((122.0 * s, 300.61 * s, 54.45 * s))
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    out = []
    for lds in (0, 20 * 640):
        ctx = api.Context(device=0, lds_root_bytes=lds)
        scene = P.hip_scene(ctx, desc)
        pipe = api.StandardPipeline(ctx, 192, 104)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(1 << 14, 776)
        for f in (1, 2):
            pipe.render(scene, cam, sky, passes, frame_index=f, rand=synth.frame_rand(7, f))
        h, sp = pipe.read_gi()
        out.append([h, sp.view(np.uint32).copy()] + [pipe.read_plane(pl) for pl in (L.PLANE_ILLUMINANCE, L.PLANE_DEPTH, L.PLANE_VOXEL_ID)])
    assert len(desc.models) > 20
    for x, y in zip(*out):
        assert np.array_equal(x, y)


def test_castle_gi_matches_oracle(monkeypatch):
    desc, cam, sky, n0, n5, st = _castle_gi_states(monkeypatch, [{}], frames=2)
    oscene = P.oracle_scene(desc)
    gi = O.GI(1 << 14, 776)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    for f in (1, 2):
        g = P.render_oracle(oscene, cam, sky, 192, 104, passes, n5[f % 4], synth.frame_rand(7, f), noise0=n0[f % 4], gi=gi, frame_index=f)
    oh, op = gi.hash(), gi.pool()
    h, sp, ill = st[0]
    assert np.array_equal(oh["fingerprint"], h[:, 0]) and np.array_equal(oh["sample_count"], h[:, 2] >> 16)
    assert np.array_equal(op["direction"], sp.reshape(-1, 4)[:, 3])
    a, b = P.half_to_float(g.illuminance)[..., :3], P.half_to_float(ill)[..., :3]
    assert np.sqrt(((a - b) ** 2).sum()) / np.sqrt((a ** 2).sum()) <= 1e-3
    assert np.array_equal(g.illuminance[..., 3], ill[..., 3])   # hit distances: bit-exact


def test_gi_on_mixed_two_and_three_level_trees_matches_oracle():
    """A sparse 4096^3 model (root -> level-2 nodes in memory -> mid nodes) and ordinary 256^3 models in one scene: with
    per-lane instance visits the lanes of one wavefront walk both kinds of hierarchy at the same time. All five passes,
    three frames, against the oracle."""
    blocks, mats = synth.procedural_deep_blocks(occupancy=3e-5, sample=True)
    pal = synth.make_palette(5)
    small = P.small_scene(seed=31, n_models=2, n_instances=5, size=(40, 36, 44))
    ctx = api.Context(device=0)
    deep = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    models = [api.Model(ctx, b, m, pal) for b, m in small.models]
    scene, oscene = api.Scene(ctx), O.Scene()
    oscene.add_model(blocks, mats, pal, extent=4096)
    for b, m in small.models:
        oscene.add_model(b, m, pal)
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = (-2048.0, -2048.0, -2048.0)
    scene.add_instance(deep, xf.reshape(12))
    oscene.add_instance(0, xf.reshape(12))
    for mid, t in small.instances:
        scene.add_instance(models[mid], t)
        oscene.add_instance(1 + mid, t)
    scene.commit()
    oscene.commit()
    sky, cam = P.sky_state(), P.camera_for((150.0, 110.0, 170.0))
    w, h = 128, 80
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 14, 1024)
    gi = O.GI(1 << 14, 1024)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    for f in range(1, 4):
        rnd = synth.frame_rand(2, f)
        pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f)
        hip = P.read_hip_gbuffer(pipe)
        P.assert_parity(P.compare_gbuffers(g, hip))
        used, valid = compare_gi(gi, pipe)
    ids = hip["voxel_id"][np.isfinite(hip["depth"])] & 0xFFFF
    assert (ids == 0).any() and (ids != 0).any()   # both kinds of model are on screen
    assert used > 20 and valid > 20


def test_candidate_list_overflow_matches_oracle():
    """More instances in a packet's way than its candidate list holds (kMaxCand = 160): the packet falls back to walking
    every instance box in index order. 220 small overlapping instances stacked in front of the camera, all five passes."""
    rng = np.random.default_rng(77)
    pal = synth.make_palette(9)
    xyzi = P.random_model(rng, (12, 12, 12))
    model_data = api.flatten_model(xyzi, (12, 12, 12), pal)
    ctx = api.Context(device=0)
    model = api.Model(ctx, model_data[0], model_data[1], pal)
    scene, oscene = api.Scene(ctx), O.Scene()
    oscene.add_model(model_data[0], model_data[1], pal)
    n = 220
    for i in range(n):
        t = np.eye(3, 4, dtype=np.float32)
        t[:, 3] = (float(rng.integers(-10, 10)), float(rng.integers(-10, 10)), float(-4 * (i % 55)))   # a deep stack along the view axis
        scene.add_instance(model, t.reshape(12))
        oscene.add_instance(0, t.reshape(12))
    scene.commit()
    oscene.commit()
    sky, cam = P.sky_state(), P.camera_for((3.0, 5.0, 40.0), target=(0.0, 0.0, -100.0))   # looking down the stack: every
    w, h = 64, 40                                                                           # central packet's bundle meets > 160 boxes
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 12, 512)
    gi = O.GI(1 << 12, 512)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    for f in range(1, 3):
        rnd = synth.frame_rand(4, f)
        pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED | L.PASS_COUNT_STATS, frame_index=f, rand=rnd)
        g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f)
        P.assert_parity(P.compare_gbuffers(g, P.read_hip_gbuffer(pipe)))
        compare_gi(gi, pipe)
    st = pipe.pass_stats(0)
    assert st.hits > 100 and st.instances_tested > st.hits   # the stack is on screen and rays cross several of its boxes


def test_clustered_apply_equals_serial_apply(monkeypatch):
    """DUST_PASS_GI_ORDERED applies the surfel pass's inserts in parallel over independent probe-window clusters; the serial
    one-wavefront loop (DUST_HIP_DEBUG bit 16) is its definition: hash, pool and radiance plane must agree bit for bit, on a
    crowded table (every window collides) and on a sparse one."""
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    s = 0.15
    sky, cam = P.sky_state(), P.camera_for((122.0 * s, 300.61 * s, 54.45 * s))
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    for capacity, pool in ((257, 4096), (1 << 20, 20000)):
        states = []
        for dbg in (None, "16"):
            monkeypatch.delenv("DUST_HIP_DEBUG", raising=False)
            if dbg:
                monkeypatch.setenv("DUST_HIP_DEBUG", dbg)
            pipe = api.StandardPipeline(ctx, 256, 144)
            pipe.set_noise(0, n0)
            pipe.set_noise(5, n5)
            pipe.configure_gi(capacity, pool)
            for f in range(1, 5):
                pipe.render(scene, cam, sky, passes, frame_index=f, rand=synth.frame_rand(9, f))
            h, sp = pipe.read_gi()
            states.append((h, sp.view(np.uint32).copy(), pipe.read_plane(L.PLANE_ILLUMINANCE)))
        monkeypatch.delenv("DUST_HIP_DEBUG", raising=False)
        assert (states[0][0][:, 0] != 0).sum() > 100
        for x, y in zip(*states):
            assert np.array_equal(x, y), capacity
