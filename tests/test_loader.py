"""Loader / flatten parity: product (dust_vox_load, dust_vox_flatten_model) vs the oracle restatement of
crates/vox/src/{loader,collector,geometry}.rs, plus .vox writer -> loader round trips."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_flatten_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    size = tuple(int(v) for v in rng.integers(5, 90, 3))
    xyzi = P.random_model(rng, size, fill=0.05 + 0.1 * seed)
    pal = synth.make_palette(seed)
    b0, m0 = O.model_build(xyzi, size, pal)
    b1, m1 = api.flatten_model(xyzi, size, pal)
    assert b0.tobytes() == b1.tobytes()          # 24-byte Block records, bit-exact
    assert np.array_equal(m0, m1)
    assert len(m1) == len(xyzi)
    assert int(sum(bin(int(m)).count("1") for m in b1["mask"])) == len(xyzi)
    # material_ptr is the exclusive prefix sum in block-major order (collector.rs:76-87)
    order = np.lexsort((b1["x"] >> 2, b1["y"] >> 2, b1["z"] >> 2))
    counts = np.array([bin(int(m)).count("1") for m in b1["mask"]])[order]
    assert np.array_equal(b1["material_ptr"][order], np.concatenate([[0], np.cumsum(counts)[:-1]]))


def test_flatten_edge_cases():
    pal = synth.make_palette(1)
    # empty model
    b, m = api.flatten_model(np.zeros((0, 4), np.uint8), (4, 4, 4), pal)
    assert len(b) == 0 and len(m) == 0
    # single voxel at the far corner of a maximum-size model: axis map (x, z, size.y-1-y), loader.rs:248-253
    b, m = api.flatten_model(np.array([[255, 0, 255, 9]], np.uint8), (256, 256, 256), pal)
    assert (int(b["x"][0]), int(b["y"][0]), int(b["z"][0])) == (252, 252, 252)
    bit = (3 << 4) | (3 << 2) | 3
    assert int(b["mask"][0]) == 1 << bit and m.tolist() == [9]
    # full 4^3 brick
    xs = np.array([[x, y, z, 3] for x in range(4) for y in range(4) for z in range(4)], np.uint8)
    b, m = api.flatten_model(xs, (4, 4, 4), pal)
    assert len(b) == 1 and int(b["mask"][0]) == 0xFFFFFFFFFFFFFFFF and len(m) == 64
    b0, _ = O.model_build(xs, (4, 4, 4), pal)
    assert b0.tobytes() == b.tobytes()
    # out-of-range voxel is rejected, not written out of bounds
    with pytest.raises(L.DustError):
        api.flatten_model(np.array([[9, 0, 0, 1]], np.uint8), (4, 4, 4), pal)


def test_vox_roundtrip_and_transforms():
    rng = np.random.default_rng(11)
    pal = synth.make_palette(3)
    sizes = [(20, 31, 12), (8, 8, 8)]
    models = []
    for sz in sizes:
        x = P.random_model(rng, sz, fill=0.2, blobs=1)
        x[:, 3] = x[:, 3] % 254 + 1  # files store 1-based colour indices
        models.append((sz, x))
    inst = [(0, (10, 20, 30), synth.ROT_IDENTITY), (1, (-5, 7, 9), synth.ROT_Z90), (0, (0, 0, 0), synth.ROT_MIRROR_X)]
    data = synth.write_vox(models, inst, pal)
    vs = api.VoxScene(data)
    assert vs.n_models == 2 and vs.n_instances == 3
    assert np.array_equal(vs.palette, pal)
    for i, (sz, x) in enumerate(models):
        x0 = x.copy()
        x0[:, 3] -= 1  # dot_vox: i = file index - 1
        b_ref, m_ref = O.model_build(x0, sz, pal)
        b, m = vs.model_data(i)
        assert b.tobytes() == b_ref.tobytes() and np.array_equal(m, m_ref)
    # instance 0: identity rotation; translation.xzy with z negated, minus half extent, plus odd-size offset
    # (loader.rs:178-204): size (20,31,12) -> engine size (20,12,31); offset (0, 0, -0.5) for odd file-y
    m0 = vs.instances[0][1].reshape(3, 4)
    assert np.array_equal(m0[:, :3], np.eye(3, dtype=np.float32))
    assert m0[:, 3].tolist() == [10 - 10.0, 30 - 6.0, -20 - 15.5 - 0.5]
    # instance 1: rotation about the file z axis (engine y); a rigid rotation keeps det = +1
    m1 = vs.instances[1][1].reshape(3, 4)
    assert round(float(np.linalg.det(m1[:, :3]))) == 1
    assert np.array_equal(np.abs(m1[:, :3]).sum(axis=0), np.ones(3))
    # instance 2: mirrored, det = -1
    m2 = vs.instances[2][1].reshape(3, 4)
    assert round(float(np.linalg.det(m2[:, :3]))) == -1
    # every transform maps the model's voxel box onto a box centred at the (axis-swapped) translation
    for (mid, t, _), (_, o2w) in zip(inst, vs.instances):
        sz = sizes[mid]
        ext = np.array([sz[0], sz[2], sz[1]], np.float64)
        corners = np.array([[(c >> k) & 1 for k in range(3)] for c in range(8)]) * ext
        w = corners @ o2w.reshape(3, 4)[:, :3].T.astype(np.float64) + o2w.reshape(3, 4)[:, 3]
        centre = (w.min(axis=0) + w.max(axis=0)) / 2
        assert np.all(np.abs(centre - np.array([t[0], t[2], -t[1]])) <= 0.5)


def test_vox_groups_compose():
    pal = synth.make_palette(2)
    x = np.array([[0, 0, 0, 1], [1, 1, 1, 2]], np.uint8)
    models = [((2, 2, 2), x)]
    inst = [(0, (4, 0, 0), synth.ROT_IDENTITY), (0, (4, 0, 0), synth.ROT_IDENTITY)]
    flat = api.VoxScene(synth.write_vox(models, inst, pal))
    grouped = api.VoxScene(synth.write_vox(models, inst, pal, groups=[((100, 0, 0), synth.ROT_Z180, [1])]))
    a = flat.instances[0][1].reshape(3, 4)
    # the group (first in file order) rotates its child by 180 degrees about engine y and shifts it by +100 x
    g = [m for m in grouped.instances][0][1].reshape(3, 4)
    assert np.array_equal(g[:, :3], np.diag([-1.0, 1.0, -1.0]).astype(np.float32))
    centre_a = a[:, :3] @ np.ones(3) + a[:, 3]
    centre_g = g[:, :3] @ np.ones(3) + g[:, 3]
    assert centre_a.tolist() == [4.0, 0.0, 0.0]
    assert centre_g.tolist() == [96.0, 0.0, 0.0]


def test_vox_without_scene_graph_and_default_palette():
    x = np.array([[1, 2, 3, 5]], np.uint8)
    vs = api.VoxScene(synth.write_vox([((8, 8, 8), x)], [], None, scene_graph=False))
    assert vs.n_instances == 1
    assert np.array_equal(vs.instances[0][1].reshape(3, 4), np.eye(3, 4, dtype=np.float32))  # loader.rs:68-84
    # default palette: entry 0 is white, the cube steps by 0x33, the ramps follow
    assert vs.palette[0].tolist() == [255, 255, 255, 255]
    assert vs.palette[1].tolist() == [255, 255, 204, 255]
    assert vs.palette[6].tolist() == [255, 204, 255, 255]
    assert vs.palette[215].tolist() == [238, 0, 0, 255]
    assert vs.palette[254].tolist() == [17, 17, 17, 255]


def test_vox_errors():
    with pytest.raises(L.DustError) as e:
        api.VoxScene(b"NOPE" + b"\0" * 32)
    assert e.value.status == L.ERR_PARSE
    good = synth.teapot_scene(24)
    with pytest.raises(L.DustError) as e:
        api.VoxScene(good[: len(good) // 2])
    assert e.value.status == L.ERR_PARSE
    # a frame count the chunk does not hold
    import struct
    idx = good.index(b"nTRN")
    bad = bytearray(good)
    off = idx + 12 + 4 + 4 + 4 + 4 + 4  # header, node id, empty dict, child, reserved, layer
    assert struct.unpack_from("<I", bad, off)[0] == 1
    struct.pack_into("<I", bad, off, 2)
    with pytest.raises(L.DustError) as e:
        api.VoxScene(bytes(bad))
    assert e.value.status == L.ERR_PARSE
    # colour index 0 is not a colour: dot_vox saturates `i - 1` at 0, nothing wraps to 255
    xyzi = np.array([[1, 1, 1, 0], [2, 1, 1, 7]], np.uint8)
    vs = api.VoxScene(synth.write_vox([((4, 4, 4), xyzi)], [(0, (0, 0, 0), synth.ROT_IDENTITY)], synth.make_palette(1)))
    _, mats = vs.model_data(0)
    assert sorted(mats.tolist()) == [0, 6]


def test_vox_animation_frames():
    """MagicaVoxel animations -- transform nodes with several keyframes, shape nodes with several models -- are
    unimplemented!() in the reference (loader.rs:103-105,149-151). Here the entry in force at the requested frame is used:
    each frame of the animated file must load exactly like the plain file that spells that frame out."""
    rng = np.random.default_rng(5)
    pal = synth.make_palette(2)
    models = []
    for sz in ((10, 12, 9), (6, 6, 6), (7, 5, 11)):
        x = P.random_model(rng, sz, fill=0.3, blobs=0)
        x[:, 3] = x[:, 3] % 254 + 1
        models.append((sz, x))
    static = [(0, (3, 4, 5), synth.ROT_IDENTITY), (1, (-20, 0, 7), synth.ROT_Z90), (2, (9, 9, 9), synth.ROT_IDENTITY)]
    keys1 = [(0, (-20, 0, 7), synth.ROT_Z90), (4, (-10, 2, 7), synth.ROT_IDENTITY), (9, (0, 4, 7), synth.ROT_MIRROR_X)]
    shapes2 = [(0, 2), (5, 1), (8, 0)]
    anim = {1: {"frames": keys1}, 2: {"models": shapes2}}
    data = synth.write_vox(models, static, pal, anim=anim, groups=[((1, 2, 3), synth.ROT_IDENTITY, [1])])

    def at(frame):
        k = max((f, t, r) for f, t, r in keys1 if f <= frame)
        m = max((f, mid) for f, mid in shapes2 if f <= frame)[1]
        return [static[0], (1, k[1], k[2]), (m, (9, 9, 9), synth.ROT_IDENTITY)]

    for frame in (0, 1, 4, 5, 8, 9, 100):
        got = api.VoxScene(data, frame=frame)
        want = api.VoxScene(synth.write_vox(models, at(frame), pal, groups=[((1, 2, 3), synth.ROT_IDENTITY, [1])]))
        assert got.n_instances == want.n_instances == 3
        key = lambda s_: sorted((m, tuple(np.round(t, 4).tolist())) for m, t in s_.instances)
        assert key(got) == key(want), frame
        for i in range(3):
            if want.model_info(i).used:
                assert got.model_data(i)[0].tobytes() == want.model_data(i)[0].tobytes()
    assert api.VoxScene(data).n_instances == 3  # dust_vox_load == frame 0


def test_castle_standin_small():
    data, info = synth.castle_scene(scale=0.12)
    desc = P.SceneDesc.from_vox(data)
    assert info["n_models"] >= 90 and info["n_instances"] >= 140
    assert len(desc.instances) == info["n_instances"]
    dets = [round(float(np.linalg.det(t.reshape(3, 4)[:, :3]))) for _, t in desc.instances]
    assert -1 in dets and 1 in dets
