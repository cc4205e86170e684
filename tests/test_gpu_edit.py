"""Device-side voxel edits (dust_hip_model_set_voxels / get_voxels; VoxGeometry::set/get, vox/src/geometry.rs:180-185):
after any sequence of edit batches the model's device arrays must be byte for byte what a host rebuild of the same voxels
uploads (dust_vox_flatten_model -> dust_hip_model_create), and frames rendered from the two must be identical."""
import time

import numpy as np
import pytest

import parity_util as P
from dust_amd import _lib as L, api, synth

pytestmark = pytest.mark.gpu


def host_model(vox, pal):
    """voxel dict {(x, y, z) tree coords: palette index} -> (blocks, materials) through the product's host flatten"""
    if not vox:
        return api.flatten_model(np.zeros((0, 4), np.uint8), (256, 256, 256), pal)
    k = np.array(list(vox.keys()), np.int64)
    v = np.array(list(vox.values()), np.int64)
    xyzi = np.stack([k[:, 0], 255 - k[:, 2], k[:, 1], v], axis=1).astype(np.uint8)  # loader.rs:248-253: engine (x, z, size.y-1-y)
    return api.flatten_model(xyzi, (256, 256, 256), pal)


def start_voxels(rng, n=6000):
    vox = {}
    for c in rng.integers(0, 3, (n, 3)) * 0 + rng.integers(20, 120, (n, 3)):
        vox[tuple(int(t) for t in c)] = int(rng.integers(0, 255))
    for x in range(40, 56):  # a solid slab: full bricks
        for y in range(40, 44):
            for z in range(40, 56):
                vox[(x, y, z)] = 7
    return vox


def test_edits_equal_a_host_rebuild():
    rng = np.random.default_rng(21)
    pal = synth.make_palette(4)
    ctx = api.Context(device=0)
    vox = start_voxels(rng)
    b0, m0 = host_model(vox, pal)
    model = api.Model(ctx, b0, m0, pal)
    got_b, got_m = model.read()
    assert got_b.tobytes() == b0.tobytes() and got_m.tobytes() == m0.tobytes()
    times = []
    for batch in range(7):
        xyz, val = [], []
        keys = list(vox.keys())
        if batch == 0:    # recolour existing voxels and add neighbours inside existing bricks
            for k in rng.choice(len(keys), 400, replace=False):
                xyz.append(keys[k]); val.append(int(rng.integers(0, 255)))
        elif batch == 1:  # new bricks, new 16^3 cells, far corners of the tree
            for c in rng.integers(0, 256, (500, 3)):
                xyz.append(tuple(int(t) for t in c)); val.append(int(rng.integers(0, 255)))
            xyz += [(0, 0, 0), (255, 255, 255), (255, 0, 128)]; val += [1, 2, 3]
        elif batch == 2:  # remove voxels, some whole bricks with them
            for k in rng.choice(len(keys), 1500, replace=False):
                xyz.append(keys[k]); val.append(-1)
            for x in range(40, 48):
                for y in range(40, 44):
                    for z in range(40, 48):
                        xyz.append((x, y, z)); val.append(-1)
        elif batch == 3:  # the same voxel several times in one batch: the last entry wins; clearing an empty voxel is a no-op
            xyz += [(10, 10, 10), (10, 10, 10), (10, 10, 10), (11, 10, 10), (11, 10, 10), (200, 3, 77)]
            val += [5, -1, 9, 4, -1, -1]
        elif batch == 4:  # a large batch
            for c in rng.integers(60, 200, (20000, 3)):
                xyz.append(tuple(int(t) for t in c)); val.append(int(rng.integers(-1, 255)))
        elif batch == 5:  # empty the model completely
            for k in keys:
                xyz.append(k); val.append(-1)
        else:             # and fill something back in
            for c in rng.integers(100, 140, (300, 3)):
                xyz.append(tuple(int(t) for t in c)); val.append(int(rng.integers(0, 255)))
        for c, v in zip(xyz, val):
            if v < 0:
                vox.pop(tuple(c), None)
            else:
                vox[tuple(c)] = v
        t0 = time.perf_counter()
        model.set_voxels(np.array(xyz, np.uint32), np.array(val, np.int32))
        times.append((time.perf_counter() - t0) * 1e3)
        want_b, want_m = host_model(vox, pal)
        got_b, got_m = model.read()
        assert len(got_b) == len(want_b) and len(got_m) == len(want_m), batch
        assert got_b.tobytes() == want_b.tobytes(), f"batch {batch}: Block records differ"
        assert got_m.tobytes() == want_m.tobytes(), f"batch {batch}: material stream differs"
        probe = np.array(list(vox.keys())[:200] + [(1, 2, 3), (250, 250, 250)], np.uint32).reshape(-1, 3)
        assert model.get_voxels(probe).tolist() == [vox.get(tuple(int(t) for t in c), -1) for c in probe]
    assert len(vox) > 100
    print("edit batches (ms, first one includes the switch to the editable form):", [round(t, 2) for t in times])
    assert max(times[1:]) < 50.0  # milliseconds, not the seconds of a host rebuild + upload


def test_frames_after_edits_equal_frames_of_a_rebuilt_model():
    rng = np.random.default_rng(22)
    pal = synth.make_palette(6)
    ctx = api.Context(device=0)
    vox = start_voxels(rng, 9000)
    b0, m0 = host_model(vox, pal)
    edited = api.Model(ctx, b0, m0, pal)
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = (-70.0, -70.0, -70.0)
    scene = api.Scene(ctx)
    scene.add_instance(edited, xf.reshape(12))
    scene.add_instance(edited, (np.array([[0, 0, 1, 40], [0, 1, 0, -60], [-1, 0, 0, 90]], np.float32)).reshape(12))
    scene.commit()
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    cam, sky = P.camera_for((150.0, 120.0, 160.0)), P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED

    def frames(sc):
        pipe = api.StandardPipeline(ctx, 320, 200)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(1 << 16, 8192)
        for f in (1, 2, 3):
            pipe.render(sc, cam, sky, passes, frame_index=f, rand=synth.frame_rand(3, f))
        h, sp = pipe.read_gi()
        return P.read_hip_gbuffer(pipe), h, sp.view(np.uint32).copy()

    frames(scene)
    xyz = rng.integers(15, 125, (4000, 3))
    val = rng.integers(-1, 255, 4000)
    for c, v in zip(xyz, val):
        if v < 0:
            vox.pop(tuple(int(t) for t in c), None)
        else:
            vox[tuple(int(t) for t in c)] = int(v)
    edited.set_voxels(xyz.astype(np.uint32), val.astype(np.int32))
    with pytest.raises(L.DustError) as e:  # bounds and the staged root may have changed: the scene must be committed again
        frames(scene)
    assert e.value.status == L.ERR_NOT_READY
    scene.commit()
    got = frames(scene)
    fresh = api.Model(ctx, *host_model(vox, pal), pal)
    ref_scene = api.Scene(ctx)
    ref_scene.add_instance(fresh, xf.reshape(12))
    ref_scene.add_instance(fresh, (np.array([[0, 0, 1, 40], [0, 1, 0, -60], [-1, 0, 0, 90]], np.float32)).reshape(12))
    ref_scene.commit()
    want = frames(ref_scene)
    for k in want[0]:
        assert want[0][k].tobytes() == got[0][k].tobytes(), k
    assert np.array_equal(want[1], got[1]) and np.array_equal(want[2], got[2])
    assert np.isfinite(want[0]["depth"]).mean() > 0.05


def test_edit_argument_checks():
    ctx = api.Context(device=0)
    pal = synth.make_palette(1)
    model = api.Model(ctx, *host_model({(1, 1, 1): 3}, pal), pal)
    with pytest.raises(L.DustError):
        model.set_voxels(np.array([[256, 0, 0]], np.uint32), np.array([1], np.int32))
    with pytest.raises(L.DustError):
        model.set_voxels(np.array([[0, 0, 0]], np.uint32), np.array([255], np.int32))
    blocks, mats = synth.procedural_deep_blocks(occupancy=2e-6, sample=True)
    deep = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    with pytest.raises(L.DustError) as e:
        deep.set_voxels(np.array([[0, 0, 0]], np.uint32), np.array([1], np.int32))
    assert e.value.status == L.ERR_UNSUPPORTED
