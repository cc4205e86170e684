"""Handle lifetimes across the C ABI (include/dust_hip.h "Ownership"): device-side handles are reference-counted inside
the library, so *_destroy may come in any order -- which is what a garbage collector does to them. Round 2's driver run died
with SIGSEGV exactly there: tests that catch a DustError (`pytest.raises(...) as e`) leave their context, models, scene and
pipeline in a reference cycle (frame <-> traceback), Python's cycle collector finalises such objects in CREATION order --
the context first --, and the models' destructors then walked a freed context. Also: asynchronous scene commits between
frames (tlas.rs:37-65), and saving / restoring the GI state."""
import gc
import time

import numpy as np
import pytest

import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth

pytestmark = pytest.mark.gpu

GI = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED


def _setup(seed=3, w=96, h=64):
    ctx = api.Context(device=0)
    desc = P.small_scene(seed=seed, n_models=3, n_instances=6)
    models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
    scene = api.Scene(ctx)
    for mid, t in desc.instances:
        scene.add_instance(models[mid], t)
    scene.commit()
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, synth.stbn_scalar(layers=4))
    pipe.set_noise(5, synth.stbn_unitvec3_cosine(layers=4))
    pipe.configure_gi(4093, 777)
    return ctx, desc, models, scene, pipe


def _frames(pipe, scene, cam, sky, first, last):
    for f in range(first, last + 1):
        pipe.render(scene, cam, sky, GI, frame_index=f, rand=synth.frame_rand(11, f))


def _state(pipe):
    h, sp = pipe.read_gi()
    return [h, sp.view(np.uint32).copy()] + [pipe.read_plane(pl) for pl in (L.PLANE_ILLUMINANCE, L.PLANE_DEPTH, L.PLANE_VOXEL_ID)]


def _destroy(obj):
    obj.__del__()  # the wrapper's destructor: calls dust_hip_*_destroy and forgets the handle


@pytest.mark.parametrize("order", ["context first", "models first", "scene, context, pipeline, models", "reverse of creation"])
def test_handles_may_be_destroyed_in_any_order(order):
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    ctx, desc, models, scene, pipe = _setup()
    _frames(pipe, scene, cam, sky, 1, 2)
    want = _state(pipe)
    _frames(pipe, scene, cam, sky, 3, 3)  # a frame in flight when the handles start to go
    if order == "context first":
        seq = [ctx] + models + [scene, pipe]
    elif order == "models first":
        seq = models + [ctx, pipe, scene]
    elif order == "scene, context, pipeline, models":
        seq = [scene, ctx, pipe] + models
    else:
        seq = [pipe, scene] + models[::-1] + [ctx]
    for o in seq:
        _destroy(o)
    # ... and the library is as good as new afterwards
    ctx2, _, models2, scene2, pipe2 = _setup()
    _frames(pipe2, scene2, cam, sky, 1, 2)
    for x, y in zip(want, _state(pipe2)):
        assert np.array_equal(x, y)


def test_objects_keep_what_they_use_alive():
    """A scene renders after the caller gave up its models and its context; a pipeline after its context."""
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    ctx, desc, models, scene, pipe = _setup()
    _frames(pipe, scene, cam, sky, 1, 3)
    want = _state(pipe)
    ctx, desc, models, scene, pipe = _setup()
    for m in models:
        _destroy(m)
    _destroy(ctx)
    _frames(pipe, scene, cam, sky, 1, 3)
    for x, y in zip(want, _state(pipe)):
        assert np.array_equal(x, y)


def test_cycle_collector_order_is_survived():
    """What `with pytest.raises(...) as e:` does to a test's locals, made explicit: context, models, scene and pipeline in one
    reference cycle, collected by gc -- finalisers run in creation order, the context's first."""
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    for _ in range(4):
        ctx, desc, models, scene, pipe = _setup()
        _frames(pipe, scene, cam, sky, 1, 2)
        cycle = [ctx, models, scene, pipe]
        cycle.append(cycle)
        del ctx, models, scene, pipe, cycle
        gc.collect()
    ctx, desc, models, scene, pipe = _setup()
    _frames(pipe, scene, cam, sky, 1, 1)
    assert np.isfinite(pipe.read_plane(L.PLANE_DEPTH)).any()


def test_commit_between_frames_changes_nothing_and_does_not_wait():
    """dust_hip_scene_commit is one stream-ordered copy from pinned memory: committing an unchanged scene between frames leaves
    every result as it was, a moved instance moves (and back), and the call costs microseconds of host time."""
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    ctx, desc, models, scene, pipe = _setup()
    _frames(pipe, scene, cam, sky, 1, 4)
    want = _state(pipe)
    ctx, desc, models, scene, pipe = _setup()
    t0 = np.array(desc.instances[2][1], np.float32)
    for f in range(1, 5):
        pipe.render(scene, cam, sky, GI, frame_index=f, rand=synth.frame_rand(11, f))
        moved = t0.copy()
        moved[3] += 8.0
        scene.set_transform(2, moved)   # there ...
        scene.commit()
        scene.set_transform(2, t0)      # ... and back again, behind the frame in flight
        scene.commit()
    for x, y in zip(want, _state(pipe)):
        assert np.array_equal(x, y)
    # host cost of a commit with one moved instance, frames in flight (tlas.rs:37-65 does this inside the frame)
    ctx.sync()
    n = 200
    t = time.perf_counter()
    for k in range(n):
        moved = t0.copy()
        moved[3] += float(k % 5)
        scene.set_transform(2, moved)
        scene.commit()
    per = (time.perf_counter() - t) / n * 1e6
    ctx.sync()
    print(f"set_transform + commit through ctypes: {per:.1f} us per call pair")
    assert per < 500.0   # (ctypes + numpy marshalling dominate here; tools/commit_timing measures the C call alone)


def test_host_a_ring_of_commits_ahead_of_the_gpu():
    """A frame loop that never waits: set_transform + commit + render, 64 times over at a frame size the GPU needs longer for than
    the host. The scene image lives in a ring of 16 copies (8 when this test was written: 64 frames go round either several times); when the ring comes round the commit waits until the frame after the
    slot's last reader has STARTED (the word its first launch writes, DustHipContext::started) -- not for the whole queue.
    Every frame goes into the running mean (PASS_ACCUMULATE) and moves the instance by its own amount: a frame that saw a
    recycled image too early, or too late, changes the mean. Equal, bit for bit, to the same loop with a wait after every frame;
    and to the loop in a pipeline with GI passes on the side stream in between (the commit then takes the careful way)."""
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_ACCUMULATE

    def loop(wait, gi_every=0):
        ctx, desc, models, scene, pipe = _setup(w=1920, h=1080)
        t0 = np.array(desc.instances[2][1], np.float32)
        for f in range(1, 65):
            moved = t0.copy()
            moved[3] += 0.37 * f
            moved[7] -= 0.11 * f
            scene.set_transform(2, moved)
            scene.commit()
            extra = (L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED) if gi_every and f % gi_every == 0 else 0
            pipe.render(scene, cam, sky, passes | extra, frame_index=f, rand=synth.frame_rand(11, f))
            if wait:
                ctx.sync()
        return [pipe.read_plane(pl) for pl in (L.PLANE_ACCUM, L.PLANE_DEPTH, L.PLANE_VOXEL_ID, L.PLANE_MOTION)]
    want = loop(True)
    got = loop(False)
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
    want = loop(True, gi_every=3)
    got = loop(False, gi_every=3)
    for x, y in zip(want[1:], got[1:]):   # (the GI passes feed the illuminance the mean is taken of; their own order is fixed by GI_ORDERED)
        assert np.array_equal(x, y)
    assert np.array_equal(want[0], got[0])


def test_gi_state_save_and_restore():
    """dust_hip_pipeline_read_gi / _write_gi: three frames, save, restore into a fresh pipeline, two more frames == five frames."""
    cam, sky = P.camera_for((90.0, 60.0, -80.0)), P.sky_state()
    ctx, desc, models, scene, pipe = _setup()
    _frames(pipe, scene, cam, sky, 1, 5)
    want = _state(pipe)
    ctx, desc, models, scene, pipe = _setup()
    _frames(pipe, scene, cam, sky, 1, 3)
    h, sp = pipe.read_gi()
    pipe2 = api.StandardPipeline(ctx, pipe.width, pipe.height)
    pipe2.set_noise(0, synth.stbn_scalar(layers=4))
    pipe2.set_noise(5, synth.stbn_unitvec3_cosine(layers=4))
    pipe2.configure_gi(4093, 777)
    pipe2.write_gi(h, sp)
    _frames(pipe2, scene, cam, sky, 4, 5)
    for x, y in zip(want, _state(pipe2)):
        assert np.array_equal(x, y)
    with pytest.raises(L.DustError):
        pipe2.write_gi(h[:-1], sp)   # a state of another capacity is refused
