"""Multi-GPU GI protocol (include/dust_hip.h, dust_hip_pipeline_gi_exchange) exercised on ONE GPU: R pipelines play R
ranks on row bands, the collectives are done by hand on the host, and every rank's spatial hash, surfel pool and band
of the radiance plane must equal the single-pipeline run bit for bit (ordered apply)."""
import ctypes

import numpy as np
import pytest

import parity_util as P
from dust_amd import _lib as L, api, sharding, synth

pytestmark = pytest.mark.gpu


def _hip():
    lib = ctypes.CDLL("libamdhip64.so")
    lib.hipMemcpy.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
    lib.hipMemcpy.restype = ctypes.c_int
    return lib


def _d2h(hip, ptr, n_items):
    out = np.empty(n_items, np.int32)
    assert hip.hipMemcpy(out.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(ptr), out.nbytes, 2) == 0
    return out


def _h2d(hip, ptr, arr):
    arr = np.ascontiguousarray(arr, np.int32)
    assert hip.hipMemcpy(ctypes.c_void_p(ptr), arr.ctypes.data_as(ctypes.c_void_p), arr.nbytes, 1) == 0


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gi_equals_single_gpu(world):
    hip = _hip()
    W, H = 192, 104
    cap, pool = 16384, 97 * 8
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    s = 0.15
    eye = (122.0 * s, 300.61 * s, 54.45 * s)
    cam, sky = P.camera_for(eye), P.sky_state()

    def make():
        p = api.StandardPipeline(ctx, W, H)
        p.set_noise(0, n0)
        p.set_noise(5, n5)
        p.configure_gi(cap, pool)
        return p

    ref = make()
    ranks = [make() for _ in range(world)]
    per = sharding.gi_band_rows(world, H)
    bands = [(min(H, r * per), min(H, (r + 1) * per)) for r in range(world)]
    exs = [p.gi_exchange(world * per) for p in ranks]
    pix = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER
    for frame in range(1, 5):
        rnd = synth.frame_rand(7, frame)
        ref.render(scene, cam, sky, pix | L.PASS_SURFEL | L.PASS_GI_ORDERED, frame, rnd)
        for r, p in enumerate(ranks):
            p.render(scene, cam, sky, pix | L.PASS_GI_SHARDED, frame, rnd, rows=bands[r])
        ctx.sync()
        # all-reduce MAX of slot_owner, all-gather of the touched bands
        owner = np.max([_d2h(hip, e.slot_owner, e.pool_size) for e in exs], axis=0)
        touched = np.zeros(world * per * W, np.int32)
        for r, e in enumerate(exs):
            t = _d2h(hip, e.touched, world * per * W)
            touched[r * per * W:(r + 1) * per * W] = t[r * per * W:(r + 1) * per * W]
        for e in exs:
            _h2d(hip, e.slot_owner, owner)
            _h2d(hip, e.touched, touched)
        for r, p in enumerate(ranks):
            p.gi_export(*bands[r])
        ctx.sync()
        parts = [_d2h(hip, e.merged, e.pool_size * 4) for e in exs]
        contributors = np.stack([q.reshape(-1, 4).any(axis=1) for q in parts]).sum(axis=0)
        assert contributors.max() <= 1  # exactly one rank holds the winning pixel of a slot
        merged = np.sum(parts, axis=0, dtype=np.int64).astype(np.int32)  # all-reduce SUM: one contributor per slot
        for e in exs:
            _h2d(hip, e.merged, merged)
        for r, p in enumerate(ranks):
            p.gi_import(bands[r][0], bands[r][1], frame)
            p.render(scene, cam, sky, L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_GI_SHARDED, frame, rnd)
        ctx.sync()
        h_ref, s_ref = ref.read_gi()
        ill_ref = ref.read_plane(L.PLANE_ILLUMINANCE)
        for r, p in enumerate(ranks):
            h, sp = p.read_gi()
            assert np.array_equal(h, h_ref), f"frame {frame} rank {r}: hash differs in {(h != h_ref).any(axis=1).sum()} entries"
            same = sp.view(np.uint32) == s_ref.view(np.uint32)  # bytes: free slots hold 0xFFFFFFFF (a NaN as float)
            assert same.all(), f"frame {frame} rank {r}: surfel pool differs in {(~same).reshape(-1, 4).any(axis=1).sum()} slots"
            ill = p.read_plane(L.PLANE_ILLUMINANCE)
            bad = (ill[bands[r][0]:bands[r][1]] != ill_ref[bands[r][0]:bands[r][1]]).any(axis=-1)
            assert not bad.any(), f"frame {frame} rank {r}: {bad.sum()} pixels of the band differ (rows {np.unique(np.nonzero(bad)[0])[:8]})"
    assert int((h_ref[:, 0] != 0).sum()) > 50 and int((s_ref["direction"] < 6).sum()) > 50


def test_gi_band_without_the_flag_is_refused():
    ctx = api.Context(device=0)
    data, _ = synth.castle_scene(scale=0.15)
    scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
    p = api.StandardPipeline(ctx, 64, 64)
    p.set_noise(0, synth.stbn_scalar(layers=1))
    p.set_noise(5, synth.stbn_unitvec3_cosine(layers=1))
    p.configure_gi(4096, 64)
    with pytest.raises(L.DustError) as e:
        p.render(scene, P.camera_for((18.0, 45.0, 8.0)), P.sky_state(), L.PASS_FINAL_GATHER, 1, 1, rows=(0, 32))
    assert e.value.status == L.ERR_UNSUPPORTED
    with pytest.raises(L.DustError) as e:  # sharded final gather before the exchange buffers exist
        p.render(scene, P.camera_for((18.0, 45.0, 8.0)), P.sky_state(), L.PASS_FINAL_GATHER | L.PASS_GI_SHARDED, 1, 1, rows=(0, 32))
    assert e.value.status == L.ERR_NOT_READY


def test_torch_aliases_the_exchange_buffers():
    torch = pytest.importorskip("torch")
    hip = _hip()
    ctx = api.Context(device=0)
    p = api.StandardPipeline(ctx, 64, 64)
    p.configure_gi(4096, 64)
    ex = p.gi_exchange(64)
    owner, touched, merged = sharding.alias_exchange_buffers(ex)
    assert owner.numel() == 64 and touched.numel() == 64 * 64 and merged.numel() == 256
    owner.copy_(torch.arange(64, dtype=torch.int32, device="cuda"))
    torch.cuda.synchronize()
    assert np.array_equal(_d2h(hip, ex.slot_owner, 64), np.arange(64, dtype=np.int32))


@pytest.mark.parametrize("w,h,world", [(88, 40, 4), (113, 80, 9), (136, 18, 5)])
def test_ranks_without_rows_take_part_in_the_exchange(w, h, world):
    """Small frames on many GPUs: the 8-row-aligned bands run out before the ranks do. A rank without rows renders no pixels but is in
    every collective -- it must export zeroes (its `merged` still holds last frame's all-reduced sum otherwise, which is what
    tools/stress_sharded.py found) and import the others' stamps. Every rank's hash and pool equal the single-pipeline run."""
    from dust_amd import sharding
    per = sharding.gi_band_rows(world, h)
    assert any(min(h, r * per) >= min(h, (r + 1) * per) for r in range(world))
    desc = P.small_scene(seed=31, n_models=2, n_instances=6)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    hsh = P.sharded_gi_vs_single_device(ctx, scene, P.camera_for((90.0, 60.0, -80.0)), P.sky_state(), w, h, world, 4, n0, n5, seed=9, gi_sizes=(4093, 777))
    assert int((hsh[:, 0] != 0).sum()) >= 5   # (the hash really was fed)
