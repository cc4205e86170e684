"""BASELINE.json configs[1] at its full size (castle stand-in, scale 1.0, 1920 x 1080): direct parity against the oracle
(hierarchical mode over the host's cores: seconds on a GPU box) and the size-independent properties the domain offers --
band union == frame, fused == separate launches, sharded GI == single-device GI."""
import ctypes
import os
import threading

import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L, api, sharding, synth

pytestmark = pytest.mark.gpu
W, H = 1920, 1080
EYE = (122.0, 300.61, 54.45)  # examples/castle.rs:126


@pytest.fixture(scope="module")
def castle():
    data, info = synth.castle_scene()
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    return ctx, desc, P.hip_scene(ctx, desc), info


def _oracle_frame(desc, cam, sky, noise5, rand, passes):
    oscene = P.oracle_scene(desc)
    g = O.GBuffer(W, H)
    n = max(1, min(os.cpu_count() or 1, H // 4))
    cuts = [H * i // n for i in range(n + 1)]
    th = [threading.Thread(target=P.render_oracle, args=(oscene, cam, sky, W, H, passes, noise5, rand),
                           kwargs={"rows": (cuts[i], cuts[i + 1]), "g": g}) for i in range(n)]
    [t.start() for t in th]
    [t.join() for t in th]
    return g


def test_full_size_primary_ao_matches_oracle(castle):
    ctx, desc, scene, info = castle
    assert info["n_voxels"] > 10_000_000 and info["n_instances"] > 100
    noise5 = synth.stbn_unitvec3_cosine()
    cam, sky = P.camera_for(EYE), P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    pipe = api.StandardPipeline(ctx, W, H)
    pipe.set_noise(5, noise5)
    rnd = synth.frame_rand(1, 1)
    pipe.render(scene, cam, sky, passes, frame_index=1, rand=rnd)
    hip = P.read_hip_gbuffer(pipe)
    g = _oracle_frame(desc, cam, sky, noise5[1 % len(noise5)], rnd, passes)
    res = P.compare_gbuffers(g, hip)
    P.assert_parity(res)
    assert res["illuminance_rel_l2"] <= 1e-3 and res.get("denoised_rel_l2", 0.0) <= 1e-3, res
    assert np.isfinite(g.depth).mean() > 0.5  # the reference camera looks down on the castle: (almost) every pixel hits


def test_full_size_bands_and_fusion(castle, monkeypatch):
    ctx, desc, scene, _ = castle
    noise5 = synth.stbn_unitvec3_cosine()
    cam, sky = P.camera_for(EYE), P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    full = api.StandardPipeline(ctx, W, H)
    full.set_noise(5, noise5)
    full.render(scene, cam, sky, passes, frame_index=2, rand=77)
    ref = P.read_hip_gbuffer(full)
    banded = api.StandardPipeline(ctx, W, H)
    banded.set_noise(5, noise5)
    for r in range(8):  # the 8-GPU partition
        banded.render(scene, cam, sky, passes, frame_index=2, rand=77, rows=sharding.band_rows(r, 8, H))
    got = P.read_hip_gbuffer(banded)
    for k in ref:
        assert ref[k].tobytes() == got[k].tobytes(), f"bands: {k}"
    monkeypatch.setenv("DUST_HIP_NO_FUSE", "1")
    sep = api.StandardPipeline(ctx, W, H)
    sep.set_noise(5, noise5)
    sep.render(scene, cam, sky, passes, frame_index=2, rand=77)
    got = P.read_hip_gbuffer(sep)
    monkeypatch.delenv("DUST_HIP_NO_FUSE")
    for k in ref:
        assert ref[k].tobytes() == got[k].tobytes(), f"separate launches: {k}"


def test_full_size_sharded_gi_equals_single_device(castle):
    """Default hash capacity (32 Mi entries) and surfel pool (345 600): two ranks on row bands, collectives by hand."""
    ctx, desc, scene, _ = castle
    h_ref = P.sharded_gi_vs_single_device(ctx, scene, P.camera_for(EYE), P.sky_state(), W, H, world=2, frames=2,
                                          n0=synth.stbn_scalar(), n5=synth.stbn_unitvec3_cosine())
    assert int((h_ref[:, 0] != 0).sum()) > 10_000


def test_full_size_gi_matches_oracle(castle):
    """BASELINE configs[2]'s passes at the reference's sizes -- 1920 x 1080, the 32 Mi-entry spatial hash (spatial_hash.glsl:1) and the
    345 600-slot surfel pool (surfel.glsl:2; standard.rs:330-358) -- against the oracle, three frames with the deterministic apply:
    every integer of the hash (fingerprints, counts, LRU stamps) and the pool bit for bit, LogLuv radiance within two quantisation
    steps, the G-buffer as in the small-frame tests, illuminance <= 1e-3 relative L2. The oracle's pixel passes are threaded over
    rows, its GI passes over bands / surfel ranges (orc_pass_*_mt: the serial passes' result, tests/test_gi_oracle.py)."""
    from test_gpu_gi import compare_gi
    ctx, desc, scene, _ = castle
    n0, n5 = synth.stbn_scalar(), synth.stbn_unitvec3_cosine()
    cam, sky = P.camera_for(EYE), P.sky_state()
    pipe = api.StandardPipeline(ctx, W, H)
    pipe.set_noise(0, n0)
    pipe.set_noise(5, n5)
    pipe.configure_gi(32 * 1024 * 1024, 720 * 480)
    gi = O.GI(32 * 1024 * 1024, 720 * 480)
    oscene = P.oracle_scene(desc)
    g = O.GBuffer(W, H)
    threads = max(1, min(os.cpu_count() or 1, H // 4))
    cuts = [H * i // threads for i in range(threads + 1)]
    pix = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    for f in range(1, 4):
        rnd = synth.frame_rand(1, f)
        pipe.render(scene, cam, sky, pix | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
        th = [threading.Thread(target=P.render_oracle, args=(oscene, cam, sky, W, H, pix, n5[f % len(n5)], rnd),
                               kwargs={"rows": (cuts[i], cuts[i + 1]), "g": g}) for i in range(threads)]
        [t.start() for t in th]
        [t.join() for t in th]
        P.render_oracle(oscene, cam, sky, W, H, L.PASS_FINAL_GATHER | L.PASS_SURFEL, n5[f % len(n5)], rnd, noise0=n0[f % len(n0)], gi=gi,
                        frame_index=f, g=g, gi_threads=threads)
        hip = P.read_hip_gbuffer(pipe)
        res = P.compare_gbuffers(g, hip)
        P.assert_parity(res)
        assert res["illuminance_rel_l2"] <= 1e-3, (f, res)
        used, valid = compare_gi(gi, pipe)
    assert used > 20_000 and valid > 300_000, (used, valid)   # (three frames: the pool is nearly full, the hash holds the bricks the surfels see)
