"""DUST_PASS_DENOISE (dust_amd/csrc/denoise.hip): the native spatiotemporal filter that stands where the reference calls NVIDIA
NRD (nrd.rs:272-617; closed SDK: PARITY UNPINNED against it). Two kinds of evidence: the kernels against the oracle's scalar
restatement of the same filter over frame sequences with a moving camera and a moving instance, and the properties any such
filter must have -- a static view accumulates exactly the running mean, history follows moving geometry, nothing ghosts
where geometry was uncovered."""
import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L, api, synth

pytestmark = pytest.mark.gpu
W, H = 256, 160


def unpack(ill):  # REBLUR_BackEnd_UnpackRadianceAndNormHitDist (nrd.glsl:107-125)
    f = ill.view(np.float16).astype(np.float32)
    t = f[..., 0] - f[..., 2]
    return np.stack([np.maximum(t + f[..., 1], 0), np.maximum(f[..., 0] + f[..., 2], 0), np.maximum(t - f[..., 1], 0)], axis=-1)


def build(ctx, seed=5):
    rng = np.random.default_rng(seed)
    pal = synth.make_palette(seed)
    floor = np.array([[x, 0, z, 1 + (x // 8 + z // 8) % 5] for x in range(120) for z in range(120)], np.uint8)  # file axes: y is depth
    floor = np.array([[x, y, 0, 3 + (x // 8 + y // 8) % 5] for x in range(120) for y in range(120)], np.uint8)
    mover = P.random_model(rng, (20, 20, 20), fill=0.5, blobs=2)
    models = [api.Model(ctx, *api.flatten_model(floor, (120, 120, 1), pal), pal), api.Model(ctx, *api.flatten_model(mover, (20, 20, 20), pal), pal)]
    scene = api.Scene(ctx)
    a = np.eye(3, 4, dtype=np.float32); a[:, 3] = (-60, 0, -60)
    scene.add_instance(models[0], a.reshape(12))
    b = np.eye(3, 4, dtype=np.float32); b[:, 3] = (-10, 6, -10)
    mid = scene.add_instance(models[1], b.reshape(12))
    scene.commit()
    return scene, models, mid, b


def mat4(o2w):
    m = np.eye(4, dtype=np.float32)
    m[:3, :] = np.asarray(o2w, np.float32).reshape(3, 4)
    return m.T.reshape(16)  # column-major


def pipeline(ctx, **kw):
    pipe = api.StandardPipeline(ctx, W, H)
    pipe.set_noise(0, synth.stbn_scalar(layers=8))
    pipe.set_noise(5, synth.stbn_unitvec3_cosine(layers=8))
    pipe.configure_gi(1 << 16, 8192)
    pipe.set_denoiser(**kw)
    return pipe


PASSES = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_DENOISE


def test_denoiser_matches_oracle_over_a_moving_sequence():
    ctx = api.Context(device=0)
    scene, models, mid, xf = build(ctx)
    sky = P.sky_state()
    pipe = pipeline(ctx)
    orc = O.Denoiser(W, H)
    prev = xf.copy()
    worst = 0.0
    for f in range(1, 7):
        eye = (70.0 + 1.5 * f, 60.0 - 0.7 * f, 80.0 + 0.9 * f)   # the camera drifts
        cam = P.camera_for(eye, target=(0.0, 5.0, 0.0))
        cur = xf.copy(); cur[0, 3] += 2.0 * f; cur[2, 3] -= 1.25 * f   # the mover slides over the floor
        scene.set_transform(mid, cur.reshape(12), mat4(prev))
        scene.commit()
        prev = cur
        pipe.render(scene, cam, sky, PASSES, frame_index=f, rand=synth.frame_rand(4, f))
        g = P.read_hip_gbuffer(pipe)
        acc = pipe.read_plane(L.PLANE_ACCUM)
        # the oracle filter runs on the SAME noisy planes (what it checks is the filter, not the passes before it); its
        # `denoised` input is the plane as the primary pass left it for miss pixels, which the filter does not touch
        want_den, want_acc = orc.frame(g, cam, f)
        hit = np.isfinite(g["depth"])
        assert hit.mean() > 0.3
        a, b = unpack(g["denoised"])[hit].astype(np.float64), unpack(want_den)[hit].astype(np.float64)
        rel = float(np.sqrt(((a - b) ** 2).sum() / max(1e-30, (b ** 2).sum())))
        rel_acc = float(np.sqrt(((acc[hit][:, :3].astype(np.float64) - want_acc[hit][:, :3]) ** 2).sum() / max(1e-30, (want_acc[hit][:, :3].astype(np.float64) ** 2).sum())))
        assert rel <= 1e-3 and rel_acc <= 1e-3, (f, rel, rel_acc)
        assert np.abs(acc[hit][:, 3] - want_acc[hit][:, 3]).max() <= 1e-3  # accumulated frame counts
        assert np.array_equal(g["denoised"][~hit], want_den[~hit])         # the sky stays as miss.rmiss wrote it
        worst = max(worst, rel, rel_acc)
        if f >= 3:  # the history really is reused while everything moves
            assert np.median(acc[hit][:, 3]) >= 2.0
    print("denoiser vs oracle, worst rel L2:", worst)


def test_static_view_accumulates_the_running_mean():
    ctx = api.Context(device=0)
    scene, _, _, _ = build(ctx)
    sky, cam = P.sky_state(), P.camera_for((70.0, 60.0, 80.0), target=(0.0, 5.0, 0.0))
    pipe = pipeline(ctx, max_accumulated_frames=6, antilag_power=0.0, max_blur_radius=0.0)
    samples = []
    for f in range(1, 10):
        pipe.render(scene, cam, sky, PASSES, frame_index=f, rand=synth.frame_rand(6, f))
        g = P.read_hip_gbuffer(pipe)
        hit = np.isfinite(g["depth"])
        samples.append(unpack(g["illuminance"]).astype(np.float64))
        acc = pipe.read_plane(L.PLANE_ACCUM)
        assert np.all(acc[hit][:, 3] == min(f, 6))
        if f <= 6:   # plain mean of the f samples
            want = np.mean(samples, axis=0)
        else:        # capped: exponential with weight 1/6
            want = want * (5.0 / 6.0) + samples[-1] / 6.0
        err = np.abs(acc[hit][:, :3] - want[hit]).max() / max(1e-9, np.abs(want[hit]).max())
        assert err < 1e-5, (f, err)
        # with the blur off the denoised plane is the accumulation, packed
        assert np.abs(unpack(g["denoised"])[hit] - acc[hit][:, :3]).max() <= 2e-3 * max(1e-9, np.abs(acc[hit][:, :3]).max())
    # DenoiserEvent::Restart
    pipe.restart_denoiser()
    pipe.render(scene, cam, sky, PASSES, frame_index=10, rand=synth.frame_rand(6, 10))
    assert np.all(pipe.read_plane(L.PLANE_ACCUM)[hit][:, 3] == 1.0)


def test_spatial_pass_lowers_noise_without_crossing_edges():
    ctx = api.Context(device=0)
    scene, _, _, _ = build(ctx)
    sky, cam = P.sky_state(), P.camera_for((70.0, 60.0, 80.0), target=(0.0, 5.0, 0.0))
    out = {}
    for radius in (0.0, 15.0):
        pipe = pipeline(ctx, max_blur_radius=radius)
        for f in range(1, 4):
            pipe.render(scene, cam, sky, PASSES, frame_index=f, rand=synth.frame_rand(8, f))
        g = P.read_hip_gbuffer(pipe)
        out[radius] = (unpack(g["denoised"]), g)
    g = out[0.0][1]
    floor = np.isfinite(g["depth"]) & ((g["voxel_id"] & 0xFFFF) == 0)
    lum = lambda x: x @ np.array([0.25, 0.5, 0.25])
    interior = floor.copy()
    interior[:2] = interior[-2:] = False
    interior[:, :2] = interior[:, -2:] = False

    def roughness(img):  # mean absolute Laplacian of the luminance over floor pixels whose 4 neighbours are floor too
        y = lum(img)
        ok = interior & np.roll(floor, 1, 0) & np.roll(floor, -1, 0) & np.roll(floor, 1, 1) & np.roll(floor, -1, 1)
        lap = 4 * y - np.roll(y, 1, 0) - np.roll(y, -1, 0) - np.roll(y, 1, 1) - np.roll(y, -1, 1)
        return float(np.abs(lap[ok]).mean())

    assert roughness(out[15.0][0]) < 0.6 * roughness(out[0.0][0])
    # energy is kept: the blur is a normalised average of pixels of the same surface
    a, b = lum(out[0.0][0])[floor].mean(), lum(out[15.0][0])[floor].mean()
    assert abs(a - b) < 0.05 * a


def test_history_follows_a_moving_instance_and_does_not_ghost():
    ctx = api.Context(device=0)
    scene, models, mid, xf = build(ctx)
    sky, cam = P.sky_state(), P.camera_for((70.0, 60.0, 80.0), target=(0.0, 5.0, 0.0))
    pipe = pipeline(ctx)
    for f in range(1, 6):   # everything at rest: histories build up
        pipe.render(scene, cam, sky, PASSES, frame_index=f, rand=synth.frame_rand(9, f))
    before = P.read_hip_gbuffer(pipe)
    was_mover = np.isfinite(before["depth"]) & ((before["voxel_id"] & 0xFFFF) == mid)
    assert was_mover.sum() > 500
    cur = xf.copy(); cur[0, 3] += 30.0   # a jump of many pixels
    scene.set_transform(mid, cur.reshape(12), mat4(xf))
    scene.commit()
    pipe.render(scene, cam, sky, PASSES, frame_index=6, rand=synth.frame_rand(9, 6))
    g = P.read_hip_gbuffer(pipe)
    acc = pipe.read_plane(L.PLANE_ACCUM)
    hit = np.isfinite(g["depth"])
    is_mover = hit & ((g["voxel_id"] & 0xFFFF) == mid)
    uncovered = was_mover & hit & ~is_mover          # floor that the mover hid a frame ago
    assert uncovered.sum() > 200 and is_mover.sum() > 500
    # uncovered floor starts over from this frame's sample: no trace of the mover's radiance
    assert np.all(acc[uncovered][:, 3] == 1.0)
    cur_sample = unpack(g["illuminance"])
    assert np.abs(acc[uncovered][:, :3] - cur_sample[uncovered]).max() <= 1e-6 * max(1.0, np.abs(cur_sample[uncovered]).max())
    # the mover took its history along (motion vectors point back to where it was)
    assert np.median(acc[is_mover][:, 3]) >= 4.0
    # floor that was visible all along keeps accumulating
    steady = hit & ~is_mover & ~was_mover
    assert np.median(acc[steady][:, 3]) >= 5.0


def test_denoise_argument_checks():
    ctx = api.Context(device=0)
    scene, _, _, _ = build(ctx)
    pipe = pipeline(ctx)
    with pytest.raises(L.DustError):
        pipe.set_denoiser(max_accumulated_frames=0)
    with pytest.raises(L.DustError) as e:   # the filter reaches across rows: whole frames only
        pipe.render(scene, P.camera_for((70.0, 60.0, 80.0)), P.sky_state(), L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_DENOISE, rows=(0, 80))
    assert e.value.status == L.ERR_UNSUPPORTED
