"""Auto exposure + tone map: oracle sanity on CPU, HIP vs oracle on the GPU."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import api, synth


def oracle_tone_map(denoised, albedo, avg0, conv, tf, min_log=-6.0, max_log=8.5, tc=0.2):
    l = O.lib()
    h, w = albedo.shape
    hist = np.zeros(256, np.uint32)
    den = np.ascontiguousarray(denoised, np.uint16)
    alb = np.ascontiguousarray(albedo, np.uint32)
    l.orc_exposure_histogram(den.ctypes.data_as(C.c_void_p), w, h, min_log, max_log - min_log, hist.ctypes.data_as(C.c_void_p))
    counts = hist.copy()
    avg = l.orc_exposure_average(hist.ctypes.data_as(C.c_void_p), w, h, min_log, max_log - min_log, tc, avg0)
    out = np.zeros((h, w, 4), np.uint16)
    l.orc_tone_map(den.ctypes.data_as(C.c_void_p), alb.ctypes.data_as(C.c_void_p), w, h, avg,
                   (C.c_float * 9)(*[float(v) for v in conv]), tf, out.ctypes.data_as(C.c_void_p))
    return counts, avg, out


def test_color_space_conversion_matrix():
    m = api.color_space_conversion(api.ACES_AP1, api.BT709).reshape(3, 3).T   # back to row-major
    # AP1 (D60 white) -> BT.709 (D65 white) without chromatic adaptation: AP1 white (1,1,1) lands on D60 in 709
    # coordinates, and the matrix undoes the XYZ round trip
    xyz_ap1 = api.primaries_to_xyz(api.ACES_AP1)
    xyz_709 = api.primaries_to_xyz(api.BT709)
    assert np.allclose(xyz_709 @ m, xyz_ap1, atol=1e-5)
    assert np.allclose(xyz_709 @ np.ones(3), [0.9505, 1.0, 1.0891], atol=2e-3)    # D65 white point of BT.709
    assert np.allclose(api.color_space_conversion(api.BT709, api.BT709).reshape(3, 3), np.eye(3), atol=1e-6)


def test_oracle_exposure_and_tone_map_basics():
    h, w = 16, 32
    den = np.zeros((h, w, 4), np.float16)
    den[..., 0] = 0.5       # Y; Co = Cg = 0 -> grey 0.5
    den[:4, :, 0] = 0.001   # below the 0.005 luminance floor -> bin 0
    alb = np.full((h, w), 0xFFFFFFFF, np.uint32)
    conv = np.eye(3, dtype=np.float32).reshape(9)
    counts, avg, out = oracle_tone_map(den.view(np.uint16), alb, 0.0, conv, 1)
    assert counts[0] == 4 * w and counts.sum() == h * w
    b = int((np.log2(0.5) + 6.0) / 14.5 * 254.0 + 1.0)
    assert counts[b] == (h - 4) * w
    # first frame: avg = 0 + (lum - 0) * 0.2, with lum rebuilt from the weighted mean bin (auto_exposure_avg.comp:41-52)
    mean_bin = b * (h - 4) / h - 1.0
    assert np.isclose(avg, 0.2 * 2.0 ** (mean_bin / 254.0 * 14.5 - 6.0), rtol=1e-5)
    o = out.view(np.float16).astype(np.float32)
    assert np.all(o[..., 3] == 1.0) and np.all(o[..., :3] >= 0) and np.all(o[..., :3] <= 1.001)
    assert o[8, 0, 0] > o[0, 0, 0]           # brighter in -> brighter out
    # transfer function 0 (linear) is darker than sRGB for mid tones
    _, _, lin = oracle_tone_map(den.view(np.uint16), alb, 0.0, conv, 0)
    assert lin.view(np.float16)[8, 0, 0] < out.view(np.float16)[8, 0, 0]


@pytest.mark.gpu
@pytest.mark.parametrize("tf", [0, 1, 4, 7])
def test_tone_map_gpu_matches_oracle(tf):
    desc = P.small_scene(seed=3)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    w, h = 160, 96
    pipe = api.StandardPipeline(ctx, w, h)
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    pipe.set_noise(5, n5)
    sky = P.sky_state()
    cam = P.camera_for((90.0, 70.0, 110.0))
    conv = api.color_space_conversion()
    avg = 0.0
    for f in range(1, 4):   # three frames: mean accumulation + exposure adaptation
        pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_ACCUMULATE, frame_index=f,
                    rand=synth.frame_rand(1, f))
        pipe.tone_map(transfer_function=tf, conversion=conv)
        den, alb = pipe.read_plane(L.PLANE_DENOISED), pipe.read_plane(L.PLANE_ALBEDO)
        counts, avg, out = oracle_tone_map(den, alb, avg, conv, tf)
        got_avg = pipe.exposure()
        assert np.isclose(got_avg, avg, rtol=2e-3), (f, got_avg, avg)    # a pixel on a bin edge may move one bin
        a = out.view(np.float16).astype(np.float32)
        b = pipe.read_plane(L.PLANE_OUTPUT).view(np.float16).astype(np.float32)
        fin = np.isfinite(a) & np.isfinite(b)   # the ACES fit dips below 0 for dark pixels: sqrt/log OETFs give NaN there
        assert (np.isfinite(a) != np.isfinite(b)).mean() < 1e-3
        rel = np.sqrt(((a[fin] - b[fin]) ** 2).sum()) / np.sqrt((a[fin] ** 2).sum())
        print('tone map rel L2', tf, f, rel)
        assert rel <= 1e-3, (f, rel)   # north_star's tolerance
        avg = got_avg   # keep both sides on the same adaptation state
    # the accumulated mean really is the mean of the per-frame illuminance
    acc = pipe.read_plane(L.PLANE_ACCUM)
    assert np.all(acc[..., 3][np.isfinite(pipe.read_plane(L.PLANE_DEPTH))] == 3.0)
