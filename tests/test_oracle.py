"""The oracle checked against what CAN pin it without the reference running: closed-form format
conversions (numpy), the committed sky fixtures, and its own brute-force mode as the semantic ground
truth for its hierarchical mode."""
import ctypes as C
import json
import os

import numpy as np
import pytest

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L

l = O.lib()


def test_f16_conversion_matches_numpy():
    rng = np.random.default_rng(0)
    vals = np.concatenate([rng.standard_normal(20000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 20000),
                           np.array([0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e5, np.inf, -np.inf, 5.96e-8, 2.98e-8, 2.99e-8, 6.1e-5],
                                    np.float32)]).astype(np.float32)
    ours = np.array([l.orc_f32_to_f16(float(v)) for v in vals], np.uint16)
    with np.errstate(over="ignore"):
        ref = vals.astype(np.float16).view(np.uint16)
    assert np.array_equal(ours, ref)
    back = np.array([l.orc_f16_to_f32(int(h)) for h in ref], np.float32)
    assert np.array_equal(back.view(np.uint32), ref.view(np.float16).astype(np.float32).view(np.uint32))


def test_rgb10a2_roundtrip():
    rng = np.random.default_rng(1)
    for _ in range(2000):
        v = rng.random(4).astype(np.float32)
        p = l.orc_pack_rgb10a2((C.c_float * 4)(*v))
        out = (C.c_float * 4)()
        l.orc_unpack_rgb10a2(p, out)
        assert abs(out[0] - v[0]) <= 0.5 / 1023 + 1e-7 and abs(out[3] - v[3]) <= 0.5 / 3 + 1e-7
    assert l.orc_pack_rgb10a2((C.c_float * 4)(1, 1, 1, 1)) == 0xFFFFFFFF
    assert l.orc_pack_rgb10a2((C.c_float * 4)(-1, float("nan"), 2, 0.5)) == (1023 << 20) | (2 << 30)


def test_normal_codec_axis_normals():
    # axis normals come back ~1e-3 off-axis through RGB10A2 (0.5 is not representable)
    for n in ((1, 0, 0), (-1, 0, 0), (0, 1, 0), (0, -1, 0), (0, 0, 1), (0, 0, -1)):
        pk = (C.c_float * 4)()
        l.orc_nrd_pack_normal(O.f3(n), 1.0, 7.0, pk)
        q = (C.c_float * 4)()
        l.orc_unpack_rgb10a2(l.orc_pack_rgb10a2(pk), q)
        out = (C.c_float * 3)()
        l.orc_nrd_unpack_normal(q, out)
        assert np.allclose(list(out), n, atol=3e-3)
        assert abs(np.linalg.norm(list(out)) - 1) < 1e-6
        assert pk[3] == 1.0  # materialID / 3 clamped


def test_face_ids():
    # normal.glsl:9-18 (the code, not its comment): +x 1, -x 0, +y 3, -y 2, +z 5, -z 4
    exp = {(1, 0, 0): 1, (-1, 0, 0): 0, (0, 1, 0): 3, (0, -1, 0): 2, (0, 0, 1): 5, (0, 0, -1): 4}
    for n, f in exp.items():
        assert l.orc_normal2faceid(O.f3(n)) == f


def test_cubed_normalize_ties():
    out = (C.c_float * 3)()
    l.orc_cubed_normalize(O.f3((0.5, -0.5, 0.1)), out)
    assert list(out) == [1.0, -1.0, 0.0]   # ties give multi-axis normals (normal.glsl:39-43)
    l.orc_cubed_normalize(O.f3((0.2, -0.5, 0.1)), out)
    assert list(out) == [0.0, -1.0, 0.0]


def test_hashes_known_values():
    # pcg / xxhash32 restated from spatial_hash.glsl:105-126; cross-checked against a numpy restatement
    def pcg(v):
        state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
        word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
        return ((word >> 22) ^ word) & 0xFFFFFFFF

    def xx(p):
        h = (p + 374761393) & 0xFFFFFFFF
        h = (668265263 * (((h << 17) | (h >> 15)) & 0xFFFFFFFF)) & 0xFFFFFFFF
        h = (2246822519 * (h ^ (h >> 15))) & 0xFFFFFFFF
        h = (3266489917 * (h ^ (h >> 13))) & 0xFFFFFFFF
        return h ^ (h >> 16)
    for v in (0, 1, 2, 12345, 0xFFFFFFFF, 0x80000000):
        assert l.orc_pcg(v) == pcg(v)
        assert l.orc_xxhash32(v) == xx(v)


def test_logluv_roundtrip():
    rng = np.random.default_rng(3)
    for _ in range(500):
        c = (rng.random(3) * 10.0 ** rng.integers(-3, 3)).astype(np.float32)
        p = l.orc_logluv_encode(O.f3(c))
        out = (C.c_float * 3)()
        l.orc_logluv_decode(p, out)
        lum = lambda v: 0.2722288 * v[0] + 0.6740818 * v[1] + 0.05368953 * v[2]
        if p:
            assert abs(lum(list(out)) / lum(c) - 1) < 0.01   # 0.17 % log steps + chroma quantisation
    assert l.orc_logluv_encode(O.f3((0, 0, 0))) == 0


def test_sky_fixture_and_eval():
    sky = P.sky_state("default")
    # the values SURVEY 8(c) records for Sunlight::default()
    assert np.allclose([sky[9], sky[25], sky[41]], [0.0492231, 0.0523076, 0.0967448], rtol=1e-6)
    assert np.allclose(sky[52:55], [2.297167e6, 2.391441e6, 2.118000e6], rtol=1e-6)
    assert np.allclose(sky[0:3], [-1.0652946, -0.15876472, 1.5408292], rtol=1e-6)
    s = O.sky_from(sky)
    out = (C.c_float * 3)()
    l.orc_sky_radiance(C.byref(s), O.f3((0.0, 1.0, 0.0)), out)
    assert all(v > 0 for v in out)
    l.orc_sun_radiance(C.byref(s), O.f3(sky[48:51]), out)
    assert all(v > 1e5 for v in out)          # looking straight at the sun disc
    l.orc_sun_radiance(C.byref(s), O.f3((0.0, 1.0, 0.0)), out)
    assert list(out) == [0.0, 0.0, 0.0]       # outside the 0.255 degree disc
    # float64 restatement of sky.glsl:1-15 for one direction
    d = np.array([0.3, 0.5, -0.4]); d /= np.linalg.norm(d)
    l.orc_sky_radiance(C.byref(s), O.f3(d), out)
    sd = sky[48:51].astype(np.float64)
    cg = float(d @ sd); g = np.arccos(cg); ct = min(max(d[1], 0), 1)
    xyz = []
    for c in range(3):
        k = sky[c * 16: c * 16 + 9].astype(np.float64)
        v = (1 + k[0] * np.exp(k[1] / (ct + 0.01))) * (k[2] + k[3] * np.exp(k[4] * g) + k[5] * cg * cg +
                                                       k[6] * (1 + cg * cg) / (1 + k[8] ** 2 - 2 * k[8] * cg) ** 1.5 + k[7] * np.sqrt(ct))
        xyz.append(v * sky[c * 16 + 9] * 683.0)
    M = np.array([[1.6410228, -0.32480323, -0.23642465], [-0.66366285, 1.6153315, 0.016756356], [0.011721907, -0.0082844375, 0.9883947]])
    assert np.allclose(list(out), M @ np.array(xyz), rtol=2e-5)


def test_dda_basics():
    t, vox, kind = C.c_float(), C.c_uint32(), C.c_int()
    full = 0xFFFFFFFF
    # ray along +x through the brick centre row, full brick: hit at entry, voxel (0,1,2)
    assert l.orc_dda(0, O.f3((-2, 1.5, 2.5)), O.f3((1, 0, 0)), full, full, 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    assert t.value == 2.0 and vox.value == (0 << 4) | (1 << 2) | 2 and kind.value == 0
    # only voxel (3,1,2) set: walks three cells
    bit = (3 << 4) | (1 << 2) | 2
    assert l.orc_dda(0, O.f3((-2, 1.5, 2.5)), O.f3((1, 0, 0)), 0, 1 << (bit - 32), 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    assert t.value == 5.0 and vox.value == bit
    # empty brick: no report; ray missing the box: no report
    assert not l.orc_dda(0, O.f3((-2, 1.5, 2.5)), O.f3((1, 0, 0)), 0, 0, 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    assert not l.orc_dda(0, O.f3((-2, 5.5, 2.5)), O.f3((1, 0, 0)), full, full, 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    # brick behind the origin is rejected (t1 <= 0)
    assert not l.orc_dda(0, O.f3((6, 1.5, 2.5)), O.f3((1, 0, 0)), full, full, 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    # the reference's tmin quirk: a brick that ends before tmin still reports t = tmin if the clamped cell is set
    assert l.orc_dda(0, O.f3((3.95, 1.5, 2.5)), O.f3((1, 0, 0)), full, full, 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    assert abs(t.value - 0.1) < 1e-7
    # AO variant: brick straddling t = 8 reports its entry with kind 1, voxel 0xFF (ambient_occlusion.rint:62-73)
    assert l.orc_dda(1, O.f3((-6, 1.5, 2.5)), O.f3((1, 0, 0)), 1, 0, 0.1, C.byref(t), C.byref(vox), C.byref(kind))
    assert t.value == 6.0 and vox.value == 0xFF and kind.value == 1
    # rough variant reports the slab entry of any non-empty brick (rough.rint:42-59)
    assert l.orc_dda_rough(O.f3((-2, 1.5, 2.5)), O.f3((1, 0, 0)), 1, 0, C.byref(t)) and t.value == 2.0
    assert not l.orc_dda_rough(O.f3((-2, 1.5, 2.5)), O.f3((1, 0, 0)), 0, 0, C.byref(t))
    # direction with an exact zero component (the default sun has dir.x == 0): terminates and hits
    assert l.orc_dda(0, O.f3((1.5, -3, 0.5)), O.f3((0.0, 0.8, 0.6)), full, full, 0.1, C.byref(t), C.byref(vox), C.byref(kind))


@pytest.mark.parametrize("seed", [1, 2])
def test_hierarchical_equals_brute_force(seed):
    """The hierarchical walk must return exactly what 'closest over ALL bricks' returns (SURVEY section 7 hard part 1)."""
    desc = P.small_scene(seed=seed, n_models=2, n_instances=4, size=(28, 24, 30))
    s = P.oracle_scene(desc)
    rng = np.random.default_rng(seed)
    n_hit = 0
    for i in range(1500):
        o = rng.uniform(-90, 90, 3)
        tgt = rng.uniform(-40, 40, 3)
        d = tgt - o
        if i % 3 == 0:
            d /= np.linalg.norm(d)
        if i % 50 == 0:
            d[int(rng.integers(0, 3))] = 0.0           # axis-parallel components
        if i % 70 == 0:
            o = np.round(o)                              # lattice origins: rays along brick faces and edges
            d = np.round(d / np.abs(d).max() * 2) / 2
            if not d.any():
                d[0] = 1.0
        for rt, anyhit, tmin, tmax in ((0, 0, 0.1, 10000.0), (1, 0, 0.1, 8.0), (1, 1, 0.1, 10000.0), (2, 0, 8.0, 10000.0)):
            a = s.trace(O.ORC_MODE_BRUTE, rt, anyhit, o, d, tmin, tmax)
            b = s.trace(O.ORC_MODE_HIER, rt, anyhit, o, d, tmin, tmax)
            if anyhit:
                assert (a is None) == (b is None), (i, rt, o, d, a, b)
            else:
                assert a == b, (i, rt, o, d, a, b)
            n_hit += a is not None
    assert n_hit > 200


def test_hierarchical_equals_brute_force_frame():
    desc = P.small_scene(seed=7, n_models=2, n_instances=3, size=(24, 24, 24))
    s = P.oracle_scene(desc)
    sky = P.sky_state()
    w, h = 40, 28
    noise5 = __import__("dust_amd.synth", fromlist=["x"]).stbn_unitvec3_cosine(layers=1)[0]
    for eye in ((70.0, 50.0, 80.0), (64.0, 32.0, 0.0), (0.5, 120.0, 0.5)):
        cam = P.camera_for(eye) if eye[0] != 0.5 else P.api.make_camera(eye, np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]], np.float32).T, P.api.PinholeProjection())
        passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
        a = P.render_oracle(s, cam, sky, w, h, passes, noise5, 99, mode=O.ORC_MODE_BRUTE)
        b = P.render_oracle(s, cam, sky, w, h, passes, noise5, 99, mode=O.ORC_MODE_HIER)
        for name in ("illuminance", "denoised", "albedo", "normal", "depth", "motion", "voxel_id"):
            assert getattr(a, name).tobytes() == getattr(b, name).tobytes(), (eye, name)
        assert np.isfinite(a.depth).sum() > 50


@pytest.mark.parametrize("seed", [500003, 500010, 500016])
def test_hierarchical_equals_brute_force_on_deep_trees(seed):
    """The same on 4096^3 models (three-level hierarchy: what the DEEP kernel variants are compared with): clusters of 16-cells holding
    1 to 40 bricks, all five passes for two frames, hash and pool included. A slice of tools/oracle_deep_sweep.py (400 scenes clean)."""
    from dust_amd import synth
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    rng = np.random.default_rng(seed)
    half = int(rng.integers(2, 6))
    c0 = int(rng.integers(half, 256 - half))
    blocks, mats, pal = P.clustered_deep_model(seed=seed, n_cells=int(rng.integers(20, 300)), cell_lo=c0 - half, cell_hi=c0 + half,
                                               max_bricks=int(rng.choice([2, 12, 40])))
    xf = np.eye(3, 4, dtype=np.float32)
    xf[:, 3] = -16.0 * c0
    oscene = O.Scene()
    oscene.add_model(blocks, mats, pal, extent=4096)
    oscene.add_instance(0, xf.reshape(12))
    oscene.commit()
    reach = 16.0 * half
    eye = np.round(rng.uniform(-1.5 * reach, 1.5 * reach, 3) / 16.0) * 16.0 if seed % 2 == 0 else rng.uniform(-1.5 * reach, 1.5 * reach, 3)
    if abs(eye[0]) + abs(eye[2]) < 1e-3:
        eye[0] = 3.0
    cam = P.camera_for(tuple(float(v) for v in eye))
    w, h = 40, 28
    states = []
    for mode in (O.ORC_MODE_HIER, O.ORC_MODE_BRUTE):
        gi = O.GI(4093, 777)
        planes = []
        for f in (1, 2):
            rnd = synth.frame_rand(seed, f)
            g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f, mode=mode)
            planes.append((g.depth.copy(), g.voxel_id.copy(), g.illuminance.copy()))
        states.append((planes, gi.hash().copy(), gi.pool().copy()))
    a, b = states
    assert np.isfinite(a[0][0][0]).mean() > 0.02
    for x, y in zip(a[0], b[0]):
        assert np.array_equal(x[0].view(np.uint32), y[0].view(np.uint32)) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2])
    assert np.array_equal(a[1]["fingerprint"], b[1]["fingerprint"]) and np.array_equal(a[2]["direction"], b[2]["direction"])
