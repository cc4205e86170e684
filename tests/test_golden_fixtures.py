"""The C oracle (and the product's host-side flatten) against the INDEPENDENT numpy witness in tests/golden/
(make_shader_fixtures.py: SURVEY 8c ii-iv). The fixtures were produced by code that shares nothing with oracle/*.c;
agreement here means two separate readings of hit.rint / ambient_occlusion.rint / rough.rint, the codecs and
VoxGeometry::from_tree arrive at the same bits. tests/test_gpu_golden.py checks the HIP device functions against the same files."""
import ctypes as C
import os

import numpy as np
import pytest

import oracle_lib as O
from dust_amd import api

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load(name):
    return np.load(os.path.join(GOLD, name))


def test_fixture_is_large_and_covers_every_ray_class():
    fx = load("dda_pairs.npz")
    n = len(fx["o"])
    assert 3 * n >= 10_000 and int(fx["dropped"][0]) < n // 100
    assert set(np.unique(fx["category"]).tolist()) == {0, 1, 2, 3, 4, 5}
    for k in (0, 1, 2):  # both outcomes well represented for every shader
        assert 0.1 * n < int(fx[f"reported{k}"].sum()) < 0.9 * n
    assert int((fx["hitkind1"] == 1).sum()) > 300  # the AO threshold early-out


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_oracle_dda_matches_independent_witness(kind):
    """(ray, mask) -> (reported, t, voxel): bit-exact for all 12 000 rays of each intersection shader."""
    fx = load("dda_pairs.npz")
    l = O.lib()
    o, d, lo, hi, tmin = fx["o"], fx["d"], fx["mask_lo"], fx["mask_hi"], fx["tmin"]
    want_rep, want_t, want_vox, want_kind = fx[f"reported{kind}"], fx[f"t{kind}"], fx[f"voxel{kind}"], fx[f"hitkind{kind}"]
    t, vox, hk = C.c_float(), C.c_uint32(), C.c_int()
    bad = []
    for i in range(len(o)):
        if kind == 2:
            rep = l.orc_dda_rough(O.f3(o[i]), O.f3(d[i]), int(lo[i]), int(hi[i]), C.byref(t))
            got = (bool(rep), t.value if rep else 0.0, 0, 0)
        else:
            rep = l.orc_dda(kind, O.f3(o[i]), O.f3(d[i]), int(lo[i]), int(hi[i]), float(tmin[i]), C.byref(t), C.byref(vox), C.byref(hk))
            got = (bool(rep), t.value if rep else 0.0, vox.value & 0xFF if rep else 0, hk.value if rep else 0)
        want = (bool(want_rep[i]), float(want_t[i]), int(want_vox[i]), int(want_kind[i]))
        same_t = np.float32(got[1]).tobytes() == np.float32(want[1]).tobytes() or (got[1] == want[1])
        if got[0] != want[0] or not same_t or got[2] != want[2] or got[3] != want[3]:
            bad.append((i, int(fx["category"][i]), got, want))
    assert not bad, f"{len(bad)} of {len(o)} differ, first: {bad[:3]}"


def test_oracle_codecs_match_independent_witness():
    cd = load("codecs.npz")
    l = O.lib()
    out3, out4 = (C.c_float * 3)(), (C.c_float * 4)()
    # LogLuv32 encode: exact except where the log-luminance / chroma sits on a quantisation step (flagged): one step there
    enc = np.array([l.orc_logluv_encode(O.f3(v)) for v in cd["logluv_rgb"]], np.uint32)
    want, border = cd["logluv_packed"], cd["logluv_borderline"]
    assert np.array_equal(enc[~border], want[~border])
    for a, b in zip(enc[border], want[border]):
        assert abs(int(a >> 18) - int(b >> 18)) <= 1 and abs(int((a >> 9) & 511) - int((b >> 9) & 511)) <= 1 and abs(int(a & 511) - int(b & 511)) <= 1
    # decode: pow(2, x) is the only transcendental; everything else is exact arithmetic
    dec = np.zeros((len(cd["logluv_words"]), 3), np.float32)
    for i, w in enumerate(cd["logluv_words"]):
        l.orc_logluv_decode(int(w), out3)
        dec[i] = out3[:]
    scale = np.abs(cd["logluv_decoded"]).max(axis=1, keepdims=True)  # the matrix product cancels: bound by the vector's size
    assert (np.abs(dec - cd["logluv_decoded"]) <= 4e-6 * scale).all()
    # NRD normal pack -> A2B10G10R10 texel, and unpack of arbitrary texels: bit-exact
    packed = np.zeros(len(cd["normal_in"]), np.uint32)
    for i, (n, mid) in enumerate(zip(cd["normal_in"], cd["normal_material_id"])):
        l.orc_nrd_pack_normal(O.f3(n), 1.0, float(mid), out4)
        packed[i] = l.orc_pack_rgb10a2(out4)
    assert np.array_equal(packed, cd["normal_packed"])
    un = np.zeros((len(cd["normal_texels"]), 3), np.float32)
    v4 = (C.c_float * 4)()
    for i, p in enumerate(cd["normal_texels"]):
        l.orc_unpack_rgb10a2(int(p), v4)
        l.orc_nrd_unpack_normal(v4, out3)
        un[i] = out3[:]
    assert un.tobytes() == cd["normal_texels_unpacked"].tobytes()
    # RGB10A2 packing incl. clamps and NaN
    got = np.array([l.orc_pack_rgb10a2((C.c_float * 4)(*[float(x) for x in v])) for v in cd["rgb10a2_in"]], np.uint32)
    assert np.array_equal(got, cd["rgb10a2_packed"])
    # fp16 conversion as the radiance planes store it
    Y = cd["radiance_in"] @ np.array([0.25, 0.5, 0.25], np.float32)
    h = np.array([l.orc_f32_to_f16(float(np.float32(y))) for y in Y.astype(np.float32)], np.uint16)
    assert np.array_equal(h, Y.astype(np.float32).astype(np.float16).view(np.uint16))
    # CubedNormalize, face ids, rotateVectorByNormal
    cub = np.zeros_like(cd["cubed_in"])
    for i, v in enumerate(cd["cubed_in"]):
        l.orc_cubed_normalize(O.f3(v), out3)
        cub[i] = out3[:]
    assert np.array_equal(cub, cd["cubed_out"])
    assert [l.orc_normal2faceid(O.f3(v)) for v in cd["face_in"]] == cd["face_id"].tolist()
    rot = np.zeros_like(cd["rotate_out"])
    for i, (n, t) in enumerate(zip(cd["rotate_normal"], cd["rotate_target"])):
        l.orc_rotate_by_normal(O.f3(n), O.f3(t), out3)
        rot[i] = out3[:]
    assert rot.tobytes() == cd["rotate_out"].tobytes()


def _check_blocks(got_b, got_m, want_b, want_m, border, who):
    assert len(got_b) == len(want_b), who
    for f in ("x", "y", "z", "w", "mask", "material_ptr"):
        assert np.array_equal(got_b[f], want_b[f]), (who, f)
    assert np.array_equal(got_m, want_m), who
    # avg_albedo: (powf(x, 1/2.4) * 1023) as u32 truncates; where a channel sits within 1e-3 of an integer (flagged by the
    # generator) a last-ulp difference between powf implementations may move it by one
    exact = ~border
    assert np.array_equal(got_b["avg_albedo"][exact], want_b["avg_albedo"][exact]), who
    for a, b in zip(got_b["avg_albedo"][border], want_b["avg_albedo"][border]):
        for sh, m in ((22, 1023), (12, 1023), (2, 1023), (0, 3)):
            assert abs(int((a >> sh) & m) - int((b >> sh) & m)) <= 1, who


def test_from_tree_matches_independent_witness():
    """Block records (position, mask, material_ptr, avg_albedo) and the material stream for six random models, one of
    them with repeated XYZI entries (ModelIndexCollector::set counts every call, collector.rs:23-34)."""
    ft = load("from_tree.npz")
    pal = ft["palette"]
    for k in range(int(ft["n_models"][0])):
        size, xyzi = ft[f"m{k}_size"], ft[f"m{k}_xyzi"]
        want_b, want_m, border = ft[f"m{k}_blocks"], ft[f"m{k}_materials"], ft[f"m{k}_borderline"]
        ob, om = O.model_build(xyzi, size, pal)
        _check_blocks(ob, om[:len(want_m)], want_b, want_m, border, f"oracle m{k}")
        pb, pm = api.flatten_model(xyzi, tuple(int(v) for v in size), pal)
        _check_blocks(pb, pm[:len(want_m)], want_b, want_m, border, f"product m{k}")
