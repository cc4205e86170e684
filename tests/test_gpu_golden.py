"""The HIP device functions against the INDEPENDENT numpy witness in tests/golden/ (make_shader_fixtures.py, SURVEY 8c ii-iv),
through the C ABI (dust_hip_device_eval runs the same inlined bodies the frame kernels use). tests/test_golden_fixtures.py checks
the C oracle against the same files on the CPU, so oracle, device code and the witness are pairwise pinned."""
import os

import numpy as np
import pytest

from dust_amd import api

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def ctx():
    return api.Context(device=0)


def words(*cols):
    """columns (float32 / uint32 arrays, 1-D or 2-D) -> (n, k) uint32 rows by bit pattern"""
    parts = []
    for c in cols:
        c = np.asarray(c)
        c = c.reshape(len(c), -1)
        parts.append(c.astype(np.float32).view(np.uint32) if c.dtype.kind == "f" else c.astype(np.uint32))
    return np.ascontiguousarray(np.concatenate(parts, axis=1))


@pytest.mark.parametrize("kind", [0, 1, 2])
def test_device_dda_matches_independent_witness(ctx, kind):
    """hit.rint / ambient_occlusion.rint / rough.rint on 12 000 (ray, mask) pairs each: reported, t and voxel bit-exact."""
    fx = np.load(os.path.join(GOLD, "dda_pairs.npz"))
    out = ctx.device_eval(kind, words(fx["o"], fx["d"], fx["tmin"], fx["mask_lo"], fx["mask_hi"]), 3)
    rep = fx[f"reported{kind}"]
    assert np.array_equal(out[:, 0] != 0, rep)
    want_t = np.where(rep, fx[f"t{kind}"], np.float32(0)).astype(np.float32)
    got_t = out[:, 1].view(np.float32)
    same = (out[:, 1] == want_t.view(np.uint32)) | (got_t == want_t)  # +0 / -0 compare equal
    assert same.all(), f"{int((~same).sum())} hit distances differ, first at {int(np.flatnonzero(~same)[0])}"
    assert np.array_equal(out[:, 2] & 0xFF, np.where(rep, fx[f"voxel{kind}"], 0))


def test_device_codecs_match_independent_witness(ctx):
    cd = np.load(os.path.join(GOLD, "codecs.npz"))
    # LogLuv32: the device takes log2 / exp2 from v_log_f32 / v_exp_f32 (1 ulp): one quantisation step of slack on the
    # luminance field where the witness flags the input as sitting on a step, exact elsewhere for chroma
    enc = ctx.device_eval(3, words(cd["logluv_rgb"]), 1)[:, 0]
    want = cd["logluv_packed"]
    dl = np.abs((enc >> 18).astype(np.int64) - (want >> 18).astype(np.int64))
    du = np.abs(((enc >> 9) & 511).astype(np.int64) - ((want >> 9) & 511).astype(np.int64))
    dv = np.abs((enc & 511).astype(np.int64) - (want & 511).astype(np.int64))
    zero = want == 0
    assert np.array_equal(enc[zero & ~cd["logluv_borderline"]], want[zero & ~cd["logluv_borderline"]])
    assert dl[~zero].max() <= 1 and du[~zero].max() <= 1 and dv[~zero].max() <= 1
    assert (dl[~cd["logluv_borderline"] & ~zero] == 0).mean() > 0.995  # off the steps the fields agree
    dec = ctx.device_eval(4, words(cd["logluv_words"]), 3).view(np.float32)
    scale = np.abs(cd["logluv_decoded"]).max(axis=1, keepdims=True)
    assert (np.abs(dec - cd["logluv_decoded"]) <= 2e-5 * scale).all()
    # exact formats
    assert np.array_equal(ctx.device_eval(5, words(cd["normal_in"], cd["normal_material_id"]), 1)[:, 0], cd["normal_packed"])
    assert ctx.device_eval(6, words(cd["normal_texels"]), 3).view(np.float32).tobytes() == cd["normal_texels_unpacked"].tobytes()
    half4 = ctx.device_eval(7, words(cd["radiance_in"], cd["radiance_hitdist"]), 2)
    assert np.array_equal(half4.view(np.uint16).reshape(-1, 4), cd["radiance_half4"])
    un = ctx.device_eval(8, np.ascontiguousarray(cd["radiance_half4"]).view(np.uint32).reshape(-1, 2), 4).view(np.float32)
    assert un.tobytes() == cd["radiance_unpacked"].tobytes()
    assert np.array_equal(ctx.device_eval(9, words(cd["rgb10a2_in"]), 1)[:, 0], cd["rgb10a2_packed"])
    cub = ctx.device_eval(10, words(cd["cubed_in"]), 4)
    assert np.array_equal(cub[:, :3].view(np.float32), cd["cubed_out"])
    faces = ctx.device_eval(10, words(cd["face_in"]), 4)
    assert faces[:, 3].tolist() == cd["face_id"].tolist()
    rot = ctx.device_eval(11, words(cd["rotate_normal"], cd["rotate_target"]), 3).view(np.float32)
    assert rot.tobytes() == cd["rotate_out"].tobytes()


def test_device_radix_sort_is_a_stable_sort(ctx):
    """radix.hip (the surfel pass's position order and the deterministic apply's cluster order) against numpy's stable sort:
    sizes around the 2048-item tile and the 512-item wave run, heavy duplicates, all 32 key bits."""
    rng = np.random.default_rng(12)
    for n, hi in ((1, 10), (63, 4), (512, 1 << 32), (2048, 300), (2049, 1 << 16), (5000, 7), (345_600, 1 << 25), (100_003, 1 << 32)):
        keys = rng.integers(0, hi, n, dtype=np.uint64).astype(np.uint32)
        rows = np.stack([keys, np.arange(n, dtype=np.uint32)], axis=1)
        out = ctx.device_eval(12, np.ascontiguousarray(rows), 2)
        order = np.argsort(keys, kind="stable")
        assert np.array_equal(out[:, 0], keys[order]) and np.array_equal(out[:, 1], order.astype(np.uint32)), (n, hi)


def test_tile_order_is_a_stable_sort_by_cost_class(ctx):
    """k_tile_order (kernels.hip; what dust_hip_render_frame hands tiles out by) against numpy: per band (8 contiguous bands of the
    tile list) a stable sort into quarter-octave classes below the band's most expensive tile, most expensive class first,
    never-timed tiles (cost 0) and everything 8 octaves down in the last class. Costs sit mid-class so that the device's
    v_log_f32 and numpy agree on every class."""
    rng = np.random.default_rng(13)
    for n in (1, 7, 8, 64, 1000, 32400, 129_600, 300_001):
        k = rng.integers(40, 121, n)
        cost = np.floor(np.exp2((k + 0.5) / 4.0)).astype(np.uint32)
        cost[rng.random(n) < 0.1] = 0
        if n > 100:
            cost[: n // 16] = 0  # a stretch nobody timed
        out = ctx.device_eval(13, np.ascontiguousarray(cost.reshape(-1, 1)), 1).reshape(-1)
        q = np.where(cost == 0, 0, 1 + k)
        per = (n + 7) // 8
        expect = np.empty(n, np.uint32)
        for b in range(8):
            lo, hi = min(b * per, n), min((b + 1) * per, n)
            if lo == hi:
                continue
            qb = q[lo:hi]
            top = qb.max()
            digit = np.where(qb == 0, 31, np.minimum(top - qb, 31))
            expect[lo:hi] = lo + np.argsort(digit, kind="stable")
        assert np.array_equal(out, expect), n


def test_tile_order_with_cost_balanced_bands(ctx):
    """device function 14 = k_tile_order as a frame uses it: the tile list is cut into 8 contiguous bands of about equal COST (not equal
    count), and inside each band the stable class sort of test_tile_order_is_a_stable_sort_by_cost_class. Against numpy: the cuts
    are where the running cost passes k/8 of the total, every band holds exactly its own tiles, and its order is that sort."""
    rng = np.random.default_rng(14)
    for n in (9, 100, 32400, 129_600):
        k = rng.integers(40, 101, n)
        cost = np.floor(np.exp2((k + 0.5) / 4.0)).astype(np.uint32)
        cost[n // 3: n // 2] = np.floor(np.exp2((rng.integers(40, 60, n // 2 - n // 3) + 0.5) / 4.0)).astype(np.uint32)   # a cheap stretch (sky)
        if n > 100:
            cost[rng.random(n) < 0.05] = 0
        out = ctx.device_eval(14, np.ascontiguousarray(cost.reshape(-1, 1)), 2)
        order, cuts = out[:, 0], [int(v) for v in out[:9, 1]]
        assert cuts[0] == 0 and cuts[8] == n and all(a <= b for a, b in zip(cuts, cuts[1:])), cuts
        eff = np.clip(cost, 1, 1 << 22).astype(np.int64)   # (a tile nobody has timed counts as cheap; the scan caps a tile at 2^22 cycles)
        run = np.concatenate([[0], np.cumsum(eff)])
        for b in range(1, 8):   # cut b = one past the first tile at which the running cost reaches b/8 of the total
            target = run[-1] // 8 * b
            want = int(np.searchsorted(run[1:], target, side="left")) + 1 if target > 0 else 0
            assert cuts[b] == min(want, n), (n, b, cuts[b], want)
        if n >= 32400:
            sums = np.array([run[cuts[b + 1]] - run[cuts[b]] for b in range(8)], np.float64)
            assert sums.max() / sums.mean() < 1.01, sums
        q = np.where(cost == 0, 0, 1 + np.floor(4.0 * np.log2(np.maximum(cost, 1).astype(np.float64))).astype(np.int64))
        for b in range(8):
            lo, hi = cuts[b], cuts[b + 1]
            if lo == hi:
                continue
            qb = q[lo:hi]
            digit = np.where(qb == 0, 31, np.minimum(qb.max() - qb, 31))
            assert np.array_equal(order[lo:hi], lo + np.argsort(digit, kind="stable")), (n, b)
