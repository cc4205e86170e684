#!/usr/bin/env python3
"""Second witness for the parts of the path nothing in the reference pins (SURVEY 8c ii-iv).

An INDEPENDENT numpy restatement -- it shares no code with oracle/*.c or the HIP kernels and binds neither
library -- of
  * the three intersection shaders' per-brick routine: primary/hit.rint:43-131, final_gather/ambient_occlusion.rint:46-134,
    final_gather/rough.rint:42-59 (all through intersectAABB, hit.rint:20-28, and encode_index, hit.rint:30-32);
  * the storage codecs: LogLuv32 (headers/spatial_hash.glsl:28-93), NRD octahedral normal + RGB10A2
    (headers/nrd.glsl:2-10,25-52,54-94), YCoCg + fp16 radiance (nrd.glsl:97-147), face ids / CubedNormalize /
    rotateVectorByNormal (headers/normal.glsl:9-43);
  * load_model + ModelIndexCollector + VoxGeometry::from_tree (crates/vox/src/loader.rs:244-274, collector.rs:23-88,
    geometry.rs:68-128) for small random models,
written vectorised over whole arrays (the C oracle is scalar, the kernels per-lane), evaluated in numpy float32 (one IEEE
rounding per operation, no contraction). It emits the fixtures tests/golden/{dda_pairs,codecs,from_tree}.npz; the tests check
the C oracle AND the HIP device functions against them (tests/test_golden_fixtures.py, tests/test_gpu_golden.py).

Semantics fixed where GLSL leaves them open -- the same conventions oracle/shade.c states, restated here so the witness is
explicit about them: min/max are IEEE minNum/maxNum (np.fmin/np.fmax: a NaN operand is ignored), comparisons with NaN are
false (so step(edge, NaN) == 1), sign(+-0) == 0, float -> UNORM and fp16 stores round to nearest even.
Rays whose evaluation leaves GLSL-defined territory (voxel position walking out of 0..3 before the exit test fires, shifts
by >= 32, NaN into an int conversion) are dropped from the fixture and counted in `dropped`.

Run:  python tests/golden/make_shader_fixtures.py      (deterministic: fixed seeds; needs only numpy)
"""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F = np.float32


# --------------------------------------------------------------------------------------------- intersection shaders
def intersect_aabb04(o, d):
    """hit.rint:20-28 with box [0,4]^3. o, d: (N,3) f32 -> (t_min, t_max)."""
    with np.errstate(all="ignore"):
        lo = (F(0.0) - o) / d
        hi = (F(4.0) - o) / d
    t1 = np.fmin(lo, hi)
    t2 = np.fmax(lo, hi)
    t_min = np.fmax(np.fmax(t1[:, 0], t1[:, 1]), t1[:, 2])
    t_max = np.fmin(np.fmin(t2[:, 0], t2[:, 1]), t2[:, 2])
    return t_min.astype(F), t_max.astype(F)


def glsl_sign(x):
    return (x > 0).astype(F) - (x < 0).astype(F)


def glsl_step(edge, x):  # 0 if x < edge else 1
    return np.where(x < edge, F(0.0), F(1.0)).astype(F)


def grid_clear(m_lo, m_hi, hit):
    """GridCheck without SHADER_INT_64 (hit.rint:13-15): true when the voxel bit is NOT set. hit: int array in 0..63."""
    bit_lo = (m_lo >> (hit & 31).astype(np.uint32)) & np.uint32(1)
    bit_hi = (m_hi >> ((hit - 32) & 31).astype(np.uint32)) & np.uint32(1)
    return np.where(hit < 32, bit_lo, bit_hi) == 0


def dda(kind, o, d, m_lo, m_hi, tmin):
    """kind 0: primary/hit.rint dda(); kind 1: ambient_occlusion.rint dda(); kind 2: rough.rint dda().
    Returns reported (bool), t (f32), voxel (u8: the value written to hitAttributes.voxelId when the guard passes),
    hitkind (u8: second argument of reportIntersectionEXT), defined (bool: evaluation stayed inside GLSL-defined behaviour)."""
    n = o.shape[0]
    reported = np.zeros(n, bool)
    t_out = np.zeros(n, F)
    voxel = np.zeros(n, np.uint8)
    hitkind = np.zeros(n, np.uint8)
    defined = np.ones(n, bool)
    t0, t1 = intersect_aabb04(o, d)
    alive = ~(t0 >= t1)  # hit.rint:49-52
    nonempty = (m_lo != 0) | (m_hi != 0)
    if kind == 2:  # rough.rint:53-59
        rep = alive & nonempty
        reported[rep] = True
        t_out[rep] = t0[rep]
        return reported, t_out, voxel, hitkind, defined
    alive &= ~(t1 <= F(0.0))  # hit.rint:53-55
    if kind == 1:  # ambient_occlusion.rint:59-70
        thr = alive & (t0 <= F(8.0)) & (F(8.0) <= t1)
        rep = thr & nonempty
        reported[rep] = True
        t_out[rep] = t0[rep]
        voxel[rep] = 0xFF
        hitkind[rep] = 1
        alive &= ~thr
    with np.errstate(all="ignore"):
        hd = np.fmax(t0, tmin).astype(F)  # hit.rint:67
        p = (o + (d * hd[:, None]).astype(F)).astype(F)
        fl = np.floor(p)
        defined &= ~(alive & np.isnan(fl).any(axis=1))
        pos = np.clip(np.nan_to_num(fl, nan=0.0, posinf=3.0, neginf=0.0), 0, 3).astype(np.int32)  # clamp(ivec3(floor(p)), 0, 3)
        step = glsl_sign(d)
        t_coef = (F(1.0) / d).astype(F)  # hit.rint:88
        t_bias = (t_coef * o).astype(F)
        t_max = (((pos.astype(F) + np.fmax(step, F(0.0))).astype(F) * t_coef).astype(F) - t_bias).astype(F)
        t_delta = ((F(1.0) * t_coef).astype(F) * step).astype(F)
    hit = (pos[:, 0] << 4) | (pos[:, 1] << 2) | pos[:, 2]
    looping = alive & grid_clear(m_lo, m_hi, hit)
    done_hit = alive & ~looping  # start voxel is solid
    for _ in range(64):
        if not looping.any():
            break
        with np.errstate(all="ignore"):
            tx, ty, tz = t_max[:, 0], t_max[:, 1], t_max[:, 2]
            comp = np.stack([glsl_step(tx, tz) * glsl_step(tx, ty),   # step(tMax.xyz, tMax.zxy) * step(tMax.xyz, tMax.yzx)
                             glsl_step(ty, tx) * glsl_step(ty, tz),
                             glsl_step(tz, ty) * glsl_step(tz, tx)], axis=1).astype(F)
            delta = (step * comp).astype(np.int32)  # i8vec3(STEP * compResult): 0 * 1 / +-1 * 1 / +-1 * 0, always finite
            pos = np.where(looping[:, None], pos + delta, pos)
            hd_new = np.fmin(np.fmin(tx, ty), tz).astype(F)
            hd = np.where(looping, hd_new, hd)
            leave = looping & ((hd + F(0.001)).astype(F) >= t1)  # hit.rint:107-109: no report
            looping = looping & ~leave
            t_max = np.where(looping[:, None], (t_max + (t_delta * comp).astype(F)).astype(F), t_max)
        out_of_range = looping & ((pos < 0) | (pos > 3)).any(axis=1)
        defined &= ~out_of_range
        looping &= ~out_of_range
        hit = np.where(looping, ((pos[:, 0] & 3) << 4) | ((pos[:, 1] & 3) << 2) | (pos[:, 2] & 3), hit)
        still = looping & grid_clear(m_lo, m_hi, hit)
        done_hit |= looping & ~still
        looping = still
    defined &= ~looping  # did not terminate within 64 steps (NaN rays)
    reported[done_hit] = True
    t_out[done_hit] = (hd / F(1.0))[done_hit]
    voxel[done_hit] = hit[done_hit].astype(np.uint8)
    return reported, t_out, voxel, hitkind, defined


def make_rays(rng):
    """(o, d, mask_lo, mask_hi, tmin, category) -- brick-local rays of every class the shaders can meet."""
    os_, ds_, cats = [], [], []

    def add(o, d, cat):
        os_.append(o.astype(F)); ds_.append(d.astype(F)); cats.append(np.full(len(o), cat, np.uint8))

    n = 5000  # 0: general position, unnormalised directions through the brick (and some that miss it)
    o = rng.uniform(-6, 10, (n, 3))
    tgt = rng.uniform(-0.5, 4.5, (n, 3))
    d = (tgt - o) * rng.uniform(0.2, 3.0, (n, 1))
    add(o, d, 0)
    n = 1500  # 1: normalised directions (secondary rays)
    o = rng.uniform(-6, 10, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    add(o, d, 1)
    n = 2000  # 2: origin inside the brick
    o = rng.uniform(0, 4, (n, 3))
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    add(o, d, 2)
    n = 1200  # 3: axis-parallel in one or two components (1/0 = inf, inf * 0 = NaN: the minNum / NaN-compare conventions decide)
    o = rng.uniform(-3, 7, (n, 3))
    d = rng.normal(size=(n, 3))
    z = rng.integers(0, 3, n)
    d[np.arange(n), z] = 0.0
    two = rng.random(n) < 0.3
    d[np.arange(n)[two], (z[two] + 1) % 3] = 0.0
    neg0 = rng.random(n) < 0.2  # -0.0 components as well
    d[neg0] = np.where(d[neg0] == 0.0, -0.0, d[neg0])
    add(o, d, 3)
    n = 1300  # 4: lattice origins and diagonal directions: exact ties in tMax (several axes step together)
    o = rng.integers(-8, 16, (n, 3)) * 0.5
    d = rng.choice([-1.0, 1.0], (n, 3)) * (2.0 ** rng.integers(-2, 3, (n, 1)))
    flat = rng.random(n) < 0.3
    d[flat, rng.integers(0, 3, flat.sum())] *= 2.0
    add(o, d, 4)
    n = 1000  # 5: the ambient-occlusion threshold: normalised rays whose brick interval straddles t = 8
    d = rng.normal(size=(n, 3)); d /= np.linalg.norm(d, axis=1, keepdims=True)
    inside = rng.uniform(0.2, 3.8, (n, 3))
    o = inside - d * rng.uniform(6.5, 9.5, (n, 1))
    add(o, d, 5)
    o = np.concatenate(os_); d = np.concatenate(ds_); cat = np.concatenate(cats)
    n = len(o)
    dens = rng.choice([1 / 64, 0.1, 0.5, 0.9], n)
    bits = rng.random((n, 64)) < dens[:, None]
    empty = rng.random(n) < 0.05
    bits[empty] = False
    w = (1 << np.arange(32, dtype=np.uint64))
    m_lo = (bits[:, :32] * w).sum(axis=1).astype(np.uint32)
    m_hi = (bits[:, 32:] * w).sum(axis=1).astype(np.uint32)
    tmin = rng.choice([0.0, 0.1, 8.0], n, p=[0.3, 0.5, 0.2]).astype(F)
    rnd = rng.random(n) < 0.2
    tmin[rnd] = rng.uniform(0, 6, rnd.sum()).astype(F)
    return o, d, m_lo, m_hi, tmin, cat


def build_dda_fixture():
    rng = np.random.default_rng(0xD0571)
    o, d, m_lo, m_hi, tmin, cat = make_rays(rng)
    out = {}
    keep = np.ones(len(o), bool)
    res = {}
    for kind in (0, 1, 2):
        res[kind] = dda(kind, o, d, m_lo, m_hi, tmin)
        keep &= res[kind][4]
    dropped = int((~keep).sum())
    for kind in (0, 1, 2):
        rep, t, vox, hk, _ = res[kind]
        out[f"reported{kind}"] = rep[keep]
        out[f"t{kind}"] = t[keep]
        out[f"voxel{kind}"] = vox[keep]
        out[f"hitkind{kind}"] = hk[keep]
    out.update(o=o[keep], d=d[keep], mask_lo=m_lo[keep], mask_hi=m_hi[keep], tmin=tmin[keep], category=cat[keep],
               dropped=np.array([dropped], np.int64))
    return out


# --------------------------------------------------------------------------------------------- codecs
M_ACESCG2XYZ = np.array([[0.66245437, 0.2722288, -0.0055746622], [0.13400422, 0.6740818, 0.00406073],
                         [0.15618773, 0.05368953, 1.0103393]], F)  # GLSL mat3 constructor: COLUMNS (spatial_hash.glsl:12-19)
M_XYZ2ACESCG = np.array([[1.6410228, -0.66366285, 0.011721907], [-0.32480323, 1.6153315, -0.0082844375],
                         [-0.23642465, 0.016756356, 0.9883947]], F)


def mat3_mul(cols, v):
    """GLSL `mat3 * vec3` with the mat3 given as its constructor's three columns; evaluated as ((c0*x + c1*y) + c2*z) in f32."""
    x, y, z = v[:, 0:1], v[:, 1:2], v[:, 2:3]
    return (((cols[0] * x).astype(F) + (cols[1] * y).astype(F)).astype(F) + (cols[2] * z).astype(F)).astype(F)


def logluv_encode(rgb):
    """spatial_hash.glsl:28-60. Returns (packed u32, borderline bool): borderline marks inputs whose log-luminance or chroma
    lands within 2e-3 of an integer step (log2 is a transcendental: implementations may differ in the last ulp there)."""
    xyz = mat3_mul(M_ACESCG2XYZ, rgb)
    with np.errstate(all="ignore"):
        log2y = np.log2(xyz[:, 1].astype(np.float64)).astype(F)  # correctly rounded f32 log2
        logy = (F(409.6) * (log2y + F(20.0)).astype(F)).astype(F)
    cl = np.fmin(np.fmax(logy, F(0.0)), F(16383.0))
    le = np.where(np.isnan(cl), 0, cl).astype(np.uint32)
    with np.errstate(all="ignore"):
        denom = ((F(-2.0) * xyz[:, 0]).astype(F) + (F(12.0) * xyz[:, 1]).astype(F)).astype(F)
        denom = (denom + (F(3.0) * ((xyz[:, 0] + xyz[:, 1]).astype(F) + xyz[:, 2]).astype(F)).astype(F)).astype(F)
        inv = (F(1.0) / denom).astype(F)
        u = ((F(4.0) * xyz[:, 0]).astype(F) * inv).astype(F)
        v = ((F(9.0) * xyz[:, 1]).astype(F) * inv).astype(F)
        cu = np.fmin(np.fmax((F(820.0) * u).astype(F), F(0.0)), F(511.0))
        cv = np.fmin(np.fmax((F(820.0) * v).astype(F), F(0.0)), F(511.0))
    ue = np.where(np.isnan(cu), 0, cu).astype(np.uint32)
    ve = np.where(np.isnan(cv), 0, cv).astype(np.uint32)
    packed = np.where(le == 0, np.uint32(0), (le << 18) | (ue << 9) | ve).astype(np.uint32)
    frac = lambda a: np.abs(a - np.rint(a))
    with np.errstate(all="ignore"):
        border = (frac(logy.astype(np.float64)) < 2e-3) | (frac(cu.astype(np.float64)) < 2e-3) | (frac(cv.astype(np.float64)) < 2e-3)
    return packed, border | ~np.isfinite(logy)


def logluv_decode(p):
    """spatial_hash.glsl:64-93 (pow(2, x) evaluated in double and rounded once)."""
    le = p >> 18
    logy = (((le.astype(F) + F(0.5)).astype(F) / F(409.6)).astype(F) - F(20.0)).astype(F)
    Y = np.exp2(logy.astype(np.float64)).astype(F)
    u = (((p >> 9) & 0x1FF).astype(F) + F(0.5)).astype(F) / F(820.0)
    v = ((p & 0x1FF).astype(F) + F(0.5)).astype(F) / F(820.0)
    u = u.astype(F); v = v.astype(F)
    inv = (F(1.0) / (((F(6.0) * u).astype(F) - (F(16.0) * v).astype(F)).astype(F) + F(12.0)).astype(F)).astype(F)
    x = ((F(9.0) * u).astype(F) * inv).astype(F)
    y = ((F(4.0) * v).astype(F) * inv).astype(F)
    s = (Y / y).astype(F)
    xyz = np.stack([(s * x).astype(F), Y, (s * ((F(1.0) - x).astype(F) - y).astype(F)).astype(F)], axis=1)
    rgb = np.fmax(mat3_mul(M_XYZ2ACESCG, xyz), F(0.0))
    rgb[le == 0] = 0.0
    return rgb.astype(F)


def unorm(v, scale):
    """float -> UNORM field, round to nearest even; NaN and negatives -> 0 (Vulkan fixed-point conversion rules)."""
    with np.errstate(all="ignore"):
        c = np.where(v > 0, np.fmin(v, F(1.0)), F(0.0)).astype(F)
        return np.rint((c * F(scale)).astype(F)).astype(np.uint32)


def pack_rgb10a2(v):  # A2B10G10R10_UNORM_PACK32
    return (unorm(v[:, 0], 1023) | (unorm(v[:, 1], 1023) << 10) | (unorm(v[:, 2], 1023) << 20) | (unorm(v[:, 3], 3) << 30)).astype(np.uint32)


def nrd_pack_normal(n, roughness, material_id):
    """nrd.glsl:2-10,25-52 with NRD_NORMAL_ENCODING_R10G10B10A2_UNORM, stored to an A2B10G10R10 image."""
    s = ((np.abs(n[:, 0]) + np.abs(n[:, 1])).astype(F) + np.abs(n[:, 2])).astype(F)  # dot(abs(v), vec3(1))
    v = (n / s[:, None]).astype(F)
    wrap_x = ((F(1.0) - np.abs(v[:, 1])).astype(F) * ((glsl_step(F(0.0), v[:, 0]) * F(2.0)).astype(F) - F(1.0)).astype(F)).astype(F)
    wrap_y = ((F(1.0) - np.abs(v[:, 0])).astype(F) * ((glsl_step(F(0.0), v[:, 1]) * F(2.0)).astype(F) - F(1.0)).astype(F)).astype(F)
    ex = np.where(v[:, 2] >= 0, v[:, 0], wrap_x)
    ey = np.where(v[:, 2] >= 0, v[:, 1], wrap_y)
    px = ((ex * F(0.5)).astype(F) + F(0.5)).astype(F)
    py = ((ey * F(0.5)).astype(F) + F(0.5)).astype(F)
    pw = np.fmin(np.fmax((material_id / F(3.0)).astype(F), F(0.0)), F(1.0))
    return pack_rgb10a2(np.stack([px, py, np.full_like(px, roughness), pw], axis=1))


def nrd_unpack_normal(p):
    """nrd.glsl:54-94 on a texel read back from the A2B10G10R10 image (UNORM -> float is k / 1023)."""
    px = ((((p & 1023).astype(F) / F(1023.0)).astype(F) * F(2.0)).astype(F) - F(1.0)).astype(F)
    py = (((((p >> 10) & 1023).astype(F) / F(1023.0)).astype(F) * F(2.0)).astype(F) - F(1.0)).astype(F)
    nz = ((F(1.0) - np.abs(px)).astype(F) - np.abs(py)).astype(F)
    t = np.fmin(np.fmax(-nz, F(0.0)), F(1.0))
    nx = (px - (t * ((glsl_step(F(0.0), px) * F(2.0)).astype(F) - F(1.0)).astype(F)).astype(F)).astype(F)
    ny = (py - (t * ((glsl_step(F(0.0), py) * F(2.0)).astype(F) - F(1.0)).astype(F)).astype(F)).astype(F)
    ln = np.sqrt((((nx * nx).astype(F) + (ny * ny).astype(F)).astype(F) + (nz * nz).astype(F)).astype(F)).astype(F)
    return np.stack([(nx / ln).astype(F), (ny / ln).astype(F), (nz / ln).astype(F)], axis=1)


def pack_radiance(r, hitdist):
    """nrd.glsl:127-147 stored to an RGBA16F image: four fp16 bit patterns (round to nearest even)."""
    hd = np.where(hitdist != 0, np.fmax(hitdist, F(1e-7)), hitdist).astype(F)
    Y = (((r[:, 0] * F(0.25)).astype(F) + (r[:, 1] * F(0.5)).astype(F)).astype(F) + (r[:, 2] * F(0.25)).astype(F)).astype(F)
    Co = (((r[:, 0] * F(0.5)).astype(F) + (r[:, 1] * F(0.0)).astype(F)).astype(F) + (r[:, 2] * F(-0.5)).astype(F)).astype(F)
    Cg = (((r[:, 0] * F(-0.25)).astype(F) + (r[:, 1] * F(0.5)).astype(F)).astype(F) + (r[:, 2] * F(-0.25)).astype(F)).astype(F)
    with np.errstate(over="ignore"):
        return np.stack([Y, Co, Cg, hd], axis=1).astype(np.float16).view(np.uint16)


def unpack_radiance(h):
    """nrd.glsl:107-125 on four fp16 values."""
    f = h.view(np.float16).astype(F)
    t = (f[:, 0] - f[:, 2]).astype(F)
    return np.stack([np.fmax((t + f[:, 1]).astype(F), F(0.0)), np.fmax((f[:, 0] + f[:, 2]).astype(F), F(0.0)),
                     np.fmax((t - f[:, 1]).astype(F), F(0.0)), f[:, 3]], axis=1).astype(F)


def cubed_normalize(d):  # normal.glsl:39-43
    a = np.abs(d)
    mx = np.fmax(a[:, 0], np.fmax(a[:, 1], a[:, 2]))
    return (glsl_sign(d) * glsl_step(mx[:, None], a)).astype(F)


def normal2faceid(n):  # normal.glsl:9-18 (GLSL round(): half cases do not occur for 0 / +-1 inputs)
    s = np.fmin(np.fmax(((n[:, 0] + n[:, 1]).astype(F) + n[:, 2]).astype(F), F(0.0)), F(1.0))
    return (np.rint(s).astype(np.uint32) + np.rint(np.abs(n[:, 2])).astype(np.uint32) * 4 + np.rint(np.abs(n[:, 1])).astype(np.uint32) * 2) & 0xFF


def rotate_by_normal(n, t):  # normal.glsl:31-37
    q = np.stack([-n[:, 1], n[:, 0], np.zeros(len(n), F), (F(1.0) + n[:, 2]).astype(F)], axis=1).astype(F)
    l2 = ((((q[:, 0] * q[:, 0]).astype(F) + (q[:, 1] * q[:, 1]).astype(F)).astype(F) + (q[:, 2] * q[:, 2]).astype(F)).astype(F)
          + (q[:, 3] * q[:, 3]).astype(F)).astype(F)
    with np.errstate(all="ignore"):
        q = (q / np.sqrt(l2).astype(F)[:, None]).astype(F)
    flip = n[:, 2] < F(-0.99999)
    q[flip] = np.array([-1.0, 0.0, 0.0, 0.0], F)
    qv, qw = q[:, :3], q[:, 3]

    def dot3(a, b):
        return (((a[:, 0] * b[:, 0]).astype(F) + (a[:, 1] * b[:, 1]).astype(F)).astype(F) + (a[:, 2] * b[:, 2]).astype(F)).astype(F)
    two_dot = (F(2.0) * dot3(qv, t)).astype(F)
    k = ((qw * qw).astype(F) - dot3(qv, qv)).astype(F)
    c = np.stack([((qv[:, 1] * t[:, 2]).astype(F) - (t[:, 1] * qv[:, 2]).astype(F)).astype(F),
                  ((qv[:, 2] * t[:, 0]).astype(F) - (t[:, 2] * qv[:, 0]).astype(F)).astype(F),
                  ((qv[:, 0] * t[:, 1]).astype(F) - (t[:, 0] * qv[:, 1]).astype(F)).astype(F)], axis=1)
    tw = (F(2.0) * qw).astype(F)
    return ((((two_dot[:, None] * qv).astype(F) + (k[:, None] * t).astype(F)).astype(F)) + (tw[:, None] * c).astype(F)).astype(F)


def build_codec_fixture():
    rng = np.random.default_rng(0xC0DEC)
    out = {}
    n = 4000
    rgb = (10.0 ** rng.uniform(-7, 5, (n, 1)) * rng.uniform(0.02, 1.0, (n, 3))).astype(F)
    rgb[:50] = 0.0
    rgb[50:100, rng.integers(0, 3)] = 0.0
    out["logluv_rgb"] = rgb
    out["logluv_packed"], out["logluv_borderline"] = logluv_encode(rgb)
    words = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    words[:20] &= np.uint32((1 << 18) - 1)  # Le == 0 -> black
    out["logluv_words"] = words
    out["logluv_decoded"] = logluv_decode(words)
    nrm = rng.normal(size=(n, 3)); nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    axes = np.array([[1, 0, 0], [-1, 0, 0], [0, 1, 0], [0, -1, 0], [0, 0, 1], [0, 0, -1]], float)
    nrm[:60] = np.tile(axes, (10, 1))
    nrm = nrm.astype(F)
    mid = rng.integers(0, 255, n).astype(F)
    out["normal_in"] = nrm
    out["normal_material_id"] = mid
    out["normal_packed"] = nrd_pack_normal(nrm, F(1.0), mid)
    out["normal_unpacked"] = nrd_unpack_normal(out["normal_packed"])
    texels = rng.integers(0, 2 ** 32, n, dtype=np.uint64).astype(np.uint32)
    out["normal_texels"] = texels
    out["normal_texels_unpacked"] = nrd_unpack_normal(texels)
    rad = (10.0 ** rng.uniform(-4, 4.5, (n, 1)) * rng.uniform(0.0, 1.0, (n, 3))).astype(F)
    hd = rng.choice([0.0, 1e-9, 0.5, 8.0, 100000.0], n).astype(F)
    out["radiance_in"] = rad
    out["radiance_hitdist"] = hd
    out["radiance_half4"] = pack_radiance(rad, hd)
    out["radiance_unpacked"] = unpack_radiance(out["radiance_half4"])
    v4 = rng.uniform(-0.2, 1.2, (n, 4)).astype(F)
    v4[:8] = np.array([[0, 0, 0, 0], [1, 1, 1, 1], [0.5, 0.5, 0.5, 0.5], [np.nan, 2, -1, 0.5]] * 2, F)
    out["rgb10a2_in"] = v4
    out["rgb10a2_packed"] = pack_rgb10a2(v4)
    dirs = rng.normal(size=(n, 3)).astype(F)
    dirs[:100] = rng.integers(-2, 3, (100, 3)).astype(F)  # ties between components
    dirs[(dirs == 0).all(axis=1)] = np.array([0, 0, 1], F)
    out["cubed_in"] = dirs
    out["cubed_out"] = cubed_normalize(dirs)
    faces = np.tile(axes, (4, 1)).astype(F)
    out["face_in"] = faces
    out["face_id"] = normal2faceid(faces)
    tv = rng.normal(size=(n, 3)); tv /= np.linalg.norm(tv, axis=1, keepdims=True)
    out["rotate_normal"] = nrm
    out["rotate_target"] = tv.astype(F)
    out["rotate_out"] = rotate_by_normal(nrm, tv.astype(F))
    return out


# --------------------------------------------------------------------------------------------- load_model + from_tree
def linear2srgb_f32(c):
    """geometry.rs:98-105: powf evaluated in double and rounded once (what a correctly rounded f32 powf returns)."""
    c = c.astype(F)
    hi = (F(1.055) * np.power(c.astype(np.float64), np.float64(F(1.0) / F(2.4))).astype(F)).astype(F) - F(0.055)
    return np.where(c <= F(0.0031308), (F(12.92) * c).astype(F), hi.astype(F)).astype(F)


def flatten_model(xyzi, size, palette):
    """xyzi: (n,4) u8 in MagicaVoxel file axes, i = 0-based palette index (dot_vox). Returns blocks (structured), materials (u8),
    borderline (bool per block: an avg_albedo channel within 1e-3 of a truncation step)."""
    x = xyzi[:, 0].astype(np.int64)
    y = xyzi[:, 2].astype(np.int64)                       # loader.rs:248-253: (x, z, size.y - 1 - y)
    z = (int(size[1]) - xyzi[:, 1].astype(np.int64) - 1) & 0xFF
    idx = xyzi[:, 3].astype(np.int64)
    # collector.rs:23-34: block-major grid, later writes win, EVERY set() counts (duplicates double count)
    block_index = (x >> 2) + (y >> 2) * 64 + (z >> 2) * 4096
    bit = (z & 3) | ((y & 3) << 2) | ((x & 3) << 4)
    counts = np.bincount(block_index, minlength=64 ** 3).astype(np.uint32)
    grid = {}
    for b, k, i in zip(block_index.tolist(), bit.tolist(), idx.tolist()):
        grid[b * 64 + k] = (i + 1) & 0xFF  # u8 arithmetic: voxel.i + 1
    running = np.concatenate([[0], np.cumsum(counts[:-1], dtype=np.uint64)]).astype(np.uint32)  # collector.rs:75-82
    cells = np.array(sorted(c for c, v in grid.items() if v != 0), np.int64)
    materials = np.array([grid[c] - 1 for c in cells.tolist()], np.uint8)
    # the tree's leaves: one per occupied 4^3 block; Tree::iter_leaf is depth first with ascending child bits, i.e. ordered by
    # (x>>4, y>>4, z>>4) at the root (index x<<8|y<<4|z) then ((x>>2)&3, (y>>2)&3, (z>>2)&3) (internal.rs:78-81, tree.rs:106-113)
    occ = {}
    for b, k in zip(block_index.tolist(), bit.tolist()):
        occ[b] = occ.get(b, 0) | (1 << k)   # leaf.rs:81-83: bit x<<4 | y<<2 | z
    keys = []
    for b in occ:
        bx, by, bz = b & 63, (b >> 6) & 63, b >> 12
        order = ((bx >> 2) << 8 | (by >> 2) << 4 | (bz >> 2)) << 6 | ((bx & 3) << 4 | (by & 3) << 2 | (bz & 3))
        keys.append((order, b))
    keys.sort()
    dt = np.dtype([("x", "<u2"), ("y", "<u2"), ("z", "<u2"), ("w", "<u2"), ("mask", "<u8"), ("material_ptr", "<u4"), ("avg_albedo", "<u4")])
    blocks = np.zeros(len(keys), dt)
    border = np.zeros(len(keys), bool)
    pal = palette.astype(np.uint32)
    for j, (_, b) in enumerate(keys):
        bx, by, bz = b & 63, (b >> 6) & 63, b >> 12
        mask = occ[b]
        nvox = bin(mask).count("1")
        ptr = int(running[b])
        col = np.zeros(4, np.uint32)
        for i in range(nvox):  # geometry.rs:88-97: the first popcount(mask) entries from material_ptr
            col += pal[materials[ptr + i]] if ptr + i < len(materials) else 0
        c = col.astype(F) / (F(nvox) * F(255.0)).astype(F)
        srgb = linear2srgb_f32(c[:3])
        scaled = np.concatenate([(srgb * F(1023.0)).astype(F), [(c[3] * F(3.0)).astype(F)]])
        r, g, bl, a = [int(v) for v in scaled]  # `as u32`: truncation
        border[j] = bool((np.abs(scaled - np.rint(scaled)) < 1e-3).any())
        blocks[j] = (bx * 4, by * 4, bz * 4, 0, mask, ptr, (r << 22) | (g << 12) | (bl << 2) | a)
    return blocks, materials, border


def build_from_tree_fixture():
    rng = np.random.default_rng(0xF207)
    out = {}
    palette = rng.integers(0, 256, (256, 4), dtype=np.uint8)
    palette[:, 3] = rng.choice([255, 255, 255, 128, 0], 256)
    out["palette"] = palette
    specs = [((5, 7, 3), 40, False), ((16, 16, 16), 900, False), ((40, 33, 21), 6000, False), ((64, 9, 50), 2500, False),
             ((126, 126, 40), 9000, False), ((12, 12, 12), 300, True)]  # the last one repeats XYZI entries (collector.rs:23-34)
    for k, (size, n, dup) in enumerate(specs):
        cells = rng.choice(size[0] * size[1] * size[2], min(n, size[0] * size[1] * size[2]), replace=False)
        xyzi = np.zeros((len(cells), 4), np.uint8)
        xyzi[:, 0] = cells % size[0]
        xyzi[:, 1] = (cells // size[0]) % size[1]
        xyzi[:, 2] = cells // (size[0] * size[1])
        xyzi[:, 3] = rng.integers(0, 255, len(cells))
        if dup:
            # Repeated entries are only well defined in the LAST block of the collector's order (bx + 64 by + 4096 bz in engine
            # axes): every set() bumps the block's count, so a repeat anywhere else shifts all later material_ptr values and
            # the last leaf's reads run past the material buffer (a panic in the reference). In the last block the later entry
            # simply wins the grid cell (collector.rs:33).
            ex, ey, ez = xyzi[:, 0].astype(int), xyzi[:, 2].astype(int), size[1] - 1 - xyzi[:, 1].astype(int)
            bi = (ex >> 2) + (ey >> 2) * 64 + (ez >> 2) * 4096
            last = np.flatnonzero(bi == bi.max())
            again = xyzi[rng.choice(last, 12)].copy()
            again[:, 3] = rng.integers(0, 255, len(again))
            xyzi = np.concatenate([xyzi, again])
        blocks, materials, border = flatten_model(xyzi, size, palette)
        out[f"m{k}_size"] = np.array(size, np.uint32)
        out[f"m{k}_xyzi"] = xyzi
        out[f"m{k}_blocks"] = blocks
        out[f"m{k}_materials"] = materials
        out[f"m{k}_borderline"] = border
    out["n_models"] = np.array([len(specs)], np.int64)
    return out


def main():
    dda_fx = build_dda_fixture()
    np.savez_compressed(os.path.join(HERE, "dda_pairs.npz"), **dda_fx)
    n = len(dda_fx["o"])
    print(f"dda_pairs.npz: {n} rays x 3 shaders = {3 * n} (ray, mask) -> (t, voxel) pairs, dropped {int(dda_fx['dropped'][0])}; "
          f"reported: primary {int(dda_fx['reported0'].sum())}, ao {int(dda_fx['reported1'].sum())} "
          f"({int((dda_fx['hitkind1'] == 1).sum())} threshold), rough {int(dda_fx['reported2'].sum())}")
    cd = build_codec_fixture()
    np.savez_compressed(os.path.join(HERE, "codecs.npz"), **cd)
    print(f"codecs.npz: {len(cd['logluv_rgb'])} vectors per codec, {int(cd['logluv_borderline'].sum())} LogLuv inputs flagged borderline")
    ft = build_from_tree_fixture()
    np.savez_compressed(os.path.join(HERE, "from_tree.npz"), **ft)
    print("from_tree.npz:", ", ".join(f"m{k}: {len(ft[f'm{k}_blocks'])} blocks / {len(ft[f'm{k}_materials'])} materials"
                                      for k in range(int(ft["n_models"][0]))))


if __name__ == "__main__":
    main()
