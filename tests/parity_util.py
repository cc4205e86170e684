"""Shared helpers for the parity tests, __graft_entry__.smoke() and bench.py's cpu_baseline leg:
build the same scene for the HIP library (through the C ABI) and for the CPU oracle, render, compare."""
import ctypes as C
import json
import os

import numpy as np

import oracle_lib as O
from dust_amd import _lib as L
from dust_amd import api, synth

from dust_amd.scenes import SceneDesc, camera_for, hip_scene  # noqa: F401  (oracle-free, shared with bench.py)
from dust_amd import scenes as _scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sky_state(name="default"):
    """The packaged default sun, or one of the named test suns of tests/golden/sky_states.json (make_sky_fixtures.py)."""
    if name == "default":
        return _scenes.sky_state()
    with open(os.path.join(ROOT, "tests", "golden", "sky_states.json")) as f:
        return np.asarray(json.load(f)[name]["state"], np.float32)


def random_model(rng, size=(48, 40, 56), fill=0.08, blobs=6):
    """A sparse random model plus a few solid blobs, returned as xyzi (0-based colour index)."""
    sx, sy, sz = size
    solid = rng.random(size) < fill
    for _ in range(blobs):
        c = [int(rng.integers(0, s)) for s in size]
        r = int(rng.integers(3, 9))
        x, y, z = np.ogrid[:sx, :sy, :sz]
        solid |= (x - c[0]) ** 2 + (y - c[1]) ** 2 + (z - c[2]) ** 2 <= r * r
    x, y, z = np.nonzero(solid)
    xyzi = np.stack([x, y, z, rng.integers(0, 255, x.size)], axis=1).astype(np.uint8)
    return xyzi


def small_scene(seed=1, n_models=3, n_instances=5, size=(48, 40, 56)):
    """Random models flattened by the product loader, instances with axis rotations/mirrors and offsets."""
    rng = np.random.default_rng(seed)
    pal = synth.make_palette(seed)
    models = []
    for _ in range(n_models):
        sz = tuple(int(v) for v in (rng.integers(size[0] // 2, size[0] + 1), rng.integers(size[1] // 2, size[1] + 1),
                                    rng.integers(size[2] // 2, size[2] + 1)))
        xyzi = random_model(rng, sz)
        models.append(api.flatten_model(xyzi, sz, pal))
    rots = [np.eye(3), np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]]), np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]]),
            np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]]), np.array([[-1, 0, 0], [0, 1, 0], [0, 0, 1]])]
    instances = []
    for i in range(n_instances):
        m = np.zeros((3, 4), np.float32)
        m[:, :3] = rots[i % len(rots)]
        m[:, 3] = rng.integers(-60, 60, 3) + (0.5 if i % 2 else 0.0)
        instances.append((int(rng.integers(0, n_models)), m.reshape(12)))
    return SceneDesc(models, pal, instances)


def clustered_deep_model(seed=7, n_cells=20000, cell_lo=96, cell_hi=160, max_bricks=12):
    """A 4096^3 model (hierarchy (4,4,2,2)) whose occupied 16-cells -- n_cells of them, drawn inside [cell_lo, cell_hi)^3 in 16-cell
    units -- hold 1 to max_bricks bricks each: the DEEP kernels test the bricks of a cell with up to four one by one and walk the
    others 4-cell by 4-cell, and a ray meets both kinds. Returns (blocks, materials, palette) in Tree::iter_leaf order."""
    rng = np.random.default_rng(seed)
    cells = np.unique(rng.integers(cell_lo, cell_hi, (n_cells, 3)), axis=0)
    per = rng.integers(1, max_bricks + 1, len(cells))
    bricks = []
    for c, n in zip(cells, per):
        sub = rng.choice(64, int(n), replace=False)
        bricks.append(np.stack([c[0] * 4 + (sub >> 4), c[1] * 4 + ((sub >> 2) & 3), c[2] * 4 + (sub & 3)], axis=1))
    b = np.concatenate(bricks).astype(np.uint64)
    key = np.zeros(len(b), np.uint64)
    for shift, bits in ((6, 4), (2, 4), (0, 2)):          # depth first, x slowest at every level
        lv = [(b[:, a] >> np.uint64(shift)) & np.uint64((1 << bits) - 1) for a in range(3)]
        key = (key << np.uint64(3 * bits)) | (lv[0] << np.uint64(2 * bits)) | (lv[1] << np.uint64(bits)) | lv[2]
    b = b[np.argsort(key, kind="stable")]
    mask = rng.integers(1, 1 << 63, len(b), dtype=np.uint64) | (rng.integers(0, 2, len(b), dtype=np.uint64) << np.uint64(63))
    sparse = rng.random(len(b)) < 0.25                    # a quarter of the bricks nearly empty: rays pass through them
    mask[sparse] &= rng.integers(1, 1 << 63, int(sparse.sum()), dtype=np.uint64) & rng.integers(1, 1 << 63, int(sparse.sum()), dtype=np.uint64)
    mask[mask == 0] = 1
    blocks = np.zeros(len(b), api.BLOCK_DTYPE)
    blocks["x"], blocks["y"], blocks["z"], blocks["mask"] = b[:, 0] * 4, b[:, 1] * 4, b[:, 2] * 4, mask
    counts = np.array([bin(int(m)).count("1") for m in mask], np.uint32)
    blocks["material_ptr"] = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32)
    mats = rng.integers(0, 255, int(counts.sum()), dtype=np.uint8)
    return blocks, mats, synth.make_palette(5)


def oracle_scene(desc: SceneDesc):
    s = O.Scene()
    for b, m in desc.models:
        s.add_model(b, m, desc.palette)
    for mid, t in desc.instances:
        s.add_instance(mid, t)
    s.commit()
    return s


def render_oracle(oscene, cam, sky, w, h, passes, noise5=None, rand=0, mode=O.ORC_MODE_HIER, rows=None, stats=None,
                  noise0=None, gi=None, frame_index=1, g=None, gi_threads=0):
    """gi_threads > 0: the GI passes on that many host threads (orc_pass_final_gather_mt / orc_pass_surfel_mt: the serial passes' result)"""
    g = g or O.GBuffer(w, h)
    oc, osky = O.camera_from(cam), O.sky_from(sky)
    y0, y1 = rows if rows else (0, h)
    st = stats if stats is not None else [O.OrcRayStats() for _ in range(6)]
    l = O.lib()
    if passes & L.PASS_PRIMARY:
        l.orc_pass_primary(oscene.h, mode, C.byref(oc), C.byref(osky), C.byref(g.c), y0, y1, C.byref(st[0]))
    if passes & L.PASS_AMBIENT_OCCLUSION:
        n5 = np.ascontiguousarray(noise5, np.uint8)
        l.orc_pass_ao(oscene.h, mode, C.byref(oc), C.byref(osky), C.byref(g.c), n5.ctypes.data_as(C.c_void_p), rand, y0, y1,
                      C.byref(st[1]), C.byref(st[2]))
    if passes & (L.PASS_FINAL_GATHER | L.PASS_SURFEL):
        n0 = np.ascontiguousarray(noise0, np.uint8)
        n5 = np.ascontiguousarray(noise5, np.uint8)
    if passes & L.PASS_FINAL_GATHER and gi_threads:
        l.orc_pass_final_gather_mt(oscene.h, mode, C.byref(oc), C.byref(osky), C.byref(g.c), n0.ctypes.data_as(C.c_void_p),
                                   n5.ctypes.data_as(C.c_void_p), rand, frame_index, gi.h, y0, y1, gi_threads, C.byref(st[3]))
    elif passes & L.PASS_FINAL_GATHER:
        l.orc_pass_final_gather(oscene.h, mode, C.byref(oc), C.byref(osky), C.byref(g.c), n0.ctypes.data_as(C.c_void_p),
                                n5.ctypes.data_as(C.c_void_p), rand, frame_index, gi.h, y0, y1, C.byref(st[3]))
    if passes & L.PASS_SURFEL and gi_threads:
        l.orc_pass_surfel_mt(oscene.h, mode, C.byref(osky), n0.ctypes.data_as(C.c_void_p), n5.ctypes.data_as(C.c_void_p), rand,
                             frame_index, gi.h, gi_threads, C.byref(st[4]), C.byref(st[5]))
    elif passes & L.PASS_SURFEL:
        l.orc_pass_surfel(oscene.h, mode, C.byref(osky), n0.ctypes.data_as(C.c_void_p), n5.ctypes.data_as(C.c_void_p), rand,
                          frame_index, gi.h, C.byref(st[4]), C.byref(st[5]))
    return g


PLANES = [("illuminance", L.PLANE_ILLUMINANCE), ("denoised", L.PLANE_DENOISED), ("albedo", L.PLANE_ALBEDO),
          ("normal", L.PLANE_NORMAL), ("depth", L.PLANE_DEPTH), ("motion", L.PLANE_MOTION), ("voxel_id", L.PLANE_VOXEL_ID)]


def read_hip_gbuffer(pipe):
    return {name: pipe.read_plane(pid) for name, pid in PLANES}


def half_to_float(a):
    return a.view(np.float16).astype(np.float32)


def compare_gbuffers(g, hip, check_ao=False):
    """Returns a dict of mismatch counts. Integer/packed planes and depth must be bit-exact where the
    reference defines them (hit pixels: all planes; miss pixels: denoised, albedo, depth, motion)."""
    depth = g.depth
    hit = np.isfinite(depth)
    res = {}
    res["depth"] = int(np.count_nonzero(depth.view(np.uint32) != hip["depth"].view(np.uint32)))
    res["albedo"] = int(np.count_nonzero(g.albedo != hip["albedo"]))
    res["motion"] = int(np.count_nonzero((g.motion != hip["motion"]).any(axis=-1)))
    res["normal"] = int(np.count_nonzero((g.normal != hip["normal"]) & hit))
    res["voxel_id"] = int(np.count_nonzero((g.voxel_id != hip["voxel_id"]) & hit))
    # radiance planes go through exp/pow/acos: compare as floats
    den_o, den_h = half_to_float(g.denoised), half_to_float(hip["denoised"])
    miss = ~hit
    def rel_l2(a, b):
        """relative L2 over the finite samples; where fp16 overflowed (looking into the sun) both sides must hold the
        same non-finite half"""
        fin = np.isfinite(a) & np.isfinite(b)
        inf_a, inf_b = np.isinf(a), np.isinf(b)
        if (not np.array_equal(np.isnan(a), np.isnan(b)) or not np.array_equal(inf_a, inf_b)
                or not np.array_equal(np.sign(a[inf_a]), np.sign(b[inf_b]))):   # NaNs (acos of 1 + ulp in the sky model) carry no sign
            return float("inf")
        a, b = a[fin].astype(np.float64), b[fin].astype(np.float64)
        return float(np.sqrt(((a - b) ** 2).sum()) / max(1e-30, np.sqrt((a ** 2).sum())))

    if miss.any():
        a, b = den_o[miss][:, :3], den_h[miss][:, :3]
        res["denoised_rel_l2"] = rel_l2(a, b)
        res["denoised_hitdist"] = int(np.count_nonzero(g.denoised[miss][:, 3] != hip["denoised"][miss][:, 3]))
    ill_o, ill_h = half_to_float(g.illuminance), half_to_float(hip["illuminance"])
    if hit.any():
        a, b = ill_o[hit][:, :3], ill_h[hit][:, :3]
        res["illuminance_rel_l2"] = rel_l2(a, b)
        res["illuminance_hitdist"] = int(np.count_nonzero(g.illuminance[hit][:, 3] != hip["illuminance"][hit][:, 3]))
    return res


def assert_parity(res):
    for k in ("depth", "albedo", "motion", "normal", "voxel_id"):
        assert res[k] == 0, f"{k}: {res[k]} pixels differ ({res})"
    for k in ("denoised_hitdist", "illuminance_hitdist"):
        assert res.get(k, 0) == 0, f"{k}: {res[k]} pixels differ ({res})"
    for k in ("denoised_rel_l2", "illuminance_rel_l2"):
        assert res.get(k, 0.0) <= 1e-3, f"{k} = {res[k]} exceeds 1e-3 ({res})"  # north_star tolerance


def sharded_gi_vs_single_device(ctx, scene, cam, sky, w, h, world, frames, n0, n5, seed=3, gi_sizes=None, shard_trace=False):
    """SURVEY 8e option i with the collectives done by hand: `world` pipelines on one GPU play the ranks (row bands for the
    pixel passes, the exchange of dust_hip_pipeline_gi_exchange, replicated ordered surfel pass); asserts that every rank's
    spatial hash, surfel pool and own illuminance band equal the single-pipeline run bit for bit. Returns the reference hash.
    shard_trace: the surfel TRACE sharded over the ranks as well (round 6): rank r traces its share of the ordered pool, a loopback group's
    dust_hip_gi_surfel_exchange_run gathers the records, repeats the stamps and applies."""
    from dust_amd import sharding
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]

    def d2h(ptr, n):
        out = np.empty(n, np.int32)
        assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2) == 0
        return out

    def h2d(ptr, arr):
        arr = np.ascontiguousarray(arr, np.int32)
        assert hip.hipMemcpy(C.c_void_p(ptr), arr.ctypes.data_as(C.c_void_p), arr.nbytes, 1) == 0

    def make():
        p = api.StandardPipeline(ctx, w, h)
        p.set_noise(0, n0)
        p.set_noise(5, n5)
        if gi_sizes:
            p.configure_gi(*gi_sizes)
        return p

    ref, ranks = make(), [make() for _ in range(world)]
    per = sharding.gi_band_rows(world, h)
    bands = [(min(h, r * per), min(h, (r + 1) * per)) for r in range(world)]
    exs = [p.gi_exchange(world * per) for p in ranks]
    pix = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER
    for frame in range(1, frames + 1):
        rnd = synth.frame_rand(seed, frame)
        ref.render(scene, cam, sky, pix | L.PASS_SURFEL | L.PASS_GI_ORDERED, frame, rnd)
        for r, p in enumerate(ranks):
            if bands[r][0] < bands[r][1]:
                p.render(scene, cam, sky, pix | L.PASS_GI_SHARDED, frame, rnd, rows=bands[r])
        ctx.sync()
        owner = np.max([d2h(e.slot_owner, e.pool_size) for e in exs], axis=0)                 # all-reduce MAX
        touched = np.zeros(world * per * w, np.int32)
        for r, e in enumerate(exs):                                                           # all-gather of the bands
            touched[r * per * w:(r + 1) * per * w] = d2h(e.touched, world * per * w)[r * per * w:(r + 1) * per * w]
        for e in exs:
            h2d(e.slot_owner, owner)
            h2d(e.touched, touched)
        for r, p in enumerate(ranks):
            p.gi_export(*bands[r]) if bands[r][0] < bands[r][1] else p.gi_export(h, h)   # (a rank without rows: zeroes into the merge)
        ctx.sync()
        merged = np.sum([d2h(e.merged, e.pool_size * 4) for e in exs], axis=0, dtype=np.int64).astype(np.int32)  # all-reduce SUM
        for r, (p, e) in enumerate(zip(ranks, exs)):
            h2d(e.merged, merged)
            if bands[r][0] < bands[r][1]:
                p.gi_import(bands[r][0], bands[r][1], frame)
            else:
                p.gi_import(h, h, frame)   # a rank past the end of the frame: every stamp is another band's (as bench.py does)
            p.render(scene, cam, sky, L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_GI_SHARDED, frame, rnd, surfel_shard=(r, world) if shard_trace else (0, 0))
        if shard_trace:
            if frame == 1:
                comms = api.Comm.local(ctx, world)
            for r, p in enumerate(ranks):
                comms[r].gi_surfel_exchange(p, frame)
        ctx.sync()
    h_ref, s_ref = ref.read_gi()
    ill_ref = ref.read_plane(L.PLANE_ILLUMINANCE)
    for r, p in enumerate(ranks):
        hh, sp = p.read_gi()
        assert np.array_equal(hh, h_ref), f"rank {r}: hash differs"
        assert np.array_equal(sp.view(np.uint32), s_ref.view(np.uint32)), f"rank {r}: surfel pool differs"
        ill = p.read_plane(L.PLANE_ILLUMINANCE)
        assert np.array_equal(ill[bands[r][0]:bands[r][1]], ill_ref[bands[r][0]:bands[r][1]]), f"rank {r}: band differs"
    return h_ref


def run_smoke():
    """__graft_entry__.smoke(): one 96x64 primary + AO frame on device 0 against the oracle."""
    desc = small_scene(seed=3)
    ctx = api.Context(device=0)
    scene = hip_scene(ctx, desc)
    w, h = 96, 64
    pipe = api.StandardPipeline(ctx, w, h)
    noise5 = synth.stbn_unitvec3_cosine(layers=2)
    pipe.set_noise(5, noise5)
    cam = camera_for((90.0, 70.0, 110.0))
    sky = sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    pipe.render(scene, cam, sky, passes, frame_index=1, rand=12345)
    ctx.sync()
    hip = read_hip_gbuffer(pipe)
    g = render_oracle(oracle_scene(desc), cam, sky, w, h, passes, noise5[1 % 2], 12345)
    res = compare_gbuffers(g, hip)
    assert_parity(res)
    # ... and the same frame as the second of three in ONE persistent launch (dust_hip_render_frames / k_primary_ao_batch: bench.py's headline path)
    more = []
    for _ in range(3):
        p2 = api.StandardPipeline(ctx, w, h)
        p2.set_noise(5, noise5)
        more.append(p2)
    api.StandardPipeline.render_frames(more, scene, cam, sky, passes, [2, 1, 3], [7, 12345, 9])
    ctx.sync()
    res2 = compare_gbuffers(g, read_hip_gbuffer(more[1]))
    assert_parity(res2)
    print("smoke ok:", res, "hit pixels:", int(np.isfinite(g.depth).sum()), "| second of three frames in one launch:", res2)
