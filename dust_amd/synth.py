"""Seeded synthetic stand-ins for the reference's assets (test / bench infrastructure).

The reference ships castle.vox, teapot.vox and the STBN noise PNGs only as Git-LFS pointer stubs
(SURVEY F3), so every scene and noise texture used by the tests and by bench.py is generated here
and labelled synthetic in the results. Scenes are written as real MagicaVoxel .vox files so that the
loader path (dust_vox_load) is exercised exactly as VoxLoader::load would be.
"""
import struct

import numpy as np

# ------------------------------------------------------------------ .vox writer
ROT_IDENTITY = 0b0000100
ROT_Z90 = 17     # rows (0,-1,0),(1,0,0),(0,0,1)
ROT_Z180 = 52    # rows (-1,0,0),(0,-1,0),(0,0,1)
ROT_Z270 = 33    # rows (0,1,0),(-1,0,0),(0,0,1)
ROT_MIRROR_X = 20  # rows (-1,0,0),(0,1,0),(0,0,1): det -1


def _chunk(cid: bytes, content: bytes, children: bytes = b"") -> bytes:
    return cid + struct.pack("<II", len(content), len(children)) + content + children


def _string(s: str) -> bytes:
    b = s.encode("ascii")
    return struct.pack("<I", len(b)) + b


def _dict(d: dict) -> bytes:
    out = struct.pack("<I", len(d))
    for k, v in d.items():
        out += _string(k) + _string(v)
    return out


def write_vox(models, instances, palette=None, groups=None, scene_graph=True, anim=None) -> bytes:
    """models: list of (size_xyz, xyzi uint8[n,4] with 1-based colour index as stored in files).
    instances: list of (model_id, (tx,ty,tz), rotation_byte) in file axes.
    groups: optional list of (translation, rotation_byte, [instance indices]) wrapping some instances in an nGRP.
    anim: optional {instance index: {"frames": [(frame, (tx,ty,tz), rotation_byte), ...], "models": [(frame, model_id), ...]}}:
          MagicaVoxel animation -- several keyframes in the instance's nTRN and / or several models in its nSHP ("_f" attributes).
    """
    anim = anim or {}
    body = b""
    for size, xyzi in models:
        xyzi = np.ascontiguousarray(xyzi, np.uint8).reshape(-1, 4)
        body += _chunk(b"SIZE", struct.pack("<III", *[int(v) for v in size]))
        body += _chunk(b"XYZI", struct.pack("<I", xyzi.shape[0]) + xyzi.tobytes())
    if scene_graph:
        nodes = []  # (id, bytes)
        next_id = [2]

        def new_id():
            i = next_id[0]
            next_id[0] += 1
            return i

        def trn(node_id, child, t, r, keys=None):
            frames = b""
            keys = keys or [(None, t, r)]
            for f, kt, kr in keys:
                frame = {}
                if kr != ROT_IDENTITY:
                    frame["_r"] = str(int(kr))
                if tuple(kt) != (0, 0, 0):
                    frame["_t"] = "%d %d %d" % tuple(int(v) for v in kt)
                if f is not None:
                    frame["_f"] = str(int(f))
                frames += _dict(frame)
            return _chunk(b"nTRN", struct.pack("<I", node_id) + _dict({}) + struct.pack("<IiiI", child, -1, 0, len(keys)) + frames)

        def shp(node_id, model, keys=None):
            keys = keys or [(None, model)]
            body_ = b"".join(struct.pack("<I", m) + _dict({} if f is None else {"_f": str(int(f))}) for f, m in keys)
            return _chunk(b"nSHP", struct.pack("<I", node_id) + _dict({}) + struct.pack("<I", len(keys)) + body_)

        def grp(node_id, children):
            return _chunk(b"nGRP", struct.pack("<I", node_id) + _dict({}) + struct.pack("<I", len(children)) +
                          b"".join(struct.pack("<I", c) for c in children))

        grouped = set()
        root_children = []
        out_nodes = b""
        for (gt, gr, members) in (groups or []):
            g_trn, g_grp = new_id(), new_id()
            kids = []
            for idx in members:
                grouped.add(idx)
                mid, t, r = instances[idx]
                a, b = new_id(), new_id()
                out_nodes += trn(a, b, t, r, anim.get(idx, {}).get("frames")) + shp(b, mid, anim.get(idx, {}).get("models"))
                kids.append(a)
            out_nodes += trn(g_trn, g_grp, gt, gr) + grp(g_grp, kids)
            root_children.append(g_trn)
        for idx, (mid, t, r) in enumerate(instances):
            if idx in grouped:
                continue
            a, b = new_id(), new_id()
            out_nodes += trn(a, b, t, r, anim.get(idx, {}).get("frames")) + shp(b, mid, anim.get(idx, {}).get("models"))
            root_children.append(a)
        body += trn(0, 1, (0, 0, 0), ROT_IDENTITY) + grp(1, root_children) + out_nodes
    if palette is not None:
        pal = np.ascontiguousarray(palette, np.uint8).reshape(256, 4)
        body += _chunk(b"RGBA", pal.tobytes())
    return b"VOX " + struct.pack("<I", 150) + _chunk(b"MAIN", b"", body)


def make_palette(seed=7):
    """256 RGBA entries as stored in an RGBA chunk (entry i colours file index i+1)."""
    rng = np.random.default_rng(seed)
    pal = np.zeros((256, 4), np.uint8)
    # stone greys, earth, roof reds, wood, foliage; then random fill
    base = np.array([[150, 150, 148], [120, 118, 115], [92, 88, 84], [180, 176, 170], [110, 84, 60], [86, 60, 40],
                     [150, 52, 40], [176, 70, 52], [70, 110, 50], [96, 140, 64], [200, 190, 150], [60, 60, 70]], np.uint8)
    for i in range(255):
        c = base[i % len(base)].astype(np.int32) + rng.integers(-14, 15, 3)
        pal[i, :3] = np.clip(c, 0, 255)
        pal[i, 3] = 255
    return pal


# ------------------------------------------------------------------ model generators (file axes: z is up)
def _grid_to_xyzi(solid, colour):
    """solid: bool[x,y,z]; colour: uint8[x,y,z] 1-based colour index."""
    x, y, z = np.nonzero(solid)
    out = np.empty((x.size, 4), np.uint8)
    out[:, 0], out[:, 1], out[:, 2] = x, y, z
    out[:, 3] = colour[x, y, z]
    return out


def _colour_field(shape, rng, choices):
    c = rng.integers(0, len(choices), size=shape, dtype=np.uint8)
    return np.asarray(choices, np.uint8)[c]


def model_box(size, rng, colours, shell=0, crenel=0, windows=0):
    sx, sy, sz = size
    solid = np.ones(size, bool)
    if shell:
        solid[shell:sx - shell, shell:sy - shell, shell:sz] = False  # open top shell, floor kept
    if crenel:
        zz = sz - crenel
        xs = (np.arange(sx) // crenel) % 2 == 0
        ys = (np.arange(sy) // crenel) % 2 == 0
        keep = xs[:, None] | ys[None, :]
        solid[:, :, zz:] &= keep[:, :, None]
    for _ in range(windows):
        wx = int(rng.integers(4, max(5, sx - 8)))
        wz = int(rng.integers(8, max(9, sz - 16)))
        solid[wx:wx + 4, : max(shell, 3), wz:wz + 8] = False  # window through the front wall
    return (size, _grid_to_xyzi(solid, _colour_field(size, rng, colours)))


def model_tower(diameter, height, rng, colours, wall=5):
    d = diameter
    xx, yy = np.meshgrid(np.arange(d) - (d - 1) / 2.0, np.arange(d) - (d - 1) / 2.0, indexing="ij")
    r = np.sqrt(xx * xx + yy * yy)
    ring = (r <= d / 2.0) & (r >= d / 2.0 - wall)
    disk = r <= d / 2.0
    solid = np.zeros((d, d, height), bool)
    solid[:, :, :] = ring[:, :, None]
    solid[:, :, :4] = disk[:, :, None]
    solid[:, :, height - 10:height - 6] = disk[:, :, None]
    ang = np.arctan2(yy, xx)
    merlon = ((ang * 8 / np.pi).astype(np.int32) % 2 == 0)
    solid[:, :, height - 6:] &= merlon[:, :, None]
    return ((d, d, height), _grid_to_xyzi(solid, _colour_field(solid.shape, rng, colours)))


def model_house(size, rng, wall_colours, roof_colours):
    sx, sy, sz = size
    wall_h = int(sz * 0.6)
    solid = np.zeros(size, bool)
    solid[:, :, :wall_h] = True
    solid[4:sx - 4, 4:sy - 4, 4:wall_h] = False
    for fz in range(20, wall_h - 4, 18):  # interior floors
        solid[:, :, fz:fz + 3] = True
    colour = _colour_field(size, rng, wall_colours)
    # gable roof along x
    for k in range(sz - wall_h):
        inset = int(k * (sy / 2.0) / max(1, sz - wall_h))
        if inset * 2 >= sy:
            break
        solid[:, inset:sy - inset, wall_h + k] = True
        if inset + 2 < sy - inset - 2 and k + 2 < sz - wall_h:
            solid[2:sx - 2, inset + 2:sy - inset - 2, wall_h + k] = False
        colour[:, :, wall_h + k] = _colour_field((sx, sy), rng, roof_colours)
    # door and windows
    solid[sx // 2 - 3:sx // 2 + 3, 0:3, 3:14] = False
    for _ in range(4):
        wx = int(rng.integers(5, max(6, sx - 10)))
        wz = int(rng.integers(8, max(9, wall_h - 8)))
        solid[wx:wx + 4, :3, wz:wz + 5] = False
        solid[wx:wx + 4, sy - 3:, wz:wz + 5] = False
    return (size, _grid_to_xyzi(solid, colour))


def model_teapot(n=96, seed=0x7EA):
    """teapot.vox stand-in: ellipsoid shell + spout + handle + lid knob, one <=128^3 model."""
    rng = np.random.default_rng(seed)
    g = np.arange(n) - (n - 1) / 2.0
    x, y, z = np.meshgrid(g, g, g, indexing="ij")
    body = (x / (0.34 * n)) ** 2 + (y / (0.34 * n)) ** 2 + ((z + 0.08 * n) / (0.27 * n)) ** 2
    shell = (body <= 1.0) & (body >= 0.80)
    # spout: tilted cylinder towards +x
    t = (x - 0.25 * n) * 0.8 + (z - 0.02 * n) * 0.6
    px, pz = x - (0.25 * n + 0.8 * t), z - (0.02 * n + 0.6 * t)
    spout = (px * px + y * y + pz * pz <= (0.05 * n) ** 2) & (t >= -0.05 * n) & (t <= 0.22 * n)
    # handle: torus section on -x
    rr = np.sqrt((x + 0.36 * n) ** 2 + (z + 0.02 * n) ** 2)
    handle = ((rr - 0.13 * n) ** 2 + y * y <= (0.035 * n) ** 2) & (x < -0.30 * n)
    knob = (x * x + y * y + (z - 0.23 * n) ** 2) <= (0.05 * n) ** 2
    solid = shell | spout | handle | knob
    colour = _colour_field(solid.shape, rng, [11, 12, 4, 1])
    return ((n, n, n), _grid_to_xyzi(solid, colour))


def teapot_scene(n=96, seed=0x7EA):
    """.vox bytes for the teapot stand-in (single model, with scene graph)."""
    return write_vox([model_teapot(n, seed)], [(0, (0, 0, 0), ROT_IDENTITY)], make_palette(seed & 0xFF))


def castle_scene(seed=0xD057, scale=1.0):
    """castle.vox stand-in: ~100 models <= 256^3, ~150 instances incl. 90-degree rotations and a
    mirrored one, ground slabs, crenellated walls, towers, a keep, stairs and houses.
    scale < 1 shrinks every model (for tests). Returns (.vox bytes, info dict)."""
    rng = np.random.default_rng(seed)

    def s(v, lo=4):
        return max(lo, int(round(v * scale)))

    stone, dark, earth, roof, wood = [1, 2, 4], [2, 3, 12], [5, 6, 9], [7, 8], [5, 6]
    models, instances = [], []

    def add_model(m):
        models.append(m)
        return len(models) - 1

    def place(mid, x, y, z, r=ROT_IDENTITY):
        instances.append((mid, (int(round(x * scale)), int(round(y * scale)), int(round(z * scale))), r))

    # ground: two slab variants, 5x5 grid of 256x256x24, z centred at -12 so the top is at z = 0
    g = s(256)
    slabs = [add_model(model_box((g, g, s(24)), rng, earth + [9, 10])), add_model(model_box((g, g, s(24)), rng, earth + stone))]
    for i in range(-2, 3):
        for j in range(-2, 3):
            place(slabs[(i + j) & 1], i * 256, j * 256, -12)
    # curtain walls: 3 variants, 256 x 24 x 96, crenellated, ring at +-384
    walls = [add_model(model_box((s(256), s(24), s(96)), rng, stone if k else stone + dark, crenel=s(6))) for k in range(3)]
    for k, off in enumerate((-256, 0, 256)):
        place(walls[k % 3], off, 384, 48)
        place(walls[(k + 1) % 3], off, -384, 48, ROT_Z180)
        place(walls[(k + 2) % 3], 384, off, 48, ROT_Z90)
        place(walls[k % 3], -384, off, 48, ROT_Z270)
    # towers: 4 variants at the corners and mid-walls
    towers = [add_model(model_tower(s(64 + 8 * k), s(200 - 10 * k), rng, stone + dark, wall=s(5, 2))) for k in range(4)]
    for k, (x, y) in enumerate(((384, 384), (-384, 384), (384, -384), (-384, -384), (128, 384), (-128, -384))):
        place(towers[k % 4], x, y, (200 - 10 * (k % 4)) / 2.0)
    # keep: hollow box 140 x 140 x 110 with crenellations and windows, a smaller turret on top
    keep = add_model(model_box((s(140), s(140), s(110)), rng, stone, shell=s(6, 2), crenel=s(8), windows=12))
    place(keep, 0, 0, 55)
    keep_top = add_model(model_box((s(72), s(72), s(40)), rng, stone + roof, shell=s(5, 2), crenel=s(6)))
    place(keep_top, 0, 0, 130, ROT_Z90)
    # stairs up to the keep
    st = s(96)
    stairs = np.zeros((st, s(48), s(64)), bool)
    for i in range(st):
        stairs[i, :, : max(1, int((i + 1) * s(64) / st))] = True
    stairs_m = add_model(((st, s(48), s(64)), _grid_to_xyzi(stairs, _colour_field(stairs.shape, rng, stone))))
    place(stairs_m, -118, 0, 32)
    place(stairs_m, 118, 0, 32, ROT_MIRROR_X)
    # houses: many seeded variants scattered in the bailey and outside the walls
    n_house_models = 85
    houses = []
    for k in range(n_house_models):
        sz = (s(int(rng.integers(56, 104))), s(int(rng.integers(48, 88))), s(int(rng.integers(48, 96))))
        houses.append((add_model(model_house(sz, rng, stone + wood, roof)), sz))
    rots = (ROT_IDENTITY, ROT_Z90, ROT_Z180, ROT_Z270)
    placed = 0
    grid = [(x, y) for x in range(-600, 601, 100) for y in range(-600, 601, 100)
            if not (abs(x) < 170 and abs(y) < 110) and not (330 < max(abs(x), abs(y)) < 440)]
    order = rng.permutation(len(grid))
    group_members = []
    for gi in order[:110]:
        x, y = grid[gi]
        mid, sz = houses[placed % n_house_models]
        place(mid, x + int(rng.integers(-10, 11)), y + int(rng.integers(-10, 11)), sz[2] / (2.0 * scale), rots[int(rng.integers(0, 4))])
        if placed % 10 == 0:
            group_members.append(len(instances) - 1)
        placed += 1
    groups = [((int(8 * scale), int(-8 * scale), 0), ROT_Z90, group_members)]
    data = write_vox(models, instances, make_palette(seed & 0xFF), groups=groups)
    info = {"n_models": len(models), "n_instances": len(instances),
            "n_voxels": int(sum(m[1].shape[0] for m in models)), "seed": seed, "scale": scale}
    return data, info


# ------------------------------------------------------------------ STBN stand-ins
def stbn_scalar(seed=0x57B0, layers=64):
    """texture [0]: 128x128xlayers R8. Plain PCG white noise, NOT spatiotemporal blue noise."""
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(layers, 128, 128), dtype=np.uint8)


def stbn_unitvec3_cosine(seed=0x57B5, layers=64):
    """texture [5]: 128x128xlayers RGBA8, cosine-weighted hemisphere about +z stored as v*0.5+0.5."""
    rng = np.random.default_rng(seed)
    u = rng.random((layers, 128, 128))
    v = rng.random((layers, 128, 128))
    r = np.sqrt(u)
    phi = 2.0 * np.pi * v
    vec = np.stack([r * np.cos(phi), r * np.sin(phi), np.sqrt(1.0 - u)], axis=-1)
    out = np.empty((layers, 128, 128, 4), np.uint8)
    out[..., :3] = np.clip(np.rint((vec * 0.5 + 0.5) * 255.0), 0, 255).astype(np.uint8)
    out[..., 3] = 255
    return out


def splitmix32(x):
    x = (x + 0x9E3779B9) & 0xFFFFFFFF
    z = x
    z = ((z ^ (z >> 16)) * 0x85EBCA6B) & 0xFFFFFFFF
    z = ((z ^ (z >> 13)) * 0xC2B2AE35) & 0xFFFFFFFF
    return (z ^ (z >> 16)) & 0xFFFFFFFF


def frame_rand(seed, frame_index):
    """`rand` push constant of a frame (standard.rs:449 uses thread_rng; SURVEY 8d fixes it to this)."""
    return splitmix32((seed ^ frame_index) & 0xFFFFFFFF)


# ------------------------------------------------------------------ config 5: procedural deep tree
def _mix32(x):
    x = (x ^ (x >> 16)) * np.uint32(0x7FEB352D)
    x = (x ^ (x >> 15)) * np.uint32(0x846CA68B)
    return x ^ (x >> 16)


def procedural_deep_blocks(seed=0xC5, occupancy=0.01, extent_log2=12, chunk=1 << 22, sample=False):
    """Blocks of a single 4096^3 model (hierarchy (4,4,2,2)): brick (bx,by,bz) is occupied iff
    hash(seed,bx,by,bz) < occupancy; its 64 voxels are set with p = 0.5. Returned in Tree::iter_leaf order
    (depth-first, x slowest at every level) as (blocks, materials) ready for dust_hip_model_create.
    sample=True draws occupancy * bricks random brick coordinates instead of hashing all 2^30 (for tests)."""
    from .api import BLOCK_DTYPE
    nb = 1 << (extent_log2 - 2)            # bricks per axis
    thr = np.uint32(min(0xFFFFFFFF, int(occupancy * 4294967296.0)))
    total = nb ** 3
    keep = []
    with np.errstate(over="ignore"):
        if sample:
            rng0 = np.random.default_rng(seed)
            c = np.unique(rng0.integers(0, total, int(round(occupancy * total)), dtype=np.uint64))
            bx, by, bz = (c // (nb * nb)).astype(np.uint32), ((c // nb) % nb).astype(np.uint32), (c % nb).astype(np.uint32)
            h = _mix32(_mix32(_mix32(bx + np.uint32(seed)) ^ by * np.uint32(0x9E3779B1)) ^ bz * np.uint32(0x85EBCA77))
            keep.append(np.stack([bx, by, bz, h], axis=1))
        for start in range(0, 0 if sample else total, chunk):
            idx = np.arange(start, min(total, start + chunk), dtype=np.uint64)
            bx = (idx // (nb * nb)).astype(np.uint32)
            by = ((idx // nb) % nb).astype(np.uint32)
            bz = (idx % nb).astype(np.uint32)
            h = _mix32(_mix32(_mix32(bx + np.uint32(seed)) ^ by * np.uint32(0x9E3779B1)) ^ bz * np.uint32(0x85EBCA77))
            sel = h < thr
            if sel.any():
                keep.append(np.stack([bx[sel], by[sel], bz[sel], h[sel]], axis=1))
        k = np.concatenate(keep) if keep else np.zeros((0, 4), np.uint32)
        # depth-first key: per level (256-cell, 16-cell, 4-cell) the child index x<<2f | y<<f | z
        bx, by, bz = (k[:, i].astype(np.uint64) for i in range(3))

        def lv(v, shift, bits):
            return (v >> np.uint64(shift)) & np.uint64((1 << bits) - 1)
        key = np.zeros(len(k), np.uint64)
        for shift, bits in ((6, 4), (2, 4), (0, 2)) if extent_log2 == 12 else ((2, 4), (0, 2)):
            key = (key << np.uint64(3 * bits)) | (lv(bx, shift, bits) << np.uint64(2 * bits)) | (lv(by, shift, bits) << np.uint64(bits)) | lv(bz, shift, bits)
        order = np.argsort(key, kind="stable")
        k = k[order]
        hh = k[:, 3]
        lo = _mix32(hh ^ np.uint32(0xA5A5A5A5))
        hi = _mix32(hh ^ np.uint32(0x5A5A5A5A))
        mask = (hi.astype(np.uint64) << np.uint64(32)) | lo.astype(np.uint64)
        mask[mask == 0] = 1
    blocks = np.zeros(len(k), BLOCK_DTYPE)
    blocks["x"], blocks["y"], blocks["z"] = k[:, 0] * 4, k[:, 1] * 4, k[:, 2] * 4
    blocks["mask"] = mask
    counts = np.array([bin(int(m)).count("1") for m in mask], np.uint32) if len(mask) < 200000 else \
        np.unpackbits(mask.view(np.uint8).reshape(-1, 8), axis=1).sum(axis=1).astype(np.uint32)
    blocks["material_ptr"] = np.concatenate([[0], np.cumsum(counts)[:-1]]).astype(np.uint32) if len(k) else 0
    rng = np.random.default_rng(seed)
    materials = rng.integers(0, 255, int(counts.sum()), dtype=np.uint8)
    blocks["avg_albedo"] = (hh & np.uint32(0xFFFFFFFC)) | np.uint32(3)
    return blocks, materials


def write_apng(frames, filters=(0, 1, 2, 3, 4), interlace=0, frame_rect=None):
    """Encode frames (layers, h, w[, channels]; uint8 or uint16) as a PNG (1 layer) or an APNG whose first frame is the
    default image -- the layout of the reference's stbn/*.png textures. Scanline filter types cycle through `filters`.
    frame_rect=(w, h, x, y) writes that (wrong on purpose) rectangle into the later frames' fcTL for the error tests."""
    import struct
    import zlib
    a = np.asarray(frames)
    if a.ndim == 3:
        a = a[..., None]
    layers, h, w, ch = a.shape
    depth = 8 if a.dtype == np.uint8 else 16
    color_type = {1: 0, 2: 4, 3: 2, 4: 6}[ch]
    bpp = ch * depth // 8

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d) & 0xFFFFFFFF)

    def encode(img):
        rows = (img.astype(">u2") if depth == 16 else img).reshape(h, -1).view(np.uint8).astype(np.int32)
        out = bytearray()
        prev = np.zeros(rows.shape[1], np.int32)
        for y in range(h):
            f = filters[y % len(filters)]
            cur = rows[y]
            left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]])
            ul = np.concatenate([np.zeros(bpp, np.int32), prev[:-bpp]])
            if f == 0:
                pred = np.zeros_like(cur)
            elif f == 1:
                pred = left
            elif f == 2:
                pred = prev
            elif f == 3:
                pred = (left + prev) >> 1
            else:
                p = left + prev - ul
                pa, pb, pc = np.abs(p - left), np.abs(p - prev), np.abs(p - ul)
                pred = np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
            out.append(f)
            out += ((cur - pred) & 255).astype(np.uint8).tobytes()
            prev = cur
        return zlib.compress(bytes(out), 6)

    png = b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, depth, color_type, 0, 0, interlace))
    seq = 0
    if layers > 1:
        png += chunk(b"acTL", struct.pack(">II", layers, 0))
    for i in range(layers):
        if layers > 1:
            fw, fh, fx, fy = (w, h, 0, 0) if (frame_rect is None or i == 0) else frame_rect
            png += chunk(b"fcTL", struct.pack(">IIIIIHHBB", seq, fw, fh, fx, fy, 1, 30, 0, 0))
            seq += 1
        z = encode(a[i])
        if i == 0:
            half = len(z) // 2   # two IDAT chunks: the stream may be split anywhere
            png += chunk(b"IDAT", z[:half]) + chunk(b"IDAT", z[half:])
        else:
            png += chunk(b"fdAT", struct.pack(">I", seq) + z)
            seq += 1
    return png + chunk(b"IEND", b"")

