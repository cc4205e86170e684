"""The top-level structures dust_hip_scene_commit builds over the instances' world boxes (what the reference hands the driver as a
TLAS, accel_struct/tlas.rs:37-117), through the host-only entry point dust_hip_top_level_build: no device needed.
 * the grid lists every instance in every cell its box overlaps -- and, for the per-ray walk of gi.hip, in every cell a ray can be
   in while inside the box: random rays are marched through the grid by a numpy restatement of the walk's rules (one axis per
   step, exit planes from integer coordinates, "listed in the previous cell = dealt with") and must come across exactly the
   instances whose box they meet;
 * the cell blocks the walk's skip rule reads are the lists' own;
 * the slot order is a permutation, and consecutive slots are neighbours (the packet cull's groups of 64 stay small)."""
import numpy as np
import pytest

from dust_amd import api


def boxes_of(n, seed, span=(600.0, 90.0, 600.0), size=(4.0, 40.0)):
    rng = np.random.default_rng(seed)
    c = rng.uniform(-0.5, 0.5, (n, 3)) * np.array(span)
    h = rng.uniform(size[0], size[1], (n, 3)) * 0.5
    return np.concatenate([c - h, c + h], axis=1).astype(np.float32)


@pytest.mark.parametrize("n,seed", [(1, 1), (7, 2), (157, 3), (1000, 4), (4157, 5)])
def test_grid_lists_every_instance_where_its_box_is(n, seed):
    b = boxes_of(n, seed)
    t = api.top_level_build(b)
    dim, lo, cell = np.array(t["dim"]), t["lo"].astype(np.float64), t["cell"].astype(np.float64)
    assert (dim >= 1).all() and (dim <= 256).all() and len(t["cells"]) == dim.prod()
    first, count = t["cells"] & ((1 << 20) - 1), t["cells"] >> 20
    assert count.sum() == len(t["items"]) and (first[1:] == (first + count)[:-1]).all()      # lists stand one behind the other
    # the grid's box holds every instance box
    assert (b[:, :3] >= lo - 1e-3).all() and (b[:, 3:] <= lo + dim * cell + 1e-3).all()
    listed = [set() for _ in range(n)]
    for c in np.nonzero(count)[0]:
        ids = t["items"][first[c]:first[c] + count[c]]
        assert (np.diff(ids.astype(np.int64)) > 0).all()                                      # ascending, no duplicates
        for i in ids:
            listed[i].add(int(c))
    for i in range(n):
        rl, rh = int(t["ranges"][i, 0]), int(t["ranges"][i, 1])
        blo = np.array([rl & 255, (rl >> 9) & 255, rl >> 18]); bhi = np.array([rh & 255, (rh >> 9) & 255, rh >> 18])
        # the block the skip rule reads == the cells the instance is listed in
        want = {int((z * dim[1] + y) * dim[0] + x) for z in range(blo[2], bhi[2] + 1) for y in range(blo[1], bhi[1] + 1) for x in range(blo[0], bhi[0] + 1)}
        assert listed[i] == want
        # ... and covers every cell the box overlaps
        clo = np.clip(np.floor((b[i, :3] - lo) / cell), 0, dim - 1).astype(int); chi = np.clip(np.floor((b[i, 3:] - lo) / cell), 0, dim - 1).astype(int)
        assert (blo <= clo).all() and (bhi >= chi).all()


def walk(t, o, d, tmax):
    """numpy restatement of gi.hip's top_begin / top_next: the instances whose box a ray is tested against, in order"""
    dim, lo, cell = np.array(t["dim"]), t["lo"].astype(np.float64), t["cell"].astype(np.float64)
    first, count = t["cells"] & ((1 << 20) - 1), t["cells"] >> 20
    hi = lo + dim * cell
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = 1.0 / d
        t0, t1 = (lo - o) * inv, (hi - o) * inv
    te, tx = np.nanmax(np.minimum(t0, t1)), np.nanmin(np.maximum(t0, t1))
    if te > tx or tx < 0:
        return []
    ts = max(te, 0.0)
    c = np.clip(np.floor((o + d * ts - lo) / cell), 0, dim - 1).astype(int)
    prev, out, t_end = None, [], min(tx, tmax)
    for _ in range(int(dim.sum()) + 4):
        idx = (c[2] * dim[1] + c[1]) * dim[0] + c[0]
        for i in t["items"][first[idx]:first[idx] + count[idx]]:
            rl, rh = int(t["ranges"][i, 0]), int(t["ranges"][i, 1])
            blo = np.array([rl & 255, (rl >> 9) & 255, rl >> 18]); bhi = np.array([rh & 255, (rh >> 9) & 255, rh >> 18])
            if prev is not None and (prev >= blo).all() and (prev <= bhi).all():
                continue                                           # listed in the cell the ray came from: dealt with there
            out.append(int(i))
        plane = lo + (c + (d > 0)) * cell
        with np.errstate(divide="ignore", invalid="ignore"):
            tk = np.where(d != 0, (plane - o) * inv, np.inf)
        a = int(np.argmin(tk))
        if not np.isfinite(tk[a]) or tk[a] > t_end:
            break
        nc = c.copy(); nc[a] += 1 if d[a] > 0 else -1
        if nc[a] < 0 or nc[a] >= dim[a]:
            break
        prev, c = c, nc
    return out


@pytest.mark.parametrize("n,seed", [(157, 11), (1200, 12)])
def test_a_ray_walking_the_grid_meets_every_box_on_its_way_once(n, seed):
    b = boxes_of(n, seed).astype(np.float64)
    t = api.top_level_build(b.astype(np.float32))
    rng = np.random.default_rng(seed)
    for k in range(300):
        o = rng.uniform(-320, 320, 3) * np.array([1.0, 0.15, 1.0])
        d = rng.normal(size=3); d /= np.linalg.norm(d)
        if k % 7 == 0:
            d[int(rng.integers(0, 3))] = 0.0; d /= np.linalg.norm(d)      # an axis-parallel plane of rays
        tmax = float(rng.choice([8.0, 200.0, 1e4]))
        seen = walk(t, o, d, tmax)
        assert len(seen) == len(set(seen)), "an instance came up twice on one path"
        with np.errstate(divide="ignore", invalid="ignore"):
            t0, t1 = (b[:, :3] - o) / d, (b[:, 3:] - o) / d
            te = np.nanmax(np.where(d != 0, np.minimum(t0, t1), -np.inf), axis=1); tx = np.nanmin(np.where(d != 0, np.maximum(t0, t1), np.inf), axis=1)
        inside = ((d != 0) | ((o >= b[:, :3]) & (o <= b[:, 3:]))).all(axis=1)
        hit = inside & (te <= tx) & (tx >= 0) & (te <= tmax)
        missing = set(np.nonzero(hit)[0].tolist()) - set(seen)
        assert not missing, (k, sorted(missing)[:5])


def test_slot_order_is_a_permutation_of_neighbours():
    n = 4157
    b = boxes_of(n, 9)
    t = api.top_level_build(b)
    order = t["slot_order"]
    assert sorted(order.tolist()) == list(range(n)) and t["n_groups"] == (n + 63) // 64
    assert api.top_level_build(boxes_of(200, 9))["n_groups"] == 0           # up to 256 instances the cull tests every box
    ctr = 0.5 * (b[:, :3] + b[:, 3:])
    ext = []
    for g in range(t["n_groups"]):
        ids = order[g * 64:(g + 1) * 64]
        ext.append(np.prod(b[ids, 3:].max(axis=0) - b[ids, :3].min(axis=0)))
    scene = np.prod(b[:, 3:].max(axis=0) - b[:, :3].min(axis=0))
    assert np.median(ext) < scene / 8.0                                     # a group box is a small part of the scene (in random order: most of it)
    assert np.linalg.norm(ctr[order[1:]] - ctr[order[:-1]], axis=1).mean() < 0.35 * np.linalg.norm(ctr[1:] - ctr[:-1], axis=1).mean()


def test_bad_arguments_are_refused():
    from dust_amd import _lib as L
    with pytest.raises(L.DustError):
        api.top_level_build(np.array([[1, 0, 0, 0, 1, 1]], np.float32))     # lo > hi
    # more boxes over one cell than a cell word counts (12 bits): refused rather than listed short
    stack = np.tile(np.array([[0, 0, 0, 1, 1, 1]], np.float32), (4200, 1))
    with pytest.raises(L.DustError):
        api.top_level_build(stack)


def test_an_elongated_scene_of_long_boxes_ends_in_one_cell_instead_of_spinning():
    """The coarsening loop of build_grid halves the density until the item list fits a cell word's 20 index bits. A corridor (aspect
    1000 : 1 : 1) of boxes that each span its length never got there: density x n <= 1 clamps the target to ONE cell, but the dims come
    from ext / cbrt(V) and stay (100, 1, 1) -- stationary state, 2 M items, a host hang in dust_hip_scene_commit (round 5's advisor).
    The loop now notices that its dims no longer change, forces one cell, and reports what that cell cannot list (more than 4095 boxes)."""
    import time
    from dust_amd import _lib as L
    n = 20000
    rng = np.random.default_rng(9)
    yz = rng.uniform(0.0, 0.9, (n, 2))
    b = np.concatenate([np.zeros((n, 1)), yz, np.full((n, 1), 1000.0), yz + 0.1], axis=1).astype(np.float32)
    t0 = time.time()
    with pytest.raises(L.DustError) as e:
        api.top_level_build(b)
    assert e.value.status == L.ERR_UNSUPPORTED and time.time() - t0 < 60
    # the same corridor with few enough boxes for one cell: a grid comes back, one cell or a few, every box listed
    t = api.top_level_build(b[:3000])
    first, count = t["cells"] & ((1 << 20) - 1), t["cells"] >> 20
    assert count.sum() == len(t["items"]) < (1 << 20) and count.max() <= 4095
