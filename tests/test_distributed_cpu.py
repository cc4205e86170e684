"""N>1 path on CPU: two gloo ranks shard a frame (row bands / samples) and gather to rank 0.
The per-rank pixels come from the oracle (no GPU here); the sharding and gather code is the one bench.py uses."""
import os
import socket
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent('''
    import os, sys
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch, torch.distributed as dist
    import oracle_lib as O, parity_util as P
    from dust_amd import _lib as L, sharding, synth
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    desc = P.small_scene(seed=4, n_models=2, n_instances=3, size=(24, 24, 24))
    s = P.oracle_scene(desc)
    sky = P.sky_state()
    cam = P.camera_for((70.0, 50.0, 80.0))
    W, H = 48, 37
    noise = synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    # --- row bands of one frame
    r0, r1 = sharding.band_rows(rank, world, H)
    g = P.render_oracle(s, cam, sky, W, H, passes, noise[1], 7, rows=(r0, r1))
    per = sharding.band_rows(0, world, H)[1]
    band = torch.zeros((per, W, 4), dtype=torch.int32)
    band[: r1 - r0] = torch.from_numpy(g.illuminance[r0:r1].astype(np.int32))
    parts = sharding.gather_to_root(dist, band)
    if rank == 0:
        full = P.render_oracle(s, cam, sky, W, H, passes, noise[1], 7)
        got = sharding.assemble_bands(parts, H).numpy().astype(np.uint16)
        assert got.shape == full.illuminance.shape and np.array_equal(got, full.illuminance), "band union != full frame"
    # --- samples of one view (spp sharding): rank r renders frame_index = step*world + r + 1
    fi = sharding.sample_frame_index(0, rank, world)
    g = P.render_oracle(s, cam, sky, W, H, passes, noise[fi % 4], synth.frame_rand(1, fi))
    frames = sharding.gather_to_root(dist, torch.from_numpy(g.illuminance.astype(np.int32)))
    if rank == 0:
        for r in range(world):
            f = sharding.sample_frame_index(0, r, world)
            ref = P.render_oracle(s, cam, sky, W, H, passes, noise[f % 4], synth.frame_rand(1, f))
            assert np.array_equal(frames[r].numpy().astype(np.uint16), ref.illuminance), f"sample {r} differs"
        assert not np.array_equal(frames[0].numpy(), frames[1].numpy()), "samples must differ (different noise slice / rand)"
    # --- multi-GPU GI exchange (sharding.gi_exchange_step; dust_hip.h dust_hip_pipeline_gi_exchange): the collectives
    # run for real over gloo, the two library kernels between them are restated in numpy (k_gi_export / k_gi_import)
    Wg, Hg, pool = 40, 27, 61
    per_rows = sharding.gi_band_rows(world, Hg)
    b0, b1 = min(Hg, rank * per_rows), min(Hg, (rank + 1) * per_rows)
    rng = np.random.default_rng(99)                       # same stream on every rank: the "whole frame" truth
    wants = rng.random(Wg * Hg) < 0.3                     # pixels whose final gather enqueues a surfel
    surf = rng.integers(1, 2**31 - 1, size=(Wg * Hg, 4)).astype(np.int32)
    stamp = np.where(rng.random(Wg * Hg) < 0.4, rng.integers(1, 500, Wg * Hg), 0).astype(np.int32)
    pix = np.arange(Wg * Hg)
    in_band = (pix // Wg >= b0) & (pix // Wg < b1)
    owner = np.zeros(pool, np.int32)
    for i in pix[wants & in_band]:                        # atomicMax(slot_owner[i % pool], i + 1) over this band only
        owner[i % pool] = max(owner[i % pool], i + 1)
    touched = np.zeros(world * per_rows * Wg, np.int32)
    touched[pix[in_band]] = stamp[in_band]
    merged = np.zeros(pool * 4, np.int32)
    t_owner, t_touched, t_merged = torch.from_numpy(owner), torch.from_numpy(touched), torch.from_numpy(merged)
    pool_state = np.full((pool, 4), -1, np.int32)
    stamped = set(stamp[in_band][stamp[in_band] != 0].tolist())   # what this rank's own final gather stamped

    def export_fn():
        o = t_owner.numpy()
        m = t_merged.numpy().reshape(pool, 4)
        m[:] = 0
        for sl in range(pool):
            if o[sl] and b0 <= (o[sl] - 1) // Wg < b1:
                m[sl] = surf[o[sl] - 1]

    def import_fn():
        t = t_touched.numpy()
        for i in pix[~in_band]:
            if t[i]:
                stamped.add(int(t[i]))
        o = t_owner.numpy()
        m = t_merged.numpy().reshape(pool, 4)
        for sl in range(pool):
            if o[sl]:
                pool_state[sl] = m[sl]
                o[sl] = 0

    sharding.gi_exchange_step(dist, rank, world, t_owner, t_touched, t_merged, per_rows * Wg, export_fn, import_fn)
    want_owner = np.zeros(pool, np.int64)
    for i in pix[wants]:
        want_owner[i % pool] = max(want_owner[i % pool], i + 1)
    want_pool = np.full((pool, 4), -1, np.int32)
    for sl in range(pool):
        if want_owner[sl]:
            want_pool[sl] = surf[want_owner[sl] - 1]
    assert np.array_equal(pool_state, want_pool), "merged surfel pool != single-process pool"
    assert stamped == set(stamp[stamp != 0].tolist()), "stamps of the other bands were not repeated"
    assert not t_owner.numpy().any()
    # --- the double-buffered asynchronous gather bench.py uses (step k's gather overlaps step k+1)
    ag = sharding.AsyncGather(dist, torch.zeros(4, dtype=torch.int32))
    for k in range(5):
        ag.submit(lambda buf, k=k: buf.fill_(100 * k + rank))
        if rank == 0 and k >= 1:
            pass
    ag.finish()
    if rank == 0:
        assert [int(x[0]) for x in ag.last()] == [400 + r for r in range(world)]
    # --- rotating root (bench.py's default: step k's frames are assembled on rank k % world, every xGMI link carries its share)
    ar = sharding.AsyncGather(dist, torch.zeros(4, dtype=torch.int32), rotate=True)
    for k in range(5):
        ar.submit(lambda buf, k=k: buf.fill_(100 * k + rank))
        ar.finish()
        assert ar.last_root() == k % world
        got = ar.last()
        if rank == k % world:
            assert [int(x[0]) for x in got] == [100 * k + r for r in range(world)]
        else:
            assert got is None
    # --- assembly by row slices (bench.py's default for spp sharding): one all-to-all, rank j gets slice j of every rank's frame
    tgs = [torch.zeros((2 * world, 3), dtype=torch.int32) for _ in range(2)]
    sl = sharding.AsyncGather(dist, tgs[0], slices=True)
    for k in range(5):
        sl.wait_slot(k % 2)
        tgs[k % 2][:] = (1000 * k + 100 * rank + torch.arange(2 * world, dtype=torch.int32)[:, None])  # row index in the last digits
        sl.submit_view(tgs[k % 2])
    sl.finish()
    got = sl.last()
    assert got.shape == tgs[0].shape
    for src in range(world):  # rows [2 * rank, 2 * rank + 2) of rank src's frame 4
        assert got[2 * src:2 * src + 2, 0].tolist() == [4000 + 100 * src + 2 * rank, 4000 + 100 * src + 2 * rank + 1]
    # --- the same without staging copies: two render targets used in turn, gathered straight from a row view of them
    targets = [torch.zeros((6, 4), dtype=torch.int32) for _ in range(2)]
    av = sharding.AsyncGather(dist, targets[0][2:4])
    for k in range(5):
        av.wait_slot(k % 2)                      # the gather that last read this target is done
        targets[k % 2].fill_(1000 * k + rank)    # "render" frame k into it
        av.submit_view(targets[k % 2][2:4])
    av.finish()
    if rank == 0:
        got = av.last()
        assert [tuple(x.shape) for x in got] == [(2, 4)] * world
        assert [int(x[0, 0]) for x in got] == [4000 + r for r in range(world)]
    # --- bench.py --shard bands: render targets padded to world * per_rows rows, every rank gathers an EQUAL-size slice
    # (its band padded to per_rows rows) straight from the target; heights where the last band is shorter (37 rows: 24 + 13)
    # and where a rank's band is empty (8 rows: 8 + 0)
    for Hb in (37, 8):
        per, rows_b, send = sharding.band_layout(rank, world, Hb)
        assert send[1] - send[0] == per and rows_b[1] - rows_b[0] <= per
        full = P.render_oracle(s, cam, sky, W, Hb, passes, noise[1], 7)
        tg = [torch.zeros((world * per, W, 4), dtype=torch.int32) for _ in range(2)]
        bg = sharding.AsyncGather(dist, tg[0][send[0]:send[1]])
        for k in range(3):
            bg.wait_slot(k % 2)
            if rows_b[0] < rows_b[1]:
                gb = P.render_oracle(s, cam, sky, W, Hb, passes, noise[1], 7, rows=rows_b)
                tg[k % 2][rows_b[0]:rows_b[1]] = torch.from_numpy(gb.illuminance[rows_b[0]:rows_b[1]].astype(np.int32))
            bg.submit_view(tg[k % 2][send[0]:send[1]])
        bg.finish()
        if rank == 0:
            frame = torch.cat(bg.last(), dim=0)[:Hb].numpy().astype(np.uint16)
            assert np.array_equal(frame, full.illuminance), f"padded band gather != full frame at {Hb} rows"
    # --- bench.py --band-cuts cost: bands of unequal height (equal measured cost), cuts agreed by broadcast from rank 0; the send
    # slice is [r0, r0 + per_rows) of a target with H + per_rows rows, the root keeps each part's first r1 - r0 rows
    Hc = 40
    mine = sharding.balanced_cuts([5.0 + rank, 5.0, 1.0, 1.0, 1.0], world, Hc)   # (each rank's own map differs a little: rank 0's counts)
    agreed = torch.tensor(mine, dtype=torch.int64)
    dist.broadcast(agreed, src=0)
    cuts = [int(v) for v in agreed.tolist()]
    assert cuts == [0, 8, 40]
    per, rows_c, send_c = sharding.layout_from_cuts(rank, cuts, Hc)
    assert per == 32 and send_c[1] - send_c[0] == per
    full = P.render_oracle(s, cam, sky, W, Hc, passes, noise[1], 7)
    tgc = [torch.zeros((Hc + per, W, 4), dtype=torch.int32) for _ in range(2)]
    cg = sharding.AsyncGather(dist, tgc[0][send_c[0]:send_c[1]], rotate=True)
    for k in range(4):
        cg.wait_slot(k % 2)
        gb = P.render_oracle(s, cam, sky, W, Hc, passes, noise[1], 7, rows=rows_c)
        tgc[k % 2][rows_c[0]:rows_c[1]] = torch.from_numpy(gb.illuminance[rows_c[0]:rows_c[1]].astype(np.int32))
        cg.submit_view(tgc[k % 2][send_c[0]:send_c[1]])
        root = cg.last_root()
        cg.finish()
        assert root == k % world
        if rank == root:
            frame = sharding.assemble_bands_from_cuts(cg.last(), cuts).numpy().astype(np.uint16)
            assert np.array_equal(frame, full.illuminance), "bands of equal cost != full frame"
    if rank == 0:
        print("distributed ok")
    dist.barrier()
    dist.destroy_process_group()
''')


def test_band_rows_partition():
    sys.path.insert(0, ROOT)
    from dust_amd import sharding
    for h, world in ((1080, 8), (1080, 2), (2160, 8), (37, 2), (8, 2), (20, 8)):
        lay = [sharding.band_layout(r, world, h) for r in range(world)]
        per = lay[0][0]
        assert per % 8 == 0 and world * per >= h
        assert [l[2] for l in lay] == [(r * per, (r + 1) * per) for r in range(world)]          # equal send slices
        assert sum(l[1][1] - l[1][0] for l in lay) == h and all(l[1][0] <= l[1][1] for l in lay)  # bands tile the frame
    for h in (1080, 37, 8, 2160):
        for world in (1, 2, 4, 8):
            bands = [sharding.band_rows(r, world, h) for r in range(world)]
            assert bands[0][0] == 0 and bands[-1][1] == h
            assert all(a[1] == b[0] for a, b in zip(bands, bands[1:]))
            assert all(a[0] % 8 == 0 for a in bands if a[1] > a[0])


def test_two_rank_gloo_gather():
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, "-c", f"ROOT={ROOT!r}\n" + WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-2000:]
    assert "distributed ok" in outs[0][0]


def test_cost_balanced_band_cuts():
    """sharding.balanced_cuts: bands of about equal measured cost (bench.py --band-cuts cost). Cuts are monotone, 8-row aligned, cover
    the frame; never worse than equal rows on the maximum band cost by more than one strip; unusable maps fall back to equal rows."""
    import numpy as np
    from dust_amd import sharding
    rng = np.random.default_rng(5)
    for _ in range(300):
        h = int(rng.integers(8, 1200))
        n = -(-h // 8)
        world = int(rng.choice([2, 3, 4, 8]))
        cost = rng.gamma(0.7, 1.0, n) * (1.0 + 4.0 * (rng.random(n) < 0.15))   # skewed, a few heavy strips
        cuts = sharding.balanced_cuts(cost, world, h)
        assert len(cuts) == world + 1 and cuts[0] == 0 and cuts[-1] == h
        assert all(a <= b for a, b in zip(cuts, cuts[1:])) and all(c % 8 == 0 or c == h for c in cuts)   # (a band past the frame's end is empty at h)
        band = lambda a, b: float(cost[a // 8:-(-b // 8)].sum())   # noqa: E731
        worst = max(band(a, b) for a, b in zip(cuts, cuts[1:]))
        per = -(-(-(-h // world)) // 8) * 8
        equal = [min(h, r * per) for r in range(world)] + [h]
        worst_equal = max(band(a, b) for a, b in zip(equal, equal[1:]))
        assert worst <= worst_equal + cost.max() + 1e-9
        lay = [sharding.layout_from_cuts(r, cuts, h) for r in range(world)]
        pers = {l[0] for l in lay}
        assert len(pers) == 1 and next(iter(pers)) % 8 == 0 and all(l[2][1] - l[2][0] == l[0] and l[1][1] - l[1][0] <= l[0] for l in lay)
        assert max(l[2][1] for l in lay) <= h + lay[0][0]   # the render target bench.py allocates (H + per_rows rows) holds every send slice
    for bad in (None, np.zeros(135), np.full(135, np.nan), np.ones(7)):
        assert sharding.balanced_cuts(bad, 8, 1080) == [0, 136, 272, 408, 544, 680, 816, 952, 1080]


def test_rebalance_cuts_converges_on_the_slowest_band():
    """sharding.rebalance_cuts (round 6): the cost map of ONE whole-frame launch mispredicts what a band's step takes under several frames
    in flight; a few rounds of proportional correction from measured band-step times bring the slowest band down to the mean. The 'true'
    cost here is the map's strip cost to the power 1.3 plus a per-band overhead (a model of the tail effects the map does not see)."""
    import numpy as np
    from dust_amd import sharding
    rng = np.random.default_rng(3)
    H, world = 1080, 8
    n = -(-H // 8)
    seen = rng.uniform(0.2, 3.0, n) * (1.0 + 2.0 * np.exp(-((np.arange(n) - 80) / 15.0) ** 2))   # what the whole-frame launch measured
    true = seen ** 1.3

    def step_ms(cuts):
        return [0.004 + true[cuts[r] // 8:-(-cuts[r + 1] // 8)].sum() * 1e-3 if cuts[r] < cuts[r + 1] else 0.0 for r in range(world)]
    cuts = sharding.balanced_cuts(seen, world, H)
    first = step_ms(cuts)
    cost = seen
    for _ in range(3):
        cuts, cost = sharding.rebalance_cuts(cost, cuts, step_ms(cuts), world, H)
        assert cuts[0] == 0 and cuts[-1] == H and all(a <= b for a, b in zip(cuts, cuts[1:])) and all(c % 8 == 0 for c in cuts[:-1])
    last = step_ms(cuts)
    assert max(last) < max(first) and max(last) / (sum(last) / world) < 1.06 < max(first) / (sum(first) / world), (first, last)
    # degenerate inputs: no map, a band without a time, more bands than strips
    c2, _ = sharding.rebalance_cuts(None, [0, 8, 16, 24], [1.0, 0.0, 2.0], 3, 24)
    assert c2[0] == 0 and c2[-1] == 24
    c3, _ = sharding.rebalance_cuts([1.0, 1.0], [0, 8, 8, 16, 16], [1.0, 0.0, 1.0, 0.0], 4, 16)
    assert c3[0] == 0 and c3[-1] == 16 and all(a <= b for a, b in zip(c3, c3[1:]))
