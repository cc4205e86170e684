#!/usr/bin/env python3
"""Randomised sweeps of the host runtime on one GPU (no oracle: the library against itself).
  bands   : a frame rendered as random row bands (any cut points, any order, random frames-in-flight setting) equals the frame rendered whole
  commits : random sequences of add_instance / set_transform / commit / render with nothing waited for in between; at random points the
            frame must equal the frame of a FRESH scene built from the instance list as it stands (the staging ring, the image
            reallocation when the scene grows, the dirty-range copy, the second stream)
  threads : several host threads, a context each, on one device at the same time
  schedule: big frames under random launch shapes against the default launch
  frames  : dust_hip_render_frames -- random numbers of frames (1 .. 19: launches of up to 8), frame sizes from a few tiles to 1080p, cameras, row bands,
            launch shapes, calls repeated so that measured tile orders come in, single-frame calls and commits in between, frames that cannot
            share a launch mixed in -- against the same frames one dust_hip_render_frame at a time
usage: stress_host.py bands|commits|threads|schedule|frames [n] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dust_amd import _lib as L, api, synth, scenes as S

PA = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
GI = PA | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
PLANES = (L.PLANE_DEPTH, L.PLANE_VOXEL_ID, L.PLANE_ALBEDO, L.PLANE_NORMAL, L.PLANE_ILLUMINANCE, L.PLANE_DENOISED, L.PLANE_MOTION)


def small_models(ctx, rng, pal, n):
    out = []
    for _ in range(n):
        size = tuple(int(v) for v in rng.integers(10, 50, 3))
        solid = rng.random(size) < 0.1
        x, y, z = np.nonzero(solid)
        xyzi = np.stack([x, y, z, rng.integers(0, 255, x.size)], axis=1).astype(np.uint8)
        if not len(xyzi):
            xyzi = np.array([[1, 1, 1, 3]], np.uint8)
        out.append(api.Model(ctx, *api.flatten_model(xyzi, size, pal), pal))
    return out


def rand_xf(rng):
    rots = [np.eye(3), np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]]), np.array([[-1, 0, 0], [0, 1, 0], [0, 0, -1]]), np.array([[1, 0, 0], [0, 0, -1], [0, 1, 0]])]
    m = np.zeros((3, 4), np.float32)
    m[:, :3] = rots[int(rng.integers(0, 4))]
    m[:, 3] = rng.uniform(-60, 60, 3)
    return m


def planes(pipe):
    return [pipe.read_plane(p) for p in PLANES]


def pipe_for(ctx, w, h, n0, n5):
    p = api.StandardPipeline(ctx, w, h)
    p.set_noise(0, n0); p.set_noise(5, n5)
    p.configure_gi(4093, 777)
    return p


def bands(n, first):
    ctx = api.Context(device=0)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = S.sky_state()
    bad = []
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        pal = synth.make_palette(seed)
        models = small_models(ctx, rng, pal, int(rng.integers(1, 4)))
        scene = api.Scene(ctx)
        for _ in range(int(rng.integers(1, 12))):
            scene.add_instance(models[int(rng.integers(0, len(models)))], rand_xf(rng).reshape(12))
        scene.commit()
        w, h = int(rng.integers(8, 300)), int(rng.integers(8, 200))
        eye = tuple(float(v) for v in rng.uniform(-100, 100, 3))
        cam = S.camera_for(eye if abs(eye[0]) + abs(eye[2]) > 1e-3 else (1.0, eye[1], eye[2]))
        whole = pipe_for(ctx, w, h, n0, n5)
        whole.render(scene, cam, sky, PA, frame_index=1, rand=seed)
        want = planes(whole)
        cuts = sorted(set([0, h] + [int(v) for v in rng.integers(1, h, int(rng.integers(1, 9)))]))
        order = list(range(len(cuts) - 1))
        rng.shuffle(order)
        part = pipe_for(ctx, w, h, n0, n5)
        part.set_frames_in_flight(int(rng.integers(1, 9)))
        for i in order:
            part.render(scene, cam, sky, PA, frame_index=1, rand=seed, rows=(cuts[i], cuts[i + 1]))
        got = planes(part)
        diff = [pl for pl, a, b in zip(PLANES, want, got) if not np.array_equal(a, b)]
        if diff:
            bad.append(seed)
            print(f"seed {seed}: {w}x{h} cuts {cuts} order {order}: planes {diff} differ", flush=True)
    return bad


def mat4(o2w):
    m = np.eye(4, dtype=np.float32)
    m[:3, :] = np.asarray(o2w, np.float32).reshape(3, 4)
    return m.T.reshape(16)


def commits(n, first):
    ctx = api.Context(device=0)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = S.sky_state()
    bad = []
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        pal = synth.make_palette(seed)
        models = small_models(ctx, rng, pal, int(rng.integers(1, 4)))
        scene = api.Scene(ctx)
        inst = []   # [model index, current transform, previous transform]
        w, h = int(rng.integers(32, 160)), int(rng.integers(24, 100))
        cam = S.camera_for(tuple(float(v) for v in rng.uniform(60, 140, 3)))
        gi = bool(rng.integers(0, 2))
        passes = GI if gi else PA
        pipe = pipe_for(ctx, w, h, n0, n5)
        frame = 0
        ok = True
        for step in range(int(rng.integers(10, 60))):
            op = int(rng.integers(0, 10))
            if op < 3 or not inst:
                mi = int(rng.integers(0, len(models)))
                t = rand_xf(rng)
                scene.add_instance(models[mi], t.reshape(12))
                inst.append([mi, t, t])
            elif op < 8:
                j = int(rng.integers(0, len(inst)))
                t = inst[j][1].copy()
                t[:, 3] += rng.uniform(-3, 3, 3).astype(np.float32)
                scene.set_transform(j, t.reshape(12), mat4(inst[j][1]))
                inst[j][2], inst[j][1] = inst[j][1], t
            else:
                scene.commit()
                frame += 1
                pipe.render(scene, cam, sky, passes, frame_index=frame, rand=seed + frame)
                if rng.integers(0, 3) == 0 and not gi:   # (a GI frame depends on the frames before it: compared at the end only)
                    fresh = api.Scene(ctx)
                    for mi, t, tp in inst:
                        fresh.add_instance(models[mi], t.reshape(12), mat4(tp))
                    fresh.commit()
                    ref = pipe_for(ctx, w, h, n0, n5)
                    ref.render(fresh, cam, sky, passes, frame_index=frame, rand=seed + frame)
                    a, b = planes(ref), planes(pipe)
                    hit = np.isfinite(a[0])
                    # (the pixel passes leave img_illuminance_denoised of a HIT pixel alone -- miss.rmiss writes it --, so the kept pipeline
                    # still holds there what an earlier frame's sky left: compared where this frame wrote it)
                    diff = [pl for pl, x, y in zip(PLANES, a, b) if not (np.array_equal(x[~hit], y[~hit]) if pl == L.PLANE_DENOISED else np.array_equal(x[hit], y[hit]))]
                    diff += ["depth"] if not np.array_equal(a[0], b[0]) else []
                    if diff:
                        ok = False
                        d1 = (a[1] != b[1]) & hit
                        ys, xs = np.nonzero(d1)
                        ex = [(int(x), int(y), hex(int(a[1][y, x])), hex(int(b[1][y, x]))) for y, x in list(zip(ys, xs))[:3]]
                        print(f"seed {seed} step {step} frame {frame}: {len(inst)} instances, planes {diff} differ from a fresh scene's; voxel_id (fresh, kept) {ex}, {int(d1.sum())} px", flush=True)
                        break
        if not ok:
            bad.append(seed)
    return bad


def threads(n, first):
    """n rounds of 6 host threads, each with a context, scene and pipeline of its own on the same device, creating, rendering and
    destroying at the same time (contexts are externally synchronised one by one; the library's process-wide state must not care):
    every thread's frames equal the ones a single thread renders afterwards."""
    import threading
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = S.sky_state()
    bad = []
    for seed in range(first, first + n):
        results, errors = {}, []

        def work(tid, keep):
            try:
                rng = np.random.default_rng(seed * 100 + tid)
                ctx = api.Context(device=0)
                pal = synth.make_palette(tid)
                models = small_models(ctx, rng, pal, 2)
                scene = api.Scene(ctx)
                for _ in range(int(rng.integers(1, 10))):
                    scene.add_instance(models[int(rng.integers(0, 2))], rand_xf(rng).reshape(12))
                scene.commit()
                w, h = int(rng.integers(40, 200)), int(rng.integers(30, 120))
                cam = S.camera_for(tuple(float(v) for v in rng.uniform(60, 140, 3)))
                pipe = pipe_for(ctx, w, h, n0, n5)
                for f in range(1, 5):
                    pipe.render(scene, cam, sky, GI, frame_index=f, rand=seed + f)
                hh, sp = pipe.read_gi()
                keep[tid] = planes(pipe) + [hh, sp.view(np.uint32).copy()]
            except Exception as e:   # noqa: BLE001
                errors.append((tid, repr(e)))

        par, ser = {}, {}
        ts = [threading.Thread(target=work, args=(t, par)) for t in range(6)]
        [t.start() for t in ts]
        [t.join() for t in ts]
        for t in range(6):
            work(t, ser)
        if errors:
            bad.append(seed); print(f"seed {seed}: errors {errors[:3]}", flush=True); continue
        for t in range(6):
            if not all(np.array_equal(a, b) for a, b in zip(par[t], ser[t])):
                bad.append(seed); print(f"seed {seed}: thread {t} differs from the single-threaded run", flush=True); break
    return bad


def schedule(n, first):
    """The tile hand-out at scale: frames of 3 600 to 32 400 tiles under random launch shapes -- block size, workgroups per CU, reserved
    slots, frames-in-flight share (down to a sixteenth of the slots: a hundred tiles per wave), dealt rounds, tile order on or off,
    several frames in a row so that measured orders are in use -- must equal the default launch plane for plane, GI state included."""
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = S.sky_state()
    bad = []
    keys = ("DUST_HIP_BLOCK", "DUST_HIP_BLOCKS_PER_CU", "DUST_HIP_RESERVE_BLOCKS", "DUST_HIP_STATIC_ROUNDS", "DUST_HIP_NO_TILE_ORDER", "DUST_HIP_NO_SIDE_STREAM")
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        for k_ in keys:
            os.environ.pop(k_, None)
        ctx = api.Context(device=0)
        pal = synth.make_palette(seed)
        models = small_models(ctx, rng, pal, 3)
        scene = api.Scene(ctx)
        for _ in range(int(rng.integers(10, 40))):
            scene.add_instance(models[int(rng.integers(0, 3))], rand_xf(rng).reshape(12))
        scene.commit()
        w, h = [(640, 360), (1280, 720), (1920, 1080)][int(rng.integers(0, 3))]
        cam = S.camera_for(tuple(float(v) for v in rng.uniform(60, 140, 3)))
        frames = int(rng.integers(2, 5))

        def run(fif):
            pipe = pipe_for(ctx, w, h, n0, n5)
            if fif > 1:
                pipe.set_frames_in_flight(fif)
            for f in range(1, frames + 1):
                pipe.render(scene, cam, sky, GI, frame_index=f, rand=seed + f)
            hh, sp = pipe.read_gi()
            return planes(pipe) + [hh, sp.view(np.uint32).copy()]

        want = run(1)
        env = {"DUST_HIP_BLOCK": str(int(rng.choice([64, 128, 256, 512]))), "DUST_HIP_BLOCKS_PER_CU": str(int(rng.choice([1, 2]))),
               "DUST_HIP_RESERVE_BLOCKS": str(int(rng.choice([0, 32, 256]))), "DUST_HIP_STATIC_ROUNDS": str(int(rng.choice([0, 1, 2, 3])))}
        if rng.random() < 0.3:
            env["DUST_HIP_NO_TILE_ORDER"] = "1"
        if rng.random() < 0.3:
            env["DUST_HIP_NO_SIDE_STREAM"] = "1"
        os.environ.update(env)
        fif = int(rng.choice([1, 2, 4, 8, 16]))
        got = run(fif)
        for k_ in keys:
            os.environ.pop(k_, None)
        diff = [i for i, (a, b) in enumerate(zip(want, got)) if not np.array_equal(a, b)]
        if diff:
            bad.append(seed)
            print(f"seed {seed}: {w}x{h}, {frames} frames, {env}, frames in flight {fif}: outputs {diff} differ", flush=True)
    return bad


def frames(n, first):
    """Several frames per launch (k_primary_ao_batch) against a launch per frame: every plane of every frame, bit for bit."""
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    n0 = synth.stbn_scalar(layers=4)
    sky = S.sky_state()
    bad = []
    keys = ("DUST_HIP_BLOCK", "DUST_HIP_BLOCKS_PER_CU", "DUST_HIP_RESERVE_BLOCKS", "DUST_HIP_STATIC_ROUNDS", "DUST_HIP_NO_TILE_ORDER", "DUST_HIP_NO_LDS_BOXES",
            "DUST_HIP_EQUAL_BANDS", "DUST_HIP_NO_WIDE_FUSED", "DUST_HIP_FORCE_MOVING")
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        for k_ in keys:
            os.environ.pop(k_, None)
        if rng.random() < 0.5:   # a random launch shape, the same for every pipeline of the case
            os.environ["DUST_HIP_BLOCK"] = str(int(rng.choice([128, 256, 512])))
            os.environ["DUST_HIP_BLOCKS_PER_CU"] = str(int(rng.choice([1, 2])))
            os.environ["DUST_HIP_STATIC_ROUNDS"] = str(int(rng.choice([0, 1, 2])))
        for name in ("NO_TILE_ORDER", "NO_LDS_BOXES", "EQUAL_BANDS", "NO_WIDE_FUSED", "FORCE_MOVING"):
            if rng.random() < 0.25:
                os.environ["DUST_HIP_" + name] = "1"
        if rng.random() < 0.2:
            os.environ["DUST_HIP_RESERVE_BLOCKS"] = "32"
        ctx = api.Context(device=0)
        pal = synth.make_palette(seed)
        models = small_models(ctx, rng, pal, int(rng.integers(1, 4)))
        scene, twin = api.Scene(ctx), api.Scene(ctx)   # (the twin: the same scene, moved and committed frame by frame for the single-frame calls)
        xfs = [rand_xf(rng).reshape(12) for _ in range(int(rng.integers(1, 30)))]
        which = [int(rng.integers(0, len(models))) for _ in xfs]
        ids = [scene.add_instance(models[w_], x) for w_, x in zip(which, xfs)]
        for w_, x in zip(which, xfs):
            twin.add_instance(models[w_], x)
        scene.commit(); twin.commit()
        with_moves = rng.random() < 0.4   # every frame comes with the instances that moved before it (DustHipFrameMoves)
        big = rng.random() < 0.15
        w, h = ((1920, 1080) if rng.random() < 0.5 else (1280, 720)) if big else (int(rng.integers(8, 400)), int(rng.integers(8, 260)))
        nf = int(rng.integers(1, 20 if not big else 7))
        rows = (0, 0)
        if rng.random() < 0.3 and h > 16:
            r0 = int(rng.integers(0, h - 8)); rows = (r0, int(rng.integers(r0 + 1, h + 1)))

        def cam():
            eye = tuple(float(v) for v in rng.uniform(-120, 120, 3))
            return S.camera_for(eye if abs(eye[0]) + abs(eye[2]) > 1e-3 else (1.0, eye[1], eye[2]))
        pipes = [api.StandardPipeline(ctx, w, h) for _ in range(nf)]
        alone = [api.StandardPipeline(ctx, w, h) for _ in range(nf)]
        for p_ in pipes + alone:
            p_.set_noise(5, n5)
        odd = int(rng.integers(0, nf)) if (nf > 2 and rng.random() < 0.3) else -1   # one pipeline of another frame size in the middle: splits the call
        if odd >= 0:
            pipes[odd] = api.StandardPipeline(ctx, w + 8, h); alone[odd] = api.StandardPipeline(ctx, w + 8, h)
            pipes[odd].set_noise(5, n5); alone[odd].set_noise(5, n5)
        rounds = int(rng.integers(1, 12 if not big else 4))
        f = 1
        for rd in range(rounds):
            cams = [cam() for _ in range(nf)] if rng.random() < 0.5 else [cam()] * nf
            idx = [f + i for i in range(nf)]
            rnd = [int(rng.integers(0, 1 << 32)) for _ in range(nf)]
            moves = None
            if with_moves:
                moves = [[(ids[int(rng.integers(0, len(ids)))], rand_xf(rng).reshape(12), mat4(rand_xf(rng))) for _ in range(int(rng.integers(0, 3)))] for _ in range(nf)]
            api.StandardPipeline.render_frames(pipes, scene, cams, sky, PA, idx, rnd, rows=rows, moves=moves)
            for i in range(nf):
                for j_, x_, p_ in (moves[i] if moves else []):
                    twin.set_transform(j_, x_, p_)
                if moves and moves[i]:
                    twin.commit()
                alone[i].render(twin, cams[i], sky, PA, frame_index=idx[i], rand=rnd[i], rows=rows)
            f += nf
            if rng.random() < 0.3:   # a frame of its own on one of the pipelines, and a moved instance, between two calls
                j = int(rng.integers(0, nf))
                pipes[j].render(scene, cams[j], sky, PA, frame_index=f, rand=7, rows=rows)
                alone[j].render(twin, cams[j], sky, PA, frame_index=f, rand=7, rows=rows)
                mv_id, mv_xf = ids[int(rng.integers(0, len(ids)))], rand_xf(rng).reshape(12)
                scene.set_transform(mv_id, mv_xf); scene.commit()
                twin.set_transform(mv_id, mv_xf); twin.commit()
        diff = [(i, pl) for i in range(nf) for pl, a, b in zip(PLANES, planes(pipes[i]), planes(alone[i])) if not np.array_equal(a, b)]
        if diff:
            bad.append(seed)
            print(f"seed {seed}: {nf} frames of {w}x{h} rows {rows}, {rounds} calls, odd {odd}, moves {with_moves}, env { {k_: os.environ[k_] for k_ in keys if k_ in os.environ} }: (frame, plane) {diff[:6]} differ", flush=True)
    for k_ in keys:
        os.environ.pop(k_, None)
    return bad


if __name__ == "__main__":
    what = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 100
    first = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    t0 = time.time()
    bad = {"bands": bands, "commits": commits, "threads": threads, "schedule": schedule, "frames": frames}[what](n, first)
    print(f"{what}: {n} cases, {len(bad)} with mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
