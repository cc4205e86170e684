#!/usr/bin/env python3
"""Randomised GPU-vs-oracle parity sweep (GPU box; minutes, not part of pytest).
Many small random scenes -- overlapping instances on aligned and half-voxel-shifted lattices, axis rotations and mirrors,
cameras inside and outside, axis-parallel views -- through all five passes for a few frames each; integer planes, hit
distances and GI state must match the oracle bit for bit, radiance within 1e-3.
usage: stress_parity.py [n_scenes] [first_seed] [position of first_seed in the sweep to reproduce]      STRESS_BIG=1: larger scenes; STRESS_FULL=1: one 256^3 model filled to its faces; STRESS_MANY=1: 90-160 models, 150-900 instances; STRESS_SWITCHES=1: a random combination of the
library's diagnostic switches per scene; STRESS_DEEP=1: 4096^3 models (run_deep); STRESS_FULLGI=1: the castle stand-in with GI at the reference's
sizes (run_fullgi: seconds per scene)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_lib as O  # noqa: E402
import parity_util as P  # noqa: E402
from dust_amd import _lib as L, api, synth  # noqa: E402

def random_switches(rng):
    """a random combination of the library's diagnostic switches (read when a pipeline is created) and a context of its own with a
    random LDS budget for staged roots"""
    for name in ("NO_FUSE", "NO_TILE_ORDER", "NO_GATHER_ORDER", "NO_SURFEL_SORT", "NO_LDS_BOXES", "NO_SIDE_STREAM",
                 "EQUAL_BANDS", "NO_DILATE", "NO_WIDE_FUSED", "FORCE_MOVING",   # (round 4: the hand-out's knobs)
                 "RAY_STREAM", "NO_STREAM_LDS", "FLAT_CULL", "PACKET_GI"):   # (round 5: the GI passes as ray streams, the cull's hierarchy off;
                                                                             #  RAY_STREAM / PACKET_GI / NO_SIDE_STREAM reach the library through api.StandardPipeline's config)
        os.environ.pop("DUST_HIP_" + name, None)
        if rng.random() < 0.3:
            os.environ["DUST_HIP_" + name] = "1"
    os.environ.pop("DUST_HIP_BLOCK", None)
    if rng.random() < 0.6:   # (unset: the fused kernel may take the 1024-thread shape)
        os.environ["DUST_HIP_BLOCK"] = str(int(rng.choice([128, 256, 512])))
    os.environ["DUST_HIP_BLOCKS_PER_CU"] = str(int(rng.choice([1, 2])))
    os.environ["DUST_HIP_STREAM_REFILL"] = str(int(rng.choice([1, 8, 16, 48, 64])))
    os.environ["DUST_HIP_GRID_DENSITY"] = str(float(rng.choice([0.01, 1.0, 12.0, 100.0])))
    return api.Context(device=0, lds_root_bytes=int(rng.choice([0, 640, 1280, 64 * 1024])))


def run(n_scenes, seed0, big=False, verbose=True, k0=0):
    """Returns the seeds whose frames did not match. (k0: position of seed0 in the sweep that is being reproduced -- some choices go by position)"""
    ctx = api.Context(device=0)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    failed = []
    many, switches = os.environ.get("STRESS_MANY") == "1", os.environ.get("STRESS_SWITCHES") == "1"
    for k in range(k0, k0 + n_scenes):
        seed = seed0 + k - k0
        rng = np.random.default_rng(seed)
        if switches:
            ctx = random_switches(rng)
        if os.environ.get("STRESS_FULL") == "1":   # one model that fills its 256^3 tree to the faces and corners, 1 to 3 instances
            sz = (256, 256, 256)
            pts = rng.integers(0, 256, (int(rng.integers(2000, 20000)), 3))
            pts = np.concatenate([pts, rng.integers(0, 256, (600, 3)) * np.array([1, 1, 0]) + np.array([0, 0, 255]), rng.integers(0, 256, (600, 3)) * np.array([0, 1, 1]),
                                  np.array([[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 0]])])
            pts = np.unique(pts, axis=0)   # (the loader refuses a voxel listed twice)
            xyzi = np.concatenate([pts, rng.integers(0, 255, (len(pts), 1))], axis=1).astype(np.uint8)
            pal = synth.make_palette(seed)
            inst = []
            for i in range(int(rng.integers(1, 4))):
                m = np.zeros((3, 4), np.float32)
                m[:, :3] = np.eye(3) if i % 2 == 0 else np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]])
                m[:, 3] = rng.integers(-160, -90, 3) + i * 37.5
                inst.append((0, m.reshape(12)))
            desc = P.SceneDesc([api.flatten_model(xyzi, sz, pal)], pal, inst)
        elif many:       # more models than LDS holds roots of, more instances than a packet's candidate list holds
            desc = P.small_scene(seed=seed, n_models=int(rng.integers(90, 160)), n_instances=int(rng.integers(150, 900)),
                                 size=tuple(int(v) for v in rng.integers(8, 20, 3)))
        else:
            desc = P.small_scene(seed=seed, n_models=int(rng.integers(1, 6 if big else 4)), n_instances=int(rng.integers(1, 25 if big else 9)),
                                 size=tuple(int(v) for v in rng.integers(12, 110 if big else 40, 3)))
        if k % 3 == 0:   # stack instances on top of each other: equal-t ties between instances
            desc.instances = [(m, t.copy()) for m, t in desc.instances]
            for j in range(1, len(desc.instances)):
                desc.instances[j][1][3::4] = desc.instances[0][1][3::4] + rng.integers(-2, 3, 3) * 4.0
        if k % 2 == 1 and os.environ.get("STRESS_AFFINE", "1") == "1":   # arbitrary rotations, non-uniform scales, fractional offsets
            desc.instances = [(m, t.copy()) for m, t in desc.instances]
            for j in range(len(desc.instances)):
                q = rng.normal(size=4)
                q /= np.linalg.norm(q)
                qw, qx, qy, qz = q
                R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                              [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                              [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
                A = (R * rng.uniform(0.5, 2.0, 3)[None, :]).astype(np.float32)
                t = desc.instances[j][1].reshape(3, 4)
                t[:, :3] = A
                t[:, 3] = rng.uniform(-50, 50, 3).astype(np.float32)
        scene, oscene = P.hip_scene(ctx, desc), P.oracle_scene(desc)
        eye = rng.uniform(-90, 90, 3) * (2.5 if os.environ.get("STRESS_FULL") == "1" else 1.0)
        if k % 5 == 0:
            eye = np.round(eye / 4.0) * 4.0   # on the lattice
        if k % 7 == 0:
            eye[int(rng.integers(0, 3))] = 0.0
        if abs(eye[0]) + abs(eye[2]) < 1e-3:
            eye[0] = 1.0   # straight up/down the y axis has no look-at frame (NaN camera): not a traversal case
        cam = P.camera_for(tuple(float(v) for v in eye), target=tuple(float(v) for v in rng.uniform(-10, 10, 3)) if k % 4 else (0.0, 0.0, 0.0))
        w, h = int(rng.integers(40, 140)), int(rng.integers(24, 90))
        cap, pool = int(rng.choice([61, 509, 4093, 1 << 14])), int(rng.choice([97, 777, 2048]))
        pipe = api.StandardPipeline(ctx, w, h)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(cap, pool)
        gi = O.GI(cap, pool)
        f = 0
        try:
            for f in range(1, 4):
                rnd = synth.frame_rand(seed, f)
                if f > 1 and k % 6 == 2 and not many:   # instances move between frames (castle.rs:287-291): transforms through
                    oscene = O.Scene()                     # dust_hip_scene_set_transform + commit, motion vectors against the previous frame's
                    for b, m in desc.models:
                        oscene.add_model(b, m, desc.palette)
                    moved = []
                    for j, (mid, t) in enumerate(desc.instances):
                        t = np.array(t, np.float32).reshape(3, 4)
                        cur = t.copy()
                        if rng.random() < 0.6:
                            cur[:, 3] += rng.uniform(-2.5, 2.5, 3).astype(np.float32)
                        prev = np.eye(4, dtype=np.float32)
                        prev[:3, :] = t
                        prev = prev.T.reshape(16)   # column-major mat4 of last frame's transform
                        scene.set_transform(j, cur.reshape(12), prev)
                        oscene.add_instance(mid, cur.reshape(12), prev)
                        moved.append((mid, cur.reshape(12)))
                    desc.instances = moved
                    scene.commit()
                    oscene.commit()
                pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
                g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f)
                res = P.compare_gbuffers(g, P.read_hip_gbuffer(pipe))
                P.assert_parity(res)
                assert res.get("illuminance_rel_l2", 0.0) <= 1e-3 and res.get("denoised_rel_l2", 0.0) <= 1e-3, res
                oh, op = gi.hash(), gi.pool()
                hh, hp = pipe.read_gi()
                assert np.array_equal(oh["fingerprint"], hh[:, 0]), "hash fingerprints"
                assert np.array_equal(oh["sample_count"], hh[:, 2] >> 16), "hash sample counts"
                assert np.array_equal(oh["last_accessed_frame"], hh[:, 2] & 0xFFFF), "hash LRU stamps"
                assert np.array_equal(op["direction"], hp["direction"]), "surfel pool"
        except AssertionError as e:
            failed.append(seed)
            if verbose:
                print(f"seed {seed}: MISMATCH frame {f} ({w}x{h}, {len(desc.instances)} instances, eye {eye}): {str(e)[:300]}", flush=True)
    return failed


def run_deep(n_scenes, seed0, verbose=True, k0=0):
    """The same sweep over 4096^3 models (three-level hierarchy): clusters of 16-cells with 1 to 12 bricks each (the DEEP kernels'
    whole-cell test and their 4-cell / octant walk in one frame), every third scene with a 256^3 model beside it (two-level
    models in a DEEP launch), cameras inside, outside and on cell planes."""
    ctx = api.Context(device=0)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    failed = []
    for k in range(k0, k0 + n_scenes):
        seed = seed0 + k - k0
        rng = np.random.default_rng(seed)
        if os.environ.get("STRESS_SWITCHES") == "1":
            ctx = random_switches(np.random.default_rng(seed ^ 0x5A5A))
        half = int(rng.integers(2, 24))                     # the cluster's half width in 16-cells
        c0 = int(rng.integers(half, 256 - half))
        fill = float(rng.choice([0.02, 0.1, 0.4]))
        n_cells = max(4, int(fill * (2 * half) ** 3))
        blocks, mats, pal = P.clustered_deep_model(seed=seed, n_cells=min(n_cells, 30000), cell_lo=c0 - half, cell_hi=c0 + half,
                                                   max_bricks=int(rng.choice([2, 6, 12, 40])))
        centre = 16.0 * c0
        xf = np.eye(3, 4, dtype=np.float32)
        xf[:, 3] = -centre
        if k % 5 == 1:   # the deep instance rotated, scaled and shifted by fractions: object-space rays with no zero or shared components
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            qw, qx, qy, qz = q
            R = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                          [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                          [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
            A = R * rng.uniform(0.6, 1.6, 3)[None, :]
            xf[:, :3] = A.astype(np.float32)
            xf[:, 3] = (-(A @ np.full(3, centre)) + rng.uniform(-3, 3, 3)).astype(np.float32)   # the cluster's centre stays near the origin
        model = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
        scene, oscene = api.Scene(ctx), O.Scene()
        oscene.add_model(blocks, mats, pal, extent=4096)
        scene.add_instance(model, xf.reshape(12))
        oscene.add_instance(0, xf.reshape(12))
        keep = [model]
        if k % 7 == 3:   # the same deep model a second time, shifted by a few voxels and turned: overlapping three-level instances
            xf2 = xf.copy()
            xf2[:, :3] = xf[:, :3] @ np.array([[0, 0, 1], [0, 1, 0], [-1, 0, 0]], np.float32)
            xf2[:, 3] = (-(xf2[:, :3] @ np.full(3, centre, np.float32)) + rng.integers(-6, 7, 3)).astype(np.float32)
            scene.add_instance(model, xf2.reshape(12))
            oscene.add_instance(0, xf2.reshape(12))
        if k % 3 == 0:
            small = P.small_scene(seed=seed, n_models=1, n_instances=2, size=(40, 40, 40))
            m2 = api.Model(ctx, small.models[0][0], small.models[0][1], pal)
            keep.append(m2)
            oscene.add_model(small.models[0][0], small.models[0][1], pal)
            for _, t in small.instances:
                scene.add_instance(m2, t)
                oscene.add_instance(1, t)
        scene.commit()
        oscene.commit()
        reach = 16.0 * half
        eye = rng.uniform(-1.5 * reach, 1.5 * reach, 3)
        if k % 4 == 0:
            eye = np.round(eye / 16.0) * 16.0   # on 16-cell planes
        if abs(eye[0]) + abs(eye[2]) < 1e-3:
            eye[0] = 3.0
        cam = P.camera_for(tuple(float(v) for v in eye), target=tuple(float(v) for v in rng.uniform(-0.3 * reach, 0.3 * reach, 3)))
        w, h = int(rng.integers(40, 140)), int(rng.integers(24, 90))
        cap, pool = int(rng.choice([509, 4093, 1 << 14])), int(rng.choice([97, 777, 2048]))
        pipe = api.StandardPipeline(ctx, w, h)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(cap, pool)
        gi = O.GI(cap, pool)
        f = 0
        try:
            for f in range(1, 3):
                rnd = synth.frame_rand(seed, f)
                pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
                g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f)
                res = P.compare_gbuffers(g, P.read_hip_gbuffer(pipe))
                P.assert_parity(res)
                assert res.get("illuminance_rel_l2", 0.0) <= 1e-3 and res.get("denoised_rel_l2", 0.0) <= 1e-3, res
                oh, op = gi.hash(), gi.pool()
                hh, hp = pipe.read_gi()
                assert np.array_equal(oh["fingerprint"], hh[:, 0]), "hash fingerprints"
                assert np.array_equal(oh["sample_count"], hh[:, 2] >> 16), "hash sample counts"
                assert np.array_equal(op["direction"], hp["direction"]), "surfel pool"
        except AssertionError as e:
            failed.append(seed)
            if verbose:
                print(f"deep seed {seed}: MISMATCH frame {f} ({w}x{h}, {len(blocks)} bricks, eye {eye}): {str(e)[:300]}", flush=True)
    return failed


def run_fullgi(n_scenes, seed0, verbose=True):
    """STRESS_FULLGI=1: GI at (nearly) the reference's sizes -- the castle stand-in at a random scale and camera, frames of up to 1920 x 1080,
    the 32 Mi-entry hash or a smaller prime-sized one, the 345 600-slot pool or a smaller one, two or three frames, the oracle's pixel
    passes threaded over rows and its GI passes over bands / surfel ranges (orc_pass_*_mt). Seconds per scene."""
    import threading
    ctx = api.Context(device=0)
    sky = P.sky_state()
    n0, n5 = synth.stbn_scalar(layers=8), synth.stbn_unitvec3_cosine(layers=8)
    failed = []
    for k in range(n_scenes):
        seed = seed0 + k
        rng = np.random.default_rng(seed)
        scale = float(rng.choice([0.25, 0.5, 1.0]))
        data, _ = synth.castle_scene(seed=0xD057 + (seed % 3), scale=scale)
        desc = P.SceneDesc.from_vox(data)
        scene, oscene = P.hip_scene(ctx, desc), P.oracle_scene(desc)
        w, h = [(1920, 1080), (1280, 720), (960, 536), (1000, 600)][int(rng.integers(0, 4))]
        cap = int(rng.choice([32 * 1024 * 1024, 4194301, 262139]))
        pool = int(rng.choice([720 * 480, 86400, 20011]))
        th = float(rng.uniform(0, 2 * np.pi))
        eye = (133.6 * scale * np.cos(th), float(rng.uniform(120, 320)) * scale, 133.6 * scale * np.sin(th))
        cam = P.camera_for(eye)
        pipe = api.StandardPipeline(ctx, w, h)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(cap, pool)
        gi, g = O.GI(cap, pool), O.GBuffer(w, h)
        threads = max(1, min(os.cpu_count() or 1, h // 4))
        cuts = [h * i // threads for i in range(threads + 1)]
        pix = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
        f = 0
        try:
            for f in range(1, int(rng.integers(3, 5))):
                rnd = synth.frame_rand(seed, f)
                pipe.render(scene, cam, sky, pix | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
                ts = [threading.Thread(target=P.render_oracle, args=(oscene, cam, sky, w, h, pix, n5[f % 8], rnd),
                                       kwargs={"rows": (cuts[i], cuts[i + 1]), "g": g}) for i in range(threads)]
                [t.start() for t in ts]
                [t.join() for t in ts]
                P.render_oracle(oscene, cam, sky, w, h, L.PASS_FINAL_GATHER | L.PASS_SURFEL, n5[f % 8], rnd, noise0=n0[f % 8], gi=gi, frame_index=f,
                                g=g, gi_threads=threads)
                res = P.compare_gbuffers(g, P.read_hip_gbuffer(pipe))
                P.assert_parity(res)
                assert res.get("illuminance_rel_l2", 0.0) <= 1e-3, res
                oh, op = gi.hash(), gi.pool()
                hh, hp = pipe.read_gi()
                assert np.array_equal(oh["fingerprint"], hh[:, 0]), "hash fingerprints"
                assert np.array_equal(oh["sample_count"], hh[:, 2] >> 16) and np.array_equal(oh["last_accessed_frame"], hh[:, 2] & 0xFFFF), "hash counts / stamps"
                assert np.array_equal(op["direction"], hp["direction"]), "surfel pool"
        except AssertionError as e:
            failed.append(seed)
            if verbose:
                print(f"full-GI seed {seed}: MISMATCH frame {f} ({w}x{h}, scale {scale}, hash {cap}, pool {pool}): {str(e)[:300]}", flush=True)
        del pipe, gi, g
    return failed


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    t0 = time.time()
    k0 = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if os.environ.get("STRESS_FULLGI") == "1":
        bad = run_fullgi(n, first)
    else:
        bad = run_deep(n, first, k0=k0) if os.environ.get("STRESS_DEEP") == "1" else run(n, first, big=os.environ.get("STRESS_BIG") == "1", k0=k0)
    print(f"{n} scenes, {len(bad)} with mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
