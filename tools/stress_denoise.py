#!/usr/bin/env python3
"""Randomised sweep of DUST_PASS_DENOISE against the oracle's scalar restatement (oracle/denoise.c): random frame sizes (not multiples of
any tile), random scenes, a drifting camera and moving instances, random filter settings; per frame the denoised and accumulation planes
within 1e-3 relative L2, accumulated frame counts equal, sky pixels untouched (what tests/test_gpu_denoise.py checks at one size).
usage: stress_denoise.py [n] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L, api, synth
from test_gpu_denoise import unpack, mat4, PASSES

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = api.Context(device=0)
n0, n5 = synth.stbn_scalar(layers=8), synth.stbn_unitvec3_cosine(layers=8)
sky = P.sky_state()
bad = []
compared = 0
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    desc = P.small_scene(seed=seed, n_models=int(rng.integers(1, 4)), n_instances=int(rng.integers(2, 8)), size=tuple(int(v) for v in rng.integers(16, 50, 3)))
    models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
    scene = api.Scene(ctx)
    xfs = []
    for mid, t in desc.instances:
        scene.add_instance(models[mid], t)
        xfs.append(np.array(t, np.float32).reshape(3, 4).copy())
    scene.commit()
    w, h = int(rng.integers(17, 220)), int(rng.integers(17, 140))
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(0, n0); pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 14, 2048)
    kw = dict(max_accumulated_frames=int(rng.choice([2, 6, 30])), antilag_power=float(rng.choice([0.0, 0.8])), max_blur_radius=float(rng.choice([0.0, 4.0, 15.0])))
    pipe.set_denoiser(**kw)
    orc = O.Denoiser(w, h, max_frames=kw["max_accumulated_frames"], power=kw["antilag_power"], radius=kw["max_blur_radius"])
    eye0 = rng.uniform(70, 130, 3)
    drift = rng.uniform(-2, 2, 3)
    mover = int(rng.integers(0, len(xfs)))
    try:
        for f in range(1, 6):
            eye = tuple(float(v) for v in eye0 + drift * f)
            cam = P.camera_for(eye, target=(0.0, 5.0, 0.0))
            cur = xfs[mover].copy(); cur[:, 3] += rng.uniform(-1.5, 1.5, 3).astype(np.float32)
            scene.set_transform(mover, cur.reshape(12), mat4(xfs[mover]))
            scene.commit()
            xfs[mover] = cur
            pipe.render(scene, cam, sky, PASSES, frame_index=f, rand=synth.frame_rand(seed, f))
            g = P.read_hip_gbuffer(pipe)
            acc = pipe.read_plane(L.PLANE_ACCUM)
            want_den, want_acc = orc.frame(g, cam, f)
            hit = np.isfinite(g["depth"])
            compared += int(hit.sum())
            if hit.any():
                a, b = unpack(g["denoised"])[hit].astype(np.float64), unpack(want_den)[hit].astype(np.float64)
                rel = float(np.sqrt(((a - b) ** 2).sum() / max(1e-30, (b ** 2).sum())))
                x, y = acc[hit][:, :3].astype(np.float64), want_acc[hit][:, :3].astype(np.float64)
                rel_acc = float(np.sqrt(((x - y) ** 2).sum() / max(1e-30, (y ** 2).sum())))
                assert rel <= 1e-3 and rel_acc <= 1e-3, f"frame {f}: rel {rel:.3g} acc {rel_acc:.3g}"
                assert np.abs(acc[hit][:, 3] - want_acc[hit][:, 3]).max() <= 1e-3, f"frame {f}: accumulated counts"
            assert np.array_equal(g["denoised"][~hit], want_den[~hit]), f"frame {f}: sky pixels changed"
    except AssertionError as e:
        bad.append(seed)
        print(f"seed {seed}: {w}x{h} {kw}: {str(e)[:200]}", flush=True)
print(f"{n} sequences, {len(bad)} with mismatches, {compared} hit pixels compared, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
