#!/usr/bin/env python3
"""Randomised sweep of auto exposure + tone map against the oracle: random frame sizes, scenes, all nine transfer functions, several
frames of adaptation (what tests/test_tone_map.py checks at one size).      usage: stress_tonemap.py [n] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util as P
from dust_amd import _lib as L, api, synth
from test_tone_map import oracle_tone_map

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
ctx = api.Context(device=0)
n5 = synth.stbn_unitvec3_cosine(layers=4)
sky = P.sky_state()
conv = api.color_space_conversion()
bad = []
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    desc = P.small_scene(seed=seed, n_models=int(rng.integers(1, 4)), n_instances=int(rng.integers(1, 8)))
    scene = P.hip_scene(ctx, desc)
    w, h = int(rng.integers(3, 260)), int(rng.integers(3, 170))
    pipe = api.StandardPipeline(ctx, w, h)
    pipe.set_noise(5, n5)
    cam = P.camera_for(tuple(float(v) for v in rng.uniform(60, 140, 3)))
    tf = int(rng.integers(0, 9))
    avg = 0.0
    try:
        for f in range(1, 4):
            pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_ACCUMULATE, frame_index=f, rand=synth.frame_rand(seed, f))
            pipe.tone_map(transfer_function=tf, conversion=conv)
            den, alb = pipe.read_plane(L.PLANE_DENOISED), pipe.read_plane(L.PLANE_ALBEDO)
            counts, avg, out = oracle_tone_map(den, alb, avg, conv, tf)
            got_avg = pipe.exposure()
            assert np.isclose(got_avg, avg, rtol=2e-3 + 3.0 / (w * h)), f"frame {f}: exposure {got_avg} vs {avg}"
            a = out.view(np.float16).astype(np.float32)
            b = pipe.read_plane(L.PLANE_OUTPUT).view(np.float16).astype(np.float32)
            fin = np.isfinite(a) & np.isfinite(b)
            assert (np.isfinite(a) != np.isfinite(b)).mean() < 2e-3 + 2.0 / (w * h), f"frame {f}: finiteness"
            if fin.any() and (a[fin] ** 2).sum() > 0:
                rel = np.sqrt(((a[fin] - b[fin]) ** 2).sum()) / np.sqrt((a[fin] ** 2).sum())
                assert rel <= 1e-3, f"frame {f}: rel {rel:.3g}"
            avg = got_avg
    except AssertionError as e:
        bad.append(seed)
        print(f"seed {seed}: {w}x{h} tf {tf}: {str(e)[:200]}", flush=True)
print(f"{n} cases, {len(bad)} with mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
