#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: procedural 4096^3 model (hierarchy (4,4,2,2)) at the given brick occupancy,
1920x1080, primary + AO and the full GI frame. usage: deep_tree_timing.py [occupancy=0.01]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
from dust_amd import scenes as P  # noqa: E402
from dust_amd import _lib as L, api, synth  # noqa: E402

occ = float(sys.argv[1]) if len(sys.argv) > 1 else 0.01
t0 = time.time()
blocks, mats = synth.procedural_deep_blocks(occupancy=occ, sample=True)
print(f"{len(blocks)} bricks, {len(mats)} voxels generated in {time.time() - t0:.1f} s", flush=True)
ctx = api.Context(device=0, timing=True)
t0 = time.time()
model = api.Model(ctx, blocks, mats, synth.make_palette(5), tree_extent_log2=12)
scene = api.Scene(ctx)
xf = np.eye(3, 4, dtype=np.float32)
xf[:, 3] = (-2048.0, -2048.0, -2048.0)
scene.add_instance(model, xf.reshape(12))
scene.commit()
print(f"hierarchy built and uploaded in {time.time() - t0:.1f} s", flush=True)
W, H = 1920, 1080
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(0, synth.stbn_scalar())
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
sky = P.sky_state()
for name, eye in (("outside", (2600.0, 1900.0, 2300.0)), ("inside", (300.0, 200.0, -150.0))):
    cam = P.camera_for(eye)
    pa = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    full = pa | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_ACCUMULATE
    for f in range(1, 6):
        pipe.render(scene, cam, sky, full | (L.PASS_COUNT_STATS if f == 5 else 0), f, synth.frame_rand(1, f))
    ctx.sync()
    st = [pipe.pass_stats(i) for i in range(6)]
    rays = sum(s.rays for s in st)
    for f in range(6, 12):
        pipe.render(scene, cam, sky, full, f, synth.frame_rand(1, f))
    ctx.sync()
    ms = [pipe.pass_stats(i).ms for i in (0, 3, 4)]
    print(f"{name}: primary+AO {ms[0]:.3f} ms, final gather {ms[1]:.3f} ms, surfel {ms[2]:.3f} ms; {rays / 1e6:.2f} M rays/frame -> "
          f"{rays / (sum(ms) * 1e-3) / 1e9:.2f} Grays/s; primary+AO alone {sum(s.rays for s in st[:3]) / (ms[0] * 1e-3) / 1e9:.2f} Grays/s; "
          f"hit fraction {st[0].hits / max(1, st[0].rays):.2f}, bricks/ray {st[0].bricks_tested / max(1, st[0].rays):.2f}, "
          f"upper descents/ray {st[0].upper_descents / max(1, st[0].rays):.2f}")
