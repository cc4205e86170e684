#!/usr/bin/env python3
"""Times the four passes of a GI frame on the bench scene (GPU box). usage: gi_timing.py [frames] [ordered]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ordered = len(sys.argv) > 2 and sys.argv[2] == "ordered"
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(0, synth.stbn_scalar())
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_ACCUMULATE | (L.PASS_GI_ORDERED if ordered else 0)
for f in range(1, frames + 1):
    t0 = time.perf_counter()
    pipe.render(scene, cam, sky, passes | (L.PASS_COUNT_STATS if f == frames else 0), f, synth.frame_rand(1, f))
    ctx.sync()
    dt = (time.perf_counter() - t0) * 1e3
    ms = [pipe.pass_stats(i).ms for i in (0, 1, 3, 4)]
    print(f"frame {f}: wall {dt:.2f} ms  primary {ms[0]:.3f}  ao {ms[1]:.3f}  final_gather {ms[2]:.3f}  surfel {ms[3]:.3f}")
for i, n in enumerate(("primary", "sun", "ao", "final_gather", "surfel_sun", "surfel_cos")):
    s = pipe.pass_stats(i)
    r = max(1, s.rays)
    print(f"{n:12s} rays {s.rays:9d} hits {s.hits / r:.3f} inst/ray {s.instances_tested / r:.2f} bricks/ray {s.bricks_tested / r:.2f}")
h, p = pipe.read_gi()
print("hash entries used", int((h[:, 0] != 0).sum()), "valid surfels", int((p["direction"] < 6).sum()))
