#!/usr/bin/env python3
"""Prints the counting build's per-ray traversal statistics for the bench scene (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene(scale=scale)
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0 * scale, 300.61 * scale, 54.45 * scale)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
pipe.render(scene, cam, P.sky_state(), L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_COUNT_STATS, 1, synth.frame_rand(1, 1))
ctx.sync()
for i, n in enumerate(("primary", "sun", "ao")):
    s = pipe.pass_stats(i)
    r = max(1, s.rays)
    print(f"{n:8s} rays {s.rays:9d} hits {s.hits / r:.3f} inst/ray {s.instances_tested / r:.2f} upper/ray {s.upper_descents / r:.2f} "
          f"mid/ray {s.mid_descents / r:.2f} bricks/ray {s.bricks_tested / r:.2f}  ms {s.ms:.3f}")
depth = pipe.read_plane(L.PLANE_DEPTH)
print("hit fraction", float(np.isfinite(depth).mean()), "depth range", float(depth[np.isfinite(depth)].min()), float(depth[np.isfinite(depth)].max()))
