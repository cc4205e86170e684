#!/usr/bin/env python3
"""Traversal counts of the final gather on the bench scene (GPU box): what the counting build says a frame's gather rays did."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth
ctx = api.Context(device=0)
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(0, synth.stbn_scalar()); pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
for f in range(1, 6):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
pipe.render(scene, cam, sky, passes | L.PASS_COUNT_STATS, 6, synth.frame_rand(1, 6))
for i in (3, 4, 5):
    s = pipe.pass_stats(i)
    print(os.environ.get("DUST_HIP_NO_RAY_LANES", "lanes"), i, "rays", s.rays, "inst", s.instances_tested, "upper", s.upper_descents, "mid", s.mid_descents, "bricks", s.bricks_tested, "hits", s.hits)
