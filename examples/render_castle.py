#!/usr/bin/env python3
"""Offline frame of the castle stand-in, the HIP counterpart of examples/castle.rs: load the .vox scene, spawn it,
render N samples with all four ray-tracing passes, accumulate, tone map, write a PNG. Needs a GPU.
usage: render_castle.py [out.png] [spp] [width] [height] [scale]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from PIL import Image

from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

out = sys.argv[1] if len(sys.argv) > 1 else "castle.png"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 8
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1920
H = int(sys.argv[4]) if len(sys.argv) > 4 else 1080
scale = float(sys.argv[5]) if len(sys.argv) > 5 else 1.0

ctx = api.Context(device=0)
data, info = synth.castle_scene(scale=scale)
desc = P.SceneDesc.from_vox(data)
teapot = P.SceneDesc.from_vox(synth.teapot_scene(96))          # examples/castle.rs:111-117 also loads teapot.vox
scene_models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
tea_model = api.Model(ctx, teapot.models[0][0], teapot.models[0][1], teapot.palette)
scene = api.Scene(ctx)
for mid, t in desc.instances:
    scene.add_instance(scene_models[mid], t)
tea = teapot.instances[0][1].reshape(3, 4).copy()
tea[:, 3] += np.array([0.0, 200.0 * scale, 0.0], np.float32)    # teapot_move_system at t = 0 (castle.rs:287-291)
scene.add_instance(tea_model, tea.reshape(12))
scene.commit()

pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(0, synth.stbn_scalar())
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0 * scale, 300.61 * scale, 54.45 * scale)            # castle.rs:126
cam = api.make_camera(eye, api.look_at_rotation(eye, (0.0, 0.0, 0.0)), api.PinholeProjection())
sky = P.sky_state("default")
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_ACCUMULATE
for f in range(1, spp + 1):
    pipe.render(scene, cam, sky, passes, frame_index=f, rand=synth.frame_rand(1, f))
    pipe.tone_map(transfer_function=1)                           # exposure adapts frame by frame like the reference
ctx.sync()
img = pipe.read_plane(L.PLANE_OUTPUT).view(np.float16).astype(np.float32)[..., :3]
img = np.nan_to_num(np.clip(img, 0.0, 1.0))
Image.fromarray((img * 255.0 + 0.5).astype(np.uint8)).save(out)
print("wrote", out, "avg luminance", pipe.exposure(), info)
