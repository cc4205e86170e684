#!/usr/bin/env python3
"""Kernel times of the deep-tree stress scene (procedural 4096^3, 1 % occupancy), primary + AO and the GI passes (GPU box)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth
label = sys.argv[sys.argv.index("--label") + 1] if "--label" in sys.argv else ""
ctx = api.Context(device=0)
blocks, mats = synth.procedural_deep_blocks(occupancy=0.01, sample=True)
model = api.Model(ctx, blocks, mats, synth.make_palette(5), tree_extent_log2=12)
scene = api.Scene(ctx)
xf = np.eye(3, 4, dtype=np.float32); xf[:, 3] = (-2048.0, -2048.0, -2048.0)
scene.add_instance(model, xf.reshape(12)); scene.commit()
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(0, synth.stbn_scalar()); pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (300.0, 200.0, -150.0)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | (L.PASS_FINAL_GATHER | L.PASS_SURFEL if "--gi" in sys.argv else 0)
for f in range(1, 13):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
ctx.sync(); pipe.kernel_times(mark=True)
for f in range(13, 33):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
ctx.sync()
ms, n = pipe.kernel_times(mark=True)
print(label or os.environ.get("DUST_HIP_LIB", "default"), " ".join(f"{('primary+ao','ao','gather','surfel')[k]} {ms[k]/n[k]:.4f}" for k in range(4) if n[k]), flush=True)
