"""Tile WORK (a -DDUST_TILE_WORK build: wave-level walk trips + visits) against tile CYCLES (the shipped build), same still view of the
castle: writes both maps under gpurun_out/ for a regression on the CPU. (GPU box; DUST_HIP_STILL_REFRESH_MAX=1 so that every launch measures)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import numpy as np
import parity_util as P
from dust_amd import _lib as L, api, synth
tag = sys.argv[1]
ctx = api.Context(device=0)
data, _ = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
cam, sky = P.camera_for((122.0, 300.61, 54.45)), P.sky_state()
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine(layers=4))
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
for f in range(1, 200):
    pipe.render(scene, cam, sky, passes, f, 7)
maps = []
for f in range(200, 232):
    pipe.render(scene, cam, sky, passes, f, 7)
    ctx.sync()
    maps.append(pipe.tile_costs(0).copy())
np.save(f"gpurun_out/tilemap_{tag}.npy", np.stack(maps))
print(tag, "mean", np.stack(maps).mean(), "first-vs-second corr", np.corrcoef(maps[0].ravel(), maps[1].ravel())[0, 1])
