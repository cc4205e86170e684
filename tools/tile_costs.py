#!/usr/bin/env python3
"""Where a frame's time goes by tile: the per-tile cycle map of the fused primary + AO kernel on the bench scene, and what the most
expensive tiles have in common (GPU box). usage: tile_costs.py [n_top]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

n_top = int(sys.argv[1]) if len(sys.argv) > 1 else 12
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
for f in range(1, 6):
    pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, f, synth.frame_rand(1, f))
ctx.sync()
c = pipe.tile_costs(0).astype(np.float64)
ms = pipe.pass_stats(0).ms
waves = 4096
print(f"kernel {ms:.4f} ms; tiles {c.shape[1]} x {c.shape[0]}; cycles: median {np.median(c):.0f}  p90 {np.percentile(c, 90):.0f}  p99 {np.percentile(c, 99):.0f}  "
      f"max {c.max():.0f}; sum / {waves} waves = {c.sum() / waves:.0f}")
depth = pipe.read_plane(L.PLANE_DEPTH)
vid = pipe.read_plane(L.PLANE_VOXEL_ID)
order = np.argsort(-c.ravel())[:n_top]
for t in order:
    ty, tx = divmod(int(t), c.shape[1])
    d = depth[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8]
    ids = vid[ty * 8:ty * 8 + 8, tx * 8:tx * 8 + 8] & 0xFFFF
    hit = np.isfinite(d)
    print(f"  tile ({tx:3d},{ty:3d}) {c[ty, tx]:8.0f} cycles = {c[ty, tx] / np.median(c):4.1f} x median; hit {hit.mean():.2f}; "
          f"depth {d[hit].min() if hit.any() else 0:.1f}..{d[hit].max() if hit.any() else 0:.1f}; instances hit {len(np.unique(ids[hit]))}")
# coarse heat map (16 x 9 cells of 15 x 15 tiles)
hm = c[:135 // 15 * 15, :240 // 15 * 15].reshape(9, 15, 16, 15).mean(axis=(1, 3)) / np.median(c)
print("mean cost / median per 120 x 120 pixel cell:")
for row in hm:
    print("   " + " ".join(f"{v:4.1f}" for v in row))
