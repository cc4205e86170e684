#!/usr/bin/env python3
"""What the row-band partition of ONE frame costs per rank (GPU box, one GPU): kernel time of every band of the bench frame for
N = 1, 2, 4, 8 ranks -- the max over a partition's bands bounds the strong-scaling step from below (gather and launch gaps on top).
usage: band_times.py [--gi]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth, sharding
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
for world in (1, 2, 4, 8):
    times = []
    for rank in range(world):
        per, rows, _ = sharding.band_layout(rank, world, H)
        pipe = api.StandardPipeline(ctx, W, H)
        pipe.set_noise(5, synth.stbn_unitvec3_cosine())
        for f in range(1, 41):
            pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f), rows=rows)
        ctx.sync(); pipe.kernel_times(mark=True)
        for f in range(41, 81):
            pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f), rows=rows)
        ctx.sync()
        ms, n = pipe.kernel_times(mark=True)
        times.append(ms[0] / n[0])
    print(f"N={world}: band kernel ms " + " ".join(f"{t:.4f}" for t in times) + f"  max {max(times):.4f}  -> at best {times and (0.0 + max(times)):.4f} ms per frame = {0.2303 / max(times):.2f}x of one GPU", flush=True)
