#!/usr/bin/env python3
"""GI frames of the bench scene with the native denoiser (DUST_PASS_DENOISE) instead of the N-frame mean; run it under
`rocprofv3 --kernel-trace --stats` for the two filter kernels' durations (GPU box). usage: denoise_timing.py [frames]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 24
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(0, synth.stbn_scalar())
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
gi = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
for passes, name in ((gi, "GI frame"), (gi | L.PASS_DENOISE, "GI frame + denoiser")):
    for f in range(1, 4):
        pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
    ctx.sync()
    t0 = time.perf_counter()
    for f in range(4, 4 + frames):
        pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
    ctx.sync()
    print(f"{name}: {(time.perf_counter() - t0) * 1e3 / frames:.4f} ms per frame (wall, {frames} frames)")
