#!/usr/bin/env python3
"""Kernel times of the bench scene without the bench's bookkeeping (GPU box): castle stand-in, 1920x1080, primary + AO (or
--gi: all four passes). The DUST_HIP_* switches in the environment apply. usage: quick_time.py [--gi] [--frames N] [--label X]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

gi = "--gi" in sys.argv
frames = int(sys.argv[sys.argv.index("--frames") + 1]) if "--frames" in sys.argv else 60
label = sys.argv[sys.argv.index("--label") + 1] if "--label" in sys.argv else ""
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
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
if gi:
    passes |= L.PASS_FINAL_GATHER | L.PASS_SURFEL
for f in range(1, 41):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
ctx.sync()
pipe.kernel_times(mark=True)
for f in range(41, 41 + frames):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
ctx.sync()
ms, n = pipe.kernel_times(mark=True)
names = ("primary(+ao)", "ao", "gather", "surfel")
print(label or os.environ.get("DUST_HIP_LIB", "default"), " ".join(f"{names[k]} {ms[k] / n[k]:.4f}" for k in range(4) if n[k]), flush=True)
try:
    import ctypes
    lib = L.load()
    buf = (ctypes.c_ulonglong * 16)()
    if lib.dust_hip_pool_stats(buf) == 0:
        names = ["fetch", "scan", "pop", "walk", "done"]
        print("  pool phases (trips, mean lanes): " + ", ".join(f"{names[i]} {buf[2*i]/frames/1e3:.0f}k x {buf[2*i+1]/max(1,buf[2*i]):.1f}" for i in range(5)))
except AttributeError:
    pass
