"""does a camera that changes by a hair every frame cost kernel time? (GPU box)"""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import scenes as P, _lib as L, api, synth
ctx = api.Context(device=0, timing=True, sparse_timing=True)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
sky = api.sky_struct(P.sky_state())
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
eye0 = (122.0, 300.61, 54.45)
def cam_eye(eye): return api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
f = 1
def run(label, cams, n=300, warm=150):
    global f
    for k in range(warm):
        pipe.render(scene, cams[k % len(cams)], sky, passes, f, synth.frame_rand(1, f)); f += 1
    ctx.sync(); pipe.mark_kernel_times()
    t0 = time.perf_counter()
    for k in range(n):
        pipe.render(scene, cams[k % len(cams)], sky, passes, f, synth.frame_rand(1, f)); f += 1
    ctx.sync(); dt = time.perf_counter() - t0
    ms, nn = pipe.kernel_times(mark=True)
    print(f"{label:50s} {dt / n * 1e3:.4f} ms/step, kernel {ms[0] / max(nn[0], 1):.4f}", flush=True)
A = cam_eye(eye0)
run("one camera", [A])
B = cam_eye((eye0[0] + 1e-3, eye0[1], eye0[2]))
run("A, copy of A alternating", [A, cam_eye(eye0)])
run("A B alternating", [A, B])
run("A A B cycling", [A, A, B])
run("A B B B cycling", [A, B, B, B])
run("B only", [B])
