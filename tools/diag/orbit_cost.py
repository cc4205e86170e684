"""still-view kernel time at several points of bench.py's orbit, then the moving orbit with different refresh settings (GPU box)"""
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
r, th0 = math.hypot(eye0[0], eye0[2]), math.atan2(eye0[2], eye0[0])
def cam_at(th):
    eye = (r * math.cos(th), eye0[1], r * math.sin(th))
    return api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
f = 1
for dth in (0.0, 0.2):
    cam = cam_at(th0 + dth)
    for _ in range(150):
        pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f)); f += 1
    ctx.sync(); pipe.mark_kernel_times()
    t0 = time.perf_counter()
    for _ in range(100):
        pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f)); f += 1
    ctx.sync(); dt = time.perf_counter() - t0
    ms, n = pipe.kernel_times(mark=True)
    print(f"still at +{dth:.1f} rad: {dt * 10:.4f} ms/step, kernel {ms[0] / max(n[0], 1):.4f}")
for omega in (float(os.environ.get('OMEGA', '0.25')) / 60, 0.05 / 60, 0.0005 / 60):
    cams = [cam_at(th0 + omega * k) for k in range(400)]
    for k in range(100):
        pipe.render(scene, cams[k], sky, passes, f, synth.frame_rand(1, f)); f += 1
    ctx.sync(); pipe.mark_kernel_times()
    t0 = time.perf_counter()
    for k in range(100, 400):
        pipe.render(scene, cams[k], sky, passes, f, synth.frame_rand(1, f)); f += 1
    ctx.sync(); dt = time.perf_counter() - t0
    ms, n = pipe.kernel_times(mark=True)
    print(f"orbit {omega * 60:.2f} rad/s: {dt / 300 * 1e3:.4f} ms/step, kernel(+sort in the bracket) {ms[0] / max(n[0], 1):.4f}")
cam = cam_at(th0)
for _ in range(150):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f)); f += 1
ctx.sync(); pipe.mark_kernel_times()
t0 = time.perf_counter()
for _ in range(100):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f)); f += 1
ctx.sync(); dt = time.perf_counter() - t0
ms, n = pipe.kernel_times(mark=True)
print(f"still at +0.0 rad again: {dt * 10:.4f} ms/step, kernel {ms[0] / max(n[0], 1):.4f}")
