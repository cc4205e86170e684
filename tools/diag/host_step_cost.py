"""host microseconds per call of the moving-view step (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import scenes as P, _lib as L, api, synth
ctx = api.Context(device=0, timing=False)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = api.sky_struct(P.sky_state())
xf = np.ascontiguousarray(desc.instances[5][1], np.float32)
prev = np.eye(4, dtype=np.float32).reshape(16)
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
for f in range(1, 50): pipe.render(scene, cam, sky, passes, f, 1)
ctx.sync()
N = 300
for name, fn in (("set_transform", lambda k: scene.set_transform(5, xf, prev)), ("commit (after set_transform)", None), ("render", lambda k: pipe.render(scene, cam, sky, passes, k + 1, k))):
    ctx.sync()
    t0 = time.perf_counter()
    if fn is None:
        for k in range(N):
            scene.set_transform(5, xf, prev); scene.commit()
    else:
        for k in range(N): fn(k)
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"{name:30s} host {1e6 * (t1 - t0) / N:7.1f} us per call; with the GPU drained {1e6 * (t2 - t0) / N:7.1f}")
