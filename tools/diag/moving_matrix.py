"""Which part of a moving view costs kernel time: the camera that changes, or the scene commits? (GPU box)
camera: still / sway; scene: static / teapot moved + commit every frame. Per-frame timing (no sparse sampling)."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import scenes as P, _lib as L, api, synth
sparse = os.environ.get("SPARSE", "0") == "1"
ctx = api.Context(device=0, timing=True, sparse_timing=sparse)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
base = P.hip_scene(ctx, desc)
tdesc = P.SceneDesc.from_vox(synth.teapot_scene(96))
tea_model = api.Model(ctx, tdesc.models[0][0], tdesc.models[0][1], tdesc.palette)
scene = api.Scene(ctx)
for model, (_, t) in zip(base._models, desc.instances):
    scene.add_instance(model, t)
home = np.asarray(tdesc.instances[0][1], np.float32).reshape(3, 4)
def tea_xf(t):
    m = home.copy(); m[:, 3] += np.array([math.sin(t) * 50.0, 200.0, 0.0], np.float32); return np.ascontiguousarray(m.reshape(12))
tea = scene.add_instance(tea_model, tea_xf(0.0))
scene.commit()
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
sky = api.sky_struct(P.sky_state())
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
eye0 = (122.0, 300.61, 54.45)
radius, th0 = math.hypot(eye0[0], eye0[2]), math.atan2(eye0[2], eye0[0])
def cam_at(th):
    eye = (radius * math.cos(th), eye0[1], radius * math.sin(th))
    return api.make_camera(eye, api.look_at_rotation(eye, (0.0, 0.0, 0.0)), api.PinholeProjection())
f = [1]
def run(label, sway, commits, n=240, warm=96):
    cams = [cam_at(th0 + (0.15 * math.sin(2 * math.pi * (k / 60.0) / 4.0) if sway else 0.0)) for k in range(n + warm)]
    for phase, lo, hi in (("warm", 0, warm), ("timed", warm, warm + n)):
        if phase == "timed":
            ctx.sync(); pipe.mark_kernel_times(); t0 = time.perf_counter()
        for k in range(lo, hi):
            if commits == 1:
                scene.set_transform(tea, tea_xf(k / 60.0)); scene.commit()
            elif commits == 2:   # a commit that changes nothing: the same transform again
                scene.set_transform(tea, tea_xf(0.0)); scene.commit()
            pipe.render(scene, cams[k], sky, passes, f[0], synth.frame_rand(1, f[0])); f[0] += 1
    ctx.sync(); dt = time.perf_counter() - t0
    ms, nn = pipe.kernel_times(mark=True)
    print(f"{label:44s} {dt / n * 1e3:.4f} ms/step   kernel {ms[0] / max(nn[0], 1):.4f} ms over {nn[0]} timed launches", flush=True)
run("still camera, static scene", False, 0)
run("still camera, static scene (again)", False, 0)
run("still camera, same transform + commit", False, 2)
run("still camera, teapot moved + commit", False, 1)
run("swaying camera, static scene", True, 0)
run("swaying camera, teapot moved + commit", True, 1)
run("still camera, static scene (again)", False, 0)
