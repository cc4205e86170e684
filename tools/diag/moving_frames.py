"""Per-frame times of bench.py's moving view (GPU box): which launches of the timed stretch are slow, and does the counting pass
in front of it (COUNT=1) or the sparse event bracket (SPARSE=1) have a part in it? Prints one line per frame."""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import scenes as P, _lib as L, api, synth
sparse = os.environ.get("SPARSE", "1") == "1"
count_first = os.environ.get("COUNT", "1") == "1"
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
settle, steps = 48, 20
n = settle + steps
cams = [cam_at(th0 + 0.15 * math.sin(2 * math.pi * (k / 60.0) / 4.0)) for k in range(n + 1)]
def frame(k, count=False):
    scene.set_transform(tea, tea_xf(k / 60.0)); scene.commit()
    pipe.render(scene, cams[k], sky, passes | (L.PASS_COUNT_STATS if count else 0), frame_index=1 + k, rand=synth.frame_rand(1, 1 + k))
if count_first:
    for k in range(settle, n):
        frame(k, count=True); ctx.sync(); pipe.pass_stats(0)
pipe.clear()
for rep in range(2):
    for k in range(settle):
        frame(k)
    ctx.sync(); pipe.mark_kernel_times()
    t0 = time.perf_counter()
    per = []
    for k in range(settle, n):
        t1 = time.perf_counter(); frame(k)
        if os.environ.get("SYNC_EACH", "0") == "1":
            ctx.sync()
        per.append((time.perf_counter() - t1) * 1e3)
    ctx.sync(); dt = time.perf_counter() - t0
    ms, nn = pipe.kernel_times(mark=True)
    print(f"rep {rep}: {dt / steps * 1e3:.4f} ms/step, kernel {ms[0] / max(nn[0], 1):.4f} ms over {nn[0]} launches; host ms per frame: " + " ".join(f"{v:.3f}" for v in per), flush=True)
