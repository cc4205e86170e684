import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dust_amd import _lib as L, api, synth, scenes as S
ctx = api.Context(device=0)
desc_rng = np.random.default_rng(3)
pal = synth.make_palette(3)
solid = desc_rng.random((40, 40, 40)) < 0.1
x, y, z = np.nonzero(solid)
xyzi = np.stack([x, y, z, desc_rng.integers(0, 255, x.size)], axis=1).astype(np.uint8)
model = api.Model(ctx, *api.flatten_model(xyzi, (40, 40, 40), pal), pal)
scene = api.Scene(ctx)
for t in ((-20, -20, -20), (10, 0, 5)):
    m = np.eye(3, 4, dtype=np.float32); m[:, 3] = t
    scene.add_instance(model, m.reshape(12))
scene.commit()
n0, n5 = synth.stbn_scalar(layers=2), synth.stbn_unitvec3_cosine(layers=2)
sky = S.sky_state()
GI = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
base = S.camera_for((60.0, 50.0, 70.0))
def cam_with(**kw):
    c = L.Camera.from_buffer_copy(base)
    for k, v in kw.items():
        f = getattr(c, k)
        if hasattr(f, "__len__"):
            for i in range(len(f)): f[i] = v[i] if hasattr(v, "__len__") else v
        else:
            setattr(c, k, v)
    return c
print([f[0] for f in L.Camera._fields_])
cases = {"nan position": dict(position=float("nan")), "inf position": dict(position=float("inf")), "huge position": dict(position=1e30),
         "nan rotation": dict(view_col0=float("nan")), "zero rotation": dict(view_col0=0.0, view_col1=0.0, view_col2=0.0),
         "inf rotation": dict(view_col2=float("inf")), "zero fov": dict(tan_half_fov=0.0), "nan fov": dict(tan_half_fov=float("nan")),
         "near beyond far": dict(near_=1e4, far_=0.1), "negative near": dict(near_=-5.0), "huge fov": dict(tan_half_fov=1e30)}
for name, kw in cases.items():
    try:
        cam = cam_with(**kw)
    except Exception as e:
        print(name, "could not build:", e); continue
    pipe = api.StandardPipeline(ctx, 64, 32)
    pipe.set_noise(0, n0); pipe.set_noise(5, n5); pipe.configure_gi(4093, 777)
    t0 = time.time()
    try:
        for f in (1, 2):
            pipe.render(scene, cam, sky, GI, frame_index=f, rand=7 + f)
        ctx.sync()
        d = pipe.read_plane(L.PLANE_DEPTH)
        print(f"{name}: ok in {time.time() - t0:.3f} s, finite depth {np.isfinite(d).mean():.2f}", flush=True)
    except L.DustError as e:
        print(f"{name}: refused: {e}", flush=True)
# a good frame still renders afterwards
pipe = api.StandardPipeline(ctx, 64, 32); pipe.set_noise(5, n5)
pipe.render(scene, base, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, frame_index=1, rand=1); ctx.sync()
print("after all that, a normal frame: finite depth", np.isfinite(pipe.read_plane(L.PLANE_DEPTH)).mean())
# degenerate instance transforms beside the good ones: singular, NaN, infinite, enormous, tiny
for name, mat in {"zero matrix": np.zeros((3, 4)), "nan matrix": np.full((3, 4), np.nan), "inf translation": np.array([[1, 0, 0, np.inf], [0, 1, 0, 0], [0, 0, 1, 0]]),
                  "rank one": np.array([[1, 1, 1, 0], [1, 1, 1, 0], [1, 1, 1, 0]]), "scale 1e30": np.eye(3, 4) * 1e30, "scale 1e-30": np.eye(3, 4) * 1e-30}.items():
    s2 = api.Scene(ctx)
    m = np.eye(3, 4, dtype=np.float32); m[:, 3] = (-20, -20, -20)
    s2.add_instance(model, m.reshape(12))
    t0 = time.time()
    try:
        s2.add_instance(model, np.asarray(mat, np.float32).reshape(12))
        s2.commit()
        pipe = api.StandardPipeline(ctx, 64, 32)
        pipe.set_noise(0, n0); pipe.set_noise(5, n5); pipe.configure_gi(4093, 777)
        for f in (1, 2):
            pipe.render(s2, base, sky, GI, frame_index=f, rand=7 + f)
        ctx.sync()
        print(f"instance with {name}: ok in {time.time() - t0:.3f} s, finite depth {np.isfinite(pipe.read_plane(L.PLANE_DEPTH)).mean():.2f}", flush=True)
    except L.DustError as e:
        print(f"instance with {name}: refused: {e}", flush=True)
