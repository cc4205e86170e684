"""tile cost maps of two cameras a hair apart, measured under the same order (GPU box)"""
import sys, os, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import scenes as P, _lib as L, api, synth
ctx = api.Context(device=0, timing=True)
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
sky = api.sky_struct(P.sky_state())
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
eye0 = (122.0, 300.61, 54.45)
def cam_eye(eye): return api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
A, B = cam_eye(eye0), cam_eye((eye0[0] + 1e-3, eye0[1], eye0[2]))
f = 1
def frames(cam, n):
    global f
    ms = []
    for _ in range(n):
        pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, 7)); f += 1   # (same noise every frame)
        ctx.sync(); ms.append(pipe.pass_stats(0).ms)
    return ms
frames(A, 40)
def corr(x, y): return np.corrcoef(x.ravel(), y.ravel())[0, 1]
for rep in range(2):
    frames(A, 4); a1 = pipe.tile_costs(0).astype(np.float64)
    frames(A, 4); a2 = pipe.tile_costs(0).astype(np.float64)
    frames(B, 4); b1 = pipe.tile_costs(0).astype(np.float64)
    frames(B, 4); b2 = pipe.tile_costs(0).astype(np.float64)
    print("corr A-A %.4f B-B %.4f A-B %.4f %.4f; >2x: A-A %d A-B %d" % (corr(a1, a2), corr(b1, b2), corr(a1, b1), corr(a2, b2), int((a2 > 2 * a1).sum()), int((b1 > 2 * a1).sum())))
    top = np.argsort(-a1.ravel())[:300]
    print("  of A's 300 most expensive tiles: in A's second map they cost %.2f of their first, in B's %.2f (median ratio)" % (np.median(a2.ravel()[top] / a1.ravel()[top]), np.median(b1.ravel()[top] / a1.ravel()[top])))
