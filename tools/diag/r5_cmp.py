"""stream vs packet GI on the small castle: which state differs after how many frames (round 5 diagnostic)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from dust_amd import scenes as P, _lib as L, api, synth

def states(env, frames, passes_extra=0):
    for k in ("DUST_HIP_PACKET_GI", "DUST_HIP_RAY_STREAM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    s = 0.15
    sky, cam = P.sky_state(), P.camera_for((122.0 * s, 300.61 * s, 54.45 * s))
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    pipe = api.StandardPipeline(ctx, 192, 104)
    pipe.set_noise(0, n0); pipe.set_noise(5, n5)
    pipe.configure_gi(1 << 14, 776)
    out = []
    for f in range(1, frames + 1):
        pipe.render(scene, cam, sky, passes, frame_index=f, rand=synth.frame_rand(7, f))
        h, sp = pipe.read_gi()
        out.append((h.copy(), sp.view(np.uint32).copy(), pipe.read_plane(L.PLANE_ILLUMINANCE).copy()))
    return out

extra = dict(kv.split("=") for kv in sys.argv[1:])
for k in extra: os.environ.pop(k, None)
b = states({"DUST_HIP_PACKET_GI": "1"}, 2)
os.environ.pop("DUST_HIP_PACKET_GI", None)
a = states(extra, 2)
print(extra)
for f, (x, y) in enumerate(zip(a, b), 1):
    print("frame", f, [int((np.asarray(p) != np.asarray(q)).sum()) for p, q in zip(x, y)], "of", [np.asarray(p).size for p in x])
    d = np.argwhere(np.asarray(x[2]).reshape(104, 192, -1) != np.asarray(y[2]).reshape(104, 192, -1))
    if len(d):
        print("  first differing pixels (y, x, c):", d[:6].tolist())
