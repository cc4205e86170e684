"""Kernel time of the fused kernel under a library variant / debug bits, through the API (no bench self-checks)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import parity_util as P
from dust_amd import _lib as L, api, synth
ctx = api.Context(device=0)
data, _ = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
cam, sky = P.camera_for((122.0, 300.61, 54.45)), P.sky_state()
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine(layers=4))
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
for f in range(1, 400):
    pipe.render(scene, cam, sky, passes, f, 7)
ctx.sync()
pipe.kernel_times(mark=True)
for f in range(400, 600):
    pipe.render(scene, cam, sky, passes, f, 7)
ctx.sync()
ms, n = pipe.kernel_times()
print(os.environ.get("DUST_HIP_LIB", "default").split("/")[-1], os.environ.get("DUST_HIP_DEBUG", ""), "kernel ms", round(ms[0] / max(1, n[0]), 4), "launches timed", n[0])
