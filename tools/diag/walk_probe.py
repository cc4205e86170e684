"""Kernel time of the primary pass ALONE (camera rays: cull + trace + shading, no sun / AO rays) under a library build, through the API.
For the walk-only probe builds (kernels.hip, DUST_WALK_PROBE): make VARIANT=wp5 EXTRA="-DDUST_WALK_PROBE=1 -DDUST_WP_T=640 -DDUST_WP_W=5
-DDUST_MAX_BLOCK=1024u"; run with DUST_HIP_LIB=.../libdust_hip_wp5.so DUST_HIP_NO_FUSE=1 DUST_HIP_BLOCK=640. (GPU box)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
import parity_util as P
from dust_amd import _lib as L, api, synth
ctx = api.Context(device=0)
data, _ = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
cam, sky = P.camera_for((122.0, 300.61, 54.45)), P.sky_state()
W, H = (3840, 2160) if "--4k" in sys.argv else (1920, 1080)
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(5, synth.stbn_unitvec3_cosine(layers=4))
passes = L.PASS_PRIMARY | (L.PASS_AMBIENT_OCCLUSION if "--ao" in sys.argv else 0)
for f in range(1, 300):
    pipe.render(scene, cam, sky, passes, f, 7)
ctx.sync()
pipe.kernel_times(mark=True)
for f in range(300, 500):
    pipe.render(scene, cam, sky, passes, f, 7)
ctx.sync()
ms, n = pipe.kernel_times()
print(os.environ.get("DUST_HIP_LIB", "default").split("/")[-1], "block", os.environ.get("DUST_HIP_BLOCK", "512"), "x", os.environ.get("DUST_HIP_BLOCKS_PER_CU", "2"),
      "primary kernel ms", round(ms[0] / max(1, n[0]), 4), "ao", round(ms[1] / max(1, n[1]), 4), "launches timed", n[0], flush=True)
