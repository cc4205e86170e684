"""shader clock of each of the last 200 launches, still view then moving view (GPU box; DUST_HIP_LIB = the -DDUST_WAVE_TIMES build)"""
import sys, os, math, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from dust_amd import scenes as P, _lib as L, api, synth
ctx = api.Context(device=0, timing=False)
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
sky = api.sky_struct(P.sky_state())
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
eye0 = (122.0, 300.61, 54.45)
r, th0 = math.hypot(eye0[0], eye0[2]), math.atan2(eye0[2], eye0[0])
def cam_at(th):
    eye = (r * math.cos(th), eye0[1], r * math.sin(th))
    return api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
lib = L.load()
def run(label, cams):
    for k in range(400):
        pipe.render(scene, cams[k % len(cams)], sky, passes, k + 1, synth.frame_rand(1, k + 1))
    ctx.sync()
    buf = np.zeros((256, 3), np.uint64)
    assert lib.dust_hip_launch_clocks(buf.ctypes.data_as(ctypes.c_void_p)) == 0
    b = buf[np.argsort(buf[:, 2])].astype(np.float64)
    mhz = b[:, 0] / (b[:, 1] / 100.0)
    dur = b[:, 1] / 100.0
    print(label, "clock MHz: min %.0f p10 %.0f median %.0f max %.0f" % (mhz.min(), np.percentile(mhz, 10), np.median(mhz), mhz.max()))
    print("   wave 0 busy us, last 32 launches:", " ".join(f"{v:.0f}" for v in dur[-32:]))
    print("   clock MHz,      last 32 launches:", " ".join(f"{v:.0f}" for v in mhz[-32:]))
run("still ", [cam_at(th0)])
run("moving", [cam_at(th0 + 0.004 * k) for k in range(400)])
