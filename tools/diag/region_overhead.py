"""where the fixed cost of a timed region goes: 20 steps, host timestamps (GPU box)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, ctypes
from dust_amd import scenes as P, _lib as L, api, synth
torch.cuda.set_device(0)
stream = torch.cuda.Stream(device=0)
torch.cuda.set_stream(stream)
ctx = api.Context(device=0, timing=True, sparse_timing=True, stream=ctypes.c_void_p(stream.cuda_stream))
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = api.sky_struct(P.sky_state())
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
f = 1
def frame():
    global f
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f)); f += 1
def sync():
    while not stream.query(): pass
    torch.cuda.synchronize()
for _ in range(1300): frame()
sync()
for K in (20, 20, 100, 20):
    t0 = time.perf_counter()
    frame()
    t1 = time.perf_counter()
    for _ in range(K - 1): frame()
    t2 = time.perf_counter()
    while not stream.query(): pass
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    print(f"K={K}: first render call {1e6*(t1-t0):.0f} us, all enqueued after {1e6*(t2-t0):.0f} us, stream idle after {1e6*(t3-t0):.0f} us (= {1e3*(t3-t0)/K:.4f} ms/step), "
          f"torch.cuda.synchronize() +{1e6*(t4-t3):.0f} us -> {1e3*(t4-t0)/K:.4f} ms/step")
