"""Does a kernel of ANOTHER queue get onto the device while persistent traversal launches hold it? (GPU box)
The N-GPU bench leaves DUST_HIP_RESERVE_BLOCKS workgroup slots free so that RCCL's send / receive kernels can become resident next
to the band frames. One GPU cannot run those kernels, but it can run a stand-in: a small kernel (256 workgroups of 256 threads, a few
registers -- the shape of a copy kernel) enqueued on a second stream while frames are in flight, timed from enqueue to completion, with
and without reserved slots."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dust_amd import scenes as P, _lib as L, api, synth
ctx = api.Context(device=0, timing=True, sparse_timing=True)
data, _ = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, 1920, 1080)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
sky = api.sky_struct(P.sky_state())
cam = api.make_camera((122.0, 300.61, 54.45), api.look_at_rotation((122.0, 300.61, 54.45), (0, 0, 0)), api.PinholeProjection())
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
side = torch.cuda.Stream()
x = torch.zeros(256 * 256, device="cuda")
for f in range(1, 200):
    pipe.render(scene, cam, sky, passes, f, 7)
ctx.sync()
lat = []
for rep in range(40):
    for f in range(6):   # six frames in the queue: ~1.3 ms of persistent launches
        pipe.render(scene, cam, sky, passes, 1000 + rep * 8 + f, 7)
    time.sleep(0.0003)   # (the first of them is running now)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record()
        x.add_(1.0)
        e1.record()
    t0 = time.perf_counter()
    e1.synchronize()
    lat.append((time.perf_counter() - t0) * 1e3)
    ctx.sync()
lat.sort()
ms, n = pipe.kernel_times()
print(f"DUST_HIP_RESERVE_BLOCKS={os.environ.get('DUST_HIP_RESERVE_BLOCKS', '0'):>3}: side kernel done after (host clock, ms) min {lat[0]:.3f} median {lat[len(lat) // 2]:.3f} max {lat[-1]:.3f};"
      f" traversal kernel {ms[0] / max(1, n[0]):.4f} ms")
