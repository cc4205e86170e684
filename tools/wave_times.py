#!/usr/bin/env python3
"""When the waves of the fused primary + AO kernel start, get past the staging barrier and run out of tiles (GPU box; needs the
-DDUST_WAVE_TIMES build: make -C dust_amd/csrc VARIANT=wt EXTRA=-DDUST_WAVE_TIMES). Also the shader clock the run had
(s_memtime ticks per 100 MHz wall tick). usage: DUST_HIP_LIB=dust_amd/libdust_hip_wt.so python tools/wave_times.py [frames] [--frames-per-launch K]
--frames-per-launch K: the timeline of a launch of K frames (dust_hip_render_frames / k_primary_ao_batch): ONE start, ONE staging and ONE tail for K frames."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

K = int(sys.argv[sys.argv.index("--frames-per-launch") + 1]) if "--frames-per-launch" in sys.argv else 1
frames = int(sys.argv[1]) if len(sys.argv) > 1 and not sys.argv[1].startswith("--") else 200
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene()
scene = P.hip_scene(ctx, P.SceneDesc.from_vox(data))
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
if K > 1:
    pipes = [pipe]
    for _ in range(K - 1):
        pipes.append(api.StandardPipeline(ctx, W, H))
        pipes[-1].set_noise(5, synth.stbn_unitvec3_cosine())
    for f in range(1, frames + 1, K):
        idx = [f + i for i in range(K)]
        api.StandardPipeline.render_frames(pipes, scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, idx, [synth.frame_rand(1, v) for v in idx])
else:
    for f in range(1, frames + 1):
        pipe.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, f, synth.frame_rand(1, f))
ctx.sync()
ms = pipe.pass_stats(0).ms
lib = L.load()
buf = np.zeros((8192, 12), np.uint64)
assert lib.dust_hip_wave_times(buf.ctypes.data_as(ctypes.c_void_p)) == 0
t = buf[buf[:, 2] > 0].astype(np.float64)
clk = (t[:, 2] - t[:, 0]).sum() / ((t[:, 4] - t[:, 3]).sum() / 100.0)  # shader ticks per microsecond
# the shader clock counters of the XCDs are not aligned with each other: starts and ends on the wall clock (10 ns), durations on the shader clock
start, end_w = t[:, 3] / 100.0, t[:, 4] / 100.0
t0, end = start.min(), end_w.max()
span = end - t0
print(f"{K} frame(s) per launch; kernel {ms:.4f} ms (HIP events); {len(t)} waves; shader clock {clk:.1f} MHz; first start -> last end {span:.1f} us")
q = lambda a: " ".join(f"{v:7.1f}" for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
print("us (min p10 p50 p90 p99 max):")
print("  wave start after the first     ", q(start - t0))
print("  staging (entry -> past barrier)", q((t[:, 1] - t[:, 0]) / clk))
print("  out of tiles after the first   ", q(end_w - t0))
print("  idle before the last wave ends ", q(end - end_w))
print(f"mean idle at the end {np.mean(end - end_w) / span:.2%} of the span, staging {np.mean(t[:, 1] - t[:, 0]) / clk / span:.2%}, late start {np.mean(start - t0) / span:.2%}; "
      f"tiles per wave min/mean/max {t[:, 5].min():.0f} {t[:, 5].mean():.2f} {t[:, 5].max():.0f}")
xcd = (np.arange(len(buf))[buf[:, 2] > 0] // 8) % 8
print("per XCD: last wave out (us after first start):", " ".join(f"{(end_w[xcd == x].max() - t0):.1f}" for x in range(8)))
if K > 1:
    sys.exit(0)   # (the last-tile records are the single-frame kernel's)
# the waves that end last: what were their last two tiles, and how long did those take
order = np.argsort(-end_w)[:16]
costs = pipe.tile_costs(0)
print("latest waves: end, [last tile (x,y) start dur us | cost map cycles], [previous tile ...]")
for w in order:
    lt, ls, pt, ps = int(t[w, 6]), t[w, 7] / 100.0, int(t[w, 8]), t[w, 9] / 100.0
    lx, ly, px_, py_ = (lt & 0xFFFFFFFF) // 8, (lt >> 32) // 8, (pt & 0xFFFFFFFF) // 8, (pt >> 32) // 8
    print(f"  {end_w[w] - t0:6.1f}  last ({lx:3d},{ly:3d}) start {ls - t0:6.1f} dur {end_w[w] - ls:6.1f} | {costs[ly, lx]:7d}   prev ({px_:3d},{py_:3d}) start {ps - t0:6.1f} dur {ls - ps:6.1f} | {costs[py_, px_]:7d}   tiles {int(t[w, 5])}")
last_start = t[:, 7] / 100.0 - t0
print("start of a wave's LAST tile, us (min p10 p50 p90 p99 max):", q(last_start), " its duration:", q(end_w - t[:, 7] / 100.0))
