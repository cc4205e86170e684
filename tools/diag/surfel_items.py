#!/usr/bin/env python3
"""Per-item cycle costs of the surfel trace (pass kind 3) on the bench scene: is the kernel as long as its work or as its longest items?
(GPU box) usage: surfel_items.py [frames [sun azimuth in degrees]]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 40
azimuth = float(sys.argv[2]) if len(sys.argv) > 2 else None   # degrees (the default sun stands at 180: x == 0)
os.environ.setdefault("DUST_HIP_NO_SIDE_STREAM", "1")
W, H = 1920, 1080
ctx = api.Context(device=0, timing=True)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(0, synth.stbn_scalar())
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
if azimuth is not None:   # the default sun turned about the vertical: same elevation, same baked coefficients
    h = float(np.hypot(sky[48], sky[50]))
    sky = sky.copy()
    sky[48], sky[50] = h * np.sin(np.deg2rad(azimuth)), h * np.cos(np.deg2rad(azimuth))
    print("sun", sky[48:51])
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_ACCUMULATE
for f in range(1, frames + 1):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
ctx.sync()
for kind, name in ((2, "final gather"), (3, "surfel trace")):
    c = pipe.tile_costs(kind).astype(np.float64).ravel()
    ms = pipe.pass_stats(4 if kind == 3 else 3).ms
    live = c[c > 0]
    clk = 2.3829e6  # cycles per ms
    print(f"{name}: pass {ms:.4f} ms; items {c.size} ({live.size} with a cost); cycles median {np.median(live):.0f} p90 {np.percentile(live, 90):.0f} "
          f"p99 {np.percentile(live, 99):.0f} max {live.max():.0f} = {live.max() / clk:.4f} ms; sum / 4096 waves = {c.sum() / 4096:.0f} = {c.sum() / 4096 / clk:.4f} ms")
    if kind == 3:
        half = c.size // 2
        print(f"   cosine items: sum {c[:half].sum() / 4096 / clk:.4f} ms/4096, max {c[:half].max() / clk:.4f} ms;  sun items: sum {c[half:].sum() / 4096 / clk:.4f}, max {c[half:].max() / clk:.4f}")
        top = np.sort(live)[::-1]
        print("   top items (ms):", " ".join(f"{v / clk:.3f}" for v in top[:16]))
        print("   share of the work in the top 1 % / 5 % / 20 % of items:", " / ".join(f"{top[:max(1, int(top.size * q))].sum() / top.sum():.2f}" for q in (0.01, 0.05, 0.2)))
