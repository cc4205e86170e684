#!/usr/bin/env python3
"""How the surfel pass's insert requests fall into probe-window clusters (the deterministic apply's unit of serial work).
usage (GPU box): cluster_stats.py [frames]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from dust_amd import scenes as P
from dust_amd import _lib as L, api, synth

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
W, H = 1920, 1080
ctx = api.Context(device=0)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
pipe = api.StandardPipeline(ctx, W, H)
pipe.set_noise(0, synth.stbn_scalar())
pipe.set_noise(5, synth.stbn_unitvec3_cosine())
eye = (122.0, 300.61, 54.45)
cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
for f in range(1, frames + 1):
    pipe.render(scene, cam, sky, passes, f, synth.frame_rand(1, f))
h, p = pipe.read_gi()
valid = p["direction"] < 6


def pcg(v):
    v = v.astype(np.uint64)
    state = (v * 747796405 + 2891336453) & 0xFFFFFFFF
    word = (((state >> ((state >> 28) + 4)) ^ state) * 277803737) & 0xFFFFFFFF
    return ((word >> 22) ^ word) & 0xFFFFFFFF


pos = (np.trunc(p["pos"][valid] / 4.0).astype(np.int64) & 0xFFFFFFFF).astype(np.uint64)
d = p["direction"][valid].astype(np.uint64)
hsh = pcg(pos[:, 0])
hsh = pcg((pos[:, 1] + hsh) & 0xFFFFFFFF)
hsh = pcg((pos[:, 2] + hsh) & 0xFFFFFFFF)
hsh = pcg((d + hsh) & 0xFFFFFFFF)
loc = np.sort(hsh % (32 * 1024 * 1024))
gap = np.diff(loc)
heads = np.concatenate([[True], gap > 2])
sizes = np.diff(np.flatnonzero(np.concatenate([heads, [True]])))
runs = np.diff(np.flatnonzero(np.concatenate([[True], gap != 0, [True]])))
print("valid surfels", int(valid.sum()), "distinct locations", int((gap != 0).sum() + 1), "clusters", len(sizes))
print("cluster size: mean %.2f  p50 %d  p99 %d  max %d;  sum of squares %.3g" % (sizes.mean(), np.percentile(sizes, 50), np.percentile(sizes, 99), sizes.max(), float((sizes.astype(float) ** 2).sum())))
print("same-location run: mean %.2f  p99 %d  max %d" % (runs.mean(), np.percentile(runs, 99), runs.max()))
multi = []
at = 0
for sz in sizes.tolist():
    if loc[at + sz - 1] != loc[at]:
        multi.append((sz, len(np.unique(loc[at:at + sz]))))
    at += sz
print("clusters spanning more than one location:", len(multi), "largest (requests, locations):", sorted(multi, reverse=True)[:8])
