#!/usr/bin/env python3
"""A view that moves, K frames per launch (GPU box): the eye sways along its circle round the castle as in bench.py's curves.moving, the scene
stands still (dust_hip_render_frames renders ONE scene state per call). K = 1: dust_hip_render_frame per frame; K > 1: K consecutive cameras
per call into K pipelines. Prints ms per frame for the moving stretch and for three of its cameras standing still.
MOVES=1 in the environment: one instance (the last one) also swings, its transform handed over with every frame (dust_hip_render_frames' moves;
K = 1: set_transform + commit + render_frame).
usage: moving_batch.py [K ...]"""
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from dust_amd import scenes as P   # noqa: E402
from dust_amd import _lib as L, api, synth   # noqa: E402

W, H = 1920, 1080
Ks = [int(v) for v in sys.argv[1:]] or [1, 4]
ctx = api.Context(device=0)
data, info = synth.castle_scene()
desc = P.SceneDesc.from_vox(data)
scene = P.hip_scene(ctx, desc)
sky = api.sky_struct(P.sky_state())
n5 = synth.stbn_unitvec3_cosine()
eye0 = (122.0, 300.61, 54.45)
radius, th0 = math.hypot(eye0[0], eye0[2]), math.atan2(eye0[2], eye0[0])
swing, period, fps = 0.15, 4.0, 60.0
settle, steps = 48, 240


def cam_at(th):
    eye = (radius * math.cos(th), eye0[1], radius * math.sin(th))
    return api.make_camera(eye, api.look_at_rotation(eye, (0.0, 0.0, 0.0)), api.PinholeProjection())


cams = [cam_at(th0 + swing * math.sin(2.0 * math.pi * (k / fps) / period)) for k in range(settle + steps + 8)]
PAO = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
MOVES = os.environ.get("MOVES") == "1"
import numpy as np   # noqa: E402
tid = len(desc.instances) - 1
home = np.asarray(desc.instances[tid][1], np.float32).reshape(3, 4)


def xf_at(k):
    m = home.copy()
    m[:, 3] += np.array([math.sin(k / fps) * 50.0, 0.0, 0.0], np.float32)
    return np.ascontiguousarray(m.reshape(12))


for K in Ks:
    pipes = []
    for _ in range(K):
        p = api.StandardPipeline(ctx, W, H)
        p.set_noise(5, n5)
        pipes.append(p)

    def run(first, n, moving=True, still_cam=None):
        k = first
        while k < first + n:
            idx = [1 + k + j for j in range(K)]
            cs = [cams[k + j] if moving else still_cam for j in range(K)]
            if K == 1:
                if MOVES and moving:
                    scene.set_transform(tid, xf_at(k))
                    scene.commit()
                pipes[0].render(scene, cs[0], sky, PAO, frame_index=idx[0], rand=synth.frame_rand(1, idx[0]))
            else:
                mv = [[(tid, xf_at(k + j), None)] for j in range(K)] if (MOVES and moving) else None
                api.StandardPipeline.render_frames(pipes, scene, cs, sky, PAO, idx, [synth.frame_rand(1, v) for v in idx], moves=mv)
            k += K
    res = []
    for _ in range(3):
        run(0, settle)
        ctx.sync()
        t0 = time.perf_counter()
        run(settle, steps)
        ctx.sync()
        res.append((time.perf_counter() - t0) / steps * 1e3)
    still = []
    for c in (cams[settle], cams[settle + steps // 2], cams[settle + steps - 1]):
        run(0, 80, moving=False, still_cam=c)
        ctx.sync()
        t0 = time.perf_counter()
        run(0, 80, moving=False, still_cam=c)
        ctx.sync()
        still.append((time.perf_counter() - t0) / 80 * 1e3)
    m = sorted(res)[1]
    s = sum(still) / len(still)
    print(f"MOVES={int(MOVES)} NO_LDS_BOXES={os.environ.get('DUST_HIP_NO_LDS_BOXES', '0')} K={K}: moving {m:.4f} ms per frame (passes {' '.join(f'{v:.4f}' for v in res)}), the same views standing still {s:.4f} "
          f"({' '.join(f'{v:.4f}' for v in still)}), moving / still {m / s:.4f}", flush=True)
