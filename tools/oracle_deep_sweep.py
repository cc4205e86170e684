#!/usr/bin/env python3
"""CPU only: the oracle against itself on 4096^3 models -- its hierarchical walk (what the GPU is compared with at sizes brute force
cannot reach) against its brute-force mode (the semantic definition: closest accepted hit over ALL bricks), all five passes for two
frames on small random clustered scenes.      usage: oracle_deep_sweep.py n_scenes"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L, api, synth
n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
bad = 0
t0 = time.time()
N = int(sys.argv[1])
for seed in range(500000, 500000 + N):
    rng = np.random.default_rng(seed)
    half = int(rng.integers(2, 8)); c0 = int(rng.integers(half, 256 - half))
    blocks, mats, pal = P.clustered_deep_model(seed=seed, n_cells=int(rng.integers(4, 600)), cell_lo=c0 - half, cell_hi=c0 + half, max_bricks=int(rng.choice([2, 6, 12, 40])))
    centre = 16.0 * c0
    xf = np.eye(3, 4, dtype=np.float32); xf[:, 3] = -centre
    oscene = O.Scene(); oscene.add_model(blocks, mats, pal, extent=4096); oscene.add_instance(0, xf.reshape(12)); oscene.commit()
    reach = 16.0 * half
    eye = rng.uniform(-1.5 * reach, 1.5 * reach, 3)
    if seed % 4 == 0: eye = np.round(eye / 16.0) * 16.0
    if abs(eye[0]) + abs(eye[2]) < 1e-3: eye[0] = 3.0
    cam = P.camera_for(tuple(float(v) for v in eye), target=tuple(float(v) for v in rng.uniform(-0.3 * reach, 0.3 * reach, 3)))
    w, h = int(rng.integers(24, 64)), int(rng.integers(16, 48))
    states = []
    for mode in (O.ORC_MODE_HIER, O.ORC_MODE_BRUTE):
        gi = O.GI(4093, 777)
        planes = []
        for f in range(1, 3):
            rnd = synth.frame_rand(seed, f)
            g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f, mode=mode)
            planes.append((g.depth.copy(), g.voxel_id.copy(), g.illuminance.copy()))
        states.append((planes, gi.hash().copy(), gi.pool().copy()))
    a, b = states
    same = all(np.array_equal(x[0].view(np.uint32), y[0].view(np.uint32)) and np.array_equal(x[1], y[1]) and np.array_equal(x[2], y[2]) for x, y in zip(a[0], b[0])) \
        and np.array_equal(a[1]["fingerprint"], b[1]["fingerprint"]) and np.array_equal(a[2]["direction"], b[2]["direction"])
    if not same:
        bad += 1; print("seed", seed, "HIER != BRUTE", flush=True)
print(N, "deep scenes, oracle hierarchical vs brute force:", bad, "differ,", round(time.time() - t0), "s")
