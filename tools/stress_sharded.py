#!/usr/bin/env python3
"""Randomised sweep of the multi-GPU GI choreography on ONE GPU (SURVEY 8e option i): random scenes, frame sizes whose height does
not divide by the band granularity, 2 to 9 "ranks" (pipelines) including ranks whose band is empty, tiny to large hash tables;
every rank's hash, surfel pool and band must equal the single-pipeline run bit for bit (parity_util.sharded_gi_vs_single_device).
usage: stress_sharded.py [n] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import parity_util as P
from dust_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50
first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
ctx = api.Context(device=0)
bad = []
t0 = time.time()
for seed in range(first, first + n):
    rng = np.random.default_rng(seed)
    desc = P.small_scene(seed=seed, n_models=int(rng.integers(1, 4)), n_instances=int(rng.integers(1, 9)), size=tuple(int(v) for v in rng.integers(12, 60, 3)))
    scene = P.hip_scene(ctx, desc)
    if seed % 4 == 0:   # a 4096^3 model beside them: the DEEP kernel variants under the same choreography
        blocks, mats, pal = P.clustered_deep_model(seed=seed, n_cells=400, cell_lo=124, cell_hi=132, max_bricks=6)
        deep = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
        xf = np.eye(3, 4, dtype=np.float32)
        xf[:, 3] = -2048.0
        scene.add_instance(deep, xf.reshape(12))
        scene.commit()
    eye = tuple(float(v) for v in rng.uniform(-90, 90, 3))
    if abs(eye[0]) + abs(eye[2]) < 1e-3:
        eye = (1.0, eye[1], eye[2])
    cam = P.camera_for(eye)
    w, h = int(rng.integers(40, 200)), int(rng.integers(9, 120))
    world = int(rng.choice([2, 3, 4, 5, 8, 9]))
    sizes = (int(rng.choice([61, 509, 4093, 1 << 14])), int(rng.choice([97, 777, 2048])))
    try:
        P.sharded_gi_vs_single_device(ctx, scene, cam, P.sky_state(), w, h, world, int(rng.integers(2, 5)), n0, n5, seed=seed, gi_sizes=sizes,
                                      shard_trace=seed % 2 == 1)   # (odd seeds: the surfel trace sharded over the ranks as well, round 6)
    except AssertionError as e:
        bad.append(seed)
        from dust_amd import sharding
        per = sharding.gi_band_rows(world, h)
        empty = sum(1 for r in range(world) if min(h, r * per) >= min(h, (r + 1) * per))
        print(f"seed {seed}: {w}x{h}, {world} ranks ({per} rows each, {empty} without rows), sizes {sizes}: {str(e)[:200]}", flush=True)
print(f"{n} cases, {len(bad)} with mismatches, {time.time() - t0:.0f} s")
if os.environ.get("STRESS_VERBOSE"):
    from dust_amd import sharding
    for seed in range(first, first + n):
        rng = np.random.default_rng(seed)
        rng.integers(1, 4); rng.integers(1, 9); rng.integers(12, 60, 3); rng.uniform(-90, 90, 3)
        w, h = int(rng.integers(40, 200)), int(rng.integers(9, 120)); world = int(rng.choice([2, 3, 4, 5, 8, 9]))
        per = sharding.gi_band_rows(world, h)
        empty = sum(1 for r in range(world) if min(h, r * per) >= min(h, (r + 1) * per))
        print(seed, "FAIL" if seed in bad else "ok", f"{w}x{h} world {world} per {per} empty {empty}")
sys.exit(1 if bad else 0)
