#!/usr/bin/env python3
"""Randomised sweep of device-side voxel edits (GPU box): random models, random batches (recolour, add anywhere including the
tree's corners, remove subsets and whole bricks, repeated voxels within a batch, clear everything, refill); after every batch
the model's device arrays must equal a host rebuild of the same voxels byte for byte, and get_voxels must agree.
usage: stress_edits.py [n_models] [first_seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from dust_amd import api, synth
from test_gpu_edit import host_model


def run(n, seed0):
    ctx = api.Context(device=0)
    bad = []
    for seed in range(seed0, seed0 + n):
        rng = np.random.default_rng(seed)
        pal = synth.make_palette(seed)
        lo = rng.integers(0, 200, 3)
        hi = np.minimum(256, lo + rng.integers(8, 120, 3))
        vox = {}
        for c in rng.integers(lo, hi, (int(rng.integers(1, 6000)), 3)):
            vox[tuple(int(t) for t in c)] = int(rng.integers(0, 255))
        b0, m0 = host_model(vox, pal)
        model = api.Model(ctx, b0, m0, pal)
        try:
            for batch in range(int(rng.integers(2, 7))):
                kind = int(rng.integers(0, 7))
                keys = list(vox.keys())
                xyz, val = [], []
                if kind == 0 and keys:
                    for k in rng.choice(len(keys), min(len(keys), int(rng.integers(1, 800))), replace=False):
                        xyz.append(keys[k]); val.append(int(rng.integers(0, 255)))
                elif kind == 1:
                    for c in rng.integers(0, 256, (int(rng.integers(1, 2000)), 3)):
                        xyz.append(tuple(int(t) for t in c)); val.append(int(rng.integers(0, 255)))
                    xyz += [(0, 0, 0), (255, 255, 255), (255, 0, 128), (0, 255, 0)]; val += [1, 2, 3, 4]
                elif kind == 2 and keys:
                    for k in rng.choice(len(keys), min(len(keys), int(rng.integers(1, 3000))), replace=False):
                        xyz.append(keys[k]); val.append(-1)
                elif kind == 3:
                    base = tuple(int(t) for t in rng.integers(0, 256, 3))
                    for _ in range(int(rng.integers(2, 40))):
                        c = tuple(int(min(255, base[a] + int(rng.integers(0, 2)))) for a in range(3))
                        xyz.append(c); val.append(int(rng.integers(-1, 255)))
                elif kind == 4:
                    c0 = rng.integers(0, 240, 3)
                    for c in rng.integers(c0, c0 + 16, (int(rng.integers(100, 20000)), 3)):
                        xyz.append(tuple(int(t) for t in c)); val.append(int(rng.integers(-1, 255)))
                elif kind == 5:
                    for k in keys:
                        xyz.append(k); val.append(-1)
                else:   # a whole brick row cleared or filled
                    bx, by, bz = (int(t) & ~3 for t in rng.integers(0, 256, 3))
                    v = int(rng.integers(-1, 255))
                    for x in range(bx, min(256, bx + 12)):
                        for y in range(by, by + 4):
                            for z in range(bz, bz + 4):
                                xyz.append((x, y, z)); val.append(v)
                if not xyz:
                    continue
                for c, v in zip(xyz, val):
                    if v < 0:
                        vox.pop(tuple(c), None)
                    else:
                        vox[tuple(c)] = v
                model.set_voxels(np.array(xyz, np.uint32), np.array(val, np.int32))
                want_b, want_m = host_model(vox, pal)
                got_b, got_m = model.read()
                assert len(got_b) == len(want_b) and len(got_m) == len(want_m), f"batch {batch} kind {kind}: sizes {len(got_b)}/{len(want_b)} {len(got_m)}/{len(want_m)}"
                assert got_b.tobytes() == want_b.tobytes(), f"batch {batch} kind {kind}: Block records differ"
                assert got_m.tobytes() == want_m.tobytes(), f"batch {batch} kind {kind}: material stream differs"
                probe = np.array(list(vox.keys())[:100] + [tuple(int(t) for t in c) for c in rng.integers(0, 256, (50, 3))], np.uint32).reshape(-1, 3)
                assert model.get_voxels(probe).tolist() == [vox.get(tuple(int(t) for t in c), -1) for c in probe], f"batch {batch}: get_voxels"
        except AssertionError as e:
            bad.append(seed)
            print(f"seed {seed}: {str(e)[:300]}", flush=True)
    return bad


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    t0 = time.time()
    bad = run(n, first)
    print(f"{n} models, {len(bad)} with mismatches, {time.time() - t0:.0f} s")
    sys.exit(1 if bad else 0)
