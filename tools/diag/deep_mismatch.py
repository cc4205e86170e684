#!/usr/bin/env python3
"""Diagnose one scene of tools/stress_parity.py's deep sweep: which pixels' gather rays and which pool / hash entries differ.
usage: deep_mismatch.py seed k"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L, api, synth

seed, k = int(sys.argv[1]), int(sys.argv[2])
check_log = sys.argv[sys.argv.index("--check-log") + 1] if "--check-log" in sys.argv else None
ctx = None if check_log else api.Context(device=0)
n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
sky = P.sky_state()
passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
rng = np.random.default_rng(seed)
half = int(rng.integers(2, 24)); c0 = int(rng.integers(half, 256 - half)); fill = float(rng.choice([0.02, 0.1, 0.4]))
n_cells = max(4, int(fill * (2 * half) ** 3))
blocks, mats, pal = P.clustered_deep_model(seed=seed, n_cells=min(n_cells, 30000), cell_lo=c0 - half, cell_hi=c0 + half,
                                           max_bricks=int(rng.choice([2, 6, 12, 40])))
centre = 16.0 * c0
xf = np.eye(3, 4, dtype=np.float32); xf[:, 3] = -centre
oscene = O.Scene()
oscene.add_model(blocks, mats, pal, extent=4096)
oscene.add_instance(0, xf.reshape(12))
if ctx:
    model = api.Model(ctx, blocks, mats, pal, tree_extent_log2=12)
    scene = api.Scene(ctx)
    scene.add_instance(model, xf.reshape(12))
    keep = [model]
if k % 3 == 0:
    small = P.small_scene(seed=seed, n_models=1, n_instances=2, size=(40, 40, 40))
    oscene.add_model(small.models[0][0], small.models[0][1], pal)
    if ctx:
        m2 = api.Model(ctx, small.models[0][0], small.models[0][1], pal); keep.append(m2)
    for _, t in small.instances:
        if ctx: scene.add_instance(m2, t)
        oscene.add_instance(1, t)
if ctx: scene.commit()
oscene.commit()
if check_log:   # SF lines of a -DDUST_SURFEL_DEBUG build: every surfel ray and what the kernel found, against the oracle's brute force
    seen = set()
    n = bad = 0
    for line in open(check_log):
        if not line.startswith("SF ") or line in seen:
            continue
        seen.add(line)
        f = line.split()
        try:   # (device printf and this script's own prints share the pipe: a torn line is skipped)
            assert len(f) == 13
            i, sun = int(f[1]), int(f[2])
            o, d = [np.float32(v) for v in f[3:6]], [np.float32(v) for v in f[6:9]]
            found, t, inst, block = int(f[9]), np.float32(f[10]), int(f[11]), int(f[12])
        except (ValueError, AssertionError):
            continue
        want = oscene.trace(O.ORC_MODE_BRUTE, 3, sun, o, d, 0.1, 10000.0)
        n += 1
        same = (want is None) == (not found) and (sun or want is None or (np.float32(want[0]) == t and want[1] == inst))
        if not same:
            bad += 1
            print("surfel", i, "sun" if sun else "cosine", "o", o, "d", d, "kernel:", (found, t, inst, block), "oracle brute:", want,
                  "oracle hier:", oscene.trace(O.ORC_MODE_HIER, 3, sun, o, d, 0.1, 10000.0))
    print(n, "distinct surfel rays checked,", bad, "differ")
    sys.exit(0)
reach = 16.0 * half
eye = rng.uniform(-1.5 * reach, 1.5 * reach, 3)
if k % 4 == 0: eye = np.round(eye / 16.0) * 16.0
if abs(eye[0]) + abs(eye[2]) < 1e-3: eye[0] = 3.0
cam = P.camera_for(tuple(float(v) for v in eye), target=tuple(float(v) for v in rng.uniform(-0.3 * reach, 0.3 * reach, 3)))
w, h = int(rng.integers(40, 140)), int(rng.integers(24, 90))
cap, pool = int(rng.choice([509, 4093, 1 << 14])), int(rng.choice([97, 777, 2048]))
print("scene", len(blocks), "bricks, centre", centre, "half", half, "eye", eye, "frame", w, h, "cap", cap, "pool", pool, "small", k % 3 == 0)
pipe = api.StandardPipeline(ctx, w, h)
pipe.set_noise(0, n0); pipe.set_noise(5, n5); pipe.configure_gi(cap, pool)
gi = O.GI(cap, pool)
for f in range(1, 3):
    rnd = synth.frame_rand(seed, f)
    pipe.render(scene, cam, sky, passes | L.PASS_GI_ORDERED, frame_index=f, rand=rnd)
    g = P.render_oracle(oscene, cam, sky, w, h, passes, n5[f % 4], rnd, noise0=n0[f % 4], gi=gi, frame_index=f)
    hip = P.read_hip_gbuffer(pipe)
    io, ih = g.illuminance.view(np.uint16).reshape(h, w, 4), hip["illuminance"].view(np.uint16).reshape(h, w, 4)
    d = np.argwhere(io[..., 3] != ih[..., 3])
    print(f"frame {f}: pixels whose stored hit distance differs: {len(d)}")
    for y, x in d[:10]:
        print("   px", x, y, "oracle", P.half_to_float(io[y, x]), "hip", P.half_to_float(ih[y, x]), "depth", g.depth[y, x])
    oh, op = gi.hash(), gi.pool()
    hh, hp = pipe.read_gi()
    bad = np.nonzero(oh["fingerprint"] != hh[:, 0])[0]
    print(f"   hash entries with different fingerprints: {len(bad)}", bad[:10], [(hex(int(oh['fingerprint'][i])), hex(int(hh[i, 0]))) for i in bad[:5]])
    badc = np.nonzero(oh["sample_count"] != (hh[:, 2] >> 16))[0]
    print(f"   hash entries with different counts: {len(badc)}", badc[:10])
    bp = np.nonzero((op["direction"] != hp["direction"]))[0]
    print(f"   pool slots with different directions: {len(bp)}", bp[:10], [(op[i], hp[i]) for i in bp[:4]])
    bpos = np.nonzero((op["x"] != hp["x"]) | (op["y"] != hp["y"]) | (op["z"] != hp["z"]))[0] if "x" in op.dtype.names else []
    print(f"   pool slots with different positions: {len(bpos)}", bpos[:10], [(op[i], hp[i]) for i in bpos[:4]])
