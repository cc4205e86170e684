#!/usr/bin/env python3
"""Where the traversal kernels' wave cycles go, by code section (s_memtime buckets compiled in with -DDUST_PROFILE).

  python tools/kernel_sections.py --build      # here (cross-compiles dust_amd/libdust_hip_prof.so, which travels with gpurun)
  gpurun -- 'python tools/kernel_sections.py [--deep]'  # on the GPU box (--deep: the 4096^3 stress tree instead of the castle)

  DUST_HIP_DEBUG=$(( (cycles / 1024) << 12 )) python tools/kernel_sections.py   # k_surfel_trace: only the work items that took at least
                                                        # `cycles` (of this instrumented build) stay in the buckets: what the longest items are made of

Buckets are INCLUSIVE wave cycles summed over all waves; the table prints exclusive shares. The timers themselves
cost ~10 instructions per mark, so treat the shares as relative, not as absolute kernel time.
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF_LIB = os.path.join(ROOT, "dust_amd", "libdust_hip_prof.so")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if "--build" in sys.argv:
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(ROOT, "dust_amd", "csrc"), "PROFILE=1"])
    print("built", PROF_LIB, "(delete it afterwards: it travels with every gpurun push)")
    sys.exit(0)

os.environ["DUST_HIP_LIB"] = PROF_LIB
from dust_amd import scenes as P  # noqa: E402
from dust_amd import _lib as L, api, synth  # noqa: E402

NAMES = ["total", "grab", "cull", "trace_ray", "instance", "find_brick", "brick_test", "screen", "advance", "stage_roots"]


def read(lib):
    buf = (ctypes.c_ulonglong * 24)()
    assert lib.dust_hip_profile_read(buf, 24) == 0
    return [int(x) for x in buf]


def report(title, b, ms):
    tot = b[0] or 1
    v = dict(zip(NAMES, b))
    excl = {
        "stage_roots": v["stage_roots"], "grab (work counters)": v["grab"], "cull (bundle vs instance boxes, sort)": v["cull"],
        "candidate loop (slab tests, early exit)": v["trace_ray"] - v["instance"],
        "instance setup: 1/d + bounds slab": b[20], "instance setup: first cell, tolerances": b[23],
        "loop control + lane wait in the walk": v["instance"] - v["find_brick"] - v["brick_test"] - v["screen"] - v["advance"] - b[20] - b[23],
        "find_brick (root/mid lookup, mask load)": v["find_brick"], "brick test (4^3 DDA)": v["brick_test"],
        "near-plane screen / neighbour queue": v["screen"], "advance (exit planes, cell step)": v["advance"],
        "primary shading + G-buffer stores": b[21], "AO pass ray setup (normal, noise, ranges)": b[22],
        "other ray setup, shading, stores": v["total"] - v["stage_roots"] - v["grab"] - v["cull"] - v["trace_ray"] - b[21] - b[22],
    }
    print(f"\n== {title}: {ms:.3f} ms (instrumented), {tot / 1e9:.2f} G wave-cycles")
    for k, x in excl.items():
        print(f"  {k:44s}{100.0 * x / tot:6.1f} %")
    nt = max(1, b[11])
    print(f"  per packet-trace (wave level): candidates {b[12] / nt:.1f}, candidate-loop iterations {b[13] / nt:.1f}, "
          f"instance visits {b[14] / nt:.2f}, traversal loop trips {b[15] / nt:.1f} ({b[15] / max(1, b[14]):.1f} per visit), "
          f"of which {100.0 * b[10] / max(1, b[15]):.1f} % call visit_neighbours")
    lt = max(1, b[16])
    print(f"  lane-level trips: {b[16] / nt / 64:.2f} per ray-slot; {100.0 * b[17] / lt:.0f} % test a brick, {100.0 * b[18] / lt:.0f} % cross an empty "
          f"4-cell, {100.0 * b[19] / lt:.0f} % an empty 16-cell (or larger)")


def report_stream(title, b, ms):
    """k_ray_walk (gi.hip): cycles by phase, and how many lanes each phase's trips served"""
    tot = b[0] or 1
    print(f"\n== {title}: {ms:.3f} ms (instrumented), {tot / 1e9:.2f} G wave-cycles")
    for name, x in (("stage (roots, grid, boxes, enter records -> LDS)", b[9]), ("walk steps", b[4]), ("top-level walk (grid steps, box tests)", b[2]),
                    ("instance set-up (walk_begin)", b[20]), ("fetch (chunk counter, ray load)", b[1]),
                    ("loop control, ballots, hit stores, idle at the end", b[0] - b[9] - b[4] - b[2] - b[20] - b[1])):
        print(f"  {name:52s}{100.0 * x / tot:6.1f} %")
    rays = max(1, b[12])
    trips = max(1, b[23])
    print(f"  rays {b[12]}; loop trips {b[23]} ({64.0 * b[23] / rays:.2f} per 64 rays)")
    for name, t, l in (("walk", b[15], b[16]), ("top-level", b[13], b[22]), ("set-up", b[14], b[21]), ("fetch", b[11], b[12])):
        print(f"  {name:10s} trips {t:10d} ({100.0 * t / trips:5.1f} % of loop trips), lanes per trip {l / max(1, t):5.1f} ({100.0 * l / max(1, t) / 64.0:4.1f} % lane activity), per ray {l / rays:.2f}")
    print(f"  top-level work per ray: {b[18] / rays:.2f} grid steps, {b[17] / rays:.2f} box tests")


def main():
    lib = L.load()
    lib.dust_hip_profile_read.restype = ctypes.c_int
    lib.dust_hip_profile_read.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
    W, H = 1920, 1080
    ctx = api.Context(device=0)
    if "--deep" in sys.argv:  # bench.py --workload deep: the 4096^3 stress tree at 1 % brick occupancy, seen from inside
        import numpy as np
        blocks, mats = synth.procedural_deep_blocks(occupancy=0.01, sample=True)
        model = api.Model(ctx, blocks, mats, synth.make_palette(5), tree_extent_log2=12)
        scene = api.Scene(ctx)
        xf = np.eye(3, 4, dtype=np.float32)
        xf[:, 3] = (-2048.0, -2048.0, -2048.0)
        scene.add_instance(model, xf.reshape(12))
        scene.commit()
        eye = (300.0, 200.0, -150.0)
    else:
        data, info = synth.castle_scene()
        desc = P.SceneDesc.from_vox(data)
        if "--props" in sys.argv:   # the castle + N scattered props (bench.py --props): the cull's 64-wide hierarchy
            P.scatter_props(desc, int(sys.argv[sys.argv.index("--props") + 1]))
        scene = P.hip_scene(ctx, desc)
        eye = (122.0, 300.61, 54.45)
    pipe = api.StandardPipeline(ctx, W, H)
    pipe.set_noise(0, synth.stbn_scalar())
    pipe.set_noise(5, synth.stbn_unitvec3_cosine())
    cam = api.make_camera(eye, api.look_at_rotation(eye, (0, 0, 0)), api.PinholeProjection())
    sky = P.sky_state()
    full = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_ACCUMULATE
    for f in range(1, 5):  # warm the surfel pool and the hash
        pipe.render(scene, cam, sky, full, f, synth.frame_rand(1, f))
    ctx.sync()
    read(lib)
    for title, passes, slot in (("k_primary_ao (fused primary + sun + AO)", L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, 0),
                                ("k_final_gather", L.PASS_FINAL_GATHER, 3), ("k_surfel_trace", L.PASS_SURFEL, 4)):
        pipe.render(scene, cam, sky, passes, 5, synth.frame_rand(1, 5))
        ctx.sync()
        b = read(lib)
        # (bucket 23 is the ray streams' trip counter AND the packet kernels' candidate-loop cycles: which kernels ran is known from the switches)
        streamed = slot >= 3 and ("DUST_HIP_RAY_STREAM" in os.environ or ("--deep" in sys.argv and slot == 3 and "DUST_HIP_PACKET_GI" not in os.environ))
        if streamed:
            report_stream(title + " -> k_ray_walk", b, pipe.pass_stats(slot).ms)
        else:
            report(title, b, pipe.pass_stats(slot).ms)


if __name__ == "__main__":
    main()
