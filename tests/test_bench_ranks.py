"""bench.py's rank function under gloo, world size 2, on CPU tensors with a recording stand-in for the pipeline: the N > 1 control
flow (both partitions in one run, band layout and padded gathers, the all-to-all by row slices, the GI exchange choreography,
the reductions behind the JSON line) executes here before the driver's 8-GPU node is the first to try it. The stand-in renders
nothing -- it stamps (rank, frame, row) patterns into the bound target so that the assembled frames can be checked -- and the
product library is not involved: this tests bench.py, not the kernels."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, json, types
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist
import bench
from dust_amd import _lib as L, sharding, synth

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
W, H = 64, 40          # 40 rows on 2 ranks: bands of 24 rows (8-aligned), the second one shorter -- the padded case
workload = os.environ["BENCH_WORKLOAD"]


class Stats:
    def __init__(self, rays=0, hits=0):
        self.rays, self.hits = rays, hits
        self.instances_tested = self.upper_descents = self.mid_descents = self.bricks_tested = 0


class StubPipe:
    """records what bench.py asks for; writes a pattern where a frame would go"""
    def __init__(self):
        self.target, self.calls, self.last = None, [], {}
        self.width, self.height = W, H
        self.n_render = 0
    def set_noise(self, *a): pass
    def set_frames_in_flight(self, n): self.calls.append(("in_flight", n))
    if os.environ.get("BENCH_REBALANCE", "0") != "0":   # (with `configure` the band cuts are rebalanced from measured band steps)
        def configure(self, frames_in_flight=None, in_flight_slots=None, **kw):
            self.calls.append(("in_flight", frames_in_flight))
    def configure_gi(self, *a): self.calls.append("configure_gi")
    def clear(self): self.calls.append("clear")
    def render(self, scene, cam, sky, passes, frame_index=1, rand=0, rows=(0, 0), surfel_shard=(0, 0)):
        r0, r1 = rows if rows[1] else (0, H)
        self.n_render += 1
        if surfel_shard[1]:
            self.calls.append(("surfel_shard",) + tuple(surfel_shard))
        self.last = {"passes": passes, "rows": (r0, r1), "frame": frame_index}
        if passes == L.PASS_DENOISE:
            self.calls.append(("denoise", frame_index, (r0, r1)))
        if passes & L.PASS_PRIMARY and self.target is None:
            self.px = (r1 - r0) * W   # (a frame into the pipeline's own plane: the cost-measuring launch before the targets exist)
        elif passes & L.PASS_PRIMARY:
            # RGBA16F stand-in: channel 0 = frame index, 1 = rank, 2 = row
            rows_t = torch.arange(r0, r1, dtype=torch.float16).view(-1, 1)
            self.target[r0:r1, :, 0] = float(frame_index % 1024)
            self.target[r0:r1, :, 1] = float(rank)
            self.target[r0:r1, :, 2] = rows_t
            self.px = (r1 - r0) * W
    def pass_stats(self, i):
        px = getattr(self, "px", 0)
        return Stats(rays=px if i < 4 else 100, hits=px // 2 if i < 4 else 50)
    def mark_kernel_times(self): self.n_render = 0
    def tile_costs(self, kind=0):
        c = np.ones((H // 8, W // 8), np.uint32)
        c[:2] = 5 + rank   # the top strips cost more -- and each rank's own measurement differs: rank 0's decides
        return c
    def kernel_times(self, mark=True):
        n = self.n_render
        self.n_render = 0
        return [0.25 * n, 0.0, 0.1 * n, 0.05 * n], [n, 0, n, n]
    def gi_exchange(self, padded_rows):
        self.ex = types.SimpleNamespace(pool_size=128, width=W, touched_rows=padded_rows)
        return self.ex
    def gi_export(self, r0, r1): self.calls.append(("export", r0, r1))
    def gi_import(self, r0, r1, f): self.calls.append(("import", r0, r1, f))


class StubComm:
    """api.Comm's interface over gloo: what bench.py's native path calls, recorded; the gather really moves the rows (in place on the
    root's target, as dust_hip_gather_bands does) so that the assembled frame can be checked"""
    log = []
    def __init__(self): self.n = 0
    def gather_bands(self, pipe, plane, cuts, root=0, dst_ptr=None, dst_bytes=0):
        cuts = [int(v) for v in cuts]
        assert plane == L.PLANE_ILLUMINANCE and cuts[0] == 0 and cuts[-1] == H and len(cuts) == world + 1
        if rank == root:
            for r in range(world):
                if r != root and cuts[r] < cuts[r + 1]:
                    buf = torch.empty_like(pipe.target[cuts[r]:cuts[r + 1]])
                    dist.recv(buf, src=r)
                    pipe.target[cuts[r]:cuts[r + 1]] = buf
            for r in range(world):   # every band of the assembled frame carries its rank and its rows
                if cuts[r] < cuts[r + 1]:
                    assert float(pipe.target[cuts[r], 0, 1]) == float(r) and float(pipe.target[cuts[r + 1] - 1, 0, 2]) == float(cuts[r + 1] - 1)
        elif cuts[rank] < cuts[rank + 1]:
            dist.send(pipe.target[cuts[rank]:cuts[rank + 1]].contiguous(), dst=root)
        self.n += 1
        StubComm.log.append(("gather", root, self.n))
        return self.n
    def gather_planes(self, pipe, planes, cuts, root=0):
        assert L.PLANE_DEPTH in planes and L.PLANE_ILLUMINANCE in planes and root == 0   # --denoise: the filter's planes, to the rank that keeps its history
        StubComm.log.append(("gather_planes", tuple(planes)))
        return self.gather_bands(pipe, L.PLANE_ILLUMINANCE, cuts, root)
    def wait(self, ticket=0): StubComm.log.append(("wait", ticket))
    def sync(self): StubComm.log.append(("sync",))
    def gi_surfel_exchange(self, pipe, frame_index): StubComm.log.append(("gi_surfel_exchange", frame_index))
    def gi_exchange(self, pipe, r0, r1, band_rows, frame_index):
        StubComm.log.append(("gi_exchange", r0, r1, band_rows, frame_index))
        pipe.gi_export(r0, r1)
        pipe.gi_import(r0, r1, frame_index)


class StubBackend:
    if os.environ.get("BENCH_NATIVE") == "1":
        def make_comm(self, dist_, ctx): return StubComm()

    def __init__(self):
        self.torch, self.L, self.sharding, self.synth = torch, L, sharding, synth
        self.rank, self.local_rank, self.world = rank, rank, world
        self.device = torch.device("cpu")
        self.pipes = []
    def sync(self): pass
    def open_lane(self, args, first):
        import contextlib
        lane = bench.Lane()
        lane.sc = {"scene": None, "cam": None, "sky": None, "info": {"n_models": 1, "n_instances": 1, "n_voxels": 1}, "n_bricks": 1,
                   "t_load": 0.0, "desc": None, "deep": None}
        lane.pipe = StubPipe()
        lane.ctx = None
        lane.enter = contextlib.nullcontext
        self.pipes.append(lane.pipe)
        return lane
    def noise(self): return None, None
    def sky_struct(self, sky): return sky
    def bind_target(self, pipe, tensor): pipe.target = tensor
    def alias_exchange(self, ex):
        return (torch.zeros(ex.pool_size, dtype=torch.int32), torch.zeros(ex.touched_rows * ex.width, dtype=torch.int32),
                torch.zeros(ex.pool_size * 4, dtype=torch.int32))
    def check_target(self, pipe, target, rows):
        assert float(target[rows[0], 0, 1]) == float(rank)


args = bench.parse(["--gpus", str(world), "--steps", "3", "--warmup", "1", "--width", str(W), "--height", str(H),
                    "--workload", workload, "--no-cpu-baseline", "--band-rebalance", os.environ.get("BENCH_REBALANCE", "0")]
                   + (["--denoise"] if os.environ.get("BENCH_DENOISE") == "1" else []))
bench.SETTLE_STEPS = 2
# the ranks' own clocks disagree about how many more settle frames are due (rank 0: as many as allowed, rank 1: none); every step
# holds a collective, so they must settle on one count or the job hangs
bench.SETTLE_SECONDS = 30.0 if rank == 0 else 0.0
bench.SETTLE_MAX_STEPS = 5
be = StubBackend()
out = bench.run_rank(args, be, dist)
if rank == 0:
    json.dumps(out)   # serialisable
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2, out
    assert out["scaling"] == "strong" and set(out["curves"]) == {"strong", "weak"}, out
    strong, weak = out["curves"]["strong"], out["curves"]["weak"]
    assert out["value"] == strong["value"]
    gi = workload == "gi"
    # bands: the two ranks' pixels add up to ONE frame; samples: one whole frame each
    extra = 200 if gi else 0   # the replicated surfel pass counts once
    classes = 4 if gi else 3
    assert strong["rays_per_step_all_gpus"] == W * H * classes + extra, strong
    assert weak["rays_per_step_all_gpus"] == 2 * (W * H * classes + extra), weak
    assert len(strong["per_rank_kernel_ms"]) == 2 and len(weak["per_rank_kernel_ms"]) == 2
    assert "bands x2" in strong["parallelism"] and "spp x2" in weak["parallelism"]
    calls = [c for p_ in be.pipes for c in p_.calls]
    assert out["curves"]["strong"]["frames_in_flight"] == (1 if gi else 4) and out["curves"]["weak"]["frames_in_flight"] == 1
    assert len(be.pipes) == (1 if gi else 4)      # row bands of a non-GI workload: four frames in flight, a pipeline each
    assert strong["settle_steps"] == weak["settle_steps"] == 2 + 5, (strong, weak)
    assert ("in_flight", 1 if gi else 4) in calls
    if not gi and os.environ.get("BENCH_REBALANCE", "0") == "0":
        assert strong["band_rows"] == [0, 8, 40], strong   # 5 strips costing 5,5,1,1,1 (rank 0's map; rank 1 measured 6,6,1,1,1): the boundary nearest to half the cost is after the first
        assert "equal measured cost" in strong["parallelism"]
    if not gi and os.environ.get("BENCH_REBALANCE", "0") != "0":   # cuts corrected from the ranks' own band-step times: rank 0's are everybody's (the gathers above checked every band's rows)
        cuts = strong["band_rows"]
        assert cuts[0] == 0 and cuts[-1] == H and len(cuts) == 3 and cuts[1] % 8 == 0, strong
    if gi:
        assert "clear" in calls and ("export", 0, 24) in calls
    if os.environ.get("BENCH_NATIVE") == "1":   # the library's own collectives: gathers with rotating roots, tickets waited for, the GI exchange in one call
        assert "libdust_hip" in strong["collectives"], strong
        kinds = [c[0] for c in StubComm.log]
        assert kinds.count("gather") >= 3 + 7 + 1 and "wait" in kinds and "sync" in kinds, kinds
        if os.environ.get("BENCH_DENOISE") == "1":   # the filter's history lives on rank 0: every gather goes there, all seven planes at once,
            assert {c[1] for c in StubComm.log if c[0] == "gather"} == {0} and "gather_planes" in kinds   # and rank 0 filters the whole frame
            assert any(c[0] == "denoise" and c[2] == (0, H) for c in calls if isinstance(c, tuple)), calls[:12]
            assert strong["denoise"] is True
        else:
            assert {c[1] for c in StubComm.log if c[0] == "gather"} == {0, 1}
        assert all(c[1] > 0 for c in StubComm.log if c[0] == "wait")
        if gi:
            assert ("gi_exchange", 0, 24, 24, 1) in StubComm.log, StubComm.log[:8]
            # the surfel trace sharded over the two ranks, completed by the library's all-gather + apply, once per frame
            assert ("surfel_shard", 0, 2) in calls and ("gi_surfel_exchange", 1) in StubComm.log and "sharded" in strong["parallelism"], strong["parallelism"]
    else:
        assert "torch.distributed" in strong["collectives"], strong
    print("BENCH_RANKS_OK", json.dumps(out)[:200])
else:
    assert out is None
    if workload == "gi":
        assert ("export", 24, 40) in be.pipes[0].calls   # the shorter, padded band
dist.barrier()
dist.destroy_process_group()
'''


def _run(workload, native=False, denoise=False, rebalance=0):
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1",
                   BENCH_WORKLOAD=workload, BENCH_NATIVE="1" if native else "0", BENCH_DENOISE="1" if denoise else "0", BENCH_REBALANCE=str(rebalance))
        procs.append(subprocess.Popen([sys.executable, "-c", f"ROOT={ROOT!r}\n" + WORKER], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, e[-3000:]
    assert "BENCH_RANKS_OK" in outs[0][0]


def test_bench_rank_function_two_ranks_primary_ao():
    _run("primary_ao")


def test_bench_rank_function_two_ranks_gi():
    _run("gi")


def test_bench_rank_function_two_ranks_native_collectives():
    """--comm native (the default): the band gather and the GI exchange through the library's communicator -- here a gloo stand-in
    with api.Comm's interface that really moves the rows, so the control flow (tickets, rotating roots, in-place assembly on the root,
    one gi_exchange call per frame) has executed before an 8-GPU node runs it over RCCL"""
    _run("primary_ao", native=True)
    _run("gi", native=True)


def test_bench_rank_function_two_ranks_rebalanced_bands():
    """--band-rebalance (round 6): every rank times its own band, the times are all-gathered, the cuts corrected and rank 0's broadcast --
    each rank rescales its OWN cost map, so without the broadcast the ranks would render different bands; the stand-in's gather asserts that
    every band of the assembled frame carries its rank and its rows"""
    _run("primary_ao", native=True, rebalance=2)


def test_bench_rank_function_two_ranks_denoised_frames():
    """--denoise on two ranks: one gather_planes per frame to rank 0 (which keeps the filter's history), then DUST_PASS_DENOISE there"""
    _run("primary_ao", native=True, denoise=True)


def test_bench_gpus_without_devices_says_so():
    """`python bench.py --gpus 2` on a node with fewer devices: a clear message, not a usage hint (here: no device at all)."""
    from dust_amd import _lib
    have = _lib.load().dust_hip_device_count()
    if have >= 2:
        return
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                       text=True, timeout=120, env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode != 0
    assert f"only {have} HIP device(s) visible" in r.stderr


def test_bench_teapot_cpu_line():
    """configs[0] has a bench line of its own: CPU traversal only, same JSON contract."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "teapot_cpu", "--steps", "2", "--warmup", "1"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    j = json.loads(r.stdout.strip().splitlines()[-1])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 0 and j["value"] > 0 and j["config"]["rays_per_step"]["primary"] == 256 * 256


import pytest  # noqa: E402


@pytest.mark.gpu
def test_bench_line_on_the_gpu_keeps_the_contract():
    """`python bench.py --gpus 1 --steps K --warmup W` as the driver runs it: one JSON line with the contract's keys, `roofline` and
    `cpu_baseline`, and numbers that are consistent with each other."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "8", "--warmup", "2", "--cpu-rows", "64"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 8 and j["warmup"] == 2 and j["unit"] == "Mrays/s" and j["vs_baseline"] is None
    rf = j["roofline"]
    assert rf["bound"] == "hbm" and rf["peak"] == 8000.0 and rf["unit"] == "GB/s" and 0.0 < rf["frac"] < 1.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-6
    # a launch cannot take longer than the steps it holds (the default line: eight frames = eight steps per launch, dust_hip_render_frames)
    assert rf["frames_per_launch"] == j["config"]["frames_per_launch"] == 8 and rf["kernel"] == "k_primary_ao_batch"
    assert 0.0 < rf["kernel_ms"] <= j["ms_per_step"] * rf["frames_per_launch"] * 1.05
    assert abs(rf["kernel_ms_per_frame"] - rf["kernel_ms"] / 8) < 1e-3
    one = j["curves"]["one_frame_per_launch"]   # rounds 1-5's headline rides along
    assert one["roofline"]["kernel"] == "k_primary_ao" and 0.0 < one["roofline"]["kernel_ms"] <= one["ms_per_step"] * 1.05
    assert abs(rf["achieved"] - rf["algorithmic_bytes_per_launch"] / (rf["kernel_ms"] * 1e-3) / 1e9) / rf["achieved"] < 1e-3
    rays = j["config"]["rays_per_step_all_gpus"]
    assert abs(j["value"] - rays / (j["ms_per_step"] * 1e-3) / 1e6) / j["value"] < 2e-3
    cb = j["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0
