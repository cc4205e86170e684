"""bench.py's --frames-per-launch control flow on CPU, with recording stand-ins for the pipeline and the binding (no GPU, no library): the
headline's eight frames per dust_hip_render_frames call, the calls marshalled before the timed region, EXACTLY K steps with the short launch
first when 8 does not divide K, the launch's bytes and frames in the roofline. The product is not involved: this tests bench.py."""
import contextlib
import json
import types

import numpy as np
import torch

import bench
from dust_amd import _lib as L, sharding, synth

W, H = 64, 40


class Stats:
    def __init__(self, rays=0, hits=0):
        self.rays, self.hits = rays, hits
        self.instances_tested = self.upper_descents = self.mid_descents = self.bricks_tested = 0


class StubPipe:
    log = []   # every launch, in order: ("frame", index) or ("frames", [indices])

    def __init__(self):
        self.target, self.launches_ms = None, []
        self.width, self.height = W, H

    def set_noise(self, *a): pass
    def configure(self, **kw): self.config = kw
    def configure_gi(self, *a): pass
    def clear(self): pass

    def render(self, scene, cam, sky, passes, frame_index=1, rand=0, rows=(0, 0), surfel_shard=(0, 0)):
        StubPipe.log.append(("frame", frame_index))
        self.launches_ms.append(0.25)
        self.px = W * H
        if self.target is not None:
            self.target[:, :, 0] = float(frame_index % 1024)

    def pass_stats(self, i): return Stats(rays=getattr(self, "px", 0), hits=getattr(self, "px", 0) // 2)
    def mark_kernel_times(self): self.launches_ms = []
    def tile_costs(self, kind=0): return np.ones((H // 8, W // 8), np.uint32)

    def kernel_times(self, mark=True):
        ms, n = sum(self.launches_ms), len(self.launches_ms)
        self.launches_ms = []
        return [ms, 0.0, 0.0, 0.0], [n, 0, 0, 0]


class StubStandardPipeline:
    @staticmethod
    def frames_call(pipes, scene, cameras, skies, passes, frame_indices, rands, rows=(0, 0), moves=None):
        idx = [int(v) for v in frame_indices]
        assert len(pipes) == len(idx) == len(rands) and len(set(id(p) for p in pipes)) == len(pipes) and len(cameras) >= len(pipes)

        def call():
            StubPipe.log.append(("frames", idx))
            pipes[0].launches_ms.append(0.21 * len(idx))   # the first pipeline's event pair brackets the launch of all frames
        return call

    @staticmethod
    def render_frames(*a, **kw):
        StubStandardPipeline.frames_call(*a, **kw)()


class StubBackend:
    def __init__(self):
        self.torch, self.L, self.sharding, self.synth = torch, L, sharding, synth
        self.api = types.SimpleNamespace(StandardPipeline=StubStandardPipeline)
        self.rank, self.local_rank, self.world = 0, 0, 1
        self.device = torch.device("cpu")

    def sync(self): pass

    def open_lane(self, args, first):
        lane = bench.Lane()
        lane.sc = {"scene": None, "cam": L.Camera(), "sky": None, "info": {"n_models": 1, "n_instances": 1, "n_voxels": 1}, "n_bricks": 1, "t_load": 0.0,
                   "desc": None, "deep": None}
        lane.pipe, lane.ctx, lane.stream, lane.enter = StubPipe(), object(), None, contextlib.nullcontext
        return lane

    def open_batch_lane(self, args, first):
        lane = bench.Lane()
        lane.sc, lane.ctx, lane.stream, lane.enter, lane.pipe = first.sc, first.ctx, first.stream, first.enter, StubPipe()
        return lane

    def noise(self): return None, None
    def sky_struct(self, sky): return L.Sky()
    def bind_target(self, pipe, tensor): pipe.target = tensor
    def check_target(self, pipe, target, rows): pass


def _run(steps, fpl=None):
    StubPipe.log = []
    argv = ["--gpus", "1", "--steps", str(steps), "--warmup", "1", "--width", str(W), "--height", str(H), "--no-cpu-baseline", "--no-extra-curves"]
    if fpl is not None:
        argv += ["--frames-per-launch", str(fpl)]
    args = bench.parse(argv)
    old = bench.SETTLE_STEPS, bench.SETTLE_SECONDS
    bench.SETTLE_STEPS, bench.SETTLE_SECONDS = 16, 0.0
    try:
        out = bench.run_rank(args, StubBackend(), None)
    finally:
        bench.SETTLE_STEPS, bench.SETTLE_SECONDS = old
    json.dumps(out)
    return out, list(StubPipe.log)


def test_default_line_renders_eight_frames_per_launch_and_exactly_k_steps():
    out, log = _run(20)
    assert out["steps"] == 20 and out["config"]["frames_per_launch"] == 8 and "8 consecutive frames per persistent launch" in out["config"]["parallelism"]
    # the counting frame alone, 16 settle steps = two launches of eight, then the timed 20 = a short launch FIRST (4), then whole ones
    assert log[0][0] == "frame"
    sizes = [len(e[1]) for e in log[1:]]
    assert all(e[0] == "frames" for e in log[1:]) and sizes == [8, 8, 4, 8, 8]
    timed = [i for e in log[3:] for i in e[1]]
    assert timed == list(range(timed[0], timed[0] + 20))   # consecutive frame indices, each once
    rf = out["roofline"]
    assert rf["kernel"] == "k_primary_ao_batch" and rf["frames_per_launch"] == 8
    frames_per_launch = 20 / 3   # what the averaged launch of the timed region carries
    assert abs(rf["kernel_ms"] - 0.21 * frames_per_launch) < 1e-6 and abs(rf["kernel_ms_per_frame"] - 0.21) < 1e-3
    assert rf["algorithmic_bytes_per_launch"] > 0


def test_a_launch_per_frame_and_other_counts():
    out, log = _run(12, fpl=1)
    assert out["config"]["frames_per_launch"] == 1 and out["roofline"]["kernel"] == "k_primary_ao" and all(e[0] == "frame" for e in log)
    out, log = _run(12, fpl=3)
    assert out["config"]["frames_per_launch"] == 3 and [len(e[1]) for e in log[1:]] == [1, 3, 3, 3, 3, 3, 3, 3, 3, 3]   # (16 settle steps: a frame, then five launches; 12 timed: four)
