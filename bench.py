#!/usr/bin/env python3
"""bench.py -- Mrays/s of the HIP ray-tracing hot path, one command per BASELINE.json configuration.

  --workload primary_ao   configs[1] (the headline, default): castle stand-in, 1920x1080, primary + sun-shadow + AO rays
  --workload gi           configs[2]/[3]: the same scene, all four passes + accumulation (--width 3840 --height 2160 for [3])
  --workload deep         configs[4]: procedural 4096^3 tree (hierarchy (4,4,2,2)) at 1 % brick occupancy, GI frame
  --workload teapot_cpu   configs[0]: teapot, 256x256, one primary ray per pixel, CPU software traversal only (no GPU touched)

One "step" = one frame of the hot path over resident inputs (StandardPipeline::render, standard.rs:477-725).
With N GPUs (one process per GPU, torch.distributed over RCCL; `python bench.py --gpus N` starts the N ranks itself when it
was not started by torch.distributed.run) the work is cut two ways, and by default BOTH are measured in one run:
  bands    ONE frame per step, cut into N row bands (DustHipFrameParams.row_begin/row_end; SURVEY 8e, north_star's partition)
           with an RCCL gather of the bands: STRONG scaling. This is `value`. GI workloads keep an identical spatial hash +
           surfel pool on every GPU through the exchange of dust_hip_pipeline_gi_exchange (three small collectives per frame)
           and the deterministic apply (DUST_PASS_GI_ORDERED); the surfel pass is replicated.
  samples  "N spp" in the reference is N consecutive frames (frame_index -> STBN slice, fresh rand): rank r renders
           sample k*N + r of the same view, one whole frame per GPU and step: WEAK scaling, reported under `curves.weak`.
--shard bands|samples measures only that one. Either way every rank's RGBA16F illuminance leaves its GPU over RCCL inside the
timed region. Whole frames (samples) are assembled by row slices -- one all-to-all per step, rank j receives slice j of every
rank's frame (--assemble slices, the default: xGMI is point to point, and a gather pushes every peer's whole frame through its
one link to the root); --assemble rotate / fixed gather them onto rank k % N / rank 0 instead. Row bands are gathered
(rotating root by default).
A ray = one traceRayEXT equivalent actually issued, counted per class by the counting build of the kernels in an untimed frame.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant kernel, algorithmic bytes / HIP-event
time) and `cpu_baseline` (the C oracle's hierarchical traversal, a port -- the reference has no CPU ray traversal at all,
SURVEY F2 -- on a bounded row sample of the same frame).
"""
import argparse
import ctypes
import gc
import json
import os
import socket
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (6.3 TB/s achievable)
NAMES = ("primary", "sun_shadow", "ambient_occlusion", "final_gather", "surfel_sun", "surfel_cosine")
# Untimed frames before the timed region, whatever --warmup says: the cost-ordered hand-out needs measured frames of the SAME
# view to settle (it re-measures every 8th launch of a still view, capi.cpp order_tiles) and the clocks ramp up over the first
# few milliseconds -- a 20-step run after 5 warm-up frames read 11 % low in round 2.
SETTLE_STEPS = 64
SETTLE_SECONDS = 0.3   # ... and at least this long, back to back, so that the timed region starts on a GPU at its sustained clocks
SETTLE_MAX_STEPS = 20000


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=96)   # (whole launches at 1, 2, 3, 4 and 8 frames per launch)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scale", type=float, default=1.0, help="castle stand-in scale (1.0 = BASELINE config)")
    ap.add_argument("--workload", choices=["primary_ao", "gi", "deep", "teapot_cpu"], default="primary_ao")
    ap.add_argument("--shard", choices=["samples", "bands", "both"], default=None,
                    help="N > 1: bands = one frame in N row bands (strong scaling), samples = one frame per GPU (weak scaling); "
                         "default both, `value` = bands")
    ap.add_argument("--gi-shard", choices=["samples", "bands"], default=None, help="older spelling of --shard")
    ap.add_argument("--assemble", choices=["slices", "rotate", "fixed"], default="slices",
                    help="N > 1, how finished frames leave their GPU: slices = one all-to-all, rank j assembles row slice j of every "
                         "frame (default; every xGMI link in use); rotate / fixed = RCCL gather of whole frames onto rank k %% N / rank 0 "
                         "(bound by each peer's single link to the root). Row bands are always gathered (default root: rotating)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="frames of the sequence on the GPU at once, each on a stream, context and G-buffer of its own (0 = auto: 4 for "
                         "row bands on N > 1 GPUs, where a rank's launch is as long as its most expensive tile and most of the GPU would "
                         "idle behind it; 1 otherwise). GI workloads run one frame at a time (every frame reads the previous one's hash)")
    ap.add_argument("--frames-per-launch", type=int, default=None,
                    help="one GPU, primary_ao: this many consecutive frames of the sequence (at most 8) go to the device in ONE call and ONE persistent "
                         "launch (dust_hip_render_frames: G-buffers of their own on one context and stream; a wavefront that finds frame i without tiles "
                         "goes on to frame i + 1, so the launch's tail, root staging and inter-launch gap are paid once per launch; every frame's planes "
                         "are the bits the frame rendered alone gives). 1 = a launch per frame (rounds 1-5's headline; the default line carries it as "
                         "curves.one_frame_per_launch, and four per launch as curves.four_frames_per_launch). Default: 8 on one GPU (offline rendering: the "
                         "reference's host keeps up to three frames in flight, rhyolite_bevy/src/lib.rs:58; three per launch measure 0.397-0.404 of the "
                         "roofline proxy, four 0.399-0.408, eight 0.402-0.412 from box to box) and 1 for GI workloads, N > 1 GPUs, emulated bands, "
                         "--camera orbit and --frames-in-flight runs")
    ap.add_argument("--in-flight-slots", choices=["auto", "share", "all"], default="auto",
                    help="with several frames in flight: share = every launch on 1/D of the workgroup slots (row bands: a band is as long as its "
                         "heaviest tile, D of them side by side fill the device); all = every launch asks for ALL slots, so the next frame's workgroups "
                         "start on the CUs the previous frame's last tiles have left (whole frames on one GPU); auto = share for row bands of an N > 1 "
                         "job (or an emulated band), all otherwise (DustHipPipelineConfig.in_flight_slots)")
    ap.add_argument("--band-cuts", choices=["cost", "rows"], default="cost",
                    help="row bands of a non-GI workload: cost = bands of about equal measured cost (one untimed whole-frame launch records every "
                         "tile's cycles, rank 0's map decides); rows = equal row counts")
    ap.add_argument("--band-rebalance", type=int, default=3,
                    help="row bands at equal cost: rounds of proportional correction of the cuts from MEASURED band-step times (every rank times its "
                         "own band under the run's frames in flight, without collectives; sharding.rebalance_cuts) before the timed region; 0 = the "
                         "cuts of the one whole-frame tile-cost map (round 5)")
    ap.add_argument("--deep-occupancy", type=float, default=0.01, help="--workload deep: occupied share of the brick lattice")
    ap.add_argument("--comm", choices=["native", "torch"], default="native",
                    help="N > 1, row bands: native = the library's own RCCL path (dust_hip_comm_create / dust_hip_gather_bands / "
                         "dust_hip_gi_exchange_run: grouped send / receive on the communicator's stream, enqueued by the call that follows the "
                         "band's render call); torch = the same collectives through torch.distributed (round 3)")
    ap.add_argument("--denoise", action="store_true",
                    help="primary_ao: every frame is also filtered (DUST_PASS_DENOISE, the reference's NRD step). On N > 1 GPUs with row bands and the "
                         "native communicator: ONE dust_hip_gather_planes moves the five planes the filter reads (+ the two the tone map reads) to rank 0, "
                         "which filters the whole frame")
    ap.add_argument("--props", type=int, default=0,
                    help="castle workloads: this many small extra instances scattered over the scene (curves.many_instances uses 4000: the packet "
                         "cull then goes through its 64-wide hierarchy, cull_instances in traverse.hpp)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--camera", choices=["still", "orbit"], default="still",
                    help="orbit: the headline becomes the MOVING view (the eye swaying along its circle round the castle, the teapot of "
                         "examples/castle.rs:287-291 swinging, set_transform + commit every frame); the default line carries it as curves.moving")
    ap.add_argument("--no-extra-curves", action="store_true",
                    help="one GPU, default workload: skip curves.moving / gi_1080p / primary_ao_4k / deep (short runs after the headline's timed region)")
    ap.add_argument("--extra-steps", type=int, default=40, help="timed steps of each extra curve (40: a 10-step region of 0.7 ms GI frames is 7 ms, of which the two synchronisations around it are percents)")
    ap.add_argument("--extra-timeout", type=float, default=150.0, help="seconds the extra curves may take before the line is printed without them")
    ap.add_argument("--assets", default=None,
                    help="directory with the reference's LFS assets (castle.vox, teapot.vox, stbn_scalar_*.png, stbn_unitvec3_cosine_*.png): a file "
                         "whose sha256 is the oid in /root/reference/assets/* replaces its stand-in, and config.workload says so (SURVEY 8d)")
    ap.add_argument("--cpu-rows", type=int, default=0, help="rows of the frame the CPU baseline traces (0 = auto)")
    a = ap.parse_args(argv)
    a.shard = a.shard or a.gi_shard or "both"
    return a


def algorithmic_bytes(st, gbuffer_bytes):
    """64 B per instance tested, 8+4 per root/mid child descended, 24 per brick tested,
    1+4 per hit (material byte + palette entry), plus the pass's G-buffer traffic."""
    return (st.instances_tested * 64 + (st.upper_descents + st.mid_descents) * 12 + st.bricks_tested * 24 + st.hits * 5
            + gbuffer_bytes)


# ---------------------------------------------------------------------------------------------------------------------
# Backends. The measurement below talks to the GPU through this object only, so that tests/test_bench_ranks.py can run the same
# rank function under gloo on CPU tensors with a recording stand-in for the pipeline (the N > 1 control flow must have executed
# somewhere before the driver's 8-GPU node is the first to try it).

class HipBackend:
    """The product: libdust_hip.so through dust_amd.api, torch for device tensors, streams and RCCL."""
    dist_backend = "nccl"

    def __init__(self, rank, local_rank, world):
        import torch
        if not torch.cuda.is_available():
            sys.exit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if local_rank >= torch.cuda.device_count():
            sys.exit(f"bench.py: rank {rank} wants device {local_rank} but only {torch.cuda.device_count()} HIP device(s) are visible")
        from dust_amd import _lib as L
        from dust_amd import api, sharding, synth
        from dust_amd import scenes as P  # scene helpers + packaged sky; the oracle is only imported in the cpu_baseline leg
        self.torch, self.L, self.api, self.sharding, self.synth, self.P = torch, L, api, sharding, synth, P
        self.rank, self.local_rank, self.world = rank, local_rank, world
        from dust_amd import assets as _assets
        self.assets = _assets.Assets(None)  # main() replaces it when --assets is given
        self.device = torch.device("cuda", local_rank)
        torch.cuda.set_device(local_rank)
        # One explicit stream for everything: the library's launches, torch's own kernels, and the point RCCL orders its
        # collectives against (torch's "current stream"). torch's DEFAULT stream has the handle 0, which the library reads as
        # "no stream given" and would answer with a private non-blocking stream that nothing of torch's is ordered with.
        # (world > 1: every pipeline leaves 32 of the 512 workgroup slots empty so that RCCL's send / receive kernels can run next to the
        #  traversal kernels, which otherwise hold every VGPR of every SIMD -- DustHipPipelineConfig.reserve_blocks, set in open_lane; the
        #  library does the same on its own once a pipeline has been through one of its collectives, DUST_RESERVE_AUTO)
        self.streams = []
        self.stream, self.ctx = self._stream_and_context()
        torch.cuda.set_stream(self.stream)

    def _stream_and_context(self):
        stream = self.torch.cuda.Stream(device=self.local_rank)
        assert stream.cuda_stream != 0
        self.streams.append(stream)
        # (kernel times from event pairs around every 4th frame's launches: a record costs the stream ~6 us, 5 % of this frame)
        return stream, self.api.Context(device=self.local_rank, timing=True, sparse_timing=True, stream=ctypes.c_void_p(stream.cuda_stream))

    def init_dist(self):
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=self.rank, world_size=self.world, device_id=self.device)
        return dist

    def sync(self):
        # poll first: an interrupt-driven wait wakes tens to hundreds of microseconds after the GPU is done, which is percents of a
        # 20-step timed region; hipStreamQuery in a loop sees the end within a couple of microseconds
        for st in self.streams:
            while not st.query():
                pass
        self.torch.cuda.synchronize()

    def open_lane(self, args, first):
        """One frame in flight: a stream, a context on it, the scene uploaded to that context, a pipeline (G-buffer). The first lane is
        the backend's own stream and context; `first` (its scene dict) spares the others the parse."""
        lane = Lane()
        if first is None:
            lane.stream, lane.ctx = self.stream, self.ctx
            lane.sc = self.build_scene(args)
        else:
            lane.stream, lane.ctx = self._stream_and_context()
            lane.sc = dict(first)
            lane.sc["scene"] = self.P.hip_scene(lane.ctx, first["desc"])
        lane.pipe = self.api.StandardPipeline(lane.ctx, args.width, args.height)
        if self.world > 1 or os.environ.get("DUST_BENCH_EMULATE_BAND"):
            lane.pipe.configure(reserve_blocks=32)
        lane.enter = lambda: self.torch.cuda.stream(lane.stream)
        return lane

    def open_batch_lane(self, args, first):
        """A further frame of a batched launch (dust_hip_render_frames): a pipeline (G-buffer) of its own on the FIRST lane's context, stream and scene."""
        lane = Lane()
        lane.stream, lane.ctx, lane.sc = first.stream, first.ctx, first.sc
        lane.pipe = self.api.StandardPipeline(lane.ctx, args.width, args.height)
        if self.world > 1 or os.environ.get("DUST_BENCH_EMULATE_BAND"):
            lane.pipe.configure(reserve_blocks=32)   # (as open_lane: frames of one launch have one configuration)
        lane.enter = first.enter
        return lane

    def build_scene(self, args, ctx=None):
        """-> dict(scene, cam, sky, info, n_bricks, t_load, desc, deep=(blocks, mats, pal, xf) or None)"""
        import numpy as np
        api, synth, P = self.api, self.synth, self.P
        t0 = time.time()
        out = {"deep": None, "desc": None}
        if args.workload == "deep":  # brick occupied iff hash(seed, bx, by, bz) < occupancy, voxels set with p = 0.5
            blocks, mats = synth.procedural_deep_blocks(occupancy=args.deep_occupancy, sample=True)
            pal = synth.make_palette(5)
            t0 = time.time()
            model = api.Model(self.ctx, blocks, mats, pal, tree_extent_log2=12)
            scene = api.Scene(self.ctx)
            xf = np.eye(3, 4, dtype=np.float32)
            xf[:, 3] = (-2048.0, -2048.0, -2048.0)
            scene.add_instance(model, xf.reshape(12))
            scene.commit()
            out.update(t_load=time.time() - t0, info={"n_models": 1, "n_instances": 1, "n_voxels": int(len(mats))},
                       n_bricks=int(len(blocks)), deep=(blocks, mats, pal, xf))
            eye, target = (300.0, 200.0, -150.0), (0.0, 0.0, 0.0)  # inside the volume
        else:
            data, info, real = self.assets.castle(args.scale)   # the reference's castle.vox if --assets holds it (sha256), else the stand-in
            t0 = time.time()
            desc = P.SceneDesc.from_vox(data)  # dust_vox_load: parse + tree build + flatten, models in parallel threads
            t_load = time.time() - t0
            if info is None:
                info = {"n_models": len(desc.models), "n_instances": len(desc.instances), "n_voxels": int(sum(len(m) for _, m in desc.models))}
            out["real_castle"] = real
            if getattr(args, "props", 0):   # seeded props: 8 small models, arbitrary rotations about the vertical (y) axis, on and above the ground
                P.scatter_props(desc, int(args.props), args.scale)
                info = dict(info, n_models=len(desc.models), n_instances=len(desc.instances), props=int(args.props))
            scene = P.hip_scene(self.ctx, desc)
            s = args.scale
            eye, target = (122.0 * s, 300.61 * s, 54.45 * s), (0.0, 0.0, 0.0)  # examples/castle.rs:120-129, fov pi/4
            out.update(t_load=t_load, info=info, n_bricks=desc.n_bricks(), desc=desc)
        out.update(scene=scene, sky=P.sky_state("default"),
                   cam=api.make_camera(eye, api.look_at_rotation(eye, target), api.PinholeProjection()))
        return out

    def make_comm(self, dist, ctx):
        """The library's RCCL communicator for one context (lane): rank 0 makes the id, torch.distributed carries it."""
        torch = self.torch
        # every rank makes the same collectives whatever fails where: the id travels unconditionally (all zero = "rank 0 could not make one"),
        # then the ranks agree on whether every one of them got its communicator -- a rank that raised alone would leave the others in a
        # collective it never joins
        uid = torch.zeros(128, dtype=torch.uint8, device=self.device)
        err = None
        if self.rank == 0:
            try:
                uid.copy_(torch.frombuffer(bytearray(self.api.Comm.unique_id()), dtype=torch.uint8))
            except Exception as e:  # noqa: BLE001 -- e.g. no librccl on this node
                err = e
        dist.broadcast(uid, src=0)
        comm = None
        if err is None and bool(uid.any().item()):
            try:
                comm = self.api.Comm.create(ctx, self.rank, self.world, bytes(uid.cpu().numpy().tobytes()))
            except Exception as e:  # noqa: BLE001
                err = e
        ok = torch.tensor([1 if comm is not None else 0], dtype=torch.int64, device=self.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if not bool(ok.item()):
            raise RuntimeError(f"native communicator unavailable on at least one rank ({err or 'another rank failed'})")
        return comm

    def noise(self):
        return self.assets.noise()   # the reference's STBN textures if --assets holds them (sha256), else the stand-ins

    def bind_target(self, pipe, tensor):
        pipe.bind_plane(self.L.PLANE_ILLUMINANCE, tensor.data_ptr(), tensor.numel() * 2)

    def alias_exchange(self, ex):
        return self.sharding.alias_exchange_buffers(ex)

    def sky_struct(self, sky):
        return self.api.sky_struct(sky)

    def check_target(self, pipe, target, rows):
        """untimed self-check of the plumbing the gather relies on: the bound torch tensor is where the frame went
        (tests/test_gpu_parity.py::test_bound_plane_equals_own_storage shows a bound target gets the pipeline's own bits)"""
        import numpy as np
        torch = self.torch
        own = torch.from_numpy(pipe.read_plane(self.L.PLANE_ILLUMINANCE)[rows[0]:rows[1]].view(np.int16))
        assert bool((own != 0).any()) and torch.equal(target[rows[0]:rows[1]].cpu().view(torch.int16), own)


class Lane:
    """a frame in flight: stream, context, scene copy and pipeline (see HipBackend.open_lane)"""
    comm = None


class NativeGather:
    """AsyncGather's interface over the library's own collectives (dust_hip_gather_bands): slot b is render target b of lane b % D;
    the gather of step k moves every rank's rows of that target to the root's copy of it, in place, on the communicator's stream."""

    def __init__(self, comms, pipes, plane, cuts, slots, rotate, world, planes=None):
        self.comms, self.pipes, self.plane, self.slots, self.rotate, self.world = comms, pipes, plane, slots, rotate, world
        self.planes = planes
        self.cuts = (ctypes.c_uint32 * len(cuts))(*[int(v) for v in cuts])
        self.tickets = [0] * slots
        self.root = 0

    def wait_slot(self, b):
        b %= self.slots
        if self.tickets[b]:
            self.comms[b % len(self.comms)].wait(self.tickets[b])   # (on the device: the lane's stream waits for the gather that last read target b)
            self.tickets[b] = 0

    def submit_slot(self, b, k):
        self.root = k % self.world if self.rotate else 0
        lane = b % len(self.comms)
        if self.planes:   # --denoise: every plane the root's filter and tone map read, in one collective
            self.tickets[b] = self.comms[lane].gather_planes(self.pipes[lane], self.planes, self.cuts, self.root)
        else:
            self.tickets[b] = self.comms[lane].gather_bands(self.pipes[lane], self.plane, self.cuts, self.root)

    def finish(self):
        for c in self.comms:
            c.sync()
        self.tickets = [0] * self.slots

    def last_root(self):
        return self.root


def measure_curve(be, dist, args, lanes, shard):
    """One timed region: the untimed counting frame, the settle + warm-up frames, K timed steps bracketed by barriers.
    shard: "bands" (one frame in world row bands) or "samples" (one frame per rank). Returns a dict (same on every rank after
    the reductions) with the whole job's rays per step, the max-over-ranks time and every rank's kernel times."""
    torch, L, sharding, synth = be.torch, be.L, be.sharding, be.synth
    rank, world = be.rank, be.world
    W, H = args.width, args.height
    sc, pipe = lanes[0].sc, lanes[0].pipe
    cam, sky = sc["cam"], be.sky_struct(sc["sky"])  # (converted once: the frame loop's host time per step is what a band-sized step is made of)
    gi_mode = args.workload in ("gi", "deep")
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    if gi_mode:  # diffuse GI through the surfel-fed spatial hash
        passes |= L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_ACCUMULATE
        if os.environ.get("DUST_BENCH_GI_ORDERED"):   # (diagnostic: one GPU with the deterministic apply an N-rank job needs)
            passes |= L.PASS_GI_ORDERED
    bands = shard == "bands"
    denoise = bool(getattr(args, "denoise", False)) and not gi_mode
    if denoise and world == 1:
        passes |= L.PASS_DENOISE   # one GPU: the filter is one more pass of the frame
    per_rows, rows, send = sharding.band_layout(rank, world, H) if bands else (H, (0, H), (0, H))
    emulate = os.environ.get("DUST_BENCH_EMULATE_BAND")  # "r/N" on ONE GPU: this rank renders band r of N, nothing is gathered --
    emulate = emulate if (emulate and bands and world == 1) else None   # what a rank of an N-GPU strong-scaling run does between collectives
    # ("all/N": every band of N is timed while the cuts are balanced, and the timed region then runs the SLOWEST one -- the frame is done when it is)
    er_all = bool(emulate) and emulate.split("/")[0] == "all"
    er, en = ((0 if er_all else int(emulate.split("/")[0])), int(emulate.split("/")[1])) if emulate else (rank, world)
    if emulate:
        per_rows, rows, send = sharding.band_layout(er, en, H)
        send = (send[0], en * per_rows)  # (sizes the padded render target as the N-rank run would)
    D = len(lanes)
    # --frames-per-launch: the D lanes are G-buffers on ONE context; D consecutive steps are one dust_hip_render_frames call = one launch
    fpl = int(getattr(args, "frames_per_launch", 0) or 0)
    batched = fpl > 1 and world == 1 and not gi_mode and D == fpl and all(ln.ctx is lanes[0].ctx for ln in lanes)
    slots_mode = getattr(args, "in_flight_slots", "auto")
    if slots_mode == "auto":
        slots_mode = "share" if (bands and (world > 1 or emulate)) else "all"
    for lane in lanes:
        if batched:
            lane.pipe.configure(frames_in_flight=1, in_flight_slots=slots_mode)   # one launch at a time, on every slot; the frames of a launch have ONE configuration
        elif hasattr(lane.pipe, "configure"):
            lane.pipe.configure(frames_in_flight=D, in_flight_slots=slots_mode)  # D launches side by side on 1/D of the workgroup slots each, or one behind the other's tail
        else:
            lane.pipe.set_frames_in_flight(D)
    band_cuts, balance_log, band_steps_ms = None, [], None
    if bands and not gi_mode and en > 1 and args.band_cuts == "cost":
        # Bands of equal COST instead of equal rows (a frame is done when its slowest band is: the castle's top half takes 0.122 ms,
        # its bottom half 0.096). One untimed whole-frame launch records every tile's cycles; rank 0's map decides the cuts for all.
        pipe.render(sc["scene"], cam, sky, passes, frame_index=1, rand=synth.frame_rand(1, 1))
        be.sync()
        costs = pipe.tile_costs(0)
        strip_cost = None if costs is None else costs.sum(axis=1)
        band_cuts = sharding.balanced_cuts(strip_cost, en, H)
        if world > 1:
            agreed = torch.tensor(band_cuts, dtype=torch.int64, device=be.device)
            dist.broadcast(agreed, src=0)
            band_cuts = [int(v) for v in agreed.tolist()]
        # ... and that map, taken under ONE launch that had the device to itself, mispredicts a band's step under D frames in flight by +-15 %
        # (round 5: seven bands of eight at 0.032-0.034 ms, one at 0.0425): a few rounds of proportional correction from measured band-step
        # times. Every rank times ITS band -- frames in flight as in the timed region, no collective inside -- and the times are shared.
        BAL_WARM, BAL_STEPS = 48, 96

        def band_step_ms(r, cuts):
            _, rws, _ = sharding.layout_from_cuts(r, cuts, H)
            if rws[0] >= rws[1]:
                return 0.0
            for phase in (BAL_WARM, BAL_STEPS):
                be.sync()
                t0 = time.perf_counter()
                for k in range(0, phase, D if batched else 1):
                    if batched:
                        idx = [1 + k + j for j in range(D)]
                        be.api.StandardPipeline.render_frames([ln.pipe for ln in lanes], lanes[0].sc["scene"], cam, sky, passes, idx,
                                                              [synth.frame_rand(1, v) for v in idx], rows=rws)
                        continue
                    ln = lanes[k % D]
                    ln.pipe.render(ln.sc["scene"], cam, sky, passes, frame_index=1 + k, rand=synth.frame_rand(1, 1 + k), rows=rws)
                be.sync()
            return (time.perf_counter() - t0) / BAL_STEPS * 1e3

        def all_band_ms(cuts):
            if world > 1:
                mine = torch.tensor([band_step_ms(rank, cuts)], dtype=torch.float64, device=be.device)
                every = [torch.empty_like(mine) for _ in range(world)]
                dist.all_gather(every, mine)
                return [float(v.item()) for v in every]
            return [band_step_ms(r, cuts) for r in range(en)] if (er_all or getattr(args, "band_rebalance", 0)) else None
        rounds = int(getattr(args, "band_rebalance", 0)) if hasattr(lanes[0].pipe, "configure") else 0
        best = None
        for rd in range(rounds + 1):
            ms_now = all_band_ms(band_cuts) if (rounds or er_all) else None
            if ms_now is None:
                break
            balance_log.append({"cuts": list(band_cuts), "band_ms": [round(v, 4) for v in ms_now]})
            if best is None or max(ms_now) < max(best[1]):
                best = (list(band_cuts), ms_now)
            if rd == rounds:
                break
            band_cuts, strip_cost = sharding.rebalance_cuts(strip_cost, band_cuts, ms_now, en, H)
            if world > 1:   # (every rank rescaled ITS OWN cost map: rank 0's cuts are everybody's)
                agreed = torch.tensor(band_cuts, dtype=torch.int64, device=be.device)
                dist.broadcast(agreed, src=0)
                band_cuts = [int(v) for v in agreed.tolist()]
        if best is not None:
            band_cuts, band_steps_ms = best   # (the best set seen: a correction may overshoot; the same on every rank: the times were all-gathered, the cuts broadcast)
            if er_all:
                er = max(range(en), key=lambda r: band_steps_ms[r])
        per_rows, rows, send = sharding.layout_from_cuts(er, band_cuts, H)
    have_rows = rows[0] < rows[1]                                        # a rank past the end of the frame renders no pixels
    gi_bands = gi_mode and bands and world > 1   # (one GPU: the band is the frame, nothing to exchange, the racy apply is fine)
    # DUST_BENCH_EMULATE_BAND=r/N on a GI workload (one GPU): what rank r of an N-rank GI job does between collectives -- its band's pixel
    # passes, the exchange's export / import on its own band, 1/N of the surfel trace (slots [r S, (r + 1) S) of the ordered pool), the
    # replicated ordering and ordered apply. The other ranks' records in the staging arrays are those of the last FULL trace (the first
    # EMULATE_FULL_STEPS steps trace the whole pool), so the apply does a frame's worth of inserts and the hash stays a converged one.
    gi_emulate = gi_mode and bool(emulate)
    EMULATE_FULL_STEPS = 24
    if gi_bands:  # one frame, row bands; the surfel TRACE sharded over the ranks too (native communicator), ordering + apply replicated
        ex = pipe.gi_exchange(world * per_rows)
        ex_owner, ex_touched, ex_merged = be.alias_exchange(ex)
    if gi_emulate:
        pipe.gi_exchange(en * per_rows)
        emu_comm = be.api.Comm.local(lanes[0].ctx, 1)[0]   # a loopback group of one: export + import, no reduction

    # Framebuffer gather: two illuminance targets in torch tensors, bound to the pipeline in turn (dust_hip_pipeline_bind_plane),
    # so RCCL moves frame k straight out of its render target while frame k+1 renders into the other one -- no staging copy.
    # With bands the targets are padded to world * per_rows rows: every rank sends a slice of the SAME size (its band padded to
    # per_rows rows; a collective with unequal counts is undefined), and rank 0 keeps the first H rows of the assembly.
    # Whole frames per rank (samples) are assembled by row slices (--assemble slices, the default): one all-to-all per step,
    # rank j receives slice j of every rank's frame over N-1 links at once; the target is padded to a multiple of N rows.
    # --assemble rotate / fixed gather whole frames onto rank k % N / rank 0 instead (bands always gather: a band is a slice).
    assemble = args.assemble if not bands else ("rotate" if args.assemble == "slices" else args.assemble)
    slices = assemble == "slices" and world > 1
    tgt_rows = max(world * per_rows, send[1], (H + per_rows) if band_cuts else 0) if bands else (-(-H // world) * world if slices else H)
    # Slots: step k renders into target k % S on lane (k % S) % D -- two targets per pipeline at least, so that a gather never
    # reads what the next frame of the same pipeline writes; with D > 1 frames in flight every lane has its own stream, and a
    # rank whose launch is held up by one expensive tile fills the rest of the GPU with the next frames' tiles.
    S = max(2, D)
    fixed_targets = S == D  # every lane renders into one target of its own: bound once, not per step
    targets = [torch.zeros((tgt_rows, W, 4), dtype=torch.float16, device=be.device) for _ in range(S)]
    if slices:
        send = (0, tgt_rows)
    # step k's exchange overlaps step k+1's rendering
    native = bands and world > 1 and getattr(args, "comm", "torch") == "native" and hasattr(be, "make_comm")
    if native:
        try:
            for lane in lanes:
                if getattr(lane, "comm", None) is None:
                    lane.comm = be.make_comm(dist, lane.ctx)
        except Exception as e:  # noqa: BLE001 -- e.g. no librccl: the torch path still works
            sys.stderr.write(f"bench.py: native communicator unavailable ({e}); using torch.distributed\n")
            native = False
        agreed = torch.tensor([1 if native else 0], dtype=torch.int64, device=be.device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MIN)   # (every rank takes the same path)
        native = bool(agreed.item())
    if native:
        row_cuts = band_cuts if band_cuts else [min(H, r * per_rows) for r in range(world)] + [H]
        # --denoise: the root keeps the filter's history, so it stays rank 0; every plane the filter and the tone map read travels
        dn_planes = (L.PLANE_ILLUMINANCE, L.PLANE_DEPTH, L.PLANE_NORMAL, L.PLANE_MOTION, L.PLANE_VOXEL_ID, L.PLANE_DENOISED, L.PLANE_ALBEDO) if denoise else None
        gather = NativeGather([lane.comm for lane in lanes], [lane.pipe for lane in lanes], L.PLANE_ILLUMINANCE, row_cuts, S,
                              rotate=assemble == "rotate" and not denoise, world=world, planes=dn_planes)
    else:
        gather = sharding.AsyncGather(dist, targets[0][send[0]:send[0] + (per_rows if bands else send[1] - send[0])], depth=S, rotate=assemble == "rotate", slices=slices)
    pix_stats = []
    if fixed_targets:
        for s_, lane in enumerate(lanes):
            be.bind_target(lane.pipe, targets[s_])

    def step(k, count=False):
        lane = lanes[(k % S) % D]
        if world > 1 and D > 1:
            with lane.enter():  # the collective below is ordered against the lane's stream
                step_on(lane.pipe, lane.sc["scene"], k, count)
        else:
            step_on(lane.pipe, lane.sc["scene"], k, count)

    def step_on(pipe, scene, k, count):
        cs = L.PASS_COUNT_STATS if count else 0
        gather.wait_slot(k % S)  # the gather that last read this target is done
        if not fixed_targets:
            be.bind_target(pipe, targets[k % S])
        if gi_bands or gi_emulate:
            frame_index = 1 + k  # every rank works on the same frame
            rnd = synth.frame_rand(1, frame_index)
            pix = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_GI_SHARDED
            if have_rows:
                pipe.render(scene, cam, sky, pix | cs, frame_index=frame_index, rand=rnd, rows=rows)
            if count:  # the second call restarts the counters: keep the pixel passes' now
                be.sync()
                pix_stats[:] = [pipe.pass_stats(i) for i in range(4)] if have_rows else []
            if gi_emulate:
                emu_comm.gi_exchange(pipe, rows[0] if have_rows else H, rows[1] if have_rows else H, en * per_rows, frame_index)
            elif native:   # the same five steps inside the library, on the context's stream
                lanes[0].comm.gi_exchange(pipe, rows[0] if have_rows else H, rows[1] if have_rows else H, per_rows, frame_index)
            else:
                sharding.gi_exchange_step(dist, rank, world, ex_owner, ex_touched, ex_merged, per_rows * W,
                                          (lambda: pipe.gi_export(*rows)) if have_rows else (lambda: pipe.gi_export(H, H)),  # (no rows: zeroes into the merge)
                                          (lambda: pipe.gi_import(rows[0], rows[1], frame_index)) if have_rows else
                                          (lambda: pipe.gi_import(H, H, frame_index)))  # empty own range: every stamp is another band's
            # the replicated surfel pass must leave the SAME hash on every GPU: deterministic apply, not the racy one
            sp = L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_GI_SHARDED | cs | (L.PASS_ACCUMULATE if have_rows else 0)
            # the surfel pass: its TRACE sharded over the ranks where the library's communicator carries the records (an all-gather of 64 B per
            # slot), its ordering and ordered apply replicated; through torch.distributed: the whole pass replicated (round 5's shape)
            shard = ((er, en) if k >= EMULATE_FULL_STEPS else (0, 1)) if gi_emulate else ((rank, world) if native else (0, 0))
            if shard[1]:
                pipe.render(scene, cam, sky, sp, frame_index=frame_index, rand=rnd, rows=rows if have_rows else (0, 0), surfel_shard=shard)
                if gi_emulate:
                    pipe.gi_surfel_finish(frame_index)
                else:
                    lanes[0].comm.gi_surfel_exchange(pipe, frame_index)
            else:
                pipe.render(scene, cam, sky, sp, frame_index=frame_index, rand=rnd, rows=rows if have_rows else (0, 0))
        elif bands:
            frame_index = 1 + k
            if have_rows:
                pipe.render(scene, cam, sky, passes | cs, frame_index=frame_index, rand=synth.frame_rand(1, frame_index), rows=rows)
        else:
            frame_index = sharding.sample_frame_index(k, rank, world)  # sample k*N + r of the spp sequence
            pipe.render(scene, cam, sky, passes | cs, frame_index=frame_index, rand=synth.frame_rand(1, frame_index))
        if world > 1 and native:
            gather.submit_slot(k % S, k)   # grouped send / receive of the bands' rows, in place in the root's target, on the communicator's stream
            if denoise and rank == 0:      # the root filters the gathered frame (its stream waits for the gather on the device, not the host)
                gather.wait_slot(k % S)
                pipe.render(scene, cam, sky, L.PASS_DENOISE, frame_index=frame_index, rand=synth.frame_rand(1, frame_index))
        elif world > 1:
            gather.submit_view(targets[k % S][send[0]:send[1]])  # asynchronous gather, straight from the target

    if batched:   # the cameras and skies of a call, as the arrays the C entry point takes: made once (one view, one sky)
        cam_arr, sky_arr = (L.Camera * D)(*([cam] * D)), (L.Sky * D)(*([sky] * D))

    def plan_steps(first, n):
        """--frames-per-launch: the dust_hip_render_frames calls of steps first .. first + n - 1, arguments marshalled (a host that knows its frames ahead
        has them ready: what a call then costs is the C entry point). D frames per call; when D does not divide n the SHORT launch comes first -- the
        device has work after the fewest preparations, and the region ends on whole launches."""
        calls, i = [], 0
        while i < n:
            m = (n % D) if (i == 0 and n % D) else min(D, n - i)
            ks = [first + i + j for j in range(m)]
            idx = [(1 + k) if bands else sharding.sample_frame_index(k, rank, world) for k in ks]
            if have_rows:
                calls.append(be.api.StandardPipeline.frames_call([lanes[(k % S) % D].pipe for k in ks], lanes[0].sc["scene"], cam_arr, sky_arr, passes, idx,
                                                                 [synth.frame_rand(1, v) for v in idx], rows=rows if bands else (0, 0)))
            i += m
        return calls

    def steps_from(first, n, planned=None):
        """steps first .. first + n - 1: one launch each, or (--frames-per-launch) D of them per dust_hip_render_frames call"""
        if not batched:
            for i in range(n):
                step(first + i)
            return
        for call in (planned if planned is not None else plan_steps(first, n)):
            call()

    def barrier():
        gather.finish()
        if world > 1:
            dist.barrier()
        be.sync()

    # untimed counting frame: rays per class and algorithmic bytes per launch
    step(0, count=True)
    barrier()
    n_classes = 6 if gi_mode else 3
    st = [pipe.pass_stats(i) for i in range(n_classes)]
    if gi_bands or gi_emulate:
        st[:4] = pix_stats if pix_stats else [L.PassStats() for _ in range(4)]
    if bands and not have_rows:
        st[:min(4, n_classes)] = [L.PassStats() for _ in range(min(4, n_classes))]
    rays_rank = sum(x.rays for x in st)
    if gi_bands and rank != 0:
        rays_rank -= st[4].rays + st[5].rays  # the replicated surfel pass counts once
    if have_rows:
        be.check_target(pipe, targets[0], rows)
    if native and rank == 0:   # step 0 gathered onto rank 0, in place: every band's rows arrived in its target
        for r in range(world):
            assert row_cuts[r] == row_cuts[r + 1] or bool((targets[0][row_cuts[r]:row_cuts[r + 1]] != 0).any()), f"band {r} did not arrive"

    gc.collect()
    gc.disable()  # no collector pause between two launches of the timed loop -- nor between the settle frames and the timed region
    # Settle frames. Every step holds a collective when world > 1, so every rank must run the SAME number of them: a count from
    # a rank's own clock would leave the ranks a few steps apart and the job hung in its gather. The fixed part first; then the
    # ranks agree (max) on how many more make up SETTLE_SECONDS at the rate the slowest of them measured.
    settle = 0
    t_settle = time.perf_counter()
    n_fixed = max(args.warmup, SETTLE_STEPS)
    steps_from(1, n_fixed)
    settle += n_fixed
    barrier()
    spent = time.perf_counter() - t_settle
    more = 0 if spent >= SETTLE_SECONDS else min(SETTLE_MAX_STEPS, int((SETTLE_SECONDS - spent) / max(spent / max(settle, 1), 1e-6)) + 1)
    if world > 1:
        agreed = torch.tensor([more], dtype=torch.int64, device=be.device)
        dist.all_reduce(agreed, op=dist.ReduceOp.MAX)
        more = int(agreed.item())
    if batched:
        more = -(-more // D) * D   # whole launches
    steps_from(1 + settle, more)
    settle += more
    barrier()
    for lane in lanes:
        lane.pipe.mark_kernel_times()  # kernel durations: the HIP-event pairs the library records around its launches FROM HERE (no wait) ...
    planned = plan_steps(1 + settle, args.steps) if batched else None   # (argument marshalling only: nothing is enqueued before t_start)
    t_start = time.perf_counter()
    steps_from(1 + settle, args.steps, planned)   # (EXACTLY args.steps frames: a FIRST launch of fewer frames if D does not divide them)
    barrier()
    elapsed = time.perf_counter() - t_start
    gc.enable()
    # ... TO HERE: the timed region's own launches, on the launch stream, read back after it (a ring of 256 pairs per pass
    # kind: with more steps than that, the last 256); nothing was synchronised per step
    ev_ms, ev_n = [0.0] * 4, [0] * 4
    for lane in lanes:
        lm, ln = lane.pipe.kernel_times(mark=True)
        ev_ms, ev_n = [a_ + b_ for a_, b_ in zip(ev_ms, lm)], [a_ + b_ for a_, b_ in zip(ev_n, ln)]
    ms = [(ev_ms[k] / ev_n[k]) if ev_n[k] else 0.0 for k in range(4)]
    if not gi_mode:
        ms[2] = ms[3] = 0.0
    # --frames-per-launch: did the frames share launches? (dust_hip_render_frames enqueues frames that do not qualify one after the other, with the
    # same results: a launch of D frames lasts about D steps, a frame's own launch one.) If they did not, the line says so and accounts per frame.
    launch_frames = (args.steps / -(-args.steps // D)) if batched else 1.0
    # (threshold: the geometric mean of the two cases' ratios, 1 and 1 / D -- a host stall of several milliseconds inside the timed region, which
    #  stretches ms_per_step, does not flip it)
    shared_launches = batched and ms[0] > (1.0 / D) ** 0.5 * launch_frames * (elapsed / args.steps * 1e3)
    if batched and not shared_launches:
        sys.stderr.write("bench.py: --frames-per-launch: the frames did not share a launch (kernel time of ONE frame); accounted per frame\n")
        launch_frames = 1.0

    t_max = torch.tensor([elapsed], dtype=torch.float64, device=be.device)
    rays_all = torch.tensor([float(rays_rank)], dtype=torch.float64, device=be.device)
    seen = torch.tensor([1.0], dtype=torch.float64, device=be.device)
    mine = torch.tensor(ms, dtype=torch.float64, device=be.device)
    per_rank = [mine]
    if world > 1:
        dist.all_reduce(t_max, op=dist.ReduceOp.MAX)
        dist.all_reduce(rays_all, op=dist.ReduceOp.SUM)
        dist.all_reduce(seen, op=dist.ReduceOp.SUM)
        per_rank = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(per_rank, mine)
    elapsed = float(t_max.item())
    rays = float(rays_all.item())
    return {"shard": shard, "scaling": "strong" if bands else "weak", "elapsed": elapsed, "rays_per_step": rays,
            "ms_per_step": elapsed / args.steps * 1e3, "mrays": rays * args.steps / elapsed / 1e6, "ranks_seen": int(seen.item()),
            "st": st, "ms": ms, "launches": max(ev_n) if ev_n else 0, "per_rank": [[float(x) for x in v.tolist()] for v in per_rank],
            "frames_per_launch": D if (batched and shared_launches) else 1,
            # frames per launch of the timed region, averaged (a last launch of fewer frames counts): what a launch's algorithmic bytes are a multiple of
            "launch_frames": launch_frames,
            "per_rows": per_rows, "assemble": assemble, "slices": slices, "settle": settle, "frames_in_flight": 1 if batched else D, "in_flight_slots": slots_mode if D > 1 else None, "band_cuts": band_cuts,
            "comm": "native" if native else "torch", "denoise": denoise, "band_balance": balance_log or None,
            "band_steps_ms": [round(v, 4) for v in band_steps_ms] if band_steps_ms else None, "emulated_band": (f"{er}/{en}" if emulate else None)}


def compact(curve, gi_mode):
    """A measured curve as the short record the default line carries under `curves` (same accounting as the headline)."""
    acct = account(curve, gi_mode)
    d = acct["dominant"]
    ms = curve["ms"]
    fused = ms[1] == 0.0
    kernels = dict({("k_primary_ao_batch" if curve.get("frames_per_launch", 1) > 1 else "k_primary_ao"): round(ms[0], 4)} if fused else {"k_primary": round(ms[0], 4), "k_ambient_occlusion": round(ms[1], 4)},
                   **acct["kernels_ms_extra"])
    return {"value": round(curve["mrays"], 2), "unit": "Mrays/s", "ms_per_step": round(curve["ms_per_step"], 4),
            "rays_per_step": int(curve["rays_per_step"]), "settle_steps": curve["settle"], "kernels_ms": kernels,
            "roofline": {"kernel": "k_" + d[0], "achieved": round(acct["achieved"], 3), "unit": "GB/s", "frac": round(acct["achieved"] / HBM_PEAK_GBPS, 6),
                         "algorithmic_bytes_per_launch": int(d[1]), "kernel_ms": round(d[2], 4)}}


def measure_moving(be, args, lane, noise5, steps, settle=48, fps=60.0, swing=0.15, period=4.0, fpl=1):
    """The reference is a real-time renderer: an FPS camera and a teapot that swings (examples/castle.rs:105-130,287-291). Here the
    eye sways along the circle round the castle that the reference's start position lies on -- angle `swing` * sin(2 pi t / `period`)
    either side of it, 60 frames a second of scene time, never at rest --, the teapot follows
    Transform::from_translation((sin t * 50, 200, 0)) -- set_transform + commit every frame, motion vectors against the previous
    frame's transform. One GPU; the castle's instances plus the teapot. (A full orbit shows a different castle every second: its
    frames cost up to 20 % more or less than the headline's view for reasons that have nothing to do with moving; the sway keeps the
    comparison with the still view a like-for-like one, and `still_same_views` times three of its cameras standing still.)
    fpl > 1: that many consecutive frames of the motion per call and per launch (dust_hip_render_frames with each frame's moves: the library
    commits the teapot's transform before preparing the frame that sees it -- a scene image per frame in the scene's ring)."""
    import math
    import numpy as np
    api, L, synth, P = be.api, be.L, be.synth, be.P
    W, H = args.width, args.height
    ctx, base = lane.ctx, lane.sc
    tea_bytes, tea_real = be.assets.teapot()
    tdesc = P.SceneDesc.from_vox(tea_bytes)
    tea_model = api.Model(ctx, tdesc.models[0][0], tdesc.models[0][1], tdesc.palette)
    scene = api.Scene(ctx)
    for model, (_, t) in zip(base["scene"]._models, base["desc"].instances):
        scene.add_instance(model, t)
    home = np.asarray(tdesc.instances[0][1], np.float32).reshape(3, 4)

    def tea_xf(t):
        m = home.copy()
        m[:, 3] += np.array([math.sin(t) * 50.0, 200.0, 0.0], np.float32)   # castle.rs:287-291
        return m

    def cols(m34):   # 3x4 row-major -> mat4 column-major (the `instances[]` entry of layout.playout:72)
        m = np.eye(4, dtype=np.float32)
        m[:3, :] = m34
        return np.ascontiguousarray(m.T).reshape(16)
    tea = scene.add_instance(tea_model, tea_xf(0.0).reshape(12))
    scene.commit()
    pipe = api.StandardPipeline(ctx, W, H)
    pipe.set_noise(5, noise5)
    pipes = [pipe]
    for _ in range(max(1, int(fpl)) - 1):
        pipes.append(api.StandardPipeline(ctx, W, H))
        pipes[-1].set_noise(5, noise5)
    fpl = len(pipes)
    steps = -(-steps // fpl) * fpl   # whole launches
    settle = -(-settle // fpl) * fpl
    sky = be.sky_struct(base["sky"])
    sc = args.scale
    eye0 = (122.0 * sc, 300.61 * sc, 54.45 * sc)
    radius, th0 = math.hypot(eye0[0], eye0[2]), math.atan2(eye0[2], eye0[0])
    n = settle + steps

    def cam_at(th):
        eye = (radius * math.cos(th), eye0[1], radius * math.sin(th))
        return api.make_camera(eye, api.look_at_rotation(eye, (0.0, 0.0, 0.0)), api.PinholeProjection())
    angle = lambda k: th0 + swing * math.sin(2.0 * math.pi * (k / fps) / period)   # noqa: E731
    cams, xfs = [], []
    for k in range(n + 1):
        cams.append(cam_at(angle(k)))
        xfs.append(tea_xf(k / fps))
    prevs = [cols(xfs[max(k - 1, 0)]) for k in range(n + 1)]
    flat = [np.ascontiguousarray(x.reshape(12)) for x in xfs]
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION

    def frame(k, count=False):
        scene.set_transform(tea, flat[k], prevs[k])
        scene.commit()
        pipe.render(scene, cams[k], sky, passes | (L.PASS_COUNT_STATS if count else 0), frame_index=1 + k, rand=synth.frame_rand(1, 1 + k))

    def stretch(first, last):
        """frames first .. last - 1: a launch per frame, or fpl of them per call"""
        if fpl == 1:
            for k in range(first, last):
                frame(k)
            return
        for k in range(first, last, fpl):
            ks = list(range(k, min(k + fpl, last)))
            api.StandardPipeline.render_frames(pipes[:len(ks)], scene, [cams[j] for j in ks], sky, passes, [1 + j for j in ks],
                                               [synth.frame_rand(1, 1 + j) for j in ks], moves=[[(tea, flat[j], prevs[j])] for j in ks])
    # untimed: the rays of exactly the frames that are timed below (they differ from frame to frame)
    rays = 0
    algo = 0   # algorithmic bytes of the same frames: every frame of a moving view has its own counts
    for k in range(settle, n):
        frame(k, count=True)
        be.sync()
        st = [pipe.pass_stats(i) for i in range(3)]
        rays += sum(x.rays for x in st)
        hit_px, miss_px = st[0].hits, st[0].rays - st[0].hits
        algo += (algorithmic_bytes(st[0], hit_px * 32 + miss_px * 24) + algorithmic_bytes(st[1], 0) + algorithmic_bytes(st[2], 0)
                 - (st[1].hits + st[2].hits) * 5 + hit_px * 24)
    pipe.clear()
    gc.collect()
    gc.disable()
    # The stretch is timed three times (settle frames, then the timed ones, each time) and the MEDIAN pass is reported: a frame loop that
    # commits every frame can be at most a ring of 8 scene images (under 2 ms of GPU work) ahead of the GPU, so ONE multi-millisecond stall
    # of the host process -- seen on these boxes about once in ten runs -- drains the queue and shows up in a 5 ms timed region as a several
    # times slower step; the median of three drops such a pass without favouring the fastest (round 5 took the faster of two: a bias the
    # other curves, single passes, do not have). All passes are listed in `passes_ms_per_step`.
    runs = []
    for _ in range(3):
        stretch(0, settle)
        be.sync()
        pipe.mark_kernel_times()
        t0 = time.perf_counter()
        stretch(settle, n)
        be.sync()
        dt_i = time.perf_counter() - t0
        runs.append((dt_i,) + tuple(pipe.kernel_times(mark=True)))
    gc.enable()
    dt, lm, ln = sorted(runs, key=lambda r: r[0])[1]
    # the same views standing still: three cameras of the timed stretch (first, middle, last), the teapot where it was, 40 + 40 frames each
    still_ms = []
    for k in (settle, settle + steps // 2, n - 1):
        scene.set_transform(tea, flat[k], prevs[k])
        scene.commit()
        for j in range(0, 80, fpl):
            if j == 40:
                be.sync()
                t1 = time.perf_counter()
            if fpl == 1:
                pipe.render(scene, cams[k], sky, passes, frame_index=1 + j, rand=synth.frame_rand(1, 1 + j))
            else:   # (the still views at the same number of frames per launch)
                api.StandardPipeline.render_frames(pipes, scene, cams[k], sky, passes, [1 + j + i for i in range(fpl)], [synth.frame_rand(1, 1 + j + i) for i in range(fpl)])
        be.sync()
        still_ms.append((time.perf_counter() - t1) / 40 * 1e3)
    speed = swing * 2.0 * math.pi / period
    k_ms = lm[0] / ln[0] if ln[0] else None
    if fpl > 1 and k_ms and k_ms < (1.0 / fpl) ** 0.5 * fpl * (dt / steps * 1e3):   # (the frames did not share launches: accounted per frame)
        fpl = 1
    achieved = (algo / steps * fpl) / (k_ms * 1e-3) / 1e9 if k_ms else None
    kname = "k_primary_ao_batch" if fpl > 1 else "k_primary_ao"
    return {"value": round(rays / dt / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(dt / steps * 1e3, 4), "steps": steps, "frames_per_launch": fpl,
            "rays_per_step": int(rays / steps), "settle_steps": settle, "passes_ms_per_step": [round(r[0] / steps * 1e3, 4) for r in runs],
            "kernels_ms": {kname: round(k_ms, 4) if k_ms else None},
            "roofline": {"bound": "hbm", "kernel": kname, "algorithmic_bytes_per_launch": int(algo / steps * fpl), "kernel_ms": round(k_ms, 4) if k_ms else None,
                         "achieved": round(achieved, 3) if achieved else None, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBPS, 6) if achieved else None,
                         "note": "mean algorithmic bytes of the timed frames (each counted in an untimed pass of its own) over the mean kernel time of the "
                                 "launches that were bracketed by HIP events (every 4th frame)"},
            "still_same_views": {"ms_per_step": [round(v, 4) for v in still_ms], "mean_ms_per_step": round(sum(still_ms) / len(still_ms), 4),
                                 "moving_over_still": round((dt / steps * 1e3) / (sum(still_ms) / len(still_ms)), 4)},
            "camera": f"sway: eye on the circle of radius {radius:.1f} at height {eye0[1]:.1f} round the origin, {swing} rad either side of the reference's "
                      f"start position with a period of {period} s at {fps:.0f} frames/s (up to {math.degrees(speed / fps):.3f} deg per frame), looking at the origin",
            "scene": f"the castle's {len(base['desc'].instances)} instances + teapot.vox {'(reference asset)' if tea_real else 'stand-in'} at "
                     "(sin t * 50, 200, 0), dust_hip_scene_set_transform + dust_hip_scene_commit every frame (castle.rs:287-291)",
            "instances": len(base["desc"].instances) + 1}


def extra_curves(args, be, noise0, noise5, base):
    """One GPU, default workload: short runs after the headline's timed region, so that the driver's single command also carries the
    moving view, the GI frame, the 4K frame and the deep tree (BASELINE configs[2], [3]'s size, [4]). Each is its own pipeline and
    timed region; none of them touches `value`. A failure is reported in its place, never raised."""
    out = {}

    def run(name, fn):
        try:
            out[name] = fn()
        except Exception as e:  # noqa: BLE001 -- the headline line must still be printed
            out[name] = {"error": f"{type(e).__name__}: {e}"}
        gc.enable()
        gc.collect()

    def curve(workload, W, H, sc, fpl=1):
        """fpl > 1 (primary + AO workloads): that many frames per launch, like the headline (dust_hip_render_frames)"""
        a = argparse.Namespace(**vars(args))
        a.workload, a.width, a.height, a.steps, a.warmup, a.frames_per_launch = workload, W, H, -(-args.extra_steps // fpl) * fpl, 0, fpl
        lanes = []
        for _ in range(fpl):
            lane = Lane()
            lane.stream, lane.ctx, lane.sc = base.stream, base.ctx, sc
            lane.pipe = be.api.StandardPipeline(base.ctx, W, H)
            lane.pipe.set_noise(5, noise5)
            lane.enter = base.enter
            lanes.append(lane)
        gi = workload != "primary_ao"
        if gi:
            lanes[0].pipe.set_noise(0, noise0)
        c = measure_curve(be, None, a, lanes, "bands")
        rec = compact(c, gi)
        rec["steps"], rec["frame"], rec["frames_per_launch"] = a.steps, [W, H], c.get("frames_per_launch", 1)
        return rec

    def deep():
        a = argparse.Namespace(**vars(args))
        a.workload = "deep"
        sc = be.build_scene(a)
        rec = curve("deep", args.width, args.height, sc)
        rec["scene"] = f"procedural 4096^3 tree, {args.deep_occupancy:.2%} brick occupancy, {sc['n_bricks']} bricks (BASELINE configs[4] on one GPU)"
        return rec
    def many_instances():
        a = argparse.Namespace(**vars(args))
        a.props = 4000
        sc = be.build_scene(a)
        rec = curve("primary_ao", args.width, args.height, sc, fpl=8)
        rec["scene"] = f"the castle + 4000 scattered props: {sc['info']['n_instances']} instances of {sc['info']['n_models']} models"
        return rec
    def pipelined(depth):
        """The headline's frames, `depth` of them in flight on streams, contexts and G-buffers of their own, each launch on 1 / depth of the
        workgroup slots (the reference's host keeps up to three frames in flight, rhyolite_bevy/src/lib.rs:58): a launch's staging, the gap
        behind it and half of its tail are hidden behind its neighbour. (Every launch asking for ALL slots -- the next frame's workgroups
        taking the CUs the previous frame's tail leaves -- measured worse: docs/EXPERIMENTS.md, round 6.)
        Per-kernel HIP-event times are inflated by the overlap: the roofline here is algorithmic bytes per STEP over ms_per_step."""
        a = argparse.Namespace(**vars(args))
        a.steps, a.warmup, a.in_flight_slots = max(args.extra_steps, 60), 0, "share"
        lanes = []
        for i in range(depth):
            lane = be.open_lane(a, base.sc)
            lane.pipe.set_noise(5, noise5)
            lanes.append(lane)
        c = measure_curve(be, None, a, lanes, "bands")
        acct = account(c, False)
        step_bytes = acct["bytes_primary"] + acct["bytes_ao"]
        achieved = step_bytes / (c["ms_per_step"] * 1e-3) / 1e9
        return {"value": round(c["mrays"], 2), "unit": "Mrays/s", "ms_per_step": round(c["ms_per_step"], 4), "steps": a.steps, "frames_in_flight": depth,
                "in_flight_slots": c["in_flight_slots"], "rays_per_step": int(c["rays_per_step"]), "settle_steps": c["settle"],
                "kernels_ms": {"k_primary_ao": round(c["ms"][0], 4)},
                "roofline": {"bound": "hbm", "kernel": "k_primary_ao", "algorithmic_bytes_per_step": int(step_bytes), "achieved": round(achieved, 3), "unit": "GB/s",
                             "peak": HBM_PEAK_GBPS, "frac": round(achieved / HBM_PEAK_GBPS, 6),
                             "note": "bytes per step / ms_per_step (whole-step rate): with launches overlapping, the HIP-event duration of one launch "
                                     "(kernels_ms) includes the time it shares the device with its neighbours' and is not what a step costs"}}
    def batched(fpl):
        """The headline's frames, `fpl` consecutive ones per call and per persistent LAUNCH (dust_hip_render_frames / k_primary_ao_batch): G-buffers of
        their own on the headline's context, stream and scene; a wavefront that finds frame i without tiles takes frame i + 1's descriptor and queue,
        so a launch's tail, root staging and the gap behind it are paid once for `fpl` frames. One launch at a time: its HIP-event duration is what
        its frames cost, and the roofline is the launch's own (fpl frames' algorithmic bytes over that duration)."""
        a = argparse.Namespace(**vars(args))
        a.steps, a.warmup, a.frames_per_launch = -(-max(args.extra_steps, 60) // fpl) * fpl, 0, fpl
        lanes = [base]
        for _ in range(fpl - 1):
            lane = be.open_batch_lane(a, base)
            lane.pipe.set_noise(5, noise5)
            lanes.append(lane)
        c = measure_curve(be, None, a, lanes, "bands")
        rec = compact(c, False)
        rec["steps"], rec["frames_per_launch"] = a.steps, c["frames_per_launch"]
        step_bytes = rec["roofline"]["algorithmic_bytes_per_launch"] / c["launch_frames"]
        rec["roofline"]["whole_step_frac"] = round(step_bytes / (c["ms_per_step"] * 1e-3) / 1e9 / HBM_PEAK_GBPS, 6)   # bytes per step / ms_per_step: gaps between launches included
        return rec
    def one_frame_per_launch():
        """rounds 1-5's headline: every frame a launch of its own (dust_hip_render_frame), one at a time"""
        a = argparse.Namespace(**vars(args))
        a.steps, a.warmup, a.frames_per_launch = max(args.extra_steps, 60), 0, 1
        rec = compact(measure_curve(be, None, a, [base], "bands"), False)
        rec["steps"], rec["frames_per_launch"] = a.steps, 1
        return rec
    run("one_frame_per_launch", one_frame_per_launch)
    run("four_frames_per_launch", lambda: batched(4))
    run("pipelined", lambda: pipelined(2))
    run("moving", lambda: measure_moving(be, args, base, noise5, max(args.extra_steps, 20)))
    run("moving_four_frames_per_launch", lambda: measure_moving(be, args, base, noise5, max(args.extra_steps, 20), fpl=4))
    run("primary_ao_4k", lambda: curve("primary_ao", 3840, 2160, base.sc, fpl=8))
    run("gi_1080p", lambda: curve("gi", args.width, args.height, base.sc))
    run("deep", deep)
    if not getattr(args, "props", 0):
        run("many_instances", many_instances)
    return out


def account(curve, gi_mode):
    """Algorithmic bytes per launch of every kernel of a measured curve, the dominant kernel and its achieved GB/s."""
    st, ms = curve["st"], curve["ms"]
    ms_primary, ms_ao, ms_fg, ms_sf = ms
    hit_px = st[0].hits
    miss_px = st[0].rays - st[0].hits
    bytes_primary = algorithmic_bytes(st[0], hit_px * 32 + miss_px * 24)
    # AO kernel: per live pixel read depth 4 + normal 4 + illuminance 8, write illuminance 8 (hit.rchit/ao.rgen)
    bytes_ao = algorithmic_bytes(st[1], 0) + algorithmic_bytes(st[2], 0) - (st[1].hits + st[2].hits) * 5 + hit_px * 24
    kernels_ms_extra = {}
    bytes_fg = bytes_sf = 0
    if gi_mode:
        # final gather: depth 4 + normal 4 + illuminance 8 read, 8 written, one 12-byte hash entry + 16-byte surfel per hit
        bytes_fg = algorithmic_bytes(st[3], st[3].rays * 24 + st[3].hits * 28) - st[3].hits * 5
        # surfel pass: 16-byte surfel read, 32-byte request + 16-byte replacement written, one hash entry read+write per surfel
        bytes_sf = (algorithmic_bytes(st[4], 0) + algorithmic_bytes(st[5], st[5].rays * (16 + 48 + 24)) - (st[4].hits + st[5].hits) * 5)
        kernels_ms_extra = {"k_final_gather": round(ms_fg, 4), "k_surfel_trace+apply": round(ms_sf, 4)}
    if gi_mode and max(ms_fg, ms_sf) > ms_primary:
        dominant = ("final_gather", bytes_fg, ms_fg) if ms_fg >= ms_sf else ("surfel_trace", bytes_sf, ms_sf)
    elif ms_ao == 0.0:   # primary + AO ran as one fused kernel (the default)
        lf = float(curve.get("launch_frames", 1.0))   # (--frames-per-launch: one launch carries that many frames' bytes)
        dominant = ("primary_ao_batch" if lf > 1.0 else "primary_ao", (bytes_primary + bytes_ao) * lf, ms_primary)
    else:
        dominant = ("ambient_occlusion", bytes_ao, ms_ao) if ms_ao >= ms_primary else ("primary", bytes_primary, ms_primary)
    achieved = dominant[1] / (dominant[2] * 1e-3) / 1e9 if dominant[2] > 0 else 0.0
    return {"bytes_primary": bytes_primary, "bytes_ao": bytes_ao, "bytes_fg": bytes_fg, "bytes_sf": bytes_sf, "dominant": dominant,
            "achieved": achieved, "kernels_ms_extra": kernels_ms_extra}


def run_rank(args, be, dist):
    """Everything one rank does. Rank 0 returns the JSON line's dict, the others None."""
    rank, world = be.rank, be.world
    W, H = args.width, args.height
    gi_mode = args.workload in ("gi", "deep")
    deep = args.workload == "deep"
    noise0, noise5 = be.noise()
    lanes = []

    def lanes_for(n):
        while len(lanes) < n:
            lane = be.open_lane(args, lanes[0].sc if lanes else None)
            lane.pipe.set_noise(5, noise5)
            if gi_mode:
                lane.pipe.set_noise(0, noise0)
            lanes.append(lane)
        return lanes[:n]

    sc, pipe = lanes_for(1)[0].sc, lanes[0].pipe

    if world == 1:
        wanted = ["bands" if args.shard in ("both", "bands") else "samples"]   # one GPU: the two partitions are the same frame
    else:
        wanted = ["bands", "samples"] if args.shard == "both" else [args.shard]
    curves = {}
    for shard in wanted:
        if len(curves) and gi_mode:  # a second curve starts from a fresh hash, like the first
            pipe.configure_gi(*(getattr(pipe, "_gi", None) or (32 * 1024 * 1024, 720 * 480)))
            pipe.clear()
        in_flight = 1 if gi_mode else (args.frames_in_flight or (4 if (world > 1 and shard == "bands") else 1))
        fpl = args.frames_per_launch
        if fpl is None:   # the default: eight frames per launch where nothing else was asked for
            fpl = 8 if (world == 1 and not gi_mode and not args.frames_in_flight and not os.environ.get("DUST_BENCH_EMULATE_BAND")
                        and not getattr(args, "denoise", False) and args.camera != "orbit") else 1   # (a moving view commits the scene between frames: a launch per frame)
        if fpl > 1 and world == 1 and not gi_mode and hasattr(be, "open_batch_lane"):
            # --frames-per-launch: that many G-buffers on the first lane's context; consecutive steps share a launch (dust_hip_render_frames)
            batch_lanes = [lanes_for(1)[0]]
            for _ in range(min(fpl, 8) - 1):
                bl = be.open_batch_lane(args, batch_lanes[0])
                bl.pipe.set_noise(5, noise5)
                batch_lanes.append(bl)
            a_b = argparse.Namespace(**vars(args))
            a_b.frames_per_launch = len(batch_lanes)
            curves[shard] = measure_curve(be, dist, a_b, batch_lanes, shard)
            continue
        curves[shard] = measure_curve(be, dist, args, lanes_for(in_flight), shard)
    main_curve = curves[wanted[0]]
    if rank != 0:
        return None

    acct = account(main_curve, gi_mode)
    st, ms = main_curve["st"], main_curve["ms"]
    ms_primary, ms_ao, ms_fg, ms_sf = ms
    bytes_primary, bytes_ao, dominant, achieved, kernels_ms_extra = (acct["bytes_primary"], acct["bytes_ao"], acct["dominant"], acct["achieved"],
                                                                     acct["kernels_ms_extra"])
    # HBM-side traffic per launch: NOT measured by this process -- rocprofv3 counter passes need their own runs
    # (tools/profile_round.sh); the committed summary of the latest one is quoted with its provenance, or nothing is
    traffic, traffic_source = None, None
    pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
    if os.path.exists(pmc_path) and world == 1 and (W, H) == (1920, 1080) and not getattr(args, "props", 0):   # (the counter passes ran on the castle as it is)
        try:
            pm = json.load(open(pmc_path))
            key = "castle-standin" if not deep else "deep-tree"
            if pm.get("workload") == key and abs(pm.get("scale", 1.0) - args.scale) < 1e-9:
                traffic = pm.get("hbm_bytes_per_launch", {}).get(dominant[0])
                if dominant[0] == "primary_ao_batch" and main_curve.get("frames_per_launch") != pm.get("primary_ao_batch_frames"):
                    traffic = None   # (the counter passes' launches carried another number of frames)
            if deep:
                traffic = pm.get("deep", {}).get("hbm_bytes_per_launch", {}).get(dominant[0])
            if traffic is not None:
                traffic_source = (f"profiles/pmc_summary.json (round {pm.get('round')}: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE "
                                  "passes of this command, 2 x FETCH + WRITE per MI355X_MICROARCH.md; not measured in this run)")
        except Exception:
            traffic, traffic_source = None, None
    fused = ms_ao == 0.0
    roofline = {"bound": "hbm", "kernel": "k_" + dominant[0], "achieved": round(achieved, 3), "peak": HBM_PEAK_GBPS,
                "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 6), "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": int(dominant[1]), "kernel_ms": round(dominant[2], 4),
                # (--frames-per-launch: ONE launch renders that many frames = steps; its bytes and its duration are those of all of them)
                "frames_per_launch": main_curve.get("frames_per_launch", 1),
                "kernel_ms_per_frame": round(dominant[2] / float(main_curve.get("launch_frames", 1.0)), 4),
                "kernel_ms_source": f"HIP events on the launch stream around the launches of every 4th frame of the timed region (a launch of 4 or more frames: every launch; {main_curve['launches']} launches "
                                    "averaged; an event record costs the stream ~6 us, so bracketing every launch would take 5 % off the rate it measures)"
                                    + (f"; {main_curve['frames_in_flight']} frames in flight on streams of their own: a launch's duration includes the time it "
                                       "shares the GPU with its neighbours'" if main_curve["frames_in_flight"] > 1 else ""),
                "kernels_ms": dict({("k_primary_ao_batch" if main_curve.get("frames_per_launch", 1) > 1 else "k_primary_ao"): round(ms_primary, 4)} if fused else
                                   {"k_primary": round(ms_primary, 4), "k_ambient_occlusion": round(ms_ao, 4)}, **kernels_ms_extra),
                "per_rank_kernel_ms": [{"primary_ao" if fused else "primary": round(v[0], 4),
                                        **({"ambient_occlusion": round(v[1], 4)} if not fused else {}),
                                        **({"final_gather": round(v[2], 4), "surfel": round(v[3], 4)} if gi_mode else {})}
                                       for v in main_curve["per_rank"]],
                "bytes_per_ray": {"primary": round(bytes_primary / max(1, st[0].rays), 1),
                                  "ao_pass": round(bytes_ao / max(1, st[1].rays + st[2].rays), 1)}}

    cpu = None
    if not args.no_cpu_baseline and world == 1:
        cpu = cpu_baseline(args, sc, noise5, be.synth)
    what = {"primary_ao": "1spp primary+shadow+AO", "gi": "1 GI frame: primary+shadow+AO+final gather+surfel",
            "deep": "1 GI frame: primary+shadow+AO+final gather+surfel"}[args.workload]
    scene_name = (f"procedural 4096^3 tree, {args.deep_occupancy:.2%} brick occupancy (synth.procedural_deep_blocks seed 0xC5)" if deep else
                  ("castle.vox (the reference's asset, sha256 verified)" if sc.get("real_castle") else
                   ("castle.vox stand-in (synth.castle_scene seed 0xD057)" if args.scale == 1.0 else f"castle stand-in at scale {args.scale}")))

    def parallelism(c):
        root_txt = "k % N for step k (rotating root)" if c["assemble"] == "rotate" and world > 1 else "0"
        if c.get("frames_per_launch", 1) > 1:
            return (f"1 GPU, {c['frames_per_launch']} consecutive frames per persistent launch (dust_hip_render_frames: G-buffers of their own, one context and "
                    "stream; every frame's planes bit-identical to the frame rendered alone, tests/test_gpu_batch.py); a step is still ONE frame")
        if c["shard"] == "bands":
            return ((f"bands x{world}: one frame in {world} row bands of about equal measured cost, cut at rows {c['band_cuts']}" if c.get("band_cuts") else
                     f"bands x{world}: one frame in {world} row bands of {c['per_rows']} rows")
                    + (", identical hash + surfel pool on every GPU (all-reduce MAX of slot owners, all-gather of hash stamps, all-reduce SUM of "
                       "winning surfels, deterministic apply), " + ("surfel TRACE sharded by pool slots with an all-gather of its records, ordering + apply replicated"
                                                                    if c.get("comm") == "native" else "surfel pass replicated") if gi_mode and world > 1 else "")
                    + (", RCCL gather of the (equal-size, padded) bands to rank " + root_txt if world > 1 else ""))
        return (f"spp x{world}: one {W}x{H} sample per GPU, " +
                ("RCCL all-to-all of the RGBA16F frames by row slices (rank j assembles slice j of every sample)" if c["slices"] or world == 1
                 else f"RCCL gather of RGBA16F frames to rank {root_txt}"))
    metric = {"primary_ao": f"Mrays/s at {W}x{H} 1spp castle.vox (primary + sun-shadow + AO rays)",
              "gi": f"Mrays/s at {W}x{H} castle.vox, diffuse GI frame (primary, shadow, AO, final gather, surfel rays)",
              "deep": f"Mrays/s at {W}x{H} procedural 4096^3 sparse vdb, diffuse GI frame (deep-tree stress)"}[args.workload]
    info = sc["info"]
    bands_main = main_curve["shard"] == "bands"
    out = {
        "metric": metric, "value": round(main_curve["mrays"], 2), "unit": "Mrays/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(main_curve["ms_per_step"], 4), "higher_is_better": True, "scaling": main_curve["scaling"], "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{scene_name}, {W}x{H}{'' if bands_main else ' per GPU'}, {what}",
                   "frame": [W, H], "spp_per_step": 1 if bands_main else world, "parallelism": parallelism(main_curve),
                   "vox_models": info["n_models"], "instances": info["n_instances"], "voxels": info["n_voxels"],
                   "bricks": sc["n_bricks"], "scene_build_s": round(sc["t_load"], 3), "untimed_steps_before_timing": main_curve["settle"],
                   "frames_in_flight": main_curve["frames_in_flight"], "in_flight_slots": main_curve.get("in_flight_slots"), "denoise": bool(main_curve.get("denoise")),
                   "frames_per_launch": main_curve.get("frames_per_launch", 1),
                   **({"emulated_band": main_curve["emulated_band"]} if main_curve.get("emulated_band") else {}),
                   **({"band_steps_ms": main_curve["band_steps_ms"], "band_balance": main_curve["band_balance"]} if main_curve.get("band_steps_ms") else {}),
                   "rays_per_step": {n: int(x.rays) for n, x in zip(NAMES, st)}, "rays_per_step_all_gpus": int(main_curve["rays_per_step"])},
        "ranks_seen": main_curve["ranks_seen"],
        "curves": {c["scaling"]: {"shard": c["shard"], "value": round(c["mrays"], 2), "ms_per_step": round(c["ms_per_step"], 4),
                                  "rays_per_step_all_gpus": int(c["rays_per_step"]), "parallelism": parallelism(c), "frames_in_flight": c["frames_in_flight"], "denoise": bool(c.get("denoise")),
                                  **({"collectives": "libdust_hip.so's own RCCL path (dust_hip_gather_bands / dust_hip_gi_exchange_run)" if c.get("comm") == "native"
                                      else "torch.distributed (RCCL)"} if world > 1 else {}),
                                  "settle_steps": c["settle"], **({"band_rows": c["band_cuts"]} if c.get("band_cuts") else {}),
                                  "per_rank_kernel_ms": [[round(x, 4) for x in v] for v in c["per_rank"]]} for c in curves.values()},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    # The extra curves: their own timed regions AFTER the headline's, never part of `value`. The headline line is complete at this
    # point; should an extra curve hang, a watchdog prints it as it stands and ends the process (a line without the extras beats no line).
    extras = {}
    if world == 1 and args.workload == "primary_ao" and hasattr(be, "assets"):
        finished, said = threading.Event(), threading.Lock()

        def watchdog():
            if not finished.wait(args.extra_timeout) and said.acquire(blocking=False):
                out["curves"]["extra_curves_error"] = f"not finished after {args.extra_timeout:.0f} s: line printed without them"
                print(json.dumps(out), flush=True)
                os._exit(0)
        threading.Thread(target=watchdog, daemon=True).start()
        if args.camera == "orbit":
            extras["moving"] = measure_moving(be, args, lanes[0], noise5, args.steps)
        elif not args.no_extra_curves and (W, H) == (1920, 1080) and args.scale == 1.0:
            extras = extra_curves(args, be, noise0, noise5, lanes[0])
        finished.set()
        if not said.acquire(blocking=False):   # the watchdog is printing: let it
            time.sleep(3600)
        if "moving" in extras and "value" in extras["moving"]:
            extras["moving"]["vs_still"] = round(extras["moving"]["value"] / max(main_curve["mrays"], 1e-9), 4)
    out["curves"].update(extras)   # moving / primary_ao_4k / gi_1080p / deep
    if hasattr(be, "assets"):
        out["config"]["assets"] = be.assets.summary()
    if args.camera == "orbit" and "moving" in extras:   # --camera orbit: the moving view IS the line; the still view stays in curves.strong
        mv = extras["moving"]
        out.update(value=mv["value"], ms_per_step=mv["ms_per_step"])
        out["config"]["workload"] += "; MOVING view: " + mv["camera"] + "; " + mv["scene"]
        out["config"]["rays_per_step_all_gpus"] = mv["rays_per_step"]
        out["config"]["instances"] = mv["instances"]
        if mv["kernels_ms"].get("k_primary_ao"):
            k_ms = mv["kernels_ms"]["k_primary_ao"]
            out["roofline"].update(kernel_ms=k_ms, kernels_ms=mv["kernels_ms"], kernel_ms_per_frame=k_ms, frames_per_launch=1,
                                   note="moving view: kernel time of the moving frames; algorithmic bytes per launch are the still frame's "
                                        "(achieved / frac are recomputed with them -- the orbit keeps the castle in view, rays per frame within a few percent)")
            out["roofline"]["achieved"] = round(out["roofline"]["algorithmic_bytes_per_launch"] / (k_ms * 1e-3) / 1e9, 3)
            out["roofline"]["frac"] = round(out["roofline"]["achieved"] / HBM_PEAK_GBPS, 6)
    return out


def cpu_baseline(args, sc, noise5, synth):
    """The oracle's hierarchical traversal on the host cores, on a bounded row sample of the same frame (a reported baseline)."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # the checker, used here only as the reported CPU baseline (never in the timed GPU path)
    import parity_util
    W, H = args.width, args.height
    deep = args.workload == "deep"
    if deep:
        blocks, mats, pal, xf = sc["deep"]
        oscene = O.Scene()
        oscene.add_model(blocks, mats, pal, extent=4096)
        oscene.add_instance(0, xf.reshape(12))
        oscene.commit()
    else:
        oscene = parity_util.oracle_scene(sc["desc"])
    cores = os.cpu_count() or 1
    n_rows = args.cpu_rows or max(cores, min(H, (2 if deep else 8) * cores))
    n_rows = min(n_rows, H)
    y0 = (H - n_rows) // 2
    g = O.GBuffer(W, H)
    oc, osky = O.camera_from(sc["cam"]), O.sky_from(sc["sky"])
    n5 = np.ascontiguousarray(noise5[1 % len(noise5)])
    stats = [[O.OrcRayStats(), O.OrcRayStats(), O.OrcRayStats()] for _ in range(cores)]
    lib = O.lib()

    def work(t):
        a = y0 + (n_rows * t) // cores
        b = y0 + (n_rows * (t + 1)) // cores
        lib.orc_pass_primary(oscene.h, O.ORC_MODE_HIER, ctypes.byref(oc), ctypes.byref(osky), ctypes.byref(g.c), a, b,
                             ctypes.byref(stats[t][0]))
        lib.orc_pass_ao(oscene.h, O.ORC_MODE_HIER, ctypes.byref(oc), ctypes.byref(osky), ctypes.byref(g.c),
                        n5.ctypes.data_as(ctypes.c_void_p), synth.frame_rand(1, 1), a, b, ctypes.byref(stats[t][1]),
                        ctypes.byref(stats[t][2]))
    t0 = time.perf_counter()
    th = [threading.Thread(target=work, args=(t,)) for t in range(cores)]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.perf_counter() - t0
    cpu_rays = sum(s_.rays for ss in stats for s_ in ss)
    return {"value": round(cpu_rays / dt / 1e6, 4), "unit": "Mrays/s", "cores": cores, "kind": "port",
            "sample": f"primary + sun-shadow + AO rays of rows {y0}..{y0 + n_rows} of the same frame ({cpu_rays} rays, {dt:.1f} s), oracle "
                      f"hierarchical mode, {cores} threads; scene build (tree build + flatten / hierarchy upload) took {sc['t_load']:.2f} s on the same cores"}


def teapot_cpu(args):
    """configs[0]: teapot.vox stand-in, 256x256, one primary ray per pixel, CPU software vdb traversal -- the reference's own
    CPU-runnable case (its plumbing: vox loader + tree build on the host; the traversal is the oracle's port). No GPU
    is touched; the product library only parses the file and builds the trees (host code). Same JSON contract."""
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O  # this whole workload is the cpu_baseline leg
    import parity_util
    from dust_amd import api, synth
    from dust_amd import scenes as P
    W = H = 256
    t0 = time.perf_counter()
    desc = P.SceneDesc.from_vox(synth.teapot_scene(96))  # dust_vox_load (product host code): parse + tree build + flatten
    t_load = time.perf_counter() - t0
    m = desc.instances[0][1].reshape(3, 4).copy()   # the teapot hovers 200 voxels up, as in tests/test_configs.py
    m[:, 3] += np.array([0.0, 200.0, 0.0], np.float32)
    desc.instances[0] = (desc.instances[0][0], m.reshape(12))
    oscene = parity_util.oracle_scene(desc)
    eye = (60.0, 250.0, 70.0)
    cam = api.make_camera(eye, api.look_at_rotation(eye, (0.0, 200.0, 0.0)), api.PinholeProjection())
    oc, osky = O.camera_from(cam), O.sky_from(P.sky_state("default"))
    g = O.GBuffer(W, H)
    cores = os.cpu_count() or 1
    threads = min(cores, H // 2)
    lib = O.lib()
    steps = max(1, args.steps)
    stats = [O.OrcRayStats() for _ in range(threads)]

    def frame():
        def work(t):
            stats[t] = O.OrcRayStats()
            lib.orc_pass_primary(oscene.h, O.ORC_MODE_HIER, ctypes.byref(oc), ctypes.byref(osky), ctypes.byref(g.c), (H * t) // threads,
                                 (H * (t + 1)) // threads, ctypes.byref(stats[t]))
        th = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
        [x.start() for x in th]
        [x.join() for x in th]
    for _ in range(max(1, args.warmup)):
        frame()
    t0 = time.perf_counter()
    for _ in range(steps):
        frame()
    dt = time.perf_counter() - t0
    rays = sum(s.rays for s in stats)
    hits = sum(s.hits for s in stats)
    assert rays == W * H and 0 < hits < rays, (rays, hits)
    v = round(rays * steps / dt / 1e6, 4)
    out = {"metric": "Mrays/s at 256x256 teapot.vox, 1 primary ray/pixel, CPU software vdb traversal", "value": v, "unit": "Mrays/s",
           "n_gpus": 0, "steps": steps, "warmup": max(1, args.warmup), "ms_per_step": round(dt / steps * 1e3, 4), "higher_is_better": True,
           "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "config": {"workload": "teapot.vox stand-in (synth.teapot_scene), 256x256, 1 primary ray/pixel, CPU only (BASELINE configs[0])",
                      "frame": [W, H], "parallelism": f"{threads} host threads over row bands", "vox_models": len(desc.models),
                      "instances": len(desc.instances), "bricks": desc.n_bricks(), "scene_build_s": round(t_load, 4),
                      "rays_per_step": {"primary": int(rays)}, "hits": int(hits)},
           "roofline": None,
           "cpu_baseline": {"value": v, "unit": "Mrays/s", "cores": threads, "kind": "port",
                            "sample": f"the whole workload: {steps} frames of {rays} primary rays, oracle hierarchical mode, {threads} threads"}}
    print(json.dumps(out))


def self_launch(args):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks of this node here (the same command line the
    driver uses), after checking that the node has N devices."""
    from dust_amd import _lib
    have = _lib.load().dust_hip_device_count()
    if have < args.gpus:
        sys.exit(f"bench.py --gpus {args.gpus}: only {have} HIP device(s) visible on this node")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool: RCCL needs it
    sys.exit(subprocess.call(cmd, env=env))


def main():
    args = parse()
    if args.workload == "teapot_cpu":
        return teapot_cpu(args)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world
    be = HipBackend(rank, local_rank, world)
    if args.assets:
        from dust_amd import assets as _assets
        be.assets = _assets.Assets(args.assets)
    dist = be.init_dist() if world > 1 else None
    out = run_rank(args, be, dist)
    if out is not None:
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
