"""The multi-GPU path behind the C ABI (dust_amd/csrc/comm.hip; include/dust_hip.h "multi-GPU"): band gather and GI exchange through
a LOOPBACK group -- R ranks on the one GPU of the test box, each with a pipeline of its own, the collectives done by the library with
device copies and reduction kernels -- against the single-pipeline frame, bit for bit; and the RCCL communicator itself at world size
1 (what one GPU can run of it: librccl opened, a communicator made, the collective entry points taken)."""
import numpy as np
import pytest

import parity_util as P
from dust_amd import _lib as L, api, sharding, synth

pytestmark = pytest.mark.gpu


def _scene(ctx, scale=0.15):
    data, _ = synth.castle_scene(scale=scale)
    return P.hip_scene(ctx, P.SceneDesc.from_vox(data))


@pytest.mark.parametrize("world,cuts_kind", [(8, "equal"), (8, "cost"), (3, "ragged"), (5, "empty")])
def test_loopback_band_gather_equals_the_single_device_frame(world, cuts_kind):
    """8 (3, 5) emulated ranks render their row bands into pipelines of their own; dust_hip_gather_bands assembles the frame on the
    root, in the root's own plane (dst NULL) and in caller memory: both equal the frame one pipeline renders, bit for bit, for
    equal bands, bands cut at unequal 8-row boundaries, bands that are not multiples of 8 rows and empty bands."""
    torch = pytest.importorskip("torch")
    W, H = 200, 152
    ctx = api.Context(device=0)
    scene = _scene(ctx)
    n5 = synth.stbn_unitvec3_cosine(layers=2)
    cam, sky = P.camera_for((122.0 * 0.15, 300.61 * 0.15, 54.45 * 0.15)), P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    ref = api.StandardPipeline(ctx, W, H)
    ref.set_noise(5, n5)
    ref.render(scene, cam, sky, passes, 3, 77)
    ctx.sync()
    want = {pl: ref.read_plane(pl) for pl in (L.PLANE_ILLUMINANCE, L.PLANE_DEPTH, L.PLANE_VOXEL_ID)}
    if cuts_kind == "equal":
        cuts = [sharding.band_rows(r, world, H)[0] for r in range(world)] + [H]
    elif cuts_kind == "cost":
        cuts = [0, 8, 40, 48, 56, 104, 120, 144, H]
    elif cuts_kind == "ragged":
        cuts = [0, 13, 100, H]   # (rows of a band need not be multiples of the 8-row packets: the kernels mask the last packet row)
    else:
        cuts = [0, 0, 64, 64, 152, H]
    comms = api.Comm.local(ctx, world)
    assert comms[2].info() == (2, world, True)
    pipes = []
    for r in range(world):
        p = api.StandardPipeline(ctx, W, H)
        p.set_noise(5, n5)
        if cuts[r] < cuts[r + 1]:
            p.render(scene, cam, sky, passes, 3, 77, rows=(cuts[r], cuts[r + 1]))
        pipes.append(p)
    root = world - 1
    for plane in want:
        for r in range(world):
            comms[r].gather_bands(pipes[r], plane, cuts, root=root)   # carried out by the call that completes the group
        comms[root].sync()
        got = pipes[root].read_plane(plane)
        assert np.array_equal(got.view(np.uint8), want[plane].view(np.uint8)), (plane, cuts)
    # into caller memory on another root
    dst = torch.zeros((H, W, 4), dtype=torch.float16, device="cuda")
    for r in range(world):
        comms[r].gather_bands(pipes[r], L.PLANE_ILLUMINANCE, cuts, root=0, dst_ptr=dst.data_ptr(), dst_bytes=dst.numel() * 2)
    comms[0].sync()
    assert np.array_equal(dst.cpu().numpy().view(np.uint16), want[L.PLANE_ILLUMINANCE].view(np.uint16).reshape(H, W, 4))


def test_loopback_refuses_what_a_collective_cannot_be():
    ctx = api.Context(device=0)
    comms = api.Comm.local(ctx, 2)
    p = [api.StandardPipeline(ctx, 64, 32) for _ in range(2)]
    with pytest.raises(L.DustError):
        comms[0].gather_bands(p[0], L.PLANE_DEPTH, [0, 16, 31])       # cuts must end at the frame's height
    comms[0].gather_bands(p[0], L.PLANE_DEPTH, [0, 16, 32])
    with pytest.raises(L.DustError):
        comms[0].gather_bands(p[0], L.PLANE_DEPTH, [0, 16, 32])       # rank 0 is already waiting in this collective
    # (a call that cannot join drops what was pending -- records that hold raw pipeline pointers are not kept for a later call to complete)
    comms[0].gather_bands(p[0], L.PLANE_DEPTH, [0, 16, 32])
    with pytest.raises(L.DustError):
        comms[1].gather_bands(p[1], L.PLANE_DEPTH, [0, 8, 32])        # ranks disagree about the cuts
    comms[0].gather_bands(p[0], L.PLANE_DEPTH, [0, 16, 32])
    with pytest.raises(L.DustError):
        comms[1].gi_exchange(p[1], 16, 32, 16, 1)                     # ranks are in different collectives
    # (the failed collective is dropped: the group starts clean)
    for r in range(2):
        comms[r].gather_bands(p[r], L.PLANE_DEPTH, [0, 16, 32])
    comms[0].sync()
    other = api.Context(device=0)
    q = api.StandardPipeline(other, 64, 32)
    with pytest.raises(L.DustError):
        comms[0].gather_bands(q, L.PLANE_DEPTH, [0, 16, 32])          # a pipeline of another context


@pytest.mark.parametrize("world,shard_trace", [(2, False), (3, False), (2, True), (3, True), (8, True)])
def test_loopback_gi_exchange_equals_single_device(world, shard_trace):
    """dust_hip_gi_exchange_run (all-reduce MAX, all-gather, export, all-reduce SUM, import inside the library) drives the sharded GI
    frame of test_gpu_gi_sharded.py: every rank's hash, pool and band equal the single-pipeline run bit for bit.
    shard_trace (round 6): the surfel TRACE is sharded too -- rank r traces slots [r S, (r + 1) S) of the position-ordered pool into slot-ordered
    staging arrays, dust_hip_gi_surfel_exchange_run all-gathers them, repeats the trace's hash stamps and applies in surfel order: still
    the single-device hash and pool, bit for bit (the pool size 776 leaves the last of 8 ranks a short share, 13 groups over 8 ranks)."""
    W, H = 192, 104
    cap, pool = 16384, 97 * 8
    ctx = api.Context(device=0)
    scene = _scene(ctx)
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    cam, sky = P.camera_for((122.0 * 0.15, 300.61 * 0.15, 54.45 * 0.15)), P.sky_state()

    def make():
        p = api.StandardPipeline(ctx, W, H)
        p.set_noise(0, n0)
        p.set_noise(5, n5)
        p.configure_gi(cap, pool)
        return p
    ref = make()
    ranks = [make() for _ in range(world)]
    comms = api.Comm.local(ctx, world)
    per = sharding.gi_band_rows(world, H)
    bands = [(min(H, r * per), min(H, (r + 1) * per)) for r in range(world)]
    pix = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER
    for frame in range(1, 5):
        rnd = synth.frame_rand(7, frame)
        ref.render(scene, cam, sky, pix | L.PASS_SURFEL | L.PASS_GI_ORDERED, frame, rnd)
        for r, p in enumerate(ranks):   # (8 ranks on 104 rows: bands of 16, the last rank has none -- it renders no pixels and takes part in everything else)
            p.gi_exchange(world * per)
            if bands[r][0] < bands[r][1]:
                p.render(scene, cam, sky, pix | L.PASS_GI_SHARDED, frame, rnd, rows=bands[r])
        for r, p in enumerate(ranks):
            comms[r].gi_exchange(p, bands[r][0], bands[r][1], per, frame)
        for r, p in enumerate(ranks):
            p.render(scene, cam, sky, L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_GI_SHARDED, frame, rnd, surfel_shard=(r, world) if shard_trace else (0, 0))
        if shard_trace:
            with pytest.raises(L.DustError):   # a pending trace must be completed before the pipeline's next GI pass
                ranks[0].render(scene, cam, sky, L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_GI_SHARDED, frame, rnd, surfel_shard=(0, world))
            for r, p in enumerate(ranks):
                comms[r].gi_surfel_exchange(p, frame)
        ctx.sync()
        h_ref, s_ref = ref.read_gi()
        ill_ref = ref.read_plane(L.PLANE_ILLUMINANCE)
        for r, p in enumerate(ranks):
            h, sp = p.read_gi()
            assert np.array_equal(h, h_ref), f"frame {frame} rank {r}: hash differs in {(h != h_ref).any(axis=1).sum()} entries"
            assert (sp.view(np.uint32) == s_ref.view(np.uint32)).all(), f"frame {frame} rank {r}: surfel pool differs"
            ill = p.read_plane(L.PLANE_ILLUMINANCE)
            assert np.array_equal(ill[bands[r][0]:bands[r][1]], ill_ref[bands[r][0]:bands[r][1]]), f"frame {frame} rank {r}"
    assert int((h_ref[:, 0] != 0).sum()) > 50


def test_rccl_communicator_of_one_rank():
    """What ONE GPU can run of the RCCL path: librccl is opened, ncclGetUniqueId / ncclCommInitRank succeed, a gather (the root's own
    rows into caller memory, on the communicator's stream behind the frame) and a GI exchange of world 1 go through, wait / sync / destroy."""
    torch = pytest.importorskip("torch")
    W, H = 128, 72
    ctx = api.Context(device=0)
    scene = _scene(ctx)
    n0, n5 = synth.stbn_scalar(layers=2), synth.stbn_unitvec3_cosine(layers=2)
    cam, sky = P.camera_for((18.0, 45.0, 8.0)), P.sky_state()
    p = api.StandardPipeline(ctx, W, H)
    p.set_noise(0, n0)
    p.set_noise(5, n5)
    p.configure_gi(4096, 256)
    uid = api.Comm.unique_id()
    assert len(uid) == 128 and any(uid)
    comm = api.Comm.create(ctx, 0, 1, uid)
    assert comm.info() == (0, 1, False)
    p.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION, 1, 5)
    dst = torch.zeros((H, W, 4), dtype=torch.float16, device="cuda")
    comm.gather_bands(p, L.PLANE_ILLUMINANCE, [0, H], root=0, dst_ptr=dst.data_ptr(), dst_bytes=dst.numel() * 2)
    comm.wait()
    comm.sync()
    assert np.array_equal(dst.cpu().numpy().view(np.uint16), p.read_plane(L.PLANE_ILLUMINANCE).view(np.uint16).reshape(H, W, 4))
    p.gi_exchange(H)
    p.render(scene, cam, sky, L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_GI_SHARDED, 2, 6)
    comm.gi_exchange(p, 0, H, H, 2)
    p.render(scene, cam, sky, L.PASS_SURFEL | L.PASS_GI_ORDERED | L.PASS_GI_SHARDED, 2, 6)
    comm.sync()
    del comm


def test_gi_exchange_refuses_buffers_prepared_for_another_size():
    """dust_hip_gi_exchange_run never re-creates the exchange buffers (that would drop the stamps the frame's final gather has just
    written): buffers prepared for another world x band_rows are an error, not a silent re-allocation."""
    ctx = api.Context(device=0)
    comms = api.Comm.local(ctx, 2)
    pipes = [api.StandardPipeline(ctx, 64, 40) for _ in range(2)]
    for p in pipes:
        p.configure_gi(1 << 10, 128)
    with pytest.raises(L.DustError):
        comms[0].gi_exchange(pipes[0], 0, 24, 24, 1)        # (never prepared: the call that completes the group reports it ...)
        comms[1].gi_exchange(pipes[1], 24, 40, 24, 1)
    for p in pipes:
        p.gi_exchange(40)                                   # prepared for 40 padded rows; the run asks for 2 x 24 = 48
    with pytest.raises(L.DustError):
        comms[0].gi_exchange(pipes[0], 0, 24, 24, 1)
        comms[1].gi_exchange(pipes[1], 24, 40, 24, 1)
    for p in pipes:
        p.gi_exchange(48)
    comms[0].gi_exchange(pipes[0], 0, 24, 24, 1)
    comms[1].gi_exchange(pipes[1], 24, 40, 24, 1)
    comms[0].sync()


@pytest.mark.parametrize("world", [8, 3])
def test_denoised_frame_on_n_ranks_equals_the_single_device_frame(world):
    """The reference denoises every frame (crates/render/src/pipeline/nrd.rs:272-617; examples/castle.rs:190-231: render -> NRD -> tone
    map). On N ranks: every rank renders its band, ONE dust_hip_gather_planes moves the five planes the filter reads (illuminance,
    depth, normal, motion, voxel id) to the root, the root filters the whole frame and tone-maps it. Four frames (the filter's history
    builds up on the root): the denoised and the display planes equal the single pipeline's, bit for bit."""
    W, H = 200, 152
    ctx = api.Context(device=0)
    scene = _scene(ctx)
    n5 = synth.stbn_unitvec3_cosine(layers=4)
    sky = P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    planes = (L.PLANE_ILLUMINANCE, L.PLANE_DEPTH, L.PLANE_NORMAL, L.PLANE_MOTION, L.PLANE_VOXEL_ID, L.PLANE_DENOISED, L.PLANE_ALBEDO)
    ref = api.StandardPipeline(ctx, W, H)
    ref.set_noise(5, n5)
    comms = api.Comm.local(ctx, world)
    pipes = []
    for r in range(world):
        p = api.StandardPipeline(ctx, W, H)
        p.set_noise(5, n5)
        pipes.append(p)
    cuts = [sharding.band_rows(r, world, H)[0] for r in range(world)] + [H]
    root = 0
    for f in range(1, 5):
        eye = (122.0 * 0.15 + 0.3 * f, 300.61 * 0.15, 54.45 * 0.15 - 0.2 * f)   # a slowly moving view: reprojection, disocclusion
        cam = P.camera_for(eye)
        rnd = synth.frame_rand(5, f)
        ref.render(scene, cam, sky, passes | L.PASS_DENOISE, f, rnd)
        ref.tone_map()
        for r in range(world):
            if cuts[r] < cuts[r + 1]:
                pipes[r].render(scene, cam, sky, passes, f, rnd, rows=(cuts[r], cuts[r + 1]))
        for r in range(world):
            comms[r].gather_planes(pipes[r], planes, cuts, root=root)
        comms[root].wait()
        pipes[root].render(scene, cam, sky, L.PASS_DENOISE, f, rnd)
        pipes[root].tone_map()
        comms[root].sync()
        for pl in (L.PLANE_DENOISED, L.PLANE_ACCUM, L.PLANE_OUTPUT):
            a, b = ref.read_plane(pl), pipes[root].read_plane(pl)
            assert np.array_equal(a.view(np.uint8), b.view(np.uint8)), (f, pl)
