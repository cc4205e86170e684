"""The GI passes as ray streams (DUST_HIP_RAY_STREAM=1; gi.hip: k_gather_rays / k_surfel_rays bin every ray over the top-level grid
of dust_hip_scene_commit, k_ray_walk walks one ray per lane with lanes refilled, k_final_gather_shade / k_surfel_shade read the hit
records): the same tests against the oracle that the packet kernels pass, and the pass statistics of the two paths side by side.
The switch is read when a pipeline is created."""
import numpy as np
import pytest

import parity_util as P
import test_gpu_gi as G
from dust_amd import _lib as L, api, synth

pytestmark = pytest.mark.gpu


@pytest.fixture
def stream(monkeypatch):
    monkeypatch.setenv("DUST_HIP_RAY_STREAM", "1")
    return monkeypatch


def test_stream_castle_gi_matches_oracle(stream):
    G.test_castle_gi_matches_oracle(stream)


def test_stream_gi_sequence_matches_oracle(stream):
    for capacity, pool in ((1 << 14, 2048), (97, 777)):   # a roomy table, and one where every probe collides and evicts
        G.test_gi_sequence_matches_oracle(capacity, pool)


def test_stream_mixed_two_and_three_level_trees(stream):
    G.test_gi_on_mixed_two_and_three_level_trees_matches_oracle()


def test_stream_more_candidates_than_a_ray_record_holds(stream):
    """220 stacked instances: most rays meet more than the seven boxes DevRay::cand lists, and the lane walks the grid itself"""
    G.test_candidate_list_overflow_matches_oracle()


@pytest.mark.parametrize("density", ["0.01", "200"])
def test_stream_grid_resolution_does_not_change_results(stream, density):
    """one cell for the whole scene (every instance in one list), and ~200 cells per instance"""
    stream.setenv("DUST_HIP_GRID_DENSITY", density)   # (read by every dust_hip_scene_commit)
    G.test_castle_gi_matches_oracle(stream)
    stream.delenv("DUST_HIP_GRID_DENSITY", raising=False)


def test_stream_sharded_gi_equals_single_gpu(stream):
    import test_gpu_gi_sharded as S
    S.test_sharded_gi_equals_single_gpu(2)


def test_stream_and_packet_statistics_agree(monkeypatch):
    """Rays and hits per GI ray class are properties of the frame, not of the kernels that trace it; the stream path enters
    fewer instances (per-ray box tests in the grid walk instead of a packet's candidate list)."""
    data, _ = synth.castle_scene(scale=0.15)
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    scene = P.hip_scene(ctx, desc)
    s = 0.15
    sky, cam = P.sky_state(), P.camera_for((122.0 * s, 300.61 * s, 54.45 * s))
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL | L.PASS_GI_ORDERED
    stats = []
    for env in (None, "1"):
        if env:
            monkeypatch.setenv("DUST_HIP_RAY_STREAM", env)
        else:
            monkeypatch.delenv("DUST_HIP_RAY_STREAM", raising=False)
        pipe = api.StandardPipeline(ctx, 192, 104)
        pipe.set_noise(0, n0)
        pipe.set_noise(5, n5)
        pipe.configure_gi(1 << 14, 776)
        for f in range(1, 4):
            pipe.render(scene, cam, sky, passes | (L.PASS_COUNT_STATS if f == 3 else 0), frame_index=f, rand=synth.frame_rand(7, f))
        stats.append([pipe.pass_stats(i) for i in range(3, 6)])
    monkeypatch.delenv("DUST_HIP_RAY_STREAM", raising=False)
    for a, b in zip(*stats):
        assert a.rays == b.rays and a.hits == b.hits and a.rays > 0, (a.rays, b.rays, a.hits, b.hits)
        assert b.bricks_tested >= b.hits
