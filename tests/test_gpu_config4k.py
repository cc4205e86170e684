"""BASELINE.json configs[3] at its stated size: castle stand-in, 3840 x 2160, the frame cut into the 8 row bands of an
8-GPU node. One GPU plays all eight ranks (the collectives are done by hand here and run over gloo in
tests/test_distributed_cpu.py): the union of the bands must be the full frame bit for bit, and after two frames of the
sharded GI protocol every rank's spatial hash, surfel pool and own band must equal the single-device run."""
import numpy as np
import pytest

import parity_util as P
from dust_amd import _lib as L, api, sharding, synth

pytestmark = pytest.mark.gpu
W, H = 3840, 2160
EYE = (122.0, 300.61, 54.45)  # examples/castle.rs:126


@pytest.fixture(scope="module")
def castle():
    data, info = synth.castle_scene()
    desc = P.SceneDesc.from_vox(data)
    ctx = api.Context(device=0)
    return ctx, P.hip_scene(ctx, desc)


def test_4k_eight_bands_equal_full_frame(castle):
    ctx, scene = castle
    noise5 = synth.stbn_unitvec3_cosine()
    cam, sky = P.camera_for(EYE), P.sky_state()
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION
    full = api.StandardPipeline(ctx, W, H)
    full.set_noise(5, noise5)
    full.render(scene, cam, sky, passes | L.PASS_COUNT_STATS, frame_index=3, rand=synth.frame_rand(1, 3))
    ref = P.read_hip_gbuffer(full)
    st = [full.pass_stats(i) for i in range(3)]
    assert st[0].rays == W * H and st[2].rays == st[0].hits and 0 < st[1].rays <= st[0].hits
    banded = api.StandardPipeline(ctx, W, H)
    banded.set_noise(5, noise5)
    rows = [sharding.band_rows(r, 8, H) for r in range(8)]
    assert rows[0][0] == 0 and rows[-1][1] == H and all(a[1] == b[0] for a, b in zip(rows, rows[1:]))
    for r in reversed(range(8)):  # any order: the bands are independent
        banded.render(scene, cam, sky, passes, frame_index=3, rand=synth.frame_rand(1, 3), rows=rows[r])
    got = P.read_hip_gbuffer(banded)
    for k in ref:
        assert ref[k].tobytes() == got[k].tobytes(), f"bands: {k}"
    assert np.isfinite(ref["depth"]).mean() > 0.5


def test_4k_sharded_gi_eight_ranks_equal_single_device(castle):
    ctx, scene = castle
    h_ref = P.sharded_gi_vs_single_device(ctx, scene, P.camera_for(EYE), P.sky_state(), W, H, world=8, frames=2,
                                          n0=synth.stbn_scalar(), n5=synth.stbn_unitvec3_cosine(), seed=4)
    assert int((h_ref[:, 0] != 0).sum()) > 10_000
