"""Spatial hash + GI passes of the oracle (spatial_hash.glsl, final_gather/*, surfel/*): CPU-only checks."""
import numpy as np

import oracle_lib as O
import parity_util as P
from dust_amd import _lib as L
from dust_amd import synth


def test_hash_insert_get_running_mean_and_cap():
    gi = O.GI(1 << 10, 16)
    key = ((3, -7, 11), 5)
    found, val, cnt = gi.get(*key, frame=1)
    assert not found and cnt == 0 and val == [0.0, 0.0, 0.0]
    gi.insert(*key, (1.0, 2.0, 3.0), 1)
    found, val, cnt = gi.get(*key, frame=2)
    assert found and cnt == 1
    assert np.allclose(val, (1.0, 2.0, 3.0), rtol=0.02)  # LogLuv: 0.17 % luminance steps, 9-bit chroma
    for i in range(2, 420):
        gi.insert(*key, (1.0, 2.0, 3.0), i)
    _, _, cnt = gi.get(*key, frame=500)
    assert cnt == 404                      # MAX_SAMPLE_COUNT (spatial_hash.glsl:179-181)
    h = gi.hash()
    e = h[h["fingerprint"] != 0]
    assert len(e) == 1 and e[0]["last_accessed_frame"] == 500 and e[0]["sample_count"] == 404
    # fingerprint / location follow the xxhash / pcg chains
    l = O.lib()
    import ctypes as C
    pos = (C.c_int32 * 3)(3, -7, 11)
    assert e[0]["fingerprint"] == l.orc_hash_fingerprint(pos, 5)
    loc = l.orc_hash_location(pos, 5, 1 << 10)
    assert h[loc]["fingerprint"] == e[0]["fingerprint"]


def test_hash_probing_and_lru_eviction():
    cap = 8
    gi = O.GI(cap, 4)
    l = O.lib()
    import ctypes as C
    # collect keys that map to the same location
    target, keys = None, []
    for x in range(4000):
        loc = l.orc_hash_location((C.c_int32 * 3)(x, 0, 0), 1, cap)
        if target is None:
            target = loc
        if loc == target:
            keys.append(x)
        if len(keys) == 4:
            break
    for f, x in enumerate(keys[:3]):
        gi.insert((x, 0, 0), 1, (1.0 + x, 1.0, 1.0), f + 1)   # fills the three probe slots
    h = gi.hash()
    assert all(h[target + i]["fingerprint"] != 0 for i in range(3))
    gi.get((keys[0], 0, 0), 1, frame=10)                       # touch the first: no longer the LRU
    gi.insert((keys[3], 0, 0), 1, (9.0, 9.0, 9.0), 11)         # must evict the second (last_accessed_frame == 2)
    h = gi.hash()
    fp3 = l.orc_hash_fingerprint((C.c_int32 * 3)(keys[3], 0, 0), 1)
    assert h[target + 1]["fingerprint"] == fp3 and h[target + 1]["sample_count"] == 1
    assert gi.get((keys[0], 0, 0), 1, frame=12)[0] and not gi.get((keys[1], 0, 0), 1, frame=12)[0]


def test_gi_frames_feed_back():
    """Three frames of primary + AO + final gather + surfel: surfels get enqueued by the final gather, the surfel
    pass fills the hash, and later final gathers find radiance there."""
    desc = P.small_scene(seed=5, n_models=2, n_instances=4, size=(28, 28, 28))
    s = P.oracle_scene(desc)
    sky = P.sky_state()
    cam = P.camera_for((80.0, 60.0, 90.0))
    w, h = 64, 40
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    gi = O.GI(1 << 14, 2048)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    filled, valid = [], []
    for f in range(1, 4):
        P.render_oracle(s, cam, sky, w, h, passes, n5[f % 4], synth.frame_rand(1, f), noise0=n0[f % 4], gi=gi, frame_index=f)
        filled.append(int((gi.hash()["fingerprint"] != 0).sum()))
        valid.append(int((gi.pool()["direction"] < 6).sum()))
    assert valid[0] > 0 and valid[-1] >= valid[0]
    assert filled[0] > 0 and filled[-1] >= filled[0]
    hh = gi.hash()
    assert (hh["sample_count"][hh["fingerprint"] != 0] >= 1).all()


def test_threaded_gi_passes_equal_the_serial_ones():
    """orc_pass_final_gather_mt / orc_pass_surfel_mt (what the full-size GPU comparison threads the oracle with) leave the G-buffer,
    the hash and the pool exactly as the serial passes do -- small tables, so that slots alias and probes collide."""
    desc = P.small_scene(seed=8, n_models=2, n_instances=5, size=(28, 28, 28))
    s = P.oracle_scene(desc)
    sky = P.sky_state()
    cam = P.camera_for((80.0, 60.0, 90.0))
    w, h = 72, 48
    n0, n5 = synth.stbn_scalar(layers=4), synth.stbn_unitvec3_cosine(layers=4)
    passes = L.PASS_PRIMARY | L.PASS_AMBIENT_OCCLUSION | L.PASS_FINAL_GATHER | L.PASS_SURFEL
    out = []
    for threads in (0, 5):
        gi = O.GI(211, 97)   # (pool slots alias every 97 pixels, far inside a thread's band and across bands)
        for f in range(1, 5):
            g = P.render_oracle(s, cam, sky, w, h, passes, n5[f % 4], synth.frame_rand(1, f), noise0=n0[f % 4], gi=gi, frame_index=f,
                                gi_threads=threads)
        out.append((gi.hash().copy(), gi.pool().copy(), g.illuminance.copy()))
    assert out[0][0].tobytes() == out[1][0].tobytes()
    assert out[0][1].tobytes() == out[1][1].tobytes()
    assert out[0][2].tobytes() == out[1][2].tobytes()
    assert int((out[0][0]["fingerprint"] != 0).sum()) >= 10 and int((out[0][1]["direction"] < 6).sum()) >= 10
