"""Sunlight::bake in product form (dust_sky_bake, dust_amd/csrc/sky.cpp; crates/render/src/pipeline/sky.rs:90-268).

Two witnesses: (1) on SYNTHETIC coefficient tables -- no reference file needed, runs anywhere -- the C++ bake must equal
the float32 numpy restatement in tests/golden/make_sky_fixtures.py bit for bit; (2) in the build container, where the
reference's dataset.bin / datasetSolar.bin exist, baking the named suns must reproduce tests/golden/sky_states.json and the
packaged sweep (dust_amd/data/sky_sweep.json) exactly."""
import importlib.util
import json
import os

import numpy as np
import pytest

from dust_amd import _lib as L, api, scenes

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/crates/render/src/pipeline"

spec = importlib.util.spec_from_file_location("make_sky_fixtures", os.path.join(HERE, "golden", "make_sky_fixtures.py"))
msf = importlib.util.module_from_spec(spec)
spec.loader.exec_module(msf)


def synthetic_tables(seed):
    rng = np.random.default_rng(seed)
    return rng.uniform(-2.0, 2.0, 1200 * 3).astype("<f4"), rng.uniform(0.0, 3.0e4, 1806 * 3).astype("<f4")


@pytest.mark.parametrize("seed", [1, 2])
def test_bake_matches_numpy_restatement_on_synthetic_tables(seed):
    ds, sol = synthetic_tables(seed)
    data = api.SkyDataset(ds.tobytes(), sol.tobytes())
    tables = msf.split_tables(ds, sol)
    rng = np.random.default_rng(seed + 10)
    identical = 0
    for k in range(200):
        turb = float(np.float32(rng.choice([1.0, 2.0, 9.0, 10.0]) if k % 4 == 0 else rng.uniform(1.0, 10.0)))
        alb = rng.uniform(0.0, 1.0, 3).astype(np.float32)
        el = rng.uniform(0.01, 1.55)
        az = rng.uniform(0, 2 * np.pi)
        d = np.array([np.cos(el) * np.sin(az), np.sin(el), np.cos(el) * np.cos(az)], np.float32)
        got = api.Sunlight(turb, alb, d).bake(data)
        want = msf.bake_with(tables, turb, alb, d)
        # powf(x, 1/3) is the one libm call on the path (glibc's powf here, numpy's float32 power there): where they differ in
        # the last bit of the Bezier parameter the random tables' cancelling sums move by a few ulp; everything else is exact
        assert np.allclose(got, want, rtol=2e-5, atol=2e-6 * np.abs(want[:48]).max()), (k, turb)
        identical += got.tobytes() == want.tobytes()
    assert identical >= 150, identical


def test_bake_rejects_bad_input():
    ds, sol = synthetic_tables(3)
    with pytest.raises(L.DustError):
        api.SkyDataset(ds.tobytes()[:-4], sol.tobytes())
    data = api.SkyDataset(ds.tobytes(), sol.tobytes())
    for turb, d in ((0.5, (0, 1, 0)), (10.5, (0, 1, 0)), (3.0, (0, -0.2, 0.98)), (3.0, (0, 0.0, 1.0))):
        with pytest.raises(L.DustError):
            api.Sunlight(turb, (0.2, 0.2, 0.2), d).bake(data)


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference's sky tables are only present in the build container")
def test_bake_reproduces_committed_fixtures_from_the_reference_tables():
    data = api.SkyDataset(open(os.path.join(REF, "dataset.bin"), "rb").read(), open(os.path.join(REF, "datasetSolar.bin"), "rb").read())
    fx = json.load(open(os.path.join(HERE, "golden", "sky_states.json")))
    for name, s in fx.items():
        got = api.Sunlight(s["turbidity"], s["albedo"], s["direction"]).bake(data)
        want = np.asarray(s["state"], np.float32)
        assert got.tobytes() == want.tobytes(), (name, int((got != want).sum()))
    # the values SURVEY 8(c) recorded for Sunlight::default()
    d = np.asarray(fx["default"]["state"], np.float32)
    assert np.allclose([d[9], d[25], d[41]], [0.0492231, 0.0523076, 0.0967448], rtol=2e-6)
    assert np.allclose(d[52:55], [2.297167e6, 2.391441e6, 2.118000e6], rtol=1e-6)
    sw = json.load(open(os.path.join(os.path.dirname(HERE), "dust_amd", "data", "sky_sweep.json")))
    for ti in (0, 4, 9):
        for ei in (0, 17, 44):
            st = np.asarray(sw["states"][ti][ei], np.float32)
            got = api.Sunlight(float(sw["turbidity"][ti]), sw["albedo"], st[48:51]).bake(data)
            assert got.tobytes() == st.tobytes()


def test_packaged_sweep():
    d = scenes.sky_state()
    assert d.shape == (56,) and abs(float(d[49]) - 0.80114365) < 1e-7
    s = scenes.sky_sweep(3.2, 41.3, azimuth_deg=90.0)
    assert abs(float(np.linalg.norm(s[48:51])) - 1.0) < 1e-6 and abs(float(s[49]) - np.sin(np.deg2rad(42.0))) < 1e-6 and s[48] > 0.7
    lo, hi = scenes.sky_sweep(2, 10), scenes.sky_sweep(2, 80)
    assert hi[25] > lo[25] > 0  # zenith luminance grows with the sun's elevation
