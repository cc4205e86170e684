"""Host-side conveniences over the C ABI that bench.py, the tools, the examples and the tests share: a flattened scene
description (what dust_vox_load produces), uploading it, the default camera and the baked sky fixtures.
Nothing here touches the oracle."""
import json
import os

import numpy as np

from . import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_SWEEP = None


def _sweep():
    global _SWEEP
    if _SWEEP is None:
        with open(os.path.join(ROOT, "dust_amd", "data", "sky_sweep.json")) as f:
            _SWEEP = json.load(f)
    return _SWEEP


def sky_state(name="default"):
    """56 baked sky floats (pipeline/sky.rs:78-85) of Sunlight::default(), from the packaged sweep (dust_amd/data/sky_sweep.json,
    made by tests/golden/make_sky_fixtures.py from the reference's tables). Hosts that hold the tables bake any sun exactly
    with api.Sunlight(...).bake(api.SkyDataset(...)) (dust_sky_bake)."""
    if name != "default":
        raise KeyError(f"packaged sky states: 'default' and sky_sweep(); not {name!r}")
    return np.asarray(_sweep()["default"]["state"], np.float32)


def sky_sweep(turbidity, elevation_deg, azimuth_deg=180.0):
    """Pre-baked state for hosts without the model's tables: the swept state nearest in (turbidity, solar elevation), with
    the sun placed at `azimuth_deg` (0 = +z, 90 = +x; the default sun stands at 180). The baked coefficients depend on the
    direction only through its elevation, so the state is exact for the swept elevation it snaps to (ground albedo 0.2)."""
    sw = _sweep()
    ti = int(np.argmin(np.abs(np.asarray(sw["turbidity"], float) - float(turbidity))))
    ei = int(np.argmin(np.abs(np.asarray(sw["elevation_deg"], float) - float(elevation_deg))))
    st = np.asarray(sw["states"][ti][ei], np.float32).copy()
    e, a = np.deg2rad(float(sw["elevation_deg"][ei])), np.deg2rad(float(azimuth_deg))
    st[48:51] = np.array([np.cos(e) * np.sin(a), min(1.0, np.sin(e)), np.cos(e) * np.cos(a)], np.float32)
    return st


class SceneDesc:
    """Flattened scene: models [(blocks, materials)], one palette, instances [(model, obj_to_world[12])]."""

    def __init__(self, models, palette, instances):
        self.models, self.palette, self.instances = models, palette, instances

    @staticmethod
    def from_vox(data: bytes):
        vs = api.VoxScene(data)
        used = sorted({m for m, _ in vs.instances})
        remap = {m: i for i, m in enumerate(used)}
        models = [vs.model_data(m) for m in used]
        return SceneDesc(models, vs.palette, [(remap[m], t) for m, t in vs.instances])

    def n_bricks(self):
        return sum(len(b) for b, _ in self.models)


def scatter_props(desc: SceneDesc, n, scale=1.0, seed=0xB0B):
    """Adds 8 small models and n instances of them to a castle scene (y up, ground at y = 0, +-640 in x and z at scale 1), with
    arbitrary rotations about the vertical axis: the scene of bench.py --props / curves.many_instances (thousands of TLAS entries,
    accel_struct/tlas.rs:79-117)."""
    rng = np.random.default_rng(seed)
    first = len(desc.models)
    for _ in range(8):
        sz = tuple(int(v) for v in rng.integers(10, 25, 3))
        solid = rng.random(sz) < 0.55
        solid[1:-1, 1:-1, 1:-1] &= rng.random((sz[0] - 2, sz[1] - 2, sz[2] - 2)) < 0.3
        x, y, z = np.nonzero(solid)
        xyzi = np.stack([x, y, z, rng.integers(0, 255, x.size)], axis=1).astype(np.uint8)
        desc.models.append(api.flatten_model(xyzi, sz, desc.palette))
    for _ in range(int(n)):
        ang = float(rng.uniform(0.0, 2.0 * np.pi))
        c, sn = np.cos(ang), np.sin(ang)
        m = np.zeros((3, 4), np.float32)
        m[:, :3] = np.array([[c, 0.0, sn], [0.0, 1.0, 0.0], [-sn, 0.0, c]], np.float32)
        m[:, 3] = (rng.uniform(-640.0, 640.0) * scale, rng.uniform(0.0, 60.0) * scale, rng.uniform(-640.0, 640.0) * scale)
        desc.instances.append((first + int(rng.integers(0, 8)), m.reshape(12)))
    return desc


def hip_scene(ctx, desc: SceneDesc):
    models = [api.Model(ctx, b, m, desc.palette) for b, m in desc.models]
    s = api.Scene(ctx)
    for mid, t in desc.instances:
        s.add_instance(models[mid], t)
    s.commit()
    return s


def camera_for(eye, target=(0.0, 0.0, 0.0), proj=None):
    proj = proj or api.PinholeProjection()
    return api.make_camera(eye, api.look_at_rotation(eye, target), proj)
