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
