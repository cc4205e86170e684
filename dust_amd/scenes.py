"""Host-side conveniences over the C ABI that bench.py, the tools, the examples and the tests share: a flattened scene
description (what dust_vox_load produces), uploading it, the default camera and the baked sky fixtures.
Nothing here touches the oracle."""
import json
import os

import numpy as np

from . import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def sky_state(name="default"):
    """56 baked sky floats (pipeline/sky.rs:78-85) from tests/golden/sky_states.json (made by tests/golden/make_sky_fixtures.py)."""
    with open(os.path.join(ROOT, "tests", "golden", "sky_states.json")) as f:
        return np.asarray(json.load(f)[name]["state"], np.float32)


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
