"""Thin Python handles over the C ABI (include/dust_hip.h) for tests and bench.py.

Names follow the reference's surface: Tree (crates/vdb/src/tree.rs), VoxLoader/VoxGeometry
(crates/vox/src/{loader,geometry}.rs), StandardPipeline/GBuffer/PinholeProjection/Sunlight
(crates/render/src/pipeline/standard.rs, projection.rs, pipeline/sky.rs). The C++ mirror of the same
surface, which is what a Rust maintainer would read, is include/dust_hip.hpp.
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L

BLOCK_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("z", "<u2"), ("w", "<u2"), ("mask", "<u8"),
                        ("material_ptr", "<u4"), ("avg_albedo", "<u4")])
assert BLOCK_DTYPE.itemsize == 24
SURFEL_DTYPE = np.dtype([("pos", "<f4", 3), ("direction", "<u4")])

PLANE_DTYPES = {
    L.PLANE_ILLUMINANCE: (np.uint16, 4), L.PLANE_DENOISED: (np.uint16, 4), L.PLANE_ALBEDO: (np.uint32, 1),
    L.PLANE_NORMAL: (np.uint32, 1), L.PLANE_DEPTH: (np.float32, 1), L.PLANE_MOTION: (np.uint16, 4),
    L.PLANE_VOXEL_ID: (np.uint32, 1), L.PLANE_ACCUM: (np.float32, 4), L.PLANE_OUTPUT: (np.uint16, 4),
}


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


class Tree:
    """dust_vdb::Tree<hierarchy!(...)> (crates/vdb/src/tree.rs:7-124)."""

    def __init__(self, *fanout_log2):
        self._lib = L.load()
        arr = (C.c_uint32 * len(fanout_log2))(*fanout_log2)
        self._h = C.c_void_p()
        L.check(self._lib.dust_vdb_tree_create(arr, len(fanout_log2), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_vdb_tree_destroy(self._h)
            self._h = None

    def set_value(self, xyz, value):
        v = -1 if value is None else (1 if value else 0)
        L.check(self._lib.dust_vdb_tree_set(self._h, int(xyz[0]), int(xyz[1]), int(xyz[2]), v))

    def get_value(self, xyz):
        out = C.c_int32()
        L.check(self._lib.dust_vdb_tree_get(self._h, int(xyz[0]), int(xyz[1]), int(xyz[2]), C.byref(out)))
        return None if out.value < 0 else bool(out.value)

    def iter(self):
        n = C.c_size_t()
        L.check(self._lib.dust_vdb_tree_iter(self._h, None, 0, C.byref(n)))
        out = np.zeros((max(n.value, 1), 3), np.uint32)
        L.check(self._lib.dust_vdb_tree_iter(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), n.value, C.byref(n)))
        return out[: n.value]

    def iter_leaf(self):
        n = C.c_size_t()
        L.check(self._lib.dust_vdb_tree_iter_leaf(self._h, None, None, None, 0, C.byref(n)))
        k = max(n.value, 1)
        xyz = np.zeros((k, 3), np.uint32)
        occ = np.zeros(k, np.uint64)
        mat = np.zeros(k, np.uint32)
        L.check(self._lib.dust_vdb_tree_iter_leaf(self._h, xyz.ctypes.data_as(C.POINTER(C.c_uint32)),
                                                  occ.ctypes.data_as(C.POINTER(C.c_uint64)),
                                                  mat.ctypes.data_as(C.POINTER(C.c_uint32)), n.value, C.byref(n)))
        return xyz[: n.value], occ[: n.value], mat[: n.value]

    def meta(self):
        mask, lvl = C.c_uint32(), C.c_uint32()
        L.check(self._lib.dust_vdb_tree_meta(self._h, C.byref(mask), C.byref(lvl)))
        return mask.value, lvl.value

    def accessor(self):
        return Accessor(self)


class Accessor:
    """dust_vdb::Accessor (crates/vdb/src/accessor.rs:5-57)."""

    def __init__(self, tree):
        self._tree = tree
        self._lib = tree._lib
        self._h = C.c_void_p()
        L.check(self._lib.dust_vdb_accessor_create(tree._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_vdb_accessor_destroy(self._h)
            self._h = None

    def get(self, xyz):
        out = C.c_int32()
        L.check(self._lib.dust_vdb_accessor_get(self._h, int(xyz[0]), int(xyz[1]), int(xyz[2]), C.byref(out)))
        return None if out.value < 0 else bool(out.value)


def lca_level(a, b, mask, root_level):
    lib = L.load()
    aa = (C.c_uint32 * 3)(*[int(v) for v in a])
    bb = (C.c_uint32 * 3)(*[int(v) for v in b])
    return lib.dust_vdb_lca_level(aa, bb, mask, root_level)


def flatten_model(xyzi, size, palette256):
    """load_model + VoxGeometry::from_tree (loader.rs:238-308, geometry.rs:55-179) -> (blocks, materials)."""
    lib = L.load()
    xyzi = np.ascontiguousarray(xyzi, np.uint8).reshape(-1, 4)
    pal = np.ascontiguousarray(palette256, np.uint8).reshape(256, 4)
    sz = (C.c_uint32 * 3)(*[int(v) for v in size])
    blocks = C.POINTER(L.Block)()
    mats = C.POINTER(C.c_uint8)()
    nb, nm = C.c_uint32(), C.c_uint64()
    L.check(lib.dust_vox_flatten_model(_ptr(xyzi), xyzi.shape[0], sz, _ptr(pal), C.byref(blocks), C.byref(nb),
                                       C.byref(mats), C.byref(nm)))
    try:
        b = np.frombuffer(C.string_at(blocks, nb.value * 24), BLOCK_DTYPE).copy()
        m = np.frombuffer(C.string_at(mats, nm.value), np.uint8).copy()
    finally:
        lib.dust_vox_free(blocks)
        lib.dust_vox_free(mats)
    return b, m


class VoxScene:
    """What VoxLoader::load yields before upload (loader.rs:322-415): models, palette, instances."""

    def __init__(self, data: bytes, frame: int = 0):
        self._lib = L.load()
        self._h = C.c_void_p()
        buf = np.frombuffer(data, np.uint8)
        if frame:  # animation frame: multi-frame nTRN / multi-model nSHP nodes (loader.rs:103-105,149-151 are unimplemented!())
            L.check(self._lib.dust_vox_load_frame(_ptr(buf), buf.size, int(frame), C.byref(self._h)))
        else:
            L.check(self._lib.dust_vox_load(_ptr(buf), buf.size, C.byref(self._h)))
        nm, ni = C.c_uint32(), C.c_uint32()
        L.check(self._lib.dust_vox_scene_counts(self._h, C.byref(nm), C.byref(ni)))
        self.n_models, self.n_instances = nm.value, ni.value
        pal = C.POINTER(C.c_uint8)()
        L.check(self._lib.dust_vox_scene_palette(self._h, C.byref(pal)))
        self.palette = np.frombuffer(C.string_at(pal, 1024), np.uint8).reshape(256, 4).copy()
        inst = (L.VoxInstance * max(ni.value, 1))()
        L.check(self._lib.dust_vox_scene_instances(self._h, inst, ni.value))
        self.instances = [(inst[i].model, np.array(inst[i].obj_to_world, np.float32)) for i in range(ni.value)]

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_vox_scene_destroy(self._h)
            self._h = None

    def model_info(self, i):
        info = L.VoxModelInfo()
        L.check(self._lib.dust_vox_scene_model_info(self._h, i, C.byref(info)))
        return info

    def model_data(self, i):
        info = self.model_info(i)
        blocks = C.POINTER(L.Block)()
        mats = C.POINTER(C.c_uint8)()
        L.check(self._lib.dust_vox_scene_model_data(self._h, i, C.byref(blocks), C.byref(mats)))
        b = np.frombuffer(C.string_at(blocks, info.n_blocks * 24), BLOCK_DTYPE).copy() if info.n_blocks else np.zeros(0, BLOCK_DTYPE)
        m = np.frombuffer(C.string_at(mats, info.n_materials), np.uint8).copy() if info.n_materials else np.zeros(0, np.uint8)
        return b, m


class PinholeProjection:
    """crates/render/src/projection.rs:3-29"""

    def __init__(self, fov=math.pi / 4, near=0.1, far=10000.0):
        self.fov, self.near, self.far = fov, near, far


def look_at_rotation(eye, target, up=(0.0, 1.0, 0.0)):
    """Rotation of a camera looking from eye to target (columns = camera x, y, z axes; -z is forward),
    as Transform::looking_at builds it for the FPS camera of examples/castle.rs:120-129."""
    eye, target, up = (np.asarray(v, np.float64) for v in (eye, target, up))
    back = eye - target
    back /= np.linalg.norm(back)
    right = np.cross(up, back)
    right /= np.linalg.norm(right)
    up2 = np.cross(back, right)
    return np.stack([right, up2, back], axis=1).astype(np.float32)  # columns


def make_camera(eye, rotation_cols, projection: PinholeProjection):
    """CameraSettings members the shaders read (standard.rs:277-302)."""
    cam = L.Camera()
    r = np.asarray(rotation_cols, np.float32)
    cam.view_col0[:] = r[:, 0].tolist()
    cam.view_col1[:] = r[:, 1].tolist()
    cam.view_col2[:] = r[:, 2].tolist()
    cam.position[:] = [float(np.float32(v)) for v in eye]
    cam.tan_half_fov = float(np.float32(np.tan(np.float32(projection.fov) / np.float32(2.0))))
    cam.far_ = projection.far
    cam.near_ = projection.near
    return cam


# ColorSpacePrimaries (rhyolite/src/utils/format.rs:573-641): r, g, b, white point chromaticities
BT709 = ((0.64, 0.33), (0.3, 0.6), (0.15, 0.06), (0.3127, 0.3290))
ACES_AP1 = ((0.713, 0.293), (0.165, 0.830), (0.128, 0.044), (0.32168, 0.33767))
DCI_P3 = ((0.68, 0.32), (0.265, 0.69), (0.15, 0.06), (0.3127, 0.3290))


def primaries_to_xyz(p):
    """ColorSpacePrimaries::to_xyz (format.rs:651-664)."""
    x = np.array([p[0][0], p[1][0], p[2][0], p[3][0]], np.float64)
    y = np.array([p[0][1], p[1][1], p[2][1], p[3][1]], np.float64)
    X, Z = x / y, (1.0 - x - y) / y
    mat = np.stack([X[:3], np.ones(3), Z[:3]])          # rows X, Y, Z; columns r, g, b
    s = np.linalg.solve(mat, np.array([X[3], 1.0, Z[3]]))
    return mat * s[None, :]


def color_space_conversion(src=ACES_AP1, dst=BT709):
    """ColorSpacePrimaries::to_color_space (format.rs:666-679), no chromatic adaptation; returned column-major
    as the nine COLOR_SPACE_CONVERSION_* specialization constants of tone_map.comp."""
    m = np.linalg.inv(primaries_to_xyz(dst)) @ primaries_to_xyz(src)
    return m.T.reshape(9).astype(np.float32)


class SkyDataset:
    """The Hosek-Wilkie tables Sunlight::bake reads (sky.rs:25-64), handed over as the bytes of the reference's
    dataset.bin and datasetSolar.bin."""

    def __init__(self, dataset_bin: bytes, dataset_solar_bin: bytes):
        self._lib = L.load()
        self._h = C.c_void_p()
        a, b = np.frombuffer(dataset_bin, np.uint8), np.frombuffer(dataset_solar_bin, np.uint8)
        L.check(self._lib.dust_sky_dataset_create(_ptr(a), a.size, _ptr(b), b.size, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_sky_dataset_destroy(self._h)
            self._h = None


class Sunlight:
    """dust_render::Sunlight (crates/render/src/pipeline/sky.rs:6-23): turbidity, ground albedo, direction eye -> sun."""

    def __init__(self, turbidity=1.0, albedo=(0.2, 0.2, 0.2), direction=(0.0, 0.80114365, -0.5984721)):
        self.turbidity, self.albedo, self.direction = float(turbidity), tuple(albedo), tuple(direction)

    def bake(self, dataset: SkyDataset):
        """Sunlight::bake (sky.rs:90-132) -> the 56 floats of SkyModelState."""
        out = L.Sky()
        alb = (C.c_float * 3)(*self.albedo)
        d = (C.c_float * 3)(*self.direction)
        L.check(dataset._lib.dust_sky_bake(dataset._h, self.turbidity, alb, d, C.byref(out)))
        return np.array(out.state, np.float32)


class Context:
    def __init__(self, device=-1, timing=True, stream=None, lds_root_bytes=0, sparse_timing=False):
        self._lib = L.load()
        flags = (L.CONTEXT_TIMING if timing else 0) | (L.CONTEXT_TIMING_SPARSE if timing and sparse_timing else 0)
        cfg = L.Config(C.sizeof(L.Config), device, stream, lds_root_bytes, flags)
        self._h = C.c_void_p()
        L.check(self._lib.dust_hip_context_create(C.byref(cfg), C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_hip_context_destroy(self._h)
            self._h = None

    def sync(self):
        L.check(self._lib.dust_hip_sync(self._h))

    def device_eval(self, fn, rows, out_words):
        """dust_hip_device_eval: rows is an (n, in_words) array of 32-bit words (float32 or uint32); returns (n, out_words) uint32."""
        rows = np.ascontiguousarray(rows)
        assert rows.dtype.itemsize == 4 and rows.ndim == 2
        out = np.zeros((rows.shape[0], out_words), np.uint32)
        L.check(self._lib.dust_hip_device_eval(self._h, fn, _ptr(rows), rows.shape[1], _ptr(out), out_words, rows.shape[0]))
        return out


class Model:
    """VoxGeometry + PaletteMaterial on the device."""

    def __init__(self, ctx, blocks, materials, palette, tree_extent_log2=8):
        self._ctx = ctx
        self._lib = ctx._lib
        blocks = np.ascontiguousarray(blocks, BLOCK_DTYPE)
        materials = np.ascontiguousarray(materials, np.uint8)
        pal = np.ascontiguousarray(np.asarray(palette, np.uint8).reshape(-1, 4)[:255])
        self._h = C.c_void_p()
        L.check(self._lib.dust_hip_model_create(ctx._h, _ptr(blocks), blocks.size, _ptr(materials), materials.size,
                                                _ptr(pal), tree_extent_log2, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_hip_model_destroy(self._h)
            self._h = None

    def set_voxels(self, xyz, values):
        """VoxGeometry::set on the device copy: xyz (n, 3) tree coordinates, values (n,) palette index or -1 to clear.
        Scenes instancing the model must be committed again afterwards."""
        xyz = np.ascontiguousarray(xyz, np.uint32).reshape(-1, 3)
        values = np.ascontiguousarray(values, np.int32).reshape(-1)
        assert len(values) == len(xyz)
        L.check(self._lib.dust_hip_model_set_voxels(self._h, _ptr(xyz), _ptr(values), len(values)))

    def get_voxels(self, xyz):
        """VoxGeometry::get: palette index per coordinate, -1 where the voxel is empty"""
        xyz = np.ascontiguousarray(xyz, np.uint32).reshape(-1, 3)
        out = np.zeros(len(xyz), np.int32)
        L.check(self._lib.dust_hip_model_get_voxels(self._h, _ptr(xyz), _ptr(out), len(out)))
        return out

    def read(self):
        """(blocks, materials) as they stand on the device"""
        nb, nm = C.c_uint32(), C.c_uint64()
        L.check(self._lib.dust_hip_model_info(self._h, C.byref(nb), C.byref(nm)))
        blocks = np.zeros(max(nb.value, 1), BLOCK_DTYPE)
        mats = np.zeros(max(nm.value, 1), np.uint8)
        L.check(self._lib.dust_hip_model_read(self._h, _ptr(blocks), len(blocks), _ptr(mats), len(mats)))
        return blocks[: nb.value], mats[: nm.value]


class Scene:
    def __init__(self, ctx):
        self._ctx = ctx
        self._lib = ctx._lib
        self._models = []
        self._h = C.c_void_p()
        L.check(self._lib.dust_hip_scene_create(ctx._h, C.byref(self._h)))

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_hip_scene_destroy(self._h)
            self._h = None

    def add_instance(self, model, obj_to_world, prev=None):
        m = np.ascontiguousarray(obj_to_world, np.float32).reshape(12)
        p = None if prev is None else np.ascontiguousarray(prev, np.float32).reshape(16)
        idx = C.c_uint32()
        L.check(self._lib.dust_hip_scene_add_instance(
            self._h, model._h, m.ctypes.data_as(C.POINTER(C.c_float)),
            None if p is None else p.ctypes.data_as(C.POINTER(C.c_float)), C.byref(idx)))
        self._models.append(model)
        return idx.value

    def set_transform(self, instance, obj_to_world, prev=None):
        """New transform for an instance (castle.rs:287-291 moves the teapot every frame); prev = last frame's
        object-to-world mat4, column-major (standard.rs:845-878), which the motion vectors are measured against."""
        m = np.ascontiguousarray(obj_to_world, np.float32).reshape(12)
        p = None if prev is None else np.ascontiguousarray(prev, np.float32).reshape(16)
        L.check(self._lib.dust_hip_scene_set_transform(
            self._h, instance, m.ctypes.data_as(C.POINTER(C.c_float)),
            None if p is None else p.ctypes.data_as(C.POINTER(C.c_float))))

    def commit(self):
        L.check(self._lib.dust_hip_scene_commit(self._h))


def top_level_build(boxes):
    """Host only: the grid and the slot order dust_hip_scene_commit builds over instance world boxes (n x 6: lo, hi) ->
    dict(dim, lo, cell, cells, items, ranges, slot_order, n_groups). For tests and tools."""
    lib = L.load()
    b = np.ascontiguousarray(boxes, np.float32).reshape(-1, 6)
    n = len(b)
    info = L.TopLevelInfo()
    info.struct_size = C.sizeof(L.TopLevelInfo)
    fp = b.ctypes.data_as(C.POINTER(C.c_float))
    L.check(lib.dust_hip_top_level_build(fp, n, C.byref(info), None, 0, None, 0, None, None))
    cells, items = np.zeros(info.n_cells, np.uint32), np.zeros(max(1, info.n_items), np.uint16)
    ranges, order = np.zeros((n, 2), np.uint32), np.zeros(n, np.uint32)
    L.check(lib.dust_hip_top_level_build(fp, n, C.byref(info), cells.ctypes.data_as(C.POINTER(C.c_uint32)), len(cells),
                                     items.ctypes.data_as(C.POINTER(C.c_uint16)), len(items), ranges.ctypes.data_as(C.POINTER(C.c_uint32)),
                                     order.ctypes.data_as(C.POINTER(C.c_uint32))))
    return {"dim": tuple(info.dim), "lo": np.array(info.lo[:], np.float32), "cell": np.array(info.cell[:], np.float32), "cells": cells,
            "items": items[:info.n_items], "ranges": ranges, "slot_order": order, "n_groups": info.n_groups}


def sky_struct(sky):
    """56 baked sky floats -> DustHipSky"""
    s = L.Sky()
    s.state[:] = np.asarray(sky, np.float32).reshape(56).tolist()
    return s


def _config_from_environment():
    """the production knobs of DustHipPipelineConfig as A/B scripts and the stress drivers spell them (read HERE, not by the library)"""
    import os
    e, cfg = os.environ, {}
    if "DUST_HIP_RAY_STREAM" in e:
        cfg["gi_path"] = "streams"
    elif "DUST_HIP_PACKET_GI" in e:
        cfg["gi_path"] = "packets"
    if "DUST_HIP_NO_SIDE_STREAM" in e:
        cfg["side_stream"] = "off"
    if e.get("DUST_HIP_SIDE_SHARE") and int(e["DUST_HIP_SIDE_SHARE"]) > 0:   # (0: calibrated, the default)
        cfg["side_share"] = min(90, max(5, int(e["DUST_HIP_SIDE_SHARE"])))
    if e.get("DUST_HIP_RESERVE_BLOCKS"):
        cfg["reserve_blocks"] = int(e["DUST_HIP_RESERVE_BLOCKS"])
    return cfg


class StandardPipeline:
    """StandardPipeline (crates/render/src/pipeline/standard.rs:51-60, :222-240) + its GBuffer (:881-917)."""

    PRIMARY_RAYTYPE = 0
    AMBIENT_OCCLUSION_RAYTYPE = 1
    FINAL_GATHER_RAYTYPE = 2
    SURFEL_RAYTYPE = 3

    def __init__(self, ctx, width, height, **config):
        """config: DustHipPipelineConfig fields (see configure). A/B runs and the stress drivers may also name them in the environment
        of THIS shim -- DUST_HIP_RAY_STREAM / DUST_HIP_PACKET_GI (gi_path), DUST_HIP_NO_SIDE_STREAM, DUST_HIP_SIDE_SHARE,
        DUST_HIP_RESERVE_BLOCKS --: the library itself reads no production knob from the environment."""
        self._ctx = ctx
        self._lib = ctx._lib
        self.width, self.height = width, height
        self._h = C.c_void_p()
        L.check(self._lib.dust_hip_pipeline_create(ctx._h, width, height, C.byref(self._h)))
        cfg = _config_from_environment()
        cfg.update(config)
        if cfg:
            self.configure(**cfg)

    def configure(self, reserve_blocks=None, gi_path=None, side_stream=None, side_share=None, frames_in_flight=None, in_flight_slots=None):
        """dust_hip_pipeline_configure: the fields given replace the pipeline's current ones.
        gi_path: "auto" | "packets" | "streams"; side_stream: "auto" | "off"; in_flight_slots: "share" | "all"; reserve_blocks: int or "auto"."""
        c = self.get_config(raw=True)
        if reserve_blocks is not None:
            c.reserve_blocks = L.RESERVE_AUTO if reserve_blocks == "auto" else int(reserve_blocks)
        if gi_path is not None:
            c.gi_path = {"auto": L.GI_PATH_AUTO, "packets": L.GI_PATH_PACKETS, "streams": L.GI_PATH_STREAMS}.get(gi_path, gi_path)
        if side_stream is not None:
            c.side_stream = {"auto": L.SIDE_STREAM_AUTO, "off": L.SIDE_STREAM_OFF}.get(side_stream, side_stream)
        if side_share is not None:
            c.side_share = int(side_share)
        if in_flight_slots is not None:
            c.in_flight_slots = {"share": L.IN_FLIGHT_SHARE, "all": L.IN_FLIGHT_ALL}.get(in_flight_slots, in_flight_slots)
        c.frames_in_flight = int(frames_in_flight) if frames_in_flight is not None else 0
        L.check(self._lib.dust_hip_pipeline_configure(self._h, C.byref(c)))

    def get_config(self, raw=False):
        c = L.PipelineConfig(C.sizeof(L.PipelineConfig))
        L.check(self._lib.dust_hip_pipeline_get_config(self._h, C.byref(c)))
        if raw:
            return c
        return {"reserve_blocks": "auto" if c.reserve_blocks == L.RESERVE_AUTO else c.reserve_blocks,
                "gi_path": ("auto", "packets", "streams")[c.gi_path], "side_stream": ("auto", "off")[c.side_stream], "side_share": c.side_share,
                "frames_in_flight": c.frames_in_flight, "in_flight_slots": ("share", "all")[c.in_flight_slots]}

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_hip_pipeline_destroy(self._h)
            self._h = None

    def set_noise(self, texture, texels):
        t = np.ascontiguousarray(texels, np.uint8)
        layers = t.size // (128 * 128 * (1 if texture == 0 else 4))
        L.check(self._lib.dust_hip_pipeline_set_noise(self._h, texture, _ptr(t), layers))

    def render(self, scene, camera, sky, passes, frame_index=1, rand=0, rows=(0, 0), surfel_shard=(0, 0)):
        """sky: 56 floats, or a DustHipSky made once with api.sky_struct() (a frame loop: the conversion is most of this call's host time).
        surfel_shard: (rank, world) -- with PASS_SURFEL | PASS_GI_SHARDED the pass only traces rank's share of the ordered pool (Comm.gi_surfel_exchange completes it)"""
        self.frame_call(scene, camera, sky, passes, frame_index, rand, rows, surfel_shard)()

    def frame_call(self, scene, camera, sky, passes, frame_index=1, rand=0, rows=(0, 0), surfel_shard=(0, 0)):
        """-> a callable that makes ONE dust_hip_render_frame call with the arguments marshalled HERE (a frame loop that knows its frames ahead builds
        the calls first: what is left per call is the C entry point -- a band-sized step of an N-GPU job is 30 us, of which render()'s marshalling was a third)"""
        s = sky if isinstance(sky, L.Sky) else sky_struct(sky)
        fp = L.FrameParams(C.sizeof(L.FrameParams), passes, frame_index, rand & 0xFFFFFFFF, rows[0], rows[1], surfel_shard[0], surfel_shard[1])
        fn, h, sh, cam_ref, sky_ref, fp_ref, check = self._lib.dust_hip_render_frame, self._h, scene._h, C.byref(camera), C.byref(s), C.byref(fp), L.check

        def call(_keep=(self, scene, camera, s, fp)):
            check(fn(h, sh, cam_ref, sky_ref, fp_ref))
        return call

    @staticmethod
    def render_frames(pipes, scene, cameras, skies, passes, frame_indices, rands, rows=(0, 0), moves=None):
        """frames_call(...)(): see there"""
        StandardPipeline.frames_call(pipes, scene, cameras, skies, passes, frame_indices, rands, rows, moves)()

    @staticmethod
    def frames_call(pipes, scene, cameras, skies, passes, frame_indices, rands, rows=(0, 0), moves=None):
        """-> a callable that makes ONE dust_hip_render_frames call with the arguments marshalled HERE (a frame loop that knows its frames ahead -- an
        offline render, bench.py's timed region -- builds the calls first: what is left per call is the C entry point).
        dust_hip_render_frames: frame i -- cameras[i], skies[i], frame_indices[i], rands[i] -- into pipes[i], the results of len(pipes)
        render() calls in that order; primary + AO frames of distinct pipelines of one context share ONE persistent launch (up to 8 frames each).
        cameras / skies: one per frame, or a single Camera / Sky for all of them. moves: per frame None or a list of (instance id, obj_to_world[12],
        prev_obj_to_world mat4[16] or None): what Scene.set_transform + Scene.commit would do before that frame."""
        n = len(pipes)
        # (a frame loop hands over ctypes arrays it made once -- at least n entries each: building them is most of this call's host time)
        if isinstance(cameras, C.Array) and getattr(cameras, "_type_", None) is L.Camera:
            cams = cameras
        else:
            cams = (L.Camera * n)(*[(cameras if isinstance(cameras, L.Camera) else cameras[i]) for i in range(n)])
        if isinstance(skies, C.Array) and getattr(skies, "_type_", None) is L.Sky:
            sk = skies
        else:
            one_sky = isinstance(skies, L.Sky) or (not isinstance(skies, (list, tuple)))
            sk = (L.Sky * n)(*[((skies if isinstance(skies, L.Sky) else sky_struct(skies)) if one_sky else
                                (skies[i] if isinstance(skies[i], L.Sky) else sky_struct(skies[i]))) for i in range(n)])
        assert len(cams) >= n and len(sk) >= n
        fps = (L.FrameParams * n)(*[L.FrameParams(C.sizeof(L.FrameParams), passes, int(frame_indices[i]), int(rands[i]) & 0xFFFFFFFF, rows[0], rows[1], 0, 0)
                                    for i in range(n)])
        hs = (C.c_void_p * n)(*[p._h for p in pipes])
        mv, keep = None, []
        if moves is not None:
            mv = (L.FrameMoves * n)()
            for i in range(n):
                ms = moves[i] or []
                if not ms:
                    continue
                ids = np.ascontiguousarray([m[0] for m in ms], np.uint32)
                xf = np.ascontiguousarray([np.asarray(m[1], np.float32).reshape(12) for m in ms], np.float32)
                have_prev = all(len(m) > 2 and m[2] is not None for m in ms)
                pv = np.ascontiguousarray([np.asarray(m[2], np.float32).reshape(16) for m in ms], np.float32) if have_prev else None
                keep += [ids, xf, pv]
                mv[i].n = len(ms)
                mv[i].instance_ids = ids.ctypes.data_as(C.POINTER(C.c_uint32))
                mv[i].obj_to_world = xf.ctypes.data_as(C.POINTER(C.c_float))
                mv[i].prev_obj_to_world = pv.ctypes.data_as(C.POINTER(C.c_float)) if pv is not None else None
        fn, sh, check = pipes[0]._lib.dust_hip_render_frames, scene._h, L.check

        def call(_keep=(keep, pipes, scene)):   # (the arrays above stay alive with the closure)
            check(fn(n, hs, sh, cams, sk, fps, mv))
        return call

    def pass_stats(self, index):
        st = L.PassStats()
        L.check(self._lib.dust_hip_pipeline_pass_stats(self._h, index, C.byref(st)))
        return st

    def read_plane(self, plane):
        dt, ch = PLANE_DTYPES[plane]
        shape = (self.height, self.width, ch) if ch > 1 else (self.height, self.width)
        out = np.zeros(shape, dt)
        L.check(self._lib.dust_hip_pipeline_read_plane(self._h, plane, _ptr(out), out.nbytes))
        return out

    def kernel_times(self, mark=True):
        """(ms_sum[4], launches[4]) per pass kind (primary/fused, AO, final gather, surfel pass) since the last mark"""
        ms, n = (C.c_float * 4)(), (C.c_uint32 * 4)()
        L.check(self._lib.dust_hip_pipeline_kernel_times(self._h, 1 if mark else 0, ms, n))
        return list(ms), list(n)

    def mark_kernel_times(self):
        """start of a timed region: later kernel_times() calls sum the launches from here on (no wait, nothing read back)"""
        L.check(self._lib.dust_hip_pipeline_kernel_times(self._h, 1, None, None))

    def tile_costs(self, pass_kind=0):
        """cycles per tile of the pass's last launch, shape (tiles_y, tiles_x); None before the first launch"""
        tx, ty = C.c_uint32(), C.c_uint32()
        L.check(self._lib.dust_hip_pipeline_tile_costs(self._h, pass_kind, None, 0, C.byref(tx), C.byref(ty)))
        if tx.value == 0:
            return None
        out = np.zeros((ty.value, tx.value), np.uint32)
        L.check(self._lib.dust_hip_pipeline_tile_costs(self._h, pass_kind, _ptr(out), out.size, C.byref(tx), C.byref(ty)))
        return out

    def plane_device_ptr(self, plane):
        p, n = C.c_void_p(), C.c_size_t()
        L.check(self._lib.dust_hip_pipeline_plane_device_ptr(self._h, plane, C.byref(p), C.byref(n)))
        return p.value, n.value

    def bind_plane(self, plane, device_ptr, nbytes):
        """Redirect a plane to caller-owned device memory (device_ptr = 0 / None: back to the pipeline's own storage)."""
        L.check(self._lib.dust_hip_pipeline_bind_plane(self._h, plane, C.c_void_p(device_ptr or None), nbytes))

    def clear(self):
        L.check(self._lib.dust_hip_pipeline_clear(self._h))

    def set_frames_in_flight(self, n):
        """the caller keeps n frames in flight on this device (a pipeline and a context each): launches take 1/n of the workgroup slots"""
        L.check(self._lib.dust_hip_pipeline_set_frames_in_flight(self._h, n))

    def set_denoiser(self, max_accumulated_frames=30, disocclusion_threshold=0.01, antilag_sigma_scale=2.0, antilag_power=0.8,
                     max_blur_radius=15.0):
        """ReblurSettings (nrd.rs:768-785) for DUST_PASS_DENOISE"""
        dp = L.DenoiseParams(C.sizeof(L.DenoiseParams), max_accumulated_frames, disocclusion_threshold, antilag_sigma_scale,
                             antilag_power, max_blur_radius)
        L.check(self._lib.dust_hip_pipeline_set_denoiser(self._h, C.byref(dp)))

    def restart_denoiser(self):
        """DenoiserEvent::Restart"""
        L.check(self._lib.dust_hip_pipeline_restart_denoiser(self._h))

    def tone_map(self, transfer_function=1, conversion=None, min_log=-6.0, max_log=8.5, time_coefficient=0.2):
        """AutoExposurePipeline + ToneMappingPipeline on the denoised plane -> PLANE_OUTPUT."""
        tp = L.ToneMapParams()
        tp.struct_size = C.sizeof(L.ToneMapParams)
        tp.transfer_function = transfer_function
        tp.color_space_conversion[:] = (color_space_conversion() if conversion is None else np.asarray(conversion, np.float32)).tolist()
        tp.min_log_luminance, tp.max_log_luminance, tp.time_coefficient = min_log, max_log, time_coefficient
        L.check(self._lib.dust_hip_tone_map(self._h, C.byref(tp)))

    def exposure(self, set_to=None):
        out = C.c_float()
        s = None if set_to is None else C.byref(C.c_float(set_to))
        L.check(self._lib.dust_hip_pipeline_exposure(self._h, C.byref(out), s))
        return out.value

    def configure_gi(self, hash_capacity=32 * 1024 * 1024, surfel_pool_size=720 * 480):
        self._gi = (hash_capacity, surfel_pool_size)
        L.check(self._lib.dust_hip_pipeline_configure_gi(self._h, hash_capacity, surfel_pool_size))

    def gi_exchange(self, padded_rows=None):
        """Device buffers of the multi-GPU GI exchange (dust_hip.h): a _lib.GiExchange with raw device pointers."""
        out = L.GiExchange()
        out.struct_size = C.sizeof(L.GiExchange)
        L.check(self._lib.dust_hip_pipeline_gi_exchange(self._h, padded_rows or self.height, C.byref(out)))
        self._gi = (self._gi[0] if getattr(self, "_gi", None) else 32 * 1024 * 1024, out.pool_size)
        return out

    def gi_export(self, row_begin, row_end):
        L.check(self._lib.dust_hip_gi_export(self._h, row_begin, row_end))

    def gi_import(self, row_begin, row_end, frame_index):
        L.check(self._lib.dust_hip_gi_import(self._h, row_begin, row_end, frame_index))

    def gi_surfel_finish(self, frame_index):
        """completes a sharded surfel trace WITHOUT an exchange (dust_hip_gi_surfel_exchange_run with no communicator): a world of one, or
        one emulated rank of N -- the other ranks' records are whatever the staging arrays hold"""
        L.check(self._lib.dust_hip_gi_surfel_exchange_run(self._h, None, frame_index))

    def read_gi(self):
        """(hash entries as uint32[capacity+2, 3], surfel pool as structured array)"""
        cap, pool = getattr(self, "_gi", None) or (32 * 1024 * 1024, 720 * 480)  # the defaults of an implicit configuration
        h = np.zeros((cap + 2, 3), np.uint32)
        L.check(self._lib.dust_hip_pipeline_read_gi(self._h, 0, _ptr(h), h.nbytes))
        s = np.zeros(pool, SURFEL_DTYPE)
        L.check(self._lib.dust_hip_pipeline_read_gi(self._h, 1, _ptr(s), s.nbytes))
        return h, s


def _write_gi(self, hash_entries, pool):
    """Restore a state saved with read_gi() into a pipeline configured with the same capacity and pool size."""
    h = np.ascontiguousarray(hash_entries, np.uint32)
    s = np.ascontiguousarray(pool, SURFEL_DTYPE)
    L.check(self._lib.dust_hip_pipeline_write_gi(self._h, 0, _ptr(h), h.nbytes))
    L.check(self._lib.dust_hip_pipeline_write_gi(self._h, 1, _ptr(s), s.nbytes))


StandardPipeline.write_gi = _write_gi


class Comm:
    """One rank's end of the multi-GPU partition: an RCCL communicator on the context's device (create), or the ranks of a loopback
    group on one device (local). Band gather and GI exchange run inside the library (comm.hip)."""

    def __init__(self, ctx, handle):
        self._ctx, self._lib, self._h = ctx, ctx._lib, handle

    @staticmethod
    def unique_id():
        buf = (C.c_uint8 * 128)()
        L.check(L.load().dust_hip_comm_unique_id(C.cast(buf, C.c_void_p)))
        return bytes(buf)

    @classmethod
    def create(cls, ctx, rank, world, unique_id: bytes):
        h = C.c_void_p()
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        L.check(ctx._lib.dust_hip_comm_create(ctx._h, rank, world, C.cast(buf, C.c_void_p), C.byref(h)))
        return cls(ctx, h)

    @classmethod
    def local(cls, ctx, world):
        hs = (C.c_void_p * world)()
        L.check(ctx._lib.dust_hip_comm_create_local(ctx._h, world, hs))
        return [cls(ctx, C.c_void_p(h)) for h in hs]

    def __del__(self):
        if getattr(self, "_h", None):
            self._lib.dust_hip_comm_destroy(self._h)
            self._h = None

    def info(self):
        r, w, l = C.c_uint32(), C.c_uint32(), C.c_uint32()
        L.check(self._lib.dust_hip_comm_info(self._h, C.byref(r), C.byref(w), C.byref(l)))
        return r.value, w.value, bool(l.value)

    def _cuts(self, cuts):
        """world + 1 row indices as the C array the library reads cuts[0..world] of (a shorter list would be an out-of-bounds host read)"""
        world = self.info()[1]
        if len(cuts) != world + 1:
            raise ValueError(f"band cuts: {len(cuts)} entries for a communicator of {world} ranks (want world + 1)")
        return cuts if isinstance(cuts, C.Array) else (C.c_uint32 * len(cuts))(*[int(v) for v in cuts])

    def gather_bands(self, pipe, plane, cuts, root=0, dst_ptr=None, dst_bytes=0):
        """-> the gather's ticket (wait(ticket) before its source target or destination is used again)"""
        c = self._cuts(cuts)
        t = C.c_uint64()
        L.check(self._lib.dust_hip_gather_bands(pipe._h, self._h, plane, c, root, C.c_void_p(dst_ptr) if dst_ptr else None, dst_bytes, C.byref(t)))
        return t.value

    def gather_planes(self, pipe, planes, cuts, root=0):
        """several planes (an iterable of DUST_PLANE_* indices) in one collective, each into the root pipeline's own plane -> ticket"""
        c = self._cuts(cuts)
        mask = 0
        for pl in planes:
            mask |= 1 << int(pl)
        t = C.c_uint64()
        L.check(self._lib.dust_hip_gather_planes(pipe._h, self._h, mask, c, root, C.byref(t)))
        return t.value

    def gi_exchange(self, pipe, row_begin, row_end, band_rows, frame_index):
        L.check(self._lib.dust_hip_gi_exchange_run(pipe._h, self._h, row_begin, row_end, band_rows, frame_index))

    def gi_surfel_exchange(self, pipe, frame_index):
        """completes a surfel pass whose trace was sharded (render(..., surfel_shard=(rank, world))): all-gather of the records, stamps, ordered apply"""
        L.check(self._lib.dust_hip_gi_surfel_exchange_run(pipe._h, self._h, frame_index))

    def wait(self, ticket=0):
        L.check(self._lib.dust_hip_comm_wait(self._h, ticket))

    def sync(self):
        L.check(self._lib.dust_hip_comm_sync(self._h))


def load_png_array(data: bytes):
    """PNG / APNG -> array (layers, height, width, channels), uint8 (big-endian uint16 for 16-bit files): the reference's
    PngLoader (rhyolite_bevy/src/loaders/png.rs:70-200). RGB comes back as RGBA with a zero fourth channel."""
    lib = L.load()
    buf = (C.c_uint8 * len(data)).from_buffer_copy(data)
    info = L.PngInfo()
    out = C.POINTER(C.c_uint8)()
    L.check(lib.dust_png_load_array(C.cast(buf, C.c_void_p), len(data), C.byref(info), C.byref(out)))
    try:
        n = info.layers * info.height * info.width * info.channels * info.bytes_per_channel
        raw = np.ctypeslib.as_array(out, shape=(n,)).copy()
    finally:
        lib.dust_vox_free(out)
    dt = np.uint8 if info.bytes_per_channel == 1 else np.dtype(">u2")
    return raw.view(dt).reshape(info.layers, info.height, info.width, info.channels)

