"""ctypes binding of oracle/_build/liboracle.so -- the CPU checker. Test infrastructure only."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.environ.get("DUST_ORACLE_LIB") or os.path.join(ROOT, "oracle", "_build", "liboracle.so")  # (override: a sanitizer build, tools/asan_host_check.sh)

BLOCK_DTYPE = np.dtype([("x", "<u2"), ("y", "<u2"), ("z", "<u2"), ("w", "<u2"), ("mask", "<u8"),
                        ("material_ptr", "<u4"), ("avg_albedo", "<u4")])


class OrcModel(C.Structure):
    _fields_ = [("blocks", C.c_void_p), ("n_blocks", C.c_uint32), ("materials", C.POINTER(C.c_uint8)),
                ("n_materials", C.c_uint64), ("palette", C.c_uint8 * (255 * 4)), ("extent", C.c_uint32)]


class OrcInstance(C.Structure):
    _fields_ = [("model", C.c_uint32), ("obj_to_world", C.c_float * 12), ("prev_obj_to_world", C.c_float * 16)]


class OrcCamera(C.Structure):
    _fields_ = [("col0", C.c_float * 3), ("col1", C.c_float * 3), ("col2", C.c_float * 3), ("pos", C.c_float * 3),
                ("tan_half_fov", C.c_float), ("far_", C.c_float), ("near_", C.c_float)]


class OrcSky(C.Structure):
    _fields_ = [("v", C.c_float * 56)]


class OrcGBuffer(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("illuminance", C.c_void_p), ("denoised", C.c_void_p),
                ("albedo", C.c_void_p), ("normal", C.c_void_p), ("depth", C.c_void_p), ("motion", C.c_void_p),
                ("voxel_id", C.c_void_p)]


class OrcRayStats(C.Structure):
    _fields_ = [("rays", C.c_uint64), ("instances_tested", C.c_uint64), ("upper_descents", C.c_uint64),
                ("mid_descents", C.c_uint64), ("bricks_tested", C.c_uint64), ("hits", C.c_uint64)]


class OrcDenoise(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("frame_index", C.c_uint32), ("have_history", C.c_uint32),
                ("cam", OrcCamera), ("prev", OrcCamera), ("max_accumulated_frames", C.c_uint32),
                ("disocclusion_threshold", C.c_float), ("antilag_sigma_scale", C.c_float), ("antilag_power", C.c_float),
                ("max_blur_radius", C.c_float),
                ("illuminance", C.c_void_p), ("denoised", C.c_void_p), ("normal", C.c_void_p), ("depth", C.c_void_p),
                ("motion", C.c_void_p), ("voxel_id", C.c_void_p),
                ("hist_in_accum", C.c_void_p), ("hist_in_depth", C.c_void_p), ("hist_in_normal", C.c_void_p), ("hist_in_id", C.c_void_p),
                ("hist_out_accum", C.c_void_p), ("hist_out_depth", C.c_void_p), ("hist_out_normal", C.c_void_p), ("hist_out_id", C.c_void_p)]


ORC_MODE_BRUTE, ORC_MODE_HIER = 0, 1
_lib = None


def lib():
    global _lib
    if _lib is None:
        l = C.CDLL(SO)
        l.orc_tree_new.restype = C.c_void_p
        l.orc_tree_new.argtypes = [C.POINTER(C.c_uint32), C.c_int]
        l.orc_tree_free.argtypes = [C.c_void_p]
        l.orc_tree_set.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int]
        l.orc_tree_get.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        l.orc_tree_iter.restype = C.c_size_t
        l.orc_tree_iter.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        l.orc_tree_iter_leaf.restype = C.c_size_t
        l.orc_tree_iter_leaf.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        l.orc_tree_meta_mask.restype = C.c_uint32
        l.orc_tree_meta_mask.argtypes = [C.c_void_p]
        l.orc_tree_root_level.restype = C.c_uint32
        l.orc_tree_root_level.argtypes = [C.c_void_p]
        l.orc_lca_level.restype = C.c_uint32
        l.orc_lca_level.argtypes = [C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_uint32, C.c_uint32]
        l.orc_accessor_new.restype = C.c_void_p
        l.orc_accessor_new.argtypes = [C.c_void_p]
        l.orc_accessor_free.argtypes = [C.c_void_p]
        l.orc_accessor_get.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32]
        l.orc_pool_new.restype = C.c_void_p
        l.orc_pool_new.argtypes = [C.c_size_t, C.c_uint]
        l.orc_pool_free_pool.argtypes = [C.c_void_p]
        l.orc_pool_alloc.restype = C.c_uint32
        l.orc_pool_alloc.argtypes = [C.c_void_p]
        l.orc_pool_free.argtypes = [C.c_void_p, C.c_uint32]
        l.orc_pool_num_chunks.restype = C.c_size_t
        l.orc_pool_num_chunks.argtypes = [C.c_void_p]
        l.orc_bitmask_set.argtypes = [C.c_void_p, C.c_size_t, C.c_int]
        l.orc_bitmask_iter.restype = C.c_size_t
        l.orc_bitmask_iter.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        l.orc_model_build.restype = C.POINTER(OrcModel)
        l.orc_model_build.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32), C.c_void_p, C.POINTER(C.c_uint32), C.c_int]
        l.orc_model_free.argtypes = [C.POINTER(OrcModel)]
        l.orc_scene_new.restype = C.c_void_p
        l.orc_scene_free.argtypes = [C.c_void_p]
        l.orc_scene_add_model.restype = C.c_uint32
        l.orc_scene_add_model.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint32]
        l.orc_scene_add_instance.restype = C.c_uint32
        l.orc_scene_add_instance.argtypes = [C.c_void_p, C.POINTER(OrcInstance)]
        l.orc_scene_commit.argtypes = [C.c_void_p]
        l.orc_dda.argtypes = [C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_float,
                              C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_int)]
        l.orc_dda_rough.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        l.orc_camera_ray_dir.argtypes = [C.POINTER(OrcCamera), C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_float)]
        l.orc_trace.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float,
                                C.c_float, C.POINTER(C.c_float), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                C.POINTER(C.c_uint32), C.POINTER(OrcRayStats)]
        l.orc_pass_primary.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcCamera), C.POINTER(OrcSky), C.POINTER(OrcGBuffer),
                                       C.c_uint32, C.c_uint32, C.POINTER(OrcRayStats)]
        l.orc_pass_ao.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcCamera), C.POINTER(OrcSky), C.POINTER(OrcGBuffer),
                                  C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(OrcRayStats), C.POINTER(OrcRayStats)]
        l.orc_gi_new.restype = C.c_void_p
        l.orc_gi_new.argtypes = [C.c_uint32, C.c_uint32]
        l.orc_gi_free.argtypes = [C.c_void_p]
        l.orc_gi_hash_ptr.restype = C.c_void_p
        l.orc_gi_hash_ptr.argtypes = [C.c_void_p]
        l.orc_gi_pool_ptr.restype = C.c_void_p
        l.orc_gi_pool_ptr.argtypes = [C.c_void_p]
        l.orc_hash_fingerprint.restype = C.c_uint32
        l.orc_hash_fingerprint.argtypes = [C.POINTER(C.c_int32), C.c_uint32]
        l.orc_hash_location.restype = C.c_uint32
        l.orc_hash_location.argtypes = [C.POINTER(C.c_int32), C.c_uint32, C.c_uint32]
        l.orc_hash_insert.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.POINTER(C.c_float), C.c_uint32]
        l.orc_hash_get.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_uint32, C.c_uint32, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]
        l.orc_pass_final_gather.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcCamera), C.POINTER(OrcSky), C.POINTER(OrcGBuffer),
                                            C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32,
                                            C.POINTER(OrcRayStats)]
        l.orc_pass_surfel.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcSky), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                      C.c_void_p, C.POINTER(OrcRayStats), C.POINTER(OrcRayStats)]
        l.orc_pass_final_gather_mt.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcCamera), C.POINTER(OrcSky), C.POINTER(OrcGBuffer),
                                               C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32,
                                               C.POINTER(OrcRayStats)]
        l.orc_pass_surfel_mt.argtypes = [C.c_void_p, C.c_int, C.POINTER(OrcSky), C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32,
                                         C.c_void_p, C.c_uint32, C.POINTER(OrcRayStats), C.POINTER(OrcRayStats)]
        l.orc_exposure_histogram.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_void_p]
        l.orc_exposure_average.restype = C.c_float
        l.orc_exposure_average.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.c_float, C.c_float, C.c_float]
        l.orc_tone_map.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_float, C.POINTER(C.c_float), C.c_uint32, C.c_void_p]
        l.orc_denoise.argtypes = [C.POINTER(OrcDenoise)]
        l.orc_pack_rgb10a2.restype = C.c_uint32
        l.orc_pack_rgb10a2.argtypes = [C.POINTER(C.c_float)]
        l.orc_unpack_rgb10a2.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        l.orc_f32_to_f16.restype = C.c_uint16
        l.orc_f32_to_f16.argtypes = [C.c_float]
        l.orc_f16_to_f32.restype = C.c_float
        l.orc_f16_to_f32.argtypes = [C.c_uint16]
        l.orc_nrd_pack_normal.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_float, C.POINTER(C.c_float)]
        l.orc_nrd_unpack_normal.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.orc_normal2faceid.restype = C.c_uint32
        l.orc_normal2faceid.argtypes = [C.POINTER(C.c_float)]
        l.orc_cubed_normalize.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.orc_rotate_by_normal.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.orc_logluv_encode.restype = C.c_uint32
        l.orc_logluv_encode.argtypes = [C.POINTER(C.c_float)]
        l.orc_logluv_decode.argtypes = [C.c_uint32, C.POINTER(C.c_float)]
        l.orc_pcg.restype = C.c_uint32
        l.orc_pcg.argtypes = [C.c_uint32]
        l.orc_xxhash32.restype = C.c_uint32
        l.orc_xxhash32.argtypes = [C.c_uint32]
        l.orc_sky_radiance.argtypes = [C.POINTER(OrcSky), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        l.orc_sun_radiance.argtypes = [C.POINTER(OrcSky), C.POINTER(C.c_float), C.POINTER(C.c_float)]
        _lib = l
    return _lib


def f3(v):
    return (C.c_float * 3)(*[float(x) for x in v])


def model_build(xyzi, size, palette256, log2s=(4, 2, 2)):
    l = lib()
    xyzi = np.ascontiguousarray(xyzi, np.uint8).reshape(-1, 4)
    pal = np.ascontiguousarray(palette256, np.uint8).reshape(256, 4)
    sz = (C.c_uint32 * 3)(*[int(v) for v in size])
    lg = (C.c_uint32 * len(log2s))(*log2s)
    m = l.orc_model_build(xyzi.ctypes.data_as(C.c_void_p), xyzi.shape[0], sz, pal.ctypes.data_as(C.c_void_p), lg, len(log2s))
    try:
        n = m.contents.n_blocks
        blocks = np.frombuffer(C.string_at(m.contents.blocks, n * 24), BLOCK_DTYPE).copy() if n else np.zeros(0, BLOCK_DTYPE)
        nm = min(m.contents.n_materials, xyzi.shape[0])
        mats = np.frombuffer(C.string_at(m.contents.materials, nm), np.uint8).copy() if nm else np.zeros(0, np.uint8)
    finally:
        l.orc_model_free(m)
    return blocks, mats


class Scene:
    """Oracle scene over flattened models (blocks, materials, palette) and instances."""

    def __init__(self):
        self.l = lib()
        self.h = self.l.orc_scene_new()
        self.keep = []

    def __del__(self):
        if getattr(self, "h", None):
            self.l.orc_scene_free(self.h)
            self.h = None

    def add_model(self, blocks, materials, palette, extent=256):
        b = np.ascontiguousarray(blocks, BLOCK_DTYPE)
        m = np.ascontiguousarray(materials, np.uint8)
        p = np.ascontiguousarray(np.asarray(palette, np.uint8).reshape(-1, 4)[:255])
        self.keep += [b, m, p]
        return self.l.orc_scene_add_model(self.h, b.ctypes.data_as(C.c_void_p), b.size, m.ctypes.data_as(C.c_void_p),
                                          m.size, p.ctypes.data_as(C.c_void_p), extent)

    def add_instance(self, model, obj_to_world, prev=None):
        inst = OrcInstance()
        inst.model = model
        o2w = np.asarray(obj_to_world, np.float32).reshape(12)
        inst.obj_to_world[:] = o2w.tolist()
        if prev is None:
            m4 = np.eye(4, dtype=np.float32)
            m4[:3, :] = o2w.reshape(3, 4)
            prev = m4.T.reshape(16)  # column-major
        inst.prev_obj_to_world[:] = np.asarray(prev, np.float32).reshape(16).tolist()
        return self.l.orc_scene_add_instance(self.h, C.byref(inst))

    def commit(self):
        self.l.orc_scene_commit(self.h)

    def trace(self, mode, raytype, any_hit, o, d, tmin, tmax, stats=None):
        t, inst, block, vox = C.c_float(), C.c_uint32(), C.c_uint32(), C.c_uint32()
        hit = self.l.orc_trace(self.h, mode, raytype, any_hit, f3(o), f3(d), tmin, tmax, C.byref(t), C.byref(inst),
                               C.byref(block), C.byref(vox), None if stats is None else C.byref(stats))
        return (t.value, inst.value, block.value, vox.value) if hit else None


class GBuffer:
    def __init__(self, w, h):
        self.w, self.h = w, h
        self.illuminance = np.zeros((h, w, 4), np.uint16)
        self.denoised = np.zeros((h, w, 4), np.uint16)
        self.albedo = np.zeros((h, w), np.uint32)
        self.normal = np.zeros((h, w), np.uint32)
        self.depth = np.zeros((h, w), np.float32)
        self.motion = np.zeros((h, w, 4), np.uint16)
        self.voxel_id = np.zeros((h, w), np.uint32)
        self.c = OrcGBuffer(w, h, *[a.ctypes.data_as(C.c_void_p) for a in
                                   (self.illuminance, self.denoised, self.albedo, self.normal, self.depth, self.motion,
                                    self.voxel_id)])


def camera_from(cam):
    """dust_amd._lib.Camera -> OrcCamera (same members)."""
    c = OrcCamera()
    c.col0[:] = list(cam.view_col0)
    c.col1[:] = list(cam.view_col1)
    c.col2[:] = list(cam.view_col2)
    c.pos[:] = list(cam.position)
    c.tan_half_fov, c.far_, c.near_ = cam.tan_half_fov, cam.far_, cam.near_
    return c


def sky_from(state):
    s = OrcSky()
    s.v[:] = np.asarray(state, np.float32).reshape(56).tolist()
    return s


HASH_DTYPE = np.dtype([("fingerprint", "<u4"), ("radiance", "<u4"), ("last_accessed_frame", "<u2"), ("sample_count", "<u2")])
SURFEL_DTYPE = np.dtype([("pos", "<f4", 3), ("direction", "<u4")])


class GI:
    """Oracle spatial hash + surfel pool."""

    def __init__(self, capacity, pool_size):
        self.l = lib()
        self.capacity, self.pool_size = capacity, pool_size
        self.h = self.l.orc_gi_new(capacity, pool_size)

    def __del__(self):
        if getattr(self, "h", None):
            self.l.orc_gi_free(self.h)
            self.h = None

    def hash(self):
        n = (self.capacity + 2) * 12
        return np.frombuffer(C.string_at(self.l.orc_gi_hash_ptr(self.h), n), HASH_DTYPE).copy()

    def pool(self):
        return np.frombuffer(C.string_at(self.l.orc_gi_pool_ptr(self.h), self.pool_size * 16), SURFEL_DTYPE).copy()

    def insert(self, pos, direction, value, frame):
        self.l.orc_hash_insert(self.h, (C.c_int32 * 3)(*pos), direction, f3(value), frame)

    def get(self, pos, direction, frame):
        out, cnt = (C.c_float * 3)(), C.c_uint32()
        f = self.l.orc_hash_get(self.h, (C.c_int32 * 3)(*pos), direction, frame, out, C.byref(cnt))
        return bool(f), list(out), cnt.value


class Denoiser:
    """The oracle's copy of the spatiotemporal filter with its own history (oracle/denoise.c)."""

    def __init__(self, w, h, max_frames=30, disocclusion=0.01, sigma=2.0, power=0.8, radius=15.0):
        self.w, self.h = w, h
        self.params = (max_frames, disocclusion, sigma, power, radius)
        self.hist = [dict(accum=np.zeros((h, w, 4), np.float32), depth=np.zeros((h, w), np.float32),
                          normal=np.zeros((h, w), np.uint32), id=np.zeros((h, w), np.uint32)) for _ in range(2)]
        self.parity = 0
        self.have = False
        self.prev_cam = None

    def frame(self, planes, cam, frame_index):
        """planes: dict with illuminance, denoised (in: sky for misses), normal, depth, motion, voxel_id (numpy, as read from a
        G-buffer). Returns (denoised plane, accumulation plane)."""
        d = OrcDenoise()
        d.width, d.height, d.frame_index, d.have_history = self.w, self.h, frame_index, 1 if self.have else 0
        d.cam = camera_from(cam)
        d.prev = self.prev_cam if self.prev_cam is not None else camera_from(cam)
        d.max_accumulated_frames, d.disocclusion_threshold, d.antilag_sigma_scale, d.antilag_power, d.max_blur_radius = self.params
        keep = {k: np.ascontiguousarray(planes[k]) for k in ("illuminance", "normal", "depth", "motion", "voxel_id")}
        den = np.ascontiguousarray(planes["denoised"]).copy()
        for k, v in keep.items():
            setattr(d, k, v.ctypes.data)
        d.denoised = den.ctypes.data
        hin, hout = self.hist[self.parity], self.hist[self.parity ^ 1]
        for k in ("accum", "depth", "normal", "id"):
            setattr(d, "hist_in_" + k, hin[k].ctypes.data)
            setattr(d, "hist_out_" + k, hout[k].ctypes.data)
        lib().orc_denoise(C.byref(d))
        self.parity ^= 1
        self.have = True
        self.prev_cam = camera_from(cam)
        return den, hout["accum"].copy()
