"""ctypes binding of libdust_hip.so -- the C ABI declared in include/dust_hip.h.

The library is the product; there is no Python or CPU fallback behind it. If the shared object is
missing the import raises, and every device entry point fails with DUST_ERR_NO_DEVICE when no GPU is
visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# DUST_HIP_LIB selects another build of the SAME library (tools/kernel_sections.py uses its -DDUST_PROFILE build)
LIB_PATH = os.environ.get("DUST_HIP_LIB") or os.path.join(_HERE, "libdust_hip.so")

OK = 0
ERR_INVALID_ARGUMENT = -1
ERR_NO_DEVICE = -2
ERR_HIP = -3
ERR_OUT_OF_MEMORY = -4
ERR_PARSE = -5
ERR_UNSUPPORTED = -6
ERR_NOT_READY = -7

PASS_PRIMARY = 1 << 0
PASS_AMBIENT_OCCLUSION = 1 << 1
PASS_FINAL_GATHER = 1 << 2
PASS_SURFEL = 1 << 3
PASS_ACCUMULATE = 1 << 4
PASS_DENOISE = 1 << 5
PASS_COUNT_STATS = 1 << 16
PASS_GI_ORDERED = 1 << 17
PASS_GI_SHARDED = 1 << 18
GI_PATH_AUTO, GI_PATH_PACKETS, GI_PATH_STREAMS = 0, 1, 2
SIDE_STREAM_AUTO, SIDE_STREAM_OFF = 0, 1
IN_FLIGHT_SHARE, IN_FLIGHT_ALL = 0, 1
RESERVE_AUTO = 0xFFFFFFFF
CONTEXT_TIMING = 1
CONTEXT_TIMING_SPARSE = 2

PLANE_ILLUMINANCE, PLANE_DENOISED, PLANE_ALBEDO, PLANE_NORMAL, PLANE_DEPTH, PLANE_MOTION, PLANE_VOXEL_ID, PLANE_ACCUM, PLANE_OUTPUT = range(9)
PLANE_BYTES_PER_PIXEL = (8, 8, 4, 4, 4, 8, 4, 16, 8)


class Block(C.Structure):  # DustHipBlock, 24 bytes
    _fields_ = [("x", C.c_uint16), ("y", C.c_uint16), ("z", C.c_uint16), ("w", C.c_uint16),
                ("mask", C.c_uint64), ("material_ptr", C.c_uint32), ("avg_albedo", C.c_uint32)]


class VoxModelInfo(C.Structure):
    _fields_ = [("size", C.c_uint32 * 3), ("n_voxels", C.c_uint32), ("n_blocks", C.c_uint32),
                ("n_materials", C.c_uint64), ("used", C.c_uint32)]


class PngInfo(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("layers", C.c_uint32), ("channels", C.c_uint32),
                ("bytes_per_channel", C.c_uint32)]


class VoxInstance(C.Structure):
    _fields_ = [("model", C.c_uint32), ("obj_to_world", C.c_float * 12)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("stream", C.c_void_p),
                ("lds_root_bytes", C.c_uint32), ("flags", C.c_uint32)]


class Camera(C.Structure):
    _fields_ = [("view_col0", C.c_float * 3), ("view_col1", C.c_float * 3), ("view_col2", C.c_float * 3),
                ("position", C.c_float * 3), ("tan_half_fov", C.c_float), ("far_", C.c_float), ("near_", C.c_float)]


class TopLevelInfo(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("dim", C.c_uint32 * 3), ("lo", C.c_float * 3), ("cell", C.c_float * 3),
                ("n_cells", C.c_uint32), ("n_items", C.c_uint32), ("n_groups", C.c_uint32)]


class Sky(C.Structure):
    _fields_ = [("state", C.c_float * 56)]


class FrameParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("passes", C.c_uint32), ("frame_index", C.c_uint32),
                ("rand", C.c_uint32), ("row_begin", C.c_uint32), ("row_end", C.c_uint32), ("surfel_rank", C.c_uint32), ("surfel_world", C.c_uint32)]


class FrameMoves(C.Structure):   # DustHipFrameMoves: the instances moved before a frame of dust_hip_render_frames
    _fields_ = [("n", C.c_uint32), ("instance_ids", C.POINTER(C.c_uint32)), ("obj_to_world", C.POINTER(C.c_float)), ("prev_obj_to_world", C.POINTER(C.c_float))]


class ToneMapParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("transfer_function", C.c_uint32), ("color_space_conversion", C.c_float * 9),
                ("min_log_luminance", C.c_float), ("max_log_luminance", C.c_float), ("time_coefficient", C.c_float)]


class DenoiseParams(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("max_accumulated_frames", C.c_uint32), ("disocclusion_threshold", C.c_float),
                ("antilag_sigma_scale", C.c_float), ("antilag_power", C.c_float), ("max_blur_radius", C.c_float)]


class PipelineConfig(C.Structure):  # DustHipPipelineConfig
    _fields_ = [("struct_size", C.c_uint32), ("reserve_blocks", C.c_uint32), ("gi_path", C.c_uint32), ("side_stream", C.c_uint32),
                ("side_share", C.c_uint32), ("frames_in_flight", C.c_uint32), ("in_flight_slots", C.c_uint32)]


class GiExchange(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("pool_size", C.c_uint32), ("width", C.c_uint32), ("touched_rows", C.c_uint32),
                ("slot_owner", C.c_void_p), ("touched", C.c_void_p), ("merged", C.c_void_p)]


class PassStats(C.Structure):
    _fields_ = [("ms", C.c_float), ("rays", C.c_uint64), ("instances_tested", C.c_uint64),
                ("upper_descents", C.c_uint64), ("mid_descents", C.c_uint64), ("bricks_tested", C.c_uint64),
                ("hits", C.c_uint64)]


# every symbol include/dust_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_u32p = C.POINTER(C.c_uint32)
_u64p = C.POINTER(C.c_uint64)
_u8p = C.POINTER(C.c_uint8)
_f32p = C.POINTER(C.c_float)
SYMBOLS = {
    "dust_hip_last_error": (C.c_char_p, []),
    "dust_hip_device_count": (C.c_int, []),
    "dust_vdb_tree_create": (C.c_int, [_u32p, C.c_uint32, C.POINTER(_P)]),
    "dust_vdb_tree_destroy": (None, [_P]),
    "dust_vdb_tree_set": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32]),
    "dust_vdb_tree_get": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]),
    "dust_vdb_tree_iter": (C.c_int, [_P, _u32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dust_vdb_tree_iter_leaf": (C.c_int, [_P, _u32p, _u64p, _u32p, C.c_size_t, C.POINTER(C.c_size_t)]),
    "dust_vdb_tree_meta": (C.c_int, [_P, _u32p, _u32p]),
    "dust_vdb_lca_level": (C.c_uint32, [_u32p, _u32p, C.c_uint32, C.c_uint32]),
    "dust_vdb_accessor_create": (C.c_int, [_P, C.POINTER(_P)]),
    "dust_vdb_accessor_destroy": (None, [_P]),
    "dust_vdb_accessor_get": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]),
    "dust_vdb_pool_create": (C.c_int, [C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "dust_vdb_pool_destroy": (None, [_P]),
    "dust_vdb_pool_alloc": (C.c_uint32, [_P]),
    "dust_vdb_pool_free": (None, [_P, C.c_uint32]),
    "dust_vdb_pool_num_chunks": (C.c_size_t, [_P]),
    "dust_vdb_bitmask_set": (None, [_u64p, C.c_size_t, C.c_int32]),
    "dust_vdb_bitmask_iter_set_bits": (C.c_size_t, [_u64p, C.c_size_t, _u32p, C.c_size_t]),
    "dust_vox_load": (C.c_int, [_P, C.c_size_t, C.POINTER(_P)]),
    "dust_vox_load_frame": (C.c_int, [_P, C.c_size_t, C.c_uint32, C.POINTER(_P)]),
    "dust_vox_scene_destroy": (None, [_P]),
    "dust_vox_scene_counts": (C.c_int, [_P, _u32p, _u32p]),
    "dust_vox_scene_model_info": (C.c_int, [_P, C.c_uint32, C.POINTER(VoxModelInfo)]),
    "dust_vox_scene_model_data": (C.c_int, [_P, C.c_uint32, C.POINTER(C.POINTER(Block)), C.POINTER(_u8p)]),
    "dust_vox_scene_palette": (C.c_int, [_P, C.POINTER(_u8p)]),
    "dust_vox_scene_instances": (C.c_int, [_P, C.POINTER(VoxInstance), C.c_uint32]),
    "dust_vox_flatten_model": (C.c_int, [_P, C.c_size_t, _u32p, _P, C.POINTER(C.POINTER(Block)), _u32p,
                                         C.POINTER(_u8p), _u64p]),
    "dust_vox_free": (None, [_P]),
    "dust_png_load_array": (C.c_int, [_P, C.c_size_t, C.POINTER(PngInfo), C.POINTER(_u8p)]),
    "dust_sky_dataset_create": (C.c_int, [_P, C.c_size_t, _P, C.c_size_t, C.POINTER(_P)]),
    "dust_sky_dataset_destroy": (None, [_P]),
    "dust_sky_bake": (C.c_int, [_P, C.c_float, _f32p, _f32p, C.POINTER(Sky)]),
    "dust_hip_context_create": (C.c_int, [C.POINTER(Config), C.POINTER(_P)]),
    "dust_hip_context_destroy": (None, [_P]),
    "dust_hip_sync": (C.c_int, [_P]),
    "dust_hip_model_create": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64, _P, C.c_uint32, C.POINTER(_P)]),
    "dust_hip_model_destroy": (None, [_P]),
    "dust_hip_model_set_voxels": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "dust_hip_model_get_voxels": (C.c_int, [_P, _P, _P, C.c_uint32]),
    "dust_hip_model_info": (C.c_int, [_P, _u32p, _u64p]),
    "dust_hip_model_read": (C.c_int, [_P, _P, C.c_uint32, _P, C.c_uint64]),
    "dust_hip_scene_create": (C.c_int, [_P, C.POINTER(_P)]),
    "dust_hip_scene_destroy": (None, [_P]),
    "dust_hip_scene_add_instance": (C.c_int, [_P, _P, _f32p, _f32p, _u32p]),
    "dust_hip_scene_set_transform": (C.c_int, [_P, C.c_uint32, _f32p, _f32p]),
    "dust_hip_scene_commit": (C.c_int, [_P]),
    "dust_hip_top_level_build": (C.c_int, [C.POINTER(C.c_float), C.c_uint32, _P, _u32p, C.c_size_t, C.POINTER(C.c_uint16), C.c_size_t, _u32p, _u32p]),
    "dust_hip_pipeline_create": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.POINTER(_P)]),
    "dust_hip_pipeline_destroy": (None, [_P]),
    "dust_hip_pipeline_set_noise": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32]),
    "dust_hip_render_frame": (C.c_int, [_P, _P, C.POINTER(Camera), C.POINTER(Sky), C.POINTER(FrameParams)]),
    "dust_hip_render_frames": (C.c_int, [C.c_uint32, C.POINTER(C.c_void_p), _P, C.POINTER(Camera), C.POINTER(Sky), C.POINTER(FrameParams), C.POINTER(FrameMoves)]),
    "dust_hip_pipeline_pass_stats": (C.c_int, [_P, C.c_uint32, C.POINTER(PassStats)]),
    "dust_hip_pipeline_kernel_times": (C.c_int, [_P, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_uint32)]),
    "dust_hip_pipeline_tile_costs": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _u32p, _u32p]),
    "dust_hip_pipeline_plane_device_ptr": (C.c_int, [_P, C.c_int, C.POINTER(_P), C.POINTER(C.c_size_t)]),
    "dust_hip_pipeline_bind_plane": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "dust_hip_pipeline_read_plane": (C.c_int, [_P, C.c_int, _P, C.c_size_t]),
    "dust_hip_pipeline_configure_gi": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "dust_hip_pipeline_read_gi": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t]),
    "dust_hip_pipeline_write_gi": (C.c_int, [_P, C.c_uint32, _P, C.c_size_t]),
    "dust_hip_pipeline_gi_exchange": (C.c_int, [_P, C.c_uint32, C.POINTER(GiExchange)]),
    "dust_hip_gi_export": (C.c_int, [_P, C.c_uint32, C.c_uint32]),
    "dust_hip_gi_import": (C.c_int, [_P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "dust_hip_tone_map": (C.c_int, [_P, C.POINTER(ToneMapParams)]),
    "dust_hip_pipeline_exposure": (C.c_int, [_P, _f32p, _f32p]),
    "dust_hip_pipeline_clear": (C.c_int, [_P]),
    "dust_hip_pipeline_set_frames_in_flight": (C.c_int, [_P, C.c_uint32]),
    "dust_hip_pipeline_configure": (C.c_int, [_P, C.POINTER(PipelineConfig)]),
    "dust_hip_pipeline_get_config": (C.c_int, [_P, C.POINTER(PipelineConfig)]),
    "dust_hip_pipeline_set_denoiser": (C.c_int, [_P, C.POINTER(DenoiseParams)]),
    "dust_hip_pipeline_restart_denoiser": (C.c_int, [_P]),
    "dust_hip_device_eval": (C.c_int, [_P, C.c_uint32, _P, C.c_uint32, _P, C.c_uint32, C.c_uint32]),
    # multi-GPU: RCCL communicator (or a loopback group on one device), band gather, GI exchange
    "dust_hip_comm_unique_id": (C.c_int, [_P]),
    "dust_hip_comm_create": (C.c_int, [_P, C.c_uint32, C.c_uint32, _P, C.POINTER(_P)]),
    "dust_hip_comm_create_local": (C.c_int, [_P, C.c_uint32, C.POINTER(_P)]),
    "dust_hip_comm_destroy": (None, [_P]),
    "dust_hip_comm_info": (C.c_int, [_P, _u32p, _u32p, _u32p]),
    "dust_hip_gather_bands": (C.c_int, [_P, _P, C.c_int, _u32p, C.c_uint32, _P, C.c_size_t, C.POINTER(C.c_uint64)]),
    "dust_hip_gather_planes": (C.c_int, [_P, _P, C.c_uint32, _u32p, C.c_uint32, C.POINTER(C.c_uint64)]),
    "dust_hip_comm_wait": (C.c_int, [_P, C.c_uint64]),
    "dust_hip_comm_sync": (C.c_int, [_P]),
    "dust_hip_gi_exchange_run": (C.c_int, [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "dust_hip_gi_surfel_exchange_run": (C.c_int, [_P, _P, C.c_uint32]),
}

_lib = None


class DustError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"dust_hip status {status}: {message}")
        self.status = status


def load():
    """dlopen libdust_hip.so (built by dust_amd.build.build()); raises if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build the HIP extension first "
                          "(python -c 'import __graft_entry__ as g; g.build()'); there is no fallback path")
    lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != OK:
        raise DustError(status, load().dust_hip_last_error().decode("utf-8", "replace"))
