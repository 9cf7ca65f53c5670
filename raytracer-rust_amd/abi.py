"""ctypes mirror of include/rtk.h (the C ABI of the device library).

Field order and types follow the header exactly; tests/test_abi.py checks the sizes
against the compiled library.
"""
import ctypes as C

RTK_OK = 0
RTK_ERR_INVALID = -1
RTK_ERR_SINGULAR_CAMERA = -2
RTK_ERR_UNSUPPORTED = -3
RTK_ERR_HIP = -4
RTK_ERR_NO_DEVICE = -5
RTK_ERR_NOMEM = -6
RTK_ERR_RCCL = -7

RTK_SPHERE = 0
RTK_PLANE = 1

RTK_TEX_SOLID = 0
RTK_TEX_CHECKER = 1
RTK_TEX_GRADIENT = 2
RTK_TEX_RING = 3
RTK_TEX_STRIPED = 4

RTK_REFERENCE_MAX_DEPTH = 5
RTK_MAX_DEPTH_LIMIT = 8
RTK_MAX_LIGHTS = 32767
RTK_TIMING_HISTORY = 64
RTK_MAX_SAMPLES_PER_AXIS = 16
RTK_TRACE_AUTO = 0
RTK_TRACE_DEPTH_FIRST = 1
RTK_TRACE_BY_LEVEL = 2
RTK_TILE_W = 32
RTK_TILE_H = 8
RTK_STATIC_MAX_SPHERES = 32768
RTK_STATIC_MAX_RECORD_PIXELS = 131072
RTK_STATIC_SCRATCH_BYTES = 3 << 20

Double3 = C.c_double * 3
Double4 = C.c_double * 4
Double16 = C.c_double * 16


class rtk_material(C.Structure):
    _fields_ = [
        ("texture_kind", C.c_uint32),
        ("pattern_transformed", C.c_uint32),
        ("color_a", Double3),
        ("color_b", Double3),
        ("ambient", C.c_double),
        ("diffuse", C.c_double),
        ("specular", C.c_double),
        ("shininess", C.c_double),
        ("transparency", C.c_double),
        ("reflectivity", C.c_double),
        ("refractivity", C.c_double),
        ("pattern_transform", Double16),
        ("inner_pattern_transforms", C.POINTER(C.c_double)),
    ]


class rtk_object(C.Structure):
    _fields_ = [
        ("kind", C.c_uint32),
        ("reserved_", C.c_uint32),
        ("transform", Double16),
        ("plane_normal", Double4),
        ("material", rtk_material),
    ]


class rtk_light(C.Structure):
    _fields_ = [("position", Double4), ("intensity", Double3)]


class rtk_scene(C.Structure):
    _fields_ = [
        ("objects", C.POINTER(rtk_object)),
        ("n_objects", C.c_uint32),
        ("lights", C.POINTER(rtk_light)),
        ("n_lights", C.c_uint32),
        ("ambient_color", Double3),
        ("max_depth", C.c_uint32),
        ("reserved_", C.c_uint32),
    ]


class rtk_camera(C.Structure):
    _fields_ = [
        ("width", C.c_uint32),
        ("height", C.c_uint32),
        ("half_width", C.c_double),
        ("half_height", C.c_double),
        ("field_of_view", C.c_double),
        ("pixel_size", C.c_double),
        ("transform", Double16),
    ]


class rtk_stats(C.Structure):
    _fields_ = [
        ("rays_primary", C.c_uint64),
        ("rays_shadow", C.c_uint64),
        ("rays_secondary", C.c_uint64),
        ("exact_tests", C.c_uint64),
        ("kernel_ms", C.c_double),
        ("rays_ms", C.c_double),
        ("gather_ms", C.c_double),
    ]

    @property
    def rays(self):
        return self.rays_primary + self.rays_shadow + self.rays_secondary


class rtk_tiling(C.Structure):
    _fields_ = [("rank", C.c_uint32), ("world", C.c_uint32)]


class rtk_banding(C.Structure):
    _fields_ = [
        ("rank", C.c_uint32),
        ("world", C.c_uint32),
        ("band_tile_rows", C.c_uint32),
        ("first_tile_row", C.c_uint32),
        ("n_tile_rows", C.c_uint32),
        ("compact", C.c_uint32),
    ]


RTK_MAX_PATTERN_LEVELS = 8
RTK_GROUP_INFO_MEMBERS = 16
RTK_ABI_VERSION = 3


class rtk_group_info(C.Structure):
    _fields_ = [
        ("n_members", C.c_uint32),
        ("exchange", C.c_uint32),
        ("root_share", C.c_uint32),
        ("rows_mode", C.c_uint32),
        ("messages", C.c_uint32),
        ("reserved_", C.c_uint32),
        ("bytes_into_root", C.c_uint64),
        ("gather_ms", C.c_double),
        ("share_tile_rows", C.c_uint32 * RTK_GROUP_INFO_MEMBERS),
        ("share_ms", C.c_double * RTK_GROUP_INFO_MEMBERS),
        ("share_rays", C.c_uint64 * RTK_GROUP_INFO_MEMBERS),
    ]


# Every symbol include/rtk.h declares: name -> (restype, argtypes).
SYMBOLS = {
    "rtk_camera_init": (C.c_int, [C.POINTER(rtk_camera), C.c_uint32, C.c_uint32, C.c_double]),
    "rtk_tiles_for_rank": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtk_compact_pixels": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtk_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "rtk_destroy": (None, [C.c_void_p]),
    "rtk_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(rtk_scene), C.POINTER(C.c_void_p)]),
    "rtk_scene_release": (None, [C.c_void_p, C.c_void_p]),
    "rtk_render_device": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(rtk_camera), C.POINTER(rtk_tiling), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rtk_band_tile_rows": (C.c_uint64, [C.c_uint32, C.POINTER(rtk_banding)]),
    "rtk_band_pixel_rows": (C.c_uint64, [C.c_uint32, C.POINTER(rtk_banding)]),
    "rtk_render_device_bands": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(rtk_camera), C.POINTER(rtk_banding), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rtk_render_device_static": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(rtk_camera), C.POINTER(rtk_tiling), C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rtk_render": (
        C.c_int,
        [C.c_void_p, C.POINTER(rtk_scene), C.POINTER(rtk_camera), C.c_void_p, C.c_void_p, C.POINTER(rtk_stats)],
    ),
    "rtk_camera_supersample": (C.c_int, [C.POINTER(rtk_camera), C.c_uint32, C.POINTER(rtk_camera)]),
    "rtk_render_device_sampled": (
        C.c_int,
        [C.c_void_p, C.c_void_p, C.POINTER(rtk_camera), C.c_uint32, C.POINTER(rtk_tiling), C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),
    "rtk_compact_pixels_sampled": (C.c_uint64, [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rtk_untile_sampled": (
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rtk_render_sampled": (
        C.c_int,
        [C.c_void_p, C.POINTER(rtk_scene), C.POINTER(rtk_camera), C.c_uint32, C.c_void_p, C.c_void_p,
         C.POINTER(rtk_stats)],
    ),
    "rtk_resolve": (
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p],
    ),
    "rtk_trace_rays": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p]),
    "rtk_untile": (
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p],
    ),
    "rtk_untile_strided": (
        C.c_int,
        [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p,
         C.c_void_p, C.c_void_p],
    ),
    "rtk_get_stats": (C.c_int, [C.c_void_p, C.POINTER(rtk_stats)]),
    "rtk_get_timing_history": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "rtk_group_create": (C.c_int, [C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "rtk_group_destroy": (None, [C.c_void_p]),
    "rtk_group_size": (C.c_int, [C.c_void_p]),
    "rtk_group_upload": (C.c_int, [C.c_void_p, C.POINTER(rtk_scene)]),
    "rtk_group_render_device": (C.c_int, [C.c_void_p, C.POINTER(rtk_camera), C.c_void_p, C.c_void_p, C.c_void_p]),
    "rtk_group_get_stats": (C.c_int, [C.c_void_p, C.POINTER(rtk_stats)]),
    "rtk_group_band_tile_rows": (C.c_uint32, [C.c_void_p, C.c_uint32]),
    "rtk_group_set_bands_per_member": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtk_group_exchange": (C.c_char_p, [C.c_void_p]),
    "rtk_group_set_root_share": (C.c_int, [C.c_void_p, C.c_uint32]),
    "rtk_group_set_rows": (C.c_int, [C.c_void_p, C.c_uint32, C.POINTER(C.c_uint32)]),
    "rtk_group_balance": (C.c_int, [C.c_void_p, C.POINTER(rtk_camera), C.c_int, C.c_int, C.c_uint32, C.c_double]),
    "rtk_group_get_info": (C.c_int, [C.c_void_p, C.POINTER(rtk_group_info)]),
    "rtk_abi_version": (C.c_uint32, []),
    "rtk_flatten_digest": (C.c_int, [C.POINTER(rtk_scene), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "rtk_set_trace_order": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint64]),
    "rtk_last_error": (C.c_char_p, []),
    "rtk_version": (C.c_char_p, []),
}


def bind(lib):
    """Attaches restype/argtypes for every declared symbol; raises if one is missing."""
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError names the missing export
        fn.restype = restype
        fn.argtypes = argtypes
    return lib
