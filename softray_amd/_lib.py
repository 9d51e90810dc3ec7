"""ctypes loader for libsoftray_hip.so (the C ABI of include/softray.h).

The library is the product: there is no Python or CPU implementation of the hot path behind it.
If the shared object is missing this module raises -- it never substitutes anything else.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsoftray_hip.so")
CSRC = os.path.join(_HERE, "csrc")

SR_OK = 0
SR_ERR_INVALID_ARG, SR_ERR_OUT_OF_RANGE, SR_ERR_NO_MODEL, SR_ERR_NOT_BUILT = -1, -2, -3, -4
SR_ERR_UNSUPPORTED, SR_ERR_NO_DEVICE, SR_ERR_HIP, SR_ERR_FORMAT = -5, -6, -7, -8
F_SHADING, F_SHADOWS, F_FOCAL_BLUR, F_POINT_LIGHT, F_SPECULAR, F_STATIC_SHADOWS = 1, 2, 4, 8, 16, 32
F_SINGLE_KERNEL = 1 << 8
F_PER_LANE_SHADOWS = 1 << 9
F_NO_SPLIT = 1 << 10
F_LITERAL_SECONDARY = 1 << 11
F_PRIMARY_STATS_ONLY = 1 << 12
MODE_REF_TREE, MODE_BRUTE, MODE_BVH = 0, 1, 2
TARGET_ROOT = 0x100
BUILD_ON_DEVICE = 0x100
BUILD_ON_HOST = 0x200
STYLE_STANDARD, STYLE_COLOR_SHUFFLE, STYLE_NEGATIVE, STYLE_DEPTH_SMOOTH, STYLE_DEPTH_BANDED = 0, 1, 2, 3, 4

# every symbol include/softray.h declares (tests/test_abi.py checks the header against this list)
SYMBOLS = [
    "sr_create", "sr_destroy", "sr_set_triangles", "sr_set_extra_geometry", "sr_build", "sr_tree_stats",
    "sr_render", "sr_render_device", "sr_frame_pixel_count", "sr_trace_rays", "sr_instance_matrices",
    "sr_default_fov_depth", "sr_area_light_offsets", "sr_load_3ds", "sr_num_triangles", "sr_get_triangles",
    "sr_reset_kernel_times", "sr_kernel_times", "sr_last_ray_stats", "sr_make_random_triangles", "sr_debug_counters", "sr_last_error", "sr_abi_version",
    "sr_post_process", "sr_post_process_device", "sr_anti_alias", "sr_anti_alias_device", "sr_reset_shadow_cache",
    "sr_debug_set", "sr_bvh_stats", "sr_bvh_digest", "sr_wide_tree_stats", "sr_create_multi", "sr_device_count", "sr_shade_points",
    "sr_trace_rays_device", "sr_rccl_unique_id", "sr_rccl_init", "sr_rccl_render", "sr_rccl_gather", "sr_set_gather",
    "sr_net_random_doubles",
]
GATHER_COPY, GATHER_RCCL = 0, 1
RCCL_ID_BYTES = 128
# sr_debug_set keys (include/softray.h)
(DBG_BAND_SAMPLES, DBG_ROUND_CAP0, DBG_ROUND_CAP1, DBG_SPLIT, DBG_FB_RAY_CAP, DBG_BVH_LEAF, DBG_KERNEL_SWITCH,
 DBG_KERNEL_TIMING, DBG_EXACT_SHADOW_TESTS, DBG_PER_LANE_SHAFT, DBG_PER_LANE_PRIMARY, DBG_ROUND2_NODES, DBG_BUILD_THREADS,
 DBG_BVH2_PACKETS, DBG_NO_PEER, DBG_LITERAL_SHADOWS) = range(16)


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("argb", C.c_uint32), ("p", C.c_double * 9)]


class Frame(C.Structure):
    """sr_frame (include/softray.h)."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("start_row", C.c_int32), ("end_row", C.c_int32),
        ("sub_pixel_res", C.c_int32),
        ("background_argb", C.c_uint32),
        ("flags", C.c_uint32),
        ("random_seed", C.c_int32),
        ("shadow_samples", C.c_int32),
        ("trace_mode", C.c_int32),
        ("strip_rows", C.c_int32), ("strip_count", C.c_int32), ("strip_index", C.c_int32),
        ("max_bounces", C.c_int32),
        ("concurrency", C.c_int32), ("reserved0", C.c_int32),
        ("transform", C.c_double * 12),
        ("inv_transform", C.c_double * 12),
        ("position_z", C.c_double),
        ("fov_depth", C.c_double),
        ("focal_depth", C.c_double), ("focal_blur_strength", C.c_double),
        ("ambient", C.c_double), ("shininess", C.c_double),
        ("light_dir_view", C.c_double * 3), ("light_pos_view", C.c_double * 3),
        ("reflectivity", C.c_double),
        ("area_light_offsets", C.c_void_p),
    ]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char_p), ("ms", C.c_float), ("launches", C.c_int32)]


def build(force=False):
    """hipcc --offload-arch=gfx950 build of the shared library (cross-compiles without a GPU)."""
    srcs = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(_HERE, "..", "include", "softray.h")]
    newest = max(os.path.getmtime(p) for p in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-C", CSRC, "-s", "-j%d" % max(1, min(8, os.cpu_count() or 1))])
    return LIB_PATH


_lib = None


def lib():
    """Load libsoftray_hip.so; raises if it is not there (no fallback of any kind)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("libsoftray_hip.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                           "(softray_amd has no CPU/Python implementation of the raytrace path)")
    try:
        # torch ships its own libamdhip64; loading it first makes this library bind to the same HIP runtime, so that
        # device pointers and streams can be shared (two runtimes in one process do not see each other's GPU state)
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, i32, i64, u32, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_uint32, C.c_double
    L.sr_create.restype = i32; L.sr_create.argtypes = [i32, C.POINTER(vp)]
    L.sr_destroy.restype = None; L.sr_destroy.argtypes = [vp]
    L.sr_set_triangles.restype = i32; L.sr_set_triangles.argtypes = [vp, vp, vp, i64, vp, vp]
    L.sr_set_extra_geometry.restype = i32; L.sr_set_extra_geometry.argtypes = [vp, vp, i32]
    L.sr_build.restype = i32; L.sr_build.argtypes = [vp, u32, i32, i32]
    L.sr_tree_stats.restype = i32; L.sr_tree_stats.argtypes = [vp, vp]
    L.sr_render.restype = i32; L.sr_render.argtypes = [vp, vp, vp, vp]
    L.sr_render_device.restype = i32; L.sr_render_device.argtypes = [vp, vp, vp, vp, vp]
    L.sr_frame_pixel_count.restype = i64; L.sr_frame_pixel_count.argtypes = [vp]
    L.sr_trace_rays.restype = i32; L.sr_trace_rays.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sr_instance_matrices.restype = None; L.sr_instance_matrices.argtypes = [vp, dbl, dbl, dbl, vp, vp]
    L.sr_default_fov_depth.restype = dbl; L.sr_default_fov_depth.argtypes = []
    L.sr_area_light_offsets.restype = None; L.sr_area_light_offsets.argtypes = [i32, i32, vp]
    L.sr_load_3ds.restype = i32; L.sr_load_3ds.argtypes = [vp, vp, C.c_size_t]
    L.sr_num_triangles.restype = i64; L.sr_num_triangles.argtypes = [vp]
    L.sr_get_triangles.restype = i32; L.sr_get_triangles.argtypes = [vp, vp, vp, vp, vp]
    L.sr_reset_kernel_times.restype = None; L.sr_reset_kernel_times.argtypes = [vp]
    L.sr_kernel_times.restype = i32; L.sr_kernel_times.argtypes = [vp, vp, i32]
    L.sr_last_ray_stats.restype = i32; L.sr_last_ray_stats.argtypes = [vp, vp]
    L.sr_make_random_triangles.restype = None
    L.sr_make_random_triangles.argtypes = [i32, i64, dbl, dbl, dbl, i32, vp, vp]
    L.sr_debug_counters.restype = i32; L.sr_debug_counters.argtypes = [vp, vp]
    L.sr_post_process.restype = i32; L.sr_post_process.argtypes = [vp, vp, i64, i32, u32]
    L.sr_post_process_device.restype = i32; L.sr_post_process_device.argtypes = [vp, vp, i64, i32, u32, vp]
    L.sr_anti_alias.restype = i32; L.sr_anti_alias.argtypes = [vp, vp, i32, i32, i32, vp]
    L.sr_anti_alias_device.restype = i32; L.sr_anti_alias_device.argtypes = [vp, vp, i32, i32, i32, vp, vp]
    L.sr_reset_shadow_cache.restype = i32; L.sr_reset_shadow_cache.argtypes = [vp]
    L.sr_debug_set.restype = i32; L.sr_debug_set.argtypes = [vp, i32, i64]
    L.sr_bvh_stats.restype = i32; L.sr_bvh_stats.argtypes = [vp, vp]
    L.sr_bvh_digest.restype = i32; L.sr_bvh_digest.argtypes = [vp, vp]
    L.sr_wide_tree_stats.restype = i32; L.sr_wide_tree_stats.argtypes = [vp, vp]
    L.sr_create_multi.restype = i32; L.sr_create_multi.argtypes = [vp, i32, C.POINTER(vp)]
    L.sr_device_count.restype = i32; L.sr_device_count.argtypes = [vp]
    L.sr_shade_points.restype = i32; L.sr_shade_points.argtypes = [vp, vp, i64, vp, vp, vp, vp]
    L.sr_trace_rays_device.restype = i32; L.sr_trace_rays_device.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.sr_rccl_unique_id.restype = i32; L.sr_rccl_unique_id.argtypes = [vp]
    L.sr_rccl_init.restype = i32; L.sr_rccl_init.argtypes = [vp, vp, i32, i32]
    L.sr_rccl_render.restype = i32; L.sr_rccl_render.argtypes = [vp, vp, vp, vp]
    L.sr_rccl_gather.restype = i32; L.sr_rccl_gather.argtypes = [vp, vp, vp, vp, vp]
    L.sr_set_gather.restype = i32; L.sr_set_gather.argtypes = [vp, i32]
    L.sr_net_random_doubles.restype = None; L.sr_net_random_doubles.argtypes = [i32, i64, i64, vp]
    L.sr_last_error.restype = C.c_char_p; L.sr_last_error.argtypes = []
    L.sr_abi_version.restype = i32; L.sr_abi_version.argtypes = []
    _lib = L
    return L
