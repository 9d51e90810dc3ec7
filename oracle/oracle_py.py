"""ctypes binding of the CPU ORACLE (oracle/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg -- never by anything under softray_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

F_SHADING, F_SHADOWS, F_FOCAL_BLUR, F_POINT_LIGHT, F_SPECULAR, F_STATIC_SHADOWS = 1, 2, 4, 8, 16, 32
MODE_REF_TREE, MODE_BRUTE, MODE_NEAREST = 0, 1, 2


class Prim(C.Structure):
    _fields_ = [("kind", C.c_int32), ("argb", C.c_uint32), ("p", C.c_double * 9)]


class Frame(C.Structure):
    """Byte-identical to sr_frame (include/softray.h) and orc_frame (softray_oracle.h)."""
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


def build(force=False):
    """Compile liboracle.so with the committed Makefile (g++, -ffp-contract=off)."""
    src = os.path.join(_HERE, "softray_oracle.cpp")
    if (force or not os.path.exists(_LIB_PATH)
            or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
        L.orc_random_new.restype = vp; L.orc_random_new.argtypes = [i32]
        L.orc_random_free.argtypes = [vp]
        L.orc_random_next.restype = i32; L.orc_random_next.argtypes = [vp]
        L.orc_random_next_max.restype = i32; L.orc_random_next_max.argtypes = [vp, i32]
        L.orc_random_next_double.restype = dbl; L.orc_random_next_double.argtypes = [vp]
        L.orc_random_next_doubles.argtypes = [vp, i64, vp]
        L.orc_random_next_ints.argtypes = [vp, i64, vp]
        L.orc_scene_new.restype = vp
        L.orc_scene_free.argtypes = [vp]
        L.orc_scene_set_triangles.restype = i32
        L.orc_scene_set_triangles.argtypes = [vp, vp, vp, i64, vp, vp]
        L.orc_scene_set_extra.restype = i32; L.orc_scene_set_extra.argtypes = [vp, vp, i32]
        L.orc_scene_build_tree.restype = i32; L.orc_scene_build_tree.argtypes = [vp, i32, i32]
        L.orc_scene_tree_stats.argtypes = [vp, vp]
        L.orc_scene_reset_shadow_cache.argtypes = [vp]
        L.orc_render.restype = i32; L.orc_render.argtypes = [vp, vp, vp, vp, i32]
        L.orc_render_window.restype = i32; L.orc_render_window.argtypes = [vp, vp, vp, vp, i32, i32, i32]
        L.orc_shade_points.restype = i32
        L.orc_shade_points.argtypes = [vp, i64, vp, vp, vp, vp, i32]
        L.orc_trace.restype = i32
        L.orc_trace.argtypes = [vp, i32, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.orc_instance_matrices.argtypes = [vp, dbl, dbl, dbl, vp, vp]
        L.orc_default_fov_depth.restype = dbl
        L.orc_area_light_offsets.argtypes = [i32, i32, vp]
        L.orc_model_load_3ds.restype = vp; L.orc_model_load_3ds.argtypes = [vp, C.c_size_t, vp, C.c_size_t]
        L.orc_model_free.argtypes = [vp]
        L.orc_model_num_triangles.restype = i64; L.orc_model_num_triangles.argtypes = [vp]
        L.orc_model_num_vertices.restype = i64; L.orc_model_num_vertices.argtypes = [vp]
        L.orc_model_get.argtypes = [vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Random:
    """System.Random (.NET Framework 4.0)."""

    def __init__(self, seed):
        self._h = lib().orc_random_new(int(seed))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_random_free(self._h)
            self._h = None

    def Next(self, max_value=None):
        if max_value is None:
            return lib().orc_random_next(self._h)
        return lib().orc_random_next_max(self._h, int(max_value))

    def NextDouble(self):
        return lib().orc_random_next_double(self._h)

    def NextInts(self, n):
        """n x Next() (= InternalSample); NextDouble() == sample * (1.0 / 2147483647)."""
        out = np.zeros(int(n), dtype=np.int32)
        lib().orc_random_next_ints(self._h, int(n), _p(out))
        return out

    def NextDoubles(self, n):
        out = np.zeros(int(n))
        lib().orc_random_next_doubles(self._h, int(n), _p(out))
        return out


def instance_matrices(position, yaw, pitch, roll):
    pos = np.asarray(position, dtype=np.float64)
    t = np.zeros(12)
    it = np.zeros(12)
    lib().orc_instance_matrices(_p(pos), yaw, pitch, roll, _p(t), _p(it))
    return t, it


def default_fov_depth():
    return lib().orc_default_fov_depth()


def area_light_offsets(seed, count=100):
    out = np.zeros((count, 3))
    lib().orc_area_light_offsets(int(seed), int(count), _p(out))
    return out


def load_3ds(data):
    """Model.Load3ds + PostProcessGeometry -> (v9[n,3,3], argb[n], min[3], max[3])."""
    buf = np.frombuffer(data, dtype=np.uint8)
    err = C.create_string_buffer(256)
    h = lib().orc_model_load_3ds(_p(buf), len(data), err, 256)
    if not h:
        raise ValueError(err.value.decode())
    try:
        n = lib().orc_model_num_triangles(h)
        v9 = np.zeros((n, 3, 3))
        argb = np.zeros(n, dtype=np.uint32)
        bmin = np.zeros(3)
        bmax = np.zeros(3)
        lib().orc_model_get(h, _p(v9), _p(argb), _p(bmin), _p(bmax))
    finally:
        lib().orc_model_free(h)
    return v9, argb, bmin, bmax


class Scene:
    def __init__(self):
        self._h = lib().orc_scene_new()
        self.num_triangles = 0

    def __del__(self):
        if getattr(self, "_h", None):
            lib().orc_scene_free(self._h)
            self._h = None

    def set_triangles(self, v9, argb, bmin, bmax):
        v9 = np.ascontiguousarray(v9, dtype=np.float64).reshape(-1, 9)
        argb = np.ascontiguousarray(argb, dtype=np.uint32)
        bmin = np.ascontiguousarray(bmin, dtype=np.float64)
        bmax = np.ascontiguousarray(bmax, dtype=np.float64)
        self.num_triangles = v9.shape[0]
        rc = lib().orc_scene_set_triangles(self._h, _p(v9), _p(argb), v9.shape[0], _p(bmin), _p(bmax))
        assert rc == 0

    def set_extra(self, prims):
        """prims: list of (kind, argb, params...)"""
        arr = (Prim * max(1, len(prims)))()
        for i, (kind, argb, params) in enumerate(prims):
            arr[i].kind = kind
            arr[i].argb = argb
            for j, v in enumerate(params):
                arr[i].p[j] = v
        rc = lib().orc_scene_set_extra(self._h, arr, len(prims))
        assert rc == 0

    def build_tree(self, max_depth=15, max_per_leaf=25):
        return lib().orc_scene_build_tree(self._h, max_depth, max_per_leaf)

    def reset_shadow_cache(self):
        lib().orc_scene_reset_shadow_cache(self._h)

    def tree_stats(self):
        out = np.zeros(4, dtype=np.int32)
        lib().orc_scene_tree_stats(self._h, _p(out))
        return tuple(int(x) for x in out)

    def render(self, frame, threads=1, out=None, cols=None):
        """cols = (begin, end): only those columns of the frame's rows are drawn (orc_render_window), the rest of `out` is untouched."""
        f = frame
        if f.strip_count > 0:
            rows = [r for r in range(max(0, f.start_row), min(f.height - 1, f.end_row) + 1)
                    if (r // f.strip_rows) % f.strip_count == f.strip_index]
            n = len(rows) * f.width
        else:
            n = f.width * f.height
        pixels = out if out is not None else np.zeros(n, dtype=np.int32)
        stats = np.zeros(4, dtype=np.uint64)
        if cols is not None:
            rc = lib().orc_render_window(self._h, C.byref(f), _p(pixels), _p(stats), threads, int(cols[0]), int(cols[1]))
        else:
            rc = lib().orc_render(self._h, C.byref(f), _p(pixels), _p(stats), threads)
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        return pixels.view(np.uint32), stats

    def trace(self, target, starts, dirs, counters=False):
        starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = starts.shape[0]
        res = dict(hit=np.zeros(n, dtype=np.uint8), ray_frac=np.zeros(n), pos=np.zeros((n, 3)),
                   normal=np.zeros((n, 3)), color=np.zeros(n, dtype=np.uint32),
                   tri_index=np.zeros(n, dtype=np.int32))
        cnt = np.zeros((n, 3), dtype=np.int32) if counters else None
        rc = lib().orc_trace(self._h, target, n, _p(starts), _p(dirs), _p(res["hit"]), _p(res["ray_frac"]),
                             _p(res["pos"]), _p(res["normal"]), _p(res["color"]), _p(res["tri_index"]), _p(cnt))
        if rc != 0:
            raise RuntimeError("orc_trace failed: %d" % rc)
        if counters:
            res["counters"] = cnt
        return res


# ---- surface passes after the raytrace (numpy restatement; Engine3D/Renderer.cs:765-767) ----
STYLE_STANDARD, STYLE_COLOR_SHUFFLE, STYLE_NEGATIVE, STYLE_DEPTH_SMOOTH, STYLE_DEPTH_BANDED = 0, 1, 2, 3, 4


def shade_points(frame, pos, normal, color, threads=1):
    """ShadingMethod.IntersectRay's colour step for recorded intersections (orc_shade_points)."""
    pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
    normal = np.ascontiguousarray(normal, dtype=np.float64).reshape(-1, 3)
    color = np.ascontiguousarray(color, dtype=np.uint32)
    out = np.zeros(pos.shape[0], dtype=np.uint32)
    rc = lib().orc_shade_points(C.byref(frame), pos.shape[0], _p(pos), _p(normal), _p(color), _p(out), int(threads))
    assert rc == 0
    return out


def post_process(pixels, style, background_color=0):
    """PostProcessImage's Surface.ApplyColorFunc lambdas (Renderer.cs:819-865, Surface.cs:226-233) on a
    uint32 array; C# unchecked uint arithmetic = numpy uint32 wrap-around."""
    x = np.ascontiguousarray(pixels).view(np.uint32).copy()
    bg = np.uint32(background_color)
    if style == STYLE_STANDARD:
        return x
    if style == STYLE_COLOR_SHUFFLE:                                   # :827-830
        return ((x & np.uint32(0xffff)) << np.uint32(8)) + ((x >> np.uint32(16)) & np.uint32(0xff))
    if style == STYLE_NEGATIVE:                                        # :832-834
        with np.errstate(over="ignore"):
            neg = np.uint32(0x00ffffff) - x
        return np.where(x == bg, bg, neg).astype(np.uint32)
    if style == STYLE_DEPTH_SMOOTH:                                    # :844-848
        return (((x >> np.uint32(8)) & np.uint32(0xff0000)) + ((x >> np.uint32(16)) & np.uint32(0xff00))
                + ((x >> np.uint32(24)) & np.uint32(0xff)))
    if style == STYLE_DEPTH_BANDED:                                    # :859-863
        return ((x >> np.uint32(24)) & np.uint32(0xff)) * np.uint32(111)
    raise ValueError("style %r is not a per-pixel colour function" % (style,))


def anti_alias(src, dst_width, dst_height, resolution):
    """AntiAliasImage (Renderer.cs:937-978): integer average of resolution^2 source pixels per channel,
    PackRgb (alpha 255, Surface.cs:98-101)."""
    n = int(resolution)
    s = np.ascontiguousarray(src).view(np.uint32).reshape(dst_height, n, dst_width, n).astype(np.int64)
    out = np.full((dst_height, dst_width), 255 << 24, dtype=np.int64)
    for shift in (16, 8, 0):
        chan = ((s >> shift) & 0xff).sum(axis=(1, 3)) // (n * n)
        out += (chan & 0xff) << shift
    return out.astype(np.uint32).reshape(-1)
