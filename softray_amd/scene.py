"""Thin object wrapper over the C ABI: one `GpuScene` per reference `Renderer` (what PreCalculate()
keeps: geometry_simple, geometry_subdivided, ExtraGeometryToRaytrace)."""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import Frame, KernelTime, Prim


class SoftrayError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("softray error %d: %s" % (code, msg))
        self.code = code


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _check(rc):
    if rc != 0:
        raise SoftrayError(rc, _lib.lib().sr_last_error().decode())


class GpuScene:
    def __init__(self, device=0, devices=None):
        """device: one HIP ordinal (-1 = host-only scene); devices: a list of ordinals -> ONE scene over several GPUs of this
        process (sr_create_multi: frames are split into interleaved 16-row strips inside the library)."""
        h = C.c_void_p()
        if devices is not None:
            arr = (C.c_int32 * len(devices))(*[int(d) for d in devices])
            _check(_lib.lib().sr_create_multi(arr, len(devices), C.byref(h)))
            device = int(devices[0]) if len(devices) else -1
        else:
            _check(_lib.lib().sr_create(int(device), C.byref(h)))
        self._h = h
        self.device = device

    def device_count(self):
        return int(_lib.lib().sr_device_count(self._h))

    def close(self):
        if getattr(self, "_h", None):
            try:
                _lib.lib().sr_destroy(self._h)
            except TypeError:                     # interpreter shutdown: the module globals are already gone
                pass
            self._h = None

    __del__ = close

    # ---- PreCalculate() ----
    def set_triangles(self, v9, argb, bmin, bmax):
        v9 = np.ascontiguousarray(v9, dtype=np.float64).reshape(-1, 9)
        argb = np.ascontiguousarray(argb, dtype=np.uint32)
        bmin = np.ascontiguousarray(bmin, dtype=np.float64)
        bmax = np.ascontiguousarray(bmax, dtype=np.float64)
        _check(_lib.lib().sr_set_triangles(self._h, _p(v9), _p(argb), v9.shape[0], _p(bmin), _p(bmax)))

    def load_3ds(self, data):
        buf = np.frombuffer(bytes(data), dtype=np.uint8)
        _check(_lib.lib().sr_load_3ds(self._h, _p(buf), buf.size))

    def num_triangles(self):
        return int(_lib.lib().sr_num_triangles(self._h))

    def get_triangles(self):
        n = self.num_triangles()
        v9 = np.zeros((n, 3, 3)); argb = np.zeros(n, dtype=np.uint32); bmin = np.zeros(3); bmax = np.zeros(3)
        _check(_lib.lib().sr_get_triangles(self._h, _p(v9), _p(argb), _p(bmin), _p(bmax)))
        return v9, argb, bmin, bmax

    def set_extra(self, prims):
        arr = (Prim * max(1, len(prims)))()
        for i, (kind, argb, params) in enumerate(prims):
            arr[i].kind = kind
            arr[i].argb = argb
            for j, v in enumerate(params):
                arr[i].p[j] = v
        _check(_lib.lib().sr_set_extra_geometry(self._h, arr, len(prims)))

    def build(self, modes=(_lib.MODE_REF_TREE,), max_depth=0, max_per_leaf=0, on_device=None):
        """on_device: None = the library's default for the own BVH (device LBVH when the scene has a device), True = insist on the
        device build, False = the host's binned-SAH builder."""
        mask = 0 if on_device is None else (_lib.BUILD_ON_DEVICE if on_device else _lib.BUILD_ON_HOST)
        for m in modes:
            mask |= 1 << m
        _check(_lib.lib().sr_build(self._h, mask, max_depth, max_per_leaf))

    def tree_stats(self):
        out = np.zeros(4, dtype=np.int32)
        _check(_lib.lib().sr_tree_stats(self._h, _p(out)))
        return tuple(int(x) for x in out)

    def bvh_stats(self):
        """(depth, inner nodes, triangles, built on device) of the library's own BVH."""
        out = np.zeros(4, dtype=np.int64)
        _check(_lib.lib().sr_bvh_stats(self._h, _p(out)))
        return tuple(int(x) for x in out)

    def wide_tree_stats(self):
        """(depth, nodes, child slots in use, leaves, triangles in leaves) of the four-wide form of the host-built BVH."""
        out = np.zeros(5, dtype=np.int64)
        _check(_lib.lib().sr_wide_tree_stats(self._h, _p(out)))
        return tuple(int(x) for x in out)

    def bvh_digest(self):
        """(hash of the node array, hash of the leaf-ordered triangle indices) of the host-built BVH."""
        out = np.zeros(2, dtype=np.uint64)
        _check(_lib.lib().sr_bvh_digest(self._h, _p(out)))
        return tuple(int(x) for x in out)

    # ---- Render() ----
    @staticmethod
    def pixel_count(frame):
        return int(_lib.lib().sr_frame_pixel_count(C.byref(frame)))

    def render(self, frame, out=None, stats=True):
        n = self.pixel_count(frame)
        if out is not None:
            if not (isinstance(out, np.ndarray) and out.dtype.itemsize == 4 and out.dtype.kind in "iu" and out.flags["C_CONTIGUOUS"]
                    and out.flags["WRITEABLE"] and out.size >= n):
                raise ValueError("out must be a writable C-contiguous int32/uint32 array of at least %d pixels" % n)
        pixels = out if out is not None else np.zeros(n, dtype=np.int32)
        st = np.zeros(4, dtype=np.uint64) if stats else None
        _check(_lib.lib().sr_render(self._h, C.byref(frame), _p(pixels), _p(st)))
        return pixels.view(np.uint32), st

    def render_device(self, frame, d_pixels_ptr, stream=0, d_stats_ptr=None):
        _check(_lib.lib().sr_render_device(self._h, C.byref(frame), C.c_void_p(d_pixels_ptr), C.c_void_p(stream),
                                           C.c_void_p(d_stats_ptr) if d_stats_ptr else None))

    # ---- PostProcessImage / AntiAliasImage (Renderer.cs:765-767) ----
    def post_process(self, pixels, style, background_color=0):
        """In place on a host int32/uint32 array."""
        assert pixels.dtype.itemsize == 4 and pixels.flags["C_CONTIGUOUS"]
        _check(_lib.lib().sr_post_process(self._h, _p(pixels), pixels.size, int(style), int(background_color) & 0xFFFFFFFF))
        return pixels

    def post_process_device(self, d_pixels_ptr, count, style, background_color=0, stream=0):
        _check(_lib.lib().sr_post_process_device(self._h, C.c_void_p(d_pixels_ptr), int(count), int(style),
                                                 int(background_color) & 0xFFFFFFFF, C.c_void_p(stream)))

    def anti_alias(self, src, dst_width, dst_height, resolution, out=None):
        src = np.ascontiguousarray(src)
        assert src.dtype.itemsize == 4 and (resolution < 1 or src.size == dst_width * dst_height * resolution * resolution)
        dst = out if out is not None else np.zeros(dst_width * dst_height, dtype=np.int32)
        _check(_lib.lib().sr_anti_alias(self._h, _p(src), int(dst_width), int(dst_height), int(resolution), _p(dst)))
        return dst.view(np.uint32)

    def anti_alias_device(self, d_src_ptr, dst_width, dst_height, resolution, d_dst_ptr, stream=0):
        _check(_lib.lib().sr_anti_alias_device(self._h, C.c_void_p(d_src_ptr), int(dst_width), int(dst_height), int(resolution),
                                               C.c_void_p(d_dst_ptr), C.c_void_p(stream)))

    def reset_shadow_cache(self):
        """Forget the static shadow cache (SR_F_STATIC_SHADOWS): what a new Renderer starts with."""
        _check(_lib.lib().sr_reset_shadow_cache(self._h))

    def ray_stats(self):
        """primary {rays, tests, nodes, leaves} + secondary {rays, tests, nodes, leaves} of the last render(stats=True)."""
        out = np.zeros(24, dtype=np.uint64)                           # SR_STATS_COUNT
        _check(_lib.lib().sr_last_ray_stats(self._h, _p(out)))
        return out

    def debug_set(self, key, value):
        """Test / experiment hook of this scene (include/softray.h SR_DBG_*); value < 0 restores the default."""
        _check(_lib.lib().sr_debug_set(self._h, int(key), int(value)))

    def debug_counters(self):
        out = np.zeros(8, dtype=np.uint32)
        _check(_lib.lib().sr_debug_counters(self._h, _p(out)))
        return [int(x) for x in out]

    def reset_kernel_times(self):
        _lib.lib().sr_reset_kernel_times(self._h)

    def kernel_times(self):
        """{kernel: (total ms, launches)} since reset_kernel_times() -- HIP events on the launch stream."""
        arr = (KernelTime * 16)()
        n = _lib.lib().sr_kernel_times(self._h, arr, 16)
        return {arr[i].name.decode(): (float(arr[i].ms), int(arr[i].launches)) for i in range(n)}

    def shade_points(self, frame, pos, normal, color):
        """ShadingMethod.IntersectRay's colour step for recorded intersections (sr_shade_points)."""
        pos = np.ascontiguousarray(pos, dtype=np.float64).reshape(-1, 3)
        normal = np.ascontiguousarray(normal, dtype=np.float64).reshape(-1, 3)
        color = np.ascontiguousarray(color, dtype=np.uint32)
        out = np.zeros(pos.shape[0], dtype=np.uint32)
        _check(_lib.lib().sr_shade_points(self._h, C.byref(frame), pos.shape[0], _p(pos), _p(normal), _p(color), _p(out)))
        return out

    # ---- IRayIntersectable.IntersectRay, batched ----
    def trace(self, target, starts, dirs, counters=False):
        starts = np.ascontiguousarray(starts, dtype=np.float64).reshape(-1, 3)
        dirs = np.ascontiguousarray(dirs, dtype=np.float64).reshape(-1, 3)
        n = starts.shape[0]
        res = dict(hit=np.zeros(n, dtype=np.uint8), ray_frac=np.zeros(n), pos=np.zeros((n, 3)),
                   normal=np.zeros((n, 3)), color=np.zeros(n, dtype=np.uint32), tri_index=np.zeros(n, dtype=np.int32))
        cnt = np.zeros((n, 3), dtype=np.int32) if counters else None
        _check(_lib.lib().sr_trace_rays(self._h, int(target), n, _p(starts), _p(dirs), _p(res["hit"]), _p(res["ray_frac"]),
                                        _p(res["pos"]), _p(res["normal"]), _p(res["color"]), _p(res["tri_index"]), _p(cnt)))
        if counters:
            res["counters"] = cnt
        return res


    def trace_device(self, target, n, d_starts, d_dirs, d_hit=0, d_ray_frac=0, d_pos=0, d_normal=0, d_color=0, d_tri_index=0, d_counters=0, stream=0):
        """sr_trace_rays_device: every array is a DEVICE pointer (e.g. tensor.data_ptr()); enqueued on `stream`, no host sync."""
        vp = lambda x: C.c_void_p(x) if x else None
        _check(_lib.lib().sr_trace_rays_device(self._h, int(target), int(n), vp(d_starts), vp(d_dirs), vp(d_hit), vp(d_ray_frac), vp(d_pos),
                                               vp(d_normal), vp(d_color), vp(d_tri_index), vp(d_counters), vp(stream)))

    # ---- the strip gather over RCCL (include/softray.h) ----
    def set_gather(self, kind):
        """Multi-device scene: _lib.GATHER_COPY (peer copies, default) or _lib.GATHER_RCCL (grouped ncclSend / ncclRecv)."""
        _check(_lib.lib().sr_set_gather(self._h, int(kind)))

    def rccl_init(self, unique_id, world, rank):
        buf = (C.c_uint8 * _lib.RCCL_ID_BYTES).from_buffer_copy(bytes(unique_id))
        _check(_lib.lib().sr_rccl_init(self._h, buf, int(world), int(rank)))

    def rccl_render(self, frame, d_full_ptr, stream=0):
        """Render this rank's strips of `frame` and gather every rank's on rank 0 (d_full_ptr: device int32[W*H] there, 0 elsewhere)."""
        _check(_lib.lib().sr_rccl_render(self._h, C.byref(frame), C.c_void_p(d_full_ptr) if d_full_ptr else None, C.c_void_p(stream) if stream else None))

    def rccl_gather(self, frame, d_strips_ptr, d_full_ptr, stream=0):
        _check(_lib.lib().sr_rccl_gather(self._h, C.byref(frame), C.c_void_p(d_strips_ptr) if d_strips_ptr else None,
                                         C.c_void_p(d_full_ptr) if d_full_ptr else None, C.c_void_p(stream) if stream else None))


def rccl_unique_id():
    """ncclGetUniqueId through the library (rank 0); hand the 128 bytes to the other ranks."""
    buf = (C.c_uint8 * _lib.RCCL_ID_BYTES)()
    _check(_lib.lib().sr_rccl_unique_id(buf))
    return bytes(buf)


def net_random_doubles(seed, n, skip=0):
    """n NextDouble() of System.Random(seed) after `skip` samples."""
    out = np.zeros(int(n))
    _lib.lib().sr_net_random_doubles(int(seed), int(skip), int(n), _p(out))
    return out


def make_random_triangles(n, seed=12345, space=100.0, extent=10.0, origin=0.0, opaque=False):
    """SpatialSubdivisionTests.MakeRandomTriangles with the library's System.Random port."""
    v9 = np.zeros((int(n), 3, 3))
    argb = np.zeros(int(n), dtype=np.uint32)
    _lib.lib().sr_make_random_triangles(int(seed), int(n), float(space), float(extent), float(origin), int(bool(opaque)),
                                        _p(v9), _p(argb))
    return v9, argb


def unit_cube_scene(n, seed=12345):
    """SURVEY 8d synthetic scene (configs 3/4): v1 in [-0.5,0.45]^3, extents U[0,0.05]^3, box [-0.5,0.5]^3."""
    v9, argb = make_random_triangles(n, seed, space=0.95, extent=0.05, origin=-0.5, opaque=True)
    return v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3)


def instance_matrices(position, yaw, pitch, roll):
    pos = np.asarray(position, dtype=np.float64)
    t = np.zeros(12); it = np.zeros(12)
    _lib.lib().sr_instance_matrices(_p(pos), float(yaw), float(pitch), float(roll), _p(t), _p(it))
    return t, it


def default_fov_depth():
    return float(_lib.lib().sr_default_fov_depth())


def area_light_offsets(seed, count=100):
    out = np.zeros((count, 3))
    _lib.lib().sr_area_light_offsets(int(seed), int(count), _p(out))
    return out
