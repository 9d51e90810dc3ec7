"""softray_amd -- MI355X (gfx950) implementation of Engine3D's per-pixel raytrace hot path
(voidstar69/softray), behind the reference's own Renderer / Instance / Surface API.

Layers:  csrc/ (HIP kernels + C ABI, include/softray.h)  ->  scene.py (ctypes handle)  ->
renderer.py (host-side mirror of Engine3D.Renderer / Instance / GeometryCollection ...).
"""
from . import _lib
from ._lib import (F_FOCAL_BLUR, F_POINT_LIGHT, F_SHADING, F_SHADOWS, F_SPECULAR, F_STATIC_SHADOWS, MODE_BRUTE, MODE_BVH,
                   MODE_REF_TREE, TARGET_ROOT, Frame)
from .scene import (GpuScene, SoftrayError, area_light_offsets, default_fov_depth, instance_matrices,
                    make_random_triangles, net_random_doubles, rccl_unique_id, unit_cube_scene)

__all__ = ["GpuScene", "SoftrayError", "Frame", "make_random_triangles", "net_random_doubles", "rccl_unique_id", "unit_cube_scene", "instance_matrices", "default_fov_depth", "area_light_offsets",
           "MODE_REF_TREE", "MODE_BRUTE", "MODE_BVH", "TARGET_ROOT", "F_SHADING", "F_SHADOWS", "F_STATIC_SHADOWS", "F_FOCAL_BLUR",
           "F_POINT_LIGHT", "F_SPECULAR", "_lib"]
