"""Shared test helpers: golden BMP reader, the RendererTests pose, seeded synthetic scenes.

The scene generators use the oracle's System.Random port so that C#, the oracle and the HIP path
can regenerate identical inputs.
"""
import math
import os
import struct
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import oracle_py as orc  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def read_bmp_rgb(path):
    """32-bpp bottom-up BMP -> uint32 array [h, w] of 0x00RRGGBB (alpha dropped: the reference
    compares Format32bppRgb, RendererTests.cs:521)."""
    data = open(path, "rb").read()
    assert data[:2] == b"BM"
    off = struct.unpack_from("<I", data, 10)[0]
    w, h = struct.unpack_from("<ii", data, 18)
    bpp = struct.unpack_from("<H", data, 28)[0]
    assert bpp == 32 and h > 0
    px = np.frombuffer(data, dtype="<u4", count=w * h, offset=off).reshape(h, w)
    return (px[::-1] & 0x00FFFFFF).astype(np.uint32)


def load_obj3ds(name="obj.3ds"):
    return orc.load_3ds(open(os.path.join(GOLDEN, name), "rb").read())


def renderer_default_light():
    """Renderer ctor defaults (Renderer.cs:207-217)."""
    d = np.array([-1.0, -1.0, 1.0])
    ln = math.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    inv = 1.0 / ln
    d = d * inv
    pos = np.array([0.0, 0.0, 1.5]) - d * 2
    return d, pos


def make_frame(res_w, res_h=None, shading=True, shadows=False, focal_blur=False, sub_pixel_res=1,
               yaw_deg=135.0, pitch_deg=-22.0, roll_deg=0.0, depth=1.0, focal_depth=None,
               background=0xff00ff, mode=orc.MODE_REF_TREE, point_light=True, specular=True,
               shadow_samples=0, start_row=None, end_row=None, strips=None, position=None, static_shadows=False,
               concurrency=0):
    """The frame RendererTests.RaytraceScenario sets up (RendererTests.cs:65-90,381-417)."""
    if res_h is None:
        res_h = res_w
    f = orc.Frame()
    f.width, f.height = res_w, res_h
    f.start_row = 0 if start_row is None else start_row
    f.end_row = res_h - 1 if end_row is None else end_row
    f.sub_pixel_res = sub_pixel_res
    f.background_argb = background & 0x00FFFFFF
    flags = 0
    if shading:
        flags |= orc.F_SHADING
    if shadows:
        flags |= orc.F_SHADOWS
    if focal_blur:
        flags |= orc.F_FOCAL_BLUR
    if point_light:
        flags |= orc.F_POINT_LIGHT
    if specular:
        flags |= orc.F_SPECULAR
    if static_shadows:
        flags |= orc.F_STATIC_SHADOWS                    # rayTraceShadowsStatic (with shadows=True)
    f.flags = flags
    f.concurrency = concurrency                         # rayTraceConcurrency, 0 => the default 4
    f.random_seed = 1234567890
    f.shadow_samples = shadow_samples
    f.trace_mode = mode
    if strips:
        f.strip_rows, f.strip_count, f.strip_index = strips
    pos = [0.0, 0.0, depth] if position is None else list(position)
    yaw = yaw_deg / 180.0 * math.pi
    pitch = pitch_deg / 180.0 * math.pi
    roll = roll_deg / 180.0 * math.pi
    t, it = orc.instance_matrices(pos, yaw, pitch, roll)
    for i in range(12):
        f.transform[i] = t[i]
        f.inv_transform[i] = it[i]
    f.position_z = pos[2]
    f.fov_depth = orc.default_fov_depth()
    f.focal_depth = (depth + 0.5) if focal_depth is None else focal_depth  # RendererTests.cs:395
    f.focal_blur_strength = 10.0
    f.ambient = 0.1
    f.shininess = 100.0
    ld, lp = renderer_default_light()
    for i in range(3):
        f.light_dir_view[i] = ld[i]
        f.light_pos_view[i] = lp[i]
    f.area_light_offsets = None
    return f


def random_triangles(n, seed=12345, space=100.0, extent=10.0, origin=0.0, mask_color=False):
    """SpatialSubdivisionTests.MakeRandomTriangles (SpatialSubdivisionTests.cs:397-411): per triangle
    9 NextDouble + 1 Next, in that order.  `origin` shifts v1 (SURVEY 8d uses [-0.5, 0.45]^3)."""
    rnd = orc.Random(seed)
    smp = rnd.NextInts(10 * n).reshape(n, 10)
    u = smp[:, :9].astype(np.float64) * (1.0 / 2147483647)      # NextDouble() = sample * (1.0 / MBIG)
    v1 = u[:, 0:3] * space + origin
    v2 = v1 + u[:, 3:6] * extent
    v3 = v1 + u[:, 6:9] * extent
    v9 = np.stack([v1, v2, v3], axis=1)
    argb = smp[:, 9].astype(np.uint32)                           # (uint)random.Next()
    if mask_color:
        argb = (argb & np.uint32(0xFFFFFF)) | np.uint32(0xFF000000)
    return v9, argb, rnd


def unit_cube_scene(n, seed=12345):
    """SURVEY 8d synthetic scene for C3/C4: v1 in [-0.5,0.45]^3, extents U[0,0.05]^3, box [-0.5,0.5]^3."""
    v9, argb, _ = random_triangles(n, seed, space=0.95, extent=0.05, origin=-0.5, mask_color=True)
    return v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3)


def c1_spheres(count=16, seed=12345):
    """SURVEY 8d config-1 extra geometry: centres U[-0.5,0.5]^3, radius U[0.05,0.15], Color constants cyclic."""
    rnd = orc.Random(seed)
    palette = [(1, 0, 0), (0, 1, 0), (0, 0, 1), (1, 1, 0), (1, 0.5, 0), (0.5, 0.25, 0), (1, 0, 1), (0, 1, 1),
               (1, 1, 1), (0.5, 0.5, 0.5)]
    prims = []
    for i in range(count):
        c = [rnd.NextDouble() - 0.5, rnd.NextDouble() - 0.5, rnd.NextDouble() - 0.5]
        r = 0.05 + rnd.NextDouble() * 0.1
        cr, cg, cb = palette[i % len(palette)]
        argb = (255 << 24) + (int(cr * 255.0) << 16) + (int(cg * 255.0) << 8) + int(cb * 255.0)
        prims.append((0, argb, c + [r]))
    return prims


def leaf_face_scene():
    """The adversarial scene for the one documented difference between SR_MODE_REF_TREE and SR_MODE_BVH (include/softray.h): two hits
    within 1e-10 of each other across a leaf face of the reference tree.  The root box is longest in z, so the tree (max 2 per leaf)
    splits at z = 0.  T (red) straddles the split plane and crosses the rays x = 0 at z = 5e-11 -- inside the z < 0 leaf's box by the
    reference's 1e-10 containment slack (AxisAlignedBox.cs:143-149, SpatialSubdivision.cs:652); T2 (green) lies entirely at z = 2e-11
    (only in the z >= 0 leaf) and is NEARER.  A ray along +z from z < 0 visits the z < 0 leaf first, accepts T there and returns it
    (SpatialSubdivision.cs:458-627); the global nearest hit is T2.  T3 only keeps the split from being rejected (:167-181)."""
    e1, e2 = 5e-11, 2e-11
    t = [(1, -1, e1 + 0.01), (-1, -1, e1 - 0.01), (0, 1.5, e1)]             # plane z - 0.01 x = e1, facing -z
    t2 = [(1, -1, e2), (-1, -1, e2), (0, 1.5, e2)]
    t3 = [(1.9, 1.5, -3.0), (1.5, 1.5, -3.0), (1.7, 1.9, -3.0)]
    v9 = np.array([t, t2, t3], dtype=np.float64)
    argb = np.array([0xFFFF0000, 0xFF00FF00, 0xFF0000FF], dtype=np.uint32)
    return v9, argb, np.array([-2.0, -2.0, -4.0]), np.array([2.0, 2.0, 4.0])
