"""Config C5 at its stated size (BASELINE.json: 10 M triangles, 4096^2, 4-bounce reflection) and the large-scene regime of
the shadow pipeline (8 % of the hit points reach the packed fallback at 10 M triangles, node indices need 23 bits, leaf
offsets 24), plus the >= 1e7-point census of Math.Pow (ocml on the device vs glibc in the oracle) that SURVEY 7 asks for.
Everything calls through the C ABI (softray_amd.GpuScene); the oracle is only the checker."""
import os

import numpy as np
import pytest

import softray_amd as sa
from helpers import make_frame, orc, unit_cube_scene

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8


def as_sr(frame, mode, **flags):
    f = sa.Frame.from_buffer_copy(bytes(frame))
    f.trace_mode = mode
    if frame.area_light_offsets:
        f.area_light_offsets = frame.area_light_offsets
    if flags.get("single_kernel"):
        f.flags |= sa._lib.F_SINGLE_KERNEL
    if flags.get("per_lane"):
        f.flags |= sa._lib.F_PER_LANE_SHADOWS
    return f


@pytest.fixture(scope="module")
def c5_scene():
    """N = 10 000 000 triangles, extent 0.02, System.Random seed 12345, unit cube."""
    v9, argb = sa.make_random_triangles(10_000_000, 12345, space=0.98, extent=0.02, origin=-0.5, opaque=True)
    bmin, bmax = np.array([-0.5] * 3), np.array([0.5] * 3)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,), on_device=False)                  # host SAH build (the device LBVH is the default; see test_c5_device_built_bvh_same_pixels)
    depth, nodes, ntri, on_dev = g.bvh_stats()
    assert ntri == 10_000_000 and nodes > (1 << 20) and not on_dev          # node indices beyond 20 bits, record offsets beyond 23
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    assert o.build_tree() == 0                                 # the reference's tree, depth 15 / 25 per leaf
    return g, o, (v9, argb, bmin, bmax)


def test_c5_four_bounces_full_size(c5_scene):
    """(i) 4096^2, 4 mirror bounces, no shadows: the wavefront bounce pipeline (packet primary walk, k_bounce per level, k_fold)
    == the one-kernel renderer over the WHOLE frame, and four row pairs == the oracle."""
    g, o, _ = c5_scene
    f = make_frame(4096, depth=1.5)
    f.max_bounces, f.reflectivity = 4, 0.5
    a, st = g.render(as_sr(f, sa.MODE_BVH))
    assert st[0] == 4096 * 4096
    single, _ = g.render(as_sr(f, sa.MODE_BVH, single_kernel=True))
    assert np.array_equal(a, single)
    plain, _ = g.render(as_sr(make_frame(4096, depth=1.5), sa.MODE_BVH))
    assert np.count_nonzero(a != plain) > 1_000_000            # the bounces really changed the image
    a2 = a.reshape(4096, 4096)
    for r0 in (700, 2047, 2900, 3500):
        fo = make_frame(4096, depth=1.5, start_row=r0, end_row=r0 + 1)
        fo.max_bounces, fo.reflectivity = 4, 0.5
        want, _ = o.render(fo, threads=NCPU)
        assert np.array_equal(want.reshape(4096, 4096)[r0:r0 + 2], a2[r0:r0 + 2]), r0


def test_c5_scene_soft_shadows_full_size(c5_scene):
    """(ii) the same scene with the reference's 100-sample shadows at 4096^2: the shaft schedule (packet walk, fp32-classified
    tests, second round, packed fallback -- which this scene really needs) == one lane per hit point (k_shadow) over the WHOLE
    frame, and row pairs == the oracle."""
    g, o, _ = c5_scene
    f = make_frame(4096, depth=1.5, shadows=True)
    a, _ = g.render(as_sr(f, sa.MODE_BVH))
    c = g.debug_counters()
    assert c[2] > 1000 and c[3] > 1000, c                      # round 2 and the exact fallback really ran at this size
    lanes, _ = g.render(as_sr(f, sa.MODE_BVH, per_lane=True))
    assert np.array_equal(a, lanes)
    a2 = a.reshape(4096, 4096)
    for r0 in (1100, 2048):
        fo = make_frame(4096, depth=1.5, shadows=True, start_row=r0, end_row=r0)
        want, _ = o.render(fo, threads=NCPU)
        assert np.array_equal(want.reshape(4096, 4096)[r0], a2[r0]), r0


def test_c5_device_built_bvh_same_pixels(c5_scene):
    """The LBVH built on the device for the 10 M-triangle scene gives the pixels of the host SAH tree (1024^2, shadows on)."""
    g, _, (v9, argb, bmin, bmax) = c5_scene
    f = as_sr(make_frame(1024, depth=1.5, shadows=True), sa.MODE_BVH)
    a, _ = g.render(f)
    g2 = sa.GpuScene(0)
    g2.set_triangles(v9, argb, bmin, bmax)
    g2.build((sa.MODE_BVH,), on_device=True)
    assert g2.bvh_stats()[3] == 1
    b, _ = g2.render(f)
    assert np.array_equal(a, b)


def test_pow_census_ten_million_surface_points():
    """ShadingMethod's Math.Pow(cos, 100) through ocml (device) vs glibc (oracle) on >= 1e7 surface points: real hit points of
    a 2048^2 frame of the 1 M-triangle scene plus 2^23 synthetic (position, normal) pairs, point and directional light.  The
    byte (byte)(255 * intensity) flips only if the two pow results straddle a truncation boundary (about 5e-14 per point)."""
    v9, argb, bmin, bmax = unit_cube_scene(1_000_000)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,))
    f = make_frame(2048, depth=1.5)
    # camera rays of the frame (Renderer.cs:1722-1743), traced through the library: recorded intersections
    t, it = sa.instance_matrices([0.0, 0.0, 1.5], 135.0 / 180.0 * np.pi, -22.0 / 180.0 * np.pi, 0.0)
    it = np.array(it).reshape(3, 4)
    cols, rows = np.meshgrid(np.arange(2048, dtype=np.float64), np.arange(2048, dtype=np.float64))
    dv = np.stack([-(cols / 2048 - 0.5), -(rows / 2048 - 0.5) * 1.0, np.full_like(cols, sa.default_fov_depth())], axis=-1).reshape(-1, 3)
    dirs = dv @ it[:, :3].T
    start = it[:, :3] @ np.array([0.0, 0.0, -1.5])
    res = g.trace(sa.MODE_BVH, np.broadcast_to(start, dirs.shape), dirs)
    hit = res["hit"].astype(bool)
    pos_h, nrm_h, col_h = res["pos"][hit], res["normal"][hit], res["color"][hit]
    assert pos_h.shape[0] > 2_000_000
    rng = np.random.default_rng(2024)
    n_syn = 1 << 23
    pos_s = rng.uniform(-0.5, 0.5, size=(n_syn, 3))
    nrm_s = rng.normal(size=(n_syn, 3))
    nrm_s /= np.linalg.norm(nrm_s, axis=1, keepdims=True)
    col_s = (0xFF000000 | rng.integers(0, 1 << 24, size=n_syn, dtype=np.uint64)).astype(np.uint32)
    pos = np.concatenate([pos_h, pos_s]); nrm = np.concatenate([nrm_h, nrm_s]); col = np.concatenate([col_h, col_s])
    assert pos.shape[0] >= 10_000_000
    total, flips = 0, 0
    for point_light in (True, False):
        fr = make_frame(64, depth=1.5, point_light=point_light)
        want = orc.shade_points(fr, pos, nrm, col, threads=NCPU)
        got = g.shade_points(as_sr(fr, sa.MODE_BVH), pos, nrm, col)
        total += pos.shape[0]
        flips += int(np.count_nonzero(want != got))
    print("pow census: %d shaded points, %d byte flips" % (total, flips))
    assert total >= 20_000_000 and flips == 0
