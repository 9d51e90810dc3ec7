"""Pins the CPU oracle against everything the reference's own tests hold for the raytrace path
(SURVEY.md 8c): System.Random fixtures, the five seeded tree KATs, the seeded tree==brute-force
differential tests, primitive KATs / hit-rate windows and 20 golden BMPs.  CPU only."""
import os

import numpy as np
import pytest

from helpers import (GOLDEN, load_obj3ds, make_frame, orc, random_triangles, read_bmp_rgb)

TREE_BOX = ([0.0, 0.0, 0.0], [110.0, 110.0, 110.0])  # GetBoundingBoxOfRandomTriangles, SpatialSubdivisionTests.cs:413-418


def test_system_random_fixtures():
    # SURVEY.md Appendix A fixtures
    r = orc.Random(12345)
    assert [r.Next() for _ in range(3)] == [143337951, 150666398, 1663795458]
    r = orc.Random(12345)
    assert [r.NextDouble() for _ in range(3)] == [0.06674693481379511, 0.07015950887937075, 0.7747651351498278]
    r = orc.Random(1234567890)
    assert [r.NextDouble() for _ in range(3)] == [0.547308153727701, 0.42220238895258044, 0.3072717289008534]
    r = orc.Random(12345)
    ints = r.NextInts(3)
    assert list(ints) == [143337951, 150666398, 1663795458]


@pytest.mark.parametrize("n,max_depth,max_geom,expected", [
    (10, 5, 3, (4, 9, 5, 4)),          # ConstructArbitraryTree   SpatialSubdivisionTests.cs:59-73
    (5, 3, 1, (2, 3, 2, 1)),           # ConstructMaxDepthTree    :75-89
    (8, 3, 1, (3, 5, 3, 2)),           # ConstructBalancedTree    :91-105
    (4, 100, 1, (2, 3, 2, 1)),         # ConstructUnbalancedTree  :107-121
    (1000, 10, 5, (10, 885, 443, 442)),  # ConstructBigTree       :123-137
])
def test_tree_construction_kats(n, max_depth, max_geom, expected):
    v9, argb, _ = random_triangles(n, seed=12345)
    s = orc.Scene()
    s.set_triangles(v9, argb, *TREE_BOX)
    assert s.build_tree(max_depth, max_geom) == 0
    assert s.tree_stats() == expected


def test_tree_ctor_edge_cases():
    # EmptyInputToConstructor_NoError / DegenerateTrianglesToConstructor_NoError (:38-57)
    s = orc.Scene()
    s.set_triangles(np.zeros((0, 3, 3)), np.zeros(0, dtype=np.uint32), [0, 0, 0], [0, 0, 0])
    assert s.build_tree() == 0
    assert s.tree_stats() == (1, 1, 1, 0)
    s = orc.Scene()
    s.set_triangles(np.zeros((1, 3, 3)), np.zeros(1, dtype=np.uint32), [0, 0, 0], [1, 1, 1])
    assert s.build_tree() == 0
    # vertex outside the box -> ArgumentOutOfRangeException (SpatialSubdivision.cs:287-295)
    s = orc.Scene()
    v = np.zeros((1, 3, 3))
    v[0, 1] = [2.0, 0.0, 0.0]
    s.set_triangles(v, np.zeros(1, dtype=np.uint32), [0, 0, 0], [1, 1, 1])
    assert s.build_tree() == -2


def _inside_out_rays(rnd, n):
    u = rnd.NextDoubles(6 * n).reshape(n, 6)
    starts = u[:, 0:3] * 100.0                  # MakeRandomVector(triangleSpaceSize)
    dirs = (1.0 - -1.0) * u[:, 3:6] + -1.0      # MakeRandomVector(-1, 1, -1, 1, -1, 1)
    return starts, dirs


def _outside_in_rays(rnd, n):
    u = rnd.NextDoubles(6 * n).reshape(n, 6)
    starts = u[:, 0:3] * 1000.0                 # MakeRandomVector(triangleSpaceSize * 10)
    ends = u[:, 3:6] * 100.0
    return starts, ends - starts


def _vec_eq(a, b):
    # Vector.operator== (Vector.cs:43-47)
    d = a - b
    return (d * d).sum(axis=1) < 1e-10


def _differential(n_tris, max_depth, max_geom, seed, n_rays, outside_in=False):
    v9, argb, rnd = random_triangles(n_tris, seed=seed)
    s = orc.Scene()
    s.set_triangles(v9, argb, *TREE_BOX)
    assert s.build_tree(max_depth, max_geom) == 0
    starts, dirs = (_outside_in_rays if outside_in else _inside_out_rays)(rnd, n_rays)
    tree = s.trace(1, starts, dirs)
    base = s.trace(0, starts, dirs)
    assert np.array_equal(tree["hit"], base["hit"])
    h = base["hit"].astype(bool)
    assert np.array_equal(tree["tri_index"][h], base["tri_index"][h])
    if outside_in:
        assert np.all(np.abs(tree["ray_frac"][h] - base["ray_frac"][h]) <= 1e-10)
    else:
        assert np.array_equal(tree["ray_frac"][h], base["ray_frac"][h])
    assert np.all(_vec_eq(tree["pos"][h], base["pos"][h]))
    assert np.all(_vec_eq(tree["normal"][h], base["normal"][h]))
    assert np.array_equal(tree["color"][h], base["color"][h])
    # the build's own BVH semantics (global nearest inside the root box) must agree too
    near = s.trace(3, starts, dirs)
    assert np.array_equal(near["hit"], tree["hit"])
    assert np.array_equal(near["tri_index"][h], tree["tri_index"][h])
    assert np.array_equal(near["ray_frac"][h], tree["ray_frac"][h])
    return int(h.sum())


def test_tree_correctness_1():           # SpatialSubdivisionTests.cs:284-290
    _differential(100, 10, 5, 12345, 100000)


def test_tree_correctness_2():           # :292-298
    _differential(20, 10, 1, 123456, 100000)


def test_tree_correctness_outside_in():  # :300-306
    _differential(20, 10, 1, 123456, 100000, outside_in=True)


def test_tree_correctness_leaf_box_regressions():   # :325-339
    _differential(100, 10, 5, 1234567, 265896)
    _differential(10000, 10, 5, 1234567, 34)


def test_tree_hit_rate_window():
    # RayIntersectTreeFromInside_Performance: 20-30 % of 10 000 rays hit (:218-233)
    hits = _differential(1000, 10, 5, 12345, 10000)
    assert 2000 < hits < 3000


def _single_tri():
    s = orc.Scene()
    v = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float64)   # origin, right, up
    s.set_triangles(v, np.array([0xffffffff], dtype=np.uint32), [-1, -1, -1], [2, 2, 2])
    return s


def test_triangle_kats():
    s = _single_tri()
    # RayHitsTriangle (TriangleTests.cs:45-54)
    r = s.trace(0, [[0, 0, 1]], [[0, 0, -1]])
    assert r["hit"][0] == 1 and r["ray_frac"][0] == 1.0
    assert np.all(_vec_eq(r["pos"], np.array([[0.0, 0, 0]]))) and np.all(_vec_eq(r["normal"], np.array([[0.0, 0, 1]])))
    assert r["color"][0] == 0xffffffff
    # RayFromTriangleVertex_HitsTriangle (:56-65)
    r = s.trace(0, [[1, 0, 0]], [[0, 0, -1]])
    assert r["hit"][0] == 1 and r["ray_frac"][0] == 0.0
    # TriangleIsOneSided_RayFromOtherSideMisses (:67-73)
    r = s.trace(0, [[0, 0, -1]], [[0, 0, 1]])
    assert r["hit"][0] == 0


def test_triangle_hit_rate_window():
    # RayIntersectTrianglePerformance: 48-51 % hits for starts in the unit cube, dir = forward (:166-183)
    s = _single_tri()
    rnd = orc.Random(2024)
    starts = rnd.NextDoubles(3 * 200000).reshape(-1, 3)
    dirs = np.tile(np.array([0.0, 0.0, -1.0]), (starts.shape[0], 1))
    r = s.trace(0, starts, dirs)
    assert 0.48 < r["hit"].mean() < 0.51


def test_sphere_hit_rates():
    # RayIntersectSphereFromInside / MostlyFromOutside: every ray hits (TriangleTests.cs:205-224, 245-263)
    s = orc.Scene()
    s.set_triangles(np.zeros((0, 3, 3)), np.zeros(0, dtype=np.uint32), [0, 0, 0], [1, 1, 1])
    s.set_extra([(0, 0xffffffff, [0.5, 0.5, 0.5, 1.0])])
    rnd = orc.Random(7)
    u = rnd.NextDoubles(6 * 100000).reshape(-1, 6)
    r = s.trace(2, u[:, :3], 2.0 * u[:, 3:] - 1.0)
    assert r["hit"].all()
    starts = 20.0 * u[:, :3] - 10.0
    r = s.trace(2, starts, u[:, 3:] - starts)
    assert r["hit"].all()
    # rayFrac is a DISTANCE for spheres (Sphere.cs:164,188-196), not a multiple of dir
    r = s.trace(2, [[0.5, 0.5, 10.0]], [[0.0, 0.0, -20.0]])
    assert r["hit"][0] == 1 and abs(r["ray_frac"][0] - 8.5) < 1e-12


def test_aabb_as_ray_intersectable():
    """AxisAlignedBox.IntersectRay (AxisAlignedBox.cs:60-95): six one-sided planes + ContainsPoint.  RayIntersectAABBPerformance's rays
    (TriangleTests.cs:322-344: starts above the unit box, directions downwards) hit 99.8-100 %; a ray from inside hits nothing (every
    plane faces outwards); the hit carries the plane's unit normal and Color.White."""
    s = orc.Scene()
    s.set_triangles(np.zeros((0, 3, 3)), np.zeros(0, dtype=np.uint32), [-1, -1, -1], [1, 1, 1])
    s.set_extra([(4, 0xff123456, [-0.5, -0.5, -0.5, 0.5, 0.5, 0.5])])
    rnd = orc.Random(12345)
    u = rnd.NextDoubles(6 * 200000).reshape(-1, 6)
    starts = np.stack([-0.3 + 0.6 * u[:, 0], -0.3 + 0.6 * u[:, 1], 0.5 + 0.5 * u[:, 2]], axis=1)
    dirs = np.stack([-0.5 + u[:, 3], -0.5 + u[:, 4], np.full(u.shape[0], -1.0)], axis=1)
    r = s.trace(2, starts, dirs, counters=True)
    assert 0.998 < r["hit"].mean() <= 1.0
    h = r["hit"].astype(bool)
    assert np.all(r["normal"][h] == np.array([0.0, 0.0, 1.0])) and np.all(r["pos"][h][:, 2] == 0.5) and np.all(r["color"][h] == 0xffffffff)
    assert np.all(r["counters"][:, 0] == 6)                               # NumRayTests: six planes (:70)
    r = s.trace(2, [[0.0, 0.0, 0.0], [2.0, 0.1, 0.2], [2.0, 0.1, 0.2]], [[1.0, 0.2, 0.3], [-1.0, 0.0, 0.0], [1.0, 0.0, 0.0]])
    assert list(r["hit"]) == [0, 1, 0]
    assert r["ray_frac"][1] == 1.5 and np.all(r["normal"][1] == [1.0, 0.0, 0.0])


def test_3ds_loader_facts():
    v9, argb, bmin, bmax = load_obj3ds()
    assert v9.shape == (152, 3, 3)                       # obj.3DS = 152 tris
    assert np.all(argb == 0xff969696)                    # diffuse 150/255 -> 0x96
    assert bmin[0] == -0.5 and bmax[0] == 0.5            # longest axis scaled to 1 (Model.cs:766-790)
    v9b, argb_b, _, _ = load_obj3ds("obj2.3DS")
    assert v9b.shape == (107, 3, 3)
    with pytest.raises(ValueError):
        orc.load_3ds(b"\x00" * 32)                       # "Not a proper 3DS file." ThreeDSFile.cs:166-169


@pytest.fixture(scope="module")
def obj_scene():
    v9, argb, bmin, bmax = load_obj3ds()
    s = orc.Scene()
    s.set_triangles(v9, argb, bmin, bmax)
    assert s.build_tree() == 0                           # defaults 15 / 25 (SpatialSubdivision.cs:269-270)
    return s


GOLDENS = [
    # name, res, kwargs  (RendererTests.RaytraceScenario naming, RendererTests.cs:419-430)
    ("noShading", 100, dict(shading=False)),
    ("shading", 100, dict()),
    ("shading_2xAA", 100, dict(sub_pixel_res=2)),
    ("shading_4xAA", 100, dict(sub_pixel_res=4)),
    ("shading_8xAA", 100, dict(sub_pixel_res=8)),
    ("noShading_4xAA", 100, dict(shading=False, sub_pixel_res=4)),
    ("shading_focalBlurx2", 100, dict(focal_blur=True, sub_pixel_res=2)),
    ("shading_focalBlurx4", 100, dict(focal_blur=True, sub_pixel_res=4)),
    ("noShading_focalBlurx2", 100, dict(shading=False, focal_blur=True, sub_pixel_res=2)),
    ("noShading_focalBlurx4", 100, dict(shading=False, focal_blur=True, sub_pixel_res=4)),
    ("shading_shadows", 100, dict(shadows=True)),
    ("noShading_shadows", 100, dict(shading=False, shadows=True)),
    ("shading_shadows_4xAA", 100, dict(shadows=True, sub_pixel_res=4)),
    ("shading_shadows_focalBlurx2", 100, dict(shadows=True, focal_blur=True, sub_pixel_res=2)),
    ("shading_shadows_focalBlurx4", 100, dict(shadows=True, focal_blur=True, sub_pixel_res=4)),
    ("noShading_shadows_4xAA", 100, dict(shading=False, shadows=True, sub_pixel_res=4)),
    ("noShading_shadows_focalBlurx2", 100, dict(shading=False, shadows=True, focal_blur=True, sub_pixel_res=2)),
    ("noShading_shadows_focalBlurx4", 100, dict(shading=False, shadows=True, focal_blur=True, sub_pixel_res=4)),
    ("shading_shadows_4xAA", 50, dict(shadows=True, sub_pixel_res=4)),
    ("shading_shadows_focalBlurx4", 50, dict(shadows=True, focal_blur=True, sub_pixel_res=4)),
]


def golden_rgb(name, res):
    return read_bmp_rgb(os.path.join(GOLDEN, "raytrace", "%dx%d" % (res, res), name + ".bmp"))


@pytest.mark.parametrize("name,res,kw", GOLDENS, ids=["%s_%d" % (g[0], g[1]) for g in GOLDENS])
def test_golden_images(obj_scene, name, res, kw):
    """0 differing RGB pixels, like RendererTests.RenderAndTest (RendererTests.cs:511-544)."""
    px, _ = obj_scene.render(make_frame(res, **kw), threads=os.cpu_count())
    got = px.reshape(res, res) & 0xFFFFFF
    assert int(np.count_nonzero(got != golden_rgb(name, res))) == 0
    assert np.all((px >> 24) == 0xFF)


def test_golden_anchor_counts():
    g = golden_rgb("noShading", 100)     # SURVEY 8c sanity anchor
    vals, counts = np.unique(g, return_counts=True)
    assert dict(zip(vals.tolist(), counts.tolist())) == {0x969696: 6716, 0xff00ff: 3284}


@pytest.mark.parametrize("mode", [orc.MODE_BRUTE, orc.MODE_NEAREST])
def test_modes_agree_on_obj(obj_scene, mode):
    a, _ = obj_scene.render(make_frame(100, shadows=True), threads=os.cpu_count())
    b, _ = obj_scene.render(make_frame(100, shadows=True, mode=mode), threads=os.cpu_count())
    assert np.array_equal(a, b)


def test_rows_and_strips(obj_scene):
    full, _ = obj_scene.render(make_frame(64), threads=4)
    full = full.reshape(64, 64)
    part, _ = obj_scene.render(make_frame(64, start_row=10, end_row=20), threads=2)
    part = part.reshape(64, 64)
    assert np.array_equal(part[10:21], full[10:21]) and not part[:10].any() and not part[21:].any()
    rebuilt = np.zeros_like(full)
    for k in range(3):
        px, _ = obj_scene.render(make_frame(64, strips=(4, 3, k)), threads=2)
        rows = [r for r in range(64) if (r // 4) % 3 == k]
        rebuilt[rows] = px.reshape(len(rows), 64)
    assert np.array_equal(rebuilt, full)


def test_threads_do_not_change_result(obj_scene):
    a, sa = obj_scene.render(make_frame(48, shadows=True, sub_pixel_res=2), threads=1)
    b, sb = obj_scene.render(make_frame(48, shadows=True, sub_pixel_res=2), threads=5)
    assert np.array_equal(a, b) and np.array_equal(sa, sb)


# ---------------------------------------------------------------- rayTraceShadowsStatic
@pytest.mark.parametrize("name,kw", [("shading_staticShadows", dict()), ("noShading_staticShadows", dict(shading=False))])
def test_static_shadow_goldens(name, kw):
    """RendererTests.RaytraceStaticShadow (RendererTests.cs:167-175): the reference fills its 128^3 shadow cache from four
    racing row-block tasks; the oracle's deterministic order (the blocks in lock step) reproduces both goldens exactly.
    Plain scan order does not (9 pixels on the block seams differ), which is what pins the definition."""
    v9, argb, bmin, bmax = load_obj3ds()
    s = orc.Scene()
    s.set_triangles(v9, argb, bmin, bmax)
    assert s.build_tree() == 0
    px, _ = s.render(make_frame(100, shadows=True, static_shadows=True, **kw))
    assert int(np.count_nonzero((px.reshape(100, 100) & 0xFFFFFF) != golden_rgb(name, 100))) == 0
    # one task = plain scan order: a different (also deterministic) fill order, different seam pixels
    s.reset_shadow_cache()
    px1, _ = s.render(make_frame(100, shadows=True, static_shadows=True, concurrency=1, **kw))
    assert 0 < int(np.count_nonzero(px1 != px)) < 50


def test_static_shadow_cache_outlives_the_frame():
    """The cache belongs to the renderer (ShadowMethod is created once, Renderer.cs:1621-1628): a second frame reuses the
    cells of the first, so rendering B after A differs from rendering B with an empty cache; reset = new Renderer."""
    v9, argb, bmin, bmax = load_obj3ds()
    s = orc.Scene()
    s.set_triangles(v9, argb, bmin, bmax)
    assert s.build_tree() == 0
    fa = make_frame(64, shadows=True, static_shadows=True)
    fb = make_frame(64, shadows=True, static_shadows=True, yaw_deg=100.0)
    a1, _ = s.render(fa)
    b_after_a, _ = s.render(fb)
    a2, _ = s.render(fa)
    assert np.array_equal(a1, a2)                       # every cell A needs is already there
    s.reset_shadow_cache()
    b_fresh, _ = s.render(fb)
    assert not np.array_equal(b_after_a, b_fresh)
    with pytest.raises(RuntimeError):                    # a frame split over ranks has no single fill order
        s.render(make_frame(32, shadows=True, static_shadows=True, strips=(4, 2, 0)))


def test_render_window_draws_the_same_pixels():
    """orc_render_window (the CPU baseline's centred crop in bench.py): the windowed pixels are orc_render's, the rest is untouched."""
    o = orc.Scene()
    o.set_triangles(*load_obj3ds())
    assert o.build_tree() == 0
    f = make_frame(48, 40, shadows=True)
    full, _ = o.render(f, threads=4)
    full = full.reshape(40, 48)
    f.start_row, f.end_row = 9, 30
    win = np.full(48 * 40, 7, dtype=np.int32)
    got, _ = o.render(f, threads=3, out=win, cols=(5, 41))
    got = got.reshape(40, 48)
    assert np.array_equal(got[9:31, 5:41], full[9:31, 5:41])
    mask = np.ones((40, 48), dtype=bool)
    mask[9:31, 5:41] = False
    assert (got[mask] == 7).all()
