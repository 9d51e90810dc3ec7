"""GPU parity tests proper: the HIP path, called through the C ABI, against the CPU oracle on the
same inputs and against the reference's golden BMPs.  Bar: bit-exact ARGB / bit-exact doubles."""
import math
import os

import numpy as np
import pytest

import softray_amd as sa
from helpers import (GOLDEN, c1_spheres, leaf_face_scene, load_obj3ds, make_frame, orc, random_triangles, read_bmp_rgb, unit_cube_scene)
from test_oracle import GOLDENS, TREE_BOX, golden_rgb

pytestmark = pytest.mark.gpu
NCPU = os.cpu_count() or 8


def as_sr(frame, mode=None, single_kernel=False, per_lane=False):
    f = sa.Frame.from_buffer_copy(bytes(frame))
    if mode is not None:
        f.trace_mode = mode
    if single_kernel:
        f.flags |= sa._lib.F_SINGLE_KERNEL
    if per_lane:
        f.flags |= sa._lib.F_PER_LANE_SHADOWS
    return f


def render_both(g, frame, mode=None):
    """The default path (k_primary -> k_shadow_packet / k_shadow -> k_resolve), the per-lane shadow kernel and the
    one-kernel renderer are three independent schedules of the same arithmetic: they must agree bit for bit."""
    a, sa_ = g.render(as_sr(frame, mode))
    b, sb_ = g.render(as_sr(frame, mode, single_kernel=True))
    assert np.array_equal(a, b), "pipeline and single-kernel renderer differ"
    used = frame.trace_mode if mode is None else mode
    if used == sa.MODE_BVH:
        # on the own BVH the node / triangle-test counters are the schedule's (the pipeline walks the tree once per 8x8-pixel
        # tile, the one-kernel renderer once per ray); rays fired is the reference's notion and must agree.  With private
        # walks in both (sr_debug_set) the counters agree entirely
        assert sa_[0] == sb_[0]
        g.debug_set(sa._lib.DBG_PER_LANE_PRIMARY, 1)
        a2, sa2 = g.render(as_sr(frame, mode))
        g.debug_set(sa._lib.DBG_PER_LANE_PRIMARY, -1)
        assert np.array_equal(a2, a) and np.array_equal(sa2, sb_)
    else:
        assert np.array_equal(sa_, sb_)
    if frame.flags & sa.F_SHADOWS:
        c, _ = g.render(as_sr(frame, mode, per_lane=True))
        assert np.array_equal(a, c), "k_shadow_packet and k_shadow differ"
    return a, sa_


@pytest.fixture(scope="module")
def obj_pair():
    v9, argb, bmin, bmax = load_obj3ds()
    g = sa.GpuScene(0)
    g.load_3ds(open(os.path.join(GOLDEN, "obj.3ds"), "rb").read())      # the product's own loader
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH))
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    assert o.build_tree() == 0
    return g, o


@pytest.fixture(scope="module")
def obj_variants(obj_pair):
    """obj.3DS in every form a host can ship it: the literal reference tree (its shadow rays on the own BVH's shaft path -- what the
    hosts' TraversalCounters.Auto / Literal run), the own BVH as the library builds it by default (on the device) and as the host's
    binned-SAH builder makes it."""
    g, o = obj_pair
    data = open(os.path.join(GOLDEN, "obj.3ds"), "rb").read()
    out = {"ref_tree": (g, sa.MODE_REF_TREE)}
    for name, on_device in (("bvh_device_built", True), ("bvh_host_built", False)):
        gv = sa.GpuScene(0)
        gv.load_3ds(data)
        gv.build((sa.MODE_BVH,), on_device=on_device)
        assert gv.bvh_stats()[3] == (1 if on_device else 0)
        out[name] = (gv, sa.MODE_BVH)
    return out, o


STATIC_GOLDENS = [("shading_staticShadows", 100, dict(shadows=True, static_shadows=True)),
                  ("noShading_staticShadows", 100, dict(shading=False, shadows=True, static_shadows=True))]


@pytest.mark.parametrize("variant", ["ref_tree", "bvh_device_built", "bvh_host_built"])
@pytest.mark.parametrize("name,res,kw", GOLDENS + STATIC_GOLDENS, ids=["%s_%d" % (g[0], g[1]) for g in GOLDENS + STATIC_GOLDENS])
def test_goldens_on_gpu(obj_variants, name, res, kw, variant):
    """All 22 reference goldens x every structure a host can ship (RendererTests.cs:381-430, 511-544): 0 differing RGB pixels, equal to
    the oracle in all 32 bits, and -- on the literal tree -- the reference's four counters."""
    variants, o = obj_variants
    g, mode = variants[variant]
    f = make_frame(res, **kw)
    if kw.get("static_shadows"):
        g.reset_shadow_cache(); o.reset_shadow_cache()
        got, gstats = g.render(as_sr(f, mode))
    else:
        got, gstats = render_both(g, f, mode)
    want, ostats = o.render(f, threads=NCPU)
    assert int(np.count_nonzero((got.reshape(res, res) & 0xFFFFFF) != golden_rgb(name, res))) == 0
    assert np.array_equal(got, want)
    if mode == sa.MODE_REF_TREE:
        assert np.array_equal(gstats, ostats)        # NumRaysFired / NumGeometryTests / NumNodeVisits / NumLeafNodeVisits
        if f.flags & sa.F_SHADOWS and not kw.get("static_shadows"):
            lit = as_sr(f, mode); lit.flags |= sa._lib.F_LITERAL_SECONDARY
            got2, gstats2 = g.render(lit)            # shadow rays through the reference tree too: same pixels, same primary counters
            assert np.array_equal(got2, want) and np.array_equal(gstats2, ostats)
    else:
        assert gstats[0] == ostats[0]


@pytest.mark.parametrize("mode,omode", [(sa.MODE_BRUTE, orc.MODE_BRUTE), (sa.MODE_BVH, orc.MODE_NEAREST)])
@pytest.mark.parametrize("kw", [dict(), dict(shadows=True), dict(shadows=True, focal_blur=True, sub_pixel_res=2),
                                dict(shading=False, sub_pixel_res=3), dict(point_light=False, shadows=True),
                                dict(specular=False)])
def test_modes_match_oracle(obj_pair, mode, omode, kw):
    g, o = obj_pair
    f = make_frame(96, 64, mode=omode, **kw)
    got, _ = render_both(g, f, mode)
    want, _ = o.render(f, threads=NCPU)
    assert np.array_equal(got, want)
    ref, _ = o.render(make_frame(96, 64, **kw), threads=NCPU)
    assert np.array_equal(got, ref)                  # and all three modes give the reference-tree image


def test_leaf_face_exception_is_what_the_header_says():
    """include/softray.h: SR_MODE_BVH is "identical to REF_TREE except for hits closer than 1e-10 to a leaf-box face of the reference
    tree".  The adversarial scene (helpers.leaf_face_scene): the literal tree returns the FARTHER triangle T (red) for rays that
    cross the split plane where T sits inside the first leaf's containment slack -- exactly what the reference does (oracle, tree
    mode) -- while brute force and the own BVH return the nearer T2 (green) -- exactly what the reference's brute-force path does
    (oracle, brute / nearest).  The difference is confined to the rays through x = 0: one pixel column of the frame."""
    v9, argb, bmin, bmax = leaf_face_scene()
    g = sa.GpuScene(0); o = orc.Scene()
    for s_ in (g, o):
        s_.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH), 5, 2); assert o.build_tree(5, 2) == 0
    assert g.tree_stats() == o.tree_stats() == (2, 3, 2, 1)
    n = 33
    starts = np.zeros((n, 3)); starts[:, 2] = -1.0; starts[:, 1] = np.linspace(-0.5, 0.5, n)
    dirs = np.tile([0.0, 0.0, 1.0], (n, 1))
    got = {}
    for target, otarget in ((sa.MODE_REF_TREE, 1), (sa.MODE_BRUTE, 0), (sa.MODE_BVH, 3)):
        a = g.trace(target, starts, dirs); b = o.trace(otarget, starts, dirs)
        for key in ("hit", "tri_index", "color", "ray_frac", "pos", "normal"):
            assert np.array_equal(a[key], b[key]), (target, key)
        got[target] = a["tri_index"]
    assert np.all(got[sa.MODE_REF_TREE] == 0) and np.all(got[sa.MODE_BRUTE] == 1) and np.all(got[sa.MODE_BVH] == 1)
    # the same through Render(): camera on the z axis looking along +z; the pixel column col = W / 2 has dir.x = 0 exactly
    for kw in (dict(shading=False), dict(), dict(shadows=True)):
        tree_f = make_frame(64, yaw_deg=0.0, pitch_deg=0.0, depth=1.0, **kw)
        near_f = make_frame(64, yaw_deg=0.0, pitch_deg=0.0, depth=1.0, mode=orc.MODE_NEAREST, **kw)
        want_tree, _ = o.render(tree_f, threads=NCPU); want_near, _ = o.render(near_f, threads=NCPU)
        a, _ = g.render(as_sr(tree_f, sa.MODE_REF_TREE)); b, _ = g.render(as_sr(tree_f, sa.MODE_BVH)); c, _ = g.render(as_sr(tree_f, sa.MODE_BRUTE))
        assert np.array_equal(a, want_tree) and np.array_equal(b, want_near) and np.array_equal(c, want_near), kw
        d = (a != b).reshape(64, 64)
        assert d.sum() == 64 and np.all(d[:, 32]), kw                # one column, every row
        if not kw:
            continue
        if kw.get("shading") is False:
            assert np.all(a.reshape(64, 64)[:, 32] == 0xFFFF0000) and np.all(b.reshape(64, 64)[:, 32] == 0xFF00FF00)


def _rays(rnd, n, outside_in):
    u = rnd.NextDoubles(6 * n).reshape(n, 6)
    if outside_in:
        starts = u[:, 0:3] * 1000.0
        return starts, u[:, 3:6] * 100.0 - starts
    return u[:, 0:3] * 100.0, 2.0 * u[:, 3:6] + -1.0


@pytest.mark.parametrize("n_tris,max_depth,max_geom,seed,n_rays,outside_in", [
    (100, 10, 5, 12345, 100000, False),      # TreeCorrectness1
    (20, 10, 1, 123456, 100000, False),      # TreeCorrectness2
    (20, 10, 1, 123456, 100000, True),       # TreeCorrectness_OutsideIn
    (100, 10, 5, 1234567, 265896, False),    # ..._EnsureIntersectionCheckedAgainstTreeNodeBoundingBox
    (10000, 10, 5, 1234567, 2000, False),    # ...BoundingBox2 (34 rays in the reference; more here)
    (1000, 10, 5, 12345, 50000, True),
])
def test_intersect_ray_batches(n_tris, max_depth, max_geom, seed, n_rays, outside_in):
    """IRayIntersectable.IntersectRay on the device == oracle, bit for bit, for the tree, brute force and the BVH."""
    v9, argb, rnd = random_triangles(n_tris, seed=seed)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, *TREE_BOX)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH), max_depth, max_geom)
    o = orc.Scene()
    o.set_triangles(v9, argb, *TREE_BOX)
    assert o.build_tree(max_depth, max_geom) == 0
    assert g.tree_stats() == o.tree_stats()
    starts, dirs = _rays(rnd, n_rays, outside_in)
    for target, otarget in ((sa.MODE_REF_TREE, 1), (sa.MODE_BRUTE, 0), (sa.MODE_BVH, 3)):
        if target == sa.MODE_BRUTE and n_tris > 1000:
            continue
        a = g.trace(target, starts, dirs, counters=True)
        b = o.trace(otarget, starts, dirs, counters=True)
        for key in ("hit", "tri_index", "color", "ray_frac", "pos", "normal"):
            assert np.array_equal(a[key], b[key]), (target, key)
        if target != sa.MODE_BVH:
            assert np.array_equal(a["counters"], b["counters"])
    assert a["hit"].any()


def _clip_edge_rays(lo, hi, seed):
    """Rays chosen to stress AxisAlignedBox.ClipLineSegment: entries exactly through corners and edges, rays that run
    inside a face plane, axis-parallel rays, starts on faces, starts with zero components, rays that miss by 1e-9..1e-12,
    and outside-in rays aimed at random points within 1e-7 of the faces, edges and corners."""
    rng = np.random.default_rng(seed)
    lo, hi = np.asarray(lo, float), np.asarray(hi, float)
    mid = 0.5 * (lo + hi)
    ext = hi - lo
    starts, dirs = [], []

    def add(s, target):
        s, target = np.asarray(s, float), np.asarray(target, float)
        starts.append(s)
        dirs.append(target - s)
    corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
    for c in corners:
        out = c + (c - mid) * 0.5                      # on the diagonal through the corner: three equal crossings
        add(out, mid)
        add(out, c)
        add(out, c + (mid - c) * 1e-12)
        for a in range(3):                             # along the three edges through the corner, in the edge line itself
            e = c.copy()
            e[a] = mid[a]
            s = c.copy()
            s[a] = c[a] + (c[a] - mid[a])
            add(s, e)
    for a in range(3):
        for side in (lo, hi):
            fc = mid.copy()
            fc[a] = side[a]
            outp = fc.copy()
            outp[a] += (side[a] - mid[a])
            add(outp, mid)                             # axis-parallel through the face centre
            add(fc, mid)                               # start exactly on the face
            b = (a + 1) % 3
            g0 = fc.copy()
            g0[b] = lo[b] - ext[b]
            g1 = fc.copy()
            g1[b] = hi[b] + ext[b]
            add(g0, g1)                                # runs inside the face plane
            for eps in (1e-9, 1e-11, 1e-12, -1e-9, -1e-11, -1e-12):
                t0, t1 = g0.copy(), g1.copy()
                t0[a] += eps
                t1[a] += eps
                add(t0, t1)                            # just inside / just outside the face plane
    add([0.0, 0.0, 0.0], mid)
    add(np.where(np.arange(3) == 0, lo - ext, 0.0), mid)
    add(lo - ext, lo - 2 * ext)                        # points away: never reaches the box
    n = 20000
    tgt = lo + rng.random((n, 3)) * ext
    axis = rng.integers(0, 3, n)
    kind = rng.integers(0, 3, n)                       # snap 1, 2 or 3 coordinates to a face -> face / edge / corner targets
    for i in range(n):
        for j in range(kind[i] + 1):
            a = (axis[i] + j) % 3
            tgt[i, a] = (lo if rng.random() < 0.5 else hi)[a] + rng.normal() * 10.0 ** rng.integers(-13, -6)
    src = mid + (rng.random((n, 3)) - 0.5) * ext * 6.0
    for i in range(n):
        add(src[i], tgt[i] + (tgt[i] - src[i]) * rng.random())
    return np.array(starts), np.array(dirs)


@pytest.mark.parametrize("flat", [False, True])
def test_box_clip_edge_cases(flat):
    """The device's entry-face shortcut must give the literal six-plane ClipLineSegment result (AxisAlignedBox.cs) in
    every degenerate configuration; checked through IntersectRay on the tree and on the BVH (start clip + rayFracOffset)."""
    v9, argb, rnd = random_triangles(3000, seed=424242)
    lo, hi = [list(x) for x in TREE_BOX]
    if flat:                                           # planar model: zero-thickness box
        v9 = v9.reshape(-1, 3, 3).copy()
        v9[:, :, 2] = 5.0
        v9 = v9.reshape(-1)
        lo[2] = hi[2] = 5.0
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, lo, hi)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH), 10, 5)
    o = orc.Scene()
    o.set_triangles(v9, argb, lo, hi)
    assert o.build_tree(10, 5) == 0
    starts, dirs = _clip_edge_rays(lo, hi if not flat else [hi[0], hi[1], 5.0], 99)
    for target, otarget in ((sa.MODE_REF_TREE, 1), (sa.MODE_BVH, 3)):
        a = g.trace(target, starts, dirs)
        b = o.trace(otarget, starts, dirs)
        for key in ("hit", "tri_index", "color", "ray_frac", "pos", "normal"):
            assert np.array_equal(a[key], b[key]), (target, key, np.flatnonzero(a["hit"] != b["hit"])[:10])
    assert 0 < a["hit"].sum() < len(starts)


def test_primitive_kats_on_gpu():
    g = sa.GpuScene(0)
    g.set_triangles(np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float64), np.array([0xffffffff], dtype=np.uint32),
                    [-1, -1, -1], [2, 2, 2])
    r = g.trace(sa.MODE_BRUTE, [[0, 0, 1], [1, 0, 0], [0, 0, -1]], [[0, 0, -1], [0, 0, -1], [0, 0, 1]])
    assert list(r["hit"]) == [1, 1, 0]               # TriangleTests.cs:45-73
    assert r["ray_frac"][0] == 1.0 and r["ray_frac"][1] == 0.0
    assert np.array_equal(r["normal"][0], [0, 0, 1]) and r["color"][0] == 0xffffffff


def test_spheres_planes_triangles_extra_geometry():
    """Config 1: extra geometry (16 spheres) + obj.3DS at depth 3; plus a plane and a triangle."""
    v9, argb, bmin, bmax = load_obj3ds()
    prims = c1_spheres()
    prims.append((1, 0xff808080, [0, -0.45, 0, 0, 1, 0]))
    prims.append((2, 0xff00ffff, [-0.6, -0.3, 0.2, 0.6, -0.3, 0.2, 0.0, 0.5, 0.2]))
    g = sa.GpuScene(0); o = orc.Scene()
    for s in (g, o):
        s.set_triangles(v9, argb, bmin, bmax)
        s.set_extra(prims)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH)); assert o.build_tree() == 0
    for kw in (dict(), dict(shadows=True), dict(sub_pixel_res=2, shadows=True, focal_blur=True)):
        res = 256 if not kw else 96
        f = make_frame(res, depth=3.0, **kw)
        want, _ = o.render(f, threads=NCPU)
        for mode in (sa.MODE_REF_TREE, sa.MODE_BRUTE, sa.MODE_BVH):
            got, _ = render_both(g, f, mode)
            assert np.array_equal(got, want), (kw, mode)
    # ray batches through the root GeometryCollection (sphere rayFrac is a distance: Sphere.cs:164)
    rnd = orc.Random(99)
    u = rnd.NextDoubles(6 * 50000).reshape(-1, 6)
    starts = 4.0 * u[:, :3] - 2.0
    dirs = (u[:, 3:] - 0.5) - starts * 0.5
    a = g.trace(sa.TARGET_ROOT | sa.MODE_REF_TREE, starts, dirs)
    b = o.trace(2, starts, dirs)
    for key in ("hit", "tri_index", "color", "ray_frac", "pos", "normal"):
        assert np.array_equal(a[key], b[key]), key


def test_random_scene_all_modes_vs_oracle_tree():
    """Config-3 style scene at a size the oracle finishes in seconds: 50k random triangles, shading + 100-sample shadows."""
    v9, argb, bmin, bmax = unit_cube_scene(50000)
    g = sa.GpuScene(0); o = orc.Scene()
    for s in (g, o):
        s.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH)); assert o.build_tree() == 0
    assert g.tree_stats() == o.tree_stats()
    f = make_frame(128, depth=1.5, shadows=True)
    want, ostats = o.render(f, threads=NCPU)
    got, gstats = render_both(g, f, sa.MODE_REF_TREE)
    assert np.array_equal(got, want) and np.array_equal(gstats, ostats)
    got_bvh, _ = render_both(g, f, sa.MODE_BVH)
    assert np.array_equal(got_bvh, want)
    assert len(np.unique(want)) > 1000


def test_tiny_models_on_the_own_bvh():
    """The host mirrors trace every subdivided model through the library's BVH by default: models of 1 ... 10 triangles (a single
    leaf, a root with one or two children, device- and host-built) give the oracle's frames, with and without the 100-sample shadows."""
    for n, both_sides in ((1, False), (1, True), (2, True), (3, True), (5, True)):
        v9, argb, bmin, bmax = unit_cube_scene(n, seed=77 + n)
        v9 = v9.copy()
        v9[:, 1:, :] = v9[:, :1, :] + (v9[:, 1:, :] - v9[:, :1, :]) * 8.0      # big enough to be seen (and to shadow each other)
        v9 = np.clip(v9, -0.5, 0.5)
        if both_sides:                                                          # triangles are one-sided: one of each pair faces the camera
            v9 = np.concatenate([v9, v9[:, [0, 2, 1], :]]); argb = np.concatenate([argb, argb])
        v9 = np.ascontiguousarray(v9); argb = np.ascontiguousarray(argb)
        o = orc.Scene(); o.set_triangles(v9, argb, bmin, bmax); assert o.build_tree() == 0
        for on_device in (True, False):
            g = sa.GpuScene(0)
            g.set_triangles(v9, argb, bmin, bmax)
            g.build((sa.MODE_BVH,), on_device=on_device)
            for kw in (dict(), dict(shadows=True), dict(shadows=True, sub_pixel_res=2)):
                f = make_frame(96, depth=1.5, **kw)
                want, _ = o.render(f, threads=NCPU)
                got, _ = g.render(as_sr(f, sa.MODE_BVH))
                assert np.array_equal(got, want), (n, on_device, kw)
                got2, _ = g.render(as_sr(f, sa.MODE_BVH), stats=False)
                assert np.array_equal(got2, want), (n, on_device, kw, "no stats")
        if both_sides:
            assert len(np.unique(want)) > 2, n


def test_rows_and_strips_on_gpu(obj_pair):
    g, o = obj_pair
    full, _ = g.render(as_sr(make_frame(80, 64, shadows=True)))
    full = full.reshape(64, 80)
    buf = np.full(64 * 80, 0x12345678, dtype=np.int32)
    part, _ = g.render(as_sr(make_frame(80, 64, shadows=True, start_row=10, end_row=20)), out=buf)
    part = part.reshape(64, 80)
    assert np.array_equal(part[10:21], full[10:21])
    assert np.all(part[:10] == 0x12345678) and np.all(part[21:] == 0x12345678)   # untouched, like Surface.DrawPixel
    rebuilt = np.zeros_like(full)
    for k in range(3):
        f = as_sr(make_frame(80, 64, shadows=True, strips=(4, 3, k)))
        px, _ = g.render(f)
        rows = [r for r in range(64) if (r // 4) % 3 == k]
        assert g.pixel_count(f) == len(rows) * 80
        rebuilt[rows] = px.reshape(len(rows), 80)
    assert np.array_equal(rebuilt, full)


def test_error_codes_on_gpu(obj_pair):
    g, _ = obj_pair
    f = as_sr(make_frame(16))
    f.max_bounces = 20
    with pytest.raises(sa.SoftrayError) as e:
        g.render(f)
    assert e.value.code == sa._lib.SR_ERR_INVALID_ARG
    g2 = sa.GpuScene(0)
    with pytest.raises(sa.SoftrayError) as e:
        g2.render(as_sr(make_frame(16)))
    assert e.value.code == sa._lib.SR_ERR_NO_MODEL   # Render() without a model draws nothing (Renderer.cs:736-739)


def test_shadow_sample_variants(obj_pair):
    """ShadowMethod with other sample tables: 1 sample / zero offset (hard-shadow variant), odd counts, a caller-supplied
    offset table (what the C# shim passes), and > 128 samples (the shaft path in chunks of 128, escape counts summed per hit point:
    130 = 128 + 2, 200, 300 = three chunks; render_both adds the one-kernel and the per-lane schedule as cross-checks)."""
    g, o = obj_pair
    rnd = orc.Random(4242)
    for count in (1, 7, 64, 65, 100, 128, 130, 200, 300):
        f = make_frame(72, 56, shadows=True, shadow_samples=count)
        if count == 1:
            table = np.zeros((1, 3))
        elif count == 100:
            table = orc.area_light_offsets(1234567890, 100) * 1.5         # a different light radius
        else:
            table = None
        if table is not None:
            table = np.ascontiguousarray(table)
            f.area_light_offsets = table.ctypes.data
        want, _ = o.render(f, threads=NCPU)
        for mode in (sa.MODE_BVH, sa.MODE_REF_TREE):
            got, _ = render_both(g, f, mode)
            assert np.array_equal(got, want), (count, mode)
        if count > 128:                                   # sub-pixel sampling (band-local sample indices), twice (the sums must be cleared)
            f2 = make_frame(72, 56, shadows=True, shadow_samples=count, sub_pixel_res=2)
            want2, _ = o.render(f2, threads=NCPU)
            for _ in range(2):
                assert np.array_equal(g.render(as_sr(f2, sa.MODE_BVH), stats=False)[0], want2), count


def test_shadow_method_shortcuts_change_no_pixel(obj_pair):
    """Two proofs let a frame skip literal shadow rays (sr_api.cpp render_common): every sample of a directional light escapes
    when the model is small against the 1000-unit start offset (ShadowMethod.cs:160-166), and a SR_MODE_REF_TREE frame whose caller
    does not read the traversal counters answers its shadow rays on the own BVH (shaft path).  Both against the literal schedule
    (SR_DBG_LITERAL_SHADOWS), the oracle, and -- for a model 3000 units across -- the case in which proof (1) does not hold."""
    g, o = obj_pair
    for kw in (dict(point_light=False, shadows=True), dict(point_light=False, shadows=True, sub_pixel_res=2), dict(shadows=True),
               dict(shadows=True, sub_pixel_res=2, focal_blur=True), dict(shadows=True, start_row=11, end_row=50)):
        f = make_frame(88, 64, **kw)
        want, _ = o.render(f, threads=NCPU)
        for mode in (sa.MODE_REF_TREE, sa.MODE_BRUTE, sa.MODE_BVH):
            fast, _ = g.render(as_sr(f, mode), stats=False)
            g.debug_set(sa._lib.DBG_LITERAL_SHADOWS, 1)
            literal, _ = g.render(as_sr(f, mode), stats=False)
            g.debug_set(sa._lib.DBG_LITERAL_SHADOWS, -1)
            with_stats, _ = g.render(as_sr(f, mode), stats=True)
            assert np.array_equal(fast, want) and np.array_equal(literal, want) and np.array_equal(with_stats, want), (kw, mode)
    # a random soup through the reference tree: shadow rays on the BVH (no counters asked) == literal tree == oracle
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    g2 = sa.GpuScene(0); o2 = orc.Scene()
    for s_ in (g2, o2):
        s_.set_triangles(v9, argb, bmin, bmax)
    g2.build((sa.MODE_REF_TREE, sa.MODE_BVH)); assert o2.build_tree() == 0
    f = make_frame(120, 90, depth=1.5, shadows=True)
    want, _ = o2.render(f, threads=NCPU)
    assert np.array_equal(g2.render(as_sr(f, sa.MODE_REF_TREE), stats=False)[0], want)
    assert np.array_equal(g2.render(as_sr(f, sa.MODE_REF_TREE), stats=True)[0], want)
    # proof (1) must NOT be used for a model that reaches the 1000-unit start offset: scale the scene to 3000 units
    scale = 3000.0
    g3 = sa.GpuScene(0); o3 = orc.Scene()
    for s_ in (g3, o3):
        s_.set_triangles(v9 * scale, argb, bmin * scale, bmax * scale)
    g3.build((sa.MODE_BVH,)); assert o3.build_tree() == 0
    f = make_frame(64, 48, depth=1.5 * scale, shadows=True, point_light=False, mode=orc.MODE_NEAREST)
    want, _ = o3.render(f, threads=NCPU)
    got, _ = g3.render(as_sr(f, sa.MODE_BVH), stats=False)
    assert np.array_equal(got, want)
    lit = make_frame(64, 48, depth=1.5 * scale, shadows=False, point_light=False, mode=orc.MODE_NEAREST)
    assert not np.array_equal(want, o3.render(lit, threads=NCPU)[0])        # (shadows do occur at this scale: the case is not vacuous)


def test_pipeline_bands_rounds_and_fallback():
    """Force tiny row bands and tiny candidate lists so that a small frame goes through several bands, the second shaft
    round and the exact wave-per-hit fallback; the image must not change."""
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    g = sa.GpuScene(0); o = orc.Scene()
    for s in (g, o):
        s.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,)); assert o.build_tree() == 0
    f = make_frame(96, 80, depth=1.5, shadows=True)
    want, _ = o.render(f, threads=NCPU)
    base, _ = g.render(as_sr(f, sa.MODE_BVH))
    assert np.array_equal(base, want)
    g.debug_set(sa._lib.DBG_BAND_SAMPLES, 2000)
    g.debug_set(sa._lib.DBG_ROUND_CAP0, 2)
    g.debug_set(sa._lib.DBG_ROUND_CAP1, 3)
    got, _ = g.render(as_sr(f, sa.MODE_BVH))
    assert np.array_equal(got, want)
    c = g.debug_counters()
    assert c[2] > 0 and c[3] > 0, c            # round 2 and the fallback were really exercised
    # the fallback's ray list too small for most entries: they take the one-wave-per-hit kernel instead
    g.debug_set(sa._lib.DBG_FB_RAY_CAP, 40)
    got, _ = g.render(as_sr(f, sa.MODE_BVH))
    assert np.array_equal(got, want)
    g.debug_set(sa._lib.DBG_FB_RAY_CAP, -1)
    f2 = make_frame(96, 80, depth=1.5, shadows=True, sub_pixel_res=2, focal_blur=True)
    want2, _ = o.render(f2, threads=NCPU)
    got2, _ = g.render(as_sr(f2, sa.MODE_BVH))
    assert np.array_equal(got2, want2)


def test_fp32_classification_against_exact_schedule():
    """k_shadow_cls (fp32 classification of (sample, triangle) pairs with a rigorous error bound, FP64 only for the pairs it
    cannot decide) against k_shadow_test (every pair in FP64) and the oracle: a soup, a scene whose triangles touch the root
    box faces and sit in a few planes (crossings exactly on edges / box faces: many undecidable pairs), a light inside the
    root box, and extra geometry."""
    # (a) soup
    v9, argb, bmin, bmax = unit_cube_scene(40000)
    # (b) axis-aligned quads split into triangles on a lattice, touching the box faces: shared edges, coplanar neighbours
    q = []
    n = 12
    for i in range(n):
        for j in range(n):
            for z in (-0.5, -0.25, 0.0, 0.5):
                x0, x1 = -0.5 + i / n, -0.5 + (i + 1) / n
                y0, y1 = -0.5 + j / n, -0.5 + (j + 1) / n
                if (i + j + int(z * 4)) % 3 == 0:
                    continue
                q.append([[x0, y0, z], [x1, y0, z], [x1, y1, z]])
                q.append([[x0, y0, z], [x1, y1, z], [x0, y1, z]])
                q.append([[x0, z, y0], [x1, z, y1], [x1, z, y0]])      # the same pattern in y = z planes, other winding
    lattice = np.array(q, dtype=np.float64)
    lat_argb = (0xFF000000 | (np.arange(len(lattice), dtype=np.uint64) * 2654435761 & 0xFFFFFF)).astype(np.uint32)
    seen = [0, 0]
    scenes = [("soup", v9, argb, {}), ("lattice", lattice, lat_argb, {}),
              ("soup_light_inside", v9[:8000], argb[:8000], dict(light_inside=True))]
    for name, tv, ta, opt in scenes:
        g = sa.GpuScene(0); o = orc.Scene()
        for s_ in (g, o):
            s_.set_triangles(tv, ta, bmin, bmax)
        g.build((sa.MODE_BVH,)); assert o.build_tree() == 0
        for kw in (dict(), dict(yaw_deg=20.0, pitch_deg=35.0), dict(sub_pixel_res=2)):
            f = make_frame(112, 96, depth=1.5, shadows=True, **kw)
            if opt.get("light_inside"):
                f.light_pos_view[0], f.light_pos_view[1], f.light_pos_view[2] = 0.05, 0.1, 1.45     # inside the unit cube at depth 1.5
            want, _ = o.render(f, threads=NCPU)
            g.debug_set(sa._lib.DBG_EXACT_SHADOW_TESTS, -1)
            got, _ = g.render(as_sr(f, sa.MODE_BVH))
            st = g.ray_stats()
            assert np.array_equal(got, want), (name, kw)
            if st[4] > 0:                                             # (k_shaft may decide every hit point of a frame by itself)
                assert st[12] > 0 and st[13] < st[12], (name, st)     # pairs were classified in fp32, only a part needed FP64
                seen[0] += int(st[12]); seen[1] += int(st[13])
            g.debug_set(sa._lib.DBG_EXACT_SHADOW_TESTS, 1)
            exact, _ = g.render(as_sr(f, sa.MODE_BVH))
            assert np.array_equal(exact, want), (name, kw)
            assert g.ray_stats()[12] == 0
        if name == "soup":
            prims = c1_spheres(5)
            g.set_extra(prims); o.set_extra(prims)
            f = make_frame(96, 80, depth=1.5, shadows=True)
            g.debug_set(sa._lib.DBG_EXACT_SHADOW_TESTS, -1)
            assert np.array_equal(g.render(as_sr(f, sa.MODE_BVH))[0], o.render(f, threads=NCPU)[0])
    assert seen[0] > 1000000 and seen[1] > 0, seen                    # both the fp32 verdicts and the FP64 path were exercised


def test_packet_shaft_walk_against_private_walks():
    """k_shaft_pkt (one wave-cooperative walk per 8x8-pixel tile of surface points, tile-aligned hit queue) and k_shaft_coop
    (later rounds: one wave per hit point) against k_shaft (one private walk per lane; bit 0: in round 1, bit 1: in round 2)
    and the oracle: tiles that straddle the silhouette, odd frame sizes (partial tiles), sub-pixel sampling (several queue
    tiles per pixel tile), tiny lists (round 2 re-collects from scratch, then the fallback)."""
    v9, argb, bmin, bmax = unit_cube_scene(30000)
    g = sa.GpuScene(0); o = orc.Scene()
    for s_ in (g, o):
        s_.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,), on_device=False); assert o.build_tree() == 0          # the host's binned-SAH tree (the other tests walk the default device LBVH)
    for kw in (dict(), dict(sub_pixel_res=2), dict(yaw_deg=10.0, pitch_deg=80.0), dict(start_row=5, end_row=70)):
        f = make_frame(117, 91, depth=1.5, shadows=True, **kw)
        want, _ = o.render(f, threads=NCPU)
        for caps in (None, (2, 3), (5, 64), (3, 200)):
            g.debug_set(sa._lib.DBG_ROUND_CAP0, caps[0] if caps else -1)
            g.debug_set(sa._lib.DBG_ROUND_CAP1, caps[1] if caps else -1)
            for per_lane in (0, 1, 2, 3):
                g.debug_set(sa._lib.DBG_PER_LANE_SHAFT, per_lane)
                got, _ = g.render(as_sr(f, sa.MODE_BVH))
                assert np.array_equal(got, want), (kw, caps, per_lane)
            # the packet walks on the binary tree with a per-step vote (round 2's kernels) instead of the four-wide, per-frame
            # ordered tree: a third schedule of the same lists
            g.debug_set(sa._lib.DBG_PER_LANE_SHAFT, 0)
            g.debug_set(sa._lib.DBG_BVH2_PACKETS, 1)
            got, _ = g.render(as_sr(f, sa.MODE_BVH))
            g.debug_set(sa._lib.DBG_BVH2_PACKETS, -1)
            assert np.array_equal(got, want), (kw, caps, "bvh2 packets")
    g.debug_set(sa._lib.DBG_PER_LANE_SHAFT, -1)
    # a light INSIDE the scene's box and one very close to the surface: the light-ordered copy of the nodes has no preferred side
    for lp in ((0.05, 0.1, -0.02), (0.3, 0.45, 0.2)):
        f = make_frame(96, 80, depth=1.5, shadows=True)
        t, it = sa.instance_matrices([0.0, 0.0, 1.5], 135.0 / 180.0 * math.pi, -22.0 / 180.0 * math.pi, 0.0)
        # light_pos_view such that inverseTransform(3x4) * pos lands on `lp` (model space): pos = transform(3x4) * lp
        for r in range(3):
            f.light_pos_view[r] = t[4 * r] * lp[0] + t[4 * r + 1] * lp[1] + t[4 * r + 2] * lp[2] + t[4 * r + 3]
        want, _ = o.render(f, threads=NCPU)
        for b2 in (-1, 1):
            g.debug_set(sa._lib.DBG_BVH2_PACKETS, b2)
            got, _ = g.render(as_sr(f, sa.MODE_BVH))
            assert np.array_equal(got, want), (lp, b2)
    g.debug_set(sa._lib.DBG_BVH2_PACKETS, -1)


def test_packet_primary_walk_against_private_walks():
    """k_primary's packet walk (one traversal per 8x8-pixel tile, camera-cone records filter the FP64 triangle tests) against
    the private per-lane walks and the oracle: odd frame sizes, AA sub-samples (same origin), focal blur (origins differ: the
    packet walk must step aside), extra geometry, a camera inside the root box, a moved camera (cone records re-made), the
    mirror-bounce pipeline, and ray statistics that stay deterministic."""
    v9, argb, bmin, bmax = unit_cube_scene(30000)
    g = sa.GpuScene(0); o = orc.Scene()
    for s_ in (g, o):
        s_.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,), on_device=False); assert o.build_tree() == 0          # the host's binned-SAH tree (the other tests walk the default device LBVH)
    cases = [dict(), dict(sub_pixel_res=3), dict(sub_pixel_res=2, focal_blur=True), dict(depth=0.2), dict(yaw_deg=300.0, pitch_deg=40.0, roll_deg=25.0),
             dict(shadows=True), dict(start_row=9, end_row=40)]
    for kw in cases:
        kw = dict(kw)
        depth = kw.pop("depth", 1.5)
        f = make_frame(107, 85, depth=depth, **kw)
        want, _ = o.render(f, threads=NCPU)
        stats = []
        for per_lane in (0, 1):
            g.debug_set(sa._lib.DBG_PER_LANE_PRIMARY, per_lane)
            got, st = g.render(as_sr(f, sa.MODE_BVH))
            assert np.array_equal(got, want), (kw, per_lane)
            again, st2 = g.render(as_sr(f, sa.MODE_BVH))
            assert np.array_equal(st, st2)
            stats.append(st)
        assert stats[0][0] == stats[1][0]                  # rays fired
        # the packet walk on the binary tree (per-step vote) instead of the four-wide, camera-ordered tree
        g.debug_set(sa._lib.DBG_PER_LANE_PRIMARY, 0)
        g.debug_set(sa._lib.DBG_BVH2_PACKETS, 1)
        got, st = g.render(as_sr(f, sa.MODE_BVH))
        g.debug_set(sa._lib.DBG_BVH2_PACKETS, -1)
        assert np.array_equal(got, want), (kw, "bvh2 packets")
        assert st[0] == stats[0][0]
    g.debug_set(sa._lib.DBG_PER_LANE_PRIMARY, -1)
    prims = c1_spheres(7)
    prims.append((1, 0xff808080, [0, -0.45, 0, 0, 1, 0]))
    g.set_extra(prims); o.set_extra(prims)
    f = make_frame(96, 72, depth=1.8)
    assert np.array_equal(g.render(as_sr(f, sa.MODE_BVH))[0], o.render(f, threads=NCPU)[0])
    f.max_bounces, f.reflectivity = 3, 0.5
    assert np.array_equal(g.render(as_sr(f, sa.MODE_BVH))[0], o.render(f, threads=NCPU)[0])
    # the golden model through the packet walk
    g2 = sa.GpuScene(0)
    g2.load_3ds(open(os.path.join(GOLDEN, "obj.3ds"), "rb").read())
    g2.build((sa.MODE_BVH,))
    got, _ = g2.render(as_sr(make_frame(100), sa.MODE_BVH))
    assert int(np.count_nonzero((got.reshape(100, 100) & 0xFFFFFF) != golden_rgb("shading", 100))) == 0


def test_multi_device_scene_in_library():
    """sr_create_multi: one scene over several devices of one process (here the same GPU several times -- the code path, the
    strip bookkeeping and the strided gathers are those of a real node): host surface and device surface, odd row ranges,
    shadows, sub-pixel sampling, statistics, extra geometry, static shadows (rendered whole by the first device)."""
    import torch
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    single = sa.GpuScene(0)
    single.set_triangles(v9, argb, bmin, bmax)
    single.build((sa.MODE_BVH, sa.MODE_REF_TREE))
    for ndev in (2, 3, 8):
        multi = sa.GpuScene(devices=[0] * ndev)
        assert multi.device_count() == ndev
        multi.set_triangles(v9, argb, bmin, bmax)
        multi.build((sa.MODE_BVH, sa.MODE_REF_TREE))
        for kw, mode in ((dict(shadows=True), sa.MODE_BVH), (dict(sub_pixel_res=2), sa.MODE_BVH), (dict(start_row=5, end_row=190), sa.MODE_REF_TREE),
                         (dict(start_row=37, end_row=41), sa.MODE_BVH), (dict(shadows=True, static_shadows=True), sa.MODE_BVH)):
            f = as_sr(make_frame(150, 203, depth=1.5, **kw), mode)
            single.reset_shadow_cache(); multi.reset_shadow_cache()
            canvas_a = np.full(150 * 203, 0x12345678, dtype=np.int32); canvas_b = canvas_a.copy()
            a, sta = single.render(f, out=canvas_a)
            b, stb = multi.render(f, out=canvas_b)
            assert np.array_equal(a, b), (ndev, kw)
            if mode == sa.MODE_REF_TREE:
                assert np.array_equal(sta, stb)                          # the reference's counters do not depend on the split
            assert sta[0] == stb[0]
        # device surface on devices[0], enqueued on a torch stream, twice in a row without a host sync in between
        f = as_sr(make_frame(150, 203, depth=1.5, shadows=True), sa.MODE_BVH)
        want, _ = single.render(f)
        dev = torch.device("cuda", 0)
        out = torch.zeros(150 * 203, dtype=torch.int32, device=dev)
        st = torch.cuda.current_stream(dev)
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(dev)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
        # the gather of a part whose memory the first device cannot read: through pinned host staging (hook: every part)
        multi.debug_set(sa._lib.DBG_NO_PEER, 1)
        out.zero_()
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(dev)
        multi.debug_set(sa._lib.DBG_NO_PEER, -1)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want)
        # a large surface (the caller's buffer is pinned for the call, every device copies its strips on its own stream)
        fl = as_sr(make_frame(1024, 1100, depth=1.5, shadows=True), sa.MODE_BVH)
        big = np.full(1024 * 1100, 7, dtype=np.int32)
        got_l, _ = multi.render(fl, out=big)
        assert np.array_equal(got_l, single.render(fl)[0])
        prims = c1_spheres(4)
        multi.set_extra(prims); single.set_extra(prims)
        f = as_sr(make_frame(96, 64, depth=1.5, shadows=True), sa.MODE_BVH)
        assert np.array_equal(multi.render(f)[0], single.render(f)[0])
        single.set_extra([])
        multi.close()


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs: sr_create_multi over distinct devices (peer gather over xGMI)")
def test_multi_device_scene_over_distinct_gpus():
    """The cross-device path of sr_create_multi on real peers (never exercised on a one-GPU box): host surface, device surface by
    peer-to-peer copies, and the staged gather, each against the single-GPU frame."""
    import torch
    ndev = min(torch.cuda.device_count(), 8)
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    single = sa.GpuScene(0)
    single.set_triangles(v9, argb, bmin, bmax)
    single.build((sa.MODE_BVH,))
    multi = sa.GpuScene(devices=list(range(ndev)))
    multi.set_triangles(v9, argb, bmin, bmax)
    multi.build((sa.MODE_BVH,))
    f = as_sr(make_frame(640, 515, depth=1.5, shadows=True), sa.MODE_BVH)
    want, _ = single.render(f)
    assert np.array_equal(multi.render(f)[0], want)
    dev = torch.device("cuda", 0)
    out = torch.zeros(640 * 515, dtype=torch.int32, device=dev)
    st = torch.cuda.current_stream(dev)
    for no_peer in (-1, 1):
        multi.debug_set(sa._lib.DBG_NO_PEER, no_peer)
        out.zero_()
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        multi.render_device(f, out.data_ptr(), st.cuda_stream)
        torch.cuda.synchronize(dev)
        assert np.array_equal(out.cpu().numpy().view(np.uint32), want), no_peer
    multi.close()


def test_frames_on_two_streams_without_host_syncs():
    """sr_render_device accepts any stream.  The per-origin / per-light records (facing partition, camera cones, ordered node copies) and
    the frame tables are written on whatever stream the frame that needs them runs on and are reused by later frames: frames with
    different cameras, lights and sample tables alternate between two streams with no host synchronisation in between -- the library
    orders them with events -- and every one must equal its blocking render."""
    import torch
    v9, argb, bmin, bmax = unit_cube_scene(60000)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,))
    dev = torch.device("cuda", 0)
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    frames = []
    for k in range(8):
        f = as_sr(make_frame(200, 160, depth=1.5, shadows=True, yaw_deg=135.0 + 17.0 * (k % 4), pitch_deg=-22.0 + 9.0 * (k // 4),
                             shadow_samples=(100, 37)[k % 2]), sa.MODE_BVH)
        if k % 3 == 2:
            f.light_pos_view[0] += 0.4
        frames.append(f)
    want = [g.render(f, stats=False)[0].copy() for f in frames]
    torch.cuda.synchronize(dev)
    outs = [torch.zeros(200 * 160, dtype=torch.int32, device=dev) for _ in frames]
    for rep in range(2):
        for k, f in enumerate(frames):
            g.render_device(f, outs[k].data_ptr(), streams[k % 2].cuda_stream)
    torch.cuda.synchronize(dev)
    for k in range(len(frames)):
        assert np.array_equal(outs[k].cpu().numpy().view(np.uint32), want[k]), k


def test_full_size_properties():
    """BASELINE-size checks (1 M triangles, up to 4096^2) through size-independent properties: the three shadow schedules
    agree, the union of interleaved strips is the frame, rendering is idempotent, the own BVH and the literal reference
    tree give the same image on the device, and five pairs of rows spread over the frame equal the CPU oracle bit for bit."""
    v9, argb, bmin, bmax = sa.unit_cube_scene(1000000)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH, sa.MODE_REF_TREE))
    # (a) 4096^2, shading + 100-sample shadows
    f = make_frame(4096, depth=1.5, shadows=True, mode=orc.MODE_REF_TREE)
    a, stats = g.render(as_sr(f, sa.MODE_BVH))
    assert stats[0] == 4096 * 4096
    again, _ = g.render(as_sr(f, sa.MODE_BVH))
    assert np.array_equal(a, again)
    lanes, _ = g.render(as_sr(f, sa.MODE_BVH, per_lane=True))
    assert np.array_equal(a, lanes)
    a2 = a.reshape(4096, 4096)
    for k in range(8):                                       # config C4's own split: 8 ranks x interleaved 16-row strips (Renderer.cs:1655-1680)
        fs = make_frame(4096, depth=1.5, shadows=True, strips=(16, 8, k))
        px, _ = g.render(as_sr(fs, sa.MODE_BVH))
        rows = [r for r in range(4096) if (r // 16) % 8 == k]
        assert np.array_equal(px.reshape(len(rows), 4096), a2[rows]), k
    assert len(np.unique(a)) > 10000
    # (b) own BVH == literal reference tree on the device (primary + shading 1024^2; shadows 384^2)
    for res, kw in ((1024, dict()), (384, dict(shadows=True))):
        fb = make_frame(res, depth=1.5, **kw)
        x, _ = g.render(as_sr(fb, sa.MODE_BVH))
        y, _ = g.render(as_sr(fb, sa.MODE_REF_TREE))
        assert np.array_equal(x, y), res
    # (c) two rows of the 4096^2 frame against the CPU oracle (reference tree 15/25)
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    assert o.build_tree() == 0
    # fixed row pairs (object centre, a silhouette, the seam of the two half-frame pipelines, two more across the object); the WHOLE
    # frame is compared with the oracle strip by strip in tests/test_gpu_frames.py (fixtures made by scripts/make_frame_fixtures.py)
    for r0 in [2047, 611, 2790, 1466, 3321]:
        fo = make_frame(4096, depth=1.5, shadows=True, start_row=r0, end_row=r0 + 1)
        want, _ = o.render(fo, threads=NCPU)
        assert np.array_equal(want.reshape(4096, 4096)[r0:r0 + 2], a2[r0:r0 + 2]), r0
    # (c2) config C3's own resolution (2048^2, shading + 100-sample shadows): rendering is idempotent, two row pairs equal the oracle
    f3 = make_frame(2048, depth=1.5, shadows=True)
    c3, _ = g.render(as_sr(f3, sa.MODE_BVH))
    assert np.array_equal(c3, g.render(as_sr(f3, sa.MODE_BVH), stats=False)[0])
    c3 = c3.reshape(2048, 2048)
    for r0 in (1023, 1433):
        fo = make_frame(2048, depth=1.5, shadows=True, start_row=r0, end_row=r0 + 1)
        want, _ = o.render(fo, threads=NCPU)
        assert np.array_equal(want.reshape(2048, 2048)[r0:r0 + 2], c3[r0:r0 + 2]), ("C3", r0)
    # (d) static shadow cache at 4096^2: a warm cache reproduces the frame (every cell it needs exists), shadow-less pixels are
    #     untouched, and every shadowed pixel is its shaded colour modulated by SOME cache byte 1..255
    fs = make_frame(4096, depth=1.5, shadows=True, static_shadows=True)
    g.reset_shadow_cache()
    s1, _ = g.render(as_sr(fs, sa.MODE_BVH))
    s2, _ = g.render(as_sr(fs, sa.MODE_BVH))
    assert np.array_equal(s1, s2)
    plain, _ = g.render(as_sr(make_frame(4096, depth=1.5), sa.MODE_BVH))
    bg = plain == 0xFFFF00FF
    assert np.array_equal(s1[bg], plain[bg]) and bg.sum() > 100000
    assert np.all(((s1 >> 16) & 255) <= ((plain >> 16) & 255)) and np.all((s1 & 255) <= (plain & 255))
    assert np.count_nonzero(s1 != plain) > 1000000


def test_reflection_extension_matches_oracle():
    """Config-5 extension (mirror bounces; no reference counterpart, the oracle is the definition): device == oracle."""
    v9, argb, bmin, bmax = load_obj3ds()
    prims = c1_spheres()
    prims.append((1, 0xff808080, [0, -0.45, 0, 0, 1, 0]))
    g = sa.GpuScene(0); o = orc.Scene()
    for s_ in (g, o):
        s_.set_triangles(v9, argb, bmin, bmax)
        s_.set_extra(prims)
    g.build((sa.MODE_REF_TREE, sa.MODE_BVH)); assert o.build_tree() == 0
    base, _ = o.render(make_frame(80, depth=3.0), threads=NCPU)
    for bounces, refl, kw in ((1, 0.5, dict()), (4, 0.3, dict(shadows=True)), (4, 1.0, dict(sub_pixel_res=2)), (2, 0.0, dict())):
        f = make_frame(80, depth=3.0, **kw)
        f.max_bounces, f.reflectivity = bounces, refl
        want, _ = o.render(f, threads=NCPU)
        for mode in (sa.MODE_REF_TREE, sa.MODE_BRUTE, sa.MODE_BVH):
            got, _ = g.render(as_sr(f, mode))
            assert np.array_equal(got, want), (bounces, refl, mode)
        if refl > 0 and not kw:
            assert not np.array_equal(want, base)
    v9, argb, bmin, bmax = unit_cube_scene(30000)
    g.set_triangles(v9, argb, bmin, bmax); o.set_triangles(v9, argb, bmin, bmax)
    g.set_extra([]); o.set_extra([])
    g.build((sa.MODE_BVH,)); assert o.build_tree() == 0
    f = make_frame(96, depth=1.5)
    f.max_bounces, f.reflectivity = 4, 0.4
    want, _ = o.render(f, threads=NCPU)
    got, _ = g.render(as_sr(f, sa.MODE_BVH))
    assert np.array_equal(got, want)
    # the wavefront form (k_primary -> k_bounce per level -> k_fold, the default on the own BVH without shadows) against the
    # one-kernel renderer, in one piece, in bands, as one pipeline and as interleaved strips
    single, st1 = g.render(as_sr(f, sa.MODE_BVH, single_kernel=True))
    assert np.array_equal(single, want)
    nosplit = as_sr(f, sa.MODE_BVH); nosplit.flags |= sa._lib.F_NO_SPLIT
    assert np.array_equal(g.render(nosplit)[0], want)
    # schedules of a level: one kernel instead of prepare / walk / finish (32); the walk with two stack levels per lane in LDS, the
    # rest in global memory (202), refilled at 60 busy lanes (160) and only when the wave is empty (100); with ray statistics
    ref_stats = None
    for hook in (32, 202, 160, 100):
        g.debug_set(sa._lib.DBG_KERNEL_SWITCH, hook)
        try:
            px, st_ = g.render(as_sr(f, sa.MODE_BVH))
        finally:
            g.debug_set(sa._lib.DBG_KERNEL_SWITCH, -1)
        assert np.array_equal(px, want), hook
        if ref_stats is None: ref_stats = st_[:8].copy()
        assert np.array_equal(st_[:8], ref_stats), hook        # the same rays, the same walks: counters do not depend on the schedule
    g.debug_set(sa._lib.DBG_BAND_SAMPLES, 3000)
    try:
        assert np.array_equal(g.render(as_sr(f, sa.MODE_BVH))[0], want)
        f2 = make_frame(96, depth=1.5, sub_pixel_res=2)
        f2.max_bounces, f2.reflectivity = 3, 0.6
        assert np.array_equal(g.render(as_sr(f2, sa.MODE_BVH))[0], o.render(f2, threads=NCPU)[0])
    finally:
        g.debug_set(sa._lib.DBG_BAND_SAMPLES, -1)
    full = want.reshape(96, 96)
    for k in range(3):
        fs = make_frame(96, depth=1.5, strips=(8, 3, k))
        fs.max_bounces, fs.reflectivity = 4, 0.4
        px, _ = g.render(as_sr(fs, sa.MODE_BVH))
        rows = [r for r in range(96) if (r // 8) % 3 == k]
        assert np.array_equal(px.reshape(len(rows), 96), full[rows]), k


def test_device_built_bvh_gives_identical_pixels():
    """the own BVH built on the GPU (LBVH) -- any conservative BVH must give the same pixels."""
    for n, res in ((300, 64), (20000, 128), (200000, 160)):
        v9, argb, bmin, bmax = unit_cube_scene(n)
        g = sa.GpuScene(0)
        g.set_triangles(v9, argb, bmin, bmax)
        g.build((sa.MODE_BVH,), on_device=False)      # host SAH build
        assert g.bvh_stats()[3] == 0
        f = make_frame(res, depth=1.5, shadows=True)
        a, _ = g.render(as_sr(f, sa.MODE_BVH))
        rnd = orc.Random(5)
        u = rnd.NextDoubles(6 * 20000).reshape(-1, 6)
        starts = 3.0 * u[:, :3] - 1.5
        dirs = (u[:, 3:] - 0.5) - starts
        ta = g.trace(sa.MODE_BVH, starts, dirs)
        g.build((sa.MODE_BVH,))                       # the default: the device LBVH build replaces it
        assert g.bvh_stats()[3] == (1 if n > 64 else 0)
        b, _ = g.render(as_sr(f, sa.MODE_BVH))
        assert np.array_equal(a, b), n
        tb = g.trace(sa.MODE_BVH, starts, dirs)
        for key in ("hit", "tri_index", "ray_frac", "pos", "normal", "color"):
            assert np.array_equal(ta[key], tb[key]), (n, key)
        c, _ = g.render(as_sr(f, sa.MODE_BVH, per_lane=True))
        assert np.array_equal(a, c)
    # the obj.3DS golden through the device-built tree
    g = sa.GpuScene(0)
    g.load_3ds(open(os.path.join(GOLDEN, "obj.3ds"), "rb").read())
    g.build((sa.MODE_BVH,), on_device=True)
    got, _ = g.render(as_sr(make_frame(100, shadows=True), sa.MODE_BVH))
    assert int(np.count_nonzero((got.reshape(100, 100) & 0xFFFFFF) != golden_rgb("shading_shadows", 100))) == 0


def test_static_shadows_match_oracle_and_goldens(obj_pair):
    """rayTraceShadowsStatic: the reference's two goldens (RendererTests.RaytraceStaticShadow), the oracle's
    lock-step fill order on other poses / sub-pixel settings / concurrencies, the cache surviving a frame, reset, and the
    combinations that have no defined fill order."""
    g, o = obj_pair
    for name, kw in (("shading_staticShadows", dict()), ("noShading_staticShadows", dict(shading=False))):
        for mode in (sa.MODE_REF_TREE, sa.MODE_BVH):
            g.reset_shadow_cache()
            got, _ = g.render(as_sr(make_frame(100, shadows=True, static_shadows=True, **kw), mode))
            assert int(np.count_nonzero((got.reshape(100, 100) & 0xFFFFFF) != golden_rgb(name, 100))) == 0, (name, mode)
    for kw in (dict(concurrency=1), dict(concurrency=3), dict(sub_pixel_res=2), dict(sub_pixel_res=3, focal_blur=True),
               dict(start_row=7, end_row=61), dict(shadow_samples=7)):
        f = make_frame(80, 72, shadows=True, static_shadows=True, **kw)
        o.reset_shadow_cache()
        want, _ = o.render(f)
        for mode in (sa.MODE_BVH, sa.MODE_REF_TREE, sa.MODE_BRUTE):
            g.reset_shadow_cache()
            got, _ = g.render(as_sr(f, mode))
            assert np.array_equal(got, want), (kw, mode)
    # the cache outlives the frame: B after A == the oracle's B after A, and differs from B on an empty cache
    fa = make_frame(64, shadows=True, static_shadows=True)
    fb = make_frame(64, shadows=True, static_shadows=True, yaw_deg=100.0)
    o.reset_shadow_cache(); g.reset_shadow_cache()
    o.render(fa); g.render(as_sr(fa, sa.MODE_BVH))
    want_b, _ = o.render(fb)
    got_b, _ = g.render(as_sr(fb, sa.MODE_BVH))
    assert np.array_equal(got_b, want_b)
    g.reset_shadow_cache()
    fresh_b, _ = g.render(as_sr(fb, sa.MODE_BVH))
    assert not np.array_equal(fresh_b, got_b)
    # a dynamic-shadow frame does not touch the cache, and static without shadows is plain shading (RendererTests.cs:420)
    dyn, _ = g.render(as_sr(make_frame(64, shadows=True), sa.MODE_BVH))
    want_dyn, _ = o.render(make_frame(64, shadows=True), threads=NCPU)
    assert np.array_equal(dyn, want_dyn)
    ns, _ = g.render(as_sr(make_frame(64, static_shadows=True), sa.MODE_BVH))
    assert np.array_equal(ns, o.render(make_frame(64), threads=NCPU)[0])
    for bad in (dict(strips=(4, 2, 0)),):
        with pytest.raises(sa.SoftrayError) as e:
            g.render(as_sr(make_frame(32, shadows=True, static_shadows=True, **bad), sa.MODE_BVH))
        assert e.value.code == sa._lib.SR_ERR_UNSUPPORTED
    fsk = as_sr(make_frame(32, shadows=True, static_shadows=True), sa.MODE_BVH, single_kernel=True)
    with pytest.raises(sa.SoftrayError):
        g.render(fsk)


def test_static_shadows_random_scene_and_extras():
    v9, argb, bmin, bmax = unit_cube_scene(5000)
    g = sa.GpuScene(0); o = orc.Scene()
    for s in (g, o):
        s.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,)); assert o.build_tree() == 0
    prims = c1_spheres(6)
    g.set_extra(prims); o.set_extra(prims)
    f = make_frame(120, 96, depth=1.5, shadows=True, static_shadows=True)
    want, _ = o.render(f)
    got, _ = g.render(as_sr(f, sa.MODE_BVH))
    assert np.array_equal(got, want)
