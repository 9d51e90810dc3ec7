"""CPU-side checks of the drop-in boundary: libsoftray_hip.so loads, exports every symbol that
include/softray.h declares, refuses compute without a device (no CPU fallback), and its HOST logic
(tree builder, 3DS loader, matrices, System.Random) agrees with the oracle."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import softray_amd as sa
from helpers import ROOT, load_obj3ds, make_frame, orc, random_triangles, unit_cube_scene

HEADER = os.path.join(ROOT, "include", "softray.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sr_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_exported():
    L = sa._lib.lib()
    names = declared_symbols()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), "libsoftray_hip.so does not export %s" % n
    assert sorted(sa._lib.SYMBOLS) == names
    assert L.sr_abi_version() == 5              # 24 ray statistics (SR_STATS_COUNT); 5: sr_trace_rays_device, the sr_rccl_* gather


def test_frame_layout_matches_oracle_frame():
    assert C.sizeof(sa.Frame) == C.sizeof(orc.Frame)
    for (n1, t1), (n2, t2) in zip(sa.Frame._fields_, orc.Frame._fields_):
        assert n1 == n2 and C.sizeof(t1) == C.sizeof(t2)
        assert getattr(sa.Frame, n1).offset == getattr(orc.Frame, n2).offset


def test_layout_contract_of_header_ctypes_and_csharp():
    """sr_frame / sr_prim offsets: the header's static asserts (compiled into the library), the ctypes mirror and the offsets
    written next to the C# struct fields must agree."""
    header = open(HEADER).read()
    assert "sizeof(sr_frame) == %d" % C.sizeof(sa.Frame) in header and "sizeof(sr_prim) == %d" % C.sizeof(sa._lib.Prim) in header
    cs = open(os.path.join(ROOT, "bindings", "csharp", "GpuRenderer.cs")).read()
    for name, _ in sa.Frame._fields_:
        off = getattr(sa.Frame, name).offset
        m = re.search(r"/\*\s*@%d\s*\*/[^;]*\b%s\b" % (off, name), cs)
        assert m, "GpuRenderer.cs: field %s is not annotated with offset @%d" % (name, off)
        hm = re.search(r"offsetof\(sr_frame, %s\) == (\d+)" % name, header)
        if hm:
            assert int(hm.group(1)) == off


def test_library_never_reads_the_environment():
    """A drop-in must not change its schedule with the host's environment: every hook goes through sr_debug_set."""
    pkg = os.path.join(ROOT, "softray_amd", "csrc")
    for f in os.listdir(pkg):
        if not os.path.isfile(os.path.join(pkg, f)) or f.endswith((".so", ".o")):
            continue
        text = open(os.path.join(pkg, f), errors="ignore").read()
        assert "getenv" not in text, "%s reads the environment" % f
    s = sa.GpuScene(device=-1)
    s.debug_set(sa._lib.DBG_ROUND_CAP0, 5)
    with pytest.raises(sa.SoftrayError):
        s.debug_set(99, 1)


def test_host_bvh_build_does_not_depend_on_the_thread_count():
    """The binned-SAH build spreads its passes and subtrees over the host's cores; tree, node numbering and triangle order must be
    those of the single-threaded build (order-independent reductions, stable partition, subtrees appended in range order)."""
    v9, argb = sa.make_random_triangles(200000, 4711, space=0.95, extent=0.05, origin=-0.5, opaque=True)
    digests, stats = set(), set()
    for threads in (1, 2, 5, 16):
        s = sa.GpuScene(device=-1)                   # host-only scene: the build is host work
        s.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
        s.debug_set(sa._lib.DBG_BUILD_THREADS, threads)
        s.build((sa.MODE_BVH,))
        digests.add(s.bvh_digest())
        stats.add(s.bvh_stats())
    assert len(digests) == 1 and len(stats) == 1, (digests, stats)
    depth, nodes, tris, on_device = next(iter(stats))
    assert tris == 200000 and on_device == 0 and 200000 / 7 <= nodes < 200000 and 15 <= depth <= 60


def test_wide_tree_is_the_binary_tree_collapsed():
    """The four-wide tree of the packet walks (sr_host.cpp collapse_bvh4) holds exactly the binary tree's leaves: every triangle in
    one leaf, as many leaves as the binary tree has (inner nodes + 1), every node but the root linked once, between a third and all of the depth."""
    for n, seed in ((1, 1), (5, 2), (9, 3), (300, 4), (50000, 5)):
        v9, argb = sa.make_random_triangles(n, seed, space=0.9, extent=0.1, origin=-0.5, opaque=True)
        s = sa.GpuScene(device=-1)
        s.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
        s.build((sa.MODE_BVH,))
        depth2, nodes2, tris, _ = s.bvh_stats()
        depth4, nodes4, slots, leaves, leaf_tris = s.wide_tree_stats()
        assert leaf_tris == tris == n
        leaves2 = nodes2 + 1 if n > 4 else 1            # (a scene that fits one leaf: a root record with one leaf child)
        assert leaves == leaves2, (n, leaves, leaves2)
        assert slots == leaves + nodes4 - 1             # every node but the root sits in exactly one slot
        assert nodes4 <= nodes2 and depth4 <= depth2 and 3 * depth4 >= depth2 - 2
        if n >= 300:
            assert slots / nodes4 > 2.8                  # the greedy collapse fills the nodes (4 wherever a node has enough descendants)


def test_no_cpu_fallback():
    s = sa.GpuScene(device=-1)                       # host-only scene
    v9, argb, bmin, bmax = load_obj3ds()
    s.set_triangles(v9, argb, bmin, bmax)
    s.build((sa.MODE_REF_TREE, sa.MODE_BVH))
    with pytest.raises(sa.SoftrayError) as e:
        s.render(make_frame(8))
    assert e.value.code == sa._lib.SR_ERR_NO_DEVICE
    with pytest.raises(sa.SoftrayError) as e:
        s.trace(sa.MODE_BRUTE, [[0, 0, 1]], [[0, 0, -1]])
    assert e.value.code == sa._lib.SR_ERR_NO_DEVICE


def test_product_never_imports_oracle():
    """The product path must not reference oracle/ in any form."""
    pkg = os.path.join(ROOT, "softray_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "oracle" not in text.lower(), "%s mentions the oracle" % f


@pytest.mark.parametrize("n,max_depth,max_geom,expected", [
    (10, 5, 3, (4, 9, 5, 4)), (5, 3, 1, (2, 3, 2, 1)), (8, 3, 1, (3, 5, 3, 2)),
    (4, 100, 1, (2, 3, 2, 1)), (1000, 10, 5, (10, 885, 443, 442)),
])
def test_product_tree_builder_kats(n, max_depth, max_geom, expected):
    # the same seeded KATs as the reference (SpatialSubdivisionTests.cs:59-137), on the product's builder
    v9, argb, _ = random_triangles(n, seed=12345)
    s = sa.GpuScene(device=-1)
    s.set_triangles(v9, argb, [0, 0, 0], [110, 110, 110])
    s.build((sa.MODE_REF_TREE,), max_depth, max_geom)
    assert s.tree_stats() == expected


def test_product_tree_errors():
    s = sa.GpuScene(device=-1)
    v = np.zeros((1, 3, 3)); v[0, 1] = [2.0, 0, 0]
    s.set_triangles(v, np.zeros(1, dtype=np.uint32), [0, 0, 0], [1, 1, 1])
    with pytest.raises(sa.SoftrayError) as e:      # ArgumentOutOfRangeException, SpatialSubdivision.cs:287-295
        s.build((sa.MODE_REF_TREE,))
    assert e.value.code == sa._lib.SR_ERR_OUT_OF_RANGE
    s2 = sa.GpuScene(device=-1)
    with pytest.raises(sa.SoftrayError) as e:
        s2.build((sa.MODE_REF_TREE,))
    assert e.value.code == sa._lib.SR_ERR_NO_MODEL
    with pytest.raises(sa.SoftrayError) as e:      # FormatException "Not a proper 3DS file."
        s2.load_3ds(b"\x00" * 64)
    assert e.value.code == sa._lib.SR_ERR_FORMAT


def test_product_tree_matches_oracle_on_big_scene():
    v9, argb, bmin, bmax = unit_cube_scene(20000)
    s = sa.GpuScene(device=-1)
    s.set_triangles(v9, argb, bmin, bmax)
    s.build((sa.MODE_REF_TREE,))
    o = orc.Scene()
    o.set_triangles(v9, argb, bmin, bmax)
    assert o.build_tree() == 0
    assert s.tree_stats() == o.tree_stats()


def test_product_host_helpers_match_oracle():
    s = sa.GpuScene(device=-1)
    for name in ("obj.3ds", "obj2.3DS"):
        s.load_3ds(open(os.path.join(ROOT, "tests", "golden", name), "rb").read())
        for a, b in zip(s.get_triangles(), load_obj3ds(name)):
            assert np.array_equal(a, b)
    for pose in ([0, 0, 1.0], 135 / 180 * np.pi, -22 / 180 * np.pi, 0.0), ([0.3, -0.2, 2.5], 0.7, 0.4, -1.1):
        t1, i1 = sa.instance_matrices(*pose)
        t2, i2 = orc.instance_matrices(*pose)
        assert np.array_equal(t1, t2) and np.array_equal(i1, i2)
    assert sa.default_fov_depth() == orc.default_fov_depth()
    for seed in (1234567890, 1, -5, 2147483647):
        assert np.array_equal(sa.area_light_offsets(seed, 100), orc.area_light_offsets(seed, 100))
    # the ray generator of the micro-benchmarks (sr_net_random_doubles) is the same System.Random stream, with and without a skip
    for seed in (12345, 1, -7):
        want = orc.Random(seed).NextDoubles(300)
        assert np.array_equal(sa.net_random_doubles(seed, 300), want)
        assert np.array_equal(sa.net_random_doubles(seed, 100, skip=200), want[200:])


def test_product_scene_generator_matches_reference_generator():
    a = sa.make_random_triangles(1000, seed=12345)
    b = random_triangles(1000, seed=12345)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    a = sa.unit_cube_scene(5000)
    b = unit_cube_scene(5000)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
