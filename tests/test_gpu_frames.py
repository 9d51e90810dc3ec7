"""Whole frames of BASELINE.json's configurations against the CPU oracle, strip by strip.

tests/golden/frames/<config>.json holds the CRC-32 of every 16-row strip of the frame the oracle renders (made by
scripts/make_frame_fixtures.py in the build container; the header of each file records the oracle commit, the command and the
coverage).  Here the HIP path renders the same frame through the C ABI -- and, for C2, through the three hosts' default modes -- and
every strip the fixture holds must have the same CRC: bit-exact ARGB over 100 % of the fixture's rows, not a sample of them."""
import json
import os
import zlib

import numpy as np
import pytest

import softray_amd as sa
from helpers import GOLDEN, make_frame, unit_cube_scene

FRAMES = os.path.join(GOLDEN, "frames")
STRIP = 16


def fixture(name):
    path = os.path.join(FRAMES, name + ".json")
    if not os.path.exists(path):
        pytest.skip("no fixture " + path)
    doc = json.load(open(path))
    assert doc["strip_rows"] == STRIP and doc["width"] == doc["height"]
    return doc


def check_strips(doc, pixels, what, need_all=False):
    res = doc["width"]
    px = np.ascontiguousarray(pixels).view(np.uint32).reshape(res, res)
    strips = {int(k): v for k, v in doc["strips"].items()}
    if need_all:
        assert len(strips) == doc["strips_total"], "the fixture of %s must cover the whole frame" % doc["config"]
    bad = [s for s, crc in strips.items() if (zlib.crc32(px[s * STRIP:(s + 1) * STRIP].astype("<u4").tobytes()) & 0xFFFFFFFF) != crc]
    assert not bad, "%s: %d of %d strips differ from the oracle's frame (first: %s)" % (what, len(bad), len(strips), bad[:8])
    return len(strips)


def as_sr(frame, mode):
    f = sa.Frame.from_buffer_copy(bytes(frame))
    f.trace_mode = mode
    return f


def fixture_frame(doc, **kw):
    fr = doc["frame"]
    f = make_frame(doc["width"], depth=fr["depth"], shadows=bool(fr.get("shadows")), **kw)
    f.max_bounces, f.reflectivity = fr["max_bounces"], fr["reflectivity"]
    return f


def test_fixture_files_are_well_formed():
    """(CPU) every committed fixture names its oracle commit, its command and its coverage; C2 / C3 / C4 / C5 (four bounces) cover the whole frame."""
    names = sorted(n[:-5] for n in os.listdir(FRAMES) if n.endswith(".json")) if os.path.isdir(FRAMES) else []
    assert "c2" in names and "c3" in names
    for n in names:
        doc = json.load(open(os.path.join(FRAMES, n + ".json")))
        assert doc["config"] == n and doc["oracle_commit"] and doc["command"].startswith("python scripts/make_frame_fixtures.py")
        assert 0 < len(doc["strips"]) <= doc["strips_total"] == doc["height"] // STRIP
        assert all(0 <= int(k) < doc["strips_total"] and 0 <= v <= 0xFFFFFFFF for k, v in doc["strips"].items())
    for n in ("c2", "c2_shadows", "c3", "c4", "c5"):
        if n in names:
            doc = json.load(open(os.path.join(FRAMES, n + ".json")))
            assert len(doc["strips"]) == doc["strips_total"], n


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c2_shadows"])
def test_c2_full_frame_every_structure(name):
    """Config C2 (obj.3DS, 1024^2) -- every strip of the oracle's frame, on the literal tree, the device-built and the host-built BVH."""
    doc = fixture(name)
    data = open(os.path.join(GOLDEN, "obj.3ds"), "rb").read()
    for modes, on_device, mode in (((sa.MODE_REF_TREE, sa.MODE_BVH), None, sa.MODE_REF_TREE), ((sa.MODE_BVH,), True, sa.MODE_BVH),
                                   ((sa.MODE_BVH,), False, sa.MODE_BVH)):
        g = sa.GpuScene(0)
        g.load_3ds(data)
        g.build(modes, on_device=on_device)
        px, _ = g.render(as_sr(fixture_frame(doc), mode))
        check_strips(doc, px, "%s mode %d on_device %s" % (name, mode, on_device), need_all=True)
        g.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2", "c2_shadows"])
def test_c2_full_frame_through_the_python_host_default_modes(name):
    """The same frame through the Renderer mirror, in each TraversalCounters mode (Auto is what a drop-in user gets)."""
    from softray_amd.renderer import Instance, Renderer, TraversalCounters, Vector
    doc = fixture(name)
    res = doc["width"]
    pixels = np.zeros(res * res, dtype=np.int32)
    for tc in (TraversalCounters.Auto, TraversalCounters.Literal, TraversalCounters.Off):
        with Renderer(0, traversalCounters=tc) as r:
            r.BackgroundColor = 0xff00ff
            r.SetRenderingSurface(res, res, pixels)
            with open(os.path.join(GOLDEN, "obj.3ds"), "rb") as stream:
                r.Load3dsModelFromStream(stream)
            r.Instances.append(Instance(r.Model, Position=Vector(0.0, 0.0, 1.0), Yaw=135.0 / 180.0 * np.pi, Pitch=-22.0 / 180.0 * np.pi, Roll=0.0))
            r.rayTrace = True
            r.rayTraceFocalBlur = False
            r.rayTraceShadows = bool(doc["frame"].get("shadows"))
            pixels[:] = 0
            r.Render()
            check_strips(doc, pixels, "%s Renderer mirror, TraversalCounters %d" % (name, tc), need_all=True)
            assert r.NumRaysFired == res * res
            assert r.TraversalCountersAvailable == (tc != TraversalCounters.Off)


@pytest.fixture(scope="module")
def million():
    v9, argb, bmin, bmax = unit_cube_scene(1_000_000)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, bmin, bmax)
    g.build((sa.MODE_BVH,))
    yield g
    g.close()


@pytest.mark.gpu
def test_c3_full_frame(million):
    """Config C3 (1 M triangles, 2048^2, shading + 100-sample shadows): every strip of the oracle's frame."""
    doc = fixture("c3")
    px, st = million.render(as_sr(fixture_frame(doc), sa.MODE_BVH))
    assert st[0] == 2048 * 2048
    n = check_strips(doc, px, "c3", need_all=True)
    print("c3: %d of %d strips equal the oracle's" % (n, doc["strips_total"]))


@pytest.mark.gpu
def test_c4_frame_and_its_8_way_split(million):
    """Config C4 (the same scene at 4096^2, 8 ranks x interleaved 16-row strips): every strip the fixture holds, rendered whole and as
    the union of the eight ranks' strips (Renderer.cs:1655-1680)."""
    doc = fixture("c4")
    f = fixture_frame(doc)
    px, _ = million.render(as_sr(f, sa.MODE_BVH))
    n = check_strips(doc, px, "c4 whole frame", need_all=True)
    union = np.zeros((4096, 4096), dtype=np.uint32)
    for k in range(8):
        fs = fixture_frame(doc, strips=(16, 8, k))
        part, _ = million.render(as_sr(fs, sa.MODE_BVH))
        rows = [r for r in range(4096) if (r // 16) % 8 == k]
        union[rows] = part.reshape(len(rows), 4096)
    check_strips(doc, union, "c4 union of 8 strip sets")
    assert np.array_equal(union.reshape(-1), px)
    print("c4: %d of %d strips (%s) equal the oracle's" % (n, doc["strips_total"], doc.get("coverage")))


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c5", "c5_shadows"])
def test_c5_frame_strips(name):
    """Config C5 (10 M triangles, 4096^2; 4 mirror bounces -- the build-defined extension, parity unpinned by the reference -- and the
    same scene with the reference's soft shadows): the strips the fixture holds (coverage in its header)."""
    doc = fixture(name)
    v9, argb = sa.make_random_triangles(10_000_000, 12345, space=0.98, extent=0.02, origin=-0.5, opaque=True)
    g = sa.GpuScene(0)
    g.set_triangles(v9, argb, np.array([-0.5] * 3), np.array([0.5] * 3))
    g.build((sa.MODE_BVH,))
    px, _ = g.render(as_sr(fixture_frame(doc), sa.MODE_BVH))
    n = check_strips(doc, px, name, need_all=(name == "c5"))            # (the four-bounce frame is complete; the shadowed one: what the round had time for)
    print("%s: %d of %d strips (%s) equal the oracle's" % (name, n, doc["strips_total"], doc.get("coverage")))
    g.close()
