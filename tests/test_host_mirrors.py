"""The host-side mirrors of the reference API above the C ABI: the C++ mirror (softray_amd/host/Engine3D.hpp, driven by
tests/cpp/renderer_tests.cpp) and the Python mirror (softray_amd/renderer.py).  The scenarios are the reference's own
RendererTests (Engine3D-Tests/Raytrace/RendererTests.cs:140-213)."""
import math
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT
from test_oracle import GOLDENS, golden_rgb


def build_cpp_tests(tmp_path):
    exe = str(tmp_path / "renderer_tests")
    lib_dir = os.path.join(ROOT, "softray_amd")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-o", exe, os.path.join(ROOT, "tests", "cpp", "renderer_tests.cpp"),
                           "-L" + lib_dir, "-lsoftray_hip", "-Wl,-rpath," + lib_dir])
    return exe


def test_cpp_mirror_builds_and_refuses_to_run_without_gpu(tmp_path):
    import torch
    exe = build_cpp_tests(tmp_path)
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu-marked test")
    r = subprocess.run([exe, GOLDEN], capture_output=True, text=True)
    assert r.returncode == 3 and "no HIP device" in r.stderr     # loud failure, no fallback


@pytest.mark.gpu
def test_cpp_mirror_reference_scenarios(tmp_path):
    """All 22 goldens x the three TraversalCounters modes, two instances, surface passes, error behaviour -- through Engine3D.hpp."""
    exe = build_cpp_tests(tmp_path)
    r = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=900)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "ALL OK" in r.stdout
    assert r.stdout.count("diff=0") >= 3 * 24


@pytest.mark.gpu
def test_cpp_mirror_c2_full_frame_in_its_default_mode(tmp_path):
    """Config C2 (obj.3DS, 1024^2, without and with the 100-sample shadows) through the C++ Renderer mirror as a drop-in user
    constructs it; every 16-row strip against the oracle's full-frame fixture."""
    import json
    import zlib
    exe = build_cpp_tests(tmp_path)
    prefix = str(tmp_path / "cpp")
    r = subprocess.run([exe, GOLDEN, "--dump-c2", prefix], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    for name in ("c2", "c2_shadows"):
        doc = json.load(open(os.path.join(GOLDEN, "frames", name + ".json")))
        px = np.fromfile(prefix + "_" + name + ".bin", dtype="<u4").reshape(1024, 1024)
        assert len(doc["strips"]) == 64
        for s, crc in doc["strips"].items():
            s = int(s)
            assert zlib.crc32(px[16 * s:16 * s + 16].tobytes()) & 0xFFFFFFFF == crc, (name, s)


# ---- the Python mirror, written like RendererTests.cs ----
pixels = np.zeros(400 * 400, dtype=np.int32)                      # RendererTests.cs:58


def RendererSetup(renderer, modelFileName, pitchDegrees, yawDegrees, rollDegrees, objectDepth, resolution):
    from softray_amd.renderer import Instance, Vector
    renderer.BackgroundColor = 0xff00ff
    renderer.SetRenderingSurface(resolution, resolution, pixels)
    with open(modelFileName, "rb") as stream:
        renderer.Load3dsModelFromStream(stream)
    renderer.Instances.append(Instance(renderer.Model, Position=Vector(0.0, 0.0, objectDepth), Yaw=yawDegrees / 180.0 * math.pi,
                                       Pitch=pitchDegrees / 180.0 * math.pi, Roll=rollDegrees / 180.0 * math.pi))


def RaytraceScenario(shading=True, focalBlur=False, shadows=False, subPixelRes=1, resolution=100, extraGeometry=None, objectDepth=1.0,
                     traversalCounters=None, staticShadows=False, firstInstance=None):
    from softray_amd.renderer import Instance, Renderer, TraversalCounters, Vector
    with Renderer(0, traversalCounters=TraversalCounters.Auto if traversalCounters is None else traversalCounters) as renderer:
        RendererSetup(renderer, os.path.join(GOLDEN, "obj.3ds"), -22.0, 135.0, 0.0, objectDepth, resolution)
        renderer.rayTrace = True
        renderer.rayTraceSubdivision = True
        renderer.rayTraceShading = shading
        renderer.rayTraceFocalBlur = focalBlur
        renderer.rayTraceFocalDepth = objectDepth + 0.5
        renderer.rayTraceSubPixelRes = subPixelRes
        renderer.rayTraceShadows = shadows
        renderer.rayTraceShadowsStatic = staticShadows
        if firstInstance is not None:                               # Renderer.cs:746-760: every instance is raytraced over the whole surface
            yaw, depth = firstInstance
            renderer.Instances.insert(0, Instance(renderer.Model, Position=Vector(0.0, 0.0, depth), Yaw=yaw / 180.0 * math.pi, Pitch=0.3, Roll=0.1))
        if extraGeometry is not None:
            renderer.ExtraGeometryToRaytrace = extraGeometry
        testName = (("shading" if shading else "noShading") + (("_staticShadows" if staticShadows else "_shadows") if shadows else "") + ("_focalBlur" if focalBlur else "") +
                    ("x%d" % subPixelRes if focalBlur else ("_%dxAA" % subPixelRes if subPixelRes > 1 else "")))
        renderer.Render()
        got = pixels[: resolution * resolution].view(np.uint32).reshape(resolution, resolution) & 0xFFFFFF
        return testName, got, renderer


@pytest.mark.gpu
def test_RaytraceAntialised():
    for n in (2, 4, 8):
        name, got, _ = RaytraceScenario(subPixelRes=n)
        assert np.array_equal(got, golden_rgb(name, 100)), name


@pytest.mark.gpu
def test_RaytraceDynamicShadow():
    from softray_amd.renderer import InvalidOperationException, TraversalCounters
    # the default (Auto): obj.3DS is a model of the reference's own size -> the literal tree for the primary rays, the reference's counters
    name, got, r = RaytraceScenario(shadows=True)
    assert np.array_equal(got, golden_rgb(name, 100))
    assert r.TraversalCountersAvailable and r.NumRaysFired == 10000 and r.NumNodeVisits > 0 and r.NumGeometryTests > 0 and r.NumLeafNodeVisits > 0
    auto = (r.NumGeometryTests, r.NumNodeVisits, r.NumLeafNodeVisits)
    name, got, r = RaytraceScenario(shadows=True, traversalCounters=TraversalCounters.Literal)
    assert np.array_equal(got, golden_rgb(name, 100))
    assert (r.NumGeometryTests, r.NumNodeVisits, r.NumLeafNodeVisits) == auto
    # Off: the own BVH; the three counters are not produced and say so -- loudly -- instead of reading zero
    name, got, r = RaytraceScenario(shadows=True, traversalCounters=TraversalCounters.Off)
    assert np.array_equal(got, golden_rgb(name, 100))
    assert r.NumRaysFired == 10000 and not r.TraversalCountersAvailable
    for counter in ("NumGeometryTests", "NumNodeVisits", "NumLeafNodeVisits"):
        with pytest.raises(InvalidOperationException):
            getattr(r, counter)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Auto", "Literal", "Off"])
def test_every_golden_through_the_python_host(mode):
    """All 22 reference goldens through Renderer.Render() in every TraversalCounters mode (RendererTests.cs:381-430, 511-544)."""
    from softray_amd.renderer import TraversalCounters
    tc = getattr(TraversalCounters, mode)
    for gname, res, kw in GOLDENS:
        name, got, _ = RaytraceScenario(shading=kw.get("shading", True), focalBlur=kw.get("focal_blur", False), shadows=kw.get("shadows", False),
                                        subPixelRes=kw.get("sub_pixel_res", 1), resolution=res, traversalCounters=tc)
        assert name == gname
        assert np.array_equal(got, golden_rgb(name, res)), (name, res, mode)
    for shading in (True, False):                                   # RaytraceStaticShadow (RendererTests.cs:167-175)
        name, got, _ = RaytraceScenario(shading=shading, shadows=True, staticShadows=True, traversalCounters=tc)
        assert np.array_equal(got, golden_rgb(name, 100)), (name, mode)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["Auto", "Off"])
def test_two_instances_the_last_one_owns_every_pixel(mode):
    """foreach (var instance in Instances) RaytraceGeometry(instance) (Renderer.cs:746-760): each instance is raytraced over the WHOLE
    surface, background included, so a frame with two instances is the frame of the last one -- here the goldens' pose, so the
    reference's own images pin it."""
    from softray_amd.renderer import TraversalCounters
    tc = getattr(TraversalCounters, mode)
    for kw in (dict(), dict(shadows=True), dict(shading=False, subPixelRes=4)):
        name, got, r = RaytraceScenario(firstInstance=(10.0, 2.0), traversalCounters=tc, **kw)
        assert len(r.Instances) == 2
        assert np.array_equal(got, golden_rgb(name, 100)), (name, mode)
        assert r.NumRaysFired == 10000 * kw.get("subPixelRes", 1) ** 2          # the last instance's frame (counters restart per RaytraceBlock, :1695-1699)
    # ... and the order matters: with the goldens' pose FIRST the frame is the other instance's (not vacuous)
    from softray_amd.renderer import Renderer
    with Renderer(0, traversalCounters=tc) as renderer:
        RendererSetup(renderer, os.path.join(GOLDEN, "obj.3ds"), -22.0, 135.0, 0.0, 1.0, 100)
        renderer.rayTrace = True
        renderer.rayTraceFocalBlur = False
        renderer.Render()
        one = pixels[:10000].copy()
        from softray_amd.renderer import Instance, Vector
        renderer.Instances.append(Instance(renderer.Model, Position=Vector(0.0, 0.0, 2.0), Yaw=10.0 / 180.0 * math.pi, Pitch=0.3, Roll=0.1))
        renderer.Render()
        two = pixels[:10000].copy()
        renderer.Instances.pop(0)
        renderer.Render()
        assert np.array_equal(pixels[:10000], two) and not np.array_equal(one, two)
        assert np.array_equal(one.view(np.uint32).reshape(100, 100) & 0xFFFFFF, golden_rgb("shading", 100))


@pytest.mark.gpu
def test_RaytraceShadowAndFocalBlur_and_AntiAlias():
    name, got, _ = RaytraceScenario(focalBlur=True, shadows=True, subPixelRes=4, resolution=50)
    assert np.array_equal(got, golden_rgb(name, 50))
    name, got, _ = RaytraceScenario(shadows=True, subPixelRes=4, resolution=50)
    assert np.array_equal(got, golden_rgb(name, 50))


@pytest.mark.gpu
def test_renderer_mirror_error_behaviour_and_extra_geometry():
    from softray_amd import SoftrayError
    from softray_amd.renderer import Color, GeometryCollection, Renderer, Sphere, Vector
    with Renderer() as r:                                           # no model: Render() returns silently (Renderer.cs:736-739)
        r.rayTrace = True
        buf = np.full(16, 7, dtype=np.int32)
        r.SetRenderingSurface(4, 4, buf)
        r.Render()
        assert np.all(buf == 7)
        with pytest.raises(NotImplementedError):
            r.rayTrace = False
            r.Render()
    with Renderer() as r:
        with pytest.raises(SoftrayError):                           # FormatException
            import io
            r.Load3dsModelFromStream(io.BytesIO(b"not a 3ds file at all........"))
        assert r.HasModelLoadFailed()
    geometryList = GeometryCollection()                              # like PathTracePrimitivesTest's scene (RendererTests.cs:247-255), no path tracing
    geometryList.Add(Sphere(Vector(-0.5, 0, -0.5), 0.5, Color=Color.Red))
    geometryList.Add(Sphere(Vector(+0.5, 0, +0.5), 0.5, Color=Color.Green))
    _, a, _ = RaytraceScenario(extraGeometry=geometryList, objectDepth=3.0, shadows=True, resolution=64)
    assert (a == 0xff00ff).sum() > 0 and len(np.unique(a)) > 50
