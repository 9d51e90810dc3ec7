"""The host-side mirrors of the reference API above the C ABI: the C++ mirror (softray_amd/host/Engine3D.hpp, driven by
tests/cpp/renderer_tests.cpp) and the Python mirror (softray_amd/renderer.py).  The scenarios are the reference's own
RendererTests (Engine3D-Tests/Raytrace/RendererTests.cs:140-213)."""
import math
import os
import subprocess

import numpy as np
import pytest

from helpers import GOLDEN, ROOT
from test_oracle import golden_rgb


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
    exe = build_cpp_tests(tmp_path)
    r = subprocess.run([exe, GOLDEN], capture_output=True, text=True, timeout=600)
    print(r.stdout, r.stderr)
    assert r.returncode == 0 and "ALL OK" in r.stdout


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
                     literalCounters=False):
    from softray_amd.renderer import Renderer
    with Renderer() as renderer:
        RendererSetup(renderer, os.path.join(GOLDEN, "obj.3ds"), -22.0, 135.0, 0.0, objectDepth, resolution)
        renderer.rayTrace = True
        renderer.rayTraceSubdivision = True
        renderer.rayTraceShading = shading
        renderer.rayTraceFocalBlur = focalBlur
        renderer.rayTraceFocalDepth = objectDepth + 0.5
        renderer.rayTraceSubPixelRes = subPixelRes
        renderer.rayTraceShadows = shadows
        renderer.gpuLiteralTraversalCounters = literalCounters
        if extraGeometry is not None:
            renderer.ExtraGeometryToRaytrace = extraGeometry
        testName = (("shading" if shading else "noShading") + ("_shadows" if shadows else "") + ("_focalBlur" if focalBlur else "") +
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
    name, got, r = RaytraceScenario(shadows=True)
    assert np.array_equal(got, golden_rgb(name, 100))
    assert r.NumRaysFired == 10000 and r.NumNodeVisits == 0        # default: no traversal counters asked for (fast path)
    name, got, r = RaytraceScenario(shadows=True, literalCounters=True)    # the literal reference-tree traversal with the reference's counters
    assert np.array_equal(got, golden_rgb(name, 100))
    assert r.NumRaysFired == 10000 and r.NumNodeVisits > 0 and r.NumGeometryTests > 0


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
