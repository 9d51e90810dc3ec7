"""Timing of the BASELINE.json configs that are not the headline (informational; printed as JSON lines)."""
import json, math, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import softray_amd as sa
from softray_amd.renderer import Renderer, Instance, Vector, Sphere, Color, GeometryCollection, TraversalCounters

def timed(r, n=5):
    r.Render()                                  # warm-up (uploads, builds)
    t = time.perf_counter()
    for _ in range(n):
        r.Render()
    return (time.perf_counter() - t) / n

def setup(res, depth, model_path=None, model=None, mode=None, counters=TraversalCounters.Auto):
    r = Renderer(0, traversalCounters=counters)
    r.BackgroundColor = 0xff00ff
    px = np.zeros(res * res, dtype=np.int32)
    r.SetRenderingSurface(res, res, px)
    if model_path:
        with open(model_path, "rb") as f:
            r.Load3dsModelFromStream(f)
    else:
        r.Model = model
    r.Instances.append(Instance(r.Model, Position=Vector(0, 0, depth), Yaw=135 / 180 * math.pi, Pitch=-22 / 180 * math.pi))
    r.rayTrace = True
    r.gpuTraceMode = mode
    return r, px

obj = os.path.join(ROOT, "tests", "golden", "obj.3ds")
out = []
# C1: 16 spheres + obj.3DS, 256^2, shading, primary only
r, px = setup(256, 3.0, model_path=obj)
g = GeometryCollection()
pal = [Color.Red, Color.Green, Color.Blue, Color.Yellow, Color.Orange, Color.Brown, Color.Pink, Color.Cyan, Color.White, Color.Grey]
import ctypes
L = sa._lib.lib()
# same System.Random stream as tests/helpers.c1_spheres, through the product's generator
v, _ = sa.make_random_triangles(8, seed=12345, space=1.0, extent=0.0)   # not the sphere stream; spheres below use numpy-free constants
rng = np.random.RandomState(1)
for i in range(16):
    c = rng.rand(3) - 0.5
    g.Add(Sphere(Vector(*c), 0.05 + 0.1 * rng.rand(), Color=pal[i % len(pal)]))
r.ExtraGeometryToRaytrace = g
dt = timed(r, 20)
out.append({"config": "C1: 16 spheres + obj.3DS, 256x256, shading, default mode of the mirror (API round trip incl. D2H)", "ms": dt * 1e3, "Mrays_s": 256 * 256 / dt / 1e6})
# C2: obj.3DS 1024^2 primary + shading, reference tree via the Renderer API
for name, counters in (("TraversalCounters.Auto = the default: literal reference tree + counters for this model size, shadow rays on the BVH", TraversalCounters.Auto),
                       ("TraversalCounters.Off: own BVH, no traversal counters", TraversalCounters.Off)):
    r, px = setup(1024, 1.0, model_path=obj, counters=counters)
    dt = timed(r, 20)
    out.append({"config": "C2: obj.3DS (152 tris), 1024x1024, shading, %s (API round trip incl. D2H)" % name, "ms": dt * 1e3, "Mrays_s": 1024 * 1024 / dt / 1e6,
                "NumRaysFired": r.NumRaysFired})
    r.rayTraceShadows = True
    dt = timed(r, 5)
    out.append({"config": "C2 + 100-sample soft shadows, %s" % name, "ms": dt * 1e3, "Mrays_s": 1024 * 1024 / dt / 1e6})
# the Auto threshold: a 1 999-triangle soup on the literal tree (Auto) against the own BVH (Off), 1024^2
from softray_amd.renderer import Model
v9s, argbs, bmins, bmaxs = sa.unit_cube_scene(1999)
for name, counters in (("Auto (literal tree + counters)", TraversalCounters.Auto), ("Off (own BVH)", TraversalCounters.Off)):
    r, px = setup(1024, 1.5, model=Model.FromTriangles(v9s, argbs, bmins, bmaxs), counters=counters)
    dt = timed(r, 10)
    out.append({"config": "1 999 random triangles, 1024x1024, shading, %s" % name, "ms": dt * 1e3, "Mrays_s": 1024 * 1024 / dt / 1e6})
    r.rayTraceShadows = True
    dt = timed(r, 5)
    out.append({"config": "1 999 random triangles + 100-sample soft shadows, %s" % name, "ms": dt * 1e3, "Mrays_s": 1024 * 1024 / dt / 1e6})
# C3: 1M random triangles + BVH, 2048^2, shading + shadows
from softray_amd.renderer import Model
v9, argb, bmin, bmax = sa.unit_cube_scene(1000000)
r, px = setup(2048, 1.5, model=Model.FromTriangles(v9, argb, bmin, bmax))       # default mode: own BVH at this size
r.rayTraceShadows = True
dt = timed(r, 5)
out.append({"config": "C3: 1M random triangles + BVH, 2048x2048, shading + 100-sample shadows (API round trip incl. D2H)", "ms": dt * 1e3, "Mrays_s": 2048 * 2048 / dt / 1e6})
for o in out:
    print(json.dumps(o))
