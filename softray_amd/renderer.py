"""Host-side mirror of the reference's public raytrace API (Engine3D.Renderer / Instance / Model /
GeometryCollection / Sphere / Plane / Triangle / Vector / Color), over the C ABI of include/softray.h.

Same names, same argument meaning, same error behaviour as the C# (file:line cites below), so the
parity tests read like Engine3D-Tests/Raytrace/RendererTests.cs.  Nothing is computed here: every
ray is traced by libsoftray_hip.so on the MI355X; this module only copies public fields into an
`sr_frame` exactly the way Renderer.RaytraceGeometry does (Renderer.cs:1501-1687).

Out of scope: the scan-line rasteriser, static-shadow / AO / light-field caches,
path tracing and voxels.  Asking for them raises NotImplementedError instead of silently differing.
"""
import math

import numpy as np

from . import _lib
from ._lib import (F_FOCAL_BLUR, F_POINT_LIGHT, F_SHADING, F_SHADOWS, F_SPECULAR, F_STATIC_SHADOWS, MODE_BRUTE, MODE_BVH,
                   MODE_REF_TREE, Frame)
from .scene import GpuScene, default_fov_depth, instance_matrices


def _to_byte(d):
    """C# unchecked (byte)(double)."""
    if not (-2147483649.0 < d < 2147483648.0):
        return 0
    return int(d) & 0xff


class Vector:
    """Engine3D.Vector (Vector.cs:9-197): three doubles."""
    __slots__ = ("x", "y", "z")

    def __init__(self, x=0.0, y=0.0, z=0.0):
        self.x, self.y, self.z = float(x), float(y), float(z)

    def __iter__(self):
        return iter((self.x, self.y, self.z))

    def __sub__(self, o):
        return Vector(self.x - o.x, self.y - o.y, self.z - o.z)

    def __mul__(self, s):
        return Vector(self.x * s, self.y * s, self.z * s)

    def Normalise(self):                      # Vector.cs:177-185: multiply by 1/len
        ln = math.sqrt(self.x * self.x + self.y * self.y + self.z * self.z)
        inv = 1.0 / ln
        self.x *= inv
        self.y *= inv
        self.z *= inv

    def __repr__(self):
        return "%r,%r,%r" % (self.x, self.y, self.z)


class Color:
    """Engine3D.Color (Color.cs:5-137)."""
    __slots__ = ("r", "g", "b")

    def __init__(self, r, g, b):
        self.r, self.g, self.b = float(r), float(g), float(b)

    def ToARGB(self):                         # Color.cs:105-111
        return (255 << 24) + (_to_byte(self.r * 255.0) << 16) + (_to_byte(self.g * 255.0) << 8) + _to_byte(self.b * 255.0)


Color.Black = Color(0.0, 0.0, 0.0); Color.White = Color(1.0, 1.0, 1.0); Color.Grey = Color(0.5, 0.5, 0.5)
Color.Red = Color(1.0, 0.0, 0.0); Color.Green = Color(0.0, 1.0, 0.0); Color.Blue = Color(0.0, 0.0, 1.0)
Color.Yellow = Color(1.0, 1.0, 0.0); Color.Orange = Color(1.0, 0.5, 0.0); Color.Brown = Color(0.5, 0.25, 0.0)
Color.Pink = Color(1.0, 0.0, 1.0); Color.Cyan = Color(0.0, 1.0, 1.0)


class Sphere:
    """Raytrace.Sphere (Sphere.cs:26-33); Color defaults to white."""

    def __init__(self, center, radius, Color=Color.White):
        if not radius > 0:
            raise ValueError("radius must be > 0")       # Contract.Requires(radius > 0)
        self.center, self.radius, self.Color = center, float(radius), Color

    def _prim(self):
        return (0, self.Color.ToARGB(), [self.center.x, self.center.y, self.center.z, self.radius])


class Plane:
    """Raytrace.Plane (Plane.cs:22-29): one-sided."""

    def __init__(self, point, normal, Color=Color.White):
        self.point, self.normal, self.Color = point, normal, Color

    def _prim(self):
        return (1, self.Color.ToARGB(), list(self.point) + list(self.normal))


class Triangle:
    """Raytrace.Triangle (Triangle.cs:29-57): one-sided, packed ARGB colour."""

    def __init__(self, v1, v2, v3, color):
        self.v1, self.v2, self.v3, self.color = v1, v2, v3, int(color)

    def _prim(self):
        return (2, self.color, list(self.v1) + list(self.v2) + list(self.v3))


class AxisAlignedBox:
    """Raytrace.AxisAlignedBox as an IRayIntersectable (AxisAlignedBox.cs:15-28, IntersectRay :60-95): six one-sided white planes."""

    def __init__(self, min, max):
        if not (min.x < max.x and min.y < max.y and min.z < max.z):
            raise ValueError("Axis aligned bounding box has bad coordinates")       # Contract.Requires, :17-19
        self.Min, self.Max = min, max

    def _prim(self):
        return (4, 0xffffffff, list(self.Min) + list(self.Max))


class GeometryCollection:
    """Raytrace.GeometryCollection (GeometryCollection.cs:8-31): ordered list of primitives."""

    def __init__(self):
        self._items = []

    def Add(self, geometry):
        self._items.append(geometry)

    @property
    def Count(self):
        return len(self._items)

    def __getitem__(self, i):
        return self._items[i]


class Model:
    """Engine3D.Model after Load3ds + PostProcessGeometry (Model.cs:522-653,750-831): triangles in the
    unit cube.  `FromTriangles` is the programmatic route (like Cloth.cs)."""

    def __init__(self):
        self.LoadingComplete = False
        self.LoadingError = False
        self._v9 = None
        self._argb = None
        self.Min = None
        self.Max = None

    @staticmethod
    def FromTriangles(v9, argb, box_min, box_max):
        m = Model()
        m._v9 = np.ascontiguousarray(v9, dtype=np.float64).reshape(-1, 3, 3)
        m._argb = np.ascontiguousarray(argb, dtype=np.uint32)
        m.Min, m.Max = Vector(*box_min), Vector(*box_max)
        m.LoadingComplete = True
        return m

    def Load3dsModelFromStream(self, stream):
        data = stream.read()
        tmp = GpuScene(device=-1)             # host-only handle: parsing is host work
        try:
            tmp.load_3ds(data)                # raises SoftrayError(SR_ERR_FORMAT) like FormatException
            v9, argb, bmin, bmax = tmp.get_triangles()
        except Exception:
            self.LoadingError = True          # Model.cs:199
            raise
        finally:
            tmp.close()
        self._v9, self._argb = v9, argb
        self.Min, self.Max = Vector(*bmin), Vector(*bmax)
        self.LoadingComplete = True
        self.LoadingError = False

    @property
    def Triangles(self):
        return self._v9


class Instance:
    """Engine3D.Instance (Instance.cs:21-51)."""

    def __init__(self, model, Position=None, Yaw=0.0, Pitch=0.0, Roll=0.0):
        if model is None:
            raise ValueError("model != null")                      # Contract.Requires(model != null)
        self.Model = model
        self.FieldOfViewDepth = 0.5                                # 90 degree FOV until Render() sets it
        self.Position = Position if Position is not None else Vector(0.0, 0.0, 1.5)
        self.Yaw, self.Pitch, self.Roll = Yaw, Pitch, Roll


class TraversalCounters:
    """How a Renderer of this library answers NumGeometryTests / NumNodeVisits / NumLeafNodeVisits (Renderer.cs:476-504) -- an
    explicit choice of the constructor, because the counters are the literal reference-tree traversal's and the fast path does
    not walk that tree:

    Literal  every model's primary rays walk the reference tree literally (SR_MODE_REF_TREE); the three counters are the
             reference's, for any model size (slow for large models);
    Auto     (default) models of fewer than gpuOwnBvhThreshold triangles -- the sizes the reference itself handles -- as Literal;
             larger ones as Off;
    Off      every subdivided model on the library's own BVH; reading one of the three counters raises InvalidOperationException
             (they are never silently zero).
    NumRaysFired is exact in every mode.  Shadow rays take the shaft path on the own BVH in all three (same pixels)."""
    Auto, Literal, Off = range(3)


class InvalidOperationException(RuntimeError):
    """System.InvalidOperationException: a traversal counter was read although the frame did not produce it."""


class Style:
    """Renderer.Style (Renderer.cs:23-33): "style of rendering (actually post-processing)"."""
    Standard, ColorShuffle, Negative, DepthSmooth, DepthBanded, Normals, Count = range(7)


class Renderer:
    """Engine3D.Renderer, raytrace half.  Public fields keep the C# names (Renderer.cs:35-139)."""

    Style = Style

    TraversalCounters = TraversalCounters

    def __init__(self, device=0, traversalCounters=TraversalCounters.Auto):
        if traversalCounters not in (TraversalCounters.Auto, TraversalCounters.Literal, TraversalCounters.Off):
            raise ValueError("traversalCounters must be TraversalCounters.Auto, .Literal or .Off")
        self.gpuTraversalCounters = traversalCounters
        self.RenderStyle = Style.Standard                          # Renderer.cs:35
        self.depthBuffer = False                                   # Renderer.cs:56-57 (only steer the two depth styles here)
        self.depthBufferHires = False
        # lighting (view space), Renderer.cs:207-217
        self.ambientLight_intensity = 0.1
        d = Vector(-1, -1, 1)
        d.Normalise()
        self.directionalLight_dir = d
        self.positionalLight_pos = Vector(0.0, 0.0, 1.5) - d * 2
        self.specularLight_shininess = 100.0
        self.pointLighting = True
        self.specularLighting = True
        # raytracing options, Renderer.cs:75-92
        self.rayTrace = False
        self.rayTraceShading = True
        self.rayTraceShadows = False
        self.rayTraceShadowsStatic = False
        self.rayTraceAmbientOcclusion = False
        self.rayTraceLightField = False
        self.rayTraceSubdivision = True
        self.rayTracePathTracing = False
        self.rayTraceVoxels = False
        self.rayTraceFocalBlur = True
        self.rayTraceFocalDepth = 1.5
        self.rayTraceFocalBlurStrength = 10.0
        self.rayTraceConcurrency = 4          # kept for API compatibility; the GPU renders exact row ranges
        self.rayTraceSubPixelRes = 1
        self.rayTraceRandomSeed = 1234567890
        self.rayTraceStartRow = 0
        self.rayTraceEndRow = 0
        # MI355X additions (not in the reference): which acceleration structure the device walks
        self.gpuTraceMode = None              # None: chosen per model (see _mode); or MODE_REF_TREE / MODE_BRUTE / MODE_BVH
        # Which structure a subdivided model's primary rays walk follows from the constructor's TraversalCounters choice: the literal
        # reference tree (with the reference's three counters) for every model (Literal), for models below gpuOwnBvhThreshold
        # triangles (Auto, the default), or for none (Off: the own BVH; the counters then raise instead of reading zero).  Same
        # pixels either way except for the documented 1e-10 leaf-face case (include/softray.h SR_MODE_BVH), which is one more
        # reason why models of the reference's own sizes keep the literal tree by default.  obj.3DS at 1024^2: 0.36 ms literal,
        # 0.17 ms on the own BVH
        self.gpuOwnBvhThreshold = 2000
        self.gpuTreeMaxDepth = 0              # 0 => SpatialSubdivision defaults 15 / 25
        self.gpuTreeMaxGeometryPerNode = 0
        self.gpuMaxBounces = 0                # config-5 extension: mirror bounces (0 = reference behaviour)
        self.gpuReflectivity = 0.0
        self.Instances = []
        self.ExtraGeometryToRaytrace = GeometryCollection()
        self.CachePath = "./cache"
        self._backgroundColor = 0
        self._fieldOfViewDepth = default_fov_depth()               # Renderer.cs:97-101
        self._width = self._height = 1
        self._pixels = np.zeros(1, dtype=np.int32)
        self._antiAliasResolution = 1                              # Renderer.cs:155-156
        self._aaSurface = None                                     # (width, height, pixels) of antiAliasedSurface
        self._modelVolatile = None
        self._model = None
        self._scene = GpuScene(device)
        self._built = set()
        self._sceneModel = None
        self._extraSig = None
        self._stats = np.zeros(4, dtype=np.uint64)
        self._haveCounters = True                                  # before the first frame the reference reads zeros

    # ---- properties ----
    @property
    def BackgroundColor(self):
        return self._backgroundColor

    @BackgroundColor.setter
    def BackgroundColor(self, value):
        self._backgroundColor = int(value) & 0x00FFFFFF           # Renderer.cs:308-321

    @property
    def BackgroundColorWithAlpha(self):
        return self._backgroundColor | 0xFF000000

    @property
    def AntiAliasResolution(self):
        return self._antiAliasResolution

    @AntiAliasResolution.setter
    def AntiAliasResolution(self, value):                          # Renderer.cs:366-413
        value = int(value)
        if value <= 0:
            raise ValueError("AntiAliasResolution must be greater than zero")
        if value == self._antiAliasResolution:
            return
        if self._aaSurface is not None:                            # restore the original surface
            w, h, pixels = self._aaSurface
        else:
            w, h, pixels = self._width, self._height, self._pixels
        self._antiAliasResolution = value
        if value == 1:
            self._aaSurface = None
        else:
            self._aaSurface = (w, h, pixels)
            w, h = w * value, h * value
            pixels = np.zeros(w * h, dtype=np.int32)               # larger surface for pre-anti-aliased rendering
        self._SetSurface(w, h, pixels)

    @property
    def Model(self):
        return self._modelVolatile

    @Model.setter
    def Model(self, value):                                        # Renderer.cs:349-364
        self._modelVolatile = value
        if value is not None:
            value.LoadingComplete = True
            value.LoadingError = False

    NumRaysFired = property(lambda self: int(self._stats[0]))     # Renderer.cs:465-504

    def _counter(self, i, name):
        if not self._haveCounters:
            raise InvalidOperationException(
                "%s: the last frame ran on the library's own BVH, which does not produce the reference tree's traversal counters; "
                "construct the Renderer with traversalCounters=TraversalCounters.Literal (or raise gpuOwnBvhThreshold) to get them" % name)
        return int(self._stats[i])

    NumGeometryTests = property(lambda self: self._counter(1, "NumGeometryTests"))
    NumNodeVisits = property(lambda self: self._counter(2, "NumNodeVisits"))
    NumLeafNodeVisits = property(lambda self: self._counter(3, "NumLeafNodeVisits"))
    TraversalCountersAvailable = property(lambda self: bool(self._haveCounters))
    RenderingSurfaceWidth = property(lambda self: self._width)
    RenderingSurfaceHeight = property(lambda self: self._height)

    # ---- public methods ----
    def SetRenderingSurface(self, width, height, pixels):
        """Renderer.cs:593-626: the caller owns `pixels` (int32[>= width*height]); it is written in place."""
        pixels = np.asarray(pixels)
        if pixels.dtype != np.int32 or pixels.size < width * height or not pixels.flags["C_CONTIGUOUS"]:
            raise ValueError("pixels must be a contiguous int32 array of at least width*height elements")
        n = self._antiAliasResolution
        if width * n == self._width and height * n == self._height:    # resolution unchanged: swap the buffer only
            if n > 1:
                self._aaSurface = (width, height, pixels)
            else:
                self._pixels = pixels
            return
        if n > 1:
            self._aaSurface = (width, height, pixels)
            width, height = width * n, height * n
            pixels = np.zeros(width * height, dtype=np.int32)
        self._SetSurface(width, height, pixels)

    def _SetSurface(self, width, height, pixels):                  # Renderer.cs:617-626
        self._pixels = pixels
        self._width, self._height = width, height
        self.rayTraceStartRow = 0
        self.rayTraceEndRow = height - 1

    def Load3dsModelFromStream(self, stream):                      # Renderer.cs:629-635
        self._modelVolatile = Model()
        self._modelVolatile.Load3dsModelFromStream(stream)

    def HasModelCompletedLoading(self):
        return self._modelVolatile is not None and self._modelVolatile.LoadingComplete

    def HasModelLoadFailed(self):
        return self._modelVolatile is not None and self._modelVolatile.LoadingError

    def _PinModel(self):                                           # Renderer.cs:791-810
        if self._modelVolatile is None:
            return False
        if self._modelVolatile.LoadingComplete:
            self._modelVolatile.LoadingComplete = False
            self._model = self._modelVolatile
            return True
        return self._model is not None

    def _literal(self):
        """Do the model's primary rays walk the reference tree (and produce its counters)?"""
        if self._model is not None and len(self._model._argb) == 0:
            return True                                            # an empty model: nothing to build a BVH from
        if self.gpuTraversalCounters == TraversalCounters.Literal:
            return True
        if self.gpuTraversalCounters == TraversalCounters.Off:
            return False
        return self._model is None or len(self._model._argb) < self.gpuOwnBvhThreshold

    def _mode(self):
        if self.gpuTraceMode is not None:
            return self.gpuTraceMode
        if not self.rayTraceSubdivision:
            return MODE_BRUTE
        return MODE_REF_TREE if self._literal() else MODE_BVH

    def PreCalculate(self):
        """Renderer.cs:673-699: MakeRayTracableGeometry_simple / _subdivided, here = upload + build on demand."""
        if not self._PinModel():
            raise RuntimeError("PreCalculate: no model is loading (the reference would spin forever, Renderer.cs:676-679)")
        if not self.rayTrace:
            return
        if self._sceneModel is not self._model:
            m = self._model
            self._scene.set_triangles(m._v9, m._argb, list(m.Min), list(m.Max))
            self._sceneModel = m
            self._built = set()
        mode = self._mode()
        want = set() if mode == MODE_BRUTE else {mode}
        if mode == MODE_REF_TREE and self.rayTraceShadows and not self.rayTraceShadowsStatic and len(self._model._argb) > 0:
            want.add(MODE_BVH)                                     # a tree frame's shadow rays take the BVH's shaft path
        need = tuple(sorted(want - self._built))
        if need:
            self._scene.build(need, self.gpuTreeMaxDepth, self.gpuTreeMaxGeometryPerNode)
            self._built.update(need)

    def Render(self):
        """Renderer.cs:701-778."""
        if not self.rayTrace:
            raise NotImplementedError("the scan-line rasteriser is out of scope of the MI355X hot path (SURVEY.md 2, row 21)")
        if not self._PinModel():
            return                                                 # silently, Renderer.cs:736-739
        for name in ("rayTraceAmbientOcclusion", "rayTraceLightField", "rayTracePathTracing", "rayTraceVoxels"):
            if getattr(self, name):
                raise NotImplementedError("%s is out of scope (RNG-order / racy-cache dependent in the reference; SURVEY.md 2)" % name)
        for instance in self.Instances:
            instance.FieldOfViewDepth = self._fieldOfViewDepth     # Renderer.cs:749
            self._RaytraceGeometry(instance)
        self._PostProcessImage()                                   # Renderer.cs:765
        self._AntiAliasImage()                                     # Renderer.cs:767

    def _PostProcessImage(self):                                   # Renderer.cs:819-897
        style = self.RenderStyle
        if style in (Style.DepthSmooth, Style.DepthBanded) and (self.depthBufferHires or not self.depthBuffer):
            return                                                 # hires branch is a TODO in the reference, :837-842
        if style == Style.Normals:
            if self.depthBuffer or self.depthBufferHires:
                raise NotImplementedError("Style.Normals reads the rasteriser's depth buffer (out of scope, SURVEY.md 2 row 21)")
            return
        if style != Style.Standard:
            self._scene.post_process(self._pixels.reshape(-1), style, self._backgroundColor)   # every pixel of the surface array

    def _AntiAliasImage(self):                                     # Renderer.cs:937-978
        if self._antiAliasResolution < 2:
            return
        w, h, pixels = self._aaSurface
        self._scene.anti_alias(self._pixels.reshape(-1)[: self._width * self._height], w, h, self._antiAliasResolution,
                               out=pixels.reshape(-1)[: w * h])

    def Dispose(self):                                             # Renderer.cs:236-256
        if self._scene is not None:
            self._scene.close()
            self._scene = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.Dispose()

    # ---- the part of RaytraceGeometry that stays on the host: copy fields into sr_frame ----
    def BuildFrame(self, instance):
        f = Frame()
        f.width, f.height = self._width, self._height
        f.start_row, f.end_row = self.rayTraceStartRow, self.rayTraceEndRow
        f.sub_pixel_res = self.rayTraceSubPixelRes
        f.background_argb = self._backgroundColor
        flags = 0
        if self.rayTraceShading:
            flags |= F_SHADING
        if self.rayTraceShadows:
            flags |= F_SHADOWS
            if self.rayTraceShadowsStatic:
                flags |= F_STATIC_SHADOWS                         # Renderer.cs:1625; cache lives in the GpuScene
        if self.rayTraceFocalBlur:
            flags |= F_FOCAL_BLUR
        if self.pointLighting:
            flags |= F_POINT_LIGHT
        if self.specularLighting:
            flags |= F_SPECULAR
        f.flags = flags | _lib.F_PRIMARY_STATS_ONLY               # Num* count primary rays (Renderer.cs:1916-1923)
        f.random_seed = self.rayTraceRandomSeed
        f.shadow_samples = 0
        f.trace_mode = self._mode()
        t, it = instance_matrices(list(instance.Position), instance.Yaw, instance.Pitch, instance.Roll)   # Instance.cs:134-135
        for i in range(12):
            f.transform[i] = t[i]
            f.inv_transform[i] = it[i]
        f.position_z = instance.Position.z
        f.fov_depth = instance.FieldOfViewDepth
        f.focal_depth = self.rayTraceFocalDepth
        f.focal_blur_strength = self.rayTraceFocalBlurStrength
        f.ambient = self.ambientLight_intensity
        f.shininess = self.specularLight_shininess
        for i, v in enumerate(self.directionalLight_dir):
            f.light_dir_view[i] = v
        for i, v in enumerate(self.positionalLight_pos):
            f.light_pos_view[i] = v
        f.max_bounces = self.gpuMaxBounces
        f.concurrency = self.rayTraceConcurrency                 # fixes the static-shadow cache fill order
        f.reflectivity = self.gpuReflectivity
        f.area_light_offsets = None
        return f

    def _RaytraceGeometry(self, instance):                         # Renderer.cs:1501-1687
        self.PreCalculate()
        sig = tuple((type(g).__name__, g._prim()[1], tuple(g._prim()[2])) for g in self.ExtraGeometryToRaytrace._items)
        if sig != self._extraSig:
            self._scene.set_extra([g._prim() for g in self.ExtraGeometryToRaytrace._items])
            self._extraSig = sig
        # Renderer.cs:1652-1653
        self.rayTraceStartRow = min(max(0, self.rayTraceStartRow), self._height - 1)
        self.rayTraceEndRow = min(max(0, self.rayTraceEndRow), self._height - 1)
        frame = self.BuildFrame(instance)
        view = self._pixels.reshape(-1)[: self._width * self._height]
        if frame.trace_mode != MODE_BVH:                           # the literal tree (or brute force): the reference's counters
            _, self._stats = self._scene.render(frame, out=view, stats=True)
            self._haveCounters = True
        else:                                                      # the own BVH does not produce them: reading one raises
            self._scene.render(frame, out=view, stats=False)
            rows = max(0, self.rayTraceEndRow - self.rayTraceStartRow + 1)
            self._stats = np.array([rows * self._width * self.rayTraceSubPixelRes ** 2, 0, 0, 0], dtype=np.uint64)   # NumRaysFired, Renderer.cs:1916
            self._haveCounters = False


__all__ = ["Renderer", "Style", "TraversalCounters", "InvalidOperationException", "Instance", "Model", "GeometryCollection", "Sphere", "Plane", "Triangle", "AxisAlignedBox", "Vector", "Color",
           "MODE_REF_TREE", "MODE_BRUTE", "MODE_BVH"]
