// GpuRenderer.cs -- P/Invoke binding of libsoftray_hip.so (include/softray.h) for the reference's C# host.
//
// SOURCE ONLY: the build image has no C# toolchain (no dotnet / mono / csc), so nobody compiles this file here; it is the
// binding a maintainer of voidstar69/softray adds to Engine3D (INTEGRATION.md shows where Renderer.cs calls it).  What WAS
// checked: every member of the reference it touches, against the reference sources --
//   Model.Triangles is ICollection<Triangle> (enumerated, never indexed), Model.Vertices[i].pos, Triangle.vertexIndex1..3,
//   Triangle.diffuseMaterial, Model.Min / Max                                  Engine3D/Model.cs:16,44-46,53,137-138,192-194
//   Surface.PackColorAndAlpha(Color, double)                                    Engine3D/Surface.cs:131
//   Instance.Position, Matrix this[row, col]                                    Engine3D/Instance.cs:47, Matrix.cs:16
//   GeometryCollection.Count / this[int]                                        Engine3D/Raytrace/GeometryCollection.cs:16,24
//   Raytrace.Triangle.Vertex1..3, .Color (uint)                                 Engine3D/Raytrace/Triangle.cs:59-67
//   Plane.Normal, Plane.DistanceToOrigin                                        Engine3D/Raytrace/Plane.cs:40,52
// and the FIVE accessors the reference lacks and INTEGRATION.md adds (private state the native side needs):
//   Sphere.Center, Sphere.Radius, Sphere.PackedColor, Plane.PackedColor, Instance.Transform / Instance.InverseTransform.
// The struct layouts mirror include/softray.h field for field (LayoutKind.Sequential, natural alignment); the byte offsets in
// the comments are the ones the header asserts at compile time (tests/test_abi.py compares them with this file).
//
// There is one entry point per Renderer call site: Upload (PreCalculate), SetExtraGeometry (ExtraGeometryToRaytrace),
// Render (RaytraceGeometry), PostProcess / AntiAlias (the two surface passes of Render()), Dispose.
using System;
using System.Runtime.InteropServices;

namespace Engine3D.Hip
{
    [StructLayout(LayoutKind.Sequential)]
    public struct SrPrim                     // sr_prim, 80 bytes
    {
        /* @0 */ public int kind;            // 0 Sphere {centre, radius}, 2 Triangle {v1, v2, v3}, 3 Plane {unit normal, originDist}
        /* @4 */ public uint argb;
        /* @8 */ [MarshalAs(UnmanagedType.ByValArray, SizeConst = 9)] public double[] p;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SrFrame                    // sr_frame, 368 bytes
    {
        /* @0   */ public int width;
        /* @4   */ public int height;
        /* @8   */ public int start_row;
        /* @12  */ public int end_row;
        /* @16  */ public int sub_pixel_res;
        /* @20  */ public uint background_argb;
        /* @24  */ public uint flags;
        /* @28  */ public int random_seed;
        /* @32  */ public int shadow_samples;
        /* @36  */ public int trace_mode;
        /* @40  */ public int strip_rows;
        /* @44  */ public int strip_count;
        /* @48  */ public int strip_index;
        /* @52  */ public int max_bounces;
        /* @56  */ public int concurrency;
        /* @60  */ public int reserved0;
        /* @64  */ [MarshalAs(UnmanagedType.ByValArray, SizeConst = 12)] public double[] transform;
        /* @160 */ [MarshalAs(UnmanagedType.ByValArray, SizeConst = 12)] public double[] inv_transform;
        /* @256 */ public double position_z;
        /* @264 */ public double fov_depth;
        /* @272 */ public double focal_depth;
        /* @280 */ public double focal_blur_strength;
        /* @288 */ public double ambient;
        /* @296 */ public double shininess;
        /* @304 */ [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public double[] light_dir_view;
        /* @328 */ [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public double[] light_pos_view;
        /* @352 */ public double reflectivity;
        /* @360 */ public IntPtr area_light_offsets;   // double[shadow_samples][3] made with the REAL System.Random, or IntPtr.Zero
    }
    // ByValArray fields of a struct passed by `ref` are marshalled inline (the marshaller copies the managed arrays into the
    // 368-byte native image and back): all five arrays must be non-null and exactly SizeConst long, which Render() guarantees.

    internal static class Native
    {
        const string Lib = "softray_hip";
        [DllImport(Lib)] public static extern int sr_create(int device, out IntPtr scene);
        [DllImport(Lib)] public static extern int sr_create_multi([In] int[] devices, int n, out IntPtr scene);
        [DllImport(Lib)] public static extern int sr_set_gather(IntPtr scene, int kind);                  // 0 peer copies (default), 1 grouped ncclSend / ncclRecv
        [DllImport(Lib)] public static extern int sr_render_device(IntPtr scene, ref SrFrame frame, IntPtr dPixels, IntPtr hipStream, IntPtr dStats);
        [DllImport(Lib)] public static extern void sr_destroy(IntPtr scene);
        [DllImport(Lib)] public static extern int sr_set_triangles(IntPtr scene, double[] v9, uint[] argb, long n, double[] boxMin, double[] boxMax);
        [DllImport(Lib)] public static extern int sr_set_extra_geometry(IntPtr scene, [In] SrPrim[] prims, int n);
        [DllImport(Lib)] public static extern int sr_build(IntPtr scene, uint modes, int maxDepth, int maxPerLeaf);
        [DllImport(Lib)] public static extern int sr_tree_stats(IntPtr scene, [Out] int[] out4);
        [DllImport(Lib)] public static extern int sr_render(IntPtr scene, ref SrFrame frame, [In, Out] int[] pixels, [Out] ulong[] stats4);
        [DllImport(Lib)] public static extern int sr_reset_shadow_cache(IntPtr scene);
        [DllImport(Lib)] public static extern int sr_load_3ds(IntPtr scene, byte[] data, UIntPtr len);
        [DllImport(Lib)] public static extern int sr_post_process(IntPtr scene, [In, Out] int[] pixels, long count, int style, uint backgroundColor);
        [DllImport(Lib)] public static extern int sr_anti_alias(IntPtr scene, [In] int[] src, int dstWidth, int dstHeight, int resolution, [In, Out] int[] dst);
        [DllImport(Lib)] public static extern IntPtr sr_last_error();
        [DllImport(Lib)] public static extern int sr_abi_version();
        public static string LastError() { return Marshal.PtrToStringAnsi(sr_last_error()); }

        public const int SR_ERR_INVALID_ARG = -1, SR_ERR_OUT_OF_RANGE = -2, SR_ERR_NO_MODEL = -3, SR_ERR_FORMAT = -8;
        public const int AbiVersion = 5;

        /// <param name="renderCall">true only for sr_render: Render() without a model draws nothing and returns (Renderer.cs:736-739)</param>
        public static void Check(int rc, bool renderCall = false)
        {
            if (rc == 0 || (renderCall && rc == SR_ERR_NO_MODEL)) return;
            string msg = LastError();
            switch (rc)
            {
                case SR_ERR_OUT_OF_RANGE: throw new ArgumentOutOfRangeException("geometry", msg);   // SpatialSubdivision.cs:293
                case SR_ERR_FORMAT: throw new FormatException(msg);                                   // Model.cs:555, ThreeDSFile.cs:168
                case SR_ERR_INVALID_ARG: throw new ArgumentException(msg);
                default: throw new InvalidOperationException(msg);
            }
        }
    }

    /// <summary>What Renderer hands to the device(s) instead of the TPL fan-out over RaytraceBlock (Renderer.cs:1655-1686).</summary>
    public sealed class SoftrayHip : IDisposable
    {
        public const uint F_SHADING = 1, F_SHADOWS = 2, F_FOCAL_BLUR = 4, F_POINT_LIGHT = 8, F_SPECULAR = 16, F_STATIC_SHADOWS = 32;
        const uint F_PRIMARY_STATS_ONLY = 1u << 12;     // Num* count primary rays (Renderer.cs:1916-1923): no counting in the shadow stage
        public const int MODE_REF_TREE = 0, MODE_BRUTE = 1, MODE_BVH = 2;
        /// How NumGeometryTests / NumNodeVisits / NumLeafNodeVisits (Renderer.cs:476-504) are answered -- an explicit choice of the
        /// constructor, because they are the literal reference-tree traversal's counters and the fast path does not walk that tree.
        ///   Literal  every model's primary rays walk the reference tree (SR_MODE_REF_TREE): the reference's counters, any model size;
        ///   Auto     (default) models of fewer than OwnBvhThreshold triangles -- the sizes the reference itself handles -- as Literal,
        ///            larger ones as Off;
        ///   Off      every subdivided model on the library's own BVH; Render() then leaves CountersAvailable false and the patched
        ///            Renderer getters throw InvalidOperationException (never a silent zero).
        /// NumRaysFired is exact in every mode; shadow rays take the shaft path on the own BVH in all three (same pixels:
        /// include/softray.h SR_MODE_BVH; obj.3DS at 1024^2: 0.36 ms literal, 0.17 ms on the own BVH).
        public enum TraversalCounters { Auto, Literal, Off }
        public readonly TraversalCounters Counters;
        public int OwnBvhThreshold = 2000;
        /// Did the last Render() produce the three traversal counters?  (false: it ran on the own BVH)
        public bool CountersAvailable { get; private set; } = true;
        IntPtr scene;
        Model uploaded;                  // the model whose triangles the scene holds
        uint builtModes;                 // structures built for `uploaded` (bit 1 << mode)
        SrFrame frame;                   // reused from call to call: its arrays are allocated once
        bool frameReady;
        int offsetsSeed;                 // seed areaLightOffsets was generated with
        double[] areaLightOffsets;       // new Random(rayTraceRandomSeed): ShadowMethod.cs:63-73
        GCHandle offsetsPin;

        /// <summary>One MI355X.</summary>
        public SoftrayHip(int device = 0, TraversalCounters counters = TraversalCounters.Auto)
        {
            Counters = counters;
            if (Native.sr_abi_version() != Native.AbiVersion) throw new InvalidOperationException("libsoftray_hip: ABI version mismatch");
            Native.Check(Native.sr_create(device, out scene));
        }

        /// <summary>A whole node from this one process: the frame's rows are split into interleaved 16-row strips over
        /// `devices` inside the library and copied straight into the caller's pixels (sr_create_multi).</summary>
        /// gatherOverRccl: frames that STAY on the first device (sr_render_device: a host that post-processes or displays from HBM)
        /// are gathered with one grouped ncclSend / ncclRecv over xGMI (sr_set_gather, include/softray.h) instead of peer copies;
        /// Render() into a managed int[] copies every device's strips over its own PCIe link either way.
        public SoftrayHip(int[] devices, TraversalCounters counters = TraversalCounters.Auto, bool gatherOverRccl = false)
        {
            Counters = counters;
            if (Native.sr_abi_version() != Native.AbiVersion) throw new InvalidOperationException("libsoftray_hip: ABI version mismatch");
            Native.Check(Native.sr_create_multi(devices, devices.Length, out scene));
            if (gatherOverRccl) Native.Check(Native.sr_set_gather(scene, 1));
        }

        /// PreCalculate() (Renderer.cs:673-699): MakeRayTracableGeometry_simple (:1452-1469) flattened, then the structure for
        /// `mode`.  Like the reference (geometry_* == null guards, :684-696) nothing is rebuilt while the model and the mode
        /// stay the same -- Renderer.Render() calls PreCalculate() every frame.
        public void Upload(Model model, int mode)
        {
            if (!ReferenceEquals(model, uploaded))
            {
                int n = model.Triangles.Count;
                var v9 = new double[n * 9]; var argb = new uint[n];
                int i = 0;
                foreach (Triangle tri in model.Triangles)                                            // triIndex = enumeration order, :1458-1467
                {
                    Vector v1 = model.Vertices[tri.vertexIndex1].pos, v2 = model.Vertices[tri.vertexIndex2].pos, v3 = model.Vertices[tri.vertexIndex3].pos;
                    v9[9 * i + 0] = v1.x; v9[9 * i + 1] = v1.y; v9[9 * i + 2] = v1.z;
                    v9[9 * i + 3] = v2.x; v9[9 * i + 4] = v2.y; v9[9 * i + 5] = v2.z;
                    v9[9 * i + 6] = v3.x; v9[9 * i + 7] = v3.y; v9[9 * i + 8] = v3.z;
                    argb[i] = Surface.PackColorAndAlpha(tri.diffuseMaterial, 1.0);                   // :1463
                    i++;
                }
                Native.Check(Native.sr_set_triangles(scene, v9, argb, n,
                    new[] { model.Min.x, model.Min.y, model.Min.z }, new[] { model.Max.x, model.Max.y, model.Max.z }));   // :1487
                uploaded = model;
                builtModes = 0;
            }
            uint bit = 1u << mode;
            if (mode == MODE_REF_TREE && UsesOwnBvh(model)) bit = 1u << MODE_BVH;                    // large model (or Off): only the own BVH is needed
            else if (mode == MODE_REF_TREE && model.Triangles.Count > 0) bit |= 1u << MODE_BVH;      // shadow rays of a tree frame take the shaft path
            if (mode != MODE_BRUTE && (builtModes & bit) != bit)
            {
                Native.Check(Native.sr_build(scene, bit & ~builtModes, 0, 0));                       // SpatialSubdivision defaults 15 / 25
                builtModes |= bit;
            }
        }

        bool UsesOwnBvh(Model model)
        {
            if (model.Triangles.Count == 0 || Counters == TraversalCounters.Literal) return false;   // (an empty model: nothing to build a BVH from)
            return Counters == TraversalCounters.Off || model.Triangles.Count >= OwnBvhThreshold;
        }

        /// The trace mode Render() should be given for `rayTraceSubdivision`: the reference tree (literal) or, for large models,
        /// the library's own BVH.
        public int TraceMode(bool rayTraceSubdivision)
        {
            if (!rayTraceSubdivision) return MODE_BRUTE;
            return (uploaded != null && UsesOwnBvh(uploaded)) ? MODE_BVH : MODE_REF_TREE;
        }

        /// ExtraGeometryToRaytrace (Renderer.cs:460, 1545-1549): the collection is scanned first to last BEFORE the model, with
        /// a strict '<' on rayFrac (GeometryCollection.cs:44-69) -- the order is kept.  Call it whenever the collection changes
        /// and before the reference appends the model's root to it (:1547).  Uses the accessors INTEGRATION.md adds.
        public void SetExtraGeometry(GeometryCollection extras)
        {
            int n = extras == null ? 0 : extras.Count;
            var prims = new SrPrim[Math.Max(n, 1)];
            for (int i = 0; i < n; i++)
            {
                var prim = new SrPrim { p = new double[9] };
                IRayIntersectable g = extras[i];
                var sphere = g as Sphere; var plane = g as Plane; var tri = g as Raytrace.Triangle; var box = g as AxisAlignedBox;
                if (sphere != null)
                {
                    prim.kind = 0; prim.argb = sphere.PackedColor;                                    // Color.ToARGB(), Sphere.cs:51-55
                    prim.p[0] = sphere.Center.x; prim.p[1] = sphere.Center.y; prim.p[2] = sphere.Center.z; prim.p[3] = sphere.Radius;
                }
                else if (plane != null)
                {
                    prim.kind = 3; prim.argb = plane.PackedColor;                                     // the stored unit normal and originDist, verbatim
                    prim.p[0] = plane.Normal.x; prim.p[1] = plane.Normal.y; prim.p[2] = plane.Normal.z; prim.p[3] = plane.DistanceToOrigin;
                }
                else if (tri != null)
                {
                    prim.kind = 2; prim.argb = tri.Color;
                    prim.p[0] = tri.Vertex1.x; prim.p[1] = tri.Vertex1.y; prim.p[2] = tri.Vertex1.z;
                    prim.p[3] = tri.Vertex2.x; prim.p[4] = tri.Vertex2.y; prim.p[5] = tri.Vertex2.z;
                    prim.p[6] = tri.Vertex3.x; prim.p[7] = tri.Vertex3.y; prim.p[8] = tri.Vertex3.z;
                }
                else if (box != null)
                {
                    prim.kind = 4; prim.argb = 0xffffffffu;                                           // six Color.White planes (AxisAlignedBox.cs:22-27, Plane.cs:28)
                    prim.p[0] = box.Min.x; prim.p[1] = box.Min.y; prim.p[2] = box.Min.z; prim.p[3] = box.Max.x; prim.p[4] = box.Max.y; prim.p[5] = box.Max.z;
                }
                else throw new NotSupportedException("ExtraGeometryToRaytrace holds a " + g.GetType().Name + ": only Sphere, Plane, Triangle and AxisAlignedBox have a device form");
                prims[i] = prim;
            }
            if (n == 0) prims[0].p = new double[9];
            Native.Check(Native.sr_set_extra_geometry(scene, prims, n));
        }

        /// The host half of RaytraceGeometry (Renderer.cs:1510-1528, 1652-1653): copy public fields into sr_frame.
        public void Render(int width, int height, int[] pixels, Instance instance, Matrix transform, Matrix inverseTransform,
                           uint backgroundColor, uint flags, int mode, int startRow, int endRow, int subPixelRes, int randomSeed,
                           double fieldOfViewDepth, double focalDepth, double focalBlurStrength, double ambient, double shininess,
                           Vector lightDirView, Vector lightPosView, ulong[] stats4, int concurrency = 4)
        {
            if (areaLightOffsets == null || offsetsSeed != randomSeed)
            {
                var random = new Random(randomSeed);                                                   // Renderer.cs:1624
                var table = new double[300];
                for (int i = 0; i < 100; i++)                                                          // softShadowQuality, ShadowMethod.cs:9
                {
                    var o = new Vector(random.NextDouble() * 2 - 1, random.NextDouble() * 2 - 1, random.NextDouble() * 2 - 1);
                    o.Normalise(); o *= 0.2;
                    table[3 * i] = o.x; table[3 * i + 1] = o.y; table[3 * i + 2] = o.z;
                }
                if (offsetsPin.IsAllocated) offsetsPin.Free();
                areaLightOffsets = table;
                offsetsPin = GCHandle.Alloc(areaLightOffsets, GCHandleType.Pinned);
                offsetsSeed = randomSeed;
            }
            if (!frameReady)
            {
                frame = new SrFrame { transform = new double[12], inv_transform = new double[12], light_dir_view = new double[3], light_pos_view = new double[3] };
                frameReady = true;
            }
            frame.width = width; frame.height = height; frame.start_row = startRow; frame.end_row = endRow; frame.sub_pixel_res = subPixelRes;
            frame.background_argb = backgroundColor; frame.flags = flags | F_PRIMARY_STATS_ONLY; frame.random_seed = randomSeed; frame.shadow_samples = 0; frame.trace_mode = mode;
            frame.concurrency = concurrency;                                 // rayTraceConcurrency: fill order of the static shadow cache
            frame.position_z = instance.Position.z; frame.fov_depth = fieldOfViewDepth; frame.focal_depth = focalDepth;
            frame.focal_blur_strength = focalBlurStrength; frame.ambient = ambient; frame.shininess = shininess;
            frame.light_dir_view[0] = lightDirView.x; frame.light_dir_view[1] = lightDirView.y; frame.light_dir_view[2] = lightDirView.z;
            frame.light_pos_view[0] = lightPosView.x; frame.light_pos_view[1] = lightPosView.y; frame.light_pos_view[2] = lightPosView.z;
            frame.area_light_offsets = offsetsPin.AddrOfPinnedObject();
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) { frame.transform[4 * r + c] = transform[r, c]; frame.inv_transform[4 * r + c] = inverseTransform[r, c]; }
            // blocking; `pixels` is only touched during the call (the library pins it for the call and copies row bands into it while
            // later bands still render).  The literal tree (and brute force) produce the reference's counters; the own BVH does not:
            // stats4 then gets the rays fired only and CountersAvailable turns false
            CountersAvailable = mode != MODE_BVH;
            ulong[] counters = CountersAvailable ? stats4 : null;
            Native.Check(Native.sr_render(scene, ref frame, pixels, counters), renderCall: true);
            if (!CountersAvailable && stats4 != null)
            {
                int a = Math.Min(Math.Max(0, startRow), height - 1), b = Math.Min(Math.Max(0, endRow), height - 1);   // Renderer.cs:1652-1653
                stats4[0] = b < a ? 0UL : (ulong)(b - a + 1) * (ulong)width * (ulong)(subPixelRes * subPixelRes);       // NumRaysFired (:1916)
                stats4[1] = stats4[2] = stats4[3] = 0;
            }
        }

        /// A new Renderer starts with an empty static shadow cache (ShadowMethod.cs:75-83)
        public void ResetShadowCache() { Native.Check(Native.sr_reset_shadow_cache(scene)); }

        /// PostProcessImage's colour functions (Renderer.cs:819-865): style = (int)Renderer.Style for Standard..DepthBanded.
        public void PostProcess(int[] pixels, int style, uint backgroundColor)
        {
            Native.Check(Native.sr_post_process(scene, pixels, pixels.LongLength, style, backgroundColor));
        }

        /// AntiAliasImage (Renderer.cs:937-978): surface.Pixels (dstWidth*res x dstHeight*res) -> antiAliasedSurface.Pixels.
        public void AntiAlias(int[] src, int dstWidth, int dstHeight, int resolution, int[] dst)
        {
            Native.Check(Native.sr_anti_alias(scene, src, dstWidth, dstHeight, resolution, dst));
        }

        public void Dispose()
        {
            if (offsetsPin.IsAllocated) offsetsPin.Free();
            if (scene != IntPtr.Zero) { Native.sr_destroy(scene); scene = IntPtr.Zero; }
        }
    }
}
