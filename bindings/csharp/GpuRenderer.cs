// GpuRenderer.cs -- P/Invoke binding of libsoftray_hip.so (include/softray.h) for the reference's C# host.
//
// SOURCE ONLY: the build image has no C# toolchain (no dotnet / mono / csc), so this file is compile-checked by
// nobody here; it is the binding a maintainer of voidstar69/softray adds to Engine3D (see INTEGRATION.md).  The
// struct layouts mirror include/softray.h field for field (LayoutKind.Sequential, natural alignment).
//
// Two ways to use it:
//   1. patch Renderer.RaytraceGeometry (Engine3D/Renderer.cs:1655-1686): replace the Task fan-out over RaytraceBlock
//      with SoftrayHip.RenderInto(this, instance)   -- the drop-in; every public field keeps its meaning;
//   2. use GpuRenderer below as a stand-alone class with the same public surface.
using System;
using System.Runtime.InteropServices;

namespace Engine3D.Hip
{
    [StructLayout(LayoutKind.Sequential)]
    public struct SrPrim
    {
        public int kind;                 // 0 Sphere, 1 Plane, 2 Triangle
        public uint argb;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 9)] public double[] p;
    }

    [StructLayout(LayoutKind.Sequential)]
    public struct SrFrame
    {
        public int width, height, start_row, end_row, sub_pixel_res;
        public uint background_argb, flags;
        public int random_seed, shadow_samples, trace_mode, strip_rows, strip_count, strip_index, max_bounces, concurrency, reserved0;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 12)] public double[] transform;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 12)] public double[] inv_transform;
        public double position_z, fov_depth, focal_depth, focal_blur_strength, ambient, shininess;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public double[] light_dir_view;
        [MarshalAs(UnmanagedType.ByValArray, SizeConst = 3)] public double[] light_pos_view;
        public double reflectivity;
        public IntPtr area_light_offsets;   // double[shadow_samples][3] produced with the REAL System.Random, or IntPtr.Zero
    }

    internal static class Native
    {
        const string Lib = "softray_hip";
        [DllImport(Lib)] public static extern int sr_create(int device, out IntPtr scene);
        [DllImport(Lib)] public static extern void sr_destroy(IntPtr scene);
        [DllImport(Lib)] public static extern int sr_set_triangles(IntPtr scene, double[] v9, uint[] argb, long n, double[] boxMin, double[] boxMax);
        [DllImport(Lib)] public static extern int sr_set_extra_geometry(IntPtr scene, [In] SrPrim[] prims, int n);
        [DllImport(Lib)] public static extern int sr_build(IntPtr scene, uint modes, int maxDepth, int maxPerLeaf);
        [DllImport(Lib)] public static extern int sr_tree_stats(IntPtr scene, [Out] int[] out4);
        [DllImport(Lib)] public static extern int sr_render(IntPtr scene, ref SrFrame frame, [In, Out] int[] pixels, [Out] ulong[] stats4);
        [DllImport(Lib)] public static extern int sr_load_3ds(IntPtr scene, byte[] data, UIntPtr len);
        [DllImport(Lib)] public static extern int sr_post_process(IntPtr scene, [In, Out] int[] pixels, long count, int style, uint backgroundColor);
        [DllImport(Lib)] public static extern int sr_anti_alias(IntPtr scene, [In] int[] src, int dstWidth, int dstHeight, int resolution, [In, Out] int[] dst);
        [DllImport(Lib)] public static extern IntPtr sr_last_error();
        public static string LastError() { return Marshal.PtrToStringAnsi(sr_last_error()); }

        public const int SR_ERR_INVALID_ARG = -1, SR_ERR_OUT_OF_RANGE = -2, SR_ERR_NO_MODEL = -3, SR_ERR_FORMAT = -8;
        public static void Check(int rc)
        {
            if (rc == 0 || rc == SR_ERR_NO_MODEL) return;                       // no model: Render() returns silently (Renderer.cs:736-739)
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

    /// <summary>What Renderer.RaytraceGeometry hands to the device instead of the TPL fan-out over RaytraceBlock.</summary>
    public sealed class SoftrayHip : IDisposable
    {
        public const uint F_SHADING = 1, F_SHADOWS = 2, F_FOCAL_BLUR = 4, F_POINT_LIGHT = 8, F_SPECULAR = 16, F_STATIC_SHADOWS = 32;
        public const int MODE_REF_TREE = 0, MODE_BRUTE = 1, MODE_BVH = 2;
        IntPtr scene;
        Model uploaded;
        double[] areaLightOffsets;       // generated once with new Random(rayTraceRandomSeed): ShadowMethod.cs:63-73
        GCHandle offsetsPin;

        public SoftrayHip(int device = 0) { Native.Check(Native.sr_create(device, out scene)); }

        /// PreCalculate(): MakeRayTracableGeometry_simple (Renderer.cs:1452-1469) flattened + sr_build
        public void Upload(Model model, int mode)
        {
            if (!ReferenceEquals(model, uploaded))
            {
                int n = model.Triangles.Count;
                var v9 = new double[n * 9]; var argb = new uint[n];
                for (int i = 0; i < n; i++)
                {
                    Triangle tri = model.Triangles[i];
                    Vector[] v = { model.Vertices[tri.vertexIndex1].pos, model.Vertices[tri.vertexIndex2].pos, model.Vertices[tri.vertexIndex3].pos };
                    for (int k = 0; k < 3; k++) { v9[9 * i + 3 * k] = v[k].x; v9[9 * i + 3 * k + 1] = v[k].y; v9[9 * i + 3 * k + 2] = v[k].z; }
                    argb[i] = Surface.PackColorAndAlpha(tri.diffuseMaterial, 1.0);                   // Renderer.cs:1463
                }
                Native.Check(Native.sr_set_triangles(scene, v9, argb, n,
                    new[] { model.Min.x, model.Min.y, model.Min.z }, new[] { model.Max.x, model.Max.y, model.Max.z }));
                uploaded = model;
            }
            if (mode != MODE_BRUTE) Native.Check(Native.sr_build(scene, 1u << mode, 0, 0));         // SpatialSubdivision defaults 15 / 25
        }

        /// The host half of RaytraceGeometry (Renderer.cs:1510-1528, 1652-1653): copy public fields into sr_frame.
        public void Render(int width, int height, int[] pixels, Instance instance, Matrix transform, Matrix inverseTransform,
                           uint backgroundColor, uint flags, int mode, int startRow, int endRow, int subPixelRes, int randomSeed,
                           double fieldOfViewDepth, double focalDepth, double focalBlurStrength, double ambient, double shininess,
                           Vector lightDirView, Vector lightPosView, ulong[] stats4, int concurrency = 4)
        {
            if (areaLightOffsets == null)
            {
                var random = new Random(randomSeed);                                                   // Renderer.cs:1624
                areaLightOffsets = new double[300];
                for (int i = 0; i < 100; i++)
                {
                    var o = new Vector(random.NextDouble() * 2 - 1, random.NextDouble() * 2 - 1, random.NextDouble() * 2 - 1);
                    o.Normalise(); o *= 0.2;
                    areaLightOffsets[3 * i] = o.x; areaLightOffsets[3 * i + 1] = o.y; areaLightOffsets[3 * i + 2] = o.z;
                }
                offsetsPin = GCHandle.Alloc(areaLightOffsets, GCHandleType.Pinned);
            }
            var f = new SrFrame
            {
                width = width, height = height, start_row = startRow, end_row = endRow, sub_pixel_res = subPixelRes,
                background_argb = backgroundColor, flags = flags, random_seed = randomSeed, shadow_samples = 0, trace_mode = mode,
                concurrency = concurrency,                                   // rayTraceConcurrency: fill order of the static shadow cache
                transform = new double[12], inv_transform = new double[12],
                position_z = instance.Position.z, fov_depth = fieldOfViewDepth, focal_depth = focalDepth,
                focal_blur_strength = focalBlurStrength, ambient = ambient, shininess = shininess,
                light_dir_view = new[] { lightDirView.x, lightDirView.y, lightDirView.z },
                light_pos_view = new[] { lightPosView.x, lightPosView.y, lightPosView.z },
                area_light_offsets = offsetsPin.AddrOfPinnedObject()
            };
            for (int r = 0; r < 3; r++) for (int c = 0; c < 4; c++) { f.transform[4 * r + c] = transform[r, c]; f.inv_transform[4 * r + c] = inverseTransform[r, c]; }
            Native.Check(Native.sr_render(scene, ref f, pixels, stats4));      // blocking; `pixels` is only touched during the call
        }

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
