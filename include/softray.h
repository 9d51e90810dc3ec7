/*
 * softray.h -- C ABI of libsoftray_hip.so: the MI355X (gfx950) implementation of Engine3D's
 * per-pixel raytrace hot path (voidstar69/softray: Engine3D/Raytrace/ + the raytrace half of
 * Engine3D/Renderer.cs).
 *
 * The reference has NO native/FFI boundary (pure C#, single process).  The seam is cut
 * where the reference fans out to worker tasks:
 *
 *     Renderer.RaytraceBlock(instance, geometry, left, top, sizeX, sizeY)      Engine3D/Renderer.cs:1690
 *
 * Everything RaytraceBlock and the decorators below it read is passed in `sr_frame`; everything
 * PreCalculate() builds once per model is held in `sr_scene`.  A C# `[DllImport("softray_hip")]`
 * shim inside a Renderer-compatible class binds exactly these entry points
 * (bindings/csharp/GpuRenderer.cs); so do the ctypes mirror (softray_amd/) and the
 * C++ mirror (softray_amd/host/).
 *
 * Conventions: plain pointers and sizes only; every call is blocking unless it takes a stream; all
 * input arrays are copied; the library never retains a caller pointer after returning; a scene is
 * single-threaded like the reference ("Must only be executed by a single thread at a time",
 * Renderer.cs:1498), distinct scenes are independent.  Return value 0 = ok, <0 = SR_ERR_*;
 * sr_last_error() gives the thread-local message the shim wraps into the matching .NET exception.
 * There is NO CPU fallback: without a usable HIP device every compute call fails with
 * SR_ERR_NO_DEVICE.
 */
#ifndef SOFTRAY_H
#define SOFTRAY_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_ABI_VERSION 5
#define SR_STATS_COUNT 24   /* entries of the ray-statistics array (sr_last_ray_stats, sr_render_device's d_stats) */

enum {
    SR_OK                 =  0,
    SR_ERR_INVALID_ARG    = -1,   /* ArgumentException / ArgumentNullException                         */
    SR_ERR_OUT_OF_RANGE   = -2,   /* ArgumentOutOfRangeException: "A triangle vertex is outside the
                                     bounding box" (SpatialSubdivision.cs:287-295)                      */
    SR_ERR_NO_MODEL       = -3,   /* no triangles set: Renderer.Render() returns silently
                                     (Renderer.cs:736-739); the shim does the same                      */
    SR_ERR_NOT_BUILT      = -4,   /* sr_build() not called for the requested trace mode                */
    SR_ERR_UNSUPPORTED    = -5,
    SR_ERR_NO_DEVICE      = -6,   /* no HIP device / host-only scene: compute is refused, never emulated */
    SR_ERR_HIP            = -7,   /* a HIP runtime call failed (message in sr_last_error)               */
    SR_ERR_FORMAT         = -8    /* FormatException (3DS loader, ThreeDSFile.cs:166-169, Model.cs:553) */
};

/* per-frame flags = the public bool fields of Renderer (Renderer.cs:56-57,76-77,84) */
enum {
    SR_F_SHADING     = 1u << 0,   /* rayTraceShading  -> ShadingMethod.Enabled  (Renderer.cs:1614)      */
    SR_F_SHADOWS     = 1u << 1,   /* rayTraceShadows (dynamic) -> ShadowMethod.Enabled (:1632)          */
    SR_F_FOCAL_BLUR  = 1u << 2,   /* rayTraceFocalBlur (only read when sub_pixel_res > 1, :1744-1790)   */
    SR_F_POINT_LIGHT = 1u << 3,   /* pointLighting   (Scene.cs:20)                                      */
    SR_F_SPECULAR    = 1u << 4,   /* specularLighting (Scene.cs:21)                                     */
    SR_F_STATIC_SHADOWS = 1u << 5, /* rayTraceShadowsStatic, with SR_F_SHADOWS (ShadowMethod.cs:75-83,103-108): the light
                                     fraction of a surface point is (byte)(fraction*254+1) looked up in a 128^3 texture over
                                     the unit cube; an empty cell is generated for whoever asks first and then kept by the
                                     scene for later frames (sr_reset_shadow_cache = a new Renderer).  The reference's worker
                                     tasks race for the cells; the library uses the deterministic order that reproduces the
                                     reference's goldens (RendererTests.RaytraceStaticShadow): the `concurrency` row blocks
                                     advance in lock step -- row r of every block, blocks ascending, before row r + 1;
                                     columns ascending; sub-samples in loop order.  Needs the whole frame in one call:
                                     SR_ERR_UNSUPPORTED with strips, mirror bounces or SR_F_SINGLE_KERNEL            */
    SR_F_SINGLE_KERNEL = 1u << 8, /* library option, not a Renderer field: trace the frame with the one-kernel
                                     renderer (k_render) instead of the k_primary/k_shadow/k_resolve pipeline.
                                     Pixels are identical; kept as an independent cross-check               */
    SR_F_NO_SPLIT    = 1u << 10,  /* library option: run the frame as ONE pipeline on the caller's stream instead of two
                                     halves on two internal streams (the default: the latency-bound tail kernels of one
                                     half overlap the other half's work, -5 % frame time).  Same pixels; used to time
                                     kernels that do not share the GPU with another kernel                       */
    SR_F_PER_LANE_SHADOWS = 1u << 9, /* library option: trace shadow samples one lane per hit point (k_shadow)
                                     instead of one wavefront per hit point with a shared shaft walk
                                     (k_shadow_packet).  Pixels are identical; cross-check                  */
    SR_F_LITERAL_SECONDARY = 1u << 11, /* library option: a SR_MODE_REF_TREE frame traces its SHADOW rays through the reference tree
                                     too.  By default (own BVH built, point light, <= 1024 samples) they are answered on the
                                     library's BVH (shaft path): "is there a hit with rayFrac <= 1.0" has the same answer, and the
                                     four statistics of sr_render count the primary rays, which keep the literal traversal either
                                     way.  With the flag the secondary counters of sr_last_ray_stats are the reference tree's  */
    SR_F_PRIMARY_STATS_ONLY = 1u << 12 /* library option: with `stats`, count the primary rays only -- the four statistics of sr_render
                                     (NumRaysFired, NumGeometryTests, NumNodeVisits, NumLeafNodeVisits).  The shadow stage then runs
                                     its uncounted kernels (counting costs atomics per hit point: obj.3DS 1024^2 + shadows 2.3 -> 1.1 ms)
                                     and entries 4.. of sr_last_ray_stats read 0.  What the hosts' Render() sets             */
};

/* how the model's triangles are intersected */
enum {
    SR_MODE_REF_TREE = 0,  /* SpatialSubdivision.IntersectRay, literal (rayTraceSubdivision = true);
                              same tree, same near/far order, same leaf-box containment rule            */
    SR_MODE_BRUTE    = 1,  /* GeometryCollection over geometry_simple (rayTraceSubdivision = false)     */
    SR_MODE_BVH      = 2   /* the library's own BVH: global nearest hit with the reference's per-triangle
                              arithmetic (same root-box clip, same rayFrac offset, hit must lie inside the
                              root box, ties -> lowest triangle index).  Identical to REF_TREE except for
                              hits closer than 1e-10 to a leaf-box face of the reference tree            */
};

/* ExtraGeometryToRaytrace element (Renderer.cs:460,1545-1549).  Order is preserved: the collection is
 * scanned first-to-last with a strict '<' on rayFrac (GeometryCollection.cs:44-69), then the model. */
typedef struct {
    int32_t  kind;      /* 0 Sphere {centre xyz, radius}          Raytrace/Sphere.cs:26-33
                           1 Plane  {point xyz, normal xyz}       Raytrace/Plane.cs:22-29
                           2 Triangle {v1, v2, v3}                Raytrace/Triangle.cs:29-57
                           3 Plane as the object holds it {Plane.Normal xyz (unit), Plane.DistanceToOrigin}
                             (Plane.cs:40-62): what a host passes for an EXISTING Plane -- kind 1 would
                             normalise and project again and could differ in the last bit
                           4 AxisAlignedBox {min xyz, max xyz} (AxisAlignedBox.cs:15-28, IntersectRay :60-95):
                             the nearest of its six one-sided planes' hits that lies on the box; the hit
                             carries the plane's colour (Color.White, Plane.cs:28): argb is not read;
                             NumRayTests = 6                                                            */
    uint32_t argb;      /* Color.ToARGB() of the primitive's Color                                      */
    double   p[9];
} sr_prim;

/* Everything RaytraceGeometry / RaytraceBlock / ShadingMethod / ShadowMethod read per frame
 * (Renderer.cs:1501-1829).  The host side (C# shim, C++ or ctypes mirror) fills it from the public
 * Renderer / Instance fields; the matrices are Instance.InitRender's (Instance.cs:134-135) so that the
 * host's own sin/cos are the ones used.  sr_instance_matrices() builds them for hosts that want it. */
typedef struct {
    int32_t  width, height;          /* SetRenderingSurface (Renderer.cs:593)                            */
    int32_t  start_row, end_row;     /* rayTraceStartRow / rayTraceEndRow, inclusive (:135-136,1652-1653)
                                        rendered EXACTLY (the reference's last task may overshoot by up to
                                        rayTraceConcurrency-1 rows, :1659-1670; those rows are identical) */
    int32_t  sub_pixel_res;          /* rayTraceSubPixelRes (:90)                                        */
    uint32_t background_argb;        /* BackgroundColor (:308); misses store it with alpha 0xFF (:1860)  */
    uint32_t flags;                  /* SR_F_*                                                           */
    int32_t  random_seed;            /* rayTraceRandomSeed (:92): area-light offsets, ShadowMethod.cs:63 */
    int32_t  shadow_samples;         /* 0 => 100 = softShadowQuality (ShadowMethod.cs:9). 1 with a zero
                                        offset table = hard-shadow variant (build-defined, unpinned)      */
    int32_t  trace_mode;             /* SR_MODE_*                                                        */
    int32_t  strip_rows, strip_count, strip_index;
                                     /* multi-GPU row interleave: this call renders rows r in
                                        [start_row,end_row] with (r / strip_rows) % strip_count ==
                                        strip_index into a COMPACT buffer (owned rows in order).
                                        strip_count == 0: off, pixels is the full W*H surface            */
    int32_t  max_bounces;            /* 0 = the reference.  1..16: EXTENSION for config 5 (no reference counterpart,
                                        parity unpinned): mirror bounces r = dir - n*(2 dir.n) from pos + n*0.001,
                                        each level coloured by the same shading/shadow chain, colours blended per
                                        channel ((s*(255-k))>>8) + ((r*k)>>8), k = (byte)(reflectivity*255)       */
    int32_t  concurrency;            /* rayTraceConcurrency (:92), <= 0 => 4.  Only read with SR_F_STATIC_SHADOWS: it fixes
                                        the order in which the shadow cache is filled (see that flag)               */
    int32_t  reserved0;              /* 0 */
    double   transform[12];          /* rows 0..2 of Instance._transform        (Instance.cs:134)        */
    double   inv_transform[12];      /* rows 0..2 of Instance._inverseTransform (Instance.cs:135)        */
    double   position_z;             /* Instance.Position.z (:1717, Instance.cs:182)                     */
    double   fov_depth;              /* Renderer.fieldOfViewDepth = 0.5 / tan(pi/8) (:101)               */
    double   focal_depth;            /* rayTraceFocalDepth (:87)                                         */
    double   focal_blur_strength;    /* rayTraceFocalBlurStrength (:88)                                  */
    double   ambient, shininess;     /* ambientLight_intensity, specularLight_shininess (:38,:41)        */
    double   light_dir_view[3];      /* directionalLight_dir (:39)                                       */
    double   light_pos_view[3];      /* positionalLight_pos (:40)                                        */
    double   reflectivity;           /* 0..1, only read when max_bounces > 0                              */
    const double* area_light_offsets;/* optional [shadow_samples][3] (e.g. produced by the C# shim with the
                                        real System.Random); NULL => derived from random_seed             */
} sr_frame;

/* Layout contract of the two structs that cross the boundary by value / by reference (x86-64 SysV and Windows x64 agree): the
 * C# shim mirrors it with [StructLayout(LayoutKind.Sequential)] (bindings/csharp/GpuRenderer.cs carries the same offsets
 * as comments), softray_amd/_lib.py with ctypes; tests/test_abi.py checks the three against each other. */
#ifdef __cplusplus
#define SR_LAYOUT_ASSERT(cond, msg) static_assert(cond, msg)
#else
#define SR_LAYOUT_ASSERT(cond, msg) _Static_assert(cond, msg)
#endif
SR_LAYOUT_ASSERT(sizeof(sr_prim) == 80, "sr_prim is 80 bytes: kind@0 argb@4 p@8");
SR_LAYOUT_ASSERT(offsetof(sr_prim, p) == 8, "sr_prim.p@8");
SR_LAYOUT_ASSERT(sizeof(sr_frame) == 368, "sr_frame is 368 bytes");
SR_LAYOUT_ASSERT(offsetof(sr_frame, flags) == 24 && offsetof(sr_frame, trace_mode) == 36 && offsetof(sr_frame, strip_rows) == 40 &&
                 offsetof(sr_frame, max_bounces) == 52 && offsetof(sr_frame, concurrency) == 56, "sr_frame int block");
SR_LAYOUT_ASSERT(offsetof(sr_frame, transform) == 64 && offsetof(sr_frame, inv_transform) == 160 && offsetof(sr_frame, position_z) == 256 &&
                 offsetof(sr_frame, fov_depth) == 264 && offsetof(sr_frame, focal_depth) == 272 && offsetof(sr_frame, ambient) == 288 &&
                 offsetof(sr_frame, light_dir_view) == 304 && offsetof(sr_frame, light_pos_view) == 328 &&
                 offsetof(sr_frame, reflectivity) == 352 && offsetof(sr_frame, area_light_offsets) == 360, "sr_frame double block");

typedef struct sr_scene sr_scene;    /* one per Renderer; freed by Dispose() (Renderer.cs:236)           */

/* device >= 0: HIP device ordinal.  device == -1: host-only scene (sr_set_*, sr_build, sr_tree_stats,
 * sr_load_3ds work; every compute call returns SR_ERR_NO_DEVICE). */
int  sr_create(int32_t device, sr_scene** out);
/* One scene over n HIP devices of this process (SURVEY 8b `sr_create(device_count, ...)`, 8e): the model and its trees are
 * replicated (sr_set_* / sr_build / sr_load_3ds act on every device; host builds run once), sr_render / sr_render_device split
 * the frame's rows into interleaved 16-row strips -- device g renders the strips s with s % n == g, all devices concurrently --
 * and the strips are copied straight into the caller's surface (device -> host over each device's own link, or peer-to-peer over
 * xGMI into the device surface, which lives on devices[0]).  The pixels do not depend on n (no reduction, no RNG).  This is
 * how a single-process host -- the C# Renderer -- uses a whole node.  Frames that need one global order (SR_F_STATIC_SHADOWS)
 * or that already carry strip_* fields are rendered by devices[0] alone.  The same ordinal may appear more than once. */
int  sr_create_multi(const int32_t* devices, int32_t n, sr_scene** out);
int32_t sr_device_count(const sr_scene*);
void sr_destroy(sr_scene*);

/* MakeRayTracableGeometry_simple (Renderer.cs:1452-1469): v9 = [n][3 vertices][xyz] in model space
 * (after Model.PostProcessGeometry), argb[n] = Surface.PackColorAndAlpha(diffuse, 1.0) (:1463),
 * box = AxisAlignedBox(model.Min, model.Max) (:1487).  TriangleIndex = position in the array (:1465). */
int  sr_set_triangles(sr_scene*, const double* v9, const uint32_t* argb, int64_t n,
                      const double box_min[3], const double box_max[3]);
/* ExtraGeometryToRaytrace (Renderer.cs:460); n == 0 clears */
int  sr_set_extra_geometry(sr_scene*, const sr_prim* prims, int32_t n);

/* PreCalculate() (Renderer.cs:673-699).  modes = bit mask (1 << SR_MODE_*) of the structures to build:
 * REF_TREE: new SpatialSubdivision(geom, box, max_depth, max_per_leaf) (SpatialSubdivision.cs:267-315;
 * <=0 => the defaults 15 / 25, :269-270; SR_ERR_OUT_OF_RANGE if a vertex is outside the box);
 * BVH: the library's own BVH; BRUTE needs nothing.  Host work + H2D copies. */
int  sr_build(sr_scene*, uint32_t modes, int32_t max_depth, int32_t max_per_leaf);
/* Where the library's own BVH is built.  Default (a scene with a device, more than 64 triangles): ON THE GPU -- Morton-ordered LBVH
 * (sr_lbvh.hip) collapsed to the four-wide form the packet walks traverse, 0.01 s for 1 M and 0.08 s for 10 M triangles (host
 * binned SAH: 0.2 s / 1.2 s); frames are within 2 % of the host tree's (the packet walks order a node's children per frame by
 * their distance from the camera / the light, which is what the Morton order lacked).  Pixels do not depend on the tree (the
 * traversal is exact for any conservative BVH).  SR_BUILD_ON_HOST (OR-ed into `modes`) asks for the host's binned-SAH builder;
 * SR_BUILD_ON_DEVICE insists on the device (SR_ERR_NO_DEVICE for a host-only scene). */
#define SR_BUILD_ON_DEVICE 0x100u
#define SR_BUILD_ON_HOST   0x200u
/* out = TreeDepth, NumNodes, NumLeafNodes, NumInternalNodes (SpatialSubdivision.cs:317-335) */
int  sr_tree_stats(const sr_scene*, int32_t out[4]);
/* the library's own BVH: out = depth, inner nodes, triangles, 1 if it was built on the device */
int  sr_bvh_stats(const sr_scene*, int64_t out[4]);
/* diagnostics: FNV-1a hashes of the host-built BVH's node array and of its leaf-ordered triangle indices (the host build must not
 * depend on the number of threads it ran on); SR_ERR_NOT_BUILT for a device-built tree */
int  sr_bvh_digest(const sr_scene*, uint64_t out[2]);
/* diagnostics: the four-children-per-node form of the host-built BVH that the wave-cooperative packet walks traverse (collapsed
 * from the binary tree: same boxes, same leaves, same leaf order): out = depth, nodes, child slots in use, leaves, triangles
 * in leaves; SR_ERR_NOT_BUILT for a device-built tree, SR_ERR_UNSUPPORTED if a link is broken */
int  sr_wide_tree_stats(const sr_scene*, int64_t out[5]);

/* Renderer.Render() for one Instance, raytrace path (Renderer.cs:701-778 -> RaytraceGeometry :1501 ->
 * RaytraceBlock :1690).  pixels = caller-owned int[W*H] ARGB, row-major pixels[row*W+col]
 * (Surface.DrawPixel, Surface.cs:174-181); only rows start_row..end_row are written; with strips the
 * buffer is the compact strip buffer.  stats (may be NULL) = NumRaysFired, NumGeometryTests,
 * NumNodeVisits, NumLeafNodeVisits (Renderer.cs:465-504) summed over the frame's PRIMARY rays
 * (deterministic, unlike the reference's racy per-block counters, :1695). */
int  sr_render(sr_scene*, const sr_frame*, int32_t* pixels, uint64_t stats[4]);
/* forget the static shadow cache (what a new Renderer / ShadowMethod starts with); sr_set_triangles does it too */
int  sr_reset_shadow_cache(sr_scene*);
/* Same, but `d_pixels` is DEVICE memory on the scene's device (e.g. a torch tensor's data_ptr) and the
 * work is enqueued on `hip_stream` (a hipStream_t; NULL = the null stream) without host sync. */
/* Ordering: the work is enqueued behind everything already on `hip_stream` and `hip_stream` continues only after it; a
 * shadowed frame is internally forked onto two library-owned streams (event fork / join), see SR_F_NO_SPLIT.  Frames of ONE scene
 * run in submission order whatever streams they are given (the scene's scratch and its per-camera / per-light records belong to one
 * frame at a time: a frame's stream waits for an event the previous frame of the scene left behind); to overlap frames, use scenes. */
int  sr_render_device(sr_scene*, const sr_frame*, void* d_pixels, void* hip_stream, uint64_t* d_stats /* device uint64[SR_STATS_COUNT] (see sr_last_ray_stats) or NULL */);
/* number of int32 pixels sr_render writes for this frame (W*H, or the compact strip size) */
int64_t sr_frame_pixel_count(const sr_frame*);

/* IRayIntersectable.IntersectRay in batch (Raytrace/IRayIntersectable.cs:31-48): the operator interface
 * every primitive, the tree and the decorators implement.  target = SR_MODE_* for the model alone, or
 * SR_TARGET_ROOT = the root geometry of the chain (extra geometry + model in `mode`, Renderer.cs:1536-1549).
 * Host arrays; outputs may be NULL.  counters[n][3] = NumRayTests, NumNodesVisited, NumLeafNodesVisited. */
#define SR_TARGET_ROOT 0x100
int  sr_trace_rays(sr_scene*, int32_t target, int64_t n, const double* starts, const double* dirs,
                   uint8_t* hit, double* ray_frac, double* pos, double* normal, uint32_t* color,
                   int32_t* tri_index, int32_t* counters);

/* The same with every array in DEVICE memory on the scene's device, enqueued on `hip_stream` without a host synchronisation: what a
 * throughput measurement of the reference's per-primitive / per-tree micro-benchmarks needs (TriangleTests.cs:100-330,
 * SpatialSubdivisionTests.cs:140-260 time IntersectRay alone, not a transfer).  Outputs may be NULL. */
int  sr_trace_rays_device(sr_scene*, int32_t target, int64_t n, const double* d_starts, const double* d_dirs,
                          uint8_t* d_hit, double* d_ray_frac, double* d_pos, double* d_normal, uint32_t* d_color,
                          int32_t* d_tri_index, int32_t* d_counters, void* hip_stream);

/* ShadingMethod.IntersectRay's colour step in batch (ShadingMethod.cs:36-68 -> CalcLighting :110-177): for n recorded
 * intersections out[i] = ModulatePackedColor(color[i], (byte)(255 * intensity)) with the frame's transform and lights (only the
 * matrices, position_z, fov_depth, lights, ambient, shininess and the POINT_LIGHT / SPECULAR flags of `frame` are read).
 * Host arrays.  The decorator's arithmetic on its own -- Math.Pow included -- without a traversal in front of it. */
int  sr_shade_points(sr_scene*, const sr_frame* frame, int64_t n, const double* pos, const double* normal, const uint32_t* color,
                     uint32_t* out);

/* n NextDouble() of new System.Random(seed) after `skip` samples have been drawn (Next() and NextDouble() consume one each):
 * hosts regenerate the reference's seeded test inputs with it (rays that continue the triangle stream, SpatialSubdivisionTests.cs:141,225) */
void sr_net_random_doubles(int32_t seed, int64_t skip, int64_t n, double* out);

/* Instance.InitRender matrices (Instance.cs:134-135, Matrix.cs:74-169): T = Trans(P)*Roll*Pitch*Yaw,
 * T^-1 = Yaw(-)*Pitch(-)*Roll(-)*Trans(-P); rows 0..2, row-major 3x4. */
void sr_instance_matrices(const double position[3], double yaw, double pitch, double roll,
                          double transform[12], double inv_transform[12]);
/* Renderer.fieldOfViewDepth (Renderer.cs:97-101) */
double sr_default_fov_depth(void);
/* ShadowMethod ctor (ShadowMethod.cs:63-73) with new Random(seed) (Renderer.cs:1624): out[count][3] */
void sr_area_light_offsets(int32_t seed, int32_t count, double* out3);

/* Model.Load3ds + Model.PostProcessGeometry (Model.cs:522-653,750-831; 3dsLoader/ThreeDSFile.cs:132-662):
 * parses a .3DS image and fills the scene's triangles / colours / box (= sr_set_triangles).  The counts
 * and the arrays can be read back with sr_get_triangles. */
int  sr_load_3ds(sr_scene*, const uint8_t* data, size_t len);
int64_t sr_num_triangles(const sr_scene*);
int  sr_get_triangles(const sr_scene*, double* v9, uint32_t* argb, double box_min[3], double box_max[3]);

/* Device time of the library's kernels (opt-in: sr_debug_set(SR_DBG_KERNEL_TIMING, 1)), measured with one HIP event pair per launch on the launch stream and
 * accumulated since sr_reset_kernel_times() (or scene creation): out[i] = {static kernel name, total ms,
 * launches}.  sr_kernel_times waits for the recorded events.  Returns the number of entries (<= cap). */
typedef struct { const char* name; float ms; int32_t launches; } sr_kernel_time;
void sr_reset_kernel_times(sr_scene*);
int  sr_kernel_times(sr_scene*, sr_kernel_time* out, int32_t cap);

/* Ray statistics of the last sr_render(..., stats != NULL): [0..3] primary rays {rays, triangle/primitive tests, nodes
 * visited, leaf nodes visited} -- the reference's notions in SR_MODE_REF_TREE / SR_MODE_BRUTE; on the own BVH the primary walk is one
 * packet walk per 8x8-pixel tile and [1..3] count what a WAVE fetched: [1] 64-byte camera-cone records consulted, [2] 64-byte
 * nodes, [3] 128-byte FP64 triangle records; [4..7] the same for secondary (shadow) rays -- on the shaft path [6],[7] are the shaft
 * walks' nodes / leaves; [8] triangle records staged through LDS by k_shadow_test, [9] hit points it processed,
 * [10] fp32 slab records read by k_shaft, [11] hit points it walked, [12] (sample, triangle) pairs k_shadow_test classified
 * in fp32, [13] pairs it had to decide with the exact FP64 test, [14] / [15] the part of [6] / [10] that came from private per-lane
 * shaft walks (later rounds) rather than from the packet walk; [16..19] the exact fallback's any-hit rays on their own {rays,
 * FP64 triangle records tested, nodes fetched per lane, leaves} (also contained in [4..7]); [20..23] the same for the mirror
 * rays of the bounce pipeline.  These are the counters the roofline's algorithmic bytes are priced from (DESIGN.md "Measurement"). */
int  sr_last_ray_stats(const sr_scene*, uint64_t out[SR_STATS_COUNT]);

/* Seeded synthetic triangle soup = SpatialSubdivisionTests.MakeRandomTriangles
 * (Engine3D-Tests/Raytrace/SpatialSubdivisionTests.cs:397-411) driven by the System.Random port: per triangle
 * v1 = U[0,space)^3 + origin, v2 = v1 + U[0,extent)^3, v3 = v1 + U[0,extent)^3, colour = (uint)Next()
 * (opaque != 0: 0xFF000000 | low 24 bits).  Used by bench.py and the tests so that C#, the CPU checker and
 * the device regenerate identical inputs. */
void sr_make_random_triangles(int32_t seed, int64_t n, double space, double extent, double origin, int32_t opaque,
                              double* v9, uint32_t* argb);

/* ---- the row-strip gather over RCCL / xGMI, native (replaces the TPL fan-out of Renderer.cs:1655-1680 across GPUs) ----
 * The frame's rows are dealt out in interleaved 16-row strips (strip s belongs to rank s % world); every rank renders its strips
 * into a compact buffer and ONE exchange step brings them to rank 0: grouped ncclSend (ranks 1..) / ncclRecv (rank 0) of
 * rows_r x W x 4 bytes each, followed on rank 0 by the row de-interleave (strided device copies) into the full surface.  No
 * reduction, no RNG: the frame does not depend on the split.  librccl is bound with dlopen at first use (a single-GPU host never
 * loads it; a process that already holds an RCCL -- PyTorch's -- keeps using that one); SR_ERR_UNSUPPORTED when it is absent.
 *
 * One process per GPU (no PyTorch needed): rank 0 calls sr_rccl_unique_id and hands the 128 bytes to the other ranks by whatever
 * means the host has (a file, a socket, MPI, a torch store); every rank calls sr_rccl_init(scene, id, world, rank) on its own
 * single-device scene (ncclCommInitRank on the scene's device), then per frame sr_rccl_render(scene, frame, d_full, stream):
 * renders this rank's strips of `frame` (strip_count must be 0: the split is the library's) and gathers; d_full (device memory,
 * W*H int32) is only written on rank 0 and may be NULL elsewhere.  sr_rccl_gather is the exchange step on its own, for a host
 * that rendered its strips itself (sr_frame.strip_rows = 16, strip_count = world, strip_index = rank) into d_strips.  Everything
 * is enqueued on `hip_stream`; consecutive frames of a scene must use the same stream.
 *
 * One process, several devices (sr_create_multi): sr_set_gather(scene, SR_GATHER_RCCL) makes sr_render_device gather the parts'
 * strips with the same grouped send / receive (ncclCommInitAll over the scene's devices, which must be distinct) instead of peer
 * copies -- SR_GATHER_COPY, the default: hipMemcpy2DAsync peer-to-peer where the devices allow it, pinned host staging where not.
 * sr_render (host surface) copies every part's strips over its own PCIe link either way. */
#define SR_RCCL_ID_BYTES 128
enum { SR_GATHER_COPY = 0, SR_GATHER_RCCL = 1 };
int  sr_rccl_unique_id(uint8_t out[SR_RCCL_ID_BYTES]);
int  sr_rccl_init(sr_scene*, const uint8_t id[SR_RCCL_ID_BYTES], int32_t world, int32_t rank);
int  sr_rccl_render(sr_scene*, const sr_frame*, void* d_full, void* hip_stream);
int  sr_rccl_gather(sr_scene*, const sr_frame*, const void* d_strips, void* d_full, void* hip_stream);
int  sr_set_gather(sr_scene* multi_device_scene, int32_t kind);

/* ---- surface passes that Renderer.Render() runs after the raytrace (Engine3D/Renderer.cs:765-767) ----
 * sr_post_process[_device]  = PostProcessImage's per-pixel colour functions (Renderer.cs:819-865, Surface.ApplyColorFunc
 *   Surface.cs:226-233), applied in place to `count` pixels.  `background_color` is Renderer.BackgroundColor (alpha
 *   masked off, Renderer.cs:304-320) and is only read by SR_STYLE_NEGATIVE.  The two depth styles are the reference's
 *   8-bit-alpha-depth twizzles (taken when depthBuffer && !depthBufferHires; the host decides).  Style.Normals reads the
 *   rasteriser's depth buffer and is outside the raytrace path: SR_ERR_UNSUPPORTED.
 * sr_anti_alias[_device]    = AntiAliasImage (Renderer.cs:937-978): src is (dst_width*resolution) x (dst_height*resolution),
 *   every destination pixel is the integer average of its resolution^2 source pixels per channel, alpha 255. */
enum {
    SR_STYLE_STANDARD = 0,       /* Style.Standard: nothing to do */
    SR_STYLE_COLOR_SHUFFLE = 1,  /* ZRGB -> 0GBR */
    SR_STYLE_NEGATIVE = 2,
    SR_STYLE_DEPTH_SMOOTH = 3,   /* ZRGB -> 0ZZZ */
    SR_STYLE_DEPTH_BANDED = 4    /* Z * 111 */
};
int  sr_post_process(sr_scene*, int32_t* pixels, int64_t count, int32_t style, uint32_t background_color);
int  sr_post_process_device(sr_scene*, void* d_pixels, int64_t count, int32_t style, uint32_t background_color, void* hip_stream);
int  sr_anti_alias(sr_scene*, const int32_t* src, int32_t dst_width, int32_t dst_height, int32_t resolution, int32_t* dst);
int  sr_anti_alias_device(sr_scene*, const void* d_src, int32_t dst_width, int32_t dst_height, int32_t resolution, void* d_dst,
                          void* hip_stream);

/* Test / experiment hooks of ONE scene.  The library never reads the process environment: a drop-in must not change its
 * schedule with the host's env.  value < 0 restores the default.  Used by tests/ and scripts/ only. */
enum {
    SR_DBG_BAND_SAMPLES   = 0,   /* samples per row band (default 16 Mi / 32 Mi): small values force several bands             */
    SR_DBG_ROUND_CAP0     = 1,   /* candidate-list length of shaft round 1 (default 40, <= 64)                                 */
    SR_DBG_ROUND_CAP1     = 2,   /* ... of round 2 (default 64, <= 1024): tiny lists force round 2 and the exact fallback      */
    SR_DBG_SPLIT          = 3,   /* concurrent part-frame pipelines (default 2, <= 4)                                          */
    SR_DBG_FB_RAY_CAP     = 4,   /* capacity of the fallback ray list                                                          */
    SR_DBG_BVH_LEAF       = 5,   /* triangles per leaf of the own BVH, host and device build (default 4, 1..15); read by the next sr_build */
    SR_DBG_KERNEL_SWITCH  = 6,   /* A/B switch of single optimisations, same pixels (0 = production): 31 the bounce pipeline walks its rays in
                                    queue order (no per-level ray sort); 61 the camera-ordered node copy keeps (lo, hi) planes; 71 no facing
                                    partition (the packet walks see every record of a leaf); 7 counts umbra decisions of the private shaft walk;
                                    32 a mirror-bounce level as ONE kernel (k_bounce) instead of prepare / walk / finish; 100 + T: the walk kernel
                                    fetches new rays at T busy lanes (default 24); 200 + K: K stack levels per lane in LDS (default 24);
                                    81 the tile kernels with one workgroup per 16x16 tile (no persistent grid); 82 k_primary on the persistent grid
                                    too (its loop form spills registers: opt-in); 84 the persistent shaft walk hands its tiles out in natural order
                                    (no longest-first lists); 830 + n: n resident workgroups per CU for it; 840 + q: a walk is long at q / 4 x the
                                    mean; 91 the first classification round on k_shadow_cls instead of k_shadow_cls_g                      */
    SR_DBG_KERNEL_TIMING  = 7,   /* > 0: record a HIP event pair around every launch (sr_kernel_times); default off           */
    SR_DBG_EXACT_SHADOW_TESTS = 8, /* > 0: k_shadow_test decides every (sample, triangle) pair with the FP64 arithmetic (no
                                    fp32 classification): an independent schedule of the same result, kept as a cross-check */
    SR_DBG_PER_LANE_SHAFT = 9,   /* bit 0: first shaft round with private per-lane walks (k_shaft) instead of the wave-cooperative
                                    packet walk (k_shaft_pkt); bit 1: later rounds with private walks instead of one wave per hit
                                    point (k_shaft_coop): same lists up to order, same pixels; cross-checks                     */
    SR_DBG_PER_LANE_PRIMARY = 10, /* > 0: primary rays with private per-lane walks instead of the packet walk + camera-cone filter */
    SR_DBG_ROUND2_NODES   = 11,  /* node budget of a private shaft walk of the later rounds (0 = unlimited): walks that exceed it hand
                                    their undecided samples to the exact fallback */
    SR_DBG_BUILD_THREADS  = 12,  /* threads of the host BVH build (default: the host's cores, at most 16); read by the next sr_build    */
    SR_DBG_BVH2_PACKETS   = 13,  /* > 0: the packet walks (k_primary, first shaft round) on the binary tree with a per-step vote instead of
                                    the four-wide tree with per-frame ordered children: same pixels; cross-check and A/B measurement */
    SR_DBG_NO_PEER        = 14,  /* > 0 (multi-device scene): sr_render_device gathers every part's strips through pinned host staging, as it
                                    does for a part whose memory the first device cannot read; test hook for that path                 */
    SR_DBG_LITERAL_SHADOWS = 15, /* > 0: no shortcut for ShadowMethod -- a directional light's samples are traced one by one although all of
                                    them provably escape, and a SR_MODE_REF_TREE frame traces its shadow rays through the reference tree
                                    (as with SR_F_LITERAL_SECONDARY); cross-checks of both shortcuts                                  */
    SR_DBG_COUNT          = 16
};
int  sr_debug_set(sr_scene*, int32_t key, int64_t value);

/* Diagnostics only: the pipeline's device counters of the last frame, summed over its concurrent part-frame pipelines (last row band of each)
 * {hit-queue entries (on the shaft path: the padded tile-queue slot count, not the hits), per-lane shadow work head, hit points that needed the long (round-2) candidate list,
 *  hit points sent to the exact per-lane fallback, fallback work head, 0, 0, 0}. */
int  sr_debug_counters(sr_scene*, uint32_t out[8]);

const char* sr_last_error(void);
int32_t     sr_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif
