/*
 * softray_oracle.h -- C interface of the CPU ORACLE.
 *
 * TEST INFRASTRUCTURE ONLY.  This is a statement-level CPU restatement (C++17, IEEE doubles,
 * -ffp-contract=off) of the raytrace hot path of voidstar69/softray (Engine3D/Raytrace/ +
 * the raytrace half of Engine3D/Renderer.cs).  Nothing under softray_amd/ may include, link,
 * import or call it: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do,
 * and there only as the checker / the reported CPU baseline.
 *
 * Pinning (see oracle/README.md and tests/test_oracle_*.py): 5 seeded tree KATs, the seeded
 * tree==brute-force differential tests, the primitive KATs and 17 golden BMPs of the
 * reference's own test-suite.  Sphere images and the hard-shadow / reflection variants have
 * no reference golden: "parity unpinned" for those.
 */
#ifndef SOFTRAY_ORACLE_H
#define SOFTRAY_ORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- System.Random (.NET Framework 4.0 BCL, Knuth subtractive) ---- */
typedef struct orc_random orc_random;
orc_random* orc_random_new(int32_t seed);
void        orc_random_free(orc_random*);
int32_t     orc_random_next(orc_random*);           /* Random.Next()                      */
int32_t     orc_random_next_max(orc_random*, int32_t max); /* Random.Next(int maxValue)  */
double      orc_random_next_double(orc_random*);    /* Random.NextDouble()                */
void        orc_random_next_doubles(orc_random*, int64_t n, double* out); /* n x NextDouble() */
void        orc_random_next_ints(orc_random*, int64_t n, int32_t* out);   /* n x Next()       */

/* ---- scene = what Renderer holds between frames ---- */
typedef struct orc_scene orc_scene;

/* extra geometry record (ExtraGeometryToRaytrace, Renderer.cs:460,1545-1549); order preserved */
typedef struct {
    int32_t  kind;      /* 0 sphere: p = centre(3), radius ; 1 plane: p = point(3), normal(3) ;
                           2 triangle: p = v1(3), v2(3), v3(3) */
    uint32_t argb;      /* packed colour, alpha forced/expected 0xFF */
    double   p[9];
} orc_prim;

/* per-frame parameters: everything RaytraceGeometry/RaytraceBlock/Shading/Shadow read
 * (Renderer.cs:1501-1829).  Layout is deliberately identical to sr_frame in include/softray.h
 * so that the tests can hand the same bytes to both. */
typedef struct {
    int32_t  width, height;          /* rendering surface (Renderer.cs:593)                     */
    int32_t  start_row, end_row;     /* rayTraceStartRow / rayTraceEndRow, inclusive (:135-136) */
    int32_t  sub_pixel_res;          /* rayTraceSubPixelRes (:90)                               */
    uint32_t background_argb;        /* BackgroundColor (:308); alpha forced to 0xFF on miss    */
    uint32_t flags;                  /* ORC_F_* below                                           */
    int32_t  random_seed;            /* rayTraceRandomSeed (:92) -> area-light offsets          */
    int32_t  shadow_samples;         /* 0 => 100 (ShadowMethod.cs:9); 1 => hard shadow variant  */
    int32_t  trace_mode;             /* ORC_MODE_*                                              */
    int32_t  strip_rows, strip_count, strip_index; /* multi-GPU row interleave; 0 => off        */
    int32_t  max_bounces;            /* config-5 extension (mirror reflection); 0 => off        */
    int32_t  concurrency;            /* rayTraceConcurrency, <= 0 => 4; read with ORC_F_STATIC_SHADOWS */
    int32_t  reserved0;
    double   transform[12];          /* Instance._transform rows 0..2 (Instance.cs:134)         */
    double   inv_transform[12];      /* Instance._inverseTransform rows 0..2 (Instance.cs:135)  */
    double   position_z;             /* Instance.Position.z                                     */
    double   fov_depth;              /* Renderer.fieldOfViewDepth (:101)                        */
    double   focal_depth, focal_blur_strength;        /* (:87-88)                               */
    double   ambient, shininess;                      /* (:38,:41)                              */
    double   light_dir_view[3], light_pos_view[3];    /* (:39-40)                               */
    double   reflectivity;           /* config-5 extension                                      */
    const double* area_light_offsets;/* optional [shadow_samples][3]; NULL => derive from seed  */
} orc_frame;

enum {
    ORC_F_SHADING     = 1u << 0,     /* rayTraceShading      */
    ORC_F_SHADOWS     = 1u << 1,     /* rayTraceShadows (dynamic)  */
    ORC_F_FOCAL_BLUR  = 1u << 2,     /* rayTraceFocalBlur    */
    ORC_F_POINT_LIGHT = 1u << 3,     /* pointLighting        */
    ORC_F_SPECULAR    = 1u << 4,     /* specularLighting     */
    ORC_F_STATIC_SHADOWS = 1u << 5   /* rayTraceShadowsStatic (with ORC_F_SHADOWS): 128^3 cache kept by the scene, filled in
                                        the lock-step row-block order defined in orc_render */
};
enum {
    ORC_MODE_REF_TREE = 0,  /* SpatialSubdivision.IntersectRay, literal                          */
    ORC_MODE_BRUTE    = 1,  /* GeometryCollection over all triangles (geometry_simple), literal  */
    ORC_MODE_NEAREST  = 2   /* clip to root box like the tree, then global nearest hit inside the
                               root box, ties -> lowest index: the semantics of the GPU's own BVH */
};

orc_scene* orc_scene_new(void);
void       orc_scene_free(orc_scene*);
/* triangles = MakeRayTracableGeometry_simple output (Renderer.cs:1452-1469): v9 = [n][3][3] */
int  orc_scene_set_triangles(orc_scene*, const double* v9, const uint32_t* argb, int64_t n,
                             const double box_min[3], const double box_max[3]);
int  orc_scene_set_extra(orc_scene*, const orc_prim* prims, int32_t n);
/* SpatialSubdivision ctor (SpatialSubdivision.cs:267-315). 0 ok; -2 vertex outside the box */
int  orc_scene_build_tree(orc_scene*, int32_t max_depth, int32_t max_per_leaf);
/* out = TreeDepth, NumNodes, NumLeafNodes, NumInternalNodes */
void orc_scene_tree_stats(const orc_scene*, int32_t out[4]);
void orc_scene_reset_shadow_cache(orc_scene*);   /* a new Renderer / ShadowMethod: empty static shadow cache */

/* Renderer.Render() for one Instance (raytrace path).  pixels = int[W*H] ARGB (strip-compact if
 * strip_count>0).  stats = rays fired, geometry tests, node visits, leaf visits (primary rays;
 * deterministic sums, not the reference's racy per-block counters).  threads<=0 => 1. */
int  orc_render(const orc_scene*, const orc_frame*, int32_t* pixels, uint64_t stats[4], int32_t threads);
/* orc_render restricted to columns [col_begin, col_end) of the frame's rows (the other pixels are left untouched): the CPU
 * baseline's centred crop (bench.py).  Test infrastructure only, like everything in oracle/. */
int  orc_render_window(const orc_scene*, const orc_frame*, int32_t* pixels, uint64_t stats[4], int32_t threads,
                       int32_t col_begin, int32_t col_end);

/* ShadingMethod.IntersectRay's colour step (ShadingMethod.cs:36-68, CalcLighting :110-177) for n recorded intersections:
 * out[i] = ModulatePackedColor(color[i], (byte)(255 * intensity(pos[i], normal[i]))) with the frame's light / transform. */
int  orc_shade_points(const orc_frame*, int64_t n, const double* pos, const double* normal, const uint32_t* color,
                      uint32_t* out, int32_t threads);

/* IRayIntersectable.IntersectRay in batch (IRayIntersectable.cs:31-48).
 * target: 0 = triangles brute (GeometryCollection), 1 = tree, 2 = root geometry of the chain
 * (extra geometry + tree, Renderer.cs:1545-1549), 3 = ORC_MODE_NEAREST semantics.
 * outputs (may be NULL): hit[n] (0/1), ray_frac[n], pos[n][3], normal[n][3], color[n], tri_index[n],
 * counters[n][3] = geometry tests, nodes visited, leaf nodes visited. */
int  orc_trace(const orc_scene*, int32_t target, int64_t n, const double* starts, const double* dirs,
               uint8_t* hit, double* ray_frac, double* pos, double* normal, uint32_t* color,
               int32_t* tri_index, int32_t* counters);

/* Instance.InitRender matrices (Instance.cs:134-135, Matrix.cs:74-169) */
void orc_instance_matrices(const double position[3], double yaw, double pitch, double roll,
                           double transform[12], double inv_transform[12]);
/* Renderer.fieldOfViewDepth (Renderer.cs:97-101) */
double orc_default_fov_depth(void);
/* ShadowMethod ctor offsets (ShadowMethod.cs:63-73) */
void orc_area_light_offsets(int32_t seed, int32_t count, double* out3);

/* ---- Model.Load3ds + PostProcessGeometry (Model.cs:522-653,750-831; 3dsLoader/) ---- */
typedef struct orc_model orc_model;
orc_model* orc_model_load_3ds(const uint8_t* data, size_t len, char* err, size_t errlen);
void       orc_model_free(orc_model*);
int64_t    orc_model_num_triangles(const orc_model*);
int64_t    orc_model_num_vertices(const orc_model*);
/* v9[n][9], argb[n] (= Surface.PackColorAndAlpha(diffuse,1.0)), min/max = Model.Min/Max */
void       orc_model_get(const orc_model*, double* v9, uint32_t* argb, double bmin[3], double bmax[3]);

#ifdef __cplusplus
}
#endif
#endif
