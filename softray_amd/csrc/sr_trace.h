// sr_trace.h -- device-side ray/primitive/tree arithmetic shared by every kernel (device code only).
// See sr_kernels.hip for the list of reference functions restated here.
#pragma once
#include "sr_device.h"

#include <float.h>

namespace sr {

// --------------------------------------------------------------------------------------------------
// FP64 3-vectors in the reference's operand order (Engine3D/Vector.cs)
// --------------------------------------------------------------------------------------------------
struct D3 {
    double x, y, z;
};
__device__ __forceinline__ D3 mk(double x, double y, double z) { D3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ D3 operator+(D3 a, D3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ D3 operator-(D3 a, D3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ D3 operator*(D3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
__device__ __forceinline__ D3 operator*(double s, D3 a) { return mk(s * a.x, s * a.y, s * a.z); }
__device__ __forceinline__ D3 neg(D3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ double dot(D3 a, D3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double length(D3 v) { return sqrt(v.x * v.x + v.y * v.y + v.z * v.z); }
__device__ __forceinline__ D3 normalise(D3 v) {            // Vector.Normalise: multiply by 1/len
    double inv = 1.0 / length(v);
    return mk(v.x * inv, v.y * inv, v.z * inv);
}
__device__ __forceinline__ double comp(D3 v, int axis) { return axis == 0 ? v.x : (axis == 1 ? v.y : v.z); }

// C# unchecked (byte)(double): truncate toward zero, keep the low 8 bits; NaN / huge -> 0 (x64 cvttsd2si)
__device__ __forceinline__ uint32_t to_byte(double d) {
    if (!(d > -2147483649.0 && d < 2147483648.0)) return 0u;
    return (uint32_t)((int32_t)d) & 0xffu;
}
// Color.ModulatePackedColor (Engine3D/Color.cs:124-133)
__device__ __forceinline__ uint32_t modulate(uint32_t color, uint32_t amount) {
    uint32_t r = (color >> 16) & 0xffu, g = (color >> 8) & 0xffu, b = color & 0xffu;
    r = ((r * amount) >> 8) & 0xffu;
    g = ((g * amount) >> 8) & 0xffu;
    b = ((b * amount) >> 8) & 0xffu;
    return (255u << 24) + (r << 16) + (g << 8) + b;
}

struct Hit {
    double   t;
    D3       pos, nrm;
    uint32_t color;
    int32_t  tri;
};
struct Ctr {
    uint32_t geom, nodes, leaves, rays;
};

// --------------------------------------------------------------------------------------------------
// primitives
// --------------------------------------------------------------------------------------------------
// Plane.IntersectRay on p[0..3] = {unit normal, originDist}
__device__ __forceinline__ bool plane_hit(const double* p, D3 s, D3 d, double& t, D3& pos) {
    double startDist = s.x * p[0] + s.y * p[1] + s.z * p[2];
    double dirDist = d.x * p[0] + d.y * p[1] + d.z * p[2];
    if (dirDist >= 0.0) return false;                      // one-sided
    double rf = p[3] - startDist;
    if (!(rf <= 0.0)) return false;
    rf = rf / dirDist;
    pos = s + d * rf;
    t = rf;
    return true;
}
// Triangle.IntersectRay on the 15-double record of sr_types.h
__device__ __forceinline__ bool tri_hit(const double* p, D3 s, D3 d, double& t, D3& pos) {
    double rf; D3 q;
    if (!plane_hit(p, s, d, rf, q)) return false;
    D3 w = mk(q.x - p[4], q.y - p[5], q.z - p[6]);
    double sv = (w.x * p[7] + w.y * p[8] + w.z * p[9]) / p[10];
    if (sv < 0.0 || sv > 1.0) return false;
    double tv = (w.x * p[11] + w.y * p[12] + w.z * p[13]) / p[14];
    if (sv >= 0.0 && tv >= 0.0 && sv + tv <= 1.0) { t = rf; pos = q; return true; }
    return false;
}
// Sphere.IntersectRay on p[0..4] = {centre, radius, radiusSqr}; rayFrac is a DISTANCE (dir is normalised)
__device__ __forceinline__ bool sphere_hit(const double* p, D3 s, D3 d, double& t, D3& pos, D3& nrm) {
    d = normalise(d);
    D3 oc = mk(s.x - p[0], s.y - p[1], s.z - p[2]);
    double b = dot(oc, d);
    if (b > p[3]) return false;
    double distSqr = oc.x * oc.x + oc.y * oc.y + oc.z * oc.z;
    double term = b * b - distSqr + p[4];
    if (term < 1e-10) return false;
    double q = sqrt(term);
    double f1 = -b - q, f2 = -b + q;
    double rf = (f1 >= 0 ? f1 : f2);
    if (rf < 0) return false;
    t = rf;
    pos = s + d * rf;
    nrm = normalise(mk(pos.x - p[0], pos.y - p[1], pos.z - p[2]));
    return true;
}

// AxisAlignedBox.IntersectRay (AxisAlignedBox.cs:60-95) on p = {min, max, originDist of the six planes in the constructor's
// order -x, -y, -z, +x, +y, +z (:22-27)}: the nearest of the six one-sided plane hits that lies on the box (ContainsPoint, 1e-10 slack)
__device__ __forceinline__ bool box_hit(const double* p, D3 s, D3 d, double& t, D3& pos, D3& nrm) {
    double best = DBL_MAX;
    bool any = false;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const double sg = i < 3 ? -1.0 : 1.0;
        const double pl[4] = {(i % 3) == 0 ? sg : 0.0, (i % 3) == 1 ? sg : 0.0, (i % 3) == 2 ? sg : 0.0, p[6 + i]};
        double rf; D3 q;
        if (plane_hit(pl, s, d, rf, q) && rf < best) {
            if (p[0] - 1e-10 < q.x && q.x < p[3] + 1e-10 && p[1] - 1e-10 < q.y && q.y < p[4] + 1e-10 && p[2] - 1e-10 < q.z && q.z < p[5] + 1e-10) {
                best = rf; pos = q; nrm = mk(pl[0], pl[1], pl[2]); any = true;
            }
        }
    }
    t = best;
    return any;
}

// one element of ExtraGeometryToRaytrace (Rec128.aux: 0 Sphere, 1 Plane, 2 Triangle, 4 AxisAlignedBox); returns its NumRayTests in `tests`
__device__ __forceinline__ bool extra_hit(const Rec128* r, D3 s, D3 d, double& t, D3& pos, D3& nrm, uint32_t& tests) {
    const int kind = r->aux;
    tests = 1;
    if (kind == 0) return sphere_hit(r->p, s, d, t, pos, nrm);
    if (kind == 4) { tests = 6; return box_hit(r->p, s, d, t, pos, nrm); }          // six planes (AxisAlignedBox.cs:70)
    nrm = mk(r->p[0], r->p[1], r->p[2]);
    if (kind == 1) return plane_hit(r->p, s, d, t, pos);
    return tri_hit(r->p, s, d, t, pos);
}

// --------------------------------------------------------------------------------------------------
// root-box clip (AxisAlignedBox.ContainsPoint / IntersectLineSegment / ClipLineSegment)
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool inside(const double* lo, const double* hi, D3 p) {
    return lo[0] < p.x && p.x < hi[0] && lo[1] < p.y && p.y < hi[1] && lo[2] < p.z && p.z < hi[2];
}
// dot with the unit normal of box plane I.  The reference evaluates the 3-term product x*nx + y*ny + z*nz with
// two of the normal's components 0.0 and one +-1.0; for finite coordinates that is exactly +-component (the 0.0
// products can only change the sign of a zero result, which no later comparison, division or sum can observe:
// every consumer divides BY (ed - sd), compares, or adds the value to other terms).
template <int I>
__device__ __forceinline__ double plane_dot(D3 v) {
    return (I == 0) ? -v.x : (I == 1) ? -v.y : (I == 2) ? -v.z : (I == 3) ? v.x : (I == 4) ? v.y : v.z;
}
template <int I>
__device__ __forceinline__ void seg_plane(const RootBox& rb, D3 a, D3 b, double& closest, D3& cpos) {
    double sd = plane_dot<I>(a), ed = plane_dot<I>(b);
    double f = (rb.pd[I] - sd) / (ed - sd);                // Plane.IntersectLineSegment, Plane.cs:111-138
    if (0.0 <= f && f <= 1.0) {
        D3 p = a + (b - a) * f;
        if (f < closest && inside(rb.lo, rb.hi, p)) { closest = f; cpos = p; }
    }
}
__device__ __forceinline__ bool box_segment(const RootBox& rb, D3 a, D3 b, D3& out) {
    double closest = DBL_MAX;
    D3 cpos = mk(0, 0, 0);
    seg_plane<0>(rb, a, b, closest, cpos);
    seg_plane<1>(rb, a, b, closest, cpos);
    seg_plane<2>(rb, a, b, closest, cpos);
    seg_plane<3>(rb, a, b, closest, cpos);
    seg_plane<4>(rb, a, b, closest, cpos);
    seg_plane<5>(rb, a, b, closest, cpos);
    if (closest == DBL_MAX) return false;
    out = cpos;
    return true;
}

// AxisAlignedBox.ClipLineSegment.  NEED_END = false: the caller never looks at `end` again (every walk except the
// reference tree's), so the second clip is not evaluated.
// (An "entry face only" shortcut -- three planes instead of six, with margin checks and this routine as fallback --
// was measured on MI355X: bit-identical but not faster; the six independent plane evaluations overlap well.)
template <bool NEED_END = true>
__device__ __forceinline__ bool clip_segment(const RootBox& rb, D3& start, D3& end) {
    bool si = inside(rb.lo, rb.hi, start), ei = inside(rb.lo, rb.hi, end);
    if (si && ei) return true;
    D3 ip;
    if (!box_segment(rb, start, end, ip)) return false;
    if (si) { end = ip; return true; }
    D3 original = start;
    start = ip;
    if (NEED_END && !ei) {
        if (box_segment(rb, end, original, ip)) end = ip;
    }
    return true;
}

// a lane's pending leaf in one word: first record | count << kLeafShift (<= 15 triangles per leaf, < 2^27 records)
constexpr int kLeafShift = 27;
constexpr int32_t kLeafMask = (1 << kLeafShift) - 1;

// Per-lane traversal stack in LDS, laid out [level][thread]: consecutive lanes hit consecutive banks.
struct Stack {
    int32_t* base;      // &lds[threadIdx]
    int32_t  stride;    // blockDim
    __device__ __forceinline__ void put(int level, int32_t v) { base[level * stride] = v; }
    __device__ __forceinline__ int32_t get(int level) const { return base[level * stride]; }
};

// --------------------------------------------------------------------------------------------------
// SpatialSubdivision.IntersectRay -- literal traversal of the reference tree
// --------------------------------------------------------------------------------------------------
__device__ bool ref_tree_intersect(const DevScene& sc, const Rec128* tris, Stack st, D3 s, D3 d, Hit& out, Ctr& c) {
    D3 end = s + d * 10000.0;
    D3 original = s;
    if (!clip_segment(sc.root, s, end)) return false;
    double offset = length(original - s) / length(d);       // originalStart.Distance(start) / dir.Length

    int sp = 0;
    st.put(sp++, 0);
    while (sp > 0) {
        int32_t ni = st.get(--sp);
        if (ni < 0) continue;                                // RecursiveRayTrace(null) -> null
        c.nodes++;
        const RefNode n = sc.rnodes[ni];
        if (n.axis < 0) {                                    // leaf: GetClosestIntersection
            c.leaves++;
            const LeafBox* lb = &sc.rboxes[n.box];
            double best = DBL_MAX;
            int32_t bestTri = -1;
            D3 bestPos = mk(0, 0, 0);
            for (int k = 0; k < n.b; ++k) {
                int32_t ti = sc.rleaf[n.a + k];
                const Rec128* r = &tris[ti];
                double t; D3 pos;
                if (tri_hit(r->p, s, d, t, pos) && t < best) {
                    if (inside(lb->lo, lb->hi, pos)) { best = t; bestTri = ti; bestPos = pos; }
                }
                c.geom++;
            }
            if (best < DBL_MAX) {
                const Rec128* r = &tris[bestTri];
                out.t = best + offset;
                out.pos = bestPos;
                out.nrm = mk(r->p[0], r->p[1], r->p[2]);
                out.color = r->color;
                out.tri = r->aux;
                return true;
            }
            continue;
        }
        bool sN = comp(s, n.axis) >= n.split;                // Point.IntersectPlane, Point.cs:35-50
        bool eN = comp(end, n.axis) >= n.split;
        int32_t nearC = sN ? n.a : n.b, farC = sN ? n.b : n.a;
        if (eN != sN) st.put(sp++, farC);
        st.put(sp++, nearC);
    }
    return false;
}

// --------------------------------------------------------------------------------------------------
// GeometryCollection over all model triangles (rayTraceSubdivision = false)
// --------------------------------------------------------------------------------------------------
template <bool ANY>
__device__ bool brute_intersect(const Rec128* tris, int ntris, D3 s, D3 d, Hit& out, Ctr& c) {
    double best = DBL_MAX;
    int32_t bestK = -1;
    D3 bestPos = mk(0, 0, 0);
    for (int k = 0; k < ntris; ++k) {
        double t; D3 pos;
        c.geom++;
        if (tri_hit(tris[k].p, s, d, t, pos) && t < best) {
            best = t; bestK = k; bestPos = pos;
            if (ANY && t <= 1.0) { out.t = t; out.tri = k; return true; }
        }
    }
    if (bestK < 0) return false;
    out.t = best;
    out.pos = bestPos;
    out.nrm = mk(tris[bestK].p[0], tris[bestK].p[1], tris[bestK].p[2]);
    out.color = tris[bestK].color;
    out.tri = tris[bestK].aux;
    return true;
}

// --------------------------------------------------------------------------------------------------
// Own BVH: fp32 conservative box culling, FP64 reference triangle arithmetic at the leaves.
// Result = nearest hit (clipped start, hit inside the root box, ties -> lowest TriangleIndex); with ANY
// it answers "is there a hit with rayFrac <= 1.0" (the only thing ShadowMethod.cs:170 asks).
// --------------------------------------------------------------------------------------------------
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 pk_fma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { f2 r = {v, v}; return r; }
// v_pk_fma_f32 with one half of a register pair broadcast to both lanes by the instruction's own operand selects (the compiler
// copies an odd register into a fresh pair first when the scalar comes from the high half of a loaded vector):
//   pk_fma_lo(P, b, c) = (P.x, P.x) * b + c      pk_fma_hi(P, b, c) = (P.y, P.y) * b + c
//   pk_fma_hi_addlo(P, b, C) = (P.y, P.y) * b + (C.x, C.x)
// bits = 2 * bits + (t > 0 ? 0 : 1): compare into the carry, add with carry (an undecided verdict -- also a NaN -- shifts in a one)
__device__ __forceinline__ uint32_t shift_in_not_positive(uint32_t bits, float t) {
    asm("v_cmp_nlt_f32 vcc, 0, %1\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(bits) : "v"(t) : "vcc");
    return bits;
}
__device__ __forceinline__ f2 pk_fma_lo(f2 P, f2 b, f2 c) {
    f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[0,1,1]" : "=v"(d) : "v"(P), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f2 pk_fma_hi(f2 P, f2 b, f2 c) {
    f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "=v"(d) : "v"(P), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ f2 pk_fma_hi_addlo(f2 P, f2 b, f2 C) {
    f2 d;
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,0]" : "=v"(d) : "v"(P), "v"(b), "v"(C));
    return d;
}

// Slab test of both children of a node as packed FMAs (v_pk_fma_f32: two floats per issue slot).  Per axis
// t = plane * inv + bias with bias = -origin * inv (walks that inflate the boxes add -+r * inv); the node's twelve
// floats are consumed in memory order as pairs (lo.x, lo.y) (lo.z, hi.x) (hi.y, hi.z), so I01 = (ix, iy),
// I20 = (iz, ix), I12 = (iy, iz) and B0/B1/B2 are the matching bias pairs.  Rounding differs from (plane - origin) * inv
// by at most a few ulp of |origin * inv|, far inside the 2^-16 * extent padding the boxes carry.  A zero direction
// component must be given a huge FINITE reciprocal (slab_inv) so that no inf - inf appears.
__device__ __forceinline__ float slab_inv(float d) { return d != 0.0f ? 1.0f / d : 1e30f; }
__device__ __forceinline__ void node_slabs(const BvhNode& n, f2 I01, f2 I20, f2 I12, f2 B0, f2 B1, f2 B2,
                                           float& a0, float& b0, float& a1, float& b1) {
    {
        const f2 T0 = pk_fma((f2){n.lo0[0], n.lo0[1]}, I01, B0), T1 = pk_fma((f2){n.lo0[2], n.hi0[0]}, I20, B1), T2 = pk_fma((f2){n.hi0[1], n.hi0[2]}, I12, B2);
        // fminf/fmaxf drop a NaN operand: conservative
        a0 = fmaxf(fmaxf(fminf(T0.x, T1.y), fminf(T0.y, T2.x)), fminf(T1.x, T2.y));
        b0 = fminf(fminf(fmaxf(T0.x, T1.y), fmaxf(T0.y, T2.x)), fmaxf(T1.x, T2.y));
    }
    {
        const f2 T0 = pk_fma((f2){n.lo1[0], n.lo1[1]}, I01, B0), T1 = pk_fma((f2){n.lo1[2], n.hi1[0]}, I20, B1), T2 = pk_fma((f2){n.hi1[1], n.hi1[2]}, I12, B2);
        a1 = fmaxf(fmaxf(fminf(T0.x, T1.y), fminf(T0.y, T2.x)), fminf(T1.x, T2.y));
        b1 = fminf(fminf(fmaxf(T0.x, T1.y), fmaxf(T0.y, T2.x)), fmaxf(T1.x, T2.y));
    }
}

// --------------------------------------------------------------------------------------------------
// fp32 pre-test of one (ray, triangle) pair for the PRIVATE walks (one ray per lane: k_shadow_rays, k_bounce, bvh_intersect):
// "Triangle.IntersectRay must fail, or its hit cannot matter", decided from the triangle's 64-byte TriSlab record (unit plane
// normal n, offset; three inward unit edge normals m_k, offsets; root-centre-relative; every value the rounding of its FP64
// source) before the 128-byte FP64 record is touched.  With G0 = n.s - d (s = the clipped start), g1 = n.dir, K0_k = m_k.s - c_k,
// K1_k = m_k.dir the crossing is at t_c = G0 / (-g1) and F_k there is B_k / (-g1), B_k = G0 K1_k - K0_k g1.  The reference's test
// (Triangle.cs:83-104, Plane.cs:52-75) needs dirDist = n.dir < 0, originDist - startDist <= 0 (G(s) >= 0) and the crossing inside
// the three edges; a caller with a distance limit (nearest hit so far / rayFrac <= 1) needs t_c below it.  Error bounds as in
// k_shadow_cls (u = 2^-24): |G0|, |K0_k| errors < a0 = 12 u s0, |g1|, |K1_k| errors < a1 = 20 u |dir|, |B_k| error
// < mc = |dir| (u (32 s0 + 32 (max |K0_k| + |G0|)) + 1e-9).  REJECT (the FP64 test cannot produce a hit that counts) iff
//   g1 > 16 a1  (back-facing)   or   G0 < -a0  (start behind the plane)   or, with g1 <= -16 a1 (surely front-facing),
//   min_k B_k < -mc  (outside an edge by > 1e-6 of the scene)   or   G0 + tlim g1 > a0 + tlim a1  (crossing beyond the limit).
// FP64 noise is nine orders of magnitude below these margins.  A degenerate triangle's record is all zeros: never rejected.
// --------------------------------------------------------------------------------------------------
struct RayF {
    f2    sdx, sdy, sdz;      // per axis (start, direction), fp32, root-centre-relative start
    float a0, a1, glo, c0, c1;
};
__device__ __forceinline__ RayF make_ray_f(const DevScene& sc, D3 s, D3 d) {
    RayF r;
    const float ex = (float)(s.x - sc.root.centre[0]), ey = (float)(s.y - sc.root.centre[1]), ez = (float)(s.z - sc.root.centre[2]);
    const float dx = (float)d.x, dy = (float)d.y, dz = (float)d.z;
    r.sdx = (f2){ex, dx}; r.sdy = (f2){ey, dy}; r.sdz = (f2){ez, dz};
    const float bx = (float)(sc.root.max[0] - sc.root.min[0]), by = (float)(sc.root.max[1] - sc.root.min[1]), bz = (float)(sc.root.max[2] - sc.root.min[2]);
    const float s0 = (__builtin_amdgcn_sqrtf(bx * bx + by * by + bz * bz) * 0.5f + fabsf(ex) + fabsf(ey) + fabsf(ez)) * 1.002f + 0.004f;
    const float dmax = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * 1.0001f + 1e-30f;
    const float u = 5.9604645e-8f;
    r.a0 = 12.0f * u * s0;
    r.a1 = 20.0f * u * dmax;
    r.glo = 16.0f * r.a1;
    r.c0 = dmax * (32.0f * u * s0 + 1e-9f);
    r.c1 = 32.0f * u * dmax;
    return r;
}
__device__ __forceinline__ bool slab_rejects(const TriSlab& t, const RayF& r, float tlim) {
    const f2 cn = {-t.d, 0.0f}, c1 = {-t.c1, 0.0f}, c2 = {-t.c2, 0.0f}, c3 = {-t.c3, 0.0f};
    const f2 N = pk_fma(splat(t.n[0]), r.sdx, pk_fma(splat(t.n[1]), r.sdy, pk_fma(splat(t.n[2]), r.sdz, cn)));
    const float G0 = N.x, g1 = N.y;
    if (g1 > r.glo || G0 < -r.a0) return true;
    if (!(g1 <= -r.glo)) return false;                                   // grazing: the FP64 test decides
    const f2 P = pk_fma(splat(t.m1[0]), r.sdx, pk_fma(splat(t.m1[1]), r.sdy, pk_fma(splat(t.m1[2]), r.sdz, c1)));
    const f2 Q = pk_fma(splat(t.m2[0]), r.sdx, pk_fma(splat(t.m2[1]), r.sdy, pk_fma(splat(t.m2[2]), r.sdz, c2)));
    const f2 T = pk_fma(splat(t.m3[0]), r.sdx, pk_fma(splat(t.m3[1]), r.sdy, pk_fma(splat(t.m3[2]), r.sdz, c3)));
    const float B1 = __builtin_fmaf(G0, P.y, -(P.x * g1)), B2 = __builtin_fmaf(G0, Q.y, -(Q.x * g1)), B3 = __builtin_fmaf(G0, T.y, -(T.x * g1));
    const float kmax = fmaxf(fmaxf(fabsf(P.x), fabsf(Q.x)), fabsf(T.x));
    const float mc = __builtin_fmaf(r.c1, kmax + fabsf(G0), r.c0);
    if (fminf(fminf(B1, B2), B3) < -mc) return true;
    return tlim < 3.0e38f && __builtin_fmaf(tlim, g1, G0) > __builtin_fmaf(tlim, r.a1, r.a0) * 1.0001f;
}
// the triangles of one leaf (<= 15 records from `first`): bit k set = record first + k survives the pre-test
__device__ __forceinline__ uint32_t leaf_survivors(const DevScene& sc, int32_t first, int32_t cn, const RayF& rf, float tlim) {
    uint32_t m = 0u;
    for (int k = 0; k < cn; ++k)
        if (!slab_rejects(sc.bslab[first + k], rf, tlim)) m |= 1u << k;
    return m;
}

template <bool ANY>
__device__ bool bvh_intersect(const DevScene& sc, Stack st, D3 s, D3 d, Hit& out, Ctr& c) {
    D3 end = s + d * 10000.0;
    D3 original = s;
    if (!clip_segment<false>(sc.root, s, end)) return false;
    double offset = length(original - s) / length(d);

    const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
    const float ix = slab_inv((float)d.x), iy = slab_inv((float)d.y), iz = slab_inv((float)d.z);
    const f2 I01 = {ix, iy}, I20 = {iz, ix}, I12 = {iy, iz};
    const f2 B0 = {-ox * ix, -oy * iy}, B1 = {-oz * iz, -ox * ix}, B2 = {-oy * iy, -oz * iz};
    const float kInfl = 1.0f + 9.5367431640625e-7f;          // 1 + 2^-20
    float tlim = FLT_MAX;
    if (ANY) {
        // occluder <=> fl(t + offset) <= 1.0 ; nothing beyond t = 1 - offset (inflated) can qualify
        double lim = 1.0 - offset;
        if (lim < 0.0) return false;
        tlim = (float)lim * kInfl + 1e-30f;
    }
    double best = DBL_MAX;
    int32_t bestIdx = 0x7fffffff, bestK = -1;
    const RayF rf = make_ray_f(sc, s, d);

    // "while-while" traversal: a lane first walks inner nodes until it owns a pending leaf (or is finished), and only
    // then the (long, FP64) triangle tests run -- so a wavefront executes the leaf code with most lanes busy instead of
    // once per node step for the few lanes that happen to be at a leaf.
    int sp = 0;
    int32_t ni = 0;                  // next inner node, -1: walk finished
    int32_t leafA = -1, leafB = -1;  // pending leaves: first record | count << kLeafShift
    for (;;) {
        while (ni >= 0 && leafA < 0) {
            const BvhNode n = sc.bnodes[ni];
            c.nodes++;
            float t0, x0, t1, x1;
            node_slabs(n, I01, I20, I12, B0, B1, B2, t0, x0, t1, x1);
            const bool h0 = n.n0 >= 0 && t0 <= x0 && x0 >= 0.0f && t0 <= tlim;
            const bool h1 = n.n1 >= 0 && t1 <= x1 && x1 >= 0.0f && t1 <= tlim;
            const bool l0 = h0 && n.n0 > 0, l1 = h1 && n.n1 > 0;
            if (l0 && l1) {                                   // nearer leaf first
                const bool first0 = t0 <= t1;
                leafA = (first0 ? n.c0 : n.c1) | ((first0 ? n.n0 : n.n1) << kLeafShift);
                leafB = (first0 ? n.c1 : n.c0) | ((first0 ? n.n1 : n.n0) << kLeafShift);
            } else if (l0) leafA = n.c0 | (n.n0 << kLeafShift);
            else if (l1) leafA = n.c1 | (n.n1 << kLeafShift);
            const bool i0 = h0 && n.n0 == 0, i1 = h1 && n.n1 == 0;
            if (i0 && i1) {
                const bool first0 = t0 <= t1;
                st.put(sp++, first0 ? n.c1 : n.c0);           // the far child is re-tested against tlim when it is popped
                ni = first0 ? n.c0 : n.c1;
            } else if (i0) ni = n.c0;
            else if (i1) ni = n.c1;
            else ni = (sp > 0) ? st.get(--sp) : -1;
        }
        if (leafA < 0) break;
        while (leafA >= 0) {
            const int32_t first = leafA & kLeafMask, cn = (leafA >> kLeafShift) & 15;
            leafA = leafB;
            leafB = -1;
            c.leaves++;
            // fp32 pre-test on the 64-byte records first (slab_rejects): the FP64 record is fetched for the survivors only
            for (uint32_t m = leaf_survivors(sc, first, cn, rf, tlim); m; m &= m - 1u) {
                const int k = first + (__ffs((int)m) - 1);
                const Rec128* r = &sc.btris[k];
                double t; D3 pos;
                c.geom++;
                if (tri_hit(r->p, s, d, t, pos) && inside(sc.root.lo, sc.root.hi, pos)) {
                    if (ANY) {
                        if (t + offset <= 1.0) { out.t = t + offset; out.tri = k; return true; }   // out.tri = record position: the caller's blocker cache
                    } else {
                        const int32_t idx = r->aux;
                        if (t < best || (t == best && idx < bestIdx)) {
                            best = t; bestIdx = idx; bestK = k;
                            tlim = (float)best * kInfl + 1e-30f;
                        }
                    }
                }
            }
        }
    }
    if (ANY || bestK < 0) return false;
    const Rec128* r = &sc.btris[bestK];
    out.t = best + offset;
    out.pos = s + d * best;          // the expression plane_hit evaluated for the winning triangle (pos = start + dir * rayFrac)
    out.nrm = mk(r->p[0], r->p[1], r->p[2]);
    out.color = r->color;
    out.tri = r->aux;
    return true;
}

// --------------------------------------------------------------------------------------------------
// The same nearest-hit search for the 64 rays of one wavefront that share their ORIGIN (the camera rays of an 8x8-pixel
// tile): the wave walks the BVH once -- shared stack of node indices in LDS (wave-uniform), node / triangle records
// fetched with scalar loads, near child first by a vote -- and every lane tests its own ray against the broadcast boxes
// and triangles.  Before the FP64 Triangle.IntersectRay a lane consults the triangle's camera-cone record (CamCone,
// sr_types.h): if the ray is back-facing or outside one of the three cone planes by more than the fp32 error bound the
// triangle cannot be hit (every condition of Triangle.cs:83-104 that fails does so by ~1e-7 relative, the FP64 noise is
// 1e-16), so the exact test is skipped; everything that can become a hit runs the reference arithmetic unchanged.
// Result per lane == bvh_intersect<false> (a different but equally conservative visiting order; the nearest hit with
// the lowest-index tie-break does not depend on the order).  All 64 lanes must call it (live = false: no ray).
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool cone_rejects(const CamCone& cm, f2 dxx, f2 dyy, f2 dzz, float dl) {
    // (c1, c2) and (c3, n.d): two packed FMA chains
    const f2 c12 = pk_fma((f2){cm.w12x[0], cm.w12x[1]}, dxx, pk_fma((f2){cm.w12y[0], cm.w12y[1]}, dyy, (f2){cm.w12z[0], cm.w12z[1]} * dzz));
    const f2 c3n = pk_fma((f2){cm.w3nx[0], cm.w3nx[1]}, dxx, pk_fma((f2){cm.w3ny[0], cm.w3ny[1]}, dyy, (f2){cm.w3nz[0], cm.w3nz[1]} * dzz));
    const f2 m12 = (f2){cm.m12[0], cm.m12[1]} * splat(dl), m3n = (f2){cm.m3n[0], cm.m3n[1]} * splat(dl);
    // outside a cone plane: c_k < -m_k |d|;  back-facing: n.d > mn |d|  (dirDist >= 0, Plane.cs:60-61)
    return fminf(fminf(c12.x + m12.x, c12.y + m12.y), c3n.x + m3n.x) < 0.0f || c3n.y > m3n.y;
}

template <bool FILTER>
__device__ bool bvh_packet_nearest(const DevScene& sc, int32_t* wnode, bool live, D3 s, D3 d, Hit& out, Ctr& c) {
    D3 end = s + d * 10000.0;
    D3 original = s;
    bool act = live;
    if (act) act = clip_segment<false>(sc.root, s, end);
    const double offset = act ? length(original - s) / length(d) : 0.0;
    const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
    const float dfx = (float)d.x, dfy = (float)d.y, dfz = (float)d.z;
    const float ix = slab_inv(dfx), iy = slab_inv(dfy), iz = slab_inv(dfz);
    const f2 I01 = {ix, iy}, I20 = {iz, ix}, I12 = {iy, iz};
    const f2 B0 = {-ox * ix, -oy * iy}, B1 = {-oz * iz, -ox * ix}, B2 = {-oy * iy, -oz * iz};
    const f2 dxx = splat(dfx), dyy = splat(dfy), dzz = splat(dfz);
    const float dl = sqrtf(dfx * dfx + dfy * dfy + dfz * dfz) * 1.000001f;
    const float kInfl = 1.0f + 9.5367431640625e-7f;          // 1 + 2^-20
    float tlim = FLT_MAX;
    double best = DBL_MAX;
    int32_t bestIdx = 0x7fffffff, bestK = -1;
    int sp = 0;                      // wave-uniform
    int32_t ni = 0;                  // wave-uniform
    if (__ballot(act) != 0ull) {
        for (;;) {
            const BvhNode n = sc.bnodes[ni];                           // wave-uniform address: scalar loads
            // counters of the packet walk are per WAVE (what the wave fetched), kept by its first participating lane:
            // nodes = 64-byte nodes, geom = 64-byte camera-cone records consulted, leaves = 128-byte FP64 records fetched
            if (act && (__ffsll((long long)__ballot(act)) - 1) == (int)(threadIdx.x & 63u)) c.nodes++;
            float t0, x0, t1, x1;
            node_slabs(n, I01, I20, I12, B0, B1, B2, t0, x0, t1, x1);
            const bool h0 = act && n.n0 >= 0 && t0 <= x0 && x0 >= 0.0f && t0 <= tlim;
            const bool h1 = act && n.n1 >= 0 && t1 <= x1 && x1 >= 0.0f && t1 <= tlim;
            const bool first0 = __popcll(__ballot(h0 && h1 && t0 <= t1)) >= __popcll(__ballot(h0 && h1 && t1 < t0));
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bool c0 = (t == 0) == first0;
                const int cn = c0 ? n.n0 : n.n1, cc = c0 ? n.c0 : n.c1;
                const bool hc = (c0 ? h0 : h1) && (c0 ? t0 : t1) <= tlim;       // tlim may have shrunk in the other leaf
                if (cn > 0 && __ballot(hc) != 0ull) {
                    const bool hc_first = hc && (__ffsll((long long)__ballot(hc)) - 1) == (int)(threadIdx.x & 63u);
                    for (int k = cc; k < cc + cn; ++k) {
                        bool cand = hc;
                        if (FILTER) {
                            const CamCone cm = sc.bcam[k];                       // scalar load
                            cand = hc && !cone_rejects(cm, dxx, dyy, dzz, dl);
                        }
                        if (hc_first) c.geom++;
                        if (__ballot(cand) != 0ull) {
                            const Rec128* r = &sc.btris[k];                       // wave-uniform address
                            if (cand && (__ffsll((long long)__ballot(cand)) - 1) == (int)(threadIdx.x & 63u)) c.leaves++;
                            if (cand) {
                                double tt; D3 pos;
                                if (tri_hit(r->p, s, d, tt, pos) && inside(sc.root.lo, sc.root.hi, pos)) {
                                    const int32_t idx = r->aux;
                                    if (tt < best || (tt == best && idx < bestIdx)) {
                                        best = tt; bestIdx = idx; bestK = k;
                                        tlim = (float)best * kInfl + 1e-30f;
                                    }
                                }
                            }
                        }
                    }
                }
            }
            const bool w0 = h0 && n.n0 == 0 && t0 <= tlim, w1 = h1 && n.n1 == 0 && t1 <= tlim;
            const bool any0 = __ballot(w0) != 0ull, any1 = __ballot(w1) != 0ull;
            if (any0 && any1) {
                wnode[sp++] = first0 ? n.c1 : n.c0;                    // (all lanes write the same word)
                ni = first0 ? n.c0 : n.c1;
            } else if (any0) ni = n.c0;
            else if (any1) ni = n.c1;
            else {
                if (sp == 0) break;
                ni = __builtin_amdgcn_readfirstlane(wnode[--sp]);
            }
        }
    }
    if (bestK < 0) return false;
    const Rec128* r = &sc.btris[bestK];
    out.t = best + offset;
    out.pos = s + d * best;          // the expression plane_hit evaluated for the winning triangle (pos = start + dir * rayFrac)
    out.nrm = mk(r->p[0], r->p[1], r->p[2]);
    out.color = r->color;
    out.tri = r->aux;
    return true;
}

// --------------------------------------------------------------------------------------------------
// The packet walk on the FOUR-WIDE tree (Bvh4Node, sr_types.h): one pair of scalar loads brings four children's boxes, so a
// tile's walk is about half as many dependent steps (and half the scalar bookkeeping) as on the binary tree.  The nodes come
// from the frame's camera-ordered copy (k_order_nodes): children are stored front to back for the rays' common origin, so
// there is no vote -- leaves are tested in slot order, the inner children that some lane still wants are pushed far to
// near and the nearest one is entered.  Per-lane arithmetic (slab tests, cone filter, FP64 triangle test) and therefore the
// result are those of bvh_packet_nearest / bvh_intersect<false>: the nearest hit with the lowest-index tie-break does not
// depend on the visiting order.  wnode: 3 * b4depth + 2 words.
// --------------------------------------------------------------------------------------------------
// KNOWN (bit a set): on axis a every ray of the frame that can reach the box meets the plane stored FIRST first -- the frame's ordered
// copy of the nodes holds (near, far) instead of (lo, hi) there (k_order_nodes: the common origin / end of the rays lies outside the
// root box's slab on that axis, so the sign of that direction component is the same for all of them) -- and the min / max of the two
// plane parameters is not computed: 2 instead of 8 min/max per child when all three axes are known.  A ray whose component has the
// other sign cannot reach any box of the scene; its "near" exceeds its "far" and the test fails, as it must.
template <int KNOWN = 0>
__device__ __forceinline__ void child_slabs(const Bvh4Child& ch, f2 I01, f2 I20, f2 I12, f2 B0, f2 B1, f2 B2, float& a, float& b) {
    const f2 T0 = pk_fma((f2){ch.lo[0], ch.lo[1]}, I01, B0), T1 = pk_fma((f2){ch.lo[2], ch.hi[0]}, I20, B1), T2 = pk_fma((f2){ch.hi[1], ch.hi[2]}, I12, B2);
    // fminf/fmaxf drop a NaN operand: conservative
    const float nx = (KNOWN & 1) ? T0.x : fminf(T0.x, T1.y), fx = (KNOWN & 1) ? T1.y : fmaxf(T0.x, T1.y);
    const float ny = (KNOWN & 2) ? T0.y : fminf(T0.y, T2.x), fy = (KNOWN & 2) ? T2.x : fmaxf(T0.y, T2.x);
    const float nz = (KNOWN & 4) ? T1.x : fminf(T1.x, T2.y), fz = (KNOWN & 4) ? T2.y : fmaxf(T1.x, T2.y);
    a = fmaxf(fmaxf(nx, ny), nz);
    b = fminf(fminf(fx, fy), fz);
}
// A record at a WAVE-UNIFORM address, read through the constant address space: scalar loads (s_load_dwordxN through the scalar cache)
// whatever the compiler can or cannot prove about the stores around it.  Left to itself it only uses scalar loads for memory it can
// show is not written between the kernel's entry and the load; inside a loop that also stores (the persistent tile loops of
// k_primary / k_shaft_pkt4) that proof fails and a 128-byte node arrives as vector loads + 32 v_readfirstlane.  Only for memory the
// kernel never writes (tree nodes, TriSlab / CamCone records).
template <class T>
__device__ __forceinline__ T load_uniform(const T* p) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "whole 32-bit words");
    typedef const uint32_t __attribute__((address_space(4))) CW4;
    CW4* w = (CW4*)(const void*)p;
    T out;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) o[i] = w[i];
    return out;
}

// value k of four with a wave-uniform k (scalar selects / v_cndmask with scalar conditions: no indexed registers, no scratch)
template <class T> __device__ __forceinline__ T pick4(int k, T a, T b, T c, T d) { return k == 0 ? a : (k == 1 ? b : (k == 2 ? c : d)); }

template <bool FILTER, int KNOWN>
__device__ bool bvh4_packet_nearest(const DevScene& sc, int32_t* wnode, bool live, D3 s, D3 d, Hit& out, Ctr& c) {
    D3 end = s + d * 10000.0;
    D3 original = s;
    bool act = live;
    if (act) act = clip_segment<false>(sc.root, s, end);
    const double offset = act ? length(original - s) / length(d) : 0.0;
    const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
    const float dfx = (float)d.x, dfy = (float)d.y, dfz = (float)d.z;
    const float ix = slab_inv(dfx), iy = slab_inv(dfy), iz = slab_inv(dfz);
    const f2 I01 = {ix, iy}, I20 = {iz, ix}, I12 = {iy, iz};
    const f2 B0 = {-ox * ix, -oy * iy}, B1 = {-oz * iz, -ox * ix}, B2 = {-oy * iy, -oz * iz};
    const f2 dxx = splat(dfx), dyy = splat(dfy), dzz = splat(dfz);
    const float dl = sqrtf(dfx * dfx + dfy * dfy + dfz * dfz) * 1.000001f;
    const float kInfl = 1.0f + 9.5367431640625e-7f;          // 1 + 2^-20
    float tlim = FLT_MAX;
    double best = DBL_MAX;
    int32_t bestIdx = 0x7fffffff, bestK = -1;
    int sp = 0;                      // wave-uniform
    int32_t ni = 0;                  // wave-uniform
    const bool counter_lane = act && (__ffsll((long long)__ballot(act)) - 1) == (int)(threadIdx.x & 63u);
    if (__ballot(act) != 0ull) {
        for (;;) {
            const Bvh4Node n = load_uniform(&sc.b4cam[ni]);            // wave-uniform address: scalar loads
            if (counter_lane) c.nodes++;                               // (per WAVE: 128-byte nodes fetched)
            float t0, x0, t1, x1, t2, x2, t3, x3;
            child_slabs<KNOWN>(n.ch[0], I01, I20, I12, B0, B1, B2, t0, x0);
            child_slabs<KNOWN>(n.ch[1], I01, I20, I12, B0, B1, B2, t1, x1);
            child_slabs<KNOWN>(n.ch[2], I01, I20, I12, B0, B1, B2, t2, x2);
            child_slabs<KNOWN>(n.ch[3], I01, I20, I12, B0, B1, B2, t3, x3);
            const bool h0 = act && n.ch[0].n >= 0 && t0 <= x0 && x0 >= 0.0f, h1 = act && n.ch[1].n >= 0 && t1 <= x1 && x1 >= 0.0f;
            const bool h2 = act && n.ch[2].n >= 0 && t2 <= x2 && x2 >= 0.0f, h3 = act && n.ch[3].n >= 0 && t3 <= x3 && x3 >= 0.0f;
            // ---- leaf children in slot order (front to back for this origin).  One copy of the triangle loop per slot, everything
            //      static: picking a slot's count / link / lane mask by a run-time index costs a chain of scalar branches per slot ----
            const auto leaf = [&](const int cn, const int cc, const bool h, const float t) __attribute__((always_inline)) {
                const bool hc = h && t <= tlim;                                      // tlim may have shrunk in an earlier leaf
                if (__ballot(hc) == 0ull) return;
                const bool hc_first = hc && (__ffsll((long long)__ballot(hc)) - 1) == (int)(threadIdx.x & 63u);
                for (int q = cc; q < cc + cn; ++q) {
                    bool cand = hc;
                    if (FILTER) {
                        const CamCone cm = load_uniform(&sc.bcam[q]);                // scalar load
                        cand = hc && !cone_rejects(cm, dxx, dyy, dzz, dl);
                    }
                    if (hc_first) c.geom++;
                    if (__ballot(cand) != 0ull) {
                        const Rec128* r = &sc.btris[q];                               // wave-uniform address
                        if (cand && (__ffsll((long long)__ballot(cand)) - 1) == (int)(threadIdx.x & 63u)) c.leaves++;
                        if (cand) {
                            double tt; D3 pos;
                            if (tri_hit(r->p, s, d, tt, pos) && inside(sc.root.lo, sc.root.hi, pos)) {
                                const int32_t idx = r->aux;
                                if (tt < best || (tt == best && idx < bestIdx)) {
                                    best = tt; bestIdx = idx; bestK = q;
                                    tlim = (float)best * kInfl + 1e-30f;
                                }
                            }
                        }
                    }
                }
            };
            if (n.ch[0].n > 0) leaf(n.ch[0].n, n.ch[0].c, h0, t0);
            if (n.ch[1].n > 0) leaf(n.ch[1].n, n.ch[1].c, h1, t1);
            if (n.ch[2].n > 0) leaf(n.ch[2].n, n.ch[2].c, h2, t2);
            if (n.ch[3].n > 0) leaf(n.ch[3].n, n.ch[3].c, h3, t3);
            // ---- inner children: far to near; the nearest one some lane wants is entered, the others wait on the stack ----
            int32_t next = -1;
            if (n.ch[3].n == 0 && __ballot(h3 && t3 <= tlim) != 0ull) next = n.ch[3].c;
            if (n.ch[2].n == 0 && __ballot(h2 && t2 <= tlim) != 0ull) { if (next >= 0) wnode[sp++] = next; next = n.ch[2].c; }
            if (n.ch[1].n == 0 && __ballot(h1 && t1 <= tlim) != 0ull) { if (next >= 0) wnode[sp++] = next; next = n.ch[1].c; }
            if (n.ch[0].n == 0 && __ballot(h0 && t0 <= tlim) != 0ull) { if (next >= 0) wnode[sp++] = next; next = n.ch[0].c; }
            if (next >= 0) ni = next;
            else {
                if (sp == 0) break;
                ni = __builtin_amdgcn_readfirstlane(wnode[--sp]);
            }
        }
    }
    if (bestK < 0) return false;
    const Rec128* r = &sc.btris[bestK];
    out.t = best + offset;
    out.pos = s + d * best;          // the expression plane_hit evaluated for the winning triangle (pos = start + dir * rayFrac)
    out.nrm = mk(r->p[0], r->p[1], r->p[2]);
    out.color = r->color;
    out.tri = r->aux;
    return true;
}

// root of the chain for a wavefront of rays with a common origin: extra geometry per lane, then the packet walk
// (WIDE: on the four-wide tree's camera-ordered copy)
template <bool EXTRA, bool FILTER, int WIDE = 0>     // WIDE: 0 binary tree, 1 four-wide tree, 2 four-wide tree with (near, far) planes on all axes
__device__ bool root_intersect_pkt(const DevScene& sc, const Rec128* extra, int32_t* wnode, bool live, D3 s, D3 d, Hit& out, Ctr& c) {
    bool any = false;
    double best = DBL_MAX;
    if (EXTRA && live) {
        for (int i = 0; i < sc.nextra; ++i) {
            const Rec128* r = &extra[i];
            double t; D3 pos, nrm;
            uint32_t tests;
            const bool ok = extra_hit(r, s, d, t, pos, nrm, tests);
            c.geom += tests;
            if (ok && t < best) {
                best = t; any = true;
                out.t = t; out.pos = pos; out.nrm = nrm; out.color = r->color; out.tri = -1;
            }
        }
    }
    Hit mh;
    const bool model = WIDE == 2 ? bvh4_packet_nearest<FILTER, 7>(sc, wnode, live, s, d, mh, c)
                     : (WIDE == 1 ? bvh4_packet_nearest<FILTER, 0>(sc, wnode, live, s, d, mh, c) : bvh_packet_nearest<FILTER>(sc, wnode, live, s, d, mh, c));
    if (model && mh.t < best) { out = mh; any = true; }
    return any;
}

// --------------------------------------------------------------------------------------------------
// Blocker cache for any-hit (shadow) rays: "does THIS triangle record occlude the ray" with exactly the
// predicate of the full traversal (same clip, same rayFrac offset, same root-box containment).  Any
// triangle that satisfies it proves rayFrac <= 1.0 for the nearest hit, which is all ShadowMethod.cs:170
// looks at -- so testing a recently found occluder first never changes a result.
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool bvh_cached_blocks(const DevScene& sc, int32_t k, D3 s, D3 d) {
    D3 end = s + d * 10000.0;
    D3 original = s;
    if (!clip_segment<false>(sc.root, s, end)) return false;
    double offset = length(original - s) / length(d);
    double t; D3 pos;
    const Rec128* r = &sc.btris[k];
    return tri_hit(r->p, s, d, t, pos) && inside(sc.root.lo, sc.root.hi, pos) && (t + offset <= 1.0);
}
__device__ __forceinline__ bool brute_cached_blocks(const Rec128* tris, int32_t k, D3 s, D3 d) {
    double t; D3 pos;
    return tri_hit(tris[k].p, s, d, t, pos) && (t <= 1.0);
}

// --------------------------------------------------------------------------------------------------
// model + root geometry
// --------------------------------------------------------------------------------------------------
enum { MODE_REF = 0, MODE_BRUTE = 1, MODE_BVH = 2 };

template <int MODE, bool ANY>
__device__ __forceinline__ bool model_intersect(const DevScene& sc, const Rec128* tris, Stack st, D3 s, D3 d, Hit& out, Ctr& c) {
    if (MODE == MODE_REF) return ref_tree_intersect(sc, tris, st, s, d, out, c);
    if (MODE == MODE_BRUTE) return brute_intersect<ANY>(tris, sc.ntris, s, d, out, c);
    return bvh_intersect<ANY>(sc, st, s, d, out, c);
}

// Root of the decorator chain (Renderer.cs:1536-1549): the model, or GeometryCollection[extra..., model].
// With ANY the caller only needs "exists rayFrac <= 1.0" (<=> nearest.rayFrac <= 1.0).
template <int MODE, bool ANY, bool EXTRA>
__device__ bool root_intersect(const DevScene& sc, const Rec128* tris, const Rec128* extra, Stack st, D3 s, D3 d, Hit& out, Ctr& c) {
    if (!EXTRA) return model_intersect<MODE, ANY>(sc, tris, st, s, d, out, c);
    bool any = false;
    double best = DBL_MAX;
    for (int i = 0; i < sc.nextra; ++i) {
        const Rec128* r = &extra[i];
        double t; D3 pos, nrm;
        uint32_t tests;
        const bool ok = extra_hit(r, s, d, t, pos, nrm, tests);
        c.geom += tests;
        if (ok && t < best) {
            best = t; any = true;
            out.t = t; out.pos = pos; out.nrm = nrm; out.color = r->color; out.tri = -1;
            if (ANY && t <= 1.0) return true;
        }
    }
    Hit mh;
    if (model_intersect<MODE, ANY>(sc, tris, st, s, d, mh, c) && mh.t < best) { out = mh; any = true; }
    return any;
}

// --------------------------------------------------------------------------------------------------
// shading (ShadingMethod.IntersectRay / CalcLighting)
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ D3 mul3x4(const double* m, D3 v) {      // Matrix.Multiply3X4, Matrix.cs:49-56
    return mk(v.x * m[0] + v.y * m[1] + v.z * m[2] + m[3],
              v.x * m[4] + v.y * m[5] + v.z * m[6] + m[7],
              v.x * m[8] + v.y * m[9] + v.z * m[10] + m[11]);
}
__device__ __forceinline__ D3 mul3x3(const double* m, D3 v) {      // Instance.TransformDirection(Reverse)
    return mk(v.x * m[0] + v.y * m[1] + v.z * m[2],
              v.x * m[4] + v.y * m[5] + v.z * m[6],
              v.x * m[8] + v.y * m[9] + v.z * m[10]);
}
__device__ uint32_t shade(const FrameConst& fc, D3 pos, D3 nrm, uint32_t color) {
    D3 v = mul3x4(fc.t, pos);                                      // Instance.TransformPosToView, Instance.cs:168-184
    v.x = v.x / v.z * fc.fov_depth;
    v.y = v.y / v.z * fc.fov_depth;
    v.z = (v.z - fc.position_z + 1.0) * 0.5;
    D3 n = mul3x3(fc.t, nrm);
    D3 L;
    if (fc.flags & 8u) L = normalise(mk(fc.light_pos_view[0] - v.x, fc.light_pos_view[1] - v.y, fc.light_pos_view[2] - v.z));
    else L = mk(-fc.light_dir_view[0], -fc.light_dir_view[1], -fc.light_dir_view[2]);
    double diff = dot(L, n);
    diff = (0.0 > diff) ? 0.0 : diff;                              // Math.Max(0.0, x)
    double spec = 0.0;
    if (fc.flags & 16u) {
        D3 V = normalise(neg(v));
        D3 R = 2.0 * dot(L, n) * n - L;
        double ca = dot(R, V);
        spec = pow(ca, fc.shininess);
        spec = (0.0 > spec) ? 0.0 : spec;
    }
    double ch = 1.0 * fc.ambient + 1.0 * diff + 1.0 * spec;       // white materials, Color * double
    ch = (ch < 1.0) ? ch : ((ch != ch) ? ch : 1.0);                // Math.Min(c, 1.0)
    return modulate(color, to_byte(255 * ch));
}

// ShadowMethod.TraceRaysForSoftShadows: fraction of the area-light samples that reach the surface point
template <int MODE, bool EXTRA>
__device__ uint32_t soft_shadow(const DevScene& sc, const FrameConst& fc, const Rec128* tris, const Rec128* extra,
                                const double* offsets, Stack st, D3 pos, D3 nrm, Ctr& c) {
    int escapes = 0;
    D3 shadowEnd = pos + nrm * 0.001;                              // shadowProbeOffset
    for (int i = 0; i < fc.shadow_samples; ++i) {
        D3 off = mk(offsets[3 * i], offsets[3 * i + 1], offsets[3 * i + 2]);
        D3 rs, rd;
        if (fc.flags & 8u) {
            D3 src = mk(fc.light_pos_model[0] + off.x, fc.light_pos_model[1] + off.y, fc.light_pos_model[2] + off.z);
            rd = shadowEnd - src;
            rs = src;
        } else {
            rd = mk(fc.light_dir_model[0], fc.light_dir_model[1], fc.light_dir_model[2]);
            rs = shadowEnd + rd * 1000.0 + off;
        }
        Hit h;
        bool blocked;
        c.rays++;
        if (MODE == MODE_REF) {
            // the reference tree returns ITS nearest hit; only that hit's rayFrac is compared with 1.0
            blocked = root_intersect<MODE, false, EXTRA>(sc, tris, extra, st, rs, rd, h, c) && !(h.t > 1.0);
        } else {
            blocked = root_intersect<MODE, true, EXTRA>(sc, tris, extra, st, rs, rd, h, c) && !(h.t > 1.0);
        }
        if (!blocked) escapes++;
    }
    double frac = (double)escapes / (double)fc.shadow_samples;
    return to_byte(frac * 255);
}

// per-channel blend of two packed colours: ((s * (255 - k)) >> 8) + ((r * k) >> 8), alpha 0xFF.
// Deliberately __noinline__ and composed with '+': inlined into the fold loop (ROCm 7.2 hipcc -O3, gfx950) the
// middle (green) byte came out one too small -- a byte-select peephole miscompile that the GPU-vs-CPU-checker test caught.
__device__ __noinline__ uint32_t blend_packed(uint32_t sfc, uint32_t refl, uint32_t k) {
    uint32_t out = 255u << 24;
    const uint32_t ik = 255u - k;
    uint32_t sr = (sfc >> 16) & 0xffu, sg = (sfc >> 8) & 0xffu, sb = sfc & 0xffu;
    uint32_t rr = (refl >> 16) & 0xffu, rg = (refl >> 8) & 0xffu, rb = refl & 0xffu;
    uint32_t cr = ((sr * ik) >> 8) + ((rr * k) >> 8);
    uint32_t cg = ((sg * ik) >> 8) + ((rg * k) >> 8);
    uint32_t cb = ((sb * ik) >> 8) + ((rb * k) >> 8);
    out += (cr & 0xffu) << 16;
    out += (cg & 0xffu) << 8;
    out += (cb & 0xffu);
    return out;
}

// TraceRayComplex with the decorator chain of one frame.
// fc.max_bounces > 0: the config-5 extension (no counterpart in the reference, definition shared with the CPU checker):
// mirror bounce r = dir - n * (2.0 * dir.n) from pos + n * 0.001, each level coloured by the same chain, packed colours
// blended per channel c = ((surface * (255 - k)) >> 8) + ((reflected * k) >> 8), k = (byte)(reflectivity * 255).
template <int MODE, bool EXTRA>
__device__ uint32_t trace_camera_ray(const DevScene& sc, const FrameConst& fc, const Rec128* tris, const Rec128* extra,
                                     const double* offsets, Stack st, D3 s, D3 d, Ctr& prim, Ctr& sec) {
    Hit h;
    if (!root_intersect<MODE, false, EXTRA>(sc, tris, extra, st, s, d, h, prim)) return fc.background;
    const int maxb = fc.max_bounces;
    constexpr int kMaxLevels = 17;                                 // max_bounces <= 16
    uint32_t surface[kMaxLevels];                                  // only ever indexed with unrolled constants: stays in registers
#pragma unroll
    for (int i = 0; i < kMaxLevels; ++i) surface[i] = 0u;
    int levels = 0;
    bool tail_is_surface = false;
    for (;;) {
        uint32_t color = h.color;
        if (fc.flags & 1u) color = shade(fc, h.pos, h.nrm, color);
        if (fc.flags & 2u) color = modulate(color, soft_shadow<MODE, EXTRA>(sc, fc, tris, extra, offsets, st, h.pos, h.nrm, sec));
        if (maxb <= 0) return color;                               // the reference's behaviour
#pragma unroll
        for (int i = 0; i < kMaxLevels; ++i) if (i == levels) surface[i] = color;
        ++levels;
        if (levels > maxb) { tail_is_surface = true; break; }
        const D3 n = h.nrm;
        const D3 r = d - n * (2.0 * dot(d, n));
        const D3 rs = h.pos + n * 0.001;
        Hit nx;
        if (!root_intersect<MODE, false, EXTRA>(sc, tris, extra, st, rs, r, nx, sec)) break;
        d = r;
        h = nx;
    }
    const uint32_t k = to_byte(fc.reflectivity * 255.0);
    // fold from the deepest level back to the camera ray
    uint32_t color = fc.background;
    const int last = tail_is_surface ? levels - 2 : levels - 1;    // deepest level that gets blended with what lies beyond it
#pragma unroll
    for (int i = kMaxLevels - 1; i >= 0; --i) {
        if (tail_is_surface && i == levels - 1) color = surface[i];
        if (i <= last) {
            color = blend_packed(surface[i], color, k);
        }
    }
    return color;
}

__device__ __forceinline__ uint32_t wave_sum(uint32_t v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// statistics are accumulated with one 64-bit atomic per wave and counter; zero contributions (idle waves of the persistent
// kernels, background tiles) are skipped: thousands of them on the same few addresses cost more than the kernel itself
__device__ __forceinline__ void stat_add(unsigned long long* p, uint32_t v) { if (v) atomicAdd(p, (unsigned long long)v); }
// the same for the kernels with one workgroup per tile (tens of thousands of workgroups): the four waves first add up in
// LDS, then one thread issues the global atomics.  Every thread of the workgroup must call it.
__device__ __forceinline__ void block_stat_add(unsigned long long* s0, unsigned long long* s1, unsigned long long* s2, unsigned long long* s3,
                                               uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    __shared__ unsigned int acc[4];
    if (threadIdx.x < 4) acc[threadIdx.x] = 0u;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        if (a) atomicAdd(&acc[0], a);
        if (b) atomicAdd(&acc[1], b);
        if (c) atomicAdd(&acc[2], c);
        if (d) atomicAdd(&acc[3], d);
    }
    __syncthreads();
    if (threadIdx.x == 0) { stat_add(s0, acc[0]); stat_add(s1, acc[1]); stat_add(s2, acc[2]); stat_add(s3, acc[3]); }
    __syncthreads();                       // a second call re-initialises acc
}

}  // namespace sr
