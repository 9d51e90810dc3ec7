// sr_types.h -- records shared by the host scene builder and the gfx950 kernels.
// Layouts are the HBM layouts described in DESIGN.md ("Data layout in HBM").
#pragma once
#include <stdint.h>

namespace sr {

// 128-byte primitive record, 16-byte aligned so a lane reads it as 8 x global_load_dwordx4 and a wave
// broadcast-reads it from LDS with ds_read_b128.
//
// Triangle (Raytrace/Triangle.cs:29-57 precomputed):
//   p[0..2] plane unit normal   p[3] plane originDist     p[4..6] vertex1
//   p[7..9] edge2Perp           p[10] edge1 . edge2Perp   p[11..13] edge1Perp   p[14] edge2 . edge1Perp
//   aux = TriangleIndex
// Sphere (Raytrace/Sphere.cs:26-33):   p[0..2] centre  p[3] radius  p[4] radiusSqr          aux = kind 0
// Plane  (Raytrace/Plane.cs:22-29):    p[0..2] unit normal  p[3] originDist                 aux = kind 1
// Extra triangles use the triangle layout with aux = kind 2.
struct alignas(16) Rec128 {
    double   p[15];
    uint32_t color;
    int32_t  aux;
};
static_assert(sizeof(Rec128) == 128, "Rec128 must be 128 bytes");

// Reference tree node (SpatialSubdivision.Node flattened, SpatialSubdivision.cs:22-46), 32 B.
//   internal: axis 0..2, split = splittingPlane.DistanceToOrigin, a = normalSide index, b = backSide index
//   leaf:     axis = -1, a = first entry in the leaf triangle-index list, b = count, box = leaf box index
struct alignas(16) RefNode {
    double  split;
    int32_t axis;
    int32_t a;
    int32_t b;
    int32_t box;
    int32_t pad[2];
};
static_assert(sizeof(RefNode) == 32, "RefNode must be 32 bytes");

// Leaf bounding box with the reference's +-1e-10 slack already applied on the host
// (AxisAlignedBox.ContainsPoint, AxisAlignedBox.cs:143-149): lo = min - eps, hi = max + eps. 48 B.
struct LeafBox {
    double lo[3];
    double hi[3];
};

// Own BVH node, 64 B = one 64-byte line: both children's boxes in fp32 (conservatively padded, in
// coordinates relative to the root-box centre) + child links.  n? > 0: leaf, c? = first record in the
// leaf-ordered triangle array; n? == 0: inner node index c?; n? < 0: empty child.
struct alignas(16) BvhNode {
    float   lo0[3], hi0[3];
    float   lo1[3], hi1[3];
    int32_t c0, c1;
    int32_t n0, n1;
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be 64 bytes");

// The same tree collapsed to four children per node for the wave-cooperative PACKET walks (k_primary, k_shaft_pkt): 128 B = two
// lines, fetched by ONE pair of scalar loads per step.  A child record is a BvhNode half: fp32 box (same frame, same padding:
// the collapse copies the BVH2 boxes bit for bit) + link + count.  n > 0: leaf of n triangles, c = first record in leaf order;
// n == 0: inner node index c (into the Bvh4Node array); n < 0: empty slot.  The children of the stored tree are in build order;
// the walks read per-frame COPIES whose children are sorted front to back for the frame's ray origin / back to front for its
// light (k_order_nodes): rays with a common origin (or a common end) meet the subtrees of a node in the same order whatever
// their direction, so a packet walk needs no per-step vote -- slot 0 first, the others pushed far to near.  The LIGHT-ordered copy also
// re-arranges a child's six planes (slots lo[0..2], hi[0..2] hold lo.x, lo.y, hi.x, hi.y, lo.z, hi.z) and, on the axes where the light lies outside
// the root box, stores (near, far) instead of (lo, hi): only k_shaft_pkt4 reads it (shaft_slabs).
struct alignas(16) Bvh4Child {
    float   lo[3], hi[3];
    int32_t c, n;
};
struct alignas(128) Bvh4Node {
    Bvh4Child ch[4];
};
static_assert(sizeof(Bvh4Node) == 128, "Bvh4Node must be 128 bytes");

// fp32 conservative description of one triangle for the shadow-shaft walk (64 B = one line), in coordinates
// relative to the root-box centre: the plane (unit normal n, offset d) and the three edge planes (unit in-plane
// normals m_k pointing inward, offsets c_k).  A point within distance rho of the triangle satisfies
// |n.x - d| <= rho and m_k.x - c_k >= -rho for k = 1..3 (the mitred offset polygon contains the rounded one).
// Degenerate triangles store all zeros: every test passes.
struct alignas(16) TriSlab {
    float n[3], d;
    float m1[3], c1;
    float m2[3], c2;
    float m3[3], c3;
};
static_assert(sizeof(TriSlab) == 64, "TriSlab must be 64 bytes");

// fp32 description of one triangle as seen from ONE ray origin O (the camera of a frame), 64 B = one line, BVH-leaf order,
// laid out in pairs for packed FMAs: with a_i = v_i - O, the cone planes w1 = -(a1 x a2), w2 = -(a2 x a3), w3 = -(a3 x a1)
// (computed in FP64, rounded once) and the reference's unit plane normal n.  A ray O + t d can only hit the triangle
// (Triangle.IntersectRay) if w_k . d >= 0 for k = 1..3 (front side) and n . d < 0; m_k = 8 * 2^-24 * |w_k| bounds the fp32
// evaluation error of w_k . d per unit |d|.  Degenerate triangles: all zeros with huge margins (never filtered).
struct alignas(16) CamCone {
    float w12x[2], w12y[2], w12z[2];   // (w1.x, w2.x) (w1.y, w2.y) (w1.z, w2.z)
    float w3nx[2], w3ny[2], w3nz[2];   // (w3.x, n.x) (w3.y, n.y) (w3.z, n.z)
    float m12[2];                      // (m1, m2)
    float m3n[2];                      // (m3, mn)
};
static_assert(sizeof(CamCone) == 64, "CamCone must be 64 bytes");

// Root box as the clip needs it (AxisAlignedBox.cs:16-28,143-149)
struct RootBox {
    double min[3], max[3];       // model.Min / model.Max
    double lo[3], hi[3];         // min - 1e-10, max + 1e-10
    double pd[6];                // originDist of the six planes (-x,-y,-z at min; +x,+y,+z at max)
    double centre[3];            // (min + max) * 0.5, for the BVH's fp32 frame
};

// library-internal FrameConst.flags bit (the caller's SR_F_* bits are below 1 << 16): every area-light sample of every hit point
// provably escapes (sr_api.cpp render_common), so ShadowMethod's factor is the constant (byte)(1.0 * 255) and no shadow ray is traced
constexpr uint32_t kFlagAllSamplesEscape = 1u << 30;

// Per-frame constants, passed by value as a kernel argument (lives in SGPRs / the kernarg segment).
struct FrameConst {
    int32_t  width, height;
    int32_t  first_row;          // first image row of this launch's row table entry 0 (see row_map)
    int32_t  num_rows;           // rows rendered by this launch (compact count when strips are on)
    int32_t  sub_pixel_res;
    uint32_t background;         // already OR-ed with 0xFF000000
    uint32_t flags;
    int32_t  shadow_samples;
    int32_t  strip_rows, strip_count, strip_index;
    int32_t  start_row;
    double   t[12];              // Instance._transform rows 0..2
    double   it[12];             // Instance._inverseTransform rows 0..2
    double   position_z, fov_depth, focal_depth, focal_blur_strength;
    double   ambient, shininess;
    double   light_dir_view[3], light_pos_view[3];
    double   light_dir_model[3], light_pos_model[3];
    double   start_world[3];     // R^-1 * (0,0,-Position.z), Renderer.cs:1717
    double   aspect;             // (double)height / (double)width, Renderer.cs:621
    int32_t  debug;              // SR_DEBUG experiment switch (0 in production)
    int32_t  max_bounces;        // config-5 extension: mirror bounces (0 = the reference's behaviour)
    double   reflectivity;
    double   light_radius;       // max |area-light offset| (0.2 for the reference table): bounds the shadow shaft
    // More than 128 samples on the shaft path: the shadow stage runs once per chunk of <= 128 samples (shadow_samples = the chunk's count)
    // and ADDS every hit point's escape count to accum[sample index] instead of finishing the pixel; k_accum_finish turns the sums
    // into ShadowMethod's byte with the total count.  nullptr: the ordinary one-pass frame.
    uint32_t* accum;
};

}  // namespace sr
