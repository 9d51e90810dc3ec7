// sr_host.h -- host-side scene preparation for libsoftray_hip: everything the reference does once
// per model in PreCalculate() (Engine3D/Renderer.cs:673-699), restated to produce flat, GPU-ready
// arrays.  Must be compiled with -ffp-contract=off: the triangle/plane precomputation has to round
// exactly like the reference's C# expressions.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>

#include "sr_types.h"

namespace sr {

struct Vec3 {
    double x, y, z;
};

// ---- System.Random (.NET Framework 4.0 BCL subtractive generator) ----
class NetRandom {
public:
    explicit NetRandom(int32_t seed);
    int32_t next();                 // Random.Next()
    double  next_double();          // Random.NextDouble()
private:
    int32_t sample();
    int32_t table_[56];
    int     inext_, inextp_;
};

// ---- primitive records ----
Rec128 make_triangle_record(Vec3 v1, Vec3 v2, Vec3 v3, uint32_t color, int32_t aux);  // Triangle.cs:29-57
Rec128 make_sphere_record(Vec3 centre, double radius, uint32_t color);                // Sphere.cs:26-33
Rec128 make_plane_record(Vec3 point, Vec3 normal, uint32_t color);                    // Plane.cs:22-29
Rec128 make_box_record(Vec3 mn, Vec3 mx);                                             // AxisAlignedBox.cs:15-28 (six Plane objects, Color.White)
RootBox make_root_box(const double bmin[3], const double bmax[3]);                    // AxisAlignedBox.cs:16-28

// ---- reference tree (SpatialSubdivision.cs:49-315) flattened ----
struct RefTree {
    std::vector<RefNode> nodes;        // nodes[0] = root
    std::vector<LeafBox> leaf_boxes;
    std::vector<int32_t> leaf_tris;    // triangle indices, list order of the reference preserved
    int32_t tree_depth = 0, num_nodes = 0, num_leaf_nodes = 0;
    int32_t max_stack = 1;             // deepest leaf level = traversal stack bound
    bool built = false;
};
// returns false when a vertex lies outside the box (ArgumentOutOfRangeException, :287-295)
bool build_ref_tree(const std::vector<double>& v9, const double bmin[3], const double bmax[3],
                    int max_depth, int max_per_leaf, RefTree& out);

// ---- own BVH ----
struct Bvh {
    std::vector<BvhNode> nodes;        // nodes[0] = root (always an inner record)
    std::vector<int32_t> order;        // leaf-ordered triangle indices: records are gathered in this order
    int32_t depth = 0;
    bool built = false;
};
// threads: 0 = the host's cores (at most 16); the tree does not depend on it
void build_bvh(const std::vector<double>& v9, const RootBox& root, Bvh& out, int leaf_max = 4, int threads = 0);
// The BVH2 collapsed to <= 4 children per node (sr_types.h Bvh4Node): a node takes its two children and, while it has fewer
// than four, replaces the inner child with the largest box by that child's two children.  Boxes, leaves and the leaf order are
// the BVH2's, so both trees describe the same hierarchy of the same records.  Returns the depth (root = 1).
int collapse_bvh4(const BvhNode* nodes, size_t num_nodes, std::vector<Bvh4Node>& out);

// ---- Instance / Renderer helpers ----
void instance_matrices(const double position[3], double yaw, double pitch, double roll,
                       double transform[12], double inv_transform[12]);                // Instance.cs:134-135
double default_fov_depth();                                                            // Renderer.cs:97-101
void area_light_offsets(int32_t seed, int32_t count, double* out3);                    // ShadowMethod.cs:63-73

// ---- Model.Load3ds + PostProcessGeometry ----
struct LoadedModel {
    std::vector<double>   v9;          // [n][3][3]
    std::vector<uint32_t> argb;        // [n]
    double bmin[3], bmax[3];
};
// returns empty string on success, else the FormatException-style message
std::string load_3ds(const uint8_t* data, size_t len, LoadedModel& out);

}  // namespace sr
