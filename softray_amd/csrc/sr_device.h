// sr_device.h -- interface between the C-ABI layer (sr_api.cpp) and the gfx950 kernels (sr_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sr_types.h"

namespace sr {

// Device-resident scene (all pointers are HBM addresses on the scene's device).
struct DevScene {
    const Rec128*  tris;        // model triangles in TriangleIndex order
    int32_t        ntris;
    const Rec128*  extra;       // ExtraGeometryToRaytrace, insertion order
    int32_t        nextra;
    // reference tree
    const RefNode* rnodes;
    const LeafBox* rboxes;
    const int32_t* rleaf;
    int32_t        rdepth;      // TreeDepth (stack bound)
    // own BVH
    const BvhNode* bnodes;
    const Rec128*  btris;       // triangle records gathered in leaf order (aux = TriangleIndex)
    const TriSlab* bslab;       // fp32 shaft-prefilter records, same order as btris
    const CamCone* bcam;        // fp32 camera-cone records of the current frame's ray origin, same order (nullptr: none)
    const double*  v9;          // model vertices [ntris][3][3], TriangleIndex order
    int32_t        bdepth;
    int32_t        bnode_bits;  // bits needed for a BVH node index (stack words pack node | bound)
    // the same tree with four children per node (packet walks), children sorted per frame: front to back for the camera rays'
    // origin / back to front for the point light (nullptr: not made -> the BVH2 packet walks run)
    const Bvh4Node* b4;         // build order (private walks: every lane orders the children for its own ray)
    const Bvh4Node* b4cam;
    const Bvh4Node* b4light;
    int32_t        b4depth;
    // axes (bit a) on which the ordered copy stores (near, far) instead of (lo, hi): the camera origin / the light lies outside the
    // root box's slab on that axis, so every ray of the frame has the same sign there (child_slabs<KNOWN>, sr_trace.h)
    int32_t        b4cam_known, b4light_known;
    RootBox        root;
    uint8_t*       shadow_cache; // static soft-shadow cache, 128^3 bytes, 0 = empty cell (SR_F_STATIC_SHADOWS frames)
};

constexpr int kShaftRounds = 2;
enum KernelId { K_RENDER = 0, K_TRACE = 1, K_PRIMARY = 2, K_SHADOW = 3, K_RESOLVE = 4, K_SHAFT = 5, K_FALLBACK = 6, K_SHAFT2 = 7, K_SHADOW2 = 8, K_POST = 9, K_ANTI_ALIAS = 10, K_BOUNCE = 11, K_COUNT = 12 };
const char* kernel_name(int id);

struct RenderLaunch {
    DevScene    sc;
    FrameConst  fc;
    int32_t     mode;           // SR_MODE_*
    const double* offsets;      // device [shadow_samples][3]
    const int32_t* row_map;     // device [num_rows]: image row of each compact row
    uint32_t*   pixels;         // device output
    unsigned long long* stats;  // device [8] or nullptr: primary {rays, tri tests, nodes, leaves}, secondary {same}
    hipStream_t stream;
};

// The one-kernel renderer: ray generation, traversal, shading, inline shadow rays, sub-pixel resolve.
hipError_t launch_render(const RenderLaunch& L);

// The default path: k_primary -> k_shadow -> k_resolve (sr_pipeline.hip).
struct PipelineLaunch {
    DevScene    sc;
    FrameConst  fc;
    int32_t     mode;
    const double*  offsets;     // device [shadow_samples][3]
    const int32_t* row_map;     // device [num_rows]
    uint32_t*   pixels;         // device output frame (full surface or compact strips)
    uint32_t*   samples;        // device [band_rows * width * n^2] sample colours (sub_pixel_res > 1 only)
    void*       hits;           // device hit queue, band_rows * width * n^2 records of pipeline_hit_record_bytes()
    unsigned int* counters;     // device uint[8]: hits, k_shadow head, items entering round 1.., fallback count, fallback head
    // shaft path (own BVH + point light), kShaftRounds rounds of (k_shaft, k_shadow_test); round 0 covers every hit
    unsigned int  round_items[kShaftRounds];      // capacity (hits) of the round's buffers (round 0: band samples)
    int           round_cap[kShaftRounds];        // candidate-list length (stride) of the round
    unsigned int* round_list[kShaftRounds];       // device hit indices entering the round (round 0: hits k_shaft left undecided)
    void*         round_state[kShaftRounds];      // device RoundState per item (round 0: nullptr)
    unsigned int* round_cand_count[kShaftRounds]; // device per-item candidate count | truncated flag
    int32_t*      round_cand[kShaftRounds];       // device [items][pipeline_round_cap(round)] (round_cand[0] == nullptr: no shaft path)
    void*         hits2;        // second ray queue of the mirror-bounce pipeline (same size as hits)
    unsigned int* ray_sort_buf; // mirror-bounce pipeline: 4 x band samples of scratch for the per-level ray order (nullptr: rays walk in queue order)
    void*         ray_sort_temp;
    size_t        ray_sort_temp_bytes;
    uint32_t*     bounce_levels; // device [band samples][max_bounces + 1]: colour of every level of a sample's mirror chain
    uint8_t*      bounce_nlev;  // device [band samples]: levels stored | 0x80 when the deepest level is a surface
    void*         bounce_prep;  // device [band samples] x 64 B: a level's rays after the FP64 clip, in walk order (k_bounce_prep)
    void*         bounce_res;   // device [band samples] x 16 B: nearest hit of every such ray (k_bounce_walk)
    int32_t*      bounce_stack; // device: the part of k_bounce_walk's per-lane stacks that does not live in LDS ([level][lane])
    size_t        bounce_stack_bytes;
    void*         static_hits;  // device HitRec[min(band samples, 128^3)]: generators of a static-shadow frame
    unsigned long long* static_claim; // device [128^3]: smallest order key that asked for an empty cell
    int32_t       static_concurrency; // rayTraceConcurrency of the frame
    unsigned int* fallback;     // device [band samples]: hits that need the exact per-lane fallback
    void*         fallback_state; // device RoundState per fallback entry: which samples are still undecided
    unsigned int* fallback_rays;  // device [fallback_ray_cap]: (entry << 7 | sample) of every undecided sample
    unsigned int  fallback_ray_cap;
    unsigned int* fallback_overflow; // device [band samples]: entries the ray list had no room for
    int32_t     band_rows;      // rows per band (multiple of 16)
    int32_t     row_first, row_limit; // compact rows [row_first, row_limit) of the frame are this launch's share
    int32_t     persistent_blocks;
    bool        per_lane_shadows; // force k_shadow (one lane per hit) instead of k_shadow_packet (cross-check)
    int32_t     tile_queue_n2, tile_queue_rows; // (set by launch_pipeline) > 0: the hit queue of this band is tile-indexed
    int32_t     round2_node_budget; // later shaft rounds: a private walk gives up after this many nodes (0 = never)
    bool        per_lane_primary; // k_primary with private walks instead of the packet walk + camera-cone filter (cross-check)
    bool        bvh2_packets;     // the packet walks on the two-wide tree with a per-step vote (round 2's kernels; cross-check)
    bool        primary_stats_only;   // `stats` counts the primary rays only (SR_F_PRIMARY_STATS_ONLY): the shadow / bounce stages run their uncounted instantiations
    int32_t     shaft_wgs_per_cu; // resident workgroups per CU of the persistent shaft walk (0 = 6)
    bool        shadows_on_bvh;   // mode != BVH: the shadow rays of a dynamic frame are traced on the own BVH (shaft path) all the same
    int32_t     per_lane_shaft;   // bit 0: k_shaft (private walks) for the first round instead of k_shaft_pkt, bit 1: for the later rounds instead of k_shaft_coop (cross-checks)
    bool        exact_shadow_tests; // k_shadow_test (every pair in FP64) instead of k_shadow_cls (fp32 classification first)
    unsigned long long* stats;  // device [8] or nullptr
    // longest-first order of the persistent shaft kernel's tiles: k_shaft_pkt4 leaves every 8x8 tile's walk length in tile_cost, k_tile_order
    // turns them into per-XCD lists (longest walks first) for the NEXT frame with the same tile grid; `tile_order_tag` is host state of the
    // scratch set: the grid the lists in tile_order were made for (0: none).  All three nullptr: natural order
    unsigned int* tile_cost;
    unsigned int* tile_order;
    unsigned long long* tile_order_tag;
    hipStream_t stream;
    void (*get_events)(void* user, int kernel_id, hipEvent_t* start, hipEvent_t* stop);   // optional per-launch timing
    // optional: called when every kernel of a row band has been enqueued on `stream` -- compact rows [row_begin, row_begin + row_count)
    // of the frame are final once what is on the stream now has run (sr_render copies a band to the host while the next ones render)
    void (*band_done)(void* user, int band_index, int row_begin, int row_count, hipStream_t stream);
    void*       user;
};
hipError_t launch_pipeline(const PipelineLaunch& L);
// per-frame pre-pass: a copy of the four-wide nodes with every node's children sorted by the distance of their box centres from
// `point` (model space), nearest first (camera origin) or farthest first (light: nearest to the surface points first)
// swap_mask (bit a): exchange lo and hi on axis a in the copy (the rays of the frame travel towards smaller coordinates there)
// live_runs (nullable): int2 per (node, slot) from launch_facing_partition -- the (offset, count) of a leaf's records that rays of this
// copy can hit
hipError_t launch_order_nodes(const Bvh4Node* in, Bvh4Node* out, int num_nodes, const RootBox& root, const double point[3], bool far_first, int swap_mask,
                              const void* live_runs, hipStream_t stream);
// per (camera origin, light) pre-pass: re-orders the records of every leaf in place so that the records a camera ray / a shadow sample ray
// can hit (the triangle faces the origin / the light) are contiguous, and writes their (offset, count) per (node, slot) (int2 each)
hipError_t launch_facing_partition(const Bvh4Node* base, int num_nodes, Rec128* btris, TriSlab* bslab, const double origin[3], bool use_cam,
                                   const double light[3], double light_radius, bool use_light, void* cam_rng, void* light_rng, hipStream_t stream);
// per-frame pre-pass: camera-cone records of every BVH triangle for the ray origin `origin` (model space)
hipError_t launch_cam_cones(const DevScene& sc, int ntris, const double origin[3], CamCone* out, hipStream_t stream);
size_t pipeline_hit_record_bytes();
size_t pipeline_static_cells();
int pipeline_round_cap(int round);
int pipeline_round_cap_max(int round);
int pipeline_bounce_lds_levels();      // stack levels per lane k_bounce_walk keeps in LDS (deeper ones live in PipelineLaunch::bounce_stack)
size_t pipeline_round_state_bytes();
size_t pipeline_counter_bytes();
size_t pipeline_tile_items(int width, int rows, int n2);

// Own BVH built on the device (sr_lbvh.hip).  Inputs in TriangleIndex order, outputs caller-allocated (n entries each).
hipError_t gather_records_device(int n, const unsigned int* d_order, const Rec128* d_tris, Rec128* d_btris, const TriSlab* d_slab_in,
                                 TriSlab* d_bslab, hipStream_t stream);
hipError_t build_bvh_device(const double* d_v9, int n, const RootBox& root, const Rec128* d_tris, const TriSlab* d_slab_in,
                            BvhNode* d_nodes, Rec128* d_btris, TriSlab* d_bslab, int* num_nodes, int* depth, hipStream_t stream, int leaf_max = 0);

// order of a bounce level's ray queue by (origin cell, direction octant) (sr_raysort.hip): order_out = permutation of [0, cap)
size_t ray_sort_temp_bytes(unsigned int cap);
hipError_t ray_sort(const void* queue, const unsigned int* d_count, unsigned int cap, const RootBox& root, unsigned int* keys, unsigned int* keys2,
                    unsigned int* idx, unsigned int* order_out, void* temp, size_t temp_bytes, hipStream_t stream);

// the four-wide tree collapsed from a device-resident binary tree (same rule as the host's collapse_bvh4); d_wide: >= num_nodes entries
hipError_t collapse_bvh4_device(const BvhNode* d_nodes, int num_nodes, Bvh4Node* d_wide, int* num_wide, int* depth, hipStream_t stream);

// fp32 TriSlab records of n triangles (TriangleIndex order) computed on the device from the FP64 vertices (sr_lbvh.hip)
hipError_t make_slabs_device(const double* d_v9, int n, const RootBox& root, TriSlab* d_out, hipStream_t stream);

// Surface passes (sr_post.hip): PostProcessImage colour functions and AntiAliasImage, Renderer.cs:819-978.
hipError_t launch_post_process(uint32_t* d_pixels, long long count, int style, uint32_t background, int num_cus, hipStream_t stream);
hipError_t launch_anti_alias(const uint32_t* d_src, uint32_t* d_dst, int dst_w, int dst_h, int res, hipStream_t stream);

struct TraceLaunch {
    DevScene sc;
    int32_t  mode;              // SR_MODE_*
    bool     with_extra;        // root geometry of the chain (extra + model)
    int64_t  n;
    const double* starts; const double* dirs;      // device [n][3]
    uint8_t* hit; double* ray_frac; double* pos; double* normal; uint32_t* color; int32_t* tri; int32_t* counters;
    hipStream_t stream;
};
hipError_t launch_trace(const TraceLaunch& L);
hipError_t launch_shade_points(const FrameConst& fc, long long n, const double* pos, const double* nrm, const uint32_t* color, uint32_t* out, hipStream_t stream);

}  // namespace sr
