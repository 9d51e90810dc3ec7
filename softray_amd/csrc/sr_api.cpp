// sr_api.cpp -- the C ABI of libsoftray_hip.so (include/softray.h): scene ownership, H2D staging,
// per-frame constant preparation and kernel dispatch.  There is no CPU compute path in here: a scene
// without a HIP device refuses every compute call with SR_ERR_NO_DEVICE.
// Compile with -ffp-contract=off (the per-frame constants must round like the reference's C#).
#include "../../include/softray.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "sr_device.h"
#include "sr_host.h"
#include "sr_rccl.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string& msg) {
    g_err = msg;
    return code;
}
int hip_fail(hipError_t e, const char* what) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return SR_ERR_HIP;
}
#define SR_HIP(call)                                         \
    do {                                                     \
        hipError_t e__ = (call);                             \
        if (e__ != hipSuccess) return hip_fail(e__, #call);  \
    } while (0)

int rccl_fail(const sr::RcclApi* api, int rc, const char* what) {
    g_err = std::string(what) + ": " + (api && api->GetErrorString ? api->GetErrorString(rc) : "RCCL error") + " (" + std::to_string(rc) + ")";
    return SR_ERR_HIP;
}
#define SR_RCCL(api, call)                                        \
    do {                                                          \
        int r__ = (call);                                         \
        if (r__ != 0) return rccl_fail(api, r__, #call);          \
    } while (0)

// growable device buffer
struct DBuf {
    void*  p = nullptr;
    size_t cap = 0;
    hipError_t reserve(size_t bytes) {
        if (bytes <= cap && p) return hipSuccess;
        if (p) { hipError_t e = hipFree(p); p = nullptr; cap = 0; if (e != hipSuccess) return e; }
        size_t want = std::max<size_t>(bytes, 256);
        hipError_t e = hipMalloc(&p, want);
        if (e == hipSuccess) cap = want;
        return e;
    }
    template <class T> hipError_t upload(const std::vector<T>& v) {
        hipError_t e = reserve(v.size() * sizeof(T));
        if (e != hipSuccess || v.empty()) return e;
        return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

}  // namespace

struct sr_scene {
    int device = -1;
    // sr_create_multi: this scene only dispatches to one complete scene per device (the model is replicated, a frame is split
    // into interleaved 16-row strips); empty for an ordinary scene
    std::vector<sr_scene*> parts;
    // host copies (what Renderer keeps between frames)
    std::vector<double>   v9;
    std::vector<uint32_t> argb;
    double bmin[3] = {0, 0, 0}, bmax[3] = {0, 0, 0};
    double vmin[3] = {0, 0, 0}, vmax[3] = {0, 0, 0};   // bounds of the vertices themselves (the caller's box need not be tight, nor, in brute-force mode, contain them)
    bool have_model = false;
    size_t ntris = 0;                      // triangles of the model (a part of a multi-device scene keeps the count, not the arrays)
    // (part of a multi-device scene) the scene whose HOST arrays -- vertices, records, reference tree, SAH nodes and order -- this
    // part uploads from: the model lives once on the host however many devices render it
    const sr_scene* host_src = nullptr;
    std::vector<sr::Rec128> tri_recs;      // geometry_simple, Renderer.cs:1452-1469
    std::vector<sr::Rec128> extra_recs;    // ExtraGeometryToRaytrace
    sr::RootBox root{};
    sr::RefTree ref;
    sr::Bvh     bvh;
    bool        bvh_on_device = false;   // built by sr_lbvh.hip: no host copy of the nodes
    size_t      bvh_num_nodes = 0;
    // device state
    DBuf d_tris, d_extra, d_rnodes, d_rboxes, d_rleaf, d_bnodes, d_btris, d_bslab, d_v9, d_bcam;
    double cam_origin[3] = {0, 0, 0};    // ray origin the camera-cone records in d_bcam (and the node order of d_b4cam) were made for
    bool   cam_valid = false;
    // four-wide tree of the packet walks: build-order nodes + the two per-frame ordered copies (camera origin / point light)
    DBuf d_b4, d_b4cam, d_b4light;
    size_t b4_num = 0;
    int    b4_depth = 0;
    bool   b4cam_valid = false, b4light_valid = false;
    int    b4cam_known = 0, b4light_known = 0;   // axes on which the ordered copy holds (near, far) planes (sr_device.h)
    // k_facing_partition: the leaves' records grouped by which of {camera rays, shadow sample rays} can hit them, for one (origin, light)
    DBuf   d_rng_cam, d_rng_light;
    bool   part_valid = false, part_cam = false, part_light = false;
    double part_origin[3] = {0, 0, 0}, part_lightpos[3] = {0, 0, 0}, part_radius = 0;
    double b4_light[3] = {0, 0, 0};      // light position the order of d_b4light was made for
    // the per-origin / per-light records above are written on whatever stream the frame that needs them runs on: `pre_ready` is
    // recorded after every rewrite and waited for by every frame (another stream may use them next), `pre_used` is recorded at the
    // end of every frame that read them and waited for before the next rewrite
    hipEvent_t pre_ready = nullptr, pre_used = nullptr;
    bool pre_ready_set = false, pre_used_set = false;
    DBuf d_shadow_cache, d_static_claim, d_static_hits;
    bool shadow_cache_empty = true;      // the device cache must be zeroed before its next use
    DBuf d_pixels, d_aa, d_stats, d_io[9];
    // per-frame tables (area-light offsets + row map): pinned host staging and device copies, double-buffered; a slot is
    // rewritten only after the frame that last used it has finished (event), and not at all when the tables are unchanged
    struct FrameTables {
        void* host = nullptr; size_t host_cap = 0;
        DBuf  dev;
        hipEvent_t used = nullptr; bool in_flight = false;
        hipEvent_t ready = nullptr; bool ready_set = false;       // recorded after the upload: a frame on ANOTHER stream that reuses the slot waits for it
        size_t off_bytes = 0, map_bytes = 0;
    } tables[2];
    int tables_cur = 0;
    bool tables_valid = false;
    // per-band scratch of the pipeline.  Two sets + two internal streams: the two halves of a frame run concurrently, so that
    // the short, latency-bound tail kernels of one half (second shaft round, fallback walks) overlap the other half's work
    static constexpr int kMaxSplit = 4;
    struct BandScratch {
        DBuf hits, hits2, bounce_levels, bounce_nlev, bounce_prep, bounce_res, bounce_stack, samples, counters, fallback, fallback_state, fallback_rays, fallback_ovf, ray_sort, ray_sort_temp;
        DBuf accum;                        // escape counts per sample index of a chunked (> 128 samples) shadow stage; zero between frames
        DBuf tile_cost, tile_order;        // walk length per 8x8 tile of the last shaft launch / the next one's longest-first lists
        unsigned long long tile_order_tag = 0;   // the tile grid tile_order was made for (0: none)
        DBuf rlist[sr::kShaftRounds], rstate[sr::kShaftRounds], rcount[sr::kShaftRounds], rcand[sr::kShaftRounds];
        hipStream_t stream = nullptr;
        hipEvent_t  done = nullptr;
        bool used_last_frame = false;
        void release() {
            tile_cost.release(); tile_order.release(); tile_order_tag = 0;
            DBuf* b[] = {&hits, &hits2, &bounce_levels, &bounce_nlev, &bounce_prep, &bounce_res, &bounce_stack, &samples, &counters, &fallback, &fallback_state, &fallback_rays, &fallback_ovf, &ray_sort, &ray_sort_temp, &accum};
            for (DBuf* x : b) x->release();
            for (int r = 0; r < sr::kShaftRounds; ++r) { rlist[r].release(); rstate[r].release(); rcount[r].release(); rcand[r].release(); }
            if (stream) (void)hipStreamDestroy(stream);
            if (done) (void)hipEventDestroy(done);
            stream = nullptr; done = nullptr;
        }
    } scratch[kMaxSplit];
    hipEvent_t fork = nullptr;
    hipEvent_t multi_done = nullptr;     // (part of a multi-device scene) this part's strips of the current frame are rendered
    // RCCL strip gather: one communicator per process-per-GPU scene (sr_rccl_init), or one per part of a multi-device
    // scene (sr_set_gather(SR_GATHER_RCCL): ncclCommInitAll); rank 0 / the first part receives into d_gather (compact, rank after rank)
    sr::RcclComm comm = nullptr;
    int rccl_world = 0, rccl_rank = -1;
    std::vector<sr::RcclComm> comms;
    int gather_kind = 0;
    DBuf d_gather;
    // the blocking calls (sr_render) never use the null stream: frames are enqueued on io_stream, their way back to the host runs on
    // copy_stream band by band (an event per row band: a band is copied while the next ones render)
    hipStream_t io_stream = nullptr, copy_stream = nullptr;
    struct BandRec { hipEvent_t ev; int band, row_begin, row_count; };
    std::vector<hipEvent_t> band_ev_pool;
    size_t band_ev_used = 0;
    std::vector<BandRec> band_recs;
    bool collect_bands = false;
    bool can_peer = true;                // (part of a multi-device scene) the first part's device can read this part's memory
    void* stage_host = nullptr; size_t stage_cap = 0;   // (part without peer access) pinned staging of its strips
    hipEvent_t staged = nullptr;
    int num_cus = 0;
    bool tris_dirty = true, extra_dirty = true, ref_dirty = true, bvh_dirty = true;
    std::vector<double>  offsets_host;
    std::vector<int32_t> rowmap_host;
    // kernel timing: one HIP event pair per launch, accumulated until sr_reset_kernel_times()
    std::vector<hipEvent_t> ev[sr::K_COUNT];       // [2*i] start, [2*i+1] stop
    int  ev_used[sr::K_COUNT] = {};
    uint64_t last_stats[SR_STATS_COUNT] = {};
    int64_t dbg[SR_DBG_COUNT];             // sr_debug_set hooks, -1 = default (never read from the environment)
    sr_scene() { for (auto& d : dbg) d = -1; }
};

namespace {

const int kMaxShaftSamples = 1024;       // area-light samples the shaft path takes (in chunks of 128); more: one lane per hit point (k_shadow)
const int kMaxTreeDepth = 62;            // (depth + 2) stack levels x 256 lanes x 4 B = 64 KB of LDS per workgroup

int use_device(sr_scene* s) {
    if (s->device < 0) return fail(SR_ERR_NO_DEVICE, "host-only scene: no HIP device bound (compute is never emulated on the CPU)");
    SR_HIP(hipSetDevice(s->device));
    return SR_OK;
}

// axes on which `p` lies outside the root box's slab (by more than the shadow probe offset and the boxes' padding): known = those axes,
// beyond = those of them on which p lies ABOVE the box
void point_outside_axes(const sr::RootBox& root, const double p[3], int& known, int& beyond) {
    known = beyond = 0;
    for (int a = 0; a < 3; ++a) {
        const double margin = 0.01 + 1e-3 * (root.max[a] - root.min[a]);
        if (p[a] > root.max[a] + margin) { known |= 1 << a; beyond |= 1 << a; }
        else if (p[a] < root.min[a] - margin) known |= 1 << a;
    }
}

// the four-wide tree of the packet walks = the binary tree collapsed on the host (sr_host.cpp collapse_bvh4)
int upload_wide_tree(sr_scene* s, const sr::BvhNode* nodes, size_t num_nodes) {
    std::vector<sr::Bvh4Node> wide;
    s->b4_depth = sr::collapse_bvh4(nodes, num_nodes, wide);
    s->b4_num = wide.size();
    s->b4cam_valid = s->b4light_valid = false;
    s->part_valid = false;
    if (s->pre_used_set) SR_HIP(hipEventSynchronize(s->pre_used));       // a frame in flight may still be walking the old tree's copies
    SR_HIP(s->d_b4.upload(wide));
    SR_HIP(s->d_b4cam.reserve(wide.size() * sizeof(sr::Bvh4Node)));
    SR_HIP(s->d_b4light.reserve(wide.size() * sizeof(sr::Bvh4Node)));
    return SR_OK;
}

int sync_geometry(sr_scene* s, uint32_t need_mode) {
    const sr_scene* h = s->host_src ? s->host_src : s;              // where the host arrays are
    if (s->tris_dirty) { SR_HIP(s->d_tris.upload(h->tri_recs)); SR_HIP(s->d_v9.upload(h->v9)); s->tris_dirty = false; s->cam_valid = false; }
    if (s->extra_dirty) { SR_HIP(s->d_extra.upload(s->extra_recs)); s->extra_dirty = false; }
    if (need_mode == SR_MODE_REF_TREE && s->ref_dirty) {
        SR_HIP(s->d_rnodes.upload(h->ref.nodes));
        SR_HIP(s->d_rboxes.upload(h->ref.leaf_boxes));
        SR_HIP(s->d_rleaf.upload(h->ref.leaf_tris));
        s->ref_dirty = false;
    }
    if (need_mode == SR_MODE_BVH && s->bvh_dirty && !s->bvh_on_device) {
        SR_HIP(s->d_bnodes.upload(h->bvh.nodes));
        // the leaf-order copies of the FP64 records and the fp32 shaft records are made on the device from the TriangleIndex-order
        // arrays that are there already: only the order (4 B per triangle) travels
        const size_t n = h->bvh.order.size();
        DBuf d_order, d_slab;
        SR_HIP(d_order.upload(h->bvh.order));
        SR_HIP(d_slab.reserve(n * sizeof(sr::TriSlab)));
        SR_HIP(sr::make_slabs_device((const double*)s->d_v9.p, (int)n, s->root, (sr::TriSlab*)d_slab.p, nullptr));
        SR_HIP(s->d_btris.reserve(n * sizeof(sr::Rec128)));
        SR_HIP(s->d_bslab.reserve(n * sizeof(sr::TriSlab)));
        SR_HIP(sr::gather_records_device((int)n, (const unsigned int*)d_order.p, (const sr::Rec128*)s->d_tris.p, (sr::Rec128*)s->d_btris.p,
                                         (const sr::TriSlab*)d_slab.p, (sr::TriSlab*)s->d_bslab.p, nullptr));
        SR_HIP(hipDeviceSynchronize());
        d_order.release();
        d_slab.release();
        s->bvh_num_nodes = h->bvh.nodes.size();
        s->bvh_dirty = false;
        s->cam_valid = false;
        int rc = upload_wide_tree(s, h->bvh.nodes.data(), h->bvh.nodes.size());
        if (rc) return rc;
    }
    return SR_OK;
}

sr::DevScene dev_scene(const sr_scene* s) {
    sr::DevScene d{};
    d.tris = (const sr::Rec128*)s->d_tris.p; d.ntris = (int32_t)s->ntris;
    d.extra = (const sr::Rec128*)s->d_extra.p; d.nextra = (int32_t)s->extra_recs.size();
    d.rnodes = (const sr::RefNode*)s->d_rnodes.p; d.rboxes = (const sr::LeafBox*)s->d_rboxes.p; d.rleaf = (const int32_t*)s->d_rleaf.p;
    d.rdepth = s->ref.tree_depth;
    d.bnodes = (const sr::BvhNode*)s->d_bnodes.p; d.btris = (const sr::Rec128*)s->d_btris.p; d.bdepth = s->bvh.depth;
    d.bslab = (const sr::TriSlab*)s->d_bslab.p;
    d.bcam = s->cam_valid ? (const sr::CamCone*)s->d_bcam.p : nullptr;
    d.v9 = (const double*)s->d_v9.p;
    // the four-wide walks stack up to three entries per level: a private walk needs (3 depth + 2) x 1 KB of LDS per workgroup, a packet
    // walk (3 depth + 2) x 528 B; a tree too deep for that is walked in its binary form (a pathological scene, not a large one:
    // 10 M triangles give depth 14)
    const size_t lv4 = (size_t)(3 * s->b4_depth + 2);
    const bool wide_private = s->b4_num > 0 && lv4 * 1024 <= 65536, wide_packets = s->b4_num > 0 && lv4 * 528 <= 65536;
    d.b4 = wide_private ? (const sr::Bvh4Node*)s->d_b4.p : nullptr;
    d.b4cam = (wide_packets && s->b4cam_valid && s->cam_valid) ? (const sr::Bvh4Node*)s->d_b4cam.p : nullptr;
    d.b4light = (wide_packets && s->b4light_valid) ? (const sr::Bvh4Node*)s->d_b4light.p : nullptr;
    d.b4depth = s->b4_depth;
    d.b4cam_known = s->b4cam_known; d.b4light_known = s->b4light_known;
    d.bnode_bits = 1;
    while ((1ull << d.bnode_bits) < s->bvh_num_nodes + 1 && d.bnode_bits < 26) d.bnode_bits++;
    d.root = s->root;
    d.shadow_cache = (uint8_t*)s->d_shadow_cache.p;
    return d;
}

int check_mode(const sr_scene* s, int mode) {
    if (!s->have_model || s->ntris == 0) return fail(SR_ERR_NO_MODEL, "no model: Render() returns without drawing (Renderer.cs:736-739)");
    if (mode == SR_MODE_REF_TREE && !s->ref.built) return fail(SR_ERR_NOT_BUILT, "SR_MODE_REF_TREE needs sr_build(1 << SR_MODE_REF_TREE)");
    // one stack word per level and lane in LDS: (depth + 2) x 256 x 4 bytes must fit the 64 KB a workgroup may ask for
    if (mode == SR_MODE_REF_TREE && s->ref.tree_depth > kMaxTreeDepth) return fail(SR_ERR_UNSUPPORTED, "reference tree deeper than 62 levels does not fit the LDS traversal stacks");
    if (mode == SR_MODE_BVH && !s->bvh.built) return fail(SR_ERR_NOT_BUILT, "SR_MODE_BVH needs sr_build(1 << SR_MODE_BVH)");
    if (mode != SR_MODE_REF_TREE && mode != SR_MODE_BRUTE && mode != SR_MODE_BVH) return fail(SR_ERR_INVALID_ARG, "unknown trace mode");
    return SR_OK;
}

bool row_owned(const sr_frame* f, int r) {
    if (f->strip_count <= 0) return true;
    return ((r / f->strip_rows) % f->strip_count) == f->strip_index;
}

int validate_frame(const sr_frame* f) {
    if (!f) return fail(SR_ERR_INVALID_ARG, "frame is NULL");
    if (f->width <= 0 || f->height <= 0) return fail(SR_ERR_INVALID_ARG, "surface size must be positive");
    if (f->sub_pixel_res < 1 || f->sub_pixel_res > 64) return fail(SR_ERR_INVALID_ARG, "sub_pixel_res out of range");
    if (f->strip_count < 0 || (f->strip_count > 0 && (f->strip_rows <= 0 || f->strip_index < 0 || f->strip_index >= f->strip_count)))
        return fail(SR_ERR_INVALID_ARG, "bad strip parameters");
    if (f->shadow_samples < 0 || f->shadow_samples > 4096) return fail(SR_ERR_INVALID_ARG, "shadow_samples out of range");
    if ((long long)f->width * f->sub_pixel_res > (1ll << 24) || (long long)f->height > (1ll << 24)) return fail(SR_ERR_INVALID_ARG, "surface too large");
    if (f->max_bounces < 0 || f->max_bounces > 16 || !(f->reflectivity >= 0.0 && f->reflectivity <= 1.0))
        return fail(SR_ERR_INVALID_ARG, "max_bounces must be 0..16 and reflectivity 0..1");
    return SR_OK;
}

void clamp_rows(const sr_frame* f, int& a, int& b) {                  // Renderer.cs:1652-1653
    a = std::min(std::max(0, f->start_row), f->height - 1);
    b = std::min(std::max(0, f->end_row), f->height - 1);                 // b < a: the row loop does not run (Renderer.cs:1666)
}

// everything RaytraceGeometry derives per frame (Renderer.cs:1510-1528, 1717)
int prepare_frame(sr_scene* s, const sr_frame* f, sr::FrameConst& fc) {
    std::memset(&fc, 0, sizeof(fc));
    fc.width = f->width; fc.height = f->height;
    fc.sub_pixel_res = f->sub_pixel_res;
    fc.background = f->background_argb | 0xFF000000u;               // BackgroundColorWithAlpha, :325-331
    fc.flags = f->flags;
    fc.shadow_samples = f->shadow_samples > 0 ? f->shadow_samples : 100;   // ShadowMethod.cs:9
    fc.strip_rows = f->strip_rows; fc.strip_count = f->strip_count; fc.strip_index = f->strip_index;
    int a, b;
    clamp_rows(f, a, b);
    fc.start_row = a;
    s->rowmap_host.clear();
    for (int r = a; r <= b; ++r) if (row_owned(f, r)) s->rowmap_host.push_back(r);
    fc.num_rows = (int32_t)s->rowmap_host.size();
    fc.first_row = fc.num_rows ? s->rowmap_host[0] : 0;
    for (int i = 0; i < 12; ++i) { fc.t[i] = f->transform[i]; fc.it[i] = f->inv_transform[i]; }
    fc.position_z = f->position_z; fc.fov_depth = f->fov_depth;
    fc.focal_depth = f->focal_depth; fc.focal_blur_strength = f->focal_blur_strength;
    fc.ambient = f->ambient; fc.shininess = f->shininess;
    const double* it = fc.it;
    for (int i = 0; i < 3; ++i) { fc.light_dir_view[i] = f->light_dir_view[i]; fc.light_pos_view[i] = f->light_pos_view[i]; }
    const double* ld = f->light_dir_view; const double* lp = f->light_pos_view;
    for (int r = 0; r < 3; ++r) {
        // Instance.TransformDirectionReverse (Instance.cs:229-235) and TransformPosFromView (:192-209, which
        // IGNORES its un-projection and returns inverseTransform(3x4) * pos -- kept)
        fc.light_dir_model[r] = ld[0] * it[4 * r] + ld[1] * it[4 * r + 1] + ld[2] * it[4 * r + 2];
        fc.light_pos_model[r] = lp[0] * it[4 * r] + lp[1] * it[4 * r + 1] + lp[2] * it[4 * r + 2] + it[4 * r + 3];
        const double vx = 0.0, vy = 0.0, vz = -f->position_z;         // new Vector(0, 0, -instance.Position.z), :1717
        fc.start_world[r] = vx * it[4 * r] + vy * it[4 * r + 1] + vz * it[4 * r + 2];
    }
    fc.aspect = (double)f->height / (double)f->width;               // Renderer.cs:621
    fc.max_bounces = f->max_bounces;
    fc.reflectivity = f->reflectivity;

    // area-light offsets (ShadowMethod.cs:63-73)
    s->offsets_host.resize((size_t)fc.shadow_samples * 3);
    if (f->area_light_offsets) std::memcpy(s->offsets_host.data(), f->area_light_offsets, s->offsets_host.size() * sizeof(double));
    else sr::area_light_offsets(f->random_seed, fc.shadow_samples, s->offsets_host.data());
    double r2max = 0;
    for (int i = 0; i < fc.shadow_samples; ++i) {
        const double* o = &s->offsets_host[3 * i];
        r2max = std::max(r2max, o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    }
    fc.light_radius = std::sqrt(r2max);
    fc.debug = s->dbg[SR_DBG_KERNEL_SWITCH] > 0 ? (int32_t)s->dbg[SR_DBG_KERNEL_SWITCH] : 0;
    return SR_OK;
}

int ensure_io_streams(sr_scene* s) {
    if (!s->io_stream) SR_HIP(hipStreamCreateWithFlags(&s->io_stream, hipStreamNonBlocking));
    if (!s->copy_stream) SR_HIP(hipStreamCreateWithFlags(&s->copy_stream, hipStreamNonBlocking));
    return SR_OK;
}

// pins the caller's (pageable) surface for the duration of one blocking call, so that the band copies are real asynchronous DMA
// straight into it; the library does not keep the pointer.  Small frames and refusals (already registered, odd mappings) simply
// take the runtime's pageable path.
struct HostPin {
    void* p = nullptr;
    HostPin(void* ptr, size_t bytes) {
        if (bytes < ((size_t)4 << 20)) return;
        if (hipHostRegister(ptr, bytes, hipHostRegisterPortable) == hipSuccess) p = ptr;
        else (void)hipGetLastError();
    }
    ~HostPin() { if (p) (void)hipHostUnregister(p); }
};

const int kMaxTimedLaunches = 4096;

// hands out the event pair for the next launch of kernel k (nullptr once the pool is exhausted)
int next_events(sr_scene* s, int k, hipEvent_t& a, hipEvent_t& b) {
    a = b = nullptr;
    if (s->dbg[SR_DBG_KERNEL_TIMING] <= 0) return SR_OK;          // opt-in: no events inside an ordinary frame
    if (s->ev_used[k] >= kMaxTimedLaunches) return SR_OK;
    if ((size_t)(2 * s->ev_used[k] + 2) > s->ev[k].size()) {
        hipEvent_t e0, e1;
        SR_HIP(hipEventCreate(&e0));
        SR_HIP(hipEventCreate(&e1));
        s->ev[k].push_back(e0);
        s->ev[k].push_back(e1);
    }
    a = s->ev[k][2 * s->ev_used[k]];
    b = s->ev[k][2 * s->ev_used[k] + 1];
    s->ev_used[k]++;
    return SR_OK;
}

int render_common(sr_scene* s, const sr_frame* f, uint32_t* d_pixels, hipStream_t stream, unsigned long long* d_stats) {
    sr::FrameConst fc;
    int rc = prepare_frame(s, f, fc);
    if (rc) return rc;
    if ((rc = sync_geometry(s, (uint32_t)f->trace_mode))) return rc;
    if (fc.num_rows == 0) return SR_OK;
    // ---- frames of one scene run in submission order whatever streams they are given: the scene's scratch (hit queues, candidate
    //      lists, counters) and its per-origin / per-light records belong to one frame at a time.  `pre_used` is recorded when everything
    //      a frame enqueues is on its stream; the next frame's stream waits for it (a no-op on the same stream) ----
    if (s->pre_used_set) SR_HIP(hipStreamWaitEvent(stream, s->pre_used, 0));
    struct MarkPreUsed {
        sr_scene* s; hipStream_t st;
        ~MarkPreUsed() {
            if (!s->pre_used && hipEventCreateWithFlags(&s->pre_used, hipEventDisableTiming) != hipSuccess) return;
            if (hipEventRecord(s->pre_used, st) == hipSuccess) s->pre_used_set = true;
        }
    } mark_pre_used{s, stream};
    // ---- frame tables -> device ----
    const size_t off_bytes = (s->offsets_host.size() * sizeof(double) + 255) / 256 * 256, map_bytes = s->rowmap_host.size() * sizeof(int32_t);
    {
        sr_scene::FrameTables* T = &s->tables[s->tables_cur];
        const bool same = s->tables_valid && T->off_bytes == off_bytes && T->map_bytes == map_bytes && T->host &&
                          std::memcmp(T->host, s->offsets_host.data(), s->offsets_host.size() * sizeof(double)) == 0 &&
                          std::memcmp((const char*)T->host + off_bytes, s->rowmap_host.data(), map_bytes) == 0;
        if (!same) {
            s->tables_cur ^= 1;
            T = &s->tables[s->tables_cur];
            if (T->in_flight) { SR_HIP(hipEventSynchronize(T->used)); T->in_flight = false; }
            const size_t need = off_bytes + map_bytes;
            if (need > T->host_cap) {
                if (T->host) SR_HIP(hipHostFree(T->host));
                T->host = nullptr; T->host_cap = 0;
                SR_HIP(hipHostMalloc(&T->host, need + 4096, hipHostMallocDefault));
                T->host_cap = need + 4096;
            }
            SR_HIP(T->dev.reserve(need));
            std::memcpy(T->host, s->offsets_host.data(), s->offsets_host.size() * sizeof(double));
            std::memcpy((char*)T->host + off_bytes, s->rowmap_host.data(), map_bytes);
            T->off_bytes = off_bytes; T->map_bytes = map_bytes;
            SR_HIP(hipMemcpyAsync(T->dev.p, T->host, need, hipMemcpyHostToDevice, stream));
            if (!T->ready) SR_HIP(hipEventCreateWithFlags(&T->ready, hipEventDisableTiming));
            SR_HIP(hipEventRecord(T->ready, stream));
            T->ready_set = true;
            s->tables_valid = true;
        } else if (T->ready_set) {
            SR_HIP(hipStreamWaitEvent(stream, T->ready, 0));        // uploaded on another stream, perhaps
        }
    }
    sr_scene::FrameTables& FT = s->tables[s->tables_cur];
    const double* d_offsets = (const double*)FT.dev.p;
    const int32_t* d_rowmap = (const int32_t*)((const char*)FT.dev.p + off_bytes);
    struct MarkUsed {                                               // the slot is busy until everything enqueued below has run
        sr_scene::FrameTables& t; hipStream_t st;
        ~MarkUsed() {
            if (!t.used && hipEventCreateWithFlags(&t.used, hipEventDisableTiming) != hipSuccess) return;
            if (hipEventRecord(t.used, st) == hipSuccess) t.in_flight = true;
        }
    } mark_used{FT, stream};
    const bool static_shadows = (f->flags & SR_F_STATIC_SHADOWS) && (f->flags & SR_F_SHADOWS);
    if (static_shadows) {
        if ((f->flags & SR_F_SINGLE_KERNEL) || f->max_bounces > 0 || f->strip_count > 1)
            return fail(SR_ERR_UNSUPPORTED, "static shadows need the whole frame in one pipeline call (no strips, mirror bounces or SR_F_SINGLE_KERNEL)");
        SR_HIP(s->d_shadow_cache.reserve(sr::pipeline_static_cells()));
        SR_HIP(s->d_static_claim.reserve(sr::pipeline_static_cells() * 8));
        if (s->shadow_cache_empty) {
            SR_HIP(hipMemsetAsync(s->d_shadow_cache.p, 0, sr::pipeline_static_cells(), stream));
            s->shadow_cache_empty = false;
        }
    } else {
        fc.flags &= ~(uint32_t)SR_F_STATIC_SHADOWS;          // meaningless without SR_F_SHADOWS (RendererTests.cs:420)
    }
    // mirror bounces: wavefront pipeline (k_primary -> k_bounce per level -> k_fold) on the own BVH without shadows; every other
    // combination is traced inline by the one-kernel renderer
    const bool bounce_pipe = f->max_bounces > 0 && f->trace_mode == SR_MODE_BVH && !(f->flags & SR_F_SHADOWS) && !(f->flags & SR_F_SINGLE_KERNEL);
    if ((f->flags & SR_F_SINGLE_KERNEL) || (f->max_bounces > 0 && !bounce_pipe)) {
        sr::RenderLaunch L{};
        L.sc = dev_scene(s);
        L.fc = fc;
        L.mode = f->trace_mode;
        L.offsets = d_offsets;
        L.row_map = d_rowmap;
        L.pixels = d_pixels;
        L.stats = d_stats;
        L.stream = stream;
        hipEvent_t e0, e1;
        if ((rc = next_events(s, sr::K_RENDER, e0, e1))) return rc;
        if (e0) SR_HIP(hipEventRecord(e0, stream));
        SR_HIP(sr::launch_render(L));
        if (e1) SR_HIP(hipEventRecord(e1, stream));
        return SR_OK;
    }
    // ---- camera-cone records of the packet primary walk: one pre-pass per (tree, ray origin) ----
    // ---- two proofs that let a frame skip literal shadow rays without changing a byte ----
    // (1) directional light (ShadowMethod.cs:160-166): a sample ray starts at E' + dir * 1000 + offset and runs along +dir, AWAY from
    //     the surface point; whatever it could hit lies at least 1000 |dir| - R from E'.  When that exceeds the diagonal of everything
    //     a hit point or a triangle can be in, no sample of no hit point is occluded: rayEscapeCount = softShadowQuality, the factor
    //     is the constant (byte)(1.0 * 255) (applied by k_primary).  Extra geometry (unbounded planes) and the static cache (its
    //     cells are renderer state) keep the literal path.
    if ((fc.flags & SR_F_SHADOWS) && !(fc.flags & SR_F_POINT_LIGHT) && !static_shadows && s->extra_recs.empty() &&
        !(f->flags & (SR_F_SINGLE_KERNEL | SR_F_PER_LANE_SHADOWS)) && f->max_bounces == 0 && s->dbg[SR_DBG_LITERAL_SHADOWS] <= 0) {
        double diag2 = 0, len2 = 0;
        for (int a = 0; a < 3; ++a) {
            const double e = std::max(s->vmax[a], s->root.max[a]) - std::min(s->vmin[a], s->root.min[a]) + 0.004;   // + the probe offset, both ends
            diag2 += e * e;
            len2 += fc.light_dir_model[a] * fc.light_dir_model[a];
        }
        if (1000.0 * std::sqrt(len2) - fc.light_radius > std::sqrt(diag2) * 1.001 + 0.01) {
            fc.flags = (fc.flags & ~(uint32_t)SR_F_SHADOWS) | sr::kFlagAllSamplesEscape;
        }
    }
    // (2) a REF_TREE frame: the shadow rays only answer "is there a hit with rayFrac <= 1.0", which the own BVH answers identically
    //     (include/softray.h SR_MODE_BVH) -- they take the shaft path.  The four statistics of sr_render count the PRIMARY rays, which
    //     keep the literal traversal, so a caller that reads them loses nothing; SR_F_LITERAL_SECONDARY asks for the literal traversal
    //     of the shadow rays too (their counters in sr_last_ray_stats are then the reference tree's)
    const bool shadows_on_bvh = (fc.flags & SR_F_SHADOWS) && f->trace_mode == SR_MODE_REF_TREE && !(f->flags & SR_F_LITERAL_SECONDARY) && s->bvh.built && !static_shadows &&
                                (fc.flags & SR_F_POINT_LIGHT) && fc.shadow_samples <= kMaxShaftSamples && f->max_bounces == 0 &&
                                !(f->flags & (SR_F_SINGLE_KERNEL | SR_F_PER_LANE_SHADOWS)) && s->dbg[SR_DBG_LITERAL_SHADOWS] <= 0;
    if (shadows_on_bvh && (rc = sync_geometry(s, SR_MODE_BVH))) return rc;
    const bool bvh_walks = f->trace_mode == SR_MODE_BVH || shadows_on_bvh;
    const bool wide = bvh_walks && s->b4_num > 0 && s->dbg[SR_DBG_BVH2_PACKETS] <= 0;
    bool rewrote = false;                                             // (frames enqueued earlier have been waited for: see pre_used above)
    // ---- which records can the frame's camera rays / shadow sample rays hit at all?  (k_facing_partition, sr_pipeline.hip) ----
    const bool pkt_primary = f->trace_mode == SR_MODE_BVH && s->dbg[SR_DBG_PER_LANE_PRIMARY] <= 0 && !((f->flags & SR_F_FOCAL_BLUR) && f->sub_pixel_res > 1);
    const bool want_cam = wide && pkt_primary, want_light = wide && (fc.flags & SR_F_SHADOWS) && (fc.flags & SR_F_POINT_LIGHT);
    if ((want_cam || want_light) && s->dbg[SR_DBG_KERNEL_SWITCH] != 71) {
        const auto same3 = [](const double* a, const double* b) { return a[0] == b[0] && a[1] == b[1] && a[2] == b[2]; };
        const bool cam_ok = !want_cam || (s->part_cam && same3(s->part_origin, fc.start_world));
        const bool light_ok = !want_light || (s->part_light && same3(s->part_lightpos, fc.light_pos_model) && s->part_radius == fc.light_radius);
        if (!s->part_valid || !cam_ok || !light_ok) {
            SR_HIP(s->d_rng_cam.reserve(s->b4_num * 4 * 8));
            SR_HIP(s->d_rng_light.reserve(s->b4_num * 4 * 8));
            rewrote = true;
            s->part_valid = false;
            s->cam_valid = false; s->b4cam_valid = false; s->b4light_valid = false;      // the records move: cone records and both copies are re-made
            SR_HIP(sr::launch_facing_partition((const sr::Bvh4Node*)s->d_b4.p, (int)s->b4_num, (sr::Rec128*)s->d_btris.p, (sr::TriSlab*)s->d_bslab.p,
                                               fc.start_world, want_cam, fc.light_pos_model, fc.light_radius, want_light, s->d_rng_cam.p, s->d_rng_light.p, stream));
            s->part_cam = want_cam; s->part_light = want_light;
            for (int i = 0; i < 3; ++i) { s->part_origin[i] = fc.start_world[i]; s->part_lightpos[i] = fc.light_pos_model[i]; }
            s->part_radius = fc.light_radius;
            s->part_valid = true;
        }
    } else if (s->part_valid && s->dbg[SR_DBG_KERNEL_SWITCH] == 71) {
        s->part_valid = false; s->b4cam_valid = false; s->b4light_valid = false;        // (hook: no live runs -- the copies are re-made without them)
    }
    if (pkt_primary) {
        const size_t nt = s->ntris;
        const bool same_origin = s->cam_origin[0] == fc.start_world[0] && s->cam_origin[1] == fc.start_world[1] && s->cam_origin[2] == fc.start_world[2];
        if (!s->cam_valid || !same_origin) {
            SR_HIP(s->d_bcam.reserve(nt * sizeof(sr::CamCone)));
            rewrote = true;
            s->cam_valid = false;
            s->b4cam_valid = false;
            SR_HIP(sr::launch_cam_cones(dev_scene(s), (int)nt, fc.start_world, (sr::CamCone*)s->d_bcam.p, stream));
            for (int i = 0; i < 3; ++i) s->cam_origin[i] = fc.start_world[i];
            s->cam_valid = true;
        }
        if (wide && !s->b4cam_valid) {                                // the four-wide nodes, children front to back for this origin
            rewrote = true;
            int known, swap;                                      // (a camera ABOVE the box on an axis looks towards smaller coordinates: hi first)
            point_outside_axes(s->root, fc.start_world, known, swap);
            // the camera-ordered copy holds (near, far) planes only when that is true on ALL axes (one extra instantiation of k_primary, not seven)
            if (known != 7 || s->dbg[SR_DBG_KERNEL_SWITCH] == 61) known = swap = 0;
            SR_HIP(sr::launch_order_nodes((const sr::Bvh4Node*)s->d_b4.p, (sr::Bvh4Node*)s->d_b4cam.p, (int)s->b4_num, s->root, fc.start_world, false, swap,
                                          (s->part_valid && s->part_cam) ? s->d_rng_cam.p : nullptr, stream));
            s->b4cam_known = known;
            s->b4cam_valid = true;
        }
    }
    if (wide && (fc.flags & SR_F_SHADOWS) && (fc.flags & SR_F_POINT_LIGHT)) {
        const bool same_light = s->b4_light[0] == fc.light_pos_model[0] && s->b4_light[1] == fc.light_pos_model[1] && s->b4_light[2] == fc.light_pos_model[2];
        if (!s->b4light_valid || !same_light) {                       // ... and nearest-to-the-surface first for this light
            rewrote = true;
            s->b4light_valid = false;
            int known, beyond;                                    // (a light BELOW the box on an axis: every shaft travels towards smaller coordinates there, hi first)
            point_outside_axes(s->root, fc.light_pos_model, known, beyond);
            if (s->dbg[SR_DBG_KERNEL_SWITCH] == 62) known = 0;                    // (hook: (lo, hi) planes on every axis)
            SR_HIP(sr::launch_order_nodes((const sr::Bvh4Node*)s->d_b4.p, (sr::Bvh4Node*)s->d_b4light.p, (int)s->b4_num, s->root, fc.light_pos_model, true, known & ~beyond,
                                          (s->part_valid && s->part_light) ? s->d_rng_light.p : nullptr, stream));
            s->b4light_known = known;
            for (int i = 0; i < 3; ++i) s->b4_light[i] = fc.light_pos_model[i];
            s->b4light_valid = true;
        }
    }
    if (rewrote) {
        if (!s->pre_ready) SR_HIP(hipEventCreateWithFlags(&s->pre_ready, hipEventDisableTiming));
        SR_HIP(hipEventRecord(s->pre_ready, stream));
        s->pre_ready_set = true;
    } else if (s->pre_ready_set) {
        SR_HIP(hipStreamWaitEvent(stream, s->pre_ready, 0));          // written on another stream, perhaps: order this frame after it
    }
    // ---- default: the primary / shadow / resolve pipeline, in row bands ----
    const long long n2 = (long long)fc.sub_pixel_res * fc.sub_pixel_res;
    const bool shadows = (fc.flags & SR_F_SHADOWS) != 0;
    // (more than 128 samples: the shaft path runs in chunks of 128, escape counts summed per hit point; not for the static cache)
    const bool shaft = shadows && bvh_walks && (fc.flags & SR_F_POINT_LIGHT) && !(f->flags & SR_F_PER_LANE_SHADOWS) &&
                       (fc.shadow_samples <= 128 || (fc.shadow_samples <= kMaxShaftSamples && !static_shadows));
    const bool chunked_shadows = shaft && fc.shadow_samples > 128;
    // samples per band: bounds the hit queue (64 B/sample) and, on the shaft path, the candidate lists (256 B/sample for
    // round 0 + 1/4 of the hits x 1 KB for round 1): 16 Mi samples = one 4096^2 frame = 10 GB of scratch in HBM
    long long kMaxBandSamples = shaft ? (16ll << 20) : (32ll << 20);
    int round_cap[sr::kShaftRounds];
    for (int r = 0; r < sr::kShaftRounds; ++r) round_cap[r] = sr::pipeline_round_cap(r);
    {   // test hooks (sr_debug_set): shrink the bands / candidate lists so that small frames exercise banding, round 2 and the fallback
        if (s->dbg[SR_DBG_BAND_SAMPLES] > 0) kMaxBandSamples = s->dbg[SR_DBG_BAND_SAMPLES];
        if (s->dbg[SR_DBG_ROUND_CAP0] > 0) round_cap[0] = (int)std::min<int64_t>(s->dbg[SR_DBG_ROUND_CAP0], sr::pipeline_round_cap_max(0));
        if (s->dbg[SR_DBG_ROUND_CAP1] > 0) round_cap[1] = (int)std::min<int64_t>(s->dbg[SR_DBG_ROUND_CAP1], sr::pipeline_round_cap_max(1));
        if (s->dbg[SR_DBG_EXACT_SHADOW_TESTS] > 0) round_cap[1] = std::min(round_cap[1], 64);   // k_shadow_test keeps a list in one wave's registers
    }
    // Two halves of the frame (16-row granularity) run as two pipelines on two internal streams, each with its own scratch
    // set; a half that exceeds its share of the band budget is processed in sequential bands on its stream.  A static
    // frame (one global fill order) and shadow-less frames (one kernel) stay whole on the first set.
    int want_split = 2;
    if (s->dbg[SR_DBG_SPLIT] > 0) want_split = (int)std::min<int64_t>(s->dbg[SR_DBG_SPLIT], (int)sr_scene::kMaxSplit);   // experiment hook
    const bool split = (shadows || bounce_pipe) && !static_shadows && want_split > 1 && fc.num_rows >= 32 * want_split && !(f->flags & SR_F_NO_SPLIT);
    const int halves = split ? want_split : 1;
    const int rows_half = split ? (int)((((long long)fc.num_rows + halves - 1) / halves + 15) / 16 * 16) : fc.num_rows;
    const long long budget = kMaxBandSamples / halves;
    // queue capacity counts whole 16x16-pixel tiles: the tile-aligned hit queue of the shaft path gives every wave (8x8 pixels
    // x one sub-sample) 64 entries, also at the right / bottom edge of the frame
    const long long wpad = ((long long)fc.width + 15) / 16 * 16;
    long long band_rows = std::max<long long>(16, (budget / (wpad * n2)) / 16 * 16);
    band_rows = std::min<long long>(band_rows, ((long long)rows_half + 15) / 16 * 16);
    const long long band_samples = band_rows * wpad * n2;
    if (shaft && band_samples >= (1ll << 25)) return fail(SR_ERR_UNSUPPORTED, "row band too large for the 25-bit fallback entry ids (surface too wide for this sub-pixel resolution)");
    if (static_shadows) {
        if (band_rows < fc.num_rows) return fail(SR_ERR_UNSUPPORTED, "static shadows: the frame does not fit one row band");
        SR_HIP(s->d_static_hits.reserve((size_t)std::min<long long>(band_samples, (long long)sr::pipeline_static_cells()) * sr::pipeline_hit_record_bytes()));
    }
    // fallback ray list: one 32-bit id (entry << 7 | sample; bands have < 2^25 entries) per undecided sample.  6 per band
    // sample = 0.4 GB for a 4096^2 frame; whatever it has no room for is taken by the one-wave-per-hit kernel.  Test hook
    // SR_FB_RAY_CAP shrinks it
    long long fallback_ray_cap = std::min<long long>(6 * band_samples, 0xfffffff0ll);
    if (s->dbg[SR_DBG_FB_RAY_CAP] > 0) fallback_ray_cap = s->dbg[SR_DBG_FB_RAY_CAP];
    unsigned round_items[sr::kShaftRounds] = {};
    for (int r = 0; r < sr::kShaftRounds; ++r)      // round 0 sees every hit; each later round is provisioned for 1/4 of the previous one
        round_items[r] = r == 0 ? (unsigned)band_samples : (unsigned)std::max<long long>(1024, (long long)round_items[r - 1] / 4);
    if (!s->num_cus) {
        hipDeviceProp_t prop;
        SR_HIP(hipGetDeviceProperties(&prop, s->device));
        s->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    if (split) {
        if (!s->fork) SR_HIP(hipEventCreateWithFlags(&s->fork, hipEventDisableTiming));
        SR_HIP(hipEventRecord(s->fork, stream));                    // the halves start after the caller's earlier work
    }
    for (auto& sc : s->scratch) sc.used_last_frame = false;
    for (int h = 0; h < halves; ++h) {
        sr_scene::BandScratch& B = s->scratch[h];
        B.used_last_frame = true;
        if (shadows || bounce_pipe) SR_HIP(B.hits.reserve((size_t)band_samples * sr::pipeline_hit_record_bytes()));
        if (bounce_pipe) {
            // level colours are indexed like the sample buffer: the frame (or compact strip buffer) for one sample per pixel, band-local otherwise
            const size_t idx_space = n2 == 1 ? (size_t)(f->strip_count > 0 ? fc.num_rows : fc.height) * fc.width : (size_t)band_samples;
            SR_HIP(B.hits2.reserve((size_t)band_samples * sr::pipeline_hit_record_bytes()));
            SR_HIP(B.bounce_levels.reserve(idx_space * (size_t)(f->max_bounces + 1) * 4));
            SR_HIP(B.bounce_nlev.reserve(idx_space));
            SR_HIP(B.bounce_prep.reserve((size_t)band_samples * 64));
            SR_HIP(B.bounce_res.reserve((size_t)band_samples * 16));
            // (k_bounce_walk keeps sr::pipeline_bounce_lds_levels() stack levels per lane in LDS, the rest of the worst case here)
            const long long deep = std::max(3 * (long long)s->b4_depth + 2, (long long)s->bvh.depth + 2) - sr::pipeline_bounce_lds_levels();
            if (deep > 0) SR_HIP(B.bounce_stack.reserve((size_t)deep * (size_t)s->num_cus * 8 * 256 * 4));
            // per-level ray order (keys, sorted keys, indices, order) + the device sort's own scratch
            SR_HIP(B.ray_sort.reserve((size_t)band_samples * 4 * 4));
            SR_HIP(B.ray_sort_temp.reserve(sr::ray_sort_temp_bytes((unsigned)band_samples)));
        }
        // (one band per part-frame pipeline: a second band would walk other tiles with the first one's lists)
        const bool order_tiles = shaft && band_rows >= rows_half;
        if (order_tiles) {
            const size_t bytes = sr::pipeline_tile_items(fc.width, (int)band_rows, (int)n2) * 4;
            if (bytes > B.tile_cost.cap) B.tile_order_tag = 0;
            SR_HIP(B.tile_cost.reserve(bytes));
            SR_HIP(B.tile_order.reserve(bytes));
        }
        if (shaft) {
            SR_HIP(B.fallback.reserve((size_t)band_samples * 4));
            SR_HIP(B.fallback_state.reserve((size_t)band_samples * sr::pipeline_round_state_bytes()));
            SR_HIP(B.fallback_rays.reserve((size_t)fallback_ray_cap * 4));
            SR_HIP(B.fallback_ovf.reserve((size_t)band_samples * 4));
            for (int r = 0; r < sr::kShaftRounds; ++r) {
                SR_HIP(B.rcount[r].reserve((size_t)round_items[r] * 4));
                SR_HIP(B.rcand[r].reserve((size_t)round_items[r] * round_cap[r] * 4));
                SR_HIP(B.rlist[r].reserve((size_t)round_items[r] * 4));     // round 0: the hits k_shaft left undecided
                if (r > 0) SR_HIP(B.rstate[r].reserve((size_t)round_items[r] * sr::pipeline_round_state_bytes()));
            }
        }
        if (n2 > 1) SR_HIP(B.samples.reserve((size_t)band_samples * 4));
        bool accum_fresh = false;
        if (chunked_shadows) {
            // indexed like the sample buffer: the frame (or the compact strips) for one sample per pixel, band-local otherwise
            const size_t idx_space = n2 == 1 ? (size_t)(f->strip_count > 0 ? fc.num_rows : fc.height) * fc.width : (size_t)band_samples;
            if (idx_space * 4 > B.accum.cap || !B.accum.p) {
                SR_HIP(B.accum.reserve(idx_space * 4));
                accum_fresh = true;                                // zeroed on the half's own stream below
            }
        }
        SR_HIP(B.counters.reserve(sr::pipeline_counter_bytes()));
        hipStream_t bs = stream;
        // the blocking call runs its part-frame pipelines one after the other on the caller's stream: a finished part travels to the host while
        // the next one renders (side by side both end together and all 64 MiB travel after the last kernel: 9.96 against 9.74 ms; the
        // persistent shaft walk of one part fills the chip anyway).  Hook 85: side by side, as the pipelined frames run
        const bool sequential_parts = s->collect_bands && s->dbg[SR_DBG_KERNEL_SWITCH] != 85;
        if (split && !sequential_parts) {
            if (!B.stream) SR_HIP(hipStreamCreateWithFlags(&B.stream, hipStreamNonBlocking));
            if (!B.done) SR_HIP(hipEventCreateWithFlags(&B.done, hipEventDisableTiming));
            bs = B.stream;
            SR_HIP(hipStreamWaitEvent(bs, s->fork, 0));
        }
        sr::PipelineLaunch P{};
        P.sc = dev_scene(s);
        P.fc = fc;
        P.fc.accum = chunked_shadows ? (uint32_t*)B.accum.p : nullptr;
        P.mode = f->trace_mode;
        P.offsets = d_offsets;
        P.row_map = d_rowmap;
        P.pixels = d_pixels;
        P.samples = (uint32_t*)B.samples.p;
        P.hits = B.hits.p;
        P.hits2 = bounce_pipe ? B.hits2.p : nullptr;
        P.ray_sort_buf = bounce_pipe ? (unsigned int*)B.ray_sort.p : nullptr;
        P.ray_sort_temp = bounce_pipe ? B.ray_sort_temp.p : nullptr;
        P.ray_sort_temp_bytes = bounce_pipe ? sr::ray_sort_temp_bytes((unsigned)band_samples) : 0;
        P.bounce_levels = bounce_pipe ? (uint32_t*)B.bounce_levels.p : nullptr;
        P.bounce_nlev = bounce_pipe ? (uint8_t*)B.bounce_nlev.p : nullptr;
        P.bounce_prep = bounce_pipe ? B.bounce_prep.p : nullptr;
        P.bounce_res = bounce_pipe ? B.bounce_res.p : nullptr;
        P.bounce_stack = bounce_pipe ? (int32_t*)B.bounce_stack.p : nullptr;
        P.bounce_stack_bytes = bounce_pipe ? B.bounce_stack.cap : 0;
        P.counters = (unsigned int*)B.counters.p;
        P.tile_cost = order_tiles ? (unsigned int*)B.tile_cost.p : nullptr;
        P.tile_order = order_tiles ? (unsigned int*)B.tile_order.p : nullptr;
        P.tile_order_tag = order_tiles ? &B.tile_order_tag : nullptr;
        P.static_hits = static_shadows ? s->d_static_hits.p : nullptr;
        P.static_claim = static_shadows ? (unsigned long long*)s->d_static_claim.p : nullptr;
        P.static_concurrency = f->concurrency;
        P.fallback = shaft ? (unsigned int*)B.fallback.p : nullptr;
        P.fallback_state = shaft ? B.fallback_state.p : nullptr;
        P.fallback_rays = shaft ? (unsigned int*)B.fallback_rays.p : nullptr;
        P.fallback_ray_cap = (unsigned int)fallback_ray_cap;
        P.fallback_overflow = shaft ? (unsigned int*)B.fallback_ovf.p : nullptr;
        for (int r = 0; r < sr::kShaftRounds; ++r) {
            P.round_items[r] = round_items[r];
            P.round_cap[r] = round_cap[r];
            P.round_list[r] = shaft ? (unsigned int*)B.rlist[r].p : nullptr;
            P.round_state[r] = (shaft && r > 0) ? B.rstate[r].p : nullptr;
            P.round_cand_count[r] = shaft ? (unsigned int*)B.rcount[r].p : nullptr;
            P.round_cand[r] = shaft ? (int32_t*)B.rcand[r].p : nullptr;
        }
        P.band_rows = (int32_t)band_rows;
        P.row_first = h * rows_half;
        P.row_limit = std::min(fc.num_rows, (h + 1) * rows_half);
        // workgroups per CU of the persistent shaft walk: two pipelines side by side leave each other room (a rank of an 8-way split: 1.41 -> 1.32 ms,
        // whole frame unchanged); a pipeline that has the chip to itself fills it (3.9 against 5.5 ms for the walk alone)
        P.shaft_wgs_per_cu = (split && !sequential_parts) ? 3 : 6;
        P.persistent_blocks = s->num_cus * 8;
        P.per_lane_shadows = (f->flags & SR_F_PER_LANE_SHADOWS) != 0;
        P.exact_shadow_tests = s->dbg[SR_DBG_EXACT_SHADOW_TESTS] > 0;
        P.per_lane_shaft = s->dbg[SR_DBG_PER_LANE_SHAFT] > 0 ? (int32_t)(s->dbg[SR_DBG_PER_LANE_SHAFT] & 3) : 0;
        P.per_lane_primary = s->dbg[SR_DBG_PER_LANE_PRIMARY] > 0;
        P.bvh2_packets = s->dbg[SR_DBG_BVH2_PACKETS] > 0;
        P.shadows_on_bvh = shadows_on_bvh;
        P.primary_stats_only = (f->flags & SR_F_PRIMARY_STATS_ONLY) != 0;
        P.round2_node_budget = s->dbg[SR_DBG_ROUND2_NODES] >= 0 ? (int32_t)std::min<int64_t>(s->dbg[SR_DBG_ROUND2_NODES], 1 << 30) : 0;
        P.stats = d_stats;
        P.stream = bs;
        P.user = s;
        P.get_events = [](void* user, int kid, hipEvent_t* a, hipEvent_t* b) {
            hipEvent_t x = nullptr, y = nullptr;
            if (next_events((sr_scene*)user, kid, x, y) != SR_OK) { x = y = nullptr; }
            *a = x; *b = y;
        };
        P.band_done = nullptr;
        if (s->collect_bands) P.band_done = [](void* user, int band, int row_begin, int row_count, hipStream_t st) {
            sr_scene* sc = (sr_scene*)user;
            if (sc->band_ev_used == sc->band_ev_pool.size()) {
                hipEvent_t e = nullptr;
                if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) { sc->collect_bands = false; return; }   // sr_render falls back to one copy
                sc->band_ev_pool.push_back(e);
            }
            hipEvent_t e = sc->band_ev_pool[sc->band_ev_used++];
            if (hipEventRecord(e, st) != hipSuccess) { sc->collect_bands = false; return; }
            sc->band_recs.push_back({e, band, row_begin, row_count});
        };
        if (accum_fresh) SR_HIP(hipMemsetAsync(B.accum.p, 0, B.accum.cap, bs));
        if (P.row_first < P.row_limit) SR_HIP(sr::launch_pipeline(P));
        if (split && !sequential_parts) {
            SR_HIP(hipEventRecord(B.done, bs));
            SR_HIP(hipStreamWaitEvent(stream, B.done, 0));          // the caller's stream continues after both halves
        }
    }
    return SR_OK;
}


// ------------------------------------------------------------------------------------------------------------------
// One scene over several devices of this process (sr_create_multi).  Rows are independent (the reference itself fans out row
// blocks, Renderer.cs:1659-1670): part g renders the 16-row strips s with s % n == g of the frame's row range into its own
// compact device buffer (sr_frame.strip_*), concurrently with the others (every part has its own device, streams and
// scratch; the calls below only ENQUEUE), and the strips travel straight to where the caller wants them: sr_render copies
// every part's strips device -> host into the caller's surface (each over its own PCIe link), sr_render_device copies them
// peer-to-peer (xGMI) into the caller's device surface.  No reduction, no RNG: the frame does not depend on the split.
// ------------------------------------------------------------------------------------------------------------------
const int kMultiStripRows = 16;

struct StripRun { int64_t compact_row, image_row, rows; };          // rows [image_row, image_row + rows) sit at compact_row of the part's buffer

// the runs of part g for the clamped row range [a, b]
std::vector<StripRun> strip_runs(int a, int b, int n, int g) {
    std::vector<StripRun> runs;
    int64_t compact = 0;
    for (int s0 = a / kMultiStripRows; s0 * kMultiStripRows <= b; ++s0) {
        if (s0 % n != g) continue;
        const int r0 = std::max(a, s0 * kMultiStripRows), r1 = std::min(b, s0 * kMultiStripRows + kMultiStripRows - 1);
        if (r1 < r0) continue;
        runs.push_back({compact, r0, r1 - r0 + 1});
        compact += r1 - r0 + 1;
    }
    return runs;
}

// copy the runs of one part to the full surface `dst` (host or device memory): the full strips between the (possibly
// partial) first and last one form an arithmetic progression -> ONE strided 2-D copy; the edges are copied on their own
hipError_t copy_runs(const std::vector<StripRun>& runs, int n, int width, const uint32_t* src, uint32_t* dst, hipMemcpyKind kind, hipStream_t st) {
    const size_t row_bytes = (size_t)width * 4;
    size_t i = 0;
    while (i < runs.size()) {
        size_t j = i;
        while (j + 1 < runs.size() && runs[j].rows == kMultiStripRows && runs[j + 1].rows == kMultiStripRows &&
               runs[j + 1].image_row - runs[j].image_row == (int64_t)n * kMultiStripRows) ++j;
        if (j > i) {
            const size_t strip_bytes = row_bytes * kMultiStripRows;
            hipError_t e = hipMemcpy2DAsync(dst + runs[i].image_row * width, strip_bytes * n, src + runs[i].compact_row * width, strip_bytes,
                                            strip_bytes, j - i + 1, kind, st);
            if (e != hipSuccess) return e;
            i = j + 1;
        } else {
            hipError_t e = hipMemcpyAsync(dst + runs[i].image_row * width, src + runs[i].compact_row * width, row_bytes * runs[i].rows, kind, st);
            if (e != hipSuccess) return e;
            ++i;
        }
    }
    return hipSuccess;
}

// frames that cannot be split (one global fill order) are rendered whole by the first part
bool multi_splittable(const sr_frame* f) { return !((f->flags & SR_F_STATIC_SHADOWS) && (f->flags & SR_F_SHADOWS)) && f->strip_count <= 0; }

// a part of a multi-device scene takes the first part's model by reference: counts, box and flags here, the arrays stay with `src`
// (sync_geometry uploads from them); nothing of the size of the model is copied on the host
void share_host_model(sr_scene* d, const sr_scene* src) {
    d->host_src = src;
    d->v9.clear(); d->argb.clear(); d->tri_recs.clear();
    d->ntris = src->ntris;
    for (int a = 0; a < 3; ++a) { d->bmin[a] = src->bmin[a]; d->bmax[a] = src->bmax[a]; d->vmin[a] = src->vmin[a]; d->vmax[a] = src->vmax[a]; }
    d->have_model = src->have_model;
    d->root = src->root;
    d->shadow_cache_empty = true;
    d->ref = sr::RefTree(); d->bvh = sr::Bvh(); d->bvh_on_device = false;
    d->tris_dirty = d->ref_dirty = d->bvh_dirty = true;
    d->cam_valid = false;
}
// ... and its host-built structures: the numbers a part needs (built / depth / counts), not the node arrays
void share_ref_tree(sr_scene* d, const sr_scene* src) {
    d->ref = sr::RefTree();
    d->ref.built = src->ref.built; d->ref.tree_depth = src->ref.tree_depth; d->ref.num_nodes = src->ref.num_nodes;
    d->ref.num_leaf_nodes = src->ref.num_leaf_nodes; d->ref.max_stack = src->ref.max_stack;
    d->ref_dirty = true;
}
void share_host_bvh(sr_scene* d, const sr_scene* src) {
    d->bvh = sr::Bvh();
    d->bvh.built = src->bvh.built; d->bvh.depth = src->bvh.depth;
    d->bvh_on_device = false;
    d->bvh_dirty = true;
}

}  // namespace

extern "C" {

int32_t sr_abi_version(void) { return SR_ABI_VERSION; }
const char* sr_last_error(void) { return g_err.c_str(); }

int sr_create(int32_t device, sr_scene** out) {
    if (!out) return fail(SR_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (device >= 0) {
        int count = 0;
        hipError_t e = hipGetDeviceCount(&count);
        if (e != hipSuccess || count <= 0) return fail(SR_ERR_NO_DEVICE, "no HIP device available (libsoftray_hip has no CPU fallback)");
        if (device >= count) return fail(SR_ERR_INVALID_ARG, "device ordinal out of range");
        SR_HIP(hipSetDevice(device));
    } else if (device != -1) {
        return fail(SR_ERR_INVALID_ARG, "device must be >= 0, or -1 for a host-only scene");
    }
    sr_scene* s = new sr_scene();
    s->device = device;
    *out = s;
    return SR_OK;
}

int sr_create_multi(const int32_t* devices, int32_t n, sr_scene** out) {
    if (!out) return fail(SR_ERR_INVALID_ARG, "out is NULL");
    *out = nullptr;
    if (!devices || n < 1 || n > 64) return fail(SR_ERR_INVALID_ARG, "sr_create_multi needs 1..64 device ordinals");
    sr_scene* m = new sr_scene();
    m->device = -2;
    for (int i = 0; i < n; ++i) {
        sr_scene* part = nullptr;
        int rc = devices[i] >= 0 ? sr_create(devices[i], &part) : fail(SR_ERR_INVALID_ARG, "device ordinals must be >= 0");
        if (rc) { for (sr_scene* q : m->parts) sr_destroy(q); delete m; return rc; }
        m->parts.push_back(part);
    }
    // sr_render_device gathers the strips into a surface on the first part's device: peer-to-peer where that device can read the
    // part's memory (asked and enabled here, the answer kept per part), through pinned host staging where it cannot
    for (int i = 1; i < n; ++i) {
        sr_scene* q = m->parts[i];
        if (devices[i] == devices[0]) { q->can_peer = true; continue; }
        int can = 0;
        q->can_peer = false;
        if (hipDeviceCanAccessPeer(&can, devices[0], devices[i]) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (!can) continue;
        hipError_t e = hipSetDevice(devices[0]);
        if (e == hipSuccess) e = hipDeviceEnablePeerAccess(devices[i], 0);
        if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) { q->can_peer = true; (void)hipGetLastError(); continue; }
        // the runtime said the devices can be peers and then refused: report it instead of running with half a configuration
        for (sr_scene* p : m->parts) sr_destroy(p);
        delete m;
        return hip_fail(e, "hipDeviceEnablePeerAccess");
    }
    *out = m;
    return SR_OK;
}

int32_t sr_device_count(const sr_scene* s) { return !s ? 0 : (s->parts.empty() ? (s->device >= 0 ? 1 : 0) : (int32_t)s->parts.size()); }

void sr_destroy(sr_scene* s) {
    if (!s) return;
    if (!s->parts.empty()) {
        if (!s->comms.empty()) {
            const sr::RcclApi* api = sr::rccl_api(nullptr);
            for (size_t g = 0; api && g < s->comms.size(); ++g)
                if (s->comms[g] && use_device(s->parts[g]) == SR_OK) (void)api->CommDestroy(s->comms[g]);
        }
        if (s->multi_done && use_device(s->parts[0]) == SR_OK) (void)hipEventDestroy(s->multi_done);
        if (use_device(s->parts[0]) == SR_OK) s->d_gather.release();
        for (sr_scene* q : s->parts) sr_destroy(q);
        delete s;
        return;
    }
    if (s->device >= 0 && hipSetDevice(s->device) == hipSuccess) {
        DBuf* bufs[] = {&s->d_tris, &s->d_extra, &s->d_rnodes, &s->d_rboxes, &s->d_rleaf, &s->d_bnodes, &s->d_btris, &s->d_bslab,
                        &s->d_v9, &s->d_bcam, &s->d_b4, &s->d_b4cam, &s->d_b4light, &s->d_rng_cam, &s->d_rng_light, &s->d_shadow_cache, &s->d_static_claim, &s->d_static_hits, &s->d_pixels, &s->d_aa, &s->d_stats};
        for (DBuf* b : bufs) b->release();
        for (auto& sc : s->scratch) sc.release();
        for (auto& t : s->tables) { t.dev.release(); if (t.host) (void)hipHostFree(t.host); if (t.used) (void)hipEventDestroy(t.used); if (t.ready) (void)hipEventDestroy(t.ready); }
        if (s->fork) (void)hipEventDestroy(s->fork);
        if (s->io_stream) (void)hipStreamDestroy(s->io_stream);
        if (s->copy_stream) (void)hipStreamDestroy(s->copy_stream);
        for (hipEvent_t e : s->band_ev_pool) (void)hipEventDestroy(e);
        if (s->stage_host) (void)hipHostFree(s->stage_host);
        if (s->staged) (void)hipEventDestroy(s->staged);
        if (s->pre_ready) (void)hipEventDestroy(s->pre_ready);
        if (s->pre_used) (void)hipEventDestroy(s->pre_used);
        if (s->multi_done) (void)hipEventDestroy(s->multi_done);
        for (DBuf& b : s->d_io) b.release();
        s->d_gather.release();
        if (s->comm) { const sr::RcclApi* api = sr::rccl_api(nullptr); if (api) (void)api->CommDestroy(s->comm); s->comm = nullptr; }
        for (int k = 0; k < sr::K_COUNT; ++k)
            for (hipEvent_t e : s->ev[k]) (void)hipEventDestroy(e);
    }
    delete s;
}

int sr_set_triangles(sr_scene* s, const double* v9, const uint32_t* argb, int64_t n, const double box_min[3], const double box_max[3]) {
    if (s && !s->parts.empty()) {                                  // host work once, the records are replicated
        int rc = sr_set_triangles(s->parts[0], v9, argb, n, box_min, box_max);
        for (size_t i = 1; i < s->parts.size() && !rc; ++i) share_host_model(s->parts[i], s->parts[0]);
        return rc;
    }
    if (!s || n < 0 || (n > 0 && (!v9 || !argb)) || !box_min || !box_max) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_set_triangles");
    if (n > 0x7fffff00) return fail(SR_ERR_INVALID_ARG, "too many triangles");
    s->shadow_cache_empty = true;                         // new model: what a new ShadowMethod starts with
    s->v9.assign(v9, v9 + 9 * n);
    s->argb.assign(argb, argb + n);
    for (int a = 0; a < 3; ++a) { s->bmin[a] = box_min[a]; s->bmax[a] = box_max[a]; }
    s->root = sr::make_root_box(box_min, box_max);
    for (int a = 0; a < 3; ++a) { s->vmin[a] = box_min[a]; s->vmax[a] = box_max[a]; }
    for (int64_t i = 0; i < 3 * n; ++i)
        for (int a = 0; a < 3; ++a) { s->vmin[a] = std::min(s->vmin[a], v9[3 * i + a]); s->vmax[a] = std::max(s->vmax[a], v9[3 * i + a]); }
    s->ntris = (size_t)n;
    s->tri_recs.resize((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const double* p = &s->v9[9 * i];
        s->tri_recs[i] = sr::make_triangle_record({p[0], p[1], p[2]}, {p[3], p[4], p[5]}, {p[6], p[7], p[8]}, argb[i], (int32_t)i);
    }
    s->have_model = true;
    s->ref = sr::RefTree();
    s->bvh = sr::Bvh();
    s->bvh_on_device = false;
    s->tris_dirty = s->ref_dirty = s->bvh_dirty = true;
    return SR_OK;
}

int sr_set_extra_geometry(sr_scene* s, const sr_prim* prims, int32_t n) {
    if (s && !s->parts.empty()) {
        for (sr_scene* q : s->parts) { int rc = sr_set_extra_geometry(q, prims, n); if (rc) return rc; }
        return SR_OK;
    }
    if (!s || n < 0 || (n > 0 && !prims)) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_set_extra_geometry");
    std::vector<sr::Rec128> recs;
    for (int i = 0; i < n; ++i) {
        const sr_prim& q = prims[i];
        switch (q.kind) {
            case 0:
                if (!(q.p[3] > 0)) return fail(SR_ERR_INVALID_ARG, "sphere radius must be > 0 (Sphere.cs:28)");
                recs.push_back(sr::make_sphere_record({q.p[0], q.p[1], q.p[2]}, q.p[3], q.argb));
                break;
            case 1: recs.push_back(sr::make_plane_record({q.p[0], q.p[1], q.p[2]}, {q.p[3], q.p[4], q.p[5]}, q.argb)); break;
            case 3: {                                              // an existing Plane object: its stored unit normal and originDist, verbatim
                sr::Rec128 r;
                std::memset(&r, 0, sizeof(r));
                r.p[0] = q.p[0]; r.p[1] = q.p[1]; r.p[2] = q.p[2]; r.p[3] = q.p[3];
                r.color = q.argb; r.aux = 1;
                recs.push_back(r);
                break;
            }
            case 2: {
                sr::Rec128 r = sr::make_triangle_record({q.p[0], q.p[1], q.p[2]}, {q.p[3], q.p[4], q.p[5]}, {q.p[6], q.p[7], q.p[8]}, q.argb, 2);
                recs.push_back(r);
                break;
            }
            case 4:                                                // AxisAlignedBox(min, max), AxisAlignedBox.cs:15-28
                if (!(q.p[0] < q.p[3] && q.p[1] < q.p[4] && q.p[2] < q.p[5])) return fail(SR_ERR_INVALID_ARG, "Axis aligned bounding box has bad coordinates (AxisAlignedBox.cs:17-19)");
                recs.push_back(sr::make_box_record({q.p[0], q.p[1], q.p[2]}, {q.p[3], q.p[4], q.p[5]}));
                break;
            default: return fail(SR_ERR_INVALID_ARG, "unknown primitive kind");
        }
    }
    s->extra_recs.swap(recs);
    s->extra_dirty = true;
    return SR_OK;
}

int sr_build(sr_scene* s, uint32_t modes, int32_t max_depth, int32_t max_per_leaf) {
    if (s && !s->parts.empty()) {
        // the host structures (reference tree, SAH BVH) are built once and copied; a device-built BVH is built by every part
        int rc = sr_build(s->parts[0], modes, max_depth, max_per_leaf);
        for (size_t i = 1; i < s->parts.size() && !rc; ++i) {
            sr_scene* q = s->parts[i];
            if (!(modes & SR_BUILD_ON_HOST) && (modes & (1u << SR_MODE_BVH)) && s->parts[0]->bvh_on_device) {
                share_ref_tree(q, s->parts[0]);
                rc = sr_build(q, modes & ~(1u << SR_MODE_REF_TREE), max_depth, max_per_leaf);
                if (!rc && (modes & (1u << SR_MODE_REF_TREE)) && (rc = use_device(q)) == SR_OK) rc = sync_geometry(q, SR_MODE_REF_TREE);
                continue;
            }
            if (modes & (1u << SR_MODE_REF_TREE)) share_ref_tree(q, s->parts[0]);
            if (modes & (1u << SR_MODE_BVH)) share_host_bvh(q, s->parts[0]);
            if ((rc = use_device(q))) break;
            if (modes & (1u << SR_MODE_REF_TREE)) if ((rc = sync_geometry(q, SR_MODE_REF_TREE))) break;
            if (modes & (1u << SR_MODE_BVH)) if ((rc = sync_geometry(q, SR_MODE_BVH))) break;
            rc = sync_geometry(q, SR_MODE_BRUTE);
        }
        return rc;
    }
    if (!s) return fail(SR_ERR_INVALID_ARG, "scene is NULL");
    if (!s->have_model) return fail(SR_ERR_NO_MODEL, "sr_build before sr_set_triangles");
    // leaves are packed as (first record | count << 27) and traversal stack words as (node | bound << bits): 2^28 records / 2^26 nodes
    if ((modes & (1u << SR_MODE_BVH)) && s->ntris >= (1u << 26))
        return fail(SR_ERR_UNSUPPORTED, "the library's BVH holds at most 2^26 - 1 triangles");
    if (modes & (1u << SR_MODE_REF_TREE)) {
        int md = max_depth > 0 ? max_depth : 15, mg = max_per_leaf > 0 ? max_per_leaf : 25;   // SpatialSubdivision.cs:269-270
        if (!sr::build_ref_tree(s->v9, s->bmin, s->bmax, md, mg, s->ref))
            return fail(SR_ERR_OUT_OF_RANGE, "A triangle vertex is outside the bounding box");
        s->ref_dirty = true;
    }
    // the own BVH is built where the triangles are: on the device (LBVH, sr_lbvh.hip), unless the scene has none, is tiny, or the
    // caller asks for the host's binned-SAH builder (SR_BUILD_ON_HOST)
    if ((modes & SR_BUILD_ON_DEVICE) && s->device < 0) return fail(SR_ERR_NO_DEVICE, "SR_BUILD_ON_DEVICE needs a HIP device");
    const bool on_device = s->device >= 0 && !(modes & SR_BUILD_ON_HOST) && s->ntris > 64;
    if ((modes & (1u << SR_MODE_BVH)) && on_device) {
        // ---- LBVH built by the GPU (sr_lbvh.hip) ----
        int rc = use_device(s);
        if (rc) return rc;
        if ((rc = sync_geometry(s, SR_MODE_BRUTE))) return rc;            // d_tris
        const size_t n = s->ntris;
        DBuf d_slab;
        DBuf& d_v9 = s->d_v9;                                             // uploaded by sync_geometry with the records
        SR_HIP(d_slab.reserve(n * sizeof(sr::TriSlab)));
        SR_HIP(sr::make_slabs_device((const double*)d_v9.p, (int)n, s->root, (sr::TriSlab*)d_slab.p, nullptr));   // (the host loop + upload cost 0.4 s at 10 M)
        SR_HIP(s->d_bnodes.reserve(n * sizeof(sr::BvhNode)));
        SR_HIP(s->d_btris.reserve(n * sizeof(sr::Rec128)));
        SR_HIP(s->d_bslab.reserve(n * sizeof(sr::TriSlab)));
        int nn = 0, depth = 0;
        hipError_t e = sr::build_bvh_device((const double*)d_v9.p, (int)n, s->root, (const sr::Rec128*)s->d_tris.p, (const sr::TriSlab*)d_slab.p,
                                            (sr::BvhNode*)s->d_bnodes.p, (sr::Rec128*)s->d_btris.p, (sr::TriSlab*)s->d_bslab.p, &nn, &depth, nullptr,
                                            s->dbg[SR_DBG_BVH_LEAF] > 0 ? (int)std::min<int64_t>(15, s->dbg[SR_DBG_BVH_LEAF]) : 0);
        d_slab.release();
        if (e != hipSuccess) return hip_fail(e, "build_bvh_device");
        if (depth > kMaxTreeDepth) return fail(SR_ERR_UNSUPPORTED, "BVH deeper than 62 levels does not fit the LDS traversal stacks");
        s->bvh = sr::Bvh();
        s->bvh.built = true;
        s->bvh.depth = depth;
        s->bvh_num_nodes = (size_t)nn;
        s->bvh_on_device = true;
        s->bvh_dirty = false;
        s->cam_valid = false;
        {   // the four-wide tree of the packet walks, collapsed where the binary nodes are
            if (s->pre_used_set) SR_HIP(hipEventSynchronize(s->pre_used));      // a frame in flight may still be walking the old tree's copies
            s->b4cam_valid = s->b4light_valid = false;
            s->part_valid = false;
            SR_HIP(s->d_b4.reserve((size_t)nn * sizeof(sr::Bvh4Node)));
            int n4 = 0, d4 = 0;
            e = sr::collapse_bvh4_device((const sr::BvhNode*)s->d_bnodes.p, nn, (sr::Bvh4Node*)s->d_b4.p, &n4, &d4, nullptr);
            if (e != hipSuccess) return hip_fail(e, "collapse_bvh4_device");
            s->b4_num = (size_t)n4;
            s->b4_depth = d4;
            SR_HIP(s->d_b4cam.reserve((size_t)n4 * sizeof(sr::Bvh4Node)));
            SR_HIP(s->d_b4light.reserve((size_t)n4 * sizeof(sr::Bvh4Node)));
        }
    } else if (modes & (1u << SR_MODE_BVH)) {
        sr::build_bvh(s->v9, s->root, s->bvh, s->dbg[SR_DBG_BVH_LEAF] > 0 ? (int)std::min<int64_t>(15, s->dbg[SR_DBG_BVH_LEAF]) : 4,
                      s->dbg[SR_DBG_BUILD_THREADS] > 0 ? (int)std::min<int64_t>(64, s->dbg[SR_DBG_BUILD_THREADS]) : 0);
        if (s->bvh.depth > kMaxTreeDepth) return fail(SR_ERR_UNSUPPORTED, "BVH deeper than 62 levels does not fit the LDS traversal stacks");
        s->bvh_on_device = false;
        s->bvh_dirty = true;
    }
    if (s->device >= 0) {
        int rc = use_device(s);
        if (rc) return rc;
        if (modes & (1u << SR_MODE_REF_TREE)) if ((rc = sync_geometry(s, SR_MODE_REF_TREE))) return rc;
        if ((modes & (1u << SR_MODE_BVH)) && !s->bvh_on_device) if ((rc = sync_geometry(s, SR_MODE_BVH))) return rc;
        if ((rc = sync_geometry(s, SR_MODE_BRUTE))) return rc;
    }
    return SR_OK;
}

int sr_tree_stats(const sr_scene* s, int32_t out[4]) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "bad argument");
    if (!s->ref.built) return fail(SR_ERR_NOT_BUILT, "reference tree not built");
    out[0] = s->ref.tree_depth; out[1] = s->ref.num_nodes; out[2] = s->ref.num_leaf_nodes; out[3] = s->ref.num_nodes - s->ref.num_leaf_nodes;
    return SR_OK;
}

int sr_bvh_stats(const sr_scene* s, int64_t out[4]) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "bad argument");
    if (!s->bvh.built) return fail(SR_ERR_NOT_BUILT, "BVH not built");
    out[0] = s->bvh.depth; out[1] = (int64_t)(s->bvh_on_device ? s->bvh_num_nodes : s->bvh.nodes.size()); out[2] = (int64_t)s->ntris; out[3] = s->bvh_on_device ? 1 : 0;
    return SR_OK;
}

int sr_wide_tree_stats(const sr_scene* s, int64_t out[5]) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "bad argument");
    if (!s->bvh.built || s->bvh_on_device) return fail(SR_ERR_NOT_BUILT, "no host-built BVH");
    std::vector<sr::Bvh4Node> wide;
    out[0] = sr::collapse_bvh4(s->bvh.nodes.data(), s->bvh.nodes.size(), wide);
    out[1] = (int64_t)wide.size();
    out[2] = out[3] = out[4] = 0;
    std::vector<char> linked(wide.size(), 0);
    for (const sr::Bvh4Node& n : wide)
        for (const sr::Bvh4Child& c : n.ch) {
            if (c.n < 0) continue;
            out[2]++;                                                  // child slots in use
            if (c.n > 0) { out[3]++; out[4] += c.n; }                  // leaves, triangles in leaves
            else if (c.c <= 0 || (size_t)c.c >= wide.size() || linked[(size_t)c.c]++) return fail(SR_ERR_UNSUPPORTED, "wide tree: bad inner link");
        }
    return SR_OK;
}

int sr_bvh_digest(const sr_scene* s, uint64_t out[2]) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "bad argument");
    if (!s->bvh.built || s->bvh_on_device) return fail(SR_ERR_NOT_BUILT, "no host-built BVH");
    auto fnv = [](const void* p, size_t bytes) {
        uint64_t h = 1469598103934665603ull;
        const unsigned char* c = (const unsigned char*)p;
        for (size_t i = 0; i < bytes; ++i) { h ^= c[i]; h *= 1099511628211ull; }
        return h;
    };
    out[0] = fnv(s->bvh.nodes.data(), s->bvh.nodes.size() * sizeof(sr::BvhNode));
    out[1] = fnv(s->bvh.order.data(), s->bvh.order.size() * sizeof(int32_t));
    return SR_OK;
}

int64_t sr_frame_pixel_count(const sr_frame* f) {
    if (!f || f->width <= 0 || f->height <= 0) return 0;
    int a, b;
    clamp_rows(f, a, b);
    if (f->strip_count <= 0) return (int64_t)f->width * f->height;
    int64_t rows = 0;
    for (int r = a; r <= b; ++r) if (row_owned(f, r)) rows++;
    return rows * f->width;
}

int sr_reset_shadow_cache(sr_scene* s) {
    if (s && !s->parts.empty()) { for (sr_scene* q : s->parts) q->shadow_cache_empty = true; return SR_OK; }
    if (!s) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_reset_shadow_cache");
    s->shadow_cache_empty = true;
    return SR_OK;
}

static int multi_render(sr_scene* m, const sr_frame* f, int32_t* host_pixels, void* d_pixels, hipStream_t user_stream, uint64_t* stats4, uint64_t* d_stats) {
    int rc = validate_frame(f);
    if (rc) return rc;
    if (d_stats) return fail(SR_ERR_UNSUPPORTED, "device-side statistics are per device: use sr_render / sr_last_ray_stats with a multi-device scene");
    const int n = (int)m->parts.size();
    if (!multi_splittable(f)) {                                     // static shadow cache / caller-made strips: the first part renders it
        rc = host_pixels ? sr_render(m->parts[0], f, host_pixels, stats4) : sr_render_device(m->parts[0], f, d_pixels, user_stream, nullptr);
        std::memcpy(m->last_stats, m->parts[0]->last_stats, sizeof(m->last_stats));     // sr_last_ray_stats(multi scene) reports this frame
        return rc;
    }
    if ((rc = check_mode(m->parts[0], f->trace_mode))) return rc;
    int a, b;
    clamp_rows(f, a, b);
    if (b < a) { if (stats4) std::memset(stats4, 0, 4 * sizeof(uint64_t)); return SR_OK; }
    // ---- every part enqueues its strips on its own device and stream (nothing below waits for a GPU until all have been enqueued) ----
    std::vector<sr_frame> fg(n, *f);
    std::vector<int64_t> counts(n, 0);
    for (int g = 0; g < n; ++g) {
        sr_scene* q = m->parts[g];
        fg[g].strip_rows = kMultiStripRows; fg[g].strip_count = n; fg[g].strip_index = g;
        counts[g] = sr_frame_pixel_count(&fg[g]);
        if (counts[g] == 0) continue;
        if ((rc = check_mode(q, f->trace_mode))) return rc;
        if ((rc = use_device(q))) return rc;
        if ((rc = ensure_io_streams(q))) return rc;
        // the previous frame's gather may still be reading this part's strips (sr_render_device returns before the copies have run)
        if (m->multi_done) SR_HIP(hipStreamWaitEvent(q->io_stream, m->multi_done, 0));
        SR_HIP(q->d_pixels.reserve((size_t)counts[g] * 4));
        unsigned long long* ds = nullptr;
        if (stats4) {
            SR_HIP(q->d_stats.reserve(SR_STATS_COUNT * sizeof(uint64_t)));
            SR_HIP(hipMemsetAsync(q->d_stats.p, 0, SR_STATS_COUNT * sizeof(uint64_t), q->io_stream));
            ds = (unsigned long long*)q->d_stats.p;
        }
        if ((rc = render_common(q, &fg[g], (uint32_t*)q->d_pixels.p, q->io_stream, ds))) return rc;
        if (!q->multi_done) SR_HIP(hipEventCreateWithFlags(&q->multi_done, hipEventDisableTiming));
        SR_HIP(hipEventRecord(q->multi_done, q->io_stream));
    }
    // ---- the strips go straight to the caller's surface ----
    if (host_pixels) {
        // every device copies its own strips over its own link, all of them at once, into the (pinned for this call) surface
        HostPin pin(host_pixels + (size_t)a * f->width, (size_t)(b - a + 1) * f->width * 4);
        for (int g = 0; g < n; ++g) {
            sr_scene* q = m->parts[g];
            const std::vector<StripRun> runs = strip_runs(a, b, n, g);
            if (runs.empty() || counts[g] == 0) continue;
            if ((rc = use_device(q))) return rc;
            SR_HIP(hipStreamWaitEvent(q->copy_stream, q->multi_done, 0));
            SR_HIP(copy_runs(runs, n, f->width, (const uint32_t*)q->d_pixels.p, (uint32_t*)host_pixels, hipMemcpyDeviceToHost, q->copy_stream));
        }
        for (int k = 0; k < SR_STATS_COUNT; ++k) m->last_stats[k] = 0;
        for (int g = 0; g < n; ++g) {                               // one wait per device, after everything has been queued
            sr_scene* q = m->parts[g];
            if (counts[g] == 0) continue;
            if ((rc = use_device(q))) return rc;
            SR_HIP(hipStreamSynchronize(q->copy_stream));
            SR_HIP(hipStreamSynchronize(q->io_stream));
            if (stats4) {
                SR_HIP(hipMemcpy(q->last_stats, q->d_stats.p, SR_STATS_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost));
                for (int k = 0; k < SR_STATS_COUNT; ++k) m->last_stats[k] += q->last_stats[k];
            }
        }
        if (stats4) std::memcpy(stats4, m->last_stats, 4 * sizeof(uint64_t));
        return SR_OK;
    }
    // device surface on the first part's device.  SR_GATHER_RCCL: the one exchange step of SURVEY 8e -- grouped ncclSend (every other
    // part, on the stream its strips were rendered on) / ncclRecv (the first part, on the caller's stream) of the compact strip
    // buffers over xGMI, then the row de-interleave as strided device copies on the first device
    if (m->gather_kind == SR_GATHER_RCCL && n > 1) {
        const sr::RcclApi* api = sr::rccl_api(nullptr);
        if (!api || (int)m->comms.size() != n) return fail(SR_ERR_NOT_BUILT, "SR_GATHER_RCCL: sr_set_gather has not created the communicators");
        std::vector<size_t> off(n, 0);
        size_t total = 0;
        for (int g = 1; g < n; ++g) { off[g] = total; total += (size_t)counts[g]; }
        if ((rc = use_device(m->parts[0]))) return rc;
        SR_HIP(m->d_gather.reserve(std::max<size_t>(total, 1) * 4));
        SR_RCCL(api, api->GroupStart());
        for (int g = 1; g < n; ++g) {
            if (counts[g] == 0) continue;
            if ((rc = use_device(m->parts[g]))) { (void)api->GroupEnd(); return rc; }
            SR_RCCL(api, api->Send(m->parts[g]->d_pixels.p, (size_t)counts[g], sr::kRcclInt32, 0, m->comms[g], m->parts[g]->io_stream));
        }
        if ((rc = use_device(m->parts[0]))) { (void)api->GroupEnd(); return rc; }
        for (int g = 1; g < n; ++g) {
            if (counts[g] == 0) continue;
            SR_RCCL(api, api->Recv((uint32_t*)m->d_gather.p + off[g], (size_t)counts[g], sr::kRcclInt32, g, m->comms[0], user_stream));
        }
        SR_RCCL(api, api->GroupEnd());
        if (counts[0]) SR_HIP(hipStreamWaitEvent(user_stream, m->parts[0]->multi_done, 0));
        for (int g = 0; g < n; ++g) {
            const std::vector<StripRun> runs = strip_runs(a, b, n, g);
            if (runs.empty() || counts[g] == 0) continue;
            const uint32_t* src = g == 0 ? (const uint32_t*)m->parts[0]->d_pixels.p : (const uint32_t*)m->d_gather.p + off[g];
            SR_HIP(copy_runs(runs, n, f->width, src, (uint32_t*)d_pixels, hipMemcpyDeviceToDevice, user_stream));
        }
        if (!m->multi_done) SR_HIP(hipEventCreateWithFlags(&m->multi_done, hipEventDisableTiming));
        SR_HIP(hipEventRecord(m->multi_done, user_stream));
        return SR_OK;
    }
    // SR_GATHER_COPY (default): peer-to-peer over xGMI where the first device can read the part's memory, through pinned host
    // staging where it cannot (sr_create_multi asked; nothing is assumed)
    for (int g = 0; g < n; ++g) {
        sr_scene* q = m->parts[g];
        const std::vector<StripRun> runs = strip_runs(a, b, n, g);
        if (runs.empty() || counts[g] == 0) continue;
        if (q->can_peer && q->dbg[SR_DBG_NO_PEER] <= 0) {
            if ((rc = use_device(m->parts[0]))) return rc;
            SR_HIP(hipStreamWaitEvent(user_stream, q->multi_done, 0));
            SR_HIP(copy_runs(runs, n, f->width, (const uint32_t*)q->d_pixels.p, (uint32_t*)d_pixels, hipMemcpyDefault, user_stream));
        } else {
            const size_t bytes = (size_t)counts[g] * 4;
            if ((rc = use_device(q))) return rc;
            if (bytes > q->stage_cap) {
                if (q->stage_host) SR_HIP(hipHostFree(q->stage_host));
                q->stage_host = nullptr; q->stage_cap = 0;
                SR_HIP(hipHostMalloc(&q->stage_host, bytes, hipHostMallocPortable));
                q->stage_cap = bytes;
            }
            if (!q->staged) SR_HIP(hipEventCreateWithFlags(&q->staged, hipEventDisableTiming));
            SR_HIP(hipStreamWaitEvent(q->copy_stream, q->multi_done, 0));
            SR_HIP(hipMemcpyAsync(q->stage_host, q->d_pixels.p, bytes, hipMemcpyDeviceToHost, q->copy_stream));
            SR_HIP(hipEventRecord(q->staged, q->copy_stream));
            if ((rc = use_device(m->parts[0]))) return rc;
            SR_HIP(hipStreamWaitEvent(user_stream, q->staged, 0));
            SR_HIP(copy_runs(runs, n, f->width, (const uint32_t*)q->stage_host, (uint32_t*)d_pixels, hipMemcpyHostToDevice, user_stream));
        }
    }
    if ((rc = use_device(m->parts[0]))) return rc;
    if (!m->multi_done) SR_HIP(hipEventCreateWithFlags(&m->multi_done, hipEventDisableTiming));
    SR_HIP(hipEventRecord(m->multi_done, user_stream));
    return SR_OK;
}

int sr_render_device(sr_scene* s, const sr_frame* f, void* d_pixels, void* hip_stream, uint64_t* d_stats) {
    if (s && !s->parts.empty()) {
        if (!d_pixels) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_render_device");
        return multi_render(s, f, nullptr, d_pixels, (hipStream_t)hip_stream, nullptr, d_stats);
    }
    if (!s || !d_pixels) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_render_device");
    int rc = validate_frame(f);
    if (rc) return rc;
    if ((rc = check_mode(s, f->trace_mode))) return rc;
    if ((rc = use_device(s))) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    if (d_stats) SR_HIP(hipMemsetAsync(d_stats, 0, SR_STATS_COUNT * sizeof(uint64_t), stream));
    return render_common(s, f, (uint32_t*)d_pixels, stream, (unsigned long long*)d_stats);
}

int sr_render(sr_scene* s, const sr_frame* f, int32_t* pixels, uint64_t stats[4]) {
    if (s && !s->parts.empty()) {
        if (!pixels) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_render");
        return multi_render(s, f, pixels, nullptr, nullptr, stats, nullptr);
    }
    if (!s || !pixels) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_render");
    int rc = validate_frame(f);
    if (rc) return rc;
    if ((rc = check_mode(s, f->trace_mode))) return rc;
    if ((rc = use_device(s))) return rc;
    int64_t count = sr_frame_pixel_count(f);
    {
        int a, b;
        clamp_rows(f, a, b);
        if (b < a || count == 0) { if (stats) std::memset(stats, 0, 4 * sizeof(uint64_t)); return SR_OK; }   // rayTraceStartRow > EndRow: nothing is drawn
    }
    SR_HIP(s->d_pixels.reserve((size_t)count * 4));
    if ((rc = ensure_io_streams(s))) return rc;
    unsigned long long* d_stats = nullptr;
    if (stats) {
        SR_HIP(s->d_stats.reserve(SR_STATS_COUNT * sizeof(uint64_t)));
        SR_HIP(hipMemsetAsync(s->d_stats.p, 0, SR_STATS_COUNT * sizeof(uint64_t), s->io_stream));
        d_stats = (unsigned long long*)s->d_stats.p;
    }
    // rows the reference would have drawn: [a, b] of the surface, or this call's strips as one compact block
    int a, b;
    clamp_rows(f, a, b);
    const bool strips = f->strip_count > 0;
    const size_t first_px = strips ? 0 : (size_t)a * f->width;
    const size_t total_px = strips ? (size_t)count : (size_t)(b - a + 1) * f->width;
    HostPin pin(pixels + first_px, total_px * 4);                  // (hidden behind the frame: nothing has been enqueued yet that we wait for)
    // ---- enqueue the frame; every row band leaves an event behind ----
    s->band_recs.clear();
    s->band_ev_used = 0;
    s->collect_bands = true;
    rc = render_common(s, f, (uint32_t*)s->d_pixels.p, s->io_stream, d_stats);
    const bool banded = s->collect_bands && !s->band_recs.empty();
    s->collect_bands = false;
    if (rc) { (void)hipStreamSynchronize(s->io_stream); return rc; }
    // ---- its way back: band by band on the copy stream, earliest bands first (the two half-frame pipelines run side by side) ----
    size_t covered = 0;
    if (banded) {
        std::stable_sort(s->band_recs.begin(), s->band_recs.end(), [](const sr_scene::BandRec& x, const sr_scene::BandRec& y) { return x.band < y.band; });
        for (const sr_scene::BandRec& r : s->band_recs) covered += (size_t)r.row_count * f->width;
    }
    if (banded && covered == total_px) {
        for (const sr_scene::BandRec& r : s->band_recs) {
            const size_t off = first_px + (size_t)r.row_begin * f->width, n = (size_t)r.row_count * f->width;
            SR_HIP(hipStreamWaitEvent(s->copy_stream, r.ev, 0));
            SR_HIP(hipMemcpyAsync(pixels + off, (const int32_t*)s->d_pixels.p + off, n * 4, hipMemcpyDeviceToHost, s->copy_stream));
        }
        SR_HIP(hipStreamSynchronize(s->copy_stream));
        SR_HIP(hipStreamSynchronize(s->io_stream));                 // (the join of the halves, the statistics)
    } else {
        // one-kernel frames (no bands): the whole range after the frame
        SR_HIP(hipStreamSynchronize(s->io_stream));
        SR_HIP(hipMemcpyAsync(pixels + first_px, (const int32_t*)s->d_pixels.p + first_px, total_px * 4, hipMemcpyDeviceToHost, s->copy_stream));
        SR_HIP(hipStreamSynchronize(s->copy_stream));
    }
    if (stats) {
        SR_HIP(hipMemcpy(s->last_stats, s->d_stats.p, SR_STATS_COUNT * sizeof(uint64_t), hipMemcpyDeviceToHost));
        std::memcpy(stats, s->last_stats, 4 * sizeof(uint64_t));
    }
    return SR_OK;
}

static int trace_prepare(sr_scene*& s, int32_t target, int64_t n, bool have_rays, int& mode, bool& with_extra) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || n < 0 || (n > 0 && !have_rays)) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_trace_rays");
    with_extra = (target & SR_TARGET_ROOT) != 0;
    mode = target & 0xff;
    if (!s->have_model) return fail(SR_ERR_NO_MODEL, "no model");
    int rc;
    if (s->ntris == 0 && mode == SR_MODE_BRUTE) { /* empty model is fine for brute force */ }
    else if ((rc = check_mode(s, mode))) return rc;
    if ((rc = use_device(s))) return rc;
    return sync_geometry(s, (uint32_t)mode);
}

int sr_trace_rays_device(sr_scene* s, int32_t target, int64_t n, const double* d_starts, const double* d_dirs, uint8_t* d_hit, double* d_ray_frac,
                         double* d_pos, double* d_normal, uint32_t* d_color, int32_t* d_tri_index, int32_t* d_counters, void* hip_stream) {
    int mode; bool with_extra;
    int rc = trace_prepare(s, target, n, d_starts && d_dirs, mode, with_extra);
    if (rc || n == 0) return rc;
    sr::TraceLaunch L{};
    L.sc = dev_scene(s);
    L.mode = mode;
    L.with_extra = with_extra && !s->extra_recs.empty();
    L.n = n;
    L.starts = d_starts; L.dirs = d_dirs;
    L.hit = d_hit; L.ray_frac = d_ray_frac; L.pos = d_pos; L.normal = d_normal; L.color = d_color; L.tri = d_tri_index; L.counters = d_counters;
    L.stream = (hipStream_t)hip_stream;
    hipEvent_t e0, e1;
    if ((rc = next_events(s, sr::K_TRACE, e0, e1))) return rc;
    if (e0) SR_HIP(hipEventRecord(e0, L.stream));
    SR_HIP(sr::launch_trace(L));
    if (e1) SR_HIP(hipEventRecord(e1, L.stream));
    return SR_OK;
}

int sr_trace_rays(sr_scene* s, int32_t target, int64_t n, const double* starts, const double* dirs, uint8_t* hit, double* ray_frac,
                  double* pos, double* normal, uint32_t* color, int32_t* tri_index, int32_t* counters) {
    int mode; bool with_extra;
    int rc = trace_prepare(s, target, n, starts && dirs, mode, with_extra);
    if (rc || n == 0) return rc;
    size_t sizes[9] = {(size_t)n * 24, (size_t)n * 24, (size_t)n, (size_t)n * 8, (size_t)n * 24, (size_t)n * 24, (size_t)n * 4, (size_t)n * 4, (size_t)n * 12};
    for (int i = 0; i < 9; ++i) SR_HIP(s->d_io[i].reserve(sizes[i]));
    SR_HIP(hipMemcpy(s->d_io[0].p, starts, sizes[0], hipMemcpyHostToDevice));
    SR_HIP(hipMemcpy(s->d_io[1].p, dirs, sizes[1], hipMemcpyHostToDevice));
    if ((rc = sr_trace_rays_device(s, target, n, (const double*)s->d_io[0].p, (const double*)s->d_io[1].p, (uint8_t*)s->d_io[2].p, (double*)s->d_io[3].p,
                                   (double*)s->d_io[4].p, (double*)s->d_io[5].p, (uint32_t*)s->d_io[6].p, (int32_t*)s->d_io[7].p, (int32_t*)s->d_io[8].p, nullptr)))
        return rc;
    SR_HIP(hipStreamSynchronize(nullptr));
    void* outs[7] = {hit, ray_frac, pos, normal, color, tri_index, counters};
    for (int i = 0; i < 7; ++i)
        if (outs[i]) SR_HIP(hipMemcpy(outs[i], s->d_io[2 + i].p, sizes[2 + i], hipMemcpyDeviceToHost));
    return SR_OK;
}

// ------------------------------------------------------------------------------------------------------------------
// The RCCL strip gather, native: no PyTorch in the data path.
// ------------------------------------------------------------------------------------------------------------------
int sr_set_gather(sr_scene* m, int32_t kind) {
    if (!m || (kind != SR_GATHER_COPY && kind != SR_GATHER_RCCL)) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_set_gather");
    if (m->parts.empty()) return fail(SR_ERR_INVALID_ARG, "sr_set_gather: not a multi-device scene (sr_create_multi)");
    if (kind == SR_GATHER_RCCL && m->comms.empty() && m->parts.size() > 1) {
        std::string why;
        const sr::RcclApi* api = sr::rccl_api(&why);
        if (!api) return fail(SR_ERR_UNSUPPORTED, why);
        std::vector<int> devs;
        for (sr_scene* q : m->parts) devs.push_back(q->device);
        for (size_t i = 0; i < devs.size(); ++i)
            for (size_t j = i + 1; j < devs.size(); ++j)
                if (devs[i] == devs[j]) return fail(SR_ERR_UNSUPPORTED, "SR_GATHER_RCCL needs distinct devices (RCCL refuses two ranks on one GPU)");
        std::vector<sr::RcclComm> comms(devs.size(), nullptr);
        SR_RCCL(api, api->CommInitAll(comms.data(), (int)devs.size(), devs.data()));
        for (sr_scene* q : m->parts) { int rc = use_device(q); if (rc) return rc; rc = ensure_io_streams(q); if (rc) return rc; }
        m->comms.swap(comms);
    }
    m->gather_kind = kind;
    return SR_OK;
}

int sr_rccl_unique_id(uint8_t out[SR_RCCL_ID_BYTES]) {
    if (!out) return fail(SR_ERR_INVALID_ARG, "out is NULL");
    std::string why;
    const sr::RcclApi* api = sr::rccl_api(&why);
    if (!api) return fail(SR_ERR_UNSUPPORTED, why);
    static_assert(SR_RCCL_ID_BYTES == sr::kRcclIdBytes, "ncclUniqueId is 128 bytes");
    sr::RcclId id;
    SR_RCCL(api, api->GetUniqueId(&id));
    std::memcpy(out, id.internal, SR_RCCL_ID_BYTES);
    return SR_OK;
}

int sr_rccl_init(sr_scene* s, const uint8_t id_bytes[SR_RCCL_ID_BYTES], int32_t world, int32_t rank) {
    if (!s || !id_bytes || world < 1 || rank < 0 || rank >= world) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_rccl_init");
    if (!s->parts.empty()) return fail(SR_ERR_INVALID_ARG, "sr_rccl_init: one scene = one rank = one device (use sr_set_gather for a multi-device scene)");
    std::string why;
    const sr::RcclApi* api = sr::rccl_api(&why);
    if (!api) return fail(SR_ERR_UNSUPPORTED, why);
    int rc = use_device(s);
    if (rc) return rc;
    if (s->comm) { SR_RCCL(api, api->CommDestroy(s->comm)); s->comm = nullptr; }
    sr::RcclId id;
    std::memcpy(id.internal, id_bytes, SR_RCCL_ID_BYTES);
    SR_RCCL(api, api->CommInitRank(&s->comm, world, id, rank));
    s->rccl_world = world; s->rccl_rank = rank;
    return SR_OK;
}

int sr_rccl_gather(sr_scene* s, const sr_frame* f, const void* d_strips, void* d_full, void* hip_stream) {
    if (!s || !s->parts.empty()) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_rccl_gather");
    if (!s->comm) return fail(SR_ERR_NOT_BUILT, "sr_rccl_gather before sr_rccl_init");
    int rc = validate_frame(f);
    if (rc) return rc;
    const sr::RcclApi* api = sr::rccl_api(nullptr);
    if (!api) return fail(SR_ERR_UNSUPPORTED, "librccl is not loaded");
    if ((rc = use_device(s))) return rc;
    const int n = s->rccl_world, me = s->rccl_rank;
    if (me == 0 && !d_full) return fail(SR_ERR_INVALID_ARG, "rank 0 needs the full surface");
    int a, b;
    clamp_rows(f, a, b);
    if (b < a) return SR_OK;
    std::vector<size_t> counts(n, 0), off(n, 0);
    size_t total = 0;
    for (int g = 0; g < n; ++g) {
        sr_frame fg = *f;
        fg.strip_rows = kMultiStripRows; fg.strip_count = n; fg.strip_index = g;
        counts[g] = (size_t)sr_frame_pixel_count(&fg);
        if (g > 0) { off[g] = total; total += counts[g]; }
    }
    if (counts[me] && !d_strips) return fail(SR_ERR_INVALID_ARG, "this rank owns rows but passed no strip buffer");
    hipStream_t stream = (hipStream_t)hip_stream;
    if (me != 0) {
        if (counts[me] == 0) return SR_OK;
        SR_RCCL(api, api->GroupStart());
        SR_RCCL(api, api->Send(d_strips, counts[me], sr::kRcclInt32, 0, s->comm, stream));
        SR_RCCL(api, api->GroupEnd());
        return SR_OK;
    }
    SR_HIP(s->d_gather.reserve(std::max<size_t>(total, 1) * 4));
    if (n > 1) {
        SR_RCCL(api, api->GroupStart());
        for (int g = 1; g < n; ++g)
            if (counts[g]) SR_RCCL(api, api->Recv((uint32_t*)s->d_gather.p + off[g], counts[g], sr::kRcclInt32, g, s->comm, stream));
        SR_RCCL(api, api->GroupEnd());
    }
    for (int g = 0; g < n; ++g) {                                     // the row de-interleave: strided device copies behind the receives
        const std::vector<StripRun> runs = strip_runs(a, b, n, g);
        if (runs.empty() || counts[g] == 0) continue;
        const uint32_t* src = g == 0 ? (const uint32_t*)d_strips : (const uint32_t*)s->d_gather.p + off[g];
        SR_HIP(copy_runs(runs, n, f->width, src, (uint32_t*)d_full, hipMemcpyDeviceToDevice, stream));
    }
    return SR_OK;
}

int sr_rccl_render(sr_scene* s, const sr_frame* f, void* d_full, void* hip_stream) {
    if (!s || !s->parts.empty()) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_rccl_render");
    if (!s->comm) return fail(SR_ERR_NOT_BUILT, "sr_rccl_render before sr_rccl_init");
    int rc = validate_frame(f);
    if (rc) return rc;
    if (f->strip_count > 0) return fail(SR_ERR_INVALID_ARG, "sr_rccl_render splits the frame itself: strip_count must be 0");
    if ((f->flags & SR_F_STATIC_SHADOWS) && (f->flags & SR_F_SHADOWS)) return fail(SR_ERR_UNSUPPORTED, "static shadows need the whole frame on one device");
    if ((rc = use_device(s))) return rc;
    sr_frame fg = *f;
    fg.strip_rows = kMultiStripRows; fg.strip_count = s->rccl_world; fg.strip_index = s->rccl_rank;
    const int64_t mine = sr_frame_pixel_count(&fg);
    if (mine > 0) {
        SR_HIP(s->d_pixels.reserve((size_t)mine * 4));
        if ((rc = sr_render_device(s, &fg, s->d_pixels.p, hip_stream, nullptr))) return rc;
    }
    return sr_rccl_gather(s, f, mine > 0 ? s->d_pixels.p : nullptr, d_full, hip_stream);
}

int sr_shade_points(sr_scene* s, const sr_frame* f, int64_t n, const double* pos, const double* normal, const uint32_t* color, uint32_t* out) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || n < 0 || (n > 0 && (!pos || !normal || !color || !out))) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_shade_points");
    int rc = validate_frame(f);
    if (rc) return rc;
    if ((rc = use_device(s))) return rc;
    if (n == 0) return SR_OK;
    sr::FrameConst fc;
    if ((rc = prepare_frame(s, f, fc))) return rc;
    size_t sizes[4] = {(size_t)n * 24, (size_t)n * 24, (size_t)n * 4, (size_t)n * 4};
    for (int i = 0; i < 4; ++i) SR_HIP(s->d_io[i].reserve(sizes[i]));
    SR_HIP(hipMemcpy(s->d_io[0].p, pos, sizes[0], hipMemcpyHostToDevice));
    SR_HIP(hipMemcpy(s->d_io[1].p, normal, sizes[1], hipMemcpyHostToDevice));
    SR_HIP(hipMemcpy(s->d_io[2].p, color, sizes[2], hipMemcpyHostToDevice));
    SR_HIP(sr::launch_shade_points(fc, n, (const double*)s->d_io[0].p, (const double*)s->d_io[1].p, (const uint32_t*)s->d_io[2].p, (uint32_t*)s->d_io[3].p, nullptr));
    SR_HIP(hipStreamSynchronize(nullptr));
    SR_HIP(hipMemcpy(out, s->d_io[3].p, sizes[3], hipMemcpyDeviceToHost));
    return SR_OK;
}

void sr_instance_matrices(const double position[3], double yaw, double pitch, double roll, double transform[12], double inv_transform[12]) {
    sr::instance_matrices(position, yaw, pitch, roll, transform, inv_transform);
}
double sr_default_fov_depth(void) { return sr::default_fov_depth(); }
void sr_area_light_offsets(int32_t seed, int32_t count, double* out3) { sr::area_light_offsets(seed, count, out3); }

int sr_load_3ds(sr_scene* s, const uint8_t* data, size_t len) {
    if (!s || !data) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_load_3ds");
    sr::LoadedModel m;
    std::string err = sr::load_3ds(data, len, m);
    if (!err.empty()) return fail(SR_ERR_FORMAT, err);
    return sr_set_triangles(s, m.v9.data(), m.argb.data(), (int64_t)m.argb.size(), m.bmin, m.bmax);
}
int64_t sr_num_triangles(const sr_scene* s) { if (s && !s->parts.empty()) s = s->parts[0]; return s ? (int64_t)s->ntris : 0; }
int sr_get_triangles(const sr_scene* s, double* v9, uint32_t* argb, double box_min[3], double box_max[3]) {
    if (s && !s->parts.empty()) s = s->parts[0];
    if (!s || !s->have_model) return fail(SR_ERR_NO_MODEL, "no model");
    if (v9) std::memcpy(v9, s->v9.data(), s->v9.size() * sizeof(double));
    if (argb) std::memcpy(argb, s->argb.data(), s->argb.size() * sizeof(uint32_t));
    for (int a = 0; a < 3; ++a) { if (box_min) box_min[a] = s->bmin[a]; if (box_max) box_max[a] = s->bmax[a]; }
    return SR_OK;
}

int sr_last_ray_stats(const sr_scene* s, uint64_t out[SR_STATS_COUNT]) {
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "bad argument");
    std::memcpy(out, s->last_stats, sizeof(s->last_stats));
    return SR_OK;
}

void sr_make_random_triangles(int32_t seed, int64_t n, double space, double extent, double origin, int32_t opaque,
                              double* v9, uint32_t* argb) {
    sr::NetRandom rnd(seed);
    for (int64_t i = 0; i < n; ++i) {
        double* p = v9 + 9 * i;
        for (int a = 0; a < 3; ++a) p[a] = rnd.next_double() * space + origin;          // v1 = MakeRandomVector(space)
        for (int a = 0; a < 3; ++a) p[3 + a] = p[a] + rnd.next_double() * extent;       // v2 = v1 + MakeRandomVector(extent)
        for (int a = 0; a < 3; ++a) p[6 + a] = p[a] + rnd.next_double() * extent;       // v3 = v1 + MakeRandomVector(extent)
        uint32_t c = (uint32_t)rnd.next();                                              // (uint)random.Next()
        argb[i] = opaque ? (0xFF000000u | (c & 0xFFFFFFu)) : c;
    }
}

void sr_net_random_doubles(int32_t seed, int64_t skip, int64_t n, double* out) {
    sr::NetRandom rnd(seed);
    for (int64_t i = 0; i < skip; ++i) (void)rnd.next();           // Next() and NextDouble() both consume one sample
    for (int64_t i = 0; i < n; ++i) out[i] = rnd.next_double();
}

// ---- surface passes, Renderer.cs:765-767 ----
int sr_post_process_device(sr_scene* s, void* d_pixels, int64_t count, int32_t style, uint32_t background_color, void* hip_stream) {
    if (s && !s->parts.empty()) s = s->parts[0];               // surface passes / timings: the first device
    if (!s || count < 0 || (count > 0 && !d_pixels)) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_post_process");
    if (style < SR_STYLE_STANDARD || style > SR_STYLE_DEPTH_BANDED)
        return fail(SR_ERR_UNSUPPORTED, "render style " + std::to_string(style) + " is not a per-pixel colour function (Style.Normals needs the rasteriser's depth buffer)");
    int rc = use_device(s);
    if (rc) return rc;
    if (style == SR_STYLE_STANDARD || count == 0) return SR_OK;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipEvent_t e0, e1;
    if ((rc = next_events(s, sr::K_POST, e0, e1))) return rc;
    if (e0) SR_HIP(hipEventRecord(e0, stream));
    SR_HIP(sr::launch_post_process((uint32_t*)d_pixels, count, style, background_color, s->num_cus, stream));
    if (e1) SR_HIP(hipEventRecord(e1, stream));
    return SR_OK;
}

int sr_post_process(sr_scene* s, int32_t* pixels, int64_t count, int32_t style, uint32_t background_color) {
    if (s && !s->parts.empty()) s = s->parts[0];               // surface passes / timings: the first device
    if (!s || count < 0 || (count > 0 && !pixels)) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_post_process");
    if (style < SR_STYLE_STANDARD || style > SR_STYLE_DEPTH_BANDED)
        return fail(SR_ERR_UNSUPPORTED, "render style " + std::to_string(style) + " is not a per-pixel colour function (Style.Normals needs the rasteriser's depth buffer)");
    int rc = use_device(s);
    if (rc) return rc;
    if (style == SR_STYLE_STANDARD || count == 0) return SR_OK;
    SR_HIP(s->d_pixels.reserve((size_t)count * 4));
    SR_HIP(hipMemcpy(s->d_pixels.p, pixels, (size_t)count * 4, hipMemcpyHostToDevice));
    if ((rc = sr_post_process_device(s, s->d_pixels.p, count, style, background_color, nullptr))) return rc;
    SR_HIP(hipMemcpy(pixels, s->d_pixels.p, (size_t)count * 4, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_anti_alias_device(sr_scene* s, const void* d_src, int32_t dst_width, int32_t dst_height, int32_t resolution, void* d_dst, void* hip_stream) {
    if (s && !s->parts.empty()) s = s->parts[0];               // surface passes / timings: the first device
    if (!s || !d_src || !d_dst || dst_width <= 0 || dst_height <= 0) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_anti_alias");
    if (resolution < 1 || resolution > 64 || (int64_t)dst_width * resolution > INT_MAX || (int64_t)dst_height * resolution > INT_MAX)
        return fail(SR_ERR_INVALID_ARG, "AntiAliasResolution must be in 1..64");           // Renderer.cs:374 (> 0)
    int rc = use_device(s);
    if (rc) return rc;
    hipStream_t stream = (hipStream_t)hip_stream;
    hipEvent_t e0, e1;
    if ((rc = next_events(s, sr::K_ANTI_ALIAS, e0, e1))) return rc;
    if (e0) SR_HIP(hipEventRecord(e0, stream));
    SR_HIP(sr::launch_anti_alias((const uint32_t*)d_src, (uint32_t*)d_dst, dst_width, dst_height, resolution, stream));
    if (e1) SR_HIP(hipEventRecord(e1, stream));
    return SR_OK;
}

int sr_anti_alias(sr_scene* s, const int32_t* src, int32_t dst_width, int32_t dst_height, int32_t resolution, int32_t* dst) {
    if (s && !s->parts.empty()) s = s->parts[0];               // surface passes / timings: the first device
    if (!s || !src || !dst || dst_width <= 0 || dst_height <= 0) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_anti_alias");
    if (resolution < 1 || resolution > 64 || (int64_t)dst_width * resolution > INT_MAX || (int64_t)dst_height * resolution > INT_MAX)
        return fail(SR_ERR_INVALID_ARG, "AntiAliasResolution must be in 1..64");
    int rc = use_device(s);
    if (rc) return rc;
    const size_t dst_bytes = (size_t)dst_width * dst_height * 4;
    const size_t src_bytes = dst_bytes * resolution * resolution;
    SR_HIP(s->d_pixels.reserve(src_bytes));
    SR_HIP(s->d_aa.reserve(dst_bytes));
    SR_HIP(hipMemcpy(s->d_pixels.p, src, src_bytes, hipMemcpyHostToDevice));
    if ((rc = sr_anti_alias_device(s, s->d_pixels.p, dst_width, dst_height, resolution, s->d_aa.p, nullptr))) return rc;
    SR_HIP(hipMemcpy(dst, s->d_aa.p, dst_bytes, hipMemcpyDeviceToHost));
    return SR_OK;
}

int sr_debug_counters(sr_scene* s, uint32_t out[8]) {
    if (s && !s->parts.empty() && out) {
        for (int i = 0; i < 8; ++i) out[i] = 0;
        for (sr_scene* q : s->parts) { uint32_t c[8]; int rc = sr_debug_counters(q, c); if (rc) return rc; for (int i = 0; i < 8; ++i) out[i] += c[i]; }
        return SR_OK;
    }
    /* diagnostics: the pipeline's device counters after the last band of the last frame:
       hit_count, k_shadow work head, fallback_count, fallback work head */
    if (!s || !out) return fail(SR_ERR_INVALID_ARG, "bad argument");
    int rc = use_device(s);
    if (rc) return rc;
    for (int i = 0; i < 8; ++i) out[i] = 0;
    SR_HIP(hipDeviceSynchronize());
    for (auto& sc : s->scratch) {                                   // every part-frame pipeline of the last frame (last band of each)
        if (!sc.counters.p || !sc.used_last_frame) continue;
        uint32_t c[8];
        SR_HIP(hipMemcpy(c, sc.counters.p, 32, hipMemcpyDeviceToHost));
        for (int i = 0; i < 8; ++i) out[i] += c[i];
    }
    return SR_OK;
}

int sr_debug_set(sr_scene* s, int32_t key, int64_t value) {
    if (s && !s->parts.empty()) { for (sr_scene* q : s->parts) { int rc = sr_debug_set(q, key, value); if (rc) return rc; } return SR_OK; }
    if (!s || key < 0 || key >= SR_DBG_COUNT) return fail(SR_ERR_INVALID_ARG, "bad argument to sr_debug_set");
    s->dbg[key] = value;
    return SR_OK;
}

void sr_reset_kernel_times(sr_scene* s) {
    if (!s) return;
    if (!s->parts.empty()) { for (sr_scene* q : s->parts) sr_reset_kernel_times(q); return; }
    for (int k = 0; k < sr::K_COUNT; ++k) s->ev_used[k] = 0;
}

int sr_kernel_times(sr_scene* s, sr_kernel_time* out, int32_t cap) {
    if (s && !s->parts.empty()) s = s->parts[0];               // surface passes / timings: the first device
    if (!s || !out || cap <= 0) return 0;
    if (use_device(s)) return 0;
    int n = 0;
    for (int k = 0; k < sr::K_COUNT && n < cap; ++k) {
        if (!s->ev_used[k]) continue;
        double total = 0;
        int ok = 0;
        for (int i = 0; i < s->ev_used[k]; ++i) {
            if (hipEventSynchronize(s->ev[k][2 * i + 1]) != hipSuccess) continue;
            float ms = 0;
            if (hipEventElapsedTime(&ms, s->ev[k][2 * i], s->ev[k][2 * i + 1]) != hipSuccess) continue;
            total += ms;
            ++ok;
        }
        if (!ok) continue;
        out[n].name = sr::kernel_name(k);
        out[n].ms = (float)total;
        out[n].launches = ok;
        ++n;
    }
    return n;
}

}  // extern "C"
