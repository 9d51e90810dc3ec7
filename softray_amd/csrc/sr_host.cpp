// sr_host.cpp -- host-side scene preparation (see sr_host.h).  Compile with -ffp-contract=off.
#include "sr_host.h"

#include <sched.h>

#include <algorithm>
#include <array>
#include <cfloat>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <atomic>
#include <map>
#include <memory>
#include <thread>

namespace sr {

// ------------------------------------------------------------------------------------------------
// small vector helpers; every expression keeps the reference's operand order (Engine3D/Vector.cs)
// ------------------------------------------------------------------------------------------------
static inline Vec3 sub(Vec3 a, Vec3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
static inline Vec3 scale(Vec3 a, double s) { return {a.x * s, a.y * s, a.z * s}; }
static inline double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }          // Vector.cs:99
static inline Vec3 cross(Vec3 a, Vec3 b) {                                                        // Vector.cs:104
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
static inline Vec3 unit(Vec3 v) {                                                                 // Vector.cs:177-185
    double len = std::sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    double inv = 1.0 / len;
    return {v.x * inv, v.y * inv, v.z * inv};
}
static inline bool is_zero_vector(Vec3 v) {                                                       // Vector.cs:140
    const double e = 1e-10;
    return -e < v.x && v.x < e && -e < v.y && v.y < e && -e < v.z && v.z < e;
}

// ------------------------------------------------------------------------------------------------
// System.Random
// ------------------------------------------------------------------------------------------------
NetRandom::NetRandom(int32_t seed) {
    const int32_t kBig = 2147483647, kSeed = 161803398;
    int32_t sub_ = (seed == std::numeric_limits<int32_t>::min()) ? kBig : (seed < 0 ? -seed : seed);
    int32_t mj = kSeed - sub_, mk = 1;
    std::memset(table_, 0, sizeof(table_));
    table_[55] = mj;
    for (int i = 1; i < 55; ++i) {
        int slot = (21 * i) % 55;
        table_[slot] = mk;
        mk = mj - mk;
        if (mk < 0) mk += kBig;
        mj = table_[slot];
    }
    for (int round = 0; round < 4; ++round)
        for (int i = 1; i <= 55; ++i) {
            uint32_t d = (uint32_t)table_[i] - (uint32_t)table_[1 + (i + 30) % 55];   // wraps like C# int
            table_[i] = (int32_t)d;
            if (table_[i] < 0) table_[i] += kBig;
        }
    inext_ = 0;
    inextp_ = 21;
}
int32_t NetRandom::sample() {
    const int32_t kBig = 2147483647;
    if (++inext_ >= 56) inext_ = 1;
    if (++inextp_ >= 56) inextp_ = 1;
    int32_t r = table_[inext_] - table_[inextp_];
    if (r == kBig) --r;
    if (r < 0) r += kBig;
    table_[inext_] = r;
    return r;
}
int32_t NetRandom::next() { return sample(); }
double NetRandom::next_double() { return sample() * (1.0 / 2147483647); }

// ------------------------------------------------------------------------------------------------
// primitive records
// ------------------------------------------------------------------------------------------------
Rec128 make_triangle_record(Vec3 v1, Vec3 v2, Vec3 v3, uint32_t color, int32_t aux) {
    Rec128 r;
    Vec3 e1 = sub(v2, v1), e2 = sub(v3, v1);
    Vec3 n = cross(e1, e2);
    if (is_zero_vector(n)) n = {1, 0, 0};                   // Triangle.cs:42-43
    Vec3 un = unit(n);                                      // Plane ctor, Plane.cs:25-27
    double d = dot(v1, un);
    Vec3 e1p = cross(e1, n), e2p = cross(e2, n);            // with the UN-normalised normal, Triangle.cs:49-50
    r.p[0] = un.x; r.p[1] = un.y; r.p[2] = un.z; r.p[3] = d;
    r.p[4] = v1.x; r.p[5] = v1.y; r.p[6] = v1.z;
    r.p[7] = e2p.x; r.p[8] = e2p.y; r.p[9] = e2p.z; r.p[10] = dot(e1, e2p);   // Triangle.cs:91
    r.p[11] = e1p.x; r.p[12] = e1p.y; r.p[13] = e1p.z; r.p[14] = dot(e2, e1p); // Triangle.cs:96
    r.color = color;
    r.aux = aux;
    return r;
}
Rec128 make_sphere_record(Vec3 c, double radius, uint32_t color) {
    Rec128 r;
    std::memset(&r, 0, sizeof(r));
    r.p[0] = c.x; r.p[1] = c.y; r.p[2] = c.z; r.p[3] = radius; r.p[4] = radius * radius;
    r.color = color;
    r.aux = 0;
    return r;
}
Rec128 make_plane_record(Vec3 point, Vec3 normal, uint32_t color) {
    Rec128 r;
    std::memset(&r, 0, sizeof(r));
    Vec3 un = unit(normal);
    r.p[0] = un.x; r.p[1] = un.y; r.p[2] = un.z; r.p[3] = dot(point, un);
    r.color = color;
    r.aux = 1;
    return r;
}
Rec128 make_box_record(Vec3 mn, Vec3 mx) {
    Rec128 r;
    std::memset(&r, 0, sizeof(r));
    r.p[0] = mn.x; r.p[1] = mn.y; r.p[2] = mn.z; r.p[3] = mx.x; r.p[4] = mx.y; r.p[5] = mx.z;
    const Vec3 nrm[6] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 6; ++i) r.p[6 + i] = make_plane_record(i < 3 ? mn : mx, nrm[i], 0).p[3];   // new Plane(min / max, normal).originDist
    r.color = 0xffffffffu;                                 // a hit carries its plane's colour: Color.White (Plane.cs:28)
    r.aux = 4;
    return r;
}
RootBox make_root_box(const double bmin[3], const double bmax[3]) {
    RootBox b;
    const double eps = 1e-10;
    Vec3 mn = {bmin[0], bmin[1], bmin[2]}, mx = {bmax[0], bmax[1], bmax[2]};
    for (int a = 0; a < 3; ++a) {
        b.min[a] = bmin[a]; b.max[a] = bmax[a];
        b.lo[a] = bmin[a] - eps; b.hi[a] = bmax[a] + eps;
        b.centre[a] = (bmin[a] + bmax[a]) * 0.5;
    }
    // the six Plane(point, normal) of AxisAlignedBox.cs:22-27; the unit() call reproduces Plane's ctor
    const Vec3 nrm[6] = {{-1, 0, 0}, {0, -1, 0}, {0, 0, -1}, {1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int i = 0; i < 6; ++i) b.pd[i] = dot(i < 3 ? mn : mx, unit(nrm[i]));
    return b;
}

// ------------------------------------------------------------------------------------------------
// reference tree
// ------------------------------------------------------------------------------------------------
namespace {
struct RefBuilder {
    const std::vector<double>& v9;
    int max_depth, max_per_leaf;
    RefTree& t;

    Vec3 vert(int tri, int k) const { const double* p = &v9[(size_t)tri * 9 + 3 * k]; return {p[0], p[1], p[2]}; }

    int32_t make_leaf(int32_t idx, const std::vector<int32_t>& geom, Vec3 mn, Vec3 mx, int depth) {
        const double eps = 1e-10;
        RefNode& n = t.nodes[idx];
        n.axis = -1;
        n.split = 0;
        n.a = (int32_t)t.leaf_tris.size();
        n.b = (int32_t)geom.size();
        n.box = (int32_t)t.leaf_boxes.size();
        t.leaf_tris.insert(t.leaf_tris.end(), geom.begin(), geom.end());
        LeafBox lb = {{mn.x - eps, mn.y - eps, mn.z - eps}, {mx.x + eps, mx.y + eps, mx.z + eps}};
        t.leaf_boxes.push_back(lb);
        t.num_leaf_nodes++;
        t.max_stack = std::max(t.max_stack, depth);
        return idx;
    }

    // Node ctor + RecursivePlaneSplit (SpatialSubdivision.cs:39-230)
    int32_t split(std::vector<int32_t>& geom, Vec3 mn, Vec3 mx, int depth) {
        int32_t idx = (int32_t)t.nodes.size();
        t.nodes.push_back(RefNode{});
        t.num_nodes++;
        t.tree_depth = std::max(t.tree_depth, depth);
        if (depth >= max_depth || (int)geom.size() <= max_per_leaf) return make_leaf(idx, geom, mn, mx, depth);

        Vec3 ext = {std::fabs(mx.x - mn.x), std::fabs(mx.y - mn.y), std::fabs(mx.z - mn.z)};
        int axis;                                                      // the exact '>' cascade of :80-101
        if (ext.x > ext.y) axis = (ext.x > ext.z) ? 0 : 2;
        else               axis = (ext.y > ext.z) ? 1 : 2;
        Vec3 c = {(mn.x + mx.x) * 0.5, (mn.y + mx.y) * 0.5, (mn.z + mx.z) * 0.5};   // AxisAlignedBox.Centre
        Vec3 pn = unit(axis == 0 ? Vec3{1, 0, 0} : axis == 1 ? Vec3{0, 1, 0} : Vec3{0, 0, 1});
        double pd = dot(c, pn);                                        // Plane ctor, Plane.cs:25-27

        std::vector<int32_t> ns, bs;
        for (int32_t g : geom) {                                       // Triangle.IntersectPlane, Triangle.cs:125-131
            bool anyN = false, anyB = false;
            for (int k = 0; k < 3; ++k) {
                if (dot(vert(g, k), pn) >= pd) anyN = true; else anyB = true;      // Point.cs:35-50
            }
            if (anyN) ns.push_back(g);
            if (anyB) bs.push_back(g);
        }
        if (ns.size() == geom.size() || bs.size() == geom.size())      // rejected split, :167-181
            return make_leaf(idx, geom, mn, mx, depth);
        std::vector<int32_t>().swap(geom);                             // geometry.Clear()

        Vec3 back_max = mx, norm_min = mn;
        if (axis == 0) back_max.x = norm_min.x = c.x;
        else if (axis == 1) back_max.y = norm_min.y = c.y;
        else back_max.z = norm_min.z = c.z;

        int32_t a = -1, b = -1;
        if (!ns.empty()) a = split(ns, norm_min, mx, depth + 1);       // normal side first, :199-203
        if (!bs.empty()) b = split(bs, mn, back_max, depth + 1);
        RefNode& n = t.nodes[idx];
        n.axis = axis;
        n.split = pd;
        n.a = a;
        n.b = b;
        n.box = -1;
        return idx;
    }
};
}  // namespace

bool build_ref_tree(const std::vector<double>& v9, const double bmin[3], const double bmax[3],
                    int max_depth, int max_per_leaf, RefTree& out) {
    out = RefTree();
    const double eps = 1e-10;
    size_t n = v9.size() / 9;
    for (size_t i = 0; i < 3 * n; ++i) {                               // ContainsPoint per vertex, :287-295
        const double* p = &v9[3 * i];
        for (int a = 0; a < 3; ++a)
            if (!(bmin[a] - eps < p[a] && p[a] < bmax[a] + eps)) return false;
    }
    RefBuilder rb{v9, max_depth, max_per_leaf, out};
    std::vector<int32_t> all(n);
    for (size_t i = 0; i < n; ++i) all[i] = (int32_t)i;
    rb.split(all, Vec3{bmin[0], bmin[1], bmin[2]}, Vec3{bmax[0], bmax[1], bmax[2]}, 1);
    out.built = true;
    return true;
}

// ------------------------------------------------------------------------------------------------
// own BVH: binned SAH over centroids, <= 7 triangles per leaf, children's boxes stored in the parent
// ------------------------------------------------------------------------------------------------
namespace {
struct Aabb {
    double lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; ++a) { lo[a] = DBL_MAX; hi[a] = -DBL_MAX; } }
    void grow(const Aabb& o) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); } }
    void grow(const double p[3]) { for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    double half_area() const {
        double dx = hi[0] - lo[0], dy = hi[1] - lo[1], dz = hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    }
};
struct ChildRef { int32_t c, n; Aabb box; };

// Binned-SAH builder.  The triangles' boxes travel WITH the order (an array of {box, index} records that every split
// partitions stably from one buffer into the other), so every pass streams through memory instead of gathering through an
// index list (the gathers made the build cache-miss bound: 8.6 s at 10 M triangles).  The work is spread over the host's
// cores without changing the tree: the top of the tree (ranges above kTaskSize triangles) is built by one thread whose
// passes (binning, partition) run in chunks on all threads; every range at or below kTaskSize becomes a task that a pool
// of threads builds into its own node vector, and the vectors are appended to the top part in range order.  min / max /
// integer counts are order independent and the partition is stable, so tree, triangle order and node numbers do not
// depend on the thread count.
struct BvhBuilder {
    static constexpr int kBins = 16;
    static constexpr int kTaskSize = 65536;      // triangles per parallel subtree task
    static constexpr int kMaxThreads = 16;
    struct Prim { double lo[3], hi[3]; int32_t idx, pad; };
    struct Bounds {
        Aabb box, cb;                // of the triangles / of their centroids
        void reset() { box.reset(); cb.reset(); }
        void grow(const Prim& p) {
            for (int a = 0; a < 3; ++a) {
                box.lo[a] = std::min(box.lo[a], p.lo[a]); box.hi[a] = std::max(box.hi[a], p.hi[a]);
                const double c = 0.5 * (p.lo[a] + p.hi[a]);
                cb.lo[a] = std::min(cb.lo[a], c); cb.hi[a] = std::max(cb.hi[a], c);
            }
        }
        void grow(const Bounds& o) { box.grow(o.box); cb.grow(o.cb); }
    };
    struct Bins {
        Aabb bb[kBins]; int cnt[kBins];
        void reset() { for (int k = 0; k < kBins; ++k) { bb[k].reset(); cnt[k] = 0; } }
    };
    std::vector<Prim> buf[2];        // ping-pong: a split reads one and writes the other
    Bvh& out;
    const RootBox& root;
    double pad;
    int kLeafMax = 4;                // triangles per leaf; set by build_bvh.  Round 3, four-wide packet walks with per-frame ordered children: 2: 11.9 ms, 3: 11.6, 4: 11.5, 5: 11.6, 7: 11.8, 10: 12.3, 14: 12.9 (round 2's binary packet walks preferred 7: 2: 22.7, 4: 20.8, 7: 19.5, 10: 19.9) (sr_debug_set(SR_DBG_BVH_LEAF) = 1..15 overrides)
    int threads = 1;

    struct Task { int b, e, depth; int32_t parent; int side; Bounds bd; int src; };
    // where a (sub)tree's nodes go: the top part pushes onto out.nodes, a task fills its own slice of one array that the
    // calling thread allocated (no allocation inside the workers: concurrent heap growth serialises on the address space)
    struct Dest {
        std::vector<BvhNode>* vec; BvhNode* base; int count; int depth; std::vector<Task>* defer;
        int32_t alloc() { if (vec) { vec->push_back(BvhNode{}); return (int32_t)vec->size() - 1; } base[count] = BvhNode{}; return count++; }
        BvhNode& at(int32_t i) { return vec ? (*vec)[i] : base[i]; }
    };

    BvhBuilder(Bvh& o, const RootBox& r, double p) : out(o), root(r), pad(p) {}

    float down(double v) const { float f = (float)v; if ((double)f > v) f = std::nextafterf(f, -INFINITY); return f; }
    float up(double v) const { float f = (float)v; if ((double)f < v) f = std::nextafterf(f, INFINITY); return f; }

    void store(float lo[3], float hi[3], const Aabb& b) const {
        for (int a = 0; a < 3; ++a) {
            lo[a] = down(b.lo[a] - root.centre[a] - pad);
            hi[a] = up(b.hi[a] - root.centre[a] + pad);
        }
    }

    // fn(chunk_begin, chunk_end, chunk_index) over [b, e): chunk 0 on the caller, the others on their own threads
    template <class F> static void chunks(int b, int e, int nchunks, F fn) {
        std::vector<std::thread> pool;
        const long long len = (long long)e - b;
        for (int k = 1; k < nchunks; ++k)
            pool.emplace_back(fn, b + (int)(len * k / nchunks), b + (int)(len * (k + 1) / nchunks), k);
        fn(b, b + (int)(len / nchunks), 0);
        for (auto& t : pool) t.join();
    }

    ChildRef build(int b, int e, int depth, const Bounds& bd, int src, Dest& d, int32_t parent, int side) {
        d.depth = std::max(d.depth, depth);
        const Prim* S = buf[src].data();
        Prim* D = buf[1 - src].data();
        if (e - b <= kLeafMax) {
            for (int i = b; i < e; ++i) out.order[i] = S[i].idx;
            return ChildRef{b, e - b, bd.box};
        }
        if (d.defer && parent >= 0 && e - b <= kTaskSize) {             // a subtree for the pool; its root lands in nodes[parent]'s child slot
            d.defer->push_back(Task{b, e, depth, parent, side, bd, src});
            return ChildRef{-1, 0, bd.box};
        }
        const int nch = (d.defer && threads > 1) ? threads : 1;         // top of the tree: passes in parallel chunks

        int axis = 0;
        double ext = bd.cb.hi[0] - bd.cb.lo[0];
        for (int a = 1; a < 3; ++a) if (bd.cb.hi[a] - bd.cb.lo[a] > ext) { ext = bd.cb.hi[a] - bd.cb.lo[a]; axis = a; }
        int mid = -1;
        Bounds bl, br;
        bl.reset(); br.reset();
        if (ext > 0) {
            const double k1 = kBins * (1.0 - 1e-9) / ext, c0 = bd.cb.lo[axis];
            auto bin_of = [&](const Prim& p) {
                int k = (int)((0.5 * (p.lo[axis] + p.hi[axis]) - c0) * k1);
                return std::min(std::max(k, 0), kBins - 1);
            };
            Bins pt1;
            std::vector<Bins> ptv;
            if (nch > 1) ptv.resize(nch);
            Bins* pt = nch > 1 ? ptv.data() : &pt1;
            auto fill = [&](int b0, int e0, int c) {
                Bins& t = pt[c];
                t.reset();
                for (int i = b0; i < e0; ++i) {
                    const int k = bin_of(S[i]);
                    for (int a = 0; a < 3; ++a) { t.bb[k].lo[a] = std::min(t.bb[k].lo[a], S[i].lo[a]); t.bb[k].hi[a] = std::max(t.bb[k].hi[a], S[i].hi[a]); }
                    t.cnt[k]++;
                }
            };
            if (nch > 1) chunks(b, e, nch, fill); else fill(b, e, 0);
            Bins bins; bins.reset();
            for (int t = 0; t < nch; ++t)
                for (int k = 0; k < kBins; ++k) { if (pt[t].cnt[k]) bins.bb[k].grow(pt[t].bb[k]); bins.cnt[k] += pt[t].cnt[k]; }
            const Aabb* bb = bins.bb; const int* cnt = bins.cnt;
            double rightA[kBins]; int rightN[kBins];
            Aabb acc; acc.reset(); int accn = 0;
            for (int k = kBins - 1; k > 0; --k) { acc.grow(bb[k]); accn += cnt[k]; rightA[k] = accn ? acc.half_area() : 0; rightN[k] = accn; }
            acc.reset(); accn = 0;
            double best = DBL_MAX; int bestk = -1;
            for (int k = 0; k < kBins - 1; ++k) {
                if (cnt[k]) acc.grow(bb[k]);
                accn += cnt[k];
                if (accn == 0 || rightN[k + 1] == 0) continue;
                double cost = acc.half_area() * accn + rightA[k + 1] * rightN[k + 1];
                if (cost < best) { best = cost; bestk = k; }
            }
            if (bestk >= 0) {
                // stable partition S -> D; every chunk knows from its bins where its left and right runs start
                int ol[kMaxThreads], orr[kMaxThreads];
                int total_l = 0;
                for (int c = 0; c < nch; ++c) for (int k = 0; k <= bestk; ++k) total_l += pt[c].cnt[k];
                for (int c = 0, l = b, r = b + total_l; c < nch; ++c) {
                    int nl = 0, nall = 0;
                    for (int k = 0; k < kBins; ++k) { nall += pt[c].cnt[k]; if (k <= bestk) nl += pt[c].cnt[k]; }
                    ol[c] = l; orr[c] = r; l += nl; r += nall - nl;
                }
                Bounds pl[kMaxThreads], pr[kMaxThreads];
                auto scatter = [&](int b0, int e0, int c) {
                    int l = ol[c], r = orr[c];
                    Bounds x, y; x.reset(); y.reset();
                    for (int i = b0; i < e0; ++i) {
                        if (bin_of(S[i]) <= bestk) { D[l++] = S[i]; x.grow(S[i]); } else { D[r++] = S[i]; y.grow(S[i]); }
                    }
                    pl[c] = x; pr[c] = y;
                };
                if (nch > 1) chunks(b, e, nch, scatter); else scatter(b, e, 0);
                for (int c = 0; c < nch; ++c) { bl.grow(pl[c]); br.grow(pr[c]); }
                mid = b + total_l;
            }
        }
        if (mid <= b || mid >= e) {                                    // degenerate: equal halves by index
            mid = (b + e) / 2;
            std::copy(S + b, S + e, D + b);
            std::nth_element(D + b, D + mid, D + e, [&](const Prim& x, const Prim& y) {
                const double cx = 0.5 * (x.lo[axis] + x.hi[axis]), cy = 0.5 * (y.lo[axis] + y.hi[axis]);
                return cx < cy || (cx == cy && x.idx < y.idx);
            });
            bl.reset(); br.reset();
            for (int i = b; i < mid; ++i) bl.grow(D[i]);
            for (int i = mid; i < e; ++i) br.grow(D[i]);
        }
        const int32_t idx = d.alloc();
        ChildRef l = build(b, mid, depth + 1, bl, 1 - src, d, idx, 0);
        ChildRef r = build(mid, e, depth + 1, br, 1 - src, d, idx, 1);
        BvhNode& n = d.at(idx);
        store(n.lo0, n.hi0, l.box); n.c0 = l.c; n.n0 = l.n;
        store(n.lo1, n.hi1, r.box); n.c1 = r.c; n.n1 = r.n;
        return ChildRef{idx, 0, bd.box};
    }

    ChildRef build_all(int n, const Bounds& all) {
        std::vector<Task> tasks;
        Dest top{&out.nodes, nullptr, 0, 0, n > 2 * kTaskSize ? &tasks : nullptr};   // (also with one thread: same node numbering)
        ChildRef r = build(0, n, 1, all, 0, top, -1, 0);
        out.depth = top.depth;
        if (tasks.empty()) return r;
        // the subtrees: a task over triangles [b, e) has fewer than e - b inner nodes and writes them (links local) to slots
        // [b, ...) of one array; tasks are taken from a shared counter
        std::unique_ptr<BvhNode[]> slots(new BvhNode[(size_t)n]);
        std::vector<int> sub_count(tasks.size(), 0), sub_depth(tasks.size(), 0);
        std::atomic<size_t> next{0};
        auto worker = [&]() {
            for (size_t k = next.fetch_add(1); k < tasks.size(); k = next.fetch_add(1)) {
                Dest d{nullptr, slots.get() + tasks[k].b, 0, 0, nullptr};
                build(tasks[k].b, tasks[k].e, tasks[k].depth, tasks[k].bd, tasks[k].src, d, -1, 0);   // (> kLeafMax triangles: its root is inner, local index 0)
                sub_count[k] = d.count;
                sub_depth[k] = d.depth;
            }
        };
        std::vector<std::thread> pool;
        for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
        // append in range order, rebase the inner links, hook every subtree root into its parent
        size_t total = out.nodes.size();
        for (int c : sub_count) total += (size_t)c;
        out.nodes.reserve(total);
        for (size_t k = 0; k < tasks.size(); ++k) {
            const int32_t base = (int32_t)out.nodes.size();
            const BvhNode* src_nodes = slots.get() + tasks[k].b;
            for (int i = 0; i < sub_count[k]; ++i) {
                BvhNode nd = src_nodes[i];
                if (nd.n0 == 0) nd.c0 += base;
                if (nd.n1 == 0) nd.c1 += base;
                out.nodes.push_back(nd);
            }
            BvhNode& p = out.nodes[tasks[k].parent];
            if (tasks[k].side == 0) p.c0 = base; else p.c1 = base;
            out.depth = std::max(out.depth, sub_depth[k]);
        }
        return r;
    }
};
}  // namespace

// cores this process may actually use: its affinity mask, capped by the cgroup's CPU quota (a shared GPU box gives every job a
// share; hardware_concurrency() would report the whole machine and oversubscribe it)
static int usable_cores() {
    int cores = (int)std::max(1u, std::thread::hardware_concurrency());
    cpu_set_t set;
    CPU_ZERO(&set);
    if (sched_getaffinity(0, sizeof(set), &set) == 0) { const int c = CPU_COUNT(&set); if (c > 0) cores = std::min(cores, c); }
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        long long per = 0;
        if (std::fscanf(f, "%63s %lld", q, &per) == 2 && std::strcmp(q, "max") != 0 && per > 0) {
            const long long quota = std::atoll(q);
            if (quota > 0) cores = (int)std::max<long long>(1, std::min<long long>(cores, (quota + per - 1) / per));
        }
        std::fclose(f);
    }
    return cores;
}

void build_bvh(const std::vector<double>& v9, const RootBox& root, Bvh& out, int leaf_max, int threads) {
    out = Bvh();
    size_t n = v9.size() / 9;
    double ext = 0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, root.max[a] - root.min[a]);
    BvhBuilder bb(out, root, std::ldexp(ext > 0 ? ext : 1.0, -16));
    bb.kLeafMax = std::min(15, std::max(1, leaf_max));
    if (threads <= 0) threads = std::min(16, usable_cores());
    bb.threads = n > (size_t)2 * BvhBuilder::kTaskSize ? std::max(1, std::min(threads, (int)BvhBuilder::kMaxThreads)) : 1;
    bb.buf[0].resize(n);
    bb.buf[1].resize(n);
    out.order.resize(n);
    std::vector<BvhBuilder::Bounds> part(bb.threads);
    auto prep = [&](int b0, int e0, int c) {
        BvhBuilder::Bounds acc; acc.reset();
        for (size_t i = (size_t)b0; i < (size_t)e0; ++i) {
            Aabb b; b.reset();
            for (int k = 0; k < 3; ++k) b.grow(&v9[i * 9 + 3 * k]);
            BvhBuilder::Prim& p = bb.buf[0][i];
            for (int a = 0; a < 3; ++a) { p.lo[a] = b.lo[a]; p.hi[a] = b.hi[a]; }
            p.idx = (int32_t)i; p.pad = 0;
            acc.grow(p);
        }
        part[c] = acc;
    };
    if (bb.threads > 1) BvhBuilder::chunks(0, (int)n, bb.threads, prep); else prep(0, (int)n, 0);
    BvhBuilder::Bounds all; all.reset();
    for (const auto& q : part) all.grow(q);
    auto empty_child = [](float lo[3], float hi[3], int32_t& c, int32_t& cn) {
        for (int a = 0; a < 3; ++a) { lo[a] = 1.0f; hi[a] = -1.0f; }
        c = 0; cn = -1;
    };
    if (n == 0) {
        BvhNode r{};
        empty_child(r.lo0, r.hi0, r.c0, r.n0);
        empty_child(r.lo1, r.hi1, r.c1, r.n1);
        out.nodes.push_back(r);
    } else {
        ChildRef top = bb.build_all((int)n, all);
        if (top.n > 0) {                                               // whole scene fits one leaf
            BvhNode r{};
            bb.store(r.lo0, r.hi0, top.box); r.c0 = top.c; r.n0 = top.n;
            empty_child(r.lo1, r.hi1, r.c1, r.n1);
            out.nodes.insert(out.nodes.begin(), r);
        }
    }
    out.built = true;
}

int collapse_bvh4(const BvhNode* nodes, size_t num_nodes, std::vector<Bvh4Node>& out) {
    out.clear();
    if (num_nodes == 0) return 0;
    out.reserve(num_nodes / 2 + 2);
    auto child_of = [](const BvhNode& n, int side) {
        Bvh4Child c;
        for (int a = 0; a < 3; ++a) { c.lo[a] = side ? n.lo1[a] : n.lo0[a]; c.hi[a] = side ? n.hi1[a] : n.hi0[a]; }
        c.c = side ? n.c1 : n.c0;
        c.n = side ? n.n1 : n.n0;
        return c;
    };
    auto area = [](const Bvh4Child& c) {
        const double dx = (double)c.hi[0] - c.lo[0], dy = (double)c.hi[1] - c.lo[1], dz = (double)c.hi[2] - c.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    struct Item { int32_t src, dst, depth; };
    std::vector<Item> stack;
    out.push_back(Bvh4Node{});
    stack.push_back({0, 0, 1});
    int depth = 0;
    while (!stack.empty()) {
        const Item it = stack.back();
        stack.pop_back();
        depth = std::max(depth, it.depth);
        Bvh4Child ch[4];
        int k = 0;
        for (int side = 0; side < 2; ++side) {
            const Bvh4Child c = child_of(nodes[it.src], side);
            if (c.n >= 0) ch[k++] = c;                                  // (empty children exist only in the root of a tiny scene)
        }
        while (k < 4) {
            int best = -1;
            double best_area = -1.0;
            for (int i = 0; i < k; ++i)
                if (ch[i].n == 0) { const double a = area(ch[i]); if (a > best_area) { best_area = a; best = i; } }
            if (best < 0) break;
            const BvhNode& m = nodes[ch[best].c];
            const Bvh4Child l = child_of(m, 0), r = child_of(m, 1);
            // the left child takes the parent's slot, the right one follows it: build (= spatial split) order is kept
            for (int i = k; i > best + 1; --i) ch[i] = ch[i - 1];
            ch[best] = l;
            ch[best + 1] = r;
            ++k;
        }
        for (int i = 0; i < k; ++i) {
            if (ch[i].n == 0) {
                const int32_t dst = (int32_t)out.size();
                out.push_back(Bvh4Node{});
                stack.push_back({ch[i].c, dst, it.depth + 1});
                ch[i].c = dst;
            }
        }
        for (int i = k; i < 4; ++i) {
            for (int a = 0; a < 3; ++a) { ch[i].lo[a] = 1.0f; ch[i].hi[a] = -1.0f; }
            ch[i].c = 0; ch[i].n = -1;
        }
        for (int i = 0; i < 4; ++i) out[it.dst].ch[i] = ch[i];
    }
    return depth;
}

// ------------------------------------------------------------------------------------------------
// Instance / Renderer helpers
// ------------------------------------------------------------------------------------------------
namespace {
struct M4 { double m[4][4]; };
M4 zero4() { M4 r; std::memset(&r, 0, sizeof(r)); return r; }
M4 mul(const M4& a, const M4& b) {                                     // Matrix.cs:74-91
    M4 r = zero4();
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0.0;
            for (int k = 0; k < 4; ++k) s += a.m[i][k] * b.m[k][j];
            r.m[i][j] = s;
        }
    return r;
}
M4 translation(Vec3 p) {                                               // Matrix.cs:94-113
    M4 r = zero4();
    r.m[0][0] = r.m[1][1] = r.m[2][2] = r.m[3][3] = 1.0;
    r.m[0][3] = p.x; r.m[1][3] = p.y; r.m[2][3] = p.z;
    return r;
}
M4 yaw_m(double a) {                                                   // Matrix.cs:116-131
    M4 r = zero4();
    r.m[0][0] = std::cos(a); r.m[2][0] = std::sin(a); r.m[1][1] = 1.0;
    r.m[0][2] = -std::sin(a); r.m[2][2] = std::cos(a); r.m[3][3] = 1.0;
    return r;
}
M4 pitch_m(double a) {                                                 // Matrix.cs:134-149
    M4 r = zero4();
    r.m[0][0] = 1.0; r.m[1][1] = std::cos(a); r.m[2][1] = std::sin(a);
    r.m[1][2] = -std::sin(a); r.m[2][2] = std::cos(a); r.m[3][3] = 1.0;
    return r;
}
M4 roll_m(double a) {                                                  // Matrix.cs:152-167
    M4 r = zero4();
    r.m[0][0] = std::cos(a); r.m[1][0] = std::sin(a);
    r.m[0][1] = -std::sin(a); r.m[1][1] = std::cos(a); r.m[2][2] = 1.0; r.m[3][3] = 1.0;
    return r;
}
}  // namespace

void instance_matrices(const double position[3], double yaw, double pitch, double roll,
                       double transform[12], double inv_transform[12]) {
    Vec3 p = {position[0], position[1], position[2]};
    Vec3 np = {-position[0], -position[1], -position[2]};
    M4 t = mul(mul(mul(translation(p), roll_m(roll)), pitch_m(pitch)), yaw_m(yaw));
    M4 it = mul(mul(mul(yaw_m(-yaw), pitch_m(-pitch)), roll_m(-roll)), translation(np));
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 4; ++c) { transform[r * 4 + c] = t.m[r][c]; inv_transform[r * 4 + c] = it.m[r][c]; }
}

double default_fov_depth() {
    const double deg = 45.0;                                           // Renderer.cs:97
    const double rad = deg / 180.0 * M_PI;                             // :100
    return 0.5 / std::tan(rad / 2);                                    // :101
}

void area_light_offsets(int32_t seed, int32_t count, double* out3) {
    NetRandom rnd(seed);
    for (int i = 0; i < count; ++i) {
        double x = rnd.next_double() * 2 - 1;
        double y = rnd.next_double() * 2 - 1;
        double z = rnd.next_double() * 2 - 1;
        Vec3 o = scale(unit(Vec3{x, y, z}), 0.2);                      // Normalise(); offset *= 0.2
        out3[3 * i] = o.x; out3[3 * i + 1] = o.y; out3[3 * i + 2] = o.z;
    }
}

// ------------------------------------------------------------------------------------------------
// .3DS -> unit-cube model (ThreeDSFile.cs:132-662, Model.cs:522-653,750-831)
// ------------------------------------------------------------------------------------------------
namespace {
struct Cursor {
    const uint8_t* base; size_t len; size_t at; bool ok;
    template <class T> T get() {
        T v{};
        if (at + sizeof(T) > len) { ok = false; return v; }
        std::memcpy(&v, base + at, sizeof(T));
        at += sizeof(T);
        return v;
    }
    std::string zstr() {
        std::string s;
        for (;;) { uint8_t c = get<uint8_t>(); if (!ok || c == 0) break; s.push_back((char)c); }
        return s;
    }
};
struct Chunk { uint16_t id; size_t begin, end; };
bool open_chunk(Cursor& c, Chunk& k) {
    k.begin = c.at;
    k.id = c.get<uint16_t>();
    uint32_t n = c.get<uint32_t>();
    if (!c.ok || n < 6) { c.ok = false; return false; }
    k.end = k.begin + n;
    return true;
}
struct Mesh {
    std::vector<Vec3> verts; std::vector<int32_t> faces; std::vector<int32_t> face_mat;
    bool got_verts = false, got_faces = false;
};
struct Parser3ds {
    Cursor c;
    std::vector<std::array<float, 3>> mat_diffuse;
    std::map<std::string, int> mat_index;
    std::vector<Mesh> meshes;

    void colour(float rgb[3]) {                                        // ProcessColorChunk: first sub-chunk only
        Chunk k;
        rgb[0] = rgb[1] = rgb[2] = 1.0f;
        if (!open_chunk(c, k)) return;
        if (k.id == 0x0010) { rgb[0] = c.get<float>(); rgb[1] = c.get<float>(); rgb[2] = c.get<float>(); }
        else if (k.id == 0x0011) {
            rgb[0] = (float)c.get<uint8_t>() / 255.0f; rgb[1] = (float)c.get<uint8_t>() / 255.0f; rgb[2] = (float)c.get<uint8_t>() / 255.0f;
        }
        c.at = k.end;
    }
    void material(size_t end) {                                        // ProcessMaterialChunk
        std::string name;
        std::array<float, 3> diffuse = {0.0f, 0.0f, 0.0f};             // Material.cs:33 default
        while (c.ok && c.at < end) {
            Chunk k;
            if (!open_chunk(c, k)) break;
            if (k.id == 0xA000) name = c.zstr();
            else if (k.id == 0xA020) colour(diffuse.data());
            c.at = k.end;
        }
        if (!mat_index.count(name)) { mat_index[name] = (int)mat_diffuse.size(); mat_diffuse.push_back(diffuse); }
    }
    void object(size_t end, Mesh& m) {                                 // ProcessObjectChunk / ProcessFaceChunk
        while (c.ok && c.at < end) {
            Chunk k;
            if (!open_chunk(c, k)) break;
            if (k.id == 0x4100) object(k.end, m);
            else if (k.id == 0x4110) {
                int n = c.get<uint16_t>();
                m.verts.resize(n);
                for (int i = 0; i < n; ++i) {
                    float a = c.get<float>(), b = c.get<float>(), d = c.get<float>();
                    m.verts[i] = {(double)a, (double)d, (double)(-b)};  // (x, z, -y), ThreeDSFile.cs:627
                }
                m.got_verts = true;
            } else if (k.id == 0x4120) {
                int n = c.get<uint16_t>();
                m.faces.resize((size_t)n * 3);
                m.face_mat.assign(n, -1);
                for (int i = 0; i < n; ++i) {
                    for (int j = 0; j < 3; ++j) m.faces[3 * i + j] = c.get<uint16_t>();
                    c.get<uint16_t>();
                }
                m.got_faces = true;
                while (c.ok && c.at < k.end) {
                    Chunk f;
                    if (!open_chunk(c, f)) break;
                    if (f.id == 0x4130) {
                        std::string nm = c.zstr();
                        auto it = mat_index.find(nm);
                        int mi = it == mat_index.end() ? -1 : it->second;
                        int cnt = c.get<uint16_t>();
                        for (int i = 0; i < cnt; ++i) {
                            int fi = c.get<uint16_t>();
                            if (fi < (int)m.face_mat.size()) m.face_mat[fi] = mi; else c.ok = false;
                        }
                    }
                    c.at = f.end;
                }
            }
            c.at = k.end;
        }
    }
    void top(size_t end) {                                             // ProcessChunk
        while (c.ok && c.at < end) {
            Chunk k;
            if (!open_chunk(c, k)) break;
            if (k.id == 0x0002) { c.get<int32_t>(); continue; }         // version: not skipped to its end (:241-244)
            if (k.id == 0x3D3D) {
                Chunk first;                                           // first sub-chunk is read and skipped (:204-212)
                if (open_chunk(c, first)) c.at = first.end;
                top(k.end);
            } else if (k.id == 0xAFFF) material(k.end);
            else if (k.id == 0x4000) {
                c.zstr();
                Mesh m;
                object(k.end, m);
                if (m.got_verts && m.got_faces) meshes.push_back(std::move(m));
            }
            c.at = k.end;
        }
    }
};
inline uint8_t to_byte(double d) {                                     // C# unchecked (byte)(double)
    if (!(d > -2147483649.0 && d < 2147483648.0)) return 0;
    return (uint8_t)(int32_t)d;
}
}  // namespace

std::string load_3ds(const uint8_t* data, size_t len, LoadedModel& out) {
    Parser3ds p;
    p.c = Cursor{data, len, 0, true};
    Chunk primary;
    if (!open_chunk(p.c, primary) || primary.id != 0x4D4D) return "Not a proper 3DS file.";
    p.top(std::min(primary.end, len));
    if (!p.c.ok) return "3DS file truncated or corrupt.";
    if (p.meshes.empty()) return "No entities in model. 3DS file may be corrupt.";

    std::vector<Vec3> verts;
    std::vector<int32_t> tri;
    out.argb.clear();
    double mn[3] = {DBL_MAX, DBL_MAX, DBL_MAX}, mx[3] = {-DBL_MAX, -DBL_MAX, -DBL_MAX};
    for (const Mesh& m : p.meshes) {
        if (m.verts.size() < 3) return "Entity has less than 3 vertices. 3DS file may be corrupt.";
        if (m.faces.empty()) return "Entity has no triangles. 3DS file may be corrupt.";
        int32_t base = (int32_t)verts.size();
        for (Vec3 v : m.verts) {                                       // Model.cs:585-610
            double q[3] = {v.x, v.y, v.z};
            for (int a = 0; a < 3; ++a) {
                if (std::isnan(q[a]) || std::isinf(q[a]) || std::fabs(q[a]) > 1e6) q[a] = 0.0;
                mn[a] = std::min(mn[a], q[a]);
                mx[a] = std::max(mx[a], q[a]);
            }
            verts.push_back({q[0], q[1], q[2]});
        }
        for (size_t f = 0; f < m.face_mat.size(); ++f) {
            for (int j = 0; j < 3; ++j) tri.push_back(base + m.faces[3 * f + j]);
            float d[3] = {0.0f, 0.0f, 0.0f};
            if (m.face_mat[f] >= 0) for (int j = 0; j < 3; ++j) d[j] = p.mat_diffuse[m.face_mat[f]][j];
            // Model.cs:98-100 (float -> double), Renderer.cs:1463 PackColorAndAlpha(diffuse, 1.0), Surface.cs:131-138
            uint32_t r = to_byte((double)d[0] * 255.0), g = to_byte((double)d[1] * 255.0), b = to_byte((double)d[2] * 255.0);
            uint32_t a = to_byte(1.0 * 255.0);
            out.argb.push_back((a << 24) + (r << 16) + (g << 8) + b);
        }
    }
    for (int32_t i : tri) if (i < 0 || i >= (int32_t)verts.size()) return "Triangle vertex index out of range.";
    // PostProcessGeometry, Model.cs:762-790
    Vec3 centre = {(mn[0] + mx[0]) / 2, (mn[1] + mx[1]) / 2, (mn[2] + mx[2]) / 2};
    double ex = mx[0] - mn[0], ey = mx[1] - mn[1], ez = mx[2] - mn[2];
    double s = 1.0 / std::max(std::max(ex, ey), ez);
    for (Vec3& v : verts) v = scale(sub(v, centre), s);
    Vec3 nmn = scale(sub(Vec3{mn[0], mn[1], mn[2]}, centre), s), nmx = scale(sub(Vec3{mx[0], mx[1], mx[2]}, centre), s);
    out.bmin[0] = nmn.x; out.bmin[1] = nmn.y; out.bmin[2] = nmn.z;
    out.bmax[0] = nmx.x; out.bmax[1] = nmx.y; out.bmax[2] = nmx.z;
    out.v9.resize(tri.size() * 3);
    for (size_t i = 0; i < tri.size(); ++i) {
        const Vec3& v = verts[tri[i]];
        out.v9[3 * i] = v.x; out.v9[3 * i + 1] = v.y; out.v9[3 * i + 2] = v.z;
    }
    return std::string();
}

}  // namespace sr
