// sr_lbvh.hip -- the own BVH built ON THE DEVICE (SURVEY.md 8f "next" row 2: at 1M-10M triangles the host build
// dominates end-to-end time).  Morton-ordered LBVH (Karras 2012): 63-bit Morton codes of the triangle-box centres ->
// radix sort (rocPRIM through hipCUB) -> binary radix tree -> bottom-up fp32 boxes -> subtrees of <= 7 triangles are
// collapsed into leaves -> the same 64-byte BvhNode / leaf-ordered record layout the SAH builder produces.
//
// Any BVH with conservative boxes gives the same pixels: the traversal (sr_trace.h) decides hits with the reference's
// FP64 triangle arithmetic and picks the nearest (ties -> lowest TriangleIndex) whatever the tree shape.  Boxes are
// rounded outward (directed double->float conversion) and padded by 2^-16 * extent exactly like sr_host.cpp::store().
#include "sr_device.h"

#include <hipcub/hipcub.hpp>

#include <algorithm>
#include <vector>

namespace sr {

namespace {

constexpr int kLeafMaxDefault = 4;   // like the host SAH builder (measured on the four-wide packet walks: 2: 11.9 ms, 4: 11.5, 7: 11.8, 14: 12.9)

struct FBox { float lo[3], hi[3]; };

__device__ __forceinline__ unsigned long long spread21(unsigned long long x) {   // 21 bits -> every third bit
    x &= 0x1fffffull;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}

// per triangle: conservative fp32 box (relative to the root centre, padded) + Morton key of its centre
__global__ void k_lbvh_keys(const double* __restrict__ v9, int n, RootBox root, float pad, FBox* __restrict__ tbox,
                            unsigned long long* __restrict__ keys, unsigned int* __restrict__ vals) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = v9 + (size_t)i * 9;
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        lo[a] = fmin(fmin(p[a], p[3 + a]), p[6 + a]);
        hi[a] = fmax(fmax(p[a], p[3 + a]), p[6 + a]);
    }
    FBox b;
    unsigned long long key = 0;
    for (int a = 0; a < 3; ++a) {
        b.lo[a] = __double2float_rd(lo[a] - root.centre[a] - (double)pad);
        b.hi[a] = __double2float_ru(hi[a] - root.centre[a] + (double)pad);
        const double ext = root.max[a] - root.min[a];
        double q = ext > 0 ? (0.5 * (lo[a] + hi[a]) - root.min[a]) / ext : 0.0;
        q = fmin(fmax(q, 0.0), 1.0);
        unsigned long long c = (unsigned long long)(q * 2097151.0);
        key |= spread21(c) << (2 - a);
    }
    tbox[i] = b;
    keys[i] = key;
    vals[i] = (unsigned)i;
}

// Karras' delta: length of the common prefix of keys i and j (ties broken by the position)
__device__ __forceinline__ int delta(const unsigned long long* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const unsigned long long a = keys[i], b = keys[j];
    if (a == b) return 64 + __clz((unsigned)(i ^ j));
    return __clzll((long long)(a ^ b));
}

// internal node i of the binary radix tree over the sorted keys; children: index >= 0 internal, ~leaf for a leaf
__global__ void k_lbvh_tree(const unsigned long long* __restrict__ keys, int n, int2* __restrict__ child, int2* __restrict__ range,
                            int* __restrict__ parent_int, int* __restrict__ parent_leaf) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    const int d = (delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1)) >= 0 ? 1 : -1;
    const int dmin = delta(keys, n, i, i - d);
    int lmax = 2;
    while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2;
    int l = 0;
    for (int t = lmax / 2; t >= 1; t /= 2)
        if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = delta(keys, n, i, j);
    int s = 0;
    for (int t = (l + 1) / 2;; t = (t + 1) / 2) {
        if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t;
        if (t == 1) break;
    }
    const int gamma = i + s * d + min(d, 0);
    const int first = min(i, j), last = max(i, j);
    const int left = (first == gamma) ? ~gamma : gamma;
    const int right = (last == gamma + 1) ? ~(gamma + 1) : gamma + 1;
    child[i] = make_int2(left, right);
    range[i] = make_int2(first, last);
    if (left >= 0) parent_int[left] = i; else parent_leaf[~left] = i;
    if (right >= 0) parent_int[right] = i; else parent_leaf[~right] = i;
    if (i == 0) parent_int[0] = -1;
}

__device__ __forceinline__ FBox merge(const FBox& a, const FBox& b) {
    FBox r;
    for (int k = 0; k < 3; ++k) { r.lo[k] = fminf(a.lo[k], b.lo[k]); r.hi[k] = fmaxf(a.hi[k], b.hi[k]); }
    return r;
}

// bottom-up boxes: the second thread to arrive at a node merges its children and continues upward
__global__ void k_lbvh_boxes(int n, const unsigned int* __restrict__ order, const FBox* __restrict__ tbox, const int2* __restrict__ child,
                             const int* __restrict__ parent_int, const int* __restrict__ parent_leaf, unsigned int* __restrict__ flags,
                             FBox* __restrict__ nbox) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    int node = parent_leaf[p];
    while (node >= 0) {
        __threadfence();
        if (atomicAdd(&flags[node], 1u) == 0u) return;            // first arrival: the sibling is not finished yet
        __threadfence();
        const int2 c = child[node];
        const FBox a = c.x >= 0 ? nbox[c.x] : tbox[order[~c.x]];
        const FBox b = c.y >= 0 ? nbox[c.y] : tbox[order[~c.y]];
        nbox[node] = merge(a, b);
        node = parent_int[node];
    }
}

__global__ void k_lbvh_mark(int n, const int2* __restrict__ range, int* __restrict__ keep, int kLeafMax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1) return;
    keep[i] = (range[i].y - range[i].x + 1) > kLeafMax ? 1 : 0;
}

__device__ __forceinline__ void put_child(float* lo, float* hi, int32_t& c, int32_t& cn, int ch, const int2* range, const int* keep,
                                          const int* outidx, const FBox* nbox, const FBox* tbox, const unsigned int* order) {
    FBox b;
    if (ch < 0) { c = ~ch; cn = 1; b = tbox[order[~ch]]; }
    else if (!keep[ch]) { c = range[ch].x; cn = range[ch].y - range[ch].x + 1; b = nbox[ch]; }
    else { c = outidx[ch]; cn = 0; b = nbox[ch]; }
    for (int k = 0; k < 3; ++k) { lo[k] = b.lo[k]; hi[k] = b.hi[k]; }
}

__global__ void k_lbvh_emit(int n, const int2* __restrict__ child, const int2* __restrict__ range, const int* __restrict__ keep,
                            const int* __restrict__ outidx, const FBox* __restrict__ nbox, const FBox* __restrict__ tbox,
                            const unsigned int* __restrict__ order, const int* __restrict__ parent_int, BvhNode* __restrict__ out,
                            int* __restrict__ max_depth) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n - 1 || !keep[i]) return;
    BvhNode nd;
    const int2 c = child[i];
    put_child(nd.lo0, nd.hi0, nd.c0, nd.n0, c.x, range, keep, outidx, nbox, tbox, order);
    put_child(nd.lo1, nd.hi1, nd.c1, nd.n1, c.y, range, keep, outidx, nbox, tbox, order);
    out[outidx[i]] = nd;
    int depth = 1;
    for (int p = parent_int[i]; p >= 0; p = parent_int[p]) ++depth;   // every ancestor of a kept node is kept
    atomicMax(max_depth, depth);
}

__global__ void k_lbvh_gather(int n, const unsigned int* __restrict__ order, const uint4* __restrict__ rec_in, uint4* __restrict__ rec_out,
                              const uint4* __restrict__ slab_in, uint4* __restrict__ slab_out) {
    const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;     // one 16-byte piece per thread
    if (t >= (long long)n * 8) return;
    const int p = (int)(t >> 3), piece = (int)(t & 7);
    const unsigned src = order[p];
    rec_out[(size_t)p * 8 + piece] = rec_in[(size_t)src * 8 + piece];
    if (piece < 4) slab_out[(size_t)p * 4 + piece] = slab_in[(size_t)src * 4 + piece];
}

struct Tmp {
    void* p = nullptr;
    ~Tmp() { if (p) (void)hipFree(p); }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, std::max<size_t>(bytes, 256)); }
    template <class T> T* as() { return (T*)p; }
};

}  // namespace

#define LB_HIP(call) do { hipError_t e__ = (call); if (e__ != hipSuccess) return e__; } while (0)

// The fp32 shaft / classification record of every triangle (TriSlab, sr_types.h), TriangleIndex order, for both the host and
// the device build: unit normal + offset, three in-plane unit edge normals pointing at the opposite vertex + offsets, computed
// in FP64 relative to the root centre and rounded once.  Degenerate / needle-thin triangles, and those whose "zero" normal the
// reference replaces by (1,0,0) (Triangle.cs:42-43: it then accepts hits in the plane x = v1.x that need not be near the
// geometric triangle), get an all-zero record = "never filtered", which is always admissible.
__global__ void k_make_slabs(const double* __restrict__ v9, int n, double cx, double cy, double cz, TriSlab* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double* p = v9 + (size_t)i * 9;
    struct V { double x, y, z; };
    auto sub = [](V a, V b) { return V{a.x - b.x, a.y - b.y, a.z - b.z}; };
    auto dot = [](V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
    auto cross = [](V a, V b) { return V{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
    auto scale = [](V a, double s) { return V{a.x * s, a.y * s, a.z * s}; };
    const V c = {cx, cy, cz};
    const V v1 = {p[0], p[1], p[2]}, v2 = {p[3], p[4], p[5]}, v3 = {p[6], p[7], p[8]};
    TriSlab t;
    t.n[0] = t.n[1] = t.n[2] = t.d = 0.0f;
    t.m1[0] = t.m1[1] = t.m1[2] = t.c1 = 0.0f; t.m2[0] = t.m2[1] = t.m2[2] = t.c2 = 0.0f; t.m3[0] = t.m3[1] = t.m3[2] = t.c3 = 0.0f;
    const TriSlab zero = t;
    bool ok = true;
    {   // the reference replaces a "zero" normal by (1,0,0) (Triangle.cs:42-43, Vector.cs:140): no planes for such triangles
        const V n0 = cross(sub(v2, v1), sub(v3, v1));
        const double e = 1e-10;
        if (-e < n0.x && n0.x < e && -e < n0.y && n0.y < e && -e < n0.z && n0.z < e) ok = false;
    }
    const V a = sub(v1, c), b = sub(v2, c), d3 = sub(v3, c);
    V nn = cross(sub(b, a), sub(d3, a));
    const double nl = sqrt(dot(nn, nn));
    const double e1 = sqrt(dot(sub(b, a), sub(b, a))), e2 = sqrt(dot(sub(d3, b), sub(d3, b))), e3 = sqrt(dot(sub(a, d3), sub(a, d3)));
    const double emax = fmax(e1, fmax(e2, e3));
    if (!(nl > 1e-12 * emax * emax) || !(emax > 0) || !isfinite(nl)) ok = false;
    if (ok) {
        nn = scale(nn, 1.0 / nl);
        const V P[3] = {a, b, d3};
        float* mm[3] = {t.m1, t.m2, t.m3};
        float* cc[3] = {&t.c1, &t.c2, &t.c3};
        for (int k = 0; k < 3; ++k) {
            const V p0 = P[k], p1 = P[(k + 1) % 3], p2 = P[(k + 2) % 3];
            V m = cross(nn, sub(p1, p0));
            const double ml = sqrt(dot(m, m));
            if (!(ml > 0)) { ok = false; break; }
            m = scale(m, 1.0 / ml);
            if (dot(m, sub(p2, p0)) < 0) m = scale(m, -1.0);
            mm[k][0] = (float)m.x; mm[k][1] = (float)m.y; mm[k][2] = (float)m.z;
            *cc[k] = (float)dot(m, p0);
        }
        t.n[0] = (float)nn.x; t.n[1] = (float)nn.y; t.n[2] = (float)nn.z;
        t.d = (float)dot(nn, a);
    }
    out[i] = ok ? t : zero;
}

// ---- the four-wide tree of the packet walks, collapsed on the device (same rule as sr_host.cpp collapse_bvh4: a node takes its two
//      children and, while it has fewer than four, replaces the inner child with the largest box by that child's two children) ----
// One launch per level of the wide tree: every item (binary node, wide-node slot) writes its wide node and appends its inner
// children to the next level's list (slots handed out by an atomic counter: the numbering depends on the schedule, the tree does not).
__global__ void k_collapse_level(const BvhNode* __restrict__ nodes, const int2* __restrict__ in, int n_in, int2* __restrict__ out,
                                 int* __restrict__ counters /* [0] next free wide node, [1] items appended */, Bvh4Node* __restrict__ wide) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in) return;
    const int2 item = in[i];
    Bvh4Child ch[4];
    int k = 0;
    const auto child_of = [](const BvhNode& n, int side) {
        Bvh4Child c;
        for (int a = 0; a < 3; ++a) { c.lo[a] = side ? n.lo1[a] : n.lo0[a]; c.hi[a] = side ? n.hi1[a] : n.hi0[a]; }
        c.c = side ? n.c1 : n.c0;
        c.n = side ? n.n1 : n.n0;
        return c;
    };
    const auto area = [](const Bvh4Child& c) {
        const float dx = c.hi[0] - c.lo[0], dy = c.hi[1] - c.lo[1], dz = c.hi[2] - c.lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    {
        const BvhNode r = nodes[item.x];
        const Bvh4Child a = child_of(r, 0), b = child_of(r, 1);
        if (a.n >= 0) ch[k++] = a;
        if (b.n >= 0) ch[k++] = b;
    }
    while (k < 4) {
        int best = -1;
        float best_area = -1.0f;
        for (int j = 0; j < k; ++j)
            if (ch[j].n == 0) { const float a = area(ch[j]); if (a > best_area) { best_area = a; best = j; } }
        if (best < 0) break;
        const BvhNode m = nodes[ch[best].c];
        for (int j = k; j > best + 1; --j) ch[j] = ch[j - 1];
        ch[best] = child_of(m, 0);
        ch[best + 1] = child_of(m, 1);
        ++k;
    }
    for (int j = 0; j < k; ++j) {
        if (ch[j].n == 0) {
            const int dst = atomicAdd(&counters[0], 1);
            out[atomicAdd(&counters[1], 1)] = make_int2(ch[j].c, dst);
            ch[j].c = dst;
        }
    }
    for (int j = k; j < 4; ++j) {
        for (int a = 0; a < 3; ++a) { ch[j].lo[a] = 1.0f; ch[j].hi[a] = -1.0f; }
        ch[j].c = 0; ch[j].n = -1;
    }
    Bvh4Node w;
    for (int j = 0; j < 4; ++j) w.ch[j] = ch[j];
    wide[item.y] = w;
}

hipError_t collapse_bvh4_device(const BvhNode* d_nodes, int num_nodes, Bvh4Node* d_wide, int* num_wide, int* depth, hipStream_t stream) {
    *num_wide = 0; *depth = 0;
    if (num_nodes <= 0) return hipSuccess;
    Tmp la, lb, ctr;
    LB_HIP(la.alloc((size_t)num_nodes * sizeof(int2)));
    LB_HIP(lb.alloc((size_t)num_nodes * sizeof(int2)));
    LB_HIP(ctr.alloc(8));
    const int2 root = make_int2(0, 0);
    int init[2] = {1, 0};
    LB_HIP(hipMemcpyAsync(la.p, &root, sizeof(root), hipMemcpyHostToDevice, stream));
    LB_HIP(hipMemcpyAsync(ctr.p, init, 8, hipMemcpyHostToDevice, stream));
    int n_in = 1, levels = 0;
    int2* in = la.as<int2>();
    int2* out = lb.as<int2>();
    while (n_in > 0) {
        ++levels;
        if (levels > 256) return hipErrorUnknown;                 // (a tree deeper than that is refused by sr_build anyway)
        hipLaunchKernelGGL(k_collapse_level, dim3((unsigned)((n_in + 255) / 256)), dim3(256), 0, stream, d_nodes, (const int2*)in, n_in, out, ctr.as<int>(), d_wide);
        LB_HIP(hipGetLastError());
        int c[2];
        LB_HIP(hipMemcpyAsync(c, ctr.p, 8, hipMemcpyDeviceToHost, stream));
        LB_HIP(hipStreamSynchronize(stream));
        n_in = c[1];
        *num_wide = c[0];
        LB_HIP(hipMemsetAsync(ctr.as<int>() + 1, 0, 4, stream));
        std::swap(in, out);
    }
    *depth = levels;
    return hipSuccess;
}

hipError_t make_slabs_device(const double* d_v9, int n, const RootBox& root, TriSlab* d_out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_make_slabs, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_v9, n, root.centre[0], root.centre[1], root.centre[2], d_out);
    return hipGetLastError();
}

// Leaf-order copies of the FP64 records and the fp32 shaft records: out[p] = in[order[p]] (the host SAH build ships only its
// order; gathering 128 + 64 bytes per triangle on the host and uploading the copies cost more than the build itself).
hipError_t gather_records_device(int n, const unsigned int* d_order, const Rec128* d_tris, Rec128* d_btris, const TriSlab* d_slab_in,
                                 TriSlab* d_bslab, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    const long long pieces = (long long)n * 8;
    hipLaunchKernelGGL(k_lbvh_gather, dim3((unsigned)((pieces + 255) / 256)), dim3(256), 0, stream, n, d_order, (const uint4*)d_tris, (uint4*)d_btris,
                       (const uint4*)d_slab_in, (uint4*)d_bslab);
    return hipGetLastError();
}

// d_v9: device double[n][9]; d_tris / d_slab_in: records in TriangleIndex order; outputs are caller-allocated:
// d_nodes (>= n entries), d_btris (n), d_bslab (n).  Returns the node count and depth.
hipError_t build_bvh_device(const double* d_v9, int n, const RootBox& root, const Rec128* d_tris, const TriSlab* d_slab_in,
                            BvhNode* d_nodes, Rec128* d_btris, TriSlab* d_bslab, int* num_nodes, int* depth, hipStream_t stream, int leaf_max) {
    const int kLeafMax = leaf_max > 0 ? std::min(leaf_max, 15) : kLeafMaxDefault;
    if (n <= kLeafMax * 2) return hipErrorInvalidValue;              // tiny scenes use the host builder
    double ext = 0;
    for (int a = 0; a < 3; ++a) ext = std::max(ext, root.max[a] - root.min[a]);
    const float pad = (float)std::ldexp(ext > 0 ? ext : 1.0, -16);
    Tmp tbox, keys, keys2, vals, vals2, sorttmp, child, range, pint, pleaf, flags, nbox, keep, outidx, scantmp, meta;
    LB_HIP(tbox.alloc((size_t)n * sizeof(FBox)));
    LB_HIP(keys.alloc((size_t)n * 8)); LB_HIP(keys2.alloc((size_t)n * 8));
    LB_HIP(vals.alloc((size_t)n * 4)); LB_HIP(vals2.alloc((size_t)n * 4));
    LB_HIP(child.alloc((size_t)n * sizeof(int2))); LB_HIP(range.alloc((size_t)n * sizeof(int2)));
    LB_HIP(pint.alloc((size_t)n * 4)); LB_HIP(pleaf.alloc((size_t)n * 4));
    LB_HIP(flags.alloc((size_t)n * 4)); LB_HIP(nbox.alloc((size_t)n * sizeof(FBox)));
    LB_HIP(keep.alloc((size_t)n * 4)); LB_HIP(outidx.alloc((size_t)n * 4));
    LB_HIP(meta.alloc(16));
    const int T = 256, B = (n + T - 1) / T;
    hipLaunchKernelGGL(k_lbvh_keys, dim3(B), dim3(T), 0, stream, d_v9, n, root, pad, tbox.as<FBox>(), keys.as<unsigned long long>(), vals.as<unsigned int>());
    LB_HIP(hipGetLastError());
    size_t sbytes = 0;
    LB_HIP(hipcub::DeviceRadixSort::SortPairs(nullptr, sbytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(), vals.as<unsigned int>(),
                                              vals2.as<unsigned int>(), n, 0, 63, stream));
    LB_HIP(sorttmp.alloc(sbytes));
    LB_HIP(hipcub::DeviceRadixSort::SortPairs(sorttmp.p, sbytes, keys.as<unsigned long long>(), keys2.as<unsigned long long>(), vals.as<unsigned int>(),
                                              vals2.as<unsigned int>(), n, 0, 63, stream));
    const unsigned long long* skeys = keys2.as<unsigned long long>();
    const unsigned int* order = vals2.as<unsigned int>();
    hipLaunchKernelGGL(k_lbvh_tree, dim3(B), dim3(T), 0, stream, skeys, n, child.as<int2>(), range.as<int2>(), pint.as<int>(), pleaf.as<int>());
    LB_HIP(hipGetLastError());
    LB_HIP(hipMemsetAsync(flags.p, 0, (size_t)n * 4, stream));
    hipLaunchKernelGGL(k_lbvh_boxes, dim3(B), dim3(T), 0, stream, n, order, tbox.as<FBox>(), child.as<int2>(), pint.as<int>(), pleaf.as<int>(),
                       flags.as<unsigned int>(), nbox.as<FBox>());
    LB_HIP(hipGetLastError());
    hipLaunchKernelGGL(k_lbvh_mark, dim3(B), dim3(T), 0, stream, n, range.as<int2>(), keep.as<int>(), kLeafMax);
    LB_HIP(hipGetLastError());
    size_t cbytes = 0;
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, cbytes, keep.as<int>(), outidx.as<int>(), n - 1, stream));
    LB_HIP(scantmp.alloc(cbytes));
    LB_HIP(hipcub::DeviceScan::ExclusiveSum(scantmp.p, cbytes, keep.as<int>(), outidx.as<int>(), n - 1, stream));
    LB_HIP(hipMemsetAsync(meta.p, 0, 16, stream));
    hipLaunchKernelGGL(k_lbvh_emit, dim3(B), dim3(T), 0, stream, n, child.as<int2>(), range.as<int2>(), keep.as<int>(), outidx.as<int>(), nbox.as<FBox>(),
                       tbox.as<FBox>(), order, pint.as<int>(), d_nodes, meta.as<int>());
    LB_HIP(hipGetLastError());
    const long long pieces = (long long)n * 8;
    hipLaunchKernelGGL(k_lbvh_gather, dim3((unsigned)((pieces + T - 1) / T)), dim3(T), 0, stream, n, order, (const uint4*)d_tris, (uint4*)d_btris,
                       (const uint4*)d_slab_in, (uint4*)d_bslab);
    LB_HIP(hipGetLastError());
    // node count = keep[n-2] + outidx[n-2]; depth from the emit kernel
    int last_keep = 0, last_idx = 0, md = 0;
    LB_HIP(hipMemcpyAsync(&last_keep, keep.as<int>() + (n - 2), 4, hipMemcpyDeviceToHost, stream));
    LB_HIP(hipMemcpyAsync(&last_idx, outidx.as<int>() + (n - 2), 4, hipMemcpyDeviceToHost, stream));
    LB_HIP(hipMemcpyAsync(&md, meta.p, 4, hipMemcpyDeviceToHost, stream));
    LB_HIP(hipStreamSynchronize(stream));
    *num_nodes = last_keep + last_idx;
    *depth = md + 1;                                                 // + the leaf level
    return hipSuccess;
}

}  // namespace sr
