// sr_pipeline.hip -- the frame as a wavefront pipeline (the default path of sr_render):
//
//   k_primary  one lane per pixel: camera ray(s) -> nearest hit -> ShadingMethod.  Misses and shadow-less
//              frames are final here.  With shadows on, the lanes that hit are compacted with
//              __ballot/__popcll into a dense hit queue in HBM (one atomicAdd per wavefront).
//   k_shadow   persistent wavefronts, one lane per queued hit: ShadowMethod's area-light samples with an
//              any-hit traversal and a per-lane blocker cache; a lane that finishes its hit immediately
//              pulls the next one from the queue (wave-aggregated atomic), so the 64 lanes stay busy even
//              though fully-shadowed hits cost ~1 traversal and fully-lit hits cost 100.
//   k_resolve  rayTraceSubPixelRes > 1 only: integer byte sums, truncating divide (Renderer.cs:1815-1826).
//
// Results are identical to the one-kernel renderer (k_render, sr_kernels.hip) by construction: the same
// device functions of sr_trace.h do all result-affecting arithmetic.
#include "sr_trace.h"

#include <algorithm>
#include <type_traits>

namespace sr {

extern __shared__ __attribute__((aligned(16))) unsigned char lds_pipe[];

// one queued surface point (64 B): what ShadowMethod.IntersectRay needs from the primary hit
struct alignas(16) HitRec {
    double   pos[3];
    double   nrm[3];
    uint32_t sample;     // index into the sample-colour buffer
    uint32_t pad[3];     // pad[0]: cell of the static shadow cache (generator hits of a static frame)
};
static_assert(sizeof(HitRec) == 64, "HitRec must be 64 bytes");
constexpr uint32_t kInvalidHit = 0xffffffffu;      // HitRec.sample of a padding entry of the tile-aligned hit queue

// ShadowMethod.IntersectRay's last step for one surface point (ShadowMethod.cs:103-119): dynamic shadows modulate the
// shaded colour with (byte)(fraction * 255); a static frame (SR_F_STATIC_SHADOWS) runs the shadow kernels only on the
// hit points that generate a cache cell and stores (byte)(fraction * 254 + 1) there (ShadowMethod.cs:80, 0 = empty cell;
// Texture3DCache.cs:124-126 replaces a generated 0 by 1) -- k_static_apply modulates every hit point afterwards.
__device__ __forceinline__ void finish_hit(const DevScene& sc, const FrameConst& fc, uint32_t* samples, uint32_t sample, uint32_t cell,
                                           uint32_t shaded, double frac) {
    if (fc.flags & 32u) {
        uint32_t v = (uint32_t)(int)(frac * 254 + 1) & 0xffu;
        sc.shadow_cache[cell] = (uint8_t)(v ? v : 1u);
    } else if (fc.accum) {
        // one chunk of a > 128-sample frame: frac = escapes / samples of THIS chunk, both <= 128, so the count comes back exactly
        fc.accum[sample] += (uint32_t)(frac * (double)fc.shadow_samples + 0.5);
    } else {
        samples[sample] = modulate(shaded, to_byte(frac * 255));
    }
}

// the last step of a chunked (> 128 samples) shadow stage: rayEscapeCount summed over the chunks -> (byte)(count / total * 255)
// (ShadowMethod.cs:113-119) for every hit point of the band's queue; the sums are cleared for the next frame
__global__ __launch_bounds__(256) void k_accum_finish(FrameConst fc, const HitRec* __restrict__ hits, const unsigned int* __restrict__ count,
                                                      uint32_t* __restrict__ samples, int total_samples) {
    const unsigned int total = *count, stride = gridDim.x * 256u;
    for (unsigned int h = blockIdx.x * 256u + threadIdx.x; h < total; h += stride) {
        const uint32_t sample = hits[h].sample;
        if (sample == kInvalidHit) continue;
        const uint32_t esc = fc.accum[sample];
        fc.accum[sample] = 0u;
        const double frac = (double)esc / (double)total_samples;
        samples[sample] = modulate(samples[sample], to_byte(frac * 255));
    }
}

__device__ __forceinline__ unsigned long long lanemask_lt() {
    unsigned lane = threadIdx.x & 63u;
    return lane == 0 ? 0ull : (~0ull >> (64u - lane));
}

// XCD-aware tile order.  Workgroups are dealt round-robin over the 8 XCDs (each with its own L2), so workgroup L runs on
// XCD L % 8; the 1-D grid is mapped so that every XCD walks through its own 8x8-tile super-tiles (128 x 128 pixels,
// neighbours share BVH nodes and triangle records in that XCD's L2), while at any time the eight XCDs work on different
// super-tile columns of the same super-tile row (skewed by the row: balanced load).  Returns false for the padding of the
// super-tile grid.  k_primary and k_shaft_pkt use the SAME mapping and the tile-aligned hit queue is indexed by the tile, so
// the shaft walk of a tile runs on the XCD whose L2 the tile's primary walk has just warmed.
__device__ __forceinline__ bool xcd_tile(int L, int width, int row_count, int& tile_x, int& tile_y) {
    const int tiles_x = (width + 15) >> 4, tiles_y = (row_count + 15) >> 4;
    const int spx = (((tiles_x + 7) >> 3) + 7) >> 3;                     // super-tile columns per XCD
    const int k = L & 7, j = L >> 3;
    const int sj = j >> 6, t = j & 63;
    const int sy = sj / spx, sx = (sj - sy * spx) * 8 + ((k + sy) & 7);
    tile_x = sx * 8 + (t & 7);
    tile_y = sy * 8 + (t >> 3);
    return tile_x < tiles_x && tile_y < tiles_y;
}
static int xcd_tile_grid(int width, int row_count) {
    const int tiles_x = (width + 15) / 16, tiles_y = (row_count + 15) / 16;
    const int spx = (((tiles_x + 7) / 8) + 7) / 8, sny = (tiles_y + 7) / 8;
    return spx * 8 * sny * 64;
}

// A kernel argument read WHERE IT IS NEEDED, from the kernel-argument segment (byte offset `offset`), through a pointer the
// compiler cannot see through: it loads every argument it can name at the top of the kernel and keeps it in scalar registers
// for good -- the root box alone is 48 of them, which a rare FP64 path needs and the loop around it then pays for with
// v_readlane / v_writelane spills.
template <typename T>
__device__ __forceinline__ T kernarg_late(size_t offset) {
    static_assert(sizeof(T) % 4 == 0 && alignof(T) >= 4, "whole 32-bit words");
    typedef const uint32_t __attribute__((address_space(4))) CW4;
    CW4* kp = (CW4*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    T out;
    uint32_t* o = reinterpret_cast<uint32_t*>(&out);
#pragma unroll
    for (size_t i = 0; i < sizeof(T) / 4; ++i) o[i] = kp[offset / 4 + i];
    return out;
}

// the first two arguments of the pipeline kernels as the kernel-argument segment holds them
struct SceneFrameArgs { DevScene sc; FrameConst fc; };

// Work distribution of the tile kernels (k_primary, k_shaft_pkt4).  Virtual block vb of the XCD-aware tile grid is one 16x16-pixel
// tile, its four waves vq = 0..3 are the tile's 8x8-pixel quadrants.  Direct mode (heads == nullptr): workgroup = virtual block,
// one item per wave.  Persistent mode: a fixed grid of resident workgroups whose WAVES pull (vb, vq) items from eight counters, one
// per XCD, each enumerating that XCD's virtual blocks (vb % 8 == xcd: the blocks the dispatcher would have dealt to it) in
// ascending order -- the order, and the XCD, of direct mode, so the XCD-local L2 sees the same tiles -- and, when its own XCD's
// list is used up, from the other XCDs' (load balance at the end of the launch).  Counters live 64 bytes apart, are zeroed by the
// host before the launch and may overshoot.  Workgroups are dealt to XCDs round-robin (blockIdx.x % 8).
constexpr int kTileHeadStride = 16;      // uints between two XCDs' counters
template <bool PERSIST>
struct TileFeed {
    unsigned int* heads;
    const unsigned int* order;       // (persistent) per-XCD lists of natural item ids, or nullptr: natural order
    unsigned int per_xcd, home, visit;
    unsigned int item;               // x * per_xcd + natural id of the item handed out last (index of its cost record)
    int wave, lane;
    bool direct_done;
    __device__ __forceinline__ TileFeed(unsigned int* h, unsigned int virtual_blocks, int w, int l, const unsigned int* ord = nullptr)
        : heads(h), order(ord), per_xcd((virtual_blocks >> 3) * 4u), home(blockIdx.x & 7u), visit(0u), item(0u), wave(w), lane(l), direct_done(false) {}
    __device__ __forceinline__ bool next(unsigned int& vb, unsigned int& vq) {
        if (!PERSIST) {
            if (direct_done) return false;
            direct_done = true;
            vb = blockIdx.x; vq = (unsigned)wave;
            return true;
        }
        while (visit < 8u) {
            const unsigned int x = (home + visit) & 7u;
            unsigned int t = 0;
            if (lane == 0) t = atomicAdd(&heads[x * kTileHeadStride], 1u);
            t = (unsigned int)__builtin_amdgcn_readfirstlane((int)t);
            if (t < per_xcd) {
                if (order) t = order[x * per_xcd + t];                       // (wave-uniform address)
                item = x * per_xcd + t;
                vb = (t >> 2) * 8u + x; vq = t & 3u;
                return true;
            }
            ++visit;
        }
        return false;
    }
};
// scene / frame of a tile kernel's iteration: the plain kernel arguments in direct mode (one iteration: nothing to hoist), read late
// from the kernel-argument segment in persistent mode (see k_shaft_pkt4)
template <bool PERSIST, class T>
__device__ __forceinline__ T tile_arg(const T& plain, size_t offset) {
    if constexpr (PERSIST) return kernarg_late<T>(offset);
    else return plain;
}

// --------------------------------------------------------------------------------------------------
// k_primary
// --------------------------------------------------------------------------------------------------
// SUB = false: one ray per pixel (rayTraceSubPixelRes == 1), compiled without the sub-pixel / focal-blur state
// PKT (own BVH, all rays of the frame share one origin): the wave walks the tree once (bvh_packet_nearest, sr_trace.h) and
// consults the frame's camera-cone records before the FP64 triangle test
// PKT = 2: the packet walk on the four-wide tree's camera-ordered copy (bvh4_packet_nearest), the default; PKT = 3: the same with the
// camera outside the root box's slab on all three axes (the copy holds (near, far) planes: no min / max per axis); PKT = 1: on the
// binary tree with a per-step vote (cross-check)
// (the packet walk on the four-wide tree at 7 waves/SIMD, 72 VGPRs: the kernel waits for its scalar loads -- 68 % VALU busy -- and a seventh wave per SIMD hides more
//  of them than the handful of spilled registers costs: 2.30 -> 2.25 ms; 8 waves, 64 VGPRs: 2.89)
template <int MODE, bool EXTRA, bool STATS, bool SUB, int PKT, bool PERSIST>
__global__ __launch_bounds__(256, SUB ? 5 : ((PKT >= 2 && !PERSIST) ? 7 : 6)) void k_primary(DevScene sc_arg, FrameConst fc_arg, const int32_t* __restrict__ row_map, int row_begin,
                                                 int row_count, uint32_t* __restrict__ samples, HitRec* __restrict__ hits,
                                                 unsigned int* __restrict__ hit_count, uint32_t* __restrict__ bounce_levels,
                                                 uint8_t* __restrict__ bounce_nlev, unsigned long long* stats, int pad_tiles, int levels,
                                                 unsigned int* __restrict__ tile_heads, unsigned int virtual_blocks) {
    const int tid = threadIdx.x;
    const int pwave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    int32_t* wnode = reinterpret_cast<int32_t*>(lds_pipe) + (size_t)pwave * levels;   // PKT: the wave's node stack
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    Ctr prim = {0, 0, 0, 0};
    // (work items: see TileFeed -- direct mode = one workgroup per 16x16 tile, persistent mode = waves pulling 8x8 quadrants)
    TileFeed<PERSIST> feed(tile_heads, virtual_blocks, pwave, lane);
    unsigned int vb, vq;
    while (feed.next(vb, vq)) {
    // Scene and frame are read from the kernel-argument segment WHERE an iteration needs them (kernarg_late): the ray set-up's inputs
    // here, the walk's before the walk, the shading's after it.  Read as plain kernel arguments, the compiler hoists them -- and what it
    // derives from them -- out of this loop and keeps it all live across the walk (133 spilled VGPRs); read in one piece at the top
    // of the loop they stay live from there to their last use.
    const FrameConst fc = tile_arg<PERSIST>(fc_arg, offsetof(SceneFrameArgs, fc));  // (ray set-up)
    const int wave = (int)vq;                                              // the tile quadrant this wave traces now
    int tile_x, tile_y;
    if (!xcd_tile((int)vb, fc.width, row_count, tile_x, tile_y)) continue;  // padding of the super-tile grid
    const int col = tile_x * 16 + (wave & 1) * 8 + (lane & 7);
    const int brow = tile_y * 16 + (wave >> 1) * 8 + (lane >> 3);          // row inside this band
    const bool live = col < fc.width && brow < row_count;
    const int crow = row_begin + brow;                                       // compact row of the frame
    const int row = live ? row_map[crow] : 0;
    const int n = SUB ? fc.sub_pixel_res : 1, n2 = n * n;
    const bool shadows = (fc.flags & 2u) != 0;
    const bool bounce = fc.max_bounces > 0;          // pipeline form of the mirror extension: never together with shadows
    const int width = fc.width, height = fc.height;
    // sample buffer: n == 1 -> the frame itself (final pixel position); n > 1 -> band-local [brow][col][n2]
    const int out_row = (fc.strip_count > 0) ? crow : row;
    const size_t sbase = (n == 1) ? ((size_t)out_row * width + col) : (((size_t)brow * width + col) * n2);
    const D3 start = mk(fc.start_world[0], fc.start_world[1], fc.start_world[2]);
    const bool blur = (fc.flags & 4u) != 0;
    D3 focal = mk(0, 0, 0);
    if (n > 1 && blur) {
        D3 dv = mk(-((double)col / width - 0.5), -((double)row / height - 0.5) * fc.aspect, fc.fov_depth);
        focal = mul3x3(fc.it, dv) * fc.focal_depth + start;
    }
    for (int si = 0; si < n2; ++si) {                                        // subX outer, subY inner (:1761-1763)
        const int sx = si / n, sy = si - sx * n;
        D3 ss = start, dw;
        if (n == 1) {                                                        // fast path, Renderer.cs:1722-1743
            D3 dv = mk(-((double)col / width - 0.5), -((double)row / height - 0.5) * fc.aspect, fc.fov_depth);
            dw = mul3x3(fc.it, dv);
        } else {
            double fx = (double)sx / (n - 1) - 0.5;
            double fy = (double)sy / (n - 1) - 0.5;
            if (blur) {
                D3 sv = mk(fx / width * fc.focal_blur_strength, fy / height * fc.focal_blur_strength, -fc.position_z);
                ss = mul3x3(fc.it, sv);
                dw = focal - ss;
            } else {
                D3 dv = mk(-((col + fx) / width - 0.5), -((row + fy) / height - 0.5) * fc.aspect, fc.fov_depth);
                dw = mul3x3(fc.it, dv);
            }
        }
        Hit h;
        bool ok = false;
        {
            const DevScene sc = tile_arg<PERSIST>(sc_arg, offsetof(SceneFrameArgs, sc));   // (the walk)
            if (PKT) {
                if (live) prim.rays++;
                ok = root_intersect_pkt<EXTRA, true, PKT == 3 ? 2 : (PKT == 2 ? 1 : 0)>(sc, sc.extra, wnode, live, ss, dw, h, prim);     // all 64 lanes take part
            } else if (live) {
                prim.rays++;
                ok = root_intersect<MODE, false, EXTRA>(sc, sc.tris, sc.extra, st, ss, dw, h, prim);
            }
        }
        const FrameConst fc = tile_arg<PERSIST>(fc_arg, offsetof(SceneFrameArgs, fc));   // (shading, queue)
        if (live) {
            uint32_t color = fc.background;
            if (ok) color = (fc.flags & 1u) ? shade(fc, h.pos, h.nrm, h.color) : h.color;
            // ShadowMethod with rayEscapeCount == softShadowQuality for every hit point (proven on the host): (byte)(1.0 * 255)
            if (ok && (fc.flags & kFlagAllSamplesEscape)) color = modulate(color, to_byte((double)fc.shadow_samples / (double)fc.shadow_samples * 255));
            samples[sbase + si] = color;
            if (bounce) {                                                    // level 0 of the mirror chain (see k_bounce)
                bounce_nlev[sbase + si] = ok ? 1 : 0;
                if (ok) bounce_levels[(sbase + si) * (size_t)(fc.max_bounces + 1)] = color;
            }
        }
        if (shadows || bounce) {
            HitRec r;
            if (ok) {
                r.pos[0] = h.pos.x; r.pos[1] = h.pos.y; r.pos[2] = h.pos.z;
                r.nrm[0] = h.nrm.x; r.nrm[1] = h.nrm.y; r.nrm[2] = h.nrm.z;
                r.sample = (uint32_t)(sbase + si);
                r.pad[0] = r.pad[1] = r.pad[2] = 0;
                if (bounce) {                                                // the queue holds the next RAY: origin, direction, level
                    const D3 refl = dw - h.nrm * (2.0 * dot(dw, h.nrm));
                    const D3 org = h.pos + h.nrm * 0.001;
                    r.pos[0] = org.x; r.pos[1] = org.y; r.pos[2] = org.z;
                    r.nrm[0] = refl.x; r.nrm[1] = refl.y; r.nrm[2] = refl.z;
                    r.pad[0] = 1u;
                }
            }
            if (pad_tiles) {
                // shaft path: the queue is indexed by the TILE -- entry ((tile * n2 + sub-sample) * 4 + wave) * 64 + lane, lanes
                // without a hit mark theirs invalid -- so a wave of k_shaft_pkt owns exactly one 8x8-pixel tile of surface
                // points, no atomic is needed, and block b of k_shaft_pkt reads what block b of this kernel wrote
                const int tiles_x = (fc.width + 15) >> 4;
                const unsigned int slot = (((unsigned)(tile_y * tiles_x + tile_x) * (unsigned)n2 + (unsigned)si) * 4u + (unsigned)wave) * 64u + (unsigned)lane;
                if (ok) hits[slot] = r;
                else hits[slot].sample = kInvalidHit;
            } else {                                                         // active-ray compaction
                const unsigned long long m = __ballot(ok);
                if (m) {
                    unsigned int base = 0;
                    const int leader = __ffsll((long long)m) - 1;
                    if (lane == leader) base = atomicAdd(hit_count, (unsigned int)__popcll(m));
                    base = __shfl(base, leader, 64);
                    if (ok) hits[base + (unsigned int)__popcll(m & lanemask_lt())] = r;
                }
            }
        }
    }
    }   // while (feed.next)
    if (STATS) {
        uint32_t a = wave_sum(prim.rays), b = wave_sum(prim.geom), c2 = wave_sum(prim.nodes), d2 = wave_sum(prim.leaves);
        block_stat_add(&stats[0], &stats[1], &stats[2], &stats[3], a, b, c2, d2);
    }
}

// --------------------------------------------------------------------------------------------------
// k_shadow
// --------------------------------------------------------------------------------------------------
// one area-light sample: is the surface point occluded?  (ShadowMethod.cs:147-177)
template <int MODE, bool EXTRA>
__device__ __forceinline__ bool sample_blocked(const DevScene& sc, const FrameConst& fc, Stack st, D3 rs, D3 rd, int32_t& cache, Ctr& c) {
    Hit h;
    h.tri = -1;
    bool blocked;
    if (MODE == MODE_REF) {
        // the reference tree returns ITS nearest hit; only that hit's rayFrac is compared with 1.0
        blocked = root_intersect<MODE, false, EXTRA>(sc, sc.tris, sc.extra, st, rs, rd, h, c) && !(h.t > 1.0);
    } else {
        blocked = root_intersect<MODE, true, EXTRA>(sc, sc.tris, sc.extra, st, rs, rd, h, c) && !(h.t > 1.0);
        if (blocked && h.tri >= 0) cache = h.tri;
    }
    return blocked;
}

template <int MODE, bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_shadow(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                const HitRec* __restrict__ hits, const unsigned int* __restrict__ hit_count,
                                                unsigned int* __restrict__ work_head, const unsigned int* __restrict__ index_list,
                                                uint32_t* __restrict__ samples, int stack_levels, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    // the area-light offset table staged in LDS behind the stacks (ShadowMethod.cs:63-73: 100 x 3 doubles)
    double* loff = reinterpret_cast<double*>(lds_pipe + (((size_t)stack_levels * 256 * 4 + 15) & ~(size_t)15));
    const int S = fc.shadow_samples;
    for (int i = tid; i < 3 * S; i += 256) loff[i] = offsets[i];
    __syncthreads();

    const unsigned int total = *hit_count;
    const bool point = (fc.flags & 8u) != 0;
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const D3 ldir = mk(fc.light_dir_model[0], fc.light_dir_model[1], fc.light_dir_model[2]);

    Ctr sec = {0, 0, 0, 0};
    int i = S;                       // next sample of the current hit; S = lane needs a new hit
    int escapes = 0;
    int32_t cache = -1;              // record position of the last occluder found for this lane
    uint32_t sample = 0, cell = 0;
    D3 shadowEnd = mk(0, 0, 0);
    bool exhausted = false;

    for (;;) {
        // ---- refill: lanes without work pull the next hits (wave-aggregated atomic) ----
        const bool need = (i >= S) && !exhausted;
        const unsigned long long m = __ballot(need);
        if (m) {
            unsigned int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(work_head, (unsigned int)__popcll(m));
            base = __shfl(base, leader, 64);
            if (need) {
                const unsigned int idx = base + (unsigned int)__popcll(m & lanemask_lt());
                if (idx < total) {
                    const HitRec r = hits[index_list ? index_list[idx] : idx];
                    D3 pos = mk(r.pos[0], r.pos[1], r.pos[2]), nrm = mk(r.nrm[0], r.nrm[1], r.nrm[2]);
                    shadowEnd = pos + nrm * 0.001;                       // shadowProbeOffset, ShadowMethod.cs:10,151
                    sample = r.sample;
                    cell = r.pad[0];
                    i = 0;
                    escapes = 0;
                    // the cache is deliberately kept across hits: neighbouring surface points share occluders
                } else {
                    exhausted = true;
                }
            }
        }
        if (!__any(i < S)) break;

        if (i < S) {
            // ---- cheap phase: samples that the cached occluder blocks cost one triangle test each ----
            if (MODE != MODE_REF) {
                while (i < S && cache >= 0) {
                    D3 off = mk(loff[3 * i], loff[3 * i + 1], loff[3 * i + 2]);
                    D3 rs, rd;
                    if (point) { rs = lpos + off; rd = shadowEnd - rs; }
                    else { rd = ldir; rs = shadowEnd + rd * 1000.0 + off; }
                    bool hitc = (MODE == MODE_BVH) ? bvh_cached_blocks(sc, cache, rs, rd) : brute_cached_blocks(sc.tris, cache, rs, rd);
                    sec.geom++;
                    if (!hitc) break;
                    sec.rays++;
                    ++i;
                }
            }
            // ---- one full any-hit traversal for the lanes whose current sample is still undecided ----
            if (i < S) {
                D3 off = mk(loff[3 * i], loff[3 * i + 1], loff[3 * i + 2]);
                D3 rs, rd;
                if (point) { rs = lpos + off; rd = shadowEnd - rs; }
                else { rd = ldir; rs = shadowEnd + rd * 1000.0 + off; }
                sec.rays++;
                if (!sample_blocked<MODE, EXTRA>(sc, fc, st, rs, rd, cache, sec)) escapes++;
                ++i;
            }
            if (i >= S) {                                                 // ShadowMethod.IntersectRay :113-119
                double frac = (double)escapes / (double)S;
                finish_hit(sc, fc, samples, sample, cell, (fc.flags & 32u) ? 0u : samples[sample], frac);
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom), c2 = wave_sum(sec.nodes), d2 = wave_sum(sec.leaves);
        if (lane == 0) {
            stat_add(&stats[4], a);
            stat_add(&stats[5], b);
            stat_add(&stats[6], c2);
            stat_add(&stats[7], d2);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Shaft shadows (own BVH + point light): k_shaft + k_shadow_test.
//
// All S sample rays of one surface point end in the same point E' and start on a sphere of radius R
// around the light: P_i(t) = C(t) + (1 - t) * off_i with the centre ray C(t) = (1-t) L + t E'.  A
// triangle can be hit by ANY of them at parameter t only if its box is within (1-t) R of C(t).
//
//   k_shaft        lane = hit point.  Walks the BVH ONCE with that shaft (fp32, conservative, children
//                  nearest to E' first) and writes the leaves it touches -- (first record, count) packed in
//                  one int -- to the hit's candidate list in HBM, at most kShaftCap leaves (then the list is
//                  flagged "truncated").  64 independent walks per wave hide the node-fetch latency.
//   k_shadow_test  wave = hit point, lanes = area-light samples.  Stages the candidates' 128-byte records
//                  through LDS (8 leaves = up to 32 records per pass, 16 B per lane, coalesced) and every
//                  lane tests its own 1-2 sample rays against them with the reference's exact FP64
//                  arithmetic.  Blocked samples drop out; when none is left the hit is finished.  If a
//                  list is exhausted the remaining samples have provably no occluder (ShadowMethod.cs:170).
//                  Only "truncated list AND a sample still undecided" goes to the fallback queue, which
//                  k_shadow (one lane per hit, full any-hit traversals) finishes.
// The result is exactly that of S independent any-hit searches.
// --------------------------------------------------------------------------------------------------
struct SampleRay {
    D3     s, d;        // clipped start (SpatialSubdivision.cs:394) and direction
    double offset;      // rayFracOffset (:401)
};

__device__ __forceinline__ bool prepare_sample(const RootBox& root, D3 rs, D3 rd, SampleRay& r) {
    D3 end = rs + rd * 10000.0;
    r.s = rs;
    r.d = rd;
    if (!clip_segment<false>(root, r.s, end)) return false;
    r.offset = length(rs - r.s) / length(rd);
    return true;
}

__device__ __forceinline__ bool prepare_sample(const DevScene& sc, D3 rs, D3 rd, SampleRay& r) { return prepare_sample(sc.root, rs, rd, r); }


template <bool EXTRA>
__device__ __forceinline__ bool extras_block(const DevScene& sc, D3 rs, D3 rd, Ctr& c) {
    for (int i = 0; i < sc.nextra; ++i) {
        const Rec128* r = &sc.extra[i];
        double t; D3 pos, nrm;
        uint32_t tests;
        const bool ok = extra_hit(r, rs, rd, t, pos, nrm, tests);
        c.geom += tests;
        if (ok && t <= 1.0) return true;
    }
    return false;
}

// "does this triangle occlude the sample": the conjunction tri_hit && inside(root) && t + offset <= 1.0 with the
// cheap rayFrac test moved in front of the barycentric divisions (a conjunction has no evaluation order)
__device__ __forceinline__ bool tri_blocks(const double* p, const SampleRay& r, const double* lo, const double* hi) {
    double startDist = r.s.x * p[0] + r.s.y * p[1] + r.s.z * p[2];
    double dirDist = r.d.x * p[0] + r.d.y * p[1] + r.d.z * p[2];
    if (dirDist >= 0.0) return false;
    double rf = p[3] - startDist;
    if (!(rf <= 0.0)) return false;
    rf = rf / dirDist;
    if (!(rf + r.offset <= 1.0)) return false;
    D3 q = r.s + r.d * rf;
    if (!inside(lo, hi, q)) return false;
    D3 w = mk(q.x - p[4], q.y - p[5], q.z - p[6]);
    double sv = (w.x * p[7] + w.y * p[8] + w.z * p[9]) / p[10];
    if (sv < 0.0 || sv > 1.0) return false;
    double tv = (w.x * p[11] + w.y * p[12] + w.z * p[13]) / p[14];
    return sv >= 0.0 && tv >= 0.0 && sv + tv <= 1.0;
}

constexpr int kPacketSlots = 2;          // samples per lane in k_shadow_test: S <= 128
constexpr int kShaftCap = 40;            // triangles per candidate list in the first round (every hit): 28: 17.8 ms, 32: 17.5, 40: 17.3, 48: 17.6
// later rounds only see the hits whose earlier candidates left samples undecided: longer lists, fewer hits
constexpr int kRoundCap[kShaftRounds] = {kShaftCap, 64};   // round 2 starts from scratch: room for round 1's candidates and as many new ones
constexpr int kRecordsPerPass = 16;      // records staged through LDS per pass (2 KB per wave)
// LDS strides (bank = address / 4 mod 64): in the (sample x candidate) layout the lanes of a wave read up to 8 different
// records and up to 64 different rays at once; 144-byte records and 80-byte rays spread those over the banks
// (128 / 64-byte strides would put them on 2 / 4 bank groups)
constexpr int kRecStride16 = 9;          // record stride in 16-byte units (128 B payload)
constexpr int kRayStride8 = 10;          // ray stride in doubles (7 used)
constexpr int kTailSlots = 64;           // k_shadow_test switches to (sample x candidate) lanes once this few samples are undecided
constexpr unsigned kTruncated = 0x80000000u;
constexpr unsigned kUmbraItem = 0x40000000u;   // cand_count of a later round: k_shaft found an umbra triangle

// The shaft is parametrised from the surface end: C(u) = E' + u (L - E'), u in [0, 1] (u = 1 - t), so that all fp32
// quantities are small near the surface point (robust for distant lights).  Every sample ray leaves E' in a direction
// D = DL + off, DL = L - E', |off| <= R.  Which triangles can such a ray hit?  With the triangle's TriSlab planes (unit
// normal n, unit in-plane edge normals m_k pointing inward; n, m_k orthonormal) and G0 = n.E' - d, N1 = n.DL,
// K0_k = m_k.E' - c_k, K1_k = m_k.DL: the ray meets the plane at u_c = -G0 / (N1 + n.off) and lies inside edge k there
// iff w_k.D >= 0 with w_k = K0_k n - G0 m_k, the normal of the plane through E' and edge k (see k_shadow_cls).  Since
//     w_k.DL = K0_k N1 - G0 K1_k =: A_k        and        |w_k| = sqrt(K0_k^2 + G0^2) =: L_k        (n is orthogonal to m_k)
// the extremes of w_k.D over the ball of offsets are A_k +- R L_k -- no vector arithmetic at all:
//   candidate  <=  N1 + R >= 0 (front-facing for some sample), G0 <= 0 (plane between light and surface point), G0 + N1 + R >= 0
//                  (some sample starts in front of the plane) and A_k + R L_k >= 0 for k = 1..3 (some direction inside edge k);
//   UMBRA      <=  N1 - R > 0, G0 < 0 and A_k - R L_k > 0 for k = 1..3 (EVERY direction of the ball inside every edge), and the
//                  crossing region inside the root box: Triangle.IntersectRay accepts the crossing of every sample with
//                  rayFrac <= 1.0 (ShadowMethod.cs:170), so rayEscapeCount = 0 whatever the samples are.
// Each necessary condition is tested on its own (not for one common offset): conservative.  fp32 error bounds as in
// k_shadow_cls: |G0|, |K0_k| errors < a0 = 12u s0, |N1|, |K1_k| errors < a1 = 20u dmax, |A_k - true| < eps_k = dmax u (15 s0 + 16 (|K0_k| + |G0|));
// the candidate tests are relaxed and the umbra tests tightened by these.  The kernels are fp32 VALU-throughput bound, so this
// is written for instruction count: packed FMAs for the eight plane functions, v_sqrt_f32 for L_k (its error is far below eps).
struct ShaftRay {
    f2    edx, edy, edz;        // per axis (E' , L - E'): value and slope of the centre ray C(u) = E' + u (L - E')
    float R, Rm, hbx, hby, hbz;
    float backface;             // R * 1.001 + 1e-6 * |L - E'|_1
    float a0, a01;              // a0; a0 + a1 + (Rm - R)
    float c0, c1;               // eps_k = c0 + c1 (|K0_k| + |G0|)
    float umargin;              // 3e-5 * extent: slack of the umbra's root-box test
};

__device__ __forceinline__ ShaftRay make_shaft_ray(const DevScene& sc, const FrameConst& fc, D3 E, D3 lpos) {
    ShaftRay sr;
    const float R = (float)fc.light_radius * 1.00001f + 1e-30f;
    const float bx = (float)(sc.root.max[0] - sc.root.min[0]), by = (float)(sc.root.max[1] - sc.root.min[1]), bz = (float)(sc.root.max[2] - sc.root.min[2]);
    const float ext = fmaxf(fmaxf(bx, by), bz);
    const float ex = (float)(E.x - sc.root.centre[0]), ey = (float)(E.y - sc.root.centre[1]), ez = (float)(E.z - sc.root.centre[2]);
    const float dx = (float)(lpos.x - E.x), dy = (float)(lpos.y - E.y), dz = (float)(lpos.z - E.z);
    sr.edx = (f2){ex, dx}; sr.edy = (f2){ey, dy}; sr.edz = (f2){ez, dz};
    sr.R = R; sr.Rm = R * 1.001f;
    sr.backface = R * 1.001f + 1e-6f * (fabsf(dx) + fabsf(dy) + fabsf(dz)) + 1e-30f;
    sr.hbx = 0.5f * bx; sr.hby = 0.5f * by; sr.hbz = 0.5f * bz;
    // |E'| may exceed the half diagonal by the probe offset; s0 as in cls_frame, evaluated with the actual |E'|
    const float s0 = (__builtin_amdgcn_sqrtf(bx * bx + by * by + bz * bz) * 0.5f + fabsf(ex) + fabsf(ey) + fabsf(ez)) * 1.002f + 0.004f;
    const float dmax = __builtin_amdgcn_sqrtf(dx * dx + dy * dy + dz * dz) * 1.0001f + R;
    const float u = 5.9604645e-8f;
    sr.a0 = 12.0f * u * s0;
    sr.a01 = sr.a0 + 20.0f * u * dmax + 0.002f * R;
    sr.c0 = 15.0f * u * s0 * dmax + 1e-9f * dmax;
    sr.c1 = 16.0f * u * dmax;
    sr.umargin = 3e-5f * ext;
    return sr;
}

// returns 0: no sample ray can hit the triangle; 1: candidate; 2: UMBRA.  Straight-line code (the kernels that call it are
// bound by instruction issue, scalar bookkeeping of nested branches included); `want`: lanes whose verdict is looked at.
// STAGED (the packet walks: all lanes test the SAME triangle with the shafts of one 8x8-pixel tile): the wave leaves after the
// first edge plane that has every lane's shaft on its outside.  Census of the headline frame: of 16.0 M wave-level filters 9.9 M get
// past the plane stage; 5.6 M of those end at the first edge, 1.5 M at the second, 0.9 M at the third, 1.9 M leave a candidate.
template <bool STAGED = false>
__device__ __forceinline__ int shaft_touches(const TriSlab s, const ShaftRay& sr, bool want) {
    // (f0, f1) of the plane and the three edge planes along the centre ray, two lanes per packed FMA
    f2 cn = {-s.d, 0.0f}, c1 = {-s.c1, 0.0f}, c2 = {-s.c2, 0.0f}, c3 = {-s.c3, 0.0f};
    const f2 N = pk_fma(splat(s.n[0]), sr.edx, pk_fma(splat(s.n[1]), sr.edy, pk_fma(splat(s.n[2]), sr.edz, cn)));
    const float G0 = N.x, N1 = N.y;
    // Triangle.IntersectRay is one-sided (dirDist >= 0 -> no hit, Plane.cs:60-61): a triangle with n.DL + R < 0 faces away from
    // EVERY sample ray; G0 > 0: the surface point is in front of the plane (crossing at u < 0, i.e. rayFrac > 1); G0 + N1 + R < 0:
    // every sample starts behind the plane
    const bool plane_ok = want && !(N1 < -sr.backface) && !(G0 > sr.a0) && !(G0 + N1 + sr.R < -sr.a01);
    if (__ballot(plane_ok) == 0ull) return 0;                          // (wave-level: nobody is left)
    const float g2 = G0 * G0, ag = fabsf(G0);
    // some direction of the ball inside every edge (each edge on its own): A_k + R L_k + eps_k >= 0
    const f2 P = pk_fma(splat(s.m1[0]), sr.edx, pk_fma(splat(s.m1[1]), sr.edy, pk_fma(splat(s.m1[2]), sr.edz, c1)));
    const float A1 = __builtin_fmaf(P.x, N1, -(G0 * P.y)), L1 = sr.Rm * __builtin_amdgcn_sqrtf(__builtin_fmaf(P.x, P.x, g2)), e1 = __builtin_fmaf(sr.c1, fabsf(P.x) + ag, sr.c0);
    const bool ok1 = plane_ok && !(A1 + L1 + e1 < 0.0f);
    if (STAGED && __ballot(ok1) == 0ull) return 0;
    const f2 Q = pk_fma(splat(s.m2[0]), sr.edx, pk_fma(splat(s.m2[1]), sr.edy, pk_fma(splat(s.m2[2]), sr.edz, c2)));
    const float A2 = __builtin_fmaf(Q.x, N1, -(G0 * Q.y)), L2 = sr.Rm * __builtin_amdgcn_sqrtf(__builtin_fmaf(Q.x, Q.x, g2)), e2 = __builtin_fmaf(sr.c1, fabsf(Q.x) + ag, sr.c0);
    const bool ok2 = ok1 && !(A2 + L2 + e2 < 0.0f);
    if (STAGED && __ballot(ok2) == 0ull) return 0;
    const f2 T = pk_fma(splat(s.m3[0]), sr.edx, pk_fma(splat(s.m3[1]), sr.edy, pk_fma(splat(s.m3[2]), sr.edz, c3)));
    const float A3 = __builtin_fmaf(T.x, N1, -(G0 * T.y)), L3 = sr.Rm * __builtin_amdgcn_sqrtf(__builtin_fmaf(T.x, T.x, g2)), e3 = __builtin_fmaf(sr.c1, fabsf(T.x) + ag, sr.c0);
    const bool cand = ok2 && !(A3 + L3 + e3 < 0.0f);
    if (STAGED && __ballot(cand) == 0ull) return 0;
    // ---- umbra: every direction of the ball inside every edge ----
    const bool pre = cand && fminf(fminf(A1 - L1 - e1, A2 - L2 - e2), A3 - L3 - e3) > 0.0f && N1 > 2.0f * sr.Rm + sr.a01 && G0 < -4.0f * sr.a0;
    bool umbra = false;
    if (__ballot(pre) != 0ull) {
        // crossing parameters of all samples lie in [ulo, uhi] (v_rcp_f32 is good to 1 ulp; the factors move the bounds outwards)
        const float ulo = -G0 * __builtin_amdgcn_rcpf(N1 + sr.Rm) * 0.999998f, uhi = -G0 * __builtin_amdgcn_rcpf(N1 - sr.Rm) * 1.000002f;
        // the crossing region must also be inside the root box (hits outside it are no hits: SpatialSubdivision.cs:652)
        const float margin = __builtin_fmaf(sr.Rm, uhi, sr.umargin);
        const f2 U = {ulo, uhi};
        const f2 X = pk_fma(U, splat(sr.edx.y), splat(sr.edx.x)), Y = pk_fma(U, splat(sr.edy.y), splat(sr.edy.x)), Z = pk_fma(U, splat(sr.edz.y), splat(sr.edz.x));
        umbra = pre && ulo > 1e-6f && uhi < 0.5f && fmaxf(fabsf(X.x), fabsf(X.y)) + margin < sr.hbx && fmaxf(fabsf(Y.x), fabsf(Y.y)) + margin < sr.hby &&
                fmaxf(fabsf(Z.x), fabsf(Z.y)) + margin < sr.hbz;
    }
    return umbra ? 2 : (cand ? 1 : 0);
}

template <bool STATS>
__global__ __launch_bounds__(256) void k_shaft(DevScene sc, FrameConst fc, const HitRec* __restrict__ hits,
                                               const unsigned int* __restrict__ hit_count, unsigned int count_cap,
                                               const unsigned int* __restrict__ index_list, int node_budget, int cap,
                                               unsigned int* __restrict__ cand_count, int32_t* __restrict__ cand,
                                               uint32_t* __restrict__ samples, unsigned int* __restrict__ work_count,
                                               unsigned int* __restrict__ work_list, unsigned long long* stats, unsigned int* dbg) {
    const int tid = threadIdx.x;
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    const unsigned int total = min(*hit_count, count_cap);
    const unsigned int slot_i = blockIdx.x * 256u + (unsigned)tid;       // list slot: the hit itself, or an entry of index_list
    uint32_t nodes = 0, leaves = 0, slabs = 0;
    bool live = slot_i < total;
    unsigned int h = 0;
    HitRec rec;
    if (live) {
        h = index_list ? index_list[slot_i] : slot_i;
        rec = hits[h];
        live = rec.sample != kInvalidHit;                                  // padding entry of the tile-aligned queue
    }
    if (live) {
        const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
        const D3 E = mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001;   // ShadowMethod.cs:151
        const float R = (float)fc.light_radius * 1.00001f + 1e-30f;
        float ext = 0.0f;
        for (int a = 0; a < 3; ++a) ext = fmaxf(ext, (float)(sc.root.max[a] - sc.root.min[a]));
        const float pad = ext * 3.0517578125e-5f;                          // 2^-15 * extent (boxes carry 2^-16 already)
        const float ex = (float)(E.x - sc.root.centre[0]), ey = (float)(E.y - sc.root.centre[1]), ez = (float)(E.z - sc.root.centre[2]);
        const float dx = (float)(lpos.x - E.x), dy = (float)(lpos.y - E.y), dz = (float)(lpos.z - E.z);
        // a zero direction component acts as a huge finite slope reciprocal: no inf - inf in the fused slab arithmetic
        const float ix = slab_inv(dx), iy = slab_inv(dy), iz = slab_inv(dz);
        // slab test of a box inflated by r, per axis: t_lo = lo*i + (-o*i - r*i), t_hi = hi*i + (-o*i + r*i); the node's
        // floats are consumed in memory order as pairs (lo.x, lo.y) (lo.z, hi.x) (hi.y, hi.z)
        const f2 I01 = {ix, iy}, I20 = {iz, ix}, I12 = {iy, iz};
        const f2 OI01 = {-ex * ix, -ey * iy}, OI20 = {-ez * iz, -ex * ix}, OI12 = {-ey * iy, -ez * iz};
        const f2 RI01 = {-ix, -iy}, RI20 = {-iz, ix}, RI12 = {iy, iz};
        const float umin = -1e-5f;                                         // hits with rayFrac rounding just beyond 1.0
        const ShaftRay sr = make_shaft_ray(sc, fc, E, lpos);
        const int nbits = sc.bnode_bits, qmax = (1 << (31 - nbits)) - 1;   // stack word = node | quantised u bound
        const float qinv = 1.0f / (float)qmax * 1.000001f;
        int32_t* out = cand + (size_t)slot_i * cap;
        int count = 0;
        bool truncated = false, umbra = false;
        int sp = 0;
        int32_t ni = 0;
        float nu = 1.0f;               // upper bound of u (distance from the surface end) inside the current subtree
        // while-while: walk inner nodes until this lane owns a pending leaf, then run the (per-triangle) slab filters
        int32_t leafA = -1, leafB = -1;            // pending leaves: first record | count << kLeafShift
        for (;;) {
            while (ni >= 0 && leafA < 0) {
                const BvhNode n = sc.bnodes[ni];
                nodes++;
                // a private walk is a chain of dependent fetches and the kernel lasts as long as its longest one: with a budget
                // (sr_debug_set, default none) a walk that has not filled its list within node_budget nodes gives up ("truncated":
                // its undecided samples go to the exact fallback)
                if (node_budget > 0 && nodes > (uint32_t)node_budget) { truncated = true; break; }
                const f2 rr = splat(__builtin_fmaf(R, fminf(1.0f, fmaxf(0.0f, nu + 1e-5f)), pad));
                const f2 B0 = pk_fma(rr, RI01, OI01), B1 = pk_fma(rr, RI20, OI20), B2 = pk_fma(rr, RI12, OI12);
                float a0, b0, a1, b1;      // child u-intervals [a, b]
                node_slabs(n, I01, I20, I12, B0, B1, B2, a0, b0, a1, b1);
                a0 = fmaxf(a0, umin); a1 = fmaxf(a1, umin);
                b0 = fminf(b0, nu); b1 = fminf(b1, nu);
                const bool h0 = n.n0 >= 0 && a0 <= b0, h1 = n.n1 >= 0 && a1 <= b1;
                // leaves: the one nearest to the surface point first
                const bool l0 = h0 && n.n0 > 0, l1 = h1 && n.n1 > 0;
                if (l0 && l1) {
                    const bool first0 = a0 <= a1;
                    leafA = (first0 ? n.c0 : n.c1) | ((first0 ? n.n0 : n.n1) << kLeafShift);
                    leafB = (first0 ? n.c1 : n.c0) | ((first0 ? n.n1 : n.n0) << kLeafShift);
                } else if (l0) leafA = n.c0 | (n.n0 << kLeafShift);
                else if (l1) leafA = n.c1 | (n.n1 << kLeafShift);
                const bool i0 = h0 && n.n0 == 0, i1 = h1 && n.n1 == 0;
                if (i0 && i1) {
                    const bool first0 = a0 <= a1;                          // the child nearest to the surface point first
                    // one stack word: node index | u upper bound quantised UP to qbits bits (conservative)
                    const float bu = fminf(1.0f, fmaxf(0.0f, first0 ? b1 : b0));
                    const int qu = min(qmax, (int)(bu * (float)qmax) + 1);
                    st.put(sp++, (first0 ? n.c1 : n.c0) | (qu << nbits));
                    ni = first0 ? n.c0 : n.c1; nu = first0 ? b0 : b1;
                } else if (i0) { ni = n.c0; nu = b0; }
                else if (i1) { ni = n.c1; nu = b1; }
                else if (sp > 0) {
                    const int w = st.get(--sp);
                    ni = w & ((1 << nbits) - 1);
                    nu = (float)((unsigned)w >> nbits) * qinv;
                } else ni = -1;
            }
            if (leafA < 0) break;
            while (leafA >= 0 && !truncated && !umbra) {
                const int first = leafA & kLeafMask, cnt = (leafA >> kLeafShift) & 15;
                leafA = leafB;
                leafB = -1;
                leaves++;
                slabs += (uint32_t)cnt;
                for (int q = 0; q < cnt; ++q) {
                    const int touch = shaft_touches(sc.bslab[first + q], sr, true);
                    if (touch == 2) umbra = true;
                    if (touch) {
                        if (count < cap) { if (count >= 0) out[count] = first + q; count++; }
                        else truncated = true;
                    }
                }
            }
            if (truncated || umbra) break;
        }
        if (umbra && !work_list) {
            // later round: k_shadow_test owns the hit point's sample masks and finishes it -- every undecided sample is blocked
            cand_count[slot_i] = kUmbraItem;
        } else if (umbra) {
            // fully shadowed: rayEscapeCount = 0 -> (byte)(0.0 * 255) = 0 -> ModulatePackedColor(color, 0) = opaque black
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], 0.0);
            cand_count[slot_i] = 0u;
            if (fc.debug == 7) atomicAdd(dbg + 7, 1u);
        } else if (work_list && count == 0 && !truncated && sc.nextra == 0) {
            // first round, nothing in the whole shaft and no extra geometry: every sample escapes,
            // rayEscapeCount = S -> (byte)(1.0 * 255) = 255 (ShadowMethod.cs:113-119); the pixel is finished here
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], (double)fc.shadow_samples / (double)fc.shadow_samples);
            cand_count[slot_i] = 0u;
        } else {
            cand_count[slot_i] = (unsigned)max(count, 0) | (truncated ? kTruncated : 0u);
            if (work_list) work_list[atomicAdd(work_count, 1u)] = h;      // the compiler aggregates this per wavefront
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(nodes), b = wave_sum(leaves), c2 = wave_sum(slabs), d2 = wave_sum(live ? 1u : 0u);
        block_stat_add(&stats[6], &stats[7], &stats[10], &stats[11], a, b, c2, d2);
        // private walks (later rounds) once more on their own, so that the packet kernel's share of [6], [10], [11] can be told apart
        block_stat_add(&stats[14], &stats[15], &stats[14], &stats[15], a, c2, 0u, 0u);
    }
}

// --------------------------------------------------------------------------------------------------
// k_shaft_coop -- the later shaft rounds, ONE WAVE PER HIT POINT.  The hit points of a later round are scattered (0.8 % of
// the frame's), so there is no packet to share a walk with, and a private walk is a chain of ~140 dependent node fetches
// and ~400 triangle filters per lane: the kernel lasted as long as one such chain however few hit points there were (the
// floor of a 1/8-frame rank), at 16 of 64 lanes active.  Here the 64 lanes of a wave work on ONE shaft: a frontier of
// pending inner nodes and a queue of pending triangles live in LDS; a node pass lets up to 64 lanes fetch one frontier
// node each, test both children's boxes and append what the shaft touches (inner children -> frontier, the triangles of
// leaf children -> triangle queue); a triangle pass lets up to 64 lanes run shaft_touches on one triangle each and append
// the candidates to the hit point's list.  The walk's depth in dependent fetches drops from hundreds to ~20 passes.
// The arithmetic per node / triangle is k_shaft's; only the order of the list differs, which no later stage depends on
// (launch_shadow_t: every round collects from scratch).  Frontier overflow = "truncated" (undecided samples go to the
// exact fallback), like a full list.  Hit points are dealt round-robin to the waves of a fixed-size grid.
// --------------------------------------------------------------------------------------------------
constexpr int kCoopFrontier = 1024;      // pending inner nodes per wave
constexpr int kCoopPasses = 1 << 16;     // passes per hit point (a 1M-triangle tree in full: 2^15 triangle passes)
constexpr int kCoopTris = 1024;          // pending triangles per wave: < 64 before a node pass + at most 64 * 2 * 7 from it (default leaves)

template <bool STATS>
__global__ __launch_bounds__(256) void k_shaft_coop(DevScene sc, FrameConst fc, const HitRec* __restrict__ hits,
                                                    const unsigned int* __restrict__ hit_count, unsigned int count_cap,
                                                    const unsigned int* __restrict__ index_list, int cap,
                                                    unsigned int* __restrict__ cand_count, int32_t* __restrict__ cand,
                                                    unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint32_t* fr = reinterpret_cast<uint32_t*>(lds_pipe) + (size_t)wave * (kCoopFrontier + kCoopTris);
    uint32_t* tq = fr + kCoopFrontier;
    const unsigned int total = min(*hit_count, count_cap);
    const unsigned long long lt = lanemask_lt();
    uint32_t nodes = 0, slabs = 0, items = 0;                         // wave-level
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const float R = (float)fc.light_radius * 1.00001f + 1e-30f;
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, (float)(sc.root.max[a] - sc.root.min[a]));
    const float pad = ext * 3.0517578125e-5f;                          // 2^-15 * extent (boxes carry 2^-16 already)
    const int nbits = sc.bnode_bits, qmax = (1 << (31 - nbits)) - 1;   // frontier word = node | quantised u bound (rounded UP)
    const float qinv = 1.0f / (float)qmax * 1.000001f;
    const float umin = -1e-5f;                                         // hits with rayFrac rounding just beyond 1.0
    // hit points are dealt to the waves of the (fixed-size) grid round-robin: wave-uniform loop bounds, every wave reaches the exit
    const unsigned int wave_id = blockIdx.x * 4u + (unsigned)wave, wave_count = gridDim.x * 4u;
    for (unsigned int item = wave_id; item < total; item += wave_count) {
        const unsigned int h = index_list[item];
        const HitRec rec = hits[h];
        items++;
        const D3 E = mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001;   // ShadowMethod.cs:151
        const float ex = (float)(E.x - sc.root.centre[0]), ey = (float)(E.y - sc.root.centre[1]), ez = (float)(E.z - sc.root.centre[2]);
        const float dx = (float)(lpos.x - E.x), dy = (float)(lpos.y - E.y), dz = (float)(lpos.z - E.z);
        const float ix = slab_inv(dx), iy = slab_inv(dy), iz = slab_inv(dz);
        const f2 I01 = {ix, iy}, I20 = {iz, ix}, I12 = {iy, iz};
        const f2 OI01 = {-ex * ix, -ey * iy}, OI20 = {-ez * iz, -ex * ix}, OI12 = {-ey * iy, -ez * iz};
        const f2 RI01 = {-ix, -iy}, RI20 = {-iz, ix}, RI12 = {iy, iz};
        const ShaftRay sr = make_shaft_ray(sc, fc, E, lpos);
        int32_t* out = cand + (size_t)item * cap;
        int count = 0, nf = 1, nt = 0;                                 // wave-uniform
        bool truncated = false, umbra = false;
        if (lane == 0) fr[0] = (uint32_t)qmax << nbits;                // the root, u <= 1
        // (a pass budget bounds the loop whatever the tree holds: a walk of that many passes gives up as "truncated")
        for (int passes = 0; (nf > 0 || nt > 0) && !truncated && !umbra; ++passes) {
            if (passes >= kCoopPasses) truncated = true;
            else if (nt >= 64 || nf == 0) {
                // ---- triangle pass: the newest (up to) 64 pending triangles ----
                const int take = min(nt, 64);
                nt -= take;
                const bool act = lane < take;
                const int tri = act ? (int)tq[nt + lane] : 0;
                slabs += (uint32_t)take;
                const int touch = shaft_touches(sc.bslab[tri], sr, act);
                const unsigned long long m = __ballot(act && touch != 0);
                const int pos = count + __popcll(m & lt);
                if (act && touch != 0 && pos < cap) out[pos] = tri;
                count += __popcll(m);
                truncated = count > cap;
                count = min(count, cap);
                umbra = __ballot(act && touch == 2) != 0ull;
            } else {
                // ---- node pass: the newest (up to) 64 frontier nodes ----
                const int take = min(nf, 64);
                nf -= take;
                const bool act = lane < take;
                const uint32_t w = act ? fr[nf + lane] : 0u;
                nodes += (uint32_t)take;
                const BvhNode n = sc.bnodes[w & ((1u << nbits) - 1u)];
                const float nu = (float)(w >> nbits) * qinv;
                const f2 rr = splat(__builtin_fmaf(R, fminf(1.0f, fmaxf(0.0f, nu + 1e-5f)), pad));
                const f2 B0 = pk_fma(rr, RI01, OI01), B1 = pk_fma(rr, RI20, OI20), B2 = pk_fma(rr, RI12, OI12);
                float a0, b0, a1, b1;                                  // child u-intervals [a, b]
                node_slabs(n, I01, I20, I12, B0, B1, B2, a0, b0, a1, b1);
                a0 = fmaxf(a0, umin); a1 = fmaxf(a1, umin);
                b0 = fminf(b0, nu); b1 = fminf(b1, nu);
                const bool h0 = act && n.n0 >= 0 && a0 <= b0, h1 = act && n.n1 >= 0 && a1 <= b1;
                const bool i0 = h0 && n.n0 == 0, i1 = h1 && n.n1 == 0;
                const unsigned long long m0 = __ballot(i0), m1 = __ballot(i1);
                const int add = __popcll(m0) + __popcll(m1);
                // the triangles of leaf children (exclusive prefix sum of the <= 5-bit counts by ballots)
                const int c0 = (h0 && n.n0 > 0) ? min(n.n0, 15) : 0, c1 = (h1 && n.n1 > 0) ? min(n.n1, 15) : 0;
                const int cl = c0 + c1;                                // <= 14 with the default <= 7 triangles per leaf
                int before = 0, sum = 0;
#pragma unroll
                for (int b = 0; b < 5; ++b) {
                    const unsigned long long mb = __ballot((cl >> b) & 1);
                    before += __popcll(mb & lt) << b;
                    sum += __popcll(mb) << b;
                }
                // no room (frontier: very wide shafts; triangle queue: only with leaves of > 7 triangles): give up as "truncated"
                truncated = nf + add > kCoopFrontier || nt + sum > kCoopTris;
                if (!truncated) {
                    // inner children -> frontier
                    if (i0) fr[nf + __popcll(m0 & lt)] = (uint32_t)n.c0 | ((uint32_t)min(qmax, (int)(fminf(1.0f, fmaxf(0.0f, b0)) * (float)qmax) + 1) << nbits);
                    if (i1) fr[nf + __popcll(m0) + __popcll(m1 & lt)] = (uint32_t)n.c1 | ((uint32_t)min(qmax, (int)(fminf(1.0f, fmaxf(0.0f, b1)) * (float)qmax) + 1) << nbits);
                    nf += add;
                    // leaf triangles -> triangle queue
                    uint32_t* dst = tq + nt + before;
                    for (int q = 0; q < c0; ++q) dst[q] = (uint32_t)(n.c0 + q);
                    for (int q = 0; q < c1; ++q) dst[c0 + q] = (uint32_t)(n.c1 + q);
                    nt += sum;
                }
                // the next pass reads what other lanes of this wave have just queued (LDS executes a wave's accesses in order;
                // the fence keeps the compiler from moving them)
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
        if (lane == 0) cand_count[item] = umbra ? kUmbraItem : ((unsigned)count | (truncated ? kTruncated : 0u));
    }
    if (STATS) {
        block_stat_add(&stats[6], &stats[10], &stats[11], &stats[14], nodes, slabs, items, nodes);
        block_stat_add(&stats[15], &stats[15], &stats[15], &stats[15], slabs, 0u, 0u, 0u);
    }
}

// --------------------------------------------------------------------------------------------------
// k_shaft_pkt -- the first shaft round as a wave-cooperative PACKET walk.  A wave owns 64 consecutive entries of the
// tile-aligned hit queue = the surface points of one 8x8-pixel tile, whose shafts all end in the same light and start
// within a few thousandths of each other: their walks visit nearly the same nodes.  So the wave walks the BVH ONCE --
// one shared stack (LDS, wave-uniform), node and TriSlab records fetched with scalar loads, near child first by a vote of
// the interested lanes -- and every lane tests its OWN shaft against the broadcast boxes / triangles, keeping its own
// u-bound per stacked subtree (-1: the lane's shaft misses it).  A lane drops out when its list is full or it found an
// umbra triangle; the walk ends when no live lane wants anything that is left.  Per-lane arithmetic and results are those
// of k_shaft; only the ORDER in which a lane meets its candidates differs (it is the wave's), which is why the second round
// re-collects from scratch instead of skipping "the first cap candidates" (launch_shadow_t).
// k_shaft kept 19 of 64 lanes busy on average (neighbouring lanes are in different phases of their private loops);
// here all lanes are at the same node at the same time.
// --------------------------------------------------------------------------------------------------
template <bool STATS>
__global__ __launch_bounds__(256, 7) void k_shaft_pkt(DevScene sc, FrameConst fc, const HitRec* __restrict__ hits,
                                                   const unsigned int* __restrict__ hit_count, int cap, int levels, int tile_n2, int tile_rows,
                                                   unsigned int* __restrict__ cand_count, int32_t* __restrict__ cand,
                                                   uint32_t* __restrict__ samples, unsigned int* __restrict__ work_count,
                                                   unsigned int* __restrict__ work_list, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t* wnode = reinterpret_cast<int32_t*>(lds_pipe) + (size_t)wave * ((size_t)levels * 33);   // [levels] stacked node
    // [levels][64] per-lane u bound, 16 bits: 0 = the lane's shaft misses the subtree, else 1 + the bound in 1/65534 rounded UP
    uint16_t* wbound = reinterpret_cast<uint16_t*>(wnode + levels) + lane;
    const unsigned int total = *hit_count;
    unsigned int slot_i = blockIdx.x * 256u + (unsigned)tid;
    if (tile_n2 > 0) {
        // tile-indexed queue (see k_primary): block b * n2 + sub-sample of this grid <-> block b of k_primary, same XCD
        int tile_x, tile_y;
        const int pb = (int)blockIdx.x / tile_n2, si = (int)blockIdx.x - pb * tile_n2;
        if (!xcd_tile(pb, fc.width, tile_rows, tile_x, tile_y)) return;
        const int tiles_x = (fc.width + 15) >> 4;
        slot_i = ((unsigned)(tile_y * tiles_x + tile_x) * (unsigned)tile_n2 + (unsigned)si) * 256u + (unsigned)tid;
    }
    HitRec rec;
    rec.sample = kInvalidHit;
    if (slot_i < total) rec = hits[slot_i];
    const bool valid = rec.sample != kInvalidHit;
    uint32_t nodes = 0, leaves = 0, slabs = 0;                    // wave-level (uniform)
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const D3 E = valid ? mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001 : lpos * 0.5;   // ShadowMethod.cs:151
    const float R = (float)fc.light_radius * 1.00001f + 1e-30f;
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, (float)(sc.root.max[a] - sc.root.min[a]));
    const float pad = ext * 3.0517578125e-5f;                          // 2^-15 * extent (boxes carry 2^-16 already)
    const float ex = (float)(E.x - sc.root.centre[0]), ey = (float)(E.y - sc.root.centre[1]), ez = (float)(E.z - sc.root.centre[2]);
    const float dx = (float)(lpos.x - E.x), dy = (float)(lpos.y - E.y), dz = (float)(lpos.z - E.z);
    const float ix = slab_inv(dx), iy = slab_inv(dy), iz = slab_inv(dz);
    const f2 I01 = {ix, iy}, I20 = {iz, ix}, I12 = {iy, iz};
    const f2 OI01 = {-ex * ix, -ey * iy}, OI20 = {-ez * iz, -ex * ix}, OI12 = {-ey * iy, -ez * iz};
    const f2 RI01 = {-ix, -iy}, RI20 = {-iz, ix}, RI12 = {iy, iz};
    const float umin = -1e-5f;                                         // hits with rayFrac rounding just beyond 1.0
    const ShaftRay sr = make_shaft_ray(sc, fc, E, lpos);
    int32_t* out = cand + (size_t)slot_i * cap;
    int count = 0;
    bool truncated = false, umbra = false;
    bool done = !valid;
    int sp = 0;                      // wave-uniform
    int32_t ni = 0;                  // wave-uniform: current inner node
    float nu = valid ? 1.0f : -1.0f; // this lane's u bound inside the current subtree; < 0: the lane's shaft misses it
    for (;;) {
        if (__ballot(!done && nu >= 0.0f) == 0ull) {
            // nobody wants the current subtree: pop until a live lane wants one
            bool found = false;
            while (sp > 0) {
                --sp;
                const uint32_t qb = wbound[sp * 64];
                const float bu = qb ? (float)(qb - 1u) * (1.0f / 65534.0f) * 1.000001f : -1.0f;
                if (__ballot(!done && bu >= 0.0f) != 0ull) { ni = __builtin_amdgcn_readfirstlane(wnode[sp]); nu = bu; found = true; break; }
            }
            if (!found) break;
        }
        const BvhNode n = sc.bnodes[ni];                               // wave-uniform address: scalar loads
        nodes++;
        const f2 rr = splat(__builtin_fmaf(R, fminf(1.0f, fmaxf(0.0f, nu + 1e-5f)), pad));
        const f2 B0 = pk_fma(rr, RI01, OI01), B1 = pk_fma(rr, RI20, OI20), B2 = pk_fma(rr, RI12, OI12);
        float a0, b0, a1, b1;                                          // child u-intervals [a, b] of this lane's shaft
        node_slabs(n, I01, I20, I12, B0, B1, B2, a0, b0, a1, b1);
        a0 = fmaxf(a0, umin); a1 = fmaxf(a1, umin);
        b0 = fminf(b0, nu); b1 = fminf(b1, nu);
        const bool h0 = !done && n.n0 >= 0 && a0 <= b0, h1 = !done && n.n1 >= 0 && a1 <= b1;
        // the child nearest to the surface points first: vote of the lanes that touch both (a lane with one child has no opinion)
        const bool first0 = __popcll(__ballot(h0 && h1 && a0 <= a1)) >= __popcll(__ballot(h0 && h1 && a1 < a0));
        // ---- leaf children: every interested lane filters the (broadcast) triangles with its own shaft ----
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const bool c0 = (t == 0) == first0;
            const int cn = c0 ? n.n0 : n.n1, cc = c0 ? n.c0 : n.c1;
            if (cn > 0 && __ballot((c0 ? h0 : h1) && !done) != 0ull) {
                const bool hc = c0 ? h0 : h1;
                leaves++;
                slabs += (uint32_t)cn;
                for (int q = 0; q < cn; ++q) {
                    const TriSlab s = sc.bslab[cc + q];               // scalar load
                    const bool live_q = hc && !done;
                    const int touch = shaft_touches<true>(s, sr, live_q);
                    const bool take = live_q && touch != 0, room = count < cap;
                    if (take && room) out[count] = cc + q;
                    count += (take && room) ? 1 : 0;
                    truncated = truncated || (take && !room);
                    umbra = umbra || (live_q && touch == 2);
                    done = done || truncated || umbra;
                }
            }
        }
        // ---- inner children ----
        const bool w0 = h0 && !done && n.n0 == 0, w1 = h1 && !done && n.n1 == 0;
        const bool any0 = __ballot(w0) != 0ull, any1 = __ballot(w1) != 0ull;
        if (any0 && any1) {
            const int32_t far_node = first0 ? n.c1 : n.c0;
            wnode[sp] = far_node;
            {
                const float fb = first0 ? (w1 ? b1 : -1.0f) : (w0 ? b0 : -1.0f);
                wbound[sp * 64] = fb < 0.0f ? (uint16_t)0 : (uint16_t)min(65535, (int)(fminf(fb, 1.0f) * 65534.0f) + 2);
            }
            sp++;
            ni = first0 ? n.c0 : n.c1;
            nu = first0 ? (w0 ? b0 : -1.0f) : (w1 ? b1 : -1.0f);
        } else if (any0) { ni = n.c0; nu = w0 ? b0 : -1.0f; }
        else if (any1) { ni = n.c1; nu = w1 ? b1 : -1.0f; }
        else nu = -1.0f;                                               // leaf-only / dead end: pop at the top of the loop
    }
    if (valid) {
        if (umbra) {
            // fully shadowed: rayEscapeCount = 0 -> (byte)(0.0 * 255) = 0 -> ModulatePackedColor(color, 0) = opaque black
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], 0.0);
            cand_count[slot_i] = 0u;
        } else if (count == 0 && !truncated && sc.nextra == 0) {
            // nothing in the whole shaft and no extra geometry: every sample escapes (ShadowMethod.cs:113-119)
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], (double)fc.shadow_samples / (double)fc.shadow_samples);
            cand_count[slot_i] = 0u;
        } else {
            cand_count[slot_i] = (unsigned)count | (truncated ? kTruncated : 0u);
            work_list[atomicAdd(work_count, 1u)] = slot_i;             // the compiler aggregates this per wavefront
        }
    }
    if (STATS) {
        const uint32_t d2 = wave_sum(valid ? 1u : 0u);
        block_stat_add(&stats[6], &stats[7], &stats[10], &stats[11], nodes, leaves, slabs, d2);
    }
}

// the slab test of one child of the LIGHT-ordered copy (plane pairs (lo.x, lo.y), (hi.x, hi.y), (lo.z, hi.z) in the slots lo[0..1], (lo[2], hi[0]),
// hi[1..2]: k_order_nodes): child_slabs' arithmetic -- the same products, the same sums -- with the pairs that need two reciprocal vectors only
template <int KNOWN = 0>
__device__ __forceinline__ void shaft_slabs(const Bvh4Child& ch, f2 Ixy, f2 Izz, f2 B0, f2 B1, f2 B2, float& a, float& b) {
    const f2 T0 = pk_fma((f2){ch.lo[0], ch.lo[1]}, Ixy, B0), T1 = pk_fma((f2){ch.lo[2], ch.hi[0]}, Ixy, B1), T2 = pk_fma((f2){ch.hi[1], ch.hi[2]}, Izz, B2);
    // fminf/fmaxf drop a NaN operand: conservative
    const float nx = (KNOWN & 1) ? T0.x : fminf(T0.x, T1.x), fx = (KNOWN & 1) ? T1.x : fmaxf(T0.x, T1.x);
    const float ny = (KNOWN & 2) ? T0.y : fminf(T0.y, T1.y), fy = (KNOWN & 2) ? T1.y : fmaxf(T0.y, T1.y);
    const float nz = (KNOWN & 4) ? T2.x : fminf(T2.x, T2.y), fz = (KNOWN & 4) ? T2.y : fmaxf(T2.x, T2.y);
    a = fmaxf(fmaxf(nx, ny), nz);
    b = fminf(fminf(fx, fy), fz);
}

// --------------------------------------------------------------------------------------------------
// k_shaft_pkt4 -- the same packet walk on the FOUR-WIDE tree (Bvh4Node): a step fetches one 128-byte node (one pair of scalar
// loads) and tests four children's boxes, so a tile's walk is about half as many dependent steps, scalar-cache round trips and
// stack operations.  The nodes come from the frame's LIGHT-ordered copy (k_order_nodes): all shafts end in the same light, so
// the children of a node are met in the same order by every shaft -- stored nearest-to-the-surface first.  No vote: leaf
// children are filtered in slot order, the inner children that a live lane wants are pushed far to near (node word + one row
// of per-lane u bounds each) and the nearest is entered.  Per-lane arithmetic (slab test, shaft_touches) is k_shaft_pkt's;
// the candidate lists may differ in order and in which candidates a truncated list holds, which no later stage depends on.
// LDS per wave: [levels] node words + [levels][64] 16-bit bounds, levels = 3 * b4depth + 2.
// --------------------------------------------------------------------------------------------------
template <bool STATS, int WAVES, bool PERSIST, int KNOWN = 0>
__global__ __launch_bounds__(256, WAVES) void k_shaft_pkt4(DevScene sc_arg, FrameConst fc_arg, const HitRec* __restrict__ hits,
                                                    const unsigned int* __restrict__ hit_count, int cap, int levels, int tile_n2, int tile_rows,
                                                    unsigned int* __restrict__ cand_count, int32_t* __restrict__ cand,
                                                    uint32_t* __restrict__ samples, unsigned int* __restrict__ work_count,
                                                    unsigned int* __restrict__ work_list, unsigned long long* stats,
                                                    unsigned int* __restrict__ tile_heads, unsigned int virtual_blocks,
                                                    const unsigned int* __restrict__ tile_order, unsigned int* __restrict__ tile_cost) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int32_t* wnode = reinterpret_cast<int32_t*>(lds_pipe) + (size_t)wave * ((size_t)levels * 33);   // [levels] stacked node
    // [levels][64] per-lane u bound, 16 bits: 0 = the lane's shaft misses the subtree, else 1 + the bound in 1/65534 rounded UP
    uint16_t* wbound = reinterpret_cast<uint16_t*>(wnode + levels) + lane;
    const unsigned int total = *hit_count;
    uint32_t nodes = 0, leaves = 0, slabs = 0, points = 0;        // wave-level (uniform)
    uint32_t walk = 0;                                            // (persistent) node steps + triangle filters of the tile being walked
    uint32_t top21 = 0, top85 = 0;                                // (STATS) node steps in the top three / four levels of the tree (level order: node < 21 / < 85)
    // A work item is ONE WAVE's 64 queue entries (an 8x8-pixel tile of surface points) of virtual block vb = the block of the one-
    // workgroup-per-16x16-tile grid that k_primary filled them from.  tile_heads == nullptr: this workgroup IS virtual block
    // blockIdx.x (one item per wave).  Otherwise the grid is persistent and every WAVE pulls items from per-XCD counters (see
    // TileFeed): a wave that finishes a short walk starts the next tile at once instead of idling until the slowest of its
    // workgroup's four walks ends, and the launch ends when the longest single walk does, not the longest workgroup.
    TileFeed<PERSIST> feed(tile_heads, virtual_blocks, wave, lane, tile_order);
    unsigned int vb, vq;
    while (feed.next(vb, vq)) {
    // Scene and frame are read from the kernel-argument segment WHERE an iteration needs them (kernarg_late): the per-tile prologue's
    // inputs here, the epilogue's there.  Read as plain kernel arguments the compiler hoists them -- and everything it derives from
    // them -- out of this loop and keeps it live across the walk (25 spilled SGPRs + 54 VGPRs); read in one piece at the top of
    // the loop they stay live from there to their uses (42 spilled VGPRs).  The walk itself only needs two pointers.
    const DevScene sc = tile_arg<PERSIST>(sc_arg, offsetof(SceneFrameArgs, sc));      // (prologue: root box)
    const FrameConst fc = tile_arg<PERSIST>(fc_arg, offsetof(SceneFrameArgs, fc));  // (prologue: light, width)
    unsigned int slot_i = vb * 256u + vq * 64u + (unsigned)lane;
    if (tile_n2 > 0) {
        // tile-indexed queue (see k_primary): block b * n2 + sub-sample of this grid <-> block b of k_primary, same XCD
        int tile_x, tile_y;
        const int pb = (int)vb / tile_n2, si = (int)vb - pb * tile_n2;
        if (!xcd_tile(pb, fc.width, tile_rows, tile_x, tile_y)) continue;
        const int tiles_x = (fc.width + 15) >> 4;
        slot_i = ((unsigned)(tile_y * tiles_x + tile_x) * (unsigned)tile_n2 + (unsigned)si) * 256u + vq * 64u + (unsigned)lane;
    }
    HitRec rec;
    rec.sample = kInvalidHit;
    if (slot_i < total) rec = hits[slot_i];
    const bool valid = rec.sample != kInvalidHit;
    if (PERSIST && tile_cost) { if (lane == 0) tile_cost[feed.item] = 0u; walk = 0u; }
    if (__ballot(valid) == 0ull) continue;                         // background tile
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const D3 E = valid ? mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001 : lpos * 0.5;   // ShadowMethod.cs:151
    const float R = (float)fc.light_radius * 1.00001f + 1e-30f;
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, (float)(sc.root.max[a] - sc.root.min[a]));
    const float pad = ext * 3.0517578125e-5f;                          // 2^-15 * extent (boxes carry 2^-16 already)
    const float ex = (float)(E.x - sc.root.centre[0]), ey = (float)(E.y - sc.root.centre[1]), ez = (float)(E.z - sc.root.centre[2]);
    const float dx = (float)(lpos.x - E.x), dy = (float)(lpos.y - E.y), dz = (float)(lpos.z - E.z);
    const float ix = slab_inv(dx), iy = slab_inv(dy), iz = slab_inv(dz);
    const f2 Ixy = {ix, iy}, Izz = {iz, iz};                           // (the light-ordered copy's plane pairs: see k_order_nodes)
    // sign of the shafts' direction on the axes where the light lies outside the root box (wave-uniform; |i| = s i there)
    const float known_sx = fc.light_pos_model[0] > sc.root.centre[0] ? 1.0f : -1.0f, known_sy = fc.light_pos_model[1] > sc.root.centre[1] ? 1.0f : -1.0f,
                known_sz = fc.light_pos_model[2] > sc.root.centre[2] ? 1.0f : -1.0f;
    const f2 Oxy = {-ex * ix, -ey * iy}, Ozz = splat(-ez * iz);
    const float umin = -1e-5f;                                         // hits with rayFrac rounding just beyond 1.0
    const ShaftRay sr = make_shaft_ray(sc, fc, E, lpos);
    int32_t* out = cand + (size_t)slot_i * cap;
    int count = 0;
    bool truncated = false, umbra = false;
    bool done = !valid;
    int sp = 0;                      // wave-uniform: entries in LDS
    // the TOP of the stack lives in registers (node: wave-uniform, bound: one encoded 16-bit value per lane): most nodes of the lowest inner level
    // have leaf children only, so every other step ends in a pop -- two dependent LDS reads (the lanes' bounds, then the node) before the next node
    // can even be requested.  With the top in registers a pop is immediate and the entry below it is fetched while the popped subtree's node is on its way.
    bool has_top = false;            // wave-uniform
    int32_t top_ni = 0;              // the same value in every lane (a vector register: the LDS read that refills it is not waited for)
    uint32_t top_b = 0u;             // this lane's encoded u bound of the top entry
    int32_t ni = 0;                  // wave-uniform: current inner node
    float nu = valid ? 1.0f : -1.0f; // this lane's u bound inside the current subtree; < 0: the lane's shaft misses it
    auto enc = [](float fb) { return fb < 0.0f ? (uint16_t)0 : (uint16_t)min(65535, (int)(fminf(fb, 1.0f) * 65534.0f) + 2); };
    for (;;) {
        if (__ballot(!done && nu >= 0.0f) == 0ull) {
            if (__ballot(!done) == 0ull) break;                        // every lane has its verdict: nothing on the stack matters
            // nobody wants the current subtree: pop until a live lane wants one
            bool found = false;
            while (has_top) {
                const uint32_t qb = top_b;
                const int32_t qn = top_ni;
                if (sp > 0) { --sp; top_b = wbound[sp * 64]; top_ni = wnode[sp]; }      // (the entry below: needed at the next pop, so nothing waits for these two reads here)
                else has_top = false;
                const float bu = qb ? (float)(qb - 1u) * (1.0f / 65534.0f) * 1.000001f : -1.0f;
                if (__ballot(!done && bu >= 0.0f) != 0ull) { ni = __builtin_amdgcn_readfirstlane(qn); nu = bu; found = true; break; }
            }
            if (!found) break;
        }
        const Bvh4Node n = load_uniform(&sc_arg.b4light[ni]);          // wave-uniform address: scalar loads
        nodes++;
        if (STATS) { top21 += ni < 21 ? 1u : 0u; top85 += ni < 85 ? 1u : 0u; }
        if (PERSIST) walk += 2u;
        // the shaft's radius moves a box's lo planes by -r and its hi planes by +r: (lo.x, lo.y), (hi.x, hi.y), (lo.z, hi.z) against the same
        // reciprocals the planes are multiplied with (no second set of per-lane constants kept across the walk)
        const float rs = __builtin_fmaf(R, fminf(1.0f, fmaxf(0.0f, nu + 1e-5f)), pad);
        // (on a KNOWN axis the copy holds (near, far): the near plane moves by -r |i|, the far plane by +r |i|; the sign of i is the frame's there)
        const float rx = (KNOWN & 1) ? rs * known_sx : rs, ry = (KNOWN & 2) ? rs * known_sy : rs, rz = (KNOWN & 4) ? rs * known_sz : rs;
        const f2 rr = {rx, ry}, rn = {-rx, -ry}, rm = {-rz, rz};
        const f2 B0 = pk_fma(rn, Ixy, Oxy), B1 = pk_fma(rr, Ixy, Oxy), B2 = pk_fma(rm, Izz, Ozz);
        float a0, b0, a1, b1, a2, b2, a3, b3;                          // child u-intervals [a, b] of this lane's shaft
        shaft_slabs<KNOWN>(n.ch[0], Ixy, Izz, B0, B1, B2, a0, b0);
        shaft_slabs<KNOWN>(n.ch[1], Ixy, Izz, B0, B1, B2, a1, b1);
        shaft_slabs<KNOWN>(n.ch[2], Ixy, Izz, B0, B1, B2, a2, b2);
        shaft_slabs<KNOWN>(n.ch[3], Ixy, Izz, B0, B1, B2, a3, b3);
        b0 = fminf(b0, nu); b1 = fminf(b1, nu); b2 = fminf(b2, nu); b3 = fminf(b3, nu);
        const bool h0 = !done && n.ch[0].n >= 0 && fmaxf(a0, umin) <= b0, h1 = !done && n.ch[1].n >= 0 && fmaxf(a1, umin) <= b1;
        const bool h2 = !done && n.ch[2].n >= 0 && fmaxf(a2, umin) <= b2, h3 = !done && n.ch[3].n >= 0 && fmaxf(a3, umin) <= b3;
        // ---- leaf children in slot order: every interested lane filters the (broadcast) triangles with its own shaft.  Four copies of
        //      the filter loop (one per slot, everything static) rather than one loop over a slot index: selecting a slot's count, link
        //      and lane mask by a run-time index costs a chain of scalar branches per slot and step ----
        const auto leaf = [&](const int cn, const int cc, const bool hc) __attribute__((always_inline)) {
            if (__ballot(hc && !done) == 0ull) return;
            leaves++;
            slabs += (uint32_t)cn;
            if (PERSIST) walk += (uint32_t)cn;
            for (int q = 0; q < cn; ++q) {
                const TriSlab s = load_uniform(&sc_arg.bslab[cc + q]); // scalar load
                const bool live_q = hc && !done;
                const int touch = shaft_touches<true>(s, sr, live_q);
                const bool take = live_q && touch != 0, room = count < cap;
                if (take && room) out[count] = cc + q;
                count += (take && room) ? 1 : 0;
                truncated = truncated || (take && !room);
                umbra = umbra || (live_q && touch == 2);
                done = done || truncated || umbra;
            }
        };
        if (n.ch[0].n > 0) leaf(n.ch[0].n, n.ch[0].c, h0);
        if (n.ch[1].n > 0) leaf(n.ch[1].n, n.ch[1].c, h1);
        if (n.ch[2].n > 0) leaf(n.ch[2].n, n.ch[2].c, h2);
        if (n.ch[3].n > 0) leaf(n.ch[3].n, n.ch[3].c, h3);
        // ---- inner children, far to near: the nearest one a live lane wants is entered, the others wait on the stack ----
        int32_t next = -1;
        float next_u = -1.0f;
        const auto push = [&](const int32_t pn, const float pu) __attribute__((always_inline)) {
            if (has_top) { wnode[sp] = top_ni; wbound[sp * 64] = (uint16_t)top_b; sp++; }      // the old top moves to LDS
            top_ni = pn; top_b = (uint32_t)enc(pu); has_top = true;
        };
        {
            const bool w3 = h3 && !done && n.ch[3].n == 0;
            if (__ballot(w3) != 0ull) { next = n.ch[3].c; next_u = w3 ? b3 : -1.0f; }
            const bool w2 = h2 && !done && n.ch[2].n == 0;
            if (__ballot(w2) != 0ull) {
                if (next >= 0) push(next, next_u);
                next = n.ch[2].c; next_u = w2 ? b2 : -1.0f;
            }
            const bool w1 = h1 && !done && n.ch[1].n == 0;
            if (__ballot(w1) != 0ull) {
                if (next >= 0) push(next, next_u);
                next = n.ch[1].c; next_u = w1 ? b1 : -1.0f;
            }
            const bool w0 = h0 && !done && n.ch[0].n == 0;
            if (__ballot(w0) != 0ull) {
                if (next >= 0) push(next, next_u);
                next = n.ch[0].c; next_u = w0 ? b0 : -1.0f;
            }
        }
        if (next >= 0) { ni = next; nu = next_u; }
        else nu = -1.0f;                                               // leaf-only / dead end: pop at the top of the loop
    }
    if (valid) {
        const DevScene sc = tile_arg<PERSIST>(sc_arg, offsetof(SceneFrameArgs, sc));      // (epilogue: shadow cache, extra geometry count)
        const FrameConst fc = tile_arg<PERSIST>(fc_arg, offsetof(SceneFrameArgs, fc));  // (epilogue: flags, sample count)
        if (umbra) {
            // fully shadowed: rayEscapeCount = 0 -> (byte)(0.0 * 255) = 0 -> ModulatePackedColor(color, 0) = opaque black
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], 0.0);
            cand_count[slot_i] = 0u;
        } else if (count == 0 && !truncated && sc.nextra == 0) {
            // nothing in the whole shaft and no extra geometry: every sample escapes (ShadowMethod.cs:113-119)
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], (double)fc.shadow_samples / (double)fc.shadow_samples);
            cand_count[slot_i] = 0u;
        } else {
            cand_count[slot_i] = (unsigned)count | (truncated ? kTruncated : 0u);
            work_list[atomicAdd(work_count, 1u)] = slot_i;             // the compiler aggregates this per wavefront
        }
    }
    if (STATS) points += (uint32_t)__popcll(__ballot(valid));
    if (PERSIST && tile_cost && lane == 0) tile_cost[feed.item] = walk;      // (the next frame walks the longest tiles first: k_tile_order)
    }   // while (feed.next)
    if (STATS) block_stat_add(&stats[6], &stats[7], &stats[10], &stats[11], nodes, leaves, slabs, points);
    if (STATS && lane == 0) { atomicAdd(work_count - 2, top21); atomicAdd(work_count - 1, top85); }      // diagnostics: sr_debug_counters [4], [5] (free on the shaft path)
}

// k_tile_order -- one workgroup per XCD list: a STABLE partition of the list's items (natural ids 0 .. per_xcd - 1) into `classes` classes by the
// walk length k_shaft_pkt4 measured for them: the longest walks (>= quarters / 4 times the list's mean; 3 x by default) first, then the ones above
// half that bound, a quarter of it, ... and the rest last, each class in natural order -- neighbouring tiles share tree nodes and records in
// the XCD's L2, so the order inside a class is left alone.  A launch ends when its last tile does: with the long walks handed out first, the
// tiles that remain at the end are short ones (two classes: below 3 x the mean; four: below 0.75 x -- what a strip-interleaved part frame
// with its few tiles per workgroup needs).
constexpr int kOrderClasses = 4, kOrderWaves = 16;
__global__ __launch_bounds__(kOrderWaves * 64) void k_tile_order(const unsigned int* __restrict__ cost, unsigned int* __restrict__ order, unsigned int per_xcd, unsigned int quarters,
                                                    int classes) {
    __shared__ unsigned long long ssum;
    __shared__ unsigned int wcnt[kOrderWaves][kOrderClasses];
    const unsigned int x = blockIdx.x, tid = threadIdx.x, lane = tid & 63u, w = tid >> 6;
    const unsigned int* c = cost + (size_t)x * per_xcd;
    unsigned int* o = order + (size_t)x * per_xcd;
    if (tid == 0) ssum = 0ull;
    __syncthreads();
    unsigned long long sum = 0ull;
    for (unsigned int i = tid; i < per_xcd; i += kOrderWaves * 64u) sum += c[i];
    for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d, 64);
    if (lane == 0) atomicAdd(&ssum, sum);
    __syncthreads();
    const unsigned int thr = (unsigned int)min(0xfffffffeull, (unsigned long long)quarters * ssum / (4ull * (unsigned long long)max(1u, per_xcd))) + 1u;   // quarters / 4 x the mean
    // class of a walk length: 0 for >= thr, k for >= thr >> k, the last class for everything else
    auto cls_of = [&](unsigned int v) { int k = 0; while (k + 1 < classes && v < (thr >> k)) ++k; return k; };
    // wave w owns a contiguous segment of the list (coalesced batches of 64 items): count its classes, then place every item
    const unsigned int seg = ((per_xcd + kOrderWaves - 1u) / kOrderWaves + 63u) / 64u * 64u, i0 = min(per_xcd, w * seg), i1 = min(per_xcd, i0 + seg);
    unsigned int cnt[kOrderClasses] = {};
    for (unsigned int b = i0; b < i1; b += 256u) {                       // (four batches of 64 in flight: the loop is load latency)
        unsigned int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned int i = b + 64u * u + lane; v[u] = i < i1 ? c[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int i = b + 64u * u + lane;
            const int k = i < i1 ? cls_of(v[u]) : -1;
#pragma unroll
            for (int q = 0; q < kOrderClasses; ++q) cnt[q] += (unsigned int)__popcll(__ballot(k == q));
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < kOrderClasses; ++q) wcnt[w][q] = cnt[q];
    }
    __syncthreads();
    unsigned int pos[kOrderClasses], base = 0;
#pragma unroll
    for (int q = 0; q < kOrderClasses; ++q) {                            // class q starts after the classes before it; this wave's share after the earlier waves'
        unsigned int before = 0, all = 0;
        for (unsigned int k = 0; k < (unsigned int)kOrderWaves; ++k) { const unsigned int v = wcnt[k][q]; if (k < w) before += v; all += v; }
        pos[q] = base + before;
        base += all;
    }
    for (unsigned int b = i0; b < i1; b += 256u) {
        unsigned int v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const unsigned int i = b + 64u * u + lane; v[u] = i < i1 ? c[i] : 0u; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned int i = b + 64u * u + lane;
            const int k = i < i1 ? cls_of(v[u]) : -1;
            const unsigned long long lt = lanemask_lt();
#pragma unroll
            for (int q = 0; q < kOrderClasses; ++q) {
                const unsigned long long m = __ballot(k == q);
                if (k == q) o[pos[q] + (unsigned int)__popcll(m & lt)] = i;
                pos[q] += (unsigned int)__popcll(m);
            }
        }
    }
}

// undecided / escaped sample masks of a hit that moves on to the next round
struct alignas(16) RoundState {
    unsigned long long alive[kPacketSlots], escaped[kPacketSlots];
};

template <bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_shadow_test(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                     const HitRec* __restrict__ hits, const unsigned int* __restrict__ hit_count,
                                                     unsigned int count_cap, const unsigned int* __restrict__ index_list,
                                                     const RoundState* __restrict__ state_in, int cap, int lists_by_hit,
                                                     const unsigned int* __restrict__ cand_count, const int32_t* __restrict__ cand,
                                                     unsigned int* __restrict__ next_count, unsigned int next_cap,
                                                     unsigned int* __restrict__ next_list, RoundState* __restrict__ state_out,
                                                     unsigned int* __restrict__ last_count, unsigned int* __restrict__ last_list,
                                                     RoundState* __restrict__ last_state,
                                                     uint32_t* __restrict__ samples, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // per-wave LDS: kRecordsPerPass records of 128 B, then kTailSlots compacted sample rays of 64 B
    uint4* wrec = reinterpret_cast<uint4*>(lds_pipe) + (size_t)wave * (kRecordsPerPass * kRecStride16 + kTailSlots * kRayStride8 / 2);
    const double* wrecd = reinterpret_cast<const double*>(wrec);
    double* wray = reinterpret_cast<double*>(wrec + kRecordsPerPass * kRecStride16);

    const int S = fc.shadow_samples;
    const unsigned int total = min(*hit_count, count_cap);
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    D3 off[kPacketSlots];
    bool valid[kPacketSlots];
#pragma unroll
    for (int k = 0; k < kPacketSlots; ++k) {
        int j = lane + 64 * k;
        valid[k] = j < S;
        int jj = valid[k] ? j : 0;
        off[k] = mk(offsets[3 * jj], offsets[3 * jj + 1], offsets[3 * jj + 2]);
    }
    Ctr sec = {0, 0, 0, 0};
    const unsigned int nwaves = gridDim.x * 4u;
    const unsigned int s0 = blockIdx.x * 4u + (unsigned)wave;
    const int pre = min(cap, 64);                  // list entries kept in registers: lane l holds entry l
    // software pipeline over this wave's items: the hit index is fetched two items ahead, the hit record / list
    // head one item ahead, the first chunk's triangle records before the (long) per-sample clip of the current item
    unsigned int h_cur = 0, h_nxt = 0;
    HitRec rec_cur;
    unsigned int cc_cur = 0;
    int32_t ent_cur = 0;
    if (s0 < total) {
        h_cur = index_list ? index_list[s0] : s0;
        rec_cur = hits[h_cur];
        const size_t li0 = lists_by_hit ? (size_t)h_cur : (size_t)s0;
        cc_cur = cand_count[li0];
        ent_cur = lane < pre ? cand[li0 * cap + lane] : 0;
    }
    if (s0 + nwaves < total) h_nxt = index_list ? index_list[s0 + nwaves] : s0 + nwaves;
    for (unsigned int slot_i = s0; slot_i < total; slot_i += nwaves) {
        const unsigned int h = h_cur;
        const HitRec rec = rec_cur;
        const unsigned int cc = cc_cur;
        const int32_t ent = ent_cur;
        sec.leaves++;
        const D3 E = mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001;
        const size_t li = lists_by_hit ? (size_t)h : (size_t)slot_i;   // round 0: lists are stored per hit, later rounds per item
        const int ntri = (int)(cc & 0xffffu);
        const bool truncated = (cc & kTruncated) != 0;
        const int32_t* list = cand + li * cap;
        const bool work = ntri > 0 || truncated;      // an empty, complete list: every sample escapes, no clipping needed
        // ---- (a) issue the record loads of the first chunk: lane -> (record slot, 16-byte piece), coalesced dwordx4 ----
        uint4 r0[kRecordsPerPass / 8];
        const int n0 = min(kRecordsPerPass, ntri);
#pragma unroll
        for (int pass = 0; pass < kRecordsPerPass / 8; ++pass) {
            const int slot = pass * 8 + (lane >> 3);
            const int32_t e = __shfl(ent, slot, 64);
            r0[pass] = make_uint4(0, 0, 0, 0);
            if (slot < n0) r0[pass] = reinterpret_cast<const uint4*>(&sc.btris[e])[lane & 7];
        }
        // the pixel's shaded colour is needed only when the item is finished: fetch it now, off the critical path
        uint32_t shaded = 0;
        if (lane == 0 && !(fc.flags & 32u)) shaded = samples[rec.sample];
        // ---- (b) prefetch the next item ----
        {
            const unsigned int sn = slot_i + nwaves, snn = sn + nwaves;
            h_cur = h_nxt;
            if (sn < total) {
                rec_cur = hits[h_cur];
                const size_t lin = lists_by_hit ? (size_t)h_cur : (size_t)sn;
                cc_cur = cand_count[lin];
                ent_cur = lane < pre ? cand[lin * cap + lane] : 0;
            }
            if (snn < total) h_nxt = index_list ? index_list[snn] : snn;
        }
        // ---- (c) per-sample rays: clip + rayFracOffset ----
        SampleRay ray[kPacketSlots];
        bool alive[kPacketSlots], escaped[kPacketSlots];
#pragma unroll
        for (int k = 0; k < kPacketSlots; ++k) {
            D3 rs = lpos + off[k];
            D3 rd = E - rs;
            alive[k] = false;
            escaped[k] = valid[k];
            if (state_in) {                                               // later round: resume from the saved masks
                const RoundState stt = state_in[slot_i];
                // (an undecided sample carries its escaped bit; an umbra triangle found by this round's walk blocks them all)
                escaped[k] = ((stt.escaped[k] >> lane) & 1ull) != 0 && !(cc & kUmbraItem);
                if (work && ((stt.alive[k] >> lane) & 1ull) != 0) alive[k] = prepare_sample(sc, rs, rd, ray[k]);
            } else if (valid[k]) {
                sec.rays++;
                bool blocked = false;
                if (EXTRA) blocked = extras_block<EXTRA>(sc, rs, rd, sec);
                if (blocked) escaped[k] = false;
                else if (work) alive[k] = prepare_sample(sc, rs, rd, ray[k]);   // outside the root box: nothing can block it
            }
        }
        bool have = __any(alive[0] || alive[1]);
        // rcur: the chunk about to be staged (the first one was requested before the clip); while a chunk is being
        // tested, the next chunk's records are already on their way (requested right after the LDS write)
        uint4 rcur[kRecordsPerPass / 8];
#pragma unroll
        for (int pass = 0; pass < kRecordsPerPass / 8; ++pass) rcur[pass] = r0[pass];
        bool tail = false;                         // wave-uniform
        int tail_sh = 5;                           // log2 of the sample-slot count of a pass: 8, 16, 32 or 64
        unsigned long long tail_alive = 0ull, tail_blocked = 0ull;   // per compacted sample slot, wave-uniform
        int tail_idx[kPacketSlots] = {0, 0};       // this lane's samples' slots
        for (int base = 0; base < ntri && have; base += kRecordsPerPass) {
            const int npass = min(kRecordsPerPass, ntri - base);
            sec.nodes += (uint32_t)npass;
            // ---- (d) stage up to 16 records (2 KB) through LDS ----
#pragma unroll
            for (int pass = 0; pass < kRecordsPerPass / 8; ++pass) {
                const int slot = pass * 8 + (lane >> 3);
                if (slot < npass) wrec[slot * kRecStride16 + (lane & 7)] = rcur[pass];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // ---- request the next chunk ----
            const int nbase = base + kRecordsPerPass;
            const int nnext = min(kRecordsPerPass, ntri - nbase);
#pragma unroll
            for (int pass = 0; pass < kRecordsPerPass / 8; ++pass) {
                const int slot = pass * 8 + (lane >> 3);
                const int32_t es = __shfl(ent, (nbase + slot) & 63, 64);      // all lanes take part in the shuffle
                if (slot < nnext) {
                    const int32_t e = (nbase + kRecordsPerPass <= pre) ? es : list[nbase + slot];
                    rcur[pass] = reinterpret_cast<const uint4*>(&sc.btris[e])[lane & 7];
                }
            }
            // ---- (e) every lane tests its undecided samples against the staged records (exact FP64) ----
            int k = 0;
            if (!tail) {
                for (; k < npass && have; ++k) {
                    const double* p = wrecd + (size_t)k * (2 * kRecStride16);
#pragma unroll
                    for (int q = 0; q < kPacketSlots; ++q) {
                        if (alive[q]) {
                            sec.geom++;
                            if (tri_blocks(p, ray[q], sc.root.lo, sc.root.hi)) { alive[q] = false; escaped[q] = false; }
                        }
                    }
                    {
                        const unsigned long long m0 = __ballot(alive[0]), m1 = __ballot(alive[1]);
                        have = (m0 | m1) != 0ull;
                        const int na = (int)__popcll(m0) + (int)__popcll(m1);
                        if (have && na <= kTailSlots) {
                            // ---- few samples left: compact their rays into LDS; from here on a lane is one
                            //      (sample, candidate) pair, so a pass tests 64 / tail_w candidates at once ----
                            tail = true;
                            tail_sh = na <= 8 ? 3 : (na <= 16 ? 4 : (na <= 32 ? 5 : 6));
                            tail_alive = na >= 64 ? ~0ull : ((1ull << na) - 1ull);
                            tail_blocked = 0ull;
                            tail_idx[0] = (int)__popcll(m0 & lanemask_lt());
                            tail_idx[1] = (int)__popcll(m0) + (int)__popcll(m1 & lanemask_lt());
#pragma unroll
                            for (int q = 0; q < kPacketSlots; ++q) {
                                if (alive[q]) {
                                    double* w = wray + (size_t)tail_idx[q] * kRayStride8;
                                    w[0] = ray[q].s.x; w[1] = ray[q].s.y; w[2] = ray[q].s.z;
                                    w[3] = ray[q].d.x; w[4] = ray[q].d.y; w[5] = ray[q].d.z;
                                    w[6] = ray[q].offset;
                                }
                            }
                            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                            __builtin_amdgcn_wave_barrier();
                            ++k;
                            break;
                        }
                    }
                }
            }
            if (tail) {
                while (k < npass && tail_alive != 0ull) {
                    const int tail_w = 1 << tail_sh, per = 64 >> tail_sh;    // sample slots, candidates per pass
                    const int a = lane & (tail_w - 1), c = k + (lane >> tail_sh);
                    const bool act = c < npass && ((tail_alive >> a) & 1ull) != 0ull;
                    bool blk = false;
                    if (act) {
                        const double* w = wray + (size_t)a * kRayStride8;
                        SampleRay r;
                        r.s = mk(w[0], w[1], w[2]);
                        r.d = mk(w[3], w[4], w[5]);
                        r.offset = w[6];
                        sec.geom++;
                        blk = tri_blocks(wrecd + (size_t)c * (2 * kRecStride16), r, sc.root.lo, sc.root.hi);
                    }
                    unsigned long long m = __ballot(blk);
                    // fold the per-pair results onto the sample slots (low tail_w bits)
                    if (tail_sh <= 5) m |= m >> 32;
                    if (tail_sh <= 4) m |= m >> 16;
                    if (tail_sh <= 3) m |= m >> 8;
                    const unsigned long long nb = m & tail_alive;
                    tail_alive &= ~nb;
                    tail_blocked |= nb;
                    k += per;
                    // ---- at most half of the slots still undecided: pack them again, twice the candidates per pass ----
                    const int left = (int)__popcll(tail_alive);
                    if (tail_sh > 3 && left > 0 && left <= (tail_w >> 1) && (k < npass || base + kRecordsPerPass < ntri)) {
#pragma unroll
                        for (int q = 0; q < kPacketSlots; ++q) {             // owners take the verdicts so far, then renumber
                            if (alive[q]) {
                                if (((tail_blocked >> tail_idx[q]) & 1ull) != 0ull) { alive[q] = false; escaped[q] = false; }
                                else tail_idx[q] = (int)__popcll(tail_alive & ((1ull << tail_idx[q]) - 1ull));
                            }
                        }
                        const bool mine = lane < tail_w && ((tail_alive >> lane) & 1ull) != 0ull;   // lane = old slot
                        double t[7];
                        if (mine) {
#pragma unroll
                            for (int i = 0; i < 7; ++i) t[i] = wray[(size_t)lane * kRayStride8 + i];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        if (mine) {
                            double* w = wray + (size_t)__popcll(tail_alive & lanemask_lt()) * kRayStride8;
#pragma unroll
                            for (int i = 0; i < 7; ++i) w[i] = t[i];
                        }
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                        tail_alive = (1ull << left) - 1ull;
                        tail_blocked = 0ull;
                        tail_sh = left <= 8 ? 3 : (left <= 16 ? 4 : 5);
                    }
                }
                have = tail_alive != 0ull;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (tail) {
            // hand the tail's verdicts back to the lanes that own the samples
#pragma unroll
            for (int q = 0; q < kPacketSlots; ++q) {
                if (alive[q] && ((tail_blocked >> tail_idx[q]) & 1ull) != 0ull) { alive[q] = false; escaped[q] = false; }
            }
            have = __any(alive[0] || alive[1]);
        }
        if (have && truncated) {
            // the list ran out before the shaft did and some sample is still undecided: next round (longer list),
            // or -- after the last round / when the next round's buffers are full -- the exact per-lane fallback
            const unsigned long long a0 = __ballot(alive[0]), a1 = __ballot(alive[1]), e0 = __ballot(escaped[0]), e1 = __ballot(escaped[1]);
            if (lane == 0) {
                unsigned int slot = next_count ? atomicAdd(next_count, 1u) : 0xffffffffu;
                RoundState o;
                o.alive[0] = a0; o.alive[1] = a1; o.escaped[0] = e0; o.escaped[1] = e1;
                if (slot < next_cap) {
                    next_list[slot] = h;
                    state_out[slot] = o;
                } else {
                    const unsigned int fi = atomicAdd(last_count, 1u);    // the fallback only traces the undecided samples
                    last_list[fi] = h;
                    last_state[fi] = o;
                }
            }
        } else {
            const int esc = (int)__popcll(__ballot(escaped[0])) + (int)__popcll(__ballot(escaped[1]));
            if (lane == 0) {
                double frac = (double)esc / (double)S;                     // ShadowMethod.IntersectRay :113-119
                finish_hit(sc, fc, samples, rec.sample, rec.pad[0], shaded, frac);
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b);
            stat_add(&stats[8], sec.nodes);          // wave-level: records staged through LDS
            stat_add(&stats[9], sec.leaves);         // wave-level: hit points processed
        }
    }
}

// --------------------------------------------------------------------------------------------------
// k_shadow_cls -- the default exact-test half of the shaft path: same items, same lists, same verdicts as k_shadow_test,
// but a (sample, triangle) pair is first CLASSIFIED in fp32 with a rigorous error bound; only the pairs the bound cannot
// decide (a crossing within ~1e-6 of a triangle edge, of the surface point, of the root box, or a grazing plane) run the
// reference's FP64 arithmetic (prepare_sample + tri_blocks, exactly as in k_shadow_test).
//
// Geometry, parametrised from the surface end like the shaft: sample ray i is X(u) = E' + u D_i, D_i = (L - E') + off_i,
// u in [0, 1] (u = 1 - rayFrac).  With the triangle's TriSlab planes (unit normals, root-centre-relative; the record
// k_shaft filtered with): plane function G(x) = n.x - d, edge functions F_k(x) = m_k.x - c_k (>= 0 inside).  Along the
// ray G = G0 + u g1_i with G0 = G(E'), g1_i = n.D_i, so the crossing is at u_c = -G0 / g1_i and there
//     F_k = K0_k + u_c (m_k.D_i) = (w_k . D_i) / g1_i,      w_k = K0_k n - G0 m_k,  K0_k = F_k(E')
// i.e. w_k is the normal of the plane through E' and edge k: the sample is inside the triangle's cone iff w_k.D_i >= 0
// for k = 1..3 -- no division, and w_k depends on (hit point, triangle) only.  Triangle.IntersectRay (Triangle.cs:83-104)
// + the tree's conditions (SpatialSubdivision.cs:394-401,652; ShadowMethod.cs:170) accept the crossing iff it is
// front-facing (dirDist = -g1_i < 0), at or behind the clipped start, rayFrac + offset <= 1.0 (u_c >= 0 <=> G0 <= 0), inside
// the root box and inside the three edges.
//
// Error bounds (u = 2^-24; every stored fp32 value is the rounding of its FP64 source; dot products are FMA chains):
//   |G0 - true|, |K0_k - true| <= 5u (|E'|_2 + |d|)  <  a0 := 12u s0,   s0 = 2 (half diagonal of the root box + 0.002)
//   |g1_i - true| <= 8u dmax  <  a1 := 20u dmax,   dmax = |L - E'|_2 + R;   "front-facing" needs g1_i >= glo := 16 a1
//   |c_k - true w_k.D_i| <= dmax u (10 s0 + 11 (|K0_k| + |G0|))  <  mc := dmax u (15 s0 + 16 (max_k |K0_k| + |G0|)) + 1e-9 dmax
//     (w_k inherits 2 a0' from G0 / K0_k and 3u (|K0_k| + |G0|) from its own rounding; the two-part dot product adds 8u |w_k| dmax)
// FP64 evaluation errors (~1e-15) are six orders of magnitude below these margins, so with g1_i >= glo:
//   BLOCKED  <=  G0 <= -a0 (u_c > 0)  and  min_k c_k > mc (inside every edge by > 1e-9)  and  G0 + umax_i g1_i > a0 + a1 (u_c < umax_i)
//   MISS     <=  min_k c_k < -mc (outside an edge)  or  G0 >= a0 (crossing behind the surface point);   also g1_i <= -glo (back-facing)
// umax_i <= 1 is a per-(hit point, sample) parameter at which X(u) is VERIFIED (position error bound pm) to lie inside the
// root box; E' is verified too, so by convexity every crossing with u_c < umax_i lies inside the box and strictly in front
// of the clipped start (SpatialSubdivision.cs:394).  Everything else is "uncertain" and decided by the FP64 code.  A
// degenerate triangle's TriSlab is all zeros: g1 = 0, always uncertain.  With g1_i >= glo every intermediate is finite.
// --------------------------------------------------------------------------------------------------
constexpr float kU24 = 5.9604645e-8f;    // 2^-24
constexpr int kClsCand = 32;             // candidates staged per chunk (5 float4 each): one bit each in a lane's "uncertain" masks
// LDS of one wave of k_shadow_cls, in float4 units: records + pair-transposed records + record indices + tail tables
constexpr int kClsWaveF4 = kClsCand * 5 + (kClsCand / 2) * 10 + kClsCand / 4 + 16 + 32 + 32 + 1;

struct ClsFrame {                         // per-frame fp32 constants of the classification
    float cx, cy, cz, hbx, hby, hbz, s0, a0, R;
};

__device__ __forceinline__ ClsFrame cls_frame(const DevScene& sc, const FrameConst& fc) {
    ClsFrame c;
    c.cx = (float)sc.root.centre[0]; c.cy = (float)sc.root.centre[1]; c.cz = (float)sc.root.centre[2];
    const float ex = (float)(sc.root.max[0] - sc.root.min[0]), ey = (float)(sc.root.max[1] - sc.root.min[1]), ez = (float)(sc.root.max[2] - sc.root.min[2]);
    c.hbx = 0.5f * ex * 0.9999999f; c.hby = 0.5f * ey * 0.9999999f; c.hbz = 0.5f * ez * 0.9999999f;   // rounded towards the centre
    c.s0 = (sqrtf(ex * ex + ey * ey + ez * ez) * 0.5f + 0.002f) * 2.002f;
    c.a0 = 12.0f * kU24 * c.s0;
    c.R = (float)fc.light_radius * 1.0001f + 1e-30f;
    return c;
}

// TAIL: compile the tail layout in.  It pays for the later rounds (few undecided samples from the start); in the first round the
// 30 registers it costs (162 instead of 128 VGPRs = 3 instead of 4 waves/SIMD) lose more than its shorter candidate loop gains.
template <bool EXTRA, bool STATS, bool TAIL>
__global__ __launch_bounds__(256) void k_shadow_cls(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                    const HitRec* __restrict__ hits, const unsigned int* __restrict__ hit_count,
                                                    unsigned int count_cap, const unsigned int* __restrict__ index_list,
                                                    const RoundState* __restrict__ state_in, int cap, int lists_by_hit,
                                                    const unsigned int* __restrict__ cand_count, const int32_t* __restrict__ cand,
                                                    unsigned int* __restrict__ next_count, unsigned int next_cap,
                                                    unsigned int* __restrict__ next_list, RoundState* __restrict__ state_out,
                                                    unsigned int* __restrict__ last_count, unsigned int* __restrict__ last_list,
                                                    RoundState* __restrict__ last_state,
                                                    uint32_t* __restrict__ samples, unsigned long long* stats) {
    // (the first round -- no tail layout -- is the one launched without saved masks and with lists stored per hit: compile-time facts)
    if (!TAIL) { state_in = nullptr; lists_by_hit = 1; }
    else lists_by_hit = 0;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    float4* wc = reinterpret_cast<float4*>(lds_pipe) + (size_t)wave * kClsWaveF4;   // per-candidate records (owner layout)
    float4* wpair = wc + kClsCand * 5;                                      // the same records transposed in pairs (tail layout)
    int32_t* wrecidx = reinterpret_cast<int32_t*>(wpair + (kClsCand / 2) * 10);   // record position of every staged candidate
    uint32_t* wids = reinterpret_cast<uint32_t*>(wrecidx + kClsCand);       // [64] sample of every tail slot
    float* wum = reinterpret_cast<float*>(wids + 64);                       // [128] umax of every sample of the current hit point
    uint32_t* wunc = reinterpret_cast<uint32_t*>(wum + 128);                // [128] undecided candidates of every sample (tail)
    uint32_t* wdead = wunc + 128;                                           // [4] samples the tail found blocked
    const int S = fc.shadow_samples;
    float* offtab = reinterpret_cast<float*>(reinterpret_cast<float4*>(lds_pipe) + 4 * kClsWaveF4);   // [128][3] area-light offsets, fp32
    // ShadowMethod.IntersectRay's byte for every possible rayEscapeCount (:113-119 / the static cache's :80), made once per
    // workgroup with the reference's FP64 expressions instead of one FP64 division per hit point
    uint32_t* light_byte = reinterpret_cast<uint32_t*>(offtab + 128 * 3);
    for (int e = tid; e < 128 * 3; e += 256) offtab[e] = e < 3 * S ? (float)offsets[e] : 0.0f;
    for (int e = tid; e <= S; e += 256) {
        const double frac = (double)e / (double)S;
        uint32_t v;
        if (fc.flags & 32u) { v = (uint32_t)(int)(frac * 254 + 1) & 0xffu; v = v ? v : 1u; }
        else v = to_byte(frac * 255);
        light_byte[e] = v;
    }
    __syncthreads();
    const unsigned int total = min(*hit_count, count_cap);
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const ClsFrame cf = cls_frame(sc, fc);
    bool valid[kPacketSlots];
    f2 OX, OY, OZ;                                                       // fp32 roundings of the lane's two offsets (the FP64 ones are fetched again by the rare exact path)
    {
        const int j1 = lane + 64 < S ? lane + 64 : 0, j0 = lane < S ? lane : 0;
        valid[0] = lane < S; valid[1] = lane + 64 < S;
        OX = (f2){(float)offsets[3 * j0], (float)offsets[3 * j1]};
        OY = (f2){(float)offsets[3 * j0 + 1], (float)offsets[3 * j1 + 1]};
        OZ = (f2){(float)offsets[3 * j0 + 2], (float)offsets[3 * j1 + 2]};
    }
    uint32_t n_rays = 0, n_items = 0, n_recs = 0, n_cls = 0, n_exact = 0;
    const unsigned int nwaves = gridDim.x * 4u;
    const unsigned int s0 = blockIdx.x * 4u + (unsigned)wave;
    // software pipeline over this wave's items: hit index two items ahead, hit record / list head one item ahead
    unsigned int h_cur = 0, h_nxt = 0;
    HitRec rec_cur;
    unsigned int cc_cur = 0;
    int32_t ent_cur = 0;
    if (s0 < total) {
        h_cur = index_list ? index_list[s0] : s0;
        rec_cur = hits[h_cur];
        const size_t li0 = lists_by_hit ? (size_t)h_cur : (size_t)s0;
        cc_cur = cand_count[li0];
        ent_cur = lane < cap ? cand[li0 * cap + lane] : 0;
    }
    if (s0 + nwaves < total) h_nxt = index_list ? index_list[s0 + nwaves] : s0 + nwaves;
    for (unsigned int slot_i = s0; slot_i < total; slot_i += nwaves) {
        const unsigned int h = h_cur;
        const unsigned int cc = cc_cur;
        int32_t ent = ent_cur;
        n_items++;
        const D3 E = mk(rec_cur.pos[0], rec_cur.pos[1], rec_cur.pos[2]) + mk(rec_cur.nrm[0], rec_cur.nrm[1], rec_cur.nrm[2]) * 0.001;
        const uint32_t rec_sample = rec_cur.sample, rec_cell = rec_cur.pad[0];
        const size_t li = lists_by_hit ? (size_t)h : (size_t)slot_i;   // round 0: lists are stored per hit, later rounds per item
        const int ntri = (int)(cc & 0xffffu);
        const bool truncated = (cc & kTruncated) != 0;
        const int32_t* list = cand + li * cap;
        const bool work = ntri > 0 || truncated;      // an empty, complete list: every sample escapes
        // ---- this lane's candidate: the fp32 record k_shaft filtered with ----
        TriSlab slab;
        if (lane < ntri) slab = sc.bslab[ent];
        uint32_t shaded = 0;
        if (lane == 0 && !(fc.flags & 32u)) shaded = samples[rec_sample];
        // ---- prefetch the next item ----
        {
            const unsigned int sn = slot_i + nwaves, snn = sn + nwaves;
            h_cur = h_nxt;
            if (sn < total) {
                rec_cur = hits[h_cur];
                const size_t lin = lists_by_hit ? (size_t)h_cur : (size_t)sn;
                cc_cur = cand_count[lin];
                ent_cur = lane < cap ? cand[lin * cap + lane] : 0;
            }
            if (snn < total) h_nxt = index_list ? index_list[snn] : snn;
        }
        // ---- per hit point: fp32 frame of the classification ----
        const D3 DLd = lpos - E;
        const float efx = (float)(E.x - sc.root.centre[0]), efy = (float)(E.y - sc.root.centre[1]), efz = (float)(E.z - sc.root.centre[2]);
        const float dlx = (float)DLd.x, dly = (float)DLd.y, dlz = (float)DLd.z;
        const float dmax = __builtin_amdgcn_sqrtf(dlx * dlx + dly * dly + dlz * dlz) * 1.0001f + cf.R;   // an upper bound is all it has to be
        const float a1 = 20.0f * kU24 * dmax, glo = 16.0f * a1;
        const float pm = 5e-7f * (cf.s0 + dmax);                        // position error bound of E' + u D_i in fp32 (>= 4u (|E'| + u |D|))
        // ---- per (hit point, sample): state + umax ----
        bool alive[kPacketSlots], escaped[kPacketSlots];
#pragma unroll
        for (int q = 0; q < kPacketSlots; ++q) {
            alive[q] = false;
            escaped[q] = valid[q];
            if (state_in) {                                               // later round: resume from the saved masks
                const RoundState stt = state_in[slot_i];
                escaped[q] = ((stt.escaped[q] >> lane) & 1ull) != 0 && !(cc & kUmbraItem);
                alive[q] = work && ((stt.alive[q] >> lane) & 1ull) != 0;
            } else if (valid[q]) {
                n_rays++;
                bool blocked = false;
                if (EXTRA) { Ctr cx = {0, 0, 0, 0}; const int sj = lane + 64 * q; const D3 rs = lpos + mk(offsets[3 * sj], offsets[3 * sj + 1], offsets[3 * sj + 2]); blocked = extras_block<EXTRA>(sc, rs, E - rs, cx); }
                if (blocked) escaped[q] = false;
                else alive[q] = work;
            }
        }
        // umax_i: the exit parameter of the ray from a box shrunk by 4 pm (any approximation will do: fast reciprocals), then
        // the point E' + umax_i D_i is VERIFIED to lie inside the root box by more than pm, and so is E'.  Packed over the
        // lane's two samples.
        f2 UM;
        {
            const float sx = cf.hbx - 4.0f * pm, sy = cf.hby - 4.0f * pm, sz = cf.hbz - 4.0f * pm;
            const bool ein = fabsf(efx) < sx && fabsf(efy) < sy && fabsf(efz) < sz;
            const f2 dxv = OX + splat(dlx), dyv = OY + splat(dly), dzv = OZ + splat(dlz);
            const f2 rx = {__builtin_amdgcn_rcpf(dxv.x), __builtin_amdgcn_rcpf(dxv.y)}, ry = {__builtin_amdgcn_rcpf(dyv.x), __builtin_amdgcn_rcpf(dyv.y)},
                     rz = {__builtin_amdgcn_rcpf(dzv.x), __builtin_amdgcn_rcpf(dzv.y)};
            const f2 tx = ((f2){__builtin_copysignf(sx, dxv.x), __builtin_copysignf(sx, dxv.y)} - splat(efx)) * rx;
            const f2 ty = ((f2){__builtin_copysignf(sy, dyv.x), __builtin_copysignf(sy, dyv.y)} - splat(efy)) * ry;
            const f2 tz = ((f2){__builtin_copysignf(sz, dzv.x), __builtin_copysignf(sz, dzv.y)} - splat(efz)) * rz;
            const f2 ut = {fminf(fminf(fminf(tx.x, ty.x), tz.x), 0.999999f), fminf(fminf(fminf(tx.y, ty.y), tz.y), 0.999999f)};
            const f2 px = pk_fma(ut, dxv, splat(efx)), py = pk_fma(ut, dyv, splat(efy)), pz = pk_fma(ut, dzv, splat(efz));
            const bool ok0 = ein && ut.x > 0.0f && fabsf(px.x) < cf.hbx - pm && fabsf(py.x) < cf.hby - pm && fabsf(pz.x) < cf.hbz - pm;
            const bool ok1 = ein && ut.y > 0.0f && fabsf(px.y) < cf.hbx - pm && fabsf(py.y) < cf.hby - pm && fabsf(pz.y) < cf.hbz - pm;
            UM = (f2){ok0 ? ut.x : -1.0f, ok1 ? ut.y : -1.0f};
        }
        const float hbm = cf.a0 + a1;
        if (TAIL) { wum[lane] = UM.x; wum[lane + 64] = UM.y; }            // (the tail layout fetches a sample's umax by its index)
        bool have = __any(alive[0] || alive[1]);
        for (int base = 0; base < ntri && have; base += kClsCand) {
            const int nc = min(kClsCand, ntri - base);
            if (base > 0) {                                               // lists longer than one chunk
                ent = lane < nc ? list[base + lane] : 0;
                if (lane < nc) slab = sc.bslab[ent];
            }
            // ---- per (hit point, candidate) constants, lane = candidate: cone planes through E' and their margins ----
            if (lane < nc) {
                const f2 edx = {efx, dlx}, edy = {efy, dly}, edz = {efz, dlz};
                const f2 cn = {-slab.d, 0.0f}, c1 = {-slab.c1, 0.0f}, c2 = {-slab.c2, 0.0f}, c3 = {-slab.c3, 0.0f};
                const f2 N = pk_fma(splat(slab.n[0]), edx, pk_fma(splat(slab.n[1]), edy, pk_fma(splat(slab.n[2]), edz, cn)));    // (G0, n.(L - E'))
                const f2 P = pk_fma(splat(slab.m1[0]), edx, pk_fma(splat(slab.m1[1]), edy, pk_fma(splat(slab.m1[2]), edz, c1)));  // (K0_1, .)
                const f2 Q = pk_fma(splat(slab.m2[0]), edx, pk_fma(splat(slab.m2[1]), edy, pk_fma(splat(slab.m2[2]), edz, c2)));
                const f2 T = pk_fma(splat(slab.m3[0]), edx, pk_fma(splat(slab.m3[1]), edy, pk_fma(splat(slab.m3[2]), edz, c3)));
                const float G0 = N.x;
                float4 W[3];
                const float K0[3] = {P.x, Q.x, T.x};
                const float* mm[3] = {slab.m1, slab.m2, slab.m3};
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const float wx = __builtin_fmaf(K0[e], slab.n[0], -(G0 * mm[e][0]));
                    const float wy = __builtin_fmaf(K0[e], slab.n[1], -(G0 * mm[e][1]));
                    const float wz = __builtin_fmaf(K0[e], slab.n[2], -(G0 * mm[e][2]));
                    W[e] = make_float4(wx, wy, wz, __builtin_fmaf(wx, dlx, __builtin_fmaf(wy, dly, wz * dlz)));
                }
                const float kmax = fmaxf(fmaxf(fabsf(P.x), fabsf(Q.x)), fabsf(T.x));
                const float mc = dmax * (kU24 * (15.0f * cf.s0 + 16.0f * (kmax + fabsf(G0))) + 1e-9f);
                float4* w = wc + lane * 5;
                // (constant term first: it is the packed FMAs' addend, broadcast from the low register of an aligned pair)
                w[0] = make_float4(N.y, slab.n[0], slab.n[1], slab.n[2]);
                w[1] = make_float4(W[0].w, W[0].x, W[0].y, W[0].z);
                w[2] = make_float4(W[1].w, W[1].x, W[1].y, W[1].z);
                w[3] = make_float4(W[2].w, W[2].x, W[2].y, W[2].z);
                // .y: margin of "inside every edge" (unreachable unless E' is definitely behind the plane);
                // .z: margin of "outside an edge" (always reached when E' is definitely in front of the plane)
                w[4] = make_float4(G0, G0 <= -cf.a0 ? mc : 1e30f, G0 >= cf.a0 ? -1e30f : mc, 0.0f);
                wrecidx[lane] = ent;
                // the same 20 floats once more, interleaved with the neighbour candidate's (pair p = lane / 2: float 2 f + (lane & 1)),
                // so that the tail layout reads both candidates of a pair as packed operands
                if (TAIL) {
                float* pr = reinterpret_cast<float*>(wpair + (lane >> 1) * 10) + (lane & 1);
                const float rec20[20] = {slab.n[0], slab.n[1], slab.n[2], N.y, W[0].x, W[0].y, W[0].z, W[0].w, W[1].x, W[1].y, W[1].z, W[1].w,
                                         W[2].x, W[2].y, W[2].z, W[2].w, G0, G0 <= -cf.a0 ? mc : 1e30f, G0 >= cf.a0 ? -1e30f : mc, 0.0f};
#pragma unroll
                for (int e = 0; e < 20; ++e) pr[2 * e] = rec20[e];
                }
            } else if (TAIL && lane == nc && (nc & 1)) {
                // odd count: the last pair's second candidate is a null record -- in front of the surface point, "decided: miss" for every sample
                float* pr = reinterpret_cast<float*>(wpair + (lane >> 1) * 10) + 1;
#pragma unroll
                for (int e = 0; e < 20; ++e) pr[2 * e] = e == 3 ? 1.0f : (e == 16 ? 1.0f : (e == 17 ? 1e30f : (e == 18 ? -1e30f : 0.0f)));
            }
            if (TAIL) {
                wunc[lane] = 0u; wunc[lane + 64] = 0u;
                if (lane < 4) wdead[lane] = 0u;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            n_recs += (uint32_t)nc;
            // ---- fp32 classification of every (sample, candidate) pair.  tblk > 0: BLOCKED; tall > 0: decided (BLOCKED or
            //      MISS); bit k of unc[q]: pair (sample q, candidate k) is undecided ----
            uint32_t unc[kPacketSlots] = {0u, 0u};
            // (wave-uniform masks instead of per-lane running maxima: the bookkeeping of "still undecided" is scalar work)
            const unsigned long long al0 = __ballot(alive[0]), al1 = __ballot(alive[1]);
            unsigned long long blk0 = 0ull, blk1 = 0ull;                  // samples some candidate of this chunk blocks
            int k = 0;
            unsigned long long am0 = 0ull, am1 = 0ull;                    // samples still undecided
            int na = 128;
            // (the record of candidate k + 1 is requested before candidate k is worked on: an LDS round trip per candidate would
            //  otherwise be waited for in full; the slot after the chunk's last record is readable LDS whose content is not used)
            typedef float f4 __attribute__((ext_vector_type(4)));
            const f4* wc4 = reinterpret_cast<const f4*>(wc);
            f4 nA = wc4[0], nB1 = wc4[1], nB2 = wc4[2], nB3 = wc4[3], nF = wc4[4];
            for (; k < nc && have; ++k) {
                if (TAIL && !(k & 1) && na <= 64) break;                  // few samples left: the tail layout takes the rest of the chunk
                const f4 A = nA, B1 = nB1, B2 = nB2, B3 = nB3, F = nF;
                nA = wc4[k * 5 + 5]; nB1 = wc4[k * 5 + 6]; nB2 = wc4[k * 5 + 7]; nB3 = wc4[k * 5 + 8]; nF = wc4[k * 5 + 9];
                // packed over the lane's two samples (.x = sample lane, .y = sample lane + 64); records: (constant, x, y, z)
                // value = x * OX + (y * OY + (z * OZ + constant)), the scalars broadcast by the FMAs' operand selects
                const auto plane = [&](const f4& r) {
                    const f2 cx = {r.x, r.y}, yz = {r.z, r.w};
                    return pk_fma_hi(cx, OX, pk_fma_lo(yz, OY, pk_fma_hi_addlo(yz, OZ, cx)));
                };
                const f2 g1 = plane(A), c1 = plane(B1), c2 = plane(B2), c3 = plane(B3);
                const f2 hb = pk_fma(UM, g1, splat(F.x));                 // G at u = umax_i: > 0 <=> the crossing comes earlier
                const f2 cmin = {fminf(fminf(c1.x, c2.x), c3.x), fminf(fminf(c1.y, c2.y), c3.y)};
                const f2 s3 = g1 - splat(glo), s1 = cmin - splat(F.y), s4 = hb - splat(hbm);
                const f2 tm0 = splat(-F.z) - cmin, bfv = splat(-glo) - g1;
                const float tblk0 = fminf(fminf(s1.x, s3.x), s4.x), tblk1 = fminf(fminf(s1.y, s3.y), s4.y);
                const float tall0 = fmaxf(fmaxf(fminf(tm0.x, s3.x), bfv.x), tblk0), tall1 = fmaxf(fmaxf(fminf(tm0.y, s3.y), bfv.y), tblk1);
                // (shifted in from the right: compare + add-with-carry; put into candidate order after the loop)
                unc[0] = shift_in_not_positive(unc[0], tall0);
                unc[1] = shift_in_not_positive(unc[1], tall1);
                blk0 |= __ballot(tblk0 > 0.0f);
                blk1 |= __ballot(tblk1 > 0.0f);
                am0 = al0 & ~blk0; am1 = al1 & ~blk1;
                if (STATS) n_cls += (uint32_t)((am0 >> lane) & 1ull) + (uint32_t)((am1 >> lane) & 1ull);
                na = (int)__popcll(am0) + (int)__popcll(am1);
                have = na != 0;
                // (the requested record must have arrived HERE, not at the top of the next iteration where the compiler would
                //  otherwise move the request to)
                asm volatile("" : "+v"(nA), "+v"(nB1), "+v"(nB2), "+v"(nB3), "+v"(nF));
            }
            if (k > 0) { unc[0] = __brev(unc[0]) >> (32 - k); unc[1] = __brev(unc[1]) >> (32 - k); }   // bit j = candidate j
            // bsum[q] > 0: some candidate blocked the sample (dead samples ran through the arithmetic too: nothing they produced is looked at)
            float bsum[kPacketSlots] = {((blk0 >> lane) & 1ull) != 0ull ? 1.0f : -1.0f, ((blk1 >> lane) & 1ull) != 0ull ? 1.0f : -1.0f};
            if (k == 0 && have) {                                         // (entered with few samples: later rounds, later chunks)
                am0 = __ballot(alive[0]); am1 = __ballot(alive[1]);
                na = (int)__popcll(am0) + (int)__popcll(am1);
            }
            // ---- tail layout: the na <= 64 undecided samples are compacted into sample slots; a lane is one (slot, candidate PAIR),
            //      so a pass tests 2 * 64 / W candidates against every remaining sample (W = 8, 16, 32 or 64 slots).  Same arithmetic,
            //      packed over the two candidates of a pair instead of over two samples ----
            if (TAIL && have && k < nc && na <= 64) {
                const bool mine0 = alive[0] && !(bsum[0] > 0.0f), mine1 = alive[1] && !(bsum[1] > 0.0f);
                if (mine0) wids[__popcll(am0 & lanemask_lt())] = (uint32_t)lane;
                if (mine1) wids[__popcll(am0) + __popcll(am1 & lanemask_lt())] = (uint32_t)lane + 64u;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                unsigned long long sm = na >= 64 ? ~0ull : ((1ull << na) - 1ull);   // (wave-uniform) live slots
                for (;;) {
                    const int nslot = (int)__popcll(sm);                  // == highest live slot + 1 right after a compaction
                    const int sh = nslot <= 8 ? 3 : (nslot <= 16 ? 4 : (nslot <= 32 ? 5 : 6));
                    const int W = 1 << sh, a = lane & (W - 1), j = lane >> sh, groups = 64 >> sh;
                    const bool slot_live = ((sm >> a) & 1ull) != 0ull;
                    const uint32_t sid = slot_live ? wids[a] : 0u;
                    const float ox = offtab[3 * sid], oy = offtab[3 * sid + 1], oz = offtab[3 * sid + 2], um = wum[sid];
                    const f2 sox = splat(ox), soy = splat(oy), soz = splat(oz);
                    unsigned long long dead = 0ull;
                    bool recompact = false;
                    while (k < nc && sm != 0ull) {
                        const int p = (k >> 1) + j;                       // this lane's candidate pair (2 p, 2 p + 1)
                        const bool act = slot_live && ((sm >> a) & 1ull) != 0ull && 2 * p < nc;
                        const float4* pq = wpair + (act ? p : 0) * 10;
                        const float4 q0 = pq[0], q1 = pq[1], q2 = pq[2], q3 = pq[3], q4 = pq[4], q5 = pq[5], q6 = pq[6], q7 = pq[7], q8 = pq[8], q9 = pq[9];
                        const f2 g1 = pk_fma((f2){q0.x, q0.y}, sox, pk_fma((f2){q0.z, q0.w}, soy, pk_fma((f2){q1.x, q1.y}, soz, (f2){q1.z, q1.w})));
                        const f2 c1 = pk_fma((f2){q2.x, q2.y}, sox, pk_fma((f2){q2.z, q2.w}, soy, pk_fma((f2){q3.x, q3.y}, soz, (f2){q3.z, q3.w})));
                        const f2 c2 = pk_fma((f2){q4.x, q4.y}, sox, pk_fma((f2){q4.z, q4.w}, soy, pk_fma((f2){q5.x, q5.y}, soz, (f2){q5.z, q5.w})));
                        const f2 c3 = pk_fma((f2){q6.x, q6.y}, sox, pk_fma((f2){q6.z, q6.w}, soy, pk_fma((f2){q7.x, q7.y}, soz, (f2){q7.z, q7.w})));
                        const f2 hb = pk_fma(splat(um), g1, (f2){q8.x, q8.y});
                        const f2 cmin = {fminf(fminf(c1.x, c2.x), c3.x), fminf(fminf(c1.y, c2.y), c3.y)};
                        const f2 s3 = g1 - splat(glo), s1 = cmin - (f2){q8.z, q8.w}, s4 = hb - splat(hbm);
                        const f2 tm0 = -(f2){q9.x, q9.y} - cmin, bfv = splat(-glo) - g1;
                        const float tb0 = fminf(fminf(s1.x, s3.x), s4.x), tb1 = fminf(fminf(s1.y, s3.y), s4.y);
                        const float ta0 = fmaxf(fmaxf(fminf(tm0.x, s3.x), bfv.x), tb0), ta1 = fmaxf(fmaxf(fminf(tm0.y, s3.y), bfv.y), tb1);
                        const bool blk = act && (tb0 > 0.0f || tb1 > 0.0f);
                        if (act && !blk) {                                // undecided pairs of a sample this pass does not block (rare)
                            const uint32_t ub = (ta0 > 0.0f ? 0u : (1u << (2 * p))) | (ta1 > 0.0f ? 0u : (2u << (2 * p)));
                            if (ub) atomicOr(&wunc[sid], ub);
                        }
                        if (STATS) n_cls += act ? 2u : 0u;
                        unsigned long long m = __ballot(blk);
                        if (sh <= 5) m |= m >> 32;
                        if (sh <= 4) m |= m >> 16;
                        if (sh <= 3) m |= m >> 8;
                        const unsigned long long nb = m & sm;
                        sm &= ~nb;
                        dead |= nb;
                        k += 2 * groups;
                        if (sh > 3 && sm != 0ull && (int)__popcll(sm) <= (W >> 1) && k < nc) { recompact = true; break; }
                    }
                    // the samples this epoch found blocked, for their owner lanes
                    if (j == 0 && ((dead >> a) & 1ull) != 0ull) atomicOr(&wdead[sid >> 5], 1u << (sid & 31u));
                    if (!recompact) break;
                    // pack the surviving slots again: twice the candidates per pass
                    const bool keep = j == 0 && ((sm >> a) & 1ull) != 0ull;
                    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if (keep) wids[__popcll(sm & ((1ull << a) - 1ull))] = sid;
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    sm = (1ull << __popcll(sm)) - 1ull;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // owners take the tail's verdicts: blocked samples, undecided pairs
#pragma unroll
                for (int q = 0; q < kPacketSlots; ++q) {
                    const uint32_t sidq = (uint32_t)lane + 64u * q;
                    if (((wdead[sidq >> 5] >> (sidq & 31u)) & 1u) != 0u) bsum[q] = 1.0f;
                    unc[q] |= wunc[sidq];
                }
                have = __any((alive[0] && !(bsum[0] > 0.0f)) || (alive[1] && !(bsum[1] > 0.0f)));
            }
#pragma unroll
            for (int q = 0; q < kPacketSlots; ++q) {
                if (alive[q] && bsum[q] > 0.0f) { alive[q] = false; escaped[q] = false; }
            }
            // ---- the undecided pairs of the samples no candidate blocked so far: the reference's FP64 arithmetic, one
            //      copy of the code for both sample slots (rare: a few percent of the items have any) ----
            if (__any((alive[0] && unc[0] != 0u) || (alive[1] && unc[1] != 0u))) {
#pragma unroll 1
                for (int q = 0; q < kPacketSlots; ++q) {
                    uint32_t m = (q ? alive[1] : alive[0]) ? (q ? unc[1] : unc[0]) : 0u;
                    if (m) {
                        const int sj = lane + 64 * q;
                        const D3 rs = lpos + mk(offsets[3 * sj], offsets[3 * sj + 1], offsets[3 * sj + 2]);
                        SampleRay ray;
                        const RootBox root = kernarg_late<RootBox>(offsetof(DevScene, root));   // (the scene is the kernel's first argument)
                        bool is_alive = prepare_sample(root, rs, E - rs, ray);  // false: the ray misses the root box, nothing can block it
                        bool blocked = false;
                        while (is_alive && m) {
                            const int k = __ffs((int)m) - 1;
                            m &= m - 1u;
                            if (STATS) n_exact++;
                            if (tri_blocks(sc.btris[wrecidx[k]].p, ray, root.lo, root.hi)) { is_alive = false; blocked = true; }
                        }
                        if (!is_alive) {
                            if (q) { alive[1] = false; if (blocked) escaped[1] = false; }
                            else { alive[0] = false; if (blocked) escaped[0] = false; }
                        }
                    }
                }
                have = __any(alive[0] || alive[1]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (have && truncated) {
            // the list ran out before the shaft did and some sample is still undecided: next round, or the exact fallback
            const unsigned long long a0m = __ballot(alive[0]), a1m = __ballot(alive[1]), e0 = __ballot(escaped[0]), e1m = __ballot(escaped[1]);
            if (lane == 0) {
                unsigned int slot = next_count ? atomicAdd(next_count, 1u) : 0xffffffffu;
                RoundState o;
                o.alive[0] = a0m; o.alive[1] = a1m; o.escaped[0] = e0; o.escaped[1] = e1m;
                if (slot < next_cap) {
                    next_list[slot] = h;
                    state_out[slot] = o;
                } else {
                    const unsigned int fi = atomicAdd(last_count, 1u);    // the fallback only traces the undecided samples
                    last_list[fi] = h;
                    last_state[fi] = o;
                }
            }
        } else {
            const int esc = (int)__popcll(__ballot(escaped[0])) + (int)__popcll(__ballot(escaped[1]));
            if (lane == 0) {                                               // finish_hit with the tabulated byte
                if (fc.flags & 32u) sc.shadow_cache[rec_cell] = (uint8_t)light_byte[esc];
                else if (fc.accum) fc.accum[rec_sample] += (uint32_t)esc;      // (one chunk of a > 128-sample frame)
                else samples[rec_sample] = modulate(shaded, light_byte[esc]);
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(n_rays), b = wave_sum(n_cls), c = wave_sum(n_exact);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b);
            stat_add(&stats[8], n_recs);             // wave-level: fp32 triangle records read
            stat_add(&stats[9], n_items);            // wave-level: hit points processed
            stat_add(&stats[12], b); stat_add(&stats[13], c);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// k_shadow_cls_g -- the FIRST round of the classification with the per-hit-point preparation done for a GROUP of hit points at once.
// A first-round list holds ~7 candidates, so k_shadow_cls' lane = candidate stage (the cone planes through the surface point: ~130
// instructions) ran at 7 of 64 lanes for every hit point, and the FP64 frame of a hit point (E', L - E', the error bounds) was
// computed 64-fold redundantly.  Here a wave takes up to kGrpItems of its hit points together:
//   stage 0   lane = hit point: index, queue record, list length, E' (FP64), the fp32 frame of the classification -> LDS item table
//   fill      lane = (hit point, candidate) pair, up to 64 pairs of as many WHOLE hit points as fit: the cone-plane records -> LDS
//   items     one hit point after the other, lanes = samples: umax, the fp32 pair loop over the hit point's records, the rare FP64
//             path for undecided pairs, the pixel -- k_shadow_cls' code, with the hit point's constants read from the item table
// Same arithmetic per (hit point, candidate, sample) as k_shadow_cls<.., TAIL = false>, same verdicts, same pixels; the later rounds
// (long lists, few hit points) keep k_shadow_cls.  SR_DBG_KERNEL_SWITCH 91 runs the first round on k_shadow_cls (cross-check, A/B).
// --------------------------------------------------------------------------------------------------
constexpr int kGrpItems = 16;            // hit points prepared together (one lane each)
constexpr int kGrpRecs = 64;             // candidate records staged per fill (one lane each) >= the longest first-round list
constexpr int kGrpQueue = 64;            // undecided (hit point, sample, record) pairs a wave collects before it runs the FP64 tests for all of them
// LDS of one wave in float4 units: records (+ one of slack: the pair loop requests the record after the one it works on) + record
// indices + the item table (7 float4 per hit point: [0..1] fp32 frame + shaded colour, [2] ids, [3] alive / [4] escaped sample masks
// (4 words = 128 samples each), [5] blocked / [6] dead verdicts of the FP64 tests) + the queue of undecided pairs (2 words each): 7.6 KB,
// five workgroups per CU
constexpr int kGrpItemF4 = 7;
constexpr int kGrpWaveF4 = (kGrpRecs + 1) * 5 + kGrpRecs / 4 + kGrpItems * kGrpItemF4 + kGrpQueue / 2;

template <bool EXTRA, bool STATS>
__global__ __launch_bounds__(256, 5) void k_shadow_cls_g(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                      const HitRec* __restrict__ hits, const unsigned int* __restrict__ hit_count,
                                                      unsigned int count_cap, const unsigned int* __restrict__ index_list, int cap,
                                                      const unsigned int* __restrict__ cand_count, const int32_t* __restrict__ cand,
                                                      unsigned int* __restrict__ next_count, unsigned int next_cap,
                                                      unsigned int* __restrict__ next_list, RoundState* __restrict__ state_out,
                                                      unsigned int* __restrict__ last_count, unsigned int* __restrict__ last_list,
                                                      RoundState* __restrict__ last_state,
                                                      uint32_t* __restrict__ samples, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    typedef float f4 __attribute__((ext_vector_type(4)));
    float4* wc = reinterpret_cast<float4*>(lds_pipe) + (size_t)wave * kGrpWaveF4;        // [kGrpRecs + 1][5] cone-plane records
    int32_t* wrecidx = reinterpret_cast<int32_t*>(wc + (kGrpRecs + 1) * 5);             // [kGrpRecs] record position of every staged candidate
    float4* witem = reinterpret_cast<float4*>(wrecidx + kGrpRecs);                      // [kGrpItems][kGrpItemF4] per-hit-point constants, masks, verdicts
    uint32_t* wmask = reinterpret_cast<uint32_t*>(witem);                               // (the same table as words: item j at wmask + j * kGrpItemF4 * 4)
    uint2* wqueue = reinterpret_cast<uint2*>(witem + kGrpItems * kGrpItemF4);           // [kGrpQueue] (item | sample << 8, record position)
    const int S = fc.shadow_samples;
    uint32_t* light_byte = reinterpret_cast<uint32_t*>(reinterpret_cast<float4*>(lds_pipe) + 4 * kGrpWaveF4);   // (see k_shadow_cls)
    for (int e = tid; e <= S; e += 256) {
        const double frac = (double)e / (double)S;
        uint32_t v;
        if (fc.flags & 32u) { v = (uint32_t)(int)(frac * 254 + 1) & 0xffu; v = v ? v : 1u; }
        else v = to_byte(frac * 255);
        light_byte[e] = v;
    }
    __syncthreads();
    const unsigned int total = min(*hit_count, count_cap);
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const ClsFrame cf = cls_frame(sc, fc);
    bool valid[kPacketSlots];
    f2 OX, OY, OZ;
    {
        const int j1 = lane + 64 < S ? lane + 64 : 0, j0 = lane < S ? lane : 0;
        valid[0] = lane < S; valid[1] = lane + 64 < S;
        OX = (f2){(float)offsets[3 * j0], (float)offsets[3 * j1]};
        OY = (f2){(float)offsets[3 * j0 + 1], (float)offsets[3 * j1 + 1]};
        OZ = (f2){(float)offsets[3 * j0 + 2], (float)offsets[3 * j1 + 2]};
    }
    uint32_t n_rays = 0, n_items = 0, n_recs = 0, n_cls = 0, n_exact = 0;
    int qn = 0;                                                          // (wave-uniform) undecided pairs waiting in wqueue
    // The reference's FP64 arithmetic for the pairs the fp32 bounds could not decide (0.1 % of them), one pair per lane, for a whole
    // group of hit points at once: inline it ran with one or two lanes active for ~400 instructions whenever a hit point had such a
    // pair -- a sixth of the kernel for a thousandth of the pairs.  A blocked sample loses its bit in the hit point's "escaped" mask, a
    // sample whose ray misses the root box is decided (nothing can block it): verdict words of the item table, ORed in.
    const auto flush_queue = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        for (int qb = 0; qb < qn; qb += 64) {
            const int p = qb + lane;
            if (p < qn) {
                const uint2 e = wqueue[p];
                const unsigned int ij = e.x & 0xffu, sj = e.x >> 8;
                const unsigned int hq = wmask[ij * (kGrpItemF4 * 4) + 8];                       // item table word [2].x: the hit index
                const HitRec rq = hits[hq];
                const D3 E = mk(rq.pos[0], rq.pos[1], rq.pos[2]) + mk(rq.nrm[0], rq.nrm[1], rq.nrm[2]) * 0.001;
                const D3 rs = lpos + mk(offsets[3 * sj], offsets[3 * sj + 1], offsets[3 * sj + 2]);
                SampleRay ray;
                const RootBox root = kernarg_late<RootBox>(offsetof(DevScene, root));   // (the scene is the kernel's first argument)
                const bool in_box = prepare_sample(root, rs, E - rs, ray);              // false: the ray misses the root box, nothing can block it
                if (STATS) n_exact++;
                uint32_t* verdict = wmask + ij * (kGrpItemF4 * 4) + 20;                  // [5] blocked: words 0..3 (sample s: word s / 32), [6] dead: words 4..7
                if (!in_box) atomicOr(&verdict[4 + (sj >> 5)], 1u << (sj & 31u));
                else if (tri_blocks(sc.btris[e.y].p, ray, root.lo, root.hi)) atomicOr(&verdict[sj >> 5], 1u << (sj & 31u));
            }
        }
        qn = 0;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    const unsigned int nwaves = gridDim.x * 4u;
    const unsigned int s0 = blockIdx.x * 4u + (unsigned)wave;
    for (unsigned int slot_g = s0; slot_g < total; slot_g += nwaves * (unsigned)kGrpItems) {
        // ---- stage 0, lane = hit point: the wave's next kGrpItems items (slot_g, slot_g + nwaves, ...) ----
        const unsigned int my_slot = slot_g + (unsigned)lane * nwaves;
        const bool have_item = lane < kGrpItems && my_slot < total;
        int my_ntri = 0;
        if (have_item) {
            const unsigned int h = index_list ? index_list[my_slot] : my_slot;
            const HitRec rec = hits[h];
            const unsigned int cc = cand_count[h];
            const D3 E = mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001;
            const D3 DLd = lpos - E;
            const float efx = (float)(E.x - sc.root.centre[0]), efy = (float)(E.y - sc.root.centre[1]), efz = (float)(E.z - sc.root.centre[2]);
            const float dlx = (float)DLd.x, dly = (float)DLd.y, dlz = (float)DLd.z;
            const float dmax = __builtin_amdgcn_sqrtf(dlx * dlx + dly * dly + dlz * dlz) * 1.0001f + cf.R;   // an upper bound is all it has to be
            const uint32_t shaded = (fc.flags & 32u) ? 0u : samples[rec.sample];
            my_ntri = (int)(cc & 0xffffu);
            witem[lane * kGrpItemF4 + 0] = make_float4(efx, efy, efz, dlx);
            witem[lane * kGrpItemF4 + 1] = make_float4(dly, dlz, dmax, __uint_as_float(shaded));
            witem[lane * kGrpItemF4 + 2] = make_float4(__uint_as_float(h), __uint_as_float(rec.sample), __uint_as_float(rec.pad[0]), __uint_as_float(cc));
            witem[lane * kGrpItemF4 + 5] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // FP64 verdicts (flush_queue ORs into them): blocked samples,
            witem[lane * kGrpItemF4 + 6] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);     // samples whose ray misses the root box
        }
        int incl = my_ntri;                                               // inclusive prefix sum of the list lengths over the group's lanes
#pragma unroll
        for (int d = 1; d < kGrpItems; d <<= 1) {
            const int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const int excl = incl - my_ntri;
        const unsigned int left = (total - slot_g + nwaves - 1u) / nwaves;
        const int nitems = (int)min((unsigned)kGrpItems, left);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        int ja = 0;
        while (ja < nitems) {
            // ---- as many whole hit points as fit kGrpRecs records ----
            const int base = __builtin_amdgcn_readlane(excl, ja);
            int jb = ja + 1;
            while (jb < nitems && __builtin_amdgcn_readlane(incl, jb) - base <= kGrpRecs) ++jb;
            const int npairs = __builtin_amdgcn_readlane(incl, jb - 1) - base;
            // ---- fill, lane = (hit point, candidate): cone planes through E' and their margins (see k_shadow_cls) ----
            if (lane < npairs) {
                int jj = ja, ofs = 0;
                for (int j = ja + 1; j < jb; ++j) {
                    const int o = __builtin_amdgcn_readlane(excl, j) - base;
                    if (lane >= o) { jj = j; ofs = o; }
                }
                const float4 i0 = witem[jj * kGrpItemF4 + 0], i1 = witem[jj * kGrpItemF4 + 1], i2 = witem[jj * kGrpItemF4 + 2];
                const float efx = i0.x, efy = i0.y, efz = i0.z, dlx = i0.w, dly = i1.x, dlz = i1.y, dmax = i1.z;
                const unsigned int hj = __float_as_uint(i2.x);
                const int32_t ent = cand[(size_t)hj * cap + (lane - ofs)];
                const TriSlab slab = sc.bslab[ent];
                const f2 edx = {efx, dlx}, edy = {efy, dly}, edz = {efz, dlz};
                const f2 cn = {-slab.d, 0.0f}, c1 = {-slab.c1, 0.0f}, c2 = {-slab.c2, 0.0f}, c3 = {-slab.c3, 0.0f};
                const f2 N = pk_fma(splat(slab.n[0]), edx, pk_fma(splat(slab.n[1]), edy, pk_fma(splat(slab.n[2]), edz, cn)));    // (G0, n.(L - E'))
                const f2 P = pk_fma(splat(slab.m1[0]), edx, pk_fma(splat(slab.m1[1]), edy, pk_fma(splat(slab.m1[2]), edz, c1)));  // (K0_1, .)
                const f2 Q = pk_fma(splat(slab.m2[0]), edx, pk_fma(splat(slab.m2[1]), edy, pk_fma(splat(slab.m2[2]), edz, c2)));
                const f2 T = pk_fma(splat(slab.m3[0]), edx, pk_fma(splat(slab.m3[1]), edy, pk_fma(splat(slab.m3[2]), edz, c3)));
                const float G0 = N.x;
                float4 W[3];
                const float K0[3] = {P.x, Q.x, T.x};
                const float* mm[3] = {slab.m1, slab.m2, slab.m3};
#pragma unroll
                for (int e = 0; e < 3; ++e) {
                    const float wx = __builtin_fmaf(K0[e], slab.n[0], -(G0 * mm[e][0]));
                    const float wy = __builtin_fmaf(K0[e], slab.n[1], -(G0 * mm[e][1]));
                    const float wz = __builtin_fmaf(K0[e], slab.n[2], -(G0 * mm[e][2]));
                    W[e] = make_float4(wx, wy, wz, __builtin_fmaf(wx, dlx, __builtin_fmaf(wy, dly, wz * dlz)));
                }
                const float kmax = fmaxf(fmaxf(fabsf(P.x), fabsf(Q.x)), fabsf(T.x));
                const float mc = dmax * (kU24 * (15.0f * cf.s0 + 16.0f * (kmax + fabsf(G0))) + 1e-9f);
                float4* w = wc + lane * 5;
                w[0] = make_float4(N.y, slab.n[0], slab.n[1], slab.n[2]);
                w[1] = make_float4(W[0].w, W[0].x, W[0].y, W[0].z);
                w[2] = make_float4(W[1].w, W[1].x, W[1].y, W[1].z);
                w[3] = make_float4(W[2].w, W[2].x, W[2].y, W[2].z);
                w[4] = make_float4(G0, G0 <= -cf.a0 ? mc : 1e30f, G0 >= cf.a0 ? -1e30f : mc, 0.0f);
                wrecidx[lane] = ent;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            n_recs += (uint32_t)npairs;
            // ---- the hit points of this fill, one after the other: lanes = samples ----
            for (int j = ja; j < jb; ++j) {
                const int off_local = __builtin_amdgcn_readlane(excl, j) - base;
                const float4 i0 = witem[j * kGrpItemF4 + 0], i1 = witem[j * kGrpItemF4 + 1], i2 = witem[j * kGrpItemF4 + 2];
                const auto uni = [](float v) { return __uint_as_float((uint32_t)__builtin_amdgcn_readfirstlane((int)__float_as_uint(v))); };
                const float efx = uni(i0.x), efy = uni(i0.y), efz = uni(i0.z), dlx = uni(i0.w), dly = uni(i1.x), dlz = uni(i1.y), dmax = uni(i1.z);
                const unsigned int h = __float_as_uint(uni(i2.x)), cc = __float_as_uint(uni(i2.w));
                n_items++;
                const int ntri = (int)(cc & 0xffffu);
                const bool truncated = (cc & kTruncated) != 0;
                const bool work = ntri > 0 || truncated;      // an empty, complete list: every sample escapes
                const float a1 = 20.0f * kU24 * dmax, glo = 16.0f * a1;
                const float pm = 5e-7f * (cf.s0 + dmax);                        // position error bound of E' + u D_i in fp32 (>= 4u (|E'| + u |D|))
                // ---- per (hit point, sample): state + umax ----
                bool alive[kPacketSlots], escaped[kPacketSlots];
#pragma unroll
                for (int q = 0; q < kPacketSlots; ++q) {
                    alive[q] = false;
                    escaped[q] = valid[q];
                    if (valid[q]) {
                        n_rays++;
                        bool blocked = false;
                        if (EXTRA) {
                            const HitRec rq = hits[h];
                            const D3 Eq = mk(rq.pos[0], rq.pos[1], rq.pos[2]) + mk(rq.nrm[0], rq.nrm[1], rq.nrm[2]) * 0.001;
                            Ctr cx = {0, 0, 0, 0}; const int sj = lane + 64 * q; const D3 rs = lpos + mk(offsets[3 * sj], offsets[3 * sj + 1], offsets[3 * sj + 2]);
                            blocked = extras_block<EXTRA>(sc, rs, Eq - rs, cx);
                        }
                        if (blocked) escaped[q] = false;
                        else alive[q] = work;
                    }
                }
                f2 UM;
                {
                    const float sx = cf.hbx - 4.0f * pm, sy = cf.hby - 4.0f * pm, sz = cf.hbz - 4.0f * pm;
                    const bool ein = fabsf(efx) < sx && fabsf(efy) < sy && fabsf(efz) < sz;
                    const f2 dxv = OX + splat(dlx), dyv = OY + splat(dly), dzv = OZ + splat(dlz);
                    const f2 rx = {__builtin_amdgcn_rcpf(dxv.x), __builtin_amdgcn_rcpf(dxv.y)}, ry = {__builtin_amdgcn_rcpf(dyv.x), __builtin_amdgcn_rcpf(dyv.y)},
                             rz = {__builtin_amdgcn_rcpf(dzv.x), __builtin_amdgcn_rcpf(dzv.y)};
                    const f2 tx = ((f2){__builtin_copysignf(sx, dxv.x), __builtin_copysignf(sx, dxv.y)} - splat(efx)) * rx;
                    const f2 ty = ((f2){__builtin_copysignf(sy, dyv.x), __builtin_copysignf(sy, dyv.y)} - splat(efy)) * ry;
                    const f2 tz = ((f2){__builtin_copysignf(sz, dzv.x), __builtin_copysignf(sz, dzv.y)} - splat(efz)) * rz;
                    const f2 ut = {fminf(fminf(fminf(tx.x, ty.x), tz.x), 0.999999f), fminf(fminf(fminf(tx.y, ty.y), tz.y), 0.999999f)};
                    const f2 px = pk_fma(ut, dxv, splat(efx)), py = pk_fma(ut, dyv, splat(efy)), pz = pk_fma(ut, dzv, splat(efz));
                    const bool ok0 = ein && ut.x > 0.0f && fabsf(px.x) < cf.hbx - pm && fabsf(py.x) < cf.hby - pm && fabsf(pz.x) < cf.hbz - pm;
                    const bool ok1 = ein && ut.y > 0.0f && fabsf(px.y) < cf.hbx - pm && fabsf(py.y) < cf.hby - pm && fabsf(pz.y) < cf.hbz - pm;
                    UM = (f2){ok0 ? ut.x : -1.0f, ok1 ? ut.y : -1.0f};
                }
                const float hbm = cf.a0 + a1;
                bool have = __any(alive[0] || alive[1]);
                const f4* wc4 = reinterpret_cast<const f4*>(wc) + (size_t)off_local * 5;
                for (int cbase = 0; cbase < ntri && have; cbase += 32) {          // (32 candidates per "undecided" mask)
                    const int nc = min(32, ntri - cbase);
                    const f4* rc4 = wc4 + (size_t)cbase * 5;
                    uint32_t unc[kPacketSlots] = {0u, 0u};
                    const unsigned long long al0 = __ballot(alive[0]), al1 = __ballot(alive[1]);
                    unsigned long long blk0 = 0ull, blk1 = 0ull;                  // samples some candidate of this chunk blocks
                    int k = 0;
                    unsigned long long am0 = 0ull, am1 = 0ull;                    // samples still undecided
                    f4 nA = rc4[0], nB1 = rc4[1], nB2 = rc4[2], nB3 = rc4[3], nF = rc4[4];
                    for (; k < nc && have; ++k) {
                        const f4 A = nA, B1 = nB1, B2 = nB2, B3 = nB3, F = nF;
                        nA = rc4[k * 5 + 5]; nB1 = rc4[k * 5 + 6]; nB2 = rc4[k * 5 + 7]; nB3 = rc4[k * 5 + 8]; nF = rc4[k * 5 + 9];
                        const auto plane = [&](const f4& r) {
                            const f2 cx = {r.x, r.y}, yz = {r.z, r.w};
                            return pk_fma_hi(cx, OX, pk_fma_lo(yz, OY, pk_fma_hi_addlo(yz, OZ, cx)));
                        };
                        const f2 g1 = plane(A), c1 = plane(B1), c2 = plane(B2), c3 = plane(B3);
                        const f2 hb = pk_fma(UM, g1, splat(F.x));                 // G at u = umax_i: > 0 <=> the crossing comes earlier
                        const f2 cmin = {fminf(fminf(c1.x, c2.x), c3.x), fminf(fminf(c1.y, c2.y), c3.y)};
                        const f2 s3 = g1 - splat(glo), s1 = cmin - splat(F.y), s4 = hb - splat(hbm);
                        const f2 tm0 = splat(-F.z) - cmin, bfv = splat(-glo) - g1;
                        const float tblk0 = fminf(fminf(s1.x, s3.x), s4.x), tblk1 = fminf(fminf(s1.y, s3.y), s4.y);
                        const float tall0 = fmaxf(fmaxf(fminf(tm0.x, s3.x), bfv.x), tblk0), tall1 = fmaxf(fmaxf(fminf(tm0.y, s3.y), bfv.y), tblk1);
                        unc[0] = shift_in_not_positive(unc[0], tall0);
                        unc[1] = shift_in_not_positive(unc[1], tall1);
                        blk0 |= __ballot(tblk0 > 0.0f);
                        blk1 |= __ballot(tblk1 > 0.0f);
                        am0 = al0 & ~blk0; am1 = al1 & ~blk1;
                        if (STATS) n_cls += (uint32_t)((am0 >> lane) & 1ull) + (uint32_t)((am1 >> lane) & 1ull);
                        have = (am0 | am1) != 0ull;
                        asm volatile("" : "+v"(nA), "+v"(nB1), "+v"(nB2), "+v"(nB3), "+v"(nF));
                    }
                    if (k > 0) { unc[0] = __brev(unc[0]) >> (32 - k); unc[1] = __brev(unc[1]) >> (32 - k); }   // bit j = candidate j
                    if (((blk0 >> lane) & 1ull) != 0ull && alive[0]) { alive[0] = false; escaped[0] = false; }
                    if (((blk1 >> lane) & 1ull) != 0ull && alive[1]) { alive[1] = false; escaped[1] = false; }
                    // ---- the undecided pairs of the samples no candidate blocked so far go to the wave's queue (FP64 tests for a whole
                    //      group at once: flush_queue); such a sample stays alive for the candidates that follow (one of them may block it) ----
                    if (fc.debug != 92 && __any((alive[0] && unc[0] != 0u) || (alive[1] && unc[1] != 0u))) {      // (hook 92: TIMING EXPERIMENT ONLY, wrong pixels)
#pragma unroll 1
                        for (int q = 0; q < kPacketSlots; ++q) {
                            uint32_t m = (q ? alive[1] : alive[0]) ? (q ? unc[1] : unc[0]) : 0u;
                            while (__any(m != 0u)) {
                                const bool has = m != 0u;
                                const unsigned long long b = __ballot(has);
                                const int np = (int)__popcll(b);
                                if (qn + np > kGrpQueue) flush_queue();                   // (wave-uniform; the verdicts are ORed into the item table)
                                if (has) {
                                    const int kk = __ffs((int)m) - 1;
                                    m &= m - 1u;
                                    wqueue[qn + (int)__popcll(b & lanemask_lt())] = make_uint2((unsigned int)j | ((unsigned int)(lane + 64 * q) << 8),
                                                                                               (unsigned int)wrecidx[off_local + cbase + kk]);
                                }
                                qn += np;
                            }
                        }
                    }
                    have = __any(alive[0] || alive[1]);
                }
                // ---- the hit point's sample masks wait in the item table for the FP64 verdicts (group epilogue below) ----
                {
                    const unsigned long long a0m = __ballot(alive[0]), a1m = __ballot(alive[1]), e0 = __ballot(escaped[0]), e1m = __ballot(escaped[1]);
                    if (lane == 0) {
                        uint32_t* w = wmask + j * (kGrpItemF4 * 4) + 12;
                        w[0] = (uint32_t)a0m; w[1] = (uint32_t)(a0m >> 32); w[2] = (uint32_t)a1m; w[3] = (uint32_t)(a1m >> 32);
                        w[4] = (uint32_t)e0; w[5] = (uint32_t)(e0 >> 32); w[6] = (uint32_t)e1m; w[7] = (uint32_t)(e1m >> 32);
                    }
                }
            }
            ja = jb;
            __builtin_amdgcn_wave_barrier();                              // (the next fill overwrites the records)
        }
        // ---- group epilogue, lane = hit point: FP64 verdicts in, pixel (or next round) out ----
        flush_queue();
        if (lane < nitems) {
            const uint32_t* w = wmask + lane * (kGrpItemF4 * 4);
            const unsigned int h = w[8], rec_sample = w[9], rec_cell = w[10], cc = w[11];
            const uint32_t shaded = w[7];
            unsigned long long al[2], es[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const unsigned long long blocked = (unsigned long long)w[20 + 2 * q] | ((unsigned long long)w[21 + 2 * q] << 32);
                const unsigned long long dead = (unsigned long long)w[24 + 2 * q] | ((unsigned long long)w[25 + 2 * q] << 32);
                al[q] = ((unsigned long long)w[12 + 2 * q] | ((unsigned long long)w[13 + 2 * q] << 32)) & ~blocked & ~dead;
                es[q] = ((unsigned long long)w[16 + 2 * q] | ((unsigned long long)w[17 + 2 * q] << 32)) & ~blocked;
            }
            if ((al[0] | al[1]) != 0ull && (cc & kTruncated) != 0u) {
                // the list ran out before the shaft did and some sample is still undecided: next round, or the exact fallback
                unsigned int slot = next_count ? atomicAdd(next_count, 1u) : 0xffffffffu;
                RoundState o;
                o.alive[0] = al[0]; o.alive[1] = al[1]; o.escaped[0] = es[0]; o.escaped[1] = es[1];
                if (slot < next_cap) {
                    next_list[slot] = h;
                    state_out[slot] = o;
                } else {
                    const unsigned int fi = atomicAdd(last_count, 1u);    // the fallback only traces the undecided samples
                    last_list[fi] = h;
                    last_state[fi] = o;
                }
            } else {
                const int esc = (int)__popcll(es[0]) + (int)__popcll(es[1]);      // finish_hit with the tabulated byte
                if (fc.flags & 32u) sc.shadow_cache[rec_cell] = (uint8_t)light_byte[esc];
                else if (fc.accum) fc.accum[rec_sample] += (uint32_t)esc;      // (one chunk of a > 128-sample frame)
                else samples[rec_sample] = modulate(shaded, light_byte[esc]);
            }
        }
        __builtin_amdgcn_wave_barrier();                                  // (the next group's stage 0 overwrites the item table)
    }
    if (STATS) {
        uint32_t a = wave_sum(n_rays), b = wave_sum(n_cls), c = wave_sum(n_exact);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b);
            stat_add(&stats[8], n_recs);             // wave-level: fp32 triangle records read
            stat_add(&stats[9], n_items);            // wave-level: hit points processed
            stat_add(&stats[12], b); stat_add(&stats[13], c);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// Exact fallback of the shaft path: hit points whose candidate lists overflowed in every round still have undecided
// samples (masks in RoundState).  Each undecided sample is one any-hit BVH walk.
//   k_fb_expand    lane = fallback entry: appends one ray id (entry << 7 | sample) per undecided sample to a ray list
//                  (wave-level prefix sum, one atomicAdd per wave).  Entries that do not fit are flagged and listed
//                  for k_shadow_wave.
//   k_shadow_rays  lane = ray id: 64 rays of one to six hit points per wave instead of one hit point per wave with
//                  most lanes idle (a hit point reaches the fallback with ~15 % of its samples undecided).  A blocked
//                  sample clears its bit in the entry's escaped mask (atomicAnd).
//   k_fb_resolve   lane = entry: rayEscapeCount = popcount of the escaped masks -> pixel.
//   k_shadow_wave  wave = hit point, lanes = samples: the entries the ray list had no room for.
// --------------------------------------------------------------------------------------------------
constexpr unsigned int kFbOverflowFlag = 0x80000000u;     // in the fallback list: entry handled by k_shadow_wave

__global__ __launch_bounds__(256) void k_fb_expand(const unsigned int* __restrict__ count, unsigned int* __restrict__ fb_list,
                                                   const RoundState* __restrict__ state, unsigned int* __restrict__ rays,
                                                   unsigned int ray_cap, unsigned int* __restrict__ ray_count,
                                                   unsigned int* __restrict__ ovf_list, unsigned int* __restrict__ ovf_count) {
    const int lane = threadIdx.x & 63;
    const unsigned int total = *count;
    const unsigned int stride = gridDim.x * 256u;
    for (unsigned int i0 = blockIdx.x * 256u; i0 < total; i0 += stride) {      // whole waves stay in the loop
        const unsigned int i = i0 + threadIdx.x;
        unsigned long long a0 = 0ull, a1 = 0ull;
        if (i < total) { a0 = state[i].alive[0]; a1 = state[i].alive[1]; }
        const unsigned int n = (unsigned int)__popcll(a0) + (unsigned int)__popcll(a1);
        unsigned int incl = n;                                                 // inclusive prefix sum over the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const unsigned int v = __shfl_up(incl, d, 64);
            if (lane >= d) incl += v;
        }
        const unsigned int wave_total = __shfl(incl, 63, 64);
        unsigned int base = 0;
        if (lane == 63 && wave_total) base = atomicAdd(ray_count, wave_total);
        base = __shfl(base, 63, 64) + incl - n;
        if (n == 0) continue;
        if (base + n > ray_cap || base + n < base) {                           // no room: the per-wave kernel takes this entry
            for (unsigned int k = base; k < ray_cap && k < base + n; ++k) rays[k] = 0xffffffffu;
            fb_list[i] |= kFbOverflowFlag;
            ovf_list[atomicAdd(ovf_count, 1u)] = i;
            continue;
        }
        unsigned int k = base;
        for (unsigned long long m = a0; m; m &= m - 1) rays[k++] = (i << 7) | (unsigned int)(__ffsll((long long)m) - 1);
        for (unsigned long long m = a1; m; m &= m - 1) rays[k++] = (i << 7) | (64u + (unsigned int)(__ffsll((long long)m) - 1));
    }
}

constexpr int kRaysRefillAt = 40;        // k_shadow_rays fetches new rays when at most this many lanes are still walking
constexpr int kBounceRefillAt = 8;       // k_bounce: ... and resolves its finished rays

// Persistent lanes with refill: the rays of this list belong to different hit points and end after very different numbers
// of steps (most are blocked early, the lit ones walk the whole ray), so with one ray per lane 8 of 64 lanes were busy on
// average.  There is no coherence to lose here, unlike in k_shaft.  The walk is bvh_intersect<true> (sr_trace.h) unrolled
// into a per-lane state machine: same clip, same rayFracOffset, same fp32 culling, same FP64 triangle test.
template <bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_shadow_rays(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                     const HitRec* __restrict__ hits, const unsigned int* __restrict__ fb_list,
                                                     RoundState* __restrict__ state, const unsigned int* __restrict__ rays,
                                                     unsigned int ray_cap, const unsigned int* __restrict__ ray_count,
                                                     unsigned int* __restrict__ ray_head, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    const unsigned int total = min(*ray_count, ray_cap);
    // the grid is sized for a full ray list; with a short one most workgroups would only queue up at the work counter (thousands
    // of atomics on one address) -- they leave before touching it: 16 lanes per ray remain
    if ((unsigned long long)blockIdx.x * 1024ull >= (unsigned long long)total * 16ull) return;
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const float kInfl = 1.0f + 9.5367431640625e-7f;          // 1 + 2^-20
    Ctr sec = {0, 0, 0, 0};
    // ---- per ray ----
    unsigned int ei = 0, ej = 0;           // fallback entry, sample
    D3 s = mk(0, 0, 0), d = mk(0, 0, 0);
    double offset = 0.0;
    f2 I01 = splat(0.0f), I20 = I01, I12 = I01, B0 = I01, B1 = I01, B2 = I01;
    RayF rf = {};                          // fp32 pre-test frame of the ray (slab_rejects, sr_trace.h)
    float tlim = 0.0f;
    int sp = 0;
    int32_t ni = -1, leafA = -1, leafB = -1;
    bool active = false;
    bool drained = false;                  // wave-uniform: the head has passed the end of the list
    for (;;) {
        const unsigned long long busy = __ballot(active);
        if (!drained && (int)__popcll(busy) <= kRaysRefillAt) {
            const unsigned long long m = ~busy;
            unsigned int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(ray_head, (unsigned int)__popcll(m));
            base = __shfl(base, leader, 64);
            drained = base + (unsigned int)__popcll(m) >= total;
            if (!active) {
                const unsigned int r = base + (unsigned int)__popcll(m & lanemask_lt());
                const unsigned int id = r < total ? rays[r] : 0xffffffffu;
                if (id != 0xffffffffu) {
                    ei = id >> 7; ej = id & 127u;
                    const HitRec rec = hits[fb_list[ei] & ~kFbOverflowFlag];
                    const D3 E = mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001;
                    const D3 rs = lpos + mk(offsets[3 * ej], offsets[3 * ej + 1], offsets[3 * ej + 2]);
                    const D3 rd = E - rs;
                    sec.rays++;
                    if (EXTRA && extras_block<EXTRA>(sc, rs, rd, sec)) {
                        atomicAnd(&state[ei].escaped[ej >> 6], ~(1ull << (ej & 63u)));
                    } else {
                        s = rs; d = rd;
                        D3 end = s + d * 10000.0;
                        if (clip_segment<false>(sc.root, s, end)) {             // SpatialSubdivision.cs:394
                            offset = length(rs - s) / length(d);                // :401
                            const double lim = 1.0 - offset;                    // occluder <=> fl(t + offset) <= 1.0
                            if (!(lim < 0.0)) {
                                const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
                                const float ix = slab_inv((float)d.x), iy = slab_inv((float)d.y), iz = slab_inv((float)d.z);
                                I01 = (f2){ix, iy}; I20 = (f2){iz, ix}; I12 = (f2){iy, iz};
                                B0 = (f2){-ox * ix, -oy * iy}; B1 = (f2){-oz * iz, -ox * ix}; B2 = (f2){-oy * iy, -oz * iz};
                                tlim = (float)lim * kInfl + 1e-30f;
                                rf = make_ray_f(sc, s, d);
                                sp = 0; ni = 0; leafA = -1; leafB = -1;
                                active = true;
                            }
                        }
                    }
                }
            }
        }
        if (!__any(active)) break;
        if (active) {
            while (ni >= 0 && leafA < 0) {
                const BvhNode n = sc.bnodes[ni];
                sec.nodes++;
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
                    st.put(sp++, first0 ? n.c1 : n.c0);
                    ni = first0 ? n.c0 : n.c1;
                } else if (i0) ni = n.c0;
                else if (i1) ni = n.c1;
                else ni = (sp > 0) ? st.get(--sp) : -1;
            }
            bool blocked = false;
            while (leafA >= 0 && !blocked) {
                const int32_t first = leafA & kLeafMask, cn = (leafA >> kLeafShift) & 15;
                leafA = leafB;
                leafB = -1;
                sec.leaves++;
                // fp32 pre-test on the 64-byte records (slab_rejects); the FP64 record is fetched for the survivors only
                for (uint32_t m = leaf_survivors(sc, first, cn, rf, tlim); m && !blocked; m &= m - 1u) {
                    const int k = first + (__ffs((int)m) - 1);
                    double t; D3 pos;
                    sec.geom++;
                    if (tri_hit(sc.btris[k].p, s, d, t, pos) && inside(sc.root.lo, sc.root.hi, pos) && (t + offset <= 1.0)) blocked = true;
                }
            }
            if (blocked) {
                atomicAnd(&state[ei].escaped[ej >> 6], ~(1ull << (ej & 63u)));
                active = false;
            } else if (ni < 0 && leafA < 0) {
                active = false;                                   // walked the whole ray: the sample escapes
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom), c2 = wave_sum(sec.nodes), d2 = wave_sum(sec.leaves);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b);
            stat_add(&stats[6], c2); stat_add(&stats[7], d2);
            stat_add(&stats[16], a); stat_add(&stats[17], b);      // the exact fallback on its own: rays, FP64 records, nodes (per lane), leaves
            stat_add(&stats[18], c2); stat_add(&stats[19], d2);
        }
    }
}

__global__ __launch_bounds__(256) void k_fb_resolve(DevScene sc, FrameConst fc, const HitRec* __restrict__ hits, const unsigned int* __restrict__ count,
                                                    const unsigned int* __restrict__ fb_list, const RoundState* __restrict__ state,
                                                    uint32_t* __restrict__ samples) {
    const unsigned int total = *count;
    const unsigned int stride = gridDim.x * 256u;
    for (unsigned int i = blockIdx.x * 256u + threadIdx.x; i < total; i += stride) {
        const unsigned int e = fb_list[i];
        if (e & kFbOverflowFlag) continue;
        const int esc = (int)__popcll(state[i].escaped[0]) + (int)__popcll(state[i].escaped[1]);
        const uint32_t sample = hits[e].sample;
        const double frac = (double)esc / (double)fc.shadow_samples;          // ShadowMethod.IntersectRay :113-119
        finish_hit(sc, fc, samples, sample, hits[e].pad[0], (fc.flags & 32u) ? 0u : samples[sample], frac);
    }
}

template <bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_shadow_wave(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                     const HitRec* __restrict__ hits, const unsigned int* __restrict__ count,
                                                     const unsigned int* __restrict__ sub_list,
                                                     const unsigned int* __restrict__ index_list, const RoundState* __restrict__ state,
                                                     uint32_t* __restrict__ samples, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    const int S = fc.shadow_samples;
    const unsigned int total = *count;                 // entries of sub_list (indices into index_list / state)
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    Ctr sec = {0, 0, 0, 0};
    const unsigned int nwaves = gridDim.x * 4u;
    for (unsigned int k = blockIdx.x * 4u + (unsigned)wave; k < total; k += nwaves) {
        const unsigned int i = sub_list[k];
        const HitRec rec = hits[index_list[i] & ~kFbOverflowFlag];
        const D3 E = mk(rec.pos[0], rec.pos[1], rec.pos[2]) + mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]) * 0.001;
        const RoundState stt = state[i];           // verdicts of the list rounds: only undecided samples are traced
        int esc = 0;
        for (int j = lane, q = 0; j < S; j += 64, ++q) {
            bool escaped = ((stt.escaped[q] >> lane) & 1ull) != 0ull;
            if (((stt.alive[q] >> lane) & 1ull) != 0ull) {
                D3 rs = lpos + mk(offsets[3 * j], offsets[3 * j + 1], offsets[3 * j + 2]);
                D3 rd = E - rs;
                Hit h;
                sec.rays++;
                escaped = !(root_intersect<MODE_BVH, true, EXTRA>(sc, sc.tris, sc.extra, st, rs, rd, h, sec) && !(h.t > 1.0));
            }
            if (escaped) esc++;
        }
        esc = (int)wave_sum((uint32_t)esc);
        if (lane == 0) {
            double frac = (double)esc / (double)S;
            finish_hit(sc, fc, samples, rec.sample, rec.pad[0], (fc.flags & 32u) ? 0u : samples[rec.sample], frac);
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom), c2 = wave_sum(sec.nodes), d2 = wave_sum(sec.leaves);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b);
            stat_add(&stats[6], c2); stat_add(&stats[7], d2);
            stat_add(&stats[16], a); stat_add(&stats[17], b);
            stat_add(&stats[18], c2); stat_add(&stats[19], d2);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// k_resolve: n x n sub-pixel average with integer byte sums and truncating division
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resolve(FrameConst fc, const int32_t* __restrict__ row_map, int row_begin, int row_count,
                                                 const uint32_t* __restrict__ samples, uint32_t* __restrict__ pixels) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long npx = (long long)row_count * fc.width;
    if (idx >= npx) return;
    const int brow = (int)(idx / fc.width), col = (int)(idx - (long long)brow * fc.width);
    const int crow = row_begin + brow;
    const int out_row = (fc.strip_count > 0) ? crow : row_map[crow];
    const int n2 = fc.sub_pixel_res * fc.sub_pixel_res;
    int sumR = 0, sumG = 0, sumB = 0;
    const uint32_t* sp = samples + (size_t)idx * n2;
    for (int k = 0; k < n2; ++k) {
        uint32_t c = sp[k];
        sumR += (c >> 16) & 0xff; sumG += (c >> 8) & 0xff; sumB += c & 0xff;
    }
    sumR /= n2; sumG /= n2; sumB /= n2;
    pixels[(size_t)out_row * fc.width + col] =
        (255u << 24) + ((uint32_t)(sumR & 0xff) << 16) + ((uint32_t)(sumG & 0xff) << 8) + (uint32_t)(sumB & 0xff);   // Surface.PackRgb
}

// --------------------------------------------------------------------------------------------------
// Mirror bounces (config-5 extension, definition in trace_camera_ray / the CPU checker) as a wavefront pipeline: the
// queue holds one RAY per surviving sample (origin = hit + n * 0.001, direction = the reflection, level); k_bounce finds
// its nearest hit, stores that level's colour and queues the next reflection; k_fold blends the levels back to front.
// Secondary rays are incoherent, so k_bounce uses persistent lanes refilled from the queue (cf. k_shadow_rays); the walk
// is root_intersect<MODE_BVH, false, EXTRA> (extra geometry first, then bvh_intersect<false>) as a per-lane state machine.
// --------------------------------------------------------------------------------------------------
// WIDE: the private walk on the four-wide tree (build-order nodes, sc.b4): a step fetches one 128-byte node and tests four boxes,
// the lane enters the nearest inner child it hits and stacks the others -- about half as many dependent fetches per ray as on
// the binary tree, which is what an incoherent ray waits for (DESIGN.md "C5").  Pending leaves (up to four per step) wait in
// registers.  Same per-ray arithmetic, same result: the nearest hit with the lowest-index tie-break is order independent.
template <bool EXTRA, bool STATS, bool WIDE>
__global__ __launch_bounds__(256) void k_bounce(DevScene sc, FrameConst fc, const HitRec* __restrict__ qin, const unsigned int* __restrict__ qin_count,
                                                HitRec* __restrict__ qout, unsigned int* __restrict__ qout_count, unsigned int* __restrict__ head,
                                                uint32_t* __restrict__ levels, uint8_t* __restrict__ nlev, unsigned long long* stats,
                                                const unsigned int* __restrict__ order) {
    const int tid = threadIdx.x, lane = tid & 63;
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    const unsigned int total = *qin_count;
    const float kInfl = 1.0f + 9.5367431640625e-7f;          // 1 + 2^-20
    const int maxb = fc.max_bounces;
    Ctr sec = {0, 0, 0, 0};
    // ---- per ray ----
    uint32_t sample = 0, level = 0;
    D3 s0 = mk(0, 0, 0), d = mk(0, 0, 0);      // the ray as queued (unclipped)
    D3 s = mk(0, 0, 0);                        // clipped start (SpatialSubdivision.cs:394)
    double offset = 0.0, best = DBL_MAX;
    int32_t bestIdx = 0x7fffffff, bestK = -1;
    double ex_t = DBL_MAX;                     // nearest extra-geometry hit (GeometryCollection order, strict '<')
    D3 ex_pos = mk(0, 0, 0), ex_nrm = mk(0, 0, 0);
    uint32_t ex_color = 0;
    f2 I01 = splat(0.0f), I20 = I01, I12 = I01, B0 = I01, B1 = I01, B2 = I01;
    RayF rf = {};                              // fp32 pre-test frame of the ray (slab_rejects, sr_trace.h)
    float tlim = FLT_MAX;
    int sp = 0;
    int32_t ni = -1, leafA = -1, leafB = -1;
    int32_t leafC = -1, leafD = -1;            // (WIDE) a four-wide node can leave four pending leaves; the queue fills from leafA
    int32_t leafE = -1, leafF = -1, leafG = -1, leafH = -1;   // (WIDE) ... and the walk goes on while at most four wait: room for eight
    bool active = false, walking = false;
    bool finished = false;                     // the walk of this lane's ray is over; its colour and its next ray are made at the next refill
    bool drained = false;
    for (;;) {
        const unsigned long long busy = __ballot(active);
        // (k_shadow_rays refills at 40 busy lanes; here every refill also runs the FP64 clip of the new rays and the shading of the finished
        //  ones for the whole wave, so fewer, fuller batches win: 56: 78 ms, 40: 62, 24: 59, 16: 57, 8: 56.2, 4: 56.3, 0: 57.6 at C5)
        const bool refill_now = (int)__popcll(busy) <= kBounceRefillAt;
        // ---- the finished rays of the wave, TOGETHER (shading, level colour, next reflection): done whenever one lane's walk ended, this
        //      block -- a pow(), a dozen FP64 products, a queue append -- ran once per ray at one or two lanes; at the refill points it runs
        //      once per ~24 rays at ~24 lanes ----
        if (refill_now && __any(finished)) {
            if (finished) {
                bool hit = false;
                D3 hpos = ex_pos, hnrm = ex_nrm;
                uint32_t hcol = ex_color;
                if (ex_t < DBL_MAX) hit = true;
                if (walking && bestK >= 0 && (best + offset) < ex_t) {
                    const Rec128* r = &sc.btris[bestK];
                    hit = true;
                    hpos = s + d * best;
                    hnrm = mk(r->p[0], r->p[1], r->p[2]);
                    hcol = r->color;
                }
                if (hit) {
                    uint32_t color = hcol;
                    if (fc.flags & 1u) color = shade(fc, hpos, hnrm, color);
                    levels[(size_t)sample * (size_t)(maxb + 1) + level] = color;
                    const uint32_t nl = level + 1u;
                    if ((int)nl > maxb) {
                        nlev[sample] = (uint8_t)(nl | 0x80u);                  // the deepest level is a surface: nothing beyond it
                    } else {
                        nlev[sample] = (uint8_t)nl;
                        HitRec o;
                        const D3 refl = d - hnrm * (2.0 * dot(d, hnrm));
                        const D3 org = hpos + hnrm * 0.001;
                        o.pos[0] = org.x; o.pos[1] = org.y; o.pos[2] = org.z;
                        o.nrm[0] = refl.x; o.nrm[1] = refl.y; o.nrm[2] = refl.z;
                        o.sample = sample;
                        o.pad[0] = nl; o.pad[1] = o.pad[2] = 0;
                        qout[atomicAdd(qout_count, 1u)] = o;                   // the compiler aggregates this per wavefront
                    }
                }
                finished = false;
            }
        }
        if (!drained && refill_now) {
            const unsigned long long m = ~busy;
            unsigned int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(head, (unsigned int)__popcll(m));
            base = __shfl(base, leader, 64);
            drained = base + (unsigned int)__popcll(m) >= total;
            if (!active) {
                const unsigned int r = base + (unsigned int)__popcll(m & lanemask_lt());
                if (r < total) {
                    const HitRec q = qin[order ? order[r] : r];                        // (sr_raysort.hip: rays of a cell and octant side by side)
                    sample = q.sample; level = q.pad[0];
                    s0 = mk(q.pos[0], q.pos[1], q.pos[2]);
                    d = mk(q.nrm[0], q.nrm[1], q.nrm[2]);
                    sec.rays++;
                    ex_t = DBL_MAX;
                    if (EXTRA) {                                                   // root_intersect: the extras, first to last
                        for (int i = 0; i < sc.nextra; ++i) {
                            const Rec128* e = &sc.extra[i];
                            double t; D3 pos, nrm;
                            uint32_t tests;
                            const bool ok = extra_hit(e, s0, d, t, pos, nrm, tests);
                            sec.geom += tests;
                            if (ok && t < ex_t) { ex_t = t; ex_pos = pos; ex_nrm = nrm; ex_color = e->color; }
                        }
                    }
                    // bvh_intersect<false>: clip, rayFracOffset, fp32 culling frame
                    best = DBL_MAX; bestIdx = 0x7fffffff; bestK = -1;
                    s = s0;
                    D3 end = s + d * 10000.0;
                    walking = clip_segment<false>(sc.root, s, end);
                    if (walking) {
                        offset = length(s0 - s) / length(d);
                        const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
                        const float ix = slab_inv((float)d.x), iy = slab_inv((float)d.y), iz = slab_inv((float)d.z);
                        I01 = (f2){ix, iy}; I20 = (f2){iz, ix}; I12 = (f2){iy, iz};
                        B0 = (f2){-ox * ix, -oy * iy}; B1 = (f2){-oz * iz, -ox * ix}; B2 = (f2){-oy * iy, -oz * iz};
                        tlim = FLT_MAX;
                        rf = make_ray_f(sc, s, d);
                        sp = 0; ni = 0; leafA = -1; leafB = -1; leafC = -1; leafD = -1; leafE = leafF = leafG = leafH = -1;
                    } else {
                        ni = -1; leafA = -1; leafB = -1; leafC = -1; leafD = -1; leafE = leafF = leafG = leafH = -1;
                    }
                    active = true;
                }
            }
        }
        if (!__any(active)) {
            if (__any(finished)) continue;                        // (drained: the last rays are resolved at the top of the loop)
            break;
        }
        if (active) {
            // speculative walk: a lane that has found a leaf walks on (the leaf waits in a register queue of eight) until five leaves wait
            // or its stack is empty -- a node step can add four -- so the lanes of a wave stay in the node phase together instead of
            // idling from their first leaf to the slowest lane's; the postponed triangles cannot prune the nodes met meanwhile (3 % more
            // nodes), the phases are four times fewer: 57.5 -> 47.3 ms at C5 (walking on while <= 2 wait: 49.9)
            while (WIDE && ni >= 0 && leafE < 0) {
                const Bvh4Node n = sc.b4[ni];
                sec.nodes++;
                // entry distance and link of every inner child the ray hits (FLT_MAX: not a candidate); leaves go to the pending queue
                float tk[4];
                int32_t ck[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t, x;
                    child_slabs(n.ch[k], I01, I20, I12, B0, B1, B2, t, x);
                    const bool h = n.ch[k].n >= 0 && t <= x && x >= 0.0f && t <= tlim;
                    tk[k] = (h && n.ch[k].n == 0) ? t : FLT_MAX;
                    ck[k] = n.ch[k].c;
                    if (h && n.ch[k].n > 0) {
                        const int32_t v = n.ch[k].c | (n.ch[k].n << kLeafShift);
                        if (leafA < 0) leafA = v; else if (leafB < 0) leafB = v; else if (leafC < 0) leafC = v; else if (leafD < 0) leafD = v; else if (leafE < 0) leafE = v; else if (leafF < 0) leafF = v; else if (leafG < 0) leafG = v; else leafH = v;
                    }
                }
                // nearest first: a nearest-hit walk prunes with the distance of the hit it has, so the order in which the stacked
                // subtrees come back matters (5 compare-exchanges, registers only)
#define SR_CE(a, b) { const bool sw = tk[b] < tk[a]; const float ta = tk[a]; const int32_t ca = ck[a]; \
                      tk[a] = sw ? tk[b] : ta; ck[a] = sw ? ck[b] : ca; tk[b] = sw ? ta : tk[b]; ck[b] = sw ? ca : ck[b]; }
                SR_CE(0, 1) SR_CE(2, 3) SR_CE(0, 2) SR_CE(1, 3) SR_CE(1, 2)
#undef SR_CE
                if (tk[3] < FLT_MAX) st.put(sp++, ck[3]);         // far to near onto the stack (re-tested against tlim when popped)
                if (tk[2] < FLT_MAX) st.put(sp++, ck[2]);
                if (tk[1] < FLT_MAX) st.put(sp++, ck[1]);
                ni = tk[0] < FLT_MAX ? ck[0] : ((sp > 0) ? st.get(--sp) : -1);
            }
            while (!WIDE && ni >= 0 && leafA < 0) {
                const BvhNode n = sc.bnodes[ni];
                sec.nodes++;
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
            while (leafA >= 0) {
                const int32_t first = leafA & kLeafMask, cn = (leafA >> kLeafShift) & 15;
                leafA = leafB;
                leafB = WIDE ? leafC : -1;
                if (WIDE) { leafC = leafD; leafD = leafE; leafE = leafF; leafF = leafG; leafG = leafH; leafH = -1; }
                sec.leaves++;
                // fp32 pre-test on the 64-byte records (slab_rejects); the FP64 record is fetched for the survivors only
                for (uint32_t m = leaf_survivors(sc, first, cn, rf, tlim); m; m &= m - 1u) {
                    const int k = first + (__ffs((int)m) - 1);
                    const Rec128* r = &sc.btris[k];
                    double t; D3 pos;
                    sec.geom++;
                    if (tri_hit(r->p, s, d, t, pos) && inside(sc.root.lo, sc.root.hi, pos)) {
                        const int32_t idx = r->aux;
                        if (t < best || (t == best && idx < bestIdx)) {
                            best = t; bestIdx = idx; bestK = k;
                            tlim = (float)best * kInfl + 1e-30f;
                        }
                    }
                }
            }
            if (ni < 0 && leafA < 0) {                            // the walk is over: the ray waits for the wave's next refill point
                finished = true;
                active = false;
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom), c2 = wave_sum(sec.nodes), d2 = wave_sum(sec.leaves);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b); stat_add(&stats[6], c2); stat_add(&stats[7], d2);
            stat_add(&stats[20], a); stat_add(&stats[21], b); stat_add(&stats[22], c2); stat_add(&stats[23], d2);   // mirror rays on their own
        }
    }
}

// --------------------------------------------------------------------------------------------------
// The same level as THREE kernels (default; k_bounce above stays as the one-kernel form, SR_DBG_KERNEL_SWITCH 32): what a ray needs before
// its walk (FP64 clip against the root box, rayFracOffset) and after it (shading with its pow(), level colour, next reflection,
// queue append) is work for a full wave over consecutive rays; inside the walk kernel it ran at the refill points for the lanes
// that happened to have finished, and -- being expensive -- forced few, late refills (8 of 64 lanes still walking).  Split off,
//   k_bounce_prep    lane = ray (walk order): clip, offset -> 64-byte BounceRay
//   k_bounce_walk    persistent lanes: a refill is one 64-byte load + the fp32 frames, so it happens at kWalkRefillAt busy lanes;
//                    the walk itself is k_bounce's, a finished lane stores (t, record) and waits for the next refill
//   k_bounce_finish  lane = ray: k_bounce's finishing block
// Same per-ray arithmetic in the same order, same results.
// --------------------------------------------------------------------------------------------------
struct alignas(16) BounceRay {
    double   s[3], d[3];         // clipped start (SpatialSubdivision.cs:394), direction as queued
    double   offset;             // rayFracOffset (:401); < 0: the ray misses the root box, no walk
    uint32_t sample, level;
};
static_assert(sizeof(BounceRay) == 64, "BounceRay must be 64 bytes");
struct alignas(16) BounceHit {
    double  best;                // rayFrac from the clipped start of the nearest triangle hit (DBL_MAX: none)
    int32_t bestK, pad;          // its record (-1: none)
};
static_assert(sizeof(BounceHit) == 16, "BounceHit must be 16 bytes");
constexpr int kWalkNodeBurst = -1, kWalkLeafBurst = -1;   // node steps / leaves per phase of k_bounce_walk (-1: unlimited); SR_DBG_KERNEL_SWITCH 300 + 10 B + L
constexpr int kWalkRefillAt = 24;        // k_bounce_walk fetches new rays when at most this many lanes are still walking (8: 43.4 ms, 24: 43.3, 32: 43.9, 48: 45.3, 62: 47.5 at C5)
// The worst case of a lane's stack is 3 * depth + 2 entries -- 60 KB of LDS per workgroup at C5, i.e. two workgroups per CU, and an
// incoherent walk lives on the number of waves that wait side by side.  Only the first kBounceLdsLevels live in LDS; the deeper
// ones (rare) in global memory, [level][lane].
constexpr int kBounceLdsLevels = 24;     // C5, four bounces: all 59 levels in LDS 44.1 ms, 32: 41.1, 24: 41.0, 16: 44.1, 12: 44.1
struct SplitStack {
    int32_t* lds;        // &lds[threadIdx], stride 256
    int32_t* deep;       // &deep[global lane], stride = lanes of the grid
    int32_t  lds_levels, deep_stride;
    __device__ __forceinline__ void put(int level, int32_t v) {
        if (level < lds_levels) lds[level * 256] = v; else deep[(size_t)(level - lds_levels) * (size_t)deep_stride] = v;
    }
    __device__ __forceinline__ int32_t get(int level) const {
        return level < lds_levels ? lds[level * 256] : deep[(size_t)(level - lds_levels) * (size_t)deep_stride];
    }
};

__global__ __launch_bounds__(256) void k_bounce_prep(DevScene sc, const HitRec* __restrict__ qin, const unsigned int* __restrict__ qin_count,
                                                     const unsigned int* __restrict__ order, BounceRay* __restrict__ prep, BounceHit* __restrict__ res) {
    const unsigned int total = *qin_count, stride = gridDim.x * 256u;
    for (unsigned int r = blockIdx.x * 256u + threadIdx.x; r < total; r += stride) {
        const HitRec q = qin[order ? order[r] : r];                            // (sr_raysort.hip: rays of a cell and octant side by side)
        const D3 s0 = mk(q.pos[0], q.pos[1], q.pos[2]), d = mk(q.nrm[0], q.nrm[1], q.nrm[2]);
        D3 s = s0;
        D3 end = s + d * 10000.0;
        const bool walking = clip_segment<false>(sc.root, s, end);
        BounceRay o;
        o.s[0] = s.x; o.s[1] = s.y; o.s[2] = s.z;
        o.d[0] = d.x; o.d[1] = d.y; o.d[2] = d.z;
        o.offset = walking ? length(s0 - s) / length(d) : -1.0;
        o.sample = q.sample; o.level = q.pad[0];
        prep[r] = o;
        BounceHit h;
        h.best = DBL_MAX; h.bestK = -1; h.pad = 0;
        res[r] = h;
    }
}

// (119 VGPRs = 4 waves/SIMD; forced to 96 for 5 it spills: 46.2 instead of 41.0 ms at C5)
template <bool STATS, bool WIDE>
__global__ __launch_bounds__(256, 5) void k_bounce_walk(DevScene sc, const BounceRay* __restrict__ prep, const unsigned int* __restrict__ count,
                                                     unsigned int* __restrict__ head, BounceHit* __restrict__ res, unsigned long long* stats, int refill_at,
                                                     int32_t* __restrict__ deep, int lds_levels, int node_burst, int leaf_burst) {
    const int tid = threadIdx.x, lane = tid & 63;
    SplitStack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, deep + (size_t)blockIdx.x * 256u + (unsigned)tid, lds_levels, (int32_t)(gridDim.x * 256u)};
    const unsigned int total = *count;
    const float kInfl = 1.0f + 9.5367431640625e-7f;          // 1 + 2^-20
    Ctr sec = {0, 0, 0, 0};
    // ---- per ray ----
    unsigned int mine = 0;                     // position of the lane's ray in prep / res
    double best = DBL_MAX;
    int32_t bestIdx = 0x7fffffff, bestK = -1;
    f2 I01 = splat(0.0f), I20 = I01, I12 = I01, B0 = I01, B1 = I01, B2 = I01;
    RayF rf = {};
    float tlim = FLT_MAX;
    int sp = 0;
    int32_t ni = -1, leafA = -1, leafB = -1, leafC = -1, leafD = -1, leafE = -1, leafF = -1, leafG = -1, leafH = -1;
    bool active = false;
    bool drained = false;
    for (;;) {
        const unsigned long long busy = __ballot(active);
        if (!drained && (int)__popcll(busy) <= refill_at) {
            const unsigned long long m = ~busy;
            unsigned int base = 0;
            const int leader = __ffsll((long long)m) - 1;
            if (lane == leader) base = atomicAdd(head, (unsigned int)__popcll(m));
            base = __shfl(base, leader, 64);
            drained = base + (unsigned int)__popcll(m) >= total;
            if (!active) {
                const unsigned int r = base + (unsigned int)__popcll(m & lanemask_lt());
                if (r < total) {
                    const BounceRay q = prep[r];
                    if (q.offset >= 0.0) {
                        mine = r;
                        const D3 s = mk(q.s[0], q.s[1], q.s[2]), d = mk(q.d[0], q.d[1], q.d[2]);   // (not carried through the walk: see the FP64 tests)
                        sec.rays++;
                        best = DBL_MAX; bestIdx = 0x7fffffff; bestK = -1;
                        const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
                        const float ix = slab_inv((float)d.x), iy = slab_inv((float)d.y), iz = slab_inv((float)d.z);
                        I01 = (f2){ix, iy}; I20 = (f2){iz, ix}; I12 = (f2){iy, iz};
                        B0 = (f2){-ox * ix, -oy * iy}; B1 = (f2){-oz * iz, -ox * ix}; B2 = (f2){-oy * iy, -oz * iz};
                        tlim = FLT_MAX;
                        rf = make_ray_f(sc, s, d);
                        sp = 0; ni = 0; leafA = -1; leafB = -1; leafC = -1; leafD = -1; leafE = leafF = leafG = leafH = -1;
                        active = true;
                    } else if (STATS) sec.rays++;
                }
            }
        }
        if (!__any(active)) {
            if (drained) break;
            continue;
        }
        if (active) {
            // (the speculative walk of k_bounce: up to eight pending leaves, on while at most four wait)
            int nb = node_burst;                                   // (node steps / leaves per phase: the lanes of a wave advance in small, equal increments)
            while (WIDE && ni >= 0 && leafE < 0 && nb-- != 0) {
                const Bvh4Node n = sc.b4[ni];
                sec.nodes++;
                float tk[4];
                int32_t ck[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float t, x;
                    child_slabs(n.ch[k], I01, I20, I12, B0, B1, B2, t, x);
                    const bool h = n.ch[k].n >= 0 && t <= x && x >= 0.0f && t <= tlim;
                    tk[k] = (h && n.ch[k].n == 0) ? t : FLT_MAX;
                    ck[k] = n.ch[k].c;
                    if (h && n.ch[k].n > 0) {
                        const int32_t v = n.ch[k].c | (n.ch[k].n << kLeafShift);
                        if (leafA < 0) leafA = v; else if (leafB < 0) leafB = v; else if (leafC < 0) leafC = v; else if (leafD < 0) leafD = v; else if (leafE < 0) leafE = v; else if (leafF < 0) leafF = v; else if (leafG < 0) leafG = v; else leafH = v;
                    }
                }
#define SR_CE(a, b) { const bool sw = tk[b] < tk[a]; const float ta = tk[a]; const int32_t ca = ck[a]; \
                      tk[a] = sw ? tk[b] : ta; ck[a] = sw ? ck[b] : ca; tk[b] = sw ? ta : tk[b]; ck[b] = sw ? ca : ck[b]; }
                SR_CE(0, 1) SR_CE(2, 3) SR_CE(0, 2) SR_CE(1, 3) SR_CE(1, 2)
#undef SR_CE
                if (tk[3] < FLT_MAX) st.put(sp++, ck[3]);
                if (tk[2] < FLT_MAX) st.put(sp++, ck[2]);
                if (tk[1] < FLT_MAX) st.put(sp++, ck[1]);
                ni = tk[0] < FLT_MAX ? ck[0] : ((sp > 0) ? st.get(--sp) : -1);
            }
            while (!WIDE && ni >= 0 && leafA < 0) {
                const BvhNode n = sc.bnodes[ni];
                sec.nodes++;
                float t0, x0, t1, x1;
                node_slabs(n, I01, I20, I12, B0, B1, B2, t0, x0, t1, x1);
                const bool h0 = n.n0 >= 0 && t0 <= x0 && x0 >= 0.0f && t0 <= tlim;
                const bool h1 = n.n1 >= 0 && t1 <= x1 && x1 >= 0.0f && t1 <= tlim;
                const bool l0 = h0 && n.n0 > 0, l1 = h1 && n.n1 > 0;
                if (l0 && l1) {
                    const bool first0 = t0 <= t1;
                    leafA = (first0 ? n.c0 : n.c1) | ((first0 ? n.n0 : n.n1) << kLeafShift);
                    leafB = (first0 ? n.c1 : n.c0) | ((first0 ? n.n1 : n.n0) << kLeafShift);
                } else if (l0) leafA = n.c0 | (n.n0 << kLeafShift);
                else if (l1) leafA = n.c1 | (n.n1 << kLeafShift);
                const bool i0 = h0 && n.n0 == 0, i1 = h1 && n.n1 == 0;
                if (i0 && i1) {
                    const bool first0 = t0 <= t1;
                    st.put(sp++, first0 ? n.c1 : n.c0);
                    ni = first0 ? n.c0 : n.c1;
                } else if (i0) ni = n.c0;
                else if (i1) ni = n.c1;
                else ni = (sp > 0) ? st.get(--sp) : -1;
            }
            int lb = leaf_burst;
            while (leafA >= 0 && lb-- != 0) {
                const int32_t first = leafA & kLeafMask, cn = (leafA >> kLeafShift) & 15;
                leafA = leafB;
                leafB = WIDE ? leafC : -1;
                if (WIDE) { leafC = leafD; leafD = leafE; leafE = leafF; leafF = leafG; leafG = leafH; leafH = -1; }
                sec.leaves++;
                uint32_t m = leaf_survivors(sc, first, cn, rf, tlim);
                if (m) {
                    // The reference's FP64 test for the pre-test's survivors (1.2 per ray).  Its 30-register record and the FP64 ray
                    // are the register peak of the kernel, so the ray is read from the prepared array here instead of being carried
                    // through the walk, and the walk's fp32 frames -- dead while the test runs -- are made again from it afterwards
                    // with the expressions of the refill (same values): 23 registers less at the peak = 5 instead of 4 waves/SIMD.
                    const BounceRay* pq = &prep[mine];
                    const D3 s = mk(pq->s[0], pq->s[1], pq->s[2]), d = mk(pq->d[0], pq->d[1], pq->d[2]);
                    for (; m; m &= m - 1u) {
                        const int k = first + (__ffs((int)m) - 1);
                        const Rec128* r = &sc.btris[k];
                        double t; D3 pos;
                        sec.geom++;
                        if (tri_hit(r->p, s, d, t, pos) && inside(sc.root.lo, sc.root.hi, pos)) {
                            const int32_t idx = r->aux;
                            if (t < best || (t == best && idx < bestIdx)) {
                                best = t; bestIdx = idx; bestK = k;
                                tlim = (float)best * kInfl + 1e-30f;
                            }
                        }
                    }
                    const float ox = (float)(s.x - sc.root.centre[0]), oy = (float)(s.y - sc.root.centre[1]), oz = (float)(s.z - sc.root.centre[2]);
                    const float ix = slab_inv((float)d.x), iy = slab_inv((float)d.y), iz = slab_inv((float)d.z);
                    I01 = (f2){ix, iy}; I20 = (f2){iz, ix}; I12 = (f2){iy, iz};
                    B0 = (f2){-ox * ix, -oy * iy}; B1 = (f2){-oz * iz, -ox * ix}; B2 = (f2){-oy * iy, -oz * iz};
                    rf = make_ray_f(sc, s, d);
                }
            }
            if (ni < 0 && leafA < 0) {                            // the walk is over
                BounceHit h;
                h.best = best; h.bestK = bestK; h.pad = 0;
                res[mine] = h;
                active = false;
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom), c2 = wave_sum(sec.nodes), d2 = wave_sum(sec.leaves);
        if (lane == 0) {
            stat_add(&stats[4], a); stat_add(&stats[5], b); stat_add(&stats[6], c2); stat_add(&stats[7], d2);
            stat_add(&stats[20], a); stat_add(&stats[21], b); stat_add(&stats[22], c2); stat_add(&stats[23], d2);   // mirror rays on their own
        }
    }
}

template <bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_bounce_finish(DevScene sc, FrameConst fc, const HitRec* __restrict__ qin, const unsigned int* __restrict__ qin_count,
                                                       const unsigned int* __restrict__ order, const BounceRay* __restrict__ prep,
                                                       const BounceHit* __restrict__ res, HitRec* __restrict__ qout, unsigned int* __restrict__ qout_count,
                                                       uint32_t* __restrict__ levels, uint8_t* __restrict__ nlev, unsigned long long* stats) {
    const unsigned int total = *qin_count, stride = gridDim.x * 256u;
    const int maxb = fc.max_bounces;
    uint32_t n_geom = 0;
    // The queue append is aggregated per WORKGROUP: one atomicAdd on the queue's counter per 256 rays.  One per wave -- 160 k atomics on one address
    // per level at 10 M rays -- is what the kernel's time was: the L2 serialises them.
    __shared__ unsigned int s_cnt[2][4], s_base[2];
    const unsigned int lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned int parity = 0u;
    for (unsigned int r0 = blockIdx.x * 256u; r0 < total; r0 += stride, parity ^= 1u) {   // whole workgroups stay in the loop
        const unsigned int r = r0 + threadIdx.x;
        bool append = false;
        HitRec o;
        if (r < total) {
        const BounceRay p = prep[r];
        const BounceHit h = res[r];
        const D3 s = mk(p.s[0], p.s[1], p.s[2]), d = mk(p.d[0], p.d[1], p.d[2]);
        const bool walking = p.offset >= 0.0;
        const uint32_t sample = p.sample, level = p.level;
        double ex_t = DBL_MAX;                     // nearest extra-geometry hit (GeometryCollection order, strict '<'), from the UNCLIPPED start
        D3 ex_pos = mk(0, 0, 0), ex_nrm = mk(0, 0, 0);
        uint32_t ex_color = 0;
        if (EXTRA) {
            const HitRec q = qin[order ? order[r] : r];
            const D3 s0 = mk(q.pos[0], q.pos[1], q.pos[2]);
            for (int i = 0; i < sc.nextra; ++i) {
                const Rec128* e = &sc.extra[i];
                double t; D3 pos, nrm;
                uint32_t tests;
                const bool ok = extra_hit(e, s0, d, t, pos, nrm, tests);
                n_geom += tests;
                if (ok && t < ex_t) { ex_t = t; ex_pos = pos; ex_nrm = nrm; ex_color = e->color; }
            }
        }
        bool hit = false;
        D3 hpos = ex_pos, hnrm = ex_nrm;
        uint32_t hcol = ex_color;
        if (ex_t < DBL_MAX) hit = true;
        if (walking && h.bestK >= 0 && (h.best + p.offset) < ex_t) {
            const Rec128* t = &sc.btris[h.bestK];
            hit = true;
            hpos = s + d * h.best;
            hnrm = mk(t->p[0], t->p[1], t->p[2]);
            hcol = t->color;
        }
        if (hit) {
            uint32_t color = hcol;
            if (fc.flags & 1u) color = shade(fc, hpos, hnrm, color);
            levels[(size_t)sample * (size_t)(maxb + 1) + level] = color;
            const uint32_t nl = level + 1u;
            if ((int)nl > maxb) {
                nlev[sample] = (uint8_t)(nl | 0x80u);                  // the deepest level is a surface: nothing beyond it
            } else {
                nlev[sample] = (uint8_t)nl;
                const D3 refl = d - hnrm * (2.0 * dot(d, hnrm));
                const D3 org = hpos + hnrm * 0.001;
                o.pos[0] = org.x; o.pos[1] = org.y; o.pos[2] = org.z;
                o.nrm[0] = refl.x; o.nrm[1] = refl.y; o.nrm[2] = refl.z;
                o.sample = sample;
                o.pad[0] = nl; o.pad[1] = o.pad[2] = 0;
                append = true;
            }
        }
        }
        const unsigned long long am = __ballot(append);
        if (lane == 0u) s_cnt[parity][wave] = (unsigned int)__popcll(am);
        __syncthreads();
        if (threadIdx.x == 0u) {
            const unsigned int n = s_cnt[parity][0] + s_cnt[parity][1] + s_cnt[parity][2] + s_cnt[parity][3];
            s_base[parity] = n ? atomicAdd(qout_count, n) : 0u;
        }
        __syncthreads();
        if (append) {
            unsigned int at = s_base[parity] + (unsigned int)__popcll(am & lanemask_lt());
            for (unsigned int w = 0; w < wave; ++w) at += s_cnt[parity][w];
            qout[at] = o;
        }
    }
    if (STATS && EXTRA) {
        const uint32_t b = wave_sum(n_geom);
        if ((threadIdx.x & 63) == 0) { stat_add(&stats[5], b); stat_add(&stats[21], b); }
    }
}

// blend the stored levels of every sample back to front (the fold of trace_camera_ray)
__global__ __launch_bounds__(256) void k_fold(FrameConst fc, long long nsamples, const uint32_t* __restrict__ levels, const uint8_t* __restrict__ nlev,
                                              uint32_t* __restrict__ samples, const int32_t* __restrict__ row_map, int row_begin, int row_count) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;            // band-local sample position
    if (i >= nsamples) return;
    const int n2 = fc.sub_pixel_res * fc.sub_pixel_res;
    // position in the sample buffer: n == 1 -> the frame (image or compact strip row), n > 1 -> band-local
    size_t sidx = (size_t)i;
    if (n2 == 1) {
        const int brow = (int)(i / fc.width), col = (int)(i - (long long)brow * fc.width);
        const int crow = row_begin + brow;
        const int out_row = (fc.strip_count > 0) ? crow : row_map[crow];
        sidx = (size_t)out_row * fc.width + col;
    }
    const uint32_t code = nlev[sidx];
    const int nl = (int)(code & 0x7fu);
    if (nl == 0) return;                                                       // the camera ray missed: background, already stored
    const bool tail_is_surface = (code & 0x80u) != 0;
    const uint32_t* sf = levels + sidx * (size_t)(fc.max_bounces + 1);
    const uint32_t k = to_byte(fc.reflectivity * 255.0);
    uint32_t color = fc.background;
    const int last = tail_is_surface ? nl - 2 : nl - 1;
    for (int l = nl - 1; l >= 0; --l) {
        if (tail_is_surface && l == nl - 1) color = sf[l];
        if (l <= last) color = blend_packed(sf[l], color, k);
    }
    samples[sidx] = color;
}

// --------------------------------------------------------------------------------------------------
// rayTraceShadowsStatic (ShadowMethod.cs:75-83,103-108, Texture3DCache.cs:95-135): the light fraction of a surface
// point is looked up in a 128^3 byte texture over the unit cube; an empty cell is generated by whoever asks first.
// The reference's worker tasks race for that; the CPU checker pins a deterministic order against the reference's two
// goldens (row r of every row block, blocks ascending, before row r + 1; columns ascending; sub-samples in loop order)
// and these three kernels reproduce it: every hit point claims its cell with an atomicMin on its order key, the
// winners become the generator list that the ordinary shadow kernels process (they store the cell byte instead of
// touching the pixel, see finish_hit), then every hit point modulates its sample with its cell.
// --------------------------------------------------------------------------------------------------
constexpr int kStaticRes = 128;          // staticShadowRes, Renderer.cs:114

__device__ __forceinline__ uint32_t static_cell(const double* pos) {
    const int n = kStaticRes;
    int kx = (int)((pos[0] + 0.5) * (n - 1)), ky = (int)((pos[1] + 0.5) * (n - 1)), kz = (int)((pos[2] + 0.5) * (n - 1));   // Texture3DCache.cs:102-104
    kx = min(max(kx, 0), n - 1); ky = min(max(ky, 0), n - 1); kz = min(max(kz, 0), n - 1);                              // the asserts of :99-101, clamped
    return (uint32_t)((kx * n + ky) * n + kz);
}
// order key of a sample: ((row-in-block * blocks + block) * width + col) * n2 + sub-sample
__device__ __forceinline__ unsigned long long static_key(const FrameConst& fc, uint32_t sample, int block_height, int nblocks) {
    const unsigned int n2 = (unsigned int)(fc.sub_pixel_res * fc.sub_pixel_res);
    const unsigned int si = sample % n2, pix = sample / n2;
    const unsigned int col = pix % (unsigned int)fc.width, rowpos = pix / (unsigned int)fc.width;
    // n == 1: the sample buffer is the frame (rowpos = image row); n > 1: band-local rows (one band = the frame's row range)
    const int i = (n2 == 1) ? (int)rowpos - fc.first_row : (int)rowpos;
    const unsigned long long r = (unsigned long long)(i % block_height), b = (unsigned long long)(i / block_height);
    return ((r * (unsigned long long)nblocks + b) * (unsigned long long)fc.width + col) * n2 + si;
}

__global__ __launch_bounds__(256) void k_static_claim(FrameConst fc, const uint8_t* __restrict__ cache, const HitRec* __restrict__ hits,
                                                      const unsigned int* __restrict__ count, unsigned long long* __restrict__ claim,
                                                      int block_height, int nblocks) {
    const unsigned int total = *count, stride = gridDim.x * 256u;
    for (unsigned int h = blockIdx.x * 256u + threadIdx.x; h < total; h += stride) {
        const uint32_t cell = static_cell(hits[h].pos);
        if (cache[cell] == 0) atomicMin(&claim[cell], static_key(fc, hits[h].sample, block_height, nblocks));
    }
}

__global__ __launch_bounds__(256) void k_static_select(FrameConst fc, const uint8_t* __restrict__ cache, const HitRec* __restrict__ hits,
                                                       const unsigned int* __restrict__ count, const unsigned long long* __restrict__ claim,
                                                       int block_height, int nblocks, HitRec* __restrict__ gen, unsigned int* __restrict__ gen_count) {
    const unsigned int total = *count, stride = gridDim.x * 256u;
    for (unsigned int h = blockIdx.x * 256u + threadIdx.x; h < total; h += stride) {
        HitRec r = hits[h];
        const uint32_t cell = static_cell(r.pos);
        if (cache[cell] == 0 && claim[cell] == static_key(fc, r.sample, block_height, nblocks)) {
            r.pad[0] = cell;
            gen[atomicAdd(gen_count, 1u)] = r;            // at most one winner per cell: <= 128^3 entries
        }
    }
}

__global__ __launch_bounds__(256) void k_static_apply(const uint8_t* __restrict__ cache, const HitRec* __restrict__ hits,
                                                      const unsigned int* __restrict__ count, uint32_t* __restrict__ samples) {
    const unsigned int total = *count, stride = gridDim.x * 256u;
    for (unsigned int h = blockIdx.x * 256u + threadIdx.x; h < total; h += stride) {
        const uint32_t sample = hits[h].sample;
        samples[sample] = modulate(samples[sample], (uint32_t)cache[static_cell(hits[h].pos)]);   // ShadowMethod.cs:118
    }
}

// --------------------------------------------------------------------------------------------------
// k_cam_cones: per-frame pre-pass of the packet primary walk -- the CamCone record (sr_types.h) of every BVH triangle for
// the frame's ray origin O: FP64 cross products of the FP64 vertices, rounded once to fp32.  64 B written + 72 B (gathered)
// + 8 B read per triangle: 0.15 GB at 1 M triangles; re-run only when the origin or the tree changed.
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_cam_cones(const Rec128* __restrict__ btris, const TriSlab* __restrict__ bslab,
                                                   const double* __restrict__ v9, int n, double ox, double oy, double oz,
                                                   CamCone* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const Rec128* r = &btris[i];
    const TriSlab sl = bslab[i];
    CamCone c;
    const bool degenerate = sl.n[0] == 0.0f && sl.n[1] == 0.0f && sl.n[2] == 0.0f;   // k_make_slabs: no usable planes
    if (degenerate) {
        for (int k = 0; k < 2; ++k) { c.w12x[k] = c.w12y[k] = c.w12z[k] = c.w3nx[k] = c.w3ny[k] = c.w3nz[k] = 0.0f; c.m12[k] = c.m3n[k] = 1e30f; }
    } else {
        const double* v = v9 + (size_t)r->aux * 9;
        const D3 O = mk(ox, oy, oz);
        const D3 a1 = mk(v[0], v[1], v[2]) - O, a2 = mk(v[3], v[4], v[5]) - O, a3 = mk(v[6], v[7], v[8]) - O;
        auto ncross = [](D3 a, D3 b) { return mk(-(a.y * b.z - a.z * b.y), -(a.z * b.x - a.x * b.z), -(a.x * b.y - a.y * b.x)); };
        const D3 w1 = ncross(a1, a2), w2 = ncross(a2, a3), w3 = ncross(a3, a1);
        const float k8u = 8.0f * 5.9604645e-8f;
        c.w12x[0] = (float)w1.x; c.w12y[0] = (float)w1.y; c.w12z[0] = (float)w1.z;
        c.w12x[1] = (float)w2.x; c.w12y[1] = (float)w2.y; c.w12z[1] = (float)w2.z;
        c.w3nx[0] = (float)w3.x; c.w3ny[0] = (float)w3.y; c.w3nz[0] = (float)w3.z;
        c.w3nx[1] = (float)r->p[0]; c.w3ny[1] = (float)r->p[1]; c.w3nz[1] = (float)r->p[2];
        c.m12[0] = k8u * (float)length(w1) * 1.0001f + 1e-37f;
        c.m12[1] = k8u * (float)length(w2) * 1.0001f + 1e-37f;
        c.m3n[0] = k8u * (float)length(w3) * 1.0001f + 1e-37f;
        c.m3n[1] = k8u * 1.0001f;
    }
    out[i] = c;
}

// --------------------------------------------------------------------------------------------------
// k_order_nodes: per-frame pre-pass of the four-wide packet walks -- a copy of the node array in which every node's children are
// sorted by the squared distance of their box centres from one point (fp32, root-centre-relative like the boxes): nearest first
// for the camera rays' origin, farthest first for the point light (= nearest to the surface points first).  Empty slots go last.
// Links are untouched (a child keeps its node index), only the slots move; the sort is stable, so equal keys keep build order.
// 128 B read + 128 B written per node: 17 MB at 1 M triangles; re-run only when the point or the tree changed.
// --------------------------------------------------------------------------------------------------
// --------------------------------------------------------------------------------------------------
// k_facing_partition: per (camera origin, light) pre-pass of the packet walks.  Triangle.IntersectRay is one-sided: a ray can only
// hit a triangle whose plane it starts in front of and runs towards (Plane.cs:52-75: dirDist < 0, originDist - startDist <= 0), so
//   * a triangle with n.O - d < 0 cannot be hit by ANY ray that starts at the camera origin O ("camera-dead"), and
//   * a triangle with n.L - d + R < 0 cannot be hit by any sample ray that starts within R of the light L ("light-dead")
// (both from the FP64 plane of the record the exact test itself uses, with a 1e-9 margin).  About half of a scene's triangles are
// dead for the camera and half for the light.  The records of every leaf are re-ordered IN PLACE (FP64 record and TriSlab together;
// a leaf keeps its range, so every other walk is unaffected) into four groups -- (camera-dead, light-live), (both live),
// (camera-live, light-dead), (both dead) -- so that the camera-live records and the light-live records are each one contiguous
// run; the two runs (offset, count) per leaf slot go to side arrays that k_order_nodes applies to the camera- / light-ordered copy.
// The packet walks then never fetch a dead record, and leaves without a live record become empty slots.  One thread per
// (node, slot); the camera-cone records are re-made afterwards (they follow the records' positions).
// --------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_facing_partition(const Bvh4Node* __restrict__ base, int n4, Rec128* __restrict__ btris, TriSlab* __restrict__ bslab,
                                                          double ox, double oy, double oz, int use_cam, double lx, double ly, double lz, double R, int use_light,
                                                          int2* __restrict__ cam_rng, int2* __restrict__ light_rng) {
    const int t = blockIdx.x * 256 + threadIdx.x;
    if (t >= n4 * 4) return;
    const Bvh4Child ch = base[t >> 2].ch[t & 3];
    if (ch.n <= 0) { cam_rng[t] = make_int2(0, ch.n); light_rng[t] = make_int2(0, ch.n); return; }
    const int c = ch.c, n = ch.n;
    const auto group = [&](int k) {
        const double* p = btris[k].p;
        const double mag = 1e-9 * (1.0 + fabs(p[3]) + fabs(ox) + fabs(oy) + fabs(oz) + fabs(lx) + fabs(ly) + fabs(lz));
        const bool cam_dead = use_cam && (p[0] * ox + p[1] * oy + p[2] * oz - p[3]) < -mag;
        const bool light_dead = use_light && (p[0] * lx + p[1] * ly + p[2] * lz - p[3]) + R * 1.000001 < -mag;
        return cam_dead ? (light_dead ? 3 : 0) : (light_dead ? 2 : 1);
    };
    int pos = c, cnt[4] = {0, 0, 0, 0};
#pragma unroll
    for (int g = 0; g < 3; ++g) {                                        // (what is left after three passes is group 3)
        for (int k = pos; k < c + n; ++k) {
            if (group(k) != g) continue;
            if (k != pos) {
                uint4* a = reinterpret_cast<uint4*>(&btris[k]); uint4* b = reinterpret_cast<uint4*>(&btris[pos]);
                for (int w = 0; w < 8; ++w) { const uint4 x = a[w]; a[w] = b[w]; b[w] = x; }
                uint4* sa_ = reinterpret_cast<uint4*>(&bslab[k]); uint4* sb_ = reinterpret_cast<uint4*>(&bslab[pos]);
                for (int w = 0; w < 4; ++w) { const uint4 x = sa_[w]; sa_[w] = sb_[w]; sb_[w] = x; }
            }
            ++pos;
            if (g == 0) cnt[0]++; else if (g == 1) cnt[1]++; else cnt[2]++;
        }
    }
    cam_rng[t] = make_int2(cnt[0], cnt[1] + cnt[2]);
    light_rng[t] = make_int2(0, cnt[0] + cnt[1]);
}

hipError_t launch_facing_partition(const Bvh4Node* base, int num_nodes, Rec128* btris, TriSlab* bslab, const double origin[3], bool use_cam,
                                   const double light[3], double light_radius, bool use_light, void* cam_rng, void* light_rng, hipStream_t stream) {
    if (num_nodes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_facing_partition, dim3((unsigned)((num_nodes * 4 + 255) / 256)), dim3(256), 0, stream, base, num_nodes, btris, bslab,
                       origin[0], origin[1], origin[2], use_cam ? 1 : 0, light[0], light[1], light[2], light_radius, use_light ? 1 : 0, (int2*)cam_rng, (int2*)light_rng);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void k_order_nodes(const Bvh4Node* __restrict__ in, Bvh4Node* __restrict__ out, int n, float px, float py, float pz, int far_first,
                                                     int swap_mask, const int2* __restrict__ rng) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    Bvh4Node nd = in[i];
    if (rng) {                                                           // the live run of every leaf for this copy's rays (k_facing_partition)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (nd.ch[k].n > 0) {
                const int2 r = rng[i * 4 + k];
                nd.ch[k].c += r.x;
                nd.ch[k].n = r.y > 0 ? r.y : -1;
            }
        }
    }
    float key[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const Bvh4Child& c = nd.ch[k];
        const float cx = 0.5f * (c.lo[0] + c.hi[0]) - px, cy = 0.5f * (c.lo[1] + c.hi[1]) - py, cz = 0.5f * (c.lo[2] + c.hi[2]) - pz;
        const float d2 = cx * cx + cy * cy + cz * cz;
        key[k] = c.n < 0 ? FLT_MAX : (far_first ? -d2 : d2);
    }
    // stable insertion sort of four (fully unrolled: registers only)
#pragma unroll
    for (int a = 1; a < 4; ++a) {
#pragma unroll
        for (int b = a; b > 0; --b) {
            if (key[b] < key[b - 1]) {
                const float tk = key[b]; key[b] = key[b - 1]; key[b - 1] = tk;
                const Bvh4Child tc = nd.ch[b]; nd.ch[b] = nd.ch[b - 1]; nd.ch[b - 1] = tc;
            }
        }
    }
    // (near, far) instead of (lo, hi) on the axes where every ray of the frame travels towards smaller coordinates
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int a = 0; a < 3; ++a)
            if ((swap_mask >> a) & 1) { const float t = nd.ch[k].lo[a]; nd.ch[k].lo[a] = nd.ch[k].hi[a]; nd.ch[k].hi[a] = t; }
    // the light-ordered copy keeps a child's planes as (lo.x, lo.y), (hi.x, hi.y), (lo.z, hi.z): k_shaft_pkt4's packed slab test then needs the
    // reciprocals (ix, iy) and (iz, iz) only -- four per-lane constants less across its walk than with (lo.x, lo.y), (lo.z, hi.x), (hi.y, hi.z)
    if (far_first) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float lz = nd.ch[k].lo[2], hx = nd.ch[k].hi[0], hy = nd.ch[k].hi[1];
            nd.ch[k].lo[2] = hx; nd.ch[k].hi[0] = hy; nd.ch[k].hi[1] = lz;
        }
    }
    out[i] = nd;
}

hipError_t launch_order_nodes(const Bvh4Node* in, Bvh4Node* out, int num_nodes, const RootBox& root, const double point[3], bool far_first, int swap_mask,
                              const void* live_runs, hipStream_t stream) {
    if (num_nodes <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_order_nodes, dim3((unsigned)((num_nodes + 255) / 256)), dim3(256), 0, stream, in, out, num_nodes,
                       (float)(point[0] - root.centre[0]), (float)(point[1] - root.centre[1]), (float)(point[2] - root.centre[2]), far_first ? 1 : 0, swap_mask,
                       (const int2*)live_runs);
    return hipGetLastError();
}

hipError_t launch_cam_cones(const DevScene& sc, int ntris, const double origin[3], CamCone* out, hipStream_t stream) {
    if (ntris <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_cam_cones, dim3((unsigned)((ntris + 255) / 256)), dim3(256), 0, stream, sc.btris, sc.bslab, sc.v9, ntris,
                       origin[0], origin[1], origin[2], out);
    return hipGetLastError();
}

// --------------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------------
static int pipe_stack_levels(const DevScene& sc, int mode) {
    if (mode == MODE_REF) return sc.rdepth + 2;
    if (mode == MODE_BVH) return sc.bdepth + 2;
    return 1;
}

// the tile kernels' work distribution (TileFeed): counters[kHeadsPrimary ..] / [kHeadsShaft ..] are the per-XCD heads of k_primary /
// k_shaft_pkt4, zeroed with the band's other counters.  A launch goes persistent when its tile grid is larger than the resident grid
// (small frames keep one workgroup per tile: nothing to balance, no atomics); SR_DBG_KERNEL_SWITCH 81 keeps every launch direct
constexpr int kHeadsPrimary = 64, kHeadsShaft = kHeadsPrimary + 8 * kTileHeadStride, kCounterWords = kHeadsShaft + 8 * kTileHeadStride;
static unsigned tile_grid(const PipelineLaunch& L, unsigned virtual_blocks, int wgs_per_cu, unsigned int* heads, unsigned int** heads_out) {
    const unsigned resident = (unsigned)(L.persistent_blocks / 8 * wgs_per_cu);
    const bool persistent = virtual_blocks > resident && L.fc.debug != 81 && (virtual_blocks & 7u) == 0u;
    *heads_out = persistent ? heads : nullptr;
    return persistent ? resident : virtual_blocks;
}

template <int MODE, bool EXTRA, bool SUB, int PKT>
static hipError_t launch_primary_p(const PipelineLaunch& L, int row_begin, int row_count, uint32_t* samples, int pad_tiles) {
    // 1-D grid over the padded super-tile grid (see the tile order in k_primary), or the resident grid pulling its tiles (TileFeed)
    const unsigned vblocks = (unsigned)xcd_tile_grid(L.fc.width, row_count);
    unsigned int* heads;
    dim3 grid(tile_grid(L, vblocks, SUB ? 5 : 6, L.counters + kHeadsPrimary, &heads));
    if (heads && L.fc.debug != 82) { heads = nullptr; grid = dim3(vblocks); }      // (k_primary's persistent form is opt-in: SR_DBG_KERNEL_SWITCH 82)
    const int levels = PKT >= 2 ? 3 * L.sc.b4depth + 2 : pipe_stack_levels(L.sc, MODE);
    size_t lds = PKT ? (size_t)levels * 4 * 4 : (size_t)levels * 256 * 4;
    const auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, L.stream, L.sc, L.fc, L.row_map, row_begin, row_count,
                           samples, (HitRec*)L.hits, L.counters, L.bounce_levels, L.bounce_nlev, L.stats, pad_tiles, levels, heads, vblocks);
    };
    if (heads) { if (L.stats) go(k_primary<MODE, EXTRA, true, SUB, PKT, true>); else go(k_primary<MODE, EXTRA, false, SUB, PKT, true>); }
    else { if (L.stats) go(k_primary<MODE, EXTRA, true, SUB, PKT, false>); else go(k_primary<MODE, EXTRA, false, SUB, PKT, false>); }
    return hipGetLastError();
}

template <int MODE, bool EXTRA, bool SUB>
static hipError_t launch_primary_s(const PipelineLaunch& L, int row_begin, int row_count, uint32_t* samples, int pad_tiles) {
    if constexpr (MODE == MODE_BVH) {
        // packet walk + camera-cone filter: the rays of the frame must share their origin (focal blur moves it per sub-sample)
        const bool common_origin = !(SUB && (L.fc.flags & 4u));
        if (L.sc.bcam && common_origin && !L.per_lane_primary) {
            if (L.sc.b4cam && !L.bvh2_packets && L.sc.b4cam_known == 7) return launch_primary_p<MODE, EXTRA, SUB, 3>(L, row_begin, row_count, samples, pad_tiles);
            if (L.sc.b4cam && !L.bvh2_packets) return launch_primary_p<MODE, EXTRA, SUB, 2>(L, row_begin, row_count, samples, pad_tiles);
            return launch_primary_p<MODE, EXTRA, SUB, 1>(L, row_begin, row_count, samples, pad_tiles);
        }
    }
    return launch_primary_p<MODE, EXTRA, SUB, 0>(L, row_begin, row_count, samples, pad_tiles);
}

template <int MODE, bool EXTRA>
static hipError_t launch_shadow_lanes_t(const PipelineLaunch& L, uint32_t* samples, long long max_hits, const unsigned int* count,
                                        unsigned int* head, const unsigned int* index_list) {
    int levels = pipe_stack_levels(L.sc, MODE);
    size_t lds = (((size_t)levels * 256 * 4 + 15) & ~(size_t)15) + (size_t)L.fc.shadow_samples * 3 * sizeof(double);
    long long want = (max_hits + 255) / 256;
    unsigned blocks = (unsigned)std::min<long long>(want, (long long)L.persistent_blocks);
    if (blocks == 0) return hipSuccess;
    if (L.stats)
        hipLaunchKernelGGL((k_shadow<MODE, EXTRA, true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits,
                           count, head, index_list, samples, levels, L.stats);
    else
        hipLaunchKernelGGL((k_shadow<MODE, EXTRA, false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits,
                           count, head, index_list, samples, levels, L.stats);
    return hipGetLastError();
}

static void pipe_events(const PipelineLaunch& L, int kid, hipEvent_t& e0, hipEvent_t& e1) {
    e0 = e1 = nullptr;
    if (L.get_events) L.get_events(L.user, kid, &e0, &e1);
}

template <int MODE>
static bool shaft_path(const PipelineLaunch& L) {
    return MODE == MODE_BVH && (L.fc.flags & 8u) && (L.fc.shadow_samples <= 64 * kPacketSlots || L.fc.accum) && !L.per_lane_shadows && L.round_cand[0];
}

// counters: [0] hit_count  [1] k_shadow work head  [2] fallback_count  [3] fallback work head
template <int MODE, bool EXTRA>
static hipError_t launch_shadow_t(const PipelineLaunch& L, uint32_t* samples, long long max_hits) {
    hipError_t e;
    hipEvent_t e0, e1;
    const bool shaft = shaft_path<MODE>(L);
    if (!shaft) {
        pipe_events(L, K_SHADOW, e0, e1);
        if (e0 && (e = hipEventRecord(e0, L.stream)) != hipSuccess) return e;
        e = launch_shadow_lanes_t<MODE, EXTRA>(L, samples, max_hits, L.counters, L.counters + 1, nullptr);
        if (e != hipSuccess) return e;
        if (e1 && (e = hipEventRecord(e1, L.stream)) != hipSuccess) return e;
        return hipSuccess;
    }
    // counters: [0] hits  [1] k_shadow head  [2..] items entering round 1, 2, ..  [2+R-1] fallback count  [6] round-0 work items
    unsigned int* fb_count = L.counters + 2 + (kShaftRounds - 1);
    unsigned int* work0 = L.counters + 6;
    unsigned int order_items = 0;                 // > 0: the persistent shaft walk left its tiles' walk lengths behind (see below)
    unsigned long long order_new_tag = 0;
    for (int round = 0; round < kShaftRounds; ++round) {
        const bool first = round == 0, last = round == kShaftRounds - 1;
        const unsigned int* count_ptr = first ? L.counters : L.counters + 1 + round;
        const unsigned count_cap = first ? 0xffffffffu : L.round_items[round];
        const unsigned int* ilist = first ? nullptr : L.round_list[round];
        const long long max_items = first ? max_hits : (long long)std::min<long long>(max_hits, count_cap);
        if (max_items <= 0) break;
        const int cap = L.round_cap[round];
        // ---- k_shaft ----
        pipe_events(L, first ? K_SHAFT : K_SHAFT2, e0, e1);
        if (e0 && (e = hipEventRecord(e0, L.stream)) != hipSuccess) return e;
        {
            const int levels = pipe_stack_levels(L.sc, MODE_BVH);
            unsigned blocks = (unsigned)((max_items + 255) / 256);
            if (first && !(L.per_lane_shaft & 1) && L.sc.b4light && !L.bvh2_packets) {
                // round 1 (default): one packet walk on the four-wide, light-ordered tree per 64 consecutive queue entries
                const int lv4 = 3 * L.sc.b4depth + 2;
                size_t lds = ((size_t)lv4 * 4 + (size_t)lv4 * 64 * 2) * 4;
                const int tn2 = L.tile_queue_n2, trows = L.tile_queue_rows;
                if (tn2 > 0) blocks = (unsigned)(xcd_tile_grid(L.fc.width, trows) * tn2);      // the grid of k_primary (x sub-samples)
                unsigned int* heads;
                const unsigned vblocks = blocks;
                blocks = tile_grid(L, vblocks, (L.fc.debug >= 830 && L.fc.debug <= 836) ? L.fc.debug - 830 : (L.shaft_wgs_per_cu > 0 ? L.shaft_wgs_per_cu : 6), L.counters + kHeadsShaft, &heads);          // ... or the resident grid pulling its tiles (hook 830 + n: n workgroups per CU)
                // (persistent) longest walks first: the lists k_tile_order made from the previous frame's walk lengths, if that frame had this tile grid
                const unsigned long long order_tag = ((unsigned long long)vblocks << 32) | ((unsigned long long)(unsigned)L.fc.width << 12) | (unsigned long long)(unsigned)tn2;
                const bool keep_cost = heads && L.tile_cost && L.tile_order && L.tile_order_tag && L.fc.debug != 84;
                const unsigned int* order = (keep_cost && *L.tile_order_tag == order_tag) ? L.tile_order : nullptr;
                unsigned int* cost = keep_cost ? L.tile_cost : nullptr;
                const auto go = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, cap, lv4, tn2, trows, L.round_cand_count[round], L.round_cand[round], samples, work0, L.round_list[0], L.stats, heads, vblocks, order, cost);
                };
                // 6 waves/SIMD (85 VGPRs): 5.16 ms on the headline frame; 7 waves (72 VGPRs, spills in the node step) 5.44; 5 waves 5.61
                // ((near, far) planes in the light-ordered copy, as in the camera-ordered one, were measured: fewer instructions, more spills at
                //  this kernel's register budget -- 5.17 -> 5.30 ms; 5.49 ms at 5 waves/SIMD)
                // one instantiation per set of axes on which the light-ordered copy holds (near, far) planes (the light lies outside the root box there)
                const auto go_known = [&](auto known) {
                    constexpr int K = decltype(known)::value;
                    if (heads) { if (L.stats) go(k_shaft_pkt4<true, 6, true, K>); else go(k_shaft_pkt4<false, 6, true, K>); }
                    else { if (L.stats) go(k_shaft_pkt4<true, 6, false, K>); else go(k_shaft_pkt4<false, 6, false, K>); }
                };
                switch (L.sc.b4light_known & 7) {
                    case 1: go_known(std::integral_constant<int, 1>()); break;
                    case 2: go_known(std::integral_constant<int, 2>()); break;
                    case 3: go_known(std::integral_constant<int, 3>()); break;
                    case 4: go_known(std::integral_constant<int, 4>()); break;
                    case 5: go_known(std::integral_constant<int, 5>()); break;
                    case 6: go_known(std::integral_constant<int, 6>()); break;
                    case 7: go_known(std::integral_constant<int, 7>()); break;
                    default: go_known(std::integral_constant<int, 0>()); break;
                }
                if (keep_cost) { order_items = (vblocks >> 3) * 4u; order_new_tag = order_tag; }
            } else if (first && !(L.per_lane_shaft & 1)) {
                // round 1 on the binary tree (cross-check): one packet walk per 64 consecutive queue entries (one 8x8-pixel tile when the queue is tile-aligned)
                size_t lds = ((size_t)levels * 4 + (size_t)levels * 64 * 2) * 4;
                const int tn2 = L.tile_queue_n2, trows = L.tile_queue_rows;
                if (tn2 > 0) blocks = (unsigned)(xcd_tile_grid(L.fc.width, trows) * tn2);      // the grid of k_primary (x sub-samples)
                if (L.stats) hipLaunchKernelGGL((k_shaft_pkt<true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, cap, levels, tn2, trows, L.round_cand_count[round], L.round_cand[round], samples, work0, L.round_list[0], L.stats);
                else hipLaunchKernelGGL((k_shaft_pkt<false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, cap, levels, tn2, trows, L.round_cand_count[round], L.round_cand[round], samples, work0, L.round_list[0], L.stats);
            } else if (!first && !(L.per_lane_shaft & 2)) {
                // later rounds (scattered hit points): one wave per hit point, see k_shaft_coop.  They collect from scratch (skip = 0):
                // the order in which round 1 met a hit point's candidates was its wave's, so "skip the first cap" would not name
                // the same triangles; re-testing a candidate is harmless (it did not block the samples that are still undecided)
                const size_t lds = 4 * (size_t)(kCoopFrontier + kCoopTris) * 4;
                // 32 KB of LDS per workgroup = 5 resident per CU; a grid of ~6 times that many lets the dispatcher balance the walks
                // (headline frame: 1280 workgroups 0.90 ms, 2048 0.78, 8192 0.73)
                const unsigned cblocks = (unsigned)std::min<long long>((max_items + 3) / 4, (long long)L.persistent_blocks * 4);
                if (L.stats) hipLaunchKernelGGL((k_shaft_coop<true>), dim3(cblocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, count_cap, ilist, cap, L.round_cand_count[round], L.round_cand[round], L.stats);
                else hipLaunchKernelGGL((k_shaft_coop<false>), dim3(cblocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, count_cap, ilist, cap, L.round_cand_count[round], L.round_cand[round], L.stats);
            } else {
                // private per-lane walks: the first round as a cross-check of the packet walk, the later rounds as one of k_shaft_coop
                size_t lds = (size_t)levels * 256 * 4;
                unsigned int* wc = first ? work0 : nullptr;
                unsigned int* wl = first ? L.round_list[0] : nullptr;
                if (L.stats) hipLaunchKernelGGL((k_shaft<true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, count_cap, ilist, first ? 0 : L.round2_node_budget, cap, L.round_cand_count[round], L.round_cand[round], samples, wc, wl, L.stats, L.counters);
                else hipLaunchKernelGGL((k_shaft<false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, (const HitRec*)L.hits, count_ptr, count_cap, ilist, first ? 0 : L.round2_node_budget, cap, L.round_cand_count[round], L.round_cand[round], samples, wc, wl, L.stats, L.counters);
            }
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (e1 && (e = hipEventRecord(e1, L.stream)) != hipSuccess) return e;
        // ---- k_shadow_test ----
        pipe_events(L, first ? K_SHADOW : K_SHADOW2, e0, e1);
        if (e0 && (e = hipEventRecord(e0, L.stream)) != hipSuccess) return e;
        {
            size_t lds = 4 * ((size_t)kRecordsPerPass * kRecStride16 * 16 + (size_t)kTailSlots * kRayStride8 * 8);
            long long want = (max_items + 3) / 4;
            unsigned blocks = (unsigned)std::min<long long>(want, (long long)L.persistent_blocks * 2);
            const RoundState* st_in = first ? nullptr : (const RoundState*)L.round_state[round];
            unsigned int* next_count = last ? nullptr : L.counters + 2 + round;
            const unsigned next_cap = last ? 0u : L.round_items[round + 1];
            unsigned int* next_list = last ? nullptr : L.round_list[round + 1];
            RoundState* st_out = last ? nullptr : (RoundState*)L.round_state[round + 1];
            // round 0 iterates the compacted list of hits that k_shaft could not decide by itself
            const unsigned int* t_count = first ? work0 : count_ptr;
            const unsigned int* t_list = first ? L.round_list[0] : ilist;
            if (!L.exact_shadow_tests) {
                // default: fp32 classification, FP64 only for the pairs it cannot decide
                const size_t lds_c = 4 * (size_t)kClsWaveF4 * sizeof(float4) + 128 * 3 * sizeof(float) + (64 * kPacketSlots + 1) * sizeof(uint32_t);
                const auto args = [&](auto kern) {
                    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_c, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, t_count, count_cap, t_list, st_in, cap, first ? 1 : 0, L.round_cand_count[round], L.round_cand[round], next_count, next_cap, next_list, st_out, fb_count, L.fallback, (RoundState*)L.fallback_state, samples, L.stats);
                };
                if (first && L.fc.debug != 91 && cap <= kGrpRecs) {
                    // first round: the per-hit-point preparation done for groups of hit points (k_shadow_cls_g)
                    const size_t lds_g = 4 * (size_t)kGrpWaveF4 * sizeof(float4) + (64 * kPacketSlots + 1) * sizeof(uint32_t);
                    const auto args_g = [&](auto kern) {
                        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds_g, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, t_count, count_cap, t_list, cap, L.round_cand_count[round], L.round_cand[round], next_count, next_cap, next_list, st_out, fb_count, L.fallback, (RoundState*)L.fallback_state, samples, L.stats);
                    };
                    if (L.stats) args_g(k_shadow_cls_g<EXTRA, true>); else args_g(k_shadow_cls_g<EXTRA, false>);
                } else if (first) { if (L.stats) args(k_shadow_cls<EXTRA, true, false>); else args(k_shadow_cls<EXTRA, false, false>); }
                else { if (L.stats) args(k_shadow_cls<EXTRA, true, true>); else args(k_shadow_cls<EXTRA, false, true>); }
            } else if (L.stats) hipLaunchKernelGGL((k_shadow_test<EXTRA, true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, t_count, count_cap, t_list, st_in, cap, first ? 1 : 0, L.round_cand_count[round], L.round_cand[round], next_count, next_cap, next_list, st_out, fb_count, L.fallback, (RoundState*)L.fallback_state, samples, L.stats);
            else hipLaunchKernelGGL((k_shadow_test<EXTRA, false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, t_count, count_cap, t_list, st_in, cap, first ? 1 : 0, L.round_cand_count[round], L.round_cand[round], next_count, next_cap, next_list, st_out, fb_count, L.fallback, (RoundState*)L.fallback_state, samples, L.stats);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        if (e1 && (e = hipEventRecord(e1, L.stream)) != hipSuccess) return e;
    }
    // ---- fallback: hits that are still undecided after the longest list ----
    pipe_events(L, K_FALLBACK, e0, e1);
    if (e0 && (e = hipEventRecord(e0, L.stream)) != hipSuccess) return e;
    {
        // counters: [10] rays in the fallback ray list  [11] entries the list had no room for  [12] k_shadow_rays' work head
        unsigned int* ray_count = L.counters + 10;
        unsigned int* ovf_count = L.counters + 11;
        RoundState* fst = (RoundState*)L.fallback_state;
        size_t lds = (size_t)pipe_stack_levels(L.sc, MODE_BVH) * 256 * 4;
        unsigned small = (unsigned)std::min<long long>((max_hits + 255) / 256, (long long)L.persistent_blocks);
        hipLaunchKernelGGL(k_fb_expand, dim3(small), dim3(256), 0, L.stream, fb_count, L.fallback, (const RoundState*)fst, L.fallback_rays,
                           L.fallback_ray_cap, ray_count, L.fallback_overflow, ovf_count);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        unsigned blocks = (unsigned)std::min<long long>(((long long)L.fallback_ray_cap + 255) / 256, (long long)L.persistent_blocks);
        if (L.stats) hipLaunchKernelGGL((k_shadow_rays<EXTRA, true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, L.fallback, fst, L.fallback_rays, L.fallback_ray_cap, ray_count, L.counters + 12, L.stats);
        else hipLaunchKernelGGL((k_shadow_rays<EXTRA, false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, L.fallback, fst, L.fallback_rays, L.fallback_ray_cap, ray_count, L.counters + 12, L.stats);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        hipLaunchKernelGGL(k_fb_resolve, dim3(small), dim3(256), 0, L.stream, L.sc, L.fc, (const HitRec*)L.hits, fb_count, L.fallback, (const RoundState*)fst, samples);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        unsigned wblocks = (unsigned)std::min<long long>((max_hits + 3) / 4, (long long)L.persistent_blocks * 2);
        if (L.stats) hipLaunchKernelGGL((k_shadow_wave<EXTRA, true>), dim3(wblocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, ovf_count, L.fallback_overflow, L.fallback, (const RoundState*)fst, samples, L.stats);
        else hipLaunchKernelGGL((k_shadow_wave<EXTRA, false>), dim3(wblocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits, ovf_count, L.fallback_overflow, L.fallback, (const RoundState*)fst, samples, L.stats);
        if ((e = hipGetLastError()) != hipSuccess) return e;
    }
    if (e1 && (e = hipEventRecord(e1, L.stream)) != hipSuccess) return e;
    // the next frame's longest-first tile lists, from this frame's walk lengths: last in the stage (nothing of this frame waits for it)
    if (order_items) {
        hipLaunchKernelGGL(k_tile_order, dim3(8), dim3(kOrderWaves * 64), 0, L.stream, (const unsigned int*)L.tile_cost, L.tile_order, order_items,
                           (L.fc.debug >= 840 && L.fc.debug < 880) ? (unsigned)(L.fc.debug - 840) : 12u,   // longest = >= 3 x the mean (2 x: 1.41 ms per rank of 8, 3 x: 1.37, 4 x: 1.40)
                           (L.fc.debug > 880 && L.fc.debug <= 880 + kOrderClasses) ? L.fc.debug - 880 : 3);   // classes (hook 880 + n): 2 / 3 / 4: slowest rank of 8 1.40 / 1.31 / 1.32 ms at 3 workgroups per CU (1.37 / 1.40 / 1.42 at 6)
        if ((e = hipGetLastError()) != hipSuccess) return e;
        *L.tile_order_tag = order_new_tag;
    }
    return hipSuccess;
}

template <int MODE, bool EXTRA>
static hipError_t launch_pipeline_t(const PipelineLaunch& L) {
    const int n2 = L.fc.sub_pixel_res * L.fc.sub_pixel_res;
    const bool shadows = (L.fc.flags & 2u) != 0;
    // rows are processed in bands so that the hit queue / sample buffer stay within their allocation
    for (int row_begin = L.row_first; row_begin < L.row_limit; row_begin += L.band_rows) {
        int row_count = std::min(L.band_rows, L.row_limit - row_begin);
        uint32_t* samples = (n2 == 1) ? L.pixels : L.samples;
        hipError_t e;
        e = hipMemsetAsync(L.counters, 0, kCounterWords * sizeof(unsigned int), L.stream);   // the band's counters and both kernels' tile heads
        if (e != hipSuccess) return e;
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (L.get_events) L.get_events(L.user, K_PRIMARY, &e0, &e1);
        if (e0) { e = hipEventRecord(e0, L.stream); if (e != hipSuccess) return e; }
        // the shaft path of a dynamic-shadow frame reads the hit queue tile by tile (k_shaft_pkt): 64-aligned entries
        const bool shaft_frame = (MODE != MODE_BVH && L.shadows_on_bvh) ? shaft_path<MODE_BVH>(L) : shaft_path<MODE>(L);
        const int pad_tiles = (shadows && !(L.fc.flags & 32u) && L.fc.max_bounces == 0 && shaft_frame) ? 1 : 0;
        e = L.fc.sub_pixel_res > 1 ? launch_primary_s<MODE, EXTRA, true>(L, row_begin, row_count, samples, pad_tiles)
                                   : launch_primary_s<MODE, EXTRA, false>(L, row_begin, row_count, samples, pad_tiles);
        if (e == hipSuccess && pad_tiles) {
            // the tile-indexed queue has one slot per pixel of every (whole) 16x16 tile and sub-sample: its "count" is its size
            const unsigned int slots = (unsigned)(((L.fc.width + 15) / 16) * ((row_count + 15) / 16)) * 256u * (unsigned)n2;
            e = hipMemsetD32Async((hipDeviceptr_t)L.counters, (int)slots, 1, L.stream);
        }
        if (e != hipSuccess) return e;
        if (e1) { e = hipEventRecord(e1, L.stream); if (e != hipSuccess) return e; }
        if (L.fc.max_bounces > 0 && L.bounce_levels) {
            // ---- mirror bounces as a wavefront pipeline: one k_bounce per level, the queues ping-pong ----
            // counters: [0] / [1] ray counts of the two queues  [2] k_bounce's work head
            const long long band_n = (long long)row_count * L.fc.width * n2;
            hipEvent_t b0 = nullptr, b1 = nullptr;
            if (L.get_events) L.get_events(L.user, K_BOUNCE, &b0, &b1);         // all levels' k_bounce launches + k_fold of this band
            if (b0 && (e = hipEventRecord(b0, L.stream)) != hipSuccess) return e;
            const bool wide = L.sc.b4 != nullptr && !L.bvh2_packets;                     // private walks on the four-wide tree
            const size_t lds = (size_t)(wide ? 3 * L.sc.b4depth + 2 : pipe_stack_levels(L.sc, MODE_BVH)) * 256 * 4;
            const unsigned blocks = (unsigned)std::min<long long>((band_n + 255) / 256, (long long)L.persistent_blocks);
            int cur = 0;
            for (int level = 1; level <= L.fc.max_bounces; ++level) {
                if ((e = hipMemsetAsync(L.counters + (1 - cur), 0, 4, L.stream)) != hipSuccess) return e;
                if ((e = hipMemsetAsync(L.counters + 2, 0, 4, L.stream)) != hipSuccess) return e;
                const HitRec* qin = (const HitRec*)(cur == 0 ? L.hits : L.hits2);
                HitRec* qout = (HitRec*)(cur == 0 ? L.hits2 : L.hits);
                // the level's rays in (origin cell, direction octant) order: neighbouring lanes walk neighbouring subtrees
                const unsigned int* order = nullptr;
                if (L.ray_sort_buf && L.fc.debug != 31) {
                    const unsigned cap = (unsigned)band_n;
                    unsigned int* b = L.ray_sort_buf;
                    if ((e = ray_sort(qin, L.counters + cur, cap, L.sc.root, b, b + (size_t)cap, b + 2 * (size_t)cap, b + 3 * (size_t)cap, L.ray_sort_temp, L.ray_sort_temp_bytes, L.stream)) != hipSuccess) return e;
                    order = b + 3 * (size_t)cap;
                }
                if (L.fc.debug == 32 || !L.bounce_prep || !L.bounce_res) {      // (hook: the level as one kernel)
                    const auto go = [&](auto kern) {
                        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, qin, L.counters + cur, qout, L.counters + (1 - cur), L.counters + 2, L.bounce_levels, L.bounce_nlev, L.stats, order);
                    };
                    if (wide) { if (L.stats) go(k_bounce<EXTRA, true, true>); else go(k_bounce<EXTRA, false, true>); }
                    else { if (L.stats) go(k_bounce<EXTRA, true, false>); else go(k_bounce<EXTRA, false, false>); }
                } else {
                    BounceRay* prep = (BounceRay*)L.bounce_prep;
                    BounceHit* res = (BounceHit*)L.bounce_res;
                    hipLaunchKernelGGL(k_bounce_prep, dim3(blocks), dim3(256), 0, L.stream, L.sc, qin, L.counters + cur, order, prep, res);
                    // stack levels in LDS: kBounceLdsLevels when the rest of the worst case fits the overflow buffer, all of them otherwise
                    const int levels_all = (int)(lds / (256 * 4));
                    int lds_levels = (L.fc.debug >= 200 && L.fc.debug < 300) ? L.fc.debug - 200 : kBounceLdsLevels;      // (hook)
                    lds_levels = std::max(1, std::min(lds_levels, levels_all));
                    if ((size_t)(levels_all - lds_levels) * (size_t)blocks * 256 * 4 > L.bounce_stack_bytes || !L.bounce_stack) lds_levels = levels_all;
                    const auto walk = [&](auto kern) {
                        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), (size_t)lds_levels * 256 * 4, L.stream, L.sc, (const BounceRay*)prep, L.counters + cur, L.counters + 2, res, L.stats,
                                           (L.fc.debug >= 100 && L.fc.debug < 164) ? L.fc.debug - 100 : kWalkRefillAt, L.bounce_stack, lds_levels,
                                           (L.fc.debug >= 300 && L.fc.debug < 400) ? ((L.fc.debug - 300) / 10 ? (L.fc.debug - 300) / 10 : -1) : kWalkNodeBurst,
                                           (L.fc.debug >= 300 && L.fc.debug < 400) ? ((L.fc.debug - 300) % 10 ? (L.fc.debug - 300) % 10 : -1) : kWalkLeafBurst);
                    };
                    if (wide) { if (L.stats) walk(k_bounce_walk<true, true>); else walk(k_bounce_walk<false, true>); }
                    else { if (L.stats) walk(k_bounce_walk<true, false>); else walk(k_bounce_walk<false, false>); }
                    const auto fin = [&](auto kern) {
                        hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, L.stream, L.sc, L.fc, qin, L.counters + cur, order, (const BounceRay*)prep, (const BounceHit*)res,
                                           qout, L.counters + (1 - cur), L.bounce_levels, L.bounce_nlev, L.stats);
                    };
                    if (L.stats) fin(k_bounce_finish<EXTRA, true>); else fin(k_bounce_finish<EXTRA, false>);
                }
                if ((e = hipGetLastError()) != hipSuccess) return e;
                cur = 1 - cur;
            }
            hipLaunchKernelGGL(k_fold, dim3((unsigned)((band_n / n2 * n2 + 255) / 256)), dim3(256), 0, L.stream, L.fc, band_n, (const uint32_t*)L.bounce_levels,
                               (const uint8_t*)L.bounce_nlev, samples, L.row_map, row_begin, row_count);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if (b1 && (e = hipEventRecord(b1, L.stream)) != hipSuccess) return e;
        } else if (shadows && (L.fc.flags & 32u)) {
            // ---- static frame: claim cells, run the shadow kernels on the generators only, apply the cache ----
            // counters: [0] hit points of the band  [13] generators  [14] copy of [0] while the generators are processed
            const long long max_hits = (long long)row_count * L.fc.width * n2;
            const int conc = L.static_concurrency > 0 ? L.static_concurrency : 4;        // Renderer.cs:92
            const int block_height = (L.fc.num_rows - 1 + conc) / conc;                    // :1661
            const int nblocks = (L.fc.num_rows - 1 + block_height) / block_height;
            const unsigned blocks = (unsigned)std::min<long long>((max_hits + 255) / 256, (long long)L.persistent_blocks);
            if ((e = hipMemsetAsync(L.static_claim, 0xff, (size_t)kStaticRes * kStaticRes * kStaticRes * 8, L.stream)) != hipSuccess) return e;
            hipLaunchKernelGGL(k_static_claim, dim3(blocks), dim3(256), 0, L.stream, L.fc, (const uint8_t*)L.sc.shadow_cache, (const HitRec*)L.hits, L.counters,
                               L.static_claim, block_height, nblocks);
            hipLaunchKernelGGL(k_static_select, dim3(blocks), dim3(256), 0, L.stream, L.fc, (const uint8_t*)L.sc.shadow_cache, (const HitRec*)L.hits, L.counters,
                               (const unsigned long long*)L.static_claim, block_height, nblocks, (HitRec*)L.static_hits, L.counters + 13);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            if ((e = hipMemcpyAsync(L.counters + 14, L.counters, 4, hipMemcpyDeviceToDevice, L.stream)) != hipSuccess) return e;
            if ((e = hipMemcpyAsync(L.counters, L.counters + 13, 4, hipMemcpyDeviceToDevice, L.stream)) != hipSuccess) return e;
            if ((e = hipMemsetAsync(L.counters + kHeadsShaft, 0, 8 * kTileHeadStride * sizeof(unsigned int), L.stream)) != hipSuccess) return e;
            PipelineLaunch G = L;
            G.hits = L.static_hits;
            if (L.primary_stats_only) G.stats = nullptr;
            e = launch_shadow_t<MODE, EXTRA>(G, samples, std::min<long long>(max_hits, (long long)kStaticRes * kStaticRes * kStaticRes));
            if (e != hipSuccess) return e;
            hipLaunchKernelGGL(k_static_apply, dim3(blocks), dim3(256), 0, L.stream, (const uint8_t*)L.sc.shadow_cache, (const HitRec*)L.hits, L.counters + 14, samples);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        } else if (shadows) {
            // (queue entries are counted in whole tiles: see pad_tiles)
            PipelineLaunch T = L;                                    // this band's queue is tile-indexed (k_primary above)
            if (L.primary_stats_only) T.stats = nullptr;             // (the four statistics of sr_render count primary rays: no atomics per hit point)
            T.tile_queue_n2 = pad_tiles ? n2 : 0;
            T.tile_queue_rows = row_count;
            const long long max_hits = (long long)((row_count + 15) / 16 * 16) * ((L.fc.width + 15) / 16 * 16) * n2;
            // > 128 samples on the shaft path: one pass of the whole shadow stage per chunk of <= 128 samples, escape counts summed per
            // hit point (FrameConst.accum), then k_accum_finish.  (The walk does not depend on the samples; repeating it keeps the rounds'
            // bookkeeping -- undecided masks are 128 bits -- untouched.  The reference's count is the constant 100: ShadowMethod.cs:9.)
            const int S = L.fc.shadow_samples, chunks = (shaft_frame && L.fc.accum && S > 64 * kPacketSlots) ? (S + 64 * kPacketSlots - 1) / (64 * kPacketSlots) : 1;
            if (chunks == 1) T.fc.accum = nullptr;
            for (int c = 0; c < chunks; ++c) {
                if (chunks > 1) {
                    T.fc.shadow_samples = std::min(64 * kPacketSlots, S - c * 64 * kPacketSlots);
                    T.offsets = L.offsets + (size_t)c * 64 * kPacketSlots * 3;
                    if (c > 0 && (e = hipMemsetAsync(L.counters + 1, 0, 15 * sizeof(unsigned int), L.stream)) != hipSuccess) return e;
                    if (c > 0 && (e = hipMemsetAsync(L.counters + kHeadsShaft, 0, 8 * kTileHeadStride * sizeof(unsigned int), L.stream)) != hipSuccess) return e;
                }
                // primary rays through the reference tree / brute force, shadow rays on the own BVH (sr_api.cpp decides when that is allowed)
                if (MODE != MODE_BVH && L.shadows_on_bvh) e = launch_shadow_t<MODE_BVH, EXTRA>(T, samples, max_hits);
                else e = launch_shadow_t<MODE, EXTRA>(T, samples, max_hits);
                if (e != hipSuccess) return e;
            }
            if (chunks > 1) {
                const unsigned blocks = (unsigned)std::min<long long>((max_hits + 255) / 256, (long long)L.persistent_blocks);
                hipLaunchKernelGGL(k_accum_finish, dim3(blocks), dim3(256), 0, L.stream, T.fc, (const HitRec*)L.hits, L.counters, samples, S);
                if ((e = hipGetLastError()) != hipSuccess) return e;
            }
        }
        if (n2 > 1) {
            e0 = e1 = nullptr;
            if (L.get_events) L.get_events(L.user, K_RESOLVE, &e0, &e1);
            if (e0) { e = hipEventRecord(e0, L.stream); if (e != hipSuccess) return e; }
            long long npx = (long long)row_count * L.fc.width;
            hipLaunchKernelGGL(k_resolve, dim3((unsigned)((npx + 255) / 256)), dim3(256), 0, L.stream, L.fc, L.row_map, row_begin, row_count,
                               (const uint32_t*)L.samples, L.pixels);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
            if (e1) { e = hipEventRecord(e1, L.stream); if (e != hipSuccess) return e; }
        }
        if (L.band_done) L.band_done(L.user, (row_begin - L.row_first) / L.band_rows, row_begin, row_count, L.stream);
    }
    return hipSuccess;
}

hipError_t launch_pipeline(const PipelineLaunch& L) {
    const bool extra = L.sc.nextra > 0;
    switch (L.mode) {
        case MODE_REF: return extra ? launch_pipeline_t<MODE_REF, true>(L) : launch_pipeline_t<MODE_REF, false>(L);
        case MODE_BRUTE: return extra ? launch_pipeline_t<MODE_BRUTE, true>(L) : launch_pipeline_t<MODE_BRUTE, false>(L);
        case MODE_BVH: return extra ? launch_pipeline_t<MODE_BVH, true>(L) : launch_pipeline_t<MODE_BVH, false>(L);
        default: return hipErrorInvalidValue;
    }
}

size_t pipeline_hit_record_bytes() { return sizeof(HitRec); }
size_t pipeline_static_cells() { return (size_t)kStaticRes * kStaticRes * kStaticRes; }
int pipeline_round_cap(int round) { return kRoundCap[round]; }   // default list length of a round
// longest list the kernels handle (sr_debug_set hooks): 64 in the first round and wherever k_shadow_test keeps the list in the
// registers of one wave; the later rounds of the default path (k_shaft_coop, k_shadow_cls) read and write it in chunks
int pipeline_round_cap_max(int round) { return round == 0 ? 64 : 1024; }
int pipeline_bounce_lds_levels() { return kBounceLdsLevels; }
size_t pipeline_round_state_bytes() { return sizeof(RoundState); }
size_t pipeline_counter_bytes() { return (size_t)kCounterWords * sizeof(unsigned int); }
// 8x8-pixel tiles (one wave's work items) of the padded super-tile grid of a band, all sub-samples
size_t pipeline_tile_items(int width, int rows, int n2) { return (size_t)xcd_tile_grid(width, rows) * (size_t)n2 * 4; }

}  // namespace sr
