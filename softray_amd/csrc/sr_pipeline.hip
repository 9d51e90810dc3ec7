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

namespace sr {

extern __shared__ __attribute__((aligned(16))) unsigned char lds_pipe[];

// one queued surface point (64 B): what ShadowMethod.IntersectRay needs from the primary hit
struct alignas(16) HitRec {
    double   pos[3];
    double   nrm[3];
    uint32_t sample;     // index into the sample-colour buffer
    uint32_t pad[3];
};
static_assert(sizeof(HitRec) == 64, "HitRec must be 64 bytes");

__device__ __forceinline__ unsigned long long lanemask_lt() {
    unsigned lane = threadIdx.x & 63u;
    return lane == 0 ? 0ull : (~0ull >> (64u - lane));
}

// --------------------------------------------------------------------------------------------------
// k_primary
// --------------------------------------------------------------------------------------------------
template <int MODE, bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_primary(DevScene sc, FrameConst fc, const int32_t* __restrict__ row_map, int row_begin,
                                                 int row_count, uint32_t* __restrict__ samples, HitRec* __restrict__ hits,
                                                 unsigned int* __restrict__ hit_count, unsigned long long* stats) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int col = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int brow = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);      // row inside this band
    Stack st{reinterpret_cast<int32_t*>(lds_pipe) + tid, 256};
    const bool live = col < fc.width && brow < row_count;
    const int crow = row_begin + brow;                                       // compact row of the frame
    const int row = live ? row_map[crow] : 0;
    const int n = fc.sub_pixel_res, n2 = n * n;
    const bool shadows = (fc.flags & 2u) != 0;
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
    Ctr prim = {0, 0, 0, 0};
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
        if (live) {
            prim.rays++;
            ok = root_intersect<MODE, false, EXTRA>(sc, sc.tris, sc.extra, st, ss, dw, h, prim);
            uint32_t color = fc.background;
            if (ok) color = (fc.flags & 1u) ? shade(fc, h.pos, h.nrm, h.color) : h.color;
            samples[sbase + si] = color;
        }
        if (shadows) {                                                       // active-ray compaction
            unsigned long long m = __ballot(ok);
            if (m) {
                unsigned int base = 0;
                const int leader = __ffsll((long long)m) - 1;
                if (lane == leader) base = atomicAdd(hit_count, (unsigned int)__popcll(m));
                base = __shfl(base, leader, 64);
                if (ok) {
                    HitRec r;
                    r.pos[0] = h.pos.x; r.pos[1] = h.pos.y; r.pos[2] = h.pos.z;
                    r.nrm[0] = h.nrm.x; r.nrm[1] = h.nrm.y; r.nrm[2] = h.nrm.z;
                    r.sample = (uint32_t)(sbase + si);
                    r.pad[0] = r.pad[1] = r.pad[2] = 0;
                    hits[base + (unsigned int)__popcll(m & lanemask_lt())] = r;
                }
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(prim.rays), b = wave_sum(prim.geom), c2 = wave_sum(prim.nodes), d2 = wave_sum(prim.leaves);
        if (lane == 0) {
            atomicAdd(&stats[0], (unsigned long long)a);
            atomicAdd(&stats[1], (unsigned long long)b);
            atomicAdd(&stats[2], (unsigned long long)c2);
            atomicAdd(&stats[3], (unsigned long long)d2);
        }
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
                                                unsigned int* __restrict__ work_head, uint32_t* __restrict__ samples,
                                                int stack_levels, unsigned long long* stats) {
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
    uint32_t sample = 0;
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
                    const HitRec r = hits[idx];
                    D3 pos = mk(r.pos[0], r.pos[1], r.pos[2]), nrm = mk(r.nrm[0], r.nrm[1], r.nrm[2]);
                    shadowEnd = pos + nrm * 0.001;                       // shadowProbeOffset, ShadowMethod.cs:10,151
                    sample = r.sample;
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
                samples[sample] = modulate(samples[sample], to_byte(frac * 255));
            }
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom), c2 = wave_sum(sec.nodes), d2 = wave_sum(sec.leaves);
        if (lane == 0) {
            atomicAdd(&stats[4], (unsigned long long)a);
            atomicAdd(&stats[5], (unsigned long long)b);
            atomicAdd(&stats[6], (unsigned long long)c2);
            atomicAdd(&stats[7], (unsigned long long)d2);
        }
    }
}

// --------------------------------------------------------------------------------------------------
// k_shadow_packet (own BVH + point light): one WAVEFRONT per queued hit, lanes = area-light samples.
//
// All S sample rays of one surface point end in the same point E' and start on a sphere of radius R
// around the light: P_i(t) = C(t) + (1 - t) * off_i with the centre ray C(t) = (1-t) L + t E'.  A
// triangle can be hit by ANY of them at parameter t only if its box is within (1-t) R of C(t).  So the
// wave walks the BVH ONCE with that shaft (fp32, conservative, children nearest to E' first), gathers
// the triangles of the leaves it touches in chunks, and every lane tests its own sample rays against a
// chunk with the reference's exact FP64 arithmetic.  Samples that are blocked drop out; when none is
// left the hit is finished early (fully shadowed points: a handful of nodes); when the shaft is
// exhausted the remaining samples have provably no occluder (ShadowMethod.cs:170: nearest hit > 1.0 or
// none).  No per-sample traversal at all; the result is exactly that of S independent any-hit searches.
// --------------------------------------------------------------------------------------------------
struct SampleRay {
    D3     s, d;        // clipped start (SpatialSubdivision.cs:394) and direction
    double offset;      // rayFracOffset (:401)
};

__device__ __forceinline__ bool prepare_sample(const DevScene& sc, D3 rs, D3 rd, SampleRay& r) {
    D3 end = rs + rd * 10000.0;
    r.s = rs;
    r.d = rd;
    if (!clip_segment(sc.root, r.s, end)) return false;
    r.offset = length(rs - r.s) / length(rd);
    return true;
}

template <bool EXTRA>
__device__ __forceinline__ bool extras_block(const DevScene& sc, D3 rs, D3 rd, Ctr& c) {
    for (int i = 0; i < sc.nextra; ++i) {
        const Rec128* r = &sc.extra[i];
        double t; D3 pos, nrm;
        bool ok;
        int kind = r->aux;
        if (kind == 0) ok = sphere_hit(r->p, rs, rd, t, pos, nrm);
        else if (kind == 1) ok = plane_hit(r->p, rs, rd, t, pos);
        else ok = tri_hit(r->p, rs, rd, t, pos);
        c.geom++;
        if (ok && t <= 1.0) return true;
    }
    return false;
}

constexpr int kPacketSlots = 2;          // samples per lane: S <= 128
constexpr int kChunk = 16;               // candidate triangles gathered before the lanes test them
constexpr int kPacketStack = 96;         // shaft-traversal stack entries per wave (node, t)

template <bool EXTRA, bool STATS>
__global__ __launch_bounds__(256) void k_shadow_packet(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                       const HitRec* __restrict__ hits, const unsigned int* __restrict__ hit_count,
                                                       uint32_t* __restrict__ samples, unsigned long long* stats) {
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // per-wave LDS: candidate chunk + shaft stack
    int32_t* wl = reinterpret_cast<int32_t*>(lds_pipe) + wave * (kChunk + 8 + 2 * kPacketStack);
    int32_t* cand = wl;
    int32_t* stk_n = wl + kChunk + 8;
    float*   stk_t = reinterpret_cast<float*>(stk_n + kPacketStack);

    const int S = fc.shadow_samples;
    const unsigned int total = *hit_count;
    const D3 lpos = mk(fc.light_pos_model[0], fc.light_pos_model[1], fc.light_pos_model[2]);
    const float R = (float)fc.light_radius * 1.00001f + 1e-30f;
    float ext = 0.0f;
    for (int a = 0; a < 3; ++a) ext = fmaxf(ext, (float)(sc.root.max[a] - sc.root.min[a]));
    const float pad = ext * 3.0517578125e-5f;                              // 2^-15 * extent (boxes carry 2^-16 already)
    const float lox = (float)(lpos.x - sc.root.centre[0]), loy = (float)(lpos.y - sc.root.centre[1]), loz = (float)(lpos.z - sc.root.centre[2]);

    // sample offsets of this lane (slot k = sample lane + 64 k)
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
    for (unsigned int h = blockIdx.x * 4u + (unsigned)wave; h < total; h += nwaves) {
        const HitRec rec = hits[h];
        const D3 pos = mk(rec.pos[0], rec.pos[1], rec.pos[2]), nrm = mk(rec.nrm[0], rec.nrm[1], rec.nrm[2]);
        const D3 E = pos + nrm * 0.001;                                    // shadowRayEnd, ShadowMethod.cs:151

        SampleRay ray[kPacketSlots];
        bool alive[kPacketSlots];      // sample still undecided (not blocked yet, may still be hit)
        bool escaped[kPacketSlots];    // sample that is decided as reaching the surface so far
#pragma unroll
        for (int k = 0; k < kPacketSlots; ++k) {
            D3 rs = lpos + off[k];
            D3 rd = E - rs;
            alive[k] = false;
            escaped[k] = valid[k];
            if (valid[k]) {
                sec.rays++;
                bool blocked = false;
                if (EXTRA) blocked = extras_block<EXTRA>(sc, rs, rd, sec);
                if (blocked) escaped[k] = false;
                else alive[k] = prepare_sample(sc, rs, rd, ray[k]);       // outside the root box: nothing can block it
            }
        }

        // ---- shaft walk (wave-uniform) ----
        const float cdx = (float)(E.x - lpos.x), cdy = (float)(E.y - lpos.y), cdz = (float)(E.z - lpos.z);
        const float ix = 1.0f / cdx, iy = 1.0f / cdy, iz = 1.0f / cdz;
        const float tmax = 1.0f + 1e-5f;
        int sp = 0, ncand = 0;
        int32_t ni = 0;
        float nt = 0.0f;               // lower bound of t inside the current subtree
        bool have = __any(alive[0] || alive[1]);
        while (have) {
            ni = __builtin_amdgcn_readfirstlane(ni);                       // wave-uniform walk: scalar node fetch
            nt = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(nt)));
            const BvhNode n = sc.bnodes[ni];
            sec.nodes++;
            const float r = R * fminf(1.0f, fmaxf(0.0f, 1.0f - nt + 1e-5f)) + pad;
            float elo[3], ehi[3];
            float t0, x0, t1, x1;
#pragma unroll
            for (int a = 0; a < 3; ++a) { elo[a] = n.lo0[a] - r; ehi[a] = n.hi0[a] + r; }
            slab(elo, ehi, lox, loy, loz, ix, iy, iz, t0, x0);
#pragma unroll
            for (int a = 0; a < 3; ++a) { elo[a] = n.lo1[a] - r; ehi[a] = n.hi1[a] + r; }
            slab(elo, ehi, lox, loy, loz, ix, iy, iz, t1, x1);
            t0 = fmaxf(t0, nt); t1 = fmaxf(t1, nt);
            x0 = fminf(x0, tmax); x1 = fminf(x1, tmax);
            const bool h0 = n.n0 >= 0 && t0 <= x0, h1 = n.n1 >= 0 && t1 <= x1;
            if (h0 && n.n0 > 0) { for (int k = 0; k < n.n0; ++k) if (ncand < kChunk + 8) { if (lane == 0) cand[ncand] = n.c0 + k; ncand++; } sec.leaves++; }
            if (h1 && n.n1 > 0) { for (int k = 0; k < n.n1; ++k) if (ncand < kChunk + 8) { if (lane == 0) cand[ncand] = n.c1 + k; ncand++; } sec.leaves++; }
            const bool i0 = h0 && n.n0 == 0, i1 = h1 && n.n1 == 0;
            bool popped = false;
            if (i0 && i1) {
                // visit the child whose interval reaches closest to E' (larger exit t) first
                const bool first0 = x0 >= x1;
                if (sp < kPacketStack) { if (lane == 0) { stk_n[sp] = first0 ? n.c1 : n.c0; stk_t[sp] = first0 ? t1 : t0; } sp++; }
                ni = first0 ? n.c0 : n.c1; nt = first0 ? t0 : t1;
            } else if (i0) { ni = n.c0; nt = t0; }
            else if (i1) { ni = n.c1; nt = t1; }
            else popped = true;

            const bool flush = ncand >= kChunk || (popped && sp == 0);
            if (flush && ncand > 0) {
                __builtin_amdgcn_wave_barrier();                           // lane 0's LDS writes are ordered before the reads below
                // ---- every lane tests its undecided samples against the gathered triangles (exact FP64) ----
                for (int k = 0; k < ncand; ++k) {
                    const Rec128* tr = &sc.btris[cand[k]];
#pragma unroll
                    for (int q = 0; q < kPacketSlots; ++q) {
                        if (alive[q]) {
                            double t; D3 hp;
                            sec.geom++;
                            if (tri_hit(tr->p, ray[q].s, ray[q].d, t, hp) && inside(sc.root.lo, sc.root.hi, hp) && (t + ray[q].offset <= 1.0)) {
                                alive[q] = false;
                                escaped[q] = false;
                            }
                        }
                    }
                }
                ncand = 0;
                have = __any(alive[0] || alive[1]);
                if (!have) break;
            }
            if (popped) {
                if (sp == 0) break;
                --sp;
                __builtin_amdgcn_wave_barrier();
                ni = __builtin_amdgcn_readfirstlane(stk_n[sp]);
                nt = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(stk_t[sp])));
            }
        }
        const int esc = (int)__popcll(__ballot(escaped[0])) + (int)__popcll(__ballot(escaped[1]));
        if (lane == 0) {
            double frac = (double)esc / (double)S;                         // ShadowMethod.IntersectRay :113-119
            samples[rec.sample] = modulate(samples[rec.sample], to_byte(frac * 255));
        }
    }
    if (STATS) {
        uint32_t a = wave_sum(sec.rays), b = wave_sum(sec.geom);
        if (lane == 0) {
            atomicAdd(&stats[4], (unsigned long long)a);
            atomicAdd(&stats[5], (unsigned long long)b);
            atomicAdd(&stats[6], (unsigned long long)sec.nodes);          // wave-uniform walk: counted once per wave
            atomicAdd(&stats[7], (unsigned long long)sec.leaves);
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
// launchers
// --------------------------------------------------------------------------------------------------
static int pipe_stack_levels(const DevScene& sc, int mode) {
    if (mode == MODE_REF) return sc.rdepth + 2;
    if (mode == MODE_BVH) return 2 * (sc.bdepth + 2);
    return 1;
}

template <int MODE, bool EXTRA>
static hipError_t launch_primary_t(const PipelineLaunch& L, int row_begin, int row_count, uint32_t* samples) {
    dim3 grid((L.fc.width + 15) / 16, (row_count + 15) / 16);
    size_t lds = (size_t)pipe_stack_levels(L.sc, MODE) * 256 * 4;
    if (L.stats)
        hipLaunchKernelGGL((k_primary<MODE, EXTRA, true>), grid, dim3(256), lds, L.stream, L.sc, L.fc, L.row_map, row_begin, row_count,
                           samples, (HitRec*)L.hits, L.hit_count, L.stats);
    else
        hipLaunchKernelGGL((k_primary<MODE, EXTRA, false>), grid, dim3(256), lds, L.stream, L.sc, L.fc, L.row_map, row_begin, row_count,
                           samples, (HitRec*)L.hits, L.hit_count, L.stats);
    return hipGetLastError();
}

template <bool EXTRA>
static hipError_t launch_shadow_packet_t(const PipelineLaunch& L, uint32_t* samples, long long max_hits) {
    size_t lds = 4 * (size_t)(kChunk + 8 + 2 * kPacketStack) * 4;
    long long want = (max_hits + 3) / 4;
    unsigned blocks = (unsigned)std::min<long long>(want, (long long)L.persistent_blocks * 2);
    if (blocks == 0) return hipSuccess;
    if (L.stats)
        hipLaunchKernelGGL((k_shadow_packet<EXTRA, true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits,
                           L.hit_count, samples, L.stats);
    else
        hipLaunchKernelGGL((k_shadow_packet<EXTRA, false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits,
                           L.hit_count, samples, L.stats);
    return hipGetLastError();
}

template <int MODE, bool EXTRA>
static hipError_t launch_shadow_t(const PipelineLaunch& L, uint32_t* samples, long long max_hits) {
    if (MODE == MODE_BVH && (L.fc.flags & 8u) && L.fc.shadow_samples <= 64 * kPacketSlots && L.sc.bdepth + 2 <= kPacketStack && !L.per_lane_shadows)
        return launch_shadow_packet_t<EXTRA>(L, samples, max_hits);
    int levels = pipe_stack_levels(L.sc, MODE);
    size_t lds = (((size_t)levels * 256 * 4 + 15) & ~(size_t)15) + (size_t)L.fc.shadow_samples * 3 * sizeof(double);
    long long want = (max_hits + 255) / 256;
    unsigned blocks = (unsigned)std::min<long long>(want, (long long)L.persistent_blocks);
    if (blocks == 0) return hipSuccess;
    if (L.stats)
        hipLaunchKernelGGL((k_shadow<MODE, EXTRA, true>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits,
                           L.hit_count, L.work_head, samples, levels, L.stats);
    else
        hipLaunchKernelGGL((k_shadow<MODE, EXTRA, false>), dim3(blocks), dim3(256), lds, L.stream, L.sc, L.fc, L.offsets, (const HitRec*)L.hits,
                           L.hit_count, L.work_head, samples, levels, L.stats);
    return hipGetLastError();
}

template <int MODE, bool EXTRA>
static hipError_t launch_pipeline_t(const PipelineLaunch& L) {
    const int n2 = L.fc.sub_pixel_res * L.fc.sub_pixel_res;
    const bool shadows = (L.fc.flags & 2u) != 0;
    // rows are processed in bands so that the hit queue / sample buffer stay within their allocation
    for (int row_begin = 0; row_begin < L.fc.num_rows; row_begin += L.band_rows) {
        int row_count = std::min(L.band_rows, L.fc.num_rows - row_begin);
        uint32_t* samples = (n2 == 1) ? L.pixels : L.samples;
        hipError_t e;
        if (shadows) {
            e = hipMemsetAsync(L.hit_count, 0, 2 * sizeof(unsigned int), L.stream);     // hit_count, work_head (adjacent)
            if (e != hipSuccess) return e;
        }
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (L.get_events) L.get_events(L.user, K_PRIMARY, &e0, &e1);
        if (e0) { e = hipEventRecord(e0, L.stream); if (e != hipSuccess) return e; }
        e = launch_primary_t<MODE, EXTRA>(L, row_begin, row_count, samples);
        if (e != hipSuccess) return e;
        if (e1) { e = hipEventRecord(e1, L.stream); if (e != hipSuccess) return e; }
        if (shadows) {
            e0 = e1 = nullptr;
            if (L.get_events) L.get_events(L.user, K_SHADOW, &e0, &e1);
            if (e0) { e = hipEventRecord(e0, L.stream); if (e != hipSuccess) return e; }
            e = launch_shadow_t<MODE, EXTRA>(L, samples, (long long)row_count * L.fc.width * n2);
            if (e != hipSuccess) return e;
            if (e1) { e = hipEventRecord(e1, L.stream); if (e != hipSuccess) return e; }
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

}  // namespace sr
