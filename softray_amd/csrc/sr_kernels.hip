// sr_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the raytrace hot path.
//
// One lane = one pixel (or one ray); a wavefront (64 lanes) covers an 8x8 pixel tile so that the lanes'
// rays are coherent and triangle / node records are fetched once per wave (all lanes read the same
// address: one 64/128-byte request, broadcast).  All result-affecting arithmetic is FP64 in the
// reference's operand order, compiled with -ffp-contract=off (no FMA contraction): see DESIGN.md
// "Numerics".  No MFMA: the intersection math is branchy scalar FP64, not a contraction.
//
// Reference functions restated here (file:line relative to the reference root):
//   Plane.IntersectRay              Engine3D/Raytrace/Plane.cs:67-103
//   Triangle.IntersectRay           Engine3D/Raytrace/Triangle.cs:83-104
//   Sphere.IntersectRay             Engine3D/Raytrace/Sphere.cs:152-219
//   GeometryCollection.IntersectRay Engine3D/Raytrace/GeometryCollection.cs:44-69
//   AxisAlignedBox.ClipLineSegment  Engine3D/Raytrace/AxisAlignedBox.cs:111-216
//   SpatialSubdivision.IntersectRay / RecursiveRayTrace / GetClosestIntersection
//                                   Engine3D/Raytrace/SpatialSubdivision.cs:381-419,458-627,629-676
//   ShadingMethod                   Engine3D/Raytrace/ShadingMethod.cs:36-68,110-177
//   ShadowMethod                    Engine3D/Raytrace/ShadowMethod.cs:93-121,144-179
//   Renderer.RaytraceBlock / TraceRayComplex  Engine3D/Renderer.cs:1690-1829,1850-1885
#include "sr_trace.h"

namespace sr {

// --------------------------------------------------------------------------------------------------
// k_render: the whole of RaytraceBlock for one 16x16 pixel tile per 256-thread workgroup
// (4 wavefronts, each an 8x8 sub-tile).
// LDSGEOM: the triangle + extra records are staged into LDS once per workgroup and every lane reads
// them from there (wave-uniform address -> broadcast); used when they fit (small models, config 1/2).
// --------------------------------------------------------------------------------------------------
extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];

template <int MODE, bool EXTRA, bool LDSGEOM, bool STATS>
__global__ __launch_bounds__(256) void k_render(DevScene sc, FrameConst fc, const double* __restrict__ offsets,
                                                const int32_t* __restrict__ row_map, uint32_t* __restrict__ pixels,
                                                int stack_levels, unsigned long long* stats) {
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    const int col = blockIdx.x * 16 + (wave & 1) * 8 + (lane & 7);
    const int crow = blockIdx.y * 16 + (wave >> 1) * 8 + (lane >> 3);   // compact row of this launch

    int32_t* stack_mem = reinterpret_cast<int32_t*>(lds_raw);
    Stack st{stack_mem + tid, 256};
    const Rec128* tris = sc.tris;
    const Rec128* extra = sc.extra;
    if (LDSGEOM) {
        // stage [triangles | extra] behind the stacks, 16 bytes per lane per step (coalesced dwordx4)
        size_t off = ((size_t)stack_levels * 256 * 4 + 15) & ~(size_t)15;
        uint4* dst = reinterpret_cast<uint4*>(lds_raw + off);
        const int nt16 = (MODE != MODE_BVH ? sc.ntris : 0) * 8, ne16 = sc.nextra * 8;
        const uint4* srcT = reinterpret_cast<const uint4*>(sc.tris);
        const uint4* srcE = reinterpret_cast<const uint4*>(sc.extra);
        for (int i = tid; i < nt16; i += 256) dst[i] = srcT[i];
        for (int i = tid; i < ne16; i += 256) dst[nt16 + i] = srcE[i];
        __syncthreads();
        if (MODE != MODE_BVH) tris = reinterpret_cast<const Rec128*>(dst);
        extra = reinterpret_cast<const Rec128*>(dst + nt16);
    }

    Ctr prim = {0, 0, 0, 0}, sec = {0, 0, 0, 0};
    uint32_t nrays = 0;
    if (col < fc.width && crow < fc.num_rows) {
        const int row = row_map[crow];
        const int width = fc.width, height = fc.height, n = fc.sub_pixel_res;
        const D3 start = mk(fc.start_world[0], fc.start_world[1], fc.start_world[2]);
        uint32_t result;
        if (n == 1) {                                               // fast path, Renderer.cs:1722-1743
            D3 dv = mk(-((double)col / width - 0.5), -((double)row / height - 0.5) * fc.aspect, fc.fov_depth);
            D3 dw = mul3x3(fc.it, dv);
            nrays++;
            result = trace_camera_ray<MODE, EXTRA>(sc, fc, tris, extra, offsets, st, start, dw, prim, sec);
        } else {                                                    // n x n sub-rays, :1744-1826
            const bool blur = (fc.flags & 4u) != 0;
            int sumR = 0, sumG = 0, sumB = 0;
            D3 focal = mk(0, 0, 0);
            if (blur) {
                D3 dv = mk(-((double)col / width - 0.5), -((double)row / height - 0.5) * fc.aspect, fc.fov_depth);
                D3 dw = mul3x3(fc.it, dv);
                focal = dw * fc.focal_depth + start;
            }
            for (int sx = 0; sx < n; ++sx) {
                for (int sy = 0; sy < n; ++sy) {
                    double fx = (double)sx / (n - 1) - 0.5;
                    double fy = (double)sy / (n - 1) - 0.5;
                    D3 ss, dw;
                    if (blur) {
                        D3 sv = mk(fx / width * fc.focal_blur_strength, fy / height * fc.focal_blur_strength, -fc.position_z);
                        ss = mul3x3(fc.it, sv);
                        dw = focal - ss;
                    } else {
                        ss = start;
                        D3 dv = mk(-((col + fx) / width - 0.5), -((row + fy) / height - 0.5) * fc.aspect, fc.fov_depth);
                        dw = mul3x3(fc.it, dv);
                    }
                    nrays++;
                    uint32_t cc = trace_camera_ray<MODE, EXTRA>(sc, fc, tris, extra, offsets, st, ss, dw, prim, sec);
                    sumR += (cc >> 16) & 0xff;
                    sumG += (cc >> 8) & 0xff;
                    sumB += cc & 0xff;
                }
            }
            sumR /= n * n; sumG /= n * n; sumB /= n * n;
            result = (255u << 24) + ((uint32_t)(sumR & 0xff) << 16) + ((uint32_t)(sumG & 0xff) << 8) + (uint32_t)(sumB & 0xff);
        }
        const int out_row = (fc.strip_count > 0) ? crow : row;       // compact strip buffer, or the full surface
        pixels[(size_t)out_row * fc.width + col] = result;            // Surface.DrawPixel, Surface.cs:174-181
    }
    if (STATS) {
        uint32_t a = wave_sum(nrays), b = wave_sum(prim.geom), c2 = wave_sum(prim.nodes), d2 = wave_sum(prim.leaves);
        block_stat_add(&stats[0], &stats[1], &stats[2], &stats[3], a, b, c2, d2);
        a = wave_sum(sec.rays); b = wave_sum(sec.geom); c2 = wave_sum(sec.nodes); d2 = wave_sum(sec.leaves);
        block_stat_add(&stats[4], &stats[5], &stats[6], &stats[7], a, b, c2, d2);
    }
}

// --------------------------------------------------------------------------------------------------
// k_trace: IRayIntersectable.IntersectRay for a batch of rays (one lane per ray)
// --------------------------------------------------------------------------------------------------
template <int MODE, bool EXTRA>
__global__ __launch_bounds__(256) void k_trace(DevScene sc, long long n, const double* __restrict__ starts,
                                               const double* __restrict__ dirs, uint8_t* hit, double* ray_frac, double* pos,
                                               double* normal, uint32_t* color, int32_t* tri, int32_t* counters) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    Stack st{reinterpret_cast<int32_t*>(lds_raw) + threadIdx.x, 256};
    if (i >= n) return;
    D3 s = mk(starts[3 * i], starts[3 * i + 1], starts[3 * i + 2]);
    D3 d = mk(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    Hit h;
    h.t = 0; h.pos = mk(0, 0, 0); h.nrm = mk(0, 0, 0); h.color = 0; h.tri = -1;
    Ctr c = {0, 0, 0, 0};
    bool ok = root_intersect<MODE, false, EXTRA>(sc, sc.tris, sc.extra, st, s, d, h, c);
    if (hit) hit[i] = ok ? 1 : 0;
    if (ray_frac) ray_frac[i] = ok ? h.t : 0.0;
    if (pos) { pos[3 * i] = ok ? h.pos.x : 0.0; pos[3 * i + 1] = ok ? h.pos.y : 0.0; pos[3 * i + 2] = ok ? h.pos.z : 0.0; }
    if (normal) { normal[3 * i] = ok ? h.nrm.x : 0.0; normal[3 * i + 1] = ok ? h.nrm.y : 0.0; normal[3 * i + 2] = ok ? h.nrm.z : 0.0; }
    if (color) color[i] = ok ? h.color : 0u;
    if (tri) tri[i] = ok ? h.tri : -1;
    if (counters) { counters[3 * i] = (int32_t)c.geom; counters[3 * i + 1] = (int32_t)c.nodes; counters[3 * i + 2] = (int32_t)c.leaves; }
}

// --------------------------------------------------------------------------------------------------
// launchers
// --------------------------------------------------------------------------------------------------
const char* kernel_name(int id) {
    switch (id) {
        case K_RENDER: return "k_render";
        case K_TRACE: return "k_trace";
        case K_PRIMARY: return "k_primary";
        case K_SHADOW: return "k_shadow";
        case K_RESOLVE: return "k_resolve";
        case K_SHAFT: return "k_shaft";
        case K_FALLBACK: return "k_shadow_fallback";
        case K_BOUNCE: return "k_bounce";
        case K_SHAFT2: return "k_shaft_round2";
        case K_SHADOW2: return "k_shadow_round2";
        case K_POST: return "k_post_process";
        case K_ANTI_ALIAS: return "k_anti_alias";
        default: return "?";
    }
}

static int stack_levels_for(const DevScene& sc, int mode) {
    if (mode == MODE_REF) return sc.rdepth + 2;
    if (mode == MODE_BVH) return sc.bdepth + 2;
    return 1;
}

static const size_t kLdsGeomBudget = 96 * 1024;     // stay below 160 KiB/CU with room for a second workgroup

template <int MODE, bool EXTRA, bool LDSGEOM>
static hipError_t launch_render_t(const RenderLaunch& L, dim3 grid, size_t lds, int levels) {
    if (L.stats)
        hipLaunchKernelGGL((k_render<MODE, EXTRA, LDSGEOM, true>), grid, dim3(256), lds, L.stream, L.sc, L.fc, L.offsets,
                           L.row_map, L.pixels, levels, L.stats);
    else
        hipLaunchKernelGGL((k_render<MODE, EXTRA, LDSGEOM, false>), grid, dim3(256), lds, L.stream, L.sc, L.fc, L.offsets,
                           L.row_map, L.pixels, levels, L.stats);
    return hipGetLastError();
}

template <int MODE>
static hipError_t launch_render_m(const RenderLaunch& L) {
    dim3 grid((L.fc.width + 15) / 16, (L.fc.num_rows + 15) / 16);
    if (grid.x == 0 || grid.y == 0) return hipSuccess;
    int levels = stack_levels_for(L.sc, MODE);
    size_t stack_bytes = ((size_t)levels * 256 * 4 + 15) & ~(size_t)15;
    size_t geom_bytes = ((size_t)(MODE != MODE_BVH ? L.sc.ntris : 0) + (size_t)L.sc.nextra) * sizeof(Rec128);
    bool extra = L.sc.nextra > 0;
    bool ldsgeom = geom_bytes > 0 && stack_bytes + geom_bytes <= kLdsGeomBudget;
    size_t lds = stack_bytes + (ldsgeom ? geom_bytes : 0);
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    if (extra) return ldsgeom ? launch_render_t<MODE, true, true>(L, grid, lds, levels) : launch_render_t<MODE, true, false>(L, grid, lds, levels);
    return ldsgeom ? launch_render_t<MODE, false, true>(L, grid, lds, levels) : launch_render_t<MODE, false, false>(L, grid, lds, levels);
}

hipError_t launch_render(const RenderLaunch& L) {
    switch (L.mode) {
        case MODE_REF: return launch_render_m<MODE_REF>(L);
        case MODE_BRUTE: return launch_render_m<MODE_BRUTE>(L);
        case MODE_BVH: return launch_render_m<MODE_BVH>(L);
        default: return hipErrorInvalidValue;
    }
}

template <int MODE>
static hipError_t launch_trace_m(const TraceLaunch& L) {
    if (L.n <= 0) return hipSuccess;
    dim3 grid((unsigned)((L.n + 255) / 256));
    int levels = stack_levels_for(L.sc, MODE);
    size_t lds = (size_t)levels * 256 * 4;
    if (L.with_extra)
        hipLaunchKernelGGL((k_trace<MODE, true>), grid, dim3(256), lds, L.stream, L.sc, (long long)L.n, L.starts, L.dirs, L.hit,
                           L.ray_frac, L.pos, L.normal, L.color, L.tri, L.counters);
    else
        hipLaunchKernelGGL((k_trace<MODE, false>), grid, dim3(256), lds, L.stream, L.sc, (long long)L.n, L.starts, L.dirs, L.hit,
                           L.ray_frac, L.pos, L.normal, L.color, L.tri, L.counters);
    return hipGetLastError();
}

// ShadingMethod.IntersectRay's colour step for recorded intersections (sr_shade_points): one lane per point
__global__ __launch_bounds__(256) void k_shade_points(FrameConst fc, long long n, const double* __restrict__ pos, const double* __restrict__ nrm,
                                                      const uint32_t* __restrict__ color, uint32_t* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    out[i] = shade(fc, mk(pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]), mk(nrm[3 * i], nrm[3 * i + 1], nrm[3 * i + 2]), color[i]);
}

hipError_t launch_shade_points(const FrameConst& fc, long long n, const double* pos, const double* nrm, const uint32_t* color, uint32_t* out, hipStream_t stream) {
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_shade_points, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, fc, n, pos, nrm, color, out);
    return hipGetLastError();
}

hipError_t launch_trace(const TraceLaunch& L) {
    switch (L.mode) {
        case MODE_REF: return launch_trace_m<MODE_REF>(L);
        case MODE_BRUTE: return launch_trace_m<MODE_BRUTE>(L);
        case MODE_BVH: return launch_trace_m<MODE_BVH>(L);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace sr
