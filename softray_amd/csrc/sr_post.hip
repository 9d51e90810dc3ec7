// sr_post.hip -- the two surface passes Renderer.Render() runs after the raytrace (Renderer.cs:765-767):
// PostProcessImage's per-pixel colour functions (:819-865) and AntiAliasImage's box down-sample (:937-978).
// Both are pure integer work on the 32-bit surface and are HBM-bound: one read + one write per source pixel
// (8 B/pixel in place) for the colour functions, 4 B read per source pixel + 4/res^2 B written for the down-sample.
#include <hip/hip_runtime.h>

#include "sr_device.h"

namespace sr {

namespace {

__device__ __forceinline__ uint32_t style_func(uint32_t x, int style, uint32_t background) {
    switch (style) {
        case 1:  return ((x & 0xffffu) << 8) + ((x >> 16) & 0xffu);                                     // ColorShuffle  :827-830
        case 2:  return x == background ? background : 0x00ffffffu - x;                                 // Negative      :832-834
        case 3:  return ((x >> 8) & 0xff0000u) + ((x >> 16) & 0xff00u) + ((x >> 24) & 0xffu);           // DepthSmooth   :844-848
        case 4:  return ((x >> 24) & 0xffu) * 111u;                                                     // DepthBanded   :859-863
        default: return x;
    }
}

// 4 pixels per lane (dwordx4), grid-stride; the scalar tail is handled by the same lanes.
// `head` (< 4) leading pixels bring the pointer to 16-byte alignment and are done by the first lanes.
__global__ void __launch_bounds__(256) k_post_process(uint32_t* __restrict__ px, long long count, int head, int style, uint32_t background) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid < head) px[gid] = style_func(px[gid], style, background);
    px += head;
    count -= head;
    const long long quads = count >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    uint4* q = reinterpret_cast<uint4*>(px);
    for (long long i = gid; i < quads; i += stride) {
        uint4 v = q[i];
        v.x = style_func(v.x, style, background);
        v.y = style_func(v.y, style, background);
        v.z = style_func(v.z, style, background);
        v.w = style_func(v.w, style, background);
        q[i] = v;
    }
    const long long tail = (quads << 2) + gid;
    if (tail < count) px[tail] = style_func(px[tail], style, background);
}

// One lane per destination pixel; a wave reads res consecutive runs of 64*res source pixels (coalesced per sub-row).
__global__ void __launch_bounds__(256) k_anti_alias(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int dst_w, int dst_h, int res) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63);
    const int dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= dst_w || dy >= dst_h) return;
    const long long src_w = (long long)dst_w * res;
    int sum_r = 0, sum_g = 0, sum_b = 0;
    for (int sy = 0; sy < res; ++sy) {
        const uint32_t* row = src + ((long long)dy * res + sy) * src_w + (long long)dx * res;
        for (int sx = 0; sx < res; ++sx) {
            const uint32_t c = row[sx];
            sum_r += (int)((c >> 16) & 0xffu);
            sum_g += (int)((c >> 8) & 0xffu);
            sum_b += (int)(c & 0xffu);
        }
    }
    const int n = res * res;                                   // integer division, :970-972
    sum_r /= n; sum_g /= n; sum_b /= n;
    dst[(long long)dy * dst_w + dx] = (255u << 24) + ((uint32_t)(sum_r & 0xff) << 16) + ((uint32_t)(sum_g & 0xff) << 8) + (uint32_t)(sum_b & 0xff);
}

}  // namespace

hipError_t launch_post_process(uint32_t* d_pixels, long long count, int style, uint32_t background, int num_cus, hipStream_t stream) {
    if (count <= 0 || style == 0) return hipSuccess;
    int head = (int)(((16 - ((uintptr_t)d_pixels & 15)) & 15) >> 2);
    if (head > count) head = (int)count;
    long long want = ((count >> 2) + 255) / 256;
    if (want < 1) want = 1;
    long long cap = (long long)(num_cus > 0 ? num_cus : 256) * 16;
    const unsigned blocks = (unsigned)(want < cap ? want : cap);
    hipLaunchKernelGGL(k_post_process, dim3(blocks), dim3(256), 0, stream, d_pixels, count, head, style, background);
    return hipGetLastError();
}

hipError_t launch_anti_alias(const uint32_t* d_src, uint32_t* d_dst, int dst_w, int dst_h, int res, hipStream_t stream) {
    if (dst_w <= 0 || dst_h <= 0) return hipSuccess;
    dim3 grid((unsigned)((dst_w + 63) / 64), (unsigned)((dst_h + 3) / 4));
    hipLaunchKernelGGL(k_anti_alias, grid, dim3(256), 0, stream, d_src, d_dst, dst_w, dst_h, res);
    return hipGetLastError();
}

}  // namespace sr
