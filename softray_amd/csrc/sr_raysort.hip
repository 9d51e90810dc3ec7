// sr_raysort.hip -- coherence for the incoherent: the ray queue of a bounce level is ordered by (cell of the origin, direction
// octant) before k_bounce walks it, so that the lanes of a wave start in the same part of the tree and head the same way
// (neighbouring lanes then fetch the same nodes and records: cache hits instead of HBM round trips).  The order is a
// permutation of queue indices (the queue itself stays where it is); which lane traces which ray changes nothing a pixel
// depends on.  Key = 7 bits per axis of the origin's cell in the root box, Morton-interleaved, with the 3 bits of direction signs between the coarse 15 and the fine 6 cell bits
// (24 bits: three 8-bit radix passes of rocPRIM's device sort through hipCUB).  Entries beyond the queue's device-side count
// get the largest key and stay at the end.
#include <hipcub/hipcub.hpp>

#include "sr_device.h"

namespace sr {
namespace {

__device__ __forceinline__ unsigned int spread7(unsigned int v) {          // 7 bits -> every third bit
    v &= 0x7fu;
    v = (v | (v << 8)) & 0x700fu;
    v = (v | (v << 4)) & 0x430c3u;
    v = (v | (v << 2)) & 0x49249u;
    return v;
}

// queue records are 64 bytes: double origin[3], double direction[3], ... (HitRec of sr_pipeline.hip)
__global__ __launch_bounds__(256) void k_ray_keys(const double* __restrict__ queue, const unsigned int* __restrict__ count, unsigned int cap,
                                                  double lx, double ly, double lz, double sx, double sy, double sz,
                                                  unsigned int* __restrict__ keys, unsigned int* __restrict__ idx) {
    const unsigned int i = blockIdx.x * 256u + threadIdx.x;
    if (i >= cap) return;
    unsigned int key = 0xffffffffu;
    if (i < *count) {
        const double* q = queue + (size_t)i * 8;
        const int cx = min(127, max(0, (int)((q[0] - lx) * sx))), cy = min(127, max(0, (int)((q[1] - ly) * sy))), cz = min(127, max(0, (int)((q[2] - lz) * sz)));
        const unsigned int oct = (q[3] < 0.0 ? 1u : 0u) | (q[4] < 0.0 ? 2u : 0u) | (q[5] < 0.0 ? 4u : 0u);
        // the direction octant sits between the coarse cell bits (32^3 cells) and the fine ones: rays of a coarse cell that head the same way
        // come first together, then spread over the cell (against octant-last: -1 %, against octant-first: -6 % at C5)
        const unsigned int mort = (spread7((unsigned)cx) << 2) | (spread7((unsigned)cy) << 1) | spread7((unsigned)cz);
        key = ((mort >> 6) << 9) | (oct << 6) | (mort & 63u);
    }
    keys[i] = key;
    idx[i] = i;
}

}  // namespace

size_t ray_sort_temp_bytes(unsigned int cap) {
    size_t bytes = 0;
    (void)hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, (const unsigned int*)nullptr, (unsigned int*)nullptr, (const unsigned int*)nullptr,
                                             (unsigned int*)nullptr, (int)cap, 0, 24);
    return bytes;
}

// order_out[j] = index of the j-th ray in key order; keys / keys2 / idx: scratch of `cap` entries each
hipError_t ray_sort(const void* queue, const unsigned int* d_count, unsigned int cap, const RootBox& root, unsigned int* keys, unsigned int* keys2,
                    unsigned int* idx, unsigned int* order_out, void* temp, size_t temp_bytes, hipStream_t stream) {
    if (cap == 0) return hipSuccess;
    double s[3];
    for (int a = 0; a < 3; ++a) { const double e = root.max[a] - root.min[a]; s[a] = e > 0 ? 128.0 / e : 0.0; }
    hipLaunchKernelGGL(k_ray_keys, dim3((cap + 255u) / 256u), dim3(256), 0, stream, (const double*)queue, d_count, cap, root.min[0], root.min[1], root.min[2],
                       s[0], s[1], s[2], keys, idx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // sorting all 32 key bits' low 24 keeps the 0xffffffff padding last as well (its low 24 bits are all ones, the largest digit in every pass)
    return hipcub::DeviceRadixSort::SortPairs(temp, temp_bytes, (const unsigned int*)keys, keys2, (const unsigned int*)idx, order_out, (int)cap, 0, 24, stream);
}

}  // namespace sr
