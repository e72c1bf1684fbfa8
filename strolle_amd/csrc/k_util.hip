// k_util.hip — measurement helper kernels that are not part of the pass graph.
#include <algorithm>
#include "k_common.h"

namespace st {
namespace ST_KNS {

// Grid-stride float4 copy: the device's own streaming ceiling (read + write), what /opt/skills/guides/MI355X_MICROARCH.md quotes
// as the achievable HBM rate (6.29 TB/s there). Four float4 per thread and iteration, nontemporal accesses, 16 blocks per CU: the best
// of tools/copy_probe.hip's variants on this pool (6.04 TB/s; one plain float4 per thread: 4.7-5.7; torch's copy_: 5.2-5.5).
// bench.py reports it beside the 8 TB/s spec figure: `frac` in the roofline object is always against the spec peak.
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_copy_float4(vf4* __restrict__ dst, const vf4* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 1024u;
    for (size_t i = (size_t)blockIdx.x * 1024u + threadIdx.x; i < n; i += stride) {
        vf4 v[4];
#pragma unroll
        for (int u = 0; u < 4; u++) if (i + (size_t)u * 256u < n) v[u] = __builtin_nontemporal_load(&src[i + (size_t)u * 256u]);
#pragma unroll
        for (int u = 0; u < 4; u++) if (i + (size_t)u * 256u < n) __builtin_nontemporal_store(v[u], &dst[i + (size_t)u * 256u]);
    }
}
void launch_copy_float4(float4* dst, const float4* src, size_t n, uint32_t blocks, hipStream_t s) {
    ST_KLAUNCH(k_copy_float4, dim3(blocks), dim3(256), s, reinterpret_cast<vf4*>(dst), reinterpret_cast<const vf4*>(src), n);
}

// Rectangle copy between two pitched images in units of T (uint4 when everything is 16-byte aligned, uint32_t otherwise): packs a
// rank's tile of the composed frame into a contiguous send buffer and unpacks received tiles into the root's frame (st_dist.cpp).
template <class T>
__global__ __launch_bounds__(256) void k_rect_copy(char* dst, size_t dst_pitch, const char* src, size_t src_pitch, uint32_t row_units, uint32_t rows) {
    const size_t n = (size_t)row_units * rows, stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t y = (uint32_t)(i / row_units), x = (uint32_t)(i - (size_t)y * row_units);
        reinterpret_cast<T*>(dst + (size_t)y * dst_pitch)[x] = reinterpret_cast<const T*>(src + (size_t)y * src_pitch)[x];
    }
}
void launch_rect_copy(void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes, uint32_t rows, hipStream_t s) {
    if (!row_bytes || !rows) return;
    const bool wide = ((reinterpret_cast<uintptr_t>(dst) | reinterpret_cast<uintptr_t>(src) | dst_pitch | src_pitch | row_bytes) & 15u) == 0u;
    const uint32_t units = (uint32_t)(row_bytes / (wide ? 16u : 4u));
    const size_t n = (size_t)units * rows;
    const uint32_t blocks = (uint32_t)std::min<size_t>((n + 255u) / 256u, 4096u);
    if (wide) ST_KLAUNCH(k_rect_copy<uint4>, dim3(blocks), dim3(256), s, static_cast<char*>(dst), dst_pitch, static_cast<const char*>(src), src_pitch, units, rows);
    else ST_KLAUNCH(k_rect_copy<uint32_t>, dim3(blocks), dim3(256), s, static_cast<char*>(dst), dst_pitch, static_cast<const char*>(src), src_pitch, units, rows);
}

}  // namespace ST_KNS
}  // namespace st
