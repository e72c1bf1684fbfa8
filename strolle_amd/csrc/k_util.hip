// k_util.hip — measurement helper kernels that are not part of the pass graph.
#include "k_common.h"

namespace st {
namespace ST_KNS {

// Grid-stride float4 copy: the device's own streaming ceiling (read + write), what /opt/skills/guides/MI355X_MICROARCH.md quotes
// as the achievable HBM rate (6.29 TB/s there). bench.py reports it beside torch's copy_ and beside the 8 TB/s spec figure:
// `frac` in the roofline object is always against the spec peak.
__global__ __launch_bounds__(256) void k_copy_float4(float4* __restrict__ dst, const float4* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) dst[i] = src[i];
}
void launch_copy_float4(float4* dst, const float4* src, size_t n, uint32_t blocks, hipStream_t s) {
    ST_KLAUNCH(k_copy_float4, dim3(blocks), dim3(256), s, dst, src, n);
}

}  // namespace ST_KNS
}  // namespace st
