// copy_probe.hip — which grid-stride float4 copy reaches this device's streaming ceiling? (hipcc --offload-arch=gfx950 -O3 tools/copy_probe.hip)
// Variants: blocks per CU, float4s per thread and iteration, plain vs nontemporal accesses. Prints GB/s (read + write) of a 1 GiB copy.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float vf4 __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k_copy(vf4* __restrict__ dst, const vf4* __restrict__ src, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256u * U;
    for (size_t i = (size_t)blockIdx.x * 256u * U + threadIdx.x; i < n; i += stride) {
        vf4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + (size_t)u * 256u < n) v[u] = NT ? __builtin_nontemporal_load(&src[i + (size_t)u * 256u]) : src[i + (size_t)u * 256u];
#pragma unroll
        for (int u = 0; u < U; u++) if (i + (size_t)u * 256u < n) { if (NT) __builtin_nontemporal_store(v[u], &dst[i + (size_t)u * 256u]); else dst[i + (size_t)u * 256u] = v[u]; }
    }
}
template <int U, bool NT>
static void run(float4* d, const float4* s, size_t n, int blocks_per_cu) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int it = 0; it < 6; it++) {
        hipEventRecord(a); hipLaunchKernelGGL((k_copy<U, NT>), dim3(256 * blocks_per_cu), dim3(256), 0, 0, reinterpret_cast<vf4*>(d), reinterpret_cast<const vf4*>(s), n); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (it && ms < best) best = ms;
    }
    printf("unroll %d %-11s %2d blocks/CU: %7.1f GB/s\n", U, NT ? "nontemporal" : "plain", blocks_per_cu, 2.0 * n * 16 / (best * 1e-3) / 1e9);
}
int main() {
    const size_t n = ((size_t)1 << 30) / 16;
    float4 *s, *d; hipMalloc(&s, n * 16); hipMalloc(&d, n * 16); hipMemset(s, 0x3c, n * 16);
    for (int bpc : {4, 8, 16, 32}) { run<1, false>(d, s, n, bpc); run<4, false>(d, s, n, bpc); run<4, true>(d, s, n, bpc); run<8, true>(d, s, n, bpc); }
    hipEventRecord(0); hipMemcpy(d, s, n * 16, hipMemcpyDeviceToDevice);
    return 0;
}
