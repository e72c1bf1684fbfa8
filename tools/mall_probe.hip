// mall_probe.hip — how fast does MI355X move a working set that fits in the 256 MB Infinity Cache (or the 8 x 4 MB L2s)
// compared with one that does not? Answers whether keeping a pass's output resident for the next pass (row-band
// pipelining of the frame) could beat the device-copy ceiling bench.py measures on a 1 GiB buffer.
//
//   hipcc --offload-arch=gfx950 -O3 tools/mall_probe.hip -o /tmp/mall_probe && timeout 120 /tmp/mall_probe
//
// For each working-set size S: a float4 copy src -> dst (S/2 each) repeated back to back, a read-only pass over S and a
// write-only pass over S; prints GB/s of bytes moved (read + written).
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_copy(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i];
}
__global__ void k_read(const float4* __restrict__ src, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 v = make_float4(0, 0, 0, 0);
    if (i < n) v = src[i];
    if (v.x + v.y + v.z + v.w == 123456.789f) out[0] = 1.0f;
}
__global__ void k_write(float4* dst, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = make_float4(1.0f, 2.0f, 3.0f, (float)(i & 1023u));
}
// producer -> consumer: pass A writes plane P from plane Q, pass B reads P and writes Q (what two passes of a frame do)
int main() {
    const size_t max_bytes = (size_t)1 << 30;
    float4* buf; float* out;
    CHECK(hipMalloc(&buf, max_bytes));
    CHECK(hipMalloc(&out, 256));
    CHECK(hipMemset(buf, 0, max_bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int reps = 40;
    for (size_t mb = 4; mb <= 1024; mb *= 2) {
        const size_t bytes = mb << 20, texels = bytes / 16, half = texels / 2;
        float ms_copy, ms_read, ms_write;
        // copy (ping-pong so that each repetition reads what the previous one wrote)
        for (int warm = 0; warm < 2; warm++) {
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) {
                const float4* s = (r & 1) ? buf + half : buf;
                float4* d = (r & 1) ? buf : buf + half;
                hipLaunchKernelGGL(k_copy, dim3((unsigned)((half + 255) / 256)), dim3(256), 0, 0, s, d, half);
            }
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_copy, e0, e1));
        }
        for (int warm = 0; warm < 2; warm++) {
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_read, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, 0, buf, texels, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_read, e0, e1));
        }
        for (int warm = 0; warm < 2; warm++) {
            CHECK(hipEventRecord(e0));
            for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_write, dim3((unsigned)((texels + 255) / 256)), dim3(256), 0, 0, buf, texels);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_write, e0, e1));
        }
        const double gb = (double)bytes * reps / 1e9;
        printf("working_set_MB %5zu  copy %8.1f GB/s  read %8.1f GB/s  write %8.1f GB/s   (us per launch: copy %.1f read %.1f write %.1f)\n", mb,
               gb / (ms_copy * 1e-3), gb / (ms_read * 1e-3), gb / (ms_write * 1e-3), ms_copy * 1e3 / reps, ms_read * 1e3 / reps, ms_write * 1e3 / reps);
    }
    return 0;
}
