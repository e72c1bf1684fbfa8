// fetch_calib.hip — calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on MI355X for THIS project's access patterns
// (MI355X_MICROARCH.md section HBM: "calibrate on a known byte count in your own access pattern before trusting an absolute").
//
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o /tmp/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- /tmp/fetch_calib     (and again with WRITE_SIZE)
//
// Kernels (each moves a known number of bytes; the program prints them as `name bytes_read bytes_written`):
//   calib_stream16   every lane reads one float4, consecutive lanes consecutive texels (how a pass reads its own pixel's plane)
//   calib_gather16   one float4 per lane at a random texel of a 2 GiB buffer                 (a surface-twin tap far away)
//   calib_gather32   two consecutive float4 (one 32-B DI reservoir) at a random 32-B slot     (di_spatial pick)
//   calib_gather64   four consecutive float4 (one 64-B GI reservoir) at a random 64-B slot    (gi_preview / gi_spatial taps)
//   calib_near64     the same, but the random slot lies within +-128 pixels of the lane's own pixel in a 1920-wide plane
//                    (the real preview pattern: neighbouring lanes' discs overlap, so part of the traffic is L2-served)
//   calib_write16    every lane writes one float4 (coalesced)
// Results are accumulated into a per-lane sum that is written once per wave so the loads cannot be optimised away.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t pcg(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    const uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}
__device__ __forceinline__ void sink(float4 acc, float* out) {
    const float s = acc.x + acc.y + acc.z + acc.w;
    if (s == 123456.789f) out[blockIdx.x] = s;   // never true for the fill pattern; keeps the loads alive
}

__global__ void calib_stream16(const float4* buf, size_t n, float* out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    float4 acc = make_float4(0, 0, 0, 0);
    if (i < n) acc = buf[i];
    sink(acc, out);
}
template <int TEXELS>
__global__ void calib_gather(const float4* buf, size_t slots, float* out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const size_t slot = ((size_t)pcg(i) * 2654435761ull + pcg(i ^ 0x9e3779b9u)) % slots;
    float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
    for (int k = 0; k < TEXELS; k++) { const float4 v = buf[slot * TEXELS + k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    sink(acc, out);
}
// lanes form 8x8 pixel tiles of a 1920x1080 plane (like the renderer's waves); each fetches the 64-B record of a pixel
// uniformly inside a disc of radius 128 around its own
__global__ void calib_near64(const float4* buf, uint32_t width, uint32_t height, uint32_t salt, float* out) {
    const uint32_t tiles_x = width / 8u;
    const uint32_t tile = blockIdx.x * (blockDim.x / 64u) + (threadIdx.x >> 6), lane = threadIdx.x & 63u;
    const uint32_t tx = tile % tiles_x, ty = tile / tiles_x;
    const int32_t px = (int32_t)(tx * 8u + (lane & 7u)), py = (int32_t)(ty * 8u + (lane >> 3));
    float4 acc = make_float4(0, 0, 0, 0);
    if (py < (int32_t)height) {
        const uint32_t r = pcg((uint32_t)(py * (int32_t)width + px) ^ salt);
        const float ang = (float)(r & 0xffffu) * (6.2831853f / 65536.0f), rad = 128.0f * sqrtf((float)(r >> 16) / 65536.0f);
        int32_t sx = px + (int32_t)(rad * cosf(ang)), sy = py + (int32_t)(rad * sinf(ang));
        sx = sx < 0 ? -sx : (sx >= (int32_t)width ? 2 * (int32_t)width - sx - 1 : sx);
        sy = sy < 0 ? -sy : (sy >= (int32_t)height ? 2 * (int32_t)height - sy - 1 : sy);
        const size_t slot = (size_t)sy * width + (size_t)sx;
#pragma unroll
        for (int k = 0; k < 4; k++) { const float4 v = buf[slot * 4 + k]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
    }
    sink(acc, out);
}
// One 64-B record per lane (a GI reservoir: buf[4 * id + k], k = 0..3), read and written the way gi_read / gi_write do it —
// four loads / stores whose lanes are 64 B apart — against the same bytes moved as four fully coalesced 1-KB rows per wave
// (what a 4x4 lane-quad transpose would turn the accesses into).
__global__ void calib_aos64_strided(const float4* in, float4* out, size_t records) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= records) return;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = in[4 * i + k];
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k].x += 1.0f; out[4 * i + k] = v[k]; }
}
__global__ void calib_aos64_coalesced(const float4* in, float4* out, size_t records) {
    const size_t wave_base = ((size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63u)) * 4;   // first texel of this wave's 64 records
    const uint32_t lane = threadIdx.x & 63u;
    if (wave_base / 4 + 64 > records) return;
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = in[wave_base + (size_t)k * 64 + lane];
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k].x += 1.0f; out[wave_base + (size_t)k * 64 + lane] = v[k]; }
}
// the same bytes with the accesses of each lane quad transposed (instruction k moves record k of every quad: 64 contiguous
// bytes per quad) and of each group of 8 lanes (instruction k moves texels 8k..8k+7 of the group's 32: 128 contiguous bytes)
template <int GROUP>
__global__ void calib_aos64_grouped(const float4* in, float4* out, size_t records) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= records) return;
    const size_t base = (i / GROUP) * GROUP * 4;   // first texel of the group's records
    const uint32_t l = (uint32_t)(i % GROUP);
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; k++) v[k] = in[base + (size_t)k * GROUP + l];
#pragma unroll
    for (int k = 0; k < 4; k++) { v[k].x += 1.0f; out[base + (size_t)k * GROUP + l] = v[k]; }
}
__global__ void calib_write16(float4* buf, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) buf[i] = make_float4(1.0f, 2.0f, 3.0f, (float)(i & 1023u));
}
__global__ void calib_fill(float4* buf, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) buf[i] = make_float4(1.0f, 0.5f, 0.25f, 0.125f);
}

int main() {
    const size_t bytes = (size_t)2 << 30, texels = bytes / 16;   // 2 GiB: eight times the 256 MiB Infinity Cache
    float4* buf; float* out;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&out, 1 << 24));
    hipLaunchKernelGGL(calib_fill, dim3(4096), dim3(256), 0, 0, buf, texels);
    CHECK(hipDeviceSynchronize());
    const uint32_t lanes = 1920u * 1080u * 4u;   // four 1080p frames worth of lanes per launch
    const uint32_t blocks = lanes / 256u;
    for (int rep = 0; rep < 5; rep++) {
        hipLaunchKernelGGL(calib_stream16, dim3(blocks), dim3(256), 0, 0, buf + (size_t)rep * lanes, (size_t)lanes, out);
        hipLaunchKernelGGL(calib_gather<1>, dim3(blocks), dim3(256), 0, 0, buf, texels, out);
        hipLaunchKernelGGL(calib_gather<2>, dim3(blocks), dim3(256), 0, 0, buf, texels / 2, out);
        hipLaunchKernelGGL(calib_gather<4>, dim3(blocks), dim3(256), 0, 0, buf, texels / 4, out);
        // a different 1080p-sized window of the buffer per repetition, so nothing is warm in the Infinity Cache
        hipLaunchKernelGGL(calib_near64, dim3(1920 * 1080 / 256), dim3(256), 0, 0, buf + (size_t)rep * 1920 * 1080 * 4 * 2, 1920u, 1080u, (uint32_t)rep * 7919u, out);
        hipLaunchKernelGGL(calib_write16, dim3(blocks), dim3(256), 0, 0, buf + (size_t)(rep + 8) * lanes, (size_t)lanes);
        {   // 1080p worth of 64-B records, source and destination 1 GiB apart, a fresh window per repetition
            const size_t records = 1920 * 1080;   // 8.3 M texels per kernel; three windows of 16.6 M texels fit in each 67 M-texel half
            const float4* src = buf + (size_t)(rep % 3) * records * 4 * 2;
            float4* dst = buf + (texels / 2) + (size_t)(rep % 3) * records * 4 * 2;
            if ((size_t)(rep % 3 + 1) * records * 8 > texels / 2) { fprintf(stderr, "window out of range\n"); return 1; }
            hipLaunchKernelGGL(calib_aos64_strided, dim3(records / 256), dim3(256), 0, 0, src, dst, records);
            hipLaunchKernelGGL(calib_aos64_coalesced, dim3(records / 256), dim3(256), 0, 0, src + records * 4, dst + records * 4, records);
            hipLaunchKernelGGL(calib_aos64_grouped<4>, dim3(records / 256), dim3(256), 0, 0, src, dst + records * 4, records);
            hipLaunchKernelGGL(calib_aos64_grouped<8>, dim3(records / 256), dim3(256), 0, 0, src + records * 4, dst, records);
        }
        CHECK(hipDeviceSynchronize());
    }
    printf("calib_stream16 %llu 0\n", (unsigned long long)lanes * 16ull);
    printf("calib_gather<1> %llu 0\n", (unsigned long long)lanes * 16ull);
    printf("calib_gather<2> %llu 0\n", (unsigned long long)lanes * 32ull);
    printf("calib_gather<4> %llu 0\n", (unsigned long long)lanes * 64ull);
    printf("calib_near64 %llu 0\n", (unsigned long long)1920 * 1080 * 64ull);
    printf("calib_write16 0 %llu\n", (unsigned long long)lanes * 16ull);
    printf("calib_aos64_strided %llu %llu\n", (unsigned long long)1920 * 1080 * 64ull, (unsigned long long)1920 * 1080 * 64ull);
    printf("calib_aos64_coalesced %llu %llu\n", (unsigned long long)1920 * 1080 * 64ull, (unsigned long long)1920 * 1080 * 64ull);
    return 0;
}
