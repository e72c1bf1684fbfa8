// tile_shape_probe.hip — does the shape of the pixel tile a wave covers matter for a purely streaming pass?
// A wave of the ReSTIR passes is an 8x8 pixel tile (eight 128-B row segments per 16-B-per-pixel access); the LDS a-trous
// passes use 32x2. This kernel reads three float4 planes and writes two at 1920x1080 with the wave laid out as
// W x (64 / W) pixels, W = 8, 16, 32, 64, and prints GB/s of the five planes.
//
//   hipcc --offload-arch=gfx950 -O3 tools/tile_shape_probe.hip -o /tmp/tile_shape_probe && timeout 60 /tmp/tile_shape_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int W>
__global__ void k_stream(const float4* __restrict__ a, const float4* __restrict__ b, const float4* __restrict__ c, float4* __restrict__ o0, float4* __restrict__ o1, int width, int height) {
    constexpr int H = 64 / W;
    const int tiles_x = (width + W - 1) / W;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ty = wave / tiles_x, tx = wave - ty * tiles_x;
    const int x = tx * W + (lane % W), y = ty * H + (lane / W);
    if (x >= width || y >= height) return;
    const size_t i = (size_t)y * width + x;
    const float4 p = a[i], q = b[i], r = c[i];
    o0[i] = make_float4(p.x + q.x, p.y + q.y, p.z + r.z, p.w);
    o1[i] = make_float4(q.x * r.x, q.y, r.z, r.w + p.w);
}
template <int W>
static void run(float4** planes, int width, int height) {
    const int tiles = ((width + W - 1) / W) * ((height + 64 / W - 1) / (64 / W));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0;
    for (int rep = 0; rep < 3; rep++) {
        CHECK(hipEventRecord(e0));
        for (int k = 0; k < 20; k++)   // rotate through 10 planes so that nothing stays in the 256 MB Infinity Cache
            hipLaunchKernelGGL(k_stream<W>, dim3((tiles + 3) / 4), dim3(256), 0, 0, planes[(5 * k) % 30], planes[(5 * k + 1) % 30], planes[(5 * k + 2) % 30], planes[(5 * k + 3) % 30], planes[(5 * k + 4) % 30], width, height);
        CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
    }
    const double bytes = 20.0 * 5 * 16 * width * height;
    printf("wave = %2d x %d pixels: %7.1f GB/s  (%.1f us per launch)\n", W, 64 / W, bytes / (ms * 1e-3) / 1e9, ms * 1e3 / 20);
}
int main() {
    const int width = 1920, height = 1080;
    float4* planes[30];
    for (int i = 0; i < 30; i++) { CHECK(hipMalloc(&planes[i], (size_t)width * height * 16)); CHECK(hipMemset(planes[i], 0, (size_t)width * height * 16)); }
    run<8>(planes, width, height); run<16>(planes, width, height); run<32>(planes, width, height); run<64>(planes, width, height);
    return 0;
}
