// tap_probe.hip — what a jittered a-trous tap costs on MI355X, by data layout ("far a-trous passes").
//
// The strides-8 / 16 passes of the SVGF denoiser (frame_denoising.rs:262-349) read, per pixel, 8 taps at
// p + jitter(p) + o * stride with a PER-PIXEL jitter of +-1 (stride 8) or +-3 (stride 16) texels; each tap needs the
// surface (normal, depth), the direct colour and the indirect colour: 3 x 16 B. Lanes of a quad therefore land on different
// rows / 64-B segments and the texture-address unit serialises them. This probe times the same access skeleton (8 taps,
// three values per tap, a trivial reduction, 32 B of output per pixel) for the candidate layouts on a 1920x1080 frame:
//
//   planes      three float4 planes, one 16-B gather each per tap (what k_denoise_wavelet_far does)
//   planes0     the same with the jitter forced to zero (the coalesced floor)
//   pair        surface plane + one 32-B (direct | indirect) record per pixel, read per lane (two 16-B loads)
//   pair_coop   the same, but the two lanes of a pair fetch one record per instruction (32 contiguous bytes) and swap halves
//               with a DPP move
//   rec64       one 64-B (surface | direct | indirect | pad) record per pixel, read per lane (three 16-B loads); the output
//               is a full record again (the next pass taps it)
//   rec64_coop  the same, but the four lanes of a quad fetch one whole record per instruction (64 contiguous bytes = one
//               segment) and a 4x4 DPP transpose hands every lane its record
//
//   hipcc --offload-arch=gfx950 -O3 tools/tap_probe.hip -o /tmp/tap_probe && /tmp/tap_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define D __device__ __forceinline__

constexpr int W = 1920, H = 1080;

template <int CTRL>
D float4 quad_dpp(float4 v) {
    float4 r;
    r.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.x), CTRL, 0xf, 0xf, true));
    r.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.y), CTRL, 0xf, 0xf, true));
    r.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.z), CTRL, 0xf, 0xf, true));
    r.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.w), CTRL, 0xf, 0xf, true));
    return r;
}
template <int CTRL>
D uint32_t quad_dpp_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
D float4 sel4(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
D void quad_transpose(float4& v0, float4& v1, float4& v2, float4& v3) {
    const bool b0 = (threadIdx.x & 1u) != 0u, b1 = (threadIdx.x & 2u) != 0u;
    { const float4 ra = quad_dpp<0xB1>(sel4(b0, v0, v1)), rb = quad_dpp<0xB1>(sel4(b0, v2, v3)); if (b0) { v0 = ra; v2 = rb; } else { v1 = ra; v3 = rb; } }
    { const float4 ra = quad_dpp<0x4E>(sel4(b1, v0, v2)), rb = quad_dpp<0x4E>(sel4(b1, v1, v3)); if (b1) { v0 = ra; v1 = rb; } else { v2 = ra; v3 = rb; } }
}
D float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
D float4 fma4(float4 a, float s, float4 c) { return make_float4(a.x * s + c.x, a.y * s + c.y, a.z * s + c.z, a.w * s + c.w); }

struct Px { int x, y; bool ok; };
D Px pixel() {
    const int tiles_x = W / 8, groups_x = tiles_x / 4;
    const int gy = blockIdx.x / groups_x, gx = blockIdx.x - gy * groups_x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    Px p; p.x = (gx * 4 + wave) * 8 + (lane & 7); p.y = gy * 8 + (lane >> 3); p.ok = p.y < H;
    return p;
}
// the reference's jitter: trunc((noise - 0.5) * (stride - 1) * 0.5) per axis, noise from a 256x256 byte table
template <int S, bool JITTER>
D void tap_ids(const uchar2* noise, Px p, uint32_t center, uint32_t* at) {
    int jx = 0, jy = 0;
    if (JITTER) {
        const uchar2 n = noise[(p.y & 255) * 256 + (p.x & 255)];
        jx = (int)(((float)n.x / 255.0f - 0.5f) * (float)(S - 1) * 0.5f);
        jy = (int)(((float)n.y / 255.0f - 0.5f) * (float)(S - 1) * 0.5f);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int k = t < 4 ? t : t + 1, ox = k % 3 - 1, oy = k / 3 - 1;
        const int sx = p.x + jx + ox * S, sy = p.y + jy + oy * S;
        at[t] = (sx >= 0 && sy >= 0 && sx < W && sy < H) ? (uint32_t)(sy * W + sx) : center;
    }
}

template <int S, bool JITTER>
__global__ __launch_bounds__(256) void k_planes(const uchar2* noise, const float4* sn, const float4* di, const float4* gi, float4* odi, float4* ogi) {
    const Px p = pixel(); if (!p.ok) return;
    const uint32_t c = (uint32_t)(p.y * W + p.x);
    uint32_t at[8]; tap_ids<S, JITTER>(noise, p, c, at);
    float w[8];
    { float4 s[8];
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] = sn[at[t]];
#pragma unroll
      for (int t = 0; t < 8; t++) w[t] = s[t].x + s[t].w; }
    __builtin_amdgcn_sched_barrier(0);
    { float4 acc = di[c], v[8];
#pragma unroll
      for (int t = 0; t < 8; t++) v[t] = di[at[t]];
#pragma unroll
      for (int t = 0; t < 8; t++) acc = fma4(v[t], w[t], acc);
      odi[c] = acc; }
    __builtin_amdgcn_sched_barrier(0);
    { float4 acc = gi[c], v[8];
#pragma unroll
      for (int t = 0; t < 8; t++) v[t] = gi[at[t]];
#pragma unroll
      for (int t = 0; t < 8; t++) acc = fma4(v[t], w[t], acc);
      ogi[c] = acc; }
}

template <int S, bool COOP>
__global__ __launch_bounds__(256) void k_pair(const uchar2* noise, const float4* sn, const float4* pair, float4* opair) {
    const Px p = pixel(); if (!p.ok) return;
    const uint32_t c = (uint32_t)(p.y * W + p.x);
    uint32_t at[8]; tap_ids<S, true>(noise, p, c, at);
    float w[8];
    { float4 s[8];
#pragma unroll
      for (int t = 0; t < 8; t++) s[t] = sn[at[t]];
#pragma unroll
      for (int t = 0; t < 8; t++) w[t] = s[t].x + s[t].w; }
    __builtin_amdgcn_sched_barrier(0);
    const bool odd = (threadIdx.x & 1u) != 0u;
    const uint32_t half = threadIdx.x & 1u;
    float4 a0, a1;
    if (COOP) {   // own record: lanes 2k, 2k+1 are consecutive pixels = 64 contiguous bytes for the pair, two instructions
        const uint32_t first = c - half;
        const float4 l0 = pair[2u * first + half], l1 = pair[2u * (first + 1u) + half];
        const float4 recv = quad_dpp<0xB1>(sel4(odd, l0, l1));
        a0 = odd ? recv : l0; a1 = odd ? l1 : recv;
    } else { a0 = pair[2u * c]; a1 = pair[2u * c + 1u]; }
#pragma unroll
    for (int h = 0; h < 2; h++) {   // two rounds of four taps: eight texels live at a time
        float4 d[4], g[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const uint32_t mine = at[h * 4 + q];
            if (COOP) {
                const uint32_t other = quad_dpp_u<0xB1>(mine);
                const float4 l0 = pair[2u * (odd ? other : mine) + half];   // the even lane's record
                const float4 l1 = pair[2u * (odd ? mine : other) + half];   // the odd lane's record
                const float4 recv = quad_dpp<0xB1>(sel4(odd, l0, l1));      // even sends l1 (the odd lane's first half), odd sends l0
                d[q] = odd ? recv : l0; g[q] = odd ? l1 : recv;
            } else { d[q] = pair[2u * mine]; g[q] = pair[2u * mine + 1u]; }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) { a0 = fma4(d[q], w[h * 4 + q], a0); a1 = fma4(g[q], w[h * 4 + q], a1); }
    }
    if (COOP) {
        const uint32_t first = c - half;
        const float4 recv = quad_dpp<0xB1>(sel4(odd, a0, a1));   // even sends a1, odd sends a0
        opair[2u * first + half] = odd ? recv : a0;               // the even lane's record
        opair[2u * (first + 1u) + half] = odd ? a1 : recv;        // the odd lane's record
    } else { opair[2u * c] = a0; opair[2u * c + 1u] = a1; }
}

template <int S, bool COOP>
__global__ __launch_bounds__(256) void k_rec64(const uchar2* noise, const float4* rec, float4* orec) {
    const Px p = pixel(); if (!p.ok) return;
    const uint32_t c = (uint32_t)(p.y * W + p.x);
    uint32_t at[8]; tap_ids<S, true>(noise, p, c, at);
    const uint32_t j = threadIdx.x & 3u;
    float4 csn, a0, a1;
    if (COOP) {
        const uint32_t first = c - j;
        float4 v0 = rec[4u * first + j], v1 = rec[4u * (first + 1u) + j], v2 = rec[4u * (first + 2u) + j], v3 = rec[4u * (first + 3u) + j];
        quad_transpose(v0, v1, v2, v3);
        csn = v0; a0 = v1; a1 = v2;
    } else { csn = rec[4u * c]; a0 = rec[4u * c + 1u]; a1 = rec[4u * c + 2u]; }
#pragma unroll
    for (int h = 0; h < 4; h++) {   // four rounds of two taps
        float4 s[2], d[2], g[2];
#pragma unroll
        for (int q = 0; q < 2; q++) {
            const uint32_t mine = at[h * 2 + q];
            if (COOP) {
                const uint32_t r0 = quad_dpp_u<0x00>(mine), r1 = quad_dpp_u<0x55>(mine), r2 = quad_dpp_u<0xAA>(mine), r3 = quad_dpp_u<0xFF>(mine);
                float4 v0 = rec[4u * r0 + j], v1 = rec[4u * r1 + j], v2 = rec[4u * r2 + j], v3 = rec[4u * r3 + j];
                quad_transpose(v0, v1, v2, v3);
                s[q] = v0; d[q] = v1; g[q] = v2;
            } else { s[q] = rec[4u * mine]; d[q] = rec[4u * mine + 1u]; g[q] = rec[4u * mine + 2u]; }
        }
#pragma unroll
        for (int q = 0; q < 2; q++) { const float w = s[q].x + s[q].w; a0 = fma4(d[q], w, a0); a1 = fma4(g[q], w, a1); }
    }
    if (COOP) {
        const uint32_t first = c - j;
        float4 v0 = csn, v1 = a0, v2 = a1, v3 = make_float4(0.f, 0.f, 0.f, 0.f);
        quad_transpose(v0, v1, v2, v3);
        orec[4u * first + j] = v0; orec[4u * (first + 1u) + j] = v1; orec[4u * (first + 2u) + j] = v2; orec[4u * (first + 3u) + j] = v3;
    } else { orec[4u * c] = csn; orec[4u * c + 1u] = a0; orec[4u * c + 2u] = a1; }
}

template <class F>
static float time_us(F&& launch, int reps = 30) {
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) launch();
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(a, nullptr));
    for (int i = 0; i < reps; i++) launch();
    CHECK(hipEventRecord(b, nullptr)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipGetLastError());
    return ms * 1e3f / reps;
}

int main() {
    const size_t n = (size_t)W * H;
    std::vector<float> host(n * 16);
    srand(7);
    for (auto& v : host) v = (float)(rand() & 1023) / 1024.0f;
    std::vector<unsigned char> noise(256 * 256 * 2);
    for (auto& v : noise) v = (unsigned char)(rand() & 255);
    uchar2* d_noise; CHECK(hipMalloc(&d_noise, noise.size())); CHECK(hipMemcpy(d_noise, noise.data(), noise.size(), hipMemcpyHostToDevice));
    float4 *sn, *di, *gi, *odi, *ogi, *pair, *opair, *rec, *orec;
    CHECK(hipMalloc(&sn, n * 16)); CHECK(hipMalloc(&di, n * 16)); CHECK(hipMalloc(&gi, n * 16)); CHECK(hipMalloc(&odi, n * 16)); CHECK(hipMalloc(&ogi, n * 16));
    CHECK(hipMalloc(&pair, n * 32)); CHECK(hipMalloc(&opair, n * 32)); CHECK(hipMalloc(&rec, n * 64)); CHECK(hipMalloc(&orec, n * 64));
    CHECK(hipMemcpy(sn, host.data(), n * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(di, host.data() + n * 4, n * 16, hipMemcpyHostToDevice)); CHECK(hipMemcpy(gi, host.data() + n * 8, n * 16, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(pair, host.data(), n * 32, hipMemcpyHostToDevice)); CHECK(hipMemcpy(rec, host.data(), n * 64, hipMemcpyHostToDevice));
    const dim3 grid((W / 32) * (H + 7) / 8 > 0 ? (W / 32) * ((H + 7) / 8) : 1), block(256);
    const double alg = 84.0 * (double)n;   // 84 B per pixel and pass
    auto report = [&](const char* name, int stride, float us, double hbm_bytes) {
        printf("%-12s stride %2d: %7.2f us   algorithmic 84 B/px -> %6.2f TB/s   layout's own compulsory bytes %6.1f MB -> %6.2f TB/s\n", name, stride, us, alg / us * 1e-6, hbm_bytes * 1e-6,
               hbm_bytes / us * 1e-6);
    };
    const double b_planes = 80.0 * n, b_pair = 80.0 * n, b_rec = 128.0 * n;
#define RUN(S)                                                                                                                                   \
    report("planes0", S, time_us([&] { hipLaunchKernelGGL((k_planes<S, false>), grid, block, 0, nullptr, d_noise, sn, di, gi, odi, ogi); }), b_planes);  \
    report("planes", S, time_us([&] { hipLaunchKernelGGL((k_planes<S, true>), grid, block, 0, nullptr, d_noise, sn, di, gi, odi, ogi); }), b_planes);    \
    report("pair", S, time_us([&] { hipLaunchKernelGGL((k_pair<S, false>), grid, block, 0, nullptr, d_noise, sn, pair, opair); }), b_pair);              \
    report("pair_coop", S, time_us([&] { hipLaunchKernelGGL((k_pair<S, true>), grid, block, 0, nullptr, d_noise, sn, pair, opair); }), b_pair);          \
    report("rec64", S, time_us([&] { hipLaunchKernelGGL((k_rec64<S, false>), grid, block, 0, nullptr, d_noise, rec, orec); }), b_rec);                   \
    report("rec64_coop", S, time_us([&] { hipLaunchKernelGGL((k_rec64<S, true>), grid, block, 0, nullptr, d_noise, rec, orec); }), b_rec);
    RUN(8)
    RUN(16)
    return 0;
}
