// k_common.h — per-kernel boilerplate: tile/pixel resolution and launch geometry.
#pragma once
#include "st_device.h"
#include "st_passes.h"
#include "st_kernels.h"

namespace st {

constexpr int kBlockThreads = 256;               // 4 wavefronts, one 8x8 tile each
// -DST_MIN_WAVES=N (experiments): ask the register allocator for at least N waves per SIMD on every pass kernel
#ifdef ST_MIN_WAVES
#define ST_KERNEL_BOUNDS __launch_bounds__(kBlockThreads, ST_MIN_WAVES)
#else
#define ST_KERNEL_BOUNDS __launch_bounds__(kBlockThreads)
#endif
// the traversal stacks of a block: dynamic LDS, [wave][entry][lane], KArgs::stack_entries x 256 slots of SE (st_device.h lane_stack)
#define ST_STACK_LDS(SE, name) extern __shared__ __align__(16) unsigned char st_stack_lds_[]; SE* name = reinterpret_cast<SE*>(st_stack_lds_)
inline uint32_t stack_lds_bytes(const KArgs& a, size_t slot_bytes) { return a.stack_entries * (uint32_t)kBlockThreads * (uint32_t)slot_bytes; }

// The window of 8x8 tiles a launch covers: all of the viewport, or the rows / columns of st_camera_set_rows / st_camera_set_window
// widened to tile boundaries (owns_pixel masks the rest). Half-resolution passes (2x1 checkerboard cells, `(size + 7) / 8 / (2, 1)`
// workgroups: passes/*_resampling.rs, gi_sampling.rs) count tiles of 8 CELLS = 16 pixels in x.
struct TileWindow { uint32_t tx0, tx1, ty0, ty1; };
__host__ __device__ inline TileWindow tile_window(const KArgs& a, bool half_x) {
    uint32_t tiles_x = (a.width + 7u) >> 3;
    if (half_x) tiles_x >>= 1;
    const uint32_t shift = half_x ? 4u : 3u, round = half_x ? 15u : 7u;
    TileWindow w;
    w.tx0 = a.col0 >> shift;
    w.tx1 = (a.col1 + round) >> shift; if (w.tx1 > tiles_x) w.tx1 = tiles_x;
    if (w.tx0 > w.tx1) w.tx0 = w.tx1;
    w.ty0 = a.row0 >> 3;
    w.ty1 = (a.row1 + 7u) >> 3;
    return w;
}
struct LaunchDims { uint32_t tiles_x, tile_x0, tile_y0, tile_y1, blocks; };
inline LaunchDims launch_dims(const KArgs& a, bool half_x) {
    const TileWindow w = tile_window(a, half_x);
    LaunchDims d;
    d.tiles_x = w.tx1 - w.tx0; d.tile_x0 = w.tx0;
    d.tile_y0 = w.ty0; d.tile_y1 = w.ty1;
    const uint32_t groups_x = (d.tiles_x + 3u) / 4u;
    d.blocks = groups_x * (d.tile_y1 - d.tile_y0);
    return d;
}

// This wave's tile (absolute tile coordinates). valid == false: outside the dispatch.
ST_D TileCoord resolve_tile(const KArgs& a, bool half_x) {
    const TileWindow w = tile_window(a, half_x);
    TileCoord tc = tile_for_thread(w.tx1 - w.tx0, w.ty1 - w.ty0, a.tile_map);
    tc.x += w.tx0; tc.y += w.ty0;
    return tc;
}
// ... for workgroup number `b` of the one-workgroup-per-four-tiles launch (a kernel whose workgroups take two of those: k_gi_sampling_ab_pool)
ST_D bool resolve_gid_of_block(const KArgs& a, bool half_x, uint32_t b, U2* gid) {
    const TileWindow w = tile_window(a, half_x);
    TileCoord tc = tile_for_block(b, w.tx1 - w.tx0, w.ty1 - w.ty0, a.tile_map);
    tc.x += w.tx0; tc.y += w.ty0;
    if (!tc.valid) return false;
    *gid = pixel_in_tile(tc);
    return true;
}
// Resolves this thread's `global_invocation_id` (gid). Returns false for lanes outside the dispatch.
ST_D bool resolve_gid(const KArgs& a, bool half_x, U2* gid) {
    const TileCoord tc = resolve_tile(a, half_x);
    if (!tc.valid) return false;
    *gid = pixel_in_tile(tc);
    return true;
}
// pixel belongs to this launch: inside the viewport (Camera::contains) and inside the window
ST_D bool owns_pixel(const KArgs& a, U2 p) { return p.x < a.width && p.y < a.height && p.y >= a.row0 && p.y < a.row1 && p.x >= a.col0 && p.x < a.col1; }

// Tracing kernels are instantiated three ways; the scene picks one at launch:
//   <true,  uint16_t>  the whole device BVH stream fits in LDS (Cornell: 55 entries = 220 float4): every block copies it in
//                      once and traversal reads ds_read_b128 instead of going through the vector L1 (dependent entry
//                      fetches are the traversal's latency chain; measured -21 % on the shadow-ray pass)
//   <false, uint16_t>  fewer than 65,536 entries (stack slots hold entry numbers = texel pointer / 4): 16-bit stack entries
//   <false, uint32_t>  anything larger
constexpr uint32_t kStack16Texels = 4u * 65536u;
// (the compact stream's pointers carry a kind bit: entry << 1 | leaf — half as many entries fit a 16-bit slot)
// (the wide stream's 16-bit form — links inside the sort keys — needs 16-bit slots, its 32-bit form 32-bit ones: KArgs::bvh_w_links16 decides)
inline uint32_t stack16_limit(const KArgs& a) {
    if (a.bvh_w) return a.bvh_w_links16 ? 0xffffffffu : 0u;
    return a.bvh_c ? kStack16Texels / 2u : kStack16Texels;   // (32-bit slots cost 0.6 % of the dungeon frame, measured)
}
#ifdef ST_NO_LDS_SCENE  // experiment switch (tools/ab_bench.sh): small scenes traverse through the vector L1 like large ones
inline bool scene_fits_lds(const KArgs&) { return false; }
#else
inline bool scene_fits_lds(const KArgs& a) { return a.bvh_len > 0u && a.bvh_len <= kLdsSceneTexels; }
#endif
// first statement of a tracing kernel whose parameter is `a_in`: defines `a`, the arguments the body uses
// ... and every tracing kernel stages the head of the light table in LDS (kLdsLights x 112 B = 1.75 KB per block): RIS over the lights
// (reservoir/ephemeral.rs:14-55: up to 16 picks per pixel, each a 112-B fetch at a per-lane random index) and the resampling passes'
// per-sample light look-ups then read LDS instead of going through the texture-address path. Lights beyond kLdsLights keep the global table.
#define ST_SCENE_PROLOGUE                                                                                                        \
    __shared__ float4 s_scene_bvh_[LDS_SCENE ? kLdsSceneTexels : 1];                                                             \
    __shared__ GpuLight s_lights_[kLdsLights];                                                                                   \
    KArgs a = a_in;                                                                                                              \
    {                                                                                                                            \
        const uint32_t n_l_ = (a_in.n_lights_buf < kLdsLights ? a_in.n_lights_buf : kLdsLights) * 7u;                            \
        for (uint32_t i_ = threadIdx.x; i_ < n_l_; i_ += kBlockThreads)                                                          \
            reinterpret_cast<float4*>(s_lights_)[i_] = reinterpret_cast<const float4*>(a_in.lights)[i_];                         \
        if (LDS_SCENE) for (uint32_t i_ = threadIdx.x; i_ < a_in.bvh_len; i_ += kBlockThreads) s_scene_bvh_[i_] = a_in.bvh[i_];  \
        __syncthreads();                                                                                                         \
        a.lights_lds = s_lights_;                                                                                                \
        if (LDS_SCENE) a.bvh = s_scene_bvh_;                                                                                     \
    }
// For a kernel that decodes MANY G-buffer texels per lane (GI spatial resampling: every candidate neighbour's): the byte tables of the
// decode (st_device.h kLut*, 4 KB) staged in LDS as well — seven per-lane table reads per decoded texel leave the texture-address path
// (dungeon gi_spatial_fused 366 -> 335 us; in the kernels that decode one pixel the staging costs what it saves, measured).
#define ST_SCENE_PROLOGUE_WITH_BYTE_TABLES                                                                                       \
    __shared__ float s_byte_luts_[kByteLutFloats];                                                                               \
    for (uint32_t i_ = threadIdx.x; i_ < kByteLutFloats / 4u; i_ += kBlockThreads)                                               \
        reinterpret_cast<float4*>(s_byte_luts_)[i_] = reinterpret_cast<const float4*>(a_in.byte_luts)[i_];                       \
    ST_SCENE_PROLOGUE                                                                                                            \
    a.byte_luts = s_byte_luts_;
#define ST_LAUNCH_TRACE(kernel_tmpl, half, stream, ...)                                                             \
    do {                                                                                                            \
        if (scene_fits_lds(a)) ST_LAUNCH_SMEM(ST_TPL2(kernel_tmpl, true, uint16_t), half, stack_lds_bytes(a, 2), stream, __VA_ARGS__);          \
        else if (a.bvh_len < stack16_limit(a)) ST_LAUNCH_SMEM(ST_TPL2(kernel_tmpl, false, uint16_t), half, stack_lds_bytes(a, 2), stream, __VA_ARGS__);   \
        else ST_LAUNCH_SMEM(ST_TPL2(kernel_tmpl, false, uint32_t), half, stack_lds_bytes(a, 4), stream, __VA_ARGS__);                           \
    } while (0)
// the same for kernels with one more leading bool (REPROJECT)
#define ST_LAUNCH_TRACE_B(kernel_tmpl, flag, half, stream, ...)                                                         \
    do {                                                                                                                \
        if (scene_fits_lds(a)) ST_LAUNCH_SMEM(ST_TPL3(kernel_tmpl, true, flag, uint16_t), half, stack_lds_bytes(a, 2), stream, __VA_ARGS__);        \
        else if (a.bvh_len < stack16_limit(a)) ST_LAUNCH_SMEM(ST_TPL3(kernel_tmpl, false, flag, uint16_t), half, stack_lds_bytes(a, 2), stream, __VA_ARGS__); \
        else ST_LAUNCH_SMEM(ST_TPL3(kernel_tmpl, false, flag, uint32_t), half, stack_lds_bytes(a, 4), stream, __VA_ARGS__);                         \
    } while (0)
#define ST_TPL(k, t) k<t>
#define ST_TPL2(k, b, t) (k<b, t>)
#define ST_TPL3(k, b, c, t) (k<b, c, t>)

#define ST_LAUNCH_SMEM(kernel, half, smem, stream, ...)                                                          \
    do {                                                                                                         \
        const LaunchDims d_ = launch_dims(a, half);                                                              \
        if (d_.blocks) ST_KLAUNCH_SMEM(kernel, dim3(d_.blocks), dim3(kBlockThreads), smem, stream, __VA_ARGS__); \
    } while (0)
#define ST_LAUNCH(kernel, half, stream, ...)                                                          \
    do {                                                                                              \
        const LaunchDims d_ = launch_dims(a, half);                                                   \
        if (d_.blocks) ST_KLAUNCH(kernel, dim3(d_.blocks), dim3(kBlockThreads), stream, __VA_ARGS__); \
    } while (0)

}  // namespace st
