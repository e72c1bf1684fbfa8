// st_kernels.h — launch interface between the host engine (st_engine.cpp) and the HIP kernels.
// Each launcher enqueues exactly one kernel on `stream`; `KArgs` travels by value in the kernarg segment.
#pragma once
#include <hip/hip_runtime.h>

#include "st_types.h"

namespace st {

// Kernel slots: index into the per-camera counter array (2 x u64 per slot: rays, traversal bytes) and into the
// profiler's table. `bytes_per_unit` = compulsory screen-space bytes one launch unit (pixel or 2x1 cell) reads +
// writes through the planes the reference binds for that pass (neighbour taps assumed cache-served).
enum KernelSlot {
    KS_BVH_HEATMAP, KS_REF_TRACING, KS_REF_SHADING, KS_PRIM_VISIBILITY, KS_FRAME_REPROJECTION,
    KS_DI_SAMPLING, KS_DI_TEMPORAL, KS_DI_SPATIAL_PICK, KS_DI_SPATIAL_TRACE, KS_DI_SPATIAL_SAMPLE, KS_DI_RESOLVING,
    KS_GI_REPROJECTION, KS_GI_SAMPLING_A, KS_GI_SAMPLING_B, KS_GI_TEMPORAL, KS_GI_SPATIAL_PICK, KS_GI_SPATIAL_TRACE,
    KS_GI_SPATIAL_SAMPLE, KS_GI_PREVIEW, KS_GI_RESOLVING, KS_DENOISE_REPROJECT, KS_DENOISE_VARIANCE, KS_DENOISE_WAVELET,
    KS_COMPOSITION,
    // fused launches (own-pixel consumer passes appended to their producer); bytes = sum of the reference passes they execute
    KS_PRIM_VISIBILITY_REPROJECTION, KS_DI_RESOLVING_REPROJECT, KS_GI_PREVIEW_RESOLVE, KS_GI_PREVIEW_RESOLVE_REPROJECT, KS_DENOISE_WAVELET_COMPOSE,
    KS_GI_REPROJECTION_TEMPORAL, KS_DI_SAMPLING_TEMPORAL, KS_DI_SPATIAL_FUSED, KS_GI_SPATIAL_FUSED,
    KS_COUNT
};
struct KernelInfo { const char* name; float bytes_per_unit; bool half; };
inline const KernelInfo& kernel_info(int slot) {
    static const KernelInfo k[KS_COUNT] = {
        {"bvh_heatmap", 16.f, false},        {"ref_tracing", 64.f, false},         {"ref_shading", 128.f, false},
        {"prim_visibility", 64.f, false},    {"frame_reprojection", 64.f, false},  {"di_sampling", 68.f, false},
        {"di_temporal", 176.f, false},       {"di_spatial_pick", 128.f, true},     {"di_spatial_trace", 48.f, false},
        {"di_spatial_sample", 192.f, true},  {"di_resolving", 128.f, false},       {"gi_reprojection", 176.f, false},
        {"gi_sampling_a", 80.f, true},       {"gi_sampling_b", 144.f, true},       {"gi_temporal", 272.f, false},
        {"gi_spatial_pick", 160.f, true},    {"gi_spatial_trace", 48.f, false},    {"gi_spatial_sample", 352.f, true},
        {"gi_preview", 160.f, false},        {"gi_resolving", 256.f, false},       {"denoise_reproject", 112.f, false},
        {"denoise_variance", 112.f, false},  {"denoise_wavelet", 84.f, false},     {"composition", 112.f, false},
        {"prim_visibility+frame_reprojection", 64.f + 64.f, false},
        {"di_resolving+denoise_reproject", 128.f + 112.f, false},
        {"gi_preview+gi_resolving", 160.f + 256.f, false},
        {"gi_preview+gi_resolving+denoise_reproject", 160.f + 256.f + 112.f, false},
        {"denoise_wavelet+composition", 84.f + 112.f, false},
        {"gi_reprojection+gi_temporal", 176.f + 272.f, false},
        {"di_sampling+di_temporal", 68.f + 176.f, false},
        {"di_spatial_pick+trace+sample", 128.f + 2.f * 48.f + 192.f, true},  // per cell: the trace pass covers both of its pixels
        {"gi_spatial_pick+trace+sample", 160.f + 2.f * 48.f + 352.f, true},
    };
    return k[slot];
}

// atmosphere LUTs (passes/atmosphere.rs: transmittance + scattering once, sky on sun-altitude change)
void launch_atmosphere_static(float4* transmittance_lut, float4* scattering_lut, hipStream_t s);
void launch_atmosphere_sky(const float4* transmittance_lut, const float4* scattering_lut, float sun_altitude, float4* sky_lut, hipStream_t s);
// ray-tracing passes
void launch_bvh_heatmap(const KArgs& a, hipStream_t s);
void launch_ref_tracing(const KArgs& a, uint32_t depth, hipStream_t s);
void launch_ref_shading(const KArgs& a, uint32_t seed, uint32_t depth, hipStream_t s);
void launch_prim_visibility(const KArgs& a, bool fuse_frame_reprojection, hipStream_t s);
void launch_build_byte_luts(float* out /* kByteLutFloats */, hipStream_t s);  // st_device.h byte decode tables
void launch_frame_reprojection(const KArgs& a, hipStream_t s);
// ReSTIR DI
void launch_di_sampling(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_di_temporal(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_di_sampling_temporal(const KArgs& a, uint32_t seed_sampling, uint32_t seed_temporal, hipStream_t s);  // both passes, one launch
void launch_di_spatial_pick(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_spatial_trace(const KArgs& a, const float4* buf_d0, const float4* buf_d1, float4* buf_d2, hipStream_t s);
void launch_di_spatial_sample(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_di_spatial_fused(const KArgs& a, uint32_t seed_pick, uint32_t seed_sample, hipStream_t s);  // pick + trace + sample per cell
void launch_di_resolving(const KArgs& a, bool fuse_denoise_reproject, hipStream_t s);
// ReSTIR GI
void launch_gi_reprojection(const KArgs& a, hipStream_t s);
void launch_gi_sampling_a(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_gi_sampling_b(const KArgs& a, uint32_t seed, hipStream_t s);
// fuse_reprojection: gi_reprojection.rs runs inside this launch (legal on tracing frames, where nothing else reads gi_res[2] before)
void launch_gi_temporal(const KArgs& a, uint32_t seed, bool fuse_reprojection, hipStream_t s);
void launch_gi_spatial_pick(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_gi_spatial_sample(const KArgs& a, uint32_t seed, hipStream_t s);
void launch_gi_spatial_fused(const KArgs& a, uint32_t seed_pick, uint32_t seed_sample, hipStream_t s);  // pick + trace + sample per cell
void launch_gi_preview(const KArgs& a, uint32_t seed, uint32_t nth, const float4* in, float4* out, hipStream_t s);
void launch_gi_resolving(const KArgs& a, uint32_t source, hipStream_t s);
// second preview pass + gi_resolving (+ the GI half of denoise reproject) in one launch
void launch_gi_preview_resolve(const KArgs& a, uint32_t seed, uint32_t nth, const float4* in, uint32_t source, bool fuse_denoise_reproject, hipStream_t s);
// SVGF + composition
void launch_denoise_reproject(const KArgs& a, const float4* prev_colors, const float4* prev_moments, const float4* samples, float4* colors,
                              float4* moments, hipStream_t s);
void launch_denoise_variance(const KArgs& a, hipStream_t s);
// sl_in / sl_out: sqrt-luma planes of the input / output colour planes (KArgs::sl). The LDS-staged strides (1, 2, 4)
// require sl_in; strides 8 and 16 take null and derive the values from the colours. sl_out null = not needed downstream.
void launch_denoise_wavelet(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                            float4* gi_out, const float2* sl_in, float2* sl_out, hipStream_t s);
// last wavelet pass + frame composition in one launch
void launch_denoise_wavelet_compose(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                                    float4* gi_out, const float2* sl_in, uint32_t camera_mode, float4* frame_out, hipStream_t s);
void launch_composition(const KArgs& a, uint32_t camera_mode, const float4* di_diff, const float4* gi_diff, void* out, uint32_t format, hipStream_t s);

}  // namespace st
