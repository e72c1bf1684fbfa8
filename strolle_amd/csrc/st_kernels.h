// st_kernels.h — launch interface between the host engine (st_engine.cpp) and the HIP kernels.
// Each launcher enqueues exactly one kernel on `stream`; `KArgs` travels by value in the kernarg segment.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include "st_types.h"

namespace st {

// Per-kernel timing without event packets between kernels (ST_PROFILE_KERNEL_EVENTS): when the engine sets this pair, the next
// launch goes through hipExtLaunchKernelGGL, which writes the dispatch's own start / stop timestamps into the two events — what
// rocprofv3's kernel trace reads too. An event RECORDED between two kernels instead makes the second wait for a barrier packet
// (3-15 us each, measured), which is how the default profiling mode times runs of launches.
// `consumed` tells the engine that a launch really took the pair (a launcher whose grid is empty enqueues nothing: its events must go
// back to the pool unrecorded — hipEventElapsedTime on a pair no dispatch wrote fails).
struct LaunchEvents { hipEvent_t start = nullptr, stop = nullptr; bool consumed = false; };
extern thread_local LaunchEvents g_launch_events;  // st_engine.cpp
#define ST_KLAUNCH_SMEM(kernel, grid, block, smem, stream, ...)                                                                                  \
    do {                                                                                                                                         \
        if (::st::g_launch_events.start && !::st::g_launch_events.consumed) {                                                                    \
            hipExtLaunchKernelGGL(kernel, grid, block, smem, stream, ::st::g_launch_events.start, ::st::g_launch_events.stop, 0, __VA_ARGS__);   \
            ::st::g_launch_events.consumed = true;                                                                                               \
        } else hipLaunchKernelGGL(kernel, grid, block, smem, stream, __VA_ARGS__);                                                               \
    } while (0)
#define ST_KLAUNCH(kernel, grid, block, stream, ...) ST_KLAUNCH_SMEM(kernel, grid, block, 0, stream, __VA_ARGS__)

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
    KS_PRIM_VISIBILITY_REPROJECTION, KS_DI_RESOLVING_REPROJECT, KS_GI_PREVIEW_RESOLVE, KS_GI_PREVIEW_RESOLVE_REPROJECT, KS_DENOISE_WAVELET_12,
    KS_GI_REPROJECTION_TEMPORAL, KS_DI_SAMPLING_TEMPORAL, KS_DI_SPATIAL_FUSED, KS_GI_SPATIAL_FUSED,
    KS_GI_PREVIEW_BOTH, KS_GI_PREVIEW_BOTH_NO_REPROJECT, KS_GI_PREVIEW_LATE, KS_DENOISE_WAVELET_COMPOSE, KS_GI_SAMPLING_AB,
    KS_DENOISE_WAVELET_FAMILY,  // profiling only (ST_PROFILE_GROUP_ATROUS): the a-trous chain's launches timed as ONE interval
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
        {"denoise_wavelet x2 (strides 1+2)", 84.f + 84.f, false},
        {"gi_reprojection+gi_temporal", 176.f + 272.f, false},
        {"di_sampling+di_temporal", 68.f + 176.f, false},
        {"di_spatial_pick+trace+sample", 128.f + 2.f * 48.f + 192.f, true},  // per cell: the trace pass covers both of its pixels
        {"gi_spatial_pick+trace+sample", 160.f + 2.f * 48.f + 352.f, true},
        {"gi_preview x2+gi_resolving+denoise_reproject", 160.f + 160.f + 256.f + 112.f, false},
        {"gi_preview x2+gi_resolving", 160.f + 160.f + 256.f, false},
        {"gi_preview 2nd pass (pixels that resample)", 0.f, false},  // its bytes are credited to the launch above
        {"denoise_wavelet+composition", 84.f + 112.f, false},
        {"gi_sampling_a+b", 80.f + 144.f, true},
        {"a-trous chain (one timed interval)", 0.f, false},  // bytes: the sum its member launches are credited
    };
    return k[slot];
}

// The launchers exist twice, in namespaces st::exact and st::fast (the two arithmetic builds of the kernel files, Makefile);
// the engine calls them through a table picked per engine (st_engine_set_arithmetic).
#define ST_LAUNCHER(name, args) void name args;
namespace exact {
#include "st_launchers.inc"
}
namespace fast {
#include "st_launchers.inc"
}
#undef ST_LAUNCHER
struct Launchers {
#define ST_LAUNCHER(name, args) void (*name) args;
#include "st_launchers.inc"
#undef ST_LAUNCHER
};
inline Launchers launchers_exact() {
    Launchers t;
#define ST_LAUNCHER(name, args) t.name = &exact::name;
#include "st_launchers.inc"
#undef ST_LAUNCHER
    return t;
}
inline Launchers launchers_fast() {
    Launchers t;
#define ST_LAUNCHER(name, args) t.name = &fast::name;
#include "st_launchers.inc"
#undef ST_LAUNCHER
    return t;
}

}  // namespace st
