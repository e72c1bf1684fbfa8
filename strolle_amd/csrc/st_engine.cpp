// st_engine.cpp — host engine of libstrolle_hip.so: construction, environment switches, destruction. See st_engine.h.
#include "st_engine.h"

namespace st {

thread_local std::string g_last_error;
thread_local LaunchEvents g_launch_events;  // st_kernels.h: set around one launch while ST_PROFILE_KERNEL_EVENTS is on

// Defaults of StTuning (what bench.py times) and the environment variables that override them when an engine is created.
StTuning default_tuning() {
    StTuning t{};
    t.struct_size = sizeof(StTuning);
    t.overlap = t.fuse = t.fuse_di_head = t.fuse_spatial = t.fuse_gi_sampling = t.fuse_gi_validation = t.fuse_gi_reprojection = 1u;
    t.fuse_wavelet = t.fuse_compose = t.preview_both = t.variance_in_reproject = t.lean_frame = 1u;
    t.skip_scratch_stores = t.di_head_on_main = t.alias_gi_history = 1u;
    t.tile_map = 1u; t.tile_map_denoise = 2u; t.side_priority = 0;
    t.staging = t.double_buffer = t.packed_base = 1u; t.tick_timing = 0u;
    t.anyhit_fast = 1u; t.compact_bvh = 1u; t.wide_bvh = 1u; t.primary_packets = 1u;
    t.allow_deep_bvh = 0u; t.device_bake = 1u;
    return t;
}
static void tuning_from_environment(StTuning& t) {
    struct Clear { const char* name; uint32_t StTuning::*field; };   // NAME=1 clears the field (NAME=0 sets it)
    static const Clear clears[] = {
        {"ST_NO_OVERLAP", &StTuning::overlap}, {"ST_NO_FUSE", &StTuning::fuse}, {"ST_NO_FUSE_DI_HEAD", &StTuning::fuse_di_head},
        {"ST_NO_FUSE_SPATIAL", &StTuning::fuse_spatial}, {"ST_NO_FUSE_GI_SAMPLING", &StTuning::fuse_gi_sampling},
        {"ST_NO_FUSE_GI_VALIDATION", &StTuning::fuse_gi_validation}, {"ST_NO_FUSE_GI_REPROJECTION", &StTuning::fuse_gi_reprojection},
        {"ST_NO_FUSE_WAVELET", &StTuning::fuse_wavelet}, {"ST_NO_FUSE_COMPOSE", &StTuning::fuse_compose}, {"ST_NO_PREVIEW_BOTH", &StTuning::preview_both},
        {"ST_NO_VARIANCE_IN_REPROJECT", &StTuning::variance_in_reproject},
        {"ST_KEEP_ALL_PLANES", &StTuning::lean_frame}, {"ST_KEEP_SCRATCH", &StTuning::skip_scratch_stores}, {"ST_NO_GI_ALIAS", &StTuning::alias_gi_history},
        {"ST_NO_STAGING", &StTuning::staging}, {"ST_NO_DOUBLE_BUFFER", &StTuning::double_buffer}, {"ST_NO_PACKED_BASE", &StTuning::packed_base},
        {"ST_NO_ANYHIT_FAST", &StTuning::anyhit_fast}, {"ST_NO_COMPACT_BVH", &StTuning::compact_bvh}, {"ST_NO_WIDE_BVH", &StTuning::wide_bvh}, {"ST_NO_PRIMARY_PACKETS", &StTuning::primary_packets},
    };
    for (const Clear& c : clears) if (const char* v = getenv(c.name)) { if (atoi(v) != 0) t.*c.field = 0u; else if (t.*c.field == 0u) t.*c.field = 1u; }
    struct Value { const char* name; uint32_t StTuning::*field; };
    static const Value values[] = {
        {"ST_DI_HEAD_ON_MAIN", &StTuning::di_head_on_main}, {"ST_TILE_MAP_DENOISE", &StTuning::tile_map_denoise}, {"ST_TICK_TIMING", &StTuning::tick_timing},
        {"ST_ALLOW_DEEP_BVH", &StTuning::allow_deep_bvh}, {"ST_DEVICE_BAKE", &StTuning::device_bake},
    };
    if (const char* v = getenv("ST_TILE_MAP")) t.tile_map = t.tile_map_denoise = (uint32_t)atoi(v);
    for (const Value& c : values) if (const char* v = getenv(c.name)) t.*c.field = (uint32_t)atoi(v);
    if (const char* v = getenv("ST_SIDE_PRIORITY")) t.side_priority = atoi(v);
}

Engine::Engine() {
    GpuLight sun{};
    sun.d0 = make_float4(0, 0, 0, 25.0f); sun.d1 = make_float4(0, 0, 0, INFINITY); sun.d2 = make_float4(b2f(1u), 0, 0, 0);
    light_buffer.push_back(sun);
    light_slot[-1] = 0;
    blue_noise.assign(256 * 256 * 4, 0);
    reset_profile_totals();
    tuning = default_tuning();
    tuning_from_environment(tuning);
    staging.enabled = tuning.staging != 0u;
    if (const char* x = getenv("ST_EXP")) exp_flags = (uint32_t)strtoul(x, nullptr, 0);
    if (const char* ex = getenv("ST_EXACT")) if (atoi(ex) != 0) { arithmetic = ST_ARITH_EXACT; L = launchers_exact(); }
}
// st_engine_set_tuning: between frames. What cannot change under a running pipeline is re-armed here.
int Engine::set_tuning(const StTuning& t) {
    if (t.struct_size != sizeof(StTuning)) return fail(ST_ERR_INVALID_ARGUMENT, "StTuning::struct_size does not match this library");
    if (t.tile_map > 2u || t.tile_map_denoise > 2u) return fail(ST_ERR_INVALID_ARGUMENT, "tile_map is 0, 1 or 2");
    if (t.wide_stack_entries != 0u && (t.wide_stack_entries < 8u || t.wide_stack_entries > 56u)) return fail(ST_ERR_INVALID_ARGUMENT, "wide_stack_entries is 0 (default) or 8 ... 56");
    if (t.wide_stack_entries != tuning.wide_stack_entries) wide_stack_rearmed = 0u;   // the caller's figure stands until a walk overflows it
    tuning = t;
    staging.enabled = tuning.staging != 0u;
    return ST_OK;
}
// st_tick found a wide walk's overflow word set (st_device.h wide_walk_overflowed): see st_engine.h walk_flags_host
void Engine::note_walk_overflow() {
    const bool lane = walk_flags_host[0] != 0u, packet = walk_flags_host[1] != 0u;
    walk_flags_host[0] = 0u; walk_flags_host[1] = 0u;   // (a frame still in flight may set them again: the next tick sees that)
    if (!lane && !packet) return;
    walk_overflows++; walk_overflow_unreported = true;
    if (packet) packets_overflowed = true;
    if (lane) {
        const uint32_t now = wide_stack_entries_now();
        wide_stack_rearmed = now < 32u ? 32u : (now < 48u ? 48u : 56u);   // 56: what 64 KB of dynamic LDS hold with 32-bit slots
    }
}
void Engine::reset_profile_totals() {
    for (int i = 0; i < KS_COUNT; i++) {
        memset(&profile_totals[i], 0, sizeof(StKernelProfile));
        snprintf(profile_totals[i].name, sizeof(profile_totals[i].name), "%s", kernel_info(i).name);
    }
}

Engine::~Engine() {
    release_dist();
    if (!has_device) return;
    (void)hipSetDevice(device);
    (void)hipDeviceSynchronize();
    for (auto& kv : cameras) release_camera(*kv.second);
    for (DeviceArray* d : {&d_byte_luts, &d_atlas, &d_blue_noise, &d_transmittance, &d_scattering, &d_sky, &d_mesh_store}) d->release();
    for (LightSet& l : light_sets) { l.buf.release(); if (l.free_ev) (void)hipEventDestroy(l.free_ev); }
    for (SceneSet& t : sets) {
        for (DeviceArray* d : {&t.bvh, &t.tri_attr, &t.xforms, &t.materials, &t.base_packed, &t.tri_geo, &t.tri_bounds, &t.entry_of_tri, &t.parent, &t.refit_local, &t.refit_items, &t.refit_batch_off, &t.bvh_compact, &t.tri_info, &t.lb_keys_a, &t.lb_keys_b, &t.lb_temp, &t.lb_seg, &t.lb_children, &t.lb_node_box, &t.lb_small, &t.bvh_wide, &t.wide_topo, &t.wide_leaf_entry, &t.bake_jobs, &t.bake_starts}) d->release();
        if (t.free_ev) (void)hipEventDestroy(t.free_ev);
    }
    if (copy_stream) (void)hipStreamDestroy(copy_stream);
    if (ev_copy) (void)hipEventDestroy(ev_copy);
    for (auto& r : profile_records) { if (r.owns_start) (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
    for (auto e : event_pool) (void)hipEventDestroy(e);
    if (ev_tick) (void)hipEventDestroy(ev_tick);
    if (walk_flags_host) (void)hipHostFree(const_cast<uint32_t*>(walk_flags_host));
    staging.release();
}

}  // namespace st
