// st_render.cpp — host engine of libstrolle_hip.so: per-camera buffers, the per-frame pass graph (camera_controller.rs:87-174) on two HIP streams, present hand-over. See st_engine.h.
#include "st_engine.h"

namespace st {

void Engine::release_camera(CameraState& c) {
    if (c.slab) (void)hipFree(c.slab);
    if (c.counters) (void)hipFree(c.counters);
    if (c.tile_mask) (void)hipFree(c.tile_mask);
    c.tile_mask = nullptr;
    c.slab = nullptr; c.counters = nullptr;
    if (c.side_stream) (void)hipStreamDestroy(c.side_stream);
    for (hipEvent_t* e : {&c.ev_di_head, &c.ev_gi_done, &c.ev_prim_ok, &c.ev_frame_done, &c.ev_setup}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    c.side_stream = nullptr; c.have_prev_frame_events = false;
    if (c.present_stream) { (void)hipStreamSynchronize(c.present_stream); (void)hipStreamDestroy(c.present_stream); c.present_stream = nullptr; }
    for (auto& p : c.present) { for (hipEvent_t* e : {&p.ev_src, &p.ev_done}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; } p = CameraState::PresentSlot(); }
}

// st_camera_present_copy: `src_device` (what st_render_camera composed into on `stream`) -> `dst_host`, asynchronously
int Engine::present_copy(CameraState& c, const void* src, void* dst, size_t bytes, hipStream_t stream) {
    if (!has_device) return fail(ST_ERR_NO_DEVICE, "present copy on a host-only engine");
    ST_HIP(hipSetDevice(device));
    if (!c.present_stream) ST_HIP(hipStreamCreateWithFlags(&c.present_stream, hipStreamNonBlocking));
    // the slot that already serves this destination, else the older one
    CameraState::PresentSlot* slot = nullptr;
    for (auto& p : c.present) if (p.dst == dst) slot = &p;
    if (!slot) { slot = &c.present[c.present_next & 1u]; c.present_next++; }
    if (slot->pending) ST_HIP(hipEventSynchronize(slot->ev_done));  // only when the caller runs more than two frames ahead
    if (!slot->ev_src) { ST_HIP(hipEventCreateWithFlags(&slot->ev_src, hipEventDisableTiming)); ST_HIP(hipEventCreateWithFlags(&slot->ev_done, hipEventDisableTiming)); }
    slot->src = src; slot->dst = dst;
    ST_HIP(hipEventRecord(slot->ev_src, stream));                    // the frame is composed
    ST_HIP(hipStreamWaitEvent(c.present_stream, slot->ev_src, 0));
    ST_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.present_stream));
    ST_HIP(hipEventRecord(slot->ev_done, c.present_stream));
    slot->pending = true;
    return ST_OK;
}

// 1 = the copy into `dst` has landed (or none was asked for), 0 = still in flight; wait != 0 blocks until it has
int Engine::present_ready(CameraState& c, const void* dst, int wait, int* ready) {
    *ready = 1;
    for (auto& p : c.present) {
        if (p.dst != dst || !p.pending) continue;
        if (wait) { ST_HIP(hipEventSynchronize(p.ev_done)); p.pending = false; }
        else {
            const hipError_t q = hipEventQuery(p.ev_done);
            if (q == hipSuccess) p.pending = false;
            else if (q == hipErrorNotReady) { (void)hipGetLastError(); *ready = 0; }
            else return fail(ST_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
        }
    }
    return ST_OK;
}

// ---- cameras (camera.rs:50-66, camera_controller.rs:27-86)
GpuCamera Engine::serialize_camera(const StCamera& c) {
    const M4 transform = m4_from_cols(c.transform), projection = m4_from_cols(c.projection);
    GpuCamera g;
    g.projection_view = m4_mul(projection, m4_inverse(transform));
    g.ndc_to_world = m4_mul(transform, m4_inverse(projection));
    g.origin = make_float4(transform.c[3].x, transform.c[3].y, transform.c[3].z, 0.0f);
    g.screen = make_float4((float)c.width, (float)c.height, 0.0f, 0.0f);
    return g;
}

int Engine::allocate_camera(CameraState& c) {
    c.row0 = 0; c.row1 = c.desc.height; c.col0 = 0; c.col1 = c.desc.width;
    if (!has_device) return ST_OK;
    ST_HIP(hipSetDevice(device));
    release_camera(c);
    const size_t n = (size_t)c.desc.width * c.desc.height;
    size_t total = 0;
    for (int i = 0; i < ST_BUF_COUNT + kInternalPlanes; i++) {
        c.plane_bytes[i] = i == ST_BUF_DBG_USED_MEMORY ? n * 4 : n * 16 * plane_texels_per_pixel(i);
        total += (c.plane_bytes[i] + 255) & ~size_t(255);
    }
    ST_HIP(hipMalloc(&c.slab, total));
    ST_HIP(hipMemset(c.slab, 0, total));  // wgpu zero-initialises resources; stale-data paths depend on it
    c.slab_bytes = total;
    size_t off = 0;
    for (int i = 0; i < ST_BUF_COUNT + kInternalPlanes; i++) { c.plane[i] = reinterpret_cast<float4*>(static_cast<char*>(c.slab) + off); off += (c.plane_bytes[i] + 255) & ~size_t(255); }
    c.gi_aliased = false;
    if (hipMalloc(reinterpret_cast<void**>(&c.counters), kCounterBytes) != hipSuccess) {
        (void)hipGetLastError(); c.counters = nullptr;
        release_camera(c);  // do not leak the slab
        return fail(ST_ERR_HIP, "hipMalloc(camera counters) failed");
    }
    ST_HIP(hipMemset(c.counters, 0, kCounterBytes));
    {
        const size_t tiles = (size_t)((c.desc.width + 7u) / 8u) * ((c.desc.height + 7u) / 8u);
        if (hipMalloc(reinterpret_cast<void**>(&c.tile_mask), 2 * tiles * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); c.tile_mask = nullptr; release_camera(c); return fail(ST_ERR_HIP, "hipMalloc(camera tile mask) failed"); }
        ST_HIP(hipMemset(c.tile_mask, 0, 2 * tiles * sizeof(unsigned long long)));  // [0, tiles): variance's, [tiles, 2 tiles): the GI preview's
        c.tile_mask_tiles = tiles;
    }
    memset(c.profiled_traversal_bytes, 0, sizeof(c.profiled_traversal_bytes));
    ST_HIP(hipDeviceSynchronize());  // the clears run on the null stream; renders may use any stream
    return ST_OK;
}

// ---- render (camera_controller.rs:87-174)
int Engine::render(CameraState& c, void* out, hipStream_t stream) {
    if (!has_device) return fail(ST_ERR_NO_DEVICE, "render_camera on a host-only engine");
    if (!scene_uploaded) return fail(ST_ERR_INVALID_ARGUMENT, "st_tick must precede st_render_camera");
    ST_HIP(hipSetDevice(device));
    if (tick_work_in_flight) ST_HIP(hipStreamWaitEvent(stream, ev_tick, 0));  // a no-op when st_tick ran on this stream
    if (copy_in_flight) ST_HIP(hipStreamWaitEvent(stream, ev_copy, 0));       // likewise (st_tick already queued this wait on its own stream)
    if (rendered_before && last_render_stream != stream) mixed_render_streams = true;  // the null stream is a stream too
    last_render_stream = stream; rendered_before = true;
    const bool alt = c.frame % 2u == 1u;
    c.last_lean = 0u; c.last_lean_composed = false;
    KArgs a{};
    a.cam = c.curr; a.prev_cam = c.prev;
    const SceneSet& scene = sets[live];
    a.bvh = static_cast<const float4*>(scene.bvh.ptr); a.tri_attr = static_cast<const float4*>(scene.tri_attr.ptr); a.instance_xforms = static_cast<const float4*>(scene.xforms.ptr);
    a.materials = static_cast<const GpuMaterial*>(scene.materials.ptr); a.material_base_packed = tuning.packed_base ? static_cast<const uint32_t*>(scene.base_packed.ptr) : nullptr; a.lights = static_cast<const GpuLight*>(light_sets[live_lights].buf.ptr);
    a.atlas = static_cast<const uchar4*>(d_atlas.ptr); a.blue_noise = static_cast<const uchar4*>(d_blue_noise.ptr); a.byte_luts = static_cast<const float*>(d_byte_luts.ptr);
    a.transmittance_lut = static_cast<const float4*>(d_transmittance.ptr); a.sky_lut = static_cast<const float4*>(d_sky.ptr);
    a.tri_slots = (uint32_t)(tri_geo.size() / 3u);
    a.count_bytes = count_bytes ? 1u : 0u;
    a.exp_flags = exp_flags;
    // the fast build's shadow rays walk the compact stream of this scene copy when it has one (k_bvh.hip k_bvh_compact)
    const bool compact = arithmetic == ST_ARITH_FAST && tuning.compact_bvh && tuning.anyhit_fast && !count_bytes && scene.compact_entries != 0u && scene.compact_entries * 4u == device_bvh_len;
    a.bvh_c = compact ? static_cast<const float4*>(scene.bvh_compact.ptr) : nullptr;
    a.bvh_c_root = (compact && device_root_is_leaf) ? 1u : 0u;
    // ... or, preferred, its wide form (k_bvh.hip k_bvh_wide)
    // (ST_BVH_BUILD_DEVICE: this copy's wide stream was built on the device and its contract stream is stale — every ray must walk the wide stream)
    const bool contract_observer = arithmetic != ST_ARITH_FAST || !tuning.wide_bvh || !tuning.compact_bvh || !tuning.anyhit_fast || count_bytes || c.desc.mode == ST_MODE_BVH_HEATMAP;
    if (scene.device_built && contract_observer)
        return fail(ST_ERR_INVALID_ARGUMENT, "the live scene copy's tree was built on the device (ST_BVH_BUILD_DEVICE) and this frame needs the contract stream (heatmap camera, exact arithmetic, "
                                             "byte counting or a switched-off wide stream): st_tick builds it on the host once it sees the observer");
    const bool wide = scene.device_built || (compact && tuning.wide_bvh && scene.wide_for_entries != 0u && scene.wide_for_entries * 4u == device_bvh_len);
    a.bvh_w = wide ? static_cast<const float4*>(scene.bvh_wide.ptr) : nullptr;
    a.bvh_w_leaf_off = wide ? scene.wide_nodes * 64u : 0u;
    a.bvh_w_root = wide ? scene.wide_root : 0u; a.bvh_w_links16 = wide ? scene.wide_links16 : 0u;
    a.primary_packets = wide && tuning.primary_packets && !packets_overflowed ? 1u : 0u;
    a.walk_flags = walk_flags_dev;
    {   // the largest link of this stream: (max(nodes, leaf records) - 1) << 1 | 1
        uint32_t bits = 16u;
        const uint32_t top = std::max(scene.wide_nodes, scene.wide_leaves);
        while (bits < 31u && (1ull << bits) <= (unsigned long long)top * 2ull + 1ull) bits++;
        a.bvh_w_link_mask = (1u << bits) - 1u;
    }
    a.anyhit_contract = (count_bytes || !tuning.anyhit_fast) ? 1u : 0u;   // the reference's used_memory is the contract loop's
    // Walks over the CONTRACT stream (exact build, heatmap pass, byte-counting mode, the compact binary stream: the contract's tree) hold what
    // that tree's deepest chain can need — proven drop-free up to kBvhStackSizeDeep. The WIDE stream is another tree: its worst case (every
    // child of every node on a path hit: 35 pending entries for the 13 k-triangle dungeon, 45 at 208 k) does not fit LDS at full occupancy and
    // no ray comes near it (deepest stack measured: 11-13); it keeps kBvhStackSize entries, and test_the_wide_walk_drops_no_push renders
    // BASELINE config 3's scene with 24 and with 48 entries (StTuning::wide_stack_entries) and finds the same bits.
    // A walk that does find the stack full says so (KArgs::walk_flags) and the next st_tick re-arms the launches with a deeper one (st_engine.h walk_flags_host).
    a.stack_entries = wide ? wide_stack_entries_now() : stack_entries;
    a.bvh_len = scene.device_built ? 0x40000000u : device_bvh_len;   // (device-built: no contract stream; any value that is neither "empty" nor "fits LDS")
    if (scene.device_built) a.bvh = nullptr; a.n_lights_buf = (uint32_t)gpu_lights.size(); a.light_count = light_count;
    a.atlas_w = atlas_w; a.atlas_h = atlas_h; a.sun_altitude = sun_altitude;
    a.sun_dir[0] = sun_dir_.x; a.sun_dir[1] = sun_dir_.y; a.sun_dir[2] = sun_dir_.z;
    auto P = [&](int id) { return c.plane[id]; };
    a.g0 = P(alt ? ST_BUF_PRIM_GBUFFER_D0_B : ST_BUF_PRIM_GBUFFER_D0_A); a.pg0 = P(alt ? ST_BUF_PRIM_GBUFFER_D0_A : ST_BUF_PRIM_GBUFFER_D0_B);
    a.g1 = P(alt ? ST_BUF_PRIM_GBUFFER_D1_B : ST_BUF_PRIM_GBUFFER_D1_A); a.pg1 = P(alt ? ST_BUF_PRIM_GBUFFER_D1_A : ST_BUF_PRIM_GBUFFER_D1_B);
    a.sm = P(alt ? ST_BUF_PRIM_SURFACE_MAP_B : ST_BUF_PRIM_SURFACE_MAP_A); a.psm = P(alt ? ST_BUF_PRIM_SURFACE_MAP_A : ST_BUF_PRIM_SURFACE_MAP_B);
    a.sn = P(ST_BUF_COUNT + (alt ? 1 : 0)); a.psn = P(ST_BUF_COUNT + (alt ? 0 : 1));
    a.reprojection = P(ST_BUF_REPROJECTION_MAP); a.velocity = P(ST_BUF_VELOCITY_MAP);
    for (int i = 0; i < 3; i++) a.di_res[i] = P(ST_BUF_DI_RESERVOIRS_0 + i);
    a.di_diff_samples = P(ST_BUF_DI_DIFF_SAMPLES); a.di_diff_prev_colors = P(ST_BUF_DI_DIFF_PREV_COLORS); a.di_diff_curr_colors = P(ST_BUF_DI_DIFF_CURR_COLORS);
    a.di_diff_moments = P(alt ? ST_BUF_DI_DIFF_MOMENTS_B : ST_BUF_DI_DIFF_MOMENTS_A); a.di_diff_prev_moments = P(alt ? ST_BUF_DI_DIFF_MOMENTS_A : ST_BUF_DI_DIFF_MOMENTS_B);
    a.di_diff_stash = P(ST_BUF_DI_DIFF_STASH); a.di_spec_samples = P(ST_BUF_DI_SPEC_SAMPLES);
    a.gi_d0 = P(ST_BUF_GI_D0); a.gi_d1 = P(ST_BUF_GI_D1); a.gi_d2 = P(ST_BUF_GI_D2);
    for (int i = 0; i < 4; i++) a.gi_res[i] = P(ST_BUF_GI_RESERVOIRS_0 + i);
    a.gi_diff_samples = P(ST_BUF_GI_DIFF_SAMPLES); a.gi_diff_prev_colors = P(ST_BUF_GI_DIFF_PREV_COLORS); a.gi_diff_curr_colors = P(ST_BUF_GI_DIFF_CURR_COLORS);
    a.gi_diff_moments = P(alt ? ST_BUF_GI_DIFF_MOMENTS_B : ST_BUF_GI_DIFF_MOMENTS_A); a.gi_diff_prev_moments = P(alt ? ST_BUF_GI_DIFF_MOMENTS_A : ST_BUF_GI_DIFF_MOMENTS_B);
    a.gi_diff_stash = P(ST_BUF_GI_DIFF_STASH); a.gi_spec_samples = P(ST_BUF_GI_SPEC_SAMPLES);
    a.ref_hits = P(ST_BUF_REF_HITS); a.ref_rays = P(ST_BUF_REF_RAYS); a.ref_colors = P(ST_BUF_REF_COLORS);
    a.dbg_used_memory = reinterpret_cast<uint32_t*>(P(ST_BUF_DBG_USED_MEMORY));
    a.width = c.desc.width; a.height = c.desc.height; a.row0 = c.row0; a.row1 = c.row1; a.col0 = c.col0; a.col1 = c.col1;
    a.frame = c.frame;
    a.tile_map = tuning.tile_map;

    const double rows = (double)(c.row1 - c.row0);
    const uint32_t cols = c.col1 - c.col0;
    auto slot_bytes = [&](int slot) {
        const KernelInfo& ki = kernel_info(slot);
        const double units = rows * (ki.half ? (double)(((cols + 7u) / 8u / 2u) * 8u) : (double)cols);
        return units * ki.bytes_per_unit;
    };
    hipStream_t cur = stream;  // stream the next launches go to (the GI chain may be diverted to side_stream)
    // `bits`: the reference passes this launch executes (StPassBit). Their unfused algorithmic bytes are what
    // kernel_info(slot) credits to the launch, so fusion shows up as a gain, not as a moved goalpost.
    last_launches.clear();
    bool mask_split = false;
    uint32_t launch_ordinal = 0u;
    auto run = [&](int slot, uint64_t bits, auto&& launch) {
        if (last_launches.empty() || last_launches.back() != bits) last_launches.push_back(bits);  // a launch group is reported once
        if ((bits & pass_mask) != bits) { mask_split |= (bits & pass_mask) != 0; return; }
        if (launch_filter != ~0ull && !((launch_filter >> (launch_ordinal++ & 63u)) & 1ull)) return;  // measurement only: the frame's state is not meaningful afterwards
        const double bytes = slot_bytes(slot);
        a.ray_counter = c.counters + kCounterWordsPerSlot * slot;
        if (profiling && profile_kernel_events) {  // the dispatch's own timestamps (what rocprofv3's kernel trace reads)
            (void)profile_close();   // a scope the run-of-launches mode left open belongs to that mode
            g_launch_events.start = take_event(); g_launch_events.stop = take_event(); g_launch_events.consumed = false;
            launch();
            if (g_launch_events.consumed) profile_records.push_back({slot, g_launch_events.start, g_launch_events.stop, bytes, 1u, true});
            else { event_pool.push_back(g_launch_events.start); event_pool.push_back(g_launch_events.stop); }   // nothing was enqueued (an empty grid)
            g_launch_events = LaunchEvents();
            return;
        }
        const bool atrous = slot == KS_DENOISE_WAVELET || slot == KS_DENOISE_WAVELET_12 || slot == KS_DENOISE_WAVELET_COMPOSE;
        profile_begin(profile_group_atrous && atrous ? (int)KS_DENOISE_WAVELET_FAMILY : slot, cur, bytes);
        launch();
    };
    auto seed = [&](uint32_t pass) { return pass_seed(base_seed, c.frame, pass); };
    const uint32_t mode = c.desc.mode;
    bool di_reprojected = false, gi_reprojected = false, composed = false, luts_generated_now = false;
    if (c.surface_map_replaced[0] || c.surface_map_replaced[1]) {  // ordered before the side stream like the LUTs
        const uint32_t cur = alt ? 1u : 0u;
        L.launch_refresh_internal_planes(a, (c.surface_map_replaced[cur] ? 1u : 0u) | (c.surface_map_replaced[cur ^ 1u] ? 2u : 0u), stream);
        c.surface_map_replaced[0] = c.surface_map_replaced[1] = false; luts_generated_now = true;
    }
    if (mode != ST_MODE_BVH_HEATMAP) {  // AtmospherePass::run (passes/atmosphere.rs:78-110)
        if (!atmosphere_initialized) {
            L.launch_atmosphere_static(static_cast<float4*>(d_transmittance.ptr), static_cast<float4*>(d_scattering.ptr), stream);
            atmosphere_initialized = true; luts_generated_now = true;
        }
        if (!sky_known || known_sun_altitude != sun_altitude) {
            L.launch_atmosphere_sky(static_cast<const float4*>(d_transmittance.ptr), static_cast<const float4*>(d_scattering.ptr), sun_altitude,
                                  static_cast<float4*>(d_sky.ptr), stream);
            sky_known = true; known_sun_altitude = sun_altitude; luts_generated_now = true;
        }
    }
    if (mode == ST_MODE_BVH_HEATMAP) {
        run(KS_BVH_HEATMAP, ST_PASS_BVH_HEATMAP, [&] { L.launch_bvh_heatmap(a, cur); });
    } else if (mode == ST_MODE_REFERENCE) {
        for (uint32_t d = 0; d <= c.desc.depth; d++) {
            run(KS_REF_TRACING, ST_PASS_REF_TRACING, [&] { L.launch_ref_tracing(a, d, cur); });
            run(KS_REF_SHADING, ST_PASS_REF_SHADING, [&] { L.launch_ref_shading(a, seed(SEED_REF_SHADING + d), d, cur); });
        }
        run(KS_REF_SHADING, ST_PASS_REF_SHADING, [&] { L.launch_ref_shading(a, seed(SEED_REF_SHADING + 255u), 255u, cur); });
    } else {
        const bool needs_di = mode == ST_MODE_IMAGE || mode == ST_MODE_DI_DIFFUSE || mode == ST_MODE_DI_SPECULAR;
        const bool needs_gi = mode == ST_MODE_IMAGE || mode == ST_MODE_GI_DIFFUSE || mode == ST_MODE_GI_SPECULAR;
        const bool denoise = c.desc.denoise != 0u;
        const bool any_objects = !instances.empty();
        const bool tracing = c.frame % 6u < 4u;
        const uint32_t gi_source = (tracing && c.frame % 2u == 1u) ? 1u : 0u;
        const uint32_t pseed = seed(SEED_GI_PREVIEW);  // one seed for both preview passes (passes/gi_preview_resampling.rs:60-74)
        // GI history hand-over by pointer swap instead of gi_resolving's copy (CameraState::gi_aliased says when)
        const bool whole_graph = pass_mask == ~0ull;  // a row window (multi-GPU band) changes which pixels a pass owns, not which passes follow it
        const bool gi_runs = needs_gi && any_objects;
        if (c.gi_aliased && gi_runs && !whole_graph) { const int rc = materialize_gi_history(c); if (rc) return rc; }
        // Fast build only: the reference's copy is a decode + re-encode of every reservoir, which is not the identity on all
        // bit patterns (the octahedral normal of a few records per frame moves by an ulp), and the exact build owes the
        // parity suite those bits.
        const bool swap_gi_history = tuning.alias_gi_history && arithmetic == ST_ARITH_FAST && gi_runs && whole_graph && gi_source == 0u;
        if (gi_runs && whole_graph) c.gi_aliased = false;  // this frame's temporal pass rewrites GI_RESERVOIRS_1 completely
        a.gi_skip_history_copy = swap_gi_history ? 1u : 0u;
        // estimate_variance's long-history branch rides in the fused reproject stages (st_passes.h denoise_reproject_finish);
        // the variance launch then serves the short-history pixels only, in place, and the strides-1+2 launch reads curr_colors
        a.tile_mask = c.tile_mask;
        // both GI preview passes + resolving in one launch for the pixels whose second pass draws no neighbour (k_gi.hip
        // k_gi_preview_both); the second-pass launch then serves the flagged rest
        a.gi_late_mask = c.tile_mask ? c.tile_mask + c.tile_mask_tiles : nullptr; a.gi_preview_late = 0u;
        const bool gi_preview_both = tuning.preview_both && whole_graph && tuning.fuse && gi_runs && a.gi_late_mask;
        a.variance_in_reproject = (tuning.variance_in_reproject && whole_graph && tuning.fuse && tuning.fuse_wavelet && denoise && needs_di && needs_gi && any_objects && c.tile_mask) ? 1u : 0u;
        // di_spatial's scratch records (di_diff_samples / curr_colors / stash as the reference binds them) are dead stores
        // when the fused launch is followed by resolving, denoise-reproject and the a-trous chain of the same frame
        const bool even_tiles_x = (((a.width + 7u) / 8u) & 1u) == 0u;
        a.lean = 0u;
        if (tuning.lean_frame && arithmetic == ST_ARITH_FAST && whole_graph && tuning.fuse && denoise && any_objects && mode == ST_MODE_IMAGE) {
            a.lean = kLeanPrim | kLeanSamples;
            if (tuning.fuse_gi_reprojection && tracing && even_tiles_x) a.lean |= kLeanGiRes2;
            if (gi_preview_both) a.lean |= kLeanGiMid;
        }
        c.last_lean = a.lean; c.last_lean_composed = false;
        // frame composition rides in the last a-trous pass (k_denoise.hip k_denoise_wavelet_far<true>)
        const bool compose_in_wavelet = tuning.fuse_compose && arithmetic == ST_ARITH_FAST && whole_graph && tuning.fuse && denoise && out != nullptr && mode == ST_MODE_IMAGE && any_objects;
        c.last_lean_composed = compose_in_wavelet && a.lean != 0u;   // the last a-trous pass's colour planes stay unwritten
        a.skip_dead_scratch = (tuning.skip_scratch_stores && whole_graph && tuning.fuse && tuning.fuse_spatial && ((((a.width + 7u) / 8u) & 1u) == 0u) && needs_di && denoise && any_objects) ? 1u : 0u;

        auto do_prim = [&] {
            if (tuning.fuse && any_objects) run(KS_PRIM_VISIBILITY_REPROJECTION, ST_PASS_PRIM_VISIBILITY | ST_PASS_FRAME_REPROJECTION, [&] { L.launch_prim_visibility(a, true, cur); });
            else run(KS_PRIM_VISIBILITY, ST_PASS_PRIM_VISIBILITY, [&] { L.launch_prim_visibility(a, false, cur); });
            if (any_objects && !tuning.fuse) run(KS_FRAME_REPROJECTION, ST_PASS_FRAME_REPROJECTION, [&] { L.launch_frame_reprojection(a, cur); });
        };
        // DI up to temporal resampling touches only the DI reservoirs and read-only frame inputs ...
        auto do_di_head = [&] {
            if (tuning.fuse && tuning.fuse_di_head) run(KS_DI_SAMPLING_TEMPORAL, ST_PASS_DI_SAMPLING | ST_PASS_DI_TEMPORAL, [&] { L.launch_di_sampling_temporal(a, seed(SEED_DI_SAMPLING), seed(SEED_DI_TEMPORAL), cur); });
            else {
                run(KS_DI_SAMPLING, ST_PASS_DI_SAMPLING, [&] { L.launch_di_sampling(a, seed(SEED_DI_SAMPLING), cur); });
                run(KS_DI_TEMPORAL, ST_PASS_DI_TEMPORAL, [&] { L.launch_di_temporal(a, seed(SEED_DI_TEMPORAL), cur); });
            }
        };
        // ... the spatial passes use the denoiser's planes as scratch (passes/di_spatial_resampling.rs binds
        // di_diff_samples / curr_colors / stash), and resolving writes the planes the denoiser reads
        auto do_di_tail = [&] {
            // the half-resolution grid drops the last tile column when the tile count is odd (`(size + 7) / 8 / (2, 1)`), while
            // the stand-alone trace pass still visits those pixels: only an even tile count lets one launch cover all three
            const bool even_tiles = (((a.width + 7u) / 8u) & 1u) == 0u;
            if (tuning.fuse && tuning.fuse_spatial && even_tiles) run(KS_DI_SPATIAL_FUSED, ST_PASS_DI_SPATIAL_PICK | ST_PASS_DI_SPATIAL_TRACE | ST_PASS_DI_SPATIAL_SAMPLE, [&] { L.launch_di_spatial_fused(a, seed(SEED_DI_SPATIAL_PICK), seed(SEED_DI_SPATIAL_SAMPLE), cur); });
            else {
                run(KS_DI_SPATIAL_PICK, ST_PASS_DI_SPATIAL_PICK, [&] { L.launch_di_spatial_pick(a, seed(SEED_DI_SPATIAL_PICK), cur); });
                run(KS_DI_SPATIAL_TRACE, ST_PASS_DI_SPATIAL_TRACE, [&] { L.launch_spatial_trace(a, a.di_diff_samples, a.di_diff_curr_colors, a.di_diff_stash, cur); });
                run(KS_DI_SPATIAL_SAMPLE, ST_PASS_DI_SPATIAL_SAMPLE, [&] { L.launch_di_spatial_sample(a, seed(SEED_DI_SPATIAL_SAMPLE), cur); });
            }
            if (tuning.fuse && denoise) { run(KS_DI_RESOLVING_REPROJECT, ST_PASS_DI_RESOLVING | ST_PASS_DENOISE_REPROJECT_DI, [&] { L.launch_di_resolving(a, true, cur); }); di_reprojected = true; }
            else run(KS_DI_RESOLVING, ST_PASS_DI_RESOLVING, [&] { L.launch_di_resolving(a, false, cur); });
        };
        auto do_di = [&] { do_di_head(); do_di_tail(); };
        // GI up to the first preview pass: touches only reservoirs, gi_d0..2 and read-only frame inputs
        auto do_gi_head = [&] {
            // on tracing frames gi_temporal is the only reader of the reprojected reservoirs and does the reprojection itself
            // ... and on validation frames of a whole frame both of its readers — the sampling launch for the half of the pixels it
            // re-traces, then gi_temporal, which stores it — do it for themselves (ST_NO_FUSE_GI_VALIDATION=1: a launch of its own)
            const bool fuse_gi_validation_now = tuning.fuse && tuning.fuse_gi_reprojection && tuning.fuse_gi_sampling && tuning.fuse_gi_validation && !tracing && whole_graph;
            const bool fuse_gi_reprojection_now = (tuning.fuse && tuning.fuse_gi_reprojection && tracing) || fuse_gi_validation_now;
            auto temporal = [&] {
                if (fuse_gi_reprojection_now) run(KS_GI_REPROJECTION_TEMPORAL, ST_PASS_GI_REPROJECTION | ST_PASS_GI_TEMPORAL, [&] { L.launch_gi_temporal(a, seed(SEED_GI_TEMPORAL), true, cur); });
                else run(KS_GI_TEMPORAL, ST_PASS_GI_TEMPORAL, [&] { L.launch_gi_temporal(a, seed(SEED_GI_TEMPORAL), false, cur); });
            };
            if (!fuse_gi_reprojection_now) run(KS_GI_REPROJECTION, ST_PASS_GI_REPROJECTION, [&] { L.launch_gi_reprojection(a, cur); });
            auto sampling = [&] {
                if (tuning.fuse && tuning.fuse_gi_sampling) { run(KS_GI_SAMPLING_AB, ST_PASS_GI_SAMPLING_A | ST_PASS_GI_SAMPLING_B, [&] { L.launch_gi_sampling_ab(a, seed(SEED_GI_SAMPLING_A), seed(SEED_GI_SAMPLING_B), fuse_gi_validation_now, cur); }); return; }
                run(KS_GI_SAMPLING_A, ST_PASS_GI_SAMPLING_A, [&] { L.launch_gi_sampling_a(a, seed(SEED_GI_SAMPLING_A), cur); });
                run(KS_GI_SAMPLING_B, ST_PASS_GI_SAMPLING_B, [&] { L.launch_gi_sampling_b(a, seed(SEED_GI_SAMPLING_B), cur); });
            };
            if (tracing) {
                if (c.frame % 2u == 0u) sampling();
                temporal();
                if (c.frame % 2u == 1u) {
                    if (tuning.fuse && tuning.fuse_spatial && ((((a.width + 7u) / 8u) & 1u) == 0u))
                        run(KS_GI_SPATIAL_FUSED, ST_PASS_GI_SPATIAL_PICK | ST_PASS_GI_SPATIAL_TRACE | ST_PASS_GI_SPATIAL_SAMPLE, [&] { L.launch_gi_spatial_fused(a, seed(SEED_GI_SPATIAL_PICK), seed(SEED_GI_SPATIAL_SAMPLE), cur); });
                    else {
                        run(KS_GI_SPATIAL_PICK, ST_PASS_GI_SPATIAL_PICK, [&] { L.launch_gi_spatial_pick(a, seed(SEED_GI_SPATIAL_PICK), cur); });
                        run(KS_GI_SPATIAL_TRACE, ST_PASS_GI_SPATIAL_TRACE, [&] { L.launch_spatial_trace(a, a.gi_d0, a.gi_d1, a.gi_d2, cur); });
                        run(KS_GI_SPATIAL_SAMPLE, ST_PASS_GI_SPATIAL_SAMPLE, [&] { L.launch_gi_spatial_sample(a, seed(SEED_GI_SPATIAL_SAMPLE), cur); });
                    }
                }
            } else {
                sampling();
                temporal();
            }
            if (!gi_preview_both) run(KS_GI_PREVIEW, ST_PASS_GI_PREVIEW_0, [&] { L.launch_gi_preview(a, pseed, 0u, gi_source == 0 ? a.gi_res[1] : a.gi_res[2], a.gi_res[3], cur); });
        };
        // second preview pass + resolving (+ reproject): the first GI stage that writes planes the denoiser/composition read
        auto do_gi_tail = [&] {
            if (gi_preview_both) {
                // one launch group of two kernels = one set of pass bits
                const uint64_t group = ST_PASS_GI_PREVIEW_0 | ST_PASS_GI_PREVIEW_1 | ST_PASS_GI_RESOLVING | (denoise ? (uint64_t)ST_PASS_DENOISE_REPROJECT_GI : 0ull);
                run(denoise ? KS_GI_PREVIEW_BOTH : KS_GI_PREVIEW_BOTH_NO_REPROJECT, group, [&] { L.launch_gi_preview_both(a, pseed, gi_source == 0 ? a.gi_res[1] : a.gi_res[2], a.gi_res[3], gi_source, denoise, cur); });
                a.gi_preview_late = 1u;
                a.gi_mid_src = (a.lean & kLeanGiMid) ? (gi_source == 0 ? a.gi_res[1] : a.gi_res[2]) : nullptr;
                run(KS_GI_PREVIEW_LATE, group, [&] { L.launch_gi_preview_resolve(a, pseed, 1u, a.gi_res[3], gi_source, denoise, cur); });
                a.gi_preview_late = 0u; a.gi_mid_src = nullptr;
                if (denoise) gi_reprojected = true;
            } else if (tuning.fuse) {
                if (denoise) { run(KS_GI_PREVIEW_RESOLVE_REPROJECT, ST_PASS_GI_PREVIEW_1 | ST_PASS_GI_RESOLVING | ST_PASS_DENOISE_REPROJECT_GI, [&] { L.launch_gi_preview_resolve(a, pseed, 1u, a.gi_res[3], gi_source, true, cur); }); gi_reprojected = true; }
                else run(KS_GI_PREVIEW_RESOLVE, ST_PASS_GI_PREVIEW_1 | ST_PASS_GI_RESOLVING, [&] { L.launch_gi_preview_resolve(a, pseed, 1u, a.gi_res[3], gi_source, false, cur); });
            } else {
                run(KS_GI_PREVIEW, ST_PASS_GI_PREVIEW_1, [&] { L.launch_gi_preview(a, pseed, 1u, a.gi_res[3], a.gi_res[0], cur); });
                run(KS_GI_RESOLVING, ST_PASS_GI_RESOLVING, [&] { L.launch_gi_resolving(a, gi_source, cur); });
            }
            if (swap_gi_history) {  // the launches above were told not to copy (KArgs::gi_skip_history_copy)
                std::swap(c.plane[ST_BUF_GI_RESERVOIRS_0], c.plane[ST_BUF_GI_RESERVOIRS_1]);
                c.gi_aliased = true;
            }
        };
        auto do_denoise = [&] {
            if (!denoise) return;
            // the denoiser can use its own block -> tile mapping (see `tile_map_denoise`)
            struct MapScope { KArgs& a; uint32_t saved; MapScope(KArgs& a_, uint32_t m) : a(a_), saved(a_.tile_map) { a.tile_map = m; } ~MapScope() { a.tile_map = saved; } } map_scope(a, tuning.tile_map_denoise);
            if (!di_reprojected) run(KS_DENOISE_REPROJECT, ST_PASS_DENOISE_REPROJECT_DI, [&] { L.launch_denoise_reproject(a, a.di_diff_prev_colors, a.di_diff_prev_moments, a.di_diff_samples, a.di_diff_curr_colors, a.di_diff_moments, cur); });
            if (!gi_reprojected) run(KS_DENOISE_REPROJECT, ST_PASS_DENOISE_REPROJECT_GI, [&] { L.launch_denoise_reproject(a, a.gi_diff_prev_colors, a.gi_diff_prev_moments, a.gi_diff_samples, a.gi_diff_curr_colors, a.gi_diff_moments, cur); });
            // ping-pong (passes/frame_denoising.rs:87-110): stash -> prev -> stash -> curr -> stash -> curr
            float4* di[3] = {a.di_diff_stash, a.di_diff_prev_colors, a.di_diff_curr_colors};
            float4* gi[3] = {a.gi_diff_stash, a.gi_diff_prev_colors, a.gi_diff_curr_colors};
            const int in_ix[5] = {0, 1, 0, 2, 0}, out_ix[5] = {1, 0, 2, 0, 2};
            uint32_t first = 0;
            if (tuning.fuse && tuning.fuse_wavelet) {
                // variance estimation + strides 1 and 2 form one launch group of two kernels: the variance pass hands its
                // output over in an internal pair of planes (k_denoise.hip k_denoise_wavelet_12 says why), so the stash
                // planes receive the stride-2 result directly. One group = one set of pass bits (st_debug_set_pass_mask).
                const uint64_t group = ST_PASS_DENOISE_VARIANCE | ST_PASS_DENOISE_WAVELET_0 | ((uint64_t)ST_PASS_DENOISE_WAVELET_0 << 1);
                // (with KArgs::variance_in_reproject the hand-over planes are the reproject stages' own outputs)
                float4* tmp_di = a.variance_in_reproject ? a.di_diff_curr_colors : P(ST_BUF_COUNT + 2);
                float4* tmp_gi = a.variance_in_reproject ? a.gi_diff_curr_colors : P(ST_BUF_COUNT + 3);
                run(KS_DENOISE_VARIANCE, group, [&] { L.launch_denoise_variance(a, tmp_di, tmp_gi, cur); });
                run(KS_DENOISE_WAVELET_12, group, [&] { L.launch_denoise_wavelet_12(a, 1.0f, 2.0f, tmp_di, di[1], di[0], tmp_gi, gi[1], gi[0], cur); });
                first = 2;
            } else run(KS_DENOISE_VARIANCE, ST_PASS_DENOISE_VARIANCE, [&] { L.launch_denoise_variance(a, a.di_diff_stash, a.gi_diff_stash, cur); });
            for (uint32_t nth = first; nth < 5; nth++) {
                if (nth == 4u && compose_in_wavelet) {
                    present_guard(c, out, cur); dist_guard(c.handle, out, cur);
                    run(KS_DENOISE_WAVELET_COMPOSE, ((uint64_t)ST_PASS_DENOISE_WAVELET_0 << nth) | ST_PASS_COMPOSITION, [&] {
                        L.launch_denoise_wavelet_compose(a, 1u << nth, (float)(1u + nth), di[in_ix[nth]], di[out_ix[nth]], gi[in_ix[nth]], gi[out_ix[nth]], mode, out, c.out_format, a.lean == 0u, cur); });
                    composed = true;
                    continue;
                }
                run(KS_DENOISE_WAVELET, (uint64_t)ST_PASS_DENOISE_WAVELET_0 << nth, [&] { L.launch_denoise_wavelet(a, 1u << nth, (float)(1u + nth), di[in_ix[nth]], di[out_ix[nth]], gi[in_ix[nth]], gi[out_ix[nth]], cur); });
            }
        };
        auto do_compose = [&] {
            if (!out || composed) return;
            present_guard(c, out, cur); dist_guard(c.handle, out, cur);
            const float4* di_diff = (denoise && (mode == ST_MODE_IMAGE || mode == ST_MODE_DI_DIFFUSE)) ? a.di_diff_curr_colors : a.di_diff_samples;
            const float4* gi_diff = (denoise && (mode == ST_MODE_IMAGE || mode == ST_MODE_GI_DIFFUSE)) ? a.gi_diff_curr_colors : a.gi_diff_samples;
            run(KS_COMPOSITION, ST_PASS_COMPOSITION, [&] { L.launch_composition(a, mode, di_diff, gi_diff, out, c.out_format, cur); });
            composed = true;
        };

        // per-kernel profiling runs the graph serially on `stream`: a launch's event pair then times that kernel alone,
        // not the kernels of the other stream it would share the chip with
        if (tuning.overlap && !profiling && launch_filter == ~0ull && needs_di && needs_gi && any_objects) {
            // Two streams, software-pipelined across frames: `side` carries primary visibility and the GI chain; `stream`
            // carries the DI passes (sampling + temporal resampling too, by default: measured 1.2 % on the dungeon, nothing
            // on Cornell, against running them behind primary visibility on `side`), the denoiser and composition. Events
            // express the true data dependencies only, so the reservoir passes of frame N+1 overlap the denoiser of frame N:
            //   prim(N+1)      after DI tail(N)       — it overwrites frame N's "previous" G-buffer + the reprojection map
            //   GI tail(N+1)   after frame N is done  — it writes gi sample/colour/moment planes the denoiser + composition read
            //   DI head(N+1)   after prim(N+1)        (ev_di_head)
            //   DI tail(N+1)   after DI head(N+1)     (and after frame N's composition by stream order: its scratch aliases
            //                                          the denoiser's planes)
            //   denoiser(N+1)  after GI tail(N+1)
            if (!c.side_stream) {
                int least = 0, greatest = 0;
                (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                const int priority = tuning.side_priority > 0 ? greatest : (tuning.side_priority < 0 ? least : 0);
                ST_HIP(hipStreamCreateWithPriority(&c.side_stream, hipStreamNonBlocking, priority));
                for (hipEvent_t* e : {&c.ev_di_head, &c.ev_gi_done, &c.ev_prim_ok, &c.ev_frame_done, &c.ev_setup}) ST_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
            }
            // LUT generation issued on `stream` in this call must precede the side stream's consumers. (Do NOT do this
            // unconditionally: an event recorded on `stream` here completes only after frame N's denoiser, which would
            // serialise prim(N+1) behind it. Uploads in st_tick are followed by a host-side stream sync.)
            if (luts_generated_now) { ST_HIP(hipEventRecord(c.ev_setup, stream)); ST_HIP(hipStreamWaitEvent(c.side_stream, c.ev_setup, 0)); }
            // copies st_tick queued without joining the stream (staged uploads, dynamic images): they sit behind frame N on
            // the tick's stream, so a frame that follows a scene change gives up the prim(N+1) / denoiser(N) overlap
            if (tick_work_in_flight) ST_HIP(hipStreamWaitEvent(c.side_stream, ev_tick, 0));
            if (copy_in_flight) ST_HIP(hipStreamWaitEvent(c.side_stream, ev_copy, 0));  // independent of frame N: the overlap stays
            if (c.have_prev_frame_events) ST_HIP(hipStreamWaitEvent(c.side_stream, c.ev_prim_ok, 0));
            cur = c.side_stream;
            do_prim();
            if (!tuning.di_head_on_main) do_di_head();
            ST_HIP(hipEventRecord(c.ev_di_head, c.side_stream));  // primary visibility (+ DI head) of this frame are through
            do_gi_head();
            if (c.have_prev_frame_events) ST_HIP(hipStreamWaitEvent(c.side_stream, c.ev_frame_done, 0));
            do_gi_tail();
            ST_HIP(hipEventRecord(c.ev_gi_done, c.side_stream));
            cur = stream;
            ST_HIP(hipStreamWaitEvent(stream, c.ev_di_head, 0));
            if (tuning.di_head_on_main) do_di_head();
            do_di_tail();
            // stand-alone denoise reprojection kernels (unfused path) still read the reprojection map: prim(N+1) may
            // only start once they are through
            const bool reproject_later = denoise && !tuning.fuse;
            if (!reproject_later) ST_HIP(hipEventRecord(c.ev_prim_ok, stream));
            ST_HIP(hipStreamWaitEvent(stream, c.ev_gi_done, 0));
            do_denoise();
            if (reproject_later) ST_HIP(hipEventRecord(c.ev_prim_ok, stream));
            do_compose();
            ST_HIP(hipEventRecord(c.ev_frame_done, stream));
            c.have_prev_frame_events = true;
        } else {
            do_prim();
            if (any_objects) {
                if (needs_di) do_di();
                if (needs_gi) { do_gi_head(); do_gi_tail(); }
            }
            do_denoise();
            do_compose();
            if (c.side_stream) { ST_HIP(hipEventRecord(c.ev_prim_ok, stream)); ST_HIP(hipEventRecord(c.ev_frame_done, stream)); }
        }
    }
    if (out && !composed) {
        present_guard(c, out, cur); dist_guard(c.handle, out, cur);
        const bool dn = c.desc.denoise != 0u;
        const float4* di_diff = (dn && (mode == ST_MODE_IMAGE || mode == ST_MODE_DI_DIFFUSE)) ? a.di_diff_curr_colors : a.di_diff_samples;
        const float4* gi_diff = (dn && (mode == ST_MODE_IMAGE || mode == ST_MODE_GI_DIFFUSE)) ? a.gi_diff_curr_colors : a.gi_diff_samples;
        run(KS_COMPOSITION, ST_PASS_COMPOSITION, [&] { L.launch_composition(a, mode, di_diff, gi_diff, out, c.out_format, cur); });
    }
    if (alternating) {  // the end of the last frame that reads this copy of the scene
        SceneSet& l = sets[live];
        if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
        ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
    }
    if (lights_alternating) {
        LightSet& l = light_sets[live_lights];
        if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
        ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
    }
    profile_close();
    ST_HIP(hipGetLastError());
    if (mask_split) return fail(ST_ERR_INVALID_ARGUMENT, "the pass mask splits a fused launch (st_debug_last_launches lists the launch groups)");
    return ST_OK;
}

}  // namespace st
