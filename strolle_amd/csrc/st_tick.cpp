// st_tick.cpp — host engine of libstrolle_hip.so: Engine::tick (lib.rs:301-395): refresh of the stores + uploads. See st_engine.h.
#include "st_engine.h"

#include <cstdlib>
namespace st {

constexpr uint32_t kDeviceRefitsPerBuild = 15;   // ST_BVH_BUILD_DEVICE: moves-only ticks answered by a refit of the device-built tree between two builds

// The host's binned-SAH tree of the scene as it is now (st_bvh.h: the reference's tree, with unchanged subtrees reused) and its flattened stream.
void Engine::rebuild_host_tree(bool timing) {
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    if (any_host_stale()) bake_stale_on_host();
    std::vector<uint8_t> blend(materials.size());
    for (size_t i = 0; i < materials.size(); i++) blend[i] = materials[i].alpha_mode == 1u;
    const auto t1 = now();
    bvh.begin_refresh();  // keeps the previous tree: unchanged subtrees are copied, not rebuilt (same result as a fresh build)
    for (size_t i = 0; i < prims.size(); i++) if (prim_alive[i]) bvh.prims.push_back(prims[i]);
    const auto t2 = now();
    bvh.run();
    const auto t3 = now();
    bvh.flatten(blend, bvh_stream);
    const auto t4 = now();
    {   // what ST_BVH_AUTO's choice of the first tree rests on (device_build_possible): the surface-area-weighted mean length of this tree's leaf runs
        double weighted = 0.0, area = 0.0;
        std::vector<uint32_t> todo;
        if (!bvh.nodes.empty() && !bvh.prims.empty()) todo.push_back(0u);
        while (!todo.empty()) {
            const auto& n = bvh.nodes[todo.back()]; todo.pop_back();
            if (n.internal) { todo.push_back(n.left); todo.push_back(n.right); continue; }
            const double a = (double)n.bounds.half_area();
            if (a > 0.0 && a < 1e300) { weighted += a * (double)(n.end - n.begin); area += a; }
        }
        host_leaf_run_weight = area > 0.0 ? (float)(weighted / area) : 1.0f;
    }
    rebuilds++; tree_version++; host_stream_stale = false; host_tree_stale = false;
    mark_internal_starts(); measure_stack_need();
    have_topology = false;
    if (timing) fprintf(stderr, "[st_tick] gather %.2f ms, bvh build %.2f ms, flatten %.2f ms (%zu triangles, %zu reused)\n", ms(t1, t2), ms(t2, t3), ms(t3, t4), bvh.prims.size(), bvh.reused_primitives());
}
// ST_BVH_BUILD_DEVICE applies while nothing observes the contract stream: the fast build's rays walk the wide stream (StTuning::wide_bvh and
// what it rests on), no camera draws the heatmap, the reference's traversal bytes are not being counted — and there is a scene to sort.
// ST_BVH_AUTO (the default since round 6): the FIRST tree of an engine is the host's — the reference's binned SAH, the better tree, paid once while a scene
// loads — and every CHANGE after it (spawn, despawn, move) goes to the device under the same conditions; scenes whose contract stream fits LDS
// (k_common.h scene_fits_lds: the Cornell box) keep the host's tree, which their kernels walk from LDS with the exact closest-hit loop.
bool Engine::device_build_possible() const {
    // ST_BVH_AUTO: a scene that fits LDS keeps its contract stream there (a leaf entry per triangle: more than 112 of them never fit); a larger one sends its
    // CHANGES to the device builder — and its FIRST tree too when the host's tree, built first, turns out to hang long leaf runs on large faces
    // (auto_first_on_device: tick() sets it when rebuild_host_tree measured host_leaf_run_weight > kAutoLeafRunLimit). Measured on 17 scene x mode rows of 13 k to 537 k
    // triangles (tools/tree_choice.py, profiles/r06_tree_choice_auto.txt): at a weight of 3.0 or less the frames over the host's binned-SAH tree are up to 7 % faster than over
    // the device's LBVH (two rows: the device's by 1 and 6 %); at 3.7 the device's is 1-2 % faster; at 4.2-4.5 (the dungeon with its level split x16: runs of up to 140
    // coplanar triangles, a wide walk's step each) 5-17 % faster. A triangle COUNT does not separate them: 16 instanced copies of the level (139 k triangles, weight 1.8)
    // render 1-7 % faster on the host's tree, the x16-split level (134 k, weight 4.5) 12-17 % faster on the device's.
    const bool automatic = bvh_refresh_mode == ST_BVH_AUTO && (scene_uploaded || auto_first_on_device) && live_prims_ > kLdsSceneTexels / 4u;
    if (!(bvh_refresh_mode == ST_BVH_BUILD_DEVICE || automatic) || !has_device || arithmetic != ST_ARITH_FAST) return false;
    if (!tuning.wide_bvh || !tuning.compact_bvh || !tuning.anyhit_fast || count_bytes) return false;
    for (const auto& kv : cameras) if (kv.second->desc.mode == ST_MODE_BVH_HEATMAP) return false;
    return live_prims_ >= 2u && prims.size() < (1u << 23);
}
// The device builder's scratch and output arrays of one scene copy, for `slots` triangle slots of which `live` are live: grown with headroom, never shrunk.
// Also called ahead of time (tick, after the first upload of an engine whose later changes will be answered on the device): a spawn tick then allocates
// nothing (round 5: the first device build of each copy took 5-15 ms, all of it hipMalloc).
int Engine::reserve_device_builder(SceneSet& t, size_t slots, uint32_t live) {
    if (live < 2u) return ST_OK;
    const uint32_t pow2 = lbvh_pow2(live);
    const size_t temp = lbvh_sort_temp_bytes((uint32_t)slots);
    auto need = [&](DeviceArray& d, size_t bytes) -> int {
        if (bytes <= d.capacity) return ST_OK;
        if (d.ptr) ST_HIP(hipFree(d.ptr));
        d.ptr = nullptr; d.capacity = 0;
        ST_HIP(hipMalloc(&d.ptr, bytes + bytes / 4)); d.capacity = bytes + bytes / 4;
        return ST_OK;
    };
    int rc;
    if ((rc = need(t.lb_keys_a, slots * 8u)) || (rc = need(t.lb_keys_b, slots * 8u)) || (rc = need(t.lb_temp, std::max<size_t>(temp + temp / 4, 16u))) || (rc = need(t.lb_seg, (size_t)pow2 * 2u * 32u)) ||
        (rc = need(t.lb_children, (size_t)live * 8u)) || (rc = need(t.lb_node_box, (size_t)live * 32u)) ||
        (rc = need(t.lb_small, 64u)) || (rc = need(t.bvh_wide, (size_t)(live - 1u) * 64u + (size_t)live * 48u + 64u))) return rc;
    return ST_OK;
}
// This device copy's triangle arrays brought up to date, then k_lbvh.hip builds its wide stream from them.
int Engine::build_on_device(SceneSet& t, hipStream_t up, bool* pageable) {
    int rc;
    const size_t slots = prims.size();
    const bool timing = tuning.tick_timing;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    if (tri_info_built_for_ != tri_info_serial_ || tri_info_.size() != slots) {   // (a tick in which instances only moved changes none of this)
        auto word = [&](size_t i) {
            const uint32_t mat = prims[i].material_id;
            const uint32_t blend = mat < materials.size() && materials[mat].alpha_mode == 1u ? 2u : 0u;
            return (prim_alive[i] ? 1u : 0u) | blend | (mat << 2);
        };
        if (info_full_ || tri_info_.size() != slots) {
            tri_info_.resize(slots);
            tri_info_live_ = 0;
            for (size_t i = 0; i < slots; i++) { tri_info_[i] = word(i); tri_info_live_ += tri_info_[i] & 1u; }
            live_prims_ = tri_info_live_;
        } else {   // only the slots an instance took or gave back since the last listing (208 k slots: the full listing was 0.25-0.35 ms of a 0.45-ms tick)
            for (size_t i = info_dirty_lo_; i < std::min(info_dirty_hi_, slots); i++) { const uint32_t w = word(i); tri_info_live_ += (w & 1u) - (tri_info_[i] & 1u); tri_info_[i] = w; }
        }
        info_full_ = false; info_dirty_lo_ = SIZE_MAX; info_dirty_hi_ = 0;
        tri_info_built_for_ = tri_info_serial_;
    }
    const uint32_t live = tri_info_live_;
    const auto t1 = now();
    const bool whole = !t.valid || t.tri_full || t.tri_geo.capacity < tri_geo.size() * sizeof(float4) || t.tri_bounds.capacity < tri_bounds.size() * sizeof(float4);
    if (whole) {
        if ((rc = t.tri_geo.upload(tri_geo.data(), tri_geo.size() * sizeof(float4), up, staging, pageable))) return rc;
        if ((rc = t.tri_bounds.upload(tri_bounds.data(), tri_bounds.size() * sizeof(float4), up, staging, pageable))) return rc;
    } else if (t.dirty_lo < t.dirty_hi) {
        if ((rc = t.tri_geo.upload_range(tri_geo.data(), 3 * t.dirty_lo * sizeof(float4), 3 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, pageable))) return rc;
        if ((rc = t.tri_bounds.upload_range(tri_bounds.data(), 2 * t.dirty_lo * sizeof(float4), 2 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, pageable))) return rc;
    }
    // attribute records of the same slots, BEFORE the device bake below (an instance the host baked a tick ago and the device moves now must end with the device's)
    if (whole || t.tri_attr.capacity < tri_attr.size() * sizeof(float4)) {
        if ((rc = t.tri_attr.upload(tri_attr.data(), tri_attr.size() * sizeof(float4), up, staging, pageable))) return rc;
    } else if (t.dirty_lo < t.dirty_hi) {
        if ((rc = t.tri_attr.upload_range(tri_attr.data(), 4 * t.dirty_lo * sizeof(float4), 4 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, pageable))) return rc;
    }
    if (t.tri_info_serial != tri_info_serial_ || t.tri_info.capacity < slots * sizeof(uint32_t)) {
        if ((rc = t.tri_info.upload(tri_info_.data(), slots * sizeof(uint32_t), up, staging, pageable))) return rc;
        t.tri_info_serial = tri_info_serial_;
    }
    // instances that only moved are baked HERE from the object-space meshes (StTuning::device_bake, k_bvh.hip k_bvh_bake): the host bakes nothing for them
    const bool refit_tree = device_tree_refit_now && t.device_built && t.lb_live == live && t.lb_serial == tri_info_serial_ && t.lb_slots == (uint32_t)slots && live >= 2u;
    t.device_built = true;   // (bake_on_device: no contract stream to patch on this copy)
    const auto t2 = now();
    if ((rc = bake_on_device(t, up, pageable))) return rc;
    const auto t3 = now();
    const uint32_t pow2 = lbvh_pow2(live);
    const size_t temp = lbvh_sort_temp_bytes((uint32_t)slots);
    if ((rc = reserve_device_builder(t, slots, live))) return rc;
    LbvhArgs a{};
    a.tri_geo = static_cast<const float4*>(t.tri_geo.ptr); a.tri_bounds = static_cast<const float4*>(t.tri_bounds.ptr); a.tri_info = static_cast<const uint32_t*>(t.tri_info.ptr);
    a.slots = (uint32_t)slots; a.live = live; a.links16 = live < 32768u ? 1u : 0u;
    { static const char* v = getenv("ST_LBVH_CELL_ASPECT"); a.cell_aspect = v ? (float)atof(v) : kLbvhCellAspect; }
    a.nodes = static_cast<float4*>(t.bvh_wide.ptr); a.leaves = a.nodes + 4u * (size_t)(live - 1u);
    a.keys_in = static_cast<unsigned long long*>(t.lb_keys_a.ptr); a.keys_out = static_cast<unsigned long long*>(t.lb_keys_b.ptr);
    a.sort_temp = t.lb_temp.ptr; a.sort_temp_bytes = temp;
    a.seg = static_cast<float4*>(t.lb_seg.ptr); a.children = static_cast<uint2*>(t.lb_children.ptr); a.node_box = static_cast<float4*>(t.lb_node_box.ptr);
    a.bounds = static_cast<int*>(t.lb_small.ptr);
    // moves only, on a copy whose last build saw these very slots: the tree keeps its shape, every box follows (k_lbvh.hip lbvh_refit: 5 launches against 22)
    if (refit_tree) {
        if (lbvh_refit(a, up) != 0) return fail(ST_ERR_HIP, "the device BVH refit failed to launch");
        if (timing) fprintf(stderr, "[st_tick] device tree refit: slot words %.3f ms, uploads %.3f, bake launches %.3f, refit launches %.3f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
        return ST_OK;
    }
    if (lbvh_build(a, up) != 0) return fail(ST_ERR_HIP, "the device BVH build failed to launch");
    if (timing) fprintf(stderr, "[st_tick] device tree build: slot words %.3f ms, uploads %.3f, bake launches %.3f, build launches %.3f\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, now()));
    t.device_built = true; t.lb_live = live; t.lb_serial = tri_info_serial_; t.lb_slots = (uint32_t)slots;
    t.wide_nodes = live - 1u; t.wide_leaves = live; t.wide_root = 0u; t.wide_links16 = a.links16; t.wide_for_entries = 0u; t.compact_entries = 0u;
    return ST_OK;
}

// ---- tick (lib.rs:301-395)
int Engine::tick(hipStream_t stream) {
    bool scene_changed = false;
    materials_changed_this_tick = materials_dirty || atlas_dirty;   // a Blend flag may have changed: the tree's topology signature has to be looked at
    if (materials_dirty || atlas_dirty) { materials_dirty = false; rebuild_gpu_materials(); scene_changed = true; }
    const bool timing = tuning.tick_timing;
    auto now = [] { return std::chrono::steady_clock::now(); };
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
    const auto t0 = now();
    const bool instances_changed = refresh_instances();
    if (instances_changed) {
        for (const auto& inst : instances) {
            float4* x = instance_xforms.data() + 8u * inst.xslot;
            const Affine* src[2] = {&inst.xform_inv, &inst.prev_xform};
            for (int k = 0; k < 2; k++) { x[4 * k] = f4(src[k]->x, 0.0f); x[4 * k + 1] = f4(src[k]->y, 0.0f); x[4 * k + 2] = f4(src[k]->z, 0.0f); x[4 * k + 3] = f4(src[k]->t, 0.0f); }
        }
    }
    // ST_BVH_BUILD_DEVICE: while nothing observes the contract stream the changed scene's tree is built on the device (below, per device copy)
    // and the host's tree falls behind; the first tick that finds an observer brings it up to date like any rebuild.
    if (materials_changed_this_tick) info_full_ = true;   // a Blend flag may have changed under any slot
    if ((instances_changed && !moved_on_device) || materials_changed_this_tick) tri_info_serial_++;   // slots, liveness, materials or Blend flags may have changed
    bool build_on_device_now = device_build_possible();
    device_tree_refit_now = false;
    if (instances_changed && build_on_device_now) {
        // instances only moved (refresh_instances left them to the device's bake): the device-built tree is refitted, not rebuilt — at most
        // kDeviceRefitsPerBuild times in a row, then a rebuild restores the tree's quality (it costs 0.3 ms more at 208 k triangles)
        static const bool no_refit = getenv("ST_NO_DEVICE_TREE_REFIT") != nullptr;
        device_tree_refit_now = moved_on_device && !materials_changed_this_tick && !no_refit && device_builds > 0 && device_refits_since_build < kDeviceRefitsPerBuild;
        if (device_tree_refit_now) { device_tree_refits++; device_refits_since_build++; } else { device_builds++; device_refits_since_build = 0; }
        host_tree_stale = true; scene_changed = true;
        if (timing) fprintf(stderr, "[st_tick] bake %.2f ms, tree: on the device\n", ms(t0, now()));
    } else if (instances_changed || (host_tree_stale && !build_on_device_now)) {
        const auto t1 = now();
        std::vector<uint8_t> blend(materials.size());
        for (size_t i = 0; i < materials.size(); i++) blend[i] = materials[i].alpha_mode == 1u;
        const bool refitting = bvh_refresh_mode == ST_BVH_REFIT || bvh_refresh_mode == ST_BVH_REFIT_DEVICE;
        // (moved_on_device: refresh_instances saw nothing but transforms change — slots, materials and Blend flags are what the last build saw)
        const uint64_t signature = moved_on_device ? topology_signature : (refitting ? topology_of(blend) : 0);
        if (refitting && have_topology && !host_tree_stale && signature == topology_signature) {
            // ST_BVH_REFIT_DEVICE: the boxes are recomputed on the device from the moved triangles' bounds (k_bvh.hip); the host's
            // copy of the stream is brought up to date only when something reads it
            if (device_refit_possible()) host_stream_stale = true; else refit_stream();
            refits++;
            if (timing) fprintf(stderr, "[st_tick] bake %.2f ms, refit %.2f ms (%zu internal nodes)\n", ms(t0, t1), ms(t1, now()), internal_positions.size());
        } else {
            rebuild_host_tree(timing);
            if (refitting) { index_stream(); topology_signature = signature; have_topology = true; }
            if (!scene_uploaded && bvh_refresh_mode == ST_BVH_AUTO) {
                // the first tree of this scene: the host's, unless it hangs long leaf runs on large faces — then the device builder's, from this very tick on
                auto_first_on_device = host_leaf_run_weight > kAutoLeafRunLimit;
                if (auto_first_on_device && device_build_possible()) { build_on_device_now = true; device_builds++; device_refits_since_build = 0; }
            }
        }
        scene_changed = true;
    } else if (scene_uploaded && sets[live].device_built && !build_on_device_now) {
        // Nothing changed, the host's tree is current (a debug read rebuilt it) — but the live device copy still holds a device-built tree and no contract
        // stream, and an observer has appeared (heatmap camera, exact arithmetic, byte counting): the host's streams are uploaded now, so that the render
        // after THIS tick finds them (include/strolle_hip.h ST_BVH_BUILD_DEVICE: "a heatmap camera created later renders after the next st_tick").
        scene_changed = true;
    }
    light_count = next_light_id;
    {   // World::sun_dir (world.rs:18-24)
        float sa, ca, sz, cz;
        sincos_(sun_altitude, &sa, &ca); sincos_(sun_azimuth, &sz, &cz);
        sun_dir_ = v3(ca * sz, sa, -ca * cz);
    }
    if (sun_dirty) {
        sun_dirty = false;
        V3 color = sun_transmittance(v3(0.0f, 6.360f + 0.0002f, 0.0f), sun_dir_);
        color = color * 20.0f * 5.0f;
        GpuLight sun{};
        const V3 pos = sun_dir_ * 1000.0f;
        sun.d0 = f4(pos, 25.0f); sun.d1 = f4(color, INFINITY); sun.d2 = make_float4(b2f(1u), 0, 0, 0);
        overwrite_light(0, -1, sun);
    }
    snapshot_lights();
    if (has_device) {
        ST_HIP(hipSetDevice(device));
        if (walk_flags_host && (walk_flags_host[0] | walk_flags_host[1]) != 0u) note_walk_overflow();   // a wide walk of an earlier frame dropped a push
        // Uploads of an earlier tick that no render has waited for yet stay pending until their event has completed: a
        // tick that uploads nothing must not make a later render on another stream forget them.
        if (tick_work_in_flight && hipEventQuery(ev_tick) == hipSuccess) tick_work_in_flight = false;
        if (copy_in_flight && hipEventQuery(ev_copy) == hipSuccess) copy_in_flight = false;
        (void)hipGetLastError();  // hipErrorNotReady from the queries is not an error
        bool copied_now = false;  // this tick queued copies on copy_stream
        bool pageable = false;  // some copy of this tick reads pageable host memory (or writes it): join the stream before returning
        staging.begin_tick();
        bool pageable_copy = false;
        if (scene_changed || !scene_uploaded) {
            int rc;
            // which copy, on which stream: the first upload and ST_NO_DOUBLE_BUFFER=1 write the live copy in place on the
            // caller's stream (behind the frames queued there); every later change goes to the other copy on copy_stream
            int target = live; hipStream_t up = stream; bool* flag = &pageable; bool other_copy = false;
            if (tuning.double_buffer && scene_uploaded && !mixed_render_streams) {
                if (!copy_stream) { ST_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); ST_HIP(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming)); }
                if (!alternating) {  // frames enqueued so far read the live copy without marking their end: mark it now, behind them
                    alternating = true;
                    SceneSet& l = sets[live];
                    if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
                    ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
                }
                target = live ^ 1; up = copy_stream; flag = &pageable_copy; other_copy = true;
                if (sets[target].busy) { ST_HIP(hipStreamWaitEvent(copy_stream, sets[target].free_ev, 0)); sets[target].busy = false; }
            } else if (mixed_render_streams) ST_HIP(hipDeviceSynchronize());  // cameras render on several streams: no single event ends their reads
            SceneSet& t = sets[target];
            bool attr_sent = false;
            bool attr_done = false;
            if (!scene_uploaded && (bvh_refresh_mode == ST_BVH_AUTO || bvh_refresh_mode == ST_BVH_BUILD_DEVICE) && arithmetic == ST_ARITH_FAST &&
                tuning.wide_bvh && live_prims_ > kLdsSceneTexels / 4u && prims.size() < (1u << 23)) {
                // this scene's changes (and, for a large scene, its first tree: device_build_possible) go to the device builder: its arrays are allocated NOW, while the
                // scene loads, for both copies — BEFORE the wide stream of this copy is written below (its allocation is one of them: growing it afterwards would throw the stream away: round 6's
                // first version did, and rendered an empty world). A failure here is not an error: the build allocates what it finds missing.
                {   // ... and the host's triangle arrays get the same headroom now, so that the first spawn does not reallocate and copy them (60 MB at 208 k triangles: 7 ms)
                    const size_t want = triangles.size() + triangles.size() / 8u;
                    triangles.reserve(want); prims.reserve(want); prim_alive.reserve(want); tri_geo.reserve(3 * want); tri_attr.reserve(4 * want); tri_bounds.reserve(2 * want);
                }
                bool reserved = true;
                for (SceneSet& c : sets) {
                    if (reserve_device_builder(c, prims.size() + prims.size() / 8u, (uint32_t)std::min<size_t>(live_prims_ + live_prims_ / 8u, prims.size() + prims.size() / 8u)) != ST_OK) { (void)hipGetLastError(); reserved = false; break; }
                }
                if (reserved && !build_on_device_now && sets[0].lb_small.ptr) lbvh_warm(static_cast<int*>(sets[0].lb_small.ptr), up);   // ... and the builder's code object is on the device before the first spawn
            }
            if (build_on_device_now) {
                // (instances the device moved stay stale on the host: this copy's pending list re-bakes them on the device even after a whole upload)
                if ((rc = build_on_device(t, up, flag))) return rc;
                attr_done = true;
                host_tree_stale = true;   // whatever made this tick upload (a material edit, too): the live copy's tree is no longer the host's
                t.tree_version = 0;       // ... and this copy's contract stream and refit arrays belong to no host tree any more
            } else t.device_built = false;
            const bool device_path = !build_on_device_now && device_refit_possible() && t.valid && !t.tri_full && t.tree_version == tree_version && t.tri_geo.capacity >= tri_geo.size() * sizeof(float4);
            // a copy that cannot be brought up to date in place is sent whole, from the host's arrays: instances the device moved must be in them
            if (!device_path && !build_on_device_now && any_host_stale()) bake_stale_on_host();
            if (device_path) {
                // This copy holds the current tree; only boxes and moved triangles are behind. Send the records and bounds of the
                // triangle slots baked since it was written and let the device patch its leaf entries and refit its boxes.
                if (t.dirty_lo < t.dirty_hi) {
                    if ((rc = t.tri_geo.upload_range(tri_geo.data(), 3 * t.dirty_lo * sizeof(float4), 3 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
                    if ((rc = t.tri_bounds.upload_range(tri_bounds.data(), 2 * t.dirty_lo * sizeof(float4), 2 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
                    L.launch_bvh_patch_leaves(static_cast<float4*>(t.bvh.ptr), static_cast<const float4*>(t.tri_geo.ptr), static_cast<const uint32_t*>(t.entry_of_tri.ptr), (uint32_t)t.dirty_lo, (uint32_t)t.dirty_hi, up);
                    // (their attribute records too, BEFORE the device bake below: an instance the host baked a tick ago and the device moves now must end with the device's)
                    if (t.tri_attr.capacity >= tri_attr.size() * sizeof(float4)) {
                        if ((rc = t.tri_attr.upload_range(tri_attr.data(), 4 * t.dirty_lo * sizeof(float4), 4 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
                        attr_sent = true;
                    }
                }
                // instances that only moved are baked HERE, from the object-space meshes (StTuning::device_bake; k_bvh.hip k_bvh_bake): this copy's
                // hit-test records (+ leaf entries), bounds and attribute records of every instance it has not followed yet
                if ((rc = bake_on_device(t, up, flag))) return rc;
                for (const auto& level : refit_levels_)   // this copy holds the current tree, so the engine's work list is its own
                    L.launch_bvh_refit(static_cast<float4*>(t.bvh.ptr), static_cast<const float4*>(t.tri_bounds.ptr), static_cast<const uint32_t*>(t.parent.ptr), static_cast<const uint32_t*>(t.refit_local.ptr),
                                       static_cast<const uint32_t*>(t.refit_items.ptr), static_cast<const uint32_t*>(t.refit_batch_off.ptr), level.first, level.second, up);
                device_refits++;
            } else if (!build_on_device_now) {
                if (host_stream_stale) { refit_stream(); host_stream_stale = false; }
                expand_stream();
                // traversal pointers are 32-bit BYTE offsets into the device stream (64 B per entry) and stack slots hold entry numbers
                if ((size_t)device_bvh_len * sizeof(float4) > 0xffffffffull) return fail(ST_ERR_INVALID_ARGUMENT, "the BVH stream exceeds 4 GiB (2^26 entries): traversal pointers are 32-bit byte offsets");
                if ((rc = t.bvh.upload(bvh_upload_.data(), bvh_upload_.size() * sizeof(float4), up, staging, flag))) return rc;
                if (device_refit_possible()) {  // what the device refit of later ticks needs beside the stream
                    index_device_tree();
                    if ((rc = t.tri_geo.upload(tri_geo.data(), tri_geo.size() * sizeof(float4), up, staging, flag))) return rc;
                    if ((rc = t.tri_bounds.upload(tri_bounds.data(), tri_bounds.size() * sizeof(float4), up, staging, flag))) return rc;
                    if ((rc = t.entry_of_tri.upload(entry_of_tri_.data(), entry_of_tri_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                    if ((rc = t.parent.upload(parent_.data(), parent_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                    if ((rc = t.refit_local.upload(refit_local_.data(), refit_local_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                    if (!refit_items_.empty() && (rc = t.refit_items.upload(refit_items_.data(), refit_items_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                    if ((rc = t.refit_batch_off.upload(refit_batch_off_.data(), refit_batch_off_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                    t.tree_version = tree_version;
                }
            }
            if (!build_on_device_now) {
                if ((rc = refresh_compact_stream(t, up))) return rc;   // the shadow rays' compact form follows every change of the contract stream
                if ((rc = refresh_wide_stream(t, up, !device_path, flag))) return rc;   // and so does the wide form (its topology only when the tree itself was sent)
            }
            // attribute records: whole the first time or after they grew, otherwise only the slots baked since this copy was written
            const bool partial = t.valid && !t.tri_full && t.tri_attr.capacity >= tri_attr.size() * sizeof(float4);
            if (attr_done) {
            } else if (!partial) {
                if ((rc = t.tri_attr.upload(tri_attr.data(), tri_attr.size() * sizeof(float4), up, staging, flag))) return rc;
            } else if (t.dirty_lo < t.dirty_hi && !attr_sent) {
                if ((rc = t.tri_attr.upload_range(tri_attr.data(), 4 * t.dirty_lo * sizeof(float4), 4 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
            }
            t.dirty_lo = SIZE_MAX; t.dirty_hi = 0; t.tri_full = false; t.valid = true;
            if ((rc = t.xforms.upload(instance_xforms.data(), instance_xforms.size() * sizeof(float4), up, staging, flag))) return rc;
            if ((rc = t.materials.upload(gpu_materials.data(), gpu_materials.size() * sizeof(GpuMaterial), up, staging, flag))) return rc;
            if ((rc = t.base_packed.upload(material_base_packed.data(), material_base_packed.size() * sizeof(uint32_t), up, staging, flag))) return rc;
            if (other_copy) copied_now = true;
            live = target; live_bvh_texels = build_on_device_now ? 0u : device_bvh_len;
            scene_uploaded = true;
            scene_changed = !other_copy;  // in-place uploads count as work on the caller's stream below
        }
        bool misc_uploaded = atlas_dirty || blue_noise_dirty, uploaded_device_images = false;
        if (atlas_dirty) { int rc = d_atlas.upload(atlas.data(), atlas.size(), stream, staging, &pageable); if (rc) return rc; }
        for (auto& kv : device_images) {
            DeviceImage& di = kv.second;
            if (!di.pending && !di.dynamic) continue;
            const ImageRec& r = images.at(kv.first);
            uint8_t* dst = static_cast<uint8_t*>(d_atlas.ptr) + ((size_t)r.y * atlas_w + r.x) * 4;
            ST_HIP(hipMemcpy2DAsync(dst, (size_t)atlas_w * 4, di.pixels, di.pitch, (size_t)r.w * 4, r.h, hipMemcpyDeviceToDevice, stream));
            if (!di.dynamic) {  // keep the host copy complete: it is what a later full upload sends
                ST_HIP(hipMemcpy2DAsync(&atlas[((size_t)r.y * atlas_w + r.x) * 4], (size_t)atlas_w * 4, di.pixels, di.pitch, (size_t)r.w * 4, r.h, hipMemcpyDeviceToHost, stream));
                misc_uploaded = true; pageable = true;  // joins the stream below before the host copy is read again
            }
            di.pending = false;
            uploaded_device_images = true;
        }
        if (blue_noise_dirty) { int rc = d_blue_noise.upload(blue_noise.data(), blue_noise.size(), stream, staging, &pageable); if (rc) return rc; blue_noise_dirty = false; }
        bool uploaded = scene_changed || misc_uploaded || uploaded_device_images;
        // lights change rarely; skipping the identical re-upload also skips the stream sync below, so the host can
        // run a frame ahead of the GPU (the reference re-uploads only dirty buffers too: mapped_storage_buffer.rs:103-121)
        if (gpu_lights.size() != uploaded_lights.size() || memcmp(gpu_lights.data(), uploaded_lights.data(), gpu_lights.size() * sizeof(GpuLight)) != 0) {
            int target = live_lights; hipStream_t up = stream; bool* flag = &pageable; bool other_copy = false;
            if (tuning.double_buffer && lights_uploaded && !mixed_render_streams) {
                if (!copy_stream) { ST_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); ST_HIP(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming)); }
                if (!lights_alternating) {  // as for the scene: the frames queued so far end here
                    lights_alternating = true;
                    LightSet& l = light_sets[live_lights];
                    if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
                    ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
                }
                target = live_lights ^ 1; up = copy_stream; flag = &pageable_copy; other_copy = true;
                if (light_sets[target].busy) { ST_HIP(hipStreamWaitEvent(copy_stream, light_sets[target].free_ev, 0)); light_sets[target].busy = false; }
            } else if (mixed_render_streams) ST_HIP(hipDeviceSynchronize());
            int rc = light_sets[target].buf.upload(gpu_lights.data(), gpu_lights.size() * sizeof(GpuLight), up, staging, flag);
            if (rc) return rc;
            live_lights = target; lights_uploaded = true;
            uploaded_lights = gpu_lights;
            if (other_copy) copied_now = true; else uploaded = true;
        }
        if (copied_now) {
            copy_in_flight = true;
            ST_HIP(hipEventRecord(ev_copy, copy_stream));
            ST_HIP(hipStreamWaitEvent(stream, ev_copy, 0));  // the caller's stream: the next frame's kernels (and the staging slot's event) come after the copies
        }
        if (int rc = staging.end_tick(stream)) return rc;
        // What was uploaded went through page-locked staging, so the caller may change the scene again at once; the next
        // frame's side stream is ordered behind these copies by an event (render). Only copies that touch pageable
        // host memory directly (staging full or disabled) make the tick wait for the stream.
        if (uploaded) {
            if (!ev_tick) ST_HIP(hipEventCreateWithFlags(&ev_tick, hipEventDisableTiming));
            ST_HIP(hipEventRecord(ev_tick, stream));
            tick_work_in_flight = true;
        }
        if (pageable_copy) ST_HIP(hipStreamSynchronize(copy_stream));
        if ((uploaded && pageable) || sync_every_tick) ST_HIP(hipStreamSynchronize(stream));
    }
    atlas_dirty = false;
    for (auto& kv : cameras) kv.second->frame = frame;  // CameraController::flush
    frame += 1;
    if (walk_overflow_unreported) {   // likewise: the tick did everything, later frames walk with the deeper stack
        walk_overflow_unreported = false;
        const std::string msg = "a traversal of the wide BVH stream found its stack full and DROPPED a push: frames rendered so far may have missed geometry behind the dropped subtrees. "
                                "Later frames walk with " + std::to_string(wide_stack_entries_now()) + " pending entries per ray" + (packets_overflowed ? " and primary rays with the per-lane walk instead of the packet walk" : "") +
                                " (st_debug_walk_overflow); StTuning::allow_deep_bvh = 1 turns this status into a warning";
        if (!tuning.allow_deep_bvh) return fail(ST_ERR_BVH_TOO_DEEP, msg);
        fprintf(stderr, "[strolle-hip] warning: %s\n", msg.c_str());
    }
    if (bvh_too_deep_unreported) {   // the tick did everything; the status says what the uploaded tree can cost (once per build)
        bvh_too_deep_unreported = false;
        if (!tuning.allow_deep_bvh)
            return fail(ST_ERR_BVH_TOO_DEEP, "the BVH is " + std::to_string(bvh_stack_need) + " internal nodes deep, the kernels' traversal stack holds " + std::to_string(stack_entries) +
                                             " pending entries (strolle-gpu/src/lib.rs:76): pushes beyond it are dropped and geometry behind them can be missed. The scene was uploaded and renders; StTuning::allow_deep_bvh = 1 accepts this");
    }
    return ST_OK;
}

// The wide stream (k_bvh.hip k_bvh_wide) of device copy `t`: topology from the host when the tree was (re)sent, boxes and leaf records from that
// copy's contract stream as it is on the device right now.
int Engine::refresh_wide_stream(SceneSet& t, hipStream_t up, bool topology_changed, bool* pageable) {
    if (!tuning.wide_bvh || !tuning.compact_bvh || device_bvh_len <= kLdsSceneTexels) { t.wide_nodes = t.wide_leaves = 0u; t.wide_for_entries = 0u; return ST_OK; }
    int rc;
    if (topology_changed || t.wide_for_entries != device_bvh_len / 4u) {
        if (!topology_changed && wide_built_for_ != tree_version) return ST_OK;   // (cannot happen: a copy on the device path holds this tree's topology)
        if (wide_built_for_ != tree_version || wide_topo_.empty()) { build_wide_topology(); wide_built_for_ = tree_version; }
        const uint32_t nodes = (uint32_t)(wide_topo_.size() / 8u), leaves = (uint32_t)wide_leaf_entry_.size();
        t.wide_nodes = t.wide_leaves = 0u; t.wide_for_entries = 0u;
        if (nodes >= (1u << 23) || leaves >= (1u << 23)) return ST_OK;   // v_mul_u32_u24 addressing and the << 5 of a node link: larger trees keep the binary streams
        if (nodes && (rc = t.wide_topo.upload(wide_topo_.data(), wide_topo_.size() * sizeof(uint32_t), up, staging, pageable))) return rc;
        if (leaves && (rc = t.wide_leaf_entry.upload(wide_leaf_entry_.data(), wide_leaf_entry_.size() * sizeof(uint32_t), up, staging, pageable))) return rc;
        // one allocation: nodes (64 B each), then the leaf records (48 B each) + one texel of slack (a node step's fourth texel is never read for a record)
        const size_t need = (size_t)nodes * 64u + (size_t)leaves * 48u + 64u;
        if (need > 0xfffffff0ull) return ST_OK;   // 32-bit byte offsets
        if (need > t.bvh_wide.capacity) {
            if (t.bvh_wide.ptr) ST_HIP(hipFree(t.bvh_wide.ptr));
            t.bvh_wide.ptr = nullptr; t.bvh_wide.capacity = 0;
            ST_HIP(hipMalloc(&t.bvh_wide.ptr, need + need / 2)); t.bvh_wide.capacity = need + need / 2;
        }
        t.wide_nodes = nodes; t.wide_leaves = leaves; t.wide_root = wide_root_; t.wide_for_entries = device_bvh_len / 4u;
        t.wide_links16 = (nodes < 32768u && leaves < 32768u) ? 1u : 0u;
        t.wide_topology_serial = wide_serial_;
    }
    if (!t.wide_for_entries) return ST_OK;
    launchers_exact().launch_bvh_wide(static_cast<const float4*>(t.bvh.ptr), static_cast<const uint32_t*>(t.wide_topo.ptr), t.wide_nodes, static_cast<const uint32_t*>(t.wide_leaf_entry.ptr), t.wide_leaves,
                                      t.wide_links16, static_cast<float4*>(t.bvh_wide.ptr), static_cast<float4*>(t.bvh_wide.ptr) + 4u * (size_t)t.wide_nodes, up);
    return ST_OK;
}

// The compact stream (k_bvh.hip k_bvh_compact) of device copy `t`, from that copy's contract stream as it is on the device right now.
int Engine::refresh_compact_stream(SceneSet& t, hipStream_t up) {
    t.compact_entries = 0;
    if (!tuning.compact_bvh || device_bvh_len <= kLdsSceneTexels) return ST_OK;   // tiny scenes live in LDS as they are
    const uint32_t entries = device_bvh_len / 4u;
    // the compact walk addresses entries with v_mul_u32_u24 (st_device.h any_hit_compact): 24 bits of entry number. A larger stream (8 M+
    // triangles) keeps compact_entries = 0 and its rays walk the contract stream with any_hit_fast / traverse
    if (entries >= (1u << 24)) return ST_OK;
    const size_t bytes = (size_t)entries * 48u;
    if (bytes > t.bvh_compact.capacity) {
        if (t.bvh_compact.ptr) ST_HIP(hipFree(t.bvh_compact.ptr));
        t.bvh_compact.ptr = nullptr; t.bvh_compact.capacity = 0;
        ST_HIP(hipMalloc(&t.bvh_compact.ptr, bytes + bytes / 2)); t.bvh_compact.capacity = bytes + bytes / 2;
    }
    launchers_exact().launch_bvh_compact(static_cast<const float4*>(t.bvh.ptr), entries, static_cast<float4*>(t.bvh_compact.ptr), up);
    t.compact_entries = entries;
    return ST_OK;
}

}  // namespace st
