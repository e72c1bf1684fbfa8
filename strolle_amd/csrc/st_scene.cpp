// st_scene.cpp — host engine of libstrolle_hip.so: scene stores: materials (materials.rs, material.rs), lights (lights.rs, light.rs), instances -> world-space triangles (instances.rs, mesh_triangle.rs, triangle.rs). See st_engine.h.
#include "st_engine.h"

namespace st {

// ---- materials (materials.rs:33-96, material.rs:29-50)
float4 Engine::image_rect(uint64_t h) const {
    if (!h) return make_float4(0, 0, 0, 0);
    auto it = images.find(h);
    if (it == images.end() || atlas_w == 0) return make_float4(0, 0, 0, 0);
    const ImageRec& r = it->second;
    return make_float4((float)r.x / (float)atlas_w, (float)r.y / (float)atlas_h, (float)r.w / (float)atlas_w, (float)r.h / (float)atlas_h);
}

void Engine::rebuild_gpu_materials() {
    gpu_materials.resize(materials.size()); material_base_packed.resize(materials.size());
    for (size_t i = 0; i < materials.size(); i++) {
        const StMaterial& m = materials[i]; GpuMaterial& g = gpu_materials[i];
        g.base_color = make_float4(m.base_color[0], m.base_color[1], m.base_color[2], m.base_color[3]);
        g.base_color_texture = image_rect(m.base_color_texture);
        g.emissive = make_float4(m.emissive[0], m.emissive[1], m.emissive[2], m.emissive[3]);
        g.emissive_texture = image_rect(m.emissive_texture);
        g.roughness = pow2_(m.perceptual_roughness);
        g.metallic = m.metallic; g.reflectance = m.reflectance; g.ior = m.ior;
        g.metallic_roughness_texture = image_rect(m.metallic_roughness_texture);
        g.normal_map_texture = image_rect(m.normal_map_texture);
        material_base_packed[i] = gbuffer_pack_base_color(g.base_color);  // st_math.h routines are bit-identical on host and device
    }
}

void Engine::overwrite_light(uint32_t slot, int64_t key, GpuLight g) {
    const GpuLight old = light_buffer[slot];
    g.prev_d0 = old.d0; g.prev_d1 = old.d1; g.prev_d2 = old.d2;
    note(lights_updated, key);
    light_buffer[slot] = g;
}

void Engine::insert_light(uint64_t id, const StLight& l) {
    GpuLight g{};
    g.d0 = make_float4(l.position[0], l.position[1], l.position[2], l.radius);
    g.d1 = make_float4(l.color[0], l.color[1], l.color[2], l.range);
    if (l.kind == ST_LIGHT_POINT) g.d2 = make_float4(b2f(1u), 0, 0, 0);
    else {
        V3 n = v3(l.direction[0], l.direction[1], l.direction[2]);  // Normal::encode (normal.rs:9-24)
        n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
        V2 e = n.z >= 0.0f ? v2(n.x, n.y) : v2(copysignf(1.0f - fabsf(n.y), n.x), copysignf(1.0f - fabsf(n.x), n.y));
        e = e * 0.5f + 0.5f;
        g.d2 = make_float4(b2f(2u), e.x, e.y, l.angle);
    }
    const int64_t key = (int64_t)id;
    auto it = light_slot.find(key);
    if (it != light_slot.end()) { overwrite_light(it->second, key, g); return; }
    if (next_light_id < light_buffer.size()) { light_buffer[next_light_id] = g; light_slot[key] = next_light_id; }
    else { light_slot[key] = (uint32_t)light_buffer.size(); light_buffer.push_back(g); }
    note(lights_created, key);
    next_light_id += 1;
}

void Engine::remove_light(uint64_t id) {
    const int64_t key = (int64_t)id;
    auto it = light_slot.find(key);
    if (it == light_slot.end()) return;  // silent no-op like the reference
    const uint32_t slot = it->second;
    light_slot.erase(it);
    light_buffer.erase(light_buffer.begin() + slot);
    light_buffer.push_back(GpuLight{});
    lights_created.erase(std::remove(lights_created.begin(), lights_created.end(), key), lights_created.end());
    lights_updated.erase(std::remove(lights_updated.begin(), lights_updated.end(), key), lights_updated.end());
    lights_remapped.erase(key);
    if (std::find(lights_killed.begin(), lights_killed.end(), slot) == lights_killed.end()) lights_killed.push_back(slot);
    next_light_id -= 1;
    for (auto& kv : light_slot)
        if (kv.second > slot) { if (!lights_remapped.count(kv.first)) lights_remapped[kv.first] = kv.second; kv.second -= 1; }
}

void Engine::snapshot_lights() {  // lights.rs:128-154: what the device sees this frame, then commit prev_* for the next one
    for (uint32_t s : lights_killed) light_buffer[s].d3.x = b2f(0xcafebabeu);
    for (auto& kv : lights_remapped) light_buffer[kv.second].d3.x = b2f(light_slot[kv.first] + 1u);
    gpu_lights = light_buffer;
    auto commit = [&](int64_t k) { GpuLight& l = light_buffer[light_slot[k]]; l.prev_d0 = l.d0; l.prev_d1 = l.d1; l.prev_d2 = l.d2; };
    for (int64_t k : lights_created) commit(k);
    for (int64_t k : lights_updated) commit(k);
    for (uint32_t s : lights_killed) light_buffer[s].d3.x = b2f(0u);
    for (auto& kv : lights_remapped) light_buffer[kv.second].d3.x = b2f(0u);
    lights_created.clear(); lights_updated.clear(); lights_remapped.clear(); lights_killed.clear();
}

// ---- instances -> world-space triangles (instances.rs:69-139, mesh_triangle.rs:47-86, triangle.rs:16-37)
void Engine::drop_instance_triangles(uint64_t id) {
    auto it = instance_triangles.find(id);
    if (it == instance_triangles.end()) return;
    triangle_free.give(it->second.first, it->second.second);
    for (size_t i = it->second.first; i < it->second.second; i++) { live_prims_ -= prim_alive[i]; prim_alive[i] = 0; }
    mark_info_dirty(it->second.first, it->second.second);
    instance_triangles.erase(it);
}

void Engine::bake(const StMeshTriangle& t, const InstanceRec& inst, uint32_t material, size_t slot) {
    // normals use transpose(inverse(xform)) (Mat4::transform_vector3 order); tangents follow the forward matrix
    const Affine& inv = inst.xform_inv;
    const V3 r0 = v3(inv.x.x, inv.y.x, inv.z.x), r1 = v3(inv.x.y, inv.y.y, inv.z.y), r2 = v3(inv.x.z, inv.y.z, inv.z.z);
    const float det = dot(inst.xform.z, cross(inst.xform.x, inst.xform.y));
    const float sign = (f2b(det) >> 31) ? -1.0f : 1.0f;
    V3 p[3], n[3]; float4 tg[3];
    for (int i = 0; i < 3; i++) {
        p[i] = affine_point(inst.xform, v3(t.positions[i][0], t.positions[i][1], t.positions[i][2]));
        const V3 nn = v3(t.normals[i][0], t.normals[i][1], t.normals[i][2]);
        // transpose(inverse): columns are the inverse's rows; the 4th row of the transposed matrix carries the
        // inverse translation in .w only, which transform_vector3 drops
        V3 acc = r0 * nn.x; acc = r1 * nn.y + acc; acc = r2 * nn.z + acc;
        n[i] = normalize(acc);
        const V3 tt = normalize(affine_vec(inst.xform, v3(t.tangents[i][0], t.tangents[i][1], t.tangents[i][2])));
        tg[i] = make_float4(tt.x, tt.y, tt.z, t.tangents[i][3] * sign);
    }
    HostTriangle h;
    h.d0 = f4(p[0], t.uvs[0][0]); h.d1 = f4(n[0], t.uvs[0][1]); h.d2 = tg[0];
    h.d3 = f4(p[1], t.uvs[1][0]); h.d4 = f4(n[1], t.uvs[1][1]); h.d5 = tg[1];
    h.d6 = f4(p[2], t.uvs[2][0]); h.d7 = f4(n[2], t.uvs[2][1]); h.d8 = tg[2];
    triangles[slot] = h;
    BuildPrim bp;
    bp.triangle_id = (uint32_t)slot; bp.material_id = material;
    bp.center = (((v3s(0.0f) + p[0]) + p[1]) + p[2]) / 3.0f;
    bp.bounds = Aabb(); bp.bounds.grow(p[0]); bp.bounds.grow(p[1]); bp.bounds.grow(p[2]);
    prims[slot] = bp; prim_alive[slot] = 1;
    tri_geo[3 * slot] = f4(p[0], 0.0f); tri_geo[3 * slot + 1] = f4(p[1] - p[0], 0.0f); tri_geo[3 * slot + 2] = f4(p[2] - p[0], 0.0f);
    tri_bounds[2 * slot] = f4(bp.bounds.lo, 0.0f); tri_bounds[2 * slot + 1] = f4(bp.bounds.hi, 0.0f);
    tri_attr[4 * slot] = f4(n[0], t.uvs[0][0]); tri_attr[4 * slot + 1] = f4(n[1], t.uvs[0][1]); tri_attr[4 * slot + 2] = f4(n[2], t.uvs[1][0]);
    tri_attr[4 * slot + 3] = make_float4(t.uvs[1][1], t.uvs[2][0], t.uvs[2][1], b2f(inst.xslot));
}

// Baking (instances.rs:100-139) writes disjoint slots and reads nothing it writes, so once every range is assigned — the arrays
// do not move any more — large refreshes are spread over the BVH builder's worker pool in chunks.
void Engine::bake_jobs_on_host(const std::vector<BakeJob>& jobs, size_t total) {
    const auto tb0 = std::chrono::steady_clock::now();
    constexpr size_t kChunk = 2048, kParallelFrom = 16384;
    unsigned threads = std::thread::hardware_concurrency();
    if (threads > 16u) threads = 16u;
    if (total < kParallelFrom || threads < 2u) {
        for (const BakeJob& j : jobs)
            for (size_t i = 0; i < j.count; i++) bake((*j.mesh)[i], *j.inst, j.material, j.first + i);
    } else {
        TaskPool pool(threads);
        for (const BakeJob& j : jobs)
            for (size_t at = 0; at < j.count; at += kChunk) {
                const size_t end = std::min(j.count, at + kChunk);
                pool.push([this, j, at, end] { for (size_t i = at; i < end; i++) bake((*j.mesh)[i], *j.inst, j.material, j.first + i); });
            }
        pool.finish();
    }
    if (tuning.tick_timing) fprintf(stderr, "[bake] %zu triangles in %zu jobs: %.2f ms\n", total, jobs.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
}

// Instances the device has moved (StTuning::device_bake) are baked on the host only when the host arrays are needed again: a rebuild
// (its primitives), a host refit or a debug read of the stream (tri_bounds), a full upload. Both device copies then receive those slots from
// the host like any other baked range, and their lists of pending device moves are void.
void Engine::bake_stale_on_host() {
    std::vector<BakeJob> jobs; size_t total = 0;
    for (auto& inst : instances) {
        if (!inst.host_stale) continue;
        inst.host_stale = false;
        auto mesh = meshes.find(inst.mesh);
        auto have = instance_triangles.find(inst.id);
        if (mesh == meshes.end() || have == instance_triangles.end() || have->second.second - have->second.first != mesh->second.size()) continue;  // the next refresh re-bakes it as dirty
        jobs.push_back({&mesh->second, &inst, inst.baked_material, have->second.first, mesh->second.size()});
        total += mesh->second.size();
        for (SceneSet& t : sets) { t.dirty_lo = std::min(t.dirty_lo, have->second.first); t.dirty_hi = std::max(t.dirty_hi, have->second.second); }
    }
    for (SceneSet& t : sets) t.pending_moves.clear();
    if (!jobs.empty()) bake_jobs_on_host(jobs, total);
}

bool Engine::refresh_instances() {
    moved_on_device = false;
    if (!instances_dirty) return false;
    instances_dirty = false;
    // Device bake: when every dirty instance only MOVED — same mesh (and version of it), same material, its slots already assigned —
    // under ST_BVH_REFIT_DEVICE with the tree's topology on record, the host bakes nothing: the device copies do (Engine::bake_on_device).
    const bool removed = instance_removed; instance_removed = false;
    // (ST_BVH_BUILD_DEVICE: the same, with the tree rebuilt on the device from the device-baked arrays instead of refitted)
    const bool build_mode = device_build_possible() && scene_uploaded;
    if (tuning.device_bake && ((device_refit_possible() && have_topology) || build_mode) && !materials_changed_this_tick && !removed) {
        bool only_moves = true; size_t moved = 0;
        for (const auto& inst : instances) {
            if (!inst.dirty) continue;
            moved++;
            auto mv = mesh_version.find(inst.mesh);
            auto mat = material_slot.find(inst.material);
            auto have = instance_triangles.find(inst.id);
            auto mesh = meshes.find(inst.mesh);
            if (!inst.baked || mv == mesh_version.end() || mat == material_slot.end() || have == instance_triangles.end() || mesh == meshes.end() ||
                inst.baked_mesh != inst.mesh || inst.baked_mesh_version != mv->second || inst.baked_material != mat->second ||
                have->second.second - have->second.first != mesh->second.size()) { only_moves = false; break; }
        }
        if (only_moves && moved) {
            for (auto& inst : instances) {
                if (!inst.dirty) continue;
                inst.dirty = false; inst.host_stale = true;
                for (SceneSet& t : sets) if (std::find(t.pending_moves.begin(), t.pending_moves.end(), inst.id) == t.pending_moves.end()) t.pending_moves.push_back(inst.id);
            }
            moved_on_device = true;
            return true;
        }
    }
    // the host bakes this refresh: instances the device moved earlier and that are not dirty now must catch up first (a rebuild reads every primitive)
    if (any_host_stale()) {
        // re-baked below anyway — but only those whose bake job WILL be queued: an instance whose mesh or material is missing is retried at a
        // later tick, and until then its host arrays must still count as stale (debug reads catch up through bake_stale_on_host)
        for (auto& inst : instances)
            if (inst.host_stale && inst.dirty && meshes.count(inst.mesh) && material_slot.count(inst.material)) inst.host_stale = false;
        bake_stale_on_host();
    }
    std::vector<BakeJob> jobs; size_t total = 0;
    {   // one reallocation at most for everything this refresh appends (a scene load appends every instance)
        size_t fresh = 0;
        for (const auto& inst : instances) {
            if (!inst.dirty || instance_triangles.count(inst.id)) continue;
            auto mesh = meshes.find(inst.mesh);
            if (mesh != meshes.end()) fresh += mesh->second.size();
        }
        if (fresh && triangles.size() + fresh > triangles.capacity()) {
            // (geometric: reserve(size + fresh) at EVERY spawn reallocated and copied all five arrays — 60 MB at 208 k triangles, 7 ms of a spawn tick
            // that otherwise costs 0.2 ms, profiles/r06_spawn_cost.txt — although the free list usually serves the new instance)
            const size_t want = std::max(triangles.size() + fresh, triangles.capacity() + triangles.capacity() / 2);
            triangles.reserve(want); prims.reserve(want); prim_alive.reserve(want); tri_geo.reserve(3 * want); tri_attr.reserve(4 * want); tri_bounds.reserve(2 * want);
        }
    }
    for (auto& inst : instances) {
        if (!inst.dirty) continue;
        inst.dirty = false;
        auto mesh = meshes.find(inst.mesh);
        auto mat = material_slot.find(inst.material);
        if (mesh == meshes.end() || mat == material_slot.end()) { inst.dirty = true; instances_dirty = true; continue; }  // retry next tick
        const size_t count = mesh->second.size();
        auto have = instance_triangles.find(inst.id);
        if (have != instance_triangles.end() && have->second.second - have->second.first != count) { drop_instance_triangles(inst.id); have = instance_triangles.end(); }
        size_t b, e;
        if (have != instance_triangles.end()) { b = have->second.first; e = have->second.second; }
        else if (!triangle_free.take(count, &b, &e)) {
            b = triangles.size(); e = b + count;
            triangles.resize(e); prims.resize(e); prim_alive.resize(e, 0); tri_geo.resize(3 * e); tri_attr.resize(4 * e); tri_bounds.resize(2 * e);
            for (SceneSet& t : sets) t.tri_full = true;
        }
        jobs.push_back({&mesh->second, &inst, mat->second, b, count});
        total += count;
        for (SceneSet& t : sets) { t.dirty_lo = std::min(t.dirty_lo, b); t.dirty_hi = std::max(t.dirty_hi, e); }  // slots each device copy still has to receive
        mark_info_dirty(b, e);
        if (have == instance_triangles.end()) live_prims_ += count;   // (the bake below sets prim_alive for the whole range)
        instance_triangles[inst.id] = {b, e};
        auto mv = mesh_version.find(inst.mesh);
        inst.baked = true; inst.baked_mesh = inst.mesh; inst.baked_mesh_version = mv == mesh_version.end() ? 0 : mv->second; inst.baked_material = mat->second; inst.host_stale = false;
    }
    bake_jobs_on_host(jobs, total);
    return true;
}

// The device half of a tick whose instances only moved: object-space meshes this copy's pending instances need (appended to the
// device mesh store once), one 128-B job per instance with its CURRENT transform, one launch of k_bvh_bake (which also patches the
// leaf entries). PCIe per tick: 128 B + 4 B per moved instance (and the instance-transform table st_tick sends anyway).
int Engine::bake_on_device(SceneSet& t, hipStream_t up, bool* pageable) {
    if (t.pending_moves.empty()) return ST_OK;
    std::unordered_map<uint64_t, const InstanceRec*> by_id;
    for (const auto& inst : instances) by_id[inst.id] = &inst;
    struct Job { float4 x, y, z, t, r0, r1, r2; uint32_t mesh_first, count, slot_first, xslot; };
    static_assert(sizeof(Job) == 128, "k_bvh.hip BakeJobDevice");
    std::vector<Job> jobs; std::vector<uint32_t> starts{0u};
    bool store_grew = false;
    for (uint64_t id : t.pending_moves) {
        auto it = by_id.find(id);
        if (it == by_id.end()) continue;   // removed since: that tick rebuilt the tree and voided the lists
        const InstanceRec& inst = *it->second;
        auto have = instance_triangles.find(id);
        auto mesh = meshes.find(inst.mesh);
        if (have == instance_triangles.end() || mesh == meshes.end()) continue;
        auto dm = device_meshes.find(inst.mesh);
        if (dm == device_meshes.end() || dm->second.version != inst.baked_mesh_version) {
            DeviceMeshRec rec{mesh_store_host.size() / 24u, mesh->second.size(), inst.baked_mesh_version};
            mesh_store_host.reserve(mesh_store_host.size() + 24u * rec.count);
            for (const StMeshTriangle& m : mesh->second) {
                for (int v = 0; v < 3; v++) for (int c = 0; c < 3; c++) mesh_store_host.push_back(m.positions[v][c]);
                for (int v = 0; v < 3; v++) for (int c = 0; c < 3; c++) mesh_store_host.push_back(m.normals[v][c]);
                for (int v = 0; v < 3; v++) for (int c = 0; c < 2; c++) mesh_store_host.push_back(m.uvs[v][c]);
            }
            device_meshes[inst.mesh] = rec; dm = device_meshes.find(inst.mesh); store_grew = true;
        }
        const Affine& inv = inst.xform_inv;
        Job j;
        j.x = f4(inst.xform.x, 0.0f); j.y = f4(inst.xform.y, 0.0f); j.z = f4(inst.xform.z, 0.0f); j.t = f4(inst.xform.t, 0.0f);
        j.r0 = make_float4(inv.x.x, inv.y.x, inv.z.x, 0.0f); j.r1 = make_float4(inv.x.y, inv.y.y, inv.z.y, 0.0f); j.r2 = make_float4(inv.x.z, inv.y.z, inv.z.z, 0.0f);
        j.mesh_first = (uint32_t)dm->second.first; j.count = (uint32_t)dm->second.count; j.slot_first = (uint32_t)have->second.first; j.xslot = inst.xslot;
        jobs.push_back(j); starts.push_back(starts.back() + j.count);
    }
    t.pending_moves.clear();
    if (jobs.empty()) return ST_OK;
    int rc;
    if (store_grew) {
        const size_t bytes = mesh_store_host.size() * sizeof(float);
        if (bytes > d_mesh_store.capacity) {   // a new allocation: everything again (hipFree waits for the kernels that read the old one)
            if ((rc = d_mesh_store.upload(mesh_store_host.data(), bytes, up, staging, pageable))) return rc;
        } else if ((rc = d_mesh_store.upload_range(mesh_store_host.data(), mesh_store_uploaded * sizeof(float), (mesh_store_host.size() - mesh_store_uploaded) * sizeof(float), up, staging, pageable))) return rc;
        mesh_store_uploaded = mesh_store_host.size();
    }
    if ((rc = t.bake_jobs.upload(jobs.data(), jobs.size() * sizeof(Job), up, staging, pageable))) return rc;
    if ((rc = t.bake_starts.upload(starts.data(), starts.size() * sizeof(uint32_t), up, staging, pageable))) return rc;
    // the EXACT build's kernel, whatever arithmetic the frames use: the baked arrays are the host's bits
    launchers_exact().launch_bvh_bake(t.bake_jobs.ptr, static_cast<const uint32_t*>(t.bake_starts.ptr), (uint32_t)jobs.size(), starts.back(), static_cast<const float*>(d_mesh_store.ptr),
                                      static_cast<float4*>(t.tri_geo.ptr), static_cast<float4*>(t.tri_bounds.ptr), static_cast<float4*>(t.tri_attr.ptr), static_cast<float4*>(t.bvh.ptr),
                                      t.device_built ? nullptr : static_cast<const uint32_t*>(t.entry_of_tri.ptr), up);
    device_bakes++; device_baked_triangles += starts.back();
    return ST_OK;
}

}  // namespace st
