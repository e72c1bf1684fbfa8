// st_abi.cpp — the C ABI of libstrolle_hip.so (include/strolle_hip.h): one entry point per strolle::Engine method
// (strolle/src/lib.rs:132-301) plus the seams this library adds. See st_engine.h.
#include "st_engine.h"

using namespace st;
static Engine* E(StEngine* e) { return reinterpret_cast<Engine*>(e); }
#define ST_REQUIRE(cond, msg) do { if (!(cond)) return fail(ST_ERR_INVALID_ARGUMENT, msg); } while (0)

extern "C" {

const char* st_last_error(void) { return g_last_error.c_str(); }
// the commit this library was built from (csrc/Makefile writes st_build_commit.inc; "+dirty": the kernel / host sources differed from that commit)
#if __has_include("st_build_commit.inc")
#include "st_build_commit.inc"
#endif
#ifndef ST_BUILD_COMMIT
#define ST_BUILD_COMMIT "unknown"
#endif
const char* st_build_commit(void) { return ST_BUILD_COMMIT; }
// st_gltf.cpp reports through the same thread-local message
extern "C" int st_internal_fail(int status, const char* message) { return fail(status, message ? message : ""); }

int st_engine_create(int device_ordinal, StEngine** out) {
    ST_REQUIRE(out, "out is NULL");
    std::unique_ptr<Engine> e(new Engine());
    if (device_ordinal >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= device_ordinal)
            return fail(ST_ERR_NO_DEVICE, "no HIP device with that ordinal (this library has no CPU rendering path)");
        ST_HIP(hipSetDevice(device_ordinal));
        e->device = device_ordinal; e->has_device = true;
        // LUT storage (zero until the first non-heatmap render generates them)
        const size_t lut_bytes[3] = {sizeof(float4) * 256 * 64, sizeof(float4) * 32 * 32, sizeof(float4) * 256 * 256};
        DeviceArray* luts[3] = {&e->d_transmittance, &e->d_scattering, &e->d_sky};
        for (int i = 0; i < 3; i++) { ST_HIP(hipMalloc(&luts[i]->ptr, lut_bytes[i])); luts[i]->capacity = lut_bytes[i]; ST_HIP(hipMemset(luts[i]->ptr, 0, lut_bytes[i])); }
        ST_HIP(hipMalloc(&e->d_byte_luts.ptr, sizeof(float) * 1024)); e->d_byte_luts.capacity = sizeof(float) * 1024;
        {   // the wide walks' overflow words (st_engine.h walk_flags_host): page-locked, mapped, written by a kernel only when a push is dropped
            void* host = nullptr; void* dev = nullptr;
            ST_HIP(hipHostMalloc(&host, 64, hipHostMallocMapped));
            memset(host, 0, 64);
            ST_HIP(hipHostGetDevicePointer(&dev, host, 0));
            e->walk_flags_host = static_cast<volatile uint32_t*>(host); e->walk_flags_dev = static_cast<uint32_t*>(dev);
        }
        e->L.launch_build_byte_luts(static_cast<float*>(e->d_byte_luts.ptr), nullptr);
        ST_HIP(hipDeviceSynchronize());
    }
    *out = reinterpret_cast<StEngine*>(e.release());
    return ST_OK;
}
void st_engine_destroy(StEngine* e) { delete E(e); }

int st_mesh_insert(StEngine* e, StHandle id, const StMeshTriangle* t, size_t count) {
    ST_REQUIRE(e && (t || count == 0), "null argument");
    if (count == 0) return fail(ST_ERR_EMPTY_MESH, "mesh contains no triangles");
    E(e)->meshes[id].assign(t, t + count);
    E(e)->mesh_version[id] = E(e)->next_mesh_version++;   // an instance baked from the earlier mesh of this handle is re-baked by the host
    return ST_OK;
}
int st_mesh_remove(StEngine* e, StHandle id) { ST_REQUIRE(e, "null engine"); E(e)->meshes.erase(id); E(e)->mesh_version.erase(id); return ST_OK; }

int st_material_insert(StEngine* e, StHandle id, const StMaterial* m) {
    ST_REQUIRE(e && m, "null argument");
    Engine* en = E(e);
    auto it = en->material_slot.find(id);
    if (it != en->material_slot.end()) en->materials[it->second] = *m;
    else {
        size_t b, end_;
        uint32_t slot;
        if (en->material_free.take(1, &b, &end_)) slot = (uint32_t)b;  // materials.rs:48-50 (the slot keeps its previous contents)
        else { en->materials.push_back(*m); slot = (uint32_t)en->materials.size() - 1u; }
        en->material_slot[id] = slot;
    }
    en->materials_dirty = true;
    return ST_OK;
}
int st_material_has(StEngine* e, StHandle id) { return e && E(e)->material_slot.count(id) ? 1 : 0; }
int st_material_remove(StEngine* e, StHandle id) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    auto it = en->material_slot.find(id);
    if (it == en->material_slot.end()) return ST_OK;
    en->material_free.give(it->second, it->second);  // `give(id..id)`: an empty range, as in materials.rs:74
    en->material_slot.erase(it);
    en->materials_dirty = true;
    return ST_OK;
}

// Images::insert (images.rs:54-105): finds the rectangle for image `id` and makes the host copy of the atlas tall enough.
static int place_image(Engine* en, StHandle id, uint32_t w, uint32_t h, Engine::ImageRec* out) {
    constexpr uint32_t kAtlasW = Engine::kAtlasW;
    if (w > kAtlasW) return fail(ST_ERR_ATLAS_FULL, "image wider than the atlas");
    auto it = en->images.find(id);
    Engine::ImageRec rec;
    if (it != en->images.end() && it->second.w == w && it->second.h == h) rec = it->second;  // same size: rewritten in place (images.rs:61-63)
    else {
        if (it != en->images.end()) {  // another size: the old rectangle is given back first (images.rs:64-66)
            en->atlas_rects.release(it->second.x, it->second.y, it->second.w);
            en->images.erase(it);
            en->device_images.erase(id);
            en->materials_dirty = true;
        }
        rec = {0, 0, w, h};
        if (!en->atlas_rects.allocate(w, h, &rec.x, &rec.y)) return fail(ST_ERR_ATLAS_FULL, "no more space in the atlas");
    }
    const uint32_t need_h = rec.y + h;
    if (en->atlas_w == 0) en->atlas_w = kAtlasW;
    if (need_h > en->atlas_h) {
        // grow in 256-row steps; rects are stored in texels, so existing materials stay valid after a rebuild
        en->atlas_h = (need_h + 255u) & ~255u;
        en->atlas.resize((size_t)en->atlas_w * en->atlas_h * 4, 0);
    }
    en->images[id] = rec;
    en->atlas_dirty = true; en->materials_dirty = true;
    *out = rec;
    return ST_OK;
}

int st_image_insert_rgba8(StEngine* e, StHandle id, uint32_t w, uint32_t h, const uint8_t* rgba, int /*srgb*/) {
    ST_REQUIRE(e && rgba && w && h && id, "bad image");
    Engine* en = E(e);
    Engine::ImageRec rec;
    if (int rc = place_image(en, id, w, h, &rec)) return rc;
    en->device_images.erase(id);
    for (uint32_t y = 0; y < h; y++) memcpy(&en->atlas[((size_t)(rec.y + y) * en->atlas_w + rec.x) * 4], rgba + (size_t)y * w * 4, (size_t)w * 4);
    return ST_OK;
}
int st_image_insert_device_rgba8(StEngine* e, StHandle id, uint32_t w, uint32_t h, const void* device_rgba, size_t row_pitch_bytes, int is_dynamic) {
    ST_REQUIRE(e && device_rgba && w && h && id, "bad image");
    ST_REQUIRE(row_pitch_bytes >= (size_t)w * 4, "row pitch smaller than a row");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "device images need a device engine");
    Engine::ImageRec rec;
    if (int rc = place_image(en, id, w, h, &rec)) return rc;
    en->device_images[id] = Engine::DeviceImage{device_rgba, row_pitch_bytes, is_dynamic != 0, true};
    return ST_OK;
}
int st_image_remove(StEngine* e, StHandle id) {  // images.rs:107-113
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    auto it = en->images.find(id);
    if (it == en->images.end()) return ST_OK;
    en->atlas_rects.release(it->second.x, it->second.y, it->second.w);
    en->images.erase(it);
    en->device_images.erase(id);
    en->materials_dirty = true;
    return ST_OK;
}
int st_debug_image_rect(StEngine* e, StHandle id, uint32_t out_xywh[4]) {
    ST_REQUIRE(e && out_xywh, "null argument");
    auto it = E(e)->images.find(id);
    if (it == E(e)->images.end()) return fail(ST_ERR_INVALID_ARGUMENT, "no such image");
    out_xywh[0] = it->second.x; out_xywh[1] = it->second.y; out_xywh[2] = it->second.w; out_xywh[3] = it->second.h;
    return ST_OK;
}

int st_instance_insert(StEngine* e, StHandle id, StHandle mesh, StHandle material, const float xform[12]) {
    ST_REQUIRE(e && xform, "null argument");
    Engine* en = E(e);
    const Affine x = affine_from12(xform);
    for (auto& r : en->instances)
        if (r.id == id) { r.prev_xform = r.xform; r.mesh = mesh; r.material = material; r.xform = x; r.xform_inv = affine_inverse(x); r.dirty = true; en->instances_dirty = true; return ST_OK; }
    uint32_t xslot;
    if (!en->xslot_free.empty()) { xslot = en->xslot_free.back(); en->xslot_free.pop_back(); }
    else { xslot = (uint32_t)(en->instance_xforms.size() / 8u); en->instance_xforms.resize(en->instance_xforms.size() + 8u, make_float4(0, 0, 0, 0)); }
    en->instances.push_back({id, mesh, material, x, affine_inverse(x), x, true, xslot});
    en->instances_dirty = true;
    return ST_OK;
}
int st_instance_remove(StEngine* e, StHandle id) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    for (size_t i = 0; i < en->instances.size(); i++)
        if (en->instances[i].id == id) { en->xslot_free.push_back(en->instances[i].xslot); en->instances.erase(en->instances.begin() + i); en->instances_dirty = true; en->instance_removed = true; break; }
    en->drop_instance_triangles(id);
    return ST_OK;
}
int st_light_insert(StEngine* e, StHandle id, const StLight* l) { ST_REQUIRE(e && l, "null argument"); E(e)->insert_light(id, *l); return ST_OK; }
int st_light_remove(StEngine* e, StHandle id) { ST_REQUIRE(e, "null engine"); E(e)->remove_light(id); return ST_OK; }
int st_sun_update(StEngine* e, float azimuth, float altitude) { ST_REQUIRE(e, "null engine"); E(e)->sun_azimuth = azimuth; E(e)->sun_altitude = altitude; E(e)->sun_dirty = true; return ST_OK; }

int st_camera_create(StEngine* e, const StCamera* c, StHandle* out) {
    ST_REQUIRE(e && c && out && c->width && c->height, "bad camera");
    Engine* en = E(e);
    std::unique_ptr<CameraState> s(new CameraState());
    s->desc = *c;
    s->curr = Engine::serialize_camera(*c); s->prev = s->curr;
    const int rc = en->allocate_camera(*s);
    if (rc) return rc;
    *out = en->next_camera++;
    s->handle = *out;
    en->cameras[*out] = std::move(s);
    return ST_OK;
}
int st_camera_update(StEngine* e, StHandle h, const StCamera* c) {
    ST_REQUIRE(e && c && c->width && c->height, "bad camera");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    CameraState& s = *it->second;
    const bool invalidated = s.desc.mode != c->mode || s.desc.denoise != c->denoise || s.desc.depth != c->depth || s.desc.width != c->width || s.desc.height != c->height;
    s.desc = *c;
    s.prev = s.curr;
    s.curr = Engine::serialize_camera(*c);
    if (invalidated) { if (en->has_device) { ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize()); } return en->allocate_camera(s); }  // camera.rs:17-48: buffers are rebuilt
    return ST_OK;
}
int st_camera_delete(StEngine* e, StHandle h) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return ST_OK;
    if (en->has_device) { ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize()); Engine::release_camera(*it->second); }
    en->dist_forget_camera(h);
    en->cameras.erase(it);
    return ST_OK;
}
int st_camera_set_window(StEngine* e, StHandle h, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    CameraState& s = *it->second;
    if (x0 == 0 && x1 == 0) x1 = s.desc.width;
    if (y0 == 0 && y1 == 0) y1 = s.desc.height;
    ST_REQUIRE(y0 < y1 && y1 <= s.desc.height, "bad row window");
    ST_REQUIRE(x0 < x1 && x1 <= s.desc.width, "bad column window");
    // half-resolution passes work on 2x1 cells in tiles of 8 cells: a window starts and ends on a multiple of 16 pixels (or at the frame's edge)
    ST_REQUIRE(x0 % 16u == 0u && (x1 % 16u == 0u || x1 == s.desc.width), "window columns must be multiples of 16 (or the frame's right edge)");
    s.row0 = y0; s.row1 = y1; s.col0 = x0; s.col1 = x1;
    return ST_OK;
}
int st_camera_set_rows(StEngine* e, StHandle h, uint32_t y0, uint32_t y1) { return st_camera_set_window(e, h, 0u, y0, 0u, y1); }

int st_camera_set_output_format(StEngine* e, StHandle h, int format) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    ST_REQUIRE(format >= ST_FORMAT_RGBA32F && format <= ST_FORMAT_BGRA8_UNORM_SRGB, "unknown output format");
    it->second->out_format = (uint32_t)format;
    return ST_OK;
}

int st_tick(StEngine* e, void* stream) { ST_REQUIRE(e, "null engine"); return E(e)->tick(static_cast<hipStream_t>(stream)); }
int st_render_camera(StEngine* e, StHandle h, void* out, void* stream) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->render(*it->second, out, static_cast<hipStream_t>(stream));
}

int st_debug_keep_all_planes(StEngine* e, int keep) { ST_REQUIRE(e, "null engine"); E(e)->tuning.lean_frame = keep == 0 ? 1u : 0u; return ST_OK; }
int st_camera_present_copy(StEngine* e, StHandle h, const void* src_device, void* dst_host, size_t bytes, void* stream) {
    ST_REQUIRE(e && src_device && dst_host && bytes, "null argument");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->present_copy(*it->second, src_device, dst_host, bytes, static_cast<hipStream_t>(stream));
}
int st_camera_present_ready(StEngine* e, StHandle h, const void* dst_host, int wait, int* ready) {
    ST_REQUIRE(e && ready, "null argument");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->present_ready(*it->second, dst_host, wait, ready);
}

int st_set_seed(StEngine* e, uint64_t seed) { ST_REQUIRE(e, "null engine"); E(e)->base_seed = seed; return ST_OK; }
int st_set_blue_noise(StEngine* e, const uint8_t* rgba, size_t bytes) {
    ST_REQUIRE(e && rgba && bytes == 256 * 256 * 4, "blue noise must be 256x256 RGBA8");
    E(e)->blue_noise.assign(rgba, rgba + bytes); E(e)->blue_noise_dirty = true;
    return ST_OK;
}
int st_debug_read_lut(StEngine* e, int what, float* out, size_t capacity_floats, size_t* written_floats) {
    ST_REQUIRE(e && what >= 0 && what < 3, "bad lut id");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine has no LUTs");
    const size_t n[3] = {256 * 64 * 4, 32 * 32 * 4, 256 * 256 * 4};
    const DeviceArray* src[3] = {&en->d_transmittance, &en->d_scattering, &en->d_sky};
    if (written_floats) *written_floats = n[what];
    if (!out) return ST_OK;
    ST_REQUIRE(capacity_floats >= n[what], "buffer too small");
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    ST_HIP(hipMemcpy(out, src[what]->ptr, n[what] * sizeof(float), hipMemcpyDeviceToHost));
    return ST_OK;
}

int st_camera_read_buffer(StEngine* e, StHandle h, int id, void* out, size_t capacity, size_t* written) {
    ST_REQUIRE(e && id >= 0 && id < ST_BUF_COUNT, "bad buffer id");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine has no camera buffers");
    CameraState& c = *it->second;
    if (written) *written = c.plane_bytes[id];
    if (!out) return ST_OK;
    ST_REQUIRE(capacity >= c.plane_bytes[id], "buffer too small");
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    const float4* src = (id == ST_BUF_GI_RESERVOIRS_1 && c.gi_aliased) ? c.plane[ST_BUF_GI_RESERVOIRS_0] : c.plane[id];
    ST_HIP(hipMemcpy(out, src, c.plane_bytes[id], hipMemcpyDeviceToHost));
    return ST_OK;
}
// Did the camera's last frame leave this plane unwritten (the lean frame, st_debug_keep_all_planes)? A read-back then returns what an
// earlier launch or frame stored there.
int st_camera_buffer_stale(StEngine* e, StHandle h, int id, int* stale) {
    ST_REQUIRE(e && stale && id >= 0 && id < ST_BUF_COUNT, "bad argument");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    const CameraState& c = *it->second;
    const uint32_t lean = c.last_lean;
    bool s = false;
    if (lean & kLeanPrim) s |= id == ST_BUF_VELOCITY_MAP || id == ST_BUF_PRIM_SURFACE_MAP_A || id == ST_BUF_PRIM_SURFACE_MAP_B;
    if (lean & kLeanSamples) s |= id == ST_BUF_DI_DIFF_SAMPLES || id == ST_BUF_GI_DIFF_SAMPLES;
    if (lean & kLeanGiRes2) s |= id == ST_BUF_GI_RESERVOIRS_2;
    if (lean & kLeanGiMid) s |= id == ST_BUF_GI_RESERVOIRS_3;
    if (c.last_lean_composed) s |= id == ST_BUF_DI_DIFF_CURR_COLORS || id == ST_BUF_GI_DIFF_CURR_COLORS;
    *stale = s ? 1 : 0;
    return ST_OK;
}
int st_camera_write_buffer(StEngine* e, StHandle h, int id, const void* data, size_t bytes) {
    ST_REQUIRE(e && data && id >= 0 && id < ST_BUF_COUNT, "bad buffer id");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine has no camera buffers");
    CameraState& c = *it->second;
    ST_REQUIRE(bytes == c.plane_bytes[id], "size does not match the buffer");
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    { const int rc = materialize_gi_history(c); if (rc) return rc; }
    ST_HIP(hipMemcpy(c.plane[id], data, bytes, hipMemcpyHostToDevice));
    if (id == ST_BUF_PRIM_SURFACE_MAP_A) c.surface_map_replaced[0] = true;
    if (id == ST_BUF_PRIM_SURFACE_MAP_B) c.surface_map_replaced[1] = true;
    return ST_OK;
}
int st_debug_variance_flags(StEngine* e, StHandle h, uint64_t* tile_mask_out, size_t capacity_tiles, size_t* tiles) {
    ST_REQUIRE(e && tiles, "null argument");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine");
    CameraState& c = *it->second;
    *tiles = c.tile_mask_tiles;
    if (!tile_mask_out) return ST_OK;
    ST_REQUIRE(capacity_tiles >= c.tile_mask_tiles, "buffer too small");
    ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize());
    ST_HIP(hipMemcpy(tile_mask_out, c.tile_mask, c.tile_mask_tiles * sizeof(uint64_t), hipMemcpyDeviceToHost));
    return ST_OK;
}
int st_debug_set_pass_mask(StEngine* e, uint64_t mask) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    if (en->has_device) { ST_HIP(hipSetDevice(en->device)); for (auto& kv : en->cameras) { const int rc = materialize_gi_history(*kv.second); if (rc) return rc; } }
    en->pass_mask = mask;
    return ST_OK;
}
int st_debug_set_launch_filter(StEngine* e, uint64_t filter) {
    ST_REQUIRE(e, "null engine");
    E(e)->launch_filter = filter;
    return ST_OK;
}
int st_debug_last_launches(StEngine* e, uint64_t* out_bits, size_t capacity, size_t* count) {
    ST_REQUIRE(e && count, "null argument");
    const std::vector<uint64_t>& v = E(e)->last_launches;
    *count = v.size();
    for (size_t i = 0; i < v.size() && i < capacity && out_bits; i++) out_bits[i] = v[i];
    return ST_OK;
}
int st_engine_set_arithmetic(StEngine* e, int arithmetic) {
    ST_REQUIRE(e, "null engine");
    ST_REQUIRE(arithmetic == ST_ARITH_FAST || arithmetic == ST_ARITH_EXACT, "unknown arithmetic");
    Engine* en = E(e);
    if (en->arithmetic == arithmetic) return ST_OK;
    en->arithmetic = arithmetic;
    en->L = arithmetic == ST_ARITH_EXACT ? launchers_exact() : launchers_fast();
    if (en->has_device) {  // frames in flight finish with the tables they were launched with; the byte tables follow the build
        ST_HIP(hipSetDevice(en->device));
        ST_HIP(hipDeviceSynchronize());
        en->L.launch_build_byte_luts(static_cast<float*>(en->d_byte_luts.ptr), nullptr);
        ST_HIP(hipDeviceSynchronize());
        // the atmosphere LUTs are regenerated by the next render with the new build's routines
        en->atmosphere_initialized = false; en->sky_known = false;
    }
    return ST_OK;
}
int st_engine_get_tuning(StEngine* e, StTuning* out) { ST_REQUIRE(e && out, "null argument"); *out = E(e)->tuning; return ST_OK; }
int st_engine_set_tuning(StEngine* e, const StTuning* t) { ST_REQUIRE(e && t, "null argument"); return E(e)->set_tuning(*t); }
int st_engine_get_arithmetic(StEngine* e, int* out) { ST_REQUIRE(e && out, "null argument"); *out = E(e)->arithmetic; return ST_OK; }
int st_camera_ray_count(StEngine* e, StHandle h, uint64_t* out, int reset) {
    ST_REQUIRE(e && out, "null argument");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine");
    unsigned long long host[2 * KS_COUNT];
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    { const int rc2 = read_counters(*it->second, host); if (rc2) return rc2; }
    uint64_t total = 0;
    for (int i = 0; i < KS_COUNT; i++) total += host[2 * i];
    *out = total;
    if (reset) { ST_HIP(hipMemset(it->second->counters, 0, kCounterBytes)); memset(it->second->profiled_traversal_bytes, 0, sizeof(it->second->profiled_traversal_bytes)); ST_HIP(hipDeviceSynchronize()); }
    return ST_OK;
}
int st_debug_read_scene(StEngine* e, int what, void* out, size_t capacity, size_t* written) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    const void* p; size_t bytes;
    if ((what == 0 || what == 1 || what == 4 || (what >= 7 && what <= 15)) && en->any_host_stale()) en->bake_stale_on_host();   // instances the device moved: the host arrays catch up
    if (en->host_tree_stale && (what == 0 || what == 4 || (what >= 7 && what <= 15))) en->rebuild_host_tree(false);   // ST_BVH_BUILD_DEVICE left the host's tree behind
    if ((what == 0 || what == 4) && en->host_stream_stale) { en->refit_stream(); en->host_stream_stale = false; }  // device refits since the host copy was current
    switch (what) {
        case 0: p = en->bvh_stream.data(); bytes = en->bvh_stream.size() * sizeof(float4); break;
        case 1: p = en->triangles.data(); bytes = en->triangles.size() * sizeof(HostTriangle); break;
        case 2: p = en->gpu_lights.data(); bytes = en->gpu_lights.size() * sizeof(GpuLight); break;
        case 3: p = en->gpu_materials.data(); bytes = en->gpu_materials.size() * sizeof(GpuMaterial); break;
        case 4: en->expand_stream(); p = en->bvh_upload_.data(); bytes = (size_t)en->device_bvh_len * sizeof(float4); break;  // as st_tick would upload it now
        case 6: {  // the device stream as it is on the device right now (the live copy): what a device refit left there
            if (!en->has_device || !en->scene_uploaded) return fail(ST_ERR_NO_DEVICE, "no device copy of the scene");
            const size_t n = en->sets[en->live].bvh.capacity ? (size_t)en->live_bvh_texels : 0;
            en->readback_.resize(n);
            ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize());
            if (n) ST_HIP(hipMemcpy(en->readback_.data(), en->sets[en->live].bvh.ptr, n * sizeof(float4), hipMemcpyDeviceToHost));
            p = en->readback_.data(); bytes = n * sizeof(float4); break;
        }
        case 7: case 8: case 9: case 10: case 11: case 12: case 13: {  // the device refit's inputs (k_bvh.hip), built here for a host-side emulation
            if (en->host_stream_stale) { en->refit_stream(); en->host_stream_stale = false; }
            en->expand_stream(); en->index_device_tree();
            en->readback_levels_.clear();
            for (const auto& l : en->refit_levels_) { en->readback_levels_.push_back(l.first); en->readback_levels_.push_back(l.second); }
            const std::vector<uint32_t>* v = what == 7 ? &en->parent_ : what == 8 ? &en->refit_local_ : what == 9 ? &en->refit_items_ : what == 10 ? &en->refit_batch_off_
                                           : what == 11 ? &en->readback_levels_ : &en->entry_of_tri_;
            if (what == 13) { p = en->tri_bounds.data(); bytes = en->tri_bounds.size() * sizeof(float4); }
            else { p = v->data(); bytes = v->size() * sizeof(uint32_t); }
            break;
        }
        case 14: case 15: {  // the wide stream's topology (k_bvh.hip k_bvh_wide), built here: 14 = 8 words per node (4 box sources, 4 links), preceded by
                             // one word holding the root's link; 15 = the contract entry of every leaf record
            if (en->host_stream_stale) { en->refit_stream(); en->host_stream_stale = false; }
            en->expand_stream(); en->build_wide_topology(); en->wide_built_for_ = ~0ull;   // (a later tick builds its own)
            en->readback_levels_.assign(1, en->wide_root_ | (en->wide_stack_need_ << 8));   // (bits 8..: the most entries a wide walk can have pending)
            en->readback_levels_.insert(en->readback_levels_.end(), en->wide_topo_.begin(), en->wide_topo_.end());
            const std::vector<uint32_t>* v = what == 14 ? &en->readback_levels_ : &en->wide_leaf_entry_;
            p = v->data(); bytes = v->size() * sizeof(uint32_t); break;
        }
        case 16: case 17: {  // the wide stream as it is on the device right now (the live copy): 16 = nodes (64 B each), 17 = leaf records (48 B each)
            if (!en->has_device || !en->scene_uploaded) return fail(ST_ERR_NO_DEVICE, "no device copy of the scene");
            const auto& t = en->sets[en->live];
            const size_t n = (t.wide_for_entries || t.device_built) ? (what == 16 ? (size_t)t.wide_nodes * 4u : (size_t)t.wide_leaves * 3u) : 0;
            en->readback_.resize(n);
            ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize());
            if (n) ST_HIP(hipMemcpy(en->readback_.data(), what == 16 ? t.bvh_wide.ptr : static_cast<const void*>(static_cast<const float4*>(t.bvh_wide.ptr) + 4u * (size_t)t.wide_nodes), n * sizeof(float4), hipMemcpyDeviceToHost));
            p = en->readback_.data(); bytes = n * sizeof(float4); break;
        }
        default: return fail(ST_ERR_INVALID_ARGUMENT, "unknown scene buffer");
    }
    if (written) *written = bytes;
    if (!out) return ST_OK;
    ST_REQUIRE(capacity >= bytes, "buffer too small");
    if (bytes) memcpy(out, p, bytes);
    return ST_OK;
}
int st_debug_world(StEngine* e, uint32_t* light_count, uint32_t* next_frame) {
    ST_REQUIRE(e && light_count && next_frame, "null argument");
    *light_count = E(e)->light_count; *next_frame = E(e)->frame;
    return ST_OK;
}

int st_set_bvh_refresh(StEngine* e, int mode) {
    ST_REQUIRE(e, "null engine");
    ST_REQUIRE(mode == ST_BVH_REBUILD || mode == ST_BVH_REFIT || mode == ST_BVH_REFIT_DEVICE || mode == ST_BVH_BUILD_DEVICE || mode == ST_BVH_AUTO, "unknown refresh mode");
    Engine* en = E(e);
    if (en->bvh_refresh_mode != mode) { en->bvh_refresh_mode = mode; en->have_topology = false; }
    return ST_OK;
}
int st_debug_device_builds(StEngine* e, uint64_t* ticks) { ST_REQUIRE(e && ticks, "null argument"); *ticks = E(e)->device_builds; return ST_OK; }
int st_debug_device_tree_refits(StEngine* e, uint64_t* ticks) { ST_REQUIRE(e && ticks, "null argument"); *ticks = E(e)->device_tree_refits; return ST_OK; }
int st_debug_bvh_depth(StEngine* e, uint32_t* deepest_internal_chain, uint32_t* stack_entries) {
    ST_REQUIRE(e && deepest_internal_chain && stack_entries, "null argument");
    *deepest_internal_chain = E(e)->bvh_stack_need; *stack_entries = E(e)->stack_entries;
    return ST_OK;
}
int st_debug_walk_overflow(StEngine* e, uint64_t* overflows, uint32_t* wide_stack_entries, uint32_t* packets_off) {
    ST_REQUIRE(e && overflows && wide_stack_entries && packets_off, "null argument");
    if (E(e)->has_device && E(e)->walk_flags_host) {   // frames still in flight count too: wait for them, then look (the next st_tick reports what is found here)
        ST_HIP(hipSetDevice(E(e)->device)); ST_HIP(hipDeviceSynchronize());
        if ((E(e)->walk_flags_host[0] | E(e)->walk_flags_host[1]) != 0u) E(e)->note_walk_overflow();
    }
    *overflows = E(e)->walk_overflows; *wide_stack_entries = E(e)->wide_stack_entries_now(); *packets_off = E(e)->packets_overflowed ? 1u : 0u;
    return ST_OK;
}
int st_debug_auto_tree(StEngine* e, float* leaf_run_weight, uint32_t* first_tree_on_device) {
    ST_REQUIRE(e && leaf_run_weight && first_tree_on_device, "null argument");
    *leaf_run_weight = E(e)->host_leaf_run_weight; *first_tree_on_device = E(e)->auto_first_on_device ? 1u : 0u;
    return ST_OK;
}
int st_debug_bvh_refits(StEngine* e, uint64_t* rebuilds, uint64_t* refits) {
    ST_REQUIRE(e && rebuilds && refits, "null argument");
    *rebuilds = E(e)->rebuilds; *refits = E(e)->refits;
    return ST_OK;
}
int st_debug_bvh_device_refits(StEngine* e, uint64_t* device_refits) {
    ST_REQUIRE(e && device_refits, "null argument");
    *device_refits = E(e)->device_refits;
    return ST_OK;
}
int st_debug_device_bakes(StEngine* e, uint64_t* ticks, uint64_t* triangles) {
    ST_REQUIRE(e && ticks && triangles, "null argument");
    *ticks = E(e)->device_bakes; *triangles = E(e)->device_baked_triangles;
    return ST_OK;
}
int st_debug_bvh_refresh(StEngine* e, uint64_t* primitives, uint64_t* reused) {
    ST_REQUIRE(e && primitives && reused, "null argument");
    *primitives = E(e)->bvh.prims.size(); *reused = E(e)->bvh.reused_primitives();
    return ST_OK;
}

// Streaming ceiling of this device by this library's own kernel: `iters` grid-stride float4 copies of `bytes` (src -> dst, both
// allocated here), best of them, as (bytes read + bytes written) / time in GB/s.
int st_debug_copy_bandwidth(StEngine* e, size_t bytes, int iters, double* out_gbps) {
    ST_REQUIRE(e && out_gbps && bytes >= 16 && iters > 0, "bad argument");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine");
    ST_HIP(hipSetDevice(en->device));
    const size_t n = bytes / 16;
    void *src = nullptr, *dst = nullptr;
    ST_HIP(hipMalloc(&src, n * 16));
    if (hipMalloc(&dst, n * 16) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(src); return fail(ST_ERR_HIP, "hipMalloc(copy destination) failed"); }
    hipEvent_t t0 = nullptr, t1 = nullptr;
    int rc = ST_OK; double best = 0.0;
    if (hipMemset(src, 0x3c, n * 16) != hipSuccess || hipEventCreate(&t0) != hipSuccess || hipEventCreate(&t1) != hipSuccess) rc = fail(ST_ERR_HIP, "copy bandwidth set-up failed");
    for (int i = 0; rc == ST_OK && i < iters + 1; i++) {   // the first pass is a warm-up
        (void)hipEventRecord(t0, nullptr);
        en->L.launch_copy_float4(static_cast<float4*>(dst), static_cast<const float4*>(src), n, 256u * 16u, nullptr);
        (void)hipEventRecord(t1, nullptr);
        if (hipEventSynchronize(t1) != hipSuccess) { rc = fail(ST_ERR_HIP, "copy kernel failed"); break; }
        float ms = 0.0f; (void)hipEventElapsedTime(&ms, t0, t1);
        if (i > 0 && ms > 0.0f) best = std::max(best, 2.0 * (double)(n * 16) / (ms * 1e-3) / 1e9);
    }
    if (t0) (void)hipEventDestroy(t0);
    if (t1) (void)hipEventDestroy(t1);
    (void)hipFree(src); (void)hipFree(dst);
    *out_gbps = best;
    return rc;
}

int st_profile_enable(StEngine* e, int enabled) { ST_REQUIRE(e, "null engine"); E(e)->profiling = (enabled & 1) != 0; E(e)->count_bytes = (enabled & 2) != 0; E(e)->profile_group_atrous = (enabled & 4) != 0; E(e)->profile_kernel_events = (enabled & 8) != 0; return ST_OK; }
int st_profile_read(StEngine* e, StKernelProfile* out, size_t capacity, size_t* count, int reset) {
    ST_REQUIRE(e && out && count, "null argument");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine");
    ST_HIP(hipSetDevice(en->device));
    const int rc = en->drain_profile();
    if (rc) return rc;
    // traversal bytes (the reference's used_memory, summed on the device) join the screen-space bytes per kernel
    ST_HIP(hipDeviceSynchronize());
    for (auto& kv : en->cameras) {
        CameraState& c = *kv.second;
        unsigned long long host[2 * KS_COUNT];
        { const int rc2 = read_counters(c, host); if (rc2) return rc2; }
        for (int i = 0; i < KS_COUNT; i++) {
            const unsigned long long total = host[2 * i + 1];
            if (total >= c.profiled_traversal_bytes[i]) {
                en->profile_totals[i].algorithmic_bytes += (double)(total - c.profiled_traversal_bytes[i]);
                en->profile_totals[i].traversal_bytes += (double)(total - c.profiled_traversal_bytes[i]);
            }
            c.profiled_traversal_bytes[i] = total;
        }
    }
    size_t n = 0;
    for (int i = 0; i < KS_COUNT && n < capacity; i++) {
        if (en->profile_totals[i].launches == 0 && en->profile_totals[i].traversal_bytes == 0.0) continue;  // bytes-only mode records no launches
        out[n] = en->profile_totals[i];
        n++;
    }
    *count = n;
    if (reset) en->reset_profile_totals();
    return ST_OK;
}

}  // extern "C"
