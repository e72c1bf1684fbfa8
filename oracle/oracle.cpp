// ORACLE — TEST INFRASTRUCTURE ONLY (see or_math.h header).
//
// C API of the CPU oracle. Mirrors the product's C ABI (include/strolle_hip.h)
// call for call with an `or_` prefix so one Python harness can drive both, plus
// a few probes used to pin the restatement against the reference's own unit
// tests. Built by oracle/Makefile into oracle/liboracle.so with
// -O2 -ffp-contract=off -fopenmp.
#include <cstdio>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <cstring>

#include "or_host.h"

using namespace orc;

extern "C" {

typedef struct OrEngine OrEngine;
static Engine* E(OrEngine* e) { return reinterpret_cast<Engine*>(e); }

int or_engine_create(OrEngine** out) { *out = reinterpret_cast<OrEngine*>(new Engine()); return 0; }
void or_engine_destroy(OrEngine* e) { delete E(e); }
int or_set_seed(OrEngine* e, uint64_t seed) { E(e)->base_seed = seed; return 0; }
int or_set_blue_noise(OrEngine* e, const uint8_t* rgba, size_t n) {
    if (n != 256 * 256 * 4) return 1;
    E(e)->blue_noise.assign(rgba, rgba + n);
    return 0;
}
// what = 0 transmittance (256x64), 1 scattering (32x32), 2 sky (256x256): RGBA32F texels holding f16-rounded values
int or_debug_read_lut(OrEngine* e, int what, float* out, size_t capacity_floats, size_t* written_floats) {
    const std::vector<Vec4>* v = what == 0 ? &E(e)->transmittance_lut : (what == 1 ? &E(e)->scattering_lut : &E(e)->sky_lut);
    if (written_floats) *written_floats = v->size() * 4;
    if (out) { if (capacity_floats < v->size() * 4) return 1; std::memcpy(out, (const void*)v->data(), v->size() * sizeof(Vec4)); }
    return 0;
}
int or_mesh_insert(OrEngine* e, uint64_t id, const ApiMeshTriangle* tris, size_t n) {
    E(e)->meshes[id] = std::vector<ApiMeshTriangle>(tris, tris + n);
    return 0;
}
int or_mesh_remove(OrEngine* e, uint64_t id) { E(e)->meshes.erase(id); return 0; }
int or_material_insert(OrEngine* e, uint64_t id, const ApiMaterial* m) { E(e)->insert_material(id, *m); return 0; }
int or_material_has(OrEngine* e, uint64_t id) { return E(e)->material_index.count(id) ? 1 : 0; }
int or_material_remove(OrEngine* e, uint64_t id) { E(e)->remove_material(id); return 0; }
int or_image_insert_rgba8(OrEngine* e, uint64_t id, uint32_t w, uint32_t h, const uint8_t* rgba, int /*srgb*/) { return E(e)->insert_image(id, w, h, rgba) ? 0 : 6; }
int or_image_remove(OrEngine* e, uint64_t id) { E(e)->remove_image(id); return 0; }
int or_debug_set_gi_neighbours(uint32_t n) { g_debug_gi_neighbours = n; return 0; }  // test-only, process-wide
// pending entries a ray may keep (24 = the reference's; 64 stands for unbounded), and what the rays since the last reset needed
int or_debug_set_stack_limit(int n) { if (n < 1 || n > BVH_STACK_MAX) return 1; g_or_stack_limit = n; return 0; }
int or_debug_stack_stats(unsigned long long* dropped_pushes, int* deepest_stack, int reset) {
    if (dropped_pushes) *dropped_pushes = __atomic_load_n(&g_or_dropped_pushes, __ATOMIC_RELAXED);
    if (deepest_stack) *deepest_stack = __atomic_load_n(&g_or_deepest_stack, __ATOMIC_RELAXED);
    if (reset) { __atomic_store_n(&g_or_dropped_pushes, 0ull, __ATOMIC_RELAXED); __atomic_store_n(&g_or_deepest_stack, 0, __ATOMIC_RELAXED); }
    return 0;
}
int or_set_bvh_refresh(OrEngine* e, int mode) { E(e)->bvh_refit_mode = mode == 1; return (mode == 0 || mode == 1) ? 0 : 1; }
int or_debug_bvh_refits(OrEngine* e, uint64_t* rebuilds, uint64_t* refits) { *rebuilds = E(e)->rebuilds; *refits = E(e)->refits; return 0; }
int or_debug_image_rect(OrEngine* e, uint64_t id, uint32_t out[4]) {
    auto it = E(e)->images.find(id);
    if (it == E(e)->images.end()) return 1;
    out[0] = it->second.x; out[1] = it->second.y; out[2] = it->second.w; out[3] = it->second.h;
    return 0;
}
int or_instance_insert(OrEngine* e, uint64_t id, uint64_t mesh, uint64_t material, const float xform[12]) {
    E(e)->insert_instance(id, mesh, material, xform);
    return 0;
}
int or_instance_remove(OrEngine* e, uint64_t id) { E(e)->remove_instance(id); return 0; }
int or_light_insert(OrEngine* e, uint64_t id, const ApiLight* l) { E(e)->insert_light(id, *l); return 0; }
int or_light_remove(OrEngine* e, uint64_t id) { E(e)->remove_light(id); return 0; }
int or_sun_update(OrEngine* e, float azimuth, float altitude) { E(e)->sun_azimuth = azimuth; E(e)->sun_altitude = altitude; E(e)->sun_dirty = true; return 0; }
int or_camera_create(OrEngine* e, const ApiCamera* c, uint64_t* out) { *out = E(e)->create_camera(*c); return 0; }
int or_camera_update(OrEngine* e, uint64_t h, const ApiCamera* c) { return E(e)->update_camera(h, *c) ? 0 : 3; }
int or_camera_delete(OrEngine* e, uint64_t h) { E(e)->cameras.erase(h); return 0; }
int or_tick(OrEngine* e) { E(e)->tick(); return 0; }
// What the render target's format does to the composed RGBA32F frame (camera.rs:170-175 `viewport.format`; the conversion
// itself is the GPU's store path, specified by the Khronos Data Format Specification 1.3, sections 10.1 "16-bit floating
// point" and 13.3 "sRGB transfer functions"): format 1 = Rgba16Float (round to nearest even), 2 = Rgba8UnormSrgb,
// 3 = Bgra8UnormSrgb (clamp to [0, 1], encode, scale by 255, round to nearest; alpha is linear and always 1 here).
static uint32_t srgb8(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 1.0f) return 255u;
    float y = x <= 0.0031308f ? x * 12.92f : 1.055f * stm_pow(x, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)(y * 255.0f + 0.5f);
}
int or_encode_output(const float* rgba32f, size_t pixels, int format, void* out) {
    if (format == 1) {
        uint16_t* o = static_cast<uint16_t*>(out);
        for (size_t i = 0; i < pixels * 4; i++) o[i] = (uint16_t)f16_bits(rgba32f[i]);
        return 0;
    }
    if (format == 2 || format == 3) {
        uint8_t* o = static_cast<uint8_t*>(out);
        for (size_t i = 0; i < pixels; i++) {
            const uint32_t r = srgb8(rgba32f[4 * i]), g = srgb8(rgba32f[4 * i + 1]), b = srgb8(rgba32f[4 * i + 2]);
            o[4 * i] = (uint8_t)(format == 2 ? r : b); o[4 * i + 1] = (uint8_t)g; o[4 * i + 2] = (uint8_t)(format == 2 ? b : r); o[4 * i + 3] = 255;
        }
        return 0;
    }
    return 1;
}
int or_debug_set_pass_mask(OrEngine* e, uint64_t mask) { E(e)->pass_mask = mask; return 0; }
int or_camera_write_buffer(OrEngine* e, uint64_t h, int buffer_id, const void* data, size_t bytes);
int or_render_camera(OrEngine* e, uint64_t h, float* out_rgba32f) { return E(e)->render_camera(h, reinterpret_cast<Vec4*>(out_rgba32f)) ? 0 : 3; }

// ---- debug / parity read-back (buffer ids shared with include/strolle_hip.h ST_BUF_*)
static const void* buffer_ptr(CameraBuffers& b, int id, size_t* bytes) {
    const Plane* p = nullptr;
    switch (id) {
        case 0: p = &b.prim_gbuffer_d0[0]; break; case 1: p = &b.prim_gbuffer_d0[1]; break;
        case 2: p = &b.prim_gbuffer_d1[0]; break; case 3: p = &b.prim_gbuffer_d1[1]; break;
        case 4: p = &b.prim_surface_map[0]; break; case 5: p = &b.prim_surface_map[1]; break;
        case 6: p = &b.reprojection_map; break; case 7: p = &b.velocity_map; break;
        case 8: p = &b.di_reservoirs[0]; break; case 9: p = &b.di_reservoirs[1]; break; case 10: p = &b.di_reservoirs[2]; break;
        case 11: p = &b.di_diff_samples; break; case 12: p = &b.di_diff_prev_colors; break; case 13: p = &b.di_diff_curr_colors; break;
        case 14: p = &b.di_diff_moments[0]; break; case 15: p = &b.di_diff_moments[1]; break; case 16: p = &b.di_diff_stash; break;
        case 17: p = &b.di_spec_samples; break;
        case 18: p = &b.gi_d0; break; case 19: p = &b.gi_d1; break; case 20: p = &b.gi_d2; break;
        case 21: p = &b.gi_reservoirs[0]; break; case 22: p = &b.gi_reservoirs[1]; break; case 23: p = &b.gi_reservoirs[2]; break; case 24: p = &b.gi_reservoirs[3]; break;
        case 25: p = &b.gi_diff_samples; break; case 26: p = &b.gi_diff_prev_colors; break; case 27: p = &b.gi_diff_curr_colors; break;
        case 28: p = &b.gi_diff_moments[0]; break; case 29: p = &b.gi_diff_moments[1]; break; case 30: p = &b.gi_diff_stash; break;
        case 31: p = &b.gi_spec_samples; break;
        case 32: p = &b.ref_hits; break; case 33: p = &b.ref_rays; break; case 34: p = &b.ref_colors; break;
        case 35: *bytes = b.dbg_used_memory.size() * 4; return b.dbg_used_memory.data();
        default: return nullptr;
    }
    *bytes = p->size() * sizeof(Vec4);
    return p->data();
}
int or_camera_read_buffer(OrEngine* e, uint64_t h, int buffer_id, void* out, size_t capacity, size_t* written) {
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return 3;
    size_t bytes = 0;
    const void* p = buffer_ptr(it->second->buffers, buffer_id, &bytes);
    if (!p) return 1;
    if (written) *written = bytes;
    if (out) { if (capacity < bytes) return 1; std::memcpy(out, p, bytes); }
    return 0;
}
int or_camera_write_buffer(OrEngine* e, uint64_t h, int buffer_id, const void* data, size_t bytes) {
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return 3;
    size_t have = 0;
    const void* p = buffer_ptr(it->second->buffers, buffer_id, &have);
    if (!p || have != bytes) return 1;
    std::memcpy(const_cast<void*>(p), data, bytes);
    return 0;
}
int or_camera_ray_count(OrEngine* e, uint64_t h, uint64_t* out, int reset) {
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return 3;
    *out = it->second->buffers.ray_count;
    if (reset) it->second->buffers.ray_count = 0;
    return 0;
}
// scene read-back: 0 = bvh (Vec4 stream), 1 = triangles (144 B each), 2 = lights (112 B), 3 = materials (112 B)
int or_debug_read_scene(OrEngine* e, int what, void* out, size_t capacity, size_t* written) {
    const void* p = nullptr; size_t bytes = 0;
    Engine* en = E(e);
    switch (what) {
        case 0: p = en->bvh_buffer.data(); bytes = en->bvh_buffer.size() * sizeof(Vec4); break;
        case 1: p = en->triangles.data(); bytes = en->triangles.size() * sizeof(Triangle); break;
        case 2: p = en->gpu_lights.data(); bytes = en->gpu_lights.size() * sizeof(Light); break;
        case 3: p = en->gpu_materials.data(); bytes = en->gpu_materials.size() * sizeof(Material); break;
        default: return 1;
    }
    if (written) *written = bytes;
    if (out) { if (capacity < bytes) return 1; if (bytes) std::memcpy(out, p, bytes); }
    return 0;
}
int or_debug_world(OrEngine* e, uint32_t* light_count, uint32_t* frame) { *light_count = E(e)->world.light_count; *frame = E(e)->frame; return 0; }

// ---- probes that pin the restatement against the reference's own unit tests
void or_probe_camera_contain(float w, float h, int32_t x, int32_t y, uint32_t* ox, uint32_t* oy) {  // camera.rs:152-175
    Camera c; c.screen = Vec4(w, h, 0, 0);
    UVec2 r = c.contain(IVec2(x, y)); *ox = r.x; *oy = r.y;
}
// in: base_color[4], normal[3], metallic, emissive[3], roughness, reflectance, depth (14 floats); out: same order
void or_probe_gbuffer_roundtrip(const float* in, float* out, float* packed8) {  // gbuffer.rs:132-164
    GBufferEntry g;
    g.base_color = Vec4(in[0], in[1], in[2], in[3]); g.normal = Vec3(in[4], in[5], in[6]); g.metallic = in[7];
    g.emissive = Vec3(in[8], in[9], in[10]); g.roughness = in[11]; g.reflectance = in[12]; g.depth = in[13];
    Vec4 p[2]; g.pack(p);
    if (packed8) std::memcpy(packed8, p, 32);
    GBufferEntry u = GBufferEntry::unpack(p[0], p[1]);
    float o[14] = {u.base_color.x, u.base_color.y, u.base_color.z, u.base_color.w, u.normal.x, u.normal.y, u.normal.z, u.metallic,
                   u.emissive.x, u.emissive.y, u.emissive.z, u.roughness, u.reflectance, u.depth};
    std::memcpy(out, o, sizeof(o));
}
// m, w, pdf, confidence, light_id(bits), light_point[3], is_occluded  -> written at slot idx of buf (2 Vec4 per slot), then read back
void or_probe_di_reservoir_write(float* buf, size_t idx, float m, float w, float pdf, float confidence, uint32_t light_id, const float* lp, int occluded) {
    DiReservoir r; r.m = m; r.w = w; r.sample.pdf = pdf; r.sample.confidence = confidence; r.sample.light_id = light_id;
    r.sample.light_point = Vec3(lp[0], lp[1], lp[2]); r.sample.is_occluded = occluded != 0;
    r.write(reinterpret_cast<Vec4*>(buf), idx);
}
void or_probe_di_reservoir_read(const float* buf, size_t idx, size_t count, float* out9) {  // reservoir/di.rs:132-162
    DiReservoir r = DiReservoir::read(reinterpret_cast<const Vec4*>(buf), idx, count);
    out9[0] = r.m; out9[1] = r.w; out9[2] = r.sample.pdf; out9[3] = r.sample.confidence; out9[4] = b2f(r.sample.light_id);
    out9[5] = r.sample.light_point.x; out9[6] = r.sample.light_point.y; out9[7] = r.sample.light_point.z; out9[8] = r.sample.is_occluded ? 1.0f : 0.0f;
}
void or_probe_reprojection_roundtrip(const float* in3, uint32_t validity, float* out3, uint32_t* out_validity) {  // reprojection.rs:81-96
    Reprojection r; r.prev_x = in3[0]; r.prev_y = in3[1]; r.confidence = in3[2]; r.validity = validity;
    Reprojection d = Reprojection::deserialize(r.serialize());
    out3[0] = d.prev_x; out3[1] = d.prev_y; out3[2] = d.confidence; *out_validity = d.validity;
}
uint32_t or_probe_u32_bytes_roundtrip(uint32_t v) {  // utils/u32_ext.rs:31-34
    uint32_t b[4]; u32_to_bytes(v, b); return u32_from_bytes(b[0], b[1], b[2], b[3]);
}
// op: 0 sin 1 cos 2 acos 3 exp 4 pow(x,y) 5 atan2(x=y_arg, y=x_arg) 6 log2 7 exp2
void or_probe_math(int op, const float* x, const float* y, float* out, size_t n) {
    for (size_t i = 0; i < n; i++) {
        switch (op) {
            case 0: out[i] = stm_sin(x[i]); break; case 1: out[i] = stm_cos(x[i]); break; case 2: out[i] = stm_acos(x[i]); break;
            case 3: out[i] = stm_exp(x[i]); break; case 4: out[i] = stm_pow(x[i], y[i]); break; case 5: out[i] = stm_atan2(x[i], y[i]); break;
            case 6: out[i] = stm_log2(x[i]); break; case 7: out[i] = stm_exp2(x[i]); break; default: out[i] = 0;
        }
    }
}
// GGX normal distribution D(n.h, roughness) and the specular sampler's (direction, pdf) for unit-test integration
float or_probe_ggx_d(float n_dot_h, float roughness) { return ggx_distribution(n_dot_h, roughness); }
// samples the layered BRDF n times for a surface with normal +Y, view direction v3: out = n x {dir.xyz, pdf, radiance.xyz}
void or_probe_brdf_samples(uint32_t seed, float metallic, float roughness, const float* base3, const float* v3, float* out7, size_t n) {
    GBufferEntry g;
    g.base_color = Vec4(base3[0], base3[1], base3[2], 1.0f); g.normal = Vec3(0.0f, 1.0f, 0.0f); g.metallic = metallic; g.emissive = Vec3();
    g.roughness = roughness; g.reflectance = 0.5f; g.depth = 1.0f;
    WhiteNoise wn{seed};
    const Vec3 v(v3[0], v3[1], v3[2]);
    for (size_t i = 0; i < n; i++) {
        const BrdfSample s = layered_brdf_sample(g, wn, v);
        float* o = out7 + 7 * i;
        o[0] = s.dir.x; o[1] = s.dir.y; o[2] = s.dir.z; o[3] = s.pdf; o[4] = s.radiance.x; o[5] = s.radiance.y; o[6] = s.radiance.z;
    }
}
// weighted reservoir sampling (reservoir.rs:24-45): streams `k` candidates with the given weights through Reservoir::update
// `trials` times (one RNG stream) and counts which candidate each trial keeps; also returns the last trial's (m, w)
void or_probe_reservoir_counts(uint32_t seed, const float* weights, uint32_t k, uint32_t trials, uint32_t* counts, float* m_w) {
    WhiteNoise wn{seed};
    for (uint32_t i = 0; i < k; i++) counts[i] = 0;
    for (uint32_t t = 0; t < trials; t++) {
        Reservoir<uint32_t> r; r.sample = 0xffffffffu;
        for (uint32_t i = 0; i < k; i++) r.update(wn, i, weights[i]);
        if (r.sample < k) counts[r.sample]++;
        m_w[0] = r.m; m_w[1] = r.w;
    }
}
uint32_t or_probe_pass_seed(uint64_t base, uint32_t frame, uint32_t pass_id) { return pass_seed(base, frame, pass_id); }
// white noise stream: n samples of sample_int() for (seed, x, y)
void or_probe_white_noise(uint32_t seed, uint32_t x, uint32_t y, uint32_t* out, size_t n) {
    WhiteNoise wn = WhiteNoise::make(seed, UVec2(x, y));
    for (size_t i = 0; i < n; i++) out[i] = wn.sample_int();
}
void or_probe_normal_codec(const float* n3, float* enc2, float* dec3) {
    Vec2 e = normal_encode(Vec3(n3[0], n3[1], n3[2])); enc2[0] = e.x; enc2[1] = e.y;
    Vec3 d = normal_decode(e); dec3[0] = d.x; dec3[1] = d.y; dec3[2] = d.z;
}
// camera ray for pixel (x,y) given transform/projection (col-major) and size: out = origin[3], dir[3]
void or_probe_camera_ray(const ApiCamera* c, uint32_t x, uint32_t y, float* out6) {
    CameraSlot s; s.api = *c;
    Ray r = s.serialize().ray(UVec2(x, y));
    out6[0] = r.origin.x; out6[1] = r.origin.y; out6[2] = r.origin.z; out6[3] = r.dir.x; out6[4] = r.dir.y; out6[5] = r.dir.z;
}
// glam 0.24.2 routines the path depends on (the crate is not under /root/reference), for the float32 numpy cross-check in
// tests/test_oracle_layouts.py: op 0 Mat4::inverse (16 -> 16), 1 Vec3::any_orthonormal_pair (3 -> 6), 2 Mat4::project_point3
// (16 + 3 -> 3), 3 Affine3A::inverse (12 -> 12), 4 Mat4 * Mat4 (16 + 16 -> 16), 5 Vec3::normalize (3 -> 3)
void or_probe_glam(int op, const float* in, float* out) {
    switch (op) {
        case 0: { Mat4 r = inverse(Mat4::from_cols_array(in)); std::memcpy(out, &r, 64); break; }
        case 1: { Vec3 t, b; any_orthonormal_pair(Vec3(in[0], in[1], in[2]), &t, &b); out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = b.x; out[4] = b.y; out[5] = b.z; break; }
        case 2: { Vec3 r = project_point3(Mat4::from_cols_array(in), Vec3(in[16], in[17], in[18])); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        case 3: { Affine3 r = inverse(Affine3::from_12(in)); const Vec3 c[4] = {r.x_axis, r.y_axis, r.z_axis, r.translation}; for (int i = 0; i < 4; i++) { out[3 * i] = c[i].x; out[3 * i + 1] = c[i].y; out[3 * i + 2] = c[i].z; } break; }
        case 4: { Mat4 r = mul(Mat4::from_cols_array(in), Mat4::from_cols_array(in + 16)); std::memcpy(out, &r, 64); break; }
        case 5: { Vec3 r = normalize(Vec3(in[0], in[1], in[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z; break; }
        default: break;
    }
}
int or_num_threads(void) {
    int n = 1;
#ifdef _OPENMP
#pragma omp parallel
    {
#pragma omp single
        n = omp_get_num_threads();
    }
#endif
    return n;
}

}  // extern "C"

extern "C" void or_probe_atmosphere(OrEngine* e, const float* sun_dir3, const float* ray_dir3, float* out3) {
    EngineView v = E(e)->view();
    Vec3 r = v.atmosphere.sample(Vec3(sun_dir3[0], sun_dir3[1], sun_dir3[2]), Vec3(ray_dir3[0], ray_dir3[1], ray_dir3[2]));
    out3[0] = r.x; out3[1] = r.y; out3[2] = r.z;
}
extern "C" void or_probe_sun_dir(OrEngine* e, float* out3) { Vec3 d = E(e)->world.sun_dir(); out3[0] = d.x; out3[1] = d.y; out3[2] = d.z; }

// trace one ray against the engine's scene: out = distance, point[3], normal[3], uv[2], material_id(bits), used_memory
extern "C" void or_probe_trace(OrEngine* e, const float* origin3, const float* dir3, float len, int any_hit, float* out11) {
    EngineView v = E(e)->view();
    Ray r = Ray::make(Vec3(origin3[0], origin3[1], origin3[2]), Vec3(dir3[0], dir3[1], dir3[2]));
    TriangleHit hit = TriangleHit::none();
    if (any_hit) { r = r.with_len(len); hit.distance = len; }
    size_t um = r.traverse(v.scene, any_hit ? ReturnFirst : ReturnClosest, &hit);
    out11[0] = hit.distance; out11[1] = hit.point.x; out11[2] = hit.point.y; out11[3] = hit.point.z;
    out11[4] = hit.normal.x; out11[5] = hit.normal.y; out11[6] = hit.normal.z; out11[7] = hit.uv.x; out11[8] = hit.uv.y;
    out11[9] = b2f(hit.material_id); out11[10] = (float)um;
}
