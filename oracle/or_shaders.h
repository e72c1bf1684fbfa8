// ORACLE — TEST INFRASTRUCTURE ONLY (see or_math.h header).
//
// CPU restatement of the reference's `strolle-shaders` crate: one function per
// SPIR-V entry point (strolle/src/shaders.rs:41-71), executed as a plain loop
// over `global_invocation_id`. Every pass is race-free in the reference (each
// thread writes only texels no other thread reads in the same dispatch), so a
// sequential / OpenMP loop reproduces the GPU result exactly.
//
// Buffers follow strolle/src/camera_controller/buffers.rs: every screen-space
// plane is a W*H array of Vec4 (Rgba32Float), zero-initialised like wgpu does.
#pragma once
#include <vector>

#include "or_gpu.h"

namespace orc {

typedef std::vector<Vec4> Plane;

struct EngineView {  // engine-level bindings (set 0 in the reference)
    SceneView scene;
    LightsView lights;
    World world;
    BlueNoiseTex blue_noise;
    Atmosphere atmosphere;
};

struct CameraBuffers {  // camera_controller/buffers.rs:7-51
    uint32_t width = 0, height = 0;
    Camera curr_camera, prev_camera;
    Plane prim_gbuffer_d0[2], prim_gbuffer_d1[2], prim_surface_map[2];
    Plane reprojection_map, velocity_map;
    Plane di_reservoirs[3];  // 2 Vec4 per pixel
    Plane di_diff_samples, di_diff_prev_colors, di_diff_curr_colors, di_diff_moments[2], di_diff_stash, di_spec_samples;
    Plane gi_d0, gi_d1, gi_d2;
    Plane gi_reservoirs[4];  // 4 Vec4 per pixel
    Plane gi_diff_samples, gi_diff_prev_colors, gi_diff_curr_colors, gi_diff_moments[2], gi_diff_stash, gi_spec_samples;
    Plane ref_hits, ref_rays, ref_colors;
    std::vector<uint32_t> dbg_used_memory;  // oracle/product extra: the heatmap's integer counter
    uint64_t ray_count = 0;                 // rays traced since last reset (closest + any-hit)

    void allocate(uint32_t w, uint32_t h) {
        width = w; height = h;
        size_t n = (size_t)w * h;
        Plane* one[] = {&prim_gbuffer_d0[0], &prim_gbuffer_d0[1], &prim_gbuffer_d1[0], &prim_gbuffer_d1[1], &prim_surface_map[0],
                        &prim_surface_map[1], &reprojection_map, &velocity_map, &di_diff_samples, &di_diff_prev_colors,
                        &di_diff_curr_colors, &di_diff_moments[0], &di_diff_moments[1], &di_diff_stash, &di_spec_samples, &gi_d0,
                        &gi_d1, &gi_d2, &gi_diff_samples, &gi_diff_prev_colors, &gi_diff_curr_colors, &gi_diff_moments[0],
                        &gi_diff_moments[1], &gi_diff_stash, &gi_spec_samples, &ref_colors};
        for (Plane* p : one) p->assign(n, Vec4());
        for (auto& p : di_reservoirs) p.assign(2 * n, Vec4());
        for (auto& p : gi_reservoirs) p.assign(4 * n, Vec4());
        ref_hits.assign(2 * n, Vec4());
        ref_rays.assign(3 * n, Vec4());
        dbg_used_memory.assign(n, 0);
    }
};

// Texture read: in-bounds by construction except where noted; out-of-range reads
// return zero (Vulkan storage-image semantics).
static inline Vec4 tex_read(const Plane& p, const CameraBuffers& b, UVec2 pos) {
    if (pos.x >= b.width || pos.y >= b.height) return Vec4();
    return p[(size_t)pos.y * b.width + pos.x];
}
static inline void tex_write(Plane& p, const CameraBuffers& b, UVec2 pos, Vec4 v) {
    if (pos.x >= b.width || pos.y >= b.height) return;
    p[(size_t)pos.y * b.width + pos.x] = v;
}
static inline GBufferEntry gbuffer_at(const Plane& d0, const Plane& d1, const CameraBuffers& b, UVec2 pos) {
    return GBufferEntry::unpack(tex_read(d0, b, pos), tex_read(d1, b, pos));
}
static inline Surface surface_at(const Plane& p, const CameraBuffers& b, UVec2 pos) { return Surface::from_texel(tex_read(p, b, pos)); }

#define ORC_FOR_EACH_PIXEL(W, H)            \
    _Pragma("omp parallel for schedule(dynamic, 4)") \
    for (int32_t gy_ = 0; gy_ < (int32_t)(H); gy_++) \
        for (uint32_t gx_ = 0; gx_ < (uint32_t)(W); gx_++)

// ---------------------------------------------------------------- bvh_heatmap.rs:3-77
static inline Vec3 heatmap_gradient(float progress) {
    const Vec3 colors[4] = {Vec3(0, 0, 1), Vec3(0, 1, 0), Vec3(1, 0, 0), Vec3(0, 0, 0)};
    const int N = 4;
    if (progress <= 0.0f) return colors[0];
    float step = 1.0f / ((float)N - 1.0f);
    for (int i = 0; i < N - 1; i++) {
        float mn = step * (float)i;
        float mx = step * ((float)i + 1.0f);
        if (progress >= mn && progress <= mx) {
            float rhs = (progress - mn) / step;
            float lhs = 1.0f - rhs;
            return lhs * colors[i] + rhs * colors[i + 1];
        }
    }
    return colors[N - 1];
}
static inline void pass_bvh_heatmap(const EngineView& e, CameraBuffers& b) {
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            size_t used = 0;
            b.curr_camera.ray(pos).trace(e.scene, &used);
            rays++;
            b.dbg_used_memory[(size_t)y * b.width + x] = (uint32_t)used;
            Vec3 c = heatmap_gradient((float)used / 8192.0f);
            tex_write(b.ref_colors, b, pos, Vec4(c, 1.0f));
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- ref_tracing.rs:3-60
static inline void pass_ref_tracing(const EngineView& e, CameraBuffers& b, uint32_t depth) {
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            size_t idx = b.curr_camera.screen_to_idx(pos);
            Ray ray;
            if (depth == 0) ray = b.curr_camera.ray(pos);
            else {
                Vec4 d0 = b.ref_rays[3 * idx], d1 = b.ref_rays[3 * idx + 1];
                if (d1 == Vec4()) continue;
                ray = Ray::make(d0.xyz(), d1.xyz());
            }
            TriangleHit hit = ray.trace(e.scene);
            rays++;
            Vec4 out[2]; hit.pack(out);
            b.ref_hits[2 * idx] = out[0]; b.ref_hits[2 * idx + 1] = out[1];
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- ref_shading.rs:3-177
static inline void pass_ref_shading(const EngineView& e, CameraBuffers& b, uint32_t seed, uint32_t depth) {
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            size_t idx = b.curr_camera.screen_to_idx(pos);
            WhiteNoise wn = WhiteNoise::make(seed, pos);
            if (depth == 255) {
                Vec4 prev = b.curr_camera.is_eq(b.prev_camera) ? tex_read(b.ref_colors, b, pos) : Vec4();
                Vec3 curr = b.ref_rays[3 * idx + 2].xyz();
                tex_write(b.ref_colors, b, pos, prev + Vec4(curr, 1.0f));
                continue;
            }
            Ray ray; Vec3 color, throughput;
            if (depth == 0) { ray = b.curr_camera.ray(pos); color = Vec3(); throughput = Vec3(1, 1, 1); }
            else {
                Vec4 d0 = b.ref_rays[3 * idx], d1 = b.ref_rays[3 * idx + 1], d2 = b.ref_rays[3 * idx + 2];
                ray = Ray::make(d0.xyz(), d1.xyz());
                color = d2.xyz();
                throughput = Vec3(d0.w, d1.w, d2.w);
            }
            TriangleHit t_hit = TriangleHit::unpack(b.ref_hits[2 * idx], b.ref_hits[2 * idx + 1]);
            if (t_hit.is_none()) {
                color += throughput * e.atmosphere.sample(e.world.sun_dir(), ray.dir);
                b.ref_rays[3 * idx] = Vec4(); b.ref_rays[3 * idx + 1] = Vec4(); b.ref_rays[3 * idx + 2] = Vec4(color, 0.0f);
                continue;
            }
            Material material = e.scene.materials[t_hit.material_id];
            if (depth > 0) material.regularize();
            Hit hit;
            hit.point = t_hit.point + t_hit.normal * Hit::NUDGE_OFFSET;
            hit.origin = ray.origin; hit.dir = ray.dir;
            hit.gbuffer.base_color = mat_base_color(material, e.scene.atlas, t_hit.uv);
            hit.gbuffer.normal = t_hit.normal;
            hit.gbuffer.metallic = material.metallic;
            hit.gbuffer.emissive = mat_emissive(material, e.scene.atlas, t_hit.uv);
            hit.gbuffer.roughness = material.roughness;
            hit.gbuffer.reflectance = material.reflectance;
            hit.gbuffer.depth = 0.0f;

            color += throughput * hit.gbuffer.emissive;
            if (e.world.light_count > 0) {
                uint32_t light_id = wn.sample_int() % e.world.light_count;
                float light_pdf = 1.0f / (float)e.world.light_count;
                Light light = e.lights.get(light_id);
                bool occluded = light.ray_wnoise(wn, hit.point).intersect(e.scene);
                rays++;
                if (!occluded) color += throughput * light.radiance(hit).sum() / light_pdf;
            }
            BrdfSample rs = layered_brdf_sample(hit.gbuffer, wn, -hit.dir);
            if (rs.is_invalid()) { b.ref_rays[3 * idx] = Vec4(); b.ref_rays[3 * idx + 1] = Vec4(); continue; }
            Ray rr = Ray::make(hit.point, rs.dir);
            throughput *= dot(rs.dir, hit.gbuffer.normal);
            throughput *= rs.radiance / rs.pdf;
            b.ref_rays[3 * idx] = Vec4(rr.origin, throughput.x);
            b.ref_rays[3 * idx + 1] = Vec4(rr.dir, throughput.y);
            b.ref_rays[3 * idx + 2] = Vec4(color, throughput.z);
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- prim_raster.rs:40-128 restated as primary rays
// CDNA has no rasteriser, so the G-buffer is produced by one closest-hit primary
// ray per pixel (pixel-centre sampling == raster sample position). Fields match
// the fragment shader: front_facing flip == the sign flip Triangle::hit applies
// (normal faces the ray), depth = distance(ray.origin, point), velocity from
// re-projecting the hit point with the instance's previous transform. Alpha-kill
// (`base_color.w < 0.01 && ior == 1.0`) falls out of traversal for Blend
// materials; for Opaque materials alpha is forced to 1 (prepare.rs:141).
// `prev_point = prev_xform * (curr_xform_inv * point)` (prim_raster.rs:21-27) uses the transforms of the instance that owns
// the hit triangle: `triangle_slot[t]` indexes the two tables (the reference passes them as per-draw push constants,
// passes/prim_raster.rs:196-230).
struct InstanceXforms { const Affine3* curr_inv; const Affine3* prev; const uint32_t* triangle_slot; };
static inline void pass_prim_visibility(const EngineView& e, CameraBuffers& b, bool alternate, const InstanceXforms* xf) {
    Plane& g0 = b.prim_gbuffer_d0[alternate], &g1 = b.prim_gbuffer_d1[alternate], &sm = b.prim_surface_map[alternate];
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            Ray ray = b.curr_camera.ray(pos);
            // closest hit with the id of the winning triangle (needed for velocity)
            TriangleHit hit = TriangleHit::none();
            uint32_t hit_triangle = 0;
            ray.traverse(e.scene, ReturnClosest, &hit, &hit_triangle);
            rays++;
            if (hit.is_none()) {  // LoadOp::Clear(TRANSPARENT): prim_raster.rs (host) :160-193
                tex_write(g0, b, pos, Vec4()); tex_write(g1, b, pos, Vec4()); tex_write(sm, b, pos, Vec4()); tex_write(b.velocity_map, b, pos, Vec4());
                continue;
            }
            const Material& material = e.scene.materials[hit.material_id];
            Vec4 base_color = mat_base_color(material, e.scene.atlas, hit.uv);
            Vec2 mr = mat_metallic_roughness(material, e.scene.atlas, hit.uv);
            Vec3 normal = hit.normal;
            float depth = distance(ray.origin, hit.point);
            GBufferEntry g;
            g.base_color = base_color; g.normal = normal; g.metallic = mr.x;
            g.emissive = mat_emissive(material, e.scene.atlas, hit.uv);
            g.roughness = mr.y; g.reflectance = material.reflectance; g.depth = depth;
            Vec4 out[2]; g.pack(out);
            tex_write(g0, b, pos, out[0]); tex_write(g1, b, pos, out[1]);
            Vec2 en = normal_encode(normal);
            tex_write(sm, b, pos, Vec4(en.x, en.y, depth, material.roughness));
            const uint32_t slot = xf->triangle_slot[hit_triangle];
            Vec3 prev_point = transform_point3(xf->prev[slot], transform_point3(xf->curr_inv[slot], hit.point));
            Vec2 velocity = b.curr_camera.clip_to_screen(b.curr_camera.world_to_clip(hit.point)) -
                            b.prev_camera.clip_to_screen(b.prev_camera.world_to_clip(prev_point));
            if (length_squared(velocity) >= 0.001f) tex_write(b.velocity_map, b, pos, Vec4(velocity.x, velocity.y, 0, 0));
            else tex_write(b.velocity_map, b, pos, Vec4());
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- frame_reprojection.rs:6-95
static inline void pass_frame_reprojection(CameraBuffers& b, bool alternate) {
    const Plane& curr_sm = b.prim_surface_map[alternate];
    const Plane& prev_sm = b.prim_surface_map[!alternate];
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        Reprojection rp;
        Surface surface = surface_at(curr_sm, b, pos);
        if (surface.is_sky()) { tex_write(b.reprojection_map, b, pos, rp.serialize()); continue; }
        Vec2 prev_screen_pos = as_vec2(pos) - tex_read(b.velocity_map, b, pos).xy();
        if (b.prev_camera.contains(round(prev_screen_pos))) {
            Surface prev_surface = surface_at(prev_sm, b, as_uvec2(round(prev_screen_pos)));
            float confidence = prev_surface.evaluate_similarity_to(surface);
            if (confidence > 0.0f) { rp.prev_x = prev_screen_pos.x; rp.prev_y = prev_screen_pos.y; rp.confidence = confidence; rp.validity = 0; }
        }
        if (rp.is_some()) {
            IVec2 p[4];
            reprojection_coords(rp.prev_x, rp.prev_y, p);
            for (int i = 0; i < 4; i++) {
                if (!b.curr_camera.contains(p[i])) continue;
                if (surface_at(prev_sm, b, as_uvec2(p[i])).evaluate_similarity_to(surface) >= 0.25f) rp.validity |= (1u << i);
            }
        }
        tex_write(b.reprojection_map, b, pos, rp.serialize());
    }
}

// ---------------------------------------------------------------- di_sampling.rs:3-94
static inline void pass_di_sampling(const EngineView& e, CameraBuffers& b, bool alt, uint32_t seed, Frame frame) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            size_t idx = b.curr_camera.screen_to_idx(pos);
            BlueNoise bn = BlueNoise::make(e.blue_noise, pos, frame);
            WhiteNoise wn = WhiteNoise::make(seed, pos);
            Hit hit = Hit::make(b.curr_camera.ray(pos), gbuffer_at(g0, g1, b, pos));
            if (hit.is_none()) continue;
            EphemeralReservoir res = EphemeralReservoir::build(wn, e.lights, e.world, hit);
            DiReservoir out;
            if (res.m > 0.0f) {
                Ray ray = e.lights.get(res.sample.light_id).ray_bnoise(bn.first_sample(), hit.point);
                bool occluded = ray.intersect(e.scene);
                rays++;
                if (occluded) res.w = 0.0f;
                out.sample.pdf = 0.0f; out.sample.confidence = 0.0f; out.sample.light_id = res.sample.light_id;
                out.sample.light_point = ray.origin; out.sample.is_occluded = occluded;
                out.m = 1.0f; out.w = res.w;
            }
            out.write(b.di_reservoirs[1].data(), idx);
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- di_temporal_resampling.rs:3-112
static inline void pass_di_temporal_resampling(const EngineView& e, CameraBuffers& b, bool alt, uint32_t seed) {
    const Plane &cg0 = b.prim_gbuffer_d0[alt], &cg1 = b.prim_gbuffer_d1[alt], &pg0 = b.prim_gbuffer_d0[!alt], &pg1 = b.prim_gbuffer_d1[!alt];
    const size_t n = (size_t)b.width * b.height;
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 lhs_pos(gx_, (uint32_t)gy_);
        size_t lhs_idx = b.curr_camera.screen_to_idx(lhs_pos);
        WhiteNoise wn = WhiteNoise::make(seed, lhs_pos);
        Hit lhs_hit = Hit::make(b.curr_camera.ray(lhs_pos), gbuffer_at(cg0, cg1, b, lhs_pos));
        if (lhs_hit.is_none()) continue;
        DiReservoir lhs = DiReservoir::read(b.di_reservoirs[1].data(), lhs_idx, n);
        if (!lhs.is_empty()) lhs.sample.pdf = lhs.sample.pdf_curr(e.lights, lhs_hit);
        DiReservoir rhs; Hit rhs_hit; bool rhs_killed = false;
        Reprojection rp = Reprojection::deserialize(tex_read(b.reprojection_map, b, lhs_pos));
        if (rp.is_some()) {
            UVec2 rhs_pos = rp.prev_pos_round();
            rhs = DiReservoir::read(b.di_reservoirs[0].data(), b.curr_camera.screen_to_idx(rhs_pos), n);
            rhs.clamp_m(64.0f);
            if (!rhs.is_empty()) {
                Light rhs_light = e.lights.get(rhs.sample.light_id);
                if (rhs_light.is_slot_killed()) { rhs.w = 0.0f; rhs_killed = true; }
                else if (rhs_light.is_slot_remapped()) rhs.sample.light_id = rhs_light.slot_remapped_to();
                rhs_hit = Hit::make(b.prev_camera.ray(rhs_pos), gbuffer_at(pg0, pg1, b, rhs_pos));
            }
        }
        DiReservoir main; float main_pdf = 0.0f;
        MisResult mis = Mis::di_temporal(e.lights, lhs, lhs_hit, rhs, rhs_hit, rhs_killed).eval();
        if (main.update(wn, lhs.sample, mis.lhs_mis * mis.lhs_pdf * lhs.w)) main_pdf = mis.lhs_pdf;
        if (main.update(wn, rhs.sample, mis.rhs_mis * mis.rhs_pdf * rhs.w)) main_pdf = mis.rhs_pdf;
        main.m = lhs.m + mis.m;
        main.sample.pdf = main_pdf;
        main.sample.confidence = rhs_killed ? 0.0f : 1.0f;
        main.norm_mis(main_pdf);
        main.write(b.di_reservoirs[1].data(), lhs_idx);
    }
}

// ---------------------------------------------------------------- di_spatial_resampling.rs:3-297
// buf_d0/d1/d2 alias di_diff_samples / di_diff_curr_colors / di_diff_stash
// (passes/di_spatial_resampling.rs:24-28).
static inline void pass_di_spatial_pick(const EngineView& e, CameraBuffers& b, bool alt, uint32_t seed, Frame frame) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    Plane &buf_d0 = b.di_diff_samples, &buf_d1 = b.di_diff_curr_colors;
    const size_t n = (size_t)b.width * b.height;
    const uint32_t gw = ((b.width + 7) / 8 / 2) * 8, gh = ((b.height + 7) / 8) * 8;  // dispatch: (size+7)/8/(2,1) groups of 8x8
    ORC_FOR_EACH_PIXEL(gw, gh) {
        UVec2 gid(gx_, (uint32_t)gy_);
        UVec2 lhs_pos = resolve_checkerboard_alt(gid, frame.id / 2);
        if (!b.curr_camera.contains(lhs_pos)) continue;
        size_t lhs_idx = b.curr_camera.screen_to_idx(lhs_pos);
        WhiteNoise wn = WhiteNoise::make(seed, lhs_pos);
        UVec2 buf_pos_a(gid.x * 2, gid.y), buf_pos_b(gid.x * 2 + 1, gid.y);
        Hit lhs_hit = Hit::make(b.curr_camera.ray(lhs_pos), gbuffer_at(g0, g1, b, lhs_pos));
        if (lhs_hit.is_none()) continue;
        DiReservoir lhs = DiReservoir::read(b.di_reservoirs[1].data(), lhs_idx, n);
        DiReservoir rhs; uint32_t rhs_nth = 0; size_t rhs_idx = 0; Hit rhs_hit;
        const uint32_t max_samples = 8; float max_radius = 128.0f;
        while (rhs_nth < max_samples) {
            rhs_nth += 1;
            UVec2 rhs_pos = b.curr_camera.contain(as_ivec2(as_vec2(lhs_pos) + wn.sample_disk() * max_radius));
            if (rhs_pos == lhs_pos) continue;
            rhs_hit = Hit::make(b.curr_camera.ray(rhs_pos), gbuffer_at(g0, g1, b, rhs_pos));
            if (rhs_hit.is_none()) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
            if (fabsf(rhs_hit.gbuffer.depth - lhs_hit.gbuffer.depth) > 0.33f * lhs_hit.gbuffer.depth) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
            if (dot(rhs_hit.gbuffer.normal, lhs_hit.gbuffer.normal) < 0.33f) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
            rhs_idx = b.curr_camera.screen_to_idx(rhs_pos);
            rhs = DiReservoir::read(b.di_reservoirs[1].data(), rhs_idx, n);
            if (!rhs.is_empty()) break;
        }
        if (rhs.is_empty()) { tex_write(buf_d1, b, buf_pos_a, Vec4()); tex_write(buf_d1, b, buf_pos_b, Vec4()); continue; }
        float lhs_rhs_pdf = lhs.sample.pdf_curr(e.lights, rhs_hit);
        float rhs_lhs_pdf = rhs.sample.pdf_curr(e.lights, lhs_hit);
        Ray ray_a, ray_b;
        std::memset((void*)&ray_a, 0, sizeof(Ray)); std::memset((void*)&ray_b, 0, sizeof(Ray));
        if (lhs_rhs_pdf > 0.0f) ray_a = lhs.sample.ray(rhs_hit.point);
        if (rhs_lhs_pdf > 0.0f) ray_b = rhs.sample.ray(lhs_hit.point);
        tex_write(buf_d0, b, buf_pos_a, Vec4(ray_a.origin, ray_a.len));
        Vec2 ea = normal_encode(ray_a.dir);
        tex_write(buf_d1, b, buf_pos_a, Vec4(ea.x, ea.y, b2f((uint32_t)rhs_idx + 1), 0.0f));
        tex_write(buf_d0, b, buf_pos_b, Vec4(ray_b.origin, ray_b.len));
        Vec2 eb = normal_encode(ray_b.dir);
        tex_write(buf_d1, b, buf_pos_b, Vec4(eb.x, eb.y, lhs_rhs_pdf, rhs_lhs_pdf));
    }
}
// shared by DI and GI (di_spatial_resampling.rs:149-209 == gi_spatial_resampling.rs:170-230)
static inline void pass_spatial_trace(const EngineView& e, CameraBuffers& b, const Plane& buf_d0, const Plane& buf_d1, Plane& buf_d2) {
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            Vec4 ray_d0 = tex_read(buf_d0, b, pos), ray_d1 = tex_read(buf_d1, b, pos);
            if (ray_d1 == Vec4()) { tex_write(buf_d2, b, pos, Vec4()); continue; }
            Ray ray = Ray::make(ray_d0.xyz(), normal_decode(ray_d1.xy())).with_len(ray_d0.w);
            bool occluded = ray.intersect(e.scene);
            rays++;
            tex_write(buf_d2, b, pos, Vec4(occluded ? 0.0f : 1.0f, ray_d1.z, ray_d1.w, 0.0f));
        }
    b.ray_count += rays;
}
static inline void pass_di_spatial_sample(CameraBuffers& b, uint32_t seed, Frame frame) {
    const Plane& buf_d2 = b.di_diff_stash;
    const Vec4* in_res = b.di_reservoirs[1].data();
    Vec4* out_res = b.di_reservoirs[2].data();
    const size_t n = (size_t)b.width * b.height;
    const uint32_t gw = ((b.width + 7) / 8 / 2) * 8, gh = ((b.height + 7) / 8) * 8;
    ORC_FOR_EACH_PIXEL(gw, gh) {
        UVec2 gid(gx_, (uint32_t)gy_);
        UVec2 lhs_pos = resolve_checkerboard_alt(gid, frame.id / 2);
        if (!b.curr_camera.contains(lhs_pos)) continue;
        size_t lhs_idx = b.curr_camera.screen_to_idx(lhs_pos);
        WhiteNoise wn = WhiteNoise::make(seed, lhs_pos);
        UVec2 buf_pos_a(gid.x * 2, gid.y), buf_pos_b(gid.x * 2 + 1, gid.y);
        Vec4 d0 = tex_read(buf_d2, b, buf_pos_a), d1 = tex_read(buf_d2, b, buf_pos_b);
        float lhs_rhs_vis = d0.x; uint32_t rhs_idx = f2b(d0.y);
        float rhs_lhs_vis = d1.x, lhs_rhs_pdf = d1.y, rhs_lhs_pdf = d1.z;
        DiReservoir lhs = DiReservoir::read(in_res, lhs_idx, n);
        if (rhs_idx > 0) {
            DiReservoir rhs = DiReservoir::read(in_res, (size_t)rhs_idx - 1, n);
            DiReservoir main; float main_pdf = 0.0f;
            Mis m; m.lhs_m = lhs.m; m.rhs_m = rhs.m; m.rhs_jacobian = 1.0f; m.lhs_lhs_pdf = lhs.sample.pdf;
            m.lhs_rhs_pdf = lhs_rhs_pdf * lhs_rhs_vis; m.rhs_lhs_pdf = rhs_lhs_pdf * rhs_lhs_vis; m.rhs_rhs_pdf = rhs.sample.pdf;
            MisResult mis = m.eval();
            if (main.update(wn, lhs.sample, mis.lhs_mis * mis.lhs_pdf * lhs.w)) main_pdf = mis.lhs_pdf;
            if (main.update(wn, rhs.sample, mis.rhs_mis * mis.rhs_pdf * rhs.w)) { main_pdf = mis.rhs_pdf; main.sample.is_occluded = lhs_rhs_vis == 0.0f; }
            main.m = lhs.m + mis.m;
            main.sample.pdf = main_pdf;
            main.norm_mis(main_pdf);
            main.write(out_res, lhs_idx);
        } else lhs.write(out_res, lhs_idx);
        UVec2 other = resolve_checkerboard(gid, frame.id / 2);
        size_t other_idx = b.curr_camera.screen_to_idx(other);
        // Deviation (documented): the reference copies unchecked; when the width is odd the
        // "other" pixel of the last cell lies outside the row and is skipped here.
        if (b.curr_camera.contains(other)) DiReservoir::read(in_res, other_idx, n).write(out_res, other_idx);
    }
}

// ---------------------------------------------------------------- di_resolving.rs:3-119
static inline void pass_di_resolving(const EngineView& e, CameraBuffers& b, bool alt) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    const size_t n = (size_t)b.width * b.height;
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)b.height; y++)
        for (uint32_t x = 0; x < b.width; x++) {
            UVec2 pos(x, (uint32_t)y);
            size_t idx = b.curr_camera.screen_to_idx(pos);
            Hit hit = Hit::make(b.curr_camera.ray(pos), gbuffer_at(g0, g1, b, pos));
            DiReservoir res = DiReservoir::read(b.di_reservoirs[2].data(), idx, n);
            float confidence; LightRadiance radiance;
            if (hit.is_some()) {
                bool occluded = res.sample.ray(hit.point).intersect(e.scene);
                rays++;
                confidence = (res.sample.is_occluded == occluded) ? res.sample.confidence : 0.0f;
                res.sample.confidence = 1.0f;
                res.sample.is_occluded = occluded;
                if (occluded) radiance = LightRadiance();
                else { radiance = e.lights.get(res.sample.light_id).radiance(hit); radiance.radiance *= res.w; }
            } else {
                confidence = 1.0f;
                radiance.radiance = e.atmosphere.sample(e.world.sun_dir(), hit.dir);
                radiance.diff_brdf = Vec3(1, 1, 1); radiance.spec_brdf = Vec3();
            }
            float diff_brdf = (1.0f - hit.gbuffer.metallic) / PI;
            tex_write(b.di_diff_samples, b, pos, Vec4(radiance.radiance * diff_brdf, confidence));
            tex_write(b.di_spec_samples, b, pos, Vec4(radiance.radiance * radiance.spec_brdf, confidence));
            res.write(b.di_reservoirs[0].data(), idx);
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- gi_reprojection.rs:3-51
static inline void pass_gi_reprojection(CameraBuffers& b, bool alt) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    const size_t n = (size_t)b.width * b.height;
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        size_t idx = b.curr_camera.screen_to_idx(pos);
        Hit hit = Hit::make(b.curr_camera.ray(pos), gbuffer_at(g0, g1, b, pos));
        if (hit.is_none()) continue;
        Reprojection rp = Reprojection::deserialize(tex_read(b.reprojection_map, b, pos));
        GiReservoir res;
        if (rp.is_some()) res = GiReservoir::read(b.gi_reservoirs[0].data(), b.curr_camera.screen_to_idx(rp.prev_pos_round()), n);
        res.confidence = 1.0f;
        res.sample.v1_point = hit.point;
        res.write(b.gi_reservoirs[2].data(), idx);
    }
}

// ---------------------------------------------------------------- gi_sampling_a.rs:3-122
static inline void pass_gi_sampling_a(const EngineView& e, CameraBuffers& b, bool alt, uint32_t seed, Frame frame) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    const size_t n = (size_t)b.width * b.height;
    const uint32_t gw = ((b.width + 7) / 8 / 2) * 8, gh = ((b.height + 7) / 8) * 8;
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)gh; y++)
        for (uint32_t x = 0; x < gw; x++) {
            UVec2 gid(x, (uint32_t)y);
            UVec2 pos = frame.is_gi_tracing() ? resolve_checkerboard(gid, frame.id / 2) : resolve_checkerboard(gid, frame.id);
            if (!b.curr_camera.contains(pos)) continue;
            size_t idx = b.curr_camera.screen_to_idx(pos);
            Ray gi_ray; float gi_ray_pdf;
            if (frame.is_gi_tracing()) {
                WhiteNoise wn = WhiteNoise::make(seed, pos);
                Hit hit = Hit::make(b.curr_camera.ray(pos), gbuffer_at(g0, g1, b, pos));
                if (hit.is_none()) continue;
                BrdfSample s = layered_brdf_sample(hit.gbuffer, wn, -hit.dir);
                gi_ray = Ray::make(hit.point, s.dir);
                gi_ray_pdf = s.pdf;
            } else {
                GiReservoir res = GiReservoir::read(b.gi_reservoirs[2].data(), idx, n);
                if (res.is_empty()) continue;
                gi_ray = Ray::make(res.sample.v1_point, res.sample.dir(res.sample.v1_point));
                gi_ray_pdf = 1.0f;
            }
            TriangleHit gi_hit = gi_ray.trace(e.scene);
            rays++;
            GBufferEntry gg;
            if (gi_hit.is_some()) {
                Material m = e.scene.materials[gi_hit.material_id];
                m.regularize();
                gg.base_color = mat_base_color(m, e.scene.atlas, gi_hit.uv);
                gg.normal = gi_hit.normal; gg.metallic = m.metallic;
                gg.emissive = mat_emissive(m, e.scene.atlas, gi_hit.uv);
                gg.roughness = m.roughness; gg.reflectance = m.reflectance;
                gg.depth = distance(gi_ray.origin, gi_hit.point);
            }
            Vec4 packed[2]; gg.pack(packed);
            // gi_d* are indexed by the half-resolution global_id (gi_sampling_a.rs:117-121)
            tex_write(b.gi_d0, b, gid, Vec4(gi_ray.dir, gi_ray_pdf));
            tex_write(b.gi_d1, b, gid, packed[0]);
            tex_write(b.gi_d2, b, gid, packed[1]);
        }
    b.ray_count += rays;
}

// ---------------------------------------------------------------- gi_sampling_b.rs:3-235
static inline void pass_gi_sampling_b(const EngineView& e, CameraBuffers& b, bool alt, uint32_t seed, Frame frame) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    const size_t n = (size_t)b.width * b.height;
    const uint32_t gw = ((b.width + 7) / 8 / 2) * 8, gh = ((b.height + 7) / 8) * 8;
    uint64_t rays = 0;
    _Pragma("omp parallel for schedule(dynamic, 4) reduction(+ : rays)")
    for (int32_t y = 0; y < (int32_t)gh; y++)
        for (uint32_t x = 0; x < gw; x++) {
            UVec2 gid(x, (uint32_t)y);
            UVec2 pos = frame.is_gi_tracing() ? resolve_checkerboard(gid, frame.id / 2) : resolve_checkerboard(gid, frame.id);
            if (!b.curr_camera.contains(pos)) continue;
            size_t idx = b.curr_camera.screen_to_idx(pos);
            Hit prim_hit = Hit::make(b.curr_camera.ray(pos), gbuffer_at(g0, g1, b, pos));
            if (prim_hit.is_none()) continue;
            Vec4 d0 = tex_read(b.gi_d0, b, gid), d1 = tex_read(b.gi_d1, b, gid), d2 = tex_read(b.gi_d2, b, gid);
            WhiteNoise wn{0}; Hit gi_hit; float gi_ray_pdf;
            if (frame.is_gi_tracing()) {
                wn = WhiteNoise::make(seed, pos);
                gi_hit = Hit::make(Ray::make(prim_hit.point, d0.xyz()), GBufferEntry::unpack(d1, d2));
                gi_ray_pdf = d0.w;
            } else {
                GiReservoir res = GiReservoir::read(b.gi_reservoirs[2].data(), idx, n);
                if (res.is_empty()) continue;
                wn = WhiteNoise{res.sample.rng};
                gi_hit = Hit::make(Ray::make(res.sample.v1_point, d0.xyz()), GBufferEntry::unpack(d1, d2));
                gi_ray_pdf = 1.0f;
            }
            uint32_t rng = wn.state;
            uint32_t light_id; float light_pdf; Vec3 light_rad; Vec3 light_dir;
            if (gi_hit.is_none()) {
                light_id = LIGHT_ID_SKY; light_pdf = 1.0f;
                light_rad = e.atmosphere.sample(e.world.sun_dir(), gi_hit.dir);
            } else {
                float atmosphere_pdf = e.world.sun_altitude <= -1.0f ? 0.0f : 0.25f;
                if (e.world.light_count == 0 || wn.sample() < atmosphere_pdf) {
                    light_id = LIGHT_ID_SKY; light_pdf = atmosphere_pdf;
                    light_dir = wn.sample_hemisphere(gi_hit.gbuffer.normal);
                    light_rad = e.atmosphere.sample(e.world.sun_dir(), light_dir) * dot(gi_hit.gbuffer.normal, light_dir);
                } else {
                    EphemeralReservoir res = EphemeralReservoir::build(wn, e.lights, e.world, gi_hit);
                    if (res.w > 0.0f) {
                        light_id = res.sample.light_id;
                        light_pdf = (1.0f / res.w) * (1.0f - atmosphere_pdf);
                        light_rad = res.sample.light_rad.radiance * (Vec3(1, 1, 1) + res.sample.light_rad.spec_brdf);
                    } else { light_id = 0; light_pdf = 1.0f; light_rad = Vec3(); }
                }
            }
            Vec3 radiance;
            if (light_pdf > 0.0f) {
                float light_vis;
                if (gi_hit.is_some()) {
                    Ray ray = (light_id == LIGHT_ID_SKY) ? Ray::make(gi_hit.point, light_dir) : e.lights.get(light_id).ray_wnoise(wn, gi_hit.point);
                    bool occluded = ray.intersect(e.scene);
                    rays++;
                    light_vis = occluded ? 0.0f : 1.0f;
                } else light_vis = 1.0f;
                radiance = light_rad * light_vis / light_pdf;
            } else radiance = Vec3();
            if (gi_hit.is_some()) { radiance *= gi_hit.gbuffer.base_color.xyz() / PI; radiance += gi_hit.gbuffer.emissive; }
            GiReservoir res;
            if (gi_ray_pdf > 0.0f) {
                Vec3 v1 = prim_hit.point, v2, v2n;
                if (gi_hit.is_some()) { v2 = gi_hit.point; v2n = gi_hit.gbuffer.normal; }
                else { v2 = v1 + gi_hit.dir * SUN_DISTANCE; v2n = -gi_hit.dir; }
                res.sample.pdf = 0.0f; res.sample.rng = rng; res.sample.radiance = radiance;
                res.sample.v1_point = v1; res.sample.v2_point = v2; res.sample.v2_normal = v2n;
                res.m = 1.0f; res.w = 1.0f / gi_ray_pdf;
                res.sample.pdf = res.sample.pdf_at(prim_hit);
            }
            res.write(b.gi_reservoirs[1].data(), idx);
        }
    b.ray_count += rays;
}

// Test-only knob (tests/test_oracle_semantics.py): the most neighbours the GI spatial and preview passes may merge. 8 is the
// reference's constant; 0 leaves the sampling + temporal estimate untouched, which isolates the bias the neighbour merges add.
static uint32_t g_debug_gi_neighbours = 8;

// ---------------------------------------------------------------- gi_temporal_resampling.rs:3-156
static inline void pass_gi_temporal_resampling(CameraBuffers& b, bool alt, uint32_t seed, Frame frame) {
    const Plane &cg0 = b.prim_gbuffer_d0[alt], &cg1 = b.prim_gbuffer_d1[alt], &pg0 = b.prim_gbuffer_d0[!alt], &pg1 = b.prim_gbuffer_d1[!alt];
    const size_t n = (size_t)b.width * b.height;
    Vec4* curr_res = b.gi_reservoirs[1].data();
    const Vec4* prev_res = b.gi_reservoirs[2].data();
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 lhs_pos(gx_, (uint32_t)gy_);
        size_t lhs_idx = b.curr_camera.screen_to_idx(lhs_pos);
        WhiteNoise wn = WhiteNoise::make(seed, lhs_pos);
        Hit lhs_hit = Hit::make(b.curr_camera.ray(lhs_pos), gbuffer_at(cg0, cg1, b, lhs_pos));
        if (lhs_hit.is_none()) { GiReservoir().write(curr_res, lhs_idx); continue; }
        bool got_sample = frame.is_gi_tracing() ? (frame.id % 2 == 0 && got_checkerboard_at(lhs_pos, frame.id / 2))
                                                : got_checkerboard_at(lhs_pos, frame.id);
        GiReservoir lhs = got_sample ? GiReservoir::read(curr_res, lhs_idx, n) : GiReservoir();
        GiReservoir rhs; Hit rhs_hit;
        Reprojection rp = Reprojection::deserialize(tex_read(b.reprojection_map, b, lhs_pos));
        if (rp.is_some()) {
            rhs = GiReservoir::read(prev_res, lhs_idx, n);
            rhs.confidence = 1.0f;
            rhs.clamp_m(128.0f);
            if (frame.is_gi_validation() && !lhs.is_empty() && !rhs.is_empty() && rhs.sample.exists()) {
                if (distance(lhs.sample.radiance, rhs.sample.radiance) > 0.33f) rhs.confidence = 0.0f;
                rhs.sample.radiance = lhs.sample.radiance;
                rhs.sample.v2_point = lhs.sample.v2_point;
                rhs.sample.v2_normal = lhs.sample.v2_normal;
            }
            if (!rhs.is_empty()) {
                UVec2 rhs_pos = rp.prev_pos_round();
                rhs_hit = Hit::make(b.prev_camera.ray(rhs_pos), gbuffer_at(pg0, pg1, b, rhs_pos));
            }
        }
        GiReservoir main; float main_pdf = 0.0f;
        if (frame.is_gi_tracing()) {
            MisResult mis = Mis::gi_temporal(lhs, lhs_hit, rhs, rhs_hit).eval();
            if (main.update(wn, lhs.sample, mis.lhs_mis * mis.lhs_pdf * lhs.w)) main_pdf = mis.lhs_pdf;
            if (main.update(wn, rhs.sample, mis.rhs_mis * mis.rhs_pdf * rhs.w)) main_pdf = mis.rhs_pdf;
            main.m = lhs.m + mis.m;
            main.confidence = 1.0f;
            main.norm_mis(main_pdf);
        } else {
            if (main.merge(wn, rhs, rhs.sample.pdf)) main_pdf = rhs.sample.pdf;
            main.confidence = rhs.confidence;
            main.norm_avg(main_pdf);
        }
        main.sample.pdf = main_pdf;
        main.sample.v1_point = lhs_hit.point;
        main.clamp_w(5.0f);
        main.write(curr_res, lhs_idx);
    }
}

// ---------------------------------------------------------------- gi_spatial_resampling.rs:3-314
static inline void pass_gi_spatial_pick(CameraBuffers& b, bool alt, uint32_t seed, Frame frame) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    Plane &buf_d0 = b.gi_d0, &buf_d1 = b.gi_d1;
    const Vec4* reservoirs = b.gi_reservoirs[1].data();
    const size_t n = (size_t)b.width * b.height;
    const uint32_t gw = ((b.width + 7) / 8 / 2) * 8, gh = ((b.height + 7) / 8) * 8;
    ORC_FOR_EACH_PIXEL(gw, gh) {
        UVec2 gid(gx_, (uint32_t)gy_);
        UVec2 lhs_pos = resolve_checkerboard_alt(gid, frame.id / 2);
        if (!b.curr_camera.contains(lhs_pos)) continue;
        size_t lhs_idx = b.curr_camera.screen_to_idx(lhs_pos);
        WhiteNoise wn = WhiteNoise::make(seed, lhs_pos);
        UVec2 buf_pos_a(gid.x * 2, gid.y), buf_pos_b(gid.x * 2 + 1, gid.y);
        Hit lhs_hit = Hit::make(b.curr_camera.ray(lhs_pos), gbuffer_at(g0, g1, b, lhs_pos));
        GiReservoir lhs = GiReservoir::read(reservoirs, lhs_idx, n);
        if (lhs_hit.is_none() || lhs.is_empty()) { tex_write(buf_d1, b, buf_pos_a, Vec4()); tex_write(buf_d1, b, buf_pos_b, Vec4()); continue; }
        GiReservoir rhs; uint32_t rhs_nth = 0; size_t rhs_idx = 0; Hit rhs_hit; float rhs_jacobian = 0.0f;
        const uint32_t max_samples = g_debug_gi_neighbours < 8u ? g_debug_gi_neighbours : 8u; float max_radius = 128.0f;
        while (rhs_nth < max_samples) {
            rhs_nth += 1;
            UVec2 rhs_pos = b.curr_camera.contain(as_ivec2(as_vec2(lhs_pos) + wn.sample_disk() * max_radius));
            if (rhs_pos == lhs_pos) continue;
            rhs_hit = Hit::make(b.curr_camera.ray(rhs_pos), gbuffer_at(g0, g1, b, rhs_pos));
            if (rhs_hit.is_none()) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
            if (fabsf(rhs_hit.gbuffer.depth - lhs_hit.gbuffer.depth) > 0.33f * lhs_hit.gbuffer.depth) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
            if (dot(rhs_hit.gbuffer.normal, lhs_hit.gbuffer.normal) < 0.33f) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
            rhs_idx = b.curr_camera.screen_to_idx(rhs_pos);
            rhs = GiReservoir::read(reservoirs, rhs_idx, n);
            if (rhs.is_empty()) continue;
            rhs_jacobian = rhs.sample.jacobian(lhs_hit.point);
            if (rhs_jacobian < 1.0f / 10.0f || rhs_jacobian > 10.0f) { rhs.m = 0.0f; continue; }
            rhs_jacobian = clampf(rhs_jacobian, 1.0f / 3.0f, 3.0f);
            break;
        }
        if (rhs.is_empty() || rhs_hit.is_none()) { tex_write(buf_d1, b, buf_pos_a, Vec4()); tex_write(buf_d1, b, buf_pos_b, Vec4()); continue; }
        float lhs_rhs_pdf = lhs.sample.pdf_at(rhs_hit);
        float rhs_lhs_pdf = rhs.sample.pdf_at(lhs_hit);
        Ray ray_a, ray_b;
        std::memset((void*)&ray_a, 0, sizeof(Ray)); std::memset((void*)&ray_b, 0, sizeof(Ray));
        if (lhs_rhs_pdf > 0.0f) ray_a = lhs.sample.ray(rhs_hit.point);
        if (rhs_lhs_pdf > 0.0f) ray_b = rhs.sample.ray(lhs_hit.point);
        tex_write(buf_d0, b, buf_pos_a, Vec4(ray_a.origin, ray_a.len));
        Vec2 ea = normal_encode(ray_a.dir);
        tex_write(buf_d1, b, buf_pos_a, Vec4(ea.x, ea.y, b2f((uint32_t)rhs_idx + 1), rhs_jacobian));
        tex_write(buf_d0, b, buf_pos_b, Vec4(ray_b.origin, ray_b.len));
        Vec2 eb = normal_encode(ray_b.dir);
        tex_write(buf_d1, b, buf_pos_b, Vec4(eb.x, eb.y, lhs_rhs_pdf, rhs_lhs_pdf));
    }
}
static inline void pass_gi_spatial_sample(CameraBuffers& b, uint32_t seed, Frame frame) {
    const Plane& buf_d2 = b.gi_d2;
    const Vec4* in_res = b.gi_reservoirs[1].data();
    Vec4* out_res = b.gi_reservoirs[2].data();
    const size_t n = (size_t)b.width * b.height;
    const uint32_t gw = ((b.width + 7) / 8 / 2) * 8, gh = ((b.height + 7) / 8) * 8;
    ORC_FOR_EACH_PIXEL(gw, gh) {
        UVec2 gid(gx_, (uint32_t)gy_);
        UVec2 pos = resolve_checkerboard_alt(gid, frame.id / 2);
        if (!b.curr_camera.contains(pos)) continue;
        size_t idx = b.curr_camera.screen_to_idx(pos);
        WhiteNoise wn = WhiteNoise::make(seed, pos);
        UVec2 buf_pos_a(gid.x * 2, gid.y), buf_pos_b(gid.x * 2 + 1, gid.y);
        Vec4 d0 = tex_read(buf_d2, b, buf_pos_a), d1 = tex_read(buf_d2, b, buf_pos_b);
        float lhs_rhs_vis = d0.x; uint32_t rhs_idx = f2b(d0.y); float rhs_jacobian = d0.z;
        float rhs_lhs_vis = d1.x, lhs_rhs_pdf = d1.y, rhs_lhs_pdf = d1.z;
        GiReservoir lhs = GiReservoir::read(in_res, idx, n);
        if (rhs_idx > 0) {
            GiReservoir rhs = GiReservoir::read(in_res, (size_t)rhs_idx - 1, n);
            GiReservoir main; float main_pdf = 0.0f;
            Mis m; m.lhs_m = lhs.m; m.rhs_m = rhs.m; m.rhs_jacobian = rhs_jacobian; m.lhs_lhs_pdf = lhs.sample.pdf;
            m.lhs_rhs_pdf = lhs_rhs_pdf * lhs_rhs_vis; m.rhs_lhs_pdf = rhs_lhs_pdf * rhs_lhs_vis; m.rhs_rhs_pdf = rhs.sample.pdf;
            MisResult mis = m.eval();
            if (main.update(wn, lhs.sample, mis.lhs_mis * mis.lhs_pdf * lhs.w)) main_pdf = mis.lhs_pdf;
            if (main.update(wn, rhs.sample, mis.rhs_mis * mis.rhs_pdf * rhs.w * rhs_jacobian)) main_pdf = mis.rhs_pdf;
            main.m = lhs.m + mis.m;
            main.confidence = 1.0f;
            main.sample.pdf = main_pdf;
            main.sample.v1_point = lhs.sample.v1_point;
            main.norm_mis(main_pdf);
            main.clamp_w(5.0f);
            main.write(out_res, idx);
        } else lhs.write(out_res, idx);
        UVec2 other = resolve_checkerboard(gid, frame.id / 2);
        size_t other_idx = b.curr_camera.screen_to_idx(other);
        if (b.curr_camera.contains(other)) GiReservoir::read(in_res, other_idx, n).write(out_res, other_idx);
    }
}

// ---------------------------------------------------------------- gi_preview_resampling.rs:3-138
static inline void pass_gi_preview_resampling(CameraBuffers& b, bool alt, uint32_t seed, uint32_t source, uint32_t nth,
                                              const Plane& in_a, const Plane& in_b, Plane& out) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt], &sm = b.prim_surface_map[alt];
    const size_t n = (size_t)b.width * b.height;
    const Vec4* in = source == 0 ? in_a.data() : in_b.data();
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 center_pos(gx_, (uint32_t)gy_);
        size_t center_idx = b.curr_camera.screen_to_idx(center_pos);
        WhiteNoise wn = WhiteNoise::make(seed, center_pos);
        Hit center_hit = Hit::make(b.curr_camera.ray(center_pos), gbuffer_at(g0, g1, b, center_pos));
        if (center_hit.is_none()) { GiReservoir().write(out.data(), center_idx); continue; }
        GiReservoir main; float main_pdf = 0.0f;
        GiReservoir center = GiReservoir::read(in, center_idx, n);
        if (main.merge(wn, center, center.sample.pdf)) main_pdf = center.sample.pdf;
        uint32_t max_samples = f2u_sat(lerpf(8.0f, 0.0f, main.m / 8.0f));
        if (max_samples > g_debug_gi_neighbours) max_samples = g_debug_gi_neighbours;
        float max_radius = nth == 0 ? 128.0f : 64.0f;
        uint32_t sample_nth = 0;
        bool bail = false;
        while (sample_nth < max_samples) {
            sample_nth += 1;
            UVec2 sample_pos = b.curr_camera.contain(as_ivec2(as_vec2(center_pos) + wn.sample_disk() * max_radius));
            if (sample_pos == center_pos) { bail = true; break; }  // `return` in the reference (:84-86): no write at all
            Surface ss = surface_at(sm, b, sample_pos);
            if (ss.is_sky()) continue;
            if (fabsf(ss.depth - center_hit.gbuffer.depth) > 0.25f * center_hit.gbuffer.depth) continue;
            if (dot(ss.normal, center_hit.gbuffer.normal) < 0.5f) continue;
            GiReservoir s = GiReservoir::read(in, b.curr_camera.screen_to_idx(sample_pos), n);
            if (s.is_empty()) continue;
            float sample_pdf = s.sample.pdf_at(center_hit);
            float sample_jacobian = s.sample.jacobian(center_hit.point);
            if (sample_jacobian < 1.0f / 10.0f || sample_jacobian > 10.0f) continue;
            sample_jacobian = clampf(sample_jacobian, 1.0f / 3.0f, 3.0f);
            if (main.merge(wn, s, sample_pdf * sample_jacobian)) main_pdf = sample_pdf;
        }
        if (bail) continue;
        main.confidence = center.confidence;
        main.sample.pdf = main_pdf;
        main.sample.v1_point = center.sample.v1_point;
        main.norm_avg(main_pdf);
        main.clamp_w(5.0f);
        main.write(out.data(), center_idx);
    }
}

// ---------------------------------------------------------------- gi_resolving.rs:3-67
static inline void pass_gi_resolving(CameraBuffers& b, bool alt, uint32_t source) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    const size_t n = (size_t)b.width * b.height;
    const Vec4* in = source == 0 ? b.gi_reservoirs[1].data() : b.gi_reservoirs[2].data();
    Vec4* out = b.gi_reservoirs[0].data();
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        size_t idx = b.curr_camera.screen_to_idx(pos);
        Hit hit = Hit::make(b.curr_camera.ray(pos), gbuffer_at(g0, g1, b, pos));
        GiReservoir res = GiReservoir::read(out, idx, n);
        float confidence; Vec3 radiance;
        if (hit.is_some()) { confidence = res.confidence; radiance = res.w * res.sample.cosine(hit) * res.sample.radiance; }
        else { confidence = 1.0f; radiance = Vec3(); }
        float diff_brdf = (1.0f - hit.gbuffer.metallic) / PI;
        Vec3 spec_brdf = res.sample.spec_brdf(hit);
        tex_write(b.gi_diff_samples, b, pos, Vec4(radiance * diff_brdf, confidence));
        tex_write(b.gi_spec_samples, b, pos, Vec4(radiance * spec_brdf, confidence));
        GiReservoir::read(in, idx, n).write(out, idx);
    }
}

// ---------------------------------------------------------------- frame_denoising.rs:3-392
static inline float denoise_sample_weight(float center_luma, const Surface& cs, float sample_luma, const Surface& ss, float luma_sigma, float depth_sigma) {
    float luma_weight = fabsf(sqrtf(center_luma) - sqrtf(sample_luma)) * luma_sigma;
    float leeway = cs.depth * depth_sigma;
    float diff = fabsf(ss.depth - cs.depth);
    float depth_weight = diff >= leeway ? 0.0f : 1.0f - diff / leeway;
    float normal_weight = stm_pow64(fmax_(dot(ss.normal, cs.normal), 0.0f));
    return stm_exp(-luma_weight) * depth_weight * normal_weight;
}
static inline void pass_denoise_reproject(CameraBuffers& b, bool alt, const Plane& prev_colors, const Plane& prev_moments, const Plane& samples,
                                          Plane& colors, Plane& moments) {
    const Plane& sm = b.prim_surface_map[alt];
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        if (surface_at(sm, b, pos).is_sky()) { tex_write(colors, b, pos, tex_read(samples, b, pos)); continue; }
        Vec4 sample = tex_read(samples, b, pos);
        float sample_luma = luma(sample.xyz());
        Reprojection rp = Reprojection::deserialize(tex_read(b.reprojection_map, b, pos));
        Vec3 color, moment;
        if (rp.is_some() && sample.w > 0.0f) {
            Vec4 pc = bilinear_reproject(rp, [&](UVec2 p) { return tex_read(prev_colors, b, p); });
            Vec4 pm = bilinear_reproject(rp, [&](UVec2 p) { return tex_read(prev_moments, b, p); });
            Vec3 prev_color = pc.xyz();
            float prev_history = pm.x, prev_m1 = pm.y, prev_m2 = pm.z;
            Vec3 curr_color = sample.xyz();
            float curr_history = fmin_(prev_history + 1.0f, 16.0f);
            float curr_m1 = sample_luma, curr_m2 = sample_luma * sample_luma;
            float alpha = 1.0f / curr_history;
            color = lerp3(prev_color, curr_color, alpha);
            moment = Vec3(curr_history, lerpf(prev_m1, curr_m1, alpha), lerpf(prev_m2, curr_m2, alpha));
        } else {
            color = sample.xyz();
            moment = Vec3(1.0f, sample_luma, sample_luma * sample_luma);
        }
        tex_write(colors, b, pos, Vec4(color, 0.0f));
        tex_write(moments, b, pos, Vec4(moment, 0.0f));
    }
}
static inline void pass_denoise_estimate_variance(CameraBuffers& b, bool alt) {
    const Plane& sm = b.prim_surface_map[alt];
    const Plane &di_colors = b.di_diff_curr_colors, &di_moments = b.di_diff_moments[alt];
    const Plane &gi_colors = b.gi_diff_curr_colors, &gi_moments = b.gi_diff_moments[alt];
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        Surface cs = surface_at(sm, b, pos);
        Vec4 cdi = tex_read(di_colors, b, pos); float cdi_luma = luma(cdi.xyz()); Vec4 cdi_m = tex_read(di_moments, b, pos);
        Vec4 cgi = tex_read(gi_colors, b, pos); float cgi_luma = luma(cgi.xyz()); Vec4 cgi_m = tex_read(gi_moments, b, pos);
        if (cs.is_sky()) { tex_write(b.di_diff_stash, b, pos, cdi); tex_write(b.gi_diff_stash, b, pos, cgi); continue; }
        float di_var, gi_var;
        if (cdi_m.x >= 4.0f) {
            di_var = cdi_m.z - sqr(cdi_m.y);
            gi_var = cgi_m.z - sqr(cgi_m.y);
        } else {
            Vec3 sum_di, sum_gi;
            IVec2 off(-2, -2);
            for (;;) {  // the reference's 29-tap window (frame_denoising.rs:128,180-189) reproduced verbatim
                IVec2 sp = as_ivec2(pos) + off;
                if (b.curr_camera.contains(sp)) {
                    UVec2 up = as_uvec2(sp);
                    Surface ss = surface_at(sm, b, up);
                    if (!ss.is_sky()) {
                        float l = luma(tex_read(di_colors, b, up).xyz());
                        float w = denoise_sample_weight(cdi_luma, cs, l, ss, 1.0f, 0.2f);
                        sum_di += Vec3(l, l * l, 1.0f) * Vec3::splat(w);
                        float lg = luma(tex_read(gi_colors, b, up).xyz());
                        float wg = denoise_sample_weight(cgi_luma, cs, lg, ss, 1.0f, 0.2f);
                        sum_gi += Vec3(lg, lg * lg, 1.0f) * Vec3::splat(wg);
                    }
                }
                off.x += 1;
                if (off.x == 3) { off.x = -3; off.y += 1; if (off.y == 3) break; }
            }
            { float m1 = sum_di.x / sum_di.z, m2 = sum_di.y / sum_di.z; di_var = fabsf(m2 - m1 * m1) * 4.0f; }
            { float m1 = sum_gi.x / sum_gi.z, m2 = sum_gi.y / sum_gi.z; gi_var = fabsf(m2 - m1 * m1) * 4.0f; }
        }
        di_var = fmax_(di_var, 0.0f); gi_var = fmax_(gi_var, 0.0f);
        tex_write(b.di_diff_stash, b, pos, Vec4(cdi.xyz(), di_var));
        tex_write(b.gi_diff_stash, b, pos, Vec4(cgi.xyz(), gi_var));
    }
}
static inline void pass_denoise_wavelet(const EngineView& e, CameraBuffers& b, bool alt, Frame frame, uint32_t stride, float strength,
                                        const Plane& di_in, Plane& di_out, const Plane& gi_in, Plane& gi_out) {
    const Plane& sm = b.prim_surface_map[alt];
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        BlueNoise bn = BlueNoise::make(e.blue_noise, pos, frame);
        Surface cs = surface_at(sm, b, pos);
        Vec4 cdi = tex_read(di_in, b, pos);
        Vec3 cdi_color = cdi.xyz(); float cdi_var = cdi.w; float cdi_luma = luma(cdi_color);
        if (cs.is_sky()) { tex_write(di_out, b, pos, Vec4(cdi_color, cdi_var)); continue; }
        Vec4 cgi = tex_read(gi_in, b, pos);
        Vec3 cgi_color = cgi.xyz(); float cgi_var = cgi.w; float cgi_luma = luma(cgi_color);
        float luma_sigma_di = lerpf(2.5f, 0.5f, sqrtf(cdi_var));
        float depth_sigma_di = 0.33f / strength;
        float luma_sigma_gi = lerpf(1.0f, 0.0f, sqrtf(cgi_var));
        float depth_sigma_gi = 0.33f / strength;
        IVec2 jitter = as_ivec2((bn.second_sample() - 0.5f) * ((float)stride - 1.0f) * 0.5f);
        float sum_di_w = 1.0f; Vec3 sum_di_c = cdi_color; float sum_di_v = cdi_var;
        float sum_gi_w = 1.0f; Vec3 sum_gi_c = cgi_color; float sum_gi_v = cgi_var;
        IVec2 off(-1, -1);
        for (;;) {
            IVec2 sp = as_ivec2(pos) + jitter + off * (int32_t)stride;
            if (b.curr_camera.contains(sp) && !(off == IVec2(0, 0))) {
                UVec2 up = as_uvec2(sp);
                Surface ss = surface_at(sm, b, up);
                if (!ss.is_sky()) {
                    Vec4 sdi = tex_read(di_in, b, up);
                    float w = denoise_sample_weight(cdi_luma, cs, luma(sdi.xyz()), ss, luma_sigma_di, depth_sigma_di);
                    if (w > 0.0f) { sum_di_w += w; sum_di_c += w * sdi.xyz(); sum_di_v += sqr(w) * sdi.w; }
                    Vec4 sgi = tex_read(gi_in, b, up);
                    float wg = denoise_sample_weight(cgi_luma, cs, luma(sgi.xyz()), ss, luma_sigma_gi, depth_sigma_gi);
                    if (wg > 0.0f) { sum_gi_w += wg; sum_gi_c += wg * sgi.xyz(); sum_gi_v += sqr(wg) * sgi.w; }
                }
            }
            off.x += 1;
            if (off.x == 2) { off.x = -1; off.y += 1; if (off.y == 2) break; }
        }
        tex_write(di_out, b, pos, Vec4(sum_di_c / sum_di_w, sum_di_v / (sum_di_w * sum_di_w)));
        tex_write(gi_out, b, pos, Vec4(sum_gi_c / sum_gi_w, sum_gi_v / (sum_gi_w * sum_gi_w)));
    }
}

// ---------------------------------------------------------------- frame_composition.rs:18-82 (as a compute pass into an RGBA32F buffer)
static inline void pass_frame_composition(CameraBuffers& b, bool alt, uint32_t camera_mode, bool denoise_di, bool denoise_gi, Vec4* out) {
    const Plane &g0 = b.prim_gbuffer_d0[alt], &g1 = b.prim_gbuffer_d1[alt];
    const Plane& di_diff = denoise_di ? b.di_diff_curr_colors : b.di_diff_samples;
    const Plane& gi_diff = denoise_gi ? b.gi_diff_curr_colors : b.gi_diff_samples;
    ORC_FOR_EACH_PIXEL(b.width, b.height) {
        UVec2 pos(gx_, (uint32_t)gy_);
        Vec3 color;
        switch (camera_mode) {
            case 0: {
                GBufferEntry g = gbuffer_at(g0, g1, b, pos);
                Vec3 dd = tex_read(di_diff, b, pos).xyz(), ds = tex_read(b.di_spec_samples, b, pos).xyz();
                Vec3 gd = tex_read(gi_diff, b, pos).xyz(), gs = tex_read(b.gi_spec_samples, b, pos).xyz();
                if (g.is_some()) color = g.emissive + (dd + gd) * g.base_color.xyz() + ds + gs;
                else color = dd;
                break;
            }
            case 1: color = tex_read(di_diff, b, pos).xyz(); break;
            case 2: color = tex_read(b.di_spec_samples, b, pos).xyz(); break;
            case 3: color = tex_read(gi_diff, b, pos).xyz(); break;
            case 4: color = tex_read(b.gi_spec_samples, b, pos).xyz(); break;
            case 5: color = tex_read(b.ref_colors, b, pos).xyz(); break;
            case 6: { Vec4 c = tex_read(b.ref_colors, b, pos); color = c.xyz() / c.w; break; }
            default: color = Vec3();
        }
        out[(size_t)pos.y * b.width + pos.x] = Vec4(color, 1.0f);
    }
}

}  // namespace orc
