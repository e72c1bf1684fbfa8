// ORACLE — TEST INFRASTRUCTURE ONLY (see or_math.h header).
//
// CPU restatement of the reference's host crate `strolle` as far as the hot
// path depends on it: world-space triangle baking, the binned-SAH BVH builder
// and its DFS serializer, light/material stores, the camera uniform and the
// per-frame pass order. Citations are to /root/reference/strolle/src.
//
// Stated deviations from the reference (also in DESIGN.md):
//  * instances are iterated in insertion order (the reference iterates a
//    HashMap, i.e. in an unspecified order: instances.rs:80);
//  * every BVH refresh is a fresh build (no subtree reuse by hash,
//    bvh/builder.rs:268-301) — tree shape is the same as the reference's
//    first build;
//  * per-pass seeds come from pass_seed(base, frame, pass_id) instead of
//    `rand::thread_rng()` (camera_controller.rs:189-194).
#pragma once
#include <algorithm>
#include <deque>
#include <map>
#include <memory>
#include <unordered_map>

#include "or_shaders.h"

namespace orc {

// ---------------------------------------------------------------- seeds (NEW seam, shared definition with the product's DESIGN.md)
static inline uint32_t seed_hash(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    uint32_t w = ((v >> ((v >> 28) + 4)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}
static inline uint32_t pass_seed(uint64_t base, uint32_t frame, uint32_t pass_id) {
    return seed_hash((uint32_t)base ^ seed_hash((uint32_t)(base >> 32) ^ seed_hash(frame ^ seed_hash(pass_id))));
}
enum PassId {
    PASS_DI_SAMPLING = 1, PASS_DI_TEMPORAL = 2, PASS_DI_SPATIAL_PICK = 3, PASS_DI_SPATIAL_TRACE = 4, PASS_DI_SPATIAL_SAMPLE = 5,
    PASS_DI_RESOLVING = 6, PASS_GI_REPROJECTION = 7, PASS_GI_SAMPLING_A = 8, PASS_GI_SAMPLING_B = 9, PASS_GI_TEMPORAL = 10,
    PASS_GI_SPATIAL_PICK = 11, PASS_GI_SPATIAL_TRACE = 12, PASS_GI_SPATIAL_SAMPLE = 13, PASS_GI_PREVIEW = 14,
    PASS_REF_TRACING = 100, PASS_REF_SHADING = 200,  // + depth
};

// ---------------------------------------------------------------- utils/bounding_box.rs:5-105
struct BoundingBox {
    Vec3 mn, mx;
    BoundingBox() : mn(Vec3::splat(F32_MAX)), mx(Vec3::splat(-F32_MAX)) {}
    Vec3 extent() const { return mx - mn; }
    float half_area() const { Vec3 e = extent(); return e.x * e.y + e.y * e.z + e.z * e.x; }
    bool is_set() const { return mn.x != F32_MAX; }
    void add(Vec3 p) { mn = vmin(mn, p); mx = vmax(mx, p); }
    void add(const BoundingBox& o) { add(o.mn); add(o.mx); }
};

// ---------------------------------------------------------------- bvh/primitive.rs, bvh/node.rs
struct BvhPrimitive {
    uint32_t triangle_id, material_id; Vec3 center; BoundingBox bounds;
    void kill() { center = Vec3::splat(F32_MAX); }
    bool is_alive() const { return center.x != F32_MAX; }
};
struct BvhNode {
    bool internal = false; BoundingBox bounds; uint32_t prim_start = 0, prim_end = 0; uint32_t left = 0, right = 0;
    float sah_cost() const { return internal ? 0.0f : (float)(prim_end - prim_start) * bounds.half_area(); }
};

// ---------------------------------------------------------------- bvh/builder.rs:15-229
struct BvhBuilder {
    static const int BINS = 12;
    std::vector<BvhNode> nodes;
    std::vector<BvhPrimitive> current;

    struct Plane_ { int axis; float split_at, split_cost; };

    bool find_splitting_plane(uint32_t node_id, Plane_* out) const {
        const BvhNode& node = nodes[node_id];
        uint32_t len = node.prim_end - node.prim_start;
        if (len <= 1) return false;
        const BvhPrimitive* prims = current.data() + node.prim_start;
        BoundingBox centroid_bb;
        for (uint32_t i = 0; i < len; i++) centroid_bb.add(prims[i].center);
        struct Bin { BoundingBox bounds; uint32_t count = 0; };
        Bin bins[3][BINS];
        Vec3 scale = (float)BINS / centroid_bb.extent();
        for (uint32_t i = 0; i < len; i++) {
            Vec3 bin_id = scale * (prims[i].center - centroid_bb.mn);
            uint32_t id[3] = {std::min(f2u_sat(bin_id.x), (uint32_t)BINS - 1), std::min(f2u_sat(bin_id.y), (uint32_t)BINS - 1),
                              std::min(f2u_sat(bin_id.z), (uint32_t)BINS - 1)};
            for (int a = 0; a < 3; a++) { bins[a][id[a]].count += 1; bins[a][id[a]].bounds.add(prims[i].bounds); }
        }
        float left_areas[3][BINS - 1], right_areas[3][BINS - 1];
        uint32_t left_counts[3][BINS - 1], right_counts[3][BINS - 1];
        BoundingBox left_bb[3], right_bb[3];
        uint32_t left_count[3] = {0, 0, 0}, right_count[3] = {0, 0, 0};
        for (int axis = 0; axis < 3; axis++)
            for (int i = 0; i < BINS - 1; i++) {
                const Bin& lb = bins[axis][i];
                left_count[axis] += lb.count; left_counts[axis][i] = left_count[axis];
                if (lb.bounds.is_set()) left_bb[axis].add(lb.bounds);
                left_areas[axis][i] = left_bb[axis].half_area();
                const Bin& rb = bins[axis][BINS - 1 - i];
                right_count[axis] += rb.count; right_counts[axis][BINS - 2 - i] = right_count[axis];
                if (rb.bounds.is_set()) right_bb[axis].add(rb.bounds);
                right_areas[axis][BINS - 2 - i] = right_bb[axis].half_area();
            }
        bool have = false; Plane_ best{0, 0, 0};
        Vec3 scale2 = centroid_bb.extent() / (float)BINS;
        for (int axis = 0; axis < 3; axis++)
            for (int i = 0; i < BINS - 1; i++) {
                float split_cost = (float)left_counts[axis][i] * left_areas[axis][i] + (float)right_counts[axis][i] * right_areas[axis][i];
                bool better = !have || split_cost <= best.split_cost;
                if (better) { have = true; best.axis = axis; best.split_at = centroid_bb.mn[axis] + scale2[axis] * (float)(i + 1); best.split_cost = split_cost; }
            }
        *out = best;
        return have;
    }

    void build() {
        nodes.clear();
        BvhNode root; root.prim_start = 0; root.prim_end = (uint32_t)current.size();
        nodes.push_back(root);
        std::deque<uint32_t> queue{0};
        while (!queue.empty()) {
            uint32_t id = queue.front(); queue.pop_front();
            Plane_ plane;
            if (!find_splitting_plane(id, &plane)) continue;
            if (!(plane.split_cost < nodes[id].sah_cost())) continue;
            // split(): bvh/builder.rs:183-304 (partition with swap-to-back)
            uint32_t start = nodes[id].prim_start, end = nodes[id].prim_end;
            BvhPrimitive* data = current.data() + start;
            int32_t li = 0, ri = (int32_t)(end - start) - 1;
            BoundingBox lb, rb;
            while (li <= ri) {
                BvhPrimitive p = data[li];
                if (p.center[plane.axis] < plane.split_at) { li += 1; lb.add(p.bounds); }
                else { std::swap(data[li], data[ri]); ri -= 1; rb.add(p.bounds); }
            }
            uint32_t pivot = start + (uint32_t)li;
            BvhNode l, r;
            l.bounds = lb; l.prim_start = start; l.prim_end = pivot;
            r.bounds = rb; r.prim_start = pivot; r.prim_end = end;
            uint32_t lid = (uint32_t)nodes.size(); nodes.push_back(l);
            uint32_t rid = (uint32_t)nodes.size(); nodes.push_back(r);
            nodes[id].internal = true; nodes[id].left = lid; nodes[id].right = rid;
            queue.push_back(lid); queue.push_back(rid);
        }
    }

    // Refit (not in the reference; the product's ST_BVH_REFIT policy restated on the tree instead of on the flat stream):
    // the topology and the order of the primitives stay, every node's box becomes the union of what is below it.
    BoundingBox refit(uint32_t id, const std::vector<BvhPrimitive>& all) {
        BvhNode& n = nodes[id];
        BoundingBox box;
        if (n.internal) { box.add(refit(n.left, all)); box.add(refit(n.right, all)); }
        else
            for (uint32_t i = n.prim_start; i < n.prim_end; i++) {
                current[i].bounds = all[current[i].triangle_id].bounds; current[i].center = all[current[i].triangle_id].center;
                box.add(current[i].bounds);
            }
        n.bounds = box;
        return box;
    }

    // bvh/serializer.rs:20-110
    uint32_t serialize(uint32_t id, const std::vector<uint8_t>& material_is_blend, std::vector<Vec4>& buffer) const {
        uint32_t ptr = (uint32_t)buffer.size();
        const BvhNode& n = nodes[id];
        if (n.internal) {
            buffer.push_back(Vec4()); buffer.push_back(Vec4()); buffer.push_back(Vec4()); buffer.push_back(Vec4());
            BoundingBox lb = nodes[n.left].bounds, rb = nodes[n.right].bounds;
            serialize(n.left, material_is_blend, buffer);
            uint32_t right_ptr = serialize(n.right, material_is_blend, buffer);
            buffer[ptr] = Vec4(lb.mn, b2f(0));
            buffer[ptr + 1] = Vec4(lb.mx, b2f(right_ptr));
            buffer[ptr + 2] = Vec4(rb.mn, 0.0f);
            buffer[ptr + 3] = Vec4(rb.mx, 0.0f);
        } else {
            uint32_t len = n.prim_end - n.prim_start;
            for (uint32_t i = 0; i < len; i++) {
                const BvhPrimitive& p = current[n.prim_start + i];
                uint32_t flags = ((i + 1 < len) ? 1u : 0u) | ((material_is_blend[p.material_id] ? 1u : 0u) << 1);
                buffer.push_back(Vec4(b2f(flags), b2f(p.triangle_id), b2f(p.material_id), b2f(1)));
            }
        }
        return ptr;
    }
};

// ---------------------------------------------------------------- utils/allocator.rs:5-61
struct RangeAllocator {
    std::vector<std::pair<size_t, size_t>> slots; bool dirty = false;
    void give(size_t s, size_t e) { if (!slots.empty()) dirty |= s <= slots.back().second; slots.push_back({s, e}); }
    bool take(size_t len, size_t* s, size_t* e) {
        compact();
        for (size_t i = 0; i < slots.size(); i++) {
            size_t sl = slots[i].second - slots[i].first;
            if (sl >= len) {
                if (sl - len > 0) { slots[i].first += len; *s = slots[i].first - len; *e = slots[i].first; }
                else { *s = slots[i].first; *e = slots[i].second; slots.erase(slots.begin() + i); }
                return true;
            }
        }
        return false;
    }
    void compact() {
        if (!dirty || slots.empty()) { dirty = false; return; }
        dirty = false;
        std::stable_sort(slots.begin(), slots.end(), [](auto& a, auto& b) { return a.first < b.first; });
        size_t idx = 0;
        while (idx + 1 < slots.size()) {
            if (slots[idx].second == slots[idx + 1].first) { slots[idx].second = slots[idx + 1].second; slots.erase(slots.begin() + idx + 1); }
            else idx++;
        }
    }
};

// ---------------------------------------------------------------- API-level PODs (same layout as include/strolle_hip.h; declared independently)
struct ApiMeshTriangle { float positions[3][3], normals[3][3], uvs[3][2], tangents[3][4]; };
struct ApiMaterial {
    float base_color[4], emissive[4];
    float perceptual_roughness, metallic, reflectance, ior;
    uint64_t base_color_texture, emissive_texture, metallic_roughness_texture, normal_map_texture;
    uint32_t alpha_mode, _pad;
};
struct ApiLight { uint32_t kind; float position[3]; float radius; float color[3]; float range; float direction[3]; float angle; };
struct ApiCamera { uint32_t mode, denoise, depth, width, height, pos_x, pos_y, _pad; float transform[16], projection[16]; };

// atmosphere/generate_transmittance_lut.rs:32-59 + atmosphere/utils.rs:3-29 (host use: lights.rs:80-95)
static inline Vec3 transmittance_eval(Vec3 pos, Vec3 sun_dir) {
    if (Ray::make(pos, sun_dir).intersect_sphere(Atmosphere::GROUND_RADIUS_MM) > 0.0f) return Vec3();
    float atmosphere_distance = Ray::make(pos, sun_dir).intersect_sphere(Atmosphere::ATMOSPHERE_RADIUS_MM);
    float t = 0.0f; Vec3 transmittance = Vec3::splat(1.0f); float i = 0.0f;
    while (i < 40.0f) {
        float new_t = ((i + 0.3f) / 40.0f) * atmosphere_distance;
        float dt = new_t - t;
        t = new_t;
        Vec3 new_pos = pos + t * sun_dir;
        float altitude_km = (length(new_pos) - Atmosphere::GROUND_RADIUS_MM) * 1000.0f;
        float rayleigh_density = stm_exp(-altitude_km / 8.0f);
        float mie_density = stm_exp(-altitude_km / 1.2f);
        Vec3 rayleigh_scattering = Vec3(5.802f, 13.558f, 33.1f) * rayleigh_density;
        float rayleigh_absorption = 0.0f;  // RAYLEIGH_ABSORPTION_BASE (0.0) * density, folded: 0 * inf must not poison the LUT (DESIGN.md deviation 9)
        float mie_scattering = 3.996f * mie_density;
        float mie_absorption = 4.4f * mie_density;
        Vec3 ozone_absorption = Vec3(0.650f, 1.881f, 0.085f) * fmax_(1.0f - fabsf(altitude_km - 25.0f) / 15.0f, 0.0f);
        Vec3 extinction = rayleigh_scattering + Vec3::splat(rayleigh_absorption) + Vec3::splat(mie_scattering) + Vec3::splat(mie_absorption) + ozone_absorption;
        Vec3 arg = -dt * extinction;
        transmittance *= Vec3(stm_exp(arg.x), stm_exp(arg.y), stm_exp(arg.z));
        i += 1.0f;
    }
    return transmittance;
}

// ---------------------------------------------------------------- atmosphere LUT generation (strolle-shaders/src/atmosphere/*.rs)
// Textures are Rgba16Float in the reference: every texel is rounded through f16 on store (quantize_f16).
struct ScatteringTerms { Vec3 rayleigh; float mie; Vec3 extinction; };
static inline ScatteringTerms eval_scattering(Vec3 pos) {  // atmosphere/utils.rs:3-29
    float altitude_km = (length(pos) - Atmosphere::GROUND_RADIUS_MM) * 1000.0f;
    float rayleigh_density = stm_exp(-altitude_km / 8.0f);
    float mie_density = stm_exp(-altitude_km / 1.2f);
    ScatteringTerms t;
    t.rayleigh = Vec3(5.802f, 13.558f, 33.1f) * rayleigh_density;
    float rayleigh_absorption = 0.0f;  // RAYLEIGH_ABSORPTION_BASE (0.0) * density, folded: 0 * inf must not poison the LUT (DESIGN.md deviation 9)
    t.mie = 3.996f * mie_density;
    float mie_absorption = 4.4f * mie_density;
    Vec3 ozone_absorption = Vec3(0.650f, 1.881f, 0.085f) * fmax_(1.0f - fabsf(altitude_km - 25.0f) / 15.0f, 0.0f);
    t.extinction = t.rayleigh + Vec3::splat(rayleigh_absorption) + Vec3::splat(t.mie) + Vec3::splat(mie_absorption) + ozone_absorption;
    return t;
}
static inline float eval_mie_phase(float cos_theta) {  // atmosphere/utils.rs:31-40
    const float G = 0.8f;
    const float SCALE = 3.0f / (8.0f * PI);
    float num = (1.0f - G * G) * (1.0f + cos_theta * cos_theta);
    float denom = (2.0f + G * G) * stm_pow(1.0f + G * G - 2.0f * G * cos_theta, 1.5f);
    return SCALE * num / denom;
}
static inline float eval_rayleigh_phase(float cos_theta) { const float K = 3.0f / (16.0f * PI); return K * (1.0f + cos_theta * cos_theta); }
static inline Vec3 exp3(Vec3 v) { return Vec3(stm_exp(v.x), stm_exp(v.y), stm_exp(v.z)); }
static inline Vec4 store_f16(Vec3 v) { return Vec4(quantize_f16(v.x), quantize_f16(v.y), quantize_f16(v.z), quantize_f16(1.0f)); }

static inline void generate_transmittance_lut(std::vector<Vec4>& out) {  // generate_transmittance_lut.rs:5-30
    out.assign(256 * 64, Vec4());
    for (uint32_t y = 0; y < 64; y++)
        for (uint32_t x = 0; x < 256; x++) {
            Vec2 uv = Vec2((float)x, (float)y) / Vec2(256.0f, 64.0f);
            float sun_cos_theta = 2.0f * uv.x - 1.0f;
            float sun_theta = stm_acos(clampf(sun_cos_theta, -1.0f, 1.0f));
            float height = lerpf(Atmosphere::GROUND_RADIUS_MM, Atmosphere::ATMOSPHERE_RADIUS_MM, uv.y);
            Vec3 pos(0.0f, height, 0.0f);
            Vec3 sun_dir = normalize(Vec3(0.0f, sun_cos_theta, -stm_sin(sun_theta)));
            out[y * 256 + x] = store_f16(transmittance_eval(pos, sun_dir));
        }
}
static inline void generate_scattering_lut(const LutTex& transmittance, std::vector<Vec4>& out) {  // generate_scattering_lut.rs
    out.assign(32 * 32, Vec4());
    const int SQ = 8;
    for (uint32_t y = 0; y < 32; y++)
        for (uint32_t x = 0; x < 32; x++) {
            Vec2 uv = Vec2((float)x, (float)y) / Vec2(32.0f, 32.0f);
            float sun_cos_theta = 2.0f * uv.x - 1.0f;
            float sun_theta = stm_acos(clampf(sun_cos_theta, -1.0f, 1.0f));
            float height = lerpf(Atmosphere::GROUND_RADIUS_MM, Atmosphere::ATMOSPHERE_RADIUS_MM, fmax_(uv.y, 0.01f));
            Vec3 pos(0.0f, height, 0.0f);
            Vec3 sun_dir = normalize(Vec3(0.0f, sun_cos_theta, -stm_sin(sun_theta)));
            Vec3 lum_total, fms;
            float inv_samples = 1.0f / (float)(SQ * SQ);
            for (int i = 0; i < SQ; i++)
                for (int j = 0; j < SQ; j++) {
                    float theta = PI * ((float)i + 0.5f) / (float)SQ;
                    float phi = stm_acos(clampf(1.0f - 2.0f * ((float)j + 0.5f) / (float)SQ, -1.0f, 1.0f));
                    float cos_phi = stm_cos(phi), sin_phi = stm_sin(phi), cos_theta_d = stm_cos(theta), sin_theta_d = stm_sin(theta);
                    Vec3 ray_dir(sin_phi * sin_theta_d, cos_phi, sin_phi * cos_theta_d);  // spherical_direction
                    float atmosphere_distance = Ray::make(pos, ray_dir).intersect_sphere(Atmosphere::ATMOSPHERE_RADIUS_MM);
                    float ground_distance = Ray::make(pos, ray_dir).intersect_sphere(Atmosphere::GROUND_RADIUS_MM);
                    float t_max = ground_distance > 0.0f ? ground_distance : atmosphere_distance;
                    float cos_theta = dot(ray_dir, sun_dir);
                    float mie_phase_value = eval_mie_phase(cos_theta);
                    float rayleigh_phase_value = eval_rayleigh_phase(-cos_theta);
                    Vec3 lum, lum_factor, transmittance_acc = Vec3::splat(1.0f);
                    float t = 0.0f, step_i = 0.0f;
                    while (step_i < 20.0f) {
                        float new_t = ((step_i + 0.3f) / 20.0f) * t_max;
                        float dt = new_t - t;
                        t = new_t;
                        Vec3 new_pos = pos + t * ray_dir;
                        ScatteringTerms sc = eval_scattering(new_pos);
                        Vec3 sample_transmittance = exp3(-dt * sc.extinction);
                        Vec3 scattering_no_phase = sc.rayleigh + Vec3::splat(sc.mie);
                        Vec3 scattering_f = (scattering_no_phase - scattering_no_phase * sample_transmittance) / sc.extinction;
                        lum_factor += transmittance_acc * scattering_f;
                        Vec3 sun_transmittance = Atmosphere::sample_lut(transmittance, new_pos, sun_dir);
                        Vec3 rayleigh_in = sc.rayleigh * rayleigh_phase_value;
                        float mie_in = sc.mie * mie_phase_value;
                        Vec3 in_scattering = (rayleigh_in + Vec3::splat(mie_in)) * sun_transmittance;
                        Vec3 scattering_integral = (in_scattering - in_scattering * sample_transmittance) / sc.extinction;
                        lum += scattering_integral * transmittance_acc;
                        transmittance_acc *= sample_transmittance;
                        step_i += 1.0f;
                    }
                    if (ground_distance > 0.0f) {
                        Vec3 hit_pos = pos + ground_distance * ray_dir;
                        if (dot(pos, sun_dir) > 0.0f) {
                            hit_pos = normalize(hit_pos) * Atmosphere::GROUND_RADIUS_MM;
                            lum += transmittance_acc * Vec3::splat(0.25f) * Atmosphere::sample_lut(transmittance, hit_pos, sun_dir);
                        }
                    }
                    fms += lum_factor * inv_samples;
                    lum_total += lum * inv_samples;
                }
            Vec3 out_val = lum_total / (Vec3::splat(1.0f) - fms);
            out[y * 32 + x] = store_f16(out_val);
        }
}
static inline void generate_sky_lut(const LutTex& transmittance, const LutTex& scattering, float sun_altitude, std::vector<Vec4>& out) {  // generate_sky_lut.rs
    out.assign(256 * 256, Vec4());
    _Pragma("omp parallel for schedule(dynamic, 4)")
    for (int32_t y = 0; y < 256; y++)
        for (uint32_t x = 0; x < 256; x++) {
            Vec2 uv = Vec2((float)x, (float)y) / Vec2(256.0f, 256.0f);
            float azimuth = (uv.x - 0.5f) * 2.0f * PI;
            float v;
            if (uv.y < 0.5f) { float coord = 1.0f - 2.0f * uv.y; v = -coord * coord; }
            else { float coord = uv.y * 2.0f - 1.0f; v = coord * coord; }
            float height = length(Atmosphere::view_pos());
            float th = sqr(height) - sqr(Atmosphere::GROUND_RADIUS_MM);
            th = sqrtf(th) / height;
            float horizon = stm_acos(clampf(th, -1.0f, 1.0f)) - 0.5f * PI;
            float altitude = v * 0.5f * PI - horizon;
            Vec3 ray_dir(stm_cos(altitude) * stm_sin(azimuth), stm_sin(altitude), -stm_cos(altitude) * stm_cos(azimuth));
            float sa = fmodf(sun_altitude, 2.0f * PI);
            Vec3 sun_dir = sa < 0.5f * PI ? Vec3(0.0f, stm_sin(sa), -stm_cos(sa)) : Vec3(0.0f, stm_sin(sa), stm_cos(sa));
            Vec3 pos = Atmosphere::view_pos();
            float atmosphere_distance = Ray::make(pos, ray_dir).intersect_sphere(Atmosphere::ATMOSPHERE_RADIUS_MM);
            float ground_distance = Ray::make(pos, ray_dir).intersect_sphere(Atmosphere::GROUND_RADIUS_MM);
            float t_max = ground_distance < 0.0f ? atmosphere_distance : ground_distance;
            float cos_theta = dot(ray_dir, sun_dir);
            float mie_phase_value = eval_mie_phase(cos_theta);
            float rayleigh_phase_value = eval_rayleigh_phase(-cos_theta);
            Vec3 lum, transmittance_acc = Vec3::splat(1.0f);
            float t = 0.0f, i = 0.0f;
            while (i < 32.0f) {
                float new_t = ((i + 0.3f) / 32.0f) * t_max;
                float dt = new_t - t;
                t = new_t;
                Vec3 new_pos = pos + t * ray_dir;
                ScatteringTerms sc = eval_scattering(new_pos);
                Vec3 sample_transmittance = exp3(-dt * sc.extinction);
                Vec3 sun_transmittance = Atmosphere::sample_lut(transmittance, new_pos, sun_dir);
                Vec3 psi_ms = Atmosphere::sample_lut(scattering, new_pos, sun_dir);
                Vec3 rayleigh_in = sc.rayleigh * (rayleigh_phase_value * sun_transmittance + psi_ms);
                Vec3 mie_in = sc.mie * (mie_phase_value * sun_transmittance + psi_ms);
                Vec3 in_scattering = rayleigh_in + mie_in;
                Vec3 scattering_integral = (in_scattering - in_scattering * sample_transmittance) / sc.extinction;
                lum += scattering_integral * transmittance_acc;
                transmittance_acc *= sample_transmittance;
                i += 1.0f;
            }
            out[(size_t)y * 256 + x] = store_f16(lum);
        }
}

// ---------------------------------------------------------------- Engine (lib.rs:105-395) + CameraController (camera_controller.rs)
struct CameraSlot {
    ApiCamera api; CameraBuffers buffers; Frame frame{0};
    Camera serialize() const {  // camera.rs:50-66
        Mat4 transform = Mat4::from_cols_array(api.transform), projection = Mat4::from_cols_array(api.projection);
        Camera c;
        c.projection_view = mul(projection, inverse(transform));
        c.ndc_to_world = mul(transform, inverse(projection));
        c.origin = Vec4(transform.c[3].xyz(), 0.0f);  // to_scale_rotation_translation().2 == w_axis.xyz
        c.screen = Vec4((float)api.width, (float)api.height, 0.0f, 0.0f);
        return c;
    }
};

struct Engine {
    // meshes.rs / materials.rs / instances.rs / triangles.rs / lights.rs (insertion-ordered)
    std::map<uint64_t, std::vector<ApiMeshTriangle>> meshes;
    struct InstanceEntry { uint64_t id, mesh, material; Affine3 xform, xform_inv, prev_xform; bool dirty; uint32_t xslot; };
    std::vector<InstanceEntry> instances;
    // per-instance transform tables for primary visibility's prev_point (prim_raster.rs push constants), indexed by a
    // stable slot; triangle_slot[t] = slot of the instance that owns triangle t
    std::vector<Affine3> xf_curr_inv, xf_prev; std::vector<uint32_t> xslot_free, triangle_slot;
    struct IndexedInstance { size_t start, end; };
    std::map<uint64_t, IndexedInstance> tri_index;
    RangeAllocator tri_alloc;
    std::vector<Triangle> triangles;
    std::vector<BvhPrimitive> prims_all;
    bool instances_dirty = false;

    std::vector<ApiMaterial> materials; std::map<uint64_t, uint32_t> material_index; RangeAllocator material_alloc;
    std::vector<Material> gpu_materials; bool materials_dirty = false;

    // lights.rs:12-172
    std::vector<Light> light_buffer; std::map<int64_t, uint32_t> light_index;  // key -1 == sun
    std::vector<int64_t> created, updated; std::map<int64_t, uint32_t> remapped; std::vector<uint32_t> killed;
    uint32_t next_light_id = 1;
    std::vector<Light> gpu_lights;  // what the device sees this frame

    float sun_azimuth = 0.0f, sun_altitude = 0.35f; bool sun_dirty = true;  // sun.rs:7-14
    World world{0, 0, 0};
    uint32_t frame = 1;  // lib.rs:152
    uint64_t base_seed = 0;
    BvhBuilder bvh; std::vector<Vec4> bvh_buffer;
    bool bvh_refit_mode = false, built_with_refit_mode = false; uint64_t rebuilds = 0, refits = 0;
    std::vector<std::pair<uint32_t, uint32_t>> built_leaves; std::vector<uint8_t> built_blend;
    std::vector<uint8_t> blue_noise;  // 256*256*4
    std::vector<Vec4> transmittance_lut, scattering_lut, sky_lut;  // passes/atmosphere.rs:78-110
    bool atmosphere_initialized = false; bool sky_known = false; float known_sun_altitude = 0.0f;
    std::vector<uint8_t> atlas; uint32_t atlas_w = 0, atlas_h = 0;
    // images: one linear RGBA8 atlas of the reference's extent (8192 x 8192, images.rs:28-29), grown in 256-row steps.
    // (The reference places rectangles with `guillotiere` 0.6.2, images.rs:54-127 — a crate that is not under
    //  /root/reference; rectangle placement is therefore this project's own policy, restated here independently of the
    //  product's st_atlas.h: shelves stacked bottom to top; a request goes to the closed shelf of least height that has a
    //  wide enough free span (leftmost such span), else to the top shelf, which may still grow taller, else to a new shelf
    //  on top; released spans merge with free neighbours and empty top shelves disappear. Only the *sampling* result, which
    //  does not depend on placement up to clamp-at-rect-border bleeding, is comparable with the reference.)
    struct ImageRec { uint32_t x, y, w, h; };
    struct AtlasPacker {
        struct Row { uint32_t y, h; std::map<uint32_t, uint32_t> holes; /* x0 -> x1 */ };
        std::vector<Row> rows;
        static constexpr uint32_t W = 8192, H = 8192;
        static bool hole_for(const Row& r, uint32_t w, uint32_t* x0) {
            for (auto& kv : r.holes) if (kv.second - kv.first >= w) { *x0 = kv.first; return true; }
            return false;
        }
        static void carve(Row& r, uint32_t x0, uint32_t w) {
            const uint32_t x1 = r.holes[x0];
            r.holes.erase(x0);
            if (x0 + w < x1) r.holes[x0 + w] = x1;
        }
        bool place(uint32_t w, uint32_t h, ImageRec* out) {
            if (!w || !h || w > W || h > H) return false;
            Row* chosen = nullptr; uint32_t x0 = 0, x = 0;
            for (size_t i = 0; i + 1 < rows.size(); i++)
                if (rows[i].h >= h && (!chosen || rows[i].h < chosen->h) && hole_for(rows[i], w, &x)) { chosen = &rows[i]; x0 = x; }
            if (!chosen && !rows.empty() && hole_for(rows.back(), w, &x) && rows.back().y + std::max(rows.back().h, h) <= H) {
                chosen = &rows.back(); x0 = x; chosen->h = std::max(chosen->h, h);
            }
            if (!chosen) {
                const uint64_t y = rows.empty() ? 0 : (uint64_t)rows.back().y + rows.back().h;
                if (y + h > H) return false;
                rows.push_back(Row{(uint32_t)y, h, {{0u, W}}});
                chosen = &rows.back(); x0 = 0;
            }
            carve(*chosen, x0, w);
            *out = ImageRec{x0, chosen->y, w, h};
            return true;
        }
        void give_back(const ImageRec& r) {
            for (Row& row : rows) {
                if (row.y != r.y) continue;
                uint32_t a = r.x, b = r.x + r.w;
                auto next = row.holes.find(b);
                if (next != row.holes.end()) { b = next->second; row.holes.erase(next); }
                for (auto it = row.holes.begin(); it != row.holes.end(); ++it)
                    if (it->second == a) { a = it->first; row.holes.erase(it); break; }
                row.holes[a] = b;
                break;
            }
            while (!rows.empty() && rows.back().holes.size() == 1 && rows.back().holes.begin()->first == 0 && rows.back().holes.begin()->second == W) rows.pop_back();
        }
    };
    std::map<uint64_t, ImageRec> images; AtlasPacker packer;
    bool insert_image(uint64_t id, uint32_t w, uint32_t h, const uint8_t* rgba) {
        if (w > AtlasPacker::W) return false;
        ImageRec rec;
        auto it = images.find(id);
        if (it != images.end() && it->second.w == w && it->second.h == h) rec = it->second;  // images.rs:61-63
        else {
            if (it != images.end()) { packer.give_back(it->second); images.erase(it); materials_dirty = true; }  // images.rs:64-66
            if (!packer.place(w, h, &rec)) return false;  // images.rs:71-79: warn and drop
        }
        if (atlas_w == 0) atlas_w = AtlasPacker::W;
        if (rec.y + h > atlas_h) { atlas_h = (rec.y + h + 255u) & ~255u; atlas.resize((size_t)atlas_w * atlas_h * 4, 0); }
        for (uint32_t y = 0; y < h; y++) std::memcpy(&atlas[((size_t)(rec.y + y) * atlas_w + rec.x) * 4], rgba + (size_t)y * w * 4, (size_t)w * 4);
        images[id] = rec;
        materials_dirty = true;
        return true;
    }
    void remove_image(uint64_t id) {  // images.rs:107-113
        auto it = images.find(id);
        if (it == images.end()) return;
        packer.give_back(it->second);
        images.erase(it);
        materials_dirty = true;
    }
    std::map<uint64_t, std::unique_ptr<CameraSlot>> cameras; uint64_t next_camera = 0;

    Engine() {
        light_buffer.push_back(Light::sun(Vec3(), Vec3()));
        light_index[-1] = 0;
        blue_noise.assign(256 * 256 * 4, 0);
        transmittance_lut.assign(256 * 64, Vec4());
        sky_lut.assign(256 * 256, Vec4());
    }

    // -------- materials (materials.rs:33-96, material.rs:29-50)
    void insert_material(uint64_t h, const ApiMaterial& m) {
        auto it = material_index.find(h);
        if (it != material_index.end()) materials[it->second] = m;
        else {
            size_t s, e; uint32_t id;
            if (material_alloc.take(1, &s, &e)) { id = (uint32_t)s; /* reference quirk: the slot keeps its old contents (materials.rs:48-50) */ }
            else { materials.push_back(m); id = (uint32_t)materials.size() - 1; }
            material_index[h] = id;
        }
        materials_dirty = true;
    }
    void remove_material(uint64_t h) {
        auto it = material_index.find(h);
        if (it == material_index.end()) return;
        material_alloc.give(it->second, it->second);  // `give(id..id)`: an empty range (materials.rs:74) — never reused
        material_index.erase(it);
        materials_dirty = true;
    }
    Vec4 lookup_image(uint64_t h) const {
        if (!h || atlas_w == 0) return Vec4();
        auto it = images.find(h);
        if (it == images.end()) return Vec4();
        const ImageRec& r = it->second;
        return Vec4((float)r.x / (float)atlas_w, (float)r.y / (float)atlas_h, (float)r.w / (float)atlas_w, (float)r.h / (float)atlas_h);
    }
    void refresh_materials() {
        gpu_materials.clear();
        for (auto& m : materials) {
            Material g;
            g.base_color = Vec4(m.base_color[0], m.base_color[1], m.base_color[2], m.base_color[3]);
            g.base_color_texture = lookup_image(m.base_color_texture);
            g.emissive = Vec4(m.emissive[0], m.emissive[1], m.emissive[2], m.emissive[3]);
            g.emissive_texture = lookup_image(m.emissive_texture);
            g.roughness = stm_pow2(m.perceptual_roughness);
            g.metallic = m.metallic; g.reflectance = m.reflectance; g.ior = m.ior;
            g.metallic_roughness_texture = lookup_image(m.metallic_roughness_texture);
            g.normal_map_texture = lookup_image(m.normal_map_texture);
            gpu_materials.push_back(g);
        }
    }

    // -------- lights
    static Light serialize_light(const ApiLight& l) {  // light.rs:25-79
        Light g{};
        g.d0 = Vec4(l.position[0], l.position[1], l.position[2], l.radius);
        g.d1 = Vec4(l.color[0], l.color[1], l.color[2], l.range);
        if (l.kind == 0) g.d2 = Vec4(b2f(Light::TYPE_POINT), 0, 0, 0);
        else { Vec2 d = normal_encode(Vec3(l.direction[0], l.direction[1], l.direction[2])); g.d2 = Vec4(b2f(Light::TYPE_SPOT), d.x, d.y, l.angle); }
        return g;
    }
    static void push_unique(std::vector<int64_t>& v, int64_t k) { if (std::find(v.begin(), v.end(), k) == v.end()) v.push_back(k); }
    void update_light(uint32_t idx, int64_t key, Light nw) {
        Light old = light_buffer[idx];
        nw.prev_d0 = old.d0; nw.prev_d1 = old.d1; nw.prev_d2 = old.d2;
        push_unique(updated, key);
        light_buffer[idx] = nw;
    }
    void insert_light(uint64_t h, const ApiLight& l) {
        Light item = serialize_light(l);
        int64_t key = (int64_t)h;
        auto it = light_index.find(key);
        if (it != light_index.end()) { update_light(it->second, key, item); return; }
        if (next_light_id < light_buffer.size()) { light_buffer[next_light_id] = item; light_index[key] = next_light_id; }
        else { light_index[key] = (uint32_t)light_buffer.size(); light_buffer.push_back(item); }
        push_unique(created, key);
        next_light_id += 1;
    }
    void remove_light(uint64_t h) {
        int64_t key = (int64_t)h;
        auto it = light_index.find(key);
        if (it == light_index.end()) return;
        uint32_t id = it->second;
        light_index.erase(it);
        light_buffer.erase(light_buffer.begin() + id);
        light_buffer.push_back(Light{});
        created.erase(std::remove(created.begin(), created.end(), key), created.end());
        updated.erase(std::remove(updated.begin(), updated.end(), key), updated.end());
        remapped.erase(key);
        if (std::find(killed.begin(), killed.end(), id) == killed.end()) killed.push_back(id);
        next_light_id -= 1;
        for (auto& kv : light_index)
            if (kv.second > id) { if (!remapped.count(kv.first)) remapped[kv.first] = kv.second; kv.second -= 1; }
    }
    void flush_lights() {  // lights.rs:128-154
        for (uint32_t id : killed) light_buffer[id].d3.x = b2f(0xcafebabeu);
        for (auto& kv : remapped) light_buffer[kv.second].d3.x = b2f(light_index[kv.first] + 1);
        gpu_lights = light_buffer;
        for (int64_t k : created) { Light& l = light_buffer[light_index[k]]; l.prev_d0 = l.d0; l.prev_d1 = l.d1; l.prev_d2 = l.d2; }
        for (int64_t k : updated) { Light& l = light_buffer[light_index[k]]; l.prev_d0 = l.d0; l.prev_d1 = l.d1; l.prev_d2 = l.d2; }
        for (uint32_t id : killed) light_buffer[id].d3.x = b2f(0);
        for (auto& kv : remapped) light_buffer[kv.second].d3.x = b2f(0);
        created.clear(); updated.clear(); remapped.clear(); killed.clear();
    }

    // -------- instances / triangles (instances.rs:29-139, triangles.rs:37-177, mesh_triangle.rs:47-86, triangle.rs:16-37)
    void insert_instance(uint64_t h, uint64_t mesh, uint64_t material, const float xf[12]) {
        Affine3 x = Affine3::from_12(xf);
        for (auto& e : instances)
            if (e.id == h) { e.prev_xform = e.xform; e.mesh = mesh; e.material = material; e.xform = x; e.xform_inv = inverse(x); e.dirty = true; instances_dirty = true; return; }
        uint32_t xslot;
        if (!xslot_free.empty()) { xslot = xslot_free.back(); xslot_free.pop_back(); }
        else { xslot = (uint32_t)xf_prev.size(); xf_prev.push_back(Affine3()); xf_curr_inv.push_back(Affine3()); }
        instances.push_back({h, mesh, material, x, inverse(x), x, true, xslot});
        instances_dirty = true;
    }
    void remove_triangles(uint64_t h) {
        auto it = tri_index.find(h);
        if (it == tri_index.end()) return;
        tri_alloc.give(it->second.start, it->second.end);
        for (size_t i = it->second.start; i < it->second.end; i++) prims_all[i].kill();
        tri_index.erase(it);
    }
    void remove_instance(uint64_t h) {
        for (size_t i = 0; i < instances.size(); i++)
            if (instances[i].id == h) { xslot_free.push_back(instances[i].xslot); instances.erase(instances.begin() + i); instances_dirty = true; break; }
        remove_triangles(h);
    }
    static Triangle bake(const ApiMeshTriangle& t, const Affine3& xform, const Affine3& xform_inv, Vec3* center, BoundingBox* bounds) {
        Vec3 pos[3], nrm[3]; Vec4 tan[3];
        Mat4 mat = transpose(mat4_from_affine(xform_inv));
        float sign = std::signbit(mat3_determinant(xform)) ? -1.0f : 1.0f;
        for (int i = 0; i < 3; i++) {
            pos[i] = transform_point3(xform, Vec3(t.positions[i][0], t.positions[i][1], t.positions[i][2]));
            nrm[i] = normalize(transform_vector3(mat, Vec3(t.normals[i][0], t.normals[i][1], t.normals[i][2])));
            Vec3 tg = normalize(mat3_mul(xform, Vec3(t.tangents[i][0], t.tangents[i][1], t.tangents[i][2])));
            tan[i] = Vec4(tg, t.tangents[i][3] * sign);
        }
        *center = (((Vec3() + pos[0]) + pos[1]) + pos[2]) / 3.0f;  // iter().sum::<Vec3>() folds from ZERO (triangle.rs:17)
        *bounds = BoundingBox(); bounds->add(pos[0]); bounds->add(pos[1]); bounds->add(pos[2]);
        Triangle g;
        g.d0 = Vec4(pos[0], t.uvs[0][0]); g.d1 = Vec4(nrm[0], t.uvs[0][1]); g.d2 = tan[0];
        g.d3 = Vec4(pos[1], t.uvs[1][0]); g.d4 = Vec4(nrm[1], t.uvs[1][1]); g.d5 = tan[1];
        g.d6 = Vec4(pos[2], t.uvs[2][0]); g.d7 = Vec4(nrm[2], t.uvs[2][1]); g.d8 = tan[2];
        return g;
    }
    bool refresh_instances() {
        if (!instances_dirty) return false;
        instances_dirty = false;
        for (auto& e : instances) {
            if (!e.dirty) continue;
            e.dirty = false;
            auto mit = meshes.find(e.mesh);
            auto mat = material_index.find(e.material);
            if (mit == meshes.end() || mat == material_index.end()) { e.dirty = true; instances_dirty = true; continue; }
            const auto& mesh = mit->second;
            auto ti = tri_index.find(e.id);
            if (ti != tri_index.end() && (ti->second.end - ti->second.start) != mesh.size()) { remove_triangles(e.id); ti = tri_index.end(); }
            size_t start, end;
            if (ti != tri_index.end()) { start = ti->second.start; end = ti->second.end; }
            else if (tri_alloc.take(mesh.size(), &start, &end)) {}
            else { start = triangles.size(); end = start + mesh.size(); triangles.resize(end); prims_all.resize(end); triangle_slot.resize(end, 0u); }
            for (size_t i = 0; i < mesh.size(); i++) {
                BvhPrimitive p; p.triangle_id = (uint32_t)(start + i); p.material_id = mat->second;
                triangles[start + i] = bake(mesh[i], e.xform, e.xform_inv, &p.center, &p.bounds);
                prims_all[start + i] = p;
                triangle_slot[start + i] = e.xslot;
            }
            tri_index[e.id] = {start, end};
        }
        return true;
    }

    // -------- tick (lib.rs:301-395)
    void tick() {
        if (materials_dirty) { materials_dirty = false; refresh_materials(); }
        if (refresh_instances()) {
            for (const auto& e : instances) { xf_curr_inv[e.xslot] = e.xform_inv; xf_prev[e.xslot] = e.prev_xform; }
            std::vector<uint8_t> blend(materials.size() + 1, 0);
            for (size_t i = 0; i < materials.size(); i++) blend[i] = materials[i].alpha_mode == 1;
            // what the leaves refer to: live (triangle slot, material) pairs; refit mode keeps the tree while this and the
            // Blend flags stay what the last build saw
            std::vector<std::pair<uint32_t, uint32_t>> leaves;
            for (auto& p : prims_all) if (p.is_alive()) leaves.push_back({p.triangle_id, p.material_id});
            if (bvh_refit_mode && built_with_refit_mode && leaves == built_leaves && blend == built_blend && !bvh.nodes.empty()) {
                bvh.refit(0, prims_all);
                refits += 1;
            } else {
                bvh.current.clear();
                for (auto& p : prims_all) if (p.is_alive()) bvh.current.push_back(p);
                bvh.build();
                built_leaves = leaves; built_blend = blend; built_with_refit_mode = bvh_refit_mode;
                rebuilds += 1;
            }
            bvh_buffer.clear();
            bvh.serialize(0, blend, bvh_buffer);
        }
        world.light_count = next_light_id; world.sun_azimuth = sun_azimuth; world.sun_altitude = sun_altitude;
        if (sun_dirty) {
            sun_dirty = false;
            Vec3 color = transmittance_eval(Atmosphere::view_pos(), world.sun_dir());
            color = color * Atmosphere::EXPOSURE * 5.0f;
            update_light(0, -1, Light::sun(world.sun_pos(), color));
        }
        flush_lights();
        for (auto& kv : cameras) kv.second->frame = Frame{frame};
        frame += 1;
    }

    EngineView view() const {
        EngineView v;
        v.scene.triangles = triangles.data(); v.scene.bvh = bvh_buffer.data(); v.scene.bvh_len = bvh_buffer.size(); v.scene.materials = gpu_materials.data();
        v.scene.atlas = Atlas{atlas.empty() ? nullptr : atlas.data(), atlas_w, atlas_h};
        v.lights = LightsView{gpu_lights.data(), gpu_lights.size()};
        v.world = world;
        v.blue_noise = BlueNoiseTex{blue_noise.data()};
        v.atmosphere.transmittance_lut = LutTex{transmittance_lut.data(), 256, 64};
        v.atmosphere.sky_lut = LutTex{sky_lut.data(), 256, 256};
        return v;
    }

    // -------- cameras (camera_controller.rs:27-201)
    uint64_t create_camera(const ApiCamera& c) {
        auto slot = std::make_unique<CameraSlot>();
        slot->api = c;
        slot->buffers.allocate(c.width, c.height);
        slot->buffers.curr_camera = slot->serialize();
        slot->buffers.prev_camera = slot->buffers.curr_camera;
        uint64_t h = next_camera++;
        cameras[h] = std::move(slot);
        return h;
    }
    bool update_camera(uint64_t h, const ApiCamera& c) {
        auto it = cameras.find(h);
        if (it == cameras.end()) return false;
        CameraSlot& s = *it->second;
        bool invalidated = s.api.mode != c.mode || s.api.denoise != c.denoise || s.api.depth != c.depth || s.api.width != c.width || s.api.height != c.height;
        s.api = c;
        s.buffers.prev_camera = s.buffers.curr_camera;
        s.buffers.curr_camera = s.serialize();
        if (invalidated) { Camera cc = s.buffers.curr_camera, pc = s.buffers.prev_camera; s.buffers = CameraBuffers(); s.buffers.allocate(c.width, c.height); s.buffers.curr_camera = cc; s.buffers.prev_camera = pc; }
        return true;
    }

    void run_atmosphere() {  // passes/atmosphere.rs:78-110
        if (!atmosphere_initialized) {
            generate_transmittance_lut(transmittance_lut);
            generate_scattering_lut(LutTex{transmittance_lut.data(), 256, 64}, scattering_lut);
            atmosphere_initialized = true;
        }
        if (!sky_known || known_sun_altitude != sun_altitude) {
            generate_sky_lut(LutTex{transmittance_lut.data(), 256, 64}, LutTex{scattering_lut.data(), 32, 32}, world.sun_altitude, sky_lut);
            sky_known = true; known_sun_altitude = sun_altitude;
        }
    }
    uint64_t pass_mask = ~0ull;  // which reference passes render_camera executes (parity tests step through a frame one launch at a time)
    bool render_camera(uint64_t h, Vec4* out) {
        auto it = cameras.find(h);
        if (it == cameras.end()) return false;
        if (it->second->api.mode != 5) run_atmosphere();
        CameraSlot& s = *it->second;
        CameraBuffers& b = s.buffers;
        EngineView e = view();
        Frame frame = s.frame;
        bool alt = frame.id % 2 == 1;
        uint32_t mode = s.api.mode;
        bool denoise = s.api.denoise != 0 && mode <= 4;
        auto seed = [&](uint32_t pass) { return pass_seed(base_seed, frame.id, pass); };
        auto on = [&](uint32_t bit) { return (pass_mask >> bit) & 1ull; };  // or_debug_set_pass_mask: bit numbers = include/strolle_hip.h StPassBit
        if (mode == 5) {
            if (on(27)) pass_bvh_heatmap(e, b);
        } else if (mode == 6) {
            for (uint32_t d = 0; d <= s.api.depth; d++) {
                if (on(28)) pass_ref_tracing(e, b, d);
                if (on(29)) pass_ref_shading(e, b, seed(PASS_REF_SHADING + d), d);
            }
            if (on(29)) pass_ref_shading(e, b, seed(PASS_REF_SHADING + 255), 255);
        } else {
            bool needs_di = mode == 0 || mode == 1 || mode == 2;
            bool needs_gi = mode == 0 || mode == 3 || mode == 4;
            const InstanceXforms xf{xf_curr_inv.data(), xf_prev.data(), triangle_slot.data()};
            if (on(0)) pass_prim_visibility(e, b, alt, &xf);
            if (!instances.empty()) {
                if (on(1)) pass_frame_reprojection(b, alt);
                if (needs_di) {
                    if (on(2)) pass_di_sampling(e, b, alt, seed(PASS_DI_SAMPLING), frame);
                    if (on(3)) pass_di_temporal_resampling(e, b, alt, seed(PASS_DI_TEMPORAL));
                    if (on(4)) pass_di_spatial_pick(e, b, alt, seed(PASS_DI_SPATIAL_PICK), frame);
                    if (on(5)) pass_spatial_trace(e, b, b.di_diff_samples, b.di_diff_curr_colors, b.di_diff_stash);
                    if (on(6)) pass_di_spatial_sample(b, seed(PASS_DI_SPATIAL_SAMPLE), frame);
                    if (on(7)) pass_di_resolving(e, b, alt);
                }
                if (needs_gi) {
                    uint32_t source;
                    if (on(8)) pass_gi_reprojection(b, alt);
                    if (frame.is_gi_tracing()) {
                        if (frame.id % 2 == 0) {
                            if (on(9)) pass_gi_sampling_a(e, b, alt, seed(PASS_GI_SAMPLING_A), frame);
                            if (on(10)) pass_gi_sampling_b(e, b, alt, seed(PASS_GI_SAMPLING_B), frame);
                        }
                        if (on(11)) pass_gi_temporal_resampling(b, alt, seed(PASS_GI_TEMPORAL), frame);
                        if (frame.id % 2 == 1) {
                            if (on(12)) pass_gi_spatial_pick(b, alt, seed(PASS_GI_SPATIAL_PICK), frame);
                            if (on(13)) pass_spatial_trace(e, b, b.gi_d0, b.gi_d1, b.gi_d2);
                            if (on(14)) pass_gi_spatial_sample(b, seed(PASS_GI_SPATIAL_SAMPLE), frame);
                            source = 1;
                        } else source = 0;
                    } else {
                        if (on(9)) pass_gi_sampling_a(e, b, alt, seed(PASS_GI_SAMPLING_A), frame);
                        if (on(10)) pass_gi_sampling_b(e, b, alt, seed(PASS_GI_SAMPLING_B), frame);
                        if (on(11)) pass_gi_temporal_resampling(b, alt, seed(PASS_GI_TEMPORAL), frame);
                        source = 0;
                    }
                    // gi_preview_resampling (host :60-74): pass 1 reads [1] or [2] -> [3]; pass 2 reads [3] (source forced to 1) -> [0]
                    uint32_t pseed = seed(PASS_GI_PREVIEW);
                    if (on(15)) pass_gi_preview_resampling(b, alt, pseed, source, 0, b.gi_reservoirs[1], b.gi_reservoirs[2], b.gi_reservoirs[3]);
                    if (on(16)) pass_gi_preview_resampling(b, alt, pseed, 1, 1, b.gi_reservoirs[1], b.gi_reservoirs[3], b.gi_reservoirs[0]);
                    if (on(17)) pass_gi_resolving(b, alt, source);
                }
            }
            if (denoise) {
                if (on(18)) pass_denoise_reproject(b, alt, b.di_diff_prev_colors, b.di_diff_moments[!alt], b.di_diff_samples, b.di_diff_curr_colors, b.di_diff_moments[alt]);
                if (on(19)) pass_denoise_reproject(b, alt, b.gi_diff_prev_colors, b.gi_diff_moments[!alt], b.gi_diff_samples, b.gi_diff_curr_colors, b.gi_diff_moments[alt]);
                if (on(20)) pass_denoise_estimate_variance(b, alt);
                struct WP { Plane *di_in, *di_out, *gi_in, *gi_out; };
                WP wp[5] = {{&b.di_diff_stash, &b.di_diff_prev_colors, &b.gi_diff_stash, &b.gi_diff_prev_colors},
                            {&b.di_diff_prev_colors, &b.di_diff_stash, &b.gi_diff_prev_colors, &b.gi_diff_stash},
                            {&b.di_diff_stash, &b.di_diff_curr_colors, &b.gi_diff_stash, &b.gi_diff_curr_colors},
                            {&b.di_diff_curr_colors, &b.di_diff_stash, &b.gi_diff_curr_colors, &b.gi_diff_stash},
                            {&b.di_diff_stash, &b.di_diff_curr_colors, &b.gi_diff_stash, &b.gi_diff_curr_colors}};
                for (uint32_t nth = 0; nth < 5; nth++)
                    if (on(21 + nth)) pass_denoise_wavelet(e, b, alt, frame, 1u << nth, (float)(1 + nth), *wp[nth].di_in, *wp[nth].di_out, *wp[nth].gi_in, *wp[nth].gi_out);
            }
        }
        if (out) {
            bool dn = s.api.denoise != 0;
            if (on(26)) pass_frame_composition(b, alt, mode, dn && (mode == 0 || mode == 1), dn && (mode == 0 || mode == 3), out);
        }
        return true;
    }
};

}  // namespace orc
