// ORACLE — TEST INFRASTRUCTURE ONLY (see or_math.h header).
//
// CPU restatement of the reference's `strolle-gpu` crate: the device-side
// library every shader entry point calls. Each block cites the reference
// file:line it follows (paths relative to /root/reference).
//
// Parity status: the reference's own tests pin only Camera::contain, the
// G-buffer / DiReservoir / Reprojection round trips and the u32 byte packing
// (SURVEY.md §4); those are reproduced in tests/test_oracle_layouts.py.
// Traversal, shading, resampling and denoising have no golden vectors in the
// reference and the reference cannot be built here => PARITY UNPINNED for them.
#pragma once
#include <vector>

#include "or_math.h"

namespace orc {

// ------------------------------------------------ utils/u32_ext.rs:10-29
static inline uint32_t u32_from_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    return a | (b << 8) | (c << 16) | (d << 24);
}
static inline void u32_to_bytes(uint32_t v, uint32_t out[4]) {
    out[0] = v & 0xff; v >>= 8; out[1] = v & 0xff; v >>= 8; out[2] = v & 0xff; v >>= 8; out[3] = v & 0xff;
}

// ------------------------------------------------ utils/vec3_ext.rs:31-62
static inline Vec3 reflect(Vec3 self, Vec3 other) { return self - 2.0f * dot(other, self) * other; }
static inline float luma(Vec3 c) { return dot(c, Vec3(0.2126f, 0.7152f, 0.0722f)); }
static inline float perc_luma(Vec3 c) { return sqrtf(luma(c)); }

// utils.rs:21-43
static inline float lerpf(float a, float b, float t) { return a + (b - a) * clampf(t, 0.0f, 1.0f); }
static inline Vec3 lerp3(Vec3 a, Vec3 b, float t) { return a + (b - a) * clampf(t, 0.0f, 1.0f); }
static inline UVec2 resolve_checkerboard(UVec2 gid, uint32_t frame) {
    return UVec2(gid.x * 2 + ((frame + gid.y) % 2), gid.y);
}
static inline UVec2 resolve_checkerboard_alt(UVec2 gid, uint32_t frame) { return resolve_checkerboard(gid, frame + 1); }
static inline bool got_checkerboard_at(UVec2 p, uint32_t frame) {
    return p == resolve_checkerboard(UVec2(p.x / 2, p.y), frame);
}

// ------------------------------------------------ normal.rs:9-34
static inline Vec2 normal_encode(Vec3 n) {
    n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
    Vec2 r;
    if (n.z >= 0.0f) {
        r = Vec2(n.x, n.y);
    } else {
        Vec2 t = 1.0f - Vec2(fabsf(n.y), fabsf(n.x));
        t.x = copysignf(t.x, n.x);
        t.y = copysignf(t.y, n.y);
        r = t;
    }
    return r * 0.5f + 0.5f;
}
static inline Vec3 normal_decode(Vec2 e) {
    e = e * 2.0f - 1.0f;
    Vec3 n(e.x, e.y, 1.0f - fabsf(e.x) - fabsf(e.y));
    float t = fmax_(-n.z, 0.0f);
    n.x -= copysignf(t, n.x);
    n.y -= copysignf(t, n.y);
    return normalize(n);
}

// ------------------------------------------------ frame.rs:10-26
struct Frame {
    uint32_t id;
    bool is_gi_tracing() const { return id % 6 < 4; }
    bool is_gi_validation() const { return !is_gi_tracing(); }
};

// ------------------------------------------------ world.rs:7-30
struct World {
    uint32_t light_count;
    float sun_azimuth, sun_altitude;
    Vec3 sun_dir() const {
        return Vec3(stm_cos(sun_altitude) * stm_sin(sun_azimuth), stm_sin(sun_altitude),
                    -stm_cos(sun_altitude) * stm_cos(sun_azimuth));
    }
    Vec3 sun_pos() const { return sun_dir() * 1000.0f; }
};
static const float SUN_DISTANCE = 1000.0f;

// ------------------------------------------------ noise/white.rs:10-83
struct WhiteNoise {
    uint32_t state;
    static WhiteNoise make(uint32_t seed, UVec2 id) { return WhiteNoise{seed ^ (48619u * id.x) ^ (95461u * id.y)}; }
    uint32_t sample_int() {
        state = state * 747796405u + 2891336453u;
        uint32_t word = ((state >> ((state >> 28) + 4)) ^ state) * 277803737u;
        return (word >> 22) ^ word;
    }
    float sample() { return (float)sample_int() / 4294967296.0f; }  // u32::MAX as f32 == 2^32
    Vec2 sample_circle() {
        float angle = sample() * PI * 2.0f;
        return Vec2(stm_cos(angle), stm_sin(angle));
    }
    Vec2 sample_disk() {
        float radius = sqrtf(sample());
        return sample_circle() * radius;
    }
    Vec3 sample_sphere() {
        float phi = sample() * 2.0f * PI;
        float cos_theta = sample() * 2.0f - 1.0f;
        float u = sample();
        float theta = stm_acos(cos_theta);
        float r = sqrtf(u);
        return Vec3(r * stm_sin(theta) * stm_cos(phi), r * stm_sin(theta) * stm_sin(phi), r * stm_cos(theta));
    }
    Vec3 sample_hemisphere(Vec3 normal) {
        float cos_theta = sample();
        float sin_theta = sqrtf(1.0f - sqr(cos_theta));
        float phi = 2.0f * PI * sample();
        Vec3 t, b;
        any_orthonormal_pair(normal, &t, &b);
        return (t * stm_cos(phi) + b * stm_sin(phi)) * sin_theta + normal * cos_theta;
    }
};

// ------------------------------------------------ noise/blue.rs:10-27
// 256x256 RGBA8 (unorm) texture; `read` returns channel/255.
struct BlueNoiseTex { const uint8_t* rgba; };
struct BlueNoise {
    BlueNoiseTex tex; UVec2 uv;
    static BlueNoise make(BlueNoiseTex tex, UVec2 id, Frame frame) {
        return BlueNoise{tex, UVec2((id.x + 71u * frame.id) % 256u, (id.y + 11u * frame.id) % 256u)};
    }
    Vec4 read() const {
        const uint8_t* p = tex.rgba + 4 * (uv.y * 256u + uv.x);
        return Vec4((float)p[0] / 255.0f, (float)p[1] / 255.0f, (float)p[2] / 255.0f, (float)p[3] / 255.0f);
    }
    Vec2 first_sample() const { return read().xy(); }
    Vec2 second_sample() const { return read().zw(); }
};

// ------------------------------------------------ camera.rs:9-150
struct Ray;
struct Camera {
    Mat4 projection_view, ndc_to_world;
    Vec4 origin, screen;

    Vec4 world_to_clip(Vec3 pos) const { return mul(projection_view, Vec4(pos, 1.0f)); }
    Vec2 clip_to_screen(Vec4 pos) const {
        Vec2 ndc = pos.xy() / pos.w;
        ndc = Vec2(ndc.x, -ndc.y);
        return (0.5f * ndc + 0.5f) * screen.xy();
    }
    Vec2 world_to_screen(Vec3 pos) const { return clip_to_screen(world_to_clip(pos)); }
    size_t screen_to_idx(UVec2 pos) const { return (size_t)(pos.y * f2u_sat(screen.x) + pos.x); }
    UVec2 screen_size() const { return as_uvec2(screen.xy()); }
    UVec2 contain(IVec2 pos) const {
        IVec2 ss = as_ivec2(screen.xy());
        if (pos.x < 0) pos.x = -pos.x;
        if (pos.y < 0) pos.y = -pos.y;
        if (pos.x >= ss.x) pos.x = ss.x - pos.x + ss.x - 1;
        if (pos.y >= ss.y) pos.y = ss.y - pos.y + ss.y - 1;
        return as_uvec2(pos);
    }
    bool contains(UVec2 p) const { UVec2 ss = as_uvec2(screen.xy()); return p.x < ss.x && p.y < ss.y; }
    bool contains(IVec2 p) const { IVec2 ss = as_ivec2(screen.xy()); return p.x >= 0 && p.y >= 0 && p.x < ss.x && p.y < ss.y; }
    bool contains(Vec2 p) const { Vec2 ss = screen.xy(); return p.x >= 0.0f && p.y >= 0.0f && p.x < ss.x && p.y < ss.y; }
    bool is_eq(const Camera& rhs) const {  // Mat4::abs_diff_eq(.., 0.0025)
        for (int i = 0; i < 4; i++) {
            const Vec4 &a = projection_view.c[i], &b = rhs.projection_view.c[i];
            if (!(fabsf(a.x - b.x) <= 0.0025f && fabsf(a.y - b.y) <= 0.0025f && fabsf(a.z - b.z) <= 0.0025f && fabsf(a.w - b.w) <= 0.0025f)) return false;
        }
        return true;
    }
    inline Ray ray(UVec2 screen_pos) const;
};

// ------------------------------------------------ material.rs:8-104
struct Material {
    Vec4 base_color, base_color_texture, emissive, emissive_texture;
    float roughness, metallic, reflectance, ior;
    Vec4 metallic_roughness_texture, normal_map_texture;
    void regularize() { roughness = fmax_(roughness, 0.75f * 0.75f); }
};
static_assert(sizeof(Material) == 112, "Material POD is 112 B (material.rs:8-23)");

// Atlas: RGBA8-sRGB texels in a linear buffer; bilinear, clamp-to-edge, lod 0
// (manual restatement of `sample_by_lod`; gfx950 has no texture unit).
struct Atlas {
    const uint8_t* rgba; uint32_t width, height;
    static float srgb_to_linear(uint8_t v) {
        float c = (float)v / 255.0f;
        if (c <= 0.04045f) return c / 12.92f;
        return stm_pow((c + 0.055f) / 1.055f, 2.4f);
    }
    Vec4 texel(int32_t x, int32_t y) const {
        if (x < 0) x = 0; if (y < 0) y = 0;
        if (x >= (int32_t)width) x = (int32_t)width - 1;
        if (y >= (int32_t)height) y = (int32_t)height - 1;
        const uint8_t* p = rgba + 4 * ((size_t)y * width + (size_t)x);
        return Vec4(srgb_to_linear(p[0]), srgb_to_linear(p[1]), srgb_to_linear(p[2]), (float)p[3] / 255.0f);
    }
    Vec4 sample(Vec2 uv) const {
        if (!rgba || width == 0) return Vec4(0, 0, 0, 0);
        if (uv.x != uv.x) uv.x = 0.0f;  // texture units sanitise NaN coordinates; so does this sampler
        if (uv.y != uv.y) uv.y = 0.0f;
        float fx = uv.x * (float)width - 0.5f, fy = uv.y * (float)height - 0.5f;
        float x0 = floorf(fx), y0 = floorf(fy);
        float tx = fx - x0, ty = fy - y0;
        int32_t ix = f2i_sat(x0), iy = f2i_sat(y0);
        Vec4 a = texel(ix, iy), b = texel(ix + 1, iy), c = texel(ix, iy + 1), d = texel(ix + 1, iy + 1);
        Vec4 top = a + (b - a) * tx;
        Vec4 bot = c + (d - c) * tx;
        return top + (bot - top) * ty;
    }
};
static inline float mat_wrap(float t) { return t > 0.0f ? fmodf(t, 1.0f) : 1.0f - fmodf(-t, 1.0f); }
static inline Vec4 sample_atlas(const Atlas& atlas, Vec2 hit_uv, Vec4 multiplier, Vec4 texture) {
    if (texture == Vec4(0, 0, 0, 0)) return multiplier;
    hit_uv.x = mat_wrap(hit_uv.x);
    hit_uv.y = mat_wrap(hit_uv.y);
    Vec2 uv = texture.xy() + hit_uv * texture.zw();
    return multiplier * atlas.sample(uv);
}
static inline Vec4 mat_base_color(const Material& m, const Atlas& a, Vec2 uv) { return sample_atlas(a, uv, m.base_color, m.base_color_texture); }
static inline Vec2 mat_metallic_roughness(const Material& m, const Atlas& a, Vec2 uv) {
    Vec4 s = sample_atlas(a, uv, Vec4(1.0f, m.roughness, m.metallic, 1.0f), m.metallic_roughness_texture);
    return Vec2(s.z, s.y);
}
static inline Vec3 mat_emissive(const Material& m, const Atlas& a, Vec2 uv) { return sample_atlas(a, uv, m.emissive, m.emissive_texture).xyz(); }

// ------------------------------------------------ triangle.rs:9-113, hit.rs:83-129
struct Triangle { Vec4 d0, d1, d2, d3, d4, d5, d6, d7, d8; };
static_assert(sizeof(Triangle) == 144, "Triangle POD is 144 B");

struct TriangleHit {
    float distance; Vec3 point, normal; Vec2 uv; uint32_t material_id;
    static TriangleHit none() { TriangleHit h; h.distance = F32_MAX; h.material_id = 0; return h; }
    bool is_some() const { return distance < F32_MAX; }
    bool is_none() const { return !is_some(); }
    void pack(Vec4 out[2]) const {
        out[0] = Vec4(point, b2f(material_id));
        Vec2 n = normal_encode(normal);
        out[1] = Vec4(n.x, n.y, uv.x, uv.y);
    }
    static TriangleHit unpack(Vec4 d0, Vec4 d1) {
        if (d0.xyz() == Vec3()) return none();
        TriangleHit h;
        h.distance = 0.0f; h.point = d0.xyz(); h.normal = normal_decode(d1.xy()); h.uv = d1.zw(); h.material_id = f2b(d0.w);
        return h;
    }
};

// ------------------------------------------------ ray.rs:14-328
struct SceneView {  // what the shaders bind: triangles, bvh, materials, atlas
    const Triangle* triangles; const Vec4* bvh; const Material* materials; Atlas atlas; size_t bvh_len;
};
enum Tracing { ReturnClosest, ReturnFirst };
static const int BVH_STACK_SIZE = 24;  // lib.rs:76
// Test-only switches of the checker (or_debug_set_stack_limit / or_debug_stack_stats): the number of pending entries a ray may keep (24 as the
// reference's lib.rs:76 by default; up to BVH_STACK_MAX stands for "unbounded": the deepest chain of any scene here is 26) and what the rays
// actually needed — pushes that were dropped at the limit and the deepest stack any ray reached. Process-wide, relaxed atomics.
static const int BVH_STACK_MAX = 64;
static int g_or_stack_limit = BVH_STACK_SIZE;
static unsigned long long g_or_dropped_pushes = 0;
static int g_or_deepest_stack = 0;

struct Ray {
    Vec3 origin, dir, inv_dir; float len;
    static Ray make(Vec3 o, Vec3 d) { Ray r; r.origin = o; r.dir = d; r.inv_dir = 1.0f / d; r.len = F32_MAX; return r; }
    Ray with_len(float l) const { Ray r = *this; r.len = l; return r; }
    Vec3 at(float t) const { return origin + dir * t; }

    float intersect_box(Vec3 bmin, Vec3 bmax) const {  // ray.rs:273-302
        float tmin = 0.0f, tmax = F32_MAX;
        Vec3 t1 = (bmin - origin) * inv_dir;
        Vec3 t2 = (bmax - origin) * inv_dir;
        tmin = fmax_(tmin, fmin_(t1.x, t2.x)); tmax = fmin_(tmax, fmax_(t1.x, t2.x));
        tmin = fmax_(tmin, fmin_(t1.y, t2.y)); tmax = fmin_(tmax, fmax_(t1.y, t2.y));
        tmin = fmax_(tmin, fmin_(t1.z, t2.z)); tmax = fmin_(tmax, fmax_(t1.z, t2.z));
        return tmin <= tmax ? tmin : F32_MAX;
    }
    float intersect_sphere(float radius) const {  // ray.rs:304-321
        float b = dot(origin, dir);
        float c = dot(origin, origin) - radius * radius;
        if (c > 0.0f && b > 0.0f) return -1.0f;
        float discr = b * b - c;
        if (discr < 0.0f) return -1.0f;
        if (discr > b * b) return -b + sqrtf(discr);
        return -b - sqrtf(discr);
    }
    // triangle.rs:64-113
    bool hit_triangle(const Triangle& t, TriangleHit* hit) const {
        Vec3 p0 = t.d0.xyz(), p1 = t.d3.xyz(), p2 = t.d6.xyz();
        Vec3 v0v1 = p1 - p0, v0v2 = p2 - p0;
        Vec3 pvec = cross(dir, v0v2);
        float det = dot(v0v1, pvec);
        if (fabsf(det) < F32_EPSILON) return false;
        float inv_det = 1.0f / det;
        Vec3 tvec = origin - p0;
        float u = dot(tvec, pvec) * inv_det;
        Vec3 qvec = cross(tvec, v0v1);
        float v = dot(dir, qvec) * inv_det;
        float distance = dot(v0v2, qvec) * inv_det;
        if ((u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (distance <= 0.0f) | (distance >= hit->distance)) return false;
        Vec3 n = u * t.d4.xyz() + v * t.d7.xyz() + (1.0f - u - v) * t.d1.xyz();
        n = normalize(n) * copysignf(1.0f, inv_det);
        Vec2 uv0(t.d0.w, t.d1.w), uv1(t.d3.w, t.d4.w), uv2(t.d6.w, t.d7.w);
        Vec2 uv = uv0 + (uv1 - uv0) * u + (uv2 - uv0) * v;
        hit->uv = uv; hit->normal = n; hit->distance = distance;
        return true;
    }
    // ray.rs:114-266. The per-thread slice of the workgroup stack is a local array here.
    // Deviation (documented): pushes beyond BVH_STACK_SIZE are dropped instead of
    // corrupting the neighbouring lane's slice (UB in the reference).
    size_t traverse(const SceneView& s, Tracing tracing, TriangleHit* hit, uint32_t* hit_triangle_id = nullptr) const {
        size_t used_memory = 0;
        if (s.bvh_len == 0) return 0;  // deviation: empty world == miss (the reference would read an empty buffer)
        uint32_t bvh_ptr = 0;
        uint32_t stack[BVH_STACK_MAX];
        int stack_ptr = 0, deepest = 0;
        const int limit = g_or_stack_limit;
        for (;;) {
            used_memory += 16;
            Vec4 d0 = s.bvh[bvh_ptr];
            bool is_internal = f2b(d0.w) == 0;
            if (is_internal) {
                used_memory += 3 * 16;
                Vec4 d1 = s.bvh[bvh_ptr + 1], d2 = s.bvh[bvh_ptr + 2], d3 = s.bvh[bvh_ptr + 3];
                uint32_t near_ptr = bvh_ptr + 4, far_ptr = f2b(d1.w);
                float near_d = intersect_box(d0.xyz(), d1.xyz());
                float far_d = intersect_box(d2.xyz(), d3.xyz());
                if (far_d < near_d) { uint32_t tp = near_ptr; near_ptr = far_ptr; far_ptr = tp; float td = near_d; near_d = far_d; far_d = td; }
                if (far_d < hit->distance) {
                    if (stack_ptr < limit) { stack[stack_ptr++] = far_ptr; if (stack_ptr > deepest) deepest = stack_ptr; }
                    else __atomic_fetch_add(&g_or_dropped_pushes, 1ull, __ATOMIC_RELAXED);
                }
                if (near_d < hit->distance) { bvh_ptr = near_ptr; continue; }
            } else {
                used_memory += sizeof(Triangle);
                uint32_t flags = f2b(d0.x);
                bool got_more = (flags & 1) == 1;
                bool has_alpha = (flags & 2) == 2;
                uint32_t triangle_id = f2b(d0.y), material_id = f2b(d0.z);
                Vec2 prev_uv = hit->uv; Vec3 prev_normal = hit->normal; float prev_distance = hit->distance;
                bool found = hit_triangle(s.triangles[triangle_id], hit);
                if (found && has_alpha) {
                    used_memory += sizeof(Material);
                    used_memory += 16;
                    Vec4 bc = mat_base_color(s.materials[material_id], s.atlas, hit->uv);
                    if (bc.w < 1.0f) { found = false; hit->uv = prev_uv; hit->normal = prev_normal; hit->distance = prev_distance; }
                }
                if (found) {
                    hit->material_id = material_id;
                    if (hit_triangle_id) *hit_triangle_id = triangle_id;
                    if (tracing == ReturnFirst) break;
                }
                if (got_more) { bvh_ptr += 1; continue; }
            }
            if (stack_ptr > 0) { stack_ptr -= 1; bvh_ptr = stack[stack_ptr]; } else break;
        }
        if (hit->is_some()) hit->point = at(hit->distance);
        if (deepest > __atomic_load_n(&g_or_deepest_stack, __ATOMIC_RELAXED)) {   // (rare after the first rays: a max, kept with a CAS loop)
            int seen = __atomic_load_n(&g_or_deepest_stack, __ATOMIC_RELAXED);
            while (deepest > seen && !__atomic_compare_exchange_n(&g_or_deepest_stack, &seen, deepest, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
        }
        return used_memory;
    }
    TriangleHit trace(const SceneView& s, size_t* used_memory = nullptr) const {  // ray.rs:55-80
        TriangleHit hit = TriangleHit::none();
        size_t um = traverse(s, ReturnClosest, &hit);
        if (used_memory) *used_memory = um;
        return hit;
    }
    bool intersect(const SceneView& s) const {  // ray.rs:84-112
        TriangleHit hit = TriangleHit::none();
        hit.distance = len;
        traverse(s, ReturnFirst, &hit);
        return hit.distance < len;
    }
};

inline Ray Camera::ray(UVec2 screen_pos) const {  // camera.rs:80-93
    Vec2 screen_size = screen.xy();
    Vec2 sp = as_vec2(screen_pos) + Vec2(0.5f, 0.5f);
    Vec2 ndc = sp * 2.0f / screen_size - Vec2(1.0f, 1.0f);
    ndc = Vec2(ndc.x, -ndc.y);
    Vec3 far_plane = project_point3(ndc_to_world, Vec3(ndc.x, ndc.y, F32_EPSILON));
    Vec3 near_plane = project_point3(ndc_to_world, Vec3(ndc.x, ndc.y, 1.0f));
    return Ray::make(near_plane, normalize(far_plane - near_plane));
}

// ------------------------------------------------ gbuffer.rs:8-124
struct GBufferEntry {
    Vec4 base_color; Vec3 normal; float metallic; Vec3 emissive; float roughness, reflectance, depth;
    GBufferEntry() : metallic(0), roughness(0), reflectance(0), depth(0) {}
    static GBufferEntry unpack(Vec4 d0, Vec4 d1) {
        GBufferEntry g;
        g.depth = d0.x;
        g.normal = normal_decode(d0.yz());
        uint32_t by[4]; u32_to_bytes(f2b(d0.w), by);
        g.metallic = (float)by[0] / 255.0f;
        g.roughness = sqr((float)by[1] / 255.0f);
        g.reflectance = (float)by[2] / 255.0f;
        g.emissive = d1.xyz();
        u32_to_bytes(f2b(d1.w), by);
        g.base_color = Vec4(stm_pow((float)by[0] / 255.0f, 2.2f), stm_pow((float)by[1] / 255.0f, 2.2f),
                            stm_pow((float)by[2] / 255.0f, 2.2f), stm_pow((float)by[3] / 63.0f, 2.2f));
        return g;
    }
    void pack(Vec4 out[2]) const {
        Vec2 n = normal_encode(normal);
        float m = clampf(metallic, 0.0f, 1.0f) * 255.0f;
        float r = clampf(sqrtf(roughness), 0.0f, 1.0f) * 255.0f;
        float rf = clampf(reflectance, 0.0f, 1.0f) * 255.0f;
        out[0] = Vec4(depth, n.x, n.y, b2f(u32_from_bytes(f2u_sat(m), f2u_sat(r), f2u_sat(rf), 1)));
        const float ig = 1.0f / 2.2f;
        float bx = clampf(stm_pow(base_color.x, ig), 0.0f, 1.0f), by = clampf(stm_pow(base_color.y, ig), 0.0f, 1.0f);
        float bz = clampf(stm_pow(base_color.z, ig), 0.0f, 1.0f), bw = clampf(stm_pow(base_color.w, ig), 0.0f, 1.0f);
        // Vec4::clamp = max(min).min(max): NaN -> min side; stm_pow never returns NaN for x >= 0
        out[1] = Vec4(emissive.x, emissive.y, emissive.z,
                      b2f(u32_from_bytes(f2u_sat(bx * 255.0f), f2u_sat(by * 255.0f), f2u_sat(bz * 255.0f), f2u_sat(bw * 63.0f))));
    }
    bool is_some() const { return depth != 0.0f; }
    float clamped_roughness() const { return clampf(roughness, 0.089f * 0.089f, 1.0f); }
};

// ------------------------------------------------ surface.rs:8-69
struct Surface {
    Vec3 normal; float depth, roughness;
    bool is_sky() const { return depth == 0.0f; }
    float evaluate_similarity_to(const Surface& other) const {
        if (is_sky() || other.is_sky()) return 0.0f;
        float d = fmax_(dot(normal, other.normal), 0.0f);
        float normal_score = d <= 0.5f ? 0.0f : 2.0f * d;
        float t = fabsf(depth - other.depth);
        float depth_score = t >= 0.1f * other.depth ? 0.0f : 1.0f;
        return normal_score * depth_score;
    }
    static Surface from_texel(Vec4 d0) { Surface s; s.normal = normal_decode(d0.xy()); s.depth = d0.z; s.roughness = d0.w; return s; }
};

// ------------------------------------------------ hit.rs:8-81
struct Hit {
    Vec3 origin, dir, point; GBufferEntry gbuffer;
    static constexpr float NUDGE_OFFSET = 0.01f;
    static Hit make(const Ray& ray, const GBufferEntry& g) {
        Hit h; h.origin = ray.origin; h.dir = ray.dir; h.point = ray.at(g.depth - NUDGE_OFFSET); h.gbuffer = g; return h;
    }
    bool is_some() const { return gbuffer.is_some(); }
    bool is_none() const { return !is_some(); }
};

// ------------------------------------------------ brdf.rs:9-186
static inline float ggx_distribution(float n_dot_h, float roughness) {
    float a2 = roughness * roughness;
    float d = (n_dot_h * a2 - n_dot_h) * n_dot_h + 1.0f;
    return a2 / (PI * d * d);
}
static inline float ggx_schlick_masking_term(float n_dot_l, float n_dot_v, float roughness) {
    float k = roughness * roughness / 2.0f;
    float g_v = n_dot_v / (n_dot_v * (1.0f - k) + k);
    float g_l = n_dot_l / (n_dot_l * (1.0f - k) + k);
    return g_v * g_l;
}
static inline Vec3 f_schlick_vec(Vec3 f0, float f90, float v_dot_h) {
    return f0 + (Vec3::splat(f90) - f0) * stm_pow5(fmax_(1.0f - v_dot_h, 0.001f));
}
static inline Vec3 ggx_schlick_fresnel(Vec3 f0, float l_dot_h) {
    float f90 = saturate(dot(f0, Vec3::splat(50.0f * 0.33f)));
    return f_schlick_vec(f0, f90, l_dot_h);
}
struct BrdfSample { Vec3 dir; float pdf; Vec3 radiance; bool is_invalid() const { return pdf == 0.0f; } };

static inline Vec3 diffuse_brdf_eval(const GBufferEntry& g) { return g.base_color.xyz() * (1.0f - g.metallic) / PI; }
static inline Vec3 specular_brdf_eval(const GBufferEntry& g, Vec3 l, Vec3 v) {
    if (g.metallic <= 0.0f) return Vec3();
    float a = g.clamped_roughness();
    Vec3 n = g.normal;
    Vec3 h = normalize(l + v);
    float n_dot_l = saturate(dot(n, l)), n_dot_h = saturate(dot(n, h)), l_dot_h = saturate(dot(l, h)), n_dot_v = saturate(dot(n, v));
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return Vec3();
    float d = ggx_distribution(n_dot_h, a);
    float gg = ggx_schlick_masking_term(n_dot_l, n_dot_v, a);
    Vec3 f0 = Vec3::splat(0.16f * g.reflectance * g.reflectance * (1.0f - g.metallic)) + g.base_color.xyz() * g.metallic;
    Vec3 f = ggx_schlick_fresnel(f0, l_dot_h);
    return d * gg * f / (4.0f * n_dot_l * n_dot_v);
}
static inline BrdfSample diffuse_brdf_sample(const GBufferEntry& g, WhiteNoise& wn) {
    BrdfSample s; s.dir = wn.sample_hemisphere(g.normal); s.pdf = 1.0f / PI; s.radiance = diffuse_brdf_eval(g); return s;
}
static inline BrdfSample specular_brdf_sample(const GBufferEntry& g, WhiteNoise& wn, Vec3 v) {
    float r0 = wn.sample(), r1 = wn.sample();
    float a = g.clamped_roughness();
    Vec3 n = g.normal;
    float a2 = sqr(a);
    Vec3 b, t;
    any_orthonormal_pair(n, &b, &t);  // brdf.rs:91: `let (b, t) = n.any_orthonormal_pair()`
    float cos_theta = sqrtf(fmax_(0.0f, (1.0f - r0) / ((a2 - 1.0f) * r0 + 1.0f)));
    float sin_theta = sqrtf(fmax_(0.0f, 1.0f - cos_theta * cos_theta));
    float phi = r1 * PI * 2.0f;
    Vec3 h = t * (sin_theta * stm_cos(phi)) + b * (sin_theta * stm_sin(phi)) + n * cos_theta;
    float n_dot_h = saturate(dot(n, h)), h_dot_v = saturate(dot(h, v));
    BrdfSample s;
    s.dir = normalize(2.0f * h_dot_v * h - v);
    s.pdf = ggx_distribution(n_dot_h, a) * n_dot_h / (4.0f * h_dot_v);
    s.radiance = specular_brdf_eval(g, s.dir, v);
    return s;
}
static inline BrdfSample layered_brdf_sample(const GBufferEntry& g, WhiteNoise& wn, Vec3 l) {
    BrdfSample s;
    if (wn.sample() < g.metallic) { s = specular_brdf_sample(g, wn, l); s.pdf /= g.metallic; }
    else { s = diffuse_brdf_sample(g, wn); s.pdf /= 1.0f - g.metallic; }
    return s;
}

// ------------------------------------------------ light.rs:12-285
struct LightRadiance {
    Vec3 radiance, diff_brdf, spec_brdf;
    Vec3 sum() const { return radiance * (diff_brdf + spec_brdf); }
};
struct Light {
    Vec4 d0, d1, d2, d3, prev_d0, prev_d1, prev_d2;
    static const uint32_t TYPE_NONE = 0, TYPE_POINT = 1, TYPE_SPOT = 2;
    static Light sun(Vec3 position, Vec3 color) {
        Light l{}; l.d0 = Vec4(position, 25.0f); l.d1 = Vec4(color, INFINITY); l.d2 = Vec4(b2f(TYPE_POINT), 0, 0, 0); return l;
    }
    Vec3 center() const { return d0.xyz(); }
    float radius() const { return d0.w; }
    Vec3 color() const { return d1.xyz(); }
    float range() const { return d1.w; }
    bool contains(Vec3 p) const { return distance(center(), p) <= radius(); }
    uint32_t ty() const { return f2b(d2.x); }
    bool is_none() const { return ty() == TYPE_NONE; }
    bool is_point() const { return ty() == TYPE_POINT; }
    Vec3 spot_dir() const { return normal_decode(d2.yz()); }
    float spot_angle() const { return d2.w; }
    bool is_slot_remapped() const { return f2b(d3.x) > 0 && f2b(d3.x) != 0xcafebabeu; }
    uint32_t slot_remapped_to() const { return f2b(d3.x) - 1; }
    bool is_slot_killed() const { return f2b(d3.x) == 0xcafebabeu; }
    void rollback() { d0 = prev_d0; d1 = prev_d1; d2 = prev_d2; }

    LightRadiance radiance(const Hit& hit) const {  // light.rs:143-207
        Vec3 l = center() - hit.point;
        float f_angle;
        if (is_point()) f_angle = 1.0f;
        else {
            float angle = angle_between(spot_dir(), hit.point - center());
            f_angle = saturate(1.0f - stm_pow3(angle / spot_angle()));
        }
        float f_dist;
        if (range() == INFINITY) f_dist = 1.0f;
        else {
            float l2 = length_squared(l);
            float inv_r2 = 1.0f / sqr(range());
            float factor = l2 * inv_r2;
            float smooth_factor = saturate(1.0f - factor * factor);
            float attenuation = smooth_factor * smooth_factor;
            f_dist = attenuation / fmax_(l2, 0.0001f);
        }
        float f_cosine = saturate(dot(hit.gbuffer.normal, normalize(l)));
        LightRadiance out;
        out.diff_brdf = diffuse_brdf_eval(hit.gbuffer);
        {
            Vec3 v = -hit.dir;
            Vec3 n = hit.gbuffer.normal;
            Vec3 r = reflect(-v, n);
            Vec3 center_to_ray = dot(l, r) * r - l;
            float t = radius() * inverse_sqrt(dot(center_to_ray, center_to_ray));
            Vec3 closest_point = l + center_to_ray * saturate(t);
            float l_spec_length_inverse = inverse_sqrt(dot(closest_point, closest_point));
            float tt = hit.gbuffer.clamped_roughness() + radius() * 0.5f * l_spec_length_inverse;
            float i_roughness = hit.gbuffer.clamped_roughness() / saturate(tt);
            float intensity = sqr(i_roughness);
            Vec3 ls = closest_point * l_spec_length_inverse;
            out.spec_brdf = intensity * specular_brdf_eval(hit.gbuffer, ls, v);
        }
        out.radiance = color() * f_angle * f_dist * f_cosine;
        return out;
    }
    Ray ray_wnoise(WhiteNoise& wn, Vec3 hit_point) const {  // light.rs:209-215
        Vec3 light_pos = center() + radius() * wn.sample_sphere();
        Vec3 light_to_hit = hit_point - light_pos;
        return Ray::make(light_pos, normalize(light_to_hit)).with_len(length(light_to_hit));
    }
    Ray ray_bnoise(Vec2 sample, Vec3 hit_point) const {  // light.rs:217-239
        Vec3 to_light = center() - hit_point;
        Vec3 light_dir = normalize(to_light);
        float light_distance = length(to_light);
        float light_radius = radius() / light_distance;
        Vec3 lt, lb;
        any_orthonormal_pair(light_dir, &lt, &lb);
        float angle = 2.0f * PI * sample.x;
        float rad = sqrtf(sample.y);
        Vec2 disk_point = Vec2(stm_sin(angle), stm_cos(angle)) * rad * light_radius;
        Vec3 ray_dir = light_dir + disk_point.x * lt + disk_point.y * lb;
        ray_dir = normalize(ray_dir);
        return Ray::make(hit_point + ray_dir * light_distance, -ray_dir).with_len(light_distance);
    }
};
static_assert(sizeof(Light) == 112, "Light POD is 112 B");
static const uint32_t LIGHT_ID_SKY = 0xffffffffu;

struct LightsView {
    const Light* items; size_t count;
    // Deviation (documented): out-of-range ids yield an all-zero (TYPE_NONE)
    // light; the reference indexes unchecked.
    Light get(uint32_t id) const { if (id < count) return items[id]; Light l{}; return l; }
    Light get_prev(uint32_t id) const { Light l = get(id); l.rollback(); return l; }
};

// ------------------------------------------------ atmosphere.rs:86-205
struct LutTex {  // RGBA f32 texels, bilinear + clamp-to-edge (manual `sample_by_lod`)
    const Vec4* texels; uint32_t width, height;
    Vec4 texel(int32_t x, int32_t y) const {
        if (x < 0) x = 0; if (y < 0) y = 0;
        if (x >= (int32_t)width) x = (int32_t)width - 1;
        if (y >= (int32_t)height) y = (int32_t)height - 1;
        return texels[(size_t)y * width + (size_t)x];
    }
    Vec4 sample(Vec2 uv) const {
        if (uv.x != uv.x) uv.x = 0.0f;  // texture units sanitise NaN coordinates; so does this sampler
        if (uv.y != uv.y) uv.y = 0.0f;
        float fx = uv.x * (float)width - 0.5f, fy = uv.y * (float)height - 0.5f;
        float x0 = floorf(fx), y0 = floorf(fy);
        float tx = fx - x0, ty = fy - y0;
        int32_t ix = f2i_sat(x0), iy = f2i_sat(y0);
        Vec4 a = texel(ix, iy), b = texel(ix + 1, iy), c = texel(ix, iy + 1), d = texel(ix + 1, iy + 1);
        Vec4 top = a + (b - a) * tx;
        Vec4 bot = c + (d - c) * tx;
        return top + (bot - top) * ty;
    }
};
struct Atmosphere {
    LutTex transmittance_lut, sky_lut;
    static constexpr float GROUND_RADIUS_MM = 6.360f, ATMOSPHERE_RADIUS_MM = 6.460f, EXPOSURE = 20.0f;
    static Vec3 view_pos() { return Vec3(0.0f, GROUND_RADIUS_MM + 0.0002f, 0.0f); }

    Vec3 sample_sky_lut(Vec3 ray_dir, Vec3 sun_dir) const {
        float height = length(view_pos());
        Vec3 up = view_pos() / height;
        float th = sqr(height) - sqr(GROUND_RADIUS_MM);
        th = sqrtf(th) / height;
        float horizon = stm_acos(clampf(th, -1.0f, 1.0f));
        float altitude = horizon - stm_acos(dot(ray_dir, up));
        float azimuth;
        if (fabsf(altitude) > (0.5f * PI - 0.0001f)) azimuth = 0.0f;
        else {
            Vec3 right = cross(sun_dir, up);
            Vec3 forward = cross(up, right);
            Vec3 projected_dir = normalize(ray_dir - up * dot(ray_dir, up));
            float sin_theta = dot(projected_dir, right);
            float cos_theta = dot(projected_dir, forward);
            azimuth = stm_atan2(sin_theta, cos_theta) + PI;
        }
        float u = azimuth / (2.0f * PI);
        float v = 0.5f + 0.5f * copysignf(sqrtf(fabsf(altitude) * 2.0f / PI), altitude);
        return sky_lut.sample(Vec2(u, v)).xyz();
    }
    static Vec3 evaluate_bloom(Vec3 ray_dir, Vec3 sun_dir) {
        const float SUN_SOLID_ANGLE = 0.53f * PI / 180.0f;
        float min_sun_cos_theta = stm_cos(SUN_SOLID_ANGLE);
        float cos_theta = dot(ray_dir, sun_dir);
        if (cos_theta >= min_sun_cos_theta) return Vec3::splat(1.0f);
        float offset = min_sun_cos_theta - cos_theta;
        float gaussian_bloom = stm_exp(-offset * 50000.0f) * 0.5f;
        float inv_bloom = 1.0f / (0.02f + offset * 300.0f) * 0.01f;
        return Vec3::splat(gaussian_bloom + inv_bloom);
    }
    static Vec3 interpolate_bloom(Vec3 bloom) {
        Vec3 t = vclamp((bloom - Vec3::splat(0.002f)) / (Vec3::splat(1.0f) - Vec3::splat(0.002f)), Vec3(), Vec3::splat(1.0f));
        return t * t * (Vec3::splat(3.0f) - 2.0f * t);
    }
    static Vec3 sample_lut(const LutTex& lut, Vec3 pos, Vec3 sun_dir) {
        float height = length(pos);
        Vec3 up = pos / height;
        float sun_cos_zenith_angle = dot(sun_dir, up);
        float u = saturate(0.5f + 0.5f * sun_cos_zenith_angle);
        float v = saturate((height - GROUND_RADIUS_MM) / (ATMOSPHERE_RADIUS_MM - GROUND_RADIUS_MM));
        return lut.sample(Vec2(u, v)).xyz();
    }
    Vec3 sample(Vec3 sun_dir, Vec3 ray_dir) const {  // atmosphere.rs:86-106
        Vec3 lum = sample_sky_lut(ray_dir, sun_dir);
        Vec3 sun_lum = evaluate_bloom(ray_dir, sun_dir);
        sun_lum = interpolate_bloom(sun_lum);
        if (length_squared(sun_lum) > 0.0f) {
            Ray ray = Ray::make(view_pos(), ray_dir);
            if (ray.intersect_sphere(GROUND_RADIUS_MM) >= 0.0f) sun_lum = Vec3();
            else sun_lum *= sample_lut(transmittance_lut, view_pos(), sun_dir);
        }
        lum += sun_lum;
        lum *= EXPOSURE;
        return lum;
    }
};

// ------------------------------------------------ reprojection.rs:5-78, utils/bilinear_filter.rs
struct Reprojection {
    float prev_x, prev_y, confidence; uint32_t validity;
    Reprojection() : prev_x(0), prev_y(0), confidence(0), validity(0) {}
    Vec4 serialize() const { return Vec4(prev_x, prev_y, confidence, b2f(validity)); }
    static Reprojection deserialize(Vec4 d) { Reprojection r; r.prev_x = d.x; r.prev_y = d.y; r.confidence = d.z; r.validity = f2b(d.w); return r; }
    bool is_some() const { return confidence > 0.0f; }
    Vec2 prev_pos() const { return Vec2(prev_x, prev_y); }
    UVec2 prev_pos_round() const { return as_uvec2(round(prev_pos())); }
    bool is_exact() const { return length_squared(fract_floor(prev_pos())) == 0.0f; }
};
static inline void reprojection_coords(float px, float py, IVec2 out[4]) {
    out[0] = IVec2(f2i_sat(floorf(px)), f2i_sat(floorf(py)));
    out[1] = IVec2(f2i_sat(ceilf(px)), f2i_sat(floorf(py)));
    out[2] = IVec2(f2i_sat(floorf(px)), f2i_sat(ceilf(py)));
    out[3] = IVec2(f2i_sat(ceilf(px)), f2i_sat(ceilf(py)));
}
// BilinearFilter::reproject with a plane sampler (weight 1.0 per valid tap).
// Deviation (documented): taps outside the plane read as zero (the reference
// performs an unchecked storage-image read, which Vulkan defines as zero).
template <class F>
static inline Vec4 bilinear_reproject(const Reprojection& r, F sample) {
    if (r.is_exact()) return sample(r.prev_pos_round());
    Vec4 s[4]; float w[4] = {0, 0, 0, 0};
    IVec2 p[4];
    reprojection_coords(r.prev_x, r.prev_y, p);
    for (int i = 0; i < 4; i++)
        if ((r.validity & (1u << i)) > 0 && p[i].x >= 0 && p[i].y >= 0) { s[i] = sample(as_uvec2(p[i])); w[i] = 1.0f; }
    Vec2 uv(fract_trunc(r.prev_x), fract_trunc(r.prev_y));
    Vec4 weights = Vec4(w[0], w[1], w[2], w[3]) *
                   Vec4((1.0f - uv.x) * (1.0f - uv.y), uv.x * (1.0f - uv.y), (1.0f - uv.x) * uv.y, uv.x * uv.y);
    float w_sum = dot(weights, Vec4(1, 1, 1, 1));
    if (w_sum == 0.0f) return Vec4();
    return (s[0] * weights.x + s[1] * weights.y + s[2] * weights.z + s[3] * weights.w) / w_sum;
}

// ------------------------------------------------ reservoir.rs:3-79, reservoir/{di,gi,ephemeral,mis}.rs
struct DiSample {
    float pdf, confidence; uint32_t light_id; Vec3 light_point; bool is_occluded;
    DiSample() : pdf(0), confidence(0), light_id(0), is_occluded(false) {}
    float pdf_ex(const Light& light, Hit hit) const {
        hit.gbuffer.base_color = Vec4(1, 1, 1, 1);
        if (!light.is_none() && light.contains(light_point)) return luma(light.radiance(hit).sum());
        return 0.0f;
    }
    float pdf_curr(const LightsView& lights, const Hit& hit) const { return pdf_ex(lights.get(light_id), hit); }
    float pdf_prev(const LightsView& lights, const Hit& hit) const { return pdf_ex(lights.get_prev(light_id), hit); }
    Ray ray(Vec3 hit_point) const {
        Vec3 dir = hit_point - light_point;
        return Ray::make(light_point, normalize(dir)).with_len(length(dir));
    }
};
struct GiSample {
    float pdf; uint32_t rng; Vec3 radiance, v1_point, v2_point, v2_normal;
    GiSample() : pdf(0), rng(0) {}
    bool exists() const { return v2_point != Vec3(); }
    Vec3 dir(Vec3 point) const { return normalize(v2_point - point); }
    float cosine(const Hit& hit) const { return fmax_(dot(dir(hit.point), hit.gbuffer.normal), 0.0f); }
    Vec3 diff_brdf(const Hit& hit) const { return diffuse_brdf_eval(hit.gbuffer); }
    Vec3 spec_brdf(const Hit& hit) const { return specular_brdf_eval(hit.gbuffer, dir(hit.point), -hit.dir); }
    float pdf_at(Hit hit) const {
        if (!exists()) return 0.0f;
        hit.gbuffer.base_color = Vec4(1, 1, 1, 1);
        float d = luma(diff_brdf(hit));
        float s = luma(spec_brdf(hit));
        return luma(radiance) * cosine(hit) * (d + s);
    }
    Ray ray(Vec3 hit_point) const { return Ray::make(hit_point, dir(hit_point)).with_len(distance(v2_point, hit_point) - 0.01f); }
    void partial_jacobian(Vec3 hit_point, float* dist, float* cosv) const {
        Vec3 vec = hit_point - v2_point;
        *dist = length(vec);
        *cosv = saturate(dot(v2_normal, vec / *dist));
    }
    float jacobian(Vec3 new_hit_point) const {
        if (!exists()) return 1.0f;
        float nd, nc, od, oc;
        partial_jacobian(new_hit_point, &nd, &nc);
        partial_jacobian(v1_point, &od, &oc);
        float x = nc * od * od, y = oc * nd * nd;
        return y == 0.0f ? 0.0f : x / y;
    }
};
struct EphemeralSample {
    uint32_t light_id; LightRadiance light_rad;
    EphemeralSample() : light_id(0) {}
    float pdf() const { return perc_luma(light_rad.radiance); }
};

template <class T>
struct Reservoir {
    T sample; float m, w;
    Reservoir() : m(0), w(0) {}
    bool update(WhiteNoise& wn, const T& s, float weight) {
        m += 1.0f; w += weight;
        if (wn.sample() * w < weight) { sample = s; return true; }
        return false;
    }
    bool merge(WhiteNoise& wn, const Reservoir& s, float pdf) {
        if (s.m <= 0.0f) return false;
        m += s.m - 1.0f;
        return update(wn, s.sample, s.w * s.m * pdf);
    }
    void clamp_m(float mx) { m = fmin_(m, mx); }
    void clamp_w(float mx) { w = fmin_(w, mx); }
    void norm(float pdf, float num, float denom_) { float denom = pdf * denom_; w = denom == 0.0f ? 0.0f : (w * num) / denom; }
    void norm_avg(float pdf) { norm(pdf, 1.0f, m); }
    void norm_mis(float pdf) { norm(pdf, 1.0f, 1.0f); }
    bool is_empty() const { return m == 0.0f; }
};

struct DiReservoir : Reservoir<DiSample> {
    // Deviation (documented): `id` beyond `count` reads as an empty reservoir
    // (the reference indexes unchecked; see di_spatial_resampling.rs:262).
    static DiReservoir read(const Vec4* buf, size_t id, size_t count) {
        DiReservoir r;
        if (id >= count) return r;
        Vec4 d0 = buf[2 * id], d1 = buf[2 * id + 1];
        uint32_t by[4]; u32_to_bytes(f2b(d0.w), by);
        r.sample.pdf = d0.z; r.sample.confidence = (float)by[1]; r.sample.light_id = f2b(d1.w);
        r.sample.light_point = d1.xyz(); r.sample.is_occluded = by[0] > 0;
        r.m = d0.x; r.w = d0.y;
        return r;
    }
    void write(Vec4* buf, size_t id) const {
        buf[2 * id] = Vec4(m, w, sample.pdf, b2f(u32_from_bytes(sample.is_occluded ? 1u : 0u, f2u_sat(sample.confidence), 0, 0)));
        buf[2 * id + 1] = Vec4(sample.light_point, b2f(sample.light_id));
    }
};
struct GiReservoir : Reservoir<GiSample> {
    float confidence;
    GiReservoir() : confidence(0) {}
    static GiReservoir read(const Vec4* buf, size_t id, size_t count) {
        GiReservoir r;
        if (id >= count) return r;
        Vec4 d0 = buf[4 * id], d1 = buf[4 * id + 1], d2 = buf[4 * id + 2], d3 = buf[4 * id + 3];
        r.sample.pdf = d2.w; r.sample.rng = f2b(d3.w); r.sample.radiance = d0.xyz(); r.sample.v1_point = d1.xyz();
        r.sample.v2_point = d2.xyz(); r.sample.v2_normal = normal_decode(d3.xy());
        r.m = d0.w; r.w = d1.w; r.confidence = d3.z;
        return r;
    }
    void write(Vec4* buf, size_t id) const {
        buf[4 * id] = Vec4(sample.radiance, m);
        buf[4 * id + 1] = Vec4(sample.v1_point, w);
        buf[4 * id + 2] = Vec4(sample.v2_point, sample.pdf);
        Vec2 n = normal_encode(sample.v2_normal);
        buf[4 * id + 3] = Vec4(n.x, n.y, confidence, b2f(sample.rng));
    }
};
struct EphemeralReservoir : Reservoir<EphemeralSample> {
    static EphemeralReservoir build(WhiteNoise& wn, const LightsView& lights, const World& world, const Hit& hit) {  // ephemeral.rs:14-55
        EphemeralReservoir res; float res_pdf = 0.0f;
        uint32_t max_samples = world.light_count < 16 ? world.light_count : 16;
        float sample_ipdf = (float)world.light_count;
        for (uint32_t nth = 0; nth < max_samples; nth++) {
            EphemeralSample s;
            s.light_id = wn.sample_int() % world.light_count;
            s.light_rad = lights.get(s.light_id).radiance(hit);
            float sample_pdf = s.pdf();
            if (res.update(wn, s, sample_pdf * sample_ipdf)) res_pdf = sample_pdf;
        }
        res.norm_avg(res_pdf);
        return res;
    }
};

struct MisResult { float m, lhs_pdf, lhs_mis, rhs_pdf, rhs_mis; };
struct Mis {  // reservoir/mis.rs:11-144
    float lhs_m, rhs_m, rhs_jacobian, lhs_lhs_pdf, lhs_rhs_pdf, rhs_lhs_pdf, rhs_rhs_pdf;
    static float mis2(float x, float y) { float sum = x + y; return sum == 0.0f ? 0.0f : x / sum; }
    static float mfac(float q0, float q1) { return q0 <= 0.0f ? 1.0f : saturate(stm_pow8(fmin_(q1 / q0, 1.0f))); }
    MisResult eval() const {
        MisResult r;
        r.m = rhs_m * fmin_(mfac(rhs_rhs_pdf, rhs_lhs_pdf), mfac(lhs_rhs_pdf, lhs_lhs_pdf));
        float t = mis2(lhs_m, rhs_m);
        r.lhs_mis = t + (1.0f - t) * mis2(lhs_m * lhs_lhs_pdf, rhs_m * lhs_rhs_pdf);
        r.rhs_mis = (1.0f - t) * mis2(rhs_m * rhs_rhs_pdf * rhs_jacobian, lhs_m * rhs_lhs_pdf);
        r.lhs_pdf = lhs_lhs_pdf; r.rhs_pdf = rhs_lhs_pdf;
        return r;
    }
    static Mis di_temporal(const LightsView& lights, const DiReservoir& lhs, const Hit& lhs_hit, const DiReservoir& rhs, const Hit& rhs_hit, bool rhs_killed) {
        Mis s;
        s.lhs_rhs_pdf = ((lhs.m > 0.0f) & rhs_hit.is_some()) ? lhs.sample.pdf_prev(lights, rhs_hit) : 0.0f;
        s.rhs_lhs_pdf = ((rhs.m > 0.0f) & !rhs_killed) ? rhs.sample.pdf_curr(lights, lhs_hit) : 0.0f;
        s.lhs_m = lhs.m; s.rhs_m = rhs.m; s.rhs_jacobian = 1.0f; s.lhs_lhs_pdf = lhs.sample.pdf; s.rhs_rhs_pdf = rhs.sample.pdf;
        return s;
    }
    static Mis gi_temporal(const GiReservoir& lhs, const Hit& lhs_hit, const GiReservoir& rhs, const Hit& rhs_hit) {
        Mis s;
        s.lhs_rhs_pdf = ((lhs.m > 0.0f) & rhs_hit.is_some()) ? lhs.sample.pdf_at(rhs_hit) : 0.0f;
        s.rhs_lhs_pdf = (rhs.m > 0.0f) ? rhs.sample.pdf_at(lhs_hit) : 0.0f;
        s.lhs_m = lhs.m; s.rhs_m = rhs.m; s.rhs_jacobian = 1.0f; s.lhs_lhs_pdf = lhs.sample.pdf; s.rhs_rhs_pdf = rhs.sample.pdf;
        return s;
    }
};

}  // namespace orc
