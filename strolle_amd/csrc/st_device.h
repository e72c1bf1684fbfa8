// st_device.h — device-side library of the MI355X hot path: primary-ray generation, BVH traversal with a
// per-wave LDS stack, Möller–Trumbore with deferred attribute fetch, G-buffer / normal / reservoir codecs,
// BRDFs, lights, MIS, atmosphere lookup. Behavioural contract: strolle-gpu/src/*.rs (cited per block);
// data layout and control flow are this library's own.
#pragma once
#include "st_types.h"

namespace st {

#define ST_D __device__ __forceinline__

// ------------------------------------------------------------------ launch geometry
// One wavefront == one 8x8 pixel tile (64 lanes); a 256-thread block owns four horizontally adjacent tiles.
// Blocks are dealt to XCDs round-robin by the hardware (block b -> XCD b % 8); the remap below gives every
// XCD one contiguous band of tile rows so that neighbour taps (spatial resampling, à-trous) of a tile stay
// in the same XCD's L2.
struct TileCoord { uint32_t x, y; bool valid; };
// map_mode 0: XCD-banded (each XCD owns a contiguous band of tile rows); 1: hardware order (block b -> XCD b % 8,
// consecutive blocks are horizontal neighbours); 2: XCD-banded in chunks of 4 tile rows (locality for short taps,
// balance for spatially clustered slow paths).
// (`b`: the workgroup's number in the launch that covers the window with one workgroup per four tiles — blockIdx.x, or a number a kernel whose
// workgroups each take several of those derives from it)
ST_D TileCoord tile_for_block(uint32_t b, uint32_t tiles_x, uint32_t tiles_y, uint32_t map_mode) {
    const uint32_t groups_x = (tiles_x + 3u) >> 2;              // blocks per tile row
    const uint32_t n_blocks = groups_x * tiles_y;
    uint32_t lin = b;
    if (map_mode == 0u) {
        const uint32_t q = n_blocks >> 3, r = n_blocks & 7u;    // bijective XCD remap (handles n % 8 != 0)
        const uint32_t xcd = b & 7u, k = b >> 3;
        lin = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + k;
    } else if (map_mode == 2u) {
        const uint32_t chunk = groups_x * 4u * 8u;              // 8 XCDs x 4 tile rows
        const uint32_t full = (n_blocks / chunk) * chunk;
        if (b < full) {
            const uint32_t c = b / chunk, i = b - c * chunk;
            const uint32_t xcd = i & 7u, k = i >> 3;            // k in [0, groups_x*4)
            lin = c * chunk + xcd * (groups_x * 4u) + k;
        }
    }
    const uint32_t wave = threadIdx.x >> 6;
    TileCoord t;
    t.y = lin / groups_x;
    t.x = (lin - t.y * groups_x) * 4u + wave;
    t.valid = b < n_blocks && t.x < tiles_x && t.y < tiles_y;
    return t;
}
ST_D TileCoord tile_for_thread(uint32_t tiles_x, uint32_t tiles_y, uint32_t map_mode) { return tile_for_block(blockIdx.x, tiles_x, tiles_y, map_mode); }
ST_D U2 pixel_in_tile(TileCoord t) {
    const uint32_t lane = threadIdx.x & 63u;
    return u2(t.x * 8u + (lane & 7u), t.y * 8u + (lane >> 3));
}
// Traversal stack in LDS, [wave][entry][lane] (consecutive lanes -> consecutive banks). Entries are BVH stream indices:
// entry numbers (byte offset / 64): device streams of fewer than 65536 entries (Cornell: 55, dungeon: ~26 k) use 16-bit
// stack slots, which halves the LDS footprint
// (12 KiB per 4-wave block) and lifts the LDS cap on occupancy from 6 to 8 waves per SIMD.
// The stack holds KArgs::stack_entries pending entries per lane: kBvhStackSize (24, strolle-gpu/src/lib.rs:76) or — when the tree's deepest chain of
// internal nodes is longer than that (the 208 k-triangle dungeon: 26) — kBvhStackSizeDeep (32), so that no push is ever dropped. It lives in
// DYNAMIC LDS sized at the launch (k_common.h ST_STACK_LDS / ST_LAUNCH): the 24-entry launches keep their occupancy.
template <class SE>
ST_D SE* lane_stack(const KArgs& a, SE* lds) { return lds + (threadIdx.x >> 6) * (a.stack_entries * 64u) + (threadIdx.x & 63u); }
// Per-kernel counters {rays traced, the reference's `used_memory` bytes}. One returning-free atomic pair per
// wavefront (hipcc's atomic optimizer reduces the uniform-address adds across the wave) lands on one of
// kCounterLines 64-byte lines picked by block id: a single hot word saturates near 88 atomics/us
// (MI355X_MICROARCH.md, row "dequeue"), which alone cost 0.7 ms per full-screen launch at 1080p.

// The byte half is off unless st_profile_enable asked for it (KArgs::count_bytes): summing a per-lane value takes a 64-bit
// cross-lane reduction per call (measured: 12-13 us of primary visibility's 89), counting rays takes a ballot.
ST_D void count_rays_n(const KArgs& a, uint32_t rays /* 0..3 */, unsigned long long used_memory) {
    unsigned long long* line = a.ray_counter + (blockIdx.x & (kCounterLines - 1u)) * 8u;
    if (rays & 1u) atomicAdd(line, 1ull);
    if (rays & 2u) atomicAdd(line, 2ull);
    if (a.count_bytes) atomicAdd(line + 1, used_memory);
}
// any number of rays per lane (the persistent-wave shadow kernel)
ST_D void count_rays_many(const KArgs& a, uint32_t rays, unsigned long long used_memory) {
    unsigned long long* line = a.ray_counter + (blockIdx.x & (kCounterLines - 1u)) * 8u;
    atomicAdd(line, (unsigned long long)rays);
    if (a.count_bytes) atomicAdd(line + 1, used_memory);
}
ST_D void count_rays(const KArgs& a, uint32_t used_memory) {
    unsigned long long* line = a.ray_counter + (blockIdx.x & (kCounterLines - 1u)) * 8u;
    atomicAdd(line, 1ull);
    if (a.count_bytes) atomicAdd(line + 1, (unsigned long long)used_memory);
}

// ------------------------------------------------------------------ small codecs
ST_D uint32_t u32_from_bytes(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return a | (b << 8) | (c << 16) | (d << 24); }
ST_D U2 resolve_checkerboard(U2 gid, uint32_t frame) { return u2(gid.x * 2u + ((frame + gid.y) % 2u), gid.y); }  // utils.rs:33-35
ST_D U2 resolve_checkerboard_alt(U2 gid, uint32_t frame) { return resolve_checkerboard(gid, frame + 1u); }
ST_D bool got_checkerboard_at(U2 p, uint32_t frame) { U2 q = resolve_checkerboard(u2(p.x / 2u, p.y), frame); return q.x == p.x && q.y == p.y; }

ST_D V2 normal_encode(V3 n) {  // normal.rs:9-24
    n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));  // V3 / float: one reciprocal in the fast build
    V2 r;
    if (n.z >= 0.0f) r = v2(n.x, n.y);
    else { r = v2(copysignf(1.0f - fabsf(n.y), n.x), copysignf(1.0f - fabsf(n.x), n.y)); }
    return r * 0.5f + 0.5f;
}
ST_D V3 normal_decode(V2 e) {  // normal.rs:26-34
    e = e * 2.0f - 1.0f;
    V3 n = v3(e.x, e.y, 1.0f - fabsf(e.x) - fabsf(e.y));
    const float t = fmax_(-n.z, 0.0f);
    n.x -= copysignf(t, n.x);
    n.y -= copysignf(t, n.y);
    return normalize(n);
}

// ------------------------------------------------------------------ noise (noise/white.rs, noise/blue.rs)
struct WhiteNoise {
    uint32_t state;
    ST_D uint32_t sample_int() {
        state = state * 747796405u + 2891336453u;
        const uint32_t word = ((state >> ((state >> 28) + 4u)) ^ state) * 277803737u;
        return (word >> 22) ^ word;
    }
    ST_D float sample() { return (float)sample_int() * 2.3283064365386963e-10f; }  // / 2^32: a power of two, exact either way
    ST_D V2 sample_circle() { const float angle = sample() * kPi * 2.0f; float s, c; sincos_(angle, &s, &c); return v2(c, s); }
    ST_D V2 sample_disk() { const float radius = fsqrt(sample()); return sample_circle() * radius; }
    ST_D V3 sample_sphere() {
        const float phi = sample() * 2.0f * kPi;
        const float cos_theta = sample() * 2.0f - 1.0f;
        const float u = sample();
        const float theta = acos_(cos_theta);
        const float r = fsqrt(u);
        float st_, ct_, sp_, cp_;
        sincos_(theta, &st_, &ct_); sincos_(phi, &sp_, &cp_);
        return v3(r * st_ * cp_, r * st_ * sp_, r * ct_);
    }
    ST_D V3 sample_hemisphere(V3 normal) {
        const float cos_theta = sample();
        const float sin_theta = fsqrt(1.0f - sqr(cos_theta));
        const float phi = 2.0f * kPi * sample();
        V3 t, b;
        any_orthonormal_pair(normal, &t, &b);
        float sp_, cp_;
#if defined(ST_ABL_HEMI_POLY)
        sincos_poly_(phi, &sp_, &cp_);
#else
        sincos_(phi, &sp_, &cp_);
#endif
        return (t * cp_ + b * sp_) * sin_theta + normal * cos_theta;
    }
};
ST_D WhiteNoise white_noise(uint32_t seed, U2 id) { WhiteNoise w; w.state = seed ^ (48619u * id.x) ^ (95461u * id.y); return w; }
ST_D float4 blue_noise_read(const KArgs& a, U2 id) {
    const uint32_t ux = (id.x + 71u * a.frame) % 256u, uy = (id.y + 11u * a.frame) % 256u;
    const uchar4 p = a.blue_noise[uy * 256u + ux];
    return make_float4(fdivc((float)p.x, 255.0f), fdivc((float)p.y, 255.0f), fdivc((float)p.z, 255.0f), fdivc((float)p.w, 255.0f));
}

// ------------------------------------------------------------------ rays & camera (ray.rs:14-53, camera.rs:19-150)
struct Ray { V3 origin, dir, inv_dir; float len; };

// ================================================================== exact island (1/3): ray generation
// Everything between the `contract(off)` pragmas is the same arithmetic in both builds of the kernels: correctly rounded
// + - * / sqrt, never contracted, no hardware approximations. It covers what decides which BVH nodes and triangles a ray
// visits — Camera::ray, the ray's reciprocal direction, intersect_box, Triangle::hit, the alpha test's texture fetch — so
// that the reference's `used_memory` counter (the BVH heatmap's integers) is bit-identical to the CPU restatement whichever
// build runs. Helpers used in here are defined in here (xe::): an inlined helper from outside would bring its own
// contraction setting with it.
#pragma clang fp contract(off)
namespace xe {
ST_D V3 add(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
ST_D V3 sub(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
ST_D V3 mul(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
ST_D V3 scale(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
ST_D float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
ST_D V3 cross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
ST_D V3 normalize(V3 a) { return scale(a, 1.0f / sqrtf(dot(a, a))); }
ST_D float4 scale4(float4 a, float s) { return make_float4(a.x * s, a.y * s, a.z * s, a.w * s); }
ST_D float4 add4(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
ST_D float4 sub4(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
ST_D float4 mul4(float4 a, float4 b) { return make_float4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
ST_D V3 project_point3(const M4& m, V3 p) {  // glam Mat4::project_point3
    float4 r = scale4(m.c[0], p.x);
    r = add4(scale4(m.c[1], p.y), r);
    r = add4(scale4(m.c[2], p.z), r);
    r = add4(m.c[3], r);
    const float rw = 1.0f / r.w;
    return v3(r.x * rw, r.y * rw, r.z * rw);
}
}  // namespace xe
ST_D Ray make_ray(V3 o, V3 d) { Ray r; r.origin = o; r.dir = d; r.inv_dir = v3(1.0f / d.x, 1.0f / d.y, 1.0f / d.z); r.len = kF32Max; return r; }
ST_D Ray zero_ray() { Ray r; r.origin = v3s(0.0f); r.dir = v3s(0.0f); r.inv_dir = v3s(0.0f); r.len = 0.0f; return r; }  // Ray::default()
ST_D V3 ray_at(const Ray& r, float t) { return r.origin + r.dir * t; }

ST_D Ray camera_ray(const GpuCamera& c, U2 pos) {
    const float spx = (float)pos.x + 0.5f, spy = (float)pos.y + 0.5f;
    const float ndc_x = spx * 2.0f / c.screen.x - 1.0f;
    const float ndc_y = -(spy * 2.0f / c.screen.y - 1.0f);
    const V3 far_plane = xe::project_point3(c.ndc_to_world, v3(ndc_x, ndc_y, kF32Eps));
    const V3 near_plane = xe::project_point3(c.ndc_to_world, v3(ndc_x, ndc_y, 1.0f));
    return make_ray(near_plane, xe::normalize(xe::sub(far_plane, near_plane)));
}
#if defined(ST_FAST_MATH)
#pragma clang fp contract(fast)
#endif
// ================================================================== end of exact island (1/3)
// Camera::ray for SHADING: the ray through a pixel whose G-buffer depth is being turned back into a surface point
// (Hit::from_gbuffer via pixel_hit below) or whose direction a BRDF needs. Nothing traverses the BVH with it, so in the fast
// build it is ordinary fast arithmetic (about a third of the instructions of the exact version above — and pixel_hit runs at
// 17 call sites, once per neighbour tap in the resampling loops); in the exact build it is the same operations as camera_ray.
ST_D Ray camera_ray_shading(const GpuCamera& c, U2 pos) {
    const V2 screen_size = v2(c.screen.x, c.screen.y);
    const V2 sp = as_v2(pos) + v2(0.5f, 0.5f);
    V2 ndc = sp * 2.0f / screen_size - v2(1.0f, 1.0f);
    ndc = v2(ndc.x, -ndc.y);
    const V3 far_plane = project_point3(c.ndc_to_world, v3(ndc.x, ndc.y, kF32Eps));
    const V3 near_plane = project_point3(c.ndc_to_world, v3(ndc.x, ndc.y, 1.0f));
    Ray r; r.origin = near_plane; r.dir = normalize(far_plane - near_plane); r.inv_dir = v3s(0.0f); r.len = kF32Max;  // inv_dir: never traversed
    return r;
}
ST_D float4 world_to_clip(const GpuCamera& c, V3 p) { return mul(c.projection_view, f4(p, 1.0f)); }
ST_D V2 clip_to_screen(const GpuCamera& c, float4 p) {
    V2 ndc = v2(p.x, p.y) / p.w;
    ndc = v2(ndc.x, -ndc.y);
    return (0.5f * ndc + 0.5f) * v2(c.screen.x, c.screen.y);
}
ST_D uint32_t screen_to_idx(const KArgs& a, U2 p) { return p.y * a.width + p.x; }
ST_D U2 camera_contain(const KArgs& a, I2 p) {  // camera.rs:53-75
    const int32_t sx = (int32_t)a.width, sy = (int32_t)a.height;
    if (p.x < 0) p.x = -p.x;
    if (p.y < 0) p.y = -p.y;
    if (p.x >= sx) p.x = sx - p.x + sx - 1;
    if (p.y >= sy) p.y = sy - p.y + sy - 1;
    return u2((uint32_t)p.x, (uint32_t)p.y);
}
ST_D bool contains_u(const KArgs& a, U2 p) { return p.x < a.width && p.y < a.height; }
ST_D bool contains_i(const KArgs& a, I2 p) { return p.x >= 0 && p.y >= 0 && p.x < (int32_t)a.width && p.y < (int32_t)a.height; }
ST_D bool contains_f(const KArgs& a, V2 p) { return p.x >= 0.0f && p.y >= 0.0f && p.x < (float)a.width && p.y < (float)a.height; }
ST_D bool camera_is_eq(const GpuCamera& a, const GpuCamera& b) {  // camera.rs:104-107
    bool ok = true;
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const float4 x = a.projection_view.c[i], y = b.projection_view.c[i];
        ok = ok && fabsf(x.x - y.x) <= 0.0025f && fabsf(x.y - y.y) <= 0.0025f && fabsf(x.z - y.z) <= 0.0025f && fabsf(x.w - y.w) <= 0.0025f;
    }
    return ok;
}
// plane access with storage-image semantics (out-of-range reads are zero, writes are dropped)
ST_D float4 tex_read(const float4* p, const KArgs& a, U2 pos) { return (pos.x < a.width && pos.y < a.height) ? p[pos.y * a.width + pos.x] : f4z(); }
ST_D void tex_write(float4* p, const KArgs& a, U2 pos, float4 v) { if (pos.x < a.width && pos.y < a.height) p[pos.y * a.width + pos.x] = v; }

// ------------------------------------------------------------------ materials & atlas (material.rs:25-104)
// Byte decodes are pure functions of 256 inputs; the engine tabulates them once on the device with these very routines
// (k_trace.hip k_build_byte_luts) so that a texel costs four table reads instead of three pow_() + a division — about
// 250 VALU operations per texel, twelve texels per textured primary hit.
// ================================================================== exact island (2/3): byte tables and atlas sampling
// The tables are built once per engine with the polynomial pow (identical in both builds); the bilinear fetch is what the
// traversal's alpha test reads (Triangle::hit's `base_color.w < 1`, ray.rs:184-214).
#pragma clang fp contract(off)
ST_D float srgb_to_linear_eval(uint32_t v) {
    const float c = (float)v / 255.0f;
    return c <= 0.04045f ? c / 12.92f : pow_poly_((c + 0.055f) / 1.055f, 2.4f);
}
ST_D float unorm8_eval(uint32_t v) { return (float)v / 255.0f; }
constexpr uint32_t kLutSrgb = 0u, kLutUnorm8 = 256u, kLutGamma8 = 512u, kLutGamma6 = 768u, kByteLutFloats = 1024u;  // layout of KArgs::byte_luts
ST_D float gamma8_eval(uint32_t v) { return pow_poly_((float)v / 255.0f, 2.2f); }  // G-buffer base colour RGB (gbuffer.rs:88-97)
ST_D float gamma6_eval(uint32_t v) { return pow_poly_((float)v / 63.0f, 2.2f); }   // ... and its 6-bit alpha
ST_D float srgb_to_linear(const KArgs& a, uint32_t v) { return a.byte_luts[kLutSrgb + v]; }
ST_D float unorm8(const KArgs& a, uint32_t v) { return a.byte_luts[kLutUnorm8 + v]; }
ST_D float4 atlas_texel(const KArgs& a, int32_t x, int32_t y) {
    x = x < 0 ? 0 : (x >= (int32_t)a.atlas_w ? (int32_t)a.atlas_w - 1 : x);
    y = y < 0 ? 0 : (y >= (int32_t)a.atlas_h ? (int32_t)a.atlas_h - 1 : y);
    const uchar4 p = a.atlas[(uint32_t)y * a.atlas_w + (uint32_t)x];
    return make_float4(srgb_to_linear(a, p.x), srgb_to_linear(a, p.y), srgb_to_linear(a, p.z), unorm8(a, p.w));
}
// gfx950 has no texture unit: bilinear, clamp-to-edge, lod 0, sRGB decode before filtering; NaN coordinates -> 0.
ST_D float4 atlas_sample(const KArgs& a, V2 uv) {
    if (a.atlas_w == 0u) return f4z();
    if (uv.x != uv.x) uv.x = 0.0f;
    if (uv.y != uv.y) uv.y = 0.0f;
    const float fx = uv.x * (float)a.atlas_w - 0.5f, fy = uv.y * (float)a.atlas_h - 0.5f;
    const float x0 = floorf(fx), y0 = floorf(fy);
    const float tx = fx - x0, ty = fy - y0;
    const int32_t ix = f2i_sat(x0), iy = f2i_sat(y0);
    const float4 p00 = atlas_texel(a, ix, iy), p10 = atlas_texel(a, ix + 1, iy), p01 = atlas_texel(a, ix, iy + 1), p11 = atlas_texel(a, ix + 1, iy + 1);
    const float4 top = xe::add4(p00, xe::scale4(xe::sub4(p10, p00), tx));
    const float4 bot = xe::add4(p01, xe::scale4(xe::sub4(p11, p01), tx));
    return xe::add4(top, xe::scale4(xe::sub4(bot, top), ty));
}
// fmodf(x, 1) for x >= 0 (material.rs:86-104 `% 1.0`): x - floor(x) is exact in f32 (the fraction of x needs no more
// significant bits than x has), NaN and +inf give NaN either way; libm's fmodf is a 40-instruction loop with two nested
// divergent branches, six times per textured hit.
ST_D float fmod1_nonneg(float x) { return x - floorf(x); }
ST_D float mat_wrap(float t) { return t > 0.0f ? fmod1_nonneg(t) : 1.0f - fmod1_nonneg(-t); }
ST_D float4 sample_atlas(const KArgs& a, V2 hit_uv, float4 multiplier, float4 texture) {
    if (is_zero(texture)) return multiplier;
    hit_uv.x = mat_wrap(hit_uv.x);
    hit_uv.y = mat_wrap(hit_uv.y);
    const V2 uv = v2(texture.x + hit_uv.x * texture.z, texture.y + hit_uv.y * texture.w);
    return xe::mul4(multiplier, atlas_sample(a, uv));
}
#if defined(ST_FAST_MATH)
#pragma clang fp contract(fast)
#endif
// ================================================================== end of exact island (2/3)

// ------------------------------------------------------------------ BVH traversal (ray.rs:114-302, triangle.rs:64-113)
struct TriangleHit { float distance; V3 point, normal; V2 uv; uint32_t material_id; uint32_t xform_slot; };  // xform_slot: owning instance (trace_closest only)
ST_D bool hit_is_some(const TriangleHit& h) { return h.distance < kF32Max; }

// ================================================================== exact island (3/3): traversal
#pragma clang fp contract(off)
ST_D float intersect_box(const Ray& r, V3 bmin, V3 bmax) {
    float tmin = 0.0f, tmax = kF32Max;
    const V3 t1 = xe::mul(xe::sub(bmin, r.origin), r.inv_dir);
    const V3 t2 = xe::mul(xe::sub(bmax, r.origin), r.inv_dir);
    tmin = fmax_(tmin, fmin_(t1.x, t2.x)); tmax = fmin_(tmax, fmax_(t1.x, t2.x));
    tmin = fmax_(tmin, fmin_(t1.y, t2.y)); tmax = fmin_(tmax, fmax_(t1.y, t2.y));
    tmin = fmax_(tmin, fmin_(t1.z, t2.z)); tmax = fmin_(tmax, fmax_(t1.z, t2.z));
    return tmin <= tmax ? tmin : kF32Max;
}
ST_D float intersect_sphere(const Ray& r, float radius) {  // ray.rs:304-321
    const float b = dot(r.origin, r.dir);
    const float c = dot(r.origin, r.origin) - radius * radius;
    if (c > 0.0f && b > 0.0f) return -1.0f;
    const float discr = b * b - c;
    if (discr < 0.0f) return -1.0f;
    if (discr > b * b) return -b + fsqrt(discr);
    return -b - fsqrt(discr);
}

struct Candidate { float t, u, v, inv_det; uint32_t tri, material; };
// Entry of the device BVH stream at BYTE offset `at` (traversal pointers are byte offsets: 64 per entry): base + a 32-bit
// offset, which the global-memory form turns into `global_load v, v_offset, s[base]` with immediate offsets for the four
// texels — no per-texel 64-bit address arithmetic (three VALU instructions per texel when indexed as bvh[ptr + k]).
ST_D const float4* bvh_entry(const float4* bvh, uint32_t at) { return reinterpret_cast<const float4*>(reinterpret_cast<const char*>(bvh) + at); }

ST_D V2 tri_uv(const KArgs& a, uint32_t tri, float u, float v) {
    const float4 q0 = a.tri_attr[4u * tri], q1 = a.tri_attr[4u * tri + 1u], q2 = a.tri_attr[4u * tri + 2u], q3 = a.tri_attr[4u * tri + 3u];
    const V2 uv0 = v2(q0.w, q1.w), uv1 = v2(q2.w, q3.x), uv2 = v2(q3.y, q3.z);
    return v2((uv0.x + (uv1.x - uv0.x) * u) + (uv2.x - uv0.x) * v, (uv0.y + (uv1.y - uv0.y) * u) + (uv2.y - uv0.y) * v);
}
// ANY_HIT: Tracing::ReturnFirst. Returns the reference's `used_memory` byte counter.
// On return `best.t` is the closest accepted distance (or the initial max_t if none).
// (Measured and dropped, round 3: ONE body per step chosen by a vote of the wave — lanes whose fetched entry is of the other kind keep it
// and wait, so that each body runs with more of its lanes: majority vote, triangles-only-when-no-node-is-pending ("while-while")
// and a 16-lane threshold all lost, dungeon 1080p 1.46 -> 1.57 / 1.60 / 1.58 ms per frame, GI sampling a 203 -> 235 / 303 / 268 us.
// The loop below already skips a body no lane wants (s_cbranch_execz); making lanes wait costs more steps than the fuller bodies save.)
template <bool ANY_HIT, class SE>
ST_D uint32_t traverse(const KArgs& a, const Ray& ray, float max_t, SE* stack, Candidate* best, bool* found_any) {
    best->t = max_t; best->tri = 0xffffffffu; best->material = 0u; best->u = 0.0f; best->v = 0.0f; best->inv_det = 1.0f;
    *found_any = false;
    if (a.bvh_len == 0u) return 0u;
    uint32_t used_memory = 0u;
    uint32_t ptr = 0u;
    int sp = 0;
    for (;;) {
        used_memory += 16u;
        // Every entry of the device stream is four texels (st_types.h "device BVH stream"): an internal node's two child
        // boxes, or a leaf entry with its triangle's hit-test record inline — one round trip per step for either kind
        // (dependent fetches are the traversal's latency chain). The empty asm keeps the four loads together: the compiler
        // otherwise sinks d1..d3 behind the d0.w test, which puts a second, dependent round trip into every step (measured
        // on the dungeon: 1.99 -> 1.87 ms/frame for the internal nodes alone).
        const float4* entry = bvh_entry(a.bvh, ptr);
        const float4 d0 = entry[0], d1 = entry[1], d2 = entry[2], d3 = entry[3];
        asm volatile("" :: "v"(d1.x), "v"(d2.x), "v"(d3.x));
        if (f2b(d0.w) == 0u) {
            used_memory += 48u;
            uint32_t near_ptr = ptr + 64u, far_ptr = f2b(d1.w);
            float near_d = intersect_box(ray, xyz(d0), xyz(d1));
            float far_d = intersect_box(ray, xyz(d2), xyz(d3));
            if (far_d < near_d) { const uint32_t tp = near_ptr; near_ptr = far_ptr; far_ptr = tp; const float td = near_d; near_d = far_d; far_d = td; }
            if (far_d < best->t) { if (sp < (int)a.stack_entries) { stack[sp * 64] = (SE)(far_ptr >> 6); sp++; } }
            if (near_d < best->t) { ptr = near_ptr; continue; }
        } else {
            used_memory += 144u;
            const uint32_t flags = f2b(d0.x), tri = f2b(d0.y), material = f2b(d0.z);
            const V3 p0 = xyz(d1), e1 = xyz(d2), e2 = xyz(d3);
            const V3 pvec = xe::cross(ray.dir, e2);
            const float det = xe::dot(e1, pvec);
            bool found = false;
            if (!(fabsf(det) < kF32Eps)) {
                const float inv_det = 1.0f / det;
                const V3 tvec = xe::sub(ray.origin, p0);
                const float u = xe::dot(tvec, pvec) * inv_det;
                const V3 qvec = xe::cross(tvec, e1);
                const float v = xe::dot(ray.dir, qvec) * inv_det;
                const float t = xe::dot(e2, qvec) * inv_det;
                if (!((u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (t <= 0.0f) | (t >= best->t))) {
                    found = true;
                    if (flags & 2u) {  // AlphaMode::Blend: the hit only counts where the base colour is opaque
                        used_memory += 112u + 16u;
                        const GpuMaterial m = a.materials[material];
                        const float4 bc = sample_atlas(a, tri_uv(a, tri, u, v), m.base_color, m.base_color_texture);
                        if (bc.w < 1.0f) found = false;
                    }
                    if (found) { best->t = t; best->u = u; best->v = v; best->inv_det = inv_det; best->tri = tri; best->material = material; *found_any = true; }
                }
            }
            if (found && ANY_HIT) break;
            if (flags & 1u) { ptr += 64u; continue; }
        }
        if (sp > 0) { sp--; ptr = (uint32_t)stack[sp * 64] << 6; } else break;
    }
    return used_memory;
}
// Ray::trace (closest hit) with attributes resolved once, for the winning triangle.
template <class SE> ST_D bool closest_hit_compact(const KArgs& a, const Ray& ray, SE* stack, Candidate* best);
template <class SE, bool EXACT_LEAF = false> ST_D bool closest_hit_wide(const KArgs& a, const Ray& ray, SE* stack, Candidate* best);
ST_D TriangleHit closest_resolve(const KArgs& a, const Ray& ray, const Candidate& c, bool any);
template <class SE>
ST_D TriangleHit trace_closest(const KArgs& a, const Ray& ray, SE* stack, uint32_t* used_memory) {
    Candidate c; bool any;
#if ST_FAST_DEVICE && !defined(ST_NO_ANYHIT_FAST)
    // the fast build's closest-hit rays outside the heatmap pass (which calls traverse() itself: its integers are the contract's) walk the
    // compact stream too when there is one: conservative boxes visit a superset of the entries, the triangle records are the same f32
    if (a.bvh_w != nullptr) { *used_memory = 0u; any = closest_hit_wide(a, ray, stack, &c); }
    else if (a.bvh_c != nullptr) { *used_memory = 0u; any = closest_hit_compact(a, ray, stack, &c); }
    else
#endif
    *used_memory = traverse<false>(a, ray, kF32Max, stack, &c, &any);
    return closest_resolve(a, ray, c, any);
}
// the winning triangle's attributes (normal, uv, instance slot), fetched once
ST_D TriangleHit closest_resolve(const KArgs& a, const Ray& ray, const Candidate& c, bool any) {
    TriangleHit h;
    h.distance = c.t; h.material_id = c.material; h.point = v3s(0.0f); h.normal = v3s(0.0f); h.uv = v2(0.0f, 0.0f); h.xform_slot = 0u;
    if (any) {
        const float4 q0 = a.tri_attr[4u * c.tri], q1 = a.tri_attr[4u * c.tri + 1u], q2 = a.tri_attr[4u * c.tri + 2u], q3 = a.tri_attr[4u * c.tri + 3u];
        V3 n = c.u * xyz(q1) + c.v * xyz(q2) + (1.0f - c.u - c.v) * xyz(q0);
        h.normal = normalize(n) * copysignf(1.0f, c.inv_det);
        const V2 uv0 = v2(q0.w, q1.w), uv1 = v2(q2.w, q3.x), uv2 = v2(q3.y, q3.z);
        h.uv = uv0 + (uv1 - uv0) * c.u + (uv2 - uv0) * c.v;
        h.xform_slot = f2b(q3.w);
    }
    if (hit_is_some(h)) h.point = ray_at(ray, h.distance);
    return h;
}
// glam Affine3A::transform_point3 with the transform stored as 4 float4 (x, y, z axes, translation)
ST_D V3 affine_point(const float4* m, V3 p) { return ((xyz(m[0]) * p.x) + (xyz(m[1]) * p.y) + (xyz(m[2]) * p.z)) + xyz(m[3]); }
// Triangle::hit's accept / reject and (t, u, v, 1 / det) for a hit-test record (p0, e1, e2), in the island's arithmetic: what traverse() computes for a leaf
// entry, for walks over OTHER streams that owe the contract walk's bits (primary rays over the wide stream: closest_hit_wide<SE, true>, closest_hit_packet)
ST_D bool triangle_hit_exact(const Ray& ray, V3 p0, V3 e1, V3 e2, float limit, float* t_out, float* u_out, float* v_out, float* inv_det_out) {
    const V3 pvec = xe::cross(ray.dir, e2);
    const float det = xe::dot(e1, pvec);
    if (fabsf(det) < kF32Eps) return false;
    const float inv_det = 1.0f / det;
    const V3 tvec = xe::sub(ray.origin, p0);
    const float u = xe::dot(tvec, pvec) * inv_det;
    const V3 qvec = xe::cross(tvec, e1);
    const float v = xe::dot(ray.dir, qvec) * inv_det;
    const float t = xe::dot(e2, qvec) * inv_det;
    *t_out = t; *u_out = u; *v_out = v; *inv_det_out = inv_det;
    return !((u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (t <= 0.0f) | (t >= limit));
}
// closest_resolve() and normal_encode() in the island's arithmetic, for PRIMARY hits (k_trace.hip k_prim_visibility; round 6). glam's
// Vec3::any_orthonormal_pair — behind every hemisphere sample (noise/white.rs:73-81) — branches on the SIGN of normal.z, and a wall whose normal lies in
// the xy-plane decodes to z = +-(an ulp): the fast build's last-bit differences in the G-buffer's encoded normal flipped that sign on 0.2 % of the
// dungeon's pixels, each flip a completely different bounce direction — the largest single consumer of the fast build's tolerance gates
// (profiles/r06_gate_headroom.json). With the primary hit's (u, v) from triangle_hit_exact and these two, the encoded normal is the CPU restatement's bit for bit
// wherever the same triangle wins.
ST_D TriangleHit closest_resolve_exact(const KArgs& a, const Ray& ray, const Candidate& c, bool any) {
    TriangleHit h;
    h.distance = c.t; h.material_id = c.material; h.point = v3s(0.0f); h.normal = v3s(0.0f); h.uv = v2(0.0f, 0.0f); h.xform_slot = 0u;
    if (any) {
        const float4 q0 = a.tri_attr[4u * c.tri], q1 = a.tri_attr[4u * c.tri + 1u], q2 = a.tri_attr[4u * c.tri + 2u], q3 = a.tri_attr[4u * c.tri + 3u];
        const V3 n = xe::add(xe::add(xe::scale(xyz(q1), c.u), xe::scale(xyz(q2), c.v)), xe::scale(xyz(q0), (1.0f - c.u) - c.v));
        h.normal = xe::scale(xe::normalize(n), copysignf(1.0f, c.inv_det));
        const float u0x = q0.w, u0y = q1.w, u1x = q2.w, u1y = q3.x, u2x = q3.y, u2y = q3.z;
        h.uv = v2((u0x + (u1x - u0x) * c.u) + (u2x - u0x) * c.v, (u0y + (u1y - u0y) * c.u) + (u2y - u0y) * c.v);
        h.xform_slot = f2b(q3.w);
    }
    if (hit_is_some(h)) h.point = xe::add(ray.origin, xe::scale(ray.dir, h.distance));
    return h;
}
ST_D V2 normal_encode_exact(V3 n) {  // normal.rs:9-24
    const float s = (fabsf(n.x) + fabsf(n.y)) + fabsf(n.z);
    n = v3(n.x / s, n.y / s, n.z / s);
    V2 r;
    if (n.z >= 0.0f) r = v2(n.x, n.y);
    else r = v2(copysignf(1.0f - fabsf(n.y), n.x), copysignf(1.0f - fabsf(n.x), n.y));
    return v2(r.x * 0.5f + 0.5f, r.y * 0.5f + 0.5f);
}
// Ray::intersect (shadow ray) as the contract states it: the reference's visiting order, arithmetic and `used_memory` count
template <class SE>
ST_D bool trace_any_contract(const KArgs& a, const Ray& ray, SE* stack, uint32_t* used_memory) {
    Candidate c; bool any;
    *used_memory = traverse<true>(a, ray, ray.len, stack, &c, &any);
    return c.t < ray.len;
}
#if defined(ST_FAST_MATH)
#pragma clang fp contract(fast)
#endif
// ================================================================== end of exact island (3/3)

// ------------------------------------------------------------------ any-hit rays of the fast build
// `Ray::intersect` (strolle-gpu/src/ray.rs:84-112) hands back ONE boolean — "some triangle is hit closer than ray.len" — and
// nothing else of the traversal (its `used_memory` is observable only through st_profile_enable(ST_PROFILE_TRAVERSAL_BYTES),
// which switches back to the contract loop above). The fast build therefore walks shadow rays with ordinary fast arithmetic:
//   * the slab test is two FMAs per plane pair against a precomputed -origin * inv_dir (the exact island's (b - o) * inv is a
//     subtraction and a multiplication: 12 VALU instructions fewer per internal node). In position space the difference is
//     one ulp of the coordinate's magnitude; direction components are floored at 1e-20 in magnitude (slab_safe_dir says why);
//   * Möller–Trumbore is contracted into FMAs and divides by v_rcp_f32 (the island's IEEE division alone is 13 instructions);
//   * near-child-first order is KEPT: tools/packet_sim.py prices "left child first, no near/far sort" at +27 % loop bodies on
//     the dungeon's DI shadow rays (occluded rays find their occluder later) against the ~15 % of an internal step the sort costs;
//     a wave-wide packet walk of the same rays (one node per step for all lanes, scalar fetch) visits 1.8x (DI) to 3x (GI) MORE
//     entries than the per-lane loop executes bodies — rays of one 8x8 tile go to 4-5 different lights — and is not built.
//   * a WORLD-SPACE LAST-OCCLUDER TABLE (key = hash of the ray's origin cell and end-point cell; the leaf entry that ended an earlier
//     ray between the same two cells is tested first) was built on top of this loop and measured NEUTRAL — dungeon 1.4244 vs 1.4212
//     ms/frame without it, DI resolving 138.4 vs 141.6 us, DI spatial 117.6 vs 115.3: packet_sim.py's 0.74 hit rate per ray holds,
//     but a wave only ends when its last lane does, and the waves that end early were not the ones the frame waits for. It is in
//     tools/experiments/occluder_table.inc.
// The boolean can differ from the contract loop's only where a ray grazes a box or a triangle edge within an ulp; the fast
// build's launch-by-launch tolerance tests (tests/test_gpu_fast_*.py) bound how often. The exact build never comes here.
#if ST_FAST_DEVICE
#if defined(ST_ABL_MT_DIV)
#define ST_MT_RCP(det) (1.0f / (det))                  // (ablation build: Moeller-Trumbore divides by an IEEE reciprocal)
#else
#define ST_MT_RCP(det) __builtin_amdgcn_rcpf(det)
#endif
// A direction component of (nearly) zero would make that axis' planes inf - inf = NaN wherever the origin and the plane have the same sign;
// v_min / v_max return the other operand for a NaN, so the axis would either collapse to one plane (a box the ray runs inside gets rejected)
// or constrain nothing (max3 / min3 below: measured — a handful of axis-parallel GI rays per frame walked every box along their other axes
// and turned a 0.3 ms launch into 3 ms). A floor of 1e-20 keeps the products finite: the planes become (bound - origin) * 1e20, which
// rejects a slab the ray runs beside and leaves one it runs inside unconstrained, as it should be.
ST_D float slab_safe_dir(float d) { return fabsf(d) < 1e-20f ? copysignf(1e-20f, d) : d; }
ST_D float any_slab(V3 lo, V3 hi, V3 inv, V3 oi) {
    const float ax = fmaf(lo.x, inv.x, oi.x), bx = fmaf(hi.x, inv.x, oi.x);
    const float ay = fmaf(lo.y, inv.y, oi.y), by = fmaf(hi.y, inv.y, oi.y);
    const float az = fmaf(lo.z, inv.z, oi.z), bz = fmaf(hi.z, inv.z, oi.z);
    const float tmin = fmax_(fmax_(fmax_(0.0f, fmin_(ax, bx)), fmin_(ay, by)), fmin_(az, bz));
    const float tmax = fmin_(fmin_(fmin_(kF32Max, fmax_(ax, bx)), fmax_(ay, by)), fmax_(az, bz));
    return tmin <= tmax ? tmin : kF32Max;
}
// Triangle::hit's accept / reject for a ray that only asks "closer than limit?" (alpha test excluded)
ST_D bool any_triangle(const Ray& ray, V3 p0, V3 e1, V3 e2, float limit, float* u_out, float* v_out) {
    const V3 pvec = cross(ray.dir, e2);
    const float det = dot(e1, pvec);
    if (fabsf(det) < kF32Eps) return false;
    const float inv_det = ST_MT_RCP(det);
    const V3 tvec = ray.origin - p0;
    const float u = dot(tvec, pvec) * inv_det;
    const V3 qvec = cross(tvec, e1);
    const float v = dot(ray.dir, qvec) * inv_det;
    const float t = dot(e2, qvec) * inv_det;
    *u_out = u; *v_out = v;
    return !((u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (t <= 0.0f) | (t >= limit));
}
template <class SE>
ST_D bool any_hit_fast(const KArgs& a, const Ray& ray, SE* stack) {
    if (a.bvh_len == 0u) return false;
    const float limit = ray.len;
    const V3 inv = v3(__builtin_amdgcn_rcpf(slab_safe_dir(ray.dir.x)), __builtin_amdgcn_rcpf(slab_safe_dir(ray.dir.y)), __builtin_amdgcn_rcpf(slab_safe_dir(ray.dir.z)));
    const V3 oi = v3(-ray.origin.x * inv.x, -ray.origin.y * inv.y, -ray.origin.z * inv.z);
    // (Loop shape: ONE loop with `continue`s and a single exit, as traverse() has it. A first version returned from inside the loop;
    // the structurizer turned its exits into an inner and an outer loop, lanes waited for each other at the inner one's end, and the
    // dungeon frame went 1.520 -> 1.604 ms although every body had become cheaper.)
    uint32_t ptr = 0u;
    int sp = 0;
    bool hit = false;
    for (;;) {
        const float4* entry = bvh_entry(a.bvh, ptr);
        const float4 d0 = entry[0], d1 = entry[1], d2 = entry[2], d3 = entry[3];
        asm volatile("" :: "v"(d1.x), "v"(d2.x), "v"(d3.x));   // one round trip for the four texels (see traverse())
        if (f2b(d0.w) == 0u) {
            uint32_t near_ptr = ptr + 64u, far_ptr = f2b(d1.w);
            float near_d = any_slab(xyz(d0), xyz(d1), inv, oi);
            float far_d = any_slab(xyz(d2), xyz(d3), inv, oi);
            if (far_d < near_d) { const uint32_t tp = near_ptr; near_ptr = far_ptr; far_ptr = tp; const float td = near_d; near_d = far_d; far_d = td; }
            if (far_d < limit) { if (sp < (int)a.stack_entries) { stack[sp * 64] = (SE)(far_ptr >> 6); sp++; } }
            if (near_d < limit) { ptr = near_ptr; continue; }
        } else {
            const uint32_t flags = f2b(d0.x);
            float u, v;
            bool found = any_triangle(ray, xyz(d1), xyz(d2), xyz(d3), limit, &u, &v);
            if (found && (flags & 2u)) {  // AlphaMode::Blend: the texel decides (exact-island fetch, as in traverse())
                const uint32_t tri = f2b(d0.y), material = f2b(d0.z);
                const GpuMaterial m = a.materials[material];
                const float4 bc = sample_atlas(a, tri_uv(a, tri, u, v), m.base_color, m.base_color_texture);
                if (bc.w < 1.0f) found = false;
            }
            if (found) { hit = true; break; }
            if (flags & 1u) { ptr += 64u; continue; }
        }
        if (sp > 0) { sp--; ptr = (uint32_t)stack[sp * 64] << 6; } else break;
    }
    return hit;
}
// The same walk over the COMPACT stream (k_bvh.hip k_bvh_compact: 48-B entries, conservative f16 child boxes; KArgs::bvh_c): two texels
// per internal step, three per leaf step instead of four. The slab test reads the f16 planes straight into v_fma_mix_f32
// (f16 x f32 + f32): the narrower boxes cost no conversion. A child's kind travels with its pointer: `cur` and the stack entries are
// (entry << 1 | is a leaf entry).
ST_D float half_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
ST_D float half_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
// One child box of a compact internal entry: a word per axis, (lower bound | upper bound << 16). The ray's direction sign per axis says
// which of the two is the entry plane, so the word is rotated by 0 or 16 bits (`rot`, per ray) and its low half is the near plane, its
// high half the far plane: no min / max per axis, max3 / min3 over the axes.
struct RaySlabs { V3 inv, oi; uint32_t rx, ry, rz; };
ST_D RaySlabs ray_slabs(const Ray& ray) {
    RaySlabs r;
    r.inv = v3(__builtin_amdgcn_rcpf(slab_safe_dir(ray.dir.x)), __builtin_amdgcn_rcpf(slab_safe_dir(ray.dir.y)), __builtin_amdgcn_rcpf(slab_safe_dir(ray.dir.z)));
    r.oi = v3(-ray.origin.x * r.inv.x, -ray.origin.y * r.inv.y, -ray.origin.z * r.inv.z);
    r.rx = (f2b(r.inv.x) >> 31) << 4; r.ry = (f2b(r.inv.y) >> 31) << 4; r.rz = (f2b(r.inv.z) >> 31) << 4;
    return r;
}
ST_D float compact_slab(uint32_t wx, uint32_t wy, uint32_t wz, const RaySlabs& r) {
    wx = __builtin_amdgcn_alignbit(wx, wx, r.rx); wy = __builtin_amdgcn_alignbit(wy, wy, r.ry); wz = __builtin_amdgcn_alignbit(wz, wz, r.rz);
    const float nx = fmaf(half_lo(wx), r.inv.x, r.oi.x), fx = fmaf(half_hi(wx), r.inv.x, r.oi.x);
    const float ny = fmaf(half_lo(wy), r.inv.y, r.oi.y), fy = fmaf(half_hi(wy), r.inv.y, r.oi.y);
    const float nz = fmaf(half_lo(wz), r.inv.z, r.oi.z), fz = fmaf(half_hi(wz), r.inv.z, r.oi.z);
    const float tmin = fmax_(fmax_(fmax_(nx, ny), nz), 0.0f);
    const float tmax = fmin_(fmin_(fx, fy), fz);
    return tmin <= tmax ? tmin : kF32Max;
}
template <class SE>
ST_D bool any_hit_compact(const KArgs& a, const Ray& ray, SE* stack) {
    if (a.bvh_len == 0u) return false;
    const float limit = ray.len;
    const RaySlabs rs = ray_slabs(ray);
    uint32_t cur = a.bvh_c_root;
    int sp = 0;
    bool hit = false;
    for (;;) {
        const bool leaf = (cur & 1u) != 0u;
        const float4* e = bvh_entry(a.bvh_c, __umul24(cur >> 1, 48u));   // v_mul_u32_u24: full rate, the 32-bit multiply is not
        const float4 t0 = e[0], t1 = e[1];
        float4 t2 = f4z();
        if (leaf) t2 = e[2];
        asm volatile("" :: "v"(t0.x), "v"(t1.x), "v"(t2.x));   // one round trip for the entry
        if (!leaf) {
            const uint32_t link = f2b(t1.z);
            float near_d = compact_slab(f2b(t0.x), f2b(t0.y), f2b(t0.z), rs);
            float far_d = compact_slab(f2b(t0.w), f2b(t1.x), f2b(t1.y), rs);
            uint32_t near_ptr = (((cur >> 1) + 1u) << 1) | (link & 1u), far_ptr = ((link >> 2) << 1) | ((link >> 1) & 1u);
            if (far_d < near_d) { const uint32_t tp = near_ptr; near_ptr = far_ptr; far_ptr = tp; const float td = near_d; near_d = far_d; far_d = td; }
            if (far_d < limit) { if (sp < (int)a.stack_entries) { stack[sp * 64] = (SE)far_ptr; sp++; } }
            if (near_d < limit) { cur = near_ptr; continue; }
        } else {
            const uint32_t head = f2b(t0.w);
            float u, v;
            bool found = any_triangle(ray, xyz(t0), xyz(t1), xyz(t2), limit, &u, &v);
            if (found && (head & 2u)) {  // AlphaMode::Blend: the texel decides (exact-island fetch, as in traverse())
                const GpuMaterial m = a.materials[f2b(t1.w)];
                const float4 bc = sample_atlas(a, tri_uv(a, head >> 2, u, v), m.base_color, m.base_color_texture);
                if (bc.w < 1.0f) found = false;
            }
            if (found) { hit = true; break; }
            if (head & 1u) { cur += 2u; continue; }   // the next entry of the run: a leaf entry too
        }
        if (sp > 0) { sp--; cur = (uint32_t)stack[sp * 64]; } else break;
    }
    return hit;
}
// Closest hit over the compact stream: the same loop with the cut-off following the best distance (triangle arithmetic contracted, as in
// any_triangle; Candidate as traverse() fills it, so trace_closest resolves attributes the same way).
template <class SE>
ST_D bool closest_hit_compact(const KArgs& a, const Ray& ray, SE* stack, Candidate* best) {
    best->t = kF32Max; best->tri = 0xffffffffu; best->material = 0u; best->u = 0.0f; best->v = 0.0f; best->inv_det = 1.0f;
    if (a.bvh_len == 0u) return false;
    const RaySlabs rs = ray_slabs(ray);
    uint32_t cur = a.bvh_c_root;
    int sp = 0;
    bool found_any = false;
    for (;;) {
        const bool leaf = (cur & 1u) != 0u;
        const float4* e = bvh_entry(a.bvh_c, __umul24(cur >> 1, 48u));   // v_mul_u32_u24: full rate, the 32-bit multiply is not
        const float4 t0 = e[0], t1 = e[1];
        float4 t2 = f4z();
        if (leaf) t2 = e[2];
        asm volatile("" :: "v"(t0.x), "v"(t1.x), "v"(t2.x));
        if (!leaf) {
            const uint32_t link = f2b(t1.z);
            float near_d = compact_slab(f2b(t0.x), f2b(t0.y), f2b(t0.z), rs);
            float far_d = compact_slab(f2b(t0.w), f2b(t1.x), f2b(t1.y), rs);
            uint32_t near_ptr = (((cur >> 1) + 1u) << 1) | (link & 1u), far_ptr = ((link >> 2) << 1) | ((link >> 1) & 1u);
            if (far_d < near_d) { const uint32_t tp = near_ptr; near_ptr = far_ptr; far_ptr = tp; const float td = near_d; near_d = far_d; far_d = td; }
            if (far_d < best->t) { if (sp < (int)a.stack_entries) { stack[sp * 64] = (SE)far_ptr; sp++; } }
            if (near_d < best->t) { cur = near_ptr; continue; }
        } else {
            const uint32_t head = f2b(t0.w);
            const V3 p0 = xyz(t0), e1 = xyz(t1), e2 = xyz(t2);
            const V3 pvec = cross(ray.dir, e2);
            const float det = dot(e1, pvec);
            if (!(fabsf(det) < kF32Eps)) {
                const float inv_det = ST_MT_RCP(det);
                const V3 tvec = ray.origin - p0;
                const float u = dot(tvec, pvec) * inv_det;
                const V3 qvec = cross(tvec, e1);
                const float v = dot(ray.dir, qvec) * inv_det;
                const float t = dot(e2, qvec) * inv_det;
                if (!((u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (t <= 0.0f) | (t >= best->t))) {
                    bool found = true;
                    if (head & 2u) {
                        const GpuMaterial m = a.materials[f2b(t1.w)];
                        const float4 bc = sample_atlas(a, tri_uv(a, head >> 2, u, v), m.base_color, m.base_color_texture);
                        if (bc.w < 1.0f) found = false;
                    }
                    if (found) { best->t = t; best->u = u; best->v = v; best->inv_det = inv_det; best->tri = head >> 2; best->material = f2b(t1.w); found_any = true; }
                }
            }
            if (head & 1u) { cur += 2u; continue; }
        }
        if (sp > 0) { sp--; cur = (uint32_t)stack[sp * 64]; } else break;
    }
    return found_any;
}

// ---- the WIDE stream (round 5; k_bvh.hip k_bvh_wide, StTuning::wide_bvh): the same rays over 4-wide nodes. Round 4's probes priced ONE more 64-B
// line per step at +44 % and tripled box arithmetic at +20 %: the loop is bound by the lines it fetches and by its dependent round trips. The
// compact binary entry spends 32 B (a quarter of them straddling two lines) on TWO child boxes; a wide node holds FOUR conservative f16 child
// boxes (4 x 3 axis words, exactly what compact_slab reads) + four links in ONE aligned 64-B line. Host model (tools/bvh4_sim.py, dungeon):
// 11.6 node steps per GI ray instead of 23.4, the same number of texels, half the lines, 0.55 x the loop iterations per wave, VALU unchanged.
// Round 3's 4-wide nodes lost with 128-B f32 nodes (two lines, seven texels per step) and ~25 instructions of ordering; here ordering is
//   key = (entry distance's high 16 bits | the child's 16-bit link), one v_perm_b32 per child,
// sorted by a 5-comparator network of v_min_u32 / v_max_u32: the link travels inside the key, a push is a 16-bit LDS store of the key itself,
// and children whose distances agree to 7 mantissa bits are visited in link order (order only, never the result).
//   node (64 B, texel 4 n of bvh_w):  texel 0: c0.x c0.y c0.z c1.x   texel 1: c1.y c1.z c2.x c2.y   texel 2: c2.z c3.x c3.y c3.z   (word = lower | upper << 16, f16)
//                                     texel 3: links — 16-bit form (fewer than 32768 nodes and leaf records): l0 | l1 << 16, l2 | l3 << 16, 0, 0
//                                                      32-bit form: l0, l1, l2, l3 (the key then keeps the link in its low 17 ... 24 bits — as many as the tree's
//                                                      largest link needs — and the distance's leading bits above them: one v_and_or_b32 per child).
//                                                      link = index << 1 | is a leaf record; an empty slot has an inverted box
//   leaf record (48 B, bvh_w_leaf_off + 48 k bytes into the same allocation): the compact stream's leaf entry; a run's records are consecutive
// One child's key: its slab test (compact_slab's arithmetic) fused with the cut-off — a miss or a child beyond `lim` is 0xffffffff, which sorts last.
template <class SE> struct WideKeys;
template <> struct WideKeys<uint16_t> {   // links ride in the keys
    static ST_D uint32_t key(float tmin, bool hit, float4 t3, int slot, uint32_t) {
        const uint32_t links = slot < 2 ? f2b(t3.x) : f2b(t3.y);
        return hit ? __builtin_amdgcn_perm(f2b(tmin), links, (slot & 1) ? 0x07060302u : 0x07060100u) : 0xffffffffu;
    }
    static ST_D uint32_t link(uint32_t k, uint32_t) { return k & 0xffffu; }
};
template <> struct WideKeys<uint32_t> {   // the link rides in the key's low KArgs::bvh_w_link_bits bits (17 ... 24), the distance keeps what is left above them
    static ST_D uint32_t key(float tmin, bool hit, float4 t3, int slot, uint32_t mask) {
        const uint32_t link = f2b(slot == 0 ? t3.x : (slot == 1 ? t3.y : (slot == 2 ? t3.z : t3.w)));
        return hit ? ((f2b(tmin) & ~mask) | link) : 0xffffffffu;   // v_and_or_b32
    }
    static ST_D uint32_t link(uint32_t k, uint32_t mask) { return k & mask; }
};
template <class SE>
ST_D uint32_t wide_key(uint32_t wx, uint32_t wy, uint32_t wz, const RaySlabs& r, float lim, float4 t3, int slot, uint32_t mask) {
    wx = __builtin_amdgcn_alignbit(wx, wx, r.rx); wy = __builtin_amdgcn_alignbit(wy, wy, r.ry); wz = __builtin_amdgcn_alignbit(wz, wz, r.rz);
    const float nx = fmaf(half_lo(wx), r.inv.x, r.oi.x), fx = fmaf(half_hi(wx), r.inv.x, r.oi.x);
    const float ny = fmaf(half_lo(wy), r.inv.y, r.oi.y), fy = fmaf(half_hi(wy), r.inv.y, r.oi.y);
    const float nz = fmaf(half_lo(wz), r.inv.z, r.oi.z), fz = fmaf(half_hi(wz), r.inv.z, r.oi.z);
    const float tmin = fmax_(fmax_(fmax_(nx, ny), nz), 0.0f);
    const float tmax = fmin_(fmin_(fx, fy), fz);
    return WideKeys<SE>::key(tmin, (tmin <= tmax) & (tmin < lim), t3, slot, mask);
}
// four keys in ascending order: sort three with v_min3 / v_med3 / v_max3, insert the fourth with two more v_med3 — 7 instructions
// (the compiler finds v_med3_u32 in min / max trees only sometimes and v_min3_u32 never: stated here)
ST_D uint32_t umin3_(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
ST_D uint32_t umax3_(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_max3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
ST_D uint32_t umed3_(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
#define ST_WIDE_SORT4(k0, k1, k2, k3)                                                                                   \
    do {                                                                                                                \
        const uint32_t lo_ = umin3_(k0, k1, k2), md_ = umed3_(k0, k1, k2), hi_ = umax3_(k0, k1, k2), d_ = k3;           \
        k0 = min(lo_, d_); k1 = umed3_(lo_, md_, d_); k2 = umed3_(md_, hi_, d_); k3 = max(hi_, d_);                     \
    } while (0)
// (ONE fetch site for both kinds of step, as in the compact loop: with a fetch in each body a wave whose lanes sit on nodes AND on leaf
// records pays two dependent round trips per iteration — measured: the incoherent GI rays lost 10 % that way while coherent rays gained.
// Nodes and leaf records therefore live in one allocation, the records `bvh_w_leaf_off` bytes behind its start.)
// A push the stack has no room for is DROPPED (the subtree behind it is never visited: geometry can be missed) — and reported: the keys are sorted, so
// whenever a node pushes at all its LAST push is k1's, and if any of its pushes found the stack full that one did too. One `else` per node step, never
// taken on any scene measured (deepest stack 11-14 of 24), sets a sticky word of page-locked host memory the engine owns (KArgs::walk_flags:
// a plain store of 1, no atomic — word 0: a per-lane walk, word 1: the primary rays' packet); the next st_tick that sees it re-arms the launches with a
// deeper stack (the packet: hands primary visibility back to the per-lane walk) and returns ST_ERR_BVH_TOO_DEEP once (st_tick.cpp).
constexpr uint32_t kWalkOverflowLane = 0u, kWalkOverflowPacket = 1u;
ST_D void wide_walk_overflowed(const KArgs& a, uint32_t word) { if (a.walk_flags) a.walk_flags[word] = 1u; }
ST_D uint32_t wide_at(const KArgs& a, uint32_t cur) { return (cur & 1u) ? a.bvh_w_leaf_off + __umul24(cur >> 1, 48u) : (cur << 5); }
template <class SE>
ST_D bool any_hit_wide(const KArgs& a, const Ray& ray, SE* stack) {
    if (a.bvh_len == 0u) return false;
    const float limit = ray.len;
    const RaySlabs rs = ray_slabs(ray);
    uint32_t cur = a.bvh_w_root;
    SE* top = stack;                                             // the stack pointer is the LDS address itself: a push / pop is one add, no index -> address step
    const SE* const stack_end = stack + a.stack_entries * 64u;
    bool hit = false;
    for (;;) {
        const bool leaf = (cur & 1u) != 0u;
        const float4* e = bvh_entry(a.bvh_w, wide_at(a, cur));
        const float4 t0 = e[0], t1 = e[1], t2 = e[2];
        float4 t3 = f4z();
        if (!leaf) t3 = e[3];
        asm volatile("" :: "v"(t0.x), "v"(t1.x), "v"(t2.x), "v"(t3.x));   // one round trip for the line
        if (!leaf) {
            uint32_t k0 = wide_key<SE>(f2b(t0.x), f2b(t0.y), f2b(t0.z), rs, limit, t3, 0, a.bvh_w_link_mask);
            uint32_t k1 = wide_key<SE>(f2b(t0.w), f2b(t1.x), f2b(t1.y), rs, limit, t3, 1, a.bvh_w_link_mask);
            uint32_t k2 = wide_key<SE>(f2b(t1.z), f2b(t1.w), f2b(t2.x), rs, limit, t3, 2, a.bvh_w_link_mask);
            uint32_t k3 = wide_key<SE>(f2b(t2.y), f2b(t2.z), f2b(t2.w), rs, limit, t3, 3, a.bvh_w_link_mask);
            ST_WIDE_SORT4(k0, k1, k2, k3);
            if (k3 != 0xffffffffu) { if (top < stack_end) { *top = (SE)WideKeys<SE>::link(k3, a.bvh_w_link_mask); top += 64; } }
            if (k2 != 0xffffffffu) { if (top < stack_end) { *top = (SE)WideKeys<SE>::link(k2, a.bvh_w_link_mask); top += 64; } }
            if (k1 != 0xffffffffu) { if (top < stack_end) { *top = (SE)WideKeys<SE>::link(k1, a.bvh_w_link_mask); top += 64; } else wide_walk_overflowed(a, kWalkOverflowLane); }
            if (k0 != 0xffffffffu) { cur = WideKeys<SE>::link(k0, a.bvh_w_link_mask); continue; }
        } else {
            const uint32_t head = f2b(t0.w);
            float u, v;
            bool found = any_triangle(ray, xyz(t0), xyz(t1), xyz(t2), limit, &u, &v);
            if (found && (head & 2u)) {  // AlphaMode::Blend: the texel decides (exact-island fetch, as in traverse())
                const GpuMaterial m = a.materials[f2b(t1.w)];
                const float4 bc = sample_atlas(a, tri_uv(a, head >> 2, u, v), m.base_color, m.base_color_texture);
                if (bc.w < 1.0f) found = false;
            }
            if (found) { hit = true; break; }
            if (head & 1u) { cur += 2u; continue; }   // the next record of the run
        }
        if (top > stack) { top -= 64; cur = (uint32_t)*top; } else break;
    }
    return hit;
}
// EXACT_LEAF (primary rays, round 6): leaf records are tested with the exact island's Triangle::hit — (t, u, v) bit-identical to the contract walk's for the same triangle
template <class SE, bool EXACT_LEAF>
ST_D bool closest_hit_wide(const KArgs& a, const Ray& ray, SE* stack, Candidate* best) {
    best->t = kF32Max; best->tri = 0xffffffffu; best->material = 0u; best->u = 0.0f; best->v = 0.0f; best->inv_det = 1.0f;
    if (a.bvh_len == 0u) return false;
    const RaySlabs rs = ray_slabs(ray);
    uint32_t cur = a.bvh_w_root;
    SE* top = stack;                                             // the stack pointer is the LDS address itself: a push / pop is one add, no index -> address step
    const SE* const stack_end = stack + a.stack_entries * 64u;
    bool found_any = false;
    for (;;) {
        const bool leaf = (cur & 1u) != 0u;
        const float4* e = bvh_entry(a.bvh_w, wide_at(a, cur));
        const float4 t0 = e[0], t1 = e[1], t2 = e[2];
        float4 t3 = f4z();
        if (!leaf) t3 = e[3];
        asm volatile("" :: "v"(t0.x), "v"(t1.x), "v"(t2.x), "v"(t3.x));
        if (!leaf) {
            const float lim = best->t;
            uint32_t k0 = wide_key<SE>(f2b(t0.x), f2b(t0.y), f2b(t0.z), rs, lim, t3, 0, a.bvh_w_link_mask);
            uint32_t k1 = wide_key<SE>(f2b(t0.w), f2b(t1.x), f2b(t1.y), rs, lim, t3, 1, a.bvh_w_link_mask);
            uint32_t k2 = wide_key<SE>(f2b(t1.z), f2b(t1.w), f2b(t2.x), rs, lim, t3, 2, a.bvh_w_link_mask);
            uint32_t k3 = wide_key<SE>(f2b(t2.y), f2b(t2.z), f2b(t2.w), rs, lim, t3, 3, a.bvh_w_link_mask);
            ST_WIDE_SORT4(k0, k1, k2, k3);
            if (k3 != 0xffffffffu) { if (top < stack_end) { *top = (SE)WideKeys<SE>::link(k3, a.bvh_w_link_mask); top += 64; } }
            if (k2 != 0xffffffffu) { if (top < stack_end) { *top = (SE)WideKeys<SE>::link(k2, a.bvh_w_link_mask); top += 64; } }
            if (k1 != 0xffffffffu) { if (top < stack_end) { *top = (SE)WideKeys<SE>::link(k1, a.bvh_w_link_mask); top += 64; } else wide_walk_overflowed(a, kWalkOverflowLane); }
            if (k0 != 0xffffffffu) { cur = WideKeys<SE>::link(k0, a.bvh_w_link_mask); continue; }
        } else {
            const uint32_t head = f2b(t0.w);
            const V3 p0 = xyz(t0), e1 = xyz(t1), e2 = xyz(t2);
            float t, u, v, inv_det;
            bool found;
            if (EXACT_LEAF) found = triangle_hit_exact(ray, p0, e1, e2, best->t, &t, &u, &v, &inv_det);
            else {
                const V3 pvec = cross(ray.dir, e2);
                const float det = dot(e1, pvec);
                inv_det = ST_MT_RCP(det);
                const V3 tvec = ray.origin - p0;
                u = dot(tvec, pvec) * inv_det;
                const V3 qvec = cross(tvec, e1);
                v = dot(ray.dir, qvec) * inv_det;
                t = dot(e2, qvec) * inv_det;
                found = !(fabsf(det) < kF32Eps) & !((u < 0.0f) | (u > 1.0f) | (v < 0.0f) | (u + v > 1.0f) | (t <= 0.0f) | (t >= best->t));
            }
            if (found) {
                if (head & 2u) {
                    const GpuMaterial m = a.materials[f2b(t1.w)];
                    const float4 bc = sample_atlas(a, tri_uv(a, head >> 2, u, v), m.base_color, m.base_color_texture);
                    if (bc.w < 1.0f) found = false;
                }
                if (found) { best->t = t; best->u = u; best->v = v; best->inv_det = inv_det; best->tri = head >> 2; best->material = f2b(t1.w); found_any = true; }
            }
            if (head & 1u) { cur += 2u; continue; }
        }
        if (top > stack) { top -= 64; cur = (uint32_t)*top; } else break;
    }
    return found_any;
}

// ---- A WAVE-WIDE PACKET over the wide stream, for coherent rays (round 5: primary visibility; StTuning::primary_packets). The 64 primary rays of
// an 8 x 8 tile walk nearly the same nodes (host model, dungeon: 13.9 node steps per ray, 15.0 for the tile's longest ray, 15.7 in the UNION of the
// tile's paths), yet in the per-lane loop every lane fetches its own node, sorts its own keys and keeps its own stack: ~89 VALU instructions and
// four vector loads per node step. Here the WAVE walks the union: `cur` and the stack are uniform, a node is fetched ONCE with scalar loads
// (64 B through the scalar cache: no vector memory instruction in the loop), every lane tests the node's four boxes against its own ray — the
// box words arrive as scalar operands —, v_cmp's result IS the ballot, a child is entered when any lane hits it, children are ordered by the
// distances of the first lane that hits each, and the stack is ONE VGPR indexed by lane (v_writelane / v_readlane with a uniform stack pointer:
// 64 entries). A lane that missed a node misses its children too (their boxes lie inside it), so no per-entry lane mask is kept. Leaf records
// are scalar loads as well; the triangle test is each lane's own. Lanes that have left the kernel are simply inactive: ballots skip them.
// Results: each lane's closest hit — the same triangle as the per-lane walk finds, except where two triangles tie.
typedef const __attribute__((address_space(4))) uint32_t* ScalarWords;
ST_D uint32_t wave_uniform(uint32_t x) { return __builtin_amdgcn_readfirstlane(x); }
// lane `lane` of `reg` = `value` (both uniform); v_writelane_b32 ignores EXEC. (This clang has the readlane builtin but no writelane one.)
ST_D uint32_t wave_writelane(uint32_t reg, uint32_t value, uint32_t lane) {
    // (two different SGPR operands would break the one-SGPR constant-bus rule of gfx9 VALU instructions: the lane select goes through M0)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(reg) : "s"(wave_uniform(value)), "s"(wave_uniform(lane)) : "m0");
    return reg;
}
// (The same walk over the LDS-resident CONTRACT stream of the Cornell box — uniform pointer, LDS broadcast reads, the exact island's box and
// triangle tests per lane — was built too: prim_visibility 59.3 -> 58.0 us, inside the noise of the frame, and the whole-frame steady-state test
// no longer passed (ties between a quad's two triangles resolve in packet order, not in the per-lane walk's). Not kept: the Cornell box keeps its
// per-lane contract walk.)
// (A software-pipelined form — the nearest child's line fetched before the others are pushed, a leaf step's successor before its triangle is
// tested, keys sorted by a branch-free min / max network with the slot in their low bits — measured SLOWER on the same box: prim_visibility
// 119 -> 132 us on the dungeon, 150 -> 173 at 208 k triangles, 440 -> 485 at 3840 x 2160: sixteen more live SGPRs for the second line spill, and
// scalar loads return out of order, so every wait for an OLD line also waits for the prefetched one. The simple loop below is the one kept.)
ST_D bool closest_hit_packet(const KArgs& a, const Ray& ray, Candidate* best) {
    best->t = kF32Max; best->tri = 0xffffffffu; best->material = 0u; best->u = 0.0f; best->v = 0.0f; best->inv_det = 1.0f;
    if (a.bvh_len == 0u) return false;
    const RaySlabs rs = ray_slabs(ray);
    const ScalarWords base = (ScalarWords)(a.bvh_w);
    const uint32_t leaf_words = a.bvh_w_leaf_off >> 2;
    uint32_t cur = a.bvh_w_root;          // uniform
    uint32_t stack = 0u;                  // lane k of this VGPR = stack entry k
    uint32_t sp = 0u;                     // uniform
    bool found_any = false;
    for (;;) {
        if (!(cur & 1u)) {
            const ScalarWords n = base + (size_t)cur * 8u;   // node index = cur >> 1, 16 words each
            const uint32_t w0 = n[0], w1 = n[1], w2 = n[2], w3 = n[3], w4 = n[4], w5 = n[5], w6 = n[6], w7 = n[7], w8 = n[8], w9 = n[9], w10 = n[10], w11 = n[11];
            const uint32_t x0 = n[12], x1 = n[13], x2 = n[14], x3 = n[15];
            uint32_t l0, l1, l2, l3;
            if (a.bvh_w_links16) { l0 = x0 & 0xffffu; l1 = x0 >> 16; l2 = x1 & 0xffffu; l3 = x1 >> 16; } else { l0 = x0; l1 = x1; l2 = x2; l3 = x3; }
            const float lim = best->t;
            const float t0 = compact_slab(w0, w1, w2, rs), t1 = compact_slab(w3, w4, w5, rs), t2 = compact_slab(w6, w7, w8, rs), t3 = compact_slab(w9, w10, w11, rs);
            const unsigned long long m0 = __builtin_amdgcn_ballot_w64(t0 < lim), m1 = __builtin_amdgcn_ballot_w64(t1 < lim), m2 = __builtin_amdgcn_ballot_w64(t2 < lim),
                                     m3 = __builtin_amdgcn_ballot_w64(t3 < lim);
            // keys: the entry distance of the first lane that hits the child (bits of a non-negative float order as integers), ~0 when no lane does
            uint32_t k0 = m0 ? (uint32_t)__builtin_amdgcn_readlane((int)f2b(t0), (int)__builtin_ctzll(m0)) : 0xffffffffu;
            uint32_t k1 = m1 ? (uint32_t)__builtin_amdgcn_readlane((int)f2b(t1), (int)__builtin_ctzll(m1)) : 0xffffffffu;
            uint32_t k2 = m2 ? (uint32_t)__builtin_amdgcn_readlane((int)f2b(t2), (int)__builtin_ctzll(m2)) : 0xffffffffu;
            uint32_t k3 = m3 ? (uint32_t)__builtin_amdgcn_readlane((int)f2b(t3), (int)__builtin_ctzll(m3)) : 0xffffffffu;
            // a 5-comparator network on (key, link) pairs: uniform values, scalar unit
#define ST_PKT_CSWAP(ka, la, kb, lb) do { if (kb < ka) { const uint32_t tk_ = ka; ka = kb; kb = tk_; const uint32_t tl_ = la; la = lb; lb = tl_; } } while (0)
            ST_PKT_CSWAP(k0, l0, k1, l1); ST_PKT_CSWAP(k2, l2, k3, l3); ST_PKT_CSWAP(k0, l0, k2, l2); ST_PKT_CSWAP(k1, l1, k3, l3); ST_PKT_CSWAP(k1, l1, k2, l2);
#undef ST_PKT_CSWAP
            if (k3 != 0xffffffffu && sp < 64u) { stack = wave_writelane(stack, l3, sp); sp++; }
            if (k2 != 0xffffffffu && sp < 64u) { stack = wave_writelane(stack, l2, sp); sp++; }
            if (k1 != 0xffffffffu) { if (sp < 64u) { stack = wave_writelane(stack, l1, sp); sp++; } else wide_walk_overflowed(a, kWalkOverflowPacket); }
            if (k0 != 0xffffffffu) { cur = l0; continue; }
        } else {
            const ScalarWords r = base + leaf_words + (size_t)(cur >> 1) * 12u;
            const V3 p0 = v3(b2f(r[0]), b2f(r[1]), b2f(r[2])), e1 = v3(b2f(r[4]), b2f(r[5]), b2f(r[6])), e2 = v3(b2f(r[8]), b2f(r[9]), b2f(r[10]));
            const uint32_t head = r[3], material = r[7];
            // (the island's Triangle::hit: primary hits carry the CPU restatement's (t, u, v) bit for bit — see closest_resolve_exact; 12 more VALU instructions per
            // record than the contracted form with v_rcp_f32, three to five records per ray)
            float t, u, v, inv_det;
            if (triangle_hit_exact(ray, p0, e1, e2, best->t, &t, &u, &v, &inv_det)) {
                bool found = true;
                if (head & 2u) {
                    const GpuMaterial m = a.materials[material];
                    const float4 bc = sample_atlas(a, tri_uv(a, head >> 2, u, v), m.base_color, m.base_color_texture);
                    if (bc.w < 1.0f) found = false;
                }
                if (found) { best->t = t; best->u = u; best->v = v; best->inv_det = inv_det; best->tri = head >> 2; best->material = material; found_any = true; }
            }
            if (head & 1u) { cur += 2u; continue; }
        }
        if (sp == 0u) break;
        sp--; cur = (uint32_t)__builtin_amdgcn_readlane((int)stack, (int)sp);
    }
    return found_any;
}
#endif
// Ray::intersect (shadow ray)
template <class SE>
ST_D bool trace_any(const KArgs& a, const Ray& ray, SE* stack, uint32_t* used_memory) {
#if ST_FAST_DEVICE && !defined(ST_NO_ANYHIT_FAST)
    if (!a.anyhit_contract) {
        *used_memory = 0u;
        if (a.bvh_w != nullptr) return any_hit_wide(a, ray, stack);
        return a.bvh_c != nullptr ? any_hit_compact(a, ray, stack) : any_hit_fast(a, ray, stack);
    }
#endif
    return trace_any_contract(a, ray, stack, used_memory);
}
ST_D void hit_pack(const TriangleHit& h, float4* d0, float4* d1) {  // hit.rs:112-120
    *d0 = f4(h.point, b2f(h.material_id));
    const V2 n = normal_encode(h.normal);
    *d1 = make_float4(n.x, n.y, h.uv.x, h.uv.y);
}
ST_D TriangleHit hit_unpack(float4 d0, float4 d1) {  // hit.rs:95-110
    TriangleHit h;
    h.xform_slot = 0u;
    if (is_zero(xyz(d0))) { h.distance = kF32Max; h.point = v3s(0.0f); h.normal = v3s(0.0f); h.uv = v2(0.0f, 0.0f); h.material_id = 0u; return h; }
    h.distance = 0.0f; h.point = xyz(d0); h.normal = normal_decode(v2(d1.x, d1.y)); h.uv = v2(d1.z, d1.w); h.material_id = f2b(d0.w);
    return h;
}

// ------------------------------------------------------------------ G-buffer, surface, hit (gbuffer.rs, surface.rs, hit.rs)
struct GBuffer { float4 base_color; V3 normal; float metallic; V3 emissive; float roughness, reflectance, depth; };
ST_D GBuffer gbuffer_zero() { GBuffer g; g.base_color = f4z(); g.normal = v3s(0.0f); g.metallic = 0.0f; g.emissive = v3s(0.0f); g.roughness = 0.0f; g.reflectance = 0.0f; g.depth = 0.0f; return g; }
ST_D GBuffer gbuffer_unpack(const KArgs& a, float4 d0, float4 d1) {
    GBuffer g;
    g.depth = d0.x;
    g.normal = normal_decode(v2(d0.y, d0.z));
    const uint32_t w0 = f2b(d0.w);
    g.metallic = unorm8(a, w0 & 0xffu);
    g.roughness = sqr(unorm8(a, (w0 >> 8) & 0xffu));
    g.reflectance = unorm8(a, (w0 >> 16) & 0xffu);
    g.emissive = xyz(d1);
    const uint32_t w1 = f2b(d1.w);
    const float* lut = a.byte_luts;
    g.base_color = make_float4(lut[kLutGamma8 + (w1 & 0xffu)], lut[kLutGamma8 + ((w1 >> 8) & 0xffu)], lut[kLutGamma8 + ((w1 >> 16) & 0xffu)],
                               lut[kLutGamma6 + ((w1 >> 24) & 0xffu)]);
    return g;
}
// gbuffer.rs:19-50. The base colour's four gamma-encoded bytes (4 pow_ = ~320 VALU operations) depend only on the colour;
// for a material without a base-colour texture that is a per-material constant, which the host packs once with this very
// routine (gbuffer_pack_base_color, st_math.h; KArgs::material_base_packed) and primary visibility / GI sampling pass in
// through `base_bits`.
ST_D void gbuffer_pack_bits(const GBuffer& g, uint32_t base_bits, float4* d0, float4* d1) {
    const V2 n = normal_encode(g.normal);
    const float m = clampf(g.metallic, 0.0f, 1.0f) * 255.0f;
    const float r = clampf(fsqrt(g.roughness), 0.0f, 1.0f) * 255.0f;
    const float rf = clampf(g.reflectance, 0.0f, 1.0f) * 255.0f;
    *d0 = make_float4(g.depth, n.x, n.y, b2f(u32_from_bytes(f2u_sat(m), f2u_sat(r), f2u_sat(rf), 1u)));
    *d1 = make_float4(g.emissive.x, g.emissive.y, g.emissive.z, b2f(base_bits));
}
ST_D void gbuffer_pack(const GBuffer& g, float4* d0, float4* d1) { gbuffer_pack_bits(g, gbuffer_pack_base_color(g.base_color), d0, d1); }
ST_D float clamped_roughness(const GBuffer& g) { return clampf(g.roughness, 0.089f * 0.089f, 1.0f); }

struct Surface { V3 normal; float depth, roughness; };
ST_D Surface surface_from(float4 d) { Surface s; s.normal = normal_decode(v2(d.x, d.y)); s.depth = d.z; s.roughness = d.w; return s; }
ST_D Surface surface_decoded(float4 d) { Surface s; s.normal = v3(d.x, d.y, d.z); s.depth = d.w; s.roughness = 0.0f; return s; }  // from KArgs::sn
ST_D float surface_similarity(const Surface& self, const Surface& other) {  // surface.rs:21-47
    if (self.depth == 0.0f || other.depth == 0.0f) return 0.0f;
    const float d = fmax_(dot(self.normal, other.normal), 0.0f);
    const float normal_score = d <= 0.5f ? 0.0f : 2.0f * d;
    const float t = fabsf(self.depth - other.depth);
    const float depth_score = t >= 0.1f * other.depth ? 0.0f : 1.0f;
    return normal_score * depth_score;
}

struct Hit { V3 origin, dir, point; GBuffer g; };
constexpr float kNudgeOffset = 0.01f;  // hit.rs:19
ST_D Hit hit_make(const Ray& ray, const GBuffer& g) { Hit h; h.origin = ray.origin; h.dir = ray.dir; h.point = ray_at(ray, g.depth - kNudgeOffset); h.g = g; return h; }
ST_D Hit hit_zero() { Hit h; h.origin = v3s(0.0f); h.dir = v3s(0.0f); h.point = v3s(0.0f); h.g = gbuffer_zero(); return h; }
ST_D bool hit_some(const Hit& h) { return h.g.depth != 0.0f; }
ST_D Hit pixel_hit(const KArgs& a, const GpuCamera& cam, const float4* g0, const float4* g1, U2 pos) {
    return hit_make(camera_ray_shading(cam, pos), gbuffer_unpack(a, tex_read(g0, a, pos), tex_read(g1, a, pos)));
}

// ------------------------------------------------------------------ BRDFs (brdf.rs)
ST_D float ggx_distribution(float n_dot_h, float roughness) {
    const float a2 = roughness * roughness;
    const float d = (n_dot_h * a2 - n_dot_h) * n_dot_h + 1.0f;
    return fdiv(a2, kPi * d * d);
}
ST_D float ggx_schlick_masking_term(float n_dot_l, float n_dot_v, float roughness) {
    const float k = roughness * roughness / 2.0f;
    const float g_v = fdiv(n_dot_v, n_dot_v * (1.0f - k) + k);
    const float g_l = fdiv(n_dot_l, n_dot_l * (1.0f - k) + k);
    return g_v * g_l;
}
ST_D V3 ggx_schlick_fresnel(V3 f0, float l_dot_h) {
    const float f90 = saturate(dot(f0, v3s(50.0f * 0.33f)));
    return f0 + (v3s(f90) - f0) * pow5_(fmax_(1.0f - l_dot_h, 0.001f));
}
ST_D V3 diffuse_eval(const GBuffer& g) { return divc3(xyz(g.base_color) * (1.0f - g.metallic), kPi); }
ST_D V3 specular_eval(const GBuffer& g, V3 l, V3 v) {
    if (g.metallic <= 0.0f) return v3s(0.0f);
    const float a = clamped_roughness(g);
    const V3 n = g.normal;
    const V3 h = normalize(l + v);
    const float n_dot_l = saturate(dot(n, l)), n_dot_h = saturate(dot(n, h)), l_dot_h = saturate(dot(l, h)), n_dot_v = saturate(dot(n, v));
    if (n_dot_l <= 0.0f || n_dot_v <= 0.0f) return v3s(0.0f);
    const float d = ggx_distribution(n_dot_h, a);
    const float gg = ggx_schlick_masking_term(n_dot_l, n_dot_v, a);
    const V3 f0 = v3s(0.16f * g.reflectance * g.reflectance * (1.0f - g.metallic)) + xyz(g.base_color) * g.metallic;
    const V3 f = ggx_schlick_fresnel(f0, l_dot_h);
    return d * gg * f / (4.0f * n_dot_l * n_dot_v);
}
struct BrdfSample { V3 dir; float pdf; V3 radiance; };
ST_D BrdfSample layered_brdf_sample(const GBuffer& g, WhiteNoise& wn, V3 v) {
    BrdfSample s;
    if (wn.sample() < g.metallic) {
        const float r0 = wn.sample(), r1 = wn.sample();
        const float a = clamped_roughness(g);
        const V3 n = g.normal;
        const float a2 = sqr(a);
        V3 b, t;
        any_orthonormal_pair(n, &b, &t);  // brdf.rs:91 `(b, t)`
        const float cos_theta = fsqrt(fmax_(0.0f, fdiv(1.0f - r0, (a2 - 1.0f) * r0 + 1.0f)));
        const float sin_theta = fsqrt(fmax_(0.0f, 1.0f - cos_theta * cos_theta));
        const float phi = r1 * kPi * 2.0f;
        float sp_, cp_; sincos_(phi, &sp_, &cp_);
        const V3 h = t * (sin_theta * cp_) + b * (sin_theta * sp_) + n * cos_theta;
        const float n_dot_h = saturate(dot(n, h)), h_dot_v = saturate(dot(h, v));
        s.dir = normalize(2.0f * h_dot_v * h - v);
        s.pdf = fdiv(ggx_distribution(n_dot_h, a) * n_dot_h, 4.0f * h_dot_v);
        s.radiance = specular_eval(g, s.dir, v);
        s.pdf = fdiv(s.pdf, g.metallic);
    } else {
        s.dir = wn.sample_hemisphere(g.normal);
        s.pdf = 1.0f / kPi;
        s.radiance = diffuse_eval(g);
        s.pdf = fdiv(s.pdf, 1.0f - g.metallic);
    }
    return s;
}

// ------------------------------------------------------------------ lights (light.rs)
struct LightRadiance { V3 radiance, diff_brdf, spec_brdf; };
ST_D V3 radiance_sum(const LightRadiance& r) { return r.radiance * (r.diff_brdf + r.spec_brdf); }
ST_D GpuLight light_zero() { GpuLight l; l.d0 = l.d1 = l.d2 = l.d3 = l.prev_d0 = l.prev_d1 = l.prev_d2 = f4z(); return l; }
ST_D GpuLight light_get(const KArgs& a, uint32_t id) {  // unchecked in the reference
    if (id >= a.n_lights_buf) return light_zero();
    if (a.lights_lds != nullptr && id < kLdsLights) return a.lights_lds[id];   // tracing kernels: the head of the table lives in LDS
    return a.lights[id];
}
ST_D GpuLight light_get_prev(const KArgs& a, uint32_t id) { GpuLight l = light_get(a, id); l.d0 = l.prev_d0; l.d1 = l.prev_d1; l.d2 = l.prev_d2; return l; }
ST_D bool light_is_none(const GpuLight& l) { return f2b(l.d2.x) == 0u; }
ST_D bool light_contains(const GpuLight& l, V3 p) { return distance(xyz(l.d0), p) <= l.d0.w; }

ST_D LightRadiance light_radiance(const GpuLight& l, const Hit& hit) {  // light.rs:143-207
    const V3 center = xyz(l.d0);
    const float radius = l.d0.w, range = l.d1.w;
    const V3 lv = center - hit.point;
    float f_angle;
    if (f2b(l.d2.x) == 1u) f_angle = 1.0f;
    else {
        const float angle = angle_between(normal_decode(v2(l.d2.y, l.d2.z)), hit.point - center);
        f_angle = saturate(1.0f - pow3_(fdiv(angle, l.d2.w)));
    }
    float f_dist;
    if (range == INFINITY) f_dist = 1.0f;
    else {
        const float l2 = length_squared(lv);
        const float inv_r2 = frcp(sqr(range));
        const float factor = l2 * inv_r2;
        const float smooth_factor = saturate(1.0f - factor * factor);
        const float attenuation = smooth_factor * smooth_factor;
        f_dist = fdiv(attenuation, fmax_(l2, 0.0001f));
    }
    const float f_cosine = saturate(dot(hit.g.normal, normalize(lv)));
    LightRadiance out;
    out.diff_brdf = diffuse_eval(hit.g);
    {
        const V3 v = -hit.dir;
        const V3 n = hit.g.normal;
        const V3 r = reflect(-v, n);
        const V3 center_to_ray = dot(lv, r) * r - lv;
        const float t = radius * inverse_sqrt(dot(center_to_ray, center_to_ray));
        const V3 closest_point = lv + center_to_ray * saturate(t);
        const float l_spec_length_inverse = inverse_sqrt(dot(closest_point, closest_point));
        const float tt = clamped_roughness(hit.g) + radius * 0.5f * l_spec_length_inverse;
        const float i_roughness = fdiv(clamped_roughness(hit.g), saturate(tt));
        const float intensity = sqr(i_roughness);
        const V3 ls = closest_point * l_spec_length_inverse;
        out.spec_brdf = intensity * specular_eval(hit.g, ls, v);
    }
    out.radiance = xyz(l.d1) * f_angle * f_dist * f_cosine;
    return out;
}
ST_D Ray light_ray_wnoise(const GpuLight& l, WhiteNoise& wn, V3 hit_point) {  // light.rs:209-215
    const V3 light_pos = xyz(l.d0) + l.d0.w * wn.sample_sphere();
    const V3 light_to_hit = hit_point - light_pos;
    Ray r = make_ray(light_pos, normalize(light_to_hit));
    r.len = length(light_to_hit);
    return r;
}
ST_D Ray light_ray_bnoise(const GpuLight& l, V2 sample, V3 hit_point) {  // light.rs:217-239
    const V3 to_light = xyz(l.d0) - hit_point;
    const V3 light_dir = normalize(to_light);
    const float light_distance = length(to_light);
    const float light_radius = fdiv(l.d0.w, light_distance);
    V3 lt, lb;
    any_orthonormal_pair(light_dir, &lt, &lb);
    const float angle = 2.0f * kPi * sample.x;
    const float rad = fsqrt(sample.y);
    float s_, c_; sincos_(angle, &s_, &c_);
    const V2 disk_point = v2(s_, c_) * rad * light_radius;
    V3 ray_dir = light_dir + disk_point.x * lt + disk_point.y * lb;
    ray_dir = normalize(ray_dir);
    Ray r = make_ray(hit_point + ray_dir * light_distance, -ray_dir);
    r.len = light_distance;
    return r;
}

// ------------------------------------------------------------------ atmosphere lookup (atmosphere.rs:86-205)
constexpr float kGroundRadiusMm = 6.360f, kAtmosphereRadiusMm = 6.460f, kExposure = 20.0f;
ST_D float4 lut_texel(const float4* lut, int32_t w, int32_t h, int32_t x, int32_t y) {
    x = x < 0 ? 0 : (x >= w ? w - 1 : x);
    y = y < 0 ? 0 : (y >= h ? h - 1 : y);
    return lut[y * w + x];
}
ST_D float4 lut_sample(const float4* lut, int32_t w, int32_t h, V2 uv) {
    if (uv.x != uv.x) uv.x = 0.0f;
    if (uv.y != uv.y) uv.y = 0.0f;
    const float fx = uv.x * (float)w - 0.5f, fy = uv.y * (float)h - 0.5f;
    const float x0 = floorf(fx), y0 = floorf(fy);
    const float tx = fx - x0, ty = fy - y0;
    const int32_t ix = f2i_sat(x0), iy = f2i_sat(y0);
    const float4 p00 = lut_texel(lut, w, h, ix, iy), p10 = lut_texel(lut, w, h, ix + 1, iy), p01 = lut_texel(lut, w, h, ix, iy + 1), p11 = lut_texel(lut, w, h, ix + 1, iy + 1);
    const float4 top = p00 + (p10 - p00) * tx;
    const float4 bot = p01 + (p11 - p01) * tx;
    return top + (bot - top) * ty;
}
ST_D V3 atmosphere_sample(const KArgs& a, V3 ray_dir) {
    const V3 sun_dir = v3(a.sun_dir[0], a.sun_dir[1], a.sun_dir[2]);
    const V3 view_pos = v3(0.0f, kGroundRadiusMm + 0.0002f, 0.0f);
    // sample_sky_lut
    V3 lum;
    {
        const float height = length(view_pos);
        const V3 up = view_pos / height;
        float th = sqr(height) - sqr(kGroundRadiusMm);
        th = fdiv(fsqrt(th), height);
        const float horizon = acos_(clampf(th, -1.0f, 1.0f));
        const float altitude = horizon - acos_(dot(ray_dir, up));
        float azimuth;
        if (fabsf(altitude) > (0.5f * kPi - 0.0001f)) azimuth = 0.0f;
        else {
            const V3 right = cross(sun_dir, up);
            const V3 forward = cross(up, right);
            const V3 projected_dir = normalize(ray_dir - up * dot(ray_dir, up));
            const float sin_theta = dot(projected_dir, right);
            const float cos_theta = dot(projected_dir, forward);
            azimuth = atan2_(sin_theta, cos_theta) + kPi;
        }
        const float u = fdivc(azimuth, 2.0f * kPi);
        const float v = 0.5f + 0.5f * copysignf(fsqrt(fdivc(fabsf(altitude) * 2.0f, kPi)), altitude);
        lum = xyz(lut_sample(a.sky_lut, 256, 256, v2(u, v)));
    }
    // evaluate_bloom + interpolate_bloom
    V3 sun_lum;
    {
        const float sun_solid_angle = 0.53f * kPi / 180.0f;
        const float min_sun_cos_theta = cos_(sun_solid_angle);
        const float cos_theta = dot(ray_dir, sun_dir);
        if (cos_theta >= min_sun_cos_theta) sun_lum = v3s(1.0f);
        else {
            const float offset = min_sun_cos_theta - cos_theta;
            const float gaussian_bloom = exp_(-offset * 50000.0f) * 0.5f;
            const float inv_bloom = frcp(0.02f + offset * 300.0f) * 0.01f;
            sun_lum = v3s(gaussian_bloom + inv_bloom);
        }
        const V3 t = vclamp((sun_lum - v3s(0.002f)) / (v3s(1.0f) - v3s(0.002f)), v3s(0.0f), v3s(1.0f));
        sun_lum = t * t * (v3s(3.0f) - 2.0f * t);
    }
    if (length_squared(sun_lum) > 0.0f) {
        const Ray ray = make_ray(view_pos, ray_dir);
        if (intersect_sphere(ray, kGroundRadiusMm) >= 0.0f) sun_lum = v3s(0.0f);
        else {
            const float height = length(view_pos);
            const V3 up = view_pos / height;
            const float sun_cos_zenith_angle = dot(sun_dir, up);
            const float u = saturate(0.5f + 0.5f * sun_cos_zenith_angle);
            const float v = saturate(fdivc(height - kGroundRadiusMm, kAtmosphereRadiusMm - kGroundRadiusMm));
            sun_lum = sun_lum * xyz(lut_sample(a.transmittance_lut, 256, 64, v2(u, v)));
        }
    }
    lum = lum + sun_lum;
    lum = lum * kExposure;
    return lum;
}

// ------------------------------------------------------------------ reprojection (reprojection.rs, utils/bilinear_filter.rs)
struct Reprojection { float prev_x, prev_y, confidence; uint32_t validity; };
ST_D Reprojection reprojection_read(float4 d) { Reprojection r; r.prev_x = d.x; r.prev_y = d.y; r.confidence = d.z; r.validity = f2b(d.w); return r; }
ST_D U2 reprojection_prev_round(const Reprojection& r) { return as_u2(round2(v2(r.prev_x, r.prev_y))); }
ST_D bool reprojection_is_exact(const Reprojection& r) {
    const float fx = r.prev_x - floorf(r.prev_x), fy = r.prev_y - floorf(r.prev_y);  // glam Vec2::fract
    return (fx * fx + fy * fy) == 0.0f;
}
ST_D float4 bilinear_reproject(const KArgs& a, const Reprojection& r, const float4* plane) {
    if (reprojection_is_exact(r)) return tex_read(plane, a, reprojection_prev_round(r));
    const float fl_x = floorf(r.prev_x), fl_y = floorf(r.prev_y), ce_x = ceilf(r.prev_x), ce_y = ceilf(r.prev_y);
    const I2 p[4] = {i2(f2i_sat(fl_x), f2i_sat(fl_y)), i2(f2i_sat(ce_x), f2i_sat(fl_y)), i2(f2i_sat(fl_x), f2i_sat(ce_y)), i2(f2i_sat(ce_x), f2i_sat(ce_y))};
    float4 s[4]; float w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        s[i] = f4z(); w[i] = 0.0f;
        if ((r.validity & (1u << i)) > 0u && p[i].x >= 0 && p[i].y >= 0) { s[i] = tex_read(plane, a, u2((uint32_t)p[i].x, (uint32_t)p[i].y)); w[i] = 1.0f; }
    }
    const float ux = r.prev_x - truncf(r.prev_x), uy = r.prev_y - truncf(r.prev_y);  // f32::fract
    const float4 weights = make_float4(w[0], w[1], w[2], w[3]) * make_float4((1.0f - ux) * (1.0f - uy), ux * (1.0f - uy), (1.0f - ux) * uy, ux * uy);
    const float w_sum = (weights.x * 1.0f) + (weights.y * 1.0f) + (weights.z * 1.0f) + (weights.w * 1.0f);
    if (w_sum == 0.0f) return f4z();
    return div4(s[0] * weights.x + s[1] * weights.y + s[2] * weights.z + s[3] * weights.w, w_sum);
}
// Two planes at once (the denoiser's colour and moment history share the reprojection): the taps of both are in flight together — one
// round trip instead of two — and each plane's arithmetic is bilinear_reproject's, operation for operation.
ST_D void bilinear_reproject2(const KArgs& a, const Reprojection& r, const float4* plane_a, const float4* plane_b, float4* out_a, float4* out_b) {
    if (reprojection_is_exact(r)) { const U2 q = reprojection_prev_round(r); *out_a = tex_read(plane_a, a, q); *out_b = tex_read(plane_b, a, q); return; }
    const float fl_x = floorf(r.prev_x), fl_y = floorf(r.prev_y), ce_x = ceilf(r.prev_x), ce_y = ceilf(r.prev_y);
    const I2 p[4] = {i2(f2i_sat(fl_x), f2i_sat(fl_y)), i2(f2i_sat(ce_x), f2i_sat(fl_y)), i2(f2i_sat(fl_x), f2i_sat(ce_y)), i2(f2i_sat(ce_x), f2i_sat(ce_y))};
    float4 sa[4], sb[4]; float w[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        sa[i] = f4z(); sb[i] = f4z(); w[i] = 0.0f;
        if ((r.validity & (1u << i)) > 0u && p[i].x >= 0 && p[i].y >= 0) {
            const U2 q = u2((uint32_t)p[i].x, (uint32_t)p[i].y);
            sa[i] = tex_read(plane_a, a, q); sb[i] = tex_read(plane_b, a, q); w[i] = 1.0f;
        }
    }
    const float ux = r.prev_x - truncf(r.prev_x), uy = r.prev_y - truncf(r.prev_y);  // f32::fract
    const float4 weights = make_float4(w[0], w[1], w[2], w[3]) * make_float4((1.0f - ux) * (1.0f - uy), ux * (1.0f - uy), (1.0f - ux) * uy, ux * uy);
    const float w_sum = (weights.x * 1.0f) + (weights.y * 1.0f) + (weights.z * 1.0f) + (weights.w * 1.0f);
    if (w_sum == 0.0f) { *out_a = f4z(); *out_b = f4z(); return; }
    *out_a = div4(sa[0] * weights.x + sa[1] * weights.y + sa[2] * weights.z + sa[3] * weights.w, w_sum);
    *out_b = div4(sb[0] * weights.x + sb[1] * weights.y + sb[2] * weights.z + sb[3] * weights.w, w_sum);
}

// ------------------------------------------------------------------ reservoirs (reservoir.rs, reservoir/*.rs)
struct DiSample { float pdf, confidence; uint32_t light_id; V3 light_point; bool is_occluded; };
struct DiReservoir { DiSample s; float m, w; };
ST_D DiReservoir di_empty() { DiReservoir r; r.s.pdf = 0.0f; r.s.confidence = 0.0f; r.s.light_id = 0u; r.s.light_point = v3s(0.0f); r.s.is_occluded = false; r.m = 0.0f; r.w = 0.0f; return r; }
// reservoir/di.rs:17-59: two texels per pixel
ST_D DiReservoir di_unpack(float4 d0, float4 d1) {
    const uint32_t w = f2b(d0.w);
    DiReservoir r;
    r.s.pdf = d0.z; r.s.confidence = (float)((w >> 8) & 0xffu); r.s.light_id = f2b(d1.w); r.s.light_point = xyz(d1); r.s.is_occluded = (w & 0xffu) > 0u;
    r.m = d0.x; r.w = d0.y;
    return r;
}
ST_D void di_pack(const DiReservoir& r, float4* d0, float4* d1) {
    *d0 = make_float4(r.m, r.w, r.s.pdf, b2f(u32_from_bytes(r.s.is_occluded ? 1u : 0u, f2u_sat(r.s.confidence), 0u, 0u)));
    *d1 = f4(r.s.light_point, b2f(r.s.light_id));
}
ST_D DiReservoir di_read(const float4* buf, uint32_t id, uint32_t count) {
    if (id >= count) return di_empty();
    return di_unpack(buf[2u * id], buf[2u * id + 1u]);
}
ST_D void di_write(float4* buf, uint32_t id, const DiReservoir& r) { di_pack(r, &buf[2u * id], &buf[2u * id + 1u]); }
// what di_read() returns for a reservoir di_write() just stored (the confidence passes through a byte)
ST_D DiReservoir di_after_store(const DiReservoir& r) { float4 d0, d1; di_pack(r, &d0, &d1); return di_unpack(d0, d1); }
ST_D float di_pdf_ex(const DiSample& s, const GpuLight& l, Hit hit) {  // reservoir/di.rs:105-116
    hit.g.base_color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    if (!light_is_none(l) && light_contains(l, s.light_point)) return luma(radiance_sum(light_radiance(l, hit)));
    return 0.0f;
}
ST_D Ray di_sample_ray(const DiSample& s, V3 hit_point) {
    const V3 dir = hit_point - s.light_point;
    Ray r = make_ray(s.light_point, normalize(dir));
    r.len = length(dir);
    return r;
}

struct GiSample { float pdf; uint32_t rng; V3 radiance, v1_point, v2_point, v2_normal; };
struct GiReservoir { GiSample s; float m, w, confidence; };
ST_D GiReservoir gi_empty() { GiReservoir r; r.s.pdf = 0.0f; r.s.rng = 0u; r.s.radiance = r.s.v1_point = r.s.v2_point = r.s.v2_normal = v3s(0.0f); r.m = 0.0f; r.w = 0.0f; r.confidence = 0.0f; return r; }
ST_D GiReservoir gi_from_texels(float4 d0, float4 d1, float4 d2, float4 d3) {
    GiReservoir r;
    r.s.pdf = d2.w; r.s.rng = f2b(d3.w); r.s.radiance = xyz(d0); r.s.v1_point = xyz(d1); r.s.v2_point = xyz(d2); r.s.v2_normal = normal_decode(v2(d3.x, d3.y));
    r.m = d0.w; r.w = d1.w; r.confidence = d3.z;
    return r;
}
ST_D GiReservoir gi_read(const float4* buf, uint32_t id, uint32_t count) {
    if (id >= count) return gi_empty();
    return gi_from_texels(buf[4u * id], buf[4u * id + 1u], buf[4u * id + 2u], buf[4u * id + 3u]);
}
ST_D void gi_write(float4* buf, uint32_t id, const GiReservoir& r) {
    buf[4u * id] = f4(r.s.radiance, r.m);
    buf[4u * id + 1u] = f4(r.s.v1_point, r.w);
    buf[4u * id + 2u] = f4(r.s.v2_point, r.s.pdf);
    const V2 n = normal_encode(r.s.v2_normal);
    buf[4u * id + 3u] = make_float4(n.x, n.y, r.confidence, b2f(r.s.rng));
}
// what gi_read() returns for a reservoir gi_write() just stored: every field is kept bit for bit except the normal, which
// passes through the octahedral code
ST_D GiReservoir gi_after_store(GiReservoir r) { r.s.v2_normal = normal_decode(normal_encode(r.s.v2_normal)); return r; }
// ---- quad-transposed reservoir I/O
// A GI reservoir is a 64-B record (four float4). gi_read / gi_write move it with four instructions whose lanes are 64 B
// apart: every instruction touches a quarter of every cache line the wave covers. Measured (tools/fetch_calib.hip,
// calib_aos64_*): a 1080p plane of such records copies at 4.1 TB/s that way and at 5.5 TB/s when the four lanes of a quad
// move one whole record per instruction (64 contiguous bytes per quad) — the same bytes, 25 % less time. So where a kernel
// streams its OWN pixel's reservoir (consecutive lanes of a quad = consecutive pixels of a row = consecutive records), the
// quad loads record k with instruction k and a 4x4 transpose across the quad (two rounds of quad_perm DPP moves — no LDS)
// hands every lane its own record; stores do the inverse. All four lanes must take part: the helpers check with a ballot
// that the whole quad is executing the call with a real pixel (`valid`) and fall back to the per-lane form otherwise
// (image edges, divergent callers), so they are safe anywhere and fast where the quad is convergent.
ST_D bool quad_all(bool p) { const unsigned long long b = __ballot(p); return ((b >> (threadIdx.x & 60u)) & 0xFull) == 0xFull; }
ST_D bool quad_any(bool p) { const unsigned long long b = __ballot(p); return ((b >> (threadIdx.x & 60u)) & 0xFull) != 0ull; }
// component-wise select (`c ? a : b` on the vector class can be lowered as a select of two stack addresses -> scratch)
ST_D float4 sel4(bool c, float4 a, float4 b) { return make_float4(c ? a.x : b.x, c ? a.y : b.y, c ? a.z : b.z, c ? a.w : b.w); }
template <int CTRL>
ST_D float4 quad_dpp(float4 v) {
    float4 r;
    r.x = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.x), CTRL, 0xf, 0xf, true));
    r.y = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.y), CTRL, 0xf, 0xf, true));
    r.z = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.z), CTRL, 0xf, 0xf, true));
    r.w = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v.w), CTRL, 0xf, 0xf, true));
    return r;
}
// v[k] in quad lane j holds M[k][j]; afterwards it holds M[j][k]
ST_D void quad_transpose(float4& v0, float4& v1, float4& v2, float4& v3) {
    const bool b0 = (threadIdx.x & 1u) != 0u, b1 = (threadIdx.x & 2u) != 0u;
    {   // lanes j and j ^ 1 (quad_perm [1,0,3,2] = 0xB1): 2x2 blocks of rows (0,1) and (2,3)
        const float4 ra = quad_dpp<0xB1>(sel4(b0, v0, v1)), rb = quad_dpp<0xB1>(sel4(b0, v2, v3));
        if (b0) { v0 = ra; v2 = rb; } else { v1 = ra; v3 = rb; }
    }
    {   // lanes j and j ^ 2 (quad_perm [2,3,0,1] = 0x4E): rows (0,2) and (1,3)
        const float4 ra = quad_dpp<0x4E>(sel4(b1, v0, v2)), rb = quad_dpp<0x4E>(sel4(b1, v1, v3));
        if (b1) { v0 = ra; v1 = rb; } else { v2 = ra; v3 = rb; }
    }
}
// The reservoir of this lane's own pixel. `valid`: `id` is a pixel of this launch (lanes of a quad then hold consecutive
// ids); `want`: this lane needs the record (a lane that does not still helps move its neighbours').
ST_D GiReservoir gi_read_own(const float4* buf, uint32_t id, bool valid, bool want) {
    want = want && valid;
    if (quad_all(valid)) {
        if (!quad_any(want)) return gi_empty();
        const uint32_t j = threadIdx.x & 3u, first = id - j;
        float4 v0 = buf[4u * first + j], v1 = buf[4u * (first + 1u) + j], v2 = buf[4u * (first + 2u) + j], v3 = buf[4u * (first + 3u) + j];
        quad_transpose(v0, v1, v2, v3);
        return want ? gi_from_texels(v0, v1, v2, v3) : gi_empty();
    }
    if (!want) return gi_empty();
    return gi_from_texels(buf[4u * id], buf[4u * id + 1u], buf[4u * id + 2u], buf[4u * id + 3u]);
}
ST_D void gi_write_own(float4* buf, uint32_t id, const GiReservoir& r, bool valid, bool want) {
    want = want && valid;
    const V2 n = normal_encode(r.s.v2_normal);
    float4 v0 = f4(r.s.radiance, r.m), v1 = f4(r.s.v1_point, r.w), v2 = f4(r.s.v2_point, r.s.pdf), v3 = make_float4(n.x, n.y, r.confidence, b2f(r.s.rng));
    if (quad_all(want)) {
        const uint32_t j = threadIdx.x & 3u, first = id - j;
        quad_transpose(v0, v1, v2, v3);
        buf[4u * first + j] = v0; buf[4u * (first + 1u) + j] = v1; buf[4u * (first + 2u) + j] = v2; buf[4u * (first + 3u) + j] = v3;
        return;
    }
    if (!want) return;
    buf[4u * id] = v0; buf[4u * id + 1u] = v1; buf[4u * id + 2u] = v2; buf[4u * id + 3u] = v3;
}
// ---- quad-cooperative gathers: ANY record per lane (spatial-neighbour reservoirs)
// A lane that fetches a neighbour's 64-B reservoir with gi_read issues four 16-B loads, and across the wave every one of them
// lands on 64 different 64-B segments: the texture-address unit serialises them (tools/tap_probe.hip: 64-B records read per
// lane 127-158 us, a whole record per quad and instruction 99 us; k_gi_spatial_fused shows the far a-trous kernel's
// signature in the SQ counters — 42 % of its waves' life stalled at issue). Here the four lanes of a quad fetch ONE lane's
// record per instruction — 64 contiguous bytes, one segment — for each lane of the quad that wants one (the record index is
// broadcast with a DPP move; quads in which no lane wants that round skip it), and the 4x4 transpose of quad_transpose
// hands every lane its own record. Pure data movement: same bytes, same values. The call must be made by all four lanes
// of a quad together (`want` says who needs a record); where the quad is not whole — divergent callers, image edges —
// the helper falls back to gi_read, so it is safe anywhere.
template <int CTRL>
ST_D uint32_t quad_dpp_u(uint32_t v) { return (uint32_t)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true); }
#ifdef ST_NO_COOP_GATHER  // A/B (tools/ab_bench.sh): every gather takes the per-lane fallback
ST_D bool quad_whole() { return false; }
#else
ST_D bool quad_whole() { const unsigned long long b = __ballot(true); return ((b >> (threadIdx.x & 60u)) & 0xFull) == 0xFull; }
#endif
ST_D GiReservoir gi_read_coop(const float4* buf, uint32_t id, uint32_t count, bool want) {
    want = want && id < count;
    if (!quad_whole()) return want ? gi_read(buf, id, count) : gi_empty();
    const uint32_t j = threadIdx.x & 3u;
    const uint32_t w = want ? 1u : 0u;
    float4 v0 = f4z(), v1 = f4z(), v2 = f4z(), v3 = f4z();
    if (quad_dpp_u<0x00>(w)) v0 = buf[4u * quad_dpp_u<0x00>(id) + j];   // lane 0's record: the quad's lanes take its four texels
    if (quad_dpp_u<0x55>(w)) v1 = buf[4u * quad_dpp_u<0x55>(id) + j];
    if (quad_dpp_u<0xAA>(w)) v2 = buf[4u * quad_dpp_u<0xAA>(id) + j];
    if (quad_dpp_u<0xFF>(w)) v3 = buf[4u * quad_dpp_u<0xFF>(id) + j];
    quad_transpose(v0, v1, v2, v3);
    return want ? gi_from_texels(v0, v1, v2, v3) : gi_empty();
}
// the inverse: every lane with `want` stores its record at its own index, one whole record per quad and instruction
ST_D void gi_write_coop(float4* buf, uint32_t id, const GiReservoir& r, bool want) {
    if (!quad_whole()) { if (want) gi_write(buf, id, r); return; }
    const V2 n = normal_encode(r.s.v2_normal);
    float4 v0 = f4(r.s.radiance, r.m), v1 = f4(r.s.v1_point, r.w), v2 = f4(r.s.v2_point, r.s.pdf), v3 = make_float4(n.x, n.y, r.confidence, b2f(r.s.rng));
    quad_transpose(v0, v1, v2, v3);   // lane j now holds texel j of lane k's record in v<k>
    const uint32_t j = threadIdx.x & 3u;
    const uint32_t w = want ? 1u : 0u;
    if (quad_dpp_u<0x00>(w)) buf[4u * quad_dpp_u<0x00>(id) + j] = v0;
    if (quad_dpp_u<0x55>(w)) buf[4u * quad_dpp_u<0x55>(id) + j] = v1;
    if (quad_dpp_u<0xAA>(w)) buf[4u * quad_dpp_u<0xAA>(id) + j] = v2;
    if (quad_dpp_u<0xFF>(w)) buf[4u * quad_dpp_u<0xFF>(id) + j] = v3;
}
// The same for 32-B records (two float4: a DI reservoir, a path tracer hit): the quad's four records are eight consecutive
// texels; instruction k moves texels 4k..4k+3 (64 contiguous bytes per quad), and lane j wants texels 2j and 2j+1 = lanes
// (2j mod 4), (2j+1 mod 4) of instruction j / 2 — two quad_perm moves per register and a select. `ok` is false (and the
// texels zero) for a lane that does not want its record. (Returned by value: output pointers ended up in scratch memory.)
struct Rec2 { float4 d0, d1; bool ok; };
ST_D Rec2 rec2_read_own(const float4* buf, uint32_t id, bool valid, bool want) {
    want = want && valid;
    Rec2 r; r.d0 = f4z(); r.d1 = f4z(); r.ok = false;
    if (quad_all(valid)) {
        if (!quad_any(want)) return r;
        const uint32_t j = threadIdx.x & 3u, first = id - j;
        const float4 v0 = buf[2u * first + j], v1 = buf[2u * first + 4u + j];
        const bool hi = j >= 2u;
        const float4 e0 = quad_dpp<0x88>(v0), e1 = quad_dpp<0x88>(v1);  // quad_perm [0,2,0,2]
        const float4 o0 = quad_dpp<0xDD>(v0), o1 = quad_dpp<0xDD>(v1);  // quad_perm [1,3,1,3]
        if (want) { r.d0 = sel4(hi, e1, e0); r.d1 = sel4(hi, o1, o0); r.ok = true; }
        return r;
    }
    if (!want) return r;
    r.d0 = buf[2u * id]; r.d1 = buf[2u * id + 1u]; r.ok = true;
    return r;
}
ST_D void rec2_write_own(float4* buf, uint32_t id, float4 d0, float4 d1, bool valid, bool want) {
    want = want && valid;
    if (quad_all(want)) {
        // lane j holds texels 2j, 2j+1 of the quad's eight; instruction k stores texel 4k + j: texel t lives in lane t / 2,
        // register t % 2, so instruction 0 takes lanes [0,0,1,1] and instruction 1 lanes [2,2,3,3], even lanes d0, odd lanes d1
        const uint32_t j = threadIdx.x & 3u, first = id - j;
        const bool odd = (j & 1u) != 0u;
        const float4 a0 = quad_dpp<0x50>(d0), a1 = quad_dpp<0x50>(d1);  // quad_perm [0,0,1,1]
        const float4 b0 = quad_dpp<0xFA>(d0), b1 = quad_dpp<0xFA>(d1);  // quad_perm [2,2,3,3]
        buf[2u * first + j] = sel4(odd, a1, a0);
        buf[2u * first + 4u + j] = sel4(odd, b1, b0);
        return;
    }
    if (!want) return;
    buf[2u * id] = d0; buf[2u * id + 1u] = d1;
}
// (Measured and not used for the DI reservoirs: same-box A/B of the Cornell frame 1.062 ms with these accessors in DI
// resolving and DI sampling + temporal against 1.053 ms without — 32-B records are half-line accesses already, and both
// kernels are bound by VALU issue, to which the ballots and DPP moves add. The 64-B GI reservoirs gain 25 %.)
ST_D bool gi_exists(const GiSample& s) { return !is_zero(s.v2_point); }
ST_D V3 gi_dir(const GiSample& s, V3 p) { return normalize(s.v2_point - p); }
ST_D float gi_cosine(const GiSample& s, const Hit& hit) { return fmax_(dot(gi_dir(s, hit.point), hit.g.normal), 0.0f); }
ST_D V3 gi_spec_brdf(const GiSample& s, const Hit& hit) { return specular_eval(hit.g, gi_dir(s, hit.point), -hit.dir); }
ST_D float gi_pdf(const GiSample& s, Hit hit) {  // reservoir/gi.rs:98-112
    if (!gi_exists(s)) return 0.0f;
    hit.g.base_color = make_float4(1.0f, 1.0f, 1.0f, 1.0f);
    const float d = luma(diffuse_eval(hit.g));
    const float sp = luma(gi_spec_brdf(s, hit));
    return luma(s.radiance) * gi_cosine(s, hit) * (d + sp);
}
ST_D Ray gi_sample_ray(const GiSample& s, V3 hit_point) {
    Ray r = make_ray(hit_point, gi_dir(s, hit_point));
    r.len = distance(s.v2_point, hit_point) - 0.01f;
    return r;
}
ST_D void gi_partial_jacobian(const GiSample& s, V3 hit_point, float* dist, float* cosv) {
    const V3 vec = hit_point - s.v2_point;
    *dist = length(vec);
    *cosv = saturate(dot(s.v2_normal, vec / *dist));
}
ST_D float gi_jacobian(const GiSample& s, V3 new_hit_point) {
    if (!gi_exists(s)) return 1.0f;
    float nd, nc, od, oc;
    gi_partial_jacobian(s, new_hit_point, &nd, &nc);
    gi_partial_jacobian(s, s.v1_point, &od, &oc);
    const float x = nc * od * od, y = oc * nd * nd;
    return y == 0.0f ? 0.0f : fdiv(x, y);
}

// Reservoir<T>::update / merge / norm as free templates (reservoir.rs:24-79)
template <class R, class S>
ST_D bool res_update(R& r, WhiteNoise& wn, const S& sample, float weight) {
    r.m += 1.0f; r.w += weight;
    if (wn.sample() * r.w < weight) { r.s = sample; return true; }
    return false;
}
template <class R>
ST_D bool res_merge(R& r, WhiteNoise& wn, const R& other, float pdf) {
    if (other.m <= 0.0f) return false;
    r.m += other.m - 1.0f;
    return res_update(r, wn, other.s, other.w * other.m * pdf);
}
template <class R>
ST_D void res_norm(R& r, float pdf, float num, float denom_) { const float denom = pdf * denom_; r.w = denom == 0.0f ? 0.0f : fdiv(r.w * num, denom); }

// EphemeralReservoir::build (reservoir/ephemeral.rs:14-55): RIS over min(16, light_count) uniformly picked lights
struct EphemeralResult { uint32_t light_id; LightRadiance light_rad; float m, w; };
ST_D EphemeralResult ephemeral_build(const KArgs& a, WhiteNoise& wn, const Hit& hit) {
    EphemeralResult res; res.light_id = 0u; res.light_rad.radiance = res.light_rad.diff_brdf = res.light_rad.spec_brdf = v3s(0.0f); res.m = 0.0f; res.w = 0.0f;
    float res_pdf = 0.0f;
    const uint32_t max_samples = a.light_count < 16u ? a.light_count : 16u;
    const float sample_ipdf = (float)a.light_count;
    for (uint32_t nth = 0; nth < max_samples; nth++) {
        const uint32_t light_id = wn.sample_int() % a.light_count;
        const LightRadiance rad = light_radiance(light_get(a, light_id), hit);
        const float sample_pdf = fsqrt(luma(rad.radiance));  // perc_luma
        const float weight = sample_pdf * sample_ipdf;
        res.m += 1.0f; res.w += weight;
        if (wn.sample() * res.w < weight) { res.light_id = light_id; res.light_rad = rad; res_pdf = sample_pdf; }
    }
    { const float denom = res_pdf * res.m; res.w = denom == 0.0f ? 0.0f : fdiv(res.w * 1.0f, denom); }  // norm_avg
    return res;
}

// defensive pairwise MIS (reservoir/mis.rs:96-144)
struct Mis { float lhs_m, rhs_m, rhs_jacobian, lhs_lhs_pdf, lhs_rhs_pdf, rhs_lhs_pdf, rhs_rhs_pdf; };
struct MisResult { float m, lhs_pdf, lhs_mis, rhs_pdf, rhs_mis; };
ST_D float mis2(float x, float y) { const float sum = x + y; return sum == 0.0f ? 0.0f : fdiv(x, sum); }
ST_D float mis_mfac(float q0, float q1) { return q0 <= 0.0f ? 1.0f : saturate(pow8_(fmin_(fdiv(q1, q0), 1.0f))); }
ST_D MisResult mis_eval(const Mis& s) {
    MisResult r;
    r.m = s.rhs_m * fmin_(mis_mfac(s.rhs_rhs_pdf, s.rhs_lhs_pdf), mis_mfac(s.lhs_rhs_pdf, s.lhs_lhs_pdf));
    const float t = mis2(s.lhs_m, s.rhs_m);
    r.lhs_mis = t + (1.0f - t) * mis2(s.lhs_m * s.lhs_lhs_pdf, s.rhs_m * s.lhs_rhs_pdf);
    r.rhs_mis = (1.0f - t) * mis2(s.rhs_m * s.rhs_rhs_pdf * s.rhs_jacobian, s.lhs_m * s.rhs_lhs_pdf);
    r.lhs_pdf = s.lhs_lhs_pdf; r.rhs_pdf = s.rhs_lhs_pdf;
    return r;
}

}  // namespace st
