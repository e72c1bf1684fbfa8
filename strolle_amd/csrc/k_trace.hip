// k_trace.hip — the ray-tracing kernels that are not tied to a ReSTIR stage: BVH heatmap, the reference
// path tracer (trace + shade + accumulate), primary visibility (the compute restatement of prim_raster),
// the shared shadow-ray pass of spatial resampling, and frame composition.
#include <cstdlib>
#include "k_common.h"

namespace st {
namespace ST_KNS {

// ---------------------------------------------------------------- bvh_heatmap.rs:3-77
ST_D V3 heatmap_gradient(float progress) {
    const V3 c0 = v3(0.0f, 0.0f, 1.0f), c1 = v3(0.0f, 1.0f, 0.0f), c2 = v3(1.0f, 0.0f, 0.0f), c3 = v3(0.0f, 0.0f, 0.0f);
    if (progress <= 0.0f) return c0;
    const float step = 1.0f / (4.0f - 1.0f);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        const float mn = step * (float)i;
        const float mx = step * ((float)i + 1.0f);
        if (progress >= mn && progress <= mx) {
            const float rhs = fdivc(progress - mn, step);
            const float lhs = 1.0f - rhs;
            const V3 a = i == 0 ? c0 : (i == 1 ? c1 : c2), b = i == 0 ? c1 : (i == 1 ? c2 : c3);
            return lhs * a + rhs * b;
        }
    }
    return c3;
}
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_bvh_heatmap(const KArgs a_in) {
    ST_SCENE_PROLOGUE
    __shared__ SE lds[kStackWords];
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    Candidate c; bool any;
    const uint32_t used = traverse<false, SE>(a, camera_ray(a.cam, pos), kF32Max, lane_stack(lds), &c, &any);
    count_rays(a, used);
    a.dbg_used_memory[screen_to_idx(a, pos)] = used;
    tex_write(a.ref_colors, a, pos, f4(heatmap_gradient((float)used / 8192.0f), 1.0f));
}
void launch_bvh_heatmap(const KArgs& a, hipStream_t s) { ST_LAUNCH_TRACE(k_bvh_heatmap, false, s, a); }

// ---------------------------------------------------------------- ref_tracing.rs:3-60
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_ref_tracing(const KArgs a_in, uint32_t depth) {
    ST_SCENE_PROLOGUE
    __shared__ SE lds[kStackWords];
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t idx = screen_to_idx(a, pos);
    Ray ray;
    if (depth == 0u) ray = camera_ray(a.cam, pos);
    else {
        const float4 d0 = a.ref_rays[3u * idx], d1 = a.ref_rays[3u * idx + 1u];
        if (is_zero(d1)) return;
        ray = make_ray(xyz(d0), xyz(d1));
    }
    const TriangleHit hit = trace_closest(a, ray, lane_stack(lds), &used_);
    count_rays(a, used_);
    float4 h0, h1;
    hit_pack(hit, &h0, &h1);
    rec2_write_own(a.ref_hits, idx, h0, h1, true, true);  // quad-transposed 32-B records (st_device.h)
}
void launch_ref_tracing(const KArgs& a, uint32_t depth, hipStream_t s) { ST_LAUNCH_TRACE(k_ref_tracing, false, s, a, depth); }

// ---------------------------------------------------------------- ref_shading.rs:3-177
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_ref_shading(const KArgs a_in, uint32_t seed, uint32_t depth) {
    ST_SCENE_PROLOGUE
    __shared__ SE lds[kStackWords];
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t idx = screen_to_idx(a, pos);
    WhiteNoise wn = white_noise(seed, pos);
    if (depth == 255u) {  // accumulate
        const float4 prev = camera_is_eq(a.cam, a.prev_cam) ? tex_read(a.ref_colors, a, pos) : f4z();
        const V3 curr = xyz(a.ref_rays[3u * idx + 2u]);
        tex_write(a.ref_colors, a, pos, prev + f4(curr, 1.0f));
        return;
    }
    Ray ray; V3 color, throughput;
    if (depth == 0u) { ray = camera_ray(a.cam, pos); color = v3s(0.0f); throughput = v3s(1.0f); }
    else {
        const float4 d0 = a.ref_rays[3u * idx], d1 = a.ref_rays[3u * idx + 1u], d2 = a.ref_rays[3u * idx + 2u];
        ray = make_ray(xyz(d0), xyz(d1));
        color = xyz(d2);
        throughput = v3(d0.w, d1.w, d2.w);
    }
    const Rec2 packed_hit = rec2_read_own(a.ref_hits, idx, true, true);
    const TriangleHit t_hit = hit_unpack(packed_hit.d0, packed_hit.d1);
    if (!hit_is_some(t_hit)) {
        color = color + throughput * atmosphere_sample(a, ray.dir);
        a.ref_rays[3u * idx] = f4z();
        a.ref_rays[3u * idx + 1u] = f4z();
        a.ref_rays[3u * idx + 2u] = f4(color, 0.0f);
        return;
    }
    GpuMaterial material = a.materials[t_hit.material_id];
    if (depth > 0u) material.roughness = fmax_(material.roughness, 0.75f * 0.75f);  // Material::regularize
    Hit hit;
    hit.point = t_hit.point + t_hit.normal * kNudgeOffset;
    hit.origin = ray.origin; hit.dir = ray.dir;
    hit.g.base_color = sample_atlas(a, t_hit.uv, material.base_color, material.base_color_texture);
    hit.g.normal = t_hit.normal;
    hit.g.metallic = material.metallic;
    hit.g.emissive = xyz(sample_atlas(a, t_hit.uv, material.emissive, material.emissive_texture));
    hit.g.roughness = material.roughness;
    hit.g.reflectance = material.reflectance;
    hit.g.depth = 0.0f;

    color = color + throughput * hit.g.emissive;
    if (a.light_count > 0u) {
        const uint32_t light_id = wn.sample_int() % a.light_count;
        const float light_pdf = frcp((float)a.light_count);
        const GpuLight light = light_get(a, light_id);
        const bool occluded = trace_any(a, light_ray_wnoise(light, wn, hit.point), lane_stack(lds), &used_);
        count_rays(a, used_);
        if (!occluded) color = color + throughput * radiance_sum(light_radiance(light, hit)) / light_pdf;
    }
    const BrdfSample rs = layered_brdf_sample(hit.g, wn, -hit.dir);
    if (rs.pdf == 0.0f) { a.ref_rays[3u * idx] = f4z(); a.ref_rays[3u * idx + 1u] = f4z(); return; }
    throughput = throughput * dot(rs.dir, hit.g.normal);
    throughput = throughput * (rs.radiance / rs.pdf);
    a.ref_rays[3u * idx] = f4(hit.point, throughput.x);
    a.ref_rays[3u * idx + 1u] = f4(rs.dir, throughput.y);
    a.ref_rays[3u * idx + 2u] = f4(color, throughput.z);
}
void launch_ref_shading(const KArgs& a, uint32_t seed, uint32_t depth, hipStream_t s) { ST_LAUNCH_TRACE(k_ref_shading, false, s, a, seed, depth); }

// ---------------------------------------------------------------- primary visibility (prim_raster.rs:40-128 as one closest-hit ray per pixel)
// frame_reprojection.rs:6-95 for one pixel, given its own fresh surface and velocity
ST_D void frame_reprojection_pixel(const KArgs& a, U2 pos, const Surface& surface, V2 velocity) {
    Reprojection rp; rp.prev_x = 0.0f; rp.prev_y = 0.0f; rp.confidence = 0.0f; rp.validity = 0u;
    if (surface.depth != 0.0f) {
        const V2 prev_screen_pos = as_v2(pos) - velocity;
        const V2 rounded = round2(prev_screen_pos);
        if (contains_f(a, rounded)) {
            const float confidence = surface_similarity(surface_decoded(tex_read(a.psn, a, as_u2(rounded))), surface);
            if (confidence > 0.0f) { rp.prev_x = prev_screen_pos.x; rp.prev_y = prev_screen_pos.y; rp.confidence = confidence; }
        }
        if (rp.confidence > 0.0f) {
            const float fl_x = floorf(rp.prev_x), fl_y = floorf(rp.prev_y), ce_x = ceilf(rp.prev_x), ce_y = ceilf(rp.prev_y);
            const I2 p[4] = {i2(f2i_sat(fl_x), f2i_sat(fl_y)), i2(f2i_sat(ce_x), f2i_sat(fl_y)), i2(f2i_sat(fl_x), f2i_sat(ce_y)), i2(f2i_sat(ce_x), f2i_sat(ce_y))};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                if (!contains_i(a, p[i])) continue;
                if (surface_similarity(surface_decoded(tex_read(a.psn, a, u2((uint32_t)p[i].x, (uint32_t)p[i].y))), surface) >= 0.25f) rp.validity |= (1u << i);
            }
        }
    }
    tex_write(a.reprojection, a, pos, make_float4(rp.prev_x, rp.prev_y, rp.confidence, b2f(rp.validity)));
}

// REPROJECT: frame_reprojection runs in the same kernel (it needs this pixel's new surface + velocity and the PREVIOUS
// frame's surfaces only).
template <bool LDS_SCENE, bool REPROJECT, class SE>
__global__ ST_KERNEL_BOUNDS void k_prim_visibility(const KArgs a_in) {
    ST_SCENE_PROLOGUE
    __shared__ SE lds[kStackWords];
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Ray ray = camera_ray(a.cam, pos);
    const TriangleHit hit = trace_closest(a, ray, lane_stack(lds), &used_);
    count_rays(a, used_);
    if (!hit_is_some(hit)) {  // LoadOp::Clear(TRANSPARENT)
        tex_write(a.g0, a, pos, f4z()); tex_write(a.g1, a, pos, f4z());
        if (!(REPROJECT && (a.lean & kLeanPrim))) { tex_write(a.sm, a, pos, f4z()); tex_write(a.velocity, a, pos, f4z()); }
        tex_write(a.sn, a, pos, f4z());
        if (REPROJECT) tex_write(a.reprojection, a, pos, f4z());
        return;
    }
    const GpuMaterial material = a.materials[hit.material_id];
    const float4 mr = sample_atlas(a, hit.uv, make_float4(1.0f, material.roughness, material.metallic, 1.0f), material.metallic_roughness_texture);
    GBuffer g;
    g.base_color = sample_atlas(a, hit.uv, material.base_color, material.base_color_texture);
    g.normal = hit.normal;
    g.metallic = mr.z;
    g.emissive = xyz(sample_atlas(a, hit.uv, material.emissive, material.emissive_texture));
    g.roughness = mr.y;
    g.reflectance = material.reflectance;
    g.depth = distance(ray.origin, hit.point);
    float4 d0, d1;
    gbuffer_pack_bits(g, (a.material_base_packed && is_zero(material.base_color_texture)) ? a.material_base_packed[hit.material_id] : gbuffer_pack_base_color(g.base_color), &d0, &d1);
    tex_write(a.g0, a, pos, d0);
    tex_write(a.g1, a, pos, d1);
    const V2 en = normal_encode(hit.normal);
    const bool lean = REPROJECT && (a.lean & kLeanPrim) != 0u;
    if (!lean) tex_write(a.sm, a, pos, make_float4(en.x, en.y, g.depth, material.roughness));
    tex_write(a.sn, a, pos, f4(normal_decode(en), g.depth));
    // prim_raster.rs:21-27: where this surface point was under its instance's previous transform
    const float4* xf = a.instance_xforms + 8u * hit.xform_slot;
    const V3 prev_point = affine_point(xf + 4, affine_point(xf, hit.point));
    const V2 velocity = clip_to_screen(a.cam, world_to_clip(a.cam, hit.point)) - clip_to_screen(a.prev_cam, world_to_clip(a.prev_cam, prev_point));
    const bool moving = dot(velocity, velocity) >= 0.001f;
    if (!lean) tex_write(a.velocity, a, pos, moving ? make_float4(velocity.x, velocity.y, 0.0f, 0.0f) : f4z());
    if (REPROJECT) {
        Surface surface; surface.normal = normal_decode(en); surface.depth = g.depth; surface.roughness = 0.0f;
        frame_reprojection_pixel(a, pos, surface, moving ? velocity : v2(0.0f, 0.0f));
    }
}
void launch_prim_visibility(const KArgs& a, bool reproject, hipStream_t s) {
    if (reproject) ST_LAUNCH_TRACE_B(k_prim_visibility, true, false, s, a); else ST_LAUNCH_TRACE_B(k_prim_visibility, false, false, s, a);
}

// Tabulates the byte decodes of st_device.h with the routines themselves (one launch at engine creation).
__global__ void k_build_byte_luts(float* out) {
    const uint32_t v = threadIdx.x;
    out[kLutSrgb + v] = srgb_to_linear_eval(v);
    out[kLutUnorm8 + v] = unorm8_eval(v);
    out[kLutGamma8 + v] = gamma8_eval(v);
    out[kLutGamma6 + v] = gamma6_eval(v);
}
void launch_build_byte_luts(float* out, hipStream_t s) { ST_KLAUNCH(k_build_byte_luts, dim3(1), dim3(256), s, out); }

// ---------------------------------------------------------------- frame_reprojection.rs:6-95
__global__ ST_KERNEL_BOUNDS void k_frame_reprojection(const KArgs a) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const float4 vel = tex_read(a.velocity, a, pos);
    frame_reprojection_pixel(a, pos, surface_decoded(tex_read(a.sn, a, pos)), v2(vel.x, vel.y));
}
void launch_frame_reprojection(const KArgs& a, hipStream_t s) { ST_LAUNCH(k_frame_reprojection, false, s, a); }

// ---------------------------------------------------------------- {di,gi}_spatial_resampling.rs `trace`
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_spatial_trace(const KArgs a_in, const float4* buf_d0, const float4* buf_d1, float4* buf_d2) {
    ST_SCENE_PROLOGUE
    __shared__ SE lds[kStackWords];
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const float4 ray_d0 = tex_read(buf_d0, a, pos), ray_d1 = tex_read(buf_d1, a, pos);
    if (is_zero(ray_d1)) { tex_write(buf_d2, a, pos, f4z()); return; }
    Ray ray = make_ray(xyz(ray_d0), normal_decode(v2(ray_d1.x, ray_d1.y)));
    ray.len = ray_d0.w;
    const bool occluded = trace_any(a, ray, lane_stack(lds), &used_);
    count_rays(a, used_);
    tex_write(buf_d2, a, pos, make_float4(occluded ? 0.0f : 1.0f, ray_d1.z, ray_d1.w, 0.0f));
}
// The same pass with ray compaction (north_star: "wavefront ballot/prefix-sum for ray compaction"). A block owns
// kGroupsPerBlock groups of 4 tiles; its pixels' ray records are gathered with a wave ballot + prefix sum into a dense
// pool in LDS (pixels without a ray write their zero and leave), and the block's four waves then run as persistent
// workers: a lane whose shadow ray ends fetches the next pool entry instead of idling until its wave's longest traversal
// is over. Which lane traces a ray changes, the ray's arithmetic does not; any-hit results do not depend on the order.
// MEASURED SLOWER, hence opt-in (ST_COMPACT=1): 137 vs 92 us on Cornell, 152 vs 83 us on the dungeon at 1080p, although
// lane utilisation of the plain kernel is only 0.56 / 0.32 — the resumable-step loop, the scattered result writes and the
// LDS pool (5 instead of 8 waves per SIMD) cost more than the idle lanes did. Kept as the starting point for round 2.
constexpr uint32_t kGroupsPerBlock = 2u, kPoolRays = kGroupsPerBlock * 256u;
template <class SE>
__global__ ST_KERNEL_BOUNDS void k_spatial_trace_compact(const KArgs a, const float4* buf_d0, const float4* buf_d1, float4* buf_d2,
                                                                          uint32_t groups_x, uint32_t tile_y0, uint32_t n_groups) {
    __shared__ SE lds[kStackWords];
    __shared__ float4 pool_d0[kPoolRays];
    __shared__ float4 pool_d1[kPoolRays];
    __shared__ uint32_t pool_px[kPoolRays];
    __shared__ uint32_t pool_n, pool_next;
    if (threadIdx.x == 0u) { pool_n = 0u; pool_next = 0u; }
    __syncthreads();
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    // contiguous ranges of blocks per XCD (hardware sends block b to XCD b % 8); shadow rays have no neighbour taps
    const uint32_t n_blocks = gridDim.x, q = n_blocks >> 3, r = n_blocks & 7u, xcd = blockIdx.x & 7u, k = blockIdx.x >> 3;
    const uint32_t sb = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + k;
    for (uint32_t g = 0; g < kGroupsPerBlock; g++) {
        const uint32_t lin = sb * kGroupsPerBlock + g;
        bool has = false; float4 d0 = f4z(), d1 = f4z(); uint32_t px = 0u;
        if (lin < n_groups) {
            const uint32_t ty = lin / groups_x, tx = (lin - ty * groups_x) * 4u + wave;
            const U2 pos = u2(tx * 8u + (lane & 7u), (ty + tile_y0) * 8u + (lane >> 3));
            if (owns_pixel(a, pos)) {
                px = pos.y * a.width + pos.x;
                d0 = buf_d0[px]; d1 = buf_d1[px];
                has = !is_zero(d1);
                if (!has) buf_d2[px] = f4z();
            }
        }
        const unsigned long long mask = __ballot(has);
        uint32_t base = 0u;
        if (lane == 0u && mask) base = atomicAdd(&pool_n, (uint32_t)__popcll(mask));
        base = __shfl(base, 0);
        if (has) { const uint32_t slot = base + (uint32_t)__popcll(mask & lt_mask); pool_d0[slot] = d0; pool_d1[slot] = d1; pool_px[slot] = px; }
    }
    __syncthreads();
    const uint32_t total = pool_n;
    SE* stack = lane_stack(lds);
    bool active = false, exhausted = false;
    Ray ray = zero_ray(); AnyHitState st = any_hit_begin(); float4 rec1 = f4z(); uint32_t px = 0u;
    uint32_t rays_done = 0u; unsigned long long bytes = 0ull;
    for (;;) {
        const unsigned long long idle = __ballot(!active);
        // refill when a quarter of the wave idles (each refill issues the ray set-up code for the whole wave)
        if (!exhausted && (__popcll(idle) >= 16 || idle == ~0ull)) {
            uint32_t base = 0u;
            if (lane == (uint32_t)__ffsll((long long)idle) - 1u) base = atomicAdd(&pool_next, (uint32_t)__popcll(idle));
            base = __shfl(base, __ffsll((long long)idle) - 1);
            if (!active) {
                const uint32_t slot = base + (uint32_t)__popcll(idle & lt_mask);
                if (slot < total) {
                    const float4 r0 = pool_d0[slot]; rec1 = pool_d1[slot]; px = pool_px[slot];
                    ray = make_ray(xyz(r0), normal_decode(v2(rec1.x, rec1.y)));
                    ray.len = r0.w;
                    st = any_hit_begin();
                    active = true;
                }
            }
            if (base + (uint32_t)__popcll(idle) >= total) exhausted = true;
        }
        if (!__ballot(active)) break;
#pragma unroll 1
        for (int burst = 0; burst < 4; burst++) {
            if (active && any_hit_step(a, ray, stack, st)) {
                buf_d2[px] = make_float4(st.found ? 0.0f : 1.0f, rec1.z, rec1.w, 0.0f);
                rays_done += 1u; bytes += st.used_memory;
                active = false;
            }
        }
    }
    if (rays_done) count_rays_many(a, rays_done, bytes);
}
void launch_spatial_trace(const KArgs& a, const float4* buf_d0, const float4* buf_d1, float4* buf_d2, hipStream_t s) {
    static const bool compact = [] { const char* e = getenv("ST_COMPACT"); return e && atoi(e) != 0; }();
    if (!compact) { ST_LAUNCH_TRACE(k_spatial_trace, false, s, a, buf_d0, buf_d1, buf_d2); return; }
    const LaunchDims d = launch_dims(a, false);
    const uint32_t groups_x = (d.tiles_x + 3u) / 4u, n_groups = d.blocks;
    if (!n_groups) return;
    const uint32_t blocks = (n_groups + kGroupsPerBlock - 1u) / kGroupsPerBlock;
    if (a.bvh_len < kStack16Texels) ST_KLAUNCH((k_spatial_trace_compact<uint16_t>), dim3(blocks), dim3(kBlockThreads), s, a, buf_d0, buf_d1, buf_d2, groups_x, d.tile_y0, n_groups);
    else ST_KLAUNCH((k_spatial_trace_compact<uint32_t>), dim3(blocks), dim3(kBlockThreads), s, a, buf_d0, buf_d1, buf_d2, groups_x, d.tile_y0, n_groups);
}

// ---------------------------------------------------------------- frame_composition.rs:18-82 as a compute pass into an RGBA32F buffer
// `format` (StOutputFormat) is what the reference leaves to the render target's format (camera.rs:170-175 viewport.format).
__global__ ST_KERNEL_BOUNDS void k_composition(const KArgs a, uint32_t camera_mode, const float4* di_diff, const float4* gi_diff, void* out, uint32_t format) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const float4 c = compose_pixel(a, pos, camera_mode, tex_read(di_diff, a, pos), tex_read(gi_diff, a, pos));
    store_output(out, pos.y * a.width + pos.x, c, format);
}
void launch_composition(const KArgs& a, uint32_t camera_mode, const float4* di_diff, const float4* gi_diff, void* out, uint32_t format, hipStream_t s) {
    ST_LAUNCH(k_composition, false, s, a, camera_mode, di_diff, gi_diff, out, format);
}

}  // namespace ST_KNS
}  // namespace st
