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
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    Candidate c; bool any;
    const uint32_t used = traverse<false, SE>(a, camera_ray(a.cam, pos), kF32Max, lane_stack(a, lds), &c, &any);
    count_rays(a, used);
    a.dbg_used_memory[screen_to_idx(a, pos)] = used;
    tex_write(a.ref_colors, a, pos, f4(heatmap_gradient((float)used / 8192.0f), 1.0f));
}
void launch_bvh_heatmap(const KArgs& a, hipStream_t s) { ST_LAUNCH_TRACE(k_bvh_heatmap, false, s, a); }

// ---------------------------------------------------------------- ref_tracing.rs:3-60
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_ref_tracing(const KArgs a_in, uint32_t depth) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
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
    TriangleHit hit;
#if ST_FAST_DEVICE
    if (!LDS_SCENE && depth == 0u && a.bvh_w != nullptr && a.primary_packets) {   // camera rays: one packet per wave (depth is uniform: the branch is)
        Candidate c;
        const bool any = closest_hit_packet(a, ray, &c);
        hit = closest_resolve(a, ray, c, any);
    } else
#endif
    hit = trace_closest(a, ray, lane_stack(a, lds), &used_);
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
    ST_STACK_LDS(SE, lds);
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
        const bool occluded = trace_any(a, light_ray_wnoise(light, wn, hit.point), lane_stack(a, lds), &used_);
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
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Ray ray = camera_ray(a.cam, pos);
    TriangleHit hit;
#if ST_FAST_DEVICE
    // PRIMARY hits are exact in the fast build too (round 6): Triangle::hit, the attribute interpolation and the normal's octahedral code in the island's
    // arithmetic (st_device.h closest_resolve_exact says why: the sign of a decoded normal's z decides every hemisphere sample's tangent frame)
    // (a scene that lives in LDS — the Cornell box — keeps round 5's path: its contract walk is the island's already, its gates have 17 dB to spare, and the
    // exact interpolation + octahedral code would cost its headline 2-3 us per frame)
    if (LDS_SCENE) hit = trace_closest(a, ray, lane_stack(a, lds), &used_);
    else {
        Candidate c; bool any;
        if (a.bvh_w != nullptr && a.primary_packets) any = closest_hit_packet(a, ray, &c);   // the tile's 64 primary rays as one packet over the wide stream
        else if (a.bvh_w != nullptr) any = closest_hit_wide<SE, true>(a, ray, lane_stack(a, lds), &c);
        else if (a.bvh_c != nullptr) any = closest_hit_compact(a, ray, lane_stack(a, lds), &c);
        else used_ = traverse<false>(a, ray, kF32Max, lane_stack(a, lds), &c, &any);
        hit = closest_resolve_exact(a, ray, c, any);
    }
#else
    hit = trace_closest(a, ray, lane_stack(a, lds), &used_);
#endif
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
#if ST_FAST_DEVICE
    const V2 en = LDS_SCENE ? normal_encode(hit.normal) : normal_encode_exact(hit.normal);
    d0.y = en.x; d0.z = en.y;
#else
    const V2 en = normal_encode(hit.normal);
#endif
    tex_write(a.g0, a, pos, d0);
    tex_write(a.g1, a, pos, d1);
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
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const float4 ray_d0 = tex_read(buf_d0, a, pos), ray_d1 = tex_read(buf_d1, a, pos);
    if (is_zero(ray_d1)) { tex_write(buf_d2, a, pos, f4z()); return; }
    Ray ray = make_ray(xyz(ray_d0), normal_decode(v2(ray_d1.x, ray_d1.y)));
    ray.len = ray_d0.w;
    const bool occluded = trace_any(a, ray, lane_stack(a, lds), &used_);
    count_rays(a, used_);
    tex_write(buf_d2, a, pos, make_float4(occluded ? 0.0f : 1.0f, ray_d1.z, ray_d1.w, 0.0f));
}
void launch_spatial_trace(const KArgs& a, const float4* buf_d0, const float4* buf_d1, float4* buf_d2, hipStream_t s) {
    ST_LAUNCH_TRACE(k_spatial_trace, false, s, a, buf_d0, buf_d1, buf_d2);
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
