// k_di.hip — ReSTIR DI: initial sampling, temporal and spatial resampling, resolving.
// Behavioural contract: strolle-shaders/src/di_{sampling,temporal_resampling,spatial_resampling,resolving}.rs.
// Reservoir roles: sampling -> [1]; temporal reads [0], RMW [1]; spatial [1] -> [2]; resolving [2] -> [0]
// (strolle/src/camera_controller/passes/di_*.rs). Spatial scratch aliases di_diff_samples /
// di_diff_curr_colors / di_diff_stash (passes/di_spatial_resampling.rs:24-28).
#include "k_common.h"

namespace st {
namespace ST_KNS {

// ---------------------------------------------------------------- di_sampling.rs:3-94
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_di_sampling(const KArgs a_in, uint32_t seed) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Hit hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    if (!hit_some(hit)) return;
    di_write(a.di_res[1], screen_to_idx(a, pos), di_sampling_pixel(a, seed, pos, hit, lane_stack(a, lds)));
}
void launch_di_sampling(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH_TRACE(k_di_sampling, false, s, a, seed); }

// ---------------------------------------------------------------- di_temporal_resampling.rs:3-112
__global__ ST_KERNEL_BOUNDS void k_di_temporal(const KArgs a, uint32_t seed) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Hit hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    if (!hit_some(hit)) return;
    di_temporal_pixel(a, seed, pos, hit, di_read(a.di_res[1], screen_to_idx(a, pos), a.width * a.height), tex_read(a.reprojection, a, pos));
}
void launch_di_temporal(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH(k_di_temporal, false, s, a, seed); }

// di_sampling.rs + di_temporal_resampling.rs in one launch: temporal resampling reads the sampling pass's reservoir only at
// its own pixel, so it takes it from registers (through the store/load codec, which is not the identity) and the pixel's hit
// is rebuilt once; di_res[1] ends with the temporal result exactly as it does after the two separate passes.
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_di_sampling_temporal(const KArgs a_in, uint32_t seed_sampling, uint32_t seed_temporal) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Hit hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    if (!hit_some(hit)) return;
    di_temporal_pixel(a, seed_temporal, pos, hit, di_after_store(di_sampling_pixel(a, seed_sampling, pos, hit, lane_stack(a, lds))), tex_read(a.reprojection, a, pos));
}
void launch_di_sampling_temporal(const KArgs& a, uint32_t seed_sampling, uint32_t seed_temporal, hipStream_t s) {
    ST_LAUNCH_TRACE(k_di_sampling_temporal, false, s, a, seed_sampling, seed_temporal);
}

// ---------------------------------------------------------------- di_spatial_resampling.rs:3-147 (pick)
// What the pick stage left in the scratch planes for its cell's two pixels: `wrote_d1` / `wrote_d0` say which texels are
// fresh (an early exit on a sky pixel writes nothing and the next stages then see stale plane contents, as in the reference).
struct SpatialRecords { bool wrote_d0, wrote_d1; float4 a0, a1, b0, b1; };
ST_D SpatialRecords di_spatial_pick_cell(const KArgs& a, uint32_t seed, U2 gid, U2 lhs_pos) {
    SpatialRecords out; out.wrote_d0 = false; out.wrote_d1 = false; out.a0 = out.a1 = out.b0 = out.b1 = f4z();
    const uint32_t n = a.width * a.height;
    const uint32_t lhs_idx = screen_to_idx(a, lhs_pos);
    WhiteNoise wn = white_noise(seed, lhs_pos);
    float4* buf_d0 = a.di_diff_samples; float4* buf_d1 = a.di_diff_curr_colors;
    const U2 buf_pos_a = u2(gid.x * 2u, gid.y), buf_pos_b = u2(gid.x * 2u + 1u, gid.y);
    const Hit lhs_hit = pixel_hit(a, a.cam, a.g0, a.g1, lhs_pos);
    if (!hit_some(lhs_hit)) return out;
    const DiReservoir lhs = di_read(a.di_res[1], lhs_idx, n);
    DiReservoir rhs = di_empty();
    uint32_t rhs_nth = 0u, rhs_idx = 0u;
    Hit rhs_hit = hit_zero();
    float max_radius = 128.0f;
    while (rhs_nth < 8u) {
        rhs_nth += 1u;
        const V2 disk = wn.sample_disk();
        const U2 rhs_pos = camera_contain(a, as_i2(as_v2(lhs_pos) + disk * max_radius));
        if (rhs_pos.x == lhs_pos.x && rhs_pos.y == lhs_pos.y) continue;
        // (Fetching the candidate's reservoir together with its G-buffer texel — one round trip per try instead of two — measured
        // slower, 92 -> 110 us on Cornell 1080p: the pass is bound by memory transactions, not by their latency.)
        rhs_hit = pixel_hit(a, a.cam, a.g0, a.g1, rhs_pos);
        if (!hit_some(rhs_hit)) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
        if (fabsf(rhs_hit.g.depth - lhs_hit.g.depth) > 0.33f * lhs_hit.g.depth) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
        if (dot(rhs_hit.g.normal, lhs_hit.g.normal) < 0.33f) { max_radius = fmax_(max_radius * 0.5f, 5.0f); continue; }
        rhs_idx = screen_to_idx(a, rhs_pos);
        rhs = di_read(a.di_res[1], rhs_idx, n);
        if (rhs.m != 0.0f) break;
    }
    out.wrote_d1 = true;
    const bool store = !a.skip_dead_scratch;  // fused launch inside a whole denoised frame: the records travel in registers and the planes are rewritten before anything can read them
    if (rhs.m == 0.0f) { if (store) { tex_write(buf_d1, a, buf_pos_a, f4z()); tex_write(buf_d1, a, buf_pos_b, f4z()); } return out; }
    const float lhs_rhs_pdf = di_pdf_ex(lhs.s, light_get(a, lhs.s.light_id), rhs_hit);
    const float rhs_lhs_pdf = di_pdf_ex(rhs.s, light_get(a, rhs.s.light_id), lhs_hit);
    const Ray ray_a = lhs_rhs_pdf > 0.0f ? di_sample_ray(lhs.s, rhs_hit.point) : zero_ray();
    const Ray ray_b = rhs_lhs_pdf > 0.0f ? di_sample_ray(rhs.s, lhs_hit.point) : zero_ray();
    const V2 ea = normal_encode(ray_a.dir), eb = normal_encode(ray_b.dir);
    out.wrote_d0 = true;
    out.a0 = f4(ray_a.origin, ray_a.len); out.a1 = make_float4(ea.x, ea.y, b2f(rhs_idx + 1u), 0.0f);
    out.b0 = f4(ray_b.origin, ray_b.len); out.b1 = make_float4(eb.x, eb.y, lhs_rhs_pdf, rhs_lhs_pdf);
    if (store) {
        tex_write(buf_d0, a, buf_pos_a, out.a0); tex_write(buf_d1, a, buf_pos_a, out.a1);
        tex_write(buf_d0, a, buf_pos_b, out.b0); tex_write(buf_d1, a, buf_pos_b, out.b1);
    }
    return out;
}
__global__ ST_KERNEL_BOUNDS void k_di_spatial_pick(const KArgs a, uint32_t seed) {
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const U2 lhs_pos = resolve_checkerboard_alt(gid, a.frame / 2u);
    if (!owns_pixel(a, lhs_pos)) return;
    (void)di_spatial_pick_cell(a, seed, gid, lhs_pos);
}
void launch_di_spatial_pick(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH(k_di_spatial_pick, true, s, a, seed); }

// ---------------------------------------------------------------- di_spatial_resampling.rs:211-297 (sample)
// d0 / d1: the trace stage's texels for the cell's two pixels (di_diff_stash at (2 gid.x, gid.y) and (2 gid.x + 1, gid.y))
ST_D void di_spatial_sample_cell(const KArgs& a, uint32_t seed, U2 gid, U2 lhs_pos, float4 d0, float4 d1) {
    const uint32_t n = a.width * a.height;
    const uint32_t lhs_idx = screen_to_idx(a, lhs_pos);
    WhiteNoise wn = white_noise(seed, lhs_pos);
    const float4* in_res = a.di_res[1];
    float4* out_res = a.di_res[2];
    const float lhs_rhs_vis = d0.x;
    const uint32_t rhs_idx = f2b(d0.y);
    const float rhs_lhs_vis = d1.x, lhs_rhs_pdf = d1.y, rhs_lhs_pdf = d1.z;
    const DiReservoir lhs = di_read(in_res, lhs_idx, n);
    if (rhs_idx > 0u) {
        const DiReservoir rhs = di_read(in_res, rhs_idx - 1u, n);
        Mis mis;
        mis.lhs_m = lhs.m; mis.rhs_m = rhs.m; mis.rhs_jacobian = 1.0f; mis.lhs_lhs_pdf = lhs.s.pdf;
        mis.lhs_rhs_pdf = lhs_rhs_pdf * lhs_rhs_vis; mis.rhs_lhs_pdf = rhs_lhs_pdf * rhs_lhs_vis; mis.rhs_rhs_pdf = rhs.s.pdf;
        const MisResult mr = mis_eval(mis);
        DiReservoir main_ = di_empty();
        float main_pdf = 0.0f;
        if (res_update(main_, wn, lhs.s, mr.lhs_mis * mr.lhs_pdf * lhs.w)) main_pdf = mr.lhs_pdf;
        if (res_update(main_, wn, rhs.s, mr.rhs_mis * mr.rhs_pdf * rhs.w)) { main_pdf = mr.rhs_pdf; main_.s.is_occluded = lhs_rhs_vis == 0.0f; }
        main_.m = lhs.m + mr.m;
        main_.s.pdf = main_pdf;
        res_norm(main_, main_pdf, 1.0f, 1.0f);
        di_write(out_res, lhs_idx, main_);
    } else di_write(out_res, lhs_idx, lhs);
    const U2 other = resolve_checkerboard(gid, a.frame / 2u);
    if (contains_u(a, other)) { const uint32_t oi = screen_to_idx(a, other); di_write(out_res, oi, di_read(in_res, oi, n)); }
}
__global__ ST_KERNEL_BOUNDS void k_di_spatial_sample(const KArgs a, uint32_t seed) {
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const U2 lhs_pos = resolve_checkerboard_alt(gid, a.frame / 2u);
    if (!owns_pixel(a, lhs_pos)) return;
    di_spatial_sample_cell(a, seed, gid, lhs_pos, tex_read(a.di_diff_stash, a, u2(gid.x * 2u, gid.y)), tex_read(a.di_diff_stash, a, u2(gid.x * 2u + 1u, gid.y)));
}

// di_spatial_resampling.rs pick + trace + sample for one 2x1 checkerboard cell in one launch. The three passes of a cell talk
// to each other only through that cell's two texels of the scratch planes, so the records and visibilities travel in
// registers; they are still stored (later passes, or the next frame's stale reads, must find what the reference leaves
// there) unless the engine knows the rest of the frame rewrites all three planes (KArgs::skip_dead_scratch), and a texel the
// pick stage did not write is read back from the plane, stale contents included.
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_di_spatial_fused(const KArgs a_in, uint32_t seed_pick, uint32_t seed_sample) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const U2 lhs_pos = resolve_checkerboard_alt(gid, a.frame / 2u);
    const bool own_lhs = owns_pixel(a, lhs_pos);
    SpatialRecords rec; rec.wrote_d0 = false; rec.wrote_d1 = false; rec.a0 = rec.a1 = rec.b0 = rec.b1 = f4z();
    if (own_lhs) rec = di_spatial_pick_cell(a, seed_pick, gid, lhs_pos);
    float4 vis[2];
    uint32_t rays = 0u; unsigned long long bytes = 0ull;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const U2 pos = u2(gid.x * 2u + (uint32_t)k, gid.y);
        if (!owns_pixel(a, pos)) { vis[k] = tex_read(a.di_diff_stash, a, pos); continue; }  // what the sample stage would read there
        const float4 r0 = rec.wrote_d0 ? (k == 0 ? rec.a0 : rec.b0) : tex_read(a.di_diff_samples, a, pos);
        const float4 r1 = rec.wrote_d1 ? (k == 0 ? rec.a1 : rec.b1) : tex_read(a.di_diff_curr_colors, a, pos);
        if (is_zero(r1)) vis[k] = f4z();
        else {
            Ray ray = make_ray(xyz(r0), normal_decode(v2(r1.x, r1.y)));
            ray.len = r0.w;
            uint32_t used_ = 0u;
            const bool occluded = trace_any(a, ray, lane_stack(a, lds), &used_);
            rays += 1u; bytes += used_;
            vis[k] = make_float4(occluded ? 0.0f : 1.0f, r1.z, r1.w, 0.0f);
        }
        if (!a.skip_dead_scratch) tex_write(a.di_diff_stash, a, pos, vis[k]);
    }
    if (rays) count_rays_n(a, rays, bytes);
    if (own_lhs) di_spatial_sample_cell(a, seed_sample, gid, lhs_pos, vis[0], vis[1]);
}
void launch_di_spatial_fused(const KArgs& a, uint32_t seed_pick, uint32_t seed_sample, hipStream_t s) {
    ST_LAUNCH_TRACE(k_di_spatial_fused, true, s, a, seed_pick, seed_sample);
}
void launch_di_spatial_sample(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH(k_di_spatial_sample, true, s, a, seed); }

// ---------------------------------------------------------------- di_resolving.rs:3-119
// REPROJECT: the DI half of frame_denoising.rs::reproject is appended (it reads only this pixel's fresh diffuse sample
// plus previous-frame planes).
// (76 VGPRs with REPROJECT = 6 waves per SIMD; asking for 7 spills 4 registers and measures 78.2 -> 79.9 us.)
template <bool LDS_SCENE, bool REPROJECT, class SE>
__global__ ST_KERNEL_BOUNDS void k_di_resolving(const KArgs a_in) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t n = a.width * a.height;
    const uint32_t idx = screen_to_idx(a, pos);
    const Hit hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    DiReservoir res = di_read(a.di_res[2], idx, n);
    ReprojectHistory history;  // fetched ahead of the shadow ray (st_passes.h)
    if (REPROJECT) history = denoise_reproject_prefetch<true>(a, pos, a.di_diff_prev_colors, a.di_diff_prev_moments);
    float confidence;
    V3 radiance, spec_brdf;
    if (hit_some(hit)) {
        const bool occluded = trace_any(a, di_sample_ray(res.s, hit.point), lane_stack(a, lds), &used_);
        count_rays(a, used_);
        confidence = (res.s.is_occluded == occluded) ? res.s.confidence : 0.0f;
        res.s.confidence = 1.0f;
        res.s.is_occluded = occluded;
        if (occluded) { radiance = v3s(0.0f); spec_brdf = v3s(0.0f); }
        else { const LightRadiance lr = light_radiance(light_get(a, res.s.light_id), hit); radiance = lr.radiance * res.w; spec_brdf = lr.spec_brdf; }
    } else {
        confidence = 1.0f;
        radiance = atmosphere_sample(a, hit.dir);
        spec_brdf = v3s(0.0f);
    }
    const float diff_brdf = fdivc(1.0f - hit.g.metallic, kPi);
    const float4 diff = f4(radiance * diff_brdf, confidence);
    if (!(REPROJECT && (a.lean & kLeanSamples))) tex_write(a.di_diff_samples, a, pos, diff);
    tex_write(a.di_spec_samples, a, pos, f4(radiance * spec_brdf, confidence));
    di_write(a.di_res[0], idx, res);
    if (REPROJECT) {
        const bool short_history = denoise_reproject_finish(a, pos, diff, history, a.di_diff_curr_colors, a.di_diff_moments);
        if (a.variance_in_reproject) {  // tell the variance kernel which pixels of this tile still need it
            const unsigned long long flagged = __ballot(short_history), active = __ballot(true);
            if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)active) - 1u) a.tile_mask[tile_mask_index(a, pos)] = flagged;
        }
    }
}
void launch_di_resolving(const KArgs& a, bool reproject, hipStream_t s) {
    if (reproject) ST_LAUNCH_TRACE_B(k_di_resolving, true, false, s, a); else ST_LAUNCH_TRACE_B(k_di_resolving, false, false, s, a);
}

}  // namespace ST_KNS
}  // namespace st
