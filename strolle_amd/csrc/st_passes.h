// st_passes.h — per-pixel bodies of passes that exist both as stand-alone kernels (one per reference entry point) and
// inside fused kernels. Fusion rule: a pass may be appended to its producer when everything it reads from the current
// frame is the producer's own pixel; reads of *other* pixels must come from buffers no fused stage writes.
#pragma once
#include "st_device.h"

namespace st {

// frame_denoising.rs:3-78 (reproject) for one pixel; `sample` is this pixel's texel of the samples plane.
// Split in two so that a fused producer can issue the history loads (reprojection texel -> previous colour / moment taps,
// two dependent round trips) BEFORE its own long latency chain (a shadow-ray traversal, a resampling loop) and consume
// them after it: the loads do not depend on the sample. Fetching history that `sample.w <= 0` then ignores is harmless.
struct ReprojectHistory { bool sky, have; float4 pc, pm; };
// (`sky` and the pixel's reprojection texel come from the caller, who fetched them beside its other first-round loads; the colour and
// the moment taps then travel together: the history costs one dependent round trip, not three)
// TOGETHER: the colour and the moment taps travel as one round trip (eight texels live at once: +13 VGPRs — pays in di_resolving,
// 134 -> 129 us on the dungeon, costs gi_preview_both a wave of occupancy, 102 -> 105 us, which therefore keeps them apart)
template <bool TOGETHER>
ST_D ReprojectHistory denoise_reproject_history(const KArgs& a, bool sky, float4 reprojection_texel, const float4* prev_colors, const float4* prev_moments) {
    ReprojectHistory h; h.have = false; h.pc = f4z(); h.pm = f4z();
    h.sky = sky;
    if (h.sky) return h;
    const Reprojection rp = reprojection_read(reprojection_texel);
    if (rp.confidence > 0.0f) {
        h.have = true;
        if (TOGETHER) bilinear_reproject2(a, rp, prev_colors, prev_moments, &h.pc, &h.pm);
        else { h.pc = bilinear_reproject(a, rp, prev_colors); h.pm = bilinear_reproject(a, rp, prev_moments); }
    }
    return h;
}
template <bool TOGETHER = false>
ST_D ReprojectHistory denoise_reproject_prefetch(const KArgs& a, U2 pos, const float4* prev_colors, const float4* prev_moments) {
    if (TOGETHER) {
        const float4 snt = tex_read(a.sn, a, pos), rpt = tex_read(a.reprojection, a, pos);   // both before the first branch: one round trip
        return denoise_reproject_history<true>(a, snt.w == 0.0f, rpt, prev_colors, prev_moments);
    }
    const bool sky = tex_read(a.sn, a, pos).w == 0.0f;
    if (sky) { ReprojectHistory h; h.have = false; h.pc = f4z(); h.pm = f4z(); h.sky = true; return h; }
    return denoise_reproject_history<false>(a, false, tex_read(a.reprojection, a, pos), prev_colors, prev_moments);
}
// Returns true where estimate_variance will take its short-history path for this pixel (frame_denoising.rs:118-121; decided
// by the DIRECT signal's history length, so only the DI caller's answer means anything). With KArgs::variance_in_reproject
// the long-history variance (frame_denoising.rs:122-126: second moment - first moment squared, clamped at 0) rides in the
// colour texel's w, where estimate_variance would put it: the plane is rewritten by the a-trous chain before the frame ends.
ST_D bool denoise_reproject_finish(const KArgs& a, U2 pos, float4 sample, const ReprojectHistory& h, float4* colors, float4* moments) {
    if (h.sky) { tex_write(colors, a, pos, sample); return false; }
    const float sample_luma = luma(xyz(sample));
    V3 color, moment;
    if (h.have && sample.w > 0.0f) {
        const float4 pc = h.pc, pm = h.pm;
        const float curr_history = fmin_(pm.x + 1.0f, 16.0f);
        const float alpha = frcp(curr_history);
        color = lerp3(xyz(pc), xyz(sample), alpha);
        moment = v3(curr_history, lerpf(pm.y, sample_luma, alpha), lerpf(pm.z, sample_luma * sample_luma, alpha));
    } else {
        color = xyz(sample);
        moment = v3(1.0f, sample_luma, sample_luma * sample_luma);
    }
    const float variance = a.variance_in_reproject ? fmax_(moment.z - sqr(moment.y), 0.0f) : 0.0f;
    tex_write(colors, a, pos, f4(color, variance));
    tex_write(moments, a, pos, f4(moment, 0.0f));
    return !(moment.x >= 4.0f);
}
// one bit per pixel of an 8x8 tile (bit = lane = pixel_in_tile's numbering), one word per tile
ST_D uint32_t tile_mask_index(const KArgs& a, U2 pos) { return (pos.y >> 3) * ((a.width + 7u) >> 3) + (pos.x >> 3); }
ST_D void denoise_reproject_pixel(const KArgs& a, U2 pos, float4 sample, const float4* prev_colors, const float4* prev_moments, float4* colors, float4* moments) {
    denoise_reproject_finish(a, pos, sample, denoise_reproject_prefetch(a, pos, prev_colors, prev_moments), colors, moments);
}

// gi_resolving.rs:3-67 for one pixel. `res` is what out_reservoirs (gi_res[0]) holds for this pixel when the pass starts.
// Returns the diffuse sample texel (for a fused reprojection stage).
// `reprojects`: the caller runs the GI half of denoise-reproject on the returned texel (the lean frame then skips its store)
ST_D float4 gi_resolve_pixel(const KArgs& a, U2 pos, uint32_t idx, const Hit& hit, const GiReservoir& res, uint32_t source, bool reprojects = false) {
    const uint32_t n = a.width * a.height;
    float confidence; V3 radiance;
    if (hit_some(hit)) { confidence = res.confidence; radiance = res.w * gi_cosine(res.s, hit) * res.s.radiance; }
    else { confidence = 1.0f; radiance = v3s(0.0f); }
    const float diff_brdf = fdivc(1.0f - hit.g.metallic, kPi);
    const V3 spec_brdf = gi_spec_brdf(res.s, hit);
    const float4 diff = f4(radiance * diff_brdf, confidence);
    if (!(reprojects && (a.lean & kLeanSamples))) tex_write(a.gi_diff_samples, a, pos, diff);
    tex_write(a.gi_spec_samples, a, pos, f4(radiance * spec_brdf, confidence));
    const float4* in = source == 0u ? a.gi_res[1] : a.gi_res[2];
    if (!a.gi_skip_history_copy) gi_write_own(a.gi_res[0], idx, gi_read_own(in, idx, true, true), true, true);  // the frame's source reservoir becomes next frame's history
    return diff;
}

// ---- di_sampling.rs:3-94 and di_temporal_resampling.rs:3-112 for one pixel (stand-alone kernels in k_di.hip; primary
// visibility can run both for its own pixel, k_trace.hip)
// the pixel's initial reservoir (RIS over the lights + one shadow ray); `hit` is a surface hit
template <class SE>
ST_D DiReservoir di_sampling_pixel(const KArgs& a, uint32_t seed, U2 pos, const Hit& hit, SE* stack) {
    uint32_t used_ = 0u;
    WhiteNoise wn = white_noise(seed, pos);
    EphemeralResult res = ephemeral_build(a, wn, hit);
    DiReservoir out = di_empty();
    if (res.m > 0.0f) {
        const float4 bn = blue_noise_read(a, pos);
        const Ray ray = light_ray_bnoise(light_get(a, res.light_id), v2(bn.x, bn.y), hit.point);
        const bool occluded = trace_any(a, ray, stack, &used_);
        count_rays(a, used_);
        if (occluded) res.w = 0.0f;
        out.s.light_id = res.light_id; out.s.light_point = ray.origin; out.s.is_occluded = occluded;
        out.m = 1.0f; out.w = res.w;
    }
    return out;
}

// `lhs` is what di_res[1] holds for this pixel (the sampling pass's reservoir), `lhs_hit` a surface hit
ST_D void di_temporal_pixel(const KArgs& a, uint32_t seed, U2 lhs_pos, const Hit& lhs_hit, DiReservoir lhs, float4 reprojection_texel) {
    const uint32_t n = a.width * a.height;
    const uint32_t lhs_idx = screen_to_idx(a, lhs_pos);
    WhiteNoise wn = white_noise(seed, lhs_pos);
    if (lhs.m != 0.0f) lhs.s.pdf = di_pdf_ex(lhs.s, light_get(a, lhs.s.light_id), lhs_hit);
    DiReservoir rhs = di_empty();
    Hit rhs_hit = hit_zero();
    bool rhs_killed = false;
    const Reprojection rp = reprojection_read(reprojection_texel);
    if (rp.confidence > 0.0f) {
        const U2 rhs_pos = reprojection_prev_round(rp);
        rhs = di_read(a.di_res[0], screen_to_idx(a, rhs_pos), n);
        rhs.m = fmin_(rhs.m, 64.0f);
        if (rhs.m != 0.0f) {
            const GpuLight rhs_light = light_get(a, rhs.s.light_id);
            const uint32_t slot = f2b(rhs_light.d3.x);
            if (slot == 0xcafebabeu) { rhs.w = 0.0f; rhs_killed = true; }
            else if (slot > 0u) rhs.s.light_id = slot - 1u;
            rhs_hit = pixel_hit(a, a.prev_cam, a.pg0, a.pg1, rhs_pos);
        }
    }
    Mis mis;
    mis.lhs_rhs_pdf = ((lhs.m > 0.0f) & hit_some(rhs_hit)) ? di_pdf_ex(lhs.s, light_get_prev(a, lhs.s.light_id), rhs_hit) : 0.0f;
    mis.rhs_lhs_pdf = ((rhs.m > 0.0f) & !rhs_killed) ? di_pdf_ex(rhs.s, light_get(a, rhs.s.light_id), lhs_hit) : 0.0f;
    mis.lhs_m = lhs.m; mis.rhs_m = rhs.m; mis.rhs_jacobian = 1.0f; mis.lhs_lhs_pdf = lhs.s.pdf; mis.rhs_rhs_pdf = rhs.s.pdf;
    const MisResult mr = mis_eval(mis);
    DiReservoir main_ = di_empty();
    float main_pdf = 0.0f;
    if (res_update(main_, wn, lhs.s, mr.lhs_mis * mr.lhs_pdf * lhs.w)) main_pdf = mr.lhs_pdf;
    if (res_update(main_, wn, rhs.s, mr.rhs_mis * mr.rhs_pdf * rhs.w)) main_pdf = mr.rhs_pdf;
    main_.m = lhs.m + mr.m;
    main_.s.pdf = main_pdf;
    main_.s.confidence = rhs_killed ? 0.0f : 1.0f;
    res_norm(main_, main_pdf, 1.0f, 1.0f);
    di_write(a.di_res[1], lhs_idx, main_);
}

// frame_composition.rs:18-82 for one pixel; di_diff / gi_diff are this pixel's texels of the (denoised or raw) diffuse planes.
ST_D float4 compose_pixel(const KArgs& a, U2 pos, uint32_t camera_mode, float4 di_diff, float4 gi_diff) {
    V3 color;
    switch (camera_mode) {
        case 0: {
            const GBuffer g = gbuffer_unpack(a, tex_read(a.g0, a, pos), tex_read(a.g1, a, pos));
            const V3 dd = xyz(di_diff), ds = xyz(tex_read(a.di_spec_samples, a, pos));
            const V3 gd = xyz(gi_diff), gs = xyz(tex_read(a.gi_spec_samples, a, pos));
            color = g.depth != 0.0f ? g.emissive + (dd + gd) * xyz(g.base_color) + ds + gs : dd;
            break;
        }
        case 1: color = xyz(di_diff); break;
        case 2: color = xyz(tex_read(a.di_spec_samples, a, pos)); break;
        case 3: color = xyz(gi_diff); break;
        case 4: color = xyz(tex_read(a.gi_spec_samples, a, pos)); break;
        case 5: color = xyz(tex_read(a.ref_colors, a, pos)); break;
        case 6: { const float4 c = tex_read(a.ref_colors, a, pos); color = xyz(c) / c.w; break; }
        default: color = v3s(0.0f);
    }
    return f4(color, 1.0f);
}

// what the render target's format does to the composed colour (camera.rs:170-175 viewport.format; StOutputFormat)
ST_D void store_output(void* out, uint32_t at, float4 c, uint32_t format) {
    if (format == 0u) static_cast<float4*>(out)[at] = c;
    else if (format == 1u) static_cast<uint2*>(out)[at] = make_uint2(f16_bits(c.x) | (f16_bits(c.y) << 16), f16_bits(c.z) | (f16_bits(c.w) << 16));
    else {
        const uint32_t r = srgb8_encode(c.x), g = srgb8_encode(c.y), b = srgb8_encode(c.z);
        static_cast<uint32_t*>(out)[at] = format == 2u ? (r | (g << 8) | (b << 16) | 0xff000000u) : (b | (g << 8) | (r << 16) | 0xff000000u);
    }
}

}  // namespace st
