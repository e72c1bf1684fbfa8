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
ST_D ReprojectHistory denoise_reproject_prefetch(const KArgs& a, U2 pos, const float4* prev_colors, const float4* prev_moments) {
    ReprojectHistory h; h.have = false; h.pc = f4z(); h.pm = f4z();
    h.sky = tex_read(a.sn, a, pos).w == 0.0f;
    if (h.sky) return h;
    const Reprojection rp = reprojection_read(tex_read(a.reprojection, a, pos));
    if (rp.confidence > 0.0f) { h.have = true; h.pc = bilinear_reproject(a, rp, prev_colors); h.pm = bilinear_reproject(a, rp, prev_moments); }
    return h;
}
ST_D void denoise_reproject_finish(const KArgs& a, U2 pos, float4 sample, const ReprojectHistory& h, float4* colors, float4* moments) {
    if (h.sky) { tex_write(colors, a, pos, sample); return; }
    const float sample_luma = luma(xyz(sample));
    V3 color, moment;
    if (h.have && sample.w > 0.0f) {
        const float4 pc = h.pc, pm = h.pm;
        const float curr_history = fmin_(pm.x + 1.0f, 16.0f);
        const float alpha = 1.0f / curr_history;
        color = lerp3(xyz(pc), xyz(sample), alpha);
        moment = v3(curr_history, lerpf(pm.y, sample_luma, alpha), lerpf(pm.z, sample_luma * sample_luma, alpha));
    } else {
        color = xyz(sample);
        moment = v3(1.0f, sample_luma, sample_luma * sample_luma);
    }
    tex_write(colors, a, pos, f4(color, 0.0f));
    tex_write(moments, a, pos, f4(moment, 0.0f));
}
ST_D void denoise_reproject_pixel(const KArgs& a, U2 pos, float4 sample, const float4* prev_colors, const float4* prev_moments, float4* colors, float4* moments) {
    denoise_reproject_finish(a, pos, sample, denoise_reproject_prefetch(a, pos, prev_colors, prev_moments), colors, moments);
}

// gi_resolving.rs:3-67 for one pixel. `res` is what out_reservoirs (gi_res[0]) holds for this pixel when the pass starts.
// Returns the diffuse sample texel (for a fused reprojection stage).
ST_D float4 gi_resolve_pixel(const KArgs& a, U2 pos, uint32_t idx, const Hit& hit, const GiReservoir& res, uint32_t source) {
    const uint32_t n = a.width * a.height;
    float confidence; V3 radiance;
    if (hit_some(hit)) { confidence = res.confidence; radiance = res.w * gi_cosine(res.s, hit) * res.s.radiance; }
    else { confidence = 1.0f; radiance = v3s(0.0f); }
    const float diff_brdf = (1.0f - hit.g.metallic) / kPi;
    const V3 spec_brdf = gi_spec_brdf(res.s, hit);
    const float4 diff = f4(radiance * diff_brdf, confidence);
    tex_write(a.gi_diff_samples, a, pos, diff);
    tex_write(a.gi_spec_samples, a, pos, f4(radiance * spec_brdf, confidence));
    const float4* in = source == 0u ? a.gi_res[1] : a.gi_res[2];
    gi_write(a.gi_res[0], idx, gi_read(in, idx, n));
    return diff;
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

}  // namespace st
