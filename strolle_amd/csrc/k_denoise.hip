// k_denoise.hip — SVGF: temporal reprojection (x2: DI, GI), variance estimation, five à-trous wavelet passes.
// Behavioural contract: strolle-shaders/src/frame_denoising.rs; plane ping-pong from
// strolle/src/camera_controller/passes/frame_denoising.rs:87-110.
#include "k_common.h"

namespace st {

ST_D float denoise_sample_weight(float center_luma, const Surface& cs, float sample_luma, const Surface& ss, float luma_sigma, float depth_sigma) {
    const float luma_weight = fabsf(sqrtf(center_luma) - sqrtf(sample_luma)) * luma_sigma;
    const float leeway = cs.depth * depth_sigma;
    const float diff = fabsf(ss.depth - cs.depth);
    const float depth_weight = diff >= leeway ? 0.0f : 1.0f - diff / leeway;
    const float normal_weight = pow64_(fmax_(dot(ss.normal, cs.normal), 0.0f));
    return exp_(-luma_weight) * depth_weight * normal_weight;
}

// ---------------------------------------------------------------- frame_denoising.rs:3-78
__global__ __launch_bounds__(kBlockThreads) void k_denoise_reproject(const KArgs a, const float4* prev_colors, const float4* prev_moments,
                                                                      const float4* samples, float4* colors, float4* moments) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    denoise_reproject_pixel(a, pos, tex_read(samples, a, pos), prev_colors, prev_moments, colors, moments);
}
void launch_denoise_reproject(const KArgs& a, const float4* prev_colors, const float4* prev_moments, const float4* samples, float4* colors,
                              float4* moments, hipStream_t s) {
    ST_LAUNCH(k_denoise_reproject, false, s, a, prev_colors, prev_moments, samples, colors, moments);
}

// ---------------------------------------------------------------- frame_denoising.rs:80-217
__global__ __launch_bounds__(kBlockThreads) void k_denoise_variance(const KArgs a) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Surface cs = surface_decoded(tex_read(a.sn, a, pos));
    const float4 cdi = tex_read(a.di_diff_curr_colors, a, pos), cdi_m = tex_read(a.di_diff_moments, a, pos);
    const float4 cgi = tex_read(a.gi_diff_curr_colors, a, pos), cgi_m = tex_read(a.gi_diff_moments, a, pos);
    if (cs.depth == 0.0f) { tex_write(a.di_diff_stash, a, pos, cdi); tex_write(a.gi_diff_stash, a, pos, cgi); return; }
    const float cdi_luma = luma(xyz(cdi)), cgi_luma = luma(xyz(cgi));
    float di_var, gi_var;
    if (cdi_m.x >= 4.0f) {
        di_var = cdi_m.z - sqr(cdi_m.y);
        gi_var = cgi_m.z - sqr(cgi_m.y);
    } else {
        V3 sum_di = v3s(0.0f), sum_gi = v3s(0.0f);
        int ox = -2, oy = -2;
        for (;;) {  // the reference's 29-tap window (frame_denoising.rs:128,180-189), kept as is
            const I2 sp = i2((int32_t)pos.x + ox, (int32_t)pos.y + oy);
            if (contains_i(a, sp)) {
                const U2 up = u2((uint32_t)sp.x, (uint32_t)sp.y);
                const Surface ss = surface_decoded(tex_read(a.sn, a, up));
                if (ss.depth != 0.0f) {
                    const float l = luma(xyz(tex_read(a.di_diff_curr_colors, a, up)));
                    const float w = denoise_sample_weight(cdi_luma, cs, l, ss, 1.0f, 0.2f);
                    sum_di = sum_di + v3(l, l * l, 1.0f) * v3s(w);
                    const float lg = luma(xyz(tex_read(a.gi_diff_curr_colors, a, up)));
                    const float wg = denoise_sample_weight(cgi_luma, cs, lg, ss, 1.0f, 0.2f);
                    sum_gi = sum_gi + v3(lg, lg * lg, 1.0f) * v3s(wg);
                }
            }
            ox += 1;
            if (ox == 3) { ox = -3; oy += 1; if (oy == 3) break; }
        }
        { const float m1 = sum_di.x / sum_di.z, m2 = sum_di.y / sum_di.z; di_var = fabsf(m2 - m1 * m1) * 4.0f; }
        { const float m1 = sum_gi.x / sum_gi.z, m2 = sum_gi.y / sum_gi.z; gi_var = fabsf(m2 - m1 * m1) * 4.0f; }
    }
    di_var = fmax_(di_var, 0.0f);
    gi_var = fmax_(gi_var, 0.0f);
    tex_write(a.di_diff_stash, a, pos, f4(xyz(cdi), di_var));
    tex_write(a.gi_diff_stash, a, pos, f4(xyz(cgi), gi_var));
}
void launch_denoise_variance(const KArgs& a, hipStream_t s) { ST_LAUNCH(k_denoise_variance, false, s, a); }

// ---------------------------------------------------------------- frame_denoising.rs:219-361
// COMPOSE: the last wavelet pass also runs frame composition for its pixel (frame_composition.rs) — the composed frame
// needs only this pixel's denoised colours, which are in registers here.
template <bool COMPOSE>
__global__ __launch_bounds__(kBlockThreads) void k_denoise_wavelet(const KArgs a, uint32_t stride, float strength, const float4* di_in, float4* di_out,
                                                                    const float4* gi_in, float4* gi_out, uint32_t camera_mode, float4* frame_out) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const Surface cs = surface_decoded(tex_read(a.sn, a, pos));
    const float4 cdi = tex_read(di_in, a, pos);
    if (cs.depth == 0.0f) {
        tex_write(di_out, a, pos, cdi);
        // composition reads gi_diff_curr_colors for this pixel, which this pass leaves untouched on sky pixels
        if (COMPOSE) frame_out[pos.y * a.width + pos.x] = compose_pixel(a, pos, camera_mode, cdi, tex_read(gi_out, a, pos));
        return;
    }
    const float4 cgi = tex_read(gi_in, a, pos);
    const float cdi_luma = luma(xyz(cdi)), cgi_luma = luma(xyz(cgi));
    const float luma_sigma_di = lerpf(2.5f, 0.5f, sqrtf(cdi.w));
    const float depth_sigma_di = 0.33f / strength;
    const float luma_sigma_gi = lerpf(1.0f, 0.0f, sqrtf(cgi.w));
    const float depth_sigma_gi = 0.33f / strength;
    const float4 bn = blue_noise_read(a, pos);
    const I2 jitter = as_i2((v2(bn.z, bn.w) - 0.5f) * ((float)stride - 1.0f) * 0.5f);
    float sum_di_w = 1.0f; V3 sum_di_c = xyz(cdi); float sum_di_v = cdi.w;
    float sum_gi_w = 1.0f; V3 sum_gi_c = xyz(cgi); float sum_gi_v = cgi.w;
#pragma unroll
    for (int oy = -1; oy <= 1; oy++) {
#pragma unroll
        for (int ox = -1; ox <= 1; ox++) {
            if (ox == 0 && oy == 0) continue;
            const I2 sp = i2((int32_t)pos.x + jitter.x + ox * (int32_t)stride, (int32_t)pos.y + jitter.y + oy * (int32_t)stride);
            if (!contains_i(a, sp)) continue;
            const U2 up = u2((uint32_t)sp.x, (uint32_t)sp.y);
            const Surface ss = surface_decoded(tex_read(a.sn, a, up));
            if (ss.depth == 0.0f) continue;
            const float4 sdi = tex_read(di_in, a, up);
            const float w = denoise_sample_weight(cdi_luma, cs, luma(xyz(sdi)), ss, luma_sigma_di, depth_sigma_di);
            if (w > 0.0f) { sum_di_w += w; sum_di_c = sum_di_c + w * xyz(sdi); sum_di_v += sqr(w) * sdi.w; }
            const float4 sgi = tex_read(gi_in, a, up);
            const float wg = denoise_sample_weight(cgi_luma, cs, luma(xyz(sgi)), ss, luma_sigma_gi, depth_sigma_gi);
            if (wg > 0.0f) { sum_gi_w += wg; sum_gi_c = sum_gi_c + wg * xyz(sgi); sum_gi_v += sqr(wg) * sgi.w; }
        }
    }
    const float4 odi = f4(sum_di_c / sum_di_w, sum_di_v / (sum_di_w * sum_di_w));
    const float4 ogi = f4(sum_gi_c / sum_gi_w, sum_gi_v / (sum_gi_w * sum_gi_w));
    tex_write(di_out, a, pos, odi);
    tex_write(gi_out, a, pos, ogi);
    if (COMPOSE) frame_out[pos.y * a.width + pos.x] = compose_pixel(a, pos, camera_mode, odi, ogi);
}
void launch_denoise_wavelet(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                            float4* gi_out, hipStream_t s) {
    ST_LAUNCH(k_denoise_wavelet<false>, false, s, a, stride, strength, di_in, di_out, gi_in, gi_out, 0u, (float4*)nullptr);
}
void launch_denoise_wavelet_compose(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                                    float4* gi_out, uint32_t camera_mode, float4* frame_out, hipStream_t s) {
    ST_LAUNCH(k_denoise_wavelet<true>, false, s, a, stride, strength, di_in, di_out, gi_in, gi_out, camera_mode, frame_out);
}

}  // namespace st
