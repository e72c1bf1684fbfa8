// k_denoise.hip — SVGF: temporal reprojection (x2: DI, GI), variance estimation, five à-trous wavelet passes.
// Behavioural contract: strolle-shaders/src/frame_denoising.rs; plane ping-pong from
// strolle/src/camera_controller/passes/frame_denoising.rs:87-110.
#include "k_common.h"

namespace st {
namespace ST_KNS {

// The SVGF sample weight (frame_denoising.rs:363-398) is exp(-|sqrt(luma_c) - sqrt(luma_s)| * luma_sigma) * depth_weight *
// normal_weight, multiplied in that order; the depth and normal factors do not depend on the signal, so the loops below
// evaluate them once per tap and the exponential for both signals at once.

// Packed pairs (direct, indirect): v_pk_mul_f32 / v_pk_add_f32 carry both signals through one issue slot each.
typedef float f2 __attribute__((ext_vector_type(2)));
typedef int i2v __attribute__((ext_vector_type(2)));
ST_D f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
ST_D f2 splat2(float x) { return mk2(x, x); }
ST_D f2 sqrt2(f2 x) { return mk2(fsqrt(x.x), fsqrt(x.y)); }
// x / d where 1 / d may be precomputed: the exact build divides, the fast build multiplies by the hoisted reciprocal
ST_D float div_by(float x, float d, float inv_d) {
#if ST_FAST_DEVICE
    return x * inv_d;
#else
    return x / d;
#endif
}
// frame_denoising.rs:343-358: colour sums / weight sum, variance sum / weight sum squared
ST_D float4 wavelet_resolve(float r, float g, float b, float v, float w, float ww) {
#if ST_FAST_DEVICE
    const float iw = frcp(w);
    return make_float4(r * iw, g * iw, b * iw, v * (iw * iw));
#else
    return make_float4(r / w, g / w, b / w, v / ww);
#endif
}

// exp_() of both halves. Inside |x| < 87 none of exp_'s range branches fire and scale2() is a single multiplication by
// 2^n with -126 <= n <= 126, so the straight-line packed evaluation is exp_() operation for operation; anything else
// (NaN, overflow, the denormal tail) takes the scalar routine.
ST_D f2 exp_pair(f2 x) {
#if ST_FAST_DEVICE
    x = x * 1.44269504088896341f;
    return mk2(__builtin_amdgcn_exp2f(x.x), __builtin_amdgcn_exp2f(x.y));
#endif
    if (!(fabsf(x.x) < 87.0f && fabsf(x.y) < 87.0f)) return mk2(exp_(x.x), exp_(x.y));
    const f2 z = __builtin_elementwise_floor(1.44269504088896341f * x + 0.5f);
    x = x - z * 0.693359375f;
    x = x - z * -2.12194440e-4f;
    const i2v n = __builtin_convertvector(z, i2v);
    const f2 zz = x * x;
    const f2 p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * zz + x + 1.0f;
    return p * mk2(b2f((uint32_t)(n.x + 127) << 23), b2f((uint32_t)(n.y + 127) << 23));
}

// ---------------------------------------------------------------- frame_denoising.rs:3-78
__global__ ST_KERNEL_BOUNDS void k_denoise_reproject(const KArgs a, const float4* prev_colors, const float4* prev_moments,
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
__global__ ST_KERNEL_BOUNDS void k_denoise_variance(const KArgs a) {
    // Window of the short-history estimate, staged per block when any of its pixels needs it: ox in [-3, 2], oy in
    // [-2, 2] around 32x8 pixels = 38 x 12 texels of (surface, direct colour, indirect colour). Only ~15 % of the waves
    // take the slow path on Cornell, but with 58 dependent loads each they set the kernel's duration; from LDS the same
    // 29 taps cost one staging round trip. Texels outside the viewport are staged as depth 0, which the loop skips just
    // as it skips out-of-bounds and sky taps (frame_denoising.rs:135-140).
    constexpr int RW = 38, RH = 12, PITCH = 40;
    __shared__ float4 s_sn[PITCH * RH];
    __shared__ float4 s_di[PITCH * RH];
    __shared__ float4 s_gi[PITCH * RH];
    const uint32_t tiles_x = (a.width + 7u) >> 3;
    const uint32_t ty0 = a.row0 >> 3, ty1 = (a.row1 + 7u) >> 3;
    TileCoord tc = tile_for_thread(tiles_x, ty1 - ty0, a.tile_map);
    const uint32_t wave = threadIdx.x >> 6;
    tc.y += ty0;
    const U2 pos = pixel_in_tile(tc);
    const bool mine = tc.valid && owns_pixel(a, pos);
    const uint32_t center = pos.y * a.width + pos.x;
    float4 csn = f4z(), cdi = f4z(), cdi_m = f4z(), cgi = f4z(), cgi_m = f4z();
    if (mine) { csn = a.sn[center]; cdi = a.di_diff_curr_colors[center]; cdi_m = a.di_diff_moments[center]; cgi = a.gi_diff_curr_colors[center]; cgi_m = a.gi_diff_moments[center]; }
    const bool slow = mine && csn.w != 0.0f && !(cdi_m.x >= 4.0f);
    if (__syncthreads_or(slow ? 1 : 0)) {
        const int32_t bx0 = (int32_t)((tc.x - wave) * 8u) - 3, by0 = (int32_t)(tc.y * 8u) - 2;
        for (int i = (int)threadIdx.x; i < RW * RH; i += kBlockThreads) {
            const int ry = i / RW, rx = i - ry * RW;
            const int32_t gx = bx0 + rx, gy = by0 + ry;
            const int li = ry * PITCH + rx;
            if (gx >= 0 && gy >= 0 && gx < (int32_t)a.width && gy < (int32_t)a.height) {
                const uint32_t at = (uint32_t)gy * a.width + (uint32_t)gx;
                s_sn[li] = a.sn[at]; s_di[li] = a.di_diff_curr_colors[at]; s_gi[li] = a.gi_diff_curr_colors[at];
            } else {
                s_sn[li] = f4z();
            }
        }
        __syncthreads();
    }
    if (!mine) return;
    if (csn.w == 0.0f) { a.di_diff_stash[center] = cdi; a.gi_diff_stash[center] = cgi; return; }  // sky
    const float cdi_luma = luma(xyz(cdi)), cgi_luma = luma(xyz(cgi));
    float di_var, gi_var;
    if (!slow) {
        di_var = cdi_m.z - sqr(cdi_m.y);
        gi_var = cgi_m.z - sqr(cgi_m.y);
    } else {
        // Short history: spatial estimate over the reference's 29-tap window (frame_denoising.rs:128,180-189: the walk
        // starts at (-2,-2) and every later row starts at -3), kept as is; the two signals share packed-f32 arithmetic
        // as in the wavelet pass.
        const V3 cn = v3(csn.x, csn.y, csn.z);
        const f2 c_sqrt_luma = mk2(fsqrt(cdi_luma), fsqrt(cgi_luma));
        const float leeway = csn.w * 0.2f, inv_leeway = frcp(leeway);
        const int lc = ((int)(pos.y & 7u) + 2) * PITCH + (int)(wave * 8u + (pos.x & 7u)) + 3;
        f2 sum_l = splat2(0.0f), sum_ll = splat2(0.0f), sum_1 = splat2(0.0f);
        for (int oy = -2; oy <= 2; oy++) {
#pragma unroll
            for (int ox = -3; ox <= 2; ox++) {
                if (ox == -3 && oy == -2) continue;
                const int lt = lc + oy * PITCH + ox;
                const float4 ssn = s_sn[lt];
                if (ssn.w == 0.0f) continue;
                const float4 sdi = s_di[lt], sgi = s_gi[lt];
                const f2 l = (mk2(sdi.x, sgi.x) * 0.2126f + mk2(sdi.y, sgi.y) * 0.7152f) + mk2(sdi.z, sgi.z) * 0.0722f;
                const f2 d = c_sqrt_luma - sqrt2(l);
                const float diff = fabsf(ssn.w - csn.w);
                const float depth_weight = diff >= leeway ? 0.0f : 1.0f - div_by(diff, leeway, inv_leeway);
                const float normal_weight = pow64_(fmax_(dot(v3(ssn.x, ssn.y, ssn.z), cn), 0.0f));
                const f2 w = exp_pair(-mk2(fabsf(d.x), fabsf(d.y))) * depth_weight * normal_weight;  // luma sigma 1: |d| * 1 == |d|
                sum_l = sum_l + l * w; sum_ll = sum_ll + (l * l) * w; sum_1 = sum_1 + w;
            }
        }
        { const float m1 = fdiv(sum_l.x, sum_1.x), m2 = fdiv(sum_ll.x, sum_1.x); di_var = fabsf(m2 - m1 * m1) * 4.0f; }
        { const float m1 = fdiv(sum_l.y, sum_1.y), m2 = fdiv(sum_ll.y, sum_1.y); gi_var = fabsf(m2 - m1 * m1) * 4.0f; }
    }
    di_var = fmax_(di_var, 0.0f);
    gi_var = fmax_(gi_var, 0.0f);
    a.di_diff_stash[center] = f4(xyz(cdi), di_var);
    a.gi_diff_stash[center] = f4(xyz(cgi), gi_var);
    a.sl[0][center] = make_float2(fsqrt(cdi_luma), fsqrt(cgi_luma));  // for the first wavelet pass's taps
}
void launch_denoise_variance(const KArgs& a, hipStream_t s) { ST_LAUNCH(k_denoise_variance, false, s, a); }

// ---------------------------------------------------------------- frame_denoising.rs:219-361
// The pass is VALU-issue-bound (SQ counters: ~100 % VALU busy, 1245 VALU/wave before this layout), so the direct and
// indirect signals are carried as the two halves of 64-bit packed-f32 operations (v_pk_mul_f32 / v_pk_add_f32): the
// same IEEE operations in the same order per half, half the issue slots. The weight's shared factors (depth, normal)
// are evaluated once per tap; a tap whose shared factor is exactly zero is dropped before its colours are loaded —
// its weight would be 0 or NaN and `w > 0` (frame_denoising.rs:318,340) rejects both.
// COMPOSE: the last wavelet pass also runs frame composition for its pixel (frame_composition.rs) — the composed frame
// needs only this pixel's denoised colours, which are in registers here.
template <bool COMPOSE>
__global__ ST_KERNEL_BOUNDS void k_denoise_wavelet(const KArgs a, uint32_t stride, float strength, const float4* di_in, float4* di_out,
                                                                    const float4* gi_in, float4* gi_out, const float2* sl_in, float2* sl_out,
                                                                    uint32_t camera_mode, float4* frame_out) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t center = pos.y * a.width + pos.x;
    const float4 csn = a.sn[center];
    const float4 cdi = di_in[center];
    if (csn.w == 0.0f) {  // sky
        di_out[center] = cdi;
        // composition reads gi_diff_curr_colors for this pixel, which this pass leaves untouched on sky pixels
        if (COMPOSE) frame_out[center] = compose_pixel(a, pos, camera_mode, cdi, gi_out[center]);
        return;
    }
    const float4 cgi = gi_in[center];
    const V3 cn = v3(csn.x, csn.y, csn.z);
    // sl_in == null: this pass's input has no sqrt-luma plane (the generic kernel serves strides 8 and 16, whose taps
    // are texture-address-bound — a fourth load per tap costs more there than the two square roots it replaces)
    f2 c_sqrt_luma;
    if (sl_in) { const float2 csl = sl_in[center]; c_sqrt_luma = mk2(csl.x, csl.y); }
    else c_sqrt_luma = mk2(fsqrt(luma(xyz(cdi))), fsqrt(luma(xyz(cgi))));
    const f2 luma_sigma = mk2(lerpf(2.5f, 0.5f, fsqrt(cdi.w)), lerpf(1.0f, 0.0f, fsqrt(cgi.w)));
    const float leeway = csn.w * (0.33f / strength), inv_leeway = frcp(leeway);  // depth sigma is the same for both signals
    I2 jitter = i2(0, 0);
    if (stride != 1u) {  // at stride 1 the jitter is (bn - 0.5) * 0 * 0.5 == 0
        const float4 bn = blue_noise_read(a, pos);
        jitter = as_i2((v2(bn.z, bn.w) - 0.5f) * ((float)stride - 1.0f) * 0.5f);
    }
    f2 sum_w = splat2(1.0f), sum_r = mk2(cdi.x, cgi.x), sum_g = mk2(cdi.y, cgi.y), sum_b = mk2(cdi.z, cgi.z), sum_v = mk2(cdi.w, cgi.w);
    // Loads are issued in two unconditional batches (8 surface texels, then the colours of the taps) so that a wave pays
    // two memory round trips instead of sixteen dependent ones; a tap that is out of bounds reads the centre texel and
    // is masked out afterwards.
    uint32_t at[8];
    float4 ssn[8];
    bool live[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int k = t < 4 ? t : t + 1, ox = k % 3 - 1, oy = k / 3 - 1;
        const I2 sp = i2((int32_t)pos.x + jitter.x + ox * (int32_t)stride, (int32_t)pos.y + jitter.y + oy * (int32_t)stride);
        live[t] = contains_i(a, sp);
        at[t] = live[t] ? (uint32_t)sp.y * a.width + (uint32_t)sp.x : center;
        ssn[t] = a.sn[at[t]];
    }
    float depth_w[8], normal_w[8];
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float diff = fabsf(ssn[t].w - csn.w);
        depth_w[t] = diff >= leeway ? 0.0f : 1.0f - div_by(diff, leeway, inv_leeway);
        normal_w[t] = pow64_(fmax_(dot(v3(ssn[t].x, ssn[t].y, ssn[t].z), cn), 0.0f));
        live[t] = live[t] && ssn[t].w != 0.0f && !(depth_w[t] == 0.0f || normal_w[t] == 0.0f);
    }
#pragma unroll
    for (int half = 0; half < 2; half++) {
        float4 sdi[4], sgi[4]; float2 ssl[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { const uint32_t i = at[half * 4 + u]; sdi[u] = di_in[i]; sgi[u] = gi_in[i]; if (sl_in) ssl[u] = sl_in[i]; }
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int t = half * 4 + u;
            if (!live[t]) continue;
            const f2 r = mk2(sdi[u].x, sgi[u].x), g = mk2(sdi[u].y, sgi[u].y), b = mk2(sdi[u].z, sgi[u].z), v = mk2(sdi[u].w, sgi[u].w);
            f2 tap_sqrt_luma;
            if (sl_in) tap_sqrt_luma = mk2(ssl[u].x, ssl[u].y);
            else { const f2 l = (r * 0.2126f + g * 0.7152f) + b * 0.0722f; tap_sqrt_luma = sqrt2(l); }
            const f2 d = c_sqrt_luma - tap_sqrt_luma;
            const f2 luma_weight = mk2(fabsf(d.x), fabsf(d.y)) * luma_sigma;
            const f2 w = exp_pair(-luma_weight) * depth_w[t] * normal_w[t];
            if (w.x > 0.0f && w.y > 0.0f) {
                sum_w = sum_w + w; sum_r = sum_r + w * r; sum_g = sum_g + w * g; sum_b = sum_b + w * b; sum_v = sum_v + (w * w) * v;
            } else {
                if (w.x > 0.0f) { sum_w.x += w.x; sum_r.x += w.x * r.x; sum_g.x += w.x * g.x; sum_b.x += w.x * b.x; sum_v.x += (w.x * w.x) * v.x; }
                if (w.y > 0.0f) { sum_w.y += w.y; sum_r.y += w.y * r.y; sum_g.y += w.y * g.y; sum_b.y += w.y * b.y; sum_v.y += (w.y * w.y) * v.y; }
            }
        }
    }
    const f2 ww = sum_w * sum_w;
    const float4 odi = wavelet_resolve(sum_r.x, sum_g.x, sum_b.x, sum_v.x, sum_w.x, ww.x);
    const float4 ogi = wavelet_resolve(sum_r.y, sum_g.y, sum_b.y, sum_v.y, sum_w.y, ww.y);
    di_out[center] = odi;
    gi_out[center] = ogi;
    if (sl_out) sl_out[center] = make_float2(fsqrt(luma(xyz(odi))), fsqrt(luma(xyz(ogi))));
    if (COMPOSE) frame_out[center] = compose_pixel(a, pos, camera_mode, odi, ogi);
}
// ---------------------------------------------------------------- the same pass for strides 1, 2 and 4, staged through LDS
// At these strides the jitter is identically zero — |(bn - 0.5) * (stride - 1) * 0.5| <= 0.75 truncates to 0
// (frame_denoising.rs:262-266) — so the taps of a block's 32x8 pixels fall on a fixed (32+2S)x(8+2S) window that the
// block loads once (1.3-2.5 texels per pixel instead of 9) and then reads from LDS. Texels outside the viewport are
// staged with depth 0: the tap loop already skips depth-0 (sky) samples, which is what `continue` on an out-of-bounds
// tap does in the reference. Row pitch 40 texels: 640 B rows put 8-lane row segments of ds_read_b128 on disjoint banks.
template <int S>
__global__ ST_KERNEL_BOUNDS void k_denoise_wavelet_lds(const KArgs a, float strength, const float4* di_in, float4* di_out,
                                                                        const float4* gi_in, float4* gi_out, const float2* sl_in, float2* sl_out) {
    constexpr int RW = 32 + 2 * S, RH = 8 + 2 * S, PITCH = 40;
    __shared__ float4 s_sn[PITCH * RH];
    __shared__ float4 s_di[PITCH * RH];
    __shared__ float4 s_gi[PITCH * RH];
    __shared__ float2 s_sl[PITCH * RH];
    const uint32_t tiles_x = (a.width + 7u) >> 3;
    const uint32_t ty0 = a.row0 >> 3, ty1 = (a.row1 + 7u) >> 3;
    TileCoord tc = tile_for_thread(tiles_x, ty1 - ty0, a.tile_map);  // tc.y is block-uniform, tc.x = first tile of the block + wave
    const uint32_t wave = threadIdx.x >> 6;
    const int32_t bx0 = (int32_t)((tc.x - wave) * 8u) - S, by0 = (int32_t)((tc.y + ty0) * 8u) - S;
    for (int i = (int)threadIdx.x; i < RW * RH; i += kBlockThreads) {
        const int ry = i / RW, rx = i - ry * RW;
        const int32_t gx = bx0 + rx, gy = by0 + ry;
        const int li = ry * PITCH + rx;
        if (gx >= 0 && gy >= 0 && gx < (int32_t)a.width && gy < (int32_t)a.height) {
            const uint32_t at = (uint32_t)gy * a.width + (uint32_t)gx;
            s_sn[li] = a.sn[at]; s_di[li] = di_in[at]; s_gi[li] = gi_in[at]; s_sl[li] = sl_in[at];
        } else {
            s_sn[li] = f4z();
        }
    }
    __syncthreads();
    if (!tc.valid) return;
    tc.y += ty0;
    const U2 pos = pixel_in_tile(tc);
    if (!owns_pixel(a, pos)) return;
    const uint32_t center = pos.y * a.width + pos.x;
    const int lc = ((int)(pos.y & 7u) + S) * PITCH + (int)(wave * 8u + (pos.x & 7u)) + S;
    const float4 csn = s_sn[lc];
    const float4 cdi = s_di[lc];
    if (csn.w == 0.0f) { di_out[center] = cdi; return; }  // sky
    const float4 cgi = s_gi[lc];
    const V3 cn = v3(csn.x, csn.y, csn.z);
    const float2 csl = s_sl[lc];
    const f2 c_sqrt_luma = mk2(csl.x, csl.y);
    const f2 luma_sigma = mk2(lerpf(2.5f, 0.5f, fsqrt(cdi.w)), lerpf(1.0f, 0.0f, fsqrt(cgi.w)));
    const float leeway = csn.w * (0.33f / strength), inv_leeway = frcp(leeway);
    f2 sum_w = splat2(1.0f), sum_r = mk2(cdi.x, cgi.x), sum_g = mk2(cdi.y, cgi.y), sum_b = mk2(cdi.z, cgi.z), sum_v = mk2(cdi.w, cgi.w);
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int k = t < 4 ? t : t + 1, ox = k % 3 - 1, oy = k / 3 - 1;
        const int lt = lc + oy * S * PITCH + ox * S;
        const float4 ssn = s_sn[lt];
        if (ssn.w == 0.0f) continue;
        const float diff = fabsf(ssn.w - csn.w);
        const float depth_weight = diff >= leeway ? 0.0f : 1.0f - div_by(diff, leeway, inv_leeway);
        const float normal_weight = pow64_(fmax_(dot(v3(ssn.x, ssn.y, ssn.z), cn), 0.0f));
        if (depth_weight == 0.0f || normal_weight == 0.0f) continue;
        const float4 sdi = s_di[lt], sgi = s_gi[lt];
        const float2 ssl = s_sl[lt];
        const f2 r = mk2(sdi.x, sgi.x), g = mk2(sdi.y, sgi.y), b = mk2(sdi.z, sgi.z), v = mk2(sdi.w, sgi.w);
        const f2 d = c_sqrt_luma - mk2(ssl.x, ssl.y);
        const f2 luma_weight = mk2(fabsf(d.x), fabsf(d.y)) * luma_sigma;
        const f2 w = exp_pair(-luma_weight) * depth_weight * normal_weight;
        if (w.x > 0.0f && w.y > 0.0f) {
            sum_w = sum_w + w; sum_r = sum_r + w * r; sum_g = sum_g + w * g; sum_b = sum_b + w * b; sum_v = sum_v + (w * w) * v;
        } else {
            if (w.x > 0.0f) { sum_w.x += w.x; sum_r.x += w.x * r.x; sum_g.x += w.x * g.x; sum_b.x += w.x * b.x; sum_v.x += (w.x * w.x) * v.x; }
            if (w.y > 0.0f) { sum_w.y += w.y; sum_r.y += w.y * r.y; sum_g.y += w.y * g.y; sum_b.y += w.y * b.y; sum_v.y += (w.y * w.y) * v.y; }
        }
    }
    const f2 ww = sum_w * sum_w;
    const float4 odi = wavelet_resolve(sum_r.x, sum_g.x, sum_b.x, sum_v.x, sum_w.x, ww.x);
    const float4 ogi = wavelet_resolve(sum_r.y, sum_g.y, sum_b.y, sum_v.y, sum_w.y, ww.y);
    di_out[center] = odi;
    gi_out[center] = ogi;
    if (sl_out) sl_out[center] = make_float2(fsqrt(luma(xyz(odi))), fsqrt(luma(xyz(ogi))));
}

void launch_denoise_wavelet(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                            float4* gi_out, const float2* sl_in, float2* sl_out, hipStream_t s) {
    if (stride == 1u) ST_LAUNCH(k_denoise_wavelet_lds<1>, false, s, a, strength, di_in, di_out, gi_in, gi_out, sl_in, sl_out);
    else if (stride == 2u) ST_LAUNCH(k_denoise_wavelet_lds<2>, false, s, a, strength, di_in, di_out, gi_in, gi_out, sl_in, sl_out);
    else if (stride == 4u) ST_LAUNCH(k_denoise_wavelet_lds<4>, false, s, a, strength, di_in, di_out, gi_in, gi_out, sl_in, sl_out);
    else ST_LAUNCH(k_denoise_wavelet<false>, false, s, a, stride, strength, di_in, di_out, gi_in, gi_out, sl_in, sl_out, 0u, (float4*)nullptr);
}
void launch_denoise_wavelet_compose(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                                    float4* gi_out, const float2* sl_in, uint32_t camera_mode, float4* frame_out, hipStream_t s) {
    ST_LAUNCH(k_denoise_wavelet<true>, false, s, a, stride, strength, di_in, di_out, gi_in, gi_out, sl_in, (float2*)nullptr, camera_mode, frame_out);
}

// ---------------------------------------------------------------- st_camera_write_buffer support
__global__ ST_KERNEL_BOUNDS void k_refresh_internal_planes(const KArgs a, float4* psn_out) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !contains_u(a, pos)) return;
    const uint32_t i = pos.y * a.width + pos.x;
    const float4 sm = a.sm[i], psm = a.psm[i];
    a.sn[i] = sm.z == 0.0f ? f4z() : f4(normal_decode(v2(sm.x, sm.y)), sm.z);   // what primary visibility writes beside the surface map
    psn_out[i] = psm.z == 0.0f ? f4z() : f4(normal_decode(v2(psm.x, psm.y)), psm.z);
    a.sl[0][i] = make_float2(fsqrt(luma(xyz(a.di_diff_stash[i]))), fsqrt(luma(xyz(a.gi_diff_stash[i]))));
    a.sl[1][i] = make_float2(fsqrt(luma(xyz(a.di_diff_prev_colors[i]))), fsqrt(luma(xyz(a.gi_diff_prev_colors[i]))));
}
void launch_refresh_internal_planes(const KArgs& a_in, hipStream_t s) {
    KArgs a = a_in; a.row0 = 0; a.row1 = a.height;
    ST_LAUNCH(k_refresh_internal_planes, false, s, a, const_cast<float4*>(a.psn));
}

}  // namespace ST_KNS
}  // namespace st
