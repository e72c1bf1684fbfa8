// k_denoise.hip — SVGF: temporal reprojection (x2: DI, GI), variance estimation, five à-trous wavelet passes.
// Behavioural contract: strolle-shaders/src/frame_denoising.rs; plane ping-pong from
// strolle/src/camera_controller/passes/frame_denoising.rs:87-110.
#include "k_common.h"

namespace st {
namespace ST_KNS {

// The SVGF sample weight (frame_denoising.rs:363-398) is exp(-|sqrt(luma_c) - sqrt(luma_s)| * luma_sigma) * depth_weight *
// normal_weight, multiplied in that order; the depth and normal factors do not depend on the signal, so the loops below
// evaluate them once per tap and the exponential for both signals at once.

// Pairs (direct, indirect) that share a tap's arithmetic. Plain structs, scalar VALU operations: measured on MI355X
// (tools/issue_probe.hip) a v_pk_fma_f32 takes 5.2 cycles per SIMD against 2.9 for v_fma_f32, so the packed form buys
// 10 % at best and loses it to the register moves that pair the operands up.
struct f2 { float x, y; };
struct i2v { int x, y; };
ST_D f2 mk2(float x, float y) { f2 r; r.x = x; r.y = y; return r; }
ST_D f2 operator+(f2 a, f2 b) { return mk2(a.x + b.x, a.y + b.y); }
ST_D f2 operator-(f2 a, f2 b) { return mk2(a.x - b.x, a.y - b.y); }
ST_D f2 operator*(f2 a, f2 b) { return mk2(a.x * b.x, a.y * b.y); }
ST_D f2 operator*(f2 a, float b) { return mk2(a.x * b, a.y * b); }
ST_D f2 operator*(float a, f2 b) { return mk2(a * b.x, a * b.y); }
ST_D f2 operator+(f2 a, float b) { return mk2(a.x + b, a.y + b); }
ST_D f2 operator-(f2 a) { return mk2(-a.x, -a.y); }
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

// exp_() of both halves
ST_D f2 exp_pair(f2 x) {
#if ST_FAST_DEVICE
    return mk2(__builtin_amdgcn_exp2f(x.x * 1.44269504088896341f), __builtin_amdgcn_exp2f(x.y * 1.44269504088896341f));
#else
    return mk2(exp_(x.x), exp_(x.y));
#endif
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

// Short history (frame_denoising.rs:128,180-189): spatial estimate over the reference's 29-tap window — the walk starts at
// (-2,-2) and every later row starts at -3 —, kept as is, from a window staged in LDS (`lc` = the pixel's texel, row pitch
// `pitch`). A tap needs its two colours only as luma and sqrt(luma): both are evaluated once per staged texel and staged as
// one texel (direct luma, indirect luma, their square roots) — two 16-B LDS reads per tap instead of three, no luma
// arithmetic per tap. The two signals share the arithmetic as in the wavelet pass.
ST_D void variance_short_history(float4 csn, float4 cdi, float4 cgi, const float4* s_sn, const float4* s_l, int lc, int pitch, float* di_var, float* gi_var) {
    const V3 cn = v3(csn.x, csn.y, csn.z);
    const f2 c_sqrt_luma = mk2(fsqrt(luma(xyz(cdi))), fsqrt(luma(xyz(cgi))));
    const float leeway = csn.w * 0.2f, inv_leeway = frcp(leeway);
    f2 sum_l = splat2(0.0f), sum_ll = splat2(0.0f), sum_1 = splat2(0.0f);
    for (int oy = -2; oy <= 2; oy++) {
#pragma unroll
        for (int ox = -3; ox <= 2; ox++) {
            if (ox == -3 && oy == -2) continue;
            const int lt = lc + oy * pitch + ox;
            const float4 ssn = s_sn[lt];
            if (ssn.w == 0.0f) continue;
            const float4 sl = s_l[lt];
            const f2 l = mk2(sl.x, sl.y);
            const f2 d = c_sqrt_luma - mk2(sl.z, sl.w);
            const float diff = fabsf(ssn.w - csn.w);
            const float depth_weight = diff >= leeway ? 0.0f : 1.0f - div_by(diff, leeway, inv_leeway);
            const float normal_weight = pow64_(fmax_(dot(v3(ssn.x, ssn.y, ssn.z), cn), 0.0f));
            const f2 w = exp_pair(-mk2(fabsf(d.x), fabsf(d.y))) * depth_weight * normal_weight;  // luma sigma 1: |d| * 1 == |d|
            sum_l = sum_l + l * w; sum_ll = sum_ll + (l * l) * w; sum_1 = sum_1 + w;
        }
    }
    { const float m1 = fdiv(sum_l.x, sum_1.x), m2 = fdiv(sum_ll.x, sum_1.x); *di_var = fabsf(m2 - m1 * m1) * 4.0f; }
    { const float m1 = fdiv(sum_l.y, sum_1.y), m2 = fdiv(sum_ll.y, sum_1.y); *gi_var = fabsf(m2 - m1 * m1) * 4.0f; }
}

// ---------------------------------------------------------------- frame_denoising.rs:80-217
__global__ ST_KERNEL_BOUNDS void k_denoise_variance(const KArgs a, float4* di_out, float4* gi_out) {
    // Window of the short-history estimate, staged per block when any of its pixels needs it: ox in [-3, 2], oy in
    // [-2, 2] around 32x8 pixels = 38 x 12 texels of (surface, direct colour, indirect colour). Only ~15 % of the waves
    // take the slow path on Cornell, but with 58 dependent loads each they set the kernel's duration; from LDS the same
    // 29 taps cost one staging round trip. Texels outside the viewport are staged as depth 0, which the loop skips just
    // as it skips out-of-bounds and sky taps (frame_denoising.rs:135-140).
    constexpr int RW = 38, RH = 12, PITCH = 40;
    __shared__ float4 s_sn[PITCH * RH];
    __shared__ float4 s_l[PITCH * RH];
    const TileCoord tc = resolve_tile(a, false);
    const uint32_t wave = threadIdx.x >> 6;
    const U2 pos = pixel_in_tile(tc);
    const bool mine = tc.valid && owns_pixel(a, pos);
    const uint32_t center = pos.y * a.width + pos.x;
    float4 csn = f4z(), cdi = f4z(), cdi_m = f4z(), cgi = f4z(), cgi_m = f4z();
    bool slow;
    if (a.variance_in_reproject) {
        // The reproject stages have already stored the long-history variance in the colour texels' w (st_passes.h
        // denoise_reproject_finish) and flagged the short-history pixels: this launch only serves those, in place. A block
        // without any leaves after one 8-byte load per wave.
        slow = mine && ((a.tile_mask[tile_mask_index(a, pos)] >> (threadIdx.x & 63u)) & 1ull) != 0ull;
        if (!__syncthreads_or(slow ? 1 : 0)) return;
        if (slow) { csn = a.sn[center]; cdi = a.di_diff_curr_colors[center]; cgi = a.gi_diff_curr_colors[center]; }
    } else {
        if (mine) { csn = a.sn[center]; cdi = a.di_diff_curr_colors[center]; cdi_m = a.di_diff_moments[center]; cgi = a.gi_diff_curr_colors[center]; cgi_m = a.gi_diff_moments[center]; }
        slow = mine && csn.w != 0.0f && !(cdi_m.x >= 4.0f);
    }
    if (a.variance_in_reproject || __syncthreads_or(slow ? 1 : 0)) {
        const int32_t bx0 = (int32_t)((tc.x - wave) * 8u) - 3, by0 = (int32_t)(tc.y * 8u) - 2;
        for (int i = (int)threadIdx.x; i < RW * RH; i += kBlockThreads) {
            const int ry = i / RW, rx = i - ry * RW;
            const int32_t gx = bx0 + rx, gy = by0 + ry;
            const int li = ry * PITCH + rx;
            if (gx >= 0 && gy >= 0 && gx < (int32_t)a.width && gy < (int32_t)a.height) {
                const uint32_t at = (uint32_t)gy * a.width + (uint32_t)gx;
                // luma and sqrt(luma) of both signals: evaluated once per staged texel, not once per tap
                const float4 tdi = a.di_diff_curr_colors[at], tgi = a.gi_diff_curr_colors[at];
                const f2 tl = (mk2(tdi.x, tgi.x) * 0.2126f + mk2(tdi.y, tgi.y) * 0.7152f) + mk2(tdi.z, tgi.z) * 0.0722f;
                const f2 ts = sqrt2(tl);
                s_sn[li] = a.sn[at]; s_l[li] = make_float4(tl.x, tl.y, ts.x, ts.y);
            } else {
                s_sn[li] = f4z();
            }
        }
        __syncthreads();
    }
    if (a.variance_in_reproject && !slow) return;
    if (!mine) return;
    if (csn.w == 0.0f) { di_out[center] = cdi; gi_out[center] = cgi; return; }  // sky
    float di_var, gi_var;
    if (!slow) {
        di_var = cdi_m.z - sqr(cdi_m.y);
        gi_var = cgi_m.z - sqr(cgi_m.y);
    } else {
        variance_short_history(csn, cdi, cgi, s_sn, s_l, ((int)(pos.y & 7u) + 2) * PITCH + (int)(wave * 8u + (pos.x & 7u)) + 3, PITCH, &di_var, &gi_var);
    }
    di_var = fmax_(di_var, 0.0f);
    gi_var = fmax_(gi_var, 0.0f);
    if (a.variance_in_reproject) {  // in place, the w component only: other blocks are staging these texels' colours right now
        reinterpret_cast<float*>(&di_out[center])[3] = di_var;
        reinterpret_cast<float*>(&gi_out[center])[3] = gi_var;
        return;
    }
    di_out[center] = f4(xyz(cdi), di_var);
    gi_out[center] = f4(xyz(cgi), gi_var);
}

// ---------------------------------------------------------------- frame_denoising.rs:219-361 (five à-trous passes)
// Measured (rocprofv3, MI355X): the LDS-staged passes run at 70-75 % of their VALU issue time and within 10-30 % of the
// device-copy rate at once, the gather passes are bound by address processing. The layout below
// keeps the traffic at the compulsory bytes:
//   * strides 1 and 2 run as ONE launch (k_denoise_wavelet_12): a block stages the (32+6) x (16+6) window of its 32x16
//     pixels once, runs the stride-1 pass for the (32+4) x (16+4) pixels the stride-2 taps will touch, keeps those results
//     in LDS, and runs the stride-2 pass from there. Both outputs are stored (the first pass's output is next frame's colour
//     history, the second feeds the stride-4 pass): 48 B x 1.63 read + 64 B written per pixel for two passes, against
//     2 x (48 B x 1.33 + 32 B) for two separate 32x8-tiled launches.
//   * stride 4 (jitter still truncates to 0, frame_denoising.rs:262-266) stages a (32+8) x (16+8) window.
//   * strides 8 and 16 have per-pixel jitter and no reuse to stage; they gather through L2 (k_denoise_wavelet_far).
// sqrt(luma) of a tap (frame_denoising.rs:370-372) is recomputed from the staged colours: it costs two hardware square
// roots per tap and saves an 8-B plane that had to be written by every pass and read back, halo included, by the next.
// The weight is exp(-|sqrt(luma_c) - sqrt(luma_s)| * luma_sigma) * depth_weight * normal_weight, multiplied in that order
// (frame_denoising.rs:363-398); depth and normal factors are shared by the direct and the indirect signal, which ride the
// two halves of packed-f32 operations. A tap whose shared factor is exactly zero is dropped before its colours are read:
// its weight would be 0 or NaN and `w > 0` (frame_denoising.rs:318,340) rejects both.
struct WaveletCenter {
    float4 sn; V3 n; f2 sqrt_luma, luma_sigma; float leeway, inv_leeway;
    f2 sum_w, sum_r, sum_g, sum_b, sum_v;
};
ST_D f2 sqrt_luma2(float4 di, float4 gi) { return sqrt2((mk2(di.x, gi.x) * 0.2126f + mk2(di.y, gi.y) * 0.7152f) + mk2(di.z, gi.z) * 0.0722f); }
ST_D WaveletCenter wavelet_begin(float4 csn, float4 cdi, float4 cgi, float strength, f2 c_sqrt_luma) {
    WaveletCenter c;
    c.sn = csn; c.n = v3(csn.x, csn.y, csn.z);
    c.sqrt_luma = c_sqrt_luma;
    c.luma_sigma = mk2(lerpf(2.5f, 0.5f, fsqrt(cdi.w)), lerpf(1.0f, 0.0f, fsqrt(cgi.w)));
    c.leeway = csn.w * (0.33f / strength); c.inv_leeway = frcp(c.leeway);  // depth sigma is the same for both signals
    c.sum_w = splat2(1.0f); c.sum_r = mk2(cdi.x, cgi.x); c.sum_g = mk2(cdi.y, cgi.y); c.sum_b = mk2(cdi.z, cgi.z); c.sum_v = mk2(cdi.w, cgi.w);
    return c;
}
// depth_weight * ... and normal_weight of a tap's surface; false = the tap contributes nothing
ST_D bool wavelet_shared(const WaveletCenter& c, float4 ssn, float* depth_weight, float* normal_weight) {
    if (ssn.w == 0.0f) return false;
    const float diff = fabsf(ssn.w - c.sn.w);
    *depth_weight = diff >= c.leeway ? 0.0f : 1.0f - div_by(diff, c.leeway, c.inv_leeway);
    *normal_weight = pow64_(fmax_(dot(v3(ssn.x, ssn.y, ssn.z), c.n), 0.0f));
    return !(*depth_weight == 0.0f || *normal_weight == 0.0f);
}
// `t_sqrt_luma`: sqrt_luma2() of the tap's two colours (the LDS passes evaluate it once per staged texel, not once per tap)
ST_D void wavelet_tap(WaveletCenter& c, float4 sdi, float4 sgi, f2 t_sqrt_luma, float depth_weight, float normal_weight) {
    const f2 r = mk2(sdi.x, sgi.x), g = mk2(sdi.y, sgi.y), b = mk2(sdi.z, sgi.z), v = mk2(sdi.w, sgi.w);
    const f2 d = c.sqrt_luma - t_sqrt_luma;
    const f2 luma_weight = mk2(fabsf(d.x), fabsf(d.y)) * c.luma_sigma;
    const f2 w = exp_pair(-luma_weight) * depth_weight * normal_weight;
    if (w.x > 0.0f && w.y > 0.0f) {
        c.sum_w = c.sum_w + w; c.sum_r = c.sum_r + w * r; c.sum_g = c.sum_g + w * g; c.sum_b = c.sum_b + w * b; c.sum_v = c.sum_v + (w * w) * v;
    } else {
        if (w.x > 0.0f) { c.sum_w.x += w.x; c.sum_r.x += w.x * r.x; c.sum_g.x += w.x * g.x; c.sum_b.x += w.x * b.x; c.sum_v.x += (w.x * w.x) * v.x; }
        if (w.y > 0.0f) { c.sum_w.y += w.y; c.sum_r.y += w.y * r.y; c.sum_g.y += w.y * g.y; c.sum_b.y += w.y * b.y; c.sum_v.y += (w.y * w.y) * v.y; }
    }
}
struct WaveletOut { float4 di, gi; bool lit; };  // lit == false: sky pixel — `di` is the copied direct colour, the indirect output is left alone
ST_D WaveletOut wavelet_end(const WaveletCenter& c) {
    const f2 ww = c.sum_w * c.sum_w;
    WaveletOut o;
    o.di = wavelet_resolve(c.sum_r.x, c.sum_g.x, c.sum_b.x, c.sum_v.x, c.sum_w.x, ww.x);
    o.gi = wavelet_resolve(c.sum_r.y, c.sum_g.y, c.sum_b.y, c.sum_v.y, c.sum_w.y, ww.y);
    o.lit = true;
    return o;
}
// one pixel of a zero-jitter pass whose window is in LDS: `lc` = the pixel's texel, taps at +-S texels / rows (pitch P).
// Texels outside the viewport were staged with depth 0, which is skipped like an out-of-bounds or sky tap. Returns false
// for a sky pixel (the reference copies the direct colour and leaves the indirect output alone).
// (Evaluating sqrt(luma) once per staged texel instead of once per tap measured neutral in round 3 — tools/experiments/README.md.)
template <int S, int P>
ST_D WaveletOut wavelet_pixel_lds(const float4* s_sn, const float4* s_di, const float4* s_gi, int lc, float strength) {
    const float4 csn = s_sn[lc], cdi = s_di[lc], cgi = s_gi[lc];
    if (csn.w == 0.0f) { WaveletOut o; o.di = cdi; o.gi = cgi; o.lit = false; return o; }
    WaveletCenter c = wavelet_begin(csn, cdi, cgi, strength, sqrt_luma2(cdi, cgi));
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const int k = t < 4 ? t : t + 1, ox = k % 3 - 1, oy = k / 3 - 1;
        const int lt = lc + oy * S * P + ox * S;
        float dw, nw;
        if (!wavelet_shared(c, s_sn[lt], &dw, &nw)) continue;
        const float4 tdi = s_di[lt], tgi = s_gi[lt];
        wavelet_tap(c, tdi, tgi, sqrt_luma2(tdi, tgi), dw, nw);
    }
    return wavelet_end(c);
}

// Block geometry of the LDS-staged passes: 32 x 16 pixels, 512 threads, thread t -> pixel (t % 32, t / 32), so a wave
// reads two 32-texel row segments from LDS (conflict-free ds_read_b128 with a row pitch = 8 mod 16 texels) and stores two
// 512-B runs per plane. Blocks are dealt to XCDs like the 32x8 tiling does (st_device.h tile_for_thread): mode 2 keeps
// two block rows (32 pixel rows) per XCD chunk so the halo rows a block shares with its vertical neighbour stay in one L2.
constexpr int kWvW = 32, kWvH = 16, kWvThreads = kWvW * kWvH;
struct WaveletBlock { int32_t x0, y0; bool valid; };
// blocks of the window: columns [col0 / BW, ceil(col1 / BW)) x block rows from the window's first tile row
__host__ __device__ inline void wavelet_grid(const KArgs& a, uint32_t bw, uint32_t* bx0, uint32_t* groups_x, uint32_t* rows) {
    const uint32_t c1 = a.col1 < a.width ? a.col1 : a.width;
    *bx0 = a.col0 / bw;
    const uint32_t bx1 = (c1 + bw - 1u) / bw;
    *groups_x = bx1 > *bx0 ? bx1 - *bx0 : 0u;
    const uint32_t ty0 = a.row0 >> 3, ty1 = (a.row1 + 7u) >> 3;
    *rows = ((ty1 - ty0) * 8u + kWvH - 1u) / kWvH;
}
template <int BW = kWvW>
ST_D WaveletBlock wavelet_block(const KArgs& a) {
    uint32_t bx0, groups_x, rows;
    wavelet_grid(a, BW, &bx0, &groups_x, &rows);
    const uint32_t ty0 = a.row0 >> 3;
    const uint32_t n_blocks = groups_x * rows, b = blockIdx.x;
    uint32_t lin = b;
    if (a.tile_map == 0u) {
        const uint32_t q = n_blocks >> 3, r = n_blocks & 7u, xcd = b & 7u, k = b >> 3;
        lin = (xcd < r ? xcd * (q + 1u) : r * (q + 1u) + (xcd - r) * q) + k;
    } else if (a.tile_map == 2u) {
        const uint32_t chunk = groups_x * 2u * 8u, full = (n_blocks / chunk) * chunk;
        if (b < full) { const uint32_t c = b / chunk, i = b - c * chunk; lin = c * chunk + (i & 7u) * (groups_x * 2u) + (i >> 3); }
    }
    WaveletBlock w;
    const uint32_t gy = lin / groups_x, gx = lin - gy * groups_x;
    w.x0 = (int32_t)((bx0 + gx) * BW); w.y0 = (int32_t)(ty0 * 8u + gy * kWvH);
    w.valid = gy < rows;
    return w;
}
inline uint32_t wavelet_blocks(const KArgs& a, uint32_t block_w = kWvW) {
    uint32_t bx0, groups_x, rows;
    wavelet_grid(a, block_w, &bx0, &groups_x, &rows);
    return groups_x * rows;
}
// the pixel belongs to this launch's window (the LDS passes address pixels as signed block offsets)
ST_D bool wavelet_owns(const KArgs& a, int32_t px, int32_t py) { return px >= 0 && py >= 0 && owns_pixel(a, u2((uint32_t)px, (uint32_t)py)); }

void launch_denoise_variance(const KArgs& a, float4* di_out, float4* gi_out, hipStream_t s) { ST_LAUNCH(k_denoise_variance, false, s, a, di_out, gi_out); }

// stages the (kWvW + 2 HALO) x (kWvH + 2 HALO) window around the block into LDS (row pitch P texels)
template <int HALO, int P, int BW = kWvW>
ST_D void wavelet_stage(const KArgs& a, const WaveletBlock& blk, const float4* di_in, const float4* gi_in, float4* s_sn, float4* s_di, float4* s_gi) {
    constexpr int WW = BW + 2 * HALO, WH = kWvH + 2 * HALO;
    for (int i = (int)threadIdx.x; i < WW * WH; i += kWvThreads) {
        const int ry = i / WW, rx = i - ry * WW;
        const int32_t gx = blk.x0 - HALO + rx, gy = blk.y0 - HALO + ry;
        const int li = ry * P + rx;
        if (gx >= 0 && gy >= 0 && gx < (int32_t)a.width && gy < (int32_t)a.height) {
            const uint32_t at = (uint32_t)gy * a.width + (uint32_t)gx;
            const float4 tsn = a.sn[at], tdi = di_in[at], tgi = gi_in[at];
            s_sn[li] = tsn; s_di[li] = tdi; s_gi[li] = tgi;
        } else {
            s_sn[li] = f4z();
        }
    }
}

// ---- strides 1 + 2 in one launch. di_in / gi_in: the variance pass's output; di_mid / gi_mid: the stride-1 pass's output
// (prev_colors planes: next frame's history); di_out / gi_out: the stride-2 pass's output (stash planes). In the reference
// the variance pass writes the stash planes too (stash -> prev -> stash); here a block would then read halo texels that a
// neighbouring block has already overwritten with its stride-2 results, so the variance pass of this launch group writes
// an internal pair of planes instead (st_engine.cpp) and the stash planes first receive the stride-2 output.
__global__ __launch_bounds__(kWvThreads) void k_denoise_wavelet_12(const KArgs a, float strength0, float strength1, const float4* di_in, float4* di_mid, float4* di_out,
                                                                   const float4* gi_in, float4* gi_mid, float4* gi_out) {
    // pitch 38: 40,128 B of LDS, four blocks per CU (pitch 40: three; measured 68.9 -> 63.3 us). A wave's ds_read_b128 covers whole 16-texel row runs either way
    constexpr int HALO = 3, WW = kWvW + 2 * HALO, WH = kWvH + 2 * HALO, P = 38;  // 38 x 22 texels
    constexpr int RW = kWvW + 4, RH = kWvH + 4;                                   // 36 x 20: where the stride-1 pass must run
    __shared__ float4 s_sn[P * WH];
    __shared__ float4 s_di[P * WH];
    __shared__ float4 s_gi[P * WH];
    const WaveletBlock blk = wavelet_block(a);
    if (!blk.valid) return;
    wavelet_stage<HALO, P>(a, blk, di_in, gi_in, s_sn, s_di, s_gi);
    __syncthreads();
    // stride-1 pass over the 36 x 20 region (row-major over the threads: 1.4 pixels each)
    float4 r_di[2], r_gi[2]; int r_at[2];
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const int idx = (int)threadIdx.x + it * kWvThreads;
        r_at[it] = -1;
        if (idx >= RW * RH) continue;
        const int ry = idx / RW, rx = idx - ry * RW;
        const int lc = (ry + 1) * P + rx + 1;
        r_at[it] = lc;
        const WaveletOut o = wavelet_pixel_lds<1, P>(s_sn, s_di, s_gi, lc, strength0);
        r_di[it] = o.di; r_gi[it] = o.gi;  // sky (or outside the viewport): the indirect colour is never read, as a tap or as a centre
        const bool lit = o.lit;
        // the block's own pixels: this is what the stand-alone stride-1 pass stores
        const int32_t px = blk.x0 - 2 + rx, py = blk.y0 - 2 + ry;
        if (rx >= 2 && rx < 2 + kWvW && ry >= 2 && ry < 2 + kWvH && wavelet_owns(a, px, py)) {
            const uint32_t center = (uint32_t)py * a.width + (uint32_t)px;
            di_mid[center] = r_di[it];
            if (lit) gi_mid[center] = r_gi[it];
        }
    }
    __syncthreads();  // every stride-1 read of the staged colours is done: replace them by the stride-1 results
#pragma unroll
    for (int it = 0; it < 2; it++) if (r_at[it] >= 0) {
        s_di[r_at[it]] = r_di[it]; s_gi[r_at[it]] = r_gi[it];
    }
    __syncthreads();
    // stride-2 pass for the block's own pixels
    const int x = (int)(threadIdx.x & 31u), y = (int)(threadIdx.x >> 5);
    const int32_t px = blk.x0 + x, py = blk.y0 + y;
    if (!wavelet_owns(a, px, py)) return;
    const uint32_t center = (uint32_t)py * a.width + (uint32_t)px;
    const WaveletOut o = wavelet_pixel_lds<2, P>(s_sn, s_di, s_gi, (y + HALO) * P + x + HALO, strength1);
    di_out[center] = o.di;
    // On a sky pixel the stride-2 pass leaves its indirect output alone, and what the reference's stash plane holds there is
    // the variance pass's copy of the input colour — which this launch group never stored there: `o.gi` is that colour.
    gi_out[center] = o.gi;
}

// ---- a single zero-jitter pass staged through LDS (stride 4; also strides 1 and 2 when run unfused)
template <int S>
__global__ __launch_bounds__(kWvThreads) void k_denoise_wavelet_lds(const KArgs a, float strength, const float4* di_in, float4* di_out, const float4* gi_in, float4* gi_out) {
    // row pitch: the smallest P >= window width with P = 8 mod 16 texels (conflict-free ds_read_b128 of two row segments per
    // wave). Stride 4: 40 texels -> 46 KB of LDS per block, three blocks per CU (56 gave 64.5 KB and two)
    constexpr int WH = kWvH + 2 * S, P = (kWvW + 2 * S + 7) / 16 * 16 + 8;
    static_assert(P >= kWvW + 2 * S && P % 16 == 8, "pitch");
    __shared__ float4 s_sn[P * WH];
    __shared__ float4 s_di[P * WH];
    __shared__ float4 s_gi[P * WH];
    const WaveletBlock blk = wavelet_block(a);
    if (!blk.valid) return;
    wavelet_stage<S, P>(a, blk, di_in, gi_in, s_sn, s_di, s_gi);
    __syncthreads();
    const int x = (int)(threadIdx.x & 31u), y = (int)(threadIdx.x >> 5);
    const int32_t px = blk.x0 + x, py = blk.y0 + y;
    if (!wavelet_owns(a, px, py)) return;
    const uint32_t center = (uint32_t)py * a.width + (uint32_t)px;
    const WaveletOut o = wavelet_pixel_lds<S, P>(s_sn, s_di, s_gi, (y + S) * P + x + S, strength);
    di_out[center] = o.di;
    if (o.lit) gi_out[center] = o.gi;
}

// ---- strides 8 and 16: jittered taps, gathered through L2. Three dependent rounds of eight independent loads each
// (surfaces, direct colours, indirect colours) with scheduling barriers between them: the shared depth / normal factors are
// reduced to two floats per tap before the colours are requested and the direct signal is finished before the indirect one
// is loaded, so that at most eight texels are live at a time (6 waves per SIMD cover the gather latency; with all 24 loads
// hoisted the kernel ran at 4). One signal at a time means scalar instead of packed arithmetic — the same IEEE operations.
struct WaveletSignal { float sqrt_luma, sigma, sw, sr, sg, sb, sv; };
ST_D WaveletSignal signal_begin(float4 c, float sigma_hi, float sigma_lo) {
    WaveletSignal s;
    s.sqrt_luma = fsqrt(luma(xyz(c))); s.sigma = lerpf(sigma_hi, sigma_lo, fsqrt(c.w));
    s.sw = 1.0f; s.sr = c.x; s.sg = c.y; s.sb = c.z; s.sv = c.w;
    return s;
}
ST_D void signal_tap(WaveletSignal& s, float4 t, float depth_weight, float normal_weight) {
    const float d = s.sqrt_luma - fsqrt(luma(xyz(t)));
    const float w = exp_(-(fabsf(d) * s.sigma)) * depth_weight * normal_weight;
    if (w > 0.0f) { s.sw += w; s.sr += w * t.x; s.sg += w * t.y; s.sb += w * t.z; s.sv += (w * w) * t.w; }
}
ST_D float4 signal_end(const WaveletSignal& s) { return wavelet_resolve(s.sr, s.sg, s.sb, s.sv, s.sw, s.sw * s.sw); }
// COMPOSE (fast build, CameraMode::Image): frame_composition.rs:18-82 runs for the same pixel at the end of the LAST pass — it
// reads this pixel's two denoised colours, in registers here, plus planes no a-trous pass writes. What composition needs
// from those planes is fetched with the centre texels, ahead of the three gather rounds, and reduced to six floats (the
// G-buffer's base colour; emissive + both specular samples): fetched at the end instead, the dependent chain G-buffer ->
// byte table -> store sat behind the last round with nothing left to overlap it (measured: 119 us against 48 + 51 us for
// the two separate launches). The sum's association differs from compose_pixel's, hence fast build only.
// `keep_colours` == 0 (the lean frame) leaves the pass's own output planes unwritten: nothing but composition reads them.
struct ComposeArgs { void* out; uint32_t format, camera_mode, keep_colours; };
#ifndef ST_FAR_COMPOSE_WAVES
#define ST_FAR_COMPOSE_WAVES 5  // the six floats composition carries through the gather rounds do not fit the 80 registers of 6 waves per SIMD without spilling
#endif
template <bool COMPOSE>
__global__ __launch_bounds__(kBlockThreads, COMPOSE ? ST_FAR_COMPOSE_WAVES : 6) void k_denoise_wavelet_far(const KArgs a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in, float4* gi_out, const ComposeArgs co) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t center = pos.y * a.width + pos.x;
    const float4 csn = a.sn[center];
    const float4 cdi = di_in[center];
    const bool store = !COMPOSE || co.keep_colours != 0u;
    if (csn.w == 0.0f) {  // sky
        if (store) di_out[center] = cdi;
        if (COMPOSE) store_output(co.out, center, f4(xyz(cdi), 1.0f), co.format);  // frame_composition.rs: depth == 0 shows the direct colour
        return;
    }
    V3 c_base = v3s(0.0f), c_add = v3s(0.0f);
    if (COMPOSE) {
        const float4 g1 = a.g1[center], ds = a.di_spec_samples[center], gs = a.gi_spec_samples[center];
        const uint32_t w1 = f2b(g1.w);
        const float* lut = a.byte_luts;
        c_base = v3(lut[kLutGamma8 + (w1 & 0xffu)], lut[kLutGamma8 + ((w1 >> 8) & 0xffu)], lut[kLutGamma8 + ((w1 >> 16) & 0xffu)]);  // gbuffer_unpack's base colour
        c_add = xyz(g1) + xyz(ds) + xyz(gs);                                                                                      // emissive + specular samples
    }
    const float4 bn = blue_noise_read(a, pos);
    const I2 jitter = as_i2((v2(bn.z, bn.w) - 0.5f) * ((float)stride - 1.0f) * 0.5f);
    uint32_t at[8];
    float dw[8], nw[8];
    {
        WaveletCenter c; c.sn = csn; c.n = v3(csn.x, csn.y, csn.z);
        c.leeway = csn.w * (0.33f / strength); c.inv_leeway = frcp(c.leeway);
        float4 ssn[8];
#pragma unroll
        for (int t = 0; t < 8; t++) {
            const int k = t < 4 ? t : t + 1, ox = k % 3 - 1, oy = k / 3 - 1;
            const I2 sp = i2((int32_t)pos.x + jitter.x + ox * (int32_t)stride, (int32_t)pos.y + jitter.y + oy * (int32_t)stride);
            const bool inside = contains_i(a, sp);
            at[t] = inside ? (uint32_t)sp.y * a.width + (uint32_t)sp.x : center;  // an out-of-bounds tap reads the centre and is masked below
            ssn[t] = a.sn[at[t]];
            if (!inside) ssn[t].w = 0.0f;
        }
#pragma unroll
        for (int t = 0; t < 8; t++) if (!wavelet_shared(c, ssn[t], &dw[t], &nw[t])) { dw[t] = 0.0f; nw[t] = 0.0f; at[t] = center; }  // a dead tap re-reads the centre's line
    }
    __builtin_amdgcn_sched_barrier(0);
    float4 res_di;
    {
        WaveletSignal sg = signal_begin(cdi, 2.5f, 0.5f);
        float4 tap[8];
#pragma unroll
        for (int t = 0; t < 8; t++) tap[t] = di_in[at[t]];
#pragma unroll
        for (int t = 0; t < 8; t++) if (dw[t] != 0.0f) signal_tap(sg, tap[t], dw[t], nw[t]);
        res_di = signal_end(sg);
        if (store) di_out[center] = res_di;
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        WaveletSignal sg = signal_begin(gi_in[center], 1.0f, 0.0f);
        float4 tap[8];
#pragma unroll
        for (int t = 0; t < 8; t++) tap[t] = gi_in[at[t]];
#pragma unroll
        for (int t = 0; t < 8; t++) if (dw[t] != 0.0f) signal_tap(sg, tap[t], dw[t], nw[t]);
        const float4 res_gi = signal_end(sg);
        if (store) gi_out[center] = res_gi;
        if (COMPOSE) store_output(co.out, center, f4(c_add + (xyz(res_di) + xyz(res_gi)) * c_base, 1.0f), co.format);
    }
}

void launch_denoise_wavelet(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in,
                            float4* gi_out, hipStream_t s) {
    const uint32_t blocks = wavelet_blocks(a);
    if (!blocks) return;
    if (stride == 1u) ST_KLAUNCH((k_denoise_wavelet_lds<1>), dim3(blocks), dim3(kWvThreads), s, a, strength, di_in, di_out, gi_in, gi_out);
    else if (stride == 2u) ST_KLAUNCH((k_denoise_wavelet_lds<2>), dim3(blocks), dim3(kWvThreads), s, a, strength, di_in, di_out, gi_in, gi_out);
    else if (stride == 4u) ST_KLAUNCH((k_denoise_wavelet_lds<4>), dim3(blocks), dim3(kWvThreads), s, a, strength, di_in, di_out, gi_in, gi_out);
    else ST_LAUNCH(k_denoise_wavelet_far<false>, false, s, a, stride, strength, di_in, di_out, gi_in, gi_out, ComposeArgs{nullptr, 0u, 0u, 1u});
}
// a gather pass (stride 8 or 16) with frame_composition.rs appended (the last pass of the chain)
void launch_denoise_wavelet_compose(const KArgs& a, uint32_t stride, float strength, const float4* di_in, float4* di_out, const float4* gi_in, float4* gi_out,
                                    uint32_t camera_mode, void* out, uint32_t format, bool keep_colours, hipStream_t s) {
    ST_LAUNCH(k_denoise_wavelet_far<true>, false, s, a, stride, strength, di_in, di_out, gi_in, gi_out, ComposeArgs{out, format, camera_mode, keep_colours ? 1u : 0u});
}
void launch_denoise_wavelet_12(const KArgs& a, float strength0, float strength1, const float4* di_in, float4* di_mid, float4* di_out, const float4* gi_in,
                               float4* gi_mid, float4* gi_out, hipStream_t s) {
    const uint32_t blocks = wavelet_blocks(a);
    if (blocks) ST_KLAUNCH(k_denoise_wavelet_12, dim3(blocks), dim3(kWvThreads), s, a, strength0, strength1, di_in, di_mid, di_out, gi_in, gi_mid, gi_out);
}

// ---------------------------------------------------------------- st_camera_write_buffer support
// `which`: bit 0 = the current frame's twin (KArgs::sn from KArgs::sm), bit 1 = the previous frame's (psn from psm) — only the
// twin of a surface map that was actually replaced: in the lean frame the encoded maps are not kept up to date
__global__ ST_KERNEL_BOUNDS void k_refresh_internal_planes(const KArgs a, float4* psn_out, uint32_t which) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !contains_u(a, pos)) return;
    const uint32_t i = pos.y * a.width + pos.x;
    if (which & 1u) { const float4 sm = a.sm[i]; a.sn[i] = sm.z == 0.0f ? f4z() : f4(normal_decode(v2(sm.x, sm.y)), sm.z); }   // what primary visibility writes beside the surface map
    if (which & 2u) { const float4 psm = a.psm[i]; psn_out[i] = psm.z == 0.0f ? f4z() : f4(normal_decode(v2(psm.x, psm.y)), psm.z); }
}
void launch_refresh_internal_planes(const KArgs& a_in, uint32_t which, hipStream_t s) {
    KArgs a = a_in; a.row0 = 0; a.row1 = a.height; a.col0 = 0; a.col1 = a.width;
    ST_LAUNCH(k_refresh_internal_planes, false, s, a, const_cast<float4*>(a.psn), which);
}

}  // namespace ST_KNS
}  // namespace st
