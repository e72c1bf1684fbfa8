// st_math.h — scalar/vector maths shared by the HIP kernels and the C++ host engine.
//
// The kernels are built twice (Makefile). In the EXACT build — and always on the host — two rules make every result
// bit-identical on gfx950 and on x86-64 (which is what lets the parity tests demand exact equality instead of tolerances):
//   1. only +, -, *, / and sqrt are used (correctly rounded on both; hipcc's default
//      -fhip-fp32-correctly-rounded-divide-sqrt is relied upon) and everything is compiled with
//      -ffp-contract=off, so no FMA is formed outside the IEEE division/sqrt expansions;
//   2. transcendentals (the reference leaves them to the SPIR-V driver: strolle-gpu/src/*.rs via
//      `spirv_std::num_traits::Float`) are evaluated by the fixed polynomial kernels below
//      (Cephes-style single precision, a few ulp — well inside Vulkan's GLSL.std.450 envelope).
// Vector operation order follows glam 0.24 scalar maths (the reference's vector library).
//
// In the FAST build (-DST_FAST_MATH=1, device code only) division, square root and the transcendentals go to the hardware
// (v_rcp_f32, v_sqrt_f32, v_rsq_f32, v_exp_f32, v_log_f32, v_sin_f32, v_cos_f32: 1 ulp each) through the helpers below
// (fdiv, frcp, fsqrt, ...), and the compiler may contract a*b+c into FMAs (-ffp-contract=fast-honor-pragmas). The reference
// itself leaves all of these to the SPIR-V driver's precision. What must stay exact in both builds — ray generation and the
// BVH traversal compare chain, whose `used_memory` integers are compared bit for bit — is written with plain `/`, sqrtf
// and its own helpers inside `#pragma clang fp contract(off)` regions ("exact island", st_device.h).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ST_HD __host__ __device__ __forceinline__
#if defined(ST_FAST_MATH) && defined(__HIP_DEVICE_COMPILE__)
#define ST_FAST_DEVICE 1
#else
#define ST_FAST_DEVICE 0
#endif

namespace st {

constexpr float kF32Max = 3.40282347e+38f;
constexpr float kF32Eps = 1.1920929e-7f;
constexpr float kPi = 3.14159265358979323846f;
constexpr float kHalfPi = 1.5707963267948966f;

ST_HD uint32_t f2b(float f) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __float_as_uint(f);
#else
    uint32_t u; memcpy(&u, &f, 4); return u;
#endif
}
ST_HD float b2f(uint32_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float(u);
#else
    float f; memcpy(&f, &u, 4); return f;
#endif
}
// Rust `as u32` / `as i32`: saturating, NaN -> 0
ST_HD uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u;
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
ST_HD int32_t f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)f;
}
ST_HD float fmin_(float a, float b) { return fminf(a, b); }  // IEEE minNum == Rust f32::min
ST_HD float fmax_(float a, float b) { return fmaxf(a, b); }
ST_HD float clampf(float x, float lo, float hi) { return x < lo ? lo : (x > hi ? hi : x); }  // Rust f32::clamp
ST_HD float saturate(float x) { return clampf(x, 0.0f, 1.0f); }
ST_HD float sqr(float x) { return x * x; }
// division, reciprocal, square root: correctly rounded in the exact build, one hardware instruction each in the fast build
#if ST_FAST_DEVICE
ST_HD float frcp(float x) { return __builtin_amdgcn_rcpf(x); }
ST_HD float fdiv(float a, float b) { return a * __builtin_amdgcn_rcpf(b); }
ST_HD float fsqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
ST_HD float inverse_sqrt(float x) { return __builtin_amdgcn_rsqf(x); }
ST_HD float fdivc(float x, float c) { return x * (1.0f / c); }  // division by a compile-time constant: its reciprocal is folded
#else
ST_HD float frcp(float x) { return 1.0f / x; }
ST_HD float fdiv(float a, float b) { return a / b; }
ST_HD float fsqrt(float x) { return sqrtf(x); }
ST_HD float inverse_sqrt(float x) { return 1.0f / sqrtf(x); }
ST_HD float fdivc(float x, float c) { return x / c; }
#endif
ST_HD float signum(float x) { return x != x ? x : copysignf(1.0f, x); }

// ------------------------------------------------------------------ deterministic transcendentals
// (exact island: these polynomial kernels are never contracted, so the byte tables and the host's uses of them are the
// same bits whichever build of the kernels runs)
#pragma clang fp contract(off)
ST_HD float scale2(float z, int n) {
    if (n > 127) { z *= b2f(0x7f000000u); n -= 127; if (n > 127) n = 127; }
    if (n < -126) { z *= b2f(0x00800000u); n += 126; if (n < -126) n = -126; }
    return z * b2f((uint32_t)(n + 127) << 23);
}
ST_HD void sincos_poly_(float x, float* s_out, float* c_out) {
    const float ax = fabsf(x);
    int j = (int)(ax * 1.27323954473516f);
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    const float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    const float z = r * r;
    const float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    const float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    const bool swap = (j & 2) != 0;       // octants 2, 6: sin <-> cos
    float s = swap ? pc : ps;
    float c = swap ? ps : pc;
    if (j == 4 || j == 6) s = -s;
    if (j == 2 || j == 4) c = -c;
    if (x < 0.0f) s = -s;
    *s_out = s; *c_out = c;
}
ST_HD void sincos_(float x, float* s_out, float* c_out) {
#if ST_FAST_DEVICE && !defined(ST_ABL_SINCOS_POLY)   // (ST_ABL_*: build variants of tools/abl_build.sh — which fast-math substitution spends the tolerance gates? DESIGN.md section 2.2)
    const float turns = x * 0.15915494309189535f;  // v_sin_f32 / v_cos_f32 take revolutions (valid to +-256 turns)
    *s_out = __builtin_amdgcn_sinf(turns); *c_out = __builtin_amdgcn_cosf(turns);
#else
    sincos_poly_(x, s_out, c_out);
#endif
}
ST_HD float sin_(float x) { float s, c; sincos_(x, &s, &c); return s; }
ST_HD float cos_(float x) { float s, c; sincos_(x, &s, &c); return c; }
ST_HD float asin_pos_(float a) {
    float x, z; const bool flag = a > 0.5f;
    if (flag) { z = 0.5f * (1.0f - a); x = fsqrt(z); } else { x = a; z = x * x; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
    if (flag) { p = p + p; p = kHalfPi - p; }
    return p;
}
ST_HD float acos_(float x) {
    if (x < -0.5f) return kPi - 2.0f * asin_pos_(fsqrt(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * asin_pos_(fsqrt(0.5f * (1.0f - x)));
    float s = asin_pos_(fabsf(x));
    if (x < 0.0f) s = -s;
    return kHalfPi - s;
}
ST_HD float exp_poly_(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.0f;
    const float z = floorf(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    const int n = (int)z;
    const float zz = x * x;
    const float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * zz + x + 1.0f;
    return scale2(p, n);
}
ST_HD float log2_poly_(float x) {
    uint32_t b = f2b(x);
    int e = 0;
    if ((b & 0x7f800000u) == 0) { x *= 8388608.0f; b = f2b(x); e = -23; }
    e += (int)((b >> 23) & 0xff) - 126;
    float m = b2f((b & 0x007fffffu) | 0x3f000000u);
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    const float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;
    float r = y * 0.44269504088896340735992f;
    r += m * 0.44269504088896340735992f;
    r += y;
    r += m;
    r += (float)e;
    return r;
}
ST_HD float exp2_poly_(float x) {
    if (x != x) return x;
    if (x > 127.999f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    const float px = floorf(x);
    int i0 = (int)px;
    x = x - px;
    if (x > 0.5f) { i0 += 1; x = x - 1.0f; }
    const float p = (((((1.535336188319500e-4f * x + 1.339887440266574e-3f) * x + 9.618437357674640e-3f) * x + 5.550332471162809e-2f) * x + 2.402264791363012e-1f) * x + 6.931472028550421e-1f) * x + 1.0f;
    return scale2(p, i0);
}
ST_HD float pow_poly_(float x, float y) {  // x >= 0, finite y > 0 (all the path needs)
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return 0.0f;
    if (x == INFINITY) return INFINITY;
    if (x == 1.0f) return 1.0f;
    return exp2_poly_(y * log2_poly_(x));
}
#if defined(ST_FAST_MATH)
#pragma clang fp contract(fast)
#endif
// the names the passes call: the fixed polynomials above in the exact build and on the host, the hardware in the fast build
#if ST_FAST_DEVICE
ST_HD float exp_(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }
ST_HD float log2_(float x) { return __builtin_amdgcn_logf(x); }
ST_HD float exp2_(float x) { return __builtin_amdgcn_exp2f(x); }
ST_HD float pow_(float x, float y) { return __builtin_amdgcn_exp2f(y * __builtin_amdgcn_logf(x)); }  // 0 -> exp2(-inf) = 0, 1 -> 1, inf -> inf, negative / NaN -> NaN
#else
ST_HD float exp_(float x) { return exp_poly_(x); }
ST_HD float log2_(float x) { return log2_poly_(x); }
ST_HD float exp2_(float x) { return exp2_poly_(x); }
ST_HD float pow_(float x, float y) { return pow_poly_(x, y); }
#endif
// integer-exponent powf calls of the reference as exact multiplication chains (cheaper and closer to a correctly
// rounded pow than exp2(y*log2(x)); x^64 is evaluated 16x per pixel per wavelet pass)
ST_HD float pow2_(float x) { return x * x; }
ST_HD float pow3_(float x) { return x * x * x; }
ST_HD float pow5_(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x; }
ST_HD float pow8_(float x) { const float x2 = x * x; const float x4 = x2 * x2; return x4 * x4; }
ST_HD float pow64_(float x) { const float x2 = x * x; const float x4 = x2 * x2; const float x8 = x4 * x4; const float x16 = x8 * x8; const float x32 = x16 * x16; return x32 * x32; }

// f32 -> f16 bits, round to nearest even, in integer arithmetic (what a store to an Rgba16Float texel does)
ST_HD uint32_t f16_bits(float f) {
    const uint32_t u = f2b(f);
    const uint32_t sign = (u >> 16) & 0x8000u;
    const uint32_t a = u & 0x7fffffffu;
    if (a >= 0x7f800000u) return sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u);
    if (a >= 0x477ff000u) return sign | 0x7c00u;
    if (a < 0x38800000u) {
        if (a < 0x33000000u) return sign;
        const uint32_t m = (a & 0x007fffffu) | 0x00800000u;
        const uint32_t sft = 126u - (a >> 23);
        uint32_t r = m >> sft;
        const uint32_t rem = m & ((1u << sft) - 1u), half = 1u << (sft - 1u);
        if (rem > half || (rem == half && (r & 1u))) r += 1u;
        return sign | r;
    }
    const uint32_t b = a + 0xfffu + ((a >> 13) & 1u);
    return sign | ((b - 0x38000000u) >> 13);
}
// linear -> 8-bit sRGB (the store to an Rgba8UnormSrgb / Bgra8UnormSrgb render target: clamp, IEC 61966-2-1 transfer
// function, round to nearest). NaN and negatives give 0.
ST_HD uint32_t srgb8_encode(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 1.0f) return 255u;
    const float y = x <= 0.0031308f ? x * 12.92f : 1.055f * pow_(x, 1.0f / 2.4f) - 0.055f;
    return (uint32_t)(y * 255.0f + 0.5f);
}

// the G-buffer's gamma-encoded base colour bytes (gbuffer.rs:37-48): RGB8 + A6. Host and device evaluate it identically.
ST_HD uint32_t gbuffer_pack_base_color(float4 c) {
    const float ig = 1.0f / 2.2f;
    const float bx = clampf(pow_(c.x, ig), 0.0f, 1.0f), by = clampf(pow_(c.y, ig), 0.0f, 1.0f);
    const float bz = clampf(pow_(c.z, ig), 0.0f, 1.0f), bw = clampf(pow_(c.w, ig), 0.0f, 1.0f);
    return f2u_sat(bx * 255.0f) | (f2u_sat(by * 255.0f) << 8) | (f2u_sat(bz * 255.0f) << 16) | (f2u_sat(bw * 63.0f) << 24);
}

ST_HD float atan_(float x) {
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = kHalfPi; x = -frcp(x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = fdiv(x - 1.0f, x + 1.0f); }
    else y = 0.0f;
    const float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return sign * y;
}
ST_HD float atan2_(float y, float x) {
    if (x != x || y != y) return x + y;
    if (x == 0.0f) {
        if (y == 0.0f) return copysignf((f2b(x) >> 31) ? kPi : 0.0f, y);
        return y > 0.0f ? kHalfPi : -kHalfPi;
    }
    float a = atan_(fdiv(y, x));
    if (x < 0.0f) a = (y < 0.0f || (y == 0.0f && (f2b(y) >> 31))) ? a - kPi : a + kPi;
    return a;
}

// ------------------------------------------------------------------ vectors
struct V2 { float x, y; };
struct V3 { float x, y, z; };
struct I2 { int32_t x, y; };
struct U2 { uint32_t x, y; };

ST_HD V2 v2(float x, float y) { V2 r; r.x = x; r.y = y; return r; }
ST_HD V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
ST_HD V3 v3s(float s) { return v3(s, s, s); }
ST_HD V3 xyz(float4 v) { return v3(v.x, v.y, v.z); }
ST_HD float4 f4(V3 v, float w) { return make_float4(v.x, v.y, v.z, w); }
ST_HD float4 f4z() { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }
ST_HD I2 i2(int32_t x, int32_t y) { I2 r; r.x = x; r.y = y; return r; }
ST_HD U2 u2(uint32_t x, uint32_t y) { U2 r; r.x = x; r.y = y; return r; }

ST_HD V2 operator+(V2 a, V2 b) { return v2(a.x + b.x, a.y + b.y); }
ST_HD V2 operator-(V2 a, V2 b) { return v2(a.x - b.x, a.y - b.y); }
ST_HD V2 operator*(V2 a, V2 b) { return v2(a.x * b.x, a.y * b.y); }
ST_HD V2 operator/(V2 a, V2 b) { return v2(fdiv(a.x, b.x), fdiv(a.y, b.y)); }
ST_HD V2 operator*(V2 a, float s) { return v2(a.x * s, a.y * s); }
ST_HD V2 operator*(float s, V2 a) { return v2(s * a.x, s * a.y); }
#if ST_FAST_DEVICE
ST_HD V2 operator/(V2 a, float s) { const float r = frcp(s); return v2(a.x * r, a.y * r); }
#else
ST_HD V2 operator/(V2 a, float s) { return v2(a.x / s, a.y / s); }
#endif
ST_HD V2 operator+(V2 a, float s) { return v2(a.x + s, a.y + s); }
ST_HD V2 operator-(V2 a, float s) { return v2(a.x - s, a.y - s); }
ST_HD float dot(V2 a, V2 b) { return a.x * b.x + a.y * b.y; }
ST_HD V2 round2(V2 a) { return v2(roundf(a.x), roundf(a.y)); }
ST_HD V2 as_v2(U2 a) { return v2((float)a.x, (float)a.y); }
ST_HD U2 as_u2(V2 a) { return u2(f2u_sat(a.x), f2u_sat(a.y)); }
ST_HD I2 as_i2(V2 a) { return i2(f2i_sat(a.x), f2i_sat(a.y)); }

ST_HD V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
ST_HD V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
ST_HD V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
ST_HD V3 operator/(V3 a, V3 b) { return v3(fdiv(a.x, b.x), fdiv(a.y, b.y), fdiv(a.z, b.z)); }
ST_HD V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
ST_HD V3 operator*(float s, V3 a) { return v3(s * a.x, s * a.y, s * a.z); }
#if ST_FAST_DEVICE
ST_HD V3 operator/(V3 a, float s) { const float r = frcp(s); return v3(a.x * r, a.y * r, a.z * r); }
#else
ST_HD V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
#endif
ST_HD V3 operator/(float s, V3 a) { return v3(fdiv(s, a.x), fdiv(s, a.y), fdiv(s, a.z)); }
ST_HD V3 divc3(V3 a, float c) { return v3(fdivc(a.x, c), fdivc(a.y, c), fdivc(a.z, c)); }
ST_HD V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
ST_HD bool is_zero(V3 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f; }
ST_HD bool is_zero(float4 a) { return a.x == 0.0f && a.y == 0.0f && a.z == 0.0f && a.w == 0.0f; }
ST_HD float dot(V3 a, V3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
ST_HD V3 cross(V3 a, V3 b) { return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
ST_HD float length_squared(V3 a) { return dot(a, a); }
ST_HD float length(V3 a) { return fsqrt(dot(a, a)); }
#if ST_FAST_DEVICE
ST_HD V3 normalize(V3 a) { return a * inverse_sqrt(dot(a, a)); }
#else
ST_HD V3 normalize(V3 a) { return a * (1.0f / length(a)); }
#endif
ST_HD float distance(V3 a, V3 b) { return length(a - b); }
ST_HD V3 vmin(V3 a, V3 b) { return v3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
ST_HD V3 vmax(V3 a, V3 b) { return v3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
ST_HD V3 vclamp(V3 a, V3 lo, V3 hi) { return vmin(vmax(a, lo), hi); }
ST_HD V3 reflect(V3 self, V3 other) { return self - 2.0f * dot(other, self) * other; }  // utils/vec3_ext.rs:32
ST_HD float luma(V3 c) { return dot(c, v3(0.2126f, 0.7152f, 0.0722f)); }                  // utils/vec3_ext.rs:51
ST_HD float lerpf(float a, float b, float t) { return a + (b - a) * clampf(t, 0.0f, 1.0f); }  // utils.rs:21-30
ST_HD V3 lerp3(V3 a, V3 b, float t) { return a + (b - a) * clampf(t, 0.0f, 1.0f); }
ST_HD void any_orthonormal_pair(V3 n, V3* t, V3* b) {  // glam Vec3::any_orthonormal_pair
    const float sign = signum(n.z);
    const float a = -frcp(sign + n.z);
    const float bb = n.x * n.y * a;
    *t = v3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    *b = v3(bb, sign + n.y * n.y * a, -n.y);
}
ST_HD float glam_acos_approx(float v) {  // glam 0.24 math::acos_approx, behind Vec3::angle_between
    const bool nonnegative = v >= 0.0f;
    const float x = fabsf(v);
    float omx = 1.0f - x;
    if (omx < 0.0f) omx = 0.0f;
    const float root = fsqrt(omx);
    float r = ((((((-0.0012624911f * x + 0.0066700901f) * x - 0.0170881256f) * x + 0.0308918810f) * x - 0.0501743046f) * x + 0.0889789874f) * x - 0.2145988016f) * x + 1.5707963050f;
    r *= root;
    return nonnegative ? r : kPi - r;
}
ST_HD float angle_between(V3 a, V3 b) { return glam_acos_approx(fdiv(dot(a, b), fsqrt(length_squared(a) * length_squared(b)))); }

// float4 arithmetic uses HIP's own component-wise operators (amd_hip_vector_types.h): +, -, * and / by scalar.
#if ST_FAST_DEVICE
ST_HD float4 div4(float4 v, float s) { return v * frcp(s); }
#else
ST_HD float4 div4(float4 v, float s) { return v / s; }
#endif

// ------------------------------------------------------------------ 4x4 / affine (column-major, glam order)
struct M4 { float4 c[4]; };
ST_HD float4 mul(const M4& m, float4 v) {
    float4 r = m.c[0] * v.x;
    r = r + m.c[1] * v.y;
    r = r + m.c[2] * v.z;
    r = r + m.c[3] * v.w;
    return r;
}
ST_HD V3 project_point3(const M4& m, V3 p) {
    float4 r = m.c[0] * p.x;
    r = m.c[1] * p.y + r;
    r = m.c[2] * p.z + r;
    r = m.c[3] + r;
    const float rw = frcp(r.w);
    return v3(r.x * rw, r.y * rw, r.z * rw);
}

}  // namespace st
