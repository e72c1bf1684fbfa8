// ORACLE — TEST INFRASTRUCTURE ONLY. Never linked into, imported by, or
// executed from the product (strolle_amd/, libstrolle_hip.so). Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// CPU restatement of the vector maths the reference gets from `glam` 0.24.2
// (Cargo.lock pin; the crate's sources are NOT under /root/reference, so the
// operation ORDER below is restated from the published glam scalar-math
// implementation — parity unpinned against glam itself) plus a deterministic
// transcendental library ("stm").
//
// stm: the reference evaluates sin/cos/acos/exp/pow/atan2 through SPIR-V
// GLSL.std.450 ops whose precision is driver-defined (strolle-gpu/src/*.rs use
// `spirv_std::num_traits::Float`). To make CPU-vs-GPU parity *bit-exact*
// instead of tolerance-based, both this oracle and the HIP product evaluate
// those functions with the same FMA-free polynomial kernels (classic
// single-precision Cephes-style range reduction + minimax polynomials, ≤ ~2 ulp,
// far inside Vulkan's precision envelope). Only +,-,*,/ and sqrt (all
// correctly rounded on x86-64 SSE and on gfx950 with hipcc defaults) are used,
// and both sides are compiled with -ffp-contract=off.
#pragma once
#include <cmath>
#include <cstdint>
#include <cstring>

namespace orc {

// ---------------------------------------------------------------- bit casts
static inline uint32_t f2b(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }
static inline float b2f(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

// Rust `as u32` / `as i32` from f32: saturating, NaN -> 0.
static inline uint32_t f2u_sat(float f) {
    if (!(f > 0.0f)) return 0u;               // NaN, negatives, -0
    if (f >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)f;
}
static inline int32_t f2i_sat(float f) {
    if (f != f) return 0;
    if (f >= 2147483648.0f) return 2147483647;
    if (f <= -2147483648.0f) return (-2147483647 - 1);
    return (int32_t)f;
}

static const float F32_MAX = 3.40282347e+38f;
static const float F32_EPSILON = 1.1920929e-7f;
static const float PI = 3.14159265358979323846f;

// Rust f32::min / max (IEEE minNum / maxNum) == C fminf / fmaxf.
static inline float fmin_(float a, float b) { return fminf(a, b); }
static inline float fmax_(float a, float b) { return fmaxf(a, b); }
// Rust f32::clamp: NaN passes through.
static inline float clampf(float x, float lo, float hi) {
    if (x < lo) return lo;
    if (x > hi) return hi;
    return x;
}
static inline float saturate(float x) { return clampf(x, 0.0f, 1.0f); }  // utils/f32_ext.rs:18
static inline float sqr(float x) { return x * x; }                       // utils/f32_ext.rs:14
static inline float inverse_sqrt(float x) { return 1.0f / sqrtf(x); }    // utils/f32_ext.rs:22
static inline float signum(float x) {                                    // Rust f32::signum
    if (x != x) return x;
    return copysignf(1.0f, x);
}

// ------------------------------------------------------------------- stm_*
static inline float stm_scale2(float z, int n) {  // z * 2^n, deterministic
    if (n > 127) { z *= b2f(0x7f000000u); n -= 127; if (n > 127) n = 127; }
    if (n < -126) { z *= b2f(0x00800000u); n += 126; if (n < -126) n = -126; }
    return z * b2f((uint32_t)(n + 127) << 23);
}

static inline void stm_sincos_core(float x, float* s_out, float* c_out) {
    // |x| < 8192 expected (callers pass angles in [0, 2pi] or small)
    float ax = fabsf(x);
    int j = (int)(ax * 1.27323954473516f);  // 4/pi
    float y = (float)j;
    if (j & 1) { j += 1; y += 1.0f; }
    j &= 7;
    float r = ((ax - y * 0.78515625f) - y * 2.4187564849853515625e-4f) - y * 3.77489497744594108e-8f;
    float z = r * r;
    float ps = ((-1.9515295891e-4f * z + 8.3321608736e-3f) * z - 1.6666654611e-1f) * z * r + r;
    float pc = ((2.443315711809948e-5f * z - 1.388731625493765e-3f) * z + 4.166664568298827e-2f) * z * z - 0.5f * z + 1.0f;
    float s, c;
    switch (j) {
        case 0: s = ps; c = pc; break;
        case 2: s = pc; c = -ps; break;
        case 4: s = -ps; c = -pc; break;
        default: s = -pc; c = ps; break;  // 6
    }
    if (x < 0.0f) s = -s;
    *s_out = s; *c_out = c;
}
static inline float stm_sin(float x) { float s, c; stm_sincos_core(x, &s, &c); return s; }
static inline float stm_cos(float x) { float s, c; stm_sincos_core(x, &s, &c); return c; }

static inline float stm_asin_pos(float a) {  // a in [0,1]
    float x, z; bool flag = false;
    if (a > 0.5f) { z = 0.5f * (1.0f - a); x = sqrtf(z); flag = true; }
    else { x = a; z = x * x; }
    float p = ((((4.2163199048e-2f * z + 2.4181311049e-2f) * z + 4.5470025998e-2f) * z + 7.4953002686e-2f) * z + 1.6666752422e-1f) * z * x + x;
    if (flag) { p = p + p; p = 1.5707963267948966f - p; }
    return p;
}
static inline float stm_acos(float x) {
    if (x < -0.5f) return PI - 2.0f * stm_asin_pos(sqrtf(0.5f * (1.0f + x)));
    if (x > 0.5f) return 2.0f * stm_asin_pos(sqrtf(0.5f * (1.0f - x)));
    float a = fabsf(x);
    float s = stm_asin_pos(a);
    if (x < 0.0f) s = -s;
    return 1.5707963267948966f - s;
}

static inline float stm_exp(float x) {
    if (x != x) return x;
    if (x > 88.72283905206835f) return INFINITY;
    if (x < -103.278929903431851103f) return 0.0f;
    float z = floorf(1.44269504088896341f * x + 0.5f);
    x -= z * 0.693359375f;
    x -= z * -2.12194440e-4f;
    int n = (int)z;
    float zz = x * x;
    float p = (((((1.9875691500e-4f * x + 1.3981999507e-3f) * x + 8.3334519073e-3f) * x + 4.1665795894e-2f) * x + 1.6666665459e-1f) * x + 5.0000001201e-1f) * zz + x + 1.0f;
    return stm_scale2(p, n);
}

static inline float stm_log2(float x) {  // x > 0, finite
    uint32_t b = f2b(x);
    int e = 0;
    if ((b & 0x7f800000u) == 0) { x *= 8388608.0f; b = f2b(x); e = -23; }  // subnormal
    e += (int)((b >> 23) & 0xff) - 126;
    float m = b2f((b & 0x007fffffu) | 0x3f000000u);  // [0.5,1)
    if (m < 0.707106781186547524f) { e -= 1; m = m + m - 1.0f; } else { m = m - 1.0f; }
    float z = m * m;
    float y = ((((((((7.0376836292e-2f * m - 1.1514610310e-1f) * m + 1.1676998740e-1f) * m - 1.2420140846e-1f) * m + 1.4249322787e-1f) * m - 1.6668057665e-1f) * m + 2.0000714765e-1f) * m - 2.4999993993e-1f) * m + 3.3333331174e-1f) * m * z;
    y = y - 0.5f * z;
    float r = y * 0.44269504088896340735992f;
    r += m * 0.44269504088896340735992f;
    r += y;
    r += m;
    r += (float)e;
    return r;
}
static inline float stm_exp2(float x) {
    if (x != x) return x;
    if (x > 127.999f) return INFINITY;
    if (x < -150.0f) return 0.0f;
    float px = floorf(x);
    int i0 = (int)px;
    x = x - px;
    if (x > 0.5f) { i0 += 1; x = x - 1.0f; }
    float p = (((((1.535336188319500e-4f * x + 1.339887440266574e-3f) * x + 9.618437357674640e-3f) * x + 5.550332471162809e-2f) * x + 2.402264791363012e-1f) * x + 6.931472028550421e-1f) * x + 1.0f;
    return stm_scale2(p, i0);
}
// powf for the domains the path uses (x >= 0, finite y > 0).
static inline float stm_pow(float x, float y) {
    if (x != x || y != y) return x + y;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return 0.0f;
    if (x == INFINITY) return INFINITY;
    if (x == 1.0f) return 1.0f;
    return stm_exp2(y * stm_log2(x));
}

// Integer-exponent powf calls of the reference (`powf(2.0)`, `3.0`, `5.0`, `8.0`, `64.0`) are evaluated as exact
// multiplication chains — closer to a correctly rounded pow than exp2(y*log2(x)) and an order of magnitude cheaper
// (the SVGF edge-stopping weight alone evaluates x^64 sixteen times per pixel per pass).
static inline float stm_pow2(float x) { return x * x; }
static inline float stm_pow3(float x) { return x * x * x; }
static inline float stm_pow5(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x; }
static inline float stm_pow8(float x) { float x2 = x * x; float x4 = x2 * x2; return x4 * x4; }
static inline float stm_pow64(float x) { float x2 = x * x; float x4 = x2 * x2; float x8 = x4 * x4; float x16 = x8 * x8; float x32 = x16 * x16; return x32 * x32; }

static inline float stm_atan(float x) {
    float sign = 1.0f;
    if (x < 0.0f) { sign = -1.0f; x = -x; }
    float y;
    if (x > 2.414213562373095f) { y = 1.5707963267948966f; x = -(1.0f / x); }
    else if (x > 0.4142135623730950f) { y = 0.7853981633974483f; x = (x - 1.0f) / (x + 1.0f); }
    else y = 0.0f;
    float z = x * x;
    y += (((8.05374449538e-2f * z - 1.38776856032e-1f) * z + 1.99777106478e-1f) * z - 3.33329491539e-1f) * z * x + x;
    return sign * y;
}
static inline float stm_atan2(float y, float x) {  // Rust f32::atan2(self=y, other=x)
    if (x != x || y != y) return x + y;
    if (x == 0.0f) {
        if (y == 0.0f) return copysignf(std::signbit(x) ? PI : 0.0f, y);
        return y > 0.0f ? 1.5707963267948966f : -1.5707963267948966f;
    }
    float a = stm_atan(y / x);
    if (x < 0.0f) { a = (y < 0.0f || (y == 0.0f && std::signbit(y))) ? a - PI : a + PI; }
    return a;
}

// Rgba16Float storage (atmosphere LUTs): f32 -> f16 (round to nearest even) -> f32, in integer arithmetic so that
// CPU and GPU agree bit for bit.
// f32 -> f16 bit pattern, round to nearest even (IEEE 754-2008 binary16; what a store to an Rgba16Float texel does)
static inline uint32_t f16_bits(float f) {
    uint32_t u = f2b(f);
    uint32_t sign = (u >> 16) & 0x8000u;
    uint32_t a = u & 0x7fffffffu;
    uint32_t h;
    if (a >= 0x7f800000u) h = sign | (a > 0x7f800000u ? 0x7e00u : 0x7c00u);
    else if (a >= 0x477ff000u) h = sign | 0x7c00u;
    else if (a < 0x38800000u) {
        if (a < 0x33000000u) h = sign;
        else {
            uint32_t m = (a & 0x007fffffu) | 0x00800000u;
            uint32_t sft = 126u - (a >> 23);
            uint32_t r = m >> sft, rem = m & ((1u << sft) - 1u), half = 1u << (sft - 1u);
            if (rem > half || (rem == half && (r & 1u))) r += 1u;
            h = sign | r;
        }
    } else {
        uint32_t b = a + 0xfffu + ((a >> 13) & 1u);
        h = sign | ((b - 0x38000000u) >> 13);
    }
    return h;
}
static inline float quantize_f16(float f) {
    uint32_t h = f16_bits(f);
    uint32_t hs = (h & 0x8000u) << 16, he = (h >> 10) & 0x1fu, hm = h & 0x3ffu;
    if (he == 0u) { float v = (float)hm * 5.9604644775390625e-8f; return b2f(f2b(v) | hs); }
    if (he == 31u) return b2f(hs | 0x7f800000u | (hm << 13));
    return b2f(hs | ((he + 112u) << 23) | (hm << 13));
}

// -------------------------------------------------------------------- Vec2
struct Vec2 {
    float x, y;
    Vec2() : x(0), y(0) {}
    Vec2(float x_, float y_) : x(x_), y(y_) {}
};
static inline Vec2 operator+(Vec2 a, Vec2 b) { return Vec2(a.x + b.x, a.y + b.y); }
static inline Vec2 operator-(Vec2 a, Vec2 b) { return Vec2(a.x - b.x, a.y - b.y); }
static inline Vec2 operator*(Vec2 a, Vec2 b) { return Vec2(a.x * b.x, a.y * b.y); }
static inline Vec2 operator/(Vec2 a, Vec2 b) { return Vec2(a.x / b.x, a.y / b.y); }
static inline Vec2 operator*(Vec2 a, float s) { return Vec2(a.x * s, a.y * s); }
static inline Vec2 operator*(float s, Vec2 a) { return Vec2(s * a.x, s * a.y); }
static inline Vec2 operator/(Vec2 a, float s) { return Vec2(a.x / s, a.y / s); }
static inline Vec2 operator+(Vec2 a, float s) { return Vec2(a.x + s, a.y + s); }
static inline Vec2 operator-(Vec2 a, float s) { return Vec2(a.x - s, a.y - s); }
static inline Vec2 operator-(float s, Vec2 a) { return Vec2(s - a.x, s - a.y); }
static inline bool operator==(Vec2 a, Vec2 b) { return a.x == b.x && a.y == b.y; }
static inline float dot(Vec2 a, Vec2 b) { return a.x * b.x + a.y * b.y; }
static inline float length_squared(Vec2 a) { return dot(a, a); }
static inline Vec2 abs(Vec2 a) { return Vec2(fabsf(a.x), fabsf(a.y)); }
static inline Vec2 round(Vec2 a) { return Vec2(roundf(a.x), roundf(a.y)); }
static inline Vec2 fract_floor(Vec2 a) { return Vec2(a.x - floorf(a.x), a.y - floorf(a.y)); }  // glam Vec2::fract
static inline float fract_trunc(float a) { return a - truncf(a); }                            // Rust f32::fract

struct IVec2 {
    int32_t x, y;
    IVec2() : x(0), y(0) {}
    IVec2(int32_t x_, int32_t y_) : x(x_), y(y_) {}
};
struct UVec2 {
    uint32_t x, y;
    UVec2() : x(0), y(0) {}
    UVec2(uint32_t x_, uint32_t y_) : x(x_), y(y_) {}
};
static inline bool operator==(UVec2 a, UVec2 b) { return a.x == b.x && a.y == b.y; }
static inline bool operator==(IVec2 a, IVec2 b) { return a.x == b.x && a.y == b.y; }
static inline IVec2 operator+(IVec2 a, IVec2 b) { return IVec2(a.x + b.x, a.y + b.y); }
static inline IVec2 operator*(IVec2 a, int32_t s) { return IVec2(a.x * s, a.y * s); }
static inline Vec2 as_vec2(UVec2 a) { return Vec2((float)a.x, (float)a.y); }
static inline Vec2 as_vec2(IVec2 a) { return Vec2((float)a.x, (float)a.y); }
static inline UVec2 as_uvec2(Vec2 a) { return UVec2(f2u_sat(a.x), f2u_sat(a.y)); }
static inline IVec2 as_ivec2(Vec2 a) { return IVec2(f2i_sat(a.x), f2i_sat(a.y)); }
static inline IVec2 as_ivec2(UVec2 a) { return IVec2((int32_t)a.x, (int32_t)a.y); }
static inline UVec2 as_uvec2(IVec2 a) { return UVec2((uint32_t)a.x, (uint32_t)a.y); }

// -------------------------------------------------------------------- Vec3
struct Vec3 {
    float x, y, z;
    Vec3() : x(0), y(0), z(0) {}
    Vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    static Vec3 splat(float v) { return Vec3(v, v, v); }
    float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};
static inline Vec3 operator+(Vec3 a, Vec3 b) { return Vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline Vec3 operator-(Vec3 a, Vec3 b) { return Vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline Vec3 operator*(Vec3 a, Vec3 b) { return Vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline Vec3 operator/(Vec3 a, Vec3 b) { return Vec3(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline Vec3 operator*(Vec3 a, float s) { return Vec3(a.x * s, a.y * s, a.z * s); }
static inline Vec3 operator*(float s, Vec3 a) { return Vec3(s * a.x, s * a.y, s * a.z); }
static inline Vec3 operator/(Vec3 a, float s) { return Vec3(a.x / s, a.y / s, a.z / s); }
static inline Vec3 operator/(float s, Vec3 a) { return Vec3(s / a.x, s / a.y, s / a.z); }
static inline Vec3 operator-(Vec3 a) { return Vec3(-a.x, -a.y, -a.z); }
static inline Vec3& operator+=(Vec3& a, Vec3 b) { a = a + b; return a; }
static inline Vec3& operator*=(Vec3& a, Vec3 b) { a = a * b; return a; }
static inline Vec3& operator*=(Vec3& a, float s) { a = a * s; return a; }
static inline bool operator==(Vec3 a, Vec3 b) { return a.x == b.x && a.y == b.y && a.z == b.z; }
static inline bool operator!=(Vec3 a, Vec3 b) { return !(a == b); }
static inline float dot(Vec3 a, Vec3 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z); }
static inline Vec3 cross(Vec3 a, Vec3 b) {
    return Vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
static inline float length_squared(Vec3 a) { return dot(a, a); }
static inline float length(Vec3 a) { return sqrtf(dot(a, a)); }
static inline Vec3 normalize(Vec3 a) { return a * (1.0f / length(a)); }
static inline float distance(Vec3 a, Vec3 b) { return length(a - b); }
static inline Vec3 vmin(Vec3 a, Vec3 b) { return Vec3(fmin_(a.x, b.x), fmin_(a.y, b.y), fmin_(a.z, b.z)); }
static inline Vec3 vmax(Vec3 a, Vec3 b) { return Vec3(fmax_(a.x, b.x), fmax_(a.y, b.y), fmax_(a.z, b.z)); }
static inline Vec3 vclamp(Vec3 a, Vec3 lo, Vec3 hi) { return vmin(vmax(a, lo), hi); }
static inline Vec3 abs(Vec3 a) { return Vec3(fabsf(a.x), fabsf(a.y), fabsf(a.z)); }
static inline float max_element(Vec3 a) { return fmax_(a.x, fmax_(a.y, a.z)); }
static inline Vec3 vlerp(Vec3 a, Vec3 b, float s) { return a + ((b - a) * s); }  // glam Vec3::lerp
// glam Vec3::any_orthonormal_pair (Pixar "Building an Orthonormal Basis, Revisited")
static inline void any_orthonormal_pair(Vec3 n, Vec3* t, Vec3* b) {
    float sign = signum(n.z);
    float a = -1.0f / (sign + n.z);
    float bb = n.x * n.y * a;
    *t = Vec3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    *b = Vec3(bb, sign + n.y * n.y * a, -n.y);
}
// glam 0.24 math::acos_approx (DirectXMath XMScalarAcos), used by Vec3::angle_between
static inline float glam_acos_approx(float v) {
    bool nonnegative = v >= 0.0f;
    float x = fabsf(v);
    float omx = 1.0f - x;
    if (omx < 0.0f) omx = 0.0f;
    float root = sqrtf(omx);
    float r = ((((((-0.0012624911f * x + 0.0066700901f) * x - 0.0170881256f) * x + 0.0308918810f) * x - 0.0501743046f) * x + 0.0889789874f) * x - 0.2145988016f) * x + 1.5707963050f;
    r *= root;
    return nonnegative ? r : PI - r;
}
static inline float angle_between(Vec3 a, Vec3 b) {
    return glam_acos_approx(dot(a, b) / sqrtf(length_squared(a) * length_squared(b)));
}

// -------------------------------------------------------------------- Vec4
struct Vec4 {
    float x, y, z, w;
    Vec4() : x(0), y(0), z(0), w(0) {}
    Vec4(float x_, float y_, float z_, float w_) : x(x_), y(y_), z(z_), w(w_) {}
    Vec4(Vec3 v, float w_) : x(v.x), y(v.y), z(v.z), w(w_) {}
    Vec3 xyz() const { return Vec3(x, y, z); }
    Vec2 xy() const { return Vec2(x, y); }
    Vec2 zw() const { return Vec2(z, w); }
    Vec2 yz() const { return Vec2(y, z); }
};
static inline Vec4 operator+(Vec4 a, Vec4 b) { return Vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
static inline Vec4 operator-(Vec4 a, Vec4 b) { return Vec4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }
static inline Vec4 operator*(Vec4 a, Vec4 b) { return Vec4(a.x * b.x, a.y * b.y, a.z * b.z, a.w * b.w); }
static inline Vec4 operator*(Vec4 a, float s) { return Vec4(a.x * s, a.y * s, a.z * s, a.w * s); }
static inline Vec4 operator/(Vec4 a, float s) { return Vec4(a.x / s, a.y / s, a.z / s, a.w / s); }
static inline bool operator==(Vec4 a, Vec4 b) { return a.x == b.x && a.y == b.y && a.z == b.z && a.w == b.w; }
static inline float dot(Vec4 a, Vec4 b) { return (a.x * b.x) + (a.y * b.y) + (a.z * b.z) + (a.w * b.w); }

// -------------------------------------------------------------------- Mat4
struct Mat4 {  // column-major, like glam
    Vec4 c[4];
    static Mat4 identity() {
        Mat4 m;
        m.c[0] = Vec4(1, 0, 0, 0); m.c[1] = Vec4(0, 1, 0, 0); m.c[2] = Vec4(0, 0, 1, 0); m.c[3] = Vec4(0, 0, 0, 1);
        return m;
    }
    static Mat4 from_cols_array(const float* a) {
        Mat4 m;
        for (int i = 0; i < 4; i++) m.c[i] = Vec4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]);
        return m;
    }
};
static inline Vec4 mul(const Mat4& m, Vec4 v) {  // glam Mat4::mul_vec4
    Vec4 r = m.c[0] * v.x;
    r = r + m.c[1] * v.y;
    r = r + m.c[2] * v.z;
    r = r + m.c[3] * v.w;
    return r;
}
static inline Mat4 mul(const Mat4& a, const Mat4& b) {
    Mat4 r;
    for (int i = 0; i < 4; i++) r.c[i] = mul(a, b.c[i]);
    return r;
}
static inline Vec3 project_point3(const Mat4& m, Vec3 p) {  // glam Mat4::project_point3
    Vec4 r = m.c[0] * p.x;
    r = m.c[1] * p.y + r;
    r = m.c[2] * p.z + r;
    r = m.c[3] + r;
    float rw = 1.0f / r.w;
    return Vec3(r.x * rw, r.y * rw, r.z * rw);
}
static inline Vec3 transform_vector3(const Mat4& m, Vec3 v) {  // glam Mat4::transform_vector3
    Vec4 r = m.c[0] * v.x;
    r = m.c[1] * v.y + r;
    r = m.c[2] * v.z + r;
    return r.xyz();
}
static inline Mat4 transpose(const Mat4& m) {
    Mat4 r;
    r.c[0] = Vec4(m.c[0].x, m.c[1].x, m.c[2].x, m.c[3].x);
    r.c[1] = Vec4(m.c[0].y, m.c[1].y, m.c[2].y, m.c[3].y);
    r.c[2] = Vec4(m.c[0].z, m.c[1].z, m.c[2].z, m.c[3].z);
    r.c[3] = Vec4(m.c[0].w, m.c[1].w, m.c[2].w, m.c[3].w);
    return r;
}
// glam 0.24 Mat4::inverse (scalar path; cofactor expansion as in GLM)
static inline Mat4 inverse(const Mat4& m) {
    float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z, m03 = m.c[0].w;
    float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z, m13 = m.c[1].w;
    float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z, m23 = m.c[2].w;
    float m30 = m.c[3].x, m31 = m.c[3].y, m32 = m.c[3].z, m33 = m.c[3].w;
    float coef00 = m22 * m33 - m32 * m23, coef02 = m12 * m33 - m32 * m13, coef03 = m12 * m23 - m22 * m13;
    float coef04 = m21 * m33 - m31 * m23, coef06 = m11 * m33 - m31 * m13, coef07 = m11 * m23 - m21 * m13;
    float coef08 = m21 * m32 - m31 * m22, coef10 = m11 * m32 - m31 * m12, coef11 = m11 * m22 - m21 * m12;
    float coef12 = m20 * m33 - m30 * m23, coef14 = m10 * m33 - m30 * m13, coef15 = m10 * m23 - m20 * m13;
    float coef16 = m20 * m32 - m30 * m22, coef18 = m10 * m32 - m30 * m12, coef19 = m10 * m22 - m20 * m12;
    float coef20 = m20 * m31 - m30 * m21, coef22 = m10 * m31 - m30 * m11, coef23 = m10 * m21 - m20 * m11;
    Vec4 fac0(coef00, coef00, coef02, coef03), fac1(coef04, coef04, coef06, coef07), fac2(coef08, coef08, coef10, coef11);
    Vec4 fac3(coef12, coef12, coef14, coef15), fac4(coef16, coef16, coef18, coef19), fac5(coef20, coef20, coef22, coef23);
    Vec4 vec0(m10, m00, m00, m00), vec1(m11, m01, m01, m01), vec2(m12, m02, m02, m02), vec3(m13, m03, m03, m03);
    Vec4 inv0 = (vec1 * fac0 - vec2 * fac1) + vec3 * fac2;
    Vec4 inv1 = (vec0 * fac0 - vec2 * fac3) + vec3 * fac4;
    Vec4 inv2 = (vec0 * fac1 - vec1 * fac3) + vec3 * fac5;
    Vec4 inv3 = (vec0 * fac2 - vec1 * fac4) + vec2 * fac5;
    Vec4 sign_a(1.0f, -1.0f, 1.0f, -1.0f), sign_b(-1.0f, 1.0f, -1.0f, 1.0f);
    Mat4 inv;
    inv.c[0] = inv0 * sign_a; inv.c[1] = inv1 * sign_b; inv.c[2] = inv2 * sign_a; inv.c[3] = inv3 * sign_b;
    Vec4 col0(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    Vec4 dot0 = m.c[0] * col0;
    float dot1 = dot0.x + dot0.y + dot0.z + dot0.w;
    float rcp_det = 1.0f / dot1;
    for (int i = 0; i < 4; i++) inv.c[i] = inv.c[i] * rcp_det;
    return inv;
}

// ---------------------------------------------------------------- Affine3A
struct Affine3 {  // glam Affine3A: matrix3 columns + translation
    Vec3 x_axis, y_axis, z_axis, translation;
    static Affine3 from_12(const float* a) {  // column-major 3x4: x_axis, y_axis, z_axis, translation
        Affine3 r;
        r.x_axis = Vec3(a[0], a[1], a[2]); r.y_axis = Vec3(a[3], a[4], a[5]);
        r.z_axis = Vec3(a[6], a[7], a[8]); r.translation = Vec3(a[9], a[10], a[11]);
        return r;
    }
};
static inline Vec3 mat3_mul(const Affine3& a, Vec3 v) {
    Vec3 r = a.x_axis * v.x;
    r = r + a.y_axis * v.y;
    r = r + a.z_axis * v.z;
    return r;
}
static inline Vec3 transform_point3(const Affine3& a, Vec3 p) {
    return ((a.x_axis * p.x) + (a.y_axis * p.y) + (a.z_axis * p.z)) + a.translation;
}
static inline float mat3_determinant(const Affine3& a) { return dot(a.z_axis, cross(a.x_axis, a.y_axis)); }
static inline Affine3 inverse(const Affine3& a) {  // glam Affine3A::inverse / Mat3A::inverse
    Vec3 tmp0 = cross(a.y_axis, a.z_axis);
    Vec3 tmp1 = cross(a.z_axis, a.x_axis);
    Vec3 tmp2 = cross(a.x_axis, a.y_axis);
    float det = dot(a.z_axis, tmp2);
    float inv_det = 1.0f / det;
    Vec3 c0 = tmp0 * inv_det, c1 = tmp1 * inv_det, c2 = tmp2 * inv_det;
    Affine3 r;  // transpose
    r.x_axis = Vec3(c0.x, c1.x, c2.x);
    r.y_axis = Vec3(c0.y, c1.y, c2.y);
    r.z_axis = Vec3(c0.z, c1.z, c2.z);
    r.translation = -mat3_mul(r, a.translation);
    return r;
}
static inline Mat4 mat4_from_affine(const Affine3& a) {
    Mat4 m;
    m.c[0] = Vec4(a.x_axis, 0.0f); m.c[1] = Vec4(a.y_axis, 0.0f); m.c[2] = Vec4(a.z_axis, 0.0f); m.c[3] = Vec4(a.translation, 1.0f);
    return m;
}

}  // namespace orc
