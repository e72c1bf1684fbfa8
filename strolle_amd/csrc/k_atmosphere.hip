// k_atmosphere.hip — sky look-up tables: transmittance (256x64, once), multi-scattering (32x32, once) and the sky view
// (256x256, whenever the sun's altitude changes). Behavioural contract: strolle-shaders/src/atmosphere/*.rs and
// strolle/src/camera_controller/passes/atmosphere.rs. The reference stores these as Rgba16Float textures; here they
// are float4 arrays whose texels are rounded through f16 on store (quantize_f16), sampled with the manual bilinear
// filter of st_device.h. Off the per-frame path: one thread per texel, no tiling tricks.
#include "k_common.h"

namespace st {
namespace ST_KNS {

ST_D float quantize_f16(float f) {
    const uint32_t h = f16_bits(f);
    const uint32_t hs = (h & 0x8000u) << 16, he = (h >> 10) & 0x1fu, hm = h & 0x3ffu;
    if (he == 0u) { const float v = (float)hm * 5.9604644775390625e-8f; return b2f(f2b(v) | hs); }
    if (he == 31u) return b2f(hs | 0x7f800000u | (hm << 13));
    return b2f(hs | ((he + 112u) << 23) | (hm << 13));
}
ST_D float4 store_f16(V3 v) { return make_float4(quantize_f16(v.x), quantize_f16(v.y), quantize_f16(v.z), quantize_f16(1.0f)); }
ST_D V3 exp3(V3 v) { return v3(exp_(v.x), exp_(v.y), exp_(v.z)); }

struct ScatteringTerms { V3 rayleigh; float mie; V3 extinction; };
ST_D ScatteringTerms eval_scattering(V3 pos) {  // atmosphere/utils.rs:3-29
    const float altitude_km = (length(pos) - kGroundRadiusMm) * 1000.0f;
    const float rayleigh_density = exp_(-altitude_km / 8.0f);
    const float mie_density = exp_(-altitude_km / 1.2f);
    ScatteringTerms t;
    t.rayleigh = v3(5.802f, 13.558f, 33.1f) * rayleigh_density;
    const float rayleigh_absorption = 0.0f;  // RAYLEIGH_ABSORPTION_BASE (0.0) * density, folded: 0 * inf must not poison the LUT (DESIGN.md deviation 9)
    t.mie = 3.996f * mie_density;
    const float mie_absorption = 4.4f * mie_density;
    const V3 ozone_absorption = v3(0.650f, 1.881f, 0.085f) * fmax_(1.0f - fabsf(altitude_km - 25.0f) / 15.0f, 0.0f);
    t.extinction = t.rayleigh + v3s(rayleigh_absorption) + v3s(t.mie) + v3s(mie_absorption) + ozone_absorption;
    return t;
}
ST_D float eval_mie_phase(float cos_theta) {
    const float G = 0.8f;
    const float SCALE = 3.0f / (8.0f * kPi);
    const float num = (1.0f - G * G) * (1.0f + cos_theta * cos_theta);
    const float denom = (2.0f + G * G) * pow_(1.0f + G * G - 2.0f * G * cos_theta, 1.5f);
    return SCALE * num / denom;
}
ST_D float eval_rayleigh_phase(float cos_theta) { const float K = 3.0f / (16.0f * kPi); return K * (1.0f + cos_theta * cos_theta); }
ST_D V3 sample_sun_lut(const float4* lut, int32_t w, int32_t h, V3 pos, V3 sun_dir) {  // Atmosphere::sample_lut (atmosphere.rs:179-204)
    const float height = length(pos);
    const V3 up = pos / height;
    const float sun_cos_zenith_angle = dot(sun_dir, up);
    const float u = saturate(0.5f + 0.5f * sun_cos_zenith_angle);
    const float v = saturate((height - kGroundRadiusMm) / (kAtmosphereRadiusMm - kGroundRadiusMm));
    return xyz(lut_sample(lut, w, h, v2(u, v)));
}
ST_D float sphere_hit(V3 origin, V3 dir, float radius) { return intersect_sphere(make_ray(origin, dir), radius); }

// generate_transmittance_lut.rs:5-59
__global__ void k_atmosphere_transmittance(float4* out) {
    const uint32_t x = blockIdx.x * 8u + (threadIdx.x & 7u), y = blockIdx.y * 8u + (threadIdx.x >> 3);
    if (x >= 256u || y >= 64u) return;
    const V2 uv = v2((float)x, (float)y) / v2(256.0f, 64.0f);
    const float sun_cos_theta = 2.0f * uv.x - 1.0f;
    const float sun_theta = acos_(clampf(sun_cos_theta, -1.0f, 1.0f));
    const float height = lerpf(kGroundRadiusMm, kAtmosphereRadiusMm, uv.y);
    const V3 pos = v3(0.0f, height, 0.0f);
    const V3 sun_dir = normalize(v3(0.0f, sun_cos_theta, -sin_(sun_theta)));
    V3 transmittance;
    if (sphere_hit(pos, sun_dir, kGroundRadiusMm) > 0.0f) transmittance = v3s(0.0f);
    else {
        const float atmosphere_distance = sphere_hit(pos, sun_dir, kAtmosphereRadiusMm);
        float t = 0.0f, i = 0.0f;
        transmittance = v3s(1.0f);
        while (i < 40.0f) {
            const float new_t = ((i + 0.3f) / 40.0f) * atmosphere_distance;
            const float dt = new_t - t;
            t = new_t;
            const ScatteringTerms sc = eval_scattering(pos + t * sun_dir);
            transmittance = transmittance * exp3(-dt * sc.extinction);
            i += 1.0f;
        }
    }
    out[y * 256u + x] = store_f16(transmittance);
}

// generate_scattering_lut.rs:5-170
__global__ void k_atmosphere_scattering(const float4* transmittance_lut, float4* out) {
    const uint32_t x = blockIdx.x * 8u + (threadIdx.x & 7u), y = blockIdx.y * 8u + (threadIdx.x >> 3);
    if (x >= 32u || y >= 32u) return;
    const V2 uv = v2((float)x, (float)y) / v2(32.0f, 32.0f);
    const float sun_cos_theta = 2.0f * uv.x - 1.0f;
    const float sun_theta = acos_(clampf(sun_cos_theta, -1.0f, 1.0f));
    const float height = lerpf(kGroundRadiusMm, kAtmosphereRadiusMm, fmax_(uv.y, 0.01f));
    const V3 pos = v3(0.0f, height, 0.0f);
    const V3 sun_dir = normalize(v3(0.0f, sun_cos_theta, -sin_(sun_theta)));
    V3 lum_total = v3s(0.0f), fms = v3s(0.0f);
    const float inv_samples = 1.0f / (float)(8 * 8);
    for (int i = 0; i < 8; i++)
        for (int j = 0; j < 8; j++) {
            const float theta = kPi * ((float)i + 0.5f) / 8.0f;
            const float phi = acos_(clampf(1.0f - 2.0f * ((float)j + 0.5f) / 8.0f, -1.0f, 1.0f));
            float sp, cp, st_, ct;
            sincos_(phi, &sp, &cp); sincos_(theta, &st_, &ct);
            const V3 ray_dir = v3(sp * st_, cp, sp * ct);
            const float atmosphere_distance = sphere_hit(pos, ray_dir, kAtmosphereRadiusMm);
            const float ground_distance = sphere_hit(pos, ray_dir, kGroundRadiusMm);
            const float t_max = ground_distance > 0.0f ? ground_distance : atmosphere_distance;
            const float cos_theta = dot(ray_dir, sun_dir);
            const float mie_phase_value = eval_mie_phase(cos_theta);
            const float rayleigh_phase_value = eval_rayleigh_phase(-cos_theta);
            V3 lum = v3s(0.0f), lum_factor = v3s(0.0f), transmittance = v3s(1.0f);
            float t = 0.0f, step_i = 0.0f;
            while (step_i < 20.0f) {
                const float new_t = ((step_i + 0.3f) / 20.0f) * t_max;
                const float dt = new_t - t;
                t = new_t;
                const V3 new_pos = pos + t * ray_dir;
                const ScatteringTerms sc = eval_scattering(new_pos);
                const V3 sample_transmittance = exp3(-dt * sc.extinction);
                const V3 scattering_no_phase = sc.rayleigh + v3s(sc.mie);
                const V3 scattering_f = (scattering_no_phase - scattering_no_phase * sample_transmittance) / sc.extinction;
                lum_factor = lum_factor + transmittance * scattering_f;
                const V3 sun_transmittance = sample_sun_lut(transmittance_lut, 256, 64, new_pos, sun_dir);
                const V3 rayleigh_in = sc.rayleigh * rayleigh_phase_value;
                const float mie_in = sc.mie * mie_phase_value;
                const V3 in_scattering = (rayleigh_in + v3s(mie_in)) * sun_transmittance;
                const V3 scattering_integral = (in_scattering - in_scattering * sample_transmittance) / sc.extinction;
                lum = lum + scattering_integral * transmittance;
                transmittance = transmittance * sample_transmittance;
                step_i += 1.0f;
            }
            if (ground_distance > 0.0f) {
                V3 hit_pos = pos + ground_distance * ray_dir;
                if (dot(pos, sun_dir) > 0.0f) {
                    hit_pos = normalize(hit_pos) * kGroundRadiusMm;
                    lum = lum + transmittance * v3s(0.25f) * sample_sun_lut(transmittance_lut, 256, 64, hit_pos, sun_dir);
                }
            }
            fms = fms + lum_factor * inv_samples;
            lum_total = lum_total + lum * inv_samples;
        }
    out[y * 32u + x] = store_f16(lum_total / (v3s(1.0f) - fms));
}

// generate_sky_lut.rs:5-158
__global__ void k_atmosphere_sky(const float4* transmittance_lut, const float4* scattering_lut, float sun_altitude, float4* out) {
    const uint32_t x = blockIdx.x * 8u + (threadIdx.x & 7u), y = blockIdx.y * 8u + (threadIdx.x >> 3);
    if (x >= 256u || y >= 256u) return;
    const V2 uv = v2((float)x, (float)y) / v2(256.0f, 256.0f);
    const float azimuth = (uv.x - 0.5f) * 2.0f * kPi;
    float v;
    if (uv.y < 0.5f) { const float coord = 1.0f - 2.0f * uv.y; v = -coord * coord; }
    else { const float coord = uv.y * 2.0f - 1.0f; v = coord * coord; }
    const V3 pos = v3(0.0f, kGroundRadiusMm + 0.0002f, 0.0f);
    const float height = length(pos);
    float th = sqr(height) - sqr(kGroundRadiusMm);
    th = sqrtf(th) / height;
    const float horizon = acos_(clampf(th, -1.0f, 1.0f)) - 0.5f * kPi;
    const float altitude = v * 0.5f * kPi - horizon;
    float sa_, ca_, sz_, cz_;
    sincos_(altitude, &sa_, &ca_); sincos_(azimuth, &sz_, &cz_);
    const V3 ray_dir = v3(ca_ * sz_, sa_, -ca_ * cz_);
    const float sa = fmodf(sun_altitude, 2.0f * kPi);
    float ss, sc_;
    sincos_(sa, &ss, &sc_);
    const V3 sun_dir = sa < 0.5f * kPi ? v3(0.0f, ss, -sc_) : v3(0.0f, ss, sc_);
    const float atmosphere_distance = sphere_hit(pos, ray_dir, kAtmosphereRadiusMm);
    const float ground_distance = sphere_hit(pos, ray_dir, kGroundRadiusMm);
    const float t_max = ground_distance < 0.0f ? atmosphere_distance : ground_distance;
    const float cos_theta = dot(ray_dir, sun_dir);
    const float mie_phase_value = eval_mie_phase(cos_theta);
    const float rayleigh_phase_value = eval_rayleigh_phase(-cos_theta);
    V3 lum = v3s(0.0f), transmittance = v3s(1.0f);
    float t = 0.0f, i = 0.0f;
    while (i < 32.0f) {
        const float new_t = ((i + 0.3f) / 32.0f) * t_max;
        const float dt = new_t - t;
        t = new_t;
        const V3 new_pos = pos + t * ray_dir;
        const ScatteringTerms sc = eval_scattering(new_pos);
        const V3 sample_transmittance = exp3(-dt * sc.extinction);
        const V3 sun_transmittance = sample_sun_lut(transmittance_lut, 256, 64, new_pos, sun_dir);
        const V3 psi_ms = sample_sun_lut(scattering_lut, 32, 32, new_pos, sun_dir);
        const V3 rayleigh_in = sc.rayleigh * (rayleigh_phase_value * sun_transmittance + psi_ms);
        const V3 mie_in = sc.mie * (mie_phase_value * sun_transmittance + psi_ms);
        const V3 in_scattering = rayleigh_in + mie_in;
        const V3 scattering_integral = (in_scattering - in_scattering * sample_transmittance) / sc.extinction;
        lum = lum + scattering_integral * transmittance;
        transmittance = transmittance * sample_transmittance;
        i += 1.0f;
    }
    out[y * 256u + x] = store_f16(lum);
}

void launch_atmosphere_static(float4* transmittance_lut, float4* scattering_lut, hipStream_t s) {
    ST_KLAUNCH(k_atmosphere_transmittance, dim3(32, 8), dim3(64), s, transmittance_lut);
    ST_KLAUNCH(k_atmosphere_scattering, dim3(4, 4), dim3(64), s, transmittance_lut, scattering_lut);
}
void launch_atmosphere_sky(const float4* transmittance_lut, const float4* scattering_lut, float sun_altitude, float4* sky_lut, hipStream_t s) {
    ST_KLAUNCH(k_atmosphere_sky, dim3(32, 32), dim3(64), s, transmittance_lut, scattering_lut, sun_altitude, sky_lut);
}

}  // namespace ST_KNS
}  // namespace st
