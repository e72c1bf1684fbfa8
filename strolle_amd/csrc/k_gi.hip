// k_gi.hip — ReSTIR GI: reprojection, sampling (trace + shade), temporal / spatial / preview resampling, resolving.
// Behavioural contract: strolle-shaders/src/gi_*.rs. Reservoir roles: reprojection [0] -> [2]; sampling_b -> [1];
// temporal reads [2], RMW [1]; spatial [1] -> [2]; preview ([1]|[2]) -> [3] -> [0]; resolving reads [0], then copies
// the frame's source ([1]|[2]) over it (strolle/src/camera_controller/passes/gi_*.rs).
#include "k_common.h"

namespace st {
namespace ST_KNS {

// ---------------------------------------------------------------- gi_reprojection.rs:3-51
__global__ ST_KERNEL_BOUNDS void k_gi_reprojection(const KArgs a) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t n = a.width * a.height;
    const Hit hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    const bool some = hit_some(hit);
    GiReservoir res = gi_empty();
    if (some) {
        const Reprojection rp = reprojection_read(tex_read(a.reprojection, a, pos));
        if (rp.confidence > 0.0f) res = gi_read(a.gi_res[0], screen_to_idx(a, reprojection_prev_round(rp)), n);
        res.confidence = 1.0f;
        res.s.v1_point = hit.point;
    }
    gi_write_own(a.gi_res[2], screen_to_idx(a, pos), res, true, some);  // a pixel without a surface leaves its slot alone
}
void launch_gi_reprojection(const KArgs& a, hipStream_t s) { ST_LAUNCH(k_gi_reprojection, false, s, a); }

ST_D bool frame_is_gi_tracing(uint32_t frame) { return frame % 6u < 4u; }  // frame.rs:19-21

// ---------------------------------------------------------------- gi_sampling_a.rs:3-122
// The two sampling passes as per-cell bodies, so that they can run as the reference's two launches or as one (k_gi_sampling_ab).
// `prim_hit`: pixel_hit() of the cell's pixel (tracing frames: pass a samples its BRDF; pass b needs it on every frame);
// `vres`: the reprojected reservoir of validation frames (both passes re-trace / re-shade its sample).
// Pass a: what it stores in gi_d0..2 — false when the pass leaves early (nothing stored, stale texels stay, as in the reference).
// ... in two halves — the cell's bounce ray, and what the pass stores for its hit — so that the trace between them can be somebody else's (round 6's lane-refill
// pool, measured and archived: tools/experiments/gi_sampling_pool.inc)
ST_D bool gi_sampling_a_ray(uint32_t seed, bool tracing, U2 pos, const Hit& prim_hit, const GiReservoir& vres, Ray* gi_ray, float* gi_ray_pdf) {
    if (tracing) {
        WhiteNoise wn = white_noise(seed, pos);
        if (!hit_some(prim_hit)) return false;
        const BrdfSample s = layered_brdf_sample(prim_hit.g, wn, -prim_hit.dir);
        *gi_ray = make_ray(prim_hit.point, s.dir);
        *gi_ray_pdf = s.pdf;
    } else {
        if (vres.m == 0.0f) return false;
        *gi_ray = make_ray(vres.s.v1_point, gi_dir(vres.s, vres.s.v1_point));
        *gi_ray_pdf = 1.0f;
    }
    return true;
}
ST_D void gi_sampling_a_shade(const KArgs& a, const Ray& gi_ray, float gi_ray_pdf, const TriangleHit& gi_hit, float4* d0, float4* d1, float4* d2);
template <class SE>
ST_D bool gi_sampling_a_cell(const KArgs& a, uint32_t seed, bool tracing, U2 pos, const Hit& prim_hit, const GiReservoir& vres, SE* stack, uint32_t* used_,
                             float4* d0, float4* d1, float4* d2) {
    Ray gi_ray; float gi_ray_pdf;
    if (!gi_sampling_a_ray(seed, tracing, pos, prim_hit, vres, &gi_ray, &gi_ray_pdf)) return false;
    const TriangleHit gi_hit = trace_closest(a, gi_ray, stack, used_);
    gi_sampling_a_shade(a, gi_ray, gi_ray_pdf, gi_hit, d0, d1, d2);
    return true;
}
ST_D void gi_sampling_a_shade(const KArgs& a, const Ray& gi_ray, float gi_ray_pdf, const TriangleHit& gi_hit, float4* d0, float4* d1, float4* d2) {
    GBuffer gg = gbuffer_zero();
    uint32_t base_bits = 0u;  // gbuffer_pack_base_color of the zero colour
    if (hit_is_some(gi_hit)) {
        GpuMaterial m = a.materials[gi_hit.material_id];
        m.roughness = fmax_(m.roughness, 0.75f * 0.75f);
        gg.base_color = sample_atlas(a, gi_hit.uv, m.base_color, m.base_color_texture);
        base_bits = (a.material_base_packed && is_zero(m.base_color_texture)) ? a.material_base_packed[gi_hit.material_id] : gbuffer_pack_base_color(gg.base_color);
        gg.normal = gi_hit.normal; gg.metallic = m.metallic;
        gg.emissive = xyz(sample_atlas(a, gi_hit.uv, m.emissive, m.emissive_texture));
        gg.roughness = m.roughness; gg.reflectance = m.reflectance;
        gg.depth = distance(gi_ray.origin, gi_hit.point);
    }
    *d0 = f4(gi_ray.dir, gi_ray_pdf);
    gbuffer_pack_bits(gg, base_bits, d1, d2);
}
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_gi_sampling_a(const KArgs a_in, uint32_t seed) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const bool tracing = frame_is_gi_tracing(a.frame);
    const U2 pos = tracing ? resolve_checkerboard(gid, a.frame / 2u) : resolve_checkerboard(gid, a.frame);
    if (!owns_pixel(a, pos)) return;
    const Hit prim_hit = tracing ? pixel_hit(a, a.cam, a.g0, a.g1, pos) : hit_zero();
    const GiReservoir vres = tracing ? gi_empty() : gi_read(a.gi_res[2], screen_to_idx(a, pos), a.width * a.height);
    float4 d0, d1, d2;
    if (!gi_sampling_a_cell(a, seed, tracing, pos, prim_hit, vres, lane_stack(a, lds), &used_, &d0, &d1, &d2)) return;
    count_rays(a, used_);
    tex_write(a.gi_d0, a, gid, d0);  // indexed by the half-resolution gid (gi_sampling_a.rs:117-121)
    tex_write(a.gi_d1, a, gid, d1);
    tex_write(a.gi_d2, a, gid, d2);
}
void launch_gi_sampling_a(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH_TRACE(k_gi_sampling_a, true, s, a, seed); }

// ---------------------------------------------------------------- gi_sampling_b.rs:3-235
// d0..2: pass a's texels for this cell. The caller has checked hit_some(prim_hit) and, on validation frames, vres.m != 0.
template <class SE>
ST_D void gi_sampling_b_cell(const KArgs& a, uint32_t seed, bool tracing, U2 pos, const Hit& prim_hit, const GiReservoir& vres, SE* stack, uint32_t* used_,
                             float4 d0, float4 d1, float4 d2) {
    const uint32_t idx = screen_to_idx(a, pos);
    WhiteNoise wn; Hit gi_hit; float gi_ray_pdf;
    if (tracing) {
        wn = white_noise(seed, pos);
        gi_hit = hit_make(make_ray(prim_hit.point, xyz(d0)), gbuffer_unpack(a, d1, d2));
        gi_ray_pdf = d0.w;
    } else {
        wn.state = vres.s.rng;
        gi_hit = hit_make(make_ray(vres.s.v1_point, xyz(d0)), gbuffer_unpack(a, d1, d2));
        gi_ray_pdf = 1.0f;
    }
    const uint32_t rng = wn.state;
    uint32_t light_id; float light_pdf; V3 light_rad; V3 light_dir = v3s(0.0f);
    if (!hit_some(gi_hit)) {
        light_id = kLightIdSky; light_pdf = 1.0f;
        light_rad = atmosphere_sample(a, gi_hit.dir);
    } else {
        const float atmosphere_pdf = a.sun_altitude <= -1.0f ? 0.0f : 0.25f;
        bool pick_sky = a.light_count == 0u;
        if (!pick_sky) pick_sky = wn.sample() < atmosphere_pdf;  // `||` short-circuits: no sample is drawn when light_count == 0
        if (pick_sky) {
            light_id = kLightIdSky; light_pdf = atmosphere_pdf;
            light_dir = wn.sample_hemisphere(gi_hit.g.normal);
            light_rad = atmosphere_sample(a, light_dir) * dot(gi_hit.g.normal, light_dir);
        } else {
            const EphemeralResult res = ephemeral_build(a, wn, gi_hit);
            if (res.w > 0.0f) {
                light_id = res.light_id;
                light_pdf = frcp(res.w) * (1.0f - atmosphere_pdf);
                light_rad = res.light_rad.radiance * (v3s(1.0f) + res.light_rad.spec_brdf);
            } else { light_id = 0u; light_pdf = 1.0f; light_rad = v3s(0.0f); }
        }
    }
    V3 radiance;
    if (light_pdf > 0.0f) {
        float light_vis;
        if (hit_some(gi_hit)) {
            const Ray ray = light_id == kLightIdSky ? make_ray(gi_hit.point, light_dir) : light_ray_wnoise(light_get(a, light_id), wn, gi_hit.point);
            uint32_t used_now = 0u;
            const bool occluded = trace_any(a, ray, stack, &used_now);
            *used_ += used_now;
            count_rays(a, used_now);
            light_vis = occluded ? 0.0f : 1.0f;
        } else light_vis = 1.0f;
        radiance = light_rad * light_vis / light_pdf;
    } else radiance = v3s(0.0f);
    if (hit_some(gi_hit)) { radiance = radiance * divc3(xyz(gi_hit.g.base_color), kPi); radiance = radiance + gi_hit.g.emissive; }
    GiReservoir res = gi_empty();
    if (gi_ray_pdf > 0.0f) {
        const V3 v1 = prim_hit.point;
        V3 v2p, v2n;
        if (hit_some(gi_hit)) { v2p = gi_hit.point; v2n = gi_hit.g.normal; }
        else { v2p = v1 + gi_hit.dir * 1000.0f; v2n = -gi_hit.dir; }  // World::SUN_DISTANCE
        res.s.pdf = 0.0f; res.s.rng = rng; res.s.radiance = radiance; res.s.v1_point = v1; res.s.v2_point = v2p; res.s.v2_normal = v2n;
        res.m = 1.0f; res.w = frcp(gi_ray_pdf);
        res.s.pdf = gi_pdf(res.s, prim_hit);
    }
    gi_write(a.gi_res[1], idx, res);
}
template <bool LDS_SCENE, class SE>
__global__ ST_KERNEL_BOUNDS void k_gi_sampling_b(const KArgs a_in, uint32_t seed) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const bool tracing = frame_is_gi_tracing(a.frame);
    const U2 pos = tracing ? resolve_checkerboard(gid, a.frame / 2u) : resolve_checkerboard(gid, a.frame);
    if (!owns_pixel(a, pos)) return;
    const Hit prim_hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    if (!hit_some(prim_hit)) return;
    const float4 d0 = tex_read(a.gi_d0, a, gid), d1 = tex_read(a.gi_d1, a, gid), d2 = tex_read(a.gi_d2, a, gid);
    GiReservoir vres = gi_empty();
    if (!tracing) { vres = gi_read(a.gi_res[2], screen_to_idx(a, pos), a.width * a.height); if (vres.m == 0.0f) return; }
    gi_sampling_b_cell(a, seed, tracing, pos, prim_hit, vres, lane_stack(a, lds), &used_, d0, d1, d2);
}
void launch_gi_sampling_b(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH_TRACE(k_gi_sampling_b, true, s, a, seed); }

// Both sampling passes of a cell in one launch: pass b takes pass a's three texels from registers (they are still stored:
// gi_d0..2 are planes of the reference), the pixel's G-buffer and, on validation frames, the reprojected reservoir are read once.
// Wherever pass b runs pass a has run (tracing: both need the pixel's surface; validation: both need a non-empty reservoir).
// 6 waves per SIMD: the allocator fits 80 VGPRs without a spill where it would take 83 (5 waves): 96.5 -> 93.2 us, dungeon 356 -> 345.
template <bool LDS_SCENE, class SE>
__global__ __launch_bounds__(kBlockThreads, 6) void k_gi_sampling_ab(const KArgs a_in, uint32_t seed_a, uint32_t seed_b, uint32_t reproject) {
    ST_SCENE_PROLOGUE
    ST_STACK_LDS(SE, lds);
    uint32_t used_ = 0u;
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const bool tracing = frame_is_gi_tracing(a.frame);
    const U2 pos = tracing ? resolve_checkerboard(gid, a.frame / 2u) : resolve_checkerboard(gid, a.frame);
    if (!owns_pixel(a, pos)) return;
    const Hit prim_hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    GiReservoir vres = gi_empty();
    if (!tracing) {
        const uint32_t n = a.width * a.height;
        if (reproject && hit_some(prim_hit)) {
            // validation frames of a whole frame: gi_reprojection.rs for this pixel runs here (and again in k_gi_temporal<true>, which
            // stores it) instead of as a launch of its own — what that pass would have stored, through the store / load codec
            GiReservoir r = gi_empty();
            const Reprojection rp = reprojection_read(tex_read(a.reprojection, a, pos));
            if (rp.confidence > 0.0f) r = gi_read(a.gi_res[0], screen_to_idx(a, reprojection_prev_round(rp)), n);
            r.confidence = 1.0f;
            r.s.v1_point = prim_hit.point;
            vres = gi_after_store(r);
        } else vres = gi_read(a.gi_res[2], screen_to_idx(a, pos), n);  // a pixel without a surface: the slot gi_reprojection leaves alone
    }
    float4 d0, d1, d2;
    if (!gi_sampling_a_cell(a, seed_a, tracing, pos, prim_hit, vres, lane_stack(a, lds), &used_, &d0, &d1, &d2)) return;
    count_rays(a, used_);
    tex_write(a.gi_d0, a, gid, d0);
    tex_write(a.gi_d1, a, gid, d1);
    tex_write(a.gi_d2, a, gid, d2);
    if (!hit_some(prim_hit)) return;  // validation frames: pass a re-traces a reservoir wherever one is, pass b wants a surface too
    gi_sampling_b_cell(a, seed_b, tracing, pos, prim_hit, vres, lane_stack(a, lds), &used_, d0, d1, d2);
}
void launch_gi_sampling_ab(const KArgs& a, uint32_t seed_a, uint32_t seed_b, bool reproject, hipStream_t s) {
    ST_LAUNCH_TRACE(k_gi_sampling_ab, true, s, a, seed_a, seed_b, reproject ? 1u : 0u);
}

// ---------------------------------------------------------------- gi_temporal_resampling.rs:3-156
// REPROJECT: gi_reprojection.rs for the same pixel runs right here (tracing frames only, where this pass is the only reader
// of the reprojected reservoir): the reservoir fetched from last frame's gi_res[0] is used directly and still stored to
// gi_res[2], so the plane ends the frame with the reference's contents, but it is not written and read back through HBM
// by two launches.
// (76 VGPRs with REPROJECT = 6 waves per SIMD; asking for 7 spills 18 registers: 67.7 -> 82.3 us.)
template <bool REPROJECT>
__global__ ST_KERNEL_BOUNDS void k_gi_temporal(const KArgs a, uint32_t seed) {
    U2 lhs_pos;
    if (!resolve_gid(a, false, &lhs_pos) || !owns_pixel(a, lhs_pos)) return;
    const uint32_t n = a.width * a.height;
    const uint32_t lhs_idx = screen_to_idx(a, lhs_pos);
    float4* curr_res = a.gi_res[1];
    const float4* prev_res = a.gi_res[2];
    WhiteNoise wn = white_noise(seed, lhs_pos);
    const Hit lhs_hit = pixel_hit(a, a.cam, a.g0, a.g1, lhs_pos);
    // The pixel's own reservoirs are streamed with the quad-transposed accessors (st_device.h), which want the whole lane quad
    // at every call: a pixel without a surface therefore runs along with `some == false` instead of leaving early.
    const bool some = hit_some(lhs_hit);
    const bool tracing = frame_is_gi_tracing(a.frame);
    const bool got_sample = tracing ? (a.frame % 2u == 0u && got_checkerboard_at(lhs_pos, a.frame / 2u)) : got_checkerboard_at(lhs_pos, a.frame);
    const GiReservoir lhs = gi_read_own(curr_res, lhs_idx, true, some && got_sample);
    const Reprojection rp = reprojection_read(tex_read(a.reprojection, a, lhs_pos));
    const bool reprojected = some && rp.confidence > 0.0f;
    GiReservoir rhs = gi_empty();
    if (REPROJECT) {  // gi_reprojection.rs:3-51 for this pixel
        GiReservoir res = reprojected ? gi_read(a.gi_res[0], screen_to_idx(a, reprojection_prev_round(rp)), n) : gi_empty();
        res.confidence = 1.0f;
        res.s.v1_point = lhs_hit.point;
        if (!(a.lean & kLeanGiRes2)) gi_write_own(a.gi_res[2], lhs_idx, res, true, some);
        if (reprojected) rhs = gi_after_store(res);
    } else {
        rhs = gi_read_own(prev_res, lhs_idx, true, reprojected);
    }
    GiReservoir main_ = gi_empty();
    if (some) {
        Hit rhs_hit = hit_zero();
        if (reprojected) {
            rhs.confidence = 1.0f;
            rhs.m = fmin_(rhs.m, 128.0f);
            if (!tracing && lhs.m != 0.0f && rhs.m != 0.0f && gi_exists(rhs.s)) {
                if (distance(lhs.s.radiance, rhs.s.radiance) > 0.33f) rhs.confidence = 0.0f;
                rhs.s.radiance = lhs.s.radiance;
                rhs.s.v2_point = lhs.s.v2_point;
                rhs.s.v2_normal = lhs.s.v2_normal;
            }
            if (rhs.m != 0.0f) rhs_hit = pixel_hit(a, a.prev_cam, a.pg0, a.pg1, reprojection_prev_round(rp));
        } else rhs = gi_empty();
        float main_pdf = 0.0f;
        if (tracing) {
            Mis mis;
            mis.lhs_rhs_pdf = ((lhs.m > 0.0f) & hit_some(rhs_hit)) ? gi_pdf(lhs.s, rhs_hit) : 0.0f;
            mis.rhs_lhs_pdf = (rhs.m > 0.0f) ? gi_pdf(rhs.s, lhs_hit) : 0.0f;
            mis.lhs_m = lhs.m; mis.rhs_m = rhs.m; mis.rhs_jacobian = 1.0f; mis.lhs_lhs_pdf = lhs.s.pdf; mis.rhs_rhs_pdf = rhs.s.pdf;
            const MisResult mr = mis_eval(mis);
            if (res_update(main_, wn, lhs.s, mr.lhs_mis * mr.lhs_pdf * lhs.w)) main_pdf = mr.lhs_pdf;
            if (res_update(main_, wn, rhs.s, mr.rhs_mis * mr.rhs_pdf * rhs.w)) main_pdf = mr.rhs_pdf;
            main_.m = lhs.m + mr.m;
            main_.confidence = 1.0f;
            res_norm(main_, main_pdf, 1.0f, 1.0f);
        } else {
            if (res_merge(main_, wn, rhs, rhs.s.pdf)) main_pdf = rhs.s.pdf;
            main_.confidence = rhs.confidence;
            res_norm(main_, main_pdf, 1.0f, main_.m);
        }
        main_.s.pdf = main_pdf;
        main_.s.v1_point = lhs_hit.point;
        main_.w = fmin_(main_.w, 5.0f);
    }
    gi_write_own(curr_res, lhs_idx, main_, true, true);  // an empty reservoir where the pixel has no surface (gi_temporal_resampling.rs:30-33)
}
void launch_gi_temporal(const KArgs& a, uint32_t seed, bool fuse_reprojection, hipStream_t s) {
    if (fuse_reprojection) ST_LAUNCH(k_gi_temporal<true>, false, s, a, seed); else ST_LAUNCH(k_gi_temporal<false>, false, s, a, seed);
}

// ---------------------------------------------------------------- gi_spatial_resampling.rs:3-168 (pick)
// SpatialRecords as in k_di.hip: what this stage left in the scratch planes (gi_d0 / gi_d1) for the cell's two pixels; every
// exit of the GI pick stage writes both gi_d1 texels
struct GiSpatialRecords { bool wrote_d0; float4 a0, a1, b0, b1; };
ST_D GiSpatialRecords gi_spatial_pick_cell(const KArgs& a, uint32_t seed, U2 gid, U2 lhs_pos) {
    GiSpatialRecords out; out.wrote_d0 = false; out.a0 = out.a1 = out.b0 = out.b1 = f4z();
    const uint32_t n = a.width * a.height;
    const uint32_t lhs_idx = screen_to_idx(a, lhs_pos);
    WhiteNoise wn = white_noise(seed, lhs_pos);
    float4* buf_d0 = a.gi_d0; float4* buf_d1 = a.gi_d1;
    const float4* reservoirs = a.gi_res[1];
    const U2 buf_pos_a = u2(gid.x * 2u, gid.y), buf_pos_b = u2(gid.x * 2u + 1u, gid.y);
    const Hit lhs_hit = pixel_hit(a, a.cam, a.g0, a.g1, lhs_pos);
    // Reservoirs are fetched with the quad-cooperative gather (st_device.h gi_read_coop): a whole 64-B record per quad and
    // instruction instead of four loads per lane that each touch 64 different segments. That wants the quad's four lanes at
    // the fetch together, so the reference's `while rhs_nth < 8 { ... continue ... break }` (gi_spatial_resampling.rs:56-113)
    // runs in lockstep: every lane goes round until no lane of the wave is searching any more; a lane outside the loop — done,
    // or never in it because its pixel has no surface / no sample — idles along. Per lane the sequence of random numbers,
    // tests and fetches is the reference's.
    const GiReservoir lhs = gi_read_coop(reservoirs, lhs_idx, n, true);
    const bool valid = hit_some(lhs_hit) && lhs.m != 0.0f;
    GiReservoir rhs = gi_empty();
    uint32_t rhs_nth = 0u, rhs_idx = 0u;
    Hit rhs_hit = hit_zero();
    float rhs_jacobian = 0.0f;
    float max_radius = 128.0f;
    bool searching = valid;
    for (;;) {
        bool fetch = false;
        uint32_t cand_idx = 0u;
        if (searching) {
            if (rhs_nth >= 8u) searching = false;
            else {
                rhs_nth += 1u;
                const V2 disk = wn.sample_disk();
                const U2 rhs_pos = camera_contain(a, as_i2(as_v2(lhs_pos) + disk * max_radius));
                if (!(rhs_pos.x == lhs_pos.x && rhs_pos.y == lhs_pos.y)) {
                    rhs_hit = pixel_hit(a, a.cam, a.g0, a.g1, rhs_pos);
                    if (!hit_some(rhs_hit)) max_radius = fmax_(max_radius * 0.5f, 5.0f);
                    else if (fabsf(rhs_hit.g.depth - lhs_hit.g.depth) > 0.33f * lhs_hit.g.depth) max_radius = fmax_(max_radius * 0.5f, 5.0f);
                    else if (dot(rhs_hit.g.normal, lhs_hit.g.normal) < 0.33f) max_radius = fmax_(max_radius * 0.5f, 5.0f);
                    else { cand_idx = screen_to_idx(a, rhs_pos); fetch = true; }
                }
            }
        }
        if (!__ballot(searching)) break;
        const GiReservoir cand = gi_read_coop(reservoirs, cand_idx, n, fetch);
        if (fetch) {
            rhs_idx = cand_idx;
            rhs = cand;
            if (rhs.m != 0.0f) {
                rhs_jacobian = gi_jacobian(rhs.s, lhs_hit.point);
                if (rhs_jacobian < 1.0f / 10.0f || rhs_jacobian > 10.0f) rhs.m = 0.0f;
                else { rhs_jacobian = clampf(rhs_jacobian, 1.0f / 3.0f, 3.0f); searching = false; }  // `break`
            }
        }
    }
    if (!valid || rhs.m == 0.0f || !hit_some(rhs_hit)) { tex_write(buf_d1, a, buf_pos_a, f4z()); tex_write(buf_d1, a, buf_pos_b, f4z()); return out; }
    const float lhs_rhs_pdf = gi_pdf(lhs.s, rhs_hit);
    const float rhs_lhs_pdf = gi_pdf(rhs.s, lhs_hit);
    const Ray ray_a = lhs_rhs_pdf > 0.0f ? gi_sample_ray(lhs.s, rhs_hit.point) : zero_ray();
    const Ray ray_b = rhs_lhs_pdf > 0.0f ? gi_sample_ray(rhs.s, lhs_hit.point) : zero_ray();
    const V2 ea = normal_encode(ray_a.dir), eb = normal_encode(ray_b.dir);
    out.wrote_d0 = true;
    out.a0 = f4(ray_a.origin, ray_a.len); out.a1 = make_float4(ea.x, ea.y, b2f(rhs_idx + 1u), rhs_jacobian);
    out.b0 = f4(ray_b.origin, ray_b.len); out.b1 = make_float4(eb.x, eb.y, lhs_rhs_pdf, rhs_lhs_pdf);
    tex_write(buf_d0, a, buf_pos_a, out.a0); tex_write(buf_d1, a, buf_pos_a, out.a1);
    tex_write(buf_d0, a, buf_pos_b, out.b0); tex_write(buf_d1, a, buf_pos_b, out.b1);
    return out;
}
__global__ ST_KERNEL_BOUNDS void k_gi_spatial_pick(const KArgs a, uint32_t seed) {
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const U2 lhs_pos = resolve_checkerboard_alt(gid, a.frame / 2u);
    if (!owns_pixel(a, lhs_pos)) return;
    (void)gi_spatial_pick_cell(a, seed, gid, lhs_pos);
}
void launch_gi_spatial_pick(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH(k_gi_spatial_pick, true, s, a, seed); }

// ---------------------------------------------------------------- gi_spatial_resampling.rs:232-314 (sample)
// d0 / d1: the trace stage's texels (gi_d2) for the cell's two pixels
ST_D void gi_spatial_sample_cell(const KArgs& a, uint32_t seed, U2 gid, U2 pos, float4 d0, float4 d1) {
    const uint32_t n = a.width * a.height;
    const uint32_t idx = screen_to_idx(a, pos);
    WhiteNoise wn = white_noise(seed, pos);
    const float4* in_res = a.gi_res[1];
    float4* out_res = a.gi_res[2];
    const float lhs_rhs_vis = d0.x;
    const uint32_t rhs_idx = f2b(d0.y);
    const float rhs_jacobian = d0.z;
    const float rhs_lhs_vis = d1.x, lhs_rhs_pdf = d1.y, rhs_lhs_pdf = d1.z;
    // quad-cooperative record I/O (st_device.h): the cell's pixel, the picked neighbour, the other checkerboard pixel
    const GiReservoir lhs = gi_read_coop(in_res, idx, n, true);
    const bool merge = rhs_idx > 0u;
    const GiReservoir rhs = gi_read_coop(in_res, rhs_idx - 1u, n, merge);
    GiReservoir result = lhs;
    if (merge) {
        Mis mis;
        mis.lhs_m = lhs.m; mis.rhs_m = rhs.m; mis.rhs_jacobian = rhs_jacobian; mis.lhs_lhs_pdf = lhs.s.pdf;
        mis.lhs_rhs_pdf = lhs_rhs_pdf * lhs_rhs_vis; mis.rhs_lhs_pdf = rhs_lhs_pdf * rhs_lhs_vis; mis.rhs_rhs_pdf = rhs.s.pdf;
        const MisResult mr = mis_eval(mis);
        GiReservoir main_ = gi_empty();
        float main_pdf = 0.0f;
        if (res_update(main_, wn, lhs.s, mr.lhs_mis * mr.lhs_pdf * lhs.w)) main_pdf = mr.lhs_pdf;
        if (res_update(main_, wn, rhs.s, mr.rhs_mis * mr.rhs_pdf * rhs.w * rhs_jacobian)) main_pdf = mr.rhs_pdf;
        main_.m = lhs.m + mr.m;
        main_.confidence = 1.0f;
        main_.s.pdf = main_pdf;
        main_.s.v1_point = lhs.s.v1_point;
        res_norm(main_, main_pdf, 1.0f, 1.0f);
        main_.w = fmin_(main_.w, 5.0f);
        result = main_;
    }
    gi_write_coop(out_res, idx, result, true);
    const U2 other = resolve_checkerboard(gid, a.frame / 2u);
    const bool has_other = contains_u(a, other);
    const uint32_t oi = has_other ? screen_to_idx(a, other) : 0u;
    gi_write_coop(out_res, oi, gi_read_coop(in_res, oi, n, has_other), has_other);
}
__global__ ST_KERNEL_BOUNDS void k_gi_spatial_sample(const KArgs a, uint32_t seed) {
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const U2 pos = resolve_checkerboard_alt(gid, a.frame / 2u);
    if (!owns_pixel(a, pos)) return;
    gi_spatial_sample_cell(a, seed, gid, pos, tex_read(a.gi_d2, a, u2(gid.x * 2u, gid.y)), tex_read(a.gi_d2, a, u2(gid.x * 2u + 1u, gid.y)));
}
void launch_gi_spatial_sample(const KArgs& a, uint32_t seed, hipStream_t s) { ST_LAUNCH(k_gi_spatial_sample, true, s, a, seed); }

// gi_spatial_resampling.rs pick + trace + sample for one 2x1 cell in one launch (see k_di_spatial_fused, k_di.hip)
// (5 waves per SIMD = 96 VGPRs: the allocator's free choice crossed to 97 — 4 waves — when the wide walk's overflow report went in; the loop is the one that waits, wait_any 0.49)
template <bool LDS_SCENE, class SE>
__global__ __launch_bounds__(kBlockThreads, 5) void k_gi_spatial_fused(const KArgs a_in, uint32_t seed_pick, uint32_t seed_sample) {
    ST_SCENE_PROLOGUE_WITH_BYTE_TABLES
    ST_STACK_LDS(SE, lds);
    U2 gid;
    if (!resolve_gid(a, true, &gid)) return;
    const U2 lhs_pos = resolve_checkerboard_alt(gid, a.frame / 2u);
    const bool own_lhs = owns_pixel(a, lhs_pos);
    GiSpatialRecords rec; rec.wrote_d0 = false; rec.a0 = rec.a1 = rec.b0 = rec.b1 = f4z();
    if (own_lhs) rec = gi_spatial_pick_cell(a, seed_pick, gid, lhs_pos);  // writes both gi_d1 texels on every path
    float4 vis[2];
    uint32_t rays = 0u; unsigned long long bytes = 0ull;
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const U2 pos = u2(gid.x * 2u + (uint32_t)k, gid.y);
        if (!owns_pixel(a, pos)) { vis[k] = tex_read(a.gi_d2, a, pos); continue; }
        const float4 r1 = own_lhs ? (k == 0 ? rec.a1 : rec.b1) : tex_read(a.gi_d1, a, pos);
        if (is_zero(r1)) vis[k] = f4z();
        else {
            const float4 r0 = rec.wrote_d0 ? (k == 0 ? rec.a0 : rec.b0) : tex_read(a.gi_d0, a, pos);
            Ray ray = make_ray(xyz(r0), normal_decode(v2(r1.x, r1.y)));
            ray.len = r0.w;
            uint32_t used_ = 0u;
            const bool occluded = trace_any(a, ray, lane_stack(a, lds), &used_);
            rays += 1u; bytes += used_;
            vis[k] = make_float4(occluded ? 0.0f : 1.0f, r1.z, r1.w, 0.0f);
        }
        tex_write(a.gi_d2, a, pos, vis[k]);
    }
    if (rays) count_rays_n(a, rays, bytes);
    if (own_lhs) gi_spatial_sample_cell(a, seed_sample, gid, lhs_pos, vis[0], vis[1]);
}
void launch_gi_spatial_fused(const KArgs& a, uint32_t seed_pick, uint32_t seed_sample, hipStream_t s) {
    ST_LAUNCH_TRACE(k_gi_spatial_fused, true, s, a, seed_pick, seed_sample);
}

// ---------------------------------------------------------------- gi_preview_resampling.rs:3-138
// RESOLVE (second preview pass only): gi_resolving.rs and, if `reproject`, the GI half of frame_denoising.rs::reproject run
// for the same pixel right here. Resolving consumes exactly what this pass would have stored at gi_res[0][pixel] — the merged
// reservoir, an empty one on sky, or (the reference's early `return`, :84-86) whatever gi_res[0] already held — and then
// overwrites that slot with the frame's source reservoir, so the intermediate store is dropped.
// One preview pass for one pixel (gi_preview_resampling.rs:60-128). `center`: the pixel's reservoir as the pass loads it;
// `center_hit`: the pixel's Hit if `hit_ready`, otherwise it is rebuilt here, and only when a neighbour is actually drawn.
// keep_stored: the reference's early `return` (:84-86) — the output slot keeps its previous contents.
struct PreviewPass { GiReservoir r; bool keep_stored; uint32_t max_samples; };
// the same for a pixel that turns out to draw no neighbour (max_samples == 0); otherwise only max_samples is meaningful
ST_D PreviewPass gi_preview_pass_if_alone(uint32_t seed, U2 center_pos, const GiReservoir& center) {
    PreviewPass o; o.r = gi_empty(); o.keep_stored = false;
    WhiteNoise wn = white_noise(seed, center_pos);
    float main_pdf = 0.0f;
    if (res_merge(o.r, wn, center, center.s.pdf)) main_pdf = center.s.pdf;
    o.max_samples = f2u_sat(lerpf(8.0f, 0.0f, o.r.m * 0.125f));
    if (o.max_samples > 0u) return o;
    o.r.confidence = center.confidence;
    o.r.s.pdf = main_pdf;
    o.r.s.v1_point = center.s.v1_point;
    res_norm(o.r, main_pdf, 1.0f, o.r.m);
    o.r.w = fmin_(o.r.w, 5.0f);
    return o;
}
// What GI_RESERVOIRS_3 holds for pixel `pos` (index `idx`) after the first preview pass, in the lean frame (kLeanGiMid): a pixel
// whose first pass drew no neighbour was not stored — its record is that pass's normalisation of the input plane's record,
// rebuilt here (through the store / load codec, as a reader of the plane would see it); a pixel that did resample (or hit the
// reference's early `return`) has its slot as always.
ST_D GiReservoir gi_mid_value(const KArgs& a, uint32_t seed, const float4* mid, U2 pos, uint32_t idx, uint32_t n) {
    const PreviewPass p = gi_preview_pass_if_alone(seed, pos, gi_read(a.gi_mid_src, idx, n));
    return p.max_samples == 0u ? gi_after_store(p.r) : gi_read(mid, idx, n);
}
ST_D PreviewPass gi_preview_pass(const KArgs& a, uint32_t seed, uint32_t nth, const float4* in, U2 center_pos, bool center_some, const GiReservoir& center,
                                 Hit center_hit, bool hit_ready) {
    PreviewPass o; o.r = gi_empty(); o.keep_stored = false; o.max_samples = 0u;
    if (!center_some) return o;
    const uint32_t n = a.width * a.height;
    WhiteNoise wn = white_noise(seed, center_pos);
    float main_pdf = 0.0f;
    if (res_merge(o.r, wn, center, center.s.pdf)) main_pdf = center.s.pdf;
    const uint32_t max_samples = f2u_sat(lerpf(8.0f, 0.0f, o.r.m * 0.125f));
    o.max_samples = max_samples;
    if (!hit_ready && max_samples > 0u) center_hit = pixel_hit(a, a.cam, a.g0, a.g1, center_pos);
    const float max_radius = nth == 0u ? 128.0f : 64.0f;
    // (Measured and not kept, round 3: this loop in lockstep with quad-cooperative neighbour fetches, as k_gi_spatial's pick loop
    // runs — bit-identical, static frame unchanged, but the moving-scene frame, where these loops actually run, 0.889 against
    // 0.878 ms: the late launch's lanes are sparse, so most quads fall back to per-lane loads and still pay for the lockstep.)
    uint32_t sample_nth = 0u;
    while (sample_nth < max_samples) {
        sample_nth += 1u;
        const V2 disk = wn.sample_disk();
        const U2 sample_pos = camera_contain(a, as_i2(as_v2(center_pos) + disk * max_radius));
        if (sample_pos.x == center_pos.x && sample_pos.y == center_pos.y) { o.keep_stored = true; break; }  // sic: `return`, not `continue`
        const Surface ss = surface_decoded(tex_read(a.sn, a, sample_pos));
        if (ss.depth == 0.0f) continue;
        if (fabsf(ss.depth - center_hit.g.depth) > 0.25f * center_hit.g.depth) continue;
        if (dot(ss.normal, center_hit.g.normal) < 0.5f) continue;
        const uint32_t sample_idx = screen_to_idx(a, sample_pos);
        // (Measured and not kept, round 6: in the launch that serves the flagged pixels, a tap's three records — surface texel, input-plane record, first-pass
        // record — fetched TOGETHER before the surface tests, one round trip instead of three: 17.0 -> 19.4 us on Cornell, 17.3 -> 18.7 on the dungeon
        // (profiles/r06_ab_vs_r05.txt against r06_ab_exact_primary.txt rows A): most taps of a flagged pixel fail the surface tests after ONE load, and now waited for nine.)
        const GiReservoir s = (nth != 0u && a.gi_mid_src) ? gi_mid_value(a, seed, in, sample_pos, sample_idx, n) : gi_read(in, sample_idx, n);
        if (s.m == 0.0f) continue;
        const float sample_pdf = gi_pdf(s.s, center_hit);
        float sample_jacobian = gi_jacobian(s.s, center_hit.point);
        if (sample_jacobian < 1.0f / 10.0f || sample_jacobian > 10.0f) continue;
        sample_jacobian = clampf(sample_jacobian, 1.0f / 3.0f, 3.0f);
        if (res_merge(o.r, wn, s, sample_pdf * sample_jacobian)) main_pdf = sample_pdf;
    }
    if (!o.keep_stored) {
        o.r.confidence = center.confidence;
        o.r.s.pdf = main_pdf;
        o.r.s.v1_point = center.s.v1_point;
        res_norm(o.r, main_pdf, 1.0f, o.r.m);
        o.r.w = fmin_(o.r.w, 5.0f);
    }
    return o;
}
// one bit per pixel of an 8x8 tile (bit = lane = pixel_in_tile's numbering), one word per tile (KArgs::gi_late_mask)
ST_D uint32_t gi_late_index(const KArgs& a, U2 pos) { return (pos.y >> 3) * ((a.width + 7u) >> 3) + (pos.x >> 3); }

// RESOLVE (second preview pass only): gi_resolving.rs and, if `reproject`, the GI half of frame_denoising.rs::reproject run
// for the same pixel right here. Resolving consumes exactly what this pass would have stored at gi_res[0][pixel] — the merged
// reservoir, an empty one on sky, or (the reference's early `return`, :84-86) whatever gi_res[0] already held — and then
// overwrites that slot with the frame's source reservoir, so the intermediate store is dropped.
// With KArgs::gi_preview_late the launch follows k_gi_preview_both and serves only the pixels that one flagged.
template <bool RESOLVE>
__global__ ST_KERNEL_BOUNDS void k_gi_preview(const KArgs a, uint32_t seed, uint32_t nth, const float4* in, float4* out, uint32_t source,
                                                              uint32_t reproject) {
    U2 center_pos;
    if (!resolve_gid(a, false, &center_pos) || !owns_pixel(a, center_pos)) return;
    if (RESOLVE && a.gi_preview_late && ((a.gi_late_mask[gi_late_index(a, center_pos)] >> (threadIdx.x & 63u)) & 1ull) == 0ull) return;
    const uint32_t n = a.width * a.height;
    const uint32_t center_idx = screen_to_idx(a, center_pos);
    // The pixel's Hit (camera ray + both G-buffer texels decoded) is needed by a neighbour tap and by resolving; whether the
    // pixel has a surface at all is the first G-buffer texel's depth. Once a reservoir's m has reached 8 — which temporal
    // resampling does within a few frames — `max_samples` is 0 and the first preview pass is a normalised copy: it then reads
    // one G-buffer texel instead of two and skips the ray reconstruction.
    const float4 center_g0 = tex_read(a.g0, a, center_pos);
    const bool center_some = center_g0.x != 0.0f;  // GBufferEntry::depth (gbuffer.rs:60) == Hit::is_some
    Hit center_hit = hit_zero();
    if (RESOLVE) center_hit = pixel_hit(a, a.cam, a.g0, a.g1, center_pos);
    ReprojectHistory history;  // fetched ahead of the resampling loop (st_passes.h)
    if (RESOLVE && reproject) history = denoise_reproject_prefetch(a, center_pos, a.gi_diff_prev_colors, a.gi_diff_prev_moments);
    GiReservoir center;
    if (RESOLVE && a.gi_mid_src) center = center_some ? gi_mid_value(a, seed, in, center_pos, center_idx, n) : gi_empty();  // lean frame: see gi_mid_value
    else center = gi_read_own(in, center_idx, true, center_some);  // quad-transposed (st_device.h): before the branch
    const PreviewPass pass = gi_preview_pass(a, seed, nth, in, center_pos, center_some, center, center_hit, RESOLVE);
    GiReservoir main_ = pass.r;
    if (!RESOLVE) {
        gi_write_own(out, center_idx, main_, true, !pass.keep_stored);
        return;
    }
    if (pass.keep_stored) main_ = gi_read(out, center_idx, n);
    const float4 diff = gi_resolve_pixel(a, center_pos, center_idx, center_hit, main_, source, reproject != 0u);
    if (reproject) denoise_reproject_finish(a, center_pos, diff, history, a.gi_diff_curr_colors, a.gi_diff_moments);
}
// Both preview passes, resolving and (if `reproject`) the GI half of denoise-reproject in one launch, for the pixels whose
// SECOND pass draws no neighbour — nearly all of them once the reservoirs have history: such a pixel's second pass reads
// nothing but what its first pass has just produced (through the store / load codec), so the 64-B round trip through
// gi_res[3] and a second decode of the G-buffer go away. The first pass's result is stored for every pixel as always (a
// neighbour's second pass may draw it); pixels whose second pass does resample — or whose first pass hit the early `return`
// — are flagged per tile and served by k_gi_preview<true> afterwards, when every first-pass result is in memory.
__global__ ST_KERNEL_BOUNDS void k_gi_preview_both(const KArgs a, uint32_t seed, const float4* in, float4* mid, uint32_t source, uint32_t reproject) {
    U2 center_pos;
    if (!resolve_gid(a, false, &center_pos) || !owns_pixel(a, center_pos)) return;
    const uint32_t center_idx = screen_to_idx(a, center_pos);
    const float4 center_g0 = tex_read(a.g0, a, center_pos);
    const bool center_some = center_g0.x != 0.0f;
    const Hit center_hit = pixel_hit(a, a.cam, a.g0, a.g1, center_pos);
    ReprojectHistory history;
    if (reproject) history = denoise_reproject_prefetch(a, center_pos, a.gi_diff_prev_colors, a.gi_diff_prev_moments);
    const GiReservoir center0 = gi_read_own(in, center_idx, true, center_some);
    const PreviewPass first = gi_preview_pass(a, seed, 0u, in, center_pos, center_some, center0, center_hit, true);
    // lean frame: a result that is the plain normalisation of the input needs no slot (gi_mid_value rebuilds it where it is read)
    gi_write_own(mid, center_idx, first.r, true, !first.keep_stored && !((a.lean & kLeanGiMid) && (!center_some || first.max_samples == 0u)));
    // second pass, if it is the neighbour-free kind: its `center` is what gi_read would return for the record just stored
    bool late = first.keep_stored;
    PreviewPass second; second.r = gi_empty(); second.keep_stored = false; second.max_samples = 0u;
    if (!late && center_some) {
        second = gi_preview_pass_if_alone(seed, center_pos, gi_after_store(first.r));
        late = second.max_samples > 0u;
    }
    const unsigned long long flagged = __ballot(late), active = __ballot(true);
    if ((threadIdx.x & 63u) == (uint32_t)__ffsll((long long)active) - 1u) a.gi_late_mask[gi_late_index(a, center_pos)] = flagged;
    if (late) return;
    const float4 diff = gi_resolve_pixel(a, center_pos, center_idx, center_hit, second.r, source, reproject != 0u);
    if (reproject) denoise_reproject_finish(a, center_pos, diff, history, a.gi_diff_curr_colors, a.gi_diff_moments);
}
void launch_gi_preview(const KArgs& a, uint32_t seed, uint32_t nth, const float4* in, float4* out, hipStream_t s) {
    ST_LAUNCH(k_gi_preview<false>, false, s, a, seed, nth, in, out, 0u, 0u);
}
void launch_gi_preview_resolve(const KArgs& a, uint32_t seed, uint32_t nth, const float4* in, uint32_t source, bool reproject, hipStream_t s) {
    ST_LAUNCH(k_gi_preview<true>, false, s, a, seed, nth, in, a.gi_res[0], source, reproject ? 1u : 0u);
}
void launch_gi_preview_both(const KArgs& a, uint32_t seed, const float4* in, float4* mid, uint32_t source, bool reproject, hipStream_t s) {
    ST_LAUNCH(k_gi_preview_both, false, s, a, seed, in, mid, source, reproject ? 1u : 0u);
}

// ---------------------------------------------------------------- gi_resolving.rs:3-67
__global__ ST_KERNEL_BOUNDS void k_gi_resolving(const KArgs a, uint32_t source) {
    U2 pos;
    if (!resolve_gid(a, false, &pos) || !owns_pixel(a, pos)) return;
    const uint32_t n = a.width * a.height;
    const uint32_t idx = screen_to_idx(a, pos);
    const Hit hit = pixel_hit(a, a.cam, a.g0, a.g1, pos);
    gi_resolve_pixel(a, pos, idx, hit, gi_read(a.gi_res[0], idx, n), source);
}
void launch_gi_resolving(const KArgs& a, uint32_t source, hipStream_t s) { ST_LAUNCH(k_gi_resolving, false, s, a, source); }

}  // namespace ST_KNS
}  // namespace st
