// st_types.h — PODs shared by the host engine and the kernels (device buffer layouts).
#pragma once
#include "st_math.h"

namespace st {

// strolle-gpu/src/camera.rs:9-17 (160 B uniform)
struct GpuCamera {
    M4 projection_view;
    M4 ndc_to_world;
    float4 origin;
    float4 screen;
};
static_assert(sizeof(GpuCamera) == 160, "Camera uniform is 160 B");

// strolle-gpu/src/material.rs:8-23 (112 B)
struct GpuMaterial {
    float4 base_color, base_color_texture, emissive, emissive_texture;
    float roughness, metallic, reflectance, ior;
    float4 metallic_roughness_texture, normal_map_texture;
};
static_assert(sizeof(GpuMaterial) == 112, "Material is 112 B");

// strolle-gpu/src/light.rs:12-43 (112 B)
struct GpuLight { float4 d0, d1, d2, d3, prev_d0, prev_d1, prev_d2; };
static_assert(sizeof(GpuLight) == 112, "Light is 112 B");

// The reference's 144-B AoS triangle (strolle-gpu/src/triangle.rs:9-21), kept on the HOST only (debug
// read-back + BVH build). The device gets:
//   tri_attr[4 float4 / triangle]: (n0, uv0.x) (n1, uv0.y) (n2, uv1.x) (uv1.y, uv2.x, uv2.y, instance slot)
//     fetched once per ray, for the winning triangle only;
//   the hit-test record (v0.xyz, 0) (v1-v0, 0) (v2-v0, 0) — all a hit test reads (48 B) — inline in the BVH stream:
// device BVH stream: the serializer's stream (bvh/serializer.rs:20-110, what st_debug_bvh_stream returns and the
//   stream-contract test compares) re-laid so that EVERY entry is four float4 and entry k starts at texel 4 k:
//     internal node  (min0.xyz, 0) (max0.xyz, far child's byte offset) (min1.xyz, -) (max1.xyz, -)   the near child is the next entry
//     leaf entry     (flags, triangle, material, marker != 0) (v0.xyz, 0) (v1-v0, 0) (v2-v0, 0)   flags & 1: another
//                    entry of the same leaf follows
//   One 64-B fetch per traversal step, whichever kind the entry is; traversal pointers are byte offsets into this form
//   (the per-lane stack keeps entry numbers). Visiting order and the `used_memory` count are those of the serializer's stream.
struct HostTriangle { float4 d0, d1, d2, d3, d4, d5, d6, d7, d8; };
static_assert(sizeof(HostTriangle) == 144, "Triangle is 144 B");

// KArgs::lean bits
constexpr uint32_t kRefitBatch = 512;  // leaves one workgroup of k_bvh_refit takes (st_engine.cpp index_device_tree): 26 KB of LDS
constexpr uint32_t kLeanPrim = 1u;     // primary visibility + frame reprojection in one launch: the velocity map (consumed in registers) and the
                                       // encoded surface map (every kernel reads its decoded twin, KArgs::sn) stay unwritten
constexpr uint32_t kLeanSamples = 2u;  // di / gi diffuse sample planes: the fused denoise-reproject stages consume them in registers
constexpr uint32_t kLeanGiRes2 = 4u;   // tracing frames: the reprojected GI reservoirs the fused temporal pass consumes in registers (on odd
                                       // tracing frames the spatial pass rewrites the whole plane anyway)
constexpr uint32_t kLeanGiMid = 8u;    // both GI preview passes in one launch: a first-pass result that is a plain normalisation of its input (the pixel drew
                                       // no neighbour: every pixel once the reservoirs have history) is not stored to GI_RESERVOIRS_3; the few pixels
                                       // whose second pass does resample rebuild such a neighbour's record from the pass's input (KArgs::gi_mid_src)
constexpr int kBvhStackSize = 24;  // strolle-gpu/src/lib.rs:76
constexpr int kBvhStackSizeDeep = 32;  // trees whose deepest chain of internal nodes exceeds kBvhStackSize (st_bvh_refresh.cpp measure_stack_need)
constexpr uint32_t kLightIdSky = 0xffffffffu;
constexpr uint32_t kLdsSceneTexels = 448;  // device streams up to this many float4 are copied into LDS by the tracing kernels (7 KiB per block: 112 entries, e.g. 56 one-triangle leaves + 55 internal nodes)
constexpr uint32_t kLdsLights = 16;  // lights the tracing kernels keep in LDS (k_common.h ST_SCENE_PROLOGUE)
constexpr uint32_t kCounterLines = 256;  // ray/byte counters are spread over this many 64-B lines per kernel slot

// Everything a per-pixel kernel can touch, passed by value as the kernel argument (scalar loads).
struct KArgs {
    GpuCamera cam, prev_cam;
    // engine-level bindings
    const float4* bvh; const float4* tri_attr;
    const float4* instance_xforms;  // 8 float4 per instance slot: curr_xform_inv (3 axes + translation), prev_xform; slot = tri_attr[4 t + 3].w
    const GpuMaterial* materials; const GpuLight* lights;
    const GpuLight* lights_lds;   // set by the tracing kernels' prologue: the first kLdsLights lights in LDS (nullptr from the host)
    const uint32_t* material_base_packed;  // per material: gbuffer_pack_base_color(base_color), valid where it has no base-colour texture
    const uchar4* atlas; const uchar4* blue_noise;
    const float* byte_luts;  // 256 sRGB->linear + 256 unorm8 values (st_device.h kLut*), generated on the device at engine creation
    const float4* transmittance_lut; const float4* sky_lut;
    uint32_t bvh_len, n_lights_buf, light_count, atlas_w, atlas_h;
    uint32_t stack_entries;    // pending entries per lane of the traversal stack: kBvhStackSize, or kBvhStackSizeDeep for deeper trees (dynamic LDS, sized at the launch)
    // k_denoise.hip "variance in the reproject stage": the fused reproject stages store each pixel's long-history variance in
    // curr_colors.w and the DI one flags short-history pixels per 8x8 tile (bit = lane) for the variance kernel
    unsigned long long* tile_mask; uint32_t variance_in_reproject;
    // k_gi.hip k_gi_preview_both: pixels whose second preview pass resamples, per 8x8 tile (bit = lane); gi_preview_late: the
    // second-pass launch serves flagged pixels only
    unsigned long long* gi_late_mask; uint32_t gi_preview_late;
    uint32_t skip_dead_scratch;  // the fused DI spatial launch keeps its pick / trace records in registers only: resolving, denoise-reproject and the a-trous chain rewrite the three scratch planes later in this frame
    uint32_t gi_skip_history_copy;  // gi_resolving leaves GI_RESERVOIRS_0 alone: the engine swaps plane pointers instead (st_engine.cpp gi_aliased)
    // The lean frame (fast build, whole pass graph, Image{denoise}; st_engine.cpp `lean_frame`): stores that nothing reads —
    // not a later pass of this frame, not the next frame — are not made. Bits: kLean*. st_debug_keep_all_planes(1) /
    // ST_KEEP_ALL_PLANES=1 keeps every plane as the reference leaves it.
    uint32_t lean;
    const float4* gi_mid_src;  // kLeanGiMid, second-pass launch only: the first preview pass's input plane (nullptr: GI_RESERVOIRS_3 holds every first-pass result)
    const float4* bvh_c;       // fast build: the compact stream the shadow rays walk (k_bvh.hip k_bvh_compact; nullptr: they walk `bvh`)
    uint32_t bvh_c_root;       // ... its entry 0 with the kind bit (entry << 1 | is a leaf entry)
    const float4* bvh_w;       // fast build: the WIDE stream's nodes (k_bvh.hip k_bvh_wide: four f16 child boxes + four links per 64-B line; nullptr: none)
    uint32_t bvh_w_leaf_off;   // ... byte offset of its leaf records (48 B each) in the same allocation
    uint32_t bvh_w_root;       // ... the root's link (index << 1 | is a leaf record)
    uint32_t primary_packets;  // ... 1: primary visibility walks it as ONE packet per wave (st_device.h closest_hit_packet; StTuning::primary_packets)
    uint32_t bvh_w_link_mask;  // ... 32-bit form: (1 << bits) - 1, bits = what the largest link needs (the sort key keeps the link there)
    uint32_t bvh_w_links16;    // ... 1: links are 16-bit (fewer than 32768 nodes and leaf records): kernels run with 16-bit stack slots
    uint32_t* walk_flags;      // ... two sticky words in page-locked host memory: [0] a per-lane wide walk, [1] the packet walk found its stack full and DROPPED a push (st_device.h wide_walk_overflowed; st_tick.cpp reads them)
    uint32_t exp_flags;        // A/B switches of experiments in flight (ST_EXP in the environment; 0 in the shipped configuration)
    uint32_t anyhit_contract;  // fast build: shadow rays walk the contract loop (set while the reference's used_memory bytes are counted, or by StTuning::anyhit_fast = 0)
    uint32_t count_bytes;  // st_profile_enable bit 1: kernels also sum the reference's used_memory over their rays
    uint32_t tri_slots;  // triangle records in tri_attr (upper bound of every triangle id in the BVH stream)
    float sun_altitude;
    float sun_dir[3];
    // per-camera planes (A/B resolved for this frame)
    float4 *g0, *g1, *sm;                // prim_gbuffer_d0/d1, prim_surface_map (curr)
    const float4 *pg0, *pg1, *psm;       // previous frame's
    // Decoded twin of the surface map: (normal.xyz, depth) per pixel, written by primary visibility. Neighbour-tap loops
    // (à-trous, variance, preview resampling, reprojection validity) read it instead of re-running the octahedral decode
    // (a sqrt and a division per tap); the bytes per tap are the same 16. Not part of the reference's buffer set.
    float4* sn; const float4* psn;
    float4 *reprojection, *velocity;
    float4* di_res[3];
    float4 *di_diff_samples, *di_diff_prev_colors, *di_diff_curr_colors, *di_diff_moments, *di_diff_stash, *di_spec_samples;
    const float4* di_diff_prev_moments;
    float4 *gi_d0, *gi_d1, *gi_d2;
    float4* gi_res[4];
    float4 *gi_diff_samples, *gi_diff_prev_colors, *gi_diff_curr_colors, *gi_diff_moments, *gi_diff_stash, *gi_spec_samples;
    const float4* gi_diff_prev_moments;
    float4 *ref_hits, *ref_rays, *ref_colors;
    uint32_t* dbg_used_memory;
    unsigned long long* ray_counter;
    uint32_t width, height, row0, row1;  // [row0,row1) x [col0,col1): the window of the viewport this launch covers (multi-GPU row bands / 2-D tiles;
    uint32_t col0, col1;                 // st_camera_set_rows / st_camera_set_window); pixels keep their absolute coordinates
    uint32_t frame;
    uint32_t tile_map;  // blockIdx -> tile mapping (st_device.h tile_for_thread)
};

}  // namespace st
