// st_engine.h — the host engine of libstrolle_hip.so: state and interfaces shared by its translation units
//   st_engine.cpp       construction, tuning (StTuning + environment), destruction
//   st_scene.cpp        scene stores: materials, images, lights, instances -> world-space triangles (baking)
//   st_bvh_refresh.cpp  device form of the BVH stream, refit (host) and the device refit's index arrays, depth check
//   st_tick.cpp         Engine::tick: refresh + uploads (double-buffered scene / light copies, page-locked staging)
//   st_render.cpp       per-camera buffers and the per-frame pass graph on two HIP streams, present hand-over
//   st_profile.cpp      per-kernel event timing
//   st_abi.cpp          the C ABI (include/strolle_hip.h)
// Behavioural contract: strolle/src/lib.rs (Engine), camera_controller.rs (pass order), lights.rs / materials.rs /
// instances.rs / triangles.rs (stores), camera.rs (camera uniform). The wgpu plumbing of the reference (bind groups, mapped
// buffers, textures) is replaced by plain device allocations and pointer swaps.
#pragma once
#include <algorithm>
#include <cstdio>
#include <chrono>
#include <cstring>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/strolle_hip.h"
#include "st_lbvh.h"
#include "st_atlas.h"
#include "st_bvh.h"
#include "st_kernels.h"

#include <thread>

namespace st {

extern thread_local std::string g_last_error;  // st_engine.cpp
inline int fail(int status, const std::string& msg) { g_last_error = msg; return status; }

#define ST_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t err_ = (call);                                                                         \
        if (err_ != hipSuccess) return fail(ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err_)); \
    } while (0)

// ------------------------------------------------------------------ host maths (glam order; see st_math.h)
inline M4 m4_from_cols(const float* a) { M4 m; for (int i = 0; i < 4; i++) m.c[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]); return m; }
inline M4 m4_mul(const M4& a, const M4& b) { M4 r; for (int i = 0; i < 4; i++) r.c[i] = mul(a, b.c[i]); return r; }
inline M4 m4_inverse(const M4& m) {  // glam 0.24 Mat4::inverse, scalar path
    const float m00 = m.c[0].x, m01 = m.c[0].y, m02 = m.c[0].z, m03 = m.c[0].w;
    const float m10 = m.c[1].x, m11 = m.c[1].y, m12 = m.c[1].z, m13 = m.c[1].w;
    const float m20 = m.c[2].x, m21 = m.c[2].y, m22 = m.c[2].z, m23 = m.c[2].w;
    const float m30 = m.c[3].x, m31 = m.c[3].y, m32 = m.c[3].z, m33 = m.c[3].w;
    const float c00 = m22 * m33 - m32 * m23, c02 = m12 * m33 - m32 * m13, c03 = m12 * m23 - m22 * m13;
    const float c04 = m21 * m33 - m31 * m23, c06 = m11 * m33 - m31 * m13, c07 = m11 * m23 - m21 * m13;
    const float c08 = m21 * m32 - m31 * m22, c10 = m11 * m32 - m31 * m12, c11 = m11 * m22 - m21 * m12;
    const float c12 = m20 * m33 - m30 * m23, c14 = m10 * m33 - m30 * m13, c15 = m10 * m23 - m20 * m13;
    const float c16 = m20 * m32 - m30 * m22, c18 = m10 * m32 - m30 * m12, c19 = m10 * m22 - m20 * m12;
    const float c20 = m20 * m31 - m30 * m21, c22 = m10 * m31 - m30 * m11, c23 = m10 * m21 - m20 * m11;
    const float4 f0 = make_float4(c00, c00, c02, c03), f1 = make_float4(c04, c04, c06, c07), f2 = make_float4(c08, c08, c10, c11);
    const float4 f3 = make_float4(c12, c12, c14, c15), f4_ = make_float4(c16, c16, c18, c19), f5 = make_float4(c20, c20, c22, c23);
    const float4 v0 = make_float4(m10, m00, m00, m00), v1 = make_float4(m11, m01, m01, m01), v2_ = make_float4(m12, m02, m02, m02), v3_ = make_float4(m13, m03, m03, m03);
    const float4 i0 = (v1 * f0 - v2_ * f1) + v3_ * f2;
    const float4 i1 = (v0 * f0 - v2_ * f3) + v3_ * f4_;
    const float4 i2_ = (v0 * f1 - v1 * f3) + v3_ * f5;
    const float4 i3 = (v0 * f2 - v1 * f4_) + v2_ * f5;
    const float4 sa = make_float4(1.0f, -1.0f, 1.0f, -1.0f), sb = make_float4(-1.0f, 1.0f, -1.0f, 1.0f);
    M4 inv;
    inv.c[0] = i0 * sa; inv.c[1] = i1 * sb; inv.c[2] = i2_ * sa; inv.c[3] = i3 * sb;
    const float4 col0 = make_float4(inv.c[0].x, inv.c[1].x, inv.c[2].x, inv.c[3].x);
    const float4 d0 = m.c[0] * col0;
    const float det = d0.x + d0.y + d0.z + d0.w;
    const float rcp = 1.0f / det;
    for (int i = 0; i < 4; i++) inv.c[i] = inv.c[i] * rcp;
    return inv;
}
struct Affine { V3 x, y, z, t; };
inline Affine affine_from12(const float* a) { Affine r; r.x = v3(a[0], a[1], a[2]); r.y = v3(a[3], a[4], a[5]); r.z = v3(a[6], a[7], a[8]); r.t = v3(a[9], a[10], a[11]); return r; }
inline V3 affine_vec(const Affine& a, V3 v) { V3 r = a.x * v.x; r = r + a.y * v.y; r = r + a.z * v.z; return r; }
inline V3 affine_point(const Affine& a, V3 p) { return ((a.x * p.x) + (a.y * p.y) + (a.z * p.z)) + a.t; }
inline Affine affine_inverse(const Affine& a) {  // glam Affine3A::inverse
    const V3 t0 = cross(a.y, a.z), t1 = cross(a.z, a.x), t2 = cross(a.x, a.y);
    const float det = dot(a.z, t2);
    const float inv_det = 1.0f / det;
    const V3 c0 = t0 * inv_det, c1 = t1 * inv_det, c2 = t2 * inv_det;
    Affine r;
    r.x = v3(c0.x, c1.x, c2.x); r.y = v3(c0.y, c1.y, c2.y); r.z = v3(c0.z, c1.z, c2.z);
    r.t = -affine_vec(r, a.t);
    return r;
}

// per-pass seeds (NEW seam): the same definition is stated in DESIGN.md
inline uint32_t seed_hash(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    const uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}
inline uint32_t pass_seed(uint64_t base, uint32_t frame, uint32_t pass_id) {
    return seed_hash((uint32_t)base ^ seed_hash((uint32_t)(base >> 32) ^ seed_hash(frame ^ seed_hash(pass_id))));
}
enum PassSeedId { SEED_DI_SAMPLING = 1, SEED_DI_TEMPORAL = 2, SEED_DI_SPATIAL_PICK = 3, SEED_DI_SPATIAL_SAMPLE = 5, SEED_GI_SAMPLING_A = 8,
                  SEED_GI_SAMPLING_B = 9, SEED_GI_TEMPORAL = 10, SEED_GI_SPATIAL_PICK = 11, SEED_GI_SPATIAL_SAMPLE = 13, SEED_GI_PREVIEW = 14,
                  SEED_REF_SHADING = 200 };

// sun light colour: atmosphere/generate_transmittance_lut.rs:32-59 evaluated on the host (lights.rs:80-95)
inline V3 sun_transmittance(V3 pos, V3 sun_dir) {
    auto sphere = [&](float radius) {
        const float b = dot(pos, sun_dir), c = dot(pos, pos) - radius * radius;
        if (c > 0.0f && b > 0.0f) return -1.0f;
        const float discr = b * b - c;
        if (discr < 0.0f) return -1.0f;
        return discr > b * b ? -b + sqrtf(discr) : -b - sqrtf(discr);
    };
    if (sphere(6.360f) > 0.0f) return v3s(0.0f);
    const float atmosphere_distance = sphere(6.460f);
    float t = 0.0f, i = 0.0f;
    V3 transmittance = v3s(1.0f);
    while (i < 40.0f) {
        const float new_t = ((i + 0.3f) / 40.0f) * atmosphere_distance;
        const float dt = new_t - t;
        t = new_t;
        const V3 new_pos = pos + t * sun_dir;
        const float altitude_km = (length(new_pos) - 6.360f) * 1000.0f;
        const float rayleigh_density = exp_(-altitude_km / 8.0f), mie_density = exp_(-altitude_km / 1.2f);
        const V3 rayleigh_scattering = v3(5.802f, 13.558f, 33.1f) * rayleigh_density;
        const float rayleigh_absorption = 0.0f;  // RAYLEIGH_ABSORPTION_BASE (0.0) * density, folded: 0 * inf must not poison the LUT (DESIGN.md deviation 9)
        const float mie_scattering = 3.996f * mie_density, mie_absorption = 4.4f * mie_density;
        const V3 ozone_absorption = v3(0.650f, 1.881f, 0.085f) * fmax_(1.0f - fabsf(altitude_km - 25.0f) / 15.0f, 0.0f);
        const V3 extinction = rayleigh_scattering + v3s(rayleigh_absorption) + v3s(mie_scattering) + v3s(mie_absorption) + ozone_absorption;
        const V3 arg = -dt * extinction;
        transmittance = transmittance * v3(exp_(arg.x), exp_(arg.y), exp_(arg.z));
        i += 1.0f;
    }
    return transmittance;
}

// ------------------------------------------------------------------ small containers
struct SlotRanges {  // utils/allocator.rs
    std::vector<std::pair<size_t, size_t>> free_; bool unsorted = false;
    void give(size_t b, size_t e) { if (!free_.empty()) unsorted |= b <= free_.back().second; free_.push_back({b, e}); }
    bool take(size_t len, size_t* b, size_t* e) {
        if (unsorted && !free_.empty()) {
            std::stable_sort(free_.begin(), free_.end(), [](const auto& l, const auto& r) { return l.first < r.first; });
            for (size_t i = 0; i + 1 < free_.size();) {
                if (free_[i].second == free_[i + 1].first) { free_[i].second = free_[i + 1].second; free_.erase(free_.begin() + i + 1); }
                else i++;
            }
        }
        unsorted = false;
        for (size_t i = 0; i < free_.size(); i++) {
            const size_t have = free_[i].second - free_[i].first;
            if (have < len) continue;
            *b = free_[i].first; *e = *b + len;
            if (have == len) free_.erase(free_.begin() + i); else free_[i].first += len;
            return true;
        }
        return false;
    }
};

// Pinned staging for the uploads of st_tick: what a tick sends is copied into one slot of page-locked memory and goes to
// the device from there, so st_tick does not have to wait for the stream before the caller may touch the scene again —
// with a scene that changes every frame the host then runs a frame ahead of the GPU instead of in lock-step with it.
// Three slots: a slot is reused only after the copies issued from it have finished (its event).
struct StagingRing {
    static constexpr int kSlots = 3;
    static constexpr size_t kMaxSlotBytes = (size_t)256 << 20;  // larger ticks go from pageable memory and join the stream
    struct Slot { char* mem = nullptr; size_t capacity = 0, used = 0; hipEvent_t done = nullptr; bool pending = false; };
    Slot slots[kSlots];
    int cur = 0;
    size_t wanted = 0;   // bytes the last tick asked for: the next slot is grown to hold that much
    bool enabled = true;

    void begin_tick() {
        if (!enabled) return;
        cur = (cur + 1) % kSlots;
        Slot& s = slots[cur];
        if (s.pending) { (void)hipEventSynchronize(s.done); s.pending = false; }
        s.used = 0;
        const size_t want = std::min(kMaxSlotBytes, std::max<size_t>(wanted + wanted / 4, (size_t)1 << 20));
        if (s.capacity < want) {
            if (s.mem) (void)hipHostFree(s.mem);
            s.mem = nullptr; s.capacity = 0;
            void* m = nullptr;
            if (hipHostMalloc(&m, want, hipHostMallocDefault) == hipSuccess) { s.mem = static_cast<char*>(m); s.capacity = want; }
            else (void)hipGetLastError();
        }
        wanted = 0;
    }
    // a page-locked copy of [src, src + bytes), or nullptr when the slot cannot take it (the caller then uploads from `src`
    // and joins the stream)
    const void* stage(const void* src, size_t bytes) {
        wanted += (bytes + 255) & ~(size_t)255;
        if (!enabled) return nullptr;
        Slot& s = slots[cur];
        const size_t at = (s.used + 255) & ~(size_t)255;
        if (!s.mem || at + bytes > s.capacity) return nullptr;
        memcpy(s.mem + at, src, bytes);
        s.used = at + bytes;
        return s.mem + at;
    }
    int end_tick(hipStream_t stream) {
        if (!enabled) return ST_OK;
        Slot& s = slots[cur];
        if (s.used == 0) return ST_OK;
        if (!s.done) ST_HIP(hipEventCreateWithFlags(&s.done, hipEventDisableTiming));
        ST_HIP(hipEventRecord(s.done, stream));
        s.pending = true;
        return ST_OK;
    }
    void release() {
        for (Slot& s : slots) {
            if (s.pending) (void)hipEventSynchronize(s.done);
            if (s.done) (void)hipEventDestroy(s.done);
            if (s.mem) (void)hipHostFree(s.mem);
            s = Slot();
        }
    }
};

struct DeviceArray {
    void* ptr = nullptr; size_t capacity = 0;
    // `pageable` is set when the copy had to be issued straight from `src`: the caller joins the stream before `src` changes
    int upload(const void* src, size_t bytes, hipStream_t stream, StagingRing& ring, bool* pageable) {
        if (bytes > capacity) {
            if (ptr) ST_HIP(hipFree(ptr));
            capacity = std::max<size_t>(bytes * 3 / 2, 4096);
            ST_HIP(hipMalloc(&ptr, capacity));
        }
        return upload_range(src, 0, bytes, stream, ring, pageable);
    }
    // part of an array that is already on the device: bytes [offset, offset + bytes) of `base`
    int upload_range(const void* base, size_t offset, size_t bytes, hipStream_t stream, StagingRing& ring, bool* pageable) {
        if (!bytes) return ST_OK;
        const void* src = static_cast<const char*>(base) + offset;
        const void* staged = ring.stage(src, bytes);
        if (!staged) *pageable = true;
        ST_HIP(hipMemcpyAsync(static_cast<char*>(ptr) + offset, staged ? staged : src, bytes, hipMemcpyHostToDevice, stream));
        return ST_OK;
    }
    void release() { if (ptr) (void)hipFree(ptr); ptr = nullptr; capacity = 0; }
};

// ------------------------------------------------------------------ per-camera state (camera_controller/buffers.rs)
constexpr int kInternalPlanes = 4;  // decoded-surface twins A/B (KArgs::sn / psn) + the pair the variance pass writes ahead of the strides-1+2 wavelet launch
struct CameraState {
    StCamera desc{};
    GpuCamera curr{}, prev{};
    uint64_t handle = 0;   // the StHandle this camera is known by (st_dist.cpp keys its partitions by it)
    uint32_t frame = 0, row0 = 0, row1 = 0, col0 = 0, col1 = 0;   // [row0,row1) x [col0,col1): the window this engine renders (st_camera_set_window)
    uint32_t out_format = 0;  // StOutputFormat (camera.rs:170-175 viewport.format)
    void* slab = nullptr; size_t slab_bytes = 0;
    float4* plane[ST_BUF_COUNT + kInternalPlanes] = {};   // + the two decoded-surface twins (KArgs::sn / psn), internal only
    size_t plane_bytes[ST_BUF_COUNT + kInternalPlanes] = {};
    unsigned long long* tile_mask = nullptr; size_t tile_mask_tiles = 0;  // two arrays of one u64 per 8x8 tile (KArgs::tile_mask, KArgs::gi_late_mask)
    unsigned long long* counters = nullptr;  // KS_COUNT x kCounterLines x 8 u64 (one 64-B line each: {rays, traversal bytes, pad})
    unsigned long long profiled_traversal_bytes[KS_COUNT] = {};  // part of counters[..][1] already reported by st_profile_read
    // The two-stream frame pipeline (render()) belongs to the camera: its side stream and the events that order frame N+1's
    // passes behind frame N's are per camera, so cameras rendered on different caller streams never wait on — or race
    // with — each other's frames.
    hipStream_t side_stream = nullptr;
    hipEvent_t ev_di_head = nullptr, ev_gi_done = nullptr, ev_prim_ok = nullptr, ev_frame_done = nullptr, ev_setup = nullptr;
    bool have_prev_frame_events = false;
    // GI history hand-over without the copy. gi_resolving ends every frame by copying the frame's source reservoirs into
    // GI_RESERVOIRS_0, next frame's history (gi_resolving.rs:60-66): 128 B per pixel of pure copy. When the source is the
    // temporal pass's output (GI_RESERVOIRS_1, four frames in six) and the whole pass graph runs, the engine swaps the two
    // plane pointers instead: GI_RESERVOIRS_0 takes over the storage temporal resampling wrote, and GI_RESERVOIRS_1 — which
    // the next temporal pass overwrites completely before anything reads it — gets the old history's storage. Until then
    // reading GI_RESERVOIRS_1 back returns GI_RESERVOIRS_0's storage (`gi_aliased`; st_camera_read_buffer), and anything
    // that could observe the difference (a pass mask, st_camera_write_buffer) first makes the copy for real
    // (`materialize_gi_history`).
    uint32_t last_lean = 0; bool last_lean_composed = false;   // KArgs::lean of the last frame: which planes it left unwritten (st_camera_buffer_stale)
    bool gi_aliased = false;
    bool surface_map_replaced[2] = {false, false};  // st_camera_write_buffer replaced PRIM_SURFACE_MAP_A / _B: regenerate its decoded twin before the next frame
    // Present hand-over (st_camera_present_copy): composed frames leave for host memory on a stream of their own, behind the
    // frame that produced them, while the next frame's kernels run. Two copies may be in flight (the caller alternates two
    // output buffers); a render into a buffer whose copy is still pending is ordered behind that copy.
    struct PresentSlot { const void* src = nullptr; void* dst = nullptr; hipEvent_t ev_src = nullptr, ev_done = nullptr; bool pending = false; };
    hipStream_t present_stream = nullptr;
    PresentSlot present[2];
    uint32_t present_next = 0;
};
inline size_t plane_texels_per_pixel(int id) {
    if (id >= ST_BUF_DI_RESERVOIRS_0 && id <= ST_BUF_DI_RESERVOIRS_2) return 2;
    if (id >= ST_BUF_GI_RESERVOIRS_0 && id <= ST_BUF_GI_RESERVOIRS_3) return 4;
    if (id == ST_BUF_REF_HITS) return 2;
    if (id == ST_BUF_REF_RAYS) return 3;
    return 1;
}

constexpr size_t kCounterWordsPerSlot = (size_t)kCounterLines * 8;
constexpr size_t kCounterBytes = sizeof(unsigned long long) * kCounterWordsPerSlot * KS_COUNT;
inline int materialize_gi_history(CameraState& c) {
    if (!c.gi_aliased) return ST_OK;
    ST_HIP(hipDeviceSynchronize());
    ST_HIP(hipMemcpy(c.plane[ST_BUF_GI_RESERVOIRS_1], c.plane[ST_BUF_GI_RESERVOIRS_0], c.plane_bytes[ST_BUF_GI_RESERVOIRS_0], hipMemcpyDeviceToDevice));
    c.gi_aliased = false;
    return ST_OK;
}
// sums the per-line counters of every kernel slot into host[2*slot + {0: rays, 1: traversal bytes}]
inline int read_counters(const CameraState& c, unsigned long long* host /* 2*KS_COUNT */) {
    std::vector<unsigned long long> raw(kCounterWordsPerSlot * KS_COUNT);
    hipError_t err = hipMemcpy(raw.data(), c.counters, kCounterBytes, hipMemcpyDeviceToHost);
    if (err != hipSuccess) return fail(ST_ERR_HIP, std::string("hipMemcpy(counters): ") + hipGetErrorString(err));
    for (int s = 0; s < KS_COUNT; s++) {
        unsigned long long rays = 0, bytes = 0;
        for (uint32_t l = 0; l < kCounterLines; l++) { rays += raw[(size_t)s * kCounterWordsPerSlot + l * 8]; bytes += raw[(size_t)s * kCounterWordsPerSlot + l * 8 + 1]; }
        host[2 * s] = rays; host[2 * s + 1] = bytes;
    }
    return ST_OK;
}


struct ProfileRecord { int slot; hipEvent_t start, stop; double bytes; uint32_t launches; bool owns_start; };

struct Light112 { GpuLight g; };

StTuning default_tuning();  // st_engine.cpp
struct DistState;           // st_dist.cpp: rank / world, transport, per-camera partition
int dist_partition(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, uint32_t rank, StDistRect* owned);
int dist_window(uint32_t width, uint32_t height, const StDistRect* owned, uint32_t apron, StDistRect* window);
int dist_grid(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, StDistGrid* out);
int dist_grid_tile(const StDistGrid* g, uint32_t rank, StDistRect* owned);
int dist_grid_rebalance(uint32_t width, uint32_t height, const StDistGrid* cur, const float* cost, uint32_t max_step, StDistGrid* out);

struct Engine {
    int device = -1;
    bool has_device = false;
    uint64_t base_seed = 0;
    uint32_t frame = 1;  // lib.rs:152

    // meshes / materials / instances / triangles
    std::unordered_map<uint64_t, std::vector<StMeshTriangle>> meshes;
    // Device bake (StTuning::device_bake; k_bvh.hip k_bvh_bake): object-space meshes live on the device too (24 floats per triangle, appended
    // when an instance of the mesh is first moved on the device; a re-inserted mesh is a new version and is appended again).
    std::unordered_map<uint64_t, uint64_t> mesh_version; uint64_t next_mesh_version = 1;
    struct DeviceMeshRec { size_t first, count; uint64_t version; };
    std::unordered_map<uint64_t, DeviceMeshRec> device_meshes;
    std::vector<float> mesh_store_host; DeviceArray d_mesh_store; size_t mesh_store_uploaded = 0;   // floats
    bool moved_on_device = false;      // this tick's refresh left the moved instances to the device
    bool materials_changed_this_tick = false;
    bool instance_removed = false;     // since the last refresh: the set of triangle slots changed, the tree will be rebuilt
    uint64_t device_bakes = 0, device_baked_triangles = 0;
    bool any_host_stale() const { for (const auto& i : instances) if (i.host_stale) return true; return false; }
    void bake_stale_on_host();         // brings the host arrays up to date (rebuilds, host refits, full uploads and debug reads need them)

    std::vector<StMaterial> materials; std::unordered_map<uint64_t, uint32_t> material_slot; SlotRanges material_free; bool materials_dirty = false;
    std::vector<GpuMaterial> gpu_materials; std::vector<uint32_t> material_base_packed;
    struct InstanceRec {
        uint64_t id, mesh, material; Affine xform, xform_inv, prev_xform; bool dirty; uint32_t xslot;
        // what the triangle slots of this instance were baked from (device bake: a later change of the transform alone can be baked on the device)
        uint64_t baked_mesh = 0, baked_mesh_version = 0; uint32_t baked_material = 0xffffffffu; bool baked = false;
        bool host_stale = false;   // moved on the device since the host arrays (triangles, prims, tri_geo / attr / bounds) were last baked
    };
    std::vector<InstanceRec> instances; bool instances_dirty = false;
    // per-instance transforms for primary visibility's prev_point (the reference's per-draw push constants,
    // passes/prim_raster.rs:196-230): 8 float4 per stable slot — curr_xform_inv (x, y, z axes, translation), then prev_xform.
    // tri_attr[4 t + 3].w holds the slot of the instance that owns triangle t.
    std::vector<float4> instance_xforms; std::vector<uint32_t> xslot_free;
    std::map<uint64_t, std::pair<size_t, size_t>> instance_triangles; SlotRanges triangle_free;
    std::vector<HostTriangle> triangles; std::vector<BuildPrim> prims; std::vector<uint8_t> prim_alive;
    size_t live_prims_ = 0;   // how many of prim_alive are set (kept by refresh_instances / drop_instance_triangles: device_build_possible asks every tick)
    std::vector<float4> tri_geo, tri_attr, tri_bounds, bvh_stream, bvh_upload_;  // tri_bounds: (lo, hi) per triangle slot (device refit)
    BvhBuild bvh;
    bool scene_uploaded = false;
    // BVH refresh policy (st_set_bvh_refresh). Refit: while the set of (triangle slot, material) pairs and the Blend flags
    // are what the last build saw — i.e. instances only moved — keep the tree and recompute the boxes bottom-up.
    int bvh_refresh_mode = ST_BVH_AUTO;   // (host-only engines, the exact build and observed contract streams: what ST_BVH_REBUILD does)
    bool have_topology = false; uint64_t topology_signature = 0;
    std::vector<uint32_t> internal_positions;  // stream offsets of the internal nodes, ascending (parents before children)
    uint64_t refits = 0, rebuilds = 0;
    uint64_t device_refits = 0;    // ticks whose boxes were recomputed on the device
    uint64_t tree_version = 0;     // bumped by every rebuild (ST_BVH_REFIT_DEVICE: a scene copy whose arrays are of this version can be refitted in place)
    bool host_stream_stale = false;  // device refits happened since bvh_stream's boxes were last recomputed (debug reads and full uploads refit it first)
    std::vector<float4> readback_; uint32_t live_bvh_texels = 0;  // st_debug_read_scene(6)
    std::vector<uint32_t> entry_of_tri_, parent_, refit_local_, refit_items_, refit_batch_off_;  // host images of the device refit's index arrays (index_device_tree)
    std::vector<uint32_t> readback_levels_;
    std::vector<std::pair<uint32_t, uint32_t>> refit_levels_;  // (first batch, batches) of each launch, leaves first
    // Deepest chain of internal nodes in the uploaded stream = the most entries a traversal can have pending (every internal
    // node on the path may push its far child). The kernels' per-lane stack holds kBvhStackSize entries (strolle-gpu/src/lib.rs:76;
    // the reference indexes past the end there, here a push beyond the end is dropped): a deeper tree is reported, not hidden.
    // (round 5) Up to kBvhStackSizeDeep the launches take a deeper stack instead (dynamic LDS): `stack_entries` is what this tree's launches hold.
    uint32_t bvh_stack_need = 0, stack_entries = (uint32_t)kBvhStackSize; bool bvh_depth_warned = false, bvh_too_deep_unreported = false;
    void measure_stack_need();
    // Device form of the stream (st_types.h "device BVH stream"): every entry four texels — an internal node as the
    // serializer wrote it (far pointer remapped), a leaf entry followed by its triangle's hit-test record — so that one
    // four-texel fetch serves a traversal step of either kind. Entry k starts at texel 4 k.
    std::vector<uint32_t> expand_map_;  // scratch: offset in bvh_stream -> texel pointer in bvh_upload_ (entry starts only)
    uint32_t device_bvh_len = 0; bool device_root_is_leaf = false;
    void expand_stream();
    void index_device_tree();
    std::vector<uint8_t> internal_start_;  // scratch of measure_stack_need: 1 where an internal node begins
    bool is_internal_start(size_t p) const { return p < internal_start_.size() && internal_start_[p]; }
    void mark_internal_starts();

    // images: a single linear RGBA8 atlas of the reference's extent (images.rs:28-29); rectangles from st_atlas.h
    static constexpr uint32_t kAtlasW = 8192, kAtlasMaxH = 8192;
    uint32_t atlas_w = 0, atlas_h = 0; std::vector<uint8_t> atlas; bool atlas_dirty = false;
    struct ImageRec { uint32_t x, y, w, h; };
    std::unordered_map<uint64_t, ImageRec> images; AtlasShelves atlas_rects{kAtlasW, kAtlasMaxH};
    // ImageData::Texture (image.rs:46-59): pixels that live in device memory. Static ones are copied into the atlas once
    // (and mirrored into the host copy, which stays the source of every later re-upload); dynamic ones at every tick
    // (images.rs:187-213). std::map: copies are issued in handle order.
    struct DeviceImage { const void* pixels; size_t pitch; bool dynamic, pending; };
    std::map<uint64_t, DeviceImage> device_images;
    StagingRing staging;
    // a tick queued copies without joining the stream: ev_tick marks their end, the next frame's streams wait for it
    bool tick_work_in_flight = false; hipEvent_t ev_tick = nullptr;

    // lights (lights.rs): slot 0 is the sun
    std::vector<GpuLight> light_buffer; std::map<int64_t, uint32_t> light_slot;
    std::vector<int64_t> lights_created, lights_updated; std::map<int64_t, uint32_t> lights_remapped; std::vector<uint32_t> lights_killed;
    uint32_t next_light_id = 1;
    std::vector<GpuLight> gpu_lights, uploaded_lights;
    bool sync_every_tick = false;
    float sun_azimuth = 0.0f, sun_altitude = 0.35f; bool sun_dirty = true;
    uint32_t light_count = 0; V3 sun_dir_ = v3s(0.0f);

    std::vector<uint8_t> blue_noise; bool blue_noise_dirty = true;
    bool atmosphere_initialized = false, sky_known = false; float known_sun_altitude = 0.0f;  // passes/atmosphere.rs:14-15,78-110

    DeviceArray d_byte_luts, d_atlas, d_blue_noise, d_transmittance, d_scattering, d_sky;
    // The arrays a scene change rewrites exist twice. A tick that changes the scene fills the copy no frame in flight reads,
    // on a stream of its own, while the previous frame still renders from the other one; the next frame switches over.
    // (Updating in place would have to wait for the previous frame, and the next frame's primary rays with it.)
    struct SceneSet {
        DeviceArray bvh, tri_attr, xforms, materials, base_packed;
        DeviceArray bvh_compact; uint32_t compact_entries = 0;   // k_bvh.hip k_bvh_compact: 48 B per entry, regenerated whenever `bvh` changed
        // k_bvh.hip k_bvh_wide: 4-wide nodes (64 B) + leaf records (48 B), regenerated whenever `bvh` changed; the topology arrays only when the tree was rebuilt
        DeviceArray bvh_wide /* the nodes, then the leaf records */, wide_topo, wide_leaf_entry; uint32_t wide_nodes = 0, wide_leaves = 0, wide_root = 0, wide_links16 = 0, wide_for_entries = 0;
        uint64_t wide_topology_serial = 0;   // which build_wide_topology() result this copy holds
        // ST_BVH_BUILD_DEVICE (k_lbvh.hip): this copy's wide stream was built on the device from its own triangle arrays; `bvh` (the contract
        // stream) is then stale and no launch may read it. lb_live: live triangles of that build. The rest is the builder's scratch.
        bool device_built = false; uint32_t lb_live = 0; uint64_t tri_info_serial = 0;
        uint64_t lb_serial = ~0ull; uint32_t lb_slots = 0;   // tri_info_serial_ / triangle slots of the copy's last device BUILD (a refit needs both unchanged)
        DeviceArray tri_info, lb_keys_a, lb_keys_b, lb_temp, lb_seg, lb_children, lb_node_box, lb_small;
        // ST_BVH_REFIT_DEVICE: what k_bvh.hip needs beside the stream — per triangle slot the hit-test record, the bounds and the
        // device entry that holds it; per entry its parent (entry << 1 | child slot); the leaf runs; an arrival counter per entry.
        // tree_version says which build of the tree these (and the stream's topology) belong to.
        DeviceArray tri_geo, tri_bounds, entry_of_tri, parent, refit_local, refit_items, refit_batch_off;
        uint64_t tree_version = 0;
        size_t dirty_lo = SIZE_MAX, dirty_hi = 0; bool tri_full = true;  // what this copy lacks of the host's triangle arrays
        std::vector<uint64_t> pending_moves;   // instances moved on the device whose current transform this copy has not baked yet
        DeviceArray bake_jobs, bake_starts;
        hipEvent_t free_ev = nullptr; bool busy = false;  // busy: frames reading this copy were enqueued since it was written; free_ev ends the last
        bool valid = false;
    };
    SceneSet sets[2]; int live = 0;
    // the light table alternates the same way, on its own schedule (a light that moves every frame does not resend the scene)
    struct LightSet { DeviceArray buf; hipEvent_t free_ev = nullptr; bool busy = false; };
    LightSet light_sets[2]; int live_lights = 0; bool lights_uploaded = false, lights_alternating = false;
    bool alternating = false, mixed_render_streams = false;
    hipStream_t copy_stream = nullptr, last_render_stream = nullptr; bool rendered_before = false; hipEvent_t ev_copy = nullptr; bool copy_in_flight = false;

    std::unordered_map<uint64_t, std::unique_ptr<CameraState>> cameras; uint64_t next_camera = 0;

    // Which build of the kernels this engine launches (st_kernels.h): fast arithmetic by default, the bit-exact build on request
    // (st_engine_set_arithmetic, or ST_EXACT=1 in the environment when the engine is created).
    int arithmetic = ST_ARITH_FAST;
    Launchers L = launchers_fast();
    std::vector<uint64_t> last_launches;  // pass bits of every launch the last render considered (st_debug_last_launches)
    uint64_t launch_filter = ~0ull;  // st_debug_set_launch_filter: which launches of the whole frame's graph (by their ordinal in the serial order) are enqueued (tools/pair_matrix.py)
    uint64_t pass_mask = ~0ull;  // st_debug_set_pass_mask: which reference passes a render executes (parity tests run one launch at a time)
    // Scheduling / tuning switches (include/strolle_hip.h StTuning says what each selects; st_engine.cpp holds the defaults and
    // the environment overrides). Notes that belong to the implementation:
    //  * lean_frame: KArgs::lean, st_types.h kLean* — fast build + whole pass graph + Image{denoise}: planes nothing reads again are
    //    not stored; st_camera_read_buffer of those planes returns what an earlier frame or launch left there.
    //  * tile_map_denoise = 2 keeps the halo rows of the LDS windows and the a-trous taps in one XCD's L2. Measured on one box: with
    //    mode 1 the two-stream frame is 1.347 instead of 1.373 ms, but a wavelet launch moves 340 instead of 205 MB through the
    //    fabric (algorithmic: 174 MB) and takes 61 instead of 56 us on its own.
    //  * tile_map = 1 measured best on MI355X with the current kernels (2 was, before the LDS-staged denoiser).
    StTuning tuning;
    uint32_t exp_flags = 0;   // KArgs::exp_flags (ST_EXP): same-box A/B of an experiment before it is kept or archived
    bool profiling = false;       // st_profile_enable bit 0: per-kernel event timing (serial execution)
    bool count_bytes = false;     // st_profile_enable bit 1: traversal-byte counters
    bool profile_kernel_events = false;  // st_profile_enable bit 3: every launch carries its own start / stop events (hipExtLaunchKernelGGL): no event packets between kernels
    bool profile_group_atrous = false;  // st_profile_enable bit 2: the a-trous chain's back-to-back launches share ONE event pair (an event between two kernels costs the second one 3-15 us)
    std::vector<ProfileRecord> profile_records; std::vector<hipEvent_t> event_pool;
    StKernelProfile profile_totals[KS_COUNT];

    Engine();
    int set_tuning(const StTuning& t);
    // multi-GPU (st_dist.cpp)
    DistState* dist = nullptr;
    void release_dist();
    void dist_forget_camera(uint64_t handle);
    // the wide stream's topology (st_bvh_refresh.cpp build_wide_topology), from bvh_upload_: 8 words per node (4 box sources, 4 links) and the
    // contract entry of every leaf record; wide_serial_ counts the builds
    std::vector<uint32_t> wide_topo_, wide_leaf_entry_; uint32_t wide_root_ = 0, wide_stack_need_ = 0; uint64_t wide_serial_ = 0; uint64_t wide_built_for_ = ~0ull;
    void build_wide_topology();
    int refresh_wide_stream(SceneSet& t, hipStream_t up, bool topology_changed, bool* pageable);
    int dist_set_partition(uint64_t handle, CameraState& c, uint32_t cols, uint32_t apron);
    int dist_set_grid(uint64_t handle, CameraState& c, const StDistGrid& grid, uint32_t apron);
    int dist_gather(uint64_t handle, CameraState& c, const void* frame, void* full, hipStream_t stream);
    int dist_wait(uint64_t handle, const void* frame, hipStream_t stream, bool host);
    void dist_guard(uint64_t handle, const void* out, hipStream_t s);
    int dist_gather_ms(uint64_t handle, float* ms);
    void reset_profile_totals();
    ~Engine();
    static void release_camera(CameraState& c);
    int present_copy(CameraState& c, const void* src, void* dst, size_t bytes, hipStream_t stream);
    int present_ready(CameraState& c, const void* dst, int wait, int* ready);

    float4 image_rect(uint64_t h) const;
    void rebuild_gpu_materials();

    // ---- lights (lights.rs:49-172, light.rs:25-79)
    static void note(std::vector<int64_t>& v, int64_t k) { if (std::find(v.begin(), v.end(), k) == v.end()) v.push_back(k); }
    void overwrite_light(uint32_t slot, int64_t key, GpuLight g);
    void insert_light(uint64_t id, const StLight& l);
    void remove_light(uint64_t id);
    void snapshot_lights();

    void drop_instance_triangles(uint64_t id);
    void bake(const StMeshTriangle& t, const InstanceRec& inst, uint32_t material, size_t slot);
    struct BakeJob { const std::vector<StMeshTriangle>* mesh; const InstanceRec* inst; uint32_t material; size_t first, count; };
    bool refresh_instances();
    void bake_jobs_on_host(const std::vector<BakeJob>& jobs, size_t total);
    int bake_on_device(SceneSet& t, hipStream_t up, bool* pageable);
    int refresh_compact_stream(SceneSet& t, hipStream_t up);

    uint64_t topology_of(const std::vector<uint8_t>& blend) const;
    void index_stream();
    Aabb subtree_box(size_t p) const;
    void refit_node(size_t p);
    void refit_span(size_t begin, size_t end);
    // One thread: at 134 k triangles the sweep is about a millisecond, less than starting a worker pool for it would buy back.
    void refit_stream() { refit_span(0, bvh_stream.size()); }
    bool device_refit_possible() const { return bvh_refresh_mode == ST_BVH_REFIT_DEVICE && has_device; }
    // ST_BVH_BUILD_DEVICE: the tree of a changed scene is built on the device (k_lbvh.hip) while nothing observes the contract stream
    bool host_tree_stale = false;    // the host's binned-SAH tree (bvh_stream and everything derived from it) is behind the scene
    uint64_t device_builds = 0, device_tree_refits = 0;   // ticks answered by a device build / by a refit of the device-built tree (moves only)
    bool device_tree_refit_now = false; uint32_t device_refits_since_build = 0;
    size_t info_dirty_lo_ = SIZE_MAX, info_dirty_hi_ = 0; bool info_full_ = true;   // slots whose word may have changed since tri_info_ was listed (spawned, removed); everything (materials changed)
    void mark_info_dirty(size_t b, size_t e) { info_dirty_lo_ = std::min(info_dirty_lo_, b); info_dirty_hi_ = std::max(info_dirty_hi_, e); }
    std::vector<uint32_t> tri_info_; uint32_t tri_info_live_ = 0; uint64_t tri_info_serial_ = 1, tri_info_built_for_ = 0;   // per slot: live | Blend << 1 | material << 2; the serial counts what can change it
    // A wide walk that found its stack full drops the push and says so in one of two sticky words of page-locked host memory (st_device.h
    // wide_walk_overflowed; KArgs::walk_flags). st_tick reads them: a per-lane walk's overflow re-arms every later launch with a deeper stack
    // (24 -> 32 -> 48 -> 56 entries: dynamic LDS, fewer waves per SIMD), the packet's (64 entries, one VGPR) hands primary visibility back to the
    // per-lane walk; either way that tick returns ST_ERR_BVH_TOO_DEEP once (StTuning::allow_deep_bvh: a line on stderr) — the frames rendered in
    // between may have missed geometry behind the dropped subtrees.
    volatile uint32_t* walk_flags_host = nullptr; uint32_t* walk_flags_dev = nullptr;
    uint32_t wide_stack_rearmed = 0u;   // 0: StTuning::wide_stack_entries (0 = 24) as set; otherwise the entries an overflow re-armed the wide walks with
    bool packets_overflowed = false;    // the packet walk overflowed once on this engine: primary rays keep the per-lane walk
    uint64_t walk_overflows = 0; bool walk_overflow_unreported = false;
    uint32_t wide_stack_entries_now() const { return wide_stack_rearmed ? wide_stack_rearmed : (tuning.wide_stack_entries ? tuning.wide_stack_entries : (uint32_t)kBvhStackSize); }
    void note_walk_overflow();
    // ST_BVH_AUTO's choice of a scene's FIRST tree (st_tick.cpp device_build_possible): host_leaf_run_weight = the surface-area-weighted mean length of the host
    // tree's leaf runs (rebuild_host_tree); above kAutoLeafRunLimit the device builder's tree renders faster (profiles/r06_tree_choice*.txt)
    static constexpr float kAutoLeafRunLimit = 3.4f;
    float host_leaf_run_weight = 1.0f; bool auto_first_on_device = false;
    bool device_build_possible() const;
    int build_on_device(SceneSet& t, hipStream_t up, bool* pageable);
    int reserve_device_builder(SceneSet& t, size_t slots, uint32_t live);
    void rebuild_host_tree(bool timing);

    int tick(hipStream_t stream);

    static GpuCamera serialize_camera(const StCamera& c);
    int allocate_camera(CameraState& c);

    hipEvent_t take_event();
    // One event pair per RUN of consecutive launches of the same slot on the same stream (the five a-trous launches, say):
    // an event between two kernels makes the second wait for a barrier packet, which adds microseconds to every launch
    // it brackets, so back-to-back launches of one slot are timed as one interval and divided by their count.
    // Consecutive runs on one stream share the event between them (the stop of one is the start of the next).
    struct OpenScope { int slot = -1; hipStream_t stream = nullptr; hipEvent_t start{}; bool owns_start = true; double bytes = 0; uint32_t launches = 0; } open_scope;
    void profile_begin(int slot, hipStream_t s, double bytes);
    hipEvent_t profile_close();
    int drain_profile();

    // a composition into a buffer whose present copy has not finished waits for that copy (callers that alternate two
    // buffers never meet this)
    static void present_guard(CameraState& c, const void* out, hipStream_t s) {
        for (auto& p : c.present) if (p.pending && p.src == out) (void)hipStreamWaitEvent(s, p.ev_done, 0);
    }

    int render(CameraState& c, void* out, hipStream_t stream);
};

}  // namespace st
