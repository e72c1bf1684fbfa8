// st_engine.cpp — host engine of libstrolle_hip.so: scene stores, world-space baking + BVH refresh, device
// buffer management, the per-frame pass graph and the C ABI (include/strolle_hip.h).
//
// Behavioural contract: strolle/src/lib.rs (Engine), camera_controller.rs (pass order), lights.rs / materials.rs /
// instances.rs / triangles.rs (stores), camera.rs (camera uniform). The wgpu plumbing of the reference
// (bind groups, mapped buffers, textures) is replaced by plain device allocations and pointer swaps.
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
#include "st_atlas.h"
#include "st_bvh.h"
#include "st_kernels.h"

namespace st {

static thread_local std::string g_last_error;
static int fail(int status, const std::string& msg) { g_last_error = msg; return status; }

#define ST_HIP(call)                                                                                      \
    do {                                                                                                  \
        hipError_t err_ = (call);                                                                         \
        if (err_ != hipSuccess) return fail(ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(err_)); \
    } while (0)

// ------------------------------------------------------------------ host maths (glam order; see st_math.h)
static M4 m4_from_cols(const float* a) { M4 m; for (int i = 0; i < 4; i++) m.c[i] = make_float4(a[4 * i], a[4 * i + 1], a[4 * i + 2], a[4 * i + 3]); return m; }
static M4 m4_mul(const M4& a, const M4& b) { M4 r; for (int i = 0; i < 4; i++) r.c[i] = mul(a, b.c[i]); return r; }
static M4 m4_inverse(const M4& m) {  // glam 0.24 Mat4::inverse, scalar path
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
static Affine affine_from12(const float* a) { Affine r; r.x = v3(a[0], a[1], a[2]); r.y = v3(a[3], a[4], a[5]); r.z = v3(a[6], a[7], a[8]); r.t = v3(a[9], a[10], a[11]); return r; }
static V3 affine_vec(const Affine& a, V3 v) { V3 r = a.x * v.x; r = r + a.y * v.y; r = r + a.z * v.z; return r; }
static V3 affine_point(const Affine& a, V3 p) { return ((a.x * p.x) + (a.y * p.y) + (a.z * p.z)) + a.t; }
static Affine affine_inverse(const Affine& a) {  // glam Affine3A::inverse
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
static uint32_t seed_hash(uint32_t v) {
    v = v * 747796405u + 2891336453u;
    const uint32_t w = ((v >> ((v >> 28) + 4u)) ^ v) * 277803737u;
    return (w >> 22) ^ w;
}
static uint32_t pass_seed(uint64_t base, uint32_t frame, uint32_t pass_id) {
    return seed_hash((uint32_t)base ^ seed_hash((uint32_t)(base >> 32) ^ seed_hash(frame ^ seed_hash(pass_id))));
}
enum PassSeedId { SEED_DI_SAMPLING = 1, SEED_DI_TEMPORAL = 2, SEED_DI_SPATIAL_PICK = 3, SEED_DI_SPATIAL_SAMPLE = 5, SEED_GI_SAMPLING_A = 8,
                  SEED_GI_SAMPLING_B = 9, SEED_GI_TEMPORAL = 10, SEED_GI_SPATIAL_PICK = 11, SEED_GI_SPATIAL_SAMPLE = 13, SEED_GI_PREVIEW = 14,
                  SEED_REF_SHADING = 200 };

// sun light colour: atmosphere/generate_transmittance_lut.rs:32-59 evaluated on the host (lights.rs:80-95)
static V3 sun_transmittance(V3 pos, V3 sun_dir) {
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
    uint32_t frame = 0, row0 = 0, row1 = 0;
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
static size_t plane_texels_per_pixel(int id) {
    if (id >= ST_BUF_DI_RESERVOIRS_0 && id <= ST_BUF_DI_RESERVOIRS_2) return 2;
    if (id >= ST_BUF_GI_RESERVOIRS_0 && id <= ST_BUF_GI_RESERVOIRS_3) return 4;
    if (id == ST_BUF_REF_HITS) return 2;
    if (id == ST_BUF_REF_RAYS) return 3;
    return 1;
}

constexpr size_t kCounterWordsPerSlot = (size_t)kCounterLines * 8;
constexpr size_t kCounterBytes = sizeof(unsigned long long) * kCounterWordsPerSlot * KS_COUNT;
static int materialize_gi_history(CameraState& c) {
    if (!c.gi_aliased) return ST_OK;
    ST_HIP(hipDeviceSynchronize());
    ST_HIP(hipMemcpy(c.plane[ST_BUF_GI_RESERVOIRS_1], c.plane[ST_BUF_GI_RESERVOIRS_0], c.plane_bytes[ST_BUF_GI_RESERVOIRS_0], hipMemcpyDeviceToDevice));
    c.gi_aliased = false;
    return ST_OK;
}
// sums the per-line counters of every kernel slot into host[2*slot + {0: rays, 1: traversal bytes}]
static int read_counters(const CameraState& c, unsigned long long* host /* 2*KS_COUNT */) {
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

thread_local LaunchEvents g_launch_events;  // st_kernels.h: set around one launch while ST_PROFILE_KERNEL_EVENTS is on

struct ProfileRecord { int slot; hipEvent_t start, stop; double bytes; uint32_t launches; bool owns_start; };

struct Light112 { GpuLight g; };

struct Engine {
    int device = -1;
    bool has_device = false;
    uint64_t base_seed = 0;
    uint32_t frame = 1;  // lib.rs:152

    // meshes / materials / instances / triangles
    std::unordered_map<uint64_t, std::vector<StMeshTriangle>> meshes;
    std::vector<StMaterial> materials; std::unordered_map<uint64_t, uint32_t> material_slot; SlotRanges material_free; bool materials_dirty = false;
    std::vector<GpuMaterial> gpu_materials; std::vector<uint32_t> material_base_packed;
    struct InstanceRec { uint64_t id, mesh, material; Affine xform, xform_inv, prev_xform; bool dirty; uint32_t xslot; };
    std::vector<InstanceRec> instances; bool instances_dirty = false;
    // per-instance transforms for primary visibility's prev_point (the reference's per-draw push constants,
    // passes/prim_raster.rs:196-230): 8 float4 per stable slot — curr_xform_inv (x, y, z axes, translation), then prev_xform.
    // tri_attr[4 t + 3].w holds the slot of the instance that owns triangle t.
    std::vector<float4> instance_xforms; std::vector<uint32_t> xslot_free;
    std::map<uint64_t, std::pair<size_t, size_t>> instance_triangles; SlotRanges triangle_free;
    std::vector<HostTriangle> triangles; std::vector<BuildPrim> prims; std::vector<uint8_t> prim_alive;
    std::vector<float4> tri_geo, tri_attr, tri_bounds, bvh_stream, bvh_upload_;  // tri_bounds: (lo, hi) per triangle slot (device refit)
    BvhBuild bvh;
    bool scene_uploaded = false;
    // BVH refresh policy (st_set_bvh_refresh). Refit: while the set of (triangle slot, material) pairs and the Blend flags
    // are what the last build saw — i.e. instances only moved — keep the tree and recompute the boxes bottom-up.
    int bvh_refresh_mode = ST_BVH_REBUILD;
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
    uint32_t bvh_stack_need = 0; bool bvh_depth_warned = false;
    void measure_stack_need() {
        std::vector<uint8_t> height(bvh_stream.size() + 1, 0);  // internal nodes only; a leaf run has height 0
        uint32_t deepest = 0;
        for (size_t p = bvh_stream.size(); p-- > 0;) {
            // walk backwards; an internal node starts where d0.w == 0 and the 3 texels after it are its own
            if (p + 3 < bvh_stream.size() && f2b(bvh_stream[p].w) == 0u && is_internal_start(p)) {
                const size_t l = p + 4, r = f2b(bvh_stream[p + 1].w);
                const uint32_t h = 1u + std::max<uint32_t>(l < height.size() ? height[l] : 0, r < height.size() ? height[r] : 0);
                height[p] = (uint8_t)std::min<uint32_t>(h, 255u);
                deepest = std::max(deepest, h);
            }
        }
        bvh_stack_need = deepest;
        if (deepest > (uint32_t)kBvhStackSize && !bvh_depth_warned) {
            bvh_depth_warned = true;
            fprintf(stderr, "[strolle-hip] warning: the BVH is %u internal nodes deep; traversal keeps %d pending entries per ray (as the reference does) and drops deeper ones — distant geometry may be missed. st_debug_bvh_depth reports this.\n", deepest, kBvhStackSize);
        }
    }
    // Device form of the stream (st_types.h "device BVH stream"): every entry four texels — an internal node as the
    // serializer wrote it (far pointer remapped), a leaf entry followed by its triangle's hit-test record — so that one
    // four-texel fetch serves a traversal step of either kind. Entry k starts at texel 4 k.
    std::vector<uint32_t> expand_map_;  // scratch: offset in bvh_stream -> texel pointer in bvh_upload_ (entry starts only)
    uint32_t device_bvh_len = 0;
    void expand_stream() {
        const size_t n = bvh_stream.size();
        expand_map_.resize(n);
        size_t entries = 0;
        for (size_t p = 0; p < n; p += f2b(bvh_stream[p].w) == 0u ? 4 : 1) expand_map_[p] = (uint32_t)(4 * entries++);
        bvh_upload_.resize(4 * std::max<size_t>(entries, 1));
        if (!entries) for (float4& t : bvh_upload_) t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        size_t o = 0;
        for (size_t p = 0; p < n; o += 4) {
            if (f2b(bvh_stream[p].w) == 0u) {
                for (int k = 0; k < 4; k++) bvh_upload_[o + k] = bvh_stream[p + k];
                bvh_upload_[o + 1].w = b2f(expand_map_[f2b(bvh_stream[p + 1].w)] * 16u);  // far pointer as a byte offset
                p += 4;
            } else {
                const uint32_t tri = f2b(bvh_stream[p].y);
                bvh_upload_[o] = bvh_stream[p];
                for (int k = 0; k < 3; k++) bvh_upload_[o + 1 + k] = tri_geo[3 * (size_t)tri + k];
                p += 1;
            }
        }
        device_bvh_len = (uint32_t)(4 * entries);
    }
    // Index arrays of the device refit, from the device form of the stream (bvh_upload_, entries of four texels): per triangle
    // slot its leaf entry, per entry its parent (entry << 1 | child slot), and the work list of k_bvh_refit.
    // The refit must not pass data between workgroups inside a launch (XCD L2s are not coherent; an agent-scope fence per node
    // measured 2.1 ms for 208 k triangles), so the tree is cut into TASKS — maximal subtrees with at most kRefitBatch leaves —,
    // each refitted bottom-up by one workgroup that keeps the child boxes in LDS. Finished task roots are the leaves of the next
    // LEVEL (their boxes already sit in the stream, in their parent's entry), one launch per level: two for 208 k triangles.
    // Small tasks share a workgroup (a batch: <= kRefitBatch items, LDS slots numbered within the batch).
    // items: leaf entry of a run's first triangle, or (1 << 31 | entry) for a finished task root; refit_local_[entry]: the LDS slot
    // of an internal node, bit 31 set on a task's root.
    void index_device_tree() {
        const uint32_t n_entries = device_bvh_len / 4u;
        entry_of_tri_.assign(tri_geo.size() / 3u, 0xffffffffu);
        parent_.assign(std::max<uint32_t>(n_entries, 1u), 0xffffffffu);
        refit_local_.assign(std::max<uint32_t>(n_entries, 1u), 0u);
        refit_items_.clear(); refit_batch_off_.assign(1, 0u); refit_levels_.clear();
        std::vector<uint32_t> internals;
        auto internal = [&](uint32_t k) { return f2b(bvh_upload_[4u * (size_t)k].w) == 0u; };
        auto child = [&](uint32_t k, int which) { return which ? f2b(bvh_upload_[4u * (size_t)k + 1u].w) / 64u : k + 1u; };
        for (uint32_t k = 0; k < n_entries; k++) {
            if (internal(k)) {
                parent_[child(k, 0)] = (k << 1) | 0u;
                parent_[child(k, 1)] = (k << 1) | 1u;
                internals.push_back(k);
            } else entry_of_tri_[f2b(bvh_upload_[4u * (size_t)k].y)] = k;
        }
        std::vector<uint32_t> leaves(n_entries, 0u);  // of an unfinished internal node: runs + finished task roots beneath it
        std::vector<uint8_t> finished(n_entries, 0);
        std::vector<uint32_t> stack;
        for (size_t left = internals.size(); left;) {
            for (size_t i = internals.size(); i-- > 0;) {   // children sit behind their parent
                const uint32_t k = internals[i];
                if (finished[k]) continue;
                uint32_t n = 0;
                for (int c = 0; c < 2; c++) { const uint32_t ck = child(k, c); n += internal(ck) && !finished[ck] ? leaves[ck] : 1u; }
                leaves[k] = n;
            }
            const uint32_t first_batch = (uint32_t)refit_batch_off_.size() - 1u;
            uint32_t in_batch = 0, slots = 0;
            for (const uint32_t k : internals) {
                if (finished[k] || leaves[k] > kRefitBatch) continue;
                // finished[] of a task's nodes is set when its root is met, so an unfinished node here has no parent in a task
                if (in_batch + leaves[k] > kRefitBatch) { refit_batch_off_.push_back((uint32_t)refit_items_.size()); in_batch = 0; slots = 0; }
                in_batch += leaves[k];
                stack.assign(1, k);
                while (!stack.empty()) {
                    const uint32_t n = stack.back(); stack.pop_back();
                    refit_local_[n] = slots++ | (n == k ? 0x80000000u : 0u);
                    finished[n] = 1; left--;
                    for (int c = 1; c >= 0; c--) {
                        const uint32_t ck = child(n, c);
                        if (!internal(ck)) refit_items_.push_back(ck);
                        else if (finished[ck]) refit_items_.push_back(ck | 0x80000000u);
                        else stack.push_back(ck);
                    }
                }
            }
            if (in_batch) refit_batch_off_.push_back((uint32_t)refit_items_.size());
            refit_levels_.push_back({first_batch, (uint32_t)refit_batch_off_.size() - 1u - first_batch});
        }
    }
    // OPT-IN (a build with -DST_WIDE_NODES=1 run with ST_WIDE_NODES=1; measured slower, st_device.h trace_any_wide says by how much):
    // 4-wide nodes for the ANY-HIT rays of the fast build: the binary tree's internal nodes collapsed
    // four children at a time (the child with the largest box is opened first), appended behind the device stream; a child is
    // another wide node or a leaf run of the binary stream itself — same leaves, same triangles, so the same occlusion answer.
    // The heatmap, closest-hit rays and the reference's `used_memory` counter keep the binary stream (the stream contract).
    // A wide node is eight texels: min.x[4] min.y[4] min.z[4] max.x[4] max.y[4] max.z[4] child byte offsets[4] (spare);
    // an unused slot holds a box no finite ray reaches.
    bool wide_nodes = false;
    uint32_t device_wide_len = 0;
    void append_wide_nodes() {
        device_wide_len = 0;
        bvh_upload_.resize(device_bvh_len ? device_bvh_len : bvh_upload_.size());
        if (!wide_nodes || device_bvh_len == 0 || f2b(bvh_upload_[0].w) != 0u) return;  // empty scene, or the root is a leaf run
        struct Child { float mn[3], mx[3]; uint32_t off; };
        const uint32_t wide_base = device_bvh_len * 16u;
        auto internal = [&](uint32_t off) { return f2b(bvh_upload_[off / 16u].w) == 0u; };
        auto children_of = [&](uint32_t off, Child* out) {
            const float4* e = &bvh_upload_[off / 16u];
            out[0] = Child{{e[0].x, e[0].y, e[0].z}, {e[1].x, e[1].y, e[1].z}, off + 64u};
            out[1] = Child{{e[2].x, e[2].y, e[2].z}, {e[3].x, e[3].y, e[3].z}, f2b(e[1].w)};
        };
        auto half_area = [](const Child& c) { const float x = c.mx[0] - c.mn[0], y = c.mx[1] - c.mn[1], z = c.mx[2] - c.mn[2]; return x * y + y * z + z * x; };
        std::vector<float4> wide(8);
        std::vector<std::pair<uint32_t, uint32_t>> work{{0u, 0u}};  // (binary internal entry, wide node number)
        while (!work.empty()) {
            const auto [off, node] = work.back(); work.pop_back();
            Child c[4]; int n = 2;
            children_of(off, c);
            while (n < 4) {
                int pick = -1; float best = -1.0f;
                for (int i = 0; i < n; i++) if (internal(c[i].off) && half_area(c[i]) > best) { best = half_area(c[i]); pick = i; }
                if (pick < 0) break;
                Child two[2]; children_of(c[pick].off, two);
                c[pick] = two[0]; c[n++] = two[1];
            }
            uint32_t ref[4];
            for (int i = 0; i < 4; i++) {
                if (i >= n) { c[i] = Child{{kF32Max, kF32Max, kF32Max}, {kF32Max, kF32Max, kF32Max}, 0u}; ref[i] = 0u; continue; }
                if (internal(c[i].off)) {
                    const uint32_t next = (uint32_t)(wide.size() / 8u);
                    wide.resize(wide.size() + 8u);
                    work.push_back({c[i].off, next});
                    ref[i] = wide_base + next * 128u;
                } else ref[i] = c[i].off;
            }
            float4* w = &wide[(size_t)node * 8u];
            for (int k = 0; k < 3; k++) {
                w[k] = make_float4(c[0].mn[k], c[1].mn[k], c[2].mn[k], c[3].mn[k]);
                w[3 + k] = make_float4(c[0].mx[k], c[1].mx[k], c[2].mx[k], c[3].mx[k]);
            }
            w[6] = make_float4(b2f(ref[0]), b2f(ref[1]), b2f(ref[2]), b2f(ref[3]));
            w[7] = make_float4(b2f((uint32_t)n), 0.0f, 0.0f, 0.0f);
        }
        if ((size_t)device_bvh_len * 16u + wide.size() * 16u > 0xffffffffull) return;  // offsets would not fit: binary stream only
        bvh_upload_.insert(bvh_upload_.end(), wide.begin(), wide.end());
        device_wide_len = (uint32_t)wide.size();
    }
    std::vector<uint8_t> internal_start_;  // scratch of measure_stack_need: 1 where an internal node begins
    bool is_internal_start(size_t p) const { return p < internal_start_.size() && internal_start_[p]; }
    void mark_internal_starts() {
        internal_start_.assign(bvh_stream.size(), 0);
        for (size_t p = 0; p < bvh_stream.size();) {
            if (f2b(bvh_stream[p].w) == 0u) { internal_start_[p] = 1; p += 4; } else p += 1;
        }
    }

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
        // ST_BVH_REFIT_DEVICE: what k_bvh.hip needs beside the stream — per triangle slot the hit-test record, the bounds and the
        // device entry that holds it; per entry its parent (entry << 1 | child slot); the leaf runs; an arrival counter per entry.
        // tree_version says which build of the tree these (and the stream's topology) belong to.
        DeviceArray tri_geo, tri_bounds, entry_of_tri, parent, refit_local, refit_items, refit_batch_off;
        uint64_t tree_version = 0;
        size_t dirty_lo = SIZE_MAX, dirty_hi = 0; bool tri_full = true;  // what this copy lacks of the host's triangle arrays
        hipEvent_t free_ev = nullptr; bool busy = false;  // busy: frames reading this copy were enqueued since it was written; free_ev ends the last
        bool valid = false;
    };
    SceneSet sets[2]; int live = 0;
    // the light table alternates the same way, on its own schedule (a light that moves every frame does not resend the scene)
    struct LightSet { DeviceArray buf; hipEvent_t free_ev = nullptr; bool busy = false; };
    LightSet light_sets[2]; int live_lights = 0; bool lights_uploaded = false, lights_alternating = false;
    bool double_buffer = true, alternating = false, mixed_render_streams = false;
    hipStream_t copy_stream = nullptr, last_render_stream = nullptr; bool rendered_before = false; hipEvent_t ev_copy = nullptr; bool copy_in_flight = false;

    std::unordered_map<uint64_t, std::unique_ptr<CameraState>> cameras; uint64_t next_camera = 0;

    // Which build of the kernels this engine launches (st_kernels.h): fast arithmetic by default, the bit-exact build on request
    // (st_engine_set_arithmetic, or ST_EXACT=1 in the environment when the engine is created).
    int arithmetic = ST_ARITH_FAST;
    Launchers L = launchers_fast();
    std::vector<uint64_t> last_launches;  // pass bits of every launch the last render considered (st_debug_last_launches)
    uint64_t pass_mask = ~0ull;  // st_debug_set_pass_mask: which reference passes a render executes (parity tests run one launch at a time)
    bool overlap = true;    // two-stream, cross-frame software pipelining of the Image-mode pass graph (ST_NO_OVERLAP=1 disables)
    bool variance_in_reproject = true;  // ST_NO_VARIANCE_IN_REPROJECT=1: estimate_variance as its own full-screen pass
    bool preview_both = true;  // ST_NO_PREVIEW_BOTH=1: the two GI preview passes as two full-screen launches
    // The lean frame (KArgs::lean, st_types.h kLean*; fast build + whole pass graph + Image-family mode with the denoiser):
    // planes nothing reads again are not stored — velocity and the encoded surface map (primary visibility), both diffuse
    // sample planes (resolving + reproject stages), the reprojected GI reservoirs of tracing frames, first-preview-pass results
    // that merely normalise their input, and the last a-trous
    // pass's colours when composition rides in that launch. st_camera_read_buffer of those planes returns what an earlier
    // frame or launch left there; ST_KEEP_ALL_PLANES=1 / st_debug_keep_all_planes(e, 1) stores everything the reference does.
    bool lean_frame = true;
    bool fuse_compose = true;  // ST_NO_FUSE_COMPOSE=1: frame composition as its own launch (the fast build's Image frames run it inside the last a-trous pass)
    bool skip_scratch_stores = true;  // ST_KEEP_SCRATCH=1: the fused DI spatial launch stores its intermediate records as the three separate passes would
    bool di_head_on_main = true;  // ST_DI_HEAD_ON_MAIN=0: DI sampling + temporal on the side stream (behind primary visibility) instead of the caller's
    bool alias_gi_history = true;  // ST_NO_GI_ALIAS=1: gi_resolving always copies the source reservoirs into the history plane
    bool fuse_wavelet = true;  // ST_NO_FUSE_WAVELET=1: strides 1 and 2 of the a-trous chain as two launches
    bool fuse_spatial = true;  // ST_NO_FUSE_SPATIAL=1: DI spatial resampling as three launches
    bool fuse_gi_sampling = true;  // ST_NO_FUSE_GI_SAMPLING=1: GI sampling passes a and b as two launches
    bool fuse_gi_valid = true;     // ST_NO_FUSE_GI_VALIDATION=1: gi_reprojection is a launch of its own on validation frames too
    bool fuse_di_head = true, fuse_gi_reproj = true;  // A/B switches for the two newest fusions (ST_NO_FUSE_DI_HEAD / ST_NO_FUSE_GI_REPROJECTION)
    bool fuse = true;       // run own-pixel consumer passes inside their producer's launch (ST_NO_FUSE=1: one launch per reference pass)
    // ... and for the SVGF passes (ST_TILE_MAP_DENOISE): mode 2 keeps the halo rows of the LDS windows and the a-trous taps
    // in one XCD's L2. Measured on one box: with mode 1 the two-stream frame is 1.347 instead of 1.373 ms, but a wavelet
    // launch moves 340 instead of 205 MB through the fabric (algorithmic: 174 MB) and takes 61 instead of 56 us on its own.
    uint32_t tile_map_denoise = 2;
    uint32_t tile_map = 1;  // blockIdx -> tile mapping (st_device.h); 1 measured best on MI355X with the current kernels (2 was, before the LDS-staged denoiser); ST_TILE_MAP overrides
    bool profiling = false;       // st_profile_enable bit 0: per-kernel event timing (serial execution)
    bool count_bytes = false;     // st_profile_enable bit 1: traversal-byte counters
    bool profile_kernel_events = false;  // st_profile_enable bit 3: every launch carries its own start / stop events (hipExtLaunchKernelGGL): no event packets between kernels
    bool profile_group_atrous = false;  // st_profile_enable bit 2: the a-trous chain's back-to-back launches share ONE event pair (an event between two kernels costs the second one 3-15 us)
    int side_priority = 0;     // ST_SIDE_PRIORITY: > 0 the side stream (primary visibility + GI chain) gets the device's highest stream priority, < 0 the lowest
    bool tick_timing = false;  // ST_TICK_TIMING=1: print the host-side cost of a scene refresh to stderr
    std::vector<ProfileRecord> profile_records; std::vector<hipEvent_t> event_pool;
    StKernelProfile profile_totals[KS_COUNT];

    Engine() {
        GpuLight sun{};
        sun.d0 = make_float4(0, 0, 0, 25.0f); sun.d1 = make_float4(0, 0, 0, INFINITY); sun.d2 = make_float4(b2f(1u), 0, 0, 0);
        light_buffer.push_back(sun);
        light_slot[-1] = 0;
        blue_noise.assign(256 * 256 * 4, 0);
        reset_profile_totals();
        if (const char* tm = getenv("ST_TILE_MAP")) tile_map = tile_map_denoise = (uint32_t)atoi(tm);
        if (const char* tm = getenv("ST_TILE_MAP_DENOISE")) tile_map_denoise = (uint32_t)atoi(tm);
        if (const char* nf = getenv("ST_NO_FUSE")) fuse = atoi(nf) == 0;
        if (const char* k = getenv("ST_NO_FUSE_DI_HEAD")) fuse_di_head = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_FUSE_SPATIAL")) fuse_spatial = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_FUSE_GI_SAMPLING")) fuse_gi_sampling = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_FUSE_GI_VALIDATION")) fuse_gi_valid = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_FUSE_GI_REPROJECTION")) fuse_gi_reproj = atoi(k) == 0;
        if (const char* no = getenv("ST_NO_OVERLAP")) overlap = atoi(no) == 0;
        if (const char* k = getenv("ST_SIDE_PRIORITY")) side_priority = atoi(k);
        if (const char* k = getenv("ST_NO_FUSE_WAVELET")) fuse_wavelet = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_GI_ALIAS")) alias_gi_history = atoi(k) == 0;
        if (const char* k = getenv("ST_DI_HEAD_ON_MAIN")) di_head_on_main = atoi(k) != 0;
        if (const char* k = getenv("ST_KEEP_SCRATCH")) skip_scratch_stores = atoi(k) == 0;
        if (const char* k = getenv("ST_KEEP_ALL_PLANES")) lean_frame = atoi(k) == 0;
#ifdef ST_WIDE_NODES
        if (const char* k = getenv("ST_WIDE_NODES")) wide_nodes = atoi(k) != 0;
#endif
        if (const char* k = getenv("ST_NO_FUSE_COMPOSE")) fuse_compose = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_PREVIEW_BOTH")) preview_both = atoi(k) == 0;
        if (const char* k = getenv("ST_NO_VARIANCE_IN_REPROJECT")) variance_in_reproject = atoi(k) == 0;
        if (const char* ns = getenv("ST_NO_STAGING")) staging.enabled = atoi(ns) == 0;
        if (const char* nd = getenv("ST_NO_DOUBLE_BUFFER")) double_buffer = atoi(nd) == 0;
        if (const char* tt = getenv("ST_TICK_TIMING")) tick_timing = atoi(tt) != 0;
        if (const char* ex = getenv("ST_EXACT")) if (atoi(ex) != 0) { arithmetic = ST_ARITH_EXACT; L = launchers_exact(); }
    }
    void reset_profile_totals() {
        for (int i = 0; i < KS_COUNT; i++) {
            memset(&profile_totals[i], 0, sizeof(StKernelProfile));
            snprintf(profile_totals[i].name, sizeof(profile_totals[i].name), "%s", kernel_info(i).name);
        }
    }
    ~Engine() {
        if (!has_device) return;
        (void)hipSetDevice(device);
        (void)hipDeviceSynchronize();
        for (auto& kv : cameras) release_camera(*kv.second);
        for (DeviceArray* d : {&d_byte_luts, &d_atlas, &d_blue_noise, &d_transmittance, &d_scattering, &d_sky}) d->release();
        for (LightSet& l : light_sets) { l.buf.release(); if (l.free_ev) (void)hipEventDestroy(l.free_ev); }
        for (SceneSet& t : sets) {
            for (DeviceArray* d : {&t.bvh, &t.tri_attr, &t.xforms, &t.materials, &t.base_packed, &t.tri_geo, &t.tri_bounds, &t.entry_of_tri, &t.parent, &t.refit_local, &t.refit_items, &t.refit_batch_off}) d->release();
            if (t.free_ev) (void)hipEventDestroy(t.free_ev);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (ev_copy) (void)hipEventDestroy(ev_copy);
        for (auto& r : profile_records) { if (r.owns_start) (void)hipEventDestroy(r.start); (void)hipEventDestroy(r.stop); }
        for (auto e : event_pool) (void)hipEventDestroy(e);
        if (ev_tick) (void)hipEventDestroy(ev_tick);
        staging.release();
    }
    static void release_camera(CameraState& c) {
        if (c.slab) (void)hipFree(c.slab);
        if (c.counters) (void)hipFree(c.counters);
        if (c.tile_mask) (void)hipFree(c.tile_mask);
        c.tile_mask = nullptr;
        c.slab = nullptr; c.counters = nullptr;
        if (c.side_stream) (void)hipStreamDestroy(c.side_stream);
        for (hipEvent_t* e : {&c.ev_di_head, &c.ev_gi_done, &c.ev_prim_ok, &c.ev_frame_done, &c.ev_setup}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
        c.side_stream = nullptr; c.have_prev_frame_events = false;
        if (c.present_stream) { (void)hipStreamSynchronize(c.present_stream); (void)hipStreamDestroy(c.present_stream); c.present_stream = nullptr; }
        for (auto& p : c.present) { for (hipEvent_t* e : {&p.ev_src, &p.ev_done}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; } p = CameraState::PresentSlot(); }
    }
    // st_camera_present_copy: `src_device` (what st_render_camera composed into on `stream`) -> `dst_host`, asynchronously
    int present_copy(CameraState& c, const void* src, void* dst, size_t bytes, hipStream_t stream) {
        if (!has_device) return fail(ST_ERR_NO_DEVICE, "present copy on a host-only engine");
        ST_HIP(hipSetDevice(device));
        if (!c.present_stream) ST_HIP(hipStreamCreateWithFlags(&c.present_stream, hipStreamNonBlocking));
        // the slot that already serves this destination, else the older one
        CameraState::PresentSlot* slot = nullptr;
        for (auto& p : c.present) if (p.dst == dst) slot = &p;
        if (!slot) { slot = &c.present[c.present_next & 1u]; c.present_next++; }
        if (slot->pending) ST_HIP(hipEventSynchronize(slot->ev_done));  // only when the caller runs more than two frames ahead
        if (!slot->ev_src) { ST_HIP(hipEventCreateWithFlags(&slot->ev_src, hipEventDisableTiming)); ST_HIP(hipEventCreateWithFlags(&slot->ev_done, hipEventDisableTiming)); }
        slot->src = src; slot->dst = dst;
        ST_HIP(hipEventRecord(slot->ev_src, stream));                    // the frame is composed
        ST_HIP(hipStreamWaitEvent(c.present_stream, slot->ev_src, 0));
        ST_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.present_stream));
        ST_HIP(hipEventRecord(slot->ev_done, c.present_stream));
        slot->pending = true;
        return ST_OK;
    }
    // 1 = the copy into `dst` has landed (or none was asked for), 0 = still in flight; wait != 0 blocks until it has
    int present_ready(CameraState& c, const void* dst, int wait, int* ready) {
        *ready = 1;
        for (auto& p : c.present) {
            if (p.dst != dst || !p.pending) continue;
            if (wait) { ST_HIP(hipEventSynchronize(p.ev_done)); p.pending = false; }
            else {
                const hipError_t q = hipEventQuery(p.ev_done);
                if (q == hipSuccess) p.pending = false;
                else if (q == hipErrorNotReady) { (void)hipGetLastError(); *ready = 0; }
                else return fail(ST_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(q));
            }
        }
        return ST_OK;
    }

    // ---- materials (materials.rs:33-96, material.rs:29-50)
    float4 image_rect(uint64_t h) const {
        if (!h) return make_float4(0, 0, 0, 0);
        auto it = images.find(h);
        if (it == images.end() || atlas_w == 0) return make_float4(0, 0, 0, 0);
        const ImageRec& r = it->second;
        return make_float4((float)r.x / (float)atlas_w, (float)r.y / (float)atlas_h, (float)r.w / (float)atlas_w, (float)r.h / (float)atlas_h);
    }
    void rebuild_gpu_materials() {
        gpu_materials.resize(materials.size()); material_base_packed.resize(materials.size());
        for (size_t i = 0; i < materials.size(); i++) {
            const StMaterial& m = materials[i]; GpuMaterial& g = gpu_materials[i];
            g.base_color = make_float4(m.base_color[0], m.base_color[1], m.base_color[2], m.base_color[3]);
            g.base_color_texture = image_rect(m.base_color_texture);
            g.emissive = make_float4(m.emissive[0], m.emissive[1], m.emissive[2], m.emissive[3]);
            g.emissive_texture = image_rect(m.emissive_texture);
            g.roughness = pow2_(m.perceptual_roughness);
            g.metallic = m.metallic; g.reflectance = m.reflectance; g.ior = m.ior;
            g.metallic_roughness_texture = image_rect(m.metallic_roughness_texture);
            g.normal_map_texture = image_rect(m.normal_map_texture);
            material_base_packed[i] = gbuffer_pack_base_color(g.base_color);  // st_math.h routines are bit-identical on host and device
        }
    }

    // ---- lights (lights.rs:49-172, light.rs:25-79)
    static void note(std::vector<int64_t>& v, int64_t k) { if (std::find(v.begin(), v.end(), k) == v.end()) v.push_back(k); }
    void overwrite_light(uint32_t slot, int64_t key, GpuLight g) {
        const GpuLight old = light_buffer[slot];
        g.prev_d0 = old.d0; g.prev_d1 = old.d1; g.prev_d2 = old.d2;
        note(lights_updated, key);
        light_buffer[slot] = g;
    }
    void insert_light(uint64_t id, const StLight& l) {
        GpuLight g{};
        g.d0 = make_float4(l.position[0], l.position[1], l.position[2], l.radius);
        g.d1 = make_float4(l.color[0], l.color[1], l.color[2], l.range);
        if (l.kind == ST_LIGHT_POINT) g.d2 = make_float4(b2f(1u), 0, 0, 0);
        else {
            V3 n = v3(l.direction[0], l.direction[1], l.direction[2]);  // Normal::encode (normal.rs:9-24)
            n = n / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
            V2 e = n.z >= 0.0f ? v2(n.x, n.y) : v2(copysignf(1.0f - fabsf(n.y), n.x), copysignf(1.0f - fabsf(n.x), n.y));
            e = e * 0.5f + 0.5f;
            g.d2 = make_float4(b2f(2u), e.x, e.y, l.angle);
        }
        const int64_t key = (int64_t)id;
        auto it = light_slot.find(key);
        if (it != light_slot.end()) { overwrite_light(it->second, key, g); return; }
        if (next_light_id < light_buffer.size()) { light_buffer[next_light_id] = g; light_slot[key] = next_light_id; }
        else { light_slot[key] = (uint32_t)light_buffer.size(); light_buffer.push_back(g); }
        note(lights_created, key);
        next_light_id += 1;
    }
    void remove_light(uint64_t id) {
        const int64_t key = (int64_t)id;
        auto it = light_slot.find(key);
        if (it == light_slot.end()) return;  // silent no-op like the reference
        const uint32_t slot = it->second;
        light_slot.erase(it);
        light_buffer.erase(light_buffer.begin() + slot);
        light_buffer.push_back(GpuLight{});
        lights_created.erase(std::remove(lights_created.begin(), lights_created.end(), key), lights_created.end());
        lights_updated.erase(std::remove(lights_updated.begin(), lights_updated.end(), key), lights_updated.end());
        lights_remapped.erase(key);
        if (std::find(lights_killed.begin(), lights_killed.end(), slot) == lights_killed.end()) lights_killed.push_back(slot);
        next_light_id -= 1;
        for (auto& kv : light_slot)
            if (kv.second > slot) { if (!lights_remapped.count(kv.first)) lights_remapped[kv.first] = kv.second; kv.second -= 1; }
    }
    void snapshot_lights() {  // lights.rs:128-154: what the device sees this frame, then commit prev_* for the next one
        for (uint32_t s : lights_killed) light_buffer[s].d3.x = b2f(0xcafebabeu);
        for (auto& kv : lights_remapped) light_buffer[kv.second].d3.x = b2f(light_slot[kv.first] + 1u);
        gpu_lights = light_buffer;
        auto commit = [&](int64_t k) { GpuLight& l = light_buffer[light_slot[k]]; l.prev_d0 = l.d0; l.prev_d1 = l.d1; l.prev_d2 = l.d2; };
        for (int64_t k : lights_created) commit(k);
        for (int64_t k : lights_updated) commit(k);
        for (uint32_t s : lights_killed) light_buffer[s].d3.x = b2f(0u);
        for (auto& kv : lights_remapped) light_buffer[kv.second].d3.x = b2f(0u);
        lights_created.clear(); lights_updated.clear(); lights_remapped.clear(); lights_killed.clear();
    }

    // ---- instances -> world-space triangles (instances.rs:69-139, mesh_triangle.rs:47-86, triangle.rs:16-37)
    void drop_instance_triangles(uint64_t id) {
        auto it = instance_triangles.find(id);
        if (it == instance_triangles.end()) return;
        triangle_free.give(it->second.first, it->second.second);
        for (size_t i = it->second.first; i < it->second.second; i++) prim_alive[i] = 0;
        instance_triangles.erase(it);
    }
    void bake(const StMeshTriangle& t, const InstanceRec& inst, uint32_t material, size_t slot) {
        // normals use transpose(inverse(xform)) (Mat4::transform_vector3 order); tangents follow the forward matrix
        const Affine& inv = inst.xform_inv;
        const V3 r0 = v3(inv.x.x, inv.y.x, inv.z.x), r1 = v3(inv.x.y, inv.y.y, inv.z.y), r2 = v3(inv.x.z, inv.y.z, inv.z.z);
        const float det = dot(inst.xform.z, cross(inst.xform.x, inst.xform.y));
        const float sign = (f2b(det) >> 31) ? -1.0f : 1.0f;
        V3 p[3], n[3]; float4 tg[3];
        for (int i = 0; i < 3; i++) {
            p[i] = affine_point(inst.xform, v3(t.positions[i][0], t.positions[i][1], t.positions[i][2]));
            const V3 nn = v3(t.normals[i][0], t.normals[i][1], t.normals[i][2]);
            // transpose(inverse): columns are the inverse's rows; the 4th row of the transposed matrix carries the
            // inverse translation in .w only, which transform_vector3 drops
            V3 acc = r0 * nn.x; acc = r1 * nn.y + acc; acc = r2 * nn.z + acc;
            n[i] = normalize(acc);
            const V3 tt = normalize(affine_vec(inst.xform, v3(t.tangents[i][0], t.tangents[i][1], t.tangents[i][2])));
            tg[i] = make_float4(tt.x, tt.y, tt.z, t.tangents[i][3] * sign);
        }
        HostTriangle h;
        h.d0 = f4(p[0], t.uvs[0][0]); h.d1 = f4(n[0], t.uvs[0][1]); h.d2 = tg[0];
        h.d3 = f4(p[1], t.uvs[1][0]); h.d4 = f4(n[1], t.uvs[1][1]); h.d5 = tg[1];
        h.d6 = f4(p[2], t.uvs[2][0]); h.d7 = f4(n[2], t.uvs[2][1]); h.d8 = tg[2];
        triangles[slot] = h;
        BuildPrim bp;
        bp.triangle_id = (uint32_t)slot; bp.material_id = material;
        bp.center = (((v3s(0.0f) + p[0]) + p[1]) + p[2]) / 3.0f;
        bp.bounds = Aabb(); bp.bounds.grow(p[0]); bp.bounds.grow(p[1]); bp.bounds.grow(p[2]);
        prims[slot] = bp; prim_alive[slot] = 1;
        tri_geo[3 * slot] = f4(p[0], 0.0f); tri_geo[3 * slot + 1] = f4(p[1] - p[0], 0.0f); tri_geo[3 * slot + 2] = f4(p[2] - p[0], 0.0f);
        tri_bounds[2 * slot] = f4(bp.bounds.lo, 0.0f); tri_bounds[2 * slot + 1] = f4(bp.bounds.hi, 0.0f);
        tri_attr[4 * slot] = f4(n[0], t.uvs[0][0]); tri_attr[4 * slot + 1] = f4(n[1], t.uvs[0][1]); tri_attr[4 * slot + 2] = f4(n[2], t.uvs[1][0]);
        tri_attr[4 * slot + 3] = make_float4(t.uvs[1][1], t.uvs[2][0], t.uvs[2][1], b2f(inst.xslot));
    }
    struct BakeJob { const std::vector<StMeshTriangle>* mesh; const InstanceRec* inst; uint32_t material; size_t first, count; };
    bool refresh_instances() {
        if (!instances_dirty) return false;
        instances_dirty = false;
        std::vector<BakeJob> jobs; size_t total = 0;
        {   // one reallocation at most for everything this refresh appends (a scene load appends every instance)
            size_t fresh = 0;
            for (const auto& inst : instances) {
                if (!inst.dirty || instance_triangles.count(inst.id)) continue;
                auto mesh = meshes.find(inst.mesh);
                if (mesh != meshes.end()) fresh += mesh->second.size();
            }
            if (fresh) {
                const size_t want = triangles.size() + fresh;
                triangles.reserve(want); prims.reserve(want); prim_alive.reserve(want); tri_geo.reserve(3 * want); tri_attr.reserve(4 * want);
            }
        }
        for (auto& inst : instances) {
            if (!inst.dirty) continue;
            inst.dirty = false;
            auto mesh = meshes.find(inst.mesh);
            auto mat = material_slot.find(inst.material);
            if (mesh == meshes.end() || mat == material_slot.end()) { inst.dirty = true; instances_dirty = true; continue; }  // retry next tick
            const size_t count = mesh->second.size();
            auto have = instance_triangles.find(inst.id);
            if (have != instance_triangles.end() && have->second.second - have->second.first != count) { drop_instance_triangles(inst.id); have = instance_triangles.end(); }
            size_t b, e;
            if (have != instance_triangles.end()) { b = have->second.first; e = have->second.second; }
            else if (!triangle_free.take(count, &b, &e)) {
                b = triangles.size(); e = b + count;
                triangles.resize(e); prims.resize(e); prim_alive.resize(e, 0); tri_geo.resize(3 * e); tri_attr.resize(4 * e); tri_bounds.resize(2 * e);
                for (SceneSet& t : sets) t.tri_full = true;
            }
            jobs.push_back({&mesh->second, &inst, mat->second, b, count});
            total += count;
            for (SceneSet& t : sets) { t.dirty_lo = std::min(t.dirty_lo, b); t.dirty_hi = std::max(t.dirty_hi, e); }  // slots each device copy still has to receive
            instance_triangles[inst.id] = {b, e};
        }
        // Baking (instances.rs:100-139) writes disjoint slots and reads nothing it writes, so once every range is assigned —
        // the arrays do not move any more — large refreshes are spread over the BVH builder's worker pool in chunks.
        const auto tb0 = std::chrono::steady_clock::now();
        constexpr size_t kChunk = 2048, kParallelFrom = 16384;
        unsigned threads = std::thread::hardware_concurrency();
        if (threads > 16u) threads = 16u;
        if (total < kParallelFrom || threads < 2u) {
            for (const BakeJob& j : jobs)
                for (size_t i = 0; i < j.count; i++) bake((*j.mesh)[i], *j.inst, j.material, j.first + i);
        } else {
            TaskPool pool(threads);
            for (const BakeJob& j : jobs)
                for (size_t at = 0; at < j.count; at += kChunk) {
                    const size_t end = std::min(j.count, at + kChunk);
                    pool.push([this, j, at, end] { for (size_t i = at; i < end; i++) bake((*j.mesh)[i], *j.inst, j.material, j.first + i); });
                }
            pool.finish();
        }
        if (tick_timing) fprintf(stderr, "[bake] %zu triangles in %zu jobs: %.2f ms\n", total, jobs.size(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb0).count());
        return true;
    }

    // ---- BVH refit (the alternative to a rebuild when instances only move)
    // What the leaves of the current tree refer to: every live (triangle slot, material) pair, plus the Blend flags baked into
    // the leaf entries. Equal signatures mean the stream's topology and leaf entries are still right; only boxes moved.
    uint64_t topology_of(const std::vector<uint8_t>& blend) const {
        uint64_t h = 0x9e3779b97f4a7c15ull;
        auto mix = [&h](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; h ^= h >> 29; };
        for (size_t i = 0; i < prims.size(); i++) if (prim_alive[i]) mix(((uint64_t)i << 32) | prims[i].material_id);
        mix(0xffffffffffffffffull);
        for (uint8_t b : blend) mix(b);
        return h;
    }
    // offsets of the internal nodes of bvh_stream, in stream order (serializer.rs:20-110: a node is internal when d0.w == 0)
    void index_stream() {
        internal_positions.clear();
        for (size_t p = 0; p < bvh_stream.size();) {
            if (f2b(bvh_stream[p].w) == 0u) { internal_positions.push_back((uint32_t)p); p += 4; }
            else p += 1;
        }
    }
    // box of the subtree that starts at stream offset p: a run of leaf entries (triangle bounds as baked) or an internal node
    // (union of the two child boxes it stores)
    Aabb subtree_box(size_t p) const {
        Aabb box;
        if (f2b(bvh_stream[p].w) == 0u) {
            box.grow(v3(bvh_stream[p].x, bvh_stream[p].y, bvh_stream[p].z)); box.grow(v3(bvh_stream[p + 1].x, bvh_stream[p + 1].y, bvh_stream[p + 1].z));
            box.grow(v3(bvh_stream[p + 2].x, bvh_stream[p + 2].y, bvh_stream[p + 2].z)); box.grow(v3(bvh_stream[p + 3].x, bvh_stream[p + 3].y, bvh_stream[p + 3].z));
            return box;
        }
        for (;; p++) {
            const float4* b = &tri_bounds[2u * (size_t)f2b(bvh_stream[p].y)];  // = prims[...].bounds, 32 B apart instead of 56
            box.grow(v3(b[0].x, b[0].y, b[0].z)); box.grow(v3(b[1].x, b[1].y, b[1].z));
            if (!(f2b(bvh_stream[p].x) & 1u)) return box;
        }
    }
    void refit_node(size_t p) {
        const Aabb l = subtree_box(p + 4), r = subtree_box(f2b(bvh_stream[p + 1].w));
        bvh_stream[p] = make_float4(l.lo.x, l.lo.y, l.lo.z, bvh_stream[p].w);
        bvh_stream[p + 1] = make_float4(l.hi.x, l.hi.y, l.hi.z, bvh_stream[p + 1].w);
        bvh_stream[p + 2] = make_float4(r.lo.x, r.lo.y, r.lo.z, bvh_stream[p + 2].w);
        bvh_stream[p + 3] = make_float4(r.hi.x, r.hi.y, r.hi.z, bvh_stream[p + 3].w);
    }
    // internal nodes whose offsets lie in [begin, end), last to first: children sit behind their parent in the stream, so a
    // backward sweep sees finished children
    void refit_span(size_t begin, size_t end) {
        const auto lo = std::lower_bound(internal_positions.begin(), internal_positions.end(), (uint32_t)begin);
        auto hi = std::lower_bound(internal_positions.begin(), internal_positions.end(), (uint32_t)end);
        while (hi != lo) refit_node(*--hi);
    }
    // One thread: at 134 k triangles the sweep is about a millisecond, less than starting a worker pool for it would buy back.
    void refit_stream() { refit_span(0, bvh_stream.size()); }
    bool device_refit_possible() const { return bvh_refresh_mode == ST_BVH_REFIT_DEVICE && has_device && !wide_nodes; }

    // ---- tick (lib.rs:301-395)
    int tick(hipStream_t stream) {
        bool scene_changed = false;
        if (materials_dirty || atlas_dirty) { materials_dirty = false; rebuild_gpu_materials(); scene_changed = true; }
        const bool timing = tick_timing;
        auto now = [] { return std::chrono::steady_clock::now(); };
        auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
        const auto t0 = now();
        if (refresh_instances()) {
            for (const auto& inst : instances) {
                float4* x = instance_xforms.data() + 8u * inst.xslot;
                const Affine* src[2] = {&inst.xform_inv, &inst.prev_xform};
                for (int k = 0; k < 2; k++) { x[4 * k] = f4(src[k]->x, 0.0f); x[4 * k + 1] = f4(src[k]->y, 0.0f); x[4 * k + 2] = f4(src[k]->z, 0.0f); x[4 * k + 3] = f4(src[k]->t, 0.0f); }
            }
            const auto t1 = now();
            std::vector<uint8_t> blend(materials.size());
            for (size_t i = 0; i < materials.size(); i++) blend[i] = materials[i].alpha_mode == 1u;
            const bool refitting = bvh_refresh_mode != ST_BVH_REBUILD;
            const uint64_t signature = refitting ? topology_of(blend) : 0;
            if (refitting && have_topology && signature == topology_signature) {
                // ST_BVH_REFIT_DEVICE: the boxes are recomputed on the device from the moved triangles' bounds (k_bvh.hip); the host's
                // copy of the stream is brought up to date only when something reads it
                if (device_refit_possible()) host_stream_stale = true; else refit_stream();
                refits++;
                if (timing) fprintf(stderr, "[st_tick] bake %.2f ms, refit %.2f ms (%zu internal nodes)\n", ms(t0, t1), ms(t1, now()), internal_positions.size());
            } else {
                bvh.begin_refresh();  // keeps the previous tree: unchanged subtrees are copied, not rebuilt (same result as a fresh build)
                for (size_t i = 0; i < prims.size(); i++) if (prim_alive[i]) bvh.prims.push_back(prims[i]);
                const auto t2 = now();
                bvh.run();
                const auto t3 = now();
                bvh.flatten(blend, bvh_stream);
                const auto t4 = now();
                rebuilds++; tree_version++; host_stream_stale = false;
                mark_internal_starts(); measure_stack_need();
                have_topology = false;
                if (refitting) { index_stream(); topology_signature = signature; have_topology = true; }
                if (timing) fprintf(stderr, "[st_tick] bake %.2f ms, gather %.2f ms, bvh build %.2f ms, flatten %.2f ms (%zu triangles, %zu reused)\n", ms(t0, t1), ms(t1, t2), ms(t2, t3), ms(t3, t4), bvh.prims.size(), bvh.reused_primitives());
            }
            scene_changed = true;
        }
        light_count = next_light_id;
        {   // World::sun_dir (world.rs:18-24)
            float sa, ca, sz, cz;
            sincos_(sun_altitude, &sa, &ca); sincos_(sun_azimuth, &sz, &cz);
            sun_dir_ = v3(ca * sz, sa, -ca * cz);
        }
        if (sun_dirty) {
            sun_dirty = false;
            V3 color = sun_transmittance(v3(0.0f, 6.360f + 0.0002f, 0.0f), sun_dir_);
            color = color * 20.0f * 5.0f;
            GpuLight sun{};
            const V3 pos = sun_dir_ * 1000.0f;
            sun.d0 = f4(pos, 25.0f); sun.d1 = f4(color, INFINITY); sun.d2 = make_float4(b2f(1u), 0, 0, 0);
            overwrite_light(0, -1, sun);
        }
        snapshot_lights();
        if (has_device) {
            ST_HIP(hipSetDevice(device));
            // Uploads of an earlier tick that no render has waited for yet stay pending until their event has completed: a
            // tick that uploads nothing must not make a later render on another stream forget them.
            if (tick_work_in_flight && hipEventQuery(ev_tick) == hipSuccess) tick_work_in_flight = false;
            if (copy_in_flight && hipEventQuery(ev_copy) == hipSuccess) copy_in_flight = false;
            (void)hipGetLastError();  // hipErrorNotReady from the queries is not an error
            bool copied_now = false;  // this tick queued copies on copy_stream
            bool pageable = false;  // some copy of this tick reads pageable host memory (or writes it): join the stream before returning
            staging.begin_tick();
            bool pageable_copy = false;
            if (scene_changed || !scene_uploaded) {
                int rc;
                // which copy, on which stream: the first upload and ST_NO_DOUBLE_BUFFER=1 write the live copy in place on the
                // caller's stream (behind the frames queued there); every later change goes to the other copy on copy_stream
                int target = live; hipStream_t up = stream; bool* flag = &pageable; bool other_copy = false;
                if (double_buffer && scene_uploaded && !mixed_render_streams) {
                    if (!copy_stream) { ST_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); ST_HIP(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming)); }
                    if (!alternating) {  // frames enqueued so far read the live copy without marking their end: mark it now, behind them
                        alternating = true;
                        SceneSet& l = sets[live];
                        if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
                        ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
                    }
                    target = live ^ 1; up = copy_stream; flag = &pageable_copy; other_copy = true;
                    if (sets[target].busy) { ST_HIP(hipStreamWaitEvent(copy_stream, sets[target].free_ev, 0)); sets[target].busy = false; }
                } else if (mixed_render_streams) ST_HIP(hipDeviceSynchronize());  // cameras render on several streams: no single event ends their reads
                SceneSet& t = sets[target];
                if (device_refit_possible() && t.valid && !t.tri_full && t.tree_version == tree_version && t.tri_geo.capacity >= tri_geo.size() * sizeof(float4)) {
                    // This copy holds the current tree; only boxes and moved triangles are behind. Send the records and bounds of the
                    // triangle slots baked since it was written and let the device patch its leaf entries and refit its boxes.
                    if (t.dirty_lo < t.dirty_hi) {
                        if ((rc = t.tri_geo.upload_range(tri_geo.data(), 3 * t.dirty_lo * sizeof(float4), 3 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
                        if ((rc = t.tri_bounds.upload_range(tri_bounds.data(), 2 * t.dirty_lo * sizeof(float4), 2 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
                        L.launch_bvh_patch_leaves(static_cast<float4*>(t.bvh.ptr), static_cast<const float4*>(t.tri_geo.ptr), static_cast<const uint32_t*>(t.entry_of_tri.ptr), (uint32_t)t.dirty_lo, (uint32_t)t.dirty_hi, up);
                    }
                    for (const auto& level : refit_levels_)   // this copy holds the current tree, so the engine's work list is its own
                        L.launch_bvh_refit(static_cast<float4*>(t.bvh.ptr), static_cast<const float4*>(t.tri_bounds.ptr), static_cast<const uint32_t*>(t.parent.ptr), static_cast<const uint32_t*>(t.refit_local.ptr),
                                           static_cast<const uint32_t*>(t.refit_items.ptr), static_cast<const uint32_t*>(t.refit_batch_off.ptr), level.first, level.second, up);
                    device_refits++;
                } else {
                    if (host_stream_stale) { refit_stream(); host_stream_stale = false; }
                    expand_stream();
                    append_wide_nodes();
                    // traversal pointers are 32-bit BYTE offsets into the device stream (64 B per entry) and stack slots hold entry numbers
                    if ((size_t)device_bvh_len * sizeof(float4) > 0xffffffffull) return fail(ST_ERR_INVALID_ARGUMENT, "the BVH stream exceeds 4 GiB (2^26 entries): traversal pointers are 32-bit byte offsets");
                    if ((rc = t.bvh.upload(bvh_upload_.data(), bvh_upload_.size() * sizeof(float4), up, staging, flag))) return rc;
                    if (device_refit_possible()) {  // what the device refit of later ticks needs beside the stream
                        index_device_tree();
                        if ((rc = t.tri_geo.upload(tri_geo.data(), tri_geo.size() * sizeof(float4), up, staging, flag))) return rc;
                        if ((rc = t.tri_bounds.upload(tri_bounds.data(), tri_bounds.size() * sizeof(float4), up, staging, flag))) return rc;
                        if ((rc = t.entry_of_tri.upload(entry_of_tri_.data(), entry_of_tri_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                        if ((rc = t.parent.upload(parent_.data(), parent_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                        if ((rc = t.refit_local.upload(refit_local_.data(), refit_local_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                        if (!refit_items_.empty() && (rc = t.refit_items.upload(refit_items_.data(), refit_items_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                        if ((rc = t.refit_batch_off.upload(refit_batch_off_.data(), refit_batch_off_.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                        t.tree_version = tree_version;
                    }
                }
                // attribute records: whole the first time or after they grew, otherwise only the slots baked since this copy was written
                const bool partial = t.valid && !t.tri_full && t.tri_attr.capacity >= tri_attr.size() * sizeof(float4);
                if (!partial) {
                    if ((rc = t.tri_attr.upload(tri_attr.data(), tri_attr.size() * sizeof(float4), up, staging, flag))) return rc;
                } else if (t.dirty_lo < t.dirty_hi) {
                    if ((rc = t.tri_attr.upload_range(tri_attr.data(), 4 * t.dirty_lo * sizeof(float4), 4 * (t.dirty_hi - t.dirty_lo) * sizeof(float4), up, staging, flag))) return rc;
                }
                t.dirty_lo = SIZE_MAX; t.dirty_hi = 0; t.tri_full = false; t.valid = true;
                if ((rc = t.xforms.upload(instance_xforms.data(), instance_xforms.size() * sizeof(float4), up, staging, flag))) return rc;
                if ((rc = t.materials.upload(gpu_materials.data(), gpu_materials.size() * sizeof(GpuMaterial), up, staging, flag))) return rc;
                if ((rc = t.base_packed.upload(material_base_packed.data(), material_base_packed.size() * sizeof(uint32_t), up, staging, flag))) return rc;
                if (other_copy) copied_now = true;
                live = target; live_bvh_texels = device_bvh_len;
                scene_uploaded = true;
                scene_changed = !other_copy;  // in-place uploads count as work on the caller's stream below
            }
            bool misc_uploaded = atlas_dirty || blue_noise_dirty, uploaded_device_images = false;
            if (atlas_dirty) { int rc = d_atlas.upload(atlas.data(), atlas.size(), stream, staging, &pageable); if (rc) return rc; }
            for (auto& kv : device_images) {
                DeviceImage& di = kv.second;
                if (!di.pending && !di.dynamic) continue;
                const ImageRec& r = images.at(kv.first);
                uint8_t* dst = static_cast<uint8_t*>(d_atlas.ptr) + ((size_t)r.y * atlas_w + r.x) * 4;
                ST_HIP(hipMemcpy2DAsync(dst, (size_t)atlas_w * 4, di.pixels, di.pitch, (size_t)r.w * 4, r.h, hipMemcpyDeviceToDevice, stream));
                if (!di.dynamic) {  // keep the host copy complete: it is what a later full upload sends
                    ST_HIP(hipMemcpy2DAsync(&atlas[((size_t)r.y * atlas_w + r.x) * 4], (size_t)atlas_w * 4, di.pixels, di.pitch, (size_t)r.w * 4, r.h, hipMemcpyDeviceToHost, stream));
                    misc_uploaded = true; pageable = true;  // joins the stream below before the host copy is read again
                }
                di.pending = false;
                uploaded_device_images = true;
            }
            if (blue_noise_dirty) { int rc = d_blue_noise.upload(blue_noise.data(), blue_noise.size(), stream, staging, &pageable); if (rc) return rc; blue_noise_dirty = false; }
            bool uploaded = scene_changed || misc_uploaded || uploaded_device_images;
            // lights change rarely; skipping the identical re-upload also skips the stream sync below, so the host can
            // run a frame ahead of the GPU (the reference re-uploads only dirty buffers too: mapped_storage_buffer.rs:103-121)
            if (gpu_lights.size() != uploaded_lights.size() || memcmp(gpu_lights.data(), uploaded_lights.data(), gpu_lights.size() * sizeof(GpuLight)) != 0) {
                int target = live_lights; hipStream_t up = stream; bool* flag = &pageable; bool other_copy = false;
                if (double_buffer && lights_uploaded && !mixed_render_streams) {
                    if (!copy_stream) { ST_HIP(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking)); ST_HIP(hipEventCreateWithFlags(&ev_copy, hipEventDisableTiming)); }
                    if (!lights_alternating) {  // as for the scene: the frames queued so far end here
                        lights_alternating = true;
                        LightSet& l = light_sets[live_lights];
                        if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
                        ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
                    }
                    target = live_lights ^ 1; up = copy_stream; flag = &pageable_copy; other_copy = true;
                    if (light_sets[target].busy) { ST_HIP(hipStreamWaitEvent(copy_stream, light_sets[target].free_ev, 0)); light_sets[target].busy = false; }
                } else if (mixed_render_streams) ST_HIP(hipDeviceSynchronize());
                int rc = light_sets[target].buf.upload(gpu_lights.data(), gpu_lights.size() * sizeof(GpuLight), up, staging, flag);
                if (rc) return rc;
                live_lights = target; lights_uploaded = true;
                uploaded_lights = gpu_lights;
                if (other_copy) copied_now = true; else uploaded = true;
            }
            if (copied_now) {
                copy_in_flight = true;
                ST_HIP(hipEventRecord(ev_copy, copy_stream));
                ST_HIP(hipStreamWaitEvent(stream, ev_copy, 0));  // the caller's stream: the next frame's kernels (and the staging slot's event) come after the copies
            }
            if (int rc = staging.end_tick(stream)) return rc;
            // What was uploaded went through page-locked staging, so the caller may change the scene again at once; the next
            // frame's side stream is ordered behind these copies by an event (render). Only copies that touch pageable
            // host memory directly (staging full or disabled) make the tick wait for the stream.
            if (uploaded) {
                if (!ev_tick) ST_HIP(hipEventCreateWithFlags(&ev_tick, hipEventDisableTiming));
                ST_HIP(hipEventRecord(ev_tick, stream));
                tick_work_in_flight = true;
            }
            if (pageable_copy) ST_HIP(hipStreamSynchronize(copy_stream));
            if ((uploaded && pageable) || sync_every_tick) ST_HIP(hipStreamSynchronize(stream));
        }
        atlas_dirty = false;
        for (auto& kv : cameras) kv.second->frame = frame;  // CameraController::flush
        frame += 1;
        return ST_OK;
    }

    // ---- cameras (camera.rs:50-66, camera_controller.rs:27-86)
    static GpuCamera serialize_camera(const StCamera& c) {
        const M4 transform = m4_from_cols(c.transform), projection = m4_from_cols(c.projection);
        GpuCamera g;
        g.projection_view = m4_mul(projection, m4_inverse(transform));
        g.ndc_to_world = m4_mul(transform, m4_inverse(projection));
        g.origin = make_float4(transform.c[3].x, transform.c[3].y, transform.c[3].z, 0.0f);
        g.screen = make_float4((float)c.width, (float)c.height, 0.0f, 0.0f);
        return g;
    }
    int allocate_camera(CameraState& c) {
        c.row0 = 0; c.row1 = c.desc.height;
        if (!has_device) return ST_OK;
        ST_HIP(hipSetDevice(device));
        release_camera(c);
        const size_t n = (size_t)c.desc.width * c.desc.height;
        size_t total = 0;
        for (int i = 0; i < ST_BUF_COUNT + kInternalPlanes; i++) {
            c.plane_bytes[i] = i == ST_BUF_DBG_USED_MEMORY ? n * 4 : n * 16 * plane_texels_per_pixel(i);
            total += (c.plane_bytes[i] + 255) & ~size_t(255);
        }
        ST_HIP(hipMalloc(&c.slab, total));
        ST_HIP(hipMemset(c.slab, 0, total));  // wgpu zero-initialises resources; stale-data paths depend on it
        c.slab_bytes = total;
        size_t off = 0;
        for (int i = 0; i < ST_BUF_COUNT + kInternalPlanes; i++) { c.plane[i] = reinterpret_cast<float4*>(static_cast<char*>(c.slab) + off); off += (c.plane_bytes[i] + 255) & ~size_t(255); }
        c.gi_aliased = false;
        if (hipMalloc(reinterpret_cast<void**>(&c.counters), kCounterBytes) != hipSuccess) {
            (void)hipGetLastError(); c.counters = nullptr;
            release_camera(c);  // do not leak the slab
            return fail(ST_ERR_HIP, "hipMalloc(camera counters) failed");
        }
        ST_HIP(hipMemset(c.counters, 0, kCounterBytes));
        {
            const size_t tiles = (size_t)((c.desc.width + 7u) / 8u) * ((c.desc.height + 7u) / 8u);
            if (hipMalloc(reinterpret_cast<void**>(&c.tile_mask), 2 * tiles * sizeof(unsigned long long)) != hipSuccess) { (void)hipGetLastError(); c.tile_mask = nullptr; release_camera(c); return fail(ST_ERR_HIP, "hipMalloc(camera tile mask) failed"); }
            ST_HIP(hipMemset(c.tile_mask, 0, 2 * tiles * sizeof(unsigned long long)));  // [0, tiles): variance's, [tiles, 2 tiles): the GI preview's
            c.tile_mask_tiles = tiles;
        }
        memset(c.profiled_traversal_bytes, 0, sizeof(c.profiled_traversal_bytes));
        ST_HIP(hipDeviceSynchronize());  // the clears run on the null stream; renders may use any stream
        return ST_OK;
    }

    // ---- profiling
    hipEvent_t take_event() {
        if (!event_pool.empty()) { hipEvent_t e = event_pool.back(); event_pool.pop_back(); return e; }
        hipEvent_t e; (void)hipEventCreate(&e); return e;
    }
    // One event pair per RUN of consecutive launches of the same slot on the same stream (the five a-trous launches, say):
    // an event between two kernels makes the second wait for a barrier packet, which adds microseconds to every launch
    // it brackets, so back-to-back launches of one slot are timed as one interval and divided by their count.
    // Consecutive runs on one stream share the event between them (the stop of one is the start of the next).
    struct OpenScope { int slot = -1; hipStream_t stream = nullptr; hipEvent_t start{}; bool owns_start = true; double bytes = 0; uint32_t launches = 0; } open_scope;
    void profile_begin(int slot, hipStream_t s, double bytes) {
        if (!profiling) return;
        if (open_scope.slot == slot && open_scope.stream == s) { open_scope.bytes += bytes; open_scope.launches += 1; return; }
        const bool chained = open_scope.slot >= 0 && open_scope.stream == s;
        hipEvent_t boundary = profile_close();
        open_scope.slot = slot; open_scope.stream = s; open_scope.bytes = bytes; open_scope.launches = 1;
        if (chained) { open_scope.start = boundary; open_scope.owns_start = false; }
        else { open_scope.start = take_event(); open_scope.owns_start = true; (void)hipEventRecord(open_scope.start, s); }
    }
    hipEvent_t profile_close() {
        if (open_scope.slot < 0) return nullptr;
        hipEvent_t stop = take_event();
        (void)hipEventRecord(stop, open_scope.stream);
        profile_records.push_back({open_scope.slot, open_scope.start, stop, open_scope.bytes, open_scope.launches, open_scope.owns_start});
        open_scope.slot = -1;
        return stop;
    }
    int drain_profile() {
        for (auto& r : profile_records) {
            ST_HIP(hipEventSynchronize(r.stop));
            float ms = 0.0f;
            ST_HIP(hipEventElapsedTime(&ms, r.start, r.stop));
            profile_totals[r.slot].launches += r.launches; profile_totals[r.slot].total_ms += ms; profile_totals[r.slot].algorithmic_bytes += r.bytes;
            if (r.owns_start) event_pool.push_back(r.start);
            event_pool.push_back(r.stop);
        }
        profile_records.clear();
        return ST_OK;
    }

    // a composition into a buffer whose present copy has not finished waits for that copy (callers that alternate two
    // buffers never meet this)
    static void present_guard(CameraState& c, const void* out, hipStream_t s) {
        for (auto& p : c.present) if (p.pending && p.src == out) (void)hipStreamWaitEvent(s, p.ev_done, 0);
    }

    // ---- render (camera_controller.rs:87-174)
    int render(CameraState& c, void* out, hipStream_t stream) {
        if (!has_device) return fail(ST_ERR_NO_DEVICE, "render_camera on a host-only engine");
        if (!scene_uploaded) return fail(ST_ERR_INVALID_ARGUMENT, "st_tick must precede st_render_camera");
        ST_HIP(hipSetDevice(device));
        if (tick_work_in_flight) ST_HIP(hipStreamWaitEvent(stream, ev_tick, 0));  // a no-op when st_tick ran on this stream
        if (copy_in_flight) ST_HIP(hipStreamWaitEvent(stream, ev_copy, 0));       // likewise (st_tick already queued this wait on its own stream)
        if (rendered_before && last_render_stream != stream) mixed_render_streams = true;  // the null stream is a stream too
        last_render_stream = stream; rendered_before = true;
        const bool alt = c.frame % 2u == 1u;
        KArgs a{};
        a.cam = c.curr; a.prev_cam = c.prev;
        const SceneSet& scene = sets[live];
        a.bvh = static_cast<const float4*>(scene.bvh.ptr); a.tri_attr = static_cast<const float4*>(scene.tri_attr.ptr); a.instance_xforms = static_cast<const float4*>(scene.xforms.ptr);
        a.materials = static_cast<const GpuMaterial*>(scene.materials.ptr); a.material_base_packed = getenv("ST_NO_PACKED_BASE") ? nullptr : static_cast<const uint32_t*>(scene.base_packed.ptr); a.lights = static_cast<const GpuLight*>(light_sets[live_lights].buf.ptr);
        a.atlas = static_cast<const uchar4*>(d_atlas.ptr); a.blue_noise = static_cast<const uchar4*>(d_blue_noise.ptr); a.byte_luts = static_cast<const float*>(d_byte_luts.ptr);
        a.transmittance_lut = static_cast<const float4*>(d_transmittance.ptr); a.sky_lut = static_cast<const float4*>(d_sky.ptr);
        a.tri_slots = (uint32_t)(tri_geo.size() / 3u);
        a.count_bytes = count_bytes ? 1u : 0u;
        a.bvh_wide_len = (arithmetic == ST_ARITH_FAST && !count_bytes) ? device_wide_len : 0u;  // any-hit rays of the fast build; the byte counters are the binary stream's
        a.bvh_len = device_bvh_len; a.n_lights_buf = (uint32_t)gpu_lights.size(); a.light_count = light_count;
        a.atlas_w = atlas_w; a.atlas_h = atlas_h; a.sun_altitude = sun_altitude;
        a.sun_dir[0] = sun_dir_.x; a.sun_dir[1] = sun_dir_.y; a.sun_dir[2] = sun_dir_.z;
        auto P = [&](int id) { return c.plane[id]; };
        a.g0 = P(alt ? ST_BUF_PRIM_GBUFFER_D0_B : ST_BUF_PRIM_GBUFFER_D0_A); a.pg0 = P(alt ? ST_BUF_PRIM_GBUFFER_D0_A : ST_BUF_PRIM_GBUFFER_D0_B);
        a.g1 = P(alt ? ST_BUF_PRIM_GBUFFER_D1_B : ST_BUF_PRIM_GBUFFER_D1_A); a.pg1 = P(alt ? ST_BUF_PRIM_GBUFFER_D1_A : ST_BUF_PRIM_GBUFFER_D1_B);
        a.sm = P(alt ? ST_BUF_PRIM_SURFACE_MAP_B : ST_BUF_PRIM_SURFACE_MAP_A); a.psm = P(alt ? ST_BUF_PRIM_SURFACE_MAP_A : ST_BUF_PRIM_SURFACE_MAP_B);
        a.sn = P(ST_BUF_COUNT + (alt ? 1 : 0)); a.psn = P(ST_BUF_COUNT + (alt ? 0 : 1));
        a.reprojection = P(ST_BUF_REPROJECTION_MAP); a.velocity = P(ST_BUF_VELOCITY_MAP);
        for (int i = 0; i < 3; i++) a.di_res[i] = P(ST_BUF_DI_RESERVOIRS_0 + i);
        a.di_diff_samples = P(ST_BUF_DI_DIFF_SAMPLES); a.di_diff_prev_colors = P(ST_BUF_DI_DIFF_PREV_COLORS); a.di_diff_curr_colors = P(ST_BUF_DI_DIFF_CURR_COLORS);
        a.di_diff_moments = P(alt ? ST_BUF_DI_DIFF_MOMENTS_B : ST_BUF_DI_DIFF_MOMENTS_A); a.di_diff_prev_moments = P(alt ? ST_BUF_DI_DIFF_MOMENTS_A : ST_BUF_DI_DIFF_MOMENTS_B);
        a.di_diff_stash = P(ST_BUF_DI_DIFF_STASH); a.di_spec_samples = P(ST_BUF_DI_SPEC_SAMPLES);
        a.gi_d0 = P(ST_BUF_GI_D0); a.gi_d1 = P(ST_BUF_GI_D1); a.gi_d2 = P(ST_BUF_GI_D2);
        for (int i = 0; i < 4; i++) a.gi_res[i] = P(ST_BUF_GI_RESERVOIRS_0 + i);
        a.gi_diff_samples = P(ST_BUF_GI_DIFF_SAMPLES); a.gi_diff_prev_colors = P(ST_BUF_GI_DIFF_PREV_COLORS); a.gi_diff_curr_colors = P(ST_BUF_GI_DIFF_CURR_COLORS);
        a.gi_diff_moments = P(alt ? ST_BUF_GI_DIFF_MOMENTS_B : ST_BUF_GI_DIFF_MOMENTS_A); a.gi_diff_prev_moments = P(alt ? ST_BUF_GI_DIFF_MOMENTS_A : ST_BUF_GI_DIFF_MOMENTS_B);
        a.gi_diff_stash = P(ST_BUF_GI_DIFF_STASH); a.gi_spec_samples = P(ST_BUF_GI_SPEC_SAMPLES);
        a.ref_hits = P(ST_BUF_REF_HITS); a.ref_rays = P(ST_BUF_REF_RAYS); a.ref_colors = P(ST_BUF_REF_COLORS);
        a.dbg_used_memory = reinterpret_cast<uint32_t*>(P(ST_BUF_DBG_USED_MEMORY));
        a.width = c.desc.width; a.height = c.desc.height; a.row0 = c.row0; a.row1 = c.row1;
        a.frame = c.frame;
        a.tile_map = tile_map;

        const double rows = (double)(c.row1 - c.row0);
        auto slot_bytes = [&](int slot) {
            const KernelInfo& ki = kernel_info(slot);
            const double units = rows * (ki.half ? (double)(((c.desc.width + 7u) / 8u / 2u) * 8u) : (double)c.desc.width);
            return units * ki.bytes_per_unit;
        };
        hipStream_t cur = stream;  // stream the next launches go to (the GI chain may be diverted to side_stream)
        // `bits`: the reference passes this launch executes (StPassBit). Their unfused algorithmic bytes are what
        // kernel_info(slot) credits to the launch, so fusion shows up as a gain, not as a moved goalpost.
        last_launches.clear();
        bool mask_split = false;
        auto run = [&](int slot, uint64_t bits, auto&& launch) {
            if (last_launches.empty() || last_launches.back() != bits) last_launches.push_back(bits);  // a launch group is reported once
            if ((bits & pass_mask) != bits) { mask_split |= (bits & pass_mask) != 0; return; }
            const double bytes = slot_bytes(slot);
            a.ray_counter = c.counters + kCounterWordsPerSlot * slot;
            if (profiling && profile_kernel_events) {  // the dispatch's own timestamps (what rocprofv3's kernel trace reads)
                g_launch_events.start = take_event(); g_launch_events.stop = take_event();
                launch();
                profile_records.push_back({slot, g_launch_events.start, g_launch_events.stop, bytes, 1u, true});
                g_launch_events = LaunchEvents();
                return;
            }
            const bool atrous = slot == KS_DENOISE_WAVELET || slot == KS_DENOISE_WAVELET_12 || slot == KS_DENOISE_WAVELET_COMPOSE;
            profile_begin(profile_group_atrous && atrous ? (int)KS_DENOISE_WAVELET_FAMILY : slot, cur, bytes);
            launch();
        };
        auto seed = [&](uint32_t pass) { return pass_seed(base_seed, c.frame, pass); };
        const uint32_t mode = c.desc.mode;
        bool di_reprojected = false, gi_reprojected = false, composed = false, luts_generated_now = false;
        if (c.surface_map_replaced[0] || c.surface_map_replaced[1]) {  // ordered before the side stream like the LUTs
            const uint32_t cur = alt ? 1u : 0u;
            L.launch_refresh_internal_planes(a, (c.surface_map_replaced[cur] ? 1u : 0u) | (c.surface_map_replaced[cur ^ 1u] ? 2u : 0u), stream);
            c.surface_map_replaced[0] = c.surface_map_replaced[1] = false; luts_generated_now = true;
        }
        if (mode != ST_MODE_BVH_HEATMAP) {  // AtmospherePass::run (passes/atmosphere.rs:78-110)
            if (!atmosphere_initialized) {
                L.launch_atmosphere_static(static_cast<float4*>(d_transmittance.ptr), static_cast<float4*>(d_scattering.ptr), stream);
                atmosphere_initialized = true; luts_generated_now = true;
            }
            if (!sky_known || known_sun_altitude != sun_altitude) {
                L.launch_atmosphere_sky(static_cast<const float4*>(d_transmittance.ptr), static_cast<const float4*>(d_scattering.ptr), sun_altitude,
                                      static_cast<float4*>(d_sky.ptr), stream);
                sky_known = true; known_sun_altitude = sun_altitude; luts_generated_now = true;
            }
        }
        if (mode == ST_MODE_BVH_HEATMAP) {
            run(KS_BVH_HEATMAP, ST_PASS_BVH_HEATMAP, [&] { L.launch_bvh_heatmap(a, cur); });
        } else if (mode == ST_MODE_REFERENCE) {
            for (uint32_t d = 0; d <= c.desc.depth; d++) {
                run(KS_REF_TRACING, ST_PASS_REF_TRACING, [&] { L.launch_ref_tracing(a, d, cur); });
                run(KS_REF_SHADING, ST_PASS_REF_SHADING, [&] { L.launch_ref_shading(a, seed(SEED_REF_SHADING + d), d, cur); });
            }
            run(KS_REF_SHADING, ST_PASS_REF_SHADING, [&] { L.launch_ref_shading(a, seed(SEED_REF_SHADING + 255u), 255u, cur); });
        } else {
            const bool needs_di = mode == ST_MODE_IMAGE || mode == ST_MODE_DI_DIFFUSE || mode == ST_MODE_DI_SPECULAR;
            const bool needs_gi = mode == ST_MODE_IMAGE || mode == ST_MODE_GI_DIFFUSE || mode == ST_MODE_GI_SPECULAR;
            const bool denoise = c.desc.denoise != 0u;
            const bool any_objects = !instances.empty();
            const bool tracing = c.frame % 6u < 4u;
            const uint32_t gi_source = (tracing && c.frame % 2u == 1u) ? 1u : 0u;
            const uint32_t pseed = seed(SEED_GI_PREVIEW);  // one seed for both preview passes (passes/gi_preview_resampling.rs:60-74)
            // GI history hand-over by pointer swap instead of gi_resolving's copy (CameraState::gi_aliased says when)
            const bool whole_graph = pass_mask == ~0ull;  // a row window (multi-GPU band) changes which pixels a pass owns, not which passes follow it
            const bool gi_runs = needs_gi && any_objects;
            if (c.gi_aliased && gi_runs && !whole_graph) { const int rc = materialize_gi_history(c); if (rc) return rc; }
            // Fast build only: the reference's copy is a decode + re-encode of every reservoir, which is not the identity on all
            // bit patterns (the octahedral normal of a few records per frame moves by an ulp), and the exact build owes the
            // parity suite those bits.
            const bool swap_gi_history = alias_gi_history && arithmetic == ST_ARITH_FAST && gi_runs && whole_graph && gi_source == 0u;
            if (gi_runs && whole_graph) c.gi_aliased = false;  // this frame's temporal pass rewrites GI_RESERVOIRS_1 completely
            a.gi_skip_history_copy = swap_gi_history ? 1u : 0u;
            // estimate_variance's long-history branch rides in the fused reproject stages (st_passes.h denoise_reproject_finish);
            // the variance launch then serves the short-history pixels only, in place, and the strides-1+2 launch reads curr_colors
            a.tile_mask = c.tile_mask;
            // both GI preview passes + resolving in one launch for the pixels whose second pass draws no neighbour (k_gi.hip
            // k_gi_preview_both); the second-pass launch then serves the flagged rest
            a.gi_late_mask = c.tile_mask ? c.tile_mask + c.tile_mask_tiles : nullptr; a.gi_preview_late = 0u;
            const bool gi_preview_both = preview_both && whole_graph && fuse && gi_runs && a.gi_late_mask;
            a.variance_in_reproject = (variance_in_reproject && whole_graph && fuse && fuse_wavelet && denoise && needs_di && needs_gi && any_objects && c.tile_mask) ? 1u : 0u;
            // di_spatial's scratch records (di_diff_samples / curr_colors / stash as the reference binds them) are dead stores
            // when the fused launch is followed by resolving, denoise-reproject and the a-trous chain of the same frame
            const bool even_tiles_x = (((a.width + 7u) / 8u) & 1u) == 0u;
            a.lean = 0u;
            if (lean_frame && arithmetic == ST_ARITH_FAST && whole_graph && fuse && denoise && any_objects && mode == ST_MODE_IMAGE) {
                a.lean = kLeanPrim | kLeanSamples;
                if (fuse_gi_reproj && tracing && even_tiles_x) a.lean |= kLeanGiRes2;
                if (gi_preview_both) a.lean |= kLeanGiMid;
            }
            // frame composition rides in the last a-trous pass (k_denoise.hip k_denoise_wavelet_far<true>)
            const bool compose_in_wavelet = fuse_compose && arithmetic == ST_ARITH_FAST && whole_graph && fuse && denoise && out != nullptr && mode == ST_MODE_IMAGE && any_objects;
            a.skip_dead_scratch = (skip_scratch_stores && whole_graph && fuse && fuse_spatial && ((((a.width + 7u) / 8u) & 1u) == 0u) && needs_di && denoise && any_objects) ? 1u : 0u;

            auto do_prim = [&] {
                if (fuse && any_objects) run(KS_PRIM_VISIBILITY_REPROJECTION, ST_PASS_PRIM_VISIBILITY | ST_PASS_FRAME_REPROJECTION, [&] { L.launch_prim_visibility(a, true, cur); });
                else run(KS_PRIM_VISIBILITY, ST_PASS_PRIM_VISIBILITY, [&] { L.launch_prim_visibility(a, false, cur); });
                if (any_objects && !fuse) run(KS_FRAME_REPROJECTION, ST_PASS_FRAME_REPROJECTION, [&] { L.launch_frame_reprojection(a, cur); });
            };
            // DI up to temporal resampling touches only the DI reservoirs and read-only frame inputs ...
            auto do_di_head = [&] {
                if (fuse && fuse_di_head) run(KS_DI_SAMPLING_TEMPORAL, ST_PASS_DI_SAMPLING | ST_PASS_DI_TEMPORAL, [&] { L.launch_di_sampling_temporal(a, seed(SEED_DI_SAMPLING), seed(SEED_DI_TEMPORAL), cur); });
                else {
                    run(KS_DI_SAMPLING, ST_PASS_DI_SAMPLING, [&] { L.launch_di_sampling(a, seed(SEED_DI_SAMPLING), cur); });
                    run(KS_DI_TEMPORAL, ST_PASS_DI_TEMPORAL, [&] { L.launch_di_temporal(a, seed(SEED_DI_TEMPORAL), cur); });
                }
            };
            // ... the spatial passes use the denoiser's planes as scratch (passes/di_spatial_resampling.rs binds
            // di_diff_samples / curr_colors / stash), and resolving writes the planes the denoiser reads
            auto do_di_tail = [&] {
                // the half-resolution grid drops the last tile column when the tile count is odd (`(size + 7) / 8 / (2, 1)`), while
                // the stand-alone trace pass still visits those pixels: only an even tile count lets one launch cover all three
                const bool even_tiles = (((a.width + 7u) / 8u) & 1u) == 0u;
                if (fuse && fuse_spatial && even_tiles) run(KS_DI_SPATIAL_FUSED, ST_PASS_DI_SPATIAL_PICK | ST_PASS_DI_SPATIAL_TRACE | ST_PASS_DI_SPATIAL_SAMPLE, [&] { L.launch_di_spatial_fused(a, seed(SEED_DI_SPATIAL_PICK), seed(SEED_DI_SPATIAL_SAMPLE), cur); });
                else {
                    run(KS_DI_SPATIAL_PICK, ST_PASS_DI_SPATIAL_PICK, [&] { L.launch_di_spatial_pick(a, seed(SEED_DI_SPATIAL_PICK), cur); });
                    run(KS_DI_SPATIAL_TRACE, ST_PASS_DI_SPATIAL_TRACE, [&] { L.launch_spatial_trace(a, a.di_diff_samples, a.di_diff_curr_colors, a.di_diff_stash, cur); });
                    run(KS_DI_SPATIAL_SAMPLE, ST_PASS_DI_SPATIAL_SAMPLE, [&] { L.launch_di_spatial_sample(a, seed(SEED_DI_SPATIAL_SAMPLE), cur); });
                }
                if (fuse && denoise) { run(KS_DI_RESOLVING_REPROJECT, ST_PASS_DI_RESOLVING | ST_PASS_DENOISE_REPROJECT_DI, [&] { L.launch_di_resolving(a, true, cur); }); di_reprojected = true; }
                else run(KS_DI_RESOLVING, ST_PASS_DI_RESOLVING, [&] { L.launch_di_resolving(a, false, cur); });
            };
            auto do_di = [&] { do_di_head(); do_di_tail(); };
            // GI up to the first preview pass: touches only reservoirs, gi_d0..2 and read-only frame inputs
            auto do_gi_head = [&] {
                // on tracing frames gi_temporal is the only reader of the reprojected reservoirs and does the reprojection itself
                // ... and on validation frames of a whole frame both of its readers — the sampling launch for the half of the pixels it
                // re-traces, then gi_temporal, which stores it — do it for themselves (ST_NO_FUSE_GI_VALIDATION=1: a launch of its own)
                const bool fuse_gi_validation = fuse && fuse_gi_reproj && fuse_gi_sampling && fuse_gi_valid && !tracing && whole_graph;
                const bool fuse_gi_reprojection = (fuse && fuse_gi_reproj && tracing) || fuse_gi_validation;
                auto temporal = [&] {
                    if (fuse_gi_reprojection) run(KS_GI_REPROJECTION_TEMPORAL, ST_PASS_GI_REPROJECTION | ST_PASS_GI_TEMPORAL, [&] { L.launch_gi_temporal(a, seed(SEED_GI_TEMPORAL), true, cur); });
                    else run(KS_GI_TEMPORAL, ST_PASS_GI_TEMPORAL, [&] { L.launch_gi_temporal(a, seed(SEED_GI_TEMPORAL), false, cur); });
                };
                if (!fuse_gi_reprojection) run(KS_GI_REPROJECTION, ST_PASS_GI_REPROJECTION, [&] { L.launch_gi_reprojection(a, cur); });
                auto sampling = [&] {
                    if (fuse && fuse_gi_sampling) { run(KS_GI_SAMPLING_AB, ST_PASS_GI_SAMPLING_A | ST_PASS_GI_SAMPLING_B, [&] { L.launch_gi_sampling_ab(a, seed(SEED_GI_SAMPLING_A), seed(SEED_GI_SAMPLING_B), fuse_gi_validation, cur); }); return; }
                    run(KS_GI_SAMPLING_A, ST_PASS_GI_SAMPLING_A, [&] { L.launch_gi_sampling_a(a, seed(SEED_GI_SAMPLING_A), cur); });
                    run(KS_GI_SAMPLING_B, ST_PASS_GI_SAMPLING_B, [&] { L.launch_gi_sampling_b(a, seed(SEED_GI_SAMPLING_B), cur); });
                };
                if (tracing) {
                    if (c.frame % 2u == 0u) sampling();
                    temporal();
                    if (c.frame % 2u == 1u) {
                        if (fuse && fuse_spatial && ((((a.width + 7u) / 8u) & 1u) == 0u))
                            run(KS_GI_SPATIAL_FUSED, ST_PASS_GI_SPATIAL_PICK | ST_PASS_GI_SPATIAL_TRACE | ST_PASS_GI_SPATIAL_SAMPLE, [&] { L.launch_gi_spatial_fused(a, seed(SEED_GI_SPATIAL_PICK), seed(SEED_GI_SPATIAL_SAMPLE), cur); });
                        else {
                            run(KS_GI_SPATIAL_PICK, ST_PASS_GI_SPATIAL_PICK, [&] { L.launch_gi_spatial_pick(a, seed(SEED_GI_SPATIAL_PICK), cur); });
                            run(KS_GI_SPATIAL_TRACE, ST_PASS_GI_SPATIAL_TRACE, [&] { L.launch_spatial_trace(a, a.gi_d0, a.gi_d1, a.gi_d2, cur); });
                            run(KS_GI_SPATIAL_SAMPLE, ST_PASS_GI_SPATIAL_SAMPLE, [&] { L.launch_gi_spatial_sample(a, seed(SEED_GI_SPATIAL_SAMPLE), cur); });
                        }
                    }
                } else {
                    sampling();
                    temporal();
                }
                if (!gi_preview_both) run(KS_GI_PREVIEW, ST_PASS_GI_PREVIEW_0, [&] { L.launch_gi_preview(a, pseed, 0u, gi_source == 0 ? a.gi_res[1] : a.gi_res[2], a.gi_res[3], cur); });
            };
            // second preview pass + resolving (+ reproject): the first GI stage that writes planes the denoiser/composition read
            auto do_gi_tail = [&] {
                if (gi_preview_both) {
                    // one launch group of two kernels = one set of pass bits
                    const uint64_t group = ST_PASS_GI_PREVIEW_0 | ST_PASS_GI_PREVIEW_1 | ST_PASS_GI_RESOLVING | (denoise ? (uint64_t)ST_PASS_DENOISE_REPROJECT_GI : 0ull);
                    run(denoise ? KS_GI_PREVIEW_BOTH : KS_GI_PREVIEW_BOTH_NO_REPROJECT, group, [&] { L.launch_gi_preview_both(a, pseed, gi_source == 0 ? a.gi_res[1] : a.gi_res[2], a.gi_res[3], gi_source, denoise, cur); });
                    a.gi_preview_late = 1u;
                    a.gi_mid_src = (a.lean & kLeanGiMid) ? (gi_source == 0 ? a.gi_res[1] : a.gi_res[2]) : nullptr;
                    run(KS_GI_PREVIEW_LATE, group, [&] { L.launch_gi_preview_resolve(a, pseed, 1u, a.gi_res[3], gi_source, denoise, cur); });
                    a.gi_preview_late = 0u; a.gi_mid_src = nullptr;
                    if (denoise) gi_reprojected = true;
                } else if (fuse) {
                    if (denoise) { run(KS_GI_PREVIEW_RESOLVE_REPROJECT, ST_PASS_GI_PREVIEW_1 | ST_PASS_GI_RESOLVING | ST_PASS_DENOISE_REPROJECT_GI, [&] { L.launch_gi_preview_resolve(a, pseed, 1u, a.gi_res[3], gi_source, true, cur); }); gi_reprojected = true; }
                    else run(KS_GI_PREVIEW_RESOLVE, ST_PASS_GI_PREVIEW_1 | ST_PASS_GI_RESOLVING, [&] { L.launch_gi_preview_resolve(a, pseed, 1u, a.gi_res[3], gi_source, false, cur); });
                } else {
                    run(KS_GI_PREVIEW, ST_PASS_GI_PREVIEW_1, [&] { L.launch_gi_preview(a, pseed, 1u, a.gi_res[3], a.gi_res[0], cur); });
                    run(KS_GI_RESOLVING, ST_PASS_GI_RESOLVING, [&] { L.launch_gi_resolving(a, gi_source, cur); });
                }
                if (swap_gi_history) {  // the launches above were told not to copy (KArgs::gi_skip_history_copy)
                    std::swap(c.plane[ST_BUF_GI_RESERVOIRS_0], c.plane[ST_BUF_GI_RESERVOIRS_1]);
                    c.gi_aliased = true;
                }
            };
            auto do_denoise = [&] {
                if (!denoise) return;
                // the denoiser can use its own block -> tile mapping (see `tile_map_denoise`)
                struct MapScope { KArgs& a; uint32_t saved; MapScope(KArgs& a_, uint32_t m) : a(a_), saved(a_.tile_map) { a.tile_map = m; } ~MapScope() { a.tile_map = saved; } } map_scope(a, tile_map_denoise);
                if (!di_reprojected) run(KS_DENOISE_REPROJECT, ST_PASS_DENOISE_REPROJECT_DI, [&] { L.launch_denoise_reproject(a, a.di_diff_prev_colors, a.di_diff_prev_moments, a.di_diff_samples, a.di_diff_curr_colors, a.di_diff_moments, cur); });
                if (!gi_reprojected) run(KS_DENOISE_REPROJECT, ST_PASS_DENOISE_REPROJECT_GI, [&] { L.launch_denoise_reproject(a, a.gi_diff_prev_colors, a.gi_diff_prev_moments, a.gi_diff_samples, a.gi_diff_curr_colors, a.gi_diff_moments, cur); });
                // ping-pong (passes/frame_denoising.rs:87-110): stash -> prev -> stash -> curr -> stash -> curr
                float4* di[3] = {a.di_diff_stash, a.di_diff_prev_colors, a.di_diff_curr_colors};
                float4* gi[3] = {a.gi_diff_stash, a.gi_diff_prev_colors, a.gi_diff_curr_colors};
                const int in_ix[5] = {0, 1, 0, 2, 0}, out_ix[5] = {1, 0, 2, 0, 2};
                uint32_t first = 0;
                if (fuse && fuse_wavelet) {
                    // variance estimation + strides 1 and 2 form one launch group of two kernels: the variance pass hands its
                    // output over in an internal pair of planes (k_denoise.hip k_denoise_wavelet_12 says why), so the stash
                    // planes receive the stride-2 result directly. One group = one set of pass bits (st_debug_set_pass_mask).
                    const uint64_t group = ST_PASS_DENOISE_VARIANCE | ST_PASS_DENOISE_WAVELET_0 | ((uint64_t)ST_PASS_DENOISE_WAVELET_0 << 1);
                    // (with KArgs::variance_in_reproject the hand-over planes are the reproject stages' own outputs)
                    float4* tmp_di = a.variance_in_reproject ? a.di_diff_curr_colors : P(ST_BUF_COUNT + 2);
                    float4* tmp_gi = a.variance_in_reproject ? a.gi_diff_curr_colors : P(ST_BUF_COUNT + 3);
                    run(KS_DENOISE_VARIANCE, group, [&] { L.launch_denoise_variance(a, tmp_di, tmp_gi, cur); });
                    run(KS_DENOISE_WAVELET_12, group, [&] { L.launch_denoise_wavelet_12(a, 1.0f, 2.0f, tmp_di, di[1], di[0], tmp_gi, gi[1], gi[0], cur); });
                    first = 2;
                } else run(KS_DENOISE_VARIANCE, ST_PASS_DENOISE_VARIANCE, [&] { L.launch_denoise_variance(a, a.di_diff_stash, a.gi_diff_stash, cur); });
                for (uint32_t nth = first; nth < 5; nth++) {
                    if (nth == 4u && compose_in_wavelet) {
                        present_guard(c, out, cur);
                        run(KS_DENOISE_WAVELET_COMPOSE, ((uint64_t)ST_PASS_DENOISE_WAVELET_0 << nth) | ST_PASS_COMPOSITION, [&] {
                            L.launch_denoise_wavelet_compose(a, 1u << nth, (float)(1u + nth), di[in_ix[nth]], di[out_ix[nth]], gi[in_ix[nth]], gi[out_ix[nth]], mode, out, c.out_format, a.lean == 0u, cur); });
                        composed = true;
                        continue;
                    }
                    run(KS_DENOISE_WAVELET, (uint64_t)ST_PASS_DENOISE_WAVELET_0 << nth, [&] { L.launch_denoise_wavelet(a, 1u << nth, (float)(1u + nth), di[in_ix[nth]], di[out_ix[nth]], gi[in_ix[nth]], gi[out_ix[nth]], cur); });
                }
            };
            auto do_compose = [&] {
                if (!out || composed) return;
                present_guard(c, out, cur);
                const float4* di_diff = (denoise && (mode == ST_MODE_IMAGE || mode == ST_MODE_DI_DIFFUSE)) ? a.di_diff_curr_colors : a.di_diff_samples;
                const float4* gi_diff = (denoise && (mode == ST_MODE_IMAGE || mode == ST_MODE_GI_DIFFUSE)) ? a.gi_diff_curr_colors : a.gi_diff_samples;
                run(KS_COMPOSITION, ST_PASS_COMPOSITION, [&] { L.launch_composition(a, mode, di_diff, gi_diff, out, c.out_format, cur); });
                composed = true;
            };

            // per-kernel profiling runs the graph serially on `stream`: a launch's event pair then times that kernel alone,
            // not the kernels of the other stream it would share the chip with
            if (overlap && !profiling && needs_di && needs_gi && any_objects) {
                // Two streams, software-pipelined across frames: `side` carries primary visibility and the GI chain; `stream`
                // carries the DI passes (sampling + temporal resampling too, by default: measured 1.2 % on the dungeon, nothing
                // on Cornell, against running them behind primary visibility on `side`), the denoiser and composition. Events
                // express the true data dependencies only, so the reservoir passes of frame N+1 overlap the denoiser of frame N:
                //   prim(N+1)      after DI tail(N)       — it overwrites frame N's "previous" G-buffer + the reprojection map
                //   GI tail(N+1)   after frame N is done  — it writes gi sample/colour/moment planes the denoiser + composition read
                //   DI head(N+1)   after prim(N+1)        (ev_di_head)
                //   DI tail(N+1)   after DI head(N+1)     (and after frame N's composition by stream order: its scratch aliases
                //                                          the denoiser's planes)
                //   denoiser(N+1)  after GI tail(N+1)
                if (!c.side_stream) {
                    int least = 0, greatest = 0;
                    (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
                    const int priority = side_priority > 0 ? greatest : (side_priority < 0 ? least : 0);
                    ST_HIP(hipStreamCreateWithPriority(&c.side_stream, hipStreamNonBlocking, priority));
                    for (hipEvent_t* e : {&c.ev_di_head, &c.ev_gi_done, &c.ev_prim_ok, &c.ev_frame_done, &c.ev_setup}) ST_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
                }
                // LUT generation issued on `stream` in this call must precede the side stream's consumers. (Do NOT do this
                // unconditionally: an event recorded on `stream` here completes only after frame N's denoiser, which would
                // serialise prim(N+1) behind it. Uploads in st_tick are followed by a host-side stream sync.)
                if (luts_generated_now) { ST_HIP(hipEventRecord(c.ev_setup, stream)); ST_HIP(hipStreamWaitEvent(c.side_stream, c.ev_setup, 0)); }
                // copies st_tick queued without joining the stream (staged uploads, dynamic images): they sit behind frame N on
                // the tick's stream, so a frame that follows a scene change gives up the prim(N+1) / denoiser(N) overlap
                if (tick_work_in_flight) ST_HIP(hipStreamWaitEvent(c.side_stream, ev_tick, 0));
                if (copy_in_flight) ST_HIP(hipStreamWaitEvent(c.side_stream, ev_copy, 0));  // independent of frame N: the overlap stays
                if (c.have_prev_frame_events) ST_HIP(hipStreamWaitEvent(c.side_stream, c.ev_prim_ok, 0));
                cur = c.side_stream;
                do_prim();
                if (!di_head_on_main) do_di_head();
                ST_HIP(hipEventRecord(c.ev_di_head, c.side_stream));  // primary visibility (+ DI head) of this frame are through
                do_gi_head();
                if (c.have_prev_frame_events) ST_HIP(hipStreamWaitEvent(c.side_stream, c.ev_frame_done, 0));
                do_gi_tail();
                ST_HIP(hipEventRecord(c.ev_gi_done, c.side_stream));
                cur = stream;
                ST_HIP(hipStreamWaitEvent(stream, c.ev_di_head, 0));
                if (di_head_on_main) do_di_head();
                do_di_tail();
                // stand-alone denoise reprojection kernels (unfused path) still read the reprojection map: prim(N+1) may
                // only start once they are through
                const bool reproject_later = denoise && !fuse;
                if (!reproject_later) ST_HIP(hipEventRecord(c.ev_prim_ok, stream));
                ST_HIP(hipStreamWaitEvent(stream, c.ev_gi_done, 0));
                do_denoise();
                if (reproject_later) ST_HIP(hipEventRecord(c.ev_prim_ok, stream));
                do_compose();
                ST_HIP(hipEventRecord(c.ev_frame_done, stream));
                c.have_prev_frame_events = true;
            } else {
                do_prim();
                if (any_objects) {
                    if (needs_di) do_di();
                    if (needs_gi) { do_gi_head(); do_gi_tail(); }
                }
                do_denoise();
                do_compose();
                if (c.side_stream) { ST_HIP(hipEventRecord(c.ev_prim_ok, stream)); ST_HIP(hipEventRecord(c.ev_frame_done, stream)); }
            }
        }
        if (out && !composed) {
            present_guard(c, out, cur);
            const bool dn = c.desc.denoise != 0u;
            const float4* di_diff = (dn && (mode == ST_MODE_IMAGE || mode == ST_MODE_DI_DIFFUSE)) ? a.di_diff_curr_colors : a.di_diff_samples;
            const float4* gi_diff = (dn && (mode == ST_MODE_IMAGE || mode == ST_MODE_GI_DIFFUSE)) ? a.gi_diff_curr_colors : a.gi_diff_samples;
            run(KS_COMPOSITION, ST_PASS_COMPOSITION, [&] { L.launch_composition(a, mode, di_diff, gi_diff, out, c.out_format, cur); });
        }
        if (alternating) {  // the end of the last frame that reads this copy of the scene
            SceneSet& l = sets[live];
            if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
            ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
        }
        if (lights_alternating) {
            LightSet& l = light_sets[live_lights];
            if (!l.free_ev) ST_HIP(hipEventCreateWithFlags(&l.free_ev, hipEventDisableTiming));
            ST_HIP(hipEventRecord(l.free_ev, stream)); l.busy = true;
        }
        profile_close();
        ST_HIP(hipGetLastError());
        if (mask_split) return fail(ST_ERR_INVALID_ARGUMENT, "the pass mask splits a fused launch (st_debug_last_launches lists the launch groups)");
        return ST_OK;
    }
};

}  // namespace st

using namespace st;
static Engine* E(StEngine* e) { return reinterpret_cast<Engine*>(e); }
#define ST_REQUIRE(cond, msg) do { if (!(cond)) return fail(ST_ERR_INVALID_ARGUMENT, msg); } while (0)

extern "C" {

const char* st_last_error(void) { return g_last_error.c_str(); }
// st_gltf.cpp reports through the same thread-local message
extern "C" int st_internal_fail(int status, const char* message) { return fail(status, message ? message : ""); }

int st_engine_create(int device_ordinal, StEngine** out) {
    ST_REQUIRE(out, "out is NULL");
    std::unique_ptr<Engine> e(new Engine());
    if (device_ordinal >= 0) {
        int n = 0;
        if (hipGetDeviceCount(&n) != hipSuccess || n <= device_ordinal)
            return fail(ST_ERR_NO_DEVICE, "no HIP device with that ordinal (this library has no CPU rendering path)");
        ST_HIP(hipSetDevice(device_ordinal));
        e->device = device_ordinal; e->has_device = true;
        // LUT storage (zero until the first non-heatmap render generates them)
        const size_t lut_bytes[3] = {sizeof(float4) * 256 * 64, sizeof(float4) * 32 * 32, sizeof(float4) * 256 * 256};
        DeviceArray* luts[3] = {&e->d_transmittance, &e->d_scattering, &e->d_sky};
        for (int i = 0; i < 3; i++) { ST_HIP(hipMalloc(&luts[i]->ptr, lut_bytes[i])); luts[i]->capacity = lut_bytes[i]; ST_HIP(hipMemset(luts[i]->ptr, 0, lut_bytes[i])); }
        ST_HIP(hipMalloc(&e->d_byte_luts.ptr, sizeof(float) * 1024)); e->d_byte_luts.capacity = sizeof(float) * 1024;
        e->L.launch_build_byte_luts(static_cast<float*>(e->d_byte_luts.ptr), nullptr);
        ST_HIP(hipDeviceSynchronize());
    }
    *out = reinterpret_cast<StEngine*>(e.release());
    return ST_OK;
}
void st_engine_destroy(StEngine* e) { delete E(e); }

int st_mesh_insert(StEngine* e, StHandle id, const StMeshTriangle* t, size_t count) {
    ST_REQUIRE(e && (t || count == 0), "null argument");
    if (count == 0) return fail(ST_ERR_EMPTY_MESH, "mesh contains no triangles");
    E(e)->meshes[id].assign(t, t + count);
    return ST_OK;
}
int st_mesh_remove(StEngine* e, StHandle id) { ST_REQUIRE(e, "null engine"); E(e)->meshes.erase(id); return ST_OK; }

int st_material_insert(StEngine* e, StHandle id, const StMaterial* m) {
    ST_REQUIRE(e && m, "null argument");
    Engine* en = E(e);
    auto it = en->material_slot.find(id);
    if (it != en->material_slot.end()) en->materials[it->second] = *m;
    else {
        size_t b, end_;
        uint32_t slot;
        if (en->material_free.take(1, &b, &end_)) slot = (uint32_t)b;  // materials.rs:48-50 (the slot keeps its previous contents)
        else { en->materials.push_back(*m); slot = (uint32_t)en->materials.size() - 1u; }
        en->material_slot[id] = slot;
    }
    en->materials_dirty = true;
    return ST_OK;
}
int st_material_has(StEngine* e, StHandle id) { return e && E(e)->material_slot.count(id) ? 1 : 0; }
int st_material_remove(StEngine* e, StHandle id) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    auto it = en->material_slot.find(id);
    if (it == en->material_slot.end()) return ST_OK;
    en->material_free.give(it->second, it->second);  // `give(id..id)`: an empty range, as in materials.rs:74
    en->material_slot.erase(it);
    en->materials_dirty = true;
    return ST_OK;
}

// Images::insert (images.rs:54-105): finds the rectangle for image `id` and makes the host copy of the atlas tall enough.
static int place_image(Engine* en, StHandle id, uint32_t w, uint32_t h, Engine::ImageRec* out) {
    constexpr uint32_t kAtlasW = Engine::kAtlasW;
    if (w > kAtlasW) return fail(ST_ERR_ATLAS_FULL, "image wider than the atlas");
    auto it = en->images.find(id);
    Engine::ImageRec rec;
    if (it != en->images.end() && it->second.w == w && it->second.h == h) rec = it->second;  // same size: rewritten in place (images.rs:61-63)
    else {
        if (it != en->images.end()) {  // another size: the old rectangle is given back first (images.rs:64-66)
            en->atlas_rects.release(it->second.x, it->second.y, it->second.w);
            en->images.erase(it);
            en->device_images.erase(id);
            en->materials_dirty = true;
        }
        rec = {0, 0, w, h};
        if (!en->atlas_rects.allocate(w, h, &rec.x, &rec.y)) return fail(ST_ERR_ATLAS_FULL, "no more space in the atlas");
    }
    const uint32_t need_h = rec.y + h;
    if (en->atlas_w == 0) en->atlas_w = kAtlasW;
    if (need_h > en->atlas_h) {
        // grow in 256-row steps; rects are stored in texels, so existing materials stay valid after a rebuild
        en->atlas_h = (need_h + 255u) & ~255u;
        en->atlas.resize((size_t)en->atlas_w * en->atlas_h * 4, 0);
    }
    en->images[id] = rec;
    en->atlas_dirty = true; en->materials_dirty = true;
    *out = rec;
    return ST_OK;
}

int st_image_insert_rgba8(StEngine* e, StHandle id, uint32_t w, uint32_t h, const uint8_t* rgba, int /*srgb*/) {
    ST_REQUIRE(e && rgba && w && h && id, "bad image");
    Engine* en = E(e);
    Engine::ImageRec rec;
    if (int rc = place_image(en, id, w, h, &rec)) return rc;
    en->device_images.erase(id);
    for (uint32_t y = 0; y < h; y++) memcpy(&en->atlas[((size_t)(rec.y + y) * en->atlas_w + rec.x) * 4], rgba + (size_t)y * w * 4, (size_t)w * 4);
    return ST_OK;
}
int st_image_insert_device_rgba8(StEngine* e, StHandle id, uint32_t w, uint32_t h, const void* device_rgba, size_t row_pitch_bytes, int is_dynamic) {
    ST_REQUIRE(e && device_rgba && w && h && id, "bad image");
    ST_REQUIRE(row_pitch_bytes >= (size_t)w * 4, "row pitch smaller than a row");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "device images need a device engine");
    Engine::ImageRec rec;
    if (int rc = place_image(en, id, w, h, &rec)) return rc;
    en->device_images[id] = Engine::DeviceImage{device_rgba, row_pitch_bytes, is_dynamic != 0, true};
    return ST_OK;
}
int st_image_remove(StEngine* e, StHandle id) {  // images.rs:107-113
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    auto it = en->images.find(id);
    if (it == en->images.end()) return ST_OK;
    en->atlas_rects.release(it->second.x, it->second.y, it->second.w);
    en->images.erase(it);
    en->device_images.erase(id);
    en->materials_dirty = true;
    return ST_OK;
}
int st_debug_image_rect(StEngine* e, StHandle id, uint32_t out_xywh[4]) {
    ST_REQUIRE(e && out_xywh, "null argument");
    auto it = E(e)->images.find(id);
    if (it == E(e)->images.end()) return fail(ST_ERR_INVALID_ARGUMENT, "no such image");
    out_xywh[0] = it->second.x; out_xywh[1] = it->second.y; out_xywh[2] = it->second.w; out_xywh[3] = it->second.h;
    return ST_OK;
}

int st_instance_insert(StEngine* e, StHandle id, StHandle mesh, StHandle material, const float xform[12]) {
    ST_REQUIRE(e && xform, "null argument");
    Engine* en = E(e);
    const Affine x = affine_from12(xform);
    for (auto& r : en->instances)
        if (r.id == id) { r.prev_xform = r.xform; r.mesh = mesh; r.material = material; r.xform = x; r.xform_inv = affine_inverse(x); r.dirty = true; en->instances_dirty = true; return ST_OK; }
    uint32_t xslot;
    if (!en->xslot_free.empty()) { xslot = en->xslot_free.back(); en->xslot_free.pop_back(); }
    else { xslot = (uint32_t)(en->instance_xforms.size() / 8u); en->instance_xforms.resize(en->instance_xforms.size() + 8u, make_float4(0, 0, 0, 0)); }
    en->instances.push_back({id, mesh, material, x, affine_inverse(x), x, true, xslot});
    en->instances_dirty = true;
    return ST_OK;
}
int st_instance_remove(StEngine* e, StHandle id) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    for (size_t i = 0; i < en->instances.size(); i++)
        if (en->instances[i].id == id) { en->xslot_free.push_back(en->instances[i].xslot); en->instances.erase(en->instances.begin() + i); en->instances_dirty = true; break; }
    en->drop_instance_triangles(id);
    return ST_OK;
}
int st_light_insert(StEngine* e, StHandle id, const StLight* l) { ST_REQUIRE(e && l, "null argument"); E(e)->insert_light(id, *l); return ST_OK; }
int st_light_remove(StEngine* e, StHandle id) { ST_REQUIRE(e, "null engine"); E(e)->remove_light(id); return ST_OK; }
int st_sun_update(StEngine* e, float azimuth, float altitude) { ST_REQUIRE(e, "null engine"); E(e)->sun_azimuth = azimuth; E(e)->sun_altitude = altitude; E(e)->sun_dirty = true; return ST_OK; }

int st_camera_create(StEngine* e, const StCamera* c, StHandle* out) {
    ST_REQUIRE(e && c && out && c->width && c->height, "bad camera");
    Engine* en = E(e);
    std::unique_ptr<CameraState> s(new CameraState());
    s->desc = *c;
    s->curr = Engine::serialize_camera(*c); s->prev = s->curr;
    const int rc = en->allocate_camera(*s);
    if (rc) return rc;
    *out = en->next_camera++;
    en->cameras[*out] = std::move(s);
    return ST_OK;
}
int st_camera_update(StEngine* e, StHandle h, const StCamera* c) {
    ST_REQUIRE(e && c && c->width && c->height, "bad camera");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    CameraState& s = *it->second;
    const bool invalidated = s.desc.mode != c->mode || s.desc.denoise != c->denoise || s.desc.depth != c->depth || s.desc.width != c->width || s.desc.height != c->height;
    s.desc = *c;
    s.prev = s.curr;
    s.curr = Engine::serialize_camera(*c);
    if (invalidated) { if (en->has_device) { ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize()); } return en->allocate_camera(s); }  // camera.rs:17-48: buffers are rebuilt
    return ST_OK;
}
int st_camera_delete(StEngine* e, StHandle h) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return ST_OK;
    if (en->has_device) { ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize()); Engine::release_camera(*it->second); }
    en->cameras.erase(it);
    return ST_OK;
}
int st_camera_set_rows(StEngine* e, StHandle h, uint32_t y0, uint32_t y1) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    CameraState& s = *it->second;
    if (y0 == 0 && y1 == 0) { y1 = s.desc.height; }
    ST_REQUIRE(y0 < y1 && y1 <= s.desc.height, "bad row window");
    s.row0 = y0; s.row1 = y1;
    return ST_OK;
}

int st_camera_set_output_format(StEngine* e, StHandle h, int format) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    ST_REQUIRE(format >= ST_FORMAT_RGBA32F && format <= ST_FORMAT_BGRA8_UNORM_SRGB, "unknown output format");
    it->second->out_format = (uint32_t)format;
    return ST_OK;
}

int st_tick(StEngine* e, void* stream) { ST_REQUIRE(e, "null engine"); return E(e)->tick(static_cast<hipStream_t>(stream)); }
int st_render_camera(StEngine* e, StHandle h, void* out, void* stream) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->render(*it->second, out, static_cast<hipStream_t>(stream));
}

int st_debug_keep_all_planes(StEngine* e, int keep) { ST_REQUIRE(e, "null engine"); E(e)->lean_frame = keep == 0; return ST_OK; }
int st_camera_present_copy(StEngine* e, StHandle h, const void* src_device, void* dst_host, size_t bytes, void* stream) {
    ST_REQUIRE(e && src_device && dst_host && bytes, "null argument");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->present_copy(*it->second, src_device, dst_host, bytes, static_cast<hipStream_t>(stream));
}
int st_camera_present_ready(StEngine* e, StHandle h, const void* dst_host, int wait, int* ready) {
    ST_REQUIRE(e && ready, "null argument");
    auto it = E(e)->cameras.find(h);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->present_ready(*it->second, dst_host, wait, ready);
}

int st_set_seed(StEngine* e, uint64_t seed) { ST_REQUIRE(e, "null engine"); E(e)->base_seed = seed; return ST_OK; }
int st_set_blue_noise(StEngine* e, const uint8_t* rgba, size_t bytes) {
    ST_REQUIRE(e && rgba && bytes == 256 * 256 * 4, "blue noise must be 256x256 RGBA8");
    E(e)->blue_noise.assign(rgba, rgba + bytes); E(e)->blue_noise_dirty = true;
    return ST_OK;
}
int st_debug_read_lut(StEngine* e, int what, float* out, size_t capacity_floats, size_t* written_floats) {
    ST_REQUIRE(e && what >= 0 && what < 3, "bad lut id");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine has no LUTs");
    const size_t n[3] = {256 * 64 * 4, 32 * 32 * 4, 256 * 256 * 4};
    const DeviceArray* src[3] = {&en->d_transmittance, &en->d_scattering, &en->d_sky};
    if (written_floats) *written_floats = n[what];
    if (!out) return ST_OK;
    ST_REQUIRE(capacity_floats >= n[what], "buffer too small");
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    ST_HIP(hipMemcpy(out, src[what]->ptr, n[what] * sizeof(float), hipMemcpyDeviceToHost));
    return ST_OK;
}

int st_camera_read_buffer(StEngine* e, StHandle h, int id, void* out, size_t capacity, size_t* written) {
    ST_REQUIRE(e && id >= 0 && id < ST_BUF_COUNT, "bad buffer id");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine has no camera buffers");
    CameraState& c = *it->second;
    if (written) *written = c.plane_bytes[id];
    if (!out) return ST_OK;
    ST_REQUIRE(capacity >= c.plane_bytes[id], "buffer too small");
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    const float4* src = (id == ST_BUF_GI_RESERVOIRS_1 && c.gi_aliased) ? c.plane[ST_BUF_GI_RESERVOIRS_0] : c.plane[id];
    ST_HIP(hipMemcpy(out, src, c.plane_bytes[id], hipMemcpyDeviceToHost));
    return ST_OK;
}
int st_camera_write_buffer(StEngine* e, StHandle h, int id, const void* data, size_t bytes) {
    ST_REQUIRE(e && data && id >= 0 && id < ST_BUF_COUNT, "bad buffer id");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine has no camera buffers");
    CameraState& c = *it->second;
    ST_REQUIRE(bytes == c.plane_bytes[id], "size does not match the buffer");
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    { const int rc = materialize_gi_history(c); if (rc) return rc; }
    ST_HIP(hipMemcpy(c.plane[id], data, bytes, hipMemcpyHostToDevice));
    if (id == ST_BUF_PRIM_SURFACE_MAP_A) c.surface_map_replaced[0] = true;
    if (id == ST_BUF_PRIM_SURFACE_MAP_B) c.surface_map_replaced[1] = true;
    return ST_OK;
}
int st_debug_set_pass_mask(StEngine* e, uint64_t mask) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    if (en->has_device) { ST_HIP(hipSetDevice(en->device)); for (auto& kv : en->cameras) { const int rc = materialize_gi_history(*kv.second); if (rc) return rc; } }
    en->pass_mask = mask;
    return ST_OK;
}
int st_debug_last_launches(StEngine* e, uint64_t* out_bits, size_t capacity, size_t* count) {
    ST_REQUIRE(e && count, "null argument");
    const std::vector<uint64_t>& v = E(e)->last_launches;
    *count = v.size();
    for (size_t i = 0; i < v.size() && i < capacity && out_bits; i++) out_bits[i] = v[i];
    return ST_OK;
}
int st_engine_set_arithmetic(StEngine* e, int arithmetic) {
    ST_REQUIRE(e, "null engine");
    ST_REQUIRE(arithmetic == ST_ARITH_FAST || arithmetic == ST_ARITH_EXACT, "unknown arithmetic");
    Engine* en = E(e);
    if (en->arithmetic == arithmetic) return ST_OK;
    en->arithmetic = arithmetic;
    en->L = arithmetic == ST_ARITH_EXACT ? launchers_exact() : launchers_fast();
    if (en->has_device) {  // frames in flight finish with the tables they were launched with; the byte tables follow the build
        ST_HIP(hipSetDevice(en->device));
        ST_HIP(hipDeviceSynchronize());
        en->L.launch_build_byte_luts(static_cast<float*>(en->d_byte_luts.ptr), nullptr);
        ST_HIP(hipDeviceSynchronize());
        // the atmosphere LUTs are regenerated by the next render with the new build's routines
        en->atmosphere_initialized = false; en->sky_known = false;
    }
    return ST_OK;
}
int st_engine_get_arithmetic(StEngine* e, int* out) { ST_REQUIRE(e && out, "null argument"); *out = E(e)->arithmetic; return ST_OK; }
int st_camera_ray_count(StEngine* e, StHandle h, uint64_t* out, int reset) {
    ST_REQUIRE(e && out, "null argument");
    Engine* en = E(e);
    auto it = en->cameras.find(h);
    if (it == en->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine");
    unsigned long long host[2 * KS_COUNT];
    ST_HIP(hipSetDevice(en->device));
    ST_HIP(hipDeviceSynchronize());
    { const int rc2 = read_counters(*it->second, host); if (rc2) return rc2; }
    uint64_t total = 0;
    for (int i = 0; i < KS_COUNT; i++) total += host[2 * i];
    *out = total;
    if (reset) { ST_HIP(hipMemset(it->second->counters, 0, kCounterBytes)); memset(it->second->profiled_traversal_bytes, 0, sizeof(it->second->profiled_traversal_bytes)); ST_HIP(hipDeviceSynchronize()); }
    return ST_OK;
}
int st_debug_read_scene(StEngine* e, int what, void* out, size_t capacity, size_t* written) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    const void* p; size_t bytes;
    if ((what == 0 || what == 4 || what == 5) && en->host_stream_stale) { en->refit_stream(); en->host_stream_stale = false; }  // device refits since the host copy was current
    switch (what) {
        case 0: p = en->bvh_stream.data(); bytes = en->bvh_stream.size() * sizeof(float4); break;
        case 1: p = en->triangles.data(); bytes = en->triangles.size() * sizeof(HostTriangle); break;
        case 2: p = en->gpu_lights.data(); bytes = en->gpu_lights.size() * sizeof(GpuLight); break;
        case 3: p = en->gpu_materials.data(); bytes = en->gpu_materials.size() * sizeof(GpuMaterial); break;
        case 4: en->expand_stream(); p = en->bvh_upload_.data(); bytes = (size_t)en->device_bvh_len * sizeof(float4); break;  // as st_tick would upload it now
        case 5: {  // the 4-wide nodes st_tick would append behind it when they are switched on (built here either way: a host-side test walks them)
            const bool was = en->wide_nodes; en->wide_nodes = true;
            en->expand_stream(); en->append_wide_nodes();
            en->wide_nodes = was;
            p = en->bvh_upload_.data() + en->device_bvh_len; bytes = (size_t)en->device_wide_len * sizeof(float4); break;
        }
        case 6: {  // the device stream as it is on the device right now (the live copy): what a device refit left there
            if (!en->has_device || !en->scene_uploaded) return fail(ST_ERR_NO_DEVICE, "no device copy of the scene");
            const size_t n = en->sets[en->live].bvh.capacity ? (size_t)en->live_bvh_texels : 0;
            en->readback_.resize(n);
            ST_HIP(hipSetDevice(en->device)); ST_HIP(hipDeviceSynchronize());
            if (n) ST_HIP(hipMemcpy(en->readback_.data(), en->sets[en->live].bvh.ptr, n * sizeof(float4), hipMemcpyDeviceToHost));
            p = en->readback_.data(); bytes = n * sizeof(float4); break;
        }
        case 7: case 8: case 9: case 10: case 11: case 12: case 13: {  // the device refit's inputs (k_bvh.hip), built here for a host-side emulation
            if (en->host_stream_stale) { en->refit_stream(); en->host_stream_stale = false; }
            en->expand_stream(); en->index_device_tree();
            en->readback_levels_.clear();
            for (const auto& l : en->refit_levels_) { en->readback_levels_.push_back(l.first); en->readback_levels_.push_back(l.second); }
            const std::vector<uint32_t>* v = what == 7 ? &en->parent_ : what == 8 ? &en->refit_local_ : what == 9 ? &en->refit_items_ : what == 10 ? &en->refit_batch_off_
                                           : what == 11 ? &en->readback_levels_ : &en->entry_of_tri_;
            if (what == 13) { p = en->tri_bounds.data(); bytes = en->tri_bounds.size() * sizeof(float4); }
            else { p = v->data(); bytes = v->size() * sizeof(uint32_t); }
            break;
        }
        default: return fail(ST_ERR_INVALID_ARGUMENT, "unknown scene buffer");
    }
    if (written) *written = bytes;
    if (!out) return ST_OK;
    ST_REQUIRE(capacity >= bytes, "buffer too small");
    if (bytes) memcpy(out, p, bytes);
    return ST_OK;
}
int st_debug_world(StEngine* e, uint32_t* light_count, uint32_t* next_frame) {
    ST_REQUIRE(e && light_count && next_frame, "null argument");
    *light_count = E(e)->light_count; *next_frame = E(e)->frame;
    return ST_OK;
}

int st_set_bvh_refresh(StEngine* e, int mode) {
    ST_REQUIRE(e, "null engine");
    ST_REQUIRE(mode == ST_BVH_REBUILD || mode == ST_BVH_REFIT || mode == ST_BVH_REFIT_DEVICE, "unknown refresh mode");
    Engine* en = E(e);
    if (en->bvh_refresh_mode != mode) { en->bvh_refresh_mode = mode; en->have_topology = false; }
    return ST_OK;
}
int st_debug_bvh_depth(StEngine* e, uint32_t* deepest_internal_chain, uint32_t* stack_entries) {
    ST_REQUIRE(e && deepest_internal_chain && stack_entries, "null argument");
    *deepest_internal_chain = E(e)->bvh_stack_need; *stack_entries = (uint32_t)kBvhStackSize;
    return ST_OK;
}
int st_debug_bvh_refits(StEngine* e, uint64_t* rebuilds, uint64_t* refits) {
    ST_REQUIRE(e && rebuilds && refits, "null argument");
    *rebuilds = E(e)->rebuilds; *refits = E(e)->refits;
    return ST_OK;
}
int st_debug_bvh_device_refits(StEngine* e, uint64_t* device_refits) {
    ST_REQUIRE(e && device_refits, "null argument");
    *device_refits = E(e)->device_refits;
    return ST_OK;
}
int st_debug_bvh_refresh(StEngine* e, uint64_t* primitives, uint64_t* reused) {
    ST_REQUIRE(e && primitives && reused, "null argument");
    *primitives = E(e)->bvh.prims.size(); *reused = E(e)->bvh.reused_primitives();
    return ST_OK;
}

int st_profile_enable(StEngine* e, int enabled) { ST_REQUIRE(e, "null engine"); E(e)->profiling = (enabled & 1) != 0; E(e)->count_bytes = (enabled & 2) != 0; E(e)->profile_group_atrous = (enabled & 4) != 0; E(e)->profile_kernel_events = (enabled & 8) != 0; return ST_OK; }
int st_profile_read(StEngine* e, StKernelProfile* out, size_t capacity, size_t* count, int reset) {
    ST_REQUIRE(e && out && count, "null argument");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "host-only engine");
    ST_HIP(hipSetDevice(en->device));
    const int rc = en->drain_profile();
    if (rc) return rc;
    // traversal bytes (the reference's used_memory, summed on the device) join the screen-space bytes per kernel
    ST_HIP(hipDeviceSynchronize());
    for (auto& kv : en->cameras) {
        CameraState& c = *kv.second;
        unsigned long long host[2 * KS_COUNT];
        { const int rc2 = read_counters(c, host); if (rc2) return rc2; }
        for (int i = 0; i < KS_COUNT; i++) {
            const unsigned long long total = host[2 * i + 1];
            if (total >= c.profiled_traversal_bytes[i]) {
                en->profile_totals[i].algorithmic_bytes += (double)(total - c.profiled_traversal_bytes[i]);
                en->profile_totals[i].traversal_bytes += (double)(total - c.profiled_traversal_bytes[i]);
            }
            c.profiled_traversal_bytes[i] = total;
        }
    }
    size_t n = 0;
    for (int i = 0; i < KS_COUNT && n < capacity; i++) {
        if (en->profile_totals[i].launches == 0 && en->profile_totals[i].traversal_bytes == 0.0) continue;  // bytes-only mode records no launches
        out[n] = en->profile_totals[i];
        n++;
    }
    *count = n;
    if (reset) en->reset_profile_totals();
    return ST_OK;
}

}  // extern "C"
