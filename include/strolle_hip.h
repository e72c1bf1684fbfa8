/* strolle_hip.h — C ABI of libstrolle_hip.so, the MI355X-native replacement for the
 * per-pixel hot path of Patryk27/strolle.
 *
 * Every entry point below replaces one method of the reference's `strolle::Engine<P>`
 * (reference paths relative to /root/reference). u64 handles stand in for the
 * `Params` associated handle types (strolle/src/lib.rs:402-409); POD structs stand in
 * for the Rust value types; `int` status codes stand in for the reference's
 * panics/asserts. See INTEGRATION.md for the Rust-side FFI stub a maintainer adds.
 *
 * Threading: like the reference (`&mut self` everywhere, lib.rs:105-395) an engine is
 * single-owner; calls on one engine must not overlap. Stream contract: everything
 * st_render_camera enqueues is complete once the `hipStream_t` it was given (0 = the null
 * stream) has drained — the engine may run part of a frame on an internal side stream, but
 * joins it into the caller's stream before the frame's last kernel. Frames may be enqueued
 * back to back without host synchronisation. st_tick queues its uploads on the stream it was given (from
 * page-locked copies, so the scene may be edited again as soon as it returns); st_render_camera orders itself behind them.
 * Scheduling and tuning switches: StTuning below (st_engine_get_tuning / st_engine_set_tuning; environment variables of the
 * same meaning override the defaults when an engine is created).
 */
#ifndef STROLLE_HIP_H
#define STROLLE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct StEngine StEngine;
typedef uint64_t StHandle;

enum StStatus {
    ST_OK = 0,
    ST_ERR_INVALID_ARGUMENT = 1,
    ST_ERR_NO_DEVICE = 2,        /* engine was created host-only, or HIP is unavailable */
    ST_ERR_UNKNOWN_CAMERA = 3,   /* reference: panic "camera does not exist" (camera_controllers.rs:21-34) */
    ST_ERR_EMPTY_MESH = 4,       /* reference: assert "contains no triangles" (triangles.rs:50-53) */
    ST_ERR_HIP = 5,              /* a HIP runtime call failed; see st_last_error() */
    ST_ERR_ATLAS_FULL = 6,       /* reference: warn + drop (images.rs:71-79) */
    ST_ERR_IO = 7,               /* scene ingest: a file could not be read */
    ST_ERR_PARSE = 8,            /* scene ingest: malformed glTF / GLB / PNG; st_last_error() says where */
    ST_ERR_UNSUPPORTED = 9,      /* scene ingest: valid file using something this loader does not read (JPEG, Draco, ...) */
    ST_ERR_BVH_TOO_DEEP = 10,    /* st_tick: the tree's deepest chain of internal nodes exceeds the deepest traversal stack the kernels take
                                  * (24 entries as strolle-gpu/src/lib.rs:76 — the reference indexes past its stack array beyond that —, grown
                                  * to the chain's own length for deeper trees, up to 32). The scene IS uploaded and
                                  * renders — pushes beyond the stack are dropped, so geometry behind them can be missed — but the tick
                                  * says so instead of returning ST_OK; StTuning::allow_deep_bvh = 1 turns the status back into a warning */
    ST_ERR_DIST = 11             /* st_dist_*: the collective transport failed (RCCL status in st_last_error()) */
};

/* strolle/src/mesh_triangle.rs:6-33 — object-space triangle */
typedef struct StMeshTriangle {
    float positions[3][3];
    float normals[3][3];
    float uvs[3][2];
    float tangents[3][4];
} StMeshTriangle;

/* strolle/src/material.rs:8-23; texture handles: 0 = None */
typedef struct StMaterial {
    float base_color[4];
    float emissive[4];
    float perceptual_roughness;
    float metallic;
    float reflectance;
    float ior;
    StHandle base_color_texture;
    StHandle emissive_texture;
    StHandle metallic_roughness_texture;
    StHandle normal_map_texture;
    uint32_t alpha_mode; /* 0 = Opaque, 1 = Blend (material.rs:72-91) */
    uint32_t _pad;
} StMaterial;

/* strolle/src/light.rs:6-22 */
enum StLightKind { ST_LIGHT_POINT = 0, ST_LIGHT_SPOT = 1 };
typedef struct StLight {
    uint32_t kind;
    float position[3];
    float radius;
    float color[3];
    float range;
    float direction[3]; /* spot only */
    float angle;        /* spot only */
} StLight;

/* strolle/src/camera.rs:8-14,83-105,170-175. Matrices are column-major (glam Mat4::to_cols_array). */
enum StCameraMode {
    ST_MODE_IMAGE = 0, ST_MODE_DI_DIFFUSE = 1, ST_MODE_DI_SPECULAR = 2, ST_MODE_GI_DIFFUSE = 3,
    ST_MODE_GI_SPECULAR = 4, ST_MODE_BVH_HEATMAP = 5, ST_MODE_REFERENCE = 6
};
typedef struct StCamera {
    uint32_t mode;     /* StCameraMode */
    uint32_t denoise;  /* CameraMode::{Image,Di*,Gi*}{denoise} */
    uint32_t depth;    /* CameraMode::Reference{depth} */
    uint32_t width, height;   /* viewport.size */
    uint32_t pos_x, pos_y;    /* viewport.position (kept for API parity; the output buffer is viewport-sized) */
    uint32_t _pad;
    float transform[16];
    float projection[16];
} StCamera;

/* ---- lifecycle: Engine::new (lib.rs:132-158).
 * device_ordinal >= 0: HIP device; -1: host-only engine (scene stores + BVH build work,
 * every call that needs the GPU returns ST_ERR_NO_DEVICE — never a CPU fallback). */
int st_engine_create(int device_ordinal, StEngine** out);
void st_engine_destroy(StEngine* e);
const char* st_last_error(void);
/* The commit libstrolle_hip.so was built from ("<hash>" or "<hash>+dirty"; "unknown" for a build outside a git tree): a property of the binary. */
const char* st_build_commit(void);

/* ---- scene: insert_xxx / remove_xxx (lib.rs:161-246) */
int st_mesh_insert(StEngine* e, StHandle id, const StMeshTriangle* triangles, size_t count);   /* lib.rs:161 */
int st_mesh_remove(StEngine* e, StHandle id);                                                    /* lib.rs:169 */
int st_material_insert(StEngine* e, StHandle id, const StMaterial* material);                   /* lib.rs:174 */
int st_material_has(StEngine* e, StHandle id);                                                   /* lib.rs:184; returns 0/1 */
int st_material_remove(StEngine* e, StHandle id);                                                /* lib.rs:192 */
int st_image_insert_rgba8(StEngine* e, StHandle id, uint32_t width, uint32_t height, const uint8_t* rgba, int srgb); /* lib.rs:198 */
int st_image_remove(StEngine* e, StHandle id);                                                   /* lib.rs:211 */
/* lib.rs:198 with ImageData::Texture{texture, is_dynamic} (image.rs:46-59): the pixels are RGBA8 in device memory,
 * rows `row_pitch_bytes` apart. Static images are copied into the atlas once, by the next st_tick; dynamic ones by every
 * st_tick (images.rs:187-213) on the stream st_tick is given — the caller keeps the buffer alive and orders its writes
 * before that tick. Needs a device engine. */
int st_image_insert_device_rgba8(StEngine* e, StHandle id, uint32_t width, uint32_t height, const void* device_rgba,
                                 size_t row_pitch_bytes, int is_dynamic);
/* xform: glam Affine3A as 12 floats, column-major (x_axis, y_axis, z_axis, translation) */
int st_instance_insert(StEngine* e, StHandle id, StHandle mesh, StHandle material, const float xform[12]); /* lib.rs:217 */
int st_instance_remove(StEngine* e, StHandle id);                                                /* lib.rs:226 */
int st_light_insert(StEngine* e, StHandle id, const StLight* light);                            /* lib.rs:232 */
int st_light_remove(StEngine* e, StHandle id);                                                   /* lib.rs:237 */
int st_sun_update(StEngine* e, float azimuth, float altitude);                                   /* lib.rs:242 */

/* ---- cameras (lib.rs:252-297) */
int st_camera_create(StEngine* e, const StCamera* camera, StHandle* out_handle);                /* lib.rs:252 */
int st_camera_update(StEngine* e, StHandle camera, const StCamera* desc);                       /* lib.rs:262 */
int st_camera_delete(StEngine* e, StHandle camera);                                              /* lib.rs:292 */

/* ---- per frame */
int st_tick(StEngine* e, void* hip_stream);                                                      /* lib.rs:301 */
/* Records and launches every pass of CameraController::render (camera_controller.rs:87-174) on
 * `hip_stream` and writes the composed HDR frame (RGBA32F, width*height*16 B, row-major) to the
 * DEVICE pointer `out_device` (width x height pixels of the camera's output format, RGBA32F unless
 * st_camera_set_output_format said otherwise; may be NULL to skip composition). Asynchronous. */
int st_render_camera(StEngine* e, StHandle camera, void* out_device, void* hip_stream); /* lib.rs:279 */

/* ---- present hand-over. Engine::render_camera (lib.rs:279-286) records the frame into the caller's wgpu encoder and
 * texture view; a facade over this library (rust/strolle-hip) owns no resource the other API can sample, so the composed
 * frame has to reach it through host memory. These two calls do that WITHOUT stalling the pipeline:
 *   st_camera_present_copy  enqueues `bytes` of `src_device` (the buffer st_render_camera just composed into on
 *                           `hip_stream`) -> `dst_host` on a copy stream the camera owns, ordered behind that frame only;
 *                           returns at once. The next frames' kernels overlap the copy (8 MB per 1080p RGBA8 frame).
 *   st_camera_present_ready *ready = 1 once the copy into `dst_host` has landed (wait != 0: block until then).
 * Use page-locked host memory (hipHostMalloc) — a pageable destination makes the copy synchronous — and alternate two
 * (src_device, dst_host) pairs: frame N-1 is presented while frame N renders (one frame of latency). Rendering into a
 * `src_device` whose copy is still in flight is safe: the engine orders that frame's composition behind the copy. */
int st_camera_present_copy(StEngine* e, StHandle camera, const void* src_device, void* dst_host, size_t bytes, void* hip_stream);
int st_camera_present_ready(StEngine* e, StHandle camera, const void* dst_host, int wait, int* ready);

/* ---- NEW seams (no counterpart in the reference) */
/* BVH refresh policy for scenes that change every frame (examples/stress-bvh.rs).
 * ST_BVH_REBUILD is the reference's behaviour: every change rebuilds the tree — with unchanged subtrees
 * reused, builder.rs:183-301 — and the result is the tree a from-scratch build gives. ST_BVH_REFIT keeps the tree
 * while instances only move (same triangles, same materials, same Blend flags) and recomputes its boxes bottom-up,
 * which costs a fraction of a rebuild; traversal stays correct, `used_memory` and the tree's quality follow the old
 * topology until something other than a transform changes (or the mode is set again), which rebuilds. */
enum StBvhRefresh { ST_BVH_REBUILD = 0, ST_BVH_REFIT = 1,
                    ST_BVH_REFIT_DEVICE = 2 /* as ST_BVH_REFIT, but the boxes are recomputed ON THE DEVICE: st_tick sends the moved triangles'
                                               hit-test records and bounds (80 B each) instead of refitting the stream on the host and
                                               re-sending all of it; k_bvh.hip patches the leaf entries and refits the boxes bottom-up (one launch per
                                               level of 512-leaf subtrees: two at 208 k triangles). Same bits as ST_BVH_REFIT. */,
                    ST_BVH_BUILD_DEVICE = 3 /* (round 5) after a scene change the tree is BUILT ON THE DEVICE, straight into the wide stream the fast
                                               build's rays walk (k_lbvh.hip: Morton codes, radix sort, the binary radix tree of Karras 2012, boxes by
                                               range queries, collapsed into 4-wide nodes): spawning or removing an instance costs a few hundred
                                               microseconds of device time instead of the host rebuild (26-28 ms at 208 k triangles). It is ANOTHER tree
                                               than the reference's binned SAH — the hits are the same, traversal costs about a fifth more for
                                               incoherent rays —, so it is used only while nothing observes the contract stream (a tick in which
                                               instances only MOVED refits that tree: st_debug_device_tree_refits): fast arithmetic, no
                                               BvhHeatmap camera, no byte counting. A tick that finds such an observer builds on the host as
                                               ST_BVH_REBUILD does; a heatmap camera created later renders after the next st_tick. */,
                    ST_BVH_AUTO = 4         /* (round 6) THE DEFAULT. The first tree of a scene is built on the host as ST_BVH_REBUILD builds it — the
                                               reference's binned SAH, paid once while the scene loads; every later change (spawn, despawn, move) is
                                               answered as ST_BVH_BUILD_DEVICE answers it, under the same conditions, so that a default engine no longer
                                               stalls for tens of milliseconds per spawn. The host's first tree is MEASURED (st_debug_auto_tree: the
                                               surface-area-weighted mean length of its leaf runs): above 3.4 — long runs of coplanar triangles on large
                                               faces, a step of the wide walk each — the device builder's tree is used from that very tick on: 5-17 %
                                               faster there; over 17 measured scene x mode rows of 13 k - 537 k triangles the default picks the faster
                                               tree, or one within 1 % of it, in 16 (profiles/r06_tree_choice_auto.txt).
                                               Scenes whose stream fits the kernels' LDS copy (at most 112
                                               entries: the Cornell box), host-only engines, the exact build and observed contract streams behave as
                                               under ST_BVH_REBUILD. */ };
/* Ticks whose tree was built on the device so far (ST_BVH_BUILD_DEVICE). */
int st_debug_device_builds(StEngine* e, uint64_t* ticks);
/* Ticks of that mode in which instances only moved and the device-built tree was REFITTED instead (same shape, every box recomputed: 5 launches
   against 22; at most 15 in a row, then the next change rebuilds; ST_NO_DEVICE_TREE_REFIT=1 in the environment rebuilds always). */
int st_debug_device_tree_refits(StEngine* e, uint64_t* ticks);
int st_set_bvh_refresh(StEngine* e, int mode);
int st_debug_bvh_refits(StEngine* e, uint64_t* rebuilds, uint64_t* refits);
int st_debug_bvh_device_refits(StEngine* e, uint64_t* device_refits);
/* Launches of the device bake so far and the triangles they baked (StTuning::device_bake: under ST_BVH_REFIT_DEVICE instances that only moved are
 * baked into world space on the device from object-space meshes uploaded once; a tick then sends 132 B per moved instance). */
int st_debug_device_bakes(StEngine* e, uint64_t* ticks, uint64_t* triangles);   /* ticks whose boxes were recomputed by k_bvh.hip (ST_BVH_REFIT_DEVICE) */
/* Depth check of the last BVH build: the longest chain of internal nodes (= the most far-child pointers one traversal can
 * have pending) against the per-ray stack the launches that walk this tree take: 24 entries (strolle-gpu/src/lib.rs:76) while that is
 * enough, the chain's own length for a deeper tree, up to 32 (dynamic LDS). The reference writes past its stack array when a tree is
 * deeper than 24; this library drops a push only beyond 32 and says so — a scene for which *deepest_internal_chain > *stack_entries can
 * miss geometry behind the dropped subtrees. */
int st_debug_bvh_depth(StEngine* e, uint32_t* deepest_internal_chain, uint32_t* stack_entries);
/* The WIDE stream (StTuning::wide_bvh) is another tree than the contract stream and its walks keep StTuning::wide_stack_entries (0 = 24) pending
 * entries per ray — every scene measured needs 11-14. A walk that does find its stack full drops the push (geometry behind it can be missed) and
 * sets a sticky word the engine owns; the next st_tick that sees it re-arms every later launch with a deeper stack (24 -> 32 -> 48 -> 56 entries;
 * the primary rays' packet walk, 64 entries in one register, is replaced by the per-lane walk) and returns ST_ERR_BVH_TOO_DEEP once
 * (StTuning::allow_deep_bvh = 1: a warning on stderr instead). *overflows: ticks that found a word set; *wide_stack_entries: what the wide walks
 * hold now; *packets_off: 1 once the packet walk overflowed. */
int st_debug_walk_overflow(StEngine* e, uint64_t* overflows, uint32_t* wide_stack_entries, uint32_t* packets_off);
/* What ST_BVH_AUTO's choice of a scene's first tree rests on. *leaf_run_weight: the surface-area-weighted mean length of the leaf runs of the host's
 * last binned-SAH build (1 = every leaf holds one triangle); *first_tree_on_device: 1 when that weight exceeded 3.4 at the scene's first tick and the
 * device builder's tree was used from that tick on (measured: profiles/r06_tree_choice*.txt). */
int st_debug_auto_tree(StEngine* e, float* leaf_run_weight, uint32_t* first_tree_on_device);

/* Deterministic seeds: every pass draws seed = pass_seed(base, frame, pass_id) instead of
 * rand::thread_rng() (camera_controller.rs:189-194; passes/ref_*.rs:49-59). */
int st_set_seed(StEngine* e, uint64_t base_seed);
/* Blue-noise texture (256x256 RGBA8), decoded by the caller from strolle/assets/blue-noise.png
 * (strolle/src/noise.rs:40-50 embeds the PNG; this library carries no image decoder). */
int st_set_blue_noise(StEngine* e, const uint8_t* rgba_256x256x4, size_t bytes);
/* Multi-GPU tiling: restrict every per-pixel launch of this camera to the window [x0, x1) x [y0, y1) of the full viewport
 * (x0 = x1 = 0: all columns; y0 = y1 = 0: all rows). Pixels keep their absolute coordinates — RNG, reprojection and every
 * neighbour tap are those of the full frame, so Reference / heatmap tiles reproduce the single-GPU image bit for bit; Image
 * mode reads neighbours, which is what the apron of st_dist_set_partition is for. x0 and x1 must be multiples of 16 (the
 * half-resolution passes work on 2x1 cells in tiles of 8) or the frame's right edge. st_camera_set_rows = all columns. */
int st_camera_set_window(StEngine* e, StHandle camera, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1);
int st_camera_set_rows(StEngine* e, StHandle camera, uint32_t y0, uint32_t y1);
/* ---- multi-GPU behind the boundary (NEW seam; SURVEY.md section 8e, BASELINE.json configs 4 and 5). One process per GPU, one
 * engine per process, the scene replicated; the frame is cut into tiles, every rank renders its tile (+ an apron of redundant
 * pixels in Image mode, whose passes read neighbours) with absolute pixel coordinates, and the ONE collective of the path
 * gathers the composed tiles to rank 0: grouped ncclSend / ncclRecv over RCCL — one point-to-point message per xGMI link into
 * the root, not a ring — on a communication stream the engine owns, ordered behind the frame by an event and overlapped with
 * the next frame. librccl is opened at run time (a process that already loaded it, e.g. through torch, shares that copy). */
typedef struct StDistRect { uint32_t x0, y0, x1, y1; } StDistRect;          /* [x0, x1) x [y0, y1) in pixels */
typedef struct StDistUniqueId { char internal[128]; } StDistUniqueId;      /* ncclUniqueId */
/* The partition rule (a pure function; `e` is not needed): `world` tiles in a grid of `cols` columns (0: the default grid — 1x1,
 * two row bands, 2x2, 3x2, 4x2, ... columns >= rows; a prime world gives row bands) whose edges sit on multiples of 16 pixels in
 * x and 8 in y; rank r owns tile (r % cols, r / cols). */
int st_dist_partition(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, uint32_t rank, StDistRect* owned);
/* `owned` widened by `apron` pixels on every side that has a neighbour, outward to the same 16 / 8 grid: what a rank renders. */
int st_dist_window(uint32_t width, uint32_t height, const StDistRect* owned, uint32_t apron, StDistRect* window);
/* Cost-weighted tiles (round 5). The equal split gives BASELINE config 5's eight tiles unequal work (one GPU rendering each tile window in
 * turn: max / mean 1.10, profiles/r05_tile_balance.json), so a grid's row edges and — per row — its column edges can be moved. A StDistGrid is
 * plain data that every rank holds identically: rank r owns column r % cols of row r / cols. st_dist_grid: the equal split (the tiles
 * st_dist_partition returns). st_dist_grid_rebalance: from the current grid and ONE cost per tile in rank order (a rank's frame time: the host
 * gathers them its own way), the grid whose rows — then each row's tiles — would cost the same if a tile's cost were spread evenly over its
 * pixels; edges stay on the 16 x 8 pixel grid, tiles stay at least 64 x 32 pixels, and with max_step != 0 no edge moves further than that per
 * call (max_step <= apron: every pixel a rank newly owns was already rendered by it, as apron, so its history is warm). The tiles of a valid
 * grid are disjoint and cover the frame. st_dist_set_grid: like st_dist_set_partition, with the grid's tiles; every rank must set the same
 * grid between the same two frames (the root sizes its receives from it). */
#define ST_DIST_MAX_SIDE 16
typedef struct StDistGrid {
    uint32_t cols, rows;
    uint32_t row_edge[ST_DIST_MAX_SIDE + 1];                    /* rows + 1 values, 0 ... height, multiples of 8 */
    uint32_t col_edge[ST_DIST_MAX_SIDE][ST_DIST_MAX_SIDE + 1];  /* per row: cols + 1 values, 0 ... width, multiples of 16 */
} StDistGrid;
int st_dist_grid(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, StDistGrid* out);
int st_dist_grid_tile(const StDistGrid* grid, uint32_t rank, StDistRect* owned);
int st_dist_grid_rebalance(uint32_t width, uint32_t height, const StDistGrid* current, const float* tile_cost, uint32_t max_step, StDistGrid* out);
/* RCCL transport: rank 0 makes an id (ncclGetUniqueId) and hands it to the other processes by its own means (the Rust host: a
 * pipe or MPI; bench.py: torch.distributed's store); every rank then joins with st_dist_init on its engine's device. */
int st_dist_unique_id(StDistUniqueId* out);
int st_dist_init(StEngine* e, int rank, int world, const StDistUniqueId* id);
/* In-process transport (tests; a single-GPU box): the engines of one process that share `group` exchange tiles through a
 * mailbox — same partition, pack / unpack and stream ordering, no RCCL. Works on host-only engines too (host frames). Within a
 * frame the non-root ranks call st_dist_gather before rank 0 does. */
int st_dist_init_local(StEngine* e, int rank, int world, uint64_t group);
int st_dist_shutdown(StEngine* e);
int st_dist_rank(StEngine* e, int* rank, int* world);
/* Sets the camera's window (st_camera_set_window) to this rank's tile + apron; reports both rectangles (either may be NULL). */
int st_dist_set_partition(StEngine* e, StHandle camera, uint32_t cols, uint32_t apron, StDistRect* owned, StDistRect* window);
int st_dist_set_grid(StEngine* e, StHandle camera, const StDistGrid* grid, uint32_t apron, StDistRect* owned, StDistRect* window);
/* `frame`: the buffer st_render_camera just composed into on `hip_stream` (full-frame sized, the camera's output format; this
 * rank's tile of it is what travels). `full_on_root`: where rank 0 assembles the frame (may be `frame` itself: its own tile is
 * then already in place); ignored on other ranks. Returns at once; the caller alternates two frame buffers so that frame N is
 * gathered while frame N+1 renders; st_render_camera into a buffer whose gather is still in flight is ordered behind that gather by the
 * engine. st_dist_wait orders `hip_stream` behind the gather that read `frame` (NULL: every gather in flight); host_wait != 0 blocks
 * the caller instead. st_dist_gather_ms: duration of the camera's last gather on the communication stream (blocks until it is through). */
int st_dist_gather(StEngine* e, StHandle camera, const void* frame, void* full_on_root, void* hip_stream);
int st_dist_wait(StEngine* e, StHandle camera, const void* frame, void* hip_stream, int host_wait);
int st_dist_gather_ms(StEngine* e, StHandle camera, float* ms);

/* camera.rs:170-175 `viewport.format`: the reference renders into a texture view of that format and the hardware converts
 * on store; here the composition kernel does. RGBA32F (default, 16 B/pixel), RGBA16F (8 B, round to nearest even),
 * RGBA8 / BGRA8 sRGB (4 B: clamp, IEC 61966-2-1 encode, round to nearest; alpha 255). The buffer handed to
 * st_render_camera must hold width x height pixels of the chosen format. */
enum StOutputFormat { ST_FORMAT_RGBA32F = 0, ST_FORMAT_RGBA16F = 1, ST_FORMAT_RGBA8_UNORM_SRGB = 2, ST_FORMAT_BGRA8_UNORM_SRGB = 3 };
int st_camera_set_output_format(StEngine* e, StHandle camera, int format);

/* ---- scheduling / tuning switches of one engine (NEW seam). Every field selects another launch structure or host policy for
 * the SAME pass graph: in the exact build every combination renders the same bits (tests/test_gpu_parity.py runs several
 * against the oracle). Defaults are what bench.py times. Get, change, set — between frames; `struct_size` must be
 * sizeof(StTuning). Environment variables (read once, when the engine is created) override the defaults:
 *   ST_NO_OVERLAP ST_NO_FUSE ST_NO_FUSE_DI_HEAD ST_NO_FUSE_SPATIAL ST_NO_FUSE_GI_SAMPLING ST_NO_FUSE_GI_VALIDATION
 *   ST_NO_FUSE_GI_REPROJECTION ST_NO_FUSE_WAVELET ST_NO_FUSE_COMPOSE ST_NO_PREVIEW_BOTH ST_NO_VARIANCE_IN_REPROJECT
 *   ST_KEEP_ALL_PLANES ST_KEEP_SCRATCH ST_NO_GI_ALIAS ST_NO_STAGING ST_NO_DOUBLE_BUFFER ST_NO_PACKED_BASE
 *   ST_NO_ANYHIT_FAST ST_NO_COMPACT_BVH (=1 clears the field), ST_ALLOW_DEEP_BVH ST_DI_HEAD_ON_MAIN ST_TILE_MAP ST_TILE_MAP_DENOISE ST_SIDE_PRIORITY
 *   ST_TICK_TIMING ST_DEVICE_BAKE (= value). */
typedef struct StTuning {
    uint32_t struct_size;
    uint32_t overlap;               /* 1: two HIP streams per camera, software-pipelined across frames */
    uint32_t fuse;                  /* 1: own-pixel consumer passes ride in their producer's launch (0: one launch per reference pass) */
    uint32_t fuse_di_head;          /* DI sampling + temporal resampling in one launch */
    uint32_t fuse_spatial;          /* DI / GI spatial resampling: pick + trace + sample per 2x1 cell in one launch */
    uint32_t fuse_gi_sampling;      /* GI sampling passes a + b in one launch */
    uint32_t fuse_gi_validation;    /* validation frames: gi_reprojection done by its two readers */
    uint32_t fuse_gi_reprojection;  /* tracing frames: gi_reprojection inside gi_temporal */
    uint32_t fuse_wavelet;          /* a-trous strides 1 and 2 as one launch */
    uint32_t fuse_compose;          /* fast build: frame composition inside the last a-trous pass */
    uint32_t preview_both;          /* both GI preview passes + resolving in one launch, flagged pixels served afterwards */
    uint32_t variance_in_reproject; /* estimate_variance's long-history branch inside the reproject stages */
    uint32_t lean_frame;            /* fast build: planes nothing reads again are not stored (st_debug_keep_all_planes) */
    uint32_t skip_scratch_stores;   /* fused DI spatial launch keeps its scratch records in registers */
    uint32_t di_head_on_main;       /* DI sampling + temporal on the caller's stream (0: on the side stream) */
    uint32_t alias_gi_history;      /* fast build: GI history hand-over by pointer swap instead of gi_resolving's copy */
    uint32_t tile_map;              /* blockIdx -> 8x8 tile mapping of the ReSTIR passes: 0 XCD bands, 1 hardware order, 2 chunks of 4 tile rows */
    uint32_t tile_map_denoise;      /* the same for the SVGF passes */
    int32_t side_priority;          /* > 0: the side stream gets the device's highest stream priority, < 0 the lowest */
    uint32_t staging;               /* st_tick uploads through page-locked staging slots (0: from pageable memory, joining the stream) */
    uint32_t double_buffer;         /* scene / light arrays exist twice on the device; a change fills the copy no frame in flight reads */
    uint32_t packed_base;           /* per-material packed base colour (0: primary visibility packs it per pixel) */
    uint32_t tick_timing;           /* 1: host-side cost of a scene refresh on stderr */
    uint32_t anyhit_fast;           /* fast build: shadow rays (boolean result only, ray.rs:84-112) walk with fast arithmetic
                                     * (st_device.h any_hit_fast); 0: the contract loop. Always 0 while traversal bytes are counted */
    uint32_t compact_bvh;           /* fast build: shadow rays walk a second, compact form of the BVH stream (48-B entries with conservative f16 child
                                     * boxes, regenerated on the device after every change: k_bvh.hip k_bvh_compact) — 2 / 3 texels per step instead of 4 */
    uint32_t allow_deep_bvh;        /* 1: a tree deeper than the traversal stack is a warning on stderr, not ST_ERR_BVH_TOO_DEEP */
    uint32_t device_bake;           /* 1: instances are baked into world space ON THE DEVICE from object-space meshes uploaded once
                                     * (k_bvh.hip k_bvh_bake) when only transforms changed under ST_BVH_REFIT_DEVICE; 0: on the host */
    uint32_t wide_bvh;              /* fast build, scenes that do not fit LDS: every ray outside the heatmap pass walks a 4-WIDE form of the BVH — four conservative
                                     * f16 child boxes + four links per aligned 64-B line (k_bvh.hip k_bvh_wide; the host collapses the binary tree once per build,
                                     * the device refills the boxes after every change) — half the dependent round trips and lines of the compact binary stream */
    uint32_t wide_stack_entries;    /* pending entries per ray of the wide walk's stack: 0 = 24 (strolle-gpu/src/lib.rs:76); tests render with 48 to show that 24 drops no push */
    uint32_t primary_packets;       /* with the wide stream: primary visibility walks it as ONE packet per wave — uniform node pointer and stack, scalar node
                                     * fetches, per-lane box and triangle tests, ballots decide the descent (st_device.h closest_hit_packet) */
} StTuning;
int st_engine_get_tuning(StEngine* e, StTuning* out);
int st_engine_set_tuning(StEngine* e, const StTuning* tuning);

/* Arithmetic of the per-pixel kernels. Both builds of every kernel live in the library (csrc/Makefile):
 * ST_ARITH_FAST (default) uses the hardware's reciprocal / square root / exp2 / log2 / sin / cos (1 ulp each) and lets the
 * compiler contract a*b+c into FMAs, everywhere except the ray-generation + BVH-traversal compare chain (Camera::ray,
 * Ray::traverse, intersect_box, Triangle::hit: strolle-gpu/src/camera.rs:32-51, ray.rs:114-302, triangle.rs:64-113),
 * which stays IEEE-exact so that BVH-heatmap integers are bit-identical to the reference restatement. The reference itself
 * leaves these functions to the SPIR-V driver's precision. Output is checked against the oracle within the per-plane
 * tolerances stated in tests/test_gpu_fast_tolerance.py.
 * ST_ARITH_EXACT evaluates everything with correctly rounded + - * / sqrt, no contraction and fixed polynomial
 * transcendentals: every buffer of every pass is then bit-identical to the CPU oracle (tests/test_gpu_parity.py).
 * ST_EXACT=1 in the environment makes new engines start in the exact build. Switching keeps all camera state. */
enum StArithmetic { ST_ARITH_FAST = 0, ST_ARITH_EXACT = 1 };
int st_engine_set_arithmetic(StEngine* e, int arithmetic);
int st_engine_get_arithmetic(StEngine* e, int* out);

/* ---- parity / measurement read-back */
enum StBufferId {
    ST_BUF_PRIM_GBUFFER_D0_A = 0, ST_BUF_PRIM_GBUFFER_D0_B = 1, ST_BUF_PRIM_GBUFFER_D1_A = 2, ST_BUF_PRIM_GBUFFER_D1_B = 3,
    ST_BUF_PRIM_SURFACE_MAP_A = 4, ST_BUF_PRIM_SURFACE_MAP_B = 5, ST_BUF_REPROJECTION_MAP = 6, ST_BUF_VELOCITY_MAP = 7,
    ST_BUF_DI_RESERVOIRS_0 = 8, ST_BUF_DI_RESERVOIRS_1 = 9, ST_BUF_DI_RESERVOIRS_2 = 10,
    ST_BUF_DI_DIFF_SAMPLES = 11, ST_BUF_DI_DIFF_PREV_COLORS = 12, ST_BUF_DI_DIFF_CURR_COLORS = 13,
    ST_BUF_DI_DIFF_MOMENTS_A = 14, ST_BUF_DI_DIFF_MOMENTS_B = 15, ST_BUF_DI_DIFF_STASH = 16, ST_BUF_DI_SPEC_SAMPLES = 17,
    ST_BUF_GI_D0 = 18, ST_BUF_GI_D1 = 19, ST_BUF_GI_D2 = 20,
    ST_BUF_GI_RESERVOIRS_0 = 21, ST_BUF_GI_RESERVOIRS_1 = 22, ST_BUF_GI_RESERVOIRS_2 = 23, ST_BUF_GI_RESERVOIRS_3 = 24,
    ST_BUF_GI_DIFF_SAMPLES = 25, ST_BUF_GI_DIFF_PREV_COLORS = 26, ST_BUF_GI_DIFF_CURR_COLORS = 27,
    ST_BUF_GI_DIFF_MOMENTS_A = 28, ST_BUF_GI_DIFF_MOMENTS_B = 29, ST_BUF_GI_DIFF_STASH = 30, ST_BUF_GI_SPEC_SAMPLES = 31,
    ST_BUF_REF_HITS = 32, ST_BUF_REF_RAYS = 33, ST_BUF_REF_COLORS = 34,
    ST_BUF_DBG_USED_MEMORY = 35, /* u32 per pixel: Ray::traverse's `used_memory` of the heatmap pass (ray.rs:125-264) */
    ST_BUF_COUNT = 36
};
/* Synchronises the device, then copies a per-camera buffer (camera_controller/buffers.rs:7-51) to host
 * memory. out == NULL: only report the size in *written.
 * After a frame every buffer holds what the reference's pass graph leaves in it — bit for bit in the exact build, within
 * the documented tolerance in the fast build, with one stated difference there: on frames whose GI source is the temporal
 * pass's output the history plane GI_RESERVOIRS_0 is that output itself (a pointer swap) instead of the reference's
 * decoded-and-re-encoded copy of it, so a few normals per frame differ by an ulp between the two planes' read-backs. */
int st_camera_read_buffer(StEngine* e, StHandle camera, int buffer_id, void* out, size_t capacity, size_t* written);
/* The inverse of st_camera_read_buffer: overwrite a per-camera buffer with host data (`bytes` must be the buffer's size).
 * With st_debug_set_pass_mask this lets a test hand one launch exactly the inputs the oracle's pass saw, so the fast
 * build is compared pass by pass instead of through ReSTIR's chaotic temporal feedback. Planes this library derives from
 * the reference's (decoded surface twins, sqrt-luma planes) are regenerated before the next render. Synchronises. */
int st_camera_write_buffer(StEngine* e, StHandle camera, int buffer_id, const void* data, size_t bytes);
/* The lean frame. In the fast build, when the whole pass graph of an Image{denoise} frame runs, planes that nothing reads
 * again — not a later pass of the frame, not the next frame — are not stored: the velocity map and the encoded surface map
 * (every kernel reads the decoded twin this library keeps), both diffuse-sample planes (the fused denoise-reproject stages
 * consume them in registers), the reprojected GI reservoirs of tracing frames, the first GI preview pass's result where it
 * is a plain normalisation of its input (the second pass rebuilds it where it reads one), and the last a-trous pass's colour
 * planes (frame composition runs inside that launch). st_camera_read_buffer of ST_BUF_VELOCITY_MAP, PRIM_SURFACE_MAP_*,
 * DI/GI_DIFF_SAMPLES, GI_RESERVOIRS_2, GI_RESERVOIRS_3 and DI/GI_DIFF_CURR_COLORS then returns what an earlier launch left.
 * keep != 0 (or ST_KEEP_ALL_PLANES=1 in the environment) makes every frame store all planes as the reference does; the exact
 * build, a pass mask and the partial camera modes always do. */
int st_debug_keep_all_planes(StEngine* e, int keep);
/* *stale = 1 when the camera's last frame left `buffer_id` unwritten (one of the planes listed above, in a lean frame): what
 * st_camera_read_buffer returns for it is an earlier launch's or frame's content. strolle_amd.api read_buffer(strict=True) refuses such a read. */
int st_camera_buffer_stale(StEngine* e, StHandle camera, int buffer_id, int* stale);
/* One bit per reference pass (strolle/src/camera_controller.rs:87-174 order). st_render_camera executes a launch only when
 * ALL the passes it covers are in the mask (a fused launch covers several); a mask that splits a fused launch is an
 * ST_ERR_INVALID_ARGUMENT. Default: all ones. Frame counters, seeds and plane ping-pong are unaffected. */
enum StPassBit {
    ST_PASS_PRIM_VISIBILITY = 1u << 0, ST_PASS_FRAME_REPROJECTION = 1u << 1,
    ST_PASS_DI_SAMPLING = 1u << 2, ST_PASS_DI_TEMPORAL = 1u << 3, ST_PASS_DI_SPATIAL_PICK = 1u << 4, ST_PASS_DI_SPATIAL_TRACE = 1u << 5,
    ST_PASS_DI_SPATIAL_SAMPLE = 1u << 6, ST_PASS_DI_RESOLVING = 1u << 7,
    ST_PASS_GI_REPROJECTION = 1u << 8, ST_PASS_GI_SAMPLING_A = 1u << 9, ST_PASS_GI_SAMPLING_B = 1u << 10, ST_PASS_GI_TEMPORAL = 1u << 11,
    ST_PASS_GI_SPATIAL_PICK = 1u << 12, ST_PASS_GI_SPATIAL_TRACE = 1u << 13, ST_PASS_GI_SPATIAL_SAMPLE = 1u << 14,
    ST_PASS_GI_PREVIEW_0 = 1u << 15, ST_PASS_GI_PREVIEW_1 = 1u << 16, ST_PASS_GI_RESOLVING = 1u << 17,
    ST_PASS_DENOISE_REPROJECT_DI = 1u << 18, ST_PASS_DENOISE_REPROJECT_GI = 1u << 19, ST_PASS_DENOISE_VARIANCE = 1u << 20,
    ST_PASS_DENOISE_WAVELET_0 = 1u << 21, /* ... wavelet pass n = ST_PASS_DENOISE_WAVELET_0 << n, n < 5 */
    ST_PASS_COMPOSITION = 1u << 26, ST_PASS_BVH_HEATMAP = 1u << 27, ST_PASS_REF_TRACING = 1u << 28, ST_PASS_REF_SHADING = 1u << 29
};
int st_debug_set_pass_mask(StEngine* e, uint64_t mask);
/* Measurement only (tools/pair_matrix.py): the frame's graph is built as always — every fusion of the whole frame — but only the launches
 * whose ordinal in the frame's serial order has its bit set are enqueued, all on the caller's stream. ~0 (default): everything, as shipped.
 * What the planes hold after a filtered frame is unspecified. */
int st_debug_set_launch_filter(StEngine* e, uint64_t filter);
/* The variance pass's short-history flags after the last frame (StTuning::variance_in_reproject): one 64-bit word per 8x8 tile, bit =
 * pixel of the tile — the pixels whose estimate_variance takes the 29-tap spatial branch (frame_denoising.rs:128-189). out == NULL: only
 * the tile count. (Steady state, 1080p: 8 % of the pixels on the Cornell box, 84 % in the dungeon — DI samples without confidence reset
 * their history every frame.) */
int st_debug_variance_flags(StEngine* e, StHandle camera, uint64_t* tile_mask_out, size_t capacity_tiles, size_t* tiles);
/* The pass bits of every launch the last st_render_camera considered, in launch order (executed or not): lets a test walk
 * the shipped launch structure without knowing it. Returns the count; writes at most `capacity` entries. */
int st_debug_last_launches(StEngine* e, uint64_t* out_bits, size_t capacity, size_t* count);
/* Rays traced for this camera since the last reset (device counters, closest-hit + any-hit). */
int st_camera_ray_count(StEngine* e, StHandle camera, uint64_t* out, int reset);
/* Host-side copies of what st_tick uploads: what = 0 BVH stream as the reference's serializer writes it (float4), 1
 * triangles in the reference's 144-B layout, 2 lights (112 B), 3 materials (112 B), 4 the BVH stream in its device form
 * (every entry four float4: internal nodes with the far child's byte offset, leaf entries followed by the triangle's
 * hit-test record; st_types.h), 6 the device form read back FROM the device (live copy),
 * 7-13 the inputs of the device refit (k_bvh.hip; uint32 unless noted): 7 parent of every entry (entry << 1 | child slot),
 * 8 LDS slot of every internal entry (bit 31: a task's root), 9 work items (bit 31: root of a finished task), 10 batch offsets
 * into 9, 11 (first batch, batches) per launch, 12 leaf entry of every triangle slot, 13 triangle bounds (two float4 per slot);
 * 14-17 the WIDE stream (StTuning::wide_bvh; uint32 unless noted): 14 its topology as the host builds it — one word with the root's
 * link in bits 0-7 and the most entries a walk over it can have pending in bits 8.., then 8 words per node: where each of the four child boxes lives in the device form (entry << 1 | 0 left box, 1 right box;
 * 0xffffffff = empty slot) and the four links (index << 1 | is a leaf record) —, 15 the device-form entry of every leaf record,
 * 16 / 17 its nodes (64 B each) and leaf records (48 B each) read back FROM the device (live copy; float4).
 * All but 6, 16 and 17 work on host-only engines. */
int st_debug_read_scene(StEngine* e, int what, void* out, size_t capacity, size_t* written);
int st_debug_world(StEngine* e, uint32_t* light_count, uint32_t* next_frame);
/* Where an image sits in the 8192-wide atlas: x, y, width, height in texels (images.rs:115-124 `lookup`). */
int st_debug_image_rect(StEngine* e, StHandle id, uint32_t out_xywh[4]);
/* The last BVH refresh (strolle/src/bvh/builder.rs:35-124 reuses subtrees whose primitives did not change): how many
 * primitives the tree holds and how many of them came over inside subtrees copied from the previous tree. The
 * uploaded stream is the one a from-scratch build of the same primitives gives, reuse or not. */
int st_debug_bvh_refresh(StEngine* e, uint64_t* primitives, uint64_t* reused);
/* Atmosphere LUTs as generated on the device (strolle-shaders/src/atmosphere): what = 0 transmittance 256x64,
 * 1 multi-scattering 32x32, 2 sky 256x256; RGBA32F texels holding f16-rounded values (the reference stores Rgba16Float). */
int st_debug_read_lut(StEngine* e, int what, float* out, size_t capacity_floats, size_t* written_floats);

/* ---- scene ingest. In the reference this step is Bevy's glTF loader plus bevy-strolle's
 * stages (bevy-strolle/src/stages/prepare.rs:20-122 meshes, :124-180 materials, :182-260 images; extract.rs instances);
 * here it is a convenience layered on the entry points above and nothing else. One mesh + instance per triangle-list
 * primitive of the default scene, numbered in depth-first node order: mesh / instance handle = first_handle + i,
 * material handle = first_handle + material index, image handle = first_image_handle + image index; KHR_lights_punctual
 * point and spot lights become lights the way bevy_gltf + bevy-strolle's extract stage would make them
 * (extract.rs:283-327), light handle = first_handle + k. Materials follow
 * prepare.rs:132-175 (Opaque forces alpha 1, Mask becomes Blend with alpha 0/1, reflectance 0.5, ior 1). PNG (all colour
 * types and bit depths, Adam7 too) and JPEG textures are decoded here; KTX2 / WebP, Draco and sparse accessors give
 * ST_ERR_UNSUPPORTED. Host-only work: valid on host-only engines. */
enum { ST_GLTF_OVERRIDE_REFLECTANCE = 1, ST_GLTF_OVERRIDE_PERCEPTUAL_ROUGHNESS = 2 };
typedef struct StGltfOptions {
    StHandle first_handle;        /* default 1 */
    StHandle first_image_handle;  /* default 1000 */
    uint32_t override_mask;       /* ST_GLTF_OVERRIDE_*: replace that field of every material (demo.rs:254-258 does this) */
    float reflectance;
    float perceptual_roughness;
    uint32_t subdivide;           /* k: every triangle is split into 4^k by midpoint subdivision (synthetic scaling), k <= 6 */
    float light_radius;           /* radius given to KHR_lights_punctual lights, which have none. 0 is what arrives through Bevy
                                   * (PointLight::radius defaults to 0) — but the reference's ReSTIR loses about half of a
                                   * zero-radius light's energy (reservoir/di.rs:105-116 tests `light.contains(point)`), which is
                                   * why its own examples set 0.15 (cornell.rs:45-54, demo.rs:169-191) */
    uint32_t _pad;
} StGltfOptions;
typedef struct StGltfSummary {
    uint32_t meshes, triangles, materials, images;
    uint32_t images_dropped;      /* did not fit the atlas: the reference warns and drops them (images.rs:71-79) */
    uint32_t primitives_skipped;  /* points, lines, strips, fans, or primitives without a single triangle */
    uint32_t lights;              /* KHR_lights_punctual point / spot lights inserted, handles first_handle + k in node order */
    uint32_t lights_skipped;      /* directional ones (strolle's sun is st_sun_update) and ones fainter than 0.0001 cd */
} StGltfSummary;
/* options == NULL: the defaults above; summary may be NULL. External buffers / images are read relative to the file. */
int st_scene_load_gltf(StEngine* e, const char* path, const StGltfOptions* options, StGltfSummary* summary);
int st_scene_load_gltf_memory(StEngine* e, const void* bytes, size_t size, const char* base_dir, const StGltfOptions* options, StGltfSummary* summary);
/* PNG -> RGBA8 (straight alpha; 16-bit samples keep their high byte), the decoder the loader uses. out_rgba == NULL
 * only reports the size. */
int st_decode_png(const void* bytes, size_t size, uint8_t* out_rgba, size_t capacity, uint32_t* width, uint32_t* height);
/* The same for either of glTF's two image formats, told apart by signature: PNG, or JPEG (baseline, extended and
 * progressive DCT; 8-bit; grey or three components). */
int st_decode_image(const void* bytes, size_t size, uint8_t* out_rgba, size_t capacity, uint32_t* width, uint32_t* height);

/* This device's streaming ceiling measured with the library's own grid-stride float4 copy kernel (k_util.hip): best of `iters`
 * copies of `bytes`, (bytes read + bytes written) / time in GB/s. bench.py reports it beside the 8 TB/s spec peak. */
int st_debug_copy_bandwidth(StEngine* e, size_t bytes, int iters, double* out_gbps);

/* Per-kernel measurement. st_profile_enable(e, flags): bit 0 (ST_PROFILE_TIMING) = HIP events around every launch on the
 * launch stream; while it is set the pass graph runs serially on the caller's stream (no two-stream overlap), so that an
 * event pair times its kernel alone. Bit 1 (ST_PROFILE_TRAVERSAL_BYTES) = the tracing kernels also sum the reference's
 * `used_memory` counter over their rays (a cross-lane reduction per ray: ~10 us per full-screen launch, which is why it
 * is not always on; rays themselves are always counted, st_camera_ray_count). The rendered bits are the same in every
 * mode. st_profile_read returns, per kernel slot i < *count: name, launches, total milliseconds (0 without bit 0),
 * algorithmic bytes (DESIGN.md "bytes per unit" x units launched; the traversal part is 0 without bit 1). */
enum { ST_PROFILE_TIMING = 1, ST_PROFILE_TRAVERSAL_BYTES = 2,
       ST_PROFILE_GROUP_ATROUS = 4 /* with TIMING: the a-trous chain's launches (4 per frame, back to back on one stream) are timed as ONE
                                      interval under the slot "a-trous chain (one timed interval)" instead of one event pair per slot */,
       ST_PROFILE_KERNEL_EVENTS = 8 /* with TIMING: every launch carries its own start / stop events (hipExtLaunchKernelGGL: the dispatch's
                                       timestamps, what rocprofv3's kernel trace reports) instead of events recorded between kernels */ };
enum { ST_PROFILE_MAX_KERNELS = 48 };  /* >= the number of kernel slots (st_kernels.h) */
typedef struct StKernelProfile {
    char name[48];
    uint32_t launches;
    float total_ms;
    double algorithmic_bytes; /* summed over launches: screen-space bytes (B) + traversal bytes (A), SURVEY.md 8(d) */
    double traversal_bytes;   /* the A part alone: the reference's used_memory counter summed over the kernel's rays */
} StKernelProfile;
int st_profile_enable(StEngine* e, int enabled);
int st_profile_read(StEngine* e, StKernelProfile* out, size_t capacity, size_t* count, int reset);

#ifdef __cplusplus
}
#endif
#endif /* STROLLE_HIP_H */
