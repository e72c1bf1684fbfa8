// st_dist.cpp — multi-GPU behind the C ABI (BASELINE.json configs 4 and 5): the tile partition of a frame
// over the ranks of one node, and the ONE collective of the path — the per-frame gather of the composed tiles to rank 0.
//
// One process per GPU, one engine per process, the scene replicated (every rank builds the same engine state). The reference has
// no counterpart (strolle renders on one device); north_star keeps the host in Rust, so the gather must be reachable through the
// C ABI and not only from Python (`strolle_amd/distributed.py` is the torch.distributed fallback for boxes without RCCL).
//
// Transports:
//   RCCL  — librccl is opened at run time (dlopen: the library stays loadable where RCCL is absent, and a process that already
//           loaded RCCL — torch does — shares that copy instead of bringing a second one). Grouped ncclSend / ncclRecv: every
//           peer sends its tile to rank 0 over its own xGMI link (point to point, not a ring), on a communication stream the
//           engine owns, ordered behind the frame by an event and overlapped with the next frame's rendering.
//   local — an in-process mailbox (several engines in ONE process, host-only or sharing one device): what the CPU tests and the
//           single-GPU box drive; same partition, same pack / unpack, same stream ordering, no RCCL.
#include <dlfcn.h>

#include <algorithm>
#include <cmath>
#include <mutex>

#include "st_engine.h"

namespace st {

// ------------------------------------------------------------------ partition (pure functions)
// Default grid of `world` tiles for a landscape frame: columns x rows with columns >= rows and both as square as the count allows
// (1: 1x1, 2: 1x2 — two row bands —, 4: 2x2, 8: 4x2, 6: 3x2, a prime p: 1 x p row bands).
static void default_grid(uint32_t world, uint32_t* cols, uint32_t* rows) {
    uint32_t r = 1;
    for (uint32_t k = 1; k * k <= world; k++) if (world % k == 0u) r = k;
    *rows = r; *cols = world / r;
    if (world == 2u) { *cols = 1u; *rows = 2u; }   // two row bands: no column seam, contiguous sends
}
static uint32_t split_edge(uint32_t extent, uint32_t parts, uint32_t i, uint32_t align) {
    if (i >= parts) return extent;
    uint64_t e = (uint64_t)extent * i / parts;
    e = (e + align / 2u) / align * align;
    return (uint32_t)std::min<uint64_t>(e, extent);
}
int dist_partition(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, uint32_t rank, StDistRect* owned) {
    if (!width || !height || !world || rank >= world || !owned) return fail(ST_ERR_INVALID_ARGUMENT, "bad partition request");
    uint32_t rows;
    if (cols == 0u) default_grid(world, &cols, &rows);
    else { if (world % cols != 0u) return fail(ST_ERR_INVALID_ARGUMENT, "world is not a multiple of the column count"); rows = world / cols; }
    const uint32_t cx = rank % cols, cy = rank / cols;
    owned->x0 = split_edge(width, cols, cx, 16u); owned->x1 = split_edge(width, cols, cx + 1u, 16u);
    owned->y0 = split_edge(height, rows, cy, 8u); owned->y1 = split_edge(height, rows, cy + 1u, 8u);
    if (owned->x0 >= owned->x1 || owned->y0 >= owned->y1) return fail(ST_ERR_INVALID_ARGUMENT, "the frame is too small for that many tiles");
    return ST_OK;
}
int dist_window(uint32_t width, uint32_t height, const StDistRect* owned, uint32_t apron, StDistRect* window) {
    if (!owned || !window || owned->x1 > width || owned->y1 > height) return fail(ST_ERR_INVALID_ARGUMENT, "bad window request");
    auto down = [](uint32_t v, uint32_t a) { return v / a * a; };
    auto up = [](uint32_t v, uint32_t a, uint32_t lim) { const uint64_t r = ((uint64_t)v + a - 1u) / a * a; return (uint32_t)std::min<uint64_t>(r, lim); };
    window->x0 = down(owned->x0 > apron ? owned->x0 - apron : 0u, 16u);
    window->y0 = down(owned->y0 > apron ? owned->y0 - apron : 0u, 8u);
    window->x1 = owned->x1 >= width ? width : up(std::min<uint64_t>((uint64_t)owned->x1 + apron, width), 16u, width);
    window->y1 = owned->y1 >= height ? height : up(std::min<uint64_t>((uint64_t)owned->y1 + apron, height), 8u, height);
    return ST_OK;
}

// ---- cost-weighted grids (round 5). The equal split above gives config 5's eight tiles unequal work (dungeon 3840x2160, one GPU rendering
// each tile window in turn: max / mean 1.10 — the upper-left tiles hold the vaults); a grid's row edges and, per row, its column edges can
// therefore be moved. A grid is plain data every rank holds identically: the host gathers each rank's frame time its own way (three floats
// per rank), every rank calls st_dist_grid_rebalance with the same numbers and gets the same grid.
int dist_grid(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, StDistGrid* g) {
    if (!g) return fail(ST_ERR_INVALID_ARGUMENT, "null grid");
    uint32_t rows;
    if (!width || !height || !world) return fail(ST_ERR_INVALID_ARGUMENT, "bad partition request");
    if (cols == 0u) default_grid(world, &cols, &rows);
    else { if (world % cols != 0u) return fail(ST_ERR_INVALID_ARGUMENT, "world is not a multiple of the column count"); rows = world / cols; }
    if (cols > ST_DIST_MAX_SIDE || rows > ST_DIST_MAX_SIDE) return fail(ST_ERR_INVALID_ARGUMENT, "a grid has at most 16 columns and 16 rows");
    memset(g, 0, sizeof(*g));
    g->cols = cols; g->rows = rows;
    for (uint32_t k = 0; k <= rows; k++) g->row_edge[k] = split_edge(height, rows, k, 8u);
    for (uint32_t k = 0; k < rows; k++) for (uint32_t c = 0; c <= cols; c++) g->col_edge[k][c] = split_edge(width, cols, c, 16u);
    for (uint32_t k = 0; k < rows; k++) if (g->row_edge[k] >= g->row_edge[k + 1]) return fail(ST_ERR_INVALID_ARGUMENT, "the frame is too small for that many tiles");
    for (uint32_t c = 0; c < cols; c++) if (g->col_edge[0][c] >= g->col_edge[0][c + 1]) return fail(ST_ERR_INVALID_ARGUMENT, "the frame is too small for that many tiles");
    return ST_OK;
}
// a grid is valid for a frame when its rows tile [0, height) and every row's columns tile [0, width), edges on the 8- / 16-pixel grid
static int grid_check(uint32_t width, uint32_t height, const StDistGrid* g) {
    if (!g || !g->cols || !g->rows || g->cols > ST_DIST_MAX_SIDE || g->rows > ST_DIST_MAX_SIDE) return fail(ST_ERR_INVALID_ARGUMENT, "bad grid");
    if (g->row_edge[0] != 0u || g->row_edge[g->rows] != height) return fail(ST_ERR_INVALID_ARGUMENT, "the grid's rows do not span the frame");
    for (uint32_t k = 0; k < g->rows; k++) {
        if (g->row_edge[k] >= g->row_edge[k + 1] || (g->row_edge[k] % 8u) != 0u) return fail(ST_ERR_INVALID_ARGUMENT, "grid rows must ascend on multiples of 8");
        if (g->col_edge[k][0] != 0u || g->col_edge[k][g->cols] != width) return fail(ST_ERR_INVALID_ARGUMENT, "a grid row's columns do not span the frame");
        for (uint32_t c = 0; c < g->cols; c++)
            if (g->col_edge[k][c] >= g->col_edge[k][c + 1] || (g->col_edge[k][c] % 16u) != 0u) return fail(ST_ERR_INVALID_ARGUMENT, "grid columns must ascend on multiples of 16");
    }
    return ST_OK;
}
int dist_grid_tile(const StDistGrid* g, uint32_t rank, StDistRect* owned) {
    if (!g || !owned || !g->cols || !g->rows || g->cols > ST_DIST_MAX_SIDE || g->rows > ST_DIST_MAX_SIDE || rank >= g->cols * g->rows) return fail(ST_ERR_INVALID_ARGUMENT, "bad grid tile request");
    const uint32_t cx = rank % g->cols, cy = rank / g->cols;
    owned->x0 = g->col_edge[cy][cx]; owned->x1 = g->col_edge[cy][cx + 1u]; owned->y0 = g->row_edge[cy]; owned->y1 = g->row_edge[cy + 1u];
    return ST_OK;
}
// Equal-cost edges of a piecewise-constant cost density: `bp` (n + 1 ascending breakpoints from 0 to extent) and `mass` (cost between
// consecutive breakpoints) -> `parts` + 1 edges, snapped to `align`, at least `min_size` apart, each within `max_step` of `old` (0: free).
static void equal_cost_edges(const std::vector<double>& bp, const std::vector<double>& mass, uint32_t extent, uint32_t parts, uint32_t align, uint32_t min_size,
                             uint32_t max_step, const uint32_t* old, uint32_t* out) {
    double total = 0.0; for (double m : mass) total += m;
    out[0] = 0u; out[parts] = extent;
    for (uint32_t j = 1; j < parts; j++) {
        double want = total * j / parts, acc = 0.0, pos = (double)extent * j / parts;
        if (total > 0.0)
            for (size_t i = 0; i < mass.size(); i++) {
                if (acc + mass[i] >= want) { pos = mass[i] > 0.0 ? bp[i] + (bp[i + 1] - bp[i]) * (want - acc) / mass[i] : bp[i]; break; }
                acc += mass[i];
            }
        int64_t e = (int64_t)((pos + align / 2.0) / align) * align;
        if (max_step) {   // an edge moves by at most max_step pixels (rounded down to the grid, at least one grid step) per call
            const int64_t step = std::max<int64_t>(align, (int64_t)max_step / align * align);
            e = std::min<int64_t>(std::max<int64_t>(e, (int64_t)old[j] - step), (int64_t)old[j] + step);
        }
        out[j] = (uint32_t)std::max<int64_t>(e, 0);
    }
    for (uint32_t j = 1; j < parts; j++) out[j] = std::max(out[j], out[j - 1] + min_size);                 // ascending, min_size apart ...
    // ... from both ends — and still ON THE GRID: the frame's far edge need not be a multiple of `align` (1080 rows, a 1000-pixel-wide frame), so the
    // back-to-front clamp rounds down (out[parts] - min_size landed off the 8 / 16 grid and grid_check refused the result)
    for (uint32_t j = parts; j-- > 1;) out[j] = std::min(out[j], (out[j + 1] - min_size) / align * align);
    // The clamps above may have carried an edge further than max_step from where it was, which would hand a rank pixels it never rendered (cold
    // history at the seam): when that happens, or the edges no longer ascend, the old edges stand for this call (they are a valid grid).
    bool keep_old = false;
    for (uint32_t j = 1; j < parts && old; j++) {
        const int64_t step = max_step ? std::max<int64_t>(align, (int64_t)max_step / align * align) : (int64_t)extent;
        if (std::llabs((int64_t)out[j] - (int64_t)old[j]) > step || out[j] <= out[j - 1] || out[j] >= out[j + 1]) keep_old = true;
    }
    if (keep_old) for (uint32_t j = 1; j < parts; j++) out[j] = old[j];
}
int dist_grid_rebalance(uint32_t width, uint32_t height, const StDistGrid* cur, const float* cost, uint32_t max_step, StDistGrid* out) {
    if (!cost || !out) return fail(ST_ERR_INVALID_ARGUMENT, "null argument");
    if (int rc = grid_check(width, height, cur)) return rc;
    const uint32_t rows = cur->rows, cols = cur->cols;
    for (uint32_t i = 0; i < rows * cols; i++) if (!(cost[i] > 0.0f) || !std::isfinite(cost[i])) return fail(ST_ERR_INVALID_ARGUMENT, "tile costs must be positive and finite");
    if (height < rows * 32u || width < cols * 64u) return fail(ST_ERR_INVALID_ARGUMENT, "the frame is too small to rebalance that many tiles");
    StDistGrid g = *cur;
    // rows first: a row's work is the sum of its tiles' (its columns are then balanced among themselves), spread evenly over its height
    std::vector<double> bp(rows + 1), mass(rows);
    for (uint32_t k = 0; k <= rows; k++) bp[k] = cur->row_edge[k];
    for (uint32_t k = 0; k < rows; k++) { mass[k] = 0.0; for (uint32_t c = 0; c < cols; c++) mass[k] += cost[k * cols + c]; }
    equal_cost_edges(bp, mass, height, rows, 8u, 32u, max_step, cur->row_edge, g.row_edge);
    // then each new row's columns, from the cost density of the old rows it overlaps (a tile's cost spread evenly over its area)
    for (uint32_t j = 0; j < rows; j++) {
        std::vector<uint32_t> cuts;
        for (uint32_t k = 0; k < rows; k++) for (uint32_t c = 0; c <= cols; c++) cuts.push_back(cur->col_edge[k][c]);
        std::sort(cuts.begin(), cuts.end()); cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
        std::vector<double> xb(cuts.begin(), cuts.end()), xm(cuts.size() - 1, 0.0);
        for (uint32_t k = 0; k < rows; k++) {
            const int64_t y0 = std::max<int64_t>(g.row_edge[j], cur->row_edge[k]), y1 = std::min<int64_t>(g.row_edge[j + 1], cur->row_edge[k + 1]);
            if (y1 <= y0) continue;
            const double share = (double)(y1 - y0) / (double)(cur->row_edge[k + 1] - cur->row_edge[k]);
            for (uint32_t c = 0; c < cols; c++) {
                const double x0 = cur->col_edge[k][c], x1 = cur->col_edge[k][c + 1];
                for (size_t i = 0; i + 1 < cuts.size(); i++) {
                    const double a = std::max<double>(x0, cuts[i]), b = std::min<double>(x1, cuts[i + 1]);
                    if (b > a) xm[i] += cost[k * cols + c] * share * (b - a) / (x1 - x0);
                }
            }
        }
        equal_cost_edges(xb, xm, width, cols, 16u, 64u, max_step, cur->col_edge[j], g.col_edge[j]);
    }
    if (int rc = grid_check(width, height, &g)) return rc;
    *out = g;
    return ST_OK;
}

// ------------------------------------------------------------------ RCCL through dlopen
namespace {
struct Rccl {
    void* lib = nullptr;
    int (*GetUniqueId)(void*) = nullptr;
    int (*CommInitRank)(void**, int, StDistUniqueId, int) = nullptr;   // ncclUniqueId is passed BY VALUE: a 128-byte struct
    int (*CommDestroy)(void*) = nullptr;
    int (*Send)(const void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*Recv)(void*, size_t, int, int, void*, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    std::string error;
    bool load() {
        if (lib) return true;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_NOLOAD);   // a copy the process already has (torch's) is shared, not doubled
            if (lib) break;
        }
        if (!lib) for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) { lib = dlopen(name, RTLD_NOW | RTLD_LOCAL); if (lib) break; }
        if (!lib) { error = std::string("librccl could not be opened: ") + (dlerror() ? dlerror() : "?"); return false; }
        auto sym = [&](const char* n) { void* p = dlsym(lib, n); if (!p) error = std::string("librccl lacks ") + n; return p; };
        GetUniqueId = reinterpret_cast<decltype(GetUniqueId)>(sym("ncclGetUniqueId"));
        CommInitRank = reinterpret_cast<decltype(CommInitRank)>(sym("ncclCommInitRank"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        return GetUniqueId && CommInitRank && CommDestroy && Send && Recv && GroupStart && GroupEnd && GetErrorString;
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
constexpr int kNcclInt8 = 0;   // rccl.h ncclDataType_t

// ---- the in-process transport: one mailbox slot per (group, sender); the sender copies its packed tile in and records an event, the
// root waits for the event on its stream and copies out. Non-root ranks of a frame must have called st_dist_gather before the root does.
// `consumed` orders the other direction: the root records it behind its copy-out, the sender's next hand-over waits for it before it
// overwrites the slot (a caller with two frames in flight would otherwise hand the root a torn or next-frame tile).
struct LocalSlot { void* mem = nullptr; size_t capacity = 0, bytes = 0; bool device = false; int device_id = -1; hipEvent_t ready = nullptr, consumed = nullptr;
                   bool consumed_recorded = false; uint64_t seq = 0, taken = 0; };
std::map<std::pair<uint64_t, int>, LocalSlot> g_local;
std::mutex g_local_mutex;
}  // namespace

struct DistState {
    int transport = 0;            // 0 none, 1 RCCL, 2 local
    int rank = 0, world = 1;
    void* comm = nullptr;         // ncclComm_t
    uint64_t group = 0;           // local transport: which mailbox
    hipStream_t stream = nullptr; // the communication stream
    // `slots`: the gathers in flight, by the frame buffer they read (the caller alternates two): a render into a buffer whose gather has
    // not finished is ordered behind it (Engine::dist_guard), st_dist_wait waits for one buffer's gather or for all.
    struct Slot { const void* frame = nullptr; hipEvent_t done = nullptr, t0 = nullptr; bool pending = false; };
    struct CamPart { StDistRect owned{}, window{}; uint32_t cols = 0, apron = 0, width = 0, height = 0;   // width x height: the frame the partition was computed for
                     bool has_grid = false; StDistGrid grid{};   // st_dist_set_grid: every rank's tile comes from this grid instead of the equal split
                     hipEvent_t rendered = nullptr; Slot slots[2]; uint32_t next = 0; int last = -1;
                     void* staging = nullptr; size_t staging_bytes = 0; };
    std::map<uint64_t, CamPart> cams;
};

static size_t bytes_per_pixel(uint32_t format) { return format == ST_FORMAT_RGBA32F ? 16u : (format == ST_FORMAT_RGBA16F ? 8u : 4u); }
static int rccl_fail(int code, const char* what) {
    return fail(ST_ERR_DIST, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(code) : "RCCL error") + " (" + std::to_string(code) + ")");
}

static void release_cam_part(Engine* en, DistState::CamPart& p) {
    if (p.rendered) (void)hipEventDestroy(p.rendered);
    for (auto& sl : p.slots) { if (sl.done) (void)hipEventDestroy(sl.done); if (sl.t0) (void)hipEventDestroy(sl.t0); }
    if (p.staging) { if (en->has_device) (void)hipFree(p.staging); else free(p.staging); }
    p = DistState::CamPart();
}
// st_camera_delete: the camera's partition, events and staging go with it
void Engine::dist_forget_camera(uint64_t handle) {
    if (!dist) return;
    auto it = dist->cams.find(handle);
    if (it == dist->cams.end()) return;
    if (has_device) { (void)hipSetDevice(device); if (dist->stream) (void)hipStreamSynchronize(dist->stream); }
    release_cam_part(this, it->second);
    dist->cams.erase(it);
}
void Engine::release_dist() {
    if (!dist) return;
    if (has_device) { (void)hipSetDevice(device); if (dist->stream) (void)hipStreamSynchronize(dist->stream); }
    for (auto& kv : dist->cams) release_cam_part(this, kv.second);
    if (dist->transport == 1 && dist->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(dist->comm);
    if (dist->transport == 2) {
        std::lock_guard<std::mutex> lock(g_local_mutex);
        auto it = g_local.find({dist->group, dist->rank});
        if (it != g_local.end()) {
            if (it->second.ready) (void)hipEventDestroy(it->second.ready);
            if (it->second.consumed) (void)hipEventDestroy(it->second.consumed);
            if (it->second.mem) { if (it->second.device) (void)hipFree(it->second.mem); else free(it->second.mem); }
            g_local.erase(it);
        }
    }
    if (dist->stream) (void)hipStreamDestroy(dist->stream);
    delete dist; dist = nullptr;
}

static int dist_begin(Engine* en, int rank, int world) {
    if (world < 1 || rank < 0 || rank >= world) return fail(ST_ERR_INVALID_ARGUMENT, "bad rank / world");
    en->release_dist();
    en->dist = new DistState();
    en->dist->rank = rank; en->dist->world = world;
    if (en->has_device) { ST_HIP(hipSetDevice(en->device)); ST_HIP(hipStreamCreateWithFlags(&en->dist->stream, hipStreamNonBlocking)); }
    return ST_OK;
}

// the tile `r` of this camera's partition, as every rank computes it
static int rect_of(const CameraState& c, const DistState& d, const DistState::CamPart& p, int r, StDistRect* out) {
    if (p.has_grid) return dist_grid_tile(&p.grid, (uint32_t)r, out);
    return dist_partition(c.desc.width, c.desc.height, (uint32_t)d.world, p.cols, (uint32_t)r, out);
}

int Engine::dist_set_partition(uint64_t handle, CameraState& c, uint32_t cols, uint32_t apron) {
    if (!dist) return fail(ST_ERR_INVALID_ARGUMENT, "st_dist_init has not been called on this engine");
    // validate first: a request that fails leaves neither a zero rectangle in the map nor the camera's window changed
    StDistRect owned{}, window{};
    if (int rc = dist_partition(c.desc.width, c.desc.height, (uint32_t)dist->world, cols, (uint32_t)dist->rank, &owned)) return rc;
    if (int rc = dist_window(c.desc.width, c.desc.height, &owned, apron, &window)) return rc;
    DistState::CamPart& p = dist->cams[handle];
    p.cols = cols; p.apron = apron; p.owned = owned; p.window = window; p.width = c.desc.width; p.height = c.desc.height; p.has_grid = false;
    c.col0 = p.window.x0; c.col1 = p.window.x1; c.row0 = p.window.y0; c.row1 = p.window.y1;
    return ST_OK;
}
// the same with a grid every rank holds (st_dist_grid / st_dist_grid_rebalance). A gather in flight still uses the tiles it was enqueued with
// (the byte counts were fixed then); the next st_dist_gather of every rank must follow the same st_dist_set_grid.
int Engine::dist_set_grid(uint64_t handle, CameraState& c, const StDistGrid& grid, uint32_t apron) {
    if (!dist) return fail(ST_ERR_INVALID_ARGUMENT, "st_dist_init has not been called on this engine");
    if (int rc = grid_check(c.desc.width, c.desc.height, &grid)) return rc;
    if (grid.cols * grid.rows != (uint32_t)dist->world) return fail(ST_ERR_INVALID_ARGUMENT, "the grid does not have one tile per rank");
    StDistRect owned{}, window{};
    if (int rc = dist_grid_tile(&grid, (uint32_t)dist->rank, &owned)) return rc;
    if (int rc = dist_window(c.desc.width, c.desc.height, &owned, apron, &window)) return rc;
    DistState::CamPart& p = dist->cams[handle];
    p.cols = grid.cols; p.apron = apron; p.owned = owned; p.window = window; p.width = c.desc.width; p.height = c.desc.height; p.has_grid = true; p.grid = grid;
    c.col0 = p.window.x0; c.col1 = p.window.x1; c.row0 = p.window.y0; c.row1 = p.window.y1;
    return ST_OK;
}

// pack / unpack on whichever side the engine lives
static void copy_rect(Engine* en, void* dst, size_t dst_pitch, const void* src, size_t src_pitch, size_t row_bytes, uint32_t rows, hipStream_t s) {
    if (en->has_device) { en->L.launch_rect_copy(dst, dst_pitch, src, src_pitch, row_bytes, rows, s); return; }
    for (uint32_t y = 0; y < rows; y++) memcpy(static_cast<char*>(dst) + (size_t)y * dst_pitch, static_cast<const char*>(src) + (size_t)y * src_pitch, row_bytes);
}
static int ensure_staging(Engine* en, DistState::CamPart& p, size_t bytes) {
    if (p.staging_bytes >= bytes) return ST_OK;
    if (p.staging) { if (en->has_device) (void)hipFree(p.staging); else free(p.staging); p.staging = nullptr; p.staging_bytes = 0; }
    if (en->has_device) ST_HIP(hipMalloc(&p.staging, bytes)); else { p.staging = malloc(bytes); if (!p.staging) return fail(ST_ERR_DIST, "out of memory"); }
    p.staging_bytes = bytes;
    return ST_OK;
}

int Engine::dist_gather(uint64_t handle, CameraState& c, const void* frame, void* full, hipStream_t stream) {
    if (!dist) return fail(ST_ERR_INVALID_ARGUMENT, "st_dist_init has not been called on this engine");
    auto it = dist->cams.find(handle);
    if (it == dist->cams.end()) return fail(ST_ERR_INVALID_ARGUMENT, "st_dist_set_partition has not been called for this camera");
    DistState& d = *dist; DistState::CamPart& p = it->second;
    const bool root = d.rank == 0;
    // The partition belongs to the frame size it was computed for. st_camera_update with another size rebuilds the camera's buffers and
    // resets its window to the whole frame (allocate_camera): peers would then send the OLD tile's byte count while the root expects the
    // NEW one's — a hung collective or a corrupt frame. Refuse until st_dist_set_partition has been called again.
    if (p.width != c.desc.width || p.height != c.desc.height || p.owned.x0 >= p.owned.x1 || p.owned.y0 >= p.owned.y1)
        return fail(ST_ERR_INVALID_ARGUMENT, "the camera's size changed since st_dist_set_partition: call it again before gathering");
    if (c.col0 > p.owned.x0 || c.col1 < p.owned.x1 || c.row0 > p.owned.y0 || c.row1 < p.owned.y1)
        return fail(ST_ERR_INVALID_ARGUMENT, "the camera's window no longer covers this rank's tile (st_camera_set_window since st_dist_set_partition)");
    if (root && !full) return fail(ST_ERR_INVALID_ARGUMENT, "rank 0 needs the destination frame");
    if (!frame) return fail(ST_ERR_INVALID_ARGUMENT, "null frame");
    const size_t bpp = bytes_per_pixel(c.out_format), pitch = (size_t)c.desc.width * bpp;
    hipStream_t cs = d.stream;
    DistState::Slot* slot = nullptr;
    if (has_device) {
        ST_HIP(hipSetDevice(device));
        if (!p.rendered) ST_HIP(hipEventCreateWithFlags(&p.rendered, hipEventDisableTiming));
        for (int k = 0; k < 2; k++) if (p.slots[k].frame == frame) { slot = &p.slots[k]; p.last = k; }
        if (!slot) { p.last = (int)(p.next & 1u); slot = &p.slots[p.last]; p.next++; }
        if (slot->pending) { ST_HIP(hipEventSynchronize(slot->done)); slot->pending = false; }   // only when the caller runs more than two frames ahead
        if (!slot->done) { ST_HIP(hipEventCreate(&slot->done)); ST_HIP(hipEventCreate(&slot->t0)); }   // timing enabled: st_dist_gather_ms
        slot->frame = frame;
        ST_HIP(hipEventRecord(p.rendered, stream));          // the frame is composed ...
        ST_HIP(hipStreamWaitEvent(cs, p.rendered, 0));       // ... before the communication stream touches it
        ST_HIP(hipEventRecord(slot->t0, cs));
    }
    // what this rank contributes: its own tile, contiguous when it spans the frame's width (a row band), packed otherwise
    auto tile_bytes = [&](const StDistRect& r) { return (size_t)(r.x1 - r.x0) * (r.y1 - r.y0) * bpp; };
    auto is_band = [&](const StDistRect& r) { return r.x0 == 0u && r.x1 == c.desc.width; };
    auto at = [&](const void* base, const StDistRect& r) { return static_cast<const char*>(base) + (size_t)r.y0 * pitch + (size_t)r.x0 * bpp; };
    if (d.world == 1) {
        if (root && full != frame) copy_rect(this, const_cast<char*>(at(full, p.owned)), pitch, at(frame, p.owned), pitch, (size_t)(p.owned.x1 - p.owned.x0) * bpp, p.owned.y1 - p.owned.y0, cs);
    } else if (!root) {
        const void* send = at(frame, p.owned);
        if (!is_band(p.owned)) {
            if (int rc = ensure_staging(this, p, tile_bytes(p.owned))) return rc;
            copy_rect(this, p.staging, (size_t)(p.owned.x1 - p.owned.x0) * bpp, at(frame, p.owned), pitch, (size_t)(p.owned.x1 - p.owned.x0) * bpp, p.owned.y1 - p.owned.y0, cs);
            send = p.staging;
        }
        if (d.transport == 1) {
            const int rc = g_rccl.Send(send, tile_bytes(p.owned), kNcclInt8, 0, d.comm, cs);
            if (rc) return rccl_fail(rc, "ncclSend");
        } else {
            std::lock_guard<std::mutex> lock(g_local_mutex);
            LocalSlot& slot = g_local[{d.group, d.rank}];
            const size_t n = tile_bytes(p.owned);
            if (slot.capacity < n || slot.device != has_device) {
                if (slot.mem) { if (slot.device) (void)hipFree(slot.mem); else free(slot.mem); slot.mem = nullptr; }   // hipFree joins the device: no read of it is left in flight
                slot.capacity = 0; slot.consumed_recorded = false;
                if (has_device) { ST_HIP(hipMalloc(&slot.mem, n)); } else { slot.mem = malloc(n); if (!slot.mem) return fail(ST_ERR_DIST, "out of memory"); }
                slot.capacity = n; slot.device = has_device; slot.device_id = device;
            }
            if (has_device) {
                if (!slot.ready) ST_HIP(hipEventCreateWithFlags(&slot.ready, hipEventDisableTiming));
                if (slot.consumed_recorded) ST_HIP(hipStreamWaitEvent(cs, slot.consumed, 0));   // the root's read of the previous tile comes first
                ST_HIP(hipMemcpyAsync(slot.mem, send, n, hipMemcpyDeviceToDevice, cs));
                ST_HIP(hipEventRecord(slot.ready, cs));
            } else memcpy(slot.mem, send, n);
            slot.bytes = n; slot.seq++;
        }
    } else {
        if (full != frame) copy_rect(this, const_cast<char*>(at(full, p.owned)), pitch, at(frame, p.owned), pitch, (size_t)(p.owned.x1 - p.owned.x0) * bpp, p.owned.y1 - p.owned.y0, cs);
        // receive every peer's tile: straight into the frame when it is a band, through the staging area and an unpack otherwise
        size_t need = 0;
        std::vector<StDistRect> rects((size_t)d.world);
        std::vector<size_t> offset((size_t)d.world, 0);
        for (int r = 1; r < d.world; r++) {
            if (int rc = rect_of(c, d, p, r, &rects[(size_t)r])) return rc;
            if (!is_band(rects[(size_t)r])) { offset[(size_t)r] = need; need += (tile_bytes(rects[(size_t)r]) + 255u) & ~(size_t)255u; }
        }
        if (need) if (int rc = ensure_staging(this, p, need)) return rc;
        if (d.transport == 1) {
            int rc = g_rccl.GroupStart();
            if (rc) return rccl_fail(rc, "ncclGroupStart");
            for (int r = 1; r < d.world && !rc; r++) {
                const StDistRect& q = rects[(size_t)r];
                void* dst = is_band(q) ? const_cast<char*>(at(full, q)) : static_cast<char*>(p.staging) + offset[(size_t)r];
                rc = g_rccl.Recv(dst, tile_bytes(q), kNcclInt8, r, d.comm, cs);
            }
            const int rc2 = g_rccl.GroupEnd();
            if (rc) return rccl_fail(rc, "ncclRecv");
            if (rc2) return rccl_fail(rc2, "ncclGroupEnd");
        } else {
            std::lock_guard<std::mutex> lock(g_local_mutex);
            for (int r = 1; r < d.world; r++) {
                const StDistRect& q = rects[(size_t)r];
                auto sit = g_local.find({d.group, r});
                if (sit == g_local.end() || sit->second.seq == sit->second.taken || sit->second.bytes != tile_bytes(q))
                    return fail(ST_ERR_DIST, "local transport: rank " + std::to_string(r) + " has not handed over this frame's tile (non-root ranks call st_dist_gather first)");
                LocalSlot& slot = sit->second;
                void* dst = is_band(q) ? const_cast<char*>(at(full, q)) : static_cast<char*>(p.staging) + offset[(size_t)r];
                if (has_device) {
                    if (slot.ready) ST_HIP(hipStreamWaitEvent(cs, slot.ready, 0));
                    ST_HIP(hipMemcpyAsync(dst, slot.mem, slot.bytes, slot.device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, cs));
                    if (slot.device) {
                        if (!slot.consumed) ST_HIP(hipEventCreateWithFlags(&slot.consumed, hipEventDisableTiming));
                        ST_HIP(hipEventRecord(slot.consumed, cs)); slot.consumed_recorded = true;
                    }
                } else memcpy(dst, slot.mem, slot.bytes);
                slot.taken = slot.seq;
            }
        }
        for (int r = 1; r < d.world; r++) {
            const StDistRect& q = rects[(size_t)r];
            if (!is_band(q)) copy_rect(this, const_cast<char*>(at(full, q)), pitch, static_cast<char*>(p.staging) + offset[(size_t)r], (size_t)(q.x1 - q.x0) * bpp, (size_t)(q.x1 - q.x0) * bpp, q.y1 - q.y0, cs);
        }
    }
    if (has_device) { ST_HIP(hipEventRecord(slot->done, cs)); slot->pending = true; }
    return ST_OK;
}

int Engine::dist_wait(uint64_t handle, const void* frame, hipStream_t stream, bool host) {
    if (!dist || !has_device) return ST_OK;
    auto it = dist->cams.find(handle);
    if (it == dist->cams.end()) return ST_OK;
    ST_HIP(hipSetDevice(device));
    for (auto& sl : it->second.slots) {
        if (!sl.pending || (frame && sl.frame != frame)) continue;
        if (host) { ST_HIP(hipEventSynchronize(sl.done)); sl.pending = false; }
        else ST_HIP(hipStreamWaitEvent(stream, sl.done, 0));
    }
    return ST_OK;
}
// a composition into a buffer whose gather has not finished waits for that gather (callers that alternate two buffers never wait long)
void Engine::dist_guard(uint64_t handle, const void* out, hipStream_t s) {
    if (!dist || !has_device || !out) return;
    auto it = dist->cams.find(handle);
    if (it == dist->cams.end()) return;
    for (auto& sl : it->second.slots) if (sl.pending && sl.frame == out) (void)hipStreamWaitEvent(s, sl.done, 0);
}
// duration of the camera's last gather on the communication stream (blocks until it has finished)
int Engine::dist_gather_ms(uint64_t handle, float* ms) {
    *ms = 0.0f;
    if (!dist || !has_device) return ST_OK;
    auto it = dist->cams.find(handle);
    if (it == dist->cams.end() || it->second.last < 0) return ST_OK;
    DistState::Slot& sl = it->second.slots[it->second.last];
    if (!sl.done) return ST_OK;
    ST_HIP(hipSetDevice(device));
    ST_HIP(hipEventSynchronize(sl.done)); sl.pending = false;
    ST_HIP(hipEventElapsedTime(ms, sl.t0, sl.done));
    return ST_OK;
}

}  // namespace st

using namespace st;
static Engine* E(StEngine* e) { return reinterpret_cast<Engine*>(e); }
#define ST_REQUIRE(cond, msg) do { if (!(cond)) return fail(ST_ERR_INVALID_ARGUMENT, msg); } while (0)

extern "C" {

int st_dist_partition(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, uint32_t rank, StDistRect* owned) { return dist_partition(width, height, world, cols, rank, owned); }
int st_dist_window(uint32_t width, uint32_t height, const StDistRect* owned, uint32_t apron, StDistRect* window) { return dist_window(width, height, owned, apron, window); }

int st_dist_unique_id(StDistUniqueId* out) {
    ST_REQUIRE(out, "null argument");
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    if (!g_rccl.load()) return fail(ST_ERR_DIST, g_rccl.error);
    const int rc = g_rccl.GetUniqueId(out);
    return rc ? rccl_fail(rc, "ncclGetUniqueId") : ST_OK;
}
int st_dist_init(StEngine* e, int rank, int world, const StDistUniqueId* id) {
    ST_REQUIRE(e && id, "null argument");
    Engine* en = E(e);
    if (!en->has_device) return fail(ST_ERR_NO_DEVICE, "the RCCL transport needs a device engine (st_dist_init_local is the in-process one)");
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        if (!g_rccl.load()) return fail(ST_ERR_DIST, g_rccl.error);
    }
    if (int rc = dist_begin(en, rank, world)) return rc;
    en->dist->transport = 1;
    const int rc = g_rccl.CommInitRank(&en->dist->comm, world, *id, rank);
    if (rc) { const int out = rccl_fail(rc, "ncclCommInitRank"); en->release_dist(); return out; }
    return ST_OK;
}
int st_dist_init_local(StEngine* e, int rank, int world, uint64_t group) {
    ST_REQUIRE(e, "null engine");
    Engine* en = E(e);
    if (int rc = dist_begin(en, rank, world)) return rc;
    en->dist->transport = 2; en->dist->group = group;
    return ST_OK;
}
int st_dist_shutdown(StEngine* e) { ST_REQUIRE(e, "null engine"); E(e)->release_dist(); return ST_OK; }
int st_dist_rank(StEngine* e, int* rank, int* world) {
    ST_REQUIRE(e && rank && world, "null argument");
    *rank = E(e)->dist ? E(e)->dist->rank : 0; *world = E(e)->dist ? E(e)->dist->world : 1;
    return ST_OK;
}
int st_dist_set_partition(StEngine* e, StHandle camera, uint32_t cols, uint32_t apron, StDistRect* owned, StDistRect* window) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(camera);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (int rc = E(e)->dist_set_partition(camera, *it->second, cols, apron)) return rc;
    const DistState::CamPart& p = E(e)->dist->cams[camera];
    if (owned) *owned = p.owned;
    if (window) *window = p.window;
    return ST_OK;
}
int st_dist_grid(uint32_t width, uint32_t height, uint32_t world, uint32_t cols, StDistGrid* out) { return dist_grid(width, height, world, cols, out); }
int st_dist_grid_tile(const StDistGrid* grid, uint32_t rank, StDistRect* owned) { return dist_grid_tile(grid, rank, owned); }
int st_dist_grid_rebalance(uint32_t width, uint32_t height, const StDistGrid* current, const float* tile_cost, uint32_t max_step, StDistGrid* out) {
    return dist_grid_rebalance(width, height, current, tile_cost, max_step, out);
}
int st_dist_set_grid(StEngine* e, StHandle camera, const StDistGrid* grid, uint32_t apron, StDistRect* owned, StDistRect* window) {
    ST_REQUIRE(e && grid, "null argument");
    auto it = E(e)->cameras.find(camera);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    if (int rc = E(e)->dist_set_grid(camera, *it->second, *grid, apron)) return rc;
    const DistState::CamPart& p = E(e)->dist->cams[camera];
    if (owned) *owned = p.owned;
    if (window) *window = p.window;
    return ST_OK;
}
int st_dist_gather(StEngine* e, StHandle camera, const void* frame, void* full_on_root, void* stream) {
    ST_REQUIRE(e, "null engine");
    auto it = E(e)->cameras.find(camera);
    if (it == E(e)->cameras.end()) return fail(ST_ERR_UNKNOWN_CAMERA, "camera does not exist");
    return E(e)->dist_gather(camera, *it->second, frame, full_on_root, static_cast<hipStream_t>(stream));
}
int st_dist_wait(StEngine* e, StHandle camera, const void* frame, void* stream, int host_wait) {
    ST_REQUIRE(e, "null engine");
    return E(e)->dist_wait(camera, frame, static_cast<hipStream_t>(stream), host_wait != 0);
}
int st_dist_gather_ms(StEngine* e, StHandle camera, float* ms) { ST_REQUIRE(e && ms, "null argument"); return E(e)->dist_gather_ms(camera, ms); }

}  // extern "C"
