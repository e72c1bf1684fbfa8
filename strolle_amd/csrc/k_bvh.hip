// k_bvh.hip — BVH refit on the device (st_set_bvh_refresh(ST_BVH_REFIT_DEVICE); SURVEY.md section 8(f).2, the step before the
// path for scenes whose instances move every frame: strolle/src/bvh/builder.rs:183-354, examples/stress-bvh.rs).
// While instances only move, the tree's topology and leaf entries stay what they are: the host bakes the moved triangles
// and sends their hit-test records and bounds (80 B per triangle) instead of refitting the stream itself and re-sending all
// of it (64 B per entry: 27 MB for the 208 k-triangle dungeon); these two kernels bring the device stream up to date.
// min / max are exact in any order, so the result is bit for bit the host's backward sweep (st_engine.cpp refit_stream)
// and the CPU restatement's refit (tests/test_gpu_parity.py test_bvh_refit_*).
#include <hip/hip_fp16.h>
#include "k_common.h"

namespace st {
namespace ST_KNS {

// the hit-test records (v0, v1 - v0, v2 - v0) of triangles [lo, hi) into their leaf entries (entry_of_tri: device entry number)
__global__ void k_bvh_patch_leaves(float4* bvh, const float4* tri_geo, const uint32_t* entry_of_tri, uint32_t lo, uint32_t hi) {
    const uint32_t t = lo + blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= hi) return;
    const uint32_t e = entry_of_tri[t];
    if (e == 0xffffffffu) return;  // a free triangle slot
    bvh[4u * e + 1u] = tri_geo[3u * t]; bvh[4u * e + 2u] = tri_geo[3u * t + 1u]; bvh[4u * e + 3u] = tri_geo[3u * t + 2u];
}
// Bottom-up box refit of one batch of tasks per workgroup (st_engine.cpp index_device_tree cuts the tree): a lane per item —
// a leaf run, whose box is computed from its triangles' bounds, or the root of a task an earlier launch finished, whose box is
// read back from the stream — stores the box in its parent's child slot (a node's box lives in its parent's entry; the tree root's
// is never stored) and in the parent's LDS slot, and climbs: the second lane to arrive at a node — an LDS counter per node tells —
// finds both child boxes in LDS, so the node's own box is their union, and carries on until it has stored a task root's box.
// Nothing is handed from workgroup to workgroup inside a launch: every stream store is read again only by later launches.
// parent[e] = (parent entry << 1) | which child slot, 0xffffffff for the root.
__global__ __launch_bounds__(256) void k_bvh_refit(float4* bvh, const float4* tri_bounds, const uint32_t* parent, const uint32_t* local, const uint32_t* items,
                                                   const uint32_t* batch_off, uint32_t first_batch) {
    __shared__ float boxes[kRefitBatch][12];  // (left min, left max, right min, right max) of the batch's internal nodes
    __shared__ uint32_t arrived[kRefitBatch];
    const uint32_t batch = first_batch + blockIdx.x, begin = batch_off[batch], end = batch_off[batch + 1u];
    for (uint32_t i = threadIdx.x; i < kRefitBatch; i += blockDim.x) arrived[i] = 0u;
    __syncthreads();
    for (uint32_t i = begin + threadIdx.x; i < end; i += blockDim.x) {
        const uint32_t item = items[i];
        uint32_t p = parent[item & 0x7fffffffu];
        // Aabb::grow exactly as the host applies it (st_bvh.h: every point goes through min AND max, utils/bounding_box.rs:83-88),
        // so that even the sign of a zero comes out as in the host's sweep
        V3 lo = v3s(kF32Max), hi = v3s(-kF32Max);
        bool stored = false;
        if (item >> 31) {
            const float4* at = bvh + 4u * (p >> 1) + 2u * (p & 1u);
            lo = xyz(at[0]); hi = xyz(at[1]); stored = true;
        } else {
            for (uint32_t k = item;; k++) {
                const float4 d0 = bvh[4u * k];
                const uint32_t tri = f2b(d0.y);
                const V3 blo = xyz(tri_bounds[2u * tri]), bhi = xyz(tri_bounds[2u * tri + 1u]);
                lo = vmin(lo, blo); hi = vmax(hi, blo); lo = vmin(lo, bhi); hi = vmax(hi, bhi);
                if (!(f2b(d0.x) & 1u)) break;
            }
        }
        for (;;) {
            const uint32_t pe = p >> 1, slot = p & 1u;
            if (!stored) {
                float* dst = reinterpret_cast<float*>(bvh + 4u * pe + 2u * slot);  // (min.xyz, w) (max.xyz, w): the w lanes keep their bits
                dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[4] = hi.x; dst[5] = hi.y; dst[6] = hi.z;
            }
            const uint32_t info = local[pe], l = info & 0x7fffffffu;
            float* mine = &boxes[l][6u * slot];
            mine[0] = lo.x; mine[1] = lo.y; mine[2] = lo.z; mine[3] = hi.x; mine[4] = hi.y; mine[5] = hi.z;
            __threadfence_block();
            if (atomicAdd(&arrived[l], 1u) == 0u) break;  // the sibling is still on its way: it will carry on from here
            __threadfence_block();
            const float* both = boxes[l];
            const V3 lmin = v3(both[0], both[1], both[2]), lmax = v3(both[3], both[4], both[5]), rmin = v3(both[6], both[7], both[8]), rmax = v3(both[9], both[10], both[11]);
            lo = v3s(kF32Max); hi = v3s(-kF32Max);
            lo = vmin(lo, lmin); hi = vmax(hi, lmin); lo = vmin(lo, lmax); hi = vmax(hi, lmax);
            lo = vmin(lo, rmin); hi = vmax(hi, rmin); lo = vmin(lo, rmax); hi = vmax(hi, rmax);
            p = parent[pe]; stored = false;
            if (p == 0xffffffffu) break;  // the tree's root
            if (info >> 31) {             // a task's root: its box goes into the stream for the next launch, and the task ends
                float* dst = reinterpret_cast<float*>(bvh + 4u * (p >> 1) + 2u * (p & 1u));
                dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[4] = hi.x; dst[5] = hi.y; dst[6] = hi.z;
                break;
            }
        }
    }
}

// ---- world-space baking on the device (strolle/src/instances.rs:100-139, mesh_triangle.rs:47-86; StTuning::device_bake).
// When instances only MOVE under ST_BVH_REFIT_DEVICE the host no longer bakes their triangles and sends 80 + 64 B for each: the
// object-space meshes are uploaded once (24 floats per triangle: positions, normals, uvs), a tick sends one 128-B job per moved
// instance, and this kernel writes what Engine::bake writes for the device — the hit-test record (also straight into the
// triangle's leaf entry, which is what k_bvh_patch_leaves did), the bounds k_bvh_refit reads, and the attribute record — with
// Engine::bake's own operations in its own order (this file's EXACT build is the one launched, whatever arithmetic the frame
// uses: st_engine launches launchers_exact().launch_bvh_bake), so the device arrays are bit for bit the host's
// (tests/test_gpu_parity.py test_device_bake_*). Tangents are host-only data (the reference's 144-B triangle) and are not made here.
struct BakeJobDevice { float4 x, y, z, t, r0, r1, r2; uint32_t mesh_first, count, slot_first, xslot; };
static_assert(sizeof(BakeJobDevice) == 128, "one bake job is 128 B");
__global__ __launch_bounds__(256) void k_bvh_bake(const BakeJobDevice* jobs, const uint32_t* job_start, uint32_t n_jobs, uint32_t total, const float* mesh,
                                                  float4* tri_geo, float4* tri_bounds, float4* tri_attr, float4* bvh, const uint32_t* entry_of_tri) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    uint32_t lo = 0u, hi = n_jobs;          // the job whose [job_start[j], job_start[j + 1]) holds i
    while (hi - lo > 1u) { const uint32_t mid = (lo + hi) >> 1; if (job_start[mid] <= i) lo = mid; else hi = mid; }
    const BakeJobDevice j = jobs[lo];
    const uint32_t k = i - job_start[lo];
    const float* m = mesh + 24u * (size_t)(j.mesh_first + k);
    const V3 ax = xyz(j.x), ay = xyz(j.y), az = xyz(j.z), at = xyz(j.t), r0 = xyz(j.r0), r1 = xyz(j.r1), r2 = xyz(j.r2);
    V3 p[3], n[3];
#pragma unroll
    for (int v = 0; v < 3; v++) {
        const V3 q = v3(m[3 * v], m[3 * v + 1], m[3 * v + 2]);
        p[v] = ((ax * q.x) + (ay * q.y) + (az * q.z)) + at;                       // glam Affine3A::transform_point3
        const V3 nn = v3(m[9 + 3 * v], m[9 + 3 * v + 1], m[9 + 3 * v + 2]);
        V3 acc = r0 * nn.x; acc = r1 * nn.y + acc; acc = r2 * nn.z + acc;           // transpose(inverse) * normal, Mat4::transform_vector3 order
        n[v] = normalize(acc);
    }
    const float* uv = m + 18;
    const uint32_t slot = j.slot_first + k;
    const float4 g0 = f4(p[0], 0.0f), g1 = f4(p[1] - p[0], 0.0f), g2 = f4(p[2] - p[0], 0.0f);
    tri_geo[3u * slot] = g0; tri_geo[3u * slot + 1u] = g1; tri_geo[3u * slot + 2u] = g2;
    V3 blo = v3s(kF32Max), bhi = v3s(-kF32Max);                                      // Aabb::grow: every point through min AND max
#pragma unroll
    for (int v = 0; v < 3; v++) { blo = vmin(blo, p[v]); bhi = vmax(bhi, p[v]); }
    tri_bounds[2u * slot] = f4(blo, 0.0f); tri_bounds[2u * slot + 1u] = f4(bhi, 0.0f);
    tri_attr[4u * slot] = f4(n[0], uv[0]); tri_attr[4u * slot + 1u] = f4(n[1], uv[1]); tri_attr[4u * slot + 2u] = f4(n[2], uv[2]);
    tri_attr[4u * slot + 3u] = make_float4(uv[3], uv[4], uv[5], b2f(j.xslot));
    if (entry_of_tri != nullptr) {   // (nullptr: ST_BVH_BUILD_DEVICE — there is no contract stream to patch, the tree is rebuilt from tri_geo / tri_bounds right after)
        const uint32_t e = entry_of_tri[slot];
        if (e != 0xffffffffu) { bvh[4u * e + 1u] = g0; bvh[4u * e + 2u] = g1; bvh[4u * e + 3u] = g2; }
    }
}
void launch_bvh_bake(const void* jobs, const uint32_t* job_start, uint32_t n_jobs, uint32_t total, const float* mesh, float4* tri_geo, float4* tri_bounds, float4* tri_attr,
                     float4* bvh, const uint32_t* entry_of_tri, hipStream_t s) {
    if (total) ST_KLAUNCH(k_bvh_bake, dim3((total + 255u) / 256u), dim3(256), s, static_cast<const BakeJobDevice*>(jobs), job_start, n_jobs, total, mesh, tri_geo, tri_bounds, tri_attr, bvh, entry_of_tri);
}

// ---- the COMPACT stream the fast build's shadow rays walk (st_device.h any_hit_compact; StTuning::compact_bvh).
// Measured (round 4, ST_EXP probes on the dungeon): one more 64-B line fetched per traversal step — same round trip, no arithmetic —
// takes `di_sampling+di_temporal` from 194 to 280 us, tripling the box arithmetic only to 233: the loop is bound by what the
// texture-address path has to serve per step more than by VALU issue. A shadow ray's answer is one boolean, so it may walk
// CONSERVATIVE boxes: entry k of the contract stream (64 B at texel 4 k) becomes entry k of this stream, 48 B at texel 3 k —
//   internal  texel 0: f16 x 8  left min.xyz, left max.xyz, right min.xy        (mins rounded DOWN, maxes UP: a box never shrinks)
//             texel 1: f16 x 4  right min.z, right max.xyz | u32 (far entry << 2 | right child is a leaf << 1 | left child is a leaf) | u32 0
//   leaf      texel 0: v0.xyz, bits(triangle << 2 | flags)   texel 1: (v1 - v0).xyz, bits(material)   texel 2: (v2 - v0).xyz, 0
// so an internal step fetches TWO texels and a leaf step three (the kind of a child travels with its pointer) where the contract
// stream's take four, the stream is a quarter smaller, and the near child still sits right behind its parent (depth-first order:
// a second form with the nodes and the leaf records in two arrays — 32-B nodes, two to a line — lost that adjacency and measured
// HALF the gain: 1.413 vs 1.388 ms against 1.435). f16 keeps 11 significant bits: a box grows by at most 2^-10 of its coordinate's
// magnitude (3 cm at 32 units), i.e. a few more entries are visited and no triangle a ray hits is ever skipped. The triangle records
// stay f32: the same hit test. The stream is a pure function of the contract stream on the device — one thread per entry — and is
// regenerated after every upload, leaf patch, device bake or refit of it.
__global__ __launch_bounds__(256) void k_bvh_compact(const float4* bvh, uint32_t n_entries, float4* out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_entries) return;
    const float4 d0 = bvh[4u * k], d1 = bvh[4u * k + 1u], d2 = bvh[4u * k + 2u], d3 = bvh[4u * k + 3u];
    if (f2b(d0.w) != 0u) {   // leaf entry
        out[3u * k] = make_float4(d1.x, d1.y, d1.z, b2f((f2b(d0.y) << 2) | (f2b(d0.x) & 3u)));
        out[3u * k + 1u] = make_float4(d2.x, d2.y, d2.z, d0.z);
        out[3u * k + 2u] = make_float4(d3.x, d3.y, d3.z, 0.0f);
        return;
    }
    const uint32_t far_entry = f2b(d1.w) >> 6;
    const uint32_t left_leaf = f2b(bvh[4u * (k + 1u)].w) != 0u ? 1u : 0u, right_leaf = f2b(bvh[4u * far_entry].w) != 0u ? 1u : 0u;
    auto dn = [](float x) { return (uint32_t)__half_as_ushort(__float2half_rd(x)); };
    auto up = [](float x) { return (uint32_t)__half_as_ushort(__float2half_ru(x)); };
    uint4 t0, t1;
    // a word per axis and box: (lower bound | upper bound << 16) — st_device.h compact_slab picks the entry plane by rotating the word
    t0.x = dn(d0.x) | (up(d1.x) << 16); t0.y = dn(d0.y) | (up(d1.y) << 16); t0.z = dn(d0.z) | (up(d1.z) << 16); t0.w = dn(d2.x) | (up(d3.x) << 16);
    t1.x = dn(d2.y) | (up(d3.y) << 16); t1.y = dn(d2.z) | (up(d3.z) << 16); t1.z = (far_entry << 2) | (right_leaf << 1) | left_leaf; t1.w = 0u;
    out[3u * k] = make_float4(b2f(t0.x), b2f(t0.y), b2f(t0.z), b2f(t0.w));
    out[3u * k + 1u] = make_float4(b2f(t1.x), b2f(t1.y), b2f(t1.z), b2f(t1.w));
    out[3u * k + 2u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
}
void launch_bvh_compact(const float4* bvh, uint32_t n_entries, float4* out, hipStream_t s) {
    if (n_entries) ST_KLAUNCH(k_bvh_compact, dim3((n_entries + 255u) / 256u), dim3(256), s, bvh, n_entries, out);
}

// ---- the WIDE stream (st_device.h any_hit_wide / closest_hit_wide; StTuning::wide_bvh): 4-wide nodes whose child boxes are the contract
// stream's own, read from the device copy as it is NOW (after an upload, leaf patch, device bake or refit), rounded outwards to f16.
// The topology — which binary nodes were collapsed into which wide node — is the host's (st_bvh_refresh.cpp build_wide_topology, once per
// build of the tree): topo[8 n + c] = where child c's box lives in the contract stream (entry << 1 | 0: left box, 1: right box; ~0: empty
// slot), topo[8 n + 4 + c] = the child's link (wide node or leaf record index << 1 | is a leaf record). leaf_entry[k] = contract entry of
// leaf record k. One thread per node, then one per leaf record.
__global__ __launch_bounds__(256) void k_bvh_wide(const float4* bvh, const uint32_t* topo, uint32_t n_nodes, const uint32_t* leaf_entry, uint32_t n_leaves, uint32_t links16,
                                                  float4* nodes, float4* leaves) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    auto dn = [](float x) { return (uint32_t)__half_as_ushort(__float2half_rd(x)); };
    auto up = [](float x) { return (uint32_t)__half_as_ushort(__float2half_ru(x)); };
    if (i < n_nodes) {
        uint32_t w[12], link[4];
#pragma unroll
        for (int c = 0; c < 4; c++) {
            const uint32_t src = topo[8u * i + c];
            link[c] = topo[8u * i + 4u + c];
            if (src == 0xffffffffu) {   // empty slot: lower = +inf, upper = -inf — no ray enters it
                w[3 * c] = w[3 * c + 1] = w[3 * c + 2] = 0xfc007c00u;
            } else {
                const float4 lo = bvh[4u * (src >> 1) + 2u * (src & 1u)], hi = bvh[4u * (src >> 1) + 2u * (src & 1u) + 1u];
                w[3 * c] = dn(lo.x) | (up(hi.x) << 16); w[3 * c + 1] = dn(lo.y) | (up(hi.y) << 16); w[3 * c + 2] = dn(lo.z) | (up(hi.z) << 16);
            }
        }
        nodes[4u * i] = make_float4(b2f(w[0]), b2f(w[1]), b2f(w[2]), b2f(w[3]));
        nodes[4u * i + 1u] = make_float4(b2f(w[4]), b2f(w[5]), b2f(w[6]), b2f(w[7]));
        nodes[4u * i + 2u] = make_float4(b2f(w[8]), b2f(w[9]), b2f(w[10]), b2f(w[11]));
        nodes[4u * i + 3u] = links16 ? make_float4(b2f(link[0] | (link[1] << 16)), b2f(link[2] | (link[3] << 16)), 0.0f, 0.0f)
                                     : make_float4(b2f(link[0]), b2f(link[1]), b2f(link[2]), b2f(link[3]));
        return;
    }
    const uint32_t k = i - n_nodes;
    if (k >= n_leaves) return;
    const uint32_t e = leaf_entry[k];
    const float4 d0 = bvh[4u * e], d1 = bvh[4u * e + 1u], d2 = bvh[4u * e + 2u], d3 = bvh[4u * e + 3u];
    leaves[3u * k] = make_float4(d1.x, d1.y, d1.z, b2f((f2b(d0.y) << 2) | (f2b(d0.x) & 3u)));
    leaves[3u * k + 1u] = make_float4(d2.x, d2.y, d2.z, d0.z);
    leaves[3u * k + 2u] = make_float4(d3.x, d3.y, d3.z, 0.0f);
}
void launch_bvh_wide(const float4* bvh, const uint32_t* topo, uint32_t n_nodes, const uint32_t* leaf_entry, uint32_t n_leaves, uint32_t links16, float4* nodes, float4* leaves, hipStream_t s) {
    const uint32_t n = n_nodes + n_leaves;
    if (n) ST_KLAUNCH(k_bvh_wide, dim3((n + 255u) / 256u), dim3(256), s, bvh, topo, n_nodes, leaf_entry, n_leaves, links16, nodes, leaves);
}

void launch_bvh_patch_leaves(float4* bvh, const float4* tri_geo, const uint32_t* entry_of_tri, uint32_t lo, uint32_t hi, hipStream_t s) {
    if (hi > lo) ST_KLAUNCH(k_bvh_patch_leaves, dim3((hi - lo + 255u) / 256u), dim3(256), s, bvh, tri_geo, entry_of_tri, lo, hi);
}
void launch_bvh_refit(float4* bvh, const float4* tri_bounds, const uint32_t* parent, const uint32_t* local, const uint32_t* items, const uint32_t* batch_off, uint32_t first_batch, uint32_t batches, hipStream_t s) {
    if (batches) ST_KLAUNCH(k_bvh_refit, dim3(batches), dim3(256), s, bvh, tri_bounds, parent, local, items, batch_off, first_batch);
}

}  // namespace ST_KNS
}  // namespace st
