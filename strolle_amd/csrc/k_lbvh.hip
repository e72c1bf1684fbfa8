// k_lbvh.hip — a BVH built ON THE DEVICE, straight into the wide stream (st_set_bvh_refresh(ST_BVH_BUILD_DEVICE); round 5).
//
// The reference rebuilds its binned-SAH tree on the host whenever an instance appears or disappears (strolle/src/bvh/builder.rs:17-228,
// bevy-strolle/examples/stress-bvh.rs:111-167); this library's host builder is the same tree bit for bit (st_bvh.h: that is what makes the
// heatmap's `used_memory` integers the reference's) and costs 26-28 ms per spawn at 208 k triangles. The fast build's rays do not need THAT
// tree — they owe the reference their hits, not their path (st_device.h closest_hit_wide) — so while no camera observes the contract stream
// (no heatmap camera, fast arithmetic, no byte counting) a scene change is followed by this builder instead:
//
//   1. centroid bounds of the live triangle slots                                   k_lbvh_bounds      (wave, then workgroup reduction: one ordered-int atomic pair per workgroup)
//   2. key = 30-bit Morton code of the centroid << 32 | triangle slot (unique)      k_lbvh_keys        dead slots: ~0, sorted to the end; cells of bounded aspect (kLbvhCellAspect)
//   3. radix sort                                                                   rocPRIM radix_sort_pairs over the 31-bit (code | dead) keys with the slot as value
//      (round 5 sorted 64-bit keys); the sort is stable and its input in slot order, so the order is the same, and k_lbvh_compose rebuilds the 64-bit keys
//      afterwards. Half the key bytes — and, measured, no faster: at 208 k keys the library runs the same 17 launches of 5-6 us either way (0.10 ms, launch-bound;
//      lbvh_sort_temp_bytes: the two replacements that were tried)
//   4. leaf records (48 B, sorted order = leaf index) + leaf boxes                  k_lbvh_leaves
//   5. min / max segment tree over the sorted leaf boxes                            k_lbvh_seg_levels  nine levels per launch, no fences
//   6. the binary radix tree of Karras 2012 — one thread per internal node finds its range and split from the keys alone —, each
//      node's box as a range query of the segment tree (no bottom-up pass: nothing is handed from workgroup to workgroup) k_lbvh_hierarchy
//   7. 4-wide nodes: a wide node is headed by a binary node and takes its children's children, largest surface area first, until it has
//      four (the rule st_bvh_refresh.cpp build_wide_topology applies to the host's tree); it lives at its head's binary index, so nothing
//      is allocated and the result does not depend on the schedule. ONE launch: every binary node writes the node it would head, and the
//      links decide which of them a walk reaches (rounds 5-6: a frontier walk, 14 launches)                             k_lbvh_wide_nodes
//
//   R. instances only moved: steps 4-6 again over the SAME sorted order, then every wide node's box words from its links           lbvh_refit
//      (5 launches against the build's 22; st_tick.cpp: at most 15 refits between two builds)
//
// 208 k triangles, one build: 0.17 ms of device time (profiles/r06_lbvh_kernel_stats_one_launch.txt, r06_lbvh_sort_config.txt; round 5: 0.34). Host model (tools/bvh4_sim.py's rays over this tree, dungeon): 13.8 node steps per primary ray against the SAH tree's 13.6, 14.3 against
// 11.6 for a GI bounce, the same number of triangle tests, deepest stack 13-14.
#include <hip/hip_fp16.h>
#include <rocprim/device/device_radix_sort.hpp>
#include "k_common.h"
#include "st_lbvh.h"

namespace st {

namespace {
constexpr int kT = 256;
__device__ inline int ordered(float f) { const int i = __float_as_int(f); return i >= 0 ? i : i ^ 0x7fffffff; }
__device__ inline float unordered(int i) { return __int_as_float(i >= 0 ? i : i ^ 0x7fffffff); }
__device__ inline uint32_t expand10(uint32_t x) {
    x &= 0x3ffu; x = (x | (x << 16)) & 0x30000ffu; x = (x | (x << 8)) & 0x300f00fu; x = (x | (x << 4)) & 0x30c30c3u; x = (x | (x << 2)) & 0x9249249u;
    return x;
}
struct Box { float lx, ly, lz, hx, hy, hz; };
__device__ inline Box box_empty() { return {kF32Max, kF32Max, kF32Max, -kF32Max, -kF32Max, -kF32Max}; }
__device__ inline Box box_union(const Box& a, const Box& b) { return {fminf(a.lx, b.lx), fminf(a.ly, b.ly), fminf(a.lz, b.lz), fmaxf(a.hx, b.hx), fmaxf(a.hy, b.hy), fmaxf(a.hz, b.hz)}; }
__device__ inline Box box_load(const float4* p) { const float4 a = p[0], b = p[1]; return {a.x, a.y, a.z, b.x, b.y, b.z}; }
__device__ inline void box_store(float4* p, const Box& b) { p[0] = make_float4(b.lx, b.ly, b.lz, 0.0f); p[1] = make_float4(b.hx, b.hy, b.hz, 0.0f); }
__device__ inline float box_area(const Box& b) {
    const float dx = fmaxf(b.hx - b.lx, 0.0f), dy = fmaxf(b.hy - b.ly, 0.0f), dz = fmaxf(b.hz - b.lz, 0.0f);
    return dx * dy + dy * dz + dz * dx;
}

__global__ void k_lbvh_init(int* bounds) {
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0x7fffffff;             // ordered(+inf-ish): min
    else if (threadIdx.x < 6) bounds[threadIdx.x] = (int)0x80000000;   // max
}
// 1. bounds of the live triangles' centroids. One ordered-int atomic pair per axis and WORKGROUP, from at most kBoundsBlocks workgroups: with one
// pair per wave of a launch that covered the slots once (3,252 waves at 208 k triangles = 19,500 atomics on one cache line) this kernel took
// 34-225 us depending on the box — the largest single item of the build (profiles/r05_lbvh_kernel_stats.txt).
constexpr uint32_t kBoundsBlocks = 256;
__global__ __launch_bounds__(kT) void k_lbvh_bounds(const float4* tri_bounds, const uint32_t* tri_info, uint32_t slots, int* bounds) {
    __shared__ float s_mn[3][kT / 64], s_mx[3][kT / 64];
    float mn[3] = {kF32Max, kF32Max, kF32Max}, mx[3] = {-kF32Max, -kF32Max, -kF32Max};
    for (uint32_t i = blockIdx.x * kT + threadIdx.x; i < slots; i += gridDim.x * kT) {
        if (!(tri_info[i] & 1u)) continue;
        const float4 lo = tri_bounds[2u * i], hi = tri_bounds[2u * i + 1u];
        const float c[3] = {(lo.x + hi.x) * 0.5f, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f};
        for (int k = 0; k < 3; k++) { mn[k] = fminf(mn[k], c[k]); mx[k] = fmaxf(mx[k], c[k]); }
    }
    for (int k = 0; k < 3; k++)
        for (int off = 32; off >= 1; off >>= 1) { mn[k] = fminf(mn[k], __shfl_xor(mn[k], off)); mx[k] = fmaxf(mx[k], __shfl_xor(mx[k], off)); }
    if ((threadIdx.x & 63u) == 0u)
        for (int k = 0; k < 3; k++) { s_mn[k][threadIdx.x >> 6] = mn[k]; s_mx[k][threadIdx.x >> 6] = mx[k]; }
    __syncthreads();
    if (threadIdx.x < 3u) {
        const int k = (int)threadIdx.x;
        float lo = s_mn[k][0], hi = s_mx[k][0];
        for (int w = 1; w < kT / 64; w++) { lo = fminf(lo, s_mn[k][w]); hi = fmaxf(hi, s_mx[k][w]); }
        atomicMin(&bounds[k], ordered(lo)); atomicMax(&bounds[3 + k], ordered(hi));
    }
}
// 2. sort keys
constexpr uint32_t kDeadCode = 0x40000000u;   // above every 30-bit Morton code: dead slots sort to the end (31 key bits)
__global__ __launch_bounds__(kT) void k_lbvh_keys(const float4* tri_bounds, const uint32_t* tri_info, uint32_t slots, const int* bounds, uint32_t* codes, uint32_t* slot_of, float cell_aspect) {
    const uint32_t i = blockIdx.x * kT + threadIdx.x;
    if (i >= slots) return;
    slot_of[i] = i;
    if (!(tri_info[i] & 1u)) { codes[i] = kDeadCode; return; }
    const float4 lo = tri_bounds[2u * i], hi = tri_bounds[2u * i + 1u];
    const float c[3] = {(lo.x + hi.x) * 0.5f, (lo.y + hi.y) * 0.5f, (lo.z + hi.z) * 0.5f};
    // Morton cells of bounded aspect (round 6, late; st_lbvh.h kLbvhCellAspect): an axis is quantised by its own extent but never finer than cell_aspect x the largest one.
    // With each axis stretched to 1,024 cells of its own (rounds 5-6) a wide, low scene got every third split along its thin axis — slabs cut where they are thinnest.
    float largest = 0.0f;
    for (int k = 0; k < 3; k++) largest = fmaxf(largest, unordered(bounds[3 + k]) - unordered(bounds[k]));
    uint32_t q[3];
    for (int k = 0; k < 3; k++) {
        const float ext = fmaxf(unordered(bounds[3 + k]) - unordered(bounds[k]), largest * cell_aspect);   // cell_aspect 1: cubic cells; 0: every axis its own 1,024
        const float t = ext > 0.0f ? (c[k] - unordered(bounds[k])) / ext : 0.0f;
        q[k] = (uint32_t)fminf(fmaxf(t * 1024.0f, 0.0f), 1023.0f);
    }
    codes[i] = expand10(q[0]) | (expand10(q[1]) << 1) | (expand10(q[2]) << 2);
}
// 3b. the 64-bit keys the hierarchy's tie-breaks want, from the sorted (code, slot) pairs: unique, and ascending because the sort is stable
__global__ __launch_bounds__(kT) void k_lbvh_compose(const uint32_t* codes, const uint32_t* slot_of, uint32_t slots, unsigned long long* keys) {
    const uint32_t i = blockIdx.x * kT + threadIdx.x;
    if (i < slots) keys[i] = codes[i] == kDeadCode ? ~0ull : (((unsigned long long)codes[i] << 32) | slot_of[i]);
}
// 4. leaf records and leaf boxes, in sorted order; the segment tree's unused leaves are empty boxes
__global__ __launch_bounds__(kT) void k_lbvh_leaves(const unsigned long long* keys, uint32_t n, uint32_t pow2, const float4* tri_geo, const float4* tri_bounds,
                                                    const uint32_t* tri_info, float4* seg, float4* leaves) {
    const uint32_t p = blockIdx.x * kT + threadIdx.x;
    if (p >= pow2) return;
    if (p >= n) { box_store(seg + 2u * (size_t)(pow2 + p), box_empty()); return; }
    const uint32_t slot = (uint32_t)keys[p];
    seg[2u * (size_t)(pow2 + p)] = tri_bounds[2u * slot]; seg[2u * (size_t)(pow2 + p) + 1u] = tri_bounds[2u * slot + 1u];
    const uint32_t info = tri_info[slot];
    const float4 g0 = tri_geo[3u * slot], g1 = tri_geo[3u * slot + 1u], g2 = tri_geo[3u * slot + 2u];
    leaves[3u * p] = make_float4(g0.x, g0.y, g0.z, b2f((slot << 2) | (info & 2u)));   // bit 0 (another record of the run follows): single-triangle leaves
    leaves[3u * p + 1u] = make_float4(g1.x, g1.y, g1.z, b2f(info >> 2));
    leaves[3u * p + 2u] = make_float4(g2.x, g2.y, g2.z, 0.0f);
}
// 5. the segment tree, up to log2(kT) + 1 levels per launch: workgroup b computes its min(count0, kT) nodes of the level that has `count0` nodes from
// their children in global memory and every ancestor those nodes have among themselves from LDS (208 k triangles: 18 levels, 2 launches)
__global__ __launch_bounds__(kT) void k_lbvh_seg_levels(float4* seg, uint32_t count0) {
    __shared__ Box s_box[2 * kT];   // the workgroup's subtree as a heap: its bottom level at [width, 2 * width)
    const uint32_t t = threadIdx.x, width = count0 < (uint32_t)kT ? count0 : (uint32_t)kT, groups = count0 / width;
    if (t < width) {
        const size_t k = (size_t)count0 + blockIdx.x * width + t;
        const Box b = box_union(box_load(seg + 4u * k), box_load(seg + 4u * k + 2u));
        s_box[width + t] = b; box_store(seg + 2u * k, b);
    }
    for (uint32_t w = width >> 1; w >= 1u; w >>= 1) {
        __syncthreads();
        if (t < w) {
            const Box b = box_union(s_box[2u * (w + t)], s_box[2u * (w + t) + 1u]);
            s_box[w + t] = b; box_store(seg + 2u * ((size_t)groups * w + blockIdx.x * w + t), b);   // the level of groups * w nodes starts at node groups * w
        }
    }
}
__device__ inline Box seg_query(const float4* seg, uint32_t pow2, uint32_t first, uint32_t last) {
    Box b = box_empty();
    for (uint32_t l = first + pow2, r = last + pow2 + 1u; l < r; l >>= 1, r >>= 1) {
        if (l & 1u) { b = box_union(b, box_load(seg + 2u * (size_t)l)); l++; }
        if (r & 1u) { r--; b = box_union(b, box_load(seg + 2u * (size_t)r)); }
    }
    return b;
}
// 6. Karras 2012: internal node i of the binary radix tree over n sorted, unique keys. children[i] = (left, right) as links (index << 1 | is a
// leaf), node_box[i] = the box of its range.
__device__ inline int lb_delta(const unsigned long long* keys, int n, int i, int j) { return (j < 0 || j >= n) ? -1 : __clzll((long long)(keys[i] ^ keys[j])); }
__global__ __launch_bounds__(kT) void k_lbvh_hierarchy(const unsigned long long* keys, uint32_t n_, uint32_t pow2, const float4* seg, uint2* children, float4* node_box) {
    const int n = (int)n_, i = (int)(blockIdx.x * kT + threadIdx.x);
    if (i >= n - 1) return;
    const int d = lb_delta(keys, n, i, i + 1) - lb_delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
    const int dmin = lb_delta(keys, n, i, i - d);
    int lmax = 2;
    while (lb_delta(keys, n, i, i + lmax * d) > dmin) lmax <<= 1;
    int l = 0;
    for (int t = lmax >> 1; t >= 1; t >>= 1) if (lb_delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
    const int j = i + l * d;
    const int dnode = lb_delta(keys, n, i, j);
    int s = 0, t = l;
    do { t = (t + 1) >> 1; if (lb_delta(keys, n, i, i + (s + t) * d) > dnode) s += t; } while (t > 1);
    const int gamma = i + s * d + (d < 0 ? d : 0);
    const int first = i < j ? i : j, last = i < j ? j : i;
    const uint32_t left = first == gamma ? ((uint32_t)gamma << 1) | 1u : (uint32_t)gamma << 1;
    const uint32_t right = last == gamma + 1 ? ((uint32_t)(gamma + 1) << 1) | 1u : (uint32_t)(gamma + 1) << 1;
    children[i] = make_uint2(left, right);
    box_store(node_box + 2u * (size_t)i, seg_query(seg, pow2, (uint32_t)first, (uint32_t)last));
}
// 7. one wide node, headed by binary node `b`: written at node index b
__device__ inline Box lb_child_box(uint32_t link, const float4* seg, uint32_t pow2, const float4* node_box) {
    return (link & 1u) ? box_load(seg + 2u * (size_t)(pow2 + (link >> 1))) : box_load(node_box + 2u * (size_t)(link >> 1));
}
__device__ inline void lb_emit(uint32_t b, const uint2* children, const float4* node_box, const float4* seg, uint32_t pow2, uint32_t links16, float4* nodes) {
    // (four slots addressed by compile-time indices only: with run-time indices — the first version shifted the slots to keep the children in
    // the binary tree's order — the arrays lived in scratch memory, 112 B per lane, and every access was a round trip through the vector L1)
    uint32_t link[4]; Box box[4]; int n = 2;
    const uint2 c = children[b];
    link[0] = c.x; link[1] = c.y; link[2] = link[3] = 1u;
    box[0] = lb_child_box(c.x, seg, pow2, node_box); box[1] = lb_child_box(c.y, seg, pow2, node_box); box[2] = box[3] = box_empty();
#pragma unroll
    for (int round = 0; round < 2; round++) {   // the internal child of the largest surface area gives way to its two children: one in its slot, one in the next free slot
        int pick = -1; float best = -1.0f; uint32_t pick_link = 0u;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (i < n && !(link[i] & 1u)) { const float ar = box_area(box[i]); if (ar > best) { best = ar; pick = i; pick_link = link[i]; } }
        if (pick < 0) break;
        const uint2 g = children[pick_link >> 1];
        const Box bx = lb_child_box(g.x, seg, pow2, node_box), by = lb_child_box(g.y, seg, pow2, node_box);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (i == pick) { link[i] = g.x; box[i] = bx; }
            if (i == n) { link[i] = g.y; box[i] = by; }
        }
        n++;
    }
    auto dn = [](float x) { return (uint32_t)__half_as_ushort(__float2half_rd(x)); };
    auto up = [](float x) { return (uint32_t)__half_as_ushort(__float2half_ru(x)); };
    uint32_t w[12], l[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (i >= n) { w[3 * i] = w[3 * i + 1] = w[3 * i + 2] = 0xfc007c00u; l[i] = 0u; continue; }
        w[3 * i] = dn(box[i].lx) | (up(box[i].hx) << 16); w[3 * i + 1] = dn(box[i].ly) | (up(box[i].hy) << 16); w[3 * i + 2] = dn(box[i].lz) | (up(box[i].hz) << 16);
        l[i] = link[i];
    }
    float4* out = nodes + 4u * (size_t)b;
    out[0] = make_float4(b2f(w[0]), b2f(w[1]), b2f(w[2]), b2f(w[3]));
    out[1] = make_float4(b2f(w[4]), b2f(w[5]), b2f(w[6]), b2f(w[7]));
    out[2] = make_float4(b2f(w[8]), b2f(w[9]), b2f(w[10]), b2f(w[11]));
    out[3] = links16 ? make_float4(b2f(l[0] | (l[1] << 16)), b2f(l[2] | (l[3] << 16)), 0.0f, 0.0f) : make_float4(b2f(l[0]), b2f(l[1]), b2f(l[2]), b2f(l[3]));
}
// The collapse, in ONE launch (round 6, late): EVERY binary node writes the wide node it would head. Which binary nodes DO head a wide node is only known
// from the top down (the root does; a wide node's internal children do) — rounds 5 and 6 walked that frontier level by level: 14 launches, 158 us of the
// 368-us build at 208 k triangles, a finishing launch with a private stack that could overflow. But a walk only ever follows links from the root: a wide
// node at a slot no link points to is never read, so nobody needs to know which slots those are. Two thirds of the launch's nodes are written for nothing
// (13 MB instead of 4.4 at 208 k triangles); it has no dependency between its threads, no frontier arrays, no counters, no stack, nothing that can fail.
__global__ __launch_bounds__(kT) void k_lbvh_wide_nodes(const uint2* children, const float4* node_box, const float4* seg, uint32_t pow2, uint32_t links16, float4* nodes, uint32_t n_nodes) {
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    if (b < n_nodes) lb_emit(b, children, node_box, seg, pow2, links16, nodes);
}
// REFIT (lbvh_refit): triangles moved, nothing else changed — the sorted order, the binary tree and the wide nodes' links stay, the boxes follow. Every
// slot of the node array (every one holds a wide node since k_lbvh_wide_nodes writes them all; empty child slots have link 0, the root's, which no child
// has) gets its children's boxes again from the recomputed node boxes / segment-tree leaves.
__global__ __launch_bounds__(kT) void k_lbvh_refit_nodes(const float4* node_box, const float4* seg, uint32_t pow2, uint32_t links16, float4* nodes, uint32_t n_nodes) {
    const uint32_t b = blockIdx.x * kT + threadIdx.x;
    if (b >= n_nodes) return;
    float4* out = nodes + 4u * (size_t)b;
    const float4 lw = out[3];
    uint32_t l[4];
    if (links16) { l[0] = f2b(lw.x) & 0xffffu; l[1] = f2b(lw.x) >> 16; l[2] = f2b(lw.y) & 0xffffu; l[3] = f2b(lw.y) >> 16; }
    else { l[0] = f2b(lw.x); l[1] = f2b(lw.y); l[2] = f2b(lw.z); l[3] = f2b(lw.w); }
    auto dn = [](float x) { return (uint32_t)__half_as_ushort(__float2half_rd(x)); };
    auto up = [](float x) { return (uint32_t)__half_as_ushort(__float2half_ru(x)); };
    uint32_t w[12];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        if (l[i] == 0u) { w[3 * i] = w[3 * i + 1] = w[3 * i + 2] = 0xfc007c00u; continue; }
        const Box bx = lb_child_box(l[i], seg, pow2, node_box);
        w[3 * i] = dn(bx.lx) | (up(bx.hx) << 16); w[3 * i + 1] = dn(bx.ly) | (up(bx.hy) << 16); w[3 * i + 2] = dn(bx.lz) | (up(bx.hz) << 16);
    }
    out[0] = make_float4(b2f(w[0]), b2f(w[1]), b2f(w[2]), b2f(w[3]));
    out[1] = make_float4(b2f(w[4]), b2f(w[5]), b2f(w[6]), b2f(w[7]));
    out[2] = make_float4(b2f(w[8]), b2f(w[9]), b2f(w[10]), b2f(w[11]));
}
}  // namespace

// The sort: rocPRIM's radix_sort_pairs, which below 1 M keys is its merge sort — one block sort and two launches per doubling of the sorted runs: with the
// default configuration (runs of 1,024) 17 launches, 99-110 us of launch latency, at 208 k keys; LbvhSort block-sorts runs of 4,096 (1,024 threads x 4 keys):
// 13 launches, 82 us — the block sort 9 -> 18 us, four merge launches of 5 us fewer; runs of 8,192: 11 launches but a 30-us block sort, 95 us
// (profiles/r06_lbvh_sort_config.txt). Two replacements were built and measured in round 6, both giving the same order, both slower:
// the library's onesweep radix sort (radix_sort_config<..., MergeSortLimit = 8192>: 6 launches, but 23-29 us per pass and 10-20 us of memsets between
// them: 163 us; profiles/r06_lbvh_onesweep.txt) and an own bucket-and-rank form (3 launches; 540 us on the dungeon, whose slot order and tori make its
// atomics collide and its buckets huge; tools/experiments/lbvh_bucket_sort.inc, profiles/r06_lbvh_bucket_sort.txt).
using LbvhSort = rocprim::radix_sort_config<rocprim::default_config, rocprim::merge_sort_config<512u, 1024u, 4u>, rocprim::default_config>;
size_t lbvh_sort_temp_bytes(uint32_t slots) {
    size_t bytes = 0;
    (void)rocprim::radix_sort_pairs<LbvhSort>(nullptr, bytes, (const uint32_t*)nullptr, (uint32_t*)nullptr, (const uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)slots, 0u, 31u, (hipStream_t) nullptr);
    return bytes;
}
uint32_t lbvh_pow2(uint32_t n) { uint32_t p = 1; while (p < n) p <<= 1; return p; }

// Ahead of the first build (st_tick.cpp, while the scene loads): this file's code object on the device — 5 ms of the first spawn's tick otherwise
// (profiles/r06_spawn_ticks.txt: "build launches 5.095 ms" against 0.15 for every later one). `bounds`: the six ints every build initialises itself.
void lbvh_warm(int* bounds, hipStream_t s) {
    hipFuncAttributes attr;
    (void)hipFuncGetAttributes(&attr, reinterpret_cast<const void*>(k_lbvh_wide_nodes));
    hipLaunchKernelGGL(k_lbvh_init, dim3(1), dim3(64), 0, s, bounds);
    (void)hipGetLastError();
}

int lbvh_build(const LbvhArgs& a, hipStream_t s) {
    if (a.live < 2u) return -1;   // the caller keeps the host path for a scene of fewer than two triangles
    const uint32_t pow2 = lbvh_pow2(a.live);
    auto grid = [](uint32_t n) { return dim3((n + kT - 1) / kT); };
    hipLaunchKernelGGL(k_lbvh_init, dim3(1), dim3(64), 0, s, a.bounds);
    hipLaunchKernelGGL(k_lbvh_bounds, dim3(std::min<uint32_t>((a.slots + kT - 1) / kT, kBoundsBlocks)), dim3(kT), 0, s, a.tri_bounds, a.tri_info, a.slots, a.bounds);
    // (code, slot) pairs: unsorted in keys_out's memory, sorted into keys_in's, the composed 64-bit keys back in keys_out (what the hierarchy and every
    // later refit read)
    uint32_t* codes_in = reinterpret_cast<uint32_t*>(a.keys_out); uint32_t* slots_in = codes_in + a.slots;
    uint32_t* codes_out = reinterpret_cast<uint32_t*>(a.keys_in); uint32_t* slots_out = codes_out + a.slots;
    hipLaunchKernelGGL(k_lbvh_keys, grid(a.slots), dim3(kT), 0, s, a.tri_bounds, a.tri_info, a.slots, a.bounds, codes_in, slots_in, a.cell_aspect);
    size_t temp = a.sort_temp_bytes;
    if (rocprim::radix_sort_pairs<LbvhSort>(a.sort_temp, temp, codes_in, codes_out, slots_in, slots_out, (size_t)a.slots, 0u, 31u, s) != hipSuccess) return -2;
    hipLaunchKernelGGL(k_lbvh_compose, grid(a.slots), dim3(kT), 0, s, codes_out, slots_out, a.slots, a.keys_out);
    hipLaunchKernelGGL(k_lbvh_leaves, grid(pow2), dim3(kT), 0, s, a.keys_out, a.live, pow2, a.tri_geo, a.tri_bounds, a.tri_info, a.seg, a.leaves);
    for (uint32_t count = pow2 >> 1; count >= 1u;) {
        const uint32_t width = std::min<uint32_t>(count, (uint32_t)kT), groups = count / width;   // after this launch: every level down to the one of `groups` nodes
        hipLaunchKernelGGL(k_lbvh_seg_levels, dim3(groups), dim3(kT), 0, s, a.seg, count);
        count = groups >> 1;
    }
    hipLaunchKernelGGL(k_lbvh_hierarchy, grid(a.live - 1u), dim3(kT), 0, s, a.keys_out, a.live, pow2, a.seg, a.children, a.node_box);
    hipLaunchKernelGGL(k_lbvh_wide_nodes, grid(a.live - 1u), dim3(kT), 0, s, a.children, a.node_box, a.seg, pow2, a.links16, a.nodes, a.live - 1u);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

// The same arguments as the build that made the tree (keys_out, children and nodes are READ: they must still hold what that build left), after the
// triangles' records and bounds changed in place: leaf records and boxes, the segment tree, the binary nodes' boxes, the wide nodes' boxes. 5 launches.
int lbvh_refit(const LbvhArgs& a, hipStream_t s) {
    if (a.live < 2u) return -1;
    const uint32_t pow2 = lbvh_pow2(a.live);
    auto grid = [](uint32_t n) { return dim3((n + kT - 1) / kT); };
    hipLaunchKernelGGL(k_lbvh_leaves, grid(pow2), dim3(kT), 0, s, a.keys_out, a.live, pow2, a.tri_geo, a.tri_bounds, a.tri_info, a.seg, a.leaves);
    for (uint32_t count = pow2 >> 1; count >= 1u;) {
        const uint32_t width = std::min<uint32_t>(count, (uint32_t)kT), groups = count / width;
        hipLaunchKernelGGL(k_lbvh_seg_levels, dim3(groups), dim3(kT), 0, s, a.seg, count);
        count = groups >> 1;
    }
    hipLaunchKernelGGL(k_lbvh_hierarchy, grid(a.live - 1u), dim3(kT), 0, s, a.keys_out, a.live, pow2, a.seg, a.children, a.node_box);   // (the children come out as they were)
    hipLaunchKernelGGL(k_lbvh_refit_nodes, grid(a.live - 1u), dim3(kT), 0, s, a.node_box, a.seg, pow2, a.links16, a.nodes, a.live - 1u);
    return hipGetLastError() == hipSuccess ? 0 : -3;
}

}  // namespace st
