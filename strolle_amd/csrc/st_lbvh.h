// st_lbvh.h — interface of the device BVH builder (k_lbvh.hip): built once into the library, whatever arithmetic the frames use.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace st {

// Everything is device memory the caller owns. tri_info: one word per triangle slot — bit 0 the slot is live, bit 1 its material is
// AlphaMode::Blend, bits 2.. its material slot. Output: `nodes` (64 B per node, node index = the head's index in the binary radix tree:
// live - 1 slots, ALL written — a walk from the root, node 0, reaches about a third of them: k_lbvh.hip k_lbvh_wide_nodes) and `leaves` (48 B per
// record, sorted order), in the wide stream's format (st_device.h closest_hit_wide). Scratch: keys_in / keys_out (8 B x slots), sort_temp
// (lbvh_sort_temp_bytes(slots)), seg (2 x lbvh_pow2(live) boxes of 32 B), children (8 B x live), node_box (32 B x live), bounds (6 ints).
// A build cannot fail half-way: no launch of it depends on a count another one produced, none keeps a stack (rounds 5 and 6 collapsed the binary tree
// frontier by frontier and finished with private stacks that could overflow; see k_lbvh_wide_nodes).
// The sort (round 6): 31-bit keys — the 30-bit Morton code, 0x40000000 for a dead slot — with the slot as value (rocPRIM radix_sort_pairs); the sort is stable and the
// input is in slot order, so the order is the one (code << 32 | slot) gives, and those 64-bit keys are composed afterwards for the hierarchy's tie-breaks: the
// tree is bit for bit the one round 5's 64-bit sort produced.
struct LbvhArgs {
    const float4* tri_geo; const float4* tri_bounds; const uint32_t* tri_info;
    uint32_t slots, live, links16;
    float cell_aspect;   // Morton cells: an axis is quantised by max(its own extent, cell_aspect x the largest extent) — 1: cubic cells, 0: 1,024 cells of its own per axis
    float4* nodes; float4* leaves;
    unsigned long long* keys_in; unsigned long long* keys_out; void* sort_temp; size_t sort_temp_bytes;
    float4* seg; uint2* children; float4* node_box; int* bounds;
};
// LbvhArgs::cell_aspect as shipped: no Morton cell more than 8 times as long along one axis as along another. Measured (tools/cell_aspect.py, profiles/r06_lbvh_cell_aspect.txt:
// nine scenes of 13 k - 537 k triangles, steady frame over the device-built tree): with every axis its own 1,024 cells (0; rounds 5-6) a wide, low scene — 16 copies of the
// dungeon's level side by side: 194 x 7.6 x 392 m — is sliced along its thin axis and renders 7-18 % slower than with 0.125; cubic cells (1) cost the plain dungeon 2-4 %.
// 0.125 is within 1.4 % of the best of {0, 0.125, 0.25, 0.5, 1} on average (0: 5.4 %, 1: 3.0 %). ST_LBVH_CELL_ASPECT in the environment overrides it (measurements).
constexpr float kLbvhCellAspect = 0.125f;
size_t lbvh_sort_temp_bytes(uint32_t slots);
uint32_t lbvh_pow2(uint32_t n);
void lbvh_warm(int* bounds, hipStream_t stream);   // before the first build: the builder's code object on the device (bounds: the six ints of LbvhArgs::bounds)
int lbvh_build(const LbvhArgs& args, hipStream_t stream);   // 0, or negative: -1 fewer than two live triangles, -2 the sort failed, -3 a launch failed
// After a build with the same arguments whose scratch (keys_out, children) and output are untouched: the triangles moved (tri_geo / tri_bounds changed in
// place, the same slots live): leaf records and every box again, the topology as it was.
int lbvh_refit(const LbvhArgs& args, hipStream_t stream);

}  // namespace st
