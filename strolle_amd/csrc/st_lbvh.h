// st_lbvh.h — interface of the device BVH builder (k_lbvh.hip): built once into the library, whatever arithmetic the frames use.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace st {

// Everything is device memory the caller owns. tri_info: one word per triangle slot — bit 0 the slot is live, bit 1 its material is
// AlphaMode::Blend, bits 2.. its material slot. Output: `nodes` (64 B per node, node index = the head's index in the binary radix tree:
// live - 1 slots, sparsely used; the root is node 0) and `leaves` (48 B per record, sorted order), in the wide stream's format
// (st_device.h closest_hit_wide). Scratch: keys_in / keys_out (8 B x slots), sort_temp (lbvh_sort_temp_bytes(slots)), seg
// (2 x lbvh_pow2(live) boxes of 32 B), children (8 B x live), node_box (32 B x live), frontier_a / frontier_b (4 B x live), bounds (6 ints),
// counters (4 words: the collapse's frontier counts, rotating by launch; [3] != 0: the finishing launch ran out of its private stack and LEFT SUBTREES
// UNBUILT — `flags_host`, if not null, is a page-locked word the same launch sets then: the caller must not use the tree).
// The sort (round 6): 31-bit keys — the 30-bit Morton code, 0x40000000 for a dead slot — with the slot as value (hipCUB SortPairs); the sort is stable and the
// input is in slot order, so the order is the one (code << 32 | slot) gives, and those 64-bit keys are composed afterwards for the hierarchy's tie-breaks: the
// tree is bit for bit the one round 5's 64-bit sort produced.
struct LbvhArgs {
    const float4* tri_geo; const float4* tri_bounds; const uint32_t* tri_info;
    uint32_t slots, live, links16;
    float4* nodes; float4* leaves;
    unsigned long long* keys_in; unsigned long long* keys_out; void* sort_temp; size_t sort_temp_bytes;
    float4* seg; uint2* children; float4* node_box; uint32_t* frontier_a; uint32_t* frontier_b; int* bounds; uint32_t* counters;
    uint32_t* flags_host;   // nullptr, or device-visible host memory: [0] |= 1 when the finishing launch dropped a subtree
};
size_t lbvh_sort_temp_bytes(uint32_t slots);
uint32_t lbvh_pow2(uint32_t n);
int lbvh_build(const LbvhArgs& args, hipStream_t stream);   // 0, or negative: -1 fewer than two live triangles, -2 the sort failed, -3 a launch failed
// After a build with the same arguments whose scratch (keys_out, children) and output are untouched: the triangles moved (tri_geo / tri_bounds changed in
// place, the same slots live): leaf records and every box again, the topology as it was.
int lbvh_refit(const LbvhArgs& args, hipStream_t stream);

}  // namespace st
