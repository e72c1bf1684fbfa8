// st_bvh_refresh.cpp — host engine of libstrolle_hip.so: device form of the BVH stream, host refit, the device refit's index arrays, stack-depth check. See st_engine.h.
#include "st_engine.h"

namespace st {

void Engine::measure_stack_need() {
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
    // A ray's pending stack never holds more entries than the deepest chain of internal nodes (every node on the path pushes at most its far
    // child): up to kBvhStackSize (24, the reference's) the launches keep the reference's stack, up to kBvhStackSizeDeep (32) they take the
    // deep one — more LDS per block, no dropped push. Beyond that a tree is reported through st_tick's status (ST_ERR_BVH_TOO_DEEP, once per
    // build) — or, with StTuning::allow_deep_bvh, once on stderr.
    stack_entries = std::min<uint32_t>(std::max<uint32_t>(deepest, (uint32_t)kBvhStackSize), (uint32_t)kBvhStackSizeDeep);   // exactly what the tree needs: LDS per block decides occupancy
    bvh_too_deep_unreported = deepest > stack_entries;
    if (bvh_too_deep_unreported && tuning.allow_deep_bvh && !bvh_depth_warned) {
        bvh_depth_warned = true;
        fprintf(stderr, "[strolle-hip] warning: the BVH is %u internal nodes deep; traversal keeps %d pending entries per ray (as the reference does) and drops deeper ones — distant geometry may be missed. st_debug_bvh_depth reports this.\n", deepest, (int)stack_entries);
    }
}

void Engine::expand_stream() {
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
    device_root_is_leaf = entries != 0 && f2b(bvh_upload_[0].w) != 0u;
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
void Engine::index_device_tree() {
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

// The WIDE stream's topology (k_bvh.hip k_bvh_wide; st_device.h closest_hit_wide): the binary tree of the device stream (bvh_upload_,
// entries of four texels) collapsed top-down into nodes of up to four children — a node's two children are replaced by their own children,
// largest surface area first, until there are four or only leaf runs are left (tools/bvh4_sim.py: 3.1 children per node on the dungeon,
// half the node steps per ray). Wide nodes are numbered in depth-first order, leaf records in stream order (a run's records stay
// consecutive). Only WHICH boxes a node holds is decided here; the boxes themselves are read on the device, from the device's stream.
void Engine::build_wide_topology() {
    const uint32_t n_entries = device_bvh_len / 4u;
    wide_topo_.clear(); wide_leaf_entry_.clear(); wide_root_ = 0u; wide_stack_need_ = 0u; wide_serial_++;
    if (!n_entries) return;
    auto internal = [&](uint32_t k) { return f2b(bvh_upload_[4u * (size_t)k].w) == 0u; };
    auto far_child = [&](uint32_t k) { return f2b(bvh_upload_[4u * (size_t)k + 1u].w) / 64u; };
    auto area = [&](uint32_t src) {   // of the box stored at (entry << 1 | slot)
        const float4 lo = bvh_upload_[4u * (size_t)(src >> 1) + 2u * (src & 1u)], hi = bvh_upload_[4u * (size_t)(src >> 1) + 2u * (src & 1u) + 1u];
        const float dx = std::max(hi.x - lo.x, 0.0f), dy = std::max(hi.y - lo.y, 0.0f), dz = std::max(hi.z - lo.z, 0.0f);
        return dx * dy + dy * dz + dz * dx;
    };
    std::vector<uint32_t> leaf_index(n_entries, 0u);
    for (uint32_t k = 0; k < n_entries; k++) if (!internal(k)) { leaf_index[k] = (uint32_t)wide_leaf_entry_.size(); wide_leaf_entry_.push_back(k); }
    if (!internal(0u)) { wide_root_ = 1u; return; }   // the whole tree is one leaf run: leaf record 0
    struct Child { uint32_t src, entry; };
    std::vector<uint32_t> head;             // binary entry that heads wide node i
    std::vector<Child> kids;                // 4 per node (src = ~0: empty)
    std::vector<uint32_t> node_of(n_entries, 0xffffffffu);
    std::vector<uint32_t> todo{0u};
    while (!todo.empty()) {
        const uint32_t k = todo.back(); todo.pop_back();
        node_of[k] = (uint32_t)head.size(); head.push_back(k);
        Child ch[4]; int n = 2;
        ch[0] = {(k << 1) | 0u, k + 1u}; ch[1] = {(k << 1) | 1u, far_child(k)};
        while (n < 4) {
            int pick = -1; float best = -1.0f;
            for (int i = 0; i < n; i++) if (internal(ch[i].entry)) { const float ar = area(ch[i].src); if (ar > best) { best = ar; pick = i; } }
            if (pick < 0) break;
            const uint32_t e = ch[pick].entry;
            for (int i = n; i > pick + 1; i--) ch[i] = ch[i - 1];   // the two grandchildren take the child's place, in order
            ch[pick] = {(e << 1) | 0u, e + 1u}; ch[pick + 1] = {(e << 1) | 1u, far_child(e)};
            n++;
        }
        for (int i = 0; i < 4; i++) kids.push_back(i < n ? ch[i] : Child{0xffffffffu, 0u});
        for (int i = n - 1; i >= 0; i--) if (internal(ch[i].entry)) todo.push_back(ch[i].entry);   // depth first: the first child's subtree follows its parent
    }
    // the most entries a walk over the wide nodes can have pending: descending into one of a node's n children leaves at most n - 1 behind
    // (children sit behind their parent in `head`, so a backward sweep sees finished children)
    {
        std::vector<uint32_t> need(head.size(), 0u);
        for (size_t i = head.size(); i-- > 0;) {
            uint32_t n = 0, below = 0;
            for (int c = 0; c < 4; c++) {
                const Child& q = kids[4u * i + c];
                if (q.src == 0xffffffffu) continue;
                n++;
                if (internal(q.entry)) below = std::max(below, need[node_of[q.entry]]);
            }
            need[i] = (n ? n - 1u : 0u) + below;
        }
        wide_stack_need_ = head.empty() ? 0u : need[0];
    }
    wide_topo_.resize(8u * head.size());
    for (size_t i = 0; i < head.size(); i++)
        for (int c = 0; c < 4; c++) {
            const Child& q = kids[4u * i + c];
            wide_topo_[8u * i + c] = q.src;
            wide_topo_[8u * i + 4u + c] = q.src == 0xffffffffu ? 0u : (internal(q.entry) ? (node_of[q.entry] << 1) : ((leaf_index[q.entry] << 1) | 1u));
        }
    wide_root_ = 0u;
}

void Engine::mark_internal_starts() {
    internal_start_.assign(bvh_stream.size(), 0);
    for (size_t p = 0; p < bvh_stream.size();) {
        if (f2b(bvh_stream[p].w) == 0u) { internal_start_[p] = 1; p += 4; } else p += 1;
    }
}

// ---- BVH refit (the alternative to a rebuild when instances only move)
// What the leaves of the current tree refer to: every live (triangle slot, material) pair, plus the Blend flags baked into
// the leaf entries. Equal signatures mean the stream's topology and leaf entries are still right; only boxes moved.
uint64_t Engine::topology_of(const std::vector<uint8_t>& blend) const {
    uint64_t h = 0x9e3779b97f4a7c15ull;
    auto mix = [&h](uint64_t v) { h = (h ^ v) * 0x100000001b3ull; h ^= h >> 29; };
    for (size_t i = 0; i < prims.size(); i++) if (prim_alive[i]) mix(((uint64_t)i << 32) | prims[i].material_id);
    mix(0xffffffffffffffffull);
    for (uint8_t b : blend) mix(b);
    return h;
}

// offsets of the internal nodes of bvh_stream, in stream order (serializer.rs:20-110: a node is internal when d0.w == 0)
void Engine::index_stream() {
    internal_positions.clear();
    for (size_t p = 0; p < bvh_stream.size();) {
        if (f2b(bvh_stream[p].w) == 0u) { internal_positions.push_back((uint32_t)p); p += 4; }
        else p += 1;
    }
}

// box of the subtree that starts at stream offset p: a run of leaf entries (triangle bounds as baked) or an internal node
// (union of the two child boxes it stores)
Aabb Engine::subtree_box(size_t p) const {
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

void Engine::refit_node(size_t p) {
    const Aabb l = subtree_box(p + 4), r = subtree_box(f2b(bvh_stream[p + 1].w));
    bvh_stream[p] = make_float4(l.lo.x, l.lo.y, l.lo.z, bvh_stream[p].w);
    bvh_stream[p + 1] = make_float4(l.hi.x, l.hi.y, l.hi.z, bvh_stream[p + 1].w);
    bvh_stream[p + 2] = make_float4(r.lo.x, r.lo.y, r.lo.z, bvh_stream[p + 2].w);
    bvh_stream[p + 3] = make_float4(r.hi.x, r.hi.y, r.hi.z, bvh_stream[p + 3].w);
}

// internal nodes whose offsets lie in [begin, end), last to first: children sit behind their parent in the stream, so a
// backward sweep sees finished children
void Engine::refit_span(size_t begin, size_t end) {
    const auto lo = std::lower_bound(internal_positions.begin(), internal_positions.end(), (uint32_t)begin);
    auto hi = std::lower_bound(internal_positions.begin(), internal_positions.end(), (uint32_t)end);
    while (hi != lo) refit_node(*--hi);
}

}  // namespace st
