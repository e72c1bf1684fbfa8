// st_bvh.h — host side of the scene -> device-buffer step: world-space triangle baking, the binned-SAH
// BVH build and its flattening into the float4 stream the traversal kernel walks.
//
// Contract (tree shape and stream layout must equal the reference's so that `used_memory` and hit order
// are identical): strolle/src/bvh/builder.rs (12 centroid bins per axis, sweep costs, `<=` tie-break,
// swap-to-back partition, breadth-first processing) and strolle/src/bvh/serializer.rs (DFS pre-order,
// internal = 4 float4 holding both children's bounds + right pointer, leaf entry = 1 float4).
// Implementation notes: iterative builder over an index-free primitive array, subtrees built concurrently for large
// scenes (the result does not depend on the schedule); subtree reuse by hash (builder.rs:205-301) is not implemented —
// every refresh is a fresh build.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

#include "st_types.h"

namespace st {

struct Aabb {
    V3 lo, hi;
    Aabb() { lo = v3s(kF32Max); hi = v3s(-kF32Max); }
    V3 extent() const { return hi - lo; }
    float half_area() const { const V3 e = extent(); return e.x * e.y + e.y * e.z + e.z * e.x; }
    bool is_set() const { return lo.x != kF32Max; }
    void grow(V3 p) { lo = vmin(lo, p); hi = vmax(hi, p); }
    void grow(const Aabb& o) { grow(o.lo); grow(o.hi); }  // utils/bounding_box.rs:83-88: min then max, literally
};

struct BuildPrim { uint32_t triangle_id, material_id; V3 center; Aabb bounds; };

inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

class BvhBuild {
  public:
    struct Node { bool internal = false; Aabb bounds; uint32_t begin = 0, end = 0, left = 0, right = 0; };
    std::vector<Node> nodes;
    std::vector<BuildPrim> prims;

    // Builds the tree over `prims` (reordered in place). The result — node contents, child links, primitive order — is a
    // function of the primitive array alone: a node's split reads only its own [begin, end) range, which its parent's
    // partition fixed, so subtrees can be built in any order or concurrently. Large scenes therefore fan the subtrees
    // out over a few worker threads (the reference builds on a scoped thread too, builder.rs:183-203); only the indices
    // nodes receive in `nodes` depend on the schedule, and nothing downstream looks at them (flatten follows the links).
    void run(unsigned max_threads = 0) {
        const uint32_t n = (uint32_t)prims.size();
        nodes.clear();
        nodes.resize(n > 0 ? 2u * (size_t)n + 1u : 1u);  // a split makes two non-empty children, so <= 2n - 1 nodes
        nodes[0] = Node(); nodes[0].begin = 0; nodes[0].end = n;
        next_node.store(1u);
        overflow.store(false);
        unsigned threads = max_threads ? max_threads : std::thread::hardware_concurrency();
        if (threads > 16u) threads = 16u;
        if (n < kParallelMin || threads < 2u) { std::vector<uint32_t> stack{0u}; build_subtree(stack, nullptr); }
        else run_parallel(threads);
        if (overflow.load()) {  // cannot happen for finite inputs; keep the sequential answer if it ever does
            nodes.clear(); nodes.resize(1); nodes.reserve(4u * (size_t)n + 16u);
            nodes[0] = Node(); nodes[0].begin = 0; nodes[0].end = n;
            run_sequential_growing();
            return;
        }
        nodes.resize(next_node.load());
    }

    // DFS flatten. `blend[m]` != 0 marks AlphaMode::Blend materials (leaf flag bit 1).
    void flatten(const std::vector<uint8_t>& blend, std::vector<float4>& out) const {
        out.clear();
        if (prims.empty()) return;
        emit(0, blend, out);
    }

  private:
    static constexpr int kBins = 12;
    static constexpr uint32_t kParallelMin = 4096;  // below this a build takes < 3 ms and threads cost more than they save
#ifndef ST_BVH_SPAWN_MIN
#define ST_BVH_SPAWN_MIN 256
#endif
    static constexpr uint32_t kSpawnMin = ST_BVH_SPAWN_MIN;  // subtrees at least this large are offered to other workers
    // on their own cache line: workers bump the counter constantly, and the vector headers above are read on every access
    struct alignas(64) Counters { std::atomic<uint32_t> next{1}; std::atomic<bool> overflow{false}; };
    Counters counters_;
    std::atomic<uint32_t>& next_node = counters_.next;
    std::atomic<bool>& overflow = counters_.overflow;

    // Splits node `id` if the SAH says so (builder.rs:60-181). Returns true and the two children when it did.
    bool split(uint32_t id, uint32_t* left, uint32_t* right) {
        int axis; float split_at, split_cost;
        if (!best_plane(nodes[id], &axis, &split_at, &split_cost)) return false;
        const float leaf_cost = (float)(nodes[id].end - nodes[id].begin) * nodes[id].bounds.half_area();
        if (!(split_cost < leaf_cost)) return false;
        const uint32_t begin = nodes[id].begin, end = nodes[id].end;
        int64_t i = 0, j = (int64_t)(end - begin) - 1;
        Aabb lb, rb;
        BuildPrim* p = prims.data() + begin;
        while (i <= j) {
            const BuildPrim cur = p[i];
            if (axis_of(cur.center, axis) < split_at) { lb.grow(cur.bounds); i++; }
            else { const BuildPrim t = p[i]; p[i] = p[j]; p[j] = t; rb.grow(cur.bounds); j--; }
        }
        const uint32_t li = next_node.fetch_add(2u);
        if ((size_t)li + 2u > nodes.size()) { overflow.store(true); return false; }
        const uint32_t ri = li + 1u;
        Node l, r;
        l.bounds = lb; l.begin = begin; l.end = begin + (uint32_t)i;
        r.bounds = rb; r.begin = begin + (uint32_t)i; r.end = end;
        nodes[li] = l; nodes[ri] = r;
        nodes[id].internal = true; nodes[id].left = li; nodes[id].right = ri;
        *left = li; *right = ri;
        return true;
    }

    struct Shared { std::mutex m; std::condition_variable cv; std::vector<uint32_t> queue; uint32_t pending = 0; };

    // Builds every node reachable from `stack`; with `shared`, large children are handed to the common queue instead.
    void build_subtree(std::vector<uint32_t>& stack, Shared* shared) {
        while (!stack.empty()) {
            const uint32_t id = stack.back();
            stack.pop_back();
            uint32_t c[2];
            if (!split(id, &c[0], &c[1])) continue;
            for (int k = 0; k < 2; k++) {
                if (shared && nodes[c[k]].end - nodes[c[k]].begin >= kSpawnMin) {
                    { std::lock_guard<std::mutex> lock(shared->m); shared->queue.push_back(c[k]); shared->pending++; }
                    shared->cv.notify_one();
                } else stack.push_back(c[k]);
            }
        }
    }

    void run_parallel(unsigned threads) {
        Shared shared;
        shared.queue.push_back(0u); shared.pending = 1;
        auto worker = [&] {
            std::vector<uint32_t> stack;
            for (;;) {
                uint32_t id;
                {
                    std::unique_lock<std::mutex> lock(shared.m);
                    shared.cv.wait(lock, [&] { return !shared.queue.empty() || shared.pending == 0; });
                    if (shared.queue.empty()) return;  // pending == 0: the tree is complete
                    id = shared.queue.back(); shared.queue.pop_back();
                }
                stack.assign(1, id);
                build_subtree(stack, &shared);
                bool done;
                { std::lock_guard<std::mutex> lock(shared.m); shared.pending--; done = shared.pending == 0; }
                if (done) shared.cv.notify_all();
            }
        };
        std::vector<std::thread> pool;
        for (unsigned t = 1; t < threads; t++) pool.emplace_back(worker);
        worker();
        for (auto& t : pool) t.join();
    }

    // the original breadth-first loop with a growing node vector (only reached through the overflow guard)
    void run_sequential_growing() {
        std::deque<uint32_t> todo;
        todo.push_back(0);
        while (!todo.empty()) {
            const uint32_t id = todo.front();
            todo.pop_front();
            int axis; float split_at, split_cost;
            if (!best_plane(nodes[id], &axis, &split_at, &split_cost)) continue;
            const float leaf_cost = (float)(nodes[id].end - nodes[id].begin) * nodes[id].bounds.half_area();
            if (!(split_cost < leaf_cost)) continue;
            const uint32_t begin = nodes[id].begin, end = nodes[id].end;
            int64_t i = 0, j = (int64_t)(end - begin) - 1;
            Aabb lb, rb;
            BuildPrim* p = prims.data() + begin;
            while (i <= j) {
                const BuildPrim cur = p[i];
                if (axis_of(cur.center, axis) < split_at) { lb.grow(cur.bounds); i++; }
                else { const BuildPrim t = p[i]; p[i] = p[j]; p[j] = t; rb.grow(cur.bounds); j--; }
            }
            Node l, r;
            l.bounds = lb; l.begin = begin; l.end = begin + (uint32_t)i;
            r.bounds = rb; r.begin = begin + (uint32_t)i; r.end = end;
            const uint32_t li = (uint32_t)nodes.size(); nodes.push_back(l);
            const uint32_t ri = (uint32_t)nodes.size(); nodes.push_back(r);
            nodes[id].internal = true; nodes[id].left = li; nodes[id].right = ri;
            todo.push_back(li); todo.push_back(ri);
        }
    }

    bool best_plane(const Node& node, int* out_axis, float* out_at, float* out_cost) const {
        const uint32_t n = node.end - node.begin;
        if (n <= 1) return false;
        const BuildPrim* p = prims.data() + node.begin;
        Aabb cb;
        for (uint32_t i = 0; i < n; i++) cb.grow(p[i].center);
        Aabb bin_bounds[3][kBins]; uint32_t bin_count[3][kBins] = {};
        const V3 scale = (float)kBins / cb.extent();
        for (uint32_t i = 0; i < n; i++) {
            const V3 f = scale * (p[i].center - cb.lo);
            const uint32_t b[3] = {f2u_sat(f.x), f2u_sat(f.y), f2u_sat(f.z)};
            for (int a = 0; a < 3; a++) {
                const uint32_t k = b[a] < (uint32_t)kBins - 1u ? b[a] : (uint32_t)kBins - 1u;
                bin_count[a][k] += 1; bin_bounds[a][k].grow(p[i].bounds);
            }
        }
        float la[3][kBins - 1], ra[3][kBins - 1]; uint32_t lc[3][kBins - 1], rc[3][kBins - 1];
        for (int a = 0; a < 3; a++) {
            Aabb lbb, rbb; uint32_t lcount = 0, rcount = 0;
            for (int i = 0; i < kBins - 1; i++) {
                lcount += bin_count[a][i]; lc[a][i] = lcount;
                if (bin_bounds[a][i].is_set()) lbb.grow(bin_bounds[a][i]);
                la[a][i] = lbb.half_area();
                rcount += bin_count[a][kBins - 1 - i]; rc[a][kBins - 2 - i] = rcount;
                if (bin_bounds[a][kBins - 1 - i].is_set()) rbb.grow(bin_bounds[a][kBins - 1 - i]);
                ra[a][kBins - 2 - i] = rbb.half_area();
            }
        }
        bool have = false; float best_cost = 0.0f; int best_axis = 0; float best_at = 0.0f;
        const V3 width = cb.extent() / (float)kBins;
        for (int a = 0; a < 3; a++)
            for (int i = 0; i < kBins - 1; i++) {
                const float cost = (float)lc[a][i] * la[a][i] + (float)rc[a][i] * ra[a][i];
                if (!have || cost <= best_cost) {
                    have = true; best_cost = cost; best_axis = a;
                    best_at = axis_of(cb.lo, a) + axis_of(width, a) * (float)(i + 1);
                }
            }
        *out_axis = best_axis; *out_at = best_at; *out_cost = best_cost;
        return have;
    }

    uint32_t emit(uint32_t id, const std::vector<uint8_t>& blend, std::vector<float4>& out) const {
        const uint32_t at = (uint32_t)out.size();
        const Node& n = nodes[id];
        if (n.internal) {
            out.resize(out.size() + 4, make_float4(0, 0, 0, 0));
            const Aabb lb = nodes[n.left].bounds, rb = nodes[n.right].bounds;
            emit(n.left, blend, out);
            const uint32_t right_at = emit(n.right, blend, out);
            out[at] = make_float4(lb.lo.x, lb.lo.y, lb.lo.z, b2f(0u));
            out[at + 1] = make_float4(lb.hi.x, lb.hi.y, lb.hi.z, b2f(right_at));
            out[at + 2] = make_float4(rb.lo.x, rb.lo.y, rb.lo.z, 0.0f);
            out[at + 3] = make_float4(rb.hi.x, rb.hi.y, rb.hi.z, 0.0f);
        } else {
            const uint32_t n_entries = n.end - n.begin;
            for (uint32_t i = 0; i < n_entries; i++) {
                const BuildPrim& p = prims[n.begin + i];
                const uint32_t more = i + 1 < n_entries ? 1u : 0u;
                const uint32_t is_blend = (p.material_id < blend.size() && blend[p.material_id]) ? 2u : 0u;
                out.push_back(make_float4(b2f(more | is_blend), b2f(p.triangle_id), b2f(p.material_id), b2f(1u)));
            }
        }
        return at;
    }
};

}  // namespace st
