// st_bvh.h — host side of the scene -> device-buffer step: world-space triangle baking, the binned-SAH
// BVH build and its flattening into the float4 stream the traversal kernel walks.
//
// Contract (tree shape and stream layout must equal the reference's so that `used_memory` and hit order
// are identical): strolle/src/bvh/builder.rs (12 centroid bins per axis, sweep costs, `<=` tie-break,
// swap-to-back partition, breadth-first processing) and strolle/src/bvh/serializer.rs (DFS pre-order,
// internal = 4 float4 holding both children's bounds + right pointer, leaf entry = 1 float4).
// Implementation notes: iterative builder over an index-free primitive array, subtrees built concurrently for large
// scenes (the result does not depend on the schedule). A refresh reuses unchanged subtrees of the previous tree the way
// builder.rs:183-301 does — an order-sensitive hash of each child's primitives, taken while partitioning, is compared with
// the previous tree's child and a match copies the subtree — with one guarantee added: the refreshed tree is the tree a
// from-scratch build of the same primitives gives (BvhBuild::begin_refresh / run, below).
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <utility>
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

struct BuildPrim { uint32_t triangle_id, material_id; V3 center; Aabb bounds; uint64_t key = 0; };  // key: hash of the other fields (BvhBuild::run)

inline float axis_of(V3 v, int a) { return a == 0 ? v.x : (a == 1 ? v.y : v.z); }

// A minimal fork-join pool for the builder: a LIFO queue of closures, `threads - 1` workers plus the calling thread.
class TaskPool {
  public:
    explicit TaskPool(unsigned threads) { for (unsigned t = 1; t < threads; t++) workers_.emplace_back([this] { work(false); }); }
    ~TaskPool() { finish(); }
    void push(std::function<void()> f) {
        { std::lock_guard<std::mutex> lock(m_); queue_.push_back(std::move(f)); pending_++; }
        cv_.notify_one();
    }
    // the calling thread works until every task (including the ones tasks pushed) is done, then the workers are joined
    void finish() {
        if (joined_) return;
        { std::lock_guard<std::mutex> lock(m_); closing_ = true; }
        cv_.notify_all();
        work(false);
        for (auto& t : workers_) t.join();
        joined_ = true;
    }
    // fork-join inside a task: fn(0 .. n-1), chunk 0 on the caller, the rest offered to the pool; the caller helps with
    // whatever is queued while it waits, so nested use cannot deadlock
    void parallel_for(uint32_t n, const std::function<void(uint32_t)>& fn) {
        if (n == 0) return;
        std::atomic<uint32_t> remaining(n - 1u);
        for (uint32_t c = 1; c < n; c++) push([&fn, &remaining, c] { fn(c); remaining.fetch_sub(1u); });
        fn(0u);
        while (remaining.load() > 0u) { if (!run_one()) std::this_thread::yield(); }
    }

  private:
    bool run_one() {
        std::function<void()> f;
        { std::lock_guard<std::mutex> lock(m_); if (queue_.empty()) return false; f = std::move(queue_.back()); queue_.pop_back(); }
        f();
        bool done;
        { std::lock_guard<std::mutex> lock(m_); pending_--; done = pending_ == 0; }
        if (done) cv_.notify_all();
        return true;
    }
    void work(bool) {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lock(m_);
                cv_.wait(lock, [this] { return !queue_.empty() || (closing_ && pending_ == 0); });
                if (queue_.empty()) return;  // closing and pending == 0: nothing left and nothing running
                f = std::move(queue_.back()); queue_.pop_back();
            }
            f();
            bool done;
            { std::lock_guard<std::mutex> lock(m_); pending_--; done = pending_ == 0; }
            if (done) cv_.notify_all();
        }
    }
    std::mutex m_; std::condition_variable cv_;
    std::vector<std::function<void()>> queue_; uint32_t pending_ = 0;
    std::vector<std::thread> workers_; bool joined_ = false, closing_ = false;
};

class BvhBuild {
  public:
    // lhash / rhash: order-sensitive hashes of the primitives the split sent left / right, in the order the partition examined
    // them (as builder.rs:205-232 hashes them) — equal hashes mean the child received the same primitives in the same order,
    // hence would be rebuilt into the same subtree. count: nodes in this subtree (set by finish_counts()).
    struct Node { bool internal = false; Aabb bounds; uint32_t begin = 0, end = 0, left = 0, right = 0; uint64_t lhash = 0, rhash = 0; uint32_t count = 1; };
    std::vector<Node> nodes;
    std::vector<BuildPrim> prims;

    // Call before refilling `prims` for the next build: keeps the finished tree and its primitive order so that subtrees whose
    // input did not change are copied instead of rebuilt (builder.rs:233-301). The outcome equals a fresh build.
    void begin_refresh() { prev_nodes_.swap(nodes); prev_prims_.swap(prims); nodes.clear(); prims.clear(); }
    size_t reused_primitives() const { return reused_.load(); }

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
        reused_.store(0);
        for (uint32_t i = 0; i < n; i++) prims[i].key = prim_key(prims[i]);
        const uint32_t root_ghost = (!prev_nodes_.empty() && !prev_prims_.empty() && prev_nodes_[0].internal) ? 0u : kNoGhost;
        unsigned threads = max_threads ? max_threads : std::thread::hardware_concurrency();
        if (threads > 16u) threads = 16u;
        threads_ = (n < kParallelMin || threads < 2u) ? 1u : threads;
        spawn_min_ = n / (threads_ * 8u) > kSpawnMin ? n / (threads_ * 8u) : kSpawnMin;  // ~8 tasks per thread and level at the top
        if (threads_ == 1u) { std::vector<Work> stack{{0u, root_ghost}}; build_subtree(stack, nullptr); }
        else {
            TaskPool pool(threads_);
            pool.push([this, &pool, root_ghost] { std::vector<Work> stack{{0u, root_ghost}}; build_subtree(stack, &pool); });
            pool.finish();
        }
        if (overflow.load()) {  // cannot happen for finite inputs; keep the sequential answer if it ever does
            nodes.clear(); nodes.resize(1); nodes.reserve(4u * (size_t)n + 16u);
            nodes[0] = Node(); nodes[0].begin = 0; nodes[0].end = n;
            run_sequential_growing();
            finish_counts();
            return;
        }
        nodes.resize(next_node.load());
        finish_counts();
    }

    // DFS flatten. `blend[m]` != 0 marks AlphaMode::Blend materials (leaf flag bit 1).
    // A node's children are always allocated after it (larger index), so one backward sweep yields every subtree's length
    // in the stream; with the offsets known, large subtrees are written concurrently into their disjoint ranges.
    void flatten(const std::vector<uint8_t>& blend, std::vector<float4>& out) const {
        out.clear();
        if (prims.empty()) return;
        std::vector<uint32_t> len(nodes.size());
        for (size_t i = nodes.size(); i-- > 0;) {
            const Node& n = nodes[i];
            len[i] = n.internal ? 4u + len[n.left] + len[n.right] : n.end - n.begin;
        }
        out.resize(len[0]);
        if (threads_ > 1u && prims.size() >= 16u * kParallelMin) {  // below ~64 k primitives one thread is done before a pool has started
            TaskPool pool(threads_);
            pool.push([&, this] { emit_at(0u, 0u, blend, len, out, &pool); });
            pool.finish();
        } else emit_at(0u, 0u, blend, len, out, nullptr);
    }

  private:
    static constexpr int kBins = 12;
    static constexpr uint32_t kParallelBinMin = 16384;  // nodes at least this large bin their primitives on all threads
    mutable unsigned threads_ = 1;
    std::vector<Node> prev_nodes_; std::vector<BuildPrim> prev_prims_;  // the previous build (begin_refresh)
    std::atomic<size_t> reused_{0};
    uint32_t spawn_min_ = kSpawnMin;
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
    static constexpr uint32_t kNoGhost = 0xffffffffu;
    struct Work { uint32_t id, ghost; };  // ghost: the node that stood at this place in the previous tree, if any
    static uint64_t prim_key(const BuildPrim& p) {
        uint64_t h = 1469598103934665603ull;
        const uint32_t w[11] = {p.triangle_id, p.material_id, f2b(p.center.x), f2b(p.center.y), f2b(p.center.z), f2b(p.bounds.lo.x), f2b(p.bounds.lo.y),
                                f2b(p.bounds.lo.z), f2b(p.bounds.hi.x), f2b(p.bounds.hi.y), f2b(p.bounds.hi.z)};
        for (uint32_t x : w) { h ^= x; h *= 1099511628211ull; }
        return h;
    }
    static uint64_t mix(uint64_t h, uint64_t key) { return (h ^ key) * 0x9e3779b97f4a7c15ull + 0x632be59bd9b4e019ull; }

    // Copies the previous tree's subtree `old_id` (and the final order of its primitives) to the range starting at `begin`.
    // The copy is laid out in pre-order — a node, its left subtree, its right subtree — so every node's place follows from
    // the subtree sizes alone and large subtrees can be copied by other workers at the same time.
    uint32_t copy_subtree(uint32_t old_id, uint32_t begin, TaskPool* pool) {
        const Node& old_root = prev_nodes_[old_id];
        const uint32_t cnt = old_root.count, n_prims = old_root.end - old_root.begin;
        const uint32_t base = next_node.fetch_add(cnt);
        if ((size_t)base + cnt > nodes.size()) { overflow.store(true); return kNoGhost; }
        reused_.fetch_add(n_prims);
        copy_nodes(old_id, base, (int64_t)begin - (int64_t)old_root.begin, pool);
        return base;
    }
    void copy_nodes(uint32_t old_id, uint32_t at, int64_t shift, TaskPool* pool) {
        std::vector<std::pair<uint32_t, uint32_t>> stack{{old_id, at}};  // (old node, its new index)
        bool first = true;
        while (!stack.empty()) {
            const uint32_t o = stack.back().first, nw = stack.back().second;
            stack.pop_back();
            const Node& src = prev_nodes_[o];
            if (pool && !first && src.end - src.begin >= spawn_min_) {
                pool->push([this, o, nw, shift, pool] { copy_nodes(o, nw, shift, pool); });
                continue;
            }
            first = false;
            Node c = src;
            c.begin = (uint32_t)((int64_t)c.begin + shift); c.end = (uint32_t)((int64_t)c.end + shift);
            if (c.internal) {
                const uint32_t l = nw + 1u, r = l + prev_nodes_[src.left].count;
                stack.push_back({src.right, r}); stack.push_back({src.left, l});
                c.left = l; c.right = r;
            } else {
                for (uint32_t i = src.begin; i < src.end; i++) prims[(uint32_t)((int64_t)i + shift)] = prev_prims_[i];
            }
            nodes[nw] = c;
        }
    }

    // Splits node `w.id` if the SAH says so (builder.rs:60-181). Children whose primitives (and their order) equal those of the
    // previous tree's child at the same place are copied from it; the others are returned as further work.
    int split(Work w, Work* out, TaskPool* pool) {
        const uint32_t id = w.id;
        int axis; float split_at, split_cost;
        if (!best_plane(nodes[id], &axis, &split_at, &split_cost, pool)) return 0;
        const float leaf_cost = (float)(nodes[id].end - nodes[id].begin) * nodes[id].bounds.half_area();
        if (!(split_cost < leaf_cost)) return 0;
        const uint32_t begin = nodes[id].begin, end = nodes[id].end;
        int64_t i = 0, j = (int64_t)(end - begin) - 1;
        Aabb lb, rb;
        uint64_t lh = 0x243f6a8885a308d3ull, rh = 0x13198a2e03707344ull;
        BuildPrim* p = prims.data() + begin;
        while (i <= j) {
            const BuildPrim cur = p[i];
            if (axis_of(cur.center, axis) < split_at) { lb.grow(cur.bounds); lh = mix(lh, cur.key); i++; }
            else { const BuildPrim t = p[i]; p[i] = p[j]; p[j] = t; rb.grow(cur.bounds); rh = mix(rh, cur.key); j--; }
        }
        const uint32_t pivot = begin + (uint32_t)i;
        const Node* ghost = (w.ghost != kNoGhost && prev_nodes_[w.ghost].internal) ? &prev_nodes_[w.ghost] : nullptr;
        uint32_t child[2]; bool reused[2] = {false, false};
        const uint32_t cb[2] = {begin, pivot}, ce[2] = {pivot, end};
        const uint64_t hs[2] = {lh, rh};
        for (int k = 0; k < 2; k++) {
            if (!ghost) continue;
            const uint32_t g = k == 0 ? ghost->left : ghost->right;
            const uint64_t gh = k == 0 ? ghost->lhash : ghost->rhash;
            if (gh == hs[k] && prev_nodes_[g].end - prev_nodes_[g].begin == ce[k] - cb[k]) {
                const uint32_t at = copy_subtree(g, cb[k], pool);
                if (at == kNoGhost) return 0;
                child[k] = at; reused[k] = true;
            }
        }
        uint32_t fresh = 0;
        for (int k = 0; k < 2; k++) fresh += reused[k] ? 0u : 1u;
        uint32_t li = 0;
        if (fresh) {
            li = next_node.fetch_add(fresh);
            if ((size_t)li + fresh > nodes.size()) { overflow.store(true); return 0; }
        }
        int n_out = 0;
        for (int k = 0; k < 2; k++) {
            if (reused[k]) continue;
            Node c; c.bounds = k == 0 ? lb : rb; c.begin = cb[k]; c.end = ce[k];
            child[k] = li++;
            nodes[child[k]] = c;
            out[n_out++] = Work{child[k], ghost ? (k == 0 ? ghost->left : ghost->right) : kNoGhost};
        }
        Node& me = nodes[id];
        me.internal = true; me.left = child[0]; me.right = child[1]; me.lhash = lh; me.rhash = rh;
        return n_out;
    }

    // Builds every node reachable from `stack`; with a pool, large children become tasks of their own.
    void build_subtree(std::vector<Work>& stack, TaskPool* pool) {
        while (!stack.empty()) {
            const Work w = stack.back();
            stack.pop_back();
            Work c[2];
            const int n = split(w, c, pool);
            for (int k = 0; k < n; k++) {
                if (pool && nodes[c[k].id].end - nodes[c[k].id].begin >= spawn_min_) {
                    const Work child = c[k];
                    pool->push([this, pool, child] { std::vector<Work> st{child}; build_subtree(st, pool); });
                } else stack.push_back(c[k]);
            }
        }
    }
    // subtree sizes (children always have larger indices than their parent)
    void finish_counts() {
        for (size_t i = nodes.size(); i-- > 0;) {
            Node& n = nodes[i];
            n.count = n.internal ? 1u + nodes[n.left].count + nodes[n.right].count : 1u;
        }
    }

    // the original breadth-first loop with a growing node vector (only reached through the overflow guard)
    void run_sequential_growing() {
        std::deque<uint32_t> todo;
        todo.push_back(0);
        while (!todo.empty()) {
            const uint32_t id = todo.front();
            todo.pop_front();
            int axis; float split_at, split_cost;
            if (!best_plane(nodes[id], &axis, &split_at, &split_cost, nullptr)) continue;
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

    struct Bins { Aabb bounds[3][kBins]; uint32_t count[3][kBins]; Bins() { for (auto& a : count) for (auto& c : a) c = 0; } };
    static void bin_range(const BuildPrim* p, uint32_t n, const Aabb& cb, V3 scale, Bins& b) {
        for (uint32_t i = 0; i < n; i++) {
            const V3 f = scale * (p[i].center - cb.lo);
            const uint32_t idx[3] = {f2u_sat(f.x), f2u_sat(f.y), f2u_sat(f.z)};
            for (int a = 0; a < 3; a++) {
                const uint32_t k = idx[a] < (uint32_t)kBins - 1u ? idx[a] : (uint32_t)kBins - 1u;
                b.count[a][k] += 1; b.bounds[a][k].grow(p[i].bounds);
            }
        }
    }
    // Centroid bounds and bin contents are min / max / count reductions: exact in any order, so the few nodes at the top
    // of a large tree (which would otherwise be the serial critical path of the build) split them over the pool.
    bool best_plane(const Node& node, int* out_axis, float* out_at, float* out_cost, TaskPool* pool) const {
        const uint32_t n = node.end - node.begin;
        if (n <= 1) return false;
        const BuildPrim* p = prims.data() + node.begin;
        Aabb cb;
        Bins bins;
        const uint32_t chunks = (pool && n >= kParallelBinMin) ? threads_ : 1u;
        if (chunks > 1u) {
            const uint32_t per = (n + chunks - 1u) / chunks;
            std::vector<Aabb> part_cb(chunks);
            pool->parallel_for(chunks, [&](uint32_t c) {
                const uint32_t b = c * per, e = b + per < n ? b + per : n;
                for (uint32_t i = b; i < e; i++) part_cb[c].grow(p[i].center);
            });
            for (const Aabb& a : part_cb) if (a.is_set()) cb.grow(a);
            const V3 scale = (float)kBins / cb.extent();
            std::vector<Bins> part(chunks);
            pool->parallel_for(chunks, [&](uint32_t c) {
                const uint32_t b = c * per, e = b + per < n ? b + per : n;
                if (b < e) bin_range(p + b, e - b, cb, scale, part[c]);
            });
            for (const Bins& pb : part)
                for (int a = 0; a < 3; a++)
                    for (int k = 0; k < kBins; k++) { bins.count[a][k] += pb.count[a][k]; if (pb.bounds[a][k].is_set()) bins.bounds[a][k].grow(pb.bounds[a][k]); }
        } else {
            for (uint32_t i = 0; i < n; i++) cb.grow(p[i].center);
            bin_range(p, n, cb, (float)kBins / cb.extent(), bins);
        }
        Aabb (&bin_bounds)[3][kBins] = bins.bounds; uint32_t (&bin_count)[3][kBins] = bins.count;
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

    // writes the subtree of `id` at out[at ...]; explicit stack, right subtrees of large nodes become pool tasks
    void emit_at(uint32_t root, uint32_t root_at, const std::vector<uint8_t>& blend, const std::vector<uint32_t>& len, std::vector<float4>& out,
                 TaskPool* pool) const {
        std::vector<std::pair<uint32_t, uint32_t>> stack{{root, root_at}};
        while (!stack.empty()) {
            const uint32_t id = stack.back().first, at = stack.back().second;
            stack.pop_back();
            const Node& n = nodes[id];
            if (n.internal) {
                const Aabb lb = nodes[n.left].bounds, rb = nodes[n.right].bounds;
                const uint32_t left_at = at + 4u, right_at = left_at + len[n.left];
                out[at] = make_float4(lb.lo.x, lb.lo.y, lb.lo.z, b2f(0u));
                out[at + 1] = make_float4(lb.hi.x, lb.hi.y, lb.hi.z, b2f(right_at));
                out[at + 2] = make_float4(rb.lo.x, rb.lo.y, rb.lo.z, 0.0f);
                out[at + 3] = make_float4(rb.hi.x, rb.hi.y, rb.hi.z, 0.0f);
                if (pool && nodes[n.right].end - nodes[n.right].begin >= 4u * spawn_min_) {
                    const uint32_t r = n.right;
                    pool->push([this, r, right_at, &blend, &len, &out, pool] { emit_at(r, right_at, blend, len, out, pool); });
                } else stack.push_back({n.right, right_at});
                stack.push_back({n.left, left_at});
            } else {
                const uint32_t n_entries = n.end - n.begin;
                for (uint32_t i = 0; i < n_entries; i++) {
                    const BuildPrim& p = prims[n.begin + i];
                    const uint32_t more = i + 1 < n_entries ? 1u : 0u;
                    const uint32_t is_blend = (p.material_id < blend.size() && blend[p.material_id]) ? 2u : 0u;
                    out[at + i] = make_float4(b2f(more | is_blend), b2f(p.triangle_id), b2f(p.material_id), b2f(1u));
                }
            }
        }
    }
};

}  // namespace st
