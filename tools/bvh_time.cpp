#include <hip/hip_runtime.h>
#include "/root/repo/strolle_amd/csrc/st_math.h"
#include "/root/repo/strolle_amd/csrc/st_bvh.h"
#include <chrono>
#include <cstdio>
#include <random>
using namespace st;
int main(int argc, char** argv) {
    size_t n = argc > 1 ? atol(argv[1]) : 134288;
    std::mt19937 rng(1); std::uniform_real_distribution<float> u(-10, 10), s(-0.1f, 0.1f);
    BvhBuild b;
    b.prims.resize(n);
    for (size_t i = 0; i < n; i++) {
        V3 c = v3(u(rng), u(rng) * 0.1f, u(rng)); Aabb bb;
        for (int k = 0; k < 3; k++) bb.grow(c + v3(s(rng), s(rng), s(rng)));
        b.prims[i].triangle_id = (uint32_t)i; b.prims[i].material_id = 0; b.prims[i].center = (bb.lo + bb.hi) * 0.5f; b.prims[i].bounds = bb;
    }
    auto prims0 = b.prims;
    for (int rep = 0; rep < 3; rep++) {
        b.prims = prims0;
        auto t0 = std::chrono::steady_clock::now();
        b.run(argc > 2 ? atoi(argv[2]) : 0);
        auto t1 = std::chrono::steady_clock::now();
        std::vector<float4> out; std::vector<uint8_t> blend(1, 0);
        b.flatten(blend, out);
        auto t2 = std::chrono::steady_clock::now();
        unsigned long long h = 1469598103934665603ull;
        for (auto& f : out) { const uint32_t* w = (const uint32_t*)&f; for (int k = 0; k < 4; k++) { h ^= w[k]; h *= 1099511628211ull; } }
        printf("n=%zu build %.2f ms flatten %.2f ms nodes %zu stream %zu hash %016llx\n", n, std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t2 - t1).count(), b.nodes.size(), out.size(), h);
    }
    // refresh after a small edit: 1 % of the primitives nudged. The reusing builder must produce the stream of a fresh build.
    {
        auto edited = prims0;
        for (size_t i = 0; i < edited.size(); i += 100) { edited[i].center.x += 0.01f; edited[i].bounds.lo.x += 0.01f; edited[i].bounds.hi.x += 0.01f; }
        auto stream_hash = [](BvhBuild& bb) {
            std::vector<float4> out; std::vector<uint8_t> blend(1, 0);
            bb.flatten(blend, out);
            unsigned long long h = 1469598103934665603ull;
            for (auto& f : out) { const uint32_t* w = (const uint32_t*)&f; for (int k = 0; k < 4; k++) { h ^= w[k]; h *= 1099511628211ull; } }
            return h;
        };
        BvhBuild fresh; fresh.prims = edited; fresh.run(argc > 2 ? atoi(argv[2]) : 0);
        b.prims = prims0; b.run(argc > 2 ? atoi(argv[2]) : 0);   // the "previous frame"
        b.begin_refresh(); b.prims = edited;
        auto t0 = std::chrono::steady_clock::now();
        b.run(argc > 2 ? atoi(argv[2]) : 0);
        auto t1 = std::chrono::steady_clock::now();
        const unsigned long long hf = stream_hash(fresh), hr = stream_hash(b);
        printf("refresh after editing 1%%: %.2f ms, %zu of %zu primitives reused, stream %s a fresh build's\n",
               std::chrono::duration<double, std::milli>(t1 - t0).count(), b.reused_primitives(), n, hf == hr ? "EQUALS" : "DIFFERS FROM");
        // identical input: everything below the root is reused
        b.begin_refresh(); b.prims = edited;
        t0 = std::chrono::steady_clock::now(); b.run(argc > 2 ? atoi(argv[2]) : 0); t1 = std::chrono::steady_clock::now();
        printf("refresh with unchanged input: %.2f ms, %zu reused, stream %s\n", std::chrono::duration<double, std::milli>(t1 - t0).count(), b.reused_primitives(),
               stream_hash(b) == hf ? "EQUALS" : "DIFFERS");
    }
}
