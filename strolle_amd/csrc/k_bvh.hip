// k_bvh.hip — BVH refit on the device (st_set_bvh_refresh(ST_BVH_REFIT_DEVICE); SURVEY.md section 8(f).2, the step before the
// path for scenes whose instances move every frame: strolle/src/bvh/builder.rs:183-354, examples/stress-bvh.rs).
// While instances only move, the tree's topology and leaf entries stay what they are: the host bakes the moved triangles
// and sends their hit-test records and bounds (80 B per triangle) instead of refitting the stream itself and re-sending all
// of it (64 B per entry: 17 MB for the 134 k-triangle dungeon); these two kernels bring the device stream up to date.
// min / max are exact in any order, so the result is bit for bit the host's backward sweep (st_engine.cpp refit_stream)
// and the oracle's refit (tests/test_gpu_parity.py test_bvh_refit_*).
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
// Bottom-up box refit: one thread per leaf run computes the run's box from its triangles' bounds, stores it in its parent's
// child slot (a node's box lives in its parent's entry; the root's is never stored) and climbs: the second thread to arrive
// at a node — a counter per node tells — finds both child boxes there, so the node's own box is their union, and carries on.
// parent[e] = (parent entry << 1) | which child slot, 0xffffffff for the root.
__global__ void k_bvh_refit(float4* bvh, const float4* tri_bounds, const uint32_t* parent, const uint32_t* runs, uint32_t n_runs, uint32_t* arrived) {
    const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n_runs) return;
    uint32_t node = runs[r];
    // Aabb::grow exactly as the host applies it (st_bvh.h: every point goes through min AND max, utils/bounding_box.rs:83-88),
    // so that even the sign of a zero comes out as in the host's sweep
    V3 lo = v3s(kF32Max), hi = v3s(-kF32Max);
    for (uint32_t k = node;; k++) {
        const float4 d0 = bvh[4u * k];
        const uint32_t tri = f2b(d0.y);
        const V3 blo = xyz(tri_bounds[2u * tri]), bhi = xyz(tri_bounds[2u * tri + 1u]);
        lo = vmin(lo, blo); hi = vmax(hi, blo); lo = vmin(lo, bhi); hi = vmax(hi, bhi);
        if (!(f2b(d0.x) & 1u)) break;
    }
    for (;;) {
        const uint32_t p = parent[node];
        if (p == 0xffffffffu) return;
        const uint32_t pe = p >> 1, slot = p & 1u;
        volatile float* dst = reinterpret_cast<volatile float*>(bvh + 4u * pe + 2u * slot);  // (min.xyz, w) (max.xyz, w): the w lanes keep their bits
        dst[0] = lo.x; dst[1] = lo.y; dst[2] = lo.z; dst[4] = hi.x; dst[5] = hi.y; dst[6] = hi.z;
        __threadfence();
        if (atomicAdd(&arrived[pe], 1u) == 0u) return;  // the sibling is still on its way: it will carry on from here
        __threadfence();
        const volatile float* both = reinterpret_cast<const volatile float*>(bvh + 4u * pe);
        const V3 lmin = v3(both[0], both[1], both[2]), lmax = v3(both[4], both[5], both[6]), rmin = v3(both[8], both[9], both[10]), rmax = v3(both[12], both[13], both[14]);
        lo = v3s(kF32Max); hi = v3s(-kF32Max);
        lo = vmin(lo, lmin); hi = vmax(hi, lmin); lo = vmin(lo, lmax); hi = vmax(hi, lmax);
        lo = vmin(lo, rmin); hi = vmax(hi, rmin); lo = vmin(lo, rmax); hi = vmax(hi, rmax);
        node = pe;
    }
}

void launch_bvh_patch_leaves(float4* bvh, const float4* tri_geo, const uint32_t* entry_of_tri, uint32_t lo, uint32_t hi, hipStream_t s) {
    if (hi > lo) ST_KLAUNCH(k_bvh_patch_leaves, dim3((hi - lo + 255u) / 256u), dim3(256), s, bvh, tri_geo, entry_of_tri, lo, hi);
}
void launch_bvh_refit(float4* bvh, const float4* tri_bounds, const uint32_t* parent, const uint32_t* runs, uint32_t n_runs, uint32_t* arrived, hipStream_t s) {
    if (n_runs) ST_KLAUNCH(k_bvh_refit, dim3((n_runs + 255u) / 256u), dim3(256), s, bvh, tri_bounds, parent, runs, n_runs, arrived);
}

}  // namespace ST_KNS
}  // namespace st
