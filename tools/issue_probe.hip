// issue_probe.hip — how many cycles does a SIMD of MI355X need per VALU / SALU instruction when 8 waves share it, and do
// the two kinds issue in parallel? (Traversal and resampling loops are full of exec-mask bookkeeping: s_and_saveexec,
// s_or_b64 exec, s_andn2 ... about one SALU instruction per VALU instruction.)
//
//   hipcc --offload-arch=gfx950 -O3 tools/issue_probe.hip -o /tmp/issue_probe && timeout 60 /tmp/issue_probe
//
// Every wave runs `iters` repetitions of a 16-slot block; the grid fills every SIMD with `waves` waves (1, 2, 4 or 8).
// Printed: cycles per block per SIMD at 2.4 GHz = elapsed * 2.4e9 / (iters * waves) — divide by the instruction counts.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define V4 "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5\n"
#define S4 "s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
#define VS4 "v_fma_f32 %0, %0, %4, %5\n s_add_u32 %6, %6, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 %7, %7, 1\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 %6, %6, 1\n v_fma_f32 %3, %3, %4, %5\n s_add_u32 %7, %7, 1\n"
#define VSS4 "v_fma_f32 %0, %0, %4, %5\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n v_fma_f32 %1, %1, %4, %5\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n v_fma_f32 %2, %2, %4, %5\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n v_fma_f32 %3, %3, %4, %5\n s_add_u32 %6, %6, 1\n s_add_u32 %7, %7, 1\n"
// the exec-mask pattern of a divergent if: compare, save exec, one VALU under the mask, restore
#define MASK4 "v_cmp_lt_f32 vcc, %0, %4\n s_and_saveexec_b64 %8, vcc\n v_fma_f32 %1, %1, %4, %5\n s_or_b64 exec, exec, %8\n v_cmp_lt_f32 vcc, %2, %4\n s_and_saveexec_b64 %8, vcc\n v_fma_f32 %3, %3, %4, %5\n s_or_b64 exec, exec, %8\n"
// packed f32: two lanes' worth of FMAs per instruction (what the SLP vectoriser makes of adjacent scalar f32 operations)
#define PK4 "v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2\n v_pk_fma_f32 %0, %0, %1, %2\n v_pk_fma_f32 %3, %3, %1, %2\n"
// dependent chain: each VALU needs the previous one's result
#define DEP4 "v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %0, %0, %4, %5\n"

template <int MODE>
__global__ void k_probe(float* out, int iters, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned s0 = 0, s1 = 0;
    unsigned long long m = 0;
    for (int i = 0; i < iters; i++) {
        if (MODE == 0) asm volatile(V4 V4 V4 V4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(s0), "s"(s1), "s"(m));
        if (MODE == 1) asm volatile(S4 S4 S4 S4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(s0), "s"(s1), "s"(m) : "scc");
        if (MODE == 2) asm volatile(VS4 VS4 VS4 VS4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(s0), "s"(s1), "s"(m) : "scc");
        if (MODE == 3) asm volatile(VSS4 VSS4 VSS4 VSS4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(s0), "s"(s1), "s"(m) : "scc");
        if (MODE == 4) asm volatile(MASK4 MASK4 MASK4 MASK4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(s0), "s"(s1), "s"(m) : "vcc", "scc");
        if (MODE == 6) { typedef float f2 __attribute__((ext_vector_type(2))); f2 p0 = {x0, x1}, p1 = {a, a}, p2 = {b, b}, p3 = {x2, x3};
                         asm volatile(PK4 PK4 PK4 PK4 : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3)); x0 = p0.x; x1 = p0.y; x2 = p3.x; x3 = p3.y; }
        if (MODE == 5) asm volatile(DEP4 DEP4 DEP4 DEP4 : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b), "s"(s0), "s"(s1), "s"(m));
    }
    if (x0 + x1 + x2 + x3 + (float)(s0 + s1) == 123456.789f) out[0] = x0;
}
// s0 / s1 are declared as inputs only; the SALU blocks overwrite them anyway (a probe: nothing reads them afterwards).

template <int MODE>
static void run(const char* what, int valu, int salu, float* out) {
    const int iters = 20000;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int waves = 1; waves <= 8; waves *= 2) {
        const int blocks = 256 * waves;  // 256-thread blocks: 4 waves, one per SIMD of a CU; `waves` blocks per CU
        float ms = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(k_probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        const double cyc = ms * 1e-3 * 2.4e9 / ((double)iters * waves);
        printf("%-34s waves/SIMD %d  cycles per block per wave-slot %7.1f  (per VALU %5.2f%s)\n", what, waves, cyc, valu ? cyc / valu : 0.0, salu ? "" : "");
        if (salu) printf("%-34s                per SALU %5.2f\n", "", cyc / salu);
    }
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256));
    run<0>("16 independent v_fma_f32", 16, 0, out);
    run<5>("16 dependent v_fma_f32 (4 chains)", 16, 0, out);
    run<6>("16 v_pk_fma_f32 (2 chains)", 16, 0, out);
    run<1>("16 s_add_u32", 0, 16, out);
    run<2>("16 v_fma + 16 s_add interleaved", 16, 16, out);
    run<3>("16 v_fma + 32 s_add interleaved", 16, 32, out);
    run<4>("8 x (v_cmp, saveexec, v_fma, s_or)", 16, 16, out);
    return 0;
}
