// valu_rate.hip — issue rate of the VALU instructions the traversal loops are made of (gfx950): cycles per wave64 instruction and SIMD, from kernels
// that are nothing but long runs of independent instructions of one kind, 8 waves per SIMD. Built and run by tools/gpu_valu_rate.sh:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
// (An inline-asm block that writes SCC — an s_and_b64, say — between a counted loop's s_cmp and its branch hangs the kernel: none here.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP16(X) X X X X X X X X X X X X X X X X
constexpr int kIters = 4096;   // x 16 instructions x 8 registers' worth of independence

#define KERNEL(name, ASM)                                                                                                   \
    __global__ __launch_bounds__(256) void name(float* out, float seed) {                                                  \
        float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;  \
        float b = seed * 0.5f, c = seed * 0.25f;                                                                           \
        for (int i = 0; i < kIters; i++) {                                                                                 \
            asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));                 \
        }                                                                                                                  \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;                                   \
    }

#define A_FMA(r) "v_fma_f32 " #r ", " #r ", %8, %9\n"
#define A_MUL(r) "v_mul_f32 " #r ", " #r ", %8\n"
#define A_FMAMIX(r) "v_fma_mix_f32 " #r ", " #r ", %8, %9 op_sel_hi:[1,0,0]\n"
#define A_FMAMIXHI(r) "v_fma_mix_f32 " #r ", " #r ", %8, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n"
#define A_ALIGN(r) "v_alignbit_b32 " #r ", " #r ", " #r ", %8\n"
#define A_PERM(r) "v_perm_b32 " #r ", " #r ", %8, %9\n"
#define A_MIN3(r) "v_min3_f32 " #r ", " #r ", %8, %9\n"
#define A_MAX(r) "v_max_f32 " #r ", " #r ", %8\n"
#define A_MED3U(r) "v_med3_u32 " #r ", " #r ", %8, %9\n"
#define A_MINU(r) "v_min_u32 " #r ", " #r ", %8\n"
#define A_CNDMASK(r) "v_cndmask_b32 " #r ", " #r ", %8, vcc\n"
#define A_CVT(r) "v_cvt_f32_f16 " #r ", " #r "\n"
#define A_ANDOR(r) "v_and_or_b32 " #r ", " #r ", %8, %9\n"
#define A_ADDU(r) "v_add_u32 " #r ", " #r ", %8\n"
#define A_LSHL(r) "v_lshlrev_b32 " #r ", 1, " #r "\n"
#define A_PKFMA(r) "v_pk_fma_f16 " #r ", " #r ", %8, %9\n"
#define A_RCP(r) "v_rcp_f32 " #r ", " #r "\n"
#define A_CMP(r) "v_cmp_lt_f32 vcc, " #r ", %8\n"
#define A_MOV(r) "v_mov_b32 " #r ", %8\n"
#define A_CNDS(r) "v_cndmask_b32 " #r ", " #r ", %8, s[10:11]\n"
#define A_CNDOTHER(r) "v_cndmask_b32 " #r ", %9, %8, vcc\n"
#define A_CMPCND(r) "v_cmp_lt_f32 vcc, " #r ", %8\n v_cndmask_b32 " #r ", " #r ", %9, vcc\n"
#define A_MAX3(r) "v_max3_f32 " #r ", " #r ", %8, %9\n"
#define A_ADD(r) "v_add_f32 " #r ", " #r ", %8\n"
#define A_SUB(r) "v_sub_f32 " #r ", " #r ", %8\n"
#define A_AND(r) "v_and_b32 " #r ", " #r ", %8\n"
#define A_OR(r) "v_or_b32 " #r ", " #r ", %8\n"
#define A_XOR(r) "v_xor_b32 " #r ", " #r ", %8\n"
#define A_LSHLADD(r) "v_lshl_add_u32 " #r ", " #r ", 2, %8\n"
#define A_ADD3(r) "v_add3_u32 " #r ", " #r ", %8, %9\n"
#define A_MAD24(r) "v_mad_u32_u24 " #r ", " #r ", %8, %9\n"
#define A_BFE(r) "v_bfe_u32 " #r ", " #r ", 3, 5\n"
#define A_PKFMA32(r) "v_pk_fma_f32 v[20:21], v[20:21], v[22:23], v[24:25]\n"
#define A_PKMUL32(r) "v_pk_mul_f32 v[20:21], v[20:21], v[22:23]\n"
#define A_FMAC(r) "v_fmac_f32 " #r ", %8, %9\n"
#define A_MIN(r) "v_min_f32 " #r ", " #r ", %8\n"
#define A_MUL24(r) "v_mul_u32_u24 " #r ", " #r ", %8\n"
#define A_MULLO(r) "v_mul_lo_u32 " #r ", " #r ", %8\n"

KERNEL(k_fma, A_FMA) KERNEL(k_mul, A_MUL) KERNEL(k_fmamix, A_FMAMIX) KERNEL(k_fmamixhi, A_FMAMIXHI) KERNEL(k_align, A_ALIGN) KERNEL(k_perm, A_PERM)
KERNEL(k_min3, A_MIN3) KERNEL(k_max, A_MAX) KERNEL(k_med3u, A_MED3U) KERNEL(k_minu, A_MINU) KERNEL(k_cndmask, A_CNDMASK) KERNEL(k_cvt, A_CVT)
KERNEL(k_andor, A_ANDOR) KERNEL(k_addu, A_ADDU) KERNEL(k_lshl, A_LSHL) KERNEL(k_pkfma, A_PKFMA) KERNEL(k_rcp, A_RCP) KERNEL(k_cmp, A_CMP) KERNEL(k_mov, A_MOV)
KERNEL(k_mul24, A_MUL24) KERNEL(k_mullo, A_MULLO)
KERNEL(k_cnds, A_CNDS) KERNEL(k_cndother, A_CNDOTHER) KERNEL(k_cmpcnd, A_CMPCND) KERNEL(k_max3, A_MAX3) KERNEL(k_add, A_ADD) KERNEL(k_sub, A_SUB) KERNEL(k_and, A_AND) KERNEL(k_or, A_OR) KERNEL(k_xor, A_XOR)
KERNEL(k_lshladd, A_LSHLADD) KERNEL(k_add3, A_ADD3) KERNEL(k_mad24, A_MAD24) KERNEL(k_bfe, A_BFE) KERNEL(k_fmac, A_FMAC) KERNEL(k_min, A_MIN)

__global__ __launch_bounds__(256) void k_cmp_then_cnd(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int i = 0; i < kIters; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %8\n"
                     "v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                     "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n"
                     "v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;
}
// ... the selects in their VOP3 encoding (vcc named as an ordinary SGPR pair)
__global__ __launch_bounds__(256) void k_cmp_then_cnd_e64(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int i = 0; i < kIters; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %8\n"
                     "v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n"
                     "v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n v_cndmask_b32_e64 %0, %0, %9, vcc\n v_cndmask_b32_e64 %1, %1, %9, vcc\n v_cndmask_b32_e64 %2, %2, %9, vcc\n"
                     "v_cndmask_b32_e64 %3, %3, %9, vcc\n v_cndmask_b32_e64 %4, %4, %9, vcc\n v_cndmask_b32_e64 %5, %5, %9, vcc\n v_cndmask_b32_e64 %6, %6, %9, vcc\n v_cndmask_b32_e64 %7, %7, %9, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;
}
// ... VOP2 again, a few idle cycles after the compare
__global__ __launch_bounds__(256) void k_cmp_then_cnd_nop(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int i = 0; i < kIters; i++) {
        asm volatile("v_cmp_lt_f32 vcc, %0, %8\n s_nop 4\n"
                     "v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n"
                     "v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n v_cndmask_b32 %0, %0, %9, vcc\n v_cndmask_b32 %1, %1, %9, vcc\n v_cndmask_b32 %2, %2, %9, vcc\n"
                     "v_cndmask_b32 %3, %3, %9, vcc\n v_cndmask_b32 %4, %4, %9, vcc\n v_cndmask_b32 %5, %5, %9, vcc\n v_cndmask_b32 %6, %6, %9, vcc\n v_cndmask_b32 %7, %7, %9, vcc\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "vcc");
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;
}
// the same with the mask in an SGPR pair written by the compare (VOP3 forms)
__global__ __launch_bounds__(256) void k_cmp_then_cnd_s(float* out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    float b = seed * 0.5f, c = seed * 0.25f;
    for (int i = 0; i < kIters; i++) {
        asm volatile("v_cmp_lt_f32 s[10:11], %0, %8\n"
                     "v_cndmask_b32 %1, %1, %8, s[10:11]\n v_cndmask_b32 %2, %2, %8, s[10:11]\n v_cndmask_b32 %3, %3, %8, s[10:11]\n v_cndmask_b32 %4, %4, %8, s[10:11]\n v_cndmask_b32 %5, %5, %8, s[10:11]\n"
                     "v_cndmask_b32 %6, %6, %8, s[10:11]\n v_cndmask_b32 %7, %7, %8, s[10:11]\n v_cndmask_b32 %0, %0, %9, s[10:11]\n v_cndmask_b32 %1, %1, %9, s[10:11]\n v_cndmask_b32 %2, %2, %9, s[10:11]\n"
                     "v_cndmask_b32 %3, %3, %9, s[10:11]\n v_cndmask_b32 %4, %4, %9, s[10:11]\n v_cndmask_b32 %5, %5, %9, s[10:11]\n v_cndmask_b32 %6, %6, %9, s[10:11]\n v_cndmask_b32 %7, %7, %9, s[10:11]\n"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c) : "s10", "s11");
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[threadIdx.x] = a0;
}
// packed fp32 (VOP3P on 64-bit register pairs): two multiply-adds per lane and instruction — does it issue at the full rate? (if it
// does, Moeller-Trumbore's cross / dot pairs are candidates)
typedef float float2v __attribute__((ext_vector_type(2)));
#define KERNEL_PK(name, ASM)                                                                                                \
    __global__ __launch_bounds__(256) void name(float* out, float seed) {                                                  \
        float2v a0 = {seed + threadIdx.x, seed}, a1 = a0 + 1.0f, a2 = a0 + 2.0f, a3 = a0 + 3.0f, a4 = a0 + 4.0f, a5 = a0 + 5.0f, a6 = a0 + 6.0f, a7 = a0 + 7.0f; \
        float2v b = {seed * 0.5f, seed * 0.75f}, c = {seed * 0.25f, seed * 0.125f};                                        \
        for (int i = 0; i < kIters; i++) {                                                                                 \
            asm volatile(ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) ASM(%0) ASM(%1) ASM(%2) ASM(%3) ASM(%4) ASM(%5) ASM(%6) ASM(%7) \
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(c));                 \
        }                                                                                                                  \
        const float2v t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                           \
        if (t.x + t.y == 12345.678f) out[threadIdx.x] = a0.x;                                                              \
    }
#define A_PKFMA_F32(r) "v_pk_fma_f32 " #r ", " #r ", %8, %9\n"
#define A_PKMUL_F32(r) "v_pk_mul_f32 " #r ", " #r ", %8\n"
#define A_PKADD_F32(r) "v_pk_add_f32 " #r ", " #r ", %8\n"
KERNEL_PK(k_pkfma_f32, A_PKFMA_F32) KERNEL_PK(k_pkmul_f32, A_PKMUL_F32) KERNEL_PK(k_pkadd_f32, A_PKADD_F32)
// the 64-bit address add the compiler emits for plane[idx] (one per global access whose offset is not provably 32-bit)
#define A_LSHLADD64(r) "v_lshl_add_u64 " #r ", " #r ", 4, %8\n"
KERNEL_PK(k_lshladd64, A_LSHLADD64)

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1e3;   // kHz -> MHz
    float* out; hipMalloc(&out, 4096);
    struct K { const char* name; void (*fn)(float*, float); };
    const std::vector<K> ks = {{"v_fma_f32", k_fma}, {"v_mul_f32", k_mul}, {"v_fma_mix_f32 (lo)", k_fmamix}, {"v_fma_mix_f32 (hi)", k_fmamixhi}, {"v_alignbit_b32", k_align}, {"v_perm_b32", k_perm},
                               {"v_min3_f32", k_min3}, {"v_max_f32", k_max}, {"v_med3_u32", k_med3u}, {"v_min_u32", k_minu}, {"v_cndmask_b32", k_cndmask}, {"v_cvt_f32_f16", k_cvt},
                               {"v_and_or_b32", k_andor}, {"v_add_u32", k_addu}, {"v_lshlrev_b32", k_lshl}, {"v_pk_fma_f16", k_pkfma}, {"v_rcp_f32", k_rcp}, {"v_cmp_lt_f32", k_cmp}, {"v_mov_b32", k_mov},
                               {"v_mul_u32_u24", k_mul24}, {"v_mul_lo_u32", k_mullo},
                               {"v_cndmask_b32 (sgpr pair)", k_cnds}, {"v_cndmask_b32 (dst not a source)", k_cndother}, {"v_cmp + v_cndmask pair (per 2)", k_cmpcnd}, {"v_max3_f32", k_max3}, {"v_add_f32", k_add}, {"v_sub_f32", k_sub},
                               {"v_and_b32", k_and}, {"v_or_b32", k_or}, {"v_xor_b32", k_xor}, {"v_lshl_add_u32", k_lshladd}, {"v_add3_u32", k_add3}, {"v_mad_u32_u24", k_mad24}, {"v_bfe_u32", k_bfe}, {"v_fmac_f32", k_fmac}, {"v_min_f32", k_min},
                               {"v_pk_fma_f32 (2 fp32 lanes per instruction)", k_pkfma_f32}, {"v_pk_mul_f32", k_pkmul_f32}, {"v_pk_add_f32", k_pkadd_f32}, {"v_lshl_add_u64 (64-bit address add)", k_lshladd64},
                               {"1 v_cmp -> vcc, 15 v_cndmask reading vcc", k_cmp_then_cnd}, {"1 v_cmp -> s[10:11], 15 v_cndmask reading it", k_cmp_then_cnd_s}, {"1 v_cmp -> vcc, 15 v_cndmask_e64 reading vcc", k_cmp_then_cnd_e64}, {"1 v_cmp -> vcc, s_nop 4, 15 v_cndmask reading vcc", k_cmp_then_cnd_nop}};
    printf("%s: %d CUs, clock %0.0f MHz (reported); 8 waves per SIMD, %d instructions per wave\n", p.gcnArchName, cus, mhz, kIters * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int waves_per_simd : {8}) {
        const int blocks = cus * waves_per_simd;   // 256 threads = 4 waves = one per SIMD; `waves_per_simd` blocks per CU
        printf("-- %d wave(s) per SIMD\n", waves_per_simd);
        for (const K& k : ks) {
            hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f);
            hipDeviceSynchronize();
            float best = 1e30f;
            for (int r = 0; r < 3; r++) {
                hipEventRecord(e0); hipLaunchKernelGGL(k.fn, dim3(blocks), dim3(256), 0, 0, out, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
            }
            const double instr_per_simd = (double)kIters * 16 * waves_per_simd;
            printf("%-52s %8.3f ms  %5.2f cycles per wave64 instruction and SIMD at 2400 MHz\n", k.name, best, best * 1e-3 * 2.4e9 / instr_per_simd);
            fflush(stdout);
        }
    }
    return 0;
}
