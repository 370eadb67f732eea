// Microbenchmark: what can run BESIDE v_mfma_f32_16x16x4_f32 on one SIMD of gfx950?
//  (A) one wave per SIMD: N independent VALU fillers (v_fma_f32 / v_exp_f32 / ds_read_b128) in every MFMA gap;
//  (B) two waves per SIMD (512 threads): an MFMA-only wave beside a VALU-only wave, each timed alone and together;
//  (C) the "two tiles" shape: every wave alternates NM MFMAs and NV VALU instructions; one wave per SIMD doing
//      (2 NM, 2 NV) per iteration against two waves per SIMD doing (NM, NV) each, the second wave half a period late.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_overlap.hip -o mfma_f32_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(i, F) "v_mfma_f32_16x16x4_f32 %" #i ", %8, %9, %" #i "\n\t" F
#define MF4(F) MF(0, F) MF(1, F) MF(2, F) MF(3, F)
#define MF16(F) MF4(F) MF4(F) MF4(F) MF4(F)

#define FMA1 "v_fma_f32 %4, %4, %8, %4\n\t"
#define FMA2 FMA1 "v_fma_f32 %5, %5, %8, %5\n\t"
#define FMA4 FMA2 "v_fma_f32 %6, %6, %8, %6\n\t" "v_fma_f32 %7, %7, %8, %7\n\t"
#define FMA6 FMA4 FMA2
#define FMA8 FMA4 FMA4
#define EXP1 "v_exp_f32 %4, %4\n\t"
#define EXP2 EXP1 "v_exp_f32 %5, %5\n\t"
#define EXP3 EXP2 "v_exp_f32 %6, %6\n\t"
#define EXP4 EXP3 "v_exp_f32 %7, %7\n\t"
#define MIX5 EXP1 "v_fma_f32 %5, %5, %8, %5\n\t" "v_fma_f32 %6, %6, %8, %6\n\t" "v_fma_f32 %7, %7, %8, %7\n\t" "v_fma_f32 %5, %5, %8, %5\n\t"

#define DEF_A(NAME, F)                                                                                            \
    __global__ __launch_bounds__(256, 1) void NAME(float* out, unsigned long long* cyc, int iters) {              \
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;                                                       \
        float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                                    \
        float a = threadIdx.x * 1e-3f, b = 0.5f;                                                                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                     \
        for (int it = 0; it < iters; ++it)                                                                        \
            asm volatile(MF16(F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) \
                         : "v"(a), "v"(b));                                                                       \
        asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));                   \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3;                  \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                          \
    }

DEF_A(ka_none, "")
DEF_A(ka_fma1, FMA1)
DEF_A(ka_fma2, FMA2)
DEF_A(ka_fma4, FMA4)
DEF_A(ka_fma6, FMA6)
DEF_A(ka_fma8, FMA8)
DEF_A(ka_exp1, EXP1)
DEF_A(ka_exp2, EXP2)
DEF_A(ka_exp3, EXP3)
DEF_A(ka_exp4, EXP4)
DEF_A(ka_mix5, MIX5)
DEF_A(ka_nop1, "s_nop 0\n\t")
DEF_A(ka_salu2, "s_add_u32 s20, s20, 1\n\ts_add_u32 s21, s21, 1\n\t")

// VALU-only streams of the same instruction counts (no MFMA): what the fillers cost alone
#define V16(F) F F F F F F F F F F F F F F F F
#define DEF_V(NAME, F)                                                                                            \
    __global__ __launch_bounds__(256, 1) void NAME(float* out, unsigned long long* cyc, int iters) {              \
        f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;                                                       \
        float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;                                    \
        float a = threadIdx.x * 1e-3f, b = 0.5f;                                                                  \
        unsigned long long t0 = __builtin_amdgcn_s_memtime();                                                     \
        for (int it = 0; it < iters; ++it)                                                                        \
            asm volatile(V16(F) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)  \
                         : "v"(a), "v"(b));                                                                       \
        unsigned long long t1 = __builtin_amdgcn_s_memtime();                                                     \
        out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3;                  \
        if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;                          \
    }
DEF_V(kv_fma4, FMA4)
DEF_V(kv_exp4, EXP4)

// (B) / (C): 512 threads = two waves per SIMD.  role(wave) decides what a wave does.
// mode 0: waves 0-3 MFMA-only (nm per iteration), waves 4-7 VALU-only (nv per iteration)
// mode 1: waves 0-3 MFMA-only, waves 4-7 exit        mode 2: waves 0-3 exit, waves 4-7 VALU-only
// mode 3: every wave alternates nm MFMAs and nv VALU; waves 4-7 start with their VALU phase
// mode 4: as 3 but waves 4-7 exit (one wave per SIMD doing the alternating stream)
// nm, nv in units of 16 instructions.  VALU mix per 16: 6 v_exp + 10 v_fma (a cell update's proportions)
#define VMIX16 EXP2 FMA4 EXP2 FMA4 EXP2 FMA2
__global__ __launch_bounds__(512, 2) void kb(float* out, unsigned long long* cyc, int iters, int nm, int nv, int mode) {
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float f0 = threadIdx.x * 1e-3f, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    float a = threadIdx.x * 1e-3f, b = 0.5f;
    const int wave = threadIdx.x >> 6;
    const bool hi = wave >= 4;
    bool do_m = true, do_v = true, v_first = false;
    if (mode == 0) { do_m = !hi; do_v = hi; }
    if (mode == 1) { do_m = !hi; do_v = false; }
    if (mode == 2) { do_m = false; do_v = hi; }
    if (mode == 3) v_first = hi;
    if (mode == 4 && hi) { do_m = do_v = false; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    if (do_m || do_v) {
        if (v_first)
            for (int j = 0; j < nv; ++j)
                asm volatile(VMIX16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));
        for (int it = 0; it < iters; ++it) {
            if (do_m)
                for (int j = 0; j < nm; ++j)
                    asm volatile(MF16("") : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));
            if (do_v) {
                if (do_m) asm volatile("s_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
                for (int j = 0; j < nv; ++j)
                    asm volatile(VMIX16 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3) : "v"(a), "v"(b));
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * 512 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

static double median(std::vector<unsigned long long> v) {
    std::sort(v.begin(), v.end());
    return (double)v[v.size() / 2];
}

template <typename K>
static void run_a(const char* name, K kern, int blocks, int nfill) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 4 * 8);
    const int iters = 400;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, blocks * 4 * 8, hipMemcpyDeviceToHost);
    const double c = median(h) / (iters * 16.0);
    printf("A %-10s fillers/gap %d : %.2f cycles per MFMA gap  (kernel %.3f ms)\n", name, nfill, c, ms);
    hipFree(out); hipFree(cyc);
}

static void run_b(int blocks, int nm, int nv, int mode, const char* what) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 512 * 4); hipMalloc(&cyc, blocks * 8 * 8);
    const int iters = 100;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(kb, dim3(blocks), dim3(512), 0, 0, out, cyc, iters, nm, nv, mode);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(blocks * 8), lo, hi;
    hipMemcpy(h.data(), cyc, blocks * 8 * 8, hipMemcpyDeviceToHost);
    for (int b = 0; b < blocks; ++b)
        for (int w = 0; w < 8; ++w) (w < 4 ? lo : hi).push_back(h[b * 8 + w]);
    printf("B mode %d nm=%3d nv=%3d (x16) %-46s: waves0-3 %9.1f  waves4-7 %9.1f cycles per iteration\n", mode, nm, nv, what,
           median(lo) / iters, median(hi) / iters);
    hipFree(out); hipFree(cyc);
}

int main(int argc, char** argv) {
    const int blocks = argc > 1 ? atoi(argv[1]) : 256;
    printf("blocks = %d\n", blocks);
    run_a("none", ka_none, blocks, 0);
    run_a("fma", ka_fma1, blocks, 1);
    run_a("fma", ka_fma2, blocks, 2);
    run_a("fma", ka_fma4, blocks, 4);
    run_a("fma", ka_fma6, blocks, 6);
    run_a("fma", ka_fma8, blocks, 8);
    run_a("exp", ka_exp1, blocks, 1);
    run_a("exp", ka_exp2, blocks, 2);
    run_a("exp", ka_exp3, blocks, 3);
    run_a("exp", ka_exp4, blocks, 4);
    run_a("exp+4fma", ka_mix5, blocks, 5);
    run_a("s_nop", ka_nop1, blocks, 1);
    run_a("salu", ka_salu2, blocks, 2);
    run_a("V fma4", kv_fma4, blocks, 4);
    run_a("V exp4", kv_exp4, blocks, 4);
    // (B)
    run_b(blocks, 22, 0, 1, "MFMA-only waves alone (352 MFMAs)");
    run_b(blocks, 0, 26, 2, "VALU-only waves alone (416 VALU)");
    run_b(blocks, 22, 26, 0, "MFMA waves beside VALU waves");
    run_b(blocks, 22, 52, 0, "MFMA waves beside VALU waves (832 VALU)");
    // (C)
    run_b(blocks, 22, 26, 4, "ONE wave/SIMD: 352 MFMA then 416 VALU");
    run_b(blocks, 11, 13, 3, "TWO waves/SIMD: each 176 MFMA then 208 VALU");
    run_b(blocks, 22, 52, 4, "ONE wave/SIMD: 352 MFMA then 832 VALU");
    run_b(blocks, 11, 26, 3, "TWO waves/SIMD: each 176 MFMA then 416 VALU");
    return 0;
}
