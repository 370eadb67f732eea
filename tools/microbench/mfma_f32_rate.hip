// Microbenchmark: cycles per v_mfma_f32_16x16x4_f32 by operand placement (VGPR vs AGPR for the
// accumulator and the B operand), 4 independent accumulators, one wave per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 mfma_f32_rate.hip -o mfma_f32_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MF(ACC, A, B, CA, CB) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+" CA(ACC) : "v"(A), CB(B))

template <int VAR>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    f32x4 acc[4];
    float b[16];
    float a[4];
    for (int i = 0; i < 4; ++i) { acc[i] = (f32x4){0, 0, 0, 0}; a[i] = threadIdx.x * 0.001f + i; }
    for (int i = 0; i < 16; ++i) b[i] = threadIdx.x * 0.002f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (VAR == 0) MF(acc[g], a[s], b[s * 4 + g], "v", "a");
                if (VAR == 1) MF(acc[g], a[s], b[s * 4 + g], "v", "v");
                if (VAR == 2) MF(acc[g], a[s], b[s * 4 + g], "a", "a");
                if (VAR == 3) MF(acc[g], a[s], b[s * 4 + g], "a", "v");
            }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int g = 0; g < 4; ++g) r += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// VAR 4: the shipped stream shape - A operand re-read from LDS (one ds_read_b128 per 16 MFMAs,
// issued one block ahead), B in AGPRs.  VAR 5: same with two blocks of read-ahead.
template <int VAR>
__global__ __launch_bounds__(256, 1) void k2(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float hs[16 * 260];
    for (int i = threadIdx.x; i < 16 * 260; i += 256) hs[i] = i * 0.001f;
    __syncthreads();
    f32x4 acc[4];
    float b[64];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    for (int i = 0; i < 64; ++i) b[i] = threadIdx.x * 0.002f + i;
    const int lane = threadIdx.x & 63;
    const float* hrow = hs + (lane & 15) * 260 + 4 * (lane >> 4);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 a0 = *(const f32x4*)(hrow);
        f32x4 a1 = *(const f32x4*)(hrow + 16);
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            f32x4 an = a0;
            if (VAR == 4) { if (j + 1 < 16) an = *(const f32x4*)(hrow + 16 * (j + 1)); }
            else { if (j + 2 < 16) an = *(const f32x4*)(hrow + 16 * (j + 2)); }
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g) MF(acc[g], a0[s], b[(j & 3) * 16 + s * 4 + g], "v", "a");
            if (VAR == 4) a0 = an; else { a0 = a1; a1 = an; }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int g = 0; g < 4; ++g) r += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

// VAR 6: the x.K stream shape - A operand (x tile) AND B operand (K slice) from LDS: one ds_read_b128 of A per 16 MFMAs,
// one of B per 4 MFMAs, B two blocks ahead, 6 k-blocks fully unrolled (96 MFMAs per iteration).  VAR 7: B one q-block
// (4 reads) ahead.
template <int VAR>
__global__ __launch_bounds__(256, 1) void k3(float* out, unsigned long long* cyc, int iters) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* hs = sm;                   // x tile [16][100]
    float* ks = sm + 16 * 100 + 64;   // K slices: 4 waves x 26 blocks x 256 floats
    for (int i = threadIdx.x; i < 16 * 100 + 64 + 4 * 26 * 256; i += 256) sm[i] = i * 0.001f;
    __syncthreads();
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0, 0, 0, 0};
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float* arow = hs + (lane & 15) * 100 + 4 * (lane >> 4);
    const float* bl = ks + wave * 26 * 256 + lane * 4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        f32x4 a = *(const f32x4*)arow;
        if (VAR == 6) {
            f32x4 b0 = *(const f32x4*)bl, b1 = *(const f32x4*)(bl + 256);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                f32x4 an = a;
                if (q + 1 < 6) an = *(const f32x4*)(arow + 16 * (q + 1));
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    f32x4 bn = b1;
                    if (4 * q + s + 2 < 24) bn = *(const f32x4*)(bl + (4 * q + s + 2) * 256);
#pragma unroll
                    for (int g = 0; g < 4; ++g) MF(acc[g], a[s], b0[g], "v", "v");
                    b0 = b1; b1 = bn;
                }
                a = an;
            }
        } else {
            f32x4 b[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) b[s] = *(const f32x4*)(bl + s * 256);
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                f32x4 an = a, bn[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) bn[s] = b[s];
                if (q + 1 < 6) {
                    an = *(const f32x4*)(arow + 16 * (q + 1));
#pragma unroll
                    for (int s = 0; s < 4; ++s) bn[s] = *(const f32x4*)(bl + (4 * (q + 1) + s) * 256);
                }
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int g = 0; g < 4; ++g) MF(acc[g], a[s], b[s][g], "v", "v");
                a = an;
#pragma unroll
                for (int s = 0; s < 4; ++s) b[s] = bn[s];
            }
        }
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = 0;
    for (int g = 0; g < 4; ++g) r += acc[g][0] + acc[g][1] + acc[g][2] + acc[g][3];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int VAR>
void run3(const char* name, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 400;
    const size_t lds = (16 * 100 + 64 + 4 * 26 * 256) * 4;
    hipFuncSetAttribute((const void*)k3<VAR>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k3<VAR>, dim3(blocks), dim3(256), lds, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("%-28s blocks=%4d: %.2f cycles per MFMA\n", name, blocks, s / blocks / (iters * 96.0));
    hipFree(out); hipFree(cyc);
}

template <int VAR>
void run2(const char* name, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 200;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k2<VAR>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("%-28s blocks=%4d: %.2f cycles per MFMA\n", name, blocks, s / blocks / (iters * 256.0));
    hipFree(out); hipFree(cyc);
}

template <int VAR>
void run(const char* name, int blocks) {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&cyc, blocks * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k<VAR>, dim3(blocks), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024]; hipMemcpy(h, cyc, blocks * 8, hipMemcpyDeviceToHost);
    double s = 0; for (int i = 0; i < blocks; ++i) s += h[i];
    printf("%-28s blocks=%4d: %.2f cycles per MFMA\n", name, blocks, s / blocks / (iters * 16.0));
    hipFree(out); hipFree(cyc);
}

int main() {
    for (int blocks : {1, 256}) {
        run<0>("acc VGPR, B AGPR (shipped)", blocks);
        run<1>("acc VGPR, B VGPR", blocks);
        run<2>("acc AGPR, B AGPR", blocks);
        run<3>("acc AGPR, B VGPR", blocks);
        run2<4>("LDS A, 1 block ahead", blocks);
        run2<5>("LDS A, 2 blocks ahead", blocks);
        run3<6>("x.K: LDS A + LDS B (2 ahead)", blocks);
        run3<7>("x.K: LDS A + LDS B (q ahead)", blocks);
    }
    return 0;
}
