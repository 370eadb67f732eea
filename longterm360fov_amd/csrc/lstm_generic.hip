// Generic LSTM sequence kernels: any H, any F, any B.  One 256-thread workgroup owns 4
// sequences; thread j owns hidden unit j (all four gate columns, so the cell update is
// thread-local); weights are streamed from L2 every step.  This is the always-available
// fallback and the independent cross-check of the cluster kernel - not the fast path.
//
// Replaces keras LSTM / Dense calls of mycode/FoV_seq2seq.py:83-97,137-178 (see include/fov360.h).
#include <atomic>

#include "fov_common.h"

namespace fov {

constexpr int GB = 4;  // sequences per workgroup

template <int ACT>
__device__ __forceinline__ void generic_step(const float* __restrict__ sx, int ldx, int F,
                                             const float* __restrict__ K, const float* __restrict__ R,
                                             const float* __restrict__ b, int H,
                                             const float* __restrict__ sh_prev, float* __restrict__ sh_next,
                                             float* __restrict__ sc, float* __restrict__ reserve = nullptr,
                                             size_t res_stride = 0, int rows = 0,
                                             const float* __restrict__ zx = nullptr, size_t zx_stride = 0) {
    const int H4 = 4 * H;
    for (int j = threadIdx.x; j < H; j += blockDim.x) {
        float acc[GB][4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const float bv = b[g * H + j];
#pragma unroll
            for (int s = 0; s < GB; ++s)
                acc[s][g] = bv + ((zx && s < rows) ? zx[(size_t)s * zx_stride + g * H + j] : 0.f);
        }
        for (int k = 0; k < (zx ? 0 : F); ++k) {
            float w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g] = K[(size_t)k * H4 + g * H + j];
#pragma unroll
            for (int s = 0; s < GB; ++s) {
                const float xv = sx[s * ldx + k];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[s][g] = fmaf(xv, w[g], acc[s][g]);
            }
        }
        for (int k = 0; k < H; ++k) {
            float w[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) w[g] = R[(size_t)k * H4 + g * H + j];
#pragma unroll
            for (int s = 0; s < GB; ++s) {
                const float hv = sh_prev[s * H + k];
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[s][g] = fmaf(hv, w[g], acc[s][g]);
            }
        }
#pragma unroll
        for (int s = 0; s < GB; ++s) {
            const float i = rec_act<ACT>(acc[s][0]);
            const float f = rec_act<ACT>(acc[s][1]);
            const float g = tanh_f(acc[s][2]);
            const float o = rec_act<ACT>(acc[s][3]);
            const float c = fmaf(f, sc[s * H + j], i * g);
            sc[s * H + j] = c;
            sh_next[s * H + j] = o * tanh_f(c);
            if (reserve && s < rows) {
                float* rp = reserve + (size_t)s * res_stride + j;
                rp[0] = i; rp[H] = f; rp[2 * H] = g; rp[3 * H] = o; rp[4 * H] = c;
            }
        }
    }
}

template <int ACT, bool DECODE>
__global__ __launch_bounds__(256) void lstm_generic_kernel(LstmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int H = p.H;
    const int ldx = (p.F > p.F_dec ? p.F : p.F_dec) + 1;
    float* sh_h = smem;               // [2][GB][H]
    float* sh_c = sh_h + 2 * GB * H;  // [GB][H]
    float* sh_x = sh_c + GB * H;      // [GB][ldx]
    const int b0 = blockIdx.x * GB;
    const int rows = (p.B - b0 < GB) ? p.B - b0 : GB;
    const int tid = threadIdx.x;

    for (int i = tid; i < GB * H; i += blockDim.x) {
        const int s = i / H, j = i - s * H;
        const bool live = s < rows;
        sh_h[i] = (live && p.h0) ? p.h0[(size_t)(b0 + s) * H + j] : 0.f;
        sh_c[i] = (live && p.c0) ? p.c0[(size_t)(b0 + s) * H + j] : 0.f;
    }
    int cur = 0;
    for (int t = 0; t < p.T; ++t) {
        if (!p.zx) {
            for (int i = tid; i < GB * p.F; i += blockDim.x) {
                const int s = i / p.F, k = i - s * p.F;
                sh_x[s * ldx + k] = (s < rows) ? p.x[((size_t)(b0 + s) * p.T + t) * p.F + k] : 0.f;
            }
        }
        __syncthreads();
        generic_step<ACT>(sh_x, ldx, p.F, p.K, p.R, p.b, H, sh_h + cur * GB * H, sh_h + (cur ^ 1) * GB * H, sh_c,
                          p.reserve ? p.reserve + (((size_t)b0 * p.T + t) * 5) * H : nullptr, (size_t)p.T * 5 * H, rows,
                          p.zx ? p.zx + ((size_t)b0 * p.T + t) * 4 * H : nullptr, (size_t)p.T * 4 * H);
        __syncthreads();
        cur ^= 1;
        if (p.hs) {
            for (int i = tid; i < rows * H; i += blockDim.x) {
                const int s = i / H, j = i - s * H;
                p.hs[((size_t)(b0 + s) * p.T + t) * H + j] = sh_h[cur * GB * H + i];
            }
        }
    }
    if (DECODE) {
        const int O = p.F_dec;
        for (int i = tid; i < GB * O; i += blockDim.x) {
            const int s = i / O, o = i - s * O;
            sh_x[s * ldx + o] = (s < rows) ? p.dec_in0[(size_t)(b0 + s) * O + o] : 0.f;
        }
        __syncthreads();
        const int wave = tid >> 6, lane = tid & 63;  // wave s computes the dense row of sequence s
        for (int t = 0; t < p.T_out; ++t) {
            generic_step<ACT>(sh_x, ldx, O, p.dK, p.dR, p.db, H, sh_h + cur * GB * H, sh_h + (cur ^ 1) * GB * H, sh_c);
            __syncthreads();
            cur ^= 1;
            const float* hrow = sh_h + cur * GB * H + wave * H;
            for (int o = 0; o < O; ++o) {
                float part = 0.f;
                for (int k = lane; k < H; k += 64) part = fmaf(hrow[k], p.dW[(size_t)k * O + o], part);
#pragma unroll
                for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m);
                if (lane == 0) {
                    const float y = tanh_f(part + p.dbias[o]);
                    sh_x[wave * ldx + o] = y;
                    if (wave < rows) p.out[((size_t)(b0 + wave) * p.T_out + t) * O + o] = y;
                }
            }
            __syncthreads();
        }
    }
    if (p.hT)
        for (int i = tid; i < rows * H; i += blockDim.x) p.hT[(size_t)b0 * H + i] = sh_h[cur * GB * H + i];
    if (p.cT)
        for (int i = tid; i < rows * H; i += blockDim.x) p.cT[(size_t)b0 * H + i] = sh_c[i];
}

static std::atomic<long> g_generic_launches{0};
long generic_launch_count() { return g_generic_launches.load(); }

int launch_generic(const LstmParams& p, bool decode, hipStream_t stream) {
    if (p.B == 0) return FOV_OK;
    ++g_generic_launches;
    const int ldx = (p.F > p.F_dec ? p.F : p.F_dec) + 1;
    const size_t lds = sizeof(float) * ((size_t)3 * GB * p.H + (size_t)GB * ldx);
    if (lds > 160 * 1024) {
        set_error("generic kernel: H=%d F=%d needs %zu B of LDS (> 160 KiB)", p.H, p.F, lds);
        return FOV_ERR_UNSUPPORTED;
    }
    const dim3 grid((p.B + GB - 1) / GB), block(256);
    void (*kern)(LstmParams) = nullptr;
    if (p.act == FOV_ACT_HARD_SIGMOID)
        kern = decode ? lstm_generic_kernel<FOV_ACT_HARD_SIGMOID, true> : lstm_generic_kernel<FOV_ACT_HARD_SIGMOID, false>;
    else
        kern = decode ? lstm_generic_kernel<FOV_ACT_SIGMOID, true> : lstm_generic_kernel<FOV_ACT_SIGMOID, false>;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    }
    hipLaunchKernelGGL(kern, grid, block, lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("generic launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
