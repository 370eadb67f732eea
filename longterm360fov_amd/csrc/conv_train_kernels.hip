// ConvLSTM2D training kernels (a8 backward: what model.fit runs under Keras/TF autodiff for
// mycode/convlstm_seq2seq.py:100-126,146-165,209-287).
//
//   convlstm_gates_train_kernel  forward gates that also keep (i,f,g,o) and c_t for the backward pass   - HBM
//   convlstm_gates_bwd_kernel    dh, dc -> dz (rows,4F), dc_{t-1}                                        - HBM
//   conv_wgrad_kernel            dW[tap][c][n] = sum_pixels x[pixel + tap][c] * dz[pixel][n]            - MFMA
//   conv_weight_transpose_kernel W'(kh-1-i, kw-1-j, n, c) = W(i, j, c, n): with it the data gradient of a
//                                'same' convolution is the forward kernel on dz (dx = conv2d_same(dz, W'))
//   softmax_bwd_kernel           dy = p * (dp - sum_c dp*p)                                              - HBM
#include "fov_common.h"

namespace fov {

// derivative of the recurrent activation expressed in its OUTPUT a (as Keras/TF autodiff evaluates it)
template <int ACT>
__device__ __forceinline__ float conv_act_grad(float a) {
    if (ACT == FOV_ACT_HARD_SIGMOID) return (a > 0.f && a < 1.f) ? 0.2f : 0.f;
    return a * (1.f - a);
}

template <int ACT>
__global__ __launch_bounds__(256) void convlstm_gates_train_kernel(const float* __restrict__ z, const float* __restrict__ c_prev,
                                                                   float* __restrict__ c_new, float* __restrict__ h, long ldh,
                                                                   float* __restrict__ gates, long rows, int F) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * F) return;
    const long m = idx / F;
    const int j = (int)(idx - m * F);
    const float* zp = z + m * 4 * F + j;
    const float i = rec_act<ACT>(zp[0]), f = rec_act<ACT>(zp[F]), gg = tanh_f(zp[2 * F]), o = rec_act<ACT>(zp[3 * F]);
    const float cn = fmaf(f, c_prev ? c_prev[idx] : 0.f, i * gg);
    float* gp = gates + m * 4 * F + j;   // may alias z
    gp[0] = i; gp[F] = f; gp[2 * F] = gg; gp[3 * F] = o;
    c_new[idx] = cn;
    h[m * ldh + j] = o * tanh_f(cn);
}

// dh (rows,F) with pixel stride lddh; dc (rows,F): in = dL/dc_t from the future, out = dL/dc_{t-1};
// gates = activated i,f,g,o of step t; c_prev may be NULL (zero initial state).  dz may alias gates.
template <int ACT>
__global__ __launch_bounds__(256) void convlstm_gates_bwd_kernel(const float* __restrict__ dh, long lddh, float* __restrict__ dc,
                                                                 const float* __restrict__ gates, const float* __restrict__ c_prev,
                                                                 const float* __restrict__ c_new, float* __restrict__ dz,
                                                                 long rows, int F) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * F) return;
    const long m = idx / F;
    const int j = (int)(idx - m * F);
    const float* gp = gates + m * 4 * F + j;
    const float i = gp[0], f = gp[F], gg = gp[2 * F], o = gp[3 * F];
    const float tc = tanh_f(c_new[idx]);
    const float dhv = dh[m * lddh + j];
    const float dct = dc[idx] + dhv * o * (1.f - tc * tc);
    const float cp = c_prev ? c_prev[idx] : 0.f;
    float* dp = dz + m * 4 * F + j;
    dp[0] = dct * gg * conv_act_grad<ACT>(i);
    dp[F] = dct * cp * conv_act_grad<ACT>(f);
    dp[2 * F] = dct * i * (1.f - gg * gg);
    dp[3 * F] = dhv * tc * conv_act_grad<ACT>(o);
    dc[idx] = dct * f;
}

__global__ __launch_bounds__(256) void conv_weight_transpose_kernel(const float* __restrict__ w, float* __restrict__ wt, int kh,
                                                                    int kw, int C, int N) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)kh * kw * C * N;
    if (idx >= total) return;
    // idx walks the OUTPUT (kh,kw,N,C) so writes are coalesced
    const int c = (int)(idx % C);
    long r = idx / C;
    const int n = (int)(r % N);
    r /= N;
    const int j = (int)(r % kw), i = (int)(r / kw);
    wt[idx] = w[(((long)(kh - 1 - i) * kw + (kw - 1 - j)) * C + c) * N + n];
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ dp, const float* __restrict__ p,
                                                          float* __restrict__ dy, long rows, int n) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* dpp = dp + row * n;
    const float* pp = p + row * n;
    float s = 0.f;
    for (int i = 0; i < n; ++i) s = fmaf(dpp[i], pp[i], s);
    float* yp = dy + row * n;
    for (int i = 0; i < n; ++i) yp[i] = pp[i] * (dpp[i] - s);
}

// ---------------------------------------------------------------------------------------
// Weight gradient of a 'same' convolution as a TN product per filter tap:
//   dW[tap] (C x N) = A^T B,  A(k = pixel, m = c) = x[pixel + shift(tap)][c] (zero outside the image),
//                             B(k = pixel, n)     = dz[pixel][n]
// Same machine as gemm_f32_kernel (k-slow operands, buffer loads, k-tiles of 16 pixels, [k][m] LDS tiles,
// deterministic split-K over the pixels): blockIdx.y = (channel tile, tap).  The tap's shift is wave-uniform
// and sits in the tile's base address; every staged element tracks the (y, x) of its pixel incrementally
// (+16 pixels per tile) and presents an out-of-range offset when the shifted pixel leaves the image.
// x must be batch-dense (batch stride = H*W*ldx): the training tape keeps every step's maps that way.
// ---------------------------------------------------------------------------------------
struct WgradArgs {
    const float* x;
    const float* dz;
    float* out;   // [split][kh*kw*C*N] partial slices, or dW itself when split == 1
    int H, W, C, N, kh, kw;
    int dil;      // dilation of the taps over x
    long P;       // pixels (B*H*W)
    long ldx;     // pixel stride of x
    int ctiles;
    int split;
    long tiles_per_split;
};

typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));

template <int MI, int NI, int WAVES_M, int AVEC, int BVEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs g) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = 16 * MI * WAVES_M, BN = 16 * NI * WAVES_N, BK = 16;
    constexpr int RA = AVEC ? (BM * 4 + 255) / 256 : BM / 16;
    constexpr int RB = BVEC ? (BN * 4 + 255) / 256 : BN / 16;
    constexpr unsigned OOR = 0x80000000u;
    // row strides and fragment reads as in gemm_f32_kernel (train_kernels.hip, OperandStage): MFMA tile i, row q of a wave is
    // operand row 4*q + (i%4) of its group of four tiles (ds_read_b128 at [k][4*li]) or 2*q + (i%2) of a trailing pair
    constexpr int LDA = BM + (MI % 4 == 0 ? 0 : 8), LDB = BN + (NI % 4 == 0 ? 0 : 8);
    static_assert(MI % 2 == 0 && NI % 2 == 0, "tiles per wave: groups of four plus at most one pair");
    __shared__ __attribute__((aligned(16))) float As[2][BK][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ctile = blockIdx.y % g.ctiles, tap = blockIdx.y / g.ctiles;
    const int c0 = ctile * BM, n0 = blockIdx.x * BN;
    const int sy = (tap / g.kw - (g.kh - 1) / 2) * g.dil, sx = (tap % g.kw - (g.kw - 1) / 2) * g.dil;
    const int wm = (wave / WAVES_N) * 16 * MI, wn = (wave % WAVES_N) * 16 * NI;
    const int li = lane & 15, lq = lane >> 4;
    const long ktiles = (g.P + BK - 1) / BK;
    const long tbeg = (long)blockIdx.z * g.tiles_per_split;
    long tend = tbeg + g.tiles_per_split;
    if (tend > ktiles) tend = ktiles;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // staging maps: consecutive threads walk the row index (channels / output channels) of one pixel
    int a_kk[RA], a_mm[RA], a_y[RA], a_x[RA];
    unsigned a_off[RA];
    const int HW = g.H * g.W;
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int e = tid + 256 * r;
        a_kk[r] = AVEC ? e / (BM / 4) : e / BM;
        a_mm[r] = AVEC ? 4 * (e - a_kk[r] * (BM / 4)) : e - a_kk[r] * BM;
        const bool ok = a_kk[r] < BK && c0 + a_mm[r] < g.C;
        a_off[r] = ok ? (unsigned)(((long)a_kk[r] * g.ldx + c0 + a_mm[r]) * 4) : OOR;
        const int pos = (int)((tbeg * BK + a_kk[r]) % HW);
        a_y[r] = pos / g.W;
        a_x[r] = pos - a_y[r] * g.W;
    }
    int b_kk[RB], b_nn[RB];
    unsigned b_off[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int e = tid + 256 * r;
        b_kk[r] = BVEC ? e / (BN / 4) : e / BN;
        b_nn[r] = BVEC ? 4 * (e - b_kk[r] * (BN / 4)) : e - b_kk[r] * BN;
        const bool ok = b_kk[r] < BK && n0 + b_nn[r] < g.N;
        b_off[r] = ok ? (unsigned)(((long)b_kk[r] * g.N + n0 + b_nn[r]) * 4) : OOR;
    }
    long f_t = tbeg;   // the tile the next fetch() loads
    float ra[AVEC ? 1 : RA], rb[BVEC ? 1 : RB];
    f32x4 va[AVEC ? RA : 1], vb[BVEC ? RB : 1];
    auto fetch = [&]() {
        const long p0 = f_t * BK;
        const long krem = g.P - p0;   // pixels left
        const float* xt = g.x + (p0 + (long)sy * g.W + sx) * g.ldx;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xt), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int yy = a_y[r] + sy, xx = a_x[r] + sx;
            const bool ok = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W && a_kk[r] < krem;
            const unsigned off = ok ? a_off[r] : OOR;
            if constexpr (AVEC) {
                const wu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
                va[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
            } else {
                ra[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
            }
        }
        const float* zt = g.dz + p0 * g.N;
        const int krows = krem < BK ? (int)krem : BK;
        const __amdgpu_buffer_rsrc_t zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(zt), 0, krows * g.N * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if constexpr (BVEC) {
                const wu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(zrs, b_off[r], 0, 0);
                vb[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
            } else {
                rb[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zrs, b_off[r], 0, 0));
            }
        }
        // the final iteration re-reads the last tile (unused): the position and the pixel trackers stay
        if (f_t + 1 < tend) {
            ++f_t;
#pragma unroll
            for (int r = 0; r < RA; ++r) {   // every element's pixel advances by 16
                a_x[r] += BK;
                while (a_x[r] >= g.W) { a_x[r] -= g.W; ++a_y[r]; }
                while (a_y[r] >= g.H) a_y[r] -= g.H;
            }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            if constexpr (AVEC) {
                if (a_kk[r] < BK) *(f32x4*)&As[buf][a_kk[r]][a_mm[r]] = va[r];
            } else {
                As[buf][a_kk[r]][a_mm[r]] = ra[r];
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if constexpr (BVEC) {
                if (b_kk[r] < BK) *(f32x4*)&Bs[buf][b_kk[r]][b_nn[r]] = vb[r];
            } else {
                Bs[buf][b_kk[r]][b_nn[r]] = rb[r];
            }
        }
    };
    float av[4][MI], bv[4][NI];
    typedef float wf32x2 __attribute__((ext_vector_type(2)));
    auto read_frags = [&](int buf) {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const float* pa = &As[buf][ks * 4 + lq][wm];
            const float* pb = &Bs[buf][ks * 4 + lq][wn];
#pragma unroll
            for (int q = 0; q < MI / 4; ++q) {
                const f32x4 v = *(const f32x4*)(pa + 64 * q + 4 * li);
                av[ks][4 * q] = v[0]; av[ks][4 * q + 1] = v[1]; av[ks][4 * q + 2] = v[2]; av[ks][4 * q + 3] = v[3];
            }
            if constexpr (MI % 4 == 2) {
                const wf32x2 v = *(const wf32x2*)(pa + 64 * (MI / 4) + 2 * li);
                av[ks][MI - 2] = v[0]; av[ks][MI - 1] = v[1];
            }
#pragma unroll
            for (int q = 0; q < NI / 4; ++q) {
                const f32x4 v = *(const f32x4*)(pb + 64 * q + 4 * li);
                bv[ks][4 * q] = v[0]; bv[ks][4 * q + 1] = v[1]; bv[ks][4 * q + 2] = v[2]; bv[ks][4 * q + 3] = v[3];
            }
            if constexpr (NI % 4 == 2) {
                const wf32x2 v = *(const wf32x2*)(pb + 64 * (NI / 4) + 2 * li);
                bv[ks][NI - 2] = v[0]; bv[ks][NI - 1] = v[1];
            }
        }
    };
    auto row_a = [&](int i, int q) { return i < 4 * (MI / 4) ? 64 * (i / 4) + 4 * q + (i & 3) : 64 * (MI / 4) + 2 * q + (i & 1); };
    auto row_b = [&](int j, int q) { return j < 4 * (NI / 4) ? 64 * (j / 4) + 4 * q + (j & 3) : 64 * (NI / 4) + 2 * q + (j & 1); };
    auto mfmas = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[ks][i], bv[ks][j], acc[i][j], 0, 0, 0);
    };
    int buf = 0;
    if (tbeg < tend) {
        fetch();
        stash(0);
    }
    __syncthreads();
    for (long t = tbeg; t < tend; ++t) {
        read_frags(buf);
        __builtin_amdgcn_sched_barrier(0);
        fetch();
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        __builtin_amdgcn_sched_barrier(0);
        stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
    float* out = g.out + (size_t)blockIdx.z * g.kh * g.kw * g.C * g.N + (size_t)tap * g.C * g.N;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + wm + row_a(i, lq * 4 + r), n = n0 + wn + row_b(j, li);
                if (c < g.C && n < g.N) out[(size_t)c * g.N + n] = acc[i][j][r];
            }
}

static int ct_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s launch: %s", what, hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

size_t conv2d_wgrad_workspace_floats(int C, int N, int kh, int kw) {
    // up to 64 split-K slices; narrow layers (conv_wgrad_lines.hip splits over the maps): up to 256 slices within 128 MB
    const size_t wn = (size_t)kh * kw * C * N;
    size_t lines = 256 * wn;
    if (lines > ((size_t)32 << 20)) lines = (size_t)32 << 20;
    return (64 * wn > lines ? 64 * wn : lines) + 64;
}

int conv2d_wgrad(const float* x, long ldx, const float* dz, float* dw, int B, int H, int W, int C, int N, int kh, int kw,
                 int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream, int dil) {
    const long P = (long)B * H * W;
    const size_t wn = (size_t)kh * kw * C * N;
    if (P == 0) {
        if (!accumulate) (void)hipMemsetAsync(dw, 0, sizeof(float) * wn, stream);
        return FOV_OK;
    }
    if ((long)16 * ldx * 4 >= (1L << 30) || (long)16 * N * 4 >= (1L << 30)) { set_error("conv2d_wgrad: strides out of range"); return FOV_ERR_UNSUPPORTED; }
    // narrow layers: every tap in one workgroup, operands read once (conv_wgrad_lines.hip)
    if (conv_wgrad_lines_takes(H, W, C, N, kh, kw, dil < 1 ? 1 : dil))
        return conv_wgrad_lines(x, ldx, dz, dw, B, H, W, C, N, kh, accumulate, scratch, scratch_floats, stream);
    int BM, BN, variant;
    if (C <= 32) { BM = 32; BN = 256; variant = 0; }
    else if (C <= 96) { BM = 96; BN = 256; variant = 1; }
    else if (N <= 32) { BM = 256; BN = 32; variant = 3; }   // skinny output (head 1024 -> 30): a 128-wide column tile was 77 % padding
    else { BM = 128; BN = 128; variant = 2; }
    // (wide layers, measured at the head's 512 -> 1024: 128 x 128 tiles 359 ms = 0.77 of peak, 128 x 256 437 ms, 256 x 128 459 ms; the
    // blocks that share an x tile pinned to one XCD: 360 ms - the launch is not bound by its operand fetches)
    WgradArgs g = {};
    g.x = x; g.dz = dz; g.H = H; g.W = W; g.C = C; g.N = N; g.kh = kh; g.kw = kw; g.P = P; g.ldx = ldx;
    g.dil = dil < 1 ? 1 : dil;
    g.ctiles = (C + BM - 1) / BM;
    const int gn = (N + BN - 1) / BN;
    const long ktiles = (P + 15) / 16;
    const long tiles = (long)g.ctiles * kh * kw * gn;
    int split = 1;
    if (tiles < 512 && ktiles >= 32) {
        split = (int)((768 + tiles - 1) / tiles);
        const long maxs = ktiles / 8;
        if (split > maxs) split = (int)maxs;
        if (split > 64) split = 64;
        if (split < 1) split = 1;
    }
    else if (tiles < 4096 && ktiles >= 4096) {
        // a few hundred long blocks (head 512 -> 1024: 800 tiles of 103 680 k-tiles each) leave the chip a quarter empty
        // in their last round (800 / 256 CUs = 3.1 rounds): slices until the block count is far above the CU count
        split = (int)((4096 + tiles - 1) / tiles);
        if (split > 16) split = 16;
    }
    while (split > 1 && (size_t)split * wn > scratch_floats) --split;
    const long tps = (ktiles + split - 1) / split;
    split = (int)((ktiles + tps - 1) / tps);
    const bool via_scratch = split > 1 || accumulate;
    if (via_scratch && (size_t)split * wn > scratch_floats) { set_error("conv2d_wgrad: scratch too small"); return FOV_ERR_WORKSPACE; }
    g.out = via_scratch ? scratch : dw;
    g.split = split;
    g.tiles_per_split = tps;
    const bool avec = (C & 3) == 0 && (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0;
    const bool bvec = (N & 3) == 0 && (((uintptr_t)dz) & 15) == 0;
    const dim3 grid(gn, g.ctiles * kh * kw, split);
#define FOV_WGRAD_LAUNCH(MI_, NI_, WM_)                                                                                 \
    do {                                                                                                                \
        if (avec && bvec) hipLaunchKernelGGL((conv_wgrad_kernel<MI_, NI_, WM_, 1, 1>), grid, dim3(256), 0, stream, g);   \
        else if (avec) hipLaunchKernelGGL((conv_wgrad_kernel<MI_, NI_, WM_, 1, 0>), grid, dim3(256), 0, stream, g);      \
        else if (bvec) hipLaunchKernelGGL((conv_wgrad_kernel<MI_, NI_, WM_, 0, 1>), grid, dim3(256), 0, stream, g);      \
        else hipLaunchKernelGGL((conv_wgrad_kernel<MI_, NI_, WM_, 0, 0>), grid, dim3(256), 0, stream, g);                \
    } while (0)
    if (variant == 0) FOV_WGRAD_LAUNCH(2, 4, 1);
    else if (variant == 1) FOV_WGRAD_LAUNCH(6, 4, 1);
    else if (variant == 3) FOV_WGRAD_LAUNCH(4, 2, 4);
    else FOV_WGRAD_LAUNCH(4, 4, 2);
#undef FOV_WGRAD_LAUNCH
    int rc = ct_check_launch("conv_wgrad");
    if (rc || !via_scratch) return rc;
    return splitk_reduce(scratch, dw, (long)wn, split, accumulate, stream);
}

int convlstm_gates_train(const float* z, const float* c_prev, float* c_new, float* h, long ldh, float* gates, long rows, int F,
                         int act, hipStream_t stream) {
    const long n = rows * F;
    if (n == 0) return FOV_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (act == FOV_ACT_HARD_SIGMOID)
        hipLaunchKernelGGL(convlstm_gates_train_kernel<FOV_ACT_HARD_SIGMOID>, grid, dim3(256), 0, stream, z, c_prev, c_new, h, ldh,
                           gates, rows, F);
    else
        hipLaunchKernelGGL(convlstm_gates_train_kernel<FOV_ACT_SIGMOID>, grid, dim3(256), 0, stream, z, c_prev, c_new, h, ldh,
                           gates, rows, F);
    return ct_check_launch("convlstm_gates_train");
}

int convlstm_gates_bwd(const float* dh, long lddh, float* dc, const float* gates, const float* c_prev, const float* c_new,
                       float* dz, long rows, int F, int act, hipStream_t stream) {
    const long n = rows * F;
    if (n == 0) return FOV_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (act == FOV_ACT_HARD_SIGMOID)
        hipLaunchKernelGGL(convlstm_gates_bwd_kernel<FOV_ACT_HARD_SIGMOID>, grid, dim3(256), 0, stream, dh, lddh, dc, gates, c_prev,
                           c_new, dz, rows, F);
    else
        hipLaunchKernelGGL(convlstm_gates_bwd_kernel<FOV_ACT_SIGMOID>, grid, dim3(256), 0, stream, dh, lddh, dc, gates, c_prev,
                           c_new, dz, rows, F);
    return ct_check_launch("convlstm_gates_bwd");
}

int conv_weight_transpose(const float* w, float* wt, int kh, int kw, int C, int N, hipStream_t stream) {
    const long total = (long)kh * kw * C * N;
    if (total == 0) return FOV_OK;
    hipLaunchKernelGGL(conv_weight_transpose_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, wt, kh, kw, C, N);
    return ct_check_launch("conv_weight_transpose");
}

int softmax_lastdim_bwd(const float* dp, const float* p, float* dy, long rows, int n, hipStream_t stream) {
    if (rows == 0) return FOV_OK;
    hipLaunchKernelGGL(softmax_bwd_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, dp, p, dy, rows, n);
    return ct_check_launch("softmax_bwd");
}

}  // namespace fov
