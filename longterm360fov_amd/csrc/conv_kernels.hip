// ConvLSTM2D building blocks (a8/a9: keras ConvLSTM2D / Conv2D / Conv1D / Softmax calls of
// mycode/convlstm_seq2seq.py:100-126,146-165,170-189,209-258).  Round-1 structure: one implicit-GEMM
// convolution launch per (layer, operand) and one pointwise gate launch per layer-step, driven from the
// host; NHWC activations, (kh,kw,C,N) kernels exactly as Keras stores them.
//
//   conv2d_igemm_kernel   y = act(conv2d_same(x, w) + b + add)  fp32 MFMA implicit GEMM: M = B*H*W pixels,
//                         K = kh*kw*C gathered on the fly with zero 'same' padding (no im2col buffer),
//                         N = output channels.  Roofline: MFMA (2*K*N FLOP per pixel).
//   convlstm_gates_kernel i,f,c,o gates + cell update on z (pixels, 4F)                         - HBM
//   softmax_lastdim       channel softmax                                                       - HBM
#include "fov_common.h"

namespace fov {

struct ConvArgs {
    const float* x;     // (B,H,W,*) with pixel stride ldx >= C and batch stride ldb >= H*W*ldx
    const float* w;     // (kh*kw*C, N)
    const float* bias;  // (N) or NULL
    const float* add;   // (B*H*W, N) or NULL (may alias y)
    float* y;           // (B*H*W, N)
    int B, H, W, C, N, kh, kw, act;   // act: 0 none, 2 relu
    long ldx;   // pixel stride
    long ldb;   // batch stride
};

template <int MI, int NI, int WAVES_M>
__global__ __launch_bounds__(256) void conv2d_igemm_kernel(ConvArgs g) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = 16 * MI * WAVES_M, BN = 16 * NI * WAVES_N, BK = 16;
    constexpr int RA = BM / 16, RB = BN / 16;
    __shared__ __attribute__((aligned(16))) float As[2][BK][BM + 4];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN + 4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long M = (long)g.B * g.H * g.W;
    const int K = g.kh * g.kw * g.C;
    const long m0 = (long)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int wm = (wave / WAVES_N) * 16 * MI, wn = (wave % WAVES_N) * 16 * NI;
    const int li = lane & 15, lq = lane >> 4;
    const int ph = (g.kh - 1) / 2, pw = (g.kw - 1) / 2;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A staging: element e = tid + 256 r -> (pixel mm = e/16, k offset kk = e%16); per element the pixel's
    // (y, x, base pointer) is fixed and the filter tap (dy, dx, c) advances by 16 channels per k-tile.
    const int a_kk = tid & 15;
    int a_y[RA], a_x[RA], a_c[RA], a_dy[RA], a_dx[RA];
    const float* a_base[RA];
    bool a_live[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const long m = m0 + ((tid + 256 * r) >> 4);
        a_live[r] = m < M;
        const long mm = a_live[r] ? m : 0;
        const int b = (int)(mm / ((long)g.H * g.W));
        const int rem = (int)(mm - (long)b * g.H * g.W);
        a_y[r] = rem / g.W;
        a_x[r] = rem - a_y[r] * g.W;
        a_base[r] = g.x + (long)b * g.ldb;
        const int tap = a_kk / g.C;
        a_c[r] = a_kk - tap * g.C;
        a_dy[r] = tap / g.kw;
        a_dx[r] = tap - a_dy[r] * g.kw;
    }
    // B staging: w is (K, N) row-major: consecutive threads walk n
    int b_nn[RB], b_kk[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int e = tid + 256 * r;
        b_kk[r] = e / BN;
        b_nn[r] = e - b_kk[r] * BN;
    }
    float ra[RA], rb[RB];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const int yy = a_y[r] + a_dy[r] - ph, xx = a_x[r] + a_dx[r] - pw;
            const bool ok = a_live[r] && (k0 + a_kk < K) && yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            ra[r] = ok ? a_base[r][((long)yy * g.W + xx) * g.ldx + a_c[r]] : 0.f;
            a_c[r] += BK;
            while (a_c[r] >= g.C) {
                a_c[r] -= g.C;
                if (++a_dx[r] == g.kw) { a_dx[r] = 0; ++a_dy[r]; }
            }
        }
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int k = k0 + b_kk[r], n = n0 + b_nn[r];
            rb[r] = (k < K && n < g.N) ? g.w[(long)k * g.N + n] : 0.f;
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int r = 0; r < RA; ++r) As[buf][a_kk][(tid + 256 * r) >> 4] = ra[r];
#pragma unroll
        for (int r = 0; r < RB; ++r) Bs[buf][b_kk[r]][b_nn[r]] = rb[r];
    };
    int buf = 0;
    fetch(0);
    stash(0);
    __syncthreads();
    for (int k0 = 0; k0 < K; k0 += BK) {
        const bool more = (k0 + BK < K);
        if (more) fetch(k0 + BK);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float av[MI], bv[NI];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = As[buf][ks * 4 + lq][wm + i * 16 + li];
#pragma unroll
            for (int j = 0; j < NI; ++j) bv[j] = Bs[buf][ks * 4 + lq][wn + j * 16 + li];
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[j], acc[i][j], 0, 0, 0);
        }
        if (more) stash(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = m0 + wm + i * 16 + lq * 4 + r;
                const int n = n0 + wn + j * 16 + li;
                if (m < M && n < g.N) {
                    float v = acc[i][j][r];
                    if (g.bias) v += g.bias[n];
                    if (g.add) v += g.add[m * g.N + n];
                    if (g.act == 2) v = fmaxf(v, 0.f);
                    g.y[m * g.N + n] = v;
                }
            }
}

// ConvLSTM2DCell gates: z (rows, 4F) channel blocks i,f,c,o; c (rows, F) in/out; h written with pixel
// stride ldh (so a layer can write straight into its slot of the channel-concatenated feature map).
template <int ACT>
__global__ __launch_bounds__(256) void convlstm_gates_kernel(const float* __restrict__ z, float* __restrict__ c,
                                                             float* __restrict__ h, long ldh, long rows, int F) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * F) return;
    const long m = idx / F;
    const int j = (int)(idx - m * F);
    const float* zp = z + m * 4 * F + j;
    const float i = rec_act<ACT>(zp[0]), f = rec_act<ACT>(zp[F]), gg = tanh_f(zp[2 * F]), o = rec_act<ACT>(zp[3 * F]);
    const float cn = fmaf(f, c[idx], i * gg);
    c[idx] = cn;
    h[m * ldh + j] = o * tanh_f(cn);
}

__global__ __launch_bounds__(256) void softmax_lastdim_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                              int n) {
    const long row = (long)blockIdx.x * 256 + threadIdx.x;
    if (row >= rows) return;
    const float* xp = x + row * n;
    float mx = xp[0];
    for (int i = 1; i < n; ++i) mx = fmaxf(mx, xp[i]);
    float s = 0.f;
    for (int i = 0; i < n; ++i) s += __expf(xp[i] - mx);
    const float inv = 1.0f / s;
    float* yp = y + row * n;
    for (int i = 0; i < n; ++i) yp[i] = __expf(xp[i] - mx) * inv;
}

static int conv_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s launch: %s", what, hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int conv2d_fwd(const float* x, long ldx, long ldb, const float* w, const float* bias, const float* add, float* y, int B, int H,
               int W, int C, int N, int kh, int kw, int act, hipStream_t stream) {
    ConvArgs g = {};
    g.x = x; g.w = w; g.bias = bias; g.add = add; g.y = y;
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N; g.kh = kh; g.kw = kw; g.act = act; g.ldx = ldx; g.ldb = ldb;
    const long M = (long)B * H * W;
    if (M == 0 || N == 0) return FOV_OK;
    if (N <= 32) {
        const dim3 grid((N + 31) / 32, (unsigned)((M + 255) / 256));
        hipLaunchKernelGGL((conv2d_igemm_kernel<4, 2, 4>), grid, dim3(256), 0, stream, g);
    } else if (N <= 64) {
        const dim3 grid((N + 63) / 64, (unsigned)((M + 255) / 256));
        hipLaunchKernelGGL((conv2d_igemm_kernel<4, 4, 4>), grid, dim3(256), 0, stream, g);
    } else {
        const dim3 grid((N + 127) / 128, (unsigned)((M + 127) / 128));
        hipLaunchKernelGGL((conv2d_igemm_kernel<4, 4, 2>), grid, dim3(256), 0, stream, g);
    }
    return conv_check_launch("conv2d_igemm");
}

int convlstm_gates(const float* z, float* c, float* h, long ldh, long rows, int F, int act, hipStream_t stream) {
    const long n = rows * F;
    if (n == 0) return FOV_OK;
    const dim3 grid((unsigned)((n + 255) / 256));
    if (act == FOV_ACT_HARD_SIGMOID)
        hipLaunchKernelGGL(convlstm_gates_kernel<FOV_ACT_HARD_SIGMOID>, grid, dim3(256), 0, stream, z, c, h, ldh, rows, F);
    else
        hipLaunchKernelGGL(convlstm_gates_kernel<FOV_ACT_SIGMOID>, grid, dim3(256), 0, stream, z, c, h, ldh, rows, F);
    return conv_check_launch("convlstm_gates");
}

int softmax_lastdim(const float* x, float* y, long rows, int n, hipStream_t stream) {
    if (rows == 0) return FOV_OK;
    hipLaunchKernelGGL(softmax_lastdim_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, stream, x, y, rows, n);
    return conv_check_launch("softmax_lastdim");
}

}  // namespace fov
