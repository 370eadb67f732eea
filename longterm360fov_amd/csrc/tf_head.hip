// The two two-layer heads of the raw-TensorFlow model as ONE launch forward and ONE launch backward
// (mycode/lstm.py:321-337, _pred_mean_var_xyz2_new, on the top layer's final state h (B,H)):
//     a1 = relu(h W1m + b1m)   mu  = tanh(a1 W2m + b2m)          (H -> M -> O, M = 32, O = 3)
//     a3 = relu(h W1v + b1v)   var = exp (a3 W2v + b2v)
// At the script's batch (32) the seven forward and thirteen backward launches of the generic Dense / activation entry points
// each do microseconds of arithmetic: the step is bound by launch rate, so the head is fused, not tuned for bandwidth.
// Shapes: B <= 64, M <= 32, O <= 8, H <= 2048 (fov_tf_head_supported); anything else stays on the generic entry points.
#include "fov_common.h"

namespace fov {

namespace {

constexpr int TH_ROWS = 4;     // rows of h per forward workgroup
constexpr int TH_KQ = 4;       // k-slices of the forward contraction
constexpr int TH_M = 32;       // hidden units per branch at most
constexpr int TH_O = 8;        // outputs per branch at most
constexpr int TH_B = 64;       // rows at most (backward keeps d1 | d3 of all rows in LDS)

struct TfHeadParams {
    const float* h;
    const float *W1m, *b1m, *W2m, *b2m, *W1v, *b1v, *W2v, *b2v;
    float *a1, *mu, *a3, *var;              // forward: outputs; backward: inputs
    const float *dmu, *dvar;
    float *gW1m, *gb1m, *gW2m, *gb2m, *gW1v, *gb1v, *gW2v, *gb2v, *dh;
    int B, H, M, O, accumulate;
};

// Forward.  Thread (col, kq): col < 32 is unit col of the mean branch, col >= 32 unit col - 32 of the variance branch; it
// contracts a quarter of H for its unit and four rows (W loads independent and coalesced over col, h from LDS as a broadcast).
__global__ __launch_bounds__(256) void tf_head_fwd_kernel(TfHeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float th_lds[];
    float* xs = th_lds;                                  // [TH_ROWS][H]
    float* red = xs + TH_ROWS * p.H;                     // [TH_KQ][TH_ROWS][64]
    float* hid = red + TH_KQ * TH_ROWS * 64;             // [TH_ROWS][64]   a1 | a3
    const int H = p.H, M = p.M, O = p.O;
    const int row0 = blockIdx.x * TH_ROWS;
    for (int e = threadIdx.x; e < TH_ROWS * H; e += 256) {
        const int r = e / H, k = e - r * H;
        xs[e] = (row0 + r < p.B) ? p.h[(size_t)(row0 + r) * H + k] : 0.f;
    }
    __syncthreads();
    const int col = threadIdx.x & 63, kq = threadIdx.x >> 6;
    const int m = col & 31;
    const float* W1 = col < 32 ? p.W1m : p.W1v;
    const int kc = (H + TH_KQ - 1) / TH_KQ;
    const int k0 = kq * kc, k1 = (k0 + kc < H) ? k0 + kc : H;
    float acc[TH_ROWS] = {0.f, 0.f, 0.f, 0.f};
    if (m < M) {
        int k = k0;
        for (; k + 8 <= k1; k += 8) {
            float w[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) w[u] = W1[(size_t)(k + u) * M + m];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int r = 0; r < TH_ROWS; ++r) acc[r] = fmaf(xs[r * H + k + u], w[u], acc[r]);
        }
        for (; k < k1; ++k) {
            const float w = W1[(size_t)k * M + m];
#pragma unroll
            for (int r = 0; r < TH_ROWS; ++r) acc[r] = fmaf(xs[r * H + k], w, acc[r]);
        }
    }
#pragma unroll
    for (int r = 0; r < TH_ROWS; ++r) red[(kq * TH_ROWS + r) * 64 + col] = acc[r];
    __syncthreads();
    {
        const int r = threadIdx.x >> 6, row = row0 + r;      // 256 threads = 4 rows x 64 columns
        float v = 0.f;
#pragma unroll
        for (int q = 0; q < TH_KQ; ++q) v += red[(q * TH_ROWS + r) * 64 + col];   // fixed order
        if (m < M) v += (col < 32 ? p.b1m : p.b1v)[m];
        v = (m < M && v > 0.f) ? v : 0.f;
        hid[r * 64 + col] = v;
        if (row < p.B && m < M) (col < 32 ? p.a1 : p.a3)[(size_t)row * M + m] = v;
    }
    __syncthreads();
    if (threadIdx.x < TH_ROWS * 2 * TH_O) {
        const int r = threadIdx.x / (2 * TH_O), q = threadIdx.x - r * 2 * TH_O, br = q / TH_O, o = q - br * TH_O;
        const int row = row0 + r;
        if (row < p.B && o < O) {
            const float* W2 = br ? p.W2v : p.W2m;
            float v = (br ? p.b2v : p.b2m)[o];
            for (int mm = 0; mm < M; ++mm) v = fmaf(hid[r * 64 + 32 * br + mm], W2[mm * O + o], v);
            if (br) p.var[(size_t)row * O + o] = __expf(v);
            else p.mu[(size_t)row * O + o] = tanh_f(v);
        }
    }
}

// Backward, one launch, three kinds of workgroup.  Every workgroup first rebuilds the hidden-layer gradients of ALL rows in LDS
// (d[n][0..31] = d1 = (d2 W2m^T) [a1 > 0] with d2 = dmu (1 - mu^2);  d[n][32..63] = d3 = (d4 W2v^T) [a3 > 0] with d4 = dvar var):
// B x 64 values from a few hundred multiply-adds - cheaper than a launch that would write them once.
//   kind A (H/4 workgroups): gW1[i][m] (+)= sum_n h[n][i] d[n][m] for four rows i of both branches;
//   kind B (ceil(H/256) x ceil(B/8)): dh[n][i] = sum_m d[n][m] W1cat[i][m] for 256 units i and eight rows n;
//   kind C (one): gW2, gb2, gb1 of both branches.
__global__ __launch_bounds__(256) void tf_head_bwd_kernel(TfHeadParams p, int blocksA, int blocksB) {
    __shared__ float d2s[TH_B][2 * TH_O];     // d2 | d4
    __shared__ float ds[TH_B][64 + 1];        // d1 | d3
    const int B = p.B, H = p.H, M = p.M, O = p.O, tid = threadIdx.x;
    for (int e = tid; e < B * 2 * TH_O; e += 256) {
        const int n = e / (2 * TH_O), q = e - n * 2 * TH_O, br = q / TH_O, o = q - br * TH_O;
        float v = 0.f;
        if (o < O) {
            if (br) v = p.dvar[(size_t)n * O + o] * p.var[(size_t)n * O + o];
            else { const float y = p.mu[(size_t)n * O + o]; v = p.dmu[(size_t)n * O + o] * (1.f - y * y); }
        }
        d2s[n][q] = v;
    }
    __syncthreads();
    for (int e = tid; e < B * 64; e += 256) {
        const int n = e >> 6, col = e & 63, br = col >> 5, m = col & 31;
        float v = 0.f;
        if (m < M) {
            const float a = (br ? p.a3 : p.a1)[(size_t)n * M + m];
            if (a > 0.f) {
                const float* W2 = br ? p.W2v : p.W2m;
                for (int o = 0; o < O; ++o) v = fmaf(d2s[n][br * TH_O + o], W2[m * O + o], v);
            }
        }
        ds[n][col] = v;
    }
    __syncthreads();
    const int blk = blockIdx.x;
    if (blk < blocksA) {
        const int i = blk * 4 + (tid >> 6), col = tid & 63, br = col >> 5, m = col & 31;
        if (i < H && m < M) {
            float v = 0.f;
            for (int n = 0; n < B; ++n) v = fmaf(p.h[(size_t)n * H + i], ds[n][col], v);
            float* g = (br ? p.gW1v : p.gW1m) + (size_t)i * M + m;
            *g = p.accumulate ? *g + v : v;
        }
    } else if (blk < blocksA + blocksB) {
        const int b2 = blk - blocksA, ichunks = (H + 255) / 256;
        const int i = (b2 % ichunks) * 256 + tid, n0 = (b2 / ichunks) * 8;
        if (i < H) {
            float wm[TH_M], wv[TH_M];
#pragma unroll
            for (int m = 0; m < TH_M; ++m) {
                wm[m] = m < M ? p.W1m[(size_t)i * M + m] : 0.f;
                wv[m] = m < M ? p.W1v[(size_t)i * M + m] : 0.f;
            }
            for (int n = n0; n < n0 + 8 && n < B; ++n) {
                float v = 0.f;
#pragma unroll
                for (int m = 0; m < TH_M; ++m) v = fmaf(ds[n][m], wm[m], v);
#pragma unroll
                for (int m = 0; m < TH_M; ++m) v = fmaf(ds[n][32 + m], wv[m], v);
                p.dh[(size_t)n * H + i] = v;
            }
        }
    } else {
        // gW2[m][o] = sum_n a[n][m] d2[n][o] for both branches; gb2[o] = sum_n d2[n][o]; gb1[m] = sum_n d[n][m]
        for (int e = tid; e < 2 * M * O; e += 256) {
            const int br = e / (M * O), q = e - br * M * O, m = q / O, o = q - m * O;
            const float* a = br ? p.a3 : p.a1;
            float v = 0.f;
            for (int n = 0; n < B; ++n) v = fmaf(a[(size_t)n * M + m], d2s[n][br * TH_O + o], v);
            float* g = (br ? p.gW2v : p.gW2m) + q;
            *g = p.accumulate ? *g + v : v;
        }
        for (int e = tid; e < 2 * O; e += 256) {
            const int br = e / O, o = e - br * O;
            float v = 0.f;
            for (int n = 0; n < B; ++n) v += d2s[n][br * TH_O + o];
            float* g = (br ? p.gb2v : p.gb2m) + o;
            *g = p.accumulate ? *g + v : v;
        }
        for (int e = tid; e < 2 * M; e += 256) {
            const int br = e / M, m = e - br * M;
            float v = 0.f;
            for (int n = 0; n < B; ++n) v += ds[n][32 * br + m];
            float* g = (br ? p.gb1v : p.gb1m) + m;
            *g = p.accumulate ? *g + v : v;
        }
    }
}

bool tf_head_shape_ok(int B, int H, int M, int O) {
    return B >= 1 && B <= TH_B && H >= 1 && H <= 2048 && M >= 1 && M <= TH_M && O >= 1 && O <= TH_O;
}

}  // namespace

}  // namespace fov

using namespace fov;

extern "C" {

int fov_tf_head_supported(int B, int H, int M, int O) { return tf_head_shape_ok(B, H, M, O) ? 1 : 0; }

int fov_tf_head_fwd(const float* h, const float* mu_W1, const float* mu_b1, const float* mu_W2, const float* mu_b2,
                    const float* var_W1, const float* var_b1, const float* var_W2, const float* var_b2, float* a1, float* mu,
                    float* a3, float* var, int B, int H, int M, int O, fov_stream_t stream) {
    if (B == 0) return FOV_OK;
    if (!h || !mu_W1 || !mu_b1 || !mu_W2 || !mu_b2 || !var_W1 || !var_b1 || !var_W2 || !var_b2 || !a1 || !mu || !a3 || !var) {
        set_error("fov_tf_head_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (!tf_head_shape_ok(B, H, M, O)) { set_error("fov_tf_head_fwd: unsupported shape (B <= 64, H <= 2048, M <= 32, O <= 8)"); return FOV_ERR_UNSUPPORTED; }
    TfHeadParams p = {};
    p.h = h; p.W1m = mu_W1; p.b1m = mu_b1; p.W2m = mu_W2; p.b2m = mu_b2; p.W1v = var_W1; p.b1v = var_b1; p.W2v = var_W2; p.b2v = var_b2;
    p.a1 = a1; p.mu = mu; p.a3 = a3; p.var = var; p.B = B; p.H = H; p.M = M; p.O = O;
    const size_t lds = sizeof(float) * ((size_t)TH_ROWS * H + TH_KQ * TH_ROWS * 64 + TH_ROWS * 64);
    hipLaunchKernelGGL(tf_head_fwd_kernel, dim3((unsigned)((B + TH_ROWS - 1) / TH_ROWS)), dim3(256), lds, (hipStream_t)stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("tf head forward launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int fov_tf_head_bwd(const float* h, const float* mu_W1, const float* mu_W2, const float* var_W1, const float* var_W2, const float* a1,
                    const float* mu, const float* a3, const float* var, const float* dmu, const float* dvar, float* g_mu_W1,
                    float* g_mu_b1, float* g_mu_W2, float* g_mu_b2, float* g_var_W1, float* g_var_b1, float* g_var_W2, float* g_var_b2,
                    float* dh, int B, int H, int M, int O, int accumulate, fov_stream_t stream) {
    if (B == 0) return FOV_OK;
    if (!h || !mu_W1 || !mu_W2 || !var_W1 || !var_W2 || !a1 || !mu || !a3 || !var || !dmu || !dvar || !g_mu_W1 || !g_mu_b1 || !g_mu_W2 ||
        !g_mu_b2 || !g_var_W1 || !g_var_b1 || !g_var_W2 || !g_var_b2 || !dh) {
        set_error("fov_tf_head_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (!tf_head_shape_ok(B, H, M, O)) { set_error("fov_tf_head_bwd: unsupported shape (B <= 64, H <= 2048, M <= 32, O <= 8)"); return FOV_ERR_UNSUPPORTED; }
    TfHeadParams p = {};
    p.h = h; p.W1m = mu_W1; p.W2m = mu_W2; p.W1v = var_W1; p.W2v = var_W2;
    p.a1 = const_cast<float*>(a1); p.mu = const_cast<float*>(mu); p.a3 = const_cast<float*>(a3); p.var = const_cast<float*>(var);
    p.dmu = dmu; p.dvar = dvar;
    p.gW1m = g_mu_W1; p.gb1m = g_mu_b1; p.gW2m = g_mu_W2; p.gb2m = g_mu_b2; p.gW1v = g_var_W1; p.gb1v = g_var_b1; p.gW2v = g_var_W2; p.gb2v = g_var_b2;
    p.dh = dh; p.B = B; p.H = H; p.M = M; p.O = O; p.accumulate = accumulate ? 1 : 0;
    const int blocksA = (H + 3) / 4, blocksB = ((H + 255) / 256) * ((B + 7) / 8);
    hipLaunchKernelGGL(tf_head_bwd_kernel, dim3((unsigned)(blocksA + blocksB + 1)), dim3(256), 0, (hipStream_t)stream, p, blocksA, blocksB);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("tf head backward launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // extern "C"
