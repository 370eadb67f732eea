// extern "C" entry points of libfov360_hip.so (declared in include/fov360.h) plus the small
// non-recurrent kernels of the path: Dense(+tanh) and the mu/sigma^2 feature op.
#include <math.h>
#include <stdarg.h>
#include <stdio.h>

#include <mutex>
#include <unordered_map>

#include "fov_common.h"
#include "xch_common.h"
#include "bf16_common.h"

namespace fov {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---------------------------------------------------------------------------------------
// Dense: y = act(x W + b).  One wave per output row; lanes stride over In, shuffle-reduce.
// Memory-bound streaming op (x is read once; W stays in L2): used for the teacher-forced
// graph's Dense over (B*T_out, H) -> F_dec and for tests.
// ---------------------------------------------------------------------------------------
constexpr int DENSE_MAX_OUT = 16;

__global__ __launch_bounds__(256) void dense_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                    const float* __restrict__ b, const float* __restrict__ add,
                                                    long add_stride, float* __restrict__ y,
                                                    int N, int In, int Out, int activation) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= N) return;
    const float* xr = x + (size_t)row * In;
    for (int o0 = 0; o0 < Out; o0 += DENSE_MAX_OUT) {
        const int no = (Out - o0 < DENSE_MAX_OUT) ? Out - o0 : DENSE_MAX_OUT;
        float acc[DENSE_MAX_OUT];
#pragma unroll
        for (int o = 0; o < DENSE_MAX_OUT; ++o) acc[o] = 0.f;
        for (int k = lane; k < In; k += 64) {
            const float xv = xr[k];
            const float* wr = W + (size_t)k * Out + o0;
#pragma unroll
            for (int o = 0; o < DENSE_MAX_OUT; ++o)
                if (o < no) acc[o] = fmaf(xv, wr[o], acc[o]);
        }
#pragma unroll
        for (int o = 0; o < DENSE_MAX_OUT; ++o)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) acc[o] += __shfl_xor(acc[o], m);
        float mine = 0.f;
#pragma unroll
        for (int o = 0; o < DENSE_MAX_OUT; ++o) mine = (lane == o) ? acc[o] : mine;
        if (lane < no) {
            float v = mine + (b ? b[o0 + lane] : 0.f);
            if (add) v += add[(size_t)row * add_stride + o0 + lane];
            if (activation == 1) v = tanh_f(v);
            y[(size_t)row * Out + o0 + lane] = v;
        }
    }
}

// Narrow heads (Out <= 8, the Dense(6) of every model here): 16 lanes per row and four rows per wave, x read
// as 16-byte pieces, W transposed in LDS (the four row groups of a wave read the same addresses - broadcast),
// four butterfly steps instead of six and only over Out values.  ~4x fewer instructions per row than the
// generic kernel above; HBM-bound on x.
template <int VEC>   // VEC = 4: In % 4 == 0 and x 16-byte aligned; VEC = 1: any In (the others' projection: In = 198)
__global__ __launch_bounds__(256) void dense_small_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ b, const float* __restrict__ add,
                                                          long add_stride, float* __restrict__ y, int N, int In, int Out,
                                                          int activation) {
    extern __shared__ __attribute__((aligned(16))) float Wt[];   // [8][In]
    for (int e = threadIdx.x; e < 8 * In; e += 256) {
        const int o = e / In, k = e - o * In;
        Wt[e] = (o < Out) ? W[(size_t)k * Out + o] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * 4; row0 < N; row0 += (long)gridDim.x * 16) {
        const long row = row0 + g;
        const bool ok = row < N;
        const float* xr = x + (size_t)(ok ? row : 0) * In;
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = 0.f;
        if constexpr (VEC == 4) {
            for (int k = 4 * l16; k < In; k += 64) {
                const f32x4 xv = *(const f32x4*)(xr + k);
#pragma unroll
                for (int o = 0; o < 8; ++o) {
                    const f32x4 wv = *(const f32x4*)&Wt[o * In + k];
                    acc[o] = fmaf(xv[0], wv[0], fmaf(xv[1], wv[1], fmaf(xv[2], wv[2], fmaf(xv[3], wv[3], acc[o]))));
                }
            }
        } else {
            for (int k0 = 0; k0 < In; k0 += 64) {   // four elements per lane in flight
                float xv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 16 * u + l16;
                    xv[u] = xr[k < In ? k : In - 1];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int k = k0 + 16 * u + l16;
                    const float xm = k < In ? xv[u] : 0.f;
                    const int kc = k < In ? k : In - 1;
#pragma unroll
                    for (int o = 0; o < 8; ++o) acc[o] = fmaf(xm, Wt[o * In + kc], acc[o]);
                }
            }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc[o] += __shfl_xor(acc[o], m);
        float mine = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) mine = (l16 == o) ? acc[o] : mine;
        if (ok && l16 < Out) {
            float v = mine + (b ? b[l16] : 0.f);
            if (add) v += add[(size_t)row * add_stride + l16];
            if (activation == 1) v = tanh_f(v);
            y[(size_t)row * Out + l16] = v;
        }
    }
}

// Few rows, a wide input and more than eight outputs (the 400 -> 32 heads of lstm.py:321-337 on a batch of 32 final states):
// the wave-per-row kernel above walks In / 64 dependent rounds of Out strided loads - 34 us for a 32 x 512 x 32 product.
// Here a workgroup takes four rows; thread (o, kq) contracts an eighth of In for output o of all four rows: its W loads are
// independent and coalesced over o (issued back to back), x comes from LDS as a broadcast, the eight partial sums meet in LDS.
constexpr int DW_ROWS = 4, DW_KQ = 8, DW_OT = 32;
__global__ __launch_bounds__(256) void dense_fewrows_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                            const float* __restrict__ b, const float* __restrict__ add,
                                                            long add_stride, float* __restrict__ y, int N, int In, int Out,
                                                            int activation) {
    extern __shared__ __attribute__((aligned(16))) float dw_lds[];   // [DW_ROWS][In] rows of x, then [DW_KQ][DW_ROWS][DW_OT] partials
    float* xs = dw_lds;
    float* red = dw_lds + DW_ROWS * In;
    const int row0 = blockIdx.x * DW_ROWS;
    for (int e = threadIdx.x; e < DW_ROWS * In; e += 256) {
        const int r = e / In, k = e - r * In;
        xs[e] = (row0 + r < N) ? x[(size_t)(row0 + r) * In + k] : 0.f;
    }
    __syncthreads();
    const int ol = threadIdx.x & (DW_OT - 1), kq = threadIdx.x / DW_OT;
    const int kc = (In + DW_KQ - 1) / DW_KQ;
    const int k0 = kq * kc, k1 = (k0 + kc < In) ? k0 + kc : In;
    for (int o0 = 0; o0 < Out; o0 += DW_OT) {
        const int o = o0 + ol;
        float acc[DW_ROWS] = {0.f, 0.f, 0.f, 0.f};
        if (o < Out) {
            int k = k0;
            for (; k + 8 <= k1; k += 8) {
                float w[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(k + u) * Out + o];
#pragma unroll
                for (int u = 0; u < 8; ++u)
#pragma unroll
                    for (int r = 0; r < DW_ROWS; ++r) acc[r] = fmaf(xs[r * In + k + u], w[u], acc[r]);
            }
            for (; k < k1; ++k) {
                const float w = W[(size_t)k * Out + o];
#pragma unroll
                for (int r = 0; r < DW_ROWS; ++r) acc[r] = fmaf(xs[r * In + k], w, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < DW_ROWS; ++r) red[(kq * DW_ROWS + r) * DW_OT + ol] = acc[r];
        __syncthreads();
        if (threadIdx.x < DW_ROWS * DW_OT) {
            const int r = threadIdx.x / DW_OT, row = row0 + r;
            float v = 0.f;
#pragma unroll
            for (int q = 0; q < DW_KQ; ++q) v += red[(q * DW_ROWS + r) * DW_OT + ol];   // fixed order
            if (row < N && o < Out) {
                v += b ? b[o] : 0.f;
                if (add) v += add[(size_t)row * add_stride + o];
                if (activation == 1) v = tanh_f(v);
                y[(size_t)row * Out + o] = v;
            }
        }
        __syncthreads();
    }
}

static int launch_dense(const float* x, const float* W, const float* b, const float* add, long add_stride, float* y, int N,
                        int In, int Out, int activation, hipStream_t stream) {
    if (Out > 8 && N <= 256 && In >= 128 && In <= 2048) {
        const size_t lds = sizeof(float) * ((size_t)DW_ROWS * In + DW_KQ * DW_ROWS * DW_OT);
        hipLaunchKernelGGL(dense_fewrows_kernel, dim3((unsigned)((N + DW_ROWS - 1) / DW_ROWS)), dim3(256), lds, stream, x, W, b, add,
                           add_stride, y, N, In, Out, activation);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) { set_error("dense launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
        return FOV_OK;
    }
    if (Out <= 8 && In <= 2048 && N >= 64) {
        long blocks = ((long)N + 15) / 16;
        if (blocks > 2048) blocks = 2048;
        if ((In & 3) == 0 && (((uintptr_t)x) & 15) == 0)
            hipLaunchKernelGGL(dense_small_kernel<4>, dim3((unsigned)blocks), dim3(256), sizeof(float) * 8 * In, stream, x, W, b, add,
                               add_stride, y, N, In, Out, activation);
        else
            hipLaunchKernelGGL(dense_small_kernel<1>, dim3((unsigned)blocks), dim3(256), sizeof(float) * 8 * In, stream, x, W, b, add,
                               add_stride, y, N, In, Out, activation);
    } else {
        hipLaunchKernelGGL(dense_kernel, dim3((N + 3) / 4), dim3(256), 0, stream, x, W, b, add, add_stride, y, N, In, Out,
                           activation);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("dense launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// ---------------------------------------------------------------------------------------
// Mixing head of the others model, one decoder step (given_others...py:127-130,168,257-265):
//   p = tanh(h W_d + b_d)                     Dense(O,'tanh') on the layer-2 output
//   m = tanh(p W_p + add)                     mixing Dense: W_p = mix_W[-O:], add = others_t . mix_W[:-O] + mix_b
// Forward: 16 lanes per row as dense_small_kernel, then the O values of p are exchanged inside the 16-lane group.
// Backward (one thread per (row, hidden unit); the O-wide row quantities are recomputed by every thread of the
// row - 2*O*O FMAs): dpre_m = dm_loss + dm_feedback * (1 - m^2), dpre_p = (dpre_m W_p^T) * (1 - p^2),
// dh = dpre_p W_d^T.  One launch each instead of two / five: the step-wise decoder is launch-bound.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mix_head_fwd_kernel(const float* __restrict__ h, const float* __restrict__ Wd,
                                                           const float* __restrict__ bd, const float* __restrict__ Wp,
                                                           const float* __restrict__ add, long add_stride, float* __restrict__ p_out,
                                                           float* __restrict__ m_out, int N, int H, int O) {
    extern __shared__ __attribute__((aligned(16))) float Wt[];   // [8][H] transposed W_d, then W_p (O x O)
    float* sWp = Wt + 8 * H;
    for (int e = threadIdx.x; e < 8 * H; e += 256) {
        const int o = e / H, k = e - o * H;
        Wt[e] = (o < O) ? Wd[(size_t)k * O + o] : 0.f;
    }
    for (int e = threadIdx.x; e < 64; e += 256) sWp[e] = (e < O * O) ? Wp[e] : 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, l16 = lane & 15, g = lane >> 4, wave = threadIdx.x >> 6;
    for (long row0 = ((long)blockIdx.x * 4 + wave) * 4; row0 < N; row0 += (long)gridDim.x * 16) {
        const long row = row0 + g;
        const bool ok = row < N;
        const float* xr = h + (size_t)(ok ? row : 0) * H;
        float acc[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) acc[o] = 0.f;
        for (int k = 4 * l16; k < H; k += 64) {
            const f32x4 xv = *(const f32x4*)(xr + k);
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                const f32x4 wv = *(const f32x4*)&Wt[o * H + k];
                acc[o] = fmaf(xv[0], wv[0], fmaf(xv[1], wv[1], fmaf(xv[2], wv[2], fmaf(xv[3], wv[3], acc[o]))));
            }
        }
#pragma unroll
        for (int o = 0; o < 8; ++o)
#pragma unroll
            for (int m = 8; m >= 1; m >>= 1) acc[o] += __shfl_xor(acc[o], m);
        // every lane of the group now holds all eight sums: p[o] = tanh(acc[o] + bd[o])
        float pv[8];
#pragma unroll
        for (int o = 0; o < 8; ++o) pv[o] = (o < O) ? tanh_f(acc[o] + bd[o]) : 0.f;
        if (ok && l16 < O) {
            float pm = 0.f, z = add[(size_t)row * add_stride + l16];
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                pm = (l16 == o) ? pv[o] : pm;
                z = fmaf(pv[o], sWp[o * O + l16], z);   // rows o >= O of sWp are never read past O*O: pv is 0 there
            }
            p_out[(size_t)row * O + l16] = pm;
            m_out[(size_t)row * O + l16] = tanh_f(z);
        }
    }
}

__global__ __launch_bounds__(256) void mix_head_bwd_kernel(const float* __restrict__ dm_loss, const float* __restrict__ dm_fb,
                                                           const float* __restrict__ m, const float* __restrict__ p,
                                                           const float* __restrict__ Wp, const float* __restrict__ Wd,
                                                           float* __restrict__ dpre_m, float* __restrict__ dpre_p,
                                                           float* __restrict__ dh, int N, int H, int O) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)N * H) return;
    const long row = idx / H;
    const int j = (int)(idx - row * H);
    float dm[8], dp[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        dm[o] = 0.f;
        if (o < O) {
            const float mv = m[row * O + o];
            dm[o] = dm_loss[row * O + o] + (dm_fb ? dm_fb[row * O + o] * (1.f - mv * mv) : 0.f);
        }
    }
    float out = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        dp[i] = 0.f;
        if (i < O) {
            float s = 0.f;
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < O) s = fmaf(dm[o], Wp[i * O + o], s);
            const float pv = p[row * O + i];
            dp[i] = s * (1.f - pv * pv);
            out = fmaf(dp[i], Wd[(size_t)j * O + i], out);
        }
    }
    dh[idx] = out;
    if (j < O) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int o = 0; o < 8; ++o) {
            a = (j == o) ? dm[o] : a;
            b = (j == o) ? dp[o] : b;
        }
        dpre_m[row * O + j] = a;
        dpre_p[row * O + j] = b;
    }
}

// ---------------------------------------------------------------------------------------
// mu / sigma^2 feature op (utility.py:483-517): one thread per (row, axis); two-pass
// population variance like numpy.var (mean first, then mean of squared deviations).
// HBM-bound: 12*fps bytes in, 24 bytes out per row.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void meanvar_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                      long rows, int fps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * 3) return;
    const long row = i / 3;
    const int ax = (int)(i - row * 3);
    const float* p = y + (size_t)row * 3 * fps + ax;
    float s = 0.f;
    for (int f = 0; f < fps; ++f) s += p[3 * f];
    const float mean = s / (float)fps;
    float v = 0.f;
    for (int f = 0; f < fps; ++f) {
        const float d = p[3 * f] - mean;
        v = fmaf(d, d, v);
    }
    out[row * 6 + ax] = mean;
    out[row * 6 + 3 + ax] = v / (float)fps;
}

// ---------------------------------------------------------------------------------------
// FoV hit rate per predicted second (the consumer of the path's output):
// xyz -> (theta, phi) as dataIO.py:77-82, +-2pi seam fix as baseline_knn_mean.py:78-85, overlap of the two
// span x span boxes over the ground-truth box area as :62-82.  One thread per (sequence, second); HBM-bound.
// ---------------------------------------------------------------------------------------
__device__ __forceinline__ float fmod_pos(float a, float m) {
    float r = fmodf(a, m);
    return r < 0.f ? r + m : r;
}

__global__ __launch_bounds__(256) void hit_rate_kernel(const float* __restrict__ pred, long pred_stride,
                                                       const float* __restrict__ gt, long gt_stride,
                                                       float* __restrict__ out, long rows, float span, float gt_span) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows) return;
    const float kPi = 3.14159265358979323846f;
    const float* p = pred + i * pred_stride;
    const float* g = gt + i * gt_stride;
    float pt = fmod_pos(atan2f(p[1], p[0]), 2.f * kPi) - kPi;
    const float pp = fmod_pos(atan2f(p[2], sqrtf(p[0] * p[0] + p[1] * p[1])) + 0.5f * kPi, kPi);
    float gth = fmod_pos(atan2f(g[1], g[0]), 2.f * kPi) - kPi;
    const float gp = fmod_pos(atan2f(g[2], sqrtf(g[0] * g[0] + g[1] * g[1])) + 0.5f * kPi, kPi);
    if (gth > 2.f / 3.f * kPi && pt < -2.f / 3.f * kPi) pt += 2.f * kPi;
    else if (gth < -2.f / 3.f * kPi && pt > 2.f / 3.f * kPi) gth += 2.f * kPi;
    const float iw = fminf(pt + 0.5f * span, gth + 0.5f * gt_span) - fmaxf(pt - 0.5f * span, gth - 0.5f * gt_span);
    const float ih = fminf(pp + 0.5f * span, gp + 0.5f * gt_span) - fmaxf(pp - 0.5f * span, gp - 0.5f * gt_span);
    out[i] = (iw > 0.f && ih > 0.f) ? iw * ih / (gt_span * gt_span) : 0.f;
}

// ---------------------------------------------------------------------------------------
// Device-side windowing: reshape2second_stacks (mycode/utility.py:264-305) as one gather.
// x:(U,S,feat) seconds of one video -> enc / future / future_input windows of T seconds every `stride`
// seconds; collapse_user: window-major (W*U, T, feat), else user-major (U, W, T, feat).  HBM-bound copy.
// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void window_stacks_kernel(const float* __restrict__ x, float* __restrict__ enc,
                                                            float* __restrict__ fut, float* __restrict__ fut_in, int U,
                                                            int S, int feat, int T, int stride, int shift, int W,
                                                            int collapse) {
    const long total = (long)W * U * T * feat;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int f = (int)(i % feat);
    long r = i / feat;
    const int t = (int)(r % T);
    r /= T;
    int w, u;
    if (collapse) { u = (int)(r % U); w = (int)(r / U); } else { w = (int)(r % W); u = (int)(r / W); }
    const float* xu = x + (size_t)u * S * feat;
    const int s_enc = stride * w + t, s_fut = stride * (w + shift) + t;
    enc[i] = xu[(size_t)s_enc * feat + f];
    fut[i] = xu[(size_t)s_fut * feat + f];
    // decoder input = future shifted right by one second, seeded with the encoder's last second
    fut_in[i] = (t == 0) ? xu[(size_t)(stride * w + T - 1) * feat + f] : xu[(size_t)(s_fut - 1) * feat + f];
}

static bool want_cluster(int impl, int F, int H, int F_dec, bool decode) {
    if (impl == FOV_IMPL_GENERIC) return false;
    // impl = auto also needs one whole group co-resident (device_cu_count honours FOV_DBG_RESIDENT_LIMIT): otherwise generic
    const bool ok = cluster_shape_ok(F, H) && (!decode || (F_dec >= 1 && F_dec <= 8)) && device_cu_count() >= H / 64;
    return impl == FOV_IMPL_CLUSTER ? true : ok;
}

static int check_ws(void* ws, size_t have, size_t need) {
    if (need == 0) return FOV_OK;
    if (!ws || have < need) {
        set_error("workspace too small: need %zu bytes, have %zu", need, ws ? have : (size_t)0);
        return FOV_ERR_WORKSPACE;
    }
    if (((uintptr_t)ws & 15) != 0) {
        set_error("workspace must be 16-byte aligned");
        return FOV_ERR_WORKSPACE;
    }
    return FOV_OK;
}

// ---- cached environment knobs ----
static EnvKnobs g_env;
static std::once_flag g_env_once;
static int env_flag(const char* name) { const char* e = getenv(name); return (e && e[0] == '1') ? 1 : 0; }
void env_reload() {
    g_env.force_safe_exchange = env_flag("FOV_FORCE_SAFE_EXCHANGE");
    g_env.two_launches = getenv("FOV_TWO_LAUNCHES") ? 1 : 0;
    const char* lim = getenv("FOV_DBG_RESIDENT_LIMIT");
    g_env.resident_limit = lim ? atoi(lim) : 0;
    g_env.no_cell_patch = env_flag("FOV_NO_CELL_PATCH");
    g_env.no_conv_patch = env_flag("FOV_NO_CONV_PATCH");
    g_env.no_wide16 = env_flag("FOV_NO_WIDE16");
    g_env.pair = env_flag("FOV_PAIR");
    g_env.no_xcd_pad = getenv("FOV_NO_XCD_PAD") ? 1 : 0;
    { const char* pm = getenv("FOV_XCD_PAD_MAX"); g_env.xcd_pad_max = pm ? atoi(pm) : 16; }
    g_env.no_bwd16_narrow = getenv("FOV_NO_BWD16_NARROW") ? 1 : 0;
    { const char* bg = getenv("FOV_BWD16_GROUPS"); g_env.bwd16_groups32 = (bg && atoi(bg) == 32) ? 1 : 0; }
    g_env.no_stack2 = env_flag("FOV_NO_STACK2");
    g_env.bwd_stepped = getenv("FOV_BWD_STEPPED") ? 1 : 0;
    g_env.no_wgrad_fusion = getenv("FOV_NO_WGRAD_FUSION") ? 1 : 0;
    g_env.no_dx_fusion = getenv("FOV_NO_DX_FUSION") ? 1 : 0;
    g_env.bwd_groups4 = getenv("FOV_BWD_GROUPS4") ? 1 : 0;
    g_env.gemm_bf16_noremap = getenv("FOV_GEMM_BF16_NOREMAP") ? 1 : 0;
    g_env.gemm_bf16_shallow = getenv("FOV_GEMM_BF16_SHALLOW") ? 1 : 0;
    g_env.no_wgrad_group = getenv("FOV_NO_WGRAD_GROUP") ? 1 : 0;
    g_env.no_wgrad_lines = getenv("FOV_NO_WGRAD_LINES") ? 1 : 0;
    g_env.no_wide16_trio = getenv("FOV_NO_WIDE16_TRIO") ? 1 : 0;
    g_env.dbg_trace = getenv("FOV_DBG_TRACE") ? 1 : 0;
    const char* gb = getenv("FOV_GEMM_BF16_SPLIT");
    g_env.gemm_bf16_split = gb ? atoi(gb) : 0;
    const char* gv = getenv("FOV_GEMM_VARIANT");
    g_env.gemm_variant = gv ? atoi(gv) : 0;
    const char* gs = getenv("FOV_GEMM_SPLIT");
    g_env.gemm_split = gs ? atoi(gs) : 0;
}
const EnvKnobs& env_knobs() {
    std::call_once(g_env_once, env_reload);
    return g_env;
}

// ---- host-side epoch accounting per workspace ----
struct XchState { unsigned long long epoch; int force_safe; };
static std::unordered_map<void*, XchState> g_xch;
static std::mutex g_xch_mu;
int xch_account(void* workspace, long span, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_xch_mu);
    XchState& st = g_xch[workspace];
    st.epoch += (unsigned long long)(span > 0 ? span : 0) + 2ull;
    if (st.epoch > 0x70000000ull) {
        // every launch on this workspace passes here, so the device-side base is at most st.epoch: re-zero the header and the
        // granule area in front of this launch (stream-ordered; launches on one workspace are serialised by contract)
        // ... all of it but the sticky timeout word (header word 0): a give-up since the last fov_check_status must stay visible
        // to the guarded optimizer and to the next check
        static_assert(ST_TIMEOUT == 0, "the re-zero below skips header word 0");
        hipError_t e = hipMemsetAsync((char*)workspace + sizeof(unsigned), 0, kStatusBytes + kXchBytes - sizeof(unsigned), stream);
        if (e == hipSuccess && st.force_safe)
            e = hipMemsetD32Async((hipDeviceptr_t)((unsigned*)workspace + ST_FORCE_SAFE), 1, 1, stream);
        if (e != hipSuccess) { set_error("epoch re-zero: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
        st.epoch = (unsigned long long)(span > 0 ? span : 0) + 2ull;
    }
    return FOV_OK;
}
void xch_forget(void* workspace) {
    std::lock_guard<std::mutex> lock(g_xch_mu);
    XchState& st = g_xch[workspace];
    st.epoch = 0;
}
void xch_note_force_safe(void* workspace, int on) {
    std::lock_guard<std::mutex> lock(g_xch_mu);
    g_xch[workspace].force_safe = on ? 1 : 0;
}
// test hook: pretend `epoch` epochs have been consumed on this workspace (the device header is set to match by the caller)
void xch_set_epoch_for_test(void* workspace, unsigned long long epoch) {
    std::lock_guard<std::mutex> lock(g_xch_mu);
    g_xch[workspace].epoch = epoch;
}

static std::unordered_map<void*, const float*> g_prepacked;
static std::mutex g_prepacked_mu;
void prepack_mark(void* workspace, const float* K2) {
    std::lock_guard<std::mutex> lock(g_prepacked_mu);
    g_prepacked[workspace] = K2;
}
bool prepack_consume(void* workspace, const float* K2) {
    std::lock_guard<std::mutex> lock(g_prepacked_mu);
    auto it = g_prepacked.find(workspace);
    if (it == g_prepacked.end()) return false;
    const bool hit = it->second == K2;
    g_prepacked.erase(it);
    return hit;
}
// the workspace was zero-filled (fov_workspace_init, the reset of fov_check_status): a packed copy marked earlier is gone
void prepack_forget(void* workspace) {
    std::lock_guard<std::mutex> lock(g_prepacked_mu);
    g_prepacked.erase(workspace);
}

}  // namespace fov

using namespace fov;

extern "C" {

const char* fov_last_error(void) { return g_err; }

void fov_reload_env(void) { env_reload(); }

int64_t fov_debug_generic_launches(void) { return (int64_t)generic_launch_count(); }

int fov_debug_set_epoch(void* workspace, size_t workspace_bytes, unsigned epoch, fov_stream_t stream) {
    if (!workspace || workspace_bytes < kStatusBytes) { set_error("fov_debug_set_epoch: invalid workspace"); return FOV_ERR_INVALID; }
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)((unsigned*)workspace + ST_EPOCH), (int)epoch, 1, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("fov_debug_set_epoch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    xch_set_epoch_for_test(workspace, epoch);
    return FOV_OK;
}
int fov_version(void) { return 100; }

int fov_cluster_supported(int F, int H) { return cluster_shape_ok(F, H) ? 1 : 0; }

// width 512 (lstm_wide.hip): header + granule area, and for F > 96 the input projection (B,T,4H) + GEMM scratch behind them
static constexpr size_t kWide512ScratchFloats = ((size_t)1 << 20) + 64;
static size_t wide512_workspace_bytes(int B, int T, int F) {
    size_t n = cluster_workspace_bytes(B, 512);
    if (F > 96) n += sizeof(float) * ((size_t)B * T * 2048 + kWide512ScratchFloats);
    return n;
}

size_t fov_lstm_seq_workspace_bytes(int B, int T, int F, int H, int impl) {
    if (B <= 0) return kStatusBytes;
    if (impl != FOV_IMPL_GENERIC && wide512_shape_ok(F, H, F > 96)) return wide512_workspace_bytes(B, T, F);
    if (impl == FOV_IMPL_AUTO && !env_knobs().no_wide16 && wide16_shape(B, F, H)) return cluster_workspace_bytes(B, H);
    if (impl == FOV_IMPL_AUTO && stepwise_preferred(B, F, H)) return kStatusBytes + sizeof(float) * stepwise_workspace_floats(B, T, H);
    if (impl != FOV_IMPL_GENERIC && wide_shape_ok(F, H)) return cluster_workspace_bytes(B, H);
    if (impl == FOV_IMPL_AUTO && wide_narrow_preferred(B, F, H)) return cluster_workspace_bytes(B, H);
    return want_cluster(impl, F, H, 0, false) && cluster_shape_ok(F, H) ? cluster_workspace_bytes(B, H) : kStatusBytes;
}

static int lstm_seq_fwd_impl(const float* x, const float* K, const float* R, const float* b, const float* h0,
                             const float* c0, float* hs, float* hT, float* cT, float* reserve, int B, int T, int F,
                             int H, int act, int impl, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || H <= 0 || !K || !R || !b || (B > 0 && T > 0 && !x) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_seq_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_lstm_seq_workspace_bytes(B, T, F, H, impl));
    if (rc) return rc;
    LstmParams p = {};
    p.x = x; p.K = K; p.R = R; p.b = b; p.h0 = h0; p.c0 = c0; p.hs = hs; p.hT = hT; p.cT = cT;
    p.reserve = reserve;
    p.B = B; p.T = T; p.F = F; p.H = H; p.act = act;
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    hipStream_t s = (hipStream_t)stream;
    // width 512 (mycode/lstm.py's LSTMCell(400), zero-padded by the caller): R register-resident over 16 workgroups per tile;
    // an input wider than 96 is projected first (one GEMM over all steps), the kernel adds it per step
    if (impl != FOV_IMPL_GENERIC && !env_knobs().no_wide16 && wide16_preferred(x, B, F, H) && wide512_shape_ok(F, H, F > 96))
        return launch_wide16(p, s);   // few tiles: 32 workgroups per tile, K and R both in registers (lstm_wide16.hip)
    if (impl != FOV_IMPL_GENERIC && wide512_shape_ok(F, H, F > 96)) {
        if (F > 96) {
            float* zx = (float*)((char*)workspace + cluster_workspace_bytes(B, H));
            if (B > 0 && T > 0) {
                rc = matmul_f32(x, K, zx, B * T, F, 4 * H, zx + (size_t)B * T * 4 * H, kWide512ScratchFloats, s);
                if (rc) return rc;
            }
            p.zx = zx;
        }
        return launch_wide(p, s);
    }
    // few tiles at width 128 / 256: H / 16 workgroups per tile (lstm_wide16.hip) - the latency regime of model.fit at batch 32
    if (impl == FOV_IMPL_AUTO && H != 512 && T > 0 && !env_knobs().no_wide16 && wide16_preferred(x, B, F, H) &&
        workspace_bytes >= cluster_workspace_bytes(B, H))
        return launch_wide16(p, s);
    // wide inputs (a stacked layer over a 256-wide sequence): K and R both register-resident (lstm_wide.hip)
    if (impl != FOV_IMPL_GENERIC && wide_shape_ok(F, H) && (((uintptr_t)x) & 15) == 0) return launch_wide(p, s);
    // narrow inputs, at most 32 tiles: groups of eight workgroups fill the chip where lstm_cluster's groups of four leave half idle
    if (impl == FOV_IMPL_AUTO && T > 0 && wide_narrow_preferred(B, F, H)) return launch_wide(p, s);
    if (want_cluster(impl, F, H, 0, false)) return launch_cluster(p, false, s);
    // widths above the persistent kernels': step-wise on the matrix-core GEMM instead of the VALU kernel
    if (impl == FOV_IMPL_AUTO && stepwise_preferred(B, F, H))
        return launch_stepwise(p, (float*)((char*)workspace + kStatusBytes), (workspace_bytes - kStatusBytes) / sizeof(float), s);
    return launch_generic(p, false, s);
}

int fov_lstm_seq_fwd(const float* x, const float* K, const float* R, const float* b, const float* h0,
                     const float* c0, float* hs, float* hT, float* cT, int B, int T, int F, int H, int act,
                     int impl, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    return lstm_seq_fwd_impl(x, K, R, b, h0, c0, hs, hT, cT, nullptr, B, T, F, H, act, impl, workspace, workspace_bytes,
                             stream);
}

int fov_lstm_seq_fwd_train(const float* x, const float* K, const float* R, const float* b, const float* h0,
                           const float* c0, float* hs, float* hT, float* cT, float* reserve, int B, int T, int F,
                           int H, int act, int impl, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (!reserve && B > 0 && T > 0) {
        set_error("fov_lstm_seq_fwd_train: reserve is NULL");
        return FOV_ERR_INVALID;
    }
    return lstm_seq_fwd_impl(x, K, R, b, h0, c0, hs, hT, cT, reserve, B, T, F, H, act, impl, workspace, workspace_bytes,
                             stream);
}

int fov_lstm_seq_fwd_bf16(const float* x, const float* K, const float* R, const float* b, const float* h0, const float* c0,
                          float* hs, float* hT, float* cT, float* reserve, int B, int T, int F, int H, int act,
                          void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || H <= 0 || !K || !R || !b || (B > 0 && T > 0 && !x) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_seq_fwd_bf16: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (!layer_bf16_shape_ok(F, H)) { set_error("fov_lstm_seq_fwd_bf16: H = 256 and F <= 256 only"); return FOV_ERR_UNSUPPORTED; }
    int rc = check_ws(workspace, workspace_bytes, B > 0 ? kStatusBytes + kXchBytes : kStatusBytes);
    if (rc) return rc;
    LstmParams p = {};
    p.x = x; p.K = K; p.R = R; p.b = b; p.h0 = h0; p.c0 = c0; p.hs = hs; p.hT = hT; p.cT = cT; p.reserve = reserve;
    p.B = B; p.T = T; p.F = F; p.H = H; p.act = act;
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    return launch_layer_bf16(p, (hipStream_t)stream);
}

int fov_lstm_stack2_supported_bf16(int B, int T, int F, int H) { return stack2_bf16_shape_ok(B, T, F, H) ? 1 : 0; }

int fov_lstm_stack2_fwd_bf16(const float* x, const float* K1, const float* R1, const float* b1, const float* K2, const float* R2,
                             const float* b2, float* hs1, float* hT1, float* cT1, float* reserve1, float* hs2, float* hT2,
                             float* cT2, float* reserve2, int B, int T, int F, int H, int act, void* workspace,
                             size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || !K1 || !R1 || !b1 || !K2 || !R2 || !b2 || (B > 0 && T > 0 && !x) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_stack2_fwd_bf16: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (B == 0) return FOV_OK;
    if (!stack2_bf16_shape_ok(B, T, F, H)) {
        set_error("fov_lstm_stack2_fwd_bf16: H = 256, F <= 96, T >= 2 and one 16-sequence tile per group of eight workgroups only");
        return FOV_ERR_UNSUPPORTED;
    }
    int rc = check_ws(workspace, workspace_bytes, kStatusBytes + kXchBytes);
    if (rc) return rc;
    return launch_stack2_bf16(x, K1, R1, b1, K2, R2, b2, hs1, hT1, cT1, reserve1, hs2, hT2, cT2, reserve2, B, T, F, act, workspace,
                              (hipStream_t)stream);
}

int fov_lstm_stack2_supported(int B, int T, int F, int H) { return wide16_pair_shape(B, T, F, H) ? 1 : 0; }

int fov_lstm_stack2_fwd(const float* x, const float* K1, const float* R1, const float* b1, const float* h0_1, const float* c0_1,
                        const float* K2, const float* R2, const float* b2, const float* h0_2, const float* c0_2, float* hs1,
                        float* hT1, float* cT1, float* reserve1, float* hs2, float* hT2, float* cT2, float* reserve2, int B, int T,
                        int F, int H, int act, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || !K1 || !R1 || !b1 || !K2 || !R2 || !b2 || (B > 0 && T > 0 && !x) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_stack2_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (B == 0) return FOV_OK;
    if (!wide16_pair_shape(B, T, F, H)) {
        set_error("fov_lstm_stack2_fwd: H = 512, F <= 96, T >= 2 and both layers' groups resident (<= 64 sequences on 256 CUs) only");
        return FOV_ERR_UNSUPPORTED;
    }
    int rc = check_ws(workspace, workspace_bytes, kStatusBytes + kXchBytes);
    if (rc) return rc;
    LstmParams a = {}, b = {};
    a.x = x; a.K = K1; a.R = R1; a.b = b1; a.h0 = h0_1; a.c0 = c0_1; a.hs = hs1; a.hT = hT1; a.cT = cT1; a.reserve = reserve1;
    b.x = nullptr; b.K = K2; b.R = R2; b.b = b2; b.h0 = h0_2; b.c0 = c0_2; b.hs = hs2; b.hT = hT2; b.cT = cT2; b.reserve = reserve2;
    a.B = b.B = B; a.T = b.T = T; a.F = F; b.F = H; a.H = b.H = H; a.act = b.act = act;
    a.status = b.status = (unsigned*)workspace;
    a.xch = b.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    return launch_wide16_pair(a, b, (hipStream_t)stream);
}

int fov_lstm_seq_fwd_zx(const float* zx, const float* R, const float* b, const float* h0, const float* c0, float* hs,
                        float* hT, float* cT, float* reserve, int B, int T, int H, int act, int impl, void* workspace,
                        size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || H <= 0 || !R || !b || (B > 0 && T > 0 && !zx) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_seq_fwd_zx: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_lstm_seq_workspace_bytes(B, T, 1, H, impl));
    if (rc) return rc;
    LstmParams p = {};
    p.zx = zx; p.R = R; p.b = b; p.h0 = h0; p.c0 = c0; p.hs = hs; p.hT = hT; p.cT = cT; p.reserve = reserve;
    p.B = B; p.T = T; p.F = 1; p.H = H; p.act = act;
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    hipStream_t s = (hipStream_t)stream;
    if (impl != FOV_IMPL_GENERIC && wide512_shape_ok(1, H, true)) return launch_wide(p, s);
    if (want_cluster(impl, 1, H, 0, false)) return launch_cluster(p, false, s);
    return launch_generic(p, false, s);
}

size_t fov_lstm_seq_bwd_workspace_bytes(int B, int T, int F, int H) {
    if (B <= 0 || T < 0 || F <= 0 || H <= 0) return 256;
    return sizeof(float) * lstm_bwd_workspace_floats(B, T, F, H);
}

int fov_lstm_seq_bwd(const float* x, const float* K, const float* R, const float* h0, const float* c0,
                     const float* hs, const float* reserve, const float* dhs, const float* dhT, const float* dcT,
                     float* dz, float* dx, float* dK, float* dR, float* db, float* dh0, float* dc0, int B, int T,
                     int F, int H, int act, int accumulate, void* workspace, size_t workspace_bytes,
                     fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || H <= 0 || !K || !R ||
        (B > 0 && T > 0 && (!x || !hs || !reserve || !dz)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_seq_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (B == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, fov_lstm_seq_bwd_workspace_bytes(B, T, F, H));
    if (rc) return rc;
    return lstm_seq_bwd(x, K, R, h0, c0, hs, reserve, dhs, dhT, dcT, dz, dx, dK, dR, db, dh0, dc0, B, T, F, H, act,
                        accumulate, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream, 0);
}

int fov_lstm_seq_bwd_bf16(const float* x, const float* K, const float* R, const float* h0, const float* c0,
                          const float* hs, const float* reserve, const float* dhs, const float* dhT, const float* dcT,
                          float* dz, float* dx, float* dK, float* dR, float* db, float* dh0, float* dc0, int B, int T,
                          int F, int H, int act, int accumulate, void* workspace, size_t workspace_bytes,
                          fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || H != 256 || !K || !R ||
        (B > 0 && T > 0 && (!x || !hs || !reserve || !dz)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_seq_bwd_bf16: invalid argument (H = 256 only)");
        return FOV_ERR_INVALID;
    }
    if (B == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, fov_lstm_seq_bwd_workspace_bytes(B, T, F, H));
    if (rc) return rc;
    return lstm_seq_bwd(x, K, R, h0, c0, hs, reserve, dhs, dhT, dcT, dz, dx, dK, dR, db, dh0, dc0, B, T, F, H, act,
                        accumulate, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream, 1);
}

int fov_lstm_stack2_bwd_supported(int B, int T, int F, int H) { return (F > 0 && bwd16_pair_shape(B, T, H)) ? 1 : 0; }

size_t fov_lstm_stack2_bwd_workspace_bytes(int B, int T, int F, int H) {
    if (B <= 0 || T < 0 || F <= 0 || H <= 0) return 256;
    return sizeof(float) * lstm_stack2_bwd_workspace_floats(B, T, F, H);
}

int fov_lstm_stack2_bwd(const float* x, const float* R1, const float* K2, const float* R2, const float* h0_1, const float* c0_1,
                        const float* h0_2, const float* c0_2, const float* hs1, const float* reserve1, const float* hs2,
                        const float* reserve2, const float* dhs2, const float* dhT2, const float* dcT2, const float* dhT1,
                        const float* dcT1, float* dz1, float* dz2, float* dK1, float* dR1, float* db1, float* dK2, float* dR2,
                        float* db2, float* dh0_1, float* dc0_1, float* dh0_2, float* dc0_2, int B, int T, int F, int H, int act,
                        int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || H <= 0 || !R1 || !K2 || !R2 ||
        (B > 0 && T > 0 && (!x || !hs1 || !reserve1 || !hs2 || !reserve2 || !dz1 || !dz2)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_lstm_stack2_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (B == 0 || T == 0) return FOV_OK;
    if (!bwd16_pair_shape(B, T, H)) {
        set_error("fov_lstm_stack2_bwd: H = 512, at most 32 sequences, three role-groups of sixteen workgroups per tile resident only");
        return FOV_ERR_UNSUPPORTED;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_lstm_stack2_bwd_workspace_bytes(B, T, F, H));
    if (rc) return rc;
    return lstm_stack2_bwd(x, R1, K2, R2, h0_1, c0_1, h0_2, c0_2, hs1, reserve1, hs2, reserve2, dhs2, dhT2, dcT2, dhT1, dcT1, dz1, dz2, dK1,
                           dR1, db1, dK2, dR2, db2, dh0_1, dc0_1, dh0_2, dc0_2, B, T, F, H, act, accumulate, (float*)workspace,
                           workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_stream_create(int priority, fov_stream_t* stream) {
    if (!stream) { set_error("fov_stream_create: stream is NULL"); return FOV_ERR_INVALID; }
    int least = 0, greatest = 0;     // numerically: least = lowest priority (largest value), greatest = highest (smallest value)
    hipError_t e = hipDeviceGetStreamPriorityRange(&least, &greatest);
    if (e != hipSuccess) { set_error("hipDeviceGetStreamPriorityRange: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    int p = priority > 0 ? least : (priority < 0 ? greatest : 0);
    if (p > least) p = least;
    if (p < greatest) p = greatest;
    hipStream_t s = nullptr;
    e = hipStreamCreateWithPriority(&s, hipStreamNonBlocking, p);
    if (e != hipSuccess) { set_error("hipStreamCreateWithPriority(%d): %s", p, hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    *stream = (fov_stream_t)s;
    return FOV_OK;
}

int fov_stream_destroy(fov_stream_t stream) {
    if (!stream) return FOV_OK;
    hipError_t e = hipStreamDestroy((hipStream_t)stream);
    if (e != hipSuccess) { set_error("hipStreamDestroy: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int fov_lstm_seq_wgrad(const float* x, const float* hs, const float* h0, const float* dz, float* dK, float* dR, float* db, int B,
                       int T, int F, int H, int accumulate, int bf16, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T < 0 || F <= 0 || H <= 0 || (B > 0 && T > 0 && (!dz || (dK && !x) || (dR && !hs))) || (bf16 && H != 256)) {
        set_error("fov_lstm_seq_wgrad: invalid argument (bf16 operands: H = 256 only)");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_lstm_seq_bwd_workspace_bytes(B, T, F, H));
    if (rc) return rc;
    return lstm_seq_wgrad(x, hs, h0, dz, dK, dR, db, B, T, F, H, accumulate, bf16 ? 1 : 0, (float*)workspace,
                          workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_lstm_seq_wgrad_pair_one_launch(int B, int T1, int T2, int H) {
    return (B > 0 && T1 > 0 && T2 > 0 && H > 0 && lstm_seq_wgrad_pair_one_launch(B, T1, T2, H)) ? 1 : 0;
}

int fov_lstm_seq_wgrad_pair(const float* x1, const float* hs1, const float* h0_1, const float* dz1, float* dK1, float* dR1, float* db1,
                            int T1, int F1, const float* x2, const float* hs2, const float* h0_2, const float* dz2, float* dK2,
                            float* dR2, float* db2, int T2, int F2, int B, int H, int accumulate, void* workspace,
                            size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T1 < 0 || T2 < 0 || F1 <= 0 || F2 <= 0 || H <= 0 || (B > 0 && T1 > 0 && (!dz1 || (dK1 && !x1) || (dR1 && !hs1))) ||
        (B > 0 && T2 > 0 && (!dz2 || (dK2 && !x2) || (dR2 && !hs2)))) {
        set_error("fov_lstm_seq_wgrad_pair: invalid argument");
        return FOV_ERR_INVALID;
    }
    const size_t n1 = fov_lstm_seq_bwd_workspace_bytes(B, T1, F1, H), n2 = fov_lstm_seq_bwd_workspace_bytes(B, T2, F2, H);
    int rc = check_ws(workspace, workspace_bytes, n1 > n2 ? n1 : n2);
    if (rc) return rc;
    return lstm_seq_wgrad_pair(x1, hs1, h0_1, dz1, dK1, dR1, db1, T1, F1, x2, hs2, h0_2, dz2, dK2, dR2, db2, T2, F2, B, H, accumulate,
                               (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

size_t fov_dense_bwd_workspace_bytes(int N, int In, int Out) {
    if (N <= 0 || In <= 0 || Out <= 0) return 256;
    size_t a = (size_t)(Out <= 8 ? 1024 : 64) * In * Out, b = (size_t)256 * Out, c = (size_t)(N + 255) / 256 + 64;
    size_t m = a > b ? a : b;
    return sizeof(float) * ((m > c ? m : c) + 64);
}

int fov_dense_bwd(const float* x, const float* W, const float* dpre, float* dx, float* dW, float* db, int N, int In,
                  int Out, int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (N < 0 || In <= 0 || Out <= 0 || (N > 0 && (!x || !W || !dpre))) {
        set_error("fov_dense_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_dense_bwd_workspace_bytes(N, In, Out));
    if (rc) return rc;
    return dense_bwd(x, W, dpre, dx, dW, db, N, In, Out, accumulate, (float*)workspace, workspace_bytes / sizeof(float),
                     (hipStream_t)stream, 0);
}

size_t fov_wgrad_fused_workspace_bytes(int64_t N, int In1, int In2, int Out) {
    if (N <= 0 || In1 <= 0 || In2 < 0 || Out <= 0) return 256;
    const size_t a = fov_dense_bwd_workspace_bytes((int)N, In1 + In2 + 1, Out);
    const size_t b = sizeof(float) * gemm_bf16_tn_scratch_floats(In1 + In2 + 1, Out);
    return a > b ? a : b;
}

int fov_wgrad_fused(const float* x1, int In1, const float* x2, int In2, const float* dpre, float* out, int64_t N, int Out,
                    int bias, int accumulate, int bf16, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (N < 0 || In1 <= 0 || In2 < 0 || Out <= 0 || !out || (N > 0 && (!x1 || !dpre || (In2 > 0 && !x2))) || N > 0x7fffffffL) {
        set_error("fov_wgrad_fused: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_wgrad_fused_workspace_bytes(N, In1, In2, Out));
    if (rc) return rc;
    float* ws = (float*)workspace;
    const size_t wsf = workspace_bytes / sizeof(float);
    const float* a2 = In2 > 0 ? x2 : nullptr;
    if (N > 0 && wgrad_fusable(x1, In1, 0, In1, a2, In2, 0, In2, dpre, Out, 0, out, Out) && (!bf16 || Out >= 64))
        return wgrad_fused(x1, In1, 0, In1, 0, a2, In2, 0, In2, 0, dpre, Out, 0, out, Out, 1, (int)N, bias, accumulate, bf16, ws, wsf,
                           (hipStream_t)stream);
    // shapes the fused product does not take: the same result from the separate products
    rc = dense_bwd(x1, nullptr, dpre, nullptr, out, bias ? out + (size_t)(In1 + In2) * Out : nullptr, (int)N, In1, Out, accumulate, ws,
                   wsf, (hipStream_t)stream, bf16);
    if (rc == 0 && In2 > 0)
        rc = dense_bwd(x2, nullptr, dpre, nullptr, out + (size_t)In1 * Out, nullptr, (int)N, In2, Out, accumulate, ws, wsf,
                       (hipStream_t)stream, bf16);
    return rc;
}

int fov_dense_bwd_bf16(const float* x, const float* W, const float* dpre, float* dx, float* dW, float* db, int N, int In,
                       int Out, int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (N < 0 || In <= 0 || Out <= 0 || (N > 0 && (!x || !W || !dpre))) {
        set_error("fov_dense_bwd_bf16: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_dense_bwd_workspace_bytes(N, In, Out));
    if (rc) return rc;
    return dense_bwd(x, W, dpre, dx, dW, db, N, In, Out, accumulate, (float*)workspace, workspace_bytes / sizeof(float),
                     (hipStream_t)stream, 1);
}

int fov_mse_dense_grad(const float* y, const float* target, float* dpre, float* loss, int64_t n, int activation,
                       void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (n < 0 || (n > 0 && (!y || !target || !dpre)) || (activation != 0 && activation != 1)) {
        set_error("fov_mse_dense_grad: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * ((size_t)(n + 255) / 256 + 64));
    if (rc) return rc;
    return mse_dense_grad(y, target, dpre, loss, (long)n, activation, (float*)workspace, workspace_bytes / sizeof(float),
                          (hipStream_t)stream);
}

int fov_mse_dense_grad_w(const float* y, const float* target, float* dpre, float* loss, int64_t n, int activation,
                         float weight, int time_major_B, int time_major_T, int O, void* workspace, size_t workspace_bytes,
                         fov_stream_t stream) {
    const bool tm = time_major_T > 0;
    if (n < 0 || (n > 0 && (!y || !target || !dpre)) || (activation != 0 && activation != 1) ||
        (tm && (time_major_B <= 0 || O <= 0 || (int64_t)time_major_B * time_major_T * O != n))) {
        set_error("fov_mse_dense_grad_w: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * ((size_t)(n + 255) / 256 + 64));
    if (rc) return rc;
    return mse_dense_grad_w(y, target, dpre, loss, (long)n, activation, weight, tm ? time_major_B : 1, tm ? time_major_T : 0,
                            tm ? O : 1, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_mse_dense_grad_db(const float* y, const float* target, float* dpre, float* loss, float* db, int64_t n, int O, int activation,
                          float weight, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (n < 0 || O < 1 || n % O || (n > 0 && (!y || !target || !dpre)) || !db || (activation != 0 && activation != 1)) {
        set_error("fov_mse_dense_grad_db: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * (9 * ((size_t)(n + 255) / 256) + 64 + (size_t)256 * O));
    if (rc) return rc;
    return mse_dense_grad_w(y, target, dpre, loss, (long)n, activation, weight, 1, 0, 1, (float*)workspace, workspace_bytes / sizeof(float),
                            (hipStream_t)stream, db, O);
}

int fov_dense_mse_head_supported(int64_t N, int H, int O) { return dense_mse_head_shape_ok((long)N, H, O) ? 1 : 0; }
size_t fov_dense_mse_head_workspace_bytes(int64_t N, int H, int O) { return sizeof(float) * dense_mse_head_scratch_floats((long)N, H, O); }
int fov_dense_mse_head(const float* hs, const float* W, const float* b, const float* target, float* y, float* dX, float* dW, float* db,
                       float* loss, int64_t N, int H, int O, int activation, float weight, void* workspace, size_t workspace_bytes,
                       fov_stream_t stream) {
    if (N < 0 || H <= 0 || O <= 0 || (N > 0 && (!hs || !W || !b || !target || !dW || !db)) || (activation != 0 && activation != 1)) {
        set_error("fov_dense_mse_head: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (N == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, fov_dense_mse_head_workspace_bytes(N, H, O));
    if (rc) return rc;
    return dense_mse_head(hs, W, b, target, y, dX, dW, db, loss, (long)N, H, O, activation, weight, (float*)workspace,
                          workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_scale(float* x, int64_t n, float s, fov_stream_t stream) {
    if (n < 0 || (n > 0 && !x)) { set_error("fov_scale: invalid argument"); return FOV_ERR_INVALID; }
    return scale_inplace(x, (long)n, s, (hipStream_t)stream);
}

int fov_act_bwd(const float* dy, const float* y, const float* base, float* out, int64_t n, int activation,
                fov_stream_t stream) {
    if (n < 0 || (n > 0 && (!dy || !y || !out)) || (activation < 0 || activation > 3)) {
        set_error("fov_act_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return act_bwd(dy, y, base, out, (long)n, activation, (hipStream_t)stream);
}

int fov_act_fwd(const float* x, float* y, int64_t n, int activation, fov_stream_t stream) {
    if (n < 0 || (n > 0 && (!x || !y)) || activation < 0 || activation > 3) {
        set_error("fov_act_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return act_fwd(x, y, (long)n, activation, (hipStream_t)stream);
}

int fov_gauss_nll_grad(const float* mu, const float* var, const float* y, float* loss, float* dmu, float* dvar, int B,
                       int T_y, int fps, float scale, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T_y <= 0 || fps <= 0 || (B > 0 && (!mu || !var || !y || !dmu || !dvar))) {
        set_error("fov_gauss_nll_grad: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (B == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * ((size_t)B + 64));
    if (rc) return rc;
    return gauss_nll_grad(mu, var, y, loss, dmu, dvar, B, T_y, fps, scale, (float*)workspace, workspace_bytes / sizeof(float),
                          (hipStream_t)stream);
}

int fov_categorical_crossentropy_grad(const float* p, const float* target, float* dp, float* loss, int64_t n_pix, int C,
                                      void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (n_pix < 0 || C <= 0 || (n_pix > 0 && (!p || !target || !dp))) {
        set_error("fov_categorical_crossentropy_grad: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (n_pix == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * ((size_t)(n_pix + 255) / 256 + 64));
    if (rc) return rc;
    return cce_grad(p, target, dp, loss, (long)n_pix, C, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_xyz_sum1_grad(const float* p, float* dp, float* reg, int64_t n_pix, int C, void* workspace, size_t workspace_bytes,
                      fov_stream_t stream) {
    if (n_pix < 0 || C < 3 || (n_pix > 0 && (!p || !dp))) {
        set_error("fov_xyz_sum1_grad: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (n_pix == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * ((size_t)(n_pix + 255) / 256 + 64));
    if (rc) return rc;
    return xyz_sum1_grad(p, dp, reg, (long)n_pix, C, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_sample_refeed_fwd(const float* mu, const float* var, const float* noise, float* x, int64_t ldx, int B, int fps,
                          int std_mode, int layout, fov_stream_t stream) {
    if (B < 0 || fps <= 0 || ldx < 3 * (int64_t)fps || (std_mode | layout) & ~1 || (B > 0 && (!mu || !var || !noise || !x))) {
        set_error("fov_sample_refeed_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return sample_refeed_fwd(mu, var, noise, x, (long)ldx, B, fps, std_mode, layout, (hipStream_t)stream);
}

int fov_sample_refeed_bwd(const float* dx, int64_t ldx, const float* var, const float* noise, float* dmu, float* dvar, int B,
                          int fps, int std_mode, int layout, int accumulate, fov_stream_t stream) {
    if (B < 0 || fps <= 0 || ldx < 3 * (int64_t)fps || (std_mode | layout) & ~1 ||
        (B > 0 && (!dx || !var || !noise || !dmu || !dvar))) {
        set_error("fov_sample_refeed_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return sample_refeed_bwd(dx, (long)ldx, var, noise, dmu, dvar, B, fps, std_mode, layout, accumulate, (hipStream_t)stream);
}

int fov_rmsprop_tf_step(float* params, const float* grads, float* ms, int64_t n, float lr, float decay, float eps,
                        float clip_value, fov_stream_t stream) {
    if (n < 0 || (n > 0 && (!params || !grads || !ms))) {
        set_error("fov_rmsprop_tf_step: invalid argument");
        return FOV_ERR_INVALID;
    }
    return rmsprop_tf_step(params, grads, ms, (long)n, lr, decay, eps, clip_value, nullptr, nullptr, (hipStream_t)stream);
}

int fov_rmsprop_tf_step_guarded(float* params, const float* grads, float* ms, int64_t n, float lr, float decay, float eps,
                                float clip_value, const void* guard0, const void* guard1, const void* guard2, int64_t* applied,
                                fov_stream_t stream) {
    if (n < 0 || (n > 0 && (!params || !grads || !ms))) {
        set_error("fov_rmsprop_tf_step_guarded: invalid argument");
        return FOV_ERR_INVALID;
    }
    const unsigned* guards[3] = {(const unsigned*)guard0, (const unsigned*)guard1, (const unsigned*)guard2};
    return rmsprop_tf_step(params, grads, ms, (long)n, lr, decay, eps, clip_value, guards, (long long*)applied, (hipStream_t)stream);
}

int fov_adam_step_guarded(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                          float beta2, float eps, int64_t step, const void* guard0, const void* guard1, const void* guard2,
                          int64_t* applied, fov_stream_t stream) {
    if (n < 0 || step < 1 || (n > 0 && (!params || !grads || !m || !v))) {
        set_error("fov_adam_step: invalid argument");
        return FOV_ERR_INVALID;
    }
    const double lr_t = (double)lr * sqrt(1.0 - pow((double)beta2, (double)step)) / (1.0 - pow((double)beta1, (double)step));
    const unsigned* guards[3] = {(const unsigned*)guard0, (const unsigned*)guard1, (const unsigned*)guard2};
    return adam_step(params, grads, m, v, (long)n, (float)lr_t, beta1, beta2, eps, guards, (long long*)applied, (hipStream_t)stream);
}

int fov_reduce_defer_begin(float* grad_base, size_t grad_floats, void* arena, size_t arena_bytes, fov_stream_t stream) {
    if (grad_base && (!arena || arena_bytes < 256)) { set_error("fov_reduce_defer_begin: invalid arena"); return FOV_ERR_INVALID; }
    return defer_begin(grad_base, grad_floats, (float*)arena, arena_bytes / sizeof(float), (hipStream_t)stream);
}
int fov_reduce_defer_flush(const float* grad_base, fov_stream_t stream) { return defer_flush(grad_base, (hipStream_t)stream); }
int fov_reduce_defer_end(const float* grad_base, fov_stream_t stream) { return defer_end(grad_base, (hipStream_t)stream); }

int fov_guard_flag(const void* guard0, const void* guard1, const void* guard2, float* out, fov_stream_t stream) {
    if (!out) { set_error("fov_guard_flag: invalid argument"); return FOV_ERR_INVALID; }
    const unsigned* guards[3] = {(const unsigned*)guard0, (const unsigned*)guard1, (const unsigned*)guard2};
    return guard_flag(guards, out, (hipStream_t)stream);
}

int fov_adam_step(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, int64_t step, fov_stream_t stream) {
    return fov_adam_step_guarded(params, grads, m, v, n, lr, beta1, beta2, eps, step, nullptr, nullptr, nullptr, nullptr, stream);
}

int fov_rmsprop_step_guarded(float* params, const float* grads, float* accum, int64_t n, float lr, float rho, float eps,
                             const void* guard0, const void* guard1, const void* guard2, int64_t* applied, fov_stream_t stream) {
    if (n < 0 || (n > 0 && (!params || !grads || !accum))) {
        set_error("fov_rmsprop_step: invalid argument");
        return FOV_ERR_INVALID;
    }
    const unsigned* guards[3] = {(const unsigned*)guard0, (const unsigned*)guard1, (const unsigned*)guard2};
    return rmsprop_step(params, grads, accum, (long)n, lr, rho, eps, guards, (long long*)applied, (hipStream_t)stream);
}

int fov_rmsprop_step(float* params, const float* grads, float* accum, int64_t n, float lr, float rho, float eps,
                     fov_stream_t stream) {
    return fov_rmsprop_step_guarded(params, grads, accum, n, lr, rho, eps, nullptr, nullptr, nullptr, nullptr, stream);
}

int fov_dense_fwd(const float* x, const float* W, const float* b, float* y, int N, int In, int Out,
                  int activation, fov_stream_t stream) {
    if (N < 0 || In <= 0 || Out <= 0 || !W || (N > 0 && (!x || !y)) || (activation != 0 && activation != 1)) {
        set_error("fov_dense_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (N == 0) return FOV_OK;
    return launch_dense(x, W, b, nullptr, 0L, y, N, In, Out, activation, (hipStream_t)stream);
}

int fov_dense_add_fwd(const float* x, const float* W, const float* b, const float* add, int64_t add_row_stride,
                      float* y, int N, int In, int Out, int activation, fov_stream_t stream) {
    if (N < 0 || In <= 0 || Out <= 0 || !W || (N > 0 && (!x || !y || !add)) || (activation != 0 && activation != 1)) {
        set_error("fov_dense_add_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (N == 0) return FOV_OK;
    return launch_dense(x, W, b, add, (long)add_row_stride, y, N, In, Out, activation, (hipStream_t)stream);
}

int fov_mix_head_fwd(const float* h, const float* dense_W, const float* dense_b, const float* mix_Wp, const float* add,
                     int64_t add_row_stride, float* p, float* m, int N, int H, int O, fov_stream_t stream) {
    if (N < 0 || H <= 0 || O <= 0 || O > 8 || (H & 3) || H > 2048 || !dense_W || !dense_b || !mix_Wp ||
        (N > 0 && (!h || !add || !p || !m)) || (((uintptr_t)h) & 15)) {
        set_error("fov_mix_head_fwd: invalid argument (O <= 8, H a multiple of 4 and <= 2048, h 16-byte aligned)");
        return FOV_ERR_INVALID;
    }
    if (N == 0) return FOV_OK;
    long blocks = ((long)N + 15) / 16;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(mix_head_fwd_kernel, dim3((unsigned)blocks), dim3(256), sizeof(float) * (8 * H + 64), (hipStream_t)stream, h,
                       dense_W, dense_b, mix_Wp, add, (long)add_row_stride, p, m, N, H, O);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_head_fwd launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int fov_mix_head_bwd(const float* dm_loss, const float* dm_feedback, const float* m, const float* p, const float* mix_Wp,
                     const float* dense_W, float* dpre_m, float* dpre_p, float* dh, int N, int H, int O, fov_stream_t stream) {
    if (N < 0 || H <= 0 || O <= 0 || O > 8 || !mix_Wp || !dense_W || (N > 0 && (!dm_loss || !m || !p || !dpre_m || !dpre_p || !dh))) {
        set_error("fov_mix_head_bwd: invalid argument (O <= 8)");
        return FOV_ERR_INVALID;
    }
    if (N == 0) return FOV_OK;
    const long n = (long)N * H;
    hipLaunchKernelGGL(mix_head_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dm_loss, dm_feedback,
                       m, p, mix_Wp, dense_W, dpre_m, dpre_p, dh, N, H, O);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_head_bwd launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

size_t fov_mix_head_wgrad_workspace_bytes(int B, int T_out, int H, int O, int n_others) {
    if (B <= 0 || T_out <= 0 || H <= 0 || O <= 0 || n_others < 0) return 256;
    return sizeof(float) * mix_head_wgrad_scratch_floats(B, T_out, H, O, n_others);
}

int fov_mix_head_wgrad(const float* h2, const float* dpre_p, const float* others, const float* p, const float* dpre_m, float* out,
                       int B, int T_out, int H, int O, int n_others, int accumulate, void* workspace, size_t workspace_bytes,
                       fov_stream_t stream) {
    if (B < 0 || T_out < 0 || H <= 0 || O <= 0 || O > 8 || n_others < 0 || !out ||
        (B > 0 && T_out > 0 && (!h2 || !dpre_p || !p || !dpre_m || (n_others > 0 && !others)))) {
        set_error("fov_mix_head_wgrad: invalid argument (O <= 8)");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_mix_head_wgrad_workspace_bytes(B, T_out, H, O, n_others));
    if (rc) return rc;
    return mix_head_wgrad(h2, dpre_p, others, p, dpre_m, out, B, T_out, H, O, n_others, accumulate, (float*)workspace,
                          workspace_bytes / sizeof(float), (hipStream_t)stream);
}

size_t fov_mix_decoder_workspace_bytes(int B, int H) {
    if (B <= 0 || H != 256) return kStatusBytes;
    return mix_decoder_workspace_bytes(B);
}

static int mix_decoder_fwd_impl(const float* dec0, const float* h1, const float* c1, const float* h2, const float* c2,
                        const float* oth_proj, int64_t oth_batch_stride, int64_t oth_step_stride, const float* dec1_K,
                        const float* dec1_R, const float* dec1_b, const float* dec2_K, const float* dec2_R,
                        const float* dec2_b, const float* dense_W, const float* dense_b, const float* mix_Wp, float* out,
                        float* h1T, float* c1T, float* h2T, float* c2T, float* P, float* H1, float* C1, float* H2, float* C2,
                        float* res1, float* res2, int B, int T_out, int H, int O, int act, void* workspace,
                        size_t workspace_bytes, fov_stream_t stream, bool bf16) {
    if (B < 0 || T_out < 0 || O <= 0 || !dec1_K || !dec1_R || !dec1_b || !dec2_K || !dec2_R || !dec2_b || !dense_W ||
        !dense_b || !mix_Wp || (B > 0 && T_out > 0 && (!dec0 || !h1 || !c1 || !h2 || !c2 || !oth_proj || !out)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_mix_decoder_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    const int ntrain = (P != nullptr) + (H1 != nullptr) + (C1 != nullptr) + (H2 != nullptr) + (C2 != nullptr) + (res1 != nullptr) +
                       (res2 != nullptr);
    if (ntrain != 0 && ntrain != 7) { set_error("fov_mix_decoder_fwd: give all training buffers or none"); return FOV_ERR_INVALID; }
    if (H != 256 || O > 8) { set_error("fov_mix_decoder_fwd: the fused decoder supports H = 256, O <= 8"); return FOV_ERR_UNSUPPORTED; }
    if (B == 0 || T_out == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, fov_mix_decoder_workspace_bytes(B, H));
    if (rc) return rc;
    MixDecParams p = {};
    p.K1 = dec1_K; p.R1 = dec1_R; p.b1 = dec1_b; p.R2 = dec2_R; p.b2 = dec2_b; p.Wd = dense_W; p.bd = dense_b; p.Wp = mix_Wp;
    p.oth_proj = oth_proj; p.oth_sb = (long)oth_batch_stride; p.oth_st = (long)oth_step_stride;
    p.dec0 = dec0; p.h1_0 = h1; p.c1_0 = c1; p.h2_0 = h2; p.c2_0 = c2;
    p.out = out; p.P = P; p.H1 = H1; p.C1 = C1; p.H2 = H2; p.C2 = C2; p.res1 = res1; p.res2 = res2;
    p.h1T = h1T; p.c1T = c1T; p.h2T = h2T; p.c2T = c2T;
    p.B = B; p.T_out = T_out; p.O = O;
    return bf16 ? mix_decoder_bf16_launch(p, dec2_K, act, ntrain == 7, workspace, (hipStream_t)stream)
                : mix_decoder_launch(p, dec2_K, act, ntrain == 7, workspace, (hipStream_t)stream);
}

int fov_mix_decoder_fwd(const float* dec0, const float* h1, const float* c1, const float* h2, const float* c2,
                        const float* oth_proj, int64_t oth_batch_stride, int64_t oth_step_stride, const float* dec1_K,
                        const float* dec1_R, const float* dec1_b, const float* dec2_K, const float* dec2_R,
                        const float* dec2_b, const float* dense_W, const float* dense_b, const float* mix_Wp, float* out,
                        float* h1T, float* c1T, float* h2T, float* c2T, float* P, float* H1, float* C1, float* H2, float* C2,
                        float* res1, float* res2, int B, int T_out, int H, int O, int act, void* workspace,
                        size_t workspace_bytes, fov_stream_t stream) {
    return mix_decoder_fwd_impl(dec0, h1, c1, h2, c2, oth_proj, oth_batch_stride, oth_step_stride, dec1_K, dec1_R, dec1_b, dec2_K,
                                dec2_R, dec2_b, dense_W, dense_b, mix_Wp, out, h1T, c1T, h2T, c2T, P, H1, C1, H2, C2, res1, res2,
                                B, T_out, H, O, act, workspace, workspace_bytes, stream, false);
}

int fov_mix_decoder_fwd_bf16(const float* dec0, const float* h1, const float* c1, const float* h2, const float* c2,
                             const float* oth_proj, int64_t oth_batch_stride, int64_t oth_step_stride, const float* dec1_K,
                             const float* dec1_R, const float* dec1_b, const float* dec2_K, const float* dec2_R,
                             const float* dec2_b, const float* dense_W, const float* dense_b, const float* mix_Wp, float* out,
                             float* h1T, float* c1T, float* h2T, float* c2T, float* P, float* H1, float* C1, float* H2,
                             float* C2, float* res1, float* res2, int B, int T_out, int H, int O, int act, void* workspace,
                             size_t workspace_bytes, fov_stream_t stream) {
    return mix_decoder_fwd_impl(dec0, h1, c1, h2, c2, oth_proj, oth_batch_stride, oth_step_stride, dec1_K, dec1_R, dec1_b, dec2_K,
                                dec2_R, dec2_b, dense_W, dense_b, mix_Wp, out, h1T, c1T, h2T, c2T, P, H1, C1, H2, C2, res1, res2,
                                B, T_out, H, O, act, workspace, workspace_bytes, stream, true);
}

size_t fov_mix_decoder_bwd_workspace_bytes(int B, int H) {
    if (B <= 0 || H != 256) return kStatusBytes;
    return mix_decoder_bwd_workspace_bytes(B);
}

static int mix_decoder_bwd_impl(const float* M, const float* P, const float* dloss, const float* res1, const float* res2,
                        const float* C1, const float* C2, const float* dec1_K, const float* dec1_R, const float* dec2_K,
                        const float* dec2_R, const float* dense_W, const float* mix_Wp, float* DZ1, float* DZ2,
                        float* dpre_m, float* dpre_p, float* dh1_0, float* dc1_0, float* dh2_0, float* dc2_0, int B,
                        int T_out, int H, int O, int act, void* workspace, size_t workspace_bytes, fov_stream_t stream, bool bf16) {
    if (B < 0 || T_out < 0 || O <= 0 || !dec1_K || !dec1_R || !dec2_K || !dec2_R || !dense_W || !mix_Wp ||
        (B > 0 && T_out > 0 && (!M || !P || !dloss || !res1 || !res2 || !C1 || !C2 || !DZ1 || !DZ2 || !dpre_m || !dpre_p ||
                                !dh1_0 || !dc1_0 || !dh2_0 || !dc2_0)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_mix_decoder_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (H != 256 || O > 8) { set_error("fov_mix_decoder_bwd: the fused decoder supports H = 256, O <= 8"); return FOV_ERR_UNSUPPORTED; }
    if (B == 0 || T_out == 0) return FOV_OK;
    int rc = check_ws(workspace, workspace_bytes, fov_mix_decoder_bwd_workspace_bytes(B, H));
    if (rc) return rc;
    MixDecBwdParams p = {};
    p.R1 = dec1_R; p.K1 = dec1_K; p.R2 = dec2_R; p.Wd = dense_W; p.Wp = mix_Wp;
    p.M = M; p.P = P; p.dloss = dloss; p.res1 = res1; p.res2 = res2; p.C1 = C1; p.C2 = C2;
    p.DZ1 = DZ1; p.DZ2 = DZ2; p.dpre_m = dpre_m; p.dpre_p = dpre_p;
    p.dh1_0 = dh1_0; p.dc1_0 = dc1_0; p.dh2_0 = dh2_0; p.dc2_0 = dc2_0;
    p.B = B; p.T_out = T_out; p.O = O;
    return bf16 ? mix_decoder_bwd_bf16_launch(p, dec2_K, act, workspace, (hipStream_t)stream)
                : mix_decoder_bwd_launch(p, dec2_K, act, workspace, (hipStream_t)stream);
}

int fov_mix_decoder_prepack(const float* dec2_K, void* workspace_fwd, size_t fwd_bytes, void* workspace_bwd, size_t bwd_bytes, int H,
                            fov_stream_t stream) {
    if (!dec2_K || H != 256) { set_error("fov_mix_decoder_prepack: invalid argument (H = 256)"); return FOV_ERR_INVALID; }
    int rc = FOV_OK;
    if (workspace_fwd) {
        rc = check_ws(workspace_fwd, fwd_bytes, mix_decoder_workspace_bytes(1));
        if (!rc) rc = mix_decoder_prepack(dec2_K, workspace_fwd, (hipStream_t)stream);
    }
    if (!rc && workspace_bwd) {
        rc = check_ws(workspace_bwd, bwd_bytes, mix_decoder_bwd_workspace_bytes(1));
        if (!rc) rc = mix_decoder_bwd_prepack(dec2_K, workspace_bwd, (hipStream_t)stream);
    }
    return rc;
}

int fov_mix_decoder_bwd(const float* M, const float* P, const float* dloss, const float* res1, const float* res2,
                        const float* C1, const float* C2, const float* dec1_K, const float* dec1_R, const float* dec2_K,
                        const float* dec2_R, const float* dense_W, const float* mix_Wp, float* DZ1, float* DZ2,
                        float* dpre_m, float* dpre_p, float* dh1_0, float* dc1_0, float* dh2_0, float* dc2_0, int B,
                        int T_out, int H, int O, int act, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    return mix_decoder_bwd_impl(M, P, dloss, res1, res2, C1, C2, dec1_K, dec1_R, dec2_K, dec2_R, dense_W, mix_Wp, DZ1, DZ2, dpre_m,
                                dpre_p, dh1_0, dc1_0, dh2_0, dc2_0, B, T_out, H, O, act, workspace, workspace_bytes, stream, false);
}

int fov_mix_decoder_bwd_bf16(const float* M, const float* P, const float* dloss, const float* res1, const float* res2,
                             const float* C1, const float* C2, const float* dec1_K, const float* dec1_R, const float* dec2_K,
                             const float* dec2_R, const float* dense_W, const float* mix_Wp, float* DZ1, float* DZ2,
                             float* dpre_m, float* dpre_p, float* dh1_0, float* dc1_0, float* dh2_0, float* dc2_0, int B,
                             int T_out, int H, int O, int act, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    return mix_decoder_bwd_impl(M, P, dloss, res1, res2, C1, C2, dec1_K, dec1_R, dec2_K, dec2_R, dense_W, mix_Wp, DZ1, DZ2, dpre_m,
                                dpre_p, dh1_0, dc1_0, dh2_0, dc2_0, B, T_out, H, O, act, workspace, workspace_bytes, stream, true);
}

size_t fov_matmul_workspace_bytes(int M, int K, int N) {
    (void)K;
    if (M <= 0 || N <= 0) return 256;
    size_t part = (size_t)64 * M * N;
    if (part > ((size_t)64 << 20)) part = (size_t)64 << 20;   // split-K partials are optional: cap at 256 MB
    return sizeof(float) * (part + 64);
}

int fov_matmul(const float* a, const float* b, float* c, int M, int K, int N, void* workspace, size_t workspace_bytes,
               fov_stream_t stream) {
    if (M < 0 || K < 0 || N < 0 || (M > 0 && N > 0 && (!c || (K > 0 && (!a || !b))))) {
        set_error("fov_matmul: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (workspace && (((uintptr_t)workspace) & 15)) { set_error("workspace must be 16-byte aligned"); return FOV_ERR_WORKSPACE; }
    return matmul_f32(a, b, c, M, K, N, (float*)workspace, workspace ? workspace_bytes / sizeof(float) : 0, (hipStream_t)stream);
}

size_t fov_seq2seq_decode_workspace_bytes(int B, int T_in, int T_out, int F_enc, int F_dec, int H, int impl) {
    (void)T_in; (void)T_out;
    if (B <= 0) return kStatusBytes;
    return want_cluster(impl, F_enc, H, F_dec, true) && cluster_shape_ok(F_enc, H) ? cluster_workspace_bytes(B, H) : kStatusBytes;
}

int fov_seq2seq_decode_fwd(const float* enc_in, const float* dec_in0, const float* enc_K, const float* enc_R,
                           const float* enc_b, const float* dec_K, const float* dec_R, const float* dec_b,
                           const float* dense_W, const float* dense_b, float* out, float* hT, float* cT, int B,
                           int T_in, int T_out, int F_enc, int F_dec, int H, int act, int impl, void* workspace,
                           size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T_in < 0 || T_out < 0 || F_enc <= 0 || F_dec <= 0 || H <= 0 || !enc_K || !enc_R || !enc_b ||
        !dec_K || !dec_R || !dec_b || !dense_W || !dense_b ||
        (B > 0 && ((T_in > 0 && !enc_in) || !dec_in0 || (T_out > 0 && !out))) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_seq2seq_decode_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (F_dec > 64) { set_error("fov_seq2seq_decode_fwd: F_dec > 64 unsupported"); return FOV_ERR_UNSUPPORTED; }
    int rc = check_ws(workspace, workspace_bytes, fov_seq2seq_decode_workspace_bytes(B, T_in, T_out, F_enc, F_dec, H, impl));
    if (rc) return rc;
    LstmParams p = {};
    p.x = enc_in; p.K = enc_K; p.R = enc_R; p.b = enc_b; p.hT = hT; p.cT = cT;
    p.dec_in0 = dec_in0; p.dK = dec_K; p.dR = dec_R; p.db = dec_b; p.dW = dense_W; p.dbias = dense_b; p.out = out;
    p.B = B; p.T = T_in; p.F = F_enc; p.H = H; p.T_out = T_out; p.F_dec = F_dec; p.act = act;
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    hipStream_t s = (hipStream_t)stream;
    // small batches (at most eight tiles): the tile spread over H / 16 workgroups instead of H / 64 (lstm_wide16.hip)
    if (impl == FOV_IMPL_AUTO && want_cluster(impl, F_enc, H, F_dec, true) && wide16_s2s_shape(B, F_enc, F_dec, H) && T_out > 0)
        return launch_wide16_s2s(p, s);
    // batches of 33 .. 64 tiles at H = 256 (the metric's 1024 sequences): two tiles per group of eight workgroups, each tile's
    // exchange under the other tile's MFMAs (lstm_pair.hip)
    if (impl == FOV_IMPL_AUTO && want_cluster(impl, F_enc, H, F_dec, true) && pair_s2s_shape(B, T_in, T_out, F_enc, F_dec, H))
        return launch_pair_s2s(p, s);
    if (want_cluster(impl, F_enc, H, F_dec, true)) return launch_cluster(p, true, s);
    return launch_generic(p, true, s);
}

int fov_seq2seq_decoder_fwd(const float* dec_in0, const float* h0, const float* c0, const float* dec_K,
                            const float* dec_R, const float* dec_b, const float* dense_W, const float* dense_b, float* out,
                            float* hT, float* cT, int B, int T_out, int F_dec, int H, int act, int impl, void* workspace,
                            size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || T_out < 0 || F_dec <= 0 || H <= 0 || !dec_K || !dec_R || !dec_b || !dense_W || !dense_b ||
        (B > 0 && (!dec_in0 || (T_out > 0 && !out))) || (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_seq2seq_decoder_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (F_dec > 64) { set_error("fov_seq2seq_decoder_fwd: F_dec > 64 unsupported"); return FOV_ERR_UNSUPPORTED; }
    int rc = check_ws(workspace, workspace_bytes, fov_seq2seq_decode_workspace_bytes(B, 0, T_out, 1, F_dec, H, impl));
    if (rc) return rc;
    LstmParams p = {};
    p.h0 = h0; p.c0 = c0; p.hT = hT; p.cT = cT;
    p.dec_in0 = dec_in0; p.dK = dec_K; p.dR = dec_R; p.db = dec_b; p.dW = dense_W; p.dbias = dense_b; p.out = out;
    p.B = B; p.T = 0; p.F = 1; p.H = H; p.T_out = T_out; p.F_dec = F_dec; p.act = act;
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    if (impl == FOV_IMPL_AUTO && want_cluster(impl, 1, H, F_dec, true) && wide16_s2s_shape(B, 1, F_dec, H) && T_out > 0) {
        p.K = dec_K; p.R = dec_R; p.b = dec_b;   // (the empty encoder phase's weight slots: any valid pointers)
        return launch_wide16_s2s(p, (hipStream_t)stream);
    }
    if (want_cluster(impl, 1, H, F_dec, true)) return launch_cluster_decoder(p, (hipStream_t)stream);
    p.K = dec_K; p.R = dec_R; p.b = dec_b;   // unused by the generic kernel's empty encoder phase (T = 0)
    return launch_generic(p, true, (hipStream_t)stream);
}

static size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

size_t fov_seq2seq_tf_workspace_bytes(int B, int T_in, int T_out, int F_enc, int F_dec, int H, int impl) {
    if (B <= 0) return kStatusBytes;
    size_t a = fov_lstm_seq_workspace_bytes(B, T_in, F_enc, H, impl);
    size_t c = fov_lstm_seq_workspace_bytes(B, T_out, F_dec, H, impl);
    size_t lstm = align256(a > c ? a : c);
    // + encoder final state (h,c) + decoder hidden sequence
    return lstm + align256((size_t)2 * B * H * sizeof(float)) + align256((size_t)B * T_out * H * sizeof(float));
}

int fov_seq2seq_tf_fwd(const float* enc_in, const float* dec_in, const float* enc_K, const float* enc_R,
                       const float* enc_b, const float* dec_K, const float* dec_R, const float* dec_b,
                       const float* dense_W, const float* dense_b, float* out, int B, int T_in, int T_out,
                       int F_enc, int F_dec, int H, int act, int impl, void* workspace, size_t workspace_bytes,
                       fov_stream_t stream) {
    if (B < 0 || T_in < 0 || T_out < 0 || F_enc <= 0 || F_dec <= 0 || H <= 0) {
        set_error("fov_seq2seq_tf_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, fov_seq2seq_tf_workspace_bytes(B, T_in, T_out, F_enc, F_dec, H, impl));
    if (rc) return rc;
    if (B == 0) return FOV_OK;
    size_t a = fov_lstm_seq_workspace_bytes(B, T_in, F_enc, H, impl);
    size_t c = fov_lstm_seq_workspace_bytes(B, T_out, F_dec, H, impl);
    const size_t lstm = align256(a > c ? a : c);
    char* base = (char*)workspace;
    float* hT = (float*)(base + lstm);
    float* cT = hT + (size_t)B * H;
    float* hs = (float*)(base + lstm + align256((size_t)2 * B * H * sizeof(float)));
    rc = fov_lstm_seq_fwd(enc_in, enc_K, enc_R, enc_b, nullptr, nullptr, nullptr, hT, cT, B, T_in, F_enc, H, act,
                          impl, workspace, lstm, stream);
    if (rc) return rc;
    rc = fov_lstm_seq_fwd(dec_in, dec_K, dec_R, dec_b, hT, cT, hs, nullptr, nullptr, B, T_out, F_dec, H, act, impl,
                          workspace, lstm, stream);
    if (rc) return rc;
    return fov_dense_fwd(hs, dense_W, dense_b, out, B * T_out, H, F_dec, 1, stream);
}

int fov_meanvar_xyz(const float* y, float* out, int64_t rows, int fps, fov_stream_t stream) {
    if (rows < 0 || fps <= 0 || (rows > 0 && (!y || !out))) {
        set_error("fov_meanvar_xyz: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (rows == 0) return FOV_OK;
    const long n = rows * 3;
    hipLaunchKernelGGL(meanvar_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, y, out,
                       (long)rows, fps);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("meanvar launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int fov_workspace_init(void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (!workspace || workspace_bytes < kStatusBytes) {
        set_error("fov_workspace_init: invalid workspace");
        return FOV_ERR_INVALID;
    }
    hipError_t e = hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("fov_workspace_init: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    xch_forget(workspace);
    prepack_forget(workspace);
    xch_note_force_safe(workspace, 0);
    return FOV_OK;
}

int fov_workspace_force_safe(void* workspace, size_t workspace_bytes, int on, fov_stream_t stream) {
    if (!workspace || workspace_bytes < kStatusBytes) {
        set_error("fov_workspace_force_safe: invalid workspace");
        return FOV_ERR_INVALID;
    }
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)((unsigned*)workspace + ST_FORCE_SAFE), on ? 1 : 0, 1, (hipStream_t)stream);
    if (e != hipSuccess) { set_error("fov_workspace_force_safe: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    xch_note_force_safe(workspace, on);
    return FOV_OK;
}

int fov_check_status(void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (!workspace || workspace_bytes < kStatusBytes) {
        set_error("fov_check_status: invalid workspace");
        return FOV_ERR_INVALID;
    }
    unsigned st[64] = {0};
    hipError_t e = hipMemcpyAsync(st, workspace, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { set_error("fov_check_status: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    // The timeout word is sticky (no launch clears it, later launches skip their bodies): reading it here is what
    // clears it.  After a give-up, or long before the 32-bit epoch tags could wrap, the whole workspace is re-zeroed;
    // the stream is idle at this point.
    if (st[ST_TIMEOUT] != 0 || st[ST_EPOCH] > 0x7fff0000u) {
        e = hipMemsetAsync(workspace, 0, workspace_bytes, (hipStream_t)stream);
        if (e == hipSuccess && st[ST_FORCE_SAFE] != 0)   // the A/B switch is the caller's setting, not launch state: it survives
            e = hipMemsetD32Async((hipDeviceptr_t)((unsigned*)workspace + ST_FORCE_SAFE), 1, 1, (hipStream_t)stream);
        if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
        if (e != hipSuccess) { set_error("fov_check_status: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
        xch_forget(workspace);
        prepack_forget(workspace);
    }
    if (st[ST_TIMEOUT] != 0) {
        set_error("a bounded in-kernel wait gave up (exchange launches on this workspace so far: %u); results since the last check are invalid",
                  st[ST_LAUNCHES]);
        return FOV_ERR_TIMEOUT;
    }
    return FOV_OK;
}

int fov_conv2d_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, const float* w, const float* b,
                   const float* add, float* y, int B, int H, int W, int C, int N, int kh, int kw, int activation,
                   fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || kh <= 0 || kw <= 0 || (kh & 1) == 0 || (kw & 1) == 0 ||
        x_pixel_stride < C || x_batch_stride < (int64_t)H * W * x_pixel_stride || !w || (B > 0 && (!x || !y)) || (activation != 0 && activation != 2)) {
        set_error("fov_conv2d_fwd: invalid argument (odd kernel sizes only, activation 0 or 2)");
        return FOV_ERR_INVALID;
    }
    return conv2d_fwd(x, (long)x_pixel_stride, (long)x_batch_stride, w, b, add, y, B, H, W, C, N, kh, kw, activation, (hipStream_t)stream);
}

int fov_conv2d_dilated_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, const float* w, const float* b,
                           const float* add, float* y, int B, int H, int W, int C, int N, int kh, int kw, int dilation, int activation,
                           fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || kh <= 0 || kw <= 0 || (kh & 1) == 0 || (kw & 1) == 0 || dilation < 1 ||
        x_pixel_stride < C || x_batch_stride < (int64_t)H * W * x_pixel_stride || !w || (B > 0 && (!x || !y)) || (activation != 0 && activation != 2)) {
        set_error("fov_conv2d_dilated_fwd: invalid argument (odd kernel sizes only, dilation >= 1, activation 0 or 2)");
        return FOV_ERR_INVALID;
    }
    return conv2d_fwd(x, (long)x_pixel_stride, (long)x_batch_stride, w, b, add, y, B, H, W, C, N, kh, kw, activation, (hipStream_t)stream, dilation);
}

int fov_conv2d_fwd2(const float* x1, int64_t x1_pixel_stride, int64_t x1_batch_stride, int C1, const float* x2,
                    int64_t x2_pixel_stride, int64_t x2_batch_stride, int C2, const float* w, const float* b, const float* add,
                    float* y, int B, int H, int W, int N, int kh, int kw, int activation, fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C1 <= 0 || C2 <= 0 || N <= 0 || kh <= 0 || kw <= 0 || (kh & 1) == 0 || (kw & 1) == 0 ||
        x1_pixel_stride < C1 || x1_batch_stride < (int64_t)H * W * x1_pixel_stride || x2_pixel_stride < C2 ||
        x2_batch_stride < (int64_t)H * W * x2_pixel_stride || !w || (B > 0 && (!x1 || !x2 || !y)) ||
        (activation != 0 && activation != 2)) {
        set_error("fov_conv2d_fwd2: invalid argument (odd kernel sizes only, activation 0 or 2)");
        return FOV_ERR_INVALID;
    }
    return conv2d_fwd2(x1, (long)x1_pixel_stride, (long)x1_batch_stride, C1, x2, (long)x2_pixel_stride, (long)x2_batch_stride, C2,
                       w, b, add, y, B, H, W, N, kh, kw, activation, (hipStream_t)stream);
}

int fov_convlstm_cell_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, int C, const float* h_prev,
                          int64_t h_prev_pixel_stride, int64_t h_prev_batch_stride, const float* w, const float* b,
                          const float* c_prev, float* c_new, float* h, int64_t h_pixel_stride, float* gates, int B, int H, int W,
                          int F, int kh, int kw, int recurrent_activation, fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1) ||
        (B > 0 && (!x || !w || !c_new || !h)) || x_pixel_stride < C || x_batch_stride < (int64_t)H * W * x_pixel_stride ||
        h_pixel_stride < F ||
        (h_prev && (h_prev_pixel_stride < F || h_prev_batch_stride < (int64_t)H * W * h_prev_pixel_stride || h_prev == h)) ||
        (recurrent_activation != FOV_ACT_SIGMOID && recurrent_activation != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_convlstm_cell_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return convlstm_cell_fwd(x, (long)x_pixel_stride, (long)x_batch_stride, C, h_prev, (long)h_prev_pixel_stride,
                             (long)h_prev_batch_stride, w, b, c_prev, c_new, h, (long)h_pixel_stride, gates, B, H, W, F, kh, kw,
                             recurrent_activation, (hipStream_t)stream);
}

int fov_convlstm_cell_dilated_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, int C, const float* h_prev,
                                  int64_t h_prev_pixel_stride, int64_t h_prev_batch_stride, const float* w, const float* b,
                                  const float* c_prev, float* c_new, float* h, int64_t h_pixel_stride, float* gates, int B, int H, int W,
                                  int F, int kh, int kw, int dilation, int recurrent_activation, fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || F <= 0 || kh <= 0 || kw <= 0 || !(kh & 1) || !(kw & 1) || dilation < 1 ||
        (B > 0 && (!x || !w || !c_new || !h)) || x_pixel_stride < C || x_batch_stride < (int64_t)H * W * x_pixel_stride ||
        h_pixel_stride < F ||
        (h_prev && (h_prev_pixel_stride < F || h_prev_batch_stride < (int64_t)H * W * h_prev_pixel_stride || h_prev == h)) ||
        (recurrent_activation != FOV_ACT_SIGMOID && recurrent_activation != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_convlstm_cell_dilated_fwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return convlstm_cell_fwd(x, (long)x_pixel_stride, (long)x_batch_stride, C, h_prev, (long)h_prev_pixel_stride,
                             (long)h_prev_batch_stride, w, b, c_prev, c_new, h, (long)h_pixel_stride, gates, B, H, W, F, kh, kw,
                             recurrent_activation, (hipStream_t)stream, dilation);
}

int fov_convlstm_gates(const float* z, float* c, float* h, int64_t h_pixel_stride, int64_t rows, int F, int act,
                       fov_stream_t stream) {
    if (rows < 0 || F <= 0 || h_pixel_stride < F || (rows > 0 && (!z || !c || !h)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_convlstm_gates: invalid argument");
        return FOV_ERR_INVALID;
    }
    return convlstm_gates(z, c, h, (long)h_pixel_stride, (long)rows, F, act, (hipStream_t)stream);
}

int fov_softmax_lastdim(const float* x, float* y, int64_t rows, int n, fov_stream_t stream) {
    if (rows < 0 || n <= 0 || (rows > 0 && (!x || !y))) {
        set_error("fov_softmax_lastdim: invalid argument");
        return FOV_ERR_INVALID;
    }
    return softmax_lastdim(x, y, (long)rows, n, (hipStream_t)stream);
}

int fov_convlstm_gates_train(const float* z, const float* c_prev, float* c_new, float* h, int64_t h_pixel_stride,
                             float* gates, int64_t rows, int F, int act, fov_stream_t stream) {
    if (rows < 0 || F <= 0 || h_pixel_stride < F || (rows > 0 && (!z || !c_new || !h || !gates)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_convlstm_gates_train: invalid argument");
        return FOV_ERR_INVALID;
    }
    return convlstm_gates_train(z, c_prev, c_new, h, (long)h_pixel_stride, gates, (long)rows, F, act, (hipStream_t)stream);
}

int fov_convlstm_gates_bwd(const float* dh, int64_t dh_pixel_stride, float* dc, const float* gates, const float* c_prev,
                           const float* c_new, float* dz, int64_t rows, int F, int act, fov_stream_t stream) {
    if (rows < 0 || F <= 0 || dh_pixel_stride < F || (rows > 0 && (!dh || !dc || !gates || !c_new || !dz)) ||
        (act != FOV_ACT_SIGMOID && act != FOV_ACT_HARD_SIGMOID)) {
        set_error("fov_convlstm_gates_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return convlstm_gates_bwd(dh, (long)dh_pixel_stride, dc, gates, c_prev, c_new, dz, (long)rows, F, act, (hipStream_t)stream);
}

size_t fov_conv2d_wgrad_workspace_bytes(int C, int N, int kh, int kw) {
    if (C <= 0 || N <= 0 || kh <= 0 || kw <= 0) return 256;
    return sizeof(float) * conv2d_wgrad_workspace_floats(C, N, kh, kw);
}

int fov_conv2d_wgrad(const float* x, int64_t x_pixel_stride, const float* dy, float* dw, int B, int H, int W, int C, int N,
                     int kh, int kw, int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || kh <= 0 || kw <= 0 || (kh & 1) == 0 || (kw & 1) == 0 ||
        x_pixel_stride < C || !dw || (B > 0 && (!x || !dy))) {
        set_error("fov_conv2d_wgrad: invalid argument (odd kernel sizes only)");
        return FOV_ERR_INVALID;
    }
    if (workspace && (((uintptr_t)workspace) & 15)) { set_error("workspace must be 16-byte aligned"); return FOV_ERR_WORKSPACE; }
    return conv2d_wgrad(x, (long)x_pixel_stride, dy, dw, B, H, W, C, N, kh, kw, accumulate, (float*)workspace,
                        workspace ? workspace_bytes / sizeof(float) : 0, (hipStream_t)stream);
}

int fov_conv2d_dilated_wgrad(const float* x, int64_t x_pixel_stride, const float* dy, float* dw, int B, int H, int W, int C, int N,
                             int kh, int kw, int dilation, int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B < 0 || H <= 0 || W <= 0 || C <= 0 || N <= 0 || kh <= 0 || kw <= 0 || (kh & 1) == 0 || (kw & 1) == 0 || dilation < 1 ||
        x_pixel_stride < C || !dw || (B > 0 && (!x || !dy))) {
        set_error("fov_conv2d_dilated_wgrad: invalid argument (odd kernel sizes only, dilation >= 1)");
        return FOV_ERR_INVALID;
    }
    if (workspace && (((uintptr_t)workspace) & 15)) { set_error("workspace must be 16-byte aligned"); return FOV_ERR_WORKSPACE; }
    return conv2d_wgrad(x, (long)x_pixel_stride, dy, dw, B, H, W, C, N, kh, kw, accumulate, (float*)workspace,
                        workspace ? workspace_bytes / sizeof(float) : 0, (hipStream_t)stream, dilation);
}

int fov_conv2d_weight_transpose(const float* w, float* wt, int kh, int kw, int C, int N, fov_stream_t stream) {
    if (kh <= 0 || kw <= 0 || C <= 0 || N <= 0 || !w || !wt) {
        set_error("fov_conv2d_weight_transpose: invalid argument");
        return FOV_ERR_INVALID;
    }
    return conv_weight_transpose(w, wt, kh, kw, C, N, (hipStream_t)stream);
}

int fov_softmax_lastdim_bwd(const float* dp, const float* p, float* dy, int64_t rows, int n, fov_stream_t stream) {
    if (rows < 0 || n <= 0 || (rows > 0 && (!dp || !p || !dy))) {
        set_error("fov_softmax_lastdim_bwd: invalid argument");
        return FOV_ERR_INVALID;
    }
    return softmax_lastdim_bwd(dp, p, dy, (long)rows, n, (hipStream_t)stream);
}

int fov_colsum(const float* x, float* out, int64_t rows, int cols, int accumulate, void* workspace, size_t workspace_bytes,
               fov_stream_t stream) {
    if (rows < 0 || cols <= 0 || !out || (rows > 0 && !x)) {
        set_error("fov_colsum: invalid argument");
        return FOV_ERR_INVALID;
    }
    int rc = check_ws(workspace, workspace_bytes, sizeof(float) * ((size_t)256 * cols + 64));
    if (rc) return rc;
    if (rows == 0) {
        if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * cols, (hipStream_t)stream);
        return FOV_OK;
    }
    return colsum(x, out, (long)rows, cols, accumulate, (float*)workspace, workspace_bytes / sizeof(float), (hipStream_t)stream);
}

int fov_fov_hit_rate(const float* pred_xyz, int64_t pred_row_stride, const float* gt_xyz, int64_t gt_row_stride,
                     float* out, int64_t rows, float span_deg, float gt_span_deg, fov_stream_t stream) {
    if (rows < 0 || pred_row_stride < 3 || gt_row_stride < 3 || span_deg <= 0.f || gt_span_deg <= 0.f ||
        (rows > 0 && (!pred_xyz || !gt_xyz || !out))) {
        set_error("fov_fov_hit_rate: invalid argument");
        return FOV_ERR_INVALID;
    }
    if (rows == 0) return FOV_OK;
    const float d2r = 3.14159265358979323846f / 180.f;
    hipLaunchKernelGGL(hit_rate_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pred_xyz,
                       (long)pred_row_stride, gt_xyz, (long)gt_row_stride, out, (long)rows, span_deg * d2r, gt_span_deg * d2r);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("hit_rate launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int64_t fov_window_count(int S, int T, int stride) {
    if (S < 2 * T || T <= 0 || stride <= 0 || stride > T) return 0;
    const int shift = T / stride;
    const int nrows = (S - T) / stride + 1;
    return nrows - shift > 0 ? nrows - shift : 0;
}

int fov_window_stacks(const float* x, float* enc, float* fut, float* fut_in, int U, int S, int feat, int T, int stride,
                      int collapse_user, fov_stream_t stream) {
    if (U < 0 || S < 0 || feat <= 0 || T <= 0 || stride <= 0 || stride > T || (U > 0 && S >= 2 * T && (!x || !enc || !fut || !fut_in))) {
        set_error("fov_window_stacks: invalid argument");
        return FOV_ERR_INVALID;
    }
    const int W = (int)fov_window_count(S, T, stride);
    if (U == 0 || W == 0) return FOV_OK;
    const long total = (long)W * U * T * feat;
    hipLaunchKernelGGL(window_stacks_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, enc,
                       fut, fut_in, U, S, feat, T, stride, T / stride, W, collapse_user ? 1 : 0);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("window_stacks launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int fov_exchange_mode(const void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (!workspace || workspace_bytes < kStatusBytes) {
        set_error("fov_exchange_mode: invalid workspace");
        return FOV_ERR_INVALID;
    }
    unsigned st[64] = {0};
    hipError_t e = hipMemcpyAsync(st, workspace, sizeof(st), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) { set_error("fov_exchange_mode: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    if (st[ST_LAUNCHES] == 0) return 1;
    return st[ST_SAFE0 + ((st[ST_LAUNCHES] - 1u) & 1u)] == 0 ? 1 : 2;   // of the last exchange launch
}

}  // extern "C"
