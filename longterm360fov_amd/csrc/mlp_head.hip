// The heads mycode/lstm.py puts on the top LSTM layer's final state when cfg.predict_mean_var is False - the branch the
// committed mycode/config.py:69,71 selects - at the script's batch (32 rows):
//   _GMM_3dgassian (lstm.py:377-400): 400 -> 64 relu -> 128 relu -> 256 relu -> 200 linear, split into 20 softmax weights,
//       60 means, 60 exp() sigmas, 60 tanh() correlations, scored by costfunc.mixture_3d_gaussian_loss (cost.py:486-549);
//   pred_cnn_model_fn (lstm.py:147-174): three conv1d (k = 5, 'same') on ONE time step = the centre taps as
//       H -> 128 relu -> 256 relu -> 90 tanh.
// Both are a chain of up to four small Dense layers on a handful of rows: every product is microseconds of arithmetic and the
// step is bound by launch rate and by the latency of each layer's weight read, so the chain is ONE launch forward (a
// workgroup carries four rows through all layers, activations in LDS), ONE launch for the backward chain (same rows,
// weights transposed through LDS so that both the global read and the LDS read are conflict-free) and ONE launch for every
// weight / bias gradient.  The mixture loss and its gradient are a third kernel: one workgroup per row, the 3x3
// covariance work (eigenvalue repair of cost.py:335-348, inverse, determinant) in fp64 - 20 matrices per row.
// Not bandwidth- or matrix-core-shaped work: at B = 32 the whole head is ~0.5 MFLOP per row.
#include "fov_common.h"

namespace fov {

// Diagnostic build only (-DFOV_STAMPS, tools/microbench/head_stamps.py): s_memtime stamps of wave 0 of workgroup 0.
#ifdef FOV_STAMPS
__device__ unsigned long long g_mh_stamps[3][48];
#define MH_STAMP(kern, slot)                                                                   \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (blockIdx.x == 0 && threadIdx.x == 0) {                                             \
            unsigned long long t_;                                                             \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
            g_mh_stamps[kern][slot] = t_;                                                      \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define MH_STAMP(kern, slot) do { } while (0)
#endif

namespace {

constexpr int MH_NT = 512;        // threads per workgroup (forward and backward chain)
constexpr int MH_ROWS = 4;        // rows a workgroup carries through the chain
constexpr int MH_MAXL = 4;        // layers at most
constexpr int MH_MAXD = 512;      // width of any layer's OUTPUT at most
constexpr int MH_MAXD0 = 2048;    // width of the input x at most (lstm.py's n_hidden, padded or not)
constexpr int GM_MAXMIX = 32;     // mixture components at most
constexpr int GM_MAXPTS = 256;    // scored frames per row at most

struct MlpParams {
    const float* x;                  // (B, D[0])
    const float* W[MH_MAXL];         // (D[l], D[l+1]) row-major
    const float* b[MH_MAXL];
    const float* mask[MH_MAXL];      // optional (B, D[l+1]) multiplier behind layer l's activation (dropout, pre-scaled)
    float* a[MH_MAXL];               // forward: output of layer l behind activation and mask; backward: the same, read
    const float* dlast;              // backward: d loss / d (pre-activation of the last layer) (B, D[L])
    float* d[MH_MAXL];               // backward: d[l] = gradient at layer l's pre-activation, l < L-1 (workspace); d[L-1] unused
    float* gW[MH_MAXL];
    float* gb[MH_MAXL];
    float* dx;                       // (B, D[0]) or NULL
    int D[MH_MAXL + 1];
    int act[MH_MAXL];                // 0 none, 1 tanh, 2 relu, 3 exp
    int L, B, final_mode, n_mix, accumulate;
};

__device__ __forceinline__ float mh_act(float v, int act) {
    return act == 1 ? tanh_f(v) : act == 2 ? fmaxf(v, 0.f) : act == 3 ? __expf(v) : v;
}
__device__ __forceinline__ float mh_dact(float y, int act) {      // derivative from the OUTPUT y
    return act == 1 ? 1.f - y * y : act == 2 ? (y > 0.f ? 1.f : 0.f) : act == 3 ? y : 1.f;
}

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // a 16-byte global access that may start on any float

static_assert(MH_NT == MH_MAXD, "the forward stages one bias per thread and layer");
constexpr int MH_PF = 16;         // forward: weight vectors a thread holds in flight (one batch of loads)
constexpr int MH_FWD_LDS = MH_ROWS * MH_MAXD0 + 2 * MH_ROWS * MH_MAXD + 4 * MH_ROWS * MH_NT + MH_MAXL * MH_MAXD;   // floats: x | ping | pong | partial sums | biases

// Forward.  Layer l: thread (c4, ks) = (tid % cgp, tid / cgp) owns FOUR neighbouring columns (one 16-byte weight load per k) and the
// k-slice ks; cgp = the number of column groups rounded up to a power of two, so the block splits into 512 / cgp slices (32 at
// width 64, 8 at width 200..256): a thread's whole slice is ONE batch of at most sixteen independent loads - the chain is bound by
// load latency, not bandwidth - and the next layer's batch is issued before this layer's slices are summed, so it flies
// behind the reduction.  The rows' inputs come from LDS as broadcasts, the k-slices meet in LDS in a fixed order.
struct FwdMap { int Din, Dout, cgp, KS, c4, ks, k0, k1; };

__device__ __forceinline__ FwdMap fwd_map(const MlpParams& p, int l, int tid) {
    FwdMap m;
    m.Din = p.D[l]; m.Dout = p.D[l + 1];
    const int cg = (m.Dout + 3) >> 2;
    m.cgp = 16;
    while (m.cgp < cg) m.cgp <<= 1;
    m.KS = MH_NT / m.cgp;
    m.c4 = tid % m.cgp; m.ks = tid / m.cgp;
    const int kc = (((m.Din + m.KS - 1) / m.KS) + 3) & ~3;
    m.k0 = m.ks * kc;
    m.k1 = (m.k0 + kc < m.Din) ? m.k0 + kc : m.Din;
    return m;
}

// w[u] = W[kb + u][4 c4 .. 4 c4 + 3]; rows at or beyond k1 and columns beyond the layer give zeros
__device__ __forceinline__ void fwd_load(const float* __restrict__ W, const FwdMap& m, int kb, f32x4 (&w)[MH_PF]) {
    const int col = 4 * m.c4;
#pragma unroll
    for (int u = 0; u < MH_PF; ++u) {
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = kb + u;
        if (k < m.k1 && col < m.Dout) {
            const float* src = W + (size_t)k * m.Dout + col;
            if (col + 3 < m.Dout) v = *reinterpret_cast<const f32x4u*>(src);
            else {
                v[0] = src[0];
                if (col + 1 < m.Dout) v[1] = src[1];
                if (col + 2 < m.Dout) v[2] = src[2];
            }
        }
        w[u] = v;
    }
}

__global__ __launch_bounds__(MH_NT) void mlp_head_fwd_kernel(MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float mh_lds[];
    float* xin = mh_lds;                                  // [MH_ROWS][MH_MAXD0]
    float* bufA = xin + MH_ROWS * MH_MAXD0;               // [MH_ROWS][MH_MAXD]
    float* bufB = bufA + MH_ROWS * MH_MAXD;
    float* red = bufB + MH_ROWS * MH_MAXD;                // [KS][MH_ROWS][cgp][4] = 4 * MH_ROWS * MH_NT floats
    float* bias = red + 4 * MH_ROWS * MH_NT;              // [MH_MAXL][MH_MAXD]
    const int tid = threadIdx.x, row0 = blockIdx.x * MH_ROWS;
    MH_STAMP(0, 0);
    FwdMap m = fwd_map(p, 0, tid);
    f32x4 w[MH_PF];
    fwd_load(p.W[0], m, m.k0, w);                         // in flight while x and the biases are staged
    {
        // x and every bias, all loads issued before the first is waited for.  (A bias read inside a layer's reduction would queue
        // BEHIND the next layer's prefetched weights - loads return in order - and put their latency back on the critical path.)
        const int D0 = p.D[0], D0p = (D0 + 3) & ~3;       // the pad up to a multiple of four is read (times a zero weight): keep it finite
        float xv[MH_ROWS * MH_MAXD0 / MH_NT], bv[MH_MAXL];
#pragma unroll
        for (int u = 0; u < MH_ROWS * MH_MAXD0 / MH_NT; ++u) {
            const int e = u * MH_NT + tid, r = e / D0p, k = e - r * D0p;
            xv[u] = (e < MH_ROWS * D0p && row0 + r < p.B && k < D0) ? p.x[(size_t)(row0 + r) * D0 + k] : 0.f;
        }
#pragma unroll
        for (int l = 0; l < MH_MAXL; ++l) bv[l] = (l < p.L && tid < p.D[l + 1]) ? p.b[l][tid] : 0.f;
#pragma unroll
        for (int u = 0; u < MH_ROWS * MH_MAXD0 / MH_NT; ++u) {
            const int e = u * MH_NT + tid, r = e / D0p, k = e - r * D0p;
            if (e < MH_ROWS * D0p) xin[r * MH_MAXD0 + k] = xv[u];
        }
#pragma unroll
        for (int l = 0; l < MH_MAXL; ++l) bias[l * MH_MAXD + tid] = bv[l];      // MH_NT == MH_MAXD: one bias per thread and layer
    }
    __syncthreads();
    MH_STAMP(0, 1);
    const float* in = xin;
    int istr = MH_MAXD0;
    float* out = bufA;
    for (int l = 0; l < p.L; ++l) {
        const float* __restrict__ W = p.W[l];
        const bool last = (l == p.L - 1);
        f32x4 acc[MH_ROWS];
#pragma unroll
        for (int r = 0; r < MH_ROWS; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int kb = m.k0; kb < m.k1; kb += MH_PF) {
            if (kb != m.k0) fwd_load(W, m, kb, w);
#pragma unroll
            for (int u4 = 0; u4 < MH_PF / 4; ++u4) {
                const int k = kb + 4 * u4;
                if (k < m.k1) {
#pragma unroll
                    for (int r = 0; r < MH_ROWS; ++r) {
                        const f32x4 x4 = *reinterpret_cast<const f32x4*>(in + r * istr + k);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[r] += x4[q] * w[4 * u4 + q];
                    }
                }
            }
        }
        MH_STAMP(0, 2 + 4 * l);
#pragma unroll
        for (int r = 0; r < MH_ROWS; ++r) *reinterpret_cast<f32x4*>(red + ((m.ks * MH_ROWS + r) * m.cgp + m.c4) * 4) = acc[r];
        const int Dout = m.Dout, KS = m.KS, cgp = m.cgp;
        if (!last) {                                      // the next layer's first batch flies behind this layer's barrier and reduction
            m = fwd_map(p, l + 1, tid);
            fwd_load(p.W[l + 1], m, m.k0, w);
        }
        __syncthreads();
        MH_STAMP(0, 3 + 4 * l);
        const int Dp = (Dout + 3) & ~3;
        for (int e = tid; e < MH_ROWS * Dp; e += MH_NT) {
            const int r = e / Dp, oc = e - r * Dp, row = row0 + r;
            float v = 0.f;
            if (oc < Dout) {
                v = bias[l * MH_MAXD + oc];
                const float* rp = red + (r * cgp + (oc >> 2)) * 4 + (oc & 3);
                const int qs = MH_ROWS * cgp * 4;
                int q = 0;
                for (; q + 4 <= KS; q += 4) {             // fixed order; four independent LDS reads in flight (KS is a power of two >= 4)
                    const float t0 = rp[q * qs], t1 = rp[(q + 1) * qs], t2 = rp[(q + 2) * qs], t3 = rp[(q + 3) * qs];
                    v += t0; v += t1; v += t2; v += t3;
                }
                for (; q < KS; ++q) v += rp[q * qs];
                if (!(last && p.final_mode)) v = mh_act(v, p.act[l]);
                if (p.mask[l] && row < p.B) v *= p.mask[l][(size_t)row * Dout + oc];
                if (row < p.B && !(last && p.final_mode)) p.a[l][(size_t)row * Dout + oc] = v;
            }
            out[r * MH_MAXD + oc] = v;
        }
        MH_STAMP(0, 4 + 4 * l);
        __syncthreads();
        MH_STAMP(0, 5 + 4 * l);
        in = out;
        istr = MH_MAXD;
        out = (out == bufA) ? bufB : bufA;
    }
    if (p.final_mode) {      // mixture split of lstm.py:386-399 on the raw last layer (now in `in`)
        const int n = p.n_mix, Dl = p.D[p.L];
        for (int e = tid; e < MH_ROWS * Dl; e += MH_NT) {
            const int r = e / Dl, j = e - r * Dl, row = row0 + r;
            if (row >= p.B) continue;
            const float v = in[r * MH_MAXD + j];
            float y;
            if (j < n) {          // exp / sum exp, no max subtraction (lstm.py:394-396)
                float s = 0.f;
                for (int q = 0; q < n; ++q) s += __expf(in[r * MH_MAXD + q]);
                y = __expf(v) / s;
            } else if (j < 4 * n) y = v;
            else if (j < 7 * n) y = __expf(v);
            else y = tanh_f(v);
            p.a[p.L - 1][(size_t)row * Dl + j] = y;
        }
    }
    MH_STAMP(0, 20);
}

// Backward chain for four rows: d_{l-1}[r][i] = act'(a_{l-1}[r][i]) mask_{l-1}[r][i] sum_j d_l[r][j] W_l[i][j] - a product ALONG the
// rows of W_l with the lanes ACROSS them, i.e. against the grain of memory.  W_l is therefore read in tiles of whole rows (one
// contiguous block, 16-byte loads, the next tile already in registers while this one is used) and laid into LDS with a row
// stride S, S / 4 odd: sixteen neighbouring rows then start in sixteen different 16-byte bank slots, so a lane reads four
// neighbouring j of ITS row as one conflict-free ds_read_b128.  The gradients sit in LDS transposed, dT[j] = the four rows' d at
// j (one ds_read_b128 for all lanes of a wave).  Thread (il, js) owns a 4 x 4 register block per step: rows il + {0,1,2,3} IC/4
// of the tile, four j of its slice js, the four batch rows - eight LDS reads feed 64 multiply-adds.  Slices meet in LDS, fixed order.
constexpr int CH_JQ = 4;          // 4-wide j steps of a slice whose gradients are held in registers (slices are 8 .. 16 wide)
constexpr int CH_PF = 13;         // 16-byte vectors of a tile per thread: 13 * 4 * 512 = 26624 floats
constexpr int CH_TILE = 128 * 204;                        // floats of a tile: 128 rows of the 200-wide mixture layer (S = 204)
constexpr int CH_LDS = CH_TILE + 16 * MH_NT + 2 * MH_ROWS * MH_MAXD;      // tile | slice sums | dT ping | dT pong

struct ChTile { int l, i0, IC, rows, Dout, S, cnt; const float* src; };

__device__ __forceinline__ ChTile ch_tile(const MlpParams& p, int l, int i0) {
    ChTile t;
    t.l = l; t.i0 = i0;
    const int Din = p.D[l];
    t.Dout = p.D[l + 1];
    const int D4 = (t.Dout + 3) & ~3;
    t.S = (D4 & 4) ? D4 : D4 + 4;                         // multiple of four with S / 4 odd
    t.IC = l > 0 ? 128 : MH_NT;                           // rows of W per tile: the largest of 512 (layer 0) / 128, ..., 32 that fits the
    while (t.IC > 32 && (t.IC * t.S > CH_TILE || t.IC >= 2 * Din)) t.IC >>= 1;      // tile and is not twice the layer
    t.rows = (Din - i0 < t.IC) ? Din - i0 : t.IC;
    t.cnt = t.rows * t.Dout;
    t.src = p.W[l] + (size_t)i0 * t.Dout;
    return t;
}

__device__ __forceinline__ void ch_load(const ChTile& t, int tid, f32x4 (&g)[CH_PF]) {
#pragma unroll
    for (int v = 0; v < CH_PF; ++v) {
        const int e = (v * MH_NT + tid) * 4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (e + 3 < t.cnt) x = *reinterpret_cast<const f32x4u*>(t.src + e);
        else if (e < t.cnt) {
            x[0] = t.src[e];
            if (e + 1 < t.cnt) x[1] = t.src[e + 1];
            if (e + 2 < t.cnt) x[2] = t.src[e + 2];
        }
        g[v] = x;
    }
}

__device__ __forceinline__ void ch_store(const ChTile& t, int tid, const f32x4 (&g)[CH_PF], float* __restrict__ wt) {
    const bool vec = (t.Dout & 3) == 0;                   // four consecutive elements never straddle a row
#pragma unroll
    for (int v = 0; v < CH_PF; ++v) {
        const int e = (v * MH_NT + tid) * 4;
        if (e < t.cnt) {
            int i = e / t.Dout, j = e - i * t.Dout;
            if (vec) *reinterpret_cast<f32x4*>(wt + i * t.S + j) = g[v];
            else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (e + q < t.cnt) wt[i * t.S + j] = g[v][q];
                    if (++j == t.Dout) { j = 0; ++i; }
                }
            }
        }
    }
    if (!vec)                                             // columns Dout .. the next multiple of four are read (times a zero gradient)
        for (int e = tid; e < t.rows * 4; e += MH_NT) {
            const int i = e >> 2, j = t.Dout + (e & 3);
            if (j < ((t.Dout + 3) & ~3)) wt[i * t.S + j] = 0.f;
        }
}

__global__ __launch_bounds__(MH_NT) void mlp_head_bwd_chain_kernel(MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float mh_lds[];
    float* wt = mh_lds;                                   // [CH_TILE]
    float* red = wt + CH_TILE;                            // [JS][IC][4] = 16 * MH_NT floats
    float* bufA = red + 16 * MH_NT;                       // dT: [MH_MAXD][MH_ROWS]
    float* bufB = bufA + MH_ROWS * MH_MAXD;
    const int tid = threadIdx.x, row0 = blockIdx.x * MH_ROWS;
    const int lmin = p.dx ? 0 : 1;                        // layer 0's product is only needed for dx
    if (p.L - 1 < lmin) return;
    MH_STAMP(1, 0);
    ChTile t = ch_tile(p, p.L - 1, 0);
    f32x4 g[CH_PF];
    ch_load(t, tid, g);
    float* dcur = bufA;
    float* dnew = bufB;
    {
        const int Dl = p.D[p.L], Dlp = (Dl + 3) & ~3;
        for (int e = tid; e < MH_ROWS * Dlp; e += MH_NT) {
            const int j = e >> 2, r = e & 3;
            dcur[e] = (row0 + r < p.B && j < Dl) ? p.dlast[(size_t)(row0 + r) * Dl + j] : 0.f;
        }
    }
    int n_tile = 0;
    for (;;) {
        MH_STAMP(1, 1 + 5 * n_tile);
        ch_store(t, tid, g, wt);
        const int l = t.l, Din = p.D[l], Dout = t.Dout, IC = t.IC, rows = t.rows, i0 = t.i0, S = t.S;
        // what the tile's outputs are multiplied with: act'(a_{l-1}) mask_{l-1} of output (i, r) = thread - in flight during the tile
        float fac = 1.f;
        if (l > 0) {
            const int i = i0 + (tid >> 2), row = row0 + (tid & 3);
            if ((tid >> 2) < rows && row < p.B) {
                fac = mh_dact(p.a[l - 1][(size_t)row * Din + i], p.act[l - 1]);
                if (p.mask[l - 1]) fac *= p.mask[l - 1][(size_t)row * Din + i];
            } else fac = 0.f;
        }
        MH_STAMP(1, 2 + 5 * n_tile);
        __syncthreads();
        MH_STAMP(1, 3 + 5 * n_tile);
        const bool layer_done = (i0 + IC >= Din);
        const bool more = !layer_done || l - 1 >= lmin;
        if (more) {                                       // the next tile: more rows of this layer, else the layer below
            t = layer_done ? ch_tile(p, l - 1, 0) : ch_tile(p, l, i0 + IC);
            ch_load(t, tid, g);
        }
        const int IC4 = IC >> 2, JS = MH_NT / IC4;
        const int il = tid % IC4, js = tid / IC4;
        const int jc = (((Dout + JS - 1) / JS) + 3) & ~3;
        const int D4 = (Dout + 3) & ~3;
        const int j0 = js * jc, j1 = (j0 + jc < D4) ? j0 + jc : D4;
        f32x4 acc[4];
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) acc[ib] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* wr = wt + il * S;
        if (jc <= 4 * CH_JQ) {                            // the slice's gradients: 16 j x 4 rows in registers, read once per tile
            f32x4 dq[4 * CH_JQ];
#pragma unroll
            for (int q = 0; q < 4 * CH_JQ; ++q)
                dq[q] = (j0 + q < j1) ? *reinterpret_cast<const f32x4*>(dcur + (j0 + q) * MH_ROWS) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int jq = 0; jq < CH_JQ; ++jq) {
                const int j = j0 + 4 * jq;
                if (j < j1) {
#pragma unroll
                    for (int ib = 0; ib < 4; ++ib) {
                        const f32x4 w = *reinterpret_cast<const f32x4*>(wr + ib * IC4 * S + j);
#pragma unroll
                        for (int q = 0; q < 4; ++q) acc[ib] += w[q] * dq[4 * jq + q];
                    }
                }
            }
        } else {
            for (int j = j0; j < j1; j += 4) {
                f32x4 dq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) dq[q] = *reinterpret_cast<const f32x4*>(dcur + (j + q) * MH_ROWS);
#pragma unroll
                for (int ib = 0; ib < 4; ++ib) {
                    const f32x4 w = *reinterpret_cast<const f32x4*>(wr + ib * IC4 * S + j);
#pragma unroll
                    for (int q = 0; q < 4; ++q) acc[ib] += w[q] * dq[4 * 0 + q];
                }
            }
        }
#pragma unroll
        for (int ib = 0; ib < 4; ++ib) *reinterpret_cast<f32x4*>(red + (js * IC + il + ib * IC4) * 4) = acc[ib];
        MH_STAMP(1, 4 + 5 * n_tile);
        __syncthreads();
        for (int e = tid; e < rows * MH_ROWS; e += MH_NT) {     // output (i, r) = e: one per thread up to 128 rows (every layer but 0)
            const int ii = e >> 2, r = e & 3, i = i0 + ii, row = row0 + r;
            float v = 0.f;
            for (int q = 0; q < JS; q += 4) {             // fixed order; JS is a multiple of four (8 .. 32)
                const float t0 = red[(q * IC) * 4 + e], t1 = red[((q + 1) * IC) * 4 + e], t2 = red[((q + 2) * IC) * 4 + e],
                            t3 = red[((q + 3) * IC) * 4 + e];
                v += t0; v += t1; v += t2; v += t3;
            }
            if (l > 0) {
                v *= fac;                                 // rows <= 128 there: e == tid
                if (row < p.B) p.d[l - 1][(size_t)row * Din + i] = v;
                dnew[i * MH_ROWS + r] = v;
            } else if (row < p.B) {
                p.dx[(size_t)row * Din + i] = v;
            }
        }
        if (l > 0 && layer_done && (Din & 3) && tid < 16) {     // zero pad of the next layer's 16-byte reads
            const int j = Din + (tid >> 2);
            if (j < ((Din + 3) & ~3)) dnew[j * MH_ROWS + (tid & 3)] = 0.f;
        }
        MH_STAMP(1, 5 + 5 * n_tile);
        __syncthreads();
        ++n_tile;
        (void)n_tile;
        if (!more) break;
        if (layer_done) { float* s_ = dcur; dcur = dnew; dnew = s_; }
    }
    MH_STAMP(1, 1 + 5 * n_tile);
}

// Weight and bias gradients of every layer in one launch: workgroup (l, i0) owns eight rows i of gW_l, thread j a column:
// gW_l[i][j] (+)= sum_n in_l[n][i] d_l[n][j], in_0 = x, in_l = a_{l-1}; the workgroup with i0 = 0 also sums gb_l.
constexpr int WG_IR = 8;
constexpr int WG_NB = 64;
__global__ __launch_bounds__(256) void mlp_head_wgrad_kernel(MlpParams p, int blk1, int blk2, int blk3) {
    __shared__ float as[WG_NB][WG_IR];
    int blk = blockIdx.x, l = 0;
    if (blk >= blk3) { l = 3; blk -= blk3; }
    else if (blk >= blk2) { l = 2; blk -= blk2; }
    else if (blk >= blk1) { l = 1; blk -= blk1; }
    const int Din = p.D[l], Dout = p.D[l + 1], i0 = blk * WG_IR, tid = threadIdx.x;
    const float* __restrict__ inp = l == 0 ? p.x : p.a[l - 1];
    const float* __restrict__ dl = (l == p.L - 1) ? p.dlast : p.d[l];
    for (int jb = 0; jb < Dout; jb += 256) {
        const int j = jb + tid;
        float acc[WG_IR], accb = 0.f;
#pragma unroll
        for (int u = 0; u < WG_IR; ++u) acc[u] = 0.f;
        for (int n0 = 0; n0 < p.B; n0 += WG_NB) {
            const int nn = (p.B - n0 < WG_NB) ? p.B - n0 : WG_NB;
            __syncthreads();
            for (int e = tid; e < nn * WG_IR; e += 256) {
                const int n = e / WG_IR, u = e - n * WG_IR;
                as[n][u] = (i0 + u < Din) ? inp[(size_t)(n0 + n) * Din + i0 + u] : 0.f;
            }
            __syncthreads();
            if (j < Dout) {
                int n = 0;
                for (; n + 8 <= nn; n += 8) {
                    float dv[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) dv[q] = dl[(size_t)(n0 + n + q) * Dout + j];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        accb += dv[q];
#pragma unroll
                        for (int u = 0; u < WG_IR; ++u) acc[u] = fmaf(as[n + q][u], dv[q], acc[u]);
                    }
                }
                for (; n < nn; ++n) {
                    const float dv = dl[(size_t)(n0 + n) * Dout + j];
                    accb += dv;
#pragma unroll
                    for (int u = 0; u < WG_IR; ++u) acc[u] = fmaf(as[n][u], dv, acc[u]);
                }
            }
        }
        if (j < Dout) {
#pragma unroll
            for (int u = 0; u < WG_IR; ++u)
                if (i0 + u < Din) {
                    float* g = p.gW[l] + (size_t)(i0 + u) * Dout + j;
                    *g = p.accumulate ? *g + acc[u] : acc[u];
                }
            if (i0 == 0) {
                float* g = p.gb[l] + j;
                *g = p.accumulate ? *g + accb : accb;
            }
        }
    }
}

// ---- 3x3 symmetric helpers (fp64) ----
struct Sym3 { double a00, a01, a02, a11, a12, a22; };

// One Jacobi rotation annihilating a[p][q] of the symmetric matrix held as a full 3x3 in registers; V accumulates the rotations.
template <int P, int Q>
__device__ __forceinline__ void jacobi_rot(double (&a)[3][3], double (&V)[3][3]) {
    const double apq = a[P][Q];
    if (fabs(apq) < 1e-300) return;
    const double theta = (a[Q][Q] - a[P][P]) / (2.0 * apq);
    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    constexpr int R = 3 - P - Q;
    const double app = a[P][P], aqq = a[Q][Q], arp = a[R][P], arq = a[R][Q];
    a[P][P] = app - t * apq;
    a[Q][Q] = aqq + t * apq;
    a[P][Q] = a[Q][P] = 0.0;
    a[R][P] = a[P][R] = c * arp - s * arq;
    a[R][Q] = a[Q][R] = s * arp + c * arq;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const double vp = V[k][P], vq = V[k][Q];
        V[k][P] = c * vp - s * vq;
        V[k][Q] = s * vp + c * vq;
    }
}

// smallest eigenvalue of S and its unit eigenvector (cyclic Jacobi, converges quadratically: 6 sweeps are past fp64 for 3x3)
__device__ void sym3_min_eig(const Sym3& S, double& lam, double (&v)[3]) {
    double a[3][3] = {{S.a00, S.a01, S.a02}, {S.a01, S.a11, S.a12}, {S.a02, S.a12, S.a22}};
    double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int sweep = 0; sweep < 6; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        if (off <= 1e-18 * (fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]))) break;      // diagonal to fp64: usually after 3-4 sweeps
        jacobi_rot<0, 1>(a, V);
        jacobi_rot<0, 2>(a, V);
        jacobi_rot<1, 2>(a, V);
    }
    int k = 0;
    lam = a[0][0];
    if (a[1][1] < lam) { lam = a[1][1]; k = 1; }
    if (a[2][2] < lam) { lam = a[2][2]; k = 2; }
#pragma unroll
    for (int r = 0; r < 3; ++r) v[r] = k == 0 ? V[r][0] : k == 1 ? V[r][1] : V[r][2];
}

struct MixSetup {      // per mixture component, LDS
    double P[6];       // inverse of the repaired covariance: p00 p01 p02 p11 p12 p22
    double mu[3];
    double v[3];       // eigenvector of the smallest eigenvalue (repair branch only)
    double lognorm;    // -0.5 log((2 pi)^3 det)
    double L[6];       // Cholesky factor l00 l10 l11 l20 l21 l22 (sampler)
    int repaired;
};

// covariance of cost.py:376-378 with the repair of cost.py:335-348, its inverse, log normaliser and Cholesky factor
__device__ void mix_setup(const float* __restrict__ prm, int n, int m, MixSetup& o) {
    const double s1 = prm[4 * n + 3 * m], s2 = prm[4 * n + 3 * m + 1], s3 = prm[4 * n + 3 * m + 2];
    const double r12 = prm[7 * n + 3 * m], r13 = prm[7 * n + 3 * m + 1], r23 = prm[7 * n + 3 * m + 2];
    Sym3 S = {s1 * s1, r12 * s1 * s2, r13 * s1 * s3, s2 * s2, r23 * s2 * s3, s3 * s3};
    double c00 = S.a11 * S.a22 - S.a12 * S.a12, c01 = S.a02 * S.a12 - S.a01 * S.a22, c02 = S.a01 * S.a12 - S.a02 * S.a11;
    double det = S.a00 * c00 + S.a01 * c01 + S.a02 * c02;
    // Sylvester: all leading minors positive <=> positive definite <=> smallest eigenvalue > 0: no repair, and the eigen-decomposition
    // (dependent fp64 divisions and square roots, most of this kernel's time) is only run for the matrices that need it
    o.repaired = 0;
    o.v[0] = o.v[1] = o.v[2] = 0.0;
    if (!(S.a00 > 0.0 && S.a00 * S.a11 - S.a01 * S.a01 > 0.0 && det > 0.0)) {
        double lam;
        sym3_min_eig(S, lam, o.v);
        o.repaired = lam < 0.0;
        if (o.repaired) {
            const double k = -10.0 * lam;
            S.a00 += k; S.a11 += k; S.a22 += k;
            c00 = S.a11 * S.a22 - S.a12 * S.a12; c01 = S.a02 * S.a12 - S.a01 * S.a22; c02 = S.a01 * S.a12 - S.a02 * S.a11;
            det = S.a00 * c00 + S.a01 * c01 + S.a02 * c02;
        }
    }
    const double id = 1.0 / det;
    o.P[0] = c00 * id; o.P[1] = c01 * id; o.P[2] = c02 * id;
    o.P[3] = (S.a00 * S.a22 - S.a02 * S.a02) * id;
    o.P[4] = (S.a01 * S.a02 - S.a00 * S.a12) * id;
    o.P[5] = (S.a00 * S.a11 - S.a01 * S.a01) * id;
    o.lognorm = -0.5 * (3.0 * 1.8378770664093453 + log(det));      // log(2 pi) = 1.83787706640934...
    o.mu[0] = prm[n + 3 * m]; o.mu[1] = prm[n + 3 * m + 1]; o.mu[2] = prm[n + 3 * m + 2];
    const double l00 = sqrt(S.a00), l10 = S.a01 / l00, l20 = S.a02 / l00;
    const double l11 = sqrt(S.a11 - l10 * l10), l21 = (S.a12 - l20 * l10) / l11;
    const double l22 = sqrt(S.a22 - l20 * l20 - l21 * l21);
    o.L[0] = l00; o.L[1] = l10; o.L[2] = l11; o.L[3] = l20; o.L[4] = l21; o.L[5] = l22;
}

// costfunc.mixture_3d_gaussian_loss and its gradient at the head's PRE-activations, one workgroup per row.
//   part[b] = sum_t -log(S_t + 1e-20),  S_t = sum_m [pi_m] N(y_t; mu_m, Sigma'_m);   loss = scale * sum_b part[b]
//   dpre (B, 10 n): zero for the pi logits unless weight_by_pi (the reference never multiplies by pi, cost.py:532-538).
// The LAST workgroup to finish (a ticket in the workspace, left at zero again) adds the B parts in row order: one launch, and
// the same loss bits whatever the order the workgroups ran in.
__global__ __launch_bounds__(256) void gmm3d_loss_grad_kernel(const float* __restrict__ params, const float* __restrict__ y, long ldy,
                                                              unsigned* __restrict__ ticket, float* __restrict__ part,
                                                              float* __restrict__ loss, float* __restrict__ dpre, int n, int npts,
                                                              int tc, float scale, int weight_by_pi) {
    extern __shared__ __attribute__((aligned(16))) double gm_lds[];
    __shared__ MixSetup ms[GM_MAXMIX];
    __shared__ double wt[GM_MAXPTS];          // w_t = -scale / (S_t + eps)
    __shared__ double acc10[GM_MAXMIX][10];   // per mixture: d/dmu (3), G (6), d/dpi
    __shared__ double lsum[256];
    __shared__ int is_last;
    double* pm = gm_lds;                      // [n][npts] densities
    double* terms = pm + n * npts;            // [n][tc][10] per-frame terms of the gradient sums, a chunk of tc frames at a time
    const int b = blockIdx.x, B = gridDim.x, tid = threadIdx.x;
    const float* prm = params + (size_t)b * 10 * n;
    const float* yb = y + (size_t)b * ldy;
    MH_STAMP(2, 0);
    if (tid < n) mix_setup(prm, n, tid, ms[tid]);
    for (int e = tid; e < GM_MAXMIX * 10; e += 256) acc10[e / 10][e % 10] = 0.0;
    __syncthreads();
    MH_STAMP(2, 1);
    for (int e = tid; e < n * npts; e += 256) {
        const int m = e / npts, t = e - m * npts;
        const MixSetup& s = ms[m];
        const double d0 = (double)yb[3 * t] - s.mu[0], d1 = (double)yb[3 * t + 1] - s.mu[1], d2 = (double)yb[3 * t + 2] - s.mu[2];
        const double q = s.P[0] * d0 * d0 + s.P[3] * d1 * d1 + s.P[5] * d2 * d2 + 2.0 * (s.P[1] * d0 * d1 + s.P[2] * d0 * d2 + s.P[4] * d1 * d2);
        pm[e] = exp(s.lognorm - 0.5 * q);
    }
    __syncthreads();
    MH_STAMP(2, 2);
    double lacc = 0.0;
    for (int t = tid; t < npts; t += 256) {
        double S = 0.0;
        for (int m = 0; m < n; ++m) S += (weight_by_pi ? (double)prm[m] : 1.0) * pm[m * npts + t];
        lacc -= log(S + 1e-20);
        wt[t] = -(double)scale / (S + 1e-20);
    }
    lsum[tid] = lacc;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if (tid < s_) lsum[tid] += lsum[tid + s_];
        __syncthreads();
    }
    MH_STAMP(2, 3);
    // the ten sums over t of every mixture m - c 0..2: sum w p a_c; 3..8: sum w p (a a^T - P) / 2 (pairs 00 01 02 11 12 22); 9: sum w p / pi.
    // Thread (m, t) forms the ten terms of its frame, thread (m, c) then adds a chunk's terms in frame order (fixed order).
    double a10 = 0.0;                          // running sum of thread (m, c) = tid
    for (int t0 = 0; t0 < npts; t0 += tc) {
        const int tn = (npts - t0 < tc) ? npts - t0 : tc;
        for (int e = tid; e < n * tn; e += 256) {
            const int m = e / tn, tl = e - m * tn, t = t0 + tl;
            const MixSetup& s = ms[m];
            const double pim = weight_by_pi ? (double)prm[m] : 1.0;
            const double p = pm[m * npts + t], wp = wt[t] * pim * p;      // dL/dp_mt * p_mt
            const double d0 = (double)yb[3 * t] - s.mu[0], d1 = (double)yb[3 * t + 1] - s.mu[1], d2 = (double)yb[3 * t + 2] - s.mu[2];
            const double a0 = s.P[0] * d0 + s.P[1] * d1 + s.P[2] * d2;
            const double a1 = s.P[1] * d0 + s.P[3] * d1 + s.P[4] * d2;
            const double a2 = s.P[2] * d0 + s.P[4] * d1 + s.P[5] * d2;
            double* o = terms + (size_t)(m * tc + tl) * 10;
            const double h = 0.5 * wp;
            o[0] = wp * a0; o[1] = wp * a1; o[2] = wp * a2;
            o[3] = h * (a0 * a0 - s.P[0]); o[4] = h * (a0 * a1 - s.P[1]); o[5] = h * (a0 * a2 - s.P[2]);
            o[6] = h * (a1 * a1 - s.P[3]); o[7] = h * (a1 * a2 - s.P[4]); o[8] = h * (a2 * a2 - s.P[5]);
            o[9] = wt[t] * p;
        }
        __syncthreads();
        for (int e = tid; e < n * 10; e += 256) {         // n <= 32: at most two per thread, the second kept in LDS
            const int m = e / 10, c = e - m * 10;
            double a = (e == tid) ? a10 : acc10[m][c];
            for (int tl = 0; tl < tn; ++tl) a += terms[(size_t)(m * tc + tl) * 10 + c];
            if (e == tid) a10 = a; else acc10[m][c] = a;
        }
        __syncthreads();
    }
    if (tid < n * 10) acc10[tid / 10][tid % 10] = a10;
    __syncthreads();
    MH_STAMP(2, 4);
    float* dp = dpre + (size_t)b * 10 * n;
    if (tid < n) {
        const int m = tid;
        const MixSetup& s = ms[m];
        double G[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) G[k] = acc10[m][3 + k];         // G00 G01 G02 G11 G12 G22
        if (s.repaired) {     // Sigma' = Sigma - 10 lambda_min I, d lambda_min / d Sigma = v v^T
            const double k = -10.0 * (G[0] + G[3] + G[5]);
            G[0] += k * s.v[0] * s.v[0]; G[1] += k * s.v[0] * s.v[1]; G[2] += k * s.v[0] * s.v[2];
            G[3] += k * s.v[1] * s.v[1]; G[4] += k * s.v[1] * s.v[2]; G[5] += k * s.v[2] * s.v[2];
        }
        const double s1 = prm[4 * n + 3 * m], s2 = prm[4 * n + 3 * m + 1], s3 = prm[4 * n + 3 * m + 2];
        const double r12 = prm[7 * n + 3 * m], r13 = prm[7 * n + 3 * m + 1], r23 = prm[7 * n + 3 * m + 2];
        const double ds1 = 2.0 * (s1 * G[0] + r12 * s2 * G[1] + r13 * s3 * G[2]);
        const double ds2 = 2.0 * (s2 * G[3] + r12 * s1 * G[1] + r23 * s3 * G[4]);
        const double ds3 = 2.0 * (s3 * G[5] + r13 * s1 * G[2] + r23 * s2 * G[4]);
        dp[n + 3 * m] = (float)acc10[m][0]; dp[n + 3 * m + 1] = (float)acc10[m][1]; dp[n + 3 * m + 2] = (float)acc10[m][2];
        dp[4 * n + 3 * m] = (float)(ds1 * s1); dp[4 * n + 3 * m + 1] = (float)(ds2 * s2); dp[4 * n + 3 * m + 2] = (float)(ds3 * s3);
        dp[7 * n + 3 * m] = (float)(2.0 * G[1] * s1 * s2 * (1.0 - r12 * r12));
        dp[7 * n + 3 * m + 1] = (float)(2.0 * G[2] * s1 * s3 * (1.0 - r13 * r13));
        dp[7 * n + 3 * m + 2] = (float)(2.0 * G[4] * s2 * s3 * (1.0 - r23 * r23));
        float g = 0.f;
        if (weight_by_pi) {   // softmax backward; acc10[.][9] = d loss / d pi
            double dot = 0.0;
            for (int q = 0; q < n; ++q) dot += (double)prm[q] * acc10[q][9];
            g = (float)((double)prm[m] * (acc10[m][9] - dot));
        }
        dp[m] = g;
    }
    MH_STAMP(2, 5);
    if (!loss) return;
    if (tid == 0) {
        __hip_atomic_store(part + b, (float)lsum[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __threadfence();
        const unsigned done = atomicAdd(ticket, 1u);
        is_last = (done == (unsigned)B - 1u);
    }
    __syncthreads();
    if (is_last) {
        __threadfence();
        double a = 0.0;
        for (int i = tid; i < B; i += 256) a += (double)__hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lsum[tid] = a;
        __syncthreads();
        for (int s_ = 128; s_ > 0; s_ >>= 1) {
            if (tid < s_) lsum[tid] += lsum[tid + s_];
            __syncthreads();
        }
        if (tid == 0) {
            loss[0] = (float)(lsum[0] * (double)scale);
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // the next call finds it at zero
        }
    }
    MH_STAMP(2, 6);
}

// One draw per frame: component by inverse CDF over pi, then mu + L z (utility.sample_mixture_3D's documented intent).
__global__ __launch_bounds__(256) void gmm3d_sample_kernel(const float* __restrict__ params, const float* __restrict__ u,
                                                           const float* __restrict__ z, float* __restrict__ out, long ldo, int n, int npts) {
    __shared__ MixSetup ms[GM_MAXMIX];
    __shared__ float cum[GM_MAXMIX];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* prm = params + (size_t)b * 10 * n;
    if (tid < n) mix_setup(prm, n, tid, ms[tid]);
    if (tid == 0) {
        float c = 0.f;
        for (int m = 0; m < n; ++m) { c += prm[m]; cum[m] = c; }
    }
    __syncthreads();
    for (int f = tid; f < npts; f += 256) {
        const float uu = u[(size_t)b * npts + f];
        int m = 0;
        while (m < n - 1 && !(cum[m] > uu)) ++m;
        const MixSetup& s = ms[m];
        const double z0 = z[((size_t)b * npts + f) * 3], z1 = z[((size_t)b * npts + f) * 3 + 1], z2 = z[((size_t)b * npts + f) * 3 + 2];
        float* o = out + (size_t)b * ldo + 3 * f;
        o[0] = (float)(s.mu[0] + s.L[0] * z0);
        o[1] = (float)(s.mu[1] + s.L[1] * z0 + s.L[2] * z1);
        o[2] = (float)(s.mu[2] + s.L[3] * z0 + s.L[4] * z1 + s.L[5] * z2);
    }
}

bool mlp_dims_ok(int B, int L, const int* dims) {
    if (B < 1 || L < 1 || L > MH_MAXL || !dims) return false;
    for (int l = 0; l <= L; ++l)
        if (dims[l] < 1 || dims[l] > (l == 0 ? MH_MAXD0 : MH_MAXD)) return false;
    return true;
}

int mh_check(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s launch: %s", what, hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace

}  // namespace fov

using namespace fov;

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_mh_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mh_stamps), sizeof(unsigned long long) * 3 * 48);
}
#endif

extern "C" {

int fov_mlp_head_supported(int B, int L, const int* dims) { return mlp_dims_ok(B, L, dims) ? 1 : 0; }

int fov_mlp_head_fwd(const float* x, const float* const* W, const float* const* b, const float* const* masks, float* const* acts,
                     const int* dims, const int* act_codes, int L, int final_mode, int n_mix, int B, fov_stream_t stream) {
    if (B == 0) return FOV_OK;
    if (!x || !W || !b || !acts || !dims || !act_codes) { set_error("fov_mlp_head_fwd: invalid argument"); return FOV_ERR_INVALID; }
    if (!mlp_dims_ok(B, L, dims)) { set_error("fov_mlp_head_fwd: unsupported shape (1..4 layers, input width <= 2048, layer widths <= 512)"); return FOV_ERR_UNSUPPORTED; }
    if (final_mode && (final_mode != 1 || n_mix < 1 || n_mix > GM_MAXMIX || dims[L] != 10 * n_mix)) {
        set_error("fov_mlp_head_fwd: the mixture split needs a last layer of 10 * n_mix <= 320 units"); return FOV_ERR_INVALID;
    }
    MlpParams p = {};
    p.x = x; p.L = L; p.B = B; p.final_mode = final_mode; p.n_mix = n_mix;
    for (int l = 0; l < L; ++l) {
        if (!W[l] || !b[l] || !acts[l]) { set_error("fov_mlp_head_fwd: null layer pointer"); return FOV_ERR_INVALID; }
        p.W[l] = W[l]; p.b[l] = b[l]; p.a[l] = acts[l]; p.mask[l] = masks ? masks[l] : nullptr; p.act[l] = act_codes[l];
    }
    for (int l = 0; l <= L; ++l) p.D[l] = dims[l];
    const size_t lds = sizeof(float) * (size_t)MH_FWD_LDS;
    int rc = ensure_dynamic_lds((const void*)mlp_head_fwd_kernel, lds, MH_NT);
    if (rc) return rc;
    hipLaunchKernelGGL(mlp_head_fwd_kernel, dim3((unsigned)((B + MH_ROWS - 1) / MH_ROWS)), dim3(MH_NT), lds, (hipStream_t)stream, p);
    return mh_check("mlp head forward");
}

size_t fov_mlp_head_bwd_workspace_bytes(int B, int L, const int* dims) {
    if (!mlp_dims_ok(B, L, dims)) return 0;
    size_t n = 0;
    for (int l = 1; l < L; ++l) n += (size_t)B * dims[l];
    return sizeof(float) * (n + 64);
}

int fov_mlp_head_bwd(const float* x, const float* const* W, const float* const* masks, const float* const* acts, const float* dlast,
                     float* const* gW, float* const* gb, float* dx, const int* dims, const int* act_codes, int L, int B,
                     int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B == 0) return FOV_OK;
    if (!x || !W || !acts || !dlast || !gW || !gb || !dims || !act_codes) { set_error("fov_mlp_head_bwd: invalid argument"); return FOV_ERR_INVALID; }
    if (!mlp_dims_ok(B, L, dims)) { set_error("fov_mlp_head_bwd: unsupported shape"); return FOV_ERR_UNSUPPORTED; }
    if (!workspace || workspace_bytes < fov_mlp_head_bwd_workspace_bytes(B, L, dims)) { set_error("fov_mlp_head_bwd: workspace too small"); return FOV_ERR_WORKSPACE; }
    MlpParams p = {};
    p.x = x; p.L = L; p.B = B; p.dlast = dlast; p.dx = dx; p.accumulate = accumulate ? 1 : 0;
    float* ws = static_cast<float*>(workspace);
    for (int l = 0; l < L; ++l) {
        if (!W[l] || !gW[l] || !gb[l] || (l < L - 1 && !acts[l])) { set_error("fov_mlp_head_bwd: null layer pointer"); return FOV_ERR_INVALID; }
        p.W[l] = W[l]; p.a[l] = const_cast<float*>(acts[l]); p.mask[l] = masks ? masks[l] : nullptr; p.act[l] = act_codes[l];
        p.gW[l] = gW[l]; p.gb[l] = gb[l];
        if (l < L - 1) { p.d[l] = ws; ws += (size_t)B * dims[l + 1]; }
    }
    for (int l = 0; l <= L; ++l) p.D[l] = dims[l];
    if (L > 1 || dx) {
        const size_t lds = sizeof(float) * (size_t)CH_LDS;
        int rc = ensure_dynamic_lds((const void*)mlp_head_bwd_chain_kernel, lds, MH_NT);
        if (rc) return rc;
        hipLaunchKernelGGL(mlp_head_bwd_chain_kernel, dim3((unsigned)((B + MH_ROWS - 1) / MH_ROWS)), dim3(MH_NT), lds, (hipStream_t)stream, p);
        rc = mh_check("mlp head backward chain");
        if (rc) return rc;
    }
    int blk[MH_MAXL + 1] = {0, 0, 0, 0, 0};
    for (int l = 0; l < L; ++l) blk[l + 1] = blk[l] + (dims[l] + WG_IR - 1) / WG_IR;
    for (int l = L; l < MH_MAXL; ++l) blk[l + 1] = blk[L] + (1 << 28);      // unused layers: never reached
    hipLaunchKernelGGL(mlp_head_wgrad_kernel, dim3((unsigned)blk[L]), dim3(256), 0, (hipStream_t)stream, p, blk[1], blk[2], blk[3]);
    return mh_check("mlp head weight gradients");
}

int fov_gmm3d_loss_grad(const float* params, const float* y, int64_t ldy, float* loss, float* dpre, int B, int n_mix, int n_pts,
                        float scale, int weight_by_pi, void* workspace, size_t workspace_bytes, fov_stream_t stream) {
    if (B == 0) return FOV_OK;
    if (!params || !y || !dpre || B < 0) { set_error("fov_gmm3d_loss_grad: invalid argument"); return FOV_ERR_INVALID; }
    if (n_mix < 1 || n_mix > GM_MAXMIX || n_pts < 1 || n_pts > GM_MAXPTS || ldy < 3 * (int64_t)n_pts) {
        set_error("fov_gmm3d_loss_grad: n_mix <= 32, n_pts <= 256, ldy >= 3 n_pts"); return FOV_ERR_UNSUPPORTED;
    }
    if (!workspace || workspace_bytes < 256 + sizeof(float) * (size_t)B) { set_error("fov_gmm3d_loss_grad: workspace too small"); return FOV_ERR_WORKSPACE; }
    unsigned* ticket = static_cast<unsigned*>(workspace);
    float* part = reinterpret_cast<float*>(static_cast<char*>(workspace) + 256);
    int tc = 6144 / (10 * n_mix);                      // frames per chunk of the gradient sums: at most 48 KB of terms
    if (tc > n_pts) tc = n_pts;
    const size_t lds = sizeof(double) * ((size_t)n_mix * n_pts + (size_t)n_mix * tc * 10);
    int rc = ensure_dynamic_lds((const void*)gmm3d_loss_grad_kernel, lds, 256);
    if (rc) return rc;
    hipLaunchKernelGGL(gmm3d_loss_grad_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, params, y, (long)ldy, ticket, part, loss,
                       dpre, n_mix, n_pts, tc, scale, weight_by_pi ? 1 : 0);
    return mh_check("gmm3d loss");
}

int fov_gmm3d_sample(const float* params, const float* u, const float* z, float* out, int64_t ldo, int B, int n_mix, int n_pts,
                     fov_stream_t stream) {
    if (B == 0) return FOV_OK;
    if (!params || !u || !z || !out || B < 0) { set_error("fov_gmm3d_sample: invalid argument"); return FOV_ERR_INVALID; }
    if (n_mix < 1 || n_mix > GM_MAXMIX || n_pts < 1 || ldo < 3 * (int64_t)n_pts) { set_error("fov_gmm3d_sample: n_mix <= 32, ldo >= 3 n_pts"); return FOV_ERR_UNSUPPORTED; }
    hipLaunchKernelGGL(gmm3d_sample_kernel, dim3((unsigned)B), dim3(256), 0, (hipStream_t)stream, params, u, z, out, (long)ldo, n_mix, n_pts);
    return mh_check("gmm3d sample");
}

}  // extern "C"
