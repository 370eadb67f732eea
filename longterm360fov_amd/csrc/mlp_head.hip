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

namespace {

constexpr int MH_NT = 512;        // threads per workgroup (forward and backward chain)
constexpr int MH_ROWS = 4;        // rows a workgroup carries through the chain
constexpr int MH_MAXL = 4;        // layers at most
constexpr int MH_MAXD = 512;      // width of any layer's OUTPUT at most
constexpr int MH_MAXD0 = 2048;    // width of the input x at most (lstm.py's n_hidden, padded or not)
constexpr int MH_TILE = 26 * 1024;   // floats of one transposed weight tile in the backward chain (104 KB of LDS)
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

// Forward.  Layer l: thread (col, ks) = (tid % cp, tid / cp), cp = the layer's width rounded up to 64 (at most the block), so a
// wave holds one k-slice and 64 neighbouring columns: W reads are coalesced over col, the rows' inputs come from LDS as a
// broadcast, the k-slices meet in LDS in a fixed order.
__global__ __launch_bounds__(MH_NT) void mlp_head_fwd_kernel(MlpParams p) {
    __shared__ __attribute__((aligned(16))) float xin[MH_ROWS * MH_MAXD0];
    __shared__ __attribute__((aligned(16))) float bufA[MH_ROWS * MH_MAXD];
    __shared__ __attribute__((aligned(16))) float bufB[MH_ROWS * MH_MAXD];
    __shared__ float red[MH_ROWS * MH_NT];
    const int tid = threadIdx.x, row0 = blockIdx.x * MH_ROWS;
    const float* in = xin;
    int istr = MH_MAXD0;              // row stride of `in`
    float* out = bufA;
    {
        const int D0 = p.D[0];
        for (int e = tid; e < MH_ROWS * D0; e += MH_NT) {
            const int r = e / D0, k = e - r * D0;
            xin[r * MH_MAXD0 + k] = (row0 + r < p.B) ? p.x[(size_t)(row0 + r) * D0 + k] : 0.f;
        }
    }
    __syncthreads();
    for (int l = 0; l < p.L; ++l) {
        const int Din = p.D[l], Dout = p.D[l + 1];
        const float* __restrict__ W = p.W[l];
        int cp = (Dout + 63) & ~63;
        if (cp > MH_NT) cp = MH_NT;
        const int KS = MH_NT / cp;                      // k-slices (threads beyond KS * cp idle in this layer)
        const int c = tid % cp, ks = tid / cp;
        const int kc = (((Din + KS - 1) / KS) + 3) & ~3;
        const int k0 = ks * kc, k1 = (k0 + kc < Din) ? k0 + kc : Din;
        const bool last = (l == p.L - 1);
        for (int col0 = 0; col0 < Dout; col0 += cp) {
            const int col = col0 + c;
            float acc[MH_ROWS] = {0.f, 0.f, 0.f, 0.f};
            if (ks < KS && col < Dout) {
                int k = k0;
                for (; k + 8 <= k1; k += 8) {
                    float w[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) w[u] = W[(size_t)(k + u) * Dout + col];
#pragma unroll
                    for (int r = 0; r < MH_ROWS; ++r) {
                        const f32x4 x0 = *reinterpret_cast<const f32x4*>(in + r * istr + k);
                        const f32x4 x1 = *reinterpret_cast<const f32x4*>(in + r * istr + k + 4);
                        acc[r] = fmaf(x0[0], w[0], acc[r]); acc[r] = fmaf(x0[1], w[1], acc[r]);
                        acc[r] = fmaf(x0[2], w[2], acc[r]); acc[r] = fmaf(x0[3], w[3], acc[r]);
                        acc[r] = fmaf(x1[0], w[4], acc[r]); acc[r] = fmaf(x1[1], w[5], acc[r]);
                        acc[r] = fmaf(x1[2], w[6], acc[r]); acc[r] = fmaf(x1[3], w[7], acc[r]);
                    }
                }
                for (; k < k1; ++k) {
                    const float w = W[(size_t)k * Dout + col];
#pragma unroll
                    for (int r = 0; r < MH_ROWS; ++r) acc[r] = fmaf(in[r * istr + k], w, acc[r]);
                }
            }
            if (ks < KS)
#pragma unroll
                for (int r = 0; r < MH_ROWS; ++r) red[(ks * MH_ROWS + r) * cp + c] = acc[r];
            __syncthreads();
            for (int e = tid; e < MH_ROWS * cp; e += MH_NT) {
                const int r = e / cp, cc = e - r * cp, oc = col0 + cc, row = row0 + r;
                if (oc < Dout) {
                    float v = p.b[l][oc];
                    for (int q = 0; q < KS; ++q) v += red[(q * MH_ROWS + r) * cp + cc];      // fixed order
                    if (!(last && p.final_mode)) v = mh_act(v, p.act[l]);
                    if (p.mask[l] && row < p.B) v *= p.mask[l][(size_t)row * Dout + oc];
                    out[r * MH_MAXD + oc] = v;
                    if (row < p.B && !(last && p.final_mode)) p.a[l][(size_t)row * Dout + oc] = v;
                }
            }
            __syncthreads();
        }
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
}

// Backward chain for four rows: d_{l-1}[r][i] = act'(a_{l-1}[r][i]) mask_{l-1}[r][i] sum_j d_l[r][j] W_l[i][j].  W_l is read in
// tiles of whole rows (a contiguous, coalesced block of global memory) into LDS with an odd row stride; thread (i, js) then
// walks its j-slice of row i: lanes differ in i, so their LDS addresses differ by an odd stride - no bank conflict.
__global__ __launch_bounds__(MH_NT) void mlp_head_bwd_chain_kernel(MlpParams p) {
    extern __shared__ __attribute__((aligned(16))) float mh_lds[];
    float* wt = mh_lds;                                   // [MH_TILE]
    float* bufA = wt + MH_TILE;                           // [MH_ROWS][MH_MAXD]
    float* bufB = bufA + MH_ROWS * MH_MAXD;
    float* red = bufB + MH_ROWS * MH_MAXD;                // [MH_ROWS][MH_NT]
    const int tid = threadIdx.x, row0 = blockIdx.x * MH_ROWS;
    float* dcur = bufA;
    float* dnew = bufB;
    {
        const int Dl = p.D[p.L];
        for (int e = tid; e < MH_ROWS * Dl; e += MH_NT) {
            const int r = e / Dl, j = e - r * Dl;
            dcur[r * MH_MAXD + j] = (row0 + r < p.B) ? p.dlast[(size_t)(row0 + r) * Dl + j] : 0.f;
        }
    }
    __syncthreads();
    for (int l = p.L - 1; l >= 0; --l) {
        if (l == 0 && !p.dx) break;
        const int Din = p.D[l], Dout = p.D[l + 1];
        const float* __restrict__ W = p.W[l];
        const int stride = Dout | 1;
        int IC = MH_NT;                                   // rows of W per tile: the largest of 512, 256, ..., 32 that fits the
        while (IC > 32 && (IC * stride > MH_TILE || IC >= 2 * Din)) IC >>= 1;      // tile and is not twice the layer
        const int JS = MH_NT / IC;                        // j-slices
        const int il = tid % IC, js = tid / IC;
        const int jc = (Dout + JS - 1) / JS;
        const int j0 = js * jc, j1 = (j0 + jc < Dout) ? j0 + jc : Dout;
        for (int i0 = 0; i0 < Din; i0 += IC) {
            const int rows = (Din - i0 < IC) ? Din - i0 : IC;
            const float* __restrict__ src = W + (size_t)i0 * Dout;
            for (int e = tid; e < rows * Dout; e += MH_NT) {
                const int i = e / Dout, j = e - i * Dout;
                wt[i * stride + j] = src[e];
            }
            __syncthreads();
            float acc[MH_ROWS] = {0.f, 0.f, 0.f, 0.f};
            if (js < JS && il < rows) {
                const float* wrow = wt + il * stride;
                for (int j = j0; j < j1; ++j) {
                    const float w = wrow[j];
#pragma unroll
                    for (int r = 0; r < MH_ROWS; ++r) acc[r] = fmaf(dcur[r * MH_MAXD + j], w, acc[r]);
                }
            }
            if (js < JS)
#pragma unroll
                for (int r = 0; r < MH_ROWS; ++r) red[(js * MH_ROWS + r) * IC + il] = acc[r];
            __syncthreads();
            for (int e = tid; e < MH_ROWS * rows; e += MH_NT) {
                const int r = e / rows, ii = e - r * rows, i = i0 + ii, row = row0 + r;
                float v = 0.f;
                for (int q = 0; q < JS; ++q) v += red[(q * MH_ROWS + r) * IC + ii];
                if (l > 0) {
                    if (row < p.B) {
                        v *= mh_dact(p.a[l - 1][(size_t)row * Din + i], p.act[l - 1]);
                        if (p.mask[l - 1]) v *= p.mask[l - 1][(size_t)row * Din + i];
                        p.d[l - 1][(size_t)row * Din + i] = v;
                    } else v = 0.f;
                    dnew[r * MH_MAXD + i] = v;
                } else if (row < p.B) {
                    p.dx[(size_t)row * Din + i] = v;
                }
            }
            __syncthreads();
        }
        float* t = dcur; dcur = dnew; dnew = t;
    }
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
                for (; n + 4 <= nn; n += 4) {
                    float dv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) dv[q] = dl[(size_t)(n0 + n + q) * Dout + j];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
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
    double lam;
    sym3_min_eig(S, lam, o.v);
    o.repaired = lam < 0.0;
    if (o.repaired) { const double k = -10.0 * lam; S.a00 += k; S.a11 += k; S.a22 += k; }
    const double c00 = S.a11 * S.a22 - S.a12 * S.a12, c01 = S.a02 * S.a12 - S.a01 * S.a22, c02 = S.a01 * S.a12 - S.a02 * S.a11;
    const double det = S.a00 * c00 + S.a01 * c01 + S.a02 * c02;
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
__global__ __launch_bounds__(256) void gmm3d_loss_grad_kernel(const float* __restrict__ params, const float* __restrict__ y, long ldy,
                                                              float* __restrict__ part, float* __restrict__ dpre, int n, int npts,
                                                              float scale, int weight_by_pi) {
    extern __shared__ __attribute__((aligned(16))) double gm_lds[];
    __shared__ MixSetup ms[GM_MAXMIX];
    __shared__ double wt[GM_MAXPTS];          // w_t = -scale / (S_t + eps)
    __shared__ double dpi[GM_MAXMIX];
    __shared__ double lsum[256];
    double* pm = gm_lds;                      // [n][npts] densities (times pi when weight_by_pi)
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* prm = params + (size_t)b * 10 * n;
    const float* yb = y + (size_t)b * ldy;
    if (tid < n) mix_setup(prm, n, tid, ms[tid]);
    __syncthreads();
    for (int e = tid; e < n * npts; e += 256) {
        const int m = e / npts, t = e - m * npts;
        const MixSetup& s = ms[m];
        const double d0 = (double)yb[3 * t] - s.mu[0], d1 = (double)yb[3 * t + 1] - s.mu[1], d2 = (double)yb[3 * t + 2] - s.mu[2];
        const double q = s.P[0] * d0 * d0 + s.P[3] * d1 * d1 + s.P[5] * d2 * d2 + 2.0 * (s.P[1] * d0 * d1 + s.P[2] * d0 * d2 + s.P[4] * d1 * d2);
        pm[e] = exp(s.lognorm - 0.5 * q);
    }
    __syncthreads();
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
    if (tid == 0) part[b] = (float)lsum[0];
    float* dp = dpre + (size_t)b * 10 * n;
    if (tid < n) {
        const int m = tid;
        const MixSetup& s = ms[m];
        const double pim = weight_by_pi ? (double)prm[m] : 1.0;
        double gmu[3] = {0, 0, 0}, G[6] = {0, 0, 0, 0, 0, 0}, gp = 0.0;
        for (int t = 0; t < npts; ++t) {
            const double p = pm[m * npts + t], wp = wt[t] * pim * p;      // dL/dp_mt * p_mt
            const double d0 = (double)yb[3 * t] - s.mu[0], d1 = (double)yb[3 * t + 1] - s.mu[1], d2 = (double)yb[3 * t + 2] - s.mu[2];
            const double a0 = s.P[0] * d0 + s.P[1] * d1 + s.P[2] * d2;
            const double a1 = s.P[1] * d0 + s.P[3] * d1 + s.P[4] * d2;
            const double a2 = s.P[2] * d0 + s.P[4] * d1 + s.P[5] * d2;
            gmu[0] += wp * a0; gmu[1] += wp * a1; gmu[2] += wp * a2;
            const double h = 0.5 * wp;
            G[0] += h * (a0 * a0 - s.P[0]); G[1] += h * (a0 * a1 - s.P[1]); G[2] += h * (a0 * a2 - s.P[2]);
            G[3] += h * (a1 * a1 - s.P[3]); G[4] += h * (a1 * a2 - s.P[4]); G[5] += h * (a2 * a2 - s.P[5]);
            gp += wt[t] * p;
        }
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
        dp[n + 3 * m] = (float)gmu[0]; dp[n + 3 * m + 1] = (float)gmu[1]; dp[n + 3 * m + 2] = (float)gmu[2];
        dp[4 * n + 3 * m] = (float)(ds1 * s1); dp[4 * n + 3 * m + 1] = (float)(ds2 * s2); dp[4 * n + 3 * m + 2] = (float)(ds3 * s3);
        dp[7 * n + 3 * m] = (float)(2.0 * G[1] * s1 * s2 * (1.0 - r12 * r12));
        dp[7 * n + 3 * m + 1] = (float)(2.0 * G[2] * s1 * s3 * (1.0 - r13 * r13));
        dp[7 * n + 3 * m + 2] = (float)(2.0 * G[4] * s2 * s3 * (1.0 - r23 * r23));
        dpi[m] = gp;          // d loss / d pi_m (used only when weight_by_pi)
    }
    __syncthreads();
    if (tid < n) {
        float g = 0.f;
        if (weight_by_pi) {   // softmax backward
            double dot = 0.0;
            for (int q = 0; q < n; ++q) dot += (double)prm[q] * dpi[q];
            g = (float)((double)prm[tid] * (dpi[tid] - dot));
        }
        dp[tid] = g;
    }
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

__global__ __launch_bounds__(256) void mh_sum_scale_kernel(const float* __restrict__ part, float* __restrict__ out, int n, float scale) {
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) a += (double)part[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if ((int)threadIdx.x < s_) red[threadIdx.x] += red[threadIdx.x + s_];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)(red[0] * (double)scale);
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
    hipLaunchKernelGGL(mlp_head_fwd_kernel, dim3((unsigned)((B + MH_ROWS - 1) / MH_ROWS)), dim3(MH_NT), 0, (hipStream_t)stream, p);
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
        const size_t lds = sizeof(float) * ((size_t)MH_TILE + 2 * MH_ROWS * MH_MAXD + MH_ROWS * MH_NT);
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
    if (!workspace || workspace_bytes < sizeof(float) * (size_t)B) { set_error("fov_gmm3d_loss_grad: workspace too small"); return FOV_ERR_WORKSPACE; }
    float* part = static_cast<float*>(workspace);
    const size_t lds = sizeof(double) * (size_t)n_mix * n_pts;
    int rc = ensure_dynamic_lds((const void*)gmm3d_loss_grad_kernel, lds, 256);
    if (rc) return rc;
    hipLaunchKernelGGL(gmm3d_loss_grad_kernel, dim3((unsigned)B), dim3(256), lds, (hipStream_t)stream, params, y, (long)ldy, part, dpre,
                       n_mix, n_pts, scale, weight_by_pi ? 1 : 0);
    rc = mh_check("gmm3d loss");
    if (rc || !loss) return rc;
    hipLaunchKernelGGL(mh_sum_scale_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, part, loss, B, scale);
    return mh_check("gmm3d loss sum");
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
