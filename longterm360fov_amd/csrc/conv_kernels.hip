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
    // second input segment (optional): the convolution runs over the channel concatenation [x | x2] without the
    // concatenated map ever existing - a ConvLSTM2D step is conv([x_t | h_{t-1}], [K ; R]) in ONE launch
    const float* x2;    // (B,H,W,*) or NULL
    int C2;             // channels of x2 (w then has C + C2 input channels per tap)
    long ldx2, ldb2;
    const float* x;     // (B,H,W,*) with pixel stride ldx >= C and batch stride ldb >= H*W*ldx
    const float* w;     // (kh*kw*(C + C2), N)
    const float* bias;  // (N) or NULL
    const float* add;   // (B*H*W, N) or NULL (may alias y)
    float* y;           // (B*H*W, N)
    int B, H, W, C, N, kh, kw, act;   // act: 0 none, 2 relu
    int dil;            // dilation of the taps over x (segment 1); segment 2 (the recurrent map of a ConvLSTM2D cell) is never dilated
    long ldx;   // pixel stride
    long ldb;   // batch stride
    // CELL form (ConvLSTM2D step in one launch): N = 4F gate columns, the epilogue applies the gates and the cell update
    const float* c_prev;   // (B*H*W, F) or NULL (zero state)
    float* c_new;          // (B*H*W, F), may alias c_prev
    float* h;              // (B*H*W, F) with pixel stride ldh
    long ldh;
    float* gates;          // (B*H*W, 4F) activated i,f,g,o in Keras column order, or NULL (training tape)
};

typedef unsigned cu32x4 __attribute__((ext_vector_type(4)));

// Implicit GEMM, M = B*H*W pixels, N = output channels, K = (tap, channel).  The k-tiles (16 wide) never
// straddle a filter tap: tile <-> (tap, c0), so the tap's spatial shift is wave-uniform and goes into the
// base address of a per-tile buffer descriptor (SALU); each thread keeps the byte offset of its pixels,
// fixed for the whole kernel, and only decides per tile whether the shifted pixel is inside the image -
// outside ('same' zero padding), past the last channel, or past the last pixel it presents an out-of-range
// offset and the hardware returns 0.  fp32 MFMA shares the VALU issue slots, so this matters: no select on
// loaded data, no 64-bit address arithmetic per element.
//   AVEC: C % 4 == 0 and 16-byte aligned pixels -> one 16-byte load per (pixel, 4 channels)
//   BVEC: N % 4 == 0 -> 16-byte loads of the (K,N) weight rows
// LDS: As[m][k] (k contiguous, stride 24 floats: conflict-free ds_read_b128), Bs[k][n].  With the MFMA k-slot
// of lane group lq in step s mapped to k = 4*lq + s (for A and B alike) one ds_read_b128 gives a lane its A
// operands of all four steps.
// Staging modes are template parameters and the tile loop has no branch between loads, MFMAs and LDS writes
// (the final iteration re-stages its own tile, unused): staged registers never cross a control-flow merge, so
// the compiler waits for the loads where they are written to LDS, after the MFMAs.
//   CELL (0 none, else 1 + FOV_ACT_* of the recurrent activation): the block's BN GEMM columns are the FOUR gates of
// BN/4 units, permuted so that the gates of a unit meet in one lane - z never goes to memory and the separate gates
// launch (a read and a write of the (pixels, 4F) map) is gone.  The permutation costs nothing: it only changes which
// 16-byte piece of a weight row a staging lane loads.
//   NI == 4: GEMM column wn + 16 j + li of a wave is Keras column j*F + unit, unit = n0/4 + wn/4 + li: accumulator tiles
//            j = 0..3 of a lane are i, f, g, o of ONE (pixel, unit).
//   NI == 2 (F <= 8, one 32-column tile pair): column 16 j + li is gate 2 j + (li >> 3) of unit n0/4 + (li & 7): a lane
//            holds (i, g) or (f, o) and gets the other pair from the lane 8 places away in its row (DPP row_ror:8).
template <int MI, int NI, int WAVES_M, int AVEC, int BVEC, int CELL = 0>
__global__ __launch_bounds__(256) void conv2d_igemm_kernel(ConvArgs g) {
    static_assert(CELL == 0 || NI == 4 || (NI == 2 && WAVES_M == 4), "cell epilogue: four column tiles per lane, or the 32-column form");
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = 16 * MI * WAVES_M, BN = 16 * NI * WAVES_N, BK = 16;
    // B fragments (plain convolution with four column tiles per wave): column tile j, lane li stands for GEMM column
    // wn + 4*li + j, so one ds_read_b128 at [k][wn + 4*li] serves all column tiles of a k-step and a lane's outputs are four
    // consecutive channels (the row permutation of gemm_f32_kernel, train_kernels.hip).  The CELL forms keep
    // the 16*j + li map their gate permutation is written for.
    constexpr bool BPERM = CELL == 0 && NI == 4;   // (NI == 2 measured slower with the pair form: 54 -> 46 TFLOP/s at N = 32)
    constexpr int LDA = 24, LDB = BPERM ? BN : BN + 4;
    constexpr int RA = AVEC ? BM / 64 : BM / 16;                 // loads per thread per tile (A)
    constexpr int RB = BVEC ? (BN * 4 + 255) / 256 : BN / 16;    // loads per thread per tile (B)
    constexpr unsigned OOR = 0x80000000u;
    __shared__ __attribute__((aligned(16))) float As[2][BM][LDA];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK][LDB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long M = (long)g.B * g.H * g.W;
    const long m0 = (long)blockIdx.y * BM;
    const int n0 = blockIdx.x * BN;
    const int wm = (wave / WAVES_N) * 16 * MI, wn = (wave % WAVES_N) * 16 * NI;
    const int li = lane & 15, lq = lane >> 4;
    const int ph = (g.kh - 1) / 2, pw = (g.kw - 1) / 2;
    const int ctiles1 = (g.C + BK - 1) / BK;
    const int ctiles = ctiles1 + (g.C2 + BK - 1) / BK;   // k-tiles per tap: segment 1, then segment 2
    const int Ctot = g.C + g.C2;
    const int ntiles = g.kh * g.kw * ctiles;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // A staging map: AVEC group e = tid + 256 r -> (pixel e/4, channel quad e%4); scalar element e -> (pixel
    // e/16, channel e%16).  Consecutive lanes walk the channels of a pixel, then the next pixel.
    int a_mm[RA], a_kc[RA], a_y[RA], a_x[RA];
    unsigned a_pix[RA], a_pix2[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int e = tid + 256 * r;
        a_mm[r] = AVEC ? (e >> 2) : (e >> 4);
        a_kc[r] = AVEC ? 4 * (e & 3) : (e & 15);
        const long m = m0 + a_mm[r];
        if (m < M) {
            const int b = (int)(m / ((long)g.H * g.W));
            const int rem = (int)(m - (long)b * g.H * g.W);
            a_y[r] = rem / g.W;
            a_x[r] = rem - a_y[r] * g.W;
            a_pix[r] = (unsigned)(((long)b * g.ldb + (long)rem * g.ldx + a_kc[r]) * 4);
            a_pix2[r] = (unsigned)(((long)b * g.ldb2 + (long)rem * g.ldx2 + a_kc[r]) * 4);
        } else {
            a_y[r] = -(1 << 20);   // never inside the image
            a_x[r] = 0;
            a_pix[r] = OOR;
            a_pix2[r] = OOR;
        }
    }
    // B staging map: (k row, n) of the (16 x BN) weight tile; byte offset from the tile's first row
    int b_kk[RB], b_nn[RB];
    unsigned b_off[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int e = tid + 256 * r;
        b_kk[r] = BVEC ? e / (BN / 4) : e / BN;
        b_nn[r] = BVEC ? 4 * (e - b_kk[r] * (BN / 4)) : e - b_kk[r] * BN;
        if constexpr (CELL != 0) {
            const int F = g.N >> 2;
            const int gate = NI == 4 ? (b_nn[r] & 63) >> 4 : 2 * (b_nn[r] >> 4) + ((b_nn[r] >> 3) & 1);
            const int unit = (n0 >> 2) + (NI == 4 ? (b_nn[r] >> 6) * 16 + (b_nn[r] & 15) : (b_nn[r] & 7));
            const bool ok = b_kk[r] < BK && unit < F;
            b_off[r] = ok ? (unsigned)(((long)b_kk[r] * g.N + gate * F + unit) * 4) : OOR;
        } else {
            const bool ok = b_kk[r] < BK && n0 + b_nn[r] < g.N;
            b_off[r] = ok ? (unsigned)(((long)b_kk[r] * g.N + n0 + b_nn[r]) * 4) : OOR;
        }
    }

    int f_dy = 0, f_dx = 0, f_ct = 0;   // the tile the next fetch() loads (wave-uniform)
    int f_left = ntiles;                // tiles not fetched yet
    float ra[AVEC ? 1 : RA], rb[BVEC ? 1 : RB];
    f32x4 va[AVEC ? RA : 1], vb[BVEC ? RB : 1];
    // Whether a pixel's shifted source lies inside the image depends on the TAP only: evaluated when the tap changes (a
    // wave-uniform branch around VALU work, no load inside), not for every 16-channel tile of it - at 512 input channels
    // that is 1 tile in 32 (fp32 MFMA shares the VALU port: the 16 compares / selects per tile came out of the MFMA rate).
    bool a_in[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) a_in[r] = false;
    auto fetch = [&]() {
        const bool seg2 = f_ct >= ctiles1;   // wave-uniform: which input segment this tile reads
        const int c0 = (seg2 ? f_ct - ctiles1 : f_ct) * BK;
        const int dl = seg2 ? 1 : g.dil;     // Keras dilates the INPUT convolution of ConvLSTM2D only (dilation_rate), not the recurrent one
        const int sy = (f_dy - ph) * dl, sx = (f_dx - pw) * dl;
        if (f_ct == 0 || (f_ct == ctiles1 && g.dil != 1)) {
#pragma unroll
            for (int r = 0; r < RA; ++r) {
                const int yy = a_y[r] + sy, xx = a_x[r] + sx;
                a_in[r] = yy >= 0 && yy < g.H && xx >= 0 && xx < g.W;
            }
        }
        const float* xt = (seg2 ? g.x2 + ((long)sy * g.W + sx) * g.ldx2 : g.x + ((long)sy * g.W + sx) * g.ldx) + c0;
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(xt), 0, 0x7fffffff, 0x00020000);
        const int crem = (seg2 ? g.C2 : g.C) - c0;   // channels left in this segment of the tap
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const bool ok = a_in[r] && a_kc[r] < crem;
            const unsigned off = ok ? (seg2 ? a_pix2[r] : a_pix[r]) : OOR;
            if constexpr (AVEC) {
                const cu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(xrs, off, 0, 0);
                va[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
            } else {
                ra[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
            }
        }
        const int tap = f_dy * g.kw + f_dx;
        const float* wt = g.w + ((long)tap * Ctot + (seg2 ? g.C : 0) + c0) * g.N;
        const int krows = crem < BK ? crem : BK;
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(wt), 0, krows * g.N * 4, 0x00020000);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            if constexpr (BVEC) {
                const cu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(wrs, b_off[r], 0, 0);
                vb[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
            } else {
                rb[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, b_off[r], 0, 0));
            }
        }
        // advance (scalar state only); after the last tile the position stays, so the extra fetch of the final
        // iteration re-reads that tile instead of branching around the loads
        if (--f_left > 0 && ++f_ct == ctiles) {
            f_ct = 0;
            if (++f_dx == g.kw) { f_dx = 0; ++f_dy; }
        }
    };
    auto stash = [&](int buf) {
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            if constexpr (AVEC) *(f32x4*)&As[buf][a_mm[r]][a_kc[r]] = va[r];
            else As[buf][a_mm[r]][a_kc[r]] = ra[r];
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
    f32x4 af[MI];
    float bv[4][NI];
    auto read_frags = [&](int buf) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[i] = *(const f32x4*)&As[buf][wm + i * 16 + li][4 * lq];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if constexpr (BPERM && NI == 4) {
                const f32x4 v = *(const f32x4*)&Bs[buf][4 * lq + s][wn + 4 * li];
                bv[s][0] = v[0]; bv[s][1] = v[1]; bv[s][2] = v[2]; bv[s][3] = v[3];
            } else {
#pragma unroll
                for (int j = 0; j < NI; ++j) bv[s][j] = Bs[buf][4 * lq + s][wn + j * 16 + li];
            }
        }
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bv[s][j], acc[i][j], 0, 0, 0);
    };
    int buf = 0;
    fetch();
    stash(0);
    __syncthreads();
    // no branch separates loads, MFMAs and LDS writes: the final iteration stages its own tile once more, unused
    for (int t = 0; t < ntiles; ++t) {
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
    if constexpr (CELL != 0) {
        const int F = g.N >> 2;
        const int half = NI == 4 ? 0 : li >> 3;       // NI == 2: which gate pair this lane accumulated
        const int unit = (n0 >> 2) + (NI == 4 ? (wn >> 2) + li : (li & 7));
        const bool mine = unit < F && half == 0;
        float bz[4] = {0.f, 0.f, 0.f, 0.f};           // NI == 2: bz[0], bz[1] = bias of the lane's own two gates
        if (g.bias && unit < F) {
#pragma unroll
            for (int j = 0; j < NI; ++j) bz[j] = g.bias[(NI == 4 ? j : 2 * j + half) * F + unit];
        }
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const long m = m0 + wm + i * 16 + lq * 4 + r;
                float zi, zf, zg, zo;
                if constexpr (NI == 4) {
                    zi = acc[i][0][r] + bz[0]; zf = acc[i][1][r] + bz[1]; zg = acc[i][2][r] + bz[2]; zo = acc[i][3][r] + bz[3];
                } else {
                    const float z0 = acc[i][0][r] + bz[0], z1 = acc[i][1][r] + bz[1];   // (i, g) in lanes 0-7, (f, o) in lanes 8-15
                    const float p0 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(z0), 0x128, 0xf, 0xf, false));
                    const float p1 = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(z1), 0x128, 0xf, 0xf, false));
                    zi = z0; zf = p0; zg = z1; zo = p1;    // right for half == 0, the lanes that store
                }
                if (mine && m < M) {
                    const float gi = rec_act<CELL - 1>(zi), gf = rec_act<CELL - 1>(zf), gg = tanh_f(zg), go = rec_act<CELL - 1>(zo);
                    const float cn = fmaf(gf, g.c_prev ? g.c_prev[m * F + unit] : 0.f, gi * gg);
                    g.c_new[m * F + unit] = cn;
                    g.h[m * g.ldh + unit] = go * tanh_f(cn);
                    if (g.gates) {
                        float* gp = g.gates + m * g.N + unit;
                        gp[0] = gi; gp[F] = gf; gp[2 * F] = gg; gp[3 * F] = go;
                    }
                }
            }
        return;
    }
    const bool vec_out = NI == 4 && (g.N & 3) == 0 && (((uintptr_t)g.y) & 15) == 0 && (!g.add || (((uintptr_t)g.add) & 15) == 0) &&
                         (!g.bias || (((uintptr_t)g.bias) & 15) == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long m = m0 + wm + i * 16 + lq * 4 + r;
            if (m >= M) continue;
            if constexpr (NI == 4) {
                const int nb = n0 + wn + 4 * li;
                if (vec_out && nb + 3 < g.N) {   // the lane's four column tiles are four consecutive channels: 16-byte traffic
                    f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if (g.bias) v += *(const f32x4*)(g.bias + nb);
                    if (g.add) v += *(const f32x4*)(g.add + m * g.N + nb);
                    if (g.act == 2) { v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f); }
                    *(f32x4*)(g.y + m * g.N + nb) = v;
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn + (BPERM ? 4 * li + j : j * 16 + li);
                if (n < g.N) {
                    float v = acc[i][j][r];
                    if (g.bias) v += g.bias[n];
                    if (g.add) v += g.add[m * g.N + n];
                    if (g.act == 2) v = fmaxf(v, 0.f);
                    g.y[m * g.N + n] = v;
                }
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
               int W, int C, int N, int kh, int kw, int act, hipStream_t stream, int dil) {
    return conv2d_fwd2(x, ldx, ldb, C, nullptr, 0, 0, 0, w, bias, add, y, B, H, W, N, kh, kw, act, stream, dil);
}

int conv2d_fwd2(const float* x, long ldx, long ldb, int C, const float* x2, long ldx2, long ldb2, int C2, const float* w,
                const float* bias, const float* add, float* y, int B, int H, int W, int N, int kh, int kw, int act,
                hipStream_t stream, int dil) {
    ConvArgs g = {};
    g.dil = dil < 1 ? 1 : dil;
    g.x = x; g.w = w; g.bias = bias; g.add = add; g.y = y;
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N; g.kh = kh; g.kw = kw; g.act = act; g.ldx = ldx; g.ldb = ldb;
    g.x2 = x2; g.C2 = x2 ? C2 : 0; g.ldx2 = ldx2; g.ldb2 = ldb2;
    const long M = (long)B * H * W;
    if (M == 0 || N == 0) return FOV_OK;
    // 31-bit byte offsets inside one buffer descriptor
    if ((long)B * ldb * 4 >= (1L << 31) || (x2 && (long)B * ldb2 * 4 >= (1L << 31)) ||
        (long)kh * kw * (C + g.C2) * N * 4 >= (1L << 31)) {
        set_error("conv2d: operand larger than 2 GiB");
        return FOV_ERR_UNSUPPORTED;
    }
    // one input segment, a map small enough to sit in LDS, enough channels to be worth it: the map-resident form (conv_patch.hip)
    if (!x2 && g.dil == 1 && conv_patch_shape_ok(x, ldx, ldb, B, H, W, C, N, kh, kw))
        return launch_conv_patch(x, ldx, ldb, w, bias, add, y, B, H, W, C, N, kh, kw, act, stream);
    const bool avec = (C & 3) == 0 && (ldx & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
                      (!x2 || ((C2 & 3) == 0 && (ldx2 & 3) == 0 && (ldb2 & 3) == 0 && (((uintptr_t)x2) & 15) == 0));
    const bool bvec = (N & 3) == 0 && (((uintptr_t)w) & 15) == 0;
#define FOV_CONV_LAUNCH(MI_, NI_, WM_, grid_)                                                                          \
    do {                                                                                                               \
        if (avec && bvec) hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 1, 1>), grid_, dim3(256), 0, stream, g); \
        else if (avec) hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 1, 0>), grid_, dim3(256), 0, stream, g);  \
        else if (bvec) hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 0, 1>), grid_, dim3(256), 0, stream, g);  \
        else hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 0, 0>), grid_, dim3(256), 0, stream, g);            \
    } while (0)
    if (N <= 32) {
        const dim3 grid((N + 31) / 32, (unsigned)((M + 255) / 256));
        FOV_CONV_LAUNCH(4, 2, 4, grid);
    } else if (N <= 64) {
        const dim3 grid((N + 63) / 64, (unsigned)((M + 255) / 256));
        FOV_CONV_LAUNCH(4, 4, 4, grid);
    } else {
        const dim3 grid((N + 127) / 128, (unsigned)((M + 127) / 128));
        FOV_CONV_LAUNCH(4, 4, 2, grid);
    }
#undef FOV_CONV_LAUNCH
    return conv_check_launch("conv2d_igemm");
}

// One ConvLSTM2D step: h, c <- cell(conv([x | h_prev], [K ; R]) + b, c_prev).  h_prev may be NULL (zero state: w then holds
// K alone).  h must not alias h_prev (neighbouring pixels read it); c_new may alias c_prev.
int convlstm_cell_fwd(const float* x, long ldx, long ldb, int C, const float* h_prev, long ldx2, long ldb2, const float* w,
                      const float* bias, const float* c_prev, float* c_new, float* h, long ldh, float* gates, int B, int H, int W,
                      int F, int kh, int kw, int act, hipStream_t stream, int dil) {
    ConvArgs g = {};
    g.dil = dil < 1 ? 1 : dil;
    g.x = x; g.w = w; g.bias = bias; g.B = B; g.H = H; g.W = W; g.C = C; g.N = 4 * F; g.kh = kh; g.kw = kw; g.ldx = ldx; g.ldb = ldb;
    g.x2 = h_prev; g.C2 = h_prev ? F : 0; g.ldx2 = ldx2; g.ldb2 = ldb2;
    g.c_prev = c_prev; g.c_new = c_new; g.h = h; g.ldh = ldh; g.gates = gates;
    const long M = (long)B * H * W;
    if (M == 0 || F == 0) return FOV_OK;
    if ((long)B * ldb * 4 >= (1L << 31) || (h_prev && (long)B * ldb2 * 4 >= (1L << 31)) ||
        (long)kh * kw * (C + g.C2) * 4 * F * 4 >= (1L << 31)) {
        set_error("convlstm_cell: operand larger than 2 GiB");
        return FOV_ERR_UNSUPPORTED;
    }
    // the LDS-resident-patch form (convlstm_patch.hip): every tap reads the same staged patch, no barrier in the k loop
    if (g.dil == 1 && cell_patch_shape_ok(x, ldx, ldb, C, h_prev, ldx2, ldb2, F, H, W, kh, kw))
        return launch_cell_patch(x, ldx, ldb, C, h_prev, ldx2, ldb2, w, bias, c_prev, c_new, h, ldh, gates, B, H, W, F, kh, kw, act, stream);
    const bool avec = (C & 3) == 0 && (ldx & 3) == 0 && (ldb & 3) == 0 && (((uintptr_t)x) & 15) == 0 &&
                      (!h_prev || ((F & 3) == 0 && (ldx2 & 3) == 0 && (ldb2 & 3) == 0 && (((uintptr_t)h_prev) & 15) == 0));
    const bool bvec = (F & 3) == 0 && (((uintptr_t)w) & 15) == 0;
    // tile shapes as conv2d_fwd2 picks them for N = 4F: 8 / 16 / 32 units per block
#define FOV_CELL_LAUNCH(MI_, NI_, WM_, grid_, ACT_)                                                                              \
    do {                                                                                                                         \
        if (avec && bvec) hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 1, 1, ACT_ + 1>), grid_, dim3(256), 0, stream, g); \
        else if (avec) hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 1, 0, ACT_ + 1>), grid_, dim3(256), 0, stream, g);  \
        else if (bvec) hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 0, 1, ACT_ + 1>), grid_, dim3(256), 0, stream, g);  \
        else hipLaunchKernelGGL((conv2d_igemm_kernel<MI_, NI_, WM_, 0, 0, ACT_ + 1>), grid_, dim3(256), 0, stream, g);            \
    } while (0)
#define FOV_CELL_SHAPES(ACT_)                                                                  \
    do {                                                                                       \
        if (F <= 8) {                                                                          \
            const dim3 grid(1, (unsigned)((M + 255) / 256));                                   \
            FOV_CELL_LAUNCH(4, 2, 4, grid, ACT_);                                              \
        } else if (F <= 16) {                                                                  \
            const dim3 grid(1, (unsigned)((M + 255) / 256));                                   \
            FOV_CELL_LAUNCH(4, 4, 4, grid, ACT_);                                              \
        } else {                                                                               \
            const dim3 grid((F + 31) / 32, (unsigned)((M + 127) / 128));                       \
            FOV_CELL_LAUNCH(4, 4, 2, grid, ACT_);                                              \
        }                                                                                      \
    } while (0)
    if (act == FOV_ACT_HARD_SIGMOID) FOV_CELL_SHAPES(FOV_ACT_HARD_SIGMOID);
    else FOV_CELL_SHAPES(FOV_ACT_SIGMOID);
#undef FOV_CELL_SHAPES
#undef FOV_CELL_LAUNCH
    return conv_check_launch("convlstm_cell");
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
