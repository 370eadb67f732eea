// Persistent LSTM layer with a WIDE input (96 < F <= 256), H = 256: the stacked layers of the others-mixing
// model (mycode/given_others_gt_mean_var_seq2seq.py:111-112: encoder layer 2 runs over the 256-wide sequence of
// layer 1).  lstm_cluster.hip keeps its input kernel in LDS and stops at F = 96; beyond that the callers had to
// form zx = x . K with a GEMM first (fov_matmul + fov_lstm_seq_fwd_zx: an extra launch and a (B,T,4H) round
// trip).  Here both K (F x 4H) and R (H x 4H) stay in registers for the whole sequence:
//   * a tile of 16 sequences is owned by a group of EIGHT workgroups, workgroup `slice` owns hidden units
//     [32*slice, +32), wave w 8 of them (two MFMA N-tiles [i | f], [g | o]; DPP half swap before the cell
//     update - the ownership scheme of mix_decoder.hip);
//   * K slice 128 AGPRs + R slice 128 AGPRs per lane;
//   * the input tile x_t (16 x F) is double-buffered in LDS (global -> registers -> LDS one step ahead), the h
//     tile (16 x 256) sits next to it; x_{t+1} . K (128 MFMAs) runs under the gather of h_t.
// NARROW variant (NJX = 6, F <= 96, any F): the same kernel with only the six k-blocks of K that exist and a scalar x
// stage.  A batch of <= 512 sequences is <= 32 tiles: lstm_cluster.hip's groups of FOUR workgroups would occupy at
// most half of the 256 CUs, groups of eight fill them, so fov_lstm_seq_fwd(impl = auto) sends such batches here
// (encoder layer 1 of configs[2] at 512 sequences per GPU).
// WIDTH 512 (round 3, template parameters HW = 512, GW = 16): the same kernel with SIXTEEN workgroups per tile, 32 units
// each, R slice 512 x 128 = 256 AGPRs per lane - the persistent form for mycode/lstm.py's LSTMCell(400) x 2 (:59,
// 218-240), whose weights the model object zero-pads 400 -> 512 (exact: models.pad_lstm).  Layer 1 (F = 90) is the
// narrow variant; layer 2's input is 512 wide, its K would be another 256 registers, so it takes the ZX form: the input
// projection zx = x . K comes precomputed (fov_matmul) and is added to the bias at the start of every step.
#include <stdlib.h>

#include "fov_common.h"
#include "xch_common.h"
#include "stage_f32.h"

namespace fov {

namespace {

constexpr int WBT = 16;
constexpr unsigned WSPIN = 1u << 20;

typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));

template <bool W_AGPR>
__device__ __forceinline__ void wm_a(f32x4& acc, float a, float w) {
    if constexpr (W_AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w));
}
__device__ __forceinline__ void wm_begin(f32x4 (&acc)[2]) { asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void wm_end(f32x4 (&acc)[2]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}
__device__ __forceinline__ float wswap(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));
}

// acc[tile] += A(tile rows in LDS, k-blocks [0, NJ)) . W (register resident: accumulation registers, or - width 512, whose
// R slice alone fills all 256 of them - the K slice in ordinary vector registers)
template <int NJ, bool W_AGPR = true>
__device__ __forceinline__ void wide_mm(f32x4 (&acc)[2], const float* arow, const float (&w)[NJ][4][2]) {
    f32x4 a = *(const f32x4*)arow;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x4 an = a;
        if (j + 1 < NJ) an = *(const f32x4*)(arow + 16 * (j + 1));
        asm volatile("s_nop 1" : "+v"(a));   // the fragment may have been moved by the compiler (VALU copy)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            wm_a<W_AGPR>(acc[0], a[s], w[j][s][0]);
            wm_a<W_AGPR>(acc[1], a[s], w[j][s][1]);
        }
        a = an;
    }
}

// NJX: k-blocks of K held in registers (16: wide input, 16-byte x pieces; 6: narrow input, scalar x stage; 0: ZX mode, the
// input projection comes precomputed in p.zx).  WH / WG: hidden width and workgroups per tile (256 / 8 or 512 / 16).
template <int ACT, int NJX, int WH, int WG>
__global__ __launch_bounds__(256, 1) void lstm_wide_kernel(LstmParams p) {
    constexpr bool XVEC = NJX == 16;   // wide inputs: 16-byte pieces (F % 4 == 0, x aligned); narrow: scalar elements
    constexpr bool ZXM = NJX == 0;
    constexpr int WLD = WH + 4;        // LDS row stride of the h tile and of the x tiles
    constexpr int WNG = WG - 1;        // 16-byte loads (two adjacent units' tagged granules) per thread and step: (WG-1) slices * 16 rows * 16 pairs / 256
    constexpr int NJR = WH / 16;       // k-blocks of R
    constexpr bool K_AGPR = WH == 256; // width 512: R takes every accumulation register, K (narrow: 48 values) stays in VGPRs
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sH = smem;                    // [16][WLD]
    float* sX = sH + WBT * WLD;          // [2][16][WLD]
    int* sFlag = (int*)(sX + 2 * WBT * WLD);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    // members 8 blocks apart on a grid padded to a multiple of eight groups: one XCD (xch_padded_groups, xch_common.h)
    const int group = (blockIdx.x / (8 * WG)) * 8 + (blockIdx.x & 7), slice = (blockIdx.x >> 3) & (WG - 1);
    if (group >= p.num_groups) { xch_spare_leaves(p.status, p.T > 1); return; }
    const int F = ZXM ? 0 : p.F, steps = p.T;
    const int unit = 32 * slice + 8 * wave + (n & 7);
    const int hi = n >> 3;
    const int col0 = hi * WH + unit, col1 = (2 + hi) * WH + unit;
    constexpr int H4 = 4 * WH;
    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    const bool xch_used = steps > 1;
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_used ? xch_arrive(p.status, sXch, group, slice) : 0u;
    const bool poisoned = xch_used && xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- resident weights through LDS (stage_f32.h; the staging buffers are the h / x tiles, filled after it): K rows >= F
    // are zero (the descriptor ends with row F - 1) ----
    constexpr unsigned OORB = 0x80000000u;   // buffer-load offset no descriptor covers: reads as 0
    float wk[NJX > 0 ? NJX : 1][4][2], wr[NJR][4][2];
    static_assert((3 * WBT * WLD) >= FST_LDS_WORDS, "the h tile and the two x tiles hold the two staging buffers");
    if constexpr (ZXM) stage_weight_set_f32<WH>(wr, p.R, WH, slice, (unsigned*)smem, []() {});
    else stage_weight_sets_f32<WH>(wk, p.K, F, wr, p.R, WH, slice, (unsigned*)smem, []() {});
    const float bv[2] = {p.b[col0], p.b[col1]};
    for (int i = tid; i < 2 * WBT * WLD; i += 256) sX[i] = 0.f;   // columns >= F stay zero

    // ---- exchange bookkeeping (the granule protocol of lstm_cluster.hip, placement-independent form) ----
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 2 * WBT * WH, 0, 2 * WBT * WH * (int)sizeof(unsigned long long), 0x00020000);
    const int my_row0 = 4 * g4 + 2 * hi;
    // granule order [row pair][unit][row of the pair]: the lane's two rows leave as ONE 16-byte store, a gather load brings
    // both rows of a unit (two tagged granules)
    const unsigned pub_off = (unsigned)((my_row0 >> 1) * WH + unit) * 16u;
    // gather: thread (row pair tid / 32, unit tid % 32) brings both rows of its unit of every partner slice with one 16-byte load
    const unsigned gvoff = (unsigned)((tid >> 5) * WH + (tid & 31)) * 16u;
    const int lbase = 2 * (tid >> 5) * WLD + (tid & 31);
    constexpr unsigned PARITY = WBT * WH * 8u;
    if (xch_used) xch_hello_poll(p.status, sXch, group, WG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    XchTicket ticket = {0u, 0u, 0u};
    if (xch_used) ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (xch_used && tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    wu32x4 v[WNG];
    auto gather_issue = [&](unsigned base) {
#pragma unroll
        for (int j = 0; j < WNG; ++j) {
            const unsigned uo = (unsigned)(((slice + 1 + j) & (WG - 1)) * 32) * 16u;
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, gvoff, base + uo, 16);
        }
    };
    // first pass: current granules go straight to the h tile, stale ones into a bit mask; retry sweeps (rare)
    // re-read into loop-local temporaries
    auto gather_finish = [&](unsigned base) {
        unsigned bad = 0;
#pragma unroll
        for (int j = 0; j < WNG; ++j) {
            const int lo = lbase + ((slice + 1 + j) & (WG - 1)) * 32;
            if (v[j].y == epoch && v[j].w == epoch) {
                sH[lo] = __uint_as_float(v[j].x);
                sH[lo + WLD] = __uint_as_float(v[j].z);
            } else {
                bad |= (1u << j);
            }
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > WSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) {
                    xch_give_up(p.status);
                    sFlag[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            constexpr int RCH = WH == 512 ? 5 : WNG;   // width 512: retry in chunks (register budget)
#pragma unroll
            for (int j0 = 0; j0 < WNG; j0 += RCH) {
                wu32x4 tv[RCH];
#pragma unroll
                for (int j = j0; j < j0 + RCH && j < WNG; ++j) {
                    const unsigned uo = (unsigned)(((slice + 1 + j) & (WG - 1)) * 32) * 16u;
                    tv[j - j0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, gvoff, base + uo, 16);
                }
#pragma unroll
                for (int j = j0; j < j0 + RCH && j < WNG; ++j) {
                    const int lo = lbase + ((slice + 1 + j) & (WG - 1)) * 32;
                    if (((bad >> j) & 1u) && tv[j - j0].y == epoch && tv[j - j0].w == epoch) {
                        sH[lo] = __uint_as_float(tv[j - j0].x);
                        sH[lo + WLD] = __uint_as_float(tv[j - j0].z);
                        bad &= ~(1u << j);
                    }
                }
            }
        }
    };

    const float* hrow = sH + n * WLD + 4 * g4;
    // x staging: thread (xrw = tid/16, xc = tid%16) moves the 16-byte pieces xc, xc+16, xc+32, xc+48 of row xrw
    // (narrow variant: the elements xc, xc+16, ..., xc+80)
    const int xrw = tid >> 4, xc = tid & 15;
    const int nx4 = F >> 2;   // 16-byte pieces per row (F % 4 == 0, host-checked)
    constexpr int NXR = XVEC ? 4 : (NJX > 0 ? NJX : 1);
    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * WBT;
        __syncthreads();   // previous tile fully consumed
        // Every global load of the tile goes through a buffer descriptor that covers exactly its live rows (a NULL tensor:
        // nothing): rows past the batch, absent tensors and masked columns read as 0 without a branch - a load inside a
        // branch is waited for at the merge, and the waits of this kernel are all vmcnt(0).
        const int live_rows = p.B - b0 < WBT ? p.B - b0 : WBT;
        const __amdgpu_buffer_rsrc_t h0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h0 ? p.h0 + (size_t)b0 * WH : nullptr), 0, p.h0 ? live_rows * WH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t c0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.c0 ? p.c0 + (size_t)b0 * WH : nullptr), 0, p.c0 ? live_rows * WH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(ZXM ? nullptr : p.x + (size_t)b0 * p.T * F), 0, ZXM ? 0 : live_rows * p.T * F * 4, 0x00020000);
        {
            float hv[WBT * WH / 256];
#pragma unroll
            for (int q = 0; q < WBT * WH / 256; ++q) hv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, (unsigned)((tid + 256 * q) * 4), 0, 0));
#pragma unroll
            for (int q = 0; q < WBT * WH / 256; ++q) {
                const int e = tid + 256 * q;
                sH[(e / WH) * WLD + (e % WH)] = hv[q];
            }
        }
        float c[2], hc[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned off = (unsigned)(((my_row0 + r) * WH + unit) * 4);
            c[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, off, 0, 0));
            hc[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, off, 0, 0));
        }
        unsigned xoff[NXR];   // byte offset of this thread's pieces of step 0 (masked pieces: out of range)
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            if constexpr (XVEC) xoff[i] = (xc + 16 * i < nx4) ? (unsigned)((xrw * p.T * F + 4 * xc + 64 * i) * 4) : OORB;
            else xoff[i] = (xc + 16 * i < F) ? (unsigned)((xrw * p.T * F + xc + 16 * i) * 4) : OORB;
        }
        auto load_x4 = [&](int i, int t) {
            const wu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xgrs, xoff[i], (unsigned)(t * F * 4), 0);
            return (f32x4){__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
        };
        auto load_x1 = [&](int i, int t) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, xoff[i], (unsigned)(t * F * 4), 0)); };
        float* xl = sX + xrw * WLD + (XVEC ? 4 : 1) * xc;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!ZXM) {
            f32x4 x4[2][XVEC ? NXR : 1];
            float x1[2][XVEC ? 1 : NXR];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int i = 0; i < NXR; ++i) {
                    const int tc = tt < steps ? tt : 0;     // steps == 1: the second tile is loaded and dropped
                    if constexpr (XVEC) x4[tt][i] = load_x4(i, tc);
                    else x1[tt][i] = load_x1(i, tc);
                }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (tt < steps) {
#pragma unroll
                    for (int i = 0; i < NXR; ++i) {
                        if constexpr (XVEC) {
                            if (xc + 16 * i < nx4) *(f32x4*)(xl + tt * WBT * WLD + 64 * i) = x4[tt][i];
                        } else {
                            if (xc + 16 * i < F) xl[tt * WBT * WLD + 16 * i] = x1[tt][i];
                        }
                    }
                }
        }
        __syncthreads();
        // ZX mode: this lane's pre-activations of step t sit at zx[row 4*g4 + r][t][col0 | col1] (the MFMA D layout): eight
        // loads per step, requested one step ahead
        const __amdgpu_buffer_rsrc_t zxrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(ZXM ? p.zx + (size_t)b0 * p.T * H4 : nullptr), 0, ZXM ? live_rows * p.T * H4 * 4 : 0, 0x00020000);
        f32x4 zn[2] = {z4, z4};   // zx of the NEXT step
        auto load_zx = [&](int t) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const unsigned o = (unsigned)((((4 * g4 + r) * p.T + t) * H4) * 4);
                zn[0][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zxrs, o + (unsigned)(col0 * 4), 0, 0));
                zn[1][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zxrs, o + (unsigned)(col1 * 4), 0, 0));
            }
        };
        // ---- pre-activations of step 0 ----
        f32x4 acc[2];
        acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
        acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
        if constexpr (ZXM) {
            if (steps > 0) {
                load_zx(0);
#pragma unroll
                for (int r = 0; r < 4; ++r) { acc[0][r] += zn[0][r]; acc[1][r] += zn[1][r]; }
                if (steps > 1) load_zx(1);
            }
        }
        if (steps > 0) {
            wm_begin(acc);
            if constexpr (NJX > 0) wide_mm<NJX, K_AGPR>(acc, sX + n * WLD + 4 * g4, wk);
            wide_mm<NJR>(acc, hrow, wr);
            wm_end(acc);
        }
        f32x4 xr[XVEC ? 4 : 1] = {z4};
        float xs[XVEC ? 1 : (NJX > 0 ? NJX : 1)] = {0.f};
        for (int t = 0; t < steps; ++t) {
            // x pipeline: x_{t+1} (requested during step t-1) registers -> LDS; then request x_{t+2}
            if (!ZXM && t > 0 && t + 1 < steps) {
                float* xb = xl + ((t + 1) & 1) * WBT * WLD;
#pragma unroll
                for (int i = 0; i < NXR; ++i) {
                    if constexpr (XVEC) {
                        if (xc + 16 * i < nx4) *(f32x4*)(xb + 64 * i) = xr[i];
                    } else {
                        if (xc + 16 * i < F) xb[16 * i] = xs[i];
                    }
                }
            }
            if (!ZXM && t + 2 < steps) {
#pragma unroll
                for (int i = 0; i < NXR; ++i) {
                    if constexpr (XVEC) xr[i] = load_x4(i, t + 2);
                    else xs[i] = load_x1(i, t + 2);
                }
            }
            // ---- cell update ----
            {
                float snd[4], rcv[4];
                snd[0] = hi ? acc[0][0] : acc[0][2];
                snd[1] = hi ? acc[0][1] : acc[0][3];
                snd[2] = hi ? acc[1][0] : acc[1][2];
                snd[3] = hi ? acc[1][1] : acc[1][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) rcv[k] = wswap(snd[k]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float zi = hi ? rcv[r] : acc[0][r];
                    const float zf = hi ? acc[0][2 + r] : rcv[r];
                    const float zg = hi ? rcv[2 + r] : acc[1][r];
                    const float zo = hi ? acc[1][2 + r] : rcv[2 + r];
                    const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
                    c[r] = fmaf(fg, c[r], ig * gg);
                    hc[r] = og * tanh_f(c[r]);
                    const int row = b0 + my_row0 + r;
                    if (row < p.B) {
                        if (p.reserve) {
                            float* rp = p.reserve + (((size_t)row * p.T + t) * 5) * WH + unit;
                            rp[0] = ig; rp[WH] = fg; rp[2 * WH] = gg; rp[3 * WH] = og; rp[4 * WH] = c[r];
                        }
                        if (p.hs) p.hs[((size_t)row * p.T + t) * WH + unit] = hc[r];
                    }
                }
            }
            const bool more = (t + 1 < steps);
            const bool do_xch = xch_used && more;   // the last h_t is needed by nobody in here
            unsigned par = 0;
            if (do_xch) {
                ++epoch;
                par = (epoch & 1u) * PARITY;
#pragma unroll
                for (int r = 0; r < 1; ++r) {
                    const wu32x4 gr = {__float_as_uint(hc[0]), epoch, __float_as_uint(hc[1]), epoch};
                    if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, pub_off, par, 1);
                    else __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, pub_off, par, 16);
                }
            }
            __syncthreads();   // barrier 1: every wave is done reading sH; x_{t+1} is in LDS
            if (do_xch) {
#pragma unroll
                for (int r = 0; r < 2; ++r) sH[(my_row0 + r) * WLD + unit] = hc[r];
            }
            acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
            acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
            if constexpr (ZXM) {
                if (more) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) { acc[0][r] += zn[0][r]; acc[1][r] += zn[1][r]; }
                    if (t + 2 < steps) load_zx(t + 2);
                }
            } else if (more) {   // x_{t+1} . K needs no remote data
                wm_begin(acc);
                wide_mm<NJX, K_AGPR>(acc, sX + ((t + 1) & 1) * WBT * WLD + n * WLD + 4 * g4, wk);
                wm_end(acc);
            }
            // The gather is requested only now: the partners published at about the same moment as this workgroup, and
            // an sc1 store takes most of a microsecond to become visible - loads issued right behind the own publish
            // came back stale and cost a second sweep (measured: 72 -> 56 us narrow, 79 -> 68 us wide, B = 512, T = 10).
            if (do_xch) gather_issue(par);
            if (do_xch) gather_finish(par);
            __syncthreads();   // barrier 2: the whole h_t tile is in LDS
            if (sFlag[0]) { aborted = true; break; }
            if (more) {
                wm_begin(acc);
                wide_mm<NJR>(acc, hrow, wr);
                wm_end(acc);
            }
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.hT) p.hT[(size_t)row * WH + unit] = hc[r];
                    if (p.cT) p.cT[(size_t)row * WH + unit] = c[r];
                }
            }
        }
    }
    if (xch_used) xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

}  // namespace

bool wide_shape_ok(int F, int H) { return H == 256 && F > 96 && F <= 256 && (F & 3) == 0; }

// F <= 96 at H = 256: the narrow variant, preferred over lstm_cluster.hip while the batch is at most 32 tiles
bool wide_narrow_preferred(int B, int F, int H) { return H == 256 && F >= 1 && F <= 96 && B > 0 && B <= 32 * WBT; }

// width 512 (16 workgroups per tile): narrow input (F <= 96) or a precomputed input projection (p.zx)
bool wide512_shape_ok(int F, int H, bool zx) { return H == 512 && (zx || (F >= 1 && F <= 96)) && device_cu_count() >= 16; }

template <int NJX, int WH, int WG>
static int launch_wide_t(LstmParams& p, hipStream_t stream) {
    p.num_tiles = (p.B + WBT - 1) / WBT;
    const int max_groups = device_cu_count() / WG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("wide LSTM layer needs at least %d CUs", WG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 2 * WBT * WH * sizeof(unsigned long long) > kXchBytes - kHelloBytes) { set_error("wide LSTM layer: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.epoch_span = p.T * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;   // no memset: tags continue from the header
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    const size_t lds = sizeof(float) * (3 * WBT * (WH + 4)) + 64;
    void (*kern)(LstmParams) = p.act == FOV_ACT_HARD_SIGMOID ? lstm_wide_kernel<FOV_ACT_HARD_SIGMOID, NJX, WH, WG>
                                                             : lstm_wide_kernel<FOV_ACT_SIGMOID, NJX, WH, WG>;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(xch_padded_groups(p.num_groups) * WG), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("wide LSTM launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// p.status / p.xch point into the caller's workspace (cluster_workspace_bytes(B, H): header + the fixed granule area)
int launch_wide(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0) return FOV_OK;
    if (p.H == 512) {
        if (p.zx) return launch_wide_t<0, 512, 16>(p, stream);
        if (p.F > 96) { set_error("wide LSTM layer, width 512: F > 96 needs the precomputed input projection"); return FOV_ERR_INVALID; }
        return launch_wide_t<6, 512, 16>(p, stream);
    }
    const bool narrow = p.F <= 96;
    if (!narrow && (((uintptr_t)p.x) & 15) != 0) { set_error("wide LSTM layer: x must be 16-byte aligned"); return FOV_ERR_INVALID; }
    return narrow ? launch_wide_t<6, 256, 8>(p, stream) : launch_wide_t<16, 256, 8>(p, stream);
}

}  // namespace fov
