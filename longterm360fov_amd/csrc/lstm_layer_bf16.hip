// Persistent LSTM layer with bf16 matrix-core operands (BASELINE.json configs[4]), H = 256, F <= 256: the encoder
// layers of the others-mixing model (mycode/given_others_gt_mean_var_seq2seq.py:108-115) when the model runs in bf16.
//
// The bf16 sibling of lstm_wide.hip.  z_t = bf16(x_t) . bf16(K) + bf16(h_{t-1}) . bf16(R) + b on
// v_mfma_f32_16x16x32_bf16 with fp32 accumulation; gates, cell state c and every stored tensor (hs, reserve, hT, cT)
// stay fp32.  K and R slices live in registers for the whole launch as packed B fragments (64 + 64 registers per
// lane at F = H = 256); x_t and h_t tiles sit in LDS as bf16 row images (one ds_read_b128 = one A fragment).
// Per step and wave the matrix work is (NKB + 8) x 2 MFMAs of 16 cycles (fp32 path: (NJX + 16) x 8 of 32 cycles): the step is
// bound by the exchange of h_t, which carries the two bf16 values a lane owns in ONE granule (half the granules of
// the fp32 kernels).
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

// Diagnostic build only (-DFOV_STAMPS, tools/stamp_bf16_layer.py): s_memtime stamps of one wave per step.
#ifdef FOV_STAMPS
constexpr int QSTAMP_SLOTS = 12;
constexpr int QSTAMP_STEPS = 32;
__device__ unsigned long long g_q_stamps[QSTAMP_STEPS][QSTAMP_SLOTS];
// branch-free and LDS-buffered (round 4, as lstm_bwd8.hip's B8_STAMP): a global store per stamp sat in the wave's vmcnt queue and
// added ~2 000 cycles to every s_waitcnt vmcnt(0) behind it; a stamp inside `if (stamp_on)` split the basic block and let the
// optimiser sink the arithmetic in front of it past it
#define Q_STAMP(slot)                                                                          \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        sStamps[(stamp_on && t < QSTAMP_STEPS - 1) ? t * QSTAMP_SLOTS + slot : (QSTAMP_STEPS - 1) * QSTAMP_SLOTS + 11] = t_; \
    } while (0)
#else
#define Q_STAMP(slot) do { } while (0)
#endif

namespace {

// NKB: 32-wide k-blocks of the input kernel (3: F <= 96, 8: F <= 256).  XVEC: x rows are read in 16-byte pieces
// (F % 4 == 0, x 16-byte aligned: the stacked layer over a 256-wide sequence); otherwise element by element.
template <int ACT, int NKB, bool XVEC>
__global__ __launch_bounds__(256, 1) void lstm_layer_bf16_kernel(LstmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short sH[QBT * QLD];
    __shared__ __attribute__((aligned(16))) unsigned short sX[2 * QBT * QLD];
    __shared__ int sFlag[4];
    __shared__ __attribute__((aligned(16))) unsigned sStage[QST_LDS_WORDS];   // the prologue's weight staging (bf16_common.h)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    if (!q_group_slice(p.num_groups, group, slice)) { q_spare_leaves(p.status, p.T > 1); return; }
    const int F = p.F, steps = p.T;
    const int unit = 32 * slice + 8 * wave + (n & 7);
    const int hi = n >> 3;
    const int col0 = hi * QH + unit, col1 = (2 + hi) * QH + unit;   // gate columns of N-tile 0 ([i | f]) / 1 ([g | o])
    constexpr int H4 = 4 * QH;
    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    const bool xch_used = steps > 1;
    __shared__ unsigned sXch[4];
    const XchHeader header = xch_arrive_request(p.status, xch_used);   // taken behind the first weight requests (xch_common.h)
    const unsigned timeout_word = xch_timeout_word(p.status);
    const unsigned arrival = 0u;
#ifdef FOV_STAMPS
    __shared__ unsigned long long sStamps[QSTAMP_STEPS * QSTAMP_SLOTS];
    const bool stamp_on = (blockIdx.x == 5 && tid == 0);
    if (stamp_on) g_q_stamps[QSTAMP_STEPS - 1][0] = __builtin_amdgcn_s_memtime();   // kernel entry
#endif

    // ---- resident weights: packed bf16 B fragments (rows of K beyond F are zero) ----
    qu32x4 wk[NKB][2], wr[8][2];
    stage_weight_sets(wk, p.K, F, wr, p.R, QH, H4, slice, sStage, [&]() {
        xch_arrive_commit(p.status, sXch, header, group, slice, xch_used);
        for (int i = tid; i < 2 * QBT * QLD; i += 256) sX[i] = 0;   // columns >= F stay zero (under the first stage's round trip)
    });
    const bool poisoned = xch_timeout_set(timeout_word) && xch_used;
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;
    const float bv[2] = {p.b[col0], p.b[col1]};

    // ---- exchange bookkeeping ----
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 2 * (Q_TILE_BYTES / 8), 0, (int)(2 * Q_TILE_BYTES), 0x00020000);
    const int my_row0 = 4 * g4 + 2 * hi;                                       // this lane's cells: rows my_row0, +1 of `unit`
    const unsigned pub_off = (unsigned)((my_row0 >> 1) * QH + unit) * 8u;
#ifdef FOV_STAMPS
    asm volatile("" :: "v"(wr[7][1]), "v"(wr[0][0]), "v"(wk[0][0]), "v"(wk[NKB - 1][1]));   // the fragments exist by now
    if (stamp_on) g_q_stamps[QSTAMP_STEPS - 1][1] = __builtin_amdgcn_s_memtime();   // weights resident
#endif
    if (xch_used) xch_hello_poll(p.status, sXch, group, QG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    XchTicket ticket = {0u, 0u, 0u};
    if (xch_used) ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (xch_used && tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    // x staging: thread (xrw = tid / 16, xc = tid % 16) moves the elements xc, xc + 16, ... of row xrw
    // (XVEC: the 16-byte pieces xc, xc + 16, ...: four elements each)
    const int xrw = tid >> 4, xc = tid & 15;
    constexpr int NXE = XVEC ? NKB / 2 : 2 * NKB;   // pieces (XVEC) / elements per thread
    const int nx4 = F >> 2;
    QGather gq;
    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * QBT;
        __syncthreads();   // previous tile fully consumed
        // Every global load of the tile goes through a buffer descriptor that covers exactly its live rows (a NULL tensor:
        // nothing): rows past the batch, absent tensors and masked columns read as 0 without a branch (a load inside a
        // branch is waited for at the merge).
        constexpr unsigned OORB = 0x80000000u;
        const int live_rows = p.B - b0 < QBT ? p.B - b0 : QBT;
        const __amdgpu_buffer_rsrc_t h0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h0 ? p.h0 + (size_t)b0 * QH : nullptr), 0, p.h0 ? live_rows * QH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t c0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.c0 ? p.c0 + (size_t)b0 * QH : nullptr), 0, p.c0 ? live_rows * QH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x + (size_t)b0 * p.T * F), 0, live_rows * p.T * F * 4, 0x00020000);
        {
            float hv[QBT * QH / 256];
#pragma unroll
            for (int q = 0; q < QBT * QH / 256; ++q) hv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, (unsigned)((tid + 256 * q) * 4), 0, 0));
#pragma unroll
            for (int q = 0; q < QBT * QH / 256; ++q) {
                const int e = tid + 256 * q;
                sH[(e >> 8) * QLD + (e & 255)] = bf16_bits(hv[q]);
            }
        }
        float c[2], hc[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const unsigned off = (unsigned)(((my_row0 + r) * QH + unit) * 4);
            c[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, off, 0, 0));
            hc[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, off, 0, 0));
        }
        unsigned xoff[NXE];   // byte offset of this thread's pieces of step 0 (masked pieces: out of range)
#pragma unroll
        for (int i = 0; i < NXE; ++i) {
            if constexpr (XVEC) xoff[i] = (xc + 16 * i < nx4) ? (unsigned)((xrw * p.T * F + 4 * xc + 64 * i) * 4) : OORB;
            else xoff[i] = (xc + 16 * i < F) ? (unsigned)((xrw * p.T * F + xc + 16 * i) * 4) : OORB;
        }
        auto load_x4 = [&](int i, int t) {
            const qu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xgrs, xoff[i], (unsigned)(t * F * 4), 0);
            return (f32x4){__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
        };
        auto load_x1 = [&](int i, int t) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, xoff[i], (unsigned)(t * F * 4), 0)); };
        unsigned short* xl = sX + xrw * QLD + (XVEC ? 4 : 1) * xc;
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        {
            f32x4 v4[2][XVEC ? NXE : 1];
            float v1[2][XVEC ? 1 : NXE];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int i = 0; i < NXE; ++i) {
                    const int tc = tt < steps ? tt : 0;     // steps == 1: the second tile is loaded and never read
                    if constexpr (XVEC) v4[tt][i] = load_x4(i, tc);
                    else v1[tt][i] = load_x1(i, tc);
                }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int i = 0; i < NXE; ++i) {
                    if constexpr (XVEC) {
                        if (xc + 16 * i < nx4)
                            *(qu32x2*)(xl + tt * QBT * QLD + 64 * i) = (qu32x2){pack_bf16(v4[tt][i][0], v4[tt][i][1]), pack_bf16(v4[tt][i][2], v4[tt][i][3])};
                    } else {
                        if (xc + 16 * i < F) xl[tt * QBT * QLD + 16 * i] = bf16_bits(v1[tt][i]);
                    }
                }
        }
        __syncthreads();
        // ---- pre-activations of step 0 ----
        f32x4 acc[2];
        acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
        acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
        if (steps > 0) {
            qmm<0, NKB, NKB>(acc, sX, n, g4, wk);
            qmm<0, 8, 8>(acc, sH, n, g4, wr);
        }
        f32x4 xr[XVEC ? NXE : 1];
        float xs[XVEC ? 1 : NXE];
#pragma unroll
        for (int i = 0; i < (XVEC ? NXE : 1); ++i) xr[i] = z4;
#pragma unroll
        for (int i = 0; i < (XVEC ? 1 : NXE); ++i) xs[i] = 0.f;
        for (int t = 0; t < steps; ++t) {
            Q_STAMP(0);
            // x pipeline: x_{t+1} (requested during step t-1) registers -> LDS; then request x_{t+2}
            if (t > 0 && t + 1 < steps) {
                unsigned short* xb = xl + ((t + 1) & 1) * QBT * QLD;
#pragma unroll
                for (int i = 0; i < NXE; ++i) {
                    if constexpr (XVEC) {
                        if (xc + 16 * i < nx4) *(qu32x2*)(xb + 64 * i) = (qu32x2){pack_bf16(xr[i][0], xr[i][1]), pack_bf16(xr[i][2], xr[i][3])};
                    } else {
                        if (xc + 16 * i < F) xb[16 * i] = bf16_bits(xs[i]);
                    }
                }
            }
            if (t + 2 < steps) {
#pragma unroll
                for (int i = 0; i < NXE; ++i) {
                    if constexpr (XVEC) xr[i] = load_x4(i, t + 2);
                    else xs[i] = load_x1(i, t + 2);
                }
            }
            Q_STAMP(1);
            // ---- cell update (fp32) ----
            float gt[2][4];   // activated gates: the tape stores wait until the publish and the gather are out
            {
                float zi[2], zf[2], zg[2], zo[2];
                gates_of_lane(acc, hi, zi, zf, zg, zo);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float ig = rec_act<ACT>(zi[r]), fg = rec_act<ACT>(zf[r]), gg = tanh_f(zg[r]), og = rec_act<ACT>(zo[r]);
                    c[r] = fmaf(fg, c[r], ig * gg);
                    hc[r] = og * tanh_f(c[r]);
                    gt[r][0] = ig; gt[r][1] = fg; gt[r][2] = gg; gt[r][3] = og;
                }
            }
            Q_STAMP(2);
            const bool more = (t + 1 < steps);
            const bool do_xch = xch_used && more;   // the last h_t is needed by nobody in here
            unsigned par = 0;
            const unsigned hpair = pack_bf16(hc[0], hc[1]);
            if (do_xch) {
                ++epoch;
                par = (epoch & 1u) * Q_TILE_BYTES;
                XCH_STORE_B64(ticket.same_xcd, ((qu32x2){hpair, epoch}), xrs, pub_off, par);
            }
            Q_STAMP(3);
            __syncthreads();   // barrier 1: every wave is done reading sH; x_{t+1} is in LDS
            Q_STAMP(4);
            if (more) {
                sH[my_row0 * QLD + unit] = (unsigned short)(hpair & 0xffffu);
                sH[(my_row0 + 1) * QLD + unit] = (unsigned short)(hpair >> 16);
            }
            acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
            acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
            // x_{t+1} . K needs no remote data; the gather is requested behind it (the partners publish at about the same
            // moment and an sc1 store takes most of a microsecond to become visible)
            if (more) qmm<0, NKB, NKB>(acc, sX + ((t + 1) & 1) * QBT * QLD, n, g4, wk);
            Q_STAMP(5);
            if (do_xch) q_gather_issue(gq, xrs, par, slice, tid);
#pragma unroll
            for (int r = 0; r < 2; ++r) {   // tape of the step, under the gather's round trip
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.reserve) {
                        float* rp = p.reserve + (((size_t)row * p.T + t) * 5) * QH + unit;
                        rp[0] = gt[r][0]; rp[QH] = gt[r][1]; rp[2 * QH] = gt[r][2]; rp[3 * QH] = gt[r][3]; rp[4 * QH] = c[r];
                    }
                    if (p.hs) p.hs[((size_t)row * p.T + t) * QH + unit] = hc[r];
                }
            }
            Q_STAMP(6);
            if (do_xch) {
                if (!q_gather_finish(gq, xrs, par, slice, tid, epoch, sH, p.status)) sFlag[0] = 1;
            }
            Q_STAMP(7);
            __syncthreads();   // barrier 2: the whole h_t tile is in LDS
            Q_STAMP(8);
            if (sFlag[0]) { aborted = true; break; }
            if (more) qmm<0, 8, 8>(acc, sH, n, g4, wr);
            Q_STAMP(9);
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.hT) p.hT[(size_t)row * QH + unit] = hc[r];
                    if (p.cT) p.cT[(size_t)row * QH + unit] = c[r];
                }
            }
        }
    }
#ifdef FOV_STAMPS
    if (stamp_on) {
        g_q_stamps[QSTAMP_STEPS - 1][2] = __builtin_amdgcn_s_memtime();   // all tiles done
        for (int i = 0; i < (QSTAMP_STEPS - 1) * QSTAMP_SLOTS; ++i) (&g_q_stamps[0][0])[i] = sStamps[i];
    }
#endif
    if (xch_used) xch_settle(p.status, ticket, (unsigned)p.epoch_span);
#ifdef FOV_STAMPS
    if (stamp_on) g_q_stamps[QSTAMP_STEPS - 1][3] = __builtin_amdgcn_s_memtime();   // left
#endif
}

}  // namespace

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_q_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_q_stamps), sizeof(unsigned long long) * QSTAMP_STEPS * QSTAMP_SLOTS);
}
#endif

bool layer_bf16_shape_ok(int F, int H) { return H == QH && F >= 1 && F <= 256; }

// p.status / p.xch point into the caller's workspace (header + the fixed granule area)
int launch_layer_bf16(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0) return FOV_OK;
    if (!layer_bf16_shape_ok(p.F, p.H)) { set_error("bf16 LSTM layer: H = 256 and F <= 256 only (got H=%d F=%d)", p.H, p.F); return FOV_ERR_UNSUPPORTED; }
    p.num_tiles = (p.B + QBT - 1) / QBT;
    const int max_groups = device_cu_count() / QG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("bf16 LSTM layer needs at least %d CUs", QG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 2 * Q_TILE_BYTES > kXchBytes - kHelloBytes) { set_error("bf16 LSTM layer: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.epoch_span = p.T * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    const bool narrow = p.F <= 96;
    const bool xvec = !narrow && (p.F & 3) == 0 && (((uintptr_t)p.x) & 15) == 0;
    const bool hs_ = p.act == FOV_ACT_HARD_SIGMOID;
    void (*kern)(LstmParams) =
        narrow ? (hs_ ? lstm_layer_bf16_kernel<FOV_ACT_HARD_SIGMOID, 3, false> : lstm_layer_bf16_kernel<FOV_ACT_SIGMOID, 3, false>)
        : xvec ? (hs_ ? lstm_layer_bf16_kernel<FOV_ACT_HARD_SIGMOID, 8, true> : lstm_layer_bf16_kernel<FOV_ACT_SIGMOID, 8, true>)
               : (hs_ ? lstm_layer_bf16_kernel<FOV_ACT_HARD_SIGMOID, 8, false> : lstm_layer_bf16_kernel<FOV_ACT_SIGMOID, 8, false>);
    hipLaunchKernelGGL(kern, dim3(q_padded_groups(p.num_groups) * QG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("bf16 LSTM layer launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
