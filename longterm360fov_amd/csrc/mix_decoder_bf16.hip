// bf16 sibling of mix_decoder.hip (BASELINE.json configs[4]): the whole unrolled no-teacher-forcing decoder of
// mycode/given_others_gt_mean_var_seq2seq.py:203-299 in ONE persistent launch, with bf16 operands into
// v_mfma_f32_16x16x32_bf16 for every product that has h or x on its left (gate GEMMs of both layers and the Dense(6)
// head), fp32 accumulation, fp32 gates / cell state / mixing layer / tape.
//
//   per step t:  h1,c1 = LSTM1(x_t; h1,c1)          K1:(O,4H)  R1:(H,4H)
//                h2,c2 = LSTM2(h1;  h2,c2)          K2:(H,4H)  R2:(H,4H)
//                p = tanh(h2 Wd + bd)
//                m = tanh(p Wp + others_t . W_oth + b_mix)      (others part hoisted by the caller, fp32)
//                x_{t+1} = m
//
// Same ownership as the fp32 kernel (tile of 16 sequences per group of 8 workgroups, workgroup `slice` owns hidden
// units [32*slice, +32) of BOTH layers).  What bf16 changes:
//   * R1, K2, R2 slices are ALL register-resident as packed B fragments (3 x 64 registers per lane) - no LDS staging of
//     K2, no pre-pack launch;
//   * a layer's matrix work per step and wave is 16 MFMAs of 16 cycles (fp32: 128 of 32 cycles): the step is bound by
//     its two exchanges, which now move ONE granule per lane ({bf16 pair, epoch}) instead of two;
//   * the h tiles in LDS are bf16 row images (one ds_read_b128 = one A fragment).
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

namespace {

template <int ACT, bool TRAIN>
__global__ __launch_bounds__(256, 1) void mix_decoder_bf16_kernel(MixDecParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short sH1[QBT * QLD];
    __shared__ __attribute__((aligned(16))) unsigned short sH2[QBT * QLD];
    __shared__ __attribute__((aligned(16))) unsigned short sX[QBT * 32];   // decoder input x_t as a bf16 A tile (k < O used)
    __shared__ float sWp[64];                                               // [8][8] mixing kernel (pred part)
    __shared__ float sPart[4 * 256];                                        // [4 waves][16 rows][16 cols] partial Dense products
    __shared__ int sFlag[4];
    __shared__ __attribute__((aligned(16))) unsigned sStage[QST_LDS_WORDS];   // the prologue's weight staging (bf16_common.h)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int O = p.O;
    int group, slice;
    if (!q_group_slice(p.num_groups, group, slice)) { q_spare_leaves(p.status); return; }
    const int unit = 32 * slice + 8 * wave + (n & 7);   // hidden unit of this lane's columns
    const int hi = n >> 3;                               // 0: columns i / g, 1: columns f / o
    const int col0 = hi * QH + unit, col1 = (2 + hi) * QH + unit;
    constexpr int H4 = 4 * QH;

    __shared__ unsigned sXch[4];
    const XchHeader header = xch_arrive_request(p.status);   // taken behind the first weight requests (xch_common.h)
    const unsigned timeout_word = xch_timeout_word(p.status);
    const unsigned arrival = 0u;
    // ---- resident weights (packed bf16 B fragments) ----
    qu32x4 w1[8][2], wk2[8][2], w2[8][2], wk1[1][2];
    load_weight_set<1>(wk1, p.K1, H4, O, g4, col0, col1);      // K1 (O <= 8 rows): one zero-padded k-block
    // the three (H, 4H) sets through LDS (bf16_common.h); K2p: here the plain (H,4H) kernel of layer 2
    stage_weight_sets(w1, p.R1, QH, wk2, p.K2p, QH, w2, p.R2, QH, H4, slice, sStage,
                      [&]() {
                          xch_arrive_commit(p.status, sXch, header, group, slice);
                          for (int e = tid; e < QBT * 32; e += 256) sX[e] = 0;   // columns >= 8 stay zero
                      });
    const bool poisoned = xch_timeout_set(timeout_word);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;
    const float b1v[2] = {p.b1[col0], p.b1[col1]}, b2v[2] = {p.b2[col0], p.b2[col1]};
    const float bdv = ((tid & 15) < O) ? p.bd[tid & 15] : 0.f;   // Dense bias of this thread's head output
    // Dense kernel as B fragments: wave w contracts hidden units [64w, 64w + 64) = k-blocks 2w, 2w + 1; column n = output
    qu32x4 wd[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (n < O) ? p.Wd[(size_t)(32 * (2 * wave + q) + 8 * g4 + j) * O + n] : 0.f;
        wd[q] = (qu32x4){pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
    }
    for (int e = tid; e < 64; e += 256) sWp[e] = ((e >> 3) < O && (e & 7) < O) ? p.Wp[(e >> 3) * O + (e & 7)] : 0.f;

    // ---- exchange bookkeeping: [layer 2][parity 2][row pair 8][unit 256] granules per group ----
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 4 * (Q_TILE_BYTES / 8), 0, (int)(4 * Q_TILE_BYTES), 0x00020000);
    const int my_row0 = 4 * g4 + 2 * hi;
    const unsigned pub_off = (unsigned)((my_row0 >> 1) * QH + unit) * 8u;
    constexpr unsigned LAYER_BYTES = 2u * Q_TILE_BYTES;
    xch_hello_poll(p.status, sXch, group, QG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)
    QGather gq;

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * QBT;
        __syncthreads();   // previous tile fully consumed
        // ---- initial state: full h tiles to LDS (bf16), own c cells to registers, x_0 ----
        for (int e = tid; e < QBT * QH; e += 256) {
            const int row = e >> 8, u = e & 255;
            const bool ok = b0 + row < p.B;
            sH1[row * QLD + u] = bf16_bits(ok ? p.h1_0[(size_t)(b0 + row) * QH + u] : 0.f);
            sH2[row * QLD + u] = bf16_bits(ok ? p.h2_0[(size_t)(b0 + row) * QH + u] : 0.f);
        }
        if (tid < QBT * 8) {
            const int row = tid >> 3, o = tid & 7;
            const float xv = (o < O && b0 + row < p.B) ? p.dec0[(size_t)(b0 + row) * O + o] : 0.f;
            sX[row * 32 + o] = bf16_bits(xv);
        }
        float c1[2], c2[2], h1c[2] = {0.f, 0.f}, h2c[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = b0 + my_row0 + r;
            c1[r] = (row < p.B) ? p.c1_0[(size_t)row * QH + unit] : 0.f;
            c2[r] = (row < p.B) ? p.c2_0[(size_t)row * QH + unit] : 0.f;
        }
        __syncthreads();
        const int hd_o = tid & 15, hd_row = b0 + (tid >> 4);   // head: thread = (sequence, output)
        f32x4 acc1[2], acc2[2];
        // recurrent half of layer 1, step 0
        acc1[0] = (f32x4){b1v[0], b1v[0], b1v[0], b1v[0]};
        acc1[1] = (f32x4){b1v[1], b1v[1], b1v[1], b1v[1]};
        qmm<0, 8, 8>(acc1, sH1, n, g4, w1);
        __syncthreads();   // every wave has read h1_0 before the first own-slice write of h1_t into the tile
        for (int t = 0; t < p.T_out; ++t) {
            // the "others" term of the mixing layer for this step: requested now, consumed in the head
            const bool oth_live = hd_o < O && hd_row < p.B;   // dead lanes read element 0, masked (no load inside a branch)
            const float oth_ld = p.oth_proj[oth_live ? (size_t)hd_row * p.oth_sb + (size_t)t * p.oth_st + hd_o : 0];
            const float othv = oth_live ? oth_ld : 0.f;
            ++epoch;
            const unsigned par = (epoch & 1u) * Q_TILE_BYTES;
            // ================= layer 1: + x_t . K1, cell update =================
            {
                const qu32x4 xa = *(const qu32x4*)(sX + n * 32 + 8 * g4);   // k = 8*g4 + j: only g4 == 0 carries data
                qmfma(acc1[0], xa, wk1[0][0]);
                qmfma(acc1[1], xa, wk1[0][1]);
            }
            float gt[2][4];   // activated gates of the step: the tape stores wait until the publish and the gather are out
            {
                float zi[2], zf[2], zg[2], zo[2];
                gates_of_lane(acc1, hi, zi, zf, zg, zo);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float ig = rec_act<ACT>(zi[r]), fg = rec_act<ACT>(zf[r]), gg = tanh_f(zg[r]), og = rec_act<ACT>(zo[r]);
                    c1[r] = fmaf(fg, c1[r], ig * gg);
                    h1c[r] = og * tanh_f(c1[r]);
                    gt[r][0] = ig; gt[r][1] = fg; gt[r][2] = gg; gt[r][3] = og;
                }
            }
            // publish h1_t, write the own cells into the tile, run h2_{t-1} . R2 under the exchange
            const unsigned h1pair = pack_bf16(h1c[0], h1c[1]);
            XCH_STORE_B64(ticket.same_xcd, ((qu32x2){h1pair, epoch}), xrs, pub_off, par);
            sH1[my_row0 * QLD + unit] = (unsigned short)(h1pair & 0xffffu);
            sH1[(my_row0 + 1) * QLD + unit] = (unsigned short)(h1pair >> 16);
            acc2[0] = (f32x4){b2v[0], b2v[0], b2v[0], b2v[0]};
            acc2[1] = (f32x4){b2v[1], b2v[1], b2v[1], b2v[1]};
            qmm<0, 8, 8>(acc2, sH2, n, g4, w2);
            q_gather_issue(gq, xrs, par, slice, tid);
            if (TRAIN) {   // tape of layer 1, under the gather's round trip
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = b0 + my_row0 + r;
                    if (row < p.B) {
                        float* rp = p.res1 + (((size_t)t * p.B + row) * 5) * QH + unit;
                        rp[0] = gt[r][0]; rp[QH] = gt[r][1]; rp[2 * QH] = gt[r][2]; rp[3 * QH] = gt[r][3]; rp[4 * QH] = c1[r];
                        p.H1[((size_t)t * p.B + row) * QH + unit] = h1c[r];
                        p.C1[((size_t)t * p.B + row) * QH + unit] = c1[r];
                    }
                }
            }
            if (!q_gather_finish(gq, xrs, par, slice, tid, epoch, sH1, p.status)) sFlag[0] = 1;
            __syncthreads();   // barrier D: the whole h1_t tile is in LDS; every wave is done reading sH2
            if (sFlag[0]) { aborted = true; break; }
            // ================= layer 2: + h1_t . K2, cell update =================
            qmm<0, 8, 8>(acc2, sH1, n, g4, wk2);
            {
                float zi[2], zf[2], zg[2], zo[2];
                gates_of_lane(acc2, hi, zi, zf, zg, zo);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float ig = rec_act<ACT>(zi[r]), fg = rec_act<ACT>(zf[r]), gg = tanh_f(zg[r]), og = rec_act<ACT>(zo[r]);
                    c2[r] = fmaf(fg, c2[r], ig * gg);
                    h2c[r] = og * tanh_f(c2[r]);
                    gt[r][0] = ig; gt[r][1] = fg; gt[r][2] = gg; gt[r][3] = og;
                }
            }
            const unsigned h2pair = pack_bf16(h2c[0], h2c[1]);
            XCH_STORE_B64(ticket.same_xcd, ((qu32x2){h2pair, epoch}), xrs, pub_off, LAYER_BYTES + par);
            sH2[my_row0 * QLD + unit] = (unsigned short)(h2pair & 0xffffu);
            sH2[(my_row0 + 1) * QLD + unit] = (unsigned short)(h2pair >> 16);
            const bool more = (t + 1 < p.T_out);
            // recurrent half of layer 1 for step t+1 under the gather of h2_t
            acc1[0] = (f32x4){b1v[0], b1v[0], b1v[0], b1v[0]};
            acc1[1] = (f32x4){b1v[1], b1v[1], b1v[1], b1v[1]};
            if (more) qmm<0, 8, 8>(acc1, sH1, n, g4, w1);
            q_gather_issue(gq, xrs, LAYER_BYTES + par, slice, tid);
            if (TRAIN) {   // tape of layer 2, under the gather's round trip
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = b0 + my_row0 + r;
                    if (row < p.B) {
                        float* rp = p.res2 + (((size_t)t * p.B + row) * 5) * QH + unit;
                        rp[0] = gt[r][0]; rp[QH] = gt[r][1]; rp[2 * QH] = gt[r][2]; rp[3 * QH] = gt[r][3]; rp[4 * QH] = c2[r];
                        p.H2[((size_t)t * p.B + row) * QH + unit] = h2c[r];
                        p.C2[((size_t)t * p.B + row) * QH + unit] = c2[r];
                    }
                }
            }
            if (!q_gather_finish(gq, xrs, LAYER_BYTES + par, slice, tid, epoch, sH2, p.status)) sFlag[0] = 1;
            __syncthreads();   // barrier G: the whole h2_t tile is in LDS
            if (sFlag[0]) { aborted = true; break; }
            // ================= head: p = tanh(h2 Wd + bd), m = tanh(p Wp + add) =================
            {
                // Dense on the matrix pipe: wave w contracts hidden units [64w, 64w + 64) of the h2 tile with Wd (column
                // n = output o, zero for n >= O); the four partial 16 x 16 products meet in LDS.
                f32x4 dacc = {0.f, 0.f, 0.f, 0.f};
                qmfma(dacc, lds_afrag(sH2, n, g4, 2 * wave), wd[0]);
                qmfma(dacc, lds_afrag(sH2, n, g4, 2 * wave + 1), wd[1]);
#pragma unroll
                for (int r = 0; r < 4; ++r) sPart[(wave * 16 + 4 * g4 + r) * 16 + n] = dacc[r];
                __syncthreads();   // the four partials are in LDS
                const int row = tid >> 4, o = tid & 15;   // the 16 lanes of a row group share a sequence
                float pv = 0.f;
                if (o < O) {
                    pv = sPart[row * 16 + o] + sPart[(16 + row) * 16 + o] + sPart[(32 + row) * 16 + o] + sPart[(48 + row) * 16 + o];
                    pv = tanh_f(pv + bdv);
                }
                const int brow = b0 + row;
                float z = othv;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float pk = __shfl(pv, (lane & ~15) | k);   // p[row][k] from the lane that owns it
                    z = fmaf(pk, sWp[k * 8 + (o & 7)], z);
                }
                const float mv = (o < O) ? tanh_f(z) : 0.f;
                if (o < 8) sX[row * 32 + o] = bf16_bits(mv);   // x_{t+1}
                if (slice == 0 && o < O && brow < p.B) {
                    p.out[((size_t)t * p.B + brow) * O + o] = mv;
                    if (TRAIN) p.P[((size_t)t * p.B + brow) * O + o] = pv;
                }
            }
            __syncthreads();   // barrier H: x_{t+1} is in LDS
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.h1T) p.h1T[(size_t)row * QH + unit] = h1c[r];
                    if (p.c1T) p.c1T[(size_t)row * QH + unit] = c1[r];
                    if (p.h2T) p.h2T[(size_t)row * QH + unit] = h2c[r];
                    if (p.c2T) p.c2T[(size_t)row * QH + unit] = c2[r];
                }
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

}  // namespace

int mix_decoder_bf16_launch(MixDecParams p, const float* K2, int act, int train, void* workspace, hipStream_t stream) {
    if (p.B == 0 || p.T_out == 0) return FOV_OK;
    p.num_tiles = (p.B + QBT - 1) / QBT;
    const int max_groups = device_cu_count() / QG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("fused mixing decoder needs at least %d CUs", QG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 4 * Q_TILE_BYTES > kXchBytes - kHelloBytes) { set_error("mix_decoder_bf16: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    p.K2p = K2;   // no packed copy: the kernel builds its register-resident fragments from the plain (H,4H) kernel
    p.epoch_span = p.T_out * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(MixDecParams) = nullptr;
    if (act == FOV_ACT_HARD_SIGMOID) kern = train ? mix_decoder_bf16_kernel<FOV_ACT_HARD_SIGMOID, true> : mix_decoder_bf16_kernel<FOV_ACT_HARD_SIGMOID, false>;
    else kern = train ? mix_decoder_bf16_kernel<FOV_ACT_SIGMOID, true> : mix_decoder_bf16_kernel<FOV_ACT_SIGMOID, false>;
    hipLaunchKernelGGL(kern, dim3(q_padded_groups(p.num_groups) * QG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder_bf16 launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
