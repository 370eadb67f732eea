// bf16 sibling of mix_decoder_bwd.hip (BASELINE.json configs[4]): BPTT through the whole unrolled others-mixing decoder
// (mixing head, layer 2, layer 1, feedback path; mycode/given_others_gt_mean_var_seq2seq.py:203-308) in ONE persistent
// launch, with bf16 operands into v_mfma_f32_16x16x32_bf16 for the four transposed products
//     dz2 . R2^T -> dh2_{t-1},   dz2 . K2^T -> dh1_t,   dz1 . R1^T -> dh1_{t-1},   dz1 . K1^T -> dx_t
// fp32 accumulation, fp32 gates backward / head backward / dc / tapes.  Same ownership (tile of 16 sequences per group
// of 8 workgroups, workgroup `slice` owns hidden units [32*slice, +32) of both layers, a lane two adjacent units of one row)
// and the same outputs as the fp32 kernel - but the products are split by OUTPUT unit (N-split), not by gate column:
//   * the fp32 kernel multiplies its own 128 gate columns of dz into partial sums for ALL 256 units and sends them
//     to their owners: 48 fp32 {value, epoch} granules per lane and step, 8-byte write-through stores - the expensive
//     side of the exchange, 18 us per step when the matrix work is bf16-fast;
//   * a bf16 product rounds dz to bf16 anyway, so here the rounded dz tile itself is all-gathered (bf16_common.h:
//     4 granules per lane and layer) and every workgroup computes ITS 32 units of each product from the whole
//     (16 x 1024) tile: wave w contracts gate w's 256 columns, the four partial tiles meet in LDS (fixed order).
//     dx_t = dz1 . K1^T (O <= 8 outputs) is computed redundantly by every workgroup from the same tile: the third
//     exchange of the fp32 kernel is gone.
// Weights: R2^T, K2^T, R1^T slices as packed B fragments (3 x 64 registers per lane, eight contiguous floats of a weight
// row each), K1^T 8 fragments.  Per step and wave 56 MFMAs of 16 cycles; the step is bound by its two all-gathers.
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

namespace {

constexpr int DRS = 33;   // row stride of a partial (16 x 32) tile in LDS

template <int ACT>
__device__ __forceinline__ float db_act_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

template <int ACT>
__global__ __launch_bounds__(256, 1) void mix_decoder_bwd_bf16_kernel(MixDecBwdParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short sDZ[QBT * QLDZ];   // the whole dz tile of the current layer, bf16
    __shared__ float sRedA[4 * QBT * DRS];       // [wave][row][unit] partial of product A (R2^T / R1^T)
    __shared__ float sRedB[4 * QBT * DRS];       // partial of product B (K2^T)
    __shared__ float sRedX[4 * QBT * 17];        // [wave][row][o] partial of dx
    __shared__ float sDP[QBT * 8];               // dpre_p of the step
    __shared__ float sDX[QBT * 8];               // dx_{t+1}
    __shared__ int sFlag[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int O = p.O, T = p.T_out;
    int group, slice;
    if (!q_group_slice(p.num_groups, group, slice)) { q_spare_leaves(p.status); return; }
    constexpr int H4 = 4 * QH;
    // pointwise phases (round 4, as lstm_bwd8n_bf16_kernel): thread tid owns row tid >> 4 of the tile and the two adjacent units
    // 2 * (tid & 15), + 1 of the workgroup's 32 - 8-byte tape accesses, whole 128-byte lines per wave instruction
    const int prow = tid >> 4;
    const int pu = 2 * (tid & 15);
    const int unit0 = 32 * slice + pu;
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) { sFlag[0] = poisoned ? 1 : 0; sFlag[1] = 0; sFlag[2] = 0; }

    // ---- resident transposed weights: k-block kb of gate `wave` (columns 256*wave + 32*kb + 8*g4 + j), N-tile nt = own
    // unit 16*nt + n: eight contiguous floats of row (32*slice + 16*nt + n) of R2 / K2 / R1; K1^T: row n (< O) of K1 ----
    qu32x4 r2q[8][2], k2q[8][2], r1q[8][2], k1q[8];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const size_t o = (size_t)(32 * slice + 16 * nt + n) * H4 + QH * wave + 32 * kb + 8 * g4;
            r2q[kb][nt] = load_bfrag_rowmajor(p.R2 + o);
            k2q[kb][nt] = load_bfrag_rowmajor(p.K2p + o);   // K2p: here the plain (H,4H) kernel of layer 2
            r1q[kb][nt] = load_bfrag_rowmajor(p.R1 + o);
        }
        k1q[kb] = load_bfrag_rowmajor(p.K1 + (size_t)(n < O ? n : 0) * H4 + QH * wave + 32 * kb + 8 * g4);
        if (n >= O) k1q[kb] = (qu32x4){0u, 0u, 0u, 0u};
    }
    float wd[2][8];   // Dense kernel rows of this lane's two units
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int o = 0; o < 8; ++o) wd[u][o] = (o < O) ? p.Wd[(size_t)(unit0 + u) * O + o] : 0.f;
    const int hrow = tid >> 4, ho = tid & 15;   // head: thread = (sequence of the tile, output)
    float wpr[8];   // row `ho` of Wp: dp[ho] = sum_k dm[k] Wp[ho][k]
#pragma unroll
    for (int k = 0; k < 8; ++k) wpr[k] = (ho < O && k < O) ? p.Wp[ho * O + k] : 0.f;

    // ---- exchange areas of this group: [layer 2][parity 2] dz tiles ----
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 4 * (Q_DZ_BYTES / 8), 0, (int)(4 * Q_DZ_BYTES), 0x00020000);
    constexpr unsigned LAYER_BYTES = 2u * Q_DZ_BYTES;
    xch_hello_poll(p.status, sXch, group, QG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * QBT;
        __syncthreads();
        if (tid < QBT * 8) sDX[tid] = 0.f;       // no feedback gradient into the last step
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        f32x2 dc1 = {0.f, 0.f}, dc2 = {0.f, 0.f}, dh1r = {0.f, 0.f}, dh2r = {0.f, 0.f};
        __syncthreads();
        const int row = b0 + prow;
        const bool lv = row < p.B;
        const size_t rowc = (size_t)(lv ? row : p.B - 1);
        // Tapes (reserve i,f,g,o,c and the previous cell state) of this lane's two cells, both layers, one step AHEAD: the
        // loads of step t-1 are issued at the top of step t.  Vector-memory operations retire in order - HBM loads issued
        // right in front of a gather would put their whole latency into the gather's wait.
        f32x2 tq2[6], tq1[6];
        // UNCONDITIONAL loads (step and row clamped into the tape; dead rows are masked where dz is formed): a load inside
        // a branch gets an s_waitcnt vmcnt(0) at the merge and the step would wait for HBM right there
        auto load_tape = [&](int t, const float* res, const float* Cst, f32x2 (&dst)[6]) {
            const size_t tc = (size_t)(t > 0 ? t : 0);
            const float* rp = res + ((tc * p.B + rowc) * 5) * QH + unit0;
#pragma unroll
            for (int k = 0; k < 5; ++k) dst[k] = *(const f32x2*)(rp + k * QH);
            dst[5] = *(const f32x2*)(Cst + (tc * p.B + rowc) * QH + unit0);
        };
        load_tape(T - 1, p.res2, p.C2, tq2);
        load_tape(T - 1, p.res1, p.C1, tq1);
        for (int t = T - 1; t >= 0; --t) {
            ++epoch;
            const unsigned par = (epoch & 1u) * Q_DZ_BYTES;
            f32x2 tp[6], tp1[6];
#pragma unroll
            for (int k = 0; k < 6; ++k) { tp[k] = tq2[k]; tp1[k] = tq1[k]; }
            load_tape(t - 1, p.res2, p.C2, tq2);
            load_tape(t - 1, p.res1, p.C1, tq1);
            // ================= head backward (every workgroup, its 16 sequences) =================
            {
                const int brow = b0 + hrow;
                const bool live = ho < O && brow < p.B;
                const size_t hidx = ((size_t)t * p.B + brow) * O + ho;
                const size_t hidc = live ? hidx : 0;     // dead lanes read element 0, masked below (no load inside a branch)
                const float mv = p.M[hidc], pv = p.P[hidc], dl = p.dloss[hidc];
                const float dm = live ? dl + sDX[hrow * 8 + (ho & 7)] * (1.f - mv * mv) : 0.f;
                float dp = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp = fmaf(__shfl(dm, (lane & ~15) | k), wpr[k], dp);
                const float dpp = dp * (1.f - pv * pv);
                if (ho < 8) sDP[hrow * 8 + ho] = live ? dpp : 0.f;
                if (slice == 0 && live) {
                    p.dpre_m[hidx] = dm;
                    p.dpre_p[hidx] = dpp;
                }
            }
            __syncthreads();   // barrier 1: dpre_p of the step is in LDS; every wave is past the previous step's LDS reads
            // ================= layer 2: gates backward for this lane's two cells, publish dz2 =================
            {
                f32x2 dz[4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    float dhd = 0.f;
#pragma unroll
                    for (int o = 0; o < 8; ++o) dhd = fmaf(sDP[prow * 8 + o], wd[u][o], dhd);
                    const float ig = tp[0][u], fg = tp[1][u], gg = tp[2][u], og = tp[3][u], ct = tp[4][u], cp = tp[5][u];
                    const float tc = tanh_f(ct);
                    const float dh = dhd + dh2r[u];
                    const float dct = dc2[u] + dh * og * (1.f - tc * tc);
                    dz[0][u] = lv ? dct * gg * db_act_grad<ACT>(ig) : 0.f;
                    dz[1][u] = lv ? dct * cp * db_act_grad<ACT>(fg) : 0.f;
                    dz[2][u] = lv ? dct * ig * (1.f - gg * gg) : 0.f;
                    dz[3][u] = lv ? dh * tc * db_act_grad<ACT>(og) : 0.f;
                    dc2[u] = dct * fg;
                }
                unsigned dzp[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) dzp[g] = pack_bf16(dz[g][0], dz[g][1]);
                q_dz_publish2(rs, LAYER_BYTES + par, prow, unit0, dzp, epoch, sDZ, ticket.same_xcd);
                if (lv) {
                    float* zp = p.DZ2 + ((size_t)t * p.B + row) * H4 + unit0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *(f32x2*)(zp + g * QH) = dz[g];
                }
            }
            if (!q_dz_gather2(rs, LAYER_BYTES + par, slice, tid, epoch, sDZ, p.status)) sFlag[1] = 1;
            __syncthreads();   // barrier 2: the whole dz2 tile is in LDS
            if (sFlag[1]) { aborted = true; break; }
            // ================= dh2_{t-1} and dh1_t of the own 32 units: this wave's gate =================
            {
                f32x4 a2[2], ak[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) { a2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ak[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
                qu32x4 a[8];
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) a[kb] = *(const qu32x4*)(sDZ + n * QLDZ + QH * wave + 32 * kb + 8 * g4);
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    qmfma(a2[0], a[kb], r2q[kb][0]);
                    qmfma(a2[1], a[kb], r2q[kb][1]);
                    qmfma(ak[0], a[kb], k2q[kb][0]);
                    qmfma(ak[1], a[kb], k2q[kb][1]);
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        sRedA[(wave * QBT + 4 * g4 + r) * DRS + 16 * nt + n] = a2[nt][r];
                        sRedB[(wave * QBT + 4 * g4 + r) * DRS + 16 * nt + n] = ak[nt][r];
                    }
            }
            __syncthreads();   // barrier 3: partial tiles in LDS; every wave is done reading the dz2 tile
            f32x2 dh1in;
            {
                const float* qa = sRedA + prow * DRS + pu;
                const float* qb = sRedB + prow * DRS + pu;
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    dh2r[u] = (qa[u] + qa[QBT * DRS + u]) + (qa[2 * QBT * DRS + u] + qa[3 * QBT * DRS + u]);
                    dh1in[u] = (qb[u] + qb[QBT * DRS + u]) + (qb[2 * QBT * DRS + u] + qb[3 * QBT * DRS + u]);
                }
            }
            // ================= layer 1: gates backward, publish dz1 =================
            {
                f32x2 dz[4];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const float ig = tp1[0][u], fg = tp1[1][u], gg = tp1[2][u], og = tp1[3][u], ct = tp1[4][u], cp = tp1[5][u];
                    const float tc = tanh_f(ct);
                    const float dh = dh1in[u] + dh1r[u];
                    const float dct = dc1[u] + dh * og * (1.f - tc * tc);
                    dz[0][u] = lv ? dct * gg * db_act_grad<ACT>(ig) : 0.f;
                    dz[1][u] = lv ? dct * cp * db_act_grad<ACT>(fg) : 0.f;
                    dz[2][u] = lv ? dct * ig * (1.f - gg * gg) : 0.f;
                    dz[3][u] = lv ? dh * tc * db_act_grad<ACT>(og) : 0.f;
                    dc1[u] = dct * fg;
                }
                unsigned dzp[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) dzp[g] = pack_bf16(dz[g][0], dz[g][1]);
                q_dz_publish2(rs, par, prow, unit0, dzp, epoch, sDZ, ticket.same_xcd);
                if (lv) {
                    float* zp = p.DZ1 + ((size_t)t * p.B + row) * H4 + unit0;
#pragma unroll
                    for (int g = 0; g < 4; ++g) *(f32x2*)(zp + g * QH) = dz[g];
                }
            }
            if (!q_dz_gather2(rs, par, slice, tid, epoch, sDZ, p.status)) sFlag[2] = 1;
            __syncthreads();   // barrier 4: the whole dz1 tile is in LDS
            if (sFlag[2]) { aborted = true; break; }
            // ================= dh1_{t-1} of the own 32 units and dx_t (all O outputs, every workgroup) =================
            {
                f32x4 a1[2], ax;
                a1[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                a1[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                ax = (f32x4){0.f, 0.f, 0.f, 0.f};
                qu32x4 a[8];
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) a[kb] = *(const qu32x4*)(sDZ + n * QLDZ + QH * wave + 32 * kb + 8 * g4);
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    qmfma(a1[0], a[kb], r1q[kb][0]);
                    qmfma(a1[1], a[kb], r1q[kb][1]);
                    qmfma(ax, a[kb], k1q[kb]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sRedA[(wave * QBT + 4 * g4 + r) * DRS + n] = a1[0][r];
                    sRedA[(wave * QBT + 4 * g4 + r) * DRS + 16 + n] = a1[1][r];
                    sRedX[(wave * QBT + 4 * g4 + r) * 17 + n] = ax[r];
                }
            }
            __syncthreads();   // barrier 5: partial tiles in LDS; every wave is done with the dz1 tile and with sDX of this step
            {
                const float* qa = sRedA + prow * DRS + pu;
#pragma unroll
                for (int u = 0; u < 2; ++u) dh1r[u] = (qa[u] + qa[QBT * DRS + u]) + (qa[2 * QBT * DRS + u] + qa[3 * QBT * DRS + u]);
            }
            if (ho < 8) {
                const float* qx = sRedX + hrow * 17 + ho;
                sDX[hrow * 8 + ho] = (qx[0] + qx[QBT * 17]) + (qx[2 * QBT * 17] + qx[3 * QBT * 17]);
            }
            // The next step's head reads sDX behind ITS barrier 1?  No: it reads sDX before that barrier, so dx_t needs
            // its own barrier here; the same barrier orders these reads of sRedA / sRedX before their next writes.
            __syncthreads();   // barrier 6: dx_t is in LDS
        }
        if (!aborted && lv) {
            *(f32x2*)(p.dh1_0 + (size_t)row * QH + unit0) = dh1r;
            *(f32x2*)(p.dc1_0 + (size_t)row * QH + unit0) = dc1;
            *(f32x2*)(p.dh2_0 + (size_t)row * QH + unit0) = dh2r;
            *(f32x2*)(p.dc2_0 + (size_t)row * QH + unit0) = dc2;
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

}  // namespace

int mix_decoder_bwd_bf16_launch(MixDecBwdParams p, const float* K2, int act, void* workspace, hipStream_t stream) {
    if (p.B == 0 || p.T_out == 0) return FOV_OK;
    p.num_tiles = (p.B + QBT - 1) / QBT;
    const int max_groups = device_cu_count() / QG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("fused mixing decoder backward needs at least %d CUs", QG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 4 * Q_DZ_BYTES > kXchBytes - kHelloBytes) { set_error("mix_decoder_bwd_bf16: granule area too small"); return FOV_ERR_WORKSPACE; }
    if ((((uintptr_t)K2) | ((uintptr_t)p.R1) | ((uintptr_t)p.R2) | ((uintptr_t)p.K1)) & 15) {
        set_error("mix_decoder_bwd_bf16: kernels must be 16-byte aligned");
        return FOV_ERR_INVALID;
    }
    if ((((uintptr_t)p.res1) | ((uintptr_t)p.res2) | ((uintptr_t)p.C1) | ((uintptr_t)p.C2) | ((uintptr_t)p.DZ1) | ((uintptr_t)p.DZ2) |
         ((uintptr_t)p.dh1_0) | ((uintptr_t)p.dc1_0) | ((uintptr_t)p.dh2_0) | ((uintptr_t)p.dc2_0)) & 7) {   // 8-byte tape accesses
        set_error("mix_decoder_bwd_bf16: tapes and state gradients must be 8-byte aligned");
        return FOV_ERR_INVALID;
    }
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    p.K2p = K2;   // no packed copy: fragments are eight contiguous floats of a row of the plain (H,4H) kernel
    p.epoch_span = p.T_out * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(MixDecBwdParams) = act == FOV_ACT_HARD_SIGMOID ? mix_decoder_bwd_bf16_kernel<FOV_ACT_HARD_SIGMOID> : mix_decoder_bwd_bf16_kernel<FOV_ACT_SIGMOID>;
    hipLaunchKernelGGL(kern, dim3(q_padded_groups(p.num_groups) * QG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder_bwd_bf16 launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
