// Two stacked LSTM layers (bf16 matrix-core operands, H = 256) in ONE launch, as a WAVEFRONT over (layer, step)
// (round 3; BASELINE.json configs[4]: the 2-layer encoder of mycode/given_others_gt_mean_var_seq2seq.py:108-112).
//
// lstm_layer_bf16.hip runs a layer as a chain of T exchange-bound steps (2.6-3.5 us each for 0.15 us of bf16 MFMAs); the
// model's two encoder layers were two such launches back to back.  A bf16 layer workgroup needs < 256 registers and 26 KB
// of LDS, so TWO of them fit a CU: here the grid is 2 x (groups x 8) workgroups - the first half runs layer 1 exactly as
// before, the second half layer 2 ONE STEP BEHIND it - and the two chains overlap: T + 1 steps instead of 2 T.
//   * layer 1 (producer) publishes h1_t into slot t of a RING of T granule tiles (instead of two parity buffers) and also
//     publishes its last step; its partners gather from the same slot;
//   * layer 2 (consumer) has no x in memory: its input tile x_t = bf16(h1_t) IS the producer's granule tile t - it gathers
//     all eight slices of slot t (the epoch tag of a granule is its ready flag) next to its own seven-slice exchange of
//     h2_{t-1}; both land in LDS between the two barriers of a step, the 32 MFMAs of x_t.K2 + h2_{t-1}.R2 follow;
//     the bf16 values are the ones lstm_layer_bf16_kernel forms from the fp32 h1 tape: results are bit-identical to the
//     two-launch form;
//   * producers are dispatched first and never wait for a consumer, and a ring slot is written once per launch: if the
//     consumers are not co-resident the launch is merely slower, never wrong.  One tile per group (<= 512 sequences).
// Zero initial state (the encoder's); every tensor the two-launch form writes is written here (hs, reserve, hT, cT).
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

struct Stack2Params {
    const float* x;                     // (B,T,F) layer-1 input, F <= 96
    const float *K1, *R1, *b1, *K2, *R2, *b2;
    float *hs1, *hT1, *cT1, *res1;      // any may be NULL
    float *hs2, *hT2, *cT2, *res2;
    int B, T, F, act;
    unsigned long long* xch;
    unsigned* status;
    int num_groups, epoch_span;
};

namespace {

constexpr int S2_NKB1 = 3;   // layer 1: F <= 96
constexpr int S2_STAGE_DEPTH = 2;   // weight staging: stages in flight ahead (two workgroups per CU: 256 registers per lane)

// load_weight_set of bf16_common.h for a kernel whose rows ALL exist (32 NKB rows): the row part that does not depend on the
// lane goes into the scalar offset, so the loads share TWO address registers (hipcc precomputed one per load - 256 of them
// for a layer - and spilled, each spill behind its own vmcnt(0)).
template <int NKB, class Behind = QNothing>
__device__ __forceinline__ void load_weight_set_full(qu32x4 (&w)[NKB][2], const float* __restrict__ W, int ld, int g4, int col0, int col1,
                                                     Behind behind_first_batch = Behind()) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, 32 * NKB * ld * 4, 0x00020000);
    const unsigned v0 = (unsigned)((8 * g4) * ld + col0) * 4u, v1 = (unsigned)((8 * g4) * ld + col1) * 4u;
#pragma unroll
    for (int c = 0; c < NKB; c += 2) {
        float v[2][2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned so = (unsigned)((32 * (c + i) + j) * ld) * 4u;   // wave-uniform
                v[i][0][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v0, so, 0));
                v[i][1][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, v1, so, 0));
            }
        if (c == 0) behind_first_batch();
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                w[c + i][t] = (qu32x4){pack_bf16(v[i][t][0], v[i][t][1]), pack_bf16(v[i][t][2], v[i][t][3]),
                                       pack_bf16(v[i][t][4], v[i][t][5]), pack_bf16(v[i][t][6], v[i][t][7])};
        // the packed fragments must exist HERE and no later load may move above this point: the batch's 32 raw values die
        // before the next batch is requested (a scheduling barrier alone let the packing sink and 29 values spill)
        asm volatile("" :: "v"(w[c][0]), "v"(w[c][1]), "v"(w[c + 1][0]), "v"(w[c + 1][1]) : "memory");
    }
}

// qmm of bf16_common.h with at most FOUR A fragments in flight (16 registers instead of 32): two workgroups share a CU's
// register file here, the kernel is held to 256 registers per lane
template <int NKB>
__device__ __forceinline__ void qmm4(f32x4 (&acc)[2], const unsigned short* tile, int n, int g4, const qu32x4 (&w)[NKB][2]) {
#pragma unroll
    for (int k0 = 0; k0 < NKB; k0 += 4) {
        qu32x4 a[4];
#pragma unroll
        for (int kb = k0; kb < k0 + 4 && kb < NKB; ++kb) a[kb - k0] = lds_afrag(tile, n, g4, kb);
#pragma unroll
        for (int kb = k0; kb < k0 + 4 && kb < NKB; ++kb) {
            qmfma(acc[0], a[kb - k0], w[kb][0]);
            qmfma(acc[1], a[kb - k0], w[kb][1]);
        }
    }
}

// one layer of the pair.  ROLE 0: producer (x from memory, ring publish), ROLE 1: consumer (x from the producer's ring)
#ifdef FOV_STAMPS
// Diagnostic build only (make stamps): thread 0 of slice 0 of group 0 of each role stamps s_memtime at the phase boundaries of
// every step into LDS (branch-free: every thread reads the clock and writes a junk slot unless it is the stamping one) and
// copies the table out once at the end -> tools/stamp_bf16_layer.py --stack2
constexpr int S2STAMP_SLOTS = 12;
constexpr int S2STAMP_STEPS = 32;
__device__ unsigned long long g_s2_stamps[2][S2STAMP_STEPS][S2STAMP_SLOTS];
#define S2_STAMP(slot)                                                                         \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        sStamps[(stamp_on && t < S2STAMP_STEPS - 2) ? t * S2STAMP_SLOTS + slot : (S2STAMP_STEPS - 1) * S2STAMP_SLOTS + 11] = t_; \
    } while (0)
#else
#define S2_STAMP(slot) do { } while (0)
#endif

template <int ACT, int ROLE>
__device__ __forceinline__ void stack2_body(const Stack2Params& p, unsigned short* sH, unsigned short* sX, unsigned* sStage, int* sFlag, unsigned* sXch,
                                            int group, int slice) {
    constexpr int NKB = ROLE == 0 ? S2_NKB1 : 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int F = ROLE == 0 ? p.F : QH, steps = p.T;
    const int unit = 32 * slice + 8 * wave + (n & 7);
    const int hi = n >> 3;
    const int col0 = hi * QH + unit, col1 = (2 + hi) * QH + unit;
    constexpr int H4 = 4 * QH;
    const float* Kp = ROLE == 0 ? p.K1 : p.K2;
    const float* Rp = ROLE == 0 ? p.R1 : p.R2;
    const float* bp = ROLE == 0 ? p.b1 : p.b2;
    float* hs = ROLE == 0 ? p.hs1 : p.hs2;
    float* hT = ROLE == 0 ? p.hT1 : p.hT2;
    float* cT = ROLE == 0 ? p.cT1 : p.cT2;
    float* reserve = ROLE == 0 ? p.res1 : p.res2;

#ifdef FOV_STAMPS
    __shared__ unsigned long long sStamps[S2STAMP_STEPS * S2STAMP_SLOTS];
    const bool stamp_on = (group == 0 && slice == 0 && tid == 0);
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 1][0] = __builtin_amdgcn_s_memtime();   // kernel entry
#endif
    const XchHeader header = xch_arrive_request(p.status);   // taken behind the first weight set (xch_common.h)
    const unsigned timeout_word = xch_timeout_word(p.status);
    const unsigned arrival = 0u;
    qu32x4 wk[NKB][2], wr[8][2];
    auto commit = [&]() {
        xch_arrive_commit(p.status, sXch, header, ROLE * p.num_groups + group, slice);
        // zero initial state; x columns >= F stay zero (under the first stage's round trip)
        for (int i = tid; i < QBT * QLD; i += 256) { sX[i] = 0; sH[i] = 0; }
    };
    // both sets through LDS (bf16_common.h); rows of K1 >= F read as zero (hardware bounds)
    stage_weight_sets<S2_STAGE_DEPTH>(wk, Kp, ROLE == 0 ? F : QH, wr, Rp, QH, H4, slice, sStage, commit);
    const bool poisoned = xch_timeout_set(timeout_word);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 2][0] = __builtin_amdgcn_s_memtime();   // first weight set packed, arrival counted, hello word out
#endif
    const float bv[2] = {bp[col0], bp[col1]};
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 2][1] = __builtin_amdgcn_s_memtime();   // weights and bias requested (and packed)
#endif
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 2][2] = __builtin_amdgcn_s_memtime();   // LDS zeroed
#endif

    // granule areas: the producer ring [group][step][tile], behind it the consumers' parity buffers [group][2][tile]
    const unsigned ring_group_bytes = (unsigned)steps * Q_TILE_BYTES;
    const __amdgpu_buffer_rsrc_t ring = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * (ring_group_bytes / 8), 0, (int)ring_group_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t own = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)p.num_groups * (ring_group_bytes / 8) + (size_t)group * 2 * (Q_TILE_BYTES / 8), 0, (int)(2 * Q_TILE_BYTES),
        0x00020000);
    const int my_row0 = 4 * g4 + 2 * hi;
    const unsigned pub_off = (unsigned)((my_row0 >> 1) * QH + unit) * 8u;
    const int b0 = group * QBT;      // one tile per group
    const int live_rows = p.B - b0 < QBT ? p.B - b0 : QBT;
    constexpr unsigned OORB = 0x80000000u;
    // ---- x of step 0 (and 1) ----
    const int xrw = tid >> 4, xc = tid & 15;
    constexpr int NXE = 2 * S2_NKB1;
    unsigned xoff[NXE];
    const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(ROLE == 0 ? p.x + (size_t)b0 * p.T * F : nullptr), 0, ROLE == 0 ? live_rows * p.T * F * 4 : 0, 0x00020000);
    auto load_x1 = [&](int i, int t) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, xoff[i], (unsigned)(t * F * 4), 0)); };
    unsigned short* xl = sX + xrw * QLD + xc;
    float xs[NXE];
    // the lower layer requests x_0 and x_1 here, in front of the handshake and the prologue's barrier: their round trip used
    // to start behind both (1 600 cycles between the barrier and x_0 in LDS, which the upper layer's first step waits for too)
    float v1[NXE];
    if constexpr (ROLE == 0) {
#pragma unroll
        for (int i = 0; i < NXE; ++i) xoff[i] = (xc + 16 * i < F) ? (unsigned)((xrw * p.T * F + xc + 16 * i) * 4) : OORB;
#pragma unroll
        for (int i = 0; i < NXE; ++i) v1[i] = load_x1(i, 0);
#pragma unroll
        for (int i = 0; i < NXE; ++i) xs[i] = load_x1(i, steps > 1 ? 1 : 0);
    }
    xch_hello_poll(p.status, sXch, ROLE * p.num_groups + group, QG, &sFlag[0]);
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 2][3] = __builtin_amdgcn_s_memtime();   // every member's hello word seen
#endif
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 2][4] = __builtin_amdgcn_s_memtime();   // first barrier passed
#endif
    const unsigned base = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);
    // epoch tags: producer step t -> base + 1 + t; consumer's own exchange of step t -> base + 1 + steps + t

    // all eight slices of the producer's tile `slot` into the x image (consumer); returns false after a give-up
    // (the first sweep is REQUESTED by gather_x_issue together with the own exchange's gather, so the two round trips overlap)
    // thread (row pair tid / 32, half (tid / 16) % 2, unit pair tid % 16): units (2p, 2p + 1) of slices 4 half + j, one 16-byte load each
    qu32x4 xg[4];
    const unsigned xvoff = (unsigned)((tid >> 5) * QH + 2 * (tid & 15)) * 8u;
    const int xsl0 = ((tid >> 4) & 1) * 4;
    auto gather_x_issue = [&](int slot) {
        const unsigned sbase = (unsigned)slot * Q_TILE_BYTES;
#pragma unroll
        for (int j = 0; j < 4; ++j) xg[j] = __builtin_amdgcn_raw_buffer_load_b128(ring, xvoff + (unsigned)((xsl0 + j) * 32) * 8u, sbase, 16);
    };
    auto gather_x = [&](int slot, bool issued) {
        const unsigned tag = base + 1u + (unsigned)slot;
        const unsigned sbase = (unsigned)slot * Q_TILE_BYTES;
        const int lbase = (tid >> 5) * 2 * QLD + 2 * (tid & 15);
        unsigned bad = 0xfu, spins = 0;
        bool first = issued;
        while (true) {
            qu32x4 tv[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) tv[j] = first ? xg[j] : __builtin_amdgcn_raw_buffer_load_b128(ring, xvoff + (unsigned)((xsl0 + j) * 32) * 8u, sbase, 16);
            first = false;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (((bad >> j) & 1u) && tv[j].y == tag && tv[j].w == tag) {
                    q_tile_put(sX, lbase + (xsl0 + j) * 32, QLD, tv[j].x, tv[j].z);
                    bad &= ~(1u << j);
                }
            if (!__any(bad != 0)) return true;
            if (++spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if ((tid & 63) == 0) xch_give_up(p.status);
                return false;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
        }
    };
    if constexpr (ROLE == 0) {
#pragma unroll
        for (int i = 0; i < NXE; ++i)
            if (xc + 16 * i < F) xl[16 * i] = bf16_bits(v1[i]);
    } else {
        if (!aborted && steps > 0 && !gather_x(0, false)) sFlag[0] = 1;
    }
    __syncthreads();
    if (sFlag[0]) aborted = true;
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 2][5] = __builtin_amdgcn_s_memtime();   // x_0 in LDS (upper role: lower's h_0 gathered)
#endif
    float c[2] = {0.f, 0.f}, hc[2] = {0.f, 0.f};
    f32x4 acc[2];
    acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
    acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
    if (steps > 0 && !aborted) qmm4<NKB>(acc, sX, n, g4, wk);   // h_{-1} = 0: no recurrent term
    QGather gq;
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 1][1] = __builtin_amdgcn_s_memtime();   // weights resident, x_0 . K done
#endif
    for (int t = 0; t < steps && !aborted; ++t) {
        const bool more = (t + 1 < steps);
        S2_STAMP(0);
        // ---- cell update (fp32) ----
        float gt[2][4];
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
        S2_STAMP(1);
        const unsigned hpair = pack_bf16(hc[0], hc[1]);
        unsigned epoch, goff;
        if constexpr (ROLE == 0) {   // ring slot t; the LAST step is published too (the consumer's x_{T-1})
            epoch = base + 1u + (unsigned)t;
            goff = (unsigned)t * Q_TILE_BYTES;
            XCH_STORE_B64(ticket.same_xcd, ((qu32x2){hpair, epoch}), ring, pub_off, goff);
        } else {
            epoch = base + 1u + (unsigned)steps + (unsigned)t;
            goff = (epoch & 1u) * Q_TILE_BYTES;
            if (more) XCH_STORE_B64(ticket.same_xcd, ((qu32x2){hpair, epoch}), own, pub_off, goff);
        }
        S2_STAMP(2);
        __syncthreads();   // barrier 1: every wave is done reading sH and sX
        S2_STAMP(3);
        if (more) {
            sH[my_row0 * QLD + unit] = (unsigned short)(hpair & 0xffffu);
            sH[(my_row0 + 1) * QLD + unit] = (unsigned short)(hpair >> 16);
        }
        if constexpr (ROLE == 0) {
            // x_{t+1}: registers -> LDS (single image: its readers passed barrier 1), then request x_{t+2}
            if (more) {
#pragma unroll
                for (int i = 0; i < NXE; ++i)
                    if (xc + 16 * i < F) xl[16 * i] = bf16_bits(xs[i]);
            }
#pragma unroll
            for (int i = 0; i < NXE; ++i) xs[i] = load_x1(i, t + 2 < steps ? t + 2 : t);
        }
        if (more) q_gather_issue(gq, ROLE == 0 ? ring : own, goff, slice, tid);
        if constexpr (ROLE == 1) {
            if (more) gather_x_issue(t + 1);
        }
        S2_STAMP(4);
#pragma unroll
        for (int r = 0; r < 2; ++r) {   // tape of the step, under the gather's round trip
            const int row = b0 + my_row0 + r;
            if (row < p.B) {
                if (reserve) {
                    float* rp = reserve + (((size_t)row * p.T + t) * 5) * QH + unit;
                    rp[0] = gt[r][0]; rp[QH] = gt[r][1]; rp[2 * QH] = gt[r][2]; rp[3 * QH] = gt[r][3]; rp[4 * QH] = c[r];
                }
                if (hs) hs[((size_t)row * p.T + t) * QH + unit] = hc[r];
            }
        }
        S2_STAMP(5);
        if (more) {
            if (!q_gather_finish(gq, ROLE == 0 ? ring : own, goff, slice, tid, epoch, sH, p.status)) sFlag[0] = 1;
        }
        S2_STAMP(6);
        if constexpr (ROLE == 1) {
            if (more && !gather_x(t + 1, true)) sFlag[0] = 1;     // the producer's tile t + 1 (it runs one step ahead)
        }
        S2_STAMP(7);
        __syncthreads();   // barrier 2: h_t and x_{t+1} are in LDS
        S2_STAMP(8);
        if (sFlag[0]) { aborted = true; break; }
        acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
        acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
        if (more) {
            qmm4<NKB>(acc, sX, n, g4, wk);
            qmm4<8>(acc, sH, n, g4, wr);
        }
        S2_STAMP(9);
    }
#ifdef FOV_STAMPS
    if (stamp_on) {
        g_s2_stamps[ROLE][S2STAMP_STEPS - 1][2] = __builtin_amdgcn_s_memtime();   // recurrence done
        for (int i = 0; i < (S2STAMP_STEPS - 2) * S2STAMP_SLOTS; ++i) (&g_s2_stamps[ROLE][0][0])[i] = sStamps[i];
    }
#endif
    if (!aborted) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = b0 + my_row0 + r;
            if (row < p.B) {
                if (hT) hT[(size_t)row * QH + unit] = hc[r];
                if (cT) cT[(size_t)row * QH + unit] = c[r];
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
#ifdef FOV_STAMPS
    if (stamp_on) g_s2_stamps[ROLE][S2STAMP_STEPS - 1][3] = __builtin_amdgcn_s_memtime();   // left
#endif
}

template <int ACT>
__global__ __launch_bounds__(256, 2) void lstm_stack2_bf16_kernel(Stack2Params p) {
    __shared__ __attribute__((aligned(16))) unsigned short sH[QBT * QLD];
    __shared__ __attribute__((aligned(16))) unsigned short sX[QBT * QLD];
    __shared__ __attribute__((aligned(16))) unsigned sStage[QST_LDS_WORDS];   // the prologue's weight staging (bf16_common.h)
    __shared__ int sFlag[4];
    __shared__ unsigned sXch[4];
    // each role's grid is padded to a multiple of eight groups: members 8 blocks apart, one XCD (q_group_slice, bf16_common.h)
    const int per_role = q_padded_groups(p.num_groups) * QG;
    const int role = (int)blockIdx.x >= per_role ? 1 : 0;     // producers first: they never wait for a consumer
    const int local = (int)blockIdx.x - role * per_role;
    const int group = (local / (8 * QG)) * 8 + (local & 7), slice = (local >> 3) & (QG - 1);
    if (group >= p.num_groups) { q_spare_leaves(p.status); return; }
    if (role == 0) stack2_body<ACT, 0>(p, sH, sX, sStage, sFlag, sXch, group, slice);
    else stack2_body<ACT, 1>(p, sH, sX, sStage, sFlag, sXch, group, slice);
}

}  // namespace

// F <= 96 into H = 256 twice, zero initial state, one tile per group, at least two steps, and room for the ring
bool stack2_bf16_shape_ok(int B, int T, int F, int H) {
    if (H != QH || F < 1 || F > 96 || B < 1 || T < 2) return false;
    const int tiles = (B + QBT - 1) / QBT;
    if (tiles > device_cu_count() / QG) return false;
    return (size_t)tiles * ((size_t)T + 2) * Q_TILE_BYTES <= kXchBytes - kHelloBytes && 2 * tiles <= 256;
}

int launch_stack2_bf16(const float* x, const float* K1, const float* R1, const float* b1, const float* K2, const float* R2,
                       const float* b2, float* hs1, float* hT1, float* cT1, float* res1, float* hs2, float* hT2, float* cT2,
                       float* res2, int B, int T, int F, int act, void* workspace, hipStream_t stream) {
    Stack2Params p = {};
    p.x = x; p.K1 = K1; p.R1 = R1; p.b1 = b1; p.K2 = K2; p.R2 = R2; p.b2 = b2;
    p.hs1 = hs1; p.hT1 = hT1; p.cT1 = cT1; p.res1 = res1; p.hs2 = hs2; p.hT2 = hT2; p.cT2 = cT2; p.res2 = res2;
    p.B = B; p.T = T; p.F = F; p.act = act;
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    p.num_groups = (B + QBT - 1) / QBT;
    p.epoch_span = 2 * T + 2;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(Stack2Params) = act == FOV_ACT_HARD_SIGMOID ? lstm_stack2_bf16_kernel<FOV_ACT_HARD_SIGMOID>
                                                             : lstm_stack2_bf16_kernel<FOV_ACT_SIGMOID>;
    hipLaunchKernelGGL(kern, dim3(2 * q_padded_groups(p.num_groups) * QG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("bf16 two-layer launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_s2_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(fov::g_s2_stamps), sizeof(unsigned long long) * 2 * fov::S2STAMP_STEPS * fov::S2STAMP_SLOTS);
}
#endif
