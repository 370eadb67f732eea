// Encoder + autoregressive decoder of the seq2seq call (mycode/FoV_seq2seq.py:83-97 + the decode loop :154-178) at H = 256 for
// batches that give every group of workgroups TWO 16-sequence tiles (round 5; the metric's B = 1024 on 256 CUs).
//
// Why.  lstm_cluster.hip gives a tile to four workgroups; per step a workgroup issues its partner-slice MFMAs, updates its cells,
// publishes its slice of h_t and then has only the 64 own-slice MFMAs (0.9 us) to cover the 1.5 us until the partners' slices
// have arrived: in-kernel stamps (tools/stamp_profile.py) show the matrix pipe idle for 12 % of a decoder step and 9 % of an
// encoder step, and nothing inside ONE tile's step can fill that - every other MFMA of the step needs the gathered h_t.  The
// time can only be filled with ANOTHER tile's work.  Here a group is EIGHT workgroups (32 hidden units x 4 gates each: the
// ownership scheme of lstm_wide.hip - a wave owns 8 units as two MFMA N-tiles [i | f], [g | o], DPP half swap before the cell
// update) and carries two tiles, A and B, in a fixed software pipeline:
//
//     block A(t):  wait for h_A(t-1)'s gather -> LDS -> barrier;  request h_B's gather;  z_A += h_A(t-1) . R  (128 MFMAs);
//                  cell update A;  publish h_A(t);  z_A(t+1) = b + x_A(t+1) . K  (48 MFMAs, encoder)
//     block B(t):  the same for B - while A's granules travel
//
// so each tile's exchange has the other tile's whole block (>= 128 MFMAs = 4096 cycles) to complete, and the weights - R slice
// 128 accumulation registers, encoder K slice 48 - serve both tiles.  64 tiles = 32 groups x 8 workgroups = 256 CUs: the same
// sequences per CU and the same MFMA count per CU and step as the four-workgroup kernel, without the idle time.
//
// Decoder phase (weights swapped in place through the LDS staging of stage_f32.h): y_{t-1} = tanh(Dense(h_{t-1})) is formed by
// every workgroup from the gathered tile - wave w reduces over units [64 w, 64 w + 64), the four partial products meet in LDS
// behind the block's second barrier, which the 128 MFMAs of h . R precede - and enters z_t as y . K (F_dec <= 8: two k-steps).
// h tiles are double-buffered by step parity (own columns of h_t are written while other waves may still read h_{t-1}), the
// granule protocol is xch_common.h's: {value, epoch} tags monotone across launches, two parity buffers per tile, bounded
// spins, sticky timeout word.  Gate order i,f,c,o; weights in Keras layout; exact expf / tanhf as everywhere.
#include <stdlib.h>

#include "fov_common.h"
#include "stage_f32.h"
#include "xch_common.h"

namespace fov {

namespace {

constexpr int PBT = 16;            // sequences per tile
constexpr int PH = 256;            // hidden width
constexpr int PG = 8;              // workgroups per group
constexpr int PLD = PH + 4;        // LDS row stride of an h tile
constexpr int PLX = 96 + 4;        // LDS row stride of an x tile (F <= 96)
constexpr int PNJR = PH / 16;      // k-blocks of R
constexpr int PNJX = 6;            // k-blocks of the encoder's K (F <= 96)
constexpr int PNG = PG - 1;        // 16-byte gather loads per thread, tile and step
constexpr unsigned PSPIN = 1u << 20;
constexpr unsigned PPARITY = PBT * PH * 8u;   // bytes of one parity buffer of one tile's granules

typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));
typedef float pf32x2 __attribute__((ext_vector_type(2)));

// LDS layout (floats): weight staging | h tiles [tile][parity][16][PLD] | x tiles [tile][buffer][16][PLX] | Dense partials
// [tile][wave][16][16] | flags
constexpr int P_OFF_H = FST_LDS_WORDS;
constexpr int P_OFF_X = P_OFF_H + 2 * 2 * PBT * PLD;
constexpr int P_OFF_W = P_OFF_X + 2 * 2 * PBT * PLX;
constexpr int P_OFF_FLAG = P_OFF_W + 2 * 4 * 16 * 16;
constexpr int P_LDS_FLOATS = P_OFF_FLAG + 16;

// Diagnostic build only (-DFOV_STAMPS, tools/stamp_pair.py): s_memtime stamps of one wave (block 5, wave 0) per block of the pipeline
#ifdef FOV_STAMPS
constexpr int PST_STEPS = 128, PST_SLOTS = 10;
__device__ unsigned long long g_pair_stamps[PST_STEPS][2][PST_SLOTS];
#define PSTAMP(slot)                                                                            \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (stamp_on && step < PST_STEPS) {                                                     \
            unsigned long long t_;                                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
            g_pair_stamps[step][i][slot] = t_;                                                  \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#else
#define PSTAMP(slot) do { } while (0)
#endif

template <bool W_AGPR>
__device__ __forceinline__ void pm_a(f32x4& acc, float a, float w) {
    if constexpr (W_AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w));
    else asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w));
}
// wait states around a run of MFMAs on two accumulators (hipcc pads no hazards around inline asm: lstm_cluster.hip)
__device__ __forceinline__ void pm_begin(f32x4 (&acc)[2]) { asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void pm_end(f32x4 (&acc)[2]) { asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ float pswap(float v) {     // lanes n and n ^ 8 of a row of 16 exchange a value
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));
}
// acc[tile] += A(tile rows in LDS, k-blocks [0, NJ)) . W (accumulation-register resident)
template <int NJ, int J0 = 0, int J1 = NJ>
__device__ __forceinline__ void pair_mm(f32x4 (&acc)[2], const float* arow, const float (&w)[NJ][4][2]) {
#ifdef FOV_PAIR_ACC2
    f32x4 a = *(const f32x4*)(arow + 16 * J0);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        f32x4 an = a;
        if (j + 1 < J1) an = *(const f32x4*)(arow + 16 * (j + 1));
        asm volatile("s_nop 1" : "+v"(a));   // the fragment may have been moved by the compiler (VALU copy)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            pm_a<true>(acc[0], a[s], w[j][s][0]);
            pm_a<true>(acc[1], a[s], w[j][s][1]);
        }
        a = an;
    }
#else
    // FOUR accumulators (odd k-steps into a second pair, added at the end): with two, every MFMA reads the accumulator the MFMA
    // two slots earlier wrote, and the run issued at 38-40 cycles per MFMA instead of 32 (tools/stamp_pair.py)
    f32x4 odd[2];
    odd[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
    odd[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
    asm volatile("s_nop 3" : "+v"(odd[0]), "+v"(odd[1]));
    f32x4 a = *(const f32x4*)(arow + 16 * J0);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        f32x4 an = a;
        if (j + 1 < J1) an = *(const f32x4*)(arow + 16 * (j + 1));
        asm volatile("s_nop 1" : "+v"(a));   // the fragment may have been moved by the compiler (VALU copy)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (s & 1) {
                pm_a<true>(odd[0], a[s], w[j][s][0]);
                pm_a<true>(odd[1], a[s], w[j][s][1]);
            } else {
                pm_a<true>(acc[0], a[s], w[j][s][0]);
                pm_a<true>(acc[1], a[s], w[j][s][1]);
            }
        }
        a = an;
    }
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(odd[0]), "+v"(odd[1]), "+v"(acc[0]), "+v"(acc[1]));
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc[0][r] += odd[0][r]; acc[1][r] += odd[1][r]; }
#endif
}

template <int ACT>
__global__ __launch_bounds__(256, 1) void lstm_pair_s2s_kernel(LstmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    unsigned* sStage = (unsigned*)smem;
    float* sH = smem + P_OFF_H;
    float* sX = smem + P_OFF_X;
    float* sW = smem + P_OFF_W;
    int* sFlag = (int*)(smem + P_OFF_FLAG);
    __shared__ unsigned sXch[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    // members 8 blocks apart on a grid padded to a multiple of eight groups: one XCD (xch_padded_groups, xch_common.h)
    const int group = (blockIdx.x / (8 * PG)) * 8 + (blockIdx.x & 7), slice = (blockIdx.x >> 3) & (PG - 1);
    if (group >= p.num_groups) { xch_spare_leaves(p.status, true); return; }
    const int F = p.F, O = p.F_dec, T = p.T, T_out = p.T_out;
    const int unit = 32 * slice + 8 * wave + (n & 7);
    const int hi = n >> 3;
    const int col0 = hi * PH + unit, col1 = (2 + hi) * PH + unit;
    constexpr int H4 = 4 * PH;
    constexpr unsigned OORB = 0x80000000u;   // buffer-load offset no descriptor covers: reads as 0

    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- encoder weights through LDS (stage_f32.h); K rows >= F read as zero ----
    float wk[PNJX][4][2], wr[PNJR][4][2];
    stage_weight_sets_f32<PH>(wk, p.K, F, wr, p.R, PH, slice, sStage, []() {});
    float bv[2] = {p.b[col0], p.b[col1]};
    for (int i = tid; i < 2 * 2 * PBT * PLX; i += 256) sX[i] = 0.f;   // columns >= F stay zero
    for (int i = tid; i < 2 * 2 * PBT * PLD; i += 256) sH[i] = 0.f;   // h_{-1} = 0

    // ---- exchange bookkeeping: the group's granule area is [tile][parity][row pair][unit][row of the pair] ----
    const int my_row0 = 4 * g4 + 2 * hi;
    const unsigned pub_off = (unsigned)((my_row0 >> 1) * PH + unit) * 16u;
    const unsigned gvoff = (unsigned)((tid >> 5) * PH + (tid & 31)) * 16u;
    const int lbase = 2 * (tid >> 5) * PLD + (tid & 31);
    __amdgpu_buffer_rsrc_t xrs[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
        xrs[i] = __builtin_amdgcn_make_buffer_rsrc(p.xch + ((size_t)group * 2 + i) * 2 * PBT * PH, 0, 2 * PBT * PH * (int)sizeof(unsigned long long), 0x00020000);
    xch_hello_poll(p.status, sXch, group, PG, &sFlag[0]);
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    unsigned epoch[2] = {ticket.base, ticket.base};   // tag of the tile's LAST publish
    bool pending[2] = {false, false};                 // published, gather not yet requested
    bool inflight[2] = {false, false};                // gather requested, not yet in LDS
    pu32x4 v[2][PNG];
    auto gather_issue = [&](int i) __attribute__((always_inline)) {
        const unsigned base = (epoch[i] & 1u) * PPARITY;
#pragma unroll
        for (int j = 0; j < PNG; ++j) {
            const unsigned uo = (unsigned)(((slice + 1 + j) & (PG - 1)) * 32) * 16u;
            v[i][j] = __builtin_amdgcn_raw_buffer_load_b128(xrs[i], gvoff, base + uo, 16);
        }
    };
    // current granules go straight to the h tile, stale ones into a bit mask; retry sweeps (rare) re-read into temporaries
    auto gather_finish = [&](int i, float* sHt) __attribute__((always_inline)) {
        const unsigned ep = epoch[i];
        const unsigned base = (ep & 1u) * PPARITY;
        unsigned bad = 0;
#pragma unroll
        for (int j = 0; j < PNG; ++j) {
            const int lo = lbase + ((slice + 1 + j) & (PG - 1)) * 32;
            if (v[i][j].y == ep && v[i][j].w == ep) {
                sHt[lo] = __uint_as_float(v[i][j].x);
                sHt[lo + PLD] = __uint_as_float(v[i][j].z);
            } else {
                bad |= (1u << j);
            }
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > PSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) {
                    xch_give_up(p.status);
                    sFlag[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            pu32x4 tv[PNG];
#pragma unroll
            for (int j = 0; j < PNG; ++j) {
                const unsigned uo = (unsigned)(((slice + 1 + j) & (PG - 1)) * 32) * 16u;
                tv[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs[i], gvoff, base + uo, 16);
            }
#pragma unroll
            for (int j = 0; j < PNG; ++j) {
                const int lo = lbase + ((slice + 1 + j) & (PG - 1)) * 32;
                if (((bad >> j) & 1u) && tv[j].y == ep && tv[j].w == ep) {
                    sHt[lo] = __uint_as_float(tv[j].x);
                    sHt[lo + PLD] = __uint_as_float(tv[j].z);
                    bad &= ~(1u << j);
                }
            }
        }
    };

    // ---- per-tile state ----
    const int pair0 = 2 * group;                      // tiles 2 g and 2 g + 1 (the second may be absent: its rows are masked)
    int b0[2], live[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        b0[i] = (pair0 + i) * PBT;
        const int left = p.B - b0[i];
        live[i] = left < 0 ? 0 : (left < PBT ? left : PBT);
    }
    float c[2][2] = {{0.f, 0.f}, {0.f, 0.f}}, hc[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    f32x4 acc[2][2];
    // x staging: thread (xrw = tid / 16, xc = tid % 16) moves the elements xc, xc + 16, ..., xc + 80 of row xrw of each tile
    const int xrw = tid >> 4, xc = tid & 15;
    __amdgpu_buffer_rsrc_t xgrs[2];
    unsigned xoff[PNJX];
#pragma unroll
    for (int k = 0; k < PNJX; ++k) xoff[k] = (xc + 16 * k < F) ? (unsigned)((xrw * T * F + xc + 16 * k) * 4) : OORB;
#pragma unroll
    for (int i = 0; i < 2; ++i)
        xgrs[i] = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(live[i] > 0 ? p.x + (size_t)b0[i] * T * F : nullptr), 0,
                                                    live[i] * T * F * 4, 0x00020000);
    float xs[2][PNJX];
    float* xl = sX + xrw * PLX + xc;                  // + (tile * 2 + buffer) * PBT * PLX + 16 k
    __syncthreads();                                  // the zero fills above
    {   // x_0, x_1 of both tiles -> LDS
        float x2[2][2][PNJX];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int k = 0; k < PNJX; ++k)
                    x2[i][tt][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs[i], tt < T ? xoff[k] : OORB, (unsigned)(tt * F * 4), 0));
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int k = 0; k < PNJX; ++k)
                    if (xc + 16 * k < F) xl[(i * 2 + tt) * PBT * PLX + 16 * k] = x2[i][tt][k];
    }
    __syncthreads();
    // z_0 = b + x_0 . K of both tiles (h_{-1} = 0)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        acc[i][0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
        acc[i][1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
        if (T > 0) {
            pm_begin(acc[i]);
            pair_mm<PNJX>(acc[i], sX + (i * 2) * PBT * PLX + n * PLX + 4 * g4, wk);
            pm_end(acc[i]);
        }
    }

    // cell update of one tile from its accumulators (lstm_wide.hip's: the lane pair n, n ^ 8 swaps half of its values, each
    // lane then owns two rows of one unit with all four gates)
    auto cell = [&](int i) __attribute__((always_inline)) {
        float snd[4], rcv[4];
        snd[0] = hi ? acc[i][0][0] : acc[i][0][2];
        snd[1] = hi ? acc[i][0][1] : acc[i][0][3];
        snd[2] = hi ? acc[i][1][0] : acc[i][1][2];
        snd[3] = hi ? acc[i][1][1] : acc[i][1][3];
#pragma unroll
        for (int k = 0; k < 4; ++k) rcv[k] = pswap(snd[k]);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float zi = hi ? rcv[r] : acc[i][0][r];
            const float zf = hi ? acc[i][0][2 + r] : rcv[r];
            const float zg = hi ? rcv[2 + r] : acc[i][1][r];
            const float zo = hi ? acc[i][1][2 + r] : rcv[2 + r];
            const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
            c[i][r] = fmaf(fg, c[i][r], ig * gg);
            hc[i][r] = og * tanh_f(c[i][r]);
        }
    };
    // publish the tile's new h (tag ++epoch) and put the own columns into the h tile of parity `par`
    auto publish = [&](int i, int par) __attribute__((always_inline)) {
        ++epoch[i];
        const unsigned base = (epoch[i] & 1u) * PPARITY;
        const pu32x4 gr = {__float_as_uint(hc[i][0]), epoch[i], __float_as_uint(hc[i][1]), epoch[i]};
        if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, xrs[i], pub_off, base, 1 /* sc0: stays in the XCD's L2 */);
        else __builtin_amdgcn_raw_buffer_store_b128(gr, xrs[i], pub_off, base, 16 /* sc1: write-through */);
        float* sHt = sH + (i * 2 + par) * PBT * PLD;
#pragma unroll
        for (int r = 0; r < 2; ++r) sHt[(my_row0 + r) * PLD + unit] = hc[i][r];
        pending[i] = true;
    };

#ifdef FOV_STAMPS
    const bool stamp_on = (blockIdx.x == 5 && tid == 0);
#endif
    // =========================== encoder phase ===========================
    int step = 0;        // steps taken so far (both phases): h_t of step `step` lives in parity step & 1
    for (int t = 0; t < T && !aborted; ++t, ++step) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            PSTAMP(0);
            // x_{t+1} (requested during the tile's previous block) registers -> LDS; then request x_{t+2}
            if (t > 0 && t + 1 < T) {
                float* xb = xl + (i * 2 + ((t + 1) & 1)) * PBT * PLX;
#pragma unroll
                for (int k = 0; k < PNJX; ++k)
                    if (xc + 16 * k < F) xb[16 * k] = xs[i][k];
            }
            if (t + 2 < T) {
#pragma unroll
                for (int k = 0; k < PNJX; ++k)
                    xs[i][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs[i], xoff[k], (unsigned)((t + 2) * F * 4), 0));
            }
            float* sHprev = sH + (i * 2 + ((step + 1) & 1)) * PBT * PLD;     // h_{t-1}: parity (step - 1) & 1
            PSTAMP(1);
            if (inflight[i]) { gather_finish(i, sHprev); inflight[i] = false; }
            PSTAMP(2);
            __syncthreads();      // the tile of h_{t-1} (own columns since the tile's last block, partners' just now) and x_{t+1} are in LDS
            PSTAMP(3);
            if (sFlag[0]) { aborted = true; break; }
            // The other tile's gather is requested half way through this block's h . R: its partners published at about the moment this
            // workgroup did - one x . K run and a barrier ago - and a store takes most of a microsecond to become visible; a sweep that
            // comes back stale is only noticed a whole block later, where its retry is a fully exposed round trip (measured: requested
            // at the top of the block the pair-step took 7.4 us instead of the 5.9 us of two four-workgroup steps).
            if (t > 0) {
                pm_begin(acc[i]);
                pair_mm<PNJR, 0, PNJR / 2>(acc[i], sHprev + n * PLD + 4 * g4, wr);
            }
            if (pending[i ^ 1]) { gather_issue(i ^ 1); pending[i ^ 1] = false; inflight[i ^ 1] = true; }
            if (t > 0) {
                pair_mm<PNJR, PNJR / 2, PNJR>(acc[i], sHprev + n * PLD + 4 * g4, wr);
                pm_end(acc[i]);
            }
            PSTAMP(4);
            cell(i);
            PSTAMP(5);
            publish(i, step & 1);      // (the last step's h is exchanged too: the decoder phase starts from the whole tile)
            PSTAMP(6);
            acc[i][0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
            acc[i][1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
            if (t + 1 < T) {           // z_{t+1} = b + x_{t+1} . K needs no remote data
                pm_begin(acc[i]);
                pair_mm<PNJX>(acc[i], sX + (i * 2 + ((t + 1) & 1)) * PBT * PLX + n * PLX + 4 * g4, wk);
                pm_end(acc[i]);
            }
            PSTAMP(7);
        }
    }

    // =========================== decoder phase ===========================
    // weights swapped in place: R slice through the staging buffers (two barriers inside: every wave has left the encoder
    // loop), the two k-steps of K (F_dec <= 8: input row k = 4 s + g4, the layout the Dense's D fragment has), bias, Dense
    float wkd[2][2], wd[4][4], bd4[2];
    {
        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dK), 0, O * H4 * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            wkd[s][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(krs, (unsigned)(((4 * s + g4) * H4 + col0) * 4), 0, 0));   // rows >= F_dec: past the descriptor
            wkd[s][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(krs, (unsigned)(((4 * s + g4) * H4 + col1) * 4), 0, 0));
        }
        const __amdgpu_buffer_rsrc_t wdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dW), 0, PH * O * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t bdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dbias), 0, O * 4, 0x00020000);
        // Dense on the matrix pipe, transposed: y^T = Wd^T . h^T.  Wave w reduces over units [64 w, 64 w + 64); lane (i = l & 15, g4)
        // keeps Wd[64 w + 16 b + 4 g4 + s][o(i)], o(i) = 4 (i & 3) + (i >> 2) (zero where o(i) >= F_dec), so that register r of
        // the D fragment on lane (n, g4) is y[n][4 r + g4]: registers 0, 1 are the A operands of the two y . K k-steps
        const int o = 4 * (n & 3) + (n >> 2);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                wd[b][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wdrs, (o < O) ? (unsigned)(((64 * wave + 16 * b + 4 * g4 + s) * O + o) * 4) : OORB, 0, 0));
#pragma unroll
        for (int s = 0; s < 2; ++s)
            bd4[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bdrs, (4 * s + g4 < O) ? (unsigned)((4 * s + g4) * 4) : OORB, 0, 0));
        bv[0] = p.db[col0];
        bv[1] = p.db[col1];
    }
    stage_weight_set_f32<PH>(wr, p.dR, PH, slice, sStage, []() {});
    // y_{-1}: the caller's first decoder input, in the A-operand layout y[n][4 s + g4]
    float y4[2][2];
    __amdgpu_buffer_rsrc_t yors[2];
    unsigned yoff[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) yoff[s] = (4 * s + g4 < O) ? (unsigned)((n * T_out * O + 4 * s + g4) * 4) : OORB;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const __amdgpu_buffer_rsrc_t y0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(live[i] > 0 ? p.dec_in0 + (size_t)b0[i] * O : nullptr), 0, live[i] * O * 4, 0x00020000);
#pragma unroll
        for (int s = 0; s < 2; ++s)
            y4[i][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(y0rs, (4 * s + g4 < O) ? (unsigned)((n * O + 4 * s + g4) * 4) : OORB, 0, 0));
        yors[i] = __builtin_amdgcn_make_buffer_rsrc((live[i] > 0 && p.out) ? p.out + (size_t)b0[i] * T_out * O : nullptr, 0,
                                                    (live[i] > 0 && p.out) ? live[i] * T_out * O * 4 : 0, 0x00020000);
    }
    // this wave's quarter of the Dense over the gathered tile -> its partial in LDS (visible behind the block's second barrier)
    auto dense_partial = [&](int i, const float* sHt) __attribute__((always_inline)) {
        const float* hq = sHt + n * PLD + 4 * g4 + 64 * wave;
        f32x4 hb[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) hb[b] = *(const f32x4*)(hq + 16 * b);
        f32x4 dacc[2];
        dacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
        dacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
        pm_begin(dacc);
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int s = 0; s < 4; ++s) pm_a<false>(dacc[s & 1], wd[b][s], hb[b][s]);
        pm_end(dacc);
#pragma unroll
        for (int s = 0; s < 4; ++s) dacc[0][s] += dacc[1][s];
        *(f32x4*)(sW + ((i * 4 + wave) * 16 + n) * 16 + 4 * g4) = dacc[0];
    };
    auto dense_sum = [&](int i, float (&y)[2]) __attribute__((always_inline)) {
        pf32x2 part[4];
#pragma unroll
        for (int w2 = 0; w2 < 4; ++w2) part[w2] = *(const pf32x2*)(sW + ((i * 4 + w2) * 16 + n) * 16 + 4 * g4);
#pragma unroll
        for (int s = 0; s < 2; ++s) y[s] = tanh_f(((part[0][s] + part[1][s]) + (part[2][s] + part[3][s])) + bd4[s]);
    };

    for (int t = 0; t < T_out && !aborted; ++t, ++step) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* sHprev = sH + (i * 2 + ((step + 1) & 1)) * PBT * PLD;     // h_{t-1} (t = 0: the encoder's h_T, or zeros when T = 0)
            if (inflight[i]) { gather_finish(i, sHprev); inflight[i] = false; }
            __syncthreads();
            if (sFlag[0]) { aborted = true; break; }
            if (t > 0) dense_partial(i, sHprev);
            acc[i][0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
            acc[i][1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
            if (step > 0) {
                pm_begin(acc[i]);
                pair_mm<PNJR, 0, PNJR / 2>(acc[i], sHprev + n * PLD + 4 * g4, wr);
            }
            if (pending[i ^ 1]) { gather_issue(i ^ 1); pending[i ^ 1] = false; inflight[i ^ 1] = true; }   // (see the encoder loop)
            if (step > 0) {
                pair_mm<PNJR, PNJR / 2, PNJR>(acc[i], sHprev + n * PLD + 4 * g4, wr);
                pm_end(acc[i]);
            }
            if (t > 0) {
                __syncthreads();      // the four waves' Dense partials are in LDS
                dense_sum(i, y4[i]);
                if (slice == 0 && wave == 0) {
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y4[i][s]), yors[i], yoff[s], (unsigned)((t - 1) * O * 4), 0);
                }
            }
            asm volatile("s_nop 1" : "+v"(y4[i][0]), "+v"(y4[i][1]));      // VALU-written A operands: two wait states in front of the MFMAs
            pm_begin(acc[i]);
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                pm_a<false>(acc[i][0], y4[i][s], wkd[s][0]);
                pm_a<false>(acc[i][1], y4[i][s], wkd[s][1]);
            }
            pm_end(acc[i]);
            cell(i);
            publish(i, step & 1);
        }
    }
    // y of the last step: one more gather + Dense per tile, no cell behind it
    if (T_out > 0 && !aborted) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            float* sHprev = sH + (i * 2 + ((step + 1) & 1)) * PBT * PLD;
            if (pending[i]) { gather_issue(i); pending[i] = false; inflight[i] = true; }
            if (inflight[i]) { gather_finish(i, sHprev); inflight[i] = false; }
        }
        __syncthreads();
        if (!sFlag[0]) {
#pragma unroll
            for (int i = 0; i < 2; ++i) dense_partial(i, sH + (i * 2 + ((step + 1) & 1)) * PBT * PLD);
            __syncthreads();
            if (slice == 0 && wave == 0) {
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    float y[2];
                    dense_sum(i, y);
#pragma unroll
                    for (int s = 0; s < 2; ++s)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y[s]), yors[i], yoff[s], (unsigned)((T_out - 1) * O * 4), 0);
                }
            }
        } else {
            aborted = true;
        }
    }
    if (!aborted) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0[i] + my_row0 + r;
                if (row < p.B) {
                    if (p.hT) p.hT[(size_t)row * PH + unit] = hc[i][r];
                    if (p.cT) p.cT[(size_t)row * PH + unit] = c[i][r];
                }
            }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

}  // namespace

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_pair_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stamps), sizeof(unsigned long long) * PST_STEPS * 2 * PST_SLOTS);
}
#endif

// One pair of tiles per group, every group resident: 33 .. 2 * (CUs / 8) tiles (below that the four-workgroup kernel does not
// fill the chip either and the sixteen-unit / eight-workgroup forms take over; above it the four-workgroup kernel's tile loop).
bool pair_s2s_shape(int B, int T_in, int T_out, int F_enc, int F_dec, int H) {
    if (!env_knobs().pair || H != PH || F_enc < 1 || F_enc > 96 || F_dec < 1 || F_dec > 8 || T_in < 1 || T_out < 1) return false;
    const int tiles = (B + PBT - 1) / PBT;
    const int groups = (tiles + 1) / 2;
    return tiles > 32 && xch_padded_groups(groups) * PG <= device_cu_count() &&
           (size_t)groups * 4 * PBT * PH * sizeof(unsigned long long) <= kXchBytes - kHelloBytes;
}

int launch_pair_s2s(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0) return FOV_OK;
    if (!pair_s2s_shape(p.B, p.T, p.T_out, p.F, p.F_dec, p.H)) { set_error("seq2seq decode (tile pairs): unsupported shape"); return FOV_ERR_UNSUPPORTED; }
    p.num_tiles = (p.B + PBT - 1) / PBT;
    p.num_groups = (p.num_tiles + 1) / 2;
    p.epoch_span = p.T + p.T_out + 2;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(LstmParams) = p.act == FOV_ACT_HARD_SIGMOID ? lstm_pair_s2s_kernel<FOV_ACT_HARD_SIGMOID> : lstm_pair_s2s_kernel<FOV_ACT_SIGMOID>;
    const size_t lds = sizeof(float) * P_LDS_FLOATS;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(xch_padded_groups(p.num_groups) * PG), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("seq2seq decode (tile pairs) launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
