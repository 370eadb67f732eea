// Fused decoder of the others-mixing model (a4): the whole unrolled no-teacher-forcing loop of
// mycode/given_others_gt_mean_var_seq2seq.py:203-299 in ONE persistent launch.
//
//   per step t:  h1,c1 = LSTM1(x_t; h1,c1)          K1:(O,4H)  R1:(H,4H)
//                h2,c2 = LSTM2(h1;  h2,c2)          K2:(H,4H)  R2:(H,4H)
//                p = tanh(h2 Wd + bd)               Dense(O,'tanh'), :127-130
//                m = tanh(p Wp + others_t . W_oth + b_mix)      mixing Dense, :257-265 (others part hoisted)
//                x_{t+1} = m                        :292-293
//
// The step-wise path pays ~45 us per decoder step in launches, handshakes and weight reloads for ~6 us
// of arithmetic.  Here H = 256 and a tile of 16 sequences is owned by a GROUP of 8 workgroups; workgroup
// `slice` owns hidden units [32*slice, +32) of BOTH layers, wave w 8 of them.  A wave's gate columns form
// two MFMA N-tiles, [i | f] and [g | o] (8 units each); after the MFMAs the two halves of a 16-lane row
// swap what the other needs (DPP row_ror:8), so that every lane ends up with all four gates of one unit for
// two sequences: the cell update is lane-local and c1, c2 never leave registers.
//   R1, R2 slices : 128 + 128 AGPRs per lane for the whole launch (MFMA B operands)
//   K2 slice      : 128 KB per workgroup; 7/8 of it sits in LDS as lane-linear B-operand fragments (one
//                   ds_read_b128 = four MFMA steps), the last eighth in 16 VGPRs - all of it would need the whole
//                   LDS next to the two h tiles.  Loaded once from a copy of K2 pre-packed in fragment order
//                   (mix_decoder_pack_k2).  (Streaming it from L2 every step, four loads in flight, was
//                   latency-bound: 19 us per step.)
//   h1, h2 tiles  : 16 x 256 each in LDS (MFMA A operands), exchanged once per layer and step as 8-byte
//                   {value, epoch} granules (sc1 stores / sc1 loads, two parity buffers, bounded spins) - the
//                   protocol of lstm_cluster.hip
//   overlap       : the gather of h1_t runs under h2_{t-1} . R2, the gather of h2_t under h1_t . R1 of the
//                   next step
//   head          : every workgroup computes p and m of its 16 sequences from the gathered h2 tile (O = 6:
//                   cheaper than a third exchange) - Dense on the matrix pipe, K split over the four waves -
//                   slice 0 stores them
// TRAIN additionally stores what the backward pass reads: reserves (i,f,g,o,c) of both layers, h1/h2/c1/c2
// of every step, p.
#include <stdlib.h>

#include "fov_common.h"
#include "xch_common.h"
#include "stage_f32.h"

namespace fov {

namespace {

constexpr int MH = 256;          // hidden units
constexpr int MG = 8;            // workgroups per tile
constexpr int MBT = 16;          // sequences per tile
constexpr int MLDH = MH + 4;     // LDS row stride of an h tile
constexpr int MNG = 7;           // 16-byte loads (TWO adjacent units' tagged granules) per thread and exchange: 7 slices * 16 rows * 16 pairs / 256
constexpr unsigned M_SPIN_LIMIT = 1u << 20;

typedef unsigned mu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned mu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned mu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void mfma_a(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}
__device__ __forceinline__ void mfma_v(f32x4& acc, float a, float w_vgpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w_vgpr));
}
__device__ __forceinline__ void mfma_begin2(f32x4 (&acc)[2]) { asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void mfma_end2(f32x4 (&acc)[2]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}
// hipcc pads nothing around inline asm and may place a VALU write of an MFMA operand (a select, a copy made under
// register pressure) right before the asm MFMA that reads it: pin such operands and spend the two wait states
__device__ __forceinline__ void mfma_guard2(float& a, float& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mfma_guard4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
// value of the lane 8 positions away inside the same 16-lane row (row_ror:8)
__device__ __forceinline__ float swap_half(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));
}

}  // namespace

// Diagnostic build only (-DFOV_STAMPS, tools/stamp_mix_decoder.py): s_memtime stamps of one wave per step.
#ifdef FOV_STAMPS
constexpr int MSTAMP_SLOTS = 12;
constexpr int MSTAMP_STEPS = 32;
__device__ unsigned long long g_mix_stamps[MSTAMP_STEPS][MSTAMP_SLOTS];
#define MIX_STAMP(slot)                                                                        \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (stamp_on && t < MSTAMP_STEPS) {                                                    \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            g_mix_stamps[t][slot] = t_;                                                        \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define MIX_STAMP(slot) do { } while (0)
#endif

// acc[tile] += A(h tile rows, LDS) . W (AGPR resident, [16 k-blocks][4][2 tiles])
template <int J0 = 0, int J1 = 16>
__device__ __forceinline__ void recur_agpr(f32x4 (&acc)[2], const float* hrow, const float (&w)[16][4][2]) {
    if (J0 >= J1) return;
    f32x4 a = *(const f32x4*)(hrow + 16 * J0);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        f32x4 an = a;
        if (j + 1 < J1) an = *(const f32x4*)(hrow + 16 * (j + 1));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            mfma_a(acc[0], a[s], w[j][s][0]);
            mfma_a(acc[1], a[s], w[j][s][1]);
        }
        a = an;
    }
}

// acc[tile] += A(h tile rows, LDS) . K2 slice: fragment blocks (j, tile), j < 14, from LDS (this wave's region,
// lane-linear: one ds_read_b128 = the four k-subs of a block), blocks of j = 14, 15 from registers.
constexpr int K2_LDS_BLOCKS = 28;
constexpr int GATHER_AFTER_BLOCK = 8;   // k-blocks of the covering MFMA product issued before the gather is requested
__device__ __forceinline__ void recur_k2(f32x4 (&acc)[2], const float* hrow, const float* sK2l, const f32x4 (&kr)[4]) {
    f32x4 a = *(const f32x4*)hrow;
    f32x4 b0 = *(const f32x4*)sK2l, b1 = *(const f32x4*)(sK2l + 256);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x4 an = a, n0 = b0, n1 = b1;
        if (j + 1 < 16) an = *(const f32x4*)(hrow + 16 * (j + 1));
        if (2 * (j + 1) < K2_LDS_BLOCKS) {
            n0 = *(const f32x4*)(sK2l + (2 * j + 2) * 256);
            n1 = *(const f32x4*)(sK2l + (2 * j + 3) * 256);
        }
        const f32x4 c0 = (2 * j < K2_LDS_BLOCKS) ? b0 : kr[2 * j - K2_LDS_BLOCKS];
        const f32x4 c1 = (2 * j < K2_LDS_BLOCKS) ? b1 : kr[2 * j + 1 - K2_LDS_BLOCKS];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            mfma_v(acc[0], a[s], c0[s]);
            mfma_v(acc[1], a[s], c1[s]);
        }
        a = an; b0 = n0; b1 = n1;
    }
}

template <int ACT, bool TRAIN>
__global__ __launch_bounds__(256, 1) void mix_decoder_kernel(MixDecParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sH1 = smem;                       // [16][MLDH]
    float* sH2 = sH1 + MBT * MLDH;           // [16][MLDH]
    float* sX = sH2 + MBT * MLDH;            // [16][8]  decoder input x_t (columns >= O are zero)
    float* sWd = sX + MBT * 8;               // [256][8] Dense kernel, rows padded to 8
    float* sWp = sWd + MH * 8;               // [8][8]   mixing kernel (pred part)
    int* sFlag = (int*)(sWp + 64);
    float* sPart = sWp + 64 + 16;            // [4 waves][16 rows][16 cols] partial Dense products
    float* sK2 = sPart + 4 * 256;            // [4 waves][28 blocks][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int O = p.O;
    // members of a group 8 blocks apart on a grid padded to a multiple of eight groups: one XCD under round-robin dispatch
    // (xch_padded_groups, xch_common.h); only a placement preference, verified by the hello handshake
    const int group = (blockIdx.x / (8 * MG)) * 8 + (blockIdx.x & 7), slice = (blockIdx.x >> 3) & (MG - 1);
    if (group >= p.num_groups) { xch_spare_leaves(p.status); return; }
    const int unit = 32 * slice + 8 * wave + (n & 7);   // hidden unit of this lane's columns
    const int hi = n >> 3;                               // 0: columns i / g, 1: columns f / o
    const int col0 = hi * MH + unit, col1 = (2 + hi) * MH + unit;   // gate columns of tile 0 / tile 1
    const int H4 = 4 * MH;

    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;
    // ---- resident weights ----
    // both (H, 4H) recurrent kernels through LDS (stage_f32.h; the staging buffers are the K2 blocks' area, filled after it)
    float w1[16][4][2], w2[16][4][2];
    static_assert(4 * K2_LDS_BLOCKS * 256 >= FST_LDS_WORDS, "the K2 area holds the two staging buffers");
    stage_weight_sets_f32<MH>(w1, p.R1, MH, w2, p.R2, MH, slice, (unsigned*)sK2, []() {});
    // K1 (O <= 8 rows): MFMA step s uses input row k = 4*s + g4
    float k1[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const int k = 4 * s + g4;
        k1[s][0] = (k < O) ? p.K1[(size_t)k * H4 + col0] : 0.f;
        k1[s][1] = (k < O) ? p.K1[(size_t)k * H4 + col1] : 0.f;
    }
    const float b1v[2] = {p.b1[col0], p.b1[col1]}, b2v[2] = {p.b2[col0], p.b2[col1]};
    const float bdv = ((tid & 15) < O) ? p.bd[tid & 15] : 0.f;   // Dense bias of this thread's head output
    for (int e = tid; e < MH * 8; e += 256) {
        const int k = e >> 3, o = e & 7;
        sWd[e] = (o < O) ? p.Wd[(size_t)k * O + o] : 0.f;
    }
    for (int e = tid; e < 64; e += 256) sWp[e] = ((e >> 3) < O && (e & 7) < O) ? p.Wp[(e >> 3) * O + (e & 7)] : 0.f;
    // this wave's 32 KB of packed K2 fragments (block b at byte b*1024, lane fragment at lane*16): blocks 0..27
    // into LDS, 28..31 into registers
    float* sK2l = sK2 + (size_t)wave * K2_LDS_BLOCKS * 256 + lane * 4;
    f32x4 k2r[4];
    {
        const f32x4* kp = (const f32x4*)p.K2p + (size_t)(slice * 4 + wave) * 32 * 64 + lane;
        for (int b = 0; b < K2_LDS_BLOCKS; ++b) *(f32x4*)(sK2l + b * 256) = kp[64 * b];
#pragma unroll
        for (int b = 0; b < 4; ++b) k2r[b] = kp[64 * (K2_LDS_BLOCKS + b)];
    }
    // ---- exchange bookkeeping ----
    unsigned long long* xg = p.xch + (size_t)group * 4 * MBT * MH;   // [layer][parity][16][256]
    const __amdgpu_buffer_rsrc_t xrs =
        __builtin_amdgcn_make_buffer_rsrc(xg, 0, 4 * MBT * MH * (int)sizeof(unsigned long long), 0x00020000);
    const int my_row0 = 4 * g4 + 2 * hi;   // this lane's two cells: rows my_row0, my_row0 + 1 of unit `unit`
    // granule order [row pair][unit][row of the pair]: the lane's two rows leave as ONE 16-byte store, a gather load brings
    // both rows of a unit (two tagged granules)
    const unsigned pub_off = (unsigned)((my_row0 >> 1) * MH + unit) * 16u;
    // gather: load j of this thread brings rows (2k, 2k + 1), k = tid / 32, of unit tid % 32 of the other slice (slice + 1 + j) mod 8 -
    // seven 16-byte loads; one per-thread offset, everything else wave-uniform
    const unsigned gvoff = (unsigned)((tid >> 5) * MH + (tid & 31)) * 16u;
    const int lbase = 2 * (tid >> 5) * MLDH + (tid & 31);
    constexpr unsigned LAYER_BYTES = 2u * MBT * MH * 8u;    // both parities of one layer
    constexpr unsigned PARITY_BYTES = MBT * MH * 8u;
#ifdef FOV_STAMPS
    const bool stamp_on = (blockIdx.x == 5 && tid == 0);
#endif
    xch_hello_poll(p.status, sXch, group, MG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    // gather: issue / complete.  v[] stays in registers between the two so MFMAs can run in between.
    mu32x4 v[MNG];
    auto gather_issue = [&](unsigned base) {
#pragma unroll
        for (int j = 0; j < MNG; ++j) {
            const unsigned uo = (unsigned)(((slice + 1 + j) & (MG - 1)) * 32) * 16u;
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, gvoff, base + uo, 16);
        }
    };
    auto gather_finish = [&](unsigned base, float* sH) {
        unsigned spins = 0;
        while (true) {
            bool ok = true;
#pragma unroll
            for (int j = 0; j < MNG; ++j) ok = ok && (v[j].y == epoch) && (v[j].w == epoch);
            if (__all(ok)) break;
            ++spins;
            if (spins > M_SPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) {
                    xch_give_up(p.status);
                    sFlag[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < MNG; ++j) {
                const unsigned uo = (unsigned)(((slice + 1 + j) & (MG - 1)) * 32) * 16u;
                v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, gvoff, base + uo, 16);
            }
        }
#pragma unroll
        for (int j = 0; j < MNG; ++j) {
            float* d = sH + lbase + ((slice + 1 + j) & (MG - 1)) * 32;
            d[0] = __uint_as_float(v[j].x);
            d[MLDH] = __uint_as_float(v[j].z);
        }
    };

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * MBT;
        __syncthreads();   // previous tile fully consumed
        // ---- initial state: full h tiles to LDS, own c cells to registers, x_0 ----
        for (int e = tid; e < MBT * MH; e += 256) {
            const int row = e >> 8, u = e & 255;
            const bool ok = b0 + row < p.B;
            sH1[row * MLDH + u] = ok ? p.h1_0[(size_t)(b0 + row) * MH + u] : 0.f;
            sH2[row * MLDH + u] = ok ? p.h2_0[(size_t)(b0 + row) * MH + u] : 0.f;
        }
        if (tid < MBT * 8) {
            const int row = tid >> 3, o = tid & 7;
            sX[tid] = (o < O && b0 + row < p.B) ? p.dec0[(size_t)(b0 + row) * O + o] : 0.f;
        }
        float c1[2], c2[2], h1c[2] = {0.f, 0.f}, h2c[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = b0 + my_row0 + r;
            c1[r] = (row < p.B) ? p.c1_0[(size_t)row * MH + unit] : 0.f;
            c2[r] = (row < p.B) ? p.c2_0[(size_t)row * MH + unit] : 0.f;
        }
        __syncthreads();
        const int hd_o = tid & 15, hd_row = b0 + (tid >> 4);   // head: thread = (sequence, output)
        const float* h1row = sH1 + n * MLDH + 4 * g4;
        const float* h2row = sH2 + n * MLDH + 4 * g4;
        f32x4 acc1[2], acc2[2];
        // recurrent half of layer 1, step 0
        acc1[0] = (f32x4){b1v[0], b1v[0], b1v[0], b1v[0]};
        acc1[1] = (f32x4){b1v[1], b1v[1], b1v[1], b1v[1]};
        mfma_begin2(acc1);
        recur_agpr(acc1, h1row, w1);
        mfma_end2(acc1);
        __syncthreads();   // every wave has read h1_0 before the first own-slice write of h1_t into the tile
        for (int t = 0; t < p.T_out; ++t) {
            MIX_STAMP(0);
            // the "others" term of the mixing layer for this step: requested now, consumed in the head
            const bool oth_live = hd_o < O && hd_row < p.B;   // dead lanes read element 0, masked (no load inside a branch)
            const float oth_ld = p.oth_proj[oth_live ? (size_t)hd_row * p.oth_sb + (size_t)t * p.oth_st + hd_o : 0];
            const float othv = oth_live ? oth_ld : 0.f;
            ++epoch;
            const unsigned par = (epoch & 1u) * PARITY_BYTES;
            // ================= layer 1: + x_t . K1, cell update =================
            {
                float xa[2];
                xa[0] = sX[n * 8 + g4];
                xa[1] = sX[n * 8 + 4 + g4];
                mfma_guard2(xa[0], xa[1]);
                mfma_begin2(acc1);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mfma_v(acc1[0], xa[s], k1[s][0]);
                    mfma_v(acc1[1], xa[s], k1[s][1]);
                }
                mfma_end2(acc1);
            }
            {
                // swap halves: lanes hi=0 keep rows 0,1 (own i,g; f,o from the partner), hi=1 keep rows 2,3
                float snd[4], rcv[4];
                snd[0] = hi ? acc1[0][0] : acc1[0][2];
                snd[1] = hi ? acc1[0][1] : acc1[0][3];
                snd[2] = hi ? acc1[1][0] : acc1[1][2];
                snd[3] = hi ? acc1[1][1] : acc1[1][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) rcv[k] = swap_half(snd[k]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float zi = hi ? rcv[r] : acc1[0][r];
                    const float zf = hi ? acc1[0][2 + r] : rcv[r];
                    const float zg = hi ? rcv[2 + r] : acc1[1][r];
                    const float zo = hi ? acc1[1][2 + r] : rcv[2 + r];
                    const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
                    c1[r] = fmaf(fg, c1[r], ig * gg);
                    h1c[r] = og * tanh_f(c1[r]);
                    if (TRAIN) {
                        const int row = b0 + my_row0 + r;
                        if (row < p.B) {
                            float* rp = p.res1 + (((size_t)t * p.B + row) * 5) * MH + unit;
                            rp[0] = ig; rp[MH] = fg; rp[2 * MH] = gg; rp[3 * MH] = og; rp[4 * MH] = c1[r];
                            p.H1[((size_t)t * p.B + row) * MH + unit] = h1c[r];
                            p.C1[((size_t)t * p.B + row) * MH + unit] = c1[r];
                        }
                    }
                }
            }
            MIX_STAMP(1);
            // publish h1_t, then start the gather and run h2_{t-1} . R2 under it
#pragma unroll
            for (int r = 0; r < 1; ++r) {
                const mu32x4 gr = {__float_as_uint(h1c[0]), epoch, __float_as_uint(h1c[1]), epoch};
                if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, pub_off, par, 1);
                else __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, pub_off, par, 16);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) sH1[(my_row0 + r) * MLDH + unit] = h1c[r];
            acc2[0] = (f32x4){b2v[0], b2v[0], b2v[0], b2v[0]};
            acc2[1] = (f32x4){b2v[1], b2v[1], b2v[1], b2v[1]};
            // the gather is requested part-way through the MFMAs: the partners publish at about the same moment and an sc1
            // store needs most of a microsecond to become visible - loads issued right behind the own publish came back stale
            // and cost a second sweep (137 -> 112 us for 10 steps at B = 512; any split point from 4 to 16 measures the same)
            mfma_begin2(acc2);
            recur_agpr<0, GATHER_AFTER_BLOCK>(acc2, h2row, w2);
            gather_issue(par);
            recur_agpr<GATHER_AFTER_BLOCK, 16>(acc2, h2row, w2);
            mfma_end2(acc2);
            MIX_STAMP(2);
            gather_finish(par, sH1);
            MIX_STAMP(3);
            __syncthreads();   // barrier D: the whole h1_t tile is in LDS; every wave is done reading sH2
            MIX_STAMP(4);
            if (sFlag[0]) { aborted = true; break; }
            // ================= layer 2: + h1_t . K2 (streamed), cell update =================
            mfma_begin2(acc2);
            recur_k2(acc2, h1row, sK2l, k2r);
            mfma_end2(acc2);
            MIX_STAMP(5);
            {
                float snd[4], rcv[4];
                snd[0] = hi ? acc2[0][0] : acc2[0][2];
                snd[1] = hi ? acc2[0][1] : acc2[0][3];
                snd[2] = hi ? acc2[1][0] : acc2[1][2];
                snd[3] = hi ? acc2[1][1] : acc2[1][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) rcv[k] = swap_half(snd[k]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float zi = hi ? rcv[r] : acc2[0][r];
                    const float zf = hi ? acc2[0][2 + r] : rcv[r];
                    const float zg = hi ? rcv[2 + r] : acc2[1][r];
                    const float zo = hi ? acc2[1][2 + r] : rcv[2 + r];
                    const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
                    c2[r] = fmaf(fg, c2[r], ig * gg);
                    h2c[r] = og * tanh_f(c2[r]);
                    if (TRAIN) {
                        const int row = b0 + my_row0 + r;
                        if (row < p.B) {
                            float* rp = p.res2 + (((size_t)t * p.B + row) * 5) * MH + unit;
                            rp[0] = ig; rp[MH] = fg; rp[2 * MH] = gg; rp[3 * MH] = og; rp[4 * MH] = c2[r];
                            p.H2[((size_t)t * p.B + row) * MH + unit] = h2c[r];
                            p.C2[((size_t)t * p.B + row) * MH + unit] = c2[r];
                        }
                    }
                }
            }
#pragma unroll
            for (int r = 0; r < 1; ++r) {
                const mu32x4 gr = {__float_as_uint(h2c[0]), epoch, __float_as_uint(h2c[1]), epoch};
                if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, pub_off, LAYER_BYTES + par, 1);
                else __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, pub_off, LAYER_BYTES + par, 16);
            }
#pragma unroll
            for (int r = 0; r < 2; ++r) sH2[(my_row0 + r) * MLDH + unit] = h2c[r];
            const bool more = (t + 1 < p.T_out);
            MIX_STAMP(6);
            // recurrent half of layer 1 for step t+1 under the gather of h2_t (requested part-way, as above)
            acc1[0] = (f32x4){b1v[0], b1v[0], b1v[0], b1v[0]};
            acc1[1] = (f32x4){b1v[1], b1v[1], b1v[1], b1v[1]};
            if (more) {
                mfma_begin2(acc1);
                recur_agpr<0, GATHER_AFTER_BLOCK>(acc1, h1row, w1);
                gather_issue(LAYER_BYTES + par);
                recur_agpr<GATHER_AFTER_BLOCK, 16>(acc1, h1row, w1);
                mfma_end2(acc1);
            } else {
                gather_issue(LAYER_BYTES + par);
            }
            MIX_STAMP(7);
            gather_finish(LAYER_BYTES + par, sH2);
            MIX_STAMP(8);
            __syncthreads();   // barrier G: the whole h2_t tile is in LDS
            MIX_STAMP(9);
            if (sFlag[0]) { aborted = true; break; }
            // ================= head: p = tanh(h2 Wd + bd), m = tanh(p Wp + add) =================
            {
                // Dense on the matrix pipe: wave w contracts hidden units [64w, 64w+64) of the h2 tile (A operand)
                // with Wd (B operand: column n = output o, zero for n >= 8); the four partial 16 x 16 products meet
                // in LDS.  (A VALU version with per-thread slices of Wd in LDS cost 12 k cycles per step in bank
                // conflicts - 39 % of the step.)
                f32x4 dacc[2];
                dacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* hq = h2row + 64 * wave;
                const float* wq = sWd + (64 * wave + 4 * g4) * 8 + (n & 7);
                f32x4 hb[4];
                float wb[4][4];
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    hb[b] = *(const f32x4*)(hq + 16 * b);
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) wb[b][ss] = (n < 8) ? wq[(16 * b + ss) * 8] : 0.f;
                    mfma_guard4(wb[b][0], wb[b][1], wb[b][2], wb[b][3]);   // written by the select above
                }
                mfma_begin2(dacc);
#pragma unroll
                for (int b = 0; b < 4; ++b)
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) mfma_v(dacc[ss & 1], hb[b][ss], wb[b][ss]);
                mfma_end2(dacc);
                // partial (rows 4*g4 + r, column n) of this wave -> sPart[wave][row][col]
#pragma unroll
                for (int r = 0; r < 4; ++r) sPart[(wave * 16 + 4 * g4 + r) * 16 + n] = dacc[0][r] + dacc[1][r];
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
                if (o < 8) sX[row * 8 + o] = mv;   // x_{t+1}
                if (slice == 0 && o < O && brow < p.B) {
                    p.out[((size_t)t * p.B + brow) * O + o] = mv;
                    if (TRAIN) p.P[((size_t)t * p.B + brow) * O + o] = pv;
                }
            }
            MIX_STAMP(10);
            __syncthreads();   // barrier H: x_{t+1} is in LDS
            MIX_STAMP(11);
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.h1T) p.h1T[(size_t)row * MH + unit] = h1c[r];
                    if (p.c1T) p.c1T[(size_t)row * MH + unit] = c1[r];
                    if (p.h2T) p.h2T[(size_t)row * MH + unit] = h2c[r];
                    if (p.c2T) p.c2T[(size_t)row * MH + unit] = c2[r];
                }
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

// K2 (H,4H) -> fragments in the order recur_stream reads them:
// [slice 8][wave 4][k-block j 16][tile 2][lane 64][k-sub s 4], value = K2[16j + 4*(lane>>4) + s][gate column]
__global__ __launch_bounds__(256) void mix_decoder_pack_k2_kernel(const float* __restrict__ K2, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;   // one float each, 256*1024 in total
    if (idx >= MH * 4 * MH) return;
    const int s = idx & 3, lane = (idx >> 2) & 63, tile = (idx >> 8) & 1, j = (idx >> 9) & 15, wave = (idx >> 13) & 3,
              slice = idx >> 15;
    const int n = lane & 15, g4 = lane >> 4;
    const int unit = 32 * slice + 8 * wave + (n & 7);
    const int gate = 2 * tile + (n >> 3);
    out[idx] = K2[(size_t)(16 * j + 4 * g4 + s) * (4 * MH) + gate * MH + unit];
}

// header + fixed granule area + the packed copy of K2
size_t mix_decoder_workspace_bytes(int B) {
    (void)B;
    return kStatusBytes + kXchBytes + sizeof(float) * (size_t)MH * 4 * MH;
}

// The packed copy of K2 depends on the weights only: a caller that runs it ahead of time (on a side stream, while the
// encoder layers run) calls this, then orders the streams, then launches - the launch finds the mark and skips its own pack.
int mix_decoder_prepack(const float* K2, void* workspace, hipStream_t stream) {
    float* k2p = (float*)((char*)workspace + kStatusBytes + kXchBytes);
    hipLaunchKernelGGL(mix_decoder_pack_k2_kernel, dim3(MH * 4 * MH / 256), dim3(256), 0, stream, K2, k2p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder_prepack launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    prepack_mark(workspace, K2);
    return FOV_OK;
}

int mix_decoder_launch(MixDecParams p, const float* K2, int act, int train, void* workspace, hipStream_t stream) {
    if (p.B == 0 || p.T_out == 0) return FOV_OK;
    p.num_tiles = (p.B + MBT - 1) / MBT;
    const int max_groups = device_cu_count() / MG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("fused mixing decoder needs at least %d CUs", MG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 4 * MBT * MH * sizeof(unsigned long long) > kXchBytes - kHelloBytes) { set_error("mix_decoder: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    float* k2p = (float*)((char*)workspace + kStatusBytes + kXchBytes);
    p.K2p = k2p;
    p.epoch_span = p.T_out * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;   // no memset: tags continue from the header
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    // the packed copy may already be there: mix_decoder_prepack (another stream, under the encoder) marks the workspace for ONE launch
    if (!prepack_consume(workspace, K2)) hipLaunchKernelGGL(mix_decoder_pack_k2_kernel, dim3(MH * 4 * MH / 256), dim3(256), 0, stream, K2, k2p);
    const size_t lds = sizeof(float) * (2 * MBT * MLDH + MBT * 8 + MH * 8 + 64 + 16 + 4 * 256 + 4 * 28 * 256);
    void (*kern)(MixDecParams) = nullptr;
    if (act == FOV_ACT_HARD_SIGMOID) kern = train ? mix_decoder_kernel<FOV_ACT_HARD_SIGMOID, true> : mix_decoder_kernel<FOV_ACT_HARD_SIGMOID, false>;
    else kern = train ? mix_decoder_kernel<FOV_ACT_SIGMOID, true> : mix_decoder_kernel<FOV_ACT_SIGMOID, false>;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(xch_padded_groups(p.num_groups) * MG), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_mix_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mix_stamps), sizeof(unsigned long long) * MSTAMP_STEPS * MSTAMP_SLOTS);
}
#endif

}  // namespace fov
