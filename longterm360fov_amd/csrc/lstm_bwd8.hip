// Persistent BPTT recurrence of one LSTM layer, H = 256, in groups of EIGHT workgroups per 16-sequence tile - fp32
// and bf16 operand forms of one kernel.  (What Keras/TF autodiff runs under model.fit for the LSTM layers of
// mycode/given_others_gt_mean_var_seq2seq.py:108-115.)
//
// lstm_bwd_cluster.hip uses groups of H/64 = 4 workgroups: at 512 sequences per GPU (BASELINE configs[2]/[4]: 32 tiles)
// that occupies 128 of the 256 CUs.  Here workgroup `slice` owns hidden units [32*slice, +32) for all four gates -
// the ownership of lstm_wide.hip / mix_decoder_bwd.hip - so 32 tiles fill the chip.  Lane (n, g4) owns the cells
// (rows 4*g4 + 2*(n>>3) + {0,1}, unit 32*slice + 8*wave + (n&7)): dc and the recurrent dh never leave registers.
// Per step t = T-1 .. 0:
//   gates backward for the lane's two cells (from the reserve i,f,g,o,c of the training forward) -> dz (4 gates), to
//   dZ (B,T,4H) for the weight-gradient products and into an LDS tile (16 x 128 own gate columns);
//   partial[16 x 256] = dz_own . R^T_own : the contribution of the own gate columns to dh_{t-1} of ALL 256 units
//   (fp32 kernel: 128 x v_mfma_f32_16x16x4_f32 per wave, R^T slice in 128 AGPRs; the bf16 kernel further down splits
//   the product by OUTPUT unit instead and all-gathers the rounded dz tile);
//   the 16 x 32 piece of every destination workgroup travels as fp32 {value, epoch} granules; each lane gathers
//   the eight pieces of its own cells and adds them in slice order (deterministic).
// The bias gradient leaves as one (tiles, 4H) partial (sum over the tile's rows and all steps), as in the 4-group kernel.
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

namespace {

constexpr int B8LDZ = 128 + 4;     // fp32 LDS row stride of the dz tile
constexpr int B8LDQ = 128 + 16;    // bf16 LDS row stride: 288 B = 72 dwords = 8 (mod 64): conflict-free ds_read_b128
constexpr size_t B8_PAR = (size_t)QG * QG * QBT * 32;   // granules per parity: [dest][src][row][unit]

struct Bwd8Params {
    const float* R;
    const float* reserve;   // (B,T,5,H)
    const float* c0;        // (B,H) or NULL
    const float* dhs;       // (B,T,H) or NULL
    const float* dhT;       // (B,H) or NULL
    const float* dcT;       // (B,H) or NULL
    float* dz;              // (B,T,4H) out
    float* dh0;             // (B,H) or NULL
    float* dc0;             // (B,H) or NULL
    float* db_part;         // (num_tiles, 4H) or NULL
    const float* K;         // DX forms: input kernel (256, 4H) of the layer ...
    float* dx;              // ... and its data gradient dx_t = dz_t K^T (B,T,256), the dhs of the layer below (bf16 kernel);
                            // fp32 kernel: EIGHT partial tapes [slice][B][T][256] - workgroup `slice` contributes its own gate
                            // columns' share, the host adds them in slice order (splitk_reduce)
    unsigned long long* xch;
    unsigned* status;
    int B, T, num_groups, num_tiles, epoch_span;
};

__device__ __forceinline__ void b8_mfma_a(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}
template <int ACT>
__device__ __forceinline__ float b8_act_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

// DXF (round 5): the layer's data gradient dx_t = dz_t K^T (a 256-wide input: the stacked encoder layer of the mixing model) is
// formed here as well.  K^T's own-gate-column slice sits in the OTHER 128 accumulation registers (the same fragments as R^T's,
// read from K), the 128 MFMAs of the partial product run between the publication of the dh partials and their gather - in
// the shadow of the exchange - and the workgroup's partial [16 x 256] leaves as a tape of its own; the eight tapes are added by
// one reduce launch.  Before: a 5120 x 1024 x 256 GEMM + reduce (54 us) behind the recurrence, which under model.fit's second
// stream also waited 90 - 120 us for the weight-gradient GEMMs' workgroups to leave the chip
// (profiles/r05_train_mixing_f32_step_timeline.txt).
template <int ACT, bool DXF>
__global__ __launch_bounds__(256, 1) void lstm_bwd8_kernel(Bwd8Params p) {
    constexpr bool BF16 = false;   // the bf16 form is lstm_bwd8n_bf16_kernel below (N-split)
    __shared__ __attribute__((aligned(16))) float sDZ[BF16 ? 4 : QBT * B8LDZ];
    __shared__ __attribute__((aligned(16))) unsigned short sDQ[BF16 ? QBT * B8LDQ : 8];
    __shared__ int sFlag[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    if (!q_group_slice(p.num_groups, group, slice)) { q_spare_leaves(p.status); return; }
    constexpr int H4 = 4 * QH;
    const int hi = n >> 3;
    const int ul = 8 * wave + (n & 7);          // unit inside the workgroup (0..31)
    const int unit = 32 * slice + ul;
    const int my_row0 = 4 * g4 + 2 * hi;
    const int T = p.T;
    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- resident R^T fragments.  Tile tl of this wave: destination slice 2*wave + (tl>>1), half tl&1; its output unit
    // on this lane is nout; k index lc is an own gate column: gate lc>>5, unit 32*slice + (lc & 31). ----
    float rt[BF16 ? 1 : 4][BF16 ? 1 : 8][4];   // fp32: [tl][jb][s], lc = 16*jb + 4*g4 + s
    float kt[DXF ? 4 : 1][DXF ? 8 : 1][4];      // DXF: the same fragments of K (input unit nout, own gate column lc)
    qu32x4 rq[BF16 ? 4 : 1][BF16 ? 4 : 1];     // bf16: [tl][kb],    lc = 32*kb + 8*g4 + j
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const int nout = 32 * (2 * wave + (tl >> 1)) + 16 * (tl & 1) + n;
        if constexpr (BF16) {
#pragma unroll
            for (int kb = 0; kb < 4; ++kb) rq[tl][kb] = load_bfrag_rowmajor(p.R + (size_t)nout * H4 + kb * QH + 32 * slice + 8 * g4);
        } else {
#pragma unroll
            for (int jb = 0; jb < 8; ++jb) {
                const int lc = 16 * jb + 4 * g4;
                const f32x4 v = *(const f32x4*)(p.R + (size_t)nout * H4 + (lc >> 5) * QH + 32 * slice + (lc & 31));
#pragma unroll
                for (int s = 0; s < 4; ++s) rt[tl][jb][s] = v[s];
                if constexpr (DXF) {
                    const f32x4 w = *(const f32x4*)(p.K + (size_t)nout * H4 + (lc >> 5) * QH + 32 * slice + (lc & 31));
#pragma unroll
                    for (int s = 0; s < 4; ++s) kt[tl][jb][s] = w[s];
                }
            }
        }
    }
    unsigned long long* gbase = p.xch + (size_t)group * 2 * B8_PAR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, (int)(2 * B8_PAR * 8), 0x00020000);
    xch_hello_poll(p.status, sXch, group, QG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * QBT;
        float dc[2], dh[2];
        bool live[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = b0 + my_row0 + r;
            live[r] = row < p.B;
            dc[r] = (live[r] && p.dcT) ? p.dcT[(size_t)row * QH + unit] : 0.f;
            dh[r] = (live[r] && p.dhT) ? p.dhT[(size_t)row * QH + unit] : 0.f;
        }
        // Tape of this lane's two cells, ONE step ahead: tp[0..3] = i,f,g,o and tp[4] = c of the step, tp[5] = c of the
        // step before it (c0 / zero in front of step 0), tp[6] = dhs of the step.  The loads are UNCONDITIONAL (rows and
        // steps clamped into the tensor, dead rows masked where dz is formed): a load inside a branch is waited for at the
        // merge and the step would wait for HBM right there.
        float cur[7][2], pre[7][2];
        auto load_step = [&](int t, float (&dst)[7][2]) {
            const int tc = t > 0 ? t : 0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                const size_t rowc = (size_t)(row < p.B ? row : p.B - 1);
                const float* rp = p.reserve + ((rowc * T + tc) * 5) * QH + unit;
#pragma unroll
                for (int q = 0; q < 5; ++q) dst[q][r] = rp[q * QH];
                const float* cp = tc > 0 ? rp - QH : (p.c0 ? p.c0 + rowc * QH + unit : rp);   // no c0: any valid address, masked at use
                dst[5][r] = *cp;
                dst[6][r] = p.dhs ? p.dhs[(rowc * T + tc) * QH + unit] : 0.f;
            }
        };
        load_step(T - 1, cur);
        float dbacc[4] = {0.f, 0.f, 0.f, 0.f};   // sum over t and this lane's 2 sequences of dz, per gate
        __syncthreads();   // the previous tile's last step is done with the dz tile

        for (int t = T - 1; t >= 0; --t) {
            ++epoch;
            const unsigned par = (epoch & 1u) * (unsigned)(B8_PAR * 8);
            // ---- pointwise: dz of this lane's two cells ----
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float ig = cur[0][r], fg = cur[1][r], gg = cur[2][r], og = cur[3][r], cc = cur[4][r];
                const float cprev = (t > 0 || p.c0) ? cur[5][r] : 0.f;   // step 0 without a given state: c_{-1} = 0
                const float dht = dh[r] + cur[6][r];
                const float tc = tanh_f(cc);
                const float dcv = dc[r] + dht * og * (1.f - tc * tc);
                float dzv[4];
                dzv[0] = live[r] ? dcv * gg * b8_act_grad<ACT>(ig) : 0.f;
                dzv[1] = live[r] ? dcv * cprev * b8_act_grad<ACT>(fg) : 0.f;
                dzv[2] = live[r] ? dcv * ig * (1.f - gg * gg) : 0.f;
                dzv[3] = live[r] ? dht * tc * b8_act_grad<ACT>(og) : 0.f;
                dc[r] = dcv * fg;
                if (live[r]) {
                    float* zp = p.dz + ((size_t)(b0 + my_row0 + r) * T + t) * H4 + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) zp[g * QH] = dzv[g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    dbacc[g] += dzv[g];
                    if constexpr (BF16) sDQ[(my_row0 + r) * B8LDQ + g * 32 + ul] = bf16_bits(dzv[g]);
                    else sDZ[(my_row0 + r) * B8LDZ + g * 32 + ul] = dzv[g];
                }
            }
            // the tape of step t-1 is requested here, a whole step before its use
            load_step(t - 1, pre);
            __syncthreads();   // barrier A: the dz tile is complete
            // ---- partial[16 x 256] = dz_own . R^T_own ; tile tl -> destination 2*wave + (tl>>1) ----
            f32x4 acc[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (BF16) {
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const qu32x4 a = *(const qu32x4*)(sDQ + n * B8LDQ + 32 * kb + 8 * g4);
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) qmfma(acc[tl], a, rq[tl][kb]);
                }
            } else {
                asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                const float* arow = sDZ + n * B8LDZ + 4 * g4;
                f32x4 a = *(const f32x4*)arow, an = a;
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) {
                    if (jb + 1 < 8) an = *(const f32x4*)(arow + 16 * (jb + 1));
                    asm volatile("s_nop 1" : "+v"(a));
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                        for (int s = 0; s < 4; ++s) b8_mfma_a(acc[tl], a[s], rt[tl][jb][s]);
                    a = an;
                }
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
            }
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                // granule order [dest][src][row pair][unit][row of the pair]: the two rows of a lane's cells are adjacent tagged
                // granules, moved by ONE 16-byte store / load (round 3: half the exchange instructions, same 8-byte atomicity)
                const int d = 2 * wave + (tl >> 1);
                const unsigned off = (unsigned)((((d * QG + slice) * (QBT / 2) + 2 * g4) * 32) + 16 * (tl & 1) + n) * 16u;
#pragma unroll
                for (int rp = 0; rp < 2; ++rp) {
                    const qu32x4 gr = {__float_as_uint(acc[tl][2 * rp]), epoch, __float_as_uint(acc[tl][2 * rp + 1]), epoch};
                    if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off + rp * 32 * 16, par, 1);
                    else __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off + rp * 32 * 16, par, 16);
                }
            }
            if constexpr (DXF) {
                // ---- this workgroup's share of dx_t = dz_t K^T while the dh partials travel ----
                f32x4 ax[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ax[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                asm volatile("s_nop 3" : "+v"(ax[0]), "+v"(ax[1]), "+v"(ax[2]), "+v"(ax[3]));
                const float* arow = sDZ + n * B8LDZ + 4 * g4;
                f32x4 a = *(const f32x4*)arow, an = a;
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) {
                    if (jb + 1 < 8) an = *(const f32x4*)(arow + 16 * (jb + 1));
                    asm volatile("s_nop 1" : "+v"(a));
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                        for (int s = 0; s < 4; ++s) b8_mfma_a(ax[tl], a[s], kt[tl][jb][s]);
                    a = an;
                }
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(ax[0]), "+v"(ax[1]), "+v"(ax[2]), "+v"(ax[3]));
#pragma unroll
                for (int tl = 0; tl < 4; ++tl) {
                    const int col = 32 * (2 * wave + (tl >> 1)) + 16 * (tl & 1) + n;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = b0 + 4 * g4 + r;
                        if (row < p.B) p.dx[(((size_t)slice * p.B + row) * T + t) * QH + col] = ax[tl][r];
                    }
                }
            }
            // ---- gather the 8 pieces of this lane's two cells (one 16-byte load per source), add in slice order ----
            {
                const unsigned voff = (unsigned)(((slice * QG) * (QBT / 2) + (my_row0 >> 1)) * 32 + ul) * 16u;
                constexpr unsigned SSTR = QBT * 32 * 8;   // src stride in bytes
                float part[2][8];
                unsigned bad = 0;
                {
                    qu32x4 v[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + s * SSTR, par, 16);
#pragma unroll
                    for (int s = 0; s < 8; ++s) {
                        part[0][s] = __uint_as_float(v[s].x);
                        part[1][s] = __uint_as_float(v[s].z);
                        if (v[s].y != epoch || v[s].w != epoch) bad |= (1u << s);
                    }
                }
                unsigned spins = 0;
                while (__any(bad != 0)) {
                    ++spins;
                    if (spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                        if (lane == 0) { xch_give_up(p.status); sFlag[0] = 1; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    qu32x4 tv[8];
#pragma unroll
                    for (int s = 0; s < 8; ++s) tv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + s * SSTR, par, 16);
#pragma unroll
                    for (int s = 0; s < 8; ++s)
                        if (((bad >> s) & 1u) && tv[s].y == epoch && tv[s].w == epoch) {
                            part[0][s] = __uint_as_float(tv[s].x);
                            part[1][s] = __uint_as_float(tv[s].z);
                            bad &= ~(1u << s);
                        }
                }
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float a = 0.f;
#pragma unroll
                    for (int s = 0; s < 8; ++s) a += part[q][s];
                    dh[q] = a;
                }
            }
#pragma unroll
            for (int q = 0; q < 7; ++q)
#pragma unroll
                for (int r = 0; r < 2; ++r) cur[q][r] = pre[q][r];   // requested most of a step ago: long landed
            __syncthreads();   // barrier B: every wave is done reading the dz tile; sFlag is uniform below
            if (sFlag[0]) { aborted = true; break; }
        }
        if (!aborted && p.db_part) {
            // the 8 lanes (g4 0..3, hi 0..1) that share a unit hold different rows: fold them in a fixed order
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = dbacc[g];
                v += __shfl_xor(v, 8);
                const float v1 = __shfl(v, (lane + 16) & 63), v2 = __shfl(v, (lane + 32) & 63), v3 = __shfl(v, (lane + 48) & 63);
                if (g4 == 0 && hi == 0) p.db_part[(size_t)tile * H4 + g * QH + unit] = (v + v1) + (v2 + v3);
            }
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.dh0) p.dh0[(size_t)row * QH + unit] = dh[r];
                    if (p.dc0) p.dc0[(size_t)row * QH + unit] = dc[r];
                }
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

// ---------------------------------------------------------------------------------------------------------------
// bf16 form (BASELINE configs[4]), N-SPLIT: the rounded dz tile is all-gathered (bf16_common.h: 4 granules per lane
// and step instead of 16 fp32 partial sums) and every workgroup computes dh_{t-1} of ITS 32 units from the whole
// (16 x 1024) tile: wave w contracts gate w's 256 columns (8 k-blocks x 2 N-tiles = 16 MFMAs, R^T slice = 64
// registers of packed B fragments: eight contiguous floats of a row of R each), the four partial 16 x 32 tiles meet
// in LDS and are added in wave order.
// ---------------------------------------------------------------------------------------------------------------
// Diagnostic build only (-DFOV_STAMPS, tools/stamp_bf16_layer.py --bwd): s_memtime stamps of one wave per step.
#ifdef FOV_STAMPS
// stamps go to LDS and leave in one piece at the kernel's end: a global store per stamp would sit in the wave's vmcnt queue and
// every s_waitcnt vmcnt(0) of the step would wait for its acknowledgement (measured: 2 000 cycles per stamp that way)
__device__ unsigned long long g_b8_stamps[32][12];
// branch-free (a stamp inside `if (stamp_on)` splits the basic block and the optimiser sinks the arithmetic in front of it to its
// uses behind it): every thread reads the clock, all but the stamping thread write a junk slot
#define B8_STAMP(slot)                                                                         \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");            \
        sStamps[(stamp_on && (T - 1 - t) < 31) ? (T - 1 - t) * 12 + slot : 31 * 12 + 11] = t_; \
    } while (0)
// the same, and the listed registers are computed BEFORE the clock is read
#define B8_STAMP_AFTER(slot, ...)                                                              \
    do {                                                                                       \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_), __VA_ARGS__::"memory"); \
        sStamps[(stamp_on && (T - 1 - t) < 31) ? (T - 1 - t) * 12 + slot : 31 * 12 + 11] = t_; \
    } while (0)
#else
#define B8_STAMP(slot) do { } while (0)
#define B8_STAMP_AFTER(slot, ...) do { } while (0)
#endif

// DX: the data gradient dx_t = dz_t K^T of a 256-wide input (the stacked layer: dx is the dhs of the layer below) is formed
// here as well, from the SAME gathered dz tile and the own 32 rows of K - 16 more MFMAs per wave and step and 64 more
// registers of fragments instead of a separate (B*T x 1024 x 256) product that re-reads dz from HBM.
// Lane -> cell mapping of the POINTWISE phase (round 4): thread tid owns row tid >> 4 of the tile and the two adjacent units
// 2 * (tid & 15), + 1 of the workgroup's 32: every tape access is an 8-byte access, a wave's instruction covers four whole
// 128-byte lines (it was one unit of two rows: 4-byte accesses, eight 32-byte pieces per instruction, twice the instructions -
// and the step was bound by the CU's address unit, 2 300 cycles to issue the 24 tape instructions of a step from four waves).
// The MFMA phase keeps the matrix layout (lane = (n, g4)); the two meet in LDS (dz tile in, partial dh tiles out) as before.
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int ACT, bool DX>
__global__ __launch_bounds__(256, 1) void lstm_bwd8n_bf16_kernel(Bwd8Params p) {
    __shared__ __attribute__((aligned(16))) unsigned short sDZ[QBT * QLDZ];   // the whole dz tile, bf16
    __shared__ float sRed[(DX ? 8 : 4) * QBT * 33];                           // [wave][row][unit] partial dh (+ partial dx)
    __shared__ int sFlag[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    if (!q_group_slice(p.num_groups, group, slice)) { q_spare_leaves(p.status); return; }
    constexpr int H4 = 4 * QH;
    const int prow = tid >> 4;                   // row of the tile (0..15)
    const int pu = 2 * (tid & 15);               // first unit of the pair inside the workgroup (0..30)
    const int unit0 = 32 * slice + pu;
    const int T = p.T;
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;
#ifdef FOV_STAMPS
    __shared__ unsigned long long sStamps[32 * 12];
    const bool stamp_on = (blockIdx.x == 5 && tid == 0);
    if (stamp_on) sStamps[31 * 12 + 0] = __builtin_amdgcn_s_memtime();
#endif

    // R^T fragments of this wave: k-block kb of gate `wave` (columns 256*wave + 32*kb + 8*g4 + j), N-tile nt (own unit 16*nt + n)
    qu32x4 rq[8][2];
#pragma unroll
    for (int kb = 0; kb < 8; ++kb)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            rq[kb][nt] = load_bfrag_rowmajor(p.R + (size_t)(32 * slice + 16 * nt + n) * H4 + QH * wave + 32 * kb + 8 * g4);
    qu32x4 kq[DX ? 8 : 1][2];   // K^T fragments, same shape: input unit 32*slice + 16*nt + n
    if constexpr (DX) {
#pragma unroll
        for (int kb = 0; kb < 8; ++kb)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                kq[kb][nt] = load_bfrag_rowmajor(p.K + (size_t)(32 * slice + 16 * nt + n) * H4 + QH * wave + 32 * kb + 8 * g4);
    }
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 2 * (Q_DZ_BYTES / 8), 0, (int)(2 * Q_DZ_BYTES), 0x00020000);
    xch_hello_poll(p.status, sXch, group, QG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * QBT;
        const int row = b0 + prow;
        const bool live = row < p.B;
        const size_t rowc = (size_t)(live ? row : p.B - 1);
        f32x2 dc = (live && p.dcT) ? *(const f32x2*)(p.dcT + rowc * QH + unit0) : (f32x2){0.f, 0.f};
        f32x2 dh = (live && p.dhT) ? *(const f32x2*)(p.dhT + rowc * QH + unit0) : (f32x2){0.f, 0.f};
        // Tape of this lane's two cells, ONE step ahead: [0..3] = i,f,g,o and [4] = c of the step, [5] = c of the step before it
        // (c0 / zero in front of step 0), [6] = dhs of the step.  The loads are UNCONDITIONAL (rows and steps clamped into the
        // tensor, dead rows masked where dz is formed): a load inside a branch gets an s_waitcnt vmcnt(0) at the merge and the
        // step would wait for HBM right there (measured: 3 800 cycles).
        f32x2 cur[7], pre[7];
        auto load_step = [&](int t, f32x2 (&dst)[7]) {
            const int tc = t > 0 ? t : 0;
            const float* rp = p.reserve + ((rowc * T + tc) * 5) * QH + unit0;
#pragma unroll
            for (int q = 0; q < 5; ++q) dst[q] = *(const f32x2*)(rp + q * QH);
            const float* cp = tc > 0 ? rp - QH : (p.c0 ? p.c0 + rowc * QH + unit0 : rp);   // no c0: any valid address, masked at use
            dst[5] = *(const f32x2*)cp;
            dst[6] = p.dhs ? *(const f32x2*)(p.dhs + (rowc * T + tc) * QH + unit0) : (f32x2){0.f, 0.f};
        };
        load_step(T - 1, cur);
        f32x2 dbacc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) dbacc[g] = (f32x2){0.f, 0.f};
        __syncthreads();   // the previous tile's last step is done with the LDS tiles

        for (int t = T - 1; t >= 0; --t) {
            B8_STAMP(0);
            ++epoch;
            const unsigned par = (epoch & 1u) * Q_DZ_BYTES;

            B8_STAMP(1);
            // ---- pointwise: dz of this lane's two cells ----
            f32x2 dzv[4];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float ig = cur[0][u], fg = cur[1][u], gg = cur[2][u], og = cur[3][u], cc = cur[4][u];
                const float cprev = (t > 0 || p.c0) ? cur[5][u] : 0.f;   // step 0 without a given state: c_{-1} = 0
                const float dht = dh[u] + cur[6][u];
                const float tc = tanh_f(cc);
                const float dcv = dc[u] + dht * og * (1.f - tc * tc);
                dzv[0][u] = live ? dcv * gg * b8_act_grad<ACT>(ig) : 0.f;
                dzv[1][u] = live ? dcv * cprev * b8_act_grad<ACT>(fg) : 0.f;
                dzv[2][u] = live ? dcv * ig * (1.f - gg * gg) : 0.f;
                dzv[3][u] = live ? dht * tc * b8_act_grad<ACT>(og) : 0.f;
                dc[u] = dcv * fg;
            }
            // publish first (the partners wait for it), then the tape traffic of this step
            unsigned dzp[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) dzp[g] = pack_bf16(dzv[g][0], dzv[g][1]);
            B8_STAMP_AFTER(2, "+v"(dzp[0]), "+v"(dzp[1]), "+v"(dzp[2]), "+v"(dzp[3]), "+v"(dc[0]), "+v"(dc[1]));
#ifndef FOV_DBG_B8_NOPUB
            q_dz_publish2(rs, par, prow, unit0, dzp, epoch, sDZ, ticket.same_xcd);
#endif
            // Everything that does not depend on the partners goes between the publish and the gather: an sc1 store
            // takes about a microsecond to become visible, a sweep issued earlier comes back stale and costs a second
            // round trip.  The tape of step t-1 is requested here, a whole step before its use.
            B8_STAMP(3);
#ifndef FOV_DBG_B8_NOTAPE     // timing experiments (wrong results): tools/b8_variants.sh
            load_step(t - 1, pre);
#endif
#pragma unroll
            for (int g = 0; g < 4; ++g) dbacc[g] += dzv[g];
#ifndef FOV_DBG_B8_NODZ
            if (live) {
                float* zp = p.dz + ((size_t)row * T + t) * H4 + unit0;
#pragma unroll
                for (int g = 0; g < 4; ++g) *(f32x2*)(zp + g * QH) = dzv[g];
            }
#endif
            B8_STAMP(4);
#ifndef FOV_DBG_B8_NOGATHER
            if (!q_dz_gather2(rs, par, slice, tid, epoch, sDZ, p.status)) sFlag[0] = 1;
#endif
            B8_STAMP(5);
            __syncthreads();   // barrier A: the whole dz tile is in LDS
            B8_STAMP(6);
            if (sFlag[0]) { aborted = true; break; }
            // ---- this wave's share of dh_{t-1}[16 x 32 own units]: gate `wave`'s 256 columns ----
            f32x4 acc[2], accx[2];
            acc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
            accx[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            {
                qu32x4 a[8];
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) a[kb] = *(const qu32x4*)(sDZ + n * QLDZ + QH * wave + 32 * kb + 8 * g4);
#pragma unroll
                for (int kb = 0; kb < 8; ++kb) {
                    qmfma(acc[0], a[kb], rq[kb][0]);
                    qmfma(acc[1], a[kb], rq[kb][1]);
                    if constexpr (DX) {
                        qmfma(accx[0], a[kb], kq[kb][0]);
                        qmfma(accx[1], a[kb], kq[kb][1]);
                    }
                }
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    sRed[(wave * QBT + 4 * g4 + r) * 33 + 16 * nt + n] = acc[nt][r];
                    if constexpr (DX) sRed[((4 + wave) * QBT + 4 * g4 + r) * 33 + 16 * nt + n] = accx[nt][r];
                }
            B8_STAMP(7);
            __syncthreads();   // barrier B: the four partial tiles are in LDS; every wave is done reading the dz tile
            B8_STAMP(8);
            {
                const float* q = sRed + prow * 33 + pu;
#pragma unroll
                for (int u = 0; u < 2; ++u) dh[u] = (q[u] + q[QBT * 33 + u]) + (q[2 * QBT * 33 + u] + q[3 * QBT * 33 + u]);
                if constexpr (DX) {
                    const float* qx = q + 4 * QBT * 33;
                    f32x2 dxv;
#pragma unroll
                    for (int u = 0; u < 2; ++u) dxv[u] = (qx[u] + qx[QBT * 33 + u]) + (qx[2 * QBT * 33 + u] + qx[3 * QBT * 33 + u]);
                    if (live) *(f32x2*)(p.dx + ((size_t)row * T + t) * QH + unit0) = dxv;
                }
            }
#pragma unroll
            for (int q = 0; q < 7; ++q) cur[q] = pre[q];   // requested most of a step ago: long landed
            B8_STAMP(9);
        }
        if (!aborted && p.db_part) {
            // the 16 rows of a unit pair sit on lanes 16 apart (4 rows per wave) and on the four waves: fold in a fixed order
            // through LDS (sRed is free: the loop ended on barrier B and its read-out)
            __syncthreads();
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int u = 0; u < 2; ++u) sRed[(prow * 4 + g) * 33 + pu + u] = dbacc[g][u];
            __syncthreads();
            if (tid < 128) {   // thread = (gate tid >> 5, unit tid & 31)
                const int g = tid >> 5, u = tid & 31;
                float v = 0.f;
#pragma unroll
                for (int r = 0; r < QBT; ++r) v += sRed[(r * 4 + g) * 33 + u];
                p.db_part[(size_t)tile * H4 + g * QH + 32 * slice + u] = v;
            }
        }
        if (!aborted && live) {
            if (p.dh0) *(f32x2*)(p.dh0 + (size_t)row * QH + unit0) = dh;
            if (p.dc0) *(f32x2*)(p.dc0 + (size_t)row * QH + unit0) = dc;
        }
    }
#ifdef FOV_STAMPS
    if (stamp_on) {
        sStamps[31 * 12 + 1] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 32 * 12; ++i) (&g_b8_stamps[0][0])[i] = sStamps[i];
    }
#endif
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

}  // namespace

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_b8_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_b8_stamps), sizeof(unsigned long long) * 32 * 12);
}
#endif

// one workgroup per CU: groups of eight fill the chip up to 32 tiles; beyond that the 4-group kernel is as good
bool bwd8_preferred(int B, int H) { return H == QH && B > 0 && B <= 32 * QBT; }

// status word + granule buffers live at `xch_ws` (kStatusBytes + kXchBytes)
int launch_bwd8(const float* R, const float* reserve, const float* c0, const float* dhs, const float* dhT, const float* dcT,
                float* dz, float* dh0, float* dc0, float* db_part, int B, int T, int act, int bf16, void* xch_ws, hipStream_t stream,
                const float* K_dx, float* dx) {
    if (B == 0 || T == 0) return FOV_OK;
    Bwd8Params p = {};
    p.K = K_dx; p.dx = dx;
    const bool with_dx = K_dx && dx;      // (fp32: dx = eight partial tapes [slice][B][T][256])
    if (with_dx && (((uintptr_t)K_dx) & 15)) { set_error("8-group BPTT kernel: K must be 16-byte aligned"); return FOV_ERR_INVALID; }
    p.R = R; p.reserve = reserve; p.c0 = c0; p.dhs = dhs; p.dhT = dhT; p.dcT = dcT; p.dz = dz; p.dh0 = dh0; p.dc0 = dc0; p.db_part = db_part;
    p.B = B; p.T = T;
    p.num_tiles = (B + QBT - 1) / QBT;
    const int max_groups = device_cu_count() / QG;
    if (max_groups < 1) { set_error("8-group BPTT kernel needs at least %d CUs", QG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 2 * (bf16 ? (size_t)Q_DZ_BYTES : B8_PAR * 8) > kXchBytes - kHelloBytes) { set_error("8-group BPTT kernel: granule area too small"); return FOV_ERR_WORKSPACE; }
    if (bf16 && (((uintptr_t)R) & 15)) { set_error("8-group BPTT kernel: R must be 16-byte aligned"); return FOV_ERR_INVALID; }
    if (bf16 && ((((uintptr_t)reserve) | ((uintptr_t)c0) | ((uintptr_t)dhs) | ((uintptr_t)dhT) | ((uintptr_t)dcT) | ((uintptr_t)dz) |
                  ((uintptr_t)dh0) | ((uintptr_t)dc0) | ((uintptr_t)dx)) & 7)) {   // 8-byte tape accesses (two adjacent units per lane)
        set_error("8-group bf16 BPTT kernel: tapes and state gradients must be 8-byte aligned");
        return FOV_ERR_INVALID;
    }
    p.status = (unsigned*)xch_ws;
    p.xch = (unsigned long long*)((char*)xch_ws + kStatusBytes);
    p.epoch_span = T * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(Bwd8Params) = nullptr;
    if (bf16 && with_dx) kern = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd8n_bf16_kernel<FOV_ACT_HARD_SIGMOID, true> : lstm_bwd8n_bf16_kernel<FOV_ACT_SIGMOID, true>;
    else if (bf16) kern = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd8n_bf16_kernel<FOV_ACT_HARD_SIGMOID, false> : lstm_bwd8n_bf16_kernel<FOV_ACT_SIGMOID, false>;
    else if (with_dx) kern = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd8_kernel<FOV_ACT_HARD_SIGMOID, true> : lstm_bwd8_kernel<FOV_ACT_SIGMOID, true>;
    else kern = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd8_kernel<FOV_ACT_HARD_SIGMOID, false> : lstm_bwd8_kernel<FOV_ACT_SIGMOID, false>;
    hipLaunchKernelGGL(kern, dim3(q_padded_groups(p.num_groups) * QG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("8-group BPTT launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
