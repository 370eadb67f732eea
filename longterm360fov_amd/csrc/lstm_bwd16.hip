// Persistent BPTT recurrence of one LSTM layer at width 512, in groups of SIXTEEN workgroups per 16-sequence tile, fp32.
// (What TF autodiff runs under the train_op of mycode/lstm.py:556-567 for its MultiRNNCell of LSTMCell(400), :218-240,
// zero-padded to the matrix-core width 512 by training.TFLSTMTrainer; any Keras LSTM(512) under model.fit as well.)
//
// The K-split form of lstm_bwd8.hip one size up.  Workgroup `slice` (0..15) owns hidden units [32*slice, +32) for all four
// gates; lane (n, g4) of wave w owns the cells (rows 4*g4 + 2*(n>>3) + {0,1}, unit 32*slice + 8*w + (n&7)): dc and the
// recurrent dh never leave registers.  Per step t = T-1 .. 0:
//   gates backward for the lane's two cells (tape i,f,g,o,c of the training forward, requested one step ahead) -> dz, to
//   dZ (B,T,4H) for the weight-gradient products and into an LDS tile (16 x 128 own gate columns);
//   partial[16 x 512] = dz_own . R^T_own : the own gate columns' contribution to dh_{t-1} of ALL 512 units - 256 x
//   v_mfma_f32_16x16x4_f32 per wave (eight 16-unit output tiles x 32 k-steps), the 128 x 512 slice of R^T resident in
//   ALL 256 accumulation registers of the lane for the whole launch (256 KB per workgroup: the reason for sixteen);
//   the 16 x 32 piece of every destination workgroup travels as fp32 {value, epoch} granules; each lane gathers the
//   sixteen pieces of its own two cells and adds them in slice order (deterministic).
// Bias gradient: one (tiles, 4H) partial, summed over the tile's rows and all steps, as in the other BPTT kernels.
// Two group sizes of one kernel: XG = 16 as described (the default), and XG = 32 - sixteen units per workgroup, one cell per
// lane, half the matrix work per step and workgroup (128 MFMAs per wave, 128 accumulation registers of R^T) - for launches of
// at most eight tiles (FOV_BWD16_GROUPS=32; it was the faster form at lstm.py's batch until XG = 16 got paired granules).
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

namespace {

constexpr int XBT = 16;            // sequences per tile
constexpr size_t x_par(int XG, int XH) { return (size_t)XG * XBT * XH; }   // granules per parity: [dest][src][row][unit of the dest]

struct Bwd16Params {
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
    unsigned long long* xch;
    unsigned* status;
    int B, T, num_groups, num_tiles, epoch_span;
    int xcd_pad;            // grid padded to 8 x XG blocks: block b = member b / 8 of group b % 8 (same-XCD placement, lstm_wide16.hip)
    // two-layer launch (lstm_bwd16_pair_kernel): rings of T slots between the roles
    unsigned long long* ring_ax;   // upper layer -> product role: its dz slices      [tile][T][16 wg][8 row pairs][128 columns][2]
    unsigned long long* ring_xb;   // product role -> lower layer: partial dx sums    [tile][T][dest 16][src 16][8 row pairs][32 units][2]
};

__device__ __forceinline__ void x_mfma_a(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}
template <int ACT>
__device__ __forceinline__ float x_act_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

// ROLE 0: one layer per launch.  ROLES 1, 2, 3: the BPTT of TWO stacked layers as ONE launch (lstm_bwd16_pair_kernel; width 512,
// sixteen workgroups per tile and role), the lower layer about a step behind the upper one on other CUs - at lstm.py's batch
// of 32 a layer occupies 32 of 256 CUs and the two recurrences plus the data-gradient product between them (dx = dz2 . K2^T, a
// split GEMM + reduce) ran one after the other: 83 + 22 + 82 us.
//   ROLE 1  upper layer: the body of ROLE 0, and every step's own dz slice (16 x 128) also goes, tagged, into slot t of a ring;
//   ROLE 2  product: workgroup `slice` holds the same 128 x 512 slice of K2^T that the upper layer's workgroup `slice` holds
//           of R2^T (K2 and R2 have one shape) in its 256 accumulation registers, takes the dz slice of step t from the ring,
//           multiplies (the 256 MFMAs of a recurrence step) and sends the 16 x 32 pieces of dx_t to the lower layer's
//           sixteen workgroups - into slot t of a second ring, in the granule order of the recurrent exchange;
//   ROLE 3  lower layer: the body of ROLE 0 with dhs_t = the sixteen pieces of slot t added in slice order (requested a step
//           ahead, like the tape) instead of a (B,T,H) tensor read from memory.
// Rings of T slots, each written once per launch (tags base + 1 + step): a producer may run any number of steps ahead, nobody
// waits for a consumer.  Ring traffic crosses XCDs: always the placement-independent sc1 stores.
template <int ACT, int XH, int XG, int ROLE>
__device__ __forceinline__ void bwd16_body(const Bwd16Params& p, const int bx, const int pair_group) {
    constexpr int UW = XH / XG;          // units per workgroup: 32 | 16
    constexpr int NT = XH / 64;          // 16-unit output tiles per wave: 8 (width 512) | 4 | 2
    constexpr int NPASS = NT >= 8 ? 2 : 1, TQ = NT / NPASS;   // tiles per pass
    constexpr int CW = 4 * UW;           // own gate columns: 128 | 64
    constexpr int XLDZ = CW + 4;         // fp32 LDS row stride of the dz tile
    constexpr int JB = CW / 16;          // 16-column k-blocks: 8 | 4
    constexpr int CPL = XBT * UW / 256;  // cells per lane: 2 | 1
    constexpr int UB = UW / 4;           // units per wave: 8 | 4
    constexpr size_t X_PAR = x_par(XG, XH);
    __shared__ __attribute__((aligned(16))) float sDZ[XBT * XLDZ];
    __shared__ int sFlag[4];
    __shared__ unsigned sXch[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    if (ROLE != 0) {   // pair launch: the caller has split the block index (bx = member, pair_group = tile)
        group = pair_group;
        slice = bx;
    } else if (p.xcd_pad) {   // fewer than eight groups: padded grid, the blocks of the absent groups count as arrived and leave
        group = bx & 7;
        slice = bx >> 3;
        if (group >= p.num_groups) {
            xch_arrive(p.status, sXch, -1, 0);
            return;
        }
    } else if ((p.num_groups & 7) == 0) {   // members 8 blocks apart: likely one XCD (placement preference only)
        group = (bx / (8 * XG)) * 8 + (bx & 7);
        slice = (bx >> 3) & (XG - 1);
    } else {
        group = bx / XG;
        slice = bx - group * XG;
    }
    // hello words (same-XCD handshake) of the role's group: the roles of a pair launch are groups of their own
    const int hgroup = ROLE == 0 ? group : (ROLE - 1) * p.num_groups + group;
    constexpr int H4 = 4 * XH;
    const int hi = n / UB;                          // row selector inside the 4-row block of g4
    const int ul = UB * wave + (n & (UB - 1));      // unit inside the workgroup
    const int unit = UW * slice + ul;
    const int my_row0 = 4 * g4 + CPL * hi;
    const int T = p.T;
    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    const unsigned arrival = xch_arrive(p.status, sXch, hgroup, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- resident R^T fragments (ROLE 2: p.R is the upper layer's input kernel K2, same shape).  Tile tl of this wave: output units 16*(8*wave + tl) .. +16 (destination slice
    // 16*(8*wave + tl) / UW); its output unit on this lane is nout; k index lc is an own gate column: gate lc / UW, unit
    // UW*slice + lc % UW. ----
    float rt[NT][JB][4];   // [tl][jb][s], lc = 16*jb + 4*g4 + s
#pragma unroll
    for (int tl = 0; tl < NT; ++tl) {
        const int nout = 16 * (NT * wave + tl) + n;
#pragma unroll
        for (int jb = 0; jb < JB; ++jb) {
            const int lc = 16 * jb + 4 * g4;
            const f32x4 v = *(const f32x4*)(p.R + (size_t)nout * H4 + (lc / UW) * XH + UW * slice + (lc % UW));
#pragma unroll
            for (int s = 0; s < 4; ++s) rt[tl][jb][s] = v[s];
        }
    }
    // ROLE 0 / 1 / 3: the recurrent exchange's two parity areas (pair launch: the upper layer's groups, then the lower layer's)
    unsigned long long* gbase = p.xch + (size_t)((ROLE == 3 ? p.num_groups : 0) + group) * 2 * X_PAR;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, (int)(2 * X_PAR * 8), 0x00020000);
    // pair launch: this tile's rings (T slots each)
    constexpr size_t AX_SLOT = (size_t)XG * XBT * CW;   // granules of one step's dz slices
    const __amdgpu_buffer_rsrc_t rs_ax = __builtin_amdgcn_make_buffer_rsrc(
        ROLE == 0 ? gbase : p.ring_ax + (size_t)group * p.T * AX_SLOT, 0, ROLE == 0 ? 0 : (int)((size_t)p.T * AX_SLOT * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_xb = __builtin_amdgcn_make_buffer_rsrc(
        ROLE == 0 ? gbase : p.ring_xb + (size_t)group * p.T * X_PAR, 0, ROLE == 0 ? 0 : (int)((size_t)p.T * X_PAR * 8), 0x00020000);
    xch_hello_poll(p.status, sXch, hgroup, XG, &sFlag[0]);   // same-XCD handshake (xch_common.h)
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    constexpr unsigned DSTR = XG * XBT * UW * 8;     // bytes per destination slice of a partial-sum area
    constexpr unsigned SSTR = XBT * UW * 8;          // source stride inside a destination's area
    constexpr bool PAIRED = CPL == 2;
    static_assert(ROLE == 0 || PAIRED, "the pair launch is built on the two-cells-per-lane form");
    // one batch of XG 16-byte loads: the pieces of this lane's two cells in area (rsrc, soff), tags `tag`; adds them in slice order
    auto gather_pair = [&](const __amdgpu_buffer_rsrc_t& gr, unsigned soff, unsigned tag, float (&out)[2], qu32x4 (&v)[XG], bool issue,
                           bool finish) {
        const unsigned voff = (unsigned)(((slice * XG) * (XBT / 2) + 2 * g4 + hi) * UW + ul) * 16u;
        if (issue) {
#pragma unroll
            for (int s_ = 0; s_ < XG; ++s_) v[s_] = __builtin_amdgcn_raw_buffer_load_b128(gr, voff, soff + s_ * SSTR, 16);
        }
        if (!finish) return;
        float part[2][XG];
        unsigned bad = 0;
#pragma unroll
        for (int s_ = 0; s_ < XG; ++s_) {
            part[0][s_] = __uint_as_float(v[s_].x);
            part[1][s_] = __uint_as_float(v[s_].z);
            if (v[s_].y != tag || v[s_].w != tag) bad |= (1u << s_);
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
#pragma unroll
            for (int h = 0; h < XG / 8; ++h) {
                qu32x4 tv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) tv[u] = __builtin_amdgcn_raw_buffer_load_b128(gr, voff, soff + (h * 8 + u) * SSTR, 16);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int s_ = h * 8 + u;
                    if (((bad >> s_) & 1u) && tv[u].y == tag && tv[u].w == tag) {
                        part[0][s_] = __uint_as_float(tv[u].x);
                        part[1][s_] = __uint_as_float(tv[u].z);
                        bad &= ~(1u << s_);
                    }
                }
            }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float a_ = 0.f;
#pragma unroll
            for (int s_ = 0; s_ < XG; ++s_) a_ += part[q][s_];
            out[q] = a_;
        }
    };

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * XBT;
        float dc[CPL], dh[CPL];
        bool live[CPL];
#pragma unroll
        for (int r = 0; r < CPL; ++r) {
            const int row = b0 + my_row0 + r;
            live[r] = row < p.B;
            dc[r] = (ROLE != 2 && live[r] && p.dcT) ? p.dcT[(size_t)row * XH + unit] : 0.f;
            dh[r] = (ROLE != 2 && live[r] && p.dhT) ? p.dhT[(size_t)row * XH + unit] : 0.f;
        }
        // Tape of this lane's two cells, ONE step ahead (lstm_bwd8.hip): [0..3] = i,f,g,o, [4] = c of the step, [5] = c of
        // the step before it, [6] = dhs of the step.  Unconditional loads, rows and steps clamped, dead rows masked at use.
        float cur[7][CPL], pre[7][CPL];
        auto load_step = [&](int t, float (&dst)[7][CPL]) {
            const int tc = t > 0 ? t : 0;
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const int row = b0 + my_row0 + r;
                const size_t rowc = (size_t)(row < p.B ? row : p.B - 1);
                const float* rp = p.reserve + ((rowc * T + tc) * 5) * XH + unit;
#pragma unroll
                for (int q = 0; q < 5; ++q) dst[q][r] = rp[q * XH];
                const float* cp = tc > 0 ? rp - XH : (p.c0 ? p.c0 + rowc * XH + unit : rp);   // no c0: any valid address, masked at use
                dst[5][r] = *cp;
                if (ROLE != 3) dst[6][r] = p.dhs ? p.dhs[(rowc * T + tc) * XH + unit] : 0.f;   // ROLE 3: from the ring (below)
            }
        };
        if (ROLE != 2) load_step(T - 1, cur);
        if constexpr (ROLE == 3) {   // dhs of the first step: the product role's pieces of slot T - 1 (tag base + 1)
            if constexpr (PAIRED) {
                qu32x4 vx[XG];
                float o2[2];
                gather_pair(rs_xb, (unsigned)(T - 1) * (unsigned)(X_PAR * 8), epoch + 1u, o2, vx, true, true);
                cur[6][0] = o2[0];
                cur[6][CPL - 1] = o2[1];
            }
        }
        float dbacc[4] = {0.f, 0.f, 0.f, 0.f};   // sum over t and this lane's 2 sequences of dz, per gate
        __syncthreads();   // the previous tile's last step is done with the dz tile

        for (int t = T - 1; t >= 0; --t) {
            ++epoch;
            const unsigned par = (epoch & 1u) * (unsigned)(X_PAR * 8);
            qu32x4 vx[ROLE == 3 ? XG : 1];   // ROLE 3: the pieces of dhs_{t-1}, in flight under the matrix work of this step
            if constexpr (ROLE != 2) {
                // ---- pointwise: dz of this lane's cells ----
                float dzv[CPL][4];
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    const float ig = cur[0][r], fg = cur[1][r], gg = cur[2][r], og = cur[3][r], cc = cur[4][r];
                    const float cprev = (t > 0 || p.c0) ? cur[5][r] : 0.f;   // step 0 without a given state: c_{-1} = 0
                    const float dht = dh[r] + cur[6][r];
                    const float tc = tanh_f(cc);
                    const float dcv = dc[r] + dht * og * (1.f - tc * tc);
                    dzv[r][0] = live[r] ? dcv * gg * x_act_grad<ACT>(ig) : 0.f;
                    dzv[r][1] = live[r] ? dcv * cprev * x_act_grad<ACT>(fg) : 0.f;
                    dzv[r][2] = live[r] ? dcv * ig * (1.f - gg * gg) : 0.f;
                    dzv[r][3] = live[r] ? dht * tc * x_act_grad<ACT>(og) : 0.f;
                    dc[r] = dcv * fg;
                }
                if constexpr (ROLE == 1) {   // the product role waits for these: first
                    const unsigned offx = (unsigned)((slice * (XBT / 2) + (my_row0 >> 1)) * CW + ul) * 16u;
                    const unsigned slot = (unsigned)t * (unsigned)(AX_SLOT * 8);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const qu32x4 gr = {__float_as_uint(dzv[0][g]), epoch, __float_as_uint(dzv[CPL - 1][g]), epoch};
                        __builtin_amdgcn_raw_buffer_store_b128(gr, rs_ax, offx + g * UW * 16, slot, 16);
                    }
                }
#pragma unroll
                for (int r = 0; r < CPL; ++r) {
                    if (live[r]) {
                        float* zp = p.dz + ((size_t)(b0 + my_row0 + r) * T + t) * H4 + unit;
#pragma unroll
                        for (int g = 0; g < 4; ++g) zp[g * XH] = dzv[r][g];
                    }
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        dbacc[g] += dzv[r][g];
                        sDZ[(my_row0 + r) * XLDZ + g * UW + ul] = dzv[r][g];
                    }
                }
                // the tape of step t-1 is requested here, a whole step before its use
                load_step(t - 1, pre);
                if constexpr (ROLE == 3 && PAIRED) {   // and the pieces of dhs_{t-1} (slot t-1, tag epoch + 1; at t == 0: slot 0 again, unused)
                    float unused[2];
                    gather_pair(rs_xb, (unsigned)(t > 0 ? t - 1 : 0) * (unsigned)(X_PAR * 8), epoch + (t > 0 ? 1u : 0u), unused, vx, true, false);
                }
            } else {
                // ---- ROLE 2: the upper layer's dz slice of step t (its workgroup `slice`), ring slot t -> LDS tile ----
                const int rp2 = tid >> 5, c0 = (tid & 31) * 4;
                const unsigned voff = (unsigned)((slice * (XBT / 2) + rp2) * CW + c0) * 16u;
                const unsigned slot = (unsigned)t * (unsigned)(AX_SLOT * 8);
                qu32x4 v[4];
                unsigned bad = 0;
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_ax, voff + j * 16, slot, 16);
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (v[j].y != epoch || v[j].w != epoch) bad |= (1u << j);
                unsigned spins = 0;
                while (__any(bad != 0)) {
                    ++spins;
                    if (spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                        if (lane == 0) { xch_give_up(p.status); sFlag[0] = 1; }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
                    qu32x4 tv[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) tv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_ax, voff + j * 16, slot, 16);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (((bad >> j) & 1u) && tv[j].y == epoch && tv[j].w == epoch) { v[j] = tv[j]; bad &= ~(1u << j); }
                }
                *(f32x4*)(sDZ + (2 * rp2) * XLDZ + c0) = (f32x4){__uint_as_float(v[0].x), __uint_as_float(v[1].x), __uint_as_float(v[2].x), __uint_as_float(v[3].x)};
                *(f32x4*)(sDZ + (2 * rp2 + 1) * XLDZ + c0) = (f32x4){__uint_as_float(v[0].z), __uint_as_float(v[1].z), __uint_as_float(v[2].z), __uint_as_float(v[3].z)};
            }
            __syncthreads();   // barrier A: the dz tile is complete
            // ---- partial[16 x 512] = dz_own . R^T_own ; tile tl -> output units 16*(8*wave + tl) .. +16.  Two passes of four tiles:
            // the granules of the first pass are on their way (an sc1 store takes about a microsecond to become visible)
            // while the matrix pipe works on the second. ----
            // one address register for all stores: the wave's first destination is part of it, the tile's goes into the scalar offset
            // CPL == 2 (PAIRED): granule order [dest][src][row pair][unit][row of the pair] - the two rows of a lane's cells
            // are adjacent 8-byte granules, each with its own tag (the protocol's unit of atomicity stays 8 bytes), moved by
            // ONE 16-byte store / load: half the exchange instructions per step on both sides
            // ROLE 2: the same pieces, of dx_t, into slot t of the ring towards the lower layer (always sc1)
            const unsigned obase = ROLE == 2 ? (unsigned)t * (unsigned)(X_PAR * 8) : par;
            const unsigned off0 = PAIRED ? (unsigned)(((slice * (XBT / 2) + 2 * g4) * UW) + n) * 16u + (unsigned)wave * (16 * NT / UW) * DSTR
                                         : (unsigned)(((slice * XBT + 4 * g4) * UW) + n) * 8u + (unsigned)wave * (16 * NT / UW) * DSTR;
            const float* arow = sDZ + n * XLDZ + 4 * g4;
#pragma unroll
            for (int pass = 0; pass < NPASS; ++pass) {
                f32x4 acc[TQ];
#pragma unroll
                for (int i = 0; i < TQ; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                if constexpr (TQ == 4) asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                else asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1]));
                f32x4 a = *(const f32x4*)arow, an = a;
#pragma unroll
                for (int jb = 0; jb < JB; ++jb) {
                    if (jb + 1 < JB) an = *(const f32x4*)(arow + 16 * (jb + 1));
                    asm volatile("s_nop 1" : "+v"(a));
#pragma unroll
                    for (int tq = 0; tq < TQ; ++tq)
#pragma unroll
                        for (int s = 0; s < 4; ++s) x_mfma_a(acc[tq], a[s], rt[TQ * pass + tq][jb][s]);
                    a = an;
                }
                if constexpr (TQ == 4) asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));
                else asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
#pragma unroll
                for (int tq = 0; tq < TQ; ++tq) {
                    const int tl = TQ * pass + tq;
                    if constexpr (PAIRED) {
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp) {
                            const qu32x4 gr = {__float_as_uint(acc[tq][2 * rp]), epoch, __float_as_uint(acc[tq][2 * rp + 1]), epoch};
                            if (ROLE == 2) __builtin_amdgcn_raw_buffer_store_b128(gr, rs_xb, off0 + ((16 * tl) % UW) * 16 + rp * UW * 16, obase + (unsigned)((16 * tl) / UW) * DSTR, 16);
                            else if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off0 + ((16 * tl) % UW) * 16 + rp * UW * 16, obase + (unsigned)((16 * tl) / UW) * DSTR, 1);
                            else __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off0 + ((16 * tl) % UW) * 16 + rp * UW * 16, obase + (unsigned)((16 * tl) / UW) * DSTR, 16);
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            XCH_STORE_B64(ticket.same_xcd, ((qu32x2){__float_as_uint(acc[tq][r]), epoch}), rs,
                                          off0 + ((16 * tl) % UW) * 8 + r * UW * 8, obase + (unsigned)((16 * tl) / UW) * DSTR);
                    }
                }
            }
            if constexpr (ROLE == 3 && PAIRED) {   // dhs_{t-1}: requested before the matrix work, long landed (the upper layer runs ahead)
                float o2[2];
                gather_pair(rs_xb, (unsigned)(t > 0 ? t - 1 : 0) * (unsigned)(X_PAR * 8), epoch + (t > 0 ? 1u : 0u), o2, vx, false, true);
                pre[6][0] = o2[0];
                pre[6][CPL - 1] = o2[1];
            }
            // ---- gather the XG pieces of each of this lane's cells, add in slice order ----
            if constexpr (ROLE == 2) {
                // nothing to gather: the product role has no recurrence
            } else if constexpr (PAIRED) {
                qu32x4 v[XG];
                float o2[2];
                gather_pair(rs, par, epoch, o2, v, true, true);
                dh[0] = o2[0];
                dh[CPL - 1] = o2[1];
            } else
            {
                const unsigned voff = (unsigned)(((slice * XG) * XBT + my_row0) * UW + ul) * 8u;
                constexpr int NP = CPL * XG;              // 32 at width 512; 16 | 8 at the narrower widths
                constexpr int CH = NP < 16 ? NP : 16;     // granules per retry sweep
                float part[NP];
                unsigned bad = 0;
                {
                    qu32x2 v[NP];
#pragma unroll
                    for (int q = 0; q < CPL; ++q)
#pragma unroll
                        for (int s = 0; s < XG; ++s) v[q * XG + s] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + q * UW * 8, par + s * SSTR, 16);
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        part[j] = __uint_as_float(v[j].x);
                        if (v[j].y != epoch) bad |= (1u << j);
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
#pragma unroll
                    for (int h = 0; h < NP / CH; ++h) {   // at most sixteen granules at a time: half the registers of a full sweep
                        qu32x2 tv[CH];
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int j = h * CH + u, q = j / XG, s_ = j % XG;
                            tv[u] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + q * UW * 8, par + s_ * SSTR, 16);
                        }
#pragma unroll
                        for (int u = 0; u < CH; ++u) {
                            const int j = h * CH + u;
                            if (((bad >> j) & 1u) && tv[u].y == epoch) {
                                part[j] = __uint_as_float(tv[u].x);
                                bad &= ~(1u << j);
                            }
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < CPL; ++q) {
                    float a = 0.f;
#pragma unroll
                    for (int s = 0; s < XG; ++s) a += part[q * XG + s];
                    dh[q] = a;
                }
            }
            if constexpr (ROLE != 2) {
#pragma unroll
                for (int q = 0; q < 7; ++q)
#pragma unroll
                    for (int r = 0; r < CPL; ++r) cur[q][r] = pre[q][r];   // requested most of a step ago: long landed
            }
            __syncthreads();   // barrier B: every wave is done reading the dz tile; sFlag is uniform below
            if (sFlag[0]) { aborted = true; break; }
        }
        if (ROLE != 2 && !aborted && p.db_part) {
            // the lanes that share a unit (all g4, all row selectors) hold different rows: fold them in a fixed order
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v = dbacc[g];
#pragma unroll
                for (int m = UB; m < 16; m <<= 1) v += __shfl_xor(v, m);
                const float v1 = __shfl(v, (lane + 16) & 63), v2 = __shfl(v, (lane + 32) & 63), v3 = __shfl(v, (lane + 48) & 63);
                if (g4 == 0 && hi == 0) p.db_part[(size_t)tile * H4 + g * XH + unit] = (v + v1) + (v2 + v3);
            }
        }
        if (ROLE != 2 && !aborted) {
#pragma unroll
            for (int r = 0; r < CPL; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.dh0) p.dh0[(size_t)row * XH + unit] = dh[r];
                    if (p.dc0) p.dc0[(size_t)row * XH + unit] = dc[r];
                }
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

template <int ACT, int XH, int XG>
__global__ __launch_bounds__(256, 1) void lstm_bwd16_kernel(Bwd16Params p) {
    bwd16_body<ACT, XH, XG, 0>(p, (int)blockIdx.x, 0);
}

// Two stacked width-512 layers, one launch.  Grid 8 x 16 blocks: block b is member b / 8 of ROLE-GROUP b % 8, so that the sixteen
// members of a role-group sit 8 blocks apart - one XCD under round-robin dispatch (verified by the hello handshake), their
// recurrent exchange stays in that XCD's L2.  Role-group k = role * tiles + tile (roles: upper layer, product, lower layer);
// the blocks of the absent role-groups count as arrived and leave.
struct Bwd16Pair {
    Bwd16Params up, mid, low;   // mid: the product role (R = K of the upper layer)
};
template <int ACT>
__global__ __launch_bounds__(256, 1) void lstm_bwd16_pair_kernel(Bwd16Pair pp) {
    const int k = (int)blockIdx.x & 7, member = (int)blockIdx.x >> 3;
    const int tiles = pp.up.num_tiles;
    if (k >= 3 * tiles) {
        __shared__ unsigned sSpare[4];
        xch_arrive(pp.up.status, sSpare, -1, 0);
        return;
    }
    const int role = k / tiles, tile = k - role * tiles;
    if (role == 0) bwd16_body<ACT, 512, 16, 1>(pp.up, member, tile);
    else if (role == 1) bwd16_body<ACT, 512, 16, 2>(pp.mid, member, tile);
    else bwd16_body<ACT, 512, 16, 3>(pp.low, member, tile);
}

}  // namespace

// Width 512 at any batch; widths 128 / 256 (sixteen units per workgroup: 8 / 16 workgroups per tile) while a launch has at most
// eight tiles and one tile per group - the latency regime of model.fit at the reference's batch of 32, where the 2- and
// 4-workgroup kernels of lstm_bwd_cluster.hip spend a step's time on ONE workgroup's share of the product.
bool bwd16_takes(int B, int H) {
    const int cus = device_cu_count();
    if (H == 512) return cus >= 16;
    if (H != 128 && H != 256) return false;
    const bool off = env_knobs().no_bwd16_narrow != 0;
    const int tiles = (B + XBT - 1) / XBT;
    return !off && tiles >= 1 && tiles <= 8 && tiles <= cus / (H / 16);
}

template <int XH, int XG>
static int launch_bwd16_t(Bwd16Params& p, int act, hipStream_t stream) {
    const int max_groups = device_cu_count() / XG;
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 2 * x_par(XG, XH) * 8 > kXchBytes - kHelloBytes) { set_error("16-unit BPTT kernel: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.epoch_span = p.T * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(Bwd16Params) = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd16_kernel<FOV_ACT_HARD_SIGMOID, XH, XG> : lstm_bwd16_kernel<FOV_ACT_SIGMOID, XH, XG>;
    const bool no_pad = env_knobs().no_xcd_pad != 0;
    const int pad_max = env_knobs().xcd_pad_max;   // members per group at most
    p.xcd_pad = (!no_pad && XG <= pad_max && p.num_groups < 8 && device_cu_count() >= 8 * XG) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(p.xcd_pad ? 8 * XG : p.num_groups * XG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("16-unit BPTT launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// status word + granule buffers live at `xch_ws` (kStatusBytes + kXchBytes)
int launch_bwd16(const float* R, const float* reserve, const float* c0, const float* dhs, const float* dhT, const float* dcT,
                 float* dz, float* dh0, float* dc0, float* db_part, int B, int T, int H, int act, void* xch_ws, hipStream_t stream) {
    if (B == 0 || T == 0) return FOV_OK;
    if (((uintptr_t)R) & 15) { set_error("16-unit BPTT kernel: R must be 16-byte aligned"); return FOV_ERR_INVALID; }
    if (!bwd16_takes(B, H)) { set_error("16-unit BPTT kernel: unsupported shape B = %d, H = %d", B, H); return FOV_ERR_UNSUPPORTED; }
    Bwd16Params p = {};
    p.R = R; p.reserve = reserve; p.c0 = c0; p.dhs = dhs; p.dhT = dhT; p.dcT = dcT; p.dz = dz; p.dh0 = dh0; p.dc0 = dc0; p.db_part = db_part;
    p.B = B; p.T = T;
    p.num_tiles = (B + XBT - 1) / XBT;
    p.status = (unsigned*)xch_ws;
    p.xch = (unsigned long long*)((char*)xch_ws + kStatusBytes);
    if (H == 128) return launch_bwd16_t<128, 8>(p, act, stream);
    if (H == 256) return launch_bwd16_t<256, 16>(p, act, stream);
    // width 512: sixteen workgroups per tile.  The thirty-two-workgroup form (one cell per lane, half the MFMAs per step) was the
    // faster one at <= 8 tiles until the sixteen-workgroup form got paired granules and the same-XCD grid (lstm.py's training
    // step 0.4206 ms with 32, 0.414 ms with 16); FOV_BWD16_GROUPS=32 still selects it.
    if (env_knobs().bwd16_groups32 && p.num_tiles <= 8 && p.num_tiles <= device_cu_count() / 32) return launch_bwd16_t<512, 32>(p, act, stream);
    return launch_bwd16_t<512, 16>(p, act, stream);
}

// Two stacked width-512 layers (lstm.py's MultiRNNCell) as ONE launch: both recurrences and the data-gradient product between
// them.  At most two ... five tiles (three role-groups of sixteen workgroups per tile must be resident and the grid is eight
// role-groups wide), T >= 1, the rings and parity areas must fit the granule area.
bool bwd16_pair_shape(int B, int T, int H) {
    if (env_knobs().no_stack2 || H != 512 || B <= 0 || T < 1) return false;
    const int tiles = (B + XBT - 1) / XBT;
    if (3 * tiles > 8 || 8 * 16 > device_cu_count()) return false;
    const size_t xpar = x_par(16, 512), ax = (size_t)16 * XBT * 128;
    const size_t granules = (size_t)tiles * (4 * xpar + (size_t)T * (ax + xpar));
    return granules * 8 <= kXchBytes - kHelloBytes;
}

int launch_bwd16_pair(const float* R2, const float* K2, const float* reserve2, const float* c0_2, const float* dhs2, const float* dhT2,
                      const float* dcT2, float* dz2, float* dh0_2, float* dc0_2, float* db_part2, const float* R1, const float* reserve1,
                      const float* c0_1, const float* dhT1, const float* dcT1, float* dz1, float* dh0_1, float* dc0_1, float* db_part1,
                      int B, int T, int act, void* xch_ws, hipStream_t stream) {
    if (B == 0 || T == 0) return FOV_OK;
    if (!bwd16_pair_shape(B, T, 512)) { set_error("two-layer BPTT launch: unsupported shape B = %d, T = %d", B, T); return FOV_ERR_UNSUPPORTED; }
    if ((((uintptr_t)R1) | ((uintptr_t)R2) | ((uintptr_t)K2)) & 15) { set_error("two-layer BPTT launch: kernels must be 16-byte aligned"); return FOV_ERR_INVALID; }
    const int tiles = (B + XBT - 1) / XBT;
    Bwd16Pair pp = {};
    Bwd16Params& u = pp.up;
    u.R = R2; u.reserve = reserve2; u.c0 = c0_2; u.dhs = dhs2; u.dhT = dhT2; u.dcT = dcT2; u.dz = dz2; u.dh0 = dh0_2; u.dc0 = dc0_2; u.db_part = db_part2;
    u.B = B; u.T = T; u.num_tiles = tiles; u.num_groups = tiles; u.epoch_span = T + 2; u.xcd_pad = 0;
    u.status = (unsigned*)xch_ws;
    u.xch = (unsigned long long*)((char*)xch_ws + kStatusBytes);
    const size_t xpar = x_par(16, 512), ax = (size_t)16 * XBT * 128;
    u.ring_ax = u.xch + (size_t)tiles * 4 * xpar;                 // behind the two layers' parity areas
    u.ring_xb = u.ring_ax + (size_t)tiles * T * ax;
    pp.mid = u;
    pp.mid.R = K2; pp.mid.reserve = nullptr; pp.mid.c0 = nullptr; pp.mid.dhs = nullptr; pp.mid.dhT = nullptr; pp.mid.dcT = nullptr;
    pp.mid.dz = nullptr; pp.mid.dh0 = nullptr; pp.mid.dc0 = nullptr; pp.mid.db_part = nullptr;
    pp.low = u;
    pp.low.R = R1; pp.low.reserve = reserve1; pp.low.c0 = c0_1; pp.low.dhs = nullptr; pp.low.dhT = dhT1; pp.low.dcT = dcT1;
    pp.low.dz = dz1; pp.low.dh0 = dh0_1; pp.low.dc0 = dc0_1; pp.low.db_part = db_part1;
    if (int rc_ = xch_account(u.status, u.epoch_span, stream)) return rc_;
    void (*kern)(Bwd16Pair) = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd16_pair_kernel<FOV_ACT_HARD_SIGMOID> : lstm_bwd16_pair_kernel<FOV_ACT_SIGMOID>;
    hipLaunchKernelGGL(kern, dim3(8 * 16), dim3(256), 0, stream, pp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("two-layer BPTT launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
