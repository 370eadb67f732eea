// Width-512 persistent LSTM layer for SMALL batches: THIRTY-TWO workgroups per 16-sequence tile, 16 hidden units each
// (round 3; mycode/lstm.py:59,128-132,218-240: two LSTMCell(400) at batch 32 = two tiles, zero-padded to 512 units).
//
// lstm_wide.hip's width-512 form (16 workgroups x 32 units) spends 256 MFMAs per wave and step on h.R (3.4 us) behind an
// exchange that nothing covers, and needs layer 2's input projection from a separate GEMM (its K slice would be another 256
// registers).  With two tiles there are 224 idle CUs, so here a tile is spread over twice as many:
//   * a wave owns FOUR units x 4 gates = ONE 16-column MFMA tile [i f g o] x 4 units; R slice 512 x 16 columns = 128 AGPRs,
//     and the K slice of a 512-wide input (layer 2) fits beside it in the other 128 - no GEMM, no (B,T,4H) round trip;
//     a narrow input (F <= 96, layer 1) keeps its six k-blocks there;
//   * per step and wave 128 MFMAs of x_{t+1}.K UNDER the exchange of h_t and 128 MFMAs of h_t.R behind it (1.7 us each);
//   * the four gates of a (sequence, unit) sit in four lanes of a wave after the MFMAs: a 16 x 16 transpose through a
//     wave-private LDS scratch (four ds_write_b32, four ds_read_b32, no barrier: one wave's LDS instructions execute in
//     order) gives every lane ONE (row, unit) cell - 64 lanes = 16 rows x 4 units, nothing redundant;
//   * exchange as everywhere (xch_common.h): one {h, epoch} granule per lane and step published, 31 gathered.
// Taken by fov_lstm_seq_fwd[_train] for H = 512 while a launch has at most one tile per group (<= 8 tiles = 128 sequences on
// 256 CUs); larger batches stay on the 16-workgroup form, whose groups are half as many CUs.
#include <stdlib.h>

#include "fov_common.h"
#include "xch_common.h"

namespace fov {

namespace {

constexpr int VBT = 16;
constexpr unsigned VSPIN = 1u << 20;

// granule offsets of the three-role launch's extra areas behind layer 1's ring (T slots per group) and layer 2's parity slots
// (2 per group): the mailboxes (T slots of W16_MAIL granules per workgroup of layer 2), then the mirror rings (T slots per group)
__host__ __device__ constexpr size_t w16_mail_base(int num_groups, int T, int WH) { return (size_t)num_groups * (T + 2) * 16 * WH; }
// `roles`: LstmParams::trio - 3: three-role launch (mailboxes in front of the mirror rings), 2: two roles (no mailboxes)
__host__ __device__ constexpr size_t w16_mirror_base(int num_groups, int T, int WH, int roles) {
    return w16_mail_base(num_groups, T, WH) + (roles == 3 ? (size_t)num_groups * (WH / 16) * T * (256 * 4) : (size_t)0);
}
constexpr size_t W16_MAIL = 256 * 4;   // granules of one mailbox slot of the three-role launch: 256 lanes x (4 values, each with its tag)
typedef unsigned vu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned vu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void vm_a(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}
__device__ __forceinline__ void vm_begin(f32x4 (&acc)[2]) { asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void vm_end(f32x4 (&acc)[2]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}

// acc[0] + acc[1] += A(tile rows in LDS, k-blocks [0, NJ)) . W (AGPR resident).  TWO accumulators, even / odd k-steps: with a
// single one hipcc shifted the accumulator tuple between two MFMAs of a run (v_mov_b64 of a result two slots after the MFMA
// that writes it: tools/isa_mfma_hazard.py), the alternating form of lstm_wide.hip compiles clean.  Fragments run two
// k-blocks ahead (4 MFMAs = 128 cycles per block do not cover an LDS round trip).
template <int NJ>
__device__ __forceinline__ void wide16_mm(f32x4 (&acc)[2], const float* arow, const float (&w)[NJ][4]) {
    f32x4 a0 = *(const f32x4*)arow;
    f32x4 a1 = NJ > 1 ? *(const f32x4*)(arow + 16) : a0;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        f32x4 a2 = a1;
        if (j + 2 < NJ) a2 = *(const f32x4*)(arow + 16 * (j + 2));
        asm volatile("s_nop 1" : "+v"(a0));   // the fragment may have been moved by the compiler (VALU copy)
#pragma unroll
        for (int s = 0; s < 4; ++s) vm_a(acc[s & 1], a0[s], w[j][s]);
        a0 = a1;
        a1 = a2;
    }
}

// ---- resident weights through LDS (round 4, late; the sixteen-unit twin of stage_f32.h) ------------------------------
// A lane's fragments are w[j][s] = W[16 j + 4 g4 + s][gate (n / 4) * WH + 16 slice + 4 wave + n % 4]: asked for directly, a wave's
// load instruction touches SIXTEEN cache lines for 256 bytes (four gates x four rows, 16 bytes of each) and a thread issues 128
// dword loads per (512, 2048) matrix.  Here the workgroup reads its slice - 4 gates x 16 units = four 64-byte pieces per matrix
// row - as dwordx4 loads (a stage is 32 rows: two loads per thread), writes the values column-major into LDS (40 words per
// column: 16-byte aligned, 8 mod 32; the row index is XOR-swizzled with the column's gate and bit 2 so that the 64 lanes of a
// ds_write_b32 cover all banks twice) and every lane picks its fragments up as ds_read_b128 (8 lanes per bank quad).
// W16_ST_DEPTH stages are requested ahead of the one being written; two LDS buffers (the h / x tiles' space), a barrier per stage.
constexpr int W16_ST_WORDS = 40;
constexpr int W16_ST_BUF = 64 * W16_ST_WORDS;
constexpr int W16_ST_LDS_WORDS = 2 * W16_ST_BUF;   // 20 480 bytes
constexpr int W16_ST_DEPTH = 4;
struct W16StageLane {
    unsigned goff;     // byte offset of (row kr, this thread's 4 columns) in the matrix
    unsigned wr[2];    // LDS word of the thread's first column for the two 16-row halves of a stage (swizzle folded in)
    unsigned rd;       // LDS word of the lane's fragment of 16-row block 0 of a stage (before the half swizzle)
    unsigned rdx;      // 1: the lane's column reads the halves swapped
};
template <int WH>
__device__ __forceinline__ W16StageLane w16_stage_lane(int slice) {
    const int tid = threadIdx.x;
    const int c = tid & 3, gate = (tid >> 2) & 3, kr = tid >> 4;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 15, g4 = lane >> 4;
    W16StageLane q;
    q.goff = (unsigned)(kr * 4 * WH + gate * WH + 16 * slice + 4 * c) * 4u;
    const unsigned wbase = (unsigned)((gate * 16 + 4 * c) * W16_ST_WORDS + (kr ^ (gate << 2)));
    q.wr[0] = wbase + 16u * (unsigned)(c & 1);
    q.wr[1] = wbase + 16u * (unsigned)((c & 1) ^ 1);
    const int gate_r = n >> 2;
    q.rd = (unsigned)((gate_r * 16 + 4 * wave + (n & 3)) * W16_ST_WORDS + 4 * (g4 ^ gate_r));
    q.rdx = (unsigned)(wave & 1);
    return q;
}
template <int WH>
__device__ __forceinline__ void w16_stage_issue(vu32x4 (&r)[2], const float* __restrict__ W, int nrows, int ls, const W16StageLane& q) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * 4 * WH * 4, 0x00020000);
#pragma unroll
    for (int it = 0; it < 2; ++it)   // rows >= nrows read as zero (the whole offset sits in the vector register)
        r[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, q.goff + (unsigned)((32 * ls + 16 * it) * 4 * WH) * 4u, 0, 0);
}
__device__ __forceinline__ void w16_stage_write(const vu32x4 (&r)[2], unsigned* buf, const W16StageLane& q) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        unsigned* wp = buf + q.wr[it];
#pragma unroll
        for (int j = 0; j < 4; ++j) wp[j * W16_ST_WORDS] = r[it][j];
    }
}
template <int NJ>
__device__ __forceinline__ void w16_stage_read(float (&w)[NJ][4], int ls, const unsigned* buf, const W16StageLane& q) {
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
        if (2 * ls + jl < NJ) {
            const vu32x4 v = *(const vu32x4*)(buf + q.rd + 16u * ((unsigned)jl ^ q.rdx));
#pragma unroll
            for (int s = 0; s < 4; ++s) w[2 * ls + jl][s] = __uint_as_float(v[s]);
        }
}
// one (HAS_A false) or two fragment sets in one pipeline; sStage: W16_ST_LDS_WORDS words, 16-byte aligned, free again at return
template <int WH, bool HAS_A, int NA, int NB>
__device__ __forceinline__ void w16_stage_weight_sets(float (&a)[NA][4], const float* __restrict__ Wa, int nra, float (&b)[NB][4],
                                                      const float* __restrict__ Wb, int nrb, int slice, unsigned* sStage) {
    constexpr int SA = HAS_A ? (NA + 1) / 2 : 0, SB = (NB + 1) / 2, NS = SA + SB, D = W16_ST_DEPTH;
    const W16StageLane q = w16_stage_lane<WH>(slice);
    vu32x4 r[D + 1][2];
    auto issue = [&](int s_) __attribute__((always_inline)) {
        if (s_ < SA) w16_stage_issue<WH>(r[s_ % (D + 1)], Wa, nra, s_, q);
        else w16_stage_issue<WH>(r[s_ % (D + 1)], Wb, nrb, s_ - SA, q);
    };
#pragma unroll
    for (int s_ = 0; s_ < D; ++s_)
        if (s_ < NS) issue(s_);
#pragma unroll
    for (int s_ = 0; s_ < NS; ++s_) {
        if (s_ + D < NS) issue(s_ + D);
        unsigned* buf = sStage + (s_ & 1) * W16_ST_BUF;
        w16_stage_write(r[s_ % (D + 1)], buf, q);
        __syncthreads();   // (buffer s & 1 is written again two stages on, behind the barrier of stage s + 1, which a wave passes after these reads)
        if (s_ < SA) w16_stage_read<NA>(a, s_, buf, q);
        else w16_stage_read<NB>(b, s_ - SA, buf, q);
    }
    __syncthreads();
}

// NJX: k-blocks of K in registers - 6 (narrow input, F <= 96, scalar x stage) or WH / 16 (input as wide as the layer, 16-byte
// x pieces)
// ROLE 0: one layer per launch.  ROLE 1 / 2: the two layers of a stack as ONE launch (lstm_wide16_pair_kernel), layer 2 a
// few steps behind layer 1 on other CUs - with two tiles a layer occupies 64 of 256 CUs.  Layer 1 (ROLE 1) publishes h_t of
// EVERY step into a ring of T slots (its partners gather from slot t as well); layer 2 (ROLE 2) takes its input from those
// granules - x_t of layer 2 IS h_t of layer 1, already travelling as {value, epoch} - two steps ahead of its use, and runs its
// own exchange in a parity area behind the ring with tags base + T + 1 + t.  `bx`: block index inside the role.
template <int ACT, int NJX, int WH, int ROLE>
__device__ __forceinline__ void wide16_body(const LstmParams& p, const int bx, float* smem) {
    constexpr int WG = WH / 16;          // workgroups per tile
    constexpr int NJR = WH / 16;         // k-blocks of R
    constexpr bool L2 = ROLE == 2 || ROLE == 4;   // layer 2 of a stack (ROLE 4: its input projection arrives from the product role)
    constexpr bool XVEC = !L2 && NJX == WH / 16;
    constexpr int WLD = WH + 8;          // LDS row stride == 8 (mod 16) floats: conflict-free ds_read_b128 fragments
    constexpr int WNG = WG / 2;          // 16-byte loads (two adjacent units' tagged granules) per thread and step
    constexpr int H4 = 4 * WH;
    constexpr unsigned OORB = 0x80000000u;
    static_assert((WG & (WG - 1)) == 0 && WG <= kHelloStride, "slice arithmetic, hello words");
    float* sH = smem;                     // [16][WLD]
    float* sX = sH + VBT * WLD;           // [2][16][WLD]
    float* sT = sX + 2 * VBT * WLD;       // [4 waves][16][17] gate transpose
    int* sFlag = (int*)(sT + 4 * 16 * 17);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    if (ROLE == 0 && p.xcd_pad) {
        // Fewer than eight groups: the grid is padded to 8 x WG blocks, block b is member b / 8 of group b % 8, so that the
        // members of a group sit 8 blocks apart - one XCD under round-robin dispatch, verified by the hello handshake - and
        // exchange through its L2 (sc0).  The blocks of the absent groups count as arrived and leave.
        group = bx & 7;
        slice = bx >> 3;
        if (group >= p.num_groups) {
            __shared__ unsigned sSpare[4];
            if (p.T > 1) xch_arrive(p.status, sSpare, -1, 0);
            return;
        }
    } else if (ROLE == 0 && (p.num_groups & 7) == 0) {   // members 8 blocks apart: likely one XCD (placement preference only)
        group = (bx / (8 * WG)) * 8 + (bx & 7);
        slice = (bx >> 3) & (WG - 1);
    } else {
        group = bx / WG;
        slice = bx - group * WG;
    }
    const int hgroup = group + (L2 ? p.num_groups : 0);   // hello words: layer 2's groups behind layer 1's
    const int F = p.F, steps = p.T;
    // MFMA column of this lane: gate n / 4 of unit n % 4 of the wave's four units
    const int unit = 16 * slice + 4 * wave + (n & 3);
    const int col = (n >> 2) * WH + unit;
    // the cell this lane updates after the transpose: row 4 g4 + n / 4 of the tile, the same unit
    const int row_o = 4 * g4 + (n >> 2);
    float* tw = sT + wave * (16 * 17);
    const bool xch_used = ROLE != 0 || steps > 1;
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_used ? xch_arrive(p.status, sXch, hgroup, slice) : 0u;
    const bool poisoned = xch_used && xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- resident weights through LDS (w16_stage_weight_sets above; the staging buffers are the h / x tiles, filled after it):
    // K rows >= F read as zero (the descriptor ends with row F - 1) ----
    float wk[NJX > 0 ? NJX : 1][4], wr[NJR][4];
    static_assert(3 * VBT * WLD >= W16_ST_LDS_WORDS, "the h tile and the two x tiles hold the two staging buffers");
    w16_stage_weight_sets<WH, (NJX > 0)>(wk, p.K, F, wr, p.R, WH, slice, (unsigned*)smem);
    const float bv = p.b[col];
    for (int i = tid; i < 2 * VBT * WLD; i += 256) sX[i] = 0.f;   // columns >= F stay zero

    // ---- exchange bookkeeping ----
    constexpr size_t SLOT = (size_t)VBT * WH;   // granules of one h tile
    // ROLE 0: two parity slots per group.  Pair: layer 1's ring of T slots per group, then layer 2's parity slots.
    const size_t xbase = ROLE == 0 ? (size_t)group * 2 * SLOT
                         : ROLE == 1 ? (size_t)group * p.T * SLOT : ((size_t)p.num_groups * p.T + (size_t)group * 2) * SLOT;
    // three-role launch, ROLE 1: the mirror of this group's ring for readers on OTHER XCDs (the product role) - the ring itself
    // may be written with sc0 stores that stay in this XCD's L2
    const __amdgpu_buffer_rsrc_t mirrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + w16_mirror_base(p.num_groups, p.T, WH, p.trio) + (size_t)group * p.T * SLOT, 0,
        (ROLE == 1 && p.trio) ? p.T * (int)(SLOT * sizeof(unsigned long long)) : 0, 0x00020000);
    // ROLE 4: this workgroup's mailbox - T slots of 256 lanes x 32 bytes, written by its partner of the product role
    // (wide16_product_body) behind layer 1's ring and layer 2's parity slots
    const __amdgpu_buffer_rsrc_t mbrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + w16_mail_base(p.num_groups, p.T, WH) + ((size_t)group * WG + slice) * p.T * W16_MAIL, 0,
        ROLE == 4 ? p.T * (int)(W16_MAIL * sizeof(unsigned long long)) : 0, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + xbase, 0, (ROLE == 1 ? p.T : 2) * (int)(SLOT * sizeof(unsigned long long)), 0x00020000);
    const __amdgpu_buffer_rsrc_t ringrs = __builtin_amdgcn_make_buffer_rsrc(   // ROLE 2: layer 1's ring of the same group (XCD-placed launch: its mirror)
        p.xch + (p.trio ? w16_mirror_base(p.num_groups, p.T, WH, p.trio) : (size_t)0) + (size_t)group * p.T * SLOT, 0,
        ROLE == 2 ? p.T * (int)(SLOT * sizeof(unsigned long long)) : 0, 0x00020000);
    const unsigned pub_off = (unsigned)(row_o * WH + unit) * 8u;
    // gather: thread (row tid / 16, half (tid / 8) % 2, unit pair tid % 8) brings units (2p, 2p + 1) of partner slices
    // slice + 1 + half * WG/2 + j, j < WG/2, with one 16-byte load each (round 3: half the gather instructions; every 8-byte
    // granule keeps its own tag).  The last load of half 1 would be the own slice: switched off (offset past the descriptor).
    const int grow = tid >> 4, ghalf = (tid >> 3) & 1, gp = tid & 7;
    const unsigned gvoff = (unsigned)(grow * WH + 2 * gp) * 8u;
    const int lbase = grow * WLD + 2 * gp;
    constexpr unsigned PARITY = VBT * WH * 8u;
    if (xch_used) xch_hello_poll(p.status, sXch, hgroup, WG, &sFlag[0]);
    __syncthreads();
    XchTicket ticket = {0u, 0u, 0u};
    if (xch_used) ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base + (L2 ? (unsigned)p.T : 0u);   // layer 2's own tags follow layer 1's T
    bool aborted = sFlag[0] != 0;
    if (xch_used && tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);

    vu32x4 v[WNG];
    auto gslice = [&](int j) { return (slice + 1 + ghalf * WNG + j) & (WG - 1); };
    auto gather_issue = [&](unsigned base) {
#pragma unroll
        for (int j = 0; j < WNG; ++j) {
            const bool on = !(ghalf == 1 && j == WNG - 1);
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, on ? gvoff + (unsigned)(gslice(j) * 16) * 8u : OORB, base, 16);
        }
    };
    auto gather_finish = [&](unsigned base) {
        unsigned bad = 0;
#pragma unroll
        for (int j = 0; j < WNG; ++j) {
            const bool on = !(ghalf == 1 && j == WNG - 1);
            const int lo = lbase + gslice(j) * 16;
            if (!on) continue;
            if (v[j].y == epoch && v[j].w == epoch) {
                sH[lo] = __uint_as_float(v[j].x);
                sH[lo + 1] = __uint_as_float(v[j].z);
            } else {
                bad |= (1u << j);
            }
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > VSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) {
                    xch_give_up(p.status);
                    sFlag[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            constexpr int RCH = WNG < 8 ? WNG : 8;   // retry in chunks (register budget)
#pragma unroll
            for (int j0 = 0; j0 < WNG; j0 += RCH) {
                vu32x4 tv[RCH];
#pragma unroll
                for (int j = j0; j < j0 + RCH; ++j) {
                    const bool on = !(ghalf == 1 && j == WNG - 1);
                    tv[j - j0] = __builtin_amdgcn_raw_buffer_load_b128(xrs, on ? gvoff + (unsigned)(gslice(j) * 16) * 8u : OORB, base, 16);
                }
#pragma unroll
                for (int j = j0; j < j0 + RCH; ++j) {
                    const int lo = lbase + gslice(j) * 16;
                    if (((bad >> j) & 1u) && tv[j - j0].y == epoch && tv[j - j0].w == epoch) {
                        sH[lo] = __uint_as_float(tv[j - j0].x);
                        sH[lo + 1] = __uint_as_float(tv[j - j0].z);
                        bad &= ~(1u << j);
                    }
                }
            }
        }
    };

    // ROLE 2: thread (row tid / 16, c = tid % 16) gathers the unit pairs c + 16 i of layer 1's h tile of step s (slot s, tags
    // base + 1 + s), one 16-byte load per pair
    constexpr int NRG = WH / 32;
    vu32x4 xg[ROLE == 2 ? NRG : 1];
    auto ring_issue = [&](int s_) {
        if constexpr (ROLE == 2) {
            const unsigned vo = (unsigned)(((tid >> 4) * WH + 2 * (tid & 15)) * 8);
#pragma unroll
            for (int i = 0; i < NRG; ++i) xg[i] = __builtin_amdgcn_raw_buffer_load_b128(ringrs, vo + i * 32 * 8, (unsigned)s_ * PARITY, 16);
        }
    };
    auto ring_store = [&](int s_, float* dst) {   // dst: the LDS tile's element [row tid / 16][2 (tid % 16)]
        if constexpr (ROLE == 2) {
            const unsigned want = ticket.base + 1u + (unsigned)s_;
            const unsigned vo = (unsigned)(((tid >> 4) * WH + 2 * (tid & 15)) * 8);
            unsigned bad = 0;
#pragma unroll
            for (int i = 0; i < NRG; ++i) {
                if (xg[i].y == want && xg[i].w == want) {
                    dst[32 * i] = __uint_as_float(xg[i].x);
                    dst[32 * i + 1] = __uint_as_float(xg[i].z);
                } else {
                    bad |= (1u << i);
                }
            }
            unsigned spins = 0;
            while (__any(bad != 0)) {
                ++spins;
                if (spins > VSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                    if (lane == 0) {
                        xch_give_up(p.status);
                        sFlag[0] = 1;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(2);
                asm volatile("" ::: "memory");
#pragma unroll
                for (int i0 = 0; i0 < NRG; i0 += 8) {
                    vu32x4 tv[8];
#pragma unroll
                    for (int i = 0; i < 8; ++i) tv[i] = __builtin_amdgcn_raw_buffer_load_b128(ringrs, vo + (i0 + i) * 32 * 8, (unsigned)s_ * PARITY, 16);
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        if (((bad >> (i0 + i)) & 1u) && tv[i].y == want && tv[i].w == want) {
                            dst[32 * (i0 + i)] = __uint_as_float(tv[i].x);
                            dst[32 * (i0 + i) + 1] = __uint_as_float(tv[i].z);
                            bad &= ~(1u << (i0 + i));
                        }
                }
            }
        }
    };

    // ROLE 4: z_s = h1_s . K2 of this lane's D-fragment positions (rows 4 g4 + r, column n), two tagged 16-byte pieces
    vu32x4 zq[2];
    auto mail_issue = [&](int s_) {
        if constexpr (ROLE == 4) {
#pragma unroll
            for (int i = 0; i < 2; ++i) zq[i] = __builtin_amdgcn_raw_buffer_load_b128(mbrs, (unsigned)(tid * 32 + 16 * i), (unsigned)s_ * (W16_MAIL * 8u), 16);
        }
    };
    auto mail_wait = [&](int s_) {   // -> the four values; a give-up poisons the workspace and flags the workgroup
        f32x4 z = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ROLE == 4) {
            const unsigned want = ticket.base + 1u + (unsigned)s_;
            unsigned spins = 0;
            while (__any(zq[0].y != want || zq[0].w != want || zq[1].y != want || zq[1].w != want)) {
#ifdef FOV_DBG_W16_NOZWAIT   // timing experiment (tools/w16_variants.sh): WRONG results
                break;
#endif
                ++spins;
                if (spins > VSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                    if (lane == 0) {
                        xch_give_up(p.status);
                        sFlag[0] = 1;
                    }
                    break;
                }
                __builtin_amdgcn_s_sleep(1);
                asm volatile("" ::: "memory");
                mail_issue(s_);
            }
            z = (f32x4){__uint_as_float(zq[0].x), __uint_as_float(zq[0].z), __uint_as_float(zq[1].x), __uint_as_float(zq[1].z)};
        }
        return z;
    };

    const float* hrow = sH + n * WLD + 4 * g4;
    // x staging: thread (xrw = tid / 16, xc = tid % 16) moves the 16-byte pieces xc, xc + 16, ... of row xrw (narrow: elements)
    const int xrw = tid >> 4, xc = tid & 15;
    const int nx4 = F >> 2;
    constexpr int NXR = XVEC ? WH / 64 : (NJX > 0 ? NJX : 1);
    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * VBT;
        __syncthreads();   // previous tile fully consumed
        // every global load of the tile goes through a descriptor that covers exactly its live rows: rows past the batch and
        // absent tensors read as 0 without a branch
        const int live_rows = p.B - b0 < VBT ? p.B - b0 : VBT;
        const __amdgpu_buffer_rsrc_t h0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h0 ? p.h0 + (size_t)b0 * WH : nullptr), 0, p.h0 ? live_rows * WH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t c0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.c0 ? p.c0 + (size_t)b0 * WH : nullptr), 0, p.c0 ? live_rows * WH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.x + (size_t)b0 * p.T * F), 0, live_rows * p.T * F * 4, 0x00020000);
        {
            float hv[VBT * WH / 256];
#pragma unroll
            for (int q = 0; q < VBT * WH / 256; ++q) hv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, (unsigned)((tid + 256 * q) * 4), 0, 0));
#pragma unroll
            for (int q = 0; q < VBT * WH / 256; ++q) {
                const int e = tid + 256 * q;
                sH[(e / WH) * WLD + (e % WH)] = hv[q];
            }
        }
        const unsigned soff = (unsigned)((row_o * WH + unit) * 4);
        float c = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, soff, 0, 0));
        float hc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, soff, 0, 0));
        unsigned xoff[NXR];
#pragma unroll
        for (int i = 0; i < NXR; ++i) {
            if constexpr (XVEC) xoff[i] = (xc + 16 * i < nx4) ? (unsigned)((xrw * p.T * F + 4 * xc + 64 * i) * 4) : OORB;
            else xoff[i] = (xc + 16 * i < F) ? (unsigned)((xrw * p.T * F + xc + 16 * i) * 4) : OORB;
        }
        auto load_x4 = [&](int i, int t) {
            const vu32x4 q = __builtin_amdgcn_raw_buffer_load_b128(xgrs, xoff[i], (unsigned)(t * F * 4), 0);
            return (f32x4){__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3])};
        };
        auto load_x1 = [&](int i, int t) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, xoff[i], (unsigned)(t * F * 4), 0)); };
        float* xl = sX + xrw * WLD + (ROLE == 2 ? 2 : XVEC ? 4 : 1) * xc;   // ROLE 2: unit pairs from the ring (ring_store)
        if constexpr (ROLE == 4) mail_issue(0);
        const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
        if constexpr (ROLE == 2) {   // x_0 and x_1 = layer 1's h_0, h_1: wait for them (layer 2 starts two steps behind)
            ring_issue(0);
            ring_store(0, xl);
            if (steps > 1) {
                ring_issue(1);
                ring_store(1, xl + VBT * WLD);
            }
        } else if constexpr (ROLE != 4) {
            f32x4 x4[2][XVEC ? NXR : 1];
            float x1[2][XVEC ? 1 : NXR];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int i = 0; i < NXR; ++i) {
                    const int tc = tt < steps ? tt : 0;
                    if constexpr (XVEC) x4[tt][i] = load_x4(i, tc);
                    else x1[tt][i] = load_x1(i, tc);
                }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (tt < steps) {
#pragma unroll
                    for (int i = 0; i < NXR; ++i) {
                        if constexpr (XVEC) {
                            if (xc + 16 * i < nx4) *(f32x4*)(xl + tt * VBT * WLD + 64 * i) = x4[tt][i];
                        } else {
                            if (xc + 16 * i < F) xl[tt * VBT * WLD + 16 * i] = x1[tt][i];
                        }
                    }
                }
        }
        __syncthreads();
        // ---- pre-activations of step 0 ----
        f32x4 acc[2] = {{bv, bv, bv, bv}, {0.f, 0.f, 0.f, 0.f}};
        if (steps > 0) {
            vm_begin(acc);
            if constexpr (NJX > 0) wide16_mm<NJX>(acc, sX + n * WLD + 4 * g4, wk);
            wide16_mm<NJR>(acc, hrow, wr);
            vm_end(acc);
        }
        f32x4 xr[XVEC ? NXR : 1] = {z4};
        float xs[XVEC ? 1 : NXR] = {0.f};
        for (int t = 0; t < steps; ++t) {
            // x pipeline: x_{t+1} (requested during step t-1) registers -> LDS; then request x_{t+2}
            if constexpr (ROLE == 2) {
                // layer 1's h_{t+1}, requested late in step t-1 (behind the partner gather: the two sets of granule registers
                // never live together): registers -> LDS, waiting here if layer 1 is not that far yet
                if (t > 0 && t + 1 < steps) ring_store(t + 1, xl + ((t + 1) & 1) * VBT * WLD);
            } else if constexpr (ROLE != 4) {
                if (t > 0 && t + 1 < steps) {
                    float* xb = xl + ((t + 1) & 1) * VBT * WLD;
#pragma unroll
                    for (int i = 0; i < NXR; ++i) {
                        if constexpr (XVEC) {
                            if (xc + 16 * i < nx4) *(f32x4*)(xb + 64 * i) = xr[i];
                        } else {
                            if (xc + 16 * i < F) xb[16 * i] = xs[i];
                        }
                    }
                }
                if (t + 2 < steps) {
#pragma unroll
                    for (int i = 0; i < NXR; ++i) {
                        if constexpr (XVEC) xr[i] = load_x4(i, t + 2);
                        else xs[i] = load_x1(i, t + 2);
                    }
                }
            }
            // ---- the four gates of a cell meet: 16 x 16 transpose through the wave's scratch ----
            if constexpr (ROLE == 4) {   // + h1_t . K2 from the product role (requested a step ago); then the request of the next one
                const f32x4 z = mail_wait(t);
                acc[1] += z;
                if (t + 1 < steps) mail_issue(t + 1);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) tw[(4 * g4 + r) * 17 + n] = acc[0][r] + acc[1][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // one wave's LDS instructions execute in order; the compiler must not reorder either
            const float zi = tw[row_o * 17 + (n & 3)], zf = tw[row_o * 17 + 4 + (n & 3)], zg = tw[row_o * 17 + 8 + (n & 3)],
                        zo = tw[row_o * 17 + 12 + (n & 3)];
            {
                const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
                c = fmaf(fg, c, ig * gg);
                hc = og * tanh_f(c);
                const int row = b0 + row_o;
                if (row < p.B) {
                    if (p.reserve) {
                        float* rp = p.reserve + (((size_t)row * p.T + t) * 5) * WH + unit;
                        rp[0] = ig; rp[WH] = fg; rp[2 * WH] = gg; rp[3 * WH] = og; rp[4 * WH] = c;
                    }
                    if (p.hs) p.hs[((size_t)row * p.T + t) * WH + unit] = hc;
                }
            }
            const bool more = (t + 1 < steps);
            const bool do_xch = xch_used && more;              // gather the partners' pieces of h_t
            const bool pub = ROLE == 1 ? true : do_xch;        // layer 1 of a pair publishes its last step too: layer 2 reads it
            unsigned par = 0;
            if (pub) {
                ++epoch;
                par = ROLE == 1 ? (unsigned)t * PARITY : (epoch & 1u) * PARITY;
                XCH_STORE_B64(ticket.same_xcd, ((vu32x2){__float_as_uint(hc), epoch}), xrs, pub_off, par);
                if constexpr (ROLE == 1) {   // (no descriptor range outside the three-role launch: the store is dropped)
                    __builtin_amdgcn_raw_buffer_store_b64(((vu32x2){__float_as_uint(hc), epoch}), mirrs, pub_off, par, 16 /* sc1 */);
                }
            }
            __syncthreads();   // barrier 1: every wave is done reading sH; x_{t+1} is in LDS
            if (do_xch) sH[row_o * WLD + unit] = hc;
            acc[0] = (f32x4){bv, bv, bv, bv};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if constexpr (NJX > 0) {
                if (more) {   // x_{t+1} . K needs no remote data: it runs under the exchange
                    vm_begin(acc);
                    wide16_mm<NJX>(acc, sX + ((t + 1) & 1) * VBT * WLD + n * WLD + 4 * g4, wk);
                    vm_end(acc);
                }
            }
            if (do_xch) gather_issue(par);
#if defined(FOV_DBG_W16_NOL2GATHER)   // timing experiments (tools/w16_variants.sh): WRONG results
            if (do_xch && ROLE != 4) gather_finish(par);
#elif defined(FOV_DBG_W16_NOL1GATHER)
            if (do_xch && ROLE != 1) gather_finish(par);
#else
            if (do_xch) gather_finish(par);
#endif
            __syncthreads();   // barrier 2: the whole h_t tile is in LDS
            if (sFlag[0]) { aborted = true; break; }
            if constexpr (ROLE == 2) {
                if (t + 2 < steps) ring_issue(t + 2);   // layer 1's h_{t+2}: consumed at the start of the next step
            }
            if (more) {
                vm_begin(acc);
                wide16_mm<NJR>(acc, hrow, wr);
                vm_end(acc);
            }
        }
        if (!aborted) {
            const int row = b0 + row_o;
            if (row < p.B) {
                if (p.hT) p.hT[(size_t)row * WH + unit] = hc;
                if (p.cT) p.cT[(size_t)row * WH + unit] = c;
            }
        }
    }
    if (xch_used) xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

template <int ACT, int NJX, int WH>
__global__ __launch_bounds__(256, 1) void lstm_wide16_kernel(LstmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    wide16_body<ACT, NJX, WH, 0>(p, (int)blockIdx.x, smem);
}

// Two stacked layers, one launch: the first num_groups * WG blocks are layer 1 (narrow input, six k-blocks of K), the rest layer 2
struct Wide16Pair {
    LstmParams l1, l2;
};

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the PRODUCT role of the three-role launch (lstm_wide16_trio_kernel).  In the two-role launch a wave of layer 2
// issues 256 MFMAs per step - 128 for x_t . K2 (x_t = layer 1's h_t) and 128 for h_{t-1} . R2 - on ONE matrix pipe:
// 3.9 us of the 6.75 us a step takes (tools/a10_prologue_probe.py), with half of the chip idle at lstm.py's batch of 32
// (2 tiles x 32 workgroups x 2 layers = 128 CUs).  x_t . K2 does not depend on layer 2's recurrence: here a third set of
// workgroups, one per workgroup of layer 2 and with the same (group, slice, wave, lane) -> gate column map, keeps the K2
// slice in its accumulation registers, gathers layer 1's h_t tile from the ring as layer 2 used to, forms
// z_t = h1_t . K2 for its 64 gate columns and hands every lane's D fragment (rows 4 g4 + r of column n) to THE SAME lane of
// its layer-2 partner through a mailbox: T slots of 256 lanes x 32 bytes, two 16-byte stores {z0, tag, z1, tag},
// {z2, tag, z3, tag} with tag = base + 1 + t - the data is the flag, as everywhere in the exchange; a ring of T slots needs
// no back-pressure.  Layer 2 (ROLE 4 of wide16_body) adds z_t to its pre-activations at the top of step t and keeps R2 only.
// ---------------------------------------------------------------------------------------------------------------
template <int WH>
__device__ __forceinline__ void wide16_product_body(const LstmParams& p, const int bx, float* smem) {
    constexpr int WG = WH / 16, NJX = WH / 16, WLD = WH + 8, H4 = 4 * WH, NRG = WH / 32;
    constexpr size_t SLOT = (size_t)VBT * WH;
    constexpr unsigned PARITY = VBT * WH * 8u;
    float* sX = smem;                       // [2][16][WLD]: layer 1's h tiles of two consecutive steps
    int* sFlag = (int*)(sX + 2 * VBT * WLD);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g4 = lane >> 4;
    const int group = bx / WG, slice = bx - group * WG;
    const int steps = p.T;
    const int unit = 16 * slice + 4 * wave + (n & 3);
    const int col = (n >> 2) * WH + unit;
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, -1, 0);   // header words + arrival count; no handshake of its own
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;
    float wk[NJX][4];
    static_assert(2 * VBT * WLD >= W16_ST_LDS_WORDS, "the two h1 tiles hold the two staging buffers");
    w16_stage_weight_sets<WH, false>(wk, p.K, WH, wk, p.K, WH, slice, (unsigned*)smem);
    const __amdgpu_buffer_rsrc_t ringrs = __builtin_amdgcn_make_buffer_rsrc(   // the MIRROR of layer 1's ring (sc1 stores: visible on every XCD)
        p.xch + w16_mirror_base(p.num_groups, p.T, WH, 3) + (size_t)group * p.T * SLOT, 0, p.T * (int)(SLOT * sizeof(unsigned long long)), 0x00020000);
    const __amdgpu_buffer_rsrc_t mbrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + w16_mail_base(p.num_groups, p.T, WH) + ((size_t)group * WG + slice) * p.T * W16_MAIL, 0,
        p.T * (int)(W16_MAIL * sizeof(unsigned long long)), 0x00020000);
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    bool aborted = sFlag[0] != 0;

    // thread (row tid / 16, c = tid % 16) gathers the unit pairs c + 16 i of layer 1's h tile of step s (slot s, tags
    // base + 1 + s), one 16-byte load per pair (as ROLE 2 of wide16_body)
    vu32x4 xg[NRG];
    const unsigned vo = (unsigned)(((tid >> 4) * WH + 2 * (tid & 15)) * 8);
    auto ring_issue = [&](int s_) {
#pragma unroll
        for (int i = 0; i < NRG; ++i) xg[i] = __builtin_amdgcn_raw_buffer_load_b128(ringrs, vo + i * 32 * 8, (unsigned)s_ * PARITY, 16);
    };
    auto ring_store = [&](int s_, float* dst) {
        const unsigned want = ticket.base + 1u + (unsigned)s_;
        unsigned bad = 0;
#pragma unroll
        for (int i = 0; i < NRG; ++i) {
            if (xg[i].y == want && xg[i].w == want) {
                dst[32 * i] = __uint_as_float(xg[i].x);
                dst[32 * i + 1] = __uint_as_float(xg[i].z);
            } else {
                bad |= (1u << i);
            }
        }
        unsigned spins = 0;
#ifdef FOV_DBG_W16_NOPRODGATHER   // timing experiment: WRONG results
        bad = 0;
#endif
        while (__any(bad != 0)) {
            ++spins;
            if (spins > VSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) {
                    xch_give_up(p.status);
                    sFlag[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
            // ONE 16-byte probe per lane and sweep while layer 1 is not that far (its last piece: the pieces of a step appear
            // within a fraction of a microsecond of each other); the 16 loads of the full sweep only behind a current probe.
            // Sweeping all of them polled 64 KB per workgroup and iteration through the fabric the layers exchange over:
            // 0.7 us per step of the whole launch (tools/w16_variants.sh, NOPRODGATHER).
            {
                const vu32x4 pv = __builtin_amdgcn_raw_buffer_load_b128(ringrs, vo + (NRG - 1) * 32 * 8, (unsigned)s_ * PARITY, 16);
                if (!__any(pv.y == want && pv.w == want)) continue;
            }
#pragma unroll
            for (int i0 = 0; i0 < NRG; i0 += 8) {
                vu32x4 tv[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) tv[i] = __builtin_amdgcn_raw_buffer_load_b128(ringrs, vo + (i0 + i) * 32 * 8, (unsigned)s_ * PARITY, 16);
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    if (((bad >> (i0 + i)) & 1u) && tv[i].y == want && tv[i].w == want) {
                        dst[32 * (i0 + i)] = __uint_as_float(tv[i].x);
                        dst[32 * (i0 + i) + 1] = __uint_as_float(tv[i].z);
                        bad &= ~(1u << (i0 + i));
                    }
            }
        }
    };
    float* xl = sX + (tid >> 4) * WLD + 2 * (tid & 15);
    if (!aborted && steps > 0) ring_issue(0);
    for (int t = 0; t < steps && !aborted; ++t) {
        ring_store(t, xl + (t & 1) * VBT * WLD);   // waits here while layer 1 is not that far yet
        __syncthreads();   // the tile of step t is in LDS; every wave is done with the MFMAs of step t - 1 (the other buffer)
        if (sFlag[0]) { aborted = true; break; }
        if (t + 1 < steps) ring_issue(t + 1);
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#ifndef FOV_DBG_W16_NOPRODMM   // timing experiment (tools/w16_variants.sh): WRONG results without
        vm_begin(acc);
        wide16_mm<NJX>(acc, sX + (t & 1) * VBT * WLD + n * WLD + 4 * g4, wk);
        vm_end(acc);
#endif
        const f32x4 z = acc[0] + acc[1];
        const unsigned tag = ticket.base + 1u + (unsigned)t;
        const unsigned so = (unsigned)t * (unsigned)(W16_MAIL * 8u);
        __builtin_amdgcn_raw_buffer_store_b128((vu32x4){__float_as_uint(z[0]), tag, __float_as_uint(z[1]), tag}, mbrs, (unsigned)(tid * 32), so, 16 /* sc1 */);
        __builtin_amdgcn_raw_buffer_store_b128((vu32x4){__float_as_uint(z[2]), tag, __float_as_uint(z[3]), tag}, mbrs, (unsigned)(tid * 32 + 16), so, 16);
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

// Three roles, one launch (at most two tiles): layer 1, the products x_t . K2, layer 2.  The grid is 8 x 32 blocks, block b is
// member b / 8 of "XCD group" b % 8 (round-robin dispatch: one XCD, verified by the layers' hello handshakes): XCD groups
// [0, tiles) are layer 1's tiles, [tiles, 2 tiles) the product role's, [2 tiles, 3 tiles) layer 2's - every 32-workgroup group
// has an XCD (32 CUs) of its own and the layers exchange through its L2 (sc0 stores, xch_common.h), 4.2 us per step against
// 5.7 with a group's members dealt over all XCDs (tools/w16_variants.sh); what crosses XCDs - layer 1's h_t to the product
// role (the mirror ring), z_t to layer 2 (the mailboxes) - is written with sc1 stores.  Spare blocks count as arrived and leave.
template <int ACT, int WH>
__global__ __launch_bounds__(256, 1) void lstm_wide16_trio_kernel(Wide16Pair pp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WG = WH / 16;
    const int tiles = pp.l1.num_groups;
    const int xg = (int)blockIdx.x & 7, member = (int)blockIdx.x >> 3;
    const int role = xg / tiles, group = xg - role * tiles;
    if (role > 2) {
        __shared__ unsigned sSpare[4];
        xch_arrive(pp.l1.status, sSpare, -1, 0);
        return;
    }
    const int bx = group * WG + member;
    if (role == 0) wide16_body<ACT, 6, WH, 1>(pp.l1, bx, smem);
    else if (role == 1) wide16_product_body<WH>(pp.l2, bx, smem);
    else wide16_body<ACT, 0, WH, 4>(pp.l2, bx, smem);
}

// Two roles (three or four tiles): layer 1 and layer 2 with K2 in its own registers, placed like the three-role launch - an XCD
// per 32-workgroup group (XCD groups [0, tiles) layer 1, [tiles, 2 tiles) layer 2), layer 2 reads layer 1's h_t from the mirror ring
template <int ACT, int WH>
__global__ __launch_bounds__(256, 1) void lstm_wide16_pair_kernel(Wide16Pair pp) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int WG = WH / 16;
    const int tiles = pp.l1.num_groups;
    const int xg = (int)blockIdx.x & 7, member = (int)blockIdx.x >> 3;
    const int role = xg / tiles, group = xg - role * tiles;
    if (role > 1) {
        __shared__ unsigned sSpare[4];
        xch_arrive(pp.l1.status, sSpare, -1, 0);
        return;
    }
    const int bx = group * WG + member;
    if (role == 0) wide16_body<ACT, 6, WH, 1>(pp.l1, bx, smem);
    else wide16_body<ACT, WH / 16, WH, 2>(pp.l2, bx, smem);
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: encoder + free-running decoder of the Keras seq2seq (mycode/FoV_seq2seq.py:19-28 encoder, :112-117 decoder,
// :154-178 the decode loop: y_t = Dense(6, tanh)(h_t) fed back as x_{t+1}) for SMALL batches at widths 128 / 256, ONE launch,
// in the sixteen-units-per-workgroup form: a 16-sequence tile is spread over WH / 16 workgroups (8 at H = 128) where
// lstm_cluster.hip's fused kernel spreads it over WH / 64 (2) - at the reference's own batch of 32 that kernel has four
// workgroups at work and spends 1.7 us of every 3.8 us step on one workgroup's MFMAs (r03: 0.076 ms per call).
// A wave owns four units x four gates (one 16-column MFMA tile); both phases' weight slices stay in registers (R: WH / 16
// k-blocks each, encoder K: six k-blocks, decoder K: one).  Every workgroup gathers the whole h_t tile for its recurrent
// product anyway, so the decoder's Dense needs NO further exchange: each workgroup forms y_t = tanh(h_t . W + b) of its tile
// redundantly from the gathered tile (thread = (row, output, half of the units), 16-byte LDS reads, one shuffle), writes it
// as x_{t+1} into LDS and - workgroup 0 of the group - into `out`.  Same granule exchange as the layer kernel above.
// ---------------------------------------------------------------------------------------------------------------
template <int ACT, int WH>
__global__ __launch_bounds__(256, 1) void wide16_s2s_kernel(LstmParams p) {
    constexpr int WG = WH / 16, NJR = WH / 16, NJX = 6, WLD = WH + 8, WNG = WG / 2, H4 = 4 * WH;
    constexpr int WDL = WH + 4;          // floats per row of the transposed Dense kernel [output][unit]
    constexpr unsigned OORB = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sH = smem;                     // [16][WLD]
    float* sX = sH + VBT * WLD;           // [2][16][WLD]: encoder x tiles (columns < F), decoder x tile (columns < F_dec) in [0]
    float* sT = sX + 2 * VBT * WLD;       // [4 waves][16][17] gate transpose
    float* sWd = sT + 4 * 16 * 17;        // [8][WDL] Dense kernel transposed, rows >= F_dec zero
    int* sFlag = (int*)(sWd + 8 * WDL);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    const int bx = (int)blockIdx.x;
    if (p.xcd_pad) {   // fewer than eight groups: padded grid, members of a group 8 blocks apart (one XCD)
        group = bx & 7;
        slice = bx >> 3;
        if (group >= p.num_groups) {
            __shared__ unsigned sSpare[4];
            xch_arrive(p.status, sSpare, -1, 0);
            return;
        }
    } else if ((p.num_groups & 7) == 0) {
        group = (bx / (8 * WG)) * 8 + (bx & 7);
        slice = (bx >> 3) & (WG - 1);
    } else {
        group = bx / WG;
        slice = bx - group * WG;
    }
    const int F = p.F, FD = p.F_dec, T_in = p.T, T_out = p.T_out;
    const int unit = 16 * slice + 4 * wave + (n & 3);
    const int col = (n >> 2) * WH + unit;
    const int row_o = 4 * g4 + (n >> 2);
    float* tw = sT + wave * (16 * 17);
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- resident weights of both phases through LDS (w16_stage_weight_sets; the staging buffers are the h / x tiles, filled after
    // it; rows past an input's width read as zero: the descriptor ends there) ----
    float wk[NJX][4], wr[NJR][4], dk[1][4], dr[NJR][4];
    static_assert(3 * VBT * WLD >= W16_ST_LDS_WORDS, "the h tile and the two x tiles hold the two staging buffers");
    w16_stage_weight_sets<WH, true>(wk, p.K, F, wr, p.R, WH, slice, (unsigned*)smem);
    w16_stage_weight_sets<WH, true>(dk, p.dK, FD, dr, p.dR, WH, slice, (unsigned*)smem);
    const float bv_e = p.b[col], bv_d = p.db[col];
    for (int i = tid; i < 2 * VBT * WLD; i += 256) sX[i] = 0.f;   // columns past the inputs' widths stay zero
    for (int i = tid; i < 8 * WDL; i += 256) {
        const int o = i / WDL, k = i - o * WDL;
        sWd[i] = (o < FD && k < WH) ? p.dW[(size_t)k * FD + o] : 0.f;
    }
    // Dense: thread (row drow, output dout < 8, half dkh of the units)
    const int drow = tid >> 4, dout = tid & 7, dkh = (tid >> 3) & 1;
    const float dbv = dout < FD ? p.dbias[dout] : 0.f;

    // ---- exchange bookkeeping (two parity slots per group) ----
    constexpr size_t SLOT = (size_t)VBT * WH;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(p.xch + (size_t)group * 2 * SLOT, 0,
                                                                         2 * (int)(SLOT * sizeof(unsigned long long)), 0x00020000);
    const unsigned pub_off = (unsigned)(row_o * WH + unit) * 8u;
    const int grow = tid >> 4, ghalf = (tid >> 3) & 1, gp = tid & 7;
    const unsigned gvoff = (unsigned)(grow * WH + 2 * gp) * 8u;
    const int lbase = grow * WLD + 2 * gp;
    constexpr unsigned PARITY = VBT * WH * 8u;
    xch_hello_poll(p.status, sXch, group, WG, &sFlag[0]);
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);

    auto gslice = [&](int j) { return (slice + 1 + ghalf * WNG + j) & (WG - 1); };
    // the partners' pieces of the h tile (tags `epoch`) -> sH; all loads of a sweep in flight together
    auto gather = [&](unsigned base) {
        vu32x4 v[WNG];
        unsigned bad = 0;
#pragma unroll
        for (int j = 0; j < WNG; ++j) {
            const bool on = !(ghalf == 1 && j == WNG - 1);   // the last load of half 1 would be the own slice
            v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, on ? gvoff + (unsigned)(gslice(j) * 16) * 8u : OORB, base, 16);
        }
#pragma unroll
        for (int j = 0; j < WNG; ++j) {
            const bool on = !(ghalf == 1 && j == WNG - 1);
            const int lo = lbase + gslice(j) * 16;
            if (!on) continue;
            if (v[j].y == epoch && v[j].w == epoch) {
                sH[lo] = __uint_as_float(v[j].x);
                sH[lo + 1] = __uint_as_float(v[j].z);
            } else {
                bad |= (1u << j);
            }
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > VSPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) { xch_give_up(p.status); sFlag[0] = 1; }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            vu32x4 tv[WNG];
#pragma unroll
            for (int j = 0; j < WNG; ++j) {
                const bool on = !(ghalf == 1 && j == WNG - 1);
                tv[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, on ? gvoff + (unsigned)(gslice(j) * 16) * 8u : OORB, base, 16);
            }
#pragma unroll
            for (int j = 0; j < WNG; ++j) {
                const int lo = lbase + gslice(j) * 16;
                if (((bad >> j) & 1u) && tv[j].y == epoch && tv[j].w == epoch) {
                    sH[lo] = __uint_as_float(tv[j].x);
                    sH[lo + 1] = __uint_as_float(tv[j].z);
                    bad &= ~(1u << j);
                }
            }
        }
    };

    const float* hrow = sH + n * WLD + 4 * g4;
    const int xrw = tid >> 4, xc = tid & 15;
    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * VBT;
        __syncthreads();   // previous tile fully consumed
        const int live_rows = p.B - b0 < VBT ? p.B - b0 : VBT;
        const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(T_in > 0 ? p.x + (size_t)b0 * T_in * F : nullptr), 0, T_in > 0 ? live_rows * T_in * F * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t d0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.dec_in0 + (size_t)b0 * FD), 0, live_rows * FD * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t h0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h0 ? p.h0 + (size_t)b0 * WH : nullptr), 0, p.h0 ? live_rows * WH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t c0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.c0 ? p.c0 + (size_t)b0 * WH : nullptr), 0, p.c0 ? live_rows * WH * 4 : 0, 0x00020000);
        {   // initial state (zero unless the caller passes one: fov_seq2seq_decoder_fwd) into the h tile
            float hv[VBT * WH / 256];
#pragma unroll
            for (int q = 0; q < VBT * WH / 256; ++q) hv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, (unsigned)((tid + 256 * q) * 4), 0, 0));
#pragma unroll
            for (int q = 0; q < VBT * WH / 256; ++q) {
                const int e = tid + 256 * q;
                sH[(e / WH) * WLD + (e % WH)] = hv[q];
            }
        }
        const unsigned soff = (unsigned)((row_o * WH + unit) * 4);
        float c = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, soff, 0, 0));
        float hc = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, soff, 0, 0));
        unsigned xoff[NJX];
#pragma unroll
        for (int i = 0; i < NJX; ++i) xoff[i] = (xc + 16 * i < F) ? (unsigned)((xrw * T_in * F + xc + 16 * i) * 4) : OORB;
        auto load_x1 = [&](int i, int t) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, xoff[i], (unsigned)(t * F * 4), 0)); };
        float* xl = sX + xrw * WLD + xc;
        const int total = T_in + T_out;           // recurrent steps; step s < T_in: encoder, else decoder step s - T_in
        {   // x of encoder steps 0 and 1; a launch without an encoder: the decoder's first input
            float x1[2][NJX];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                for (int i = 0; i < NJX; ++i) x1[tt][i] = load_x1(i, tt < T_in ? tt : 0);
            const float d0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d0rs, xc < FD ? (unsigned)((xrw * FD + xc) * 4) : OORB, 0, 0));
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (tt < T_in) {
#pragma unroll
                    for (int i = 0; i < NJX; ++i)
                        if (xc + 16 * i < F) xl[tt * VBT * WLD + 16 * i] = x1[tt][i];
                }
            if (T_in == 0 && xc < 16) xl[0] = d0;     // (columns FD .. 15 of the decoder tile: zero, d0 read out of range)
        }
        __syncthreads();
        // ---- pre-activations of step 0 ----
        f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (total > 0) {
            const float bv0 = T_in > 0 ? bv_e : bv_d;
            acc[0] = (f32x4){bv0, bv0, bv0, bv0};
            vm_begin(acc);
            if (T_in > 0) {
                wide16_mm<NJX>(acc, sX + n * WLD + 4 * g4, wk);
                wide16_mm<NJR>(acc, hrow, wr);
            } else {
                wide16_mm<1>(acc, sX + n * WLD + 4 * g4, dk);
                wide16_mm<NJR>(acc, hrow, dr);
            }
            vm_end(acc);
        }
        float xs[NJX] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < total; ++s) {
            const bool next_enc = s + 1 < T_in;                 // the NEXT step is an encoder step: its x comes from memory
            const bool next_dec = !next_enc && s + 1 < total;   // ... a decoder step: its x is dec_in0 (s + 1 == T_in) or y of this step
            // encoder x pipeline: x_{s+1} (requested during step s-1) registers -> LDS; then request x_{s+2}
            if (s > 0 && next_enc) {
                float* xb = xl + ((s + 1) & 1) * VBT * WLD;
#pragma unroll
                for (int i = 0; i < NJX; ++i)
                    if (xc + 16 * i < F) xb[16 * i] = xs[i];
            }
            if (s + 2 < T_in) {
#pragma unroll
                for (int i = 0; i < NJX; ++i) xs[i] = load_x1(i, s + 2);
            }
            float d0 = 0.f;
            if (s + 1 == T_in) d0 = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(d0rs, xc < FD ? (unsigned)((xrw * FD + xc) * 4) : OORB, 0, 0));
            // ---- the four gates of a cell meet: 16 x 16 transpose through the wave's scratch ----
#pragma unroll
            for (int r = 0; r < 4; ++r) tw[(4 * g4 + r) * 17 + n] = acc[0][r] + acc[1][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const float zi = tw[row_o * 17 + (n & 3)], zf = tw[row_o * 17 + 4 + (n & 3)], zg = tw[row_o * 17 + 8 + (n & 3)],
                        zo = tw[row_o * 17 + 12 + (n & 3)];
            {
                const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
                c = fmaf(fg, c, ig * gg);
                hc = og * tanh_f(c);
            }
            // every step publishes: the decoder's Dense needs the whole h tile of every decoder step, the encoder's last step
            // hands its tile to the decoder; only an encoder-only tail (T_out == 0, last step) has no reader
            const bool need_tile = s + 1 < total || s >= T_in;
            unsigned par = 0;
            if (need_tile) {
                ++epoch;
                par = (epoch & 1u) * PARITY;
                XCH_STORE_B64(ticket.same_xcd, ((vu32x2){__float_as_uint(hc), epoch}), xrs, pub_off, par);
            }
            __syncthreads();   // barrier 1: every wave is done reading sH and the x tile of this step
            if (need_tile) sH[row_o * WLD + unit] = hc;
            if (s + 1 == T_in && xc < 16) xl[0] = d0;   // the decoder's first input into tile 0 (columns FD .. 15: zero)
            const float bvn = next_enc ? bv_e : bv_d;
            acc[0] = (f32x4){bvn, bvn, bvn, bvn};
            acc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
            if (next_enc) {   // x_{s+1} . K needs no remote data: it runs under the exchange
                vm_begin(acc);
                wide16_mm<NJX>(acc, sX + ((s + 1) & 1) * VBT * WLD + n * WLD + 4 * g4, wk);
                vm_end(acc);
            }
            if (need_tile) gather(par);
            __syncthreads();   // barrier 2: the whole h tile of this step is in LDS
            if (sFlag[0]) { aborted = true; break; }
            if (s >= T_in) {
                // ---- y = tanh(h . W + b) of this decoder step: every workgroup, its whole tile ----
                const float* hr = sH + drow * WLD + dkh * (WH / 2);
                const float* wrow = sWd + dout * WDL + dkh * (WH / 2);
                float a = 0.f;
#pragma unroll 8
                for (int k = 0; k < WH / 2; k += 4) {
                    const f32x4 hv4 = *(const f32x4*)(hr + k), wv4 = *(const f32x4*)(wrow + k);
                    a = fmaf(hv4[0], wv4[0], a); a = fmaf(hv4[1], wv4[1], a); a = fmaf(hv4[2], wv4[2], a); a = fmaf(hv4[3], wv4[3], a);
                }
                a += __shfl_xor(a, 8);
                const float y = tanh_f(a + dbv);
                if (dkh == 0 && dout < FD) {
                    if (next_dec) sX[drow * WLD + dout] = y;      // x of the next decoder step (tile 0)
                    if (slice == 0 && b0 + drow < p.B) p.out[((size_t)(b0 + drow) * T_out + (s - T_in)) * FD + dout] = y;
                }
                if (next_dec) __syncthreads();   // barrier 3: the next input is in LDS
            }
            if (next_enc) {
                vm_begin(acc);
                wide16_mm<NJR>(acc, hrow, wr);
                vm_end(acc);
            } else if (next_dec) {
                vm_begin(acc);
                wide16_mm<1>(acc, sX + n * WLD + 4 * g4, dk);
                wide16_mm<NJR>(acc, hrow, dr);
                vm_end(acc);
            }
        }
        if (!aborted) {
            const int row = b0 + row_o;
            if (row < p.B) {
                if (p.hT) p.hT[(size_t)row * WH + unit] = hc;
                if (p.cT) p.cT[(size_t)row * WH + unit] = c;
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

template <int NJX, int WH>
int launch_wide16_t(LstmParams& p, hipStream_t stream) {
    constexpr int WG = WH / 16;
    p.num_tiles = (p.B + VBT - 1) / VBT;
    const int max_groups = device_cu_count() / WG;
    if (max_groups < 1) { set_error("width-%d LSTM layer (%d workgroups per tile) needs at least %d CUs", WH, WG, WG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * 2 * VBT * WH * sizeof(unsigned long long) > kXchBytes - kHelloBytes) { set_error("wide LSTM layer: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.epoch_span = p.T * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    const size_t lds = sizeof(float) * (3 * VBT * (WH + 8) + 4 * 16 * 17) + 64;
    void (*kern)(LstmParams) = p.act == FOV_ACT_HARD_SIGMOID ? lstm_wide16_kernel<FOV_ACT_HARD_SIGMOID, NJX, WH>
                                                             : lstm_wide16_kernel<FOV_ACT_SIGMOID, NJX, WH>;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    // fewer than eight groups of at most sixteen workgroups: same-XCD placement through a padded grid (FOV_NO_XCD_PAD=1: off)
    const bool no_pad = env_knobs().no_xcd_pad != 0;
    const int pad_max = env_knobs().xcd_pad_max;   // members per group at most
    p.xcd_pad = (!no_pad && WG <= pad_max && p.num_groups < 8 && device_cu_count() >= 8 * WG) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(p.xcd_pad ? 8 * WG : p.num_groups * WG), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("wide16 LSTM launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace

// Widths 128 / 256 / 512 (H / 16 workgroups per tile), at most eight tiles and one tile per group, and an input the kernel
// keeps in registers: narrow (F <= 96) or as wide as the layer (F = H, 16-byte aligned rows).  The latency regime: with
// so few tiles the other kernels leave most CUs idle and a step is the exchange latency plus ALL of a workgroup's MFMAs.
bool wide16_shape(int B, int F, int H) {
    if (!(H == 128 || H == 256 || H == 512) || B <= 0) return false;
    const int tiles = (B + VBT - 1) / VBT;
    const int max_groups = device_cu_count() / (H / 16);
    if (tiles > 8 || tiles > max_groups) return false;
    return (F >= 1 && F <= 96) || F == H;
}
bool wide16_preferred(const float* x, int B, int F, int H) {
    return wide16_shape(B, F, H) && (F <= 96 || (((uintptr_t)x) & 15) == 0);
}

// Two stacked layers (F <= 96 -> 512 -> 512) as one launch: both layers' groups must be resident (2 x tiles x 32 workgroups),
// T >= 2, the ring (tiles x T slots) and layer 2's parity slots must fit the granule area.
bool wide16_pair_shape(int B, int T, int F, int H) {
    const bool off = env_knobs().no_stack2 != 0;
    if (off || H != 512 || B <= 0 || T < 2 || F < 1 || F > 96) return false;
    const int tiles = (B + VBT - 1) / VBT;
    if (tiles > 4 || 8 * (H / 16) > device_cu_count()) return false;   // an XCD per group: at most 8 groups of 32 workgroups
    // the rings, layer 2's parity slots and the mirror rings must fit the granule area (the three-role form of at most two tiles
    // also needs its mailboxes: launch_wide16_pair falls back to two roles where they do not fit)
    return (w16_mirror_base(tiles, T, H, 2) + (size_t)tiles * T * VBT * H) * sizeof(unsigned long long) <= kXchBytes - kHelloBytes;
}

int launch_wide16_pair(const LstmParams& a, const LstmParams& b, hipStream_t stream) {
    constexpr int WH = 512, WG = WH / 16;
    if (a.B == 0) return FOV_OK;
    if (!wide16_pair_shape(a.B, a.T, a.F, a.H)) { set_error("two-layer width-512 launch: unsupported shape"); return FOV_ERR_UNSUPPORTED; }
    Wide16Pair pp = {a, b};
    for (LstmParams* q : {&pp.l1, &pp.l2}) {
        q->num_tiles = (a.B + VBT - 1) / VBT;
        q->num_groups = q->num_tiles;               // one tile per group
        q->epoch_span = 2 * a.T + 1;                // layer 1's tags base + 1 .. base + T, layer 2's base + T + 1 .. base + 2T
        q->xcd_pad = 0;
    }
    if (int rc_ = xch_account(pp.l1.status, pp.l1.epoch_span, stream)) return rc_;
    const size_t lds = sizeof(float) * (3 * VBT * (WH + 8) + 4 * 16 * 17) + 64;
    // three roles (layer 2's input projection on workgroups of its own) while they all fit the chip and the mailboxes the
    // granule area: lstm.py's batch of 32 (two tiles)
    const int tiles = pp.l1.num_tiles;
    const bool trio = !env_knobs().no_wide16_trio && tiles <= 2 &&
                      (w16_mirror_base(tiles, a.T, WH, 3) + (size_t)tiles * a.T * VBT * WH) * sizeof(unsigned long long) <= kXchBytes - kHelloBytes;
    pp.l1.trio = pp.l2.trio = trio ? 3 : 2;   // both launches are XCD-placed: layer 1 mirrors its ring for the readers on other XCDs
    void (*kern)(Wide16Pair) = trio ? (a.act == FOV_ACT_HARD_SIGMOID ? lstm_wide16_trio_kernel<FOV_ACT_HARD_SIGMOID, WH>
                                                                     : lstm_wide16_trio_kernel<FOV_ACT_SIGMOID, WH>)
                                    : (a.act == FOV_ACT_HARD_SIGMOID ? lstm_wide16_pair_kernel<FOV_ACT_HARD_SIGMOID, WH>
                                                                     : lstm_wide16_pair_kernel<FOV_ACT_SIGMOID, WH>);
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(8 * WG), dim3(256), lds, stream, pp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("two-layer width-512 launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// Encoder + free-running decoder in one launch: widths 128 / 256, narrow inputs, at most eight outputs fed back, one tile per group
// and at most eight tiles (the latency regime the kernel is built for; larger batches stay on lstm_cluster.hip's fused kernel).
bool wide16_s2s_shape(int B, int F_enc, int F_dec, int H) {
    if (env_knobs().no_wide16 || !(H == 128 || H == 256) || B <= 0) return false;
    const int tiles = (B + VBT - 1) / VBT;
    if (tiles > 8 || tiles > device_cu_count() / (H / 16)) return false;
    return F_enc >= 1 && F_enc <= 96 && F_dec >= 1 && F_dec <= 8;
}

template <int WH>
static int launch_wide16_s2s_t(LstmParams& p, hipStream_t stream) {
    constexpr int WG = WH / 16;
    p.num_tiles = (p.B + VBT - 1) / VBT;
    p.num_groups = p.num_tiles;
    if ((size_t)p.num_groups * 2 * VBT * WH * sizeof(unsigned long long) > kXchBytes - kHelloBytes) { set_error("seq2seq decode: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.epoch_span = p.T + p.T_out + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    const size_t lds = sizeof(float) * (3 * VBT * (WH + 8) + 4 * 16 * 17 + 8 * (WH + 4)) + 64;
    void (*kern)(LstmParams) = p.act == FOV_ACT_HARD_SIGMOID ? wide16_s2s_kernel<FOV_ACT_HARD_SIGMOID, WH> : wide16_s2s_kernel<FOV_ACT_SIGMOID, WH>;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    const bool no_pad = env_knobs().no_xcd_pad != 0;
    p.xcd_pad = (!no_pad && WG <= env_knobs().xcd_pad_max && p.num_groups < 8 && device_cu_count() >= 8 * WG) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3(p.xcd_pad ? 8 * WG : p.num_groups * WG), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("seq2seq decode (wide16) launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

int launch_wide16_s2s(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0 || p.T + p.T_out == 0) return FOV_OK;
    if (!wide16_s2s_shape(p.B, p.F, p.F_dec, p.H)) { set_error("seq2seq decode (wide16): unsupported shape"); return FOV_ERR_UNSUPPORTED; }
    return p.H == 128 ? launch_wide16_s2s_t<128>(p, stream) : launch_wide16_s2s_t<256>(p, stream);
}

int launch_wide16(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0) return FOV_OK;
    const bool narrow = p.F <= 96;
    switch (p.H) {
        case 128: return narrow ? launch_wide16_t<6, 128>(p, stream) : launch_wide16_t<8, 128>(p, stream);
        case 256: return narrow ? launch_wide16_t<6, 256>(p, stream) : launch_wide16_t<16, 256>(p, stream);
        case 512: return narrow ? launch_wide16_t<6, 512>(p, stream) : launch_wide16_t<32, 512>(p, stream);
    }
    set_error("wide16 LSTM layer: unsupported width %d", p.H);
    return FOV_ERR_UNSUPPORTED;
}

}  // namespace fov
