// Shared pieces of the bf16 kernels (BASELINE.json configs[4]: the config-3 model with bf16 operands into
// v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 gates / cell state / master weights).
//
// Fragment conventions (cdna_hip_programming.md section 3), lane l = (n = l & 15, g4 = l >> 4):
//   A (16 x 32): lane holds A[row n][k = 8*g4 + j], j = 0..7   - 16 bytes of a bf16 row image in LDS
//   B (32 x 16): lane holds B[k = 8*g4 + j][col n]             - packed once from the fp32 weights, register-resident
//   D (16 x 16): lane holds D[row 4*g4 + r][col n], r = 0..3   - same as the fp32 MFMA the fp32 kernels use
// Ownership is the fp32 eight-workgroup kernels' (lstm_wide.hip, mix_decoder.hip): a 16-sequence tile per group of 8
// workgroups, workgroup `slice` owns hidden units [32*slice, +32), a wave 8 of them as two N-tiles [i | f], [g | o];
// after a DPP half swap every lane holds all four gates of ONE unit for TWO sequences (rows row0, row0 + 1).
// Those two h values travel as ONE granule {bf16 pair, epoch}: half the granules of the fp32 exchange.
#pragma once
#include "fov_common.h"
#include "xch_common.h"

namespace fov {

constexpr int QH = 256;     // hidden units
constexpr int QG = 8;       // workgroups per tile
constexpr int QBT = 16;     // sequences per tile
constexpr int QLD = 272;    // bf16 elements per LDS row of an activation tile: 544 B, conflict-free ds_read_b128 of A fragments
constexpr int QNG = 4;      // 16-byte loads (two adjacent units' tagged granules) per thread and exchange: 7 slices * 8 row pairs * 16 unit pairs / 256, rounded up
constexpr unsigned Q_SPIN = 1u << 20;

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned qu32x4 __attribute__((ext_vector_type(4)));
typedef unsigned qu32x2 __attribute__((ext_vector_type(2)));

// round-to-nearest-even (v_cvt_pk_bf16_f32); lo in bits 0..15
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ unsigned short bf16_bits(float v) { return (unsigned short)(pack_bf16(v, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf16_to_f32(unsigned short b) { return __uint_as_float((unsigned)b << 16); }

__device__ __forceinline__ void qmfma(f32x4& acc, qu32x4 a, qu32x4 b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
}

// B fragment of k-block kb for gate column `col` from a row-major fp32 matrix W (rows x ld); rows >= nrows are zero
__device__ __forceinline__ qu32x4 load_bfrag(const float* __restrict__ W, int ld, int nrows, int kb, int g4, int col) {
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int k = 32 * kb + 8 * g4 + j;
        v[j] = (k < nrows) ? W[(size_t)k * ld + col] : 0.f;
    }
    return (qu32x4){pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]), pack_bf16(v[6], v[7])};
}
// The [NKB][2] fragment set of one (nrows x ld) fp32 kernel for this lane's two gate columns.
// Loads go through a buffer descriptor whose num_records ends at the last real row: rows beyond it read as zero in
// hardware, so every load is UNCONDITIONAL and hipcc keeps a whole batch in flight.  (Written as `k < nrows ? W[..] :
// 0` each load sits in its own exec-masked branch with an s_waitcnt vmcnt(0) behind it: 2 500 cycles per fragment,
// 81 000 cycles = 38 us for the 32 fragments of a layer kernel - tools/stamp_bf16_layer.py.)
// `behind_first_batch()` runs once, between the requests of the first batch and its packing: the place of
// xch_arrive_commit (xch_common.h) in a kernel's prologue.
struct QNothing {
    __device__ __forceinline__ void operator()() const {}
};
template <int NKB, int CHMAX = 4, class Behind = QNothing>
__device__ __forceinline__ void load_weight_set(qu32x4 (&w)[NKB][2], const float* __restrict__ W, int ld, int nrows, int g4, int col0,
                                                int col1, Behind behind_first_batch = Behind()) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * ld * 4, 0x00020000);
    constexpr int CH = NKB < CHMAX ? NKB : CHMAX;   // k-blocks per batch: 16*CH loads in flight (CHMAX 2: kernels held to 256 registers)
    static_assert(NKB % CH == 0, "NKB must be a multiple of the batch");
#pragma unroll
    for (int c = 0; c < NKB / CH; ++c) {
        float v[CH][2][8];
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const unsigned row = (unsigned)(32 * (c * CH + i) + 8 * g4 + j) * (unsigned)ld;
                v[i][0][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (row + (unsigned)col0) * 4u, 0, 0));
                v[i][1][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (row + (unsigned)col1) * 4u, 0, 0));
            }
        if (c == 0) behind_first_batch();
#pragma unroll
        for (int i = 0; i < CH; ++i)
#pragma unroll
            for (int t = 0; t < 2; ++t)
                w[c * CH + i][t] = (qu32x4){pack_bf16(v[i][t][0], v[i][t][1]), pack_bf16(v[i][t][2], v[i][t][3]),
                                            pack_bf16(v[i][t][4], v[i][t][5]), pack_bf16(v[i][t][6], v[i][t][7])};
        // a kernel held to 256 registers: keep the scheduler from hoisting the NEXT batches' loads above this batch's packing
        // (it clustered all 256 loads of a layer and spilled 34 of them, each behind its own vmcnt(0))
        if constexpr (CHMAX < 4) __builtin_amdgcn_sched_barrier(0);
    }
}

// ---- the same fragment sets through LDS (round 4) ------------------------------------------------------------------
// load_weight_set asks for a fragment the way the MFMA wants it - lane (n, g4) reads 8 rows of ONE column - so a wave's
// load instruction touches 8 cache lines for 256 useful bytes, every line of the workgroup's slice is requested by all
// four waves, and the L1's line rate bounds the prologue: 12 300 cycles for the two sets of a layer kernel, 22 000 for
// the upper layer of the two-layer kernel that shares its CU with a second workgroup (tools/stamp_bf16_layer.py).
// Here the WORKGROUP reads its slice as whole lines - a stage is 64 rows x (4 gates x 32 units), 8 dwordx4 loads per
// thread, 8 lanes to a 128-byte line -, packs row pairs to bf16, transposes through LDS (column-major, 36 words per
// column, k-words XOR-swizzled so that the 64 lanes of a ds_write_b32 spread over all banks) and every lane picks its
// fragments up as ds_read_b128.  Two register sets and two LDS buffers: the loads of stage s + 1 are in flight while
// stage s is packed, written and read; one barrier per stage.  The matrices need 4-byte alignment only (a parameter view
// inside a trainer's flat buffer): a dwordx4 load from such an address returns the right four dwords on gfx950
// (tools/microbench/unaligned_b128.hip; tests: ..._at_any_four_byte_offset_...).
constexpr int QST_WORDS = 36;                 // words per column: 32 (64 rows as bf16 pairs) + 4 (16-byte aligned, 4 mod 32)
constexpr int QST_BUF = 128 * QST_WORDS;      // one buffer, in 32-bit words: 18 432 bytes
constexpr int QST_DEPTH = 2;                  // stages requested ahead of the one being packed (DEPTH + 1 register sets of 32)
constexpr int QST_LDS_WORDS = 2 * QST_BUF;    // __shared__ __attribute__((aligned(16))) unsigned sStage[QST_LDS_WORDS]

struct QStageLane {
    unsigned goff;      // byte offset of (row 2 * kp, this thread's 4 columns) in the fp32 matrix
    unsigned wr_even;   // LDS word of the thread's first column for even / odd 16-row groups of a stage (swizzle folded in)
    unsigned wr_odd;
    unsigned rd[2];     // LDS word of fragment (k-block 0 of the stage, N-tile t)
};
__device__ __forceinline__ QStageLane q_stage_lane(int ld, int slice) {
    const int tid = threadIdx.x;
    const int c = tid & 7, kp = (tid >> 3) & 7, gate = tid >> 6;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 15, g4 = lane >> 4, hi = n >> 3;
    QStageLane q;
    q.goff = (unsigned)((2 * kp) * ld + gate * QH + 32 * slice + 4 * c) * 4u;
    const int sbw = (c >> 1) & 1;                    // swizzle bit of the columns this thread writes (column bit 3)
    const unsigned wbase = (unsigned)((32 * gate + 4 * c) * QST_WORDS + kp);
    q.wr_even = wbase + 8u * sbw;                    // 16-row group it: word 8 * (it ^ sb) + kp
    q.wr_odd = wbase - 8u * sbw;
    const int sbr = wave & 1;                        // ... of the columns it reads: 8 * wave + (n & 7)
#pragma unroll
    for (int t = 0; t < 2; ++t)
        q.rd[t] = (unsigned)(((2 * t + hi) * 32 + 8 * wave + (n & 7)) * QST_WORDS + 4 * (g4 ^ (2 * sbr)));
    return q;
}
// requests of stage `ls` (rows [64 ls, 64 ls + 64)) of W; rows >= nrows read as zero (whole offset in the vector register:
// the hardware's range check does not see a scalar offset)
__device__ __forceinline__ void q_stage_issue(f32x4 (&r)[8], const float* __restrict__ W, int ld, int nrows, int ls, const QStageLane& q) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * ld * 4, 0x00020000);
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const qu32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, q.goff + (unsigned)((64 * ls + 16 * it + e) * ld) * 4u, 0, 0);
            r[2 * it + e] = __builtin_bit_cast(f32x4, v);
        }
}
__device__ __forceinline__ void q_stage_write(const f32x4 (&r)[8], unsigned* buf, const QStageLane& q) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        unsigned* wp = buf + ((it & 1) ? q.wr_odd : q.wr_even) + 8 * it;
#pragma unroll
        for (int j = 0; j < 4; ++j) wp[j * QST_WORDS] = pack_bf16(r[2 * it][j], r[2 * it + 1][j]);
    }
}
template <int NKB>
__device__ __forceinline__ void q_stage_read(qu32x4 (&w)[NKB][2], int ls, const unsigned* buf, const QStageLane& q) {
#pragma unroll
    for (int kbl = 0; kbl < 2; ++kbl)
        if (2 * ls + kbl < NKB) {
#pragma unroll
            for (int t = 0; t < 2; ++t) w[2 * ls + kbl][t] = *(const qu32x4*)(buf + q.rd[t] + 16 * kbl);
        }
}
// Two / three fragment sets in one pipeline (every thread of the 256 calls it; the buffers are free again at return, after
// a barrier).  `behind_first_stage()` runs once behind the first stage's requests (xch_arrive_commit).
template <bool HAS_C, int DEPTH, int NA, int NB, int NC, class Behind>
__device__ __forceinline__ void stage_weight_sets_impl(qu32x4 (&a)[NA][2], const float* __restrict__ Wa, int nra, qu32x4 (&b)[NB][2],
                                                       const float* __restrict__ Wb, int nrb, qu32x4 (&c)[NC][2],
                                                       const float* __restrict__ Wc, int nrc, int ld, int slice, unsigned* sStage,
                                                       Behind behind_first_stage) {
    constexpr int SA = (NA + 1) / 2, SB = (NB + 1) / 2, SC = HAS_C ? (NC + 1) / 2 : 0, NS = SA + SB + SC;
    const QStageLane q = q_stage_lane(ld, slice);
    f32x4 r[DEPTH + 1][8];
    auto issue = [&](int s) __attribute__((always_inline)) {
        if (s < SA) q_stage_issue(r[s % (DEPTH + 1)], Wa, ld, nra, s, q);
        else if (s < SA + SB) q_stage_issue(r[s % (DEPTH + 1)], Wb, ld, nrb, s - SA, q);
        else q_stage_issue(r[s % (DEPTH + 1)], Wc, ld, nrc, s - SA - SB, q);
    };
    issue(0);
    behind_first_stage();
#pragma unroll
    for (int s = 1; s < DEPTH; ++s)
        if (s < NS) issue(s);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + DEPTH < NS) issue(s + DEPTH);
        unsigned* buf = sStage + (s & 1) * QST_BUF;
        q_stage_write(r[s % (DEPTH + 1)], buf, q);
        __syncthreads();   // (buffer s & 1 is written again two stages on, behind the barrier of stage s + 1, which a wave passes after these reads)
        if (s < SA) q_stage_read<NA>(a, s, buf, q);
        else if (s < SA + SB) q_stage_read<NB>(b, s - SA, buf, q);
        else q_stage_read<NC>(c, s - SA - SB, buf, q);
    }
    __syncthreads();
}
template <int DEPTH = QST_DEPTH, int NA, int NB, class Behind>
__device__ __forceinline__ void stage_weight_sets(qu32x4 (&a)[NA][2], const float* __restrict__ Wa, int nra, qu32x4 (&b)[NB][2],
                                                  const float* __restrict__ Wb, int nrb, int ld, int slice, unsigned* sStage,
                                                  Behind behind_first_stage) {
    stage_weight_sets_impl<false, DEPTH>(a, Wa, nra, b, Wb, nrb, b, Wb, nrb, ld, slice, sStage, behind_first_stage);
}
template <int DEPTH = QST_DEPTH, int NA, int NB, int NC, class Behind>
__device__ __forceinline__ void stage_weight_sets(qu32x4 (&a)[NA][2], const float* __restrict__ Wa, int nra, qu32x4 (&b)[NB][2],
                                                  const float* __restrict__ Wb, int nrb, qu32x4 (&c)[NC][2], const float* __restrict__ Wc,
                                                  int nrc, int ld, int slice, unsigned* sStage, Behind behind_first_stage) {
    stage_weight_sets_impl<true, DEPTH>(a, Wa, nra, b, Wb, nrb, c, Wc, nrc, ld, slice, sStage, behind_first_stage);
}

// B fragment of a TRANSPOSED product: B[k][col] = W[row = out][k index along a row]; the 8 values are contiguous
__device__ __forceinline__ qu32x4 load_bfrag_rowmajor(const float* __restrict__ wrow) {
    const f32x4 a = *(const f32x4*)wrow, b = *(const f32x4*)(wrow + 4);
    return (qu32x4){pack_bf16(a[0], a[1]), pack_bf16(a[2], a[3]), pack_bf16(b[0], b[1]), pack_bf16(b[2], b[3])};
}

// A fragment (row n, k-block kb) of a bf16 tile image in LDS with row stride QLD
__device__ __forceinline__ qu32x4 lds_afrag(const unsigned short* tile, int n, int g4, int kb) {
    return *(const qu32x4*)(tile + n * QLD + 32 * kb + 8 * g4);
}

// acc[tile] += A(tile image, k-blocks [KB0, KB1)) . W   (W[kb][nt] register-resident B fragments)
template <int KB0, int KB1, int NKB>
__device__ __forceinline__ void qmm(f32x4 (&acc)[2], const unsigned short* tile, int n, int g4, const qu32x4 (&w)[NKB][2]) {
    qu32x4 a[KB1 - KB0 > 0 ? KB1 - KB0 : 1];   // all LDS reads in flight before the first MFMA
#pragma unroll
    for (int kb = KB0; kb < KB1; ++kb) a[kb - KB0] = lds_afrag(tile, n, g4, kb);
#pragma unroll
    for (int kb = KB0; kb < KB1; ++kb) {
        qmfma(acc[0], a[kb - KB0], w[kb][0]);
        qmfma(acc[1], a[kb - KB0], w[kb][1]);
    }
}

// value of the lane 8 positions away inside the same 16-lane row (row_ror:8)
__device__ __forceinline__ float qswap(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));
}

// The four gate pre-activations of this lane's two cells out of the two D tiles ([i | f], [g | o]).
// hi = (n >> 3): lanes hi = 0 keep rows 0,1 of their 4-row block, lanes hi = 1 rows 2,3.
__device__ __forceinline__ void gates_of_lane(const f32x4 (&acc)[2], int hi, float (&zi)[2], float (&zf)[2], float (&zg)[2],
                                              float (&zo)[2]) {
    float snd[4], rcv[4];
    snd[0] = hi ? acc[0][0] : acc[0][2];
    snd[1] = hi ? acc[0][1] : acc[0][3];
    snd[2] = hi ? acc[1][0] : acc[1][2];
    snd[3] = hi ? acc[1][1] : acc[1][3];
#pragma unroll
    for (int k = 0; k < 4; ++k) rcv[k] = qswap(snd[k]);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        zi[r] = hi ? rcv[r] : acc[0][r];
        zf[r] = hi ? acc[0][2 + r] : rcv[r];
        zg[r] = hi ? rcv[2 + r] : acc[1][r];
        zo[r] = hi ? acc[1][2 + r] : rcv[2 + r];
    }
}

// Exchange of one (16 x 256) bf16 activation tile among the 8 workgroups of a group.
// Granule buffer of one tile exchange: [row pair 8][unit 256] x 8 bytes = 16 KB (per parity).
constexpr unsigned Q_TILE_BYTES = 8u * QH * 8u;

struct QGather {
    qu32x4 v[QNG];
};

// Thread tid = (row pair tid / 32, half (tid / 16) % 2, unit pair tid % 16) gathers units (2p, 2p + 1) of the slices
// slice + 1 + 4 half + j, j < 4, with one 16-byte load each (round 3: adjacent granules of the [row pair][unit] order, each with
// its own tag).  The fourth load of half 1 would be the own slice: switched off (offset past the descriptor).
__device__ __forceinline__ unsigned q_gather_voff(int tid) { return (unsigned)((tid >> 5) * QH + 2 * (tid & 15)) * 8u; }
__device__ __forceinline__ int q_gather_slice(int slice, int tid, int j) { return (slice + 1 + ((tid >> 4) & 1) * QNG + j) & (QG - 1); }
__device__ __forceinline__ bool q_gather_on(int tid, int j) { return !(((tid >> 4) & 1) == 1 && j == QNG - 1); }

__device__ __forceinline__ void q_gather_issue(QGather& g, const __amdgpu_buffer_rsrc_t rs, unsigned base, int slice, int tid) {
    const unsigned voff = q_gather_voff(tid);
#pragma unroll
    for (int j = 0; j < QNG; ++j)
        g.v[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, q_gather_on(tid, j) ? voff + (unsigned)(q_gather_slice(slice, tid, j) * 32) * 8u : 0x80000000u, base, 16);
}
// two granules {rows 2k, 2k+1 of unit u} and {.. of unit u+1} -> two 32-bit LDS stores (units u, u+1 are adjacent bf16)
__device__ __forceinline__ void q_tile_put(unsigned short* tile, int lo, int ld, unsigned x0, unsigned x1) {
    *(unsigned*)(tile + lo) = (x0 & 0xffffu) | (x1 << 16);
    *(unsigned*)(tile + lo + ld) = (x0 >> 16) | (x1 & 0xffff0000u);
}
// first pass: current granules go straight into the tile image, stale ones into a bit mask; retry sweeps (rare)
// re-read into loop-local temporaries.  Returns false after a give-up (status word set, caller drains).
__device__ __forceinline__ bool q_gather_finish(QGather& g, const __amdgpu_buffer_rsrc_t rs, unsigned base, int slice, int tid,
                                                unsigned epoch, unsigned short* tile, unsigned* status) {
    const int lbase = (tid >> 5) * 2 * QLD + 2 * (tid & 15);
    unsigned bad = 0;
#pragma unroll
    for (int j = 0; j < QNG; ++j) {
        if (!q_gather_on(tid, j)) continue;
        const int lo = lbase + q_gather_slice(slice, tid, j) * 32;
        if (g.v[j].y == epoch && g.v[j].w == epoch) q_tile_put(tile, lo, QLD, g.v[j].x, g.v[j].z);
        else bad |= (1u << j);
    }
    unsigned spins = 0;
    bool ok = true;
    while (__any(bad != 0)) {
        ++spins;
        if (spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(status))) {
            if ((tid & 63) == 0) xch_give_up(status);
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const unsigned voff = q_gather_voff(tid);
        qu32x4 tv[QNG];
#pragma unroll
        for (int j = 0; j < QNG; ++j)
            tv[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, q_gather_on(tid, j) ? voff + (unsigned)(q_gather_slice(slice, tid, j) * 32) * 8u : 0x80000000u, base, 16);
#pragma unroll
        for (int j = 0; j < QNG; ++j) {
            const int lo = lbase + q_gather_slice(slice, tid, j) * 32;
            if (((bad >> j) & 1u) && tv[j].y == epoch && tv[j].w == epoch) {
                q_tile_put(tile, lo, QLD, tv[j].x, tv[j].z);
                bad &= ~(1u << j);
            }
        }
    }
    return ok;
}

// ---------------------------------------------------------------------------------------------------------------
// Backward kernels: all-gather of the (16 x 1024) dz tile.  A bf16 product rounds dz to bf16 anyway, so what travels
// is the rounded dz itself, two rows per granule, instead of fp32 partial sums of dz . R^T: a lane publishes 4
// granules per step (K-split with partial sums: 16 to 48) and every workgroup then forms ITS 32 output units of
// dz . R^T from the whole tile (N-split) - no reduction across workgroups, the 8-byte write-through stores (the
// expensive side of the exchange) shrink 4 to 12 times.
// Granules of one tile and parity: [row pair 8][unit 256][gate 4] (round 3): the four gate granules of a lane's cells are 32
// adjacent bytes - two 16-byte stores to publish, two 16-byte loads per (row pair, unit, slice) to gather, each 8-byte
// granule with its own tag.
// ---------------------------------------------------------------------------------------------------------------
constexpr int QLDZ = 1024 + 16;                   // bf16 per LDS row of a dz tile: 520 dwords = 8 (mod 64): conflict-free b128 reads
constexpr unsigned Q_DZ_BYTES = 4u * 8u * QH * 8u;   // 64 KB
constexpr int QNDZ = 14;                          // 16-byte loads per thread: 7 slices * 2 gate pairs

// this lane's cells: rows row0, row0 + 1 of `unit`; dzp[g] = packed bf16 pair of gate g
__device__ __forceinline__ void q_dz_publish(const __amdgpu_buffer_rsrc_t rs, unsigned base, int row0, int unit, const unsigned (&dzp)[4],
                                             unsigned epoch, unsigned short* tile, bool same_xcd) {
    const unsigned off = (unsigned)(((row0 >> 1) * QH + unit) * 4) * 8u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const qu32x4 gr = {dzp[2 * h], epoch, dzp[2 * h + 1], epoch};
        if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off + h * 16, base, 1);
        else __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off + h * 16, base, 16);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        tile[row0 * QLDZ + g * QH + unit] = (unsigned short)(dzp[g] & 0xffffu);
        tile[(row0 + 1) * QLDZ + g * QH + unit] = (unsigned short)(dzp[g] >> 16);
    }
}
__device__ __forceinline__ bool q_dz_gather(const __amdgpu_buffer_rsrc_t rs, unsigned base, int slice, int tid, unsigned epoch,
                                            unsigned short* tile, unsigned* status) {
    const int p = tid >> 5, u = tid & 31;
    const unsigned voff = (unsigned)((p * QH + u) * 4) * 8u;
    const int lbase = 2 * p * QLDZ + u;
    auto put = [&](int j, int h, const qu32x4& q) {   // gates 2h, 2h + 1 of (row pair p, unit u of slice slice + 1 + j)
        const int lo = lbase + ((slice + 1 + j) & (QG - 1)) * 32 + 2 * h * QH;
        tile[lo] = (unsigned short)(q.x & 0xffffu);
        tile[lo + QLDZ] = (unsigned short)(q.x >> 16);
        tile[lo + QH] = (unsigned short)(q.z & 0xffffu);
        tile[lo + QH + QLDZ] = (unsigned short)(q.z >> 16);
    };
    unsigned bad = 0;
    {
        qu32x4 v[QNDZ];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                v[j * 2 + h] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + h * 16, base + (unsigned)(((slice + 1 + j) & (QG - 1)) * 32 * 4) * 8u, 16);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (v[j * 2 + h].y == epoch && v[j * 2 + h].w == epoch) put(j, h, v[j * 2 + h]);
                else bad |= (1u << (j * 2 + h));
            }
    }
    unsigned spins = 0;
    bool ok = true;
    while (__any(bad != 0)) {
        ++spins;
        if (spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(status))) {
            if ((tid & 63) == 0) xch_give_up(status);
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        // a whole new sweep, all loads in flight together (re-reading slot by slot serialises the round trips: measured
        // 10 000 cycles per stale step against 3 000 for the first sweep, tools/stamp_bf16_layer.py --bwd)
        qu32x4 tv[QNDZ];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                tv[j * 2 + h] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + h * 16, base + (unsigned)(((slice + 1 + j) & (QG - 1)) * 32 * 4) * 8u, 16);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned bit = 1u << (j * 2 + h);
                if ((bad & bit) && tv[j * 2 + h].y == epoch && tv[j * 2 + h].w == epoch) {
                    put(j, h, tv[j * 2 + h]);
                    bad &= ~bit;
                }
            }
    }
    return ok;
}

// UNIT-PAIR form of the same exchange (round 4): a lane owns the cells (row, units u0, u0 + 1) - two ADJACENT units of ONE row
// instead of one unit of two rows - so that every tape access of the pointwise phase is an 8-byte access that a wave
// coalesces into whole 128-byte lines (the kernels were bound by the number of VMEM instructions the CU's one address unit
// processes per step, tools/stamp_bf16_layer.py --bwd), and the bf16 pair of a granule is one 32-bit LDS word of the tile.
// Granule order [row 16][unit pair 128][gate 4] of {bf16 pair, epoch}: 32 bytes per (row, pair) = two 16-byte stores.
// dzp[g] = packed bf16 (unit u0 low, u0 + 1 high) of gate g; u0 even, global unit index inside the 256.
__device__ __forceinline__ void q_dz_publish2(const __amdgpu_buffer_rsrc_t rs, unsigned base, int row, int u0, const unsigned (&dzp)[4],
                                              unsigned epoch, unsigned short* tile, bool same_xcd) {
    const unsigned off = (unsigned)(row * (QH / 2) + (u0 >> 1)) * 32u;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const qu32x4 gr = {dzp[2 * h], epoch, dzp[2 * h + 1], epoch};
        if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off + h * 16, base, 1);
        else __builtin_amdgcn_raw_buffer_store_b128(gr, rs, off + h * 16, base, 16);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) *(unsigned*)(tile + row * QLDZ + g * QH + u0) = dzp[g];
}
// thread tid fetches (row tid >> 4, pair tid & 15) of each of the seven other slices
__device__ __forceinline__ bool q_dz_gather2(const __amdgpu_buffer_rsrc_t rs, unsigned base, int slice, int tid, unsigned epoch,
                                             unsigned short* tile, unsigned* status) {
    const int row = tid >> 4, pr = tid & 15;
    const unsigned voff = (unsigned)(row * (QH / 2) + pr) * 32u;
    unsigned short* lrow = tile + row * QLDZ + 2 * pr;
    auto put = [&](int j, int h, const qu32x4& q) {   // gates 2h, 2h + 1 of (row, pair pr of slice slice + 1 + j)
        unsigned short* lo = lrow + ((slice + 1 + j) & (QG - 1)) * 32 + 2 * h * QH;
        *(unsigned*)lo = q.x;
        *(unsigned*)(lo + QH) = q.z;
    };
    unsigned bad = 0;
    {
        qu32x4 v[QNDZ];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                v[j * 2 + h] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + h * 16, base + (unsigned)(((slice + 1 + j) & (QG - 1)) * 16) * 32u, 16);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                if (v[j * 2 + h].y == epoch && v[j * 2 + h].w == epoch) put(j, h, v[j * 2 + h]);
                else bad |= (1u << (j * 2 + h));
            }
    }
    unsigned spins = 0;
    bool ok = true;
    while (__any(bad != 0)) {
        ++spins;
        if (spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(status))) {
            if ((tid & 63) == 0) xch_give_up(status);
            ok = false;
            break;
        }
        __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        qu32x4 tv[QNDZ];   // a whole new sweep, all loads in flight together (see q_dz_gather)
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h)
                tv[j * 2 + h] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + h * 16, base + (unsigned)(((slice + 1 + j) & (QG - 1)) * 16) * 32u, 16);
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const unsigned bit = 1u << (j * 2 + h);
                if ((bad & bit) && tv[j * 2 + h].y == epoch && tv[j * 2 + h].w == epoch) {
                    put(j, h, tv[j * 2 + h]);
                    bad &= ~bit;
                }
            }
    }
    return ok;
}

// group / slice of a workgroup: the members of a group sit 8 blocks apart - one XCD under round-robin dispatch (a placement
// preference that the hello handshake verifies at run time; the exchange falls back to the placement-independent sc1 protocol).
// The grid is padded to a multiple of eight groups (q_padded_groups; round 4, late: a group count that is no multiple of 8 used
// to deal a group's members over all XCDs, and an exchange step costs 1.4 us more there - DESIGN 4.25): -> false for a block of
// an absent group, which counts as arrived (q_spare_leaves) and leaves.
__host__ __device__ constexpr int q_padded_groups(int num_groups) { return xch_padded_groups(num_groups); }
__device__ __forceinline__ bool q_group_slice(int num_groups, int& group, int& slice) {
    group = (blockIdx.x / (8 * QG)) * 8 + (blockIdx.x & 7);
    slice = (blockIdx.x >> 3) & (QG - 1);
    return group < num_groups;
}
// a spare block of the padded grid: it takes part in the launch protocol's arrival count (xch_settle expects the whole grid)
__device__ __forceinline__ void q_spare_leaves(unsigned* status, bool exchanging = true) { xch_spare_leaves(status, exchanging); }

// host side
int launch_layer_bf16(const LstmParams& p, hipStream_t stream);   // lstm_layer_bf16.hip
bool layer_bf16_shape_ok(int F, int H);
// lstm_stack2_bf16.hip: two stacked layers as one wavefront launch (F <= 96 -> 256 -> 256, zero initial state, one tile per group)
bool stack2_bf16_shape_ok(int B, int T, int F, int H);
int launch_stack2_bf16(const float* x, const float* K1, const float* R1, const float* b1, const float* K2, const float* R2,
                       const float* b2, float* hs1, float* hT1, float* cT1, float* res1, float* hs2, float* hT2, float* cT2,
                       float* res2, int B, int T, int F, int act, void* workspace, hipStream_t stream);
// lstm_bwd8.hip: BPTT recurrence in groups of eight workgroups (H = 256), fp32 or bf16 operands
bool bwd8_preferred(int B, int H);
int launch_bwd8(const float* R, const float* reserve, const float* c0, const float* dhs, const float* dhT, const float* dcT,
                float* dz, float* dh0, float* dc0, float* db_part, int B, int T, int act, int bf16, void* xch_ws, hipStream_t stream,
                const float* K_dx = nullptr, float* dx = nullptr);
// gemm_bf16.hip
size_t gemm_bf16_tn_scratch_floats(int M, int N);
int gemm_bf16_tn_fused(const float* a1, long lda1, long a1_so, int M1, int shift1, const float* a2, long lda2, long a2_so, int M2,
                       int shift2, const float* b, long ldb, long b_so, float* c, int ldc, int N, int RO, int RI, int bias_row,
                       int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream);
int gemm_bf16_tn(const float* a, long lda, long a_so, const float* b, long ldb, long b_so, float* c, int ldc, int M, int N, int RO,
                 int RI, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream);
int gemm_bf16_nt(const float* a, long lda, const float* b, long ldb, float* c, int ldc, int M, int N, int K, hipStream_t stream);

}  // namespace fov
