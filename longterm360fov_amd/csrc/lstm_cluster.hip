// Persistent cluster LSTM kernel for gfx950 (MI355X) - the fast path.
//
// Replaces the keras LSTM / Dense calls of mycode/FoV_seq2seq.py:83-97 and the host-driven
// decode loop :154-178 (see include/fov360.h).  Gate order i,f,c,o; weights in Keras layout.
//
// Decomposition (H = 64*G, G in {1,2,4}):
//   * a TILE is 16 sequences (the M of v_mfma_f32_16x16x4_f32);
//   * a GROUP of G workgroups owns one tile at a time; workgroup `slice` owns hidden units
//     [64*slice, 64*slice+64) for ALL FOUR gates, so the cell update is lane-local and c never
//     leaves registers;
//   * wave w of a workgroup owns 16 of those units; lane l = (g4 = l>>4, n = l&15) holds, in the
//     MFMA D layout, gate pre-activations of unit n for sequences 4*g4 .. 4*g4+3;
//   * the recurrent weight slice R[:, 4 gates x 16 units] of a wave lives in H VGPRs/AGPRs per
//     lane for the whole sequence (H=256: 256 registers), used directly as MFMA B operands;
//   * the input-kernel slice K[:, ...] sits in LDS in B-operand order (lane-linear b128 reads);
//   * the running h tile (16 x H) sits in LDS and is the MFMA A operand; x_{t+1} is prefetched
//     global -> registers -> LDS behind the MFMAs of step t;
//   * per step each workgroup publishes its 16x64 slice of h_t as 8-byte {epoch,value}
//     granules (the data is the flag) and sweeps the other G-1 slices with
//     sc1 loads until every tag equals the epoch (cdna_hip_programming.md Guideline 16, R2).
//     Since the end of round 3 granules travel in PAIRS: order [row pair][unit][row of the pair], one 16-byte
//     store / load moves two tagged granules (each half is still checked on its own tag).
//     Two parity buffers per group make the reuse race-free (a workgroup can only publish
//     epoch e+2 after every partner has consumed epoch e).
//   * every spin is bounded; a give-up sets status[0] and all workgroups drain.
//
// Two launch modes share one step body: MODE_LAYER (an LSTM layer over x) and MODE_DECODE (the
// autoregressive decoder with the Dense+tanh feedback inside the loop).  The fused inference
// path is an encoder launch followed by a decoder launch seeded from the encoder's final
// (h, c): one kernel holding both weight sets made hipcc spill.
//
// k order inside a 16-wide block is permuted so that one ds_read_b128 feeds four MFMAs:
// MFMA s of block q uses k = 16q + 4*g4 + s on lanes with (l>>4) == g4, for A and B alike.
#include <stdlib.h>

#include <atomic>
#include <mutex>

#include "fov_common.h"
#include "xch_common.h"

namespace fov {

constexpr int BT = 16;          // sequences per tile
constexpr int XR = 6;           // x prefetch registers per thread: 16 rows * F <= 256 * XR
constexpr int KPAD = 2;         // spare K-slice blocks per wave for the run-ahead LDS reads
constexpr int NXBUF = 3;        // x tiles in LDS: x_{t+2} is written while x_{t+1} and x_t may be read
constexpr int CL_MAX_F = 96;
constexpr int CL_MAX_O = 8;
constexpr unsigned SPIN_LIMIT = 1u << 20;

constexpr int MODE_LAYER = 0;   // T steps over x:(B,T,F) from (h0,c0); optional hs / hT / cT
constexpr int MODE_DECODE = 1;  // T_out autoregressive steps from (h0,c0): y_t = tanh(h_t W + bias) fed back
constexpr int MODE_LAYER_ZX = 2;   // MODE_LAYER with the input projection precomputed by the caller (p.zx).  A mode of its own:
                                // as a run-time flag its loads sat in a uniform branch, and the s_waitcnt vmcnt(0) the compiler
                                // puts at that merge also waited for the x_{t+2} request issued just before it - one exposed
                                // memory latency in every step of the plain layer

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ unsigned long long ld_granule(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_granule(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // sc1: write-through
}
// Same-XCD fast path only: an sc0 store leaves the line in the XCD's L2, where the partners'
// sc1 (L1-bypassing, L2-served) loads find it at L2-hit latency instead of a fabric round trip.
__device__ __forceinline__ void st_granule_l2(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}

__host__ __device__ constexpr int round16(int v) { return (v + 15) & ~15; }

// Diagnostic build only (-DFOV_STAMPS, tools/stamp_profile.py): per-step s_memtime stamps of one
// wave.  The shipped library compiles none of this (cdna_hip_programming.md section 7, In-kernel stamps).
#ifdef FOV_STAMPS
constexpr int STAMP_SLOTS = 12;
constexpr int STAMP_STEPS = 64;
__device__ unsigned long long g_stamps[2][STAMP_STEPS][STAMP_SLOTS];
#define FOV_STAMP(slot)                                                                         \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (stamp_on && t < STAMP_STEPS) {                                                      \
            unsigned long long t_;                                                              \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
            g_stamps[MODE & 1][t][slot] = t_;                                                       \
        }                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#define FOV_PSTAMP(slot)                                                                        \
    do {                                                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                      \
        if (blockIdx.x == 5 && threadIdx.x == 0) g_stamps[MODE & 1][STAMP_STEPS - 1][slot] = __builtin_amdgcn_s_memtime(); \
        __builtin_amdgcn_sched_barrier(0);                                                      \
    } while (0)
#else
#define FOV_STAMP(slot) do { } while (0)
#define FOV_PSTAMP(slot) do { } while (0)
#endif

struct ClusterLds {
    int ldx, ldh;
    int off_k, off_h, off_x, off_w, off_bd, off_flag, total_floats;
};
__host__ __device__ inline ClusterLds cluster_lds(int H, int F, bool decode) {
    ClusterLds L;
    const int fp = round16(F);
    L.ldx = fp + 4;
    L.ldh = H + 4;
    L.off_k = 0;
    L.off_h = L.off_k + fp * 256 + 4 * KPAD * 256;
    L.off_x = L.off_h + BT * L.ldh;
    L.off_w = L.off_x + NXBUF * BT * L.ldx;
    L.off_bd = L.off_w + (decode ? 2 * 4 * 128 : 0);   // DECODE: this workgroup's four per-wave Dense partials (16 x 8 each: two floats per lane), two step parities
    L.off_flag = L.off_bd + 8;
    L.total_floats = L.off_flag + 8 + 64;   // + tail pad for run-ahead reads of the x tile
    return L;
}

// v_mfma_f32_16x16x4_f32 through inline asm.  hipcc's allocator keeps at most 256 values in
// arch VGPRs and "spills" the rest of the weight set to AGPRs, re-reading each through
// v_accvgpr_read + s_nop before its MFMA; fp32 MFMA shares the VALU datapath, so those two
// extra issues per MFMA cost about 10 of 42 cycles (measured, profiles/r01_stamps_v2.txt).  With
// the "a" constraint the weights live in AGPRs and are MFMA B operands directly; accumulators
// and A operands stay in VGPRs.
// hipcc pads no hazards around asm (cdna_hip_programming.md 5.7), and a separate "nop" statement behind a run of MFMAs is
// not enough: where control flow merges hipcc copies accumulators between register sets - VALU reads of MFMA results -
// and puts those copies BETWEEN the last MFMA statement and the nop statement (round 3: wrong x.K terms with the
// fully unrolled input projection).  So the wait states live INSIDE the strings: the first MFMA ON EACH ACCUMULATOR of a
// run opens with two (a VALU-written accumulator / operand -> MFMA read; hipcc also moves bias-initialised accumulators
// into place one by one, right in front of their first MFMA), the last one ends with twelve (8-pass MFMA result -> any
// reader).  A run that is followed by another MFMA run on the same accumulators needs neither.
// tools/isa_mfma_hazard.py checks the generated code for both hazards.
constexpr int MF_MID = 0, MF_FIRST = 1, MF_LAST = 2;
#define FOV_MFMA_STR(pre, post) pre "v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" post
// `pos` is a constant after unrolling / inlining: exactly one of the four statements survives
template <bool W_AGPR>
__device__ __forceinline__ void mfma_f32(f32x4& acc, float a, float w, int pos) {
    if constexpr (W_AGPR) {
        if (pos == MF_MID) asm volatile(FOV_MFMA_STR("", "") : "+v"(acc) : "v"(a), "a"(w));
        else if (pos == MF_FIRST) asm volatile(FOV_MFMA_STR("s_nop 1\n\t", "") : "+v"(acc) : "v"(a), "a"(w));
        else if (pos == MF_LAST) asm volatile(FOV_MFMA_STR("", "\n\ts_nop 7\n\ts_nop 3") : "+v"(acc) : "v"(a), "a"(w));
        else asm volatile(FOV_MFMA_STR("s_nop 1\n\t", "\n\ts_nop 7\n\ts_nop 3") : "+v"(acc) : "v"(a), "a"(w));
    } else {
        if (pos == MF_MID) asm volatile(FOV_MFMA_STR("", "") : "+v"(acc) : "v"(a), "v"(w));
        else if (pos == MF_FIRST) asm volatile(FOV_MFMA_STR("s_nop 1\n\t", "") : "+v"(acc) : "v"(a), "v"(w));
        else if (pos == MF_LAST) asm volatile(FOV_MFMA_STR("", "\n\ts_nop 7\n\ts_nop 3") : "+v"(acc) : "v"(a), "v"(w));
        else asm volatile(FOV_MFMA_STR("s_nop 1\n\t", "\n\ts_nop 7\n\ts_nop 3") : "+v"(acc) : "v"(a), "v"(w));
    }
}
// position of MFMA number i of a run of n: the flags say whether the run opens / closes an MFMA sequence
__host__ __device__ constexpr int mf_pos(bool first_run, bool last_run, bool is_first, bool is_last) {
    return ((first_run && is_first) ? MF_FIRST : 0) | ((last_run && is_last) ? MF_LAST : 0);
}

// Load one LSTM's weights for this wave: R slice -> registers, K slice -> LDS, bias -> registers.
// Register block j holds the k-block of hidden units ((slice + j/4) mod G)*64 + (j%4)*16 .. +16:
// the workgroup's OWN 64 units come first (j < 4), so the part of h_t . R that needs no remote
// data can start before the gather of the partner slices has landed.
// Two calls: R blocks [J0, J1) only (REST = false), then the remaining blocks, bias and K (REST = true).
template <int H, bool DEC_KMAP, int J0, int J1, bool REST>
__device__ __forceinline__ void load_weights(float (&wR)[H / 16][4][4], float (&bias)[4], float* sKw,
                                             const float* K, const float* R, const float* b, int F, int Fp,
                                             int col0, int slice, int lane) {
    constexpr int G = H / 64;
    const int n = lane & 15, g4 = lane >> 4;
    const int H4 = 4 * H;
    // Buffer loads: one per-lane offset for the whole slice, the (j, s, g) part of the address is wave-uniform and goes
    // into the instruction's scalar offset - no per-load address arithmetic, and nothing keeps the loads from being
    // issued back to back (the counter allows 63 in flight).
    {
        const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R), 0, H * H4 * 4, 0x00020000);
        const unsigned voff = (unsigned)((4 * g4 * H4 + col0 + n) * 4);
#pragma unroll
        for (int j = J0; j < J1; ++j) {
            const unsigned kb = (unsigned)((((slice + (j >> 2)) & (G - 1)) * 64 + (j & 3) * 16) * H4 * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    wR[j][s][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, voff, kb + (unsigned)((s * H4 + g * H) * 4), 0));
        }
    }
    if constexpr (!REST) return;
    {
        const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, b ? H4 * 4 : 0, 0x00020000);
#pragma unroll
        for (int g = 0; g < 4; ++g) bias[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)((g * H + col0 + n) * 4), 0, 0));
    }
    // K slice in B-operand order: block (q,s) = 64 lanes x {i,f,c,o} of input row k = 16q+4*g4+s.
    // DECODE (F <= 8, one q block): k = 4*s + g4 instead, so that the two MFMA steps s = 0,1 cover
    // k = 0..7 and the Dense output fragment (y[n][4r + g4] in register r) is their A operand.
    // All loads are unconditional (the descriptor ends with row F - 1: rows k >= F read as 0) and issued before the
    // first LDS store: a load inside a branch is waited for at the merge - with a branch per (q, s) block the prologue
    // was up to 24 dependent memory round trips long.
    const int nq = Fp >> 4;
    constexpr int QMAX = CL_MAX_F / 16;
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, F * H4 * 4, 0x00020000);
    f32x4 kv[DEC_KMAP ? 1 : QMAX][4];
#pragma unroll
    for (int q = 0; q < (DEC_KMAP ? 1 : QMAX); ++q)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int k = DEC_KMAP ? (s < 2 ? 4 * s + g4 : F) : 16 * q + 4 * g4 + s;
            const unsigned voff = (unsigned)((k * H4 + col0 + n) * 4);   // k >= F: past the descriptor
#pragma unroll
            for (int g = 0; g < 4; ++g)
                kv[q][s][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(krs, voff + (unsigned)(g * H * 4), 0, 0));
        }
#pragma unroll
    for (int q = 0; q < (DEC_KMAP ? 1 : QMAX); ++q)
        if (q < nq) {
#pragma unroll
            for (int s = 0; s < 4; ++s) *(f32x4*)(sKw + ((q * 4 + s) * 64 + lane) * 4) = kv[q][s];
        }
}

// One k-block (16 loads) of the wave's R slice; `j` is a constant after unrolling.
template <int H>
__device__ __forceinline__ void load_r_block(float (&wR)[H / 16][4][4], int j, const float* R, int col0, int slice, int lane) {
    constexpr int G = H / 64;
    const int n = lane & 15, g4 = lane >> 4;
    const int H4 = 4 * H;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R), 0, H * H4 * 4, 0x00020000);
    const unsigned voff = (unsigned)((4 * g4 * H4 + col0 + n) * 4);
    const unsigned kb = (unsigned)((((slice + (j >> 2)) & (G - 1)) * 64 + (j & 3) * 16) * H4 * 4);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            wR[j][s][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, voff, kb + (unsigned)((s * H4 + g * H) * 4), 0));
}
// DECODE: bias and the two K blocks of this lane (input row k = 4*s + g4, see load_weights) straight into registers
template <int H>
__device__ __forceinline__ void load_dec_small(float (&bias)[4], f32x4 (&kb)[2], const float* K, const float* b, int F, int col0, int lane) {
    const int n = lane & 15, g4 = lane >> 4;
    const int H4 = 4 * H;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, b ? H4 * 4 : 0, 0x00020000);
#pragma unroll
    for (int g = 0; g < 4; ++g) bias[g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)((g * H + col0 + n) * 4), 0, 0));
    const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(K), 0, F * H4 * 4, 0x00020000);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        const unsigned voff = (unsigned)(((4 * s + g4) * H4 + col0 + n) * 4);   // k >= F: past the descriptor
#pragma unroll
        for (int g = 0; g < 4; ++g) kb[s][g] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(krs, voff + (unsigned)(g * H * 4), 0, 0));
    }
}

// acc += A(16 x Fp, LDS rows of stride ldx) . Kslice(LDS); B reads run two (q,s) blocks ahead.
// No clamping: the K slice carries KPAD spare blocks per wave and the x tiles a spare tail, so
// the run-ahead reads of the last iterations stay inside LDS (their values are never used) and
// every block offset is an instruction immediate instead of VALU work.  The last k-block is peeled: its last MFMA
// carries the closing wait states.
template <bool FIRST, bool LAST>
__device__ __forceinline__ void input_proj(f32x4 (&acc)[4], const float* arow, const float* sKw, int nq, int lane) {
    if (nq <= 0) return;
    const float* bl = sKw + lane * 4;
    f32x4 a = *(const f32x4*)arow;
    f32x4 b0 = *(const f32x4*)bl;
    f32x4 b1 = *(const f32x4*)(bl + 256);
    if (FIRST) asm volatile("s_nop 1" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]));   // (rare path; the first MFMA sits in a loop or in the peeled block)
    for (int q = 0; q < nq - 1; ++q) {
        const f32x4 an = *(const f32x4*)(arow + 16 * (q + 1));
        const float* bq = bl + q * 1024;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 bn = *(const f32x4*)(bq + (s + 2) * 256);
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_f32<false>(acc[g], a[s], b0[g], MF_MID);
            b0 = b1;
            b1 = bn;
        }
        a = an;
    }
    const float* bq = bl + (nq - 1) * 1024;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const f32x4 bn = *(const f32x4*)(bq + (s + 2) * 256);
#pragma unroll
        for (int g = 0; g < 4; ++g) mfma_f32<false>(acc[g], a[s], b0[g], mf_pos(false, LAST, false, s == 3 && g == 3));
        b0 = b1;
        b1 = bn;
    }
}

// The same with the block count known at compile time: every LDS offset is an instruction immediate and the operand
// registers are renamed instead of rotated - the run-time loop above carries two v_mov_b64 and two v_add_u32 per 16
// MFMAs, and on this chip a VALU instruction is never hidden behind an fp32 MFMA (tools/microbench/mfma_f32_overlap.hip:
// 32 -> 46 cycles for ONE v_fma_f32 in an MFMA gap).
template <int NQC, bool FIRST, bool LAST>
__device__ __forceinline__ void input_proj_fixed(f32x4 (&acc)[4], const float* arow, const float* sKw, int lane) {
    const float* bl = sKw + lane * 4;
    f32x4 a = *(const f32x4*)arow;
    f32x4 b0 = *(const f32x4*)bl;
    f32x4 b1 = *(const f32x4*)(bl + 256);
#pragma unroll
    for (int q = 0; q < NQC; ++q) {
        f32x4 an = a;
        if (q + 1 < NQC) an = *(const f32x4*)(arow + 16 * (q + 1));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            f32x4 bn = b1;
            if (4 * q + s + 2 < 4 * NQC) bn = *(const f32x4*)(bl + (4 * q + s + 2) * 256);
#pragma unroll
            for (int g = 0; g < 4; ++g)
                mfma_f32<false>(acc[g], a[s], b0[g], mf_pos(FIRST, LAST, q == 0 && s == 0, q == NQC - 1 && s == 3 && g == 3));
            b0 = b1;
            b1 = bn;
        }
        a = an;
    }
}
template <bool FIRST, bool LAST>
__device__ __forceinline__ void input_proj_any(f32x4 (&acc)[4], const float* arow, const float* sKw, int nq, int lane) {
#ifndef FOV_DBG_NO_XK_UNROLL
    if (nq == 6) input_proj_fixed<6, FIRST, LAST>(acc, arow, sKw, lane);   // F in (80, 96]: the reference's 90-wide input (FoV_seq2seq.py:24-26)
    else
#endif
        input_proj<FIRST, LAST>(acc, arow, sKw, nq, lane);
}

// DECODE: acc += y(16 x 8) . Kslice, y fragment and the two K blocks already in registers (y is VALU-written: the run
// always opens with the two wait states)
template <bool LAST>
__device__ __forceinline__ void input_proj_reg(f32x4 (&acc)[4], f32x4 y4, const f32x4 (&kb)[2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int g = 0; g < 4; ++g) mfma_f32<false>(acc[g], y4[s], kb[s][g], mf_pos(true, LAST, s == 0, s == 1 && g == 3));
}

// acc += h tile (LDS, columns in rotated slice order) . register-resident R blocks [J0, J1)
template <int H, int J0, int J1, bool FIRST, bool LAST>
__device__ __forceinline__ void recurrent(f32x4 (&acc)[4], const float* hrow, const float (&wR)[H / 16][4][4]) {
    if (J0 >= J1) return;
    f32x4 a = *(const f32x4*)(hrow + 16 * J0);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        f32x4 an = a;
        if (j + 1 < J1) an = *(const f32x4*)(hrow + 16 * (j + 1));
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                mfma_f32<true>(acc[g], a[s], wR[j][s][g], mf_pos(FIRST, LAST, j == J0 && s == 0, j == J1 - 1 && s == 3 && g == 3));
        a = an;
    }
}

// What the encoder phase of the fused (one-launch) encode + decode kernel hands to its decoder phase: the epoch / ticket
// of the launch, the outcome of the hello handshake, and the final (h, c) of this lane's four cells.  The complete h_T
// tile stays in LDS (the encoder phase exchanges its last h as well).
struct ClusterCarry {
    unsigned epoch;
    XchTicket ticket;
    bool same_xcd, aborted, xch_used;
    float c[4], h[4];
};

// FUSED: 0 = the whole launch is this phase; 1 = encoder phase of the fused kernel (its last step is exchanged too, the
// state leaves through `cy` instead of memory, nothing is settled); 2 = decoder phase (no header read, no handshake, no
// state loads: everything comes from `cy` and the h tile the encoder phase left in LDS).  One tile per group only.
template <int H, int ACT, int MODE, int FUSED>
__device__ __forceinline__ void cluster_body(const LstmParams& p, ClusterCarry& cy) {
    constexpr int G = H / 64;
    constexpr int NQ = H / 16;
    constexpr int NG = (G - 1) * 2;  // 16-byte loads (two adjacent units' tagged granules) per thread per step
    constexpr bool LAYER = (MODE != MODE_DECODE);
    constexpr bool ZX = (MODE == MODE_LAYER_ZX);      // input projection precomputed by the caller
    constexpr bool F1 = (FUSED == 1), F2 = (FUSED == 2);
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    constexpr unsigned OORB = 0x80000000u;   // buffer-load offset no descriptor covers: reads as 0
#ifdef FOV_STAMPS
    if (blockIdx.x == 5 && tid == 0) g_stamps[MODE & 1][STAMP_STEPS - 1][0] = __builtin_amdgcn_s_memtime();   // phase entry
#endif
    // decoder phase of the fused kernel: the h_T tile sits where the ENCODER phase's LDS layout put it; take it into
    // registers before this phase's layout (its K slice is smaller, everything behind it moves) overwrites anything
    float hcarry[BT * H / 256];
    if constexpr (F2) {
        const ClusterLds L1 = cluster_lds(H, p.F, false);
#pragma unroll
        for (int q = 0; q < BT * H / 256; ++q) {
            const int i = tid + 256 * q;
            hcarry[q] = smem[L1.off_h + (i / H) * L1.ldh + (i - (i / H) * H)];
        }
        __syncthreads();
    }
    // Blocks b and b+8 are observed to share an XCD (round-robin dispatch), so when the group
    // count allows it the G members of a group are 8 blocks apart.  This is a SPEED choice only:
    // whether the members really share an XCD is verified at run time below.
    int group, slice;
    if (G > 1 && ((p.num_groups & 7) == 0 || p.xcd_pad)) {
        group = (blockIdx.x / (8 * G)) * 8 + (blockIdx.x & 7);
        slice = (blockIdx.x >> 3) & (G - 1);
    } else {
        group = blockIdx.x / G;
        slice = blockIdx.x - group * G;
    }
    const int col0 = slice * 64 + wave * 16;  // first hidden unit of this wave

    // weights / input of the phase this launch runs
    const float* Kp = LAYER ? p.K : p.dK;
    const float* Rp = LAYER ? p.R : p.dR;
    const float* bp = LAYER ? p.b : p.db;
    const int F = LAYER ? (ZX ? 0 : p.F) : p.F_dec;
    const int steps = LAYER ? p.T : p.T_out;
    const int Fp = round16(F);
    const int nq = Fp >> 4;

    const ClusterLds L = cluster_lds(H, F, !LAYER);
    const int LDX = L.ldx, LDH = L.ldh;
    float* sK = smem + L.off_k;
    float* sH = smem + L.off_h;   // h tile, column = ((unit/64 - slice) mod G)*64 + unit%64 (own slice first)
    float* sX = smem + L.off_x;   // LAYER: three x tiles (t mod 3); DECODE: the y tile
    float* sY = smem + L.off_w;   // DECODE: [step parity][wave][lane][2]: the four waves' Dense partials of this workgroup
    int* sFlag = (int*)(smem + L.off_flag);
    float* sKw = sK + wave * (nq * 4 + KPAD) * 256;  // this wave's K slice in B-operand order

    // A layer's last step publishes nothing (no later step reads h_T from the partners), so a one-step layer
    // launch - the unit the step-wise decoders of a4 are built from - needs no exchange and no handshake.
    const bool xch_used = (G > 1) && (!LAYER || p.T > 1 || F1);
    // Epoch tags continue from the workspace header (xch_common.h): no memset between launches.  A workspace whose
    // sticky timeout word is set is poisoned: the body is skipped (fail-stop) until fov_check_status clears it.
    //
    // Order (round 3, prologue stamps of tools/stamp_profile.py): thread 0 requests the header words, the first 48 weight
    // loads go out, thread 0 PUBLISHES its hello word and counts the workgroup as arrived (xch_common.h), the other
    // 300-odd weight loads of every wave follow (5-6 us), and only then are the partners' hello words polled.  With the publish behind all weight
    // loads every group waited for its slowest member's loads PLUS a store-to-visible round trip: 5.6 us between the
    // ticket and the end of the handshake.
    unsigned* sXch = (unsigned*)(sFlag + 4);     // base / launch index of this launch (thread 0 -> all, xch_common.h)
    const bool hdr = xch_used && !F2;
    // thread 0 requests the header words; their round trip runs under the first weight loads
    unsigned hdr_base = 0, hdr_launch = 0;
    if (hdr && tid == 0) {
        hdr_base = xch_status_load(p.status + ST_EPOCH);
        hdr_launch = xch_status_load(p.status + ST_LAUNCHES);
    }
    float wR[NQ][4][4];   // [k-block j][k-sub s][gate g], AGPR-resident
    float bias[4];
    constexpr int JA = NQ < 3 ? NQ : 3;
    // Decoder phase of the fused kernel: the R slice is STREAMED - three k-blocks here, the others three blocks ahead of the
    // MFMAs of z_0 that consume them (256 loads per wave take 4 us, z_0's 264 MFMAs another 4: now side by side); bias, K
    // and the Dense kernel go out first and stay in registers (no LDS round trip, no barrier).
    constexpr int STREAM_LA = 3;
    f32x4 kb[2];   // DECODE: the two K-slice blocks of this lane (loop invariant)
    if constexpr (F2) {
        load_dec_small<H>(bias, kb, Kp, bp, F, col0, lane);
    } else {
        load_weights<H, !LAYER, 0, JA, false>(wR, bias, sKw, Kp, Rp, bp, F, Fp, col0, slice, lane);
    }
    unsigned long long* hello = p.xch + (size_t)p.num_groups * 2 * BT * H + (size_t)group * G;
    const unsigned my_xcc = xcc_id();
    if (hdr && tid == 0) {
        sXch[0] = hdr_base;
        sXch[1] = hdr_launch;
        // hello handshake (safe sc1 protocol): do all members of this group sit on one XCD?  The tag is larger than any tag
        // of an earlier launch.
        st_granule(hello + slice, (((unsigned long long)hdr_base + 1ull) << 32) | my_xcc);
        xch_count_arrival(p.status);   // both header words have landed (stored above)
    }
    FOV_PSTAMP(3);
    if constexpr (!F2) load_weights<H, !LAYER, JA, NQ, true>(wR, bias, sKw, Kp, Rp, bp, F, Fp, col0, slice, lane);
    FOV_PSTAMP(2);
    const bool poisoned = hdr && xch_poisoned(p.status);   // one wave-wide load per wave; wave 0's value decides (sFlag[0])
    const unsigned arrival = 0;
    if (tid == 0) { sFlag[0] = (poisoned || (F2 && cy.aborted)) ? 1 : 0; sFlag[1] = 0; }
    if (hdr && !poisoned && tid < G) {   // (lanes of wave 0: thread 0 has just written sFlag, program order)
        const unsigned long long hello_tag = (unsigned long long)sXch[0] + 1ull;
        unsigned long long hv = 0;
        unsigned spins = 0;
        while (true) {
            hv = ld_granule(hello + tid);
            if ((hv >> 32) == hello_tag) break;
            if (++spins > SPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                xch_give_up(p.status);
                sFlag[0] = 1;
                break;
            }
            __builtin_amdgcn_s_sleep(4);
        }
        if ((unsigned)hv != my_xcc) sFlag[1] = 1;   // a partner lives on another XCD
    }
    FOV_PSTAMP(4);
    // x_0, x_1 of the FIRST tile are requested here, in front of the prologue's barrier (their round trip used to sit between
    // that barrier and the tile's: 1.5 us of the launch's fixed part)
    float xpre[2][XR];
    {
        const int xrw0 = tid >> 4, xcl0 = tid & 15;
        const bool xuse0 = LAYER && !ZX;
        const int b00 = group * BT;
        const int live0 = (group < p.num_tiles) ? (p.B - b00 < BT ? p.B - b00 : BT) : 0;
        const __amdgpu_buffer_rsrc_t xg0 = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xuse0 && live0 > 0 ? p.x + (size_t)b00 * p.T * F : nullptr), 0, xuse0 ? live0 * p.T * F * 4 : 0, 0x00020000);
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < XR; ++i)
                xpre[tt][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                    xg0, (tt < steps && xcl0 + 16 * i < F) ? (unsigned)((xrw0 * p.T * F + xcl0 + 16 * i) * 4) : OORB, (unsigned)(tt * F * 4), 0));
    }
    // zero the x tiles once: pad columns [F, Fp) are never written afterwards
    for (int i = tid; i < NXBUF * BT * LDX; i += 256) sX[i] = 0.f;

    // DECODE: Dense(F_dec, tanh) runs on the matrix pipe, transposed: y^T = Wd^T . h^T, and is DISTRIBUTED (round 5): every wave
    // forms the partial product over ITS OWN 16 units (4 MFMAs on the columns it wrote to LDS) in the shadow of the gather of
    // h_t, the four waves' partials meet in LDS at barrier 2, wave 0 publishes their sum - the workgroup's partial, 16 x 8
    // values as tagged granules - and every wave fetches the G - 1 partner partials in the middle of the next step's
    // partner-slice MFMAs: y_t is complete when those MFMAs are, nothing of the Dense sits between the gather and them and it
    // has no barrier of its own.  (Until round 4: 16 MFMAs per wave over a quarter of the gathered tile right behind the
    // gather, four partials meeting in LDS behind a third barrier.)
    // Lane (i = l&15, g4) keeps Wd[unit = col0 + 4*g4 + s][o(i)], o(i) = 4*(i&3) + (i>>2), as MFMA A operands (zero where
    // o(i) >= F_dec).  With that row order register r of the D fragment on lane (n, g4) is y[n][4r + g4]: registers 0, 1
    // are what is published and, summed over the partials, the A operands of the two y . K MFMA steps (batch on the
    // lane, o = 4*g4 + r in the registers): y never leaves that layout.
    float wd[4];
    float bd4[4];
    if (!LAYER) {
        const int O = p.F_dec;
        // unconditional buffer loads (out-of-range offset = 0): a load inside a branch is waited for at the merge
        const __amdgpu_buffer_rsrc_t wdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dW), 0, H * O * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t bdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dbias), 0, O * 4, 0x00020000);
#pragma unroll
        for (int ss = 0; ss < 4; ++ss) {
            const int unit = col0 + 4 * g4 + ss;
            const int o = 4 * (n & 3) + (n >> 2);
            wd[ss] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wdrs, (o < O) ? (unsigned)((unit * O + o) * 4) : OORB, 0, 0));
        }
#pragma unroll
        for (int ss = 0; ss < 4; ++ss)
            bd4[ss] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bdrs, (4 * ss + g4 < O) ? (unsigned)((4 * ss + g4) * 4) : OORB, 0, 0));
    }
    if constexpr (F2) {
#pragma unroll
        for (int j = 0; j < STREAM_LA; ++j) load_r_block<H>(wR, j, Rp, col0, slice, lane);
    } else if (!LAYER) {
        __syncthreads();   // K slice written by load_weights above
        kb[0] = *(const f32x4*)(sKw + lane * 4);
        kb[1] = *(const f32x4*)(sKw + 256 + lane * 4);
    }

    __syncthreads();
    FOV_PSTAMP(5);
    XchTicket ticket = {0u, 0u, 0u};
    if (xch_used && !F2) ticket = xch_ticket(sXch, arrival);
    if constexpr (F2) ticket = cy.ticket;
    unsigned epoch = F2 ? cy.epoch : ticket.base;
    const bool same_xcd = F2 ? cy.same_xcd : (xch_used && (sFlag[1] == 0) && (p.force_safe_exchange == 0));
    bool aborted = F2 ? cy.aborted : (xch_used && sFlag[0] != 0);   // poisoned workspace, or a partner never showed up: drain
    if (xch_used && !F2 && tid == 0 && !same_xcd && !aborted)   // number of workgroups on the safe (cross-XCD) exchange
        xch_count_safe(p.status, ticket);
#ifdef FOV_STAMPS
    const bool stamp_on = (blockIdx.x == 5 && tid == 0);
#endif
    const float* hrow = sH + n * LDH + 4 * g4;   // this lane's A-operand row of the h tile

    // Granule I/O goes through one buffer descriptor per group (both parity buffers) with 32-bit
    // per-lane offsets computed once: fp32 MFMA shares the VALU, so 64-bit address arithmetic
    // inside the step would come straight out of the MFMA rate.
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 2 * BT * H, 0, 2 * BT * H * (int)sizeof(unsigned long long), 0x00020000);
    // granule order [row pair][unit][row of the pair] (round 3): a lane publishes its four rows as TWO 16-byte stores, a gather
    // load brings both rows of one unit; every 8-byte granule keeps its own tag
    const unsigned pub_off = (unsigned)((2 * g4) * H + col0 + n) * 16u;
    unsigned goff[NG > 0 ? NG : 1];
    int loff[NG > 0 ? NG : 1];
#pragma unroll
    for (int j = 0; j < NG; ++j) {
        // load `within` of partner slice `osl`: row pair within / 64, unit within % 64 - both rows of the unit in one load
        const int idx = j * 256 + tid;
        const int rot = (idx >> 9) + 1, within = idx & 511;
        const int osl = (slice + rot) & (G - 1);
        goff[j] = (unsigned)((within >> 6) * H + osl * 64 + (within & 63)) * 16u;
        loff[j] = 2 * (within >> 6) * LDH + rot * 64 + (within & 63);
    }

    // DECODE: the workgroups' Dense partials travel like h - tagged granules in an area of their own behind the hello words,
    // [group][parity][slice][lane] x 16 bytes {y[n][g4], epoch, y[n][4 + g4], epoch}; every lane of every wave fetches its own
    // element of each workgroup's partial.
    constexpr int NYG = (LAYER || G == 1) ? 0 : G;    // 16-byte loads per lane per step for them (its own workgroup's comes the same way)
    const __amdgpu_buffer_rsrc_t yrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)p.num_groups * 2 * BT * H + (size_t)p.num_groups * G + (size_t)group * 2 * G * 128, 0,
        LAYER ? 0 : 2 * G * 128 * (int)sizeof(unsigned long long), 0x00020000);
    const unsigned ypub_off = (unsigned)((slice * 64 + lane) * 16);
    const unsigned ygoff = (unsigned)(lane * 16);      // + q * 1024: the partial of slice q (an instruction immediate)
    constexpr unsigned YPAR_BYTES = G * 64 * 16;   // one parity of the granule area
    constexpr int YPAR_LDS = 4 * 128;              // one parity of the LDS partials (floats)
    // y = tanh(sum of the G workgroup partials in absolute slice order + bias), each partial ((w0 + w1) + (w2 + w3)): every
    // workgroup of the group forms the same y bit for bit
    auto own_partial = [&](int par) -> f32x2 {
        const float* yb = sY + par * YPAR_LDS + lane * 2;
        const f32x2 a0 = *(const f32x2*)yb, a1 = *(const f32x2*)(yb + 128), a2 = *(const f32x2*)(yb + 256), a3 = *(const f32x2*)(yb + 384);
        return (a0 + a1) + (a2 + a3);
    };

    const bool h_zero = !F2 && p.h0 == nullptr;   // (the decoder phase of the fused kernel starts from the encoder's state)
    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * BT;
        // ---- initial state (the previous tile ended on a barrier) ----
        // Every global load of the tile goes through a buffer descriptor that covers exactly the tile's live rows (a NULL
        // tensor: nothing): rows past the batch, absent tensors and masked columns read as 0 WITHOUT a branch, so the
        // loads are issued back to back and nothing waits for them before their first use.
        const int live_rows = p.B - b0 < BT ? p.B - b0 : BT;
        const __amdgpu_buffer_rsrc_t c0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.c0 ? p.c0 + (size_t)b0 * H : nullptr), 0, p.c0 ? live_rows * H * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t h0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h0 ? p.h0 + (size_t)b0 * H : nullptr), 0, p.h0 ? live_rows * H * 4 : 0, 0x00020000);
        float c[4], hcur[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const unsigned off = (unsigned)(((4 * g4 + r) * H + col0 + n) * 4);
            if constexpr (F2) {   // handed over in registers by the encoder phase
                c[r] = cy.c[r];
                hcur[r] = cy.h[r];
            } else {
                c[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, off, 0, 0));
                hcur[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, off, 0, 0));
            }
        }
        {
            float hv[BT * H / 256];
#pragma unroll
            for (int q = 0; q < BT * H / 256; ++q) {
                const int i = tid + 256 * q;
                const int row = i / H, pos = i - row * H;
                const int unit = ((slice + (pos >> 6)) & (G - 1)) * 64 + (pos & 63);
                if constexpr (F2) hv[q] = hcarry[q];   // the tile the encoder phase left in LDS (same rotated column order)
                else hv[q] = h_zero ? 0.f : __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, (unsigned)((row * H + unit) * 4), 0, 0));
            }
#pragma unroll
            for (int q = 0; q < BT * H / 256; ++q) {
                const int i = tid + 256 * q;
                const int row = i / H, pos = i - row * H;
                sH[row * LDH + pos] = hv[q];
            }
        }
        // x staging: thread (xrw = tid/16, xcl = tid%16) moves columns xcl + 16*i of row xrw
        const int xrw = tid >> 4, xcl = tid & 15;
        const bool xuse = LAYER && !ZX;
        const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(xuse ? p.x + (size_t)b0 * p.T * F : nullptr), 0, xuse ? live_rows * p.T * F * 4 : 0, 0x00020000);
        unsigned xoff[XR];   // byte offset of this thread's elements of step 0 (masked columns: out of range)
#pragma unroll
        for (int i = 0; i < XR; ++i) xoff[i] = (xcl + 16 * i < F) ? (unsigned)((xrw * p.T * F + xcl + 16 * i) * 4) : OORB;
        // ZX mode: this lane's 16 pre-activations of step t live at (r*T + t)*4H + g*H of the tile's rows
        const __amdgpu_buffer_rsrc_t zxrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(ZX ? p.zx + (size_t)b0 * p.T * 4 * H : nullptr), 0, ZX ? live_rows * p.T * 4 * H * 4 : 0, 0x00020000);
        const unsigned zxoff = (unsigned)((4 * g4 * p.T * 4 * H + col0 + n) * 4);
        f32x4 zr[4];   // prefetched pre-activations of the NEXT step (ZX mode)
        float* xl = sX + xrw * LDX + xcl;
        if (xuse) {
            float x2[2][XR];
            if (tile == group) {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int i = 0; i < XR; ++i) x2[tt][i] = xpre[tt][i];
            } else {
#pragma unroll
                for (int tt = 0; tt < 2; ++tt)
#pragma unroll
                    for (int i = 0; i < XR; ++i)
                        x2[tt][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, tt < steps ? xoff[i] : OORB, (unsigned)(tt * F * 4), 0));
            }
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (tt < steps) {
#pragma unroll
                    for (int i = 0; i < XR; ++i)
                        if (xcl + 16 * i < F) xl[tt * BT * LDX + 16 * i] = x2[tt][i];
                }
        }
        // DECODE output: out[b0 + n][t][4*ss + g4]; the descriptor ends with the tile's last live row, masked outputs are out of range
        const __amdgpu_buffer_rsrc_t yors = __builtin_amdgcn_make_buffer_rsrc(
            (!LAYER && p.out) ? p.out + (size_t)b0 * p.T_out * p.F_dec : nullptr, 0, (!LAYER && p.out) ? live_rows * p.T_out * p.F_dec * 4 : 0, 0x00020000);
        unsigned yoff[2];
#pragma unroll
        for (int ss = 0; ss < 2; ++ss)
            yoff[ss] = (!LAYER && 4 * ss + g4 < p.F_dec) ? (unsigned)((n * p.T_out * p.F_dec + 4 * ss + g4) * 4) : OORB;
        f32x4 y4 = (f32x4){0.f, 0.f, 0.f, 0.f};   // DECODE: y_{t-1}[n][4*s + g4], the A fragment of y . K
        if (!LAYER) {
            // (buffer loads with an out-of-range offset for the masked elements: a `cond ? load : 0` sits in a branch whose
            // merge waits for every older load - here the streamed R slice)
            const __amdgpu_buffer_rsrc_t y0rs = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float*>(p.dec_in0 + (size_t)b0 * F), 0, live_rows * F * 4, 0x00020000);
#pragma unroll
            for (int ss = 0; ss < 4; ++ss)
                y4[ss] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(y0rs, (4 * ss + g4 < F) ? (unsigned)((n * F + 4 * ss + g4) * 4) : OORB, 0, 0));
        }
        __syncthreads();
        FOV_PSTAMP(6);

        // ---- pre-activations of step 0 that need no remote data ----
        f32x4 acc[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) acc[g] = (f32x4){bias[g], bias[g], bias[g], bias[g]};
        if (ZX && steps > 0) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    acc[g][r] = bias[g] + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(zxrs, zxoff + (unsigned)((r * p.T * 4 * H + g * H) * 4), 0, 0));
        }
        if (steps > 0) {
            if (LAYER) input_proj_any<true, true>(acc, sX + n * LDX + 4 * g4, sKw, nq, lane);
            else input_proj_reg<true>(acc, y4, kb);
            // a zero initial state (the encoder of every seq2seq call): h_0 . R is exactly 0 - its 4H/16 k-blocks (256 of the
            // 352 MFMAs of an encoder step at H = 256) are skipped here and at the top of step 0
            if constexpr (F2) {
                // the whole of h_T . R here (the loop's first partner-slice run is skipped), block j + 3 requested before the
                // MFMAs of block j
                f32x4 a = *(const f32x4*)hrow;
#pragma unroll
                for (int j = 0; j < NQ; ++j) {
                    if (j + STREAM_LA < NQ) load_r_block<H>(wR, j + STREAM_LA, Rp, col0, slice, lane);
                    f32x4 an = a;
                    if (j + 1 < NQ) an = *(const f32x4*)(hrow + 16 * (j + 1));
#pragma unroll
                    for (int s2 = 0; s2 < 4; ++s2)
#pragma unroll
                        for (int g = 0; g < 4; ++g)
                            mfma_f32<true>(acc[g], a[s2], wR[j][s2][g], mf_pos(false, true, false, j == NQ - 1 && s2 == 3 && g == 3));
                    a = an;
                }
            } else if (!h_zero) {
                recurrent<H, 0, 4, true, true>(acc, hrow, wR);
            }
        }

        float xr[XR];
#pragma unroll
        for (int i = 0; i < XR; ++i) xr[i] = 0.f;
        for (int t = 0; t < steps; ++t) {
            FOV_STAMP(0);
#ifdef FOV_STAMPS
            if (stamp_on && t < STAMP_STEPS) g_stamps[MODE & 1][t][9] = __builtin_amdgcn_s_memrealtime();
#endif
            // ---- x pipeline: x_{t+1} (loaded during step t-1) goes registers -> LDS now; its tile
            // was last read two steps ago and is next read after barrier 1b of this step.  Then
            // x_{t+2} is requested, so every load has a full step to land. ----
            if (LAYER && !ZX && t > 0 && t + 1 < steps) {
                float* xb = xl + ((t + 1) % 3) * BT * LDX;
#pragma unroll
                for (int i = 0; i < XR; ++i)
                    if (xcl + 16 * i < F) xb[16 * i] = xr[i];
            }
            if (ZX && t + 1 < steps) {
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        zr[g][r] = bias[g] + __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(
                                                 zxrs, zxoff + (unsigned)(((r * p.T + (t + 1)) * 4 * H + g * H) * 4), 0, 0));
            }
            if (LAYER && !ZX && t + 2 < steps) {
#pragma unroll
                for (int i = 0; i < XR; ++i)
                    xr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, xoff[i], (unsigned)((t + 2) * F * 4), 0));
            }
            // ---- the part of h_{t-1} . R that needed the partner slices ----
            if (LAYER || t == 0) {
                if (!((h_zero || F2) && t == 0)) recurrent<H, 4, NQ, true, true>(acc, hrow, wR);
            } else {
                // DECODE: y_{t-1}.  This workgroup's four wave partials are in LDS since barrier 2 of the previous step; the partners'
                // partials were published right behind THEIR barrier 2, about when this step began - their loads go out half way
                // through these MFMAs (which do not need y) and have the other half to land; then y_{t-1} . K completes z_t
                constexpr int JM = 4 + (NQ - 4) / 2;
                recurrent<H, 4, JM, true, false>(acc, hrow, wR);
                u32x4g vy[NYG > 0 ? NYG : 1];
                const unsigned yso = (epoch & 1u) * YPAR_BYTES;      // (the epoch is still the one h_{t-1} and y_{t-1} were tagged with)
#pragma unroll
                for (int q = 0; q < NYG; ++q) vy[q] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ygoff, yso + (unsigned)(q * 1024), 16);
                recurrent<H, JM, NQ, false, false>(acc, hrow, wR);
                // the sums / tanh BEHIND the MFMA run, not sprinkled into it: a VALU instruction in an fp32-MFMA gap is never
                // hidden and costs a pipeline turn-around on top (tools/microbench/mfma_f32_overlap.hip)
                __builtin_amdgcn_sched_barrier(0);
                f32x2 ysum;
                if constexpr (NYG > 0) {
                    unsigned spins = 0;
                    while (true) {
                        bool ok = true;
#pragma unroll
                        for (int q = 0; q < NYG; ++q) ok = ok && (vy[q].y == epoch) && (vy[q].w == epoch);
                        if (__all(ok)) break;
                        if (++spins > SPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                            if (lane == 0) { xch_give_up(p.status); sFlag[0] = 1; }      // (seen by every wave behind this step's barrier 2)
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
#pragma unroll
                        for (int q = 0; q < NYG; ++q) vy[q] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ygoff, yso + (unsigned)(q * 1024), 16);
                    }
                    f32x2 P[NYG];
#pragma unroll
                    for (int q = 0; q < NYG; ++q) P[q] = (f32x2){__uint_as_float(vy[q].x), __uint_as_float(vy[q].z)};
#pragma unroll
                    for (int w2 = NYG / 2; w2 >= 1; w2 >>= 1)
#pragma unroll
                        for (int q = 0; q < w2; ++q) P[q] += P[q + w2];
                    ysum = P[0];
                } else {
                    ysum = own_partial((t - 1) & 1);
                }
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) y4[ss] = tanh_f(ysum[ss] + bd4[ss]);
                // y_{t-1} leaves through a buffer store: lane offset computed once per tile, the step in the scalar offset
                if (slice == 0 && wave == 0) {
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y4[ss]), yors, yoff[ss], (unsigned)((t - 1) * p.F_dec * 4), 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                input_proj_reg<true>(acc, y4, kb);
            }
            FOV_STAMP(1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#ifdef FOV_DBG_NOCELL   // timing experiment only (wrong results): no transcendentals in the cell update
                const float ig = acc[0][r], fg = acc[1][r], gg = acc[2][r], og = acc[3][r];
                c[r] = fmaf(fg, c[r], ig * gg);
                hcur[r] = og * c[r];
#else
                const float ig = rec_act<ACT>(acc[0][r]);
                const float fg = rec_act<ACT>(acc[1][r]);
                const float gg = tanh_f(acc[2][r]);
                const float og = rec_act<ACT>(acc[3][r]);
                c[r] = fmaf(fg, c[r], ig * gg);
                hcur[r] = og * tanh_f(c[r]);
#endif
                if (LAYER && p.reserve) {   // training forward: gates and cell state for BPTT
                    const int row = b0 + 4 * g4 + r;
                    if (row < p.B) {
                        float* rp = p.reserve + (((size_t)row * p.T + t) * 5) * H + col0 + n;
                        rp[0] = ig; rp[H] = fg; rp[2 * H] = gg; rp[3 * H] = og; rp[4 * H] = c[r];
                    }
                }
            }
            unsigned xsoff = 0;
            // h_t of a layer's last step is needed by nobody inside the kernel: no publish, no gather
            const bool do_xch = (G > 1) && (!LAYER || t + 1 < steps || F1);
            if (do_xch) {
                // publish this workgroup's slice of h_t: one 8-byte {value, epoch} granule each
                ++epoch;
                xsoff = (epoch & 1u) * (unsigned)(BT * H * sizeof(unsigned long long));
                if (same_xcd) {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4g){__float_as_uint(hcur[2 * r]), epoch, __float_as_uint(hcur[2 * r + 1]), epoch},
                                                               xrs, pub_off + r * H * 16, xsoff, 1 /* sc0: stays in L2 */);
                } else {
#pragma unroll
                    for (int r = 0; r < 2; ++r)
                        __builtin_amdgcn_raw_buffer_store_b128((u32x4g){__float_as_uint(hcur[2 * r]), epoch, __float_as_uint(hcur[2 * r + 1]), epoch},
                                                               xrs, pub_off + r * H * 16, xsoff, 16 /* sc1: write-through */);
                }
            }
            if (LAYER && p.hs) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = b0 + 4 * g4 + r;
                    if (row < p.B) p.hs[((size_t)row * p.T + t) * H + col0 + n] = hcur[r];
                }
            }
            FOV_STAMP(2);
            // (No barrier in front of these stores since round 3.  The own-slice columns of the tile were last read by the
            // own-slice MFMAs of the previous step, which every wave finished before that step's barrier 2; the partner
            // columns, which other waves may still be reading for z_t, are only written behind barrier 1b; the x tile written
            // at the top of this step is one of three and was last read three steps ago.)
            FOV_STAMP(3);
#pragma unroll
            for (int r = 0; r < 4; ++r) sH[(4 * g4 + r) * LDH + wave * 16 + n] = hcur[r];
            __syncthreads();  // barrier 1b: the own slice of h_t is visible to all four waves
            FOV_STAMP(4);
            const bool more = (t + 1 < steps);
            // Pre-activations of step t+1 that need no remote data are computed while the partner
            // slices are in flight: x_{t+1} . K first, then the first sweep of the gather is
            // ISSUED (its ~0.9 us round trip runs under the MFMAs that follow), then the own-slice
            // k-blocks of h_t . R, and only then are the granule tags checked.
            if (more) {
#pragma unroll
                for (int g = 0; g < 4; ++g) acc[g] = ZX ? zr[g] : (f32x4){bias[g], bias[g], bias[g], bias[g]};
                if (LAYER) input_proj_any<true, false>(acc, sX + ((t + 1) % 3) * BT * LDX + n * LDX + 4 * g4, sKw, nq, lane);
            }
            FOV_STAMP(5);
            u32x4g v[NG > 0 ? NG : 1];
            // DECODE has no x . K block between the publish and the gather: one own-slice k-block goes first, so the sweep is
            // not requested right behind the partners' publish (decoder 0.191 -> 0.188 ms over three paired runs; the
            // eight-workgroup kernels, whose stores are sc1, gain far more from the same delay - lstm_wide.hip)
            constexpr int GJ = LAYER ? 0 : 1;
            // (one branch around both MFMA runs: where two `if (more)` met, hipcc copied the accumulators between them)
            if (more) {
                recurrent<H, 0, GJ, true, false>(acc, hrow, wR);
                if (do_xch) {
#pragma unroll
                    for (int j = 0; j < NG; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j], xsoff, 16);
                }
                // (a ZX layer has no x . K run in front: this run opens; with partner slices the next reader of the accumulators is
                // the partner-slice MFMA run: no closing wait states)
                recurrent<H, GJ, 4, LAYER, (G == 1)>(acc, hrow, wR);
            } else if (do_xch) {
#pragma unroll
                for (int j = 0; j < NG; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j], xsoff, 16);
            }
            FOV_STAMP(11);
            if (!LAYER) {
                // this wave's Dense partial over its own 16 units, in the shadow of the gather: the columns it wrote in front of
                // barrier 1b as the B operand h^T of y^T = Wd^T . h^T; the four waves' partials meet in LDS at barrier 2
                const f32x4 hb = *(const f32x4*)(hrow + 16 * wave);
                f32x4 dacc[2];
                dacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
#ifndef FOV_DBG_NODENSE   // (timing experiment only, wrong results: the partial stays zero)
#pragma unroll
                for (int ss = 0; ss < 4; ++ss)
                    mfma_f32<false>(dacc[ss & 1], wd[ss], hb[ss], mf_pos(true, true, ss < 2, ss == 3));
#endif
                *(f32x2*)(sY + (t & 1) * YPAR_LDS + wave * 128 + lane * 2) = (f32x2){dacc[0][0] + dacc[1][0], dacc[0][1] + dacc[1][1]};
            }
            if (do_xch) {
                // complete the gather: sweep again until every tag equals the epoch
                unsigned spins = 0;
#ifdef FOV_DBG_NOGATHER   // timing experiment only (wrong results): take whatever the first sweep returned
                while (false) {
#else
                while (true) {
#endif
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < NG; ++j) ok = ok && (v[j].y == epoch) && (v[j].w == epoch);
                    if (__all(ok)) break;
                    ++spins;
                    if (spins > SPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                        if (lane == 0) {
                            xch_give_up(p.status);
                            sFlag[0] = 1;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");   // the sweep below must really re-read memory
#pragma unroll
                    for (int j = 0; j < NG; ++j) v[j] = __builtin_amdgcn_raw_buffer_load_b128(xrs, goff[j], xsoff, 16);
                }
#ifdef FOV_STAMPS
                if (stamp_on && t < STAMP_STEPS) g_stamps[MODE & 1][t][10] = spins;
#endif
#pragma unroll
                for (int j = 0; j < NG; ++j) {
                    sH[loff[j]] = __uint_as_float(v[j].x);
                    sH[loff[j] + LDH] = __uint_as_float(v[j].z);
                }
            }
            FOV_STAMP(6);
            __syncthreads();  // barrier 2: the whole h_t tile is in LDS
            FOV_STAMP(7);
            if (G > 1 && sFlag[0]) { aborted = true; break; }
            if (!LAYER && G > 1 && wave == 0) {      // the workgroup's Dense partial of y_t, tagged like h_t
                const f32x2 mine = own_partial(t & 1);
                const u32x4g yg = (u32x4g){__float_as_uint(mine[0]), epoch, __float_as_uint(mine[1]), epoch};
                const unsigned yso = (epoch & 1u) * YPAR_BYTES;
                if (same_xcd) __builtin_amdgcn_raw_buffer_store_b128(yg, yrs, ypub_off, yso, 1 /* sc0: stays in L2 */);
                else __builtin_amdgcn_raw_buffer_store_b128(yg, yrs, ypub_off, yso, 16 /* sc1: write-through */);
            }
        }
        if (!LAYER && steps > 0 && !aborted) {
            // y of the last step: the workgroups' partials were published behind the last barrier 2 - one wave of the tile waits for them
            if (slice == 0 && wave == 0) {
                f32x2 ysum;
                bool got = true;
                if constexpr (NYG > 0) {
                    u32x4g vy[NYG];
                    const unsigned yso = (epoch & 1u) * YPAR_BYTES;
                    unsigned spins = 0;
                    while (true) {
#pragma unroll
                        for (int q = 0; q < NYG; ++q) vy[q] = __builtin_amdgcn_raw_buffer_load_b128(yrs, ygoff, yso + (unsigned)(q * 1024), 16);
                        bool ok = true;
#pragma unroll
                        for (int q = 0; q < NYG; ++q) ok = ok && (vy[q].y == epoch) && (vy[q].w == epoch);
                        if (__all(ok)) break;
                        if (++spins > SPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                            if (lane == 0) xch_give_up(p.status);
                            got = false;
                            break;
                        }
                        __builtin_amdgcn_s_sleep(1);
                        asm volatile("" ::: "memory");
                    }
                    f32x2 P[NYG];
#pragma unroll
                    for (int q = 0; q < NYG; ++q) P[q] = (f32x2){__uint_as_float(vy[q].x), __uint_as_float(vy[q].z)};
#pragma unroll
                    for (int w2 = NYG / 2; w2 >= 1; w2 >>= 1)
#pragma unroll
                        for (int q = 0; q < w2; ++q) P[q] += P[q + w2];
                    ysum = P[0];
                } else {
                    ysum = own_partial((steps - 1) & 1);
                }
                if (got) {
#pragma unroll
                    for (int ss = 0; ss < 2; ++ss)
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tanh_f(ysum[ss] + bd4[ss])), yors, yoff[ss],
                                                              (unsigned)((steps - 1) * p.F_dec * 4), 0);
                }
            }
        }
        if constexpr (F1) {
#pragma unroll
            for (int r = 0; r < 4; ++r) { cy.c[r] = c[r]; cy.h[r] = hcur[r]; }
        }
        if (!aborted && !F1) {   // (the encoder phase of the fused kernel hands its state over in registers)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = b0 + 4 * g4 + r;
                if (row < p.B) {
                    if (p.hT) p.hT[(size_t)row * H + col0 + n] = hcur[r];
                    if (p.cT) p.cT[(size_t)row * H + col0 + n] = c[r];
                }
            }
        }
#ifdef FOV_STAMPS
        if (blockIdx.x == 5 && tid == 0) g_stamps[MODE & 1][STAMP_STEPS - 1][1] = __builtin_amdgcn_s_memtime();   // time loop left
#endif
        if constexpr (FUSED != 0) break;   // one tile per group (host-checked)
    }
    if constexpr (F1) {
        cy.epoch = epoch; cy.ticket = ticket; cy.same_xcd = same_xcd; cy.aborted = aborted; cy.xch_used = xch_used;
        __syncthreads();   // every wave is done with this phase's LDS regions before the decoder phase lays out its own
    } else {
        if (xch_used) xch_settle(p.status, ticket, (unsigned)p.epoch_span);
    }
}

template <int H, int ACT, int MODE>
__global__ __launch_bounds__(256, 1) void lstm_cluster_kernel(LstmParams p) {
    if constexpr (H >= 128) {
        // padded grid (see the fused kernel below): a block of an absent group counts as arrived - if this launch exchanges at
        // all (cluster_body's xch_used) - and leaves
        if (p.xcd_pad && (int)((blockIdx.x / (8 * (H / 64))) * 8 + (blockIdx.x & 7)) >= p.num_groups) {
            const bool exchanging = (MODE == MODE_DECODE) || p.T > 1;
            if (threadIdx.x == 0 && exchanging) xch_count_arrival(p.status);
            return;
        }
    }
    ClusterCarry cy;
    cluster_body<H, ACT, MODE, 0>(p, cy);
}

// Encoder + autoregressive decoder of the fused call in ONE launch (one 16-sequence tile per group): no second dispatch,
// no second header read / handshake, the state stays in registers and the h_T tile in LDS; only the weights are swapped.
template <int H, int ACT>
__global__ __launch_bounds__(256, 1) void lstm_cluster_fused_kernel(LstmParams p) {
    if constexpr (H >= 128) {
        // a group count that is no multiple of eight (the reference's batch of 32 = two groups; any batch that is no multiple of
        // 128 sequences): the grid is padded to the next multiple so that a group's members sit 8 blocks apart = on one XCD
        // (lstm_wide16.hip, xch_common.h: xch_padded_groups); the blocks of the absent groups count as arrived and leave
        if (p.xcd_pad && (int)((blockIdx.x / (8 * (H / 64))) * 8 + (blockIdx.x & 7)) >= p.num_groups) {
            if (threadIdx.x == 0) xch_count_arrival(p.status);
            return;
        }
    }
    ClusterCarry cy;
    cluster_body<H, ACT, MODE_LAYER, 1>(p, cy);
    cluster_body<H, ACT, MODE_DECODE, 2>(p, cy);
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
// CUs of the CURRENT device (cached per device ordinal: a process may drive several GPUs)
int device_cu_count() {
    static std::atomic<int> cus[64];    // (threads that race on the first use store the same value)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    const int lim = env_knobs().resident_limit;   // FOV_DBG_RESIDENT_LIMIT: pretend fewer CUs are available (tests of the fallbacks)
    return (lim > 0 && lim < n) ? lim : n;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) once per (device, kernel, size) instead of on every launch
int ensure_dynamic_lds(const void* kern, size_t lds, int block) {
    struct Seen { const void* k; size_t lds; int dev; };
    static Seen seen[256];
    static int n_seen = 0;
    static std::mutex mu;
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lock(mu);
    for (int i = 0; i < n_seen; ++i)
        if (seen[i].k == kern && seen[i].dev == dev && seen[i].lds >= lds) return FOV_OK;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    // the persistent grids are sized as ONE workgroup per CU: ask the runtime once per (device, kernel, LDS size) that a
    // workgroup of this shape is admitted at all (registers, LDS, waves) instead of assuming it from the CU count
    int per_cu = 0;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, block, lds);
    if (e != hipSuccess || per_cu < 1) {
        (void)hipGetLastError();
        set_error("persistent kernel: a %d-thread workgroup with %zu B of LDS is not resident on this device (occupancy %d)", block, lds, per_cu);
        return FOV_ERR_UNSUPPORTED;
    }
    if (n_seen < 256) seen[n_seen++] = Seen{kern, lds, dev};
    return FOV_OK;
}

bool cluster_shape_ok(int F, int H) {
    return (H == 64 || H == 128 || H == 256) && F >= 1 && F <= CL_MAX_F;
}

int cluster_num_groups(int B, int H) {
    const int G = H / 64;
    const int tiles = (B + BT - 1) / BT;
    const int groups = device_cu_count() / G;
    if (groups < 1) return 0;   // not even one group is co-resident: the caller reports FOV_ERR_UNSUPPORTED (explicit impl) or falls back (auto)
    return tiles < groups ? (tiles > 0 ? tiles : 1) : groups;
}

static size_t cluster_xch_bytes(int B, int H) {
    const size_t groups = (size_t)cluster_num_groups(B, H);
    const size_t b = (groups * 2 * BT * H + groups * (H / 64) + groups * 2 * (H / 64) * 128) * sizeof(unsigned long long);   // granules + hello + (decode) the workgroups' Dense partials
    return (b + 255) & ~(size_t)255;
}

// header + the fixed granule area + (fused decode) the encoder's final (h, c)
size_t cluster_workspace_bytes(int B, int H) {
    (void)cluster_xch_bytes;
    return kStatusBytes + kXchBytes + (size_t)2 * B * H * sizeof(float);
}

template <int H>
static int launch_cluster_h(const LstmParams& p, int mode, hipStream_t stream) {
    void (*kern)(LstmParams) = nullptr;
    if (mode == MODE_LAYER && p.zx) mode = MODE_LAYER_ZX;
    if (p.act == FOV_ACT_HARD_SIGMOID)
        kern = mode == MODE_DECODE ? lstm_cluster_kernel<H, FOV_ACT_HARD_SIGMOID, MODE_DECODE>
               : mode == MODE_LAYER_ZX ? lstm_cluster_kernel<H, FOV_ACT_HARD_SIGMOID, MODE_LAYER_ZX>
                                       : lstm_cluster_kernel<H, FOV_ACT_HARD_SIGMOID, MODE_LAYER>;
    else
        kern = mode == MODE_DECODE ? lstm_cluster_kernel<H, FOV_ACT_SIGMOID, MODE_DECODE>
               : mode == MODE_LAYER_ZX ? lstm_cluster_kernel<H, FOV_ACT_SIGMOID, MODE_LAYER_ZX>
                                       : lstm_cluster_kernel<H, FOV_ACT_SIGMOID, MODE_LAYER>;
    const int F = mode == MODE_DECODE ? p.F_dec : (p.zx ? 0 : p.F);
    const ClusterLds L = cluster_lds(H, F, mode == MODE_DECODE);
    const size_t lds = (size_t)L.total_floats * sizeof(float);
    if (lds > 160 * 1024) {
        set_error("cluster kernel: H=%d F=%d needs %zu B of LDS (> 160 KiB)", H, F, lds);
        return FOV_ERR_UNSUPPORTED;
    }
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    LstmParams q = p;
    const int padded = xch_padded_groups(q.num_groups);
    q.xcd_pad = (H >= 128 && !env_knobs().no_xcd_pad && (q.num_groups & 7) != 0 && device_cu_count() >= padded * (H / 64)) ? 1 : 0;
    const dim3 grid((q.xcd_pad ? padded : q.num_groups) * (H / 64)), block(256);
    hipLaunchKernelGGL(kern, grid, block, lds, stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("cluster launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

template <int H>
static int launch_cluster_fused_h(const LstmParams& p, hipStream_t stream) {
    void (*kern)(LstmParams) = p.act == FOV_ACT_HARD_SIGMOID ? lstm_cluster_fused_kernel<H, FOV_ACT_HARD_SIGMOID>
                                                             : lstm_cluster_fused_kernel<H, FOV_ACT_SIGMOID>;
    const ClusterLds L1 = cluster_lds(H, p.F, false), L2 = cluster_lds(H, p.F_dec, true);
    const size_t lds = (size_t)(L1.total_floats > L2.total_floats ? L1.total_floats : L2.total_floats) * sizeof(float);
    if (lds > 160 * 1024) { set_error("fused cluster kernel: needs %zu B of LDS (> 160 KiB)", lds); return FOV_ERR_UNSUPPORTED; }
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    LstmParams q = p;
    const bool no_pad = env_knobs().no_xcd_pad != 0;
    const int padded = (q.num_groups + 7) & ~7;
    q.xcd_pad = (H >= 128 && !no_pad && (q.num_groups & 7) != 0 && device_cu_count() >= padded * (H / 64)) ? 1 : 0;
    hipLaunchKernelGGL(kern, dim3((q.xcd_pad ? padded : q.num_groups) * (H / 64)), dim3(256), lds, stream, q);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("fused cluster launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

static int launch_cluster_fused(const LstmParams& p, hipStream_t stream) {
    switch (p.H) {
        case 64: return launch_cluster_fused_h<64>(p, stream);
        case 128: return launch_cluster_fused_h<128>(p, stream);
        default: return launch_cluster_fused_h<256>(p, stream);
    }
}

static int launch_cluster_mode(const LstmParams& p, int mode, hipStream_t stream) {
    switch (p.H) {
        case 64: return launch_cluster_h<64>(p, mode, stream);
        case 128: return launch_cluster_h<128>(p, mode, stream);
        default: return launch_cluster_h<256>(p, mode, stream);
    }
}

// decode == false: one LSTM layer over x.  decode == true: encoder launch (final state into the
// workspace) followed by the autoregressive decoder launch seeded from it.
int launch_cluster(const LstmParams& p_in, bool decode, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0) return FOV_OK;
    if (p.zx) p.F = 1;   // input projection supplied by the caller: F only sizes (empty) LDS regions
    if (!cluster_shape_ok(p.F, p.H) || (decode && (p.F_dec < 1 || p.F_dec > CL_MAX_O))) {
        set_error("cluster kernel supports H in {64,128,256}, 1<=F<=%d, F_dec<=%d (got H=%d F=%d F_dec=%d)",
                  CL_MAX_F, CL_MAX_O, p.H, p.F, p.F_dec);
        return FOV_ERR_UNSUPPORTED;
    }
    p.num_tiles = (p.B + BT - 1) / BT;
    p.num_groups = cluster_num_groups(p.B, p.H);
    if (p.num_groups < 1) { set_error("cluster kernel: fewer than %d CUs available for one group", p.H / 64); return FOV_ERR_UNSUPPORTED; }
    p.force_safe_exchange = env_knobs().force_safe_exchange;   // cached (fov_reload_env re-reads the environment)
    if (cluster_xch_bytes(p.B, p.H) > kXchBytes - kHelloBytes) { set_error("cluster kernel: granule area exceeds the workspace's"); return FOV_ERR_WORKSPACE; }
    // No memset: epoch tags continue from the workspace header.  A group visits ceil(tiles / groups) tiles and
    // advances its epoch once per step of each.
    const int visits = (p.num_tiles + p.num_groups - 1) / p.num_groups;
    if (!decode) {
        p.epoch_span = p.T * visits + 1;
        if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
        return launch_cluster_mode(p, MODE_LAYER, stream);
    }

    if (visits == 1 && p.T > 0 && p.T_out > 0 && !p.hs && !p.reserve && !env_knobs().two_launches) {
        // one tile per group: encoder and decoder as ONE launch (state in registers, h_T tile in LDS)
        LstmParams f = p;
        f.epoch_span = p.T + p.T_out + 2;
        if (int rc_ = xch_account(p.status, f.epoch_span, stream)) return rc_;
        return launch_cluster_fused(f, stream);
    }
    LstmParams enc = p;
    float* state = (float*)((char*)p.status + kStatusBytes + kXchBytes);
    enc.hT = state;
    enc.cT = state + (size_t)p.B * p.H;
    enc.hs = nullptr;
    enc.epoch_span = p.T * visits + 1;
    if (int rc_ = xch_account(p.status, enc.epoch_span + p.T_out * visits + 1, stream)) return rc_;
    int rc = launch_cluster_mode(enc, MODE_LAYER, stream);
    if (rc) return rc;
    // the decoder launch reads the base the encoder launch left behind: its tags (and its hello tag) are larger
    // than every stale one
    LstmParams dec = p;
    dec.h0 = enc.hT;
    dec.c0 = enc.cT;
    dec.epoch_span = p.T_out * visits + 1;
    return launch_cluster_mode(dec, MODE_DECODE, stream);
}

// Decoder alone: T_out autoregressive steps from a given state (p.h0, p.c0) - the sampling loop of
// FoV_seq2seq.py:154-178 after encoder_model.predict, in one launch.
int launch_cluster_decoder(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0 || p.T_out == 0) return FOV_OK;
    if (!(p.H == 64 || p.H == 128 || p.H == 256) || p.F_dec < 1 || p.F_dec > CL_MAX_O) {
        set_error("cluster decoder supports H in {64,128,256}, F_dec<=%d (got H=%d F_dec=%d)", CL_MAX_O, p.H, p.F_dec);
        return FOV_ERR_UNSUPPORTED;
    }
    p.num_tiles = (p.B + BT - 1) / BT;
    p.num_groups = cluster_num_groups(p.B, p.H);
    if (p.num_groups < 1) { set_error("cluster decoder: fewer than %d CUs available for one group", p.H / 64); return FOV_ERR_UNSUPPORTED; }
    p.force_safe_exchange = env_knobs().force_safe_exchange;
    p.epoch_span = p.T_out * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    return launch_cluster_mode(p, MODE_DECODE, stream);
}

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 2 * STAMP_STEPS * STAMP_SLOTS);
}
#endif

}  // namespace fov
