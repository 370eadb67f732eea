// Resident fp32 weight fragments of the eight-workgroup kernels (lstm_wide.hip, mix_decoder.hip) through LDS - the fp32
// twin of stage_weight_sets in bf16_common.h (round 4; the reasoning and the measurements are in the comment there).
//
// Those kernels keep, per lane (n, g4) of v_mfma_f32_16x16x4_f32, w[j][s][t] = W[k = 16 j + 4 g4 + s][column t of the lane]
// with the lane's columns col0 = hi * H + unit, col1 = (2 + hi) * H + unit, unit = 32 slice + 8 wave + (n & 7), hi = n >> 3.
// Asked for directly that is one dword per lane and load: 8 cache lines per wave instruction for 256 bytes, every line of
// the workgroup's slice requested by all four waves, 256 loads per thread for two (256, 1024) matrices.
// Here the workgroup reads its slice - 4 gates x 32 units = four 128-byte lines per matrix row - as dwordx4 loads, 8 lanes
// to a line, writes the values column-major into LDS (36 words per column: 32 rows of a stage + 4; the row index is
// XOR-swizzled with bit 3 of the column so that the 64 lanes of a ds_write_b32 spread over all banks) and every lane picks
// its fragments up as ds_read_b128: w[j][0..3][t] are four consecutive words of column t.
// A stage is 32 rows (4 loads = 16 registers per thread); DEPTH stages are requested ahead of the one being written.
// The matrices need 4-byte alignment only (a parameter view inside a trainer's flat buffer): a dwordx4 load from such an
// address returns the right four dwords on gfx950 (tools/microbench/unaligned_b128.hip; tests: ..._at_any_four_byte_offset_...).
#pragma once
#include "fov_common.h"

namespace fov {

constexpr int FST_WORDS = 36;                 // words per column of a staging buffer (16-byte aligned, 4 mod 32)
constexpr int FST_BUF = 128 * FST_WORDS;      // one buffer: the workgroup's 128 gate columns x 32 rows
constexpr int FST_LDS_WORDS = 2 * FST_BUF;    // two buffers: 36 864 bytes
constexpr int FST_DEPTH = 3;

typedef unsigned fst_u32x4 __attribute__((ext_vector_type(4)));

struct FStageLane {
    unsigned goff;      // byte offset of (row kr, this thread's 4 columns) in the matrix
    unsigned wr_even;   // LDS word of the thread's first column for even / odd 8-row groups of a stage (swizzle folded in)
    unsigned wr_odd;
    unsigned rd[2];     // LDS word of fragment (16-row block 0 of the stage, column t)
};
template <int H>
__device__ __forceinline__ FStageLane f_stage_lane(int slice) {
    const int tid = threadIdx.x;
    const int c = tid & 7, kr = (tid >> 3) & 7, gate = tid >> 6;
    const int lane = tid & 63, wave = tid >> 6, n = lane & 15, g4 = lane >> 4, hi = n >> 3;
    FStageLane q;
    q.goff = (unsigned)(kr * 4 * H + gate * H + 32 * slice + 4 * c) * 4u;
    const int sbw = (c >> 1) & 1;                    // swizzle bit of the columns this thread writes (column bit 3)
    const unsigned wbase = (unsigned)((32 * gate + 4 * c) * FST_WORDS + kr);
    q.wr_even = wbase + 8u * sbw;                    // 8-row group it: word 8 * (it ^ sb) + kr
    q.wr_odd = wbase - 8u * sbw;
    const int sbr = wave & 1;                        // ... of the columns it reads: 8 * wave + (n & 7)
#pragma unroll
    for (int t = 0; t < 2; ++t)
        q.rd[t] = (unsigned)(((2 * t + hi) * 32 + 8 * wave + (n & 7)) * FST_WORDS + 4 * (g4 ^ (2 * sbr)));
    return q;
}
// requests of stage `ls` (rows [32 ls, 32 ls + 32)) of W; rows >= nrows read as zero (whole offset in the vector register:
// the hardware's range check does not see a scalar offset)
template <int H>
__device__ __forceinline__ void f_stage_issue(fst_u32x4 (&r)[4], const float* __restrict__ W, int nrows, int ls, const FStageLane& q) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, nrows * 4 * H * 4, 0x00020000);
#pragma unroll
    for (int it = 0; it < 4; ++it)
        r[it] = __builtin_amdgcn_raw_buffer_load_b128(rs, q.goff + (unsigned)((32 * ls + 8 * it) * 4 * H) * 4u, 0, 0);
}
__device__ __forceinline__ void f_stage_write(const fst_u32x4 (&r)[4], unsigned* buf, const FStageLane& q) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        unsigned* wp = buf + ((it & 1) ? q.wr_odd : q.wr_even) + 8 * it;
#pragma unroll
        for (int j = 0; j < 4; ++j) wp[j * FST_WORDS] = r[it][j];
    }
}
template <int NJ>
__device__ __forceinline__ void f_stage_read(float (&w)[NJ][4][2], int ls, const unsigned* buf, const FStageLane& q) {
#pragma unroll
    for (int jl = 0; jl < 2; ++jl)
        if (2 * ls + jl < NJ) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const fst_u32x4 v = *(const fst_u32x4*)(buf + q.rd[t] + 16 * jl);
#pragma unroll
                for (int s = 0; s < 4; ++s) w[2 * ls + jl][s][t] = __uint_as_float(v[s]);
            }
        }
}
// One or two fragment sets in one pipeline (every thread of the 256 calls it; sStage: FST_LDS_WORDS words, 16-byte aligned,
// free again at return, after a barrier).  `behind_first_stage()` runs once behind the first stage's requests.
// H: hidden width (row length 4 H; a workgroup owns 32 units of each gate).
template <int H, bool HAS_A, int DEPTH, int NA, int NB, class Behind>
__device__ __forceinline__ void stage_weight_sets_f32_impl(float (&a)[NA][4][2], const float* __restrict__ Wa, int nra, float (&b)[NB][4][2],
                                                           const float* __restrict__ Wb, int nrb, int slice, unsigned* sStage,
                                                           Behind behind_first_stage) {
    constexpr int SA = HAS_A ? (NA + 1) / 2 : 0, SB = (NB + 1) / 2, NS = SA + SB;
    const FStageLane q = f_stage_lane<H>(slice);
    fst_u32x4 r[DEPTH + 1][4];
    auto issue = [&](int s) __attribute__((always_inline)) {
        if (s < SA) f_stage_issue<H>(r[s % (DEPTH + 1)], Wa, nra, s, q);
        else f_stage_issue<H>(r[s % (DEPTH + 1)], Wb, nrb, s - SA, q);
    };
    issue(0);
    behind_first_stage();
#pragma unroll
    for (int s = 1; s < DEPTH; ++s)
        if (s < NS) issue(s);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + DEPTH < NS) issue(s + DEPTH);
        unsigned* buf = sStage + (s & 1) * FST_BUF;
        f_stage_write(r[s % (DEPTH + 1)], buf, q);
        __syncthreads();   // (buffer s & 1 is written again two stages on, behind the barrier of stage s + 1, which a wave passes after these reads)
        if (s < SA) f_stage_read<NA>(a, s, buf, q);
        else f_stage_read<NB>(b, s - SA, buf, q);
    }
    __syncthreads();
}
template <int H, int DEPTH = FST_DEPTH, int NA, int NB, class Behind>
__device__ __forceinline__ void stage_weight_sets_f32(float (&a)[NA][4][2], const float* __restrict__ Wa, int nra, float (&b)[NB][4][2],
                                                      const float* __restrict__ Wb, int nrb, int slice, unsigned* sStage,
                                                      Behind behind_first_stage) {
    stage_weight_sets_f32_impl<H, true, DEPTH>(a, Wa, nra, b, Wb, nrb, slice, sStage, behind_first_stage);
}
template <int H, int DEPTH = FST_DEPTH, int NB, class Behind>
__device__ __forceinline__ void stage_weight_set_f32(float (&b)[NB][4][2], const float* __restrict__ Wb, int nrb, int slice, unsigned* sStage,
                                                     Behind behind_first_stage) {
    stage_weight_sets_f32_impl<H, false, DEPTH>(b, Wb, nrb, b, Wb, nrb, slice, sStage, behind_first_stage);
}

}  // namespace fov
