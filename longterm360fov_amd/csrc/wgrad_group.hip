// Weight gradients of LSTM layers with FEW rows (batch x time <= ~1000) as ONE launch, fp32 (round 4).
//
// model.fit's backward forms, per layer, dK = x^T dz, dR = h_{t-1}^T dz (h_{-1} = h0) and db = column sums of dz
// (mycode/lstm.py:556-567 under its train_op; FoV_seq2seq.py:103,112-117 at the reference's batch of 32).  At 320 rows the
// split-K GEMM of train_kernels.hip spent 64 us in seven launches on the two layers of lstm.py (two fused products, two
// h0^T dz_0 products, one skinny product, reduces): every launch is a ~5 us stub around a few microseconds of MFMAs.
// Here every (problem, 128 x 128 output tile) is one workgroup that walks ALL rows - no split, no partials, no reduce, a fixed
// summation order - and the problems of a call share the launch:
//   problem = { A (rows x M, row (ro, ri) at a + ro*a_so + (ri - shift)*lda; shift = 1 reads h_{t-1}: row ri = 0 comes from
//               h0 + ro*ldh0 or is zero), B = dz (rows x N), C (M x N) (+)=, optional bias row: db (N) (+)= column sums of B }
// Tile: 4 waves as 2 x 2, a wave owns 64 x 64 = 4 x 4 MFMA tiles of v_mfma_f32_16x16x4_f32; 16 rows (k) per stage, both
// operands are k-slow and go into LDS as they lie ([k][128 + 16] floats: the four k rows of an MFMA step land on disjoint
// banks), double-buffered, loads one stage ahead of the 64 MFMAs of a stage (2 048 cycles: the kernel is MFMA-bound from the
// second stage on).  Row cursors advance by additions (gemm_bf16.hip, tn_row_split).  The result tile leaves through LDS as
// 16-byte stores.
#include <stdlib.h>

#include "fov_common.h"

namespace fov {

namespace {

typedef unsigned wu32x4 __attribute__((ext_vector_type(4)));

constexpr int WT = 128;          // output tile
constexpr int WKB = 16;          // rows per stage
constexpr int WLD = WT + 16;     // floats per LDS row of an operand image: 144 = 16 (mod 64), the four k rows of an MFMA step on disjoint banks
constexpr int WCLD = 68;         // floats per row of a wave's 64 x 64 result tile in LDS (epilogue)
constexpr int kMaxProb = 8;

struct WgProb {
    const float* a;      // rows x M
    const float* h0;     // RO x M or NULL (shift only)
    const float* b;      // rows x N
    float* c;            // M x N, row stride ldc
    float* bias;         // N or NULL: column sums of b
    long lda, a_so, ldh0, ldb, b_so;
    int M, N, ldc, shift;
    int block0;          // first block of this problem; tiles are n-fastest
    int grid_n;
};

struct WgBatch {
    WgProb p[kMaxProb];
    int count, RO, RI, accumulate;
    unsigned ri_magic;   // floor(2^32 / RI)
};

__device__ __forceinline__ void wg_mfma(f32x4& acc, float a, float b) {
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
}

#ifdef FOV_STAMPS
__device__ unsigned long long g_wr_trace[4096][6];      // per workgroup of wgrad_rows_kernel: entry, MFMAs done, exit, {xcc, hw id}
__device__ unsigned long long g_wg_stamps[4][8];
#define WG_STAMP(slot) do { if (stamp_slot >= 0 && threadIdx.x == 0) g_wg_stamps[stamp_slot][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define WG_STAMP(slot) do { } while (0)
#endif

constexpr size_t WG_IMG = (size_t)2 * 3 * WKB * WLD * sizeof(float), WG_TILES = (size_t)4 * 64 * WCLD * sizeof(float);
constexpr size_t WG_LDS = WG_IMG > WG_TILES ? WG_IMG : WG_TILES;

// AVEC (block-uniform, chosen per problem): 8-byte loads of A rows (else two 4-byte loads).  A template parameter, not a run-time `if` around the loads:
// loads inside a branch meet at a merge where the compiler waits for them - the next stage's loads would stall the MFMAs.
template <bool AVEC, bool H0>   // H0: the shifted operand's first row of every sequence comes from h0 (a second, mutually exclusive load)
__device__ __forceinline__ void wgrad_group_body(const WgBatch& g, const int pi, unsigned char* sRaw, float (*sBias)[WT]) {
    float* sA = reinterpret_cast<float*>(sRaw);            // [3][WKB][WLD]
    float* sB = sA + 3 * WKB * WLD;                        // [3][WKB][WLD]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
#ifdef FOV_STAMPS
    const int stamp_slot = blockIdx.x == 0 ? 0 : blockIdx.x == 70 ? 1 : blockIdx.x == 150 ? 2 : blockIdx.x == gridDim.x - 1 ? 3 : -1;
#endif
    WG_STAMP(0);
    const WgProb& q = g.p[pi];
    const int tile = (int)blockIdx.x - q.block0;
    const int by = tile / q.grid_n, bx = tile - by * q.grid_n;
    const int m0 = by * WT, n0 = bx * WT;
    const unsigned rows = (unsigned)g.RO * (unsigned)g.RI, RI = (unsigned)g.RI;
    const bool bias_blk = q.bias != nullptr && by == 0;
    // Staging: a WAVE moves one whole row per instruction (lane = two columns, 8-byte loads): rows wave, wave + 4, wave + 8,
    // wave + 12 of a stage.  Everything about a row - (ro, ri), the h_{-1} case, its byte offsets - is then wave-uniform and
    // lives in scalar registers and instructions; a lane contributes its constant column offset.  (fp32 MFMAs share the issue
    // port with the VALU: with per-lane row cursors, ~100 VALU instructions per stage, a stage took 4 000 cycles for 2 048
    // cycles of MFMAs.)
    const int scol = 2 * lane;
    constexpr unsigned OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.a), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.h0 ? q.h0 : q.a), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(q.b), 0, 0x7fffffff, 0x00020000);
    const unsigned a_so4 = (unsigned)q.a_so * 4u, a_ld4 = (unsigned)q.lda * 4u, b_so4 = (unsigned)q.b_so * 4u, b_ld4 = (unsigned)q.ldb * 4u,
                   h_ld4 = (unsigned)q.ldh0 * 4u;
    // lane part of the offsets; a column past the matrix: out of range for good
    const unsigned a_col0 = (m0 + scol < q.M) ? (unsigned)(m0 + scol) * 4u : OOR;
    const unsigned a_col1 = (m0 + scol + 1 < q.M) ? (unsigned)(m0 + scol + 1) * 4u : OOR;
    const unsigned b_col = (n0 + scol + 1 < q.N) ? (unsigned)(n0 + scol) * 4u : OOR;      // N % 4 == 0
    typedef float wf32x2 __attribute__((ext_vector_type(2)));
    typedef unsigned wu32x2 __attribute__((ext_vector_type(2)));
    struct Regs { wf32x2 a[4], b[4]; };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};   // [0..1]: this lane's two columns
    unsigned next_row = (unsigned)wave;   // wave-uniform: the stage's first row of this wave
    auto load_stage = [&](Regs& R) {   // the NEXT 16 rows (calls are in stage order); every load unconditional, out of range reads as 0
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned r = next_row + 4u * i;                      // scalar
            unsigned ro = __umulhi(r, g.ri_magic);
            unsigned ri = r - ro * RI;
            if (ri >= RI) { ++ro; ri -= RI; }
            const bool rok = r < rows;
            const bool first = q.shift && ri == 0;                     // h_{-1}: from h0 (or zero)
            // the row's tape source is off for h_{-1}; with H0 a second load fetches that row from h0 - exactly one of the two is
            // in range, the sum is the value (no branch around a load: 4.13)
            const unsigned arow = ro * a_so4 + (ri - (unsigned)q.shift) * a_ld4;
            const bool a_on = rok && !first;
            const bool h_on = H0 && rok && first;
            const unsigned hrow = ro * h_ld4;
            const unsigned brow = ro * b_so4 + ri * b_ld4;
            if constexpr (AVEC) {
                const wu32x2 t = __builtin_amdgcn_raw_buffer_load_b64(ars, a_on ? a_col0 : OOR, arow, 0);
                R.a[i] = (wf32x2){__uint_as_float(t[0]), __uint_as_float(t[1])};
                if constexpr (H0) {
                    const wu32x2 th = __builtin_amdgcn_raw_buffer_load_b64(hrs, h_on ? a_col0 : OOR, hrow, 0);
                    R.a[i][0] += __uint_as_float(th[0]);
                    R.a[i][1] += __uint_as_float(th[1]);
                }
            } else {
                const unsigned t0 = __builtin_amdgcn_raw_buffer_load_b32(ars, a_on ? a_col0 : OOR, arow, 0);
                const unsigned t1 = __builtin_amdgcn_raw_buffer_load_b32(ars, a_on ? a_col1 : OOR, arow, 0);
                R.a[i] = (wf32x2){__uint_as_float(t0), __uint_as_float(t1)};
                if constexpr (H0) {
                    R.a[i][0] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(hrs, h_on ? a_col0 : OOR, hrow, 0));
                    R.a[i][1] += __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(hrs, h_on ? a_col1 : OOR, hrow, 0));
                }
            }
            const wu32x2 tb = __builtin_amdgcn_raw_buffer_load_b64(brs, rok ? b_col : OOR, brow, 0);
            R.b[i] = (wf32x2){__uint_as_float(tb[0]), __uint_as_float(tb[1])};
        }
        next_row += WKB;
    };
    auto store_stage = [&](int buf, const Regs& R) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(wf32x2*)(sA + (buf * WKB + wave + 4 * i) * WLD + scol) = R.a[i];
            *(wf32x2*)(sB + (buf * WKB + wave + 4 * i) * WLD + scol) = R.b[i];
        }
        if (bias_blk) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { bsum[0] += R.b[i][0]; bsum[1] += R.b[i][1]; }
        }
    };
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // THREE stages of loads in flight (three register sets, three LDS images, a branch-free loop over a stage count rounded up to
    // a multiple of three - the structure of gemm_bf16_tn3_kernel): with one stage ahead a stage cost 4 400 cycles, the 2 048
    // cycles of its MFMAs plus what was left of a ~2 us memory latency at one workgroup per CU.
    const int nst = (int)((rows + WKB - 1) / WKB);
    const int nstages = (nst + 2) / 3 * 3;
    Regs R0, R1, R2;
    load_stage(R0);
    load_stage(R1);
    load_stage(R2);
    WG_STAMP(1);
    store_stage(0, R0);
    load_stage(R0);
    __syncthreads();
    WG_STAMP(2);
    auto iter = [&](int buf, int nbuf, Regs& Rn) {   // multiply the stage in image buf; the next stage (set Rn) -> image nbuf
        const float* ab = sA + buf * WKB * WLD + wm * 64 + li;
        const float* bb = sB + buf * WKB * WLD + wn * 64 + li;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            float af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = ab[(4 * ks + lq) * WLD + 16 * i];
                bf[i] = bb[(4 * ks + lq) * WLD + 16 * i];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) wg_mfma(acc[i][j], af[i], bf[j]);
        }
        store_stage(nbuf, Rn);
        load_stage(Rn);
        __syncthreads();
    };
    for (int s = 0; s < nstages; s += 3) {
        iter(0, 1, R1);
        iter(1, 2, R2);
        iter(2, 0, R0);
    }
    WG_STAMP(3);
    // ---- bias row: the four waves' column sums, folded in a fixed order ----
    if (bias_blk) {
        sBias[wave][scol] = bsum[0];
        sBias[wave][scol + 1] = bsum[1];
        __syncthreads();
        if (tid < WT && n0 + tid < q.N) {
            const float t = (sBias[0][tid] + sBias[1][tid]) + (sBias[2][tid] + sBias[3][tid]);
            q.bias[n0 + tid] = g.accumulate ? q.bias[n0 + tid] + t : t;
        }
    }
    // ---- result tile through LDS (the loop ended on a barrier): 16-byte stores of four consecutive columns ----
    {
        float* tl = reinterpret_cast<float*>(sRaw) + wave * (64 * WCLD);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) tl[(16 * i + 4 * lq + r) * WCLD + 16 * j + li] = acc[i][j][r];
        // (a wave reads back only what it wrote; its LDS operations complete in order)
        const int rr = lane >> 4, c4 = (lane & 15) * 4;
        const bool vec_ok = (q.ldc & 3) == 0 && (((uintptr_t)q.c) & 15) == 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int rl = rr + 4 * k;
            const f32x4 v = *(const f32x4*)(tl + rl * WCLD + c4);
            const int row = m0 + wm * 64 + rl, col = n0 + wn * 64 + c4;
            if (row < q.M) {
                float* cp = q.c + (size_t)row * q.ldc + col;
                if (vec_ok && col + 3 < q.N) {
                    f32x4 o = v;
                    if (g.accumulate) o += *(const f32x4*)cp;
                    *(f32x4*)cp = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < q.N) cp[e] = g.accumulate ? cp[e] + v[e] : v[e];
                }
            }
        }
    }
#ifdef FOV_STAMPS
    if (stamp_slot >= 0 && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_wg_stamps[stamp_slot][4] = __builtin_amdgcn_s_memtime();
        g_wg_stamps[stamp_slot][5] = (unsigned long long)nstages;
    }
#endif
}

__global__ __launch_bounds__(256) void wgrad_group_kernel(WgBatch g) {
    __shared__ __attribute__((aligned(16))) unsigned char sRaw[WG_LDS];
    __shared__ float sBias[4][WT];
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.count; ++i)
        if ((int)blockIdx.x >= g.p[i].block0) pi = i;
    const WgProb& q = g.p[pi];
    const bool avec = ((q.lda | q.a_so | q.ldh0) & 1) == 0 && (q.M & 1) == 0 && ((((uintptr_t)q.a) | ((uintptr_t)q.h0)) & 7) == 0;   // 8-byte loads of A
    const bool h0 = q.shift && q.h0 != nullptr;
    if (avec && h0) wgrad_group_body<true, true>(g, pi, sRaw, sBias);
    else if (avec) wgrad_group_body<true, false>(g, pi, sRaw, sBias);
    else if (h0) wgrad_group_body<false, true>(g, pi, sRaw, sBias);
    else wgrad_group_body<false, false>(g, pi, sRaw, sBias);
}


// ---- few rows AND few outputs (round 5): the reference's own batch at its widths (32 sequences x 10 steps, H <= 256) -------------
// At 320 rows a 128 x 128 tile per workgroup leaves a dozen workgroups walking 20 stages each (36 us), and the split products of
// train_kernels.hip cost seven launches (fused product, h0^T dz_0, dK, two batch reduces: 45 us of config 1's 160 us step).  Here a
// workgroup owns a 16 x 64 output tile and its FOUR WAVES SPLIT THE ROWS: a wave issues all the loads of its quarter at once (one A
// value and four consecutive dz values per lane and MFMA step, straight into the MFMA operand registers - no LDS staging, one
// memory round trip), multiplies, and the four partial tiles are added through LDS in a fixed order.  Problems of a call (both
// layers' dR + db and dK, each with its own time length) share the launch; nothing is split across workgroups, so there are no
// partials in memory and no reduce launch.  The four MFMA tiles of a wave take the columns n0 + 4 li + j (j = tile): a lane's four
// dz values are one 16-byte load and its four results per row one 16-byte store.
constexpr int WR_TN = 64;
constexpr int WR_MAXROWS = 640;      // a multiple of every 32 NB in use
constexpr int WR_ZCOLS = 2048;       // widest A operand
__device__ float g_wr_zero[WR_ZCOLS];      // the row of zeros
typedef __attribute__((address_space(1))) float wr_gfloat;

struct WrProb {
    const float* a;      // rows x M
    const float* h0;     // RO x M or NULL (shift only)
    const float* b;      // rows x N
    float* c;            // M x N, row stride ldc
    float* bias;         // N or NULL
    int lda, a_so, ldh0, ldb, b_so;
    int M, N, ldc, shift, RO, RI;
    unsigned ri_magic;   // floor(2^32 / RI)
    int block0, grid_n;
};

struct WrBatch {
    WrProb p[kMaxProb];
    int count, accumulate;
};

// MT: MFMA tiles along M per workgroup (tile = 16 MT x 64; MT = 4 for wide layers: a quarter of the dz traffic per MFMA, the launch
// is then MFMA-bound).  NB: MFMA steps (4 rows each) per wave and batch of loads: 4 waves x NB x 4 rows per round trip.
// The accumulators stay where they are for the whole kernel ("+a"): left to the register allocator, the two-batch loop body
// permuted all 16 MT accumulator registers through VGPRs once per iteration (128 v_accvgpr moves on the MFMAs' issue port).
// hipcc pads no hazards around inline asm: operands come straight from loads (s_waitcnt), an accumulator is reused 4 MT MFMAs
// later, and wr_settle() stands between the last MFMA and the first read of a result.
__device__ __forceinline__ void wr_mfma(f32x4& acc, float a, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}
template <int MT>
__device__ __forceinline__ void wr_settle(f32x4 (&acc)[MT][4]) {
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
        asm volatile("s_nop 15\n\ts_nop 3" : "+a"(acc[mi][0]), "+a"(acc[mi][1]), "+a"(acc[mi][2]), "+a"(acc[mi][3]));
}

template <int MT, int NB>
__global__ __launch_bounds__(256) void wgrad_rows_kernel(WrBatch g) {
    __shared__ __attribute__((aligned(16))) float sP[4][MT * 16][64];      // [wave][(tile i * 4 + tile j) * 4 + r][lane]
    __shared__ __attribute__((aligned(16))) float sBs[4][16][4];          // [wave][li][j]
    __shared__ const float* sArow[WR_MAXROWS];
    __shared__ unsigned sBoff[WR_MAXROWS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    int pi = 0;
#pragma unroll 1
    for (int i = 1; i < g.count; ++i)
        if ((int)blockIdx.x >= g.p[i].block0) pi = i;
    const WrProb& q = g.p[pi];
#ifdef FOV_STAMPS
    const int stamp_slot = blockIdx.x == 0 ? 0 : blockIdx.x == 70 ? 1 : blockIdx.x == 150 ? 2 : blockIdx.x == gridDim.x - 1 ? 3 : -1;
#endif
    WG_STAMP(0);
#ifdef FOV_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        unsigned xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        g_wr_trace[blockIdx.x][0] = __builtin_amdgcn_s_memtime();
        g_wr_trace[blockIdx.x][3] = ((unsigned long long)xcc << 32) | hw;
        g_wr_trace[blockIdx.x][4] = __builtin_amdgcn_s_memrealtime();
    }
#endif
    const int tile = (int)blockIdx.x - q.block0;
    const int by = tile / q.grid_n, bx = tile - by * q.grid_n;
    const int m0 = by * 16 * MT, n0 = bx * WR_TN;
    const unsigned RI = (unsigned)q.RI, rows = (unsigned)q.RO * RI;
    const bool bias_blk = q.bias != nullptr && by == 0;
    // Row table, once per workgroup: where row r of the A operand starts (the tape row, h0's row for h_{-1}, or a row of zeros - rows
    // past the end, h_{-1} without h0) and the byte offset of its dz row.  The loop then spends two LDS reads and one 64-bit add per
    // load on addressing: fp32 MFMAs share the issue port with the VALU, and with the row arithmetic per lane and step (~30
    // instructions) the H = 512 launch ran at a third of the matrix pipe whatever its tile shape.
    const unsigned rows_pad = (rows + 32u * NB - 1u) / (32u * NB) * (32u * NB);      // an even number of batches
    for (unsigned r = (unsigned)tid; r < rows_pad; r += 256u) {
        unsigned ro = __umulhi(r, q.ri_magic);
        unsigned ri = r - ro * RI;
        if (ri >= RI) { ++ro; ri -= RI; }
        const bool rok = r < rows;
        const bool first = q.shift && ri == 0;                     // h_{-1}: from h0 (or zero)
        const float* ap = first ? (q.h0 ? q.h0 + (long)ro * q.ldh0 : g_wr_zero)
                                : q.a + ((long)ro * q.a_so + (long)((int)ri - q.shift) * q.lda);
        sArow[r] = rok ? ap : g_wr_zero;
        sBoff[r] = rok ? (ro * (unsigned)q.b_so + ri * (unsigned)q.ldb) * 4u : 0u;      // (a row past the end: 0 x a finite dz row)
    }
    // a lane's columns, clamped into the matrix (a column past M feeds result rows that are not stored)
    int acol[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) acol[mi] = min(m0 + 16 * mi + li, q.M - 1);
    const char* const bcol = reinterpret_cast<const char*>(q.b + (n0 + 4 * li));
    f32x4 acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    WG_STAMP(1);
    // MFMA step s covers rows 4s .. 4s + 3 (lane group lq takes row 4s + lq); wave w takes steps w, w + 4, w + 8, ... in batches of
    // NB; the loads of batch b + 1 are in flight under the MFMAs of batch b (two register sets).  A workgroup on its own spent a
    // quarter of its time issuing loads (the CU's 64 bytes per clock) before its first MFMA.
    const unsigned nbt = rows_pad / (16u * NB);
    const unsigned rbase = 4u * (unsigned)wave + (unsigned)lq;
    struct Set { float a[NB][MT]; f32x4 b[NB]; };
    auto load_batch = [&](Set& S, unsigned bt) {      // every load unconditional; a batch past the end reads the last one again (not used)
        const unsigned r0 = rbase + 16u * NB * (bt < nbt ? bt : nbt - 1u);
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const unsigned r = r0 + 16u * i;
            const wr_gfloat* ap = (const wr_gfloat*)sArow[r];      // (a pointer out of LDS: say that it is global memory, or the load is a flat one)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi) S.a[i][mi] = ap[acol[mi]];
            S.b[i] = *(const f32x4*)(bcol + sBoff[r]);
        }
    };
    auto mma_batch = [&](const Set& S, unsigned bt) {
        if (bias_blk) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                if (rbase + 16u * (NB * bt + i) < rows) bsum += S.b[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                for (int j = 0; j < 4; ++j) wr_mfma(acc[mi][j], S.a[i][mi], S.b[i][j]);
    };
    Set S0, S1;
    load_batch(S0, 0);
    WG_STAMP(2);
    for (unsigned bt = 0; bt < nbt; bt += 2) {
        load_batch(S1, bt + 1);
        mma_batch(S0, bt);
        load_batch(S0, bt + 2);
        mma_batch(S1, bt + 1);      // (nbt is even: no branch here - a merge point sends every accumulator through a VGPR and back)
    }
    wr_settle<MT>(acc);
    WG_STAMP(3);
#ifdef FOV_STAMPS
    if (threadIdx.x == 0 && blockIdx.x < 4096) g_wr_trace[blockIdx.x][1] = __builtin_amdgcn_s_memtime();
#endif
    // ---- the four waves' partial tiles, added in a fixed order; thread (wave', lane') stores rows r = wave' of lane' ----
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sP[wave][(mi * 4 + j) * 4 + r][lane] = acc[mi][j][r];
    if (bias_blk) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float t = bsum[j];
            t += __shfl_xor(t, 16);
            t += __shfl_xor(t, 32);
            bsum[j] = t;
        }
        if (lq == 0) *(f32x4*)&sBs[wave][li][0] = bsum;
    }
    __syncthreads();
    {
        const int r = wave, col = n0 + 4 * li;
#pragma unroll
        for (int mi = 0; mi < MT; ++mi) {
            const int row = m0 + 16 * mi + 4 * lq + r;
            f32x4 o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int e = (mi * 4 + j) * 4 + r;
                o[j] = (sP[0][e][lane] + sP[1][e][lane]) + (sP[2][e][lane] + sP[3][e][lane]);
            }
            if (row < q.M) {
                float* cp = q.c + (size_t)row * q.ldc + col;
                if (g.accumulate) o += *(const f32x4*)cp;
                *(f32x4*)cp = o;
            }
        }
    }
    if (bias_blk && tid < 16) {
        f32x4 t = (*(const f32x4*)&sBs[0][tid][0] + *(const f32x4*)&sBs[1][tid][0]) + (*(const f32x4*)&sBs[2][tid][0] + *(const f32x4*)&sBs[3][tid][0]);
        float* bp = q.bias + n0 + 4 * tid;
        if (g.accumulate) t += *(const f32x4*)bp;
        *(f32x4*)bp = t;
    }
#ifdef FOV_STAMPS
    if (stamp_slot >= 0 && threadIdx.x == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_wg_stamps[stamp_slot][4] = __builtin_amdgcn_s_memtime();
        g_wg_stamps[stamp_slot][5] = (unsigned long long)(nbt * NB);
    }
    if (threadIdx.x == 0 && blockIdx.x < 4096) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_wr_trace[blockIdx.x][2] = __builtin_amdgcn_s_memtime();
        g_wr_trace[blockIdx.x][5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

}  // namespace

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_wr_trace(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wr_trace), sizeof(unsigned long long) * 4096 * 6);
}
extern "C" int fov_debug_read_wg_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wg_stamps), sizeof(unsigned long long) * 4 * 8);
}
#endif

// rows small enough that one workgroup per output tile walking all of them beats the split products
// ... AND tiles enough to occupy the chip: a workgroup walks all rows of its tile at ~2 400 cycles per 16 rows, so a product of a
// dozen tiles (H = 128: 8) is faster split over the rows (measured: 36 us against 24 us for config 1's layers)
bool wgrad_group_takes(int B, int T, int H) {
    return !env_knobs().no_wgrad_group && (long)B * T <= 1024 && H >= 512 && (H & 3) == 0 && B > 0 && T > 0;
}

// Weight gradients of up to four layers in one launch.  Layer l: x_l (B,T,F_l), hs_l (B,T,H), optional h0_l (B,H), dz_l (B,T,4H) ->
// dK_l (F_l,4H), dR_l (H,4H), db_l (4H); any of dK / dR / db may be NULL (db alone is not supported: it rides on dR's launch
// blocks, or on dK's when dR is NULL).
int wgrad_group_layers(int L, const float* const* x, const int* F, const float* const* hs, const float* const* h0, const float* const* dz,
                       float* const* dK, float* const* dR, float* const* db, int B, int T, int H, int accumulate, hipStream_t stream) {
    if (L < 1 || 2 * L > kMaxProb) { set_error("wgrad_group: at most %d layers", kMaxProb / 2); return FOV_ERR_INVALID; }
    WgBatch g = {};
    g.RO = B; g.RI = T; g.accumulate = accumulate ? 1 : 0;
    g.ri_magic = T <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)T);
    const int N = 4 * H;
    int blocks = 0;
    for (int l = 0; l < L; ++l) {
        if (((uintptr_t)dz[l]) & 15) { set_error("wgrad_group: dz must be 16-byte aligned"); return FOV_ERR_INVALID; }
        float* bias = db[l];
        for (int which = 0; which < 2; ++which) {       // 0: dR (with the bias row), 1: dK
            float* c = which == 0 ? dR[l] : dK[l];
            if (!c) continue;
            WgProb& q = g.p[g.count++];
            q.a = which == 0 ? hs[l] : x[l];
            q.M = which == 0 ? H : F[l];
            q.lda = q.M; q.a_so = (long)T * q.M;
            q.shift = which == 0 ? 1 : 0;
            q.h0 = which == 0 ? h0[l] : nullptr;
            q.ldh0 = H;
            q.b = dz[l]; q.ldb = N; q.b_so = (long)T * N; q.N = N;
            q.c = c; q.ldc = N;
            q.bias = bias; bias = nullptr;               // the first problem of the layer carries db
            q.grid_n = (N + WT - 1) / WT;
            q.block0 = blocks;
            blocks += q.grid_n * ((q.M + WT - 1) / WT);
            if (((long)B * T * (q.M > N ? q.M : N)) * 4 >= (1L << 31)) { set_error("wgrad_group: operand larger than 2 GiB"); return FOV_ERR_UNSUPPORTED; }
        }
        if (bias) { set_error("wgrad_group: db needs dK or dR of the same layer"); return FOV_ERR_INVALID; }
    }
    if (blocks == 0) return FOV_OK;
    hipLaunchKernelGGL(wgrad_group_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("wgrad_group launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}


// few rows and narrow layers: one 16 x 64 tile per workgroup, rows split over its waves (wgrad_rows_kernel)
bool wgrad_rows_takes(long rows, int H) {
    return !env_knobs().no_wgrad_group && rows > 0 && rows <= WR_MAXROWS && H <= 512 && (H & 15) == 0;
}

// Weight gradients of up to four layers in one launch, every layer with its own time length T[l] (an encoder / decoder pair).
int wgrad_rows_layers(int L, const float* const* x, const int* F, const int* T, const float* const* hs, const float* const* h0,
                      const float* const* dz, float* const* dK, float* const* dR, float* const* db, int B, int H, int accumulate, hipStream_t stream) {
    if (L < 1 || 2 * L > kMaxProb) { set_error("wgrad_rows: at most %d layers", kMaxProb / 2); return FOV_ERR_INVALID; }
    WrBatch g = {};
    g.accumulate = accumulate ? 1 : 0;
    const int N = 4 * H;
    int blocks = 0;
    // wide layers: 32 x 64 tiles (half the dz traffic per MFMA).  Measured at lstm.py's shape (H = 512, 320 rows, 2.1 GFLOP): 64 x 64
    // tiles 36-38 us (one or two workgroups per CU by registers), 32 x 64 34 us, 16 x 64 40-42 us; the 128 x 128 LDS-staged
    // tiles of wgrad_group_kernel 43.5 us (profiles/r05_wgrad_rows_probe.txt)
    const int MT = H >= 512 ? 2 : 1;
    for (int l = 0; l < L; ++l) {
        if (T[l] <= 0 || B <= 0) { set_error("wgrad_rows: empty layer"); return FOV_ERR_INVALID; }
        float* bias = db[l];
        for (int which = 0; which < 2; ++which) {       // 0: dR (with the bias row), 1: dK
            float* c = which == 0 ? dR[l] : dK[l];
            if (!c) continue;
            WrProb& q = g.p[g.count++];
            q.a = which == 0 ? hs[l] : x[l];
            q.M = which == 0 ? H : F[l];
            q.lda = q.M; q.a_so = T[l] * q.M;
            q.shift = which == 0 ? 1 : 0;
            q.h0 = which == 0 ? h0[l] : nullptr;
            q.ldh0 = H;
            q.b = dz[l]; q.ldb = N; q.b_so = T[l] * N; q.N = N;
            q.c = c; q.ldc = N;
            q.RO = B; q.RI = T[l];
            q.ri_magic = T[l] <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)T[l]);
            q.bias = bias; bias = nullptr;
            q.grid_n = N / WR_TN;
            q.block0 = blocks;
            blocks += q.grid_n * ((q.M + 16 * MT - 1) / (16 * MT));
            if (q.M > WR_ZCOLS || (long)B * T[l] > WR_MAXROWS || (long)B * T[l] * N * 4 >= (1L << 31)) { set_error("wgrad_rows: operand too large"); return FOV_ERR_UNSUPPORTED; }
            if ((((uintptr_t)q.b) | ((uintptr_t)q.c) | ((uintptr_t)q.bias)) & 15) { set_error("wgrad_rows: dz, dK, dR, db must be 16-byte aligned"); return FOV_ERR_INVALID; }
        }
        if (bias) { set_error("wgrad_rows: db needs dK or dR of the same layer"); return FOV_ERR_INVALID; }
    }
    if (blocks == 0) return FOV_OK;
    if (env_knobs().dbg_trace) fprintf(stderr, "[fov trace] wgrad_rows: %d problems, %d workgroups, M tiles %d\n", g.count, blocks, MT);
    const dim3 grid((unsigned)blocks), blk(256);
    if (MT == 2) hipLaunchKernelGGL((wgrad_rows_kernel<2, 10>), grid, blk, 0, stream, g);
    else hipLaunchKernelGGL((wgrad_rows_kernel<1, 5>), grid, blk, 0, stream, g);      // (batches of 10: H = 256 11.1 -> 14.2 us)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("wgrad_rows launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
