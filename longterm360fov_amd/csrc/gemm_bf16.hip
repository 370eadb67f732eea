// bf16 matrix-core GEMMs of the bf16 training step (BASELINE.json configs[4]): fp32 tensors in HBM, operands rounded to
// bf16 (RNE) while they are staged into LDS, v_mfma_f32_16x16x32_bf16, fp32 accumulation, fp32 result.
//
//   gemm_bf16_tn : C (M,N) (+)= sum_r A[r][m] * B[r][n]        weight gradients  dK = x^T dz,  dR = h_prev^T dz
//                  rows r = (ro, ri), ro < RO, ri < RI; A row at a + ro*a_so + ri*lda, B row at b + ro*b_so + ri*ldb:
//                  the two-level row index contracts (batch, time) pairs with a time shift without copies
//                  (dR = sum_b sum_{t>=1} hs[b][t-1]^T dz[b][t]).  Split over the rows (deterministic: partials +
//                  fixed-order reduce, no float atomics).
//   gemm_bf16_nt : C (M,N) = sum_k A[m][k] * B[n][k]           data gradient  dx = dz . K^T
//
// TN: both operands have the contraction index on their slow axis.  They are staged as they lie ([r][m], [r][n] bf16
// images, 16-byte global loads along the fast axis, ds_write_b64) and the MFMA fragments - 8 consecutive-k values of
// one column - come out of the image through ds_read_b64_tr_b16, the hardware transpose read (cdna_hip_programming.md
// T10): no transposed copy, no ds_permute.  The k order inside a 32-block is permuted (element j of lane group g4 is
// row 16*(j>>2) + 4*g4 + (j&3)) - the same for A and B, so the product is unchanged - which makes the eight rows a
// 32-lane half touches contiguous: with a row stride of 72 dwords every transposed read is bank-conflict-free.
// Both kernels are HBM/L2-bound at the sizes of this path (a 5120 x 256 x 1024 product is 2.7 GFLOP = 1 us of bf16
// matrix time against 26 MB of fp32 operands).
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

namespace {

typedef short qs16x4 __attribute__((ext_vector_type(4)));

constexpr int GT = 128;          // output tile (GT x GT), 4 waves as 2 x 2, each 64 x 64 = 4 x 4 MFMA tiles
constexpr int GKB = 32;          // rows (k) per stage
constexpr int GLD_TN = 144;      // bf16 per LDS row of a [k][GT] image: 72 dwords = 8 (mod 64)
constexpr int GKN = 64;          // k-values per stage of the NT product
constexpr int GLD_NT = 72;       // bf16 per LDS row of a [GT][k] image: 36 dwords = 4 (mod 32), conflict-free 16-byte reads

struct GemmTN {
    const float* a;
    const float* b;
    float* c;            // split == 1: C (ldc); else partials [slice][M + bias_row][N]
    int M, N, RO, RI;
    long lda, ldb, a_so, b_so;
    int ldc;
    int split;           // row slices (grid.z)
    long rows_per_split; // multiple of GKB
    int add_c;           // split == 1 only: C += A^T B
    int xcd_remap, grid_n, grid_m;   // 1-D grid, slice s on XCD s % 8 (split >= 8): the blocks of a slice read the same rows of
                                     // A and B, so each XCD's L2 fetches 1/8 of the operands once instead of all of A
    // fused weight gradient [dK ; dR ; db] = [x | h_prev | 1]^T dz (one product, one reduce):
    const float* a2;     // second A operand: rows [M1, M) of C come from it (M1 % GT == 0), or NULL
    long lda2, a2_so;
    int M1;
    int a_shift, a2_shift;   // 1: operand row (ro, ri) is its element (ro, ri - 1), zero at ri == 0 (h_{t-1} of a (B,T,H) tape)
    int bias_row;        // 1: row M of C = column sums of B (fp32, taken from the staged values before they are rounded)
    unsigned ri_magic;   // floor(2^32 / RI) (0xffffffff for RI = 1): r / RI = umulhi(r, magic) or that + 1 (tn_row_split)
};

// (ro, ri) = divmod(r, RI) in five instructions.  The loaders form it for every row of every stage: with the 32-bit integer
// division the compiler emits (~35 instructions) and 64-bit offset arithmetic the products were VALU-bound on ADDRESS arithmetic
// - 19 us for a product whose loads, LDS traffic and MFMAs need 7 (r04: tools/bench_wgrad_bf16.py under rocprofv3).
__device__ __forceinline__ void tn_row_split(unsigned r, unsigned RI, unsigned magic, unsigned& ro, unsigned& ri) {
    unsigned q = __umulhi(r, magic);
    unsigned rem = r - q * RI;
    if (rem >= RI) { ++q; rem -= RI; }
    ro = q; ri = rem;
}

// two transposed reads = the 8 k-values of one column for this lane group (k permutation in the file header)
__device__ __forceinline__ qu32x4 tr_frag(const unsigned short* img, int g4, int n, int col0) {
    const int q = n >> 2, p = n & 3;
    const unsigned short* p0 = img + (4 * g4 + q) * GLD_TN + col0 + 4 * p;
    const qs16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((qs16x4 __attribute__((address_space(3)))*)p0);
    const qs16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((qs16x4 __attribute__((address_space(3)))*)(p0 + 16 * GLD_TN));
    const qu32x2 l2 = __builtin_bit_cast(qu32x2, lo), h2 = __builtin_bit_cast(qu32x2, hi);
    return (qu32x4){l2.x, l2.y, h2.x, h2.y};
}

template <bool AVEC>   // AVEC: 16-byte loads of A rows (lda % 4 == 0, a 16-byte aligned, M % 4 == 0); B always is
__global__ __launch_bounds__(256) void gemm_bf16_tn_kernel(GemmTN g) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][GKB * GLD_TN];
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][GKB * GLD_TN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_remap) {
        const int nb = g.grid_n * g.grid_m;
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        bz = xcd + 8 * (w / nb);
        if (bz >= g.split) return;   // whole block, before any barrier
        const int tile = w - (w / nb) * nb;
        by = tile / g.grid_n;
        bx = tile - by * g.grid_n;
    }
    const int m0 = by * GT, n0 = bx * GT;
    // which A operand this row tile reads (block-uniform)
    const bool second = g.a2 != nullptr && m0 >= g.M1;
    const float* abase = second ? g.a2 : g.a;
    const long a_ld = second ? g.lda2 : g.lda, a_so = second ? g.a2_so : g.a_so;
    const int a_shift = second ? g.a2_shift : g.a_shift;
    const int am0 = second ? m0 - g.M1 : m0;                           // first row of the tile inside its operand
    const int a_M = second ? g.M - g.M1 : (g.a2 ? g.M1 : g.M);          // width of that operand
    const bool bias_blk = g.bias_row && by == 0;
    const long rows = (long)g.RO * g.RI;
    const long r_lo = (long)bz * g.rows_per_split;
    long r_hi = r_lo + g.rows_per_split;
    if (r_hi > rows) r_hi = rows;
    // staging: thread = (row tid >> 5 (+ 8 i), 4 columns (tid & 31) * 4)
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[4], rb[4];
    f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};   // bias_blk: column sums of this thread's B elements
    // Every load is UNCONDITIONAL: buffer loads through descriptors of the whole operands, rows past the slice, the
    // zero row of a shifted operand and columns past the matrix present an out-of-range offset and read as 0.  (A load
    // inside a branch gets an s_waitcnt vmcnt(0) at the merge: the eight loads of a stage then complete one after the
    // other - measured 1.7 us per stage instead of one memory latency.)
    constexpr unsigned OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, 0x7fffffff, 0x00020000);
    const bool a_col_ok = am0 + scol < a_M;             // AVEC: a_M % 4 == 0, the whole quad is in or out
    const bool b_col_ok = n0 + scol + 3 < g.N;          // N % 4 == 0 (host-checked)
    const unsigned a_so32 = (unsigned)a_so, a_ld32 = (unsigned)a_ld, b_so32 = (unsigned)g.b_so, b_ld32 = (unsigned)g.ldb;
    auto load_stage = [&](long r0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const long r = r0 + srow + 8 * i;
            unsigned ro, ri;
            tn_row_split((unsigned)r, (unsigned)g.RI, g.ri_magic, ro, ri);
            const bool rok = r < r_hi;
            const bool a_ok = rok && !(a_shift && ri == 0);
            // byte offsets < 2^31 (host-checked): 32-bit arithmetic is exact (the shifted row of ri == 0 is masked)
            const unsigned aoff = (ro * a_so32 + (ri - (unsigned)a_shift) * a_ld32 + (unsigned)(am0 + scol)) * 4u;
            const unsigned boff = (ro * b_so32 + ri * b_ld32 + (unsigned)(n0 + scol)) * 4u;
            if (AVEC) {
                const qu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(ars, (a_ok && a_col_ok) ? aoff : OOR, 0, 0);
                ra[i] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
            } else {
#pragma unroll
                for (int v = 0; v < 4; ++v)
                    ra[i][v] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ars, (a_ok && am0 + scol + v < a_M) ? aoff + 4 * v : OOR, 0, 0));
            }
            const qu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(brs, (rok && b_col_ok) ? boff : OOR, 0, 0);
            rb[i] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *(qu32x2*)(sA[buf] + (srow + 8 * i) * GLD_TN + scol) = (qu32x2){pack_bf16(ra[i][0], ra[i][1]), pack_bf16(ra[i][2], ra[i][3])};
            *(qu32x2*)(sB[buf] + (srow + 8 * i) * GLD_TN + scol) = (qu32x2){pack_bf16(rb[i][0], rb[i][1]), pack_bf16(rb[i][2], rb[i][3])};
        }
        if (bias_blk) {   // every stage is stored exactly once
#pragma unroll
            for (int i = 0; i < 4; ++i) bsum += rb[i];
        }
    };
    const long nstages = (r_hi > r_lo) ? (r_hi - r_lo + GKB - 1) / GKB : 0;
    // (two stages in flight from one block - a second register set - was measured slower: 322 registers, one block per CU
    // instead of three; the other blocks of the CU are what keeps more loads in flight)
    if (nstages > 0) {
        load_stage(r_lo);
        store_stage(0);
    }
    __syncthreads();
    for (long s = 0; s < nstages; ++s) {
        const int buf = (int)(s & 1);
        if (s + 1 < nstages) load_stage(r_lo + (s + 1) * GKB);   // global loads of the next stage fly under the MFMAs
        qu32x4 af[4], bf[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            af[i] = tr_frag(sA[buf], g4, n, wm * 64 + i * 16);
            bf[i] = tr_frag(sB[buf], g4, n, wn * 64 + i * 16);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) qmfma(acc[i][j], af[i], bf[j]);
        if (s + 1 < nstages) store_stage(buf ^ 1);
        __syncthreads();
    }
    float* cbase = g.split > 1 ? g.c + (size_t)bz * (g.M + g.bias_row) * g.N : g.c;
    const int ldc = g.split > 1 ? g.N : g.ldc;
    if (bias_blk) {   // row M: the eight row groups' column sums, folded in a fixed order (the loop ended on a barrier)
        float* red = reinterpret_cast<float*>(sA[0]);
        *(f32x4*)(red + srow * GT + scol) = bsum;
        __syncthreads();
        if (tid < GT && n0 + tid < g.N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 8; ++q) t += red[q * GT + tid];
            float* cp = cbase + (size_t)g.M * ldc + n0 + tid;
            *cp = (g.split == 1 && g.add_c) ? *cp + t : t;
        }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + 4 * g4 + r;
                if (row < g.M && col < g.N) {
                    float* cp = cbase + (size_t)row * ldc + col;
                    *cp = (g.split == 1 && g.add_c) ? *cp + acc[i][j][r] : acc[i][j][r];
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------------------------
// Round 4: the same TN product with THREE stages of global loads in flight per block.  The kernel above keeps one stage in
// flight (its 32 staging registers per thread and stage leave room for no more at three blocks per CU): a block's 13 stages
// each cost one memory latency (~1.3 us), 21 us for a 5120 x 513 x 1024 product that moves 31 MB - a quarter of the HBM rate.
// Here a block has 512 threads (8 waves as 2 x 4, a wave owns 64 x 32 of the 128 x 128 tile): the staging of a stage is 16
// registers per thread, three register sets rotate (loads of stages s+2, s+3, s+4 fly while stage s is multiplied and
// stage s+1 is rounded into LDS), three LDS images rotate with them, one barrier per stage.  One block per CU, so the
// row slices are cut to give about one block per CU (fewer, longer slices: 12 MB of partials instead of 27).
// Same arithmetic per output element (k order inside a slice, fixed-order reduce over slices); slices differ from the
// kernel above, so the two agree to rounding, not bit for bit.  Takes the 16-byte-aligned forms (AVEC); the rest stays above.
// ---------------------------------------------------------------------------------------------------------------
constexpr int G3T = 512;                 // threads
#ifdef FOV_STAMPS
__device__ unsigned long long g_tn3_stamps[4][8];   // [block 0, 100, 200, last][entry, loads issued, first stage in LDS, loop end, bias done, stored]
#define TN3_STAMP(slot) do { if (stamp_slot >= 0 && tid == 0) g_tn3_stamps[stamp_slot][slot] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define TN3_STAMP(slot) do { } while (0)
#endif

constexpr int G3KB = 32;                 // rows (k) per stage (64-row stages were measured: 2 050 cycles per stage against 933 for
                                         // 32 - no gain per row, a longer prologue and more padding of the stage count)
constexpr int G3NR = G3KB / 16;          // rows per thread and stage
constexpr int G3_CLD = 36;               // floats per row of a wave's 64 x 32 result tile in LDS (epilogue)
constexpr size_t G3_IMG = (size_t)2 * 3 * G3KB * GLD_TN * sizeof(unsigned short), G3_TILES = (size_t)8 * 64 * G3_CLD * sizeof(float);
constexpr size_t G3_LDS = G3_IMG > G3_TILES ? G3_IMG : G3_TILES;
struct G3Regs { f32x4 a[G3NR], b[G3NR]; };

template <bool AVEC>   // AVEC: 16-byte loads of A rows; else four 4-byte loads (width or row stride not a multiple of 4: the 90-wide encoder input)
__global__ __launch_bounds__(G3T, 1) void gemm_bf16_tn3_kernel(GemmTN g) {
    __shared__ __attribute__((aligned(16))) unsigned char sRaw[G3_LDS];   // [A | B][image 3][row][column] bf16; the epilogue's tiles
    unsigned short (*sA)[G3KB * GLD_TN] = reinterpret_cast<unsigned short (*)[G3KB * GLD_TN]>(sRaw);
    unsigned short (*sB)[G3KB * GLD_TN] = sA + 3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 2, wn = wave & 3;          // 2 x 4 waves: rows 64*wm, columns 32*wn
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (g.xcd_remap) {
        const int nb = g.grid_n * g.grid_m;
        const int xcd = blockIdx.x & 7, w = blockIdx.x >> 3;
        bz = xcd + 8 * (w / nb);
        if (bz >= g.split) return;   // whole block, before any barrier
        const int tile = w - (w / nb) * nb;
        by = tile / g.grid_n;
        bx = tile - by * g.grid_n;
    }
    const int m0 = by * GT, n0 = bx * GT;
#ifdef FOV_STAMPS
    const int stamp_slot = blockIdx.x == 0 ? 0 : blockIdx.x == 100 ? 1 : blockIdx.x == 200 ? 2 : blockIdx.x == gridDim.x - 1 ? 3 : -1;
#endif
    TN3_STAMP(0);
    const bool second = g.a2 != nullptr && m0 >= g.M1;
    const float* abase = second ? g.a2 : g.a;
    const long a_ld = second ? g.lda2 : g.lda, a_so = second ? g.a2_so : g.a_so;
    const int a_shift = second ? g.a2_shift : g.a_shift;
    const int am0 = second ? m0 - g.M1 : m0;
    const int a_M = second ? g.M - g.M1 : (g.a2 ? g.M1 : g.M);
    const bool bias_blk = g.bias_row && by == 0;
    const long rows = (long)g.RO * g.RI;
    const long r_lo = (long)bz * g.rows_per_split;
    long r_hi = r_lo + g.rows_per_split;
    if (r_hi > rows) r_hi = rows;
    // staging: thread = (row tid >> 5 (+ 16 i, i < G3NR), 4 columns (tid & 31) * 4)
    const int srow = tid >> 5, scol = (tid & 31) * 4;
    f32x4 acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = (f32x4){0.f, 0.f, 0.f, 0.f};
    constexpr unsigned OOR = 0x80000000u;   // unconditional buffer loads, out-of-range offsets read as 0 (see the kernel above)
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(abase), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, 0x7fffffff, 0x00020000);
    const bool a_col_ok = am0 + scol < a_M;
    const bool b_col_ok = n0 + scol + 3 < g.N;
    // Row cursors of this thread's rows, advanced by one stage per load_stage call - additions and one wrap
    // test instead of a division and four integer multiplies per row and stage (quarter-rate instructions: the loop was bound
    // by instruction issue, 1 236 cycles per 32-row stage of which ~300 were address arithmetic; tools/stamp_tn3.py).
    //   r = (ro, ri);  G3KB = dq * RI + dr:  ri += dr, ro += dq, one carry;  the byte offsets follow with the same carry.
    const unsigned RI = (unsigned)g.RI;
    const unsigned dq = (unsigned)G3KB / RI, dr = (unsigned)G3KB - dq * RI;
    const unsigned a_so4 = (unsigned)a_so * 4u, a_ld4 = (unsigned)a_ld * 4u, b_so4 = (unsigned)g.b_so * 4u, b_ld4 = (unsigned)g.ldb * 4u;
    const unsigned a_step = dq * a_so4 + dr * a_ld4, b_step = dq * b_so4 + dr * b_ld4;       // per stage, no carry
    const unsigned a_carry = a_so4 - RI * a_ld4, b_carry = b_so4 - RI * b_ld4;               // added when ri wraps (mod 2^32)
    unsigned c_ri[G3NR], c_aoff[G3NR], c_boff[G3NR], c_r[G3NR];   // rows < 2^31 (host-checked)
    const unsigned r_hi32 = (unsigned)r_hi;
#pragma unroll
    for (int i = 0; i < G3NR; ++i) {
        c_r[i] = (unsigned)r_lo + srow + 16 * i;
        unsigned ro;
        tn_row_split(c_r[i], RI, g.ri_magic, ro, c_ri[i]);
        // byte offsets < 2^31 (host-checked): 32-bit arithmetic is exact (the shifted row of ri == 0 is masked)
        c_aoff[i] = ro * a_so4 + (c_ri[i] - (unsigned)a_shift) * a_ld4 + (unsigned)(am0 + scol) * 4u;
        c_boff[i] = ro * b_so4 + c_ri[i] * b_ld4 + (unsigned)(n0 + scol) * 4u;
    }
    auto load_stage = [&](G3Regs& R) {   // the NEXT stage of the slice (calls are in stage order)
#pragma unroll
        for (int i = 0; i < G3NR; ++i) {
            const bool rok = c_r[i] < r_hi32;
            const bool a_ok = rok && !(a_shift && c_ri[i] == 0);
            if constexpr (AVEC) {
                const qu32x4 ta = __builtin_amdgcn_raw_buffer_load_b128(ars, (a_ok && a_col_ok) ? c_aoff[i] : OOR, 0, 0);
                R.a[i] = (f32x4){__uint_as_float(ta[0]), __uint_as_float(ta[1]), __uint_as_float(ta[2]), __uint_as_float(ta[3])};
            } else {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    R.a[i][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(ars, (a_ok && am0 + scol + e < a_M) ? c_aoff[i] + 4u * e : OOR, 0, 0));
            }
            const qu32x4 tb = __builtin_amdgcn_raw_buffer_load_b128(brs, (rok && b_col_ok) ? c_boff[i] : OOR, 0, 0);
            R.b[i] = (f32x4){__uint_as_float(tb[0]), __uint_as_float(tb[1]), __uint_as_float(tb[2]), __uint_as_float(tb[3])};
            c_r[i] += G3KB;
            c_ri[i] += dr;
            c_aoff[i] += a_step;
            c_boff[i] += b_step;
            const bool wrap = c_ri[i] >= RI;
            c_ri[i] -= wrap ? RI : 0u;
            c_aoff[i] += wrap ? a_carry : 0u;
            c_boff[i] += wrap ? b_carry : 0u;
        }
    };
    auto store_stage = [&](int buf, const G3Regs& R) {
#pragma unroll
        for (int i = 0; i < G3NR; ++i) {
            *(qu32x2*)(sA[buf] + (srow + 16 * i) * GLD_TN + scol) = (qu32x2){pack_bf16(R.a[i][0], R.a[i][1]), pack_bf16(R.a[i][2], R.a[i][3])};
            *(qu32x2*)(sB[buf] + (srow + 16 * i) * GLD_TN + scol) = (qu32x2){pack_bf16(R.b[i][0], R.b[i][1]), pack_bf16(R.b[i][2], R.b[i][3])};
        }
        if (bias_blk) {   // every stage is stored exactly once
#pragma unroll
            for (int i = 0; i < G3NR; ++i) bsum += R.b[i];
        }
    };
    // The loop body is BRANCH-FREE: the stage count is rounded up to a multiple of three and every load / LDS store runs
    // unconditionally (rows past the slice present an out-of-range offset and read as zeros: a zero stage adds nothing).  A
    // load inside an `if` makes its registers a phi of old and new values - the compiler then copies them behind an
    // s_waitcnt vmcnt(0) right after the issue, and the three stages in flight collapse into one (first version: 22 us).
    const long nst = (r_hi > r_lo) ? (r_hi - r_lo + G3KB - 1) / G3KB : 0;
    const long nstages = (nst + 2) / 3 * 3;
    G3Regs R0, R1, R2;
    // set (s % 3) holds stage s until it is rounded into LDS image (s % 3); it then takes the loads of stage s + 3
    load_stage(R0);
    load_stage(R1);
    load_stage(R2);
    TN3_STAMP(1);
    store_stage(0, R0);
    load_stage(R0);
    __syncthreads();
    TN3_STAMP(2);
    auto iter = [&](int buf, int nbuf, G3Regs& Rn) {   // multiply the stage in image buf; the next stage (set Rn) -> image nbuf
        qu32x4 af[G3KB / 32][4], bf[G3KB / 32][2];
#pragma unroll
        for (int kb = 0; kb < G3KB / 32; ++kb) {
#pragma unroll
            for (int i = 0; i < 4; ++i) af[kb][i] = tr_frag(sA[buf] + kb * 32 * GLD_TN, g4, n, wm * 64 + i * 16);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[kb][j] = tr_frag(sB[buf] + kb * 32 * GLD_TN, g4, n, wn * 32 + j * 16);
        }
#pragma unroll
        for (int kb = 0; kb < G3KB / 32; ++kb)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) qmfma(acc[i][j], af[kb][i], bf[kb][j]);
        store_stage(nbuf, Rn);
        load_stage(Rn);
        __syncthreads();
    };
    for (long s = 0; s < nstages; s += 3) {
        iter(0, 1, R1);
        iter(1, 2, R2);
        iter(2, 0, R0);
    }
    TN3_STAMP(3);
    float* cbase = g.split > 1 ? g.c + (size_t)bz * (g.M + g.bias_row) * g.N : g.c;
    const int ldc = g.split > 1 ? g.N : g.ldc;
    float* red = reinterpret_cast<float*>(sRaw);      // the loop ended on a barrier: the images are free
    if (bias_blk) {   // row M: the sixteen row groups' column sums, folded in a fixed order
        *(f32x4*)(red + srow * GT + scol) = bsum;
        __syncthreads();
        if (tid < GT && n0 + tid < g.N) {
            float t = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) t += red[q * GT + tid];
            float* cp = cbase + (size_t)g.M * ldc + n0 + tid;
            *cp = (g.split == 1 && g.add_c) ? *cp + t : t;
        }
        __syncthreads();
    }
    // Result tile through LDS: a lane holds four ROWS of one column (the MFMA's layout) - 32 four-byte stores per thread, one
    // 64-byte piece per row and instruction (3 500 cycles of the block's life).  Each wave lays its 64 x 32 tile down in LDS
    // (row stride 36: the four row groups of a store land 16 banks apart) and reads it back as four consecutive COLUMNS per
    // lane: eight 16-byte stores of 128 contiguous bytes per row.
    {
        float* tile = red + wave * (64 * G3_CLD);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) tile[(i * 16 + 4 * g4 + r) * G3_CLD + j * 16 + n] = acc[i][j][r];
        // (a wave reads back only what it wrote: no barrier, the LDS operations of a wave complete in order)
        const int rr = lane >> 3, c4 = (lane & 7) * 4;
        const bool vec_ok = (ldc & 3) == 0 && (((uintptr_t)cbase) & 15) == 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int rl = rr + 8 * k;
            const f32x4 v = *(const f32x4*)(tile + rl * G3_CLD + c4);
            const int row = m0 + wm * 64 + rl, col = n0 + wn * 32 + c4;
            if (row < g.M) {
                float* cp = cbase + (size_t)row * ldc + col;
                if (vec_ok && col + 3 < g.N) {
                    f32x4 o = v;
                    if (g.split == 1 && g.add_c) o += *(const f32x4*)cp;
                    *(f32x4*)cp = o;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (col + e < g.N) cp[e] = (g.split == 1 && g.add_c) ? cp[e] + v[e] : v[e];
                }
            }
        }
    }
#ifdef FOV_STAMPS
    if (stamp_slot >= 0 && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        g_tn3_stamps[stamp_slot][4] = __builtin_amdgcn_s_memtime();
        g_tn3_stamps[stamp_slot][5] = (unsigned long long)nstages;
    }
#endif
}

struct GemmNT {
    const float* a;   // (M, K) row-major, lda
    const float* b;   // (N, K) row-major, ldb
    float* c;         // (M, N), ldc
    int M, N, K;
    long lda, ldb;
    int ldc;
};

// K % 4 == 0, lda % 4 == 0, ldb % 4 == 0, 16-byte aligned bases (host-checked).  64 k-values per stage: a block's
// serial chain of (global load -> convert -> LDS -> MFMA) stages is what bounds this product (80 blocks of a
// 5120 x 256 x 1024 data gradient, 1.2 us per stage whatever its size), so the stages are made large and few.
__global__ __launch_bounds__(256) void gemm_bf16_nt_kernel(GemmNT g) {
    __shared__ __attribute__((aligned(16))) unsigned short sA[2][GT * GLD_NT];
    __shared__ __attribute__((aligned(16))) unsigned short sB[2][GT * GLD_NT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.y * GT, n0 = blockIdx.x * GT;
    // staging: thread = (row tid >> 4 (+ 16 i), 4 k-values (tid & 15) * 4)
    const int srow = tid >> 4, scol = (tid & 15) * 4;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 ra[8], rb[8];
    // unconditional buffer loads (see the TN kernel): rows past the matrix and k past K present an out-of-range offset
    constexpr unsigned OOR = 0x80000000u;
    const __amdgpu_buffer_rsrc_t ars = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.a), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.b), 0, 0x7fffffff, 0x00020000);
    auto load_stage = [&](int k0) {
        const bool kin = k0 + scol < g.K;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int row = srow + 16 * i;
            const unsigned aoff = (unsigned)(((long)(m0 + row) * g.lda + k0 + scol) * 4);
            const unsigned boff = (unsigned)(((long)(n0 + row) * g.ldb + k0 + scol) * 4);
            const qu32x4 ta = __builtin_amdgcn_raw_buffer_load_b128(ars, (kin && m0 + row < g.M) ? aoff : OOR, 0, 0);
            const qu32x4 tb = __builtin_amdgcn_raw_buffer_load_b128(brs, (kin && n0 + row < g.N) ? boff : OOR, 0, 0);
            ra[i] = (f32x4){__uint_as_float(ta[0]), __uint_as_float(ta[1]), __uint_as_float(ta[2]), __uint_as_float(ta[3])};
            rb[i] = (f32x4){__uint_as_float(tb[0]), __uint_as_float(tb[1]), __uint_as_float(tb[2]), __uint_as_float(tb[3])};
        }
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            *(qu32x2*)(sA[buf] + (srow + 16 * i) * GLD_NT + scol) = (qu32x2){pack_bf16(ra[i][0], ra[i][1]), pack_bf16(ra[i][2], ra[i][3])};
            *(qu32x2*)(sB[buf] + (srow + 16 * i) * GLD_NT + scol) = (qu32x2){pack_bf16(rb[i][0], rb[i][1]), pack_bf16(rb[i][2], rb[i][3])};
        }
    };
    const int nstages = (g.K + GKN - 1) / GKN;
    load_stage(0);
    store_stage(0);
    __syncthreads();
    for (int s = 0; s < nstages; ++s) {
        const int buf = s & 1;
        if (s + 1 < nstages) load_stage((s + 1) * GKN);
#pragma unroll
        for (int kb = 0; kb < GKN / 32; ++kb) {
            qu32x4 af[4], bf[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                af[i] = *(const qu32x4*)(sA[buf] + (wm * 64 + i * 16 + n) * GLD_NT + 32 * kb + 8 * g4);
                bf[i] = *(const qu32x4*)(sB[buf] + (wn * 64 + i * 16 + n) * GLD_NT + 32 * kb + 8 * g4);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) qmfma(acc[i][j], af[i], bf[j]);
        }
        if (s + 1 < nstages) store_stage(buf ^ 1);
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int col = n0 + wn * 64 + j * 16 + n;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + 4 * g4 + r;
                if (row < g.M && col < g.N) g.c[(size_t)row * g.ldc + col] = acc[i][j][r];
            }
        }
}

}  // namespace

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_tn3_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_tn3_stamps), sizeof(unsigned long long) * 4 * 8);
}
#endif

size_t gemm_bf16_tn_scratch_floats(int M, int N) { return (size_t)32 * (M + 1) * N; }

// C (M + bias_row, N) (+)= [A1 | A2 | 1]^T B over rows (ro, ri); scratch holds the split partials.  A2 may be NULL (M2 = 0).
int gemm_bf16_tn_fused(const float* a1, long lda1, long a1_so, int M1, int shift1, const float* a2, long lda2, long a2_so, int M2,
                       int shift2, const float* b, long ldb, long b_so, float* c, int ldc, int N, int RO, int RI, int bias_row,
                       int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    const int M = M1 + (a2 ? M2 : 0);
    if (M <= 0 || N <= 0) return FOV_OK;
    const long rows = (long)RO * RI;
    if ((ldb & 3) || (b_so & 3) || (((uintptr_t)b) & 15) || (N & 3)) { set_error("gemm_bf16_tn: B must be 16-byte aligned with N, ldb % 4 == 0"); return FOV_ERR_INVALID; }
    if (a2 && (M1 % GT)) { set_error("gemm_bf16_tn: the first operand of a fused product must be a multiple of %d wide", GT); return FOV_ERR_INVALID; }
    {   // 31-bit byte offsets inside one buffer descriptor per operand
        auto span = [&](long so, long ld, int width) { return ((long)(RO - 1) * so + (long)RI * ld + width) * 4; };
        if (span(a1_so, lda1, M1) >= (1L << 31) || (a2 && span(a2_so, lda2, M2) >= (1L << 31)) || span(b_so, ldb, N) >= (1L << 31) ||
            rows >= (1L << 31)) {
            set_error("gemm_bf16_tn: operand larger than 2 GiB");
            return FOV_ERR_UNSUPPORTED;
        }
    }
    if (rows <= 0) {
        if (!accumulate) (void)hipMemsetAsync(c, 0, sizeof(float) * (size_t)(M + bias_row) * ldc, stream);
        return FOV_OK;
    }
    GemmTN g = {};
    g.a = a1; g.b = b; g.M = M; g.N = N; g.RO = RO; g.RI = RI; g.lda = lda1; g.ldb = ldb; g.a_so = a1_so; g.b_so = b_so; g.ldc = ldc;
    g.a2 = a2; g.lda2 = lda2; g.a2_so = a2_so; g.M1 = M1; g.a_shift = shift1; g.a2_shift = shift2; g.bias_row = bias_row ? 1 : 0;
    g.ri_magic = RI <= 1 ? 0xffffffffu : (unsigned)((1ull << 32) / (unsigned)RI);
    const int tiles = ((M + GT - 1) / GT) * ((N + GT - 1) / GT);
    const bool avec = (lda1 & 3) == 0 && (a1_so & 3) == 0 && (((uintptr_t)a1) & 15) == 0 && (M1 & 3) == 0 &&
                      (!a2 || ((lda2 & 3) == 0 && (a2_so & 3) == 0 && (((uintptr_t)a2) & 15) == 0 && (M2 & 3) == 0));
    const bool deep = !env_knobs().gemm_bf16_shallow;   // three stages in flight, one 512-thread block per CU
    // enough row slices to fill the chip, at least 8 stages each, at most 32 slices and what the scratch holds; the deep
    // kernel runs one block per CU: as many slices as give at most one block per CU (a second, partial round would double
    // the product's time)
    int split = deep ? device_cu_count() / tiles : (2 * device_cu_count() + tiles - 1) / tiles;
    if (split < 1) split = 1;
    const long max_by_rows = rows / (8 * GKB);
    if (split > max_by_rows) split = (int)max_by_rows;
    if (split > 32) split = 32;
    if (const int v = env_knobs().gemm_bf16_split) { if (v >= 1 && v <= max_by_rows) split = v; }   // tuning knob, FOV_GEMM_BF16_SPLIT
    while (split > 1 && (size_t)split * (M + g.bias_row) * N > scratch_floats) --split;
    if (split < 1) split = 1;
    long rps = (rows + split - 1) / split;
    rps = (rps + GKB - 1) / GKB * GKB;
    split = (int)((rows + rps - 1) / rps);
    g.split = split;
    g.rows_per_split = rps;
    g.add_c = accumulate;
    bool deferred = false;
    if (split > 1) {
        if (float* arena = defer_alloc(c, (size_t)(M + g.bias_row) * N, (size_t)split * (M + g.bias_row) * N, stream)) { scratch = arena; deferred = true; }
    } else if (int rc_ = defer_touch(c, (size_t)(M + g.bias_row) * ldc, stream)) {
        return rc_;
    }
    g.c = split > 1 ? scratch : c;
    if (split > 1 && ldc != N) { set_error("gemm_bf16_tn: split products need a dense C"); return FOV_ERR_INVALID; }
    g.grid_n = (N + GT - 1) / GT;
    g.grid_m = (M + GT - 1) / GT;
    g.xcd_remap = (split >= 8 && !env_knobs().gemm_bf16_noremap) ? 1 : 0;
    const dim3 grid = g.xcd_remap ? dim3((unsigned)(8 * ((split + 7) / 8) * g.grid_n * g.grid_m)) : dim3(g.grid_n, g.grid_m, split);
    if (deep && avec) hipLaunchKernelGGL(gemm_bf16_tn3_kernel<true>, grid, dim3(G3T), 0, stream, g);
    else if (deep) hipLaunchKernelGGL(gemm_bf16_tn3_kernel<false>, grid, dim3(G3T), 0, stream, g);
    else if (avec) hipLaunchKernelGGL(gemm_bf16_tn_kernel<true>, grid, dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(gemm_bf16_tn_kernel<false>, grid, dim3(256), 0, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("gemm_bf16_tn launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    if (split > 1) return reduce_or_defer(deferred, scratch, c, (long)(M + g.bias_row) * N, split, accumulate, stream, "splitk_reduce");
    return FOV_OK;
}

// C (M,N) (+)= sum over rows (ro, ri) of A[row][m] B[row][n]
int gemm_bf16_tn(const float* a, long lda, long a_so, const float* b, long ldb, long b_so, float* c, int ldc, int M, int N, int RO,
                 int RI, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    return gemm_bf16_tn_fused(a, lda, a_so, M, 0, nullptr, 0, 0, 0, 0, b, ldb, b_so, c, ldc, N, RO, RI, 0, accumulate, scratch,
                              scratch_floats, stream);
}

int gemm_bf16_nt(const float* a, long lda, const float* b, long ldb, float* c, int ldc, int M, int N, int K, hipStream_t stream) {
    if (M <= 0 || N <= 0) return FOV_OK;
    if ((K & 3) || (lda & 3) || (ldb & 3) || (((uintptr_t)a) & 15) || (((uintptr_t)b) & 15) || K <= 0) {
        set_error("gemm_bf16_nt: K, lda, ldb must be multiples of 4 and the operands 16-byte aligned");
        return FOV_ERR_INVALID;
    }
    if (((long)M * lda + K) * 4 >= (1L << 31) || ((long)N * ldb + K) * 4 >= (1L << 31)) { set_error("gemm_bf16_nt: operand larger than 2 GiB"); return FOV_ERR_UNSUPPORTED; }
    GemmNT g = {};
    g.a = a; g.b = b; g.c = c; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
    hipLaunchKernelGGL(gemm_bf16_nt_kernel, dim3((N + GT - 1) / GT, (M + GT - 1) / GT), dim3(256), 0, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("gemm_bf16_nt launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
