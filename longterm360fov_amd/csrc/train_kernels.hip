// Training-side kernels of the hot path (a6: BPTT + Keras optimizers; mycode/FoV_seq2seq.py:103,
// 112-117 delegate all of it to Keras/TensorFlow autodiff).  Round-1 structure: the backward
// recurrence is stepped from the host (one pointwise launch + one MFMA GEMM launch per time
// step); the weight gradients are large split-K MFMA GEMMs over B*T.  Everything is
// deterministic (no float atomics): split-K partials are summed in a fixed order.
//
// Kernels and the roofline that bounds each:
//   gemm_f32_kernel       fp32 MFMA (v_mfma_f32_16x16x4_f32), LDS-tiled (32|96|128) x (256|128) x 16 - MFMA
//   splitk_reduce_kernel  sums S partial C tiles                                     - HBM
//   lstm_bwd_pointwise    dz_t, dc from the reserve (24 B read + 16 B written/elem)   - HBM
//   colsum_*              bias gradients                                             - HBM
//   mse_dense_grad_kernel dL/d(pre-tanh) of the Dense head + loss partials           - HBM
//   adam_kernel / rmsprop_kernel  flat-buffer optimizer step (16-28 B per parameter) - HBM
#include <stdlib.h>

#include <mutex>
#include <unordered_map>

#include "fov_common.h"
#include "xch_common.h"
#include "bf16_common.h"

namespace fov {

// ---------------------------------------------------------------------------------------
// C[m][n] = sum_k A(m,k) * B(k,n);  k = ko*KI + ki.
//   A(m,k) = a[m*a_sm + ko*a_sko + ki*a_ski],  B(k,n) = b[n*b_sn + ko*b_sko + ki*b_ski]
// The two-level k lets (batch, time) row pairs with a time shift be contracted without copies
// (dR = sum_{b,t} h_{t-1}^T dz_t).  blockIdx.z = split-K slice; slice s writes C + s*M*N when
// `split` > 1 (partials), else C directly.
// ---------------------------------------------------------------------------------------
struct GemmArgs {
    const float* a;
    const float* b;
    float* c;
    int M, N, KO, KI;
    long a_sm, a_sko, a_ski;
    long b_sn, b_sko, b_ski;
    int ldc;
    int split;       // number of K slices (grid.z)
    int tiles_per_split; // k-tiles (of 16, never straddling a ko row) per slice
    int xcd_remap, grid_n, grid_m;   // 1-D grid with slices pinned to XCDs (split >= 8)
    int add_c;       // single-slice accumulate: C += A.B in the epilogue (one writer per element, deterministic)
    // fused weight gradient [dK ; dR ; db] = [x | h_prev | 1]^T dz (one product, one reduce); k-slow operands only
    const float* a2;   // second A operand: rows [M1, M) of C come from it (M1 a multiple of the row tile), or NULL
    long a2_sko, a2_ski;
    int M1;
    int a_period, a2_period;   // T > 0 (KO == 1): k index r of the operand is its row r - 1, zero where r % T == 0 - h_{t-1}
                               // read from the (B,T,H) tape of h_t with the rows kept flat (full 16-row k-tiles)
    int bias_row;      // 1: row M of C = column sums of B (B staged with 16-byte loads)
};

constexpr int GBK = 16;
typedef unsigned gu32x4 __attribute__((ext_vector_type(4)));

// Block tile BM x BN x 16 with BM = 16*MI*WAVES_M, BN = 16*NI*(4/WAVES_M); each of the 4 waves owns
// MI x NI MFMA tiles.  Global->LDS staging goes through registers one k-tile ahead.
//
// fp32 MFMA shares the VALU issue slots, so staging arithmetic is paid for in MFMA time.  The k-tiles
// therefore never straddle a ko row: tile t <-> (ko = t / ntpr, kt = t % ntpr) with ntpr = ceil(KI/16),
// its k = ko*KI + kt*16 + kk.  The tile's base address is wave-uniform (SALU) and sits in a per-tile buffer
// descriptor; each thread adds a 32-bit offset fixed for the whole kernel; invalid elements present an
// out-of-range offset and the hardware returns 0 (<= 15 zero columns per ko row; KI = 29 wastes 9 % of the
// MFMAs and saves far more).
//
// Staging mode of an operand (template parameter, so the staged registers never meet at a control-flow
// merge - a merge makes the compiler wait for the loads before the MFMAs of the current tile):
//   0  k-slow ([k][row] storage, row index contiguous), 16-byte buffer loads, LDS [k][rows]
//   1  k-slow, 4-byte buffer loads (row stride or width not a multiple of 4), LDS [k][rows]
//   2  anything else: global loads, select at stash time, LDS [k][rows]
//   3  k-fast ([row][k] storage, k contiguous, KI % 4 == 0), 16-byte buffer loads along k, LDS [rows][k]
// The MFMA k-slot of lane group lq in step s is k = 4*lq + s (for A and B alike): an operand stored
// [rows][k] gives a lane its four steps with ONE ds_read_b128, an operand stored [k][rows] reads rows
// 4*lq + s (row stride = 4 mod 8 floats: the four lane groups land on disjoint banks).
// Fragments of a [k][rows] tile (round 3): a lane used to read its four k-steps of every 16-row MFMA tile with four
// ds_read_b32 each - 32 LDS instructions per k-tile at 4 x 4 tiles per wave.  The MFMA row <-> operand row assignment is
// free as long as the epilogue uses the same one, so: MFMA tile i, row q of a wave's NTW tiles is operand row
// 64*(i/4) + 4*q + (i%4) for tiles in complete groups of four (one ds_read_b128 at [k][4*li] serves four tiles) and
// 2*q + (i%2) for a trailing pair (ds_read_b64): 8 LDS instructions per k-tile, and a lane's four column tiles are four
// CONSECUTIVE output columns (16-byte stores).  Row stride: a multiple of 16 floats keeps the b128 lane groups on distinct
// 16-byte slots, == 8 (mod 16) the two halves of a b64 read on distinct banks.  Same products in the same order: results
// are bit-identical to the old mapping.
template <int ROWS, int MODE, int NTW>
struct OperandStage {
    static_assert(NTW % 2 == 0, "tiles per wave: complete groups of four plus at most one pair");
    static constexpr bool VEC = (MODE == 0 || MODE == 3);
    static constexpr bool TR = (MODE == 3);
    static constexpr int LD = TR ? 24 : ROWS + (NTW % 4 == 0 ? 0 : 8);
    static constexpr int LDS_FLOATS = TR ? ROWS * 24 : GBK * LD;   // one buffer
    static constexpr int R = VEC ? (ROWS * 4 + 255) / 256 : ROWS / 16;     // loads per thread per k-tile
    static constexpr unsigned OOR = 0x80000000u;
    int kk[R], rr[R];
    unsigned off[R];   // byte offset from the tile base (OOR when the row is outside the matrix)
    bool ok[R];
    float sc[VEC ? 1 : R];
    f32x4 vv[VEC ? R : 1];
    int kmax;          // valid k columns of the staged tile
    int ph[R];         // periodic-shift operands: (k row of this thread's element) % period

    __device__ __forceinline__ void init(int tid, int row0, int nrows, long s_row, long s_ki) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int e = tid + 256 * r;
            if (MODE == 0) { kk[r] = e / (ROWS / 4); rr[r] = 4 * (e - kk[r] * (ROWS / 4)); }
            else if (MODE == 3) { rr[r] = e >> 2; kk[r] = 4 * (e & 3); }
            else if (MODE == 2 && s_ki == 1) { rr[r] = e >> 4; kk[r] = e & 15; }   // consecutive threads walk k
            else { kk[r] = e / ROWS; rr[r] = e - kk[r] * ROWS; }
            ok[r] = rr[r] < ROWS && kk[r] < GBK && row0 + rr[r] < nrows;
            off[r] = ok[r] ? (unsigned)(((long)rr[r] * s_row + (long)kk[r] * s_ki) * 4) : OOR;
        }
        kmax = 0;
    }
    // issue the loads of one tile (base = first element of the tile's first row / k)
    __device__ __forceinline__ void init_phase(long k0, int period) {
#pragma unroll
        for (int r = 0; r < R; ++r) ph[r] = period > 0 ? (int)((k0 + kk[r]) % period) : 1;
    }
    // period > 0 (k-slow modes only): elements whose k row is a multiple of the period read as zero (`base` already points
    // one row back); inc = GBK % period advances the phases to the next tile
    __device__ __forceinline__ void fetch(const float* base, int tile_kmax, long s_ki, int period = 0, int inc = 0) {
        kmax = tile_kmax;
        if constexpr (MODE == 2) {
#pragma unroll
            for (int r = 0; r < R; ++r) sc[r] = base[(ok[r] && kk[r] < tile_kmax) ? (off[r] >> 2) : 0u];
        } else {
            const int krows = tile_kmax < GBK ? tile_kmax : GBK;
            // k-slow: the descriptor ends after the last valid k row; k-fast: per-lane k test (below)
            const int nrec = TR ? 0x7fffffff : krows * (int)s_ki * 4;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, nrec, 0x00020000);
#pragma unroll
            for (int r = 0; r < R; ++r) {
                unsigned o = (TR && kk[r] >= tile_kmax) ? OOR : off[r];
                if (!TR && period > 0) {
                    o = ph[r] == 0 ? OOR : o;
                    ph[r] += inc;
                    ph[r] -= ph[r] >= period ? period : 0;
                }
                if constexpr (VEC) {
                    const gu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, o, 0, 0);
                    vv[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
                } else {
                    sc[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, o, 0, 0));
                }
            }
        }
    }
    __device__ __forceinline__ void stash(float* lds) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if constexpr (MODE == 3) {
                if (rr[r] < ROWS) *(f32x4*)(lds + rr[r] * LD + kk[r]) = vv[r];
            } else if constexpr (MODE == 0) {
                if (kk[r] < GBK) *(f32x4*)(lds + kk[r] * LD + rr[r]) = vv[r];
            } else if constexpr (MODE == 1) {
                lds[kk[r] * LD + rr[r]] = sc[r];
            } else {
                lds[kk[r] * LD + rr[r]] = (ok[r] && kk[r] < kmax) ? sc[r] : 0.f;   // select here, after the MFMAs
            }
        }
    }
    // operand row (relative to the wave's first row) that MFMA tile i, row q stands for
    static __device__ __forceinline__ int row_of(int i, int q) {
        if constexpr (TR) return 16 * i + q;
        else return i < 4 * (NTW / 4) ? 64 * (i / 4) + 4 * q + (i & 3) : 64 * (NTW / 4) + 2 * q + (i & 1);
    }
    // fragments of the wave's NTW tiles (rows from `row0`) for this lane: out[i][s] = element (row_of(i, li), k = 4*lq + s)
    static __device__ __forceinline__ void frags(const float* lds, int row0, int li, int lq, f32x4 (&out)[NTW]) {
        if constexpr (TR) {
#pragma unroll
            for (int i = 0; i < NTW; ++i) out[i] = *(const f32x4*)(lds + (row0 + 16 * i + li) * LD + 4 * lq);
        } else {
#pragma unroll
            for (int s_ = 0; s_ < 4; ++s_) {
                const float* p = lds + (4 * lq + s_) * LD + row0;
#pragma unroll
                for (int g_ = 0; g_ < NTW / 4; ++g_) {
                    const f32x4 v = *(const f32x4*)(p + 64 * g_ + 4 * li);
                    out[4 * g_][s_] = v[0]; out[4 * g_ + 1][s_] = v[1]; out[4 * g_ + 2][s_] = v[2]; out[4 * g_ + 3][s_] = v[3];
                }
                if constexpr (NTW % 4 == 2) {
                    typedef float f32x2_ __attribute__((ext_vector_type(2)));
                    const f32x2_ v = *(const f32x2_*)(p + 64 * (NTW / 4) + 2 * li);
                    out[NTW - 2][s_] = v[0]; out[NTW - 1][s_] = v[1];
                }
            }
        }
    }
};

template <int MI, int NI, int WAVES_M, int AMODE, int BMODE>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmArgs g) {
    constexpr int WAVES_N = 4 / WAVES_M;
    constexpr int BM = 16 * MI * WAVES_M, BN = 16 * NI * WAVES_N;
    using StA = OperandStage<BM, AMODE, MI>;
    using StB = OperandStage<BN, BMODE, NI>;
    __shared__ __attribute__((aligned(16))) float As[2][StA::LDS_FLOATS];
    __shared__ __attribute__((aligned(16))) float Bs[2][StB::LDS_FLOATS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // Block -> (n tile, m tile, K slice).  Workgroups are dealt round-robin to the 8 XCDs, each with its own
    // L2.  With split-K the blocks of one slice read the same K range of A and B, so a slice is kept on one
    // XCD (1-D grid, xcd = id % 8): its operands cross the fabric once instead of once per XCD.
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
    const int m0 = by * BM, n0 = bx * BN;
    const int ntpr = (g.KI + GBK - 1) / GBK;
    const int tiles_total = g.KO * ntpr;
    const int tbeg = bz * g.tiles_per_split;
    int tend = tbeg + g.tiles_per_split;
    if (tend > tiles_total) tend = tiles_total;
    const int wm = (wave / WAVES_N) * 16 * MI, wn = (wave % WAVES_N) * 16 * NI;
    const int li = lane & 15, lq = lane >> 4;

    f32x4 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    StA sa;
    StB sb;
    // which A operand this row tile reads (block-uniform)
    const bool second = g.a2 != nullptr && m0 >= g.M1;
    const int am0 = second ? m0 - g.M1 : m0;
    const long a_sko = second ? g.a2_sko : g.a_sko, a_ski = second ? g.a2_ski : g.a_ski;
    const int a_period = second ? g.a2_period : g.a_period;
    const int a_inc = a_period > 0 ? GBK % a_period : 0;
    sa.init(tid, am0, second ? g.M - g.M1 : (g.a2 ? g.M1 : g.M), g.a_sm, a_ski);
    sa.init_phase((long)tbeg * GBK, a_period);
    sb.init(tid, n0, g.N, g.b_sn, g.b_ski);
    const float* a_blk = (second ? g.a2 : g.a) + (long)am0 * g.a_sm;
    const float* b_blk = g.b + (long)n0 * g.b_sn;
    int f_ko = tbeg / ntpr, f_kt = tbeg - f_ko * ntpr;   // the tile the next fetch() loads (wave-uniform)
    int f_left = tend - tbeg;   // tiles not fetched yet
    auto fetch = [&]() {
        const int kmax = g.KI - f_kt * GBK;   // >= 16 except for the row's last tile
        // The shifted operand's phases advance only towards a tile that is really fetched next: the extra fetch of the final
        // iteration re-reads the LAST tile and must mask the same elements again.  (Round 4: with advanced phases it read
        // element k = 0 for real - the row BEFORE the tensor, `base` points one row back - whenever the slice's only tile
        // starts at k = 0, i.e. batch x time <= 16: a memory access fault when the tape begins a mapped region.)
        const int inc_now = f_left > 1 ? a_inc : 0;
        sa.fetch(a_blk + (long)f_ko * a_sko + (long)(f_kt * GBK - (a_period > 0 ? 1 : 0)) * a_ski, kmax, a_ski, a_period, inc_now);
        sb.fetch(b_blk + (long)f_ko * g.b_sko + (long)(f_kt * GBK) * g.b_ski, kmax, g.b_ski);
        // advance (scalar state only); after the last tile the position stays, so the extra fetch of the final
        // iteration re-reads that tile instead of branching around the loads
        if (--f_left > 0 && ++f_kt == ntpr) { f_kt = 0; ++f_ko; }
    };
    // Per tile: all fragment reads first (one exposed LDS latency per tile instead of one per k-step), then the
    // global loads of the next tile, then 16*MI*NI/4 back-to-back MFMAs, then the staged tile goes to the other
    // LDS buffer.  No branch separates loads, MFMAs and LDS writes (a merge would make the compiler wait for the
    // loads early): the final iteration simply stages its own tile once more, unused.
    f32x4 af[MI], bf[NI];
    auto read_frags = [&](int buf) {
        StA::frags(As[buf], wm, li, lq, af);
        StB::frags(Bs[buf], wn, li, lq, bf);
    };
    auto mfmas = [&]() {
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][ks], bf[j][ks], acc[i][j], 0, 0, 0);
    };
    // bias row: column sums of B from the staged registers (row tile 0 only; B in 16-byte mode, host-checked)
    const bool bias_blk = g.bias_row && by == 0;
    f32x4 bsum[StB::VEC ? StB::R : 1];
#pragma unroll
    for (int r = 0; r < (StB::VEC ? StB::R : 1); ++r) bsum[r] = (f32x4){0.f, 0.f, 0.f, 0.f};
    auto add_bias = [&]() {
        if constexpr (BMODE == 0) {
#pragma unroll
            for (int r = 0; r < StB::R; ++r) bsum[r] += sb.vv[r];
        }
    };
    int buf = 0;
    if (tbeg < tend) {
        fetch();
        sa.stash(As[0]);
        sb.stash(Bs[0]);
        if (bias_blk) add_bias();
    }
    __syncthreads();
    for (int t = tbeg; t < tend; ++t) {
        read_frags(buf);
        __builtin_amdgcn_sched_barrier(0);
        fetch();
        __builtin_amdgcn_sched_barrier(0);
        mfmas();
        __builtin_amdgcn_sched_barrier(0);
        sa.stash(As[buf ^ 1]);
        sb.stash(Bs[buf ^ 1]);
        if (bias_blk && t + 1 < tend) add_bias();   // the final iteration re-stages its own tile: not summed twice
        __syncthreads();
        buf ^= 1;
    }
    float* c = g.c + (g.split > 1 ? (size_t)bz * (g.M + g.bias_row) * g.ldc : 0);
    if constexpr (BMODE == 0) {
        if (bias_blk) {   // fold the k rows of the staging map (kk = e / (BN/4)) in a fixed order through LDS
            float* red = Bs[0];
            constexpr int KR = 256 * StB::R / (BN / 4);   // distinct k rows a column quad is staged by
            static_assert(KR * BN <= 2 * StB::LDS_FLOATS, "bias reduction scratch");
#pragma unroll
            for (int r = 0; r < StB::R; ++r)
                if (sb.kk[r] < KR && sb.rr[r] < BN) *(f32x4*)(red + sb.kk[r] * BN + sb.rr[r]) = bsum[r];
            __syncthreads();
            if (tid < BN && n0 + tid < g.N) {
                float t = 0.f;
                for (int q = 0; q < KR; ++q) t += red[q * BN + tid];
                float* cp = c + (size_t)g.M * g.ldc + n0 + tid;
                *cp = g.add_c ? *cp + t : t;
            }
        }
    }
    // k-slow B with four column tiles per wave: a lane's tiles j = 0..3 are four consecutive columns -> 16-byte stores
    const bool vec_out = !StB::TR && NI == 4 && (g.ldc & 3) == 0 && (((uintptr_t)c) & 15) == 0;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int m = m0 + wm + StA::row_of(i, lq * 4 + r);
            if (m >= g.M) continue;
            if constexpr (!StB::TR && NI == 4) {
                const int nb = n0 + wn + 4 * li;
                if (vec_out && nb + 3 < g.N) {
                    f32x4* cp = (f32x4*)(c + (size_t)m * g.ldc + nb);
                    f32x4 v = {acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]};
                    if (g.add_c) v += *cp;
                    *cp = v;
                    continue;
                }
            }
#pragma unroll
            for (int j = 0; j < NI; ++j) {
                const int n = n0 + wn + StB::row_of(j, li);
                if (n < g.N) {
                    float* cp = c + (size_t)m * g.ldc + n;
                    *cp = g.add_c ? *cp + acc[i][j][r] : acc[i][j][r];
                }
            }
        }
}

// out[i] = (accumulate ? out[i] : 0) + sum_s part[s][i].  Block = 64 outputs x 4 slice groups: group q adds
// slices q, q+4, q+8, ... (eight loads in flight per thread), then the four group sums are folded in a fixed
// order through LDS - deterministic for a given (n, S), and short products with many slices are not
// serialised on one thread per output.
// VEC: four consecutive outputs per lane through 16-byte loads (n % 4 == 0, 16-byte aligned slices and output) - the
// batched reduce of a training step reads ~80 MB of slices and ran at 1.8 TB/s with 4-byte loads.  The order of additions
// per output is the same in both forms.
template <bool VEC>
__device__ __forceinline__ void splitk_reduce_body(const float* __restrict__ part, float* __restrict__ out, long n, int S,
                                                   int accumulate, long block) {
    constexpr int W = VEC ? 4 : 1;
    __shared__ float red[4][64 * W];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const long i = (block * 64 + li) * W;
    float s[W];
#pragma unroll
    for (int w = 0; w < W; ++w) s[w] = 0.f;
    if (i < n) {
        int k = q;
        for (; k + 28 < S; k += 32) {
            float v[8][W];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                if constexpr (VEC) {
                    const f32x4 t = *(const f32x4*)(part + (size_t)(k + 4 * u) * n + i);
                    v[u][0] = t[0]; v[u][1] = t[1]; v[u][2] = t[2]; v[u][3] = t[3];
                } else {
                    v[u][0] = part[(size_t)(k + 4 * u) * n + i];
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int w = 0; w < W; ++w) s[w] += v[u][w];
        }
        for (; k < S; k += 4) {
            if constexpr (VEC) {
                const f32x4 t = *(const f32x4*)(part + (size_t)k * n + i);
                s[0] += t[0]; s[1] += t[1]; s[2] += t[2]; s[3] += t[3];
            } else {
                s[0] += part[(size_t)k * n + i];
            }
        }
    }
#pragma unroll
    for (int w = 0; w < W; ++w) red[q][li * W + w] = s[w];
    __syncthreads();
    if (q == 0 && i < n) {
        float t[W];
#pragma unroll
        for (int w = 0; w < W; ++w) t[w] = ((red[0][li * W + w] + red[1][li * W + w]) + (red[2][li * W + w] + red[3][li * W + w]));
        if constexpr (VEC) {
            f32x4* op = (f32x4*)(out + i);
            f32x4 r = {t[0], t[1], t[2], t[3]};
            if (accumulate) r += *op;
            *op = r;
        } else {
            out[i] = accumulate ? out[i] + t[0] : t[0];
        }
    }
}
static inline bool splitk_reduce_vec(const float* part, const float* out, long n) {
    return (n & 3) == 0 && (((uintptr_t)part) & 15) == 0 && (((uintptr_t)out) & 15) == 0;
}
template <bool VEC>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ part, float* __restrict__ out,
                                                            long n, int S, int accumulate) {
    splitk_reduce_body<VEC>(part, out, n, S, accumulate, (long)blockIdx.x);
}

static inline dim3 splitk_reduce_grid(size_t n, bool vec = false) { return dim3((unsigned)((n + (vec ? 255 : 63)) / (vec ? 256 : 64))); }

// ---------------------------------------------------------------------------------------
// Deferred split reductions (round 3).  A training step runs 7-8 split products whose partial slices each needed their
// own reduce launch (4.5-9 us apiece, 30-55 us per step: 11 % of model.fit's step at the reference's batch of 32).
// Between fov_reduce_defer_begin and fov_reduce_defer_flush a product whose output lies inside the registered gradient
// buffer writes its slices into the caller's ARENA instead of the call's scratch and only RECORDS {slices, out, n, S,
// accumulate}; the flush sums all recorded products in ONE launch, with the arithmetic of splitk_reduce_kernel (same
// order: bit-identical results).  Any later write of a producer into a range that a pending record covers flushes first,
// so in-stream order is preserved; code that touches the gradient buffer by other means must flush itself
// (training.FlatParamTrainer does, before the optimizer / the all-reduce / a gradient rescale).
// ---------------------------------------------------------------------------------------
static int check_launch(const char* what);

constexpr int kDeferMax = 16;
struct DeferEntry { const float* part; float* out; long n; int S; int accumulate; int block0; int vec; };
struct DeferTable { int count; int blocks; DeferEntry e[kDeferMax]; };

__global__ __launch_bounds__(256) void splitk_reduce_batch_kernel(DeferTable t) {
    int cur = 0;
#pragma unroll 1
    for (int i = 1; i < t.count; ++i)
        if ((int)blockIdx.x >= t.e[i].block0) cur = i;
    const DeferEntry e = t.e[cur];
    if (e.vec) splitk_reduce_body<true>(e.part, e.out, e.n, e.S, e.accumulate, (long)((int)blockIdx.x - e.block0));
    else splitk_reduce_body<false>(e.part, e.out, e.n, e.S, e.accumulate, (long)((int)blockIdx.x - e.block0));
}

// The deferral state is keyed by the registered gradient buffer: every trainer (thread, stream, device) that opened a region
// with fov_reduce_defer_begin owns ONE entry of the registry below - its arena, its record table - and a product finds its
// region by the address of its output.  Regions never overlap (begin replaces an overlapping one after flushing it), so two
// trainers on two threads or two GPUs of one process never see each other's arena or records; the registry itself is guarded
// by one mutex.
namespace {
struct DeferState {
    const float* gbase = nullptr; const float* gend = nullptr;
    float* arena = nullptr; size_t arena_floats = 0, used = 0;
    DeferTable table = {};
};
constexpr int kDeferRegions = 16;     // open regions per process (one per trainer in flight; begin fails beyond that)
DeferState g_defer[kDeferRegions];
int g_defer_open = 0;
std::mutex g_defer_mu;

DeferState* defer_region_of(const float* out, size_t n) {       // the region that holds [out, out + n), or NULL
    for (int i = 0; i < g_defer_open; ++i)
        if (out >= g_defer[i].gbase && out + n <= g_defer[i].gend) return &g_defer[i];
    return nullptr;
}
bool defer_overlaps(const DeferState& st, const float* out, size_t n) {
    for (int i = 0; i < st.table.count; ++i) {
        const DeferEntry& e = st.table.e[i];
        if (out < e.out + e.n && e.out < out + n) return true;
    }
    return false;
}
int defer_flush_locked(DeferState& st, hipStream_t stream) {
    DeferTable& t = st.table;
    if (t.count == 0) return FOV_OK;
    hipLaunchKernelGGL(splitk_reduce_batch_kernel, dim3((unsigned)t.blocks), dim3(256), 0, stream, t);
    t.count = 0; t.blocks = 0;
    // the arena is NOT rewound here: a caller may flush on one stream (its products run on a side stream) and go on recording
    // on another - slices are written once per begin ... end, what does not fit any more is reduced at once
    return check_launch("splitk_reduce_batch");
}
void defer_close_locked(int i) {
    g_defer[i] = g_defer[g_defer_open - 1];
    g_defer[--g_defer_open] = DeferState();
}
}  // namespace

int defer_begin(float* grad_base, size_t grad_floats, float* arena, size_t arena_floats, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    if (!grad_base || grad_floats == 0) return FOV_OK;
    // a region over (part of) the same buffer is replaced: its pending records go first
    for (int i = g_defer_open - 1; i >= 0; --i)
        if (grad_base < g_defer[i].gend && g_defer[i].gbase < grad_base + grad_floats) {
            int rc = defer_flush_locked(g_defer[i], stream);
            defer_close_locked(i);
            if (rc) return rc;
        }
    if (!arena || arena_floats == 0) return FOV_OK;
    if (g_defer_open == kDeferRegions) { set_error("fov_reduce_defer_begin: too many open regions (16 per process)"); return FOV_ERR_UNSUPPORTED; }
    DeferState& st = g_defer[g_defer_open++];
    st = DeferState();
    st.gbase = grad_base; st.gend = grad_base + grad_floats;
    st.arena = arena; st.arena_floats = arena_floats;
    return FOV_OK;
}
// grad_base = the buffer a region was opened on (any address inside it); NULL = every open region of the process
int defer_flush(const float* grad_base, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    int rc = FOV_OK;
    for (int i = 0; i < g_defer_open; ++i)
        if (!grad_base || (grad_base >= g_defer[i].gbase && grad_base < g_defer[i].gend))
            if (int r = defer_flush_locked(g_defer[i], stream)) rc = r;
    return rc;
}
int defer_end(const float* grad_base, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    int rc = FOV_OK;
    for (int i = g_defer_open - 1; i >= 0; --i)
        if (!grad_base || (grad_base >= g_defer[i].gbase && grad_base < g_defer[i].gend)) {
            if (int r = defer_flush_locked(g_defer[i], stream)) rc = r;
            defer_close_locked(i);
        }
    return rc;
}
// A producer is about to write [out, out + n) (directly, or through an immediate reduce): pending records over that range
// go first.
int defer_touch(const float* out, size_t n, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    for (int i = 0; i < g_defer_open; ++i) {
        DeferState& st = g_defer[i];
        if (out < st.gend && st.gbase < out + n && st.table.count > 0 && defer_overlaps(st, out, n))
            if (int rc = defer_flush_locked(st, stream)) return rc;
    }
    return FOV_OK;
}
// Where a split product with output [out, out + n) may put its `floats` of partial slices, or NULL: reduce at once.
float* defer_alloc(const float* out, size_t n, size_t floats, hipStream_t stream) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    DeferState* st = defer_region_of(out, n);
    if (!st) return nullptr;
    if (st->table.count > 0 && defer_overlaps(*st, out, n)) { if (defer_flush_locked(*st, stream)) return nullptr; }
    const size_t need = (floats + 63) & ~(size_t)63;
    if (st->table.count == kDeferMax && defer_flush_locked(*st, stream)) return nullptr;
    if (st->used + need > st->arena_floats) return nullptr;   // arena exhausted for this step: reduce at once
    float* p = st->arena + st->used;
    st->used += need;
    return p;
}
// (only behind a successful defer_alloc for the same output: the region exists and has a free record)
void defer_record(const float* part, float* out, long n, int S, int accumulate) {
    std::lock_guard<std::mutex> lock(g_defer_mu);
    DeferState* st = defer_region_of(out, (size_t)n);
    if (!st) return;
    DeferTable& t = st->table;
    DeferEntry& e = t.e[t.count++];
    e.part = part; e.out = out; e.n = n; e.S = S; e.accumulate = accumulate; e.block0 = t.blocks;
    e.vec = splitk_reduce_vec(part, out, n) ? 1 : 0;
    t.blocks += (int)splitk_reduce_grid((size_t)n, e.vec != 0).x;
}
// the common tail of a split product: record (slices already in the arena) or reduce now
int reduce_or_defer(bool deferred, const float* part, float* out, long n, int S, int accumulate, hipStream_t stream, const char* what) {
    if (deferred) { defer_record(part, out, n, S, accumulate); return FOV_OK; }
    int rc = defer_touch(out, (size_t)n, stream);
    if (rc) return rc;
    if (splitk_reduce_vec(part, out, n))
        hipLaunchKernelGGL(splitk_reduce_kernel<true>, splitk_reduce_grid((size_t)n, true), dim3(256), 0, stream, part, out, n, S, accumulate);
    else
        hipLaunchKernelGGL(splitk_reduce_kernel<false>, splitk_reduce_grid((size_t)n), dim3(256), 0, stream, part, out, n, S, accumulate);
    return check_launch(what);
}

// ---------------------------------------------------------------------------------------
// BPTT pointwise step t (Keras LSTMCell backward):
//   dh = dhs[:,t] + dh_rec;  tc = tanh(c_t);  do = dh*tc;  dc += dh*o*(1-tc^2)
//   dz = [dc*g*s'(i), dc*c_{t-1}*s'(f), dc*i*(1-g^2), do*s'(o)];  dc *= f
// s'(a) = a(1-a) (sigmoid) or 0.2*[0<a<1] (hard_sigmoid).  dh_rec / dc live in (B,H) buffers.
// ---------------------------------------------------------------------------------------
template <int ACT>
__device__ __forceinline__ float rec_act_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

template <int ACT>
__global__ __launch_bounds__(256) void lstm_bwd_pointwise(const float* __restrict__ reserve, const float* __restrict__ c0,
                                                          const float* __restrict__ dhs, const float* __restrict__ dh_rec,
                                                          float* __restrict__ dc, float* __restrict__ dz, int B, int T,
                                                          int H, int t) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)B * H) return;
    const int b = (int)(idx / H), j = (int)(idx - (long)b * H);
    const float* rp = reserve + (((size_t)b * T + t) * 5) * H + j;
    const float i = rp[0], f = rp[H], g = rp[2 * H], o = rp[3 * H], c = rp[4 * H];
    const float cprev = (t > 0) ? rp[4 * H - 5 * (long)H] : (c0 ? c0[idx] : 0.f);
    float dh = dh_rec[idx];
    if (dhs) dh += dhs[((size_t)b * T + t) * H + j];
    const float tc = tanh_f(c);
    const float dov = dh * tc;
    const float dcv = dc[idx] + dh * o * (1.f - tc * tc);
    float* zp = dz + ((size_t)b * T + t) * 4 * H + j;
    zp[0] = dcv * g * rec_act_grad<ACT>(i);
    zp[H] = dcv * cprev * rec_act_grad<ACT>(f);
    zp[2 * H] = dcv * i * (1.f - g * g);
    zp[3 * H] = dov * rec_act_grad<ACT>(o);
    dc[idx] = dcv * f;
}

// partial column sums: block (x = column block of 256, y = row chunk) -> part[y][col]
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ x, float* __restrict__ part,
                                                             long rows, int cols, long rows_per_chunk) {
    const int col = blockIdx.x * 256 + threadIdx.x;
    if (col >= cols) return;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    long r = r0;
    for (; r + 4 <= r1; r += 4) {
        s0 += x[(size_t)r * cols + col];
        s1 += x[(size_t)(r + 1) * cols + col];
        s2 += x[(size_t)(r + 2) * cols + col];
        s3 += x[(size_t)(r + 3) * cols + col];
    }
    for (; r < r1; ++r) s0 += x[(size_t)r * cols + col];
    part[(size_t)blockIdx.y * cols + col] = (s0 + s1) + (s2 + s3);
}

// Dense(tanh|linear) + Keras mean_squared_error: dpre = 2 (y - target) / n * act'(y); per-block
// partial sums of (y - target)^2 into loss_part[blockIdx.x].
// Tickets of the loss kernels' "last block adds the partials" step (round 4: the separate one-block sum launch is gone, 5 us of
// every training step's critical path).  Zero at module load, left at zero again by the block that takes the last ticket; every
// stream of a device has its own slot (loss_ticket_of), so launches in flight never share one.
constexpr int kLossTickets = 1024;
__device__ unsigned g_loss_tickets[kLossTickets];

__global__ __launch_bounds__(256) void mse_dense_grad_kernel(const float* __restrict__ y, const float* __restrict__ target,
                                                             float* __restrict__ dpre, float* __restrict__ loss_part,
                                                             long n, float scale, int activation, int tmB, int tmT, int O,
                                                             unsigned* __restrict__ ticket, float* __restrict__ loss_out, float loss_scale,
                                                             float* __restrict__ col_part = nullptr, float* __restrict__ db_out = nullptr,
                                                             int dbO = 0) {
    // tmT > 0: y / dpre are time-major (T,B,O) against a batch-major target (B,T,O) - the unrolled decoders keep
    // their tape time-major, no transposed copies
    // dbO > 0 (batch-major only, with a ticket): the Dense bias gradient db[o] = sum over rows of dpre[row][o] as well - per-block
    // column sums in a fixed order, added over the blocks (in block order) by the block that takes the last ticket: the
    // column-sum launches of dense_bwd (7.6 us at the reference's batch) are gone
    __shared__ float red[256];
    __shared__ float gval[256];
    __shared__ int is_last;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float sq = 0.f;
    if (dbO > 0) gval[threadIdx.x] = 0.f;
    if (i < n) {
        long ti = i;
        if (tmT > 0) {   // i = (t*B + b)*O + o  ->  (b*T + t)*O + o
            const long row = i / O;
            const int o = (int)(i - row * O);
            const long t = row / tmB, b = row - t * tmB;
            ti = (b * tmT + t) * O + o;
        }
        const float yv = y[i], d = yv - target[ti];
        sq = d * d;
        float gsc = 2.f * d * scale;
        if (activation == 1) gsc *= (1.f - yv * yv);
        dpre[i] = gsc;
        if (dbO > 0) gval[threadIdx.x] = gsc;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sq;
    __syncthreads();
    __shared__ float cpart[32][8];
    const int co = threadIdx.x & 7, cq = threadIdx.x >> 3;   // column, one of 32 strands of it
    if (dbO > 0) {
        // element j of the block is i0 + j, its column (i0 + j) % dbO: strand cq of column co takes every 32nd of them, the
        // 32 strand sums are added in strand order (a fixed order: the same bits every time)
        const int c0 = (int)(((long)blockIdx.x * 256) % dbO);
        float a = 0.f;
        if (co < dbO)
            for (int j = (co - c0 + dbO) % dbO + dbO * cq; j < 256; j += dbO * 32) a += gval[j];
        cpart[cq][co] = a;
        __syncthreads();
        if (threadIdx.x < 8) {
            float sum = 0.f;
            for (int q = 0; q < 32; ++q) sum += cpart[q][threadIdx.x];
            gval[threadIdx.x] = sum;   // (every strand has been read: the barrier above; columns >= dbO are zero)
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float part = red[0] + red[1] + red[2] + red[3];
        if (ticket) {
            __hip_atomic_store(loss_part + blockIdx.x, part, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            // the block's column sums leave through the SAME thread and fence (a fence per storing thread made the kernel
            // 32 us at 720 blocks where it had been 4.8)
            // two 16-byte stores (the fence below publishes them; col_part rows are 32 bytes, the area 16-byte aligned)
            if (dbO > 0) {
                f32x4* cp = (f32x4*)(col_part + (size_t)blockIdx.x * 8);
                cp[0] = (f32x4){gval[0], gval[1], gval[2], gval[3]};
                cp[1] = (f32x4){gval[4], gval[5], gval[6], gval[7]};
            }
            __threadfence();
            is_last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
        } else {
            loss_part[blockIdx.x] = part;
            is_last = 0;
        }
    }
    __syncthreads();
    if (is_last) {   // the arithmetic of sum_scale_kernel (same order: bit-identical loss), by the block that finished last
        __threadfence();
        if (dbO > 0) {   // (block-uniform) the blocks' column sums: 32 strands per column over the blocks, then in strand order
            // (L1-bypassing buffer loads, eight in flight: agent-scope atomic loads are issued one round trip at a time - 23 of them
            // per lane made this kernel 36 us at 720 blocks)
            const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(col_part, 0, (int)(gridDim.x * 8u * 4u), 0x00020000);
            float a = 0.f;
            for (int b0 = cq; b0 < (int)gridDim.x; b0 += 32 * 8) {
                float v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u)   // blocks beyond the grid: offset past the descriptor, reads as 0
                    v[u] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, (unsigned)(((b0 + 32 * u) * 8 + co) * 4), 0, 16 /* sc1 */));
#pragma unroll
                for (int u = 0; u < 8; ++u) a += v[u];
            }
            if (co >= dbO) a = 0.f;
            cpart[cq][co] = a;
            __syncthreads();
            if ((int)threadIdx.x < dbO) {
                float sum = 0.f;
                for (int q = 0; q < 32; ++q) sum += cpart[q][threadIdx.x];
                db_out[threadIdx.x] = sum;
            }
            __syncthreads();
        }
        float a = 0.f;
        for (int i = threadIdx.x; i < (int)gridDim.x; i += 256) a += __hip_atomic_load(loss_part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[threadIdx.x] = a;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            loss_out[0] = red[0] * loss_scale;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the slot's next user finds it at zero
        }
    }
}

// x *= s (fallback gradient weighting of the trainers that do not fold the weight into their loss kernel)
__global__ __launch_bounds__(256) void scale_kernel(float* __restrict__ x, long n, float s) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < n) x[i] *= s;
}

__global__ __launch_bounds__(256) void sum_scale_kernel(const float* __restrict__ part, float* __restrict__ out, int n,
                                                        float scale) {
    __shared__ float red[256];
    float s = 0.f;
    for (int i = threadIdx.x; i < n; i += 256) s += part[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if (threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// out = base + dy * act'(y)   (act' = 1 - y^2 for tanh, [y > 0] for relu, y for exp, 1 for linear); base may be NULL; out may alias base/dy
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                      const float* base, float* out, long n, int activation) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float yv = y[i];
    const float d = dy[i] * (activation == 1 ? (1.f - yv * yv) : (activation == 2 ? (yv > 0.f ? 1.f : 0.f) : (activation == 3 ? yv : 1.f)));
    out[i] = (base ? base[i] : 0.f) + d;
}

// Keras-2.2 optimizers on one flat buffer (oracle/fov_oracle.py::adam_step / rmsprop_step)
// guard0..2 (each may be NULL): sticky timeout words of the workspaces the step's persistent kernels used
// (xch_common.h).  If one is set the gradients are garbage: the update is skipped on the device - no host sync - and
// the parameters stay as they were until fov_check_status reports the failure.
__device__ __forceinline__ bool optimizer_poisoned(const unsigned* g0, const unsigned* g1, const unsigned* g2) {
    return (g0 && *g0 != 0u) || (g1 && *g1 != 0u) || (g2 && *g2 != 0u);
}

// data parallelism: this rank's guard words as ONE float slot of the gradient buffer, so that the SUM all-reduce of that
// buffer tells every rank whether ANY rank's step failed and all of them skip (or apply) the update together
__global__ void guard_flag_kernel(const unsigned* g0, const unsigned* g1, const unsigned* g2, float* out) {
    *out = optimizer_poisoned(g0, g1, g2) ? 1.f : 0.f;
}
int guard_flag(const unsigned* const* guards, float* out, hipStream_t stream) {
    hipLaunchKernelGGL(guard_flag_kernel, dim3(1), dim3(1), 0, stream, guards[0], guards[1], guards[2], out);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("guard_flag launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps,
                                                   const unsigned* g0, const unsigned* g1, const unsigned* g2, long long* applied) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || optimizer_poisoned(g0, g1, g2)) return;
    if (i == 0 && applied) *applied += 1;      // updates that really ran: what the host's step counter is set back to after a failure
    const float gi = g[i];
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    p[i] = p[i] - lr_t * mi / (sqrtf(vi) + eps);
}

// Four parameters per thread, 16-byte accesses (round 4: the one-element form moved 47 MB in 11.4 us at config 3's 1.68 M
// parameters - a quarter of the waves, a quarter of the memory instructions).  Same arithmetic per element.
__global__ __launch_bounds__(256) void adam_kernel4(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, long n, float lr_t, float b1, float b2, float eps,
                                                    const unsigned* g0, const unsigned* g1, const unsigned* g2, long long* applied) {
    const long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i >= n || optimizer_poisoned(g0, g1, g2)) return;
    if (i == 0 && applied) *applied += 1;
    if (i + 3 < n) {
        const f32x4 gi = *(const f32x4*)(g + i), m0 = *(const f32x4*)(m + i), v0 = *(const f32x4*)(v + i), p0 = *(const f32x4*)(p + i);
        f32x4 mi, vi, pi;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            mi[e] = b1 * m0[e] + (1.f - b1) * gi[e];
            vi[e] = b2 * v0[e] + (1.f - b2) * gi[e] * gi[e];
            pi[e] = p0[e] - lr_t * mi[e] / (sqrtf(vi[e]) + eps);
        }
        *(f32x4*)(m + i) = mi;
        *(f32x4*)(v + i) = vi;
        *(f32x4*)(p + i) = pi;
    } else {
        for (long k = i; k < n; ++k) {
            const float gk = g[k];
            const float mk = b1 * m[k] + (1.f - b1) * gk;
            const float vk = b2 * v[k] + (1.f - b2) * gk * gk;
            m[k] = mk;
            v[k] = vk;
            p[k] = p[k] - lr_t * mk / (sqrtf(vk) + eps);
        }
    }
}

__global__ __launch_bounds__(256) void rmsprop_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ a,
                                                      long n, float lr, float rho, float eps, const unsigned* g0,
                                                      const unsigned* g1, const unsigned* g2, long long* applied) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || optimizer_poisoned(g0, g1, g2)) return;
    if (i == 0 && applied) *applied += 1;
    const float gi = g[i];
    const float ai = rho * a[i] + (1.f - rho) * gi * gi;
    a[i] = ai;
    p[i] = p[i] - lr * gi / (sqrtf(ai) + eps);
}

// y = act(x): 1 tanh, 2 relu, 3 exp (heads of mycode/lstm.py:321-337); in place allowed
__global__ __launch_bounds__(256) void act_fwd_kernel(const float* x, float* y, long n, int activation) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float v = x[i];
    y[i] = activation == 1 ? tanh_f(v) : (activation == 2 ? fmaxf(v, 0.f) : (activation == 3 ? __expf(v) : v));
}

// Gaussian negative log-likelihood of mycode/cost.py:190-229 (likelihood_loss_tf): per sequence b, target second t,
// frame f and axis a:  l = log(var_a + eps) + (y - mu_a)^2 / (var_a + eps), clipped to [-10, 10];
// loss = scale * mean_b sum_{t,f,a} l.  One block per sequence: partial loss into part[b], dmu / dvar (B,3) with the
// clip's zero gradient outside [-10, 10].
__global__ __launch_bounds__(256) void gauss_nll_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                        const float* __restrict__ y, float* __restrict__ part,
                                                        float* __restrict__ dmu, float* __restrict__ dvar, int B, int Ty,
                                                        int fps, float scale, unsigned* __restrict__ ticket, float* __restrict__ loss_out,
                                                        float loss_scale) {
    __shared__ float red[256][7];
    __shared__ int is_last;
    const int b = blockIdx.x;
    const int per = Ty * fps * 3;
    float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // loss, dmu[3], dvar[3]
    const float eps = 1e-20f;
    for (int e = threadIdx.x; e < per; e += 256) {
        const int a = e % 3;
        const float m = mu[b * 3 + a], v = var[b * 3 + a] + eps;
        const float d = y[(size_t)b * per + e] - m;
        const float l = __logf(v) + d * d / v;
        acc[0] += fminf(fmaxf(l, -10.f), 10.f);
        if (l > -10.f && l < 10.f) {
            acc[1 + a] += -2.f * d / v;
            acc[4 + a] += 1.f / v - d * d / (v * v);
        }
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) red[threadIdx.x][k] = acc[k];
    __syncthreads();
    for (int s_ = 128; s_ > 0; s_ >>= 1) {
        if ((int)threadIdx.x < s_)
#pragma unroll
            for (int k = 0; k < 7; ++k) red[threadIdx.x][k] += red[threadIdx.x + s_][k];
        __syncthreads();
    }
    if (threadIdx.x < 3) {
        dmu[b * 3 + threadIdx.x] = red[0][1 + threadIdx.x] * scale / (float)B;
        dvar[b * 3 + threadIdx.x] = red[0][4 + threadIdx.x] * scale / (float)B;
    }
    if (threadIdx.x == 0) {
        if (ticket) {
            __hip_atomic_store(part + b, red[0][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __threadfence();
            is_last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
        } else {
            part[b] = red[0][0];
            is_last = 0;
        }
    }
    __syncthreads();
    if (is_last) {   // the arithmetic of sum_scale_kernel, by the block that finished last (round 4: one launch less per step)
        __threadfence();
        float a = 0.f;
        for (int i = threadIdx.x; i < B; i += 256) a += __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        red[threadIdx.x][0] = a;
        __syncthreads();
        for (int m = 128; m >= 1; m >>= 1) {
            if ((int)threadIdx.x < m) red[threadIdx.x][0] += red[threadIdx.x + m][0];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            loss_out[0] = red[0][0] * loss_scale;
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

// Sampled re-feed (mycode/lstm.py:460-468 via utility.py:83-89; lstm_keras.py:39-44,139-149): one second of fps frames
// drawn around the predicted mean, x = mu_a + sd(var_a) * noise, noise ~ N(0,1) supplied by the caller in the layout of x.
// std_mode 0: sd = sqrt(var) (lstm.py); 1: sd = var (lstm_keras.py passes the variance as stddev).  layout 0: frames
// interleaved x,y,z (tf.stack axis=-1 + reshape); 1: planar [x*fps | y*fps | z*fps] (Concatenate axis=-1).
// One wave per sequence; x / dx rows are ldx floats apart (a slot of the (B,T,3*fps) window).
__global__ __launch_bounds__(64) void sample_refeed_fwd_kernel(const float* __restrict__ mu, const float* __restrict__ var,
                                                               const float* __restrict__ noise, float* __restrict__ x, long ldx,
                                                               int fps, int std_mode, int layout) {
    const int b = blockIdx.x;
    for (int e = threadIdx.x; e < 3 * fps; e += 64) {
        const int a = layout ? e / fps : e % 3;
        const float v = var[b * 3 + a];
        x[(size_t)b * ldx + e] = mu[b * 3 + a] + (std_mode ? v : sqrtf(v)) * noise[(size_t)b * 3 * fps + e];
    }
}

// dmu_a (+)= sum_frames dx;  dvar_a (+)= sum_frames dx * noise * sd'(var_a)   (sd' = 1/(2 sqrt(var)) or 1, IEEE as in TF)
__global__ __launch_bounds__(64) void sample_refeed_bwd_kernel(const float* __restrict__ dx, long ldx, const float* __restrict__ var,
                                                               const float* __restrict__ noise, float* __restrict__ dmu,
                                                               float* __restrict__ dvar, int fps, int std_mode, int layout,
                                                               int accumulate) {
    const int b = blockIdx.x;
    float acc[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int e = threadIdx.x; e < 3 * fps; e += 64) {
        const int a = layout ? e / fps : e % 3;
        const float d = dx[(size_t)b * ldx + e], n = noise[(size_t)b * 3 * fps + e];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            acc[k] += a == k ? d : 0.f;
            acc[3 + k] += a == k ? d * n : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k)
        for (int off = 32; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, 64);
    if (threadIdx.x < 3) {
        const int a = threadIdx.x;
        const float v = var[b * 3 + a];
        const float gm = a == 0 ? acc[0] : (a == 1 ? acc[1] : acc[2]);
        const float gs = a == 0 ? acc[3] : (a == 1 ? acc[4] : acc[5]);
        const float gv = std_mode ? gs : gs * 0.5f / sqrtf(v);
        dmu[b * 3 + a] = (accumulate ? dmu[b * 3 + a] : 0.f) + gm;
        dvar[b * 3 + a] = (accumulate ? dvar[b * 3 + a] : 0.f) + gv;
    }
}

// Unit-norm regulariser of costfunc._mse under cfg.add_xyz_sum1 (mycode/cost.py:20-29): per pixel (row of C >= 3 channels)
// r = ux^2 + uy^2 + uz^2 - 1;  reg = 0.5 * mean_pixels r^2;  d reg / d u_k = 2 r u_k / n_pix, ADDED to dp.  Block partials of
// sum r^2 go to part[] (reduced in fixed order by sum_scale_kernel, so the loss is deterministic).
__global__ __launch_bounds__(256) void xyz_sum1_kernel(const float* __restrict__ p, float* __restrict__ dp, float* __restrict__ part,
                                                       long n_pix, int C) {
    __shared__ float red[256];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float r2 = 0.f;
    if (i < n_pix) {
        const float x = p[i * C], y = p[i * C + 1], z = p[i * C + 2];
        const float r = x * x + y * y + z * z - 1.f;
        const float gsc = 2.f * r / (float)n_pix;
        dp[i * C] += gsc * x;
        dp[i * C + 1] += gsc * y;
        dp[i * C + 2] += gsc * z;
        r2 = r * r;
    }
    red[threadIdx.x] = r2;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// Keras-2.2 categorical_crossentropy on probabilities (convlstm_heatmap.py:192: model.compile(loss='categorical_crossentropy',
// optimizer='adam')), TensorFlow backend form: q = p / sum_c p;  q' = clip(q, 1e-7, 1 - 1e-7);  l = - sum_c t_c log q'_c per pixel,
// loss = mean over pixels.  Gradient w.r.t. p through the clip (zero outside) and the renormalisation:
//   g_c = -t_c / q'_c [eps < q_c < 1 - eps];   dl/dp_k = (g_k - sum_c g_c q_c) / S;   dp = that / n_pix.
// One thread per pixel (C <= 64 channels re-read from cache for the second pass).
__global__ __launch_bounds__(256) void cce_grad_kernel(const float* __restrict__ p, const float* __restrict__ t, float* __restrict__ dp,
                                                       float* __restrict__ part, long n_pix, int C) {
    __shared__ float red[256];
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float l = 0.f;
    if (i < n_pix) {
        const float* pi = p + i * C;
        const float* ti = t + i * C;
        const float eps = 1e-7f;
        float S = 0.f;
        for (int c = 0; c < C; ++c) S += pi[c];
        const float inv = 1.f / S;
        float dot = 0.f;
        for (int c = 0; c < C; ++c) {
            const float q = pi[c] * inv;
            const float qc = fminf(fmaxf(q, eps), 1.f - eps);
            l -= ti[c] * __logf(qc);
            const float g = (q > eps && q < 1.f - eps) ? -ti[c] / qc : 0.f;
            dot += g * q;
        }
        const float sc = inv / (float)n_pix;
        for (int c = 0; c < C; ++c) {
            const float q = pi[c] * inv;
            const float qc = fminf(fmaxf(q, eps), 1.f - eps);
            const float g = (q > eps && q < 1.f - eps) ? -ti[c] / qc : 0.f;
            dp[i * C + c] = (g - dot) * sc;
        }
    }
    red[threadIdx.x] = l;
    __syncthreads();
    for (int m = 128; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) red[threadIdx.x] += red[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}

// ---------------------------------------------------------------------------------------
// Dense(O, tanh | linear) head + Keras mean_squared_error, FORWARD AND BACKWARD IN ONE LAUNCH (round 5) - what model.fit runs
// around the decoder's hidden sequence in mycode/FoV_seq2seq.py:96-103: y = act(hs . W + b), loss = w mean (y - target)^2,
// dpre = dloss/d(pre-activation), dX = dpre . W^T (into the decoder's BPTT), dW = hs^T . dpre, db = column sums of dpre.
// At the reference's batch (32 x 10 = 320 rows, H = 128, O = 6) these were five launches of 4-8 us each - the Dense forward,
// the loss kernel, two GEMMs and a column sum - for 0.5 MFLOP: a sixth of the training step.  A block takes 64 rows: the hs
// tile and W sit in LDS, four lanes share a row's dot products, dX rows leave as whole 16-byte-per-lane lines, the block's
// partial dW / db / squared-error sum go to scratch and the block that takes the last ticket adds the partials IN BLOCK ORDER
// (deterministic).  Rows <= 4096 (64 blocks); larger heads keep the separate launches, which are <= 1 % of their step.
// ---------------------------------------------------------------------------------------
constexpr int HD_RB = 64;        // rows per block
constexpr int HD_MAXB = 64;      // blocks per launch at most
__host__ __device__ inline size_t dense_mse_head_lds_floats(int H) { return (size_t)HD_RB * (H + 4) + (size_t)H * 8 + HD_RB * 8 + 16; }
__global__ __launch_bounds__(256) void dense_mse_head_kernel(const float* __restrict__ hs, const float* __restrict__ W, const float* __restrict__ bias,
                                                             const float* __restrict__ target, float* __restrict__ y_out, float* __restrict__ dX,
                                                             float* __restrict__ part, float* __restrict__ dW, float* __restrict__ db,
                                                             float* __restrict__ loss, int N, int H, int O, int activation, float scale,
                                                             unsigned* __restrict__ ticket, int chunks_per_block, int finish) {
    extern __shared__ __attribute__((aligned(16))) float hd_sm[];
    const int LDH = H + 4;
    float* sHs = hd_sm;                       // [64][H + 4]
    float* sWt = sHs + HD_RB * LDH;           // [H][8] (columns >= O zero)
    float* sD = sWt + H * 8;                  // dpre [64][8] (rows >= N, columns >= O zero)
    float* sRed = sD + HD_RB * 8;             // 4 wave sums of the squared error
    __shared__ int is_last;
    const int tid = threadIdx.x;
    const int C4 = H >> 2;                    // 16-byte pieces per row
    // More than 64 x 64 rows (round 5: the 30 720 rows of configs[1]'s training step): a block walks `chunks_per_block` tiles of 64
    // rows, its partial dW / db / squared error stay in registers across them, and a second launch adds the blocks' partials
    // (finish = 0) - the last-block reduce below is one block reading every partial.
    float aw[2][8], adb = 0.f, sq_tot = 0.f;      // dW rows tid and tid + 256 (H <= 512), db[tid], squared error (thread 0)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk)
#pragma unroll
        for (int o = 0; o < 8; ++o) aw[kk][o] = 0.f;
    const int chunk0 = blockIdx.x * chunks_per_block;
    for (int ch = 0; ch < chunks_per_block; ++ch) {
    const int r0 = (chunk0 + ch) * HD_RB;
    if (r0 >= N) break;
    if (ch > 0) __syncthreads();              // the previous tile's readers are done
    // ---- stage the hs tile (rows past N: zero) and W ----
    // (every request of the tile goes out before the first LDS store: a load-then-store loop is one memory round trip per
    // iteration on a CU that runs one wave per SIMD - 8 us of the kernel's first 18)
    {
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(hs + (size_t)r0 * H), 0,
                                                                             (N - r0 < HD_RB ? N - r0 : HD_RB) * H * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(W), 0, H * O * 4, 0x00020000);
        constexpr int HIT = HD_RB * 128 / 256;      // 16-byte pieces per thread at H = 512
        typedef unsigned hu32x4 __attribute__((ext_vector_type(4)));
        hu32x4 hv[HIT];
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int i = tid + 256 * it;
            hv[it] = __builtin_amdgcn_raw_buffer_load_b128(hrs, i < HD_RB * C4 ? (unsigned)(i * 16) : 0x80000000u, 0, 0);   // rows past N: beyond the descriptor = 0
        }
        float wv[16];
#pragma unroll
        for (int it = 0; it < 16; ++it) {      // (W: the block's first tile only - later passes request nothing)
            const int i = tid + 256 * it, k = i >> 3, o = i & 7;
            wv[it] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, (ch == 0 && i < H * 8 && o < O) ? (unsigned)((k * O + o) * 4) : 0x80000000u, 0, 0));
        }
#pragma unroll
        for (int it = 0; it < HIT; ++it) {
            const int i = tid + 256 * it;
            if (i < HD_RB * C4) {
                const int r = i / C4, c4 = i - r * C4;
                *(hu32x4*)(sHs + r * LDH + 4 * c4) = hv[it];
            }
        }
#pragma unroll
        for (int it = 0; it < 16; ++it) {
            const int i = tid + 256 * it;
            if (ch == 0 && i < H * 8) sWt[i] = wv[it];
        }
    }
    __syncthreads();
    // ---- forward + loss gradient: lanes 4r .. 4r+3 share row r (k = 4j + q: the four read neighbouring words) ----
    const int r = tid >> 2, q = tid & 3;
    float acc[8];
#pragma unroll
    for (int o = 0; o < 8; ++o) acc[o] = 0.f;
#pragma unroll 8
    for (int j = 0; j < C4; ++j) {
        const int k = 4 * j + q;
        const float h = sHs[r * LDH + k];
        const f32x4 w0 = *(const f32x4*)(sWt + k * 8), w1 = *(const f32x4*)(sWt + k * 8 + 4);
#pragma unroll
        for (int o = 0; o < 4; ++o) { acc[o] = fmaf(h, w0[o], acc[o]); acc[4 + o] = fmaf(h, w1[o], acc[4 + o]); }
    }
#pragma unroll
    for (int o = 0; o < 8; ++o) {
        acc[o] += __shfl_xor(acc[o], 1);
        acc[o] += __shfl_xor(acc[o], 2);
    }
    float sq = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {    // lane q owns the outputs q and q + 4
        const int o = q + 4 * half;
        float g = 0.f;
        if (o < O && r0 + r < N) {
            const float lo = q == 0 ? acc[0] : (q == 1 ? acc[1] : (q == 2 ? acc[2] : acc[3]));      // (register arrays take no lane-dependent index)
            const float hi2 = q == 0 ? acc[4] : (q == 1 ? acc[5] : (q == 2 ? acc[6] : acc[7]));
            const float pre = (half ? hi2 : lo) + bias[o];
            const float yv = activation == 1 ? tanh_f(pre) : pre;
            const size_t e = (size_t)(r0 + r) * O + o;
            if (y_out) y_out[e] = yv;
            const float d = yv - target[e];
            sq += d * d;
            g = 2.f * d * scale;
            if (activation == 1) g *= (1.f - yv * yv);
        }
        sD[r * 8 + o] = g;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) sq += __shfl_xor(sq, m);
    if ((tid & 63) == 0) sRed[tid >> 6] = sq;
    __syncthreads();
    // ---- dX = dpre . W^T: a thread keeps its four k rows of W, a wave instruction writes whole rows ----
    if (dX) {
        const int c4 = tid % C4, rsub = tid / C4, rstep = 256 / C4;      // (H <= 512: C4 <= 128, at least two rows per pass)
        if (rsub < rstep) {
            float wk[4][8];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 w0 = *(const f32x4*)(sWt + (4 * c4 + i) * 8), w1 = *(const f32x4*)(sWt + (4 * c4 + i) * 8 + 4);
#pragma unroll
                for (int o = 0; o < 4; ++o) { wk[i][o] = w0[o]; wk[i][4 + o] = w1[o]; }
            }
            for (int rr = rsub; rr < HD_RB && r0 + rr < N; rr += rstep) {
                const f32x4 g0 = *(const f32x4*)(sD + rr * 8), g1 = *(const f32x4*)(sD + rr * 8 + 4);
                f32x4 out;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float a = 0.f;
#pragma unroll
                    for (int o = 0; o < 4; ++o) a = fmaf(g0[o], wk[i][o], fmaf(g1[o], wk[i][4 + o], a));
                    out[i] = a;
                }
                *(f32x4*)(dX + (size_t)(r0 + rr) * H + 4 * c4) = out;
            }
        }
    }
    // ---- the block's partial dW[k][o] += sum_r hs[r][k] dpre[r][o], db[o] += sum_r dpre[r][o], squared error ----
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = tid + 256 * kk;
        if (k < H) {
#pragma unroll 8
            for (int rr = 0; rr < HD_RB; ++rr) {
                const float h = sHs[rr * LDH + k];
                const f32x4 g0 = *(const f32x4*)(sD + rr * 8), g1 = *(const f32x4*)(sD + rr * 8 + 4);
#pragma unroll
                for (int o = 0; o < 4; ++o) { aw[kk][o] = fmaf(h, g0[o], aw[kk][o]); aw[kk][4 + o] = fmaf(h, g1[o], aw[kk][4 + o]); }
            }
        }
    }
    if (tid < O) {
        float a = 0.f;
        for (int rr = 0; rr < HD_RB; ++rr) a += sD[rr * 8 + tid];
        adb += a;
    }
    if (tid == 0) sq_tot += (sRed[0] + sRed[1]) + (sRed[2] + sRed[3]);
    }   // tiles of this block
    const int NE = H * O + O + 1;
    float* mypart = part + (size_t)blockIdx.x * NE;
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        const int k = tid + 256 * kk;
        if (k < H)
        {
#pragma unroll
            for (int o = 0; o < 8; ++o)
                if (o < O) mypart[(size_t)k * O + o] = aw[kk][o];
        }
    }
    if (tid < O) mypart[(size_t)H * O + tid] = adb;
    if (tid == 0) mypart[NE - 1] = sq_tot;
    if (!finish) return;                      // a reduce launch adds the blocks' partials (dense_mse_head_reduce_kernel)
    __threadfence();
    __syncthreads();
    if (tid == 0) is_last = atomicAdd(ticket, 1u) == gridDim.x - 1u;
    __syncthreads();
    if (is_last) {
        __threadfence();
        // L1-bypassing loads (another launch's partials may sit in this CU's L1 under the same addresses), ALL of an element's
        // requested before the first is used: as a chain of load-then-add the five partials of the reference's batch cost a
        // memory round trip each (the kernel took 19 us, 14 of them here)
        const __amdgpu_buffer_rsrc_t prs = __builtin_amdgcn_make_buffer_rsrc(part, 0, (int)(gridDim.x * (unsigned)NE * 4u), 0x00020000);
        for (int e = tid; e < NE; e += 256) {
            float v[HD_MAXB];
#pragma unroll
            for (int blk = 0; blk < HD_MAXB; ++blk)
                v[blk] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(prs, (unsigned)((blk * NE + e) * 4), 0, 16 /* sc1 */));   // blocks past the grid: out of range = 0
            float a = 0.f;
#pragma unroll
            for (int blk = 0; blk < HD_MAXB; ++blk) a += v[blk];      // block order: deterministic (absent blocks add +0)
            if (e < H * O) dW[e] = a;
            else if (e < H * O + O) db[e - H * O] = a;
            else if (loss) loss[0] = a * scale;
        }
        if (tid == 0) __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // the slot is this stream's again
    }
}

// The second launch of the head at more than 4 096 rows: element e of (dW | db | squared error) = the blocks' partials in block
// order.  16 elements x 16 partial groups per block (group q adds partials q, q + 16, ...; the 16 group sums are folded in a fixed
// order): a hundred blocks instead of one block reading everything.
__global__ __launch_bounds__(256) void dense_mse_head_reduce_kernel(const float* __restrict__ part, int nb, int NE, int HO, int O,
                                                                    float* __restrict__ dW, float* __restrict__ db, float* __restrict__ loss,
                                                                    float scale) {
    __shared__ float red[16][17];
    const int el = threadIdx.x & 15, q = threadIdx.x >> 4;
    const int e = blockIdx.x * 16 + el;
    float a = 0.f;
    if (e < NE) {
        int b = q;
        for (; b + 7 * 16 < nb; b += 8 * 16) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + 16 * u) * NE + e];
#pragma unroll
            for (int u = 0; u < 8; ++u) a += v[u];
        }
        for (; b < nb; b += 16) a += part[(size_t)b * NE + e];
    }
    red[q][el] = a;
    __syncthreads();
    if (q == 0 && e < NE) {
        float t = 0.f;
#pragma unroll
        for (int g = 0; g < 16; ++g) t += red[g][el];
        if (e < HO) dW[e] = t;
        else if (e < HO + O) db[e - HO] = t;
        else if (loss) loss[0] = t * scale;
    }
}

// tf.train.RMSPropOptimizer (TF 1.x, momentum 0, not centered; mycode/lstm.py:556-567) with the script's optional
// clip_by_value(grad, -clip, clip):  ms = decay*ms + (1-decay) g^2;  p -= lr * g / sqrt(ms + eps).  (eps INSIDE the
// root, ms initialised to ONE - both unlike Keras RMSprop.)
__global__ __launch_bounds__(256) void rmsprop_tf_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ ms,
                                                         long n, float lr, float decay, float eps, float clip, const unsigned* g0,
                                                         const unsigned* g1, const unsigned* g2, long long* applied) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n || optimizer_poisoned(g0, g1, g2)) return;     // fail-stop: a persistent kernel of the step gave up -> no update
    if (i == 0 && applied) *applied += 1;
    float gi = g[i];
    if (clip > 0.f) gi = fminf(fmaxf(gi, -clip), clip);
    const float m = decay * ms[i] + (1.f - decay) * gi * gi;
    ms[i] = m;
    p[i] = p[i] - lr * gi / sqrtf(m + eps);
}

// ---------------------------------------------------------------------------------------
// host helpers
// ---------------------------------------------------------------------------------------
static int check_launch(const char* what) {
    if (env_knobs().dbg_trace) fprintf(stderr, "[fov trace] launched: %s\n", what);   // FOV_DBG_TRACE=1 (with HIP_LAUNCH_BLOCKING=1: the last line names the launch BEFORE a faulting one)
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("%s launch: %s", what, hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// C (+)= A.B.  Few output tiles and a long K -> split-K: slices write partial tiles into
// `scratch` ([split][M][N]) and a fixed-order reduce adds them (deterministic, no atomics).
int gemm_f32(GemmArgs g, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (g.M <= 0 || g.N <= 0) return FOV_OK;
    const long K = (long)g.KO * g.KI;
    if (K <= 0) {
        if (!accumulate) (void)hipMemsetAsync(g.c, 0, sizeof(float) * (size_t)g.M * g.ldc, stream);
        return FOV_OK;
    }
    const size_t mn = (size_t)(g.M + g.bias_row) * g.N;
    // tile shape by M: short-and-wide weight-gradient products (M = F or Out) would waste a 128-row tile
    int BM, BN, variant;
    if (g.M <= 32) { BM = 32; BN = 256; variant = 0; }
    else if (g.M <= 96) { BM = 96; BN = 256; variant = 1; }
    else { BM = 128; BN = 128; variant = 2; }
    // mid-size outputs (e.g. 512 x 1024 of a per-step projection): 128 x 128 tiles would occupy a fraction of
    // the 256 CUs and K is too short to split -> 64 x 64 tiles
    if (variant == 2 && ((g.M + 127) / 128) * ((g.N + 127) / 128) < 128 && K <= 2048) { BM = 64; BN = 64; variant = 3; }
    // short K (the weight gradients of a wide layer at a small batch: 1025 x 2048 x 320 for lstm.py's LSTMCell(400) at batch 32,
    // 144 tiles of 128 x 128): nothing to split, 64 x 64 tiles fill the chip twice over (its training step 0.415 -> 0.397 ms)
    if (variant == 2 && K <= 512 && ((g.M + 127) / 128) * ((g.N + 127) / 128) < 256) { BM = 64; BN = 64; variant = 3; }
    // the same regime with a 33..96-row output (dK of a 90-wide input: eight 96 x 256 tiles): 64 x 64 tiles as well
    if (variant == 1 && K <= 512 && g.N >= 1024) { BM = 64; BN = 64; variant = 3; }
    // mid-length K on a few dozen 128 x 128 tiles (the 512 x 1024 x 5120 weight gradients of configs[2] at T = 10: 32 tiles x 8 slices):
    // 64 x 128 tiles halve the slices - and the partials written and re-read - for the same block count; measured on the training
    // step 0.788 -> 0.775 ms.  Longer K stays (T = 30, K = 15 360: 1.955 -> 1.969 ms; configs[1], K = 30 720: 1.231 -> 1.287 ms).
    if (variant == 2 && K > 2048 && K <= 8192 && ((g.M + 127) / 128) * ((g.N + 127) / 128) <= 64) { BM = 64; BN = 128; variant = 4; }
    if (const int v = env_knobs().gemm_variant) {   // tuning knob (experiments), FOV_GEMM_VARIANT
        if (v == 2) { BM = 128; BN = 128; variant = 2; }
        if (v == 3) { BM = 64; BN = 64; variant = 3; }
        if (v == 4) { BM = 64; BN = 128; variant = 4; }
    }
    // per-thread staging offsets span one block tile: they must fit 32 bits
    {
        const long amax = (long)BM * (g.a_sm < 0 ? -g.a_sm : g.a_sm) + 16 * (g.a_ski < 0 ? -g.a_ski : g.a_ski);
        const long bmax = (long)BN * (g.b_sn < 0 ? -g.b_sn : g.b_sn) + 16 * (g.b_ski < 0 ? -g.b_ski : g.b_ski);
        if (amax >= (1L << 30) || bmax >= (1L << 30) || g.a_sm < 0 || g.a_ski < 0 || g.b_sn < 0 || g.b_ski < 0) {
            set_error("gemm_f32: operand strides out of range");
            return FOV_ERR_UNSUPPORTED;
        }
    }
    const int ntpr = (g.KI + GBK - 1) / GBK;
    const long ktiles = (long)g.KO * ntpr;
    if (ktiles > 0x7fffffffL) { set_error("gemm_f32: K too large"); return FOV_ERR_UNSUPPORTED; }
    const int tiles = ((g.M + BM - 1) / BM) * ((g.N + BN - 1) / BN);
    int split = 1;
    if (tiles < 512 && ktiles >= 32) {
        // fewer tiles than two per CU and a long K: slices.  (256 <= tiles < 512, e.g. the 5120 x 256 x 1024 data gradient of
        // the stacked layer on 64 x 64 tiles: four slices measured 55 -> 43 us)
        // The blocks of a launch are co-resident (2-3 per CU) and share the matrix pipe, so the launch takes
        // ceil(blocks / CUs) block-times: 40 tiles x 13 slices = 520 blocks on 256 CUs cost THREE rounds for 2.03 rounds of
        // work (the 513 x 1024 weight gradients of configs[2] ran at 58 % of peak).  Among the slice counts up to the
        // old target (about 512 resp. 1024 blocks) take the one that fills its last round best; ties go to fewer slices.
        const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
        const int want = ((tiles < 256 ? 512 : 1024) + tiles - 1) / tiles;
        long maxs = ktiles / 4;   // >= 4 k-tiles (64 k) per slice
        if (maxs > 64) maxs = 64;
        if (maxs > want) maxs = want;
        double best = -1.0;
        for (int sct = 1; sct <= (int)maxs; ++sct) {
            const long blocks = (long)tiles * sct;
            const long rounds = (blocks + cus - 1) / cus;
            double fill = (double)blocks / (double)(rounds * cus);
            if (blocks < cus) fill *= 0.5;                       // less than one block per CU: only if nothing else is possible
            else if (rounds == 1) fill *= 0.9;                   // one round: a second one overlaps latencies better
            if (fill > best + 1e-9) { best = fill; split = sct; }
        }
        if (split < 1) split = 1;
    } else if (tiles < 128 && ktiles >= 8) {
        // SHORT K on a few tiles (model.fit at the reference's batch of 32: B*T = 320 rows -> 20 k-tiles, dK of a layer is
        // two 96 x 256 tiles): a block's time is its k-tiles x ~2.2 us of unhidden global -> LDS -> barrier latency, so two
        // blocks took 46 us for 30 MFLOP.  Slices of >= 2 k-tiles up to about one block per CU; the reduce launch (4.5 us)
        // is paid back from 4 k-tiles per block on.
        split = (256 + tiles - 1) / tiles;
        const long maxs = ktiles / 2;
        if (split > maxs) split = (int)maxs;
        if (split > 64) split = 64;
        if (split < 1) split = 1;
    }
    if (const int v = env_knobs().gemm_split) { if (v >= 1 && v <= ktiles) split = v; }   // tuning knob, FOV_GEMM_SPLIT
    while (split > 1 && (size_t)split * mn > scratch_floats) --split;
    const long tps = (ktiles + split - 1) / split;
    split = (int)((ktiles + tps - 1) / tps);
    // accumulate with a single K slice happens in the epilogue (C += A.B, one writer per element)
    g.add_c = (accumulate && split == 1) ? 1 : 0;
    const bool via_scratch = (split > 1);
    float* c_final = g.c;
    bool deferred = false;
    if (via_scratch) {
        if (mn * split > scratch_floats) { set_error("gemm_f32: scratch too small (%zu floats needed)", mn * split); return FOV_ERR_WORKSPACE; }
        if (g.ldc != g.N) { set_error("gemm_f32: accumulate/split-K needs a dense C (ldc == N)"); return FOV_ERR_INVALID; }
        if (float* arena = defer_alloc(c_final, mn, mn * split, stream)) { scratch = arena; deferred = true; }
        g.c = scratch;
    } else {
        if (int rc_ = defer_touch(c_final, (size_t)(g.M + g.bias_row) * g.ldc, stream)) return rc_;
    }
    g.split = split;
    g.tiles_per_split = (int)tps;
    // staging modes.  k-slow: all row offsets of one k row stay below the k stride and 16 k rows fit a 31-bit
    // byte range -> buffer loads; 16-byte form when the row index is contiguous, every row start is 16-byte
    // aligned and a group of four never straddles the matrix edge.
    const int Ma = g.a2 ? g.M1 : g.M;   // width of the first A operand
    const bool a_ks = g.a_ski >= (long)(Ma - 1) * g.a_sm + 1 && g.a_ski < (1L << 24) && (((uintptr_t)g.a) & 3) == 0;
    const bool b_ks = g.b_ski >= (long)(g.N - 1) * g.b_sn + 1 && g.b_ski < (1L << 24) && (((uintptr_t)g.b) & 3) == 0;
    const bool a_v4 = a_ks && g.a_sm == 1 && (g.M & 3) == 0 && (g.a_sko & 3) == 0 && (g.a_ski & 3) == 0 && (((uintptr_t)g.a) & 15) == 0 &&
                      (!g.a2 || ((g.M1 & 3) == 0 && (g.a2_sko & 3) == 0 && (g.a2_ski & 3) == 0 && (((uintptr_t)g.a2) & 15) == 0 &&
                                 g.a2_ski >= (long)(g.M - g.M1) && g.a2_ski < (1L << 24)));
    const bool b_v4 = b_ks && g.b_sn == 1 && (g.N & 3) == 0 && (g.b_sko & 3) == 0 && (g.b_ski & 3) == 0 && (((uintptr_t)g.b) & 15) == 0;
    // k-fast: k is the contiguous index, rows and ko rows start 16-byte aligned, KI a multiple of 4
    const bool a_kf = g.a_ski == 1 && (g.KI & 3) == 0 && (g.a_sm & 3) == 0 && (g.a_sko & 3) == 0 && (((uintptr_t)g.a) & 15) == 0;
    const bool b_kf = g.b_ski == 1 && (g.KI & 3) == 0 && (g.b_sn & 3) == 0 && (g.b_sko & 3) == 0 && (((uintptr_t)g.b) & 15) == 0;
    const int amode = a_v4 ? 0 : (a_ks ? 1 : (a_kf ? 3 : 2)), bmode = b_v4 ? 0 : (b_ks ? 1 : (b_kf ? 3 : 2));
    if ((g.bias_row && bmode != 0) || ((g.a_period || g.a2_period) && (amode > 1 || g.KO != 1)) || (g.a2 && (g.M1 % BM || amode != 0))) {
        set_error("gemm_f32: fused weight gradient needs 16-byte-aligned k-slow operands and M1 a multiple of the row tile");
        return FOV_ERR_UNSUPPORTED;
    }
    g.grid_n = (g.N + BN - 1) / BN;
    g.grid_m = (g.M + BM - 1) / BM;
    g.xcd_remap = split >= 8 ? 1 : 0;
    const dim3 grid = g.xcd_remap ? dim3((unsigned)(8 * ((split + 7) / 8) * g.grid_n * g.grid_m)) : dim3(g.grid_n, g.grid_m, split);
    if (env_knobs().dbg_trace)
        fprintf(stderr, "[fov trace] gemm_f32 next: M=%d N=%d KO=%d KI=%d variant=%d amode=%d bmode=%d split=%d a2=%d periods=%d/%d bias=%d add_c=%d\n", g.M, g.N,
                g.KO, g.KI, variant, amode, bmode, split, g.a2 ? 1 : 0, g.a_period, g.a2_period, g.bias_row, g.add_c);
#define FOV_GEMM_LAUNCH(MI_, NI_, WM_, A_, B_) \
    hipLaunchKernelGGL((gemm_f32_kernel<MI_, NI_, WM_, A_, B_>), grid, dim3(256), 0, stream, g)
#define FOV_GEMM_MODES(MI_, NI_, WM_)                                                            \
    switch (amode * 4 + bmode) {                                                                 \
        case 0: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 0, 0); break;                                     \
        case 1: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 0, 1); break;                                     \
        case 2: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 0, 2); break;                                     \
        case 3: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 0, 3); break;                                     \
        case 4: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 1, 0); break;                                     \
        case 5: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 1, 1); break;                                     \
        case 6: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 1, 2); break;                                     \
        case 7: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 1, 3); break;                                     \
        case 8: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 2, 0); break;                                     \
        case 9: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 2, 1); break;                                     \
        case 10: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 2, 2); break;                                    \
        case 11: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 2, 3); break;                                    \
        case 12: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 3, 0); break;                                    \
        case 13: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 3, 1); break;                                    \
        case 14: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 3, 2); break;                                    \
        default: FOV_GEMM_LAUNCH(MI_, NI_, WM_, 3, 3); break;                                    \
    }
    if (variant == 0) { FOV_GEMM_MODES(2, 4, 1) }
    else if (variant == 1) { FOV_GEMM_MODES(6, 4, 1) }
    else if (variant == 3) { FOV_GEMM_MODES(2, 2, 2) }
    else if (variant == 4) { FOV_GEMM_MODES(2, 4, 2) }
    else { FOV_GEMM_MODES(4, 4, 2) }
#undef FOV_GEMM_MODES
#undef FOV_GEMM_LAUNCH
    int rc = check_launch("gemm_f32");
    if (rc || !via_scratch) return rc;
    return reduce_or_defer(deferred, scratch, c_final, (long)mn, split, accumulate, stream, "splitk_reduce");
}

// Skinny weight gradients: out[s][w] = sum_r S[r][s] * Wd[r][w] with ns <= 8 (dK of the decoder LSTM, F_dec = 6;
// dW of Dense(6)).  An MFMA tile would be >= 80 % padding and the product is HBM-bound anyway (Wd is read
// once): thread = column w, block = (256 columns, chunk of rows), the ns values of a row are wave-uniform
// scalar loads.  Partials land in scratch already in the output layout (element s*os + w*ow of slice
// `chunk`), and splitk_reduce adds the slices in a fixed order.
template <int NS, int VEC>
__global__ __launch_bounds__(256) void skinny_tn_kernel(const float* __restrict__ S, long ss, const float* __restrict__ Wd,
                                                        long ldw, float* __restrict__ part, long rows, int nw,
                                                        long rows_per_chunk, long os, long ow) {
    // block = 64*VEC columns x 4 row groups (wave q takes rows r0+q, r0+q+4, ...); eight rows in flight per
    // thread, VEC = 4 reads 16 bytes per lane and row
    __shared__ float red[4][NS][64 * VEC];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int w = (blockIdx.x * 64 + li) * VEC;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    float acc[NS][VEC];
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) acc[i][v] = 0.f;
    if (w < nw) {
        for (long r = r0 + q; r < r1; r += 32) {
            float wv[8][VEC];
            long rr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = r + 4 * u < r1;
                rr[u] = ok ? r + 4 * u : r;   // rows past the chunk re-read row r and contribute zero
                if (VEC == 4) {
                    const f32x4 t = *(const f32x4*)(Wd + rr[u] * ldw + w);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) wv[u][v] = ok ? t[v] : 0.f;
                } else {
                    wv[u][0] = ok ? Wd[rr[u] * ldw + w] : 0.f;
                }
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const float sv = S[rr[u] * ss + i];
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[i][v] = fmaf(sv, wv[u][v], acc[i][v]);
                }
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i)
#pragma unroll
        for (int v = 0; v < VEC; ++v) red[q][i][li * VEC + v] = acc[i][v];
    __syncthreads();
    if (q == 0 && w < nw) {
        float* pp = part + (long)blockIdx.y * NS * nw;
#pragma unroll
        for (int i = 0; i < NS; ++i)
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int c = li * VEC + v;
                pp[i * os + (w + v) * ow] = (red[0][i][c] + red[1][i][c]) + (red[2][i][c] + red[3][i][c]);
            }
    }
}

// returns 1 when the product was done here, 0 when the caller should use the MFMA GEMM, < 0 on error
static int skinny_tn(const float* S, long ss, int ns, const float* Wd, long ldw, int nw, long rows, float* out, long os,
                     long ow, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (ns < 1 || ns > 8 || rows < 1024) return 0;
    const bool vec = nw >= 512 && (nw & 3) == 0 && (ldw & 3) == 0 && (((uintptr_t)Wd) & 15) == 0;
    const int colblocks = (nw + (vec ? 255 : 63)) / (vec ? 256 : 64);
    long chunks = (2048 + colblocks - 1) / colblocks;   // ~2048 blocks when the rows allow it
    if (chunks > rows / 32) chunks = rows / 32;         // >= 32 rows per chunk
    if (chunks > 1024) chunks = 1024;
    const size_t n = (size_t)ns * nw;
    while (chunks > 1 && (size_t)chunks * n > scratch_floats) chunks >>= 1;
    if (chunks < 1 || (size_t)chunks * n > scratch_floats) return 0;
    const long rpc = (rows + chunks - 1) / chunks;
    const dim3 grid(colblocks, (unsigned)chunks);
    bool deferred = false;
    if (float* arena = defer_alloc(out, n, (size_t)chunks * n, stream)) { scratch = arena; deferred = true; }
#define FOV_SKINNY(NSV)                                                                                                       \
    case NSV:                                                                                                                 \
        if (vec) hipLaunchKernelGGL((skinny_tn_kernel<NSV, 4>), grid, dim3(256), 0, stream, S, ss, Wd, ldw, scratch, rows, nw, \
                                    rpc, os, ow);                                                                             \
        else hipLaunchKernelGGL((skinny_tn_kernel<NSV, 1>), grid, dim3(256), 0, stream, S, ss, Wd, ldw, scratch, rows, nw,    \
                                rpc, os, ow);                                                                                 \
        break
    switch (ns) {
        FOV_SKINNY(1); FOV_SKINNY(2); FOV_SKINNY(3); FOV_SKINNY(4); FOV_SKINNY(5); FOV_SKINNY(6); FOV_SKINNY(7); FOV_SKINNY(8);
    }
#undef FOV_SKINNY
    int rc = check_launch("skinny_tn");
    if (rc) return rc;
    rc = reduce_or_defer(deferred, scratch, out, (long)n, (int)chunks, accumulate, stream, "skinny_reduce");
    return rc ? rc : 1;
}

// Weight gradients of the others-mixing head (given_others...py:127-130,166-168,257-265 under model.fit) in ONE launch and
// one reduce: with rows r = (t, b) of the unrolled decoder's tape,
//     [dense_W ; dense_b]       = [h2_t | 1]^T dpre_p              (H + 1, O)
//     [mix_W ; mix_b]           = [others_t | p_t | 1]^T dpre_m    (n_oth + O + 1, O)
// written as one (H + 1 + n_oth + O + 1, O) block - the layout of dense_W, dense_b, mix_W, mix_b in a trainer's flat
// gradient buffer.  Same scheme as skinny_tn (thread = one column of the wide operand, block = 64 columns x 4 row
// groups, eight rows in flight, the O values of a row are broadcast loads), the column decides which array it reads;
// `others` stays in its (B, T, n_oth) layout (row (t, b) is others[b][t]).
template <int NS>
__global__ __launch_bounds__(256) void mix_head_wgrad_kernel(const float* __restrict__ h2, const float* __restrict__ dpre_p,
                                                             const float* __restrict__ others, const float* __restrict__ p,
                                                             const float* __restrict__ dpre_m, float* __restrict__ part, int B,
                                                             int T, int H, int n_oth, long rows_per_chunk) {
    __shared__ float red[4][NS][64];
    const int li = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int w = blockIdx.x * 64 + li;
    const int ncol = H + 1 + n_oth + NS + 1;
    const long rows = (long)B * T;
    const long r0 = (long)blockIdx.y * rows_per_chunk;
    long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    // what this thread's column reads: kind 0 = array with row stride ld (row r), 1 = ones, 2 = others[b][t][col]
    int kind = 1, col = 0;
    long ld = 0;
    const float* src = dpre_p;
    const float* S = dpre_p;
    if (w < H) { kind = 0; src = h2; ld = H; col = w; }
    else if (w == H) { kind = 1; }
    else {
        S = dpre_m;
        const int j = w - (H + 1);
        if (j < n_oth) { kind = 2; src = others; col = j; }
        else if (j < n_oth + NS) { kind = 0; src = p; ld = NS; col = j - n_oth; }
    }
    float acc[NS];
#pragma unroll
    for (int i = 0; i < NS; ++i) acc[i] = 0.f;
    if (w < ncol) {
        for (long r = r0 + q; r < r1; r += 32) {
            float wv[8];
            long rr[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const bool ok = r + 4 * u < r1;
                rr[u] = ok ? r + 4 * u : r;
                // one unconditional load per row whatever the column reads (a load inside a branch is waited for at the merge)
                const int t = (int)((unsigned)rr[u] / (unsigned)B), b = (int)rr[u] - t * B;
                const long off = kind == 2 ? ((long)b * T + t) * n_oth + col : rr[u] * ld + col;   // kind 1: element 0 of dpre_p, unused
                const float v = src[off];
                wv[u] = ok ? (kind == 1 ? 1.f : v) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) acc[i] = fmaf(S[rr[u] * NS + i], wv[u], acc[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < NS; ++i) red[q][i][li] = acc[i];
    __syncthreads();
    if (q == 0 && w < ncol) {
        float* pp = part + (long)blockIdx.y * NS * ncol + (long)w * NS;
#pragma unroll
        for (int i = 0; i < NS; ++i) pp[i] = (red[0][i][li] + red[1][i][li]) + (red[2][i][li] + red[3][i][li]);
    }
}

size_t mix_head_wgrad_scratch_floats(int B, int T, int H, int O, int n_oth) {
    const long rows = (long)B * T;
    long chunks = rows / 32;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    return (size_t)chunks * (H + 1 + n_oth + O + 1) * O + 64;
}

int mix_head_wgrad(const float* h2, const float* dpre_p, const float* others, const float* p, const float* dpre_m, float* out, int B,
                   int T, int H, int O, int n_oth, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    const long rows = (long)B * T;
    const int ncol = H + 1 + n_oth + O + 1;
    const size_t n = (size_t)ncol * O;
    if (rows <= 0) {
        if (!accumulate) (void)hipMemsetAsync(out, 0, sizeof(float) * n, stream);
        return FOV_OK;
    }
    long chunks = rows / 32;
    if (chunks > 256) chunks = 256;
    if (chunks < 1) chunks = 1;
    if ((size_t)chunks * n > scratch_floats) { set_error("mix_head_wgrad: scratch too small"); return FOV_ERR_WORKSPACE; }
    const long rpc = (rows + chunks - 1) / chunks;
    chunks = (rows + rpc - 1) / rpc;
    const dim3 grid((ncol + 63) / 64, (unsigned)chunks);
    bool deferred = false;
    if (float* arena = defer_alloc(out, n, (size_t)chunks * n, stream)) { scratch = arena; deferred = true; }
#define FOV_HEADW(NSV) \
    case NSV: hipLaunchKernelGGL(mix_head_wgrad_kernel<NSV>, grid, dim3(256), 0, stream, h2, dpre_p, others, p, dpre_m, scratch, B, T, H, n_oth, rpc); break
    switch (O) {
        FOV_HEADW(1); FOV_HEADW(2); FOV_HEADW(3); FOV_HEADW(4); FOV_HEADW(5); FOV_HEADW(6); FOV_HEADW(7); FOV_HEADW(8);
        default: set_error("mix_head_wgrad: O <= 8"); return FOV_ERR_UNSUPPORTED;
    }
#undef FOV_HEADW
    int rc = check_launch("mix_head_wgrad");
    if (rc) return rc;
    return reduce_or_defer(deferred, scratch, out, (long)n, (int)chunks, accumulate, stream, "mix_head_wgrad reduce");
}

// Narrow matrices (cols <= 16, e.g. the Dense(6) bias gradient over B*T rows): the column-per-thread kernel
// above would keep 6 lanes busy.  Here a block owns a chunk of rows, threads stride over the rows with all
// columns in registers, and a fixed-order LDS tree folds the 256 per-thread sums - deterministic.
template <int COLS>
__global__ __launch_bounds__(256) void colsum_narrow_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                            long rows, int cols, long rows_per_chunk) {
    __shared__ float red[256][COLS + 1];
    const long r0 = (long)blockIdx.x * rows_per_chunk;
    long r1 = r0 + rows_per_chunk;
    if (r1 > rows) r1 = rows;
    float acc[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) acc[c] = 0.f;
    for (long r = r0 + threadIdx.x; r < r1; r += 256) {
        const float* xp = x + r * cols;
#pragma unroll
        for (int c = 0; c < COLS; ++c)
            if (c < cols) acc[c] += xp[c];
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) red[threadIdx.x][c] = acc[c];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s)
#pragma unroll
            for (int c = 0; c < COLS; ++c) red[threadIdx.x][c] += red[threadIdx.x + s][c];
        __syncthreads();
    }
    if ((int)threadIdx.x < cols) partial[(long)blockIdx.x * cols + threadIdx.x] = red[0][threadIdx.x];
}

int colsum(const float* x, float* out, long rows, int cols, int accumulate, float* scratch, size_t scratch_floats,
           hipStream_t stream) {
    if (cols <= 0) return FOV_OK;
    int chunks = (int)((rows + 127) / 128);
    if (chunks < 1) chunks = 1;
    if (chunks > 256) chunks = 256;
    if ((size_t)chunks * cols > scratch_floats) { set_error("colsum: scratch too small"); return FOV_ERR_WORKSPACE; }
    long rpc = (rows + chunks - 1) / chunks;
    bool deferred = false;
    if (float* arena = defer_alloc(out, (size_t)cols, (size_t)chunks * cols, stream)) { scratch = arena; deferred = true; }   // (the narrow form below needs fewer slices)
    if (cols <= 16 && rows >= 4096) {
        chunks = (int)((rows + 2047) / 2048);
        if (chunks > 256) chunks = 256;
        rpc = (rows + chunks - 1) / chunks;
        if (cols <= 8)
            hipLaunchKernelGGL(colsum_narrow_kernel<8>, dim3(chunks), dim3(256), 0, stream, x, scratch, rows, cols, rpc);
        else
            hipLaunchKernelGGL(colsum_narrow_kernel<16>, dim3(chunks), dim3(256), 0, stream, x, scratch, rows, cols, rpc);
    } else {
        hipLaunchKernelGGL(colsum_partial_kernel, dim3((cols + 255) / 256, chunks), dim3(256), 0, stream, x, scratch, rows,
                           cols, rpc > 0 ? rpc : 1);
    }
    int rc = check_launch("colsum_partial");
    if (rc) return rc;
    return reduce_or_defer(deferred, scratch, out, (long)cols, chunks, accumulate, stream, "colsum_reduce");
}

size_t lstm_bwd_workspace_floats(int B, int T, int F, int H) {
    // dh_rec (B,H) + dc (B,H) + split-K scratch for the largest weight-gradient GEMM / colsum
    size_t wg = (size_t)64 * (size_t)(F > H ? F : H) * 4 * H;   // split-K partials of dK / dR (<= 64 slices)
    size_t cs = (size_t)256 * 4 * H + (size_t)((B + 15) / 16) * 4 * H;   // colsum partials + per-tile db partials
    size_t st = (size_t)8 * B * H;                               // split-K partials of the per-step dh GEMM
    size_t m = wg > cs ? wg : cs;
    // fp32, 256 -> 256 on the eight-workgroup BPTT kernel: the eight partial tapes of dx = dz K^T behind the per-tile db partials
    const size_t dxp = (F == 256 && H == 256) ? (size_t)8 * B * T * 256 + (size_t)((B + 15) / 16) * 4 * H + 64 : 0;
    if (dxp > m) m = dxp;
    // head: status word + granule buffers of the persistent BPTT kernel (when the shape allows it)
    size_t head = (kStatusBytes + kXchBytes) / sizeof(float);   // header + the fixed granule area (xch_common.h)
    return head + (size_t)2 * B * H + (m > st ? m : st) + 64;
}

// ---------------------------------------------------------------------------------------------------------------
// Step-wise layer forward on the matrix-core GEMM, for hidden widths the persistent kernels do not take (H > 256: the
// raw-TensorFlow model's LSTMCell(400), mycode/lstm.py:59,128-132).  zx = x K for ALL steps is one product; a step is
// zr = h_{t-1} R (one product over all sequences, every CU takes a tile) + one pointwise launch (gates, c, h, tape).
// The generic kernel walks the steps inside one launch but streams all of K and R through every workgroup in every
// step on the VALU: 6.3 ms for two layers at (32, 10, 90, H = 400); this path is launch-bound at ~3 launches a step.
// ---------------------------------------------------------------------------------------------------------------
template <int ACT>
__global__ __launch_bounds__(256) void lstm_pointwise_fwd_kernel(const float* __restrict__ zx, long ldzx, const float* __restrict__ zr,
                                                                 const float* __restrict__ b, const float* __restrict__ c_in,
                                                                 float* __restrict__ c_out, float* __restrict__ h_out, long ldh,
                                                                 float* __restrict__ reserve, long ldres, float* __restrict__ hT,
                                                                 float* __restrict__ cT, int B, int H) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * H) return;
    const int row = (int)(i / H), u = (int)(i - (long)row * H);
    const float* zp = zx + (size_t)row * ldzx + u;
    const float* rp = zr ? zr + (size_t)row * 4 * H + u : nullptr;
    float z[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) z[g] = zp[g * H] + b[g * H + u] + (rp ? rp[g * H] : 0.f);
    const float ig = rec_act<ACT>(z[0]), fg = rec_act<ACT>(z[1]), gg = tanh_f(z[2]), og = rec_act<ACT>(z[3]);
    const float c = fmaf(fg, c_in ? c_in[i] : 0.f, ig * gg);
    const float h = og * tanh_f(c);
    c_out[i] = c;
    h_out[(size_t)row * ldh + u] = h;
    if (reserve) {
        float* q = reserve + (size_t)row * ldres + u;
        q[0] = ig; q[H] = fg; q[2 * H] = gg; q[3 * H] = og; q[4 * H] = c;
    }
    if (hT) hT[i] = h;
    if (cT) cT[i] = c;
}

bool stepwise_preferred(int B, int F, int H) { return B > 0 && F > 0 && H > 256; }

size_t stepwise_workspace_floats(int B, int T, int H) {
    return (size_t)B * T * 4 * H + (size_t)B * 4 * H + (size_t)3 * B * H + ((size_t)1 << 20) + 64;
}

int launch_stepwise(const LstmParams& p, float* ws, size_t ws_floats, hipStream_t stream) {
    const int B = p.B, T = p.T, F = p.F, H = p.H;
    if (B == 0) return FOV_OK;
    if (ws_floats < stepwise_workspace_floats(B, T, H)) { set_error("step-wise layer: workspace too small"); return FOV_ERR_WORKSPACE; }
    const size_t bh = (size_t)B * H;
    if (T == 0) {
        const size_t nb = sizeof(float) * bh;
        if (p.hT) (void)(p.h0 ? hipMemcpyAsync(p.hT, p.h0, nb, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(p.hT, 0, nb, stream));
        if (p.cT) (void)(p.c0 ? hipMemcpyAsync(p.cT, p.c0, nb, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(p.cT, 0, nb, stream));
        return FOV_OK;
    }
    float* zx = ws;
    float* zr = zx + (size_t)B * T * 4 * H;
    float* cbuf = zr + (size_t)B * 4 * H;
    float* hbuf[2] = {cbuf + bh, cbuf + 2 * bh};
    float* scratch = cbuf + 3 * bh;
    const size_t scratch_floats = ws_floats - (size_t)(scratch - ws);
    // zx (B*T, 4H) = x (B*T, F) . K (F, 4H)
    int rc = matmul_f32(p.x, p.K, zx, B * T, F, 4 * H, scratch, scratch_floats, stream);
    if (rc) return rc;
    const dim3 pgrid((unsigned)((bh + 255) / 256));
    const float* hprev = p.h0;
    long ldhp = H;
    const float* cprev = p.c0;
    for (int t = 0; t < T; ++t) {
        const float* zrp = nullptr;
        if (hprev) {   // zr (B, 4H) = h_{t-1} (B, H) . R (H, 4H)
            GemmArgs g = {};
            g.a = hprev; g.b = p.R; g.c = zr; g.M = B; g.N = 4 * H; g.KO = 1; g.KI = H;
            g.a_sm = ldhp; g.a_ski = 1; g.b_sn = 1; g.b_ski = 4 * H; g.ldc = 4 * H;
            rc = gemm_f32(g, 0, scratch, scratch_floats, stream);
            if (rc) return rc;
            zrp = zr;
        }
        float* hout = p.hs ? p.hs + (size_t)t * H : hbuf[t & 1];
        const long ldh = p.hs ? (long)T * H : H;
        const bool last = (t == T - 1);
        float* res = p.reserve ? p.reserve + (size_t)t * 5 * H : nullptr;
        if (p.act == FOV_ACT_HARD_SIGMOID)
            hipLaunchKernelGGL(lstm_pointwise_fwd_kernel<FOV_ACT_HARD_SIGMOID>, pgrid, dim3(256), 0, stream, zx + (size_t)t * 4 * H,
                               (long)T * 4 * H, zrp, p.b, cprev, cbuf, hout, ldh, res, (long)T * 5 * H, last ? p.hT : nullptr,
                               last ? p.cT : nullptr, B, H);
        else
            hipLaunchKernelGGL(lstm_pointwise_fwd_kernel<FOV_ACT_SIGMOID>, pgrid, dim3(256), 0, stream, zx + (size_t)t * 4 * H,
                               (long)T * 4 * H, zrp, p.b, cprev, cbuf, hout, ldh, res, (long)T * 5 * H, last ? p.hT : nullptr,
                               last ? p.cT : nullptr, B, H);
        rc = check_launch("lstm_pointwise_fwd");
        if (rc) return rc;
        hprev = hout;
        ldhp = ldh;
        cprev = cbuf;      // in place: element i is read and written by the same thread
    }
    return FOV_OK;
}

// [dK ; dR ; db] = [A1 | A2 | 1]^T B as ONE product and one reduce: c is a dense (M1 + M2 + bias_row, N) matrix - the
// layout of a layer's kernel, recurrent kernel and bias in a trainer's flat gradient buffer.  Rows of the product are
// (ro, ri) pairs; a shifted operand presents its element (ro, ri - 1) at (ro, ri) and zero at ri == 0 (h_{t-1} read
// from the (B,T,H) tape of h_t).  A2 may be NULL.
bool wgrad_fusable(const float* a1, long lda1, long a1_so, int M1, const float* a2, long lda2, long a2_so, int M2, const float* b,
                   long ldb, long b_so, const float* c, int N) {
    auto al = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
    if (!al(a1) || !al(b) || !al(c) || (lda1 & 3) || (a1_so & 3) || (M1 & 3) || (ldb & 3) || (b_so & 3) || (N & 3) || M1 <= 0) return false;
    if (a2 && (!al(a2) || (lda2 & 3) || (a2_so & 3) || (M2 & 3) || (M1 % 128) || M2 <= 0)) return false;
    return M1 + (a2 ? M2 : 0) > 96;   // the 128-row tile shape (gemm_f32) / the 128 x 128 bf16 tile
}

int wgrad_fused(const float* a1, long lda1, long a1_so, int M1, int shift1, const float* a2, long lda2, long a2_so, int M2, int shift2,
                const float* b, long ldb, long b_so, float* c, int N, int RO, int RI, int bias_row, int accumulate, int bf16,
                float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (bf16)
        return gemm_bf16_tn_fused(a1, lda1, a1_so, M1, shift1, a2, lda2, a2_so, M2, shift2, b, ldb, b_so, c, N, N, RO, RI, bias_row,
                                  accumulate, scratch, scratch_floats, stream);
    GemmArgs g = {};
    g.a = a1; g.b = b; g.c = c; g.M = M1 + (a2 ? M2 : 0); g.N = N; g.KO = RO; g.KI = RI;
    g.a_sm = 1; g.a_sko = a1_so; g.a_ski = lda1; g.b_sn = 1; g.b_sko = b_so; g.b_ski = ldb; g.ldc = N;
    g.a2 = a2; g.a2_sko = a2_so; g.a2_ski = lda2; g.M1 = M1; g.bias_row = bias_row ? 1 : 0;
    if (shift1 || shift2) {
        // the (ro, ri) rows are kept flat - full 16-row k-tiles instead of tiles that end with every ro row - and the
        // shifted operand is masked where ri == 0; needs rows that are contiguous across ro
        if (a1_so != (long)RI * lda1 || (a2 && a2_so != (long)RI * lda2) || b_so != (long)RI * ldb) {
            set_error("wgrad_fused: a shifted operand needs (ro, ri) rows that are contiguous");
            return FOV_ERR_UNSUPPORTED;
        }
        g.KO = 1; g.KI = RO * RI; g.a_sko = g.a2_sko = g.b_sko = 0;
        g.a_period = shift1 ? RI : 0;
        g.a2_period = shift2 ? RI : 0;
    }
    return gemm_f32(g, accumulate, scratch, scratch_floats, stream);
}

// The weight-gradient half of a layer's BPTT, from its dz tape: dK = x^T dz, dR = h_{t-1}^T dz (h_{-1} = h0 or 0), db = colsum(dz).
// fuse_kr / fuse_r: the adjacent-in-memory forms (one product for all three / for dR and db); db_done: the caller's persistent
// kernel has produced db already (unfused forms only).
static int lstm_seq_weight_products(const float* x, const float* hs, const float* h0, const float* dz, float* dK, float* dR, float* db,
                                    int B, int T, int F, int H, int accumulate, int bf16, bool fuse_kr, bool fuse_r, bool db_done,
                                    float* scratch, size_t scratch_floats, hipStream_t stream) {
    const bool persistent = db_done;
    int rc;
    const long BT = (long)B * T;
    if (fuse_kr || fuse_r) {
        rc = fuse_kr ? wgrad_fused(x, F, (long)T * F, F, 0, hs, H, (long)T * H, H, 1, dz, 4 * H, (long)T * 4 * H, dK, 4 * H, B, T, 1,
                                   accumulate, bf16, scratch, scratch_floats, stream)
                     : wgrad_fused(hs, H, (long)T * H, H, 1, nullptr, 0, 0, 0, 0, dz, 4 * H, (long)T * 4 * H, dR, 4 * H, B, T, 1,
                                   accumulate, bf16, scratch, scratch_floats, stream);
        if (rc) return rc;
        if (h0) {   // dR += h0^T dz[:, 0]
            GemmArgs g0 = {};
            g0.a = h0; g0.b = dz; g0.c = dR; g0.M = H; g0.N = 4 * H; g0.KO = 1; g0.KI = B;
            g0.a_sm = 1; g0.a_ski = H; g0.b_sn = 1; g0.b_ski = (long)T * 4 * H; g0.ldc = 4 * H;
            rc = bf16 ? gemm_bf16_tn(h0, H, 0, dz, (long)T * 4 * H, 0, dR, 4 * H, H, 4 * H, 1, B, 1, scratch, scratch_floats, stream)
                      : gemm_f32(g0, 1, scratch, scratch_floats, stream);
            if (rc) return rc;
        }
        if (fuse_kr) dK = nullptr;
        dR = db = nullptr;
    }
    if (dK) {   // dK (F,4H) = x^T dz : A(m=f,k=(b,t)) = x[k][f], B(k,n) = dz[k][n]
        GemmArgs g = {};
        g.a = x; g.b = dz; g.c = dK; g.M = F; g.N = 4 * H; g.KO = 1; g.KI = (int)BT;
        g.a_sm = 1; g.a_ski = F; g.b_sn = 1; g.b_ski = 4 * H; g.ldc = 4 * H;
        rc = skinny_tn(x, F, F, dz, 4 * H, 4 * H, BT, dK, 4 * H, 1, accumulate, scratch, scratch_floats, stream);
        if (rc == 0) rc = bf16 ? gemm_bf16_tn(x, F, 0, dz, 4 * H, 0, dK, 4 * H, F, 4 * H, 1, (int)BT, accumulate, scratch, scratch_floats, stream)
                               : gemm_f32(g, accumulate, scratch, scratch_floats, stream);
        if (rc < 0) return rc;
    }
    if (dR) {   // dR (H,4H) = sum_b sum_{t>=1} hs[b][t-1]^T dz[b][t]  +  h0^T dz[:,0]
        GemmArgs g = {};
        g.a = hs; g.b = dz + (size_t)4 * H; g.c = dR; g.M = H; g.N = 4 * H; g.KO = B; g.KI = T - 1;
        g.a_sm = 1; g.a_sko = (long)T * H; g.a_ski = H;
        g.b_sn = 1; g.b_sko = (long)T * 4 * H; g.b_ski = 4 * H; g.ldc = 4 * H;
        if (T > 1) {
            rc = bf16 ? gemm_bf16_tn(hs, H, (long)T * H, dz + (size_t)4 * H, 4 * H, (long)T * 4 * H, dR, 4 * H, H, 4 * H, B, T - 1,
                                     accumulate, scratch, scratch_floats, stream)
                      : gemm_f32(g, accumulate, scratch, scratch_floats, stream);
            if (rc) return rc;
        } else if (!accumulate) {
            (void)hipMemsetAsync(dR, 0, sizeof(float) * (size_t)H * 4 * H, stream);
        }
        if (h0 && T > 0) {
            GemmArgs g0 = {};
            g0.a = h0; g0.b = dz; g0.c = dR; g0.M = H; g0.N = 4 * H; g0.KO = 1; g0.KI = B;
            g0.a_sm = 1; g0.a_ski = H; g0.b_sn = 1; g0.b_ski = (long)T * 4 * H; g0.ldc = 4 * H;
            rc = bf16 ? gemm_bf16_tn(h0, H, 0, dz, (long)T * 4 * H, 0, dR, 4 * H, H, 4 * H, 1, B, 1, scratch, scratch_floats, stream)
                      : gemm_f32(g0, 1, scratch, scratch_floats, stream);
            if (rc) return rc;
        }
    }
    if (db && !persistent) {
        rc = colsum(dz, db, BT, 4 * H, accumulate, scratch, scratch_floats, stream);
        if (rc) return rc;
    }
    return FOV_OK;
}

// few rows (batch x time <= ~1000), fp32: all of a layer's weight gradients by wgrad_group.hip - one launch, no split, no reduce
static bool lstm_seq_wgrad_grouped(int bf16, int B, int T, int H, const float* dz, float* dK, float* dR, float* db) {
    return !bf16 && wgrad_group_takes(B, T, H) && (dK || dR) && (((uintptr_t)dz) & 15) == 0 && (!db || dK || dR);
}

// few rows AND narrow layers (the reference's batch of 32 at H <= 256): wgrad_rows_kernel, 16 x 64 tiles, rows split over the waves
static bool lstm_seq_wgrad_rows(int bf16, int B, int T, int F, int H, const float* dz, float* dK, float* dR, float* db) {
    return !bf16 && wgrad_rows_takes((long)B * T, H) && F <= 2048 && (dK || dR) && (!db || dK || dR) &&
           ((((uintptr_t)dz) | ((uintptr_t)dK) | ((uintptr_t)dR) | ((uintptr_t)db)) & 15) == 0;
}

static int wgrad_few_rows(bool rows_form, const float* x, const float* hs, const float* h0, const float* dz, float* dK, float* dR, float* db,
                          int B, int T, int F, int H, int accumulate, hipStream_t stream) {
    return rows_form ? wgrad_rows_layers(1, &x, &F, &T, &hs, &h0, &dz, &dK, &dR, &db, B, H, accumulate, stream)
                     : wgrad_group_layers(1, &x, &F, &hs, &h0, &dz, &dK, &dR, &db, B, T, H, accumulate, stream);
}

static void lstm_seq_wgrad_fusion(const float* x, const float* hs, const float* dz, float* dK, float* dR, float* db, int T, int F, int H,
                                  bool* fuse_kr, bool* fuse_r) {
    const int N4 = 4 * H;
    *fuse_kr = dK && dR && db && T > 1 && dR == dK + (size_t)F * N4 && db == dR + (size_t)H * N4 &&
               wgrad_fusable(x, F, (long)T * F, F, hs, H, (long)T * H, H, dz, N4, (long)T * N4, dK, N4) && !env_knobs().no_wgrad_fusion;
    *fuse_r = !*fuse_kr && dR && db && T > 1 && db == dR + (size_t)H * N4 &&
              wgrad_fusable(hs, H, (long)T * H, H, nullptr, 0, 0, 0, dz, N4, (long)T * N4, dR, N4) && !env_knobs().no_wgrad_fusion;
}

// Weight gradients of one layer from a dz tape its BPTT left behind (fov_lstm_seq_bwd* with dK = dR = db = NULL): lets a trainer
// put a layer's products on another stream than the next layer's recurrence.  Same arithmetic, same order as inside lstm_seq_bwd.
int lstm_seq_wgrad(const float* x, const float* hs, const float* h0, const float* dz, float* dK, float* dR, float* db, int B, int T,
                   int F, int H, int accumulate, int bf16, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (bf16 && H != 256) { set_error("lstm_seq_wgrad: the bf16 path is built for H = 256"); return FOV_ERR_UNSUPPORTED; }
    if (T == 0 || B == 0) {
        if (!accumulate) {
            if (dK) (void)hipMemsetAsync(dK, 0, sizeof(float) * (size_t)F * 4 * H, stream);
            if (dR) (void)hipMemsetAsync(dR, 0, sizeof(float) * (size_t)H * 4 * H, stream);
            if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)4 * H, stream);
        }
        return FOV_OK;
    }
    const bool rows_form = lstm_seq_wgrad_rows(bf16, B, T, F, H, dz, dK, dR, db);
    if (rows_form || lstm_seq_wgrad_grouped(bf16, B, T, H, dz, dK, dR, db))
        return wgrad_few_rows(rows_form, x, hs, h0, dz, dK, dR, db, B, T, F, H, accumulate, stream);
    bool fuse_kr, fuse_r;
    lstm_seq_wgrad_fusion(x, hs, dz, dK, dR, db, T, F, H, &fuse_kr, &fuse_r);
    return lstm_seq_weight_products(x, hs, h0, dz, dK, dR, db, B, T, F, H, accumulate, bf16, fuse_kr, fuse_r, false, scratch,
                                    scratch_floats, stream);
}

// Weight gradients of TWO layers that share the batch but not the time length (an encoder and its decoder: FoV_seq2seq.py:68-93)
// from the dz tapes their BPTT calls left.  At the reference's batch all six gradients are ONE launch (wgrad_rows_kernel).
bool lstm_seq_wgrad_pair_one_launch(int B, int T1, int T2, int H) {
    return T1 > 0 && T2 > 0 && wgrad_rows_takes((long)B * (T1 > T2 ? T1 : T2), H);
}

int lstm_seq_wgrad_pair(const float* x1, const float* hs1, const float* h0_1, const float* dz1, float* dK1, float* dR1, float* db1, int T1, int F1,
                        const float* x2, const float* hs2, const float* h0_2, const float* dz2, float* dK2, float* dR2, float* db2, int T2, int F2,
                        int B, int H, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (B > 0 && T1 > 0 && T2 > 0 && lstm_seq_wgrad_rows(0, B, T1, F1, H, dz1, dK1, dR1, db1) && lstm_seq_wgrad_rows(0, B, T2, F2, H, dz2, dK2, dR2, db2)) {
        const float* xs[2] = {x1, x2};
        const int Fs[2] = {F1, F2}, Ts[2] = {T1, T2};
        const float* hss[2] = {hs1, hs2};
        const float* h0s[2] = {h0_1, h0_2};
        const float* dzs[2] = {dz1, dz2};
        float* dKs[2] = {dK1, dK2};
        float* dRs[2] = {dR1, dR2};
        float* dbs[2] = {db1, db2};
        return wgrad_rows_layers(2, xs, Fs, Ts, hss, h0s, dzs, dKs, dRs, dbs, B, H, accumulate, stream);
    }
    int rc = lstm_seq_wgrad(x1, hs1, h0_1, dz1, dK1, dR1, db1, B, T1, F1, H, accumulate, 0, scratch, scratch_floats, stream);
    if (rc) return rc;
    return lstm_seq_wgrad(x2, hs2, h0_2, dz2, dK2, dR2, db2, B, T2, F2, H, accumulate, 0, scratch, scratch_floats, stream);
}

// BPTT of one layer.  dz:(B,T,4H) is an output (kept: the caller may need it for dx of the layer
// below).  dK,dR,db are overwritten (accumulate = 0) or added to (accumulate = 1).
int lstm_seq_bwd(const float* x, const float* K, const float* R, const float* h0, const float* c0, const float* hs,
                 const float* reserve, const float* dhs, const float* dhT, const float* dcT, float* dz, float* dx,
                 float* dK, float* dR, float* db, float* dh0, float* dc0, int B, int T, int F, int H, int act,
                 int accumulate, float* ws, size_t ws_floats, hipStream_t stream, int bf16) {
    if (bf16 && H != 256) { set_error("lstm_seq_bwd: the bf16 path is built for H = 256"); return FOV_ERR_UNSUPPORTED; }
    if (ws_floats < lstm_bwd_workspace_floats(B, T, F, H)) { set_error("lstm_seq_bwd: workspace too small"); return FOV_ERR_WORKSPACE; }
    if (T == 0) {
        if (!accumulate) {
            if (dK) (void)hipMemsetAsync(dK, 0, sizeof(float) * (size_t)F * 4 * H, stream);
            if (dR) (void)hipMemsetAsync(dR, 0, sizeof(float) * (size_t)H * 4 * H, stream);
            if (db) (void)hipMemsetAsync(db, 0, sizeof(float) * (size_t)4 * H, stream);
        }
        const size_t bh0 = sizeof(float) * (size_t)B * H;
        if (dh0) (void)(dhT ? hipMemcpyAsync(dh0, dhT, bh0, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(dh0, 0, bh0, stream));
        if (dc0) (void)(dcT ? hipMemcpyAsync(dc0, dcT, bh0, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(dc0, 0, bh0, stream));
        return FOV_OK;
    }
    const bool wide16 = !bf16 && bwd16_takes(B, H) && (((uintptr_t)R) & 15) == 0;   // width 512, small batches at 128 / 256: lstm_bwd16.hip
    const bool persistent = (bwd_cluster_shape_ok(H) || wide16) && !env_knobs().bwd_stepped;
    bool fuse_kr = false, fuse_r = false, dx_in_kernel = false, grouped = false, rows_form = false;
    const size_t head = (kStatusBytes + kXchBytes) / sizeof(float);
    float* dh_rec = ws + head;
    float* dc = dh_rec + (size_t)B * H;
    float* scratch = dc + (size_t)B * H;
    const size_t scratch_floats = ws_floats - head - (size_t)2 * B * H;
    const size_t bh = sizeof(float) * (size_t)B * H;
    hipError_t e = hipSuccess;
    if (persistent) {
        // one launch for the whole recurrence: dz (B,T,4H), dh0, dc0
        // bias gradient: per-tile partials from the kernel (db_part lives in the split-K scratch), summed below
        // dK, dR, db adjacent (a trainer's flat gradient buffer): ONE product [x | h_{t-1} | 1]^T dz (h_{-1} = 0; a given initial
        // state adds its h0^T dz_0 afterwards)
        // below gives all three (no bias partials from the kernel, no column-sum launches); F too narrow for a row tile:
        // [h_{t-1} | 1]^T dz gives dR and db
        lstm_seq_wgrad_fusion(x, hs, dz, dK, dR, db, T, F, H, &fuse_kr, &fuse_r);
        rows_form = lstm_seq_wgrad_rows(bf16, B, T, F, H, dz, dK, dR, db);
        grouped = rows_form || lstm_seq_wgrad_grouped(bf16, B, T, H, dz, dK, dR, db);
        if (grouped) fuse_kr = fuse_r = false;
        float* db_part = (db && !fuse_kr && !fuse_r && !grouped) ? scratch : nullptr;
        // bf16, 256-wide input (the stacked layer): the BPTT kernel forms dx = dz K^T from the dz tile it has gathered anyway
        dx_in_kernel = bf16 && dx && F == 256 && (((uintptr_t)K) & 15) == 0 && !env_knobs().no_dx_fusion;
        // fp32 on the eight-workgroup kernel (round 5): every workgroup forms its gate columns' share of dx in the shadow of the
        // exchange; eight partial tapes in the scratch, one reduce launch (was: a GEMM + reduce behind the recurrence)
        const bool on_bwd8 = !wide16 && !bf16 && bwd8_preferred(B, H) && !env_knobs().bwd_groups4;
        float* dx_parts = nullptr;
        if (on_bwd8 && dx && F == 256 && (((uintptr_t)K) & 15) == 0 && (((uintptr_t)dx) & 15) == 0 && !env_knobs().no_dx_fusion) {
            const size_t off = ((size_t)((B + 15) / 16) * 4 * H + 63) & ~(size_t)63;      // behind the db partials
            if (off + (size_t)8 * B * T * 256 <= scratch_floats) { dx_parts = scratch + off; dx_in_kernel = true; }
        }
        // H = 256: groups of eight workgroups fill the chip up to 32 tiles (the 4-group kernel leaves half of it idle
        // at 512 sequences); bf16 operands exist in the 8-group kernel only
        if (env_knobs().dbg_trace) fprintf(stderr, "[fov trace] lstm_seq_bwd B=%d T=%d F=%d H=%d wide16=%d fuse_kr=%d fuse_r=%d grouped=%d acc=%d: recurrence next\n", B, T, F, H, (int)wide16, (int)fuse_kr, (int)fuse_r, (int)grouped, accumulate);
        int rc = wide16 ? launch_bwd16(R, reserve, c0, dhs, dhT, dcT, dz, dh0 ? dh0 : dh_rec, dc0 ? dc0 : dc, db_part, B, T, H, act, ws, stream)
                 : (bf16 || (bwd8_preferred(B, H) && !env_knobs().bwd_groups4))
                     ? launch_bwd8(R, reserve, c0, dhs, dhT, dcT, dz, dh0 ? dh0 : dh_rec, dc0 ? dc0 : dc, db_part, B, T, act, bf16, ws, stream,
                                   dx_in_kernel ? K : nullptr, dx_in_kernel ? (dx_parts ? dx_parts : dx) : nullptr)
                     : launch_bwd_cluster(R, reserve, c0, dhs, dhT, dcT, dz, dh0 ? dh0 : dh_rec, dc0 ? dc0 : dc, db_part, B, T, H,
                                          act, ws, stream);
        if (rc) return rc;
        if (dx_parts) {      // dx = the eight workgroups' shares in slice order (before anything else touches the scratch)
            rc = splitk_reduce(dx_parts, dx, (long)B * T * 256, 8, 0, stream);
            if (rc) return rc;
        }
        if (env_knobs().dbg_trace) fprintf(stderr, "[fov trace] lstm_seq_bwd: recurrence done\n");
        if (db_part) {
            const int tiles = (B + 15) / 16;
            if ((size_t)tiles * 4 * H + (size_t)256 * 4 * H > scratch_floats) { set_error("lstm_seq_bwd: scratch too small for db"); return FOV_ERR_WORKSPACE; }
            rc = colsum(db_part, db, tiles, 4 * H, accumulate, scratch + (size_t)tiles * 4 * H,
                        scratch_floats - (size_t)tiles * 4 * H, stream);
            if (rc) return rc;
        }
    } else {
        e = dhT ? hipMemcpyAsync(dh_rec, dhT, bh, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(dh_rec, 0, bh, stream);
        if (e == hipSuccess) e = dcT ? hipMemcpyAsync(dc, dcT, bh, hipMemcpyDeviceToDevice, stream) : hipMemsetAsync(dc, 0, bh, stream);
        if (e != hipSuccess) { set_error("lstm_seq_bwd init: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
        const long nelem = (long)B * H;
        const dim3 pgrid((unsigned)((nelem + 255) / 256));
        for (int t = T - 1; t >= 0; --t) {
            if (act == FOV_ACT_HARD_SIGMOID)
                hipLaunchKernelGGL(lstm_bwd_pointwise<FOV_ACT_HARD_SIGMOID>, pgrid, dim3(256), 0, stream, reserve, c0, dhs,
                                   dh_rec, dc, dz, B, T, H, t);
            else
                hipLaunchKernelGGL(lstm_bwd_pointwise<FOV_ACT_SIGMOID>, pgrid, dim3(256), 0, stream, reserve, c0, dhs, dh_rec,
                                   dc, dz, B, T, H, t);
            int rc = check_launch("lstm_bwd_pointwise");
            if (rc) return rc;
            // dh_rec (B,H) = dz_t (B,4H) . R^T :  A(m,k) = dz[m][t][k], B(k,n) = R[n][k]
            GemmArgs g = {};
            g.a = dz + (size_t)t * 4 * H; g.b = R; g.c = dh_rec;
            g.M = B; g.N = H; g.KO = 1; g.KI = 4 * H;
            g.a_sm = (long)T * 4 * H; g.a_sko = 0; g.a_ski = 1;
            g.b_sn = 4 * H; g.b_sko = 0; g.b_ski = 1;
            g.ldc = H;
            rc = gemm_f32(g, 0, scratch, scratch_floats, stream);
            if (rc) return rc;
        }
        if (dh0) { e = hipMemcpyAsync(dh0, dh_rec, bh, hipMemcpyDeviceToDevice, stream); if (e != hipSuccess) { set_error("dh0 copy"); return FOV_ERR_LAUNCH; } }
        if (dc0) { e = hipMemcpyAsync(dc0, dc, bh, hipMemcpyDeviceToDevice, stream); if (e != hipSuccess) { set_error("dc0 copy"); return FOV_ERR_LAUNCH; } }
    }
    int rc = grouped ? wgrad_few_rows(rows_form, x, hs, h0, dz, dK, dR, db, B, T, F, H, accumulate, stream)
                     : lstm_seq_weight_products(x, hs, h0, dz, dK, dR, db, B, T, F, H, accumulate, bf16, fuse_kr, fuse_r, /*db_done=*/persistent,
                                                scratch, scratch_floats, stream);
    if (rc) return rc;
    if (dx && !dx_in_kernel) {   // dx (B*T,F) = dz . K^T : A(m,k) = dz[m][k], B(k,n) = K[n][k]
        const long BT = (long)B * T;
        GemmArgs g = {};
        g.a = dz; g.b = K; g.c = dx; g.M = (int)BT; g.N = F; g.KO = 1; g.KI = 4 * H;
        g.a_sm = 4 * H; g.a_ski = 1; g.b_sn = 4 * H; g.b_ski = 1; g.ldc = F;
        rc = (bf16 && (((uintptr_t)K) & 15) == 0) ? gemm_bf16_nt(dz, 4 * H, K, 4 * H, dx, F, (int)BT, F, 4 * H, stream)
                                                 : gemm_f32(g, 0, scratch, scratch_floats, stream);
        if (rc) return rc;
    }
    return FOV_OK;
}

// BPTT of two stacked width-512 layers in ONE launch (lstm_bwd16.hip: both recurrences and dx = dz2 . K2^T between them as three
// roles of one persistent kernel), then the two layers' weight-gradient products.  Same outputs as two lstm_seq_bwd calls
// (upper layer first, its dx as the lower layer's dhs) except that the intermediate dx is not materialised.
size_t lstm_stack2_bwd_workspace_floats(int B, int T, int F, int H) {
    const size_t tiles = (size_t)(B + 15) / 16;
    return lstm_bwd_workspace_floats(B, T, F > H ? F : H, H) + 2 * tiles * 4 * H + (size_t)4 * B * H + 64;
}

int lstm_stack2_bwd(const float* x, const float* R1, const float* K2, const float* R2, const float* h0_1, const float* c0_1,
                    const float* h0_2, const float* c0_2, const float* hs1, const float* res1, const float* hs2, const float* res2,
                    const float* dhs2, const float* dhT2, const float* dcT2, const float* dhT1, const float* dcT1, float* dz1, float* dz2,
                    float* dK1, float* dR1, float* db1, float* dK2, float* dR2, float* db2, float* dh0_1, float* dc0_1, float* dh0_2,
                    float* dc0_2, int B, int T, int F, int H, int act, int accumulate, float* ws, size_t ws_floats, hipStream_t stream) {
    if (!bwd16_pair_shape(B, T, H)) { set_error("lstm_stack2_bwd: unsupported shape (H = 512, <= 32 sequences on 256 CUs)"); return FOV_ERR_UNSUPPORTED; }
    if (ws_floats < lstm_stack2_bwd_workspace_floats(B, T, F, H)) { set_error("lstm_stack2_bwd: workspace too small"); return FOV_ERR_WORKSPACE; }
    const size_t head = (kStatusBytes + kXchBytes) / sizeof(float);
    const size_t tiles = (size_t)(B + 15) / 16, bh = (size_t)B * H;
    float* dbp2 = ws + head;
    float* dbp1 = dbp2 + tiles * 4 * H;
    float* spare = dbp1 + tiles * 4 * H;          // state gradients the caller does not ask for
    float* scratch = spare + 4 * bh;
    const size_t scratch_floats = ws_floats - head - 2 * tiles * 4 * H - 4 * bh;
    bool kr1, r1, kr2, r2;
    lstm_seq_wgrad_fusion(x, hs1, dz1, dK1, dR1, db1, T, F, H, &kr1, &r1);
    lstm_seq_wgrad_fusion(hs1, hs2, dz2, dK2, dR2, db2, T, H, H, &kr2, &r2);
    // few rows: ALL weight gradients of both layers in one launch behind the recurrences (wgrad_group.hip)
    const bool rows_form = lstm_seq_wgrad_rows(0, B, T, F, H, dz1, dK1, dR1, db1) && lstm_seq_wgrad_rows(0, B, T, H, H, dz2, dK2, dR2, db2);
    const bool grouped = rows_form || (lstm_seq_wgrad_grouped(0, B, T, H, dz1, dK1, dR1, db1) && lstm_seq_wgrad_grouped(0, B, T, H, dz2, dK2, dR2, db2));
    if (grouped) kr1 = r1 = kr2 = r2 = true;      // (no bias partials from the kernel)
    float* db_part1 = (db1 && !kr1 && !r1) ? dbp1 : nullptr;
    float* db_part2 = (db2 && !kr2 && !r2) ? dbp2 : nullptr;
    int rc = launch_bwd16_pair(R2, K2, res2, c0_2, dhs2, dhT2, dcT2, dz2, dh0_2 ? dh0_2 : spare, dc0_2 ? dc0_2 : spare + bh, db_part2, R1,
                               res1, c0_1, dhT1, dcT1, dz1, dh0_1 ? dh0_1 : spare + 2 * bh, dc0_1 ? dc0_1 : spare + 3 * bh, db_part1, B,
                               T, act, ws, stream);
    if (rc) return rc;
    if (grouped) {
        const float* xs[2] = {x, hs1};
        const int Fs[2] = {F, H};
        const float* hss[2] = {hs1, hs2};
        const float* h0s[2] = {h0_1, h0_2};
        const float* dzs[2] = {dz1, dz2};
        float* dKs[2] = {dK1, dK2};
        float* dRs[2] = {dR1, dR2};
        float* dbs[2] = {db1, db2};
        const int Ts[2] = {T, T};
        return rows_form ? wgrad_rows_layers(2, xs, Fs, Ts, hss, h0s, dzs, dKs, dRs, dbs, B, H, accumulate, stream)
                         : wgrad_group_layers(2, xs, Fs, hss, h0s, dzs, dKs, dRs, dbs, B, T, H, accumulate, stream);
    }
    for (int l = 2; l >= 1; --l) {
        float* part = l == 2 ? db_part2 : db_part1;
        float* db = l == 2 ? db2 : db1;
        if (part) {
            if (tiles * 4 * H + (size_t)256 * 4 * H > scratch_floats) { set_error("lstm_stack2_bwd: scratch too small for db"); return FOV_ERR_WORKSPACE; }
            rc = colsum(part, db, (long)tiles, 4 * H, accumulate, scratch, scratch_floats, stream);
            if (rc) return rc;
        }
        rc = l == 2 ? lstm_seq_weight_products(hs1, hs2, h0_2, dz2, dK2, dR2, db2, B, T, H, H, accumulate, 0, kr2, r2, true, scratch, scratch_floats, stream)
                    : lstm_seq_weight_products(x, hs1, h0_1, dz1, dK1, dR1, db1, B, T, F, H, accumulate, 0, kr1, r1, true, scratch, scratch_floats, stream);
        if (rc) return rc;
    }
    return FOV_OK;
}

// Dense backward given dpre (N,Out): dW (In,Out) = x^T dpre, db = colsum(dpre), dx (N,In) = dpre W^T
int dense_bwd(const float* x, const float* W, const float* dpre, float* dx, float* dW, float* db, int N, int In, int Out,
              int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream, int bf16) {
    int rc;
    if (dW && bf16 && Out >= 64 && (Out & 3) == 0 && (((uintptr_t)dpre) & 15) == 0) {
        // dW (In,Out) = x^T dpre with bf16 operands (the unrolled decoder's K / R gradients: one product over all steps)
        rc = gemm_bf16_tn(x, In, 0, dpre, Out, 0, dW, Out, In, Out, 1, N, accumulate, scratch, scratch_floats, stream);
        if (rc) return rc;
        dW = nullptr;
    }
    if (dW) {
        GemmArgs g = {};
        g.a = x; g.b = dpre; g.c = dW; g.M = In; g.N = Out; g.KO = 1; g.KI = N;
        g.a_sm = 1; g.a_ski = In; g.b_sn = 1; g.b_ski = Out; g.ldc = Out;
        // skinny forms: out[s = output unit][w = input unit] stored transposed into dW (In,Out) for a narrow output (Dense(6)),
        // out[s = input unit][w = output unit] for a narrow input (the decoder LSTM's 6-wide kernel over all steps)
        rc = skinny_tn(dpre, Out, Out, x, In, In, N, dW, 1, Out, accumulate, scratch, scratch_floats, stream);
        if (rc == 0) rc = skinny_tn(x, In, In, dpre, Out, Out, N, dW, Out, 1, accumulate, scratch, scratch_floats, stream);
        if (rc == 0) rc = gemm_f32(g, accumulate, scratch, scratch_floats, stream);
        if (rc < 0) return rc;
    }
    if (db) {
        rc = colsum(dpre, db, N, Out, accumulate, scratch, scratch_floats, stream);
        if (rc) return rc;
    }
    if (dx) {
        GemmArgs g = {};
        g.a = dpre; g.b = W; g.c = dx; g.M = N; g.N = In; g.KO = 1; g.KI = Out;
        g.a_sm = Out; g.a_ski = 1; g.b_sn = Out; g.b_ski = 1; g.ldc = In;
        rc = gemm_f32(g, 0, scratch, scratch_floats, stream);
        if (rc) return rc;
    }
    return FOV_OK;
}

// The ticket word of a loss launch: one slot of the device's ticket table PER STREAM (first use of a stream on a device takes
// the next free slot; the device address of g_loss_tickets is looked up once per device).  Launches on one stream run in
// stream order, so the word a launch finds is the zero its predecessor on that stream left: two launches in flight never share
// a slot however many there are, whichever threads issued them.  Slots are never handed back (a stream handle may still have a
// launch in flight when its owner forgets it): a device on which kLossTickets = 1024 distinct streams have issued loss calls
// gets NULL for further ones - the callers below fall back to the separate sum / column-sum launches, fov_dense_mse_head refuses.
// (A captured graph replayed on two streams AT ONCE would share the slot of its capture stream: not supported.)
static unsigned* loss_ticket_of(hipStream_t stream) {
    struct PerDevice { unsigned* base = nullptr; std::unordered_map<hipStream_t, int> slot; };
    static std::mutex mu;
    static PerDevice devs[64];
    std::lock_guard<std::mutex> lock(mu);
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    PerDevice& d = devs[dev];
    if (!d.base) {
        void* p = nullptr;
        if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_loss_tickets)) != hipSuccess) return nullptr;
        d.base = static_cast<unsigned*>(p);
    }
    auto it = d.slot.find(stream);
    if (it != d.slot.end()) return d.base + it->second;
    if ((int)d.slot.size() == kLossTickets) return nullptr;
    const int idx = (int)d.slot.size();
    d.slot.emplace(stream, idx);
    return d.base + idx;
}

int mse_dense_grad(const float* y, const float* target, float* dpre, float* loss, long n, int activation, float* scratch,
                   size_t scratch_floats, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    const long blocks = (n + 255) / 256;
    if ((size_t)blocks > scratch_floats) { set_error("mse_dense_grad: scratch too small"); return FOV_ERR_WORKSPACE; }
    unsigned* ticket = loss ? loss_ticket_of(stream) : nullptr;
    hipLaunchKernelGGL(mse_dense_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, target, dpre, scratch, n,
                       1.0f / (float)n, activation, 0, 0, 1, ticket, loss, 1.0f / (float)n);
    int rc = check_launch("mse_dense_grad");
    if (rc) return rc;
    if (loss && !ticket) {   // (no ticket slot: the separate sum launch)
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, scratch, loss, (int)blocks, 1.0f / (float)n);
        rc = check_launch("sum_scale");
    }
    return rc;
}

// The same with a weight on the gradient AND the loss (data parallelism: this rank's share n_local / n_global of the
// global mean, so that a SUM all-reduce of gradients and loss gives the global-batch values without a scaling pass)
// and optionally a time-major prediction.
int mse_dense_grad_w(const float* y, const float* target, float* dpre, float* loss, long n, int activation, float weight,
                     int tmB, int tmT, int O, float* scratch, size_t scratch_floats, hipStream_t stream, float* db, int dbO) {
    if (n <= 0) {
        if (db && dbO > 0) (void)hipMemsetAsync(db, 0, sizeof(float) * dbO, stream);
        return FOV_OK;
    }
    const long blocks = (n + 255) / 256;
    if ((size_t)blocks * (db ? 9 : 1) + (db ? 4 : 0) > scratch_floats) { set_error("mse_dense_grad_w: scratch too small"); return FOV_ERR_WORKSPACE; }
    unsigned* ticket = loss ? loss_ticket_of(stream) : nullptr;
    const bool fused_db = db && ticket && dbO >= 1 && dbO <= 8 && tmT == 0 && n % dbO == 0;
    hipLaunchKernelGGL(mse_dense_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, y, target, dpre, scratch, n,
                       weight / (float)n, activation, tmB, tmT, O, ticket, loss, weight / (float)n,
                       fused_db ? scratch + ((blocks + 3) & ~3L) : nullptr, fused_db ? db : nullptr, fused_db ? dbO : 0);
    int rc = check_launch("mse_dense_grad_w");
    if (rc) return rc;
    if (db && !fused_db) {   // (no ticket slot, wide or time-major head: the column-sum launches)
        if (dbO < 1 || n % dbO) { set_error("mse_dense_grad_w: db needs the head's width"); return FOV_ERR_INVALID; }
        rc = colsum(dpre, db, (int)(n / dbO), dbO, 0, scratch + ((blocks + 3) & ~3L), scratch_floats - ((blocks + 3) & ~3L), stream);
        if (rc) return rc;
    }
    if (loss && !ticket) {
        hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, scratch, loss, (int)blocks, weight / (float)n);
        rc = check_launch("sum_scale");
    }
    return rc;
}

bool dense_mse_head_shape_ok(long N, int H, int O) {
    return N >= 1 && N <= (1L << 20) && H >= 4 && H <= 512 && (H & 3) == 0 && O >= 1 && O <= 8;
}
size_t dense_mse_head_scratch_floats(long N, int H, int O) { return (size_t)((N + HD_RB - 1) / HD_RB) * ((size_t)H * O + O + 1) + 64; }
// weight: this rank's share of the global batch (data parallelism), 1 otherwise; loss = weight * mean over all N * O elements
int dense_mse_head(const float* hs, const float* W, const float* b, const float* target, float* y, float* dX, float* dW, float* db,
                   float* loss, long N, int H, int O, int activation, float weight, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (!dense_mse_head_shape_ok(N, H, O)) { set_error("dense_mse_head: rows <= 2^20, H <= 512 (a multiple of 4), O <= 8 only"); return FOV_ERR_UNSUPPORTED; }
    if (dense_mse_head_scratch_floats(N, H, O) > scratch_floats) { set_error("dense_mse_head: scratch too small"); return FOV_ERR_WORKSPACE; }
    unsigned* ticket = loss_ticket_of(stream);
    if (!ticket) { set_error("dense_mse_head: no ticket slot left for this stream (1024 distinct streams have issued loss calls on the device)"); return FOV_ERR_UNSUPPORTED; }
    int rc = defer_touch(dW, (size_t)H * O, stream);     // pending deferred reductions over the gradients written here go first
    if (!rc) rc = defer_touch(db, (size_t)O, stream);
    if (rc) return rc;
    const size_t lds = dense_mse_head_lds_floats(H) * sizeof(float);
    rc = ensure_dynamic_lds((const void*)dense_mse_head_kernel, lds);
    if (rc) return rc;
    const int chunks = (int)((N + HD_RB - 1) / HD_RB);
    const float scale = weight / (float)(N * O);
    if (chunks <= HD_MAXB) {      // one launch: the last block to finish adds the partials
        hipLaunchKernelGGL(dense_mse_head_kernel, dim3((unsigned)chunks), dim3(256), lds, stream, hs, W, b, target, y, dX, scratch, dW, db, loss,
                           (int)N, H, O, activation, scale, ticket, 1, 1);
        return check_launch("dense_mse_head");
    }
    // many rows: about one block per CU walking several tiles (two per CU measured no faster: 39.9 vs 37.1 us at 30 720 rows), then the reduce launch
    const int cus = device_cu_count() > 0 ? device_cu_count() : 256;
    const int cpb = (chunks + cus - 1) / cus;
    const int blocks = (chunks + cpb - 1) / cpb;
    hipLaunchKernelGGL(dense_mse_head_kernel, dim3((unsigned)blocks), dim3(256), lds, stream, hs, W, b, target, y, dX, scratch, dW, db, loss,
                       (int)N, H, O, activation, scale, ticket, cpb, 0);
    rc = check_launch("dense_mse_head");
    if (rc) return rc;
    const int NE = H * O + O + 1;
    hipLaunchKernelGGL(dense_mse_head_reduce_kernel, dim3((unsigned)((NE + 15) / 16)), dim3(256), 0, stream, scratch, blocks, NE, H * O, O, dW, db,
                       loss, scale);
    return check_launch("dense_mse_head_reduce");
}

int scale_inplace(float* x, long n, float s, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    hipLaunchKernelGGL(scale_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, n, s);
    return check_launch("scale");
}

// C (M,N) = A (M,K) . B (K,N), all row-major dense
int matmul_f32(const float* a, const float* b, float* c, int M, int K, int N, float* scratch, size_t scratch_floats,
               hipStream_t stream) {
    GemmArgs g = {};
    g.a = a; g.b = b; g.c = c; g.M = M; g.N = N; g.KO = 1; g.KI = K;
    g.a_sm = K; g.a_ski = 1; g.b_sn = 1; g.b_ski = N; g.ldc = N;
    return gemm_f32(g, 0, scratch, scratch_floats, stream);
}

int splitk_reduce(const float* part, float* out, long n, int S, int accumulate, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    return reduce_or_defer(false, part, out, n, S, accumulate, stream, "splitk_reduce");
}

int act_bwd(const float* dy, const float* y, const float* base, float* out, long n, int activation, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    hipLaunchKernelGGL(act_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, dy, y, base, out, n, activation);
    return check_launch("act_bwd");
}

int adam_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2, float eps,
              const unsigned* const* guards, long long* applied, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    if (((((uintptr_t)p) | ((uintptr_t)g) | ((uintptr_t)m) | ((uintptr_t)v)) & 15) == 0 && n >= 1024)
        hipLaunchKernelGGL(adam_kernel4, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, stream, p, g, m, v, n, lr_t, b1, b2, eps,
                           guards ? guards[0] : nullptr, guards ? guards[1] : nullptr, guards ? guards[2] : nullptr, applied);
    else
        hipLaunchKernelGGL(adam_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, m, v, n, lr_t, b1, b2, eps,
                           guards ? guards[0] : nullptr, guards ? guards[1] : nullptr, guards ? guards[2] : nullptr, applied);
    return check_launch("adam");
}

int rmsprop_step(float* p, const float* g, float* a, long n, float lr, float rho, float eps, const unsigned* const* guards,
                 long long* applied, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    hipLaunchKernelGGL(rmsprop_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, a, n, lr, rho, eps,
                       guards ? guards[0] : nullptr, guards ? guards[1] : nullptr, guards ? guards[2] : nullptr, applied);
    return check_launch("rmsprop");
}

int act_fwd(const float* x, float* y, long n, int activation, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    hipLaunchKernelGGL(act_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, x, y, n, activation);
    return check_launch("act_fwd");
}

int gauss_nll_grad(const float* mu, const float* var, const float* y, float* loss, float* dmu, float* dvar, int B, int Ty,
                   int fps, float scale, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (B <= 0) return FOV_OK;
    if ((size_t)B > scratch_floats) { set_error("gauss_nll_grad: scratch too small"); return FOV_ERR_WORKSPACE; }
    unsigned* ticket = loss ? loss_ticket_of(stream) : nullptr;
    hipLaunchKernelGGL(gauss_nll_kernel, dim3(B), dim3(256), 0, stream, mu, var, y, scratch, dmu, dvar, B, Ty, fps, scale, ticket, loss,
                       scale / (float)B);
    int rc = check_launch("gauss_nll");
    if (rc || !loss || ticket) return rc;
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, scratch, loss, B, scale / (float)B);
    return check_launch("sum_scale");
}

int cce_grad(const float* p, const float* t, float* dp, float* loss, long n_pix, int C, float* scratch, size_t scratch_floats,
             hipStream_t stream) {
    if (n_pix <= 0) return FOV_OK;
    const long blocks = (n_pix + 255) / 256;
    if ((size_t)blocks > scratch_floats) { set_error("categorical_crossentropy_grad: scratch too small"); return FOV_ERR_WORKSPACE; }
    hipLaunchKernelGGL(cce_grad_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, t, dp, scratch, n_pix, C);
    int rc = check_launch("cce_grad");
    if (rc || !loss) return rc;
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, scratch, loss, (int)blocks, 1.0f / (float)n_pix);
    return check_launch("sum_scale");
}

int xyz_sum1_grad(const float* p, float* dp, float* reg, long n_pix, int C, float* scratch, size_t scratch_floats, hipStream_t stream) {
    if (n_pix <= 0) return FOV_OK;
    const long blocks = (n_pix + 255) / 256;
    if ((size_t)blocks > scratch_floats) { set_error("xyz_sum1_grad: scratch too small"); return FOV_ERR_WORKSPACE; }
    hipLaunchKernelGGL(xyz_sum1_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, p, dp, scratch, n_pix, C);
    int rc = check_launch("xyz_sum1");
    if (rc || !reg) return rc;
    hipLaunchKernelGGL(sum_scale_kernel, dim3(1), dim3(256), 0, stream, scratch, reg, (int)blocks, 0.5f / (float)n_pix);
    return check_launch("sum_scale");
}

int sample_refeed_fwd(const float* mu, const float* var, const float* noise, float* x, long ldx, int B, int fps, int std_mode,
                      int layout, hipStream_t stream) {
    if (B <= 0) return FOV_OK;
    hipLaunchKernelGGL(sample_refeed_fwd_kernel, dim3(B), dim3(64), 0, stream, mu, var, noise, x, ldx, fps, std_mode, layout);
    return check_launch("sample_refeed_fwd");
}

int sample_refeed_bwd(const float* dx, long ldx, const float* var, const float* noise, float* dmu, float* dvar, int B, int fps,
                      int std_mode, int layout, int accumulate, hipStream_t stream) {
    if (B <= 0) return FOV_OK;
    hipLaunchKernelGGL(sample_refeed_bwd_kernel, dim3(B), dim3(64), 0, stream, dx, ldx, var, noise, dmu, dvar, fps, std_mode, layout,
                       accumulate);
    return check_launch("sample_refeed_bwd");
}

int rmsprop_tf_step(float* p, const float* g, float* ms, long n, float lr, float decay, float eps, float clip,
                    const unsigned* const* guards, long long* applied, hipStream_t stream) {
    if (n <= 0) return FOV_OK;
    hipLaunchKernelGGL(rmsprop_tf_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, p, g, ms, n, lr, decay, eps, clip,
                       guards ? guards[0] : nullptr, guards ? guards[1] : nullptr, guards ? guards[2] : nullptr, applied);
    return check_launch("rmsprop_tf");
}

}  // namespace fov
