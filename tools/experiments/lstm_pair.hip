// Fused encoder + autoregressive decoder for H = 256 with TWO 16-sequence tiles per workgroup (gfx950).
//
// Replaces the keras LSTM / Dense calls of mycode/FoV_seq2seq.py:83-97 and the host-driven decode loop :154-178
// (include/fov360.h, fov_seq2seq_decode_fwd) at the headline shape - the same arithmetic as lstm_cluster.hip, laid
// out for what tools/microbench/mfma_f32_overlap.hip measured on this chip: v_mfma_f32_16x16x4_f32 runs on the
// vector ALU's own datapath.  A VALU instruction next to it - of the same wave or of another wave on the SIMD - is
// never hidden (32 -> 46 cycles per MFMA gap with ONE v_fma_f32 in it, an MFMA-only wave beside a VALU-only wave
// takes the sum of both), so the only time a second wave can win back is the time a wave spends WAITING: exchange
// round trips, barriers, LDS and load latencies - 18-22 % of lstm_cluster's step.
//
// Layout:
//   * a workgroup is 512 threads = two SETS of four waves; each set owns its own 16-sequence tile and runs
//     free of the other (no s_barrier inside the time loops: the four waves of a set meet through flag words in
//     LDS), so one set's waits are covered by the other set's MFMAs on the same SIMDs;
//   * a GROUP of 8 workgroups owns a pair of tiles; workgroup `slice` owns hidden units [32*slice, +32), wave w of
//     a set 8 of them for ALL FOUR gates: 32 gate columns = two MFMA M-tiles in the TRANSPOSED product
//     z^T = W^T . a^T (weights are the A operand, activations the B operand).  M row m of tile tau is
//     (unit 4*tau + m/4, gate m%4), so register r of the D fragment on lane (n, g4) is gate r of unit 4*tau + g4
//     for sequence n: all four gates of a cell in one lane, no cross-lane traffic, c never leaves registers;
//   * R slice: 256 k-rows x 32 columns = 128 AGPRs per lane for the whole phase (MFMA A operands);
//     K slice (encoder): in LDS in A-operand order, shared by the two sets (lane-linear ds_read_b128);
//   * h tile (16 x 256) per set in LDS, own slice rotated to columns 0..31; exchanged per step as 8-byte
//     {value, epoch} granules (protocol of lstm_cluster.hip / xch_common.h: sc1 stores, or sc0 when a hello
//     handshake shows the whole group on one XCD; bounded spins; sticky timeout);
//   * decoder: Dense(F_dec, tanh) on the matrix pipe, transposed, K split over the four waves of the set; its
//     partial sums meet in LDS UNDER the partner-slice MFMAs of the next step (which do not need y_t).
// One pair of tiles per group only (B <= 16 * 2 * CUs/8): larger batches take lstm_cluster.hip's persistent loop.
#include <stdlib.h>

#include "fov_common.h"
#include "xch_common.h"

namespace fov {

namespace {

constexpr int PH = 256;            // hidden units
constexpr int PG = 8;              // workgroups per group
constexpr int PBT = 16;            // sequences per tile
constexpr int PLDH = PH + 4;       // LDS row stride of an h tile
constexpr int PNG = 14;            // granules gathered per thread and step: 7 slices * 16 rows * 32 units / 256
constexpr int P_MAX_F = 96;
constexpr int P_MAX_O = 8;
constexpr int PXR = 6;             // x prefetch registers per thread: 16 rows * F <= 256 * PXR
constexpr int PKPAD = 2;           // spare K blocks per wave slice for the run-ahead reads
constexpr unsigned P_SPIN_LIMIT = 1u << 20;
constexpr unsigned P_OORB = 0x80000000u;   // buffer-load offset no descriptor covers: reads as 0

typedef unsigned pu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned p_xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(x));
    return x;
}
__device__ __forceinline__ unsigned long long p_ld_granule(const unsigned long long* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void p_st_granule(unsigned long long* p, unsigned long long v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// weights are the A operand (src0): AGPR-resident (R) or from LDS / registers (K, Dense)
__device__ __forceinline__ void mfma_aw(f32x4& acc, float w_agpr, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(w_agpr), "v"(b));
}
__device__ __forceinline__ void mfma_vw(f32x4& acc, float w_vgpr, float b) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(w_vgpr), "v"(b));
}
// hipcc pads no hazards around inline asm (cdna_hip_programming.md 5.7): VALU write -> MFMA read, MFMA write -> VALU read
__device__ __forceinline__ void p_begin(f32x4 (&acc)[2]) { asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void p_end(f32x4 (&acc)[2]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}
__device__ __forceinline__ void p_guard2(float& a, float& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }

__device__ __forceinline__ unsigned lds_addr(const void* p) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const void*)p;
}
// Flag words of a set: wave w stores its progress (the epoch of the step it has completed the phase of) into word w;
// a waiter reads the four words with one ds_read_b128.  LDS instructions of a wave execute in order and the LDS serves
// one instruction at a time, so data written before the flag store is visible to whoever has seen the flag.
__device__ __forceinline__ void set_signal(unsigned flag_addr, unsigned val, int lane) {
    if (lane == 0) asm volatile("ds_write_b32 %0, %1" ::"v"(flag_addr), "v"(val) : "memory");
    else asm volatile("" ::: "memory");
}
// true = give up (abort flag of the workgroup set, or the wait ran into its bound)
__device__ __forceinline__ bool set_wait(unsigned flags_addr, unsigned val, volatile int* abortw) {
    unsigned spins = 0;
    while (true) {
        pu32x4 f;
        asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(f) : "v"(flags_addr) : "memory");
        const int d0 = (int)(f.x - val), d1 = (int)(f.y - val), d2 = (int)(f.z - val), d3 = (int)(f.w - val);
        const int m = min(min(d0, d1), min(d2, d3));
        if (__builtin_amdgcn_readfirstlane(m) >= 0) return false;
        ++spins;
        if ((spins & 31u) == 0 && *abortw != 0) return true;
        if (spins > (P_SPIN_LIMIT << 2)) { *abortw = 1; return true; }
        __builtin_amdgcn_s_sleep(1);
    }
}

// Diagnostic build only (-DFOV_STAMPS, tools/stamp_pair.py): s_memtime stamps of wave 0 of both sets of one workgroup
#ifdef FOV_STAMPS
constexpr int PSTAMP_SLOTS = 12;
constexpr int PSTAMP_STEPS = 64;
__device__ unsigned long long g_pair_stamps[2][2][PSTAMP_STEPS][PSTAMP_SLOTS];   // [set][phase][step][slot]
#define PAIR_STAMP(phase, slot)                                                                \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (stamp_on && t < PSTAMP_STEPS) {                                                    \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            g_pair_stamps[set][phase][t][slot] = t_;                                           \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define PAIR_STAMP(phase, slot) do { } while (0)
#endif

struct PairLds {
    int ldx;
    int off_k, off_h, off_x, off_w, off_be, off_bd, off_wd, off_sync, total_floats;
};
__host__ __device__ inline PairLds pair_lds(int F) {
    PairLds L;
    const int fp = (F + 15) & ~15;
    const int nq = fp >> 4;
    L.ldx = fp + 4;
    L.off_k = 0;                                              // [4 wave slices][2*nq blocks][64 lanes][4] + PKPAD spare blocks
    L.off_h = L.off_k + (4 * 2 * nq + PKPAD) * 256;           // [2 sets][16][PLDH]
    L.off_x = L.off_h + 2 * PBT * PLDH;                       // [2 sets][2 buffers][16][ldx] (+ tail pad)
    L.off_w = L.off_x + 2 * 2 * PBT * L.ldx + 64;             // [2 sets][4 waves][16][16] Dense partials
    L.off_be = L.off_w + 2 * 4 * 256;                         // [4 waves][64 lanes][8] encoder bias in accumulator layout
    L.off_bd = L.off_be + 4 * 64 * 8;                         // the same for the decoder LSTM
    L.off_wd = L.off_bd + 4 * 64 * 8;                         // [4 waves][4 blocks][64 lanes][4] Dense kernel, A-operand order
    L.off_sync = L.off_wd + 4 * 4 * 64 * 4;                   // [2 sets][3 flags][4 words], then abort / hello / ticket words
    L.total_floats = L.off_sync + 2 * 3 * 4 + 16;
    return L;
}

// acc[tau] += W[rows of blocks J0..J1) . h^T : h tile rows in LDS (B operand), W in AGPRs [16 k-blocks][4][2 tiles]
template <int J0, int J1>
__device__ __forceinline__ void p_recur(f32x4 (&acc)[2], const float* hrow, const float (&w)[16][4][2]) {
    if (J0 >= J1) return;
    f32x4 a = *(const f32x4*)(hrow + 16 * J0);
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        f32x4 an = a;
        if (j + 1 < J1) an = *(const f32x4*)(hrow + 16 * (j + 1));
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            mfma_aw(acc[0], w[j][s][0], a[s]);
            mfma_aw(acc[1], w[j][s][1], a[s]);
        }
        a = an;
    }
}

// acc[tau] += K^T . x^T: x tile rows in LDS (B operand), K slice in LDS in A-operand order: block 2q + hh of a wave
// slice holds, per lane, {tau0, tau1} of k-steps s = 2hh, 2hh + 1 of k-block q.  Reads run one q ahead (spare blocks /
// tail pad keep them inside LDS; their values are never used).
__device__ __forceinline__ void p_input_proj(f32x4 (&acc)[2], const float* xrow, const float* sKl, int nq) {
    if (nq <= 0) return;
    f32x4 b = *(const f32x4*)xrow;
    f32x4 k0 = *(const f32x4*)sKl, k1 = *(const f32x4*)(sKl + 256);
    for (int q = 0; q < nq; ++q) {
        const f32x4 bn = *(const f32x4*)(xrow + 16 * (q + 1));
        const f32x4 n0 = *(const f32x4*)(sKl + (2 * q + 2) * 256), n1 = *(const f32x4*)(sKl + (2 * q + 3) * 256);
        mfma_vw(acc[0], k0[0], b[0]);
        mfma_vw(acc[1], k0[1], b[0]);
        mfma_vw(acc[0], k0[2], b[1]);
        mfma_vw(acc[1], k0[3], b[1]);
        mfma_vw(acc[0], k1[0], b[2]);
        mfma_vw(acc[1], k1[1], b[2]);
        mfma_vw(acc[0], k1[2], b[3]);
        mfma_vw(acc[1], k1[3], b[3]);
        b = bn; k0 = n0; k1 = n1;
    }
}

// R slice of this wave -> registers.  Block j covers the hidden units ((slice + j/2) mod 8)*32 + (j%2)*16 .. +16: the
// workgroup's OWN 32 units are blocks 0, 1.  Lane (m = lane%16, g4): column (gate m%4, unit ubase + m/4 [+4 for tau 1]),
// k row of MFMA step s = block base + 4*g4 + s.  Sixteen loads at a time: with all 128 in flight the values pass
// through 128 VGPRs the kernel does not have (two waves per SIMD: 128 + 128), and what hipcc spills then is the
// weights - for the whole launch.
__device__ __forceinline__ void p_load_r(float (&w)[16][4][2], const float* R, int slice, int ucol, int lane) {
    const int m = lane & 15, g4 = lane >> 4;
    constexpr int H4 = 4 * PH;
    const __amdgpu_buffer_rsrc_t rrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(R), 0, PH * H4 * 4, 0x00020000);
    const unsigned voff = (unsigned)((4 * g4 * H4 + (m & 3) * PH + ucol + (m >> 2)) * 4);
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const unsigned kb = (unsigned)((((slice + (j >> 1)) & (PG - 1)) * 32 + (j & 1) * 16) * H4 * 4);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int tau = 0; tau < 2; ++tau)
                w[j][s][tau] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rrs, voff, kb + (unsigned)((s * H4 + 4 * tau) * 4), 0));
        if (j & 1) __builtin_amdgcn_sched_barrier(0);
    }
}

// bias of this wave's 32 gate columns in accumulator layout ([tau][gate r] of unit ucol + 4*tau + g4) -> LDS
__device__ __forceinline__ void p_stage_bias(float* dst, const float* b, int ucol, int lane) {
    const int g4 = lane >> 4;
    const __amdgpu_buffer_rsrc_t brs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(b), 0, 4 * PH * 4, 0x00020000);
    f32x4 v[2];
#pragma unroll
    for (int tau = 0; tau < 2; ++tau)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            v[tau][r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(brs, (unsigned)((r * PH + ucol + 4 * tau + g4) * 4), 0, 0));
    *(f32x4*)(dst + lane * 8) = v[0];
    *(f32x4*)(dst + lane * 8 + 4) = v[1];
}

}  // namespace

template <int ACT>
__global__ __launch_bounds__(512) void lstm_pair_fused_kernel(LstmParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, ts = tid & 255;
    // wave-uniform by construction; readfirstlane tells the compiler (a divergent `set` made the granule buffer descriptor
    // divergent and every granule load / store a waterfall loop)
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int set = wave8 >> 2, wave = wave8 & 3;
    const int n = lane & 15, g4 = lane >> 4;
    constexpr int H4 = 4 * PH;

    // members of a group 8 blocks apart: round-robin dispatch puts them on one XCD when num_groups % 8 == 0.  A speed
    // choice only - whether they really share an XCD is verified below (hello handshake) before sc0 stores are used.
    int group, slice;
    if ((p.num_groups & 7) == 0) {
        group = (blockIdx.x / (8 * PG)) * 8 + (blockIdx.x & 7);
        slice = (blockIdx.x >> 3) & (PG - 1);
    } else {
        group = blockIdx.x / PG;
        slice = blockIdx.x - group * PG;
    }
    const int tile = 2 * group + set;
    const bool live = tile < p.num_tiles;
    const int b0 = tile * PBT;
    const int ucol = 32 * slice + 8 * wave;          // first hidden unit of this wave
    const int F = p.F, O = p.F_dec;
    const int Fp = (F + 15) & ~15, nq = Fp >> 4;
    const int T = p.T, T_out = p.T_out;
#ifdef FOV_STAMPS
    const bool stamp_on = (blockIdx.x == 13 && wave == 0 && lane == 0);
#endif

    const PairLds L = pair_lds(F);
    const int LDX = L.ldx;
    float* sKw = smem + L.off_k + wave * (2 * nq) * 256;           // this wave's K slice (shared with wave + 4 of the other set)
    float* sH = smem + L.off_h + set * PBT * PLDH;
    float* sX = smem + L.off_x + set * 2 * PBT * LDX;
    float* sW = smem + L.off_w + set * 4 * 256;
    const float* sBe = smem + L.off_be + wave * 512 + lane * 8;
    const float* sBd = smem + L.off_bd + wave * 512 + lane * 8;
    const float* sWd = smem + L.off_wd + wave * 1024 + lane * 4;
    unsigned* sSync = (unsigned*)(smem + L.off_sync) + set * 12;    // [3][4]
    int* sFlag = (int*)(smem + L.off_sync) + 24;                    // [0] abort, [1] "a partner lives on another XCD"
    unsigned* sXch = (unsigned*)(sFlag + 4);                        // base / launch index (xch_common.h)
    volatile int* abortw = sFlag;
    const unsigned fa_addr = lds_addr(sSync), fb_addr = lds_addr(sSync + 4), fd_addr = lds_addr(sSync + 8);

    // ---- launch protocol (xch_common.h): epoch base, arrival ticket, poison check, hello handshake ----
    const unsigned arrival = xch_arrive(p.status, sXch);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) { sFlag[0] = poisoned ? 1 : 0; sFlag[1] = 0; }
    if (!poisoned && tid < 64) {   // wave 0: thread 0 has just written sXch (same wave: program order)
        const unsigned epoch_base = sXch[0];
        unsigned long long* hello = p.xch + (size_t)p.num_groups * 2 * 2 * PBT * PH + (size_t)group * PG;
        const unsigned mine = p_xcc_id();
        const unsigned long long hello_tag = (unsigned long long)epoch_base + 1ull;
        if (tid == 0) p_st_granule(hello + slice, (hello_tag << 32) | mine);
        if (tid < PG) {
            unsigned long long hv = 0;
            unsigned spins = 0;
            while (true) {
                hv = p_ld_granule(hello + tid);
                if ((hv >> 32) == hello_tag) break;
                if (++spins > P_SPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                    xch_give_up(p.status);
                    sFlag[0] = 1;
                    break;
                }
                __builtin_amdgcn_s_sleep(4);
            }
            if ((unsigned)hv != mine) sFlag[1] = 1;
        }
    }
    // ---- staging that needs no AGPRs (the R slice comes last, see p_load_r) ----
    // K slice -> LDS (A-operand order); set s stages the k-blocks q = s, s + 2, ... of its wave's slice
    {
        const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.K), 0, F * H4 * 4, 0x00020000);
        const unsigned cvo = (unsigned)(((n & 3) * PH + ucol + (n >> 2)) * 4);
        constexpr int QH = P_MAX_F / 32;   // k-blocks per set at most
        f32x4 kv[QH][2];
#pragma unroll
        for (int qi = 0; qi < QH; ++qi) {
            const int q = 2 * qi + set;
#pragma unroll
            for (int hh = 0; hh < 2; ++hh)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int s = 2 * hh + (e >> 1), tau = e & 1;
                    const int k = 16 * q + 4 * g4 + s;   // k >= F: past the descriptor, reads as 0
                    kv[qi][hh][e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(krs, cvo + (unsigned)(k * H4 * 4), (unsigned)(4 * tau * 4), 0));
                }
        }
#pragma unroll
        for (int qi = 0; qi < QH; ++qi) {
            const int q = 2 * qi + set;
            if (q < nq) {
                *(f32x4*)(sKw + ((2 * q) * 64 + lane) * 4) = kv[qi][0];
                *(f32x4*)(sKw + ((2 * q + 1) * 64 + lane) * 4) = kv[qi][1];
            }
        }
    }
    __builtin_amdgcn_sched_barrier(0);
    if (set == 0) {
        p_stage_bias(smem + L.off_be + wave * 512, p.b, ucol, lane);
        p_stage_bias(smem + L.off_bd + wave * 512, p.db, ucol, lane);
    } else {
        // Dense, transposed: y^T = Wd^T . h^T.  A operand: row i = lane%16 is output o(i) = 4*(i&3) + (i>>2), k = rotated tile
        // position 64*wave + 16*b + 4*g4 + ss.  Register r of the D fragment on lane (n, g4) is then y[n][4r + g4]: registers
        // 0, 1 are the B operands of the two y . K MFMA steps.
        const __amdgpu_buffer_rsrc_t wdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dW), 0, PH * O * 4, 0x00020000);
        const int o = 4 * (n & 3) + (n >> 2);
        f32x4 wv[4];
#pragma unroll
        for (int b = 0; b < 4; ++b)
#pragma unroll
            for (int ss = 0; ss < 4; ++ss) {
                const int pos = 64 * wave + 16 * b + 4 * g4 + ss;
                const int unit = ((slice + (pos >> 5)) & (PG - 1)) * 32 + (pos & 31);
                wv[b][ss] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wdrs, (o < O) ? (unsigned)((unit * O + o) * 4) : P_OORB, 0, 0));
            }
#pragma unroll
        for (int b = 0; b < 4; ++b) *(f32x4*)(smem + L.off_wd + wave * 1024 + (b * 64 + lane) * 4) = wv[b];
    }
    __builtin_amdgcn_sched_barrier(0);
    // zero the x tiles once (pad columns [F, Fp) and the tail are never written afterwards)
    for (int i = ts; i < 2 * PBT * LDX + 32; i += 256) sX[i] = 0.f;

    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    const bool same_xcd = (sFlag[1] == 0) && (p.force_safe_exchange == 0);
    bool aborted = sFlag[0] != 0;
    if (tid == 0 && !same_xcd && !aborted) xch_count_safe(p.status, ticket);
    if (lane == 0) { sSync[wave] = epoch; sSync[4 + wave] = epoch; sSync[8 + wave] = epoch; }

    // ---- exchange bookkeeping: one descriptor per (group, set), 32-bit lane offsets computed once ----
    constexpr unsigned PARITY_BYTES = PBT * PH * 8u;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + ((size_t)group * 2 + set) * 2 * PBT * PH, 0, 2 * PBT * PH * (int)sizeof(unsigned long long), 0x00020000);
    const unsigned pub_off = (unsigned)(n * PH + ucol + g4) * 8u;          // + 4*8 for tau = 1
    // gather: granule j of this thread = (partner slice (slice + 1 + j/2) mod 8, row 8*(j&1) + ts/32, unit ts%32)
    const unsigned gvoff = (unsigned)((ts >> 5) * PH + (ts & 31)) * 8u;
    const int gl_off = (ts >> 5) * PLDH + (ts & 31);
    const int hrow_off = n * PLDH + 4 * g4;                                  // this lane's B-operand row of the h tile
    const int own_off = n * PLDH + 8 * wave + g4;                            // + 4 for tau = 1

    auto gather_issue = [&](pu32x2 (&v)[PNG], unsigned par) {
#pragma unroll
        for (int j = 0; j < PNG; ++j) {
            const unsigned uo = (unsigned)((j & 1) * 8 * PH + ((slice + 1 + (j >> 1)) & (PG - 1)) * 32) * 8u;
            v[j] = __builtin_amdgcn_raw_buffer_load_b64(xrs, gvoff, par + uo, 16);
        }
    };
    auto gather_store = [&](const pu32x2 (&v)[PNG]) {
        float* gl = sH + gl_off;
#pragma unroll
        for (int j = 0; j < PNG; ++j) gl[(j & 1) * 8 * PLDH + ((j >> 1) + 1) * 32] = __uint_as_float(v[j].x);
    };
    auto gather_ok = [&](const pu32x2 (&v)[PNG]) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < PNG; ++j) ok = ok && (v[j].y == epoch);
        return __all(ok) != 0;
    };
    // the first sweep was requested earlier (v); a sweep that came back incomplete is repeated into temporaries of the retry
    // loop - an array carried around that loop costs about nine registers per granule (tools/experiments/README.md)
    auto gather_finish = [&](pu32x2 (&v)[PNG], unsigned par) {
        if (gather_ok(v)) {
            gather_store(v);
            return;
        }
        unsigned spins = 0;
#pragma clang loop unroll(disable)
        while (true) {
            ++spins;
            if (spins > P_SPIN_LIMIT || *abortw != 0 || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                if (lane == 0) {
                    xch_give_up(p.status);
                    *abortw = 1;
                }
                aborted = true;
                return;
            }
            __builtin_amdgcn_s_sleep(1);
            pu32x2 w[PNG];
            gather_issue(w, par);
            if (gather_ok(w)) {
                gather_store(w);
                return;
            }
        }
    };
    auto publish = [&](const float (&h)[2], unsigned par) {
        if (same_xcd) {
#pragma unroll
            for (int tau = 0; tau < 2; ++tau)
                __builtin_amdgcn_raw_buffer_store_b64((pu32x2){__float_as_uint(h[tau]), epoch}, xrs, pub_off + tau * 32, par, 1 /* sc0 */);
        } else {
#pragma unroll
            for (int tau = 0; tau < 2; ++tau)
                __builtin_amdgcn_raw_buffer_store_b64((pu32x2){__float_as_uint(h[tau]), epoch}, xrs, pub_off + tau * 32, par, 16 /* sc1 */);
        }
    };
    auto cell = [&](const f32x4 (&acc)[2], float (&c)[2], float (&h)[2]) {
#pragma unroll
        for (int tau = 0; tau < 2; ++tau) {
            const float ig = rec_act<ACT>(acc[tau][0]);
            const float fg = rec_act<ACT>(acc[tau][1]);
            const float gg = tanh_f(acc[tau][2]);
            const float og = rec_act<ACT>(acc[tau][3]);
            c[tau] = fmaf(fg, c[tau], ig * gg);
            h[tau] = og * tanh_f(c[tau]);
        }
    };

    // ---- initial state and the first two x tiles ----
    const int live_rows = !live ? 0 : (p.B - b0 < PBT ? p.B - b0 : PBT);
    float c[2], hcur[2];
    {
        const __amdgpu_buffer_rsrc_t c0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.c0 && live ? p.c0 + (size_t)b0 * PH : nullptr), 0, p.c0 ? live_rows * PH * 4 : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t h0rs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(p.h0 && live ? p.h0 + (size_t)b0 * PH : nullptr), 0, p.h0 ? live_rows * PH * 4 : 0, 0x00020000);
#pragma unroll
        for (int tau = 0; tau < 2; ++tau) {
            const unsigned off = (unsigned)((n * PH + ucol + 4 * tau + g4) * 4);
            c[tau] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(c0rs, off, 0, 0));
            hcur[tau] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, off, 0, 0));
        }
        const int unit = ((slice + (ts >> 5)) & (PG - 1)) * 32 + (ts & 31);   // rotated position ts of every row
#pragma unroll
        for (int q0 = 0; q0 < PBT; q0 += 4) {
            float hv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                hv[q] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(h0rs, (unsigned)(((q0 + q) * PH + unit) * 4), 0, 0));
#pragma unroll
            for (int q = 0; q < 4; ++q) sH[(q0 + q) * PLDH + ts] = hv[q];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // x staging: thread (xrw = ts/16, xcl = ts%16) moves columns xcl + 16*i of row xrw
    const int xrw = ts >> 4, xcl = ts & 15;
    const __amdgpu_buffer_rsrc_t xgrs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(live ? p.x + (size_t)b0 * T * F : nullptr), 0, live_rows * T * F * 4, 0x00020000);
    unsigned xbase = (unsigned)((xrw * T * F + xcl) * 4);
    const int xl_off = xrw * LDX + xcl;
    float xr[PXR];
    {
        float x2[2][PXR];
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < PXR; ++i)
                x2[tt][i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, (tt < T && xcl + 16 * i < F) ? xbase + 64u * i : P_OORB, (unsigned)(tt * F * 4), 0));
#pragma unroll
        for (int i = 0; i < PXR; ++i)
            xr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, (2 < T && xcl + 16 * i < F) ? xbase + 64u * i : P_OORB, (unsigned)(2 * F * 4), 0));
#pragma unroll
        for (int tt = 0; tt < 2; ++tt)
#pragma unroll
            for (int i = 0; i < PXR; ++i)
                if (xcl + 16 * i < F) sX[xl_off + tt * PBT * LDX + 16 * i] = x2[tt][i];
    }
    __builtin_amdgcn_sched_barrier(0);

    // ---- encoder R slice -> AGPRs ----
    float wR[16][4][2];
    p_load_r(wR, p.R, slice, ucol, lane);
    __syncthreads();   // K slice, biases, Dense kernel, h_0 tile, x_0, x_1, flag words: visible to everybody

    f32x4 acc[2];
    const float* sKl = sKw + lane * 4;
    const float* hrow = sH + hrow_off;
    if (live && !aborted) {
        // ---- z_0 of the encoder ----
        acc[0] = *(const f32x4*)sBe;
        acc[1] = *(const f32x4*)(sBe + 4);
        p_begin(acc);
        p_input_proj(acc, sX + n * LDX + 4 * g4, sKl, nq);
        p_recur<0, 16>(acc, hrow, wR);
    }
    __syncthreads();   // every wave has read the h_0 tile before the first own-slice write of h_t into it

    if (live && !aborted) {
        // =============================== encoder phase ===============================
#pragma clang loop unroll(disable)
        for (int t = 0; t < T; ++t) {
            PAIR_STAMP(0, 0);
#ifdef FOV_STAMPS
            if (stamp_on && t < PSTAMP_STEPS) g_pair_stamps[set][0][t][11] = __builtin_amdgcn_s_memrealtime();
#endif
            p_end(acc);
            cell(acc, c, hcur);
            ++epoch;
            const unsigned par = (epoch & 1u) * PARITY_BYTES;
            publish(hcur, par);
            sH[own_off] = hcur[0];
            sH[own_off + 4] = hcur[1];
            set_signal(fa_addr + 4 * wave, epoch, lane);
            const bool more = (t + 1 < T);
            PAIR_STAMP(0, 1);
            if (more) {
                acc[0] = *(const f32x4*)sBe;
                acc[1] = *(const f32x4*)(sBe + 4);
                p_begin(acc);
                p_input_proj(acc, sX + ((t + 1) & 1) * PBT * LDX + n * LDX + 4 * g4, sKl, nq);
            }
            PAIR_STAMP(0, 2);
            pu32x2 v[PNG];
            gather_issue(v, par);
            if (set_wait(fa_addr, epoch, abortw)) { aborted = true; break; }
            PAIR_STAMP(0, 3);
            if (more) p_recur<0, 2>(acc, hrow, wR);
            PAIR_STAMP(0, 4);
            gather_finish(v, par);
            PAIR_STAMP(0, 5);
            if (aborted) break;
            // x_{t+2} (requested a step ago) -> LDS buffer t&1, last read by x_t . K before this step's flag A; then x_{t+3}
            asm volatile("" : "+v"(xbase));   // offsets and column masks are recomputed per step: hoisted they cost 12 registers
            if (t + 2 < T) {
                float* xb = sX + xl_off + (t & 1) * PBT * LDX;
#pragma unroll
                for (int i = 0; i < PXR; ++i)
                    if (xcl + 16 * i < F) xb[16 * i] = xr[i];
            }
            set_signal(fb_addr + 4 * wave, epoch, lane);
#pragma unroll
            for (int i = 0; i < PXR; ++i)
                xr[i] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(xgrs, (t + 3 < T && xcl + 16 * i < F) ? xbase + 64u * i : P_OORB, (unsigned)((t + 3) * F * 4), 0));
            if (set_wait(fb_addr, epoch, abortw)) { aborted = true; break; }
            PAIR_STAMP(0, 6);
            if (more) p_recur<2, 16>(acc, hrow, wR);
            PAIR_STAMP(0, 7);
        }
    }

    // =============================== decoder phase ===============================
    // weights of the decoder LSTM: R -> the same 128 AGPRs, K (F_dec <= 8 rows: k = 4*s + g4) in VGPRs
    if (live && !aborted) {
        p_load_r(wR, p.dR, slice, ucol, lane);
        float kd[2][2];
        float bd2[2];
        {
            const __amdgpu_buffer_rsrc_t krs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dK), 0, O * H4 * 4, 0x00020000);
            const __amdgpu_buffer_rsrc_t bdrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.dbias), 0, O * 4, 0x00020000);
            const unsigned cvo = (unsigned)(((n & 3) * PH + ucol + (n >> 2)) * 4);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int tau = 0; tau < 2; ++tau)   // k = 4*s + g4 >= O: past the descriptor, reads as 0
                    kd[s][tau] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(krs, cvo + (unsigned)((4 * s + g4) * H4 * 4), (unsigned)(4 * tau * 4), 0));
#pragma unroll
            for (int s = 0; s < 2; ++s)
                bd2[s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(bdrs, (4 * s + g4 < O) ? (unsigned)((4 * s + g4) * 4) : P_OORB, 0, 0));
        }
        float y2[2];   // y_{t-1}[n][4*s + g4]: the B fragment of K^T . y^T
#pragma unroll
        for (int s = 0; s < 2; ++s)
            y2[s] = (4 * s + g4 < O && b0 + n < p.B) ? p.dec_in0[(size_t)(b0 + n) * O + 4 * s + g4] : 0.f;
        // z_0 of the decoder from the complete h_T tile the encoder phase left in LDS
        acc[0] = *(const f32x4*)sBd;
        acc[1] = *(const f32x4*)(sBd + 4);
        p_guard2(y2[0], y2[1]);
        p_begin(acc);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            mfma_vw(acc[0], kd[s][0], y2[s]);
            mfma_vw(acc[1], kd[s][1], y2[s]);
        }
        p_recur<0, 16>(acc, hrow, wR);
        // every wave of the set has read the h_T tile before the first own-slice write of the decoder.  Flag D is unused so
        // far (still at the base) and the encoder's last epoch is larger; the epoch itself must NOT advance here: granule
        // tags and parity buffers alternate strictly with the exchanges.
        set_signal(fd_addr + 4 * wave, epoch, lane);
        if (set_wait(fd_addr, epoch, abortw)) aborted = true;
        const float* hq = hrow + 64 * wave;
        float* yo = p.out + ((size_t)(b0 + n) * T_out) * O + g4;
#pragma clang loop unroll(disable)
        for (int t = 0; t < T_out && !aborted; ++t) {
            PAIR_STAMP(1, 0);
#ifdef FOV_STAMPS
            if (stamp_on && t < PSTAMP_STEPS) g_pair_stamps[set][1][t][11] = __builtin_amdgcn_s_memrealtime();
#endif
            p_end(acc);
            cell(acc, c, hcur);
            ++epoch;
            const unsigned par = (epoch & 1u) * PARITY_BYTES;
            publish(hcur, par);
            sH[own_off] = hcur[0];
            sH[own_off + 4] = hcur[1];
            set_signal(fa_addr + 4 * wave, epoch, lane);
            const bool more = (t + 1 < T_out);
            PAIR_STAMP(1, 1);
            if (set_wait(fa_addr, epoch, abortw)) { aborted = true; break; }
            PAIR_STAMP(1, 2);
            if (more) {
                acc[0] = *(const f32x4*)sBd;
                acc[1] = *(const f32x4*)(sBd + 4);
                p_begin(acc);
                p_recur<0, 2>(acc, hrow, wR);
            }
            PAIR_STAMP(1, 3);
            {
                pu32x2 v[PNG];
                gather_issue(v, par);
                gather_finish(v, par);
            }
            PAIR_STAMP(1, 4);
            if (aborted) break;
            set_signal(fb_addr + 4 * wave, epoch, lane);
            if (set_wait(fb_addr, epoch, abortw)) { aborted = true; break; }
            PAIR_STAMP(1, 5);
            // Dense partial of this wave's 64 tile positions
            {
                f32x4 dacc[2];
                dacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                f32x4 hb = *(const f32x4*)hq, wb = *(const f32x4*)sWd;
                p_begin(dacc);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    f32x4 hn = hb, wn = wb;
                    if (b + 1 < 4) {
                        hn = *(const f32x4*)(hq + 16 * (b + 1));
                        wn = *(const f32x4*)(sWd + 256 * (b + 1));
                    }
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) mfma_vw(dacc[ss & 1], wb[ss], hb[ss]);
                    hb = hn; wb = wn;
                }
                p_end(dacc);
#pragma unroll
                for (int ss = 0; ss < 4; ++ss) dacc[0][ss] += dacc[1][ss];
                *(f32x4*)(sW + (wave * 16 + n) * 16 + 4 * g4) = dacc[0];
            }
            set_signal(fd_addr + 4 * wave, epoch, lane);
            PAIR_STAMP(1, 6);
            // the partner-slice part of z_{t+1} does not need y_t: it covers the meeting of the four partials
            if (more) {
                p_begin(acc);
                p_recur<2, 16>(acc, hrow, wR);
            }
            PAIR_STAMP(1, 7);
            if (set_wait(fd_addr, epoch, abortw)) { aborted = true; break; }
            PAIR_STAMP(1, 8);
            f32x4 ysum = *(const f32x4*)(sW + n * 16 + 4 * g4);
#pragma unroll
            for (int w2 = 1; w2 < 4; ++w2) {
                const f32x4 part = *(const f32x4*)(sW + (w2 * 16 + n) * 16 + 4 * g4);
#pragma unroll
                for (int ss = 0; ss < 2; ++ss) ysum[ss] += part[ss];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) y2[s] = tanh_f(ysum[s] + bd2[s]);
            if (slice == 0 && wave == 0 && b0 + n < p.B) {
#pragma unroll
                for (int s = 0; s < 2; ++s)
                    if (4 * s + g4 < O) yo[(size_t)t * O + 4 * s] = y2[s];
            }
            if (more) {
                p_guard2(y2[0], y2[1]);
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    mfma_vw(acc[0], kd[s][0], y2[s]);
                    mfma_vw(acc[1], kd[s][1], y2[s]);
                }
            }
            PAIR_STAMP(1, 9);
        }
        if (!aborted && b0 + n < p.B) {
#pragma unroll
            for (int tau = 0; tau < 2; ++tau) {
                if (p.hT) p.hT[(size_t)(b0 + n) * PH + ucol + 4 * tau + g4] = hcur[tau];
                if (p.cT) p.cT[(size_t)(b0 + n) * PH + ucol + 4 * tau + g4] = c[tau];
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
bool pair_shape_ok(int B, int T, int T_out, int F, int F_dec, int H) {
    if (H != PH || F < 1 || F > P_MAX_F || F_dec < 1 || F_dec > P_MAX_O || T < 1 || T_out < 1 || B < 1) return false;
    const int tiles = (B + PBT - 1) / PBT;
    const int groups = (tiles + 1) / 2;
    return groups <= device_cu_count() / PG;
}

int launch_pair_fused(const LstmParams& p_in, hipStream_t stream) {
    LstmParams p = p_in;
    p.num_tiles = (p.B + PBT - 1) / PBT;
    p.num_groups = (p.num_tiles + 1) / 2;
    if ((size_t)p.num_groups * (2 * 2 * PBT * PH + PG) * sizeof(unsigned long long) > kXchBytes - kHelloBytes) {
        set_error("pair kernel: granule area exceeds the workspace's");
        return FOV_ERR_WORKSPACE;
    }
    p.epoch_span = p.T + p.T_out + 2;
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    void (*kern)(LstmParams) = p.act == FOV_ACT_HARD_SIGMOID ? lstm_pair_fused_kernel<FOV_ACT_HARD_SIGMOID>
                                                             : lstm_pair_fused_kernel<FOV_ACT_SIGMOID>;
    const PairLds L = pair_lds(p.F);
    const size_t lds = (size_t)L.total_floats * sizeof(float);
    if (lds > 160 * 1024) { set_error("pair kernel: needs %zu B of LDS (> 160 KiB)", lds); return FOV_ERR_UNSUPPORTED; }
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(p.num_groups * PG), dim3(512), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("pair kernel launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_pair_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_pair_stamps), sizeof(unsigned long long) * 2 * 2 * PSTAMP_STEPS * PSTAMP_SLOTS);
}
#endif

}  // namespace fov
