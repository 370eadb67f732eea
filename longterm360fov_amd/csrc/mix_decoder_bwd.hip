// Backward counterpart of mix_decoder.hip: BPTT through the whole unrolled others-mixing decoder (mixing head,
// layer 2, layer 1, feedback path) in ONE persistent launch - what Keras/TF autodiff runs under model.fit for
// mycode/given_others_gt_mean_var_seq2seq.py:203-308.  It produces the DATA gradients only: dz of both layers and
// the pre-activation gradients of the two head layers for every step (the weight gradients are then one product
// per layer over all steps, formed by the caller), and the gradient with respect to the decoder's initial state.
//
// Same ownership as the forward kernel: a tile of 16 sequences per group of 8 workgroups, workgroup `slice` owns
// hidden units [32*slice, +32) of both layers, lane (n, g4) the cells (rows 4*g4 + 2*(n>>3) + {0,1}, unit
// 32*slice + 8*wave + (n&7)): dc1, dc2 and the recurrent dh never leave registers.
// Per step t = T-1 .. 0:
//   head      dpre_m = dL/dz_m + dx_{t+1} (1 - m^2);  dpre_p = (dpre_m Wp^T)(1 - p^2);  dh2 += dpre_p Wd^T   (every
//             workgroup redundantly: O = 6)
//   layer 2   gates backward (lane-local, from the reserve) -> dz2 (own 128 gate columns) -> LDS
//             partial[16 x 512] = dz2_own . [R2^T | K2^T]_own  : the contribution of the own gate columns to dh2_{t-1}
//             (all 256 units) and to dh1_t (all 256 units); 256 MFMAs per wave.  R2^T slice in 128 AGPRs, K2^T slice
//             28/32 fragment blocks in LDS + 4 in VGPRs (the forward kernel's arrangement).
//             The 16 x 32 piece of each destination workgroup travels as {value, epoch} granules; every lane
//             gathers the 8 pieces of its own cells and adds them in slice order (deterministic).
//   layer 1   the same with R1^T (128 AGPRs) -> dh1_{t-1}; dx_t = dz1 . K1^T (16 x O, on the VALU: a 17th MFMA tile on one
//             wave made that wave the straggler of every barrier) is summed the same way and feeds the head of step
//             t-1 (x_t = m_{t-1}).
// MFMA-bound in arithmetic (392 per wave and step, as the forward), in practice bound by its two exposed
// exchanges per step.
#include <stdlib.h>

#include "fov_common.h"
#include "xch_common.h"

namespace fov {

namespace {

constexpr int BH = 256;
constexpr int BG = 8;
constexpr int BBT = 16;
constexpr int BLDZ = 128 + 4;    // LDS row stride of the dz tile (16 x 128 own gate columns)
constexpr unsigned B_SPIN = 1u << 20;
constexpr int BK2_LDS_BLOCKS = 28;

typedef unsigned bwu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned bwu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bm_a(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}
__device__ __forceinline__ void bm_v(f32x4& acc, float a, float w_vgpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w_vgpr));
}
template <int N>
__device__ __forceinline__ void bm_begin(f32x4 (&acc)[N]) {
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(acc[i]));
    asm volatile("s_nop 3");
}
template <int N>
__device__ __forceinline__ void bm_end(f32x4 (&acc)[N]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" : "+v"(acc[i]));
}
template <int ACT>
__device__ __forceinline__ float bact_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

}  // namespace

// Diagnostic build only (-DFOV_STAMPS, tools/stamp_mix_decoder.py --bwd): s_memtime stamps of one wave per step.
#ifdef FOV_STAMPS
constexpr int BSTAMP_SLOTS = 12;
constexpr int BSTAMP_STEPS = 32;
__device__ unsigned long long g_mixb_stamps[BSTAMP_STEPS][BSTAMP_SLOTS];
#define MIXB_STAMP(slot)                                                                       \
    do {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        if (stamp_on && (T - 1 - t) < BSTAMP_STEPS) {                                          \
            unsigned long long t_;                                                             \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");        \
            g_mixb_stamps[T - 1 - t][slot] = t_;                                               \
        }                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    } while (0)
#else
#define MIXB_STAMP(slot) do { } while (0)
#endif

// granule areas of one group, in granules (8 bytes each); two parities of each
constexpr size_t MB2 = (size_t)BG * BG * 2 * BBT * 32;   // [dest][src][product][row pair][unit][row of the pair]
constexpr size_t MB1 = (size_t)BG * BG * BBT * 32;       // [dest][src][row pair][unit][row of the pair]
constexpr size_t MBX = (size_t)BG * BBT * 8;             // [src][row][o]
constexpr size_t MB_GROUP = 2 * (MB2 + MB1 + MBX);

template <int ACT>
__global__ __launch_bounds__(256, 1) void mix_decoder_bwd_kernel(MixDecBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* sDZ = smem;                         // [16][BLDZ]
    float* sDP = sDZ + BBT * BLDZ;             // [16][8]  dpre_p of the step
    float* sDX = sDP + BBT * 8;                // [16][8]  dx_{t+1} (sum over the workgroups)
    float* sK1T = sDX + BBT * 8;               // [128][8] K1^T rows of the own gate columns
    int* sFlag = (int*)(sK1T + 128 * 8);
    float* sK2 = sK1T + 128 * 8 + 16;          // [4 waves][28 blocks][64 lanes][4]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int O = p.O, T = p.T_out;
    // (grid padded to a multiple of eight groups, members 8 blocks apart: xch_padded_groups, xch_common.h)
    const int group = (blockIdx.x / (8 * BG)) * 8 + (blockIdx.x & 7), slice = (blockIdx.x >> 3) & (BG - 1);
    if (group >= p.num_groups) { xch_spare_leaves(p.status); return; }
    constexpr int H4 = 4 * BH;
    const int hi = n >> 3;
    const int ul = 8 * wave + (n & 7);          // unit inside the workgroup (0..31)
    const int unit = 32 * slice + ul;
    const int my_row0 = 4 * g4 + 2 * hi;
    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    __shared__ unsigned sXch[4];
    const unsigned arrival = xch_arrive(p.status, sXch, group, slice);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // ---- resident transposed weights.  Tile tl of this wave: destination slice 2*wave + (tl>>1), half tl&1;
    // its output unit on this lane is nout; k index lc = 16*jb + 4*g4 + s is an own gate column:
    // gate lc>>5, unit 32*slice + (lc & 31). ----
    float r2t[4][8][4], r1t[4][8][4];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const int nout = 32 * (2 * wave + (tl >> 1)) + 16 * (tl & 1) + n;
#pragma unroll
        for (int jb = 0; jb < 8; ++jb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int lc = 16 * jb + 4 * g4 + s;
                const int gcol = (lc >> 5) * BH + 32 * slice + (lc & 31);
                r2t[tl][jb][s] = p.R2[(size_t)nout * H4 + gcol];
                r1t[tl][jb][s] = p.R1[(size_t)nout * H4 + gcol];
            }
    }
    // K2^T fragments of this wave (packed by mix_decoder_bwd_pack_k2: block b = tl*8 + jb): 28 blocks to LDS, 4 to registers
    float* sK2l = sK2 + (size_t)wave * BK2_LDS_BLOCKS * 256 + lane * 4;
    f32x4 k2r[4];
    {
        const f32x4* kp = (const f32x4*)p.K2p + (size_t)(slice * 4 + wave) * 32 * 64 + lane;
        for (int b = 0; b < BK2_LDS_BLOCKS; ++b) *(f32x4*)(sK2l + b * 256) = kp[64 * b];
#pragma unroll
        for (int b = 0; b < 4; ++b) k2r[b] = kp[64 * (BK2_LDS_BLOCKS + b)];
    }
    for (int e = tid; e < 128 * 8; e += 256) {
        const int lc = e >> 3, o = e & 7;
        sK1T[e] = (o < O) ? p.K1[(size_t)o * H4 + (lc >> 5) * BH + 32 * slice + (lc & 31)] : 0.f;
    }
    float wd[8];   // Dense kernel row of this lane's unit
#pragma unroll
    for (int o = 0; o < 8; ++o) wd[o] = (o < O) ? p.Wd[(size_t)unit * O + o] : 0.f;
    const int hrow = tid >> 4, ho = tid & 15;   // head: thread = (sequence of the tile, output)
    float wpr[8];   // row `ho` of Wp: dp[ho] = sum_k dm[k] Wp[ho][k]
#pragma unroll
    for (int k = 0; k < 8; ++k) wpr[k] = (ho < O && k < O) ? p.Wp[ho * O + k] : 0.f;

    // ---- exchange areas of this group ----
    unsigned long long* gbase = p.xch + (size_t)group * MB_GROUP;
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, (int)(2 * MB2 * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(gbase + 2 * MB2, 0, (int)(2 * MB1 * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(gbase + 2 * MB2 + 2 * MB1, 0, (int)(2 * MBX * 8), 0x00020000);
    unsigned epoch = 0;
    bool aborted = false;
#ifdef FOV_STAMPS
    const bool stamp_on = (blockIdx.x == 5 && tid == 0);
#endif
    auto give_up = [&]() {
        if (lane == 0) {
            xch_give_up(p.status);
            sFlag[0] = 1;
        }
    };
    // one quantity from the 8 sources (the feedback gradient dx), added in slice order
    auto gather_one = [&](const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned sstride, unsigned base) -> float {
        float part[8];
        unsigned bad = 0;
        {
            bwu32x2 v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + s * sstride, base, 16);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                part[s] = __uint_as_float(v[s].x);
                if (v[s].y != epoch) bad |= (1u << s);
            }
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > B_SPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                give_up();
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            bwu32x2 tv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) tv[s] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff + s * sstride, base, 16);
#pragma unroll
            for (int s = 0; s < 8; ++s)
                if (((bad >> s) & 1u) && tv[s].y == epoch) {
                    part[s] = __uint_as_float(tv[s].x);
                    bad &= ~(1u << s);
                }
        }
        float acc = 0.f;
#pragma unroll
        for (int s = 0; s < 8; ++s) acc += part[s];
        return acc;
    };
    // The partial sums of a lane's TWO cells (rows 2k, 2k + 1 of one unit) are adjacent tagged granules (round 3): one 16-byte
    // load per source brings both - eight loads instead of sixteen per gather, half the stores on the publishing side; every
    // 8-byte granule keeps its own epoch tag, the protocol's unit of atomicity is unchanged.
    auto gather_pair = [&](const __amdgpu_buffer_rsrc_t rs, unsigned voff, unsigned sstride, unsigned base, float (&out)[2]) {
        float part[2][8];
        unsigned bad = 0;
        {
            bwu32x4 v[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) v[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + s * sstride, base, 16);
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
            if (spins > B_SPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                give_up();
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            bwu32x4 tv[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) tv[s] = __builtin_amdgcn_raw_buffer_load_b128(rs, voff + s * sstride, base, 16);
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
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc += part[q][s];
            out[q] = acc;
        }
    };
    xch_hello_poll(p.status, sXch, group, BG, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    const XchTicket ticket = xch_ticket(sXch, arrival);
    epoch = ticket.base;
    aborted = sFlag[0] != 0;
    if (tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * BBT;
        __syncthreads();
        if (tid < BBT * 8) sDX[tid] = 0.f;       // no feedback gradient into the last step
        float dc1[2] = {0.f, 0.f}, dc2[2] = {0.f, 0.f}, dh1r[2] = {0.f, 0.f}, dh2r[2] = {0.f, 0.f};
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            MIXB_STAMP(0);
            ++epoch;
            const unsigned par2 = (epoch & 1u) * (unsigned)(MB2 * 8), par1 = (epoch & 1u) * (unsigned)(MB1 * 8),
                           parx = (epoch & 1u) * (unsigned)(MBX * 8);
            // tape of layer 2 for this lane's two cells: requested now, consumed after the head
            // (unconditional loads, rows clamped into the batch and masked where dz is formed: a load inside a branch is
            // waited for at the merge, i.e. right here)
            float tp[2][6];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                const size_t rowc = (size_t)(row < p.B ? row : p.B - 1);
                const float* rp = p.res2 + (((size_t)t * p.B + rowc) * 5) * BH + unit;
#pragma unroll
                for (int k = 0; k < 5; ++k) tp[r][k] = rp[k * BH];
                tp[r][5] = p.C2[((size_t)t * p.B + rowc) * BH + unit];
            }
            // ================= head backward (every workgroup, its 16 sequences) =================
            {
                const int brow = b0 + hrow;
                const bool live = ho < O && brow < p.B;
                const size_t hidx = ((size_t)t * p.B + brow) * O + ho;
                const size_t hidc = live ? hidx : 0;     // dead lanes read element 0, masked below
                const float mv = p.M[hidc], pv = p.P[hidc], dl = p.dloss[hidc];
                const float dm = live ? dl + sDX[hrow * 8 + (ho & 7)] * (1.f - mv * mv) : 0.f;
                float dp = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) dp = fmaf(__shfl(dm, (lane & ~15) | k), wpr[k], dp);
                const float dpp = dp * (1.f - pv * pv);
                if (ho < 8) sDP[hrow * 8 + ho] = live ? dpp : 0.f;
                if (slice == 0 && live) {
                    p.dpre_m[hidx] = dm;
                    p.dpre_p[hidx] = dpp;
                }
            }
            MIXB_STAMP(1);
            __syncthreads();   // dpre_p of the step is in LDS; every wave is past the previous step's MFMAs
            MIXB_STAMP(2);
            // ================= layer 2: gates backward for this lane's two cells =================
            float dz[2][4];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                float dhd = 0.f;
#pragma unroll
                for (int o = 0; o < 8; ++o) dhd = fmaf(sDP[(my_row0 + r) * 8 + o], wd[o], dhd);
                const float ig = tp[r][0], fg = tp[r][1], gg = tp[r][2], og = tp[r][3], ct = tp[r][4], cp = tp[r][5];
                const float tc = tanh_f(ct);
                const float dh = dhd + dh2r[r];
                const float dct = dc2[r] + dh * og * (1.f - tc * tc);
                const bool rl = row < p.B;
                dz[r][0] = rl ? dct * gg * bact_grad<ACT>(ig) : 0.f;
                dz[r][1] = rl ? dct * cp * bact_grad<ACT>(fg) : 0.f;
                dz[r][2] = rl ? dct * ig * (1.f - gg * gg) : 0.f;
                dz[r][3] = rl ? dh * tc * bact_grad<ACT>(og) : 0.f;
                dc2[r] = rl ? dct * fg : 0.f;
                if (row < p.B) {
                    float* zp = p.DZ2 + ((size_t)t * p.B + row) * H4 + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) zp[g * BH] = dz[r][g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) sDZ[(my_row0 + r) * BLDZ + g * 32 + ul] = (row < p.B) ? dz[r][g] : 0.f;
            }
            MIXB_STAMP(3);
            __syncthreads();   // the dz2 tile is in LDS
            // ================= partial[16 x 512] = dz2_own . [R2^T | K2^T]_own =================
            {
                f32x4 acc[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                bm_begin<8>(acc);
                const float* arow = sDZ + n * BLDZ + 4 * g4;
                // software pipeline over the 32 (jb, tl) blocks: the A fragment of the next jb and the K2^T fragment of
                // the next block are requested one iteration ahead (an LDS read right before its MFMAs costs ~100
                // cycles each time)
                auto k2frag = [&](int it) -> f32x4 {   // it = jb*4 + tl -> packed block tl*8 + jb
                    const int blk = (it & 3) * 8 + (it >> 2);
                    return (blk < BK2_LDS_BLOCKS) ? *(const f32x4*)(sK2l + blk * 256) : k2r[blk - BK2_LDS_BLOCKS];
                };
                f32x4 a = *(const f32x4*)arow, an = a;
                f32x4 kb = k2frag(0), kn = kb;
#pragma unroll
                for (int it = 0; it < 32; ++it) {
                    const int jb = it >> 2, tl = it & 3;
                    if (it + 1 < 32) kn = k2frag(it + 1);
                    if (tl == 0 && jb + 1 < 8) an = *(const f32x4*)(arow + 16 * (jb + 1));
                    if (tl == 0) asm volatile("s_nop 1" : "+v"(a));
                    asm volatile("s_nop 1" : "+v"(kb));
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        bm_a(acc[tl], a[s], r2t[tl][jb][s]);
                        bm_v(acc[4 + tl], a[s], kb[s]);
                    }
                    kb = kn;
                    if (tl == 3) a = an;
                }
                bm_end<8>(acc);
                // publish: tile tl of product q goes to destination d = 2*wave + (tl>>1), rows 4*g4 + r, unit 16*(tl&1) + n
#pragma unroll
                for (int q = 0; q < 2; ++q)
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        const int d = 2 * wave + (tl >> 1);
                        const unsigned off = (unsigned)(((((d * BG + slice) * 2 + q) * (BBT / 2) + 2 * g4) * 32) + 16 * (tl & 1) + n) * 16u;
#pragma unroll
                        for (int rp = 0; rp < 2; ++rp) {
                            const bwu32x4 gr = {__float_as_uint(acc[q * 4 + tl][2 * rp]), epoch, __float_as_uint(acc[q * 4 + tl][2 * rp + 1]), epoch};
                            if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, rs2, off + rp * 32 * 16, par2, 1);
                            else __builtin_amdgcn_raw_buffer_store_b128(gr, rs2, off + rp * 32 * 16, par2, 16);
                        }
                    }
            }
            // tape of layer 1: requested under the exchange wait below
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                const size_t rowc = (size_t)(row < p.B ? row : p.B - 1);
                const float* rp = p.res1 + (((size_t)t * p.B + rowc) * 5) * BH + unit;
#pragma unroll
                for (int k = 0; k < 5; ++k) tp[r][k] = rp[k * BH];
                tp[r][5] = p.C1[((size_t)t * p.B + rowc) * BH + unit];
            }
            MIXB_STAMP(4);
            __syncthreads();   // every wave is done reading the dz2 tile
            // ================= gather the 8 pieces of this lane's cells: dh2_{t-1} and dh1_t =================
            float dh1in[2];
            {
                // [dest = slice][src][q][row][unit]: src stride = 2*16*32 granules
                float sum[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const unsigned vo = (unsigned)((((slice * BG) * 2 + q) * (BBT / 2) + (my_row0 >> 1)) * 32 + ul) * 16u;
                    gather_pair(rs2, vo, 2 * BBT * 32 * 8, par2, sum);
                    if (q == 0) { dh2r[0] = sum[0]; dh2r[1] = sum[1]; }
                    else { dh1in[0] = sum[0]; dh1in[1] = sum[1]; }
                }
            }
            MIXB_STAMP(5);
            if (sFlag[0]) { aborted = true; }
            // ================= layer 1: gates backward =================
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                const float ig = tp[r][0], fg = tp[r][1], gg = tp[r][2], og = tp[r][3], ct = tp[r][4], cp = tp[r][5];
                const float tc = tanh_f(ct);
                const float dh = dh1in[r] + dh1r[r];
                const float dct = dc1[r] + dh * og * (1.f - tc * tc);
                const bool rl = row < p.B;
                dz[r][0] = rl ? dct * gg * bact_grad<ACT>(ig) : 0.f;
                dz[r][1] = rl ? dct * cp * bact_grad<ACT>(fg) : 0.f;
                dz[r][2] = rl ? dct * ig * (1.f - gg * gg) : 0.f;
                dz[r][3] = rl ? dh * tc * bact_grad<ACT>(og) : 0.f;
                dc1[r] = rl ? dct * fg : 0.f;
                if (row < p.B) {
                    float* zp = p.DZ1 + ((size_t)t * p.B + row) * H4 + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) zp[g * BH] = dz[r][g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) sDZ[(my_row0 + r) * BLDZ + g * 32 + ul] = (row < p.B) ? dz[r][g] : 0.f;
            }
            MIXB_STAMP(6);
            __syncthreads();   // the dz1 tile is in LDS
            // ================= dx partial = dz1_own . K1^T_own (16 x O): thread (sequence, output, half of the columns) ======
            {
                const int o = ho & 7, half = ho >> 3;
                const float* zr = sDZ + hrow * BLDZ + 64 * half;
                const float* kr = sK1T + (64 * half) * 8 + o;
                float sx = 0.f;
#pragma unroll
                for (int c4 = 0; c4 < 16; ++c4) {
                    const f32x4 zv = *(const f32x4*)(zr + 4 * c4);
#pragma unroll
                    for (int k = 0; k < 4; ++k) sx = fmaf(zv[k], kr[(4 * c4 + k) * 8], sx);
                }
                sx += __shfl_xor(sx, 8);   // the two column halves of one (sequence, output)
                if (ho < 8)
                    XCH_STORE_B64(ticket.same_xcd, ((bwu32x2){__float_as_uint(sx), epoch}), rsx,
                                                          (unsigned)((slice * BBT + hrow) * 8 + ho) * 8u, parx);
            }
            // ================= partial[16 x 256] = dz1_own . R1^T_own =================
            {
                f32x4 acc[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
                bm_begin<4>(acc);
                const float* arow = sDZ + n * BLDZ + 4 * g4;
                f32x4 a = *(const f32x4*)arow, an = a;
#pragma unroll
                for (int jb = 0; jb < 8; ++jb) {
                    if (jb + 1 < 8) an = *(const f32x4*)(arow + 16 * (jb + 1));
                    asm volatile("s_nop 1" : "+v"(a));
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl)
#pragma unroll
                        for (int s = 0; s < 4; ++s) bm_a(acc[tl], a[s], r1t[tl][jb][s]);
                    a = an;
                }
                bm_end<4>(acc);
#pragma unroll
                for (int tl = 0; tl < 4; ++tl) {
                    const int d = 2 * wave + (tl >> 1);
                    const unsigned off = (unsigned)((((d * BG + slice) * (BBT / 2) + 2 * g4) * 32) + 16 * (tl & 1) + n) * 16u;
#pragma unroll
                    for (int rp = 0; rp < 2; ++rp) {
                        const bwu32x4 gr = {__float_as_uint(acc[tl][2 * rp]), epoch, __float_as_uint(acc[tl][2 * rp + 1]), epoch};
                        if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, rs1, off + rp * 32 * 16, par1, 1);
                        else __builtin_amdgcn_raw_buffer_store_b128(gr, rs1, off + rp * 32 * 16, par1, 16);
                    }
                }
            }
            MIXB_STAMP(7);
            // ================= gather: dh1_{t-1} of this lane's cells, and dx_t (every workgroup needs all of it) =================
            {
                float sum[2];
                gather_pair(rs1, (unsigned)(((slice * BG) * (BBT / 2) + (my_row0 >> 1)) * 32 + ul) * 16u, BBT * 32 * 8, par1, sum);
                dh1r[0] = sum[0]; dh1r[1] = sum[1];
                // dx: thread (hrow, ho < 8) sums the 8 sources
                sum[0] = gather_one(rsx, (unsigned)(hrow * 8 + (ho & 7)) * 8u, BBT * 8 * 8, parx);
                __syncthreads();   // every wave is done with the dz1 tile and with sDX of this step
                if (ho < 8) sDX[hrow * 8 + ho] = sum[0];
            }
            MIXB_STAMP(8);
            if (sFlag[0]) aborted = true;
            __syncthreads();   // dx_t is in LDS (also makes `aborted` uniform below)
            if (sFlag[0]) { aborted = true; break; }
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    p.dh1_0[(size_t)row * BH + unit] = dh1r[r];
                    p.dc1_0[(size_t)row * BH + unit] = dc1[r];
                    p.dh2_0[(size_t)row * BH + unit] = dh2r[r];
                    p.dc2_0[(size_t)row * BH + unit] = dc2[r];
                }
            }
        }
    }
    xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

// K2 (H,4H) -> K2^T fragments in the order the kernel reads them:
// [slice 8][wave 4][block = tl*8 + jb, 32][lane 64][s 4] = K2[nout][gcol], nout = 32*(2*wave + (tl>>1)) + 16*(tl&1) + (lane&15),
// gcol = (lc>>5)*H + 32*slice + (lc&31), lc = 16*jb + 4*(lane>>4) + s
__global__ __launch_bounds__(256) void mix_decoder_bwd_pack_k2_kernel(const float* __restrict__ K2, float* __restrict__ out) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= BH * 4 * BH) return;
    const int s = idx & 3, lane = (idx >> 2) & 63, blk = (idx >> 8) & 31, wave = (idx >> 13) & 3, slice = idx >> 15;
    const int tl = blk >> 3, jb = blk & 7;
    const int nout = 32 * (2 * wave + (tl >> 1)) + 16 * (tl & 1) + (lane & 15);
    const int lc = 16 * jb + 4 * (lane >> 4) + s;
    const int gcol = (lc >> 5) * BH + 32 * slice + (lc & 31);
    out[idx] = K2[(size_t)nout * (4 * BH) + gcol];
}

// header + fixed granule area + the packed copy of K2^T
size_t mix_decoder_bwd_workspace_bytes(int B) {
    (void)B;
    return kStatusBytes + kXchBytes + sizeof(float) * (size_t)BH * 4 * BH;
}

// The packed copy of K2 depends on the weights only: a caller that runs it ahead of time (on a side stream, while the
// encoder layers run) calls this, then orders the streams, then launches - the launch finds the mark and skips its own pack.
int mix_decoder_bwd_prepack(const float* K2, void* workspace, hipStream_t stream) {
    float* k2p = (float*)((char*)workspace + kStatusBytes + kXchBytes);
    hipLaunchKernelGGL(mix_decoder_bwd_pack_k2_kernel, dim3(BH * 4 * BH / 256), dim3(256), 0, stream, K2, k2p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder_bwd_prepack launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    prepack_mark(workspace, K2);
    return FOV_OK;
}

int mix_decoder_bwd_launch(MixDecBwdParams p, const float* K2, int act, void* workspace, hipStream_t stream) {
    if (p.B == 0 || p.T_out == 0) return FOV_OK;
    p.num_tiles = (p.B + BBT - 1) / BBT;
    const int max_groups = device_cu_count() / BG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("fused mixing decoder backward needs at least %d CUs", BG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * MB_GROUP * sizeof(unsigned long long) > kXchBytes - kHelloBytes) { set_error("mix_decoder_bwd: granule area too small"); return FOV_ERR_WORKSPACE; }
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    float* k2p = (float*)((char*)workspace + kStatusBytes + kXchBytes);
    p.K2p = k2p;
    p.epoch_span = p.T_out * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;   // no memset: tags continue from the header
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    // the packed copy may already be there: mix_decoder_bwd_prepack (another stream, under the encoder) marks the workspace for ONE launch
    if (!prepack_consume(workspace, K2)) hipLaunchKernelGGL(mix_decoder_bwd_pack_k2_kernel, dim3(BH * 4 * BH / 256), dim3(256), 0, stream, K2, k2p);
    const size_t lds = sizeof(float) * (BBT * BLDZ + BBT * 8 + BBT * 8 + 128 * 8 + 16 + 4 * BK2_LDS_BLOCKS * 256);
    void (*kern)(MixDecBwdParams) = act == FOV_ACT_HARD_SIGMOID ? mix_decoder_bwd_kernel<FOV_ACT_HARD_SIGMOID> : mix_decoder_bwd_kernel<FOV_ACT_SIGMOID>;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3(xch_padded_groups(p.num_groups) * BG), dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder_bwd launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

#ifdef FOV_STAMPS
extern "C" int fov_debug_read_mixb_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_mixb_stamps), sizeof(unsigned long long) * BSTAMP_STEPS * BSTAMP_SLOTS);
}
#endif

}  // namespace fov
