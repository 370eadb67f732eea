// bf16 sibling of mix_decoder_bwd.hip (BASELINE.json configs[4]): BPTT through the whole unrolled others-mixing decoder
// (mixing head, layer 2, layer 1, feedback path; mycode/given_others_gt_mean_var_seq2seq.py:203-308) in ONE persistent
// launch, with bf16 operands into v_mfma_f32_16x16x32_bf16 for the three transposed products
//     dz2 . R2^T -> dh2_{t-1},   dz2 . K2^T -> dh1_t,   dz1 . R1^T -> dh1_{t-1}
// (dz rounded to bf16 in its LDS image, weights packed once into register-resident B fragments: 3 x 64 registers per
// lane, no LDS staging, no pre-pack launch), fp32 accumulation, fp32 partial sums on the wire, fp32 gates backward,
// head backward and dx = dz1 . K1^T (O <= 8 columns, VALU).  Same ownership, same exchange and the same outputs as the
// fp32 kernel: dz of both layers and the pre-activation gradients of the two head layers for every step (the weight
// gradients are then one product per layer over all steps) and the gradient w.r.t. the decoder's initial state.
// Per step and wave: 48 MFMAs of 16 cycles (fp32: 384 of 32 cycles) - the step is bound by its two exchanges.
#include <stdlib.h>

#include "bf16_common.h"

namespace fov {

namespace {

constexpr int DBLDQ = 128 + 16;    // bf16 LDS row stride of the dz tile (16 x 128 own gate columns): 72 dwords = 8 (mod 64)
constexpr int DBLDF = 128 + 4;     // fp32 copy of the dz1 tile for the VALU dx product

template <int ACT>
__device__ __forceinline__ float db_act_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

// granule areas of one group, in granules (8 bytes each); two parities of each
constexpr size_t QB2 = (size_t)QG * QG * 2 * QBT * 32;   // [dest][src][product][row][unit]
constexpr size_t QB1 = (size_t)QG * QG * QBT * 32;       // [dest][src][row][unit]
constexpr size_t QBX = (size_t)QG * QBT * 8;             // [src][row][o]
constexpr size_t QB_GROUP = 2 * (QB2 + QB1 + QBX);

template <int ACT>
__global__ __launch_bounds__(256, 1) void mix_decoder_bwd_bf16_kernel(MixDecBwdParams p) {
    __shared__ __attribute__((aligned(16))) unsigned short sDQ[QBT * DBLDQ];   // dz tile, bf16 (MFMA A operand)
    __shared__ __attribute__((aligned(16))) float sDZ[QBT * DBLDF];            // dz1 tile, fp32 (dx product)
    __shared__ float sDP[QBT * 8];               // dpre_p of the step
    __shared__ float sDX[QBT * 8];               // dx_{t+1} (sum over the workgroups)
    __shared__ float sK1T[128 * 8];              // K1^T rows of the own gate columns
    __shared__ int sFlag[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    const int O = p.O, T = p.T_out;
    int group, slice;
    q_group_slice(p.num_groups, group, slice);
    constexpr int H4 = 4 * QH;
    const int hi = n >> 3;
    const int ul = 8 * wave + (n & 7);          // unit inside the workgroup (0..31)
    const int unit = 32 * slice + ul;
    const int my_row0 = 4 * g4 + 2 * hi;
    const unsigned epoch_base = xch_epoch_base(p.status);
    const bool poisoned = xch_poisoned(p.status);
    if (tid == 0) { sFlag[0] = poisoned ? 1 : 0; sFlag[1] = 0; sFlag[2] = 0; }

    // ---- resident transposed weights as packed B fragments.  Tile tl of this wave: destination slice 2*wave + (tl>>1),
    // half tl&1; its output unit on this lane is nout; k index lc = 32*kb + 8*g4 + j is an own gate column: gate kb,
    // unit 32*slice + 8*g4 + j - eight CONTIGUOUS floats of row nout of R2 / K2 / R1. ----
    qu32x4 r2q[4][4], k2q[4][4], r1q[4][4];
#pragma unroll
    for (int tl = 0; tl < 4; ++tl) {
        const int nout = 32 * (2 * wave + (tl >> 1)) + 16 * (tl & 1) + n;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            const size_t o = (size_t)nout * H4 + kb * QH + 32 * slice + 8 * g4;
            r2q[tl][kb] = load_bfrag_rowmajor(p.R2 + o);
            k2q[tl][kb] = load_bfrag_rowmajor(p.K2p + o);   // K2p: here the plain (H,4H) kernel of layer 2
            r1q[tl][kb] = load_bfrag_rowmajor(p.R1 + o);
        }
    }
    for (int e = tid; e < 128 * 8; e += 256) {
        const int lc = e >> 3, o = e & 7;
        sK1T[e] = (o < O) ? p.K1[(size_t)o * H4 + (lc >> 5) * QH + 32 * slice + (lc & 31)] : 0.f;
    }
    float wd[8];   // Dense kernel row of this lane's unit
#pragma unroll
    for (int o = 0; o < 8; ++o) wd[o] = (o < O) ? p.Wd[(size_t)unit * O + o] : 0.f;
    const int hrow = tid >> 4, ho = tid & 15;   // head: thread = (sequence of the tile, output)
    float wpr[8];   // row `ho` of Wp: dp[ho] = sum_k dm[k] Wp[ho][k]
#pragma unroll
    for (int k = 0; k < 8; ++k) wpr[k] = (ho < O && k < O) ? p.Wp[ho * O + k] : 0.f;

    // ---- exchange areas of this group ----
    unsigned long long* gbase = p.xch + (size_t)group * QB_GROUP;
    const __amdgpu_buffer_rsrc_t rs2 = __builtin_amdgcn_make_buffer_rsrc(gbase, 0, (int)(2 * QB2 * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc(gbase + 2 * QB2, 0, (int)(2 * QB1 * 8), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(gbase + 2 * QB2 + 2 * QB1, 0, (int)(2 * QBX * 8), 0x00020000);
    unsigned epoch = epoch_base;
    bool aborted = false;
    // A give-up is recorded in the LDS flag of its gather phase (1: layer-2 pieces, 2: layer-1 pieces and dx).  Each flag
    // is read by every thread right after a workgroup barrier that follows all writes of that phase, and the next write
    // to the same flag lies behind a later barrier: the break below is uniform over the workgroup.
    auto give_up = [&](int slot) {
        if (lane == 0) {
            xch_give_up(p.status);
            sFlag[slot] = 1;
        }
    };
    // gather 2 x 8 granules {src 0..7} for this lane's two quantities and add them in slice order.
    auto gather_sum = [&](const __amdgpu_buffer_rsrc_t rs, const unsigned (&voff)[2], unsigned sstride, unsigned base, float (&out)[2],
                          int slot) {
        float part[16];
        unsigned bad = 0;
        {
            qu32x2 v[16];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int s = 0; s < 8; ++s) v[q * 8 + s] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff[q] + s * sstride, base, 16);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                part[j] = __uint_as_float(v[j].x);
                if (v[j].y != epoch) bad |= (1u << j);
            }
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > Q_SPIN || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                give_up(slot);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            qu32x2 tv[16];
#pragma unroll
            for (int q = 0; q < 2; ++q)
#pragma unroll
                for (int s = 0; s < 8; ++s) tv[q * 8 + s] = __builtin_amdgcn_raw_buffer_load_b64(rs, voff[q] + s * sstride, base, 16);
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if (((bad >> j) & 1u) && tv[j].y == epoch) {
                    part[j] = __uint_as_float(tv[j].x);
                    bad &= ~(1u << j);
                }
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float acc = 0.f;
#pragma unroll
            for (int s = 0; s < 8; ++s) acc += part[q * 8 + s];
            out[q] = acc;
        }
    };
    // publish the four 16 x 16 tiles of one product: tile tl -> destination 2*wave + (tl>>1), rows 4*g4 + r, unit 16*(tl&1) + n
    auto publish4 = [&](const __amdgpu_buffer_rsrc_t rs, const f32x4 (&acc)[4], int nprod, int q, unsigned par) {
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int d = 2 * wave + (tl >> 1);
            const unsigned off = (unsigned)(((((d * QG + slice) * nprod + q) * QBT + 4 * g4) * 32) + 16 * (tl & 1) + n) * 8u;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                __builtin_amdgcn_raw_buffer_store_b64((qu32x2){__float_as_uint(acc[tl][r]), epoch}, rs, off + r * 32 * 8, par, 16);
        }
    };
    __syncthreads();
    aborted = sFlag[0] != 0;

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * QBT;
        __syncthreads();
        if (tid < QBT * 8) sDX[tid] = 0.f;       // no feedback gradient into the last step
        float dc1[2] = {0.f, 0.f}, dc2[2] = {0.f, 0.f}, dh1r[2] = {0.f, 0.f}, dh2r[2] = {0.f, 0.f};
        __syncthreads();
        for (int t = T - 1; t >= 0; --t) {
            ++epoch;
            const unsigned par2 = (epoch & 1u) * (unsigned)(QB2 * 8), par1 = (epoch & 1u) * (unsigned)(QB1 * 8),
                           parx = (epoch & 1u) * (unsigned)(QBX * 8);
            // tape of layer 2 for this lane's two cells: requested now, consumed after the head
            float tp[2][6];
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
#pragma unroll
                for (int k = 0; k < 6; ++k) tp[r][k] = 0.f;
                if (row < p.B) {
                    const float* rp = p.res2 + (((size_t)t * p.B + row) * 5) * QH + unit;
#pragma unroll
                    for (int k = 0; k < 5; ++k) tp[r][k] = rp[k * QH];
                    tp[r][5] = p.C2[((size_t)t * p.B + row) * QH + unit];
                }
            }
            // ================= head backward (every workgroup, its 16 sequences) =================
            {
                const int brow = b0 + hrow;
                const bool live = ho < O && brow < p.B;
                const size_t hidx = ((size_t)t * p.B + brow) * O + ho;
                const float mv = live ? p.M[hidx] : 0.f, pv = live ? p.P[hidx] : 0.f;
                const float dm = live ? p.dloss[hidx] + sDX[hrow * 8 + (ho & 7)] * (1.f - mv * mv) : 0.f;
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
            __syncthreads();   // dpre_p of the step is in LDS; every wave is past the previous step's MFMAs
            // ================= layer 2: gates backward for this lane's two cells =================
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
                float dz[4];
                dz[0] = dct * gg * db_act_grad<ACT>(ig);
                dz[1] = dct * cp * db_act_grad<ACT>(fg);
                dz[2] = dct * ig * (1.f - gg * gg);
                dz[3] = dh * tc * db_act_grad<ACT>(og);
                dc2[r] = dct * fg;
                if (row < p.B) {
                    float* zp = p.DZ2 + ((size_t)t * p.B + row) * H4 + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) zp[g * QH] = dz[g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) sDQ[(my_row0 + r) * DBLDQ + g * 32 + ul] = bf16_bits((row < p.B) ? dz[g] : 0.f);
            }
            __syncthreads();   // the dz2 tile is in LDS
            // ================= partial[16 x 512] = dz2_own . [R2^T | K2^T]_own =================
            {
                f32x4 a2[4], ak[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { a2[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; ak[i] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const qu32x4 a = *(const qu32x4*)(sDQ + n * DBLDQ + 32 * kb + 8 * g4);
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        qmfma(a2[tl], a, r2q[tl][kb]);
                        qmfma(ak[tl], a, k2q[tl][kb]);
                    }
                }
                publish4(rs2, a2, 2, 0, par2);
                publish4(rs2, ak, 2, 1, par2);
            }
            // tape of layer 1: requested under the exchange wait below
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
#pragma unroll
                for (int k = 0; k < 6; ++k) tp[r][k] = 0.f;
                if (row < p.B) {
                    const float* rp = p.res1 + (((size_t)t * p.B + row) * 5) * QH + unit;
#pragma unroll
                    for (int k = 0; k < 5; ++k) tp[r][k] = rp[k * QH];
                    tp[r][5] = p.C1[((size_t)t * p.B + row) * QH + unit];
                }
            }
            __syncthreads();   // every wave is done reading the dz2 tile
            // ================= gather the 8 pieces of this lane's cells: dh2_{t-1} and dh1_t =================
            float dh1in[2];
            {
                unsigned voff[2];
                float sum[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    voff[0] = (unsigned)((((slice * QG) * 2 + q) * QBT + my_row0) * 32 + ul) * 8u;
                    voff[1] = voff[0] + 32 * 8;
                    gather_sum(rs2, voff, 2 * QBT * 32 * 8, par2, sum, 1);
                    if (q == 0) { dh2r[0] = sum[0]; dh2r[1] = sum[1]; }
                    else { dh1in[0] = sum[0]; dh1in[1] = sum[1]; }
                }
            }
            // ================= layer 1: gates backward =================
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                const float ig = tp[r][0], fg = tp[r][1], gg = tp[r][2], og = tp[r][3], ct = tp[r][4], cp = tp[r][5];
                const float tc = tanh_f(ct);
                const float dh = dh1in[r] + dh1r[r];
                const float dct = dc1[r] + dh * og * (1.f - tc * tc);
                float dz[4];
                dz[0] = dct * gg * db_act_grad<ACT>(ig);
                dz[1] = dct * cp * db_act_grad<ACT>(fg);
                dz[2] = dct * ig * (1.f - gg * gg);
                dz[3] = dh * tc * db_act_grad<ACT>(og);
                dc1[r] = dct * fg;
                if (row < p.B) {
                    float* zp = p.DZ1 + ((size_t)t * p.B + row) * H4 + unit;
#pragma unroll
                    for (int g = 0; g < 4; ++g) zp[g * QH] = dz[g];
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float v = (row < p.B) ? dz[g] : 0.f;
                    sDQ[(my_row0 + r) * DBLDQ + g * 32 + ul] = bf16_bits(v);
                    sDZ[(my_row0 + r) * DBLDF + g * 32 + ul] = v;
                }
            }
            __syncthreads();   // the dz1 tile is in LDS (also makes a give-up of the gather above uniform)
            if (sFlag[1]) { aborted = true; break; }
            // ================= dx partial = dz1_own . K1^T_own (16 x O): thread (sequence, output, half of the columns) ======
            {
                const int o = ho & 7, half = ho >> 3;
                const float* zr = sDZ + hrow * DBLDF + 64 * half;
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
                    __builtin_amdgcn_raw_buffer_store_b64((qu32x2){__float_as_uint(sx), epoch}, rsx,
                                                          (unsigned)((slice * QBT + hrow) * 8 + ho) * 8u, parx, 16);
            }
            // ================= partial[16 x 256] = dz1_own . R1^T_own =================
            {
                f32x4 a1[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) a1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int kb = 0; kb < 4; ++kb) {
                    const qu32x4 a = *(const qu32x4*)(sDQ + n * DBLDQ + 32 * kb + 8 * g4);
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) qmfma(a1[tl], a, r1q[tl][kb]);
                }
                publish4(rs1, a1, 1, 0, par1);
            }
            // ================= gather: dh1_{t-1} of this lane's cells, and dx_t (every workgroup needs all of it) =================
            {
                unsigned voff[2];
                float sum[2];
                voff[0] = (unsigned)(((slice * QG) * QBT + my_row0) * 32 + ul) * 8u;
                voff[1] = voff[0] + 32 * 8;
                gather_sum(rs1, voff, QBT * 32 * 8, par1, sum, 2);
                dh1r[0] = sum[0]; dh1r[1] = sum[1];
                // dx: thread (hrow, ho < 8) sums the 8 sources; the second quantity of gather_sum re-reads the same
                voff[0] = (unsigned)(hrow * 8 + (ho & 7)) * 8u;
                voff[1] = voff[0];
                gather_sum(rsx, voff, QBT * 8 * 8, parx, sum, 2);
                __syncthreads();   // every wave is done with the dz1 tile and with sDX of this step
                if (ho < 8) sDX[hrow * 8 + ho] = sum[0];
            }
            __syncthreads();   // dx_t is in LDS (also makes a give-up uniform below)
            if (sFlag[2]) { aborted = true; break; }
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    p.dh1_0[(size_t)row * QH + unit] = dh1r[r];
                    p.dc1_0[(size_t)row * QH + unit] = dc1[r];
                    p.dh2_0[(size_t)row * QH + unit] = dh2r[r];
                    p.dc2_0[(size_t)row * QH + unit] = dc2[r];
                }
            }
        }
    }
    xch_leave(p.status, (unsigned)p.epoch_span);
}

}  // namespace

int mix_decoder_bwd_bf16_launch(MixDecBwdParams p, const float* K2, int act, void* workspace, hipStream_t stream) {
    if (p.B == 0 || p.T_out == 0) return FOV_OK;
    p.num_tiles = (p.B + QBT - 1) / QBT;
    const int max_groups = device_cu_count() / QG;   // one workgroup per CU: every group must be co-resident
    if (max_groups < 1) { set_error("fused mixing decoder backward needs at least %d CUs", QG); return FOV_ERR_UNSUPPORTED; }
    p.num_groups = p.num_tiles < max_groups ? p.num_tiles : max_groups;
    if ((size_t)p.num_groups * QB_GROUP * sizeof(unsigned long long) > kXchBytes) { set_error("mix_decoder_bwd_bf16: granule area too small"); return FOV_ERR_WORKSPACE; }
    if ((((uintptr_t)K2) | ((uintptr_t)p.R1) | ((uintptr_t)p.R2)) & 15) { set_error("mix_decoder_bwd_bf16: kernels must be 16-byte aligned"); return FOV_ERR_INVALID; }
    p.status = (unsigned*)workspace;
    p.xch = (unsigned long long*)((char*)workspace + kStatusBytes);
    p.K2p = K2;   // no packed copy: fragments are eight contiguous floats of a row of the plain (H,4H) kernel
    p.epoch_span = p.T_out * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;
    void (*kern)(MixDecBwdParams) = act == FOV_ACT_HARD_SIGMOID ? mix_decoder_bwd_bf16_kernel<FOV_ACT_HARD_SIGMOID> : mix_decoder_bwd_bf16_kernel<FOV_ACT_SIGMOID>;
    hipLaunchKernelGGL(kern, dim3(p.num_groups * QG), dim3(256), 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("mix_decoder_bwd_bf16 launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
