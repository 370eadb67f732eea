// Persistent cluster LSTM kernel, H = 256, EIGHT workgroups per 16-sequence tile and TWO workgroups per CU.
//
// lstm_cluster.hip (four workgroups per tile, 64 units each) keeps a 256-register weight slice per lane, so one
// wave per SIMD: while that wave updates cells, waits for the partner slices or sits at a barrier, the matrix pipe
// idles (70 % / 63 % MFMA issue on the encoder / decoder of config 2).  Here a workgroup owns 32 hidden units, the
// recurrent slice is 128 AGPRs per lane and two workgroups - of different tiles - share a CU: the non-MFMA phases
// of one wave run under the MFMAs of the other.  At batch 1024 the grid is 64 tiles x 8 = 512 workgroups = exactly
// two per CU, all co-resident (the bounded waits below need every member of a group on the machine).
//
// Ownership as in mix_decoder.hip: wave w of workgroup `slice` owns units 32*slice + 8*w + (0..7); its gate
// columns form two MFMA N-tiles, [i | f] and [g | o]; after the MFMAs the halves of each 16-lane row swap what the
// other needs (DPP row_ror:8), so a lane holds all four gates of one unit for two sequences and c never leaves
// registers.  K slice (F <= 96) in LDS as lane-linear B fragments; the h tile (16 x 256) in LDS as the A operand;
// h_t exchanged as 8-byte {value, epoch} granules (sc1 stores / loads, two parity buffers, bounded spins).
// MODE_LAYER: an LSTM layer over x (optional hs, reserve, final state).  MODE_DECODE: the autoregressive decoder
// with Dense(F_dec,'tanh') inside the loop (Dense on the matrix pipe, K split over the four waves).
#include <stdlib.h>

#include "fov_common.h"

namespace fov {

namespace {

constexpr int H8 = 256;
constexpr int G8 = 8;
constexpr int BT8 = 16;
constexpr int LDH8 = H8 + 4;
constexpr int NG8 = 14;           // granules gathered per thread: 7 slices * 16 rows * 32 units / 256
constexpr int XR8 = 6;            // x prefetch registers per thread (16 rows * F <= 256 * XR8)
constexpr int C8_MAX_F = 96;
constexpr int C8_MAX_O = 8;
constexpr unsigned SPIN8 = 1u << 20;
constexpr int M8_LAYER = 0, M8_DECODE = 1;

typedef unsigned cu32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void mf_a(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}
__device__ __forceinline__ void mf_v(f32x4& acc, float a, float w_vgpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(w_vgpr));
}
__device__ __forceinline__ void mf_begin(f32x4 (&acc)[2]) { asm volatile("s_nop 3" : "+v"(acc[0]), "+v"(acc[1])); }
__device__ __forceinline__ void mf_end(f32x4 (&acc)[2]) {
    asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}
// hipcc pads nothing around inline asm: an MFMA operand that the compiler may have just written with a VALU
// instruction (a select, a copy made under register pressure) needs two wait states before the MFMA reads it
__device__ __forceinline__ void mf_guard2(float& a, float& b) { asm volatile("s_nop 1" : "+v"(a), "+v"(b)); }
__device__ __forceinline__ void mf_guard4(float& a, float& b, float& c, float& d) {
    asm volatile("s_nop 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
}
__device__ __forceinline__ float swap8(float v) {
    return __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(v), 0x128, 0xf, 0xf, false));
}
__host__ __device__ constexpr int r16(int v) { return (v + 15) & ~15; }

struct Lds8 {
    int ldx, off_k, off_h, off_x, off_wd, off_part, off_flag, total;
};
__host__ __device__ inline Lds8 lds8(int F, bool decode) {
    Lds8 L;
    const int fp = r16(F);
    L.ldx = fp + 4;
    L.off_h = 0;
    L.off_k = L.off_h + BT8 * LDH8;
    // LAYER: per wave (fp/16)*2 fragment blocks of 256 floats (+2 spare blocks for the run-ahead reads)
    const int kblocks = decode ? 0 : 4 * ((fp / 16) * 2 + 2);
    L.off_x = L.off_k + kblocks * 256;
    L.off_wd = L.off_x + (decode ? BT8 * 8 : 2 * BT8 * L.ldx + 64);
    L.off_part = L.off_wd + (decode ? H8 * 8 : 0);
    L.off_flag = L.off_part + (decode ? 4 * 256 : 0);
    L.total = L.off_flag + 16;
    return L;
}

// acc[tile] += A(h tile rows, LDS) . W (AGPR resident, [16 k-blocks][4][2 tiles])
__device__ __forceinline__ void recur8(f32x4 (&acc)[2], const float* hrow, const float (&w)[16][4][2]) {
    f32x4 a = *(const f32x4*)hrow;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        f32x4 an = a;
        if (j + 1 < 16) an = *(const f32x4*)(hrow + 16 * (j + 1));
        asm volatile("s_nop 1" : "+v"(a));   // a may have been moved by the compiler
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            mf_a(acc[0], a[s], w[j][s][0]);
            mf_a(acc[1], a[s], w[j][s][1]);
        }
        a = an;
    }
}

// acc[tile] += A(x tile rows, LDS) . K slice (LDS, lane-linear blocks (q, tile)).  No run-ahead registers: the
// kernel lives on 128 VGPRs and the second wave of the SIMD covers the LDS latency.
__device__ __forceinline__ void inproj8(f32x4 (&acc)[2], const float* xrow, const float* sKl, int nq) {
    for (int q = 0; q < nq; ++q) {
        const f32x4 a = *(const f32x4*)(xrow + 16 * q);
        const f32x4 b0 = *(const f32x4*)(sKl + (2 * q) * 256), b1 = *(const f32x4*)(sKl + (2 * q + 1) * 256);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            mf_v(acc[0], a[s], b0[s]);
            mf_v(acc[1], a[s], b1[s]);
        }
    }
}

template <int ACT, int MODE>
__global__ __launch_bounds__(256, 2) void lstm8_kernel(LstmParams p) {
    constexpr bool LAYER = (MODE == M8_LAYER);
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    int group, slice;
    if ((p.num_groups & 7) == 0) {   // members 8 blocks apart: likely one XCD (placement preference only)
        group = (blockIdx.x / (8 * G8)) * 8 + (blockIdx.x & 7);
        slice = (blockIdx.x >> 3) & (G8 - 1);
    } else {
        group = blockIdx.x / G8;
        slice = blockIdx.x - group * G8;
    }
    const float* Kp = LAYER ? p.K : p.dK;
    const float* Rp = LAYER ? p.R : p.dR;
    const float* bp = LAYER ? p.b : p.db;
    const int F = LAYER ? p.F : p.F_dec;
    const int steps = LAYER ? p.T : p.T_out;
    const int Fp = r16(F), nq = Fp >> 4;
    const Lds8 L = lds8(F, !LAYER);
    const int LDX = L.ldx;
    float* sH = smem + L.off_h;
    float* sK = smem + L.off_k;
    float* sX = smem + L.off_x;
    float* sWd = smem + L.off_wd;
    float* sPart = smem + L.off_part;
    int* sFlag = (int*)(smem + L.off_flag);

    const int unit = 32 * slice + 8 * wave + (n & 7);
    const int hi = n >> 3;
    const int col0 = hi * H8 + unit, col1 = (2 + hi) * H8 + unit;
    constexpr int H4 = 4 * H8;
    if (tid == 0) sFlag[0] = 0;
    if (p.clear_status && blockIdx.x == 0 && tid == 0) { p.status[0] = 0; p.status[1] = 0; }

    // ---- resident weights ----
    float w[16][4][2];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const size_t k = (size_t)(16 * j + 4 * g4 + s) * H4;
            w[j][s][0] = Rp[k + col0];
            w[j][s][1] = Rp[k + col1];
        }
    const float bv[2] = {bp ? bp[col0] : 0.f, bp ? bp[col1] : 0.f};
    float* sKl = sK + (size_t)wave * (nq * 2 + 2) * 256 + lane * 4;
    float k1[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
    float bdv = 0.f;
    if (LAYER) {
        for (int q = 0; q < nq; ++q)
#pragma unroll
            for (int tl = 0; tl < 2; ++tl) {
                f32x4 v;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k = 16 * q + 4 * g4 + s;
                    v[s] = (k < F) ? Kp[(size_t)k * H4 + (tl ? col1 : col0)] : 0.f;
                }
                *(f32x4*)(sKl + (2 * q + tl) * 256) = v;
            }
    } else {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int k = 4 * s + g4;
            k1[s][0] = (k < F) ? Kp[(size_t)k * H4 + col0] : 0.f;
            k1[s][1] = (k < F) ? Kp[(size_t)k * H4 + col1] : 0.f;
        }
        for (int e = tid; e < H8 * 8; e += 256) sWd[e] = ((e & 7) < F) ? p.dW[(size_t)(e >> 3) * F + (e & 7)] : 0.f;
        bdv = ((tid & 15) < F) ? p.dbias[tid & 15] : 0.f;
    }
    // zero the x tiles once: pad columns [F, Fp) and the spare tail are never written afterwards
    for (int i = tid; i < (LAYER ? 2 * BT8 * LDX + 64 : BT8 * 8); i += 256) sX[i] = 0.f;

    // ---- exchange bookkeeping ----
    const bool xch_used = !LAYER || steps > 1;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 2 * BT8 * H8, 0, 2 * BT8 * H8 * (int)sizeof(unsigned long long), 0x00020000);
    const int my_row0 = 4 * g4 + 2 * hi;
    const unsigned pub_off = (unsigned)(my_row0 * H8 + unit) * 8u;
    const unsigned gvoff = (unsigned)((tid >> 5) * H8 + (tid & 31)) * 8u;
    const int lbase = (tid >> 5) * LDH8 + (tid & 31);
    constexpr unsigned PARITY = BT8 * H8 * 8u;
    unsigned epoch = (unsigned)p.epoch_start;
    bool aborted = false;
    __syncthreads();

    cu32x2 v[NG8];
    auto gather_issue = [&](unsigned base) {
#pragma unroll
        for (int j = 0; j < NG8; ++j) {
            const unsigned uo = (unsigned)((j & 1) * 8 * H8 + ((slice + 1 + (j >> 1)) & (G8 - 1)) * 32) * 8u;
            v[j] = __builtin_amdgcn_raw_buffer_load_b64(xrs, gvoff, base + uo, 16);
        }
    };
    // First pass over the granules requested before the MFMAs: current ones go straight to the h tile, stale ones
    // are remembered in a bit mask; v[] is dead afterwards.  Retry sweeps (rare) re-read everything into
    // temporaries that live only inside the loop - a loop-carried copy of v[] cost ~9 VGPRs per granule.
    auto gather_finish = [&](unsigned base) {
        unsigned bad = 0;
#pragma unroll
        for (int j = 0; j < NG8; ++j) {
            const int lo = lbase + (j & 1) * 8 * LDH8 + ((slice + 1 + (j >> 1)) & (G8 - 1)) * 32;
            if (v[j].y == epoch) sH[lo] = __uint_as_float(v[j].x);
            else bad |= (1u << j);
        }
        unsigned spins = 0;
        while (__any(bad != 0)) {
            ++spins;
            if (spins > SPIN8 ||
                ((spins & 63u) == 0 && __hip_atomic_load(p.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                if (lane == 0) {
                    __hip_atomic_store(p.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    sFlag[0] = 1;
                }
                break;
            }
            __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            cu32x2 tv[NG8];
#pragma unroll
            for (int j = 0; j < NG8; ++j) {
                const unsigned uo = (unsigned)((j & 1) * 8 * H8 + ((slice + 1 + (j >> 1)) & (G8 - 1)) * 32) * 8u;
                tv[j] = __builtin_amdgcn_raw_buffer_load_b64(xrs, gvoff, base + uo, 16);
            }
#pragma unroll
            for (int j = 0; j < NG8; ++j) {
                const int lo = lbase + (j & 1) * 8 * LDH8 + ((slice + 1 + (j >> 1)) & (G8 - 1)) * 32;
                if (((bad >> j) & 1u) && tv[j].y == epoch) {
                    sH[lo] = __uint_as_float(tv[j].x);
                    bad &= ~(1u << j);
                }
            }
        }
    };

    const float* hrow = sH + n * LDH8 + 4 * g4;
    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * BT8;
        __syncthreads();   // previous tile fully consumed
        // ---- initial state ----
        for (int e = tid; e < BT8 * H8; e += 256) {
            const int row = e >> 8, u = e & 255;
            sH[row * LDH8 + u] = (b0 + row < p.B && p.h0) ? p.h0[(size_t)(b0 + row) * H8 + u] : 0.f;
        }
        float c[2], hc[2] = {0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = b0 + my_row0 + r;
            c[r] = (row < p.B && p.c0) ? p.c0[(size_t)row * H8 + unit] : 0.f;
            hc[r] = (row < p.B && p.h0) ? p.h0[(size_t)row * H8 + unit] : 0.f;
        }
        // x staging (LAYER): thread (xrw = tid/16, xcl = tid%16) moves columns xcl + 16*i of row xrw
        const int xrw = tid >> 4, xcl = tid & 15;
        const bool xlive = LAYER && (b0 + xrw < p.B);
        const float* xt = LAYER ? p.x + ((size_t)(b0 + xrw) * p.T) * F + xcl : nullptr;
        float* xl = sX + xrw * LDX + xcl;
        if (LAYER) {
#pragma unroll
            for (int tt = 0; tt < 2; ++tt)
                if (tt < steps) {
#pragma unroll
                    for (int i = 0; i < XR8; ++i)
                        if (xcl + 16 * i < F) xl[tt * BT8 * LDX + 16 * i] = xlive ? xt[(size_t)tt * F + 16 * i] : 0.f;
                }
        } else if (tid < BT8 * 8) {
            const int row = tid >> 3, o = tid & 7;
            sX[tid] = (o < F && b0 + row < p.B) ? p.dec_in0[(size_t)(b0 + row) * F + o] : 0.f;
        }
        __syncthreads();
        // ---- pre-activations of step 0 ----
        f32x4 acc[2];
        acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
        acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
        if (steps > 0) {
            mf_begin(acc);
            if (LAYER) {
                inproj8(acc, sX + n * LDX + 4 * g4, sKl, nq);
            } else {
                float xa0 = sX[n * 8 + g4], xa1 = sX[n * 8 + 4 + g4];
                mf_guard2(xa0, xa1);
                mf_guard4(k1[0][0], k1[0][1], k1[1][0], k1[1][1]);
                mf_v(acc[0], xa0, k1[0][0]); mf_v(acc[1], xa0, k1[0][1]);
                mf_v(acc[0], xa1, k1[1][0]); mf_v(acc[1], xa1, k1[1][1]);
            }
            recur8(acc, hrow, w);
            mf_end(acc);
        }
        float xr[XR8];
#pragma unroll
        for (int i = 0; i < XR8; ++i) xr[i] = 0.f;
        for (int t = 0; t < steps; ++t) {
            // ---- x pipeline (LAYER): x_{t+1}, requested during step t-1, goes registers -> LDS (its tile was last
            // read during step t-1, before that step's barriers); then x_{t+2} is requested ----
            if (LAYER && t > 0 && t + 1 < steps) {
                float* xb = xl + ((t + 1) & 1) * BT8 * LDX;
#pragma unroll
                for (int i = 0; i < XR8; ++i)
                    if (xcl + 16 * i < F) xb[16 * i] = xr[i];
            }
            if (LAYER && t + 2 < steps) {
                const float* xn = xt + (size_t)(t + 2) * F;
#pragma unroll
                for (int i = 0; i < XR8; ++i) xr[i] = (xlive && xcl + 16 * i < F) ? xn[16 * i] : 0.f;
            }
            // ---- cell update: swap halves, then two (sequence, unit) cells per lane ----
            {
                float snd[4], rcv[4];
                snd[0] = hi ? acc[0][0] : acc[0][2];
                snd[1] = hi ? acc[0][1] : acc[0][3];
                snd[2] = hi ? acc[1][0] : acc[1][2];
                snd[3] = hi ? acc[1][1] : acc[1][3];
#pragma unroll
                for (int k = 0; k < 4; ++k) rcv[k] = swap8(snd[k]);
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const float zi = hi ? rcv[r] : acc[0][r];
                    const float zf = hi ? acc[0][2 + r] : rcv[r];
                    const float zg = hi ? rcv[2 + r] : acc[1][r];
                    const float zo = hi ? acc[1][2 + r] : rcv[2 + r];
                    const float ig = rec_act<ACT>(zi), fg = rec_act<ACT>(zf), gg = tanh_f(zg), og = rec_act<ACT>(zo);
                    c[r] = fmaf(fg, c[r], ig * gg);
                    hc[r] = og * tanh_f(c[r]);
                    if (LAYER) {
                        const int row = b0 + my_row0 + r;
                        if (row < p.B) {
                            if (p.reserve) {
                                float* rp = p.reserve + (((size_t)row * p.T + t) * 5) * H8 + unit;
                                rp[0] = ig; rp[H8] = fg; rp[2 * H8] = gg; rp[3 * H8] = og; rp[4 * H8] = c[r];
                            }
                            if (p.hs) p.hs[((size_t)row * p.T + t) * H8 + unit] = hc[r];
                        }
                    }
                }
            }
            const bool more = (t + 1 < steps);
            const bool do_xch = xch_used && (!LAYER || more);   // a layer's last h_t is needed by nobody in here
            unsigned par = 0;
            if (do_xch) {
                ++epoch;
                par = (epoch & 1u) * PARITY;
#pragma unroll
                for (int r = 0; r < 2; ++r)
                    __builtin_amdgcn_raw_buffer_store_b64((cu32x2){__float_as_uint(hc[r]), epoch}, xrs, pub_off + r * H8 * 8, par, 16);
            }
            __syncthreads();   // barrier 1: every wave is done reading sH (and, LAYER, x_{t+1} is in LDS)
            if (do_xch) {
#pragma unroll
                for (int r = 0; r < 2; ++r) sH[(my_row0 + r) * LDH8 + unit] = hc[r];
                gather_issue(par);
            }
            // pre-activations of step t+1 that need no remote data run under the gather
            acc[0] = (f32x4){bv[0], bv[0], bv[0], bv[0]};
            acc[1] = (f32x4){bv[1], bv[1], bv[1], bv[1]};
            if (LAYER && more) {
                mf_begin(acc);
                inproj8(acc, sX + ((t + 1) & 1) * BT8 * LDX + n * LDX + 4 * g4, sKl, nq);
                mf_end(acc);
            }
            if (do_xch) gather_finish(par);
            __syncthreads();   // barrier 2: the whole h_t tile is in LDS
            if (sFlag[0]) { aborted = true; break; }
            if (!LAYER) {
                // ---- y_t = tanh(h_t Wd + bias) on the matrix pipe: wave w contracts units [64w, 64w+64) ----
                f32x4 dacc[2];
                dacc[0] = (f32x4){0.f, 0.f, 0.f, 0.f};
                dacc[1] = (f32x4){0.f, 0.f, 0.f, 0.f};
                const float* hq = hrow + 64 * wave;
                const float* wq = sWd + (64 * wave + 4 * g4) * 8 + (n & 7);
                mf_begin(dacc);
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const f32x4 hb = *(const f32x4*)(hq + 16 * b);
                    float wb[4];
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) wb[ss] = (n < 8) ? wq[(16 * b + ss) * 8] : 0.f;
                    mf_guard4(wb[0], wb[1], wb[2], wb[3]);   // written by the select above
#pragma unroll
                    for (int ss = 0; ss < 4; ++ss) mf_v(dacc[ss & 1], hb[ss], wb[ss]);
                }
                mf_end(dacc);
#pragma unroll
                for (int r = 0; r < 4; ++r) sPart[(wave * 16 + 4 * g4 + r) * 16 + n] = dacc[0][r] + dacc[1][r];
                __syncthreads();   // barrier 3: the four partial products are in LDS
                const int row = tid >> 4, o = tid & 15;
                if (o < 8) {
                    float yv = 0.f;
                    if (o < F) {
                        yv = sPart[row * 16 + o] + sPart[(16 + row) * 16 + o] + sPart[(32 + row) * 16 + o] + sPart[(48 + row) * 16 + o];
                        yv = tanh_f(yv + bdv);
                        if (slice == 0 && b0 + row < p.B) p.out[((size_t)(b0 + row) * p.T_out + t) * F + o] = yv;
                    }
                    sX[row * 8 + o] = yv;   // fed back
                }
                __syncthreads();   // barrier 4: y_t is in LDS
                if (more) {
                    float xa0 = sX[n * 8 + g4], xa1 = sX[n * 8 + 4 + g4];
                    mf_guard4(k1[0][0], k1[0][1], k1[1][0], k1[1][1]);
                    mf_guard2(xa0, xa1);
                    mf_begin(acc);
                    mf_v(acc[0], xa0, k1[0][0]); mf_v(acc[1], xa0, k1[0][1]);
                    mf_v(acc[0], xa1, k1[1][0]); mf_v(acc[1], xa1, k1[1][1]);
                    recur8(acc, hrow, w);
                    mf_end(acc);
                }
            } else if (more) {
                mf_begin(acc);
                recur8(acc, hrow, w);
                mf_end(acc);
            }
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int row = b0 + my_row0 + r;
                if (row < p.B) {
                    if (p.hT) p.hT[(size_t)row * H8 + unit] = hc[r];
                    if (p.cT) p.cT[(size_t)row * H8 + unit] = c[r];
                }
            }
        }
    }
}

typedef void (*Kern8)(LstmParams);
Kern8 pick8(int act, int mode) {
    if (act == FOV_ACT_HARD_SIGMOID) return mode == M8_DECODE ? lstm8_kernel<FOV_ACT_HARD_SIGMOID, M8_DECODE> : lstm8_kernel<FOV_ACT_HARD_SIGMOID, M8_LAYER>;
    return mode == M8_DECODE ? lstm8_kernel<FOV_ACT_SIGMOID, M8_DECODE> : lstm8_kernel<FOV_ACT_SIGMOID, M8_LAYER>;
}

int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
        if (cus <= 0) cus = 256;
    }
    return cus;
}

}  // namespace

bool cluster8_shape_ok(int F, int H, bool decode, int F_dec) {
    if (H != H8) return false;
    if (decode) return F >= 1 && F <= C8_MAX_F && F_dec >= 1 && F_dec <= C8_MAX_O;
    return F >= 1 && F <= C8_MAX_F;
}

// Resident groups for one launch: every member of a group must be on the machine at once, so the group count
// comes from the occupancy the runtime reports for the kernel with its LDS size (2 workgroups per CU when the
// registers and LDS allow it).
static int groups8(Kern8 kern, size_t lds, int tiles) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void*)kern, 256, lds) != hipSuccess || per_cu < 1) per_cu = 1;
    if (per_cu > 2) per_cu = 2;
    int groups = device_cus() * per_cu / G8;
    if (groups > 64) groups = 64;   // the granule area of the workspace is sized for 64 groups
    if (groups < 1) groups = 1;
    return tiles < groups ? (tiles > 0 ? tiles : 1) : groups;
}

static int launch8_mode(LstmParams p, int mode, hipStream_t stream) {
    Kern8 kern = pick8(p.act, mode);
    const int F = mode == M8_DECODE ? p.F_dec : p.F;
    const size_t lds = sizeof(float) * (size_t)lds8(F, mode == M8_DECODE).total;
    hipError_t e = hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { set_error("hipFuncSetAttribute: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    p.num_groups = groups8(kern, lds, p.num_tiles);
    hipLaunchKernelGGL(kern, dim3(p.num_groups * G8), dim3(256), lds, stream, p);
    e = hipGetLastError();
    if (e != hipSuccess) { set_error("cluster8 launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// Same contract as launch_cluster (lstm_cluster.hip): p.status / p.xch point into the caller's workspace, whose
// granule area (sized by cluster_workspace_bytes for up to 64 groups) and state area are laid out identically.
int launch_cluster8(const LstmParams& p_in, bool decode, size_t xch_bytes_with_status, hipStream_t stream) {
    LstmParams p = p_in;
    if (p.B == 0) return FOV_OK;
    p.num_tiles = (p.B + BT8 - 1) / BT8;
    p.force_safe_exchange = 1;
    p.epoch_start = 0;
    p.clear_status = (!decode && p.T <= 1) ? 1 : 0;
    if (!p.clear_status) {
        hipError_t e = hipMemsetAsync((void*)p.status, 0, xch_bytes_with_status, stream);
        if (e != hipSuccess) { set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    }
    if (!decode) return launch8_mode(p, M8_LAYER, stream);
    LstmParams enc = p;
    float* state = (float*)((char*)p.status + xch_bytes_with_status);
    enc.hT = state;
    enc.cT = state + (size_t)p.B * p.H;
    enc.hs = nullptr;
    int rc = launch8_mode(enc, M8_LAYER, stream);
    if (rc) return rc;
    // the decoder launch continues the epoch count on the same granule buffers (no second memset): every stale
    // tag is smaller than any tag it waits for.  A group visits at most `tiles` tiles with T epochs each.
    LstmParams dec = p;
    dec.h0 = enc.hT;
    dec.c0 = enc.cT;
    dec.clear_status = 0;
    dec.epoch_start = p.T * p.num_tiles;
    return launch8_mode(dec, M8_DECODE, stream);
}

// Decoder launch of a fused call whose encoder ran on the four-workgroup kernel: same granule buffers (tags of
// the encoder's epochs are all <= epoch_start), state seeded by the caller.
int launch_cluster8_decoder(const LstmParams& dec_in, int epoch_start, hipStream_t stream) {
    LstmParams dec = dec_in;
    dec.num_tiles = (dec.B + BT8 - 1) / BT8;
    dec.clear_status = 0;
    dec.epoch_start = epoch_start;
    return launch8_mode(dec, M8_DECODE, stream);
}

}  // namespace fov
