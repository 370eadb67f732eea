// Persistent cluster BPTT kernel for gfx950 (MI355X): the backward recurrence of one LSTM layer in
// ONE launch, the sibling of lstm_cluster.hip.  (The reference delegates this to TensorFlow
// autodiff under model.fit - mycode/FoV_seq2seq.py:103,112-117.)
//
// Same decomposition as the forward: a group of G = H/64 workgroups owns a 16-sequence tile,
// workgroup `slice` owns hidden units [64*slice, 64*slice+64) for all four gates; wave w / lane
// (n, g4) owns unit 16w+n of that slice for sequences 4*g4 .. 4*g4+3 and keeps their running dc
// and dh in registers.
//
// Per step t (descending): the lane turns dh_t, dc into the four pre-activation gradients dz of its
// elements (reserve = i,f,g,o,c from the training forward), stores them to dZ (B,T,4H) for the
// weight-gradient GEMMs, and writes them into an LDS tile (16 x 256 own gate columns, MFMA A
// layout).  dh_{t-1} = dz_t . R^T is then split by OUTPUT unit: wave w computes, for every
// destination slice d, the 16 units [64d+16w, 64d+16w+16) from the workgroup's own 256 gate columns
// (64 MFMAs per destination; R^T fragment = H AGPRs per lane for the whole sequence).  That
// (16 x 16) partial belongs to wave w of workgroup d, SAME lane: remote partials travel as 8-byte
// {value, epoch} granules (4 per lane per destination), the local one never leaves its registers.
// Destinations are visited remote-first and the gather sweep is issued before the local
// destination's MFMAs, so most of the exchange latency hides under them.  The G partials of an
// element are summed in slice order 0..G-1 (deterministic).
#include <stdlib.h>

#include "fov_common.h"
#include "xch_common.h"

namespace fov {

constexpr int BBT = 16;
constexpr unsigned BSPIN_LIMIT = 1u << 20;
#ifndef FOV_GATHER_AFTER_Q
#define FOV_GATHER_AFTER_Q 8
#endif
constexpr int GATHER_AFTER_Q = FOV_GATHER_AFTER_Q;   // k-blocks of the local destination's product issued before the gather is requested
                                    // (0 = right behind the last publish: 225 us; 8: 206 us; 15: 212 us at B = 1024, T = 30)
typedef unsigned bu32x2 __attribute__((ext_vector_type(2)));
typedef unsigned bu32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void bmfma_va(f32x4& acc, float a, float w_agpr) {
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(w_agpr));
}

struct BwdParams {
    const float* R;
    const float* reserve;   // (B,T,5,H)
    const float* c0;        // (B,H) or NULL
    const float* dhs;       // (B,T,H) or NULL
    const float* dhT;       // (B,H) or NULL
    const float* dcT;       // (B,H) or NULL
    float* dz;              // (B,T,4H) out
    float* dh0;             // (B,H) or NULL
    float* dc0;             // (B,H) or NULL
    float* db_part;         // (num_tiles, 4H) per-tile bias-gradient partials (sum over the tile's rows and t), or NULL
    unsigned long long* xch;
    unsigned* status;
    int B, T, H, num_groups, num_tiles;
    int epoch_span;         // epochs this launch may consume (xch_common.h)
};

template <int ACT>
__device__ __forceinline__ float bwd_act_grad(float a) {
    return ACT == FOV_ACT_HARD_SIGMOID ? ((a > 0.f && a < 1.f) ? 0.2f : 0.f) : a * (1.f - a);
}

template <int H, int ACT>
__global__ __launch_bounds__(256, 1) void lstm_bwd_cluster_kernel(BwdParams p) {
    constexpr int G = H / 64;
    constexpr int LDZ = 256 + 4;
    __shared__ __attribute__((aligned(16))) float sZ[BBT * LDZ];   // dz tile, columns kc = gate*64 + unit_in_slice
    __shared__ int sFlag[4];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 15, g4 = lane >> 4;
    // G > 1: members 8 blocks apart on a grid padded to a multiple of eight groups - one XCD (xch_padded_groups, xch_common.h)
    int group, slice;
    if (G > 1) {
        group = (blockIdx.x / (8 * G)) * 8 + (blockIdx.x & 7);
        slice = (blockIdx.x >> 3) & (G - 1);
        if (group >= p.num_groups) { xch_spare_leaves(p.status); return; }
    } else {
        group = blockIdx.x;
        slice = 0;
    }
    const int unit = slice * 64 + wave * 16 + n;   // the hidden unit this lane owns
    // epoch tags continue from the workspace header, a poisoned workspace skips the body (xch_common.h)
    __shared__ unsigned sXch[4];
    const unsigned arrival = G > 1 ? xch_arrive(p.status, sXch, group, slice) : 0u;
    const bool poisoned = G > 1 && xch_poisoned(p.status);
    if (tid == 0) sFlag[0] = poisoned ? 1 : 0;

    // R^T fragments: destination dd visits slice d(dd) = (slice + 1 + dd) mod G, so the local one is last.
    // wB[dd][q][s] = R[j = 64 d + 16 wave + n][col(kc = 16q + 4 g4 + s)], col = gate*H + 64 slice + unit_in_slice
    float wB[G][16][4];
    {
        const int H4 = 4 * H;
#pragma unroll
        for (int dd = 0; dd < G; ++dd) {
            const int d = (slice + 1 + dd) & (G - 1);
            const float* rrow = p.R + (size_t)(64 * d + 16 * wave + n) * H4 + 64 * slice;
#pragma unroll
            for (int q = 0; q < 16; ++q)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kc = 16 * q + 4 * g4 + s;
                    wB[dd][q][s] = rrow[(kc >> 6) * H + (kc & 63)];
                }
        }
    }

    // granule buffers of this group: [parity][dst slice][src slice][wave][r][lane]
    constexpr int CHUNK = 4 * 4 * 64;   // granules one workgroup sends to one destination per step
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(
        p.xch + (size_t)group * 2 * G * G * CHUNK, 0, 2 * G * G * CHUNK * (int)sizeof(unsigned long long), 0x00020000);
    // chunk order [wave][register pair][lane][register of the pair]: a lane's registers 2k, 2k + 1 are adjacent tagged granules,
    // moved by ONE 16-byte store / load (round 3: half the exchange instructions, the 8-byte unit of atomicity unchanged)
    const unsigned lane_off = (unsigned)((wave * 2) * 64 + lane) * 16u;   // + k*64*16 per register pair

    if (G > 1) xch_hello_poll(p.status, sXch, group, G, &sFlag[0]);   // same-XCD handshake (xch_common.h): partners' words, published at entry
    __syncthreads();
    XchTicket ticket = {0u, 0u, 0u};
    if (G > 1) ticket = xch_ticket(sXch, arrival);
    unsigned epoch = ticket.base;
    bool aborted = sFlag[0] != 0;
    if (G > 1 && tid == 0 && !ticket.same_xcd && !aborted) xch_count_safe(p.status, ticket);   // (fov_exchange_mode)

    for (int tile = group; tile < p.num_tiles && !aborted; tile += p.num_groups) {
        const int b0 = tile * BBT;
        float dc[4], dh[4];
        bool live[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = b0 + 4 * g4 + r;
            live[r] = row < p.B;
            dc[r] = (live[r] && p.dcT) ? p.dcT[(size_t)row * H + unit] : 0.f;
            dh[r] = (live[r] && p.dhT) ? p.dhT[(size_t)row * H + unit] : 0.f;
        }
        // reserve pipeline: cur = step t, nxt = step t-1 (its c is c_{t-1} of step t).  The loads are UNCONDITIONAL - rows
        // and steps clamped into the tensors, absent tensors replaced by a valid address, everything masked where it is
        // used: a load inside a branch is waited for at the merge, and the prefetch of step t-2 then stalls step t.
        float cur[5][4], nxt[5][4], dhs_cur[4], dhs_nxt[4];
        auto load_step = [&](int t, float (&dst)[5][4], float (&dd_)[4]) {
            const int tc = t > 0 ? t : 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = b0 + 4 * g4 + r;
                const size_t rowc = (size_t)(row < p.B ? row : p.B - 1);
                const float* rp = p.reserve + ((rowc * p.T + tc) * 5) * H + unit;
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q][r] = rp[q * H];
                const float* cp = t >= 0 ? rp + 4 * H : (p.c0 ? p.c0 + rowc * H + unit : rp);   // c_{-1} = c0 (none: masked at use)
                dst[4][r] = *cp;
                const float* dp = p.dhs ? p.dhs + (rowc * p.T + tc) * H + unit : rp;
                dd_[r] = *dp;
            }
        };
        load_step(p.T - 1, cur, dhs_cur);
        load_step(p.T - 2, nxt, dhs_nxt);
        float dbacc[4] = {0.f, 0.f, 0.f, 0.f};   // sum over t and this lane's 4 sequences of dz, per gate

        for (int t = p.T - 1; t >= 0; --t) {
            // ---- pointwise: dz of this lane's four elements ----
            float dzv[4][4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ig = cur[0][r], fg = cur[1][r], gg = cur[2][r], og = cur[3][r], cc = cur[4][r];
                const float cprev = (t > 0 || p.c0) ? nxt[4][r] : 0.f;   // step 0 without a given state: c_{-1} = 0
                const float dht = dh[r] + (p.dhs ? dhs_cur[r] : 0.f);
                const float tc = tanh_f(cc);
                const float dcv = dc[r] + dht * og * (1.f - tc * tc);
                dzv[0][r] = live[r] ? dcv * gg * bwd_act_grad<ACT>(ig) : 0.f;
                dzv[1][r] = live[r] ? dcv * cprev * bwd_act_grad<ACT>(fg) : 0.f;
                dzv[2][r] = live[r] ? dcv * ig * (1.f - gg * gg) : 0.f;
                dzv[3][r] = live[r] ? dht * tc * bwd_act_grad<ACT>(og) : 0.f;
                dc[r] = live[r] ? dcv * fg : 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) dbacc[g] += dzv[g][r];
#pragma unroll
                for (int g = 0; g < 4; ++g) sZ[(4 * g4 + r) * LDZ + g * 64 + wave * 16 + n] = dzv[g][r];
            }
            // rotate the reserve pipeline and request step t-2
#pragma unroll
            for (int q = 0; q < 5; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) cur[q][r] = nxt[q][r];
#pragma unroll
            for (int r = 0; r < 4; ++r) dhs_cur[r] = dhs_nxt[r];
            load_step(t - 2, nxt, dhs_nxt);
            __syncthreads();   // barrier A: the dz tile is complete
            // a give-up of the PREVIOUS step's gather is acted on here, where every wave sees the same flag
            if (G > 1 && sFlag[0]) { aborted = true; break; }

            // ---- A fragments of the whole tile row (16 x b128), reused for every destination ----
            f32x4 afr[16];
            const float* arow = sZ + n * LDZ + 4 * g4;
#pragma unroll
            for (int q = 0; q < 16; ++q) afr[q] = *(const f32x4*)(arow + 16 * q);

            ++epoch;
            const unsigned xsoff = (epoch & 1u) * (unsigned)(G * G * CHUNK * sizeof(unsigned long long));
            f32x4 part = (f32x4){0.f, 0.f, 0.f, 0.f};   // the local destination's partial
            bu32x4 v[G > 1 ? (G - 1) * 2 : 1];
#pragma unroll
            for (int dd = 0; dd < G; ++dd) {
                f32x4 a4[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) a4[s] = (f32x4){0.f, 0.f, 0.f, 0.f};
                asm volatile("s_nop 1" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]));
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    if (dd == G - 1 && G > 1 && q == GATHER_AFTER_Q) {
                        // all remote partials are on their way, the last of them only just: the gather sweep is requested
                        // part-way through the local destination's MFMAs (an sc1 store takes most of a microsecond to
                        // become visible; a sweep issued right behind the partners' last publish came back stale)
#pragma unroll
                        for (int k = 0; k < G - 1; ++k) {
                            const int s_src = (slice + 1 + k) & (G - 1);
#pragma unroll
                            for (int r = 0; r < 2; ++r)
                                v[k * 2 + r] = __builtin_amdgcn_raw_buffer_load_b128(
                                    xrs, (unsigned)((slice * G + s_src) * CHUNK) * 8u + lane_off + r * 1024u, xsoff, 16);
                        }
                    }
#pragma unroll
                    for (int s = 0; s < 4; ++s) bmfma_va(a4[s], afr[q][s], wB[dd][q][s]);
                }
                asm volatile("s_nop 7\n\ts_nop 7\n\ts_nop 7" : "+v"(a4[0]), "+v"(a4[1]), "+v"(a4[2]), "+v"(a4[3]));
                f32x4 sum;
#pragma unroll
                for (int r = 0; r < 4; ++r) sum[r] = (a4[0][r] + a4[1][r]) + (a4[2][r] + a4[3][r]);
                if (dd == G - 1) {
                    part = sum;
                } else {
                    const int d = (slice + 1 + dd) & (G - 1);
#pragma unroll
                    for (int r = 0; r < 2; ++r) {
                        const bu32x4 gr = {__float_as_uint(sum[2 * r]), epoch, __float_as_uint(sum[2 * r + 1]), epoch};
                        if (ticket.same_xcd) __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, (unsigned)((d * G + slice) * CHUNK) * 8u + lane_off + r * 1024u, xsoff, 1);
                        else __builtin_amdgcn_raw_buffer_store_b128(gr, xrs, (unsigned)((d * G + slice) * CHUNK) * 8u + lane_off + r * 1024u, xsoff, 16);
                    }
                }
            }
            // dz of the step leaves for the weight-gradient products FROM THE LDS TILE, as 16-byte pieces: a wave instruction covers one
            // sequence's four gate runs of 64 units (4 x 256 contiguous bytes: whole cache lines) - 4 store instructions per step
            // instead of 16 dword stores from the registers (one dword per lane, 64-byte runs), behind the MFMAs where the
            // waves wait for barrier B anyway: config-2 step 1.2600 -> 1.2559 ms.  (It does NOT change the kernel's WRITE_SIZE,
            // 610 MB per step against a 252 MB tape in profiles/r05_pmcstep_train_f32.txt as in r04: the rest is the exchange -
            // three destinations' partials as 8-byte granules, 24 KB per workgroup and step, written through for visibility.)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int row = wave + 4 * k, gate = lane >> 4, c4 = lane & 15;
                const f32x4 zq = *(const f32x4*)(sZ + row * LDZ + gate * 64 + 4 * c4);
                if (b0 + row < p.B) *(f32x4*)(p.dz + ((size_t)(b0 + row) * p.T + t) * 4 * H + gate * H + slice * 64 + 4 * c4) = zq;
            }
            __syncthreads();   // barrier B: every wave is done with the dz tile
            if (G > 1) {
                unsigned spins = 0;
                while (true) {
                    bool ok = true;
#pragma unroll
                    for (int j = 0; j < (G - 1) * 2; ++j) ok = ok && (v[j].y == epoch) && (v[j].w == epoch);
                    if (__all(ok)) break;
                    ++spins;
                    if (spins > BSPIN_LIMIT || ((spins & 63u) == 0 && xch_poisoned(p.status))) {
                        if (lane == 0) {
                            xch_give_up(p.status);
                            sFlag[0] = 1;
                        }
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                    asm volatile("" ::: "memory");
#pragma unroll
                    for (int k = 0; k < G - 1; ++k) {
                        const int s_src = (slice + 1 + k) & (G - 1);
#pragma unroll
                        for (int r = 0; r < 2; ++r)
                            v[k * 2 + r] = __builtin_amdgcn_raw_buffer_load_b128(
                                xrs, (unsigned)((slice * G + s_src) * CHUNK) * 8u + lane_off + r * 1024u, xsoff, 16);
                    }
                }
            }
            // dh_{t-1}: partials summed in slice order 0..G-1 (own slice in its place)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float acc = 0.f;
#pragma unroll
                for (int s_abs = 0; s_abs < G; ++s_abs) {
                    // k such that (slice + 1 + k) mod G == s_abs, or the local partial when s_abs == slice
                    float term = part[r];
                    if (G > 1) {
#pragma unroll
                        for (int k = 0; k < G - 1; ++k)
                            if (((slice + 1 + k) & (G - 1)) == s_abs) term = __uint_as_float((r & 1) ? v[k * 2 + (r >> 1)].z : v[k * 2 + (r >> 1)].x);
                    }
                    acc += term;
                }
                dh[r] = acc;
            }
        }
        if (!aborted && p.db_part) {
            // the four g4 lane groups hold the same unit: fold them (fixed order), lanes with g4 == 0 store
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = dbacc[g];
                const float v1 = __shfl(v0, n + 16), v2 = __shfl(v0, n + 32), v3 = __shfl(v0, n + 48);
                if (g4 == 0) p.db_part[(size_t)tile * 4 * H + g * H + unit] = (v0 + v1) + (v2 + v3);
            }
        }
        if (!aborted) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = b0 + 4 * g4 + r;
                if (row < p.B) {
                    if (p.dh0) p.dh0[(size_t)row * H + unit] = dh[r];
                    if (p.dc0) p.dc0[(size_t)row * H + unit] = dc[r];
                }
            }
        }
        __syncthreads();
    }
    if (G > 1) xch_settle(p.status, ticket, (unsigned)p.epoch_span);
}

// --------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------
bool bwd_cluster_shape_ok(int H) { return H == 64 || H == 128 || H == 256; }

size_t bwd_cluster_xch_bytes(int B, int H) {
    const size_t G = H / 64;
    const size_t groups = (size_t)cluster_num_groups(B, H);
    const size_t b = groups * 2 * G * G * (4 * 4 * 64) * sizeof(unsigned long long);
    return (b + 255) & ~(size_t)255;
}

template <int H>
static int launch_bwd_h(const BwdParams& p, int act, hipStream_t stream) {
    void (*kern)(BwdParams) = act == FOV_ACT_HARD_SIGMOID ? lstm_bwd_cluster_kernel<H, FOV_ACT_HARD_SIGMOID>
                                                          : lstm_bwd_cluster_kernel<H, FOV_ACT_SIGMOID>;
    const dim3 grid((H > 64 ? xch_padded_groups(p.num_groups) : p.num_groups) * (H / 64)), block(256);
    hipLaunchKernelGGL(kern, grid, block, 0, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("bwd cluster launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

// status word + granule buffers live at `xch_ws` (kStatusBytes + bwd_cluster_xch_bytes)
int launch_bwd_cluster(const float* R, const float* reserve, const float* c0, const float* dhs, const float* dhT,
                       const float* dcT, float* dz, float* dh0, float* dc0, float* db_part, int B, int T, int H, int act,
                       void* xch_ws, hipStream_t stream) {
    if (B == 0 || T == 0) return FOV_OK;
    BwdParams p = {};
    p.R = R; p.reserve = reserve; p.c0 = c0; p.dhs = dhs; p.dhT = dhT; p.dcT = dcT; p.dz = dz; p.dh0 = dh0; p.dc0 = dc0; p.db_part = db_part;
    p.B = B; p.T = T; p.H = H;
    p.num_tiles = (B + BBT - 1) / BBT;
    p.num_groups = cluster_num_groups(B, H);
    p.status = (unsigned*)xch_ws;
    p.xch = (unsigned long long*)((char*)xch_ws + kStatusBytes);
    if (bwd_cluster_xch_bytes(B, H) > kXchBytes - kHelloBytes) { set_error("BPTT kernel: granule area exceeds the workspace's"); return FOV_ERR_WORKSPACE; }
    p.epoch_span = T * ((p.num_tiles + p.num_groups - 1) / p.num_groups) + 1;   // no memset: tags continue from the header
    if (int rc_ = xch_account(p.status, p.epoch_span, stream)) return rc_;
    switch (H) {
        case 64: return launch_bwd_h<64>(p, act, stream);
        case 128: return launch_bwd_h<128>(p, act, stream);
        default: return launch_bwd_h<256>(p, act, stream);
    }
}

}  // namespace fov
