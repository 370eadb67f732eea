// ConvLSTM2D step with the input PATCH resident in LDS (round 3; a8: mycode/convlstm_seq2seq.py:100-126,146-165,209-220).
//
// conv_kernels.hip runs the cell as an implicit GEMM whose k-tiles (16 channels of one filter tap) are gathered from
// global memory one by one: 25 taps x 4 channel blocks = 100 tiles for the first layer, each with its own global gather,
// LDS round trip and workgroup barrier in front of 64 MFMAs per wave - 55 % of the fp32 MFMA peak.  A 5x5 'same'
// convolution reads every input pixel 25 times, though: here a workgroup owns ROWS image rows of one map (6 x 18 = 108
// pixels = 7 MFMA row tiles), loads the (ROWS + kh - 1) x (W + kw - 1) halo patch of the channel concatenation
// [x_t | h_{t-1}] into LDS ONCE, and every tap is the same patch read at a shifted address:
//   * A operand: one ds_read_b128 per (row tile, 16-channel block) = four MFMA k-steps (k-slot of lane group lq in step s
//     is channel 4 lq + s); the tap shift is one add per row tile and tap, the channel block an immediate offset.  Pixel
//     stride CP + 8 floats (== 8 mod 16): the four 16-lane groups of a ds_read_b128 fall on 16 distinct 16-byte slots.
//   * B operand (weights, (kh, kw, C + F, 4F) as Keras stores them, L2-resident): straight from global memory into
//     registers, one tap ahead - each register is reloaded right behind the MFMAs that read it, so a load has most of a
//     tap (7 k cycles) to arrive.  No LDS, no barrier: after the patch is staged the workgroup never synchronises again.
//   * wave = 8 units x 4 gates (F = 32: two 16-column tiles, a lane holds (i, g) or (f, o) and takes the other pair from the
//     lane 8 places away) or 4 units x 4 gates (F = 16 / 8: one tile, gates meet by three in-row reads) for ALL 7 row tiles
//     (F = 8: two waves share the units, row tiles split 4 + 3).
//   * epilogue as in the implicit-GEMM cell: bias, gates, c_{t-1} -> c_t, h_t into its slice of the concatenated map,
//     optional activated-gates tape; z never exists in memory.
// Two workgroups per CU (patch 63 KB at 64 channels): one stages / stores while the other multiplies.
// The accumulation order per output (taps outer, 16-channel blocks, k-steps) equals conv2d_igemm_kernel's whenever C is
// a multiple of 16, so the two forms agree bit for bit there (tests/test_gpu_convlstm.py).
#include "fov_common.h"

namespace fov {

namespace {

struct CellPatchArgs {
    const float* x;        // (B,H,W,*) pixel stride ldx, batch stride ldb, C channels
    const float* h_prev;   // (B,H,W,*) or NULL (zero state: the weights then hold K alone)
    const float* w;        // (kh*kw*(C + C2), 4F)
    const float* bias;     // (4F) or NULL
    const float* c_prev;   // (B*H*W, F) or NULL
    float* c_new;
    float* h;              // pixel stride ldh
    float* gates;          // (B*H*W, 4F) or NULL
    long ldx, ldb, ldx2, ldb2, ldh;
    int B, H, W, C, C2, F, kh, kw;
    int rows;              // image rows per workgroup
    int groups;            // workgroups per map = ceil(H / rows)
};

typedef unsigned pu32x4 __attribute__((ext_vector_type(4)));

constexpr int PMT = 7;   // MFMA row tiles (16 pixels) per workgroup: rows * W <= 112

template <int UW, int WAVES_N, int CB, int ACT>
__global__ __launch_bounds__(256, 2) void convlstm_cell_patch_kernel(CellPatchArgs g) {
    constexpr int NT = UW / 4;                          // 16-column tiles per wave (4 gates x UW units)
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int MTW = (PMT + WAVES_M - 1) / WAVES_M;  // row tiles per wave
    constexpr int CP = 16 * CB, CPS = CP + 8;
    constexpr unsigned OOR = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int b = blockIdx.x / g.groups, rg = blockIdx.x - b * g.groups;
    const int y0 = rg * g.rows;
    const int rows_here = g.H - y0 < g.rows ? g.H - y0 : g.rows;
    const int npix = rows_here * g.W;
    const int PW = g.W + g.kw - 1, PH = g.rows + g.kh - 1;
    const int ph = (g.kh - 1) / 2, pw = (g.kw - 1) / 2;
    const int F = g.F, N = 4 * F, Ctot = g.C + g.C2;

    // ---- stage the halo patch: [x | h_prev | zero channels up to CP] per patch pixel; outside the image: zeros ----
    {
        const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x + (long)b * g.ldb), 0, 0x7fffffff, 0x00020000);
        const __amdgpu_buffer_rsrc_t hrs = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(g.h_prev ? g.h_prev + (long)b * g.ldb2 : nullptr), 0, g.h_prev ? 0x7fffffff : 0, 0x00020000);
        const int q1 = g.C >> 2, q2 = g.C2 >> 2, qz = (CP - Ctot) >> 2;
        const int npp = PH * PW;
        auto stage = [&](const __amdgpu_buffer_rsrc_t& rs, int nq, long ld, int ch0, bool zero) {
            const int total = npp * nq;
#pragma unroll 4
            for (int e = tid; e < total; e += 256) {
                const int pp = e / nq, qd = e - pp * nq;
                const int py = pp / PW, px = pp - py * PW;
                const int iy = y0 - ph + py, ix = px - pw;
                const bool ok = !zero && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
                const unsigned off = ok ? (unsigned)((((long)iy * g.W + ix) * ld + 4 * qd) * 4) : OOR;
                const pu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                *(pu32x4*)&patch[pp * CPS + ch0 + 4 * qd] = t;
            }
        };
        stage(xrs, q1, g.ldx, 0, false);
        if (q2 > 0) stage(hrs, q2, g.ldx2, g.C, false);
        if (qz > 0) stage(hrs, qz, 0, Ctot, true);
    }
    __syncthreads();

    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int unit0 = wn * UW;
    // A: LDS float index of the window origin of this lane's pixel in each row tile (+ the lane group's channel quad)
    int abase[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) {
        int p = 16 * (wm * MTW + i) + li;
        p = p < npix ? p : npix - 1;   // pixels past the block: a valid address, results dropped
        const int y = p / g.W, x = p - y * g.W;
        abase[i] = (y * PW + x) * CPS + 4 * lq;
    }
    // B: byte offset of (row 4 lq, this lane's gate column) inside a 16-row weight block; the last block of a tap may be short
    unsigned bvoff[NT], bvoff_last[NT];
    const int rem_last = Ctot - 16 * (CB - 1);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = 16 * j + li;
        const int col = (q / UW) * F + unit0 + (q % UW);
        bvoff[j] = (unsigned)(((4 * lq) * N + col) * 4);
        bvoff_last[j] = 4 * lq < rem_last ? bvoff[j] : OOR;
    }
    const int ntaps = g.kh * g.kw;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w), 0, ntaps * Ctot * N * 4, 0x00020000);
    float bw[CB][NT][4];
    auto load_b = [&](int tap, int cb) {
        const int row0 = tap * Ctot + 16 * cb;
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                bw[cb][j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, cb == CB - 1 ? bvoff_last[j] : bvoff[j],
                                                                                     (unsigned)((row0 + s) * N * 4), 0));
    };
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) load_b(0, cb);

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    f32x4 af[MTW];
#pragma unroll
    for (int i = 0; i < MTW; ++i) af[i] = *(const f32x4*)&patch[abase[i]];
    int dy = 0, dx = 0;
    for (int tap = 0; tap < ntaps; ++tap) {
        // position of the NEXT tap (the last one re-reads itself: no branch around loads)
        int ndx = dx + 1, ndy = dy;
        if (ndx == g.kw) { ndx = 0; ++ndy; }
        const bool last = tap + 1 == ntaps;
        const int tap_n = last ? tap : tap + 1;
        const int noff = last ? (dy * PW + dx) * CPS : (ndy * PW + ndx) * CPS;
        const int coff = (dy * PW + dx) * CPS;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            f32x4 an[MTW];
#pragma unroll
            for (int i = 0; i < MTW; ++i)
                an[i] = cb + 1 < CB ? *(const f32x4*)&patch[abase[i] + coff + 16 * (cb + 1)] : *(const f32x4*)&patch[abase[i] + noff];
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int i = 0; i < MTW; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i][s], bw[cb][j][s], acc[i][j], 0, 0, 0);
            load_b(tap_n, cb);   // the registers just read: the next tap's weights have a whole tap to arrive
#pragma unroll
            for (int i = 0; i < MTW; ++i) af[i] = an[i];
        }
        dx = ndx; dy = ndy;
    }

    // ---- epilogue: gates and cell update, lane-local after the gates of a unit have met ----
    const int unit = unit0 + (li % UW);
    const bool mine = li < UW;
    float bz[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int q = 16 * j + li;
        bz[j] = g.bias ? g.bias[(q / UW) * F + unit0 + (q % UW)] : 0.f;
    }
    const long mbase = ((long)b * g.H + y0) * g.W;
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int p = 16 * (wm * MTW + i) + 4 * lq + r;
            float zi, zf, zg, zo;
            if constexpr (UW == 8) {
                const float z0 = acc[i][0][r] + bz[0], z1 = acc[i][1][r] + bz[1];   // (i, g) in lanes 0-7, (f, o) in lanes 8-15
                zi = z0; zg = z1;
                zf = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(z0), 0x128, 0xf, 0xf, false));
                zo = __uint_as_float(__builtin_amdgcn_update_dpp(0u, __float_as_uint(z1), 0x128, 0xf, 0xf, false));
            } else {
                const float z = acc[i][0][r] + bz[0];   // gate li / 4 of unit li % 4
                zi = z;
                zf = __shfl(z, (lane & ~15) + ((li + 4) & 15));
                zg = __shfl(z, (lane & ~15) + ((li + 8) & 15));
                zo = __shfl(z, (lane & ~15) + ((li + 12) & 15));
            }
            if (mine && p < npix && wm * MTW + i < PMT) {
                const long m = mbase + p;
                const float gi = rec_act<ACT>(zi), gf = rec_act<ACT>(zf), gg = tanh_f(zg), go = rec_act<ACT>(zo);
                const float cn = fmaf(gf, g.c_prev ? g.c_prev[m * F + unit] : 0.f, gi * gg);
                g.c_new[m * F + unit] = cn;
                g.h[m * g.ldh + unit] = go * tanh_f(cn);
                if (g.gates) {
                    float* gp = g.gates + m * N + unit;
                    gp[0] = gi; gp[F] = gf; gp[2 * F] = gg; gp[3 * F] = go;
                }
            }
        }
}

template <int UW, int WAVES_N, int CB>
int launch_patch_t(const CellPatchArgs& g, int act, size_t lds, hipStream_t stream) {
    void (*kern)(CellPatchArgs) = act == FOV_ACT_HARD_SIGMOID ? convlstm_cell_patch_kernel<UW, WAVES_N, CB, FOV_ACT_HARD_SIGMOID>
                                                              : convlstm_cell_patch_kernel<UW, WAVES_N, CB, FOV_ACT_SIGMOID>;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.B * g.groups)), dim3(256), lds, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("convlstm_cell_patch launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace

// Shapes the patch form takes: F in {8, 16, 32}, at most 64 input channels per tap (x and h_prev together), 16-byte
// aligned channel vectors, maps at most 112 pixels wide whose patch fits half of the LDS.
bool cell_patch_shape_ok(const float* x, long ldx, long ldb, int C, const float* h_prev, long ldx2, long ldb2, int F, int H, int W,
                         int kh, int kw) {
    if (env_knobs().no_cell_patch) return false;
    if (!(F == 8 || F == 16 || F == 32)) return false;
    const int C2 = h_prev ? F : 0, Ctot = C + C2;
    if (C <= 0 || (C & 3) || Ctot > 64) return false;
    if ((ldx & 3) || (ldb & 3) || (((uintptr_t)x) & 15)) return false;
    if (h_prev && ((ldx2 & 3) || (ldb2 & 3) || (((uintptr_t)h_prev) & 15))) return false;
    if (W < 1 || W > 16 * PMT || kh < 1 || kw < 1 || kh * kw > 64) return false;
    int rows = (16 * PMT) / W;
    if (rows > H) rows = H;
    const int CB = (Ctot + 15) / 16;
    const size_t lds = sizeof(float) * (size_t)(rows + kh - 1) * (W + kw - 1) * (16 * CB + 8);
    return lds <= 80 * 1024;
}

int launch_cell_patch(const float* x, long ldx, long ldb, int C, const float* h_prev, long ldx2, long ldb2, const float* w,
                      const float* bias, const float* c_prev, float* c_new, float* h, long ldh, float* gates, int B, int H, int W,
                      int F, int kh, int kw, int act, hipStream_t stream) {
    CellPatchArgs g = {};
    g.x = x; g.h_prev = h_prev; g.w = w; g.bias = bias; g.c_prev = c_prev; g.c_new = c_new; g.h = h; g.gates = gates;
    g.ldx = ldx; g.ldb = ldb; g.ldx2 = ldx2; g.ldb2 = ldb2; g.ldh = ldh;
    g.B = B; g.H = H; g.W = W; g.C = C; g.C2 = h_prev ? F : 0; g.F = F; g.kh = kh; g.kw = kw;
    g.rows = (16 * PMT) / W;
    if (g.rows > H) g.rows = H;
    g.groups = (H + g.rows - 1) / g.rows;
    const int CB = (g.C + g.C2 + 15) / 16;
    const size_t lds = sizeof(float) * (size_t)(g.rows + kh - 1) * (W + kw - 1) * (16 * CB + 8);
#define FOV_PATCH_CB(UW_, WN_)                                                     \
    switch (CB) {                                                                  \
        case 1: return launch_patch_t<UW_, WN_, 1>(g, act, lds, stream);           \
        case 2: return launch_patch_t<UW_, WN_, 2>(g, act, lds, stream);           \
        case 3: return launch_patch_t<UW_, WN_, 3>(g, act, lds, stream);           \
        default: return launch_patch_t<UW_, WN_, 4>(g, act, lds, stream);          \
    }
    if (F == 32) { FOV_PATCH_CB(8, 4) }
    if (F == 16) { FOV_PATCH_CB(4, 4) }
    FOV_PATCH_CB(4, 2)
#undef FOV_PATCH_CB
}

}  // namespace fov
