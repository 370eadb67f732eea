// 'same' Conv2D with the WHOLE input map resident in LDS, one 16-channel slab at a time (round 4; a9: the prediction head of
// mycode/convlstm_seq2seq.py:176-181,231-238 - Conv2D 56 -> 512 -> 1024 -> 30, k = 5, on 36 x 18 heat maps - and the data
// gradients of those layers in model.fit).
//
// conv2d_igemm_kernel gathers every k-tile (16 channels of one tap) from global memory: a 5 x 5 convolution reads each input
// value 25 times, the 512 -> 1024 layer - 70 % of the model's time - fetched 311 GB per predict call for 20 GB of
// activations, and every k-tile costs a global gather, an LDS round trip and a workgroup barrier in front of 64 MFMAs.
// Here a workgroup owns one MAP (all H x W pixels = up to 41 MFMA row tiles) and BN output channels:
//   * the halo patch (H + kh - 1) x (W + kw - 1) x 16 channels sits in LDS (84 KB at 36 x 18, pixel stride 24 floats: == 8
//     mod 16, so the four lane groups of a ds_read_b128 fall on sixteen distinct 16-byte slots); every tap of the slab is the
//     same patch read at a shifted address - 25 x 16 k-steps between two barriers instead of 16; the NEXT slab is fetched
//     into registers at the start of a slab and written behind the barrier that ends it (one LDS buffer);
//   * A operand: one ds_read_b128 per row tile and tap = four MFMA k-steps (k-slot of lane group lq in step s is channel
//     4 lq + s); row tile i + PER is row tile i moved down a whole number of image rows (16 PER pixels = RS rows; 36 x 18:
//     PER = 9, RS = 8), so a lane keeps PER base addresses and the rest are immediate offsets;
//   * B operand (weights (kh, kw, C, N) as Keras stores them): straight from global memory (L2-resident: a workgroup's
//     share is 25 x 16 x 32 floats per slab and wave), one (tap, slab) ahead, no LDS;
//   * a wave owns 32 output channels (two 16-column tiles) for MTW row tiles: WAVES_N = 2 -> 21 row tiles, BN = 64 (the wide
//     layers: 168 accumulation registers per lane, 168 MFMAs per tap and slab behind 21 LDS reads and 8 weight loads);
//     WAVES_N = 1 -> 11 row tiles, BN = 32 (the 30-channel output layer).
// Blocks are ordered n-block major: all resident workgroups work on the same BN columns, whose weights (3.3 MB at C = 512)
// stay in every XCD's L2; an input map is read once per n-block (16 times for the 512 -> 1024 layer instead of 25 x 8).
#include <stdlib.h>

#include "fov_common.h"

namespace fov {

namespace {

struct ConvPatchArgs {
    const float* x;        // (B,H,W,*) pixel stride ldx, batch stride ldb, C channels
    const float* w;        // (kh*kw*C, N)
    const float* bias;     // (N) or NULL
    const float* add;      // (B*H*W, N) or NULL
    float* y;              // (B*H*W, N)
    long ldx, ldb;
    int B, H, W, C, N, kh, kw, act;
    int per, rs;           // row tile i + per = row tile i moved down rs image rows
    int nblocks;           // ceil(N / BN)
};

typedef unsigned cpu32x4 __attribute__((ext_vector_type(4)));

constexpr int CPQ = 16;           // channels per slab
constexpr int CPS = CPQ + 8;      // pixel stride in LDS (floats)
constexpr int CP_MAXPER = 9;      // base addresses a lane keeps
constexpr int CP_SLACK_ROWS = 5;  // patch rows allocated below the halo: row tiles past the end of the map (41 tiles = 656 pixels for 648,
                                  // up to 44 tiles when the waves split the rows) read there; their results are never stored
constexpr int CP_STAGE = 14;      // 16-byte vectors of the next slab per thread: 14 * 256 >= (40 * 22) * 4

// TSTEP: floats between row tile i and row tile i + 9 as a compile-time constant (36 x 18 maps under a 5-wide kernel: 8 rows x 22
// patch pixels x 24 = 4224), so that a row tile's LDS address is base register + IMMEDIATE: a v_add per tile in front of its eight
// MFMAs costs ~14 cycles of the fp32 matrix pipe each (VALU and fp32 MFMA share one port).  0 = run-time step.
template <int WAVES_N, int TSTEP>
__global__ __launch_bounds__(256, 1) void conv2d_patch_kernel(ConvPatchArgs g) {
    constexpr int WAVES_M = 4 / WAVES_N;
    constexpr int MT = 41;                                  // row tiles of a map at most (656 pixels)
    constexpr int MTW = (MT + WAVES_M - 1) / WAVES_M;       // row tiles per wave: 41 | 21 | 11
    constexpr int NT = 2;                                   // 16-column tiles per wave
    constexpr int BN = 16 * NT * WAVES_N;
    constexpr unsigned OOR = 0x80000000u;
    extern __shared__ __attribute__((aligned(16))) float patch[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int nb = blockIdx.x / g.B, b = blockIdx.x - nb * g.B;      // n-block major: neighbours share the weights
    const int PW = g.W + g.kw - 1, PH = g.H + g.kh - 1;
    const int ph = (g.kh - 1) / 2, pw = (g.kw - 1) / 2;
    const int npix = g.H * g.W, npp = PH * PW;
    const int N = g.N;
    const int wn = wave % WAVES_N, wm = wave / WAVES_N;
    const int n0 = nb * BN + wn * 16 * NT;
    const int nslab = (g.C + CPQ - 1) / CPQ;
    const int ntaps = g.kh * g.kw;

    // ---- the slab loader: patch pixel pp, channel quad qd of the slab <- x[b][iy][ix][16 slab + 4 qd ..]; zeros outside the image
    // and beyond C.  Element e = tid + 256 v -> (pp = e / 4, qd = e % 4): offsets are computed ONCE (they do not depend on the slab).
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.x + (long)b * g.ldb), 0, 0x7fffffff, 0x00020000);
    // byte offset of the pixel's channel 4 qd, or OOR (outside the image / past the patch): kept in LDS behind the patch - fourteen
    // registers the 328-accumulator form does not have
    unsigned* soff = (unsigned*)(patch + (size_t)(PH + CP_SLACK_ROWS) * PW * CPS);
    const bool cvec = (g.C & 3) == 0;
#pragma unroll
    for (int v = 0; v < CP_STAGE; ++v) {
        const int e = tid + 256 * v, pp = e >> 2, qd = e & 3;
        const int py = pp / PW, px = pp - py * PW;
        const int iy = py - ph, ix = px - pw;
        const bool ok = pp < npp && iy >= 0 && iy < g.H && ix >= 0 && ix < g.W;
        soff[e] = ok ? (unsigned)((((long)iy * g.W + ix) * g.ldx + 4 * qd) * 4) : OOR;      // read back by the same thread only
    }
    cpu32x4 st[CP_STAGE];
    auto stage_load = [&](int slab) {
        const int c0 = slab * CPQ;
#pragma unroll
        for (int v = 0; v < CP_STAGE; ++v) {
            const int qd = (tid + 256 * v) & 3;
            const unsigned so = soff[tid + 256 * v];
            if (cvec) {
                const bool in = c0 + 4 * qd < g.C;       // C a multiple of four: a quad is inside or outside as a whole
                st[v] = __builtin_amdgcn_raw_buffer_load_b128(xrs, in ? so : OOR, (unsigned)(c0 * 4), 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const bool in = c0 + 4 * qd + q < g.C;
                    st[v][q] = __builtin_amdgcn_raw_buffer_load_b32(xrs, (in && so != OOR) ? so + 4 * q : OOR, (unsigned)(c0 * 4), 0);
                }
            }
        }
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int v = 0; v < CP_STAGE; ++v) {
            const int e = tid + 256 * v, pp = e >> 2, qd = e & 3;
            if (pp < npp) *(cpu32x4*)&patch[pp * CPS + 4 * qd] = st[v];
        }
    };

    // A: LDS float index of the window origin of this lane's pixel in row tiles 0 .. per-1 of the wave (+ the lane group's quad)
    int abase[CP_MAXPER];
    const int tile0 = wm * MTW;
#pragma unroll
    for (int i = 0; i < CP_MAXPER; ++i) {
        int p = 16 * (tile0 + i) + li;
        p = p < npix ? p : npix - 1;         // pixels past the map: a valid address, results dropped
        const int yy = p / g.W, xx = p - yy * g.W;
        abase[i] = (yy * PW + xx) * CPS + 4 * lq;
    }
    const int tile_step = TSTEP ? TSTEP : g.rs * PW * CPS;   // floats from row tile i to row tile i + per
    // the last row tile of the map may hang over its end (648 = 40.5 tiles): its lanes past the map re-read the last pixel
    // B: weights of (tap, slab): rows 16 slab + 4 lq + s of the tap, columns n0 + 16 j + li
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g.w), 0, ntaps * g.C * N * 4, 0x00020000);
    unsigned bcol[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) bcol[j] = (n0 + 16 * j + li < N) ? (unsigned)((n0 + 16 * j + li) * 4) : OOR;
    float bw[NT][4], bwn[NT][4];
    auto load_b = [&](int tap, int slab, float (&dst)[NT][4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int c = slab * CPQ + 4 * lq + s;
            const unsigned row = (unsigned)(((long)tap * g.C + c) * N * 4);
#pragma unroll
            for (int j = 0; j < NT; ++j)
                dst[j][s] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(wrs, (c < g.C) ? bcol[j] + row : OOR, 0, 0));
        }
    };

    f32x4 acc[MTW][NT];
#pragma unroll
    for (int i = 0; i < MTW; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // row tiles this wave really has: the map's 41 (ceil(npix / 16)) dealt MTW per wave - the last wave's share is shorter, and
    // its missing tiles are skipped as a whole (wave-uniform), not multiplied and dropped
    const int ntile_map = (npix + 15) >> 4;
    const int my_tiles = ntile_map - tile0 < MTW ? (ntile_map - tile0 > 0 ? ntile_map - tile0 : 0) : MTW;

    stage_load(0);
    load_b(0, 0, bw);
    stage_write();
    __syncthreads();
    for (int slab = 0; slab < nslab; ++slab) {
        const bool more = slab + 1 < nslab;
        if (more) stage_load(slab + 1);          // lands during this slab's 25 taps
        int dy = 0, dx = 0;
        for (int tap = 0; tap < ntaps; ++tap) {
            // weights of the next (tap, slab) pair: a whole tap (thousands of MFMA cycles) to arrive
            const bool last_tap = tap + 1 == ntaps;
            const int tap_n = last_tap ? 0 : tap + 1;
            const int slab_n = last_tap ? (more ? slab + 1 : slab) : slab;
            load_b(tap_n, slab_n, bwn);
            const int coff = (dy * PW + dx) * CPS;
            const float* ab[CP_MAXPER];         // the tap's shift goes into the nine base addresses once, not into every read
#pragma unroll
            for (int r = 0; r < CP_MAXPER; ++r) ab[r] = patch + abase[r] + coff;
            // Two row tiles ahead: the read a tile's MFMAs consume is issued two tiles (512 matrix cycles) earlier.  The scheduling
            // barriers keep it there - left alone, hipcc sinks every read to just in front of its first use (one register set for all
            // tiles) and the wave, alone on its SIMD, sits out a full LDS latency per tile (117 -> 121 TFLOP/s on the 512 -> 1024 layer).
            f32x4 a0 = *(const f32x4*)ab[0];
            f32x4 a1 = MTW > 1 ? *(const f32x4*)(ab[1 % CP_MAXPER] + (1 / CP_MAXPER) * tile_step) : a0;
#pragma unroll
            for (int i = 0; i < MTW; ++i) {
                f32x4 an = a0;
                if (i + 2 < MTW) an = *(const f32x4*)(ab[(i + 2) % CP_MAXPER] + ((i + 2) / CP_MAXPER) * tile_step);
                __builtin_amdgcn_sched_barrier(0);
                // a tile past the wave's share (wave-uniform: 41 tiles = 21 + 20 with two waves along the rows, 11 + 11 + 11 + 8 with
                // four) is skipped, not multiplied and dropped; only trailing tiles can be missing
                if (i < MTW - 3 || i < my_tiles) {
#pragma unroll
                    for (int s = 0; s < 4; ++s)
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[s], bw[j][s], acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a0 = a1;
                a1 = an;
            }
            // (tried: the two weight sets swapping roles tap by tap instead of these eight copies, one flat loop over (tap, slab)
            // pairs - 121 -> 105 TFLOP/s: the extra control flow spilled 32 scalar registers into the loop)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int s = 0; s < 4; ++s) bw[j][s] = bwn[j][s];
            if (++dx == g.kw) { dx = 0; ++dy; }
        }
        if (more) {
            __syncthreads();                     // every wave is done reading this slab
            stage_write();
            __syncthreads();
        }
    }

    // ---- epilogue: bias, residual, activation; D fragment: rows 4 lq + r of the tile, column li ----
    const long mbase = (long)b * npix;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int col = n0 + 16 * j + li;
        if (col >= N) continue;
        const float bz = g.bias ? g.bias[col] : 0.f;
#pragma unroll
        for (int i = 0; i < MTW; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int p = 16 * (tile0 + i) + 4 * lq + r;
                if (p < npix) {
                    const long o = (mbase + p) * N + col;
                    float v = acc[i][j][r] + bz;
                    if (g.add) v += g.add[o];
                    g.y[o] = g.act == 2 ? fmaxf(v, 0.f) : v;
                }
            }
    }
}

int gcd_i(int a, int b) { return b ? gcd_i(b, a % b) : a; }

}  // namespace

// Shapes the map-resident form takes: maps of at most 656 pixels whose halo patch (16-channel slab) fits the LDS, row tiles that
// repeat after nine (W / gcd(16, W) == 9: 36 x 18 and its transpose), at least 32 input channels (below that the tap-gathering kernel has as
// little to re-read), C * kh * kw rows of weights addressable in 31 bits.
bool conv_patch_shape_ok(const float* x, long ldx, long ldb, int B, int H, int W, int C, int N, int kh, int kw) {
    if (env_knobs().no_conv_patch) return false;
    if (H < 1 || W < 1 || H * W > 656 || H * W < 64 || C < 32 || N < 1 || B < 1) return false;
    if ((ldx & 3) || (ldb & 3) || (((uintptr_t)x) & 15)) return false;
    if (kh < 1 || kw < 1 || !(kh & 1) || !(kw & 1) || kh * kw > 49) return false;
    const int per = W / gcd_i(16, W);
    if (per != CP_MAXPER) return false;       // W = 9, 18, 36, 72: the kernel's row tiles repeat with period nine
    const size_t lds = sizeof(float) * ((size_t)(H + kh - 1 + CP_SLACK_ROWS) * (W + kw - 1) * CPS + CP_STAGE * 256);
    if (lds > 150 * 1024 || (size_t)(H + kh - 1) * (W + kw - 1) * 4 > (size_t)CP_STAGE * 256) return false;
    return true;
}

int launch_conv_patch(const float* x, long ldx, long ldb, const float* w, const float* bias, const float* add, float* y, int B, int H,
                      int W, int C, int N, int kh, int kw, int act, hipStream_t stream) {
    ConvPatchArgs g = {};
    g.x = x; g.w = w; g.bias = bias; g.add = add; g.y = y; g.ldx = ldx; g.ldb = ldb;
    g.B = B; g.H = H; g.W = W; g.C = C; g.N = N; g.kh = kh; g.kw = kw; g.act = act;
    g.per = W / gcd_i(16, W);
    g.rs = 16 * g.per / W;
    const size_t lds = sizeof(float) * ((size_t)(H + kh - 1 + CP_SLACK_ROWS) * (W + kw - 1) * CPS + CP_STAGE * 256);
    void (*kern)(ConvPatchArgs);
    int bn;
    const bool fixed = g.rs * (W + kw - 1) * CPS == 4224;      // the heat maps' geometry: immediate row-tile offsets
    // (WAVES_N = 4 - all 41 row tiles x 32 columns per wave, 128 columns per workgroup, 328 accumulation registers - compiles, but
    // hipcc then shuttles accumulators between the two register files around the MFMAs: 332 v_accvgpr moves per two taps, 118 -> 116
    // TFLOP/s with either width on the 512 -> 1024 layer; not instantiated)
    if (N > 32) { kern = fixed ? conv2d_patch_kernel<2, 4224> : conv2d_patch_kernel<2, 0>; bn = 64; }
    else { kern = fixed ? conv2d_patch_kernel<1, 4224> : conv2d_patch_kernel<1, 0>; bn = 32; }
    g.nblocks = (N + bn - 1) / bn;
    int rc = ensure_dynamic_lds((const void*)kern, lds);
    if (rc) return rc;
    hipLaunchKernelGGL(kern, dim3((unsigned)(g.nblocks * B)), dim3(256), lds, stream, g);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("conv2d_patch launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    return FOV_OK;
}

}  // namespace fov
