// Weight gradient of a 'same' k x k convolution, ALL TAPS PER WORKGROUP (round 5): the ConvLSTM cells' and the small head layers'
// weight gradients of mycode/convlstm_seq2seq.py:100-126,146-165,209-287 under model.fit.
//
// conv_wgrad_kernel (conv_train_kernels.hip) is a TN GEMM per filter tap: blockIdx.y = (channel tile, tap), so x and dz are read
// once PER TAP - 25 times at k = 5.  For the head's 512 -> 1024 layer that is fine (0.78 of the matrix peak: 128 x 128 tiles carry
// 32 FLOP per byte).  For the cells (8..32 channels -> 32..128) and the head's 56 -> 512 and 1024 -> 30 layers it is not: one
// training step spent 176 ms on 7 TFLOP (0.25 of peak), each launch bound by 25 passes over its operands (32 -> 128: 21 GB).
//
// Here a workgroup owns a (16 MI channels) x (16 NI outputs) tile of EVERY tap and walks the maps LINE by line (a line = an image
// row, or an image column when that length is the multiple of 4 - 36 x 18 maps: columns of 36).  Per line it needs the dz line
// (L pixels x 16 NI) and the k lines of x around it, each with k/2 zero pixels at both ends, in LDS: a ring of k + 1 line slots, ONE
// new x line per step - x and dz of a channel / output slice are read once.  Tap (u, v) = (across lines, along the line) multiplies
// ring line s + u, LDS pixels p + v, with the dz line: the shift along the line is an LDS row offset, the shift across lines a ring
// slot; zero padding is the zero ends / the zero lines between two maps (one stream of lines: h zero lines, a map, h zero lines,
// the next map, ...).  Operands: A = x [pixel][channel], B = dz [pixel][output], both k-slow as they lie; LDS pixel strides of 16 or
// 48 (mod 64) floats put the four pixel rows of an MFMA step on disjoint banks.  Waves split the (tap, tile) products: tiles if
// there are four or more, taps otherwise.  Split over the maps (blockIdx.z), partial slices, fixed-order reduce (deterministic).
#include "fov_common.h"

namespace fov {

namespace {

typedef unsigned lu32x4 __attribute__((ext_vector_type(4)));

struct WlineArgs {
    const float* x;
    const float* dz;
    float* out;          // [split][ks*ks*C*N]
    long ldx;            // pixel stride of x (floats)
    long map_px;         // pixels per map (H * W)
    int C, N;
    int L, NL;           // pixels per line, lines per map
    int sp, sl;          // pixel index = line * sl + p * sp
    int colmode;         // lines are image columns: tap (u across, v along) is filter element (v, u)
    int maps, maps_per_split;
    int avec, bvec;
};

constexpr unsigned L_OOR = 0x80000000u;

// LDS pixel stride (floats) of a line image `width` floats wide: >= width and = 16 or 48 (mod 64) - the four pixel rows of an MFMA
// step then sit on disjoint banks
constexpr int lds_stride(int width) { return width <= 16 ? 16 : width <= 48 ? 48 : width <= 80 ? 80 : 112; }

// MI, NI: MFMA tiles of the workgroup's tile; KS: filter size; AV / BV: 16-byte staging loads of x / dz
template <int MI, int NI, int KS, int AV, int BV>
__global__ __launch_bounds__(256) void conv_wgrad_lines_kernel(WlineArgs g) {
    constexpr int TT = MI * NI, NT = KS * KS, HALO = KS / 2, RS = KS + 1;
    constexpr int XS = lds_stride(16 * MI), BS = lds_stride(16 * NI);      // LDS pixel strides of the x ring and the dz lines
    // a wave's share: TT >= 4: tiles t = wave, wave + 4, ... of every tap; TT < 4: tile wave % TT of the taps wave / TT, + 4 / TT, ...
    constexpr int TPW = TT >= 4 ? TT / 4 : 1;                   // tiles per wave
    constexpr int TSTEP = TT >= 4 ? 1 : 4 / TT;                 // tap stride of a wave
    constexpr int NA = (NT + TSTEP - 1) / TSTEP;                // taps per wave (upper bound)
    static_assert(TT == 1 || TT == 2 || TT % 4 == 0, "tiles per tap: 1, 2 or a multiple of 4");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 15, lq = lane >> 4;
    const int L4 = (g.L + 3) & ~3, LP = L4 + KS - 1;           // line length padded to MFMA steps; with the zero ends
    float* const xring = lds;                                   // [RS][LP][xs]
    float* const dzl = lds + (size_t)RS * LP * XS;            // [2][L4][bs]
    const int c0 = blockIdx.y * 16 * MI, n0 = blockIdx.x * 16 * NI;
    const int m_beg = blockIdx.z * g.maps_per_split;
    const int m_end = min(g.maps, m_beg + g.maps_per_split);
    // zero the whole LDS image once: zero ends, padding channels / outputs, padding pixels never change afterwards
    {
        const int total = RS * LP * XS + 2 * L4 * BS;
        for (int e = tid; e < total; e += 256) lds[e] = 0.f;
    }
    f32x4 acc[NA][TPW];
#pragma unroll
    for (int a = 0; a < NA; ++a)
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[a][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const int tap0 = TT >= 4 ? 0 : wave / TT;
    int tile_i[TPW], tile_j[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const int tl = TT >= 4 ? wave + 4 * t : wave % TT;
        tile_i[t] = tl / NI; tile_j[t] = tl - tile_i[t] * NI;
    }
    // ---- staging maps: thread -> (pixel, 4 channels) of an x line / (pixel, 4 outputs) of a dz line ----
    constexpr int RA = MI, RB = NI;                             // 16-byte pieces per thread (L <= 64)
    int a_p[RA], a_c[RA], b_p[RB], b_n[RB];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int e = tid + 256 * r;
        a_p[r] = e / (4 * MI); a_c[r] = 4 * (e - a_p[r] * (4 * MI));
    }
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int e = tid + 256 * r;
        b_p[r] = e / (4 * NI); b_n[r] = 4 * (e - b_p[r] * (4 * NI));
    }
    f32x4 va[RA], vb[RB];
    // stream position s of this workgroup's maps: position = (m - m_beg) * (NL + HALO) + line, line in [0, NL + HALO); lines >= NL are
    // the zero lines between two maps; the stream starts with HALO zero lines (the ring's initial zeros)
    const int per_map = g.NL + HALO;
    const long total_pos = (long)(m_end - m_beg) * per_map;
    auto fetch_x = [&](long pos) {      // x line at stream position pos -> va (zeros for a zero line / past the end)
        const int mi = (int)(pos / per_map);
        const int line = (int)(pos - (long)mi * per_map);
        const bool real = pos < total_pos && line < g.NL;
        const float* base = g.x + ((long)(m_beg + mi) * g.map_px + (long)line * g.sl) * g.ldx + c0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(real ? base : g.x), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < RA; ++r) {
            const bool ok = real && a_p[r] < g.L;
            const unsigned off = (unsigned)(((long)a_p[r] * g.sp * g.ldx + a_c[r]) * 4);
            if constexpr (AV) {
                const lu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, ok && c0 + a_c[r] < g.C ? off : L_OOR, 0, 0);
                va[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
#pragma unroll
                for (int k = 1; k < 4; ++k)
                    if (c0 + a_c[r] + k >= g.C) va[r][k] = 0.f;      // (a 16-byte piece that straddles C: the pixel stride holds it)
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    va[r][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ok && c0 + a_c[r] + k < g.C ? off + 4u * k : L_OOR, 0, 0));
            }
        }
    };
    auto fetch_dz = [&](long pos) {
        const int mi = (int)(pos / per_map);
        const int line = (int)(pos - (long)mi * per_map);
        const bool real = pos < total_pos && line < g.NL;
        const float* base = g.dz + ((long)(m_beg + mi) * g.map_px + (long)line * g.sl) * g.N + n0;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(real ? base : g.dz), 0, 0x7fffffff, 0x00020000);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const bool ok = real && b_p[r] < g.L;
            const unsigned off = (unsigned)(((long)b_p[r] * g.sp * g.N + b_n[r]) * 4);
            if constexpr (BV) {
                const lu32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, ok && n0 + b_n[r] < g.N ? off : L_OOR, 0, 0);
                vb[r] = (f32x4){__uint_as_float(t[0]), __uint_as_float(t[1]), __uint_as_float(t[2]), __uint_as_float(t[3])};
            } else {
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    vb[r][k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, ok && n0 + b_n[r] + k < g.N ? off + 4u * k : L_OOR, 0, 0));
            }
        }
    };
    auto stash_x = [&](int slot) {
        float* dst = xring + (size_t)slot * LP * XS;
#pragma unroll
        for (int r = 0; r < RA; ++r)
            if (a_p[r] < L4) *(f32x4*)(dst + (size_t)(a_p[r] + HALO) * XS + a_c[r]) = va[r];
    };
    auto stash_dz = [&](int buf) {
        float* dst = dzl + (size_t)buf * L4 * BS;
#pragma unroll
        for (int r = 0; r < RB; ++r)
            if (b_p[r] < L4) *(f32x4*)(dst + (size_t)b_p[r] * BS + b_n[r]) = vb[r];
    };
    // ring slot of stream position p: (p + HALO) % RS  (positions -HALO .. -1 are the initial zero lines)
    __syncthreads();
    // prologue: x lines 0 .. HALO and dz line 0
    for (int p = 0; p <= HALO; ++p) {
        fetch_x(p);
        stash_x((p + HALO) % RS);
    }
    fetch_dz(0);
    stash_dz(0);
    __syncthreads();
    const int ksteps = L4 >> 2;
    for (long s = 0; s < total_pos; ++s) {
        // the next x line and the next dz line travel under this line's MFMAs
        fetch_x(s + HALO + 1);
        fetch_dz(s + 1);
        const int mi = (int)(s / per_map);
        const int line = (int)(s - (long)mi * per_map);
        if (line < g.NL) {
            const float* bline = dzl + (size_t)(s & 1) * L4 * BS + (size_t)lq * BS + li;
            const int sbase = (int)(s % RS);
            auto slot_off = [&](int u) { const int q = sbase + u; return (q >= RS ? q - RS : q) * LP * XS; };      // ring line s + u - HALO
#pragma unroll 1
            for (int ks = 0; ks < ksteps; ++ks) {
                float bf[TPW];
#pragma unroll
                for (int t = 0; t < TPW; ++t) bf[t] = bline[(size_t)(4 * ks) * BS + 16 * tile_j[t]];
                const float* apix = xring + (size_t)(4 * ks + lq) * XS + li;
#pragma unroll
                for (int a = 0; a < NA; ++a) {
                    const int tap = tap0 + a * TSTEP;       // (u across lines, v along the line)
                    if (TSTEP > 1 && tap >= NT) break;
                    const int u = tap / KS, v = tap - u * KS;
                    const float* ap = apix + slot_off(u) + v * XS;
#pragma unroll
                    for (int t = 0; t < TPW; ++t) {
                        const float af = ap[16 * tile_i[t]];
                        acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af, bf[t], acc[a][t], 0, 0, 0);
                    }
                }
            }
        }
        stash_x((int)((s + KS) % RS));      // line s + HALO + 1: the slot line s - HALO - 1 has left
        stash_dz((int)((s + 1) & 1));
        __syncthreads();
    }
    // ---- partial slice: out[z][filter element][c][n] ----
    float* out = g.out + (size_t)blockIdx.z * NT * g.C * g.N;
#pragma unroll
    for (int a = 0; a < NA; ++a) {
        const int tap = tap0 + a * TSTEP;
        if (tap >= NT) break;
        const int u = tap / KS, v = tap - u * KS;
        const int felem = g.colmode ? v * KS + u : u * KS + v;      // (row, column) of the filter
#pragma unroll
        for (int t = 0; t < TPW; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int c = c0 + 16 * tile_i[t] + 4 * lq + r, n = n0 + 16 * tile_j[t] + li;
                if (c < g.C && n < g.N) out[((size_t)felem * g.C + c) * g.N + n] = acc[a][t][r];
            }
    }
}

}  // namespace

// Shapes the line kernel takes: square odd filters 3 / 5, no dilation, lines of at most 64 pixels, and an operand regime where one
// pass per tap loses (narrow layers; the wide 512 -> 1024 head layer stays on the tap-wise GEMM)
bool conv_wgrad_lines_takes(int H, int W, int C, int N, int kh, int kw, int dil) {
    if (env_knobs().no_wgrad_lines || dil != 1 || kh != kw || (kh != 3 && kh != 5) || H > 64 || W > 64 || H < 1 || W < 1) return false;
    return C <= 96 || N <= 32;
}

int conv_wgrad_lines(const float* x, long ldx, const float* dz, float* dw, int B, int H, int W, int C, int N, int ks, int accumulate,
                     float* scratch, size_t scratch_floats, hipStream_t stream) {
    const size_t wn = (size_t)ks * ks * C * N;
    WlineArgs g = {};
    g.x = x; g.dz = dz; g.ldx = ldx; g.map_px = (long)H * W; g.C = C; g.N = N; g.maps = B;
    // lines: rows unless the column length is the multiple of 4 and the row length is not (36 x 18 maps: columns of 36)
    g.colmode = (W & 3) != 0 && (H & 3) == 0;
    if (g.colmode) { g.L = H; g.NL = W; g.sp = W; g.sl = 1; }
    else { g.L = W; g.NL = H; g.sp = 1; g.sl = W; }
    int MI, NI;
    if (C <= 16) { MI = 1; NI = 2; }
    else if (C <= 32) { MI = 2; NI = 2; }
    else { MI = 4; NI = 2; }
    if (N <= 16) NI = 1;
    if (MI == 4 && NI == 1) { MI = 4; NI = 1; }
    const int ctiles = (C + 16 * MI - 1) / (16 * MI), gn = (N + 16 * NI - 1) / (16 * NI);
    const int xs = lds_stride(16 * MI), bs = lds_stride(16 * NI);
    const int L4 = (g.L + 3) & ~3;
    const size_t lds_bytes = sizeof(float) * ((size_t)(ks + 1) * (L4 + ks - 1) * xs + (size_t)2 * L4 * bs);
    if (lds_bytes > 160 * 1024) { set_error("conv_wgrad_lines: LDS image too large"); return FOV_ERR_UNSUPPORTED; }
    if ((long)B * H * W * (ldx > N ? ldx : N) * 4 >= (1L << 40)) { set_error("conv_wgrad_lines: operand too large"); return FOV_ERR_UNSUPPORTED; }
    // split over the maps until the chip is full (about four workgroups per CU), within the scratch the caller gave
    const long tiles = (long)ctiles * gn;
    long split = (1024 + tiles - 1) / tiles;
    if (split > B) split = B;
    if (split > 512) split = 512;
    while (split > 1 && (size_t)split * wn > scratch_floats) --split;
    if (split < 1) split = 1;
    g.maps_per_split = (int)((B + split - 1) / split);
    split = (B + g.maps_per_split - 1) / g.maps_per_split;
    const bool via_scratch = split > 1 || accumulate;
    if (via_scratch && (size_t)split * wn > scratch_floats) { set_error("conv_wgrad_lines: scratch too small"); return FOV_ERR_WORKSPACE; }
    g.out = via_scratch ? scratch : dw;
    // 16-byte staging loads: x when the pixel stride holds whole pieces, dz when N is a multiple of 4
    g.avec = (ldx & 3) == 0 && (((uintptr_t)x) & 15) == 0 && ((C + 3) & ~3) <= ldx;
    g.bvec = (N & 3) == 0 && (((uintptr_t)dz) & 15) == 0;
    const dim3 grid((unsigned)gn, (unsigned)ctiles, (unsigned)split);
#define FOV_WLINES(MI_, NI_, KS_)                                                                                                     \
    do {                                                                                                                              \
        auto k11 = conv_wgrad_lines_kernel<MI_, NI_, KS_, 1, 1>; auto k10 = conv_wgrad_lines_kernel<MI_, NI_, KS_, 1, 0>;             \
        auto k01 = conv_wgrad_lines_kernel<MI_, NI_, KS_, 0, 1>; auto k00 = conv_wgrad_lines_kernel<MI_, NI_, KS_, 0, 0>;             \
        auto kern = g.avec ? (g.bvec ? k11 : k10) : (g.bvec ? k01 : k00);                                                             \
        if (lds_bytes > 64 * 1024) (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds_bytes, stream, g);                                                              \
    } while (0)
    if (ks == 5) {
        if (MI == 1 && NI == 1) FOV_WLINES(1, 1, 5);
        else if (MI == 1) FOV_WLINES(1, 2, 5);
        else if (MI == 2 && NI == 1) FOV_WLINES(2, 1, 5);
        else if (MI == 2) FOV_WLINES(2, 2, 5);
        else if (NI == 1) FOV_WLINES(4, 1, 5);
        else FOV_WLINES(4, 2, 5);
    } else {
        if (MI == 1 && NI == 1) FOV_WLINES(1, 1, 3);
        else if (MI == 1) FOV_WLINES(1, 2, 3);
        else if (MI == 2 && NI == 1) FOV_WLINES(2, 1, 3);
        else if (MI == 2) FOV_WLINES(2, 2, 3);
        else if (NI == 1) FOV_WLINES(4, 1, 3);
        else FOV_WLINES(4, 2, 3);
    }
#undef FOV_WLINES
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { set_error("conv_wgrad_lines launch: %s", hipGetErrorString(e)); return FOV_ERR_LAUNCH; }
    if (!via_scratch) return FOV_OK;
    return splitk_reduce(scratch, dw, (long)wn, (int)split, accumulate, stream);
}

}  // namespace fov
