// Shared device/host helpers for libfov360_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fov360.h"

namespace fov {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// Gate activations.  v_exp_f32 / v_rcp_f32 based (about 1 ulp each); absolute error of the
// logistic and of tanh is below 3e-7, far inside the 1e-3 relative parity bound on outputs.
// __builtin_amdgcn_rcpf is the bare v_rcp_f32: __frcp_rn expands to the 10-instruction
// correctly-rounded division sequence, which tripled the cell-update time.
__device__ __forceinline__ float sigmoid_f(float x) {
    return __builtin_amdgcn_rcpf(1.0f + __expf(-x));
}
__device__ __forceinline__ float hard_sigmoid_f(float x) {
    return fminf(fmaxf(fmaf(0.2f, x, 0.5f), 0.0f), 1.0f);
}
__device__ __forceinline__ float tanh_f(float x) {
    // 1 - 2/(1+e^{2x}); saturates cleanly to +-1, no NaN for any finite x
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + __expf(2.0f * x));
}
template <int ACT>
__device__ __forceinline__ float rec_act(float x) {
    return ACT == FOV_ACT_HARD_SIGMOID ? hard_sigmoid_f(x) : sigmoid_f(x);
}

// Everything one launch of the LSTM kernels needs.  Phase 1 ("layer"/encoder): T steps over
// x:(B,T,F) with weights K,R,b from (h0,c0).  Phase 2 (decode only): T_out autoregressive
// steps with weights dK,dR,db fed by y_{t-1} = tanh(h W + bias), y_{-1} = dec_in0.
struct LstmParams {
    const float* x;
    const float* zx;           // optional (B,T,4H): precomputed x.K (stacked layers); then x/K are unused, b is still added
    const float* K;
    const float* R;
    const float* b;
    const float* h0;
    const float* c0;
    float* hs;
    float* hT;
    float* cT;
    float* reserve;            // training only: (B,T,5,H) = i,f,g,o,c of every step (phase 1), or NULL
    const float* dec_in0;
    const float* dK;
    const float* dR;
    const float* db;
    const float* dW;
    const float* dbias;
    float* out;
    int B, T, F, H;
    int T_out, F_dec;
    int act;
    // cluster kernel only
    unsigned long long* xch;   // granule exchange buffers
    unsigned* status;          // status[0] = timeout flag
    int num_groups;            // resident groups (persistent loop over 16-sequence tiles)
    int num_tiles;
    int epoch_span;            // epochs this launch may consume: the last workgroup to leave adds it to the header's base
    int force_safe_exchange;   // 1: never take the same-XCD fast path (tests)
    int xcd_pad;               // lstm_wide16.hip: grid padded to 8 x (workgroups per tile), block b = member b / 8 of group b % 8
    int trio;                  // lstm_wide16.hip, stacked launch on the XCD-per-group grid: 3 three roles, 2 two roles (0: not stacked) -
                               // layer 1 also publishes h_t (sc1) into a mirror ring for its readers on other XCDs
};

int launch_generic(const LstmParams& p, bool decode, hipStream_t stream);
long generic_launch_count();   // launches of the generic (VALU) LSTM kernel so far in this process (diagnostic)
bool wide16_s2s_shape(int B, int F_enc, int F_dec, int H);   // lstm_wide16.hip: encoder + free-running decoder, small batches
int launch_wide16_s2s(const LstmParams& p, hipStream_t stream);
bool pair_s2s_shape(int B, int T_in, int T_out, int F_enc, int F_dec, int H);   // lstm_pair.hip: H = 256, two tiles per group of eight workgroups
int launch_pair_s2s(const LstmParams& p, hipStream_t stream);
int launch_cluster(const LstmParams& p, bool decode, hipStream_t stream);
int launch_cluster_decoder(const LstmParams& p, hipStream_t stream);   // MODE_DECODE alone, from (p.h0, p.c0)
bool cluster_shape_ok(int F, int H);
size_t cluster_workspace_bytes(int B, int H);
int cluster_num_groups(int B, int H);
int device_cu_count();                               // CUs of the current device a persistent grid may count on (FOV_DBG_RESIDENT_LIMIT caps it: tests)
// getenv results cached at first use (fov_reload_env re-reads them): nothing on a launch path calls getenv
struct EnvKnobs {
    int force_safe_exchange, two_launches, resident_limit;
    int bwd_stepped, no_wgrad_fusion, no_dx_fusion, bwd_groups4, gemm_bf16_split, gemm_bf16_noremap, gemm_bf16_shallow, no_wgrad_group, dbg_trace, no_wide16_trio, no_wgrad_lines;   // experiment switches (tools/)
    int gemm_variant, gemm_split;   // FOV_GEMM_VARIANT / FOV_GEMM_SPLIT: tile shape / K slices of the fp32 GEMM forced (experiments)
    int pair;            // FOV_PAIR=1: the fused seq2seq call takes the tile-pair kernel (lstm_pair.hip) at 33 .. 64 tiles, H = 256 (an experiment that lost: off by default)
    int no_wide16;       // FOV_NO_WIDE16=1: width-512 layers stay on the 16-workgroup form (tests / A-B timing)
    int no_cell_patch;   // FOV_NO_CELL_PATCH=1: ConvLSTM2D steps stay on the implicit-GEMM cell (tests compare the two forms)
    int no_conv_patch;   // FOV_NO_CONV_PATCH=1: Conv2D layers stay on the tap-gathering implicit GEMM (tests / A-B timing; conv_patch.hip)
    int no_xcd_pad;      // FOV_NO_XCD_PAD=1: no padded grids for same-XCD placement (lstm_wide16 / lstm_bwd16 / fused H = 128 kernel)
    int xcd_pad_max;     // FOV_XCD_PAD_MAX: members per group up to which a grid is padded (default 16; 32 measured slower)
    int no_bwd16_narrow; // FOV_NO_BWD16_NARROW=1: widths 128 / 256 keep the 2- / 4- / 8-workgroup BPTT kernels at small batches
    int bwd16_groups32;  // FOV_BWD16_GROUPS=32: width-512 BPTT on thirty-two workgroups per tile up to eight tiles (default: sixteen)
    int no_stack2;       // FOV_NO_STACK2=1: two stacked width-512 layers as two launches (fov_lstm_stack2_supported -> 0)
};
const EnvKnobs& env_knobs();
void env_reload();
// Host-side epoch accounting of a workspace (xch_common.h): every exchange launch adds its span; long before the 32-bit
// tags could wrap, the header and the granule area are re-zeroed IN STREAM ORDER in front of the launch - whether or not
// the caller ever calls fov_check_status.
int xch_account(void* workspace, long span, hipStream_t stream);
void xch_forget(void* workspace);                        // the workspace was (re)initialised
void xch_note_force_safe(void* workspace, int on);
void xch_set_epoch_for_test(void* workspace, unsigned long long epoch);
int ensure_dynamic_lds(const void* kern, size_t lds, int block = 256);  // cached hipFuncSetAttribute(MaxDynamicSharedMemorySize) + occupancy check (>= 1 workgroup per CU)
void set_error(const char* fmt, ...);
// wide-input layer, H = 256, 96 < F <= 256 (lstm_wide.hip)
bool wide_shape_ok(int F, int H);
bool wide_narrow_preferred(int B, int F, int H);
// width 512 on 32 workgroups per tile (lstm_wide16.hip): small batches (one tile per group), K in registers for F <= 96 or F = 512
bool wide16_shape(int B, int F, int H);
bool wide16_preferred(const float* x, int B, int F, int H);
int launch_wide16(const LstmParams& p, hipStream_t stream);
bool wide512_shape_ok(int F, int H, bool zx);   // width 512: 16 workgroups per tile; F <= 96 in-kernel, else precomputed x.K
int launch_wide(const LstmParams& p, hipStream_t stream);
// step-wise layer forward on the matrix-core GEMM (train_kernels.hip): hidden widths above the persistent kernels' 256
bool stepwise_preferred(int B, int F, int H);
size_t stepwise_workspace_floats(int B, int T, int H);
int launch_stepwise(const LstmParams& p, float* ws, size_t ws_floats, hipStream_t stream);

constexpr size_t kStatusBytes = 256;  // head of every workspace: status words (layout: xch_common.h)

// ConvLSTM building blocks (conv_kernels.hip)
// dil: dilation of the taps over x (Keras dilation_rate; 'same' padding grows with it); the second segment of conv2d_fwd2 / the
// recurrent map of convlstm_cell_fwd is never dilated (Keras ConvLSTM2D dilates its input convolution only)
int conv2d_fwd(const float* x, long ldx, long ldb, const float* w, const float* bias, const float* add, float* y, int B, int H,
               int W, int C, int N, int kh, int kw, int act, hipStream_t stream, int dil = 1);
int conv2d_fwd2(const float* x, long ldx, long ldb, int C, const float* x2, long ldx2, long ldb2, int C2, const float* w,
                const float* bias, const float* add, float* y, int B, int H, int W, int N, int kh, int kw, int act,
                hipStream_t stream, int dil = 1);
int convlstm_gates(const float* z, float* c, float* h, long ldh, long rows, int F, int act, hipStream_t stream);
// LDS-resident-patch form of the ConvLSTM2D step (convlstm_patch.hip); convlstm_cell_fwd takes it when the shape allows
// 'same' Conv2D with the input map resident in LDS (conv_patch.hip): the wide layers of the ConvLSTM prediction head
bool conv_patch_shape_ok(const float* x, long ldx, long ldb, int B, int H, int W, int C, int N, int kh, int kw);
int launch_conv_patch(const float* x, long ldx, long ldb, const float* w, const float* bias, const float* add, float* y, int B, int H,
                      int W, int C, int N, int kh, int kw, int act, hipStream_t stream);
bool cell_patch_shape_ok(const float* x, long ldx, long ldb, int C, const float* h_prev, long ldx2, long ldb2, int F, int H, int W,
                         int kh, int kw);
int launch_cell_patch(const float* x, long ldx, long ldb, int C, const float* h_prev, long ldx2, long ldb2, const float* w,
                      const float* bias, const float* c_prev, float* c_new, float* h, long ldh, float* gates, int B, int H, int W,
                      int F, int kh, int kw, int act, hipStream_t stream);
int convlstm_cell_fwd(const float* x, long ldx, long ldb, int C, const float* h_prev, long ldx2, long ldb2, const float* w,
                      const float* bias, const float* c_prev, float* c_new, float* h, long ldh, float* gates, int B, int H, int W,
                      int F, int kh, int kw, int act, hipStream_t stream, int dil = 1);
int softmax_lastdim(const float* x, float* y, long rows, int n, hipStream_t stream);

// ConvLSTM training (conv_train_kernels.hip)
bool conv_wgrad_lines_takes(int H, int W, int C, int N, int kh, int kw, int dil);      // conv_wgrad_lines.hip
int conv_wgrad_lines(const float* x, long ldx, const float* dz, float* dw, int B, int H, int W, int C, int N, int ks, int accumulate,
                     float* scratch, size_t scratch_floats, hipStream_t stream);
size_t conv2d_wgrad_workspace_floats(int C, int N, int kh, int kw);
int conv2d_wgrad(const float* x, long ldx, const float* dz, float* dw, int B, int H, int W, int C, int N, int kh, int kw,
                 int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream, int dil = 1);
int convlstm_gates_train(const float* z, const float* c_prev, float* c_new, float* h, long ldh, float* gates, long rows, int F,
                         int act, hipStream_t stream);
int convlstm_gates_bwd(const float* dh, long lddh, float* dc, const float* gates, const float* c_prev, const float* c_new,
                       float* dz, long rows, int F, int act, hipStream_t stream);
int conv_weight_transpose(const float* w, float* wt, int kh, int kw, int C, int N, hipStream_t stream);
int softmax_lastdim_bwd(const float* dp, const float* p, float* dy, long rows, int n, hipStream_t stream);
int splitk_reduce(const float* part, float* out, long n, int S, int accumulate, hipStream_t stream);
// deferred split reductions of a training step (train_kernels.hip): one reduce launch per step instead of one per product
int defer_begin(float* grad_base, size_t grad_floats, float* arena, size_t arena_floats, hipStream_t stream);
int defer_flush(const float* grad_base, hipStream_t stream);
int defer_end(const float* grad_base, hipStream_t stream);
int defer_touch(const float* out, size_t n, hipStream_t stream);
float* defer_alloc(const float* out, size_t n, size_t floats, hipStream_t stream);
void defer_record(const float* part, float* out, long n, int S, int accumulate);
int reduce_or_defer(bool deferred, const float* part, float* out, long n, int S, int accumulate, hipStream_t stream, const char* what);
int colsum(const float* x, float* out, long rows, int cols, int accumulate, float* scratch, size_t scratch_floats,
           hipStream_t stream);

// fused decoder of the others-mixing model (mix_decoder.hip)
struct MixDecParams {
    const float *K1, *R1, *b1, *K2p, *R2, *b2, *Wd, *bd, *Wp;
    const float* oth_proj;   // others_t . W_oth + b_mix, element (b, t, o) at b*oth_sb + t*oth_st + o
    long oth_sb, oth_st;
    const float* dec0;       // (B,O)
    const float *h1_0, *c1_0, *h2_0, *c2_0;   // (B,H)
    float* out;              // (T_out,B,O) step-major: m_t
    float* P;                // TRAIN: (T_out,B,O)
    float *H1, *C1, *H2, *C2;   // TRAIN: (T_out,B,H), state after step t
    float *res1, *res2;      // TRAIN: (T_out,B,5,H) activated i,f,g,o and c of step t
    float *h1T, *c1T, *h2T, *c2T;   // final state (B,H), optional
    unsigned long long* xch; // granules: [group][layer 2][parity 2][16][256]
    unsigned* status;
    int B, T_out, O, num_groups, num_tiles;
    int epoch_span;          // epochs this launch may consume (xch_common.h)
};
// one-shot marks 'the packed K2 of this workspace is current' (fov_api.hip)
void prepack_mark(void* workspace, const float* K2);
bool prepack_consume(void* workspace, const float* K2);
void prepack_forget(void* workspace);                    // the workspace was zero-filled: no packed copy in it any more
int mix_decoder_prepack(const float* K2, void* workspace, hipStream_t stream);
int mix_decoder_bwd_prepack(const float* K2, void* workspace, hipStream_t stream);
size_t mix_decoder_workspace_bytes(int B);
int mix_decoder_launch(MixDecParams p, const float* K2, int act, int train, void* workspace, hipStream_t stream);
int mix_decoder_bf16_launch(MixDecParams p, const float* K2, int act, int train, void* workspace, hipStream_t stream);   // mix_decoder_bf16.hip

// backward of the fused decoder (mix_decoder_bwd.hip)
struct MixDecBwdParams {
    const float *R1, *K1, *R2, *K2p, *Wd, *Wp;
    const float *M, *P, *dloss;      // (T_out,B,O): m_t, p_t, dL/d(pre-tanh of m_t) from the loss
    const float *res1, *res2;        // (T_out,B,5,H) reserves of the forward
    const float *C1, *C2;            // (T_out,B,H): cell state BEFORE step t (row 0 = the decoder's initial state)
    float *DZ1, *DZ2;                // (T_out,B,4H) out
    float *dpre_m, *dpre_p;          // (T_out,B,O) out
    float *dh1_0, *dc1_0, *dh2_0, *dc2_0;   // (B,H) out: gradient w.r.t. the decoder's initial state
    unsigned long long* xch;
    unsigned* status;
    int B, T_out, O, num_groups, num_tiles;
    int epoch_span;          // epochs this launch may consume (xch_common.h)
};
size_t mix_decoder_bwd_workspace_bytes(int B);
int mix_decoder_bwd_launch(MixDecBwdParams p, const float* K2, int act, void* workspace, hipStream_t stream);
int mix_decoder_bwd_bf16_launch(MixDecBwdParams p, const float* K2, int act, void* workspace, hipStream_t stream);   // mix_decoder_bwd_bf16.hip

// persistent BPTT recurrence (lstm_bwd_cluster.hip)
bool bwd_cluster_shape_ok(int H);
// lstm_wide16.hip: two stacked width-512 layers as one launch (layer 2 a few steps behind layer 1 on other CUs)
bool wide16_pair_shape(int B, int T, int F, int H);
int launch_wide16_pair(const LstmParams& a, const LstmParams& b, hipStream_t stream);
// lstm_bwd16.hip: BPTT recurrence with 16 / 32 units per workgroup (fp32): width 512, and 128 / 256 at small batches
bool wgrad_rows_takes(long rows, int H);       // wgrad_group.hip: few rows AND narrow layers - 16 x 64 tiles, rows split over a workgroup's waves
int wgrad_rows_layers(int L, const float* const* x, const int* F, const int* T, const float* const* hs, const float* const* h0,
                      const float* const* dz, float* const* dK, float* const* dR, float* const* db, int B, int H, int accumulate, hipStream_t stream);
bool lstm_seq_wgrad_pair_one_launch(int B, int T1, int T2, int H);
int lstm_seq_wgrad_pair(const float* x1, const float* hs1, const float* h0_1, const float* dz1, float* dK1, float* dR1, float* db1, int T1, int F1,
                        const float* x2, const float* hs2, const float* h0_2, const float* dz2, float* dK2, float* dR2, float* db2, int T2, int F2,
                        int B, int H, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream);
bool wgrad_group_takes(int B, int T, int H);   // wgrad_group.hip: few rows - every (product, output tile) one workgroup, one launch
int wgrad_group_layers(int L, const float* const* x, const int* F, const float* const* hs, const float* const* h0, const float* const* dz,
                       float* const* dK, float* const* dR, float* const* db, int B, int T, int H, int accumulate, hipStream_t stream);
bool bwd16_pair_shape(int B, int T, int H);
int launch_bwd16_pair(const float* R2, const float* K2, const float* reserve2, const float* c0_2, const float* dhs2, const float* dhT2,
                      const float* dcT2, float* dz2, float* dh0_2, float* dc0_2, float* db_part2, const float* R1, const float* reserve1,
                      const float* c0_1, const float* dhT1, const float* dcT1, float* dz1, float* dh0_1, float* dc0_1, float* db_part1,
                      int B, int T, int act, void* xch_ws, hipStream_t stream);
size_t lstm_stack2_bwd_workspace_floats(int B, int T, int F, int H);
int lstm_stack2_bwd(const float* x, const float* R1, const float* K2, const float* R2, const float* h0_1, const float* c0_1,
                    const float* h0_2, const float* c0_2, const float* hs1, const float* res1, const float* hs2, const float* res2,
                    const float* dhs2, const float* dhT2, const float* dcT2, const float* dhT1, const float* dcT1, float* dz1, float* dz2,
                    float* dK1, float* dR1, float* db1, float* dK2, float* dR2, float* db2, float* dh0_1, float* dc0_1, float* dh0_2,
                    float* dc0_2, int B, int T, int F, int H, int act, int accumulate, float* ws, size_t ws_floats, hipStream_t stream);
bool bwd16_takes(int B, int H);
int launch_bwd16(const float* R, const float* reserve, const float* c0, const float* dhs, const float* dhT, const float* dcT,
                 float* dz, float* dh0, float* dc0, float* db_part, int B, int T, int H, int act, void* xch_ws, hipStream_t stream);
size_t bwd_cluster_xch_bytes(int B, int H);
int launch_bwd_cluster(const float* R, const float* reserve, const float* c0, const float* dhs, const float* dhT,
                       const float* dcT, float* dz, float* dh0, float* dc0, float* db_part, int B, int T, int H, int act,
                       void* xch_ws, hipStream_t stream);

// training side (train_kernels.hip)
size_t lstm_bwd_workspace_floats(int B, int T, int F, int H);
int lstm_seq_wgrad(const float* x, const float* hs, const float* h0, const float* dz, float* dK, float* dR, float* db, int B, int T,
                   int F, int H, int accumulate, int bf16, float* scratch, size_t scratch_floats, hipStream_t stream);
int lstm_seq_bwd(const float* x, const float* K, const float* R, const float* h0, const float* c0, const float* hs,
                 const float* reserve, const float* dhs, const float* dhT, const float* dcT, float* dz, float* dx,
                 float* dK, float* dR, float* db, float* dh0, float* dc0, int B, int T, int F, int H, int act,
                 int accumulate, float* ws, size_t ws_floats, hipStream_t stream, int bf16 = 0);
size_t mix_head_wgrad_scratch_floats(int B, int T, int H, int O, int n_oth);
int mix_head_wgrad(const float* h2, const float* dpre_p, const float* others, const float* p, const float* dpre_m, float* out, int B,
                   int T, int H, int O, int n_oth, int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream);
bool wgrad_fusable(const float* a1, long lda1, long a1_so, int M1, const float* a2, long lda2, long a2_so, int M2, const float* b,
                   long ldb, long b_so, const float* c, int N);
int wgrad_fused(const float* a1, long lda1, long a1_so, int M1, int shift1, const float* a2, long lda2, long a2_so, int M2, int shift2,
                const float* b, long ldb, long b_so, float* c, int N, int RO, int RI, int bias_row, int accumulate, int bf16,
                float* scratch, size_t scratch_floats, hipStream_t stream);
int dense_bwd(const float* x, const float* W, const float* dpre, float* dx, float* dW, float* db, int N, int In, int Out,
              int accumulate, float* scratch, size_t scratch_floats, hipStream_t stream, int bf16 = 0);
int mse_dense_grad(const float* y, const float* target, float* dpre, float* loss, long n, int activation, float* scratch,
                   size_t scratch_floats, hipStream_t stream);
bool dense_mse_head_shape_ok(long N, int H, int O);
size_t dense_mse_head_scratch_floats(long N, int H, int O);
int dense_mse_head(const float* hs, const float* W, const float* b, const float* target, float* y, float* dX, float* dW, float* db,
                   float* loss, long N, int H, int O, int activation, float weight, float* scratch, size_t scratch_floats, hipStream_t stream);
int mse_dense_grad_w(const float* y, const float* target, float* dpre, float* loss, long n, int activation, float weight,
                     int tmB, int tmT, int O, float* scratch, size_t scratch_floats, hipStream_t stream, float* db = nullptr, int dbO = 0);
int scale_inplace(float* x, long n, float s, hipStream_t stream);
int act_bwd(const float* dy, const float* y, const float* base, float* out, long n, int activation, hipStream_t stream);
int matmul_f32(const float* a, const float* b, float* c, int M, int K, int N, float* scratch, size_t scratch_floats,
               hipStream_t stream);
int adam_step(float* p, const float* g, float* m, float* v, long n, float lr_t, float b1, float b2, float eps,
              const unsigned* const* guards, long long* applied, hipStream_t stream);   // guards: NULL or three (nullable) timeout words; applied: nullable counter of updates that ran
int rmsprop_step(float* p, const float* g, float* a, long n, float lr, float rho, float eps, const unsigned* const* guards,
                 long long* applied, hipStream_t stream);
int guard_flag(const unsigned* const* guards, float* out, hipStream_t stream);
int act_fwd(const float* x, float* y, long n, int activation, hipStream_t stream);
int gauss_nll_grad(const float* mu, const float* var, const float* y, float* loss, float* dmu, float* dvar, int B, int Ty,
                   int fps, float scale, float* scratch, size_t scratch_floats, hipStream_t stream);
int rmsprop_tf_step(float* p, const float* g, float* ms, long n, float lr, float decay, float eps, float clip,
                    const unsigned* const* guards, long long* applied, hipStream_t stream);
int cce_grad(const float* p, const float* t, float* dp, float* loss, long n_pix, int C, float* scratch, size_t scratch_floats,
             hipStream_t stream);
int xyz_sum1_grad(const float* p, float* dp, float* reg, long n_pix, int C, float* scratch, size_t scratch_floats, hipStream_t stream);
int sample_refeed_fwd(const float* mu, const float* var, const float* noise, float* x, long ldx, int B, int fps, int std_mode,
                      int layout, hipStream_t stream);
int sample_refeed_bwd(const float* dx, long ldx, const float* var, const float* noise, float* dmu, float* dvar, int B, int fps,
                      int std_mode, int layout, int accumulate, hipStream_t stream);

}  // namespace fov
