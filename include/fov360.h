/*
 * fov360.h - C ABI of the MI355X-native seq2seq-LSTM hot path (libfov360_hip.so).
 *
 * Drop-in boundary for the path BASELINE.json:north_star names in ChengeLi/LongTerm360FoV:
 * the reference has no native code; every entry point below replaces a Keras/TensorFlow
 * layer call made by the reference's model scripts (file:line cited per function, relative to
 * /root/reference/).  A maintainer of the reference binds these with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / HIP types in the signatures
 *     (fov_stream_t is an opaque hipStream_t; NULL = the default stream).
 *   - All tensors are DEVICE pointers, row-major contiguous fp32.  The caller owns every
 *     buffer; the library allocates NO DEVICE MEMORY.  Scratch comes from a caller-provided
 *     workspace whose size is queried first (fov_*_workspace_bytes).
 *   - Weights use the Keras layout so weight files round-trip:
 *       kernel K:(F,4H), recurrent_kernel R:(H,4H), bias b:(4H); gate column blocks i,f,c,o.
 *   - A workspace handed to a persistent-kernel entry point (every call that takes `workspace` and whose
 *     fov_*_workspace_bytes is larger than a few KB) is STATEFUL: zero-fill it ONCE after allocating it
 *     (fov_workspace_init, or hipMemset) and then leave its contents alone between calls.  It carries a
 *     256-byte header and a 64 MB granule area whose {value, epoch} words use epoch tags that increase
 *     monotonically across calls, so no call has to clear anything (csrc/xch_common.h).  One workspace
 *     serves one stream at a time; different entry points may share it.  The header's timeout word is
 *     sticky: once a bounded in-kernel wait has given up, every later call on that workspace returns
 *     without computing until fov_check_status has reported (and cleared) the failure.
 *   - Every function returns FOV_OK (0) or a negative FOV_ERR_*; fov_last_error() gives the
 *     message of the calling thread's last failure.  No host synchronisation inside a call
 *     (except fov_check_status, which is the explicit "did the persistent kernel finish
 *     cleanly" query).
 *   - Threading and state.  One stream per call; calls from different threads are safe as long as they do not
 *     share a workspace / scratch / gradient buffer (those are single-owner: one stream at a time).  What the
 *     library keeps between calls, all of it on the HOST, all of it behind mutexes and keyed so that callers do
 *     not meet:
 *       * a thread-local error string (fov_last_error);
 *       * read-only caches filled on first use: device properties per device, and the FOV_* environment knobs
 *         (diagnostic switches between kernel forms; read once, re-read by fov_reload_env);
 *       * per WORKSPACE POINTER: the host copy of the workspace's epoch counter / exchange mode and the address
 *         of a weight matrix a caller pre-packed into it (fov_workspace_init forgets both; call it again when a
 *         workspace's memory is freed and the address comes back from the allocator);
 *       * per open fov_reduce_defer_begin region, keyed by its gradient buffer: the table of pending reductions
 *         (see there);
 *       * per (device, stream): the index of that stream's word in a 1024-word device-side ticket table of the
 *         loss entry points (a module-level __device__ array, not an allocation).  A device on which more than
 *         1024 distinct streams have issued loss calls takes the two-launch form for the later ones
 *         (fov_dense_mse_head refuses there);
 *       * streams exist only where the caller asked for one (fov_stream_create) and belong to the caller.
 *     One process per GPU - the deployment this library is written for - and several devices / trainers /
 *     threads in one process are both within this contract (tests/test_gpu_threads.py drives two trainers
 *     from two threads on two streams against the serial result).
 */
#ifndef FOV360_H
#define FOV360_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* fov_stream_t; /* hipStream_t */

/* recurrent_activation of keras.layers.LSTM.  Keras < 2.3 (the reference's era) defaults to
 * hard_sigmoid = clip(0.2x+0.5,0,1); BASELINE.json:north_star names sigmoid. */
enum { FOV_ACT_SIGMOID = 0, FOV_ACT_HARD_SIGMOID = 1 };

/* kernel family selection */
enum {
    FOV_IMPL_AUTO = 0,    /* cluster kernel when the shape is supported, else generic        */
    FOV_IMPL_GENERIC = 1, /* one workgroup per 4 sequences, weights streamed from L2; any H,F */
    FOV_IMPL_CLUSTER = 2  /* persistent MFMA kernel: H/64 workgroups per 16-sequence tile,
                             recurrent weights resident in registers, h exchanged per step   */
};

enum {
    FOV_OK = 0,
    FOV_ERR_INVALID = -1,     /* bad argument (null pointer, negative size, ...)              */
    FOV_ERR_UNSUPPORTED = -2, /* shape not supported by the requested implementation          */
    FOV_ERR_WORKSPACE = -3,   /* workspace missing or too small                               */
    FOV_ERR_LAUNCH = -4,      /* HIP runtime error (message in fov_last_error)                */
    FOV_ERR_TIMEOUT = -5      /* a bounded in-kernel wait gave up (fov_check_status)          */
};

const char* fov_last_error(void);
int fov_version(void);

/* 1 if FOV_IMPL_CLUSTER supports (F,H) for a layer / (F_enc,F_dec,H) for the fused decode. */
int fov_cluster_supported(int F, int H);

/* ---------------------------------------------------------------------------------------
 * keras.layers.LSTM(H, return_sequences=True, return_state=True)(x, initial_state=[h0,c0])
 *   replaces: mycode/FoV_seq2seq.py:83-86 (encoder), :93-95 (teacher-forced decoder),
 *             mycode/given_others_gt_mean_var_seq2seq.py:108-112.
 *   x:(B,T,F)  h0,c0:(B,H) or NULL (= zeros)  hs:(B,T,H) or NULL  hT,cT:(B,H) or NULL
 * ------------------------------------------------------------------------------------- */
size_t fov_lstm_seq_workspace_bytes(int B, int T, int F, int H, int impl);
int fov_lstm_seq_fwd(const float* x, const float* K, const float* R, const float* b,
                     const float* h0, const float* c0, float* hs, float* hT, float* cT,
                     int B, int T, int F, int H, int act, int impl,
                     void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * keras.layers.Dense(Out, activation)(x)   y = act(x W + b), W:(In,Out)
 *   replaces: mycode/FoV_seq2seq.py:96-97; given_others...py:127-130,168.
 *   activation: 0 = linear, 1 = tanh.   x:(N,In)  y:(N,Out)
 * ------------------------------------------------------------------------------------- */
int fov_dense_fwd(const float* x, const float* W, const float* b, float* y,
                  int N, int In, int Out, int activation, fov_stream_t stream);

/* y = act(x W + b + add[row*add_row_stride + :]) - Dense with an additive per-row term, used to fold the
 * precomputed "others" part of the mixing layer into the decoder step:
 *   replaces: Concatenate(axis=1)([others_t, pred]) -> Flatten -> Dense(6,'tanh')
 *             (mycode/given_others_gt_mean_var_seq2seq.py:257-265), with W = mix_W[-6:], add = others_t . mix_W[:-6]. */
int fov_dense_add_fwd(const float* x, const float* W, const float* b, const float* add, int64_t add_row_stride,
                      float* y, int N, int In, int Out, int activation, fov_stream_t stream);

/* One decoder step of the others-mixing head in ONE launch (given_others...py:127-130,168,257-265):
 *   p (N,O) = tanh(h dense_W + dense_b);   m (N,O) = tanh(p mix_Wp + add[row*add_row_stride + :])
 * with mix_Wp = mix_W[-O:] (O x O) and add = others_t . mix_W[:-O] + mix_b (fov_dense_fwd, hoisted out of the loop).
 * O <= 8, H % 4 == 0.  Backward of the same step: dm_loss = dL/d(pre-tanh of m) from the loss, dm_feedback =
 * dL/dm arriving through x_{t+1} = m_t (NULL at the last step); writes dpre_m, dpre_p (N,O: the pre-activation
 * gradients the weight-gradient products need) and dh (N,H) = dL/dh.  Outputs must not alias inputs. */
int fov_mix_head_fwd(const float* h, const float* dense_W, const float* dense_b, const float* mix_Wp, const float* add,
                     int64_t add_row_stride, float* p, float* m, int N, int H, int O, fov_stream_t stream);
int fov_mix_head_bwd(const float* dm_loss, const float* dm_feedback, const float* m, const float* p, const float* mix_Wp,
                     const float* dense_W, float* dpre_m, float* dpre_p, float* dh, int N, int H, int O,
                     fov_stream_t stream);

/* Weight gradients of that head over ALL steps of the unrolled decoder in one launch + one reduce (what Keras/TF
 * autodiff forms as six separate reductions, given_others...py:308): with rows r = (t, b),
 *     [dense_W ; dense_b] = [h2_t | 1]^T dpre_p,      [mix_W ; mix_b] = [others_t | p_t | 1]^T dpre_m
 * written into out as one (H + 1 + n_others + O + 1, O) block - dense_W, dense_b, mix_W (others' rows, then the
 * prediction's), mix_b as they lie adjacent in a flat gradient buffer.  h2 (T_out*B, H), dpre_p, p, dpre_m (T_out*B, O)
 * time-major; others (B, T_out, n_others) batch-major as the model receives it (no transposed copy).  O <= 8. */
size_t fov_mix_head_wgrad_workspace_bytes(int B, int T_out, int H, int O, int n_others);
int fov_mix_head_wgrad(const float* h2, const float* dpre_p, const float* others, const float* p, const float* dpre_m,
                       float* out, int B, int T_out, int H, int O, int n_others, int accumulate,
                       void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* The whole unrolled no-teacher-forcing decoder of the others-mixing model in ONE persistent launch
 * (given_others...py:203-299): per step LSTM1(x_t) -> LSTM2 -> Dense(O,'tanh') -> mixing Dense -> x_{t+1}.
 *   dec0 (B,O); h1,c1,h2,c2 (B,H) = encoder states; oth_proj: others_t . mix_W[:-O] + mix_b, element (b,t,o) at
 *   b*oth_batch_stride + t*oth_step_stride + o; weights in Keras layout; mix_Wp = mix_W[-O:] (O,O).
 *   out (T_out,B,O) step-major = m_t.  h1T..c2T (B,H): final states or NULL.
 *   Training buffers (all or none): P (T_out,B,O) = p_t; H1,C1,H2,C2 (T_out,B,H) = states after step t;
 *   res1,res2 (T_out,B,5,H) = activated i,f,g,o and c (the reserve layout of fov_lstm_seq_fwd_train with T = 1).
 * Supported: H = 256, O <= 8 (FOV_ERR_UNSUPPORTED otherwise: use the per-step entry points). */
size_t fov_mix_decoder_workspace_bytes(int B, int H);
int fov_mix_decoder_fwd(const float* dec0, const float* h1, const float* c1, const float* h2, const float* c2,
                        const float* oth_proj, int64_t oth_batch_stride, int64_t oth_step_stride,
                        const float* dec1_K, const float* dec1_R, const float* dec1_b,
                        const float* dec2_K, const float* dec2_R, const float* dec2_b,
                        const float* dense_W, const float* dense_b, const float* mix_Wp,
                        float* out, float* h1T, float* c1T, float* h2T, float* c2T,
                        float* P, float* H1, float* C1, float* H2, float* C2, float* res1, float* res2,
                        int B, int T_out, int H, int O, int act,
                        void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* =======================================================================================
 * bf16 forms (BASELINE.json configs[4]): the SAME fp32 tensors in and out, but every product with x or h on its left
 * (gate GEMMs, the Dense(6) head of the fused decoder) takes bf16 operands - round-to-nearest-even of the fp32 values,
 * made on the fly - into v_mfma_f32_16x16x32_bf16 with fp32 accumulation; gates, cell state, mixing layer, tapes and
 * the master weights stay fp32.  H = 256 (F <= 256 for the layer).  Outputs differ from the fp32 entry points by the
 * operand rounding: ~1e-2 absolute on tanh-range outputs after 20 recurrent steps (bound stated in the tests).
 * ======================================================================================= */
int fov_mix_decoder_fwd_bf16(const float* dec0, const float* h1, const float* c1, const float* h2, const float* c2,
                             const float* oth_proj, int64_t oth_batch_stride, int64_t oth_step_stride,
                             const float* dec1_K, const float* dec1_R, const float* dec1_b,
                             const float* dec2_K, const float* dec2_R, const float* dec2_b,
                             const float* dense_W, const float* dense_b, const float* mix_Wp, float* out,
                             float* h1T, float* c1T, float* h2T, float* c2T,
                             float* P, float* H1, float* C1, float* H2, float* C2, float* res1, float* res2,
                             int B, int T_out, int H, int O, int act,
                             void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* fov_mix_decoder_bwd with bf16 operands in its three transposed products (dz2 R2^T, dz2 K2^T, dz1 R1^T). */
int fov_mix_decoder_bwd_bf16(const float* M, const float* P, const float* dloss, const float* res1, const float* res2,
                             const float* C1, const float* C2,
                             const float* dec1_K, const float* dec1_R, const float* dec2_K, const float* dec2_R,
                             const float* dense_W, const float* mix_Wp,
                             float* DZ1, float* DZ2, float* dpre_m, float* dpre_p,
                             float* dh1_0, float* dc1_0, float* dh2_0, float* dc2_0,
                             int B, int T_out, int H, int O, int act,
                             void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* fov_lstm_seq_bwd / fov_dense_bwd with bf16 operands: the recurrence dh_{t-1} = dz_t R^T (eight-workgroup BPTT
 * kernel), the weight-gradient products x^T dz, h_prev^T dz and the data gradient dz K^T all round their operands to
 * bf16 on the fly and accumulate in fp32; gates backward, dz, dc stay fp32.  Same arguments, same workspaces.
 * fov_dense_bwd_bf16 uses the bf16 product for dW when Out >= 64 (the K / R gradients of an unrolled decoder). */
int fov_lstm_seq_bwd_bf16(const float* x, const float* K, const float* R, const float* h0, const float* c0,
                          const float* hs, const float* reserve, const float* dhs, const float* dhT, const float* dcT,
                          float* dz, float* dx, float* dK, float* dR, float* db, float* dh0, float* dc0,
                          int B, int T, int F, int H, int act, int accumulate,
                          void* workspace, size_t workspace_bytes, fov_stream_t stream);
int fov_dense_bwd_bf16(const float* x, const float* W, const float* dpre, float* dx, float* dW, float* db,
                       int N, int In, int Out, int accumulate,
                       void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* BPTT of fov_lstm_stack2_fwd's two layers (mycode/lstm.py:218-240 under the train_op of :556-567) in ONE persistent launch: the
 * upper layer's recurrence, the data-gradient product dx = dz2 . K2^T and the lower layer's recurrence run as three roles on
 * disjoint CUs, the lower layer about a step behind the upper one (as two fov_lstm_seq_bwd calls: recurrence, split product,
 * reduce, recurrence, one after the other).  Arguments are those of the two calls: layer 1 = lower (input x (B,T,F), tape hs1,
 * reserve1), layer 2 = upper (input hs1, tape hs2, reserve2); dhs2 (B,T,H) / dhT2 / dcT2 / dhT1 / dcT1 optional upstream
 * gradients; dz1, dz2 (B,T,4H) out; dK / dR / db of either layer optional (NULL: data path only); dh0 / dc0 optional.  The
 * gradient with respect to x is NOT produced (lstm.py's first layer needs none).  Results equal the two-call path's up to the
 * summation order of dx (sixteen partial sums added in slice order instead of a split GEMM's reduce).
 * Shapes: fov_lstm_stack2_bwd_supported (H = 512, at most 32 sequences on 256 CUs).  workspace: stateful like every exchange
 * workspace (zero-filled once), fov_lstm_stack2_bwd_workspace_bytes. */
int fov_lstm_stack2_bwd_supported(int B, int T, int F, int H);
size_t fov_lstm_stack2_bwd_workspace_bytes(int B, int T, int F, int H);
int fov_lstm_stack2_bwd(const float* x, const float* R1, const float* K2, const float* R2,
                        const float* h0_1, const float* c0_1, const float* h0_2, const float* c0_2,
                        const float* hs1, const float* reserve1, const float* hs2, const float* reserve2,
                        const float* dhs2, const float* dhT2, const float* dcT2, const float* dhT1, const float* dcT1,
                        float* dz1, float* dz2, float* dK1, float* dR1, float* db1, float* dK2, float* dR2, float* db2,
                        float* dh0_1, float* dc0_1, float* dh0_2, float* dc0_2,
                        int B, int T, int F, int H, int act, int accumulate,
                        void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* The weight-gradient half of fov_lstm_seq_bwd[_bf16] on its own: dK (F,4H) = x^T dz, dR (H,4H) = h_{t-1}^T dz (h_{-1} = h0, or
 * zero when h0 is NULL), db (4H) = column sums of dz, from the dz tape (B,T,4H) a call with dK = dR = db = NULL left behind.  Same
 * products in the same order (one fused product when dK, dR, db lie adjacent), so the results equal the single call's bit for bit.
 * It exists so that a trainer can enqueue a layer's products on ANOTHER stream, under the next layer's latency-bound recurrence
 * (Keras computes the same gradients inside `model.fit`, given_others_gt_mean_var_seq2seq.py:494-506).  Any of dK / dR / db may be
 * NULL.  bf16 != 0: operands rounded to bf16 (H = 256).  workspace: fov_lstm_seq_bwd_workspace_bytes(B, T, F, H) bytes of plain
 * scratch - NOT the workspace a concurrently running fov_lstm_seq_bwd uses. */
int fov_lstm_seq_wgrad(const float* x, const float* hs, const float* h0, const float* dz, float* dK, float* dR, float* db,
                       int B, int T, int F, int H, int accumulate, int bf16,
                       void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* The same for an encoder / decoder pair of layers that share the batch (FoV_seq2seq.py:68-93: layer 1 = encoder over T1 steps,
 * layer 2 = decoder over T2 steps, h0_2 = the encoder's final h): all six gradients from the two dz tapes.  When
 * fov_lstm_seq_wgrad_pair_one_launch(B, T1, T2, H) is 1 (few rows, H <= 256: the reference's batch of 32) they are ONE launch
 * without split products or reduces - a trainer then runs both BPTT calls with dK = dR = db = NULL and calls this once; otherwise
 * the call equals two fov_lstm_seq_wgrad calls.  workspace: max over the two layers of fov_lstm_seq_bwd_workspace_bytes. */
int fov_lstm_seq_wgrad_pair_one_launch(int B, int T1, int T2, int H);
int fov_lstm_seq_wgrad_pair(const float* x1, const float* hs1, const float* h0_1, const float* dz1, float* dK1, float* dR1, float* db1,
                            int T1, int F1,
                            const float* x2, const float* hs2, const float* h0_2, const float* dz2, float* dK2, float* dR2, float* db2,
                            int T2, int F2,
                            int B, int H, int accumulate,
                            void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* A HIP stream of a given priority for such side work: priority < 0 high, 0 normal, > 0 low (clamped to the device's range).
 * Work enqueued on a LOW-priority stream yields the compute units to the caller's stream whenever both have workgroups ready:
 * weight-gradient products put there fill the gaps of the persistent recurrence kernels instead of delaying their launch.
 * Destroy with fov_stream_destroy after synchronising it. */
int fov_stream_create(int priority, fov_stream_t* stream);
int fov_stream_destroy(fov_stream_t stream);

/* The three weight gradients of an (unrolled) LSTM layer as ONE product and one reduce - what Keras/TF autodiff
 * computes as three (given_others...py:308 under model.fit):
 *     out (In1 + In2 + bias, Out) (+)= [x1 | x2 | 1]^T dpre        over the N rows (all steps x sequences)
 * i.e. rows [0, In1) = dK = x^T dz, rows [In1, In1+In2) = dR = h_prev^T dz, last row = db = column sums of dz: the layout
 * of a layer's kernel, recurrent_kernel and bias when they are adjacent in a flat gradient buffer.  x1 (N,In1), x2
 * (N,In2) or NULL with In2 = 0, dpre (N,Out), all dense row-major.  dpre is read once instead of three times and the
 * split partials are reduced by one launch.  bf16 != 0: operands rounded to bf16 (the bias row is summed in fp32 from
 * the unrounded values).  Shapes the fused kernel does not take (operands not 16-byte aligned, In1 not a multiple of
 * 128 with x2 given, In1 + In2 <= 96) produce the same result from the separate products.
 * fov_lstm_seq_bwd[_bf16] does the same internally when it is handed adjacent dK, dR, db and no initial state. */
size_t fov_wgrad_fused_workspace_bytes(int64_t N, int In1, int In2, int Out);
int fov_wgrad_fused(const float* x1, int In1, const float* x2, int In2, const float* dpre, float* out,
                    int64_t N, int Out, int bias, int accumulate, int bf16,
                    void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* fov_lstm_seq_fwd_train with bf16 operands (reserve may be NULL for inference); workspace >= 256 + 64 MB. */
int fov_lstm_seq_fwd_bf16(const float* x, const float* K, const float* R, const float* b, const float* h0,
                          const float* c0, float* hs, float* hT, float* cT, float* reserve, int B, int T, int F, int H,
                          int act, void* workspace, size_t workspace_bytes, fov_stream_t stream);
/* Two stacked fp32 LSTM layers (F <= 96 -> 512 -> 512) in ONE launch - mycode/lstm.py:128-132,218-240: MultiRNNCell of two
 * LSTMCell(400) under dynamic_rnn with a fed state, zero-padded to the matrix-core width 512 (models.pad_lstm).  At the
 * script's batch a layer occupies 64 of 256 CUs, so layer 2 runs beside layer 1, a few steps behind: layer 1 publishes h_t of
 * every step as {value, epoch} granules into a ring of T slots and layer 2 takes its input from there (its 512-wide input never
 * comes from HBM).  At most 32 sequences: THREE roles - a third set of workgroups forms layer 2's input projection h1_t . K2 and
 * hands it over in tagged mailboxes, layer 2 keeps R2 only; every 32-workgroup group runs on an XCD of its own.  Same results
 * as two fov_lstm_seq_fwd[_train] calls: bit-identical for layer 1 and for the two-role form (33..64 sequences,
 * FOV_NO_WIDE16_TRIO=1), up to the order of layer 2's fp32 sums (2e-5 relative) in the three-role form.  h0_* / c0_* (B,H) or
 * NULL; reserve* (B,T,5,H) or NULL (training tape); hs1 may be NULL when only the top layer's sequence is wanted.
 * Shapes: fov_lstm_stack2_supported (H = 512, F <= 96, T >= 2, both layers' groups resident: <= 64 sequences on 256 CUs);
 * workspace >= fov_lstm_seq_workspace_bytes of a width-512 layer (header + granule area). */
int fov_lstm_stack2_supported(int B, int T, int F, int H);
int fov_lstm_stack2_fwd(const float* x, const float* K1, const float* R1, const float* b1, const float* h0_1, const float* c0_1,
                        const float* K2, const float* R2, const float* b2, const float* h0_2, const float* c0_2, float* hs1,
                        float* hT1, float* cT1, float* reserve1, float* hs2, float* hT2, float* cT2, float* reserve2, int B, int T,
                        int F, int H, int act, void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* BOTH encoder layers of the others-mixing model with bf16 operands in ONE launch, as a wavefront over (layer, step)
 * (given_others_gt_mean_var_seq2seq.py:108-112: LSTM(F -> 256) then LSTM(256 -> 256), zero initial states): layer 2 runs one
 * step behind layer 1 on the same CUs and takes its input tile straight from layer 1's exchange granules - T + 1
 * exchange-bound steps instead of 2 T.  Same results bit for bit as two fov_lstm_seq_fwd_bf16 calls.  H = 256, F <= 96,
 * T >= 2, at most CUs / 8 tiles of 16 sequences (fov_lstm_stack2_supported_bf16); hs / hT / cT / reserve of either layer may
 * be NULL; workspace as fov_lstm_seq_fwd_bf16. */
int fov_lstm_stack2_supported_bf16(int B, int T, int F, int H);
int fov_lstm_stack2_fwd_bf16(const float* x, const float* K1, const float* R1, const float* b1, const float* K2,
                             const float* R2, const float* b2, float* hs1, float* hT1, float* cT1, float* reserve1,
                             float* hs2, float* hT2, float* cT2, float* reserve2, int B, int T, int F, int H, int act,
                             void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* Backward of fov_mix_decoder_fwd (data path): BPTT through the whole unrolled decoder in ONE persistent launch.
 *   in : M, P (T_out,B,O) from the forward; dloss (T_out,B,O) = dL/d(pre-tanh of m_t) from the loss;
 *        res1, res2 (T_out,B,5,H); C1, C2 (T_out,B,H) = cell state BEFORE step t (row 0 = initial state);
 *        weights as in the forward (mix_Wp = mix_W[-O:]).
 *   out: DZ1, DZ2 (T_out,B,4H) gate pre-activation gradients of both layers and dpre_m, dpre_p (T_out,B,O)
 *        pre-activation gradients of the mixing / Dense layers, for every step (each weight gradient is then one
 *        product over all steps: dK = x^T dz, dR = h_prev^T dz, dW = in^T dpre); dh1_0..dc2_0 (B,H) gradient
 *        w.r.t. the decoder's initial state (= the encoder's final state).
 * Supported: H = 256, O <= 8. */
size_t fov_mix_decoder_bwd_workspace_bytes(int B, int H);
/* The fp32 fused decoder kernels keep a fragment-ordered copy of dec2_K in their workspaces, packed at every launch (the
 * weights change every optimizer step).  The pack depends on the weights only: this call runs it for the forward and / or
 * the backward workspace (either may be NULL) on `stream` - a side stream, under the encoder layers - and marks the
 * workspaces; the NEXT fov_mix_decoder_fwd / _bwd on a marked workspace with the same dec2_K skips its own pack.  The caller
 * orders the streams (the launch must come behind the pack).  The mark is valid for that one launch and only while dec2_K's
 * CONTENTS are unchanged (it is keyed by pointer: a caller that updates the weights in place packs again); fov_workspace_init and
 * the reset inside fov_check_status erase it, since they zero the packed copy. */
int fov_mix_decoder_prepack(const float* dec2_K, void* workspace_fwd, size_t fwd_bytes, void* workspace_bwd, size_t bwd_bytes,
                            int H, fov_stream_t stream);
int fov_mix_decoder_bwd(const float* M, const float* P, const float* dloss, const float* res1, const float* res2,
                        const float* C1, const float* C2, const float* dec1_K, const float* dec1_R,
                        const float* dec2_K, const float* dec2_R, const float* dense_W, const float* mix_Wp,
                        float* DZ1, float* DZ2, float* dpre_m, float* dpre_p,
                        float* dh1_0, float* dc1_0, float* dh2_0, float* dc2_0,
                        int B, int T_out, int H, int O, int act,
                        void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* C (M,N) = A (M,K) . B (K,N), row-major dense fp32 (keras.backend.dot on 2-D operands).  The
 * workspace is optional (NULL allowed): with it, short-and-wide products use split-K. */
size_t fov_matmul_workspace_bytes(int M, int K, int N);
int fov_matmul(const float* a, const float* b, float* c, int M, int K, int N,
               void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* LSTM layer whose input projection zx = x.K (B,T,4H) was computed by the caller (fov_matmul): the
 * form stacked layers use - given_others...py:111-112 (encoder2 over encoder1's sequence) and
 * :217 (decoder_lstm2) - because an (H,4H) input kernel does not fit beside the recurrent one.
 * Bias b is added inside.  Other arguments as fov_lstm_seq_fwd_train (reserve may be NULL). */
int fov_lstm_seq_fwd_zx(const float* zx, const float* R, const float* b, const float* h0, const float* c0,
                        float* hs, float* hT, float* cT, float* reserve,
                        int B, int T, int H, int act, int impl,
                        void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Fused inference path: encoder LSTM over T_in steps from zero state, then T_out
 * autoregressive decoder steps, each = LSTM step + Dense(F_dec,'tanh') + feedback, all on
 * device in ONE launch.
 *   replaces: the host loop mycode/FoV_seq2seq.py:154-178 over encoder_model / decoder_model
 *             (:137-148), batched.
 *   enc_in:(B,T_in,F_enc)  dec_in0:(B,1,F_dec)  out:(B,T_out,F_dec)
 *   hT,cT: final decoder state (B,H) or NULL.
 * ------------------------------------------------------------------------------------- */
size_t fov_seq2seq_decode_workspace_bytes(int B, int T_in, int T_out, int F_enc, int F_dec, int H, int impl);
int fov_seq2seq_decode_fwd(const float* enc_in, const float* dec_in0,
                           const float* enc_K, const float* enc_R, const float* enc_b,
                           const float* dec_K, const float* dec_R, const float* dec_b,
                           const float* dense_W, const float* dense_b,
                           float* out, float* hT, float* cT,
                           int B, int T_in, int T_out, int F_enc, int F_dec, int H,
                           int act, int impl,
                           void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* The decoder half alone: T_out autoregressive steps from a GIVEN state (h0, c0) (B,H) (NULL = zeros) - what the
 * reference's sampling loop does after encoder_model.predict (FoV_seq2seq.py:156-178: decoder_model.predict fed its own
 * output), any batch, one launch.  out (B,T_out,F_dec); hT, cT (B,H) optional final state.  Workspace as for
 * fov_seq2seq_decode_fwd. */
int fov_seq2seq_decoder_fwd(const float* dec_in0, const float* h0, const float* c0, const float* dec_K,
                            const float* dec_R, const float* dec_b, const float* dense_W, const float* dense_b,
                            float* out, float* hT, float* cT, int B, int T_out, int F_dec, int H, int act, int impl,
                            void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * Teacher-forced training-graph forward: encoder, decoder seeded with the encoder state over
 * GT decoder inputs, Dense(tanh) on every decoder output.
 *   replaces: model.predict / the forward half of model.fit on the graph built at
 *             mycode/FoV_seq2seq.py:82-101.
 *   dec_in:(B,T_out,F_dec)  out:(B,T_out,F_dec)
 *   workspace additionally holds the decoder hidden sequence (B,T_out,H).
 * ------------------------------------------------------------------------------------- */
size_t fov_seq2seq_tf_workspace_bytes(int B, int T_in, int T_out, int F_enc, int F_dec, int H, int impl);
int fov_seq2seq_tf_fwd(const float* enc_in, const float* dec_in,
                       const float* enc_K, const float* enc_R, const float* enc_b,
                       const float* dec_K, const float* dec_R, const float* dec_b,
                       const float* dense_W, const float* dense_b,
                       float* out,
                       int B, int T_in, int T_out, int F_enc, int F_dec, int H,
                       int act, int impl,
                       void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * mu / sigma^2 feature op: per row of `fps` interleaved xyz frames -> [mx,my,mz,vx,vy,vz]
 * (population variance, ddof = 0).
 *   replaces: mycode/utility.py:483-500 (get_gt_target_xyz) and :505-517 (_oth); the (N,T,90)
 *   and (N,T,30,3) layouts are the same bytes, so `rows` = N*T (or N*T*U for others).
 *   y:(rows, 3*fps)  out:(rows, 6)
 * ------------------------------------------------------------------------------------- */
int fov_meanvar_xyz(const float* y, float* out, int64_t rows, int fps, fov_stream_t stream);

/* =======================================================================================
 * Training side (a6/a7): what `model.fit` runs under Keras - mycode/FoV_seq2seq.py:103,112-117
 * (model.compile(optimizer='Adam', loss='mean_squared_error'); TensorFlow autodiff = BPTT).
 * ===================================================================================== */

/* fov_lstm_seq_fwd that also writes the reserve (B,T,5,H) = gates i,f,g,o and cell state c of
 * every step, which the backward consumes. */
int fov_lstm_seq_fwd_train(const float* x, const float* K, const float* R, const float* b,
                           const float* h0, const float* c0, float* hs, float* hT, float* cT,
                           float* reserve, int B, int T, int F, int H, int act, int impl,
                           void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* BPTT of one LSTM layer.
 *   in : x (B,T,F), K, R, h0/c0 (B,H) or NULL, hs (B,T,H) and reserve from the forward,
 *        dhs (B,T,H) or NULL (gradient w.r.t. every h_t), dhT/dcT (B,H) or NULL (w.r.t. final state)
 *   out: dz (B,T,4H) pre-activation gradients (always), dx (B,T,F) or NULL,
 *        dK (F,4H), dR (H,4H), db (4H) (each may be NULL), dh0/dc0 (B,H) or NULL
 *   accumulate != 0 adds into dK/dR/db instead of overwriting. */
size_t fov_lstm_seq_bwd_workspace_bytes(int B, int T, int F, int H);
int fov_lstm_seq_bwd(const float* x, const float* K, const float* R, const float* h0, const float* c0,
                     const float* hs, const float* reserve,
                     const float* dhs, const float* dhT, const float* dcT,
                     float* dz, float* dx, float* dK, float* dR, float* db, float* dh0, float* dc0,
                     int B, int T, int F, int H, int act, int accumulate,
                     void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* Dense backward from dpre = dL/d(pre-activation) (N,Out): dW (In,Out) = x^T dpre, db = column sums,
 * dx (N,In) = dpre W^T; each output may be NULL. */
size_t fov_dense_bwd_workspace_bytes(int N, int In, int Out);
int fov_dense_bwd(const float* x, const float* W, const float* dpre, float* dx, float* dW, float* db,
                  int N, int In, int Out, int accumulate,
                  void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* Keras mean_squared_error averaged over all n elements, fused with the Dense activation
 * derivative: dpre = 2 (y - target)/n * (activation == 1 ? 1 - y^2 : 1); *loss = mean((y-target)^2)
 * (loss may be NULL).  workspace >= 4*((n+255)/256 + 64) bytes. */
int fov_mse_dense_grad(const float* y, const float* target, float* dpre, float* loss, int64_t n,
                       int activation, void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* The same with (1) `weight` applied to dpre AND to *loss - under data parallelism a rank passes
 * n_local / n_global, so a plain SUM all-reduce of the flat gradient buffer (which also carries the loss) yields the
 * global-batch mean without a scaling pass - and (2) optionally a time-major prediction: time_major_T > 0 means y
 * and dpre are (T,B,O) while target stays (B,T,O) (the unrolled decoders of given_others...py keep their tape
 * time-major); then n must equal B*T*O.  time_major_T = 0: dense, same order (B, O unused). */
int fov_mse_dense_grad_w(const float* y, const float* target, float* dpre, float* loss, int64_t n, int activation,
                         float weight, int time_major_B, int time_major_T, int O, void* workspace, size_t workspace_bytes,
                         fov_stream_t stream);

/* fov_mse_dense_grad_w (batch-major) that also leaves the Dense head's BIAS gradient: db[o] = sum over the n / O rows of
 * dpre[row][o] (written, not added) - the `db` half of fov_dense_bwd (keras Dense(O) behind mean_squared_error,
 * mycode/FoV_seq2seq.py:96-101,267) from the launch that forms dpre; widths O <= 8 inside the kernel, wider heads by a column-sum
 * launch.  Pass db = NULL to fov_dense_bwd afterwards. */
int fov_mse_dense_grad_db(const float* y, const float* target, float* dpre, float* loss, float* db, int64_t n, int O, int activation,
                          float weight, void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* Dense(O, tanh | linear) head + Keras mean_squared_error, FORWARD AND BACKWARD as ONE launch (round 5; two beyond 4 096 rows: the
 * blocks' partial sums are added by a second one) - what model.fit runs around the
 * decoder's hidden sequence in mycode/FoV_seq2seq.py:96-103: hs (N,H), W (H,O), b (O), target (N,O) ->
 *   y (N,O) = act(hs . W + b) (may be NULL), *loss = weight * mean (y - target)^2 (may be NULL), dX (N,H) = dloss/dhs (may be NULL),
 *   dW (H,O), db (O).  activation 0 linear, 1 tanh; weight as in fov_mse_dense_grad_w.  N <= 2^20, H <= 512 (multiple of 4), O <= 8
 *   (fov_dense_mse_head_supported); other shapes: fov_dense_fwd + fov_mse_dense_grad_db + fov_dense_bwd.  Deterministic. */
int fov_dense_mse_head_supported(int64_t N, int H, int O);
size_t fov_dense_mse_head_workspace_bytes(int64_t N, int H, int O);
int fov_dense_mse_head(const float* hs, const float* W, const float* b, const float* target, float* y, float* dX, float* dW,
                       float* db, float* loss, int64_t N, int H, int O, int activation, float weight, void* workspace,
                       size_t workspace_bytes, fov_stream_t stream);

/* x[0..n) *= s */
int fov_scale(float* x, int64_t n, float s, fov_stream_t stream);

/* Backward of an elementwise activation given its OUTPUT y: out = base + dy * act'(y), act' = 1 - y^2 for tanh
 * (activation 1), [y > 0] for relu (activation 2), y for exp (activation 3) or 1 (activation 0); base may be NULL; out may alias base or dy.  Used where a Dense output is
 * fed back as the next decoder input (given_others...py:292-293): the feedback gradient joins the loss gradient. */
int fov_act_bwd(const float* dy, const float* y, const float* base, float* out, int64_t n, int activation,
                fov_stream_t stream);

/* Pieces of the raw-TensorFlow model's training graph (mycode/lstm.py:218-240,321-337,556-567, cost.py:190-229).
 *   fov_act_fwd: y = act(x), 0 identity, 1 tanh, 2 relu, 3 exp (the two-layer heads of _pred_mean_var_xyz2_new);
 *     fov_act_bwd takes the same codes (exp: dy * y).
 *   fov_gauss_nll_grad: likelihood_loss_tf - mu, var (B,3); y (B,T_y,3*fps) interleaved x,y,z; per element
 *     l = log(var + 1e-20) + (y - mu)^2 / (var + 1e-20) clipped to [-10,10]; *loss = scale * mean_b sum l (scale =
 *     1/(running_length*fps) under cfg.process_in_seconds); dmu, dvar (B,3).  workspace >= 4*(B + 64) bytes.
 *   fov_rmsprop_tf_step: tf.train.RMSPropOptimizer (momentum 0): g' = clip_by_value(g, -clip, clip) if clip > 0;
 *     ms = decay*ms + (1-decay) g'^2; p -= lr * g' / sqrt(ms + eps)   (TF initialises ms to ONE, eps = 1e-10). */
int fov_act_fwd(const float* x, float* y, int64_t n, int activation, fov_stream_t stream);
int fov_gauss_nll_grad(const float* mu, const float* var, const float* y, float* loss, float* dmu, float* dvar,
                       int B, int T_y, int fps, float scale, void* workspace, size_t workspace_bytes,
                       fov_stream_t stream);
int fov_rmsprop_tf_step(float* params, const float* grads, float* ms, int64_t n, float lr, float decay, float eps,
                        float clip_value, fov_stream_t stream);
/* The same, fail-stop like fov_adam_step_guarded: skipped on the device when the sticky timeout word of one of the guard
 * workspaces (each may be NULL) is set - under data parallelism the all-reduced poison slot of the flat gradient buffer;
 * *applied += 1 by every update that ran (lstm.py:583-660's loop keeps going, fov_check_status reports once per epoch). */
int fov_rmsprop_tf_step_guarded(float* params, const float* grads, float* ms, int64_t n, float lr, float decay, float eps,
                                float clip_value, const void* guard0, const void* guard1, const void* guard2,
                                int64_t* applied, fov_stream_t stream);

/* The two two-layer heads of _pred_mean_var_xyz2_new (mycode/lstm.py:321-337) on the top layer's final state h (B,H), as one
 * launch each way: a1 = relu(h mu_W1 + mu_b1), mu = tanh(a1 mu_W2 + mu_b2), a3 = relu(h var_W1 + var_b1),
 * var = exp(a3 var_W2 + var_b2); W1 (H,M), W2 (M,O).  fov_tf_head_bwd takes d loss / d mu and d loss / d var (B,O) and writes
 * (accumulate != 0: adds) the eight weight gradients and dh (B,H) (always overwritten).  Shapes the fused kernels take:
 * fov_tf_head_supported (B <= 64, H <= 2048, M <= 32, O <= 8: the script runs B = 32, H = 400, M = 32, O = 3); for others the
 * same graph is available from fov_dense_fwd / fov_act_fwd / fov_dense_bwd / fov_act_bwd. */
int fov_tf_head_supported(int B, int H, int M, int O);
int fov_tf_head_fwd(const float* h, const float* mu_W1, const float* mu_b1, const float* mu_W2, const float* mu_b2,
                    const float* var_W1, const float* var_b1, const float* var_W2, const float* var_b2, float* a1, float* mu,
                    float* a3, float* var, int B, int H, int M, int O, fov_stream_t stream);
int fov_tf_head_bwd(const float* h, const float* mu_W1, const float* mu_W2, const float* var_W1, const float* var_W2,
                    const float* a1, const float* mu, const float* a3, const float* var, const float* dmu, const float* dvar,
                    float* g_mu_W1, float* g_mu_b1, float* g_mu_W2, float* g_mu_b2, float* g_var_W1, float* g_var_b1,
                    float* g_var_W2, float* g_var_b2, float* dh, int B, int H, int M, int O, int accumulate, fov_stream_t stream);

/* The heads of mycode/lstm.py's other two branches - the ones taken when cfg.predict_mean_var is False, i.e. under the committed
 * mycode/config.py:69,71 - as a fused chain of up to four small Dense layers on the top layer's final state (B rows, the script runs 32):
 *   _GMM_3dgassian (lstm.py:377-400): dims {H, 64, 128, 256, 10 n_mix}, activations relu (2) x3, final_mode 1 = the mixture split
 *     of :386-399 on the last layer: [n softmax weights (exp / sum exp) | 3n means | 3n sigmas = exp | 3n correlations = tanh];
 *   pred_cnn_model_fn (lstm.py:147-174): dims {H, 128, 256, 3 fps}, activations relu, relu, tanh (1), final_mode 0 - the centre taps
 *     of the three k = 5 'same' conv1d kernels, all that one time step ever meets.
 * W[l] (dims[l], dims[l+1]) row-major, b[l] (dims[l+1]); masks (array or NULL) hold optional (B, dims[l+1]) multipliers applied behind
 * layer l's activation (tf.layers.dropout, already scaled; the script calls it with training=False, so NULL is the reference);
 * acts[l] (B, dims[l+1]) receives layer l's output (behind activation and mask; the last one behind the split) - the tape of the
 * backward.  fov_mlp_head_bwd takes dlast = d loss / d PRE-activation of the last layer (what fov_gmm3d_loss_grad and
 * fov_mse_dense_grad produce) and writes (accumulate != 0: adds) gW[l], gb[l] and, if dx != NULL, dx (B, dims[0]).
 * Shapes: 1..4 layers, dims[0] <= 2048, the other widths <= 512 (fov_mlp_head_supported); pointer arrays live on the HOST, what they point to on the device. */
int fov_mlp_head_supported(int B, int L, const int* dims);
int fov_mlp_head_fwd(const float* x, const float* const* W, const float* const* b, const float* const* masks, float* const* acts,
                     const int* dims, const int* act_codes, int L, int final_mode, int n_mix, int B, fov_stream_t stream);
size_t fov_mlp_head_bwd_workspace_bytes(int B, int L, const int* dims);
int fov_mlp_head_bwd(const float* x, const float* const* W, const float* const* masks, const float* const* acts, const float* dlast,
                     float* const* gW, float* const* gb, float* dx, const int* dims, const int* act_codes, int L, int B,
                     int accumulate, void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* costfunc.mixture_3d_gaussian_loss (mycode/cost.py:486-549) with its gradient.  params (B, 10 n_mix) as fov_mlp_head_fwd's mixture
 * split lays them out (mixture m: means params[n+3m..], sigmas params[4n+3m..], rho12, rho13, rho23 params[7n+3m..]); y: n_pts frames
 * (x, y, z interleaved) per row, rows ldy floats apart (cfg.process_in_seconds: second 0 of y (B,T_y,3 fps), ldy = T_y * 3 fps, n_pts =
 * fps; per frame: y (B,T,3), n_pts = T).  Per (row, mixture): covariance [[s1^2, r12 s1 s2, r13 s1 s3], ...], repaired as
 * cost.py:335-348 (smallest eigenvalue lambda < 0: Sigma - 10 lambda I), density N(y_t; mu_m, Sigma'_m);
 *   *loss = scale * sum_{b,t} -log(sum_m [pi_m] N_bmt + 1e-20)      (scale = 1 / (batch_size * running_length [* fps]), cost.py:544-549)
 * weight_by_pi = 0 is the reference: cost.py:532-538 never multiplies by mixture_pi, so the softmax weights get no gradient.
 * dpre (B, 10 n_mix) = d loss / d the head's PRE-activations (through softmax / exp / tanh and the eigenvalue repair).  The 3x3
 * work runs in fp64.  n_mix <= 32, n_pts <= 256.  workspace >= 256 + 4 * B bytes, STATEFUL like the exchange workspaces: its first
 * word is the ticket by which the last workgroup to finish adds the B per-row parts in row order (one launch, run-to-run identical
 * loss); it must be zero on entry (zero-fill the buffer once), every call leaves it zero, and the buffer must not be shared
 * with calls on other streams.
 * fov_gmm3d_sample: one draw per frame - component = first m with cumsum(pi)[m] > u (u (B, n_pts) uniform), value mu_m + chol(Sigma'_m) z
 * (z (B, n_pts, 3) normal) -> out rows of 3 n_pts floats, ldo apart: what utility.sample_mixture_3D (utility.py:178-208) documents and
 * the GMM test loop (lstm.py:735-745,820-825) feeds back; the committed function itself stops in pdb and indexes two dimensions. */
int fov_gmm3d_loss_grad(const float* params, const float* y, int64_t ldy, float* loss, float* dpre, int B, int n_mix, int n_pts,
                        float scale, int weight_by_pi, void* workspace, size_t workspace_bytes, fov_stream_t stream);
int fov_gmm3d_sample(const float* params, const float* u, const float* z, float* out, int64_t ldo, int B, int n_mix, int n_pts,
                     fov_stream_t stream);

/* Keras-2.2 `categorical_crossentropy` on probabilities, TensorFlow backend form - the loss the heat-map fork compiles
 * (mycode/convlstm_heatmap.py:192): per pixel (row of C channels) q = p / sum p, q' = clip(q, 1e-7, 1 - 1e-7),
 * l = - sum_c target_c log q'_c; *loss (may be NULL) = mean over the n_pix rows; dp = d loss / d p, through the clip (zero
 * outside it) and the renormalisation.  p, target, dp (n_pix, C) dense; workspace >= 4*(n_pix/256 + 65) bytes. */
int fov_categorical_crossentropy_grad(const float* p, const float* target, float* dp, float* loss, int64_t n_pix, int C,
                                      void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* The optional term of costfunc._mse under cfg.add_xyz_sum1 (mycode/cost.py:20-29): reg = 0.5 * mean over pixels of
 * (ux^2 + uy^2 + uz^2 - 1)^2 on the first three channels of p (n_pix, C), C >= 3.  Its gradient 2 (s - 1) u_k / n_pix is
 * ADDED into dp (which holds the MSE gradient); *reg (may be NULL) receives the term.  workspace >= 4*(n_pix/256 + 65) bytes. */
int fov_xyz_sum1_grad(const float* p, float* dp, float* reg, int64_t n_pix, int C, void* workspace, size_t workspace_bytes,
                      fov_stream_t stream);

/* Sampled re-feed: the next input second drawn around the predicted mean / variance - what lstm.py:460-468 builds from
 * utility.generate_fake_batch_tf (utility.py:83-89) in its predict_len > 1 training graph, and lstm_keras.py:39-44,139-149
 * in its cfg.sample_and_refeed sampling model.  x[b, e] = mu[b, a(e)] + sd(var[b, a(e)]) * noise[b, e] for the 3*fps
 * elements e of one second; noise ~ N(0,1) (B, 3*fps) comes from the caller (TF's generator is not reproducible anyway)
 * and has the layout of x.  std_mode 0: sd = sqrt(var) (lstm.py), 1: sd = var (lstm_keras.py hands the variance to
 * `stddev`).  layout 0: a(e) = e % 3, frames interleaved x,y,z; 1: a(e) = e / fps, planar.  Rows of x / dx are ldx floats
 * apart, so a slot of the (B,T,3*fps) window is written / read in place.  Backward (reparameterisation, what TF
 * differentiates): dmu[b,a] = sum_e dx, dvar[b,a] = sum_e dx * noise * sd'(var); accumulate != 0 adds into dmu / dvar. */
int fov_sample_refeed_fwd(const float* mu, const float* var, const float* noise, float* x, int64_t ldx, int B, int fps,
                          int std_mode, int layout, fov_stream_t stream);
int fov_sample_refeed_bwd(const float* dx, int64_t ldx, const float* var, const float* noise, float* dmu, float* dvar,
                          int B, int fps, int std_mode, int layout, int accumulate, fov_stream_t stream);

/* Keras-2.2 optimizers on one flat parameter buffer.
 *   Adam   : lr_t = lr*sqrt(1-beta2^step)/(1-beta1^step); p -= lr_t*m/(sqrt(v)+eps)   (step >= 1)
 *   RMSprop: a = rho*a + (1-rho) g^2; p -= lr*g/(sqrt(a)+eps) */
int fov_adam_step(float* params, const float* grads, float* m, float* v, int64_t n,
                  float lr, float beta1, float beta2, float eps, int64_t step, fov_stream_t stream);
int fov_rmsprop_step(float* params, const float* grads, float* accum, int64_t n,
                     float lr, float rho, float eps, fov_stream_t stream);
/* The same, fail-stop: guard0..2 (each may be NULL) are the workspaces the training step's persistent-kernel calls
 * used.  If the sticky timeout word of one of them is set the gradients are garbage and the update is skipped ON THE
 * DEVICE (no host synchronisation); the parameters stay as they were until fov_check_status reports the failure.
 * applied (device pointer, may be NULL): *applied += 1 by every update that was NOT skipped - a caller that counts steps on the
 * host (Adam's bias correction) and only looks at the status now and then reads back how many updates really ran. */
int fov_adam_step_guarded(float* params, const float* grads, float* m, float* v, int64_t n, float lr, float beta1,
                          float beta2, float eps, int64_t step, const void* guard0, const void* guard1,
                          const void* guard2, int64_t* applied, fov_stream_t stream);
int fov_rmsprop_step_guarded(float* params, const float* grads, float* accum, int64_t n, float lr, float rho, float eps,
                             const void* guard0, const void* guard1, const void* guard2, int64_t* applied,
                             fov_stream_t stream);
/* Deferred split reductions of a training step (model.fit's backward, FoV_seq2seq.py:103,112-117): between _begin and
 * _flush / _end every weight-gradient entry point of this library whose OUTPUT lies inside [grad_base, grad_base +
 * grad_floats) - the caller's flat gradient buffer - keeps the partial slices of its split product in `arena` (caller-owned
 * device memory, any size >= 256 bytes: what does not fit is reduced at once) and the flush sums all of them in ONE launch
 * instead of one per product; results are bit-identical.  In-stream order is kept for writes made through this library (a
 * later product over a pending range flushes first); a caller that reads or writes the gradient buffer by other means
 * between _begin and _end calls _flush before.
 * A region is KEYED BY ITS GRADIENT BUFFER: each caller (trainer, thread, stream, device) opens its own with _begin and names
 * it again by grad_base (any address inside the buffer) in _flush / _end; regions of different buffers share nothing - their
 * arenas and pending records are separate - so several may be open at once from different threads (at most 16 per process;
 * _begin on a buffer that overlaps an open region flushes and replaces that region).  The library holds the record table of
 * an open region (16 records, host memory) until _end; it allocates no device memory.  _flush / _end with grad_base = NULL
 * address every open region of the process; _begin with arena = NULL only closes what overlaps grad_base. */
int fov_reduce_defer_begin(float* grad_base, size_t grad_floats, void* arena, size_t arena_bytes, fov_stream_t stream);
int fov_reduce_defer_flush(const float* grad_base, fov_stream_t stream);
int fov_reduce_defer_end(const float* grad_base, fov_stream_t stream);
/* Data-parallel training (the all-reduce of model.fit's gradients, given_others_gt_mean_var_seq2seq.py:494-506):
 * *out = 1.0f if the timeout word of one of the workspaces is set, else 0.0f.  The trainers keep `out` inside the flat
 * gradient buffer: after the SUM all-reduce it is nonzero on EVERY rank if any rank's step failed, and handed to the
 * guarded optimizer as its guard (a nonzero float is a nonzero word) all replicas skip the update together. */
int fov_guard_flag(const void* guard0, const void* guard1, const void* guard2, float* out, fov_stream_t stream);

/* =======================================================================================
 * ConvLSTM2D seq2seq building blocks (a8/a9) - mycode/convlstm_seq2seq.py:100-126,146-165 (ConvLSTM2D),
 * :170-189,224-258 (Conv2D / Conv1D heads, channel Softmax).  NHWC activations, Keras kernel layout
 * (kh,kw,C,N), zero 'same' padding, stride 1, dilation 1, odd kernel sizes.
 * ===================================================================================== */

/* y (B,H,W,N) = act(conv2d_same(x, w) + b + add);  x may be strided: x_pixel_stride >= C floats (a layer can
 * read its input from a slot of a channel-concatenated map) and x_batch_stride >= H*W*x_pixel_stride (one time
 * step of a (B,T,H,W,C) sequence); b, add (B,H,W,N) may be NULL; add may alias y.  activation: 0 none, 2 relu.
 * A Conv1D(k) over (B,W,C) is kh = 1, kw = k, H = 1. */
int fov_conv2d_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, const float* w, const float* b,
                   const float* add, float* y, int B, int H, int W, int C, int N, int kh, int kw, int activation,
                   fov_stream_t stream);

/* The same convolution over the channel concatenation [x1 | x2] (w: (kh,kw,C1+C2,N)) without the concatenated
 * map ever existing: one ConvLSTM2D step is z = conv([x_t | h_{t-1}], [K ; R]) + b in ONE launch
 * (convlstm_seq2seq.py:100-126: Keras runs input_conv and recurrent_conv separately and adds). */
int fov_conv2d_fwd2(const float* x1, int64_t x1_pixel_stride, int64_t x1_batch_stride, int C1,
                    const float* x2, int64_t x2_pixel_stride, int64_t x2_batch_stride, int C2,
                    const float* w, const float* b, const float* add, float* y,
                    int B, int H, int W, int N, int kh, int kw, int activation, fov_stream_t stream);

/* One whole ConvLSTM2D step in ONE launch (convlstm_seq2seq.py:100-126,146-165: Keras' ConvLSTM2DCell.call =
 * input_conv + recurrent_conv + bias, gate activations, c and h update):
 *     z = conv_same([x_t | h_prev], [K ; R]) + b;  i,f,o = recurrent_activation(z_i, z_f, z_o);  g = tanh(z_c)
 *     c_new = f * c_prev + i * g;  h = o * tanh(c_new)
 * x (B,H,W,C) and h_prev (B,H,W,F) with their pixel / batch strides; w (kh,kw,C+F,4F) with Keras' gate order i,f,c,o
 * in the last axis (w = K alone, (kh,kw,C,4F), when h_prev is NULL = zero state); b (4F) or NULL; c_prev (B*H*W,F) or
 * NULL; c_new may alias c_prev; h is written with pixel stride h_pixel_stride >= F and must NOT alias h_prev; gates
 * (B*H*W,4F) or NULL receives the ACTIVATED i,f,g,o (the tape fov_convlstm_gates_bwd reads).  The gate columns of a
 * unit are brought into one lane by the weight staging, so z never exists in memory.  Results are bit-identical to
 * fov_conv2d_fwd2 followed by fov_convlstm_gates[_train]. */
int fov_convlstm_cell_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, int C,
                          const float* h_prev, int64_t h_prev_pixel_stride, int64_t h_prev_batch_stride,
                          const float* w, const float* b, const float* c_prev, float* c_new,
                          float* h, int64_t h_pixel_stride, float* gates,
                          int B, int H, int W, int F, int kh, int kw, int recurrent_activation, fov_stream_t stream);

/* ConvLSTM2DCell gates on z (rows, 4F) = conv(x,K)+b+conv(h,R), channel blocks i,f,c,o; c (rows,F) is
 * updated in place; h is written with pixel stride h_pixel_stride >= F. */
int fov_convlstm_gates(const float* z, float* c, float* h, int64_t h_pixel_stride, int64_t rows, int F,
                       int act, fov_stream_t stream);

/* Softmax over the last dimension of (rows, n) - keras.layers.Softmax(axis=-1), convlstm_seq2seq.py:236. */
int fov_softmax_lastdim(const float* x, float* y, int64_t rows, int n, fov_stream_t stream);

/* ---------------------------------------------------------------------------------------
 * ConvLSTM2D training (a8 backward): what model.fit runs under Keras/TF autodiff for the graph of
 * mycode/convlstm_seq2seq.py:100-126,146-165,209-282 compiled at :287 (RMSprop + costfunc._mse).
 * ------------------------------------------------------------------------------------- */

/* Forward gates that keep what the backward pass needs: gates (rows,4F) = activated i,f,g,o (may alias z),
 * c_new (rows,F) = f*c_prev + i*g (c_prev may be NULL = zero state), h = o*tanh(c_new) with pixel stride. */
int fov_convlstm_gates_train(const float* z, const float* c_prev, float* c_new, float* h, int64_t h_pixel_stride,
                             float* gates, int64_t rows, int F, int act, fov_stream_t stream);

/* Backward of the gates: dh (rows,F) with pixel stride; dc (rows,F) holds dL/dc_t on entry and dL/dc_{t-1} on
 * return; gates, c_prev (NULL = zero), c_new from the forward; dz (rows,4F) out (may alias gates). */
int fov_convlstm_gates_bwd(const float* dh, int64_t dh_pixel_stride, float* dc, const float* gates,
                           const float* c_prev, const float* c_new, float* dz, int64_t rows, int F, int act,
                           fov_stream_t stream);

/* Weight gradient of y = conv2d_same(x, w): dw (kh,kw,C,N) (+)= sum over the B*H*W pixels of
 * x[pixel + tap][c] * dy[pixel][n].  x: batch-dense NHWC with pixel stride x_pixel_stride >= C; dy (B*H*W, N).
 * workspace >= fov_conv2d_wgrad_workspace_bytes (split-K partials; deterministic fixed-order reduce). */
size_t fov_conv2d_wgrad_workspace_bytes(int C, int N, int kh, int kw);
int fov_conv2d_wgrad(const float* x, int64_t x_pixel_stride, const float* dy, float* dw, int B, int H, int W,
                     int C, int N, int kh, int kw, int accumulate, void* workspace, size_t workspace_bytes,
                     fov_stream_t stream);

/* The same three with a dilation (Keras `dilation_rate`; cfg.dilation_rate of mycode/config.py:105 reaches the six ConvLSTM2D
 * layers of mycode/convlstm_seq2seq.py:100-126,146-165): tap (i, j) reads the pixel (i - kh/2, j - kw/2) * dilation away, 'same'
 * zero padding grows with it.  Keras's ConvLSTM2D dilates its INPUT convolution only: fov_convlstm_cell_dilated_fwd applies
 * `dilation` to the taps over x and leaves those over h_prev at 1.  The data gradient of a dilated convolution is the dilated
 * convolution with the transposed weights (fov_conv2d_weight_transpose + fov_conv2d_dilated_fwd).  dilation = 1 is exactly the
 * entry points above; dilation > 1 runs on the tap-gathering implicit GEMM (the LDS-resident forms are built for dilation 1). */
int fov_conv2d_dilated_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, const float* w, const float* b,
                           const float* add, float* y, int B, int H, int W, int C, int N, int kh, int kw, int dilation,
                           int activation, fov_stream_t stream);
int fov_convlstm_cell_dilated_fwd(const float* x, int64_t x_pixel_stride, int64_t x_batch_stride, int C, const float* h_prev,
                                  int64_t h_prev_pixel_stride, int64_t h_prev_batch_stride, const float* w, const float* b,
                                  const float* c_prev, float* c_new, float* h, int64_t h_pixel_stride, float* gates, int B,
                                  int H, int W, int F, int kh, int kw, int dilation, int recurrent_activation,
                                  fov_stream_t stream);
int fov_conv2d_dilated_wgrad(const float* x, int64_t x_pixel_stride, const float* dy, float* dw, int B, int H, int W, int C,
                             int N, int kh, int kw, int dilation, int accumulate, void* workspace, size_t workspace_bytes,
                             fov_stream_t stream);

/* wt (kh,kw,N,C) = w (kh,kw,C,N) flipped in both spatial axes and transposed in the channel axes: the data
 * gradient of y = conv2d_same(x, w) is dx = conv2d_same(dy, wt) (fov_conv2d_fwd). */
int fov_conv2d_weight_transpose(const float* w, float* wt, int kh, int kw, int C, int N, fov_stream_t stream);

/* Backward of fov_softmax_lastdim given its output p: dy = p * (dp - sum_c dp*p); dy may alias dp. */
int fov_softmax_lastdim_bwd(const float* dp, const float* p, float* dy, int64_t rows, int n, fov_stream_t stream);

/* out (cols) (+)= column sums of x (rows, cols): bias gradients.  workspace >= 4*(256*cols + 64) bytes. */
int fov_colsum(const float* x, float* out, int64_t rows, int cols, int accumulate, void* workspace,
               size_t workspace_bytes, fov_stream_t stream);

/* Device-side windowing, the step in front of the path: reshape2second_stacks
 * (mycode/utility.py:264-305).  x:(U,S,feat) whole seconds of one video -> W = fov_window_count(S,T,stride)
 * windows; enc / fut / fut_in are (W*U, T, feat) window-major when collapse_user != 0, else (U, W, T, feat).
 * fut is `T/stride` windows ahead of enc; fut_in is fut shifted right one second, seeded with enc's last second. */
int64_t fov_window_count(int S, int T, int stride);
int fov_window_stacks(const float* x, float* enc, float* fut, float* fut_in, int U, int S, int feat, int T,
                      int stride, int collapse_user, fov_stream_t stream);

/* FoV hit rate per predicted second, the evaluation step that consumes the path's output:
 * centres as unit xyz vectors -> (theta, phi) (mycode/dataIO.py:77-82), +-2pi seam fix
 * (baseline_knn_mean.py:78-85), area(pred box ^ gt box) / area(gt box) (:62-82).  Row strides in floats
 * (>= 3) let `pred_xyz` be the first three columns of the model's (N*T, 6) mean/variance output. */
int fov_fov_hit_rate(const float* pred_xyz, int64_t pred_row_stride, const float* gt_xyz, int64_t gt_row_stride,
                     float* out, int64_t rows, float span_deg, float gt_span_deg, fov_stream_t stream);

/* Zero-fills a freshly allocated workspace (asynchronously on `stream`): required once before its first use by
 * a persistent-kernel entry point, and again whenever the buffer is re-allocated. */
int fov_workspace_init(void* workspace, size_t workspace_bytes, fov_stream_t stream);

/* Synchronises `stream`, reads the sticky status word the persistent-kernel calls leave in `workspace` and
 * returns FOV_OK or FOV_ERR_TIMEOUT (= some call since the previous check gave up a bounded wait; its outputs and
 * those of every later call on this workspace are invalid).  Reporting a failure clears it (the workspace is
 * re-zeroed and usable again).  Workspaces of non-persistent calls report FOV_OK. */
int fov_check_status(void* workspace, size_t workspace_bytes, fov_stream_t stream);
/* Test switch: on != 0 makes every exchanging kernel launched on this workspace keep the placement-independent
 * (write-through) granule exchange even where its run-time handshake finds a whole group on one XCD.  Results are
 * bit-identical either way (tests/test_gpu_parity.py, tests/test_gpu_train.py); fov_exchange_mode tells which ran. */
int fov_workspace_force_safe(void* workspace, size_t workspace_bytes, int on, fov_stream_t stream);
/* The library reads its environment knobs (FOV_FORCE_SAFE_EXCHANGE, FOV_PAIR, FOV_TWO_LAUNCHES, FOV_DBG_RESIDENT_LIMIT)
 * once, at first use; call this after changing them.  No launch path calls getenv. */
void fov_reload_env(void);
/* Diagnostic: launches of the generic (VALU, any-shape) LSTM kernel so far in this process - lets a test assert that a
 * call was served by the persistent matrix-core kernels. */
int64_t fov_debug_generic_launches(void);
/* Test hook: set the workspace's epoch base (device header and the host-side accounting) - lets a test reach the
 * re-zero threshold of the 32-bit epoch tags without 10^7 launches. */
int fov_debug_set_epoch(void* workspace, size_t workspace_bytes, unsigned epoch, fov_stream_t stream);

/* Diagnostic: which h-exchange protocol the last persistent-kernel call on `workspace` used.
 * 1 = every group verified (HW_REG_XCC_ID handshake) that its workgroups share an XCD and took the
 *     L2-resident fast path; 2 = at least one workgroup used the placement-independent
 *     write-through path (also forced by the environment variable FOV_FORCE_SAFE_EXCHANGE=1).
 * Results are identical either way.  Synchronises `stream`. */
int fov_exchange_mode(const void* workspace, size_t workspace_bytes, fov_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* FOV360_H */
