/*
 * CPU oracle (C restatement) for the seq2seq-LSTM hot path of ChengeLi/LongTerm360FoV.
 *
 * TEST INFRASTRUCTURE ONLY: linked/loaded by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, never by the product path (longterm360fov_amd/).
 *
 * PARITY STATUS: "parity unpinned" for the cell arithmetic - the reference delegates it to
 * Keras 2.1-2.2 / TensorFlow 1.x (mycode/FoV_seq2seq.py:2-4,83-97), which is neither vendored
 * in /root/reference nor installable here, and the reference holds no golden vectors for it.
 * This file restates the published Keras-2.2 LSTMCell / Dense equations (see
 * oracle/fov_oracle.py for the equations) and is checked against fov_oracle.py and
 * torch.nn.LSTM in tests/test_oracle.py.
 *
 * Follows:
 *   oracle_lstm_layer_f32      - keras LSTM(return_sequences, return_state) as used at
 *                                mycode/FoV_seq2seq.py:83-86,93-95
 *   oracle_seq2seq_decode_f32  - encoder + autoregressive decoder loop,
 *                                mycode/FoV_seq2seq.py:137-178 (batched)
 *   oracle_seq2seq_tf_f32      - teacher-forced training graph forward, :82-101
 *
 * Layouts: row-major contiguous; K:(F,4H) R:(H,4H) b:(4H) gate blocks i,f,c,o.
 * Threads: OpenMP over batch tiles (FOV_TILE sequences share one pass over the weights).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define FOV_TILE 8

static inline float act_sigmoid(float x) { return 1.0f / (1.0f + expf(-x)); }
static inline float act_hard_sigmoid(float x) {
    float y = 0.2f * x + 0.5f;
    return y < 0.f ? 0.f : (y > 1.f ? 1.f : y);
}

/* z[s][n] = b[n] + sum_k a[s][k] W[k][n]  accumulated on top of existing z when add!=0 */
static void tile_gemm(const float* a, int lda, int rows, const float* W, int Kdim, int N, float* z, int ldz) {
    for (int k = 0; k < Kdim; ++k) {
        const float* w = W + (size_t)k * N;
        for (int s = 0; s < rows; ++s) {
            const float av = a[(size_t)s * lda + k];
            float* zs = z + (size_t)s * ldz;
            for (int n = 0; n < N; ++n) zs[n] += av * w[n];
        }
    }
}

static void cell_update(const float* z, int rows, int H, int act, float* h, float* c) {
    for (int s = 0; s < rows; ++s) {
        const float* zs = z + (size_t)s * 4 * H;
        float* hs = h + (size_t)s * H;
        float* cs = c + (size_t)s * H;
        for (int j = 0; j < H; ++j) {
            float i, f, o;
            if (act == 1) {
                i = act_hard_sigmoid(zs[j]); f = act_hard_sigmoid(zs[H + j]); o = act_hard_sigmoid(zs[3 * H + j]);
            } else {
                i = act_sigmoid(zs[j]); f = act_sigmoid(zs[H + j]); o = act_sigmoid(zs[3 * H + j]);
            }
            const float g = tanhf(zs[2 * H + j]);
            const float cn = f * cs[j] + i * g;
            cs[j] = cn;
            hs[j] = o * tanhf(cn);
        }
    }
}

/* one LSTM step for a tile of `rows` sequences; x rows have stride ldx */
static void tile_step(const float* x, int ldx, int rows, int F, int H, const float* K, const float* R,
                      const float* b, int act, float* h, float* c, float* z) {
    for (int s = 0; s < rows; ++s) memcpy(z + (size_t)s * 4 * H, b, sizeof(float) * 4 * H);
    tile_gemm(x, ldx, rows, K, F, 4 * H, z, 4 * H);
    tile_gemm(h, H, rows, R, H, 4 * H, z, 4 * H);
    cell_update(z, rows, H, act, h, c);
}

static void tile_dense_tanh(const float* h, int rows, int H, const float* W, const float* bias, int O, float* y, int ldy) {
    for (int s = 0; s < rows; ++s) {
        float acc[64];
        for (int o = 0; o < O; ++o) acc[o] = bias[o];
        for (int k = 0; k < H; ++k) {
            const float hv = h[(size_t)s * H + k];
            for (int o = 0; o < O; ++o) acc[o] += hv * W[(size_t)k * O + o];
        }
        for (int o = 0; o < O; ++o) y[(size_t)s * ldy + o] = tanhf(acc[o]);
    }
}

void oracle_set_num_threads(int n) {
    if (n > 0) omp_set_num_threads(n);
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int oracle_lstm_layer_f32(const float* x, const float* K, const float* R, const float* b, const float* h0,
                          const float* c0, float* hs, float* hT, float* cT, int B, int T, int F, int H, int act) {
    if (B < 0 || T < 0 || F <= 0 || H <= 0) return -1;
    int err = 0;
#pragma omp parallel
    {
        float* z = (float*)malloc(sizeof(float) * FOV_TILE * 4 * H);
        float* h = (float*)malloc(sizeof(float) * FOV_TILE * H);
        float* c = (float*)malloc(sizeof(float) * FOV_TILE * H);
        if (!z || !h || !c) {
#pragma omp atomic write
            err = -2;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int b0 = 0; b0 < B; b0 += FOV_TILE) {
                const int rows = (B - b0 < FOV_TILE) ? B - b0 : FOV_TILE;
                if (h0) memcpy(h, h0 + (size_t)b0 * H, sizeof(float) * rows * H); else memset(h, 0, sizeof(float) * rows * H);
                if (c0) memcpy(c, c0 + (size_t)b0 * H, sizeof(float) * rows * H); else memset(c, 0, sizeof(float) * rows * H);
                for (int t = 0; t < T; ++t) {
                    tile_step(x + ((size_t)b0 * T + t) * F, T * F, rows, F, H, K, R, b, act, h, c, z);
                    if (hs)
                        for (int s = 0; s < rows; ++s)
                            memcpy(hs + (((size_t)(b0 + s)) * T + t) * H, h + (size_t)s * H, sizeof(float) * H);
                }
                if (hT) memcpy(hT + (size_t)b0 * H, h, sizeof(float) * rows * H);
                if (cT) memcpy(cT + (size_t)b0 * H, c, sizeof(float) * rows * H);
            }
        }
        free(z); free(h); free(c);
    }
    return err;
}

/* weights: enc (K,R,b), dec (K,R,b), dense (W:(H,O), b:(O)) */
int oracle_seq2seq_decode_f32(const float* enc_in, const float* dec_in0, const float* eK, const float* eR,
                              const float* eb, const float* dK, const float* dR, const float* db,
                              const float* dW, const float* dbias, float* out, int B, int T_in, int T_out,
                              int F_enc, int F_dec, int H, int act) {
    if (B < 0 || F_dec > 64) return -1;
    int err = 0;
#pragma omp parallel
    {
        float* z = (float*)malloc(sizeof(float) * FOV_TILE * 4 * H);
        float* h = (float*)malloc(sizeof(float) * FOV_TILE * H);
        float* c = (float*)malloc(sizeof(float) * FOV_TILE * H);
        float* y = (float*)malloc(sizeof(float) * FOV_TILE * F_dec);
        if (!z || !h || !c || !y) {
#pragma omp atomic write
            err = -2;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int b0 = 0; b0 < B; b0 += FOV_TILE) {
                const int rows = (B - b0 < FOV_TILE) ? B - b0 : FOV_TILE;
                memset(h, 0, sizeof(float) * rows * H);
                memset(c, 0, sizeof(float) * rows * H);
                for (int t = 0; t < T_in; ++t)
                    tile_step(enc_in + ((size_t)b0 * T_in + t) * F_enc, T_in * F_enc, rows, F_enc, H, eK, eR, eb, act, h, c, z);
                for (int s = 0; s < rows; ++s)
                    memcpy(y + (size_t)s * F_dec, dec_in0 + (size_t)(b0 + s) * F_dec, sizeof(float) * F_dec);
                for (int t = 0; t < T_out; ++t) {
                    tile_step(y, F_dec, rows, F_dec, H, dK, dR, db, act, h, c, z);
                    tile_dense_tanh(h, rows, H, dW, dbias, F_dec, y, F_dec);
                    for (int s = 0; s < rows; ++s)
                        memcpy(out + (((size_t)(b0 + s)) * T_out + t) * F_dec, y + (size_t)s * F_dec, sizeof(float) * F_dec);
                }
            }
        }
        free(z); free(h); free(c); free(y);
    }
    return err;
}

int oracle_seq2seq_tf_f32(const float* enc_in, const float* dec_in, const float* eK, const float* eR,
                          const float* eb, const float* dK, const float* dR, const float* db,
                          const float* dW, const float* dbias, float* out, int B, int T_in, int T_out,
                          int F_enc, int F_dec, int H, int act) {
    if (B < 0 || F_dec > 64) return -1;
    int err = 0;
#pragma omp parallel
    {
        float* z = (float*)malloc(sizeof(float) * FOV_TILE * 4 * H);
        float* h = (float*)malloc(sizeof(float) * FOV_TILE * H);
        float* c = (float*)malloc(sizeof(float) * FOV_TILE * H);
        if (!z || !h || !c) {
#pragma omp atomic write
            err = -2;
        } else {
#pragma omp for schedule(dynamic, 1)
            for (int b0 = 0; b0 < B; b0 += FOV_TILE) {
                const int rows = (B - b0 < FOV_TILE) ? B - b0 : FOV_TILE;
                memset(h, 0, sizeof(float) * rows * H);
                memset(c, 0, sizeof(float) * rows * H);
                for (int t = 0; t < T_in; ++t)
                    tile_step(enc_in + ((size_t)b0 * T_in + t) * F_enc, T_in * F_enc, rows, F_enc, H, eK, eR, eb, act, h, c, z);
                for (int t = 0; t < T_out; ++t) {
                    tile_step(dec_in + ((size_t)b0 * T_out + t) * F_dec, T_out * F_dec, rows, F_dec, H, dK, dR, db, act, h, c, z);
                    tile_dense_tanh(h, rows, H, dW, dbias, F_dec, out + ((size_t)b0 * T_out + t) * F_dec, T_out * F_dec);
                }
            }
        }
        free(z); free(h); free(c);
    }
    return err;
}
