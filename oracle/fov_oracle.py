"""CPU oracle for the seq2seq-LSTM hot path of ChengeLi/LongTerm360FoV.

TEST INFRASTRUCTURE ONLY.  Nothing under ``longterm360fov_amd/`` may import this module; only
``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` do, and
only as the checker.  The product path is the HIP library behind ``include/fov360.h``.

PARITY STATUS
  * Data side (windowing, mu/sigma^2 features, clip): PINNED against the reference's own NumPy
    code, executed in the authoring container on seeded inputs
    (``tests/golden/make_data_fixtures.py`` -> ``tests/golden/data_helpers.npz``).
  * LSTM / Dense arithmetic: **parity unpinned**.  The reference delegates it to Keras 2.1-2.2
    on TensorFlow 1.x (``from keras.layers import LSTM, Dense`` - mycode/FoV_seq2seq.py:2-4),
    neither of which is present in /root/reference or installable here, and the reference holds
    no tests, golden vectors or trained weights for it.  This file
    restates the published Keras-2.2 ``LSTMCell`` / ``Dense`` definitions; the restatement is
    cross-checked against an independent implementation with the same gate order
    (``torch.nn.LSTM`` on CPU, sigmoid mode) and against hand-computed known-answer cases for
    ``hard_sigmoid`` in ``tests/test_oracle.py``.

Equations (Keras 2.2 ``LSTMCell.call``; kernel K:(F,4H), recurrent kernel R:(H,4H), bias b:(4H,),
gate column blocks in the order i, f, c, o):
    z   = x_t @ K + b + h_{t-1} @ R
    i   = s(z_i); f = s(z_f); o = s(z_o); g = tanh(z_c)
    c_t = f * c_{t-1} + i * g
    h_t = o * tanh(c_t)
with s = ``hard_sigmoid`` = clip(0.2 x + 0.5, 0, 1) (Keras < 2.3 default for
``recurrent_activation``; the reference never overrides it) or ``sigmoid`` (what
BASELINE.json's north_star names).  Both are implemented; callers choose with ``act``.
"""
import numpy as np

ACT_SIGMOID = 0
ACT_HARD_SIGMOID = 1
_ACT_NAMES = {"sigmoid": ACT_SIGMOID, "hard_sigmoid": ACT_HARD_SIGMOID,
              ACT_SIGMOID: ACT_SIGMOID, ACT_HARD_SIGMOID: ACT_HARD_SIGMOID}


def act_code(act):
    return _ACT_NAMES[act]


def sigmoid(x):
    # numerically stable logistic in the array's own dtype
    x = np.asarray(x)
    out = np.empty_like(x)
    pos = x >= 0
    out[pos] = 1 / (1 + np.exp(-x[pos]))
    e = np.exp(x[~pos])
    out[~pos] = e / (1 + e)
    return out


def hard_sigmoid(x):
    """Keras backend hard_sigmoid: clip(0.2*x + 0.5, 0, 1)."""
    x = np.asarray(x)
    return np.clip(x.dtype.type(0.2) * x + x.dtype.type(0.5), 0, 1).astype(x.dtype)


def _rec_act(act):
    return hard_sigmoid if act_code(act) == ACT_HARD_SIGMOID else sigmoid


# --------------------------------------------------------------------------------------
# a1/a2: Keras LSTM layer  (mycode/FoV_seq2seq.py:83-86, 93-95)
# --------------------------------------------------------------------------------------
# configs[4] (bf16): the HIP path feeds bf16 operands (round-to-nearest-even of the fp32 value) into the matrix cores
# and accumulates in fp32; cell state, gates and everything elementwise stay fp32.  `bf16_operands()` makes the
# matrix products of the LSTM steps and of the Dense(6) head below round BOTH operands the same way, so the
# restatement can be compared with the bf16 kernels far more tightly than the full-precision one.
OPERAND_ROUND = None


def round_bf16(a):
    """Round-to-nearest-even to bfloat16, returned in the input's dtype (what v_cvt_pk_bf16_f32 does to an fp32)."""
    a32 = np.ascontiguousarray(a, dtype=np.float32)
    u = a32.view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000).astype(np.uint32)
    return r.view(np.float32).reshape(a32.shape).astype(np.asarray(a).dtype)


class bf16_operands:
    """with bf16_operands(): ... -> matrix products of lstm_step / the Dense head see bf16-rounded operands."""

    def __enter__(self):
        global OPERAND_ROUND
        self._prev, OPERAND_ROUND = OPERAND_ROUND, round_bf16
        return self

    def __exit__(self, *exc):
        global OPERAND_ROUND
        OPERAND_ROUND = self._prev
        return False


def _mm(a, w):
    if OPERAND_ROUND is not None:
        a, w = OPERAND_ROUND(a), OPERAND_ROUND(w)
    return a @ w


def lstm_step(x, h, c, K, R, b, act="sigmoid"):
    """One LSTMCell step.  x:(B,F) h,c:(B,H) -> (h', c')."""
    H = h.shape[1]
    z = _mm(x, K) + b + _mm(h, R)
    s = _rec_act(act)
    i = s(z[:, 0 * H:1 * H])
    f = s(z[:, 1 * H:2 * H])
    g = np.tanh(z[:, 2 * H:3 * H])
    o = s(z[:, 3 * H:4 * H])
    c_new = f * c + i * g
    h_new = o * np.tanh(c_new)
    return h_new.astype(x.dtype), c_new.astype(x.dtype)


def lstm_layer(x, K, R, b, h0=None, c0=None, act="sigmoid"):
    """LSTM over x:(B,T,F).  Returns (hs:(B,T,H), h_T, c_T).  Zero initial state by default
    (Keras ``get_initial_state``); ``initial_state=[h, c]`` as in FoV_seq2seq.py:94-95."""
    B, T, _ = x.shape
    H = R.shape[0]
    h = np.zeros((B, H), x.dtype) if h0 is None else h0.astype(x.dtype)
    c = np.zeros((B, H), x.dtype) if c0 is None else c0.astype(x.dtype)
    hs = np.empty((B, T, H), x.dtype)
    for t in range(T):
        h, c = lstm_step(x[:, t], h, c, K, R, b, act)
        hs[:, t] = h
    return hs, h, c


# a3: Dense(num_decoder_tokens, activation='tanh')  (mycode/FoV_seq2seq.py:96-97)
def dense(x, W, b, activation="tanh", matrix_core=False):
    """matrix_core: this product runs on the matrix cores in the bf16 path (operands rounded under bf16_operands())."""
    y = (_mm(x, W) if matrix_core else x @ W) + b
    if activation == "tanh":
        y = np.tanh(y)
    return y.astype(x.dtype)


# --------------------------------------------------------------------------------------
# Target-only seq2seq  (mycode/FoV_seq2seq.py)
# weights: dict(enc_K, enc_R, enc_b, dec_K, dec_R, dec_b, dense_W, dense_b)
# --------------------------------------------------------------------------------------
def seq2seq_teacher_forced(enc_in, dec_in, w, act="sigmoid"):
    """Training-graph forward, FoV_seq2seq.py:82-101: decoder consumes GT inputs (B,T_out,6)."""
    _, h, c = lstm_layer(enc_in, w["enc_K"], w["enc_R"], w["enc_b"], act=act)
    hs, _, _ = lstm_layer(dec_in, w["dec_K"], w["dec_R"], w["dec_b"], h, c, act=act)
    B, T, H = hs.shape
    return dense(hs.reshape(B * T, H), w["dense_W"], w["dense_b"]).reshape(B, T, -1)


def seq2seq_decode(enc_in, dec_in0, w, T_out, act="sigmoid"):
    """Autoregressive inference, FoV_seq2seq.py:137-178 (batched; the reference runs batch 1):
    states = encoder(enc_in); target = dec_in0 (B,1,6);
    repeat T_out: y,h,c = decoder(target,h,c); y = dense(y); target = y."""
    _, h, c = lstm_layer(enc_in, w["enc_K"], w["enc_R"], w["enc_b"], act=act)
    y = dec_in0[:, 0].astype(enc_in.dtype)
    out = []
    for _ in range(T_out):
        h, c = lstm_step(y, h, c, w["dec_K"], w["dec_R"], w["dec_b"], act)
        y = dense(h, w["dense_W"], w["dense_b"])
        out.append(y)
    return np.stack(out, axis=1)


def onelayer_tar_seq2seq_forward(enc_in, dec_in0, w, T_out, act="sigmoid", decoder_no_init_state=True,
                                 add_residual_link=False, enc_last_out_as_dec_in=False, dense_activation="tanh",
                                 embed_frame_state_enc2dec=False, has_reconstruct_loss=False):
    """Unrolled no-teacher-forcing target-only model, FoV_seq2seq_no_teac_forc.py:37-149 (onelayer_tar_seq2seq):
    encoder LSTM (:42-44); decoder input = dec_in0 (B,1,O), or Dense(encoder output) when
    cfg.enc_last_out_as_dec_in (:75-78); step 0 of the decoder starts from ZERO state when the script's
    `decoder_no_init_state` is set (:29,98-99), later steps carry the decoder's own state (:118);
    y_t = Dense(h_t) [+ residual_dense(decoder input), cfg.add_residual_link, :103-107]; y_t is fed back (:115).
    cfg.embed_frame_state_enc2dec (:47-52): the encoder's final h and c each pass a Dense(latent_dim, tanh) before they
    seed the decoders (the decoder INPUT under enc_last_out_as_dec_in still comes from the raw encoder output, :78).
    cfg.has_reconstruct_loss (:56-59,90-95,120-126): a second self-fed LSTM + Dense(num_encoder_tokens, tanh), seeded with
    the same (embedded) states and fed Dense_recons(encoder output) first, emits T_out reconstructed input seconds;
    the function then returns (prediction, reconstruction).
    weights: enc_*, dec_*, dense_W/b; res_W (O,O) / res_b; emb1_W/b, emb2_W/b (H,H); rec_K/R/b, recd_W (H,F) / recd_b."""
    fa = (lambda v: np.tanh(v)) if dense_activation == "tanh" else (lambda v: np.maximum(v, 0))
    _, h_enc, c_enc = lstm_layer(enc_in, w["enc_K"], w["enc_R"], w["enc_b"], act=act)
    sh, sc = h_enc, c_enc
    if embed_frame_state_enc2dec:
        sh = np.tanh(h_enc @ w["emb1_W"] + w["emb1_b"])
        sc = np.tanh(c_enc @ w["emb2_W"] + w["emb2_b"])
    if enc_last_out_as_dec_in:
        x0 = fa(h_enc @ w["dense_W"] + w["dense_b"])
    else:
        x0 = dec_in0[:, 0].astype(enc_in.dtype)
    h, c = (np.zeros_like(sh), np.zeros_like(sc)) if decoder_no_init_state else (sh, sc)
    r = fa(x0 @ w["res_W"] + w["res_b"]) if add_residual_link else 0.0
    x, out = x0, []
    for _ in range(T_out):
        h, c = lstm_step(x, h, c, w["dec_K"], w["dec_R"], w["dec_b"], act)
        x = fa(h @ w["dense_W"] + w["dense_b"]) + r
        out.append(x)
    y = np.stack(out, axis=1)
    if not has_reconstruct_loss:
        return y
    x, h, c, rec = np.tanh(h_enc @ w["recd_W"] + w["recd_b"]), sh, sc, []
    for _ in range(T_out):
        h, c = lstm_step(x, h, c, w["rec_K"], w["rec_R"], w["rec_b"], act)
        x = np.tanh(h @ w["recd_W"] + w["recd_b"])
        rec.append(x)
    return y, np.stack(rec, axis=1)


def stacked_seq2seq_forward(enc_in, dec_in, w, num_layers, act="sigmoid", T_out=None):
    """L-layer target-only seq2seq, Fov_seq2seq_2layers.py:232-272 / 3layers.py:222-277.  weights enc{l}_K/R/b, dec{l}_K/R/b,
    dense_W/b.  T_out None: teacher-forced graph on dec_in (B,T_out,O).  T_out given: the autoregressive loop of
    :399-430 from dec_in (B,1,O), each Dense output fed back."""
    states, inp = [], enc_in
    for l in range(num_layers):
        inp, h, c = lstm_layer(inp, w["enc%d_K" % l], w["enc%d_R" % l], w["enc%d_b" % l], act=act)
        states.append([h, c])
    if T_out is None:
        inp = dec_in
        for l in range(num_layers):
            inp, _, _ = lstm_layer(inp, w["dec%d_K" % l], w["dec%d_R" % l], w["dec%d_b" % l], states[l][0], states[l][1], act=act)
        return dense(inp, w["dense_W"], w["dense_b"])
    x, out = dec_in[:, 0].astype(enc_in.dtype), []
    for _ in range(T_out):
        for l in range(num_layers):
            states[l] = list(lstm_step(x, states[l][0], states[l][1], w["dec%d_K" % l], w["dec%d_R" % l], w["dec%d_b" % l], act))
            x = states[l][0]
        x = dense(x, w["dense_W"], w["dense_b"])
        out.append(x)
    return np.stack(out, axis=1)


def single_lstm_keras_forward(x, w, T_out=None, unrolled=False, noise=None, act="sigmoid"):
    """mycode/lstm_keras.py, weights K, R, b, dense_W, dense_b.  unrolled False: 1st part (:70-80), one input second per
    step from zero state, Dense(6,tanh) on every h.  unrolled True: the sampling model / 2nd part (:131-153,218-240) on
    ONE input second (B,1,F); noise None: `this_inputs` is never replaced, the same second feeds every step; noise
    (T_out-1,B,3*fps): cfg.predict_mean_var and cfg.sample_and_refeed - the next input is
    [N(mu_x, var_x) * fps | N(mu_y, var_y) * fps | N(mu_z, var_z) * fps] with the predicted VARIANCE used as stddev
    (:39-44) = mu + var * noise in that planar layout (:147-149)."""
    if not unrolled:
        hs, _, _ = lstm_layer(x, w["K"], w["R"], w["b"], act=act)
        return dense(hs, w["dense_W"], w["dense_b"])
    B, F = x.shape[0], x.shape[2]
    fps = F // 3
    H = w["R"].shape[0]
    h, c = np.zeros((B, H), x.dtype), np.zeros((B, H), x.dtype)
    xin, out = x[:, 0], []
    for t in range(T_out):
        h, c = lstm_step(xin, h, c, w["K"], w["R"], w["b"], act)
        y = dense(h, w["dense_W"], w["dense_b"])
        out.append(y)
        if noise is not None and t < T_out - 1:
            xin = (y[:, :3, None] + y[:, 3:6, None] * noise[t].reshape(B, 3, fps)).reshape(B, F)
    return np.stack(out, axis=1)


# --------------------------------------------------------------------------------------
# a4: target + others mixing, 2-layer, no teacher forcing
# (mycode/given_others_gt_mean_var_seq2seq.py:98-130, 203-299)
# weights: enc1_*, enc2_*, dec1_*, dec2_*, dense_W/b (H,6), mix_W (6*U,6), mix_b
# --------------------------------------------------------------------------------------
def others_mixing_forward(enc_in, others, dec_in0, w, act="sigmoid", mixing="mlp"):
    """enc_in:(B,T_in,F) others:(B,T_out,U-1,6) dec_in0:(B,1,6) -> (B,T_out,6).
    Per step: d1=LSTM1(x); d2=LSTM2(d1); p=tanh(d2 Wd+bd);
    mixing 'mlp' (mlp_mixing, :166-168,262-265): m=tanh(flatten(concat_axis1[others[:,t], p]) Wm + bm) (user-major flatten,
    pred last); mixing 'conv' (conv_mixing, :188-197,284-290): the (U,6) stack is permuted to a 1x6 map with the U users
    as channels and passes three Conv2D(1x3, same, relu) layers with 8, 8 and 1 filters (weights mixc{0,1,2}_W (1,3,C,N),
    mixc{0,1,2}_b); the single output channel is m.  x=m is fed back."""
    hs1, h1, c1 = lstm_layer(enc_in, w["enc1_K"], w["enc1_R"], w["enc1_b"], act=act)
    _, h2, c2 = lstm_layer(hs1, w["enc2_K"], w["enc2_R"], w["enc2_b"], act=act)
    x = dec_in0[:, 0].astype(enc_in.dtype)
    B, T_out = others.shape[0], others.shape[1]
    out = []
    for t in range(T_out):
        h1, c1 = lstm_step(x, h1, c1, w["dec1_K"], w["dec1_R"], w["dec1_b"], act)
        h2, c2 = lstm_step(h1, h2, c2, w["dec2_K"], w["dec2_R"], w["dec2_b"], act)
        p = dense(h2, w["dense_W"], w["dense_b"], matrix_core=True)    # (B,6)
        cat = np.concatenate([others[:, t].astype(x.dtype), p[:, None, :]], axis=1)  # (B,U,6)
        if mixing == "conv":
            a = np.transpose(cat, (0, 2, 1))[:, None]                   # Permute((2,1)) + expand_dims(1): (B,1,6,U)
            for i in range(3):
                a = np.maximum(conv2d_same(a, w["mixc%d_W" % i], w["mixc%d_b" % i]), 0)
            x = a[:, 0, :, 0]                                           # get_dim1 + Permute((2,1)): (B,6)
        else:
            x = dense(cat.reshape(B, -1), w["mix_W"], w["mix_b"])       # (B,6)
        out.append(x)
    return np.stack(out, axis=1)


def bidirectional_lstm(x, wf, wb, init=None, act="sigmoid"):
    """Keras Bidirectional(LSTM(return_sequences, return_state), merge_mode='concat'): the backward layer reads the
    sequence reversed and its outputs are reversed back before the concat.  wf / wb = (K, R, b); init = (fh, fc, bh, bc) or
    None.  -> (seq (B,T,2H), fh, fc, bh, bc)."""
    i = (None,) * 4 if init is None else init
    f_seq, fh, fc = lstm_layer(x, wf[0], wf[1], wf[2], i[0], i[1], act=act)
    b_seq, bh, bc = lstm_layer(x[:, ::-1], wb[0], wb[1], wb[2], i[2], i[3], act=act)
    return np.concatenate([f_seq, b_seq[:, ::-1]], axis=-1), fh, fc, bh, bc


def others_context_forward(enc_in, others, dec_in0, w, T_out, mode, act="sigmoid"):
    """The other decoder heads of given_others_gt_mean_var_seq2seq.py (2+2-layer model, no teacher forcing, :203-299):
      'target_user_only' (:219-220)  y_t = decoder_dense(h2_t)
      'others_mlp'       (:153-156,223-233)  ctx_t = relu(relu(flatten(others_t) W1 + b1) W2 + b2);  y_t = decoder_dense([ctx_t ; h2_t])
      'others_lstm'      (:157-166,234-240)  ctx = BiLSTM2(BiLSTM1(others reshaped (B,T,(U-1)*6))): calling the second
                         Bidirectional on the LIST the first returns makes layer 1's final states its initial states;
                         y_t = decoder_dense([ctx_t (2H) ; h2_t])
    y_t is fed back as the next decoder input.  weights: enc1/enc2/dec1/dec2 _K/_R/_b, dense_W ((Cc+H),6), dense_b, and
    oth_W1/b1/W2/b2 or ol{1,2}{f,b}_K/_R/_b."""
    hs1, h1, c1 = lstm_layer(enc_in, w["enc1_K"], w["enc1_R"], w["enc1_b"], act=act)
    _, h2, c2 = lstm_layer(hs1, w["enc2_K"], w["enc2_R"], w["enc2_b"], act=act)
    B = enc_in.shape[0]
    ctx = None
    if mode == "others_mlp":
        o = others.reshape(B, T_out, -1).astype(enc_in.dtype)
        ctx = np.maximum(np.maximum(o @ w["oth_W1"] + w["oth_b1"], 0) @ w["oth_W2"] + w["oth_b2"], 0)
    elif mode == "others_lstm":
        o = others.reshape(B, T_out, -1).astype(enc_in.dtype)
        g = lambda n: (w[n + "_K"], w[n + "_R"], w[n + "_b"])
        s1, fh, fc, bh, bc = bidirectional_lstm(o, g("ol1f"), g("ol1b"), act=act)
        ctx = bidirectional_lstm(s1, g("ol2f"), g("ol2b"), (fh, fc, bh, bc), act=act)[0]
    x, out = dec_in0[:, 0].astype(enc_in.dtype), []
    for t in range(T_out):
        h1, c1 = lstm_step(x, h1, c1, w["dec1_K"], w["dec1_R"], w["dec1_b"], act)
        h2, c2 = lstm_step(h1, h2, c2, w["dec2_K"], w["dec2_R"], w["dec2_b"], act)
        x = dense(h2 if ctx is None else np.concatenate([ctx[:, t], h2], axis=1), w["dense_W"], w["dense_b"])
        out.append(x)
    return np.stack(out, axis=1)


# --------------------------------------------------------------------------------------
# a5: mu / sigma^2 features  (mycode/utility.py:483-517)
# --------------------------------------------------------------------------------------
def meanvar_xyz(y):
    """(N,T,90) interleaved xyzxyz.. or (N,T,30,3) -> (N,T,6) = [mx,my,mz,vx,vy,vz]; ddof=0.
    Slices per axis exactly as utility.py:484-499 does, so float64 results are bit-identical."""
    if y.shape[-1] == 3:
        assert y.ndim == 4
        comps = [y[:, :, :, a] for a in range(3)]
    else:
        assert y.ndim == 3 and y.shape[-1] % 3 == 0
        comps = [y[:, :, a::3] for a in range(3)]
    means = [np.mean(c, axis=-1)[:, :, np.newaxis] for c in comps]
    variances = [np.var(c, axis=-1)[:, :, np.newaxis] for c in comps]
    return np.concatenate(means + variances, axis=-1)


def meanvar_xyz_oth(y):
    """(N,T,U,30,3) -> (N,T,U,6)   (utility.py:505-517)."""
    assert y.ndim == 5 and y.shape[-1] == 3
    comps = [y[:, :, :, :, a] for a in range(3)]
    means = [np.mean(c, axis=-1)[:, :, :, np.newaxis] for c in comps]
    variances = [np.var(c, axis=-1)[:, :, :, np.newaxis] for c in comps]
    return np.concatenate(means + variances, axis=-1)


# a6: Keras mean_squared_error + sample mean  (mycode/cost.py:20-22)
def mse(y_true, y_pred):
    return float(np.mean(np.mean((y_pred.astype(np.float64) - y_true.astype(np.float64)) ** 2, axis=-1)))


# --------------------------------------------------------------------------------------
# Keras default initialisers (glorot_uniform kernel, orthogonal recurrent, unit_forget_bias)
# --------------------------------------------------------------------------------------
def _glorot_uniform(rng, fan_in, fan_out, dtype):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, (fan_in, fan_out)).astype(dtype)


def _orthogonal(rng, rows, cols, dtype):
    a = rng.standard_normal((rows, cols))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    q = u if u.shape == (rows, cols) else vt
    return q.astype(dtype)


def init_lstm(rng, F, H, dtype=np.float32):
    K = _glorot_uniform(rng, F, 4 * H, dtype)
    R = _orthogonal(rng, H, 4 * H, dtype)
    b = np.zeros(4 * H, dtype)
    b[H:2 * H] = 1  # unit_forget_bias
    return K, R, b


def init_seq2seq(seed, F_enc=90, F_dec=6, H=256, dtype=np.float32, bias_noise=0.0):
    rng = np.random.default_rng(seed)
    w = {}
    w["enc_K"], w["enc_R"], w["enc_b"] = init_lstm(rng, F_enc, H, dtype)
    w["dec_K"], w["dec_R"], w["dec_b"] = init_lstm(rng, F_dec, H, dtype)
    w["dense_W"] = _glorot_uniform(rng, H, F_dec, dtype)
    w["dense_b"] = np.zeros(F_dec, dtype)
    if bias_noise:
        for k in ("enc_b", "dec_b", "dense_b"):
            w[k] = (w[k] + bias_noise * rng.standard_normal(w[k].shape)).astype(dtype)
    return w


def init_others_mixing(seed, F_enc=90, F_dec=6, H=256, num_user=34, dtype=np.float32, bias_noise=0.0):
    rng = np.random.default_rng(seed)
    w = {}
    for name, F in (("enc1", F_enc), ("enc2", H), ("dec1", F_dec), ("dec2", H)):
        w[name + "_K"], w[name + "_R"], w[name + "_b"] = init_lstm(rng, F, H, dtype)
    w["dense_W"] = _glorot_uniform(rng, H, F_dec, dtype)
    w["dense_b"] = np.zeros(F_dec, dtype)
    w["mix_W"] = _glorot_uniform(rng, num_user * F_dec, F_dec, dtype)
    w["mix_b"] = np.zeros(F_dec, dtype)
    if bias_noise:
        for k in list(w):
            if k.endswith("_b"):
                w[k] = (w[k] + bias_noise * rng.standard_normal(w[k].shape)).astype(dtype)
    return w


# --------------------------------------------------------------------------------------
# Synthetic trajectories of SURVEY.md section 8(d)
# --------------------------------------------------------------------------------------
def synthetic_xyz(rng, B, T, fps=30, dtype=np.float32):
    """(B, T, 3*fps) interleaved xyz per frame, smooth unit-sphere trajectories."""
    n = T * fps
    yaw = rng.uniform(-np.pi, np.pi, (B, 1)) + np.cumsum(rng.normal(0, 0.02, (B, n)), axis=1)
    pitch = np.clip(rng.normal(0, 0.3, (B, 1)) + np.cumsum(rng.normal(0, 0.01, (B, n)), axis=1), -np.pi / 2, np.pi / 2)
    xyz = np.stack([np.cos(pitch) * np.cos(yaw), np.cos(pitch) * np.sin(yaw), np.sin(pitch)], axis=-1)
    return xyz.reshape(B, T, fps * 3).astype(dtype)


def synthetic_batch(seed, B, T_in, T_out, num_others=0, fps=30, dtype=np.float32):
    """enc_in (B,T_in,90), dec_in0 (B,1,6), target (B,T_out,6) [, others (B,T_out,U-1,6)]."""
    rng = np.random.default_rng(seed)
    traj = synthetic_xyz(rng, B, T_in + T_out, fps, np.float64)
    enc_in = traj[:, :T_in]
    target = meanvar_xyz(traj[:, T_in:])
    dec_in0 = meanvar_xyz(enc_in[:, -1:])
    out = [enc_in.astype(dtype), dec_in0.astype(dtype), target.astype(dtype)]
    if num_others:
        oth = synthetic_xyz(rng, B * num_others, T_out, fps, np.float64).reshape(B, num_others, T_out, fps, 3)
        out.append(meanvar_xyz_oth(oth.transpose(0, 2, 1, 3, 4)).astype(dtype))
    return tuple(out)


# --------------------------------------------------------------------------------------
# a6: training-graph backward (BPTT) and the Keras optimizers
# model.compile(optimizer='Adam', loss='mean_squared_error') - mycode/FoV_seq2seq.py:103
# The arithmetic is Keras/TensorFlow autodiff in the reference; this is the closed-form
# restatement, checked against torch.autograd and finite differences in tests/test_oracle.py.
# --------------------------------------------------------------------------------------
def _rec_act_grad(a, act):
    """d s(z)/dz expressed through the activation value a = s(z)."""
    if act_code(act) == ACT_HARD_SIGMOID:
        return np.where((a > 0) & (a < 1), a.dtype.type(0.2), a.dtype.type(0))
    return a * (1 - a)


def lstm_layer_train(x, K, R, b, h0=None, c0=None, act="sigmoid"):
    """Forward that also returns the reserve (i,f,g,o,c per step) the backward needs."""
    B, T, _ = x.shape
    H = R.shape[0]
    h = np.zeros((B, H), x.dtype) if h0 is None else h0.astype(x.dtype)
    c = np.zeros((B, H), x.dtype) if c0 is None else c0.astype(x.dtype)
    s = _rec_act(act)
    hs = np.empty((B, T, H), x.dtype)
    res = np.empty((B, T, 5, H), x.dtype)
    for t in range(T):
        z = x[:, t] @ K + b + h @ R
        i, f, g, o = s(z[:, :H]), s(z[:, H:2 * H]), np.tanh(z[:, 2 * H:3 * H]), s(z[:, 3 * H:])
        c = f * c + i * g
        h = o * np.tanh(c)
        hs[:, t] = h
        res[:, t, 0], res[:, t, 1], res[:, t, 2], res[:, t, 3], res[:, t, 4] = i, f, g, o, c
    return hs, h, c, res


def lstm_layer_backward(x, K, R, h0, c0, hs, res, dhs=None, dhT=None, dcT=None, act="sigmoid"):
    """BPTT of one LSTM layer.  Returns dict(dx, dK, dR, db, dh0, dc0, dz)."""
    B, T, F = x.shape
    H = R.shape[0]
    dt = x.dtype
    h0 = np.zeros((B, H), dt) if h0 is None else h0
    c0 = np.zeros((B, H), dt) if c0 is None else c0
    dh = np.zeros((B, H), dt) if dhT is None else dhT.astype(dt).copy()
    dc = np.zeros((B, H), dt) if dcT is None else dcT.astype(dt).copy()
    dz_all = np.empty((B, T, 4 * H), dt)
    for t in range(T - 1, -1, -1):
        i, f, g, o, c = (res[:, t, q] for q in range(5))
        c_prev = res[:, t - 1, 4] if t > 0 else c0
        if dhs is not None:
            dh = dh + dhs[:, t]
        tc = np.tanh(c)
        do = dh * tc
        dc = dc + dh * o * (1 - tc * tc)
        dz = np.concatenate([dc * g * _rec_act_grad(i, act), dc * c_prev * _rec_act_grad(f, act),
                             dc * i * (1 - g * g), do * _rec_act_grad(o, act)], axis=1)
        dz_all[:, t] = dz
        dc = dc * f
        dh = dz @ R.T
    hprev = np.concatenate([h0[:, None], hs[:, :-1]], axis=1)
    dz2 = dz_all.reshape(B * T, 4 * H)
    return {"dx": (dz2 @ K.T).reshape(B, T, F), "dK": x.reshape(B * T, F).T @ dz2,
            "dR": hprev.reshape(B * T, H).T @ dz2, "db": dz2.sum(axis=0), "dh0": dh, "dc0": dc, "dz": dz_all}


def seq2seq_loss_and_grads(enc_in, dec_in, target, w, act="sigmoid"):
    """Teacher-forced graph (FoV_seq2seq.py:82-103): loss = Keras mean_squared_error averaged over
    every sample and step; returns (loss, grads dict keyed like the weights, prediction)."""
    ehs, eh, ec, eres = lstm_layer_train(enc_in, w["enc_K"], w["enc_R"], w["enc_b"], act=act)
    dhs_, _, _, dres = lstm_layer_train(dec_in, w["dec_K"], w["dec_R"], w["dec_b"], eh, ec, act=act)
    B, T, H = dhs_.shape
    y = np.tanh(dhs_.reshape(B * T, H) @ w["dense_W"] + w["dense_b"])
    t2 = target.reshape(B * T, -1)
    loss = float(np.mean((y - t2) ** 2))
    dy = 2 * (y - t2) / y.size
    dpre = dy * (1 - y * y)
    g = {"dense_W": dhs_.reshape(B * T, H).T @ dpre, "dense_b": dpre.sum(axis=0)}
    d_dec_hs = (dpre @ w["dense_W"].T).reshape(B, T, H)
    bd = lstm_layer_backward(dec_in, w["dec_K"], w["dec_R"], eh, ec, dhs_, dres, dhs=d_dec_hs, act=act)
    g["dec_K"], g["dec_R"], g["dec_b"] = bd["dK"], bd["dR"], bd["db"]
    be = lstm_layer_backward(enc_in, w["enc_K"], w["enc_R"], None, None, ehs, eres, dhT=bd["dh0"], dcT=bd["dc0"], act=act)
    g["enc_K"], g["enc_R"], g["enc_b"] = be["dK"], be["dR"], be["db"]
    return loss, g, y.reshape(B, T, -1)


def adam_step(p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras-2.2 Adam.get_updates: lr_t = lr*sqrt(1-b2^t)/(1-b1^t); p -= lr_t * m / (sqrt(v) + eps).
    `t` is the 1-based step count.  In place on (p, m, v)."""
    lr_t = lr * np.sqrt(1 - beta2 ** t) / (1 - beta1 ** t)
    m[...] = beta1 * m + (1 - beta1) * g
    v[...] = beta2 * v + (1 - beta2) * g * g
    p[...] = p - lr_t * m / (np.sqrt(v) + eps)


def rmsprop_step(p, g, a, lr=1e-3, rho=0.9, eps=1e-7):
    """Keras-2.2 RMSprop: a = rho*a + (1-rho)*g^2; p -= lr * g / (sqrt(a) + eps).  In place."""
    a[...] = rho * a + (1 - rho) * g * g
    p[...] = p - lr * g / (np.sqrt(a) + eps)


# --------------------------------------------------------------------------------------
# a10: tf.contrib.rnn.LSTMCell / MultiRNNCell / tf.nn.dynamic_rnn as used by mycode/lstm.py:218-240
# (TensorFlow 1.x contrib, not in the repo; restated from its published definition: one fused
# kernel W:(F+H,4H) applied to [x, h], gate column order i, j(=g), f, o, plain sigmoid,
# forget_bias added inside the cell, state tuple (c, h)).
# --------------------------------------------------------------------------------------
def tf_lstm_cell_step(x, c, h, W, b, forget_bias=1.0):
    H = h.shape[1]
    z = np.concatenate([x, h], axis=1) @ W + b
    i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
    c_new = sigmoid(f + forget_bias) * c + sigmoid(i) * np.tanh(j)
    h_new = sigmoid(o) * np.tanh(c_new)
    return c_new.astype(x.dtype), h_new.astype(x.dtype)


def tf_dynamic_rnn(x, cells, init_state=None, forget_bias=1.0):
    """cells: list of (W, b) per layer; init_state: (L,2,B,H) with [l,0]=c, [l,1]=h (lstm.py:128-132).
    Returns (states_series (B,T,H) of the top layer, current_state (L,2,B,H))."""
    B, T, _ = x.shape
    L = len(cells)
    H = cells[0][1].shape[0] // 4
    st = np.zeros((L, 2, B, H), x.dtype) if init_state is None else init_state.astype(x.dtype).copy()
    out = np.empty((B, T, H), x.dtype)
    for t in range(T):
        inp = x[:, t]
        for l, (W, b) in enumerate(cells):
            st[l, 0], st[l, 1] = tf_lstm_cell_step(inp, st[l, 0], st[l, 1], W, b, forget_bias)
            inp = st[l, 1]
        out[:, t] = inp
    return out, st


def tf_mean_var_head(h, head):
    """_pred_mean_var_xyz2_new (lstm.py:321-337): relu -> tanh for the means, relu -> linear -> exp for the variances."""
    mu = np.tanh(np.maximum(h @ head["mu_W1"] + head["mu_b1"], 0) @ head["mu_W2"] + head["mu_b2"])
    var = np.exp(np.maximum(h @ head["var_W1"] + head["var_b1"], 0) @ head["var_W2"] + head["var_b2"])
    return mu, var


def tf_lstm_sampled_rollout(x, cells, head, init_state, noise, forget_bias=1.0):
    """Test-time loop of lstm.py:714-740 (cfg.use_xyz, cfg.predict_mean_var): every step runs the stack over the current
    window from the state the PREVIOUS run returned, predicts (mu, var), draws one second around it
    (utility.generate_fake_batch_numpy, utility.py:73-80: normal(mu, sqrt(var)), here mu + sqrt(var) * noise[k], frames
    interleaved x,y,z by np.stack axis=-1) and shifts it into the window.  -> (mus (P,B,3), vars (P,B,3), state)."""
    win, st = x.copy(), init_state
    B, fps = x.shape[0], x.shape[2] // 3
    mus, vs = [], []
    for k in range(noise.shape[0]):
        _, st = tf_dynamic_rnn(win, cells, st, forget_bias)
        mu, var = tf_mean_var_head(st[-1, 1], head)
        mus.append(mu); vs.append(var)
        smp = (mu[:, None, :] + np.sqrt(var)[:, None, :] * noise[k].reshape(B, fps, 3)).reshape(B, 1, 3 * fps)
        win = np.concatenate([win[:, 1:], smp], axis=1)
    return np.stack(mus), np.stack(vs), st


# --------------------------------------------------------------------------------------
# a8/a9: ConvLSTM2D seq2seq (mycode/convlstm_seq2seq.py:100-282).  Keras-2.2 ConvLSTM2DCell restated:
#   x_g = conv2d(x, K_g, 'same') + b_g ;  h_g = conv2d(h, R_g, 'same')        (cross-correlation, NHWC)
#   i = s(x_i+h_i); f = s(x_f+h_f); c' = f*c + i*tanh(x_c+h_c); o = s(x_o+h_o); h' = o*tanh(c')
# kernel K:(kh,kw,C,4F), recurrent R:(kh,kw,F,4F), bias (4F), channel blocks i,f,c,o; s = hard_sigmoid
# by default (the reference never overrides it).  Dropout 0.3 on the inputs acts in training only.
# --------------------------------------------------------------------------------------
def conv2d_same(x, w, b=None, dilation=1):
    """x:(B,H,W,C), w:(kh,kw,C,N) -> (B,H,W,N); zero 'same' padding, stride 1, no kernel flip.  dilation = Keras
    `dilation_rate` (odd kernels: tap (i,j) reads the pixel (i - kh//2, j - kw//2) * dilation away)."""
    B, H, W, C = x.shape
    kh, kw, _, N = w.shape
    d = int(dilation)
    ph, pw = d * ((kh - 1) // 2), d * ((kw - 1) // 2)
    xp = np.zeros((B, H + d * (kh - 1), W + d * (kw - 1), C), x.dtype)
    xp[:, ph:ph + H, pw:pw + W] = x
    out = np.zeros((B, H, W, N), x.dtype)
    for dy in range(kh):
        for dx in range(kw):
            out += xp[:, d * dy:d * dy + H, d * dx:d * dx + W] @ w[dy, dx]
    if b is not None:
        out = out + b
    return out.astype(x.dtype)


def convlstm2d_step(x, h, c, K, R, b, act="hard_sigmoid", dilation=1):
    """Keras 2.2 ConvLSTM2DCell.call: `dilation_rate` reaches input_conv only; recurrent_conv is never dilated."""
    F = R.shape[2]
    z = conv2d_same(x, K, b, dilation) + conv2d_same(h, R)
    s = _rec_act(act)
    i, f, g, o = s(z[..., :F]), s(z[..., F:2 * F]), np.tanh(z[..., 2 * F:3 * F]), s(z[..., 3 * F:])
    c_new = f * c + i * g
    return (o * np.tanh(c_new)).astype(x.dtype), c_new.astype(x.dtype)


def convlstm2d_layer(x, K, R, b, h0=None, c0=None, act="hard_sigmoid", dilation=1):
    """x:(B,T,H,W,C) -> (hs:(B,T,H,W,F), h_T, c_T)."""
    B, T, H, W, _ = x.shape
    F = R.shape[2]
    h = np.zeros((B, H, W, F), x.dtype) if h0 is None else h0
    c = np.zeros((B, H, W, F), x.dtype) if c0 is None else c0
    hs = np.empty((B, T, H, W, F), x.dtype)
    for t in range(T):
        h, c = convlstm2d_step(x[:, t], h, c, K, R, b, act, dilation)
        hs[:, t] = h
    return hs, h, c


def softmax_last(x):
    e = np.exp(x - x.max(axis=-1, keepdims=True))
    return (e / e.sum(axis=-1, keepdims=True)).astype(x.dtype)


def convlstm_seq2seq_forward(enc_in, dec_in0, w, T_out, head="conv2d", act="hard_sigmoid", dilation=1):
    """3-layer ConvLSTM2D encoder, mirrored decoder unrolled T_out times with state hand-off, channel
    concat of the three layer outputs, head, output fed back (convlstm_seq2seq.py:100-126,146-165,209-282).
      head 'conv2d' (cfg.use_one_hot): Conv2D -> Conv2D -> Conv2D (relu each) + channel softmax
      head 'conv1d' (xyz mode, H == 1): Conv1D k=7 relu, relu, softmax over the 3 output channels
      head 'dense'  (cfg.predict_mean_var + cfg.input_mean_var, 1x1 maps of 6 channels): Flatten + Dense(6)
    dilation = cfg.dilation_rate (config.py:105), passed to the six ConvLSTM2D layers (:102,110,120,148,155,162), not to the head.
    enc_in:(B,T_in,H,W,C)  dec_in0:(B,1,H,W,C)  ->  (B,T_out,H,W,Cout)   ('dense': (B,T_out,6))."""
    x = enc_in
    states = []
    for l in range(3):
        x, h, c = convlstm2d_layer(x, w["enc%d_K" % l], w["enc%d_R" % l], w["enc%d_b" % l], act=act, dilation=dilation)
        states.append((h, c))
    inp = dec_in0[:, 0]
    outs = []
    for _ in range(T_out):
        feats = []
        cur = inp
        for l in range(3):
            h, c = convlstm2d_step(cur, states[l][0], states[l][1], w["dec%d_K" % l], w["dec%d_R" % l], w["dec%d_b" % l], act,
                                   dilation)
            states[l] = (h, c)
            feats.append(h)
            cur = h
        y = np.concatenate(feats, axis=-1)
        if head == "dense":      # cfg.predict_mean_var: Flatten + Dense(6, linear) (convlstm_seq2seq.py:171,225-227)
            y = y.reshape(y.shape[0], -1) @ w["head0_W"] + w["head0_b"]
            outs.append(y)
            # fed back as a 1x1 map of 6 channels (cfg.input_mean_var, :272-273)
            inp = y.reshape(y.shape[0], 1, 1, -1)
            continue
        y = np.maximum(conv2d_same(y, w["head0_W"], w["head0_b"]), 0)
        y = np.maximum(conv2d_same(y, w["head1_W"], w["head1_b"]), 0)
        y = conv2d_same(y, w["head2_W"], w["head2_b"])
        if head == "conv2d":
            y = softmax_last(np.maximum(y, 0))        # relu in the layer, then Softmax(axis=-1)
        else:
            y = softmax_last(y)                       # Conv1D(..., activation='softmax')
        outs.append(y)
        inp = y
    return np.stack(outs, axis=1)


def init_convlstm_seq2seq(seed, C=30, latent_dim=16, k=5, head="conv2d", head_filters=(512, 1024), dtype=np.float32,
                          map_hw=(1, 1)):
    """Keras initialisers: glorot_uniform kernels (fan = receptive field x channels), orthogonal recurrent
    kernels (flattened), unit forget bias."""
    rng = np.random.default_rng(seed)
    filters = (latent_dim * 2, latent_dim, latent_dim // 2)
    kh, kw = (k, k) if head == "conv2d" else (k, k)
    w = {}

    def glorot(shape):
        rf = int(np.prod(shape[:-2]))
        lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        return rng.uniform(-lim, lim, shape).astype(dtype)

    for part in ("enc", "dec"):
        cin = C
        for l, F in enumerate(filters):
            w["%s%d_K" % (part, l)] = glorot((kh, kw, cin, 4 * F))
            w["%s%d_R" % (part, l)] = _orthogonal(rng, kh * kw * F, 4 * F, dtype).reshape(kh, kw, F, 4 * F)
            b = (0.05 * rng.standard_normal(4 * F)).astype(dtype)
            b[F:2 * F] += 1
            w["%s%d_b" % (part, l)] = b
            cin = F
    cat = sum(filters)
    if head == "dense":
        n_in = map_hw[0] * map_hw[1] * cat
        lim = np.sqrt(6.0 / (n_in + 6))
        w["head0_W"] = rng.uniform(-lim, lim, (n_in, 6)).astype(dtype)
        w["head0_b"] = (0.05 * rng.standard_normal(6)).astype(dtype)
        return w
    hk = (k, k) if head == "conv2d" else (1, 7)
    chans = (cat,) + tuple(head_filters) + ((C,) if head == "conv2d" else (3,))
    for i in range(3):
        w["head%d_W" % i] = glorot(hk + (chans[i], chans[i + 1]))
        w["head%d_b" % i] = (0.05 * rng.standard_normal(chans[i + 1])).astype(dtype)
    return w


# --------------------------------------------------------------------------------------
# the consumer of the path's output - FoV hit rate per predicted second
#   xyz2thetaphi           mycode/dataIO.py:77-82
#   boundary_cases         mycode/baseline_knn_mean.py:78-85
#   bbox_overlaps_hit_rate mycode/baseline_knn_mean.py:62-82 via get_iou_or_hitrate :48-60
# --------------------------------------------------------------------------------------
def xyz2thetaphi(x, y, z):
    theta = np.mod(np.arctan2(y, x), 2 * np.pi) - np.pi
    phi = np.mod(np.arctan2(z, np.sqrt(x ** 2 + y ** 2)) + np.pi / 2, np.pi)
    return theta, phi


def fov_hit_rate(pred_xyz, gt_xyz, span_deg=120.0, gt_span_deg=120.0):
    """pred_xyz, gt_xyz: (..., 3) FoV-centre unit vectors (per-second means).  Returns the hit rate
    = area(pred box ∩ gt box) / area(gt box) of the two (theta,phi) boxes of the given angular spans,
    with the reference's +-2pi wrap fix when the two centres straddle the theta seam."""
    pt, pp = xyz2thetaphi(pred_xyz[..., 0], pred_xyz[..., 1], pred_xyz[..., 2])
    gt, gp = xyz2thetaphi(gt_xyz[..., 0], gt_xyz[..., 1], gt_xyz[..., 2])
    pt, gt = pt.copy(), gt.copy()
    c1 = (gt > 2 / 3.0 * np.pi) & (pt < -2 / 3.0 * np.pi)
    pt[c1] += 2 * np.pi
    c2 = (gt < -2 / 3.0 * np.pi) & (pt > 2 / 3.0 * np.pi)
    gt[c2] += 2 * np.pi
    s, gs = span_deg / 180.0 * np.pi, gt_span_deg / 180.0 * np.pi
    iw = np.minimum(pt + s / 2, gt + gs / 2) - np.maximum(pt - s / 2, gt - gs / 2)
    ih = np.minimum(pp + s / 2, gp + gs / 2) - np.maximum(pp - s / 2, gp - gs / 2)
    return np.where((iw > 0) & (ih > 0), iw * ih / (gs * gs), 0.0)


# --------------------------------------------------------------------------------------
# a10, the branch mycode/config.py:8,69,71 actually selects (use_xyz, predict_mean_var = False, use_GMM = True):
# the mixture-density head _GMM_3dgassian (lstm.py:377-400) with costfunc.mixture_3d_gaussian_loss
# (cost.py:486-549, density cost.py:352-383, PSD repair cost.py:335-348), and the third branch (predict raw,
# lstm.py:147-174,486-508) with costfunc.pred_raw_loss_tf (cost.py:634-641).  TensorFlow-1.x pieces restated from
# their published definitions (tf.contrib.layers.fully_connected = x W + b then the activation;
# tf.contrib.distributions.MultivariateNormalFullCovariance.prob = the N(mu, Sigma) density; tf.self_adjoint_eig
# = ascending eigenvalues): parity unpinned like the rest of the arithmetic; the density is cross-checked
# against scipy.stats.multivariate_normal in tests/test_oracle.py.
#
# Reference quirks kept on purpose (each one changes the numbers):
#   * tf.layers.dropout(internal, rate=0.2) is called WITHOUT training=True (lstm.py:380,382): it is the
#     identity in every run of the script.  `masks` exists for a caller who wants the two dropouts.
#   * mixture_3d_gaussian_loss never multiplies by mixture_pi (cost.py:532-538: `gaussian` is the bare
#     density, the expanded mixture_pi is dropped; the 2-D loss at cost.py:466 does multiply): the 20
#     softmax weights get no gradient.  weight_by_pi=True is the textbook mixture.
#   * process_in_seconds scores only second 0 of y (cost.py:502-505) and divides by
#     batch_size * running_length * fps whatever the number of frames summed.
# --------------------------------------------------------------------------------------
GMM_KEYS = ("fc1_W", "fc1_b", "fc2_W", "fc2_b", "fc3_W", "fc3_b", "fc4_W", "fc4_b")


def tf_gmm3d_head(h, head, masks=None, n_mix=20):
    """_GMM_3dgassian: h (B,H) -> 64 relu -> 128 relu -> 256 relu -> 10*n_mix linear, split
    [n pi-logits | 3n means | 3n log-sigmas | 3n atanh-rhos].  -> (pi, us, sigmas, rhos), activations (a1,a2,a3)."""
    a1 = np.maximum(h @ head["fc1_W"] + head["fc1_b"], 0)
    if masks is not None and masks[0] is not None:
        a1 = a1 * masks[0]
    a2 = np.maximum(a1 @ head["fc2_W"] + head["fc2_b"], 0)
    if masks is not None and masks[1] is not None:
        a2 = a2 * masks[1]
    a3 = np.maximum(a2 @ head["fc3_W"] + head["fc3_b"], 0)
    pred = a3 @ head["fc4_W"] + head["fc4_b"]
    n = n_mix
    e = np.exp(pred[:, :n])                       # lstm.py:394-396: exp / sum, no max subtraction
    pi = e / e.sum(1, keepdims=True)
    return (pi, pred[:, n:4 * n], np.exp(pred[:, 4 * n:7 * n]), np.tanh(pred[:, 7 * n:10 * n])), (a1, a2, a3)


def gmm3d_covariance(sig, rho):
    """(...,3) sigmas and (...,3) rhos (rho12, rho13, rho23) -> covariance (...,3,3) after the reference's repair
    (cost.py:335-348): if the smallest eigenvalue is negative, subtract 10 * min_eig * I."""
    s1, s2, s3 = sig[..., 0], sig[..., 1], sig[..., 2]
    r12, r13, r23 = rho[..., 0], rho[..., 1], rho[..., 2]
    S = np.stack([np.stack([s1 * s1, r12 * s1 * s2, r13 * s1 * s3], -1),
                  np.stack([r12 * s1 * s2, s2 * s2, r23 * s2 * s3], -1),
                  np.stack([r13 * s1 * s3, r23 * s2 * s3, s3 * s3], -1)], -2)
    lam = np.linalg.eigvalsh(S)[..., 0]
    shift = np.where(lam < 0, -10.0 * lam, 0.0)
    return S + shift[..., None, None] * np.eye(3)


def mvn3_prob(y, mu, cov):
    """N(y; mu, cov) for y (...,3), mu (...,3), cov (...,3,3)."""
    d = y - mu
    sol = np.linalg.solve(cov, d[..., None])[..., 0]
    q = (d * sol).sum(-1)
    return np.exp(-0.5 * q) / np.sqrt((2 * np.pi) ** 3 * np.linalg.det(cov))


def mixture_3d_gaussian_loss(y_true, params, batch_size, running_length, fps=30, process_in_seconds=True,
                             weight_by_pi=False):
    """cost.py:486-549.  y_true (B,T_y,3*fps) [process_in_seconds: second 0 only] or (B,T,3) [per frame]."""
    pi, us, sig, rho = params
    B, n = pi.shape
    pts = y_true[:, 0].reshape(B, -1, 3) if process_in_seconds else y_true      # (B,P,3)
    mu = us.reshape(B, n, 3)                     # us[:, 0::3], [1::3], [2::3] = x, y, z of mixture m
    cov = gmm3d_covariance(sig.reshape(B, n, 3), rho.reshape(B, n, 3))          # (B,n,3,3)
    p = mvn3_prob(pts[:, None, :, :], mu[:, :, None, :], cov[:, :, None, :, :])  # (B,n,P)
    if weight_by_pi:
        p = p * pi[:, :, None]
    loss = -np.log(p.sum(1) + 1e-20).sum()
    return loss / batch_size / (running_length * fps if process_in_seconds else running_length)


def sample_mixture_3d(params, u, z):
    """One draw per frame from the 3-D mixture, what utility.sample_mixture_3D's docstring promises ("randomly one sample
    from 3D GMM", utility.py:178-208).  The committed function cannot run (pdb.set_trace(), a module-level batch_size
    that does not exist, 2-D indexing of 3-D parameters, a (B,1,2) result fed to a (B,1,90) placeholder), so the
    intent is restated: frame f of row b picks component m = first index with cumsum(pi)[m] > u[b,f] and draws
    mu_m + L_m z[b,f], L_m the Cholesky factor of the repaired covariance.  u (B,P) uniform, z (B,P,3) normal
    -> (B,1,3*P) interleaved x,y,z."""
    pi, us, sig, rho = params
    B, n = pi.shape
    P = u.shape[1]
    mu = us.reshape(B, n, 3)
    L = np.linalg.cholesky(gmm3d_covariance(sig.reshape(B, n, 3), rho.reshape(B, n, 3)))
    cum = np.cumsum(pi, 1)
    out = np.empty((B, P, 3), pi.dtype)
    for b in range(B):
        for f in range(P):
            m = min(int(np.searchsorted(cum[b], u[b, f], side="right")), n - 1)
            out[b, f] = mu[b, m] + L[b, m] @ z[b, f]
    return out.reshape(B, 1, 3 * P)


def tf_lstm_gmm_rollout(x, cells, head, init_state, u, z, forget_bias=1.0):
    """GMM test loop (lstm.py:690-698,735-745,820-825): only the LAST second is fed (x (B,1,90)), the state is carried,
    every step predicts the mixture and feeds back one sampled second.  u (P,B,fps), z (P,B,fps,3).
    -> (samples (P,B,1,90), final state)."""
    win, st = x.copy(), init_state
    outs = []
    for k in range(u.shape[0]):
        _, st = tf_dynamic_rnn(win, cells, st, forget_bias)
        params, _ = tf_gmm3d_head(st[-1, 1], head)
        smp = sample_mixture_3d(params, u[k], z[k])
        outs.append(smp)
        win = np.concatenate([win[:, 1:], smp], axis=1)
    return np.stack(outs), st


RAW_KEYS = ("conv1_W", "conv1_b", "conv2_W", "conv2_b", "conv3_W", "conv3_b")


def tf_raw_head(h, head):
    """pred_cnn_model_fn (lstm.py:147-174): three tf.layers.conv1d (k = 5, 'same', relu, relu, tanh; 128, 256, 3*fps
    filters) on h expanded to ONE time step.  With one step and zero padding only the centre tap k//2 of each kernel
    (k, C_in, C_out) meets data ("equivalent to 3 fc layers", :151).  -> (B,1,3*fps), activations."""
    c = head["conv1_W"].shape[0] // 2
    a1 = np.maximum(h @ head["conv1_W"][c] + head["conv1_b"], 0)
    a2 = np.maximum(a1 @ head["conv2_W"][c] + head["conv2_b"], 0)
    out = np.tanh(a2 @ head["conv3_W"][c] + head["conv3_b"])
    return out[:, None, :], (a1, a2)


def total_variation_loss_tf(pred):
    """cost.py:608-618 on (B,T,3*fps): differences along axis ONE (time steps), not along the frames of a second."""
    x, y, z = pred[:, :, 0::3], pred[:, :, 1::3], pred[:, :, 2::3]
    d = (x[:, :-1] - x[:, 1:]) ** 2 + (y[:, :-1] - y[:, 1:]) ** 2 + (z[:, :-1] - z[:, 1:]) ** 2
    return (d ** 1.25).sum()


def sum1reg_tf(pred):
    """cost.py:622-631: sum (x^2 + y^2 + z^2 - 1)^2."""
    x, y, z = pred[:, :, 0::3], pred[:, :, 1::3], pred[:, :, 2::3]
    return ((x * x + y * y + z * z - 1) ** 2).sum()


def pred_raw_loss_tf(this_y, pred, use_reg=False):
    """cost.py:634-641: tf.losses.mean_squared_error (mean over every element) + 0.1 * TV (+ 0.1 * sum-to-one).  In
    lstm.py every call passes ONE time step (:493-508), so the TV term slices an empty range and is exactly 0."""
    loss = ((this_y - pred) ** 2).mean() + 0.1 * total_variation_loss_tf(pred)
    if use_reg:
        loss = loss + 0.1 * sum1reg_tf(pred)
    return loss


def tf_lstm_raw_refeed_loss(x, y, cells, head, init_state, use_reg=False, forget_bias=1.0):
    """Training graph of the raw branch, predict_len > 1 (lstm.py:486-508): second k+1 is scored after re-running the
    stack, from the same fed state, on the window shifted by one second whose last slot is prediction k itself.
    -> (loss, predictions (P,B,1,3*fps))."""
    win, preds, loss = x, [], 0.0
    for k in range(y.shape[1]):
        _, st = tf_dynamic_rnn(win, cells, init_state, forget_bias)
        p, _ = tf_raw_head(st[-1, 1], head)
        loss = loss + pred_raw_loss_tf(y[:, k:k + 1], p, use_reg)
        preds.append(p)
        win = np.concatenate([win[:, 1:], p], axis=1)
    return loss, np.stack(preds)


# --------------------------------------------------------------------------------------
# Second half of mycode/FoV_seq2seq_no_teac_forc.py (:420-486): the unrolled no-teacher-forcing decoder whose Dense head
# also sees the OTHER users' future through a ConvLSTM2D branch.
#   encoder LSTM(latent_dim) -> (h, c) seed the decoder (:431-434; this half has no decoder_no_init_state switch)
#   ConvLSTM2D(filters = latent_dim, kernel_size = (num_user-1, 3), 'same', return_sequences) over the others' future
#       (B, T_out, num_user-1, fps, 3) from zero state (:438-448) - it does not depend on the decoder
#   step t: d_t = LSTM(x_t; h, c);  s_t = Dense(latent_dim, linear)(Flatten(convlstm output t))  (:468-470)
#           y_t = Dense(6, tanh)(concat[d_t, s_t]) (:471-472);  x_{t+1} = y_t (:476)
# --------------------------------------------------------------------------------------
OTHERS_FUTURE_ORDER = ("enc_K", "enc_R", "enc_b", "oth_K", "oth_R", "oth_b", "flat_W", "flat_b", "dec_K", "dec_R", "dec_b",
                       "dense_W", "dense_b")


def others_future_convlstm_forward(enc_in, others_fut, dec_in0, w, act="sigmoid", conv_act="hard_sigmoid"):
    """enc_in (B,T_in,F), others_fut (B,T_out,U-1,fps,3), dec_in0 (B,1,O) -> (B,T_out,O)."""
    _, h, c = lstm_layer(enc_in, w["enc_K"], w["enc_R"], w["enc_b"], act=act)
    hs, _, _ = convlstm2d_layer(others_fut, w["oth_K"], w["oth_R"], w["oth_b"], act=conv_act)     # (B,T,U-1,fps,H)
    B, T = hs.shape[:2]
    s = hs.reshape(B, T, -1) @ w["flat_W"] + w["flat_b"]          # Keras Flatten of a channels-last map: (row, column, filter)
    x, out = dec_in0[:, 0].astype(enc_in.dtype), []
    for t in range(T):
        h, c = lstm_step(x, h, c, w["dec_K"], w["dec_R"], w["dec_b"], act)
        x = np.tanh(np.concatenate([h, s[:, t]], axis=1) @ w["dense_W"] + w["dense_b"])
        out.append(x)
    return np.stack(out, axis=1)


def init_others_future_convlstm(seed, F_enc=90, F_dec=6, H=64, num_user=34, fps=30, dtype=np.float32, bias_noise=0.0):
    rng = np.random.default_rng(seed)
    w = {}
    w["enc_K"], w["enc_R"], w["enc_b"] = init_lstm(rng, F_enc, H, dtype)
    w["dec_K"], w["dec_R"], w["dec_b"] = init_lstm(rng, F_dec, H, dtype)
    kh, kw = num_user - 1, 3

    def glorot(shape):
        rf = int(np.prod(shape[:-2]))
        lim = np.sqrt(6.0 / (rf * shape[-2] + rf * shape[-1]))
        return rng.uniform(-lim, lim, shape).astype(dtype)

    w["oth_K"] = glorot((kh, kw, 3, 4 * H))
    w["oth_R"] = _orthogonal(rng, kh * kw * H, 4 * H, dtype).reshape(kh, kw, H, 4 * H)
    b = np.zeros(4 * H, dtype)
    b[H:2 * H] = 1
    w["oth_b"] = b
    n_flat = (num_user - 1) * fps * H
    w["flat_W"] = _glorot_uniform(rng, n_flat, H, dtype)
    w["flat_b"] = np.zeros(H, dtype)
    w["dense_W"] = _glorot_uniform(rng, 2 * H, F_dec, dtype)
    w["dense_b"] = np.zeros(F_dec, dtype)
    if bias_noise:
        for k in ("enc_b", "dec_b", "oth_b", "flat_b", "dense_b"):
            w[k] = (w[k] + bias_noise * rng.standard_normal(w[k].shape)).astype(dtype)
    return w
