"""Keras-shaped model objects for the hot path, backed by libfov360_hip.so.

Mirrors what the reference's model scripts do with ``keras.models.Model``:
    mycode/FoV_seq2seq.py:82-101     training graph        -> Seq2SeqLSTM.predict / fit
    mycode/FoV_seq2seq.py:137-148    sampling models       -> .encoder_model / .decoder_model
    mycode/FoV_seq2seq.py:154-178    decode_sequence_fov   -> .decode_sequence (one fused device call)
Inputs and outputs are NumPy arrays in the reference's positional order and shapes; tensors are
staged through torch (device memory and streams only).  Weights keep the Keras layout and order
(``[kernel, recurrent_kernel, bias]`` per LSTM, ``[kernel, bias]`` for Dense) so files round-trip.
"""
import os

import numpy as np

from .config import cfg

_W_ORDER = ("enc_K", "enc_R", "enc_b", "dec_K", "dec_R", "dec_b", "dense_W", "dense_b")


def glorot_uniform(rng, fan_in, fan_out, shape=None):
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rng.uniform(-lim, lim, (fan_in, fan_out) if shape is None else shape).astype(np.float32)


def orthogonal(rng, rows, cols):
    a = rng.standard_normal((rows, cols))
    u, _, vt = np.linalg.svd(a, full_matrices=False)
    return (u if u.shape == (rows, cols) else vt).astype(np.float32)


def init_lstm_weights(rng, F, H):
    """Keras LSTM defaults: glorot_uniform kernel, orthogonal recurrent kernel, zero bias with the
    forget block at one (unit_forget_bias)."""
    b = np.zeros(4 * H, np.float32)
    b[H:2 * H] = 1.0
    return glorot_uniform(rng, F, 4 * H), orthogonal(rng, H, 4 * H), b


def _as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


# ---- MFMA coverage for widths the persistent kernels are not built for (the reference ships latent_dim = 32,
# given_others_gt_mean_var_seq2seq.py:38): the layer is run at the next supported width with zero-padded weights.
# This is EXACT: a padded unit has zero kernel / recurrent columns and zero bias, so its gates are sigma(0), tanh(0) = 0
# and, from a zero state, c stays 0 and h = o * tanh(0) = 0 for every step; zero recurrent ROWS keep it from reaching
# the real units. ----
MFMA_WIDTHS = (64, 128, 256)


def padded_width(H, widths=MFMA_WIDTHS):
    """The width a layer of H units runs at on the MFMA kernels (H itself when supported), or None (H > 256)."""
    for w in widths:
        if H <= w:
            return w
    return None


def pad_lstm(K, R, b, Hp, pad_input=False):
    """Keras LSTM weights (F,4H), (H,4H), (4H,) -> (F',4Hp), (Hp,4Hp), (4Hp,): every gate block padded with zero columns,
    R (and K when the layer's input is itself a padded hidden sequence) with zero rows."""
    F, H = K.shape[0], R.shape[0]
    Fp = Hp if pad_input else F
    Kp = np.zeros((Fp, 4 * Hp), np.float32)
    Rp = np.zeros((Hp, 4 * Hp), np.float32)
    bp = np.zeros(4 * Hp, np.float32)
    for g in range(4):
        Kp[:F, g * Hp:g * Hp + H] = K[:, g * H:(g + 1) * H]
        Rp[:H, g * Hp:g * Hp + H] = R[:, g * H:(g + 1) * H]
        bp[g * Hp:g * Hp + H] = b[g * H:(g + 1) * H]
    return Kp, Rp, bp


def pad_rows(W, Hp):
    out = np.zeros((Hp,) + W.shape[1:], np.float32)
    out[:W.shape[0]] = W
    return out


def pad_cols(a, Hp):
    """(N,H) state -> (N,Hp) with zero columns."""
    a = _as_f32(a)
    out = np.zeros((a.shape[0], Hp), np.float32)
    out[:, :a.shape[1]] = a
    return out


def _keras_fit(model, trainer, inputs, y, batch_size, epochs, validation_split, shuffle, callbacks, initial_epoch,
               validation_data):
    """Keras `Model.fit` loop shared by the model objects: the LAST `validation_split` fraction is held out
    BEFORE shuffling, train indices are permuted every epoch (np.random), the last partial batch is used,
    callbacks see {'loss','val_loss','lr'}.  Under torch.distributed every rank takes its contiguous shard
    of each global batch (one gradient all-reduce per step inside trainer.train_step)."""
    import torch
    from . import parallel
    from .callbacks import History
    inputs = [_as_f32(a) for a in inputs]
    tgt = _as_f32(y)
    n = inputs[0].shape[0]
    val = None
    n_train = n
    if validation_data is not None:
        val = ([_as_f32(a) for a in validation_data[0]], _as_f32(validation_data[1]))
    elif validation_split and 0.0 < validation_split < 1.0:
        n_train = int(n * (1.0 - validation_split))
        val = ([a[n_train:] for a in inputs], tgt[n_train:])
    hist = History()
    cbs = [hist] + list(callbacks or [])
    for cb in cbs:
        cb.set_model(model)
        cb.on_train_begin()
    model.stop_training = False
    rank, world = parallel.world()
    d = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(model.device)
    # The training set lives on the device for the whole fit() when it fits (FOV_FIT_RESIDENT_BYTES, default 8 GiB of the
    # 288): a batch is then a device gather by the epoch's permutation (or a plain slice without shuffling) instead of a
    # NumPy fancy-index + three pageable uploads per step - at the reference's batch of 32 the host path was a quarter of
    # the step (fit() 0.345 ms per step against 0.26 ms for the bare train_step, tools/fit_epoch_time.py).
    limit = int(os.environ.get("FOV_FIT_RESIDENT_BYTES", str(8 << 30)))
    val_dev = None
    resident = sum(a.nbytes for a in inputs) + tgt.nbytes <= limit
    dev_arrays = [d(a) for a in inputs] + [d(tgt)] if resident else None
    for epoch in range(initial_epoch, epochs):
        idx = np.arange(n_train)
        if shuffle:
            np.random.shuffle(idx)
            # every rank must slice the SAME permutation: rank 0's is broadcast (the ranks' np.random states are
            # not synchronised), then each takes its contiguous shard of every global batch
            idx = parallel.broadcast_index(idx)
        idx_dev = torch.from_numpy(np.ascontiguousarray(idx, dtype=np.int64)).to(model.device) if (resident and shuffle) else None
        # ... and when twice the training set fits, the WHOLE epoch is gathered once (three launches per epoch): a batch is then
        # a contiguous slice - at the reference's batch of 32 the three per-step gathers were 11 us of GPU time and as much host
        # time again in a 0.13 ms step
        epoch_arrays = None
        if idx_dev is not None and 2 * (sum(a.nbytes for a in inputs) + tgt.nbytes) <= limit:
            epoch_arrays = [t.index_select(0, idx_dev) for t in dev_arrays]
        # the epoch's loss sum stays on the device (fp64): no host synchronisation per step, so the launches of step k + 1 are
        # queued while step k runs (the returned loss tensor is overwritten by the NEXT step: the add is queued before it)
        tot_t, cnt = torch.zeros(1, dtype=torch.float64, device=model.device), 0
        step = trainer.train_step
        for lo in range(0, n_train, batch_size):
            gidx = idx[lo:lo + batch_size]
            a, b = parallel.shard_range(len(gidx), rank, world)
            lidx = gidx[a:b]
            if not resident:
                batch = [d(arr[lidx]) for arr in inputs] + [d(tgt[lidx])]
            elif epoch_arrays is not None:
                batch = [t[lo + a:lo + b] for t in epoch_arrays]
            elif shuffle:
                sel = idx_dev[lo + a:lo + b]
                batch = [t.index_select(0, sel) for t in dev_arrays]
            else:
                batch = [t[lo + a:lo + b] for t in dev_arrays]   # idx is the identity: contiguous rows, no copy
            loss = step(*batch, n_global=len(gidx))
            tot_t.add_(loss.reshape(1), alpha=float(len(gidx)))      # (fp32 into the fp64 sum: one launch)
            cnt += len(gidx)
        tot = float(tot_t.item())
        logs = {"loss": tot / max(cnt, 1), "lr": trainer.lr}
        if val is not None and len(val[1]):
            vt, vc = 0.0, 0
            if val_dev is None and sum(a.nbytes for a in val[0]) + val[1].nbytes <= limit:
                val_dev = [d(a) for a in val[0]] + [d(val[1])]      # uploaded once per fit()
            vt_t = torch.zeros(1, dtype=torch.float64, device=model.device)
            for lo in range(0, len(val[1]), max(batch_size, 1)):
                sl = slice(lo, lo + batch_size)
                k = len(val[1][sl])
                vb = [t[sl] for t in val_dev] if val_dev is not None else [d(arr[sl]) for arr in val[0]] + [d(val[1][sl])]
                vt_t.add_(trainer.eval_loss(*vb).reshape(1), alpha=float(k))
                vc += k
            vt = float(vt_t.item())
            logs["val_loss"] = vt / vc
        trainer.check()      # a persistent kernel that gave up poisons its workspace and the optimizer skips the update
        model._w = trainer.weights_numpy()
        model._dw = None
        for cb in cbs:
            cb.on_epoch_end(epoch, logs)
        if model.stop_training:
            break
    for cb in cbs:
        cb.on_train_end()
    trainer.check()
    return hist


class KerasModelSurface:
    """The part of the Keras `Model` surface every model object here shares (what the reference's scripts call on their
    models: get/set_weights, save / load_weights, compile, fit, train_on_batch, the `lr` the callbacks adjust).  A
    subclass fills self._w (name -> float32 array), calls _init_surface(order, impl, device) and provides
    _make_trainer() and predict().  Weight files are .npz with the arrays under their names (h5py is not available,
    DESIGN section 1 row f4); the ORDER of get_weights / set_weights is Keras's: per layer kernel, recurrent_kernel,
    bias, layers in creation order."""

    def _init_surface(self, order, impl, device):
        self._order = tuple(order)
        self.impl, self.device = impl, device
        self._dw = self._ws = self._trainer = None       # device copies / workspace (lazy), trainer (lazy)
        self.optimizer = self.loss = None
        self.metrics = []
        self._lr = 1e-3
        self.stop_training = False

    # ---- weights ----
    def get_weights(self):
        return [self._w[k].copy() for k in self._order]

    def set_weights(self, weights):
        weights = list(weights)
        if len(weights) != len(self._order):
            raise ValueError("expected %d arrays, got %d" % (len(self._order), len(weights)))
        for k, a in zip(self._order, weights):
            a = _as_f32(a)
            if a.shape != self._w[k].shape:
                raise ValueError("%s: expected shape %s, got %s" % (k, self._w[k].shape, a.shape))
            self._w[k] = a
        self._dw = None
        if self._trainer is not None:     # Keras keeps the optimizer state across set_weights / load_weights
            self._trainer.load_weights_(self._w)

    @staticmethod
    def weights_path(path):
        """The file a weights path names: '.npz', '.h5', '.hdf5' as given, anything else gets '.npz' appended - so
        save_weights(p) / load_weights(p) / ModelCheckpoint(p) agree for every p."""
        path = str(path)
        return path if path.lower().endswith((".npz", ".h5", ".hdf5")) else path + ".npz"

    def _keras_layers(self):
        """[(layer, [(weight_name, array), ...])] the way Keras groups them: consecutive tensors that share a prefix form
        one layer - '<p>_K', '<p>_R', '<p>_b' an LSTM / ConvLSTM2D (kernel, recurrent_kernel, bias), '<p>_W', '<p>_b' a
        Dense / Conv (kernel, bias) - in creation order (= get_weights order)."""
        kinds = {"K": "kernel:0", "R": "recurrent_kernel:0", "b": "bias:0", "W": "kernel:0"}
        layers = []
        for k in self._order:
            prefix, _, suffix = k.rpartition("_")
            if suffix not in kinds or not prefix:
                prefix, suffix = k, "W"
            if not layers or layers[-1][0] != prefix:
                layers.append((prefix, []))
            layers[-1][1].append(("%s/%s" % (prefix, kinds[suffix]), self._w[k]))
        return layers

    def save_weights(self, path):
        """'.h5' / '.hdf5' names: a Keras-layout HDF5 weight file (keras_h5.py: superblock 0, one group per layer with
        `weight_names`, contiguous float32 datasets - what keras.Model.load_weights reads); otherwise '.npz'."""
        path = self.weights_path(path)
        # Data parallelism: the replicas are bit-identical, every rank runs the callbacks (and the scripts' final
        # model.save_weights) - only rank 0 writes, through a temporary file, so nobody ever reads a half-written one
        from . import parallel
        if parallel.world()[0] != 0:
            return
        tmp = "%s.tmp%d" % (path, os.getpid())
        if path.lower().endswith((".h5", ".hdf5")):
            from .keras_h5 import write_keras_layers
            write_keras_layers(tmp, self._keras_layers())
        else:
            with open(tmp, "wb") as f:
                np.savez(f, **self._w)
        os.replace(tmp, path)

    save = save_weights

    def load_weights(self, path):
        """A Keras HDF5 weight file (model.save_weights / ModelCheckpoint / model.save of the reference,
        given_others...py:484,570; read by the package's own HDF5 subset reader, keras_h5.py) or an '.npz' written by
        save_weights.  The file's tensors must match the model's shapes one to one, in Keras's order."""
        path = self.weights_path(path)
        with open(path, "rb") as f:
            magic = f.read(8)
        if magic == b"\x89HDF\r\n\x1a\n":
            from .keras_h5 import read_keras_weights
            self.set_weights(read_keras_weights(path, expected_shapes=[self._w[k].shape for k in self._order]))
            return
        with np.load(path) as z:
            self.set_weights([z[k] for k in self._order])

    def count_params(self):
        return int(sum(v.size for v in self._w.values()))

    def _device_weights(self):
        import torch
        from . import ops   # imported lazily so that host-only use does not need the .so
        if self._dw is None:
            self._dw = {k: torch.from_numpy(v).to(self.device) for k, v in self._w.items()}
            self._ws = ops.Workspace()
        return self._dw

    def _to_device(self, a):
        import torch
        return torch.from_numpy(_as_f32(a)).to(self.device)

    # ---- training surface ----
    def compile(self, optimizer="Adam", loss="mean_squared_error", metrics=None):
        """Keras `compile`.  Accepted: optimizer 'Adam' | 'RMSprop' (Keras defaults), loss 'mean_squared_error' | 'mse'
        (FoV_seq2seq.py:103, given_others...py:308, convlstm_seq2seq.py:287)."""
        opt = optimizer if isinstance(optimizer, str) else getattr(optimizer, "name", str(optimizer))
        if opt.lower() not in ("adam", "rmsprop"):
            raise ValueError("unsupported optimizer %r" % (optimizer,))
        if str(getattr(loss, "__name__", loss)).lower().lstrip("_") not in ("mean_squared_error", "mse"):
            raise ValueError("unsupported loss %r" % (loss,))
        self.optimizer, self.loss, self.metrics = opt.lower(), "mse", list(metrics or [])

    @property
    def lr(self):
        return self._lr

    @lr.setter
    def lr(self, value):
        self._lr = float(value)
        if self._trainer is not None:
            self._trainer.lr = self._lr

    _default_optimizer = "adam"

    def _make_trainer(self, optimizer):
        raise NotImplementedError

    def _get_trainer(self):
        if self._trainer is None:
            self._trainer = self._make_trainer(self.optimizer or self._default_optimizer)
            self._trainer.lr = self._lr
        return self._trainer

    def _fit_inputs(self, x):
        """Model inputs as the list of arrays the trainer's train_step takes (before the target)."""
        return [_as_f32(a) for a in x] if isinstance(x, (list, tuple)) else [_as_f32(x)]

    def fit(self, x, y, batch_size=32, epochs=1, validation_split=0.0, shuffle=True, callbacks=None, initial_epoch=0,
            verbose=0, validation_data=None):
        """Keras `Model.fit` (FoV_seq2seq.py:112-117); see _keras_fit for the semantics.  Returns a History."""
        if self.optimizer is None:
            raise RuntimeError("call compile() before fit()")
        if validation_data is not None:
            validation_data = (self._fit_inputs(validation_data[0]), validation_data[1])
        return _keras_fit(self, self._get_trainer(), self._fit_inputs(x), y, batch_size, epochs, validation_split, shuffle,
                          callbacks, initial_epoch, validation_data)

    def train_on_batch(self, x, y, **kw):
        tr = self._get_trainer()
        loss = tr.train_step(*[self._to_device(a) for a in self._fit_inputs(x)], self._to_device(y),
                             **{k: (None if v is None else self._to_device(v)) for k, v in kw.items()})
        value = float(loss.item())
        tr.check()
        self._w = tr.weights_numpy()
        self._dw = None
        return value


class _SubModel:
    def __init__(self, fn):
        self._fn = fn

    def predict(self, x, batch_size=None, verbose=0):
        return self._fn(x)

    predict_on_batch = predict


class Seq2SeqLSTM(KerasModelSurface):
    """Target-only seq2seq LSTM (1-layer encoder, 1-layer decoder, Dense(tanh) head)."""

    def __init__(self, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=64, recurrent_activation=None,
                 seed=None, impl="auto", device="cuda"):
        self.num_encoder_tokens = 3 * cfg.fps if num_encoder_tokens is None else int(num_encoder_tokens)
        self.num_decoder_tokens = int(num_decoder_tokens)
        self.latent_dim = int(latent_dim)
        self.recurrent_activation = recurrent_activation or cfg.recurrent_activation
        if self.recurrent_activation not in ("sigmoid", "hard_sigmoid"):
            raise ValueError("recurrent_activation must be 'sigmoid' or 'hard_sigmoid'")
        rng = np.random.default_rng(seed)
        w = {}
        w["enc_K"], w["enc_R"], w["enc_b"] = init_lstm_weights(rng, self.num_encoder_tokens, self.latent_dim)
        w["dec_K"], w["dec_R"], w["dec_b"] = init_lstm_weights(rng, self.num_decoder_tokens, self.latent_dim)
        w["dense_W"] = glorot_uniform(rng, self.latent_dim, self.num_decoder_tokens)
        w["dense_b"] = np.zeros(self.num_decoder_tokens, np.float32)
        self._w = w
        self._init_surface(_W_ORDER, impl, device)
        self.encoder_model = _SubModel(self._encoder_predict)
        self.decoder_model = _SubModel(self._decoder_predict)

    def _ops(self):
        from . import ops   # imported lazily so that host-only use does not need the .so
        return ops

    _dev = KerasModelSurface._to_device

    def _run_width(self):
        """Width the inference kernels run at: latent_dim, or the next MFMA width with zero-padded weights (exact)."""
        H = self.latent_dim
        if self.impl == "generic" or H in MFMA_WIDTHS:
            return H
        return padded_width(H) or H

    def _device_weights(self):
        import torch
        if self._dw is None:
            Hp, w = self._run_width(), self._w
            if Hp != self.latent_dim:
                pw = {}
                pw["enc_K"], pw["enc_R"], pw["enc_b"] = pad_lstm(w["enc_K"], w["enc_R"], w["enc_b"], Hp)
                pw["dec_K"], pw["dec_R"], pw["dec_b"] = pad_lstm(w["dec_K"], w["dec_R"], w["dec_b"], Hp)
                pw["dense_W"], pw["dense_b"] = pad_rows(w["dense_W"], Hp), w["dense_b"]
                for k in w:                    # subclasses' extra tensors: padded by their own rule (_pad_extra)
                    if k not in pw:
                        pw[k] = self._pad_extra(k, w, Hp)
                w = pw
            self._dw = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in w.items()}
            self._ws = self._ops().Workspace()
        return self._dw

    def _pad_extra(self, k, w, Hp):
        """A subclass tensor at the padded run width Hp (this class has none; width-independent ones pass through)."""
        return w[k]

    # ---- inference -------------------------------------------------------------------------
    def predict(self, x, batch_size=None, verbose=0):
        """Training-graph forward: x = [encoder_input (N,T_in,F_enc), decoder_input (N,T_out,F_dec)]
        -> (N,T_out,F_dec).  `batch_size` chunks the device calls (None = all at once)."""
        enc, dec_in = x
        enc, dec_in = _as_f32(enc), _as_f32(dec_in)
        ops, dw = self._ops(), self._device_weights()
        n = enc.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            o = ops.seq2seq_teacher_forced(self._dev(enc[lo:lo + bs]), self._dev(dec_in[lo:lo + bs]), dw,
                                           act=self.recurrent_activation, impl=self.impl, workspace=self._ws)
            outs.append(o.cpu().numpy())
        self._ws.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, dec_in.shape[1], self.num_decoder_tokens), np.float32)

    predict_on_batch = predict

    def decode_sequence(self, input_seq, first_decoder_input=None, predict_step=None, batch_size=None):
        """Autoregressive inference (FoV_seq2seq.py:154-178, any batch): encoder, then `predict_step`
        decoder steps feeding each Dense output back.  `first_decoder_input` (N,1,F_dec) defaults to
        the mu/sigma^2 of the last encoder second when the encoder input is raw (N,T,90) xyz."""
        from .utility import get_gt_target_xyz
        input_seq = _as_f32(input_seq)
        T_out = cfg.predict_step if predict_step is None else int(predict_step)
        if first_decoder_input is None:
            if input_seq.shape[-1] == self.num_decoder_tokens:
                first_decoder_input = input_seq[:, -1:, :]
            else:
                first_decoder_input = get_gt_target_xyz(input_seq[:, -1:, :].astype(np.float64))
        first_decoder_input = _as_f32(first_decoder_input)
        ops, dw = self._ops(), self._device_weights()
        n = input_seq.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            o = ops.seq2seq_decode(self._dev(input_seq[lo:lo + bs]), self._dev(first_decoder_input[lo:lo + bs]), dw,
                                   T_out, act=self.recurrent_activation, impl=self.impl, workspace=self._ws)
            outs.append(o.cpu().numpy())
        self._ws.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out, self.num_decoder_tokens), np.float32)

    def _encoder_predict(self, input_seq):
        """encoder_model.predict(input_seq) -> [state_h, state_c]   (FoV_seq2seq.py:137,156)."""
        ops, dw = self._ops(), self._device_weights()
        _, hT, cT = ops.lstm_seq(self._dev(input_seq), dw["enc_K"], dw["enc_R"], dw["enc_b"],
                                 act=self.recurrent_activation, impl=self.impl, return_sequences=False,
                                 workspace=self._ws)
        self._ws.check()
        H = self.latent_dim
        return [hT.cpu().numpy()[:, :H], cT.cpu().numpy()[:, :H]]

    def _decoder_predict(self, x):
        """decoder_model.predict([target_seq (N,T,F_dec), h, c]) -> [outputs (N,T,F_dec), h, c]
        (FoV_seq2seq.py:139-148,167-168)."""
        target_seq, h, c = x
        ops, dw = self._ops(), self._device_weights()
        H, Hp = self.latent_dim, self._run_width()
        if Hp != H:
            h, c = pad_cols(h, Hp), pad_cols(c, Hp)
        hs, hT, cT = ops.lstm_seq(self._dev(target_seq), dw["dec_K"], dw["dec_R"], dw["dec_b"], self._dev(h),
                                  self._dev(c), act=self.recurrent_activation, impl=self.impl, workspace=self._ws)
        y = ops.dense(hs, dw["dense_W"], dw["dense_b"], activation="tanh")
        self._ws.check()
        return [y.cpu().numpy(), hT.cpu().numpy()[:, :H], cT.cpu().numpy()[:, :H]]

    # ---- training surface: KerasModelSurface (FoV_seq2seq.py:103 compile, :112-117 fit) ----
    def _make_trainer(self, optimizer):
        from .training import PaddedTrainer, Seq2SeqTrainer
        make = lambda w: Seq2SeqTrainer(w, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer, lr=self._lr,
                                        device=self.device)
        Hp = self._run_width()
        if Hp != self.latent_dim:      # latent_dim = 32 / 40 / ...: train on the matrix-core kernels at the next supported width (exact)
            return PaddedTrainer(make, self._w, self.latent_dim, Hp)
        return make(self._w)


class NoTeacherForcingSeq2Seq(Seq2SeqLSTM):
    """One-layer target-only seq2seq trained and run WITHOUT teacher forcing
    (mycode/FoV_seq2seq_no_teac_forc.py:37-149, `onelayer_tar_seq2seq`): the decoder is unrolled `predict_step`
    times on its own Dense output.  Inputs as the script's Model: [encoder_input (N,T_in,F), decoder_input (N,1,O)],
    or encoder_input alone under cfg.enc_last_out_as_dec_in (:128-146); returns (N,T_out,O).

    Flags default to the script's: `decoder_no_init_state=True` (:29 — decoder step 0 starts from zero state),
    cfg.add_residual_link / cfg.enc_last_out_as_dec_in / cfg.rescale_input / cfg.embed_frame_state_enc2dec /
    cfg.has_reconstruct_loss False.  With the reconstruction decoder (:56-59,90-95,120-139) the model has two outputs,
    as the script's `Model(..., [decoder_outputs, recons_decoder_outputs])`: predict returns [prediction (N,T_out,O),
    reconstruction (N,T_out,F)], fit / train_on_batch take y = [decoder_target, reconstruction_target] (the script passes
    encoder_input_data[:, ::-1, :], :267) and minimise MSE + MSE; its Model lists [encoder_inputs, time_ind_input] as
    inputs (:135) - the second one is never read by the graph and is ignored here too.  Not built: BatchNorm on the
    feedback (add_bn, a module constant that is False)."""

    def __init__(self, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=64, recurrent_activation=None, seed=None,
                 impl="auto", device="cuda", predict_step=None, decoder_no_init_state=True, add_residual_link=None,
                 enc_last_out_as_dec_in=None, rescale_input=None, embed_frame_state_enc2dec=None, has_reconstruct_loss=None):
        super().__init__(num_encoder_tokens, num_decoder_tokens, latent_dim, recurrent_activation, seed, impl, device)
        from .training import self_fed_weight_order
        knob = lambda v, name: bool(getattr(cfg, name, False)) if v is None else bool(v)
        self.predict_step = cfg.predict_step if predict_step is None else int(predict_step)
        self.decoder_no_init_state = bool(decoder_no_init_state)
        self.add_residual_link = knob(add_residual_link, "add_residual_link")
        self.enc_last_out_as_dec_in = knob(enc_last_out_as_dec_in, "enc_last_out_as_dec_in")
        self.dense_activation = "relu" if knob(rescale_input, "rescale_input") else "tanh"
        self.embed_frame_state_enc2dec = knob(embed_frame_state_enc2dec, "embed_frame_state_enc2dec")
        self.has_reconstruct_loss = knob(has_reconstruct_loss, "has_reconstruct_loss")
        H, F, O = self.latent_dim, self.num_encoder_tokens, self.num_decoder_tokens
        rng = np.random.default_rng(None if seed is None else seed + 1)
        if self.add_residual_link:
            self._w["res_W"] = glorot_uniform(rng, O, O)
            self._w["res_b"] = np.zeros(O, np.float32)
        if self.embed_frame_state_enc2dec:
            for n in ("emb1", "emb2"):
                self._w[n + "_W"] = glorot_uniform(rng, H, H)
                self._w[n + "_b"] = np.zeros(H, np.float32)
        if self.has_reconstruct_loss:
            self._w["rec_K"], self._w["rec_R"], self._w["rec_b"] = init_lstm_weights(rng, F, H)
            self._w["recd_W"] = glorot_uniform(rng, H, F)
            self._w["recd_b"] = np.zeros(F, np.float32)
        self._order = self_fed_weight_order(self.add_residual_link, self.embed_frame_state_enc2dec, self.has_reconstruct_loss)

    def _pad_extra(self, k, w, Hp):
        """The optional layers that carry the hidden width (FoV_seq2seq_no_teac_forc.py:47-59): the state embeddings
        Dense(latent_dim, tanh) get zero rows AND columns (tanh(0) = 0: a padded unit's embedded state stays exactly 0), the
        reconstruction LSTM is padded like the other two, its Dense gets zero rows.  The residual Dense is O x O."""
        if k in ("emb1_W", "emb2_W"):
            out = np.zeros((Hp, Hp), np.float32)
            out[:w[k].shape[0], :w[k].shape[1]] = w[k]
            return out
        if k in ("emb1_b", "emb2_b"):
            out = np.zeros(Hp, np.float32)
            out[:w[k].shape[0]] = w[k]
            return out
        if k in ("rec_K", "rec_R", "rec_b"):
            return dict(zip(("rec_K", "rec_R", "rec_b"), pad_lstm(w["rec_K"], w["rec_R"], w["rec_b"], Hp)))[k]
        if k == "recd_W":
            return pad_rows(w[k], Hp)
        return w[k]

    def _split_inputs(self, x):
        if self.enc_last_out_as_dec_in:
            enc = x[0] if isinstance(x, (list, tuple)) else x
            return _as_f32(enc), None
        return _as_f32(x[0]), _as_f32(x[1])

    def predict(self, x, batch_size=None, verbose=0):
        import torch
        enc, dec0 = self._split_inputs(x)
        ops, dw, act = self._ops(), self._device_weights(), self.recurrent_activation
        T_out, O, H, F = self.predict_step, self.num_decoder_tokens, self.latent_dim, self.num_encoder_tokens
        plain = not (self.add_residual_link or self.enc_last_out_as_dec_in or self.dense_activation != "tanh" or
                     self.has_reconstruct_loss or (self.embed_frame_state_enc2dec and not self.decoder_no_init_state))
        n = enc.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs, recs = [], []

        def dense(v, W, b, dact=None):
            dact = dact or self.dense_activation
            y = ops.dense(v, W, b, activation="tanh" if dact == "tanh" else None)
            return y if dact == "tanh" else ops.act_fwd(y, "relu", out=y)

        def unroll(pre, head, xin, h, c, r, dact, width):
            o = torch.empty((xin.shape[0], T_out, width), dtype=torch.float32, device=self.device)
            for t in range(T_out):
                _, h, c = ops.lstm_seq(xin.reshape(-1, 1, width), dw[pre + "_K"], dw[pre + "_R"], dw[pre + "_b"], h, c, act=act,
                                       impl=self.impl, return_sequences=False, workspace=self._ws)
                xin = dense(h, dw[head + "_W"], dw[head + "_b"], dact)
                if r is not None:
                    xin = ops.act_bwd(r, r, base=xin, activation=None)     # y_t = Dense(h_t) + residual
                o[:, t] = xin
            return o
        for lo in range(0, n, max(bs, 1)):
            e = self._dev(enc[lo:lo + bs])
            if plain:     # ONE fused call; a zero-length encoder input is the zero initial state of :98-99
                o = ops.seq2seq_decode(e[:, :0] if self.decoder_no_init_state else e, self._dev(dec0[lo:lo + bs]), dw, T_out,
                                       act=act, impl=self.impl, workspace=self._ws)
                outs.append(o.cpu().numpy())
                continue
            B = e.shape[0]
            _, h_enc, c_enc = ops.lstm_seq(e, dw["enc_K"], dw["enc_R"], dw["enc_b"], act=act, impl=self.impl, return_sequences=False,
                                           workspace=self._ws)
            sh, sc = h_enc, c_enc
            if self.embed_frame_state_enc2dec:
                sh, sc = dense(h_enc, dw["emb1_W"], dw["emb1_b"], "tanh"), dense(c_enc, dw["emb2_W"], dw["emb2_b"], "tanh")
            xin = dense(h_enc, dw["dense_W"], dw["dense_b"]) if self.enc_last_out_as_dec_in else self._dev(dec0[lo:lo + bs]).reshape(B, O)
            h, c = (torch.zeros_like(sh), torch.zeros_like(sc)) if self.decoder_no_init_state else (sh, sc)
            r = dense(xin, dw["res_W"], dw["res_b"]) if self.add_residual_link else None
            outs.append(unroll("dec", "dense", xin, h, c, r, None, O).cpu().numpy())
            if self.has_reconstruct_loss:
                xr = dense(h_enc, dw["recd_W"], dw["recd_b"], "tanh")
                recs.append(unroll("rec", "recd", xr, sh, sc, None, "tanh", F).cpu().numpy())
        self._ws.check()
        y = np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out, O), np.float32)
        if self.has_reconstruct_loss:
            return [y, np.concatenate(recs, axis=0) if recs else np.zeros((0, T_out, F), np.float32)]
        return y

    predict_on_batch = predict

    def _make_trainer(self, optimizer):
        from .training import SelfFedSeq2SeqTrainer
        return SelfFedSeq2SeqTrainer(
            self._w, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer, lr=self._lr, device=self.device,
            decoder_no_init_state=self.decoder_no_init_state, add_residual_link=self.add_residual_link,
            enc_last_out_as_dec_in=self.enc_last_out_as_dec_in, dense_activation=self.dense_activation,
            embed_frame_state_enc2dec=self.embed_frame_state_enc2dec, has_reconstruct_loss=self.has_reconstruct_loss)

    def _fit_inputs(self, x):
        enc, dec0 = self._split_inputs(x)
        if dec0 is None:      # the trainer's positional signature keeps a decoder-input slot; it is not read
            dec0 = np.zeros((enc.shape[0], 1, self.num_decoder_tokens), np.float32)
        return [enc, dec0]

    def _fit_target(self, y):
        """[decoder_target (N,T,O), reconstruction_target (N,T,F)] -> one (N,T,O+F) array for the shared fit loop."""
        if not self.has_reconstruct_loss:
            return y
        if not isinstance(y, (list, tuple)) or len(y) != 2:
            raise ValueError("a model with the reconstruction decoder takes y = [decoder_target, reconstruction_target]")
        a, b = _as_f32(y[0]), _as_f32(y[1])
        if a.shape[:2] != b.shape[:2] or b.shape[2] != self.num_encoder_tokens:
            raise ValueError("reconstruction target %s does not match (N, %d, %d)" % (b.shape, a.shape[1], self.num_encoder_tokens))
        return np.concatenate([a, b], -1)

    def fit(self, x, y, validation_data=None, **kw):
        if validation_data is not None:
            validation_data = (validation_data[0], self._fit_target(validation_data[1]))
        return super().fit(x, self._fit_target(y), validation_data=validation_data, **kw)

    def train_on_batch(self, x, y, **kw):
        return super().train_on_batch(x, self._fit_target(y), **kw)


class StackedSeq2SeqLSTM(KerasModelSurface):
    """L-layer target-only seq2seq of mycode/Fov_seq2seq_2layers.py (:232-272, sampling models :360-397, host loop :399-430)
    and 3layers.py: every layer `latent_dim` wide (the scripts pass latent_dim//2), encoder layer l seeds decoder layer l,
    Dense(6, tanh) on the top layer.  predict = the teacher-forced training graph, decode_sequence = the autoregressive
    loop (batched).  3layers.py wires its third DECODER layer through `decoder_lstm2` in the training graph (:270) and
    through the never-trained `decoder_lstm3` at inference (:356); here layer 3 simply has its own weights."""

    def __init__(self, num_encoder_tokens=6, num_decoder_tokens=6, latent_dim=32, num_layers=2, recurrent_activation=None,
                 seed=None, impl="auto", device="cuda"):
        from .training import stacked_weight_order
        self.F, self.O, self.H, self.L = int(num_encoder_tokens), int(num_decoder_tokens), int(latent_dim), int(num_layers)
        self.recurrent_activation = recurrent_activation or cfg.recurrent_activation
        rng = np.random.default_rng(seed)
        w = {}
        for side, f0 in (("enc", self.F), ("dec", self.O)):
            for l in range(self.L):
                w["%s%d_K" % (side, l)], w["%s%d_R" % (side, l)], w["%s%d_b" % (side, l)] = \
                    init_lstm_weights(rng, f0 if l == 0 else self.H, self.H)
        w["dense_W"] = glorot_uniform(rng, self.H, self.O)
        w["dense_b"] = np.zeros(self.O, np.float32)
        self._w = w
        self._init_surface(stacked_weight_order(self.L), impl, device)

    def _device(self):
        from . import ops
        return ops, self._device_weights()

    def _encode(self, ops, dw, e):
        states, inp = [], e
        for l in range(self.L):
            inp, h, c = ops.lstm_seq(inp, dw["enc%d_K" % l], dw["enc%d_R" % l], dw["enc%d_b" % l], act=self.recurrent_activation,
                                     impl=self.impl, workspace=self._ws)
            states.append((h, c))
        return states

    def predict(self, x, batch_size=None, verbose=0):
        """[encoder_input (N,T_in,F), decoder_input (N,T_out,O)] -> (N,T_out,O), teacher-forced graph (:332)."""
        import torch
        enc, dec_in = _as_f32(x[0]), _as_f32(x[1])
        ops, dw = self._device()
        n = enc.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            states = self._encode(ops, dw, torch.from_numpy(enc[lo:lo + bs]).to(self.device))
            inp = torch.from_numpy(dec_in[lo:lo + bs]).to(self.device)
            for l in range(self.L):
                inp, _, _ = ops.lstm_seq(inp, dw["dec%d_K" % l], dw["dec%d_R" % l], dw["dec%d_b" % l], states[l][0], states[l][1],
                                         act=self.recurrent_activation, impl=self.impl, workspace=self._ws)
            outs.append(ops.dense(inp, dw["dense_W"], dw["dense_b"], activation="tanh").cpu().numpy())
        self._ws.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, dec_in.shape[1], self.O), np.float32)

    predict_on_batch = predict

    def decode_sequence(self, input_seq, first_decoder_input, predict_step=None, batch_size=None):
        """Autoregressive loop of :399-430, any batch: encoder states, then `predict_step` steps through all decoder
        layers, each Dense output fed back as the next input."""
        import torch
        enc, d0 = _as_f32(input_seq), _as_f32(first_decoder_input)
        T_out = cfg.predict_step if predict_step is None else int(predict_step)
        ops, dw = self._device()
        n = enc.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            states = self._encode(ops, dw, torch.from_numpy(enc[lo:lo + bs]).to(self.device))
            B = states[0][0].shape[0]
            xin = torch.from_numpy(d0[lo:lo + bs]).to(self.device).reshape(B, 1, self.O)
            y = torch.empty((B, T_out, self.O), dtype=torch.float32, device=self.device)
            for t in range(T_out):
                inp = xin
                for l in range(self.L):
                    inp, h, c = ops.lstm_seq(inp, dw["dec%d_K" % l], dw["dec%d_R" % l], dw["dec%d_b" % l], states[l][0], states[l][1],
                                             act=self.recurrent_activation, impl=self.impl, workspace=self._ws)
                    states[l] = (h, c)
                xin = ops.dense(inp, dw["dense_W"], dw["dense_b"], activation="tanh")
                y[:, t] = xin[:, 0]
            outs.append(y.cpu().numpy())
        self._ws.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out, self.O), np.float32)

    def _make_trainer(self, optimizer):
        from .training import StackedSeq2SeqTrainer
        return StackedSeq2SeqTrainer(self._w, self.L, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer,
                                     lr=self._lr, device=self.device)


class OthersContextSeq2Seq(KerasModelSurface):
    """The other decoder heads of mycode/given_others_gt_mean_var_seq2seq.py (2+2-layer model, no teacher forcing), chosen by
    the script's module flags (:48-56):
      mode='target_user_only'  y_t = decoder_dense(h2_t)                                         (:219-220)
      mode='others_mlp'        others_t -> Dense(256, relu) -> Dense(latent_dim, relu), concatenated with h2_t   (:153-156,223-233)
      mode='others_lstm'       two Bidirectional LSTMs over the others' future mu/var, concatenated with h2_t    (:157-166,234-240)
    (`mlp_mixing`, the flag the script ships, is OthersMixingSeq2Seq.)  Inputs as the script's Model (:301-307):
    [encoder_input (N,T_in,F), others_fut_input (N,T_out,U-1,6), decoder_input (N,1,6)] -> (N,T_out,6);
    'target_user_only' takes [encoder_input, decoder_input]."""

    def __init__(self, mode, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=32, num_user=34, recurrent_activation=None,
                 seed=None, impl="auto", device="cuda", predict_step=None):
        from .training import others_context_order
        if mode not in ("target_user_only", "others_mlp", "others_lstm"):
            raise ValueError("mode must be 'target_user_only', 'others_mlp' or 'others_lstm'")
        self.mode = mode
        self.F = 3 * cfg.fps if num_encoder_tokens is None else int(num_encoder_tokens)
        self.O, self.H, self.U = int(num_decoder_tokens), int(latent_dim), int(num_user)
        self.recurrent_activation = recurrent_activation or cfg.recurrent_activation
        self.predict_step = cfg.predict_step if predict_step is None else int(predict_step)
        rng = np.random.default_rng(seed)
        w, H, n_oth = {}, self.H, (self.U - 1) * self.O
        for name, f in (("enc1", self.F), ("enc2", H), ("dec1", self.O), ("dec2", H)):
            w[name + "_K"], w[name + "_R"], w[name + "_b"] = init_lstm_weights(rng, f, H)
        Cc = {"target_user_only": 0, "others_mlp": H, "others_lstm": 2 * H}[mode]
        w["dense_W"] = glorot_uniform(rng, Cc + H, self.O)
        w["dense_b"] = np.zeros(self.O, np.float32)
        if mode == "others_mlp":
            w["oth_W1"], w["oth_b1"] = glorot_uniform(rng, n_oth, 256), np.zeros(256, np.float32)
            w["oth_W2"], w["oth_b2"] = glorot_uniform(rng, 256, H), np.zeros(H, np.float32)
        elif mode == "others_lstm":
            for j, f in ((1, n_oth), (2, 2 * H)):
                for d in ("f", "b"):
                    n = "ol%d%s" % (j, d)
                    w[n + "_K"], w[n + "_R"], w[n + "_b"] = init_lstm_weights(rng, f, H)
        self._w = w
        self._init_surface(others_context_order(mode), impl, device)

    def _inputs(self, x):
        if self.mode == "target_user_only":
            enc, dec0 = _as_f32(x[0]), _as_f32(x[-1])
            return enc, np.zeros((enc.shape[0], self.predict_step, self.U - 1, self.O), np.float32), dec0
        return _as_f32(x[0]), _as_f32(x[1]), _as_f32(x[2])

    def _context(self, ops, dw, oth):
        """(B,T_out,U-1,6) device tensor -> [ctx_t W_c + b] (B,T_out,O), or None."""
        import torch
        B, T = oth.shape[0], oth.shape[1]
        H, act = self.H, self.recurrent_activation
        if self.mode == "target_user_only":
            return None
        if self.mode == "others_mlp":
            a1 = ops.dense(oth.reshape(B * T, -1), dw["oth_W1"], dw["oth_b1"], activation=None)
            ops.act_fwd(a1, "relu", out=a1)
            ctx = ops.dense(a1, dw["oth_W2"], dw["oth_b2"], activation=None)
            ops.act_fwd(ctx, "relu", out=ctx)
        else:
            seq, init = oth.reshape(B, T, -1), {"f": (None, None), "b": (None, None)}
            for j in (1, 2):
                outs = {}
                for d, xin in (("f", seq), ("b", torch.flip(seq, (1,)))):
                    n = "ol%d%s" % (j, d)
                    outs[d] = ops.lstm_seq(xin, dw[n + "_K"], dw[n + "_R"], dw[n + "_b"], init[d][0], init[d][1], act=act,
                                           impl=self.impl, workspace=self._ws)
                seq = torch.cat([outs["f"][0], torch.flip(outs["b"][0], (1,))], 2)
                init = {d: (outs[d][1], outs[d][2]) for d in ("f", "b")}     # the list the first Bidirectional returns
            ctx = seq.reshape(B * T, 2 * H)
        Cc = dw["dense_W"].shape[0] - H
        return ops.dense(ctx, dw["dense_W"][:Cc], dw["dense_b"], activation=None).reshape(B, T, self.O)

    def predict(self, x, batch_size=None, verbose=0):
        import torch
        from . import ops
        enc, oth, dec0 = self._inputs(x)
        dw, act, H, O = self._device_weights(), self.recurrent_activation, self.H, self.O
        T_out = oth.shape[1]
        W_h = dw["dense_W"][dw["dense_W"].shape[0] - H:].contiguous()
        n = enc.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            d = lambda a: torch.from_numpy(a[lo:lo + bs]).to(self.device)
            e = d(enc)
            B = e.shape[0]
            ctx_proj = self._context(ops, dw, d(oth))
            hs1, h1, c1 = ops.lstm_seq(e, dw["enc1_K"], dw["enc1_R"], dw["enc1_b"], act=act, impl=self.impl, workspace=self._ws)
            _, h2, c2 = ops.lstm_seq(hs1, dw["enc2_K"], dw["enc2_R"], dw["enc2_b"], act=act, impl=self.impl, return_sequences=False,
                                     workspace=self._ws)
            xin = d(dec0).reshape(B, 1, O)
            y = torch.empty((B, T_out, O), dtype=torch.float32, device=self.device)
            for t in range(T_out):
                _, h1, c1 = ops.lstm_seq(xin, dw["dec1_K"], dw["dec1_R"], dw["dec1_b"], h1, c1, act=act, impl=self.impl,
                                         return_sequences=False, workspace=self._ws)
                _, h2, c2 = ops.lstm_seq(h1.view(B, 1, H), dw["dec2_K"], dw["dec2_R"], dw["dec2_b"], h2, c2, act=act, impl=self.impl,
                                         return_sequences=False, workspace=self._ws)
                yt = ops.dense(h2, W_h, dw["dense_b"], activation="tanh") if ctx_proj is None else \
                    ops.dense_add(h2, W_h, None, ctx_proj[:, t], activation="tanh")
                y[:, t] = yt
                xin = yt.view(B, 1, O)
            outs.append(y.cpu().numpy())
        self._ws.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out, O), np.float32)

    predict_on_batch = predict

    def _make_trainer(self, optimizer):
        from .training import OthersContextTrainer
        return OthersContextTrainer(self._w, self.mode, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer,
                                    lr=self._lr, device=self.device)

    def _fit_inputs(self, x):
        return list(self._inputs(x))

    def fit(self, x, y, **kw):
        """Keras `Model.fit` as the script calls it (:500-506)."""
        self.predict_step = _as_f32(y).shape[1]
        return super().fit(x, y, **kw)

    def train_on_batch(self, x, y):
        self.predict_step = _as_f32(y).shape[1]
        return super().train_on_batch(x, y)


class NoTeacherForcingOthersConvLSTM(KerasModelSurface):
    """The second model of mycode/FoV_seq2seq_no_teac_forc.py (:420-486): no teacher forcing, the decoder's Dense head
    concatenates the decoder output with Dense(latent_dim)(Flatten(.)) of a ConvLSTM2D(latent_dim, kernel (num_user-1, 3),
    'same') run over the other users' future.  Inputs as the script's Model (:486): [encoder_input (N,T_in,F),
    others_fut_input (N,T_out,num_user-1,fps,3), decoder_input (N,1,6)] -> (N,T_out,6); compile('Adam', 'mean_squared_error'),
    fit / predict as :553-558,565-570 (training.OthersFutureConvLSTMTrainer)."""

    def __init__(self, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=64, num_user=34, fps=None, recurrent_activation=None,
                 seed=None, impl="auto", device="cuda"):
        from .training import OTHERS_FUTURE_ORDER
        self.F = 3 * cfg.fps if num_encoder_tokens is None else int(num_encoder_tokens)
        self.O, self.H, self.U = int(num_decoder_tokens), int(latent_dim), int(num_user)
        self.fps = cfg.fps if fps is None else int(fps)
        self.recurrent_activation = recurrent_activation or cfg.recurrent_activation
        rng = np.random.default_rng(seed)
        w, H = {}, self.H
        w["enc_K"], w["enc_R"], w["enc_b"] = init_lstm_weights(rng, self.F, H)
        kh, kw = self.U - 1, 3
        rf = kh * kw
        lim = np.sqrt(6.0 / (rf * 3 + rf * 4 * H))               # glorot_uniform over the receptive field
        w["oth_K"] = rng.uniform(-lim, lim, (kh, kw, 3, 4 * H)).astype(np.float32)
        w["oth_R"] = orthogonal(rng, kh * kw * H, 4 * H).reshape(kh, kw, H, 4 * H)
        w["oth_b"] = np.zeros(4 * H, np.float32)
        w["oth_b"][H:2 * H] = 1.0                                # unit_forget_bias
        w["flat_W"] = glorot_uniform(rng, kh * self.fps * H, H)
        w["flat_b"] = np.zeros(H, np.float32)
        w["dec_K"], w["dec_R"], w["dec_b"] = init_lstm_weights(rng, self.O, H)
        w["dense_W"] = glorot_uniform(rng, 2 * H, self.O)
        w["dense_b"] = np.zeros(self.O, np.float32)
        self._w = w
        self._init_surface(OTHERS_FUTURE_ORDER, impl, device)

    def _make_trainer(self, optimizer):
        from .training import OthersFutureConvLSTMTrainer
        return OthersFutureConvLSTMTrainer(self._w, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer, lr=self._lr,
                                           device=self.device)

    def predict(self, x, batch_size=None, verbose=0):
        import torch
        from . import ops
        enc, oth, dec0 = (_as_f32(a) for a in x)
        tr = self._get_trainer()
        if self._dw is None:          # weights were set from outside since the trainer last saw them
            tr.load_weights_(self._w)
            self._dw = tr.w
        n, T_out = enc.shape[0], oth.shape[1]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            d = lambda a: torch.from_numpy(a[lo:lo + bs]).to(self.device)
            e = d(enc)
            B = e.shape[0]
            _, hT, cT = ops.lstm_seq(e, tr.w["enc_K"], tr.w["enc_R"], tr.w["enc_b"], act=self.recurrent_activation, impl=self.impl,
                                     return_sequences=False, workspace=tr.ws)
            S, _ = tr.branch_forward(d(oth), tape=False)
            tp = tr.decode(d(dec0).reshape(B, self.O), hT, cT, S, T_out, tape=False)
            outs.append(tp["XA"][1:].transpose(0, 1).cpu().numpy())
        tr.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out, self.O), np.float32)

    predict_on_batch = predict


class KerasSingleLSTM(KerasModelSurface):
    """Single-layer model of mycode/lstm_keras.py: ONE LSTM from zero state + Dense(6, tanh) per step, Adam + MSE.
      unrolled=False  1st part (:59-83): x (N,T,F) -> (N,T,6), one input second per step;
      unrolled=True   sampling model / 2nd part (:120-157,214-241): x (N,1,F) -> (N,predict_step,6); under
                      cfg.predict_mean_var and cfg.sample_and_refeed (`sample_and_refeed`) the next input is a sampled
                      second (mean mu, stddev = the predicted variance, planar x|y|z layout, :39-44,139-149), otherwise
                      the same input second is shown to every step.
    Weights in Keras order [kernel, recurrent_kernel, bias, dense kernel, dense bias]."""

    def __init__(self, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=64, recurrent_activation=None, seed=None,
                 impl="auto", device="cuda", unrolled=False, sample_and_refeed=None, predict_step=None):
        self.F = 3 * cfg.fps if num_encoder_tokens is None else int(num_encoder_tokens)
        self.O, self.H = int(num_decoder_tokens), int(latent_dim)
        self.recurrent_activation = recurrent_activation or cfg.recurrent_activation
        self.unrolled = bool(unrolled)
        if sample_and_refeed is None:
            sample_and_refeed = bool(cfg.predict_mean_var and cfg.sample_and_refeed)
        self.sample_and_refeed = bool(unrolled and sample_and_refeed)
        if self.sample_and_refeed and (self.O != 6 or self.F % 3):
            raise ValueError("the sampled re-feed needs num_decoder_tokens == 6 and a 3*fps-wide input")
        self.predict_step = cfg.predict_step if predict_step is None else int(predict_step)
        rng = np.random.default_rng(seed)
        w = {}
        w["K"], w["R"], w["b"] = init_lstm_weights(rng, self.F, self.H)
        w["dense_W"] = glorot_uniform(rng, self.H, self.O)
        w["dense_b"] = np.zeros(self.O, np.float32)
        self._w = w
        self._init_surface(("K", "R", "b", "dense_W", "dense_b"), impl, device)

    def predict(self, x, batch_size=None, verbose=0, noise=None):
        """`noise` (predict_step-1, N, F) standard normal for the sampled re-feed (drawn with torch.randn if None)."""
        import torch
        from . import ops
        x = _as_f32(x)
        dw, act = self._device_weights(), self.recurrent_activation
        n = x.shape[0]
        bs = n if not batch_size else int(batch_size)
        P = self.predict_step
        outs = []
        for lo in range(0, n, max(bs, 1)):
            xd = torch.from_numpy(x[lo:lo + bs]).to(self.device)
            B = xd.shape[0]
            if self.unrolled and not self.sample_and_refeed:
                xd = xd.expand(B, P, self.F).contiguous()          # the same second is shown to every step (:131-153)
            if not self.sample_and_refeed:
                hs, _, _ = ops.lstm_seq(xd, dw["K"], dw["R"], dw["b"], act=act, impl=self.impl, workspace=self._ws)
                outs.append(ops.dense(hs, dw["dense_W"], dw["dense_b"], activation="tanh").cpu().numpy())
                continue
            nz = torch.randn((P - 1, B, self.F), dtype=torch.float32, device=self.device) if noise is None \
                else torch.from_numpy(_as_f32(noise[:, lo:lo + bs])).to(self.device)
            y = torch.empty((B, P, self.O), dtype=torch.float32, device=self.device)
            xin, h, c = xd.reshape(B, 1, self.F), None, None
            for t in range(P):
                _, h, c = ops.lstm_seq(xin, dw["K"], dw["R"], dw["b"], h, c, act=act, impl=self.impl, return_sequences=False,
                                       workspace=self._ws)
                yt = ops.dense(h, dw["dense_W"], dw["dense_b"], activation="tanh")
                y[:, t] = yt
                if t < P - 1:
                    xin = ops.sample_refeed(yt[:, :3].contiguous(), yt[:, 3:].contiguous(), nz[t], std="var", planar=True).view(B, 1, self.F)
            outs.append(y.cpu().numpy())
        self._ws.check()
        T = P if self.unrolled else x.shape[1]
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T, self.O), np.float32)

    predict_on_batch = predict

    def _make_trainer(self, optimizer):
        from .training import SingleLSTMTrainer
        return SingleLSTMTrainer(self._w, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer, lr=self._lr,
                                 device=self.device, unrolled=self.unrolled, sample_and_refeed=self.sample_and_refeed)

    def train_on_batch(self, x, y, noise=None):
        """`noise` (predict_step-1, N, F): the normal draws of the sampled re-feed (torch.randn when None)."""
        return super().train_on_batch(x, y, noise=noise)


_MIX_ORDER = ("enc1_K", "enc1_R", "enc1_b", "enc2_K", "enc2_R", "enc2_b", "dec1_K", "dec1_R", "dec1_b",
              "dec2_K", "dec2_R", "dec2_b", "dense_W", "dense_b", "mix_W", "mix_b")


class OthersMixingSeq2Seq(KerasModelSurface):
    """Target + others mu/sigma^2 mixing seq2seq (mycode/given_others_gt_mean_var_seq2seq.py:98-308 with
    mlp_mixing, cfg.predict_mean_var=True, no teacher forcing): 2-layer LSTM encoder, 2-layer decoder
    unrolled `predict_step` times feeding its own output back, per-step Dense(6,tanh) and a mixing
    Dense(6,tanh) over [others_t (U-1,6) ; prediction (1,6)] flattened user-major.

    predict([encoder_input (N,T_in,F_enc), others_fut_input (N,T_out,U-1,6), decoder_input (N,1,6)])
    -> (N,T_out,6).  H = 256: encoder layer 2 on the wide-input layer kernel and the whole unrolled decoder in ONE
    persistent launch (fov_mix_decoder_fwd); other widths run per-step library calls (layer-1 step, the layer-2
    input projection as an MFMA GEMM, layer-2 step, fused head).  The "others" half of the mixing product is
    hoisted out of the loop as one GEMV batch."""

    fused_decoder = True   # H = 256: run the unrolled decoder as ONE launch (fov_mix_decoder_fwd); False = step-wise calls

    def __init__(self, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=32, num_user=34,
                 recurrent_activation=None, seed=None, impl="auto", device="cuda", dtype="f32"):
        """dtype 'bf16' (BASELINE configs[4], latent_dim = 256 only): gate GEMMs and the Dense head take bf16
        operands on the matrix cores; accumulation, gates, cell state and the weights the optimizer updates stay fp32."""
        if dtype not in ("f32", "bf16"):
            raise ValueError("dtype must be 'f32' or 'bf16'")
        if dtype == "bf16" and int(latent_dim) != 256:
            raise ValueError("the bf16 path is built for latent_dim = 256")
        self.dtype = dtype
        self.num_encoder_tokens = 3 * cfg.fps if num_encoder_tokens is None else int(num_encoder_tokens)
        self.num_decoder_tokens = int(num_decoder_tokens)
        self.latent_dim, self.num_user = int(latent_dim), int(num_user)
        self.recurrent_activation = recurrent_activation or cfg.recurrent_activation
        rng = np.random.default_rng(seed)
        H, O = self.latent_dim, self.num_decoder_tokens
        w = {}
        for name, F in (("enc1", self.num_encoder_tokens), ("enc2", H), ("dec1", O), ("dec2", H)):
            w[name + "_K"], w[name + "_R"], w[name + "_b"] = init_lstm_weights(rng, F, H)
        w["dense_W"], w["dense_b"] = glorot_uniform(rng, H, O), np.zeros(O, np.float32)
        w["mix_W"], w["mix_b"] = glorot_uniform(rng, self.num_user * O, O), np.zeros(O, np.float32)
        self._w = w
        self._init_surface(_MIX_ORDER, impl, device)

    # ---- training surface: KerasModelSurface (given_others...py:308 compile, :500-506 fit) + fit_generator (:494-498) ----
    def _make_trainer(self, optimizer):
        from .training import OthersMixingTrainer, PaddedTrainer
        make = lambda w: OthersMixingTrainer(w, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer, lr=self._lr,
                                             device=self.device, dtype=self.dtype)
        Hp = self._run_width()
        if Hp != self.latent_dim:      # the script's latent_dim = 32: the fused H = 256 kernels on zero-padded weights (exact)
            return PaddedTrainer(make, self._w, self.latent_dim, Hp, hidden_inputs=("enc2_K", "dec2_K"))
        return make(self._w)

    def fit_generator(self, generator, steps_per_epoch, epochs=1, validation_data=None, validation_steps=None,
                      callbacks=None, use_multiprocessing=False, shuffle=True, initial_epoch=0, verbose=0):
        """Keras `fit_generator` (given_others...py:494-498): `steps_per_epoch` batches per epoch from a generator
        yielding ([enc, others, dec_in], target); validation_data may be a generator (validation_steps batches)."""
        import torch
        from .callbacks import History
        if self.optimizer is None:
            raise RuntimeError("call compile() before fit_generator()")
        tr = self._get_trainer()
        d = lambda a: torch.from_numpy(_as_f32(a)).to(self.device)
        hist = History()
        cbs = [hist] + list(callbacks or [])
        for cb in cbs:
            cb.set_model(self)
            cb.on_train_begin()
        self.stop_training = False
        for epoch in range(initial_epoch, epochs):
            tot_t, cnt = torch.zeros(1, dtype=torch.float64, device=self.device), 0   # no host synchronisation per step (_keras_fit)
            for _ in range(steps_per_epoch):
                xb, yb = next(generator)
                loss = tr.train_step(*[d(a) for a in xb], d(yb))
                tot_t.add_(loss.reshape(1).double(), alpha=float(len(yb)))
                cnt += len(yb)
            tot = float(tot_t.item())
            logs = {"loss": tot / max(cnt, 1), "lr": tr.lr}
            if validation_data is not None and validation_steps:
                vt, vc = 0.0, 0
                for _ in range(validation_steps):
                    xb, yb = next(validation_data)
                    vt += float(tr.eval_loss(*[d(a) for a in xb], d(yb)).item()) * len(yb)
                    vc += len(yb)
                logs["val_loss"] = vt / max(vc, 1)
            tr.check()
            self._w = tr.weights_numpy()
            self._dw = None
            for cb in cbs:
                cb.on_epoch_end(epoch, logs)
            if self.stop_training:
                break
        for cb in cbs:
            cb.on_train_end()
        return hist

    def _run_width(self):
        """Width the inference kernels run at.  H < 256 (the script's latent_dim = 32): 256 with zero-padded weights,
        so the prediction takes the fused path (wide-input layer kernel + ONE decoder launch) instead of per-step
        launches on the generic kernels - 8x the arithmetic at H = 32 and still far faster."""
        H = self.latent_dim
        if self.impl == "generic" or H >= 256 or not self.fused_decoder:
            return H
        return 256

    def _device_weights(self):
        import torch
        from . import ops
        if self._dw is None:
            Hp, w = self._run_width(), self._w
            if Hp != self.latent_dim:
                pw = dict(w)
                for name, hidden_in in (("enc1", False), ("enc2", True), ("dec1", False), ("dec2", True)):
                    pw[name + "_K"], pw[name + "_R"], pw[name + "_b"] = pad_lstm(w[name + "_K"], w[name + "_R"], w[name + "_b"], Hp,
                                                                                 pad_input=hidden_in)
                pw["dense_W"] = pad_rows(w["dense_W"], Hp)
                w = pw
            self._dw = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in w.items()}
            n_oth = (self.num_user - 1) * self.num_decoder_tokens
            self._dw["mix_W_oth"] = self._dw["mix_W"][:n_oth].contiguous()
            self._dw["mix_W_pred"] = self._dw["mix_W"][n_oth:].contiguous()
            self._ws = ops.Workspace()
        return self._dw

    def predict_device(self, e, oth, xin):
        """The whole prediction on device tensors: e (B,T_in,F_enc), oth (B,T_out,U-1,6), xin (B,1,6) -> (B,T_out,6)
        device tensor (a transposed view of the step-major buffer the kernels write)."""
        import torch
        from . import ops
        dw = self._device_weights()
        act, impl, ws = self.recurrent_activation, self.impl, self._ws
        H, O = self._run_width(), self.num_decoder_tokens
        B, T_in = e.shape[0], e.shape[1]
        T_out = oth.shape[1]
        if self.dtype == "bf16":   # configs[4]: both encoder layers and the fused decoder with bf16 matrix-core operands
            e3 = lambda *s_: torch.empty(s_, dtype=torch.float32, device=self.device)
            if ops.lstm_stack2_bf16_supported(B, T_in, e.shape[2], H):   # one wavefront launch for the two layers (only the final states leave)
                (_, h1, c1, _), (_, h2, c2, _) = ops.lstm_stack2_bf16(
                    e, (dw["enc1_K"], dw["enc1_R"], dw["enc1_b"]), (dw["enc2_K"], dw["enc2_R"], dw["enc2_b"]), act=act, workspace=ws,
                    out1=(None, e3(B, H), e3(B, H), None), out2=(None, e3(B, H), e3(B, H), None))
            else:
                hs1, h1, c1, _ = ops.lstm_seq_bf16(e, dw["enc1_K"], dw["enc1_R"], dw["enc1_b"], act=act, workspace=ws, reserve=False)
                _, h2, c2, _ = ops.lstm_seq_bf16(hs1, dw["enc2_K"], dw["enc2_R"], dw["enc2_b"], act=act, workspace=ws,
                                                 out=(None, e3(B, H), e3(B, H), None))
            oth_proj = ops.dense(oth.reshape(B * T_out, -1), dw["mix_W_oth"], dw["mix_b"], activation=None).reshape(B, T_out, O)
            out = torch.empty((T_out, B, O), dtype=torch.float32, device=self.device)
            ops.mix_decoder(xin, h1, c1, h2, c2, oth_proj, dw, dw["mix_W_pred"], T_out, act=act, workspace=ws, out=out, dtype="bf16")
            return out.transpose(0, 1)
        hs1, h1, c1 = ops.lstm_seq(e, dw["enc1_K"], dw["enc1_R"], dw["enc1_b"], act=act, impl=impl, workspace=ws)
        if H == 256 and impl != "generic":   # layer 2 over the 256-wide sequence: K2 and R2 register-resident
            _, h2, c2 = ops.lstm_seq(hs1, dw["enc2_K"], dw["enc2_R"], dw["enc2_b"], act=act, impl=impl,
                                     return_sequences=False, workspace=ws)
        else:                                # other widths: input projection as a GEMM, then the layer on zx
            zx = ops.matmul(hs1.reshape(B * T_in, H), dw["enc2_K"]).reshape(B, T_in, 4 * H)
            _, h2, c2 = ops.lstm_seq_zx(zx, dw["enc2_R"], dw["enc2_b"], act=act, impl=impl, return_sequences=False, workspace=ws)
        # others half of the mixing layer for every step at once (bias folded in)
        oth_proj = ops.dense(oth.reshape(B * T_out, -1), dw["mix_W_oth"], dw["mix_b"], activation=None).reshape(B, T_out, O)
        out = torch.empty((T_out, B, O), dtype=torch.float32, device=self.device)   # step-major: row t = m_t
        if self.fused_decoder and ops.mix_decoder_supported(H, O):
            # the whole unrolled decoder + mixing head in one persistent launch
            ops.mix_decoder(xin, h1, c1, h2, c2, oth_proj, dw, dw["mix_W_pred"], T_out, act=act, workspace=ws, out=out)
            return out.transpose(0, 1)
        p = torch.empty((B, O), dtype=torch.float32, device=self.device)
        fused_head = O <= 8 and H % 4 == 0
        for t in range(T_out):
            _, h1, c1 = ops.lstm_seq(xin.reshape(B, 1, O), dw["dec1_K"], dw["dec1_R"], dw["dec1_b"], h1, c1, act=act,
                                     impl=impl, return_sequences=False, workspace=ws)
            zx = ops.matmul(h1, dw["dec2_K"]).reshape(B, 1, 4 * H)
            _, h2, c2 = ops.lstm_seq_zx(zx, dw["dec2_R"], dw["dec2_b"], h2, c2, act=act, impl=impl,
                                        return_sequences=False, workspace=ws)
            if fused_head:   # Dense(tanh) + mixing Dense(tanh) in one launch, written straight into the output row
                ops.mix_head_fwd(h2, dw["dense_W"], dw["dense_b"], dw["mix_W_pred"], oth_proj[:, t], p, out[t])
            else:
                pp = ops.dense(h2, dw["dense_W"], dw["dense_b"], activation="tanh")
                ops.dense_add(pp, dw["mix_W_pred"], None, oth_proj[:, t], activation="tanh", out=out[t])
            xin = out[t]
        return out.transpose(0, 1)

    def predict(self, x, batch_size=None, verbose=0):
        import torch
        enc, others, dec0 = (_as_f32(a) for a in x)
        self._device_weights()
        n = enc.shape[0]
        T_out, O = others.shape[1], self.num_decoder_tokens
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            d = lambda a: torch.from_numpy(a[lo:lo + bs]).to(self.device)
            outs.append(self.predict_device(d(enc), d(others), d(dec0)).cpu().numpy())
        self._ws.check()
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out, O), np.float32)

    predict_on_batch = predict


_CONV_MIX_ORDER = tuple(k for k in _MIX_ORDER if not k.startswith("mix_")) + \
    ("mixc0_W", "mixc0_b", "mixc1_W", "mixc1_b", "mixc2_W", "mixc2_b")


class OthersConvMixingSeq2Seq(OthersMixingSeq2Seq):
    """The `conv_mixing` form of the same script (given_others_gt_mean_var_seq2seq.py:56,188-197,284-290): instead of the
    mixing Dense, the per-step stack [others_t (U-1,6) ; prediction (1,6)] is permuted into a 1 x 6 map with the U users
    as channels and passes Conv2D(8, (1,3)) -> Conv2D(8, (1,3)) -> Conv2D(1, (1,3)), all 'same' + relu; the single output
    channel is the step's output and the next step's input.  Runs step-wise on the layer kernels and the implicit-GEMM
    convolution (the convolution reads [others_t | prediction] as two maps: the concatenation never exists); same predict /
    fit surface and inputs as OthersMixingSeq2Seq."""

    def __init__(self, num_encoder_tokens=None, num_decoder_tokens=6, latent_dim=32, num_user=34, recurrent_activation=None,
                 seed=None, impl="auto", device="cuda", filters=8):
        super().__init__(num_encoder_tokens, num_decoder_tokens, latent_dim, num_user, recurrent_activation, seed, impl, device)
        rng = np.random.default_rng(None if seed is None else seed + 1)
        w = {k: v for k, v in self._w.items() if not k.startswith("mix_")}
        for i, (c, n) in enumerate(((self.num_user, filters), (filters, filters), (filters, 1))):
            w["mixc%d_W" % i] = glorot_uniform(rng, 3 * c, 3 * n, shape=(1, 3, c, n))     # Keras: fan_in = kh*kw*C, fan_out = kh*kw*N
            w["mixc%d_b" % i] = np.zeros(n, np.float32)
        self._w = w
        self._init_surface(_CONV_MIX_ORDER, impl, device)

    def _make_trainer(self, optimizer):
        from .training import OthersConvMixingTrainer
        return OthersConvMixingTrainer(self._w, act=self.recurrent_activation, impl=self.impl, optimizer=optimizer, lr=self._lr,
                                       device=self.device)

    def _device_weights(self):
        import torch
        from . import ops
        if self._dw is None:
            self._dw = {k: torch.from_numpy(np.ascontiguousarray(v)).to(self.device) for k, v in self._w.items()}
            self._ws = ops.Workspace()
        return self._dw

    def predict_device(self, e, oth, xin):
        import torch
        from . import ops
        dw = self._device_weights()
        act, impl, ws = self.recurrent_activation, self.impl, self._ws
        H, O = self.latent_dim, self.num_decoder_tokens
        B, T_in = e.shape[0], e.shape[1]
        T_out = oth.shape[1]
        hs1, h1, c1 = ops.lstm_seq(e, dw["enc1_K"], dw["enc1_R"], dw["enc1_b"], act=act, impl=impl, workspace=ws)
        _, h2, c2 = ops.lstm_seq(hs1, dw["enc2_K"], dw["enc2_R"], dw["enc2_b"], act=act, impl=impl, return_sequences=False, workspace=ws)
        othT = oth.permute(1, 0, 3, 2).contiguous()          # (T_out,B,6,U-1): Permute((2,1)) of every step, once
        out = torch.empty((T_out, B, O), dtype=torch.float32, device=self.device)
        xin = xin.reshape(B, O)
        for t in range(T_out):
            _, h1, c1 = ops.lstm_seq(xin.reshape(B, 1, O), dw["dec1_K"], dw["dec1_R"], dw["dec1_b"], h1, c1, act=act, impl=impl,
                                     return_sequences=False, workspace=ws)
            _, h2, c2 = ops.lstm_seq(h1.reshape(B, 1, H), dw["dec2_K"], dw["dec2_R"], dw["dec2_b"], h2, c2, act=act, impl=impl,
                                     return_sequences=False, workspace=ws)
            p = ops.dense(h2, dw["dense_W"], dw["dense_b"], activation="tanh")
            a = ops.conv2d_cat(othT[t].view(B, 1, O, -1), p.view(B, 1, O, 1), dw["mixc0_W"], dw["mixc0_b"], activation="relu")
            a = ops.conv2d(a, dw["mixc1_W"], dw["mixc1_b"], activation="relu")
            ops.conv2d(a, dw["mixc2_W"], dw["mixc2_b"], activation="relu", out=out[t].view(B, 1, O, 1))
            xin = out[t]
        return out.transpose(0, 1)


def convert_tf_lstmcell(W, b, forget_bias=1.0):
    """tf.contrib.rnn.LSTMCell variables -> Keras layout used by the kernels.
    W:(F+H,4H) fused kernel over [x, h] with gate columns i, j, f, o;  b:(4H).
    Returns (K:(F,4H), R:(H,4H), b:(4H)) with gate columns i, f, c, o and the cell's forget_bias
    folded into the f block (the kernels apply plain sigmoid, as LSTMCell does)."""
    W, b = _as_f32(W), _as_f32(b)
    H = b.shape[0] // 4
    F = W.shape[0] - H
    perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
    Wk = W[:, perm]
    bk = b[perm].copy()
    bk[H:2 * H] += np.float32(forget_bias)
    return np.ascontiguousarray(Wk[:F]), np.ascontiguousarray(Wk[F:]), bk


class StackedTFLSTM:
    """MultiRNNCell([LSTMCell(n_hidden)] * num_layers) under tf.nn.dynamic_rnn with a fed initial state
    (mycode/lstm.py:128-132,218-240), inference form (the DropoutWrapper is the identity at keep_prob 1).
    predict(x (B,T,F), init_state (L,2,B,H) or None) -> (states_series (B,T,H), current_state (L,2,B,H));
    state tuples are (c, h) as in LSTMStateTuple.
    Widths between 257 and 512 (the script's n_hidden = 400, lstm.py:59) run zero-padded at width 512 on the persistent
    register-resident kernel (lstm_wide.hip, sixteen workgroups per tile) - exact, see pad_lstm; impl='generic' or
    pad=False keeps the layer at its own width (step-wise on the GEMM / the VALU kernel)."""

    def __init__(self, cells, forget_bias=1.0, impl="auto", device="cuda", pad=True):
        self.layers = [convert_tf_lstmcell(W, b, forget_bias) for W, b in cells]
        self.impl, self.device = impl, device
        self.n_hidden = self.layers[0][1].shape[0]
        self.run_width = padded_width(self.n_hidden, MFMA_WIDTHS + (512,)) if (pad and impl != "generic") else None
        if self.run_width is None or self.n_hidden <= MFMA_WIDTHS[-1]:
            self.run_width = self.n_hidden       # the supported widths need no padding; above 512: as is
        self._dw = None

    def predict(self, x, init_state=None):
        import torch
        from . import ops
        H, Hp = self.n_hidden, self.run_width
        if self._dw is None:
            layers = self.layers
            if Hp != H:
                layers = [pad_lstm(K, R, b, Hp, pad_input=(l > 0)) for l, (K, R, b) in enumerate(layers)]
            self._dw = [tuple(torch.from_numpy(a).to(self.device) for a in layer) for layer in layers]
            self._ws = ops.Workspace()
        inp = torch.from_numpy(_as_f32(x)).to(self.device)
        B = inp.shape[0]
        st = None
        if init_state is not None:
            st = torch.zeros((len(self._dw), 2, B, Hp), dtype=torch.float32, device=self.device)
            st[..., :H] = torch.from_numpy(_as_f32(init_state)).to(self.device)
        states = []
        T, F = inp.shape[1], inp.shape[2]
        if len(self._dw) == 2 and self.impl == "auto" and ops.lstm_stack2_supported(B, T, F, Hp):
            # both layers as ONE launch, layer 2 a few steps behind layer 1 on other CUs (fov_lstm_stack2_fwd)
            sts = [None if st is None else (st[l, 1].contiguous(), st[l, 0].contiguous()) for l in range(2)]
            o1, o2 = ops.lstm_stack2(inp, self._dw[0], self._dw[1], sts[0], sts[1], act="sigmoid", workspace=self._ws)
            states = [torch.stack([o1[2], o1[1]], dim=0), torch.stack([o2[2], o2[1]], dim=0)]
            inp = o2[0]
        else:
            for l, (K, R, b) in enumerate(self._dw):
                c0 = None if st is None else st[l, 0]
                h0 = None if st is None else st[l, 1]
                hs, hT, cT = ops.lstm_seq(inp, K, R, b, h0, c0, act="sigmoid", impl=self.impl, workspace=self._ws)
                states.append(torch.stack([cT, hT], dim=0))
                inp = hs
        self._ws.check()
        return inp[..., :H].cpu().numpy(), torch.stack(states, dim=0)[..., :H].cpu().numpy()


class ConvLSTMSeq2Seq(KerasModelSurface):
    """ConvLSTM2D seq2seq (mycode/convlstm_seq2seq.py:100-282): 3-layer ConvLSTM2D encoder (filters
    2L, L, L/2, k x k, padding 'same'), mirrored decoder unrolled `predict_step` times with state hand-off,
    channel concat of the three layer outputs, head, output fed back as the next input.
      head 'conv2d' (cfg.use_one_hot, 36x18x30 maps): Conv2D 512 -> 1024 -> 30 (relu) + channel Softmax
      head 'conv1d' (xyz mode, (1,30,3) "images"): Conv1D k=7 512 -> 1024 -> 3, relu, relu, softmax
      head 'dense'  (cfg.predict_mean_var with cfg.input_mean_var, 1x1 maps of 6 channels, :171,225-227,272-273):
                    Flatten + Dense(6, linear), output (N,T_out,6) fed back as the next 1x1x6 input map
    predict([encoder_input (N,T_in,H,W,C), decoder_input (N,1,H,W,C)]) -> (N,T_out,H,W,C_out).
    compile('RMSprop', loss=costfunc._mse | 'mean_squared_error') / fit / train_on_batch train the same unrolled
    graph (convlstm_seq2seq.py:287,396-420) through training.ConvLSTMTrainer, with Keras's per-gate input
    dropout when dropout_rate > 0 (training only).  dilation_rate (default cfg.dilation_rate) spreads the taps of the
    six input convolutions, as Keras's ConvLSTM2D does (its recurrent convolution stays dense).  Weights: dict with
    enc{l}_K/R/b, dec{l}_K/R/b (Keras ConvLSTM2D layout (kh,kw,C,4F)) and head{i}_W/b."""

    _default_optimizer = "rmsprop"      # convlstm_seq2seq.py:287

    def __init__(self, weights, head="conv2d", recurrent_activation="hard_sigmoid", device="cuda", dropout_rate=0.0,
                 add_xyz_sum1=None, dilation_rate=None):
        from .training import convlstm_weight_order
        self.add_xyz_sum1 = bool(cfg.add_xyz_sum1 if add_xyz_sum1 is None else add_xyz_sum1)
        # cfg.dilation_rate (config.py:105) -> the six ConvLSTM2D layers' input convolutions (:102,110,120,148,155,162)
        self.dilation_rate = int(cfg.dilation_rate if dilation_rate is None else dilation_rate)
        if self.dilation_rate < 1:
            raise ValueError("dilation_rate must be >= 1")
        self.head, self.act = head, recurrent_activation
        self._w = {k: _as_f32(v) for k, v in weights.items()}
        self._init_surface(convlstm_weight_order(self._w), None, device)
        self.dropout_rate = float(dropout_rate)

    def compile(self, optimizer="RMSprop", loss="mean_squared_error", metrics=None):
        """Keras `compile` (convlstm_seq2seq.py:287; the heat-map fork convlstm_heatmap.py:192 compiles
        loss='categorical_crossentropy', optimizer='adam').  optimizer 'RMSprop' | 'Adam' (Keras defaults); loss
        'mean_squared_error' | 'mse' | a callable named `_mse` (cost.py:20-29) | 'categorical_crossentropy'."""
        if str(loss).lower() == "categorical_crossentropy":
            super().compile(optimizer, "mse", metrics)
            self.loss = "categorical_crossentropy"
        else:
            super().compile(optimizer, loss, metrics)
        self._trainer = None

    def _make_trainer(self, optimizer):
        from .training import ConvLSTMTrainer
        return ConvLSTMTrainer(self._w, head=self.head, act=self.act, optimizer=optimizer, lr=self._lr, device=self.device,
                               dropout_rate=self.dropout_rate, add_xyz_sum1=self.add_xyz_sum1, loss=self.loss or "mse",
                               dilation_rate=self.dilation_rate)

    def predict(self, x, batch_size=None, predict_step=None, verbose=0):
        import torch
        from . import ops
        enc, dec0 = _as_f32(x[0]), _as_f32(x[1])
        T_out = cfg.predict_step if predict_step is None else int(predict_step)
        n = enc.shape[0]
        bs = n if not batch_size else int(batch_size)
        outs = []
        for lo in range(0, n, max(bs, 1)):
            out = self.predict_device(torch.from_numpy(enc[lo:lo + bs]).to(self.device), torch.from_numpy(dec0[lo:lo + bs]).to(self.device), T_out)
            outs.append(out.cpu().numpy())
        return np.concatenate(outs, axis=0) if outs else np.zeros((0, T_out), np.float32)

    def predict_device(self, xe, dec0, predict_step=None):
        """predict on device-resident inputs: xe (B,T_in,H,W,C), dec0 (B,1,H,W,C) float32 tensors on self.device -> the
        prediction as a device tensor (B,T_out,H,W,C_out) / (B,T_out,6); no host transfer (what bench.py times)."""
        import torch
        from . import ops
        T_out = cfg.predict_step if predict_step is None else int(predict_step)
        if self._dw is None:
            self._dw = {k: torch.from_numpy(v).to(self.device) for k, v in self._w.items()}
            for side in ("enc", "dec"):     # [K ; R] stacked along the input-channel axis: one convolution per cell step
                for l in range(3):
                    K, R = self._dw["%s%d_K" % (side, l)], self._dw["%s%d_R" % (side, l)]
                    pad = (-K.shape[2]) % 4 if l == 0 else 0
                    if pad:   # zero weight rows for the zero channels the input map is padded with (16-byte pixel gather)
                        K = torch.cat([K, torch.zeros(K.shape[:2] + (pad, K.shape[3]), dtype=K.dtype, device=K.device)], 2)
                    self._dw["%s%d_KR" % (side, l)] = torch.cat([K, R], 2).contiguous()
        dw, act = self._dw, self.act
        filters = [dw["enc%d_R" % l].shape[2] for l in range(3)]
        cat = sum(filters)
        offs = [0, filters[0], filters[0] + filters[1]]
        e4 = lambda *s: torch.empty(s, dtype=torch.float32, device=self.device)
        inp = dec0[:, 0]
        B, T_in, H, W, C_in = xe.shape
        pad = (-C_in) % 4
        if pad:       # channel-pad the input maps once: 30 -> 32 keeps every pixel 16-byte aligned
            xe = torch.cat([xe, torch.zeros((B, T_in, H, W, pad), dtype=torch.float32, device=self.device)], -1)
            inp = torch.cat([inp, torch.zeros(inp.shape[:-1] + (pad,), dtype=torch.float32, device=self.device)], -1)
        # encoder: layer l runs over the whole sequence of layer l-1 (return_sequences=True)
        seq = [xe[:, t] for t in range(T_in)]
        states = []
        for l, F in enumerate(filters):
            h = torch.zeros((B, H, W, F), dtype=torch.float32, device=self.device)
            c = torch.zeros((B, H, W, F), dtype=torch.float32, device=self.device)
            KR, b = dw["enc%d_KR" % l], dw["enc%d_b" % l]
            nxt = []
            for t in range(T_in):
                hn = e4(B, H, W, F)
                # conv(x_t, K) + conv(h, R) + b, gates, c / h update: one launch
                ops.convlstm_cell(seq[t], h, KR, b, c, hn, act, dilation=self.dilation_rate)
                h = hn
                nxt.append(h)
            seq = nxt
            states.append([h, c])
        # decoder: three cells per step, each h written straight into its slot of the concat map
        dense_head = self.head == "dense"
        out = e4(B, T_out, 6) if dense_head else e4(B, T_out, H, W, dw["head2_W"].shape[3])
        for t in range(T_out):
            feat = e4(B, H, W, cat)
            cur = inp
            for l, F in enumerate(filters):
                hslot = feat[..., offs[l]:offs[l] + F]
                ops.convlstm_cell(cur, states[l][0], dw["dec%d_KR" % l], dw["dec%d_b" % l], states[l][1], hslot, act,
                                  dilation=self.dilation_rate)
                states[l][0] = hslot
                cur = hslot
            if dense_head:   # Flatten + Dense(6): cfg.predict_mean_var, output fed back as a 1x1x6 map
                y = ops.dense(feat.reshape(B, H * W * cat), dw["head0_W"], dw["head0_b"], activation=None)
                out[:, t] = y
                if pad:
                    inp[..., :C_in] = y.reshape(B, 1, 1, 6)
                else:
                    inp = y.reshape(B, 1, 1, 6)
                continue
            y = ops.conv2d(feat, dw["head0_W"], dw["head0_b"], activation="relu")
            y = ops.conv2d(y, dw["head1_W"], dw["head1_b"], activation="relu")
            y = ops.conv2d(y, dw["head2_W"], dw["head2_b"], activation="relu" if self.head == "conv2d" else None)
            y = ops.softmax_lastdim(y)
            out[:, t] = y
            if pad:
                inp[..., :C_in] = y
            else:
                inp = y
        return out

    predict_on_batch = predict
