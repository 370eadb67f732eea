"""Host-side mirror of the reference interface (no GPU): cfg knobs, data helpers against fixtures
produced by the reference's own code, the generator, and the model object's weight handling."""
import os

import numpy as np
import pytest

from longterm360fov_amd import utility as U
from longterm360fov_amd.config import cfg, default_config
from longterm360fov_amd.models import Seq2SeqLSTM


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "data_helpers.npz"))


def _datadb(g, prefix):
    return {v: {a: g["%s_%s_%s" % (prefix, v, a)].copy() for a in "xyz"} for v in ("v00", "v01")}


def test_cfg_matches_reference_values(g):
    c = default_config()
    for name, val in zip(g["cfg_names"], g["cfg_values"]):
        assert float(c[str(name)]) == float(val), name


def test_clip_and_get_data_bit_exact(g):
    db = U.clip_xyz(_datadb(g, "raw"))
    for v in db:
        for a in "xyz":
            np.testing.assert_array_equal(db[v][a], g["clip_%s_%s" % (v, a)])
    enc, fut, fut_in = U.get_data(db, pick_user=False)
    np.testing.assert_array_equal(enc, g["enc"])
    np.testing.assert_array_equal(fut, g["fut"])
    np.testing.assert_array_equal(fut_in, g["fut_in"])
    assert (enc[:, -1] == fut_in[:, 0]).all()       # the reference's own sanity check (given_others...py:315)
    np.testing.assert_array_equal(U.get_gt_target_xyz(fut), g["gt_fut"])
    np.testing.assert_array_equal(U.get_gt_target_xyz(fut_in), g["gt_fut_in"])
    np.testing.assert_array_equal(U.get_gt_target_xyz(fut.reshape(12, 10, 30, 3)), g["gt_fut_4d"])


def test_pick_user_and_others(g):
    db = _datadb(g, "clip")
    tar, tar_fut, tar_fut_in, oth, oth_fut, oth_fut_in = U.get_data(db, pick_user=True, num_user=3)
    np.testing.assert_array_equal(tar, g["pu_tar"])
    np.testing.assert_array_equal(tar_fut, g["pu_tar_fut"])
    np.testing.assert_array_equal(oth_fut, g["pu_oth_fut"])
    assert oth.shape == (2, 12, 10, 90) and oth_fut_in.shape == oth.shape
    o5 = U.reshape_others_data(oth_fut)
    assert o5.shape == (12, 10, 2, 30, 3)
    np.testing.assert_array_equal(U.get_gt_target_xyz_oth(o5), g["gt_oth_fut"])
    np.testing.assert_array_equal(U.slice_layer(1, 2, 3)(g["gt_oth_fut"]), g["slice_1_2_3"])


def test_windowing_strides(g):
    a, b, c = U.reshape2second_stacks(g["s1_in"], collapse_user=True, stride=1, purelly_testing=False)
    np.testing.assert_array_equal(a, g["s1_enc"]); np.testing.assert_array_equal(b, g["s1_fut"])
    np.testing.assert_array_equal(c, g["s1_fut_in"])
    a, b, c = U.reshape2second_stacks(g["s1_in"], collapse_user=False, stride=5, purelly_testing=False)
    np.testing.assert_array_equal(a, g["s5_enc"]); np.testing.assert_array_equal(b, g["s5_fut"])
    np.testing.assert_array_equal(c, g["s5_fut_in"])
    with pytest.raises(AssertionError):
        U.reshape2second_stacks(g["s1_in"][:, :15], collapse_user=True)    # needs >= 2 x running_length seconds


def test_others_padding_uses_np_random(g):
    db = _datadb(g, "clip")
    np.random.seed(3)
    out = U.get_data(db, pick_user=True, num_user=6)     # 2 real others padded to 5 by duplication
    assert out[3].shape == (5, 12, 10, 90)
    for u in range(2, 5):                                 # every padded user duplicates a real one
        for i in range(out[4].shape[1]):                  # (the duplicate is drawn per video/target)
            assert any((out[4][u, i] == out[4][r, i]).all() for r in range(2))


def test_generator_train2_shapes(g):
    db = _datadb(g, "clip")
    old = (cfg.batch_size, cfg.predict_mean_var, cfg.input_mean_var)
    try:
        cfg.batch_size, cfg.predict_mean_var, cfg.input_mean_var = 3, True, False
        gen = U.generator_train2(db, phase="train", num_user=4)
        (enc, oth, dec_in), tgt = next(gen)
        assert enc.shape == (2, 10, 90) and oth.shape == (2, 10, 3, 90) and dec_in.shape == (2, 1, 90)
        assert tgt.shape == (2, 10, 6)
        cfg.input_mean_var = True
        (enc, oth, dec_in), tgt = next(U.generator_train2(db, phase="train", num_user=4))
        assert enc.shape == (2, 10, 6) and oth.shape == (2, 10, 3, 6) and dec_in.shape == (2, 1, 6)
        np.testing.assert_array_equal(dec_in[:, 0], enc[:, -1])
        (_, _, _), tgt = next(U.generator_train2(db, phase="test", num_user=4))
        assert tgt.shape == (2, 10, 1, 30, 3)
    finally:
        cfg.batch_size, cfg.predict_mean_var, cfg.input_mean_var = old


def test_model_weight_handling(tmp_path):
    m = Seq2SeqLSTM(latent_dim=16, seed=1)
    w = m.get_weights()
    assert [a.shape for a in w] == [(90, 64), (16, 64), (64,), (6, 64), (16, 64), (64,), (16, 6), (6,)]
    assert m.count_params() == 90 * 64 + 16 * 64 + 64 + 6 * 64 + 16 * 64 + 64 + 16 * 6 + 6
    np.testing.assert_array_equal(w[2][16:32], 1.0)           # unit_forget_bias
    np.testing.assert_allclose(w[1] @ w[1].T, np.eye(16), atol=1e-5)   # orthogonal recurrent kernel rows
    p = str(tmp_path / "w.npz")
    m.save_weights(p)
    m2 = Seq2SeqLSTM(latent_dim=16, seed=2)
    assert not np.array_equal(m2.get_weights()[0], w[0])
    m2.load_weights(p)
    for a, b in zip(m2.get_weights(), w):
        np.testing.assert_array_equal(a, b)
    with pytest.raises(ValueError):
        m2.set_weights(w[:-1])
    with pytest.raises(ValueError):
        Seq2SeqLSTM(recurrent_activation="relu")
    m.compile(optimizer="Adam", loss="mean_squared_error")
    with pytest.raises(ValueError):
        m.compile(optimizer="sgd")


def test_missing_extension_fails_loudly(monkeypatch):
    from longterm360fov_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libfov360_hip.so")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _lib.lib()


def test_trainer_host_helpers_on_cpu():
    """Host-side pieces of the trainers that need no GPU: the ConvLSTM parameter order, the block-diagonal kernel and
    channel stacking behind Keras's per-gate input dropout (the stacked convolution equals four per-gate
    convolutions of the masked inputs), and the tf.contrib <-> Keras LSTM layout conversion."""
    import torch
    import torch.nn.functional as TF
    from longterm360fov_amd.models import convert_tf_lstmcell
    from longterm360fov_amd.training import ConvLSTMTrainer, convlstm_weight_order
    w = {"%s%d_%s" % (s, l, n): 0 for s in ("enc", "dec") for l in range(3) for n in "KRb"}
    w.update({"head0_W": 0, "head0_b": 0, "head1_W": 0, "head1_b": 0})
    order = convlstm_weight_order(w)
    assert order[:3] == ["enc0_K", "enc0_R", "enc0_b"] and order[-4:] == ["head0_W", "head0_b", "head1_W", "head1_b"] and len(order) == 22
    rng = np.random.default_rng(0)
    B, H, W, C, F = 2, 5, 4, 3, 2
    K = torch.tensor(rng.standard_normal((3, 3, C, 4 * F)))
    x = torch.tensor(rng.standard_normal((B, H, W, C)))
    m4 = torch.tensor((rng.random((4, B, H, W, C)) < 0.7) / 0.7)
    conv = lambda a, k: TF.conv2d(a.permute(0, 3, 1, 2), k.permute(3, 2, 0, 1), padding=1).permute(0, 2, 3, 1)
    stacked = conv(ConvLSTMTrainer._stack_masked(x, m4), ConvLSTMTrainer._block_diag_kernel(K))
    per_gate = torch.cat([conv(x * m4[g], K[..., g * F:(g + 1) * F]) for g in range(4)], -1)
    assert torch.allclose(stacked, per_gate, atol=1e-12)
    # tf.contrib LSTMCell (i, j, f, o; forget_bias inside the cell) -> Keras layout (i, f, c, o; bias carries it)
    Hh, Fi = 5, 3
    Wt = rng.standard_normal((Fi + Hh, 4 * Hh)).astype(np.float32)
    bt = rng.standard_normal(4 * Hh).astype(np.float32)
    Kk, Rk, bk = convert_tf_lstmcell(Wt, bt, forget_bias=1.0)
    xx, hh = rng.standard_normal((2, Fi)), rng.standard_normal((2, Hh))
    z_tf = np.concatenate([xx, hh], 1) @ Wt + bt
    z_k = xx @ Kk + hh @ Rk + bk
    i, j, f, o = (z_tf[:, k * Hh:(k + 1) * Hh] for k in range(4))
    assert np.allclose(z_k[:, :Hh], i, atol=1e-5) and np.allclose(z_k[:, Hh:2 * Hh], f + 1.0, atol=1e-5)
    assert np.allclose(z_k[:, 2 * Hh:3 * Hh], j, atol=1e-5) and np.allclose(z_k[:, 3 * Hh:], o, atol=1e-5)


def test_pickled_interchange_round_trip(tmp_path):
    """§8(f) rank 4, the pickled half: a Python-2 style dataset pickle (protocol 2) loads and is clipped; the
    decoded_sentence / gt_sentence_list files are plain pickles of per-batch arrays as the reference's plotting scripts read."""
    import pickle
    from longterm360fov_amd import utility as U
    rng = np.random.default_rng(0)
    db = {"v%d" % i: {a: rng.uniform(-1.2, 1.2, (3, 90)) for a in "xyz"} for i in range(2)}
    p = tmp_path / "db.p"
    with open(p, "wb") as f:
        pickle.dump(db, f, protocol=2)
    got = U.load_datadb(str(p))
    assert set(got) == set(db) and all(np.abs(got[k][a]).max() <= 1.0 for k in got for a in "xyz")
    dec = [rng.standard_normal((4, 10, 6)).astype(np.float32) for _ in range(3)]
    gt = [rng.standard_normal((4, 10, 90)) for _ in range(3)]
    a, b = U.save_decoded_sentences(dec, gt, tag="_t", directory=str(tmp_path))
    assert a.endswith("decoded_sentence_t.p") and b.endswith("gt_sentence_list_t.p")
    back = pickle.load(open(a, "rb"))
    assert len(back) == 3 and np.array_equal(back[1], dec[1]) and np.array_equal(pickle.load(open(b, "rb"))[2], gt[2])


# ---------------------------------------------------------------------------------------------------------------
# f4: Keras HDF5 weight files (longterm360fov_amd/keras_h5.py)
# ---------------------------------------------------------------------------------------------------------------
def _real_h5_value(k, shape):
    """Values tests/golden/make_keras_h5_real.c wrote into the k-th tensor of its files."""
    i = np.arange(int(np.prod(shape)), dtype=np.int64)
    return (((37 * i + 11 * k) % 1000 - 500) / 256.0).astype(np.float32).reshape(shape)


@pytest.mark.parametrize("name", ["keras_real_weights.h5", "keras_real_model.h5", "keras_real_weights_vlen.h5",
                                  "keras_real_model_vlen.h5"])
def test_keras_h5_reader_against_files_written_by_the_hdf5_library(golden_dir, name):
    """Ground truth: files written by the real HDF5 C library (1.10.6) in the Keras layout - model.save_weights form and
    model.save form (tree under model_weights/).  The *_vlen files carry backend / keras_version / model_config /
    training_config as VARIABLE-LENGTH strings (global-heap references): what h5py writes for a Python bytes / str scalar,
    i.e. what a file saved by Keras really looks like."""
    from longterm360fov_amd import keras_h5
    layers = keras_h5.read_keras_layers(os.path.join(golden_dir, name))
    assert [n for n, _ in layers] == ["input_1", "input_2", "lstm_1", "lstm_2", "dense_1"]
    assert [w for w, _ in layers[2][1]] == ["lstm_1/kernel:0", "lstm_1/recurrent_kernel:0", "lstm_1/bias:0"]
    assert [a.shape for _, a in layers[2][1]] == [(5, 16), (4, 16), (16,)]
    assert [a.shape for _, a in layers[4][1]] == [(4, 3), (3,)]
    k = 0
    for _, ws in layers:
        for _, a in ws:
            assert a.dtype == np.float32
            np.testing.assert_array_equal(a, _real_h5_value(k, a.shape))
            k += 1
    assert k == 8


def test_keras_h5_variable_length_string_attributes(golden_dir):
    """The scalar attributes Keras adds are decoded through the global heap, not merely skipped; an attribute whose datatype
    the reader cannot decode is reported as None instead of failing the file."""
    from longterm360fov_amd import keras_h5
    with open(os.path.join(golden_dir, "keras_real_model_vlen.h5"), "rb") as fh:
        f = keras_h5._File(fh.read())
    attrs = f.attributes(f.root)
    assert attrs["backend"].item() == b"tensorflow" and attrs["keras_version"].item() == b"2.2.4"
    assert attrs["model_config"].item().startswith(b'{"class_name": "Model"')
    assert b"mean_squared_error" in attrs["training_config"].item()
    assert f.skipped_attributes == []
    # corrupt the datatype class of one attribute message (class 9 -> 6, compound): skipped, everything else still reads
    data = bytearray(open(os.path.join(golden_dir, "keras_real_weights_vlen.h5"), "rb").read())
    f2 = keras_h5._File(data)
    hit = 0
    for mtype, _, body, _ in f2.messages(f2.root["header"]):
        if mtype == 0x000C and f2._attribute_name(body) == "backend":
            nsz = f2.u16(body + 2)
            o = body + 8 + ((nsz + 7) & ~7 if f2.u8(body) == 1 else nsz + (1 if f2.u8(body) == 3 else 0))
            assert data[o] & 0x0F == 9
            data[o] = (data[o] & 0xF0) | 6
            hit += 1
    assert hit == 1
    f3 = keras_h5._File(bytes(data))
    a3 = f3.attributes(f3.root)
    assert a3["backend"] is None and [n for n, _ in f3.skipped_attributes] == ["backend"]
    assert list(keras_h5._decode(a3["layer_names"])) == ["input_1", "input_2", "lstm_1", "lstm_2", "dense_1"]


def test_keras_h5_loads_into_the_model_object_and_refuses_other_topologies(golden_dir):
    from longterm360fov_amd.models import Seq2SeqLSTM
    m = Seq2SeqLSTM(num_encoder_tokens=5, num_decoder_tokens=3, latent_dim=4, seed=0)
    m.load_weights(os.path.join(golden_dir, "keras_real_weights.h5"))
    for k, a in enumerate(m.get_weights()):
        np.testing.assert_array_equal(a, _real_h5_value(k, a.shape))
    m.load_weights(os.path.join(golden_dir, "keras_real_model.h5"))       # model.save() form
    wrong = Seq2SeqLSTM(num_encoder_tokens=5, num_decoder_tokens=3, latent_dim=8, seed=0)
    with pytest.raises(ValueError):
        wrong.load_weights(os.path.join(golden_dir, "keras_real_weights.h5"))


def test_keras_h5_writer_round_trip_and_fixture(golden_dir, tmp_path):
    """save_weights('x.h5') writes a Keras-layout HDF5 file (checked with the real library's h5ls / h5dump when the
    fixture was made); load_weights reads it back bit for bit; the committed fixture is what the writer produces."""
    import importlib.util
    from longterm360fov_amd import keras_h5
    from longterm360fov_amd.models import Seq2SeqLSTM
    m = Seq2SeqLSTM(num_encoder_tokens=7, num_decoder_tokens=6, latent_dim=12, seed=3)
    for name in ("w.h5", "w.hdf5", "w.npz", "w"):
        p = str(tmp_path / name)
        m.save_weights(p)
        assert os.path.exists(m.weights_path(p))
        m2 = Seq2SeqLSTM(num_encoder_tokens=7, num_decoder_tokens=6, latent_dim=12, seed=4)
        m2.load_weights(p)
        for a, b in zip(m.get_weights(), m2.get_weights()):
            np.testing.assert_array_equal(a, b)
    layers = keras_h5.read_keras_layers(str(tmp_path / "w.h5"))
    assert [n for n, _ in layers] == ["enc", "dec", "dense"]
    assert [w for w, _ in layers[0][1]] == ["enc/kernel:0", "enc/recurrent_kernel:0", "enc/bias:0"]
    spec = importlib.util.spec_from_file_location("mk", os.path.join(golden_dir, "make_keras_h5_fixture.py"))
    mk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mk)
    keras_h5.write_keras_layers(str(tmp_path / "tiny.h5"), mk.tiny_layers())
    assert open(str(tmp_path / "tiny.h5"), "rb").read() == open(os.path.join(golden_dir, "keras_seq2seq_tiny.h5"), "rb").read()
    ref = mk.tiny_layers()
    for fixture in ("keras_seq2seq_tiny.h5", "keras_seq2seq_tiny_full.h5"):
        got = keras_h5.read_keras_layers(os.path.join(golden_dir, fixture))
        assert [n for n, _ in got] == [n for n, _ in ref]
        for (_, ws), (_, ws2) in zip(got, ref):
            for (wn, a), (wn2, b) in zip(ws, ws2):
                assert wn == wn2
                np.testing.assert_array_equal(a, b)


def test_model_checkpoint_h5_name_round_trips(tmp_path):
    """ModelCheckpoint('...{epoch:02d}-{val_loss:.4f}.h5') (given_others...py:484) and load_weights of the same name."""
    from longterm360fov_amd.callbacks import ModelCheckpoint
    from longterm360fov_amd.models import Seq2SeqLSTM
    m = Seq2SeqLSTM(num_encoder_tokens=7, num_decoder_tokens=6, latent_dim=8, seed=5)
    cb = ModelCheckpoint(str(tmp_path / "fov{epoch:02d}-{val_loss:.4f}.h5"))
    cb.set_model(m)
    cb.on_epoch_end(2, {"val_loss": 0.1234})
    path = str(tmp_path / "fov03-0.1234.h5")
    assert cb.saved == [path] and os.path.exists(path)
    m2 = Seq2SeqLSTM(num_encoder_tokens=7, num_decoder_tokens=6, latent_dim=8, seed=6)
    m2.load_weights(path)
    for a, b in zip(m.get_weights(), m2.get_weights()):
        np.testing.assert_array_equal(a, b)


def test_lstm_py_driver_loop_on_a_recording_trainer(tmp_path):
    """lstm_driver.LSTMPyDriver's host logic without a GPU (the arithmetic lives in the trainer): the order of calls of
    mycode/lstm.py:583-660 - save + learning-rate change on even epochs BEFORE the epoch's steps, the state handed from step to
    step, display steps by the script's `count` rule (the mean / variance branch takes the display step's state), the final
    save - and :590-592's restore-then-start-at-`training_epochs` quirk; total_batch_of against the script's formula."""
    import torch
    from longterm360fov_amd.config import default_config
    from longterm360fov_amd.lstm_driver import LSTMPyDriver, total_batch_of

    class Recorder:
        L, H, device, head_kind = 2, 4, "cpu", "meanvar"

        def __init__(self):
            self.lr, self.log, self.n = 1e-3, [], 0

        def train_step(self, x, y, state, masks=None, n_global=None, state_view=False):
            assert state_view
            self.n += 1
            self.log.append(("step", self.n, float(state.sum()), self.lr, None if masks is None else len(masks)))
            return torch.tensor([1.0 / self.n]), state + 1.0

        def eval_loss(self, x, y, state, masks=None, state_view=False):
            self.log.append(("eval", self.n, float(state.sum())))
            return torch.tensor([0.5]), state + 100.0

        def check(self):
            self.log.append(("check", self.n))

        def state_dict(self):
            return {"Variable": np.float64(self.lr), "n": np.int64(self.n)}

        def load_state_dict(self, sd):
            self.lr, self.n = float(sd["Variable"]), int(sd["n"])

    class Data:
        def _get_next_minibatch(self, datadb, batch_size):
            return (np.zeros((batch_size, 3, 6), np.float32), np.zeros((batch_size, 1, 6), np.float32), None, np.zeros((batch_size, 2, 6)), None, None)

    cfg = default_config()
    cfg.LEARNING_RATE, cfg.lr_epoch_step = 1e-3, 10
    tr = Recorder()
    drv = LSTMPyDriver(tr, cfg, model_path=str(tmp_path / "LSTM_x.ckpt"), dropout=0.1)
    drv.fit(Data(), total_batch=3, training_epochs=3, batch_size=5)
    steps = [e for e in tr.log if e[0] == "step"]
    assert len(steps) == 9 and all(e[4] == 1 for e in steps)                       # one DropoutWrapper mask (L - 1 layers) per step
    assert [e[3] for e in steps[:3]] == [1e-3 * 0.5 ** 0.2] * 3 and [e[3] for e in steps[3:6]] == [1e-3 * 0.5 ** 0.2] * 3   # epoch 2: set; epoch 3: kept
    assert [e[3] for e in steps[6:]] == [1e-3 * 0.5 ** 0.4] * 3                  # epoch 4: LEARNING_RATE * 0.5 ** (4 / 10)
    assert [os.path.basename(p) for p in drv.saved] == ["LSTM_xepoch2.ckpt.npz", "LSTM_xepoch4.ckpt.npz", "LSTM_xepoch4.ckpt.npz"]
    # count = (step + 1) * batch_size + epoch * total_batch * batch_size; display_step 10 below 200: epoch 2 -> 35, 40, 45 ...
    evals = [e for e in tr.log if e[0] == "eval"]
    counts = [(s + 1) * 5 + ep * 15 for ep in (2, 3, 4) for s in range(3)]
    assert len(evals) == sum(1 for c in counts if c % (10 if c < 200 else 200) == 0) == len(drv.history)
    assert [c for c, _ in drv.history] == [c for c in counts if c % 10 == 0]
    # the state is carried (each step adds 1 to every element, a display step on this branch adds 100): strictly increasing sums
    sums = [e[2] for e in steps]
    assert sums[0] == 0.0 and all(b > a for a, b in zip(sums, sums[1:])) and sums[-1] >= 8 * 2 * 2 * 5 * 4
    assert [e for e in tr.log if e[0] == "check"] == [("check", 3), ("check", 6), ("check", 9)]
    # restore-then-start-at-training_epochs: the checkpoint of epoch starting_epoch - 1 = 1 exists
    os.replace(drv.epoch_path(2) + ".npz", drv.epoch_path(1) + ".npz")
    tr2 = Recorder()
    tr2.lr = 123.0
    d2 = LSTMPyDriver(tr2, cfg, model_path=str(tmp_path / "LSTM_x.ckpt"), dropout=0.0)
    d2.fit(Data(), total_batch=2, training_epochs=5, batch_size=5)
    st2 = [e for e in tr2.log if e[0] == "step"]
    assert tr2.n == 0 + 10 and len(st2) == 10                                      # restored (n = 0 at that save), epochs 5..9
    assert st2[0][3] == 1e-3 and st2[0][4] is None                                 # epoch 5 is odd: the restored rate stays; no masks at dropout 0
    assert st2[2][3] == 1e-3 * 0.5 ** 0.6 and st2[-1][3] == 1e-3 * 0.5 ** 0.8     # epochs 6 and 8
    assert [os.path.basename(p) for p in d2.saved] == ["LSTM_xepoch6.ckpt.npz", "LSTM_xepoch8.ckpt.npz", "LSTM_xepoch9.ckpt.npz"]
    datadb = {0: {"x": np.zeros((48, 1800))}, 1: {"x": np.zeros((48, 3600))}, 2: {"x": np.zeros((48, 900))}}
    cfg.test_video_ind = 2
    assert total_batch_of(datadb, cfg) == int((5400 - 300) / 10 / 32) * 48
