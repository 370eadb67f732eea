"""Every run-time FOV_* knob that selects another kernel form or launch arrangement, against the default path on the same
inputs (KNOBS.md lists all knobs and the test that covers each; the ones exercised in other test files are named there).
A knob changes the ORDER of floating-point sums at most, never the arithmetic: gradients agree to 2e-5 of each tensor's scale
(fp32) / 2e-3 (bf16 operands), losses to 1e-6 / 1e-4 relative.  A knob that silently did nothing would pass here, so every case
also states what must differ or where the knob is read (the library re-reads its environment through ops._sync_env)."""
import os

import numpy as np
import pytest
import torch

from oracle import fov_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _mixing(dtype, B=48, U=5, T_in=3, T_out=3):
    from longterm360fov_amd.training import OthersMixingTrainer
    w = O.init_others_mixing(70, H=256, num_user=U, bias_noise=0.1)
    enc, dec0, tgt, oth = O.synthetic_batch(71, B, T_in, T_out, num_others=U - 1)
    tr = OthersMixingTrainer(w, dtype=dtype)
    loss, _ = tr.forward_backward(dev(enc), dev(oth), dev(dec0), dev(tgt))
    l0 = float(loss.item())
    g0 = {k: v.clone() for k, v in tr.g.items()}
    l1 = float(tr.train_step(dev(enc), dev(oth), dev(dec0), dev(tgt)).item())
    tr.check()
    return l0, l1, g0


def _seq2seq(H=128, B=32, T_in=5, T_out=4):
    from longterm360fov_amd.training import Seq2SeqTrainer
    w = O.init_seq2seq(80 + H, H=H, bias_noise=0.05)
    enc, dec0, tgt = O.synthetic_batch(81, B, T_in, T_out)
    dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
    tr = Seq2SeqTrainer(w)
    loss, _ = tr.forward_backward(dev(enc), dev(dec_in), dev(tgt))
    l0 = float(loss.item())
    g0 = {k: v.clone() for k, v in tr.g.items()}
    l1 = float(tr.train_step(dev(enc), dev(dec_in), dev(tgt)).item())
    tr.check()
    return l0, l1, g0


def _tf_lstm():      # lstm.py's stack at its batch: few rows, H = 512 after padding -> the grouped weight-gradient launch
    from test_gpu_lstm_driver import _batch, _trainer
    x, y, init = _batch(91, 32, 10)
    tr = _trainer(93, 400, "meanvar")
    loss, _, _, _ = tr.forward_backward(dev(x), dev(y), dev(init))
    l0 = float(loss.item())
    g0 = {k: v.clone() for k, v in tr.g.items()}
    l1 = float(tr.train_step(dev(x), dev(y), dev(init))[0].item())
    tr.check()
    return l0, l1, g0


WORK = {"mixing_f32": lambda: _mixing("f32"), "mixing_bf16": lambda: _mixing("bf16"), "mixing_f32_512": lambda: _mixing("f32", B=512, U=34, T_in=4, T_out=4),
        "mixing_bf16_512": lambda: _mixing("bf16", B=512, U=34, T_in=4, T_out=4), "seq2seq_128": _seq2seq, "seq2seq_256": lambda: _seq2seq(256, 40),
        "tf_lstm": _tf_lstm}

CASES = [
    # fp32 GEMM (gemm_f32_kernel): tile shape / K slices forced
    ("FOV_GEMM_VARIANT", "2", "mixing_f32"), ("FOV_GEMM_VARIANT", "3", "mixing_f32_512"), ("FOV_GEMM_VARIANT", "4", "mixing_f32_512"),
    ("FOV_GEMM_SPLIT", "1", "mixing_f32_512"), ("FOV_GEMM_SPLIT", "7", "mixing_f32_512"),
    # bf16 weight-gradient GEMM (gemm_bf16.hip): K slices, two-stage pipeline, no XCD-aware tile order
    ("FOV_GEMM_BF16_SPLIT", "3", "mixing_bf16_512"), ("FOV_GEMM_BF16_SHALLOW", "1", "mixing_bf16_512"), ("FOV_GEMM_BF16_NOREMAP", "1", "mixing_bf16_512"),
    # the three weight gradients of a layer as three products instead of one (fov_lstm_seq_bwd)
    ("FOV_NO_WGRAD_FUSION", "1", "seq2seq_256"), ("FOV_NO_WGRAD_FUSION", "1", "mixing_f32"),
    # the stacked encoder layer's data gradient dz K^T inside the fp32 eight-workgroup BPTT kernel switched off (a GEMM + reduce instead)
    ("FOV_NO_DX_FUSION", "1", "mixing_f32_512"), ("FOV_NO_DX_FUSION", "1", "mixing_f32"),
    # all weight gradients of a few-row layer stack / encoder-decoder pair in one launch (wgrad_group.hip: tile and row-split forms) switched off
    ("FOV_NO_WGRAD_GROUP", "1", "tf_lstm"), ("FOV_NO_WGRAD_GROUP", "1", "seq2seq_128"), ("FOV_NO_WGRAD_GROUP", "1", "mixing_f32"),
    # BPTT kernel family: four-workgroup groups instead of eight (H = 256, <= 512 sequences); the sixteen-unit narrow form off
    ("FOV_BWD_GROUPS4", "1", "mixing_f32_512"), ("FOV_NO_BWD16_NARROW", "1", "seq2seq_128"), ("FOV_NO_BWD16_NARROW", "1", "seq2seq_256"),
    # grid padding to whole XCDs limited to groups of at most N members
    ("FOV_XCD_PAD_MAX", "4", "seq2seq_128"), ("FOV_XCD_PAD_MAX", "32", "tf_lstm"),
    # the mixing trainer's second stream: off / on, which products go there, its priority
    ("FOV_WGRAD_STREAM", "0", "mixing_f32_512"), ("FOV_WGRAD_STREAM", "1", "mixing_bf16_512"), ("FOV_WGRAD_SPLIT", "1", "mixing_f32_512"),
    ("FOV_WGRAD_SPLIT", "0", "mixing_bf16_512"), ("FOV_WGRAD_ENC_SIDE", "1", "mixing_f32_512"), ("FOV_SIDE_PRIORITY", "normal", "mixing_f32_512"),
    # deferred split reductions: off, and an arena that holds almost nothing
    ("FOV_DEFER_ARENA_MB", "0", "mixing_f32_512"), ("FOV_DEFER_ARENA_MB", "1", "seq2seq_256"),
]


@pytest.mark.parametrize("knob,value,work", CASES)
def test_knob_path_agrees_with_the_default_path(knob, value, work):
    # (the mixing trainer takes its side stream only for shapes it has seen before: the train_step behind forward_backward does)
    assert knob not in os.environ
    ref = WORK[work]()
    os.environ[knob] = value
    try:
        got = WORK[work]()
    finally:
        del os.environ[knob]
    bf16 = "bf16" in work
    tol_l, tol_g = (1e-4, 2e-3) if bf16 else (1e-6, 2e-5)
    assert abs(got[0] - ref[0]) <= tol_l * abs(ref[0]) + 1e-9 and abs(got[1] - ref[1]) <= 10 * tol_l * abs(ref[1]) + 1e-9, (got[:2], ref[:2])
    worst = 0.0
    for k in ref[2]:
        scale = ref[2][k].abs().max().item()
        d = (got[2][k] - ref[2][k]).abs().max().item()
        worst = max(worst, d / (scale + 1e-30))
        assert d <= tol_g * scale + 1e-9, (knob, value, k, d, scale)
    print("%s=%s on %s: worst gradient difference %.2e of its scale" % (knob, value, work, worst))


def test_fit_with_the_training_set_streamed_from_the_host(tmp_path):
    """FOV_FIT_RESIDENT_BYTES=0: model.fit uploads every batch instead of keeping the training set on the device - same
    losses and weights after two epochs, bit for bit (the batches are the same arrays)."""
    from longterm360fov_amd.models import Seq2SeqLSTM

    def run():
        m = Seq2SeqLSTM(latent_dim=64)
        m.compile(optimizer="Adam", loss="mean_squared_error")
        enc, dec0, tgt = O.synthetic_batch(55, 96, 5, 4)
        dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
        w0 = O.init_seq2seq(56, H=64, bias_noise=0.05)
        m.set_weights([w0[k] for k in ("enc_K", "enc_R", "enc_b", "dec_K", "dec_R", "dec_b", "dense_W", "dense_b")])
        h = m.fit([enc, dec_in], tgt, batch_size=32, epochs=2, validation_split=0.25, shuffle=False, verbose=0)
        return h.history["loss"], h.history["val_loss"], m.get_weights()

    ref = run()
    os.environ["FOV_FIT_RESIDENT_BYTES"] = "0"
    try:
        got = run()
    finally:
        del os.environ["FOV_FIT_RESIDENT_BYTES"]
    assert got[0] == ref[0] and got[1] == ref[1]
    for a, b in zip(got[2], ref[2]):
        np.testing.assert_array_equal(a, b)


def test_launch_trace_names_every_launch(capfd):
    """FOV_DBG_TRACE=1: one stderr line per launch of the training kernels' host helpers (with HIP_LAUNCH_BLOCKING=1 the last line
    names the launch in front of a faulting one); off again, nothing is printed."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(0)
    y = dev(np.tanh(rng.standard_normal((64, 6))))
    tg = dev(rng.uniform(-1, 1, (64, 6)))
    os.environ["FOV_DBG_TRACE"] = "1"
    try:
        ops.mse_dense_grad(y, tg, "tanh", scratch=ops.Scratch())      # (a scratch fetch is where the wrapper syncs the library's knobs)
        torch.cuda.synchronize()
    finally:
        del os.environ["FOV_DBG_TRACE"]
    err = capfd.readouterr().err
    assert "[fov trace] launched: mse_dense_grad" in err
    ops.mse_dense_grad(y, tg, "tanh", scratch=ops.Scratch())
    torch.cuda.synchronize()
    assert "[fov trace]" not in capfd.readouterr().err
