"""GPU parity of the bf16 path (BASELINE.json configs[4]: the config-3 model with bf16 operands into
v_mfma_f32_16x16x32_bf16, fp32 accumulate / gates / cell state / master weights), through the C ABI.

Two references, both from oracle/fov_oracle.py in fp64:
  * the restatement under `bf16_operands()` - it rounds the operands of exactly those matrix products that the HIP
    path runs in bf16 (round-to-nearest-even, as v_cvt_pk_bf16_f32).  The kernels must match it TIGHTLY (bound 1e-3
    absolute on values in (-1, 1); measured ~1e-4: what is left are accumulation order, the fast exp/rcp and the rare
    flip of an h value that sits on a bf16 rounding boundary);
  * the full-precision restatement: north_star's 1e-3 relative bound cannot hold for bf16 operands (8 mantissa bits,
    20 recurrent steps).  The bound stated and asserted here is 5e-2 absolute on the tanh-range outputs; the measured
    maximum is printed (about 1e-2).
"""
import numpy as np
import pytest
import torch

from oracle import fov_oracle as O

pytestmark = pytest.mark.gpu

TIGHT = 1e-3     # vs the bf16-operand restatement
LOOSE = 5e-2     # vs the full-precision restatement


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def f64(w):
    return {k: v.astype(np.float64) for k, v in w.items()}


@pytest.mark.parametrize("B,T,F,act,state", [(37, 5, 90, "sigmoid", False), (16 * 33 + 5, 3, 256, "sigmoid", True),
                                             (21, 4, 6, "hard_sigmoid", True), (1, 1, 256, "sigmoid", False),
                                             (48, 2, 200, "hard_sigmoid", False)])
def test_bf16_layer_matches_bf16_operand_oracle(B, T, F, act, state):
    from longterm360fov_amd import ops
    rng = np.random.default_rng(B * 131 + T)
    H = 256
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    ws = ops.Workspace()
    hs, hT, cT, res = ops.lstm_seq_bf16(dev(x), dev(K), dev(R), dev(b), None if h0 is None else dev(h0),
                                        None if c0 is None else dev(c0), act=act, workspace=ws)
    ws.check()
    to64 = lambda a: None if a is None else a.astype(np.float64)
    with O.bf16_operands():
        rhs, rh, rc = O.lstm_layer(to64(x), to64(K), to64(R), to64(b), to64(h0), to64(c0), act=act)
    fhs, _, _ = O.lstm_layer(to64(x), to64(K), to64(R), to64(b), to64(h0), to64(c0), act=act)
    e_t = max(np.abs(hs.cpu().numpy() - rhs).max(), np.abs(hT.cpu().numpy() - rh).max(), np.abs(cT.cpu().numpy() - rc).max())
    e_l = np.abs(hs.cpu().numpy() - fhs).max()
    print("bf16 layer B=%d T=%d F=%d %s: vs bf16-operand oracle %.2e, vs fp64 oracle %.2e" % (B, T, F, act, e_t, e_l))
    assert e_t <= TIGHT and e_l <= LOOSE
    # the tape the backward pass reads: reserve = activated i, f, g, o and c of every step; h = o * tanh(c)
    r = res.cpu().numpy()
    np.testing.assert_allclose(r[:, :, 3] * np.tanh(r[:, :, 4]), hs.cpu().numpy(), atol=2e-6)
    np.testing.assert_allclose(r[:, -1, 4], cT.cpu().numpy(), atol=0)


def test_bf16_layer_is_deterministic_and_batch_independent():
    from longterm360fov_amd import ops
    rng = np.random.default_rng(5)
    B, T, F, H = 100, 6, 90, 256
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    ws = ops.Workspace()
    a = ops.lstm_seq_bf16(dev(x), dev(K), dev(R), dev(b), workspace=ws)[0].cpu().numpy()
    b2 = ops.lstm_seq_bf16(dev(x), dev(K), dev(R), dev(b), workspace=ws)[0].cpu().numpy()
    perm = rng.permutation(B)
    c = ops.lstm_seq_bf16(dev(x[perm]), dev(K), dev(R), dev(b), workspace=ws)[0].cpu().numpy()
    d = ops.lstm_seq_bf16(dev(x[:23]), dev(K), dev(R), dev(b), workspace=ws)[0].cpu().numpy()
    ws.check()
    np.testing.assert_array_equal(a, b2)           # bitwise repeatable
    np.testing.assert_array_equal(a[perm], c)      # a sequence does not see its tile-mates
    np.testing.assert_array_equal(a[:23], d)


@pytest.mark.parametrize("B,T_in,T_out,act", [(37, 4, 5, "sigmoid"), (21, 3, 3, "hard_sigmoid"), (16 * 32 + 9, 2, 2, "sigmoid")])
def test_bf16_mixing_model_matches_bf16_operand_oracle(B, T_in, T_out, act):
    """Whole configs[4] inference path: two bf16 encoder layers + the fused bf16 decoder launch."""
    from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
    U = 34
    w = O.init_others_mixing(11, H=256, num_user=U, bias_noise=0.05)
    enc, dec0, _, oth = O.synthetic_batch(12, B, T_in, T_out, num_others=U - 1)
    m = OthersMixingSeq2Seq(latent_dim=256, num_user=U, recurrent_activation=act, dtype="bf16")
    m.set_weights([w[k] for k in _MIX_ORDER])
    got = m.predict([enc, oth, dec0])
    with O.bf16_operands():
        ref_t = O.others_mixing_forward(enc.astype(np.float64), oth.astype(np.float64), dec0.astype(np.float64), f64(w), act=act)
    ref_l = O.others_mixing_forward(enc.astype(np.float64), oth.astype(np.float64), dec0.astype(np.float64), f64(w), act=act)
    e_t, e_l = np.abs(got - ref_t).max(), np.abs(got - ref_l).max()
    print("bf16 mixing model B=%d %d->%d %s: vs bf16-operand oracle %.2e, vs fp64 oracle %.2e" % (B, T_in, T_out, act, e_t, e_l))
    assert e_t <= TIGHT and e_l <= LOOSE
    # fp32 path of the same object family for scale: bf16 and fp32 outputs differ by the operand rounding only
    m32 = OthersMixingSeq2Seq(latent_dim=256, num_user=U, recurrent_activation=act)
    m32.set_weights([w[k] for k in _MIX_ORDER])
    assert np.abs(m32.predict([enc, oth, dec0]) - got).max() <= LOOSE


def test_bf16_config5_full_size_and_properties():
    """configs[4] at its per-GPU size (512 sequences, T 10 -> 10, 33 others): 64 sequences against both oracles, all 512
    through size-independent properties (bitwise repeatability, batch-permutation equivariance, output range)."""
    from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
    B, T, U = 512, 10, 34
    w = O.init_others_mixing(21, H=256, num_user=U, bias_noise=0.05)
    enc, dec0, _, oth = O.synthetic_batch(22, B, T, T, num_others=U - 1)
    m = OthersMixingSeq2Seq(latent_dim=256, num_user=U, dtype="bf16")
    m.set_weights([w[k] for k in _MIX_ORDER])
    got = m.predict([enc, oth, dec0])
    n = 64
    with O.bf16_operands():
        ref_t = O.others_mixing_forward(enc[:n].astype(np.float64), oth[:n].astype(np.float64), dec0[:n].astype(np.float64), f64(w))
    ref_l = O.others_mixing_forward(enc[:n].astype(np.float64), oth[:n].astype(np.float64), dec0[:n].astype(np.float64), f64(w))
    e_t, e_l = np.abs(got[:n] - ref_t).max(), np.abs(got[:n] - ref_l).max()
    print("bf16 config 5 full size: vs bf16-operand oracle %.2e, vs fp64 oracle %.2e (max), %.2e (mean)"
          % (e_t, e_l, np.abs(got[:n] - ref_l).mean()))
    assert e_t <= TIGHT and e_l <= LOOSE
    assert np.isfinite(got).all() and np.abs(got).max() < 1.0
    np.testing.assert_array_equal(got, m.predict([enc, oth, dec0]))
    perm = np.random.default_rng(3).permutation(B)
    np.testing.assert_array_equal(got[perm], m.predict([enc[perm], oth[perm], dec0[perm]]))
