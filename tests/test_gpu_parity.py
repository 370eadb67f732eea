"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on identical seeded
inputs.  Tolerance (BASELINE.json north_star): outputs within 1e-3 relative of the reference CPU
path on fp32 trajectory coordinates; asserted here as |gpu - ref| <= 1e-3 * |ref| + 1e-5, and
additionally a much tighter absolute bound (2e-5 vs the fp64 oracle) that a correct fp32 kernel
meets with margin, so an indexing bug cannot hide inside the loose bound.
"""
import os

import numpy as np
import pytest
import torch

from oracle import c_oracle as C
from oracle import fov_oracle as O

pytestmark = pytest.mark.gpu

RTOL, ATOL_FLOOR = 1e-3, 1e-5
TIGHT = 2e-5


def _ops():
    from longterm360fov_amd import ops
    return ops


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def devw(w):
    return {k: dev(v) for k, v in w.items()}


def f64(w):
    return {k: v.astype(np.float64) for k, v in w.items()}


def assert_parity(gpu, ref, what, tight=TIGHT):
    gpu = gpu.detach().cpu().numpy().astype(np.float64) if isinstance(gpu, torch.Tensor) else np.asarray(gpu, np.float64)
    ref = np.asarray(ref, np.float64)
    assert gpu.shape == ref.shape, (what, gpu.shape, ref.shape)
    assert np.isfinite(gpu).all(), what + ": non-finite values"
    err = np.abs(gpu - ref)
    bound = RTOL * np.abs(ref) + ATOL_FLOOR
    worst = float(err.max()) if err.size else 0.0
    print("%s: max abs err %.3e (max |ref| %.3f)" % (what, worst, float(np.abs(ref).max()) if ref.size else 0))
    assert (err <= bound).all(), "%s: max err %.3e exceeds 1e-3 relative" % (what, worst)
    assert worst <= tight, "%s: max err %.3e exceeds tight bound %.1e" % (what, worst, tight)


# ---------------------------------------------------------------------------------------
# LSTM layer (a1/a2)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl,H,F,B,T", [
    ("generic", 32, 6, 5, 6), ("generic", 32, 90, 9, 4), ("generic", 256, 90, 7, 3), ("generic", 40, 11, 3, 5),
    ("cluster", 64, 90, 37, 5), ("cluster", 128, 90, 37, 5), ("cluster", 256, 90, 37, 5),
    ("cluster", 256, 6, 16, 4), ("cluster", 128, 13, 1, 3), ("cluster", 256, 96, 33, 2),
])
@pytest.mark.parametrize("act", ["sigmoid", "hard_sigmoid"])
def test_lstm_layer(impl, H, F, B, T, act):
    ops = _ops()
    rng = np.random.default_rng(100 + H + F + B)
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32)
    ws = ops.Workspace()
    for init in (False, True):
        a0, a1 = (h0, c0) if init else (None, None)
        ref = O.lstm_layer(x.astype(np.float64), K.astype(np.float64), R.astype(np.float64), b.astype(np.float64),
                           None if a0 is None else a0.astype(np.float64), None if a1 is None else a1.astype(np.float64), act)
        hs, hT, cT = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), None if a0 is None else dev(a0),
                                  None if a1 is None else dev(a1), act=act, impl=impl, workspace=ws)
        ws.check()
        tag = "%s H%d F%d B%d T%d %s init=%s" % (impl, H, F, B, T, act, init)
        assert_parity(hs, ref[0], "hs " + tag)
        assert_parity(hT, ref[1], "hT " + tag)
        assert_parity(cT, ref[2], "cT " + tag)
        # return_sequences=False path (hs pointer NULL)
        _, hT2, cT2 = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), None if a0 is None else dev(a0),
                                   None if a1 is None else dev(a1), act=act, impl=impl, return_sequences=False, workspace=ws)
        assert torch.equal(hT2, hT) and torch.equal(cT2, cT)


# ---------------------------------------------------------------------------------------
# Fused encoder + autoregressive decoder (a2+a3), teacher-forced graph, Dense
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl,H,B,T_in,T_out", [
    ("generic", 32, 5, 6, 4), ("generic", 128, 32, 10, 10),
    ("cluster", 64, 20, 4, 3), ("cluster", 128, 32, 10, 10), ("cluster", 256, 48, 6, 5), ("cluster", 256, 17, 3, 7),
])
@pytest.mark.parametrize("act", ["sigmoid", "hard_sigmoid"])
def test_seq2seq_decode(impl, H, B, T_in, T_out, act):
    ops = _ops()
    w = O.init_seq2seq(7 + H, H=H, bias_noise=0.1)
    enc, dec0, _ = O.synthetic_batch(8 + B, B, T_in, T_out)
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), f64(w), T_out, act)
    ws = ops.Workspace()
    out = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, act=act, impl=impl, workspace=ws)
    ws.check()
    assert_parity(out, ref, "decode %s H%d B%d %d->%d %s" % (impl, H, B, T_in, T_out, act))


@pytest.mark.parametrize("impl,H,B", [("generic", 32, 5), ("cluster", 128, 32), ("cluster", 256, 40)])
def test_seq2seq_teacher_forced(impl, H, B):
    ops = _ops()
    T_in, T_out = 6, 5
    w = O.init_seq2seq(17 + H, H=H, bias_noise=0.1)
    enc, dec0, tgt = O.synthetic_batch(18, B, T_in, T_out)
    dec_in = np.concatenate([dec0, tgt[:, :-1]], axis=1)
    ref = O.seq2seq_teacher_forced(enc.astype(np.float64), dec_in.astype(np.float64), f64(w))
    ws = ops.Workspace()
    out = ops.seq2seq_teacher_forced(dev(enc), dev(dec_in), devw(w), impl=impl, workspace=ws)
    ws.check()
    assert_parity(out, ref, "teacher-forced %s H%d" % (impl, H))


def test_dense():
    ops = _ops()
    rng = np.random.default_rng(3)
    # (N >= 64, Out <= 8) runs the narrow-head kernel (16-byte row loads when In % 4 == 0, scalar ones otherwise: the others'
    # projection has In = 198), the rest the generic one
    for N, In, Out in ((7, 256, 6), (130, 204, 6), (5, 33, 20), (1000, 256, 6), (4099, 40, 3), (65, 2048, 8), (333, 30, 6),
                       (70, 512, 1), (5120, 198, 6), (100, 65, 8), (64, 1, 2),
                       # few rows, In >= 128, Out > 8: the four-rows-per-workgroup kernel (lstm.py's 400 -> 32 heads, padded to 512)
                       (32, 512, 32), (1, 128, 9), (37, 400, 33), (256, 2047, 70), (5, 131, 64)):
        x = rng.standard_normal((N, In)).astype(np.float32)
        W = (rng.standard_normal((In, Out)) / np.sqrt(In)).astype(np.float32)
        b = rng.standard_normal(Out).astype(np.float32)
        ref = O.dense(x.astype(np.float64), W.astype(np.float64), b.astype(np.float64))
        assert_parity(ops.dense(dev(x), dev(W), dev(b)), ref, "dense %dx%dx%d" % (N, In, Out))
        ref = x.astype(np.float64) @ W.astype(np.float64) + b
        assert_parity(ops.dense(dev(x), dev(W), dev(b), activation=None), ref, "dense linear")
        # the others-mixing form: + a strided row view (given_others...py:257-265)
        addbuf = rng.standard_normal((N, Out + 5)).astype(np.float32)
        ref = np.tanh(x.astype(np.float64) @ W.astype(np.float64) + b + addbuf[:, 2:2 + Out])
        assert_parity(ops.dense_add(dev(x), dev(W), dev(b), dev(addbuf)[:, 2:2 + Out]), ref, "dense_add")


# ---------------------------------------------------------------------------------------
# committed golden vectors
# ---------------------------------------------------------------------------------------
def test_golden_lstm_vectors(golden_dir):
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "lstm_small.npz"))
    w = {k[2:]: g[k] for k in g.files if k.startswith("w_")}
    T_out = int(g["T_out"])
    for act in (0, 1):
        out = ops.seq2seq_decode(dev(g["enc"]), dev(g["dec0"]), devw(w), T_out, act=act)
        assert_parity(out, g["decode_act%d" % act], "golden decode act%d" % act)
        dec_in = np.concatenate([g["dec0"], g["tgt"][:, :-1]], axis=1)
        out = ops.seq2seq_teacher_forced(dev(g["enc"]), dev(dec_in), devw(w), act=act)
        assert_parity(out, g["tf_act%d" % act], "golden teacher-forced act%d" % act)


def test_meanvar_against_reference_fixture(golden_dir):
    """mu/sigma^2 op against values produced by the reference's own get_gt_target_xyz[_oth]."""
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "data_helpers.npz"))
    out = ops.meanvar_xyz(dev(g["fut"]))
    assert_parity(out, g["gt_fut"], "meanvar (N,T,90)", tight=2e-6)
    out = ops.meanvar_xyz(dev(g["fut"].reshape(12, 10, 30, 3)))
    assert_parity(out, g["gt_fut_4d"], "meanvar (N,T,30,3)", tight=2e-6)
    oth = g["pu_oth_fut"].transpose(1, 2, 0, 3).reshape(12, 10, 2, 30, 3)
    out = ops.meanvar_xyz(dev(oth))
    assert_parity(out, g["gt_oth_fut"], "meanvar others", tight=2e-6)
    assert ops.meanvar_xyz(dev(np.zeros((0, 10, 90)))).shape == (0, 10, 6)


# ---------------------------------------------------------------------------------------
# edge cases
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl,H", [("generic", 32), ("cluster", 256)])
def test_empty_and_degenerate(impl, H):
    ops = _ops()
    w = O.init_seq2seq(5, H=H)
    dw = devw(w)
    out = ops.seq2seq_decode(dev(np.zeros((0, 4, 90))), dev(np.zeros((0, 1, 6))), dw, 3, impl=impl)
    assert out.shape == (0, 3, 6)
    # single sequence, single step in and out
    enc, dec0, _ = O.synthetic_batch(2, 1, 1, 1)
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), f64(w), 1)
    assert_parity(ops.seq2seq_decode(dev(enc), dev(dec0), dw, 1, impl=impl), ref, "B=1 T=1 " + impl)
    # T_in = 0: decoder starts from the zero state
    enc0 = np.zeros((3, 0, 90), np.float32)
    dec0 = np.random.default_rng(1).uniform(-1, 1, (3, 1, 6)).astype(np.float32)
    ref = O.seq2seq_decode(enc0.astype(np.float64), dec0.astype(np.float64), f64(w), 2)
    assert_parity(ops.seq2seq_decode(dev(enc0), dev(dec0), dw, 2, impl=impl), ref, "T_in=0 " + impl)


def test_error_reporting():
    ops = _ops()
    from longterm360fov_amd import _lib
    w = O.init_seq2seq(5, H=48)
    enc, dec0, _ = O.synthetic_batch(2, 4, 2, 2)
    with pytest.raises(_lib.FovError) as e:
        ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), 2, impl="cluster")   # H=48 unsupported
    assert e.value.code == _lib.ERR_UNSUPPORTED
    # AUTO falls back to the generic kernel for the same shape
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), f64(w), 2)
    assert_parity(ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), 2, impl="auto"), ref, "auto H48")
    with pytest.raises(TypeError):
        ops.dense(torch.zeros(2, 3), dev(np.zeros((3, 2))), dev(np.zeros(2)))   # CPU tensor refused


# ---------------------------------------------------------------------------------------
# BASELINE.json full sizes: config 1 and config 2, plus size-independent properties
# ---------------------------------------------------------------------------------------
def test_config1_reference_native_shape():
    """configs[0]: H=128, B=32, T 10->10 (the reference's own operating point)."""
    ops = _ops()
    w = O.init_seq2seq(1234, H=128, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(1234, 32, 10, 10)
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), f64(w), 10)
    for impl in ("generic", "cluster"):
        assert_parity(ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), 10, impl=impl), ref, "config1 " + impl)


def test_two_width_512_layers_one_launch():
    """fov_lstm_stack2_fwd: two stacked width-512 layers as one launch (layer 2 behind layer 1, input through the granule
    ring) against two separate layer launches (bit-identical: same arithmetic per layer) and the fp64 oracle; fed states,
    ragged batches, the training tape, repeated launches on one workspace."""
    ops = _ops()
    rng = np.random.default_rng(21)
    H, F = 512, 90
    l1 = O.init_lstm(rng, F, H, np.float32)
    l2 = O.init_lstm(rng, H, H, np.float32)
    d1, d2 = tuple(dev(a) for a in l1), tuple(dev(a) for a in l2)
    ws = ops.Workspace()
    for B, T, with_state, reserve in ((32, 10, True, True), (7, 2, False, False), (64, 5, False, True), (37, 13, True, False)):
        assert ops.lstm_stack2_supported(B, T, F, H)
        x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
        st = [None, None]
        if with_state:
            st = [(dev((0.3 * rng.standard_normal((B, H))).astype(np.float32)), dev((0.3 * rng.standard_normal((B, H))).astype(np.float32)))
                  for _ in range(2)]
        for _ in range(2):
            o1, o2 = ops.lstm_stack2(dev(x), d1, d2, st[0], st[1], workspace=ws, reserve=reserve)
        ws.check()
        h01, c01 = st[0] if st[0] else (None, None)
        h02, c02 = st[1] if st[1] else (None, None)
        r1 = ops.lstm_seq_train(dev(x), *d1, h01, c01)
        r2 = ops.lstm_seq_train(r1[0], *d2, h02, c02)
        # at most two tiles: THREE roles (layer 2's input projection h1_t . K2 on workgroups of its own, lstm_wide16.hip) -
        # layer 2 then adds that product to bias + h . R2 instead of accumulating all three in one chain: same values up to
        # the order of the fp32 sums
        three_roles = B <= 32
        for li, (got, ref) in enumerate(((o1, r1), (o2, r2))):
            for k in range(4 if reserve else 3):
                if li == 1 and three_roles:
                    assert torch.allclose(got[k], ref[k], rtol=2e-5, atol=2e-6), (B, T, k, float((got[k] - ref[k]).abs().max()))
                else:
                    assert torch.equal(got[k], ref[k]), (B, T, k)
        ref = x.astype(np.float64)
        for l, (K, R, b) in enumerate((l1, l2)):
            s0 = st[l]
            ref, _, _ = O.lstm_layer(ref, K.astype(np.float64), R.astype(np.float64), b.astype(np.float64),
                                     None if s0 is None else s0[0].cpu().numpy().astype(np.float64),
                                     None if s0 is None else s0[1].cpu().numpy().astype(np.float64), act="sigmoid")
        assert_parity(o2[0], ref, "two width-512 layers, one launch B=%d T=%d" % (B, T))
    assert not ops.lstm_stack2_supported(80, 10, F, H) and not ops.lstm_stack2_supported(32, 1, F, H) and not ops.lstm_stack2_supported(32, 10, F, 256)


@pytest.mark.parametrize("B,T,with_state,reserve", [(1, 2, False, False), (16, 3, True, True), (17, 5, False, True), (32, 2, True, False),
                                                     (31, 7, True, True), (16, 200, False, False)])
def test_two_width_512_layers_three_roles_two_roles_two_launches(B, T, with_state, reserve):
    """At most two tiles: the one-launch form runs as THREE roles (layer 1, the products h1 . K2, layer 2; lstm_wide16.hip).
    FOV_NO_WIDE16_TRIO=1 keeps two roles on the same XCD-per-group grid - bit-identical to two separate layer launches;
    the three-role form differs from both in the order of layer 2's fp32 sums only.  Fed states, the tape, ragged tiles,
    repeated launches on one workspace, and the fp64 oracle.  T = 200: the three-role form's mailboxes no longer fit the granule
    area - the launch falls back to two roles by itself."""
    import os
    from longterm360fov_amd import _lib
    ops = _ops()
    rng = np.random.default_rng(B * 100 + T)
    H, F = 512, 90
    l1 = O.init_lstm(rng, F, H, np.float32)
    l2 = O.init_lstm(rng, H, H, np.float32)
    d1, d2 = tuple(dev(a) for a in l1), tuple(dev(a) for a in l2)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    st = [None, None]
    if with_state:
        st = [(dev((0.3 * rng.standard_normal((B, H))).astype(np.float32)), dev((0.3 * rng.standard_normal((B, H))).astype(np.float32)))
              for _ in range(2)]
    ws = ops.Workspace()
    for _ in range(3):
        t1, t2 = ops.lstm_stack2(dev(x), d1, d2, st[0], st[1], workspace=ws, reserve=reserve)
    os.environ["FOV_NO_WIDE16_TRIO"] = "1"
    _lib.lib().fov_reload_env()
    try:
        for _ in range(2):
            p1, p2 = ops.lstm_stack2(dev(x), d1, d2, st[0], st[1], workspace=ws, reserve=reserve)
    finally:
        os.environ.pop("FOV_NO_WIDE16_TRIO", None)
        _lib.lib().fov_reload_env()
    ws.check()
    h01, c01 = st[0] if st[0] else (None, None)
    h02, c02 = st[1] if st[1] else (None, None)
    r1 = ops.lstm_seq_train(dev(x), *d1, h01, c01)
    r2 = ops.lstm_seq_train(r1[0], *d2, h02, c02)
    for k in range(4 if reserve else 3):
        assert torch.equal(p1[k], r1[k]) and torch.equal(p2[k], r2[k]) and torch.equal(t1[k], r1[k]), (B, T, k)
        assert torch.allclose(t2[k], r2[k], rtol=2e-5, atol=2e-6), (B, T, k, float((t2[k] - r2[k]).abs().max()))
    assert ops.lstm_stack2_supported(B, T, F, H)
    ref = x.astype(np.float64)
    for l, (K, R, b) in enumerate((l1, l2)):
        s0 = st[l]
        ref, _, _ = O.lstm_layer(ref, K.astype(np.float64), R.astype(np.float64), b.astype(np.float64),
                                 None if s0 is None else s0[0].cpu().numpy().astype(np.float64),
                                 None if s0 is None else s0[1].cpu().numpy().astype(np.float64), act="sigmoid")
    assert_parity(t2[0], ref, "three roles B=%d T=%d" % (B, T))


def test_fused_decode_padded_grid_group_counts():
    """H = 128 fused encoder + decoder with a group count that is no multiple of eight: the grid is padded so that a group's
    members share an XCD (spare workgroups leave at once); 2, 9 and 13 tiles (the last ragged), repeated launches on one
    workspace (the spare blocks take part in the arrival count of the launch protocol)."""
    ops = _ops()
    w = O.init_seq2seq(77, H=128, bias_noise=0.05)
    ws = ops.Workspace()
    for B in (32, 16 * 9, 16 * 12 + 5):
        enc, dec0, _ = O.synthetic_batch(B, B, 7, 6)
        ref = C.seq2seq_decode(enc, dec0, w, 6)
        for _ in range(3):
            out = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), 6, impl="cluster", workspace=ws)
        ws.check()
        assert_parity(out, ref.astype(np.float64), "padded fused grid B=%d" % B)
        os.environ["FOV_NO_XCD_PAD"] = "1"     # the unpadded grid (placement-independent exchange): bit-identical
        try:
            plain = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), 6, impl="cluster", workspace=ws)
        finally:
            os.environ.pop("FOV_NO_XCD_PAD", None)
        ws.check()
        assert torch.equal(plain, out), B


def test_config2_full_size_and_properties():
    """configs[1]: H=256, B=1024, T 30->30 on one GPU (the bench workload).  All 1024 sequences
    against the C oracle, plus determinism, batch-permutation equivariance, independence from
    tile-mates (a sequence's result must not depend on which other sequences share its tile)."""
    ops = _ops()
    B, T_in, T_out, H = 1024, 30, 30, 256
    w = O.init_seq2seq(1234, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(1234, B, T_in, T_out)
    ref = C.seq2seq_decode(enc, dec0, w, T_out)                      # fp32 C oracle, all sequences
    ref64 = O.seq2seq_decode(enc[:64].astype(np.float64), dec0[:64].astype(np.float64), f64(w), T_out)
    dw = devw(w)
    ws = ops.Workspace()
    out = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, impl="cluster", workspace=ws)
    ws.check()
    assert_parity(out, ref, "config2 cluster vs C oracle (1024 seq)", tight=5e-5)
    assert_parity(out[:64], ref64, "config2 cluster vs fp64 oracle (64 seq)", tight=5e-5)
    out2 = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, impl="cluster", workspace=ws)
    assert torch.equal(out, out2), "not deterministic"
    perm = np.random.default_rng(0).permutation(B)
    outp = ops.seq2seq_decode(dev(enc[perm]), dev(dec0[perm]), dw, T_out, impl="cluster", workspace=ws)
    assert torch.equal(outp, out[torch.from_numpy(perm).cuda()]), "not batch-permutation equivariant"
    outr = ops.seq2seq_decode(dev(enc[:1000]), dev(dec0[:1000]), dw, T_out, impl="cluster", workspace=ws)
    assert torch.equal(outr, out[:1000]), "result depends on batch padding"
    gen = ops.seq2seq_decode(dev(enc[:128]), dev(dec0[:128]), dw, T_out, impl="generic")
    assert_parity(gen, ref[:128], "config2 generic vs C oracle (128 seq)", tight=5e-5)
    assert float(out.abs().max()) <= 1.0   # tanh head


def test_persistent_tile_loop_more_tiles_than_groups():
    """B > 16 * (CUs / G): every group walks several tiles; epochs keep counting across tiles."""
    ops = _ops()
    B, T_in, T_out, H = 16 * 64 * 2 + 16 * 7 + 5, 3, 3, 256
    w = O.init_seq2seq(99, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(98, B, T_in, T_out)
    ref = C.seq2seq_decode(enc, dec0, w, T_out)
    ws = ops.Workspace()
    out = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, impl="cluster", workspace=ws)
    ws.check()
    assert_parity(out, ref, "multi-tile cluster B=%d" % B, tight=5e-5)
    w1 = O.init_seq2seq(97, H=128, bias_noise=0.05)
    ref = C.seq2seq_decode(enc, dec0, w1, T_out)
    out = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w1), T_out, impl="cluster", workspace=ws)
    ws.check()
    assert_parity(out, ref, "multi-tile cluster H128 B=%d" % B, tight=5e-5)


def test_exchange_paths_agree_bitwise():
    """The same-XCD fast exchange (sc0 stores kept in the XCD's L2, taken only after the
    HW_REG_XCC_ID handshake proves co-location) and the placement-independent write-through
    exchange must give bit-identical results; FOV_FORCE_SAFE_EXCHANGE=1 selects the latter."""
    ops = _ops()
    B, T_in, T_out, H = 1024, 8, 8, 256
    w = O.init_seq2seq(4321, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(4321, B, T_in, T_out)
    dw = devw(w)
    ws = ops.Workspace()
    outs = {}
    try:
        for mode in ("0", "1"):
            os.environ["FOV_FORCE_SAFE_EXCHANGE"] = mode
            outs[mode] = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, impl="cluster", workspace=ws).clone()
            ws.check()
            got = ws.exchange_mode()
            print("FOV_FORCE_SAFE_EXCHANGE=%s -> exchange mode %d (1 = same-XCD fast, 2 = write-through)" % (mode, got))
            if mode == "1":
                assert got == 2
    finally:
        os.environ.pop("FOV_FORCE_SAFE_EXCHANGE", None)
    assert torch.equal(outs["0"], outs["1"])
    ref = C.seq2seq_decode(enc, dec0, w, T_out)
    assert_parity(outs["0"], ref, "exchange paths vs C oracle", tight=5e-5)


# ---------------------------------------------------------------------------------------
# a4: target + others mixing, 2+2 layers, no teacher forcing (given_others_gt_mean_var_seq2seq.py)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,B,U,T_in,T_out,act", [(32, 5, 4, 6, 4, "sigmoid"), (64, 20, 34, 5, 4, "hard_sigmoid"),
                                                  (256, 48, 34, 10, 10, "sigmoid"),
                                                  (256, 48, 34, 30, 30, "sigmoid"), (256, 512, 34, 30, 30, "hard_sigmoid")])   # the metric's horizon (config.py:20-23)
def test_others_mixing_forward(H, B, U, T_in, T_out, act):
    from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
    w = O.init_others_mixing(60 + H, H=H, num_user=U, bias_noise=0.1)
    enc, dec0, _, oth = O.synthetic_batch(61 + B, B, T_in, T_out, num_others=U - 1)
    ref = O.others_mixing_forward(enc.astype(np.float64), oth.astype(np.float64), dec0.astype(np.float64), f64(w), act)
    m = OthersMixingSeq2Seq(latent_dim=H, num_user=U, recurrent_activation=act)
    m.set_weights([w[k] for k in _MIX_ORDER])
    out = m.predict([enc, oth, dec0])
    assert_parity(out, ref, "others-mixing H%d B%d U%d %s" % (H, B, U, act))
    assert m.count_params() == sum(v.size for v in w.values())


def test_golden_others_mixing_vector(golden_dir):
    from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
    g = np.load(os.path.join(golden_dir, "lstm_small.npz"))
    wm = {k[3:]: g[k] for k in g.files if k.startswith("wm_")}
    m = OthersMixingSeq2Seq(latent_dim=wm["enc1_R"].shape[0], num_user=g["oth"].shape[2] + 1)
    m.set_weights([wm[k] for k in _MIX_ORDER])
    assert_parity(m.predict([g["enc"], g["oth"], g["dec0"]]), g["mix_act0"], "golden others-mixing")


def test_matmul_and_zx_layer():
    ops = _ops()
    rng = np.random.default_rng(9)
    for M, K, N in ((7, 33, 5), (300, 256, 1024), (1, 256, 1024), (130, 6, 70), (20, 64, 1024), (90, 100, 256),
                    (96, 257, 260), (513, 40, 132)):   # every tile variant, vector and scalar staging, ragged edges
        a = rng.standard_normal((M, K)).astype(np.float32); b = rng.standard_normal((K, N)).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        got = ops.matmul(dev(a), dev(b)).cpu().numpy()
        assert np.abs(got - ref).max() <= 1e-5 * np.abs(ref).max() + 1e-6, (M, K, N)
    # a stacked layer through zx equals the same layer given x directly
    H, F, B, T = 128, 40, 21, 5
    Kk, R, b = O.init_lstm(rng, F, H, np.float32)
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    ref = O.lstm_layer(x.astype(np.float64), Kk.astype(np.float64), R.astype(np.float64), b.astype(np.float64))
    for impl in ("cluster", "generic"):
        zx = ops.matmul(dev(x.reshape(B * T, F)), dev(Kk)).reshape(B, T, 4 * H)
        hs, hT, cT = ops.lstm_seq_zx(zx, dev(R), dev(b), impl=impl)
        assert_parity(hs, ref[0], "zx layer hs " + impl)
        assert_parity(cT, ref[2], "zx layer cT " + impl)


def test_tf_contrib_stacked_lstm_mapping():
    """a10: MultiRNNCell[2 x LSTMCell(400)] + dynamic_rnn with a fed state (mycode/lstm.py:218-240)
    through the Keras-layout kernels via convert_tf_lstmcell (gate order i,j,f,o -> i,f,c,o,
    forget_bias folded into the bias).  Oracle = restated tf.contrib LSTMCell."""
    from longterm360fov_amd.models import StackedTFLSTM
    rng = np.random.default_rng(17)
    for H, F, B, T in ((400, 90, 32, 10), (64, 90, 7, 3)):
        cells = []
        for l in range(2):
            Fin = F if l == 0 else H
            W = (rng.standard_normal((Fin + H, 4 * H)) / np.sqrt(Fin + H)).astype(np.float32)
            b = (0.1 * rng.standard_normal(4 * H)).astype(np.float32)
            cells.append((W, b))
        x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
        st0 = (0.3 * rng.standard_normal((2, 2, B, H))).astype(np.float32)
        ref_out, ref_st = O.tf_dynamic_rnn(x.astype(np.float64), [(W.astype(np.float64), b.astype(np.float64)) for W, b in cells],
                                           st0.astype(np.float64))
        m = StackedTFLSTM(cells)
        out, st = m.predict(x, st0)
        assert_parity(out, ref_out, "tf stacked LSTM H%d states_series" % H)
        assert_parity(st, ref_st, "tf stacked LSTM H%d current_state" % H)
        out0, _ = m.predict(x)                       # zero initial state
        ref0, _ = O.tf_dynamic_rnn(x.astype(np.float64), [(W.astype(np.float64), b.astype(np.float64)) for W, b in cells])
        assert_parity(out0, ref0, "tf stacked LSTM H%d zero state" % H)
        if H == 400:   # 400 units run zero-padded at width 512 on the persistent kernel; the unpadded step-wise form agrees
            assert m.run_width == 512
            u = StackedTFLSTM(cells, pad=False)
            assert u.run_width == 400
            out_u, st_u = u.predict(x, st0)
            assert np.abs(out_u - out).max() <= 2e-5 and np.abs(st_u - st).max() <= 2e-5


def test_fov_hit_rate_against_oracle(golden_dir):
    """the evaluation step fed by the model output, incl. the theta-seam cases."""
    ops = _ops()
    rng = np.random.default_rng(12)
    N, T = 300, 10
    gt = rng.standard_normal((N, T, 3)); gt /= np.linalg.norm(gt, axis=-1, keepdims=True)
    pred6 = np.concatenate([gt + 0.4 * rng.standard_normal((N, T, 3)), rng.uniform(0, 0.1, (N, T, 3))], axis=-1)
    # force seam cases: gt just below +pi in theta, prediction just above -pi (and the reverse)
    gt[:20, :, 0], gt[:20, :, 1] = 1.0, -0.05
    pred6[:20, :, 0], pred6[:20, :, 1] = 1.0, 0.05
    gt[20:40, :, 0], gt[20:40, :, 1] = 1.0, 0.05
    pred6[20:40, :, 0], pred6[20:40, :, 1] = 1.0, -0.05
    ref = O.fov_hit_rate(pred6[..., :3].astype(np.float32).astype(np.float64), gt.astype(np.float32).astype(np.float64))
    got = ops.fov_hit_rate(dev(pred6), dev(gt)).cpu().numpy()
    assert got.shape == (N, T)
    assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()
    assert (ref[:40] > 0.5).all() and (got[:40] > 0.5).all()      # seam cases overlap instead of scoring 0
    g = np.load(os.path.join(golden_dir, "data_helpers.npz"))    # reference-pinned angles feed the same formula
    one = ops.fov_hit_rate(dev(g["eval_xyz"]), dev(g["eval_xyz"])).cpu().numpy()
    np.testing.assert_allclose(one, 1.0, atol=1e-6)


def test_device_windowing_matches_reference_fixture(golden_dir):
    """reshape2second_stacks as a device gather, against windows produced by the
    reference's own function (strides 1 and 5, both user layouts) - exact, it is a copy."""
    ops = _ops()
    g = np.load(os.path.join(golden_dir, "data_helpers.npz"))
    x = dev(g["s1_in"])                                   # (2, 23, 90) seconds
    enc, fut, fut_in = ops.window_stacks(x, T=10, stride=1, collapse_user=True)
    for got, key in ((enc, "s1_enc"), (fut, "s1_fut"), (fut_in, "s1_fut_in")):
        np.testing.assert_array_equal(got.cpu().numpy(), g[key].astype(np.float32))
    enc, fut, fut_in = ops.window_stacks(x, T=10, stride=5, collapse_user=False)
    for got, key in ((enc, "s5_enc"), (fut, "s5_fut"), (fut_in, "s5_fut_in")):
        np.testing.assert_array_equal(got.cpu().numpy(), g[key].astype(np.float32))
    assert (enc[:, :, -1] == fut_in[:, :, 0]).all()       # the reference's sanity check


@pytest.mark.parametrize("B,T_in,T_out,act", [(16, 3, 4, "sigmoid"), (37, 2, 3, "hard_sigmoid"), (512, 4, 10, "sigmoid"),
                                              (600, 2, 2, "sigmoid"), (48, 30, 30, "sigmoid"), (512, 30, 30, "sigmoid")])
def test_fused_mixing_decoder_matches_oracle(B, T_in, T_out, act):
    """a4: the unrolled no-teacher-forcing decoder with others mixing in ONE launch (fov_mix_decoder_fwd) against the
    oracle's others_mixing_forward; encoder states come from the library's layer kernels.  B = 600 makes a group
    visit more than one tile; B = 37 leaves a ragged last tile."""
    ops = _ops()
    H, U, NO = 256, 34, 6
    w = O.init_others_mixing(300 + B, H=H, num_user=U, bias_noise=0.1)
    enc, dec0, tgt, oth = O.synthetic_batch(301 + B, B, T_in, T_out, num_others=U - 1)
    ref = O.others_mixing_forward(enc.astype(np.float64), oth.astype(np.float64), dec0.astype(np.float64),
                                  {k: v.astype(np.float64) for k, v in w.items()}, act=act)
    dw = devw(w)
    n_oth = (U - 1) * NO
    Wm_o, Wm_p = dw["mix_W"][:n_oth].contiguous(), dw["mix_W"][n_oth:].contiguous()
    hs1, h1, c1 = ops.lstm_seq(dev(enc), dw["enc1_K"], dw["enc1_R"], dw["enc1_b"], act=act)
    zx = ops.matmul(hs1.reshape(B * T_in, H), dw["enc2_K"]).reshape(B, T_in, 4 * H)
    _, h2, c2 = ops.lstm_seq_zx(zx, dw["enc2_R"], dw["enc2_b"], act=act, return_sequences=False)
    oth_proj = ops.dense(dev(oth).reshape(B * T_out, n_oth), Wm_o, dw["mix_b"], activation=None).reshape(B, T_out, NO)
    ws = ops.Workspace()
    out = ops.mix_decoder(dev(dec0), h1, c1, h2, c2, oth_proj, dw, Wm_p, T_out, act=act, workspace=ws)
    ws.check()
    assert_parity(out.transpose(0, 1), ref, "fused mixing decoder B=%d" % B)


@pytest.mark.parametrize("B,T,F", [(37, 5, 256), (16, 1, 256), (600, 3, 128), (20, 4, 100),
                                   (512, 10, 90), (37, 3, 6), (100, 4, 96), (17, 5, 33), (5, 1, 90)])
@pytest.mark.parametrize("act", ["sigmoid", "hard_sigmoid"])
def test_wide_input_layer(B, T, F, act):
    """A stacked layer over a wide sequence (given_others...py:111-112: encoder layer 2 over the 256-wide output of
    layer 1): K and R both register-resident (lstm_wide.hip), no precomputed input projection.  Also the
    training form (reserve) and a given initial state; B = 600 makes a group visit several tiles.  F <= 96 with at
    most 512 sequences (impl auto) is the narrow variant of the same kernel: groups of eight workgroups fill the chip
    where the cluster kernel's groups of four would leave half of it idle (encoder layer 1 of configs[2])."""
    ops = _ops()
    H = 256
    rng = np.random.default_rng(B + F)
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32)
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32)
    d = lambda a: a.astype(np.float64)
    ref = O.lstm_layer(d(x), d(K), d(R), d(b), d(h0), d(c0), act=act)
    ws = ops.Workspace()
    hs, hT, cT = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), dev(h0), dev(c0), act=act, workspace=ws)
    ws.check()
    assert_parity(hs, ref[0], "wide layer hs F=%d" % F)
    assert_parity(hT, ref[1], "wide layer hT")
    assert_parity(cT, ref[2], "wide layer cT")
    hs2, hT2, cT2, res = ops.lstm_seq_train(dev(x), dev(K), dev(R), dev(b), dev(h0), dev(c0), act=act, workspace=ws)
    ws.check()
    hs_g, _, _, res_g = ops.lstm_seq_train(dev(x), dev(K), dev(R), dev(b), dev(h0), dev(c0), act=act, impl="generic")
    assert torch.equal(hs2, hs)                                   # deterministic, same kernel
    assert (res - res_g).abs().max().item() < 2e-6                 # reserve agrees with the generic kernel's
    assert (hs2 - hs_g).abs().max().item() < 2e-6


@pytest.mark.parametrize("B,T,F,state", [(32, 10, 90, False), (32, 10, 512, True), (37, 5, 200, True), (5, 1, 7, False),
                                         (16, 1, 512, True), (600, 3, 90, True), (300, 4, 400, False), (128, 6, 512, False),
                                         (100, 7, 33, True), (129, 3, 512, True)])
@pytest.mark.parametrize("act", ["sigmoid", "hard_sigmoid"])
def test_width_512_persistent_layer(B, T, F, state, act):
    """H = 512 (mycode/lstm.py's LSTMCell(400) x 2 zero-padded, :59,218-240): the recurrent kernel stays in registers across
    SIXTEEN workgroups per 16-sequence tile (lstm_wide.hip).  F <= 96: the input kernel sits in registers too; wider inputs
    are projected by one GEMM inside the same C-ABI call and added per step.  Against the fp64 oracle and the generic
    kernel: hidden sequence, final states, the training tape, a fed initial state, ragged batches, more tiles than groups
    (B = 600: 38 tiles on 16 groups), and the entry point that takes the projection precomputed."""
    from longterm360fov_amd import _lib
    ops = _ops()
    H = 512
    rng = np.random.default_rng(B + F)
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    d = lambda a: None if a is None else a.astype(np.float64)
    dv = lambda a: None if a is None else dev(a)
    hs_ref, hT_ref, cT_ref, res_ref = O.lstm_layer_train(d(x), d(K), d(R), d(b), d(h0), d(c0), act=act)
    ws = ops.Workspace()
    before = _lib.lib().fov_debug_generic_launches()
    hs, hT, cT = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), dv(h0), dv(c0), act=act, workspace=ws)
    ws.check()
    assert _lib.lib().fov_debug_generic_launches() == before       # not the VALU kernel
    assert_parity(hs, hs_ref, "width-512 layer hs F=%d" % F)
    assert_parity(hT, hT_ref, "width-512 layer hT")
    assert_parity(cT, cT_ref, "width-512 layer cT")
    hs2, hT2, cT2, res = ops.lstm_seq_train(dev(x), dev(K), dev(R), dev(b), dv(h0), dv(c0), act=act, workspace=ws)
    ws.check()
    assert torch.equal(hs2, hs) and torch.equal(cT2, cT)           # deterministic, same kernel
    assert np.abs(res.cpu().numpy() - res_ref).max() <= 2e-5
    g_hs, _, g_cT = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), dv(h0), dv(c0), act=act, impl="generic", workspace=ops.Workspace())
    assert (g_hs - hs).abs().max().item() <= 2e-5 and (g_cT - cT).abs().max().item() <= 2e-5
    _, hT3, cT3 = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), dv(h0), dv(c0), act=act, return_sequences=False, workspace=ws)
    assert torch.equal(hT3, hT) and torch.equal(cT3, cT)
    # small batches (one tile per group of THIRTY-TWO workgroups, lstm_wide16.hip: K and R both in registers for F <= 96 and
    # F = 512) against the sixteen-workgroup form of the same layer
    os.environ["FOV_NO_WIDE16"] = "1"
    try:
        hs16, _, cT16 = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), dv(h0), dv(c0), act=act, workspace=ws)
    finally:
        del os.environ["FOV_NO_WIDE16"]
    ws.check()
    assert (hs16 - hs).abs().max().item() <= 2e-6 and (cT16 - cT).abs().max().item() <= 4e-6
    zx = ops.matmul(dev(x).reshape(B * T, F), dev(K)).reshape(B, T, 4 * H)
    hs4, _, cT4 = ops.lstm_seq_zx(zx, dev(R), dev(b), dv(h0), dv(c0), act=act, workspace=ws)
    ws.check()
    assert_parity(hs4, hs_ref, "width-512 layer, projection given")
    assert (cT4 - cT).abs().max().item() <= 2e-5


def test_config3_full_size_and_properties():
    """configs[2]: the 2+2-layer others-mixing model, H=256, 34 users, T 10->10; global batch 4096 = 8 ranks x 512.  One
    rank's shard (512) through the model object against the fp64 oracle; the whole global batch in ONE call (groups walk
    several tiles) must agree with the eight shard calls (replicas only: a rank's result depends on nothing but its own
    shard), and a shard call is deterministic and batch-permutation equivariant bit for bit."""
    from longterm360fov_amd.models import OthersMixingSeq2Seq
    B, T_in, T_out, H, U = 4096, 10, 10, 256, 34
    w = O.init_others_mixing(1234, H=H, num_user=U, bias_noise=0.05)
    enc, dec0, tgt, oth = O.synthetic_batch(1234, B, T_in, T_out, num_others=U - 1)
    m = OthersMixingSeq2Seq(latent_dim=H, num_user=U, recurrent_activation="sigmoid")
    m.set_weights([w[k] for k in ("enc1_K", "enc1_R", "enc1_b", "enc2_K", "enc2_R", "enc2_b", "dec1_K", "dec1_R", "dec1_b",
                                  "dec2_K", "dec2_R", "dec2_b", "dense_W", "dense_b", "mix_W", "mix_b")])
    full = m.predict([enc, oth, dec0])
    assert full.shape == (B, T_out, 6) and np.isfinite(full).all() and np.abs(full).max() <= 1.0
    ref = O.others_mixing_forward(enc[:512].astype(np.float64), oth[:512].astype(np.float64), dec0[:512].astype(np.float64), f64(w))
    assert_parity(torch.from_numpy(full[:512]), ref, "config3 shard 0 (512 seq) vs fp64 oracle", tight=5e-5)
    shards = []
    for r in range(8):   # a 512-sequence shard takes the eight-workgroup layer kernel for encoder layer 1, the 4096 call the
        sl = slice(512 * r, 512 * (r + 1))      # four-workgroup one: same arithmetic, different summation order
        shards.append(m.predict([enc[sl], oth[sl], dec0[sl]]))
        assert np.abs(shards[r] - full[sl]).max() <= 2e-6, "rank %d shard differs from the global-batch call" % r
    twice = m.predict([enc[:512], oth[:512], dec0[:512]])
    assert np.array_equal(twice, shards[0]), "not deterministic"
    perm = np.random.default_rng(0).permutation(512)
    outp = m.predict([enc[:512][perm], oth[:512][perm], dec0[:512][perm]])
    assert np.array_equal(outp, shards[0][perm]), "not batch-permutation equivariant"


@pytest.mark.parametrize("H", [32, 40, 100, 200])
def test_unsupported_widths_run_padded_on_the_mfma_kernels(H):
    """Widths the persistent kernels are not built for run at the next MFMA width with zero-padded weights - exact, not
    an approximation: the model object's answers (fused decode, encoder / decoder sampling models) match the fp64
    oracle at the model's own width to the usual bound, and the generic kernel at the true width agrees."""
    from longterm360fov_amd.models import Seq2SeqLSTM, _W_ORDER
    m = Seq2SeqLSTM(latent_dim=H, seed=H)
    assert m._run_width() in (64, 128, 256) and m._run_width() >= H
    w = dict(zip(_W_ORDER, m.get_weights()))
    enc, dec0, _ = O.synthetic_batch(H, 37, 6, 5)
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), {k: v.astype(np.float64) for k, v in w.items()}, 5)
    got = m.decode_sequence(enc, dec0, predict_step=5)
    assert np.abs(got - ref).max() <= 2e-5
    h, c = m.encoder_model.predict(enc)
    assert h.shape == (37, H) and c.shape == (37, H)
    _, rh, rc = O.lstm_layer(enc.astype(np.float64), *(w[k].astype(np.float64) for k in ("enc_K", "enc_R", "enc_b")))
    assert np.abs(h - rh).max() <= 2e-5 and np.abs(c - rc).max() <= 2e-5
    y, h2, c2 = m.decoder_model.predict([dec0, h, c])
    assert y.shape == (37, 1, 6) and h2.shape == (37, H)
    np.testing.assert_allclose(y[:, 0], got[:, 0], atol=2e-6)
    g = Seq2SeqLSTM(latent_dim=H, seed=H, impl="generic")
    assert g._run_width() == H
    np.testing.assert_allclose(g.decode_sequence(enc, dec0, predict_step=5), got, atol=2e-5)


def test_others_mixing_reference_width_takes_the_fused_path():
    """given_others_gt_mean_var_seq2seq.py:38 ships latent_dim = 32: the prediction runs zero-padded at 256 on the
    wide-input layer kernel + the ONE-launch decoder and matches the oracle at width 32."""
    from longterm360fov_amd.models import OthersMixingSeq2Seq, _MIX_ORDER
    U = 34
    w = O.init_others_mixing(5, H=32, num_user=U, bias_noise=0.05)
    enc, dec0, _, oth = O.synthetic_batch(6, 45, 4, 5, num_others=U - 1)
    m = OthersMixingSeq2Seq(latent_dim=32, num_user=U)
    m.set_weights([w[k] for k in _MIX_ORDER])
    assert m._run_width() == 256
    got = m.predict([enc, oth, dec0])
    ref = O.others_mixing_forward(enc.astype(np.float64), oth.astype(np.float64), dec0.astype(np.float64),
                                  {k: v.astype(np.float64) for k, v in w.items()})
    assert np.abs(got - ref).max() <= 2e-5
    g = OthersMixingSeq2Seq(latent_dim=32, num_user=U, impl="generic")
    g.set_weights([w[k] for k in _MIX_ORDER])
    np.testing.assert_allclose(g.predict([enc, oth, dec0]), got, atol=2e-5)


@pytest.mark.parametrize("B,T,F,H,act,state", [(32, 10, 90, 400, "sigmoid", False), (19, 3, 90, 400, "hard_sigmoid", True),
                                               (40, 4, 33, 300, "sigmoid", True), (5, 1, 7, 257, "sigmoid", False)])
def test_wide_hidden_layer_runs_step_wise_on_the_matrix_core_gemm(B, T, F, H, act, state):
    """H > 256 (lstm.py's LSTMCell(400)): impl='auto' runs the layer step-wise on the fp32 MFMA GEMM (x K for all steps as one
    product, h_{t-1} R + one pointwise launch per step) instead of the VALU kernel - against the fp64 oracle and against
    impl='generic'; hidden sequence, final states, the training tape, and the states-only form."""
    from longterm360fov_amd import ops
    from oracle import fov_oracle as O
    rng = np.random.default_rng(H + B)
    K, R, b = O.init_lstm(rng, F, H)
    b = (b + 0.1 * rng.standard_normal(b.shape)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    h0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    c0 = (0.3 * rng.standard_normal((B, H))).astype(np.float32) if state else None
    d64 = lambda a: None if a is None else a.astype(np.float64)
    hs_ref, hT_ref, cT_ref, res_ref = O.lstm_layer_train(d64(x), d64(K), d64(R), d64(b), d64(h0), d64(c0), act=act)
    dv = lambda a: None if a is None else torch.from_numpy(a).cuda()
    ws = ops.Workspace()
    hs, hT, cT = ops.lstm_seq(dv(x), dv(K), dv(R), dv(b), dv(h0), dv(c0), act=act, impl="auto", workspace=ws)
    ws.check()
    for got, ref, tag in ((hs, hs_ref, "hs"), (hT, hT_ref, "hT"), (cT, cT_ref, "cT")):
        err = np.abs(got.cpu().numpy() - ref).max()
        print("step-wise H=%d %s err %.3e" % (H, tag, err))
        assert err <= 2e-5, (tag, err)
    g_hs, g_hT, g_cT = ops.lstm_seq(dv(x), dv(K), dv(R), dv(b), dv(h0), dv(c0), act=act, impl="generic", workspace=ops.Workspace())
    assert (g_hs - hs).abs().max().item() <= 2e-5 and (g_cT - cT).abs().max().item() <= 2e-5
    hs2, hT2, cT2, res = ops.lstm_seq_train(dv(x), dv(K), dv(R), dv(b), dv(h0), dv(c0), act=act, workspace=ws)
    assert torch.equal(hs2, hs) and torch.equal(cT2, cT)
    assert np.abs(res.cpu().numpy() - res_ref).max() <= 2e-5
    _, hT3, cT3 = ops.lstm_seq(dv(x), dv(K), dv(R), dv(b), dv(h0), dv(c0), act=act, impl="auto", return_sequences=False, workspace=ws)
    assert torch.equal(hT3, hT) and torch.equal(cT3, cT)


@pytest.mark.parametrize("B,T_in,T_out,H,act", [(1024, 30, 30, 256, "sigmoid"), (37, 3, 5, 256, "hard_sigmoid"), (48, 4, 4, 128, "sigmoid"),
                                                (20, 2, 3, 64, "sigmoid"), (16 * 70, 3, 2, 256, "sigmoid")])
def test_one_launch_encode_decode_equals_two_launches(B, T_in, T_out, H, act):
    """The fused call runs encoder and decoder as ONE launch when every group has one tile (state handed over in registers,
    the h_T tile in LDS): bit-identical to the two-launch form (FOV_TWO_LAUNCHES=1), which more tiles than groups still take
    (the last case: 70 tiles on 64 groups), and to the oracle within the usual bound."""
    from longterm360fov_amd import ops
    w = O.init_seq2seq(400 + H, H=H, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(401 + B, B, T_in, T_out)
    dw = {k: torch.from_numpy(v).cuda() for k, v in w.items()}
    d_enc, d_dec0 = torch.from_numpy(enc).cuda(), torch.from_numpy(dec0).cuda()
    ws = ops.Workspace()
    one = ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, act=act, impl="cluster", workspace=ws).clone()
    ws.check()
    os.environ["FOV_TWO_LAUNCHES"] = "1"
    try:
        two = ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, act=act, impl="cluster", workspace=ws).clone()
    finally:
        del os.environ["FOV_TWO_LAUNCHES"]
    ws.check()
    assert torch.equal(one, two)
    n = min(B, 48)
    ref = O.seq2seq_decode(enc[:n].astype(np.float64), dec0[:n].astype(np.float64), {k: v.astype(np.float64) for k, v in w.items()},
                           T_out, act=act)
    assert np.abs(one[:n].cpu().numpy() - ref).max() <= 2e-5
    # a launch after the fused one continues from the header it left behind
    again = ops.seq2seq_decode(d_enc, d_dec0, dw, T_out, act=act, impl="cluster", workspace=ws)
    ws.check()
    assert torch.equal(again, one)


def test_launch_contract_resident_limit_and_epoch_rezero():
    """Hardening of the launch contract (include/fov360.h): (1) when fewer CUs are available than one group of workgroups
    needs (FOV_DBG_RESIDENT_LIMIT pretends so), an explicit impl='cluster' call is refused with FOV_ERR_UNSUPPORTED and
    impl='auto' falls back to the generic kernel - same result, no second-long spin; (2) the 32-bit epoch tags are re-zeroed
    by the launch path itself once the host-side account nears the limit, without fov_check_status ever being called."""
    from longterm360fov_amd import _lib
    ops = _ops()
    H, B, T_in, T_out = 256, 40, 4, 3
    w = O.init_seq2seq(77, H=H, bias_noise=0.1)
    enc, dec0, _ = O.synthetic_batch(78, B, T_in, T_out)
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), f64(w), T_out, "sigmoid")
    try:
        os.environ["FOV_DBG_RESIDENT_LIMIT"] = "2"       # < 4 workgroups of an H = 256 group
        ws = ops.Workspace()
        with pytest.raises(_lib.FovError) as ei:
            ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, impl="cluster", workspace=ws)
        assert ei.value.code == _lib.ERR_UNSUPPORTED
        out = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, impl="auto", workspace=ws)
        ws.check()
        assert_parity(out, ref, "auto falls back to the generic kernel under a resident limit")
    finally:
        os.environ.pop("FOV_DBG_RESIDENT_LIMIT", None)
    ws = ops.Workspace()
    out0 = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, impl="cluster", workspace=ws).clone()
    L = _lib.lib()
    check = _lib.check
    check(L.fov_debug_set_epoch(ws.buf.data_ptr(), ws.buf.numel(), 0x70000000 - 5, torch.cuda.current_stream().cuda_stream))
    for _ in range(4):      # the first of these crosses the threshold: header and granule area are re-zeroed in front of it
        out1 = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, impl="cluster", workspace=ws)
    torch.cuda.synchronize()
    hdr = ws.buf[:32].cpu().numpy().view(np.uint32)
    assert hdr[3] < 1000, "epoch base %d: the launch path did not re-zero the workspace" % hdr[3]
    ws.check()
    assert torch.equal(out0, out1)


# ---------------------------------------------------------------------------------------
# Round 4: the same call at small batches runs on H / 16 workgroups per tile (lstm_wide16.hip: wide16_s2s_kernel)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("H,B,T_in,T_out,F_enc,F_dec,act", [
    (128, 32, 10, 10, 90, 6, "sigmoid"),        # configs[0]
    (128, 1, 1, 1, 90, 6, "hard_sigmoid"), (128, 128, 3, 2, 96, 8, "sigmoid"), (128, 17, 0, 4, 90, 6, "sigmoid"),
    (256, 33, 2, 3, 6, 1, "hard_sigmoid"), (256, 100, 5, 1, 90, 6, "sigmoid"), (256, 16, 7, 5, 33, 3, "sigmoid"),
    (128, 50, 4, 0, 90, 6, "sigmoid"),          # no decoder step: the call returns an empty prediction, the states are the encoder's
])
def test_reference_batch_decode_on_sixteen_units_per_workgroup(H, B, T_in, T_out, F_enc, F_dec, act):
    """fov_seq2seq_decode_fwd at <= 128 sequences, impl = auto: encoder + free-running decoder in one launch on H / 16 workgroups
    per tile, the Dense formed from the gathered tile (FoV_seq2seq.py:154-178).  Against the fp64 oracle, against the VALU
    kernel, final states against the teacher-free oracle rollout; FOV_NO_WIDE16=1 (the H / 64 form) gives the same numbers to
    rounding."""
    import os
    ops = _ops()
    from longterm360fov_amd import _lib
    w = O.init_seq2seq(3 + H + B, F_enc=F_enc, F_dec=F_dec, H=H, bias_noise=0.1)
    rng = np.random.default_rng(B + T_in)
    enc = rng.uniform(-1, 1, (B, T_in, F_enc)).astype(np.float32)
    dec0 = rng.uniform(-1, 1, (B, 1, F_dec)).astype(np.float32)
    ws = ops.Workspace()
    out = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, act=act, impl="auto", workspace=ws)
    ws.check()
    assert tuple(out.shape) == (B, T_out, F_dec)
    if T_out == 0:
        return
    ref = O.seq2seq_decode(enc.astype(np.float64), dec0.astype(np.float64), f64(w), T_out, act)
    assert_parity(out, ref, "wide16 decode H%d B%d %d->%d F%d/%d %s" % (H, B, T_in, T_out, F_enc, F_dec, act))
    gen = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, act=act, impl="generic")
    assert float((gen - out).abs().max()) < 5e-6
    # twice on the same workspace (epoch tags continue): bit-identical
    again = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, act=act, impl="auto", workspace=ws)
    ws.check()
    assert torch.equal(again, out)
    os.environ["FOV_NO_WIDE16"] = "1"
    _lib.lib().fov_reload_env()
    try:
        old = ops.seq2seq_decode(dev(enc), dev(dec0), devw(w), T_out, act=act, impl="auto", workspace=ws)
        ws.check()
    finally:
        del os.environ["FOV_NO_WIDE16"]
        _lib.lib().fov_reload_env()
    assert float((old - out).abs().max()) < 5e-6


def test_ragged_batches_take_the_exchange_path_of_aligned_ones():
    """Grids are padded to a multiple of eight groups so that a group's workgroups sit 8 blocks apart - one XCD - at ANY batch
    (xch_common.h: xch_padded_groups; lstm_cluster.hip: xcd_pad).  Which exchange a launch takes is decided at run time by the
    hello handshake, so the check is relative: a batch whose tile count is no multiple of eight takes the same path as one
    whose is (both the fast one on an idle GPU) - for the fused four-workgroup kernel and for the eight-workgroup bf16 layer
    kernel.  FOV_NO_XCD_PAD=1 (the unpadded grid of the cluster kernels) deals the groups over the XCDs: mode 2."""
    from longterm360fov_amd import _lib
    ops = _ops()
    H, T_in, T_out = 256, 4, 4
    w = O.init_seq2seq(77, H=H, bias_noise=0.05)
    dw = devw(w)
    ws = ops.Workspace()
    modes = {}
    for B in (1024, 1000, 40):
        enc, dec0, _ = O.synthetic_batch(77, B, T_in, T_out)
        ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, impl="cluster", workspace=ws)
        ws.check()
        modes[B] = ws.exchange_mode()
    print("fused decode: exchange modes", modes)
    assert modes[1000] == modes[1024] and modes[40] == modes[1024]
    os.environ["FOV_NO_XCD_PAD"] = "1"
    _lib.lib().fov_reload_env()
    try:
        enc, dec0, _ = O.synthetic_batch(77, 1000, T_in, T_out)
        out = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, impl="cluster", workspace=ws)
        ws.check()
        assert ws.exchange_mode() == 2
    finally:
        os.environ.pop("FOV_NO_XCD_PAD", None)
        _lib.lib().fov_reload_env()
    out2 = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, impl="cluster", workspace=ws)
    assert torch.equal(out, out2)   # the two exchanges agree bitwise
    rng = np.random.default_rng(5)
    K, R, b = (dev(a) for a in O.init_lstm(rng, 90, 256, np.float32))
    lm = {}
    for B in (128, 100, 17):
        ops.lstm_seq_bf16(dev(rng.uniform(-1, 1, (B, 3, 90))), K, R, b, workspace=ws)
        ws.check()
        lm[B] = ws.exchange_mode()
    print("bf16 layer: exchange modes", lm)
    assert lm[100] == lm[128] and lm[17] == lm[128]


@pytest.mark.parametrize("B,T_in,T_out,F_enc,F_dec,act", [(1024, 30, 30, 90, 6, "sigmoid"), (530, 3, 4, 90, 6, "hard_sigmoid"), (1000, 2, 1, 33, 8, "sigmoid"),
                                                          (513, 1, 3, 96, 1, "sigmoid"), (784, 5, 2, 7, 3, "hard_sigmoid")])
def test_tile_pair_kernel_matches_oracle_and_the_four_workgroup_kernel(B, T_in, T_out, F_enc, F_dec, act):
    """lstm_pair.hip (round 5, FOV_PAIR=1: an experiment that did not beat the four-workgroup kernel and stays off by default): the
    fused seq2seq call at 33 .. 64 tiles, H = 256, on groups of eight workgroups that carry TWO tiles each (one tile's exchange
    under the other's MFMAs).  Against the C / fp64 oracles (1e-3 relative with the 2e-5 floor) and against the four-workgroup
    kernel (same arithmetic, another summation order); odd tile counts (the last group's second tile is absent), ragged last tiles, one-step phases,
    narrow and full-width inputs, final states, repeated launches on one workspace."""
    ops = _ops()
    from longterm360fov_amd import _lib
    H = 256
    assert _lib.lib().fov_version() >= 100
    w = O.init_seq2seq(900 + B, F_enc, F_dec, H, bias_noise=0.05)
    rng = np.random.default_rng(901 + B)
    enc = rng.uniform(-1, 1, (B, T_in, F_enc)).astype(np.float32)
    dec0 = rng.uniform(-1, 1, (B, 1, F_dec)).astype(np.float32)
    dw = devw(w)
    ws = ops.Workspace()
    hT, cT = (torch.empty((B, H), dtype=torch.float32, device="cuda") for _ in range(2))
    hT4, cT4 = (torch.empty((B, H), dtype=torch.float32, device="cuda") for _ in range(2))
    four = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, act=act, workspace=ws, hT=hT4, cT=cT4).clone()
    ws.check()
    os.environ["FOV_PAIR"] = "1"
    try:
        out = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, act=act, workspace=ws, hT=hT, cT=cT).clone()
        ws.check()
        again = ops.seq2seq_decode(dev(enc), dev(dec0), dw, T_out, act=act, workspace=ws)
        ws.check()
        perm = np.random.default_rng(3).permutation(B)
        outp = ops.seq2seq_decode(dev(enc[perm]), dev(dec0[perm]), dw, T_out, act=act, workspace=ws)
        ws.check()
    finally:
        del os.environ["FOV_PAIR"]
    assert torch.equal(again, out)
    assert torch.equal(outp, out[perm])         # a sequence does not see its tile-mates
    d = (out - four).abs().max().item()
    print("tile-pair vs four-workgroup kernel B=%d: max |diff| %.3e (state %.3e)" % (B, d, (hT - hT4).abs().max().item()))
    assert 0 < d <= 5e-6, d                     # (not bit-identical: the pair kernel really ran)
    assert (hT - hT4).abs().max().item() <= 5e-6 and (cT - cT4).abs().max().item() <= 2e-5
    n = min(B, 96)
    idx = np.r_[0:n // 2, B - n // 2:B]         # first and last sequences (the ragged / absent tile end)
    ref = O.seq2seq_decode(enc[idx].astype(np.float64), dec0[idx].astype(np.float64), f64(w), T_out, act=act)
    assert_parity(out[idx], ref, "tile-pair kernel B=%d" % B)
