"""GPU parity of the ConvLSTM2D path (a8/a9, mycode/convlstm_seq2seq.py) against the NumPy oracle:
implicit-GEMM conv2d, ConvLSTM2D cell, channel softmax and the 3+3-layer seq2seq with both heads."""
import os

import numpy as np
import pytest
import torch

from oracle import fov_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def close(got, ref, tag, tol=2e-5):
    got = got.detach().cpu().numpy().astype(np.float64) if isinstance(got, torch.Tensor) else np.asarray(got, np.float64)
    err = np.abs(got - ref)
    scale = max(np.abs(ref).max(), 1e-6)
    print("%s: max err %.3e (max |ref| %.3f)" % (tag, err.max(), scale))
    assert got.shape == ref.shape and np.isfinite(got).all()
    assert (err <= 1e-3 * np.abs(ref) + 1e-5 * scale).all(), tag
    assert err.max() <= tol * scale, tag


@pytest.mark.parametrize("B,H,W,C,N,kh,kw", [(2, 4, 5, 3, 7, 5, 5), (3, 36, 18, 30, 128, 5, 5), (2, 1, 30, 3, 128, 5, 5),
                                             (2, 1, 30, 56, 40, 1, 7), (1, 6, 7, 17, 33, 3, 3), (2, 9, 4, 8, 64, 5, 5)])
def test_conv2d_same(B, H, W, C, N, kh, kw):
    from longterm360fov_amd import ops
    rng = np.random.default_rng(B * 100 + N)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, C, N)) / np.sqrt(kh * kw * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((B, H, W, N)).astype(np.float32)
    ref = O.conv2d_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    close(ops.conv2d(dev(x), dev(w), dev(b)), ref, "conv2d")
    close(ops.conv2d(dev(x), dev(w), dev(b), add=dev(add), activation="relu"), np.maximum(ref + add, 0), "conv2d+add+relu")
    # strided input: a channel slice of a wider map
    wide = rng.standard_normal((B, H, W, C + 9)).astype(np.float32)
    ref2 = O.conv2d_same(wide[..., 4:4 + C].astype(np.float64), w.astype(np.float64))
    close(ops.conv2d(dev(wide)[..., 4:4 + C], dev(w)), ref2, "conv2d strided input")
    # one time step of a (B,T,H,W,C) sequence: batch stride != H*W*C
    seq = rng.standard_normal((B, 3, H, W, C)).astype(np.float32)
    ref3 = O.conv2d_same(seq[:, 1].astype(np.float64), w.astype(np.float64))
    close(ops.conv2d(dev(seq)[:, 1], dev(w)), ref3, "conv2d batch-strided input")


@pytest.mark.parametrize("act", ["hard_sigmoid", "sigmoid"])
def test_convlstm_cell_and_softmax(act):
    from longterm360fov_amd import ops
    rng = np.random.default_rng(5)
    B, H, W, C, F = 3, 6, 5, 4, 8
    K = (rng.standard_normal((5, 5, C, 4 * F)) * 0.2).astype(np.float32)
    R = (rng.standard_normal((5, 5, F, 4 * F)) * 0.2).astype(np.float32)
    b = rng.standard_normal(4 * F).astype(np.float32)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    h = (0.5 * rng.standard_normal((B, H, W, F))).astype(np.float32)
    c = (0.5 * rng.standard_normal((B, H, W, F))).astype(np.float32)
    d = lambda a: a.astype(np.float64)
    h_ref, c_ref = O.convlstm2d_step(d(x), d(h), d(c), d(K), d(R), d(b), act)
    z = ops.conv2d(dev(x), dev(K), dev(b))
    z = ops.conv2d(dev(h), dev(R), None, add=z, out=z)
    cc = dev(c)
    wide = torch.zeros((B, H, W, F + 5), dtype=torch.float32, device="cuda")
    ops.convlstm_gates(z, cc, wide[..., 3:3 + F], act)
    close(wide[..., 3:3 + F], h_ref, "convlstm h " + act)
    close(cc, c_ref, "convlstm c " + act)
    assert float(wide[..., :3].abs().max()) == 0 and float(wide[..., 3 + F:].abs().max()) == 0
    y = rng.standard_normal((7, 3, 30)).astype(np.float32) * 3
    close(ops.softmax_lastdim(dev(y)), O.softmax_last(d(y)), "softmax", tol=1e-6)


@pytest.mark.parametrize("head,B,T_in,T_out,H,W,C,L,hf", [("conv2d", 2, 3, 2, 9, 6, 10, 8, (24, 40)),
                                                           ("conv1d", 3, 4, 3, 1, 30, 3, 16, (32, 48)),
                                                           ("conv2d", 1, 2, 2, 36, 18, 30, 16, (64, 96))])
def test_convlstm_seq2seq(head, B, T_in, T_out, H, W, C, L, hf):
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    w = O.init_convlstm_seq2seq(3, C=C, latent_dim=L, head=head, head_filters=hf)
    rng = np.random.default_rng(4)
    if head == "conv2d":     # one-hot maps: one active cell per frame channel
        enc = np.zeros((B, T_in, H, W, C), np.float32)
        idx = rng.integers(0, H * W, (B, T_in, C))
        for bb in range(B):
            for t in range(T_in):
                for ch in range(C):
                    enc[bb, t].reshape(H * W, C)[idx[bb, t, ch], ch] = 1
    else:
        enc = O.synthetic_xyz(rng, B, T_in, 30).reshape(B, T_in, 1, 30, 3)
    dec0 = enc[:, -1:]
    d = lambda a: a.astype(np.float64)
    ref = O.convlstm_seq2seq_forward(d(enc), d(dec0), {k: d(v) for k, v in w.items()}, T_out, head)
    m = ConvLSTMSeq2Seq(w, head=head)
    out = m.predict([enc, dec0], predict_step=T_out)
    close(out, ref, "convlstm seq2seq " + head, tol=5e-5)
    np.testing.assert_allclose(out.sum(-1), 1.0, atol=1e-5)


# ---------------------------------------------------------------------------------------
# training kernels (a8 backward): checked against torch.autograd in fp64 on the CPU
# ---------------------------------------------------------------------------------------
def _tconv(x, w, dilation=1):
    """conv2d_same on NHWC / (kh,kw,C,N) operands with torch (independent reference)."""
    import torch.nn.functional as TF
    kh, kw = w.shape[:2]
    return TF.conv2d(x.permute(0, 3, 1, 2), w.permute(3, 2, 0, 1), padding=(dilation * (kh // 2), dilation * (kw // 2)),
                     dilation=dilation).permute(0, 2, 3, 1)


@pytest.mark.parametrize("B,H,W,C,N,kh,kw", [(2, 4, 5, 3, 7, 5, 5), (3, 36, 18, 30, 128, 5, 5), (2, 1, 30, 3, 128, 5, 5),
                                             (2, 1, 30, 56, 40, 1, 7), (1, 6, 7, 17, 33, 3, 3), (2, 9, 4, 8, 64, 5, 5),
                                             (2, 6, 6, 128, 12, 3, 3), (1, 6, 6, 40, 260, 3, 3), (5, 3, 3, 4, 4, 5, 5), (1, 36, 18, 260, 30, 5, 5)])
def test_conv2d_backward_kernels(B, H, W, C, N, kh, kw):
    from longterm360fov_amd import ops
    rng = np.random.default_rng(B * 1000 + C * 10 + N)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, C, N)) / np.sqrt(kh * kw * C)).astype(np.float32)
    dy = rng.standard_normal((B, H, W, N)).astype(np.float32)
    tx = torch.tensor(x.astype(np.float64), requires_grad=True)
    tw = torch.tensor(w.astype(np.float64), requires_grad=True)
    (_tconv(tx, tw) * torch.tensor(dy.astype(np.float64))).sum().backward()
    dw_ref, dx_ref = tw.grad.numpy(), tx.grad.numpy()
    dw = ops.conv2d_wgrad(dev(x), dev(dy), kh, kw)
    close(dw, dw_ref, "wgrad")
    ops.conv2d_wgrad(dev(x), dev(dy), kh, kw, dw=dw, accumulate=True)
    close(dw, 2 * dw_ref, "wgrad accumulate")
    dx = ops.conv2d(dev(dy), ops.conv2d_weight_transpose(dev(w)))
    close(dx, dx_ref, "dgrad = conv(dy, w')")
    # x as a channel slice of a wider map, time-major stack of two steps (leading dims flatten into the batch)
    wide = rng.standard_normal((2, B, H, W, C + 8)).astype(np.float32)
    dy2 = rng.standard_normal((2, B, H, W, N)).astype(np.float32)
    tx2 = torch.tensor(wide[..., 4:4 + C].reshape(2 * B, H, W, C).astype(np.float64))
    tw2 = torch.tensor(w.astype(np.float64), requires_grad=True)
    (_tconv(tx2, tw2) * torch.tensor(dy2.reshape(2 * B, H, W, N).astype(np.float64))).sum().backward()
    close(ops.conv2d_wgrad(dev(wide)[..., 4:4 + C], dev(dy2), kh, kw), tw2.grad.numpy(), "wgrad on a channel slice")


@pytest.mark.parametrize("B,H,W,C,N,k,slice_x", [(7, 36, 18, 30, 128, 5, True), (5, 36, 18, 8, 32, 5, False), (3, 36, 18, 56, 512, 5, False),
                                                 (2, 36, 18, 1024, 30, 5, False), (4, 18, 36, 16, 64, 5, True), (3, 7, 9, 20, 24, 3, False),
                                                 (70, 5, 6, 32, 64, 5, False), (2, 64, 64, 12, 16, 3, False), (3, 10, 11, 100, 16, 5, True)])
def test_conv2d_weight_gradient_all_taps_per_workgroup(B, H, W, C, N, k, slice_x):
    """conv_wgrad_lines.hip (round 5): the weight gradient of a narrow 'same' convolution with every tap in one workgroup and the
    maps walked line by line (rows, or columns when only that length is a multiple of 4) - against fp64, against the tap-wise kernel
    it replaces (FOV_NO_WGRAD_LINES=1), with accumulate, on a channel slice of a wider map, and bit for bit the same twice.  Shapes: the ConvLSTM model's cells and narrow head layers (convlstm_seq2seq.py:100-126,
    209-282 at 36 x 18), 18 x 36, odd line lengths, more maps than slices, 64-pixel lines, a channel count that is no multiple of 16."""
    import os
    from longterm360fov_amd import ops
    rng = np.random.default_rng(B * 100 + C + N)
    wide = rng.standard_normal((B, H, W, C + 8)).astype(np.float32)
    xw = dev(wide)
    x = xw[..., 4:4 + C] if slice_x else dev(np.ascontiguousarray(wide[..., :C]))
    xn = wide[..., 4:4 + C] if slice_x else wide[..., :C]
    dy = rng.standard_normal((B, H, W, N)).astype(np.float32)
    h = k // 2
    xp = np.pad(xn.astype(np.float64), ((0, 0), (h, h), (h, h), (0, 0)))
    ref = np.zeros((k, k, C, N))
    for i in range(k):
        for j in range(k):
            ref[i, j] = np.einsum("bhwc,bhwn->cn", xp[:, i:i + H, j:j + W], dy.astype(np.float64))
    scale = np.abs(ref).max()
    dw = ops.conv2d_wgrad(x, dev(dy), k, k, scratch=ops.Scratch())
    assert np.abs(dw.cpu().numpy() - ref).max() <= 3e-6 * scale * max(1.0, np.sqrt(B * H * W / 1000.0))
    dw2 = ops.conv2d_wgrad(x, dev(dy), k, k, scratch=ops.Scratch())
    assert torch.equal(dw, dw2)
    os.environ["FOV_NO_WGRAD_LINES"] = "1"
    try:
        old = ops.conv2d_wgrad(x, dev(dy), k, k, scratch=ops.Scratch())
    finally:
        del os.environ["FOV_NO_WGRAD_LINES"]
    ops.Scratch().get(256, dw.device)      # (the library re-reads its knobs)
    assert (dw - old).abs().max().item() <= 3e-6 * scale * max(1.0, np.sqrt(B * H * W / 1000.0))
    acc = dw.clone()
    ops.conv2d_wgrad(x, dev(dy), k, k, dw=acc, accumulate=True, scratch=ops.Scratch())
    assert (acc - 2 * dw).abs().max().item() <= 1e-6 * scale


@pytest.mark.parametrize("act", ["hard_sigmoid", "sigmoid"])
def test_convlstm_gates_backward_softmax_relu_colsum(act):
    from longterm360fov_amd import ops
    rng = np.random.default_rng(11)
    B, H, W, F = 3, 5, 4, 6
    z = rng.standard_normal((B, H, W, 4 * F)).astype(np.float32) * 2
    cp = rng.standard_normal((B, H, W, F)).astype(np.float32)
    dh = rng.standard_normal((B, H, W, F)).astype(np.float32)
    dc = rng.standard_normal((B, H, W, F)).astype(np.float32)
    s = torch.sigmoid if act == "sigmoid" else (lambda v: torch.clamp(0.2 * v + 0.5, 0, 1))
    tz = torch.tensor(z.astype(np.float64), requires_grad=True)
    tcp = torch.tensor(cp.astype(np.float64), requires_grad=True)
    i, f, g, o = s(tz[..., :F]), s(tz[..., F:2 * F]), torch.tanh(tz[..., 2 * F:3 * F]), s(tz[..., 3 * F:])
    cn = f * tcp + i * g
    hn = o * torch.tanh(cn)
    ((hn * torch.tensor(dh.astype(np.float64))).sum() + (cn * torch.tensor(dc.astype(np.float64))).sum()).backward()
    wide = torch.zeros((B, H, W, F + 3), dtype=torch.float32, device="cuda")
    h_out, c_new, gates = ops.convlstm_gates_train(dev(z), dev(cp), wide[..., 2:2 + F], act)
    close(h_out, hn.detach().numpy(), "train gates h")
    close(c_new, cn.detach().numpy(), "train gates c")
    dwide = torch.zeros((B, H, W, F + 5), dtype=torch.float32, device="cuda")
    dwide[..., 1:1 + F] = dev(dh)
    dcv = dev(dc)
    dz = ops.convlstm_gates_bwd(dwide[..., 1:1 + F], dcv, gates, dev(cp), c_new, act)
    close(dz, tz.grad.numpy(), "gates bwd dz " + act, tol=5e-5)
    close(dcv, tcp.grad.numpy(), "gates bwd dc_prev " + act, tol=5e-5)
    # zero initial state form
    h0, c0n, g0 = ops.convlstm_gates_train(dev(z), None, torch.empty((B, H, W, F), device="cuda"), act)
    close(c0n, (i * g).detach().numpy(), "train gates c from zero state")
    # softmax / relu backward, column sums
    y = rng.standard_normal((7, 3, 30)).astype(np.float32) * 2
    dp = rng.standard_normal((7, 3, 30)).astype(np.float32)
    ty = torch.tensor(y.astype(np.float64), requires_grad=True)
    (torch.softmax(ty, -1) * torch.tensor(dp.astype(np.float64))).sum().backward()
    close(ops.softmax_lastdim_bwd(dev(dp), ops.softmax_lastdim(dev(y))), ty.grad.numpy(), "softmax bwd", tol=1e-5)
    r = np.maximum(y, 0)
    close(ops.act_bwd(dev(dp), dev(r), activation="relu"), dp * (r > 0), "relu bwd", tol=1e-6)
    big = rng.standard_normal((5000, 40)).astype(np.float32)
    close(ops.colsum(dev(big)), big.astype(np.float64).sum(0), "colsum")


def _torch_convlstm_graph(enc, dec0, tgt, w, head, act, masks=None, xyz_sum1=False, xent=False, dilation=1):
    """Independent fp64 restatement of the ConvLSTM seq2seq training graph (convlstm_seq2seq.py:100-287) on
    torch.autograd: loss = mean squared error of the unrolled, self-fed decoder.  `masks` (optional): Keras
    ConvLSTM2D input dropout - per layer call four masks, gate g's kernel slice convolves x * mask_g."""
    t = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in w.items()}
    s = torch.sigmoid if act == "sigmoid" else (lambda z: torch.clamp(0.2 * z + 0.5, 0, 1))

    def cell(x, h, c, K, R, b, m4=None):
        F = R.shape[2]
        if m4 is None:
            zx = _tconv(x, K, dilation)
        else:
            zx = torch.cat([_tconv(x * m4[g], K[..., g * F:(g + 1) * F], dilation) for g in range(4)], -1)
        z = zx + b + _tconv(h, R)
        i, f, g, o = s(z[..., :F]), s(z[..., F:2 * F]), torch.tanh(z[..., 2 * F:3 * F]), s(z[..., 3 * F:])
        c = f * c + i * g
        return o * torch.tanh(c), c

    e, d0, tg = (torch.tensor(a.astype(np.float64)) for a in (enc, dec0, tgt))
    B, T_in, H, W, _ = e.shape
    seq = [e[:, tt] for tt in range(T_in)]
    states = []
    for l in range(3):
        F = w["enc%d_R" % l].shape[2]
        h = torch.zeros(B, H, W, F, dtype=torch.float64)
        c = torch.zeros(B, H, W, F, dtype=torch.float64)
        nxt = []
        m4 = None if masks is None else torch.tensor(masks["enc%d" % l].astype(np.float64))
        for tt in range(T_in):
            h, c = cell(seq[tt], h, c, t["enc%d_K" % l], t["enc%d_R" % l], t["enc%d_b" % l], m4)
            nxt.append(h)
        seq = nxt
        states.append([h, c])
    inp = d0[:, 0]
    outs = []
    for tt in range(tg.shape[1]):
        cur, feats = inp, []
        for l in range(3):
            m4 = None if masks is None else torch.tensor(masks["dec%d" % l][tt].astype(np.float64))
            h, c = cell(cur, states[l][0], states[l][1], t["dec%d_K" % l], t["dec%d_R" % l], t["dec%d_b" % l], m4)
            states[l] = [h, c]
            feats.append(h)
            cur = h
        y = torch.cat(feats, -1)
        if head == "dense":
            y = y.reshape(B, -1) @ t["head0_W"] + t["head0_b"]
            outs.append(y)
            inp = y.reshape(B, 1, 1, -1)
            continue
        y = torch.relu(_tconv(y, t["head0_W"]) + t["head0_b"])
        y = torch.relu(_tconv(y, t["head1_W"]) + t["head1_b"])
        y = _tconv(y, t["head2_W"]) + t["head2_b"]
        y = torch.softmax(torch.relu(y) if head == "conv2d" else y, -1)
        outs.append(y)
        inp = y
    P = torch.stack(outs, 1)
    loss = torch.mean((P - tg) ** 2)
    if xent:          # Keras-2.2 categorical_crossentropy, TF backend (convlstm_heatmap.py:192)
        q = torch.clamp(P / P.sum(-1, keepdim=True), 1e-7, 1 - 1e-7)
        loss = torch.mean(-(tg * torch.log(q)).sum(-1))
    if xyz_sum1:      # costfunc._mse under cfg.add_xyz_sum1 (cost.py:23-28)
        loss = loss + 0.5 * torch.mean((1 - (P[..., 0] ** 2 + P[..., 1] ** 2 + P[..., 2] ** 2)) ** 2)
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in t.items()}, P.detach().numpy()


@pytest.mark.parametrize("head,B,T_in,T_out,H,W,C,L,hf,act", [("conv2d", 2, 3, 3, 9, 6, 10, 8, (24, 40), "hard_sigmoid"),
                                                               ("conv1d", 3, 2, 3, 1, 30, 3, 16, (32, 48), "sigmoid"),
                                                               ("conv2d", 1, 2, 2, 36, 18, 30, 16, (40, 136), "sigmoid"),
                                                               ("dense", 5, 3, 4, 1, 1, 6, 8, (), "hard_sigmoid")])
def test_convlstm_seq2seq_gradients_and_training(head, B, T_in, T_out, H, W, C, L, hf, act):
    """a8 backward: gradients of the whole unrolled graph (feedback path through the head included) against
    torch.autograd in fp64; the torch graph itself is pinned to the NumPy oracle's forward; then RMSprop steps
    reduce the loss."""
    from longterm360fov_amd.training import ConvLSTMTrainer
    w = O.init_convlstm_seq2seq(5, C=C, latent_dim=L, head=head, head_filters=hf, map_hw=(H, W))
    rng = np.random.default_rng(8)
    for k in w:
        if k.endswith("_b"):
            w[k] = (w[k] + 0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
    enc = rng.random((B, T_in, H, W, C)).astype(np.float32)
    dec0 = enc[:, -1:].copy()
    if head == "dense":     # predict_mean_var + input_mean_var: 1x1 maps of (mu, sigma^2), output (B,T_out,6)
        enc = (2 * enc - 1).astype(np.float32)
        dec0 = enc[:, -1:].copy()
        tgt = (2 * rng.random((B, T_out, 6)) - 1).astype(np.float32)
    else:
        tgt = rng.random((B, T_out, H, W, C)).astype(np.float32)
        tgt /= tgt.sum(-1, keepdims=True)
    loss_ref, g_ref, P_ref = _torch_convlstm_graph(enc, dec0, tgt, w, head, act)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    P_orc = O.convlstm_seq2seq_forward(enc.astype(np.float64), dec0.astype(np.float64), w64, T_out, head=head, act=act)
    assert np.abs(P_orc - P_ref).max() < 1e-10        # the autograd graph computes the oracle's function
    tr = ConvLSTMTrainer(w, head=head, act=act)
    loss, P = tr.forward_backward(dev(enc), dev(dec0), dev(tgt))
    close(P, P_ref, "train-mode forward")
    assert abs(float(loss.item()) - loss_ref) <= 1e-5 * loss_ref + 1e-9
    for k in tr.order:
        a = tr.g[k].detach().cpu().numpy().astype(np.float64)
        scale = np.abs(g_ref[k]).max()
        err = np.abs(a - g_ref[k]).max()
        print("convlstm %s grad %-8s max|ref| %.3e  max err %.3e" % (head, k, scale, err))
        assert err <= 2e-4 * scale + 1e-9, (k, err, scale)
    losses = [float(tr.train_step(dev(enc), dev(dec0), dev(tgt)).item()) for _ in range(4)]
    assert losses[-1] < losses[0]


def test_convlstm_fit_surface():
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    np.random.seed(2)
    rng = np.random.default_rng(3)
    w = O.init_convlstm_seq2seq(9, C=3, latent_dim=8, head="conv1d", head_filters=(16, 24))
    enc = rng.random((24, 3, 1, 30, 3)).astype(np.float32)
    tgt = rng.random((24, 2, 1, 30, 3)).astype(np.float32)
    tgt /= tgt.sum(-1, keepdims=True)
    m = ConvLSTMSeq2Seq(w, head="conv1d")
    with pytest.raises(RuntimeError):
        m.fit([enc, enc[:, -1:]], tgt)
    m.compile(optimizer="RMSprop", loss="mean_squared_error")
    h = m.fit([enc, enc[:, -1:]], tgt, batch_size=8, epochs=3, validation_split=0.25, shuffle=True)
    assert len(h.history["loss"]) == 3 and h.history["loss"][-1] < h.history["loss"][0] and "val_loss" in h.history
    assert m.predict([enc[:4], enc[:4, -1:]], predict_step=2).shape == (4, 2, 1, 30, 3)
    md = ConvLSTMSeq2Seq(w, head="conv1d", dropout_rate=0.3)      # Keras default of the script (cfg.dropout_rate)
    md.compile(optimizer="RMSprop", loss="mse")
    assert np.isfinite(md.train_on_batch([enc[:8], enc[:8, -1:]], tgt[:8]))


def test_convlstm_dense_head_predict():
    """cfg.predict_mean_var + cfg.input_mean_var (convlstm_seq2seq.py:96,171,225-227,272-273): 1x1 maps of the six
    (mu, sigma^2) values, Flatten + Dense(6) head, output (N,T_out,6) fed back as the next input map."""
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    rng = np.random.default_rng(21)
    w = O.init_convlstm_seq2seq(4, C=6, latent_dim=16, head="dense", map_hw=(1, 1))
    enc = (2 * rng.random((7, 5, 1, 1, 6)) - 1).astype(np.float32)
    d = lambda a: a.astype(np.float64)
    ref = O.convlstm_seq2seq_forward(d(enc), d(enc[:, -1:]), {k: d(v) for k, v in w.items()}, 4, "dense")
    out = ConvLSTMSeq2Seq(w, head="dense").predict([enc, enc[:, -1:]], predict_step=4)
    assert out.shape == (7, 4, 6)
    close(out, ref, "convlstm seq2seq dense head", tol=5e-5)


@pytest.mark.parametrize("head,B,T_in,T_out,H,W,C,L,hf", [("conv2d", 2, 3, 2, 7, 5, 6, 8, (12, 20)),
                                                           ("conv1d", 2, 2, 3, 1, 30, 3, 8, (16, 24))])
def test_convlstm_input_dropout_gradients(head, B, T_in, T_out, H, W, C, L, hf):
    """ConvLSTM2D(dropout=cfg.dropout_rate): four per-gate input masks per layer call (encoder: one set per
    batch; decoder: one per unrolled step).  With the same masks the gradients match torch.autograd in fp64."""
    from longterm360fov_amd.training import ConvLSTMTrainer
    w = O.init_convlstm_seq2seq(6, C=C, latent_dim=L, head=head, head_filters=hf)
    rng = np.random.default_rng(12)
    enc = rng.random((B, T_in, H, W, C)).astype(np.float32)
    dec0 = enc[:, -1:].copy()
    tgt = rng.random((B, T_out, H, W, C)).astype(np.float32)
    tgt /= tgt.sum(-1, keepdims=True)
    tr = ConvLSTMTrainer(w, head=head, act="hard_sigmoid", dropout_rate=0.3, seed=5)
    masks = tr.sample_masks(B, H, W, C, T_out)
    keep = [float((m > 0).float().mean()) for m in masks.values()]
    assert all(0.5 < k < 0.9 for k in keep) and abs(float(masks["enc0"].max()) - 1 / 0.7) < 1e-6
    loss_ref, g_ref, P_ref = _torch_convlstm_graph(enc, dec0, tgt, w, head, "hard_sigmoid",
                                                   masks={k: v.cpu().numpy() for k, v in masks.items()})
    loss, P = tr.forward_backward(dev(enc), dev(dec0), dev(tgt), masks=masks)
    close(P, P_ref, "dropout forward")
    assert abs(float(loss.item()) - loss_ref) <= 1e-5 * loss_ref + 1e-9
    for k in tr.order:
        a = tr.g[k].detach().cpu().numpy().astype(np.float64)
        scale = np.abs(g_ref[k]).max()
        assert np.abs(a - g_ref[k]).max() <= 2e-4 * scale + 1e-9, k
    # evaluation never drops anything
    P_eval, _ = tr._forward(dev(enc), dev(dec0), T_out)
    _, _, P_plain = _torch_convlstm_graph(enc, dec0, tgt, w, head, "hard_sigmoid")
    close(P_eval.transpose(0, 1), P_plain, "eval forward without dropout")


@pytest.mark.parametrize("B,H,W,C1,C2,N,k", [(2, 6, 5, 30, 32, 128, 5), (3, 36, 18, 32, 16, 64, 5), (2, 1, 30, 3, 8, 32, 5),
                                             (1, 7, 4, 16, 8, 32, 3), (2, 5, 5, 17, 20, 12, 3)])
def test_conv2d_over_channel_concatenation(B, H, W, C1, C2, N, k):
    """One ConvLSTM2D step as ONE convolution: conv([x | h], [K ; R]) over two separate maps (fov_conv2d_fwd2) equals the
    convolution of the concatenated map, for vector and scalar channel paths, channel-slice and batch-strided views."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(C1 * 10 + C2)
    x1 = rng.standard_normal((B, 2, H, W, C1)).astype(np.float32)            # a (B,T,...) sequence: batch-strided x_t
    wide = rng.standard_normal((B, H, W, C2 + 8)).astype(np.float32)         # h as a channel slice of a wider map
    w = (rng.standard_normal((k, k, C1 + C2, N)) / np.sqrt(k * k * (C1 + C2))).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    cat = np.concatenate([x1[:, 1], wide[..., 4:4 + C2]], -1)
    ref = O.conv2d_same(cat.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = ops.conv2d_cat(dev(x1)[:, 1], dev(wide)[..., 4:4 + C2], dev(w), dev(b))
    close(got, ref, "conv over [x | h]")
    close(ops.conv2d_cat(dev(x1)[:, 1], dev(wide)[..., 4:4 + C2], dev(w), dev(b), activation="relu"), np.maximum(ref, 0), "relu")


@pytest.mark.parametrize("act", ["hard_sigmoid", "sigmoid"])
@pytest.mark.parametrize("B,H,W,C,F,k", [(3, 36, 18, 32, 128, 5), (2, 6, 5, 30, 32, 5), (2, 1, 30, 3, 8, 5), (1, 7, 4, 17, 20, 3),
                                         (2, 9, 4, 8, 50, 3), (5, 36, 18, 128, 64, 5), (3, 36, 18, 32, 16, 5), (3, 36, 18, 16, 8, 5),
                                         (2, 5, 4, 6, 6, 3), (2, 5, 4, 7, 13, 3)])
def test_convlstm_cell_one_launch(B, H, W, C, F, k, act):
    """fov_convlstm_cell_fwd (convolution over [x | h], gates, c / h update and the gates tape in ONE launch, the four gate
    columns of a unit brought into one lane by the weight staging) against the oracle's ConvLSTM2D step, and bit for bit
    against the two-launch form (fov_conv2d_fwd2 + fov_convlstm_gates_train); ragged F (not a multiple of 32 / of 4), scalar and
    vector channel paths, channel-slice and batch-strided views, zero initial state, in-place c."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(C * 7 + F)
    K = (rng.standard_normal((k, k, C, 4 * F)) / np.sqrt(k * k * C)).astype(np.float32)
    R = (rng.standard_normal((k, k, F, 4 * F)) / np.sqrt(k * k * F)).astype(np.float32)
    b = rng.standard_normal(4 * F).astype(np.float32)
    xs = rng.standard_normal((B, 2, H, W, C)).astype(np.float32)                 # x_t of a (B,T,...) sequence
    hw = (0.5 * rng.standard_normal((B, H, W, F + 8))).astype(np.float32)        # h_prev: channel slice of a wider map
    c = (0.5 * rng.standard_normal((B, H, W, F))).astype(np.float32)
    x, h = xs[:, 1], hw[..., 4:4 + F]
    d = lambda a: a.astype(np.float64)
    h_ref, c_ref = O.convlstm2d_step(d(x), d(h), d(c), d(K), d(R), d(b), act)
    KR = torch.cat([dev(K), dev(R)], 2).contiguous()
    xd, hd = dev(xs)[:, 1], dev(hw)[..., 4:4 + F]
    wide = torch.zeros((B, H, W, F + 5), dtype=torch.float32, device="cuda")
    gates = torch.empty((B, H, W, 4 * F), dtype=torch.float32, device="cuda")
    c_new = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
    ops.convlstm_cell(xd, hd, KR, dev(b), dev(c), wide[..., 3:3 + F], act, c_new=c_new, gates=gates)
    close(wide[..., 3:3 + F], h_ref, "cell h")
    close(c_new, c_ref, "cell c")
    assert float(wide[..., :3].abs().max()) == 0 and float(wide[..., 3 + F:].abs().max()) == 0
    # the two-launch form computes the same sums in the same order
    z = ops.conv2d_cat(xd, hd, KR, dev(b))
    h2 = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
    _, c2, g2 = ops.convlstm_gates_train(z, dev(c), h2, act)
    assert torch.equal(h2, wide[..., 3:3 + F]) and torch.equal(c2, c_new) and torch.equal(g2, gates)
    # in-place cell state, no tape
    cc = dev(c)
    h3 = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
    ops.convlstm_cell(xd, hd, KR, dev(b), cc, h3, act)
    assert torch.equal(cc, c_new) and torch.equal(h3, h2)
    # zero initial state: K alone
    h0_ref, c0_ref = O.convlstm2d_step(d(x), np.zeros_like(d(h)), np.zeros_like(d(c)), d(K), d(R), d(b), act)
    h4 = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
    _, c4 = ops.convlstm_cell(xd, None, dev(K), dev(b), None, h4, act)
    close(h4, h0_ref, "cell h, zero state")
    close(c4, c0_ref, "cell c, zero state")


@pytest.mark.parametrize("act", ["hard_sigmoid", "sigmoid"])
@pytest.mark.parametrize("B,H,W,C,F,k,state", [(3, 36, 18, 32, 32, 5, True), (2, 20, 18, 32, 32, 5, False), (2, 36, 18, 16, 32, 5, True),
                                               (2, 7, 9, 12, 16, 3, True), (1, 36, 18, 16, 8, 5, True), (2, 5, 112, 8, 8, 3, True),
                                               (2, 3, 40, 4, 16, 5, False), (1, 36, 18, 32, 16, 5, True)])
def test_convlstm_cell_patch_form(B, H, W, C, F, k, state, act):
    """The LDS-resident-patch form of the ConvLSTM2D step (convlstm_patch.hip: the halo patch of [x | h_prev] staged once, every
    tap the same patch at a shifted address, weights straight from L2, no barrier in the k loop) for the widths the model
    ships (convlstm_seq2seq.py: filters 32 / 16 / 8): against the fp64 oracle and against the implicit-GEMM cell
    (FOV_NO_CELL_PATCH=1).  Covers a partial last row group (H = 20: 6 + 6 + 6 + 2 rows), channel counts that are not a
    multiple of 16 (short last block, zero-padded in LDS), a map as wide as the block allows, the zero initial state, the
    gates tape and channel-slice views.  With C a multiple of 16 the two forms add in the same order: bit-identical."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(C * 11 + F + H)
    K = (rng.standard_normal((k, k, C, 4 * F)) / np.sqrt(k * k * C)).astype(np.float32)
    R = (rng.standard_normal((k, k, F, 4 * F)) / np.sqrt(k * k * F)).astype(np.float32)
    b = rng.standard_normal(4 * F).astype(np.float32)
    xs = rng.standard_normal((B, 2, H, W, C)).astype(np.float32)
    hw = (0.5 * rng.standard_normal((B, H, W, F + 8))).astype(np.float32)
    c = (0.5 * rng.standard_normal((B, H, W, F))).astype(np.float32)
    x, h = xs[:, 1], hw[..., 4:4 + F]
    d = lambda a: a.astype(np.float64)
    if state:
        h_ref, c_ref = O.convlstm2d_step(d(x), d(h), d(c), d(K), d(R), d(b), act)
        w = torch.cat([dev(K), dev(R)], 2).contiguous()
    else:
        h_ref, c_ref = O.convlstm2d_step(d(x), np.zeros_like(d(h)), np.zeros_like(d(c)), d(K), d(R), d(b), act)
        w = dev(K)
    xd, hd = dev(xs)[:, 1], (dev(hw)[..., 4:4 + F] if state else None)

    def run():
        wide = torch.zeros((B, H, W, F + 5), dtype=torch.float32, device="cuda")
        gates = torch.empty((B, H, W, 4 * F), dtype=torch.float32, device="cuda")
        c_new = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
        ops.convlstm_cell(xd, hd, w, dev(b), dev(c) if state else None, wide[..., 3:3 + F], act, c_new=c_new, gates=gates)
        return wide, c_new, gates
    wide, c_new, gates = run()
    close(wide[..., 3:3 + F], h_ref, "patch cell h")
    close(c_new, c_ref, "patch cell c")
    assert float(wide[..., :3].abs().max()) == 0 and float(wide[..., 3 + F:].abs().max()) == 0
    os.environ["FOV_NO_CELL_PATCH"] = "1"
    try:
        wide2, c2, g2 = run()
    finally:
        del os.environ["FOV_NO_CELL_PATCH"]
    if C % 16 == 0:
        assert torch.equal(wide2, wide) and torch.equal(c2, c_new) and torch.equal(g2, gates)
    else:
        assert (wide2 - wide).abs().max().item() <= 2e-6 and (c2 - c_new).abs().max().item() <= 2e-6
        assert (g2 - gates).abs().max().item() <= 2e-6
    wide3, c3, g3 = run()                                         # back on the patch form, deterministic
    assert torch.equal(wide3, wide) and torch.equal(c3, c_new) and torch.equal(g3, gates)


def test_config4_full_size_and_properties():
    """configs[3]: 36x18 equirectangular heat maps, 30 one-hot channels, ConvLSTM 32/16/8 + Conv2D 512 -> 1024 -> 30 head,
    B = 256, T 10 -> 10 on one GPU.  Seventeen sequences of the full batch against the NumPy oracle, plus size-independent
    properties: every pixel's channel softmax sums to one, and a sequence's result does not depend on its batch-mates."""
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    B, T, H, W, C = 256, 10, 36, 18, 30
    w = O.init_convlstm_seq2seq(1234, C=C, latent_dim=16, k=5, head="conv2d")
    assert w["enc0_R"].shape[2] == 32 and w["enc2_R"].shape[2] == 8 and w["head1_W"].shape[2:] == (512, 1024)
    rng = np.random.default_rng(1234)
    x = np.zeros((B, T, H, W, C), np.float32)          # one-hot of a 10-degree bin per frame, as cfg.use_one_hot builds it
    idx = rng.integers(0, H * W, size=(B, T, C))
    bi, ti, ci = np.meshgrid(np.arange(B), np.arange(T), np.arange(C), indexing="ij")
    x[bi, ti, idx // W, idx % W, ci] = 1.0
    m = ConvLSTMSeq2Seq(w, head="conv2d")
    out = m.predict([x, x[:, -1:]], predict_step=T)
    assert out.shape == (B, T, H, W, C) and np.isfinite(out).all()
    np.testing.assert_allclose(out.sum(-1), 1.0, atol=1e-5)
    rows = list(range(0, B, 17)) + [B - 1]          # 17 sequences spread over the batch (round 3 checked three)
    ref = O.convlstm_seq2seq_forward(x[rows].astype(np.float64), x[rows, -1:].astype(np.float64),
                                     {k: v.astype(np.float64) for k, v in w.items()}, T, head="conv2d")
    err = np.abs(out[rows] - ref)
    print("config4 full size: max abs err %.3e (max |ref| %.3e)" % (err.max(), np.abs(ref).max()))
    assert (err <= 1e-3 * np.abs(ref) + 1e-5).all() and err.max() <= 2e-5
    small = m.predict([x[rows], x[rows, -1:]], predict_step=T)
    np.testing.assert_allclose(small, out[rows], atol=1e-6)


def test_convlstm_mse_with_xyz_sum1_term():
    """costfunc._mse with cfg.add_xyz_sum1 (cost.py:20-29): MSE + 0.5 * MSE(1, ux^2 + uy^2 + uz^2) on the xyz head; loss and
    every gradient against torch.autograd fp64, and the stand-alone kernel on a ragged pixel count."""
    from longterm360fov_amd import ops
    from longterm360fov_amd.training import ConvLSTMTrainer
    head, B, T_in, T_out, H, W, C, L, hf, act = "conv1d", 3, 2, 3, 1, 30, 3, 16, (32, 48), "hard_sigmoid"
    w = O.init_convlstm_seq2seq(6, C=C, latent_dim=L, head=head, head_filters=hf, map_hw=(H, W))
    rng = np.random.default_rng(9)
    enc = rng.random((B, T_in, H, W, C)).astype(np.float32)
    dec0 = enc[:, -1:].copy()
    tgt = rng.random((B, T_out, H, W, C)).astype(np.float32)
    tgt /= tgt.sum(-1, keepdims=True)
    loss_ref, g_ref, _ = _torch_convlstm_graph(enc, dec0, tgt, w, head, act, xyz_sum1=True)
    loss_plain, _, _ = _torch_convlstm_graph(enc, dec0, tgt, w, head, act)
    assert loss_ref > loss_plain * 1.05
    tr = ConvLSTMTrainer(w, head=head, act=act, add_xyz_sum1=True)
    loss, _ = tr.forward_backward(dev(enc), dev(dec0), dev(tgt))
    assert abs(float(loss.item()) - loss_ref) <= 1e-5 * loss_ref + 1e-9
    assert abs(float(tr.eval_loss(dev(enc), dev(dec0), dev(tgt)).item()) - loss_ref) <= 1e-5 * loss_ref + 1e-9
    for k in tr.order:
        a = tr.g[k].detach().cpu().numpy().astype(np.float64)
        scale = np.abs(g_ref[k]).max()
        assert np.abs(a - g_ref[k]).max() <= 2e-4 * scale + 1e-9, k
    p = rng.standard_normal((1001, 5)).astype(np.float32)
    tp = torch.tensor(p.astype(np.float64), requires_grad=True)
    reg_ref = 0.5 * torch.mean((tp[:, 0] ** 2 + tp[:, 1] ** 2 + tp[:, 2] ** 2 - 1) ** 2)
    reg_ref.backward()
    dp = torch.full((1001, 5), 0.25, dtype=torch.float32, device="cuda")
    reg = ops.xyz_sum1_grad(dev(p), dp)
    assert abs(float(reg.item()) - float(reg_ref)) <= 1e-5 * float(reg_ref)
    close(dp - 0.25, tp.grad.numpy(), "xyz_sum1 gradient", tol=1e-5)


def test_convlstm_heatmap_fork_categorical_crossentropy():
    """mycode/convlstm_heatmap.py:192 compiles the same graph with loss='categorical_crossentropy', optimizer='adam': loss and
    every gradient against torch.autograd fp64; the stand-alone kernel incl. its clip (a zero probability) and its
    renormalisation (rows that do not sum to one); model-object surface."""
    from longterm360fov_amd import ops
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    from longterm360fov_amd.training import ConvLSTMTrainer
    head, B, T_in, T_out, H, W, C, L, hf, act = "conv2d", 2, 3, 3, 9, 6, 10, 8, (24, 40), "hard_sigmoid"
    w = O.init_convlstm_seq2seq(5, C=C, latent_dim=L, head=head, head_filters=hf, map_hw=(H, W))
    rng = np.random.default_rng(12)
    enc = rng.random((B, T_in, H, W, C)).astype(np.float32)
    dec0 = enc[:, -1:].copy()
    tgt = np.zeros((B, T_out, H, W, C), np.float32)      # one-hot targets
    np.put_along_axis(tgt, rng.integers(0, C, (B, T_out, H, W, 1)), 1.0, -1)
    loss_ref, g_ref, _ = _torch_convlstm_graph(enc, dec0, tgt, w, head, act, xent=True)
    tr = ConvLSTMTrainer(w, head=head, act=act, optimizer="adam", loss="categorical_crossentropy")
    loss, _ = tr.forward_backward(dev(enc), dev(dec0), dev(tgt))
    assert abs(float(loss.item()) - loss_ref) <= 1e-5 * loss_ref
    assert abs(float(tr.eval_loss(dev(enc), dev(dec0), dev(tgt)).item()) - loss_ref) <= 1e-5 * loss_ref
    for k in tr.order:
        a = tr.g[k].detach().cpu().numpy().astype(np.float64)
        scale = np.abs(g_ref[k]).max()
        assert np.abs(a - g_ref[k]).max() <= 2e-4 * scale + 1e-9, k
    p = rng.random((777, 30)).astype(np.float32) + 0.01
    p[5, 3] = 0.0                                          # clipped: contributes log(1e-7), no gradient
    t = np.zeros_like(p)
    t[np.arange(777), rng.integers(0, 30, 777)] = 1.0
    t[5] = 0.0; t[5, 3] = 1.0
    tp = torch.tensor(p.astype(np.float64), requires_grad=True)
    lr = torch.mean(-(torch.tensor(t.astype(np.float64)) * torch.log(torch.clamp(tp / tp.sum(-1, keepdim=True), 1e-7, 1 - 1e-7))).sum(-1))
    lr.backward()
    dp, l = ops.categorical_crossentropy_grad(dev(p), dev(t))
    assert abs(float(l.item()) - float(lr)) <= 1e-5 * float(lr)
    close(dp, tp.grad.numpy(), "categorical crossentropy gradient", tol=2e-5)
    m = ConvLSTMSeq2Seq(w, head=head)
    m.compile(loss="categorical_crossentropy", optimizer="adam", metrics=["accuracy"])
    losses = [m.train_on_batch([enc, dec0], tgt) for _ in range(4)]
    assert losses[-1] < losses[0]


@pytest.mark.parametrize("B,H,W,C,N,kh,kw", [(3, 36, 18, 56, 512, 5, 5), (2, 36, 18, 512, 200, 5, 5), (2, 36, 18, 128, 30, 5, 5),
                                             (2, 18, 36, 64, 56, 3, 3), (5, 36, 18, 33, 130, 5, 3), (1, 27, 9, 48, 64, 5, 5)])
def test_conv2d_map_resident_form(B, H, W, C, N, kh, kw):
    """conv_patch.hip (round 4): the prediction head's convolutions with the whole input map resident in LDS - all three wave
    arrangements (128 / 64 / 32 output channels per workgroup), channel counts that are no multiple of 16 or of 4, output widths that
    are no multiple of the block, 3 x 3 and 5 x 3 kernels, strided inputs - against the NumPy oracle, and against the tap-gathering
    implicit GEMM (FOV_NO_CONV_PATCH=1), which sums in another order."""
    from longterm360fov_amd import ops, _lib
    rng = np.random.default_rng(B * 1000 + C + N)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, C, N)) / np.sqrt(kh * kw * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((B, H, W, N)).astype(np.float32)
    ref = O.conv2d_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64))
    got = ops.conv2d(dev(x), dev(w), dev(b))
    close(got, ref, "map-resident conv2d")
    close(ops.conv2d(dev(x), dev(w), dev(b), add=dev(add), activation="relu"), np.maximum(ref + add, 0), "map-resident conv2d+add+relu")
    close(ops.conv2d(dev(x), dev(w)), ref - b, "map-resident conv2d, no bias")
    if C % 4 == 0:
        wide = rng.standard_normal((B, H, W, C + 12)).astype(np.float32)
        ref2 = O.conv2d_same(wide[..., 8:8 + C].astype(np.float64), w.astype(np.float64))
        close(ops.conv2d(dev(wide)[..., 8:8 + C], dev(w)), ref2, "map-resident conv2d, channel slice of a wider map")
    seq = rng.standard_normal((B, 2, H, W, C)).astype(np.float32)
    ref3 = O.conv2d_same(seq[:, 1].astype(np.float64), w.astype(np.float64))
    close(ops.conv2d(dev(seq)[:, 1], dev(w)), ref3, "map-resident conv2d, batch-strided input")
    try:
        os.environ["FOV_NO_CONV_PATCH"] = "1"
        _lib.lib().fov_reload_env()
        old = ops.conv2d(dev(x), dev(w), dev(b))
    finally:
        os.environ.pop("FOV_NO_CONV_PATCH", None)
        _lib.lib().fov_reload_env()
    d = (got - old).abs().max().item()
    assert d <= 2e-5 * float(np.abs(ref).max()), d
    assert d > 0 or C % 4 != 0        # two kernels really ran (an input whose pixels are not 16-byte aligned stays on the old one)


# ---------------------------------------------------------------------------------------
# cfg.dilation_rate (config.py:105 -> the six ConvLSTM2D layers, convlstm_seq2seq.py:102,110,120,148,155,162)
# ---------------------------------------------------------------------------------------
@pytest.mark.parametrize("B,H,W,C,N,kh,kw,d", [(2, 9, 7, 3, 7, 5, 5, 2), (3, 36, 18, 32, 128, 5, 5, 2), (2, 1, 30, 3, 128, 1, 7, 3),
                                               (1, 6, 7, 17, 33, 3, 3, 2), (2, 36, 18, 56, 64, 5, 5, 3), (2, 4, 4, 8, 16, 5, 5, 4),
                                               (1, 36, 18, 30, 30, 5, 5, 9)])
def test_dilated_conv2d_forward_and_backward_kernels(B, H, W, C, N, kh, kw, d):
    """Dilated 'same' convolution, its data gradient (the dilated convolution with the transposed weights) and its weight
    gradient, against the oracle (forward) and torch.autograd in fp64; taps that fall wholly outside the map (d*(k//2) >= H)
    contribute zero."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(B * 1000 + C * 10 + N + d)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    w = (rng.standard_normal((kh, kw, C, N)) / np.sqrt(kh * kw * C)).astype(np.float32)
    b = rng.standard_normal(N).astype(np.float32)
    add = rng.standard_normal((B, H, W, N)).astype(np.float32)
    dy = rng.standard_normal((B, H, W, N)).astype(np.float32)
    ref = O.conv2d_same(x.astype(np.float64), w.astype(np.float64), b.astype(np.float64), dilation=d)
    close(ops.conv2d(dev(x), dev(w), dev(b), dilation=d), ref, "dilated conv")
    close(ops.conv2d(dev(x), dev(w), dev(b), add=dev(add), activation="relu", dilation=d), np.maximum(ref + add, 0), "dilated conv + add, relu")
    wide = dev(np.concatenate([rng.standard_normal((B, H, W, 4)).astype(np.float32), x, np.zeros((B, H, W, 4), np.float32)], -1))
    close(ops.conv2d(wide[..., 4:4 + C], dev(w), dev(b), dilation=d), ref, "dilated conv on a channel slice")
    tx = torch.tensor(x.astype(np.float64), requires_grad=True)
    tw = torch.tensor(w.astype(np.float64), requires_grad=True)
    out = _tconv(tx, tw, d)
    assert np.abs(out.detach().numpy() + b - ref).max() < 1e-10          # the oracle and torch agree on what dilation means
    (out * torch.tensor(dy.astype(np.float64))).sum().backward()
    dw = ops.conv2d_wgrad(dev(x), dev(dy), kh, kw, dilation=d)
    close(dw, tw.grad.numpy(), "dilated wgrad")
    ops.conv2d_wgrad(wide[..., 4:4 + C], dev(dy), kh, kw, dw=dw, accumulate=True, dilation=d)
    close(dw, 2 * tw.grad.numpy(), "dilated wgrad accumulate, channel slice")
    close(ops.conv2d(dev(dy), ops.conv2d_weight_transpose(dev(w)), dilation=d), tx.grad.numpy(), "dilated dgrad")


@pytest.mark.parametrize("act", ["hard_sigmoid", "sigmoid"])
@pytest.mark.parametrize("B,H,W,C,F,k,d", [(3, 36, 18, 32, 32, 5, 2), (2, 6, 5, 30, 32, 5, 2), (2, 1, 30, 3, 8, 5, 3), (1, 7, 4, 17, 20, 3, 2),
                                           (2, 36, 18, 128, 64, 5, 2)])
def test_dilated_convlstm_cell_one_launch(B, H, W, C, F, k, d, act):
    """Keras's ConvLSTM2DCell dilates input_conv only: in the one-launch cell the taps over x are spread, those over h_prev
    are not; with and without a previous state, gates tape included."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(C + F + d)
    f64 = lambda a: a.astype(np.float64)
    x = rng.standard_normal((B, H, W, C)).astype(np.float32)
    h = rng.standard_normal((B, H, W, F)).astype(np.float32) * 0.5
    c = rng.standard_normal((B, H, W, F)).astype(np.float32)
    K = (rng.standard_normal((k, k, C, 4 * F)) / np.sqrt(k * k * C)).astype(np.float32)
    R = (rng.standard_normal((k, k, F, 4 * F)) / np.sqrt(k * k * F)).astype(np.float32)
    b = rng.standard_normal(4 * F).astype(np.float32) * 0.1
    h_ref, c_ref = O.convlstm2d_step(f64(x), f64(h), f64(c), f64(K), f64(R), f64(b), act, dilation=d)
    h_plain, _ = O.convlstm2d_step(f64(x), f64(h), f64(c), f64(K), f64(R), f64(b), act)
    assert np.abs(h_ref - h_plain).max() > 1e-3        # the case distinguishes dilation d from 1
    KR = dev(np.concatenate([K, R], 2))
    feat = torch.zeros((B, H, W, F + 8), dtype=torch.float32, device="cuda")
    gates = torch.empty((B, H, W, 4 * F), dtype=torch.float32, device="cuda")
    c_new = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
    ops.convlstm_cell(dev(x), dev(h), KR, dev(b), dev(c), feat[..., 4:4 + F], act, c_new=c_new, gates=gates, dilation=d)
    close(feat[..., 4:4 + F], h_ref, "dilated cell h", tol=5e-5)
    close(c_new, c_ref, "dilated cell c", tol=5e-5)
    s = (lambda v: 1 / (1 + np.exp(-v))) if act == "sigmoid" else (lambda v: np.clip(0.2 * v + 0.5, 0, 1))
    z = O.conv2d_same(f64(x), f64(K), f64(b), dilation=d) + O.conv2d_same(f64(h), f64(R))
    g_ref = np.concatenate([s(z[..., :F]), s(z[..., F:2 * F]), np.tanh(z[..., 2 * F:3 * F]), s(z[..., 3 * F:])], -1)
    close(gates, g_ref, "dilated cell gates tape", tol=5e-5)
    # zero initial state: w is K alone
    h0_ref, c0_ref = O.convlstm2d_step(f64(x), np.zeros_like(f64(h)), np.zeros_like(f64(c)), f64(K), f64(R), f64(b), act, dilation=d)
    h0 = torch.empty((B, H, W, F), dtype=torch.float32, device="cuda")
    _, c0 = ops.convlstm_cell(dev(x), None, dev(K), dev(b), None, h0, act, dilation=d)
    close(h0, h0_ref, "dilated cell h, zero state", tol=5e-5)
    close(c0, c0_ref, "dilated cell c, zero state", tol=5e-5)


@pytest.mark.parametrize("head,B,T_in,T_out,H,W,C,L,hf,act,d,drop", [("conv2d", 2, 3, 3, 9, 6, 10, 8, (24, 40), "hard_sigmoid", 2, False),
                                                                      ("conv1d", 3, 2, 3, 1, 30, 3, 16, (32, 48), "sigmoid", 3, False),
                                                                      ("conv2d", 1, 2, 2, 36, 18, 30, 16, (40, 136), "sigmoid", 2, False),
                                                                      ("conv2d", 2, 3, 2, 7, 5, 6, 8, (12, 20), "hard_sigmoid", 2, True)])
def test_dilated_convlstm_seq2seq_predict_gradients_and_training(head, B, T_in, T_out, H, W, C, L, hf, act, d, drop):
    """The whole model at cfg.dilation_rate = d: predict against the oracle, the unrolled training graph's gradients against
    torch.autograd in fp64 (with Keras's per-gate input dropout in the last case), and RMSprop steps reduce the loss."""
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    from longterm360fov_amd.training import ConvLSTMTrainer
    w = O.init_convlstm_seq2seq(5, C=C, latent_dim=L, head=head, head_filters=hf, map_hw=(H, W))
    rng = np.random.default_rng(8 + d)
    for k in w:
        if k.endswith("_b"):
            w[k] = (w[k] + 0.1 * rng.standard_normal(w[k].shape)).astype(np.float32)
    enc = rng.random((B, T_in, H, W, C)).astype(np.float32)
    dec0 = enc[:, -1:].copy()
    tgt = rng.random((B, T_out, H, W, C)).astype(np.float32)
    tgt /= tgt.sum(-1, keepdims=True)
    w64 = {k: v.astype(np.float64) for k, v in w.items()}
    P_orc = O.convlstm_seq2seq_forward(enc.astype(np.float64), dec0.astype(np.float64), w64, T_out, head=head, act=act, dilation=d)
    P_one = O.convlstm_seq2seq_forward(enc.astype(np.float64), dec0.astype(np.float64), w64, T_out, head=head, act=act)
    assert np.abs(P_orc - P_one).max() > 1e-5
    close(ConvLSTMSeq2Seq(w, head=head, recurrent_activation=act, dilation_rate=d).predict([enc, dec0], predict_step=T_out), P_orc,
          "dilated predict", tol=5e-5)
    tr = ConvLSTMTrainer(w, head=head, act=act, dilation_rate=d, dropout_rate=0.3 if drop else 0.0, seed=5)
    masks = tr.sample_masks(B, H, W, C, T_out) if drop else None
    loss_ref, g_ref, P_ref = _torch_convlstm_graph(enc, dec0, tgt, w, head, act, dilation=d,
                                                   masks=None if masks is None else {k: v.cpu().numpy() for k, v in masks.items()})
    if not drop:
        assert np.abs(P_orc - P_ref).max() < 1e-10
    loss, P = tr.forward_backward(dev(enc), dev(dec0), dev(tgt), masks=masks)
    close(P, P_ref, "dilated train-mode forward")
    assert abs(float(loss.item()) - loss_ref) <= 1e-5 * loss_ref + 1e-9
    for k in tr.order:
        a = tr.g[k].detach().cpu().numpy().astype(np.float64)
        scale = np.abs(g_ref[k]).max()
        err = np.abs(a - g_ref[k]).max()
        assert err <= 2e-4 * scale + 1e-9, (k, err, scale)
    losses = [float(tr.train_step(dev(enc), dev(dec0), dev(tgt)).item()) for _ in range(4)]
    assert losses[-1] < losses[0]


def test_dilation_rate_defaults_to_cfg_and_rejects_zero():
    from longterm360fov_amd import ops
    from longterm360fov_amd.config import cfg
    from longterm360fov_amd.models import ConvLSTMSeq2Seq
    from longterm360fov_amd._lib import FovError
    w = O.init_convlstm_seq2seq(9, C=3, latent_dim=8, head="conv1d", head_filters=(16, 24))
    assert ConvLSTMSeq2Seq(w, head="conv1d").dilation_rate == cfg.dilation_rate == 1
    with pytest.raises(ValueError):
        ConvLSTMSeq2Seq(w, head="conv1d", dilation_rate=0)
    x = torch.zeros((1, 4, 4, 4), device="cuda")
    with pytest.raises(FovError):
        ops.conv2d(x, torch.zeros((3, 3, 4, 4), device="cuda"), dilation=0)
