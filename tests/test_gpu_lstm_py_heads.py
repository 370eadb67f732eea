"""GPU parity of the two branches of mycode/lstm.py the committed mycode/config.py:69,71 leaves open when
cfg.predict_mean_var is False: the mixture-density head (_GMM_3dgassian, lstm.py:377-400) with
costfunc.mixture_3d_gaussian_loss (cost.py:486-549) - the one the config SELECTS - and the raw head
(pred_cnn_model_fn, lstm.py:147-174) with MSE / pred_raw_loss_tf (cost.py:634-641).  Through the C ABI
(fov_mlp_head_fwd / _bwd, fov_gmm3d_loss_grad, fov_gmm3d_sample) against the NumPy fp64 oracle and independent
torch.autograd fp64 graphs (torch.linalg.eigvalsh + torch.distributions.MultivariateNormal for the density).
Tolerances: outputs 1e-3 relative (north_star) with the stated absolute floors; gradients 2e-4 of each tensor's max."""
import numpy as np
import pytest
import torch

from oracle import fov_oracle as O

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()


def _gmm_head(rng, H, n=20, rho_bias=0.0):
    dims = [H, 64, 128, 256, 10 * n]
    head = {}
    for l in range(4):
        head["fc%d_W" % (l + 1)] = (rng.standard_normal((dims[l], dims[l + 1])) / np.sqrt(dims[l])).astype(np.float32)
        head["fc%d_b" % (l + 1)] = (0.1 * rng.standard_normal(dims[l + 1])).astype(np.float32)
    # large correlations make cost.py:335-348's repair branch common (a 3x3 "correlation" matrix of independent tanh()s
    # is often indefinite); log-sigmas near -1 keep the densities at a useful scale
    head["fc4_b"][7 * n:] += rho_bias * rng.choice([-1.0, 1.0], 3 * n).astype(np.float32)
    head["fc4_b"][4 * n:7 * n] -= 1.0
    return head


def _raw_head(rng, H, fps=30):
    dims = [H, 128, 256, 3 * fps]
    head = {}
    for l in range(3):
        head["conv%d_W" % (l + 1)] = (rng.standard_normal((5, dims[l], dims[l + 1])) / np.sqrt(dims[l])).astype(np.float32)
        head["conv%d_b" % (l + 1)] = (0.1 * rng.standard_normal(dims[l + 1])).astype(np.float32)
    return head


def _cells(rng, F, H):
    cells = []
    for l in range(2):
        Fin = F if l == 0 else H
        cells.append(((rng.standard_normal((Fin + H, 4 * H)) / np.sqrt(Fin + H)).astype(np.float32),
                      (0.1 * rng.standard_normal(4 * H)).astype(np.float32)))
    return cells


def _torch_gmm3d_loss(pred, pts, n, scale, weight_by_pi):
    """pred (B,10n) PRE-activations, pts (B,P,3): lstm.py:386-399 + cost.py:486-549 on torch fp64, independent of the oracle."""
    B = pred.shape[0]
    pi = torch.exp(pred[:, :n])
    pi = pi / pi.sum(1, keepdim=True)
    mu = pred[:, n:4 * n].reshape(B, n, 3)
    s1, s2, s3 = torch.exp(pred[:, 4 * n:7 * n]).reshape(B, n, 3).unbind(-1)
    r12, r13, r23 = torch.tanh(pred[:, 7 * n:]).reshape(B, n, 3).unbind(-1)
    S = torch.stack([torch.stack([s1 * s1, r12 * s1 * s2, r13 * s1 * s3], -1),
                     torch.stack([r12 * s1 * s2, s2 * s2, r23 * s2 * s3], -1),
                     torch.stack([r13 * s1 * s3, r23 * s2 * s3, s3 * s3], -1)], -2)
    lam = torch.linalg.eigvalsh(S)[..., 0]
    shift = torch.where(lam < 0, -10.0 * lam, torch.zeros_like(lam))
    S2 = S + shift[..., None, None] * torch.eye(3, dtype=pred.dtype)
    dist = torch.distributions.MultivariateNormal(mu[:, :, None, :], covariance_matrix=S2[:, :, None])
    p = dist.log_prob(pts[:, None]).exp()
    if weight_by_pi:
        p = p * pi[:, :, None]
    return -(torch.log(p.sum(1) + 1e-20)).sum() * scale, int((lam < 0).sum())


@pytest.mark.parametrize("B,n,P,ldpad,weight_by_pi,rho_bias", [
    (32, 20, 30, 270, False, 0.0), (32, 20, 30, 0, False, 2.0), (7, 20, 30, 90, True, 1.5), (5, 3, 10, 0, False, 2.5),
    (1, 32, 256, 0, True, 1.0),
])
def test_gmm3d_loss_and_gradient(B, n, P, ldpad, weight_by_pi, rho_bias):
    """fov_gmm3d_loss_grad vs torch.autograd fp64 (and the NumPy oracle's loss): the script's shape (32 rows, 20 mixtures,
    second 0 of ten), rows dominated by the eigenvalue repair, the textbook pi-weighted mixture, per-frame scoring, the caps."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(100 * B + n + P)
    pre = (0.6 * rng.standard_normal((B, 10 * n))).astype(np.float32)
    pre[:, 4 * n:7 * n] -= 0.7
    pre[:, 7 * n:] += rho_bias * rng.choice([-1.0, 1.0], (B, 3 * n)).astype(np.float32)
    y = rng.uniform(-1, 1, (B, 3 * P + ldpad)).astype(np.float32)
    scale = 1.0 / (B * 10 * 30)
    tp = torch.tensor(pre.astype(np.float64), requires_grad=True)
    pts = torch.tensor(y[:, :3 * P].astype(np.float64)).reshape(B, P, 3)
    loss_ref, n_rep = _torch_gmm3d_loss(tp, pts, n, scale, weight_by_pi)
    loss_ref.backward()
    g_ref = tp.grad.numpy()
    if rho_bias >= 2.0:
        assert n_rep > B * n // 4          # the repair branch is what this case is about
    # activations as the head's kernel leaves them (fp64 here: the loss kernel is what is under test)
    e = np.exp(pre[:, :n].astype(np.float64))
    params = np.concatenate([e / e.sum(1, keepdims=True), pre[:, n:4 * n], np.exp(pre[:, 4 * n:7 * n].astype(np.float64)),
                             np.tanh(pre[:, 7 * n:].astype(np.float64))], 1).astype(np.float32)
    p64 = params.astype(np.float64)
    loss_or = O.mixture_3d_gaussian_loss(y[:, None, :3 * P].astype(np.float64), (p64[:, :n], p64[:, n:4 * n], p64[:, 4 * n:7 * n], p64[:, 7 * n:]),
                                         B, 10, 30, process_in_seconds=True, weight_by_pi=weight_by_pi)
    loss, dpre = ops.gmm3d_loss_grad(dev(params), dev(y), P, scale, weight_by_pi=weight_by_pi)
    lv = float(loss.item())
    print("gmm3d loss B%d n%d P%d: gpu %.6f torch %.6f oracle %.6f, %d of %d covariances repaired" % (B, n, P, lv, float(loss_ref.detach()), loss_or, n_rep, B * n))
    assert abs(lv - loss_or) <= 2e-5 * abs(loss_or) + 1e-7
    assert abs(lv - float(loss_ref)) <= 1e-4 * abs(float(loss_ref)) + 1e-7     # fp32 parameters vs fp64 pre-activations
    g = dpre.cpu().numpy().astype(np.float64)
    assert np.isfinite(g).all()
    if not weight_by_pi:
        assert np.all(g[:, :n] == 0.0)      # cost.py:532-538 drops mixture_pi: no gradient reaches the softmax
    err = np.abs(g - g_ref)
    print("   dpre max|ref| %.3e max err %.3e" % (np.abs(g_ref).max(), err.max()))
    assert err.max() <= 2e-4 * np.abs(g_ref).max() + 1e-9
    assert (err <= 2e-3 * np.abs(g_ref) + 2e-4 * np.abs(g_ref).max()).all()


def _torch_mlp(h, head, kind, masks=None):
    """-> PRE-activation of the last layer (the mixture split / the tanh are applied by the caller)."""
    if kind == "gmm":
        a = h
        for l in (1, 2, 3):
            a = torch.relu(a @ head["fc%d_W" % l] + head["fc%d_b" % l])
            if masks is not None and l <= 2 and masks[l - 1] is not None:
                a = a * torch.tensor(masks[l - 1].astype(np.float64))
        return a @ head["fc4_W"] + head["fc4_b"]
    a = torch.relu(h @ head["conv1_W"][2] + head["conv1_b"])
    a = torch.relu(a @ head["conv2_W"][2] + head["conv2_b"])
    return a @ head["conv3_W"][2] + head["conv3_b"]


@pytest.mark.parametrize("kind,B,H,with_masks", [("gmm", 32, 512, False), ("gmm", 5, 400, True), ("gmm", 70, 40, True), ("raw", 32, 512, False),
                                                 ("raw", 3, 600, False), ("gmm", 1, 2048, False)])
def test_fused_head_chain_forward_backward(kind, B, H, with_masks):
    """fov_mlp_head_fwd / _bwd (one launch forward, two backward) on both heads vs torch.autograd fp64: every layer's output,
    every weight / bias gradient, dh; accumulate adds."""
    from longterm360fov_amd import ops
    rng = np.random.default_rng(B + H)
    head = _gmm_head(rng, H) if kind == "gmm" else _raw_head(rng, H)
    h = rng.standard_normal((B, H)).astype(np.float32)
    masks = None
    if with_masks:
        masks = [((rng.random((B, 64)) < 0.8) / 0.8).astype(np.float32), ((rng.random((B, 128)) < 0.8) / 0.8).astype(np.float32)]
    th = {k: torch.tensor(v.astype(np.float64), requires_grad=True) for k, v in head.items()}
    hh = torch.tensor(h.astype(np.float64), requires_grad=True)
    pre_ref = _torch_mlp(hh, th, kind, masks)
    dlast = rng.standard_normal(tuple(pre_ref.shape)).astype(np.float32)       # gradient at the last layer's PRE-activation
    (pre_ref * torch.tensor(dlast.astype(np.float64))).sum().backward()
    dw = {k: dev(v) for k, v in head.items()}
    if kind == "gmm":
        layers = [(dw["fc%d_W" % l], dw["fc%d_b" % l], "relu" if l < 4 else None) for l in (1, 2, 3, 4)]
    else:
        layers = [(dw["conv%d_W" % l][2], dw["conv%d_b" % l], "relu" if l < 3 else "tanh") for l in (1, 2, 3)]
    dm = None if masks is None else [dev(masks[0]), dev(masks[1]), None, None]
    acts = ops.mlp_head_fwd(dev(h), layers, masks=dm, n_mix=20 if kind == "gmm" else 0)
    out = acts[-1].cpu().numpy().astype(np.float64)
    if kind == "gmm":
        (pi, us, sg, rh), (a1, a2, a3) = O.tf_gmm3d_head(h.astype(np.float64), {k: v.astype(np.float64) for k, v in head.items()},
                                                          masks=None if masks is None else [m.astype(np.float64) for m in masks])
        ref = np.concatenate([pi, us, sg, rh], 1)
        np.testing.assert_allclose(acts[0].cpu().numpy(), a1, atol=2e-5)
        np.testing.assert_allclose(acts[2].cpu().numpy(), a3, atol=5e-5)
        assert abs(out[:, :20].sum(1) - 1).max() < 1e-5
    else:
        ref = O.tf_raw_head(h.astype(np.float64), {k: v.astype(np.float64) for k, v in head.items()})[0][:, 0]
        np.testing.assert_allclose(out, np.tanh(pre_ref.detach().numpy()), atol=2e-5)
    assert (np.abs(out - ref) <= 1e-3 * np.abs(ref) + 2e-5).all(), np.abs(out - ref).max()
    gW = [torch.full_like(W, 0.25) for W, _, _ in layers]
    gb = [torch.full_like(b, -0.5) for _, b, _ in layers]
    dx = ops.mlp_head_bwd(dev(h), layers, acts, dev(dlast), gW, gb, masks=dm, need_dx=True, accumulate=True)
    names = [("fc%d_W" % l, "fc%d_b" % l) for l in (1, 2, 3, 4)] if kind == "gmm" else [("conv%d_W" % l, "conv%d_b" % l) for l in (1, 2, 3)]
    for l, (kw, kb) in enumerate(names):
        rw = th[kw].grad.numpy() if kind == "gmm" else th[kw].grad.numpy()[2]
        rb = th[kb].grad.numpy()
        ew = np.abs(gW[l].cpu().numpy() - 0.25 - rw).max()
        eb = np.abs(gb[l].cpu().numpy() + 0.5 - rb).max()
        print("%s B%d H%d layer %d: gW max|ref| %.3e err %.3e  gb err %.3e" % (kind, B, H, l, np.abs(rw).max(), ew, eb))
        assert ew <= 2e-4 * np.abs(rw).max() + 1e-6 and eb <= 2e-4 * np.abs(rb).max() + 1e-6
    if kind == "raw":      # the four off-centre taps of every kernel never meet data: zero gradient
        for kw, _ in names:
            g = th[kw].grad.numpy()
            assert np.abs(g[[0, 1, 3, 4]]).max() == 0.0
    rdx = hh.grad.numpy()
    assert np.abs(dx.cpu().numpy() - rdx).max() <= 2e-4 * np.abs(rdx).max() + 1e-7
    # overwrite mode, no dx
    gW2 = [torch.full_like(W, 7.0) for W, _, _ in layers]
    gb2 = [torch.full_like(b, 7.0) for _, b, _ in layers]
    assert ops.mlp_head_bwd(dev(h), layers, acts, dev(dlast), gW2, gb2, masks=dm, need_dx=False, accumulate=False) is None
    for l in range(len(layers)):
        np.testing.assert_allclose(gW2[l].cpu().numpy(), gW[l].cpu().numpy() - 0.25, atol=1e-5 * max(1.0, float(gW2[l].abs().max())))
        np.testing.assert_allclose(gb2[l].cpu().numpy(), gb[l].cpu().numpy() + 0.5, atol=1e-5 * max(1.0, float(gb2[l].abs().max())))


def _torch_stack(x, cw, st, masks, forget_bias=1.0):
    """tf.contrib LSTMCell stack under dynamic_rnn from a fed state (lstm.py:218-240) -> top layer's final h."""
    inp = x
    for l, (W, b) in enumerate(cw):
        c, h = st[l, 0], st[l, 1]
        H = h.shape[1]
        outs = []
        for tt in range(inp.shape[1]):
            z = torch.cat([inp[:, tt], h], 1) @ W + b
            i, j, f, o = z[:, :H], z[:, H:2 * H], z[:, 2 * H:3 * H], z[:, 3 * H:]
            c = torch.sigmoid(f + forget_bias) * c + torch.sigmoid(i) * torch.tanh(j)
            h = torch.sigmoid(o) * torch.tanh(c)
            outs.append(h)
        hs = torch.stack(outs, 1)
        inp = hs if (masks is None or l == len(cw) - 1) else hs * torch.tensor(masks[l].astype(np.float64))
    return h


def _check_stack_grads(tr, cg, hg, H, head_names, raw=False):
    tg = tr.grads_numpy()
    assert tr.padded_slices_are_zero()
    for k in head_names:
        r = hg[k]
        print("head grad %-8s max|ref| %.3e err %.3e" % (k, np.abs(r).max(), np.abs(tg[k] - r).max()))
        assert tg[k].shape == r.shape
        assert np.abs(tg[k] - r).max() <= 2e-4 * np.abs(r).max() + 1e-9, k
    for l in range(2):
        K, R, b = (tg["%s%d" % (n, l)] for n in ("K", "R", "b"))
        perm = np.concatenate([np.arange(0, H), np.arange(2 * H, 3 * H), np.arange(H, 2 * H), np.arange(3 * H, 4 * H)])
        Wg = np.empty_like(cg[l][0]); Wg[:, perm] = np.concatenate([K, R], 0)
        bg = np.empty_like(cg[l][1]); bg[perm] = b
        print("cell %d grad max|ref| %.3e err %.3e" % (l, np.abs(cg[l][0]).max(), np.abs(Wg - cg[l][0]).max()))
        assert np.abs(Wg - cg[l][0]).max() <= 2e-4 * np.abs(cg[l][0]).max() + 1e-9, ("W", l)
        assert np.abs(bg - cg[l][1]).max() <= 2e-4 * np.abs(cg[l][1]).max() + 1e-9, ("b", l)


@pytest.mark.parametrize("H,B,T,seconds,with_masks", [(400, 32, 10, True, False), (40, 9, 4, True, True), (64, 6, 5, False, False)])
def test_lstm_py_gmm_training_graph(H, B, T, seconds, with_masks):
    """The graph the committed config trains (lstm.py:482-485 at n_hidden 400, batch 32, ten seconds in, y = ten seconds of which
    cost.py:502-505 scores second 0): loss, mixture parameters and every gradient vs torch.autograd fp64; then RMSProp steps."""
    from longterm360fov_amd.training import TFLSTMTrainer
    from longterm360fov_amd.config import default_config
    rng = np.random.default_rng(3 * H + B)
    fps = 30
    F = 3 * fps if seconds else 3
    cells, head = _cells(rng, F, H), _gmm_head(rng, H, rho_bias=1.0)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    y = rng.uniform(-1, 1, (B, 10, F)).astype(np.float32)
    init = (0.2 * rng.standard_normal((2, 2, B, H))).astype(np.float32)
    masks = head_masks = None
    if with_masks:
        masks = [((rng.random((B, T, H)) < 0.9) / 0.9).astype(np.float32), None]
        head_masks = [((rng.random((B, 64)) < 0.8) / 0.8).astype(np.float32), ((rng.random((B, 128)) < 0.8) / 0.8).astype(np.float32)]
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    cw = [(t(W), t(b)) for W, b in cells]
    hw = {k: t(v) for k, v in head.items()}
    hT = _torch_stack(torch.tensor(x.astype(np.float64)), cw, torch.tensor(init.astype(np.float64)), masks)
    pre = _torch_mlp(hT, hw, "gmm", head_masks)
    pts = torch.tensor(y.astype(np.float64))
    pts = pts[:, 0].reshape(B, fps, 3) if seconds else pts
    scale = 1.0 / (B * 10 * (fps if seconds else 1))
    loss_ref, n_rep = _torch_gmm3d_loss(pre, pts, 20, scale, False)
    loss_ref.backward()
    cfg = default_config()
    cfg.process_in_seconds = seconds
    cfg.batch_size = B
    assert TFLSTMTrainer.head_kind_of(cfg) == "gmm"        # what mycode/config.py:69,71 selects
    tr = TFLSTMTrainer.from_cfg(cfg, cells, head, lr=1e-3)
    assert tr.head_kind == "gmm" and tr.clip == 1.0
    dm = None if masks is None else [dev(masks[0]), None]
    dhm = None if head_masks is None else [dev(head_masks[0]), dev(head_masks[1]), None, None]
    loss, params, _, state = tr.forward_backward(dev(x), dev(y), dev(init), masks=dm, head_masks=dhm)
    print("gmm graph H%d B%d: loss gpu %.6f ref %.6f (%d repaired)" % (H, B, float(loss.item()), float(loss_ref), n_rep))
    assert abs(float(loss.item()) - float(loss_ref)) <= 2e-4 * abs(float(loss_ref)) + 1e-7
    assert state.shape == (2, 2, B, H) and params.shape == (B, 200)
    pr = pre.detach().numpy()
    e = np.exp(pr[:, :20])
    ref = np.concatenate([e / e.sum(1, keepdims=True), pr[:, 20:80], np.exp(pr[:, 80:140]), np.tanh(pr[:, 140:])], 1)
    got = params.cpu().numpy()
    assert (np.abs(got - ref) <= 1e-3 * np.abs(ref) + 5e-5).all(), np.abs(got - ref).max()
    _check_stack_grads(tr, [(W.grad.numpy(), b.grad.numpy()) for W, b in cw], {k: v.grad.numpy() for k, v in hw.items()}, H,
                       TFLSTMTrainer.GMM_HEAD)
    # weights round-trip at the caller's width; a few optimizer steps reduce the loss
    wn = tr.weights_numpy()
    for k in TFLSTMTrainer.GMM_HEAD:
        assert np.array_equal(wn[k], head[k])
    l0 = float(tr.train_step(dev(x), dev(y), dev(init))[0].item())
    for _ in range(5):
        l1 = float(tr.train_step(dev(x), dev(y), dev(init))[0].item())
    assert l1 < l0
    assert tr.padded_slices_are_zero()


@pytest.mark.parametrize("H,B,T,P,use_reg", [(40, 9, 3, 5, False), (400, 12, 4, 3, True), (64, 8, 5, 1, False)])
def test_lstm_py_raw_refeed_training_graph(H, B, T, P, use_reg):
    """The raw branch (lstm.py:486-508): prediction k re-fed as the window's last second for prediction k+1 (P > T: early
    predictions shift out), MSE summed over the P seconds (+ the sum-to-one term), against torch.autograd fp64 and the oracle."""
    from longterm360fov_amd.training import TFLSTMTrainer
    from longterm360fov_amd.config import default_config
    rng = np.random.default_rng(11 * H + B)
    F, fps = 90, 30
    cells, head = _cells(rng, F, H), _raw_head(rng, H)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    y = rng.uniform(-1, 1, (B, P, F)).astype(np.float32)
    init = (0.2 * rng.standard_normal((2, 2, B, H))).astype(np.float32)
    t = lambda a: torch.tensor(np.asarray(a, np.float64), requires_grad=True)
    cw = [(t(W), t(b)) for W, b in cells]
    hw = {k: t(v) for k, v in head.items()}
    st = torch.tensor(init.astype(np.float64))
    win, yt, loss_ref = torch.tensor(x.astype(np.float64)), torch.tensor(y.astype(np.float64)), 0.0
    for k in range(P):
        if k > 0:
            win = torch.cat([win[:, 1:], pred[:, None, :]], 1)
        pred = torch.tanh(_torch_mlp(_torch_stack(win, cw, st, None), hw, "raw"))
        loss_ref = loss_ref + ((yt[:, k] - pred) ** 2).mean()
        if use_reg:
            p3 = pred.reshape(B, fps, 3)
            loss_ref = loss_ref + 0.1 * (((p3 ** 2).sum(-1) - 1) ** 2).sum()
    loss_ref.backward()
    c64 = [(W.astype(np.float64), b.astype(np.float64)) for W, b in cells]
    loss_or, preds_or = O.tf_lstm_raw_refeed_loss(x.astype(np.float64), y.astype(np.float64), c64, {k: v.astype(np.float64) for k, v in head.items()},
                                                  init.astype(np.float64), use_reg=use_reg)
    assert abs(float(loss_ref) - loss_or) <= 1e-9 * abs(loss_or) + 1e-12
    cfg = default_config()
    cfg.use_GMM = False
    assert TFLSTMTrainer.head_kind_of(cfg) == "raw"
    tr = TFLSTMTrainer.from_cfg(cfg, cells, head, lr=1e-3, use_reg=use_reg)
    loss, pred_gpu, _, state = tr.forward_backward(dev(x), dev(y), dev(init))
    print("raw graph H%d B%d P%d: loss gpu %.6f ref %.6f" % (H, B, P, float(loss.item()), float(loss_ref)))
    assert abs(float(loss.item()) - float(loss_ref)) <= 2e-5 * abs(float(loss_ref)) + 1e-7
    np.testing.assert_allclose(pred_gpu.cpu().numpy(), preds_or[-1][:, 0], atol=3e-5)
    hg = {k: v.grad.numpy() for k, v in hw.items()}
    _check_stack_grads(tr, [(W.grad.numpy(), b.grad.numpy()) for W, b in cw], hg, H, TFLSTMTrainer.RAW_HEAD)
    losses = [float(tr.train_step(dev(x), dev(y), dev(init))[0].item()) for _ in range(6)]
    assert losses[-1] < losses[0]
    # the test-time loop (lstm.py:747-757): state carried, the prediction itself shifted in
    tr2 = TFLSTMTrainer(cells, head, head_kind="raw")
    outs, st2 = tr2.rollout_raw(dev(x), dev(init), 4)
    win, stn = x.astype(np.float64), init.astype(np.float64)
    for k in range(4):
        _, stn = O.tf_dynamic_rnn(win, c64, stn)
        p, _ = O.tf_raw_head(stn[-1, 1], {k2: v.astype(np.float64) for k2, v in head.items()})
        np.testing.assert_allclose(outs[k].cpu().numpy(), p[:, 0], atol=5e-5)
        win = np.concatenate([win[:, 1:], p], 1)
    np.testing.assert_allclose(st2.cpu().numpy(), stn, atol=5e-5)


def test_lstm_py_gmm_sampled_rollout():
    """GMM test loop (lstm.py:690-698,735-745,820-825): one fed second, carried state, one sampled second fed back per step, with
    the uniform / normal draws given explicitly.  Step by step from the oracle's window and state: mixture parameters, then
    fov_gmm3d_sample vs the oracle's inverse-CDF + Cholesky draw (frames whose u sits within 1e-4 of a cumulative weight may
    legitimately pick the neighbouring component in fp32 and are skipped); then the object's own loop end to end."""
    from longterm360fov_amd.training import TFLSTMTrainer
    from longterm360fov_amd import ops
    rng = np.random.default_rng(5)
    H, B, P, fps = 400, 32, 5, 30
    F = 3 * fps
    cells, head = _cells(rng, F, H), _gmm_head(rng, H, rho_bias=1.0)
    x = rng.uniform(-1, 1, (B, 1, F)).astype(np.float32)
    init = (0.2 * rng.standard_normal((2, 2, B, H))).astype(np.float32)
    u = rng.random((P, B, fps)).astype(np.float32)
    z = rng.standard_normal((P, B, fps, 3)).astype(np.float32)
    c64 = [(W.astype(np.float64), b.astype(np.float64)) for W, b in cells]
    h64 = {k: v.astype(np.float64) for k, v in head.items()}
    tr = TFLSTMTrainer(cells, head, head_kind="gmm")
    win, st = x.astype(np.float64), init.astype(np.float64)
    n_skipped = 0
    for k in range(P):
        params_gpu, _, st_gpu = tr.predict(dev(win), dev(st))
        _, st = O.tf_dynamic_rnn(win, c64, st)
        (pi, us, sg, rh), _ = O.tf_gmm3d_head(st[-1, 1], h64)
        ref = np.concatenate([pi, us, sg, rh], 1)
        got = params_gpu.cpu().numpy()
        assert (np.abs(got - ref) <= 1e-3 * np.abs(ref) + 5e-5).all(), (k, np.abs(got - ref).max())
        np.testing.assert_allclose(st_gpu.cpu().numpy(), st, atol=5e-5)
        smp_ref = O.sample_mixture_3d((pi, us, sg, rh), u[k].astype(np.float64), z[k].astype(np.float64))      # (B,1,90)
        smp = ops.gmm3d_sample(params_gpu, dev(u[k]), dev(z[k])).cpu().numpy()
        near = (np.abs(np.cumsum(pi, 1)[:, None, :] - u[k].astype(np.float64)[:, :, None]).min(-1) < 1e-4)   # (B,fps)
        n_skipped += int(near.sum())
        d = np.abs(smp - smp_ref[:, 0]).reshape(B, fps, 3).max(-1)
        scale = np.abs(smp_ref).max()
        assert (d[~near] <= 1e-3 * scale).all(), (k, d[~near].max())
        win = np.concatenate([win[:, 1:], smp_ref], 1)
    assert n_skipped <= B * fps * P // 100
    # the object's loop = the same entry points chained on the device (bit-identical to chaining them by hand)
    outs, st_loop = tr.rollout_gmm(dev(x), dev(init), dev(u), dev(z))
    wing, stg = dev(x), dev(init)
    for k in range(P):
        pg, _, stg = tr.predict(wing, stg)
        nxt = ops.gmm3d_sample(pg, dev(u[k]), dev(z[k]))
        assert torch.equal(outs[k], nxt)
        wing = nxt[:, None, :].contiguous()
    assert torch.equal(st_loop, stg)
    assert outs.shape == (P, B, F) and bool(torch.isfinite(outs).all())
