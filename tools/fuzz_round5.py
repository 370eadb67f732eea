#!/usr/bin/env python3
"""Random-shape checks of the round-5 kernels against fp64 NumPy (run on the GPU box; not part of the test suite):
fov_lstm_seq_wgrad_pair (wgrad_rows_kernel / fallbacks), fov_conv2d_wgrad (conv_wgrad_lines.hip / tap-wise kernel),
fov_dense_mse_head (one and two launches).  usage: python3 tools/fuzz_round5.py [cases per kernel] [seed]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
dev = lambda a: None if a is None else torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
worst = {"wgrad_pair": 0.0, "conv_wgrad": 0.0, "dense_head": 0.0}

for case in range(n_cases):
    H = int(rng.choice([16, 32, 48, 64, 96, 128, 160, 256, 400 // 16 * 16, 512]))
    B = int(rng.integers(1, 70)); T1 = int(rng.integers(1, 24)); T2 = int(rng.integers(1, 24))
    F1 = int(rng.integers(1, 100)); F2 = int(rng.choice([3, 6, 7, 64, H]))
    u = lambda *s: rng.uniform(-1, 1, s).astype(np.float32)
    x1, hs1, dz1, x2, hs2, dz2 = u(B, T1, F1), u(B, T1, H), u(B, T1, 4 * H), u(B, T2, F2), u(B, T2, H), u(B, T2, 4 * H)
    h01 = u(B, H) if rng.random() < 0.5 else None
    h02 = u(B, H) if rng.random() < 0.7 else None
    def ref(x, hs, h0, dz):
        x, hs, dz = x.astype(np.float64), hs.astype(np.float64), dz.astype(np.float64)
        hp = np.concatenate([(h0.astype(np.float64) if h0 is not None else np.zeros((x.shape[0], H)))[:, None], hs[:, :-1]], 1)
        return np.einsum("btf,btn->fn", x, dz), np.einsum("bth,btn->hn", hp, dz), dz.sum((0, 1))
    r = ref(x1, hs1, h01, dz1) + ref(x2, hs2, h02, dz2)
    pad = int(rng.integers(0, 2)) * 4      # sometimes 16-byte aligned pieces, sometimes not
    sizes = [F1 * 4 * H, H * 4 * H, 4 * H, F2 * 4 * H, H * 4 * H, 4 * H]
    offs = np.concatenate([[pad], pad + np.cumsum(sizes)])
    flat = torch.zeros(int(offs[-1]) + 4, device="cuda")
    shp = [(F1, 4 * H), (H, 4 * H), (4 * H,), (F2, 4 * H), (H, 4 * H), (4 * H,)]
    v = [flat[int(offs[i]):int(offs[i + 1])].view(*shp[i]) for i in range(6)]
    ops.lstm_seq_wgrad_pair((dev(x1), dev(hs1), dev(h01), dev(dz1)) + tuple(v[:3]), (dev(x2), dev(hs2), dev(h02), dev(dz2)) + tuple(v[3:]), scratch=ops.Scratch())
    torch.cuda.synchronize()
    for got, want in zip(v, r):
        e = np.abs(got.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-6)
        worst["wgrad_pair"] = max(worst["wgrad_pair"], e)
        assert e <= 3e-5, ("wgrad_pair", B, T1, T2, F1, F2, H, e)

for case in range(n_cases):
    k = int(rng.choice([3, 5])); Hh = int(rng.integers(1, 40)); Ww = int(rng.integers(1, 40)); B = int(rng.integers(1, 9))
    C = int(rng.choice([1, 3, 8, 16, 30, 32, 56, 100, 130])); N = int(rng.choice([1, 4, 12, 16, 30, 32, 64, 128, 200]))
    extra = int(rng.choice([0, 2, 8]))
    wide = rng.standard_normal((B, Hh, Ww, C + extra)).astype(np.float32)
    x = dev(wide)[..., :C]
    dy = rng.standard_normal((B, Hh, Ww, N)).astype(np.float32)
    h = k // 2
    xp = np.pad(wide[..., :C].astype(np.float64), ((0, 0), (h, h), (h, h), (0, 0)))
    want = np.zeros((k, k, C, N))
    for i in range(k):
        for j in range(k):
            want[i, j] = np.einsum("bhwc,bhwn->cn", xp[:, i:i + Hh, j:j + Ww], dy.astype(np.float64))
    got = ops.conv2d_wgrad(x, dev(dy), k, k, scratch=ops.Scratch())
    torch.cuda.synchronize()
    e = np.abs(got.cpu().numpy() - want).max() / (np.abs(want).max() + 1e-6)
    worst["conv_wgrad"] = max(worst["conv_wgrad"], e)
    assert e <= 3e-5, ("conv_wgrad", B, Hh, Ww, C, N, k, e)

for case in range(n_cases):
    N = int(rng.choice([1, 63, 64, 65, 320, 4095, 4097, 8192, 20000])) if case % 2 else int(rng.integers(1, 9000))
    H = int(rng.choice([4, 32, 100, 128, 256, 400, 512])); O = int(rng.integers(1, 9)); act = "tanh" if rng.random() < 0.7 else None
    hs = rng.uniform(-1, 1, (N, H)).astype(np.float32)
    W = (rng.standard_normal((H, O)) / np.sqrt(H)).astype(np.float32); b = (0.1 * rng.standard_normal(O)).astype(np.float32)
    tg = rng.uniform(-1, 1, (N, O)).astype(np.float32)
    pre = hs.astype(np.float64) @ W.astype(np.float64) + b
    y = np.tanh(pre) if act else pre
    d = y - tg
    dpre = 2.0 * d / (N * O) * ((1 - y ** 2) if act else 1.0)
    want = [dpre @ W.astype(np.float64).T, hs.astype(np.float64).T @ dpre, dpre.sum(0), np.array([np.mean(d ** 2)])]
    flat = torch.zeros(H * O + O + 1, device="cuda")
    yg, dX, loss = ops.dense_mse_head(dev(hs), dev(W), dev(b), dev(tg), act, dW=flat[:H * O].view(H, O), db=flat[H * O:H * O + O], loss=flat[H * O + O:], scratch=ops.Scratch())
    torch.cuda.synchronize()
    for got, w in zip((dX, flat[:H * O].view(H, O), flat[H * O:H * O + O], flat[H * O + O:]), want):
        e = np.abs(got.cpu().numpy() - w).max() / (np.abs(w).max() + 1e-12)
        worst["dense_head"] = max(worst["dense_head"], e)
        assert e <= 5e-5, ("dense_head", N, H, O, act, e)
    assert np.abs(yg.cpu().numpy() - y).max() <= 3e-6
print("ok: %d cases per kernel; worst relative errors %s" % (n_cases, {k_: "%.1e" % v_ for k_, v_ in worst.items()}))
