"""usage: python tools/bisect_check.py <lib.so>: layer (H=128/256, hs) and fused decode against the fp64 oracle."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from longterm360fov_amd import _lib  # noqa: E402

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
for H in (64, 128, 256):
    F, B, T = 90, 37, 5
    rng = np.random.default_rng(100 + H + F + B)
    K, R, b = O.init_lstm(rng, F, H, np.float32)
    x = rng.uniform(-1, 1, (B, T, F)).astype(np.float32)
    ref = O.lstm_layer(x.astype(np.float64), K.astype(np.float64), R.astype(np.float64), b.astype(np.float64), None, None, "sigmoid")
    ws = ops.Workspace()
    hs, hT, cT = ops.lstm_seq(dev(x), dev(K), dev(R), dev(b), None, None, act="sigmoid", impl="cluster", workspace=ws)
    ws.check()
    e = np.abs(hs.cpu().numpy() - ref[0])
    print("layer H=%d: max err %.3e per step %s" % (H, e.max(), np.array2string(e.max(axis=(0, 2)), precision=2)))
for B in (48, 1024):
    w = O.init_seq2seq(1, H=256, bias_noise=0.05)
    enc, dec0, _ = O.synthetic_batch(2, B, 6, 5)
    dw = {k: dev(v) for k, v in w.items()}
    ws = ops.Workspace()
    out = ops.seq2seq_decode(dev(enc), dev(dec0), dw, 5, impl="cluster", workspace=ws).cpu().numpy()
    ws.check()
    ref = O.seq2seq_decode(enc[:48].astype(np.float64), dec0[:48].astype(np.float64), {k: v.astype(np.float64) for k, v in w.items()}, 5)
    e = np.abs(out[:48] - ref)
    print("decode B=%d: max err %.3e per step %s" % (B, e.max(), np.array2string(e.max(axis=(0, 2)), precision=2)))
