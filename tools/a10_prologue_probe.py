"""How much of lstm.py's two-layer forward launch (fov_lstm_stack2_fwd at B = 32, F = 90, H = 512) is its prologue?  Times the
launch at several sequence lengths: the intercept of the line is what a launch costs before its first step.
usage: python tools/a10_prologue_probe.py [batch]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

B, F, H = (int(sys.argv[1]) if len(sys.argv) > 1 else 32), 90, 512
rng = np.random.default_rng(0)
layers = [O.init_lstm(rng, F, H), O.init_lstm(rng, H, H)]
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
dl = [tuple(d(a) for a in l) for l in layers]
ws = ops.Workspace()
pts = []
for T in (2, 4, 6, 10, 14, 20):
    x = d(rng.uniform(-1, 1, (B, T, F)))
    if not ops.lstm_stack2_supported(B, T, F, H):
        print("T=%d: not supported" % T)
        continue
    for _ in range(20):
        ops.lstm_stack2(x, dl[0], dl[1], workspace=ws)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(300):
        ops.lstm_stack2(x, dl[0], dl[1], workspace=ws)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 300
    pts.append((T, ms * 1e3))
    print("T=%2d  %.1f us per launch" % (T, ms * 1e3))
ws.check()
t = np.array([p[0] for p in pts], dtype=np.float64)
u = np.array([p[1] for p in pts], dtype=np.float64)
a, b = np.polyfit(t, u, 1)
print("fit: %.2f us per step (two layers, one step apart) + %.1f us per launch" % (a, b))
