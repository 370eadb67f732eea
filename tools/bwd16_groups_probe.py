"""Width-512 BPTT recurrence of ONE layer at lstm.py's batch (B = 32, T = 2 ... 20): per-step and per-launch cost with 16 or 32
workgroups per tile (FOV_BWD16_GROUPS=32).  usage: [FOV_BWD16_GROUPS=32] python tools/bwd16_groups_probe.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from longterm360fov_amd import ops  # noqa: E402
from oracle import fov_oracle as O  # noqa: E402

B, F, H = 32, 90, 512
rng = np.random.default_rng(0)
d = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
K, R, b = (d(a) for a in O.init_lstm(rng, F, H))
sc = ops.Scratch()
pts = []
for T in (2, 6, 10, 20):
    x = d(rng.uniform(-1, 1, (B, T, F)))
    hs, hT, cT, res = ops.lstm_seq_train(x, K, R, b)
    dhs = d(0.1 * rng.standard_normal((B, T, H)))
    dz = torch.empty((B, T, 4 * H), device="cuda")
    run = lambda: ops.lstm_seq_bwd(x, K, R, hs, res, dhs=dhs, need_state_grads=True, scratch=sc, dz=dz, need_weight_grads=False)
    for _ in range(10):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200):
        run()
    e1.record()
    torch.cuda.synchronize()
    pts.append((T, e0.elapsed_time(e1) / 200 * 1e3))
sc.check()
a, c = np.polyfit([p[0] for p in pts], [p[1] for p in pts], 1)
print("FOV_BWD16_GROUPS=%s: %s us -> %.2f us per step + %.1f us per launch" % (os.environ.get("FOV_BWD16_GROUPS", "16"), ["%.1f" % p[1] for p in pts], a, c))
